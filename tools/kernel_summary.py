#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results .db (kernel-trace): name, calls, avg us, total ms, %."""
import sqlite3
import sys


def main(path, top=12):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = db.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 "
        f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows) or 1.0
    for r in rows[:top]:
        print(f"{r[0][:72]:72s} calls={r[1]:6d} avg_us={r[2]:9.1f} total_ms={r[3]:9.1f} pct={100 * r[3] / tot:5.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
