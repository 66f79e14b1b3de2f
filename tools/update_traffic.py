#!/usr/bin/env python3
"""Write profiles/traffic.json: HBM/fabric bytes per launch of an SpMV kernel from a rocprofv3 PMC
summary (tools/profile_spmv.sh), keyed "<matrix>:<kernel>", together with the SHA-1 of the kernel's
source file, so that bench.py only reports the figure for the code it was measured on.
usage: tools/update_traffic.py <matrix>:<kernel> <source file> <summary.txt> [<label for "from">]
       tools/update_traffic.py --bytes <key> <traffic bytes> <source file>[,<source file>...] <label for "from">
         (the secondary configurations of bench.py: HBM bytes of one solve call / factorisation / product, summed over its
         kernels by tools/pmc_solve.sh, pmc_factor.sh, profile_spgemm.sh; tied to the SHA-1 of ALL the listed sources)"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if sys.argv[1] == "--bytes":
        key, nbytes, sources, label = sys.argv[2], int(float(sys.argv[3])), sys.argv[4].split(","), sys.argv[5]
        h = hashlib.sha1()
        for rel in sources:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        path = os.path.join(ROOT, "profiles", "traffic.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = {"traffic_bytes": nbytes, "sources": sources, "sources_sha1": h.hexdigest(), "from": label}
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
        print(key, nbytes)
        return
    key, source, summary = sys.argv[1:4]
    label = sys.argv[4] if len(sys.argv) > 4 else os.path.relpath(summary, ROOT)
    text = open(summary).read()

    def mean(name):
        m = re.search(r"%s\s+n=\d+ mean=([0-9.e+]+)" % re.escape(name), text)
        return float(m.group(1)) if m else None
    rd = mean("TCC_EA0_RDREQ_sum")
    wr_kb = mean("WRITE_SIZE")
    if rd is None or wr_kb is None:
        sys.exit("summary lacks TCC_EA0_RDREQ_sum / WRITE_SIZE")
    # every read request is a full 128-byte line on this path (TCC_EA0_RDREQ_32B/64B ~ 0); WRITE_SIZE is in KB
    traffic = int(rd * 128 + wr_kb * 1024)
    path = os.path.join(ROOT, "profiles", "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = {"traffic_bytes": traffic, "source": source,
                 "source_sha1": hashlib.sha1(open(os.path.join(ROOT, source), "rb").read()).hexdigest(), "from": label}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(key, traffic)


if __name__ == "__main__":
    main()
