#!/usr/bin/env python3
"""Rewrites the solve-ladder tables of DESIGN.md §4.5 from the JSON lines in profiles/ (so that the
document and the committed measurements cannot drift apart)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def steady(g):
    """factor time with the memory coming from the library's pool (the same matrix factored again) and the
    total built on it; the first factorisation of a size is in the JSON as factor_s / total_s"""
    f = g.get("refactor_s", g["factor_s"])
    return f, g["analyze_s"] + f + g["solve_s"]


def fmt(x):
    return "%.3f" % x if x < 1 else "%.2f" % x if x < 10 else "%.1f" % x


def main():
    P = os.path.join(ROOT, "profiles")
    rows3 = [json.loads(l) for l in open(os.path.join(P, "r01_solve_ladder_poisson3d.json"))]
    rows2 = [json.loads(l) for l in open(os.path.join(P, "r01_solve_ladder_poisson2d.json"))]
    band = {json.loads(l)["m"]: json.loads(l) for l in open(os.path.join(P, "r01_solve_ladder_poisson3d_band.json"))}
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for d in rows3:
        m, g = d["m"], d["gpu"]
        if m < 32:
            continue
        name = r"\*\*200³ \(config C5\)\*\*" if m == 200 else "%d³" % m
        label = "**200³ (config C5)**" if m == 200 else "%d³" % m
        b = ("%s s" % fmt(steady(band[m]["gpu"])[1]) + (" (%.0f GB)" % band[m]["factorisation"]["device_GB"] if m == 100 else "")) if m in band else "does not fit"
        fac, total = steady(g)
        tot = "**%s s**" % fmt(total) if m == 200 else "%s s" % fmt(total)
        new = "| %s | %s | %s | %s | %s | %s | %s | %.1e |\n" % (
            label, format(d["n"], ",").replace(",", " "), fmt(g["analyze_s"]), fmt(fac), fmt(g["solve_s"]), tot, b,
            d["max_rel_err_vs_manufactured"])
        s, k = re.subn(r"\| %s \| [^\n]*\n" % name, lambda _m: new, s, count=1)
        assert k == 1, m
    for d in rows2:
        m, g = d["m"], d["gpu"]
        fac, total = steady(g)
        new = "| %d² | %s | %s | %s | %s | %s s | %.1e |\n" % (
            m, format(d["n"], ",").replace(",", " "), fmt(g["analyze_s"]), fmt(fac), fmt(g["solve_s"]),
            fmt(total), d["max_rel_err_vs_manufactured"])
        s, k = re.subn(r"\| %d² \| [^\n]*\n" % m, lambda _m: new, s, count=1)
        assert k == 1, m
    c5 = [d for d in rows3 if d["m"] == 200][0]["gpu"]
    open(path, "w").write(s)
    print("C5:", c5)


if __name__ == "__main__":
    main()
