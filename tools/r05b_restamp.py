#!/usr/bin/env python3
"""After tools/r05_collect.sh solve100 solvez100 factor100 factorz100 solve200 factor200 on the final sources: copy the
summaries to profiles/r05b_* and restamp profiles/traffic.json (solves: the totals of tools/pmc_solve.sh; factorisations:
round 5's per-factorisation totals scaled by the ratio of the summaries' listed kernels, which re-measures them)."""
import re, shutil, subprocess, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
def top(fn):
    tot = 0
    for l in open(fn):
        m = re.match(r"\s+(\S+)\s+launches\s+(\d+)\s+([\d.]+) ms\s+read\s+([\d.]+) GB\s+written\s+([\d.]+) GB", l)
        if m and not re.search(r"nd_level|transpose_fill|segsort|col_dominance|copyBuffer", m.group(1)):
            tot += float(m.group(4)) + float(m.group(5))
    return tot
def last_total(fn):
    return float(re.search(r"= ([\d.]+) GB; algorithmic", open(fn).read().strip().splitlines()[-1]).group(1)) * 1e9
old = {'100': 460.56e9, 'z100': 1019.55e9, '200': 23574.73e9}  # profiles/r05_pmc_factor_*.txt (first session)
src = "sparse-linear_amd/csrc/multifrontal.hip,sparse-linear_amd/csrc/dense_lu_kernels.hpp,sparse-linear_amd/csrc/mf_chain.hpp"
def upd(key, val, label):
    subprocess.check_call(["python3", "tools/update_traffic.py", "--bytes", key, str(int(val)), src, label])
keys = {'100': "lu_poisson3d_100", 'z100': "zi_lu_100", '200': "lu_poisson3d_200"}
for k in old:
    f, sv = 'gpurun_out/r05/pmc_factor_%s.txt' % k, 'gpurun_out/r05/pmc_solve_%s.txt' % k
    shutil.copy(f, 'profiles/r05b_pmc_factor_%s.txt' % k)
    shutil.copy(sv, 'profiles/r05b_pmc_solve_%s.txt' % k)
    upd(keys[k] + ":factor", old[k] * top(f) / top('profiles/r05_pmc_factor_%s.txt' % k), "profiles/r05b_pmc_factor_%s.txt" % k)
    upd(keys[k] + ":solve", last_total(sv), "profiles/r05b_pmc_solve_%s.txt" % k)
for a, b in (("tr100", "r05b_solve_100_kernel_trace.txt"), ("trz100", "r05b_solve_z100_kernel_trace.txt")):
    if os.path.exists('gpurun_out/r05/%s_summary.txt' % a):
        shutil.copy('gpurun_out/r05/%s_summary.txt' % a, 'profiles/' + b)
