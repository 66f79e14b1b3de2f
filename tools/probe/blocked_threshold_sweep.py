#!/usr/bin/env python3
"""Reference-order SpMV on matrices whose rows share no x lines, x = 1 .. 64 MB: from which size on does the column-blocked
lockstep image beat the CSR-stream kernel WHEN IT IS SHAPED FOR THE WHOLE CHIP?  (Round 4's sweep built the blocked image
with the default 16 x 1024-row shape: at n = 2^17 .. 2^21 only 8 .. 128 of the 256 CUs had a workgroup.)
For every n: wavefronts per CU nw in {4, 8, 16}, panel height R = rows / (CUs * nw) (several full generations when the LDS
is too small), x window 2^w columns for w = 13 .. 18; random n x n with 20 draws per row, and the R-MAT matrix of config C4.
usage: python3 tools/probe/blocked_threshold_sweep.py [lg ...]"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
ffi = pkg._ffi
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def time_spmv(H, reps=20):
    inf = H.info()
    n, nnz = inf["nrows_local"], inf["nnz"]
    s = torch.cuda.current_stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    ffi.check("vec", ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), s.cuda_stream))
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(s)
    for _ in range(reps):
        H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    B = 12 * nnz + 4 * (n + 1) + 16 * n
    return ms, B / ms / 1e6, y


def shapes(n):
    for nw in (4, 8, 16):
        rmax = (160 * 1024 // 8) // nw
        slots = CUS * nw
        ngen = (n + slots * rmax - 1) // (slots * rmax)
        R = max(64, (n + ngen * slots - 1) // (ngen * slots))
        yield nw, R


def sweep(make, label):
    H = make()
    H.set_variant(0)
    t_stream, g_stream, y_ref = time_spmv(H)
    H.free()
    print("%s | stream %.4f ms %.0f GB/s" % (label, t_stream, g_stream), flush=True)
    H = make()
    H.optimize()
    t_opt, g_opt, y_opt = time_spmv(H)
    print("    optimize() [reference order]: kernel %d %.4f ms %.0f GB/s bit-identical %s" % (H.spmv_kernel(), t_opt, g_opt, bool(torch.equal(y_opt, y_ref))), flush=True)
    H.free()
    best = None
    for nw, R in shapes(H_n[0]):
        cells = []
        for w in (13, 14, 15, 16, 17, 18):
            if (R << w) > 0x7fffffff:
                continue
            os.environ["SPL_BLOCKED_LOCKSTEP"] = str(nw)
            H = make()
            try:
                H.build_blocked(R, w, 0)
                H.set_variant(8)
                ms, gbs, y = time_spmv(H)
                same = bool(torch.equal(y, y_ref))
                cells.append("w%d %.4f%s" % (w, ms, "" if same else " DIFFERS"))
                if same and (best is None or ms < best[0]):
                    best = (ms, nw, R, w)
            except Exception as e:
                cells.append("w%d %s" % (w, e))
            H.free()
        print("    nw=%2d R=%5d | %s" % (nw, R, " | ".join(cells)), flush=True)
    os.environ.pop("SPL_BLOCKED_LOCKSTEP", None)
    print("    best blocked: %.4f ms (nw=%d R=%d w=%d) vs stream %.4f -> %s" % (best + (t_stream, "blocked" if best[0] < t_stream else "stream")), flush=True)


H_n = [0]
lgs = [int(a) for a in sys.argv[1:]] or [17, 18, 19, 20, 21, 22, 23]
for lg in lgs:
    n = 1 << lg
    H_n[0] = n
    sweep(lambda: pkg.DeviceMatrix.synthetic("random", n, 20), "random n=2^%d x=%.0f MB" % (lg, n * 8 / 2**20))
H_n[0] = 1 << 20
sweep(lambda: pkg.DeviceMatrix.rmat(20, 32, (0.25, 0.25, 0.25)), "R-MAT scale 20 (config C4's matrix) x=8 MB")
