#!/usr/bin/env python3
"""bench.py — fp64 CSR SpMV effective GB/s (and % of the HBM roofline) on N MI355X.

Metric and configuration are BASELINE.json's: config C2 = 1e7 x 1e7 random CSR,
20 draws/row, y = A x in fp64, int32 indices; at N > 1 the rows
are split into N contiguous blocks (one process per GPU), every GPU holds the
full x, and each step ends with an RCCL all-gather of y over xGMI so that y can
be the next x.  A "step" is one SpMV over the whole matrix (plus the all-gather
at N > 1).  Inputs are generated in HBM by counter-based generators
(include/spl_synth.h) and are resident before the timed region starts.

  value     = algorithmic bytes of ONE whole-matrix SpMV / (time per step)   [GB/s]
              B = 12*nnz + 4*(nrows+1) + 8*ncols + 8*nrows   
  roofline  = the SpMV kernel alone on this rank: its own algorithmic bytes per
              launch / average launch duration from HIP events on the launch stream
  cpu_baseline (rank 0, N = 1 only) = the CPU oracle's restatement of the
              reference's mulV (serial CSC scatter, 64-bit indices,
              Sparse.hs:433-471) on the same matrix, 1 core; its result is also
              compared element-wise with the GPU's y (1e-10 relative).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under
python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def spmv_bytes(nnz, nrows, ncols):
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=10_000_000)  # --rows: unambiguous under torchrun
    ap.add_argument("--draws", type=int, default=20)
    ap.add_argument("--matrix", default="random", choices=["random", "banded"])
    ap.add_argument("--variant", type=int, default=0, help="SpMV kernel variant (ablation)")
    ap.add_argument("--blocked", default="", help="force the column-blocked image: rows_per_panel,cols_log2[,unroll]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        if world > 1:
            sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # SPL_BENCH_REHEARSAL=1 (tests only): every rank on cuda:0 and the gloo backend, so that the whole
    # N > 1 flow (row blocks, all-gather of y, max over ranks) can be rehearsed on a one-GPU box.  The
    # numbers of such a run mean nothing; the driver's runs use one GPU per rank over RCCL.
    rehearsal = os.environ.get("SPL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = load_package()
    ffi = pkg._ffi
    n, N = args.n, world
    if n % N:
        sys.exit("n must be divisible by the number of GPUs (equal all-gather slices)")
    r0, r1 = n * rank // N, n * (rank + 1) // N

    # ---- inputs resident in HBM before anything is timed
    H = pkg.DeviceMatrix.synthetic(args.matrix, n, args.draws, seed=0x5EED, row0=r0, row1=r1)
    if args.blocked:
        H.build_blocked(*[int(t) for t in args.blocked.split(",")])
        H.set_variant(8)
    elif args.variant:
        H.set_variant(args.variant)
    else:
        H.optimize()  # one-time analysis (not timed): picks CSR-stream or the column-blocked image
    info = H.info()
    stream = torch.cuda.current_stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    ffi.check("spl_vector_synthetic_dev",
              ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), stream.cuda_stream))
    dist_mod = sys.modules["sparse_linear_amd.dist"]
    op = dist_mod.RowBlockSpMV(n, dist_mod.equal_row_bounds(n, N), rank, N,
                               dist_mod.hip_local_spmv(H, lambda: stream.cuda_stream), "cuda")
    y_local, y_full = op.y_local, op.y_full

    def spmv():
        op.local_spmv(x, y_local)

    def step():
        op.step(x)  # local CSR-stream kernel, then (N > 1) the RCCL all-gather of y

    def barrier():
        if N > 1:
            dist.barrier()

    nnz_t = torch.tensor([info["nnz"]], dtype=torch.int64, device="cuda")
    if N > 1:
        dist.all_reduce(nnz_t)
    nnz_total = int(nnz_t.item())
    B_total = spmv_bytes(nnz_total, n, n)
    B_local = spmv_bytes(info["nnz"], r1 - r0, n)

    # ---- timed region: W warm-ups, then exactly K steps between barrier+sync pairs
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if N > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = B_total / (elapsed / args.steps) / 1e9

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel only
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record(stream)
    for _ in range(args.steps):
        spmv()
    ev1.record(stream)
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    achieved = B_local / (kern_ms * 1e-3) / 1e9
    info = H.info()
    kernel = ("spmv_blocked_lockstep" if info["blocked_rows"] > 0 else
              "spmv_sell" if info["blocked_rows"] < 0 else "spmv_stream")
    traffic = None  # HBM bytes per launch from the committed rocprofv3 PMC passes (N = 1, default sizes only)
    tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if N == 1 and n == 10_000_000 and args.draws == 20 and os.path.exists(tfile):
        traffic = json.load(open(tfile)).get("%s:%s" % (args.matrix, kernel))
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "kernel": kernel, "kernel_ms": round(kern_ms, 4), "bytes_per_launch": B_local}
    if info["blocked_rows"] > 0:
        roofline["image"] = "column-blocked: %d rows/panel, 2^%d columns/block" % (info["blocked_rows"], info["blocked_cols_log2"])

    out = {
        "metric": "fp64 CSR SpMV effective GB/s", "value": round(value, 1), "unit": "GB/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s CSR %dx%d, %d draws/row, nnz=%d, y=A*x fp64, int32 indices%s"
                               % (args.matrix, n, n, args.draws, nnz_total,
                                  "" if N == 1 else ", %d row blocks + RCCL all-gather of y" % N),
                   "algorithmic_bytes": B_total, "hbm_frac_of_%dx8TBps" % N: round(value / (N * HBM_PEAK_GBPS), 4),
                   "variant": args.variant, "blocked": args.blocked or "auto"},
        "roofline": roofline,
    }

    # ---- CPU baseline beside it (rank 0, N = 1): the reference's own algorithm, 1 core
    if N == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O
        y_gpu = y_local.cpu().numpy()
        cp, ri, v = H.export_csc()  # CSC(A): the reference's Matrix fields
        A = (n, n, cp, ri.astype(np.int64), v)  # 64-bit Int indices as in the reference
        xh = x.cpu().numpy()
        times = []
        y_cpu = None
        for _ in range(max(1, args.cpu_reps)):
            t = time.perf_counter()
            y_cpu = O.mulV(A, xh)  # thaw x, zero y, axpy_, freeze y  (Sparse.hs:464-471)
            times.append(time.perf_counter() - t)
        t_cpu = sorted(times)[len(times) // 2]
        bad = O.count_not_close(y_gpu, y_cpu, 1e-10)
        out["cpu_baseline"] = {"value": round(B_total / t_cpu / 1e9, 3), "unit": "GB/s", "cores": 1,
                               "kind": "port",
                               "sample": "whole %s matrix (nnz=%d), median of %d serial CSC-scatter mulV runs "
                                         "(oracle restatement of Sparse.hs:433-471, 64-bit indices), %.2f s each"
                                         % (args.matrix, nnz_total, len(times), t_cpu)}
        out["parity"] = {"checked": int(n), "not_close_1e-10": int(bad),
                         "bit_identical": bool(np.array_equal(y_gpu, y_cpu))}
        # a fair CPU (NOT the reference): OpenMP CSR gather on all host cores
        rp, ci, vv = H.export_csr()
        rp32 = rp.astype(np.int32)
        yo = np.zeros(n)
        O.csr_spmv_omp(rp32, ci, vv, xh, yo)
        t = time.perf_counter()
        O.csr_spmv_omp(rp32, ci, vv, xh, yo)
        t_omp = time.perf_counter() - t
        out["cpu_fair_openmp"] = {"value": round(B_total / t_omp / 1e9, 3), "unit": "GB/s",
                                  "cores": O.omp_threads(), "note": "not the reference: OpenMP CSR gather, int32"}

    if rank == 0:
        print(json.dumps(out))
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
