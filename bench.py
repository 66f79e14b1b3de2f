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

  secondary (N = 1 only, after the timed region; --no-secondary skips it) = the other configurations of
              BASELINE.json the driver's one line can carry: the banded variant of C2, C4 (SpGEMM A*A, R-MAT
              scale 20) and the C5 ladder (sparse LU + solves, 100^3 and, memory permitting, 200^3), each with
              its own roofline and parity fields (tools/bench_secondary.py).

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 either under
python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... (one rank per GPU), or plainly as
python bench.py --gpus N: without WORLD_SIZE in the environment the script starts that launcher itself, as a
child process and before anything touches a GPU, and exits with its status.
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
    ap.add_argument("--order", default="free", choices=["free", "reference"],
                    help="order of the floating-point sums: free = any order (1e-10 contract, column-sorted panel "
                         "kernel where it pays), reference = the reference's order, bit-identical (spl_matrix_set_spmv_order)")
    ap.add_argument("--panel", default="", help="force the column-sorted panel image: rows_per_panel,cols_log2[,unroll[,form]]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true", help="skip the banded / C4 / C5 block after the timed region")
    ap.add_argument("--secondary", default="c5:200:cpu32,banded,spmv:poisson3d:200,spmv:rmat:20,c4,c5:100,zi:100,feast:80",
                    help="which secondary configurations to run, in this order, each in a child process of its own (comma "
                         "separated: banded, c4[:<scale>], c5:<m>[:cpu<ms>], zi:<m>, feast:<m>, spmv:poisson3d:<m>, spmv:rmat:<scale>); the "
                         "200^3 factorisation comes first: it asks the driver for 255 GB, and memory other configurations have "
                         "just released is still being wiped in the background (DESIGN.md, Device memory)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one rank per GPU ourselves.  A fresh child process runs the
        # launcher; this parent has not imported torch or touched a GPU, and only passes the child's status on.
        import socket
        import subprocess
        port = os.environ.get("MASTER_PORT")
        if not port:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    # dmabuf IPC is the only kind this platform's host driver supports: without it hipIpcGetMemHandle fails ("invalid
    # argument") — for RCCL's own buffers as for the one-sided exchange of csrc/peer.hip.  The pool's boxes export it
    # already; a box that does not gets it here, before anything initialises HIP (a value that is set is left alone).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
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
    elif args.panel:
        H.build_panel(*[int(t) for t in args.panel.split(",")])
        H.set_variant(16)
    elif args.variant:
        H.set_variant(args.variant)
    else:
        if args.order == "free":
            H.set_spmv_order(H.ORDER_FREE)
        H.optimize()  # one-time analysis (not timed): picks CSR-stream, sliced ELL, column-blocked or panel image
    info = H.info()
    stream = torch.cuda.current_stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    ffi.check("spl_vector_synthetic_dev",
              ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), stream.cuda_stream))
    dist_mod = sys.modules["sparse_linear_amd.dist"]
    op = dist_mod.RowBlockSpMV(n, dist_mod.equal_row_bounds(n, N), rank, N,
                               dist_mod.hip_local_spmv(H, lambda: stream.cuda_stream), "cuda")
    y_local, y_full = op.y_local, op.y_full

    def barrier():
        if N > 1:
            dist.barrier()

    def agree(flag, how):  # the same decision on every rank
        t = torch.tensor([float(flag)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=how)
        return float(t.item())

    def seconds_per_step(fn, warm, reps):
        for _ in range(warm):
            fn()
        barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        barrier()
        return agree((time.perf_counter() - t) / reps, dist.ReduceOp.MAX)

    # ---- N > 1: how y is exchanged is chosen by measurement before anything is timed.  Two kinds —
    # "rccl": all_gather_into_tensor; "peer": one-sided device-to-device copies into the peers' buffers
    # (csrc/peer.hip: no collective kernel takes CUs from the SpMV) — each with the rank's rows in one piece
    # or cut into 2 or 4 chunks whose exchange runs under the kernel of the next chunk
    # (dist.PipelinedRowBlockSpMV / dist.PeerStoreRowBlockSpMV).  A candidate must reproduce the plain step's y
    # bit for bit on every rank (to 1e-10 relative under --order free, whose sums may differ at rounding level
    # between launches) and is kept only if it is faster.  SPL_BENCH_CHUNKS=k forces the number of chunks
    # (1 = one piece), SPL_BENCH_EXCHANGE=rccl|peer the kind.
    exchange = "one RCCL all-gather of y after the kernel"
    H0 = H  # the rank's whole row block
    handles = [H]
    kernels = [dist_mod.hip_local_spmv(H, lambda: stream.cuda_stream)]
    pieces = [(r0, r1)]
    tuning = None
    if N > 1 and not args.blocked and not args.variant:
        forced = int(os.environ.get("SPL_BENCH_CHUNKS", "0"))
        want = os.environ.get("SPL_BENCH_EXCHANGE", "")
        chunk_counts = [c for c in ([forced] if forced >= 1 else [1, 2, 4]) if n % (c * N) == 0]
        kinds = [want] if want in ("rccl", "peer") else ["rccl", "peer"]
        y_plain = op.step(x).clone()
        best_t = seconds_per_step(lambda: op.step(x), 3, 10)
        best_key = "rccl1"
        tuning = {"rccl1": round(best_t * 1e3, 4)}
        have_allowed = "rccl" in kinds and 1 in chunk_counts  # is the plain step itself a permitted choice?
        piece_handles = {1: [H0]}  # per chunk count: the device matrices of this rank's pieces
        winner_handles = [H0]

        def describe(kind, C):
            if kind == "rccl":
                return ("one RCCL all-gather of y after the kernel" if C == 1 else
                        "%d chunks per rank, the RCCL all-gather of a chunk under the kernel of the next" % C)
            return ("one-sided peer stores after the kernel: a device-to-device copy per peer and stream, step flags" if C == 1 else
                    "%d chunks per rank, one-sided peer stores of a chunk under the kernel of the next" % C)

        for C in chunk_counts:
            for kind in kinds:
                key = "%s%d" % (kind, C)
                if key == "rccl1":
                    continue  # the plain step, measured above
                cand, ok = None, 1.0
                try:
                    if C not in piece_handles:
                        bnd = dist_mod.pipelined_piece_bounds(n, N, C)
                        hs = []
                        for c in range(C):
                            q = c * N + rank
                            h = pkg.DeviceMatrix.synthetic(args.matrix, n, args.draws, seed=0x5EED, row0=bnd[q], row1=bnd[q + 1])
                            if args.order == "free":
                                h.set_spmv_order(h.ORDER_FREE)
                            h.optimize()
                            hs.append(h)
                        piece_handles[C] = hs
                    hs = piece_handles[C]
                    spmvs = [dist_mod.hip_local_spmv(h, lambda: stream.cuda_stream) for h in hs]
                    if kind == "peer":
                        cand = dist_mod.PeerStoreRowBlockSpMV(n, rank, N, C, spmvs, "cuda", lambda: stream.cuda_stream)
                        if not cand.flags_finegrained() and want != "peer":
                            # flags a kernel polls while a peer's copy engine writes them belong in fine-grained
                            # memory; without it the one-sided exchange only runs when asked for by name
                            cand.close()
                            raise RuntimeError("no fine-grained device memory for the step flags")
                    elif C == 1:
                        cand = op
                    else:
                        cand = dist_mod.PipelinedRowBlockSpMV(n, rank, N, C, spmvs, "cuda")
                except Exception as e:  # e.g. out of memory or no peer access on one rank: every rank drops the candidate
                    sys.stderr.write("rank %d: exchange %s not available: %s\n" % (rank, key, e))
                    ok = 0.0
                if agree(ok, dist.ReduceOp.MIN) < 1.0:
                    continue
                y_c = cand.step(x)
                torch.cuda.synchronize()
                if args.order == "free":  # order-free sums: rounding-level differences between launches are legitimate
                    same = float(((y_c - y_plain).abs() <= 1e-10 * (y_c + y_plain).abs()).all().item())
                else:
                    same = float(torch.equal(y_c, y_plain))
                if kind == "peer" and cand.failed():
                    same = 0.0
                if agree(same, dist.ReduceOp.MIN) < 1.0:
                    if rank == 0:
                        sys.stderr.write("exchange %s differs from the plain step: dropped\n" % key)
                    if kind == "peer":
                        cand.close()
                    continue
                t_c = seconds_per_step(lambda: cand.step(x), 3, 10)
                tuning[key] = round(t_c * 1e3, 4)
                if t_c < best_t or not have_allowed:  # (the plain step may be excluded by the forcing variables)
                    have_allowed = True
                    if hasattr(op, "close") and op is not cand:
                        op.close()
                    best_t, best_key = t_c, key
                    op, winner_handles = cand, hs
                    kernels = spmvs
                    pieces = [cand.rows_of(c) for c in range(C)] if C > 1 else [(r0, r1)]
                    exchange = describe(kind, C)
                elif kind == "peer":
                    cand.close()
        for C, hs in piece_handles.items():  # free the pieces no schedule uses any more
            if hs is not winner_handles and C != 1:
                for h in hs:
                    h.free()
        handles = winner_handles
        del y_plain
        if rank == 0:  # on stderr, so that a run that fails later can still be diagnosed from its tail
            sys.stderr.write("exchange of y, ms per step before the timed region: %s -> %s (%s)\n"
                             % (", ".join("%s %.4f" % kv for kv in tuning.items()), best_key, exchange))
            sys.stderr.flush()
    y_pieces = op.y_local if isinstance(op.y_local, list) else [op.y_local]

    def spmv():  # the kernel(s) of one step alone
        for k, yp in zip(kernels, y_pieces):
            k(x, yp)

    def step():
        op.step(x)  # local kernel(s) and, at N > 1, the RCCL all-gather(s) of y

    info = handles[0].info()
    nnz_local = sum(h.info()["nnz"] for h in handles)
    nnz_t = torch.tensor([nnz_local], dtype=torch.int64, device="cuda")
    if N > 1:
        dist.all_reduce(nnz_t)
    nnz_total = int(nnz_t.item())
    B_total = spmv_bytes(nnz_total, n, n)
    B_local = sum(spmv_bytes(h.info()["nnz"], b - a, n) for h, (a, b) in zip(handles, pieces))

    # ---- timed region: W warm-ups, then exactly K steps between barrier+sync pairs
    def timed_region():
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
        el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
        if N > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    elapsed = timed_region()
    # one-sided exchange: a bounded wait that gave up during the warm-up or the timed steps leaves an incomplete y
    # behind (its flag is sticky).  Such a run is not a result: every rank goes back to the plain step — the whole
    # row block, one RCCL all-gather — and the region is timed again.
    exchange_failed = 0.0
    fell_back = False
    if N > 1:
        exchange_failed = agree(1.0 if (hasattr(op, "failed") and op.failed()) else 0.0, dist.ReduceOp.MAX)
        if exchange_failed:
            if rank == 0:
                sys.stderr.write("the one-sided exchange timed out during the timed region: falling back to the RCCL all-gather\n")
            barrier()
            op.close()
            if handles[0] is not H0:
                for h in handles:
                    h.free()
            handles, pieces = [H0], [(r0, r1)]
            kernels = [dist_mod.hip_local_spmv(H0, lambda: stream.cuda_stream)]
            op = dist_mod.RowBlockSpMV(n, dist_mod.equal_row_bounds(n, N), rank, N, kernels[0], "cuda")
            y_pieces = [op.y_local]
            exchange = "one RCCL all-gather of y after the kernel (fallback: the one-sided exchange timed out)"
            fell_back = True
            nnz_local = H0.info()["nnz"]
            B_local = spmv_bytes(nnz_local, r1 - r0, n)
            info = H0.info()
            elapsed = timed_region()
            exchange_failed = 0.0
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
    kcode = handles[0].spmv_kernel()
    kernel = {8: "spmv_blocked_lockstep", 15: "spmv_sell", 16: "spmv_panel"}.get(kcode, "spmv_stream")
    # HBM/fabric bytes per launch from the committed rocprofv3 PMC passes (N = 1, default sizes only).  The
    # figure belongs to the kernel SOURCE it was measured on: profiles/traffic.json carries the SHA-1 of that
    # file (tools/update_traffic.py writes both), and a figure whose source has changed since is not reported.
    traffic, traffic_note = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if N == 1 and n == 10_000_000 and args.draws == 20 and os.path.exists(tfile):
        import hashlib
        ent = json.load(open(tfile)).get("%s:%s" % (args.matrix, kernel))
        if ent:
            src = os.path.join(ROOT, ent["source"])
            now = hashlib.sha1(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
            if now == ent["source_sha1"]:
                traffic = ent["traffic_bytes"]
            else:
                traffic_note = "stale: %s changed since %s was collected" % (ent["source"], ent["from"])
    # What the memory pipeline allows for this mix.  The L2 takes one request per channel and
    # clock: 263 G requests/s chip-wide whatever their size (tools/probe/gather_probe2, row A) — request_bound_ms
    # is the kernel's L2 requests (x gathers that do not share a line + the lines of the matrix stream) at that
    # rate, i.e. the floor if the stream overlapped the gathers perfectly.  It does not: a CU's HBM stream and
    # its L2 gathers share the vector L1's miss slots and their times add up (tools/probe/tcp_mix_probe, every
    # cache policy: tcp_policy_probe) — slot_model_ms = gather requests / 263e9 + stream bytes / 6.5e12.
    request_bound_ms = slot_model_ms = None
    if kernel in ("spmv_panel", "spmv_blocked_lockstep") and N == 1:
        per_entry = 0.75 if kernel == "spmv_panel" else 1.0  # gather requests per stored entry (PMC: TCP_TCC_READ_REQ)
        request_bound_ms = round(1e3 * (per_entry * nnz_local + 12.0 * nnz_local / 128.0) / 263e9, 4)
        slot_model_ms = round(1e3 * (per_entry * nnz_local / 263e9 + 12.0 * nnz_local / 6.5e12), 4)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "kernel": kernel, "kernel_ms": round(kern_ms, 4), "bytes_per_launch": B_local,
                "request_bound_ms": request_bound_ms, "slot_model_ms": slot_model_ms}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    if kcode == 8:
        roofline["image"] = "blocked %d rows x 2^%d cols" % (info["blocked_rows"], info["blocked_cols_log2"])
    elif kcode == 16:
        roofline["image"] = "panels %d rows x 2^%d cols" % (info["blocked_rows"], info["blocked_cols_log2"])

    out = {
        "metric": "fp64 CSR SpMV effective GB/s", "value": round(value, 1), "unit": "GB/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s CSR %dx%d, %d draws/row, nnz=%d%s"
                               % (args.matrix, n, n, args.draws, nnz_total,
                                  "" if N == 1 else ", %d row blocks, exchange of y: %s" % (N, exchange)),
                   "algorithmic_bytes": B_total, "hbm_frac_of_%dx8TBps" % N: round(value / (N * HBM_PEAK_GBPS), 4),
                   "sum_order": args.order},
        "roofline": roofline,
    }
    if args.variant or args.blocked or args.panel:  # forced kernel / image (ablations): said only when not the default choice
        out["config"]["forced"] = {"variant": args.variant, "blocked": args.blocked, "panel": args.panel}
    # fingerprint of the whole y every rank holds after the last step: equal for every N and exchange
    import hashlib
    step()
    torch.cuda.synchronize()
    barrier()  # one-sided exchange: no rank may release its receive buffers while a peer is still storing
    yh = op.y_full.cpu().numpy()
    out["y_sha1"] = hashlib.sha1(yh.tobytes()).hexdigest()  # equal for every N under --order reference
    out["y_sum"], out["y_norm2"] = float("%.13g" % yh.sum()), float("%.13g" % np.sqrt((yh * yh).sum()))  # equal to ~1e-15 under either order
    if tuning:
        out["config"]["exchange_ms_per_step_by_chunks"] = tuning  # measured before the timed region
    if fell_back:
        out["config"]["exchange_fallback"] = "the one-sided exchange gave up during the timed region; the region was timed again over RCCL"
    if len(handles) > 1:
        roofline["launches_per_step"] = len(handles)

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
                               "sample": "whole matrix, median of %d serial CSC-scatter mulV runs of the oracle, %.2f s each" % (len(times), t_cpu)}
        out["parity"] = {"checked": int(n), "not_close_1e-10": int(bad),
                         "bit_identical": bool(np.array_equal(y_gpu, y_cpu))}
        # a fair CPU (NOT the reference): OpenMP CSR gather on all host cores, arrays placed by parallel first
        # touch (the exported arrays were written by one thread: all their pages would sit on one NUMA node)
        rp, ci, vv = H.export_csr()
        rp32 = rp.astype(np.int32)
        yo = np.zeros(n)
        t_omp = O.csr_spmv_omp_timed(rp32, ci, vv, xh, yo, reps=5)
        if t_omp > 0:
            out["cpu_fair_openmp"] = {"value": round(B_total / t_omp / 1e9, 3), "unit": "GB/s",
                                      "cores": O.omp_threads(), "bit_identical_to_reference_order": bool(np.array_equal(yo, y_cpu)),
                                      "note": "not the reference: OpenMP CSR gather"}

    if exchange_failed:
        out["invalid"] = "the one-sided exchange of y timed out on at least one rank: y is incomplete, the figures mean nothing"

    # ---- the other configurations of BASELINE.json (N = 1): C5 LU ladder, banded C2, C4 SpGEMM, SpMV on their matrices.
    # Each runs in a child process of its own (tools/bench_secondary.py --item ...) with a time limit: whatever
    # happens there — a device fault, an out-of-memory kill, a hang — costs that item, never the headline line
    # measured above; and a factorisation measured in a fresh process IS the one-shot case.  This parent
    # releases its device memory first and only waits; it is never replaced by another program.
    if N == 1 and rank == 0 and not args.no_secondary and args.secondary:
        import gc
        import subprocess
        for h in handles:
            h.free()
        del x, op, y_local, y_full, y_pieces
        gc.collect()
        torch.cuda.empty_cache()
        ffi.release_cached_memory()
        secondary = {}
        limit = float(os.environ.get("SPL_BENCH_ITEM_TIMEOUT", "420"))
        script = os.path.join(ROOT, "tools", "bench_secondary.py")
        for item in args.secondary.split(","):
            t_item = time.perf_counter()
            cmd = [sys.executable, script, "--item", item, "--n", str(n), "--draws", str(args.draws), "--steps", str(args.steps)]
            try:
                done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True)
                rec = None
                for line in reversed(done.stdout.splitlines()):
                    if line.startswith("{"):
                        rec = json.loads(line)
                        break
                if rec is None:
                    tail = (done.stderr or "").strip().splitlines()[-3:]
                    secondary[item] = {"error": "child exited with status %d and no result line" % done.returncode, "stderr_tail": tail}
                else:
                    secondary[rec["key"]] = rec["result"]
            except subprocess.TimeoutExpired:  # (subprocess.run has killed that child, and only it)
                secondary[item] = {"error": "no result within %.0f s: child process stopped" % limit}
            except Exception as e:
                secondary[item] = {"error": "%s: %s" % (type(e).__name__, e)}
            if item.startswith("c5:200") or item.startswith("zi:"):
                # what that child held goes back to the driver, which wipes it in the background (~40 GB/s); the next
                # child's clock should not start inside that
                time.sleep(min(8.0, 1.0 + (time.perf_counter() - t_item) * 0.2))
        out["secondary"] = secondary

    if rank == 0 and N > 1:
        # the last line on stderr, <= 500 bytes: what a failed or odd-looking scaling run is diagnosed from
        brief = {"n_gpus": N, "value": out["value"], "ms_per_step": out["ms_per_step"], "kernel_ms": roofline["kernel_ms"],
                 "exchange": exchange[:120], "tournament": tuning, "fell_back": fell_back}
        line = json.dumps(brief, separators=(",", ":"))
        sys.stderr.write("\n" + (line if len(line) <= 500 else json.dumps({k: brief[k] for k in ("n_gpus", "value", "ms_per_step", "kernel_ms", "fell_back")})) + "\n")
        sys.stderr.flush()
    if rank == 0:
        # The driver keeps the last 8 KB of stdout: the line stays under 7 500 bytes (tests/test_gpu_bench_rehearsal.py
        # asserts it on the full list of secondary items; round 4's 13.9 KB line lost three of them).  What every key
        # means is DESIGN.md §7's legend, not prose in the line.
        line = json.dumps(out, separators=(",", ":"))
        if len(line) > 7500:
            sys.stderr.write("bench.py: the result line has %d bytes (> 7500)\n" % len(line))
        print(line)
    if exchange_failed:
        sys.exit(3)
    if N > 1:
        if hasattr(op, "close"):
            barrier()
            op.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
