#!/usr/bin/env python3
"""A/B of the tree solves under environment switches, one child process per setting: 3-D Poisson m^3 (`z`: the complex
shifted matrix), seconds of factorisation and of a solve in its steady state (b and x in HBM; eight columns; the transposed
system), errors against the manufactured solution, the solve report.
usage: python3 tools/solve_ab.py m [z] SETTING [SETTING ...]   with SETTING = NAME=VALUE[,NAME=VALUE...] or `default`"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(m, cplx):
    import numpy as np, torch, scipy.sparse as sp
    from __graft_entry__ import load_package
    pkg = load_package(); torch.cuda.set_device(0); U = pkg.umfpack
    H = pkg.DeviceMatrix.synthetic("poisson3d", m); rp, ci, v = H.export_csr(); H.free()
    n = m ** 3
    S = sp.csc_matrix((v, ci, rp), shape=(n, n))
    rng = np.random.default_rng(0xBEEF)
    if cplx:
        S = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n) - S); S.sort_indices()
        xs = rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(0.5, 1.5, n)
    else:
        xs = rng.uniform(0.5, 1.5, n)
    A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    b = np.asarray(S @ xs).ravel()
    an = U.analyze(A); fa = U.factor(A, an); torch.cuda.synchronize()
    del fa
    t = time.perf_counter(); fa = U.factor(A, an); torch.cuda.synchronize(); tf = time.perf_counter() - t
    x = U.linearSolve_(fa, U.UmfpackNormal, A, b)
    dev = torch.device("cuda", 0)
    Bd = torch.from_numpy(np.ascontiguousarray(b)).to(dev).reshape(1, -1)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        Xd = U.linearSolveManyDevice_(fa, U.UmfpackNormal, A, Bd)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    rep = dict(fa.solve_report)
    B8 = torch.from_numpy(np.ascontiguousarray(np.tile(b, (8, 1)))).to(dev)
    U.linearSolveManyDevice_(fa, U.UmfpackNormal, A, B8); torch.cuda.synchronize()
    t = time.perf_counter(); X8 = U.linearSolveManyDevice_(fa, U.UmfpackNormal, A, B8); torch.cuda.synchronize(); t8 = time.perf_counter() - t
    xt = U.linearSolve_(fa, U.UmfpackTrans, A, b)
    t = time.perf_counter(); xt = U.linearSolve_(fa, U.UmfpackTrans, A, b); tt = time.perf_counter() - t
    xd = Xd[0].cpu().numpy()
    x8 = X8[3].cpu().numpy()
    print("RESULT factor %.4f s | solve %.5f s (%s) | 8 columns %.5f s | transposed %.5f s | err %.2e err8 %.2e errT %.2e | walks %d bwd err %.2e | device GB %.2f | chain %.2f GB %.2f ms span %d | sha1 %s" % (
        tf, min(ts), " ".join("%.5f" % v for v in ts), t8, tt, float(np.max(np.abs(xd - xs) / np.abs(xs))), float(np.max(np.abs(x8 - xs) / np.abs(xs))),
        float(np.max(np.abs(xt - xs) / np.abs(xs))), rep["walks"], rep["backward_error"], fa.stats["device_bytes"] * 1e-9, rep["chain_bytes"] * 1e-9, rep["chain_build_ms"], rep["chain_span"],
        hashlib.sha1(xd.tobytes()).hexdigest()[:10]), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3] == "1")
        sys.exit(0)
    m = int(sys.argv[1]); args = sys.argv[2:]
    cplx = bool(args) and args[0] == "z"
    if cplx: args = args[1:]
    for setting in args or ["default"]:
        env = dict(os.environ)
        if setting != "default":
            for kv in setting.split(","):
                k, v = kv.split("=", 1); env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(m), "1" if cplx else "0"], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("m=%d %s [%s]: %s" % (m, "complex" if cplx else "real", setting, line[0][7:] if line else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])), flush=True)
