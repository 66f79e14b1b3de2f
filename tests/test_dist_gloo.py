"""N > 1 path on CPU ranks: world_size-2 (and 3) gloo process groups drive the product's
RowBlockSpMV (partition, ragged/equal all-gather of y).  The local product is supplied
by the oracle here (CPU ranks cannot launch HIP kernels); on the GPU the same class is
driven by the HIP kernel (tests/test_gpu_dist.py, bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, balanced, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg = g.load_package()
        from oracle import oracle as O
        n = 3001
        rp, ci, v = O.gen_random_csr(n, 20)
        if balanced:
            # skew the matrix: drop most entries of the first third of the rows
            keep = np.ones(len(ci), dtype=bool)
            for r in range(0, n // 3):
                keep[rp[r] + 2:rp[r + 1]] = False
            lens = np.array([keep[rp[r]:rp[r + 1]].sum() for r in range(n)])
            ci, v = ci[keep], v[keep]
            rp = np.concatenate([[0], np.cumsum(lens)])
            bounds = pkg.dist.nnz_balanced_bounds(rp, world)
        else:
            bounds = pkg.dist.equal_row_bounds(n, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        lrp = (rp[r0:r1 + 1] - rp[r0]).astype(np.int32)
        lci, lv = ci[rp[r0]:rp[r1]], v[rp[r0]:rp[r1]]

        def local_spmv(x, y_local):
            y = np.zeros(r1 - r0)
            O.csr_gaxpy32(lrp, lci, lv, x.numpy(), y)
            y_local.copy_(torch.from_numpy(y))

        op = pkg.dist.RowBlockSpMV(n, bounds, rank, world, local_spmv, "cpu")
        x = torch.from_numpy(O.gen_vector(n))
        y = op.step(x)
        y = op.step(x)  # a second step reuses every buffer
        y_ref = np.zeros(n)
        O.csr_gaxpy32(rp.astype(np.int32), ci, v, x.numpy(), y_ref)
        ok = np.array_equal(y.numpy(), y_ref)
        # blocks are contiguous, disjoint and cover all rows; nnz-balanced ones are balanced
        cover = bounds[0] == 0 and bounds[-1] == n and all(b0 <= b1 for b0, b1 in zip(bounds, bounds[1:]))
        bal = True
        if balanced:
            per = [int(rp[bounds[p + 1]] - rp[bounds[p]]) for p in range(world)]
            bal = max(per) - min(per) <= 2 * 20 + 2
        with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
            f.write("%d %d %d %d" % (ok, cover, bal, int(op.equal)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (2, True), (3, False)])
def test_rowblock_spmv_gloo(tmp_path, world, balanced):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, balanced, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, cover, bal, equal = open(tmp_path / ("rank%d" % r)).read().split()
        assert ok == "1" and cover == "1" and bal == "1"
        if balanced or world == 3:
            assert equal == "0"  # the ragged (padded) gather path was exercised


def test_bounds_helpers():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    assert pkg.dist.equal_row_bounds(10_000_000, 8) == [1_250_000 * p for p in range(9)]
    assert pkg.dist.equal_row_bounds(7, 3) == [0, 2, 4, 7]
    rp = np.array([0, 5, 5, 5, 10, 20, 40])
    b = pkg.dist.nnz_balanced_bounds(rp, 2)
    assert b[0] == 0 and b[-1] == 6 and b[1] == int(np.searchsorted(rp, 20, side="left"))
    assert pkg.dist.nnz_balanced_bounds(np.zeros(5, dtype=np.int64), 4) == [0, 0, 0, 0, 4]
