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


def _pipelined_worker(rank, world, port, chunks, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg = g.load_package()
        from oracle import oracle as O
        n = 4800  # divisible by chunks * world for every case below
        rp, ci, v = O.gen_random_csr(n, 20)
        bounds = pkg.dist.pipelined_piece_bounds(n, world, chunks)
        calls = []

        def piece_product(c):
            q = c * world + rank
            r0, r1 = bounds[q], bounds[q + 1]
            lrp = (rp[r0:r1 + 1] - rp[r0]).astype(np.int32)
            lci, lv = ci[rp[r0]:rp[r1]], v[rp[r0]:rp[r1]]

            def run(x, y_piece):
                y = np.zeros(r1 - r0)
                O.csr_gaxpy32(lrp, lci, lv, x.numpy(), y)
                y_piece.copy_(torch.from_numpy(y))
                calls.append(c)
            return run

        op = pkg.dist.PipelinedRowBlockSpMV(n, rank, world, chunks, [piece_product(c) for c in range(chunks)], "cpu")
        rows_ok = all(op.rows_of(c) == (bounds[c * world + rank], bounds[c * world + rank + 1]) for c in range(chunks))
        x = torch.from_numpy(O.gen_vector(n))
        y = op.step(x)
        y = op.step(y.clone() * 1e-3 + x)  # the y of one step feeds the next: nothing crosses step boundaries
        y1_ref = np.zeros(n)
        O.csr_gaxpy32(rp.astype(np.int32), ci, v, x.numpy(), y1_ref)
        x2 = y1_ref * 1e-3 + x.numpy()
        y2_ref = np.zeros(n)
        O.csr_gaxpy32(rp.astype(np.int32), ci, v, x2, y2_ref)
        ok = np.array_equal(y.numpy(), y2_ref)
        order_ok = calls == list(range(chunks)) * 2
        with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
            f.write("%d %d %d" % (ok, rows_ok, order_ok))
    finally:
        dist.destroy_process_group()


def _spgemm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        pkg = g.load_package()
        from oracle import oracle as O
        from tests.helpers import mat_to_tuple, tuple_to_mat, tuples_equal
        n = 700
        rp, ci, v = O.gen_random_csr(n, 6)       # CSR arrays of a matrix == CSC arrays of its transpose
        A = pkg.Matrix(n, n, rp, ci, v)
        rp2, ci2, v2 = O.gen_random_csr(n, 9, seed=77)
        # skew B: the last third of its columns is nearly empty, so balanced blocks are ragged
        keep = np.ones(len(ci2), dtype=bool)
        for c in range(2 * n // 3, n):
            keep[rp2[c] + 1:rp2[c + 1]] = False
        lens = np.array([keep[rp2[c]:rp2[c + 1]].sum() for c in range(n)])
        B = pkg.Matrix(n, n, np.concatenate([[0], np.cumsum(lens)]), ci2[keep], v2[keep])

        def local_mm(a, b):  # CPU ranks: the oracle stands in for the HIP SpGEMM
            return tuple_to_mat(pkg, O.mm(mat_to_tuple(a), mat_to_tuple(b)))

        op = pkg.dist.ColBlockSpGEMM(A, B, rank, world, local_mm)
        op.step()
        C = op.gather()
        ref = O.mm(mat_to_tuple(A), mat_to_tuple(B))
        ok = tuples_equal(mat_to_tuple(C), ref)
        off = op.pointer_offsets()
        ok_off = int(off[-1]) == int(ref[2][-1]) and len(off) == world + 1
        b = op.bounds
        a_len = np.diff(A.pointers)
        work = [int(a_len[B.indices[B.pointers[b[p]]:B.pointers[b[p + 1]]]].sum()) for p in range(world)]
        bal = max(work) - min(work) <= 2 * int(a_len.max()) * 9
        ragged = len({b[p + 1] - b[p] for p in range(world)}) > 1
        with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
            f.write("%d %d %d %d" % (ok, ok_off, bal, ragged))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_colblock_spgemm_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_spgemm_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, ok_off, bal, ragged = open(tmp_path / ("rank%d" % r)).read().split()
        assert ok == "1" and ok_off == "1" and bal == "1" and ragged == "1"


@pytest.mark.parametrize("world,balanced", [(2, False), (2, True), (3, False), (8, False), (8, True)])  # 8 = config C3's rank count
def test_rowblock_spmv_gloo(tmp_path, world, balanced):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, balanced, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, cover, bal, equal = open(tmp_path / ("rank%d" % r)).read().split()
        assert ok == "1" and cover == "1" and bal == "1"
        if balanced or world in (3, 8):
            assert equal == "0"  # the ragged (padded) gather path was exercised


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 4), (3, 2), (8, 2)])
def test_pipelined_rowblock_spmv_gloo(tmp_path, world, chunks):
    """chunked ownership + one asynchronous all-gather per chunk == the serial product, bit for bit"""
    port = _free_port()
    mp.spawn(_pipelined_worker, args=(world, port, chunks, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, rows_ok, order_ok = open(tmp_path / ("rank%d" % r)).read().split()
        assert ok == "1" and rows_ok == "1" and order_ok == "1"


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
    assert pkg.dist.pipelined_piece_bounds(24, 2, 3) == [0, 4, 8, 12, 16, 20, 24]
    with pytest.raises(ValueError):
        pkg.dist.pipelined_piece_bounds(25, 2, 3)
    one = pkg.dist.PipelinedRowBlockSpMV(12, 0, 1, 3, [lambda x, y, c=c: y.fill_(c) for c in range(3)], "cpu")
    assert one.step(None).tolist() == [0.0] * 4 + [1.0] * 4 + [2.0] * 4  # a single rank writes y in place
    # x = op.step(x) would hand the operator its own result vector as the next x: refused (the kernel, or
    # the gather of an earlier chunk, would overwrite the x that is still being read)
    with pytest.raises(ValueError):
        one.step(one.y_full)
    with pytest.raises(ValueError):
        one.step(one.y_full[4:])
    plain = pkg.dist.RowBlockSpMV(12, [0, 12], 0, 1, lambda x, y: y.fill_(1.0), "cpu")
    with pytest.raises(ValueError):
        plain.step(plain.step(None))
    assert plain.step(plain.step(None).clone()).tolist() == [1.0] * 12
