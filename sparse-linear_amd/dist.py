"""1-D row-block SpMV across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm).  Rank p owns the contiguous row block [r_p, r_{p+1}) of the CSR image —
the zero-copy sub-ranges of pointers/indices/values that the reference's
``basicUnsafeSlice`` would take (Sparse.hs:132-142) — and the full ``x``.  One
step is:  y_p = A_p x  (no communication: row sums never cross ranks, so the
result is bit-identical to the 1-GPU result)  followed by ONE collective, the
all-gather of y, so that y can be the next x.  There is no other exchange step
on this path; nothing else is communicated.

The local product is supplied as a callable so that the partition / gather
logic can be exercised on CPU ranks (gloo) in tests; the product path always
passes the HIP kernel launch (``DeviceMatrix.spmv_dev``).
"""
import numpy as np


def equal_row_bounds(n, nparts):
    """row-count-balanced contiguous blocks; == nnz-balanced for uniform rows"""
    return [n * p // nparts for p in range(nparts + 1)]


def nnz_balanced_bounds(rowptr, nparts):
    """Block p starts at the first row whose pointer is >= nnz*p/nparts — the same
    rule as spl_matrix_create_rowblock (csrc/abi.hip), so host and device agree."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = len(rowptr) - 1
    nnz = int(rowptr[-1])
    bounds = [0]
    for p in range(1, nparts):
        target = (nnz * p) // nparts
        bounds.append(int(np.searchsorted(rowptr, target, side="left")))
    bounds.append(n)
    return bounds


class RowBlockSpMV(object):
    """y = A x with A split by rows over the ranks of a process group.

    local_spmv(x, y_local) must fill y_local (a tensor view of this rank's rows)
    with A_p x on the current stream."""

    def __init__(self, n, bounds, rank, world, local_spmv, device, dtype=None, group=None):
        import torch
        self.torch = torch
        self.n, self.bounds, self.rank, self.world = int(n), list(bounds), int(rank), int(world)
        assert len(self.bounds) == world + 1 and self.bounds[0] == 0 and self.bounds[-1] == n
        self.local_spmv = local_spmv
        self.group = group
        dtype = dtype or torch.float64
        sizes = [self.bounds[p + 1] - self.bounds[p] for p in range(world)]
        self.sizes = sizes
        self.equal = all(s == sizes[0] for s in sizes)
        self.y_full = torch.zeros(n, dtype=dtype, device=device)
        r0, r1 = self.bounds[rank], self.bounds[rank + 1]
        if world == 1:
            self.y_local = self.y_full[r0:r1]
            self.pad = None
        elif self.equal:
            self.y_local = torch.zeros(r1 - r0, dtype=dtype, device=device)
            self.pad = None
        else:  # ragged blocks: gather fixed-size padded slices, then compact
            self.maxlen = max(sizes)
            self.send = torch.zeros(self.maxlen, dtype=dtype, device=device)
            self.y_local = self.send[: r1 - r0]
            self.pad = torch.zeros(world * self.maxlen, dtype=dtype, device=device)

    def step(self, x):
        """one SpMV over the whole matrix; returns the full y on every rank"""
        self.local_spmv(x, self.y_local)
        if self.world == 1:
            return self.y_full
        import torch.distributed as dist
        if self.equal:
            dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group)
        else:
            dist.all_gather_into_tensor(self.pad, self.send, group=self.group)
            for p in range(self.world):
                self.y_full[self.bounds[p]:self.bounds[p + 1]] = \
                    self.pad[p * self.maxlen: p * self.maxlen + self.sizes[p]]
        return self.y_full


def hip_local_spmv(handle, stream_getter):
    """the product's local operator: the HIP CSR-stream kernel on the current stream"""
    def run(x, y_local):
        handle.spmv_dev(x.data_ptr(), y_local.data_ptr(), accumulate=False, stream=stream_getter())
    return run
