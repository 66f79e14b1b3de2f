"""1-D row-block SpMV (and column-block SpGEMM, at the end) across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm).  Rank p owns the contiguous row block [r_p, r_{p+1}) of the CSR image —
the zero-copy sub-ranges of pointers/indices/values that the reference's
``basicUnsafeSlice`` would take (Sparse.hs:132-142) — and the full ``x``.  One
step is:  y_p = A_p x  (no communication: row sums never cross ranks, so the
result is bit-identical to the 1-GPU result)  followed by ONE collective, the
all-gather of y, so that y can be the next x.  There is no other exchange step
on this path; nothing else is communicated.  PipelinedRowBlockSpMV cuts a rank's
rows into chunks and runs the gather of one chunk under the kernel of the next.

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
        """one SpMV over the whole matrix; returns the full y on every rank.  x must not alias the returned
        y (``x = op.step(x)`` does): the kernel / the gather would overwrite the x it is still reading —
        pass a copy (``op.step(y.clone())``) or keep two operators."""
        _refuse_alias(x, self.y_full)
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


def _refuse_alias(x, y_full):
    """x sharing storage with the operator's own y would be overwritten while it is being read"""
    if x is None:
        return
    try:
        same = x.untyped_storage().data_ptr() == y_full.untyped_storage().data_ptr()
    except AttributeError:  # not a torch tensor (CPU test doubles): compare the objects
        same = x is y_full
    if same:
        raise ValueError("step(x): x aliases the operator's result vector y_full; pass a copy "
                         "(x = op.step(x).clone()) — the product would overwrite the x it is reading")


def pipelined_piece_bounds(n, world, chunks):
    """Row ownership of PipelinedRowBlockSpMV: the rows are cut into chunks * world equal pieces in
    natural order; piece q = c * world + p belongs to rank p, chunk c.  Returns the chunks * world + 1
    piece boundaries."""
    pieces = chunks * world
    if n % pieces:
        raise ValueError("n = %d is not divisible by chunks * world = %d" % (n, pieces))
    return [q * (n // pieces) for q in range(pieces + 1)]


class PipelinedRowBlockSpMV(object):
    """The same step as RowBlockSpMV with the exchange hidden behind the kernels of the step itself.

    Rank p owns `chunks` row ranges instead of one (pieces c * world + p of pipelined_piece_bounds),
    so that the rows [c n/chunks, (c+1) n/chunks) of y are exactly what the ranks produce in their
    chunk c, in rank order: ONE all_gather_into_tensor per chunk fills them in place, in natural
    row order.  The gather of chunk c is started asynchronously as soon as the kernel of chunk c is
    queued and runs on the backend's own stream while the kernel of chunk c + 1 computes; the step
    ends when the last gather has landed.  Nothing crosses step boundaries: step k + 1 may use the
    y of step k as its x.  Row sums never cross ranks or chunks: y is bit-identical to RowBlockSpMV's.

    local_spmvs[c](x, y_piece) must fill y_piece with the product of this rank's rows of chunk c."""

    def __init__(self, n, rank, world, chunks, local_spmvs, device, dtype=None, group=None):
        import torch
        self.torch = torch
        self.n, self.rank, self.world, self.chunks = int(n), int(rank), int(world), int(chunks)
        assert len(local_spmvs) == chunks
        self.bounds = pipelined_piece_bounds(self.n, self.world, self.chunks)
        self.local_spmvs = list(local_spmvs)
        self.group = group
        dtype = dtype or torch.float64
        self.piece = self.n // (self.chunks * self.world)
        self.y_full = torch.zeros(self.n, dtype=dtype, device=device)
        span = self.piece * self.world  # rows of one chunk over all ranks
        self.out = [self.y_full[c * span:(c + 1) * span] for c in range(self.chunks)]
        if world == 1:
            self.y_local = list(self.out)  # a single rank writes y in place
        else:
            self.y_local = [torch.zeros(self.piece, dtype=dtype, device=device) for _ in range(self.chunks)]

    def rows_of(self, c):
        """[first, last) global rows of this rank's piece of chunk c"""
        q = c * self.world + self.rank
        return self.bounds[q], self.bounds[q + 1]

    def step(self, x):
        """as RowBlockSpMV.step; x must not alias the returned y either (the gather of chunk c would land in
        y_full while later chunks still read it as x)"""
        _refuse_alias(x, self.y_full)
        if self.world == 1:
            for c in range(self.chunks):
                self.local_spmvs[c](x, self.y_local[c])
            return self.y_full
        import torch.distributed as dist
        works = []
        for c in range(self.chunks):
            self.local_spmvs[c](x, self.y_local[c])
            works.append(dist.all_gather_into_tensor(self.out[c], self.y_local[c], group=self.group, async_op=True))
        for w in works:
            w.wait()
        return self.y_full


class _DevicePointerView(object):
    """zero-copy torch view of library-owned device memory (``__cuda_array_interface__``)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


class PeerStoreRowBlockSpMV(object):
    """The step of RowBlockSpMV / PipelinedRowBlockSpMV with a ONE-SIDED exchange instead of the collective
    (csrc/peer.hip): every rank pushes its piece of y into every peer's copy with device-to-device copies on
    one stream per peer (one per xGMI link, the copy engines move the data: no collective kernel takes CUs
    from the SpMV), then a step flag; a one-thread kernel on the compute stream waits for the peers' flags.
    Peers' buffers are mapped through IPC handles exchanged once over the process group.  With `chunks` > 1
    the rank owns pieces c * world + rank of pipelined_piece_bounds (one local product per chunk) and the
    copies of chunk c run under the kernel of chunk c + 1.  Two receive buffers alternate, so the tensor a
    step returns stays valid until the step after next.  Results are those of RowBlockSpMV (row sums never
    cross ranks or chunks)."""

    def __init__(self, n, rank, world, chunks, local_spmvs, device, stream_getter, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _ffi
        self.torch, self._ffi, self._C = torch, _ffi, C
        self.n, self.rank, self.world, self.chunks = int(n), int(rank), int(world), int(chunks)
        self.bounds = pipelined_piece_bounds(self.n, world, chunks) if chunks > 1 else equal_row_bounds(self.n, world)
        self.local_spmvs, self.stream_getter = list(local_spmvs), stream_getter
        assert len(self.local_spmvs) == self.chunks
        self.y_local = []
        for c in range(self.chunks):
            a, b = self.rows_of(c)
            self.y_local.append(torch.zeros(b - a, dtype=torch.float64, device=device))
        L = _ffi.lib()
        self._h = C.c_void_p()
        mine = C.create_string_buffer(192)
        barr = (C.c_int64 * len(self.bounds))(*self.bounds)
        # Both stages can fail on ONE rank (no memory, IPC mapping refused): the outcome is shared through the
        # process group before anyone raises, so that no rank is left waiting in a collective the failing rank
        # never enters — every rank raises, or none.
        st = L.spl_peer_exchange_create(rank, world, self.chunks, self.n, barr, mine, C.byref(self._h))
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, mine.raw if st == 0 else None, group=group)
        else:
            handles[0] = mine.raw if st == 0 else None
        if any(h is None for h in handles):
            self.close()
            raise RuntimeError("spl_peer_exchange_create failed on rank(s) %s (status %d here)"
                               % ([r for r, h in enumerate(handles) if h is None], st))
        st = L.spl_peer_exchange_connect(self._h, b"".join(handles))
        oks = [st == 0] * world
        if world > 1:
            dist.all_gather_object(oks, st == 0, group=group)  # also: every rank has mapped every peer before anyone stores
        if not all(oks):
            self.close()
            raise RuntimeError("spl_peer_exchange_connect failed on rank(s) %s (status %d here)"
                               % ([r for r, o in enumerate(oks) if not o], st))
        self._views = {}
        self.y_full = None

    def rows_of(self, c):
        q = c * self.world + self.rank
        return self.bounds[q], self.bounds[q + 1]

    def step(self, x):
        C, L = self._C, self._ffi.lib()
        s = C.c_void_p(self.stream_getter())
        for c in range(self.chunks):
            self.local_spmvs[c](x, self.y_local[c])
            self._ffi.check("spl_peer_exchange_push", L.spl_peer_exchange_push(self._h, c, C.c_void_p(self.y_local[c].data_ptr()), s))
        out = C.c_void_p()
        self._ffi.check("spl_peer_exchange_finish", L.spl_peer_exchange_finish(self._h, s, C.byref(out)))
        v = self._views.get(out.value)
        if v is None:
            v = self.torch.as_tensor(_DevicePointerView(out.value, self.n), device=self.y_local[0].device)
            self._views[out.value] = v
        self.y_full = v
        return v

    def failed(self):
        return bool(self._ffi.lib().spl_peer_exchange_failed(self._h))

    def flags_finegrained(self):
        """True when the step flags live in fine-grained device memory (csrc/peer.hip, 'Visibility')"""
        return self._ffi.lib().spl_peer_exchange_flags_finegrained(self._h) == 1

    def close(self):
        if self._h.value:
            getattr(self, "_views", {}).clear()
            self.y_full = None
            self._ffi.lib().spl_peer_exchange_free(self._C.byref(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hip_local_spmv(handle, stream_getter):
    """the product's local operator: the HIP CSR-stream kernel on the current stream"""
    def run(x, y_local):
        handle.spmv_dev(x.data_ptr(), y_local.data_ptr(), accumulate=False, stream=stream_getter())
    return run


# ---- SpGEMM: column blocks of the right operand ----------------------------------
def product_balanced_col_bounds(a_pointers, b_pointers, b_indices, nparts):
    """Contiguous column blocks of B with about equal numbers of scalar products a_ik * b_kj
    (the work of Gustavson's column j is sum_{k in B(:,j)} nnz(A(:,k)), Sparse.hs:691-702)."""
    a_len = np.diff(np.asarray(a_pointers, dtype=np.int64))
    b_pointers = np.asarray(b_pointers, dtype=np.int64)
    per_entry = a_len[np.asarray(b_indices, dtype=np.int64)]
    cum = np.concatenate([[0], np.cumsum(per_entry)])[b_pointers]  # products before column j
    total = int(cum[-1])
    ncols = len(b_pointers) - 1
    bounds = [0]
    for p in range(1, nparts):
        bounds.append(int(np.searchsorted(cum, (total * p) // nparts, side="left")))
    bounds.append(ncols)
    return [min(max(b, 0), ncols) for b in bounds]


class ColBlockSpGEMM(object):
    """C = A B with the columns of B (hence of C) split over the ranks of a process group.

    Column j of C depends on A and on column j of B only (the reference's `mm` is a map over
    the columns of B, Sparse.hs:691-702), so rank p multiplies the replicated A by the
    zero-copy column slice B(:, c_p:c_{p+1}) and owns C(:, c_p:c_{p+1}); there is no data-path
    collective.  `pointer_offsets()` is the one size exchange needed to address the blocks as
    one CSC matrix; `gather()` assembles the full C on every rank (tests / small results).

    local_mm(A, B_block) -> Matrix is the product's HIP SpGEMM (`sparse.mm`) in production."""

    def __init__(self, matA, matB, rank, world, local_mm, bounds=None, group=None):
        if matA.ncols != matB.nrows:
            raise ValueError("ColBlockSpGEMM: inner dimension mismatch")
        self.A, self.B, self.rank, self.world = matA, matB, int(rank), int(world)
        self.local_mm, self.group = local_mm, group
        self.bounds = list(bounds) if bounds is not None else \
            product_balanced_col_bounds(matA.pointers, matB.pointers, matB.indices, world)
        assert len(self.bounds) == world + 1 and self.bounds[0] == 0 and self.bounds[-1] == matB.ncols
        self.C_local = None

    def block_of_b(self, p):
        c0, c1 = self.bounds[p], self.bounds[p + 1]
        B = self.B
        s, e = int(B.pointers[c0]), int(B.pointers[c1])
        return type(B)(c1 - c0, B.nrows, B.pointers[c0:c1 + 1] - s, B.indices[s:e], B.values[s:e])

    def step(self):
        """this rank's column block of C"""
        self.C_local = self.local_mm(self.A, self.block_of_b(self.rank))
        return self.C_local

    def pointer_offsets(self):
        """nnz of the blocks before each rank's block (all-gather of one count per rank)"""
        import torch
        mine = torch.tensor([int(self.C_local.pointers[-1])], dtype=torch.int64)
        if self.world == 1:
            counts = mine
        else:
            import torch.distributed as dist
            counts = torch.zeros(self.world, dtype=torch.int64)
            dist.all_gather_into_tensor(counts, mine, group=self.group)
        return np.concatenate([[0], np.cumsum(counts.numpy())]).astype(np.int64)

    def gather(self):
        """the full C on every rank"""
        import torch
        off = self.pointer_offsets()
        Cl = self.C_local
        if self.world == 1:
            return Cl
        import torch.distributed as dist
        maxnz = int(np.max(np.diff(off)))
        maxc = max(self.bounds[p + 1] - self.bounds[p] for p in range(self.world))

        def padded(a, n, dtype):
            t = torch.zeros(n, dtype=dtype)
            t[: len(a)] = torch.from_numpy(np.ascontiguousarray(a))
            return t

        outs = []
        for arr, n, dt in ((Cl.pointers[1:], maxc, torch.int64), (Cl.indices, maxnz, torch.int64),
                           (Cl.values, maxnz, torch.float64)):
            buf = torch.zeros(self.world * n, dtype=dt)
            dist.all_gather_into_tensor(buf, padded(arr, n, dt), group=self.group)
            outs.append(buf.numpy().reshape(self.world, n))
        ptrs, idx, val = [np.zeros(1, dtype=np.int64)], [], []
        for p in range(self.world):
            nc, nz = self.bounds[p + 1] - self.bounds[p], int(off[p + 1] - off[p])
            ptrs.append(outs[0][p, :nc] + off[p])
            idx.append(outs[1][p, :nz])
            val.append(outs[2][p, :nz])
        return type(Cl)(self.B.ncols, self.A.nrows, np.concatenate(ptrs), np.concatenate(idx), np.concatenate(val))
