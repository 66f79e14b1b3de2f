// nd_levels.hip — level structures of the large regions of the nested dissection on the GPU (round 3).
//
// The ordering of mf_symbolic.hpp cuts a region at a level of a breadth-first level structure rooted at a pseudo-
// peripheral vertex: two traversals of the whole region.  On the host these traversals are bound by memory latency
// (a miss per vertex in xadj / adj and per edge in the marks) and they are the serial part of the analysis: the top
// regions come one after the other down the tree (config C5: 0.9 of the 1.4 s of umfpack_di_symbolic).  Here the graph
// lives in HBM for the duration of the analysis and a traversal is a chain of small launches, one per level: every
// frontier vertex claims its unreached neighbours with an atomic compare-and-swap on their mark and appends them to
// ONE queue (a wavefront reserves its slots with a single atomic), so the queue is grouped by level as the host
// algorithm expects; the last workgroup of a launch records where the next level ends.  The number of levels is not
// known in advance: the host enqueues the launches in batches and looks at the level pointers after each batch (a
// launch on an empty frontier does nothing).  The order INSIDE a level depends on the atomics; it is made
// deterministic afterwards by a radix sort of (level, vertex) keys (radix_sort.hip) — the tree must not depend on scheduling.
// What comes back: the queue (vertices by level, ascending ids inside a level) and the level pointers; the marks of
// this file are its own (the host's are not touched), so a region the GPU finds disconnected is simply traversed
// again by the host code.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>

#include "common.hpp"
#include "mf_symbolic.hpp"

namespace spl {
namespace {

constexpr int kBfsThreads = 256;
constexpr int kBfsBlocks = 128;
constexpr int kBfsShare = 4;  // lanes per frontier vertex
constexpr int kBatch = 64;  // levels enqueued between two looks at the level pointers

__global__ __launch_bounds__(256) void nd_stamp_kernel(const int *__restrict__ verts, int size, int *__restrict__ mark,
                                                       int stamp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < size) mark[verts[i]] = stamp;
}

// root of a traversal: level 0 = {root}
// (a root that does not carry `accept` — a caller's hint from outside the region — leaves an empty level 0: the
// traversal reaches nothing and the caller's host code takes the region)
__global__ void nd_root_kernel(const int64_t *__restrict__ xadj, int root, int *__restrict__ mark, int accept, int stamp,
                               int *__restrict__ queue, int64_t *__restrict__ qbeg, int *__restrict__ qcnt,
                               int *__restrict__ levptr, int *__restrict__ tail, unsigned *__restrict__ blocks_done) {
  *blocks_done = 0;
  if (mark[root] != accept) {
    levptr[0] = 0;
    levptr[1] = 0;
    *tail = 0;
    return;
  }
  mark[root] = stamp;
  queue[0] = root;
  qbeg[0] = xadj[root];
  qcnt[0] = (int)(xadj[root + 1] - xadj[root]);
  levptr[0] = 0;
  levptr[1] = 1;
  *tail = 1;
}

// level l: the frontier queue[levptr[l], levptr[l + 1]) claims its unreached neighbours (mark == accept -> stamp) and
// appends them from *tail on; the last workgroup to finish writes levptr[l + 2]
// (round 4: a level is a chain of dependent memory round trips — level pointers, frontier vertex, its adjacency range,
// a neighbour, its mark, the claim, the slot, the append, the last workgroup's bookkeeping: ~10 us whatever the size of
// the frontier.  Two of them are gone: the queue carries every vertex's adjacency range, fetched by whoever claimed it
// beside the slot reservation, and a neighbour is claimed by the compare-and-swap alone, without reading its mark first.)
__global__ __launch_bounds__(kBfsThreads) void nd_level_kernel(const int64_t *__restrict__ xadj,
                                                               const int *__restrict__ adj, int *__restrict__ mark,
                                                               int accept, int stamp, int *__restrict__ queue,
                                                               int64_t *__restrict__ qbeg, int *__restrict__ qcnt,
                                                               int *__restrict__ levptr, int l, int *__restrict__ tail,
                                                               unsigned *__restrict__ blocks_done) {
  const int beg = levptr[l], end = levptr[l + 1];
  const int lane = threadIdx.x & 63;
  // kBfsShare lanes share a vertex and take every kBfsShare-th neighbour: a level is a chain of dependent round trips
  // (neighbour, its mark, the claim, the append) per neighbour a lane walks, and the levels of a mesh are short — a
  // 2-D mesh of 2e6 vertices has 2 800 of at most 2 000 vertices (round 3: 4 lanes per vertex, 10.5 -> 7 us per level).
  // Whole wavefronts stay in the loop together (the slot reservation below is a wavefront operation)
  constexpr int per_wave = 64 / kBfsShare;
  for (int base = beg + (int)(blockIdx.x * (kBfsThreads / kBfsShare)) + (int)(threadIdx.x >> 6) * per_wave; base < end;
       base += (int)(gridDim.x * (kBfsThreads / kBfsShare))) {
    const int i = base + lane / kBfsShare;
    int64_t p = 0, pe = 0;
    if (i < end) {
      const int64_t b = qbeg[i];
      p = b + lane % kBfsShare;
      pe = b + qcnt[i];
    }
    while (__any(p < pe)) {
      int u = -1;
      if (p < pe) {
        const int cand = adj[p];
        p += kBfsShare;
        if (atomicCAS(&mark[cand], accept, stamp) == accept) u = cand;
      }
      const unsigned long long won = __ballot(u >= 0);
      if (won) {
        int64_t ub = 0, ue = 0;
        if (u >= 0) {  // (requested before the slot: the two round trips overlap)
          ub = xadj[u];
          ue = xadj[u + 1];
        }
        int slot = 0;
        if (lane == __ffsll((long long)won) - 1) slot = atomicAdd(tail, __popcll(won));
        slot = __shfl(slot, __ffsll((long long)won) - 1, 64);
        if (u >= 0) {
          const int at = slot + __popcll(won & ((1ull << lane) - 1ull));
          queue[at] = u;
          qbeg[at] = ub;
          qcnt[at] = (int)(ue - ub);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned done = atomicAdd(blocks_done, 1u);
    if (done == gridDim.x - 1) {
      levptr[l + 2] = atomicAdd(tail, 0);
      *blocks_done = 0;
    }
  }
}

// key of queue position i: (level << vbits) | vertex, vbits = the bits a vertex id takes (the sort's passes are per 8
// bits of key: 23 + 10 bits at config C5, five passes where (level << 32) | vertex would take six)
__global__ __launch_bounds__(256) void nd_keys_kernel(const int *__restrict__ queue, const int *__restrict__ levptr,
                                                      int nlev, int reached, int vbits,
                                                      unsigned long long *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= reached) return;
  int lo = 0, hi = nlev - 1;  // level of position i: levptr[lev] <= i < levptr[lev + 1]
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (levptr[mid] <= i) lo = mid; else hi = mid - 1;
  }
  keys[i] = ((unsigned long long)(unsigned)lo << vbits) | (unsigned)queue[i];
}

__global__ __launch_bounds__(256) void nd_unkey_kernel(const unsigned long long *__restrict__ keys, int reached, int vbits,
                                                       int *__restrict__ queue) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < reached) queue[i] = (int)(unsigned)(keys[i] & ((1ull << vbits) - 1ull));
}

// ---- boundary lists of the finished tree ---------------------------------------------------------------------------
// The boundary of front f lies in the separators above it.  Coordinates of f: its ancestors' pivots laid end to end,
// parent first — up(f) = the number of pivots above f of them; position h of ancestor a has coordinate
// up(f) - up(a) - np(a) + (h - p0(a)), and ascending coordinates are ascending positions (an ancestor further up comes
// later in the post-order).  One bitmap of up(f) bits per front:
//   * every edge (v, u) with u eliminated after v's front sets the bit of u in the bitmap of v's front (one pass over
//     the adjacency, atomic OR);
//   * a child's coordinates are its parent's shifted by np(parent): level by level from the bottom, a front ORs its
//     children's bitmaps, shifted right by its own pivot count, into its own;
//   * the set bits, counted and then written out in order, are the lists: sorted and free of duplicates by construction.
// The host does the same with sorted lists merged level by level (mf_symbolic.hpp), bound there by the six random
// reads of inv[] per vertex and the sorts; here it is a few hundred microseconds of kernels around the transfers.
constexpr int kHeavyDegree = 32;  // vertices with more neighbours are taken by a whole wavefront each

struct BndTree {  // per front, on the device
  const int *p0, *np, *parent, *up, *sub0, *c0, *c1;
  const int64_t *boff;  // first word of the front's bitmap
};

__device__ inline void bnd_set(const BndTree &t, int f, int lastf, int h, const int *__restrict__ front_of,
                               unsigned long long *__restrict__ bits, int *__restrict__ error) {
  const int a = front_of[h];
  // a must be an ancestor of f: f's pivots inside a's subtree, before a's own
  if (!(t.sub0[a] <= t.p0[f] && lastf <= t.p0[a])) { *error = 1; return; }
  const int c = t.up[f] - t.up[a] - t.np[a] + (h - t.p0[a]);
  if (c < 0 || c >= t.up[f]) { *error = 1; return; }
  atomicOr(&bits[t.boff[f] + (c >> 6)], 1ull << (c & 63));
}

__global__ __launch_bounds__(256) void bnd_own_kernel(int n, const int64_t *__restrict__ xadj, const int *__restrict__ adj,
                                                      const int *__restrict__ inv, const int *__restrict__ front_of,
                                                      BndTree t, unsigned long long *__restrict__ bits,
                                                      int *__restrict__ heavy, int *__restrict__ nheavy,
                                                      int *__restrict__ error) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  const int64_t b = xadj[v], e = xadj[v + 1];
  if (e - b > kHeavyDegree) {
    heavy[atomicAdd(nheavy, 1)] = v;
    return;
  }
  const int f = front_of[inv[v]], lastf = t.p0[f] + t.np[f];
  for (int64_t p = b; p < e; ++p) {
    const int h = inv[adj[p]];
    if (h >= lastf) bnd_set(t, f, lastf, h, front_of, bits, error);
  }
}

__global__ __launch_bounds__(256) void bnd_own_heavy_kernel(const int64_t *__restrict__ xadj, const int *__restrict__ adj,
                                                            const int *__restrict__ inv, const int *__restrict__ front_of,
                                                            BndTree t, unsigned long long *__restrict__ bits,
                                                            const int *__restrict__ heavy, const int *__restrict__ nheavy,
                                                            int *__restrict__ error) {
  const int lane = threadIdx.x & 63, nh = *nheavy;
  for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nh; k += gridDim.x * 4) {
    const int v = heavy[k], f = front_of[inv[v]], lastf = t.p0[f] + t.np[f];
    for (int64_t p = xadj[v] + lane; p < xadj[v + 1]; p += 64) {
      const int h = inv[adj[p]];
      if (h >= lastf) bnd_set(t, f, lastf, h, front_of, bits, error);
    }
  }
}

// fronts[0 .. count) of one tree level: bitmap |= each child's bitmap >> np(front)
__global__ __launch_bounds__(256) void bnd_merge_kernel(const int *__restrict__ fronts, BndTree t,
                                                        unsigned long long *__restrict__ bits) {
  const int f = fronts[blockIdx.x];
  const int nw = (t.up[f] + 63) >> 6;
  if (nw == 0) return;
  const int npf = t.np[f], s = npf >> 6, r = npf & 63;
  unsigned long long *dst = bits + t.boff[f];
  for (int which = 0; which < 2; ++which) {
    const int ch = which ? t.c1[f] : t.c0[f];
    if (ch < 0) continue;
    const int cw = (t.up[ch] + 63) >> 6;
    const unsigned long long *src = bits + t.boff[ch];
    for (int i = threadIdx.x; i < nw; i += 256) {
      const unsigned long long lo = i + s < cw ? src[i + s] : 0ull;
      unsigned long long val = lo;
      if (r) {
        const unsigned long long hi = i + s + 1 < cw ? src[i + s + 1] : 0ull;
        val = (lo >> r) | (hi << (64 - r));
      }
      if (val) dst[i] |= val;
    }
  }
}

__global__ __launch_bounds__(256) void bnd_count_kernel(int nf, BndTree t, const unsigned long long *__restrict__ bits,
                                                        int *__restrict__ nb) {
  const int lane = threadIdx.x & 63;
  for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < nf; f += gridDim.x * 4) {
    const int nw = (t.up[f] + 63) >> 6;
    const unsigned long long *w = bits + t.boff[f];
    int cnt = 0;
    for (int i = lane; i < nw; i += 64) cnt += __popcll(w[i]);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) nb[f] = cnt;
  }
}

__global__ __launch_bounds__(256) void bnd_emit_kernel(int nf, BndTree t, const unsigned long long *__restrict__ bits,
                                                       const int64_t *__restrict__ bptr, int *__restrict__ bidx) {
  const int lane = threadIdx.x & 63;
  for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < nf; f += gridDim.x * 4) {
    const int upf = t.up[f], nw = (upf + 63) >> 6;
    const unsigned long long *words = bits + t.boff[f];
    int64_t out0 = bptr[f];
    for (int i0 = 0; i0 < nw; i0 += 64) {  // (wave-uniform trip count)
      const int i = i0 + lane;
      unsigned long long w = i < nw ? words[i] : 0ull;
      const int cnt = __popcll(w);
      int pre = cnt;  // inclusive scan over the lanes
      for (int off = 1; off < 64; off <<= 1) {
        const int got = __shfl_up(pre, off, 64);
        if (lane >= off) pre += got;
      }
      const int total = __shfl(pre, 63, 64);
      int64_t out = out0 + (pre - cnt);
      int a = t.parent[f];
      while (w) {
        const int c = (i << 6) + (__ffsll((long long)w) - 1);
        w &= w - 1;
        while (a >= 0 && c >= upf - t.up[a]) a = t.parent[a];  // the ancestor whose pivots hold coordinate c
        if (a < 0) break;  // (cannot happen: c < up(f))
        bidx[out++] = c - (upf - t.up[a] - t.np[a]) + t.p0[a];
      }
      out0 += total;
    }
  }
}

// Buffers of one traversal in flight.  The regions of a depth are dissected side by side by host threads; each call
// takes a free slot (its own stream), so the level structures of siblings overlap on the device — they are chains of
// small launches and leave most of it idle.  The marks are shared: concurrent regions are disjoint vertex sets and
// every call uses stamps of its own.
struct Slot {
  int *queue = nullptr, *verts = nullptr, *levptr = nullptr, *counters = nullptr;  // counters: [0] tail, [1] blocks_done
  int64_t *qbeg = nullptr;  // adjacency range of queue[i]: adj[qbeg[i] .. qbeg[i] + qcnt[i])
  int *qcnt = nullptr;
  unsigned long long *keys = nullptr, *keys_alt = nullptr;
  char *sort_temp = nullptr;
  std::vector<int> h_levptr;
  hipStream_t s = nullptr;
  bool busy = false;
};

struct GpuLevels : mf::LevelService {
  static constexpr int kSlots = 4;
  int device = 0;
  int n = 0;
  // Everything lives in ONE block of at least 1 GiB: blocks of that size go back to the library's pool when the
  // analysis is over (device_pool.hip) instead of to the driver, whose background wipe of freshly released memory the
  // allocations of the numeric factorisation that follows would otherwise wait for (seconds at config C5).
  DBuf<char> slab;
  int64_t *xadj = nullptr;
  int *adj = nullptr, *mark = nullptr;
  size_t sort_temp_bytes = 0;
  Slot slots[kSlots];
  std::atomic<int> stamp{0};
  std::mutex mu;
  std::condition_variable freed;

  ~GpuLevels() override {
    if (ready.valid()) {
      try {
        ready.get();
      } catch (...) {
      }
    }
    for (Slot &sl : slots)
      if (sl.s) {
        (void)hipStreamSynchronize(sl.s);
        pooled_stream_give(device, sl.s);
      }
  }

  // Two steps (round 5): the slab and the streams are taken by a thread of their own from the moment the analysis
  // begins — 5 + 4 ms the first time a process gets here, beside the construction of the adjacency on the host — and
  // the graph follows when it exists (graph()).  max_adj: an upper bound on the adjacency's length.
  std::future<void> ready;
  size_t adj_capacity = 0;

  GpuLevels(int n_, int64_t max_adj) : n(n_) {
    SPL_HIP(hipGetDevice(&device));
    adj_capacity = (size_t)std::max<int64_t>(max_adj, 1);
    ready = std::async(std::launch::async, [this] { prepare(); });
  }

  void prepare() {
    DeviceGuard g(device);  // (HIP's current device belongs to the thread)
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[nd_levels] service: %-20s %6.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    const size_t N = (size_t)n;
    sort_temp_bytes = radix_sort_u64_temp_bytes(n);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t per_slot = up(N * 4) * 3 + up((N + kBatch + 4) * 4) + up(16) + up(N * 8) * 3 + up(sort_temp_bytes ? sort_temp_bytes : 1);
    const size_t total = up((N + 1) * 8) + up(adj_capacity * 4) + up(N * 4) + kSlots * per_slot;
    slab.alloc(std::max<size_t>(total, (size_t)1 << 30));
    lap("slab");
    char *at = slab.get();
    auto take = [&](size_t b) { char *p = at; at += up(b); return p; };
    xadj = reinterpret_cast<int64_t *>(take((N + 1) * 8));
    adj = reinterpret_cast<int *>(take(adj_capacity * 4));
    mark = reinterpret_cast<int *>(take(N * 4));
    for (Slot &sl : slots) {
      sl.queue = reinterpret_cast<int *>(take(N * 4));
      sl.verts = reinterpret_cast<int *>(take(N * 4));
      sl.qcnt = reinterpret_cast<int *>(take(N * 4));
      sl.qbeg = reinterpret_cast<int64_t *>(take(N * 8));
      sl.levptr = reinterpret_cast<int *>(take((N + kBatch + 4) * 4));
      sl.counters = reinterpret_cast<int *>(take(16));
      sl.keys = reinterpret_cast<unsigned long long *>(take(N * 8));
      sl.keys_alt = reinterpret_cast<unsigned long long *>(take(N * 8));
      sl.sort_temp = take(sort_temp_bytes ? sort_temp_bytes : 1);
      sl.s = pooled_stream_take(device);
      if (!sl.s) throw DeviceError{SPL_ERROR_internal};  // (the destructor gives back the streams taken so far)
    }
    SPL_HIP(hipMemsetAsync(mark, 0, N * sizeof(int), slots[0].s));
    lap("streams");
  }

  // the graph to the device; false: no service (the preparation failed, the graph is larger than announced, a copy
  // failed): the caller drops the object
  bool graph(const int64_t *h_xadj, const int *h_adj) override {
    try {
      if (ready.valid()) ready.get();
      DeviceGuard g(device);
      const size_t nnz = (size_t)h_xadj[n], N = (size_t)n;
      if (nnz > adj_capacity) return false;
      hipStream_t s = slots[0].s;
      SPL_HIP(hipMemcpyAsync(xadj, h_xadj, (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
      if (nnz) SPL_HIP(hipMemcpyAsync(adj, h_adj, nnz * sizeof(int), hipMemcpyHostToDevice, s));
      SPL_HIP(hipStreamSynchronize(s));
      return true;
    } catch (...) {
      return false;
    }
  }

  Slot &acquire() {
    std::unique_lock<std::mutex> lk(mu);
    Slot *got = nullptr;
    freed.wait(lk, [&] {
      for (Slot &sl : slots)
        if (!sl.busy) { got = &sl; return true; }
      return false;
    });
    got->busy = true;
    return *got;
  }
  void release(Slot &sl) {
    {
      std::lock_guard<std::mutex> lk(mu);
      sl.busy = false;
    }
    freed.notify_one();
  }

  // one traversal from `root` over the vertices whose mark is `accept`; they get `st`.  Returns the number of
  // levels; h_levptr[0 .. levels] are the level pointers.
  int traverse(Slot &sl, int root, int accept, int st) {
    hipStream_t s = sl.s;
    std::vector<int> &h_levptr = sl.h_levptr;
    unsigned *blocks_done = reinterpret_cast<unsigned *>(sl.counters + 1);
    hipLaunchKernelGGL(nd_root_kernel, dim3(1), dim3(1), 0, s, xadj, root, mark, accept, st, sl.queue, sl.qbeg, sl.qcnt,
                       sl.levptr, sl.counters, blocks_done);
    h_levptr.assign(2, 0);
    h_levptr[1] = 1;  // (0 on the device if the root was refused: every later pointer is then 0 and the count below too)
    int l = 0;  // next level to expand
    for (;;) {
      for (int k = 0; k < kBatch; ++k)
        hipLaunchKernelGGL(nd_level_kernel, dim3(kBfsBlocks), dim3(kBfsThreads), 0, s, xadj, adj, mark,
                           accept, st, sl.queue, sl.qbeg, sl.qcnt, sl.levptr, l + k, sl.counters, blocks_done);
      h_levptr.resize((size_t)l + kBatch + 2);
      SPL_HIP(hipMemcpyAsync(h_levptr.data() + l + 2, sl.levptr + l + 2, (size_t)kBatch * sizeof(int),
                             hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
      for (int k = 0; k < kBatch; ++k)
        if (h_levptr[(size_t)l + k + 2] == h_levptr[(size_t)l + k + 1]) {  // level l + k + 1 is empty: done
          h_levptr.resize((size_t)l + k + 2);
          return l + k + 1;
        }
      l += kBatch;
    }
  }

  // the boundary lists of the finished tree (kernels above); the traversals are over: slot 0's stream and buffers
  bool boundaries(mf::Tree &T) override {
    const int nf = T.nfronts;
    if (nf <= 0 || T.n != n || (int)T.inv.size() != n || (int)T.front_of.size() != n) return false;
    if (T.maxdepth > 2048) return false;  // (a launch per level: chains of hub peels stay on the host)
    DeviceGuard g(device);
    static const bool timing = getenv("SPL_MF_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    // per front: pivots above it, first position of its subtree, children, bitmap offsets
    std::vector<int> up((size_t)nf, 0), sub0(T.p0), c0((size_t)nf, -1), c1((size_t)nf, -1);
    std::vector<int64_t> boff((size_t)nf + 1, 0);
    for (int f = nf - 1; f >= 0; --f) {
      const int p = T.parent[(size_t)f];
      if (p >= 0 && p <= f) return false;  // not a post-order
      up[(size_t)f] = p < 0 ? 0 : up[(size_t)p] + T.np[(size_t)p];
    }
    for (int f = 0; f < nf; ++f) {
      const int p = T.parent[(size_t)f];
      if (p < 0) continue;
      sub0[(size_t)p] = std::min(sub0[(size_t)p], sub0[(size_t)f]);
      if (c0[(size_t)p] < 0) c0[(size_t)p] = f;
      else if (c1[(size_t)p] < 0) c1[(size_t)p] = f;
      else return false;  // more than two children
    }
    for (int f = 0; f < nf; ++f) boff[(size_t)f + 1] = boff[(size_t)f] + ((up[(size_t)f] + 63) >> 6);
    const int64_t words = boff[(size_t)nf];
    if (words > ((int64_t)1 << 29)) return false;  // 4 GiB of bitmaps: the host's lists then
    std::vector<int> order;  // fronts by depth
    std::vector<int> lev_ptr((size_t)T.maxdepth + 2, 0);
    order.reserve((size_t)nf);
    for (int d = 0; d <= T.maxdepth; ++d) {
      lev_ptr[(size_t)d] = (int)order.size();
      order.insert(order.end(), T.by_depth[(size_t)d].begin(), T.by_depth[(size_t)d].end());
    }
    lev_ptr[(size_t)T.maxdepth + 1] = (int)order.size();
    if ((int)order.size() != nf) return false;
    // one upload of everything per front: seven int arrays, the order, then the offsets
    const size_t F = (size_t)nf;
    std::vector<int> packed(8 * F);
    std::copy(T.p0.begin(), T.p0.end(), packed.begin());
    std::copy(T.np.begin(), T.np.end(), packed.begin() + F);
    std::copy(T.parent.begin(), T.parent.end(), packed.begin() + 2 * F);
    std::copy(up.begin(), up.end(), packed.begin() + 3 * F);
    std::copy(sub0.begin(), sub0.end(), packed.begin() + 4 * F);
    std::copy(c0.begin(), c0.end(), packed.begin() + 5 * F);
    std::copy(c1.begin(), c1.end(), packed.begin() + 6 * F);
    std::copy(order.begin(), order.end(), packed.begin() + 7 * F);
    Slot &sl = slots[0];
    hipStream_t s = sl.s;
    DBuf<int> d_packed(8 * F), d_nb(F);
    DBuf<int64_t> d_boff(F + 1), d_bptr(F + 1);
    DBuf<unsigned long long> d_bits((size_t)std::max<int64_t>(words, 1));
    int *d_inv = sl.queue, *d_front_of = sl.verts, *d_heavy = sl.qcnt;  // n ints each: idle now
    int *d_flags = sl.counters;                                          // [0] heavy count, [1] error
    SPL_HIP(hipMemcpyAsync(d_inv, T.inv.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    SPL_HIP(hipMemcpyAsync(d_front_of, T.front_of.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    SPL_HIP(hipMemcpyAsync(d_packed.get(), packed.data(), packed.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SPL_HIP(hipMemcpyAsync(d_boff.get(), boff.data(), (F + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    SPL_HIP(hipMemsetAsync(d_bits.get(), 0, (size_t)std::max<int64_t>(words, 1) * sizeof(unsigned long long), s));
    SPL_HIP(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), s));
    BndTree t;
    t.p0 = d_packed.get();
    t.np = d_packed.get() + F;
    t.parent = d_packed.get() + 2 * F;
    t.up = d_packed.get() + 3 * F;
    t.sub0 = d_packed.get() + 4 * F;
    t.c0 = d_packed.get() + 5 * F;
    t.c1 = d_packed.get() + 6 * F;
    t.boff = d_boff.get();
    const int *d_order = d_packed.get() + 7 * F;
    hipLaunchKernelGGL(bnd_own_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, xadj, adj, d_inv, d_front_of, t,
                       d_bits.get(), d_heavy, d_flags, d_flags + 1);
    hipLaunchKernelGGL(bnd_own_heavy_kernel, dim3(256), dim3(256), 0, s, xadj, adj, d_inv, d_front_of, t, d_bits.get(),
                       d_heavy, d_flags, d_flags + 1);
    for (int d = T.maxdepth - 1; d >= 0; --d) {
      const int count = lev_ptr[(size_t)d + 1] - lev_ptr[(size_t)d];
      if (count > 0)
        hipLaunchKernelGGL(bnd_merge_kernel, dim3((unsigned)count), dim3(256), 0, s, d_order + lev_ptr[(size_t)d], t, d_bits.get());
    }
    const unsigned per_front_grid = (unsigned)std::min<int64_t>(((int64_t)nf + 3) / 4, 4096);
    hipLaunchKernelGGL(bnd_count_kernel, dim3(per_front_grid), dim3(256), 0, s, nf, t, d_bits.get(), d_nb.get());
    std::vector<int> nb(F);
    int flags[2] = {0, 0};
    SPL_HIP(hipMemcpyAsync(nb.data(), d_nb.get(), F * sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipMemcpyAsync(flags, d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    if (flags[1]) return false;  // an edge to a front that is no ancestor: not a tree this scheme understands
    std::vector<int64_t> bptr(F + 1, 0);
    for (int f = 0; f < nf; ++f) bptr[(size_t)f + 1] = bptr[(size_t)f] + nb[(size_t)f];
    const int64_t total = bptr[F];
    std::vector<int> bidx((size_t)total);
    if (total > 0) {
      DBuf<int> d_bidx((size_t)total);
      SPL_HIP(hipMemcpyAsync(d_bptr.get(), bptr.data(), (F + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(bnd_emit_kernel, dim3(per_front_grid), dim3(256), 0, s, nf, t, d_bits.get(), d_bptr.get(), d_bidx.get());
      SPL_HIP(hipMemcpyAsync(bidx.data(), d_bidx.get(), (size_t)total * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
      SPL_HIP(hipGetLastError());
    }
    T.nb.swap(nb);
    T.bptr.swap(bptr);
    T.bidx.swap(bidx);
    if (timing)
      fprintf(stderr, "[nd_levels] boundaries: %d fronts, %.1f MB of bitmaps, %lld indices, %.2f ms\n", nf, (double)words * 8 / 1e6,
              (long long)total, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return true;
  }

  int levels(const int *region, int size, std::vector<int> &out_queue, std::vector<int64_t> &out_level_ptr,
             int root) override {
    DeviceGuard g(device);
    Slot &sl = acquire();
    struct Release {
      GpuLevels *self;
      Slot *sl;
      ~Release() { self->release(*sl); }
    } releaser{this, &sl};
    hipStream_t s = sl.s;
    std::vector<int> &h_levptr = sl.h_levptr;
    const int region_stamp = stamp.fetch_add(3) + 1, first = region_stamp + 1, second = region_stamp + 2;
    if (region_stamp > 0x7ffffff0) throw DeviceError{SPL_ERROR_internal};  // (never in practice: three stamps per call)
    static const bool timing = mf::detailed_timing();
    auto t_last = std::chrono::steady_clock::now();
    double laps[5] = {0, 0, 0, 0, 0};
    auto lap = [&](int k) {
      if (!timing) return;
      (void)hipStreamSynchronize(s);
      const auto now = std::chrono::steady_clock::now();
      laps[k] = std::chrono::duration<double, std::milli>(now - t_last).count();
      t_last = now;
    };
    SPL_HIP(hipMemcpyAsync(sl.verts, region, (size_t)size * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(nd_stamp_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, s, sl.verts, size,
                       mark, region_stamp);
    lap(0);
    int nlev = traverse(sl, root >= 0 ? root : region[0], region_stamp, first);
    int reached = h_levptr[(size_t)nlev];
    lap(1);
    if (reached == size && root < 0) {
      // once more from the far end: the smallest vertex id of the last level (a choice that does not depend on the
      // order the atomics produced)
      const int lb = h_levptr[(size_t)nlev - 1], le = h_levptr[(size_t)nlev];
      std::vector<int> last((size_t)(le - lb));
      SPL_HIP(hipMemcpyAsync(last.data(), sl.queue + lb, last.size() * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
      int root2 = last[0];
      for (int v : last) root2 = v < root2 ? v : root2;
      nlev = traverse(sl, root2, first, second);
      reached = h_levptr[(size_t)nlev];
    }
    lap(2);
    if (reached < size) return reached;  // disconnected: the caller's host code takes the region
    // deterministic order inside the levels
    SPL_HIP(hipMemcpyAsync(sl.levptr, h_levptr.data(), ((size_t)nlev + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    const unsigned gb = (unsigned)((reached + 255) / 256);
    int level_bits = 1, vbits = 1;
    while ((1ll << level_bits) < nlev) ++level_bits;
    while ((1ll << vbits) < n) ++vbits;
    hipLaunchKernelGGL(nd_keys_kernel, dim3(gb), dim3(256), 0, s, sl.queue, sl.levptr, nlev, reached, vbits,
                       sl.keys);
    const unsigned long long *sorted = radix_sort_u64(sl.keys, sl.keys_alt, reached, vbits + level_bits, sl.sort_temp, s);
    hipLaunchKernelGGL(nd_unkey_kernel, dim3(gb), dim3(256), 0, s, sorted, reached, vbits, sl.queue);
    lap(3);
    out_queue.resize((size_t)reached);
    SPL_HIP(hipMemcpyAsync(out_queue.data(), sl.queue, (size_t)reached * sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    out_level_ptr.assign(h_levptr.begin(), h_levptr.begin() + nlev + 1);
    lap(4);
    if (timing)
      fprintf(stderr, "[nd_levels] %9d vertices, %5d levels: upload + stamp %.1f, traversals %.1f + %.1f, order %.1f, download %.1f ms\n",
              size, nlev, laps[0], laps[1], laps[2], laps[3], laps[4]);
    return reached;
  }
};

}  // namespace

// nullptr when no device is usable (the analysis then runs on the host alone)
std::unique_ptr<mf::LevelService> make_gpu_level_service(int n, int64_t max_adj) {
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;
    return std::unique_ptr<mf::LevelService>(new GpuLevels(n, max_adj));
  } catch (...) {
    return nullptr;
  }
}

}  // namespace spl
