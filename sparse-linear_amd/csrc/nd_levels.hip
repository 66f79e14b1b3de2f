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
__global__ void nd_root_kernel(const int64_t *__restrict__ xadj, int root, int *__restrict__ mark, int stamp,
                               int *__restrict__ queue, int64_t *__restrict__ qbeg, int *__restrict__ qcnt,
                               int *__restrict__ levptr, int *__restrict__ tail, unsigned *__restrict__ blocks_done) {
  mark[root] = stamp;
  queue[0] = root;
  qbeg[0] = xadj[root];
  qcnt[0] = (int)(xadj[root + 1] - xadj[root]);
  levptr[0] = 0;
  levptr[1] = 1;
  *tail = 1;
  *blocks_done = 0;
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
    for (Slot &sl : slots)
      if (sl.s) {
        (void)hipStreamSynchronize(sl.s);
        pooled_stream_give(device, sl.s);
      }
  }

  GpuLevels(int n_, const int64_t *h_xadj, const int *h_adj) : n(n_) {
    SPL_HIP(hipGetDevice(&device));
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[nd_levels] service: %-20s %6.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    const size_t nnz = (size_t)h_xadj[n], N = (size_t)n;
    sort_temp_bytes = radix_sort_u64_temp_bytes(n);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t per_slot = up(N * 4) * 3 + up((N + kBatch + 4) * 4) + up(16) + up(N * 8) * 3 + up(sort_temp_bytes ? sort_temp_bytes : 1);
    const size_t total = up((N + 1) * 8) + up((nnz ? nnz : 1) * 4) + up(N * 4) + kSlots * per_slot;
    slab.alloc(std::max<size_t>(total, (size_t)1 << 30));
    lap("slab");
    char *at = slab.get();
    auto take = [&](size_t b) { char *p = at; at += up(b); return p; };
    xadj = reinterpret_cast<int64_t *>(take((N + 1) * 8));
    adj = reinterpret_cast<int *>(take((nnz ? nnz : 1) * 4));
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
      if (!sl.s) {
        for (Slot &made : slots)
          if (made.s) pooled_stream_give(device, made.s);
        throw DeviceError{SPL_ERROR_internal};
      }
    }
    lap("streams");
    try {
      hipStream_t s = slots[0].s;
      SPL_HIP(hipMemcpyAsync(xadj, h_xadj, (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
      if (nnz) SPL_HIP(hipMemcpyAsync(adj, h_adj, nnz * sizeof(int), hipMemcpyHostToDevice, s));
      SPL_HIP(hipMemsetAsync(mark, 0, N * sizeof(int), s));
      SPL_HIP(hipStreamSynchronize(s));
      lap("copies");
    } catch (...) {  // (the destructor does not run for an object whose constructor throws)
      for (Slot &sl : slots)
        if (sl.s) pooled_stream_give(device, sl.s);
      throw;
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
    hipLaunchKernelGGL(nd_root_kernel, dim3(1), dim3(1), 0, s, xadj, root, mark, st, sl.queue, sl.qbeg, sl.qcnt, sl.levptr,
                       sl.counters, blocks_done);
    h_levptr.assign(2, 0);
    h_levptr[1] = 1;
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
    static const bool timing = getenv("SPL_MF_TIMING") != nullptr;
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
std::unique_ptr<mf::LevelService> make_gpu_level_service(int n, const int64_t *xadj, const int *adj) {
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;
    return std::unique_ptr<mf::LevelService>(new GpuLevels(n, xadj, adj));
  } catch (...) {
    return nullptr;
  }
}

}  // namespace spl
