// mf_symbolic.hpp — host side of the multifrontal LU: nested-dissection ordering of the pattern of
// A + A^T and the frontal (assembly) tree built on it.  Plain C++17, no HIP: the same header is
// compiled into the library (umfpack_di_symbolic) and into the CPU self-check tests/mf_check.cpp.
//
// Why: the band profile of umfpack.hip stores n x (kl + ku + 1) entries — 160 GB at 100^3 unknowns of
// a 3-D grid, impossible at 200^3 (config C5).  Nested dissection cuts the fill to O(n^{4/3}) and the
// work to O(n^2): each tree node is a dense frontal matrix [pivots | boundary] that is partially
// factored by the same blocked fp64-MFMA kernels as the band path (a dense matrix is a band whose
// kl, ku exceed its size), and passes its Schur complement to its parent (multifrontal.hip).
//
// Ordering: "automatic nested dissection" (George & Liu): a breadth-first level structure from a
// pseudo-peripheral vertex of the current region, a middle level as vertex separator (smallest
// level among the balanced ones, thinned by one pass), recursion on the two sides; regions of at most
// `leaf` vertices become leaves.  Fronts are numbered in post-order, unknowns by front.  The two
// sides of a large region are dissected by separate threads (they share nothing but the graph).
#pragma once

#include <stdint.h>
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <exception>
#include <stdexcept>
#include <vector>

namespace spl {
namespace mf {

// The per-vertex arrays of the analysis (adjacency, marks, levels, the vertex array: tens of megabytes at 10^6 unknowns,
// read and written at random by the traversals) come from their own mappings with transparent huge pages asked for:
// on 4 KB pages every access to them is a TLB miss as well as a cache miss (the GPU box runs THP in `madvise` mode,
// which the C library's allocator never asks for), a fresh mapping costs a page fault per 4 KB the first time it is
// touched (half of what a one-shot analysis pays over a repeated one), and where the C library happens to put the
// arrays relative to each other moved the dissection by a quarter from one build to the next.
template <typename T>
struct HugeAlloc {
  using value_type = T;
  static constexpr size_t kHuge = (size_t)2 << 20, kFrom = (size_t)1 << 20;
  HugeAlloc() = default;
  template <typename U>
  HugeAlloc(const HugeAlloc<U> &) {}
  static size_t mapped_bytes(size_t bytes) { return (bytes + kHuge - 1) / kHuge * kHuge; }
  T *allocate(size_t count) {
    const size_t bytes = count * sizeof(T);
    if (bytes < kFrom) {
      void *p = malloc(bytes ? bytes : 1);
      if (!p) throw std::bad_alloc();
      return static_cast<T *>(p);
    }
    // (over-map by one huge page and give back the unaligned ends: the mapping then starts on a 2 MB boundary)
    const size_t len = mapped_bytes(bytes), span = len + kHuge;
    char *raw = static_cast<char *>(mmap(nullptr, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (raw == MAP_FAILED) throw std::bad_alloc();
    char *at = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) / kHuge * kHuge);
    if (at > raw) munmap(raw, (size_t)(at - raw));
    if (at + len < raw + span) munmap(at + len, (size_t)(raw + span - (at + len)));
    (void)madvise(at, len, MADV_HUGEPAGE);  // advisory: plain pages if the kernel declines
    return reinterpret_cast<T *>(at);
  }
  void deallocate(T *p, size_t count) noexcept {
    const size_t bytes = count * sizeof(T);
    if (bytes < kFrom) free(p);
    else munmap(p, mapped_bytes(bytes));
  }
  // resize(n) / vector(n) leave the elements uninitialised (they are plain integers, written before they are read):
  // no pass of zeros over tens of megabytes, and the first touch of a page is its user's, often one of several threads
  template <typename U>
  void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <typename U, typename... Args>
  void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
  template <typename U>
  bool operator==(const HugeAlloc<U> &) const { return true; }
  template <typename U>
  bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <typename T>
using BigVec = std::vector<T, HugeAlloc<T>>;

constexpr int kBlock = 64;  // pivot block of the dense kernels (NB of band_nopiv.hip)
// regions up to this size start their level structure from the end vertex their parent hands down
// (one BFS instead of two); larger regions, whose separators carry the flops, search properly
constexpr int kHintBelow = 40000;
// ... and regions of any size whose PARENT's level structure was that of a planar-like graph: levels^2 >= vertices / 2
// (a 2-D mesh of s x s vertices has 1.5 s .. 2 s levels; a 3-D one of s^3 has 3 s, which passes only below 18^3).
// There the end the parent hands down is as good a root as the search would find, and the traversals — thousands of
// narrow levels: sequential on the host, a launch each on the device — are the whole cost of the analysis.
// Level structures of regions of kTeamRegion vertices and more are built by a team of threads
// (team_bfs): these traversals are the serial part of the dissection — the top regions, one after
// the other down the tree.  A region's team is its share of the threads (the regions of a depth run
// side by side).
constexpr int kTeamRegion = 200000;
constexpr unsigned kMaxTeam = 16;
constexpr int kTeamFrontier = 4096;  // narrower levels are expanded by one thread

struct Tree {
  int n = 0, nfronts = 0, maxdepth = 0;
  std::vector<int> perm, inv;            // new -> old, old -> new
  std::vector<int> parent, slot, depth;  // per front; children have smaller ids than their parent;
                                         // slot = position among the parent's children (0 or 1)
  std::vector<int> p0, np, nb;           // pivots are the new indices [p0, p0 + np); nb boundary indices
  std::vector<int64_t> bptr;             // nfronts + 1
  std::vector<int> bidx;                 // boundary indices of every front (new numbering, ascending)
  std::vector<int> front_of;             // front that eliminates a new index
  std::vector<int> ld;                   // leading dimension of the dense front (>= np + nb)
  // Storage.  While a level is being factored its fronts are whole dense matrices in a transient
  // region (levels alternate between two regions: a level and its children are alive together);
  // afterwards only the factor panels are kept, compactly: P = columns [0, np) of the front
  // (fs x np, leading dimension ldp: L11\U11 and L21), U = rows [0, np) of the remaining columns
  // (np x nb, leading dimension ldu: U12).
  std::vector<int64_t> foff;             // offset of the whole front inside its level's region
  std::vector<int64_t> poff, uoff;       // offsets of the panels in the factor arena
  std::vector<int> ldp, ldu;
  std::vector<int64_t> ioff, woff, roff;  // offsets: inverse blocks, work vector, rel map
  std::vector<int64_t> level_elems;      // whole fronts of each tree level
  int64_t region_elems[2] = {0, 0};      // the two transient regions (even / odd levels)
  int64_t front_elems = 0;               // sum of all whole fronts (what resident fronts would take)
  int64_t panel_elems = 0, inv_elems = 0, work_elems = 0, rel_elems = 0;
  double flops = 0.0;
  std::vector<std::vector<int>> by_depth;  // fronts of each tree level
  // levels of the level structure the root region was cut with (0: the graph is disconnected).  A breadth-first
  // search has at most diameter + 1 levels and every ordering has bandwidth >= (n - 1) / diameter: a lower bound of
  // what a band factorisation of this pattern costs, without computing a band ordering (umfpack.hip)
  int root_levels = 0;
  // a region without separators was left as ONE leaf after kMaxPeels unbalanced cuts (dissect): its front is not meant
  // to be factored — a matrix that is not column dominant goes straight to static pivoting (umfpack.hip; ADVICE r4: this
  // used to be inferred from the memory plan's allocation failure, after a dense front of up to 1e5 rows had been tried)
  bool gave_up = false;
  int fs(int f) const { return np[(size_t)f] + nb[(size_t)f]; }
  // What a numeric factorisation derives from the tree alone — its arrays on the device, the positions of every
  // boundary index inside the parent's front — is built by the first factorisation and kept here for the later
  // ones with the same analysis (a FEAST-style caller refactors once per contour point): opaque to this header,
  // owned by multifrontal.hip.
  mutable std::shared_ptr<void> device_cache;
  mutable std::mutex device_cache_mu;
  Tree() = default;
  Tree(const Tree &) = delete;
  Tree &operator=(const Tree &o) {  // (build_tree starts from `T = Tree()`; the cache never travels)
    n = o.n; nfronts = o.nfronts; maxdepth = o.maxdepth;
    perm = o.perm; inv = o.inv; parent = o.parent; slot = o.slot; depth = o.depth; p0 = o.p0; np = o.np; nb = o.nb;
    bptr = o.bptr; bidx = o.bidx; front_of = o.front_of; ld = o.ld; foff = o.foff; poff = o.poff; uoff = o.uoff;
    ldp = o.ldp; ldu = o.ldu; ioff = o.ioff; woff = o.woff; roff = o.roff; level_elems = o.level_elems;
    region_elems[0] = o.region_elems[0]; region_elems[1] = o.region_elems[1];
    front_elems = o.front_elems; panel_elems = o.panel_elems; inv_elems = o.inv_elems; work_elems = o.work_elems;
    rel_elems = o.rel_elems; flops = o.flops; by_depth = o.by_depth; root_levels = o.root_levels; gave_up = o.gave_up;
    device_cache.reset();
    return *this;
  }
};

// Optional accelerator for the level structures of large regions (csrc/nd_levels.hip builds them on the GPU; this
// header stays free of HIP).  levels(): the pseudo-peripheral level structure of the region region[0 .. size) — a
// traversal from region[0], then, if that reached everything, another from the far end — as `queue` (the vertices
// grouped by level, in a deterministic order inside a level) and `level_ptr`; returns the number of vertices reached
// (< size: the region is disconnected and the outputs are not meaningful).  Thread-safe.
// root >= 0: ONE traversal, from that vertex (a hint from the parent region).
// boundaries(): the boundary lists of every front of the finished tree (T.nb, T.bptr, T.bidx from T.inv, T.front_of,
// T.p0, T.np, T.parent, T.depth), exactly what build_tree's host code makes level by level; false: not done (too large
// for the device's scheme, a tree it does not understand, any failure there) and the outputs are untouched.
// graph(): the adjacency the service was announced (ServiceFactory) — false: no service after all.
struct LevelService {
  virtual ~LevelService() {}
  virtual bool graph(const int64_t *xadj, const int *adj) = 0;
  virtual int levels(const int *region, int size, std::vector<int> &queue, std::vector<int64_t> &level_ptr, int root = -1) = 0;
  virtual bool boundaries(Tree &) { return false; }
};

// SPL_MF_TIMING set: phase times of the analysis on stderr; any value but "phases" also the laps of the top regions,
// which cost stream synchronisations inside the device's level structures (nd_levels.hip)
inline bool detailed_timing() {
  const char *e = getenv("SPL_MF_TIMING");
  return e && strcmp(e, "phases") != 0;
}

namespace detail {

// every off-diagonal entry (i, j) has its partner (j, i)?  Rows ascending inside a column (else: false at worst).
inline bool structurally_symmetric(int n, const int *Ap, const int *Ai) {
  std::atomic<bool> ok{true};
  auto check = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      if (!ok.load(std::memory_order_relaxed)) return;
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int r = Ai[p];
        if (r == j) continue;
        if (r < 0 || r >= n) { ok.store(false, std::memory_order_relaxed); return; }
        const int *first = Ai + Ap[r], *last = Ai + Ap[r + 1];
        const int *it = std::lower_bound(first, last, j);
        if (it == last || *it != j) { ok.store(false, std::memory_order_relaxed); return; }
      }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 8 ? 8 : nt) : 1;
  if (Ap[n] < 200000 || nt < 2) {
    check(0, n);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) {
      const int j0 = (int)((int64_t)n * t / nt), j1 = (int)((int64_t)n * (t + 1) / nt);
      try {
        pool.emplace_back(check, j0, j1);
      } catch (...) {
        check(j0, j1);
      }
    }
    check(0, (int)((int64_t)n / nt));
    for (std::thread &th : pool) th.join();
  }
  return ok.load();
}

// vertices from which a graph's analysis gets a level service at all: SPL_ND_GPU_MIN (0: never), 10^6 by default
inline int service_threshold() {
  const char *e = getenv("SPL_ND_GPU_MIN");
  return e ? atoi(e) : 1000000;
}
// ... from 200 000 if the graph is bulky (bulky_graph below): a 3-D mesh has a few hundred wide levels, which the device
// walks at ~10 us each while the host pays per vertex (80^3: 0.064 -> 0.043 s, a third of the CPU time); the thousands
// of narrow levels of a 2-D mesh of that size stay with the host's threads (800^2: 0.044 s there, 0.057 s here)
constexpr int kServiceFromBulky = 200000;

// Does the graph grow like a volume rather than like a surface?  Levels a breadth-first search from vertex 0 takes to
// reach 4096 vertices: a 2-D mesh needs 45 from an interior vertex (90 from a corner), a 3-D mesh 15 (29 from a corner).
template <typename XAdj, typename Adj>
inline bool bulky_graph(int n, const XAdj &xadj, const Adj &adj) {
  constexpr int kReach = 4096, kLevels = 36;
  if (n < kReach) return false;
  std::vector<char> seen((size_t)n, 0);
  std::vector<int> frontier(1, 0), next;
  seen[0] = 1;
  int reached = 1, levels = 0;
  while (!frontier.empty() && reached < kReach) {
    next.clear();
    for (int v : frontier)
      for (int64_t p = xadj[(size_t)v]; p < xadj[(size_t)v + 1]; ++p) {
        const int u = adj[(size_t)p];
        if (!seen[(size_t)u]) { seen[(size_t)u] = 1; next.push_back(u); }
      }
    reached += (int)next.size();
    frontier.swap(next);
    if (++levels > kLevels) return false;
  }
  return reached >= kReach;
}

struct Node {
  int left = -1, right = -1;  // indices in the same vector (post-order: children before the parent)
  std::vector<int> piv;       // pivots (separator, or all vertices of a leaf), original numbering
};

// State shared by all workers: the graph, the vertex array (workers own disjoint ranges of it) and
// two per-vertex arrays (a worker only touches the vertices of its own region).  Stamps come from
// one atomic counter so that they are unique across workers.
struct Shared {
  int n;
  const BigVec<int64_t> &xadj;
  const BigVec<int> &adj;
  int leaf;
  // mark[v] = stamp of the last thing that happened to v: put into a region (region stamp), or
  // reached by a BFS (that BFS's stamp).  One word per vertex answers "in my region and not yet
  // reached" during a traversal.
  BigVec<int> verts, mark, level;
  // claim[u] = (stamp of a team traversal << 32) | ~position of the frontier vertex that takes u
  // (team_bfs); allocated only when the graph is large enough for a team
  BigVec<uint64_t> claim;
  int max_team = 1;
  int team_region = kTeamRegion, team_frontier = kTeamFrontier;  // SPL_ND_TEAM_REGION / _FRONTIER (experiments)
  bool timing = false;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();  // (timing: when the dissection began)
  LevelService *service = nullptr;  // level structures of regions of service_min vertices and more (nullptr: none)
  // The device serves one region at a time (siblings wait for each other) while the host dissects the regions of a
  // depth side by side: the GPU pays where regions are few and large — the top of the tree, the serial part.
  int service_min = 1000000;        // SPL_ND_GPU_MIN (0: never)
  // ... 150 000 below a parent whose level structure is not planar-like (3-D meshes: a few hundred wide levels per
  // region — 100^3 analysis 0.133 -> 0.111 s, 160^3 0.485 -> 0.434 s; the thousands of narrow levels of a 2-D mesh
  // cost the device more than the host's threads below 10^6 vertices: 3000^2 0.65 -> 0.91 s with this limit for all)
  int service_min_bulky = 150000;
  bool root_bulky = false;  // the whole graph is bulky (bulky_graph): its root region too goes to the device from service_min_bulky on
  int root_levels = 0;  // levels of the root region's final level structure (written by the depth-0 call only)
  std::atomic<int> gave_up{0};  // a region was left as one leaf because it has no separators (Tree::gave_up)
  std::atomic<int> stamp{0};
  Shared(int n_, const BigVec<int64_t> &xa, const BigVec<int> &ad, int leaf_)
      : n(n_), xadj(xa), adj(ad), leaf(leaf_), verts((size_t)n_), mark((size_t)n_, 0), level((size_t)n_, 0) {
    for (int i = 0; i < n; ++i) verts[(size_t)i] = i;
    const unsigned hw = std::thread::hardware_concurrency();
    max_team = (int)std::min<unsigned>(hw ? hw : 1, kMaxTeam);
    timing = detailed_timing();
    if (const char *e = getenv("SPL_ND_TEAM")) max_team = std::max(1, std::min(atoi(e), 64));
    if (const char *e = getenv("SPL_ND_TEAM_REGION")) team_region = std::max(1024, atoi(e));
    if (const char *e = getenv("SPL_ND_TEAM_FRONTIER")) team_frontier = std::max(64, atoi(e));
    if (getenv("SPL_ND_GPU_MIN")) service_min = service_min_bulky = service_threshold();
    if (n >= team_region && max_team > 1) claim.assign((size_t)n, 0);
  }
};

// Barrier of a team of threads that exists for one traversal: spins briefly, then yields (the
// process may own fewer cores than it sees).
struct TeamBarrier {
  const int n;
  std::atomic<int> waiting{0};
  std::atomic<int> generation{0};
  explicit TeamBarrier(int n_) : n(n_) {}
  void wait() {
    const int gen = generation.load(std::memory_order_acquire);
    if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      waiting.store(0, std::memory_order_relaxed);
      generation.store(gen + 1, std::memory_order_release);
      return;
    }
    for (int spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins)
      if (spins > 2000) std::this_thread::yield();
  }
};

struct Worker {
  Shared &S;
  std::vector<int> queue;
  std::vector<int64_t> level_ptr;
  int region_stamp = 0, bfs_stamp = 0;
  explicit Worker(Shared &s) : S(s) {}

  // The traversals are bound by memory latency (on a mesh in its natural order the frontier strides through the
  // arrays: every vertex is a miss in xadj, adj, mark and level).  Three software-prefetch stages ahead of the
  // scan position i of a frontier q[0, end): the pointers of q[i + 24], the adjacency of q[i + 16], the marks
  // of the neighbours of q[i + 8].
  inline void prefetch_ahead(const int *q, size_t i, size_t end, bool claims = false) const {
    if (i + 24 < end) __builtin_prefetch(&S.xadj[(size_t)q[i + 24]]);
    if (i + 16 < end) __builtin_prefetch(&S.adj[(size_t)S.xadj[(size_t)q[i + 16]]]);
    if (i + 8 < end) {
      const int v = q[i + 8];
      const int64_t a = S.xadj[(size_t)v], b = std::min(S.xadj[(size_t)v + 1], a + 8);
      for (int64_t p = a; p < b; ++p) {
        __builtin_prefetch(&S.mark[(size_t)S.adj[(size_t)p]]);
        if (claims) __builtin_prefetch(&S.claim[(size_t)S.adj[(size_t)p]], 1);  // team traversal: the claim words
        else __builtin_prefetch(&S.level[(size_t)S.adj[(size_t)p]], 1);
      }
    }
  }

  // BFS over the vertices whose mark is `accept`; they get this BFS's stamp.  Fills queue (BFS
  // order), level[], level_ptr; returns #reached
  int bfs(int root, int accept) {
    bfs_stamp = ++S.stamp;
    queue.clear();
    level_ptr.clear();
    queue.push_back(root);
    S.mark[(size_t)root] = bfs_stamp;
    S.level[(size_t)root] = 0;
    level_ptr.push_back(0);
    size_t head = 0;
    int cur = 0;
    while (head < queue.size()) {
      prefetch_ahead(queue.data(), head, queue.size());
      const int v = queue[head];
      if (S.level[(size_t)v] != cur) {
        cur = S.level[(size_t)v];
        level_ptr.push_back((int64_t)head);
      }
      ++head;
      for (int64_t p = S.xadj[(size_t)v]; p < S.xadj[(size_t)v + 1]; ++p) {
        const int u = S.adj[(size_t)p];
        if (S.mark[(size_t)u] == accept) {
          S.mark[(size_t)u] = bfs_stamp;
          S.level[(size_t)u] = cur + 1;
          queue.push_back(u);
        }
      }
    }
    level_ptr.push_back((int64_t)queue.size());
    return (int)queue.size();
  }

  // The same traversal by a team of threads, with the same result: queue order, level[] and
  // level_ptr are exactly those of bfs().  A level is expanded in two sweeps over the frontier, each
  // thread taking a contiguous slice of it: first every unreached neighbour u is claimed for the
  // frontier position of the vertex that sees it (atomic max of stamp : ~position — the smallest
  // position wins, as it would in the sequential scan; claims of older traversals carry smaller
  // stamps); then every thread collects, in adjacency order, the neighbours its positions won, and
  // the slices are concatenated in order.
  int team_bfs(int root, int accept, int team, int cap) {
    bfs_stamp = ++S.stamp;
    const uint64_t tag = (uint64_t)(uint32_t)bfs_stamp << 32;
    queue.assign((size_t)cap, 0);  // sized once for the whole region; qsize entries are valid
    level_ptr.clear();
    queue[0] = root;
    S.mark[(size_t)root] = bfs_stamp;
    S.level[(size_t)root] = 0;
    level_ptr.push_back(0);
    std::vector<std::vector<int>> found((size_t)team);
    TeamBarrier barrier(team);
    size_t lo = 0, hi = 1;  // current frontier = queue[lo, hi)
    int cur = 0;
    bool more = true;
    std::atomic<uint64_t> *claim = reinterpret_cast<std::atomic<uint64_t> *>(S.claim.data());
    auto body = [&](int t) {
      for (;;) {
        // The shared state (more, lo, hi, cur) is only written by the first thread, in the two
        // places marked (W); a barrier separates each of them from the reads of every thread on
        // either side, so all threads take the same branch here.
        if (!more) return;
        if (hi - lo < (size_t)S.team_frontier) {
          // narrow levels are not worth a barrier each: the first thread expands them by itself,
          // as bfs() does, until the frontier is wide again or the traversal ends
          barrier.wait();  // everyone has read the state
          if (t == 0) {    // (W)
            while (more && hi - lo < (size_t)S.team_frontier) {
              size_t end = hi;
              for (size_t i = lo; i < hi; ++i) {
                prefetch_ahead(queue.data(), i, hi);
                const int v = queue[i];
                for (int64_t p = S.xadj[(size_t)v]; p < S.xadj[(size_t)v + 1]; ++p) {
                  const int u = S.adj[(size_t)p];
                  if (S.mark[(size_t)u] != accept) continue;
                  S.mark[(size_t)u] = bfs_stamp;
                  S.level[(size_t)u] = cur + 1;
                  queue[end++] = u;
                }
              }
              more = end > hi;
              if (more) {
                level_ptr.push_back((int64_t)hi);
                lo = hi;
                hi = end;
                ++cur;
              }
            }
          }
          barrier.wait();
          continue;
        }
        const size_t len = hi - lo, a = lo + len * (size_t)t / (size_t)team, b = lo + len * (size_t)(t + 1) / (size_t)team;
        for (size_t i = a; i < b; ++i) {
          prefetch_ahead(queue.data(), i, b, true);
          const int v = queue[i];
          const uint64_t mine = tag | (uint32_t)~(uint32_t)i;
          for (int64_t p = S.xadj[(size_t)v]; p < S.xadj[(size_t)v + 1]; ++p) {
            const int u = S.adj[(size_t)p];
            if (__atomic_load_n(&S.mark[(size_t)u], __ATOMIC_RELAXED) != accept) continue;
            uint64_t seen = claim[(size_t)u].load(std::memory_order_relaxed);
            while (seen < mine && !claim[(size_t)u].compare_exchange_weak(seen, mine, std::memory_order_relaxed)) {
            }
          }
        }
        barrier.wait();
        std::vector<int> &out = found[(size_t)t];
        out.clear();
        for (size_t i = a; i < b; ++i) {
          prefetch_ahead(queue.data(), i, b, true);
          const int v = queue[i];
          const uint64_t mine = tag | (uint32_t)~(uint32_t)i;
          for (int64_t p = S.xadj[(size_t)v]; p < S.xadj[(size_t)v + 1]; ++p) {
            const int u = S.adj[(size_t)p];
            if (__atomic_load_n(&S.mark[(size_t)u], __ATOMIC_RELAXED) != accept) continue;
            if (claim[(size_t)u].load(std::memory_order_relaxed) != mine) continue;
            __atomic_store_n(&S.mark[(size_t)u], bfs_stamp, __ATOMIC_RELAXED);  // also stops a repeated neighbour
            S.level[(size_t)u] = cur + 1;
            out.push_back(u);
          }
        }
        barrier.wait();
        size_t before = hi, total = 0;
        for (int k = 0; k < team; ++k) {
          if (k < t) before += found[(size_t)k].size();
          total += found[(size_t)k].size();
        }
        std::copy(out.begin(), out.end(), queue.begin() + (int64_t)before);
        barrier.wait();
        if (t == 0) {  // (W)
          more = total > 0;
          if (more) {
            level_ptr.push_back((int64_t)hi);
            lo = hi;
            hi += total;
            ++cur;
          }
        }
        barrier.wait();
      }
    };
    std::vector<std::thread> others;
    for (int t = 1; t < team; ++t) others.emplace_back(body, t);
    body(0);
    for (std::thread &th : others) th.join();
    queue.resize(hi);
    level_ptr.push_back((int64_t)queue.size());
    return (int)queue.size();
  }

  // bfs or team_bfs, by the size of the region
  int traverse(int root, int accept, int size) {
    // regions of one depth run side by side: each gets its share of the threads
    const int team = (int)(((int64_t)S.max_team * size + S.n / 2) / std::max(S.n, 1));
    if (team >= 2 && size >= S.team_region && !S.claim.empty()) return team_bfs(root, accept, team, size);
    return bfs(root, accept);
  }

  static void append(std::vector<Node> &dst, std::vector<Node> &&src) {
    const int off = (int)dst.size();
    for (Node &nd : src) {
      if (nd.left >= 0) nd.left += off;
      if (nd.right >= 0) nd.right += off;
      dst.push_back(std::move(nd));
    }
  }

  // subtrees of the two vertex ranges, then `top` as their parent; large ranges get their own thread
  // hint_l / hint_r: an extreme vertex of each side to start its level structure from (-1: none)
  // peels: consecutive cuts above these ranges that left more than nine tenths of their region on one side
  std::vector<Node> join(int lo, int mid, int hi, Node &&top, int depth, int hint_l = -1, int hint_r = -1,
                         bool thin = false, int peels = 0) {
    std::vector<Node> left, right;
    const bool fork = depth < 8 && mid - lo >= 20000 && hi - mid >= 20000;
    if (fork) {
      std::future<std::vector<Node>> other = std::async(std::launch::async, [this, lo, mid, depth, hint_l, thin, peels] {
        Worker w(S);
        return w.dissect(lo, mid, depth + 1, hint_l, thin, peels);
      });
      right = dissect(mid, hi, depth + 1, hint_r, thin, peels);
      left = other.get();
    } else {
      if (mid > lo) left = dissect(lo, mid, depth + 1, hint_l, thin, peels);
      if (hi > mid) right = dissect(mid, hi, depth + 1, hint_r, thin, peels);
    }
    if (S.timing && depth <= 5)  // when the subtrees of the chain of large regions were complete
      fprintf(stderr, "[dissect] depth %d, %9d vertices: closed at %8.1f ms\n", depth, hi - lo + (int)top.piv.size(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - S.t0).count());
    std::vector<Node> out;
    out.reserve(left.size() + right.size() + 1);
    const bool has_l = !left.empty(), has_r = !right.empty();
    append(out, std::move(left));
    const int l = has_l ? (int)out.size() - 1 : -1;
    append(out, std::move(right));
    const int r = has_r ? (int)out.size() - 1 : -1;
    top.left = l;
    top.right = r;
    out.push_back(std::move(top));
    return out;
  }

  std::vector<Node> make_leaf(int lo, int hi) {
    Node nd;
    nd.piv.assign(S.verts.begin() + lo, S.verts.begin() + hi);
    std::vector<Node> out;
    out.push_back(std::move(nd));
    return out;
  }

  // hint: a vertex of the region known to lie at one of its ends (the root or the deepest vertex of
  // the parent's level structure): one BFS from it replaces the two of the pseudo-peripheral search
  // A graph without separators — the pattern of A + A^T of a mesh matrix whose rows arrive in random order is an
  // expander: half a dozen levels, the smallest interior one next to the root — would be peeled ten vertices at a time,
  // every peel a traversal of the whole region (round 3: 18 s of analysis at 4e5 unknowns, tools/fuzz_lu_scale.py family
  // perm2d, for a tree nobody can factor).  After kMaxPeels such cuts in a row the region is left as it is: a leaf
  // whose front cannot fit, so the numeric phase goes straight to static pivoting, whose transversal gives the
  // matrix its mesh pattern back (umfpack.hip).  Hubs taken out one at a time (arrow matrices) get a longer rope.
  static constexpr int kMaxPeels = 6, kMaxHubPeels = 64;
  std::vector<Node> dissect(int lo, int hi, int depth, int hint = -1, bool thin_parent = false, int peels = 0) {
    const int size = hi - lo;
    if (size <= S.leaf) return make_leaf(lo, hi);
    if ((peels >= kMaxPeels && peels < 1000 && size > 16 * S.leaf) || (peels >= 1000 + kMaxHubPeels && size > 16 * S.leaf)) {
      S.gave_up.store(1, std::memory_order_relaxed);
      return make_leaf(lo, hi);
    }
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {  // SPL_MF_TIMING: the phases of the top regions
      if (!S.timing || depth > 2) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[dissect] depth %d, %9d vertices: %-22s %8.1f ms\n", depth, size, what,
              std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    region_stamp = ++S.stamp;
    auto stamp_region = [this, lo, hi] {
      for (int i = lo; i < hi; ++i) S.mark[(size_t)S.verts[(size_t)i]] = region_stamp;
    };
    int reached = -1;
    bool from_service = false;
    const bool to_service = S.service && S.service_min > 0 && size >= (thin_parent || (depth == 0 && !S.root_bulky) ? S.service_min : S.service_min_bulky);
    const bool hint_wanted = hint >= 0 && (size <= kHintBelow || thin_parent);
    bool use_hint = false;
    if (to_service) {
      // The host's marks of the region (what the host code needs if the device hands the region back, and what a
      // child's hint is checked against) are written beside the device's traversals, not before them: a pass of random
      // writes over the region, 0.5 - 1 ms per region on the chain of large regions down the tree at 10^6 unknowns.
      // The device checks the hint against its own marks (nd_levels.hip: a root outside the region reaches nothing).
      std::future<void> stamping;
      try {
        stamping = std::async(std::launch::async, stamp_region);
      } catch (...) {
        stamp_region();
      }
      try {
        reached = S.service->levels(S.verts.data() + lo, size, queue, level_ptr, hint_wanted ? hint : -1);
      } catch (...) {  // no memory on the device, or any other failure there: the host code below takes the region
        reached = -1;
      }
      if (stamping.valid()) stamping.get();
      use_hint = hint_wanted && S.mark[(size_t)hint] == region_stamp;
      from_service = reached == size;
      if (from_service) {
        lap("level structure (GPU)");
        // the separator is thinned below by looking at the marks and levels of its neighbours: give the level after
        // the cut what a host traversal would have left there (the cut level is chosen further down, so this is
        // done for the candidates' successors lazily: see `touches`)
        bfs_stamp = ++S.stamp;
      }
    } else {
      stamp_region();
      lap("region stamp");
      use_hint = hint_wanted && S.mark[(size_t)hint] == region_stamp;
    }
    if (from_service) {
      // nothing else to do here
    } else if (use_hint) {
      reached = traverse(hint, region_stamp, size);
    } else {
      reached = traverse(S.verts[(size_t)lo], region_stamp, size);
      lap("first level structure");
      // connected: once more from the far end (deeper, narrower levels); every vertex now carries
      // the first traversal's stamp
      if (reached == size) reached = traverse(queue.back(), bfs_stamp, size);
      lap("second level structure");
    }
    if (reached < size) {
      // disconnected region: no separator needed.  All components are found and dealt into two
      // groups of about equal size (largest first, each to the lighter group), so that a region
      // with many small fragments does not become a chain of single-fragment nodes.
      std::vector<int> comp_verts;
      std::vector<int64_t> comp_ptr(1, 0);
      comp_verts.reserve((size_t)size);
      comp_verts.insert(comp_verts.end(), queue.begin(), queue.end());
      comp_ptr.push_back((int64_t)comp_verts.size());
      for (int i = lo; i < hi; ++i) {  // vertices not reached yet still carry the region stamp
        const int v = S.verts[(size_t)i];
        if (S.mark[(size_t)v] != region_stamp) continue;
        bfs(v, region_stamp);
        comp_verts.insert(comp_verts.end(), queue.begin(), queue.end());
        comp_ptr.push_back((int64_t)comp_verts.size());
      }
      const int ncomp = (int)comp_ptr.size() - 1;
      std::vector<int> order((size_t)ncomp);
      for (int c = 0; c < ncomp; ++c) order[(size_t)c] = c;
      std::sort(order.begin(), order.end(), [&](int a, int b) {
        const int64_t sa = comp_ptr[(size_t)a + 1] - comp_ptr[(size_t)a], sb = comp_ptr[(size_t)b + 1] - comp_ptr[(size_t)b];
        return sa != sb ? sa > sb : a < b;
      });
      std::vector<int> ga, gb;
      int64_t wa = 0, wb = 0;
      for (int c : order) {
        const int64_t sz = comp_ptr[(size_t)c + 1] - comp_ptr[(size_t)c];
        std::vector<int> &g = wa <= wb ? ga : gb;
        (wa <= wb ? wa : wb) += sz;
        g.insert(g.end(), comp_verts.begin() + comp_ptr[(size_t)c], comp_verts.begin() + comp_ptr[(size_t)c + 1]);
      }
      std::copy(ga.begin(), ga.end(), S.verts.begin() + lo);
      std::copy(gb.begin(), gb.end(), S.verts.begin() + lo + (int)ga.size());
      const int na = (int)ga.size();
      std::vector<int>().swap(ga);
      std::vector<int>().swap(gb);
      std::vector<int>().swap(comp_verts);
      return join(lo, lo + na, hi, Node(), depth);
    }
    const int nlev = (int)level_ptr.size() - 1;
    if (depth == 0) S.root_levels = nlev;
    if (nlev < 3) {
      // No interior level: every vertex is within one step of both ends.  A small region, or a
      // genuinely dense one, becomes a leaf.  A large one held together by a few hubs (a dense row
      // and column: an arrow matrix) must not become one dense front: the vertex of largest degree
      // is taken out as a separator of its own and the rest is dissected again (it usually falls
      // apart into components).
      if (size <= 4 * S.leaf) return make_leaf(lo, hi);
      int hub = lo;
      int64_t best_deg = -1, edges = 0;
      for (int i = lo; i < hi; ++i) {
        const int v = S.verts[(size_t)i];
        const int64_t deg = S.xadj[(size_t)v + 1] - S.xadj[(size_t)v];
        edges += deg;
        if (deg > best_deg) { best_deg = deg; hub = i; }
      }
      // dense on the whole (more than a quarter of all pairs are edges): a leaf after all
      if ((double)edges > 0.25 * (double)size * (double)size) return make_leaf(lo, hi);
      Node top;
      top.piv.push_back(S.verts[(size_t)hub]);
      std::swap(S.verts[(size_t)hub], S.verts[(size_t)hi - 1]);
      // (hub peels count from 1000 on: a separate, longer budget)
      return join(lo, hi - 1, hi - 1, std::move(top), depth, -1, -1, false, peels >= 1000 ? peels + 1 : 1001);
    }
    // smallest level among the balanced ones; the balance requirement is relaxed until one exists
    int best = -1;
    for (double frac : {0.35, 0.20, 0.10, 0.0}) {
      int64_t best_size = -1;
      for (int t = 1; t + 1 < nlev; ++t) {
        const int64_t before = level_ptr[(size_t)t], after = (int64_t)size - level_ptr[(size_t)t + 1];
        if ((double)std::min(before, after) < frac * size) continue;
        const int64_t sz = level_ptr[(size_t)t + 1] - level_ptr[(size_t)t];
        if (best < 0 || sz < best_size) { best = t; best_size = sz; }
      }
      if (best >= 0) break;
    }
    const int t = best;
    if (from_service)  // the host's marks and levels of level t + 1, as a host traversal would have left them
      for (int64_t q = level_ptr[(size_t)t + 1]; q < level_ptr[(size_t)t + 2]; ++q) {
        S.mark[(size_t)queue[(size_t)q]] = bfs_stamp;
        S.level[(size_t)queue[(size_t)q]] = t + 1;
      }
    // queue = [levels < t | level t | levels > t]; thin the separator: a vertex of level t without a
    // neighbour in level t+1 can join the first side
    std::vector<int> side1(queue.begin(), queue.begin() + level_ptr[(size_t)t]);
    std::vector<int> side2(queue.begin() + level_ptr[(size_t)t + 1], queue.end());
    Node top;
    for (int64_t q = level_ptr[(size_t)t]; q < level_ptr[(size_t)t + 1]; ++q) {
      const int v = queue[(size_t)q];
      bool touches = false;
      for (int64_t p = S.xadj[(size_t)v]; p < S.xadj[(size_t)v + 1] && !touches; ++p) {
        const int u = S.adj[(size_t)p];
        touches = S.mark[(size_t)u] == bfs_stamp && S.level[(size_t)u] == t + 1;  // reached by this BFS
      }
      if (touches) top.piv.push_back(v); else side1.push_back(v);
    }
    std::copy(side1.begin(), side1.end(), S.verts.begin() + lo);
    std::copy(side2.begin(), side2.end(), S.verts.begin() + lo + (int)side1.size());
    const int n1 = (int)side1.size(), n2 = (int)side2.size();
    lap("separator and sides");
    const int end1 = queue.front(), end2 = queue.back();  // the two ends of this level structure
    std::vector<int>().swap(side1);
    std::vector<int>().swap(side2);
    const bool thin = (int64_t)nlev * nlev * 2 >= (int64_t)size;
    const bool peel = (double)std::max(n1, n2) > 0.9 * (double)size;
    return join(lo, lo + n1, lo + n1 + n2, std::move(top), depth, end1, end2, thin, peel ? (peels >= 1000 ? 1 : peels + 1) : 0);
  }
};

}  // namespace detail

// Pattern of A given as CSC arrays (n x n); leaf = largest region that is not dissected further.
// mult > 1: (Ap, Ai) is the pattern of an n x n matrix of dense mult x mult blocks — the complex matrix behind
// the real embedding of umfpack_zi.hip, mult = 2, unknowns interleaved — and the tree is the one of the expanded
// (n mult) x (n mult) matrix: ordered on the small graph (a half of the vertices, a quarter of the edges), every
// vertex then replaced by its mult unknowns, which stay together in their front.  `leaf` counts small vertices.
inline void expand_tree(Tree &T, int mult);
inline void layout_tree(Tree &T);
// unexpanded != nullptr (mult > 1): also receives the tree of the small graph itself, laid out for fronts of its own
// n unknowns (native complex fronts, multifrontal.hip: one dissection serves both)
// pattern_symmetric: 1 / 0 if the caller has already run detail::structurally_symmetric, -1: not known
// make_service: called when the analysis begins with n and an upper bound on the adjacency's length (the service gets
// its memory and streams ready beside the construction of the adjacency, which it is handed by graph()); may return nullptr
using ServiceFactory = std::unique_ptr<LevelService> (*)(int, int64_t);
inline void build_tree(int n, const int *Ap, const int *Ai, int leaf, Tree &T, int mult = 1, Tree *unexpanded = nullptr,
                       int pattern_symmetric_hint = -1, ServiceFactory make_service = nullptr) {
  T = Tree();
  T.n = n;
  const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // phase times on stderr (diagnostic)
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[build_tree] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // adjacency of A + A^T without the diagonal.  A structurally symmetric pattern (meshes, FEAST's z B - A; rows
  // ascending inside a column, as the entry points validate) IS its own adjacency: the column lists minus the diagonal,
  // half the edges the general construction below would make the traversals scan (there every entry contributes to
  // both of its ends, so a symmetric pair appears twice: harmless for BFS, removed from the boundary lists by sort +
  // unique, but twice the edge work).
  const bool pattern_symmetric = pattern_symmetric_hint >= 0 ? pattern_symmetric_hint != 0 : detail::structurally_symmetric(n, Ap, Ai);
  std::unique_ptr<LevelService> service;
  {
    const int service_min = detail::service_threshold();
    const int from = getenv("SPL_ND_GPU_MIN") ? service_min : std::min(service_min, detail::kServiceFromBulky);
    if (make_service && service_min > 0 && n >= from) service = make_service(n, (int64_t)Ap[n] * (pattern_symmetric ? 1 : 2));
  }
  BigVec<int64_t> xadj((size_t)n + 1, 0);
  BigVec<int> adj;
  if (pattern_symmetric) {
    // column ranges on threads: lengths, one sequential prefix sum, then the lists
    unsigned nt = std::thread::hardware_concurrency();
    nt = (Ap[n] < 2000000 || nt < 2) ? 1 : (nt > 8 ? 8 : nt);
    auto on_ranges = [&](auto body) {
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; ++t) {
        const int j0 = (int)((int64_t)n * t / nt), j1 = (int)((int64_t)n * (t + 1) / nt);
        try {
          pool.emplace_back(body, j0, j1);
        } catch (...) {
          body(j0, j1);
        }
      }
      body(0, (int)((int64_t)n / nt));
      for (std::thread &th : pool) th.join();
    };
    on_ranges([&](int j0, int j1) {
      for (int j = j0; j < j1; ++j) {
        int64_t len = 0;
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) len += Ai[p] != j;
        xadj[(size_t)j + 1] = len;
      }
    });
    for (int j = 0; j < n; ++j) xadj[(size_t)j + 1] += xadj[(size_t)j];
    adj.resize((size_t)xadj[(size_t)n]);
    on_ranges([&](int j0, int j1) {
      for (int j = j0; j < j1; ++j) {
        int64_t q = xadj[(size_t)j];
        for (int p = Ap[j]; p < Ap[j + 1]; ++p)
          if (Ai[p] != j) adj[(size_t)q++] = Ai[p];
      }
    });
  } else {
    for (int j = 0; j < n; ++j)
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int i = Ai[p];
        if (i != j) { ++xadj[(size_t)i + 1]; ++xadj[(size_t)j + 1]; }
      }
    for (int i = 0; i < n; ++i) xadj[(size_t)i + 1] += xadj[(size_t)i];
    adj.resize((size_t)xadj[(size_t)n]);
    std::vector<int64_t> cur(xadj.begin(), xadj.end() - 1);
    for (int j = 0; j < n; ++j)
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int i = Ai[p];
        if (i != j) { adj[(size_t)cur[(size_t)i]++] = j; adj[(size_t)cur[(size_t)j]++] = i; }
      }
  }
  lap(pattern_symmetric ? "adjacency (symmetric pattern)" : "adjacency of A + A^T");
  detail::Shared shared(n, xadj, adj, leaf);
  if (service && n < shared.service_min) {  // announced for a bulky graph only
    shared.root_bulky = detail::bulky_graph(n, xadj, adj);
    if (!shared.root_bulky) service.reset();
  }
  if (service) {
    if (!service->graph(xadj.data(), adj.data())) service.reset();
    shared.service = service.get();
    lap(service ? "graph to the GPU" : "no GPU level service");
  }
  std::vector<detail::Node> nodes;
  if (n > 0) {
    detail::Worker w(shared);
    nodes = w.dissect(0, n, 0);
    T.root_levels = shared.root_levels;
    T.gave_up = shared.gave_up.load(std::memory_order_relaxed) != 0;
  }
  lap("dissection");
  // the node vector is a post-order: children before their parent
  const int nf = (int)nodes.size();
  T.nfronts = nf;
  T.parent.assign((size_t)nf, -1);
  T.slot.assign((size_t)nf, 0);
  T.depth.assign((size_t)nf, 0);
  T.p0.assign((size_t)nf, 0);
  T.np.assign((size_t)nf, 0);
  T.nb.assign((size_t)nf, 0);
  T.perm.assign((size_t)n, 0);
  T.inv.assign((size_t)n, 0);
  T.front_of.assign((size_t)n, 0);
  int next = 0;
  for (int f = 0; f < nf; ++f) {
    const detail::Node &nd = nodes[(size_t)f];
    if (nd.left >= 0) { T.parent[(size_t)nd.left] = f; T.slot[(size_t)nd.left] = 0; }
    if (nd.right >= 0) { T.parent[(size_t)nd.right] = f; T.slot[(size_t)nd.right] = nd.left >= 0 ? 1 : 0; }
    T.p0[(size_t)f] = next;
    T.np[(size_t)f] = (int)nd.piv.size();
    for (int v : nd.piv) {
      T.perm[(size_t)next] = v;
      T.inv[(size_t)v] = next;
      T.front_of[(size_t)next] = f;
      ++next;
    }
  }
  for (int f = nf - 1; f >= 0; --f)
    T.depth[(size_t)f] = T.parent[(size_t)f] < 0 ? 0 : T.depth[(size_t)T.parent[(size_t)f]] + 1;
  T.maxdepth = 0;
  for (int f = 0; f < nf; ++f) T.maxdepth = std::max(T.maxdepth, T.depth[(size_t)f]);
  T.by_depth.assign((size_t)T.maxdepth + 1, std::vector<int>());
  for (int f = 0; f < nf; ++f) T.by_depth[(size_t)T.depth[(size_t)f]].push_back(f);
  // boundaries, bottom-up: later-eliminated neighbours of the pivots + what the children pass on.
  // The fronts of one tree level are independent (they only read their children's lists), so each
  // level is shared out among threads; the lists are concatenated in front order afterwards.
  T.bptr.assign((size_t)nf + 1, 0);
  lap("permutation, levels");
  auto host_boundaries = [&] {
    std::vector<std::vector<int>> bnd((size_t)nf);
    auto boundary_of = [&](int f) {
      const detail::Node &nd = nodes[(size_t)f];
      const int last = T.p0[(size_t)f] + T.np[(size_t)f];
      std::vector<int> &b = bnd[(size_t)f];
      // the pivots' own later neighbours: few, sorted here; the children's lists arrive sorted and are merged in
      std::vector<int> own;
      for (int g = T.p0[(size_t)f]; g < last; ++g) {
        const int v = T.perm[(size_t)g];
        for (int64_t p = xadj[(size_t)v]; p < xadj[(size_t)v + 1]; ++p) {
          const int h = T.inv[(size_t)adj[(size_t)p]];
          if (h >= last) own.push_back(h);
        }
      }
      std::sort(own.begin(), own.end());
      own.erase(std::unique(own.begin(), own.end()), own.end());
      const std::vector<int> none;
      const std::vector<int> &l = nd.left >= 0 ? bnd[(size_t)nd.left] : none, &r = nd.right >= 0 ? bnd[(size_t)nd.right] : none;
      // three-way merge without duplicates of what lies beyond this front's pivots
      size_t a = std::lower_bound(l.begin(), l.end(), last) - l.begin(), c = std::lower_bound(r.begin(), r.end(), last) - r.begin(), o = 0;
      b.reserve(own.size() + (l.size() - a) + (r.size() - c));
      const int kEnd = 0x7fffffff;
      for (;;) {
        const int x = a < l.size() ? l[a] : kEnd, y = c < r.size() ? r[c] : kEnd, z = o < own.size() ? own[o] : kEnd;
        const int m = std::min(x, std::min(y, z));
        if (m == kEnd) break;
        b.push_back(m);
        if (x == m) ++a;
        if (y == m) ++c;
        if (z == m) ++o;
      }
    };
    // One team of threads for all levels (round 4: a pool spawned and joined per level cost more than the work of most
    // of the 26 levels of a 10^6-unknown mesh): the main thread publishes a level, everyone takes chunks of its fronts,
    // the level ends when all have reported; levels with little work stay with the main thread.
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = (int)std::min<unsigned>(hw ? hw : 1, 16);
    std::atomic<int> generation{0}, reported{0}, stop{0};
    std::atomic<size_t> next_item{0};
    const std::vector<int> *cur = nullptr;
    size_t chunk = 1;
    auto take_chunks = [&] {
      const std::vector<int> &L = *cur;
      for (size_t i = next_item.fetch_add(chunk); i < L.size(); i = next_item.fetch_add(chunk))
        for (size_t k = i; k < std::min(i + chunk, L.size()); ++k) boundary_of(L[k]);
    };
    std::vector<std::thread> team;
    // The team is stopped and joined on EVERY way out of this block (a bad_alloc in boundary_of on the main
    // thread destroyed joinable threads -> std::terminate instead of UMFPACK_ERROR_out_of_memory); what a worker throws
    // is kept and rethrown on the main thread after the level's barrier.
    std::exception_ptr worker_error;
    std::mutex worker_error_mu;
    struct StopTeam {
      std::atomic<int> &stop;
      std::vector<std::thread> &team;
      ~StopTeam() {
        stop.store(1, std::memory_order_release);
        for (std::thread &th : team)
          if (th.joinable()) th.join();
      }
    } stop_team{stop, team};
    if (nthreads > 1 && nf >= 512)
      for (int w = 1; w < nthreads; ++w) {
        try {
          team.emplace_back([&] {
            int seen = 0;
            for (;;) {
              int now, spins = 0;
              while ((now = generation.load(std::memory_order_acquire)) == seen) {
                if (stop.load(std::memory_order_acquire)) return;
                if (++spins > 200) { std::this_thread::yield(); spins = 0; }
              }
              seen = now;
              try {
                take_chunks();
              } catch (...) {
                std::lock_guard<std::mutex> lk(worker_error_mu);
                if (!worker_error) worker_error = std::current_exception();
              }
              reported.fetch_add(1, std::memory_order_release);
            }
          });
        } catch (...) {
          break;
        }
      }
    for (int d = T.maxdepth; d >= 0; --d) {
      const std::vector<int> &L = T.by_depth[(size_t)d];
      int64_t pivots = 0;
      for (int f : L) pivots += T.np[(size_t)f];
      if (team.empty() || L.size() < 2 || (L.size() < 8 && pivots < 4096)) {
        for (int f : L) boundary_of(f);
        continue;
      }
      // (the few fronts of the top levels are the largest: one each per thread; lower down in chunks)
      cur = &L;
      chunk = L.size() < 64 ? 1 : 16;
      next_item.store(0, std::memory_order_relaxed);
      reported.store(0, std::memory_order_relaxed);
      generation.fetch_add(1, std::memory_order_release);
      std::exception_ptr mine;
      try {
        take_chunks();
      } catch (...) {
        mine = std::current_exception();
        next_item.store(L.size(), std::memory_order_relaxed);  // nothing more to hand out: the level ends at once
      }
      while (reported.load(std::memory_order_acquire) < (int)team.size()) std::this_thread::yield();
      if (mine) std::rethrow_exception(mine);
      if (worker_error) std::rethrow_exception(worker_error);
    }
    stop.store(1, std::memory_order_release);  // (the guard above does the same on the exceptional ways out)
    for (std::thread &th : team) th.join();
    for (int f = 0; f < nf; ++f) {
      T.nb[(size_t)f] = (int)bnd[(size_t)f].size();
      T.bptr[(size_t)f + 1] = T.bptr[(size_t)f] + (int64_t)bnd[(size_t)f].size();
    }
    T.bidx.resize((size_t)T.bptr[(size_t)nf]);
    for (int f = 0; f < nf; ++f) std::copy(bnd[(size_t)f].begin(), bnd[(size_t)f].end(), T.bidx.begin() + T.bptr[(size_t)f]);
  };
  // On the device when the graph is there already (nd_levels.hip: a bitmap per front over its ancestors' pivots, own
  // neighbours set by one pass over the edges, children merged by shifts level by level: 18 -> 4 ms at 10^6 unknowns,
  // 146 ms at 8 10^6); SPL_ND_BOUNDARIES=host keeps it here, =check makes both and compares them (tests).
  {
    const char *mode = getenv("SPL_ND_BOUNDARIES");
    const bool host_only = mode && strcmp(mode, "host") == 0, check = mode && strcmp(mode, "check") == 0;
    bool done = false;
    if (service && !host_only) {
      try {
        done = service->boundaries(T);
      } catch (...) {
        done = false;
      }
    }
    if (done && check) {
      std::vector<int> nb_d, bidx_d;
      std::vector<int64_t> bptr_d;
      nb_d.swap(T.nb);
      bptr_d.swap(T.bptr);
      bidx_d.swap(T.bidx);
      T.nb.assign((size_t)nf, 0);
      T.bptr.assign((size_t)nf + 1, 0);
      host_boundaries();
      if (nb_d != T.nb || bptr_d != T.bptr || bidx_d != T.bidx)
        throw std::logic_error("mf::build_tree: the device's boundary lists differ from the host's");
      if (timing) fprintf(stderr, "[build_tree] boundaries: device == host (%zu indices)\n", T.bidx.size());
    } else if (!done) {
      if (check && service) throw std::logic_error("mf::build_tree: SPL_ND_BOUNDARIES=check and the device made no boundary lists");
      host_boundaries();
    }
    if (timing) fprintf(stderr, "[build_tree] boundaries made %s\n", done ? "on the device" : "on the host");
  }
  lap("boundaries");
  if (mult > 1) {
    if (unexpanded) {
      *unexpanded = T;
      layout_tree(*unexpanded);
    }
    expand_tree(T, mult);
    lap("expansion");
  }
  layout_tree(T);
}

// every vertex replaced by its mult unknowns, which stay together in their front
inline void expand_tree(Tree &T, int mult) {
  const int n = T.n, nf = T.nfronts;
  {
    const size_t m = (size_t)mult;
    std::vector<int> perm((size_t)n * m), inv((size_t)n * m), front_of((size_t)n * m), bidx(T.bidx.size() * m);
    for (size_t k = 0; k < (size_t)n; ++k)
      for (size_t h = 0; h < m; ++h) {
        perm[k * m + h] = (int)((size_t)T.perm[k] * m + h);
        front_of[k * m + h] = T.front_of[k];
      }
    for (size_t k = 0; k < (size_t)n * m; ++k) inv[(size_t)perm[k]] = (int)k;
    for (size_t k = 0; k < T.bidx.size(); ++k)
      for (size_t h = 0; h < m; ++h) bidx[k * m + h] = (int)((size_t)T.bidx[k] * m + h);
    T.perm.swap(perm);
    T.inv.swap(inv);
    T.front_of.swap(front_of);
    T.bidx.swap(bidx);
    for (int f = 0; f < nf; ++f) {
      T.p0[(size_t)f] *= mult;
      T.np[(size_t)f] *= mult;
      T.nb[(size_t)f] *= mult;
    }
    for (int f = 0; f <= nf; ++f) T.bptr[(size_t)f] *= mult;
    T.n = n * mult;
  }
}

inline void layout_tree(Tree &T) {
  const int nf = T.nfronts;
  T.front_elems = T.panel_elems = T.inv_elems = T.work_elems = T.rel_elems = 0;
  T.region_elems[0] = T.region_elems[1] = 0;
  T.flops = 0.0;
  // storage layout and work estimate
  T.ld.assign((size_t)nf, 0);
  T.foff.assign((size_t)nf, 0);
  T.poff.assign((size_t)nf, 0);
  T.uoff.assign((size_t)nf, 0);
  T.ldp.assign((size_t)nf, 0);
  T.ldu.assign((size_t)nf, 0);
  T.ioff.assign((size_t)nf, 0);
  T.woff.assign((size_t)nf, 0);
  T.roff.assign((size_t)nf, 0);
  T.level_elems.assign((size_t)T.maxdepth + 1, 0);
  // Leading dimensions are odd multiples of 16 doubles: a column then starts on a 128-byte line, so the 64 rows a
  // wavefront takes of it are four whole lines (round 4: with ld = fs a 512-byte segment straddled five — the PMC pass
  // of the solve kernels read 1.19x the panels' bytes, tools/pmc_solve.sh), and the column stride is no multiple of 256
  // bytes (the accesses along a row of the transposed kernels spread over the channels).
  auto pad_ld = [](int64_t v) {
    v = (std::max<int64_t>(v, 1) + 15) / 16 * 16;
    return v % 32 == 0 ? v + 16 : v;
  };
  for (int f = 0; f < nf; ++f) {
    const int64_t fs = T.fs(f), p = T.np[(size_t)f], q = T.nb[(size_t)f];
    const int64_t ld = pad_ld(fs);
    T.ld[(size_t)f] = (int)ld;
    int64_t &lev = T.level_elems[(size_t)T.depth[(size_t)f]];
    T.foff[(size_t)f] = lev;
    lev += (ld * std::max<int64_t>(fs, 1) + 15) / 16 * 16;
    T.front_elems += (ld * std::max<int64_t>(fs, 1) + 15) / 16 * 16;
    T.ldp[(size_t)f] = (int)pad_ld(fs);
    T.ldu[(size_t)f] = (int)pad_ld(p);
    T.poff[(size_t)f] = T.panel_elems;
    T.panel_elems += ((int64_t)T.ldp[(size_t)f] * p + 15) / 16 * 16;
    T.uoff[(size_t)f] = T.panel_elems;
    T.panel_elems += ((int64_t)T.ldu[(size_t)f] * q + 15) / 16 * 16;
    T.ioff[(size_t)f] = T.inv_elems;
    T.inv_elems += ((p + kBlock - 1) / kBlock) * 2 * kBlock * kBlock;
    T.woff[(size_t)f] = T.work_elems;
    T.work_elems += fs;
    T.roff[(size_t)f] = T.rel_elems;
    T.rel_elems += q;
    T.flops += 2.0 / 3.0 * (double)p * p * p + 2.0 * (double)p * p * q + 2.0 * (double)p * q * q;
  }
  for (int d = 0; d <= T.maxdepth; ++d)
    T.region_elems[d & 1] = std::max(T.region_elems[d & 1], T.level_elems[(size_t)d]);
}

}  // namespace mf
}  // namespace spl
