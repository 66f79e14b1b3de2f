// device_pool.hip — where DBuf gets its memory.
//
// Small blocks go straight to hipMalloc / hipFree.  Blocks of kCacheMin bytes and more are kept
// when they are released and handed out again to the next request they fit: on this platform the
// driver wipes released VRAM in the background and a later hipMalloc that lands on memory still
// being wiped waits for it — measured with tools/probe/malloc_probe.hip: 3–6 s for a request that
// reaches into ~100 GB released a moment before, during which no kernel of the process runs.  A
// sparse LU at 8 M unknowns holds 172 GB of factor panels and 83 GB of transient fronts; FEAST
// factors one such matrix per contour point.  Kept blocks are given back to the driver when a
// hipMalloc fails (then retried), by spl_release_cached_memory(), or never kept at all with
// SPL_CACHE_DEVICE_MEMORY=0.
// Blocks between kSmallMin and kCacheMin are kept as well, up to kSmallKeptMax bytes in total: a hipFree
// synchronises the device and costs 0.1-0.2 ms, and an operation like the ordered SpGEMM allocates and releases some
// twenty work arrays of 1-500 MB per call — 3 ms of a 19 ms product on config C4 before this tier existed.
#include <atomic>
#include <chrono>
#include <future>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace spl {
namespace {

constexpr size_t kCacheMin = (size_t)1 << 30;
constexpr size_t kSmallMin = (size_t)256 << 10;
constexpr size_t kSmallKeptMax = (size_t)2 << 30;

struct Pool {
  std::mutex mu;
  struct Block { size_t bytes; int device; };
  std::unordered_map<void *, Block> live;               // big blocks handed out
  std::multimap<size_t, std::pair<void *, int>> kept;   // size -> (block, device)
  size_t kept_bytes = 0;
  size_t kept_small_bytes = 0;  // of which in blocks below kCacheMin
  bool enabled = true;
  Pool() {
    const char *e = getenv("SPL_CACHE_DEVICE_MEMORY");
    if (e && atoi(e) == 0) enabled = false;
  }
  // caller holds mu
  void release_kept(int device) {
    for (auto it = kept.begin(); it != kept.end();) {
      if (device < 0 || it->second.second == device) {
        DeviceGuard g(it->second.second);
        (void)hipFree(it->second.first);
        kept_bytes -= it->first;
        if (it->first < kCacheMin) kept_small_bytes -= it->first;
        it = kept.erase(it);
      } else {
        ++it;
      }
    }
  }
};

// seconds this process has spent inside hipMalloc on behalf of device_alloc (spl_device_alloc_seconds): a request that
// reaches into memory the driver is still wiping waits there, and nothing the process has queued runs meanwhile
// (tools/probe/malloc_overlap_probe.hip) — the one part of a first factorisation that the steady state does not have
std::atomic<long long> g_alloc_ns{0};
hipError_t timed_malloc(void **p, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc(p, bytes);
  g_alloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return e;
}

Pool &pool() {
  static Pool *p = new Pool();  // never destroyed: DBufs of static lifetime may outlive any order
  return *p;
}

}  // namespace

namespace {
void warm_streams_once();  // (below, behind the stream pool it fills)
}  // namespace

void *device_alloc(size_t bytes) {
  warm_streams_once();
  Pool &P = pool();
  void *p = nullptr;
  if (bytes < kSmallMin || !P.enabled) {
    hipError_t e = timed_malloc(&p, bytes);
    if (e == hipErrorOutOfMemory && P.kept_bytes) {
      (void)hipGetLastError();
      std::lock_guard<std::mutex> lk(P.mu);
      P.release_kept(-1);
      e = timed_malloc(&p, bytes);
    }
    SPL_HIP(e);
    return p;
  }
  int device = 0;
  SPL_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lk(P.mu);
  // smallest kept block of this device that holds the request without wasting more than half of it
  for (auto it = P.kept.lower_bound(bytes); it != P.kept.end() && it->first <= bytes + bytes / 2; ++it) {
    if (it->second.second != device) continue;
    p = it->second.first;
    P.live[p] = Pool::Block{it->first, device};
    P.kept_bytes -= it->first;
    if (it->first < kCacheMin) P.kept_small_bytes -= it->first;
    P.kept.erase(it);
    return p;
  }
  hipError_t e = timed_malloc(&p, bytes);
  if (e == hipErrorOutOfMemory && P.kept_bytes) {
    (void)hipGetLastError();
    P.release_kept(-1);
    e = timed_malloc(&p, bytes);
  }
  SPL_HIP(e);
  P.live[p] = Pool::Block{bytes, device};
  return p;
}

void device_free(void *p) noexcept {
  if (!p) return;
  Pool &P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      const Pool::Block b = it->second;
      P.live.erase(it);
      if (b.bytes < kCacheMin && P.kept_small_bytes + b.bytes > kSmallKeptMax) {  // the small tier is full
        (void)hipFree(p);
        return;
      }
      // hipFree waits for the device; a kept block must be just as idle before its next owner
      int cur = 0;
      (void)hipGetDevice(&cur);
      if (cur != b.device) (void)hipSetDevice(b.device);
      (void)hipDeviceSynchronize();
      if (cur != b.device) (void)hipSetDevice(cur);
      P.kept.emplace(b.bytes, std::make_pair(p, b.device));
      P.kept_bytes += b.bytes;
      if (b.bytes < kCacheMin) P.kept_small_bytes += b.bytes;
      return;
    }
  }
  (void)hipFree(p);
}

// free device memory as a planner should see it: what the driver reports plus what this library
// would give back on demand
size_t device_free_bytes() {
  size_t free_b = 0, total_b = 0;
  SPL_HIP(hipMemGetInfo(&free_b, &total_b));
  int device = 0;
  SPL_HIP(hipGetDevice(&device));
  Pool &P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  for (const auto &kv : P.kept)
    if (kv.second.second == device) free_b += kv.first;
  return free_b;
}

double device_alloc_seconds() { return (double)g_alloc_ns.load() * 1e-9; }

namespace {
struct StreamPool {
  std::mutex mu;
  std::vector<std::pair<int, hipStream_t>> idle;
};
StreamPool &stream_pool() {
  static StreamPool *p = new StreamPool();  // never destroyed, like the streams in it
  return *p;
}
}  // namespace

hipStream_t pooled_stream_take(int device) {
  StreamPool &P = stream_pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (size_t i = 0; i < P.idle.size(); ++i)
      if (P.idle[i].first == device) {
        hipStream_t s = P.idle[i].second;
        P.idle.erase(P.idle.begin() + (ptrdiff_t)i);
        return s;
      }
  }
  DeviceGuard g(device);
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return s;
}

void pooled_stream_give(int device, hipStream_t s) {
  if (!s) return;
  StreamPool &P = stream_pool();
  std::lock_guard<std::mutex> lk(P.mu);
  P.idle.emplace_back(device, s);
}

void pooled_streams_prewarm(int device, int count) {
  StreamPool &P = stream_pool();
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(P.mu);
      int have = 0;
      for (const auto &e : P.idle) have += e.first == device;
      if (have >= count) return;
    }
    DeviceGuard g(device);
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    pooled_stream_give(device, s);
  }
}

namespace {
// The first stream a process creates costs ~15 ms (every other one 0.3 - 0.6 ms), and an analysis needs four of them
// 10 ms after it begins (nd_levels.hip), a factorisation seventeen: the streams of the pool are made by a thread of
// their own from the first allocation a process makes on a device — a one-shot umfpack_di_symbolic that follows the
// construction of its matrix finds them there (round 5: 17 - 20 ms of a first analysis at 10^6 unknowns were the wait
// for them).  SPL_PREWARM_STREAMS=0: never.  The jobs are joined when the library is unloaded.
struct StreamWarmers {
  std::mutex mu;
  std::vector<std::pair<int, std::future<void>>> jobs;
  void start(int device) {
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &j : jobs)
      if (j.first == device) return;
    const char *e = getenv("SPL_PREWARM_STREAMS");
    std::future<void> job;
    if (!(e && atoi(e) == 0)) {
      try {
        job = std::async(std::launch::async, [device] { pooled_streams_prewarm(device, 21); });
      } catch (...) {  // no thread to be had: the streams are made where they are needed
      }
    }
    jobs.emplace_back(device, std::move(job));
  }
  ~StreamWarmers() {
    for (auto &j : jobs)
      if (j.second.valid()) j.second.wait();
  }
};
void warm_streams_once() {
  (void)stream_pool();  // constructed before W, so destroyed after the jobs W joins
  static StreamWarmers W;
  static std::atomic<uint64_t> seen{0};  // one bit per device: the common case is a load and a test
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  const uint64_t bit = 1ull << (device & 63);
  if (seen.load(std::memory_order_relaxed) & bit) return;
  seen.fetch_or(bit, std::memory_order_relaxed);
  W.start(device);
}
}  // namespace


size_t device_release_cached() {
  Pool &P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  const size_t had = P.kept_bytes;
  P.release_kept(-1);
  return had;
}

}  // namespace spl
