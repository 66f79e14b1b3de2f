// peer.hip — one-sided exchange of the result blocks of a row-partitioned SpMV (last row of
// §5: "direct peer stores on 7 links in parallel").
//
// The step of the multi-GPU path ends with every rank holding the whole y (so that y can be the next x).
// With RCCL that is one all_gather: a ring over point-to-point xGMI links, whose kernels also take CUs from
// an SpMV kernel that wants one resident workgroup on every CU.  The alternative built here uses no CU at
// all for the data: every rank pushes its own block straight into every peer's copy of y with N - 1
// device-to-device copies on N - 1 streams (one per peer = one per xGMI link, all links busy at once, the
// copy engines do the work), followed on the same stream by a 4-byte store of the step number into the
// peer's flag word for this rank; a one-thread kernel on the compute stream waits until all N - 1 flags of
// the rank show the step.  One process per GPU: peers' buffers are reached through IPC memory handles
// (hipIpcGetMemHandle / hipIpcOpenMemHandle), exchanged by the caller over its own channel
// (torch.distributed in bench.py).
//
// Visibility: the flag words are polled by a kernel while a peer's copy engine writes them, so they
// live in FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained): system-scope loads
// are not served from a stale L2 line); plain hipMalloc is the fallback when the runtime refuses such memory or
// its IPC handle, and spl_peer_exchange_flags_finegrained() tells which one is in use so that a caller can keep
// the fallback out of an unattended run.  The y buffers stay coarse-grained (they are the next x of the SpMV,
// whose gathers live on L2 hits): they are only ever read by kernels launched after the wait kernel has
// finished, i.e. behind a kernel boundary, which is where coarse-grained memory is made coherent.
// SPL_PEER_FINEGRAINED_DATA=1 puts them in fine-grained memory as well (for a box where that boundary proves
// not to be enough).
//
// Reuse: a peer may run ahead by at most one step — it starts step k + 1 only after it has seen this rank's
// flag of step k — so two receive buffers alternate: step k lands in buffer k % 2 while the owner may still
// be reading buffer (k - 1) % 2.
#include "common.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace spl {
namespace {

constexpr uint32_t kPeerMagic = 0x53504C58u;  // "SPLX"

struct PeerExchange {
  uint32_t magic = kPeerMagic;
  int device = 0, rank = 0, world = 1, chunks = 1;
  int64_t n = 0;
  std::vector<int64_t> bounds;          // chunks * world + 1: piece q = c * world + p belongs to rank p, chunk c
  double *buf[2] = {nullptr, nullptr};  // this rank's two copies of y (hipMalloc: IPC needs whole allocations)
  unsigned *flags = nullptr;            // one word per piece: flags[q] = last step whose piece q has landed
  unsigned *error = nullptr;            // set by the wait kernel when it gives up
  std::vector<double *> peer_buf[2];    // the same of every peer (mapped); [rank] = own
  std::vector<unsigned *> peer_flags;
  std::vector<hipStream_t> streams;     // one per peer
  hipEvent_t ready = nullptr;
  unsigned step = 0;
  bool connected = false;
  bool flags_fine = false, data_fine = false;
  // tests only (SPL_PEER_TEST_FAIL_AT_STEP=k[:rank]): the wait of step k gives up at once, as if a peer had not
  // delivered within its bound — the caller's failure path (bench.py: fall back to RCCL, time the region again) can
  // then be exercised without a broken link
  unsigned fail_at_step = 0;
};

// fine-grained device memory whose IPC handle can be taken, or nullptr
void *alloc_finegrained_ipc(size_t bytes) {
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  return p;
}

inline PeerExchange *as_px(void *p) {
  PeerExchange *x = static_cast<PeerExchange *>(p);
  return (x && x->magic == kPeerMagic) ? x : nullptr;
}

// one thread: wait until the flag of every piece not owned by `self` shows at least `step`; bounded (about 2 s)
__global__ void peer_wait_kernel(const unsigned *__restrict__ flags, int npieces, int world, int self, unsigned step,
                                 unsigned *__restrict__ error, int give_up) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (give_up) { *error = 1; return; }  // (tests: PeerExchange::fail_at_step)
  const unsigned long long t0 = wall_clock64();
  for (int q = 0; q < npieces; ++q) {
    if (q % world == self) continue;
    while ((int)(__hip_atomic_load(flags + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
      if (wall_clock64() - t0 > 200000000ull) { *error = 1; return; }
      __builtin_amdgcn_s_sleep(16);
    }
  }
}

}  // namespace
}  // namespace spl

using namespace spl;

extern "C" {

// Allocate the exchange of rank `rank` of `world`: y has n entries cut into chunks * world pieces, piece
// q = c * world + p = [bounds[q], bounds[q+1]) belongs to rank p (chunks = 1: one block per rank; more: the
// pieces of a chunk can be pushed while the kernel of the next chunk runs).  handles_out receives 3 x 64
// bytes (the IPC handles of the two buffers and of the flag array) to be passed to every peer.
int spl_peer_exchange_create(int rank, int world, int chunks, int64_t n, const int64_t *bounds,
                             unsigned char *handles_out, void **X) {
  if (!X || !bounds || !handles_out || world < 1 || chunks < 1 || rank < 0 || rank >= world || n < 0)
    return SPL_ERROR_argument_missing;
  *X = nullptr;
  PeerExchange *px = new (std::nothrow) PeerExchange();
  if (!px) return SPL_ERROR_out_of_memory;
  try {
    SPL_HIP(hipGetDevice(&px->device));
    px->rank = rank;
    px->world = world;
    px->chunks = chunks;
    px->n = n;
    px->bounds.assign(bounds, bounds + (size_t)chunks * world + 1);
    if (const char *tf = getenv("SPL_PEER_TEST_FAIL_AT_STEP")) {
      const char *colon = strchr(tf, ':');
      if (!colon || atoi(colon + 1) == rank) px->fail_at_step = (unsigned)std::max(0, atoi(tf));
    }
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
    const char *fd = getenv("SPL_PEER_FINEGRAINED_DATA");
    const bool want_fine_data = fd && fd[0] == '1';
    px->data_fine = want_fine_data;
    for (int b = 0; b < 2; ++b) {
      if (want_fine_data) px->buf[b] = static_cast<double *>(alloc_finegrained_ipc(bytes));
      if (!px->buf[b]) {
        px->data_fine = false;
        SPL_HIP(hipMalloc(reinterpret_cast<void **>(&px->buf[b]), bytes));
      }
      SPL_HIP(hipMemset(px->buf[b], 0, bytes));
    }
    const size_t npieces = (size_t)chunks * world;
    const char *ff = getenv("SPL_PEER_FINEGRAINED_FLAGS");
    if (!(ff && ff[0] == '0')) px->flags = static_cast<unsigned *>(alloc_finegrained_ipc((npieces + 1) * sizeof(unsigned)));
    px->flags_fine = px->flags != nullptr;
    if (!px->flags) SPL_HIP(hipMalloc(reinterpret_cast<void **>(&px->flags), (npieces + 1) * sizeof(unsigned)));
    SPL_HIP(hipMemset(px->flags, 0, (npieces + 1) * sizeof(unsigned)));
    px->error = px->flags + npieces;
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    SPL_HIP(hipIpcGetMemHandle(&h, px->buf[0]));
    memcpy(handles_out, &h, 64);
    SPL_HIP(hipIpcGetMemHandle(&h, px->buf[1]));
    memcpy(handles_out + 64, &h, 64);
    SPL_HIP(hipIpcGetMemHandle(&h, px->flags));
    memcpy(handles_out + 128, &h, 64);
    SPL_HIP(hipEventCreateWithFlags(&px->ready, hipEventDisableTiming));
  } catch (const DeviceError &e) {
    if (px->buf[0]) (void)hipFree(px->buf[0]);
    if (px->buf[1]) (void)hipFree(px->buf[1]);
    if (px->flags) (void)hipFree(px->flags);
    delete px;
    return e.status;
  }
  *X = px;
  return SPL_OK;
}

// all_handles: world x 192 bytes, entry q = what rank q's create returned
int spl_peer_exchange_connect(void *X, const unsigned char *all_handles) {
  PeerExchange *px = as_px(X);
  if (!px || !all_handles) return SPL_ERROR_invalid_handle;
  try {
    DeviceGuard g(px->device);
    px->peer_buf[0].assign((size_t)px->world, nullptr);
    px->peer_buf[1].assign((size_t)px->world, nullptr);
    px->peer_flags.assign((size_t)px->world, nullptr);
    px->streams.assign((size_t)px->world, nullptr);
    for (int q = 0; q < px->world; ++q) {
      if (q == px->rank) {
        px->peer_buf[0][(size_t)q] = px->buf[0];
        px->peer_buf[1][(size_t)q] = px->buf[1];
        px->peer_flags[(size_t)q] = px->flags;
        continue;
      }
      hipIpcMemHandle_t h;
      void *p = nullptr;
      memcpy(&h, all_handles + (size_t)q * 192, 64);
      SPL_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
      px->peer_buf[0][(size_t)q] = static_cast<double *>(p);
      memcpy(&h, all_handles + (size_t)q * 192 + 64, 64);
      SPL_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
      px->peer_buf[1][(size_t)q] = static_cast<double *>(p);
      memcpy(&h, all_handles + (size_t)q * 192 + 128, 64);
      SPL_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
      px->peer_flags[(size_t)q] = static_cast<unsigned *>(p);
      SPL_HIP(hipStreamCreateWithFlags(&px->streams[(size_t)q], hipStreamNonBlocking));
    }
    px->connected = true;
    return SPL_OK;
  } catch (const DeviceError &e) {
    return e.status;
  }
}

// A step = `chunks` pushes + one finish.  push: this rank's piece of chunk c (d_piece, produced on `stream`)
// goes into every rank's y of the current step; finish: the wait for every other rank's pieces is enqueued on
// `stream`; when it has run, *y_full (device pointer, n doubles, valid until the step after next) holds the
// whole y.  Nothing is synchronised on the host.
int spl_peer_exchange_push(void *X, int chunk, const double *d_piece, void *stream) {
  PeerExchange *px = as_px(X);
  if (!px || !px->connected) return SPL_ERROR_invalid_handle;
  if (!d_piece || chunk < 0 || chunk >= px->chunks) return SPL_ERROR_argument_missing;
  try {
    DeviceGuard g(px->device);
    hipStream_t s = as_stream(stream);
    const unsigned step = px->step + 1;
    const int b = (int)(step & 1u);
    const size_t q = (size_t)chunk * px->world + px->rank;
    const int64_t r0 = px->bounds[q], r1 = px->bounds[q + 1];
    const size_t bytes = (size_t)(r1 - r0) * sizeof(double);
    SPL_HIP(hipEventRecord(px->ready, s));  // the kernel that wrote d_piece
    for (int p = 0; p < px->world; ++p) {
      if (p == px->rank) continue;
      hipStream_t t = px->streams[(size_t)p];
      SPL_HIP(hipStreamWaitEvent(t, px->ready, 0));
      if (bytes) SPL_HIP(hipMemcpyAsync(px->peer_buf[b][(size_t)p] + r0, d_piece, bytes, hipMemcpyDeviceToDevice, t));
      // stream order: the flag lands after the piece
      SPL_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(px->peer_flags[(size_t)p] + q), (int)step, 1, t));
    }
    if (bytes) SPL_HIP(hipMemcpyAsync(px->buf[b] + r0, d_piece, bytes, hipMemcpyDeviceToDevice, s));
    return SPL_OK;
  } catch (const DeviceError &e) {
    return e.status;
  }
}

int spl_peer_exchange_finish(void *X, void *stream, double **y_full) {
  PeerExchange *px = as_px(X);
  if (!px || !px->connected) return SPL_ERROR_invalid_handle;
  if (!y_full) return SPL_ERROR_argument_missing;
  try {
    DeviceGuard g(px->device);
    hipStream_t s = as_stream(stream);
    const unsigned step = ++px->step;
    if (px->world > 1)
      hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, px->flags, px->chunks * px->world, px->world, px->rank,
                         step, px->error, px->fail_at_step != 0 && step == px->fail_at_step ? 1 : 0);
    *y_full = px->buf[step & 1u];
    SPL_HIP(hipGetLastError());
    return SPL_OK;
  } catch (const DeviceError &e) {
    return e.status;
  }
}

// 1: the flag words are in fine-grained device memory; 0: plain hipMalloc (fallback, see the header comment)
int spl_peer_exchange_flags_finegrained(void *X) {
  PeerExchange *px = as_px(X);
  if (!px) return SPL_ERROR_invalid_handle;
  return px->flags_fine ? 1 : 0;
}

// 1 if a wait gave up (a peer did not deliver within ~2 s): the results since are incomplete
int spl_peer_exchange_failed(void *X) {
  PeerExchange *px = as_px(X);
  if (!px) return SPL_ERROR_invalid_handle;
  unsigned e = 0;
  if (hipMemcpy(&e, px->error, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return SPL_ERROR_device;
  return e ? 1 : 0;
}

void spl_peer_exchange_free(void **X) {
  if (!X || !*X) return;
  PeerExchange *px = as_px(*X);
  *X = nullptr;
  if (!px) return;
  int prev = -1;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  (void)hipSetDevice(px->device);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < px->world && px->connected; ++q) {
    if (q == px->rank) continue;
    if (px->peer_buf[0][(size_t)q]) (void)hipIpcCloseMemHandle(px->peer_buf[0][(size_t)q]);
    if (px->peer_buf[1][(size_t)q]) (void)hipIpcCloseMemHandle(px->peer_buf[1][(size_t)q]);
    if (px->peer_flags[(size_t)q]) (void)hipIpcCloseMemHandle(px->peer_flags[(size_t)q]);
    if (px->streams[(size_t)q]) (void)hipStreamDestroy(px->streams[(size_t)q]);
  }
  if (px->ready) (void)hipEventDestroy(px->ready);
  (void)hipFree(px->buf[0]);
  (void)hipFree(px->buf[1]);
  (void)hipFree(px->flags);
  px->magic = 0;
  delete px;
  if (have_prev) (void)hipSetDevice(prev);
}

}  // extern "C"
