// edt_multi.hip -- ONE process driving SEVERAL GPUs behind the C ABI (host buffers in, host buffers out).
//
// north_star: "host code stays C++ ... volumes shard along Z across the 8 GPUs of one node".  edt/distributed.py is
// the one-process-per-GPU form over torch.distributed / RCCL; this file is the same partition for a C++ (or
// ctypes / Cython) host that just calls edt::edt<T>() and owns no communicator: a host thread per device, the
// slab-record phases of edt_shard_api.hip on every device, and the ONE exchange between them as peer-to-peer copies
// over xGMI (hipMemcpyPeerAsync: every ordered pair of devices is its own transfer on its own link).
//
//   device g:  labels[z in Z_g] (+ the slice below: the one-slice halo)  --H2D-->
//              X and Y passes on the slab, written as per-destination slab records        (edt_hip_shard_xy_records_device)
//              its own records straight into its receive buffer, the others to peers      (hipMemcpyPeerAsync x (n-1))
//              Z pass over the gathered records [all z][its y rows]                        (edt_hip_shard_z_records_device)
//              rows [ys_g, ye_g) of every xy-slice  --one strided D2H-->  the caller's output
//
// The slab is processed in z-chunks (default 4, EDT_HIP_MULTI_CHUNKS): the peer copies of chunk k run on a copy stream
// under the X / Y kernels of chunk k+1, so only the last chunk's exchange is exposed.  Device buffers, streams and
// events live in a per-slot pool between calls (edt_hip_release_cache frees them): a transform allocates nothing once
// the pool is warm.  Distinct ordinals need peer access (xGMI); a pair without it is an error naming the pair
// (EDT_HIP_ALLOW_STAGED_PEER=1 accepts the runtime's staging through host memory instead).
//
// The same device ordinal may appear several times in the list ("virtual devices"): that is how the whole driver is
// tested on a one-GPU box.  Volumes the slab-record form does not cover (sx, sy or sz > 2048, fewer y words or
// z slices than devices): edt_hip_edt3dsq_multi reports EDT_ERR_UNSUPPORTED (query: edt_hip_multi_supported); the
// edt_hip_set_devices route runs them on the first listed device and says so once on stderr.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

namespace {

struct Barrier {  // (C++17: no std::barrier).  abort() releases every waiter, now and later: wait() returns false
  std::mutex m;
  std::condition_variable cv;
  int count, waiting = 0, phase = 0;
  std::atomic<bool> broken{false};  // (also read outside the mutex, by the launching thread after the join)
  explicit Barrier(int n) : count(n) {}
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const int ph = phase;
    if (++waiting == count) {
      waiting = 0;
      ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != ph || broken; });
    }
    return !broken;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
};

struct Range { int64_t lo, hi; };

std::vector<Range> balanced(int64_t n, int parts) {
  std::vector<Range> out;
  const int64_t base = n / parts, extra = n % parts;
  int64_t at = 0;
  for (int i = 0; i < parts; ++i) {
    const int64_t len = base + (i < extra ? 1 : 0);
    out.push_back({at, at + len});
    at += len;
  }
  return out;
}

struct Shared {
  std::vector<float *> recv;      // per device: its receive buffer [sz][rec]
  std::vector<int> rc;            // per device: first error
  std::vector<std::string> msg;
};

#define MULTI_TRY(expr)                                                        \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess && rc == EDT_OK) {                                    \
      rc = EDT_ERR_HIP;                                                        \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
    }                                                                          \
  } while (0)

// Per-slot resources kept between calls.  A slot is a POSITION in the device list (the same ordinal may back several
// slots: virtual devices), so buffers are never shared between two workers of one call.
struct Slot {
  int dev = -1;
  hipStream_t compute = nullptr, copy = nullptr;
  std::vector<hipEvent_t> chunk_done;
  struct Buf { void *p = nullptr; size_t cap = 0; };
  Buf labels, halo, ws, recv;
  std::vector<Buf> send;
  void release() {  // (with the owning device current)
    auto drop = [](Buf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    drop(labels); drop(halo); drop(ws); drop(recv);
    for (auto &b : send) drop(b);
    for (auto e : chunk_done) (void)hipEventDestroy(e);
    chunk_done.clear();
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
    compute = copy = nullptr;
    dev = -1;
  }
};
std::mutex g_multi_mutex;          // one multi-device transform at a time per process
std::vector<Slot> g_slots;

hipError_t grow(Slot::Buf &b, size_t bytes) {
  if (bytes == 0) bytes = 256;
  if (b.cap >= bytes) return hipSuccess;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  const hipError_t e = hipMalloc(&b.p, bytes);
  if (e == hipSuccess) b.cap = bytes;
  else b.p = nullptr;
  return e;
}

int multi_chunks(int64_t min_slab) {
  int want = 4;
  if (const char *e = std::getenv("EDT_HIP_MULTI_CHUNKS")) want = std::atoi(e);
  if (want < 1) want = 1;
  return (int)std::min<int64_t>(want, std::max<int64_t>(1, min_slab));
}

}  // namespace

bool multi_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, int n) {
  if (n < 2) return false;
  if (!edt_hip_shard_records_supported(dtype, sx, sy, sz)) return false;
  return sz >= n && ceil_div(sy, kBandRows) >= n;
}

void multi_release() {
  std::lock_guard<std::mutex> lock(g_multi_mutex);
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
  for (auto &s : g_slots) {
    if (s.dev < 0) continue;
    if (hipSetDevice(s.dev) != hipSuccess) { (void)hipGetLastError(); continue; }
    s.release();
  }
  g_slots.clear();
  (void)hipSetDevice(cur);
}

int run_multi(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
              int flags, float *output, const int *devices, int n) {
  std::lock_guard<std::mutex> call_lock(g_multi_mutex);
  const int esize = dtype_size(dtype);
  const int64_t sxy = sx * sy;
  const std::vector<Range> zparts = balanced(sz, n);
  std::vector<Range> yparts = balanced(ceil_div(sy, kBandRows), n);   // in 32-row words
  for (auto &r : yparts) { r.lo *= kBandRows; r.hi = std::min<int64_t>(r.hi * kBandRows, sy); }
  std::vector<int64_t> y_splits, rec;
  for (const auto &r : yparts) {
    y_splits.push_back(r.lo);
    rec.push_back((int64_t)edt_hip_shard_record_floats(sx, r.hi - r.lo));
  }
  y_splits.push_back(sy);
  int64_t min_slab = sz;
  for (const auto &r : zparts) min_slab = std::min(min_slab, r.hi - r.lo);
  const int nchunks = multi_chunks(min_slab);

  // peer access between distinct ordinals, checked up front: a pair without it would be staged through host memory
  // by the runtime -- an order of magnitude slower than xGMI and not what this route is for
  {
    const char *allow = std::getenv("EDT_HIP_ALLOW_STAGED_PEER");
    const bool staged_ok = allow && allow[0] == '1';
    for (int a = 0; a < n && !staged_ok; ++a)
      for (int b = 0; b < n; ++b) {
        if (devices[a] == devices[b]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) {
          (void)hipGetLastError();
          set_error("devices " + std::to_string(devices[a]) + " and " + std::to_string(devices[b]) +
                    " have no peer access (the exchange would be staged through host memory; "
                    "EDT_HIP_ALLOW_STAGED_PEER=1 accepts that)");
          return EDT_ERR_UNSUPPORTED;
        }
      }
  }

  if ((int)g_slots.size() < n) g_slots.resize(n);
  Shared sh;
  sh.recv.assign(n, nullptr);
  sh.rc.assign(n, EDT_OK);
  sh.msg.assign(n, "");
  Barrier barrier(n);
  const int caller_mode = debug_mode();

  auto worker = [&](int g) {
    set_thread_debug_mode(caller_mode);  // (the diagnostics mode is thread-local)
    int rc = EDT_OK;
    std::string err;
    const int dev = devices[g];
    const int64_t zs = zparts[g].lo, ze = zparts[g].hi, szl = ze - zs;
    const int64_t ylen = yparts[g].hi - yparts[g].lo;
    Slot &slot = g_slots[g];
    if (slot.dev >= 0 && slot.dev != dev) {  // the slot served another device last time
      if (hipSetDevice(slot.dev) == hipSuccess) slot.release();
      else (void)hipGetLastError();
    }
    MULTI_TRY(hipSetDevice(dev));
    slot.dev = dev;
    if (!slot.compute) MULTI_TRY(hipStreamCreateWithFlags(&slot.compute, hipStreamNonBlocking));
    if (!slot.copy) MULTI_TRY(hipStreamCreateWithFlags(&slot.copy, hipStreamNonBlocking));
    while ((int)slot.chunk_done.size() < nchunks && rc == EDT_OK) {
      hipEvent_t e = nullptr;
      MULTI_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      if (e) slot.chunk_done.push_back(e);
    }
    for (int h = 0; h < n && rc == EDT_OK; ++h) {  // (checked above; "already enabled" is fine)
      if (devices[h] == dev) continue;
      const hipError_t e = hipDeviceEnablePeerAccess(devices[h], 0);
      if (e != hipSuccess) (void)hipGetLastError();
    }
    const size_t wbytes = std::max(edt_hip_shard_records_workspace_bytes(dtype, sx, sy, szl),
                                   edt_hip_shard_records_workspace_bytes(EDT_U8, sx, ylen, sz));
    MULTI_TRY(grow(slot.labels, (size_t)(szl * sxy) * esize));
    if (g > 0) MULTI_TRY(grow(slot.halo, (size_t)sxy * esize));
    MULTI_TRY(grow(slot.ws, wbytes));
    MULTI_TRY(grow(slot.recv, (size_t)(sz * rec[g]) * sizeof(float)));
    if ((int)slot.send.size() < n) slot.send.resize(n);
    for (int h = 0; h < n; ++h)
      if (h != g) MULTI_TRY(grow(slot.send[h], (size_t)(szl * rec[h]) * sizeof(float)));
    char *d_labels = static_cast<char *>(slot.labels.p);
    float *d_recv = static_cast<float *>(slot.recv.p);
    hipStream_t stream = slot.compute;
    sh.recv[g] = d_recv;
    sh.rc[g] = rc;
    bool all_ok = barrier.wait();  // every receive buffer exists: peers may write into it from now on
    for (int h = 0; h < n; ++h) all_ok = all_ok && sh.rc[h] == EDT_OK;
    // ---- labels up (this slab, and the slice below it: the one-slice halo) ----
    const char *src = static_cast<const char *>(labels);
    if (all_ok) {
      MULTI_TRY(hipMemcpyAsync(d_labels, src + (size_t)(zs * sxy) * esize, (size_t)(szl * sxy) * esize,
                               hipMemcpyHostToDevice, stream));
      if (g > 0)
        MULTI_TRY(hipMemcpyAsync(slot.halo.p, src + (size_t)((zs - 1) * sxy) * esize, (size_t)sxy * esize,
                                 hipMemcpyHostToDevice, stream));
    }
    // ---- X and Y passes chunk by chunk -> per-destination records; the peer copies of chunk k start as soon as its
    // kernels are done and run on the copy stream under the kernels of chunk k+1 ----
    const std::vector<Range> chunks = balanced(szl, nchunks);
    for (int k = 0; k < nchunks && all_ok && rc == EDT_OK; ++k) {
      const int64_t c0 = chunks[k].lo, c1 = chunks[k].hi;
      std::vector<void *> blocks(n);
      for (int h = 0; h < n; ++h)
        blocks[h] = h == g ? (void *)(d_recv + (zs + c0) * rec[g])
                           : (void *)(static_cast<float *>(slot.send[h].p) + c0 * rec[h]);
      // the slice below a later chunk is the last slice of the previous chunk, already on the device
      const void *halo = k == 0 ? (g > 0 ? slot.halo.p : nullptr) : (const void *)(d_labels + (size_t)((c0 - 1) * sxy) * esize);
      const int r = edt_hip_shard_xy_records_device(d_labels + (size_t)(c0 * sxy) * esize, halo, dtype, sx, sy, c1 - c0, wx,
                                                    wy, flags & EDT_FLAG_BLACK_BORDER, n, y_splits.data(), blocks.data(),
                                                    slot.ws.p, wbytes, stream);
      if (r != EDT_OK) { rc = r; err = edt_hip_last_error(); break; }
      MULTI_TRY(hipEventRecord(slot.chunk_done[k], stream));
      MULTI_TRY(hipStreamWaitEvent(slot.copy, slot.chunk_done[k], 0));
      for (int j = 1; j < n && rc == EDT_OK; ++j) {
        const int h = (g + j) % n;  // (start at different peers so that the links are used evenly)
        MULTI_TRY(hipMemcpyPeerAsync(sh.recv[h] + (zs + c0) * rec[h], devices[h],
                                     static_cast<float *>(slot.send[h].p) + c0 * rec[h], dev,
                                     (size_t)((c1 - c0) * rec[h]) * sizeof(float), slot.copy));
      }
    }
    if (slot.compute) MULTI_TRY(hipStreamSynchronize(slot.compute));
    if (slot.copy) MULTI_TRY(hipStreamSynchronize(slot.copy));
    sh.rc[g] = rc;
    all_ok = barrier.wait();  // every record of every device has arrived
    for (int h = 0; h < n; ++h) all_ok = all_ok && sh.rc[h] == EDT_OK;
    // ---- Z pass over [all z][my y rows], then my rows of every slice back to the host ----
    if (all_ok) {
      const int r = edt_hip_shard_z_records_device_w(d_recv, sx, ylen, sz, wx, wy, wz, flags & (EDT_FLAG_BLACK_BORDER | EDT_FLAG_SQRT),
                                                   slot.ws.p, wbytes, stream);
      if (r != EDT_OK) { rc = r; err = edt_hip_last_error(); }
      if (rc == EDT_OK)
        MULTI_TRY(hipMemcpy2DAsync(output + yparts[g].lo * sx, (size_t)sxy * sizeof(float), d_recv,
                                   (size_t)rec[g] * sizeof(float), (size_t)(ylen * sx) * sizeof(float), (size_t)sz,
                                   hipMemcpyDeviceToHost, stream));
      MULTI_TRY(hipStreamSynchronize(stream));
    }
    (void)barrier.wait();  // nobody reuses a buffer a peer may still be copying from
    if (rc == EDT_OK && barrier.broken) { rc = EDT_ERR_NOMEM; err = "another worker could not be started"; }
    sh.rc[g] = rc;
    sh.msg[g] = err;
  };

  int prev_dev = 0;
  (void)hipGetDevice(&prev_dev);
  std::vector<std::thread> threads;
  int started = 0;
  try {
    for (int g = 0; g < n; ++g) { threads.emplace_back(worker, g); ++started; }
  } catch (const std::system_error &) {
    // not every worker could be started: the ones that run must not wait for the missing ones
    barrier.abort();
    for (int g = started; g < n; ++g) { sh.rc[g] = EDT_ERR_NOMEM; sh.msg[g] = "could not start a host thread"; }
  }
  for (auto &t : threads) t.join();
  (void)hipSetDevice(prev_dev);
  for (int g = 0; g < n; ++g)
    if (sh.rc[g] != EDT_OK) {
      set_error("device " + std::to_string(devices[g]) + " (slab " + std::to_string(g) + "): " + sh.msg[g]);
      return sh.rc[g];
    }
  return EDT_OK;
}

}  // namespace edt_amd
