// edt_multi.hip -- ONE process driving SEVERAL GPUs behind the C ABI (host buffers in, host buffers out).
//
// north_star: "host code stays C++ ... volumes shard along Z across the 8 GPUs of one node".  edt/distributed.py is
// the one-process-per-GPU form over torch.distributed / RCCL; this file is the same partition for a C++ (or
// ctypes / Cython) host that just calls edt::edt<T>() and owns no communicator: a host thread per device, the
// slab-record phases of edt_api.hip on every device, and the ONE exchange between them as peer-to-peer copies
// over xGMI (hipMemcpyPeerAsync: every ordered pair of devices is its own transfer on its own link).
//
//   device g:  labels[z in Z_g] (+ the slice below: the one-slice halo)  --H2D-->
//              X and Y passes on the slab, written as per-destination slab records        (edt_hip_shard_xy_records_device)
//              its own records straight into its receive buffer, the others to peers      (hipMemcpyPeerAsync x (n-1))
//              Z pass over the gathered records [all z][its y rows]                        (edt_hip_shard_z_records_device)
//              rows [ys_g, ye_g) of every xy-slice  --one strided D2H-->  the caller's output
//
// The same device ordinal may appear several times in the list ("virtual devices"): that is how the whole driver is
// tested on a one-GPU box.  Volumes the slab-record form does not cover (sx > 1024, sy or sz > 2048, fewer y words or
// z slices than devices) run on the first device alone.
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

namespace {

struct Barrier {  // (C++17: no std::barrier)
  std::mutex m;
  std::condition_variable cv;
  int count, waiting = 0, phase = 0;
  explicit Barrier(int n) : count(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int ph = phase;
    if (++waiting == count) {
      waiting = 0;
      ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != ph; });
    }
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

}  // namespace

bool multi_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, int n) {
  if (n < 2) return false;
  if (!edt_hip_shard_records_supported(dtype, sx, sy, sz)) return false;
  return sz >= n && ceil_div(sy, kBandRows) >= n;
}

int run_multi(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
              int flags, float *output, const int *devices, int n) {
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
    hipStream_t stream = nullptr;
    void *d_labels = nullptr, *d_halo = nullptr, *d_ws = nullptr;
    float *d_recv = nullptr;
    std::vector<float *> d_send(n, nullptr);
    MULTI_TRY(hipSetDevice(dev));
    MULTI_TRY(hipStreamCreate(&stream));
    for (int h = 0; h < n && rc == EDT_OK; ++h) {  // peer access where the ordinals differ (ignore "already enabled")
      if (devices[h] == dev) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, dev, devices[h]) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(devices[h], 0);
        if (e != hipSuccess) (void)hipGetLastError();
      }
    }
    const size_t wbytes = std::max(edt_hip_shard_records_workspace_bytes(dtype, sx, sy, szl),
                                   edt_hip_shard_records_workspace_bytes(EDT_U8, sx, ylen, sz));
    MULTI_TRY(hipMalloc(&d_labels, (size_t)(szl * sxy) * esize));
    if (g > 0) MULTI_TRY(hipMalloc(&d_halo, (size_t)sxy * esize));
    MULTI_TRY(hipMalloc(&d_ws, wbytes));
    MULTI_TRY(hipMalloc((void **)&d_recv, (size_t)(sz * rec[g]) * sizeof(float)));
    for (int h = 0; h < n; ++h)
      if (h != g) MULTI_TRY(hipMalloc((void **)&d_send[h], (size_t)(szl * rec[h]) * sizeof(float)));
    sh.recv[g] = d_recv;
    // ---- labels up (this slab, and the slice below it: the one-slice halo) ----
    const char *src = static_cast<const char *>(labels);
    if (rc == EDT_OK) {
      MULTI_TRY(hipMemcpyAsync(d_labels, src + (size_t)(zs * sxy) * esize, (size_t)(szl * sxy) * esize,
                               hipMemcpyHostToDevice, stream));
      if (g > 0)
        MULTI_TRY(hipMemcpyAsync(d_halo, src + (size_t)((zs - 1) * sxy) * esize, (size_t)sxy * esize,
                                 hipMemcpyHostToDevice, stream));
    }
    // ---- X and Y passes of the slab -> per-destination records ----
    if (rc == EDT_OK) {
      std::vector<void *> blocks(n);
      for (int h = 0; h < n; ++h) blocks[h] = h == g ? (void *)(d_recv + zs * rec[g]) : (void *)d_send[h];
      const int r = edt_hip_shard_xy_records_device(d_labels, d_halo, dtype, sx, sy, szl, wx, wy,
                                                    flags & EDT_FLAG_BLACK_BORDER, n, y_splits.data(), blocks.data(),
                                                    d_ws, wbytes, stream);
      if (r != EDT_OK) { rc = r; err = edt_hip_last_error(); }
    }
    if (stream) MULTI_TRY(hipStreamSynchronize(stream));
    sh.rc[g] = rc;
    barrier.wait();  // every receive buffer exists, every slab's records are written
    bool all_ok = true;
    for (int h = 0; h < n; ++h) all_ok = all_ok && sh.rc[h] == EDT_OK;
    // ---- the exchange: my records for destination h go to rows [zs, ze) of h's receive buffer ----
    if (all_ok) {
      for (int k = 1; k < n; ++k) {
        const int h = (g + k) % n;  // (start at different peers so that the links are used evenly)
        MULTI_TRY(hipMemcpyPeerAsync(sh.recv[h] + zs * rec[h], devices[h], d_send[h], dev,
                                     (size_t)(szl * rec[h]) * sizeof(float), stream));
      }
      MULTI_TRY(hipStreamSynchronize(stream));
    }
    sh.rc[g] = rc;
    barrier.wait();  // every record has arrived
    all_ok = true;
    for (int h = 0; h < n; ++h) all_ok = all_ok && sh.rc[h] == EDT_OK;
    // ---- Z pass over [all z][my y rows], then my rows of every slice back to the host ----
    if (all_ok) {
      const int r = edt_hip_shard_z_records_device(d_recv, sx, ylen, sz, wz, flags & (EDT_FLAG_BLACK_BORDER | EDT_FLAG_SQRT),
                                                   d_ws, wbytes, stream);
      if (r != EDT_OK) { rc = r; err = edt_hip_last_error(); }
      if (rc == EDT_OK)
        MULTI_TRY(hipMemcpy2DAsync(output + yparts[g].lo * sx, (size_t)sxy * sizeof(float), d_recv,
                                   (size_t)rec[g] * sizeof(float), (size_t)(ylen * sx) * sizeof(float), (size_t)sz,
                                   hipMemcpyDeviceToHost, stream));
      MULTI_TRY(hipStreamSynchronize(stream));
    }
    barrier.wait();  // nobody frees a buffer a peer may still be copying from
    for (auto p : d_send) if (p) (void)hipFree(p);
    if (d_recv) (void)hipFree(d_recv);
    if (d_ws) (void)hipFree(d_ws);
    if (d_halo) (void)hipFree(d_halo);
    if (d_labels) (void)hipFree(d_labels);
    if (stream) (void)hipStreamDestroy(stream);
    sh.rc[g] = rc;
    sh.msg[g] = err;
  };

  int prev_dev = 0;
  (void)hipGetDevice(&prev_dev);
  std::vector<std::thread> threads;
  for (int g = 0; g < n; ++g) threads.emplace_back(worker, g);
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
