// edt_host.hip -- the C ABI (include/edt_hip.h), part 2 of 3: the host-buffer entry points -- device memory kept between
// calls, first touch of the result pages while the labels travel, the device list of the one-process multi-GPU route, sdf and
// the voxel-graph transform on host buffers.  Every voxel is still touched by HIP kernels only (run_device, edt_api.hip).
#include <system_error>
#include <thread>

#include <sys/mman.h>

#include "edt_api_internal.h"

namespace edt_amd {

// ---- host-buffer staging -----------------------------------------------------------------
// Device memory of the host-buffer entry points is kept between calls: hipMalloc / hipFree of the
// gigabyte-sized label, output and scratch buffers cost more than the transfers (measured: 64 ms per
// 512^3 uint32 call with fresh allocations, of which 2 x 9.5 ms are PCIe and 0.7 ms kernels).  One
// process-wide pool, one host call at a time (the mutex is held for the whole call); released by
// edt_hip_release_cache() or at exit.  EDT_HIP_NO_CACHE=1 restores allocate-per-call.
struct DevicePool {
  static constexpr int kSlots = 6;
  void *p[kSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[kSlots] = {0, 0, 0, 0, 0, 0};
  std::mutex m;
  void release() {  // (call with the owning device current)
    for (int i = 0; i < kSlots; ++i) {
      if (p[i]) (void)hipFree(p[i]);
      p[i] = nullptr;
      cap[i] = 0;
    }
  }
  ~DevicePool() { /* the runtime may already be gone at static destruction: leak on purpose */ }
};
// one pool per device ordinal: a host-buffer call uses the pool of the device that is current on the
// calling thread, so buffers are never handed to kernels running on another device
constexpr int kMaxDevices = 64;
static DevicePool g_pools[kMaxDevices];

static DevicePool *current_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return (dev >= 0 && dev < kMaxDevices) ? &g_pools[dev] : nullptr;
}

struct DeviceBuf {
  DevicePool *pool;  // nullptr: private allocations only
  void *p = nullptr;
  bool owned = false;
  explicit DeviceBuf(DevicePool *pl) : pool(pl) {}
  ~DeviceBuf() { if (p && owned) (void)hipFree(p); }
  // slot < 0 (or no pool): private allocation, freed with the object; otherwise the pool slot is (re)used.
  // The caller holds pool->m when it uses slots.
  int alloc(size_t bytes, int slot = -1) {
    if (bytes == 0) bytes = 256;
    if (slot >= 0 && pool) {
      if (pool->cap[slot] < bytes) {
        if (pool->p[slot]) (void)hipFree(pool->p[slot]);
        pool->p[slot] = nullptr;
        pool->cap[slot] = 0;
        const hipError_t e = hipMalloc(&pool->p[slot], bytes);
        if (e != hipSuccess) {
          pool->p[slot] = nullptr;
          (void)hipGetLastError();
          pool->release();  // give everything back and let the caller see the failure
          set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
          return EDT_ERR_NOMEM;
        }
        pool->cap[slot] = bytes;
      }
      p = pool->p[slot];
      owned = false;
      return EDT_OK;
    }
    const hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
      return EDT_ERR_NOMEM;
    }
    owned = true;
    return EDT_OK;
  }
};

static bool pool_enabled() {
  const char *e = std::getenv("EDT_HIP_NO_CACHE");
  return !(e && e[0] == '1');
}

// First touch of a large, freshly allocated result array is what dominated the host-buffer path: the kernel
// zero-fills every page on its first write, one core at a time inside the device-to-host copy (measured:
// ~35 of the 57 ms of a 512^3 call, against 2 x 9.5 ms of PCIe and 0.7 ms of kernels).  The pages are
// therefore touched by a few threads WHILE the labels travel to the device and the kernels run; the copy
// back then proceeds at PCIe speed.  (Every byte of the buffer is overwritten by the result afterwards.)
struct Prefault {
  std::vector<std::thread> threads;
  Prefault(void *buf, size_t bytes) {
    constexpr size_t kPage = 4096, kMin = size_t(32) << 20;
    const char *off = std::getenv("EDT_HIP_NO_PREFAULT");
    if (bytes < kMin || (off && off[0] == '1')) return;
    unsigned n = std::thread::hardware_concurrency();
    n = n == 0 ? 4 : (n > 16 ? 16 : n);
    const size_t chunk = align_up((bytes + n - 1) / n, kPage);
    volatile char *base = static_cast<volatile char *>(buf);
#ifdef MADV_HUGEPAGE
    {
      // transparent huge pages for the part of the buffer that can have them (the box runs THP in
      // "madvise" mode): 2 MiB per fault instead of 4 KiB, and a cheaper unmap when the array is freed
      const char *thp = std::getenv("EDT_HIP_NO_THP");
      const uintptr_t lo = align_up(reinterpret_cast<uintptr_t>(buf), kPage);
      const uintptr_t hi = (reinterpret_cast<uintptr_t>(buf) + bytes) & ~(uintptr_t)(kPage - 1);
      if (hi > lo && !(thp && thp[0] == '1')) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
#endif
    for (unsigned t = 0; t < n; ++t) {
      const size_t lo = (size_t)t * chunk, hi = std::min(bytes, lo + chunk);
      if (lo >= hi) break;
      try {
        threads.emplace_back([base, lo, hi] {
          for (size_t o = lo; o < hi; o += kPage) base[o] = 0;
          base[hi - 1] = 0;
        });
      } catch (const std::system_error &) {
        break;  // no more threads to be had: the remaining pages are touched by the copy itself (slower, not wrong)
      }
    }
  }
  void join() {
    for (auto &t : threads) t.join();
    threads.clear();
  }
  ~Prefault() { join(); }
};

// Devices of the one-process multi-GPU route (edt_multi.hip); empty = single device.
static std::mutex g_devices_mutex;
static std::vector<int> g_devices = [] {
  std::vector<int> v;
  if (const char *e = std::getenv("EDT_HIP_DEVICES")) {
    const char *p = e;
    while (*p) {
      char *end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p) break;
      v.push_back((int)d);
      p = (*end == ',') ? end + 1 : end;
    }
  }
  return v;
}();

constexpr int EDT_FLAG_SINGLE_DEVICE = 0x4000;  // internal: do not take the multi-GPU route

// The first device of the list (edt_hip_set_devices / EDT_HIP_DEVICES) for the duration of one host-buffer call that
// is not sharded; no list: the caller's current device stays.
struct ListedDevice {
  int prev = -1;
  bool switched = false;
  ListedDevice() {
    int first = -1;
    {
      std::lock_guard<std::mutex> lock(g_devices_mutex);
      if (!g_devices.empty()) first = g_devices[0];
    }
    if (first >= 0 && hipGetDevice(&prev) == hipSuccess && prev != first && hipSetDevice(first) == hipSuccess) switched = true;
  }
  ~ListedDevice() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static int run_host(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz,
                    float wx, float wy, float wz, int flags, float *output) {
  int rc = check_shape(dtype, ndim, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(ndim, wx, wy, wz)) != EDT_OK) return rc;
  const int64_t voxels = sx * sy * sz;
  if (voxels == 0) return EDT_OK;
  if (!labels || !output) { set_error("null host pointer"); return EDT_ERR_BAD_ARG; }
  rc = require_device();
  if (rc != EDT_OK) return rc;
  if (env_force_generic()) flags |= EDT_FLAG_FORCE_GENERIC;
  // The device list (edt_hip_set_devices / EDT_HIP_DEVICES) is honoured by EVERY host-buffer call: a 3-D volume the
  // slab-record form can cut is Z-sharded over the listed devices, everything else (1-D, 2-D, stacks of images, the binary
  // route, the forced generic kernels, volumes that cannot be cut) runs on the FIRST listed device.
  if (!(flags & EDT_FLAG_SINGLE_DEVICE)) {
    std::vector<int> devs;
    {
      std::lock_guard<std::mutex> lock(g_devices_mutex);
      devs = g_devices;
    }
    const bool shardable = ndim == 3 && !(flags & (EDT_FLAG_FORCE_GENERIC | EDT_FLAG_BATCH_2D | EDT_FLAG_BINARY_YZ));
    if (shardable && devs.size() >= 2 && multi_supported(dtype, sx, sy, sz, (int)devs.size())) {
      Prefault touch(output, (size_t)voxels * sizeof(float));
      touch.join();
      return run_multi(labels, dtype, sx, sy, sz, wx, wy, wz, flags, output, devs.data(), (int)devs.size());
    }
    if (!devs.empty()) {
      // a one-entry list, or a call the slab-record form does not cover: the FIRST listed device does it alone
      if (shardable && devs.size() >= 2) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
          fprintf(stderr, "[edt_hip] note: a %lld x %lld x %lld volume cannot be Z-sharded over %zu devices (slab records: "
                          "sx, sy and sz <= 2048, >= 1 z-slice and >= 32 y-rows per device); device %d runs it alone\n",
                  (long long)sx, (long long)sy, (long long)sz, devs.size(), devs[0]);
      }
      int prev = 0;
      EDT_HIP_TRY(hipGetDevice(&prev));
      if (prev != devs[0]) {
        EDT_HIP_TRY(hipSetDevice(devs[0]));
        rc = run_host(labels, dtype, ndim, sx, sy, sz, wx, wy, wz, flags | EDT_FLAG_SINGLE_DEVICE, output);
        (void)hipSetDevice(prev);
        return rc;
      }
    }
  }

  const size_t lbytes = (size_t)voxels * dtype_size(dtype);
  const size_t obytes = (size_t)voxels * sizeof(float);
  const size_t wbytes = edt_hip_workspace_bytes_flags(dtype, ndim, sx, sy, sz, flags);
  const bool pooled = pool_enabled();
  DevicePool *pool = pooled ? current_pool() : nullptr;
  std::unique_lock<std::mutex> pool_lock;
  if (pool) pool_lock = std::unique_lock<std::mutex>(pool->m);
  DeviceBuf d_labels(pool), d_out(pool), d_ws(pool);
  if ((rc = d_labels.alloc(lbytes, pooled ? 0 : -1)) != EDT_OK) return rc;
  if ((rc = d_out.alloc(obytes, pooled ? 1 : -1)) != EDT_OK) return rc;
  if ((rc = d_ws.alloc(wbytes, pooled ? 2 : -1)) != EDT_OK) return rc;
  Prefault touch(output, obytes);  // the result pages, while the labels travel and the kernels run
  EDT_HIP_TRY(hipMemcpy(d_labels.p, labels, lbytes, hipMemcpyHostToDevice));
  rc = run_device(d_labels.p, dtype, ndim, sx, sy, sz, wx, wy, wz, flags, (float *)d_out.p, d_ws.p,
                  wbytes, nullptr);
  if (rc != EDT_OK) return rc;
  touch.join();
  EDT_HIP_TRY(hipMemcpy(output, d_out.p, obytes, hipMemcpyDeviceToHost));
  return EDT_OK;
}


// sdf / sdfsq on host buffers in ONE round trip (reference: src/edt.pyx:121-202, two transforms and a
// subtraction on the host): labels up once, the SIGNED transform on the device (one transform: EDT_FLAG_SIGNED; shapes it does
// not serve: edt(labels), the background mask, edt(mask) and the subtraction), the difference down once.
static int sdf_host(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx,
                    float wy, float wz, int flags, float *output) {
  int rc = check_shape(dtype, ndim, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(ndim, wx, wy, wz)) != EDT_OK) return rc;
  const int64_t voxels = sx * sy * sz;
  if (voxels == 0) return EDT_OK;
  if (!labels || !output) { set_error("null host pointer"); return EDT_ERR_BAD_ARG; }
  if ((rc = require_device()) != EDT_OK) return rc;
  ListedDevice on_listed_device;
  if (env_force_generic()) flags |= EDT_FLAG_FORCE_GENERIC;
  const size_t lbytes = (size_t)voxels * dtype_size(dtype), obytes = (size_t)voxels * sizeof(float);
  const size_t wbytes = std::max(edt_hip_workspace_bytes_flags(dtype, ndim, sx, sy, sz, flags),
                                 edt_hip_workspace_bytes_flags(EDT_U8, ndim, sx, sy, sz, flags));
  const bool pooled = pool_enabled();
  DevicePool *pool = pooled ? current_pool() : nullptr;
  std::unique_lock<std::mutex> pool_lock;
  if (pool) pool_lock = std::unique_lock<std::mutex>(pool->m);
  DeviceBuf d_labels(pool), d_a(pool), d_ws(pool), d_mask(pool), d_b(pool);
  if ((rc = d_labels.alloc(lbytes, pooled ? 0 : -1)) != EDT_OK) return rc;
  if ((rc = d_a.alloc(obytes, pooled ? 1 : -1)) != EDT_OK) return rc;
  if ((rc = d_ws.alloc(wbytes, pooled ? 2 : -1)) != EDT_OK) return rc;
  if (signed_transform_supported(dtype, ndim, sx, sy, sz, flags)) {
    // ONE transform (EDT_FLAG_SIGNED, edt_api.hip): label 0 measured like every label, its voxels negated at the end
    Prefault touch(output, obytes);
    EDT_HIP_TRY(hipMemcpy(d_labels.p, labels, lbytes, hipMemcpyHostToDevice));
    rc = run_device(d_labels.p, dtype, ndim, sx, sy, sz, wx, wy, wz, flags | EDT_FLAG_SIGNED, (float *)d_a.p, d_ws.p, wbytes, nullptr);
    if (rc != EDT_OK) return rc;
    touch.join();
    EDT_HIP_TRY(hipMemcpy(output, d_a.p, obytes, hipMemcpyDeviceToHost));
    return EDT_OK;
  }
  if ((rc = d_mask.alloc((size_t)voxels, pooled ? 3 : -1)) != EDT_OK) return rc;
  if ((rc = d_b.alloc(obytes, pooled ? 4 : -1)) != EDT_OK) return rc;
  Prefault touch(output, obytes);
  EDT_HIP_TRY(hipMemcpy(d_labels.p, labels, lbytes, hipMemcpyHostToDevice));
  rc = run_device(d_labels.p, dtype, ndim, sx, sy, sz, wx, wy, wz, flags, (float *)d_a.p, d_ws.p, wbytes, nullptr);
  if (rc != EDT_OK) return rc;
  rc = launch_is_background(dtype, d_labels.p, (uint8_t *)d_mask.p, voxels, nullptr);
  if (rc != EDT_OK) return rc;
  rc = run_device(d_mask.p, EDT_U8, ndim, sx, sy, sz, wx, wy, wz, flags, (float *)d_b.p, d_ws.p, wbytes, nullptr);
  if (rc != EDT_OK) return rc;
  rc = launch_subtract((const float *)d_a.p, (const float *)d_b.p, (float *)d_a.p, voxels, nullptr);
  if (rc != EDT_OK) return rc;
  touch.join();
  EDT_HIP_TRY(hipMemcpy(output, d_a.p, obytes, hipMemcpyDeviceToHost));
  return EDT_OK;
}

static int voxel_graph_host(const void *labels, int dtype, const uint8_t *graph, int ndim, int64_t sx,
                            int64_t sy, int64_t sz, float wx, float wy, float wz, int black_border,
                            float *output) {
  int rc = check_shape(dtype, ndim, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(ndim, wx, wy, wz)) != EDT_OK) return rc;
  const int64_t voxels = sx * sy * sz;
  if (voxels == 0) return EDT_OK;
  if (!labels || !graph || !output) { set_error("null host pointer"); return EDT_ERR_BAD_ARG; }
  if ((rc = require_device()) != EDT_OK) return rc;
  ListedDevice on_listed_device;
  const size_t lbytes = (size_t)voxels * dtype_size(dtype);
  const size_t wbytes = edt_hip_voxel_graph_workspace_bytes(ndim, sx, sy, sz);
  const bool pooled = pool_enabled();
  DevicePool *pool = pooled ? current_pool() : nullptr;
  std::unique_lock<std::mutex> pool_lock;
  if (pool) pool_lock = std::unique_lock<std::mutex>(pool->m);
  DeviceBuf d_labels(pool), d_graph(pool), d_ws(pool), d_out(pool);
  if ((rc = d_labels.alloc(lbytes, pooled ? 0 : -1)) != EDT_OK) return rc;
  if ((rc = d_out.alloc((size_t)voxels * sizeof(float), pooled ? 1 : -1)) != EDT_OK) return rc;
  if ((rc = d_ws.alloc(wbytes, pooled ? 2 : -1)) != EDT_OK) return rc;
  if ((rc = d_graph.alloc((size_t)voxels, pooled ? 3 : -1)) != EDT_OK) return rc;
  Prefault touch(output, (size_t)voxels * sizeof(float));
  EDT_HIP_TRY(hipMemcpy(d_labels.p, labels, lbytes, hipMemcpyHostToDevice));
  EDT_HIP_TRY(hipMemcpy(d_graph.p, graph, (size_t)voxels, hipMemcpyHostToDevice));
  rc = edt_hip_edtsq_voxel_graph_device(d_labels.p, dtype, (const uint8_t *)d_graph.p, ndim, sx, sy, sz, wx, wy, wz,
                                        black_border ? EDT_FLAG_BLACK_BORDER : 0, (float *)d_out.p, d_ws.p, wbytes,
                                        nullptr);
  if (rc != EDT_OK) return rc;
  touch.join();
  EDT_HIP_TRY(hipMemcpy(output, d_out.p, (size_t)voxels * sizeof(float), hipMemcpyDeviceToHost));
  return EDT_OK;
}

}  // namespace edt_amd

using namespace edt_amd;

extern "C" {

int edt_hip_squared_edt_1d_multi_seg(const void *labels, int dtype, float *dest, int64_t n,
                                     int64_t stride, float anisotropy, int black_border) {
  if (stride != 1) {
    set_error("stride != 1 is not supported (no reference caller uses it)");
    return EDT_ERR_UNSUPPORTED;
  }
  return run_host(labels, dtype, 1, n, 1, 1, anisotropy, 1.0f, 1.0f,
                  black_border ? EDT_FLAG_BLACK_BORDER : 0, dest);
}

int edt_hip_edt2dsq(const void *labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                    int black_border, int /*parallel*/, float *output) {
  return run_host(labels, dtype, 2, sx, sy, 1, wx, wy, 1.0f,
                  black_border ? EDT_FLAG_BLACK_BORDER : 0, output);
}

int edt_hip_edt3dsq(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx,
                    float wy, float wz, int black_border, int /*parallel*/, float *output) {
  return run_host(labels, dtype, 3, sx, sy, sz, wx, wy, wz,
                  black_border ? EDT_FLAG_BLACK_BORDER : 0, output);
}

int edt_hip_edt2d(const void *labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                  int black_border, int /*parallel*/, float *output) {
  return run_host(labels, dtype, 2, sx, sy, 1, wx, wy, 1.0f,
                  (black_border ? EDT_FLAG_BLACK_BORDER : 0) | EDT_FLAG_SQRT, output);
}

int edt_hip_edt3d(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx,
                  float wy, float wz, int black_border, int /*parallel*/, float *output) {
  return run_host(labels, dtype, 3, sx, sy, sz, wx, wy, wz,
                  (black_border ? EDT_FLAG_BLACK_BORDER : 0) | EDT_FLAG_SQRT, output);
}

int edt_hip_binary_edtsq(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx,
                         float wy, float wz, int black_border, int take_sqrt, float *output) {
  if (ndim != 2 && ndim != 3) { set_error("binary route: ndim must be 2 or 3"); return EDT_ERR_BAD_ARG; }
  return run_host(labels, dtype, ndim, sx, sy, sz, wx, wy, wz,
                  (black_border ? EDT_FLAG_BLACK_BORDER : 0) | (take_sqrt ? EDT_FLAG_SQRT : 0) | EDT_FLAG_BINARY_YZ,
                  output);
}

int edt_hip_edt2dsq_batch(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t count, float wx, float wy,
                          int black_border, int take_sqrt, float *output) {
  return run_host(labels, dtype, 3, sx, sy, count, wx, wy, 1.0f,
                  (black_border ? EDT_FLAG_BLACK_BORDER : 0) | (take_sqrt ? EDT_FLAG_SQRT : 0) | EDT_FLAG_BATCH_2D,
                  output);
}

int edt_hip_set_devices(const int *devices, int n_devices) {
  if (n_devices < 0 || (n_devices > 0 && !devices)) { set_error("bad device list"); return EDT_ERR_BAD_ARG; }
  const int have = edt_hip_device_count();
  for (int i = 0; i < n_devices; ++i)
    if (devices[i] < 0 || devices[i] >= have) { set_error("device ordinal out of range"); return EDT_ERR_BAD_ARG; }
  std::lock_guard<std::mutex> lock(g_devices_mutex);
  g_devices.assign(devices, devices + n_devices);
  return EDT_OK;
}

int edt_hip_edt3dsq_multi(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                          float wz, int black_border, int take_sqrt, float *output, const int *devices,
                          int n_devices) {
  int rc = check_shape(dtype, 3, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(3, wx, wy, wz)) != EDT_OK) return rc;
  if (sx == 0 || sy == 0 || sz == 0) return EDT_OK;
  if (!labels || !output || !devices || n_devices < 1) { set_error("null pointer / empty device list"); return EDT_ERR_BAD_ARG; }
  if ((rc = require_device()) != EDT_OK) return rc;
  const int have = edt_hip_device_count();
  for (int i = 0; i < n_devices; ++i)
    if (devices[i] < 0 || devices[i] >= have) { set_error("device ordinal out of range"); return EDT_ERR_BAD_ARG; }
  const int flags = (black_border ? EDT_FLAG_BLACK_BORDER : 0) | (take_sqrt ? EDT_FLAG_SQRT : 0);
  if (n_devices >= 2 && !multi_supported(dtype, sx, sy, sz, n_devices)) {
    set_error("this volume cannot be Z-sharded over " + std::to_string(n_devices) + " devices (slab records: sx, sy "
              "and sz <= 2048, at least one z-slice and 32 y-rows per device; edt_hip_multi_supported tells)");
    return EDT_ERR_UNSUPPORTED;
  }
  if (n_devices == 1) {  // a list of one: that device does it
    int prev = 0;
    EDT_HIP_TRY(hipGetDevice(&prev));
    EDT_HIP_TRY(hipSetDevice(devices[0]));
    rc = run_host(labels, dtype, 3, sx, sy, sz, wx, wy, wz, flags | EDT_FLAG_SINGLE_DEVICE, output);
    (void)hipSetDevice(prev);
    return rc;
  }
  Prefault touch(output, (size_t)(sx * sy * sz) * sizeof(float));
  touch.join();
  return run_multi(labels, dtype, sx, sy, sz, wx, wy, wz, flags, output, devices, n_devices);
}

int edt_hip_multi_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, int n_devices) {
  if (check_shape(dtype, 3, sx, sy, sz) != EDT_OK) return 0;
  return (n_devices == 1 || multi_supported(dtype, sx, sy, sz, n_devices)) ? 1 : 0;
}

int edt_hip_sdf(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                float wz, int black_border, int squared, float *output) {
  return sdf_host(labels, dtype, ndim, sx, sy, sz, wx, wy, wz,
                  (black_border ? EDT_FLAG_BLACK_BORDER : 0) | (squared ? 0 : EDT_FLAG_SQRT), output);
}

int edt_hip_release_cache(void) {
  multi_release();
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return EDT_OK; }
  for (int d = 0; d < kMaxDevices; ++d) {
    std::lock_guard<std::mutex> lock(g_pools[d].m);
    bool any = false;
    for (int i = 0; i < DevicePool::kSlots; ++i) any = any || g_pools[d].p[i] != nullptr;
    if (!any) continue;
    if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
    g_pools[d].release();
  }
  (void)hipSetDevice(cur);
  return EDT_OK;
}

int edt_hip_edt2dsq_voxel_graph(const void *labels, int dtype, const uint8_t *graph, int64_t sx,
                                int64_t sy, float wx, float wy, int black_border,
                                float *workspace) {
  return voxel_graph_host(labels, dtype, graph, 2, sx, sy, 1, wx, wy, 2.0f, black_border, workspace);
}

int edt_hip_edt3dsq_voxel_graph(const void *labels, int dtype, const uint8_t *graph, int64_t sx,
                                int64_t sy, int64_t sz, float wx, float wy, float wz,
                                int black_border, float *workspace) {
  return voxel_graph_host(labels, dtype, graph, 3, sx, sy, sz, wx, wy, wz, black_border, workspace);
}

}  // extern "C"
