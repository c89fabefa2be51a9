// edt_api_internal.h -- what the three translation units of the C ABI share (edt_api.hip: plan + dispatch on device-resident
// data; edt_host.hip: host-buffer staging; edt_shard_api.hip: the Z-sharded phases).  Internal: nothing here is exported.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

// ---- workspace carving -----------------------------------------------------------------
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *p) : base((char *)p) {}
  template <typename T>
  T *take(size_t count) {
    off = align_up(off, 256);
    T *p = base ? (T *)(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

AxisGeom make_geom_y(int64_t sx, int64_t sy, int64_t sz);
AxisGeom make_geom_z(int64_t sx, int64_t sy, int64_t sz);
// In-place LDS-tiled column pass where one applies (edt_api.hip)
bool column_inplace_supported(const AxisGeom &g);
int launch_column_inplace(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb, int epi,
                          hipStream_t stream, const TileList &list = TileList());
int launch_row_bits(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sx,
                    int64_t sy, int64_t sz, float w, int bb, int to_finite, hipStream_t stream);
bool env_force_generic();  // EDT_HIP_FORCE_GENERIC=1: every call takes the fallback kernels (test hook)
int check_shape(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz);
int check_voxel_sizes(int naxes, float &wx, float &wy, float &wz);  // (drops the sign of wy / wz: they enter as squares)
int check_column_voxel_size(float &w);
bool signed_transform_supported(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags);  // EDT_FLAG_SIGNED
int require_device();
// the pass pipeline on device-resident data
int run_device(const void *d_labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
               int flags, float *d_out, void *d_ws, size_t ws_bytes, hipStream_t stream);
const char *last_error_cstr();

// ---- per-pass event timing (bench.py reads this through edt_hip_get_pass_times) ----------------
struct PassLog {
  std::atomic<bool> enabled{false};
  std::vector<hipEvent_t> pool;          // reused events
  std::vector<std::pair<int, int>> span;  // (start, stop) indices of the last call
  std::vector<std::string> names;
  int used = 0;
};
extern PassLog g_log;
extern std::mutex g_log_mutex;
hipEvent_t log_event();
void log_begin_call();

struct ScopedPass {
  hipStream_t stream;
  int start = -1;
  bool named;
  ScopedPass(const char *name, hipStream_t s) : stream(s), named(name != nullptr) {  // (no name: not a pass of its own)
    if (!named) return;
    if (debug_mode() & 0x1000) fprintf(stderr, "[edt_hip] pass start: %s\n", name);
    if (!g_log.enabled.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_log_mutex);  // (only while profiling is switched on)
    hipEvent_t e = log_event();
    if (!e) return;
    start = g_log.used - 1;
    (void)hipEventRecord(e, stream);
    g_log.names.push_back(name);
  }
  ~ScopedPass() {
    if (!named) return;
    if (debug_mode() & 0x1000) {  // diagnostics: name every pass as it completes
      const hipError_t e = hipStreamSynchronize(stream);
      fprintf(stderr, "[edt_hip] pass done: %s\n", hipGetErrorString(e));
    }
    if (start < 0) return;
    std::lock_guard<std::mutex> lock(g_log_mutex);
    hipEvent_t e = log_event();
    if (!e) return;
    (void)hipEventRecord(e, stream);
    g_log.span.push_back({start, g_log.used - 1});
  }
};


}  // namespace edt_amd
