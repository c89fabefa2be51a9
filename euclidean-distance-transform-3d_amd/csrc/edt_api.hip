// edt_api.hip -- the C ABI (include/edt_hip.h): orchestration of the passes, workspace
// carving, host-buffer staging, per-pass event timing.  No compute happens on the host and
// there is no CPU fallback: every entry point needs a HIP device.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

static thread_local std::string g_last_error = "";
// diagnostics mode: thread-local, preset per thread from EDT_HIP_DEBUG_MODE; see edt_common.h for the bits
#ifdef EDT_DIAG
constexpr int kDiagMask = ~0;
#else
constexpr int kDiagMask = kDiagFormBits;
#endif
static int env_debug_mode() {
  static const int v = [] {
    const char *e = std::getenv("EDT_HIP_DEBUG_MODE");
    return e ? (int)std::strtol(e, nullptr, 0) : 0;
  }();
  return v;
}
static thread_local int g_debug_mode = env_debug_mode() & kDiagMask;

void set_error(const std::string &msg) { g_last_error = msg; }
int debug_mode() { return g_debug_mode; }
void set_thread_debug_mode(int mode) { g_debug_mode = mode & kDiagMask; }



// ---- per-pass event timing (bench.py reads this) -------------------------------------
struct PassLog {
  std::atomic<bool> enabled{false};
  std::vector<hipEvent_t> pool;          // reused events
  std::vector<std::pair<int, int>> span;  // (start, stop) indices of the last call
  std::vector<std::string> names;
  int used = 0;
};
static PassLog g_log;
static std::mutex g_log_mutex;

static hipEvent_t log_event() {
  if (g_log.used == (int)g_log.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_log.pool.push_back(e);
  }
  return g_log.pool[g_log.used++];
}

struct ScopedPass {
  hipStream_t stream;
  int start = -1;
  bool named;
  ScopedPass(const char *name, hipStream_t s) : stream(s), named(name != nullptr) {  // (no name: not a pass of its own)
    if (!named) return;
    if (g_debug_mode & 0x1000) fprintf(stderr, "[edt_hip] pass start: %s\n", name);
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
    if (g_debug_mode & 0x1000) {  // diagnostics: name every pass as it completes
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

static void log_begin_call() {
  g_log.used = 0;
  g_log.span.clear();
  g_log.names.clear();
}

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

struct Plan {
  int ndim;
  int64_t sx, sy, sz, voxels;
  AxisGeom gy, gz;
  // carved pointers
  float *bufB = nullptr;
  int32_t *stack = nullptr;
  uint32_t *nz_y = nullptr, *rs_y = nullptr, *zs_y = nullptr, *nz_z = nullptr, *rs_z = nullptr;
  unsigned char *line_ws = nullptr;  // scratch of the line pipeline when pass 1 runs over rows no row kernel takes (> 4096 voxels)
  uint16_t *codes = nullptr;  // 16-bit distance indices of pass 1 (index form), one slab of xy_slab slices
  int64_t xy_slab = 0;        // slices per slab of the slab-wise X/Y passes (0: no index form for this shape)
  // the tiles the 16-bit integer column kernel hands to the fp32 kernel (edt_colq16.hip): kQ16Slots counters, one per
  // column-pass launch of a call, and one array of tile ids (launches are stream-ordered: the array is reused)
  uint32_t *q16_counts = nullptr, *q16_ids = nullptr;
  // which tiles of pass Y left their results in the 16-bit plane (= codes): one bit per (x-tile, z), behind the counters
  uint32_t *q16_map = nullptr;
  int q16_map_words = 0;  // words per x-tile
  int64_t q16_id_capacity = 0;  // tile ids q16_ids holds
  size_t bytes = 0;
};

static AxisGeom make_geom_y(int64_t sx, int64_t sy, int64_t sz) {
  AxisGeom g;
  g.sx = sx; g.n = sy; g.stride = sx; g.nouter = sz; g.outer_stride = sx * sy;
  g.nbands = ceil_div(sy, kBandRows);
  return g;
}
static AxisGeom make_geom_z(int64_t sx, int64_t sy, int64_t sz) {
  AxisGeom g;
  g.sx = sx; g.n = sz; g.stride = sx * sy; g.nouter = sy; g.outer_stride = sx;
  g.nbands = ceil_div(sz, kBandRows);
  return g;
}

// What a call needs besides the bit planes: the second fp32 volume + hull stacks of the size-agnostic
// column pass (only when an axis is too long for the in-place LDS kernels, or the caller forces the
// generic kernels), and one slab of 16-bit distance indices for the index form of pass 1.
static bool plan_needs_pingpong(int ndim, int64_t sx, int64_t sy, int64_t sz, int flags);

// Index form of pass 1 (edt_rowwave.hip C16 -> edt_colwave_kernel.h XF): pass 1 hands the first column pass 2 bytes
// per voxel instead of 4.  The indices live in the workspace, never more than kCodeSlabVoxels of them: passes X and
// Y touch one z-slice at a time, so larger volumes run them slab by slab (a slab of 2^27 voxels is a whole 512^3
// volume's worth of parallelism) -- 256 MiB of scratch whatever the volume.  Shapes: the register-resident pass 1
// and the wave column kernel, rows of whole 8-byte granules.  (debug bit 0x100000 switches the form off.)
constexpr int64_t kCodeSlabVoxels = (int64_t)1 << 27;
constexpr int kQ16Slots = 256;  // counters of the 16-bit integer column kernel's hand-over lists (one per launch)
constexpr int EDT_FLAG_NO_INDEX_FORM = 0x8000;  // internal: plan without the index buffer
static bool env_force_generic();
static int64_t plan_code_slab(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (ndim < 2 || sx % 4 != 0 || sx * sy > kCodeSlabVoxels || (flags & (EDT_FLAG_NO_INDEX_FORM | EDT_FLAG_SMALL_WORKSPACE))) return 0;
  if ((flags & EDT_FLAG_FORCE_GENERIC) || env_force_generic() || (g_debug_mode & (0x100000 | 64 | 32))) return 0;
  if (!row_pass_wave_supported(dtype, sx, sy, sz) || !column_pass_wave_supported(make_geom_y(sx, sy, sz))) return 0;
  const int64_t slab = kCodeSlabVoxels / (sx * sy);
  return slab < sz ? slab : sz;
}

static Plan make_plan(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, void *ws, int flags) {
  Plan p;
  p.ndim = ndim; p.sx = sx; p.sy = sy; p.sz = sz; p.voxels = sx * sy * sz;
  p.gy = make_geom_y(sx, sy, sz);
  p.gz = make_geom_z(sx, sy, sz);
  Carver c(ws);
  if (ndim >= 2) {
    if (plan_needs_pingpong(ndim, sx, sy, sz, flags)) {
      p.bufB = c.take<float>((size_t)p.voxels);
      p.stack = c.take<int32_t>((size_t)p.voxels);
    }
    const size_t wy = (size_t)(p.gy.sx * p.gy.nbands * p.gy.nouter);
    p.nz_y = c.take<uint32_t>(wy);
    p.rs_y = c.take<uint32_t>(wy);
  }
  if (ndim >= 3 && !(flags & EDT_FLAG_BATCH_2D)) {
    // the z-packed planes are built AFTER the y pass: its run-start plane is dead by then and lends its
    // storage to the z run starts (four planes in all: 1/8 byte per voxel each, 0.5 GiB for 1024^3)
    const size_t wy = (size_t)(p.gy.sx * p.gy.nbands * p.gy.nouter);
    const size_t wz = (size_t)(p.gz.sx * p.gz.nbands * p.gz.nouter);
    p.nz_z = c.take<uint32_t>(wz);
    if (wz <= wy) p.rs_z = p.rs_y;
    else p.rs_z = c.take<uint32_t>(wz);
    p.zs_y = c.take<uint32_t>(wy);
  }
  if (ndim >= 2) {
    p.xy_slab = plan_code_slab(dtype, ndim, sx, sy, sz, flags);
    if (p.xy_slab > 0) p.codes = c.take<uint16_t>((size_t)(p.xy_slab * sx * sy));
  }
  if (ndim >= 2 && !(flags & EDT_FLAG_FORCE_GENERIC) && !env_force_generic()) {
    // (ids in the fp32 kernel's geometry: 16-column tiles for axes of more than 512 rows)
    // (pass Y runs over all sz slices at once whenever the index form is not taken at run time: sized for that)
    const int64_t ty = ceil_div(sx, 16) * (ceil_div(sz, 8) * 8);
    const int64_t tz = ceil_div(sx, 16) * (ceil_div(sy, 8) * 8);
    p.q16_map_words = (int)ceil_div(sz, 32);
    p.q16_counts = c.take<uint32_t>(kQ16Slots + (size_t)(ceil_div(sx, 32) * p.q16_map_words));  // (zeroed together)
    p.q16_map = p.q16_counts ? p.q16_counts + kQ16Slots : nullptr;
    p.q16_id_capacity = std::max(ty, tz);
    p.q16_ids = c.take<uint32_t>((size_t)p.q16_id_capacity);
  }
  if (ndim == 1) (void)c.take<unsigned char>(line_workspace_bytes(sx));  // block scan + table of the 1-D pipeline
  // rows too long for the row kernels (more than 4096 voxels; more than 2048 where the wave kernel does not apply): pass 1
  // runs as the line pipeline over the stack of rows and needs its scratch
  if (ndim >= 2 && !row_pass_tiled_supported(sx) && !row_pass_wave_supported(dtype, sx, sy, sz) &&
      !(flags & EDT_FLAG_FORCE_GENERIC) && !env_force_generic())
    p.line_ws = c.take<unsigned char>(rows_line_workspace_bytes(sx, sy * sz));
  p.bytes = align_up(c.off, 256) + 256;
  return p;
}

// In-place LDS-tiled column pass: the wave-autonomous kernel where the axis fits its register
// budget, the workgroup-phased kernel for longer axes.  (debug bit 64 forces the latter.)
static bool column_inplace_supported(const AxisGeom &g) {
  return column_pass_wave_supported(g) || column_pass_tiled_supported(g);
}
static int launch_column_inplace(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                                 float w, int bb, int epi, hipStream_t stream, const TileList &list = TileList()) {
  // (a list -- the tiles the 16-bit integer kernel refused -- only exists for axes of the wave kernel)
  if (list.count != nullptr) return launch_column_pass_wave(F, nz, rs, g, w, bb, epi, stream, nullptr, ColumnOut(), list);
  // axes of at most 32 rows with many columns: a thread per column (edt_short.hip); the LDS-tiled kernels would
  // launch a single-wave workgroup per 32 columns.  (debug bit 0x1000000 keeps them on the wave kernel.)
  if (column_pass_short_supported(g) && g.sx * g.nouter >= 4096 && !(g_debug_mode & (64 | 0x1000000)))
    return launch_column_pass_short(F, nz, rs, g, w, bb, epi, stream);
  if (column_pass_wave_supported(g) && !(g_debug_mode & 64))
    return launch_column_pass_wave(F, nz, rs, g, w, bb, epi, stream);
  return launch_column_pass_tiled(F, nz, rs, g, w, bb, epi, stream);
}

// Pass 1 with the bit planes of the column passes as a by-product: the register-resident wave kernel for rows of up
// to 4096 voxels (one, two or four waves per row), the LDS-staged workgroup kernel (rows of up to 2048 voxels) where
// that one does not apply.  (debug bit 32 forces the latter.)
static int launch_row_bits(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                           uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                           int to_finite, hipStream_t stream) {
  if (row_pass_wave_supported(dtype, sx, sy, sz) && !(g_debug_mode & 32))
    return launch_row_pass_wave(dtype, labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream);
  return launch_row_pass_tiled(dtype, labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream);
}

static bool env_force_generic() {
  const char *e = std::getenv("EDT_HIP_FORCE_GENERIC");  // test hook: every call takes the fallback kernels
  return e && e[0] == '1';
}

static bool force_generic_1d(int flags) { return (flags & EDT_FLAG_FORCE_GENERIC) || env_force_generic(); }

static bool plan_needs_pingpong(int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (ndim < 2) return false;
  if ((flags & EDT_FLAG_FORCE_GENERIC) || env_force_generic()) return true;
  if (!column_inplace_supported(make_geom_y(sx, sy, sz))) return true;
  return ndim == 3 && !(flags & EDT_FLAG_BATCH_2D) && !column_inplace_supported(make_geom_z(sx, sy, sz));
}

static int check_shape(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz) {
  if (dtype_size(dtype) == 0) { set_error("unknown dtype code"); return EDT_ERR_BAD_ARG; }
  if (ndim < 1 || ndim > 3) { set_error("ndim must be 1, 2 or 3"); return EDT_ERR_BAD_ARG; }
  if (sx < 0 || sy < 0 || sz < 0) { set_error("negative extent"); return EDT_ERR_BAD_ARG; }
  if ((ndim < 3 && sz != 1) || (ndim < 2 && sy != 1)) {
    set_error("unused extents must be 1");
    return EDT_ERR_BAD_ARG;
  }
  if (sx > INT32_MAX || sy > INT32_MAX || sz > INT32_MAX) {
    set_error("extent exceeds 2^31-1");
    return EDT_ERR_UNSUPPORTED;
  }
  return EDT_OK;
}

// Voxel sizes must be positive and finite.  (The reference does not validate them: a negative size makes its pass 1 cross
// label boundaries -- the unguarded backward fminf sweep, src/edt.hpp:107-109 -- and NaN / inf / 0 give NaN or all-zero
// fields; no kernel here reproduces that, so the call is refused instead of answered differently by different kernels.)
static int check_voxel_sizes(int naxes, float wx, float wy, float wz) {
  const float w[3] = {wx, wy, wz};
  for (int i = 0; i < naxes && i < 3; ++i) {
    if (!(w[i] > 0.0f) || !(w[i] <= FLT_MAX)) {
      set_error("voxel sizes (anisotropy) must be positive and finite");
      return EDT_ERR_BAD_ARG;
    }
  }
  return EDT_OK;
}

static int require_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (this library has no CPU fallback)");
    return EDT_ERR_NO_DEVICE;
  }
  return EDT_OK;
}

// ---- the pass pipeline on device-resident data -----------------------------------------
static int run_device(const void *d_labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz,
                      float wx, float wy, float wz, int flags, float *d_out, void *d_ws,
                      size_t ws_bytes, hipStream_t stream) {
  int rc = check_shape(dtype, ndim, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(ndim, wx, wy, wz)) != EDT_OK) return rc;
  if (sx == 0 || sy == 0 || sz == 0) return EDT_OK;
  if (!d_labels || !d_out) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  Plan p = make_plan(dtype, ndim, sx, sy, sz, d_ws, flags);
  // (a workspace sized while the index form of pass 1 was switched off still serves: fp32 form)
  if (d_ws && ws_bytes < p.bytes && p.codes != nullptr) p = make_plan(dtype, ndim, sx, sy, sz, d_ws, flags | EDT_FLAG_NO_INDEX_FORM);
  // (1-D: the parallel line pipeline needs its block-scan scratch; a call without any -- the ABI of round 1 -- is
  // served by the thread-per-row kernel instead of being refused)
  const bool line_without_ws = ndim == 1 && (!d_ws || ws_bytes < p.bytes);
  if (!line_without_ws && (!d_ws || ws_bytes < p.bytes)) {
    set_error("workspace too small: need " + std::to_string(p.bytes) +
              " bytes (edt_hip_workspace_bytes_flags with the flags of this call)");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const int want_sqrt = (flags & EDT_FLAG_SQRT) ? 1 : 0;
  const int last_epi = (bb ? 0 : kEpiToInf) | (want_sqrt ? kEpiSqrt : 0);
  // a stack of 2-D images is a volume without a z pass
  if ((flags & EDT_FLAG_BATCH_2D) && ndim != 3) { set_error("EDT_FLAG_BATCH_2D needs ndim = 3 (sz = image count)"); return EDT_ERR_BAD_ARG; }
  const bool zpass = ndim == 3 && !(flags & EDT_FLAG_BATCH_2D);
  // the reference's binary route for multi-valued labels: labels split runs in pass 1 only (edt_generic.hip:
  // k_planes_one_run).  Boolean input gives the same values either way and keeps the ordinary planes.
  const bool binary_yz = (flags & EDT_FLAG_BINARY_YZ) != 0 && dtype != EDT_BOOL;
  // lower bounds of the non-zero values passes Y and Z read (AxisGeom::fmin): pass X leaves fl32(T[k]^2) >= fl32(wx^2),
  // pass Y's results are at least the smaller of that and fl32(wy^2) (every candidate is its row's own value or carries
  // a c_d >= w2y; the border terms are >= w2y).  Not for the fused sqrt of a 2-D call (pass Z does not exist then).
  const float fmin_y = edt_hip_field_floor(wx, wx);
  const float fmin_z = edt_hip_field_floor(wx, wy);
  p.gy.fmin = fmin_y;
  p.gz.fmin = fmin_z;

  // the pass log is process-wide and only touched (under its mutex) while profiling is switched on
  if (g_log.enabled.load(std::memory_order_relaxed)) {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    log_begin_call();
  }

  // The 16-bit integer form of the column passes (edt_colq16.hip): voxel sizes that share a quantum.  Every column pass
  // is then two launches: the integer kernel over all tiles, and the fp32 kernel over the list of tiles it refused.
  float q16_q = 1.0f;
  uint32_t q16_a[3] = {1u, 1u, 1u};
  bool q16 = false;
  int q16_slot = 0;
  bool q16_counts_zeroed = false;
  if (ndim >= 2 && p.q16_counts != nullptr) {
    const float ws3[3] = {wx, wy, wz};
    q16 = q16_quantum(ws3, (ndim == 3 && !(flags & EDT_FLAG_BATCH_2D)) ? 3 : 2, &q16_q, q16_a);
  }
  constexpr int kQ16Off = 16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000;
  // When can the integer kernel refuse no tile at all?  With a black border every row of pass X has a boundary and its
  // index is at most ceil(sx / 2); in the index form the values are integers by construction (k^2 * ax quanta), and so
  // are the integer kernel's own results; so if ceil(sx / 2)^2 * ax is within the range of a pass (its 16-bit form, or the
  // wide form: q16_value_limit) and everything that pass reads was written by pass X or by an integer pass that could
  // not refuse either, the fp32 launch over the hand-over list has nothing to do and is not made -- nor is the list's
  // counter zeroed.  (debug bit 0x20000000: always launch.)
  const uint64_t q16_vmax = (uint64_t)ceil_div(sx, 2) * (uint64_t)ceil_div(sx, 2) * q16_a[0];
  auto q16_cannot_refuse = [&](int axis) {
    return q16 && bb && !(g_debug_mode & 0x20000000) && q16_vmax <= q16_value_limit(q16_q, q16_a[axis]);
  };
  // F in place (codes == nullptr) or from the 16-bit indices of pass X; returns the list for the fp32 launch that follows
  // (launched: the integer kernel ran; sure: the caller vouches for what the pass reads -- see above)
  auto q16_pass = [&](float *F, const uint16_t *codes, const uint32_t *rs, const AxisGeom &g, int axis, int epi,
                      TileList &list, uint16_t *plane, bool sure, bool &launched) -> int {
    list = TileList();
    launched = false;
    // (the bits that force one form of the fp32 kernel on every tile -- the test tiers' way to cover them -- keep the call there)
    if (!q16 || q16_slot >= kQ16Slots || !column_pass_q16_supported(g) || !column_pass_wave_supported(g) ||
        !column_pass_q16_aligned(F, codes, plane) || (g_debug_mode & kQ16Off))
      return EDT_OK;
    // (the id array was sized for these tile counts: make_plan)
    if (ceil_div(g.sx, 16) * (ceil_div(g.nouter, 8) * 8) > p.q16_id_capacity) return EDT_OK;
    sure = sure && q16_cannot_refuse(axis);
    if (!sure && !q16_counts_zeroed) {
      EDT_HIP_TRY(hipMemsetAsync(p.q16_counts, 0, kQ16Slots * sizeof(uint32_t), stream));
      q16_counts_zeroed = true;
    }
    uint32_t *count = p.q16_counts + q16_slot++;
    const int r = launch_column_pass_q16(F, codes, rs, g, q16_q, q16_a[axis], q16_a[0], bb, epi, count, p.q16_ids, stream,
                                         nullptr, plane, p.q16_map, p.q16_map_words);
    if (r != EDT_OK) return r;
    launched = true;
    list.count = count;
    list.ids = p.q16_ids;
    list.none = sure;
    return EDT_OK;
  };

  if (ndim == 1) {
    ScopedPass t("x_pass", stream);
    if (force_generic_1d(flags) || line_without_ws) return launch_row_pass_serial(dtype, d_labels, d_out, sx, 1, wx, bb, 0, want_sqrt, stream);
    return launch_line_pass(dtype, d_labels, d_out, sx, wx, bb, want_sqrt, d_ws, stream);
  }

  // Column passes are in place when the LDS-tiled kernel applies, otherwise they ping-pong
  // between two volumes.  Start in the buffer that makes the last pass land in d_out.
  const bool force_generic = (flags & EDT_FLAG_FORCE_GENERIC) != 0 || env_force_generic();
  const bool tiled_y = !force_generic && column_inplace_supported(p.gy);
  const bool tiled_z = !force_generic && column_inplace_supported(p.gz);
  const int swaps = (tiled_y ? 0 : 1) + ((zpass && !tiled_z) ? 1 : 0);
  float *cur = (swaps % 2 == 0) ? d_out : p.bufB;
  float *other = (cur == d_out) ? p.bufB : d_out;
  // (pass 1 on the row kernels: the wave kernel up to 4096 voxels per row, the workgroup-phased one up to 2048)
  const bool tiled_x = !force_generic && (row_pass_tiled_supported(sx) || row_pass_wave_supported(dtype, sx, sy, sz));
  // Index form of pass 1 (see plan_code_slab): pass 1 stores 16-bit distance indices, the first column pass turns
  // them into F while it fills its tile -- 2 B less written and 2 B less read per voxel.  Bit-identical only where
  // every multiple k * wx of the row is exact in fp32 (row_codes_exact); other voxel sizes keep the fp32 form.
  const bool index_form = p.codes != nullptr && tiled_x && tiled_y && row_codes_exact(wx, sx);
  // The 16-bit plane between passes Y and Z: where both run on the integer kernel and the indices of pass X cover the whole
  // volume (one slab), the tiles of pass Y that qualify write their results over their indices -- 2 bytes per voxel out
  // of pass Y and into pass Z instead of 4 -- and pass Z reads every row from wherever pass Y left it.  (debug bit
  // 0x10000000: fp32 between the passes.)
  bool plane16 = q16 && index_form && zpass && p.xy_slab >= sz && !(g_debug_mode & (kQ16Off | 0x10000000)) &&
                       column_pass_q16_supported(p.gy) && column_pass_q16_supported(p.gz) &&
                       column_pass_wave_supported(p.gy) && column_pass_wave_supported(p.gz);
  bool y_sure = true;  // every tile of pass Y was served by the integer kernel, provably (q16_cannot_refuse)
  if (index_form) {
    const int64_t sxy = sx * sy, wpl = p.gy.sx * p.gy.nbands;  // voxels / bit words per slice
    const size_t lsz = dtype_size(dtype);
    const bool one = p.xy_slab >= sz;
    ScopedPass whole(one ? nullptr : "xy_pass", stream);
    for (int64_t z0 = 0; z0 < sz; z0 += p.xy_slab) {
      const int64_t zc = std::min<int64_t>(p.xy_slab, sz - z0);
      const char *lab = static_cast<const char *>(d_labels) + (size_t)(z0 * sxy) * lsz;
      {
        // (slice 0 of a later slab compares against the slice below it through the halo pointer of the sharded path)
        ScopedPass t(one ? "x_pass" : nullptr, stream);
        rc = launch_row_pass_wave(dtype, lab, nullptr, p.nz_y + z0 * wpl, p.rs_y + z0 * wpl,
                                  zpass ? p.zs_y + z0 * wpl : nullptr, sx, sy, zc, wx, bb, bb ? 0 : 1, stream,
                                  z0 > 0 ? lab - (size_t)sxy * lsz : nullptr, p.codes);
        if (rc != EDT_OK) return rc;
        if (binary_yz) {
          AxisGeom gb = p.gy;
          gb.nouter = zc;
          rc = launch_planes_one_run(p.nz_y + z0 * wpl, p.rs_y + z0 * wpl, zpass ? p.zs_y + z0 * wpl : nullptr, gb, z0, stream);
          if (rc != EDT_OK) return rc;
        }
      }
      {
        ScopedPass t(one ? "y_pass" : nullptr, stream);
        AxisGeom g = p.gy;
        g.nouter = zc;
        TileList list;
        bool launched = false;
        rc = q16_pass(cur + z0 * sxy, p.codes, p.rs_y + z0 * wpl, g, 1, zpass ? 0 : last_epi, list, plane16 ? p.codes : nullptr,
                      true, launched);
        if (rc != EDT_OK) return rc;
        if (!launched) plane16 = false;  // (nothing wrote the plane or said where the rows are: pass Z reads fp32 values)
        y_sure = y_sure && list.none;
        if (!list.none)
          rc = launch_column_pass_wave_codes(cur + z0 * sxy, p.codes, p.nz_y + z0 * wpl, p.rs_y + z0 * wpl, g, wy, bb,
                                             zpass ? 0 : last_epi, wx, bb ? 0 : 1, stream, nullptr, list);
        if (rc != EDT_OK) return rc;
      }
    }
  } else if (tiled_x) {
    // labels are read once: pass 1 also emits the run bit-planes of the y and z axes
    {
      ScopedPass t("x_pass", stream);
      rc = launch_row_bits(dtype, d_labels, cur, p.nz_y, p.rs_y, zpass ? p.zs_y : nullptr, sx, sy, sz,
                           wx, bb, bb ? 0 : 1, stream);
      if (rc != EDT_OK) return rc;
      if (binary_yz) rc = launch_planes_one_run(p.nz_y, p.rs_y, zpass ? p.zs_y : nullptr, p.gy, 0, stream);
      if (rc != EDT_OK) return rc;
    }
  } else {
    {
      ScopedPass t("x_pass", stream);
      // rows of more than 2048 voxels: one thread per VOXEL through the line pipeline (edt_line.hip); the
      // thread-per-row kernel stays behind EDT_FLAG_FORCE_GENERIC as the cross-check it is
      if (p.line_ws != nullptr) rc = launch_rows_line_pass(dtype, d_labels, cur, sx, sy * sz, wx, bb, bb ? 0 : 1, p.line_ws, stream);
      else rc = launch_row_pass_serial(dtype, d_labels, cur, sx, sy * sz, wx, bb, bb ? 0 : 1, 0, stream);
      if (rc != EDT_OK) return rc;
    }
    {
      ScopedPass t("y_bits", stream);
      rc = launch_axis_bits(dtype, d_labels, nullptr, p.nz_y, p.rs_y, p.gy, stream);
      if (rc != EDT_OK) return rc;
      if (binary_yz) rc = launch_planes_one_run(p.nz_y, p.rs_y, nullptr, p.gy, 0, stream);
      if (rc != EDT_OK) return rc;
    }
  }
  if (!index_form) {
    ScopedPass t("y_pass", stream);
    const int epi = zpass ? 0 : last_epi;
    if (tiled_y) {
      TileList list;
      bool launched = false;
      // (fp32 values of pass X: (k * wx)^2 need not be on the quantum grid for large k -- the list stays)
      rc = q16_pass(cur, nullptr, p.rs_y, p.gy, 1, epi, list, nullptr, false, launched);
      if (rc != EDT_OK) return rc;
      y_sure = false;
      rc = launch_column_inplace(cur, p.nz_y, p.rs_y, p.gy, wy, bb, epi, stream, list);
    } else {
      rc = launch_column_pass_serial(cur, other, p.nz_y, p.rs_y, p.stack, p.gy, wy, bb, epi, stream);
      std::swap(cur, other);
    }
    if (rc != EDT_OK) return rc;
  }
  if (zpass) {
    // z-packed planes (after the y pass: rs_z may live in the y pass's run-start plane)
    ScopedPass t("z_bits", stream);
    if (tiled_x) rc = launch_bits_transpose_yz(p.nz_y, p.zs_y, p.nz_z, p.rs_z, sx, sy, sz, stream);
    else {
      rc = launch_axis_bits(dtype, d_labels, nullptr, p.nz_z, p.rs_z, p.gz, stream);
      if (rc == EDT_OK && binary_yz) rc = launch_planes_one_run(p.nz_z, p.rs_z, nullptr, p.gz, 0, stream);
    }
    if (rc != EDT_OK) return rc;
  }
  if (zpass) {
    ScopedPass t("z_pass", stream);
    if (tiled_z) {
      TileList list;
      bool launched = false;
      rc = q16_pass(cur, nullptr, p.rs_z, p.gz, 2, last_epi, list, plane16 ? p.codes : nullptr, index_form && y_sure, launched);
      if (rc != EDT_OK) return rc;
      if (!list.none) rc = launch_column_inplace(cur, p.nz_z, p.rs_z, p.gz, wz, bb, last_epi, stream, list);
    } else {
      rc = launch_column_pass_serial(cur, other, p.nz_z, p.rs_z, p.stack, p.gz, wz, bb, last_epi,
                                     stream);
      std::swap(cur, other);
    }
    if (rc != EDT_OK) return rc;
  }
  if (cur != d_out) { set_error("internal: result buffer mismatch"); return EDT_ERR_HIP; }
  return EDT_OK;
}

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
// subtraction on the host): labels up once, edt(labels), the background mask and edt(mask) on the device, the
// difference down once.
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

// ---- Z-sharded phases ------------------------------------------------------------------------
struct ShardPlan {
  float *bufB = nullptr;
  int32_t *stack = nullptr;
  uint32_t *nz = nullptr, *rs = nullptr;
  size_t bytes = 0;
};

static ShardPlan make_shard_plan(int64_t sx, int64_t sy, int64_t sz, void *ws) {
  // sized for the larger of the two phases run on an (sx, sy, sz) block
  ShardPlan p;
  Carver c(ws);
  const int64_t voxels = sx * sy * sz;
  p.bufB = c.take<float>((size_t)voxels);
  p.stack = c.take<int32_t>((size_t)voxels);
  const AxisGeom gy = make_geom_y(sx, sy, sz), gz = make_geom_z(sx, sy, sz);
  const size_t words = (size_t)std::max(gy.sx * gy.nbands * gy.nouter, gz.sx * gz.nbands * gz.nouter);
  p.nz = c.take<uint32_t>(words);
  p.rs = c.take<uint32_t>(words);
  p.bytes = align_up(c.off, 256) + 256;
  return p;
}

// ---- slab records: the fast variant of the two sharded phases (edt_shard.hip) -------------------
static int64_t record_floats(int64_t sx, int64_t ylen) {
  return ylen * sx + 2 * ceil_div(ylen, kBandRows) * sx;
}

// (records of 16-bit values, where every pass runs on the integer column kernel: the rows as packed 16-bit pairs)
static int64_t record16_words(int64_t sx, int64_t ylen) {
  return ylen * sx / 2 + 2 * ceil_div(ylen, kBandRows) * sx;
}

struct RecordPlan {
  float *F = nullptr;                                      // pass 1 output of the slab (XY phase)
  uint32_t *nz_y = nullptr, *ys_y = nullptr, *zs_y = nullptr;  // y-packed planes of the slab
  uint32_t *nz_z = nullptr, *rs_z = nullptr;               // z-packed planes (Z phase)
  BandScatter *table = nullptr;
  uint32_t *q16_counts = nullptr, *q16_ids = nullptr;      // hand-over list of the integer column kernel (one phase per call)
  uint32_t *ones_map = nullptr;                            // 16-bit records, Z phase: "every row of every tile is in the plane"
  size_t bytes = 0;
};

// sized for either phase on an (sx, sy, sz) block
static RecordPlan make_record_plan(int64_t sx, int64_t sy, int64_t sz, void *ws) {
  RecordPlan p;
  Carver c(ws);
  const size_t wy = (size_t)(sx * ceil_div(sy, kBandRows) * sz);
  const size_t wz = (size_t)(sx * ceil_div(sz, kBandRows) * sy);
  p.F = c.take<float>((size_t)(sx * sy * sz));
  p.nz_y = c.take<uint32_t>(wy);
  p.ys_y = c.take<uint32_t>(wy);
  p.zs_y = c.take<uint32_t>(wy);
  p.nz_z = c.take<uint32_t>(wz);
  p.rs_z = c.take<uint32_t>(wz);
  p.table = c.take<BandScatter>(1);
  p.q16_counts = c.take<uint32_t>(4);
  p.q16_ids = c.take<uint32_t>((size_t)(ceil_div(sx, 16) * (ceil_div(std::max(sy, sz), 8) * 8)));
  p.ones_map = c.take<uint32_t>((size_t)(ceil_div(sx, 32) * ceil_div(std::max(sy, sz), 32)));
  p.bytes = align_up(c.off, 256) + 256;
  return p;
}

}  // namespace edt_amd

using namespace edt_amd;

extern "C" {

int edt_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char *edt_hip_last_error(void) { return g_last_error.c_str(); }

int edt_hip_index_form_exact(float wx, int64_t sx) { return (sx >= 1 && row_codes_exact(wx, sx)) ? 1 : 0; }

const char *edt_hip_version(void) { return "edt_hip 0.1 (gfx950)"; }

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

size_t edt_hip_workspace_bytes_flags(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (check_shape(dtype, ndim, sx, sy, sz) != EDT_OK) return 0;
  if (sx == 0 || sy == 0 || sz == 0) return 256;
  return make_plan(dtype, ndim, sx, sy, sz, nullptr, flags).bytes;
}

size_t edt_hip_workspace_bytes(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz) {
  return edt_hip_workspace_bytes_flags(dtype, ndim, sx, sy, sz, 0);
}

int edt_hip_edtsq_device(const void *d_labels, int dtype, int ndim, int64_t sx, int64_t sy,
                         int64_t sz, float wx, float wy, float wz, int flags, float *d_output,
                         void *d_workspace, size_t workspace_bytes, void *stream) {
  return run_device(d_labels, dtype, ndim, sx, sy, sz, wx, wy, wz, flags, d_output, d_workspace,
                    workspace_bytes, (hipStream_t)stream);
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

int edt_hip_set_debug_mode(int mode) {
  set_thread_debug_mode(mode);
  return EDT_OK;
}

int edt_hip_get_debug_mode(void) { return debug_mode(); }

int edt_hip_set_profiling(int enabled) {
  std::lock_guard<std::mutex> lock(g_log_mutex);
  g_log.enabled.store(enabled != 0);
  log_begin_call();  // the shard phases append to the log (several calls make one step): start clean
  return EDT_OK;
}

int edt_hip_get_pass_times(float *ms, int capacity) {
  std::lock_guard<std::mutex> lock(g_log_mutex);
  int n = (int)g_log.span.size();
  for (int i = 0; i < n && i < capacity; ++i) {
    float t = 0.0f;
    if (hipEventElapsedTime(&t, g_log.pool[g_log.span[i].first], g_log.pool[g_log.span[i].second]) !=
        hipSuccess) {
      (void)hipGetLastError();
      t = -1.0f;
    }
    ms[i] = t;
  }
  return n;
}

const char *edt_hip_get_pass_name(int index) {
  std::lock_guard<std::mutex> lock(g_log_mutex);
  if (index < 0 || index >= (int)g_log.names.size()) return "";
  return g_log.names[index].c_str();
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

size_t edt_hip_shard_workspace_bytes(int dtype, int64_t sx, int64_t sy, int64_t sz) {
  if (check_shape(dtype, 3, sx, sy, sz) != EDT_OK) return 0;
  if (sx == 0 || sy == 0 || sz == 0) return 256;
  return make_shard_plan(sx, sy, sz, nullptr).bytes;
}

int edt_hip_shard_xy_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx,
                            int64_t sy, int64_t sz_local, float wx, float wy, int flags,
                            float *d_partial, uint8_t *d_zflags, void *d_workspace,
                            size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(dtype, 3, sx, sy, sz_local);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(2, wx, wy, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy == 0 || sz_local == 0) return EDT_OK;
  if (!d_labels || !d_partial || !d_zflags) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  ShardPlan p = make_shard_plan(sx, sy, sz_local, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const bool force_generic = (flags & EDT_FLAG_FORCE_GENERIC) != 0;
  AxisGeom gy = make_geom_y(sx, sy, sz_local);
  gy.fmin = edt_hip_field_floor(wx, wx);  // (pass Y reads the results of pass X: AxisGeom::fmin)
  const bool tiled_x = !force_generic && (row_pass_tiled_supported(sx) || row_pass_wave_supported(dtype, sx, sy, sz_local));
  const bool tiled_y = !force_generic && column_inplace_supported(gy);
  float *xout = tiled_y ? d_partial : p.bufB;  // the tiled y pass runs in place
  if (tiled_x) {
    rc = launch_row_bits(dtype, d_labels, xout, p.nz, p.rs, nullptr, sx, sy, sz_local, wx, bb, bb ? 0 : 1,
                         stream);
    if (rc != EDT_OK) return rc;
  } else {
    // rows of more than 2048 voxels: the line pipeline (a thread per voxel), its scratch borrowed from the hull
    // stacks, which only the size-agnostic column pass uses -- later on this stream
    if (!force_generic && rows_line_workspace_bytes(sx, sy * sz_local) <= (size_t)(sx * sy * sz_local) * sizeof(int32_t))
      rc = launch_rows_line_pass(dtype, d_labels, xout, sx, sy * sz_local, wx, bb, bb ? 0 : 1, p.stack, stream);
    else
      rc = launch_row_pass_serial(dtype, d_labels, xout, sx, sy * sz_local, wx, bb, bb ? 0 : 1, 0, stream);
    if (rc != EDT_OK) return rc;
    rc = launch_axis_bits(dtype, d_labels, nullptr, p.nz, p.rs, gy, stream);
    if (rc != EDT_OK) return rc;
  }
  if (tiled_y) rc = launch_column_inplace(d_partial, p.nz, p.rs, gy, wy, bb, 0, stream);
  else rc = launch_column_pass_serial(p.bufB, d_partial, p.nz, p.rs, p.stack, gy, wy, bb, 0, stream);
  if (rc != EDT_OK) return rc;
  return launch_zflags(dtype, d_labels, d_halo, d_zflags, sx * sy, sz_local, stream);
}

int edt_hip_shard_z_device(float *d_partial, const uint8_t *d_zflags, int64_t sx, int64_t sy_local,
                           int64_t sz, float wz, int flags, void *d_workspace,
                           size_t workspace_bytes, void *stream_) {
  return edt_hip_shard_z_device_ex(d_partial, d_zflags, sx, sy_local, sz, wz, 0.0f, flags, d_workspace, workspace_bytes,
                                   stream_);
}

int edt_hip_shard_z_device_ex(float *d_partial, const uint8_t *d_zflags, int64_t sx, int64_t sy_local,
                              int64_t sz, float wz, float field_floor, int flags, void *d_workspace,
                              size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(EDT_U8, 3, sx, sy_local, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(1, wz, 1.0f, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy_local == 0 || sz == 0) return EDT_OK;
  if (!d_partial || !d_zflags) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  ShardPlan p = make_shard_plan(sx, sy_local, sz, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0);
  AxisGeom gz = make_geom_z(sx, sy_local, sz);
  gz.fmin = field_floor > 0.0f ? field_floor : 0.0f;  // (AxisGeom::fmin; NaN and negatives: unknown)
  rc = launch_bits_from_flags(d_zflags, p.nz, p.rs, gz, stream);
  if (rc != EDT_OK) return rc;
  if (!(flags & EDT_FLAG_FORCE_GENERIC) && column_inplace_supported(gz))
    return launch_column_inplace(d_partial, p.nz, p.rs, gz, wz, bb, epi, stream);
  rc = launch_column_pass_serial(d_partial, p.bufB, p.nz, p.rs, p.stack, gz, wz, bb, epi, stream);
  if (rc != EDT_OK) return rc;
  EDT_HIP_TRY(hipMemcpyAsync(d_partial, p.bufB, (size_t)(sx * sy_local * sz) * sizeof(float),
                             hipMemcpyDeviceToDevice, stream));
  return EDT_OK;
}

// min(fl32(wx*wx), fl32(wy*wy)): what every non-zero value of a field is at least after passes X and Y (run_device has
// the argument); 0 where a voxel size is not a positive finite number
float edt_hip_field_floor(float wx, float wy) {
  const float a = wx * wx, b = wy * wy;
  if (!(a > 0.0f) || !(b > 0.0f) || !(a < INFINITY) || !(b < INFINITY)) return 0.0f;
  return a < b ? a : b;
}

int edt_hip_shard_records_supported(int dtype, int64_t sx, int64_t sy, int64_t sz) {
  if (dtype_size(dtype) == 0 || sx < 1 || sy < 1 || sz < 1) return 0;
  if (g_debug_mode & (32 | 64)) return 0;  // diagnostics: forced fallback kernels
  // pass 1 by the register-resident row kernel (two waves per row beyond 1024 voxels), both column passes by the wave kernel
  return (sx <= 2048 && row_pass_wave_supported(dtype, sx, sy, sz) && sy <= 2048 && sz <= 2048) ? 1 : 0;
}

size_t edt_hip_shard_record_floats(int64_t sx, int64_t y_rows) {
  if (sx < 0 || y_rows < 0) return 0;
  return (size_t)record_floats(sx, y_rows);
}

size_t edt_hip_shard_records_workspace_bytes(int dtype, int64_t sx, int64_t sy, int64_t sz) {
  if (check_shape(dtype, 3, sx, sy, sz) != EDT_OK) return 0;
  if (sx == 0 || sy == 0 || sz == 0) return 256;
  return make_record_plan(sx, sy, sz, nullptr).bytes;
}

int edt_hip_shard_xy_records_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx,
                                    int64_t sy, int64_t sz_local, float wx, float wy, int flags,
                                    int nparts, const int64_t *y_splits, void *const *d_blocks,
                                    void *d_workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(dtype, 3, sx, sy, sz_local);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(2, wx, wy, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy == 0 || sz_local == 0) return EDT_OK;
  if (!d_labels || !y_splits || !d_blocks || nparts < 1) { set_error("null argument"); return EDT_ERR_BAD_ARG; }
  if (!edt_hip_shard_records_supported(dtype, sx, sy, sz_local)) {
    set_error("slab records need sx <= 2048 and sy <= 2048 (use edt_hip_shard_xy_device)");
    return EDT_ERR_UNSUPPORTED;
  }
  if (y_splits[0] != 0 || y_splits[nparts] != sy) { set_error("y_splits must run from 0 to sy"); return EDT_ERR_BAD_ARG; }
  for (int h = 0; h < nparts; ++h) {
    if (y_splits[h + 1] <= y_splits[h] || (y_splits[h] % kBandRows) != 0) {
      set_error("y_splits must be increasing multiples of 32 (the last one is sy)");
      return EDT_ERR_BAD_ARG;
    }
    if (!d_blocks[h]) { set_error("null destination block"); return EDT_ERR_BAD_ARG; }
  }
  if (g_log.enabled.load(std::memory_order_relaxed)) {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    if (g_log.used > 2048) log_begin_call();  // nobody is reading the log
  }
  RecordPlan p = make_record_plan(sx, sy, sz_local, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  AxisGeom gy = make_geom_y(sx, sy, sz_local);
  gy.fmin = edt_hip_field_floor(wx, wx);  // (pass Y reads the results of pass X: AxisGeom::fmin)
  // destination map: every 32-row band of y lies inside one part
  BandScatter sc;
  bool aligned = (sx % 4) == 0;
  for (int b = 0, h = 0; b < BandScatter::kBands; ++b) {
    if (b >= gy.nbands) { sc.rows[b] = nullptr; sc.bits[b] = nullptr; sc.ostride[b] = 0; sc.plane[b] = 0; continue; }
    while ((int64_t)b * kBandRows >= y_splits[h + 1]) ++h;
    const int64_t ys = y_splits[h], ylen = y_splits[h + 1] - ys, words = ceil_div(ylen, kBandRows);
    float *blk = static_cast<float *>(d_blocks[h]);
    sc.rows[b] = blk + ((int64_t)b * kBandRows - ys) * sx;
    sc.bits[b] = reinterpret_cast<uint32_t *>(blk + ylen * sx) + ((int64_t)b - ys / kBandRows) * sx;
    sc.ostride[b] = record_floats(sx, ylen);
    sc.plane[b] = words * sx;
    aligned = aligned && (reinterpret_cast<uintptr_t>(blk) % 16) == 0;
  }
  if (!aligned && (sx % 4) == 0) { set_error("destination blocks must be 16-byte aligned"); return EDT_ERR_BAD_ARG; }
  // (index form of pass 1 where the voxel size allows it, see run_device: the slab's pass-1 buffer then holds 16-bit
  // indices in its first half)
  const bool index_form = (sx % 4) == 0 && !(g_debug_mode & 0x100000) && row_codes_exact(wx, sx);
  uint16_t *codes = index_form ? reinterpret_cast<uint16_t *>(p.F) : nullptr;
  {
    ScopedPass t("x_pass", stream);
    rc = launch_row_pass_wave(dtype, d_labels, p.F, p.nz_y, p.ys_y, p.zs_y, sx, sy, sz_local, wx, bb,
                              bb ? 0 : 1, stream, d_halo, codes);
    if (rc != EDT_OK) return rc;
  }
  {
    ScopedPass t("pack_bits", stream);
    rc = launch_pack_record_bits(p.nz_y, p.zs_y, sc, p.table, sx, gy.nbands, sz_local, stream);
    if (rc != EDT_OK) return rc;
  }
  ScopedPass t("y_pass", stream);
  // the integer column kernel where wx and wy share a quantum (edt_colq16.hip), the tiles it refuses to the fp32 kernel
  TileList list;
  {
    const float w2[2] = {wx, wy};
    float q = 1.0f;
    uint32_t a[3];
    if (!(g_debug_mode & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000)) && q16_quantum(w2, 2, &q, a) &&
        column_pass_q16_supported(gy) && column_pass_wave_supported(gy) && aligned) {
      EDT_HIP_TRY(hipMemsetAsync(p.q16_counts, 0, 4 * sizeof(uint32_t), stream));
      rc = launch_column_pass_q16(p.F, codes, p.ys_y, gy, q, a[1], a[0], bb, 0, p.q16_counts, p.q16_ids, stream, p.table);
      if (rc != EDT_OK) return rc;
      list.count = p.q16_counts;
      list.ids = p.q16_ids;
    }
  }
  if (index_form)
    return launch_column_pass_wave_codes(p.F, codes, p.nz_y, p.ys_y, gy, wy, bb, 0, wx, bb ? 0 : 1, stream, p.table, list);
  return launch_column_pass_wave(p.F, p.nz_y, p.ys_y, gy, wy, bb, 0, stream, p.table, ColumnOut(), list);
}

int edt_hip_shard_z_records_device(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wz,
                                   int flags, void *d_workspace, size_t workspace_bytes, void *stream_) {
  return edt_hip_shard_z_records_device_ex(d_records, sx, sy_local, sz, wz, 0.0f, flags, d_workspace, workspace_bytes,
                                           stream_);
}

static int shard_z_records(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wz, float field_floor,
                           const float *w3, int flags, void *d_workspace, size_t workspace_bytes, void *stream_);

int edt_hip_shard_z_records_device_ex(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wz,
                                      float field_floor, int flags, void *d_workspace, size_t workspace_bytes,
                                      void *stream_) {
  return shard_z_records(d_records, sx, sy_local, sz, wz, field_floor, nullptr, flags, d_workspace, workspace_bytes, stream_);
}

int edt_hip_shard_z_records_device_w(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wx, float wy,
                                     float wz, int flags, void *d_workspace, size_t workspace_bytes, void *stream_) {
  const float w3[3] = {wx, wy, wz};
  return shard_z_records(d_records, sx, sy_local, sz, wz, edt_hip_field_floor(wx, wy), w3, flags, d_workspace,
                         workspace_bytes, stream_);
}

// w3 != nullptr: the caller named all three voxel sizes -- the integer column kernel where they share a quantum
static int shard_z_records(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wz, float field_floor,
                           const float *w3, int flags, void *d_workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(EDT_U8, 3, sx, sy_local, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(1, wz, 1.0f, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy_local == 0 || sz == 0) return EDT_OK;
  if (!d_records) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  if (!edt_hip_shard_records_supported(EDT_U8, sx, sy_local, sz)) {
    set_error("slab records need sx <= 2048 and sz <= 2048 (use edt_hip_shard_z_device)");
    return EDT_ERR_UNSUPPORTED;
  }
  RecordPlan p = make_record_plan(sx, sy_local, sz, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0);
  const int64_t rec = record_floats(sx, sy_local), words = ceil_div(sy_local, kBandRows);
  const uint32_t *nz_y = reinterpret_cast<const uint32_t *>(d_records + sy_local * sx);
  {
    ScopedPass t("z_bits", stream);
    rc = launch_bits_transpose_yz(nz_y, nz_y + words * sx, p.nz_z, p.rs_z, sx, sy_local, sz, stream, rec);
    if (rc != EDT_OK) return rc;
  }
  AxisGeom gz;  // z-columns of the record buffer: consecutive z are one record apart
  gz.sx = sx; gz.n = sz; gz.stride = rec; gz.nouter = sy_local; gz.outer_stride = sx;
  gz.nbands = ceil_div(sz, kBandRows);
  gz.fmin = field_floor > 0.0f ? field_floor : 0.0f;
  ScopedPass t("z_pass", stream);
  TileList list;
  if (w3 != nullptr) {
    float q = 1.0f;
    uint32_t a[3];
    if (!(g_debug_mode & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000)) && q16_quantum(w3, 3, &q, a) &&
        column_pass_q16_supported(gz) && column_pass_wave_supported(gz) && (reinterpret_cast<uintptr_t>(d_records) % 16) == 0) {
      EDT_HIP_TRY(hipMemsetAsync(p.q16_counts, 0, 4 * sizeof(uint32_t), stream));
      rc = launch_column_pass_q16(d_records, nullptr, p.rs_z, gz, q, a[2], a[0], bb, epi, p.q16_counts, p.q16_ids, stream);
      if (rc != EDT_OK) return rc;
      list.count = p.q16_counts;
      list.ids = p.q16_ids;
    }
  }
  return launch_column_pass_wave(d_records, p.nz_z, p.rs_z, gz, wz, bb, epi, stream, nullptr, ColumnOut(), list);
}

// ---- slab records of 16-bit values ---------------------------------------------------------------------------------------
// Where the three voxel sizes share a quantum (edt_colq16.hip) and both column axes fit the integer kernel, the Y pass's
// results are integers N < 2^16 (in quanta): a record then carries its rows as 16-bit values -- 2.25 bytes per voxel over
// the links instead of 4.25 -- and the Z phase reads them as they are.  A tile the integer kernel cannot take (values beyond
// 16 bits, rows without a boundary) has no 16-bit form: the XY phase COUNTS such tiles in *d_refused (a device counter the
// caller zeroes and reads; it accumulates over calls) and leaves their rows unspecified -- a caller that finds it non-zero
// repeats the step with the fp32 records above (edt/distributed.py does).
static bool records16_common_ok(int64_t sx, float wx, float wy, float wz) {
  if (sx % 4 != 0 || (g_debug_mode & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x100000 | 0x8000000 | 0x10000000))) return false;
  if (!row_codes_exact(wx, sx)) return false;
  const float w3[3] = {wx, wy, wz};
  float q = 1.0f;
  uint32_t a[3];
  return q16_quantum(w3, 3, &q, a);
}
// the XY phase of a slab of sz_local slices / the Z phase of a slab of sy_local rows: the scan axis on the integer kernel
static bool records16_xy_ok(int dtype, int64_t sx, int64_t sy, int64_t sz_local, float wx, float wy, float wz) {
  if (!edt_hip_shard_records_supported(dtype, sx, sy, sz_local) || !records16_common_ok(sx, wx, wy, wz)) return false;
  const AxisGeom gy = make_geom_y(sx, sy, sz_local);
  return column_pass_q16_supported(gy) && column_pass_wave_supported(gy);
}
static bool records16_z_ok(int64_t sx, int64_t sy_local, int64_t sz, float wx, float wy, float wz) {
  if (!edt_hip_shard_records_supported(EDT_U8, sx, sy_local, sz) || !records16_common_ok(sx, wx, wy, wz)) return false;
  const AxisGeom gz = make_geom_z(sx, sy_local, sz);
  return column_pass_q16_supported(gz) && column_pass_wave_supported(gz);
}

int edt_hip_shard_records16_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz) {
  if (dtype_size(dtype) == 0 || sx < 1 || sy < 1 || sz < 1) return 0;
  // (whatever part of z or y a rank holds, its scan axis is whole: sy for the XY phase, sz for the Z phase)
  return (records16_xy_ok(dtype, sx, sy, 1, wx, wy, wz) && records16_z_ok(sx, 32, sz, wx, wy, wz)) ? 1 : 0;
}

size_t edt_hip_shard_record16_words(int64_t sx, int64_t y_rows) {
  if (sx < 0 || y_rows < 0 || sx % 2 != 0) return 0;
  return (size_t)record16_words(sx, y_rows);
}

int edt_hip_shard_xy_records16_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx, int64_t sy,
                                      int64_t sz_local, float wx, float wy, float wz, int flags, int nparts,
                                      const int64_t *y_splits, void *const *d_blocks, uint32_t *d_refused, void *d_workspace,
                                      size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(dtype, 3, sx, sy, sz_local);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(2, wx, wy, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy == 0 || sz_local == 0) return EDT_OK;
  if (!d_labels || !y_splits || !d_blocks || !d_refused || nparts < 1) { set_error("null argument"); return EDT_ERR_BAD_ARG; }
  if (y_splits[0] != 0 || y_splits[nparts] != sy) { set_error("y_splits must run from 0 to sy"); return EDT_ERR_BAD_ARG; }
  for (int h = 0; h < nparts; ++h) {
    if (y_splits[h + 1] <= y_splits[h] || (y_splits[h] % kBandRows) != 0) {
      set_error("y_splits must be increasing multiples of 32 (the last one is sy)");
      return EDT_ERR_BAD_ARG;
    }
    // (4-byte stores of packed pairs and bit words; the Z phase reads 8 bytes at a time: records are an even number of words)
    if (!d_blocks[h] || (reinterpret_cast<uintptr_t>(d_blocks[h]) % 8) != 0) {
      set_error("destination blocks must be non-null and 8-byte aligned");
      return EDT_ERR_BAD_ARG;
    }
  }
  const float w3[3] = {wx, wy, wz};
  float q = 1.0f;
  uint32_t a[3];
  AxisGeom gy = make_geom_y(sx, sy, sz_local);
  if (!records16_xy_ok(dtype, sx, sy, sz_local, wx, wy, wz) || !q16_quantum(w3, 3, &q, a)) {
    set_error("16-bit slab records do not apply to these extents / voxel sizes (edt_hip_shard_records16_supported)");
    return EDT_ERR_UNSUPPORTED;
  }
  RecordPlan p = make_record_plan(sx, sy, sz_local, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  // destination map in 4-byte words: a record = ylen * sx / 2 words of 16-bit pairs, then the two bit planes
  BandScatter sc;
  for (int b = 0, h = 0; b < BandScatter::kBands; ++b) {
    if (b >= gy.nbands) { sc.rows[b] = nullptr; sc.bits[b] = nullptr; sc.ostride[b] = 0; sc.plane[b] = 0; continue; }
    while ((int64_t)b * kBandRows >= y_splits[h + 1]) ++h;
    const int64_t ys = y_splits[h], ylen = y_splits[h + 1] - ys, words = ceil_div(ylen, kBandRows);
    float *blk = static_cast<float *>(d_blocks[h]);
    sc.rows[b] = blk + (((int64_t)b * kBandRows - ys) * sx) / 2;
    sc.bits[b] = reinterpret_cast<uint32_t *>(blk + ylen * sx / 2) + ((int64_t)b - ys / kBandRows) * sx;
    sc.ostride[b] = record16_words(sx, ylen);
    sc.plane[b] = words * sx;
  }
  uint16_t *codes = reinterpret_cast<uint16_t *>(p.F);
  {
    ScopedPass t("x_pass", stream);
    rc = launch_row_pass_wave(dtype, d_labels, p.F, p.nz_y, p.ys_y, p.zs_y, sx, sy, sz_local, wx, bb, bb ? 0 : 1, stream,
                              d_halo, codes);
    if (rc != EDT_OK) return rc;
  }
  {
    ScopedPass t("pack_bits", stream);
    rc = launch_pack_record_bits(p.nz_y, p.zs_y, sc, p.table, sx, gy.nbands, sz_local, stream);
    if (rc != EDT_OK) return rc;
  }
  ScopedPass t("y_pass", stream);
  // (plane: any non-null value selects the 16-bit output; the destinations are the table's)
  return launch_column_pass_q16(p.F, codes, p.ys_y, gy, q, a[1], a[0], bb, 0, d_refused, nullptr, stream, p.table, codes);
}

int edt_hip_shard_z_records16_device(const void *d_records, float *d_out, int64_t sx, int64_t sy_local, int64_t sz, float wx,
                                     float wy, float wz, int flags, void *d_workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(EDT_U8, 3, sx, sy_local, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(1, wz, 1.0f, 1.0f)) != EDT_OK) return rc;
  if (sx == 0 || sy_local == 0 || sz == 0) return EDT_OK;
  if (!d_records || !d_out) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  const float w3[3] = {wx, wy, wz};
  float q = 1.0f;
  uint32_t a[3];
  AxisGeom gz = make_geom_z(sx, sy_local, sz);  // the dense output: z-columns one (sy_local, sx) slice apart
  gz.fmin = edt_hip_field_floor(wx, wy);
  if (!records16_z_ok(sx, sy_local, sz, wx, wy, wz) || !q16_quantum(w3, 3, &q, a) ||
      ((reinterpret_cast<uintptr_t>(d_records) | reinterpret_cast<uintptr_t>(d_out)) % 16) != 0) {
    set_error("16-bit slab records do not apply to these extents / voxel sizes (edt_hip_shard_records16_supported)");
    return EDT_ERR_UNSUPPORTED;
  }
  RecordPlan p = make_record_plan(sx, sy_local, sz, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0);
  const int64_t rec = record16_words(sx, sy_local), words = ceil_div(sy_local, kBandRows);
  const uint32_t *base = static_cast<const uint32_t *>(d_records);
  const uint32_t *nz_y = base + sy_local * sx / 2;
  {
    ScopedPass t("z_bits", stream);
    rc = launch_bits_transpose_yz(nz_y, nz_y + words * sx, p.nz_z, p.rs_z, sx, sy_local, sz, stream, rec);
    if (rc != EDT_OK) return rc;
  }
  ScopedPass t("z_pass", stream);
  // every row out of the records (16-bit elements: consecutive z are 2 * rec of them apart), results to the dense array; a
  // tile beyond THIS pass's limits gets its rows written there as fp32 values and goes to the fp32 kernel, in place
  EDT_HIP_TRY(hipMemsetAsync(p.q16_counts, 0, 4 * sizeof(uint32_t), stream));
  const int map_words = (int)ceil_div(sz, 32);
  EDT_HIP_TRY(hipMemsetAsync(p.ones_map, 0xFF, (size_t)(ceil_div(sx, 32) * map_words) * sizeof(uint32_t), stream));
  uint16_t *plane = reinterpret_cast<uint16_t *>(const_cast<void *>(d_records));
  rc = launch_column_pass_q16(d_out, nullptr, p.rs_z, gz, q, a[2], a[0], bb, epi, p.q16_counts, p.q16_ids, stream, nullptr, plane,
                              p.ones_map, map_words, nullptr, 2 * rec, sx);
  if (rc != EDT_OK) return rc;
  TileList list;
  list.count = p.q16_counts;
  list.ids = p.q16_ids;
  return launch_column_pass_wave(d_out, p.nz_z, p.rs_z, gz, wz, bb, epi, stream, nullptr, ColumnOut(), list);
}

int edt_hip_subtract_device(const float *d_a, const float *d_b, float *d_out, int64_t count,
                            void *stream) {
  return launch_subtract(d_a, d_b, d_out, count, (hipStream_t)stream);
}

// Voxel-graph transform on device-resident data.  Native form (edt_voxel_graph.hip: no doubled volume) where
// the wave column kernel covers the doubled axes, else the up-sampled form: workspace = [2x uint8 volume |
// its fp32 transform | the ordinary workspace of the 2x volume].  Debug bit 0x20000 forces the latter.
static bool vg_use_native(int ndim, int64_t sx, int64_t sy, int64_t sz) {
  return !(g_debug_mode & 0x20000) && vg_native_supported(ndim, sx, sy, sz);
}

size_t edt_hip_voxel_graph_workspace_bytes(int ndim, int64_t sx, int64_t sy, int64_t sz) {
  if (check_shape(EDT_U8, ndim, sx, sy, sz) != EDT_OK || ndim < 2) return 0;
  if (sx == 0 || sy == 0 || sz == 0) return 256;
  if (vg_use_native(ndim, sx, sy, sz)) return vg_native_workspace_bytes(ndim, sx, sy, sz);
  const int64_t X = 2 * sx, Y = 2 * sy, Z = (ndim == 3) ? 2 * sz : 1;
  const size_t big = (size_t)(X * Y * Z);
  return align_up(big, 256) + align_up(big * sizeof(float), 256) + edt_hip_workspace_bytes(EDT_U8, ndim, X, Y, Z) + 256;
}

int edt_hip_edtsq_voxel_graph_device(const void *d_labels, int dtype, const uint8_t *d_graph, int ndim,
                                     int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                                     int flags, float *d_output, void *d_workspace, size_t workspace_bytes,
                                     void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape(dtype, ndim, sx, sy, sz);
  if (rc != EDT_OK) return rc;
  if ((rc = check_voxel_sizes(ndim, wx, wy, wz)) != EDT_OK) return rc;
  if (ndim < 2) { set_error("voxel_graph needs a 2-D or 3-D volume"); return EDT_ERR_BAD_ARG; }
  if (sx == 0 || sy == 0 || sz == 0) return EDT_OK;
  if (!d_labels || !d_graph || !d_output) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  const size_t need = edt_hip_voxel_graph_workspace_bytes(ndim, sx, sy, sz);
  if (!d_workspace || workspace_bytes < need) {
    set_error("workspace too small: need " + std::to_string(need) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  if (vg_use_native(ndim, sx, sy, sz)) {
    ScopedPass t("voxel_graph", stream);
    return launch_vg_native(dtype, d_labels, d_graph, ndim, sx, sy, sz, wx, wy, wz, bb,
                            (flags & EDT_FLAG_SQRT) ? 1 : 0, d_output, d_workspace, stream);
  }
  const int64_t X = 2 * sx, Y = 2 * sy, Z = (ndim == 3) ? 2 * sz : 1;
  const size_t big = (size_t)(X * Y * Z);
  Carver c(d_workspace);
  uint8_t *d_big = c.take<uint8_t>(big);
  float *d_bigdt = c.take<float>(big);
  const size_t wbytes = edt_hip_workspace_bytes(EDT_U8, ndim, X, Y, Z);
  void *d_ws = c.take<unsigned char>(wbytes);
  rc = launch_vg_expand(dtype, d_labels, d_graph, d_big, sx, sy, sz, ndim, bb, stream);
  if (rc != EDT_OK) return rc;
  // half voxel size on the 2x grid (src/edt_voxel_graph.hpp:96-101, :189-193)
  rc = run_device(d_big, EDT_U8, ndim, X, Y, Z, wx / 2, wy / 2, wz / 2,
                  (bb ? EDT_FLAG_BLACK_BORDER : 0) | (flags & EDT_FLAG_SQRT), d_bigdt, d_ws, wbytes, stream);
  if (rc != EDT_OK) return rc;
  return launch_vg_gather(d_bigdt, d_output, sx, sy, sz, ndim, stream);
}

int edt_hip_select_label_device(const void *d_labels, int dtype, const float *d_dt, const void *key,
                                float *d_out, int64_t count, void *stream) {
  if (count < 0 || dtype_size(dtype) == 0) { set_error("bad argument"); return EDT_ERR_BAD_ARG; }
  if (count == 0) return EDT_OK;
  if (!d_labels || !d_dt || !d_out || !key) { set_error("null pointer"); return EDT_ERR_BAD_ARG; }
  return launch_select_label(dtype, d_labels, d_dt, d_out, key, count, (hipStream_t)stream);
}

size_t edt_hip_runs_workspace_bytes(int64_t count) { return count < 0 ? 0 : runs_workspace_bytes(count); }

int edt_hip_extract_runs_device(const void *d_labels, int dtype, int64_t count, int64_t *d_starts, int64_t capacity,
                                int64_t *d_count, void *d_workspace, size_t workspace_bytes, void *stream) {
  if (count < 0 || capacity < 0 || dtype_size(dtype) == 0) { set_error("bad argument"); return EDT_ERR_BAD_ARG; }
  if (!d_count) { set_error("null pointer"); return EDT_ERR_BAD_ARG; }
  if (count == 0) { EDT_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int64_t), (hipStream_t)stream)); return EDT_OK; }
  if (!d_labels || !d_workspace || workspace_bytes < runs_workspace_bytes(count)) {
    set_error("null pointer or workspace too small (edt_hip_runs_workspace_bytes)");
    return EDT_ERR_BAD_ARG;
  }
  return launch_extract_runs(dtype, d_labels, count, d_starts, capacity, d_count, d_workspace, (hipStream_t)stream);
}

int edt_hip_is_background_device(const void *d_labels, int dtype, uint8_t *d_mask, int64_t count,
                                 void *stream) {
  return launch_is_background(dtype, d_labels, d_mask, count, (hipStream_t)stream);
}

}  // extern "C"
