// edt_api.hip -- the C ABI (include/edt_hip.h), part 1 of 3: the plan of a call and the dispatch of its passes on
// device-resident data (run_device), workspace carving, per-pass event timing, the device entry points that are not
// sharded.  Part 2: edt_host.hip (host-buffer staging); part 3: edt_shard_api.hip (the phases of the Z-sharded path).
// No compute happens on the host and there is no CPU fallback: every entry point needs a HIP device.
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

#include "edt_api_internal.h"

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



// ---- per-pass event timing (bench.py reads this): edt_api_internal.h ------------------
PassLog g_log;
std::mutex g_log_mutex;

hipEvent_t log_event() {
  if (g_log.used == (int)g_log.pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_log.pool.push_back(e);
  }
  return g_log.pool[g_log.used++];
}

void log_begin_call() {
  g_log.used = 0;
  g_log.span.clear();
  g_log.names.clear();
}



struct Plan {
  int ndim;
  int64_t sx, sy, sz, voxels;
  AxisGeom gy, gz;
  // carved pointers
  float *bufB = nullptr;
  int32_t *stack = nullptr;
  uint32_t *nz_y = nullptr, *rs_y = nullptr, *zs_y = nullptr, *nz_z = nullptr, *rs_z = nullptr;
  unsigned char *line_ws = nullptr;  // scratch of the line pipeline when pass 1 runs over rows no row kernel takes (> 4096 voxels)
  uint16_t *codes = nullptr;  // 16-bit distance indices of pass 1 (index form): one slab of xy_slab slices, or the whole volume
  int64_t xy_slab = 0;        // slices per slab of the slab-wise X/Y passes (0: no index form for this shape)
  bool codes_whole = false;   // the index buffer holds every slice (volumes of several slabs: round 6) -- the 16-bit plane between
                              // passes Y and Z then exists there too
  int64_t code_pitch = 0;     // elements between the slices of the index buffer: sx * sy, or padded (plane_pad_elems)
  // the tiles the 16-bit integer column kernel hands to the fp32 kernel (edt_colq16.hip): kQ16Slots counters, one per
  // column-pass launch of a call, and one array of tile ids (launches are stream-ordered: the array is reused)
  uint32_t *q16_counts = nullptr, *q16_ids = nullptr;
  // which tiles of pass Y left their results in the 16-bit plane (= codes): one bit per (x-tile, z), behind the counters
  uint32_t *q16_map = nullptr;
  int q16_map_words = 0;  // words per x-tile
  int64_t q16_id_capacity = 0;  // tile ids q16_ids holds
  size_t bytes = 0;
};

AxisGeom make_geom_y(int64_t sx, int64_t sy, int64_t sz) {
  AxisGeom g;
  g.sx = sx; g.n = sy; g.stride = sx; g.nouter = sz; g.outer_stride = sx * sy;
  g.nbands = ceil_div(sy, kBandRows);
  return g;
}
AxisGeom make_geom_z(int64_t sx, int64_t sy, int64_t sz) {
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
// Volumes of several slabs (1024^3: eight) keep the indices of EVERY slab where that costs at most this many bytes (2 GiB at
// 1024^3, against 288 GB of HBM): pass Y then leaves its results in the 16-bit plane there as well and pass Z reads 2 bytes per
// voxel instead of 4 (round 4 measured it at cfg4 and did not keep it for the workspace; round 6 does).  Passes X and Y still run
// slab by slab, so that a slab's indices are read back while they are warm.  EDT_HIP_WHOLE_INDEX_BYTES overrides the limit (0: off).
static int64_t whole_index_limit() {
  static const int64_t v = [] {
    const char *e = std::getenv("EDT_HIP_WHOLE_INDEX_BYTES");
    return e ? (int64_t)std::strtoll(e, nullptr, 0) : (int64_t)8 << 30;
  }();
  return v;
}
// The pitch of the index buffer = of the 16-bit plane between passes Y and Z (round 6).  Pass Z walks columns whose rows are one
// slice apart: 2 * sx * sy bytes in the plane it reads, 4 * sx * sy in the caller's array it writes.  Where the plane's slice is a
// whole multiple of 1 MiB both streams of a tile fall onto the same few channels at once and the pass loses a third of its rate
// (tools/zorder_probe.hip, profiles/r06_zorder_probe.txt: the bare access pattern at 1024^3 2.10 ms, 1.36 with the plane's slices 8 KiB
// further apart; the caller's 4 MiB stride alone -- stores only -- costs nothing).  The caller's array is not ours; the plane is:
// slices that are a multiple of 2 MiB lie 8 KiB further apart, slices that are a multiple of 1 MiB 4 KiB (8 KiB is the one bad
// choice there), every other shape is left alone (512^3: 512 KiB slices are best as they are).  EDT_HIP_PLANE_PAD_BYTES overrides
// (0: never; a multiple of 8).
static int64_t plane_pad_elems(int64_t sx, int64_t sy) {
  // (read per plan, not once: the test tiers force a pad onto small shapes; a workspace sized under another value is refused
  // by the size check of the call, never overrun)
  const char *e = std::getenv("EDT_HIP_PLANE_PAD_BYTES");
  const int64_t forced = (e && *e) ? (int64_t)std::strtoll(e, nullptr, 0) : (int64_t)-1;
  if (forced >= 0) return (forced & ~(int64_t)7) / 2;
  const int64_t slice = sx * sy * (int64_t)sizeof(uint16_t);
  if (slice % ((int64_t)2 << 20) == 0) return 8192 / 2;
  if (slice % ((int64_t)1 << 20) == 0) return 4096 / 2;
  return 0;
}
constexpr int kQ16Slots = 256;  // counters of the 16-bit integer column kernel's hand-over lists (one per launch)
constexpr int EDT_FLAG_NO_INDEX_FORM = 0x8000;  // internal: plan without the index buffer
static int64_t plan_code_slab(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (ndim < 2 || sx % 4 != 0 || sx * sy > kCodeSlabVoxels || (flags & (EDT_FLAG_NO_INDEX_FORM | EDT_FLAG_SMALL_WORKSPACE))) return 0;
  if ((flags & EDT_FLAG_FORCE_GENERIC) || env_force_generic() || (g_debug_mode & (0x100000 | 64 | 32))) return 0;
  if (!row_pass_wave_supported(dtype, sx, sy, sz) || !column_pass_wave_supported(make_geom_y(sx, sy, sz))) return 0;
  int64_t slab = kCodeSlabVoxels / (sx * sy);
  if (slab >= sz) return sz;
  // (several slabs: whole words of the per-slice map of the 16-bit plane per slab)
  if (slab >= 32) slab &= ~(int64_t)31;
  return slab;
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
    p.codes_whole = p.xy_slab >= sz || (p.xy_slab > 0 && p.xy_slab % 32 == 0 && ndim == 3 && !(flags & EDT_FLAG_BATCH_2D) &&
                                        sx * sy * sz * (int64_t)sizeof(uint16_t) <= whole_index_limit());
    // (a padded pitch only where the plane exists -- a whole-volume buffer of a 3-D call; a single slice needs none)
    p.code_pitch = sx * sy + ((p.codes_whole && ndim == 3 && sz > 1 && !(flags & EDT_FLAG_BATCH_2D)) ? plane_pad_elems(sx, sy) : 0);
    if (p.xy_slab > 0) p.codes = c.take<uint16_t>((size_t)((p.codes_whole ? sz : p.xy_slab) * p.code_pitch));
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
bool column_inplace_supported(const AxisGeom &g) {
  return column_pass_wave_supported(g) || column_pass_tiled_supported(g);
}
int launch_column_inplace(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                          float w, int bb, int epi, hipStream_t stream, const TileList &list) {
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
int launch_row_bits(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                           uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                           int to_finite, hipStream_t stream) {
  if (row_pass_wave_supported(dtype, sx, sy, sz) && !(g_debug_mode & 32))
    return launch_row_pass_wave(dtype, labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream);
  return launch_row_pass_tiled(dtype, labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream);
}

bool env_force_generic() {
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

int check_shape(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz) {
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

// Voxel sizes.  The reference does not validate them.  Along x a size enters pass 1 as itself (src/edt.hpp:86-113): a negative
// one makes the unguarded backward fminf sweep cross label boundaries, and NaN / inf / 0 give NaN or all-zero fields; no
// kernel here reproduces that, so such a call is refused instead of answered differently by different kernels.  Along y and
// z a size enters only as its square (w2 = anisotropy * anisotropy, src/edt.hpp:181, :258): a negative wy / wz gives the field
// of |w| there, and does here -- the sign is dropped before anything else looks at it.
static bool voxel_size_usable(float w) { return w > 0.0f && w <= FLT_MAX; }
int check_voxel_sizes(int naxes, float &wx, float &wy, float &wz) {
  float *w[3] = {&wx, &wy, &wz};
  for (int i = 0; i < naxes && i < 3; ++i) {
    if (i > 0) *w[i] = fabsf(*w[i]);
    if (!voxel_size_usable(*w[i])) {
      set_error(i == 0 ? "the voxel size along x must be positive and finite" : "voxel sizes (anisotropy) must be non-zero and finite");
      return EDT_ERR_BAD_ARG;
    }
  }
  return EDT_OK;
}
// the voxel size of a column pass on its own (the Z phase of the sharded path)
int check_column_voxel_size(float &w) {
  w = fabsf(w);
  if (!voxel_size_usable(w)) { set_error("voxel sizes (anisotropy) must be non-zero and finite"); return EDT_ERR_BAD_ARG; }
  return EDT_OK;
}

int require_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (this library has no CPU fallback)");
    return EDT_ERR_NO_DEVICE;
  }
  return EDT_OK;
}

// ---- the pass pipeline on device-resident data -----------------------------------------
int run_device(const void *d_labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz,
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
  // The SIGNED transform (EDT_FLAG_SIGNED; sdf / sdfsq of src/edt.pyx:121-202 = edt(x) - edt(x == 0)) as ONE transform: the
  // two fields have disjoint supports, and a voxel's value depends on nothing but the voxels of its own label runs -- so
  // measuring label 0 like every other label gives edt(x) on the foreground and edt(x == 0) on the background at once, bit
  // for bit; the background's sign follows in one streaming kernel.  Pass X's register kernel takes the flag (zero_label).
  const int signed_tf = (flags & EDT_FLAG_SIGNED) ? 1 : 0;
  if (signed_tf && !signed_transform_supported(dtype, ndim, sx, sy, sz, flags)) {
    set_error("EDT_FLAG_SIGNED: shape not served by the one-transform form (edt_hip_signed_supported)");
    return EDT_ERR_UNSUPPORTED;
  }
  const int last_epi = (bb ? 0 : kEpiToInf) | (want_sqrt ? kEpiSqrt : 0) | kEpiStream;
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
  // Where the integer kernel provably refuses no tile (index form, bounded values: q16_no_refusals, edt_colq16.hip) and
  // everything the pass reads was written by pass X or by an integer pass that could not refuse either, the fp32 launch over
  // the hand-over list has nothing to do and is not made -- nor is the list's counter zeroed.
  auto q16_cannot_refuse = [&](int axis, const AxisGeom &g) {
    return q16 && q16_no_refusals(q16_q, q16_a, axis, sx, sy, g.n, bb);
  };
  // F in place (codes == nullptr) or from the 16-bit indices of pass X; returns the list for the fp32 launch that follows
  // (launched: the integer kernel ran; sure: the caller vouches for what the pass reads -- see above)
  // will q16_pass launch the integer kernel for this axis?  (the shape and mode part of its test; buffers: column_pass_q16_aligned)
  auto q16_applies = [&](const AxisGeom &g) {
    return q16 && column_pass_q16_supported(g) && column_pass_wave_supported(g) && !(g_debug_mode & kQ16Off) &&
           ceil_div(g.sx, 16) * (ceil_div(g.nouter, 8) * 8) <= p.q16_id_capacity;
  };
  // map_words_off: the slab's first word in every x-tile's row of the plane's map (pass Y of the slabs after the first)
  auto q16_pass = [&](float *F, const uint16_t *codes, const uint32_t *rs, const AxisGeom &g, int axis, int epi,
                      TileList &list, uint16_t *plane, bool sure, bool &launched, int64_t map_words_off = 0,
                      const uint32_t *signbits = nullptr) -> int {
    list = TileList();
    launched = false;
    // (the bits that force one form of the fp32 kernel on every tile -- the test tiers' way to cover them -- keep the call there)
    if (!q16 || q16_slot >= kQ16Slots || !column_pass_q16_supported(g) || !column_pass_wave_supported(g) ||
        !column_pass_q16_aligned(F, codes, plane) || (g_debug_mode & kQ16Off))
      return EDT_OK;
    // (the id array was sized for these tile counts: make_plan)
    if (ceil_div(g.sx, 16) * (ceil_div(g.nouter, 8) * 8) > p.q16_id_capacity) return EDT_OK;
    sure = sure && q16_cannot_refuse(axis, g);
    if (!sure && !q16_counts_zeroed) {
      EDT_HIP_TRY(hipMemsetAsync(p.q16_counts, 0, kQ16Slots * sizeof(uint32_t), stream));
      q16_counts_zeroed = true;
    }
    uint32_t *count = p.q16_counts + q16_slot++;
    // (the index buffer's own pitch: the outer stride of codes -- and of the plane written over them -- in pass Y, the row stride
    // of the plane pass Z reads)
    const bool reads_plane = codes == nullptr && plane != nullptr;
    const int r = launch_column_pass_q16(F, codes, rs, g, q16_q, q16_a[axis], q16_a[0], bb, epi, count, p.q16_ids, stream,
                                         nullptr, plane, p.q16_map + map_words_off, p.q16_map_words, nullptr,
                                         reads_plane ? p.code_pitch : 0, reads_plane ? sx : 0,
                                         // (pass Y into the plane: may a tile of nothing but +inf stay there for pass Z?)
                                         (axis == 1 && plane != nullptr && codes != nullptr && q16_value_limit(q16_q, q16_a[2], sz, bb) != 0u) ? 1 : 0,
                                         signbits, (codes != nullptr && axis == 1) ? p.code_pitch : 0);
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
  bool plane16 = q16 && index_form && zpass && p.codes_whole && !(g_debug_mode & (kQ16Off | 0x10000000)) &&
                       column_pass_q16_supported(p.gy) && column_pass_q16_supported(p.gz) &&
                       column_pass_wave_supported(p.gy) && column_pass_wave_supported(p.gz);
  bool y_sure = true;  // every tile of pass Y was served by the integer kernel, provably (q16_cannot_refuse)
  // The signed transform's sign as the EPILOGUE of pass Z (round 6): where both column passes provably run on the integer kernel
  // alone -- it never reads the foreground plane -- pass X keeps the TRUE label != 0 bits there (zero_label = 2), the transposer
  // carries them to the z axis, and the last pass negates the voxels whose bit is clear (kEpiSign): no pass of its own over
  // labels and field (1.2 GB at 512^3).  Anywhere else: all-ones planes and k_negate_background.  (debug bit 0x400: never.)
  // q16_only: both column passes of the call provably run on the integer kernel alone (what q16_pass will decide, decided here
  // for the whole call; a pass that left that kernel after all is an internal error below)
  const bool q16_only = zpass && index_form && tiled_z && q16_applies(p.gy) && q16_applies(p.gz) && q16_cannot_refuse(1, p.gy) &&
                        q16_cannot_refuse(2, p.gz) && column_pass_q16_aligned(cur, p.codes, p.codes) &&
                        ceil_div(sz, p.xy_slab > 0 ? p.xy_slab : sz) + 1 <= kQ16Slots;
  const bool fuse_sign = signed_tf && q16_only && !(g_debug_mode & 0x400);
  const int zero_label = fuse_sign ? 2 : signed_tf;
  // ... and then nobody reads a foreground plane (the integer kernel knows background as N = 0): pass X does not write one, the
  // transposer carries the run starts alone -- 48 MiB less per 512^3 step (debug bit 0x400 keeps the planes)
  const bool skip_nz = q16_only && !signed_tf && !binary_yz && !(g_debug_mode & 0x400);
  uint32_t *const nzy = skip_nz ? nullptr : p.nz_y, *const nzz = skip_nz ? nullptr : p.nz_z;
  if (index_form) {
    const int64_t sxy = sx * sy, wpl = p.gy.sx * p.gy.nbands;  // voxels / bit words per slice
    const size_t lsz = dtype_size(dtype);
    const bool one = p.xy_slab >= sz;
    ScopedPass whole(one ? nullptr : "xy_pass", stream);
    for (int64_t z0 = 0; z0 < sz; z0 += p.xy_slab) {
      const int64_t zc = std::min<int64_t>(p.xy_slab, sz - z0);
      const char *lab = static_cast<const char *>(d_labels) + (size_t)(z0 * sxy) * lsz;
      // (one slab of indices taken in turn, or -- codes_whole -- every slab's own part of a whole-volume buffer: the 16-bit plane)
      uint16_t *slab_codes = p.codes + ((p.codes_whole && !one) ? z0 * p.code_pitch : 0);
      {
        // (slice 0 of a later slab compares against the slice below it through the halo pointer of the sharded path)
        ScopedPass t(one ? "x_pass" : nullptr, stream);
        rc = launch_row_pass_wave(dtype, lab, nullptr, nzy ? nzy + z0 * wpl : nullptr, p.rs_y + z0 * wpl,
                                  zpass ? p.zs_y + z0 * wpl : nullptr, sx, sy, zc, wx, bb, bb ? 0 : 1, stream,
                                  z0 > 0 ? lab - (size_t)sxy * lsz : nullptr, slab_codes, zero_label, p.code_pitch);
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
        rc = q16_pass(cur + z0 * sxy, slab_codes, p.rs_y + z0 * wpl, g, 1, zpass ? 0 : last_epi, list, plane16 ? slab_codes : nullptr,
                      true, launched, (p.codes_whole && !one) ? z0 / 32 : 0);
        if (rc != EDT_OK) return rc;
        if (!launched) plane16 = false;  // (nothing wrote the plane or said where the rows are: pass Z reads fp32 values)
        y_sure = y_sure && list.none;
        if (q16_only && (!launched || !list.none)) { set_error("internal: pass Y left the integer kernel"); return EDT_ERR_HIP; }
        if (!list.none)
          rc = launch_column_pass_wave_codes(cur + z0 * sxy, slab_codes, p.nz_y + z0 * wpl, p.rs_y + z0 * wpl, g, wy, bb,
                                             zpass ? 0 : last_epi, wx, bb ? 0 : 1, stream, nullptr, list, ColumnOut(), p.code_pitch);
        if (rc != EDT_OK) return rc;
      }
    }
  } else if (tiled_x) {
    // labels are read once: pass 1 also emits the run bit-planes of the y and z axes
    {
      ScopedPass t("x_pass", stream);
      if (signed_tf)  // (signed_transform_supported: the register kernel of pass X serves this shape)
        rc = launch_row_pass_wave(dtype, d_labels, cur, p.nz_y, p.rs_y, zpass ? p.zs_y : nullptr, sx, sy, sz, wx, bb, bb ? 0 : 1,
                                  stream, nullptr, nullptr, 1);
      else
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
    if (tiled_x) rc = launch_bits_transpose_yz(nzy, p.zs_y, nzz, p.rs_z, sx, sy, sz, stream);
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
      rc = q16_pass(cur, nullptr, p.rs_z, p.gz, 2, last_epi | (fuse_sign ? kEpiSign : 0), list, plane16 ? p.codes : nullptr,
                    index_form && y_sure, launched, 0, fuse_sign ? p.nz_z : nullptr);
      if (rc != EDT_OK) return rc;
      if (q16_only && (!launched || !list.none)) { set_error("internal: pass Z left the integer kernel"); return EDT_ERR_HIP; }
      if (!list.none) rc = launch_column_inplace(cur, p.nz_z, p.rs_z, p.gz, wz, bb, last_epi, stream, list);
    } else {
      rc = launch_column_pass_serial(cur, other, p.nz_z, p.rs_z, p.stack, p.gz, wz, bb, last_epi,
                                     stream);
      std::swap(cur, other);
    }
    if (rc != EDT_OK) return rc;
  }
  if (cur != d_out) { set_error("internal: result buffer mismatch"); return EDT_ERR_HIP; }
  if (signed_tf && !fuse_sign) {
    ScopedPass t("sign", stream);
    rc = launch_negate_background(dtype, d_labels, d_out, p.voxels, stream);
    if (rc != EDT_OK) return rc;
  }
  return EDT_OK;
}

// The one-transform form of sdf / sdfsq needs pass X on the register kernel (the only pass-X kernel that takes zero_label)
// and in-place column passes (their kernels never look at label values: run-start bits and field values only).
bool signed_transform_supported(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (ndim < 2 || ndim > 3 || dtype_size(dtype) == 0 || sx < 1 || sy < 1 || sz < 1) return false;
  if ((flags & (EDT_FLAG_FORCE_GENERIC | EDT_FLAG_BINARY_YZ)) || env_force_generic() || (g_debug_mode & 32)) return false;
  if (!row_pass_wave_supported(dtype, sx, sy, sz)) return false;
  if (!column_inplace_supported(make_geom_y(sx, sy, sz))) return false;
  const bool zpass = ndim == 3 && !(flags & EDT_FLAG_BATCH_2D);
  return !zpass || column_inplace_supported(make_geom_z(sx, sy, sz));
}

const char *last_error_cstr() { return g_last_error.c_str(); }

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

const char *edt_hip_last_error(void) { return last_error_cstr(); }

int edt_hip_index_form_exact(float wx, int64_t sx) { return (sx >= 1 && row_codes_exact(wx, sx)) ? 1 : 0; }

const char *edt_hip_version(void) { return "edt_hip 0.1 (gfx950)"; }

size_t edt_hip_workspace_bytes_flags(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (check_shape(dtype, ndim, sx, sy, sz) != EDT_OK) return 0;
  if (sx == 0 || sy == 0 || sz == 0) return 256;
  return make_plan(dtype, ndim, sx, sy, sz, nullptr, flags).bytes;
}

int edt_hip_signed_supported(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags) {
  if (check_shape(dtype, ndim, sx, sy, sz) != EDT_OK) return 0;
  return signed_transform_supported(dtype, ndim, sx, sy, sz, flags) ? 1 : 0;
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

int edt_hip_set_debug_mode(int mode) {
  set_thread_debug_mode(mode);
  return EDT_OK;
}

int edt_hip_get_debug_mode(void) { return debug_mode(); }

int edt_hip_q16_no_refusals(int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz, int ndim, int black_border,
                            int *pass_y, int *pass_z) {
  if (pass_y) *pass_y = 0;
  if (pass_z) *pass_z = 0;
  if (ndim < 2 || ndim > 3 || sx < 1 || sy < 1 || (ndim == 3 && sz < 1)) return 0;
  const float w3[3] = {wx, wy, wz};
  float q = 1.0f;
  uint32_t a[3] = {1u, 1u, 1u};
  if (!q16_quantum(w3, ndim, &q, a)) return 0;
  const bool y = q16_no_refusals(q, a, 1, sx, sy, sy, black_border);
  const bool z = ndim == 3 && y && q16_no_refusals(q, a, 2, sx, sy, sz, black_border);
  if (pass_y) *pass_y = y ? 1 : 0;
  if (pass_z) *pass_z = z ? 1 : 0;
  return 1;
}

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

// min(fl32(wx*wx), fl32(wy*wy)): what every non-zero value of a field is at least after passes X and Y (run_device has
// the argument); 0 where a voxel size is not a positive finite number
float edt_hip_field_floor(float wx, float wy) {
  const float a = wx * wx, b = wy * wy;
  if (!(a > 0.0f) || !(b > 0.0f) || !(a < INFINITY) || !(b < INFINITY)) return 0.0f;
  return a < b ? a : b;
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
