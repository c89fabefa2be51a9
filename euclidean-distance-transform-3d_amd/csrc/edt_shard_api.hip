// edt_shard_api.hip -- the C ABI (include/edt_hip.h), part 3 of 3: the two phases of the Z-sharded path as entry points
// on device-resident slabs (byte flags; slab records of fp32 rows; slab records of 16-bit rows).  The kernels are in
// edt_shard.hip and the column / row kernels; the driver above them is edt/distributed.py (one process per GPU) or
// edt_multi.hip (one process, several GPUs).
#include "edt_api_internal.h"

namespace edt_amd {

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
  { float w1 = 1.0f; if ((rc = check_voxel_sizes(2, wx, wy, w1)) != EDT_OK) return rc; }
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
  if ((rc = check_column_voxel_size(wz)) != EDT_OK) return rc;
  if (sx == 0 || sy_local == 0 || sz == 0) return EDT_OK;
  if (!d_partial || !d_zflags) { set_error("null device pointer"); return EDT_ERR_BAD_ARG; }
  ShardPlan p = make_shard_plan(sx, sy_local, sz, d_workspace);
  if (!d_workspace || workspace_bytes < p.bytes) {
    set_error("shard workspace too small: need " + std::to_string(p.bytes) + " bytes");
    return EDT_ERR_BAD_ARG;
  }
  const int bb = (flags & EDT_FLAG_BLACK_BORDER) ? 1 : 0;
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0) | kEpiStream;  // (the Z phase writes the call's results)
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

int edt_hip_shard_records_supported(int dtype, int64_t sx, int64_t sy, int64_t sz) {
  if (dtype_size(dtype) == 0 || sx < 1 || sy < 1 || sz < 1) return 0;
  if (debug_mode() & (32 | 64)) return 0;  // diagnostics: forced fallback kernels
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
  { float w1 = 1.0f; if ((rc = check_voxel_sizes(2, wx, wy, w1)) != EDT_OK) return rc; }
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
  const bool index_form = (sx % 4) == 0 && !(debug_mode() & 0x100000) && row_codes_exact(wx, sx);
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
    if (!(debug_mode() & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000)) && q16_quantum(w2, 2, &q, a) &&
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
  if ((rc = check_column_voxel_size(wz)) != EDT_OK) return rc;
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
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0) | kEpiStream;  // (the Z phase writes the call's results)
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
    if (!(debug_mode() & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000)) && q16_quantum(w3, 3, &q, a) &&
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
  if (sx % 4 != 0 || (debug_mode() & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x100000 | 0x8000000 | 0x10000000))) return false;
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
  { float w1 = 1.0f; if ((rc = check_voxel_sizes(2, wx, wy, w1)) != EDT_OK) return rc; }
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
  if ((rc = check_column_voxel_size(wz)) != EDT_OK) return rc;
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
  const int epi = (bb ? 0 : kEpiToInf) | ((flags & EDT_FLAG_SQRT) ? kEpiSqrt : 0) | kEpiStream;  // (the Z phase writes the call's results)
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

}  // extern "C"
