// edt_kernels.h -- device helpers shared by the kernels + launcher declarations (internal).
#pragma once

#include "edt_common.h"

#pragma clang fp contract(off)

namespace edt_amd {

// (double)d squared, exact for |d| < 2^26  (reference: pyedt::sq, src/edt.hpp:34-37)
__device__ __forceinline__ double sqd(int64_t d) {
  const double x = (double)d;
  return x * x;
}

// Numerator of the abscissa where parabola q (height Fq) overtakes parabola p < q:
//   s(p,q) = hull_num / (2 * (q - p) * w2)
// Same operation order as the reference's `ff[i] - ff[v[k]] + factor1 * factor2`
// (src/edt.hpp:206-208).  Comparing s(b,i) <= s(a,b) is done by cross-multiplication,
// which needs no division:   hull_num(b,i) * (b-a)  <=  hull_num(a,b) * (i-b).
__device__ __forceinline__ double hull_num(double Fp, double Fq, int64_t p, int64_t q, double w2) {
  const double f1 = (double)(q - p) * w2;
  const double f2 = (double)(q + p);
  return (Fq - Fp) + f1 * f2;
}

// Last-pass epilogue: FLT_MAX sentinel back to +INF (toinfinite, src/edt.hpp:47-53) and the
// optional correctly rounded sqrt (src/edt.hpp:599-601 / np.sqrt at src/edt.pyx:242).
__device__ __forceinline__ float finish(float m, int epi) {
  if ((epi & kEpiToInf) && m >= FLT_MAX) m = INFINITY;
  if (epi & kEpiSqrt) m = sqrtf(m);
  return m;
}

// ---- generic (any extent) kernels: edt_generic.hip ------------------------------------
int launch_row_pass_serial(int dtype, const void *labels, float *out, int64_t sx, int64_t nrows,
                           float w, int bb, int to_finite, int take_sqrt, hipStream_t stream);
// ---- the 1-D transform as a parallel pipeline for lines of any length: edt_line.hip -----------------
size_t line_workspace_bytes(int64_t n);
int launch_line_pass(int dtype, const void *labels, float *out, int64_t n, float w, int bb, int take_sqrt,
                     void *ws, hipStream_t stream);
// pass 1 over rows of any length (sx > 2048): the line pipeline with a run start forced at every row's first voxel
size_t rows_line_workspace_bytes(int64_t sx, int64_t nrows);
int launch_rows_line_pass(int dtype, const void *labels, float *out, int64_t sx, int64_t nrows, float w, int bb,
                          int to_finite, void *ws, hipStream_t stream);
size_t runs_workspace_bytes(int64_t n);
int launch_extract_runs(int dtype, const void *labels, int64_t n, int64_t *starts, int64_t capacity, int64_t *total,
                        void *ws, hipStream_t stream);
int launch_axis_bits(int dtype, const void *labels, const void *halo, uint32_t *nz, uint32_t *rs,
                     const AxisGeom &g, hipStream_t stream);
// planes of the binary route (EDT_FLAG_BINARY_YZ): every column one all-foreground run
int launch_planes_one_run(uint32_t *nz, uint32_t *rs, uint32_t *zs, const AxisGeom &g, int64_t o0, hipStream_t stream);
int launch_column_pass_serial(const float *fin, float *fout, const uint32_t *nz, const uint32_t *rs,
                              int32_t *stack, const AxisGeom &g, float w, int bb, int epi,
                              hipStream_t stream);
int launch_subtract(const float *a, const float *b, float *out, int64_t count, hipStream_t stream);
int launch_is_background(int dtype, const void *labels, uint8_t *mask, int64_t count,
                         hipStream_t stream);
int launch_negate_background(int dtype, const void *labels, float *f, int64_t count, hipStream_t stream);  // f = labels == 0 ? -f : f
int launch_select_label(int dtype, const void *labels, const float *dt, float *out, const void *key,
                        int64_t count, hipStream_t stream);

}  // namespace edt_amd

namespace edt_amd {
// ---- voxel-graph helpers: edt_voxel_graph.hip -----------------------------------------
int launch_vg_expand(int dtype, const void *labels, const uint8_t *graph, uint8_t *big, int64_t sx,
                     int64_t sy, int64_t sz, int ndim, int bb, hipStream_t stream);
int launch_vg_gather(const float *big, float *out, int64_t sx, int64_t sy, int64_t sz, int ndim,
                     hipStream_t stream);
// native form (no doubled volume): only the cells the result depends on are computed
bool vg_native_supported(int ndim, int64_t sx, int64_t sy, int64_t sz);
size_t vg_native_workspace_bytes(int ndim, int64_t sx, int64_t sy, int64_t sz);
int launch_vg_native(int dtype, const void *labels, const uint8_t *graph, int ndim, int64_t sx, int64_t sy,
                     int64_t sz, float wx, float wy, float wz, int bb, int want_sqrt, float *out, void *ws,
                     hipStream_t stream);
// ---- Z-sharded helpers: edt_shard.hip ---------------------------------------------------
int launch_zflags(int dtype, const void *labels, const void *halo, uint8_t *flags, int64_t sxy,
                  int64_t szl, hipStream_t stream);
int launch_bits_from_flags(const uint8_t *flags, uint32_t *nz, uint32_t *rs, const AxisGeom &g,
                           hipStream_t stream);
// y-packed planes [z][band][x] of a slab -> the bit part of the slab records; also publishes the
// destination map in device memory for the scattering column pass
int launch_pack_record_bits(const uint32_t *nz_y, const uint32_t *zs_y, const BandScatter &sc,
                            BandScatter *d_table, int64_t sx, int64_t nby, int64_t szl,
                            hipStream_t stream);
// ---- short axes (at most 32 rows): a thread per column, rows in registers: edt_short.hip ----
bool column_pass_short_supported(const AxisGeom &g);
int launch_column_pass_short(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb,
                             int epi, hipStream_t stream);
// ---- LDS-tiled column pass: edt_tiled.hip ---------------------------------------------------
bool column_pass_tiled_supported(const AxisGeom &g);
int launch_column_pass_tiled(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                             float w, int bb, int epi, hipStream_t stream);
// ---- wave-per-row-group pass 1 + bit-plane transposer: edt_rows.hip ---------------------------
bool row_pass_tiled_supported(int64_t sx);
int launch_row_pass_tiled(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                          uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                          int to_finite, hipStream_t stream);
// in_zstride: words between consecutive z of the y-packed planes (0 = dense, nby * sx)
int launch_bits_transpose_yz(const uint32_t *nz_y, const uint32_t *zs_y, uint32_t *nz_z,
                             uint32_t *rs_z, int64_t sx, int64_t sy, int64_t sz, hipStream_t stream,
                             int64_t in_zstride = 0);
}  // namespace edt_amd

namespace edt_amd {
// Arguments of the index form of pass 1 (XF kernels of edt_colwave_kernel.h): pass 1 stored 16-bit distance
// indices k (edt_rowwave.hip, C16) instead of F; the first column pass rebuilds F = fl32(fl32(k * w)^2) while it fills
// its tile.  codes = nullptr: the ordinary in-place pass.
struct XFuse {
  const uint16_t *codes;  // [outer][row][x], same strides (in elements) as F ...
  int64_t c_outer;        // ... but for the outer stride where the index buffer has a padded pitch (0: g.outer_stride)
  float w;                // voxel size of pass 1 (k * w exact: row_codes_exact)
  int flim;               // bit pattern of FLT_MAX (tofinite) or +inf
};
// ---- wave-autonomous LDS-tiled column pass: edt_colwave.hip -----------------------------------
bool column_pass_wave_supported(const AxisGeom &g);
// Which rows of a column the caller reads again, and where they go (the doubled grids of the voxel-graph transform).
struct ColumnOut {
  int stride = 1;            // 2: only the even rows of every column are needed (tiles on the windowed path evaluate
                             // just those; tiles on the hull path evaluate every row)
  float *compact = nullptr;  // stride 2 only: the even rows are written HERE instead of in place -- row r of the
                             // column (x, outer index o) goes to compact[x + o * outer + (r / 2) * row2]
  int64_t outer = 0, row2 = 0;
};
// The tiles a launch of the fp32 column kernel serves: every tile of its grid (count = nullptr), or -- list mode -- the
// tiles the 16-bit integer kernel (edt_colq16.hip) handed over: workgroup b takes ids[b] if b < *count (device memory).
struct TileList {
  const uint32_t *count = nullptr;
  const uint32_t *ids = nullptr;
  bool none = false;  // the integer kernel served EVERY tile and the host can prove it (edt_api.hip): no fp32 launch at all
};
// scatter != nullptr (device table): the rows are written to the slab records instead of F
// out: see ColumnOut (default: every row, in place) -- (tiles on the windowed path evaluate and write
// just those; tiles on the hull path still write every row)
int launch_column_pass_wave(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                            float w, int bb, int epi, hipStream_t stream,
                            const BandScatter *scatter = nullptr, const ColumnOut &out = ColumnOut(),
                            const TileList &list = TileList());
// the same reading pass 1 as 16-bit distance indices (F is write-only): see XFuse
int launch_column_pass_wave_codes(float *F, const uint16_t *codes, const uint32_t *nz, const uint32_t *rs,
                                  const AxisGeom &g, float w, int bb, int epi, float wx, int to_finite,
                                  hipStream_t stream, const BandScatter *scatter = nullptr, const TileList &list = TileList(),
                                  const ColumnOut &out = ColumnOut(), int64_t codes_outer = 0);
// ---- 16-bit integer column pass: edt_colq16.hip ---------------------------------------------------
// the quantum of a call: w_i^2 = a[i] * q (false: the voxel sizes share none, the fp32 kernels keep the call)
bool q16_quantum(const float *w, int naxes, float *q, uint32_t *a);
bool column_pass_q16_supported(const AxisGeom &g);
// the largest value, in quanta, a tile of a pass with c_d = a * d^2 may hold and stay on the integer kernel (its 16-bit
// form, or the wide form: two half-tiles with 32-bit lanes) -- what a host that knows a bound of the field compares with
uint32_t q16_value_limit(float q, uint32_t a, int64_t n, int bb);
// the host's proof that the integer kernel refuses NO tile of a column pass of a call in the index form (axis 1: pass Y over
// columns of n = sy rows; axis 2: pass Z over n = sz rows behind a pass Y that could not refuse either) -- edt_colq16.hip
bool q16_no_refusals(float q, const uint32_t *a, int axis, int64_t sx, int64_t sy, int64_t n, int bb);
// the kernel's vector accesses: 16-byte loads of fp32 rows, 8-byte stores of result pairs, 8-byte loads of index / plane
// rows (a 4-byte-aligned view handed in through DLPack stays on the fp32 kernel, which gates its vector accesses itself)
inline bool column_pass_q16_aligned(const float *F, const uint16_t *codes, const uint16_t *plane, const float *compact = nullptr) {
  return (reinterpret_cast<uintptr_t>(F) % 16) == 0 && (reinterpret_cast<uintptr_t>(codes) % 8) == 0 &&
         (reinterpret_cast<uintptr_t>(plane) % 8) == 0 && (reinterpret_cast<uintptr_t>(compact) % 8) == 0;
}
// codes != nullptr: pass X in index form (N = k^2 * ain), F is only written; else F is read (N = F / q, verified) and
// written in place.  a: c_d = a * d^2 quanta.  Tiles that do not qualify are appended to (count, ids) for the fp32 kernel.
// plane / map (volumes whose indices fit one slab): with codes, the results stay 16-bit -- written over the indices
// (plane == codes), the tile's bit set in map [x-tile][map_words]; without codes, every row is read from the plane where
// map says so and from F elsewhere (the pass after such a pass).
int launch_column_pass_q16(float *F, const uint16_t *codes, const uint32_t *rs, const AxisGeom &g, float q, uint32_t a,
                           uint32_t ain, int bb, int epi, uint32_t *count, uint32_t *ids, hipStream_t stream,
                           const BandScatter *scatter = nullptr, uint16_t *plane = nullptr, uint32_t *map = nullptr,
                           int map_words = 0, const ColumnOut *out = nullptr, int64_t plane_stride = 0, int64_t plane_outer = 0,
                           int plane_inf_ok = 0,   // the pass that reads the 16-bit plane this one writes carries +inf
                           const uint32_t *signbits = nullptr,  // kEpiSign: the true foreground plane of this axis
                           int64_t codes_outer = 0);  // outer stride of codes (and of the plane written over them), 0: g.outer_stride
}  // namespace edt_amd

namespace edt_amd {
// ---- register-resident pass 1 (rows up to 512 voxels): edt_rowwave.hip -------------------------
bool row_pass_wave_supported(int dtype, int64_t sx, int64_t sy, int64_t sz);
// halo: the xy-slice below slice 0 (Z-sharded slabs), or nullptr = slice 0 starts every z-run
// codes != nullptr: 16-bit distance indices are written there INSTEAD of `out` (see XFuse)
int launch_row_pass_wave(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                         uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                         int to_finite, hipStream_t stream, const void *halo = nullptr, uint16_t *codes = nullptr,
                         int zero_label = 0,  // zero_label: label 0 is measured like every label (the signed transform)
                         int64_t codes_pitch = 0);  // elements between the slices of codes (0: sx * sy)
// k * w exact for every k of a row of sx voxels: the 16-bit index form (codes) is bit-identical
bool row_codes_exact(float w, int64_t sx);

}  // namespace edt_amd

namespace edt_amd {
// ---- one process, several GPUs (host buffers): edt_multi.hip ----------------------------------------
bool multi_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, int n_devices);
void multi_release();  // frees the per-slot pool (edt_hip_release_cache)
int run_multi(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
              int flags, float *output, const int *devices, int n_devices);
}  // namespace edt_amd
