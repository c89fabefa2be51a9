// edt_voxel_graph.hip -- voxel-connectivity-graph EDT (reference: _edt2dsq_voxel_graph /
// _edt3dsq_voxel_graph, src/edt_voxel_graph.hpp:54-117, :120-214).
//
// Formulation (same as the reference so results are bit-identical): binarise the labels,
// up-sample 2x per axis into a uint8 volume in which a forbidden +x/+y/+z step (graph bit
// 0x01 / 0x04 / 0x10 clear) becomes a background half-voxel, transform that volume at half
// the voxel size with the ordinary pipeline, keep every other sample.  Expand and gather
// are two streaming kernels around edt_hip_edtsq_device.
#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

// One thread per (original x, up-sampled y, up-sampled z): writes the two bytes
// (2x, Y, Z), (2x+1, Y, Z) -- coalesced 128 B per wavefront.
template <typename T>
__global__ void k_vg_expand(const T *__restrict__ labels, const uint8_t *__restrict__ graph,
                            uint8_t *__restrict__ big, int64_t sx, int64_t sy, int64_t sz, int ndim,
                            int bb) {
  const int64_t Y = 2 * sy, Z = (ndim == 3) ? 2 * sz : 1;
  const int64_t total = sx * Y * Z;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += step) {
    const int64_t x = idx % sx;
    const int64_t yy = (idx / sx) % Y;
    const int64_t zz = idx / (sx * Y);
    const int64_t y = yy >> 1, z = zz >> 1;
    const int dy = (int)(yy & 1), dz = (int)(zz & 1);
    const int64_t src = x + sx * (y + sy * z);
    const bool fg = labels[src] > 0;  // `labels[loc] > 0` (src/edt_voxel_graph.hpp:76, :143)
    const uint8_t g = graph[src];
    // dx = 0 cell
    bool v0 = fg;
    if (dy == 1 && dz == 0) v0 = fg && (g & 0x04);
    if (dy == 0 && dz == 1) v0 = fg && (g & 0x10);
    // dx = 1 cell: only the pure +x half-step consults the graph
    bool v1 = fg;
    if (dy == 0 && dz == 0) v1 = fg && (g & 0x01);
    if (bb) {  // black border trims the outermost up-sampled faces (:78-90, :156-187)
      if (x == sx - 1) v1 = false;
      if (dy == 1 && y == sy - 1) { v0 = false; v1 = false; }
      if (dz == 1 && z == sz - 1) { v0 = false; v1 = false; }
    }
    uchar2 o;
    o.x = v0 ? 1 : 0;
    o.y = v1 ? 1 : 0;
    reinterpret_cast<uchar2 *>(big)[x + sx * (yy + Y * zz)] = o;
  }
}

__global__ void k_vg_gather(const float *__restrict__ big, float *__restrict__ out, int64_t sx,
                            int64_t sy, int64_t sz, int ndim) {
  const int64_t X = 2 * sx, Y = 2 * sy;
  const int64_t total = sx * sy * sz;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += step) {
    const int64_t x = idx % sx;
    const int64_t y = (idx / sx) % sy;
    const int64_t z = idx / (sx * sy);
    const int64_t zz = (ndim == 3) ? 2 * z : 0;
    out[idx] = big[2 * x + X * (2 * y + Y * zz)];
  }
}

int launch_vg_expand(int dtype, const void *labels, const uint8_t *graph, uint8_t *big, int64_t sx,
                     int64_t sy, int64_t sz, int ndim, int bb, hipStream_t stream) {
  const int threads = 256;
  const int64_t total = sx * 2 * sy * (ndim == 3 ? 2 * sz : 1);
  int64_t blocks = ceil_div(total, threads);
  if (blocks > 16384) blocks = 16384;
#define LAUNCH_VG(T)                                                                            \
  hipLaunchKernelGGL(k_vg_expand<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,          \
                     (const T *)labels, graph, big, sx, sy, sz, ndim, bb)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_VG(uint8_t); break;
    case EDT_U16: LAUNCH_VG(uint16_t); break;
    case EDT_U32: LAUNCH_VG(uint32_t); break;
    case EDT_U64: LAUNCH_VG(uint64_t); break;
    case EDT_F32: LAUNCH_VG(float); break;
    case EDT_F64: LAUNCH_VG(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_VG
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_vg_gather(const float *big, float *out, int64_t sx, int64_t sy, int64_t sz, int ndim,
                     hipStream_t stream) {
  const int threads = 256;
  int64_t blocks = ceil_div(sx * sy * sz, threads);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_vg_gather, dim3((unsigned)blocks), dim3(threads), 0, stream, big, out, sx, sy,
                     sz, ndim);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}


// =============================================================================================
// Native form: the doubled grid is never materialised.
//
// The result is read at the even cells (2x, 2y, 2z) only (src/edt_voxel_graph.hpp:201-210), and the
// transform is separable, so of the doubled grid's 8 * voxels cells
//   pass X has to produce the even-X cells of every doubled row        (4 * voxels values),
//   pass Y runs on the even-X columns only (all Y, all Z; in place),
//   pass Z runs on the even-X, even-Y columns only                     (2 * voxels values),
// and every value that IS computed goes through exactly the reference's arithmetic in the reference's
// order (binary labels on the doubled grid at half the voxel size: fp32 sequential sums and square
// in pass X, the fp64 envelope with fp32 rounding after each of passes Y and Z) -- the results are
// bit-identical to the up-sampled formulation above, which stays as the fallback for axes the wave
// column kernel does not cover.  The doubled rows are described by two bit rows ("even cell is
// background", "odd cell is background") built from the label and graph bytes on the fly:
//   even cell (2x, Y, Z):  fg                       for (Y, Z) even/even and odd/odd
//                          fg && graph & 0x04 (+y)  for Y odd, Z even;   fg && graph & 0x10 (+z)  for Y even, Z odd
//   odd cell (2x+1, Y, Z): fg && graph & 0x01 (+x)  for Y, Z even;       fg otherwise
//   black_border: cell 2sx-1 of every row, row 2sy-1 of every slice and slice 2sz-1 are background
//   (src/edt_voxel_graph.hpp:145-187).
// Traffic per voxel: 16 B (pass X) + 32 B (pass Y) + 16 B (pass Z) + 8 B (gather) against ~200 B of the
// up-sampled formulation; scratch 4 * voxels floats + bit planes instead of 8 * voxels bytes + 8 * voxels floats.
// =============================================================================================
namespace {

template <typename T>
__device__ __forceinline__ bool vg_fg(T v) { return v > 0; }  // `labels[loc] > 0`

// background-ness of the even / odd cell of voxel x in the doubled row (yp, zp = parities of Y, Z)
__device__ __forceinline__ bool vg_even_cell(bool fg, uint32_t g, int yp, int zp) {
  if (yp == 1 && zp == 0) return fg && (g & 0x04u);
  if (yp == 0 && zp == 1) return fg && (g & 0x10u);
  return fg;
}
__device__ __forceinline__ bool vg_odd_cell(bool fg, uint32_t g, int yp, int zp) {
  if (yp == 0 && zp == 0) return fg && (g & 0x01u);
  return fg;
}

constexpr int kVgFar = 1 << 28;  // "no background cell / border on this side"

// T[0] = 0, T[k] = fl32(T[k-1] + w): the sequential fp32 sums pass 1 accumulates (src/edt.hpp:92-114);
// T[count] = +inf.
__global__ void k_vg_ttab(float *__restrict__ ttab, float w, int count) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float acc = 0.0f;
  ttab[0] = 0.0f;
  for (int k = 1; k < count; ++k) {
    acc = acc + w;
    ttab[k] = acc;
  }
  ttab[count] = INFINITY;
}

// Pass X of the doubled grid, even cells only: one wave per VOXEL row (y, z) produces its doubled rows
// (Y, Z) = (2y + yp, 2z + zp) -- four in 3-D, two in 2-D -- from ONE read of the row's label and graph bytes;
// the four independent computations also give the scalar / shuffle chains something to overlap with.
template <typename T>
__global__ void __launch_bounds__(256)
k_vg_rows(const T *__restrict__ labels, const uint8_t *__restrict__ graph, const float *__restrict__ ttab,
          float *__restrict__ F1, int sx, int sy, int sz, int Y2, int Z2, int bb, int idx_inf) {
  extern __shared__ float Tl[];
  for (int i = (int)threadIdx.x; i <= idx_inf; i += (int)blockDim.x) Tl[i] = ttab[i];
  __syncthreads();
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int NC = (sx + 63) >> 6;  // <= 32 (launcher)
  const int NQ = Z2 > 1 ? 4 : 2;  // doubled rows per voxel row
  const int64_t nvrows = (int64_t)sy * (Z2 > 1 ? sz : 1);
  const uint64_t below = (1ull << lane) - 1ull;                         // lanes before this one
  const uint64_t above = lane < 63 ? ~((2ull << lane) - 1ull) : 0ull;   // lanes after it
  for (int64_t vrow = (int64_t)blockIdx.x * 4 + wave; vrow < nvrows; vrow += (int64_t)gridDim.x * 4) {
    const int y = (int)(vrow % sy), z = (int)(vrow / sy);
    const T *lrow = labels + vrow * sx;
    const uint8_t *grow = graph + vrow * sx;
    // a doubled row trimmed by black_border is all background (src/edt_voxel_graph.hpp:156-187)
    bool dead[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dead[q] = bb && (((q & 1) && y == sy - 1) || ((q >> 1) && Z2 > 1 && z == sz - 1));
    // ---- phase A: the cells of every 64-voxel chunk; last / first background cell per chunk and row ----
    uint32_t flagsE[4] = {0, 0, 0, 0}, flagsO[4] = {0, 0, 0, 0};  // bit c: the lane's even / odd cell is FOREGROUND
    int lastz[4], firstz[4];  // lane c: last / first background cell (doubled coordinate) of chunk c
#pragma unroll
    for (int q = 0; q < 4; ++q) { lastz[q] = -kVgFar; firstz[q] = kVgFar; }
    for (int c = 0; c < NC; ++c) {
      const int x = c * 64 + lane;
      const bool valid = x < sx;
      bool fg = false;
      uint32_t g = 0;
      if (valid) {
        fg = vg_fg(lrow[x]);
        g = grow[x];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= NQ) break;
        const bool E = !dead[q] && vg_even_cell(fg, g, q & 1, q >> 1);
        bool O = !dead[q] && vg_odd_cell(fg, g, q & 1, q >> 1);
        if (bb && x == sx - 1) O = false;
        flagsE[q] |= (E ? 1u : 0u) << c;
        flagsO[q] |= (O ? 1u : 0u) << c;
        const uint64_t zE = __ballot(valid && !E), zO = __ballot(valid && !O);
        int lz = -kVgFar, fz = kVgFar;
        if (zE) {
          lz = 2 * (c * 64 + 63 - __builtin_clzll(zE));
          fz = 2 * (c * 64 + __builtin_ctzll(zE));
        }
        if (zO) {
          const int l2 = 2 * (c * 64 + 63 - __builtin_clzll(zO)) + 1, f2 = 2 * (c * 64 + __builtin_ctzll(zO)) + 1;
          lz = l2 > lz ? l2 : lz;
          fz = f2 < fz ? f2 : fz;
        }
        if (lane == c) { lastz[q] = lz; firstz[q] = fz; }
      }
    }
    // exclusive prefix max of lastz / exclusive suffix min of firstz over the chunks (lanes), seeded with the
    // border sites just outside the row (black_border) or "none"
    int prev[4], next[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      prev[q] = __shfl_up(lastz[q], 1);
      next[q] = __shfl_down(firstz[q], 1);
      if (lane == 0) prev[q] = bb ? -1 : -kVgFar;
      if (lane >= NC - 1) next[q] = bb ? 2 * sx : kVgFar;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tp = __shfl_up(prev[q], d), tn = __shfl_down(next[q], d);
        if (lane >= d) prev[q] = tp > prev[q] ? tp : prev[q];
        if (lane + d < 64) next[q] = tn < next[q] ? tn : next[q];
      }
    }
    // ---- phase B: distances to the nearest background cell on either side, table look-up, square ----
    for (int c = 0; c < NC; ++c) {
      const int x = c * 64 + lane;
      const bool valid = x < sx;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= NQ) break;
        const bool E = (flagsE[q] >> c) & 1u, O = (flagsO[q] >> c) & 1u;
        const uint64_t zE = __ballot(valid && !E), zO = __ballot(valid && !O);
        int Xl = __shfl(prev[q], c), Xr = __shfl(next[q], c);
        const uint64_t mE = zE & below, mO = zO & below;
        if (mE) { const int p = 2 * (c * 64 + 63 - __builtin_clzll(mE)); Xl = p > Xl ? p : Xl; }
        if (mO) { const int p = 2 * (c * 64 + 63 - __builtin_clzll(mO)) + 1; Xl = p > Xl ? p : Xl; }
        const uint64_t nE = zE & above, nO = zO & (above | (1ull << lane));  // the own odd cell lies to the right
        if (nE) { const int p = 2 * (c * 64 + __builtin_ctzll(nE)); Xr = p < Xr ? p : Xr; }
        if (nO) { const int p = 2 * (c * 64 + __builtin_ctzll(nO)) + 1; Xr = p < Xr ? p : Xr; }
        const int X = 2 * x;
        int il = X - Xl, ir = Xr - X;
        il = il < idx_inf ? il : idx_inf;
        ir = ir < idx_inf ? ir : idx_inf;
        const float tl = Tl[il], tr = Tl[ir];
        const float d = tl < tr ? tl : tr;
        float f = d * d;                                    // `d[i] *= d[i]` (src/edt.hpp:116-118)
        if (!bb && f >= INFINITY) f = 3.402823466e+38f;     // tofinite (src/edt.hpp:39-45)
        if (!E) f = 0.0f;
        if (valid) F1[((int64_t)(2 * z + (q >> 1)) * Y2 + (2 * y + (q & 1))) * sx + x] = f;
      }
    }
  }
}

// Bit planes of the column passes for the even-X columns.  Along Y: words [Z][Y/32][x] over the doubled Y
// axis of slice Z; along Z: words [y][Z/32][x] over the doubled Z axis of the even row 2y.  nz = the cell is
// foreground, rs = it differs from the cell below it (row 0: always set), as edt_rowwave.hip defines them.
template <typename T, bool ALONG_Z>
__global__ void __launch_bounds__(256)
k_vg_bits(const T *__restrict__ labels, const uint8_t *__restrict__ graph, uint32_t *__restrict__ nz,
          uint32_t *__restrict__ rs, int sx, int sy, int sz, int Y2, int Z2, int nwords, int bb) {
  // ALONG_Z: outer = y (sy of them), axis = Z2;  else: outer = Z (Z2 of them), axis = Y2.
  // A word covers 32 doubled rows = 16 voxels along the axis, each read once (two cells per voxel).
  const int64_t nouter = ALONG_Z ? sy : Z2;
  const int64_t total = (int64_t)sx * nwords * nouter;
  const int n2 = ALONG_Z ? Z2 : Y2, nvox = ALONG_Z ? sz : sy;
  const int64_t vstride = ALONG_Z ? (int64_t)sx * sy : sx;  // between consecutive voxels along the axis
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % sx);
    const int wd = (int)((idx / sx) % nwords);
    const int o = (int)(idx / ((int64_t)sx * nwords));
    // the fixed coordinates: ALONG_Z -> the even row 2*o (yp = 0); else slice Z = o (zp = o & 1, z = o >> 1)
    const int zfix = ALONG_Z ? 0 : (o >> 1), zp = ALONG_Z ? 0 : (o & 1);
    if (!ALONG_Z && bb && zp && Z2 > 1 && zfix == sz - 1) {  // trimmed slice: all background
      nz[idx] = 0u;
      rs[idx] = wd == 0 ? 1u : 0u;
      continue;
    }
    const int64_t base = ALONG_Z ? x + (int64_t)sx * o : x + (int64_t)sx * sy * zfix;
    // the two cells of voxel v along the axis: even position 2v, odd position 2v+1
    auto cells = [&](int v, bool &c0, bool &c1) {
      const int64_t src = base + vstride * v;
      const bool fg = vg_fg(labels[src]);
      const uint32_t g = graph[src];
      if (ALONG_Z) {  // (Y even) Z = 2v: fg; Z = 2v+1: fg && +z edge
        c0 = fg;
        c1 = fg && (g & 0x10u);
        if (bb && v == sz - 1) c1 = false;
      } else {        // Y = 2v: even cell of (yp = 0, zp); Y = 2v+1: (yp = 1, zp)
        c0 = vg_even_cell(fg, g, 0, zp);
        c1 = vg_even_cell(fg, g, 1, zp);
        if (bb && v == sy - 1) c1 = false;
      }
    };
    uint32_t w = 0;
    const int v0 = wd * 16;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      if (v0 + k >= nvox) break;
      bool c0, c1;
      cells(v0 + k, c0, c1);
      w |= (c0 ? 1u : 0u) << (2 * k) | (c1 ? 1u : 0u) << (2 * k + 1);
    }
    uint32_t carry = 0;
    if (wd > 0) {
      bool c0, c1;
      cells(v0 - 1, c0, c1);
      carry = c1 ? 1u : 0u;
    }
    uint32_t rsw = w ^ ((w << 1) | carry);
    if (wd == 0) rsw |= 1u;
    const int valid = n2 - wd * 32;  // rows of this word that exist
    if (valid < 32) rsw &= (1u << valid) - 1u;
    nz[idx] = w;
    rs[idx] = rsw;
  }
}

__global__ void k_vg_gather_even(const float *__restrict__ F1, float *__restrict__ out, int64_t sx, int64_t sy,
                                 int64_t sz, int64_t Y2, int ndim) {
  const int64_t total = sx * sy * sz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = idx % sx, y = (idx / sx) % sy, z = idx / (sx * sy);
    out[idx] = F1[x + sx * (2 * y + Y2 * (ndim == 3 ? 2 * z : 0))];
  }
}

}  // namespace

bool vg_native_supported(int ndim, int64_t sx, int64_t sy, int64_t sz) {
  if (ndim < 2 || sx > 2048) return false;
  AxisGeom gy;
  gy.sx = sx; gy.n = 2 * sy; gy.stride = sx; gy.nouter = 1; gy.outer_stride = 0; gy.nbands = ceil_div(2 * sy, kBandRows);
  if (!column_pass_wave_supported(gy)) return false;
  if (ndim == 3) {
    gy.n = 2 * sz;
    gy.nbands = ceil_div(2 * sz, kBandRows);
    if (!column_pass_wave_supported(gy)) return false;
  }
  return true;
}

size_t vg_native_workspace_bytes(int ndim, int64_t sx, int64_t sy, int64_t sz) {
  const int64_t Y2 = 2 * sy, Z2 = ndim == 3 ? 2 * sz : 1;
  size_t b = align_up((size_t)(sx * Y2 * Z2) * sizeof(float), 256);
  b += 2 * align_up((size_t)(sx * ceil_div(Y2, kBandRows) * Z2) * sizeof(uint32_t), 256);
  if (ndim == 3) b += 2 * align_up((size_t)(sx * ceil_div(Z2, kBandRows) * sy) * sizeof(uint32_t), 256);
  b += align_up((size_t)(2 * sx + 4) * sizeof(float), 256);
  return b + 256;
}

template <typename T>
static int vg_native_t(const void *labels_, const uint8_t *graph, int ndim, int64_t sx, int64_t sy, int64_t sz,
                       float wx, float wy, float wz, int bb, int want_sqrt, float *out, void *ws, hipStream_t stream) {
  const T *labels = static_cast<const T *>(labels_);
  const int64_t Y2 = 2 * sy, Z2 = ndim == 3 ? 2 * sz : 1;
  const int64_t nbY = ceil_div(Y2, kBandRows), nbZ = ceil_div(Z2, kBandRows);
  char *p = static_cast<char *>(ws);
  auto take = [&](size_t bytes) { char *q = p; p += align_up(bytes, 256); return q; };
  float *F1 = reinterpret_cast<float *>(take((size_t)(sx * Y2 * Z2) * sizeof(float)));
  uint32_t *nzY = reinterpret_cast<uint32_t *>(take((size_t)(sx * nbY * Z2) * sizeof(uint32_t)));
  uint32_t *rsY = reinterpret_cast<uint32_t *>(take((size_t)(sx * nbY * Z2) * sizeof(uint32_t)));
  uint32_t *nzZ = nullptr, *rsZ = nullptr;
  if (ndim == 3) {
    nzZ = reinterpret_cast<uint32_t *>(take((size_t)(sx * nbZ * sy) * sizeof(uint32_t)));
    rsZ = reinterpret_cast<uint32_t *>(take((size_t)(sx * nbZ * sy) * sizeof(uint32_t)));
  }
  float *ttab = reinterpret_cast<float *>(take((size_t)(2 * sx + 4) * sizeof(float)));
  const int idx_inf = (int)(2 * sx + 2);
  // half voxel size on the doubled grid (src/edt_voxel_graph.hpp:96-101, :189-193)
  const float hx = wx / 2, hy = wy / 2, hz = wz / 2;
  hipLaunchKernelGGL(k_vg_ttab, dim3(1), dim3(64), 0, stream, ttab, hx, idx_inf);
  {
    const int64_t nvrows = sy * (ndim == 3 ? sz : 1);  // one wave per voxel row (its 2 or 4 doubled rows)
    int64_t blocks = ceil_div(nvrows, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    static std::atomic<uint64_t> attr_done{0};
    EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_vg_rows<T>)));
    hipLaunchKernelGGL(k_vg_rows<T>, dim3((unsigned)blocks), dim3(256), (size_t)(idx_inf + 1) * sizeof(float), stream,
                       labels, graph, ttab, F1, (int)sx, (int)sy, (int)sz, (int)Y2, (int)Z2, bb, idx_inf);
  }
  {
    const int64_t total = sx * nbY * Z2;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((k_vg_bits<T, false>), dim3((unsigned)blocks), dim3(256), 0, stream, labels, graph, nzY, rsY,
                       (int)sx, (int)sy, (int)sz, (int)Y2, (int)Z2, (int)nbY, bb);
  }
  EDT_HIP_TRY(hipGetLastError());
  const int last_epi = (bb ? 0 : kEpiToInf) | (want_sqrt ? kEpiSqrt : 0);
  AxisGeom gy;
  gy.sx = sx; gy.n = Y2; gy.stride = sx; gy.nouter = Z2; gy.outer_stride = sx * Y2; gy.nbands = nbY;
  // (only the even rows of the doubled columns are read again: by the z pass / the gather)
  int rc = launch_column_pass_wave(F1, nzY, rsY, gy, hy, bb, ndim == 2 ? last_epi : 0, stream, nullptr, 2);
  if (rc != EDT_OK) return rc;
  if (ndim == 3) {
    const int64_t total = sx * nbZ * sy;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((k_vg_bits<T, true>), dim3((unsigned)blocks), dim3(256), 0, stream, labels, graph, nzZ, rsZ,
                       (int)sx, (int)sy, (int)sz, (int)Y2, (int)Z2, (int)nbZ, bb);
    EDT_HIP_TRY(hipGetLastError());
    AxisGeom gz;  // the even rows only: outer index = y, two doubled rows apart
    gz.sx = sx; gz.n = Z2; gz.stride = sx * Y2; gz.nouter = sy; gz.outer_stride = 2 * sx; gz.nbands = nbZ;
    rc = launch_column_pass_wave(F1, nzZ, rsZ, gz, hz, bb, last_epi, stream, nullptr, 2);
    if (rc != EDT_OK) return rc;
  }
  int64_t blocks = ceil_div(sx * sy * sz, 256);
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_vg_gather_even, dim3((unsigned)blocks), dim3(256), 0, stream, F1, out, sx, sy,
                     ndim == 3 ? sz : 1, Y2, ndim);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_vg_native(int dtype, const void *labels, const uint8_t *graph, int ndim, int64_t sx, int64_t sy, int64_t sz,
                     float wx, float wy, float wz, int bb, int want_sqrt, float *out, void *ws, hipStream_t stream) {
#define VG_NATIVE(T) \
  return vg_native_t<T>(labels, graph, ndim, sx, sy, sz, wx, wy, wz, bb, want_sqrt, out, ws, stream)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: VG_NATIVE(uint8_t);
    case EDT_U16: VG_NATIVE(uint16_t);
    case EDT_U32: VG_NATIVE(uint32_t);
    case EDT_U64: VG_NATIVE(uint64_t);
    case EDT_F32: VG_NATIVE(float);
    case EDT_F64: VG_NATIVE(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef VG_NATIVE
}

}  // namespace edt_amd
