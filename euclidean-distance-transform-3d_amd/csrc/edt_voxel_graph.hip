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
// (Y, Z) = (2y + yp, 2z + zp) -- four in 3-D, two in 2-D -- from ONE read of the row's label and graph bytes.
//
// The four doubled rows differ only in WHICH half-cells the graph turns into background, so the row is described
// by four voxel masks (bit = "this voxel contributes a background cell"):
//   F: the voxel is background (both its cells, in every doubled row)
//   X: F or its +x link is cut  -> the ODD  cell 2x+1 of row (yp, zp) = (0, 0)
//   Y: F or its +y link is cut  -> the EVEN cell 2x   of row (1, 0)
//   Z: F or its +z link is cut  -> the EVEN cell 2x   of row (0, 1)
// and per row: (0,0): even cells F, odd cells X;  (1,0): even Y, odd F;  (0,1): even Z, odd F;  (1,1): even F, odd F.
// Seen from the even cell 2x an even-type cell of voxel v is 2|x - v| away, an odd-type one 2(x - v) - 1 to the left
// (v < x) and 2(v - x) + 1 to the right (v >= x: the voxel's own odd cell lies to its right).  So a lane needs, per
// mask, the nearest member below and above its voxel: two bit scans of the chunk's ballot mask, with the carry
// from the other chunks kept on the scalar unit -- the forward sweep (phase A) leaves the last member before each
// chunk in lane c, the backward sweep (phase B) carries the first member after it along.  Real voxel graphs cut few
// links: a doubled row whose mask equals F in EVERY chunk (wave-uniform test) is the (1,1) row and is not computed
// again.  black_border: the border site left of the row is cell -1, the trimmed odd cell 2sx-1 is the nearest
// background on the right of every row (src/edt_voxel_graph.hpp:156-187), rows 2sy-1 / slices 2sz-1 are background.
// C16: the index form (edt_colwave_lane.h: code_value) -- where every multiple of the half voxel size is exact the rows leave as
// 16-bit distance indices (in half cells; 0 = background, 0xFFFF = no boundary) instead of fp32 values: half the bytes out of
// this kernel and into the first column pass.
template <typename T, bool C16>
__global__ void __launch_bounds__(256)
k_vg_rows(const T *__restrict__ labels, const uint8_t *__restrict__ graph, const float *__restrict__ ttab,
          float *__restrict__ F1, uint16_t *__restrict__ codes, int sx, int sy, int sz, int Y2, int Z2, int bb, int idx_inf,
          float exact_w) {
  extern __shared__ float Tl[];
  // (exact_w > 0: every multiple of the half voxel size is exact in fp32 -- row_codes_exact --, so the sequential sums
  // ARE the multiples and no table kernel ran)
  if constexpr (!C16) {
    for (int i = (int)threadIdx.x; i <= idx_inf; i += (int)blockDim.x)
      Tl[i] = exact_w > 0.0f ? (i < idx_inf ? (float)i * exact_w : INFINITY) : ttab[i];
    __syncthreads();
  }
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int NC = (sx + 63) >> 6;  // <= 32 (launcher)
  const bool three = Z2 > 1;
  const int64_t nvrows = (int64_t)sy * (three ? sz : 1);
  const uint64_t below = (1ull << lane) - 1ull;                         // lanes before this one
  const uint64_t above = lane < 63 ? ~((2ull << lane) - 1ull) : 0ull;   // lanes after it
  const uint64_t above_own = ~below;                                    // this lane and the lanes after it
  // (the value of a cell: fp32, or -- C16 -- its index, carried through the same float variables as an exact small integer)
  auto eval = [&](int il, int ir) -> float {
    il = il < idx_inf ? il : idx_inf;
    ir = ir < idx_inf ? ir : idx_inf;
    if constexpr (C16) {
      // the table is non-decreasing: the smaller of the two values is the value at the smaller index
      const int k = il < ir ? il : ir;
      return (float)(k < idx_inf ? k : 0xFFFF);
    } else {
      const float tl = Tl[il], tr = Tl[ir];
      const float d = tl < tr ? tl : tr;
      float f = d * d;                                    // `d[i] *= d[i]` (src/edt.hpp:116-118)
      if (!bb && f >= INFINITY) f = 3.402823466e+38f;     // tofinite (src/edt.hpp:39-45)
      return f;
    }
  };
  auto put = [&](int64_t at, float v) {
    if constexpr (C16) codes[at] = (uint16_t)(int)v;
    else F1[at] = v;
  };
  for (int64_t vrow = (int64_t)blockIdx.x * 4 + wave; vrow < nvrows; vrow += (int64_t)gridDim.x * 4) {
    const int y = (int)(vrow % sy), z = (int)(vrow / sy);
    const T *lrow = labels + vrow * sx;
    const uint8_t *grow = graph + vrow * sx;
    // a doubled row trimmed by black_border is all background (src/edt_voxel_graph.hpp:156-187)
    const bool deadY = bb && y == sy - 1, deadZ = bb && three && z == sz - 1;
    // ---- phase A (forward): the masks of every 64-voxel chunk; last member before each chunk ----
    uint32_t fl[4] = {0, 0, 0, 0};  // bit c: the lane's voxel of chunk c is in mask F / X / Y / Z
    int prevv[4];                   // lane c: last member of the mask in the chunks before c
    int run[4];                     // (wave-uniform)
    uint32_t diff = 0;              // (wave-uniform) bit m: mask m differs from F somewhere in the row
#pragma unroll
    for (int m = 0; m < 4; ++m) { prevv[m] = -kVgFar; run[m] = -kVgFar; }
    // (round 6: the row's voxels eight chunks at a time, every load from a valid address and all sixteen in flight together --
    // as a load per chunk inside the loop each chunk waited for its own trip to memory, eight dependent trips per 512-voxel row)
    for (int c0 = 0; c0 < NC; c0 += 8) {
    T lv[8];
    uint8_t gv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = (c0 + k) * 64 + lane, xc = x < sx ? x : sx - 1;
      lv[k] = lrow[xc];
      gv[k] = grow[xc];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + k;
      if (c >= NC) break;
      const int x = c * 64 + lane;
      const bool valid = x < sx;
      const bool fg = valid && vg_fg(lv[k]);
      const uint32_t g = valid ? gv[k] : 0u;
      bool in[4];
      in[0] = valid && !fg;
      in[1] = valid && !(fg && (g & 0x01u));
      in[2] = valid && !(fg && (g & 0x04u));
      in[3] = valid && three && !(fg && (g & 0x10u));
      uint64_t M[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        M[m] = __ballot(in[m]);
        fl[m] |= (in[m] ? 1u : 0u) << c;
        if (lane == c) prevv[m] = run[m];
        if (M[m]) run[m] = c * 64 + 63 - __builtin_clzll(M[m]);
      }
      diff |= (M[1] != M[0] ? 1u : 0u) | (M[2] != M[0] ? 2u : 0u) | (M[3] != M[0] ? 4u : 0u);
    }
    }
    // ---- phase B (backward): nearest members on either side, table look-up, square ----
    int nxt[4];  // (wave-uniform) first member of the mask in the chunks after the current one
#pragma unroll
    for (int m = 0; m < 4; ++m) nxt[m] = kVgFar;
    for (int c = NC - 1; c >= 0; --c) {
      const int x = c * 64 + lane;
      const bool valid = x < sx;
      uint64_t M[4];
      int L[4], R[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        M[m] = __ballot(((fl[m] >> c) & 1u) != 0u);
        L[m] = 0;
        R[m] = 0;
      }
      auto search = [&](int m, uint64_t right_lanes) {
        const int p = __builtin_amdgcn_readlane(prevv[m], c);
        const uint64_t lo = M[m] & below, hi = M[m] & right_lanes;
        L[m] = lo ? c * 64 + 63 - __builtin_clzll(lo) : p;
        R[m] = hi ? c * 64 + __builtin_ctzll(hi) : nxt[m];
      };
      search(0, above);  // (the voxel itself being background makes every one of its even cells 0: never "own")
      const int bl = bb ? 2 * x + 1 : kVgFar, br = bb ? 2 * sx - 1 - 2 * x : kVgFar;
      auto mn = [](int a, int b) { return a < b ? a : b; };
      // row (1,1): even F, odd F -- the nearest cells are the odd one on the left, the even one on the right
      const float f3 = eval(mn(bl, 2 * (x - L[0]) - 1), mn(br, 2 * (R[0] - x)));
      const bool ownF = (fl[0] >> c) & 1u;
      float f0 = f3, f1 = f3, f2 = f3;
      bool own1 = ownF, own2 = ownF;
      if (diff & 1u) {  // row (0,0): even F, odd X (own odd cell included)
        search(1, above_own);
        f0 = eval(mn(bl, 2 * (x - L[1]) - 1), mn(br, mn(2 * (R[0] - x), 2 * (R[1] - x) + 1)));
      }
      if (diff & 2u) {  // row (1,0): even Y, odd F
        search(2, above);
        f1 = eval(mn(bl, mn(2 * (x - L[2]), 2 * (x - L[0]) - 1)), mn(br, 2 * (R[2] - x)));
        own1 = (fl[2] >> c) & 1u;
      }
      if (three && (diff & 4u)) {  // row (0,1): even Z, odd F
        search(3, above);
        f2 = eval(mn(bl, mn(2 * (x - L[3]), 2 * (x - L[0]) - 1)), mn(br, 2 * (R[3] - x)));
        own2 = (fl[3] >> c) & 1u;
      }
      if (valid) {
        const int64_t r00 = ((int64_t)(2 * z) * Y2 + 2 * y) * sx + x;
        put(r00, ownF ? 0.0f : f0);
        put(r00 + sx, (own1 || deadY) ? 0.0f : f1);
        if (three) {
          const int64_t r01 = r00 + (int64_t)Y2 * sx;
          put(r01, (own2 || deadZ) ? 0.0f : f2);
          put(r01 + sx, (ownF || deadY || deadZ) ? 0.0f : f3);
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (M[m]) nxt[m] = c * 64 + __builtin_ctzll(M[m]);
    }
  }
}

// Bit planes of the column passes for the even-X columns.  nz = the cell is foreground, rs = it differs from the
// cell below it (row 0: always set), as edt_rowwave.hip defines them.  A word covers 32 doubled rows = 16 voxels
// along the axis; every voxel is loaded unconditionally (clamped), all 17 loads of a thread in flight together.
//
// Along Y: words [Z][Y/32][x] over the doubled Y axis of slice Z.  One thread builds the words of the two slices
// Z = 2z, 2z+1 from one read of its voxels: slice 2z holds (fg, fg && +y) per voxel, slice 2z+1 (fg && +z, fg).
template <typename T>
__global__ void __launch_bounds__(256)
k_vg_bits_y(const T *__restrict__ labels, const uint8_t *__restrict__ graph, uint32_t *__restrict__ nz,
            uint32_t *__restrict__ rs, int sx, int sy, int sz, int Y2, int three, int nwords, int bb) {
  const int64_t total = (int64_t)sx * nwords * sz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % sx);
    const int wd = (int)((idx / sx) % nwords);
    const int z = (int)(idx / ((int64_t)sx * nwords));
    const int64_t base = x + (int64_t)sx * sy * z;
    const int v0 = wd * 16;
    T lab[17];
    uint8_t gr[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) {  // k = 0: the voxel before the word (its odd cell is the carry)
      int v = v0 + k - 1;
      v = v < 0 ? 0 : (v < sy ? v : sy - 1);
      lab[k] = labels[base + (int64_t)sx * v];
      gr[k] = graph[base + (int64_t)sx * v];
    }
    uint32_t w0 = 0, w1 = 0, carry0 = 0, carry1 = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
      const int v = v0 + k - 1;
      const bool fg = vg_fg(lab[k]);
      const uint32_t g = gr[k];
      bool e0 = fg, o0 = fg && (g & 0x04u);  // slice 2z:   Y = 2v, 2v+1
      bool e1 = fg && (g & 0x10u), o1 = fg;  // slice 2z+1
      if (bb && v == sy - 1) { o0 = false; o1 = false; }
      if (k == 0) {
        carry0 = o0 ? 1u : 0u;
        carry1 = o1 ? 1u : 0u;
      } else if (v < sy) {
        w0 |= (e0 ? 1u : 0u) << (2 * k - 2) | (o0 ? 1u : 0u) << (2 * k - 1);
        w1 |= (e1 ? 1u : 0u) << (2 * k - 2) | (o1 ? 1u : 0u) << (2 * k - 1);
      }
    }
    if (wd == 0) { carry0 = 0; carry1 = 0; }
    uint32_t r0 = w0 ^ ((w0 << 1) | carry0), r1 = w1 ^ ((w1 << 1) | carry1);
    if (wd == 0) { r0 |= 1u; r1 |= 1u; }
    const int valid = Y2 - wd * 32;  // rows of this word that exist
    if (valid < 32) { r0 &= (1u << valid) - 1u; r1 &= (1u << valid) - 1u; }
    const int64_t o0 = x + (int64_t)sx * (wd + (int64_t)nwords * (three ? 2 * z : 0));
    nz[o0] = w0;
    rs[o0] = r0;
    if (three) {
      if (bb && z == sz - 1) {  // trimmed slice: all background
        w1 = 0u;
        r1 = wd == 0 ? 1u : 0u;
      }
      nz[o0 + (int64_t)sx * nwords] = w1;
      rs[o0 + (int64_t)sx * nwords] = r1;
    }
  }
}

// Along Z: words [y][Z/32][x] over the doubled Z axis of the even row 2y: (fg, fg && +z) per voxel.
template <typename T>
__global__ void __launch_bounds__(256)
k_vg_bits_z(const T *__restrict__ labels, const uint8_t *__restrict__ graph, uint32_t *__restrict__ nz,
            uint32_t *__restrict__ rs, int sx, int sy, int sz, int Z2, int nwords, int bb) {
  const int64_t total = (int64_t)sx * nwords * sy;
  const int64_t sxy = (int64_t)sx * sy;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % sx);
    const int wd = (int)((idx / sx) % nwords);
    const int y = (int)(idx / ((int64_t)sx * nwords));
    const int64_t base = x + (int64_t)sx * y;
    const int v0 = wd * 16;
    T lab[17];
    uint8_t gr[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) {
      int v = v0 + k - 1;
      v = v < 0 ? 0 : (v < sz ? v : sz - 1);
      lab[k] = labels[base + sxy * v];
      gr[k] = graph[base + sxy * v];
    }
    uint32_t w = 0, carry = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
      const int v = v0 + k - 1;
      const bool fg = vg_fg(lab[k]);
      bool o = fg && (gr[k] & 0x10u);
      if (bb && v == sz - 1) o = false;
      if (k == 0) carry = o ? 1u : 0u;
      else if (v < sz) w |= (fg ? 1u : 0u) << (2 * k - 2) | (o ? 1u : 0u) << (2 * k - 1);
    }
    if (wd == 0) carry = 0;
    uint32_t r = w ^ ((w << 1) | carry);
    if (wd == 0) r |= 1u;
    const int valid = Z2 - wd * 32;
    if (valid < 32) r &= (1u << valid) - 1u;
    nz[idx] = w;
    rs[idx] = r;
  }
}

// The even cells of the even rows of the even slices -> the caller's array.
__global__ void k_vg_gather_even(const float *__restrict__ F1, float *__restrict__ out, int64_t sx, int64_t sy,
                                 int64_t sz, int64_t Y2, int ndim) {
  const int64_t total = sx * sy * sz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = idx % sx, y = (idx / sx) % sy, z = idx / (sx * sy);
    out[idx] = F1[x + sx * (2 * y + Y2 * (ndim == 3 ? 2 * z : 0))];
  }
}
// (rows of whole 16-byte granules: four cells per thread)
__global__ void k_vg_gather_even4(const float *__restrict__ F1, float *__restrict__ out, int64_t sx4, int64_t sy,
                                  int64_t sz, int64_t Y2, int ndim) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int64_t total = sx4 * sy * sz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = idx % sx4, y = (idx / sx4) % sy, z = idx / (sx4 * sy);
    reinterpret_cast<v4f *>(out)[idx] =
        __builtin_nontemporal_load(reinterpret_cast<const v4f *>(F1) + x + sx4 * (2 * y + Y2 * (ndim == 3 ? 2 * z : 0)));
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
  // hand-over list of the integer column kernel (edt_colq16.hip): two counters + the tile ids of the larger pass
  b += align_up(8 * sizeof(uint32_t), 256);
  b += align_up((size_t)(ceil_div(sx, 16) * (ceil_div(std::max<int64_t>(Z2, sy), 8) * 8)) * sizeof(uint32_t), 256);
  return b + 256;
}

static bool g_vg_debug_gather() { return (debug_mode() & 0x200000) != 0; }

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
  uint32_t *q16_counts = reinterpret_cast<uint32_t *>(take(8 * sizeof(uint32_t)));
  uint32_t *q16_ids = reinterpret_cast<uint32_t *>(take((size_t)(ceil_div(sx, 16) * (ceil_div(std::max<int64_t>(Z2, sy), 8) * 8)) * sizeof(uint32_t)));
  const int idx_inf = (int)(2 * sx + 2);
  // half voxel size on the doubled grid (src/edt_voxel_graph.hpp:96-101, :189-193)
  const float hx = wx / 2, hy = wy / 2, hz = wz / 2;
  const bool exact = row_codes_exact(hx, idx_inf);  // k * hx exact for every index of the table
  if (!exact) hipLaunchKernelGGL(k_vg_ttab, dim3(1), dim3(64), 0, stream, ttab, hx, idx_inf);
  // Index form (3-D, rows of whole granules, exact multiples): pass X writes 16-bit indices into the SECOND half of the F1
  // allocation; the Y pass reads them and writes its even rows -- all the Z pass reads -- compactly into the FIRST half
  // ([z2][y][x], one row per voxel row).  (debug bits 0x100000 / 0x200000: the fp32 form / the separate gather pass.)
  const bool index_form = exact && ndim == 3 && sx % 4 == 0 && idx_inf < 0xFFFF && !g_vg_debug_gather() &&
                          !(debug_mode() & (0x100000 | 64));
  uint16_t *codes = index_form ? reinterpret_cast<uint16_t *>(F1 + sx * sy * Z2) : nullptr;
  {
    const int64_t nvrows = sy * (ndim == 3 ? sz : 1);  // one wave per voxel row (its 2 or 4 doubled rows)
    int64_t blocks = ceil_div(nvrows, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (index_form) {
      hipLaunchKernelGGL((k_vg_rows<T, true>), dim3((unsigned)blocks), dim3(256), 0, stream, labels, graph, ttab, F1, codes,
                         (int)sx, (int)sy, (int)sz, (int)Y2, (int)Z2, bb, idx_inf, hx);
    } else {
      static std::atomic<uint64_t> attr_done{0};
      EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_vg_rows<T, false>)));
      hipLaunchKernelGGL((k_vg_rows<T, false>), dim3((unsigned)blocks), dim3(256), (size_t)(idx_inf + 1) * sizeof(float), stream,
                         labels, graph, ttab, F1, codes, (int)sx, (int)sy, (int)sz, (int)Y2, (int)Z2, bb, idx_inf,
                         exact ? hx : 0.0f);
    }
  }
  {
    const int64_t total = sx * nbY * (ndim == 3 ? sz : 1);  // one thread per word of slice 2z AND of slice 2z+1
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((k_vg_bits_y<T>), dim3((unsigned)blocks), dim3(256), 0, stream, labels, graph, nzY, rsY,
                       (int)sx, (int)sy, (int)(ndim == 3 ? sz : 1), (int)Y2, ndim == 3 ? 1 : 0, (int)nbY, bb);
  }
  EDT_HIP_TRY(hipGetLastError());
  const int last_epi = (bb ? 0 : kEpiToInf) | (want_sqrt ? kEpiSqrt : 0) | kEpiStream;  // (the last pass writes the call's results: streamed)
  AxisGeom gy;
  gy.sx = sx; gy.n = Y2; gy.stride = sx; gy.nouter = Z2; gy.outer_stride = sx * Y2; gy.nbands = nbY;
  // Only the even rows of the doubled columns are read again: by the z pass (in place), or -- last pass -- by the
  // caller, whose array the column kernel writes them to directly (ColumnOut: no gather pass)
  ColumnOut even;
  even.stride = 2;
  ColumnOut last = even;
  last.compact = out;
  last.outer = ndim == 3 ? sx : 0;          // 3-D: outer index of the z pass = y
  last.row2 = ndim == 3 ? sx * sy : sx;     // two doubled rows = one voxel slice / one voxel row
  if (g_vg_debug_gather()) last = even;     // (diagnostics: debug bit 0x200000 keeps the separate gather pass)
  // The integer column kernel (edt_colq16.hip, output stride 2) where the half voxel sizes share a quantum and pass X left exact
  // multiples; the tiles it refuses go to the fp32 kernel through the list, as everywhere.
  float q16_q = 1.0f;
  uint32_t q16_a[3] = {1u, 1u, 1u};
  bool q16 = false;
  {
    const float h3[3] = {hx, hy, hz};
    q16 = exact && !(debug_mode() & (16 | 64 | 0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x40000)) && q16_quantum(h3, ndim, &q16_q, q16_a) &&
          (reinterpret_cast<uintptr_t>(out) % 8) == 0;
    if (q16) EDT_HIP_TRY(hipMemsetAsync(q16_counts, 0, 8 * sizeof(uint32_t), stream));
  }
  // in16 != nullptr: the pass reads pass X as 16-bit indices (and writes compactly: co.compact)
  auto column_pass = [&](const uint16_t *in16, const uint32_t *nzp, const uint32_t *rsp, const AxisGeom &g, float h, int axis, int epi,
                         const ColumnOut &co, uint32_t *count) -> int {
    TileList list;
    if (q16 && column_pass_q16_supported(g) && column_pass_wave_supported(g) && column_pass_q16_aligned(F1, in16, nullptr, co.compact)) {
      const int r = launch_column_pass_q16(F1, in16, rsp, g, q16_q, q16_a[axis], q16_a[0], bb, epi, count, q16_ids, stream, nullptr,
                                           nullptr, nullptr, 0, &co);
      if (r != EDT_OK) return r;
      list.count = count;
      list.ids = q16_ids;
    }
    if (in16 != nullptr)
      return launch_column_pass_wave_codes(F1, in16, nzp, rsp, g, h, bb, epi, hx, bb ? 0 : 1, stream, nullptr, list, co);
    return launch_column_pass_wave(F1, nzp, rsp, g, h, bb, epi, stream, nullptr, co, list);
  };
  ColumnOut ycompact = even;  // index form: the even rows of the Y pass, one per voxel row, in the first half of F1
  ycompact.compact = F1;
  ycompact.outer = sx * sy;
  ycompact.row2 = sx;
  int rc = column_pass(codes, nzY, rsY, gy, hy, 1, ndim == 2 ? last_epi : 0, index_form ? ycompact : (ndim == 2 ? last : even), q16_counts);
  if (rc != EDT_OK) return rc;
  if (ndim == 3) {
    const int64_t total = sx * nbZ * sy;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL((k_vg_bits_z<T>), dim3((unsigned)blocks), dim3(256), 0, stream, labels, graph, nzZ, rsZ,
                       (int)sx, (int)sy, (int)sz, (int)Z2, (int)nbZ, bb);
    EDT_HIP_TRY(hipGetLastError());
    AxisGeom gz;  // the even rows only: outer index = y, two doubled rows apart (index form: the compact rows of the Y pass)
    gz.sx = sx; gz.n = Z2; gz.stride = sx * Y2; gz.nouter = sy; gz.outer_stride = 2 * sx; gz.nbands = nbZ;
    if (index_form) { gz.stride = sx * sy; gz.outer_stride = sx; }
    rc = column_pass(nullptr, nzZ, rsZ, gz, hz, 2, last_epi, last, q16_counts + 1);
    if (rc != EDT_OK) return rc;
  }
  if (last.compact != nullptr) return EDT_OK;
  const int64_t nz_out = ndim == 3 ? sz : 1;
  if (sx % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) {
    int64_t blocks = ceil_div(sx / 4 * sy * nz_out, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_vg_gather_even4, dim3((unsigned)blocks), dim3(256), 0, stream, F1, out, sx / 4, sy, nz_out, Y2, ndim);
  } else {
    int64_t blocks = ceil_div(sx * sy * nz_out, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_vg_gather_even, dim3((unsigned)blocks), dim3(256), 0, stream, F1, out, sx, sy, nz_out, Y2, ndim);
  }
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
