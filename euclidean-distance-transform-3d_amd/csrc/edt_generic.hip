// edt_generic.hip -- size-agnostic fallback kernels.
//
// These kernels accept ANY extents (they keep their per-column state in global memory)
// and are the path taken when a row/column is too long for the LDS-tiled kernels of
// edt_tiled.hip, or when EDT_FLAG_FORCE_GENERIC is set (the tests use that to
// cross-check the two implementations against each other).
// They are written for clarity and exactness first; the tuned path lives in edt_tiled.hip.
//
// What is computed (algorithm-independent statement; reference: src/edt.hpp:70-119,
// :168-377, :411-484):
//   pass 1 (x):  per row, per maximal run of one non-zero label, distance to the nearer
//                run end by sequential fp32 additions of wx, squared in fp32;
//   pass 2/3  :  per column, per maximal run [a,b] of one non-zero label,
//                  out[p] = fl32( min_j  w2*(p-j)^2 + F[j] ),  j in [a,b],  fp64, no FMA,
//                then min'ed with the border parabolas fl32(w2*(p-a+1)^2) / fl32(w2*(b-p+1)^2)
//                where a border (volume edge with black_border, or a label change) exists.
// The lower envelope is built as the lower convex hull of the points (j, F[j] + w2*j^2)
// with a division-free orientation test; only the value of the minimum reaches the
// output, evaluated with the reference's own expression (src/edt.hpp:230, :307).
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

namespace edt_amd {

// ------------------------------------------------------------------------------------
// Pass 1, one thread per row, RUN BY RUN (size-agnostic fallback: EDT_FLAG_FORCE_GENERIC, 1-D calls without a
// workspace).  A voxel's value is fl32(T[k]^2): T[k] the k-fold sequential fp32 sum of w (what the reference's two
// sweeps accumulate, src/edt.hpp:83-114), k its distance in voxels to the nearer end of its run that has a border (a
// label change, or the row's end with black_border); T is non-decreasing, so the minimum of the two sweeps is the value at
// the smaller index.  The thread finds a maximal run [s, e] and fills it from its bordered end(s) inwards with ONE running
// sum -- no forward / backward sweep over the row, no second visit of a voxel.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void k_row_pass_runs(const T *__restrict__ labels, float *__restrict__ out,
                                int64_t sx, int64_t nrows, float w, int bb, int to_finite,
                                int take_sqrt) {
  const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  const T *seg = labels + row * sx;
  float *d = out + row * sx;
  auto value = [&](float t) {
    float v = t * t;                                   // `d[i] *= d[i]` (src/edt.hpp:116-118)
    if (to_finite && isinf(v)) v = FLT_MAX;            // tofinite (src/edt.hpp:39-45)
    return take_sqrt ? sqrtf(v) : v;
  };
  int64_t s = 0;
  while (s < sx) {
    const T lab = seg[s];
    int64_t e = s;
    while (e + 1 < sx && seg[e + 1] == lab) ++e;
    const int64_t len = e - s + 1;
    if (lab == 0) {
      for (int64_t i = s; i <= e; ++i) d[i] = 0.0f;
    } else {
      const bool left = bb || s > 0, right = bb || e < sx - 1;
      if (!left && !right) {
        const float v = value(INFINITY);               // a row inside one label: no boundary at all
        for (int64_t i = s; i <= e; ++i) d[i] = v;
      } else {
        // j-th voxel from a bordered end: T[j + 1]; with both borders the two fills meet in the middle
        const int64_t count = (left && right) ? (len + 1) / 2 : len;
        float t = 0.0f;
        for (int64_t j = 0; j < count; ++j) {
          t = t + w;
          const float v = value(t);
          if (left) d[s + j] = v;
          if (right) d[e - j] = v;
        }
      }
    }
    s = e + 1;
  }
}

template <typename T>
static int launch_row_serial_t(const void *labels, float *out, int64_t sx, int64_t nrows, float w,
                               int bb, int to_finite, int take_sqrt, hipStream_t stream) {
  const int threads = 64;
  const int64_t blocks = ceil_div(nrows, threads);
  hipLaunchKernelGGL(k_row_pass_runs<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,
                     (const T *)labels, out, sx, nrows, w, bb, to_finite, take_sqrt);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_row_pass_serial(int dtype, const void *labels, float *out, int64_t sx, int64_t nrows,
                           float w, int bb, int to_finite, int take_sqrt, hipStream_t stream) {
  switch (dtype) {
    case EDT_U8: case EDT_BOOL:
      return launch_row_serial_t<uint8_t>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    case EDT_U16: return launch_row_serial_t<uint16_t>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    case EDT_U32: return launch_row_serial_t<uint32_t>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    case EDT_U64: return launch_row_serial_t<uint64_t>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    case EDT_F32: return launch_row_serial_t<float>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    case EDT_F64: return launch_row_serial_t<double>(labels, out, sx, nrows, w, bb, to_finite, take_sqrt, stream);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
}

// ------------------------------------------------------------------------------------
// Run structure of a scan axis as two bit-volumes (labels are compared exactly once, here):
//   nz word (o, b, x): bit r set  <=>  label(x, row 32b+r, o) != 0
//   rs word (o, b, x): bit r set  <=>  row 32b+r starts a run (row 0, or label differs
//                                       from the previous row; src/edt.hpp:355-368)
// Word layout [o][b][x] so that the column passes read them coalesced across x.
// One thread per word.  `halo`, when given, supplies the row "-1" of every column
// (used by the Z-sharded path where the previous slab lives on another GPU).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void k_axis_bits(const T *__restrict__ labels, const T *__restrict__ halo,
                            uint32_t *__restrict__ nzbits, uint32_t *__restrict__ rsbits,
                            AxisGeom g) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = g.sx * g.nbands * g.nouter;
  if (idx >= total) return;
  const int64_t x = idx % g.sx;
  const int64_t b = (idx / g.sx) % g.nbands;
  const int64_t o = idx / (g.sx * g.nbands);
  const T *col = labels + x + o * g.outer_stride;
  const int64_t r0 = b * kBandRows;
  T prev = 0;
  bool have_prev = false;
  if (r0 > 0) { prev = col[(r0 - 1) * g.stride]; have_prev = true; }
  else if (halo != nullptr) { prev = halo[x + o * g.outer_stride]; have_prev = true; }
  uint32_t nz = 0, rs = 0;
  for (int r = 0; r < kBandRows; ++r) {
    const int64_t row = r0 + r;
    if (row >= g.n) break;
    const T here = col[row * g.stride];
    if (here != 0) nz |= 1u << r;
    if (!have_prev || here != prev) rs |= 1u << r;
    prev = here;
    have_prev = true;
  }
  nzbits[idx] = nz;
  rsbits[idx] = rs;
}

template <typename T>
static int launch_bits_t(const void *labels, const void *halo, uint32_t *nz, uint32_t *rs,
                         const AxisGeom &g, hipStream_t stream) {
  const int threads = 256;
  const int64_t total = g.sx * g.nbands * g.nouter;
  hipLaunchKernelGGL(k_axis_bits<T>, dim3((unsigned)ceil_div(total, threads)), dim3(threads), 0,
                     stream, (const T *)labels, (const T *)halo, nz, rs, g);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_axis_bits(int dtype, const void *labels, const void *halo, uint32_t *nz, uint32_t *rs,
                     const AxisGeom &g, hipStream_t stream) {
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: return launch_bits_t<uint8_t>(labels, halo, nz, rs, g, stream);
    case EDT_U16: return launch_bits_t<uint16_t>(labels, halo, nz, rs, g, stream);
    case EDT_U32: return launch_bits_t<uint32_t>(labels, halo, nz, rs, g, stream);
    case EDT_U64: return launch_bits_t<uint64_t>(labels, halo, nz, rs, g, stream);
    case EDT_F32: return launch_bits_t<float>(labels, halo, nz, rs, g, stream);
    case EDT_F64: return launch_bits_t<double>(labels, halo, nz, rs, g, stream);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
}

// ------------------------------------------------------------------------------------
// Bit planes of the reference's *binary* route (EDT_FLAG_BINARY_YZ; pyedt::_binary_edt{2,3}dsq<T>,
// src/edt.hpp:528-567, :722-755): passes 2 and 3 do not split a column at label changes -- every column is ONE
// envelope from its first non-zero value to its end, background voxels taking part as height-0 parabolas.  Rows
// before that first value hold 0 and keep it (their own parabola), and the border site the reference puts just
// before the first value is such a row, so the whole column [0, n) as a single all-foreground run gives the same
// values.  The planes therefore do not depend on the labels: foreground everywhere, one run start at row 0.
// zs (y-packed "differs from z-1" plane of a slab starting at slice o0, or nullptr): set on slice 0 only.
// ------------------------------------------------------------------------------------
__global__ void k_planes_one_run(uint32_t *__restrict__ nzbits, uint32_t *__restrict__ rsbits,
                                 uint32_t *__restrict__ zsbits, AxisGeom g, int64_t o0) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = g.sx * g.nbands * g.nouter;
  if (idx >= total) return;
  const int64_t b = (idx / g.sx) % g.nbands;
  const int64_t o = idx / (g.sx * g.nbands);
  const int64_t left = g.n - b * kBandRows;  // rows of this band that exist (>= 1)
  const uint32_t mask = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u);
  nzbits[idx] = mask;
  rsbits[idx] = b == 0 ? 1u : 0u;
  if (zsbits != nullptr) zsbits[idx] = (o + o0 == 0) ? mask : 0u;
}

int launch_planes_one_run(uint32_t *nz, uint32_t *rs, uint32_t *zs, const AxisGeom &g, int64_t o0, hipStream_t stream) {
  const int threads = 256;
  const int64_t total = g.sx * g.nbands * g.nouter;
  if (total <= 0) return EDT_OK;
  hipLaunchKernelGGL(k_planes_one_run, dim3((unsigned)ceil_div(total, threads)), dim3(threads), 0, stream, nz, rs, zs,
                     g, o0);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// ------------------------------------------------------------------------------------
// Passes 2/3, one thread per column, hull vertices kept in a global-memory stack that
// shares the volume's addressing (entry k of column c lives at stack[c + k*stride], so
// lanes that agree on k access it coalesced).  Not in place: reads `fin`, writes `fout`.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t next_run_start(const uint32_t *rsw, int64_t word_stride,
                                                  int64_t nbands, int64_t after, int64_t n) {
  int64_t q = after + 1;
  int64_t wi = q >> 5;
  if (wi >= nbands) return n;
  uint32_t wv = rsw[wi * word_stride] & (~0u << (q & 31));
  while (true) {
    if (wv) {
      const int64_t pos = wi * 32 + __builtin_ctz(wv);
      return pos < n ? pos : n;
    }
    if (++wi >= nbands) return n;
    wv = rsw[wi * word_stride];
  }
}

__global__ void k_column_pass_serial(const float *__restrict__ fin, float *__restrict__ fout,
                                     const uint32_t *__restrict__ nzbits,
                                     const uint32_t *__restrict__ rsbits,
                                     int32_t *__restrict__ stack, AxisGeom g, float w, int bb,
                                     int epi) {
  const int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (col >= g.sx * g.nouter) return;
  const int64_t x = col % g.sx, o = col / g.sx;
  const int64_t base = x + o * g.outer_stride;
  const uint32_t *nzw = nzbits + o * g.nbands * g.sx + x;
  const uint32_t *rsw = rsbits + o * g.nbands * g.sx + x;
  const float *f = fin + base;
  float *out = fout + base;
  int32_t *stk = stack + base;
  const int64_t n = g.n, st = g.stride;
  const double w2 = (double)(w * w);  // fp32 product widened (src/edt.hpp:181, :258)

  // ---- sweep 1: lower convex hull of every run -----------------------------------
  int64_t k = 0, kb = 0;
  int64_t ia = -1, ib = -1;
  double Fa = 0.0, Fb = 0.0;
  uint32_t nzword = 0, rsword = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int r = (int)(i & 31);
    if (r == 0) {
      nzword = nzw[(i >> 5) * g.sx];
      rsword = rsw[(i >> 5) * g.sx];
    }
    if ((rsword >> r) & 1u) kb = k;
    if (!((nzword >> r) & 1u)) continue;
    const double Fi = (double)f[i * st];
    while (k - kb >= 2) {
      const double lhs = hull_num(Fb, Fi, ib, i, w2) * (double)(ib - ia);
      const double rhs = hull_num(Fa, Fb, ia, ib, w2) * (double)(i - ib);
      if (!(lhs <= rhs)) break;
      --k;  // vertex ib is on or above the chord (ia, i): drop it
      ib = ia;
      Fb = Fa;
      if (k - kb >= 2) {
        ia = stk[(k - 2) * st];
        Fa = (double)f[ia * st];
      }
    }
    stk[k * st] = (int32_t)i;
    ++k;
    ia = ib; Fa = Fb;
    ib = i;  Fb = Fi;
  }

  // ---- sweep 2: evaluate the envelope -----------------------------------------------
  const int64_t ktot = k;
  int64_t kk = 0;
  int64_t run_lo = 0, run_hi = -1;
  int64_t j = 0, jn = -1;
  double Fj = 0.0, Fjn = 0.0;
  for (int64_t p = 0; p < n; ++p) {
    const int r = (int)(p & 31);
    if (r == 0) {
      nzword = nzw[(p >> 5) * g.sx];
      rsword = rsw[(p >> 5) * g.sx];
    }
    const bool nz = (nzword >> r) & 1u;
    if ((rsword >> r) & 1u) {
      run_lo = p;
      run_hi = next_run_start(rsw, g.sx, g.nbands, p, n) - 1;
      if (nz) {
        while (stk[kk * st] < p) ++kk;  // skip hull vertices of earlier runs
        j = p;
        Fj = (double)f[p * st];
        jn = -1;
        if (kk + 1 < ktot) {
          jn = stk[(kk + 1) * st];
          if (jn > run_hi) jn = -1; else Fjn = (double)f[jn * st];
        }
      }
    }
    if (!nz) {
      out[p * st] = 0.0f;
      continue;
    }
    double best = w2 * sqd(p - j) + Fj;
    while (jn >= 0) {
      const double cand = w2 * sqd(p - jn) + Fjn;
      if (!(cand < best)) break;
      best = cand;
      ++kk;
      j = jn; Fj = Fjn;
      jn = -1;
      if (kk + 1 < ktot) {
        jn = stk[(kk + 1) * st];
        if (jn > run_hi) jn = -1; else Fjn = (double)f[jn * st];
      }
    }
    float m = (float)best;
    if (bb || run_lo > 0) m = fminf((float)(w2 * sqd(p - run_lo + 1)), m);
    if (bb || run_hi < n - 1) m = fminf((float)(w2 * sqd(run_hi - p + 1)), m);
    out[p * st] = finish(m, epi);
  }
}

int launch_column_pass_serial(const float *fin, float *fout, const uint32_t *nz, const uint32_t *rs,
                              int32_t *stack, const AxisGeom &g, float w, int bb, int epi,
                              hipStream_t stream) {
  const int threads = 64;
  const int64_t cols = g.sx * g.nouter;
  hipLaunchKernelGGL(k_column_pass_serial, dim3((unsigned)ceil_div(cols, threads)), dim3(threads), 0,
                     stream, fin, fout, nz, rs, stack, g, w, bb, epi);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// ------------------------------------------------------------------------------------
// Small elementwise helpers.
// ------------------------------------------------------------------------------------
__global__ void k_subtract(const float *__restrict__ a, const float *__restrict__ b,
                           float *__restrict__ out, int64_t count) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += step) out[i] = a[i] - b[i];
}

int launch_subtract(const float *a, const float *b, float *out, int64_t count, hipStream_t stream) {
  if (count <= 0) return EDT_OK;
  const int threads = 256;
  int64_t blocks = ceil_div(count, threads);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_subtract, dim3((unsigned)blocks), dim3(threads), 0, stream, a, b, out, count);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// The sign of the SIGNED transform (sdf / sdfsq, src/edt.pyx:121-202: edt(x) - edt(x == 0); edt_api.hip, EDT_FLAG_SIGNED):
// the field of the transform that measured label 0 like every label, negated where the label is 0.  Four voxels per
// thread (the volume's tail one by one).
template <typename T>
__global__ void k_negate_background(const T *__restrict__ labels, float *__restrict__ f, int64_t count) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int64_t quads = count >> 2;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = i; q < quads; q += step) {
    T l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = labels[4 * q + k];
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(f) + q);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = l[k] == T(0) ? -v[k] : v[k];
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(f) + q);
  }
  for (int64_t j = 4 * quads + i; j < count; j += step) f[j] = labels[j] == T(0) ? -f[j] : f[j];
}

int launch_negate_background(int dtype, const void *labels, float *f, int64_t count, hipStream_t stream) {
  if (count <= 0) return EDT_OK;
  if (reinterpret_cast<uintptr_t>(f) % 16 != 0) { set_error("output must be 16-byte aligned"); return EDT_ERR_BAD_ARG; }
  const int threads = 256;
  int64_t blocks = ceil_div(ceil_div(count, 4), threads);
  if (blocks > 16384) blocks = 16384;
#define LAUNCH_NB(T)                                                                              \
  hipLaunchKernelGGL(k_negate_background<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,   \
                     (const T *)labels, f, count)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_NB(uint8_t); break;
    case EDT_U16: LAUNCH_NB(uint16_t); break;
    case EDT_U32: LAUNCH_NB(uint32_t); break;
    case EDT_U64: LAUNCH_NB(uint64_t); break;
    case EDT_F32: LAUNCH_NB(float); break;
    case EDT_F64: LAUNCH_NB(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_NB
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

template <typename T>
__global__ void k_is_background(const T *__restrict__ labels, uint8_t *__restrict__ mask,
                                int64_t count) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += step) mask[i] = (labels[i] == 0) ? 1 : 0;
}

int launch_is_background(int dtype, const void *labels, uint8_t *mask, int64_t count,
                         hipStream_t stream) {
  if (count <= 0) return EDT_OK;
  const int threads = 256;
  int64_t blocks = ceil_div(count, threads);
  if (blocks > 8192) blocks = 8192;
#define LAUNCH_BG(T)                                                                          \
  hipLaunchKernelGGL(k_is_background<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,    \
                     (const T *)labels, mask, count)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_BG(uint8_t); break;
    case EDT_U16: LAUNCH_BG(uint16_t); break;
    case EDT_U32: LAUNCH_BG(uint32_t); break;
    case EDT_U64: LAUNCH_BG(uint64_t); break;
    case EDT_F32: LAUNCH_BG(float); break;
    case EDT_F64: LAUNCH_BG(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_BG
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// each(): the distance transform restricted to ONE label (src/edt.pyx:950-994 builds this image on
// the host from run lists: zeros + transfer_run_voxels, src/edt_voxel_graph.hpp:290-310).  One
// streaming pass: 4 voxels per lane so that the fp32 side moves in 16-byte pieces.
template <typename T>
__global__ void k_select_label(const T *__restrict__ labels, const float *__restrict__ dt,
                               float *__restrict__ out, T key, int64_t count) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < count; i += step) {
    if (i + 4 <= count && (reinterpret_cast<uintptr_t>(dt + i) % 16) == 0 &&
        (reinterpret_cast<uintptr_t>(out + i) % 16) == 0) {
      const float4 v = *reinterpret_cast<const float4 *>(dt + i);
      float4 r;
      r.x = labels[i] == key ? v.x : 0.0f;
      r.y = labels[i + 1] == key ? v.y : 0.0f;
      r.z = labels[i + 2] == key ? v.z : 0.0f;
      r.w = labels[i + 3] == key ? v.w : 0.0f;
      *reinterpret_cast<float4 *>(out + i) = r;
    } else {
      for (int64_t j = i; j < count && j < i + 4; ++j) out[j] = labels[j] == key ? dt[j] : 0.0f;
    }
  }
}

int launch_select_label(int dtype, const void *labels, const float *dt, float *out, const void *key,
                        int64_t count, hipStream_t stream) {
  if (count <= 0) return EDT_OK;
  const int threads = 256;
  int64_t blocks = ceil_div(ceil_div(count, 4), threads);
  if (blocks > 16384) blocks = 16384;
#define LAUNCH_SEL(T)                                                                         \
  hipLaunchKernelGGL(k_select_label<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,     \
                     (const T *)labels, dt, out, *(const T *)key, count)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_SEL(uint8_t); break;
    case EDT_U16: LAUNCH_SEL(uint16_t); break;
    case EDT_U32: LAUNCH_SEL(uint32_t); break;
    case EDT_U64: LAUNCH_SEL(uint64_t); break;
    case EDT_F32: LAUNCH_SEL(float); break;
    case EDT_F64: LAUNCH_SEL(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_SEL
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace edt_amd
