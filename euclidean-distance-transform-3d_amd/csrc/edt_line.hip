// edt_line.hip -- the 1-D transform (pyedt::squared_edt_1d_multi_seg on a whole array, reference:
// src/edt.hpp:70-119; Python: edt1d / edt1dsq and edt() of a 1-D array, src/edt.pyx:316-356) as a data-parallel
// pipeline for lines of any length.
//
// Closed form of the reference's two fp32 sweeps: for voxel i inside the maximal run [s, e) of
// one non-zero label,  d = min(T[i-s+1], T[e-i])  with  T[k] = the k-fold sequential fp32 sum of w
// (a side counts only if there is a neighbour run / background there, or black_border), result fl32(d*d).
// Run starts are found per 1024-voxel block; the last start before and the first start after every block
// come from ONE small scan over the blocks, so a run may span any number of blocks.
//   k_line_marks : per block, position of its last / first run start
//   k_line_scan  : exclusive prefix max / suffix min over the blocks (one workgroup)
//   k_line_eval  : per voxel, nearest run start on either side (block scan in LDS + the carried values),
//                  table look-up, square (+ optional sqrt)
// T is exact as k*w when w = m * 2^e with m*(n+1) < 2^24 (all the usual anisotropies); otherwise it is
// tabulated once by a single thread -- the sums are sequentially rounded, there is no closed form.
#include "edt_common.h"
#include "edt_kernels.h"

#include <cmath>

#pragma clang fp contract(off)

namespace edt_amd {

namespace {

constexpr int kLineBlock = 1024;
constexpr int64_t kNone = -1;

// a run starts where the label changes -- and at the first voxel of every row when the array is a stack of rows
// of `row` voxels (row = 0: one line)
template <typename T>
__device__ __forceinline__ bool line_starts(const T *lab, int64_t i, int64_t row = 0) {
  return i == 0 || lab[i] != lab[i - 1] || (row > 0 && i % row == 0);
}

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_line_marks(const T *__restrict__ lab, int64_t n, int64_t *__restrict__ blk_last, int64_t *__restrict__ blk_first,
             int64_t row) {
  __shared__ int64_t s_last, s_first;
  if (threadIdx.x == 0) { s_last = 0; s_first = INT64_MAX; }  // s_last holds position + 1 (0: none)
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kLineBlock + threadIdx.x;
  if (i < n && line_starts(lab, i, row)) {
    atomicMax((unsigned long long *)&s_last, (unsigned long long)(i + 1));
    atomicMin((unsigned long long *)&s_first, (unsigned long long)i);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_last[blockIdx.x] = s_last - 1;
    blk_first[blockIdx.x] = s_first;
  }
}

// in place: blk_last[b] <- last run start in blocks < b (or -1); blk_first[b] <- first run start in blocks > b (or n)
__global__ void __launch_bounds__(1024) k_line_scan(int64_t *blk_last, int64_t *blk_first, int64_t nblk, int64_t n) {
  __shared__ int64_t buf[1024];
  // prefix max, chunk by chunk
  int64_t carry = kNone;
  for (int64_t base = 0; base < nblk; base += 1024) {
    const int64_t b = base + threadIdx.x;
    int64_t v = b < nblk ? blk_last[b] : kNone;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int64_t t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : kNone;
      __syncthreads();
      if (t > buf[threadIdx.x]) buf[threadIdx.x] = t;
      __syncthreads();
    }
    const int64_t incl = buf[threadIdx.x];
    const int64_t excl = threadIdx.x > 0 ? buf[threadIdx.x - 1] : kNone;
    const int64_t total = buf[1023];
    __syncthreads();
    if (b < nblk) blk_last[b] = excl > carry ? excl : carry;
    carry = total > carry ? total : carry;
    (void)incl;
  }
  // suffix min, chunk by chunk from the end
  int64_t carry2 = n;
  for (int64_t top = nblk; top > 0; top -= 1024) {
    const int64_t b = top - 1 - threadIdx.x;  // thread 0 takes the last block of the chunk
    int64_t v = b >= 0 ? blk_first[b] : INT64_MAX;
    if (v == INT64_MAX) v = n;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int64_t t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : n;
      __syncthreads();
      if (t < buf[threadIdx.x]) buf[threadIdx.x] = t;
      __syncthreads();
    }
    const int64_t excl = threadIdx.x > 0 ? buf[threadIdx.x - 1] : n;
    const int64_t total = buf[1023];
    __syncthreads();
    if (b >= 0) blk_first[b] = excl < carry2 ? excl : carry2;
    carry2 = total < carry2 ? total : carry2;
  }
}

__global__ void k_line_ttab(float *__restrict__ ttab, float w, int64_t count) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float acc = 0.0f;
  ttab[0] = 0.0f;
  for (int64_t k = 1; k < count; ++k) {
    acc = acc + w;
    ttab[k] = acc;
  }
}

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_line_eval(const T *__restrict__ lab, float *__restrict__ out, int64_t n, const int64_t *__restrict__ blk_last,
            const int64_t *__restrict__ blk_first, const float *__restrict__ ttab, float w, int bb, int take_sqrt,
            int64_t row, int to_finite) {
  __shared__ int64_t s_lo[kLineBlock], s_hi[kLineBlock];
  const int64_t i = (int64_t)blockIdx.x * kLineBlock + threadIdx.x;
  const bool in = i < n;
  const bool st = in && line_starts(lab, i, row);
  // last start at or before i / first start after i, inside the block
  s_lo[threadIdx.x] = st ? i : kNone;
  s_hi[threadIdx.x] = st ? i : INT64_MAX;
  __syncthreads();
  for (int d = 1; d < kLineBlock; d <<= 1) {
    const int64_t a = threadIdx.x >= (unsigned)d ? s_lo[threadIdx.x - d] : kNone;
    const int64_t b = threadIdx.x + d < kLineBlock ? s_hi[threadIdx.x + d] : INT64_MAX;
    __syncthreads();
    if (a > s_lo[threadIdx.x]) s_lo[threadIdx.x] = a;
    if (b < s_hi[threadIdx.x]) s_hi[threadIdx.x] = b;
    __syncthreads();
  }
  if (!in) return;
  int64_t s = s_lo[threadIdx.x];
  if (s == kNone) s = blk_last[blockIdx.x];  // (>= 0: voxel 0 starts a run)
  int64_t e = threadIdx.x + 1 < kLineBlock ? s_hi[threadIdx.x + 1] : INT64_MAX;
  if (e == INT64_MAX) e = blk_first[blockIdx.x];
  float v = 0.0f;
  if (lab[i] != 0) {
    auto tk = [&](int64_t k) -> float { return ttab ? ttab[k] : (float)k * w; };
    // the ends of the line (of the voxel's row, for a stack of rows) are borders only with black_border
    const int64_t r0 = row > 0 ? (i / row) * row : 0, r1 = row > 0 ? r0 + row : n;
    const float dl = (s > r0 || bb) ? tk(i - s + 1) : INFINITY;
    const float dr = (e < r1 || bb) ? tk(e - i) : INFINITY;
    const float d = dl < dr ? dl : dr;
    v = d * d;                       // `d[i] *= d[i]` (src/edt.hpp:116-118)
    if (to_finite && v > 3.402823466e+38f) v = 3.402823466e+38f;  // tofinite (src/edt.hpp:39-45)
    if (take_sqrt) v = sqrtf(v);
  }
  out[i] = v;
}

// k*w exact for every k <= kmax, and so is every partial sum: w = m * 2^e with m * kmax < 2^24
bool multiples_exact(float w, int64_t kmax) {
  if (!(w > 0.0f) || !std::isfinite(w)) return false;
  int e = 0;
  double m = std::frexp((double)w, &e);  // w = m * 2^e, 0.5 <= m < 1
  for (int b = 0; b < 24 && m != std::floor(m); ++b) m *= 2.0;
  if (m != std::floor(m)) return false;
  return m * (double)kmax < 16777216.0;
}

template <typename T>
int launch_line_t(const void *labels, float *out, int64_t n, float w, int bb, int take_sqrt, void *ws,
                  hipStream_t stream, int64_t row = 0, int to_finite = 0) {
  const T *lab = static_cast<const T *>(labels);
  const int64_t nblk = ceil_div(n, kLineBlock);
  if (nblk > 0x7FFFFFFF) { set_error("line too long"); return EDT_ERR_UNSUPPORTED; }
  char *p = static_cast<char *>(ws);
  int64_t *blk_last = reinterpret_cast<int64_t *>(p);
  int64_t *blk_first = blk_last + nblk;
  float *ttab = nullptr;
  const int64_t longest = row > 0 ? row : n;  // no run is longer than a row
  if (!multiples_exact(w, longest + 1)) {
    ttab = reinterpret_cast<float *>(p + align_up((size_t)(2 * nblk) * sizeof(int64_t), 256));
    hipLaunchKernelGGL(k_line_ttab, dim3(1), dim3(64), 0, stream, ttab, w, longest + 2);
  }
  hipLaunchKernelGGL(k_line_marks<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, n, blk_last, blk_first, row);
  hipLaunchKernelGGL(k_line_scan, dim3(1), dim3(1024), 0, stream, blk_last, blk_first, nblk, n);
  hipLaunchKernelGGL(k_line_eval<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, out, n, blk_last,
                     blk_first, ttab, w, bb, take_sqrt, row, to_finite);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Run extraction on the device (pyedt::extract_runs, reference: src/edt_voxel_graph.hpp:238-268): the start
// offsets of the maximal constant runs of the flattened label array, in order.  Count per block, scan the
// blocks, scatter.
// ---------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_runs_count(const T *__restrict__ lab, int64_t n, int64_t *__restrict__ blk_count) {
  const int64_t i = (int64_t)blockIdx.x * kLineBlock + threadIdx.x;
  const bool st = i < n && line_starts(lab, i);
  const int c = __syncthreads_count(st ? 1 : 0);
  if (threadIdx.x == 0) blk_count[blockIdx.x] = c;
}

// exclusive prefix sum over the blocks, in place; the total goes to *total
__global__ void __launch_bounds__(1024) k_runs_scan(int64_t *blk, int64_t nblk, int64_t *total) {
  __shared__ int64_t buf[1024];
  int64_t carry = 0;
  for (int64_t base = 0; base < nblk; base += 1024) {
    const int64_t b = base + threadIdx.x;
    const int64_t v = b < nblk ? blk[b] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int64_t t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    const int64_t incl = buf[threadIdx.x], sum = buf[1023];
    __syncthreads();
    if (b < nblk) blk[b] = carry + incl - v;
    carry += sum;
  }
  if (threadIdx.x == 0) *total = carry;
}

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_runs_write(const T *__restrict__ lab, int64_t n, const int64_t *__restrict__ blk_off, int64_t *__restrict__ starts,
             int64_t capacity) {
  __shared__ int s_rank[kLineBlock];
  const int64_t i = (int64_t)blockIdx.x * kLineBlock + threadIdx.x;
  const bool st = i < n && line_starts(lab, i);
  s_rank[threadIdx.x] = st ? 1 : 0;
  __syncthreads();
  for (int d = 1; d < kLineBlock; d <<= 1) {
    const int t = threadIdx.x >= (unsigned)d ? s_rank[threadIdx.x - d] : 0;
    __syncthreads();
    s_rank[threadIdx.x] += t;
    __syncthreads();
  }
  if (st) {
    const int64_t slot = blk_off[blockIdx.x] + s_rank[threadIdx.x] - 1;
    if (slot < capacity) starts[slot] = i;
  }
}

template <typename T>
int launch_runs_t(const void *labels, int64_t n, int64_t *starts, int64_t capacity, int64_t *total, void *ws,
                  hipStream_t stream) {
  const T *lab = static_cast<const T *>(labels);
  const int64_t nblk = ceil_div(n, kLineBlock);
  if (nblk > 0x7FFFFFFF) { set_error("array too long"); return EDT_ERR_UNSUPPORTED; }
  int64_t *blk = static_cast<int64_t *>(ws);
  hipLaunchKernelGGL(k_runs_count<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, n, blk);
  hipLaunchKernelGGL(k_runs_scan, dim3(1), dim3(1024), 0, stream, blk, nblk, total);
  if (capacity > 0 && starts != nullptr)
    hipLaunchKernelGGL(k_runs_write<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, n, blk, starts, capacity);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace

size_t runs_workspace_bytes(int64_t n) { return align_up((size_t)ceil_div(n, kLineBlock) * sizeof(int64_t), 256) + 256; }

int launch_extract_runs(int dtype, const void *labels, int64_t n, int64_t *starts, int64_t capacity, int64_t *total,
                        void *ws, hipStream_t stream) {
#define RUNS(T) return launch_runs_t<T>(labels, n, starts, capacity, total, ws, stream)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: RUNS(uint8_t);
    case EDT_U16: RUNS(uint16_t);
    case EDT_U32: RUNS(uint32_t);
    case EDT_U64: RUNS(uint64_t);
    case EDT_F32: RUNS(float);
    case EDT_F64: RUNS(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef RUNS
}

size_t line_workspace_bytes(int64_t n) {
  const int64_t nblk = ceil_div(n, kLineBlock);
  return align_up((size_t)(2 * nblk) * sizeof(int64_t), 256) + align_up((size_t)(n + 2) * sizeof(float), 256) + 256;
}

// Pass 1 of a volume whose rows are too long for the wave / workgroup row kernels (sx > 2048), as ONE line of
// nrows * sx voxels with a forced run start at every row's first voxel: block scan + table look-up, every voxel its
// own thread (the size-agnostic kernel this replaces gave every ROW one thread).
size_t rows_line_workspace_bytes(int64_t sx, int64_t nrows) {
  const int64_t nblk = ceil_div(sx * nrows, kLineBlock);
  return align_up((size_t)(2 * nblk) * sizeof(int64_t), 256) + align_up((size_t)(sx + 2) * sizeof(float), 256) + 256;
}

int launch_rows_line_pass(int dtype, const void *labels, float *out, int64_t sx, int64_t nrows, float w, int bb,
                          int to_finite, void *ws, hipStream_t stream) {
#define ROWS(T) return launch_line_t<T>(labels, out, sx * nrows, w, bb, 0, ws, stream, sx, to_finite)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: ROWS(uint8_t);
    case EDT_U16: ROWS(uint16_t);
    case EDT_U32: ROWS(uint32_t);
    case EDT_U64: ROWS(uint64_t);
    case EDT_F32: ROWS(float);
    case EDT_F64: ROWS(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef ROWS
}

int launch_line_pass(int dtype, const void *labels, float *out, int64_t n, float w, int bb, int take_sqrt,
                     void *ws, hipStream_t stream) {
#define LINE(T) return launch_line_t<T>(labels, out, n, w, bb, take_sqrt, ws, stream)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LINE(uint8_t);
    case EDT_U16: LINE(uint16_t);
    case EDT_U32: LINE(uint32_t);
    case EDT_U64: LINE(uint64_t);
    case EDT_F32: LINE(float);
    case EDT_F64: LINE(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LINE
}

}  // namespace edt_amd
