// edt_line.hip -- the 1-D transform (pyedt::squared_edt_1d_multi_seg on a whole array, reference:
// src/edt.hpp:70-119; Python: edt1d / edt1dsq and edt() of a 1-D array, src/edt.pyx:316-356) as a data-parallel
// pipeline for lines of any length.
//
// Closed form of the reference's two fp32 sweeps: for voxel i inside the maximal run [s, e) of
// one non-zero label,  d = min(T[i-s+1], T[e-i])  with  T[k] = the k-fold sequential fp32 sum of w
// (a side counts only if there is a neighbour run / background there, or black_border), result fl32(d*d).
// Run starts are found per 1024-voxel block; the last start before and the first start after every block
// come from ONE small scan over the blocks, so a run may span any number of blocks.
//   k_line_marks : per block, position of its last / first run start (one ballot per wave)
//   k_line_scan  : exclusive prefix max / suffix min over the blocks (one workgroup)
//   k_line_eval  : per voxel, nearest run start on either side (bit scans of the block's sixteen ballot masks + the
//                  carried values), table look-up, square (+ optional sqrt)
// T is exact as k*w when w = m * 2^e with m*(n+1) < 2^24 (all the usual anisotropies); otherwise it is
// tabulated -- the sums are sequentially rounded -- in parallel: every chunk of the table starts from a value reached by
// jumping through the binades (edt_seqsum.h).
#include "edt_common.h"
#include "edt_kernels.h"
#include "edt_seqsum.h"

#include <cmath>

#pragma clang fp contract(off)

namespace edt_amd {

namespace {

constexpr int kLineBlock = 1024;
constexpr int kNoneLo = -1;           // "no run start below" (positions are row-relative 32-bit ints: extents < 2^31)
constexpr int kNoneHi = 0x7FFFFFFF;   // "no run start above"

// a run starts where the label changes (run extraction: one flat array)
template <typename T>
__device__ __forceinline__ bool line_starts(const T *lab, int64_t i) { return i == 0 || lab[i] != lab[i - 1]; }

// Where a block of the line pipeline sits.  The array is `rows` rows of `row` voxels (one line: rows = 1); every row is
// cut into `bpr` blocks of 1024 voxels of its own (a block never straddles two rows, so no voxel needs a 64-bit
// division to find its row: one 32-bit division of the block index per thread).  Positions INSIDE the kernels are
// row-relative 32-bit integers (x); only the final addresses are 64-bit.
struct LinePos {
  int64_t r0;     // first voxel of this thread's row (global index)
  int x;          // position of this thread's voxel in its row
  unsigned bx;    // block of the row
  bool in;        // the voxel exists
};
__device__ __forceinline__ LinePos line_pos(int64_t row, unsigned bpr) {
  const unsigned b = blockIdx.x, r = b / bpr;
  LinePos p;
  p.bx = b - r * bpr;
  p.x = (int)(p.bx * kLineBlock + threadIdx.x);
  p.r0 = (int64_t)r * row;
  p.in = (int64_t)p.x < row;
  return p;
}

// The run starts of a 1024-voxel block as sixteen 64-bit masks (one ballot per wave) in LDS, plus ONE 16-bit word
// saying which waves have any: the nearest start below / above a voxel is then a handful of bit operations -- own
// mask, else the nearest wave with a start (one bit scan of the 16-bit word) and a bit scan of that wave's mask, else
// the block's carried value.  No loops, no 64-bit positions.  (Round 2 ran two ten-step Hillis-Steele scans over int64
// positions in LDS per block; the first ballot version walked the masks in per-lane loops and paid ~450 instructions per
// wave, most of them scalar: profiles/r03_long_shapes_*.txt.)
constexpr int kLineWaves = kLineBlock / 64;

__device__ __forceinline__ int hi_bit64(unsigned long long m) { return 63 - __builtin_clzll(m); }
__device__ __forceinline__ int lo_bit64(unsigned long long m) { return __builtin_ctzll(m); }

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_line_marks(const T *__restrict__ lab, int *__restrict__ blk_last, int *__restrict__ blk_first, int64_t row,
             unsigned bpr) {
  __shared__ unsigned long long s_mask[kLineWaves];
  const LinePos p = line_pos(row, bpr);
  const int64_t i = p.r0 + p.x;
  const unsigned long long m = __ballot(p.in && (p.x == 0 || lab[i] != lab[i - 1]));
  if ((threadIdx.x & 63) == 0) s_mask[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    const unsigned long long mw = lane < kLineWaves ? s_mask[lane] : 0ull;
    const unsigned has = (unsigned)__ballot(mw != 0ull);
    if (lane == 0) {
      const int base = (int)(p.bx * kLineBlock);
      int last = kNoneLo, first = kNoneHi;
      if (has) {
        const int wl = 31 - __builtin_clz(has), wf = __builtin_ctz(has);
        last = base + wl * 64 + hi_bit64(s_mask[wl]);
        first = base + wf * 64 + lo_bit64(s_mask[wf]);
      }
      blk_last[blockIdx.x] = last;
      blk_first[blockIdx.x] = first;
    }
  }
}

// One line only: in place, blk_last[b] <- last run start in blocks < b (or -1); blk_first[b] <- first run start in
// blocks > b (or n).
__global__ void __launch_bounds__(1024) k_line_scan(int *blk_last, int *blk_first, int64_t nblk, int n) {
  __shared__ int buf[1024];
  int carry = kNoneLo;
  for (int64_t base = 0; base < nblk; base += 1024) {
    const int64_t b = base + threadIdx.x;
    buf[threadIdx.x] = b < nblk ? blk_last[b] : kNoneLo;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : kNoneLo;
      __syncthreads();
      if (t > buf[threadIdx.x]) buf[threadIdx.x] = t;
      __syncthreads();
    }
    const int excl = threadIdx.x > 0 ? buf[threadIdx.x - 1] : kNoneLo;
    const int total = buf[1023];
    __syncthreads();
    if (b < nblk) blk_last[b] = excl > carry ? excl : carry;
    carry = total > carry ? total : carry;
  }
  int carry2 = n;
  for (int64_t top = nblk; top > 0; top -= 1024) {
    const int64_t b = top - 1 - threadIdx.x;  // thread 0 takes the last block of the chunk
    int v = b >= 0 ? blk_first[b] : kNoneHi;
    if (v == kNoneHi) v = n;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int t = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : n;
      __syncthreads();
      if (t < buf[threadIdx.x]) buf[threadIdx.x] = t;
      __syncthreads();
    }
    const int excl = threadIdx.x > 0 ? buf[threadIdx.x - 1] : n;
    const int total = buf[1023];
    __syncthreads();
    if (b >= 0) blk_first[b] = excl < carry2 ? excl : carry2;
    carry2 = total < carry2 ? total : carry2;
  }
}

// The table T[0 .. count) of the sequential fp32 sums of w, built in PARALLEL: a thread jumps to the first entry of its
// chunk through the binades (edt_seqsum.h: bit-identical to walking there) and fills the chunk with real additions.
// (Rounds 1-2: one thread, count dependent additions -- seconds for a 2^31-voxel line at a voxel size like 0.1.)
constexpr int64_t kTabChunk = 1024;
__global__ void k_line_ttab(float *__restrict__ ttab, float w, int64_t count) {
  const int64_t k0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * kTabChunk;
  if (k0 >= count) return;
  const int64_t k1 = k0 + kTabChunk < count ? k0 + kTabChunk : count;
  float t = edt_seq_sum_at(w, k0);
  for (int64_t k = k0; k < k1; ++k) {
    ttab[k] = t;
    t = t + w;
  }
}

template <typename T>
__global__ void __launch_bounds__(kLineBlock)
k_line_eval(const T *__restrict__ lab, float *__restrict__ out, const int *__restrict__ blk_last,
            const int *__restrict__ blk_first, const float *__restrict__ ttab, float w, int bb, int take_sqrt,
            int64_t row, unsigned bpr, int to_finite, int bpr_local) {
  __shared__ unsigned long long s_mask[kLineWaves];
  __shared__ int s_before, s_after;
  const LinePos p = line_pos(row, bpr);
  const int64_t i = p.r0 + p.x;
  const bool in = p.in;
  const T here = in ? lab[i] : T(0);
  const unsigned long long mine = __ballot(in && (p.x == 0 || here != lab[i - 1]));
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  if (lane == 0) s_mask[wv] = mine;
  // The nearest start before / after the whole block is the same for every voxel of it: looked up ONCE per block (by
  // the first wave, its loads in flight next to everybody's label loads) and shared through LDS.
  if (wv == 0) {
    int sb = kNoneLo, sa = kNoneHi;
    if (bpr_local) {
      // a stack of rows: the marks are per block (no scan over all blocks); a row has a handful of blocks and its first
      // one starts with a run.  lanes 0..31: the blocks before this one, nearest first; lanes 32..63: the blocks after
      for (unsigned k0 = 0; k0 < bpr; k0 += 32) {
        const unsigned k = k0 + (lane & 31) + 1;
        int v = (lane < 32) ? kNoneLo : kNoneHi;
        if (lane < 32) { if (k <= p.bx) v = blk_last[blockIdx.x - k]; }
        else if (p.bx + k < bpr) v = blk_first[blockIdx.x + k];
        const unsigned long long has = __ballot(lane < 32 ? v != kNoneLo : v != kNoneHi);
        const unsigned lo = (unsigned)(has & 0xFFFFFFFFull), hi = (unsigned)(has >> 32);
        if (sb == kNoneLo && lo) sb = __shfl(v, __builtin_ctz(lo));
        if (sa == kNoneHi && hi) sa = __shfl(v, 32 + __builtin_ctz(hi));
        if ((sb != kNoneLo || k0 + 32 >= p.bx) && (sa != kNoneHi || p.bx + k0 + 33 >= bpr)) break;
      }
      if (sa == kNoneHi) sa = (int)row;  // the next row's first voxel (or the end of the array)
    } else {
      sb = blk_last[blockIdx.x];   // (scanned: last start before / first start after the block; >= 0 resp. <= n)
      sa = blk_first[blockIdx.x];
    }
    if (lane == 0) { s_before = sb; s_after = sa; }
  }
  __syncthreads();
  // which waves of the block hold a start (every wave computes the word for itself: 16 LDS reads, one ballot)
  const unsigned wmask = (unsigned)__ballot(lane < kLineWaves && s_mask[lane < kLineWaves ? lane : 0] != 0ull);
  if (!in) return;
  const int base = (int)(p.bx * kLineBlock);
  // last start at or before x
  int sx_;
  {
    const unsigned long long below = mine & (~0ull >> (63 - lane));  // bits 0 .. lane
    const unsigned wb = wmask & ((1u << wv) - 1u);                    // waves before this one that hold a start
    if (below) sx_ = base + wv * 64 + hi_bit64(below);
    else if (wb) { const int kw = 31 - __builtin_clz(wb); sx_ = base + kw * 64 + hi_bit64(s_mask[kw]); }
    else sx_ = s_before;
  }
  // first start after x
  int ex_;
  {
    const unsigned long long above = lane < 63 ? (mine & (~0ull << (lane + 1))) : 0ull;  // bits lane+1 .. 63
    const unsigned wa = wv < kLineWaves - 1 ? (wmask & (~0u << (wv + 1))) : 0u;
    if (above) ex_ = base + wv * 64 + lo_bit64(above);
    else if (wa) { const int kw = __builtin_ctz(wa); ex_ = base + kw * 64 + lo_bit64(s_mask[kw]); }
    else ex_ = s_after;
  }
  float v = 0.0f;
  if (here != 0) {
    auto tk = [&](int k) -> float { return ttab ? ttab[k] : (float)k * w; };
    // the ends of the row (of the line) are borders only with black_border
    const float dl = (sx_ > 0 || bb) ? tk(p.x - sx_ + 1) : INFINITY;
    const float dr = ((int64_t)ex_ < row || bb) ? tk(ex_ - p.x) : INFINITY;
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
  if (row <= 0) row = n;                       // one line = one row
  const int64_t rows = n / row;
  const int64_t bpr = ceil_div(row, kLineBlock);
  const int64_t nblk = rows * bpr;
  if (nblk > 0x7FFFFFFF) { set_error("line too long"); return EDT_ERR_UNSUPPORTED; }
  char *p = static_cast<char *>(ws);
  int *blk_last = reinterpret_cast<int *>(p);   // (row-relative positions: 32 bits; the carve below keeps 8 bytes per entry)
  int *blk_first = blk_last + nblk;
  float *ttab = nullptr;
  if (!multiples_exact(w, row + 1)) {          // (no run is longer than a row)
    ttab = reinterpret_cast<float *>(p + align_up((size_t)(2 * nblk) * sizeof(int64_t), 256));
    const int64_t tthreads = ceil_div(row + 2, kTabChunk);
    hipLaunchKernelGGL(k_line_ttab, dim3((unsigned)ceil_div(tthreads, 64)), dim3(64), 0, stream, ttab, w, row + 2);
  }
  hipLaunchKernelGGL(k_line_marks<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, blk_last, blk_first, row,
                     (unsigned)bpr);
  // one line: the marks are turned into "last start before / first start after the block" by a scan over the blocks;
  // a stack of rows needs none (every row's first voxel starts a run: k_line_eval looks at the row's own few blocks)
  const int local = rows > 1 ? 1 : 0;
  if (!local) hipLaunchKernelGGL(k_line_scan, dim3(1), dim3(1024), 0, stream, blk_last, blk_first, nblk, (int)n);
  hipLaunchKernelGGL(k_line_eval<T>, dim3((unsigned)nblk), dim3(kLineBlock), 0, stream, lab, out, blk_last,
                     blk_first, ttab, w, bb, take_sqrt, row, (unsigned)bpr, to_finite, local);
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
  const int64_t nblk = nrows * ceil_div(sx, kLineBlock);
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
