// edt_short.hip -- passes 2 and 3 over SHORT axes (at most 32 rows: one bit word per column).
//
// The LDS-tiled wave kernel gives every 32 columns of such an axis a workgroup of one wave: a 4096 x 4096 x 8 volume
// is half a million single-wave workgroups for its z pass (1.2 ms for 2^27 voxels, all of it launch rate).  Here a
// thread owns a whole column: its <= 32 rows live in registers (row r of 64 adjacent columns is one coalesced load),
// and with so few rows the envelope is taken by brute force over ALL rows of the column --
//     result[p] = min( B_p,  min_j fl64( w2 * (p - j)^2 + F[j] ) ),   B_p = min(F[p], border parabolas of p's run)
// (rows of other runs and background rows are harmless candidates: they lie at or beyond a border site of p's run,
// edt_colwave_lane.h, windowed path) -- in the hull path's own arithmetic: fp64 fma with an exact product, narrowed
// once, then the fp32 border term (src/edt.hpp:230, :233-242, :307-311), fused toinfinite / sqrt (src/edt.hpp:47-53,
// :599-601).
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

namespace edt_amd {

namespace {

template <int N, bool BB>
__global__ void __launch_bounds__(256)
k_column_pass_short(float *__restrict__ F, const uint32_t *__restrict__ nzbits, const uint32_t *__restrict__ rsbits,
                    AxisGeom g, float w, int epi) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= g.sx * g.nouter) return;
  const int64_t o = idx / g.sx, x = idx - o * g.sx;
  const int n = (int)g.n;
  float *col = F + x + o * g.outer_stride;
  const uint32_t nz = nzbits[o * g.sx + x];         // (one band per column: the word index is (o * 1 + 0) * sx + x)
  const uint32_t rs = rsbits[o * g.sx + x] | 1u;    // row 0 starts a run
  if (nz == 0u) return;                             // background columns keep their zeros
  float f[N];
#pragma unroll
  for (int r = 0; r < N; ++r) f[r] = r < n ? col[(int64_t)r * g.stride] : INFINITY;
  const double w2 = (double)(w * w);                // fp32 product widened (src/edt.hpp:181, :258)
  const float w2f = w * w;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    if (p < n && ((nz >> p) & 1u)) {
      double best = (double)f[p];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j != p) {
          const double d = (double)(p - j);
          best = fmin(best, __builtin_fma(w2 * d, d, (double)f[j]));   // (+inf rows beyond the axis never win)
        }
      }
      // borders of p's run: last run start at or below p, first run start above it
      const uint32_t lowm = rs & (0xFFFFFFFFu >> (31 - p));
      const int s = 31 - __builtin_clz(lowm);
      const uint32_t him = p < 31 ? (rs & (0xFFFFFFFEu << p)) : 0u;
      int e = him ? __builtin_ctz(him) - 1 : n - 1;
      if (e > n - 1) e = n - 1;
      float res = (float)best;
      float dm = INFINITY;
      if (BB || s > 0) dm = (float)(p - s + 1);
      if (BB || e < n - 1) dm = fminf(dm, (float)(e + 1 - p));
      if (dm < INFINITY) res = fminf(res, w2f * (dm * dm));
      col[(int64_t)p * g.stride] = finish(res, epi);
    }
  }
}

template <int N>
int launch_short_n(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb, int epi,
                   hipStream_t stream) {
  const int64_t cols = g.sx * g.nouter;
  if (cols <= 0) return EDT_OK;
  const int64_t blocks = ceil_div(cols, 256);
  if (blocks > 0x7FFFFFFF) { set_error("too many columns"); return EDT_ERR_UNSUPPORTED; }
  if (bb) hipLaunchKernelGGL((k_column_pass_short<N, true>), dim3((unsigned)blocks), dim3(256), 0, stream, F, nz, rs, g, w, epi & 3);
  else hipLaunchKernelGGL((k_column_pass_short<N, false>), dim3((unsigned)blocks), dim3(256), 0, stream, F, nz, rs, g, w, epi & 3);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace

bool column_pass_short_supported(const AxisGeom &g) { return g.nbands == 1 && g.n >= 1 && g.n <= 32; }

int launch_column_pass_short(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb,
                             int epi, hipStream_t stream) {
  if (g.n <= 8) return launch_short_n<8>(F, nz, rs, g, w, bb, epi, stream);
  if (g.n <= 16) return launch_short_n<16>(F, nz, rs, g, w, bb, epi, stream);
  return launch_short_n<32>(F, nz, rs, g, w, bb, epi, stream);
}

}  // namespace edt_amd
