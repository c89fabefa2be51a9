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

// X32: c_d = w2 * d^2 is exactly representable in fp32 for every d < N (voxel sizes like 1, 6, 30, 40, 0.5; checked on
// the host, brute_exact_prefix): fl32(fl64(c_d + F)) is then the plain fp32 sum (double rounding is innocuous for a sum
// when the wide format has >= 2p + 2 bits), so a candidate is one v_add_f32 and one integer minimum (non-negative
// floats and +inf order like their bit patterns) instead of an fp64 fma, an fp64 minimum and a conversion.
template <int N, bool BB, bool X32>
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
      float res;
      if constexpr (X32) {
        uint32_t best = __float_as_uint(f[p]);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          if (j != p) {
            const float cd = w2f * (float)((p - j) * (p - j));              // exact (the host checked)
            const uint32_t c = __float_as_uint(f[j] + cd);                  // (+inf rows beyond the axis never win)
            best = c < best ? c : best;
          }
        }
        res = __uint_as_float(best);
      } else {
        double best = (double)f[p];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          if (j != p) {
            const double d = (double)(p - j);
            best = fmin(best, __builtin_fma(w2 * d, d, (double)f[j]));   // (+inf rows beyond the axis never win)
          }
        }
        res = (float)best;
      }
      // borders of p's run: last run start at or below p, first run start above it
      const uint32_t lowm = rs & (0xFFFFFFFFu >> (31 - p));
      const int s = 31 - __builtin_clz(lowm);
      const uint32_t him = p < 31 ? (rs & (0xFFFFFFFEu << p)) : 0u;
      int e = him ? __builtin_ctz(him) - 1 : n - 1;
      if (e > n - 1) e = n - 1;
      float dm = INFINITY;
      if (BB || s > 0) dm = (float)(p - s + 1);
      if (BB || e < n - 1) dm = fminf(dm, (float)(e + 1 - p));
      if (dm < INFINITY) res = fminf(res, w2f * (dm * dm));
      col[(int64_t)p * g.stride] = finish(res, epi);
    }
  }
}

// c_d = fl32(w * w) * d^2 exactly representable in fp32 (and finite) for every d <= dmax
bool short_exact_c(float w, int dmax) {
  const double w2 = (double)(w * w);
  for (int d = 1; d <= dmax; ++d) {
    const double c = w2 * (double)(d * d);
    if ((double)(float)c != c || !(c < 3.0e38)) return false;
  }
  return true;
}

template <int N>
int launch_short_n(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb, int epi,
                   hipStream_t stream) {
  const int64_t cols = g.sx * g.nouter;
  if (cols <= 0) return EDT_OK;
  const int64_t blocks = ceil_div(cols, 256);
  if (blocks > 0x7FFFFFFF) { set_error("too many columns"); return EDT_ERR_UNSUPPORTED; }
  // fp32 candidates where every c_d the column can meet is exact in fp32 (debug bit 0x8000: fp64 candidates, as on the
  // windowed path)
  const bool x32 = !(debug_mode() & 0x8000) && w * w >= 1.17549435e-38f && short_exact_c(w, N - 1);
#define SHORT_GO(B, X) hipLaunchKernelGGL((k_column_pass_short<N, B, X>), dim3((unsigned)blocks), dim3(256), 0, stream, F, nz, rs, g, w, epi & 3)
  if (bb) { if (x32) SHORT_GO(true, true); else SHORT_GO(true, false); }
  else { if (x32) SHORT_GO(false, true); else SHORT_GO(false, false); }
#undef SHORT_GO
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
