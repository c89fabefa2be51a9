// edt_colq16.hip -- the 16-bit integer column pass (passes Y and Z) for gfx950.
//
// Replaces squared_edt_1d_parabolic_multi_seg over squared_edt_1d_parabolic (src/edt.hpp:168-377) wherever the voxel sizes
// of the call share a quantum (edt_colq16_lane.h: quantum_of -- (1,1,1), (6,6,30), (4,4,40), (0.5,0.5,1) ...), i.e. on
// every configuration of BASELINE.json; the mathematics, the exactness argument and the per-lane code are in
// edt_colq16_lane.h.  This file is the workgroup around it:
//
//   workgroup = one tile of 32 adjacent columns x the whole scan axis (as in edt_colwave_kernel.h): 256 threads up to 512
//               rows, 512 beyond;
//   LDS image = the tile as 16-bit integers N = F / q, row-major, 64 bytes per row, kPad = 16 rows of +inf (0xFFFF) before
//               and after it: 34 KiB for a 512-row axis, 39.3 KiB with the planes (the fp32 kernel: 80 KiB) -- FOUR
//               workgroups per CU (measured: 2 / 3 / 4 per CU -> 0.88 / 0.77 / 0.74 ms per cfg3 step); 68 KiB for 1024 rows --
//               two workgroups of eight waves with whole 128-byte lines per row (the fp32 kernel needs 16-column tiles there);
//   fill      = through VGPRs: the 16-bit distance indices k of pass X (index form, edt_rowwave.hip C16: N = k^2 * ax), or
//               fp32 values (N = F / q, checked to be exact), or -- behind a pass that left its results 16-bit -- per row the
//               16-bit plane or fp32 values; a tile that holds a value outside the 16-bit range (objects more than ~250 voxels
//               deep; rows without any boundary: +inf) is worked on again in the WIDE form -- 32-bit lanes, below -- where that
//               form's range holds it; a tile with a value off the quantum grid or beyond that range too is NOT processed:
//               its id goes to a list in device memory and the fp32 kernel (edt_colwave_kernel.h, list mode) takes it
//               afterwards;
//   scans     = run extents across bands by one thread per column and direction over the band words (LDS), break bits per
//               block of 8 rows and column pair;
//   windows   = lane = (column pair, block of 8 rows): a wave works on 16 pairs x the four blocks of one band, a contiguous
//               32 x 32 patch, and walks the bands wave, wave + T/64, ...  (output stride 2 -- the voxel graph's doubled
//               grids --: blocks of 16 rows whose even rows are evaluated, 16 pairs x four such blocks per wave);
//   results   = converted once at the end, (float)N * q (exact), optional correctly rounded sqrt, 8-byte stores (16 lanes =
//               one 128-byte line per row); or the 16-bit values themselves, over the tile's indices (the plane between
//               passes Y and Z); or the rows of the slab records of the Z-sharded path, fp32 or 16-bit (edt_shard_api.hip).
#include "edt_common.h"
#include "edt_kernels.h"

#include <cstdlib>
#include <cstring>

#define EDT_LANE __device__ __forceinline__
#define EDT_LANE_MEMBER __device__ __forceinline__
#include "edt_colq16_lane.h"

// Fill loads.  Round 5 (tools/halfline_probe.hip, profiles/r05_halfline_probe.txt): a 32-column tile reads 16-bit rows as 64-byte
// pieces -- HALF lines whose other half belongs to the neighbouring tile, dispatched right behind on the same XCD.  A
// non-temporal load lets the line go once it has served its 64 bytes, and the neighbour fetches it AGAIN: 340 MB from memory for a
// 268 MB array, 54.5 us against 37.9 us with plain loads.  Plain loads for the 16-bit rows (indices, plane): cfg2 0.634 -> 0.591 ms,
// cfg3 0.728 -> 0.715.  The fp32 rows (whole 128-byte pieces, 16 bytes per lane) stay non-temporal: pass Z of the 1024^3
// volume 2.39 ms against 2.60 with plain loads.  (EDT_Q16_NT_FILL=1 brings the non-temporal ones back for A/B runs.)
#ifndef EDT_Q16_NT_FILL
#define EDT_Q16_NT_FILL 0
#endif
// A/B (round 6): wave priority of the fill phase -- a workgroup that has just started gets its loads out ahead of the ALU work of
// its neighbours on the SIMD (s_setprio; 0 = off).  Measured with priority 2, same box, two runs each: cfg2 pass Y 0.1793 ->
// 0.1780 ms, pass Z 0.2118 -> 0.2598 (the fp32 fill's conversions run at the raised priority too and starve the stores of the
// neighbours); cfg3 0.695 -> 0.744 ms per step.  Left off.
#ifndef EDT_Q16_PRIO
#define EDT_Q16_PRIO 0
#endif
#if EDT_Q16_NT_FILL
#define EDT_Q16_FILL_LOAD(p) __builtin_nontemporal_load(p)
#else
#define EDT_Q16_FILL_LOAD(p) (*(p))
#endif

namespace edt_amd {

struct Q16Args {
  const uint16_t *codes;  // index form of pass X: [outer][row][x] with the strides of F; nullptr: F holds fp32 values
  float q, rq;            // the quantum and its reciprocal (rounded; the conversion is verified value by value)
  uint32_t a;             // c_d = a * d^2 quanta
  uint32_t ain;           // index form: N = k^2 * ain
  uint32_t kmax;          // index form: largest k with k^2 * ain <= nlim
  uint32_t nlim;          // largest N a tile may hold: a * dmax^2
  uint32_t dmax;          // largest d with a * d^2 <= 65534
  // The wide form (edt_colq16_lane.h: V<true>): a tile that holds values beyond nlim but on the quantum grid and at most
  // nlimw is worked on as two half-tiles of 16 columns with 32-bit lanes instead of being handed over.  nlimw == nlim: no
  // wide form (output stride 2, 16-bit slab records, quanta whose odd part leaves no room).
  uint32_t nlimw, dmaxw, kmaxw;
  uint32_t fwmax_bits;    // bit pattern of (float)nlimw * q (exact)
  uint32_t inf_ok;        // the wide form carries +inf (no black border, short enough columns: edt_colq16_lane.h, q16_wide_range)
  const uint32_t *signbits;  // kEpiSign: the true foreground plane of this axis ([outer][band][x] words, like the run starts)
  uint32_t plane_inf_ok;  // O16: the pass that reads the 16-bit plane carries +inf too -- a tile of nothing but +inf may stay there as 0xFFFF
  uint32_t *count;        // tiles handed to the fp32 kernel: *count of them ...
  uint32_t *ids;          // ... their tile ids (outer index * x-tiles + x-tile) in the fp32 kernel's geometry:
  int list_cols;          // its tiles are 32 columns wide, or 16 (axes of more than 512 rows: two ids per refused tile)
  // The 16-bit plane between passes Y and Z (volumes whose indices fit one slab): a tile of pass Y that qualifies writes its
  // results N over its indices (plane == codes, in place) instead of fp32 values to F and sets its bit in `map`
  // ([x-tile][outer index / 32], zeroed by the caller); pass Z takes every row from wherever pass Y left it.
  uint16_t *plane;
  uint32_t *map;          // (slab records: all ones where every row is read from the plane; nullptr where a plane is written and no map kept)
  int map_words;          // words per x-tile
  int64_t pst, p_outer;   // the plane's own row / outer strides in 16-bit elements (slab records: edt_shard_api.hip; the padded
                          // pitch of the index buffer: edt_api.hip, code_pitch), else g's
  int64_t cd_outer;       // outer stride of codes, and of the plane a pass writes over them (g.outer_stride, or the padded pitch)
  // output stride 2 (S = 2: the doubled grids of the voxel-graph transform) only: nullptr = the even rows go to their places
  // in F; else row r of column (x, outer o) goes to compact[x + o * c_outer + (r / 2) * c_row2]   (edt_kernels.h: ColumnOut)
  float *compact;
  int64_t c_outer, c_row2;
};
enum : int { kQ16InF32 = 0, kQ16InCodes = 1, kQ16InMixed = 2 };

namespace {

// what a wide image word holds for an index of pass X / a value of the 16-bit plane: 0xFFFF is +inf in either
__device__ __forceinline__ uint32_t q16_index_value(uint32_t k, uint32_t ain) { return k == 0xFFFFu ? edt_q16::kInfW : k * k * ain; }
__device__ __forceinline__ uint32_t q16_plane_value(uint32_t v) { return v == 0xFFFFu ? edt_q16::kInfW : v; }

__host__ __device__ constexpr int q16_lds_words(int NB) {
  // image (NB bands of 32 rows + 2 kPad rows, 16 words each) + run-start plane + lo/hi plane + break masks (16 pairs x 6
  // words) + flags
  // + the mask of the tile's over-range columns and the column map of a wide pass (8 + 16 words)
  return (NB * 32 + 2 * edt_q16::kPad) * edt_q16::kRowWords + NB * 32 + NB * 32 + 16 * 6 + 8 + 8 + 16;
}

}  // namespace

// (no static LDS anywhere in this kernel -- __syncthreads_or has some -- or hipFuncSetAttribute refuses the full 160 KiB of
// dynamic LDS the 1024-row image asks for)
// IN: where the tile comes from (fp32 values / indices of pass X / per row the 16-bit plane or fp32 values);
// O16: the results go to the 16-bit plane (in place over the indices) instead of F; SC: ... to the slab records
// T: threads of the workgroup -- 256 (four waves) up to 512 rows, 512 beyond (a 1024-row image leaves room for two
// workgroups per CU: eight waves each keep the SIMDs as busy as the four workgroups of four waves of the shorter axes)
// S: output stride -- 1 = every row; 2 = blocks of 16 rows whose even rows are evaluated and written (fp32 values in, or
// indices in and a compact destination out)
template <bool BB, int IN, bool O16, bool SC, int T, int S = 1>
__global__ void __launch_bounds__(T, 4)
k_column_pass_q16(float *__restrict__ F, const uint32_t *__restrict__ rsbits, AxisGeom g, int tiles_x, int epi, int dbg,
                  Q16Args qa, const BandScatter *__restrict__ scatter) {
  using namespace edt_q16;
  extern __shared__ __attribute__((aligned(16))) uint32_t q16_smem[];
  const int n = (int)g.n;
  const int NB = (int)g.nbands;
  const int nb32 = NB * 32;
  uint32_t *img = q16_smem;                                // [nb32 + 2 kPad][16]
  uint32_t *rsp = img + (nb32 + 2 * kPad) * kRowWords;     // [NB][32]
  uint32_t *lohi = rsp + NB * 32;                          // [NB][32]: (lo_in + 1) | (hi_out + 1) << 16
  uint32_t *bm = lohi + NB * 32;                           // [16][6]: break bits of the pair's blocks, words 1..4 (0, 5: zero)
  uint32_t *flags = bm + 16 * 6;                           // [T / 64]: per wave, "the tile does not qualify"
  uint32_t *ovm = flags + 8;                               // [0]: the columns of the tile that hold a value beyond 16 bits
  uint32_t *wcol = ovm + 8;                                // [16]: tile column of every image column of a wide pass
  const int t = (int)threadIdx.x;

  // ---- tile -> (x-tile, outer index): the XCD-aware order of edt_colwave_kernel.h ----
  int64_t tile_id = blockIdx.x;
  const uint32_t utx = (uint32_t)tiles_x;
  if (!(dbg & 0x800)) {
    const uint32_t tt = (uint32_t)tile_id, x = tt & 7u, j = tt >> 3;
    const uint32_t jq = j / utx, jr = j - jq * utx;
    tile_id = (int64_t)((jq * 8u + x) * utx + jr);
    if (tile_id >= (int64_t)tiles_x * g.nouter) return;
  }
  const uint32_t oq = (uint32_t)tile_id / utx;
  const int64_t xt = (uint32_t)tile_id - oq * utx, o = oq;
  const int64_t x0 = xt * 32;
  const int64_t st = g.stride;
  const int cols_left = (int)(g.sx - x0);

  // ---- phase 0: the tile, HBM -> 16-bit LDS image ------------------------------------------
#if EDT_Q16_PRIO
  __builtin_amdgcn_s_setprio(EDT_Q16_PRIO);
#endif
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int RPS = T / 8;            // rows per sweep of the workgroup: 8 threads x 4 columns per row
  const int r_in = t >> 3, cg = t & 7;
  const bool col_ok = 4 * cg < cols_left;
  uint32_t rsany = 0u;  // a run start behind row 0 somewhere in the tile (the words this thread loads)
  bool bad = false;   // the tile has no integer form at all: handed to the fp32 kernel
  bool over = false;  // ... no 16-bit form (a value beyond nlim): the wide form if `bad` stays false
  // which of the thread's columns (4 cg .. 4 cg + 3) hold such a value: the two halves of ov01 / ov23 (index, plane rows), the
  // low bits of ovq (fp32 rows)
  pk ov01 = 0u, ov23 = 0u;
  uint32_t ovq = 0u;
  if (t < 64) bm[t] = 0u, bm[t + 32] = 0u;  // (96 words)
  if (t == 0) ovm[0] = 0u, ovm[1] = 0u, ovm[2] = 0u;
  if (t < 8 * kPad) {
    // +inf around the column: 2 x kPad rows x 16 words, one 16-byte store per thread
    const int row = t < 4 * kPad ? -kPad + (t >> 2) : nb32 + ((t - 4 * kPad) >> 2);
    *reinterpret_cast<v4u *>(img + (row + kPad) * kRowWords + 4 * (t & 3)) = (v4u){~0u, ~0u, ~0u, ~0u};
  }
  // The run-start words of the tile ([outer][band][x]: 32 words per band, NB * 32 <= 2 T of them -- launch_q16_k) as at most two
  // loads per thread from addresses that are valid for every thread (a thread without a word reads word 0 of the tile and drops
  // it), issued AHEAD of the fill's loads and put into LDS behind them: straight-line code, one trip to memory for all of it.
  // (Round 6: as a loop of load -> LDS store the compiler waited for each word before the next load and before the fill's first
  // load went out -- three dependent trips per tile, tools/yorder_probe.hip has the bare pattern at half the pass's time.)
  uint32_t rsw[2];
  bool rsw_ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = t + k * T;
    rsw_ok[k] = u < NB * 32 && (u & 31) < cols_left;
    const int uc = rsw_ok[k] ? u : 0;
    rsw[k] = rsbits[(o * g.nbands + (uc >> 5)) * g.sx + x0 + (uc & 31)];
  }
  auto put_run_starts = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int u = t + k * T;
      if (u < NB * 32) {
        const uint32_t w = rsw_ok[k] ? rsw[k] : 0u;
        rsp[u] = w;
        rsany |= u < 32 ? (w & ~1u) : w;  // (row 0 starts a run in every column)
      }
    }
  };
  // (index form: the whole tile in ONE sweep of sixteen loads per thread -- nb32 <= 16 RPS: 512 rows at 256 threads, 1024 at
  // 512: launch_q16_k)
  if constexpr (IN == kQ16InCodes) {
    v2u kk[16];
    const uint16_t *src = qa.codes + x0 + o * qa.cd_outer + 4 * cg;
    const pk kmaxpk = pk_both(qa.kmax), kmaxw1pk = pk_both(qa.kmaxw + qa.inf_ok), ainpk = pk_both(qa.ain);
    const pk infadd = qa.inf_ok ? 0x00010001u : 0u;
    {
      constexpr int i0 = 0;
      // (every load from a valid address -- rows behind the column's end read its last row, columns behind the row's end the
      // tile's first four -- and dropped afterwards: sixteen loads back to back, no branch between them)
      const uint16_t *srcv = col_ok ? src : src - 4 * cg;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = i0 + RPS * j + r_in;
        kk[j] = EDT_Q16_FILL_LOAD(reinterpret_cast<const v2u *>(srcv + (int64_t)(row < n ? row : n - 1) * st));
      }
      put_run_starts();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = i0 + RPS * j + r_in;
        if (!(row < n && col_ok)) kk[j] = (v2u){0u, 0u};
        if (row < nb32) {
          // k > kmax: the tile has no 16-bit form (k^2 may have wrapped: never used); kmaxw < k < 0xFFFF: no wide form either
          ov01 |= pk_subs(kk[j][0], kmaxpk);
          ov23 |= pk_subs(kk[j][1], kmaxpk);
          // (0xFFFF -- no boundary in the row: +inf -- wraps to 0 and passes where the wide form carries +inf)
          bad |= (pk_subs(pk_add(kk[j][0], infadd), kmaxw1pk) | pk_subs(pk_add(kk[j][1], infadd), kmaxw1pk)) != 0u;
          v2u v = {pk_mul(pk_mul(kk[j][0], kk[j][0]), ainpk), pk_mul(pk_mul(kk[j][1], kk[j][1]), ainpk)};
          if (row >= n) v = (v2u){~0u, ~0u};
          *reinterpret_cast<v2u *>(img + (row + kPad) * kRowWords + 2 * cg) = v;
        }
      }
    }
  } else {
    const float *src = F + x0 + o * g.outer_stride + 4 * cg;
    const uint16_t *src16 = qa.plane + x0 + o * qa.p_outer + 4 * cg;
    const int64_t pst = qa.pst;
    const uint32_t *mapw = qa.map + xt * qa.map_words;  // (IN == kQ16InMixed: bit z = row z of this x-tile is in the plane)
    const pk nlimpk = pk_both(qa.nlim);
    const float flim = (float)qa.nlim + 1.0f;
    // EVERY ROW FROM THE PLANE (round 6) -- the usual case: pass Y left all of this x-tile's slices there.  The tile's map words
    // (NB of them, the same for every tile of an x-tile) are read up front, one wait for all of them; where they are all ones
    // the fill is sixteen plane loads back to back from addresses valid for every thread, as the index form's above.  (Before:
    // one scalar load AND its wait ahead of every single row load, and a branch around each -- sixteen dependent trips to the
    // scalar cache in the fill of a tile.)
    bool all16 = false;
    if constexpr (IN == kQ16InMixed) {
      // (one vector load: lane l takes word min(l, NB - 1) -- NB <= 32 --, in flight together with the run-start words; as
      // scalar loads the compiler waited for every word before it asked for the next)
      const int k = (t & 63) < NB ? (t & 63) : NB - 1;
      const uint32_t w = mapw[k];
      // (rows behind the column's end are nobody's)
      const uint32_t beyond = 32 * k + 32 <= n ? 0u : ~0u << (n - 32 * k);
      all16 = __ballot((w | beyond) != ~0u) == 0ull;
    }
    if (all16) {
      v2u pv[16];
      const uint16_t *srcv = col_ok ? src16 : src16 - 4 * cg;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = RPS * j + r_in;
        pv[j] = EDT_Q16_FILL_LOAD(reinterpret_cast<const v2u *>(srcv + (int64_t)(row < n ? row : n - 1) * pst));
      }
      put_run_starts();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = RPS * j + r_in;
        if (row < nb32) {
          v2u v = col_ok ? pv[j] : (v2u){0u, 0u};
          if (row < n) {
            ov01 |= pk_subs(v[0], nlimpk);
            ov23 |= pk_subs(v[1], nlimpk);
            bad |= !qa.inf_ok && (pk_subs(v[0], 0xFFFEFFFEu) | pk_subs(v[1], 0xFFFEFFFEu)) != 0u;
          } else {
            v = (v2u){~0u, ~0u};
          }
          *reinterpret_cast<v2u *>(img + (row + kPad) * kRowWords + 2 * cg) = v;
        }
      }
    } else {
    put_run_starts();
    // (eight loads per thread in flight; all sixteen of a 512-row tile at once measured no faster -- cfg2 Z 0.237 vs
    // 0.239 ms -- and cost 40-90 VGPRs)
    constexpr int NL = 8;
    for (int i0 = 0; i0 < nb32; i0 += RPS * NL) {
      v4u raw[NL];
      uint32_t in16 = 0;  // bit j: row i0 + 32 j + r_in comes from the 16-bit plane
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int row = i0 + RPS * j + r_in;
        raw[j] = (v4u){0u, 0u, 0u, 0u};
        // (a wave covers 8 consecutive rows: the map word is wave-uniform -- a scalar load, not a vector load the fill
        // would have to wait for: cfg3's Z pass 0.276 -> 0.233 ms)
        bool p16 = false;
        if constexpr (IN == kQ16InMixed) {
          const int mrow = row < n ? row : 0;
          const uint32_t mw = mapw[__builtin_amdgcn_readfirstlane(mrow >> 5)];
          p16 = row < n && ((mw >> (row & 31)) & 1u) != 0u;
        }
        if (p16) {
          in16 |= 1u << j;
          if (col_ok) {
            const v2u v = EDT_Q16_FILL_LOAD(reinterpret_cast<const v2u *>(src16 + (int64_t)row * pst));
            raw[j][0] = v[0];
            raw[j][1] = v[1];
          }
        } else if (row < n && col_ok) {
          raw[j] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + (int64_t)row * st));
        }
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int row = i0 + RPS * j + r_in;
        if (row < nb32 && ((in16 >> j) & 1u)) {
          // (pass Y's limit may be the larger one; a 16-bit value is always within the wide form's range)
          ov01 |= pk_subs(raw[j][0], nlimpk);
          ov23 |= pk_subs(raw[j][1], nlimpk);
          // (no 16-bit-output pass stores 0xFFFF; should a plane -- the caller's, in the sharded path -- hold one where this pass
          // does not carry +inf, the tile has no integer form: handed over, never a silent +inf in the wide fill)
          bad |= !qa.inf_ok && (pk_subs(raw[j][0], 0xFFFEFFFEu) | pk_subs(raw[j][1], 0xFFFEFFFEu)) != 0u;
          *reinterpret_cast<v2u *>(img + (row + kPad) * kRowWords + 2 * cg) = (v2u){raw[j][0], raw[j][1]};
        } else if (row < nb32) {
          uint32_t u[4];
          float err = 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            // N = F / q, exact or the tile does not qualify: F - N * q as one fma (the product is exact)
            const float f = __uint_as_float(raw[j][c]);
            const float tq = fminf(f * qa.rq, flim);
            u[c] = (uint32_t)(tq + 0.5f);
            const float e = fmaf(-(float)u[c], qa.q, f);
            err = fmaxf(err, fabsf(e));  // (+inf in, NaN out: fmaxf keeps the other operand -- caught by the range test)
          }
          // a value beyond nlim: the conversion above was clamped (err says nothing) -- the quad is verified with the wide
          // form's conversion instead (rare: a divergent branch)
          if (max(max(u[0], u[1]), max(u[2], u[3])) > qa.nlim) {
            over = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              ovq |= (u[c] > qa.nlim ? 1u : 0u) << c;
              uint32_t uw;  // (u[c] stays the clamped value: it must not spill into the neighbour's half of the image word)
              bad |= !wide_value(__uint_as_float(raw[j][c]), qa.q, qa.rq, qa.nlimw, qa.fwmax_bits, uw) || (uw == kInfW && !qa.inf_ok);
            }
          } else {
            bad |= !(err == 0.0f);
          }
          v2u v = {u[0] | (u[1] << 16), u[2] | (u[3] << 16)};
          if (row >= n) v = (v2u){~0u, ~0u};
          *reinterpret_cast<v2u *>(img + (row + kPad) * kRowWords + 2 * cg) = v;
        }
      }
    }
    }  // (!all16)
  }
#if EDT_Q16_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  // (every wave publishes its own verdict: no initialisation to order against, and no static LDS -- __syncthreads_or has
  // some, and hipFuncSetAttribute then refuses the full 160 KiB of dynamic LDS)
  ovq |= ((ov01 & 0xFFFFu) ? 1u : 0u) | ((ov01 >> 16) ? 2u : 0u) | ((ov23 & 0xFFFFu) ? 4u : 0u) | ((ov23 >> 16) ? 8u : 0u);
  over |= ovq != 0u;
  if ((t & 63) == 0) flags[t >> 6] = 0u;
  {
    const uint32_t v = (__ballot(bad) != 0ull ? 1u : 0u) | (__ballot(over) != 0ull ? 2u : 0u) | (__ballot(rsany != 0u) != 0ull ? 4u : 0u);
    if (v != 0u && (t & 63) == 0) flags[t >> 6] = v;
  }
  __syncthreads();
  uint32_t verdict = 0;
#pragma unroll
  for (int i = 0; i < T / 64; ++i) verdict |= flags[i];
  const bool has_run_start = (verdict & 4u) != 0u;
  verdict &= 3u;
  if constexpr (S == 1 && !SC) {
    // Round 6: a tile WITHOUT STRUCTURE along the scan axis -- no run start behind row 0 in any of its columns and every row equal
    // to row 0 (the inside of a box: the headline's single-label volume, large objects) -- has nothing for a window to find: every
    // candidate a * d^2 + N[j] carries the column's own value, so result = min(N, the border parabola of the column's ends) (with a
    // black border; without one: N).  Such a tile is answered from its image, row by row by the threads that filled it: no scans,
    // no break bits, no blocks.  The test costs a tile with run starts one OR per run-start word; a tile without any compares its
    // rows once (a cold branch, one more barrier).  debug bit 0x80: no short cut.
    if (verdict == 0u && !has_run_start && !(dbg & 0x80) && !(epi & kEpiSign)) {
      const v2u ref = *reinterpret_cast<const v2u *>(img + kPad * kRowWords + 2 * cg);
      uint32_t diff = 0u;
      for (int row = r_in; row < n; row += RPS) {
        const v2u v = *reinterpret_cast<const v2u *>(img + (row + kPad) * kRowWords + 2 * cg);
        diff |= (v[0] ^ ref[0]) | (v[1] ^ ref[1]);
      }
      if (diff != 0u) atomicOr(ovm + 2, 1u);
      __syncthreads();
      if (ovm[2] == 0u) {
        if constexpr (O16) {
          if (t == 0) atomicOr(qa.map + xt * qa.map_words + (int)(o >> 5), 1u << (o & 31));
        }
        if (col_ok) {
          float *dstF = F + x0 + o * g.outer_stride + 4 * cg;
          uint16_t *dst16 = qa.plane + x0 + o * qa.cd_outer + 4 * cg;
          for (int row = r_in; row < n; row += RPS) {
            pk b = ~0u;
            if constexpr (BB) {
              // (a * min(d, dmax + 1)^2: beyond dmax it is above every value of the tile, as the clamp of the general form)
              uint32_t d = (uint32_t)(row + 1 < n - row ? row + 1 : n - row);
              d = d < qa.dmax + 1u ? d : qa.dmax + 1u;
              const uint32_t c = qa.a * d * d;
              b = pk_both(c < kInf ? c : kInf);
            }
            const v2u r = {pk_min(ref[0], b), pk_min(ref[1], b)};
            if constexpr (O16) {
              *reinterpret_cast<v2u *>(dst16 + (int64_t)row * st) = r;
            } else {
              v4f f = {(float)(r[0] & 0xFFFFu) * qa.q, (float)(r[0] >> 16) * qa.q, (float)(r[1] & 0xFFFFu) * qa.q, (float)(r[1] >> 16) * qa.q};
              if (epi & kEpiSqrt) f = (v4f){sqrtf(f[0]), sqrtf(f[1]), sqrtf(f[2]), sqrtf(f[3])};
              if (epi & kEpiStream) __builtin_nontemporal_store(f, reinterpret_cast<v4f *>(dstF + (int64_t)row * st));
              else *reinterpret_cast<v4f *>(dstF + (int64_t)row * st) = f;
            }
          }
        }
        return;
      }
    }
  }
  if constexpr (!BB && S == 1 && !SC) {
    // Round 6: a tile of NOTHING BUT +inf whose columns hold no run start behind row 0 (no black border: a tile inside one object
    // that spans the volume along the earlier axes) has neither a border nor a finite site -- its results are +inf row for row.
    // Such a tile arrives here as "over" (+inf is beyond every 16-bit limit); before it goes through two 32-bit passes that
    // find nothing to do it is looked at once more -- its inputs again (they are in the L2), its run-start words -- and answered
    // from here.  Only tiles that are over pay for the look (the multi-label configurations: a few per cent of the tiles;
    // tracking it in the fill itself cost every tile of a border-less call 3 %).  debug bit 0x80: no short cut.
    if (verdict == 2u && qa.inf_ok && !(dbg & 0x80) && !(epi & kEpiSign)) {
      uint32_t notinf = 0u;
      for (int u = t; u < NB * 32; u += T) notinf |= (u >> 5) == 0 ? (rsp[u] & ~1u) : rsp[u];  // (row 0 starts a run in every column)
      if (col_ok) {
        if constexpr (IN == kQ16InCodes) {
          const uint16_t *src = qa.codes + x0 + o * qa.cd_outer + 4 * cg;
          for (int row = r_in; row < n && notinf == 0u; row += RPS) {
            const v2u kk = *reinterpret_cast<const v2u *>(src + (int64_t)row * st);
            notinf |= ~(kk[0] & kk[1]);
          }
        } else {
          const float *src = F + x0 + o * g.outer_stride + 4 * cg;
          const uint16_t *src16 = qa.plane + x0 + o * qa.p_outer + 4 * cg;
          for (int row = r_in; row < n && notinf == 0u; row += RPS) {
            bool p16 = false;
            if constexpr (IN == kQ16InMixed) p16 = ((qa.map[xt * qa.map_words + (row >> 5)] >> (row & 31)) & 1u) != 0u;
            if (p16) {
              const v2u v = *reinterpret_cast<const v2u *>(src16 + (int64_t)row * qa.pst);
              notinf |= ~(v[0] & v[1]);  // (+inf: 0xFFFF in a row of the plane)
            } else {
              const v4u v = *reinterpret_cast<const v4u *>(src + (int64_t)row * st);
              notinf |= (v[0] ^ 0x7F7FFFFFu) | (v[1] ^ 0x7F7FFFFFu) | (v[2] ^ 0x7F7FFFFFu) | (v[3] ^ 0x7F7FFFFFu);  // (FLT_MAX in fp32 rows)
            }
          }
        }
      }
      if (notinf != 0u) atomicOr(ovm + 1, 1u);
      __syncthreads();
      if (ovm[1] == 0u) {
        if constexpr (O16) {
          const bool stays = qa.plane_inf_ok != 0u;  // (the indices of pass X, 0xFFFF, ARE the plane's +inf)
          if (t == 0) {
            if (stays) atomicOr(qa.map + xt * qa.map_words + (int)(o >> 5), 1u << (o & 31));
            else atomicAnd(qa.map + xt * qa.map_words + (int)(o >> 5), ~(1u << (o & 31)));
          }
          if (stays) return;
        }
        const float finf = (epi & kEpiToInf) ? INFINITY : FLT_MAX;  // (sqrt of either is itself)
        float *dstF = F + x0 + o * g.outer_stride + 4 * cg;
        if (col_ok)
          for (int row = r_in; row < n; row += RPS) {
            if (epi & kEpiStream) __builtin_nontemporal_store((v4f){finf, finf, finf, finf}, reinterpret_cast<v4f *>(dstF + (int64_t)row * st));
            else *reinterpret_cast<v4f *>(dstF + (int64_t)row * st) = (v4f){finf, finf, finf, finf};
          }
        return;
      }
    }
  }
  // the wide form: every row of every column, fp32 results (16-bit slab records cannot carry them; the stride-2 form keeps
  // the hand-over)
  constexpr bool kWide = S == 1 && !(O16 && SC);
  const bool go_wide = kWide && verdict == 2u && qa.nlimw > qa.nlim;
  if (verdict != 0u && !go_wide) {
    if constexpr (O16 && !SC) {
      // (the map is never zeroed: every tile of the pass says where it left its rows)
      if (t == 0) atomicAnd(qa.map + xt * qa.map_words + (int)(o >> 5), ~(1u << (o & 31)));
    }
    if constexpr (IN == kQ16InMixed) {
      // the fp32 kernel reads F: the rows this tile has in the 16-bit plane become fp32 values there first (exact)
      const uint32_t *mapw = qa.map + xt * qa.map_words;
      float *dstF = F + x0 + o * g.outer_stride + 4 * cg;
      for (int row = r_in; row < n; row += RPS) {
        if (((mapw[row >> 5] >> (row & 31)) & 1u) != 0u && col_ok) {
          const v2u v = *reinterpret_cast<const v2u *>(img + (row + kPad) * kRowWords + 2 * cg);
          // (0xFFFF: +inf -- a tile of nothing but +inf that pass Y left in the plane as the indices it was, round 6: FLT_MAX between
          // the passes, as tofinite leaves it, src/edt.hpp:39-45)
          const uint32_t h[4] = {v[0] & 0xFFFFu, v[0] >> 16, v[1] & 0xFFFFu, v[1] >> 16};
          v4f f;
#pragma unroll
          for (int c = 0; c < 4; ++c) f[c] = h[c] == 0xFFFFu ? FLT_MAX : (float)h[c] * qa.q;
          *reinterpret_cast<v4f *>(dstF + (int64_t)row * st) = f;
        }
      }
    }
    if (t == 0) {
      if (qa.ids == nullptr) {
        atomicAdd(qa.count, 1u);  // (slab records of 16-bit values: the caller counts, nobody can serve the tile)
      } else if (qa.list_cols == 32) {
        const uint32_t idx = atomicAdd(qa.count, 1u);
        qa.ids[idx] = (uint32_t)tile_id;
      } else {
        const uint32_t tx16 = (uint32_t)((g.sx + 15) >> 4), first = oq * tx16 + 2u * (uint32_t)xt;
        const uint32_t k = 2u * (uint32_t)xt + 1u < tx16 ? 2u : 1u;
        const uint32_t idx = atomicAdd(qa.count, k);
        qa.ids[idx] = first;
        if (k == 2u) qa.ids[idx + 1] = first + 1u;
      }
    }
    return;
  }
  // What becomes of a tile with values beyond 16 bits (and nothing worse):
  //   redo      at most 16 of its columns hold such values (the middle of a 512-voxel row: ONE column per tile): those columns
  //             become background in the 16-bit image, the tile is worked on as usual -- columns are independent of one another --
  //             with fp32 results, and only the marked columns are worked on again afterwards, 32-bit lanes on an image of
  //             their own (the same LDS): one wide pass over up to 16 columns instead of two over all of them;
  //   full_wide more columns than that: two wide passes, columns 0..15 and 16..31, the 16-bit form not at all.
  uint32_t redo = 0u;
  bool full_wide = false;
  if constexpr (kWide) {
    if (go_wide) {
      if (ovq != 0u) atomicOr(ovm, ovq << (4 * cg));
      __syncthreads();
      const uint32_t mask = ovm[0];
      if (__builtin_popcount(mask) > 16 || (dbg & 0x40000000)) {
        full_wide = true;
      } else {
        redo = mask;
        // the marked columns leave the 16-bit image: background (columns are independent of one another, and a column of
        // zeros asks for no window -- a column of +inf would: its border parabolas reach a * dmax^2, beyond what the 64-block
        // view of the break bits can call flat)
        for (int row = t; row < n; row += T) {
          uint16_t *r16 = reinterpret_cast<uint16_t *>(img + (row + kPad) * kRowWords);
          for (uint32_t m = mask; m != 0u; m &= m - 1u) r16[__builtin_ctz(m)] = 0u;
        }
        __syncthreads();
      }
    }
  }
  // (O16: the results of a tile with marked columns -- all of its columns -- are fp32 values in F, like a wide tile's)
  const bool to_f32 = !O16 || redo != 0u;
  if (!full_wide) {
    if constexpr (O16 && !SC) {
      if (t == 0) {
        if (to_f32) atomicAnd(qa.map + xt * qa.map_words + (int)(o >> 5), ~(1u << (o & 31)));
        else atomicOr(qa.map + xt * qa.map_words + (int)(o >> 5), 1u << (o & 31));
      }
    }

    // ---- phase 1: run extents across bands (one thread per column and direction), break bits per block and pair ----
    {
      uint16_t *lohi16 = reinterpret_cast<uint16_t *>(lohi);
      if (t < 32) scan_runs_lo(rsp + t, 32, NB, lohi16 + 2 * t, 64);
      else if (t < 64) scan_runs_hi(rsp + (t - 32), 32, NB, n, lohi16 + 2 * (t - 32) + 1, 64);
      const pk apk = pk_both(qa.a);
      for (int u = t; u < 16 * NB; u += T) {
        const int cp = u & 15, band = u >> 4;
        const int valid = n - 32 * band;
        const uint32_t bits = band_breaks(img + (32 * band + kPad) * kRowWords + cp, apk, band == 0, valid < 32 ? valid : 32);
        if (bits) atomicOr(&bm[cp * 6 + 1 + (band >> 3)], bits << (4 * (band & 7)));
      }
    }
    __syncthreads();

    // ---- phase 2: the blocks ----------------------------------------------------------------------
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    const int cp = lane & 15, bq = lane >> 4;
    const bool store_ok = 2 * cp < cols_left;
    typedef float v2f __attribute__((ext_vector_type(2)));
  #pragma unroll 1
    for (int sb = wave; sb * 32 * S < nb32; sb += T / 64) {
      Block L;
      L.img = img;
      L.cp = cp;
      L.p0 = 32 * S * sb + 8 * S * bq;
      // (S = 2 and an odd number of bands: the last two blocks of the last iteration do not exist -- their lanes repeat the
      // column's last block and store nothing, so that the wave-wide exit tests stay what they are)
      const bool lane_on = L.p0 < nb32;
      if (!lane_on) L.p0 = nb32 - 8 * S;
      const int s = L.p0 >> 5;  // the block's band
      L.n = n;
      L.nb32 = nb32;
      {
        const v2u rs2 = *reinterpret_cast<const v2u *>(rsp + s * 32 + 2 * cp);
        const v2u lh2 = *reinterpret_cast<const v2u *>(lohi + s * 32 + 2 * cp);
        L.rswA = rs2[0];
        L.rswB = rs2[1];
        L.loA = (int)(lh2[0] & 0xFFFFu) - 1;
        L.hiA = (int)(lh2[0] >> 16) - 1;
        L.loB = (int)(lh2[1] & 0xFFFFu) - 1;
        L.hiB = (int)(lh2[1] >> 16) - 1;
      }
      L.a = qa.a;
      L.dmax = qa.dmax;
      {
        // the break bits of blocks gi - 32 .. gi + 31 (gi = p0 / 8): bits gi .. gi + 63 of the padded mask
        const int gi = L.p0 >> 3, wi = gi >> 5, sh = gi & 31;
        const uint32_t *m = bm + cp * 6 + wi;
        const uint32_t e0 = m[0], e1 = m[1], e2 = m[2];
        const uint32_t lo = (uint32_t)((((uint64_t)e1 << 32) | e0) >> sh);
        const uint32_t hi = (uint32_t)((((uint64_t)e2 << 32) | e1) >> sh);
        L.win = ((uint64_t)hi << 32) | lo;
        L.bmw = bm + cp * 6;
      }
      pk best[kB];
      block_eval<BB, S>(L, best);
      // ---- results: (float)N * q is exact; sqrt of the last pass (src/edt.hpp:599-601) ----
      float *dst;
      int64_t dstep = st;  // between consecutive EVALUATED rows: S rows of the column, or one row of a compact destination
      if constexpr (SC) {
        const int b = s < BandScatter::kBands ? s : 0;
        dst = scatter->rows[b] + o * scatter->ostride[b] + x0 + 2 * cp - (int64_t)s * 32 * st;
      } else if (S == 2 && qa.compact != nullptr) {  // (wave-uniform: kernel argument)
        dst = qa.compact + x0 + 2 * cp + o * qa.c_outer;
        dstep = qa.c_row2;
      } else {
        dst = F + x0 + o * g.outer_stride + 2 * cp;
        dstep = st * S;
      }
      if (O16 && !to_f32) {
        // (SC: the record's rows are 16-bit here -- the table's pointers and strides count 4-byte words, a row is st / 2 of them)
        auto *ndst = SC ? (__attribute__((address_space(1))) uint32_t *)dst - ((x0 + 2 * cp) >> 1) + (((int64_t)s * 32 * st) >> 1)
                        : (__attribute__((address_space(1))) uint32_t *)(qa.plane + x0 + o * qa.cd_outer + 2 * cp);
  #pragma unroll
        for (int j = 0; j < kB; ++j) {
          const int row = L.p0 + j;
          if (row < n && store_ok && lane_on) ndst[((int64_t)row * st) >> 1] = best[j];  // (st % 4 == 0: the pair is a whole word)
        }
        continue;
      }
      auto *gdst = (__attribute__((address_space(1))) float *)dst;
      const float q = qa.q;
      v2f out[kB];
  #pragma unroll
      for (int j = 0; j < kB; ++j) out[j] = (v2f){(float)(best[j] & 0xFFFFu) * q, (float)(best[j] >> 16) * q};
      if (epi & kEpiSqrt) {
  #pragma unroll
        for (int j = 0; j < kB; ++j) out[j] = (v2f){sqrtf(out[j].x), sqrtf(out[j].y)};
      }
      if (epi & kEpiSign) {
        // the signed transform: a voxel of label 0 gets the negated value (bit k0 + j of the band's foreground words)
        const int k0 = L.p0 & 31;
        v2u fgw = (v2u){~0u, ~0u};
        if (store_ok) fgw = *reinterpret_cast<const v2u *>(qa.signbits + (o * g.nbands + s) * g.sx + x0 + 2 * cp);
        const uint32_t sa = ~fgw[0] >> k0, sb = ~fgw[1] >> k0;
  #pragma unroll
        for (int j = 0; j < kB; ++j)
          out[j] = (v2f){__uint_as_float(__float_as_uint(out[j].x) ^ (((sa >> j) & 1u) << 31)),
                         __uint_as_float(__float_as_uint(out[j].y) ^ (((sb >> j) & 1u) << 31))};
      }
      if (redo == 0u && (epi & kEpiStream)) {
        // (the call's results: streamed -- edt_common.h: kEpiStream)
  #pragma unroll
        for (int j = 0; j < kB; ++j) {
          const int row = L.p0 + S * j;
          if (row < n && store_ok && lane_on)
            __builtin_nontemporal_store(out[j], reinterpret_cast<__attribute__((address_space(1))) v2f *>(gdst + (int64_t)(row / S) * dstep));
        }
      } else if (redo == 0u) {
  #pragma unroll
        for (int j = 0; j < kB; ++j) {
          const int row = L.p0 + S * j;
          if (row < n && store_ok && lane_on)
            *reinterpret_cast<__attribute__((address_space(1))) v2f *>(gdst + (int64_t)(row / S) * dstep) = out[j];
        }
      } else {
        // a marked column is not written here: what is in F there may be the INPUT its wide pass still has to read (the
        // passes that work in place), and its result comes from that pass
        const uint32_t pm = (redo >> (2 * cp)) & 3u;
  #pragma unroll
        for (int j = 0; j < kB; ++j) {
          const int row = L.p0 + S * j;
          if (row < n && store_ok && lane_on) {
            if (!(pm & 1u)) gdst[(int64_t)(row / S) * dstep] = out[j].x;
            if (!(pm & 2u)) gdst[(int64_t)(row / S) * dstep + 1] = out[j].y;
          }
        }
      }
    }

  }  // (!full_wide)

  if constexpr (kWide) {
    if (full_wide || redo != 0u) {
      // ---- wide passes: 16 columns at a time, one 32-bit value per image word, the same lane code (V<true>) ----
      if constexpr (O16) {
        if (full_wide && t == 0) atomicAnd(qa.map + xt * qa.map_words + (int)(o >> 5), ~(1u << (o & 31)));  // (fp32 values in F)
      }
      if (full_wide) {
        uint16_t *lohi16 = reinterpret_cast<uint16_t *>(lohi);
        if (t < 32) scan_runs_lo(rsp + t, 32, NB, lohi16 + 2 * t, 64);
        else if (t < 64) scan_runs_hi(rsp + (t - 32), 32, NB, n, lohi16 + 2 * (t - 32) + 1, 64);
      }
      const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
      const int lane = t & 63;
      constexpr int RPSW = T / 4;  // rows per sweep of the vector fill: 4 threads x 4 columns per row
      const int rw = t >> 2, cgw = t & 3;
      const int npass = full_wide ? (cols_left > 16 ? 2 : 1) : 1;
#pragma unroll 1
      for (int h = 0; h < npass; ++h) {
        __syncthreads();  // (everybody has read the verdict / is done with the image of the pass before)
        if (t < 64) bm[t] = 0u, bm[t + 32] = 0u;
        if (t < 8 * kPad) {
          const int row = t < 4 * kPad ? -kPad + (t >> 2) : nb32 + ((t - 4 * kPad) >> 2);
          *reinterpret_cast<v4u *>(img + (row + kPad) * kRowWords + 4 * (t & 3)) = (v4u){kInfW, kInfW, kInfW, kInfW};
        }
        // image column -> tile column (32: none).  Marked columns alone: each is REPLICATED over rep = 16 / (their number) image
        // columns, and block b of a marked column is worked on in replica b % rep -- a lane per block of ONE image column would
        // put all 64 lanes of a wave on one LDS bank (rows 8 apart are 128 words apart): measured 15 us per tile.
        const int nact = full_wide ? 16 : __builtin_popcount(redo);
        const int rep = full_wide ? 1 : 16 / nact;
        if (t < 16) {
          uint32_t c = 32u;
          if (full_wide) {
            c = 16u * (uint32_t)h + (uint32_t)t;
          } else if (t / rep < nact) {
            uint32_t m = redo;
            for (int i = 0; i < t / rep; ++i) m &= m - 1u;
            c = (uint32_t)__builtin_ctz(m);
          }
          wcol[t] = c < (uint32_t)cols_left ? c : 32u;
        }
        if (full_wide) {
          if constexpr (IN == kQ16InCodes) {
            // the first fill's mapping (whole 64-byte row pieces), by the threads that hold this half's columns (keeping the
            // first fill's registers alive across the verdict instead spills: 46 scratch instructions)
            if ((cg >> 2) == h) {
              const uint16_t *src = qa.codes + x0 + o * qa.cd_outer + 4 * cg;
  #pragma unroll 1
              for (int jb = 0; jb < 16; jb += 8) {  // (eight loads in flight: sixteen tip this cold path into scratch)
                v2u kk[8];
  #pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int row = RPS * (jb + j) + r_in;
                  kk[j] = (v2u){0u, 0u};
                  if (row < n && col_ok) kk[j] = *reinterpret_cast<const v2u *>(src + (int64_t)row * st);
                }
  #pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int row = RPS * (jb + j) + r_in;
                  if (row < nb32) {
                    const uint32_t k0 = kk[j][0] & 0xFFFFu, k1 = kk[j][0] >> 16, k2 = kk[j][1] & 0xFFFFu, k3 = kk[j][1] >> 16;
                    v4u v = (v4u){q16_index_value(k0, qa.ain), q16_index_value(k1, qa.ain), q16_index_value(k2, qa.ain), q16_index_value(k3, qa.ain)};
                    if (row >= n) v = (v4u){kInfW, kInfW, kInfW, kInfW};
                    *reinterpret_cast<v4u *>(img + (row + kPad) * kRowWords + 4 * (cg & 3)) = v;
                  }
                }
              }
            }
          } else {
            const int c0 = 16 * h + 4 * cgw;  // the thread's four columns inside the tile
            const bool cw_ok = c0 < cols_left;
  #pragma unroll 4
            for (int row = rw; row < nb32; row += RPSW) {
              v4u v = (v4u){kInfW, kInfW, kInfW, kInfW};
              if (row < n) {
                v = (v4u){0u, 0u, 0u, 0u};
                if (cw_ok) {
                  bool p16 = false;
                  if constexpr (IN == kQ16InMixed) p16 = ((qa.map[xt * qa.map_words + (row >> 5)] >> (row & 31)) & 1u) != 0u;
                  if (p16) {
                    const v2u pv = *reinterpret_cast<const v2u *>(qa.plane + x0 + o * qa.p_outer + c0 + (int64_t)row * qa.pst);
                    v = (v4u){q16_plane_value(pv[0] & 0xFFFFu), q16_plane_value(pv[0] >> 16), q16_plane_value(pv[1] & 0xFFFFu), q16_plane_value(pv[1] >> 16)};
                  } else {
                    const v4f f = *reinterpret_cast<const v4f *>(F + x0 + o * g.outer_stride + c0 + (int64_t)row * st);
  #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                      uint32_t u;
                      (void)wide_value(f[c], qa.q, qa.rq, qa.nlimw, qa.fwmax_bits, u);  // (verified by the first fill)
                      v[c] = u;
                    }
                  }
                }
              }
              *reinterpret_cast<v4u *>(img + (row + kPad) * kRowWords + 4 * cgw) = v;
            }
          }
        } else {
          // the marked columns, value by value (their lines are in the L2 since the first fill): zeros everywhere first (an
          // image column without a tile column is background), then every thread its rows of every marked column -- all
          // of a thread's loads in flight at once
          for (int row = t; row < nb32; row += T) {
            const uint32_t z = row < n ? 0u : kInfW;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<v4u *>(img + (row + kPad) * kRowWords + 4 * c4) = (v4u){z, z, z, z};
          }
          __syncthreads();  // (wcol; the zeros)
#pragma unroll 4
          for (int j = 0; j < nact; ++j) {
            const uint32_t c = wcol[j * rep];
            if (c >= 32u) continue;
            for (int row = t; row < n; row += T) {
              uint32_t v;
              bool p16 = false;
              if constexpr (IN == kQ16InMixed) p16 = ((qa.map[xt * qa.map_words + (row >> 5)] >> (row & 31)) & 1u) != 0u;
              if constexpr (IN == kQ16InCodes) {
                v = q16_index_value(qa.codes[x0 + o * qa.cd_outer + c + (int64_t)row * st], qa.ain);
              } else if (p16) {
                v = q16_plane_value(qa.plane[x0 + o * qa.p_outer + c + (int64_t)row * qa.pst]);
              } else {
                (void)wide_value(F[x0 + o * g.outer_stride + c + (int64_t)row * st], qa.q, qa.rq, qa.nlimw, qa.fwmax_bits, v);
              }
              for (int r = 0; r < rep; ++r) img[(row + kPad) * kRowWords + j * rep + r] = v;
            }
          }
        }
        __syncthreads();
        for (int u = t; u < 16 * NB; u += T) {
          const int c = u & 15, band = u >> 4;
          const int valid = n - 32 * band;
          const uint32_t bits = band_breaks<true>(img + (32 * band + kPad) * kRowWords + c, qa.a, band == 0, valid < 32 ? valid : 32);
          if (bits) atomicOr(&bm[c * 6 + 1 + (band >> 3)], bits << (4 * (band & 7)));
        }
        __syncthreads();
        // lane -> (active column, block): the blocks of the active columns dealt densely (16 columns: 16 columns x the four
        // blocks of a band per wave, as in the 16-bit form; one marked column: its 64 blocks are ONE iteration of one wave)
        const int nblk = nb32 >> 3;
#pragma unroll 1
        for (int it = wave; it * 64 < nact * nblk; it += T / 64) {
          const int item = it * 64 + lane;
          int blk = item / nact;
          const int jm = item - blk * nact;       // which active column
          const bool lane_on = blk < nblk;
          if (!lane_on) blk = nblk - 1;
          const int cw = jm * rep + blk % rep;    // its image column (replica)
          const uint32_t col = wcol[cw];
          const bool store_ok = lane_on && col < 32u;
          const uint32_t colc = col < 32u ? col : 0u;
          Block L;
          L.img = img;
          L.cp = cw;
          L.p0 = 8 * blk;
          const int s = L.p0 >> 5;
          L.n = n;
          L.nb32 = nb32;
          L.rswA = col < 32u ? rsp[s * 32 + colc] : 0u;
          const uint32_t lh = lohi[s * 32 + colc];
          L.loA = (int)(lh & 0xFFFFu) - 1;
          L.hiA = (int)(lh >> 16) - 1;
          L.rswB = 0u;
          L.loB = L.hiB = 0;
          L.a = qa.a;
          L.dmax = qa.dmaxw;
          {
            const int gi = L.p0 >> 3, wi = gi >> 5, sh = gi & 31;
            const uint32_t *m = bm + cw * 6 + wi;
            const uint32_t e0 = m[0], e1 = m[1], e2 = m[2];
            const uint32_t lo = (uint32_t)((((uint64_t)e1 << 32) | e0) >> sh);
            const uint32_t hi = (uint32_t)((((uint64_t)e2 << 32) | e1) >> sh);
            L.win = ((uint64_t)hi << 32) | lo;
            L.reach = flat_reach_full(bm + cw * 6, gi);
            L.bmw = bm + cw * 6;
          }
          pk best[kB];
          block_eval<BB, 1, true>(L, best);
          float *dst;
          if constexpr (SC) {
            const int b = s < BandScatter::kBands ? s : 0;
            dst = scatter->rows[b] + o * scatter->ostride[b] + x0 + colc - (int64_t)s * 32 * st;
          } else {
            dst = F + x0 + o * g.outer_stride + colc;
          }
          auto *gdst = (__attribute__((address_space(1))) float *)dst;
          // (+inf: "no boundary anywhere" -- FLT_MAX between the passes, +INF behind the last one: tofinite / toinfinite,
          // src/edt.hpp:39-53)
          const float finf = (epi & kEpiToInf) ? INFINITY : FLT_MAX;
          float out[kB];
#pragma unroll
          for (int j = 0; j < kB; ++j) out[j] = best[j] >= kInfW ? finf : (float)best[j] * qa.q;
          if (epi & kEpiSqrt) {
#pragma unroll
            for (int j = 0; j < kB; ++j) out[j] = sqrtf(out[j]);
          }
          if (epi & kEpiSign) {
            uint32_t fgw1 = ~0u;
            if (store_ok) fgw1 = qa.signbits[(o * g.nbands + s) * g.sx + x0 + colc];
            const uint32_t sw = ~fgw1 >> (L.p0 & 31);
#pragma unroll
            for (int j = 0; j < kB; ++j) out[j] = __uint_as_float(__float_as_uint(out[j]) ^ (((sw >> j) & 1u) << 31));
          }
#pragma unroll
          for (int j = 0; j < kB; ++j) {
            const int row = L.p0 + j;
            if (row < n && store_ok) gdst[(int64_t)row * st] = out[j];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
bool column_pass_q16_supported(const AxisGeom &g) {
  // whole 16-byte granules per row piece; axes the LDS image fits twice per CU (two workgroups per CU keep the fill of
  // one under the windows of the other); shorter axes than 4 bands leave most of the workgroup without a band
  return g.sx % 4 == 0 && g.stride % 4 == 0 && g.outer_stride % 4 == 0 && g.nbands >= 4 && g.nbands <= 32 &&
         !(debug_mode() & 0x8000000);
}

template <bool BB, int IN, bool O16, bool SC, int T, int S = 1>
static int launch_q16_kt(float *F, const uint32_t *rs, const AxisGeom &g, const Q16Args &qa, int epi, hipStream_t stream,
                         const BandScatter *scatter) {
  const int NB = (int)g.nbands;
  // (EDT_Q16_EXTRA_LDS: experiments -- bytes of LDS asked for on top, i.e. fewer workgroups per CU)
  static const size_t extra_lds = [] { const char *e = getenv("EDT_Q16_EXTRA_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
  const size_t lds = (size_t)q16_lds_words(NB) * sizeof(uint32_t) + extra_lds;
  const int64_t tiles_x = ceil_div(g.sx, 32);
  int64_t tiles = tiles_x * g.nouter;
  if (tiles <= 0) return EDT_OK;
  if (!(debug_mode() & 0x800)) tiles = tiles_x * (ceil_div(g.nouter, 8) * 8);
  if (tiles > 0x7FFFFFFF) { set_error("too many tiles"); return EDT_ERR_UNSUPPORTED; }
  static std::atomic<uint64_t> attr_done{0};
  EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_column_pass_q16<BB, IN, O16, SC, T, S>)));
  hipLaunchKernelGGL((k_column_pass_q16<BB, IN, O16, SC, T, S>), dim3((unsigned)tiles), dim3(T), lds, stream, F, rs, g,
                     (int)tiles_x, epi, debug_mode(), qa, scatter);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

template <bool BB, int IN, bool O16, bool SC>
static int launch_q16_k(float *F, const uint32_t *rs, const AxisGeom &g, const Q16Args &qa, int epi, hipStream_t stream,
                        const BandScatter *scatter) {
  if (g.nbands > 16) return launch_q16_kt<BB, IN, O16, SC, 512>(F, rs, g, qa, epi, stream, scatter);
  return launch_q16_kt<BB, IN, O16, SC, 256>(F, rs, g, qa, epi, stream, scatter);
}

template <bool BB>
static int launch_q16_b(float *F, const uint32_t *rs, const AxisGeom &g, const Q16Args &qa, int in, bool o16, int epi,
                        hipStream_t stream, const BandScatter *scatter, int out_stride) {
  if (out_stride == 2) {
    if (scatter != nullptr || o16 || in == kQ16InMixed || (in == kQ16InCodes && qa.compact == nullptr)) {
      set_error("internal: output stride 2 takes fp32 values (in place or compact) or indices (compact)");
      return EDT_ERR_BAD_ARG;
    }
    if (in == kQ16InCodes) {
      if (g.nbands > 16) return launch_q16_kt<BB, kQ16InCodes, false, false, 512, 2>(F, rs, g, qa, epi, stream, nullptr);
      return launch_q16_kt<BB, kQ16InCodes, false, false, 256, 2>(F, rs, g, qa, epi, stream, nullptr);
    }
    if (g.nbands > 16) return launch_q16_kt<BB, kQ16InF32, false, false, 512, 2>(F, rs, g, qa, epi, stream, nullptr);
    return launch_q16_kt<BB, kQ16InF32, false, false, 256, 2>(F, rs, g, qa, epi, stream, nullptr);
  }
  if (scatter != nullptr) {
    if (in == kQ16InMixed || (o16 && in != kQ16InCodes)) { set_error("internal: slab records take indices or fp32 rows"); return EDT_ERR_BAD_ARG; }
    if (o16) return launch_q16_k<BB, kQ16InCodes, true, true>(F, rs, g, qa, epi, stream, scatter);  // 16-bit records
    return in == kQ16InCodes ? launch_q16_k<BB, kQ16InCodes, false, true>(F, rs, g, qa, epi, stream, scatter)
                             : launch_q16_k<BB, kQ16InF32, false, true>(F, rs, g, qa, epi, stream, scatter);
  }
  if (in == kQ16InCodes)
    return o16 ? launch_q16_k<BB, kQ16InCodes, true, false>(F, rs, g, qa, epi, stream, nullptr)
               : launch_q16_k<BB, kQ16InCodes, false, false>(F, rs, g, qa, epi, stream, nullptr);
  if (o16) { set_error("internal: 16-bit plane without the index form"); return EDT_ERR_BAD_ARG; }
  return in == kQ16InMixed ? launch_q16_k<BB, kQ16InMixed, false, false>(F, rs, g, qa, epi, stream, nullptr)
                           : launch_q16_k<BB, kQ16InF32, false, false>(F, rs, g, qa, epi, stream, nullptr);
}

// a: c_d = a * d^2 quanta of this pass; ain: quanta per squared index of pass X (codes != nullptr).
// plane / map != nullptr: with codes -- the results go to the 16-bit plane (= codes, in place) and the tile's bit is set in
// map; without -- the rows are taken from the plane wherever map says so (the pass after such a pass).
// Slab records of 16-bit values (edt_shard_api.hip): with codes, a scatter table AND a plane (any non-null value) the results go to
// the table's destinations as 16-bit rows, refused tiles are only counted (ids == nullptr); without codes, a map of ones
// and plane_stride > 0 every row is read from the plane at its own strides (16-bit elements) and F is only written.
int launch_column_pass_q16(float *F, const uint16_t *codes, const uint32_t *rs, const AxisGeom &g, float q, uint32_t a,
                           uint32_t ain, int bb, int epi, uint32_t *count, uint32_t *ids, hipStream_t stream,
                           const BandScatter *scatter, uint16_t *plane, uint32_t *map, int map_words, const ColumnOut *out,
                           int64_t plane_stride, int64_t plane_outer, int plane_inf_ok, const uint32_t *signbits,
                           int64_t codes_outer) {
  Q16Args qa;
  qa.cd_outer = codes_outer > 0 ? codes_outer : g.outer_stride;
  qa.plane_inf_ok = plane_inf_ok ? 1u : 0u;
  qa.signbits = signbits;
  if ((epi & kEpiSign) && (signbits == nullptr || scatter != nullptr || (out && (out->stride == 2 || out->compact != nullptr)) ||
                           (codes != nullptr && plane != nullptr))) {
    set_error("internal: the sign epilogue belongs to a last pass with fp32 results in place");
    return EDT_ERR_BAD_ARG;
  }
  qa.codes = codes;
  qa.q = q;
  qa.rq = 1.0f / q;
  qa.a = a;
  qa.ain = ain;
  qa.dmax = edt_q16::q16_dmax(a);
  qa.nlim = a * qa.dmax * qa.dmax;
  uint32_t kmax = 0;
  while ((uint64_t)(kmax + 1) * (kmax + 1) * ain <= qa.nlim && kmax < 65534u) ++kmax;
  qa.kmax = kmax;
  {
    // the wide form's range (nlimw == nlim: there is none)
    edt_q16::WideRange wr = {0u, 0u, false};
    if (!(debug_mode() & 0x20000000)) wr = edt_q16::q16_wide_range(a, q, g.n, bb != 0, qa.nlim);
    qa.dmaxw = wr.nlim ? wr.dmax : qa.dmax;
    qa.nlimw = wr.nlim ? wr.nlim : qa.nlim;
    qa.inf_ok = wr.inf ? 1u : 0u;
    uint32_t kw = kmax;
    while ((uint64_t)(kw + 1) * (kw + 1) * ain <= qa.nlimw && kw < 65534u) ++kw;
    qa.kmaxw = kw;
    const float fw = (float)qa.nlimw * q;  // exact: nlimw * odd(q) < 2^24 (nlim: < 2^16 * 255)
    memcpy(&qa.fwmax_bits, &fw, sizeof(fw));
  }
  qa.count = count;
  qa.ids = ids;
  qa.list_cols = g.nbands > 16 ? 16 : 32;  // (edt_colwave_lane.h: TileGeom -- 16-column tiles for the 1- and 2-column waves)
  qa.plane = plane;
  qa.map = map;
  qa.map_words = map_words;
  qa.pst = plane_stride > 0 ? plane_stride : g.stride;
  qa.p_outer = plane_stride > 0 ? plane_outer : g.outer_stride;
  qa.compact = out ? out->compact : nullptr;
  qa.c_outer = out ? out->outer : 0;
  qa.c_row2 = out ? out->row2 : 0;
  const int ostride = (out && (out->stride == 2 || out->compact != nullptr)) ? 2 : 1;
  if (qa.compact != nullptr && ((reinterpret_cast<uintptr_t>(qa.compact) % 8) != 0 || (qa.c_outer % 2) != 0 || (qa.c_row2 % 2) != 0)) {
    set_error("internal: compact destination of the integer kernel must take 8-byte stores");
    return EDT_ERR_BAD_ARG;
  }
  const int in = codes ? kQ16InCodes : (plane ? kQ16InMixed : kQ16InF32);
  const bool o16 = codes != nullptr && plane != nullptr;
  return bb ? launch_q16_b<true>(F, rs, g, qa, in, o16, epi, stream, scatter, ostride)
            : launch_q16_b<false>(F, rs, g, qa, in, o16, epi, stream, scatter, ostride);
}

// the largest value (in quanta) a tile of a pass with c_d = a * d^2 over columns of n rows may hold without being handed to the
// fp32 kernel, whatever else it holds (without a black border: +inf as well -- 0 where the pass does not carry that)
uint32_t q16_value_limit(float q, uint32_t a, int64_t n, int bb) {
  const uint32_t d16 = edt_q16::q16_dmax(a), n16 = a * d16 * d16;
  edt_q16::WideRange wr = {0u, 0u, false};
  if (!(debug_mode() & 0x20000000)) wr = edt_q16::q16_wide_range(a, q, n, bb != 0, n16);
  if (!bb) return wr.inf ? wr.nlim : 0u;
  return wr.nlim ? wr.nlim : n16;
}

// When can the integer kernel refuse no tile at all?  In the index form the values are integers by construction (k^2 * ax
// quanta; 0xFFFF = no boundary in the row = +inf, which the wide form carries where the columns are short enough -- round 5),
// and so are the integer kernel's own results.  With a black border every row has a boundary on both sides: an index is at
// most ceil(sx / 2), and no pass raises a value.  Without, a run may touch one edge of the volume -- an index is at most sx
// -- and a row that was +inf after pass X leaves pass Y with a border parabola or a sum N[j] + ay * d^2: at most sy^2 * ay
// more.  If that bound is within the range of the pass (its 16-bit form, or the wide form: q16_value_limit) and everything
// the pass reads was written by pass X or by an integer pass that could not refuse either (the caller's part), the fp32
// launch over the hand-over list has nothing to do.  (debug bit 0x20000000: never proven.)
// tests/test_q16_logic.py plays passes Y and Z of whole volumes through the lane logic and holds this proof against them.
bool q16_no_refusals(float q, const uint32_t *a, int axis, int64_t sx, int64_t sy, int64_t n, int bb) {
  if (debug_mode() & 0x20000000) return false;
  if (axis != 1 && axis != 2) return false;
  // (the proof stands on its own: extents an index of pass X cannot describe -- 0xFFFF is "no boundary" -- prove nothing, and
  // the products below stay far from 2^64: kmax, sy < 2^16, a <= 16384)
  if (sx < 1 || sy < 1 || n < 1 || sx > 65534 || sy > 65534 || n > 65534 || a[0] > 16384u || a[1] > 16384u || a[2] > 16384u) return false;
  const uint64_t kmax = bb ? (uint64_t)((sx + 1) / 2) : (uint64_t)sx;
  uint64_t vmax = kmax * kmax * a[0];
  if (axis == 2 && !bb) vmax += (uint64_t)sy * (uint64_t)sy * a[1];
  return vmax <= q16_value_limit(q, a[axis], n, bb);
}

// the quantum of a call (edt_colq16_lane.h: quantum_of), host side
bool q16_quantum(const float *w, int naxes, float *q, uint32_t *a) {
  const edt_q16::Quantum Q = edt_q16::quantum_of(w, naxes);
  if (!Q.ok) return false;
  *q = Q.q;
  for (int i = 0; i < 3; ++i) a[i] = Q.a[i];
  return true;
}

}  // namespace edt_amd
