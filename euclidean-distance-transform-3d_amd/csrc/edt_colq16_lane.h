// edt_colq16_lane.h -- per-lane logic of the 16-bit integer column pass (passes Y and Z; edt_colq16.hip).
//
// Where the voxel sizes share a quantum q -- w_i^2 = a_i * q with small integers a_i: (1,1,1), (6,6,30), (4,4,40),
// (0.5,0.5,1) ... -- every value the column passes meet is an integer multiple N * q: pass X leaves (k*wx)^2 = k^2 * ax * q,
// a candidate of the envelope is c_d + F[j] = (a * d^2 + N[j]) * q, the border parabolas are a * d^2 * q.  As long as
// N < 2^16 and N * (odd part of q) < 2^24 all of these are exact in fp32 AND in the reference's fp64 intermediates
// (src/edt.hpp:181, :230, :258, :307: w2 * sq(d) + ff[j], rounded once to fp32), so the reference's result is the exact
// integer minimum times q and the whole pass can run on 16-bit integers, TWO ADJACENT COLUMNS PER LANE in packed
// instructions (v_pk_min_u16, v_pk_add_u16 clamp: 3 instructions per step for two voxels against 2.2 per voxel in the
// fp32 form, edt_colwave_lane.h), on an LDS tile of half the size.  Tiles that do not qualify (a value that is not a
// multiple of q, or too large even for the wide form V<true> below -- one column per lane, 32-bit values, +inf for rows
// without any boundary) are handed to the fp32 kernel through a list (edt_colq16.hip).
//
// The mathematics is that of the windowed path (edt_colwave_lane.h, "brute"):
//     result[p] = min( B_p, min_{1<=d<=R} ( c_d + min(N[p-d], N[p+d]) ) ),   B_p = min(N[p], border parabolas),
// for every R with c_{R+1} >= B_p, with no label test inside the window; a block whose neighbourhood has no "break"
// (|N[r] - N[r-1]| > a) within reach skips the window.  Differences: the border distances are packed counters, the breaks
// are kept per BLOCK of 8 rows and per column PAIR (conservative: fewer skips, never a wrong one) and count every link,
// also those across a run boundary (where a window is needed anyway), and the 64 blocks a wave works on at a time are
// CONTIGUOUS (16 column pairs x the four blocks of one 32-row band).
//
// Compiled twice like edt_colwave_lane.h: by hipcc into the kernel and by g++ into tests/q16_emul.cpp, which plays every
// lane on the host against the oracle (tests/test_q16_logic.py).
#pragma once

#include <stdint.h>
#include <string.h>
#include <math.h>

#ifndef EDT_LANE
#error "define EDT_LANE (function qualifiers) before including edt_colq16_lane.h"
#endif
#ifndef EDT_HOSTFN
#define EDT_HOSTFN static inline  // host-side helpers (the quantum of a call)
#endif

namespace edt_q16 {

typedef uint32_t pk;  // two unsigned 16-bit values: low half = the even column of the pair, high half = the odd one

#ifndef EDT_Q16_K
#define EDT_Q16_K 16  // (16: the image of a 512-row axis with its planes is 39.3 KiB, four workgroups per CU; 32: 41.4 KiB, three;
                      //  8 with the rolled loop from step 9 on: cfg3 0.72 -> 0.76 ms)
#endif
constexpr int kK = EDT_Q16_K;     // register-resident radius of the window (compile-time steps); further steps: rolled loop
constexpr int kPad = kK;          // rows of +inf (0xFFFF) before row 0 and after the last band of the image
constexpr int kRowWords = 16;     // 32-bit words per image row (32 columns x 16 bit)
constexpr int kB = 8;             // rows per block
#ifndef EDT_Q16_REFRESH
#define EDT_Q16_REFRESH 8  // (measured: 2 -> +2.9 %, 8 -> -1.3 % of a cfg3 step against 4)
#endif
constexpr int kRefresh = EDT_Q16_REFRESH;  // the exit bound is refreshed from the current minima every so many steps (2 or 4)
constexpr uint32_t kInf = 0xFFFFu;
constexpr uint32_t kFar = 0x4000u;  // "no border on this side" distance (stays below 2^15 after n <= 2048 increments)

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned short q16_us2 __attribute__((ext_vector_type(2)));
typedef short q16_s2 __attribute__((ext_vector_type(2)));
EDT_LANE q16_us2 q16_v(pk x) { return __builtin_bit_cast(q16_us2, x); }
EDT_LANE pk q16_p(q16_us2 v) { return __builtin_bit_cast(pk, v); }
EDT_LANE pk pk_min(pk a, pk b) { return q16_p(__builtin_elementwise_min(q16_v(a), q16_v(b))); }
EDT_LANE pk pk_max(pk a, pk b) { return q16_p(__builtin_elementwise_max(q16_v(a), q16_v(b))); }
EDT_LANE pk pk_adds(pk a, pk b) { return q16_p(__builtin_elementwise_add_sat(q16_v(a), q16_v(b))); }
EDT_LANE pk pk_subs(pk a, pk b) { return q16_p(__builtin_elementwise_sub_sat(q16_v(a), q16_v(b))); }
EDT_LANE pk pk_add(pk a, pk b) { return q16_p(q16_v(a) + q16_v(b)); }
EDT_LANE pk pk_mul(pk a, pk b) { return q16_p(q16_v(a) * q16_v(b)); }
template <int S>
EDT_LANE pk pk_shl(pk a) { return q16_p(q16_v(a) << (unsigned short)S); }
EDT_LANE pk pk_sar15(pk a) { return __builtin_bit_cast(pk, __builtin_bit_cast(q16_s2, a) >> (short)15); }
EDT_LANE int q16_clz(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int q16_ctz(uint32_t v) { return __builtin_ctz(v); }
#define EDT_Q16_ANY(cond) (__ballot(cond) != 0ull)
// wave-wide minimum / maximum of a small non-negative integer, as a wave-uniform value (the wide form's window limits: rare path)
EDT_LANE int q16_wave_min(int v) {
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o < v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
EDT_LANE int q16_wave_max(int v) {
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o > v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
#define EDT_Q16_UNROLL _Pragma("unroll")
#define EDT_Q16_ROLLED _Pragma("unroll 1")
// (keeps the compiler from re-deriving a select mask as two 16-bit compares, two selects and a byte permute)
#define EDT_Q16_OPAQUE(x) asm volatile("" : "+v"(x))
#else
EDT_LANE uint32_t q16_sat(uint32_t v) { return v > 0xFFFFu ? 0xFFFFu : v; }
EDT_LANE pk q16_mk(uint32_t lo, uint32_t hi) { return (lo & 0xFFFFu) | (hi << 16); }
EDT_LANE pk pk_min(pk a, pk b) {
  const uint32_t al = a & 0xFFFFu, bl = b & 0xFFFFu, ah = a >> 16, bh = b >> 16;
  return q16_mk(al < bl ? al : bl, ah < bh ? ah : bh);
}
EDT_LANE pk pk_max(pk a, pk b) {
  const uint32_t al = a & 0xFFFFu, bl = b & 0xFFFFu, ah = a >> 16, bh = b >> 16;
  return q16_mk(al > bl ? al : bl, ah > bh ? ah : bh);
}
EDT_LANE pk pk_adds(pk a, pk b) { return q16_mk(q16_sat((a & 0xFFFFu) + (b & 0xFFFFu)), q16_sat((a >> 16) + (b >> 16))); }
EDT_LANE pk pk_subs(pk a, pk b) {
  const uint32_t al = a & 0xFFFFu, bl = b & 0xFFFFu, ah = a >> 16, bh = b >> 16;
  return q16_mk(al > bl ? al - bl : 0u, ah > bh ? ah - bh : 0u);
}
EDT_LANE pk pk_add(pk a, pk b) { return q16_mk((a & 0xFFFFu) + (b & 0xFFFFu), ((a >> 16) + (b >> 16)) & 0xFFFFu); }
EDT_LANE pk pk_mul(pk a, pk b) { return q16_mk((a & 0xFFFFu) * (b & 0xFFFFu), ((a >> 16) * (b >> 16)) & 0xFFFFu); }
template <int S>
EDT_LANE pk pk_shl(pk a) { return q16_mk((a & 0xFFFFu) << S, ((a >> 16) << S) & 0xFFFFu); }
EDT_LANE pk pk_sar15(pk a) { return q16_mk((a & 0x8000u) ? 0xFFFFu : 0u, (a & 0x80000000u) ? 0xFFFFu : 0u); }
EDT_LANE int q16_clz(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int q16_ctz(uint32_t v) { return __builtin_ctz(v); }
#define EDT_Q16_ANY(cond) (cond)
EDT_LANE int q16_wave_min(int v) { return v; }  // (the emulation plays every lane on its own)
EDT_LANE int q16_wave_max(int v) { return v; }
#define EDT_Q16_UNROLL
#define EDT_Q16_ROLLED
#define EDT_Q16_OPAQUE(x) ((void)0)
#endif

EDT_LANE pk pk_both(uint32_t v) { return (v & 0xFFFFu) * 0x10001u; }
EDT_LANE pk pk_sel(pk mask, pk a, pk b) { return (a & mask) | (b & ~mask); }  // v_bfi_b32

// ---------------------------------------------------------------------------------------
// What a 32-bit word of the image holds.  V<false>: two adjacent columns as 16-bit values (everything above).  V<true>,
// the WIDE form: ONE column as a 32-bit value -- the same lane code on tiles that hold values beyond 16 bits (objects
// deeper than ~255 voxels; the middle of a 512-voxel row: k = 256, N = 2^16), which the kernel then works on as two
// half-tiles of 16 columns, one after the other, on the same LDS image (edt_colq16.hip).  Exact by the same argument while
// N * odd(q) < 2^24; +inf is kInfW, every sum stays below 2^31 (no saturation needed).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kInfW = 0x3FFFFFFFu;
template <bool W>
struct V;
template <>
struct V<false> {
  static constexpr uint32_t kInfWord = 0xFFFFFFFFu;  // +inf in (both halves of) an image word
  EDT_LANE_MEMBER static pk vmin(pk a, pk b) { return pk_min(a, b); }
  EDT_LANE_MEMBER static pk vmax(pk a, pk b) { return pk_max(a, b); }
  EDT_LANE_MEMBER static pk adds(pk a, pk b) { return pk_adds(a, b); }
  EDT_LANE_MEMBER static pk subs(pk a, pk b) { return pk_subs(a, b); }
  EDT_LANE_MEMBER static pk add(pk a, pk b) { return pk_add(a, b); }
  EDT_LANE_MEMBER static pk mul(pk a, pk b) { return pk_mul(a, b); }
  EDT_LANE_MEMBER static pk both(uint32_t v) { return pk_both(v); }
  // a wave-uniform candidate offset c (may exceed the range: +inf then)
  EDT_LANE_MEMBER static pk cval(uint64_t c) { return pk_both(c < kInf ? (uint32_t)c : kInf); }
  template <int J>
  EDT_LANE_MEMBER static pk bitmask(pk starts) { return pk_sar15(pk_shl<15 - J>(starts)); }  // bit J of either half -> that half all ones
  // per-column quantities of the pair (A: the even column, B: the odd one)
  EDT_LANE_MEMBER static pk cols(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }
};
template <>
struct V<true> {
  static constexpr uint32_t kInfWord = kInfW;
  EDT_LANE_MEMBER static pk vmin(pk a, pk b) { return a < b ? a : b; }
  EDT_LANE_MEMBER static pk vmax(pk a, pk b) { return a > b ? a : b; }
  EDT_LANE_MEMBER static pk adds(pk a, pk b) { return a + b; }  // (operands below 2^30)
  EDT_LANE_MEMBER static pk subs(pk a, pk b) { return a > b ? a - b : 0u; }
  EDT_LANE_MEMBER static pk add(pk a, pk b) { return a + b; }
  EDT_LANE_MEMBER static pk mul(pk a, pk b) { return a * b; }
  EDT_LANE_MEMBER static pk both(uint32_t v) { return v; }
  EDT_LANE_MEMBER static pk cval(uint64_t c) { return c < kInfW ? (uint32_t)c : kInfW; }
  template <int J>
  EDT_LANE_MEMBER static pk bitmask(pk starts) { return 0u - ((starts >> J) & 1u); }
  EDT_LANE_MEMBER static pk cols(uint32_t a, uint32_t) { return a; }
};

// ---------------------------------------------------------------------------------------
// The quantum of a call (host): w_i^2 = a_i * q for every axis of the call, q = odd * 2^e with odd <= 255, so that
// N * q is exact in fp32 for every N < 2^16.  ok = false: no such quantum (the fp32 kernels keep the call).
// ---------------------------------------------------------------------------------------
struct Quantum {
  bool ok;
  float q;         // the quantum
  uint32_t a[3];   // w_i^2 / q  (1 for an unused axis)
};

EDT_HOSTFN uint64_t q16_gcd(uint64_t a, uint64_t b) {
  while (b) { const uint64_t t = a % b; a = b; b = t; }
  return a;
}

EDT_HOSTFN Quantum quantum_of(const float *w, int naxes) {
  Quantum Q;
  Q.ok = false;
  Q.q = 1.0f;
  Q.a[0] = Q.a[1] = Q.a[2] = 1u;
  uint64_t m[3];
  int e[3];
  int emin = 1 << 30;
  for (int i = 0; i < naxes; ++i) {
    const float w2 = w[i] * w[i];  // the reference's fp32 product (src/edt.hpp:181, :258)
    if (!(w2 >= 1.17549435e-38f) || !(w2 < 1.0e30f)) return Q;
    if ((double)w2 != (double)w[i] * (double)w[i]) return Q;  // w^2 must be exact: pass X leaves (k*w)^2 = k^2 * w^2
    int ex;
    const double fr = frexp((double)w2, &ex);      // w2 = fr * 2^ex, fr in [0.5, 1)
    uint64_t mi = (uint64_t)ldexp(fr, 24);         // 24-bit integer mantissa
    int ei = ex - 24;
    while (!(mi & 1u)) { mi >>= 1; ++ei; }
    m[i] = mi;
    e[i] = ei;
    if (ei < emin) emin = ei;
  }
  uint64_t g = 0;
  for (int i = 0; i < naxes; ++i) {
    if (e[i] - emin > 20) return Q;
    m[i] <<= (e[i] - emin);
    if (m[i] >= (1ull << 40)) return Q;
    g = q16_gcd(g, m[i]);
  }
  // q = g * 2^emin; its odd part must leave room for a 16-bit factor in an fp32 significand
  uint64_t odd = g;
  int e2 = emin;
  while (!(odd & 1u)) { odd >>= 1; ++e2; }
  if (odd > 255u) return Q;
  if (e2 < -140 || e2 > 100) return Q;
  for (int i = 0; i < naxes; ++i) {
    const uint64_t ai = m[i] / g;
    if (ai == 0 || ai > 16384u) return Q;  // (a window needs at least two rows: a * 2^2 < 2^16)
    Q.a[i] = (uint32_t)ai;
  }
  Q.q = (float)ldexp((double)odd, e2);
  Q.ok = true;
  return Q;
}

// largest d with a * d^2 <= 65534 (0xFFFF is +inf)
EDT_HOSTFN uint32_t q16_dmax(uint32_t a) {
  uint32_t d = 1;
  while ((uint64_t)a * (d + 1) * (d + 1) <= 65534u) ++d;
  return d;
}

// wide form: the largest N a tile may hold -- N * odd(q) < 2^24 (exact in fp32), N <= a * d^2 for a d <= 2047 (the axis
// has at most 1024 rows: the clamp of the border distance changes no minimum) -- and that d
EDT_HOSTFN uint32_t q16_odd_of(float q) {
  int ex;
  const double fr = frexp((double)q, &ex);
  uint64_t m = (uint64_t)ldexp(fr, 24);
  while (m && !(m & 1u)) m >>= 1;
  return (uint32_t)m;
}
EDT_HOSTFN uint32_t q16_dmax_wide(uint32_t a, float q) {
  const uint64_t cap = ((1ull << 24) - 1) / q16_odd_of(q);
  uint32_t d = 1;
  while (d < 2047u && (uint64_t)a * (d + 1) * (d + 1) <= cap) ++d;
  return (uint64_t)a * d * d <= cap ? d : 0u;  // 0: not even one row (the wide form does not apply)
}

// The wide form's range for one pass: values up to nlim, border distances up to dmax (nlim == 0: no wide form beyond the 16-bit
// one, whose limit is nlim16).  Without a black border a tile may hold +inf -- rows without any boundary so far -- and such a
// row's result is a border parabola or a sum N[j] + a * d^2 that no value of its own bounds: +inf is carried (inf) only where
// every distance of the column is within dmax (n <= dmax: no clamp of a border distance, no window constant beyond the range)
// and the largest such sum, nlim + a * n^2, is still exact in fp32 -- for which nlim is lowered as far as that takes.
struct WideRange { uint32_t dmax, nlim; bool inf; };
EDT_HOSTFN WideRange q16_wide_range(uint32_t a, float q, int64_t n, bool bb, uint32_t nlim16) {
  WideRange R = {0u, 0u, false};
  const uint32_t dw = q16_dmax_wide(a, q);
  const uint64_t cap = ((1ull << 24) - 1) / q16_odd_of(q);
  uint64_t nw = (uint64_t)a * dw * dw;
  if (nw <= nlim16) return R;
  if (!bb && n <= (int64_t)dw) {
    const uint64_t an2 = (uint64_t)a * (uint64_t)n * (uint64_t)n;
    if (an2 < cap && cap - an2 > nlim16) {
      nw = nw < cap - an2 ? nw : cap - an2;
      R.inf = true;
    }
  }
  R.dmax = dw;
  R.nlim = (uint32_t)nw;
  return R;
}

// N = f / q as a 32-bit integer (kInfW for FLT_MAX), exact or not at all (wide form; also the verdict on a value the 16-bit conversion of
// edt_colq16.hip had to clamp): false unless f == N * q for an N <= nlimw.  fwmax_bits: bit pattern of (float)nlimw * q
// (non-negative floats order like their bit patterns; negative values, NaN and +inf lie above every one of them).
// f * rq is within 2 of N (rq and the product are rounded); the remainder f - u0 * q is a small multiple of q, exact as
// ONE fma, and puts that right.
EDT_LANE bool wide_value(float f, float q, float rq, uint32_t nlimw, uint32_t fwmax_bits, uint32_t &u) {
  uint32_t fb;
  memcpy(&fb, &f, sizeof(fb));
  u = 0u;
  // FLT_MAX: "no boundary along the earlier axes" (tofinite, src/edt.hpp:39-45) -- +inf here: such a site never wins, a voxel
  // that sees nothing else comes out as FLT_MAX again / as +INF behind the last pass (toinfinite, :47-53)
  if (fb == 0x7F7FFFFFu) { u = kInfW; return true; }
  if (fb > fwmax_bits) return false;
  const float u0 = rintf(f * rq);
  const float r = fmaf(-u0, q, f);
  const float un = u0 + rintf(r * rq);
  const float e = fmaf(-un, q, f);
  u = (uint32_t)un;
  return e == 0.0f && u <= nlimw;
}

// ---------------------------------------------------------------------------------------
// Breaks of one band of a column pair: bit k of the result = some link r-1 -> r with r in block k of the band
// (rows 8k .. 8k+7) has |N[r] - N[r-1]| > a in either column.  rows[0] = the row before the band, rows[1..32] the band.
// (|x - y| > a  <=>  x > y + a or y > x + a: one saturating add per row, shared by the two links the row is part of)
// ---------------------------------------------------------------------------------------
// band0: word of (first row of the band, pair).  top: the band is the column's first (no link into its first row).
// valid: rows of the band that are rows of the column (32 but for the last band; the +inf rows after them are no links).
template <bool W = false>
EDT_LANE uint32_t band_breaks(const uint32_t *band0, pk apk, bool top, int valid) {
  uint32_t bits = 0;
  pk y = top ? band0[0] : band0[-kRowWords];
  pk ty = V<W>::adds(y, apk);
  EDT_Q16_UNROLL
  for (int k = 0; k < 4; ++k) {
    pk acc = 0;
    EDT_Q16_UNROLL
    for (int j = 0; j < 8; ++j) {
      pk x = band0[(8 * k + j) * kRowWords];
      if (valid < 32 && 8 * k + j >= valid) x = y;
      const pk tx = V<W>::adds(x, apk);
      acc |= V<W>::subs(x, ty) | V<W>::subs(y, tx);
      y = x;
      ty = tx;
    }
    bits |= acc ? (1u << k) : 0u;
  }
  return bits;
}

// Flat reach of block gi of a column pair (in rows): no break lies within this distance of the block's rows.
// bm: the break bits of the pair's blocks gi-32 .. gi+31 (bit 32 = the block itself).
// S = 2: the block is 16 rows, i.e. the blocks gi and gi + 1 of the break bits.
template <int S = 1>
EDT_LANE int flat_reach(uint64_t win) {
  if ((win >> 32) & (S == 2 ? 3u : 1u)) return 0;
  const uint32_t below = (uint32_t)win, above = (uint32_t)(win >> (32 + S));
  // nearest break block below: gi - 32 + hb  ->  its last row is at least 8 * (32 - hb) - 7 rows below the block
  const int dlo = below ? 8 * (q16_clz(below) + 1) - 7 : 8 * 33 - 7;
  const int dhi = above ? 8 * q16_ctz(above) : 8 * 31;
  return dlo < dhi ? dlo : dhi;
}

// The same over the WHOLE column (the wide form: its windows may be hundreds of rows long, and a flat column -- the middle
// of a single-label volume -- must not pay for them): m = the six mask words of the column (word 0 and 5 zero, block g = bit
// g of words 1..4), gi = the block.  No break anywhere: 1 << 20.
EDT_LANE int flat_reach_full(const uint32_t *m, int gi) {
  const uint64_t lo64 = ((uint64_t)m[2] << 32) | m[1], hi64 = ((uint64_t)m[4] << 32) | m[3];  // blocks 0..63, 64..127
  const int gb = gi & 63;
  const uint64_t own = gi < 64 ? lo64 : hi64;
  if ((own >> gb) & 1u) return 0;
  const uint64_t lt = gb ? own & (~0ull >> (64 - gb)) : 0ull, gt = gb < 63 ? own & (~0ull << (gb + 1)) : 0ull;
  int below = -1, above = -1;  // nearest break block on either side
  if (lt) below = (gi & 64) + 63 - __builtin_clzll(lt);
  else if (gi >= 64 && lo64) below = 63 - __builtin_clzll(lo64);
  if (gt) above = (gi & 64) + __builtin_ctzll(gt);
  else if (gi < 64 && hi64) above = 64 + __builtin_ctzll(hi64);
  const int dlo = below >= 0 ? 8 * (gi - below) - 7 : (1 << 20);
  const int dhi = above >= 0 ? 8 * (above - gi - 1) : (1 << 20);
  return dlo < dhi ? dlo : dhi;
}

// ---------------------------------------------------------------------------------------
// One block of a lane: rows p0 .. p0+7 of the column pair cp.
// ---------------------------------------------------------------------------------------
struct Block {
  const uint32_t *img;   // the image: word of (row r, pair cp) = img[(r + kPad) * kRowWords + cp]
  int cp;                // column pair inside the tile (0..15)
  int p0;                // first row of the block (a multiple of 8)
  int n;                 // rows of the column
  int nb32;              // rows of the image without the padding (whole bands)
  uint32_t rswA, rswB;   // run-start words of the band, even / odd column
  int loA, loB;          // last run start in an earlier band (-1: none), even / odd column
  int hiA, hiB;          // row before the first run start in a later band (n-1: none)
  uint32_t a;            // c_d = a * d^2
  uint32_t dmax;         // q16_dmax(a)
  uint64_t win;          // break bits around the block (flat_reach)
  int reach;             // wide form: the flat reach over the whole column (flat_reach_full); unused otherwise
  const uint32_t *bmw;   // the six mask words of the block's column pair / (wide form) image column
  // wide form, set by block_eval (wave-uniform): windows of +inf rows.  dskip: every step d <= dskip looks at +inf rows only,
  // on both sides, in every lane of the wave (the wave's blocks lie inside stretches of rows without any boundary: such a
  // window STARTS at the stretch's end instead of walking there); dend: beyond this distance no lane of the wave has a
  // finite row left in its column (the window ENDS there: c_d = +inf) -- round 6, ADVICE r5: a column with ONE finite
  // row ran its windows to sqrt(d0^2 + N / a) steps, hundreds, for nothing
  mutable int dskip, dend;
};

// (wide form) the rows of a column that may hold finite values: all of them lie in [jfirst, jlast] (jfirst > jlast: none).
// m: the six mask words of the image column (block g = bit g of words 1..4); first / last: the column's rows 0 and n - 1.
// A stretch of finite rows inside +inf rows has a break at either end (|N - inf| > a); one that touches an end of the column
// has none there, which the end rows themselves tell.
EDT_LANE void finite_extent(const uint32_t *m, uint32_t first, uint32_t last, int n, int &jfirst, int &jlast) {
  const uint64_t lo64 = ((uint64_t)m[2] << 32) | m[1], hi64 = ((uint64_t)m[4] << 32) | m[3];
  const bool any = (lo64 | hi64) != 0ull;
  const int fb = lo64 ? __builtin_ctzll(lo64) : (hi64 ? 64 + __builtin_ctzll(hi64) : 0);
  const int lb = hi64 ? 127 - __builtin_clzll(hi64) : (lo64 ? 63 - __builtin_clzll(lo64) : 0);
  jfirst = first < kInfW ? 0 : (any ? 8 * fb : n);
  jlast = last < kInfW ? n - 1 : (any ? 8 * lb + 6 : -1);
  if (jlast > n - 1) jlast = n - 1;
}

// S = output stride: 1 = every row of the block is evaluated; 2 = a block is 16 rows of which the even ones are evaluated (the
// doubled grids of the voxel-graph transform, whose odd rows are never read again) -- every row is a candidate either way.
template <bool BB, int S = 1, bool W = false>
struct Steps {
  typedef V<W> X;
  static constexpr int K = kK, B = kB, RW = kRowWords, NR = S * B;  // NR rows per block
  const Block &L;
  pk (&w)[NR + 2 * K];
  pk (&best)[B];
  const uint32_t *PB;  // word of (row p0 - K, pair cp)
  pk bmax;             // upper bound of the current minima of the block (both halves)
  uint32_t a;

  EDT_LANE_MEMBER void refresh_bound() {
    pk m = best[0];
    EDT_Q16_UNROLL
    for (int i = 1; i < B; ++i) m = X::vmax(m, best[i]);
    bmax = m;
  }
  // c_d as a (packed) constant (wave-uniform: scalar arithmetic), +inf once it leaves the range
  EDT_LANE_MEMBER pk cpk(int d) const {
    // (wide form: beyond the column's length -- beyond the last finite row of any of the wave's columns: L.dend -- there are
    // only +inf rows: c_d = +inf there ends a window whose minima are +inf themselves, rows without any boundary, which no
    // finite c_d ever reaches; as a scalar select, not as the loop's bound: a variable bound tips the kernel into scratch)
    if constexpr (W) return d > L.dend ? kInfW : X::cval((uint64_t)a * (uint32_t)(d * d));
    const uint32_t c = a * (uint32_t)(d * d);
    return pk_both(c < kInf ? c : kInf);
  }

  template <int D>
  EDT_LANE_MEMBER void run() {
    if constexpr (D < K) {
      if constexpr (D > 1 && (D - 1) % kRefresh == 0) refresh_bound();
      const pk c1 = cpk(D), c2 = cpk(D + 1);
      // a candidate at distance d is at least c_d: once c_d >= every current minimum of the wave nothing further away
      // can lower any of them
      if (!EDT_Q16_ANY(X::subs(bmax, c1) != 0u)) return;
      w[K - D] = PB[(K - D) * RW];
      w[K + NR - 1 + D] = PB[(K + NR - 1 + D) * RW];
      w[K - D - 1] = PB[(K - D - 1) * RW];
      w[K + NR + D] = PB[(K + NR + D) * RW];
      EDT_Q16_UNROLL
      for (int ii = 0; ii < B; ++ii) {
        // (the first and the last row of the block need the rows just requested: they come last)
        const int i = ii < B - 2 ? ii + 1 : (ii == B - 2 ? 0 : B - 1);
        const pk m1 = X::vmin(w[K + S * i - D], w[K + S * i + D]);
        const pk m2 = X::vmin(w[K + S * i - D - 1], w[K + S * i + D + 1]);
        best[i] = X::vmin(X::vmin(best[i], X::adds(m1, c1)), X::adds(m2, c2));
      }
      run<D + 2>();
    } else if constexpr (S == 2) {
      // Windows beyond the register-resident part, blocks of 16 rows with the even ones evaluated: at step d output i looks
      // at the rows p0+2i-d (entered at step d-2i) and p0+2i+d (entered at step d-15+2i) -- the last 16 entries of either
      // side: rings of 16 registers indexed by d mod 16, one step per exit test.
      constexpr int R = 16;
      static_assert((K % R) == 0 && NR == R, "ring phase / size");
      pk rlo[R], rhi[R];
      EDT_Q16_UNROLL
      for (int s = K - R + 2; s <= K; ++s) {  // the last 15 rows of the register-resident window on either side
        rlo[s % R] = w[K - s];
        rhi[s % R] = w[K + NR - 1 + s];
      }
      const uint32_t *slo = L.img, *shi = L.img;
      for (int d0 = K + 1; d0 < 4096; d0 += R) {
        bool done = false;
        EDT_Q16_UNROLL
        for (int e = 0; e < R; ++e) {  // step d = d0 + e;  d mod R == (1 + e) mod R
          const int d = d0 + e;
          if (e % kRefresh == 0) refresh_bound();
          const pk c1 = cpk(d);
          if (!EDT_Q16_ANY(X::subs(bmax, c1) != 0u)) { done = true; break; }
          if (e % 8 == 0) {
            int rl = L.p0 - d - 7, rh = L.p0 + NR - 1 + d;  // the rows of the next eight steps: rl .. rl+7, rh .. rh+7
            rl = rl < -kPad ? -kPad : rl;
            rh = rh > L.nb32 + kPad - 8 ? L.nb32 + kPad - 8 : rh;
            slo = L.img + (rl + kPad) * RW + L.cp;
            shi = L.img + (rh + kPad) * RW + L.cp;
          }
          const int sl = (1 + e) % R;
          rlo[sl] = slo[(7 - e % 8) * RW];
          rhi[sl] = shi[(e % 8) * RW];
          EDT_Q16_UNROLL
          for (int i = 0; i < B; ++i) {
            const pk m = X::vmin(rlo[(sl - 2 * i + 2 * R) % R], rhi[(sl - (NR - 1) + 2 * i + 2 * R) % R]);
            best[i] = X::vmin(best[i], X::adds(m, c1));
          }
        }
        if (done) break;
      }
    } else {
      // Windows beyond the register-resident part: the same step as a rolled loop.  At step d row i looks at the rows
      // p0+i-d and p0+i+d, i.e. at the B rows that entered the window most recently on either side: rings of 16 registers
      // indexed by d mod 16, static once the loop body covers 16 consecutive steps.  Eight consecutive steps read eight
      // consecutive rows on either side through one address; a stretch beyond the image is moved onto +inf rows.
      constexpr int R = 16;
      static_assert((K % R) == 0 && B < R, "ring phase / size");
      pk rlo[R], rhi[R];
      EDT_Q16_UNROLL
      for (int s = K - B + 2; s <= K; ++s) {
        rlo[s % R] = w[K - s];
        rhi[s % R] = w[K + B - 1 + s];
      }
      const uint32_t *slo = L.img, *shi = L.img;
      int dstart = K + 1;
      if constexpr (W) {
        // the wave's blocks lie inside stretches of +inf rows: every step up to L.dskip would look at +inf rows on both sides
        // in every lane -- the window starts at the last ring phase before the stretch ends, its rings holding what those
        // steps would have left there: +inf (wave-uniform branch)
        if (L.dskip > K + R) {
          dstart = K + 1 + ((L.dskip - K) & ~(R - 1));
          EDT_Q16_UNROLL
          for (int i = 0; i < R; ++i) rlo[i] = rhi[i] = kInfW;
        }
      }
      for (int d0 = dstart; d0 < 4096; d0 += R) {
        bool done = false;
        EDT_Q16_UNROLL
        for (int e = 0; e < R; e += 2) {  // steps d, d + 1 with d = d0 + e;  d mod R == (1 + e) mod R
          const int d = d0 + e;
          if (e % kRefresh == 0) refresh_bound();
          const pk c1 = cpk(d), c2 = cpk(d + 1);
          if (!EDT_Q16_ANY(X::subs(bmax, c1) != 0u)) { done = true; break; }
          if (e % 8 == 0) {
            int rl = L.p0 - d - 7, rh = L.p0 + B - 1 + d;  // the rows of the next eight steps: rl .. rl+7, rh .. rh+7
            rl = rl < -kPad ? -kPad : rl;
            rh = rh > L.nb32 + kPad - 8 ? L.nb32 + kPad - 8 : rh;
            slo = L.img + (rl + kPad) * RW + L.cp;
            shi = L.img + (rh + kPad) * RW + L.cp;
          }
          const int s1 = (1 + e) % R, s2 = (2 + e) % R;
          rlo[s1] = slo[(7 - e % 8) * RW];
          rhi[s1] = shi[(e % 8) * RW];
          rlo[s2] = slo[(6 - e % 8) * RW];
          rhi[s2] = shi[(e % 8 + 1) * RW];
          EDT_Q16_UNROLL
          for (int i = 0; i < B; ++i) {
            // row p0+i-d entered at step d-i, row p0+i+d at step d-(B-1-i)
            const pk m1 = X::vmin(rlo[(s1 - i + R) % R], rhi[(s1 - (B - 1 - i) + R) % R]);
            const pk m2 = X::vmin(rlo[(s2 - i + R) % R], rhi[(s2 - (B - 1 - i) + R) % R]);
            best[i] = X::vmin(X::vmin(best[i], X::adds(m1, c1)), X::adds(m2, c2));
          }
        }
        if (done) break;
      }
    }
  }
};

// distance of the row before the block to the row before ITS run (kFar: that run has no border below), one column
template <bool BB>
EDT_LANE uint32_t dist_below(uint32_t rsw, int lo_in, int row0, int k0) {
  const uint32_t lowm = k0 > 0 ? rsw & (0xFFFFFFFFu >> (32 - k0)) : 0u;  // run starts at rows < k0 of this band
  const int s = lowm ? row0 + 31 - q16_clz(lowm) : lo_in;               // first row of the run of row p0 - 1
  return (BB || s > 0) ? (uint32_t)(row0 + k0 - s) : kFar;
}
// distance of the row after the block to the first row of the next run (kFar: no border above), one column
template <bool BB>
EDT_LANE uint32_t dist_above(uint32_t rsw, int hi_out, int row0, int k0, int n, int nr = kB) {
  const uint32_t m = k0 + nr < 32 ? rsw & (0xFFFFFFFFu << (k0 + nr)) : 0u;
  const int e = m ? row0 + q16_ctz(m) : hi_out + 1;
  return (BB || e < n) ? (uint32_t)(e - (row0 + k0 + nr)) : kFar;
}

// best[i] = result of row p0 + S * i (both columns), in quanta.  S = 2: the block is the 16 rows p0 .. p0 + 15 (p0 a multiple
// of 16), its even rows are evaluated.
template <bool BB, int S = 1, bool W = false>
EDT_LANE void block_eval(const Block &L, pk (&best)[kB]) {
  typedef V<W> X;
  constexpr int K = kK, B = kB, RW = kRowWords, NR = S * B;
  const int row0 = L.p0 & ~31, k0 = L.p0 & 31;
  const uint32_t *PB = L.img + (L.p0 - K + kPad) * RW + L.cp;
  pk w[NR + 2 * K];
  EDT_Q16_UNROLL
  for (int j = 0; j < NR; ++j) w[K + j] = PB[(K + j) * RW];
  // ---- B_p: border distances as packed counters from run start to run start ----
  constexpr uint32_t kRowBits = S == 2 ? 0xFFFFu : 0xFFu;
  // (wide form: one column per lane -- the B members of the block are not looked at)
  const pk starts = X::cols((L.rswA >> k0) & kRowBits, W ? 0u : (L.rswB >> k0) & kRowBits);  // bit j of a half: a run starts at row p0 + j
  pk dl = X::cols(dist_below<BB>(L.rswA, L.loA, row0, k0), W ? 0u : dist_below<BB>(L.rswB, L.loB, row0, k0));
  // (a block that reaches beyond the column's last row has a "negative" distance above: the counters are 16-bit modular,
  // the rows of the column come out right and the others are not rows)
  pk dr = X::cols(dist_above<BB>(L.rswA, L.hiA, row0, k0, L.n, NR), W ? 0u : dist_above<BB>(L.rswB, L.hiB, row0, k0, L.n, NR));
  const pk one = X::both(1u);
  // a run that starts at row 0 of the column has a border below it only with black_border
  const pk first0 = (BB || L.p0 > 0) ? one : X::both(kFar);
  const pk dmaxpk = X::both(L.dmax), apk = X::both(L.a);
  pk bmax = 0;
  // Two wave-uniform short cuts (round 5; the general form below costs ~100 of the ~250 instructions a block takes before
  // its window).  No block of the wave holds a run start: the border distances are linear in the row, dl + j + 1 below
  // and dr + NR - j above.  And if no border is nearer than dmax to any of those rows either, the border parabolas are at
  // least a * dmax^2 >= every value of the tile: B_p = N.  (A block that reaches beyond the column's last row has a
  // modular "negative" dr: it takes the linear form, whose additions put that right.)
  const bool has_start = EDT_Q16_ANY(starts != 0u);
  if (!has_start) {
    const bool ends_here = L.p0 + NR > L.n;
    if (!EDT_Q16_ANY(ends_here || (X::subs(dmaxpk, dl) | X::subs(dmaxpk, dr)) != 0u)) {
      EDT_Q16_UNROLL
      for (int i = 0; i < B; ++i) best[i] = w[K + S * i];
    } else {
      EDT_Q16_UNROLL
      for (int j = 0; j < NR; j += S) {
        const pk dnear = X::vmin(X::add(dl, X::both((uint32_t)(j + 1))), X::add(dr, X::both((uint32_t)(NR - j))));
        const pk dm = X::vmin(dnear, dmaxpk);
        pk bord = X::mul(X::mul(dm, dm), apk);
        if constexpr (W && !BB) bord = dnear >= kFar ? kInfW : bord;  // (see the general form below)
        best[j / S] = X::vmin(w[K + j], bord);
      }
    }
  } else {
    pk mask[NR], dlv[B];
    {
  #define EDT_Q16_ROW_UP(J)                                                   \
      mask[J] = X::template bitmask<J>(starts);                               \
      EDT_Q16_OPAQUE(mask[J]);                                                \
      dl = pk_sel(mask[J], J == 0 ? first0 : one, X::add(dl, one));           \
      if ((J) % S == 0) dlv[(J) / S] = dl;
      EDT_Q16_ROW_UP(0) EDT_Q16_ROW_UP(1) EDT_Q16_ROW_UP(2) EDT_Q16_ROW_UP(3)
      EDT_Q16_ROW_UP(4) EDT_Q16_ROW_UP(5) EDT_Q16_ROW_UP(6) EDT_Q16_ROW_UP(7)
      if constexpr (S == 2) {
        EDT_Q16_ROW_UP(8) EDT_Q16_ROW_UP(9) EDT_Q16_ROW_UP(10) EDT_Q16_ROW_UP(11)
        EDT_Q16_ROW_UP(12) EDT_Q16_ROW_UP(13) EDT_Q16_ROW_UP(14) EDT_Q16_ROW_UP(15)
      }
  #undef EDT_Q16_ROW_UP
    }
    EDT_Q16_UNROLL
    for (int j = NR - 1; j >= 0; --j) {
      dr = X::add(dr, one);
      if (j % S == 0) {
        const pk dnear = X::vmin(dlv[j / S], dr);
        const pk dm = X::vmin(dnear, dmaxpk);
        // a * min(d, dmax)^2 fits the range; a tile on this path holds no value above a * dmax^2, so the clamp changes no minimum
        pk bord = X::mul(X::mul(dm, dm), apk);
        // (wide form, no black border: a run without a border on either side has no border parabola at all -- its rows may be
        // +inf, "no boundary along the earlier axes", and must stay so; the 16-bit form never sees such a tile)
        if constexpr (W && !BB) bord = dnear >= kFar ? kInfW : bord;
        best[j / S] = X::vmin(w[K + j], bord);
      }
      dr = dr & ~mask[j];  // (a set bit is a real row: the border site of the rows below it)
    }
  }
  // rows that complete the last band are not rows of the column
  if (L.p0 + NR > L.n) {
    EDT_Q16_UNROLL
    for (int i = 0; i < B; ++i)
      if (L.p0 + S * i >= L.n) best[i] = 0u;
  }
  EDT_Q16_UNROLL
  for (int i = 0; i < B; ++i) bmax = X::vmax(bmax, best[i]);
  // ---- flat neighbourhood: nothing within reach can improve any row of the wave's blocks ----
  {
    const int r1 = W ? L.reach : flat_reach<S>(L.win);
    uint32_t D1 = (uint32_t)r1 + 1u;
    D1 = D1 < L.dmax + 1u ? D1 : L.dmax + 1u;
    const uint64_t cD = (uint64_t)L.a * D1 * D1;
    // (wide form: a column without any break holds one value in all its rows -- nothing can improve anything, whatever the
    // minima are; this is what keeps a column of +inf rows, "no boundary anywhere", from running a window to the column's end)
    const bool flat_column = W && r1 >= (1 << 20);
    if (!EDT_Q16_ANY(!flat_column && X::subs(bmax, X::cval(cD)) != 0u)) return;
    if constexpr (!W && S == 1) {
      // The 64-block view of the break bits ends 248 rows away.  A lane that saw no break in it looks at the whole column
      // before the wave runs hundreds of steps over a flat one (round 5: columns of 65025 = 255^2 next to the middle of a
      // 512-voxel row -- 255 steps for nothing; values that large did not reach this kernel's windows before).
      if (EDT_Q16_ANY(r1 >= 8 * 31)) {
        const int r2 = r1 >= 8 * 31 ? flat_reach_full(L.bmw, L.p0 >> 3) : r1;
        uint32_t D2 = (uint32_t)(r2 < (1 << 20) ? r2 : (1 << 20)) + 1u;
        D2 = D2 < L.dmax + 1u ? D2 : L.dmax + 1u;
        if (!EDT_Q16_ANY(X::subs(bmax, X::cval((uint64_t)L.a * D2 * D2)) != 0u)) return;
      }
    }
  }
  L.dskip = 0;
  L.dend = L.n;
  if constexpr (W) {
    // Windows over +inf rows (round 6).  A block without a break whose first row is +inf holds +inf rows only, and so do the
    // rows within its flat reach (a finite neighbour of a +inf row is always a break): the steps up to the smallest such
    // reach of the wave are skipped.  And no window needs to look further than the finite rows of its column reach.
    const int r1 = L.reach;
    const bool own_inf = r1 > 0 && w[K] >= kInfW;
    L.dskip = q16_wave_min(own_inf ? (r1 < 4096 ? r1 : 4096) : 0);
    int jf, jl;
    finite_extent(L.bmw, L.img[kPad * RW + L.cp], L.img[(L.n - 1 + kPad) * RW + L.cp], L.n, jf, jl);
    int lane_end = 0;
    if (jf <= jl) {
      const int e1 = L.p0 + NR - 1 - jf, e2 = jl - L.p0;
      lane_end = e1 > e2 ? e1 : e2;
      lane_end = lane_end > 0 ? lane_end : 0;
    }
    const int de = q16_wave_max(lane_end);
    L.dend = de < L.n ? de : L.n;
  }
  Steps<BB, S, W> steps{L, w, best, PB, bmax, L.a};
  steps.template run<1>();
}

// ---------------------------------------------------------------------------------------
// Run structure across the bands of one column (one thread per column and direction walks the band words):
// lo[b] = last run start in a band before b (-1: none), hi[b] = row before the first run start in a band after b
// (n - 1: none).  Stored as lo + 1 / hi + 1 in 16-bit planes.
// ---------------------------------------------------------------------------------------
EDT_LANE void scan_runs_lo(const uint32_t *rs_col, int stride, int NB, uint16_t *lo_col, int ostride) {
  int last = -1;
  for (int b = 0; b < NB; ++b) {
    lo_col[b * ostride] = (uint16_t)(last + 1);
    const uint32_t r = rs_col[b * stride];
    if (r) last = 32 * b + 31 - q16_clz(r);
  }
}
EDT_LANE void scan_runs_hi(const uint32_t *rs_col, int stride, int NB, int n, uint16_t *hi_col, int ostride) {
  int first = n;
  for (int b = NB - 1; b >= 0; --b) {
    hi_col[b * ostride] = (uint16_t)first;  // (= hi_out + 1)
    const uint32_t r = rs_col[b * stride];
    if (r) first = 32 * b + q16_ctz(r);
  }
}

}  // namespace edt_q16
