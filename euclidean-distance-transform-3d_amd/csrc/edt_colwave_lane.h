// edt_colwave_lane.h -- per-lane logic of the wave-autonomous column pass (passes 2 and 3).
//
// The column pass of edt_colwave.hip gives every wavefront a private set of whole columns:
//   lane = (column c, band b), a band = 32 consecutive rows = one bit-word of the run masks;
//   a 64-lane wave therefore owns CW = 64/NBP columns x NBP bands (NBP = bands per column,
//   a power of two), and the three phases below only ever talk to lanes of the SAME wave
//   (through LDS), so no workgroup barrier separates them.
//
// This header holds the three phases as plain per-lane functions so that the very same
// source is compiled twice: by hipcc into the kernel, and by g++ into a lane-by-lane host
// emulation that tests/test_lane_logic.py checks against the CPU oracle without a GPU
// (the emulation is a test fixture; it is never shipped or called by the library).
//
// Algorithm (see also edt_tiled.hip, which uses the same mathematics with workgroup-wide
// phases): the lower envelope of the parabolas  w2*(p-j)^2 + F[j]  over one label run is the
// lower convex hull of the points (j, F[j] + w2*j^2); a hull is a SUBSET of the rows and is
// stored as one `alive` bit per row.
//   phase 1: every lane builds the hull of its own 32 rows (monotone chain, rows in VGPRs);
//   phase 2: log2(NBP) rounds of pairwise hull merges across band-group boundaries (bridge
//            walk), only where a label run crosses the boundary;
//   phase 3: every lane sweeps its 32 rows over the merged hull and evaluates the
//            reference's own expression fl32(w2*(p-j)^2 + F[j]) (src/edt.hpp:230, :307),
//            the border parabolas (src/edt.hpp:233-242, :310-311) and the fused
//            toinfinite / sqrt epilogue (src/edt.hpp:47-53, :599-601).
//
// Arithmetic notes (bit parity with the reference):
//   * w2 is the fp32 product w*w widened to fp64 (src/edt.hpp:181, :258);
//   * w2 * d^2 with d < 2^14 is EXACT in fp64 (24 x 28 significant bits), so
//     fma(w2, d^2, F) == w2*d^2 + F bit for bit: one rounding either way;
//   * hull orientation tests compare crossing abscissae by cross-multiplication, no divide.
#pragma once

#include <math.h>
#include <stdint.h>

#ifndef EDT_LANE
#error "define EDT_LANE (function qualifiers) before including edt_colwave_lane.h"
#endif

namespace edt_lane {

constexpr int kTileCols = 32;  // columns of a workgroup tile = floats per LDS tile row

#if defined(__HIP_DEVICE_COMPILE__)
EDT_LANE int mul24(int a, int b) { return __mul24(a, b); }
EDT_LANE int clz32(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int ctz32(uint32_t v) { return __builtin_ctz(v); }
EDT_LANE double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }
#else
EDT_LANE int mul24(int a, int b) { return a * b; }
EDT_LANE int clz32(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int ctz32(uint32_t v) { return __builtin_ctz(v); }
EDT_LANE double fma64(double a, double b, double c) { return fma(a, b, c); }
#endif

enum : int { kLaneEpiToInf = 1, kLaneEpiSqrt = 2 };

// LDS addressing of one lane.  The fp32 tile is row-major [row][32]; the 16-byte granule
// that holds the lane's column is XOR-rotated by its band so that the 32 lanes of a
// half-wave (32/CW bands x CW columns) read 32 different banks when each reads "its" row.
// The bit planes ([band][32] words) use the same rotation.
template <int CW>
EDT_LANE int addr_swz(int band) { return ((band * ((CW / 4) & 7)) & 7) << 2; }
template <int CW>
EDT_LANE int addr_tile(int colc, int row) { return (row << 5) + (colc ^ addr_swz<CW>(row >> 5)); }
template <int CW>
EDT_LANE int addr_word(int colc, int band) { return (band << 5) + (colc ^ addr_swz<CW>(band)); }

struct Lane {
  float *tile;           // LDS fp32 tile of the workgroup
  uint32_t *alive;       // LDS [NBP][32]: hull vertices
  const uint32_t *rsp;   // LDS [NBP][32]: run starts
  int colc;              // column inside the workgroup tile (0..31)
  int band;              // this lane's band
  int row0;              // band * 32
  int n;                 // rows of the scan axis
  double w2;
  uint32_t nzw, rsw;     // foreground / run-start bits of this lane's 32 rows
};

// exact (double)(d*d) for |d| < 4096
EDT_LANE double sq_i(int d) { return (double)mul24(d, d); }

// value of parabola j (height Fj) at row p -- the reference's output expression
EDT_LANE double para(int p, int j, double Fj, double w2) { return fma64(w2, sq_i(p - j), Fj); }

// numerator of the crossing abscissa of parabolas p < q:  (Fq - Fp) + w2*(q-p)*(q+p)
// (src/edt.hpp:206-208: ff[i] - ff[v[k]] + factor1 * factor2; the product is exact)
EDT_LANE double edge_num(int p, double Fp, int q, double Fq, double w2) {
  return fma64(w2, (double)mul24(q - p, q + p), Fq - Fp);
}

template <int CW>
EDT_LANE double ldF(const Lane &L, int row) { return (double)L.tile[addr_tile<CW>(L.colc, row)]; }

// Highest set bit p with lo <= p < from in this lane's column of a bit plane.  -1 if none.
template <int CW>
EDT_LANE int prev_set(const uint32_t *plane, int colc, int from, int lo) {
  if (from <= lo) return -1;
  int wi = (from - 1) >> 5;
  const int wlo = lo >> 5;
  uint32_t m = plane[addr_word<CW>(colc, wi)] & (0xFFFFFFFFu >> (31 - ((from - 1) & 31)));
  while (true) {
    if (wi == wlo) m &= 0xFFFFFFFFu << (lo & 31);
    if (m) return wi * 32 + 31 - clz32(m);
    if (wi == wlo) return -1;
    --wi;
    m = plane[addr_word<CW>(colc, wi)];
  }
}

// Lowest set bit p with after < p <= hi.  -1 if none.
template <int CW>
EDT_LANE int next_set(const uint32_t *plane, int colc, int after, int hi) {
  if (after >= hi) return -1;
  int wi = (after + 1) >> 5;
  const int whi = hi >> 5;
  uint32_t m = plane[addr_word<CW>(colc, wi)] & (0xFFFFFFFFu << ((after + 1) & 31));
  while (true) {
    if (wi == whi) m &= 0xFFFFFFFFu >> (31 - (hi & 31));
    if (m) return wi * 32 + ctz32(m);
    if (wi == whi) return -1;
    ++wi;
    m = plane[addr_word<CW>(colc, wi)];
  }
}

// ---------------------------------------------------------------------------------------
// phase 1: hull of the lane's own band.  f[r] = F(row0 + r) (registers).  Returns the alive
// word.  The stack of the monotone chain IS the alive word; its top two entries (ia, ib) and
// the numerator / width of the edge between them are cached in registers.
//   pop while  s(ib,row) <= s(ia,ib)  <=>  num(ib,row)*(ib-ia) <= num(ia,ib)*(row-ib)
//   (src/edt.hpp:210, :287).  A stack with fewer than two entries has nab = -inf.
// ---------------------------------------------------------------------------------------
template <int CW>
EDT_LANE uint32_t phase1_hull(const Lane &L, const float *f) {
  const uint32_t rs1 = L.rsw | 1u;  // the band's first row starts a (local) chain
  const double w2 = L.w2;
  uint32_t aw = 0, seg = 0xFFFFFFFFu;
  int ia = L.row0, ib = L.row0;
  double Fa = 0.0, Fb = 0.0, nab = -INFINITY, dab = 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 32; ++r) {
    if ((L.nzw >> r) & 1u) {
      const int row = L.row0 + r;
      const double Fi = (double)f[r];
      const bool fresh = (rs1 >> r) & 1u;
      if (fresh) {
        nab = -INFINITY;
        seg = 0xFFFFFFFFu << r;
      }
      double nbi = edge_num(ib, Fb, row, Fi, w2);
      double dbi = (double)(row - ib);
      while (nbi * dab <= nab * dbi) {  // the top vertex lies on or above the chord
        aw &= ~(1u << (ib - L.row0));
        ib = ia;
        Fb = Fa;
        nbi = edge_num(ib, Fb, row, Fi, w2);
        dbi = (double)(row - ib);
        const uint32_t below = aw & seg & ((1u << (ib - L.row0)) - 1u);
        if (below) {
          ia = L.row0 + 31 - clz32(below);
          Fa = ldF<CW>(L, ia);
          nab = edge_num(ia, Fa, ib, Fb, w2);
          dab = (double)(ib - ia);
        } else {
          nab = -INFINITY;
          dab = 1.0;
        }
      }
      aw |= 1u << r;
      ia = ib;
      Fa = Fb;
      nab = fresh ? -INFINITY : nbi;
      dab = fresh ? 1.0 : dbi;
      ib = row;
      Fb = Fi;
    }
  }
  return aw;
}

// ---------------------------------------------------------------------------------------
// phase 2, one round: the lane whose band is the first of the right group joins the hull of
// the `half` bands to its left with the hull of the `half` bands starting at its own, if a
// label run crosses the boundary.  Walks the common tangent and clears the bits in between.
// ---------------------------------------------------------------------------------------
template <int CW>
EDT_LANE void phase2_merge(const Lane &L, int half) {
  if ((L.band & (2 * half - 1)) != half) return;
  const int R = L.row0;  // first row of the right group
  if (R >= L.n || !(L.nzw & 1u) || (L.rsw & 1u)) return;
  const double w2 = L.w2;
  const int glo = (L.band - half) * 32;
  int ghi = (L.band + half) * 32;
  if (ghi > L.n) ghi = L.n;
  ghi -= 1;
  int Llo = prev_set<CW>(L.rsp, L.colc, R, glo);
  if (Llo < 0) Llo = glo;
  const int nxt = next_set<CW>(L.rsp, L.colc, R, ghi);
  const int Rhi = nxt < 0 ? ghi : nxt - 1;

  int u = R - 1;  // last vertex of the left hull (always alive)
  int v = R;      // first vertex of the right hull (always alive)
  double Fu = ldF<CW>(L, u), Fv = ldF<CW>(L, v);
  int up = prev_set<CW>(L.alive, L.colc, u, Llo);
  int vn = next_set<CW>(L.alive, L.colc, v, Rhi);
  double Fup = up >= 0 ? ldF<CW>(L, up) : 0.0;
  double Fvn = vn >= 0 ? ldF<CW>(L, vn) : 0.0;
  while (true) {
    const double nuv = edge_num(u, Fu, v, Fv, w2);
    if (up >= 0 && nuv * (double)(u - up) <= edge_num(up, Fup, u, Fu, w2) * (double)(v - u)) {
      L.alive[addr_word<CW>(L.colc, u >> 5)] &= ~(1u << (u & 31));
      u = up;
      Fu = Fup;
      up = prev_set<CW>(L.alive, L.colc, u, Llo);
      Fup = up >= 0 ? ldF<CW>(L, up) : 0.0;
    } else if (vn >= 0 &&
               edge_num(v, Fv, vn, Fvn, w2) * (double)(v - u) <= nuv * (double)(vn - v)) {
      L.alive[addr_word<CW>(L.colc, v >> 5)] &= ~(1u << (v & 31));
      v = vn;
      Fv = Fvn;
      vn = next_set<CW>(L.alive, L.colc, v, Rhi);
      Fvn = vn >= 0 ? ldF<CW>(L, vn) : 0.0;
    } else {
      break;
    }
  }
}

EDT_LANE float finish_f(float m, int epi) {
  if ((epi & kLaneEpiToInf) && m >= 3.402823466e+38f) m = INFINITY;
  if (epi & kLaneEpiSqrt) m = sqrtf(m);
  return m;
}

// ---------------------------------------------------------------------------------------
// phase 3: evaluate the envelope on the lane's 32 rows.  f[r] holds F(row0+r) on entry and
// the result on exit (background rows keep their 0).  `aw` = this band's merged alive word.
// ---------------------------------------------------------------------------------------
template <int CW, int EPI, bool BB>
EDT_LANE void phase3_eval(const Lane &L, uint32_t aw, float *f) {
  constexpr int kFar = 1 << 14;  // "no border on this side" distance
  const double w2 = L.w2;
  const int row0 = L.row0, n = L.n;
  const uint32_t nzw = L.nzw, rsw = L.rsw;
  if (nzw == 0) return;

  // last row of the run that is still open when the band ends
  int hi_carry = row0 + 31;
  if (row0 + 31 < n - 1 && (nzw >> 31)) {
    const int nx = next_set<CW>(L.rsp, L.colc, row0 + 31, n - 1);
    hi_carry = nx < 0 ? n - 1 : nx - 1;
  }
  if (hi_carry > n - 1) hi_carry = n - 1;

  int j = row0, jn = -1, run_hi = row0;
  int lo1 = -kFar, hi1 = 2 * kFar;  // run_lo - 1 / run_hi + 1 where that side has a border
  double Fj = 0.0, Fjn = 0.0;
  if ((nzw & 1u) && !(rsw & 1u)) {
    // The band begins inside a run that started in an earlier band: find the hull vertex that
    // owns row0 -- start from the last vertex at or before it and walk down the (unimodal)
    // values towards earlier vertices.
    const int run_lo = prev_set<CW>(L.rsp, L.colc, row0, 0);
    const uint32_t above = rsw & 0xFFFFFFFEu;
    run_hi = above ? row0 + ctz32(above) - 1 : hi_carry;
    lo1 = (BB || run_lo > 0) ? run_lo - 1 : -kFar;
    hi1 = (BB || run_hi < n - 1) ? run_hi + 1 : 2 * kFar;
    j = prev_set<CW>(L.alive, L.colc, row0 + 1, run_lo);
    Fj = ldF<CW>(L, j);
    double vj = para(row0, j, Fj, w2);
    while (true) {
      const int jp = prev_set<CW>(L.alive, L.colc, j, run_lo);
      if (jp < 0) break;
      const double Fjp = ldF<CW>(L, jp);
      const double vp = para(row0, jp, Fjp, w2);
      if (!(vp < vj)) break;
      j = jp;
      Fj = Fjp;
      vj = vp;
    }
    // next hull vertex after j
    if (j >= row0) {
      const uint32_t m = aw & 0xFFFFFFFEu;  // j == row0
      if (m && row0 + ctz32(m) <= run_hi) jn = row0 + ctz32(m);
      else jn = run_hi > row0 + 31 ? next_set<CW>(L.alive, L.colc, row0 + 31, run_hi) : -1;
    } else {
      jn = next_set<CW>(L.alive, L.colc, j, run_hi);
    }
    Fjn = jn >= 0 ? ldF<CW>(L, jn) : 0.0;
  }

#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 32; ++r) {
    if ((nzw >> r) & 1u) {
      const int p = row0 + r;
      bool need = false;  // jn / Fjn must be (re)loaded
      if ((rsw >> r) & 1u) {  // a run starts at p: its first hull vertex is p
        const uint32_t above = r < 31 ? (rsw & (0xFFFFFFFEu << r)) : 0u;
        run_hi = above ? row0 + ctz32(above) - 1 : hi_carry;
        lo1 = (BB || p > 0) ? p - 1 : -kFar;
        hi1 = (BB || run_hi < n - 1) ? run_hi + 1 : 2 * kFar;
        j = p;
        Fj = (double)f[r];
        need = true;
      }
      double best = para(p, j, Fj, w2);
      while (true) {
        if (need) {
          // next hull vertex after j: a bit scan of this lane's own alive word ...
          const unsigned rj = (unsigned)(j - row0);
          int q = -1;
          if (rj < 32u) {
            const uint32_t m = rj < 31u ? (aw & (0xFFFFFFFEu << rj)) : 0u;
            if (m != 0u) {
              q = row0 + ctz32(m);
              if (q > run_hi) q = -1;
            } else if (run_hi > row0 + 31) {
              // ... unless the run continues past this band
              q = next_set<CW>(L.alive, L.colc, row0 + 31, run_hi);
            }
          } else {  // j sits in an earlier band
            q = next_set<CW>(L.alive, L.colc, j, run_hi);
          }
          jn = q;
          if (q == p + 1 && r < 31) Fjn = (double)f[r < 31 ? r + 1 : r];  // still the input value
          else if (q >= 0) Fjn = ldF<CW>(L, q);
          need = false;
        }
        if (jn < 0) break;
        const double cand = para(p, jn, Fjn, w2);
        if (!(cand < best)) break;
        best = cand;
        j = jn;
        Fj = Fjn;
        need = true;
      }
      // border parabolas of height 0 just outside the run.  fp32 rounding is monotone, so one
      // narrowing of the fp64 minimum equals the reference's separate narrowings
      // (src/edt.hpp:233-242, :310-311); the nearer border dominates the farther one.
      const int dl = p - lo1, dr = hi1 - p;
      const int dm = dl < dr ? dl : dr;
      if (dm < kFar) best = fmin(best, w2 * sq_i(dm));
      f[r] = finish_f((float)best, EPI);
    }
  }
}

}  // namespace edt_lane
