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
//   phase 1: every lane builds the hull of its own 32 rows (monotone chain, rows in VGPRs) and
//            notes where the field is flat (|F[r]-F[r-1]| <= w2);
//   phase 2: log2(NBP) rounds of pairwise hull merges across band-group boundaries (bridge
//            walk), only where a label run crosses the boundary -- skipped altogether by a wave
//            whose boundaries are all "quiet" (boundary_quiet);
//   phase 3: rows that own themselves (own_mask: flat on both sides of an alive row) take
//            min(F, border) directly; the others are swept over the merged hull with the
//            reference's own expression fl32(w2*(p-j)^2 + F[j]) (src/edt.hpp:230, :307), the
//            border parabolas (src/edt.hpp:233-242, :310-311) and the fused toinfinite / sqrt
//            epilogue (src/edt.hpp:47-53, :599-601).
//
// Arithmetic notes (bit parity with the reference):
//   * w2 is the fp32 product w*w widened to fp64 (src/edt.hpp:181, :258);
//   * w2 * d^2 with d < 2^14 is EXACT in fp64 (24 x 28 significant bits), so
//     fma(w2, d^2, F) == w2*d^2 + F bit for bit: one rounding either way;
//   * hull orientation tests compare crossing abscissae by cross-multiplication, no divide.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef EDT_LANE
#error "define EDT_LANE (function qualifiers) before including edt_colwave_lane.h"
#endif

namespace edt_lane {

// Geometry of the workgroup tile per wave shape.  Axes up to 512 rows (CW >= 4): 32 columns, i.e.
// whole 128-byte lines per row, at most 64 KiB of fp32 -> two workgroups per CU.  1024-row axes
// (CW = 2): a 32-column tile would fill the LDS of a CU (128 KiB) and leave ONE workgroup per CU,
// whose compute phase nothing overlaps (measured: 0.35 ms per pass against 0.23 ms for 512 rows, the
// memory phases alone 0.245 ms); so those use 16-column tiles (64-byte row pieces, 64.5 KiB), two
// workgroups per CU, the two halves of a line handled back to back on one XCD (edt_colwave.hip).
template <int CW>
struct TileGeom {
  static constexpr int kCols = CW <= 2 ? 16 : 32;                    // floats per LDS tile row
  static constexpr int kBandFloats = CW <= 2 ? 32 * 16 + 4 : 32 * 32;  // LDS floats per band of 32 rows
  static constexpr int kBandWords = CW <= 2 ? 18 : 32;               // bit-plane words per band
};

#if defined(__HIP_DEVICE_COMPILE__)
// w = 2*w + cond: one add-with-carry whose carry-in is the wave-wide compare mask.  Being a
// volatile asm it also pins the accumulation to the row it belongs to (written as plain C the
// compiler gathers the 32 compares into an OR tree at the end and keeps every operand alive).
#define EDT_SHIFT_IN(w, cond)                                                                  \
  asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(w) : "s"(__ballot(cond)) : "vcc")
EDT_LANE uint32_t brev32(uint32_t v) { return __brev(v); }
EDT_LANE int mul24(int a, int b) { return __mul24(a, b); }
EDT_LANE int clz32(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int ctz32(uint32_t v) { return __builtin_ctz(v); }
EDT_LANE double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }
#else
#define EDT_SHIFT_IN(w, cond) (w) = ((w) << 1) | ((cond) ? 1u : 0u)
#ifdef EDT_LANE_STATS
// host emulation only: event counters (::edt_lane_stat is declared by tests/lane_stats.cpp, which
// aggregates them per wave and row)
#define EDT_STAT(kind, row, count) ::edt_lane_stat((kind), (row), (count))
#endif
EDT_LANE uint32_t brev32(uint32_t v) {
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
EDT_LANE int mul24(int a, int b) { return a * b; }
EDT_LANE int clz32(uint32_t v) { return __builtin_clz(v); }
EDT_LANE int ctz32(uint32_t v) { return __builtin_ctz(v); }
EDT_LANE double fma64(double a, double b, double c) { return fma(a, b, c); }
#endif

enum : int { kLaneEpiToInf = 1, kLaneEpiSqrt = 2 };

#ifndef EDT_STAT
#define EDT_STAT(kind, row, count) ((void)0)
#endif
// event kinds of EDT_STAT
enum : int { kStPop = 0, kStBridgeCall, kStBridgeStep, kStOwnRow, kStGeneralRow, kStResync, kStPrologueStep,
             kStAdvance, kStFindNext, kStFindNextLds, kStFresh, kStKinds };

// LDS addressing of one lane.  32-column tiles: the fp32 tile is row-major [row][32]; the 16-byte
// granule that holds the lane's column is XOR-rotated by its band so that the 32 lanes of a
// half-wave (32/CW bands x CW columns) read 32 different banks when each reads "its" row.
// The bit planes ([band][32] words) use the same rotation.
// 16-column tiles (CW = 2): rows are 16 floats, no rotation (so that 16-byte granules stay whole);
// instead every band is followed by 4 floats of padding, which spreads "every lane reads its own
// row r" (2 columns x 32 bands) over 16 banks -- a 4-way conflict on 64 accesses per lane and tile,
// cheap next to what the rotation would cost in 4-byte tile traffic -- and the bit planes use 18
// words per band (2-way, the minimum for 64 lanes).
template <int CW>
EDT_LANE int addr_swz(int band) { return CW <= 2 ? 0 : (band * CW) & 31; }  // rotate by one wave's columns per band
template <int CW>
EDT_LANE int addr_tile(int colc, int row) {
  if (CW <= 2) return (row >> 5) * TileGeom<CW>::kBandFloats + ((row & 31) << 4) + colc;
  return (row << 5) + (colc ^ addr_swz<CW>(row >> 5));
}
template <int CW>
EDT_LANE int addr_word(int colc, int band) {
  if (CW <= 2) return band * TileGeom<CW>::kBandWords + colc;
  return (band << 5) + (colc ^ addr_swz<CW>(band));
}

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
  // run structure carried across bands (one wave-level scan over the column's bands):
  int lo_in;             // first row of the run that is open when the band begins (last run
                         //   start in an earlier band; -1 if there is none)
  int hi_out;            // last row of the run that is open when the band ends (row before the
                         //   first run start in a later band, n-1 if there is none)
  uint32_t own;          // rows of this band that are self-owned (own_mask), set after the merges
};

// ---------------------------------------------------------------------------------------
// Index form of pass 1.  Pass 1 (src/edt.hpp:70-119) has the closed form
//     F = fl32(d*d),  d = T[k],  k = min(x-s+1, e-x+1),  T[k] = k-fold sequential fp32 sum of wx,
// for a voxel x inside the maximal run [s,e] of one non-zero label (a side without a boundary does not count;
// no boundary at all: +inf).  When k*wx is exactly representable for every k of the row (edt_rowwave.hip:
// row_codes_exact) the sequential sums are exact, T[k] = k*wx, and pass 1 may hand the first column pass the 16-bit
// index k instead of the fp32 value: 2 bytes less written and 2 bytes less read per voxel.  The column kernel turns
// the index back into F while it fills its tile: k = 0 (background) -> 0, kCodeInf -> +inf, then tofinite
// (src/edt.hpp:39-45) as pass 1 applies it.  flim = bit pattern of FLT_MAX (tofinite) or of +inf; with a black
// border every run has two borders, no index is kCodeInf and flim is +inf.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kCodeInf = 0xFFFFu;
EDT_LANE float code_value(uint32_t k, float w, int flim) {
  const float d = k == kCodeInf ? INFINITY : (float)k * w;  // exact product
  const float sq = d * d;                                   // `d[i] *= d[i]` (src/edt.hpp:116-118)
  int f;
  memcpy(&f, &sq, 4);
  f = f < flim ? f : flim;  // non-negative floats order like their bit patterns
  float out;
  memcpy(&out, &f, 4);
  return out;
}

// How one wave-wide memory instruction of the tile fill / write-back maps to the tile: lane l of
// instruction i moves G consecutive floats of tile row `row` whose LDS words are linear
// (row*32 + phys ..) -- that is what global_load_lds requires -- and whose global columns are
// phys ^ swz(band).  G = 4 (16-byte granules) needs the band rotation to be a multiple of 4
// columns (CW >= 4) and 16-byte aligned rows in memory; G = 1 serves the 2-column waves of
// 1024-row axes and volumes whose x extent is not a multiple of 4.
template <int CW>
struct TileIO {
  static constexpr int kGran = 4;  // widest granule the wave shape allows (rotation / rows are multiples of 4)
  // wave-wide instructions that move the whole tile with granules of G floats
  static constexpr int count(int NBP, int G) { return NBP * TileGeom<CW>::kCols / (2 * G); }
};
template <int CW, int G>
EDT_LANE int io_row(int i, int lane) {
  return (64 * G / TileGeom<CW>::kCols) * i + (lane * G) / TileGeom<CW>::kCols;
}
template <int CW, int G>
EDT_LANE int io_gcol(int i, int lane) {
  return ((lane * G) & (TileGeom<CW>::kCols - 1)) ^ addr_swz<CW>(io_row<CW, G>(i, lane) >> 5);
}
// LDS word of lane `lane` of instruction i: linear in the lane (global_load_lds writes lane l at
// base + l * 4 * G bytes); an instruction never straddles a band, so the band padding of the
// 16-column tiles only moves the instruction's base
template <int CW, int G>
EDT_LANE int io_lds_word(int i, int lane) {
  if (CW <= 2) {
    const int row = (64 * G / 16) * i;  // first row of the instruction
    return (row >> 5) * TileGeom<CW>::kBandFloats + ((row & 31) << 4) + lane * G;
  }
  return (i * 64 + lane) * G;
}

// per-band inputs of that scan
EDT_LANE int band_last_start(uint32_t rsw, int row0) { return rsw ? row0 + 31 - clz32(rsw) : -1; }
EDT_LANE int band_first_start(uint32_t rsw, int row0, int n) { return rsw ? row0 + ctz32(rsw) : n; }

// value of parabola j (height Fj) at row p -- the reference's output expression
// w2*sq(p-j) + Fj.  d = p - j as a double: w2*d is exact (24 + 12 bits) and the fma multiplies
// it by d exactly, so the only rounding is the final addition, as in the reference.
EDT_LANE double para_d(int d, double Fj, double w2) {
  const double dd = (double)d;
  return fma64(w2 * dd, dd, Fj);
}
EDT_LANE double para(int p, int j, double Fj, double w2) { return para_d(p - j, Fj, w2); }

// numerator of the crossing abscissa of parabolas p < q:  (Fq - Fp) + w2*(q-p)*(q+p)
// (src/edt.hpp:206-208: ff[i] - ff[v[k]] + factor1 * factor2; the product is exact)
EDT_LANE double edge_num(int p, double Fp, int q, double Fq, double w2) {
  return fma64(w2 * (double)(q - p), (double)(q + p), Fq - Fp);
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
// word.  The stack of the monotone chain IS the alive word (it starts as the foreground word
// and loses a bit per pop).  In a run every row is pushed, so when row r is examined the top of
// the stack is row r-1 (F in a register) and the orientation test
//     pop while  s(ib,row) <= s(ia,ib)  <=>  num(ib,row)*(ib-ia) <= num(ia,ib)*(row-ib)
// (src/edt.hpp:210, :287) reduces, for its first and usually only evaluation, to
//     num(r-1,r) * dab <= nab,     num(r-1,r) = (F[r]-F[r-1]) + w2*(2*row-1)
// with (nab, dab) = numerator / width of the edge under the top, carried in registers.  Rows
// whose test cannot fire (background, run start, row after a run start) are masked by `dis`;
// the pop loop proper is the rare slow path and re-derives the stack from the alive word.
// ---------------------------------------------------------------------------------------
// what phase 1 leaves behind besides the alive word
struct Hull1 {
  uint32_t aw;    // alive word of the band
  uint32_t flat;  // bit r: |F[r] - F[r-1]| <= w2
  double nb0, nb1, nb31;  // crossing numerators num(r-1, r) of the band's rows 0, 1 and 31
};

// The flat bits alone (bit r: |F[r] - F[r-1]| <= w2, exact in fp64; row 0 against the last row of the
// band below).  If EVERY foreground row of a column that continues a run is flat, every row of the run
// owns itself: for rows p and j of one run, F[p] - F[j] is a sum of |p-j| steps of at most w2, hence
// F[p] <= F[j] + w2*|p-j| <= F[j] + w2*(p-j)^2, i.e. no parabola of the run lies below the row's own
// value -- the envelope is F itself and neither hulls nor merges nor a sweep are needed (the result
// min(F, border) is what the reference rounds to, bit for bit: the own term is exact).  The kernel
// tests that wave-wide (`need` = nzw & ~rsw) before it builds any hull.
EDT_LANE uint32_t flat_word(const Lane &L, const float *f, float fprev) {
  const double w2 = L.w2;
  uint32_t fl = 0;  // built most-significant-row first, flipped at the end
  double Fb = (double)f[0];
  EDT_SHIFT_IN(fl, fabs(Fb - (double)fprev) <= w2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 1; r < 32; ++r) {
    const double Fi = (double)f[r];
    EDT_SHIFT_IN(fl, fabs(Fi - Fb) <= w2);
    Fb = Fi;
  }
  return brev32(fl);
}

template <int CW>
EDT_LANE Hull1 phase1_hull(const Lane &L, const float *f, float fprev, uint32_t flat) {
  const uint32_t rs1 = L.rsw | 1u;  // the band's first row starts a (local) chain
  const uint32_t dis = ~L.nzw | rs1 | (rs1 << 1);
  const double w2 = L.w2, w2x2 = w2 + w2;
  uint32_t aw = L.nzw;
  double nab = -INFINITY, dab = 1.0;
  double Fb = (double)f[0];
  // (`flat`, bit r: |F[r] - F[r-1]| <= w2, comes from flat_word: where it holds on both sides of an
  // alive row, the row's own parabola is the envelope there, see own_mask)
  double c = w2 * (double)(2 * L.row0 - 1);  // w2*(2*row-1) for row = row0; exact, and so are its updates
  Hull1 H;
  H.nb0 = (Fb - (double)fprev) + c;
  H.nb1 = 0.0;
  H.nb31 = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 1; r < 32; ++r) {
    c += w2x2;
    const double Fi = (double)f[r];
    const double t = Fi - Fb;
    double nbi = t + c;
    if (r == 1) H.nb1 = nbi;
    if (r == 31) H.nb31 = nbi;
    double dbi = 1.0;
    if (!((dis >> r) & 1u) && nbi * dab <= nab) {
      // slow path: pop.  top = r-1, the entries below it come from the alive word
      const int row = L.row0 + r;
      const uint32_t rsm = rs1 & (0xFFFFFFFFu >> (31 - r));
      const uint32_t segmask = 0xFFFFFFFFu << (31 - clz32(rsm));
      int ib = r - 1;
      uint32_t below = aw & segmask & ((1u << ib) - 1u);
      int ia = 31 - clz32(below);  // exists: the test only fires with two entries on the stack
      double Fa = ldF<CW>(L, L.row0 + ia);
      double Ftop = Fb;
      while (true) {
        EDT_STAT(kStPop, r, 1);
        aw &= ~(1u << ib);
        ib = ia;
        Ftop = Fa;
        below = aw & segmask & ((1u << ib) - 1u);
        if (below) {
          ia = 31 - clz32(below);
          Fa = ldF<CW>(L, L.row0 + ia);
          nab = edge_num(L.row0 + ia, Fa, L.row0 + ib, Ftop, w2);
          dab = (double)(ib - ia);
        } else {
          nab = -INFINITY;
          dab = 1.0;
        }
        nbi = edge_num(L.row0 + ib, Ftop, row, Fi, w2);
        dbi = (double)(r - ib);
        if (!(nbi * dab <= nab * dbi)) break;
      }
    }
    nab = nbi;  // push r: the edge under the new top is (old top, r)
    dab = dbi;
    Fb = Fi;
  }
  H.flat = flat;
  H.aw = aw;
  return H;
}

// ---------------------------------------------------------------------------------------
// Rows whose own parabola is the envelope value at the row itself ("self-owned").  On a lower
// hull the values para(p, j) over consecutive vertices j are unimodal, so vertex p owns row p
// iff it is no worse there than its two hull neighbours; when those are the adjacent rows that
// is |F[p] - F[p-1]| <= w2 and |F[p+1] - F[p]| <= w2 (the `flat` bits) -- a side that ends the
// run has no neighbour (the border parabola is min'ed in separately).  Inputs are this band's
// words plus bit 31 of the band below and bit 0 of the band above (0 where there is none).
// ---------------------------------------------------------------------------------------
EDT_LANE uint32_t own_mask(uint32_t nzw, uint32_t rsw, uint32_t aw, uint32_t flat, uint32_t aw_prev31,
                           uint32_t nz_next0, uint32_t rs_next0, uint32_t aw_next0, uint32_t flat_next0) {
  const uint32_t alive_m1 = (aw << 1) | (aw_prev31 & 1u);
  const uint32_t alive_p1 = (aw >> 1) | (aw_next0 << 31);
  const uint32_t flat_p1 = (flat >> 1) | (flat_next0 << 31);
  const uint32_t nz_p1 = (nzw >> 1) | (nz_next0 << 31);
  const uint32_t rs_p1 = (rsw >> 1) | (rs_next0 << 31);
  const uint32_t ends = rs_p1 | ~nz_p1;  // the run ends at this row
  const uint32_t P = rsw | (alive_m1 & flat);
  const uint32_t N = ends | (alive_p1 & flat_p1);
  return nzw & aw & P & N;
}

// ---------------------------------------------------------------------------------------
// Is the band boundary below this lane "quiet", i.e. would the bridge walk of phase 2 find the
// tangent (R-1, R) at once and remove nothing?  With the neighbours R-2 and R+1 alive and
// adjacent, the two orientation tests of the walk are exactly the pop tests phase 1 would have
// made for rows R and R+1 had the column not been cut:  num(R-1,R) <= num(R-2,R-1)  and
// num(R,R+1) <= num(R-1,R)  (same expressions, same rounding).  A side on which the run ends has
// no neighbour and cannot lose a vertex.  If every boundary of a wave is quiet the merge rounds
// would change nothing at any level and are skipped altogether.
// prev_* : alive / run-start words and nb31 of the band below, after phase 1.
// ---------------------------------------------------------------------------------------
EDT_LANE bool boundary_quiet(const Lane &L, const Hull1 &H, uint32_t prev_aw, uint32_t prev_rsw,
                             double prev_nb31) {
  const bool cross = L.row0 > 0 && L.row0 < L.n && (L.nzw & 1u) && !(L.rsw & 1u);
  if (!cross) return true;  // no run crosses: nothing to merge
  const bool has_up = !((prev_rsw >> 31) & 1u);                       // row R-2 belongs to the run
  const bool has_vn = ((L.nzw >> 1) & 1u) && !((L.rsw >> 1) & 1u);     // row R+1 belongs to the run
  const bool ok_l = !has_up || (((prev_aw >> 30) & 1u) && !(H.nb0 <= prev_nb31));
  const bool ok_r = !has_vn || (((H.aw >> 1) & 1u) && !(H.nb1 <= H.nb0));
  return ok_l && ok_r;
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
  const int Llo = L.lo_in > glo ? L.lo_in : glo;
  const uint32_t above = L.rsw & 0xFFFFFFFEu;
  const int run_hi = above ? R + ctz32(above) - 1 : L.hi_out;
  const int Rhi = run_hi < ghi ? run_hi : ghi;

  // The walk almost always stays inside the two bit words that meet at the boundary (a handful of
  // steps against 32 rows per word), so those two words live in registers: a step is then bit
  // arithmetic plus ONE LDS read (the height of the new neighbour) instead of a read-modify-write of
  // the alive plane followed by a dependent search through it.  Only a walk that leaves the two
  // words goes back to the plane.  (No other lane touches these words in this round: the groups of
  // the merging lanes of a column are disjoint.)
  const int wl0 = R - 32;  // first row of the left word
  uint32_t wl = L.alive[addr_word<CW>(L.colc, L.band - 1)];
  uint32_t wr = L.alive[addr_word<CW>(L.colc, L.band)];
  const uint32_t lmask = Llo > wl0 ? 0xFFFFFFFFu << (Llo - wl0) : 0xFFFFFFFFu;        // rows >= Llo
  const uint32_t rmask = Rhi < R + 31 ? 0xFFFFFFFFu >> (31 - (Rhi - R)) : 0xFFFFFFFFu;  // rows <= Rhi
  auto prev_of = [&](int q) -> int {  // hull vertex below q inside [Llo, q)
    if (q >= wl0) {
      const uint32_t m = wl & lmask & ~(0xFFFFFFFFu << (q - wl0));
      if (m) return wl0 + 31 - clz32(m);
      if (Llo >= wl0) return -1;
      return prev_set<CW>(L.alive, L.colc, wl0, Llo);
    }
    return prev_set<CW>(L.alive, L.colc, q, Llo);
  };
  auto next_of = [&](int q) -> int {  // hull vertex above q inside (q, Rhi]
    if (q <= R + 31) {
      const uint32_t m = wr & rmask & (0xFFFFFFFEu << (q - R));
      if (m) return R + ctz32(m);
      if (Rhi <= R + 31) return -1;
      return next_set<CW>(L.alive, L.colc, R + 31, Rhi);
    }
    return next_set<CW>(L.alive, L.colc, q, Rhi);
  };

  int u = R - 1;  // last vertex of the left hull (always alive)
  int v = R;      // first vertex of the right hull (always alive)
  double Fu = ldF<CW>(L, u), Fv = ldF<CW>(L, v);
  int up = prev_of(u);
  int vn = next_of(v);
  double Fup = up >= 0 ? ldF<CW>(L, up) : 0.0;
  double Fvn = vn >= 0 ? ldF<CW>(L, vn) : 0.0;
  EDT_STAT(kStBridgeCall, half, 1);
  while (true) {
    EDT_STAT(kStBridgeStep, half, 1);
    const double nuv = edge_num(u, Fu, v, Fv, w2);
    if (up >= 0 && nuv * (double)(u - up) <= edge_num(up, Fup, u, Fu, w2) * (double)(v - u)) {
      if (u >= wl0) wl &= ~(1u << (u - wl0));
      else L.alive[addr_word<CW>(L.colc, u >> 5)] &= ~(1u << (u & 31));
      u = up;
      Fu = Fup;
      up = prev_of(u);
      Fup = up >= 0 ? ldF<CW>(L, up) : 0.0;
    } else if (vn >= 0 &&
               edge_num(v, Fv, vn, Fvn, w2) * (double)(v - u) <= nuv * (double)(vn - v)) {
      if (v <= R + 31) wr &= ~(1u << (v - R));
      else L.alive[addr_word<CW>(L.colc, v >> 5)] &= ~(1u << (v & 31));
      v = vn;
      Fv = Fvn;
      vn = next_of(v);
      Fvn = vn >= 0 ? ldF<CW>(L, vn) : 0.0;
    } else {
      break;
    }
  }
  L.alive[addr_word<CW>(L.colc, L.band - 1)] = wl;
  L.alive[addr_word<CW>(L.colc, L.band)] = wr;
}

EDT_LANE float finish_f(float m, int epi) {
  if ((epi & kLaneEpiToInf) && m >= 3.402823466e+38f) m = INFINITY;
  if (epi & kLaneEpiSqrt) m = sqrtf(m);
  return m;
}

// ---------------------------------------------------------------------------------------
// phase 3: evaluate the envelope on the lane's 32 rows.  f[r] holds F(row0+r) on entry and
// the result on exit (background rows keep their 0).  `aw` = this band's merged alive word.
//
// Sweep state (row indices relative to row0, so they may be negative or exceed 31):
//   jr / Fj    the hull vertex owning the current row,
//   jnr / Fjn  the next hull vertex of the run (Fjn = +inf when there is none),
//   awrun      alive bits of this band that belong to the current run,
//   lo1r/hi1r  the rows just outside the run where a border parabola sits (far away if none).
// Common path per row: two parabola evaluations, one compare, the border term in fp32.
// ---------------------------------------------------------------------------------------
// next hull vertex after the one at relative row jr (run up to run_hir, alive bits of this band that
// belong to the run in awrun).  r = the row being evaluated; fnext = F(row0+r+1) from a register.
template <int CW>
EDT_LANE void find_next(const Lane &L, int jr, int r, uint32_t awrun, int run_hir, float fnext,
                        int &jnr, double &Fjn, double &dn) {
  const int row0 = L.row0;
  Fjn = INFINITY;
  EDT_STAT(kStFindNext, r, 1);
  const uint32_t m = (unsigned)jr < 31u ? (awrun & (0xFFFFFFFEu << jr)) : 0u;
  if (m) {
    jnr = ctz32(m);
    if (jnr == r + 1) Fjn = (double)fnext;  // still the input value
    else Fjn = ldF<CW>(L, row0 + jnr);
  } else if (jr < 0 || run_hir > 31) {
    // the owner still sits in an earlier band, or the run continues past this band
    EDT_STAT(kStFindNextLds, r, 1);
    const int q = next_set<CW>(L.alive, L.colc, (jr < 0 || jr >= 31) ? row0 + jr : row0 + 31,
                               row0 + run_hir);
    if (q >= 0) {
      jnr = q - row0;
      Fjn = ldF<CW>(L, q);
    }
  }
  dn = (double)(jnr - r);
}

template <int CW, bool BB>
EDT_LANE void phase3_eval(const Lane &L, uint32_t aw, float *f, int epi) {
  constexpr int kFar = 1 << 14;  // "no border on this side" distance
  const double w2 = L.w2;
  const float w2f = (float)w2;  // exactly the fp32 product w*w
  const int row0 = L.row0, n = L.n;
  const uint32_t nzw = L.nzw, rsw = L.rsw, own = L.own;
  if (nzw == 0) return;
  const int hi_carry = L.hi_out - row0;  // last row (relative) of the run open at the band's end

  // Run state (every foreground row): run_hir, awrun, and the distances to the rows just outside
  // the run where a border parabola sits, as floats stepped by +-1 per row (a missing border is
  // a distance >= kFar: dl only grows, dr shrinks by at most n <= 2048).
  int run_hir = 0;
  uint32_t awrun = 0;
  float dl = (float)kFar, dr = (float)(4 * kFar);
  const bool midrun = (nzw & 1u) && !(rsw & 1u);  // the band begins inside a run of an earlier band
  if (midrun) {
    const int run_lo = L.lo_in;
    const uint32_t above = rsw & 0xFFFFFFFEu;
    run_hir = above ? ctz32(above) - 1 : hi_carry;
    awrun = aw & (0xFFFFFFFFu >> (31 - (run_hir < 31 ? run_hir : 31)));
    dl = (BB || run_lo > 0) ? (float)(row0 - run_lo + 1) : (float)kFar;
    dr = (BB || row0 + run_hir < n - 1) ? (float)(run_hir + 1) : (float)(4 * kFar);
  }
  // Sweep state (only kept across consecutive rows that are NOT self-owned): owner vertex jr / Fj,
  // next vertex jnr / Fjn (Fjn = +inf: none), and dj = r - jr, dn = jnr - r as doubles.
  int jr = 0, jnr = 0;
  double Fj = 0.0, Fjn = INFINITY, dj = 0.0, dn = 0.0;

#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 32; ++r) {
    if ((nzw >> r) & 1u) {
      const bool fresh = (rsw >> r) & 1u;
      if (fresh) {  // a run starts at this row
        EDT_STAT(kStFresh, r, 1);
        const uint32_t above = r < 31 ? (rsw & (0xFFFFFFFEu << r)) : 0u;
        run_hir = above ? ctz32(above) - 1 : hi_carry;
        awrun = aw & (0xFFFFFFFFu >> (31 - (run_hir < 31 ? run_hir : 31))) & (0xFFFFFFFFu << r);
        dl = (BB || row0 + r > 0) ? 1.0f : (float)kFar;
        dr = (BB || row0 + run_hir < n - 1) ? (float)(run_hir + 1 - r) : (float)(4 * kFar);
      }
      float res;
      if ((own >> r) & 1u) {
        EDT_STAT(kStOwnRow, r, 1);
        res = f[r];  // the row's own parabola is the envelope here: no fp64 work at all
      } else {
        EDT_STAT(kStGeneralRow, r, 1);
        const float fnext = f[r < 31 ? r + 1 : r];
        // (re)establish the sweep state where the previous row did not leave one
        bool need = false;  // the next vertex must be (re)loaded
        if (r == 0 && !fresh) {
          // The band begins inside a run: find the hull vertex that owns row0 -- start from the
          // last vertex at or before it and walk down the (unimodal) values towards earlier ones.
          const int run_lo = L.lo_in;
          int j = prev_set<CW>(L.alive, L.colc, row0 + 1, run_lo);
          Fj = ldF<CW>(L, j);
          double vj = para(row0, j, Fj, w2);
          while (true) {
            EDT_STAT(kStPrologueStep, r, 1);
            const int jp = prev_set<CW>(L.alive, L.colc, j, run_lo);
            if (jp < 0) break;
            const double Fjp = ldF<CW>(L, jp);
            const double vp = para(row0, jp, Fjp, w2);
            if (!(vp < vj)) break;
            j = jp;
            Fj = Fjp;
            vj = vp;
          }
          jr = j - row0;
          dj = (double)(-jr);
          need = true;
        } else if (fresh) {
          jr = r;  // the first hull vertex of a run is its first row
          dj = 0.0;
          Fj = (double)f[r];
          need = true;
        } else if ((own >> (r > 0 ? r - 1 : 0)) & 1u) {
          EDT_STAT(kStResync, r, 1);
          jr = r - 1;  // the previous row owned itself (its register already holds the result)
          dj = 1.0;
          Fj = ldF<CW>(L, row0 + r - 1);
          need = true;
        }
        double best = fma64(w2 * dj, dj, Fj);
        while (true) {
          if (need) find_next<CW>(L, jr, r, awrun, run_hir, fnext, jnr, Fjn, dn);
          const double cand = fma64(w2 * dn, dn, Fjn);
          if (!(cand < best)) break;  // (Fjn = +inf never wins)
          EDT_STAT(kStAdvance, r, 1);
          best = cand;  // the next vertex takes over
          jr = jnr;
          Fj = Fjn;
          dj = -dn;
          need = true;
        }
        res = (float)best;
        dj += 1.0;
        dn -= 1.0;
      }
      // border parabolas of height 0 just outside the run (src/edt.hpp:233-242, :310-311):
      // fl32(w2 * d^2) is one exact-product fp32 multiply; the nearer border dominates.
      const float dm = fminf(dl, dr);
      if (BB || dm < (float)kFar) res = fminf(res, w2f * (dm * dm));
      f[r] = res;
    }
    dl += 1.0f;
    dr -= 1.0f;
  }
  // fused epilogue of the last pass, as whole-band loops behind wave-uniform branches (inside
  // the row loop the compiler would evaluate the sqrt sequence unconditionally and select)
  if (epi & kLaneEpiToInf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 32; ++r) f[r] = f[r] >= 3.402823466e+38f ? INFINITY : f[r];
  }
  if (epi & kLaneEpiSqrt) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 32; ++r) f[r] = sqrtf(f[r]);
  }
}


// ---------------------------------------------------------------------------------------
// The windowed path ("brute"): tiles whose field is small everywhere (many small label runs:
// dense segmentations) do not build hulls at all.
//
// For a foreground row p of the run [a,b] let  B_p = min(F[p], border parabolas)  (the row's own
// parabola and the two parabolas of height 0 just outside the run, src/edt.hpp:233-242, :310-311).
// A row j at distance d = |p-j| can only lower the result if  c_d = w2*d^2 < B_p;  such a row lies
// inside the run (d is smaller than the distance to either border site), and for ANY row j of the
// column  c_d + F[j] >= c_d >= B_p  otherwise, because F >= 0.  Hence
//     result[p] = fl32( min( B_p,  min_{1<=d<=R} ( c_d + min(F[p-d], F[p+d]) ) ) )
// for every R with c_{R+1} >= B_p, with NO label test inside the window: rows of other runs, rows
// beyond a border and the +inf rows around the tile are harmless candidates.  (fl32 is monotone,
// so rounding commutes with the min; c_d is exact in fp64.)
//
// Mapping: lane = one column, a (64/TC)-th of a wave = the 32 rows of one band, processed as four
// blocks of 8 rows.  The rows p-R..p+R of a block live in a register window that grows by two LDS
// reads per step d; a step costs three (fp32) or four (fp64) vector instructions per row, and the
// loop ends for the whole wave as soon as c_d >= B_p for all of its rows -- adjacent columns share
// their borders and have similar fields, so the wave-wide window is close to the per-row one
// (dense 512^3 segmentation: mean window 8 rows per voxel, 15-20 per wave step).
// Every lane executes the same instructions: no pops, no bridge walks, no per-lane searches.
//
// Arithmetic: when c_d is exactly representable in fp32 for every d the path may use (X32; true
// for anisotropies like 1, 6, 30, 0.5, checked on the host), fl32(fl64(c_d + F)) equals the fp32
// sum (double rounding is innocuous for sums when the wide format has >= 2p+2 bits), so a
// candidate is ONE v_add_f32.  Otherwise candidates are formed in fp64 exactly as in the hull path.
// Non-negative floats order like their bit patterns, so minima of raw LDS values are integer minima.
// ---------------------------------------------------------------------------------------
constexpr int kBruteK = 32;  // register window radius = the rows of +inf padding on either side of the tile
constexpr int kBruteB = 8;   // rows per block

struct BruteLane {
  const float *tile;  // LDS tile, row 0 (one band of +inf rows on either side; rows >= n are +inf)
  int col;            // column inside the workgroup tile
  int band, row0, n;
  uint32_t rsw;       // run-start bits of the band (0 for a lane without a column)
  uint32_t brk;       // bit r: row r continues a run and is NOT flat against row r-1 (a "break")
  int blo_in;         // last break row in an earlier band of the column (-1: none)
  int bhi_out;        // first break row in a later band (>= n: none)
  int lo_in, hi_out;  // as in Lane
  bool live;          // the lane has a column (else its LDS words are whatever was there)
  double w2;
  float w2f;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define EDT_ANY(cond) (__ballot(cond) != 0ull)
#define EDT_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define EDT_ANY(cond) (cond)
#define EDT_OPAQUE(x) ((void)0)
#endif

EDT_LANE uint32_t f2u(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
EDT_LANE float u2f(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
EDT_LANE float minpos(float a, float b) {  // min of two non-negative floats (or +inf), no NaN handling
  const uint32_t ua = f2u(a), ub = f2u(b);
  return u2f(ua < ub ? ua : ub);
}
EDT_LANE float min3pos(float a, float b, float c) {
  uint32_t ua = f2u(a);
  const uint32_t ub = f2u(b), uc = f2u(c);
  ua = ua < ub ? ua : ub;
  return u2f(ua < uc ? ua : uc);
}

// Flat blocks need no window at all.  If every link (row r-1 -> r inside a run) at distance <= D around the
// block is flat, then for rows p, j of one run with |p-j| <= D:  F[j] >= F[p] - w2*|p-j| >= F[p] - c_|p-j|,
// i.e. no row within D improves p; rows further away cannot either once c_(D+1) >= B_p.  Returns D for the
// block of rows k0..k0+nrows-1: the distance to the nearest break of the whole column (breaks of other bands come
// from the same kind of scan over the bands as the run structure).
EDT_LANE int brute_flat_reach(const BruteLane &L, int k0, int nrows) {
  const uint32_t lowm = L.brk & (0xFFFFFFFFu >> (32 - nrows - k0));  // breaks at rows < k0 + nrows of this band
  const int h = lowm ? L.row0 + 31 - clz32(lowm) : L.blo_in;
  const uint32_t him = k0 + nrows < 32 ? L.brk & (0xFFFFFFFFu << (k0 + nrows)) : 0u;
  const int l = him ? L.row0 + ctz32(him) : L.bhi_out;
  const int p0 = L.row0 + k0;
  const int dlo = h >= 0 ? p0 - h : 4095;
  const int dhi = l < L.n ? l - p0 - nrows : 4095;  // (bhi_out >= n: no break above)
  const int d = dlo < dhi ? dlo : dhi;
  return d > 0 ? d : 0;
}

// The steps of the window as a compile-time recursion (every index into the register window is static):
// steps D and D+1 share one exit test; past the register-resident part the rows come straight from the tile.
// S = output stride: 1 = every row of the block is evaluated; 2 = a block is 16 rows of which the even ones
// are evaluated (the doubled grids of the voxel-graph transform, whose odd rows are never read again) --
// every row is a candidate either way.
template <int CW, bool X32, int S>
struct BruteSteps {
  static constexpr int K = kBruteK, B = kBruteB, TC = TileGeom<CW>::kCols, NR = S * B;  // NR rows per block
  const BruteLane &L;
  float (&w)[NR + 2 * K];
  float (&best)[B];
  double (&best64)[B];
  const float *PL0, *PL1, *PH0, *PH1;
  int k0;
  float bmaxf;
  double bmax64;
  int nb32;
  float w2f;   // per-block copies of L.w2f / L.w2 (see brute_band: keeps the c_d next to their use)
  double w2;
  uint32_t cbias;  // X32 only: 0 = c_d exact in fp32; 1 = c_d rounded (fma candidates, brute_f32e_prefix): the exit tests
                   // then use the float just below fl32(c_d), which is <= the true c_d

  // The exit bound follows the minima: a candidate at distance d is at least c_d, so once c_d >= every current
  // minimum of the wave's blocks nothing further away can lower any of them (ties change nothing).  Refreshed every
  // fourth step -- the bound of the block's START (its B_p) is the window of the INPUT field, the refreshed one that
  // of the RESULT, which is what the rows of a cell interior need: 24 -> 17 steps per block on the dense
  // segmentation's Y pass, 62 -> 27 on the doubled blobs of the voxel-graph configuration (host simulation).
  EDT_LANE_MEMBER void refresh_bound() {
    if (X32) {
      uint32_t m = f2u(best[0]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 1; i < B; ++i) { const uint32_t u = f2u(best[i]); m = u > m ? u : m; }
      bmaxf = u2f(m);
      bmax64 = (double)bmaxf;
    } else {
      double m = best64[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 1; i < B; ++i) m = fmax(m, best64[i]);
      bmax64 = m;
    }
  }

  template <int D>
  EDT_LANE_MEMBER void run() {
    if constexpr (D < K) {
      if constexpr (D > 1 && (D - 1) % 4 == 0) refresh_bound();
      // X32: a candidate is fl32(w2 * d^2 + F) -- ONE rounding of the exact sum (fma with an exact product).  Where c_d is
      // exactly representable that is the plain fp32 sum; where it is not, it is still the reference's value wherever the
      // fp64 sum the reference forms is exact (brute_f32e_prefix).  The exit test compares (a lower bound of) c_d.
      const float d1f = (float)(D * D), d2f = (float)((D + 1) * (D + 1));
      const float c1f = u2f(f2u(w2f * d1f) - cbias);
      const double c1 = w2 * (double)(D * D), c2 = w2 * (double)((D + 1) * (D + 1));
      if (X32 ? !EDT_ANY(c1f < bmaxf) : !EDT_ANY(c1 < bmax64)) return;
      // the rows that enter the window in these two steps
      w[K - D] = (D <= k0 ? PL0 : PL1)[(K - D) * TC];
      w[K + NR - 1 + D] = (D <= 32 - NR - k0 ? PH0 : PH1)[D * TC];
      w[K - D - 1] = (D + 1 <= k0 ? PL0 : PL1)[(K - D - 1) * TC];
      w[K + NR + D] = (D + 1 <= 32 - NR - k0 ? PH0 : PH1)[(D + 1) * TC];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int ii = 0; ii < B; ++ii) {
        // (the first and the last row of the block need the rows just requested: they come last)
        const int i = ii < B - 2 ? ii + 1 : (ii == B - 2 ? 0 : B - 1);
        const float m1 = minpos(w[K + S * i - D], w[K + S * i + D]);
        const float m2 = minpos(w[K + S * i - D - 1], w[K + S * i + D + 1]);
        // (sums of non-negative terms: integer minima, no canonicalisation of the operands)
        if (X32) best[i] = min3pos(best[i], fmaf(w2f, d1f, m1), fmaf(w2f, d2f, m2));
        else best64[i] = fmin(best64[i], fmin((double)m1 + c1, (double)m2 + c2));
      }
      run<D + 2>();
    } else if constexpr (X32 && S == 2) {
      // Windows beyond the register-resident part, blocks of 16 rows with the even ones evaluated: at step d
      // output i looks at the rows p0+2i-d (entered at step d-2i) and p0+2i+d (entered at step d-15+2i) --
      // the last 16 entries of either side: rings of 16 registers indexed by d mod 16, one step per exit test.
      constexpr int R = 16;
      static_assert((K % R) == 0 && NR == R, "ring phase / size");
      const int p0 = L.row0 + k0;
      float rlo[R], rhi[R];  // slot s: the row that entered at a step congruent to s (mod R)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int s = K - R + 2; s <= K; ++s) {  // the last 15 rows of the register-resident window on either side
        rlo[s % R] = w[K - s];
        rhi[s % R] = w[K + NR - 1 + s];
      }
      // (eight consecutive steps read one band's worth of rows on either side through one address each: see the
      // stride-1 form below; p0 and NR are multiples of 8 here too)
      const float *slo = L.tile, *shi = L.tile;
      // d^2 and its first difference as floats, stepped by exact additions (integers below 2^24); candidates are
      // fl32(w2 * d^2 + F) by fma, the exit test sees (a lower bound of) fl32(w2 * d^2)
      float ddf = (float)((K + 1) * (K + 1)), gdd = (float)(2 * (K + 1) + 1);
      for (int d0 = K + 1; d0 < 4096; d0 += R) {
        bool done = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e = 0; e < R; ++e, ddf += gdd, gdd += 2.0f) {  // step d = d0 + e;  d mod R == (1 + e) mod R
          const int d = d0 + e;
          if (e % 4 == 0) refresh_bound();
          if (!EDT_ANY(u2f(f2u(w2f * ddf) - cbias) < bmaxf)) { done = true; break; }
          if (e % 8 == 0) {
            int rl = p0 - d - 7, rh = p0 + NR - 1 + d;
            rl = rl < -32 ? -32 : rl;
            rh = rh > nb32 + 24 ? nb32 + 24 : rh;
            slo = L.tile + addr_tile<CW>(L.col, rl);
            shi = L.tile + addr_tile<CW>(L.col, rh);
          }
          const int sl = (1 + e) % R;
          rlo[sl] = slo[(7 - e % 8) * TC];
          rhi[sl] = shi[(e % 8) * TC];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
          for (int i = 0; i < B; ++i) {
            const float m = minpos(rlo[(sl - 2 * i + 2 * R) % R], rhi[(sl - (NR - 1) + 2 * i + 2 * R) % R]);
            best[i] = minpos(best[i], fmaf(w2f, ddf, m));
          }
        }
        if (done) break;
      }
    } else if constexpr (!X32 || S != 1) {
      // Windows beyond the register-resident part, fp64 candidates (the rarer form: 16 more registers of
      // minima) and strided blocks: one step at a time, every row straight from the tile.
      for (int d = K + 1; d < 4096; ++d) {
        if ((d & 3) == 1) refresh_bound();
        const double cd = w2 * (double)(d * d);  // exact
        if (!EDT_ANY(cd < bmax64)) break;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < B; ++i) {
          int rl = L.row0 + k0 + S * i - d, rh = L.row0 + k0 + S * i + d;
          rl = rl < -1 ? -1 : rl;        // row -1 and row nb32 are +inf rows
          rh = rh > nb32 ? nb32 : rh;
          const float m = minpos(L.tile[addr_tile<CW>(L.col, rl)], L.tile[addr_tile<CW>(L.col, rh)]);
          if (X32) best[i] = minpos(best[i], m + (float)cd);  // (X32: c_d is exact in fp32)
          else best64[i] = fmin(best64[i], (double)m + cd);
        }
      }
    } else {
      // Windows beyond the register-resident part: the same two-steps-per-test scheme as a rolled loop.
      // At step d row i looks at the rows p0+i-d and p0+i+d, i.e. at the B rows that entered the window
      // most recently on either side.  With two steps in flight that is a ring of B + 1 live rows: rings
      // of 16 registers indexed by d mod 16, which is static once the loop body covers 16 consecutive
      // steps.  Rows are addressed individually (a window of this size leaves the +inf padding: rows
      // beyond the column are clamped to the +inf rows -1 / nb32).
      constexpr int R = 16;
      static_assert((K % R) == 0 && B < R, "ring phase / size");
      const int p0 = L.row0 + k0;
      float rlo[R], rhi[R];  // slot s: the row that entered at a step congruent to s (mod R)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int s = K - B + 2; s <= K; ++s) {  // the last B - 1 rows of the register-resident window
        rlo[s % R] = w[K - s];
        rhi[s % R] = w[K + B - 1 + s];
      }
      // Eight consecutive steps read the rows p0-d-7 .. p0-d and p0+B-1+d .. p0+B+6+d: p0 is a multiple of 8 and
      // d = 1 (mod 8) at the first of them, so either stretch starts at a multiple of 8 and lies inside ONE band --
      // one address (with that band's column rotation) serves all eight rows through constant offsets.  A stretch
      // beyond the column is moved onto +inf rows of the padding (rows -32 .. -1 / n .. nb32 + 31 all hold +inf).
      const float *slo = L.tile, *shi = L.tile;
      float ddf = (float)((K + 1) * (K + 1)), gdd = (float)(2 * (K + 1) + 1);
      for (int d0 = K + 1; d0 < 4096; d0 += R) {
        bool done = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e = 0; e < R; e += 2) {  // steps d, d + 1 with d = d0 + e;  d mod R == (1 + e) mod R
          const int d = d0 + e;
          if (e % 4 == 0) refresh_bound();
          // d^2, (d+1)^2 by exact additions (see the stride-2 form above)
          const float d1f = ddf, d2f = ddf + gdd;
          ddf = d2f + (gdd + 2.0f);
          gdd += 4.0f;
          if (!EDT_ANY(u2f(f2u(w2f * d1f) - cbias) < bmaxf)) { done = true; break; }
          if (e % 8 == 0) {
            int rl = p0 - d - 7, rh = p0 + B - 1 + d;
            rl = rl < -32 ? -32 : rl;
            rh = rh > nb32 + 24 ? nb32 + 24 : rh;
            slo = L.tile + addr_tile<CW>(L.col, rl);
            shi = L.tile + addr_tile<CW>(L.col, rh);
          }
          const int s1 = (1 + e) % R, s2 = (2 + e) % R;
          rlo[s1] = slo[(7 - e % 8) * TC];
          rhi[s1] = shi[(e % 8) * TC];
          rlo[s2] = slo[(6 - e % 8) * TC];
          rhi[s2] = shi[(e % 8 + 1) * TC];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
          for (int i = 0; i < B; ++i) {
            // row p0+i-d entered at step d-i, row p0+i+d at step d-(B-1-i)
            const float m1 = minpos(rlo[(s1 - i + R) % R], rhi[(s1 - (B - 1 - i) + R) % R]);
            const float m2 = minpos(rlo[(s2 - i + R) % R], rhi[(s2 - (B - 1 - i) + R) % R]);
            best[i] = min3pos(best[i], fmaf(w2f, d1f, m1), fmaf(w2f, d2f, m2));  // (this form is X32 only)
          }
        }
        if (done) break;
      }
    }
  }
};

template <int CW, bool BB, bool X32, int S, class Store>
EDT_LANE void brute_band(const BruteLane &L, int epi, Store &&store) {
  constexpr int K = kBruteK, B = kBruteB, TC = TileGeom<CW>::kCols, NR = S * B;
  const int row0 = L.row0, n = L.n;
  const uint32_t rsw = L.rsw;
  // the lane's column in its own band and in the bands below / above (the band rotation of the tile
  // differs from band to band; inside a band rows are TC floats apart)
  const float *A0 = L.tile + addr_tile<CW>(L.col, row0);
  const float *Am = L.tile + addr_tile<CW>(L.col, row0 - 32);
  const float *Ap = L.tile + addr_tile<CW>(L.col, row0 + 32);
  const int nb32 = ((n + 31) >> 5) << 5;  // rows of the tile incl. the +inf rows that complete the last band
  // distance of the row before the band to the row before ITS run (+inf: that run has no border below)
  float dl = (BB || L.lo_in > 0) ? (float)(row0 - L.lo_in) : INFINITY;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int k0 = 0; k0 < 32; k0 += NR) {
    float w[NR + 2 * K];  // w[K + j] = F(row0 + k0 + j); the window grows by one row per side and step
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NR; ++j) w[K + j] = A0[(k0 + j) * TC];
    // rows below k0 come from this band while d <= k0, from the band below afterwards; likewise above
    const float *PL0 = A0 + (k0 - K) * TC, *PL1 = Am + (k0 + 32 - K) * TC;
    const float *PH0 = A0 + (k0 + NR - 1) * TC, *PH1 = Ap + (k0 + NR - 1 - 32) * TC;
    // ---- B_p of the block's rows ----
    // Distances to the border sites just outside the run, as floats counted from run start to run start
    // (exact small integers; +inf where there is no border on that side, which then stays +inf through
    // the count, the square and the product).  dl is carried from block to block.
    const uint32_t s8 = rsw >> k0;  // bit j: a run starts at row k0 + j
    float dlv[B];  // of the evaluated rows (every S-th row of the block)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < NR; ++j) {
      // a run that starts at row 0 of the column has a border below it only with black_border
      const float first = (BB || j > 0) ? 1.0f : (row0 + k0 > 0 ? 1.0f : INFINITY);
      dl = ((s8 >> j) & 1u) ? first : dl + 1.0f;
      if (j % S == 0) dlv[j / S] = dl;
    }
    float dr;  // distance to the first row of the next run, as seen from the row above the block
    {
      const uint32_t m = k0 + NR < 32 ? rsw & (0xFFFFFFFFu << (k0 + NR)) : 0u;
      const int e = m ? row0 + ctz32(m) : L.hi_out + 1;
      dr = (BB || e < n) ? (float)(e - (row0 + k0 + NR)) : INFINITY;
    }
    float best[B];
    double best64[B];
    uint32_t bmax = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = NR - 1; j >= 0; --j) {
      dr += 1.0f;
      if (j % S == 0) {
        const int i = j / S;
        const float dm = minpos(dlv[i], dr);
        // fl32(w2 * d^2) is one exact-product fp32 multiply (as in phase3_eval); background rows hold 0
        const float bord = L.w2f * (dm * dm);
        float b = minpos(w[K + j], bord);
        // rows that complete the last band (no real row holds +inf) and lanes without a column
        if (f2u(w[K + j]) == 0x7f800000u || !L.live) b = 0.0f;
        best[i] = b;
        if (!X32) best64[i] = (double)b;
        const uint32_t ub = f2u(b);
        bmax = ub > bmax ? ub : bmax;
      }
      if ((s8 >> j) & 1u) dr = 0.0f;  // (a set bit is a real row: the border site of the rows below it)
    }
    const float bmaxf = u2f(bmax);
    const double bmax64 = (double)bmaxf;
    // ---- flat neighbourhood: nothing within reach can improve any row of the wave's blocks ----
    bool open = true;  // some row of the wave may still improve
    EDT_STAT(13, k0, 1);
    {
      const int D = brute_flat_reach(L, k0, NR);
      const double cD = L.w2 * (double)((D + 1) * (D + 1));
      if (!EDT_ANY(cD < bmax64)) open = false;
#ifdef EDT_DIAG
      if (epi & 0x200) open = false;  // diagnostics: the fixed cost of the path (results are wrong)
#endif
    }
    // ---- the window: register-resident part (two steps per exit test), then straight from the tile ----
    if (open) {
      EDT_STAT(12, k0, 1);
      // The c_d = w2 * d^2 are wave-uniform but live in vector registers (no scalar fp multiply); hoisted
      // out of the block loop all 32 of them would stay alive across it.  An opaque copy of w2 per block
      // keeps each product in the step that uses it.
      float w2f = L.w2f;
      double w2 = L.w2;
      EDT_OPAQUE(w2f);
      EDT_OPAQUE(w2);
      BruteSteps<CW, X32, S> steps{L, w, best, best64, PL0, PL1, PH0, PH1, k0, bmaxf, bmax64, nb32, w2f, w2,
                                    (epi & 0x800) ? 1u : 0u};
      steps.template run<1>();
    }
    // ---- epilogue (src/edt.hpp:47-53, :599-601) and the rows leave ----
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < B; ++i) {
      float r = X32 ? best[i] : (float)best64[i];
      if ((epi & kLaneEpiToInf) && r >= 3.402823466e+38f) r = INFINITY;
      best[i] = r;
    }
    if (epi & kLaneEpiSqrt) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int i = 0; i < B; ++i) best[i] = sqrtf(best[i]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < B; ++i) store(row0 + k0 + S * i, best[i]);
  }
}


// The largest D <= want such that c_d = w2 * d^2 is exactly representable in fp32 for every d <= D (then the
// candidates of the windowed path may be fp32 sums up to windows of D rows).
inline int brute_exact_prefix(float w, int want) {
  const double w2 = (double)(w * w);
  int d = 0;
  while (d < want) {
    const double c = w2 * (double)(d + 1) * (double)(d + 1);
    if ((double)(float)c != c || !(c < 3.0e38)) break;
    ++d;
  }
  return d;
}
inline bool brute_exact32(float w, int want) { return brute_exact_prefix(w, want) >= want; }

// fp32 candidates for voxel sizes whose c_d = w2 * d^2 are NOT exactly representable in fp32.  The reference forms
// fl32( fl64( w2 * d^2 + F ) ): the product is exact in fp64 (24 x 24 bits at most), the sum is rounded to fp64, the
// result narrowed.  An fp32 fma gives fl32( w2 * d^2 + F ) -- one rounding of the exact sum -- which is the same value
// whenever the fp64 sum is EXACT, i.e. whenever all set bits of w2 * d^2 and of F fit one 53-bit window.  The lowest set
// bit of w2 * d^2 is at least that of w2; the lowest set bit of a non-zero field value F >= fmin is at least
// 2^(ilogb(fmin) - 23); on a tile of the windowed path F <= c_T and d <= T, so every sum is below 2 * c_T.  Returns the
// largest T <= want for which 2 * w2 * T^2 < 2^(low + 53), low = the smaller of the two lowest-bit exponents
// (0: never -- no lower bound on the field is known, or the voxel sizes are too far apart).
// fmin: a lower bound of the non-zero values the pass reads (pass Y: fl32(wx^2); pass Z: the smaller of that and fl32(wy^2)).
inline int brute_f32e_prefix(float w, float fmin, int want) {
  const float w2f = w * w;
  if (!(w2f >= 1.17549435e-38f) || !(fmin >= 1.17549435e-38f) || !((double)w2f < 1.0e30) || !(fmin < 3.0e38f)) return 0;
  uint32_t bits;
  memcpy(&bits, &w2f, 4);
  const uint32_t man = (bits & 0x7FFFFFu) | 0x800000u;          // 24-bit significand of w2 (normal)
  const int ew = (int)((bits >> 23) & 0xFF) - 127 - 23;           // exponent of its last bit
  int tz = 0;
  while (!((man >> tz) & 1u)) ++tz;
  const int low_w = ew + tz;                                     // lowest SET bit of w2
  const int low_f = ilogbf(fmin) - 23;                           // lowest possible set bit of a field value >= fmin
  const int low = low_w < low_f ? low_w : low_f;
  const double cap = ldexp(1.0, low + 53);
  int T = want > 4095 ? 4095 : want;                             // (d^2 as an exact float: d < 4096)
  while (T > 0 && !(2.0 * (double)w2f * (double)T * (double)T < cap)) --T;
  return T;
}

}  // namespace edt_lane
