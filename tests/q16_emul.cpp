// tests/q16_emul.cpp -- TEST FIXTURE: lane-by-lane host emulation of the 16-bit integer column pass
// (euclidean-distance-transform-3d_amd/csrc/edt_colq16_lane.h + the workgroup logic of edt_colq16.hip).
//
// Compiles the SAME per-lane header the HIP kernel is built from with g++ and plays the fill, the scans and every block of
// every tile in sequence, ordinary arrays standing in for LDS, so that the CPU tier (tests/test_q16_logic.py) can hold the
// packed border counters, the break bits, the window steps and the tile qualification against the oracle without a GPU.
// Never linked into the product library.
//
//   g++ -O2 -ffp-contract=off -shared -fPIC -I<csrc> tests/q16_emul.cpp -o libq16_emul.so
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define EDT_LANE static inline
#define EDT_LANE_MEMBER inline
#include "edt_colq16_lane.h"

using namespace edt_q16;

namespace {

bool g_no_wide = false;
bool g_full_wide = false;  // tests: a tile beyond 16 bits always as two wide passes over all its columns

// The wide form of a tile (edt_colq16.hip, go_wide): two half-tiles of 16 columns, one 32-bit value per image word.
template <bool BB>
void tile_pass_wide(const float *Fin, const uint16_t *codes, const uint32_t *rsbits, float *out, int64_t sx, int n, int64_t x0,
                    float q, uint32_t a, uint32_t ain, int epi, const uint16_t *plane_in, const uint8_t *row_in_plane,
                    uint32_t dmaxw, uint32_t nlimw, uint32_t fwmax_bits, uint32_t colmask = 0xFFFFFFFFu) {
  const int NB = (n + 31) / 32, nb32 = NB * 32;
  const int cols_left = (int)(sx - x0);
  std::vector<uint32_t> rsp((size_t)NB * 32, 0), lohi((size_t)NB * 32, 0);
  for (int band = 0; band < NB; ++band)
    for (int col = 0; col < 32; ++col)
      rsp[(size_t)band * 32 + col] = col < cols_left ? rsbits[(size_t)band * sx + x0 + col] : 0u;
  uint16_t *lohi16 = reinterpret_cast<uint16_t *>(lohi.data());
  for (int t = 0; t < 32; ++t) {
    scan_runs_lo(rsp.data() + t, 32, NB, lohi16 + 2 * t, 64);
    scan_runs_hi(rsp.data() + t, 32, NB, n, lohi16 + 2 * t + 1, 64);
  }
  // the passes: columns 0..15 and 16..31 (two wide passes over the whole tile), or the marked columns alone (colmask)
  const bool subset = colmask != 0xFFFFFFFFu;
  for (int h = 0; h < (subset ? 1 : (cols_left > 16 ? 2 : 1)); ++h) {
    int wcol[16];
    for (int i = 0; i < 16; ++i) {
      int c = 32;
      if (!subset) c = 16 * h + i;
      else {
        uint32_t m = colmask;
        for (int k = 0; k < i && m; ++k) m &= m - 1;
        if (m) c = __builtin_ctz(m);
      }
      wcol[i] = c < cols_left ? c : 32;
    }
    std::vector<uint32_t> img((size_t)(nb32 + 2 * kPad) * kRowWords, kInfW), bm(16 * 6, 0);
    for (int row = 0; row < n; ++row)
      for (int c = 0; c < 16; ++c) {
        const int col = wcol[c];
        uint32_t v = 0;
        if (col < 32) {
          if (plane_in && row_in_plane[row]) {
            v = plane_in[(int64_t)row * sx + x0 + col];
            if (v == 0xFFFFu) v = kInfW;  // (+inf)
          } else if (codes) {
            const uint32_t k = codes[(int64_t)row * sx + x0 + col];
            v = k == 0xFFFFu ? kInfW : k * k * ain;  // (0xFFFF: no boundary in the row)
          } else {
            (void)wide_value(Fin[(int64_t)row * sx + x0 + col], q, 1.0f / q, nlimw, fwmax_bits, v);
          }
        }
        img[(size_t)(row + kPad) * kRowWords + c] = v;
      }
    for (int u = 0; u < 16 * NB; ++u) {
      const int c = u & 15, band = u >> 4;
      const int valid = n - 32 * band;
      const uint32_t bits = band_breaks<true>(img.data() + (size_t)(32 * band + kPad) * kRowWords + c, a, band == 0, valid < 32 ? valid : 32);
      bm[c * 6 + 1 + (band >> 3)] |= bits << (4 * (band & 7));
    }
    for (int sb = 0; sb * 32 < nb32; ++sb)
      for (int lane = 0; lane < 64; ++lane) {
        const int cw = lane & 15, bq = lane >> 4;
        const int colc = wcol[cw] < 32 ? wcol[cw] : 0;
        Block L;
        L.img = img.data();
        L.cp = cw;
        L.p0 = 32 * sb + 8 * bq;
        const int s = L.p0 >> 5;
        L.n = n;
        L.nb32 = nb32;
        L.rswA = wcol[cw] < 32 ? rsp[(size_t)s * 32 + colc] : 0u;
        const uint32_t lh = lohi[(size_t)s * 32 + colc];
        L.loA = (int)(lh & 0xFFFFu) - 1;
        L.hiA = (int)(lh >> 16) - 1;
        L.rswB = 0u;
        L.loB = L.hiB = 0;
        L.a = a;
        L.dmax = dmaxw;
        {
          const int gi = L.p0 >> 3, wi = gi >> 5, sh = gi & 31;
          const uint32_t *m = bm.data() + cw * 6 + wi;
          const uint32_t e0 = m[0], e1 = m[1], e2 = m[2];
          const uint32_t lo = (uint32_t)((((uint64_t)e1 << 32) | e0) >> sh);
          const uint32_t hi = (uint32_t)((((uint64_t)e2 << 32) | e1) >> sh);
          L.win = ((uint64_t)hi << 32) | lo;
          L.reach = flat_reach_full(bm.data() + cw * 6, gi);
          L.bmw = bm.data() + cw * 6;
        }
        pk best[kB];
        block_eval<BB, 1, true>(L, best);
        for (int j = 0; j < kB; ++j) {
          const int row = L.p0 + j, col = wcol[cw];
          if (row >= n || col >= 32) continue;
          // (+inf: FLT_MAX between the passes, +INF behind the last one -- epi bit 0 = toinfinite)
          float v = best[j] >= kInfW ? ((epi & 1) ? INFINITY : 3.402823466e+38f) : (float)best[j] * q;
          if (epi & 2) v = sqrtf(v);
          out[(int64_t)row * sx + x0 + col] = v;
        }
      }
  }
}

// one tile: columns x0 .. x0+31.  Returns 0 if the tile does not qualify (nothing is written then), 1 if it was worked on in
// its 16-bit form, 2 in the wide form (fp32 results in `out` whatever plane_out says).
template <bool BB, int S = 1>
int tile_pass(const float *Fin, const uint16_t *codes, const uint32_t *rsbits, float *out, int64_t sx, int n, int64_t x0,
               float q, uint32_t a, uint32_t ain, int epi, long *steps_taken, const uint16_t *plane_in = nullptr,
               const uint8_t *row_in_plane = nullptr, uint16_t *plane_out = nullptr) {
  const int NB = (n + 31) / 32, nb32 = NB * 32;
  const int cols_left = (int)(sx - x0);
  const uint32_t dmax = q16_dmax(a), nlim = a * dmax * dmax;
  uint32_t kmax = 0;
  while ((uint64_t)(kmax + 1) * (kmax + 1) * ain <= nlim && kmax < 65534u) ++kmax;
  // the wide form's range (edt_colq16.hip: launch_column_pass_q16)
  WideRange wr = {0u, 0u, false};
  if (!g_no_wide) wr = q16_wide_range(a, q, n, BB, nlim);
  const uint32_t dmaxw = wr.nlim ? wr.dmax : dmax, nlimw = wr.nlim ? wr.nlim : nlim;
  const bool inf_ok = wr.inf;
  uint32_t kmaxw = kmax;
  while ((uint64_t)(kmaxw + 1) * (kmaxw + 1) * ain <= nlimw && kmaxw < 65534u) ++kmaxw;
  const float fw = (float)nlimw * q;
  uint32_t fwmax_bits;
  memcpy(&fwmax_bits, &fw, sizeof(fw));
  bool over = false;
  uint32_t overmask = 0;  // the tile's columns that hold a value beyond the 16-bit form
  std::vector<uint32_t> img((size_t)(nb32 + 2 * kPad) * kRowWords, 0xFFFFFFFFu);
  std::vector<uint32_t> rsp((size_t)NB * 32, 0), lohi((size_t)NB * 32, 0), bm(16 * 6, 0);
  bool bad = false;
  // ---- fill (edt_colq16.hip, phase 0) ----
  auto put = [&](int row, int col, uint32_t v) {
    uint32_t &wd = img[(size_t)(row + kPad) * kRowWords + col / 2];
    wd = (col & 1) ? ((wd & 0xFFFFu) | (v << 16)) : ((wd & 0xFFFF0000u) | (v & 0xFFFFu));
  };
  for (int row = 0; row < nb32; ++row)
    for (int col = 0; col < 32; ++col) {
      uint32_t v = 0;
      if (row >= n) v = 0xFFFFu;
      else if (col < cols_left) {
        if (plane_in && row_in_plane[row]) {  // (mixed input: this row of the tile is in the 16-bit plane)
          v = plane_in[(int64_t)row * sx + x0 + col];
          if (v > nlim) { over = true; overmask |= 1u << col; }
        } else if (codes) {
          const uint32_t k = codes[(int64_t)row * sx + x0 + col];
          if (k > kmax) { over = true; overmask |= 1u << col; }
          if (k > kmaxw && !(k == 0xFFFFu && inf_ok)) bad = true;  // (0xFFFF: no boundary in the row -- +inf, which the wide form may carry)
          v = (uint32_t)(uint16_t)((uint16_t)(k * k) * (uint16_t)ain);  // (wraps like the packed multiply; unused if bad)
        } else {
          const float f = Fin[(int64_t)row * sx + x0 + col];
          const float flim = (float)nlim + 1.0f;
          const float tq = fminf(f * (1.0f / q), flim);
          const uint32_t u = (uint32_t)(tq + 0.5f);
          const float e = fmaf(-(float)u, q, f);
          if (u > nlim) {  // (the conversion was clamped: the wide form's conversion gives the verdict)
            over = true;
            overmask |= 1u << col;
            uint32_t uw;
            if (!wide_value(f, q, 1.0f / q, nlimw, fwmax_bits, uw) || (uw == kInfW && !inf_ok)) bad = true;
          } else if (!(fabsf(e) == 0.0f)) bad = true;
          v = u & 0xFFFFu;
        }
      }
      put(row, col, v);
    }
  for (int band = 0; band < NB; ++band)
    for (int col = 0; col < 32; ++col)
      rsp[(size_t)band * 32 + col] = col < cols_left ? rsbits[(size_t)band * sx + x0 + col] : 0u;
  const bool go_wide = S == 1 && !bad && over && nlimw > nlim;
  uint32_t redo = 0;  // (edt_colq16.hip: at most 16 marked columns -- the 16-bit form with those columns at +inf, then ONE wide pass over them)
  if (go_wide) {
    if (__builtin_popcount(overmask) > 16 || g_full_wide) {
      if constexpr (S == 1)
        tile_pass_wide<BB>(Fin, codes, rsbits, out, sx, n, x0, q, a, ain, epi, plane_in, row_in_plane, dmaxw, nlimw, fwmax_bits);
      return 2;
    }
    redo = overmask;
    for (int row = 0; row < n; ++row)
      for (uint32_t m = redo; m; m &= m - 1) put(row, __builtin_ctz(m), 0u);
    plane_out = nullptr;  // (fp32 results for the whole tile)
  }
  if (bad || (over && redo == 0)) {
    // mixed input: the rows the tile has in the plane become fp32 values (the fp32 kernel reads F)
    if (plane_in)
      for (int row = 0; row < n; ++row)
        if (row_in_plane[row])
          for (int col = 0; col < 32 && col < cols_left; ++col)
            out[(int64_t)row * sx + x0 + col] = plane_in[(int64_t)row * sx + x0 + col] == 0xFFFFu
                                                    ? 3.402823466e+38f : (float)plane_in[(int64_t)row * sx + x0 + col] * q;  // (0xFFFF: +inf)
    return 0;
  }
  // ---- scans + breaks (phase 1) ----
  uint16_t *lohi16 = reinterpret_cast<uint16_t *>(lohi.data());
  for (int t = 0; t < 32; ++t) {
    scan_runs_lo(rsp.data() + t, 32, NB, lohi16 + 2 * t, 64);
    scan_runs_hi(rsp.data() + t, 32, NB, n, lohi16 + 2 * t + 1, 64);
  }
  const pk apk = pk_both(a);
  for (int u = 0; u < 16 * NB; ++u) {
    const int cp = u & 15, band = u >> 4;
    const int valid = n - 32 * band;
    const uint32_t bits = band_breaks(img.data() + (size_t)(32 * band + kPad) * kRowWords + cp, apk, band == 0, valid < 32 ? valid : 32);
    bm[cp * 6 + 1 + (band >> 3)] |= bits << (4 * (band & 7));
  }
  // ---- blocks (phase 2): a wave works on 16 pairs x four consecutive blocks (of 8 rows, or -- S = 2 -- of 16) ----
  for (int sb = 0; sb * 32 * S < nb32; ++sb)
    for (int lane = 0; lane < 64; ++lane) {
      const int cp = lane & 15, bq = lane >> 4;
      Block L;
      L.img = img.data();
      L.cp = cp;
      L.p0 = 32 * S * sb + 8 * S * bq;
      if (L.p0 >= nb32) continue;
      const int s = L.p0 >> 5;  // the block's band
      L.n = n;
      L.nb32 = nb32;
      L.rswA = rsp[(size_t)s * 32 + 2 * cp];
      L.rswB = rsp[(size_t)s * 32 + 2 * cp + 1];
      const uint32_t lhA = lohi[(size_t)s * 32 + 2 * cp], lhB = lohi[(size_t)s * 32 + 2 * cp + 1];
      L.loA = (int)(lhA & 0xFFFFu) - 1;
      L.hiA = (int)(lhA >> 16) - 1;
      L.loB = (int)(lhB & 0xFFFFu) - 1;
      L.hiB = (int)(lhB >> 16) - 1;
      L.a = a;
      L.dmax = dmax;
      {
        const int gi = L.p0 >> 3, wi = gi >> 5, sh = gi & 31;
        const uint32_t *m = bm.data() + cp * 6 + wi;
        const uint32_t e0 = m[0], e1 = m[1], e2 = m[2];
        const uint32_t lo = (uint32_t)((((uint64_t)e1 << 32) | e0) >> sh);
        const uint32_t hi = (uint32_t)((((uint64_t)e2 << 32) | e1) >> sh);
        L.win = ((uint64_t)hi << 32) | lo;
        L.bmw = bm.data() + cp * 6;
      }
      pk best[kB];
      block_eval<BB, S>(L, best);
      (void)steps_taken;
      for (int j = 0; j < kB; ++j) {
        const int row = L.p0 + S * j;
        if (row >= n) continue;
        for (int h = 0; h < 2; ++h) {
          const int col = 2 * cp + h;
          if (col >= cols_left || ((redo >> col) & 1u)) continue;  // (a marked column's result comes from its wide pass)
          if (plane_out) { plane_out[(int64_t)row * sx + x0 + col] = (uint16_t)((best[j] >> (16 * h)) & 0xFFFFu); continue; }
          float v = (float)((best[j] >> (16 * h)) & 0xFFFFu) * q;
          if (epi & 2) v = sqrtf(v);
          out[(int64_t)row * sx + x0 + col] = v;
        }
      }
    }
  if (redo != 0) {
    if constexpr (S == 1)
      tile_pass_wide<BB>(Fin, codes, rsbits, out, sx, n, x0, q, a, ain, epi, plane_in, row_in_plane, dmaxw, nlimw, fwmax_bits, redo);
    return 3;
  }
  return 1;
}

}  // namespace

// (tests: the 16-bit form only, as in round 4)
extern "C" void q16_emul_set_no_wide(int v) { g_no_wide = v != 0; }
extern "C" void q16_emul_set_full_wide(int v) { g_full_wide = v != 0; }

// labels [n][sx] uint32 (the run structure along the scan axis), Fin [n][sx] fp32 or codes [n][sx] u16 (exactly one of
// them), out [n][sx].  tile_ok[i] = 1 if x-tile i qualified (and was written).  Returns the number of tiles that did.
extern "C" int q16_emul_column_pass(const uint32_t *labels, const float *Fin, const uint16_t *codes, float *out, int64_t sx,
                                    int64_t n, float q, uint32_t a, uint32_t ain, int bb, int epi, uint8_t *tile_ok) {
  const int NB = (int)((n + 31) / 32);
  std::vector<uint32_t> rs((size_t)NB * sx, 0);
  for (int64_t x = 0; x < sx; ++x)
    for (int64_t y = 0; y < n; ++y) {
      const uint32_t lab = labels[y * sx + x];
      const bool start = (y == 0) || lab != labels[(y - 1) * sx + x];
      if (start) rs[(size_t)(y / 32) * sx + x] |= 1u << (y % 32);
    }
  int ok = 0;
  for (int64_t x0 = 0, i = 0; x0 < sx; x0 += 32, ++i) {
    const int r = bb ? tile_pass<true>(Fin, codes, rs.data(), out, sx, (int)n, x0, q, a, ain, epi, nullptr)
                      : tile_pass<false>(Fin, codes, rs.data(), out, sx, (int)n, x0, q, a, ain, epi, nullptr);
    tile_ok[i] = (uint8_t)r;
    ok += r ? 1 : 0;
  }
  return ok;
}

// the 16-bit plane: plane_in + row_in_plane[n] (rows of every tile that come from the plane; Fin serves the others),
// plane_out (the results stay 16-bit: written there instead of out)
extern "C" int q16_emul_column_pass_plane(const uint32_t *labels, const float *Fin, const uint16_t *codes, float *out, int64_t sx,
                                          int64_t n, float q, uint32_t a, uint32_t ain, int bb, int epi, uint8_t *tile_ok,
                                          const uint16_t *plane_in, const uint8_t *row_in_plane, uint16_t *plane_out) {
  const int NB = (int)((n + 31) / 32);
  std::vector<uint32_t> rs((size_t)NB * sx, 0);
  for (int64_t x = 0; x < sx; ++x)
    for (int64_t y = 0; y < n; ++y) {
      const uint32_t lab = labels[y * sx + x];
      const bool start = (y == 0) || lab != labels[(y - 1) * sx + x];
      if (start) rs[(size_t)(y / 32) * sx + x] |= 1u << (y % 32);
    }
  int ok = 0;
  for (int64_t x0 = 0, i = 0; x0 < sx; x0 += 32, ++i) {
    const int r = bb ? tile_pass<true>(Fin, codes, rs.data(), out, sx, (int)n, x0, q, a, ain, epi, nullptr, plane_in, row_in_plane, plane_out)
                      : tile_pass<false>(Fin, codes, rs.data(), out, sx, (int)n, x0, q, a, ain, epi, nullptr, plane_in, row_in_plane, plane_out);
    tile_ok[i] = (uint8_t)r;
    ok += r ? 1 : 0;
  }
  return ok;
}

// output stride 2: only the even rows are evaluated and written (the doubled grids of the voxel-graph transform)
extern "C" int q16_emul_column_pass_even(const uint32_t *labels, const float *Fin, float *out, int64_t sx, int64_t n, float q,
                                         uint32_t a, int bb, int epi, uint8_t *tile_ok) {
  const int NB = (int)((n + 31) / 32);
  std::vector<uint32_t> rs((size_t)NB * sx, 0);
  for (int64_t x = 0; x < sx; ++x)
    for (int64_t y = 0; y < n; ++y) {
      const uint32_t lab = labels[y * sx + x];
      const bool start = (y == 0) || lab != labels[(y - 1) * sx + x];
      if (start) rs[(size_t)(y / 32) * sx + x] |= 1u << (y % 32);
    }
  int ok = 0;
  for (int64_t x0 = 0, i = 0; x0 < sx; x0 += 32, ++i) {
    const int r = bb ? tile_pass<true, 2>(Fin, nullptr, rs.data(), out, sx, (int)n, x0, q, a, 1u, epi, nullptr)
                      : tile_pass<false, 2>(Fin, nullptr, rs.data(), out, sx, (int)n, x0, q, a, 1u, epi, nullptr);
    tile_ok[i] = (uint8_t)r;
    ok += r ? 1 : 0;
  }
  return ok;
}

extern "C" int q16_emul_quantum(const float *w, int naxes, float *q, uint32_t *a) {
  const Quantum Q = quantum_of(w, naxes);
  *q = Q.q;
  for (int i = 0; i < 3; ++i) a[i] = Q.a[i];
  return Q.ok ? 1 : 0;
}
