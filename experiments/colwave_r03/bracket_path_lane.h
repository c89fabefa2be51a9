// experiments/colwave_r03/bracket_path_lane.h -- the bracket path ("mono") of the fp32 column kernel: per-lane code as it stood
// in csrc/edt_colwave_lane.h at the end of round 3 (namespace edt_lane, after brute_band).  NOT compiled into the library.
// See README.md in this directory.

// ---------------------------------------------------------------------------------------
// The bracket path ("mono"): tiles whose field is too large for short windows but whose arithmetic is exact.
//
// As in the windowed path, result[p] = min(B_p, G[p]) with B_p = min(F[p], border parabolas) and
//     G[p] = min over ALL rows j of the column of  M[p][j] = c_|p-j| + F[j]
// (rows outside p's run and the +inf rows around the column are harmless candidates, see above).  The matrix M is
// strictly Monge -- M[q][j2] - M[q][j1] = M[p][j2] - M[p][j1] - 2*w2*(q-p)*(j2-j1) for q > p, j2 > j1 -- so every
// minimiser of a row p lies at or after every minimiser of an earlier row and at or before every minimiser of a later
// one: once rows a < b are done, a row between them only needs the candidates A(a) .. A(b) between their argmins
// (ANY argmin serves as either end).  The cost of a row is then the length of its bracket, not the window: the
// brackets of the rows of one level tile the column.  Levels, lane = (column, band of 32 rows):
//   0  the band's first row ("anchor"): the window search of the windowed path for ONE row, with its argmin
//      (mono_anchor); the anchors' argmins travel through one plane of LDS words,
//   1  row 16 between this band's anchor and the next one's,   2  rows 8 and 24,
//   3  the four gaps of seven rows between those, all seven rows of a gap against the gap's bracket at once.
// Rows whose envelope value cannot beat B_p are the only ones a truncated search may get "wrong" (an anchor's window
// ends once c_d >= min(B_a, best): candidates beyond it lie beyond a border site of every row that could want them),
// and their result is B_p either way -- the argument is spelled out in DESIGN.md.
//
// Exactness: the path is only taken where every candidate value is computed without rounding that could reorder
// candidates: c_d exactly representable in fp32 for every d it can meet (brute_exact_prefix), and field values below
// 2^23 * w2 (one ulp of any candidate is then at most w2 < 2*w2 = the least amount by which the order of two
// candidates changes from one row to the next).  The host (launcher) turns both into the bit pattern of the largest
// tile maximum the path accepts.  c_d and its first differences are updated by exact fp32 additions.
// ---------------------------------------------------------------------------------------
struct MonoLane {
  const float *tile;  // LDS tile, row 0 (one band of +inf rows on either side; rows >= n are +inf)
  int col, band, row0, n;
  uint32_t rsw;       // run-start bits of the band
  int lo_in, hi_out;  // as in Lane
  bool live;          // the lane has a column and the band has rows
  float w2f;
};

// B_p of row r (relative) of the band: the row's own value and the parabolas of height 0 just outside its run
template <int CW, bool BB>
EDT_LANE float mono_bound(const MonoLane &L, int r, float Fp) {
  const int p = L.row0 + r;
  const uint32_t lowm = L.rsw & (0xFFFFFFFFu >> (31 - r));
  const int s = lowm ? L.row0 + 31 - clz32(lowm) : L.lo_in;             // first row of p's run
  const uint32_t him = r < 31 ? (L.rsw & (0xFFFFFFFEu << r)) : 0u;
  const int e = him ? L.row0 + ctz32(him) - 1 : L.hi_out;                // last row of p's run
  float dm = INFINITY;
  if (BB || s > 0) dm = (float)(p - s + 1);
  if (BB || e < L.n - 1) dm = fminf(dm, (float)(e + 1 - p));
  return minpos(Fp, L.w2f * (dm * dm));  // (+inf * +inf = +inf: no border at all)
}

// Level 0: the band's first row a.  Returns G-or-bound information in `best` (the least candidate met, the row's own
// value included) and its row in `arg`.  The loop runs for the whole wave until c_d >= min(B_a, best) for every lane.
template <int CW>
EDT_LANE void mono_anchor(const MonoLane &L, float Ba, float Fa, float &best, int &arg) {
  constexpr int TC = TileGeom<CW>::kCols;
  const int a = L.row0;
  const int nb32 = ((L.n + 31) >> 5) << 5;
  best = Fa;
  int off = 0;
  float bnd = L.live ? minpos(Ba, Fa) : 0.0f;
#if defined(EDT_MONO_SKIP) && (EDT_MONO_SKIP & 1)
  bnd = 0.0f;  // (cost measurement: no anchor search; wrong results)
#endif
  // (rows a - d, a + d for d <= 32 exist in the LDS image whatever a is: one band of +inf rows on either side)
  const float *P = L.tile + addr_tile<CW>(L.col, a);
  const float *Pm = L.tile + addr_tile<CW>(L.col, a - 32);  // the band below (its own column rotation)
  float c = L.w2f, g = 3.0f * L.w2f;  // c_1 and c_2 - c_1 (exact)
  const float g2 = L.w2f + L.w2f;
  int d = 1;
  for (; d <= 32; ++d) {
    if (!EDT_ANY(c < bnd)) break;
    const float flo = Pm[(32 - d) * TC], fhi = d < 32 ? P[d * TC] : L.tile[addr_tile<CW>(L.col, a + 32)];
    const float m = minpos(flo, fhi);
    const float cand = m + c;
    const bool win = f2u(cand) < f2u(best);
    const int side = f2u(flo) <= f2u(fhi) ? -d : d;
    best = minpos(best, cand);
    bnd = minpos(bnd, cand);
    off = win ? side : off;
    c += g;
    g += g2;
  }
  if (d > 32) {
    for (;; ++d) {
      if (!EDT_ANY(c < bnd)) break;
      int rl = a - d, rh = a + d;
      rl = rl < -1 ? -1 : rl;        // rows -1 and nb32 are +inf rows
      rh = rh > nb32 ? nb32 : rh;
      const float flo = L.tile[addr_tile<CW>(L.col, rl)], fhi = L.tile[addr_tile<CW>(L.col, rh)];
      const float m = minpos(flo, fhi);
      const float cand = m + c;
      const bool win = f2u(cand) < f2u(best);
      const int side = f2u(flo) <= f2u(fhi) ? -d : d;
      best = minpos(best, cand);
      bnd = minpos(bnd, cand);
      off = win ? side : off;
      c += g;
      g += g2;
    }
  }
  arg = a + off;
  // (a clamped far row is a +inf row: it never wins, so arg is a real row or a itself)
}

// One row p against the candidates lo .. hi (rows of the column, lo <= hi): least value (the incoming best included)
// and its row.  Per-lane trip counts: the loop is an ordinary divergent one.
template <int CW>
EDT_LANE void mono_row(const MonoLane &L, int p, int lo, int hi, float &best, int &arg) {
  // c = w2 * (p - j)^2 and its difference to the next candidate, stepped exactly: g = c_(j+1) - c_j = w2 * (1 - 2*(p - j))
  const float dj = (float)(p - lo);
  float c = L.w2f * (dj * dj);
  float g = L.w2f * (1.0f - (dj + dj));
  const float g2 = L.w2f + L.w2f;
  for (int j = lo; j <= hi; ++j) {
    const float F = L.tile[addr_tile<CW>(L.col, j)];
    const float cand = F + c;
    const bool win = f2u(cand) < f2u(best);
    best = minpos(best, cand);
    arg = win ? j : arg;
    c += g;
    g += g2;
  }
}

// Seven rows p0+1 .. p0+7 (best[0..6]) against the candidates lo .. hi, no argmin.
template <int CW>
EDT_LANE void mono_gap(const MonoLane &L, int p0, int lo, int hi, float *best) {
  float d[8];  // d[i] = (p0 + 1 + i) - j as a float (|d| < 2^12), stepped by -1 per candidate
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 8; ++i) d[i] = (float)(p0 + 1 + i - lo);
  for (int j = lo; j <= hi; ++j) {
    const float F = L.tile[addr_tile<CW>(L.col, j)];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 7; ++i) {
      // fl32(w2 * d^2 + F): the product is exact (c_d is representable), one rounding -- the same value as c_d + F
      const float cand = fmaf(L.w2f, d[i] * d[i], F);
      best[i] = minpos(best[i], cand);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 8; ++i) d[i] -= 1.0f;
  }
}

// Levels 1-3 of one lane's band.  A0 / A32: argmins of this band's anchor and of the next band's (n - 1 where there
// is none); best0 = the anchor's least candidate.  store(row, value) takes the finished rows.
template <int CW, bool BB, class Store>
EDT_LANE void mono_band(const MonoLane &L, float best0, float B0, int A0, int A32, int epi, Store &&store) {
  constexpr int TC = TileGeom<CW>::kCols;
  const int row0 = L.row0, n = L.n;
  const float *own = L.tile + addr_tile<CW>(L.col, row0);
  auto done = [&](int r, float v, float B) {
    v = minpos(v, B);
    store(row0 + r, finish_f(v, epi));
  };
  if (!L.live) return;
  done(0, best0, B0);
  // (rows beyond the column do not exist: nothing is computed for them and their "argmin" is the last row)
  auto level_row = [&](int r, int lo, int hi) -> int {
    if (row0 + r >= n) return n - 1;
#if defined(EDT_MONO_SKIP) && (EDT_MONO_SKIP & 2)
    hi = lo;  // (cost measurement: one candidate per level row; wrong results)
#endif
    const float Fp = own[r * TC];
    float best = Fp;
    int arg = row0 + r;
    mono_row<CW>(L, row0 + r, lo, hi, best, arg);
    done(r, best, mono_bound<CW, BB>(L, r, Fp));
    return arg;
  };
  // (an inverted bracket can only come from rows whose value is their bound anyway: order the ends)
  auto lo_of = [](int x, int y) { return x < y ? x : y; };
  auto hi_of = [](int x, int y) { return x < y ? y : x; };
  const int A16 = level_row(16, lo_of(A0, A32), hi_of(A0, A32));
  const int A8 = level_row(8, lo_of(A0, A16), hi_of(A0, A16));
  const int A24 = level_row(24, lo_of(A16, A32), hi_of(A16, A32));
  // (written out four times: the argmins stay in registers, no indexed array)
  auto gap = [&](int q, int Alo, int Ahi) {
    const int p0 = row0 + 8 * q;
    if (p0 + 1 >= n) return;
    float best[7], Fp[7];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 7; ++i) {
      Fp[i] = own[(8 * q + 1 + i) * TC];
      best[i] = Fp[i];
    }
#if defined(EDT_MONO_SKIP) && (EDT_MONO_SKIP & 4)
    mono_gap<CW>(L, p0, lo_of(Alo, Ahi), lo_of(Alo, Ahi), best);  // (cost measurement: one candidate per gap)
#else
    mono_gap<CW>(L, p0, lo_of(Alo, Ahi), hi_of(Alo, Ahi), best);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 7; ++i)
      if (p0 + 1 + i < n) done(8 * q + 1 + i, best[i], mono_bound<CW, BB>(L, 8 * q + 1 + i, Fp[i]));
  };
  gap(0, A0, A8);
  gap(1, A8, A16);
  gap(2, A16, A24);
  gap(3, A24, A32);
}


// Which tiles may take the bracket path (mono_band), as bit patterns of the tile's largest field value v:
// lo_bits < bits(v) <= hi_bits.  A tile whose largest value is at most c_T never looks further than T + 32 rows (the
// anchors' windows end by T, a bracket reaches at most one band beyond an anchor's argmin), so T = (largest d with
// c_d exact in fp32) - 33; values below 2^23 * w2 keep one ulp of every candidate (field + c_(T+32)) at or below w2,
// less than the 2 * w2 by which the order of two candidates moves from row to row.  `from`: tiles with windows of
// up to that many rows stay on the windowed path.  Returns false (hi_bits = 0) where the path never applies.
inline bool mono_limits(float w, int n, int from, uint32_t &lo_bits, uint32_t &hi_bits) {
  lo_bits = hi_bits = 0u;
  if (!(w * w >= 1.17549435e-38f) || !((double)w * (double)w < 1.0e30)) return false;
  const int T = brute_exact_prefix(w, n + 64) - 33;
  if (T < 48) return false;
  const double w2 = (double)(w * w);
  double hi = w2 * (double)T * (double)T;
  const double mag = w2 * 8388608.0 - w2 * (double)(T + 32) * (double)(T + 32);
  if (mag < hi) hi = mag;
  auto bits_below = [](double v) -> uint32_t {
    if (!(v > 0.0)) return 0u;
    float f = v < 3.0e38 ? (float)v : 3.0e38f;
    if ((double)f > v) f = nextafterf(f, 0.0f);
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
  };
  hi_bits = bits_below(hi);
  lo_bits = bits_below(w2 * (double)from * (double)from);
  return hi_bits != 0u;
}
