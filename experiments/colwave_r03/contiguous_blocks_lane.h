// experiments/colwave_r03/contiguous_blocks_lane.h -- the EDT_CONTIG form of the windowed path: (1) the branch of
// BruteSteps::run that selects the window pointers per lane, (2) brute_block.  As it stood in csrc/edt_colwave_lane.h at
// the end of round 3.  NOT compiled into the library.  See README.md in this directory.

// ---- (1) inside BruteSteps::run<D>(), in place of the four w[...] = (D <= k0 ? PL0 : PL1)[...] loads ----
#ifdef EDT_CONTIG
      // k0 differs from lane to lane (brute_block): the window leaves the lane's band at a step that does too -- but only
      // at d = k0 + 1 below and d = 33 - NR - k0 above, both congruent to 1 (mod 8): four static steps where the pointer of
      // either side moves to the neighbouring band by a per-lane select (PL0 / PH0 hold the CURRENT pointers)
      if constexpr (D % 8 == 1) {
        // (the neighbouring bands' addresses are rebuilt here rather than kept alive across the block: the kernel has
        // no registers to spare)
        const float *below = L.tile + addr_tile<CW>(L.col, L.row0 - 32) + (k0 + 32 - K) * TC;
        const float *above = L.tile + addr_tile<CW>(L.col, L.row0 + 32) + (k0 + NR - 1 - 32) * TC;
        PL0 = (k0 == D - 1) ? below : PL0;
        PH0 = (k0 == 33 - NR - D) ? above : PH0;
      }
      w[K - D] = PL0[(K - D) * TC];
      w[K + NR - 1 + D] = PH0[D * TC];
      w[K - D - 1] = PL0[(K - D - 1) * TC];
      w[K + NR + D] = PH0[(D + 1) * TC];
#else
      w[K - D] = (D <= k0 ? PL0 : PL1)[(K - D) * TC];
      w[K + NR - 1 + D] = (D <= 32 - NR - k0 ? PH0 : PH1)[D * TC];
      w[K - D - 1] = (D + 1 <= k0 ? PL0 : PL1)[(K - D - 1) * TC];
      w[K + NR + D] = (D + 1 <= 32 - NR - k0 ? PH0 : PH1)[(D + 1) * TC];
#endif

// ---- (2) after brute_band ----
#ifdef EDT_CONTIG
// One block of a lane, the block's position k0 inside the band being the LANE's own (experiment, DESIGN.md 7.1: a wave
// then works on 64 CONTIGUOUS blocks -- 16 columns x the four blocks of one band -- whose windows are alike, instead of
// block k of bands 32 rows apart).  Same arithmetic as brute_band's loop body; what was carried from block to block (the
// distance of the row before the block to its run start) is computed from the run-start word, and the window's band
// crossings are per-lane selects (BruteSteps, EDT_CONTIG).
template <int CW, bool BB, bool X32, int S, class Store>
EDT_LANE void brute_block(const BruteLane &L, int k0, int epi, Store &&store) {
  constexpr int K = kBruteK, B = kBruteB, TC = TileGeom<CW>::kCols, NR = S * B;
  const int row0 = L.row0, n = L.n;
  const uint32_t rsw = L.rsw;
  const float *A0 = L.tile + addr_tile<CW>(L.col, row0);
  const float *Am = L.tile + addr_tile<CW>(L.col, row0 - 32);
  const float *Ap = L.tile + addr_tile<CW>(L.col, row0 + 32);
  const int nb32 = ((n + 31) >> 5) << 5;
  // distance of the row before the block to the row before ITS run (+inf: that run has no border below)
  float dl;
  {
    const uint32_t lowm = k0 > 0 ? rsw & (0xFFFFFFFFu >> (32 - k0)) : 0u;  // run starts at rows < k0 of this band
    const int s = lowm ? row0 + 31 - clz32(lowm) : L.lo_in;                 // first row of the run of row p0 - 1
    dl = (BB || s > 0) ? (float)(row0 + k0 - s) : INFINITY;
  }
  float w[NR + 2 * K];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NR; ++j) w[K + j] = A0[(k0 + j) * TC];
  const float *PL0 = A0 + (k0 - K) * TC, *PL1 = Am + (k0 + 32 - K) * TC;
  const float *PH0 = A0 + (k0 + NR - 1) * TC, *PH1 = Ap + (k0 + NR - 1 - 32) * TC;
  const uint32_t s8 = rsw >> k0;
  float dlv[B];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < NR; ++j) {
    const float first = (BB || j > 0) ? 1.0f : (row0 + k0 > 0 ? 1.0f : INFINITY);
    dl = ((s8 >> j) & 1u) ? first : dl + 1.0f;
    if (j % S == 0) dlv[j / S] = dl;
  }
  float dr;
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
      const float bord = L.w2f * (dm * dm);
      float b = minpos(w[K + j], bord);
      if (f2u(w[K + j]) == 0x7f800000u || !L.live) b = 0.0f;
      best[i] = b;
      if (!X32) best64[i] = (double)b;
      const uint32_t ub = f2u(b);
      bmax = ub > bmax ? ub : bmax;
    }
    if ((s8 >> j) & 1u) dr = 0.0f;
  }
  const float bmaxf = u2f(bmax);
  const double bmax64 = (double)bmaxf;
  bool open = true;
  {
    const int D = brute_flat_reach(L, k0, NR);
    const double cD = L.w2 * (double)((D + 1) * (D + 1));
    if (!EDT_ANY(cD < bmax64)) open = false;
  }
  if (open) {
    float w2f = L.w2f;
    double w2 = L.w2;
    EDT_OPAQUE(w2f);
    EDT_OPAQUE(w2);
    BruteSteps<CW, X32, S> steps{L, w, best, best64, PL0, PL1, PH0, PH1, k0, bmaxf, bmax64, nb32, w2f, w2,
                                  (epi & 0x800) ? 1u : 0u};
    steps.template run<1>();
  }
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
#endif  // EDT_CONTIG