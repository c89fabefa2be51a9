// experiments/colwave_r03/contiguous_blocks_kernel.h -- what csrc/edt_colwave_kernel.h carried for EDT_CONTIG at the end of
// round 3: brute_tile_contig and its call site.  NOT compiled.  See README.md.

#ifdef EDT_CONTIG
// Experiment: the 64 blocks a wave works on at a time are CONTIGUOUS -- lane = column x block, 64 / NBLK
// columns x the NBLK blocks of ONE band -- and the wave's (band, column) pairs are walked in NBLK groups.  `dst_of(col,
// band)`: row 0 of the column the lane writes (see the caller).
template <int CW, bool BB, bool X32, class DstOf>
__device__ EDT_BRUTE_INLINE void brute_tile_contig(float *tile, const uint32_t *alive, const uint32_t *rsp,
                                                   const uint32_t *lohi, const uint32_t *bscan, int n, int NB,
                                                   int cols_left, int wave, int lane, float w, int epi,
                                                   DstOf &&dst_of, int64_t dstride) {
  using namespace edt_lane;
  constexpr int TC = TileGeom<CW>::kCols;
  const bool s2 = (epi & 0x100) != 0;          // blocks of 16 rows (even rows evaluated): two per band
  const int nblk = s2 ? 2 : 4, per = 64 / nblk;  // blocks per band, (band, column) pairs per group
  const int k0 = (lane / per) * (32 / nblk);
  const bool compact = (epi & 0x400) != 0;
#pragma unroll 1
  for (int g = 0; g < nblk; ++g) {
    const int q = g * per + lane % per;          // this lane's (band, column) pair among the wave's 64
    const int band = wave * (64 / TC) + q / TC, col = q % TC;
    BruteLane BL;
    BL.tile = tile;
    BL.col = col;
    BL.band = band;
    BL.row0 = band * 32;
    BL.n = n;
    BL.rsw = rsp[addr_word<CW>(col, band)];
    BL.brk = alive[addr_word<CW>(col, band)];
    const uint32_t bs = bscan[addr_word<CW>(col, band)];
    BL.blo_in = (int)(bs & 0xFFFFu) - 1;
    BL.bhi_out = (int)(bs >> 16);
    const uint32_t lh = lohi[addr_word<CW>(col, band)];
    BL.lo_in = (int)(lh & 0xFFFFu) - 1;
    BL.hi_out = (int)(lh >> 16) - 1;
    BL.w2 = (double)(w * w);
    BL.w2f = w * w;
    BL.live = col < cols_left && band < NB;
    const bool colok = col < cols_left;
    auto *gdst = (__attribute__((address_space(1))) float *)dst_of(col, band);
    auto store = [&](int row, float v) {
      if (row < n && colok) gdst[(int64_t)(compact ? row >> 1 : row) * dstride] = v;
    };
    if (s2) brute_block<CW, BB, X32, 2>(BL, k0, epi & 0xA03, store);
    else brute_block<CW, BB, X32, 1>(BL, k0, epi & 0xA03, store);
  }
}
#endif

// ----------------------------------------------------------------
#ifdef EDT_CONTIG
        {
          // (the same destinations as dst0 above, as a function of the (column, band) pair a lane works on)
          auto dst_of = [&](int c, int b) -> float * {
            if constexpr (SC) {
              const int bb2 = b < BandScatter::kBands ? b : 0;
              return scatter->rows[bb2] + o * scatter->ostride[bb2] + x0 + c - (int64_t)b * 32 * st;
            } else {
              if (ba.compact != nullptr) return ba.compact + x0 + c + o * ba.c_outer;
              return Ftile + c;
            }
          };
          if (ba.x32) brute_tile_contig<CW, BB, true>(tile, alive, rsp, lohi, bscan, n, NB, cols_left, wave, lane, w, epi_s, dst_of, dstep);
          else brute_tile_contig<CW, BB, false>(tile, alive, rsp, lohi, bscan, n, NB, cols_left, wave, lane, w, epi_s, dst_of, dstep);
          return;
        }
#endif
