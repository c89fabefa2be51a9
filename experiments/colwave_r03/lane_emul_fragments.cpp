// experiments/colwave_r03/lane_emul_fragments.cpp -- what tests/lane_emul.cpp carried to play the bracket path (mode 6) and the
// EDT_CONTIG block form on the host.  NOT compiled.  See README.md.

static long g_mono_tiles = 0;  // tiles that took the bracket path (tests make sure the path is really exercised)
extern "C" long lane_emul_mono_tiles() { return g_mono_tiles; }

// ----------------------------------------------------------------
  // mode 6: every tile the exactness conditions allow takes the bracket path (edt_colwave_lane.h: mono_band), the
  // others fall through to the hulls -- the kernel under debug bit 0x400000
  if (mode == 6) {
    uint32_t lo_bits, hi_bits;
    float fmaxv = 0.0f;
    for (auto &P : lanes)
      if (P.L.colc < cols_left && P.L.band < NB)
        for (int r = 0; r < 32; ++r) fmaxv = std::max(fmaxv, P.f[r]);
    if (mono_limits(w, n, 0, lo_bits, hi_bits) && f2u(fmaxv) <= hi_bits) {
      ++g_mono_tiles;
      for (int row = -32; row < (NB + 1) * 32; ++row)
        if (row < 0 || row >= n)
          for (int c = 0; c < TC; ++c) tile[addr_tile<CW>(c, row)] = INFINITY;
      auto lane_at = [&](int colc, int band) -> PerLane * {
        for (auto &Q : lanes)
          if (Q.L.colc == colc && Q.L.band == band) return &Q;
        return nullptr;
      };
      std::vector<float> res((size_t)NBP * 32 * TC, 0.0f);
      std::vector<int> anchor((size_t)NBP * TC, 0);
      std::vector<float> best0((size_t)NBP * TC, 0.0f), bound0((size_t)NBP * TC, 0.0f);
      std::vector<MonoLane> ml((size_t)NBP * TC);
      for (int band = 0; band < NBP; ++band)
        for (int col = 0; col < TC; ++col) {
          PerLane *P = lane_at(col, band);
          MonoLane &ML = ml[(size_t)band * TC + col];
          ML.tile = tile; ML.col = col; ML.band = band; ML.row0 = band * 32; ML.n = n;
          ML.rsw = P->L.rsw; ML.lo_in = P->L.lo_in; ML.hi_out = P->L.hi_out;
          ML.w2f = w * w;
          ML.live = col < cols_left && band < NB;
          if (!ML.live) ML.rsw = 0;
          const float Fa = tile[addr_tile<CW>(col, ML.row0)];
          const float Ba = mono_bound<CW, BB>(ML, 0, Fa);
          // (the kernel runs the anchor search wave-wide: a lane keeps going while any lane of its wave does; the
          // extra candidates are valid ones, so a lane-by-lane search may only see fewer -- both are exact)
          mono_anchor<CW>(ML, Ba, Fa, best0[(size_t)band * TC + col], anchor[(size_t)band * TC + col]);
          bound0[(size_t)band * TC + col] = Ba;
        }
      for (int band = 0; band < NBP; ++band)
        for (int col = 0; col < TC; ++col) {
          const MonoLane &ML = ml[(size_t)band * TC + col];
          const int A32 = (ML.row0 + 32 < n) ? anchor[(size_t)(band + 1) * TC + col] : n - 1;
          auto store = [&](int row, float v) { res[(size_t)row * TC + col] = v; };
          mono_band<CW, BB>(ML, best0[(size_t)band * TC + col], bound0[(size_t)band * TC + col],
                            anchor[(size_t)band * TC + col], A32, epi, store);
        }
      for (int row = 0; row < n; ++row)
        for (int c = 0; c < TC && c < cols_left; ++c) F[x0 + (int64_t)row * rstride + c] = res[(size_t)row * TC + c];
      return;
    }
    mode = 0;
  }

// ----------------------------------------------------------------
#ifdef EDT_CONTIG
          // the experiment's form: one call per block, the block's position handed in (lane = column x block)
          for (int k0 = 0; k0 < 32; k0 += 8 * stride) {
            if (stride == 2) {
              if (x32) brute_block<CW, BB, true, 2>(BL, k0, epi, store);
              else brute_block<CW, BB, false, 2>(BL, k0, epi, store);
            } else {
              if (x32) brute_block<CW, BB, true, 1>(BL, k0, epi, store);
              else brute_block<CW, BB, false, 1>(BL, k0, epi, store);
            }
          }
