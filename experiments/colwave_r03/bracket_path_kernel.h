// experiments/colwave_r03/bracket_path_kernel.h -- what csrc/edt_colwave_kernel.h carried for the bracket path (EDT_MONO) at the
// end of round 3: mono_tile, the BruteArgs fields, the per-tile choice, the launcher's limits.  NOT compiled.  See README.md.

// The bracket path of one lane (edt_colwave_lane.h: mono_anchor / mono_band) -- EXPERIMENT, compiled into the kernel
// only with -DEDT_MONO (make VARIANT=mono EXTRA=-DEDT_MONO): bit-exact (GPU parity suite + fuzz under debug bit 0x400000,
// host emulation in the CPU tier) but not faster than the better of the two shipped forms at any cell size
// (profiles/r03_mono_v1_*.txt), so the shipped kernels do not carry its code.  The anchors' argmins cross bands
// through one plane of LDS words (`anchors`: the break-scan plane of the windowed path, unused here) and one
// workgroup barrier -- every thread of the workgroup takes this path together (the choice is per tile).
#ifdef EDT_MONO
template <int CW, bool BB>
__device__ __forceinline__ void mono_tile(float *tile, const uint32_t *rsp, const uint32_t *lohi, uint32_t *anchors,
                                          int n, int NB, int cols_left, int band, int col, float w, int epi,
                                          float *dst0, int64_t dstride) {
  using namespace edt_lane;
  MonoLane ML;
  ML.tile = tile;
  ML.col = col;
  ML.band = band;
  ML.row0 = band * 32;
  ML.n = n;
  ML.rsw = rsp[addr_word<CW>(col, band)];
  const uint32_t lh = lohi[addr_word<CW>(col, band)];
  ML.lo_in = (int)(lh & 0xFFFFu) - 1;
  ML.hi_out = (int)(lh >> 16) - 1;
  ML.w2f = w * w;
  ML.live = col < cols_left && band < NB;
  if (!ML.live) ML.rsw = 0;
  const float Fa = tile[addr_tile<CW>(col, ML.row0)];
  const float Ba = mono_bound<CW, BB>(ML, 0, Fa);
  float best0;
  int A0;
  mono_anchor<CW>(ML, Ba, Fa, best0, A0);
  anchors[addr_word<CW>(col, band)] = (uint32_t)A0;
  __syncthreads();
  const int A32 = (ML.row0 + 32 < n) ? (int)anchors[addr_word<CW>(col, band + 1)] : n - 1;
  auto *gdst = (__attribute__((address_space(1))) float *)dst0;
  const bool colok = col < cols_left;
  auto store = [&](int row, float v) {
    if (row < n && colok) gdst[(int64_t)row * dstride] = v;
  };
  mono_band<CW, BB>(ML, best0, Ba, A0, A32, epi & 3, store);
}
#endif  // EDT_MONO

// ----------------------------------------------------------------
  // the bracket path (edt_colwave_lane.h: mono_band): tiles whose largest field value v satisfies
  // mono_lo_bits < bits(v) <= mono_hi_bits (mono_hi_bits = 0: never; mono_force: every tile up to mono_hi_bits)
  uint32_t mono_lo_bits, mono_hi_bits;
  int mono_force;

// ----------------------------------------------------------------
#ifdef EDT_MONO
        if (mono) {
          mono_tile<CW, BB>(tile, rsp, lohi, bscan, n, NB, cols_left, band2, col2, w, epi, dst0, dstep);
          return;
        }
#endif

// ----------------------------------------------------------------
  // The bracket path (edt_colwave_lane.h: mono_limits has the conditions).  (debug bits: 0x800000 never, 0x400000
  // every tile the exactness conditions allow, whatever its windows.)
  ba.mono_lo_bits = ba.mono_hi_bits = 0u;
  ba.mono_force = 0;

// ----------------------------------------------------------------
#ifdef EDT_MONO
  if (!(debug_mode() & 0x800000) && ba.stride == 1 && ba.compact == nullptr &&
      edt_lane::mono_limits(w, (int)g.n, mono_from(), ba.mono_lo_bits, ba.mono_hi_bits)) {
    ba.mono_force = (debug_mode() & 0x400000) ? 1 : 0;
    if (ba.mono_hi_bits <= ba.mono_lo_bits && !ba.mono_force) ba.mono_hi_bits = 0u;
  }
#endif

// ---- csrc/edt_colwave.hip ----
// Where the bracket path takes over from the windowed path (edt_colwave_lane.h: mono_band): tiles that would need
// windows of more than this many rows.  EDT_HIP_MONO_FROM overrides the default (experiments).
int mono_from() {
  static const int v = [] {
    const char *e = getenv("EDT_HIP_MONO_FROM");
    const int t = e ? atoi(e) : 56;
    return t < 0 ? 0 : (t > 4096 ? 4096 : t);
  }();
  return v;
}

