// edt_colwave.hip -- dispatcher of the wave-autonomous column pass: picks the wave shape (CW columns x
// 64/CW bands per wave) for the length of the scan axis.  The kernels live in edt_colwave_kernel.h and are
// instantiated in edt_colwave_cw*.hip, one translation unit per wave shape.
#include "edt_common.h"
#include "edt_kernels.h"

#include <cstdlib>

namespace edt_amd {

template <int CW>
int launch_wave_c(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w, int bb, int epi,
                  const XFuse *xf, hipStream_t stream, const BandScatter *scatter, bool sc_al, const ColumnOut &out_stride,
                  const TileList &list);
extern template int launch_wave_c<32>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);
extern template int launch_wave_c<16>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);
extern template int launch_wave_c<8>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);
extern template int launch_wave_c<4>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);
extern template int launch_wave_c<2>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);
extern template int launch_wave_c<1>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int, const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &, const TileList &);

// Largest window of the windowed path (edt_colwave_lane.h: brute_band): a tile takes it when no row can be
// improved by a row further than this away.  EDT_HIP_WINDOW_LIMIT overrides the default (experiments).
// Round 3: since the far part of the window (d > 32) reads its rows through one address per eight steps, the
// windowed path beats the hull path at EVERY cell size of the sweep (profiles/r03_window_sweep_far_addressing.txt:
// 256-voxel cells 2.69 ms per 512^3 step against 2.93 ms on hulls), so the default limit is the longest axis the
// wave kernels take; the hull path keeps the tiles that hold FLT_MAX (rows without any boundary, black_border off),
// the all-flat tiles (its shortcut is cheaper) and voxel sizes whose c_d are not exact in fp32 far enough.
int window_limit() {
  static const int v = [] {
    const char *e = getenv("EDT_HIP_WINDOW_LIMIT");
    const int t = e ? atoi(e) : 1024;
    return t < 0 ? 0 : (t > 1024 ? 1024 : t);
  }();
  return v;
}

bool column_pass_wave_supported(const AxisGeom &g) {
  // rows in VGPRs: one band per lane, at most 64 bands per column (n <= 2048)
  return g.nbands >= 1 && g.nbands <= 64;
}

static int launch_wave_any(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w,
                           int bb, int epi, const XFuse *xf, hipStream_t stream,
                           const BandScatter *sc = nullptr, bool sc_al = false, const ColumnOut &out_stride = ColumnOut(),
                           const TileList &list = TileList()) {
  const int64_t NB = g.nbands;
  if (NB <= 2) return launch_wave_c<32>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  if (NB <= 4) return launch_wave_c<16>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  if (NB <= 8) return launch_wave_c<8>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  if (NB <= 16) return launch_wave_c<4>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  if (NB <= 32) return launch_wave_c<2>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  if (NB <= 64) return launch_wave_c<1>(F, nz, rs, g, w, bb, epi, xf, stream, sc, sc_al, out_stride, list);
  set_error("axis too long for the wave column pass");
  return EDT_ERR_UNSUPPORTED;
}

int launch_column_pass_wave(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                            float w, int bb, int epi, hipStream_t stream, const BandScatter *scatter, const ColumnOut &out_stride,
                            const TileList &list) {
  // (the caller of the scattering variant guarantees 16-byte aligned destinations when sx % 4 == 0)
  return launch_wave_any(F, nz, rs, g, w, bb, epi, nullptr, stream, scatter, scatter != nullptr, out_stride, list);
}

// First column pass reading pass 1 as 16-bit distance indices (edt_rowwave.hip, C16): F is only written.
int launch_column_pass_wave_codes(float *F, const uint16_t *codes, const uint32_t *nz, const uint32_t *rs,
                                  const AxisGeom &g, float w, int bb, int epi, float wx, int to_finite,
                                  hipStream_t stream, const BandScatter *scatter, const TileList &list, const ColumnOut &out_stride,
                                  int64_t codes_outer) {
  XFuse xf;
  xf.codes = codes;
  xf.c_outer = codes_outer > 0 ? codes_outer : g.outer_stride;
  xf.w = wx;
  xf.flim = to_finite ? 0x7f7fffff : 0x7f800000;
  return launch_wave_any(F, nz, rs, g, w, bb, epi, &xf, stream, scatter, scatter != nullptr, out_stride, list);
}

}  // namespace edt_amd
