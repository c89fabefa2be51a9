// edt_colwave_cw1.hip -- the column-pass kernels of the wave shape CW = 1 (64 bands per column) in their
// own translation unit / device code object (see edt_colwave_kernel.h).
#include "edt_colwave_kernel.h"

namespace edt_amd {
template int launch_wave_c<1>(float *, const uint32_t *, const uint32_t *, const AxisGeom &, float, int, int,
                               const XFuse *, hipStream_t, const BandScatter *, bool, const ColumnOut &,
                               const TileList &);
}  // namespace edt_amd
