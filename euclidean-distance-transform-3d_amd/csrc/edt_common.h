// edt_common.h -- shared host/device helpers for the MI355X EDT library (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <string>

#include "edt_hip.h"

// All envelope arithmetic must be evaluated exactly as written: fp64 multiply and add as
// separate IEEE operations (the reference CPU build has no FMA).  The build also passes
// -ffp-contract=off; the pragma makes the intent local and robust.
#pragma clang fp contract(off)

namespace edt_amd {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBandRows = 32;    // rows of the scan axis covered by one bit-word

// ---- error plumbing ---------------------------------------------------------------
void set_error(const std::string &msg);
// Diagnostics mode (edt_hip_set_debug_mode / EDT_HIP_DEBUG_MODE).  THREAD-LOCAL: a call sees the mode of the thread
// that makes it.  The default build honours only the bits that select between RESULT-PRESERVING forms of the same
// computation (kDiagFormBits: used by the test tiers to drive every kernel family and both per-tile forms over the
// whole parity suite); the bits that switch phases off and produce wrong results (1, 2, 4, 8, 0x200, 0x40000,
// 0x80000: measuring what a phase costs) exist only in a library built with -DEDT_DIAG and are compiled out of the
// kernels otherwise.
//   16 / 0x10000  no all-flat shortcut / no self-owned rows      32 / 64   tiled row pass / tiled column pass
//   256           plain group order in pass X                    0x800     plain tile order in the column pass
//   0x1000        name every pass on stderr                      0x2000    no tile takes the windowed path
//   0x4000        every tile takes the windowed path             0x8000    fp64 candidates on the windowed path
//   0x20000       voxel graph: up-sampled formulation            0x100000  fp32 form of pass X (no 16-bit indices)
//   0x200000      voxel graph: separate gather pass              0x400000, 0x800000  (round 3's bracket-path experiment: no effect
//                 now, experiments/colwave_r03)                  0x1000000 short axes (<= 32 rows) stay on the wave kernel
//   0x2000000     inexact voxel sizes: fp64 candidates even where fp32 fma candidates are exact
//   0x4000000     rows of 1025..2048 voxels: the workgroup-phased kernel of pass X, not the two-wave form
//   0x8000000     no 16-bit integer column kernel (edt_colq16.hip): every tile on the fp32 kernels
//   0x10000000    integer column kernels: fp32 values between passes Y and Z, not the 16-bit plane
//   0x20000000    integer column kernels: no wide form (tiles beyond 16 bits go to the fp32 kernel, as in round 4); with it
//                 the fp32 launch over the hand-over list is never skipped
//   0x40000000    integer column kernels: a tile beyond 16 bits always as two wide passes over all its columns (no column subset)
//   0x80          integer column kernels: no short cuts for whole tiles -- a tile of nothing but +inf goes through the wide form like any
//                 other, a tile without structure along the scan axis through scans, break bits and blocks
//   0x400         no short cuts from "both column passes provably on the integer kernel": the foreground planes are written and
//                 transposed although nobody reads them; the signed transform's sign is a pass of its own, not the last pass's epilogue
constexpr int kDiagFormBits = 16 | 32 | 64 | 0x80 | 256 | 0x400 | 0x800 | 0x1000 | 0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x20000 |
                              0x100000 | 0x200000 | 0x400000 | 0x800000 | 0x1000000 | 0x2000000 | 0x4000000 | 0x8000000 | 0x10000000 |
                              0x20000000 | 0x40000000;
#ifdef EDT_DIAG
#define EDT_DIAG_BITS(dbg, bits) ((dbg) & (bits))
#else
#define EDT_DIAG_BITS(dbg, bits) 0
#endif
int debug_mode();                 // the calling thread's mode (masked with kDiagFormBits unless EDT_DIAG)
void set_thread_debug_mode(int);  // (worker threads of one call inherit the caller's mode)

#define EDT_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::edt_amd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
      return EDT_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// Kernels that need more than 64 KiB of dynamic LDS must be given the attribute once per kernel AND per
// device (a host process may drive several GPUs from several threads): `done` is the call site's static
// mask, one bit per device ordinal.  Evaluates to hipSuccess or the failing call's error.
#define EDT_LDS_ATTR_ONCE(done, ...)                                                            \
  [&]() -> hipError_t {                                                                         \
    int dev_ = 0;                                                                               \
    hipError_t e_ = hipGetDevice(&dev_);                                                        \
    if (e_ != hipSuccess) return e_;                                                            \
    const uint64_t bit_ = 1ull << (dev_ & 63);                                                  \
    if ((done).load(std::memory_order_acquire) & bit_) return hipSuccess;                       \
    for (const void *f_ : {__VA_ARGS__}) {                                                      \
      e_ = hipFuncSetAttribute(f_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      if (e_ != hipSuccess) return e_;                                                          \
    }                                                                                           \
    (done).fetch_or(bit_, std::memory_order_release);                                           \
    return hipSuccess;                                                                          \
  }()

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline int dtype_size(int dtype) {
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: return 1;
    case EDT_U16: return 2;
    case EDT_U32: case EDT_F32: return 4;
    case EDT_U64: case EDT_F64: return 8;
    default: return 0;
  }
}

// Geometry of one separable pass over a volume whose x axis is contiguous.
//   column c = (x, o):  first voxel at  x + o * outer_stride,  rows `stride` apart.
//   Y pass: n = sy, stride = sx,    outer = z (nouter = sz, outer_stride = sx*sy)
//   Z pass: n = sz, stride = sx*sy, outer = y (nouter = sy, outer_stride = sx)
struct AxisGeom {
  int64_t sx;            // contiguous extent (columns per outer index)
  int64_t n;             // length of the scan axis
  int64_t stride;        // element stride between consecutive rows of a column
  int64_t nouter;        // number of outer indices
  int64_t outer_stride;  // element stride between outer indices
  int64_t nbands;        // ceil(n / 32): bit-words per column
  // a lower bound of the NON-ZERO values the pass reads, where the caller knows one (pass Y: fl32(wx^2); pass Z: the
  // smaller of that and fl32(wy^2)); 0 = unknown.  Lets the windowed path use fp32 fma candidates for voxel sizes whose
  // c_d are not exact in fp32 (edt_colwave_lane.h: brute_f32e_prefix).
  float fmin = 0.0f;
};

// Destination map of a column pass that scatters its rows into per-destination "slab records"
// (the Z-sharded path, edt_hip.h: edt_hip_shard_xy_records_device): one entry per 32-row band of
// the scan axis.  Every band lies inside one destination block, so a lane (= one band of one
// column) needs a single look-up.
struct BandScatter {
  static constexpr int kBands = 64;  // axes of up to 2048 rows
  float *rows[kBands];      // where row 32*band of outer index 0, column 0 goes
  uint32_t *bits[kBands];   // where the nz word of that band, outer index 0, column 0 goes
  int64_t ostride[kBands];  // 4-byte elements between consecutive outer indices in that destination
  int64_t plane[kBands];    // words between the nz and the zs plane of one outer index
};

// epilogue of the last pass (fused tofinite/toinfinite/sqrt: src/edt.hpp:39-53, :599-601)
// kEpiStream: the pass's results are the CALL's results -- nothing of this call reads them again: the integer column kernel
// writes them with non-temporal stores (round 5: the 512 MiB a 512^3 pass Z leaves in the caches otherwise drain under the next
// call's pass X -- 0.5896 -> 0.573 ms per cfg2 step, 0.5725 -> 0.5583 with two volumes taken in turn)
// kEpiSign: the signed transform (EDT_FLAG_SIGNED) -- the integer column kernel of the call's last pass negates the results of the
// voxels whose foreground bit (Q16Args::signbits: the true label != 0 plane of that axis) is clear, instead of a pass of its own
enum : int { kEpiToInf = 1, kEpiSqrt = 2, kEpiStream = 4, kEpiSign = 8 };

}  // namespace edt_amd
