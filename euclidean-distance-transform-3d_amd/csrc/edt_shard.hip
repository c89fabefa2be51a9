// edt_shard.hip -- helpers of the Z-sharded (multi-GPU) path.
//
// The volume is cut into contiguous Z-slabs, one per GPU.  X and Y passes are local to a
// slab.  The Z pass needs whole z-columns, so the host layer re-partitions the fp32 partial
// result from Z-slabs to Y-slabs with one all-to-all (RCCL); instead of shipping the labels
// (4-8 B/voxel) a one-byte flag per voxel travels with it:
//     bit0: voxel is foreground (label != 0)
//     bit1: voxel starts a run along z (label differs from the voxel below; for the first
//           slice of a slab the voxel below lives in the previous rank's halo slice)
// After the exchange the flags are packed into the bit-words the column pass consumes.
#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

template <typename T>
__global__ void k_zflags(const T *__restrict__ labels, const T *__restrict__ halo,
                         uint8_t *__restrict__ flags, int64_t sxy, int64_t szl) {
  const int64_t total = sxy * szl;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += step) {
    const T here = labels[i];
    uint8_t f = (here != 0) ? 1 : 0;
    bool starts;
    if (i >= sxy) starts = here != labels[i - sxy];
    else if (halo != nullptr) starts = here != halo[i];
    else starts = true;
    if (starts) f |= 2;
    flags[i] = f;
  }
}

int launch_zflags(int dtype, const void *labels, const void *halo, uint8_t *flags, int64_t sxy,
                  int64_t szl, hipStream_t stream) {
  const int threads = 256;
  int64_t blocks = ceil_div(sxy * szl, threads);
  if (blocks <= 0) return EDT_OK;
  if (blocks > 16384) blocks = 16384;
#define LAUNCH_ZF(T)                                                                          \
  hipLaunchKernelGGL(k_zflags<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,           \
                     (const T *)labels, (const T *)halo, flags, sxy, szl)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_ZF(uint8_t); break;
    case EDT_U16: LAUNCH_ZF(uint16_t); break;
    case EDT_U32: LAUNCH_ZF(uint32_t); break;
    case EDT_U64: LAUNCH_ZF(uint64_t); break;
    case EDT_F32: LAUNCH_ZF(float); break;
    case EDT_F64: LAUNCH_ZF(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_ZF
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// flags (one byte per voxel, volume addressing) -> nz / rs words of the given axis geometry
__global__ void k_bits_from_flags(const uint8_t *__restrict__ flags, uint32_t *__restrict__ nzbits,
                                  uint32_t *__restrict__ rsbits, AxisGeom g) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = g.sx * g.nbands * g.nouter;
  if (idx >= total) return;
  const int64_t x = idx % g.sx;
  const int64_t b = (idx / g.sx) % g.nbands;
  const int64_t o = idx / (g.sx * g.nbands);
  const uint8_t *col = flags + x + o * g.outer_stride;
  uint32_t nz = 0, rs = 0;
  for (int r = 0; r < kBandRows; ++r) {
    const int64_t row = b * kBandRows + r;
    if (row >= g.n) break;
    const uint8_t f = col[row * g.stride];
    nz |= (uint32_t)(f & 1) << r;
    rs |= (uint32_t)((f >> 1) & 1) << r;
  }
  if (b == 0) rs |= 1u;  // row 0 always starts a run
  nzbits[idx] = nz;
  rsbits[idx] = rs;
}

int launch_bits_from_flags(const uint8_t *flags, uint32_t *nz, uint32_t *rs, const AxisGeom &g,
                           hipStream_t stream) {
  const int threads = 256;
  const int64_t total = g.sx * g.nbands * g.nouter;
  if (total <= 0) return EDT_OK;
  hipLaunchKernelGGL(k_bits_from_flags, dim3((unsigned)ceil_div(total, threads)), dim3(threads), 0,
                     stream, flags, nz, rs, g);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

// ---- slab records (the fast Z-sharded path) -------------------------------------------------
// A destination rank h owns the rows [ys_h, ye_h) of every column along y (ys_h a multiple of 32).
// What it needs from one xy-slice is ONE contiguous record:
//     [ (ye_h - ys_h) * sx floats : the slice after the X and Y passes                      ]
//     [ words_h * sx uint32       : foreground bits, 32 rows of y per word (nz plane)       ]
//     [ words_h * sx uint32       : "starts a run along z" bits, same packing (zs plane)    ]
// so a slab of szl slices is szl records per destination, and the exchange is one contiguous
// message per peer that lands where the receiver's Z pass reads it.  The fp32 part is written by
// the Y pass itself (k_column_pass_wave with a BandScatter table); this kernel moves the two bit
// planes of pass 1 into place (2 bits per voxel) and publishes the table.
__global__ void k_pack_record_bits(const uint32_t *__restrict__ nz_y, const uint32_t *__restrict__ zs_y,
                                   BandScatter sc, BandScatter *__restrict__ d_table, int64_t sx,
                                   int64_t nby, int64_t szl) {
  if (blockIdx.x == 0 && threadIdx.x < BandScatter::kBands) {
    const int b = (int)threadIdx.x;
    d_table->rows[b] = sc.rows[b];
    d_table->bits[b] = sc.bits[b];
    d_table->ostride[b] = sc.ostride[b];
    d_table->plane[b] = sc.plane[b];
  }
  const int64_t total = sx * nby * szl;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += step) {
    const int64_t x = i % sx;
    const int64_t b = (i / sx) % nby;
    const int64_t z = i / (sx * nby);
    uint32_t *dst = sc.bits[b] + z * sc.ostride[b] + x;
    dst[0] = nz_y[i];
    dst[sc.plane[b]] = zs_y[i];
  }
}

int launch_pack_record_bits(const uint32_t *nz_y, const uint32_t *zs_y, const BandScatter &sc,
                            BandScatter *d_table, int64_t sx, int64_t nby, int64_t szl,
                            hipStream_t stream) {
  const int threads = 256;
  int64_t blocks = ceil_div(sx * nby * szl, threads);
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_pack_record_bits, dim3((unsigned)blocks), dim3(threads), 0, stream, nz_y, zs_y, sc,
                     d_table, sx, nby, szl);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace edt_amd
