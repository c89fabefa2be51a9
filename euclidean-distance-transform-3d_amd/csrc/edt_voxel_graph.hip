// edt_voxel_graph.hip -- voxel-connectivity-graph EDT (reference: _edt2dsq_voxel_graph /
// _edt3dsq_voxel_graph, src/edt_voxel_graph.hpp:54-117, :120-214).
//
// Formulation (same as the reference so results are bit-identical): binarise the labels,
// up-sample 2x per axis into a uint8 volume in which a forbidden +x/+y/+z step (graph bit
// 0x01 / 0x04 / 0x10 clear) becomes a background half-voxel, transform that volume at half
// the voxel size with the ordinary pipeline, keep every other sample.  Expand and gather
// are two streaming kernels around edt_hip_edtsq_device.
#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

// One thread per (original x, up-sampled y, up-sampled z): writes the two bytes
// (2x, Y, Z), (2x+1, Y, Z) -- coalesced 128 B per wavefront.
template <typename T>
__global__ void k_vg_expand(const T *__restrict__ labels, const uint8_t *__restrict__ graph,
                            uint8_t *__restrict__ big, int64_t sx, int64_t sy, int64_t sz, int ndim,
                            int bb) {
  const int64_t Y = 2 * sy, Z = (ndim == 3) ? 2 * sz : 1;
  const int64_t total = sx * Y * Z;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += step) {
    const int64_t x = idx % sx;
    const int64_t yy = (idx / sx) % Y;
    const int64_t zz = idx / (sx * Y);
    const int64_t y = yy >> 1, z = zz >> 1;
    const int dy = (int)(yy & 1), dz = (int)(zz & 1);
    const int64_t src = x + sx * (y + sy * z);
    const bool fg = labels[src] > 0;  // `labels[loc] > 0` (src/edt_voxel_graph.hpp:76, :143)
    const uint8_t g = graph[src];
    // dx = 0 cell
    bool v0 = fg;
    if (dy == 1 && dz == 0) v0 = fg && (g & 0x04);
    if (dy == 0 && dz == 1) v0 = fg && (g & 0x10);
    // dx = 1 cell: only the pure +x half-step consults the graph
    bool v1 = fg;
    if (dy == 0 && dz == 0) v1 = fg && (g & 0x01);
    if (bb) {  // black border trims the outermost up-sampled faces (:78-90, :156-187)
      if (x == sx - 1) v1 = false;
      if (dy == 1 && y == sy - 1) { v0 = false; v1 = false; }
      if (dz == 1 && z == sz - 1) { v0 = false; v1 = false; }
    }
    uchar2 o;
    o.x = v0 ? 1 : 0;
    o.y = v1 ? 1 : 0;
    reinterpret_cast<uchar2 *>(big)[x + sx * (yy + Y * zz)] = o;
  }
}

__global__ void k_vg_gather(const float *__restrict__ big, float *__restrict__ out, int64_t sx,
                            int64_t sy, int64_t sz, int ndim) {
  const int64_t X = 2 * sx, Y = 2 * sy;
  const int64_t total = sx * sy * sz;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += step) {
    const int64_t x = idx % sx;
    const int64_t y = (idx / sx) % sy;
    const int64_t z = idx / (sx * sy);
    const int64_t zz = (ndim == 3) ? 2 * z : 0;
    out[idx] = big[2 * x + X * (2 * y + Y * zz)];
  }
}

int launch_vg_expand(int dtype, const void *labels, const uint8_t *graph, uint8_t *big, int64_t sx,
                     int64_t sy, int64_t sz, int ndim, int bb, hipStream_t stream) {
  const int threads = 256;
  const int64_t total = sx * 2 * sy * (ndim == 3 ? 2 * sz : 1);
  int64_t blocks = ceil_div(total, threads);
  if (blocks > 16384) blocks = 16384;
#define LAUNCH_VG(T)                                                                            \
  hipLaunchKernelGGL(k_vg_expand<T>, dim3((unsigned)blocks), dim3(threads), 0, stream,          \
                     (const T *)labels, graph, big, sx, sy, sz, ndim, bb)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: LAUNCH_VG(uint8_t); break;
    case EDT_U16: LAUNCH_VG(uint16_t); break;
    case EDT_U32: LAUNCH_VG(uint32_t); break;
    case EDT_U64: LAUNCH_VG(uint64_t); break;
    case EDT_F32: LAUNCH_VG(float); break;
    case EDT_F64: LAUNCH_VG(double); break;
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef LAUNCH_VG
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_vg_gather(const float *big, float *out, int64_t sx, int64_t sy, int64_t sz, int ndim,
                     hipStream_t stream) {
  const int threads = 256;
  int64_t blocks = ceil_div(sx * sy * sz, threads);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_vg_gather, dim3((unsigned)blocks), dim3(threads), 0, stream, big, out, sx, sy,
                     sz, ndim);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace edt_amd
