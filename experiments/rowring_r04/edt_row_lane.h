// edt_row_lane.h -- what the two forms of the register-resident pass X (edt_rowwave.hip: rows pipelined through VGPRs,
// edt_rowring.hip: rows staged in an LDS ring by LDS-DMA) share: the bit-word accumulator, buffer addressing, the
// XCD-aware order of the row groups.
#pragma once

#include "edt_common.h"

namespace edt_amd {
namespace rowlane {

constexpr int kRowWaves = 4;  // waves per workgroup of edt_rowwave.hip (they only share the T table)

// w = 2*w + bit(lane) of `mask`
__device__ __forceinline__ void shift_in(uint32_t &w, unsigned long long mask) {
  asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(w) : "s"(mask) : "vcc");
}

__device__ __forceinline__ int as_int(float v) { return __float_as_int(v); }

// Buffer addressing: a wave-uniform descriptor (SGPRs) + a wave-uniform byte offset (the row) +
// a per-lane 32-bit byte offset (the voxel).  Unlike flat 64-bit per-lane pointers this keeps the
// 24 loads of a row down to 16 offset registers and no address arithmetic at all.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <typename T>
__device__ __forceinline__ T buf_load(rsrc_t r, uint32_t voff, uint32_t soff) {
  if constexpr (sizeof(T) == 1) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0));
  else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
  else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// XCD-aware schedule (see the kernels): worth it when every XCD gets at least one y-band and z is
// long enough to have neighbours in flight; returns 1 and rounds the grid to 8 workgroup columns.
inline int row_xcd_schedule(int64_t nby, int64_t sz, int64_t *blocks, int groups_per_block = kRowWaves,
                            int64_t max_per_xcd = 256) {
  if (nby < 8 || sz < 2 || debug_mode() & 256) return 0;
  const int64_t per_xcd = ceil_div(nby, 8) * sz;  // groups of the busiest XCD
  int64_t bx = ceil_div(per_xcd, groups_per_block);
  if (bx > max_per_xcd) bx = max_per_xcd;
  *blocks = bx * 8;
  return 1;
}

}  // namespace rowlane
}  // namespace edt_amd
