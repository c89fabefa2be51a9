// edt_rowring.hip -- pass 1 (x axis), index form, rows staged in an LDS ring by LDS-DMA.
//
// The register-pipelined kernel of edt_rowwave.hip is bound by memory LATENCY, not by bandwidth or instruction issue:
// a wave holds one row in flight (its next row's loads, in VGPRs), a CU 16 of them (four waves per SIMD at ~120 VGPRs),
// and a row comes back ~2 us after it was asked for -- 262 144 rows of 512 uint32 labels take 0.205 ms whether the
// labels are 4 bytes wide or 1 (measured: 0.171 ms on uint8).  More rows in flight need more registers, which costs
// the waves that hold them.  Here the rows in flight do not live in registers at all: a wave asks for row q + D while
// it works on row q, `buffer_load ... lds` writes the labels (and the row of the slice below, for the z bits) straight
// into slot (q + D) % D of the wave's ring in LDS, and the wave reads a row out of its slot with ds_read when its turn
// comes.  What bounds the rows in flight is then the CU's 160 KB of LDS: ~36 rows of 512 uint32 labels (+ as many of
// the slice below) instead of 16, with fewer registers per wave (no prefetch registers, no left-neighbour loads: the
// left neighbour is the same LDS row one element down).
//
// One wave per workgroup (nothing is shared: the index form needs no T table), walking its row groups -- 32
// consecutive rows of one z-slice, as in edt_rowwave.hip, same XCD-aware order -- as ONE sequence of rows: the ring
// runs across group boundaries, the row above a group's first row (for the y bits) is fetched just before that row.
//
// Counters.  LDS-DMA and the result stores retire through vmcnt in issue order; every iteration issues the same
// number of operations (rows past the end of a short band or past the wave's last group are fetched and stored again,
// same values), so "row q has landed" is `s_waitcnt vmcnt(N)` with a compile-time N = everything issued after row q's
// DMA.  The compiler cannot see that a ds_read depends on an LDS-DMA three iterations back (it would wait for
// vmcnt(0): every row in flight), so the reads are inline assembly with their own lgkmcnt wait.
//
// Everything about the distances themselves -- the closed form over ballot masks, the scalar carries, the bit words
// of the column passes -- is edt_rowwave.hip's (src/edt.hpp:83-118 is what both replace).
#include "edt_common.h"
#include "edt_kernels.h"
#include "edt_row_lane.h"

#pragma clang fp contract(off)

namespace edt_amd {

using namespace rowlane;

namespace {

using lds_ptr = __attribute__((address_space(3))) void *;

// bytes per lane and row (PER) as pieces of 16 / 4 / 2 / 1 bytes per lane: piece j covers row bytes [o_j, o_j + 64 s_j)
constexpr int dma_pieces(int per) { return per / 16 + (per % 16) / 4 + (per % 4) / 2 + per % 2; }

template <int PER, int OFF = 0>
__device__ __forceinline__ void dma_row(rsrc_t rs, unsigned char *lds, uint32_t soff, uint32_t lane) {
  if constexpr (PER >= 16) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + OFF), 16, lane * 16u, soff + OFF, 0, 0);
    dma_row<PER - 16, OFF + 64 * 16>(rs, lds, soff, lane);
  } else if constexpr (PER >= 4) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + OFF), 4, lane * 4u, soff + OFF, 0, 0);
    dma_row<PER - 4, OFF + 64 * 4>(rs, lds, soff, lane);
  } else if constexpr (PER >= 2) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + OFF), 2, lane * 2u, soff + OFF, 0, 0);
    dma_row<PER - 2, OFF + 64 * 2>(rs, lds, soff, lane);
  } else if constexpr (PER >= 1) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + OFF), 1, lane, soff + OFF, 0, 0);
  }
}

// one element of SZ bytes out of LDS into a register of its own (no wait: lds_landed; no conversion before lds_tie: the
// register is not valid yet, nothing may be derived from it)
template <int SZ> struct LdsWord { using type = uint32_t; };
template <> struct LdsWord<8> { using type = uint64_t; };
template <int SZ, int OFF>
__device__ __forceinline__ void lds_get(typename LdsWord<SZ>::type &v, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset");
  if constexpr (SZ == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  else if constexpr (SZ == 4) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  else if constexpr (SZ == 2) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  else asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
// the reads issued so far have landed; a word is only usable through lds_tie AFTER this (the compiler does not know
// about lgkmcnt: volatile asm statements keep their order, and the tie makes every use depend on that order)
__device__ __forceinline__ void lds_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ T lds_tie(typename LdsWord<(int)sizeof(T)>::type &v) {
  asm volatile("" : "+v"(v));
  if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, v);
  else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, v);
  else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(T, (uint16_t)v);
  else return __builtin_bit_cast(T, (uint8_t)v);
}

// chunk C .. NC-1 of one row: labels, left neighbours, the slice below
template <typename T, int NC, bool HAS_Z, int ROWB, int C = 0>
__device__ __forceinline__ void read_row(typename LdsWord<(int)sizeof(T)>::type (&lab)[NC], typename LdsWord<(int)sizeof(T)>::type (&left)[NC],
                                         typename LdsWord<(int)sizeof(T)>::type (&below)[NC], uint32_t a, uint32_t a_left0) {
  constexpr int SZ = (int)sizeof(T);
  if constexpr (C < NC) {
    lds_get<SZ, C * 64 * SZ>(lab[C], a);
    if constexpr (C == 0) lds_get<SZ, 0>(left[C], a_left0);  // (voxel 0 is its own left neighbour)
    else lds_get<SZ, C * 64 * SZ - SZ>(left[C], a);
    if constexpr (HAS_Z) lds_get<SZ, ROWB + C * 64 * SZ>(below[C], a);
    read_row<T, NC, HAS_Z, ROWB, C + 1>(lab, left, below, a, a_left0);
  }
}
template <typename T, int NC, int C = 0>
__device__ __forceinline__ void read_above(typename LdsWord<(int)sizeof(T)>::type (&ab)[NC], uint32_t a) {
  if constexpr (C < NC) {
    lds_get<(int)sizeof(T), C * 64 * (int)sizeof(T)>(ab[C], a);
    read_above<T, NC, C + 1>(ab, a);
  }
}

}  // namespace

// (FULL rows only: sx == 64 NC; the index form only: the kernel stores the 16-bit index of the voxel's distance)
template <typename T, int NC, bool HAS_Z, int D>
__global__ void __launch_bounds__(64)
k_row_pass_ring(const T *__restrict__ labels, uint16_t *__restrict__ out, uint32_t *__restrict__ nz_y, uint32_t *__restrict__ ys_y,
                uint32_t *__restrict__ zs_y, int sy, int sz, int bb, int nby, int ngroups, int xcd_sched,
                const T *__restrict__ halo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SZ = (int)sizeof(T);
  constexpr int sx = 64 * NC;
  constexpr int ROWB = sx * SZ;                   // bytes of one row
  constexpr int SLOT = (HAS_Z ? 2 : 1) * ROWB;    // a ring slot: the row, then the same row of the slice below
  constexpr int PER = NC * SZ;                    // bytes per lane and row
  constexpr int K = dma_pieces(PER) * (HAS_Z ? 2 : 1);  // DMA operations per row
  constexpr int NS = NC;                                // result stores per row
  // issued after row q's DMA before iteration q waits for it: the rest of iteration q - D (its stores), then D - 1 whole
  // iterations (a group's first row adds the row above -- issued BEFORE that row, so older -- and a group's last row its
  // bit words: more than this is outstanding then, never less)
  constexpr int NWAIT_ = NS + (D - 1) * (K + NS);
  constexpr int NWAIT = NWAIT_ < 63 ? NWAIT_ : 63;
  constexpr int NWARM = (D - 1) * K < NWAIT ? (D - 1) * K : NWAIT;  // the wave's first D rows: the prologue's other fetches
  using word_t = typename LdsWord<SZ>::type;
  unsigned char *ring = smem;                     // [D][SLOT]
  unsigned char *above_slot = smem + D * SLOT;    // [ROWB]
  const uint32_t lane = threadIdx.x;
  const int64_t sxy = (int64_t)sx * sy;
  const unsigned long long le_mask = ~0ull >> (63 - lane);  // bits 0..lane
  const unsigned long long gt_mask = ~le_mask;              // bits lane+1..63
  const int pre0 = bb ? 0 : -(1 << 20);
  const int suf0 = bb ? sx : (1 << 20);

  // the wave's groups (edt_rowwave.hip: every XCD takes the y-bands congruent to its index and walks z in order)
  const bool by_xcd = xcd_sched != 0;
  const int xcd = (int)(blockIdx.x & 7), nyk = by_xcd ? (nby - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int count = by_xcd ? nyk * sz : ngroups;
  if (first >= count) return;
  const int last = first + ((count - 1 - first) / step) * step;  // the wave's last group
  auto locate = [&](int i, int &z, int &yb) {
    z = by_xcd ? i / nyk : i / nby;
    yb = by_xcd ? xcd + 8 * (i - z * nyk) : i - z * nby;
  };

  // ---- the fetch cursor: row fr of group fi goes to slot fs -------------------------------------------------------------
  int fi = first, fr = 0, fs = 0;
  rsrc_t f_lab = make_rsrc(labels), f_bel = make_rsrc(labels);
  int f_nrows = 1;
  auto fetch = [&]() {
    if (fr == 0) {
      int z, yb;
      locate(fi, z, yb);
      const int y0 = yb * 32;
      f_nrows = (sy - y0) < 32 ? (sy - y0) : 32;
      const T *base = labels + ((int64_t)z * sy + y0) * sx;
      f_lab = make_rsrc(base);
      f_bel = make_rsrc((HAS_Z && z > 0) ? base - sxy : (HAS_Z && halo != nullptr) ? halo + (int64_t)y0 * sx : base);
      dma_row<PER>(make_rsrc(y0 > 0 ? base - sx : base), above_slot, 0, lane);  // (row 0 of the volume: its own row, not used)
    }
    const uint32_t soff = (uint32_t)((fr < f_nrows ? fr : f_nrows - 1) * ROWB);
    dma_row<PER>(f_lab, ring + fs * SLOT, soff, lane);
    if constexpr (HAS_Z) dma_row<PER>(f_bel, ring + fs * SLOT + ROWB, soff, lane);
    fs = fs + 1 == D ? 0 : fs + 1;
    if (++fr == 32) {
      fr = 0;
      fi = fi + step <= last ? fi + step : last;  // (past the end: the last group again -- same counts, nobody reads it)
    }
  };
#pragma unroll 1
  for (int d = 0; d < D; ++d) fetch();

  int slot = 0, warm = 0;
  const uint32_t lane_b = lane * (uint32_t)SZ;
  const uint32_t lane_left0 = (lane > 0 ? lane - 1 : 0) * (uint32_t)SZ;
#pragma unroll 1
  for (int i = first; i < count; i += step) {
    int z, yb;
    locate(i, z, yb);
    const int y0 = yb * 32;
    const int nrows = (sy - y0) < 32 ? (sy - y0) : 32;
    const rsrc_t rs_out = make_rsrc(reinterpret_cast<char *>(out) + (size_t)(((int64_t)z * sy + y0) * sx) * 2u);
    T above[NC];
    uint32_t nzw[NC], ysw[NC], zsw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { nzw[c] = 0; ysw[c] = 0; zsw[c] = 0; above[c] = T(0); }
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
      // ---- row r out of its slot ----------------------------------------------------------------------------------------
      // (the wave's first D rows: only the prologue's younger fetches are behind them)
      if (warm < D) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWARM) : "memory"); ++warm; }
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
      word_t labv[NC], leftv[NC], belowv[NC];
      {
        const uint32_t sbase = (uint32_t)(uintptr_t)(lds_ptr)(ring + slot * SLOT);
        read_row<T, NC, HAS_Z, ROWB>(labv, leftv, belowv, sbase + lane_b, sbase + lane_left0);
      }
      if (r == 0 && y0 > 0) {
        word_t abv[NC];
        read_above<T, NC>(abv, (uint32_t)(uintptr_t)(lds_ptr)above_slot + lane_b);
        lds_landed();
#pragma unroll
        for (int c = 0; c < NC; ++c) above[c] = lds_tie<T>(abv[c]);
      } else {
        lds_landed();
      }
      T lab[NC], left[NC], below[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        lab[c] = lds_tie<T>(labv[c]);
        left[c] = lds_tie<T>(leftv[c]);
        if constexpr (HAS_Z) below[c] = lds_tie<T>(belowv[c]);
        else below[c] = lab[c];
      }
      slot = slot + 1 == D ? 0 : slot + 1;
      fetch();  // row q + D into the slot just read
      // ---- compares -> masks (SGPRs), bit words ---------------------------------------------------------------------------
      unsigned long long M[NC];
      unsigned long long any_start = 0;
      uint32_t all_fg = 0;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        M[c] = __ballot(lab[c] != left[c]);
        const unsigned long long fg = __ballot(lab[c] != T(0));
        shift_in(nzw[c], fg);
        shift_in(ysw[c], __ballot(lab[c] != above[c]));
        if (HAS_Z) shift_in(zsw[c], __ballot(lab[c] != below[c]));
        above[c] = lab[c];
        any_start |= M[c];
        all_fg |= (fg == ~0ull ? 1u : 0u) << c;
      }
      // ---- run starts / ends carried across chunks (scalar unit) ------------------------------------------------------------
      int pre[NC], suf[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { pre[c] = pre0; suf[c] = suf0; }
      if (any_start) {
        int lastp = pre0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          pre[c] = lastp;
          if (M[c]) lastp = c * 64 + 63 - __builtin_clzll(M[c]);
        }
        int nxt = suf0;
#pragma unroll
        for (int c = NC - 1; c >= 0; --c) {
          suf[c] = nxt;
          if (M[c]) nxt = c * 64 + __builtin_ctzll(M[c]);
        }
      }
      // ---- distance indices, stored at once ----------------------------------------------------------------------------------
      const uint32_t ooff = (uint32_t)((r < nrows ? r : nrows - 1) * sx) * 2u;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int x = c * 64 + (int)lane;
        int il, ir;
        if (M[c] == 0) {
          il = x - pre[c] + 1;
          ir = suf[c] - x;
        } else {
          const unsigned long long m1 = M[c] & le_mask;
          const unsigned long long m2 = M[c] & gt_mask;
          const int s = m1 ? c * 64 + 63 - __builtin_clzll(m1) : pre[c];
          const int e1 = m2 ? c * 64 + __builtin_ctzll(m2) : suf[c];
          il = x - s + 1;
          ir = e1 - x;
        }
        const int k = il < ir ? il : ir;
        int f = k < 0xFFFF ? k : 0xFFFF;
        if (!((all_fg >> c) & 1u)) f = lab[c] != T(0) ? f : 0;
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)f, rs_out, (uint32_t)x * 2u, ooff, 0);
      }
    }
    // ---- the three bit words of this (z, y-band): 32 rows went in, row 0 ended up in bit 31 --------------------------------
    const uint32_t keep = 0xFFFFFFFFu >> (32 - nrows);
    const int64_t wbase = ((int64_t)z * nby + yb) * sx;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int x = c * 64 + (int)lane;
      const uint32_t ys = (__brev(ysw[c]) & keep) | (y0 == 0 ? 1u : 0u);
      const uint32_t zs = (z == 0 && halo == nullptr) ? keep : (__brev(zsw[c]) & keep);
      nz_y[wbase + x] = __brev(nzw[c]) & keep;
      ys_y[wbase + x] = ys;
      if (HAS_Z) zs_y[wbase + x] = zs;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing may land in LDS after the wave has gone)
}

namespace {

template <typename T, int NC, int D>
int launch_ring_tnd(const void *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sy, int64_t sz,
                    int bb, hipStream_t stream, const void *halo) {
  constexpr int ROWB = 64 * NC * (int)sizeof(T);
  const int64_t nby = ceil_div(sy, kBandRows);
  const int64_t ngroups = nby * sz;
  if (ngroups <= 0) return EDT_OK;
  const size_t lds = (size_t)D * (zs_y != nullptr ? 2 : 1) * ROWB + ROWB;
  int64_t per_cu = (int64_t)(160 * 1024 / lds);
  if (per_cu > 32) per_cu = 32;
  if (per_cu < 1) per_cu = 1;
  int64_t blocks = ngroups;
  const int64_t resident = 256 * per_cu;
  if (blocks > resident) blocks = resident;
  const int xcd_sched = row_xcd_schedule(nby, sz, &blocks, 1, resident / 8);
  if (zs_y != nullptr)
    hipLaunchKernelGGL((k_row_pass_ring<T, NC, true, D>), dim3((unsigned)blocks), dim3(64), lds, stream, (const T *)labels, codes, nz_y,
                       ys_y, zs_y, (int)sy, (int)sz, bb, (int)nby, (int)ngroups, xcd_sched, (const T *)halo);
  else
    hipLaunchKernelGGL((k_row_pass_ring<T, NC, false, D>), dim3((unsigned)blocks), dim3(64), lds, stream, (const T *)labels, codes, nz_y,
                       ys_y, zs_y, (int)sy, (int)sz, bb, (int)nby, (int)ngroups, xcd_sched, (const T *)halo);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int ring_depth_knob() {
  static const int v = [] {
    const char *e = getenv("EDT_ROW_RING");
    return e ? atoi(e) : 0;
  }();
  return v;
}

template <typename T, int NC>
int launch_ring_tn(const void *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sy, int64_t sz, int bb,
                   hipStream_t stream, const void *halo) {
  int d = ring_depth_knob();
  if (d == 0) d = 3;
  // (a workgroup's LDS stays within the 64 KB every kernel may ask for without an attribute)
  constexpr int ROWB = 64 * NC * (int)sizeof(T);
  while (d > 2 && (size_t)d * 2 * ROWB + ROWB > 64 * 1024) --d;
  if (d >= 4) return launch_ring_tnd<T, NC, 4>(labels, codes, nz_y, ys_y, zs_y, sy, sz, bb, stream, halo);
  if (d == 3) return launch_ring_tnd<T, NC, 3>(labels, codes, nz_y, ys_y, zs_y, sy, sz, bb, stream, halo);
  return launch_ring_tnd<T, NC, 2>(labels, codes, nz_y, ys_y, zs_y, sy, sz, bb, stream, halo);
}

template <typename T>
int launch_ring_t(const void *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sx, int64_t sy,
                  int64_t sz, int bb, hipStream_t stream, const void *halo) {
  switch (sx) {
    case 512: return launch_ring_tn<T, 8>(labels, codes, nz_y, ys_y, zs_y, sy, sz, bb, stream, halo);
    case 1024: return launch_ring_tn<T, 16>(labels, codes, nz_y, ys_y, zs_y, sy, sz, bb, stream, halo);
    default: set_error("internal: row ring kernel called for an unsupported row length"); return EDT_ERR_BAD_ARG;
  }
}

}  // namespace

// rows of exactly 512 / 1024 voxels, labels (and the halo slice) 16-byte aligned, index form; debug bit 0x400000: the
// register-pipelined kernel for these too
bool row_pass_ring_supported(int dtype, const void *labels, const void *halo, int64_t sx, int64_t sy, int64_t sz) {
  if (debug_mode() & 0x400000) return false;
  if (!(dtype == EDT_U32 || dtype == EDT_U8 || dtype == EDT_BOOL)) return false;
  if (sx != 512 && sx != 1024) return false;
  if ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(halo)) & 15u) return false;
  return sy >= 1 && sz >= 1 && sy * sz < ((int64_t)1 << 30);
}

int launch_row_pass_ring(int dtype, const void *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sx,
                         int64_t sy, int64_t sz, int bb, hipStream_t stream, const void *halo) {
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: return launch_ring_t<uint8_t>(labels, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, halo);
    case EDT_U32: return launch_ring_t<uint32_t>(labels, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, halo);
    default: set_error("internal: row ring kernel called for an unsupported dtype"); return EDT_ERR_BAD_ARG;
  }
}

}  // namespace edt_amd
