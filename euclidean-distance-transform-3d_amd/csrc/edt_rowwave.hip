// edt_rowwave.hip -- pass 1 (x axis) for gfx950, register-resident: one wavefront per group of
// 32 consecutive rows of one z-slice for rows of up to 1024 voxels (16 chunks of 64 lanes), two / four wavefronts
// (one per part of the row, k_row_pass_wave<..., H = 2 / 4>) for rows of 1025..2048 / 2049..4096 voxels.
//
// Pass 1 is a label-aware 1-D distance along contiguous rows.  The reference walks each row
// twice with fp32 recurrences (src/edt.hpp:83-118); the result has the closed form
//     d(i) = min( L, R ),   L = T[i-s+1]  (if a boundary exists on the left,  else +inf)
//                           R = T[e-i+1]  (if a boundary exists on the right, else +inf)
//     T[0] = 0, T[k] = fl32(T[k-1] + w)        (the SAME sequential fp32 sums)
// for a voxel i inside the maximal run [s,e] of one non-zero label, and F = fl32(d*d).
//
// The kernel is bound by VALU issue, not by HBM, so everything that can live on the scalar
// unit does: a row is 8 wave-wide compares ("label differs from its left neighbour") whose
// results ARE the 64-bit run-start masks, held in SGPRs; run starts / ends that lie in another
// chunk are carried with scalar find-first-bit instructions; per voxel only the bit scan of
// its own chunk mask, two table look-ups (T lives in LDS), a min and a multiply remain.
//
// The same sweep emits, per voxel, the three bits the column passes need, so that labels are
// read from HBM exactly ONCE by the whole pipeline:
//     nz : label != 0
//     ys : label differs from the voxel at y-1 (run start along y)
//     zs : label differs from the voxel at z-1 (run start along z)
// packed 32 consecutive y per word, layout [z][y/32][x].  A lane builds its three words with
// one add-with-carry per row: w = 2*w + (this row's compare bit), the compare mask being the
// carry-in -- the words come out bit-reversed and are flipped once at the end.
#include "edt_common.h"
#include "edt_kernels.h"

#include <cmath>

#pragma clang fp contract(off)

// cache policy of the result stores (experiment knob; 0 = default, 2 = nt)
#ifndef EDT_ROW_STORE_AUX
#define EDT_ROW_STORE_AUX 0
#endif

namespace edt_amd {

namespace {

constexpr int kRowWaves = 4;  // waves per workgroup (they only share the T table)

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

}  // namespace

// FULL: sx == 64*NC exactly, so no lane ever falls off the end of a row (no clamped offsets, no
// store guards).  The wave-uniform fast paths below (a chunk without run starts, a chunk without
// background, a row without any start) skip whole groups of per-voxel and scalar instructions: they
// brought the kernel from its instruction-issue limit (~1 instruction per SIMD per 4-5 cycles over
// all types, SQ_ACTIVE_INST_ANY ~ 85 %) down to where it now sits on uint32 labels: 1.2 GB at
// 4.8 TB/s, 88 % of a plain copy on the same box (DESIGN.md 4.1 has the two experiments that show
// neither fewer VALU instructions nor 16-byte loads move it any further).
// XCD-aware schedule (see the kernels): worth it when every XCD gets at least one y-band and z is
// long enough to have neighbours in flight; returns 1 and rounds the grid to 8 workgroup columns.
static int row_xcd_schedule(int64_t nby, int64_t sz, int64_t *blocks, int groups_per_block = kRowWaves,
                            int64_t max_per_xcd = 256) {
  if (nby < 8 || sz < 2 || debug_mode() & 256) return 0;
  const int64_t per_xcd = ceil_div(nby, 8) * sz;  // groups of the busiest XCD
  int64_t bx = ceil_div(per_xcd, groups_per_block);
  if (bx > max_per_xcd) bx = max_per_xcd;
  *blocks = bx * 8;
  return 1;
}

// C16: the kernel stores, instead of F, the 16-bit INDEX k = min(i-s+1, e-i+1) of the voxel's distance (0 for
// background, 0xFFFF where neither side has a boundary): half the bytes of the fp32 value, and the first column pass
// rebuilds F = fl32(fl32(k*w)^2) while it fills its tile (edt_colwave_kernel.h, XF).  Only used when k*w is exact for
// every k of the row (row_codes_exact): the sequential sums T[k] of the reference then ARE k*w.
// H = 2 / 4: rows of 1025..2048 / 2049..4096 voxels as H parts of NC chunks each, one wave per part, a workgroup = the
// H waves of one group of rows.  All a part needs from the others is one position per side and row -- the last run start
// to its left (where its first run begins) and the first run start to its right (where its last run ends) -- so every
// wave publishes the last and the first start of its own part in LDS, ONE workgroup barrier per row, and takes the
// nearest ones on either side; all waves walk the same groups and the same rows, so they meet at every barrier.
template <typename T, int NC, bool HAS_Z, bool FULL, bool C16, int H = 1>
__global__ void __launch_bounds__((H >= 2 ? H : kRowWaves) * 64)
k_row_pass_wave(const T *__restrict__ labels, float *__restrict__ out, uint32_t *__restrict__ nz_y,
                uint32_t *__restrict__ ys_y, uint32_t *__restrict__ zs_y, int sx, int sy, int sz, float w,
                int bb, int to_finite, int nby, int ngroups, int xcd_sched, const T *__restrict__ halo, int zero_label,
                int64_t opitch) {
  // (opitch: elements between the slices of `out` -- sx * sy, or the padded pitch of the index buffer: edt_api.hip, code_pitch)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *Ttab = reinterpret_cast<float *>(smem);  // [sx + 3]: T[0..sx+1], then +inf
  int *xchg = reinterpret_cast<int *>(smem) + ((sx + 3 + 3) & ~3);  // H >= 2: [row parity][part][last, first] boundary positions
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  constexpr int GW = H >= 2 ? 1 : kRowWaves;  // groups of rows a workgroup works on at a time
  const int gslot = H >= 2 ? 0 : wave;        // which of them this wave takes
  const int xb = H >= 2 ? wave * (NC * 64) : 0;  // first voxel of this wave's part of the row

  // The reference's sequential fp32 sums of the voxel size (src/edt.hpp:97, :113).
  if (!C16 && threadIdx.x == 0) {
    float acc = 0.0f;
    Ttab[0] = 0.0f;
    for (int k = 1; k <= sx + 1; ++k) {
      acc = acc + w;
      Ttab[k] = acc;
    }
    Ttab[sx + 2] = INFINITY;
  }
  __syncthreads();

  const int idx_inf = sx + 2;
  const int64_t sxy = (int64_t)sx * sy;
  const unsigned long long le_mask = ~0ull >> (63 - lane);  // bits 0..lane
  const unsigned long long gt_mask = ~le_mask;              // bits lane+1..63
  const int flim = to_finite ? 0x7f7fffff : 0x7f800000;     // FLT_MAX / +inf bit patterns
  // start of the row = initial carry: position 0 with a black border, far to the left without
  // (voxel 0 never marks itself -- it is its own left neighbour); end of the row likewise
  const int pre0 = bb ? 0 : -(1 << 20);
  const int suf0 = bb ? sx : (1 << 20);
  const unsigned long long force_fg = zero_label == 1 ? ~0ull : 0ull;  // (see the row loop)
  const uint32_t keep_all = zero_label != 0 ? ~0u : 0u;

  // Work distribution.  The `zs` bits need the labels of slice z-1, which some other wave reads as
  // ITS slice: when both run on the same XCD at about the same time the second read hits in that
  // XCD's L2 instead of crossing the fabric again (observed placement: workgroup b runs on XCD
  // b % 8 -- used for speed only).  So every XCD takes the y-bands congruent to its index and
  // walks z in order: slices z-1 and z of one band are then neighbouring waves of one XCD.
  const bool by_xcd = xcd_sched != 0;
  const int xcd = (int)(blockIdx.x & 7), nyk = by_xcd ? (nby - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) * GW + gslot : (int)blockIdx.x * GW + gslot;
  const int step = by_xcd ? (int)(gridDim.x >> 3) * GW : (int)gridDim.x * GW;
  const int count = by_xcd ? nyk * sz : ngroups;
  int par = 0;  // H >= 2: which set of exchange words the next row uses
  for (int i = first; i < count; i += step) {
    const int z = by_xcd ? i / nyk : i / nby;
    const int yb = by_xcd ? xcd + 8 * (i - z * nyk) : i - z * nby;
    const int y0 = yb * 32;
    const int nrows = (sy - y0) < 32 ? (sy - y0) : 32;
    const T *base = labels + ((int64_t)z * sy + y0) * sx;  // row y0 of this slice
    constexpr uint32_t OB = C16 ? 2u : 4u;  // bytes per stored voxel
    char *obase = reinterpret_cast<char *>(out) + (size_t)((int64_t)z * opitch + (int64_t)y0 * sx) * OB;
    const rsrc_t rs_lab = make_rsrc(base);
    // the slice below: inside the volume, or -- for slice 0 of a Z-sharded slab -- the halo slice
    const rsrc_t rs_bel = make_rsrc((HAS_Z && z > 0) ? base - sxy
                                    : (HAS_Z && halo != nullptr) ? halo + (int64_t)y0 * sx : base);
    const rsrc_t rs_out = make_rsrc(obase);

    uint32_t xs[NC], xl[NC];  // per-lane BYTE offsets inside a row
    T above[NC];
    uint32_t nzw[NC], ysw[NC], zsw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int x = xb + c * 64 + lane;
      // every load is unconditional, clamped to a voxel that exists; lanes past the end of the row
      // read the last voxel as their own AND as their left neighbour, so they never mark a start
      xs[c] = (uint32_t)((FULL || x < sx) ? x : sx - 1) * (uint32_t)sizeof(T);
      xl[c] = (uint32_t)((FULL || x < sx) ? (x > 0 ? x - 1 : 0) : sx - 1) * (uint32_t)sizeof(T);
      nzw[c] = 0; ysw[c] = 0; zsw[c] = 0;
      above[c] = y0 > 0 ? buf_load<T>(make_rsrc(base - sx), xs[c], 0) : T(0);
    }

    // Software pipeline.  On gfx9 loads and stores retire through ONE in-order counter (vmcnt), so
    // a row whose loads are issued after the previous row's stores cannot be consumed before those
    // stores have completed: load latency and store latency add up per row.  Here the loads of
    // row r+1 are issued BEFORE the stores of row r; waiting for them then leaves the 8 younger
    // stores in flight (s_waitcnt vmcnt(8)) and they drain behind the next row's arithmetic.
    T lab[NC], left[NC], below[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      lab[c] = buf_load<T>(rs_lab, xs[c], 0);
      left[c] = buf_load<T>(rs_lab, xl[c], 0);
      below[c] = HAS_Z ? buf_load<T>(rs_bel, xs[c], 0) : lab[c];
    }
    int pend[NC];  // results of the previous row, stored one iteration late (see below)
#pragma unroll
    for (int c = 0; c < NC; ++c) pend[c] = 0;
#pragma unroll 1
    for (int r = 0; r < nrows; ++r) {
      // ---- compares -> masks (SGPRs), bit words --------------------------------------------
      unsigned long long M[NC];
      unsigned long long any_start = 0;  // OR of the start masks of the row
      uint32_t all_fg = 0;               // bit c: chunk c has no background voxel
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        M[c] = __ballot(lab[c] != left[c]);
        // (zero_label: the SIGNED transform -- label 0 is a label like any other, its runs are measured like every run.  1: the
        // foreground plane says "everything" (the fp32 column kernels read it); 2: it keeps the truth -- the call's column passes
        // are integer passes that never read it, and the last of them takes the background's sign from it: edt_api.hip.  As two
        // masks made once per kernel: this loop is bound by its scalar instructions)
        const unsigned long long fg = __ballot(lab[c] != T(0)) | force_fg;
        shift_in(nzw[c], fg);
        shift_in(ysw[c], __ballot(lab[c] != above[c]));
        if (HAS_Z) shift_in(zsw[c], __ballot(lab[c] != below[c]));
        above[c] = lab[c];
        any_start |= M[c];
        all_fg |= (fg == ~0ull ? 1u : 0u) << c;
      }
      all_fg |= keep_all;  // (zero_label: no voxel is zeroed)
      // ---- the previous row's results leave now: issued after this row's loads have been waited
      //      for and a whole distance stage before the next wait, they retire off the critical path
      if (r > 0) {
        const uint32_t poff = (uint32_t)((r - 1) * sx) * OB;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int x = xb + c * 64 + lane;
          if (FULL || x < sx) {
            if (C16) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[c], rs_out, (uint32_t)x * 2u, poff, EDT_ROW_STORE_AUX);
            else __builtin_amdgcn_raw_buffer_store_b32((uint32_t)pend[c], rs_out, (uint32_t)x * 4u, poff, EDT_ROW_STORE_AUX);
          }
        }
      }
      // ---- next row's loads (the last row reloads itself: those hits cost nothing) ---------------
      T cur[NC];  // this row's labels for the background select below
      {
        const int rn = r + 1 < nrows ? r + 1 : r;
        const uint32_t soff = (uint32_t)(rn * sx) * (uint32_t)sizeof(T);  // wave-uniform row offset
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          cur[c] = lab[c];
          lab[c] = buf_load<T>(rs_lab, xs[c], soff);
          left[c] = buf_load<T>(rs_lab, xl[c], soff);
          below[c] = HAS_Z ? buf_load<T>(rs_bel, xs[c], soff) : lab[c];
        }
      }
      // ---- run starts / ends carried across chunks (scalar unit) ----------------------------
      int pre_in = pre0, suf_in = suf0;  // what the row looks like to the left / right of this wave's part
      if constexpr (H >= 2) {
        // this part's boundary positions for the other parts: its LAST run start (where the first run of the part to
        // its right begins) and its FIRST one (where the last run of the part to its left ends); kNone: no start in
        // this part -- the others then look further, and past the row's ends see pre0 / suf0
        constexpr int kNone = INT32_MIN;
        int my_last = kNone, my_first = kNone;
        if (any_start) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (M[c]) my_last = xb + c * 64 + 63 - __builtin_clzll(M[c]);
#pragma unroll
          for (int c = NC - 1; c >= 0; --c)
            if (M[c]) my_first = xb + c * 64 + __builtin_ctzll(M[c]);
        }
        // (two sets of words, alternating from row to row ACROSS groups: the barrier of the next row separates this
        // row's reads from the writes of the row after it)
        int *slot = xchg + 2 * H * par;
        par ^= 1;
        if (lane == 0) { slot[2 * wave] = my_last; slot[2 * wave + 1] = my_first; }
        __syncthreads();
        const int mine_or_theirs = slot[lane & (2 * H - 1)];  // lane i holds word i (i < 2 H)
#pragma unroll
        for (int v = 0; v < H; ++v) {  // nearest part to the left that has a start: ascending v, the last hit stays
          const int last_v = __builtin_amdgcn_readlane(mine_or_theirs, 2 * v);
          if (v < wave && last_v != kNone) pre_in = last_v;
        }
#pragma unroll
        for (int v = H - 1; v >= 0; --v) {  // nearest part to the right: descending v
          const int first_v = __builtin_amdgcn_readlane(mine_or_theirs, 2 * v + 1);
          if (v > wave && first_v != kNone) suf_in = first_v;
        }
      }
      int pre[NC], suf[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { pre[c] = pre_in; suf[c] = suf_in; }
      if (any_start) {
        int last = pre_in;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          pre[c] = last;
          if (M[c]) last = xb + c * 64 + 63 - __builtin_clzll(M[c]);
        }
        int nxt = suf_in;
#pragma unroll
        for (int c = NC - 1; c >= 0; --c) {
          suf[c] = nxt;
          if (M[c]) nxt = xb + c * 64 + __builtin_ctzll(M[c]);
        }
      }
      // ---- distances ---------------------------------------------------------------------------
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int x = xb + c * 64 + lane;
        int il, ir;
        if (M[c] == 0) {
          // no run starts inside this chunk (wave-uniform test): every voxel belongs to the run
          // carried in from the left and out to the right -- no bit scans, no selects
          il = x - pre[c] + 1;
          ir = suf[c] - x;
        } else {
          const unsigned long long m1 = M[c] & le_mask;
          const unsigned long long m2 = M[c] & gt_mask;
          const int s = m1 ? xb + c * 64 + 63 - __builtin_clzll(m1) : pre[c];   // first voxel of the run
          const int e1 = m2 ? xb + c * 64 + __builtin_ctzll(m2) : suf[c];        // one past its last voxel
          il = x - s + 1;
          ir = e1 - x;
        }
        int f;
        if (C16) {
          // T is non-decreasing: min(T[il], T[ir]) = T[min(il, ir)] -- the index is all the next pass needs
          const int k = il < ir ? il : ir;
          // ("no boundary" is an index far beyond the row: 0xFFFF = +inf; edt_colwave_lane.h: code_value)
          f = k < 0xFFFF ? k : 0xFFFF;
        } else {
          il = il < idx_inf ? il : idx_inf;
          ir = ir < idx_inf ? ir : idx_inf;
          const int dL = as_int(Ttab[il]), dR = as_int(Ttab[ir]);
          const float d = __int_as_float(dL < dR ? dL : dR);  // positive floats order like integers
          f = as_int(d * d);
          f = f < flim ? f : flim;                             // tofinite (src/edt.hpp:39-45)
        }
        if (!((all_fg >> c) & 1u)) f = cur[c] != T(0) ? f : 0;  // (wave-uniform test)
        pend[c] = f;
      }
    }
    {
      const uint32_t poff = (uint32_t)((nrows - 1) * sx) * OB;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int x = xb + c * 64 + lane;
        if (FULL || x < sx) {
          if (C16) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pend[c], rs_out, (uint32_t)x * 2u, poff, EDT_ROW_STORE_AUX);
          else __builtin_amdgcn_raw_buffer_store_b32((uint32_t)pend[c], rs_out, (uint32_t)x * 4u, poff, EDT_ROW_STORE_AUX);
        }
      }
    }

    // ---- the three bit words of this (z, y-band) -----------------------------------------------
    const int sh = 32 - nrows;
    const int64_t wbase = ((int64_t)z * nby + yb) * sx;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int x = xb + c * 64 + lane;
      if (FULL || x < sx) {
        // row 0 of the volume starts a run along y, slice 0 starts every run along z
        const uint32_t ys = (__brev(ysw[c]) >> sh) | (y0 == 0 ? 1u : 0u);
        const uint32_t zs = (z == 0 && halo == nullptr) ? (0xFFFFFFFFu >> sh) : (__brev(zsw[c]) >> sh);
        if (nz_y != nullptr) nz_y[wbase + x] = __brev(nzw[c]) >> sh;  // (nullptr: nobody will read a foreground plane, edt_api.hip)
        ys_y[wbase + x] = ys;
        if (HAS_Z) zs_y[wbase + x] = zs;
      }
    }
  }
}

// k * w is exactly representable for every k <= sx + 1 (w = m * 2^e with an odd integer m and m * (sx + 1) < 2^24):
// the reference's sequential sums T[k] = fl32(T[k-1] + w) are then exact too, i.e. T[k] = k * w, and the 16-bit
// index form of pass 1 may be used.
bool row_codes_exact(float w, int64_t sx) {
  if (!(w >= 1.0e-30f) || !(w <= 1.0e30f) || sx + 2 >= 0xFFFF) return false;  // (normal range: no flushed products)
  int e = 0;
  const float fr = std::frexp(w, &e);                  // w = fr * 2^e, fr in [0.5, 1)
  uint64_t m = (uint64_t)std::ldexp((double)fr, 24);   // the 24-bit significand as an integer
  while (m != 0 && !(m & 1u)) m >>= 1;
  if (m * (uint64_t)(sx + 1) >= (1ull << 24)) return false;
  return (double)w * (double)(sx + 1) < 3.0e38;         // (no overflow before the square)
}

bool row_pass_wave_supported(int dtype, int64_t sx, int64_t sy, int64_t sz) {
  // (rows of 1025..2048 voxels: two waves per row, the H = 2 form of the kernel; debug bit 0x4000000 leaves them to the
  // workgroup-phased kernel of edt_rows.hip)
  // (rows of 2049..4096 voxels: four waves per row; debug bit 32 -- the workgroup-phased kernel, which ends at 2048 --
  // sends those to the line pipeline as before)
  const int64_t widest = (debug_mode() & 0x4000000) ? 1024 : (debug_mode() & 32) ? 2048 : 4096;
  return sx >= 1 && sx <= widest && sy * sz < (int64_t)1 << 30 && sx * sy * sz < ((int64_t)1 << 40);
}

template <typename T, int NC, int H = 1>
static int launch_row_wave_tn(const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                              uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                              int to_finite, hipStream_t stream, const void *halo, bool codes, int zero_label, int64_t opitch) {
  const int64_t nby = ceil_div(sy, kBandRows);
  const int64_t ngroups = nby * sz;
  if (ngroups <= 0) return EDT_OK;
  constexpr int GW = H >= 2 ? 1 : kRowWaves;        // groups a workgroup works on at a time
  constexpr int WAVES = H >= 2 ? H : kRowWaves;     // its waves
  const size_t lds = (size_t)(((sx + 3 + 3) & ~(int64_t)3) + 4 * H) * sizeof(float);  // T table + the H >= 2 exchange words
  int64_t blocks = ceil_div(ngroups, GW);
  const int64_t resident = 256 * 8 * (kRowWaves / WAVES > 0 ? kRowWaves / WAVES : 1);  // persistent grid: the T table is built once per workgroup
  if (blocks > resident) blocks = resident;
  const int xcd_sched = row_xcd_schedule(nby, sz, &blocks, GW, resident / 8);
#define LAUNCH(Z, F, C)                                                                                   \
  hipLaunchKernelGGL((k_row_pass_wave<T, NC, Z, F, C, H>), dim3((unsigned)blocks), dim3(WAVES * 64), lds, stream,  \
                     (const T *)labels, out, nz_y, ys_y, zs_y, (int)sx, (int)sy, (int)sz, w, bb, to_finite, \
                     (int)nby, (int)ngroups, xcd_sched, (const T *)halo, zero_label, opitch)
#define LAUNCH_C(Z, F) do { if (codes) LAUNCH(Z, F, true); else LAUNCH(Z, F, false); } while (0)
  const bool full = sx == 64 * NC * H;
  if (zs_y != nullptr) { if (full) LAUNCH_C(true, true); else LAUNCH_C(true, false); }
  else { if (full) LAUNCH_C(false, true); else LAUNCH_C(false, false); }
#undef LAUNCH_C
#undef LAUNCH
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

template <typename T>
static int launch_row_wave_t(const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                             uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                             int to_finite, hipStream_t stream, const void *halo, bool codes, int zero_label, int64_t opitch) {
  const int64_t nc = ceil_div(sx, 64);
#define GO(N) return launch_row_wave_tn<T, N>(labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream, halo, codes, zero_label, opitch)
  if (nc <= 1) GO(1);
  if (nc <= 2) GO(2);
  if (nc <= 4) GO(4);
  if (nc <= 8) GO(8);
  if (nc <= 16) GO(16);
#undef GO
  // rows of 1025..2048 voxels: two waves per row (H = 2), halves of 10, 12, 14 or 16 chunks
#define GO2(N) return launch_row_wave_tn<T, N, 2>(labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream, halo, codes, zero_label, opitch)
  if (nc <= 20) GO2(10);
  if (nc <= 24) GO2(12);
  if (nc <= 28) GO2(14);
  if (nc <= 32) GO2(16);
#undef GO2
  // rows of 2049..4096 voxels: four waves per row (H = 4), parts of 12 or 16 chunks
#define GO4(N) return launch_row_wave_tn<T, N, 4>(labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream, halo, codes, zero_label, opitch)
  if (nc <= 48) GO4(12);
  GO4(16);
#undef GO4
}

int launch_row_pass_wave(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                         uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                         int to_finite, hipStream_t stream, const void *halo, uint16_t *codes, int zero_label, int64_t codes_pitch) {
  // codes != nullptr: the 16-bit distance indices go there and `out` is not touched; codes_pitch > 0: its slices lie that many
  // elements apart (the padded pitch of the index buffer), else sx * sy like everything else
  const int64_t opitch = (codes != nullptr && codes_pitch > 0) ? codes_pitch : sx * sy;
  if (codes != nullptr) out = reinterpret_cast<float *>(codes);
#define ROW_WAVE(T) \
  return launch_row_wave_t<T>(labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream, halo, codes != nullptr, zero_label, opitch)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: ROW_WAVE(uint8_t);
    case EDT_U16: ROW_WAVE(uint16_t);
    case EDT_U32: ROW_WAVE(uint32_t);
    case EDT_U64: ROW_WAVE(uint64_t);
    case EDT_F32: ROW_WAVE(float);
    case EDT_F64: ROW_WAVE(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef ROW_WAVE
}

}  // namespace edt_amd
