// edt_rowquad.hip -- pass 1 (x axis) for gfx950 with FOUR CONSECUTIVE VOXELS PER LANE: 16-byte loads, 8-byte stores.
//
// The same closed form as edt_rowwave.hip (k_row_pass_wave: d(i) = min(T[i-s+1], T[e-i+1]) inside the maximal run [s, e] of
// one non-zero label; reference: squared_edt_1d_multi_seg, src/edt.hpp:70-119) in its 16-bit index form (the kernel stores
// k = min(i-s+1, e-i+1), 0 for background, 0xFFFF where neither side has a boundary: row_codes_exact), and the same three
// bit words per voxel column and band of 32 rows for the column passes (nz, ys, zs; layout [z][y/32][x]).
//
// Why a second kernel: round 5's probes (profiles/r05_rowmap_probe.txt) showed what the 0.205 ms of the one-voxel-per-lane
// kernel are made of -- a bare copy with its mapping takes 0.157 ms, the same copy that also reads the slice below (the `zs`
// bits; an L2 hit) 0.221 ms, with the left neighbour too 0.230 ms: the RE-READS cost 70 us, as 256-byte requests.  With 16
// bytes per lane the slice below costs 5 us (0.156 -> 0.161 ms) and the left neighbour is a DPP shift inside the wave.  So here
// a lane holds voxels 4l .. 4l+3 of a 256-voxel piece of the row:
//   run starts   f_j = (label_j != label_{j-1}), label_{-1} from lane l-1 (wave_shr:1; the piece before through readlane);
//   masks        B_j = ballot(f_j) (SGPRs), A = B_0 | .. | B_3: "this lane has a start"; the last / first start of a piece,
//                carried from piece to piece on the scalar unit as in k_row_pass_wave;
//   per lane     the nearest lane with a start on either side by a bit scan of A, WHICH of its four voxels through one
//                ds_bpermute of the lanes' 4-bit start nibbles; then per voxel the starts of the own nibble first;
//   fast paths   (wave-uniform) a piece without any start, a piece without background;
//   bit words    w = 2 w + (compare) per voxel and plane: v_cmp + v_addc_co_u32, flipped once per 32 rows, stored 16 bytes
//                at a time.
// Serves uint32 labels, rows of whole 16-byte granules up to 1024 voxels, the index form; everything else stays on
// k_row_pass_wave (debug bit 0x40000000 keeps every call there).
#include "edt_common.h"
#include "edt_kernels.h"

namespace edt_amd {

namespace {

constexpr int kQuadWaves = 4;  // waves per workgroup (independent: each takes its own groups of 32 rows)

using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
// w = 2*w + (a != b)
__device__ __forceinline__ void shift_in_ne(uint32_t &w, uint32_t a, uint32_t b) {
  asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(a), "v"(b) : "vcc");
}
// the value of `v` in lane l-1; lane 0 keeps `first`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t first) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

}  // namespace

// NQ pieces of 256 voxels per row (sx <= 256 NQ); FULL: sx == 256 NQ.
template <int NQ, bool HAS_Z, bool FULL>
__global__ void __launch_bounds__(kQuadWaves * 64)
k_row_pass_quad(const uint32_t *__restrict__ labels, uint16_t *__restrict__ codes, uint32_t *__restrict__ nz_y,
                uint32_t *__restrict__ ys_y, uint32_t *__restrict__ zs_y, int sx, int sy, int sz, int bb, int nby, int ngroups,
                int xcd_sched, const uint32_t *__restrict__ halo) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int64_t sxy = (int64_t)sx * sy;
  const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes below this one
  const unsigned long long gt_mask = lane < 63 ? (~0ull << (lane + 1)) : 0ull;
  const int pre0 = bb ? 0 : -(1 << 20);   // where the first run of a row starts, as the distances see it
  const int suf0 = bb ? sx : (1 << 20);   // ... and one past the last voxel of the last one

  // work distribution: edt_rowwave.hip (every XCD takes the y-bands congruent to its index and walks z in order, so
  // that the slice below is an L2 hit)
  const bool by_xcd = xcd_sched != 0;
  const int xcd = (int)(blockIdx.x & 7), nyk = by_xcd ? (nby - xcd + 7) >> 3 : 0;
  const int first = by_xcd ? (int)(blockIdx.x >> 3) * kQuadWaves + wave : (int)blockIdx.x * kQuadWaves + wave;
  const int step = by_xcd ? (int)(gridDim.x >> 3) * kQuadWaves : (int)gridDim.x * kQuadWaves;
  const int count = by_xcd ? nyk * sz : ngroups;
  for (int i = first; i < count; i += step) {
    const int z = by_xcd ? i / nyk : i / nby;
    const int yb = by_xcd ? xcd + 8 * (i - z * nyk) : i - z * nby;
    const int y0 = yb * 32;
    const int nrows = (sy - y0) < 32 ? (sy - y0) : 32;
    const uint32_t *base = labels + ((int64_t)z * sy + y0) * sx;
    const rsrc_t rs_lab = make_rsrc(base);
    const rsrc_t rs_bel = make_rsrc((HAS_Z && z > 0) ? base - sxy : (HAS_Z && halo != nullptr) ? halo + (int64_t)y0 * sx : base);
    const rsrc_t rs_out = make_rsrc(codes + ((int64_t)z * sy + y0) * sx);

    uint32_t xo[NQ];   // per-lane byte offset of the quad inside a row of labels
    bool ok[NQ];       // the quad exists (rows end on a quad boundary: sx % 4 == 0)
    v4u above[NQ];
    uint32_t nzw[NQ][4], ysw[NQ][4], zsw[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int x = 256 * q + 4 * lane;
      ok[q] = FULL || x < sx;
      xo[q] = (uint32_t)(ok[q] ? x : sx - 4) * 4u;  // (a quad that exists: its compares are masked below)
#pragma unroll
      for (int j = 0; j < 4; ++j) nzw[q][j] = ysw[q][j] = zsw[q][j] = 0u;
      above[q] = (v4u){0u, 0u, 0u, 0u};
      if (y0 > 0) above[q] = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(base - sx), xo[q], 0, 0);
    }
    // loads of row r+1 are in flight under the arithmetic of row r (one in-order counter for loads and stores on gfx9:
    // the stores of row r-1 are issued after them -- edt_rowwave.hip)
    v4u lab[NQ], below[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      lab[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_lab, xo[q], 0, 0);
      below[q] = HAS_Z ? __builtin_amdgcn_raw_buffer_load_b128(rs_bel, xo[q], 0, 0) : lab[q];
    }
    v2u pend[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) pend[q] = (v2u){0u, 0u};
#pragma unroll 1
    for (int r = 0; r < nrows; ++r) {
      // ---- run starts: masks in SGPRs, the own nibble in a VGPR; bit words ----
      unsigned long long B[NQ][4], A[NQ];
      uint32_t nib[NQ];
      uint32_t all_fg = 0;
      unsigned long long any_start = 0;
      uint32_t carry = 0;  // the voxel before the piece (piece 0: voxel 0 is its own left neighbour)
      v4u cur[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const v4u v = lab[q];
        cur[q] = v;
        const uint32_t left0 = from_lane_below(v[3], q == 0 ? v[0] : carry);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)v[3], 63);
        const bool f0 = ok[q] && v[0] != left0, f1 = ok[q] && v[1] != v[0], f2 = ok[q] && v[2] != v[1], f3 = ok[q] && v[3] != v[2];
        B[q][0] = __ballot(f0); B[q][1] = __ballot(f1); B[q][2] = __ballot(f2); B[q][3] = __ballot(f3);
        A[q] = B[q][0] | B[q][1] | B[q][2] | B[q][3];
        nib[q] = (f0 ? 1u : 0u) | (f1 ? 2u : 0u) | (f2 ? 4u : 0u) | (f3 ? 8u : 0u);
        any_start |= A[q];
        const bool fg = (v[0] != 0u) & (v[1] != 0u) & (v[2] != 0u) & (v[3] != 0u);
        all_fg |= (__ballot(fg || !ok[q]) == ~0ull ? 1u : 0u) << q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          shift_in_ne(nzw[q][j], v[j], 0u);
          shift_in_ne(ysw[q][j], v[j], above[q][j]);
          if (HAS_Z) shift_in_ne(zsw[q][j], v[j], below[q][j]);
        }
        above[q] = v;
      }
      // ---- the previous row's indices leave; the next row's labels are asked for ----
      if (r > 0) {
        const uint32_t poff = (uint32_t)((r - 1) * sx) * 2u;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (ok[q]) __builtin_amdgcn_raw_buffer_store_b64(pend[q], rs_out, (uint32_t)(256 * q + 4 * lane) * 2u, poff, 0);
      }
      {
        const int rn = r + 1 < nrows ? r + 1 : r;
        const uint32_t soff = (uint32_t)(rn * sx) * 4u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          lab[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_lab, xo[q], soff, 0);
          below[q] = HAS_Z ? __builtin_amdgcn_raw_buffer_load_b128(rs_bel, xo[q], soff, 0) : lab[q];
        }
      }
      // ---- last / first start of every piece, carried across pieces (scalar unit) ----
      int pre[NQ], suf[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { pre[q] = pre0; suf[q] = suf0; }
      if (any_start) {
        int last = pre0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          pre[q] = last;
          if (A[q]) {
            const int pl = 63 - __builtin_clzll(A[q]);
            const int j = ((B[q][3] >> pl) & 1u) ? 3 : ((B[q][2] >> pl) & 1u) ? 2 : ((B[q][1] >> pl) & 1u) ? 1 : 0;
            last = 256 * q + 4 * pl + j;
          }
        }
        int nxt = suf0;
#pragma unroll
        for (int q = NQ - 1; q >= 0; --q) {
          suf[q] = nxt;
          if (A[q]) {
            const int pr = __builtin_ctzll(A[q]);
            const int j = ((B[q][0] >> pr) & 1u) ? 0 : ((B[q][1] >> pr) & 1u) ? 1 : ((B[q][2] >> pr) & 1u) ? 2 : 3;
            nxt = 256 * q + 4 * pr + j;
          }
        }
      }
      // ---- indices ----
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int x0 = 256 * q + 4 * lane;
        int k[4];
        if (A[q] == 0) {
          // no start inside the piece (wave-uniform): one run in from the left and out to the right
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int il = x0 + j - pre[q] + 1, ir = suf[q] - (x0 + j);
            k[j] = il < ir ? il : ir;
          }
        } else {
          // the nearest lanes with a start on either side, and which of their voxels it is
          const unsigned long long m1 = A[q] & lt_mask, m2 = A[q] & gt_mask;
          const int pl = m1 ? 63 - __builtin_clzll(m1) : lane, pr = m2 ? __builtin_ctzll(m2) : lane;
          const uint32_t nl = (uint32_t)__builtin_amdgcn_ds_bpermute(pl << 2, (int)nib[q]);
          const uint32_t nr = (uint32_t)__builtin_amdgcn_ds_bpermute(pr << 2, (int)nib[q]);
          const int s_in = m1 ? 256 * q + 4 * pl + (31 - __builtin_clz(nl | 0u ? nl : 1u)) : pre[q];
          const int e_in = m2 ? 256 * q + 4 * pr + __builtin_ctz(nr ? nr : 1u) : suf[q];
          const uint32_t nb = nib[q];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t lo = nb & ((2u << j) - 1u);          // starts at or before voxel j
            const uint32_t hi = j < 3 ? nb >> (j + 1) : 0u;     // starts after it
            const int s = lo ? x0 + 31 - __builtin_clz(lo) : s_in;
            const int e1 = hi ? x0 + j + 1 + __builtin_ctz(hi) : e_in;
            const int il = x0 + j - s + 1, ir = e1 - (x0 + j);
            k[j] = il < ir ? il : ir;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = k[j] < 0xFFFF ? k[j] : 0xFFFF;  // ("no boundary": far beyond the row)
        if (!((all_fg >> q) & 1u)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) k[j] = cur[q][j] != 0u ? k[j] : 0;
        }
        pend[q] = (v2u){(uint32_t)k[0] | ((uint32_t)k[1] << 16), (uint32_t)k[2] | ((uint32_t)k[3] << 16)};
      }
    }
    {
      const uint32_t poff = (uint32_t)((nrows - 1) * sx) * 2u;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (ok[q]) __builtin_amdgcn_raw_buffer_store_b64(pend[q], rs_out, (uint32_t)(256 * q + 4 * lane) * 2u, poff, 0);
    }
    // ---- the three bit words of this (z, y-band): 16 bytes per lane and plane ----
    const int sh = 32 - nrows;
    const int64_t wbase = ((int64_t)z * nby + yb) * sx;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (!ok[q]) continue;
      const int x = 256 * q + 4 * lane;
      v4u nzv, ysv, zsv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        nzv[j] = __brev(nzw[q][j]) >> sh;
        // row 0 of the volume starts a run along y, slice 0 starts every run along z
        ysv[j] = (__brev(ysw[q][j]) >> sh) | (y0 == 0 ? 1u : 0u);
        zsv[j] = (z == 0 && halo == nullptr) ? (0xFFFFFFFFu >> sh) : (__brev(zsw[q][j]) >> sh);
      }
      *reinterpret_cast<v4u *>(nz_y + wbase + x) = nzv;
      *reinterpret_cast<v4u *>(ys_y + wbase + x) = ysv;
      if (HAS_Z) *reinterpret_cast<v4u *>(zs_y + wbase + x) = zsv;
    }
  }
}

bool row_pass_quad_supported(int dtype, const void *labels, const void *halo, const uint16_t *codes, const uint32_t *nz_y,
                             const uint32_t *ys_y, const uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz) {
  if (dtype != EDT_U32 || codes == nullptr || sx < 4 || sx > 1024 || (sx % 4) != 0 || (debug_mode() & 0x40000000)) return false;
  if (sy * sz >= ((int64_t)1 << 30) || sx * sy * sz >= ((int64_t)1 << 40)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(halo) | reinterpret_cast<uintptr_t>(nz_y) |
                       reinterpret_cast<uintptr_t>(ys_y) | reinterpret_cast<uintptr_t>(zs_y);
  return (al % 16) == 0 && (reinterpret_cast<uintptr_t>(codes) % 8) == 0;
}

template <int NQ>
static int launch_row_quad_n(const uint32_t *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sx,
                             int64_t sy, int64_t sz, int bb, hipStream_t stream, const uint32_t *halo) {
  const int64_t nby = ceil_div(sy, kBandRows);
  const int64_t ngroups = nby * sz;
  if (ngroups <= 0) return EDT_OK;
  int64_t blocks = ceil_div(ngroups, kQuadWaves);
  const int64_t resident = 256 * 8;
  if (blocks > resident) blocks = resident;
  int xcd_sched = 0;
  if (nby >= 8 && sz >= 2 && !(debug_mode() & 256)) {
    // (edt_rowwave.hip: row_xcd_schedule)
    int64_t bx = ceil_div(ceil_div(nby, 8) * sz, kQuadWaves);
    if (bx > resident / 8) bx = resident / 8;
    blocks = bx * 8;
    xcd_sched = 1;
  }
#define LAUNCH(Z, F)                                                                                                        \
  hipLaunchKernelGGL((k_row_pass_quad<NQ, Z, F>), dim3((unsigned)blocks), dim3(kQuadWaves * 64), 0, stream, labels, codes, nz_y, \
                     ys_y, zs_y, (int)sx, (int)sy, (int)sz, bb, (int)nby, (int)ngroups, xcd_sched, halo)
  const bool full = sx == 256 * NQ;
  if (zs_y != nullptr) { if (full) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (full) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_row_pass_quad(const void *labels, uint16_t *codes, uint32_t *nz_y, uint32_t *ys_y, uint32_t *zs_y, int64_t sx, int64_t sy,
                         int64_t sz, int bb, hipStream_t stream, const void *halo) {
  const uint32_t *lab = static_cast<const uint32_t *>(labels), *hl = static_cast<const uint32_t *>(halo);
  const int64_t nq = ceil_div(sx, 256);
  if (nq <= 1) return launch_row_quad_n<1>(lab, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, hl);
  if (nq <= 2) return launch_row_quad_n<2>(lab, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, hl);
  if (nq <= 3) return launch_row_quad_n<3>(lab, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, hl);
  return launch_row_quad_n<4>(lab, codes, nz_y, ys_y, zs_y, sx, sy, sz, bb, stream, hl);
}

}  // namespace edt_amd
