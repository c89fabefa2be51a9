#pragma once
#ifndef EDT_Q16_NT_FILL
#define EDT_Q16_NT_FILL 0
#endif
#if EDT_Q16_NT_FILL
#define EDT_COLWAVE_CODE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define EDT_COLWAVE_CODE_LOAD(p) (*(p))
#endif
// edt_colwave_kernel.h -- wave-autonomous LDS-tiled column pass (passes 2 and 3) for gfx950: the kernel
// template and its per-wave-shape launcher.  Included by edt_colwave_cw*.hip, ONE translation unit (= one
// device code object) per wave shape CW: the kernel is ~85 KB of code, more than the instruction cache,
// and its speed turned out to depend on where the loader places it (adding the CW = 1 variants to a
// common code object moved the CW = 4 kernel and cost the 512-row y pass 10 %); separate code objects
// keep the placement of each shape independent of the others (and the build parallel).
//
// Work decomposition
//   workgroup = one tile of 32 adjacent columns (128 contiguous bytes per row, so every global
//               access is a full cache line) x the WHOLE scan axis, staged once in LDS by
//               direct global->LDS loads (global_load_lds_dwordx4: no VGPR round trip);
//   wave      = CW = 64/NBP of those columns x all NBP bands (a band = 32 rows): the hull
//               build, the hull merges and the evaluation of a column only involve lanes of
//               ONE wave, so after the tile has landed the waves run without any workgroup
//               barrier and hide each other's LDS/fp64 latencies (the previous design,
//               edt_tiled.hip, synchronised 1024 threads four times per tile while most of
//               them idled in the merge rounds);
//   lane      = (column c, band b): its 32 rows live in VGPRs for the whole kernel.
// The tile is read from HBM exactly once and written exactly once (8 B/voxel of traffic
// against 12 B/voxel in the reference's data-movement model, which re-reads the labels).
// Two forms per tile after the fill: hulls (hull_tile) or, where the field is small everywhere, a branch-free
// window over the rows (brute_tile).  Variants: XF -- the FIRST column pass reads pass 1 as 16-bit distance indices
// (6 B/voxel) and turns them into the fp32 tile through VGPRs (edt_colwave_lane.h: code_value); SC -- the rows leave
// for the slab records of a Z-sharded run instead of their places in F.
//
// LDS image: fp32 tile [rows][32], the columns of a wave XOR-rotated by CW columns per band
// (edt_colwave_lane.h: addr_tile) so that "every lane reads its own row" is bank-conflict free;
// because global_load_lds writes LDS linearly (lane i -> base + 16*i, or 4*i for the 2-column waves
// of 1024-row axes and for rows that are not 16-byte aligned), the rotation is applied to the
// per-lane SOURCE address, and again when the results are streamed back.
// Axes: up to 2048 rows (NBP = 2..64 bands per column, CW = 32..1 columns per wave; 16-column tiles for CW <= 2).
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

// cache policy of the streamed tile (read once, written once): experiment knobs
#ifndef EDT_TILE_LOAD_AUX
#define EDT_TILE_LOAD_AUX 2  // nt: ~5 % faster tile fill than the default policy (measured)
#endif
#ifdef EDT_TILE_STORE_NT
#define EDT_TILE_STORE(p, v) __builtin_nontemporal_store(v, p)
#else
#define EDT_TILE_STORE(p, v) (*(p) = (v))
#endif

#define EDT_LANE __device__ __forceinline__
#define EDT_LANE_MEMBER __device__ __forceinline__
#include "edt_colwave_lane.h"

namespace edt_amd {

// arguments of the windowed path of the column kernel
struct BruteArgs {
  uint32_t limit_bits;  // a tile takes the path when the bit pattern of its largest field value is <= this; 0 = never
  int x32;              // 1: candidates are fp32 sums, c_d exactly representable up to the limit; 2: fp32 fma candidates of
                        //    a rounded c_d, equal to the reference's values because its fp64 sums are exact (brute_f32e_prefix)
  int force;            // diagnostics: every tile takes the path
  int stride;           // 1: every row is evaluated; 2: only the even rows are (and written), see BruteSteps
  // stride 2 only, or nullptr: the even rows leave for a compact array instead of their places in F --
  // row r of column (x, outer o) -> compact[x + o * c_outer + (r / 2) * c_row2]   (edt_kernels.h: ColumnOut)
  float *compact;
  int64_t c_outer, c_row2;
  int c_al;             // its rows are whole 16-byte granules
  // list mode (nullptr: every tile of the grid): the launch serves the tiles the 16-bit integer kernel handed over
  // (edt_colq16.hip) -- workgroup b takes tile list_ids[b] (already in the XCD-aware order) if b < *list_count
  const uint32_t *list_count, *list_ids;
};
int window_limit();  // edt_colwave.hip: largest window (rows) the windowed path is used for

namespace {

__device__ __forceinline__ void wave_sync() {
  // lanes of one wave exchange data through LDS: the hardware executes a wave's LDS
  // instructions in order, the fence keeps the compiler from reordering them
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Run structure across the bands of a column: lane = c + CW*band, so "the bands below / above"
// are the lanes CW, 2*CW, ... away.  Exclusive prefix max of the last run start, exclusive
// suffix min of the first run start (Hillis-Steele over lane shuffles, log2(NBP) steps each).
template <int CW>
__device__ __forceinline__ void scan_runs(edt_lane::Lane &L, int lane) {
  int x = __shfl_up(edt_lane::band_last_start(L.rsw, L.row0), CW);
  if (lane < CW) x = -1;
  int y = __shfl_down(edt_lane::band_first_start(L.rsw, L.row0, L.n) - 1, CW);
  if (lane >= 64 - CW) y = L.n - 1;
#pragma unroll
  for (int d = CW; d < 64; d <<= 1) {
    const int tx = __shfl_up(x, d);
    const int ty = __shfl_down(y, d);
    if (lane >= d) x = tx > x ? tx : x;
    if (lane + d < 64) y = ty < y ? ty : y;
  }
  L.lo_in = x;
  L.hi_out = y;
}

// The same scan over the "break" words (edt_colwave_lane.h: brute_flat_reach): last break row in an earlier
// band of the column (-1: none), first break row in a later band (n: none).
template <int CW>
__device__ __forceinline__ void scan_breaks(uint32_t brk, int row0, int n, int lane, int &blo_in, int &bhi_out) {
  int x = __shfl_up(brk ? row0 + 31 - __builtin_clz(brk) : -1, CW);
  if (lane < CW) x = -1;
  int y = __shfl_down(brk ? row0 + __builtin_ctz(brk) : n, CW);
  if (lane >= 64 - CW) y = n;
#pragma unroll
  for (int d = CW; d < 64; d <<= 1) {
    const int tx = __shfl_up(x, d);
    const int ty = __shfl_down(y, d);
    if (lane >= d) x = tx > x ? tx : x;
    if (lane + d < 64) y = ty < y ? ty : y;
  }
  blo_in = x;
  bhi_out = y;
}

}  // namespace

// The windowed path of one lane (edt_colwave_lane.h: brute_band).  History of where it lives, because it matters:
// inlined next to a hull path that kept its 32 rows in registers from the tile choice on, the two pushed each other
// into scratch (the hull path alone needed 127 of 128 VGPRs); as a non-inlined function it had its own allocation,
// but every call saved and restored 16-19 callee-saved registers through scratch -- 0.43 GB of extra fabric traffic
// per 512^3 launch (PMC).  Since the hull path re-reads its rows from the LDS tile (hull_tile below), every kernel
// builds with 114-121 VGPRs and NO scratch with both paths inlined; that is the default.  EDT_BRUTE_INLINE can
// be set to __attribute__((noinline)) to get the call back; tools/check_spills.py / tests/test_kernel_resources.py
// watch the scratch instructions of the built kernels.
#ifndef EDT_BRUTE_INLINE
#define EDT_BRUTE_INLINE __forceinline__
#endif
template <int CW, bool BB, bool X32>
__device__ EDT_BRUTE_INLINE void brute_tile(float *tile, const uint32_t *alive, const uint32_t *rsp,
                                                     const uint32_t *lohi, const uint32_t *bscan, int n, int NB,
                                                     int cols_left, int band, int col, float w, int epi,
                                                     float *dst0, int64_t dstride) {
  using namespace edt_lane;
  BruteLane BL;
  BL.tile = tile;
  BL.col = col;
  BL.band = band;
  BL.row0 = band * 32;
  BL.n = n;
  BL.rsw = rsp[addr_word<CW>(col, band)];
  BL.brk = alive[addr_word<CW>(col, band)];
  const uint32_t bs = bscan[addr_word<CW>(col, band)];
  BL.blo_in = (int)(bs & 0xFFFFu) - 1;
  BL.bhi_out = (int)(bs >> 16);
  const uint32_t lh = lohi[addr_word<CW>(col, band)];
  BL.lo_in = (int)(lh & 0xFFFFu) - 1;
  BL.hi_out = (int)(lh >> 16) - 1;
  BL.w2 = (double)(w * w);
  BL.w2f = w * w;
  BL.live = col < cols_left && band < NB;
  const bool colok = col < cols_left;
  // (a pointer that crosses a call is generic: say that it is global memory, or the rows leave through flat stores)
  auto *gdst = (__attribute__((address_space(1))) float *)dst0;
  // (bit 10 of epi: dstride is the distance between EVEN rows of a compact destination)
  const bool compact = (epi & 0x400) != 0;
  // (bit 2 of epi = kEpiStream: the call's results, streamed)
  const bool stream = (epi & kEpiStream) != 0;
  auto store = [&](int row, float v) {
    if (row < n && colok) {
      if (stream) __builtin_nontemporal_store(v, &gdst[(int64_t)(compact ? row >> 1 : row) * dstride]);
      else gdst[(int64_t)(compact ? row >> 1 : row) * dstride] = v;
    }
  };
  // (bit 8 of epi: only the even rows are evaluated and written -- the doubled grids of the voxel-graph transform;
  // carried in an existing argument: the kernel around this call is sensitive to its signature, see hull path)
  // (bit 11 of epi: the fp32 candidates are fma's of a ROUNDED c_d -- brute_f32e_prefix -- the exit tests allow for it)
  if (epi & 0x100) brute_band<CW, BB, X32, 2>(BL, epi & 0xA03, store);
  else brute_band<CW, BB, X32, 1>(BL, epi & 0xA03, store);
}



// The hull path of one lane (phases 1-3 of edt_colwave_lane.h), inlined into the kernel (as a callee it would save
// and restore 48 callee-saved registers per tile: cfg2 0.69 -> 1.04 ms, measured).  It RE-READS the lane's 32 rows
// from the LDS tile instead of inheriting the registers the tile choice loaded them into: 32 LDS reads per lane
// that keep those 32 live ranges out of everything in between -- with the rows held from the top of the kernel the
// hull code sat at 127 VGPRs and any change anywhere in the kernel body (another call, another argument) tipped its
// hot loops into scratch with byte-identical hull code (cfg2 0.69 -> 1.05 ms, measured twice).  It leaves the
// results in the tile for the workgroup's write-back.
template <int CW, bool BB>
static __device__ __forceinline__ void hull_tile(float *tile, uint32_t *alive, const uint32_t *rsp, int colc, int band,
                                                    int n, float w, uint32_t nzw, uint32_t rsw, int lo_in, int hi_out,
                                                    uint32_t fl0, uint32_t need, int epi, int dbg, int lane) {
  using namespace edt_lane;
  constexpr int NBP = 64 / CW;
  constexpr int TC = TileGeom<CW>::kCols;
  Lane L;
  L.tile = tile;
  L.alive = alive;
  L.rsp = rsp;
  L.colc = colc;
  L.band = band;
  L.row0 = band * 32;
  L.n = n;
  L.w2 = (double)(w * w);
  L.nzw = nzw;
  L.rsw = rsw;
  L.lo_in = lo_in;
  L.hi_out = hi_out;
  L.own = 0;
  float f[32];
  {
    const float *own = tile + addr_tile<CW>(colc, L.row0);
#pragma unroll
    for (int r = 0; r < 32; ++r) f[r] = own[r * TC];
  }
  const float fprev = __shfl_up(f[31], CW);  // last row of the band below (unused for band 0)
  // (dbg: diagnostics only -- bit1 skips the hull build, bit2 the merges, bit3 the evaluation)
  // All-flat shortcut (edt_colwave_lane.h: flat_word): a wave whose columns are flat wherever a run
  // continues needs no hull at all -- every foreground row owns itself.  (debug bit 16 switches it off.)
  bool all_flat = false;
  if (!(EDT_DIAG_BITS(dbg, 2) | (dbg & (16 | 0x10000)))) all_flat = __ballot((fl0 & need) != need) == 0ull;
  uint32_t aw = L.nzw;
  if (all_flat) {
    L.own = L.nzw;
  } else {
    Hull1 H;
    if (EDT_DIAG_BITS(dbg, 2)) { H.aw = L.nzw; H.flat = 0; H.nb0 = H.nb1 = H.nb31 = 0.0; }
    else H = phase1_hull<CW>(L, f, fprev, fl0);
    aw = H.aw;
    const uint32_t flat = H.flat;
    alive[addr_word<CW>(L.colc, L.band)] = aw;
    wave_sync();
    if (!EDT_DIAG_BITS(dbg, 4)) {
      // the merge rounds change nothing for a wave whose band boundaries are all quiet
      uint32_t prev_aw = __shfl_up(aw, CW), prev_rs = __shfl_up(L.rsw, CW);
      const double prev_nb31 = __shfl_up(H.nb31, CW);
      if (lane < CW) { prev_aw = 0; prev_rs = 0; }
      const bool quiet = boundary_quiet(L, H, prev_aw, prev_rs, prev_nb31);
      if (__ballot(!quiet) != 0ull || (dbg & 32)) {
#pragma unroll
        for (int half = 1; half < NBP; half <<= 1) {
          phase2_merge<CW>(L, half);
          wave_sync();
        }
      }
    }
    aw = alive[addr_word<CW>(L.colc, L.band)];
    {
      // self-owned rows need bit 31 of the band below and bit 0 of the band above
      uint32_t prev31 = __shfl_up(aw >> 31, CW);
      uint32_t next0 = __shfl_down((L.nzw & 1u) | ((L.rsw & 1u) << 1) | ((aw & 1u) << 2) | ((flat & 1u) << 3), CW);
      if (lane < CW) prev31 = 0;
      if (lane >= 64 - CW) next0 = 0;
      L.own = own_mask(L.nzw, L.rsw, aw, flat, prev31, next0 & 1u, (next0 >> 1) & 1u, (next0 >> 2) & 1u,
                       (next0 >> 3) & 1u);
      if (dbg & 16) L.own = 0;  // diagnostics: no self-owned shortcut
    }
  }  // !all_flat
  if (!EDT_DIAG_BITS(dbg, 8)) phase3_eval<CW, BB>(L, aw, f, epi);
  wave_sync();  // every lane of the wave is done reading the tile

  // ---- results -> LDS (in place); the workgroup streams the tile back after its barrier ----
  float *own = tile + addr_tile<CW>(L.colc, L.row0);
  if (!EDT_DIAG_BITS(dbg, 0x200)) {  // (diagnostics: bit 9 leaves the tile as it was loaded)
#pragma unroll
    for (int r = 0; r < 32; ++r) own[r * TC] = f[r];
  }
}

template <int CW, bool BB, bool XF, bool SC>
__global__ void __launch_bounds__(64 * edt_lane::TileGeom<CW>::kCols / CW, 4)
k_column_pass_wave(float *__restrict__ F, const uint32_t *__restrict__ nzbits,
                   const uint32_t *__restrict__ rsbits, AxisGeom g, float w, int tiles_x,
                   int epi, int dbg, int aligned16, XFuse xf, const BandScatter *__restrict__ scatter,
                   BruteArgs ba) {
  using namespace edt_lane;
  constexpr int NBP = 64 / CW;  // bands per column handled by a wave (power of two)
  using TG = TileGeom<CW>;
  constexpr int TC = TG::kCols;  // columns of the tile (32, or 16 for the 1024-row wave shape)
  constexpr int W = TC / CW;     // waves per workgroup
  using IO = TileIO<CW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS image: [one band of padding][the tile: NBP bands][one band of padding][alive][rsp][lohi][bscan]; the
  // padding bands hold +inf for the windowed path (edt_colwave_lane.h: brute_band)
  constexpr int kPad = TG::kBandFloats;
  float *tile = reinterpret_cast<float *>(smem) + kPad;                            // [NBP*32][TC] (+ band padding)
  uint32_t *alive = reinterpret_cast<uint32_t *>(tile + NBP * TG::kBandFloats + kPad);  // [NBP][TC]
  uint32_t *rsp = alive + NBP * TG::kBandWords;                                    // [NBP][TC]
  uint32_t *lohi = rsp + NBP * TG::kBandWords;     // [NBP][TC]: (lo_in + 1) | (hi_out + 1) << 16 (windowed path)
  uint32_t *bscan = lohi + NBP * TG::kBandWords;   // [NBP][TC]: the same for the breaks
  // 2 words: largest field value of the tile, "some link of the tile is not flat".  They sit in the lower padding band (so that the LDS image of the
  // 512-row shape is exactly half of a CU's 160 KiB), which is only filled with +inf once every thread has read them.
  uint32_t *tmax = reinterpret_cast<uint32_t *>(smem);

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int n = (int)g.n;
  const int NB = (int)g.nbands;
  int64_t tile_id = blockIdx.x;
  // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement, used for speed only).  XCD x
  // takes the outer indices congruent to x (mod 8) and walks all x-tiles of one outer index back to
  // back, so the tiles that share DRAM pages (the rows of one slice in the Y pass are 2 KiB apart, a
  // tile takes 128 bytes of each) are in flight together on one XCD: Y pass of 512^3 0.231 -> 0.214 ms
  // in back-to-back A/B runs, and no more slow mode; the Z pass is indifferent.  For the 16-column
  // tiles it also puts the two halves of every 128-byte line back to back on one XCD (the second
  // half is an L2 hit).  The grid is rounded up to whole groups of 8 outer indices; debug bit 11
  // restores the plain order.
  // (tile ids fit 31 bits -- the launcher checks: 32-bit divisions, not 64-bit ones, ahead of every wave's first load)
  const uint32_t utx = (uint32_t)tiles_x;
  if (ba.list_count != nullptr) {  // (wave-uniform: kernel argument)
    if (blockIdx.x >= *ba.list_count) return;
    tile_id = ba.list_ids[blockIdx.x];
  } else if (!(dbg & 0x800)) {
    const uint32_t t = (uint32_t)tile_id, x = t & 7u, j = t >> 3;
    const uint32_t jq = j / utx, jr = j - jq * utx;
    tile_id = (int64_t)((jq * 8u + x) * utx + jr);
    if (tile_id >= (int64_t)tiles_x * g.nouter) return;
  }
  const uint32_t oq = (uint32_t)tile_id / utx;
  const int64_t xt = (uint32_t)tile_id - oq * utx, o = oq;
  const int64_t x0 = xt * TC;
  const int64_t st = g.stride;
  float *Ftile = F + x0 + o * g.outer_stride;
  const int cols_left = (int)(g.sx - x0);  // columns of this tile that exist

  // cache policy of the tile fill: streaming (nt) for whole-line tiles; the 16-column tiles must leave
  // their lines in L2 for the workgroup that takes the other half
  constexpr int kLoadAux = CW <= 2 ? 0 : EDT_TILE_LOAD_AUX;
  Lane L;
  L.tile = tile;
  L.alive = alive;
  L.rsp = rsp;
  L.colc = wave * CW + (lane % CW);
  L.band = lane / CW;
  L.row0 = L.band * 32;
  L.n = n;
  L.w2 = (double)(w * w);  // fp32 product widened (src/edt.hpp:181, :258)
  L.nzw = 0;
  L.rsw = 0;
  const bool active = L.colc < cols_left && L.band < NB;
  auto load_bits = [&] {
    if (active) {
      const int64_t widx = (o * g.nbands + L.band) * g.sx + x0 + L.colc;
      L.nzw = nzbits[widx];
      L.rsw = rsbits[widx];
    }
  };
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  constexpr int NI = IO::count(NBP, 4) / W;  // 16-byte tile instructions per wave (8 for every wave shape)
  static_assert(NI * W == IO::count(NBP, 4), "tile instructions divide evenly among the waves");
  v2u q[XF ? NI : 1];  // XF: the wave's share of the tile as 16-bit indices, four per lane and instruction
  if constexpr (!XF) {
    // ---- phase 0: the whole tile, HBM -> LDS --------------------------------------------
    // one instruction = 64 lanes x G floats = 2*G rows of 128 B, all rows in one band
    if (IO::kGran == 4 && aligned16) {
      // (direct global->LDS loads: tools/tileprobe.hip shows the same fill through VGPRs 7 % faster in
      // isolation, but inside this kernel it costs 0.035 ms -- the loading workgroup then competes
      // for issue slots with the one that is computing on the same CU; measured and rejected)
      for (int i = wave; i < IO::count(NBP, 4); i += W) {
        const int row = io_row<CW, 4>(i, lane), gc = io_gcol<CW, 4>(i, lane);
        if (row < n && gc < cols_left)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(Ftile + (int64_t)row * st + gc),
              (__attribute__((address_space(3))) void *)(tile + io_lds_word<CW, 4>(i, 0)), 16, 0, kLoadAux);
      }
    } else {
      for (int i = wave; i < IO::count(NBP, 1); i += W) {
        const int row = io_row<CW, 1>(i, lane), gc = io_gcol<CW, 1>(i, lane);
        if (row < n && gc < cols_left)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(Ftile + (int64_t)row * st + gc),
              (__attribute__((address_space(3))) void *)(tile + io_lds_word<CW, 1>(i, 0)), 4, 0, kLoadAux);
      }
    }
    load_bits();
  } else {
    // ---- phase 0 (index form of pass 1), first half: the bit words, then the indices of the whole tile are
    // requested (8 bytes per lane and instruction); they are turned into the fp32 tile further down, after the run
    // scan, which only waits for the bit words
    load_bits();
    if (IO::kGran == 4 && aligned16) {
      const uint16_t *Ctile = xf.codes + x0 + o * xf.c_outer;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int i = wave + j * W;
        const int row = io_row<CW, 4>(i, lane), gc = io_gcol<CW, 4>(i, lane);
        q[j] = (v2u){0u, 0u};
        if (row < n && gc < cols_left)
          // (plain loads: the indices of a 32-column tile are 64-byte HALF lines -- a non-temporal load lets the line go and
          // the neighbouring tile fetches it again: edt_colq16.hip, profiles/r05_halfline_probe.txt)
          q[j] = EDT_COLWAVE_CODE_LOAD(reinterpret_cast<const v2u *>(Ctile + (int64_t)row * st + gc));
      }
    }
  }
  rsp[addr_word<CW>(L.colc, L.band)] = L.rsw;
  scan_runs<CW>(L, lane);
  if constexpr (XF) {
    // ---- phase 0, second half: indices -> fp32 tile (edt_colwave_lane.h: code_value, four at a time) ----
    const uint16_t *Ctile = xf.codes + x0 + o * xf.c_outer;
    if (IO::kGran == 4 && aligned16) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f ww = {xf.w, xf.w};
      const bool finite_only = xf.flim == 0x7f800000;  // black border: no "no boundary" index, no tofinite (wave-uniform)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int i = wave + j * W;
        const uint32_t q0 = q[j][0], q1 = q[j][1];
        v2f a = {(float)(q0 & 0xFFFFu), (float)(q0 >> 16)};
        v2f b = {(float)(q1 & 0xFFFFu), (float)(q1 >> 16)};
        a = a * ww;  // exact products (row_codes_exact)
        b = b * ww;
        a = a * a;
        b = b * b;
        v4f v = {a.x, a.y, b.x, b.y};
        if (!finite_only) {
          v.x = __int_as_float(min((q0 & 0xFFFFu) == kCodeInf ? 0x7f800000 : __float_as_int(v.x), xf.flim));
          v.y = __int_as_float(min((q0 >> 16) == kCodeInf ? 0x7f800000 : __float_as_int(v.y), xf.flim));
          v.z = __int_as_float(min((q1 & 0xFFFFu) == kCodeInf ? 0x7f800000 : __float_as_int(v.z), xf.flim));
          v.w = __int_as_float(min((q1 >> 16) == kCodeInf ? 0x7f800000 : __float_as_int(v.w), xf.flim));
        }
        *reinterpret_cast<v4f *>(tile + io_lds_word<CW, 4>(i, lane)) = v;
      }
    } else {
      for (int i = wave; i < IO::count(NBP, 1); i += W) {
        const int row = io_row<CW, 1>(i, lane), gc = io_gcol<CW, 1>(i, lane);
        if (row < n && gc < cols_left)
          tile[io_lds_word<CW, 1>(i, lane)] = code_value(Ctile[(int64_t)row * st + gc], xf.w, xf.flim);
      }
    }
  }
  // the rows that complete the last band are not part of the column: 0 for the tile maximum below
  for (int i = (int)threadIdx.x; i < (NB * 32 - n) * TC; i += (int)blockDim.x)
    tile[addr_tile<CW>(i % TC, n + i / TC)] = 0.0f;
  if (threadIdx.x < 2) tmax[threadIdx.x] = 0u;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- own rows -> registers ---------------------------------------------------------------
  float f[32];
  {
    const float *own = tile + addr_tile<CW>(L.colc, L.row0);
#pragma unroll
    for (int r = 0; r < 32; ++r) f[r] = own[r * TC];
  }

  // ---- tiles whose field is small everywhere take the windowed path (edt_colwave_lane.h: brute_band) ------
  const float fprev = __shfl_up(f[31], CW);  // last row of the band below (unused for band 0)
  const uint32_t fl0 = EDT_DIAG_BITS(dbg, 2) ? 0u : flat_word(L, f, fprev);
  const uint32_t need = L.nzw & ~(L.rsw | (L.band == 0 ? 1u : 0u));  // rows that continue a run
  {
    if (ba.limit_bits != 0u) {  // (wave-uniform: kernel argument)
      uint32_t fm = 0;
#pragma unroll
      for (int r = 0; r < 32; ++r) fm = max(fm, __float_as_uint(f[r]));  // non-negative floats order like ints
      if (!active) fm = 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) fm = max(fm, (uint32_t)__shfl_xor((int)fm, d));
      const uint32_t brk = need & ~fl0;  // links that are not flat
      const bool any_brk = __ballot(brk != 0u) != 0ull;
      if (lane == 0) {
        atomicMax(tmax, fm);
        if (any_brk) tmax[1] = 1u;  // (a tile without any is left to the all-flat shortcut of the hull path)
      }
      int blo_in, bhi_out;
      scan_breaks<CW>(brk, L.row0, n, lane, blo_in, bhi_out);
      alive[addr_word<CW>(L.colc, L.band)] = brk;
      lohi[addr_word<CW>(L.colc, L.band)] = (uint32_t)(L.lo_in + 1) | ((uint32_t)(L.hi_out + 1) << 16);
      bscan[addr_word<CW>(L.colc, L.band)] = (uint32_t)(blo_in + 1) | ((uint32_t)bhi_out << 16);
      __syncthreads();
      const uint32_t tile_max = tmax[0], tile_brk = tmax[1];
      // two forms: windows (fields that are small everywhere) or hulls (the rest)
      if (tile_max <= ba.limit_bits && (tile_brk != 0u || ba.force)) {
        __syncthreads();  // (every thread has read tmax: the padding band it sits in may be filled now)
        // +inf around the columns: the padding bands and the rows that complete the last band
        for (int i = (int)threadIdx.x; i < 32 * TC; i += (int)blockDim.x)
          tile[addr_tile<CW>(i % TC, -32 + i / TC)] = INFINITY;
        for (int i = (int)threadIdx.x; i < ((NB + 1) * 32 - n) * TC; i += (int)blockDim.x)
          tile[addr_tile<CW>(i % TC, n + i / TC)] = INFINITY;
        __syncthreads();
        // (its own function: the windowed path and the hull path each get a register allocation of their own)
        const int band2 = wave * (64 / TC) + lane / TC, col2 = lane % TC;
        float *dst0;  // row 0 of the column this lane writes
        int64_t dstep = st;
        bool compact = false;
        if constexpr (SC) {
          const int b = band2 < BandScatter::kBands ? band2 : 0;
          dst0 = scatter->rows[b] + o * scatter->ostride[b] + x0 + col2 - (int64_t)band2 * 32 * st;
        } else {
          dst0 = Ftile + col2;
          if (ba.compact != nullptr) {  // (wave-uniform: kernel argument)
            dst0 = ba.compact + x0 + col2 + o * ba.c_outer;
            dstep = ba.c_row2;
            compact = true;
          }
        }
        // (bit 9, diagnostics: debug bit 0x80000 = no window at all, i.e. the fixed cost of the path; wrong results)
        const int epi_s = epi | (ba.stride == 2 ? 0x100 : 0) | (EDT_DIAG_BITS(dbg, 0x80000) ? 0x200 : 0) | (compact ? 0x400 : 0) |
                          (ba.x32 == 2 ? 0x800 : 0);
        if (ba.x32) brute_tile<CW, BB, true>(tile, alive, rsp, lohi, bscan, n, NB, cols_left, band2, col2, w, epi_s, dst0, dstep);
        else brute_tile<CW, BB, false>(tile, alive, rsp, lohi, bscan, n, NB, cols_left, band2, col2, w, epi_s, dst0, dstep);
        return;
      }
    }
  }

  // ---- phase 1 / 2 / 3 (wave-local): hull_tile, a function of its own (see there) ---------------------
  hull_tile<CW, BB>(tile, alive, rsp, L.colc, L.band, n, w, L.nzw, L.rsw, L.lo_in, L.hi_out, fl0, need, epi, dbg, lane);
  __syncthreads();
  typedef float v4f __attribute__((ext_vector_type(4)));
  if constexpr (SC) {
    // Z-sharded path: the rows leave for the slab records of their destination (one look-up per
    // band; a band never straddles two destinations).  Same granules as the in-place stores.
    if (IO::kGran == 4 && aligned16) {
      for (int i = wave; i < IO::count(NBP, 4); i += W) {
        const int row = io_row<CW, 4>(i, lane), gc = io_gcol<CW, 4>(i, lane);
        if (row < n && gc < cols_left) {
          const int b = row >> 5;
          float *dst = scatter->rows[b] + o * scatter->ostride[b] + (int64_t)(row & 31) * st + x0 + gc;
          *reinterpret_cast<v4f *>(dst) = *reinterpret_cast<const v4f *>(tile + io_lds_word<CW, 4>(i, lane));
        }
      }
    } else {
      for (int i = wave; i < IO::count(NBP, 1); i += W) {
        const int row = io_row<CW, 1>(i, lane), gc = io_gcol<CW, 1>(i, lane);
        if (row < n && gc < cols_left) {
          const int b = row >> 5;
          scatter->rows[b][o * scatter->ostride[b] + (int64_t)(row & 31) * st + x0 + gc] =
              tile[io_lds_word<CW, 1>(i, lane)];
        }
      }
    }
  } else if (ba.compact != nullptr) {
    // only the even rows are read again, from a compact array (ColumnOut): float by float, the destination
    // rows need not be 16-byte aligned
    float *cdst = ba.compact + x0 + o * ba.c_outer;
    if (IO::kGran == 4 && aligned16 && ba.c_al) {
      for (int i = wave; i < IO::count(NBP, 4); i += W) {
        const int row = io_row<CW, 4>(i, lane), gc = io_gcol<CW, 4>(i, lane);
        if (row < n && !(row & 1) && gc < cols_left)
          *reinterpret_cast<v4f *>(cdst + (int64_t)(row >> 1) * ba.c_row2 + gc) = *reinterpret_cast<const v4f *>(tile + io_lds_word<CW, 4>(i, lane));
      }
    } else
    for (int i = wave; i < IO::count(NBP, 1); i += W) {
      const int row = io_row<CW, 1>(i, lane), gc = io_gcol<CW, 1>(i, lane);
      if (row < n && !(row & 1) && gc < cols_left) cdst[(int64_t)(row >> 1) * ba.c_row2 + gc] = tile[io_lds_word<CW, 1>(i, lane)];
    }
  } else if (IO::kGran == 4 && aligned16) {
    for (int i = wave; i < IO::count(NBP, 4); i += W) {
      const int row = io_row<CW, 4>(i, lane), gc = io_gcol<CW, 4>(i, lane);
      if (row < n && gc < cols_left) {
        const v4f v = *reinterpret_cast<const v4f *>(tile + io_lds_word<CW, 4>(i, lane));
        // (kEpiStream: the call's results -- streamed, edt_common.h; wave-uniform)
        if (epi & kEpiStream) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(Ftile + (int64_t)row * st + gc));
        else EDT_TILE_STORE(reinterpret_cast<v4f *>(Ftile + (int64_t)row * st + gc), v);
      }
    }
  } else {
    for (int i = wave; i < IO::count(NBP, 1); i += W) {
      const int row = io_row<CW, 1>(i, lane), gc = io_gcol<CW, 1>(i, lane);
      if (row < n && gc < cols_left) Ftile[(int64_t)row * st + gc] = tile[io_lds_word<CW, 1>(i, lane)];
    }
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
template <int CW, bool BB, bool XF, bool SC>
static int launch_wave_cbx_sc(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w,
                           int epi, const XFuse &xf, hipStream_t stream, const BandScatter *scatter,
                           bool scatter_aligned, const ColumnOut &out_stride, const TileList &list) {
  constexpr int NBP = 64 / CW;
  using TG = edt_lane::TileGeom<CW>;
  constexpr int TC = TG::kCols;
  const size_t lds = (size_t)(NBP + 2) * TG::kBandFloats * sizeof(float) + 4 * (size_t)NBP * TG::kBandWords * sizeof(uint32_t);
  // the windowed path (edt_colwave_lane.h: brute_band): tiles whose largest field value is at most c_T
  BruteArgs ba;
  ba.limit_bits = 0u;
  ba.x32 = 0;
  ba.force = 0;
  ba.stride = (out_stride.stride == 2 && !EDT_DIAG_BITS(debug_mode(), 0x40000)) ? 2 : 1;  // (debug bit 0x40000: evaluate every row)
  ba.compact = (out_stride.stride == 2 && !SC) ? out_stride.compact : nullptr;
  ba.c_outer = out_stride.outer;
  ba.c_row2 = out_stride.row2;
  ba.c_al = (reinterpret_cast<uintptr_t>(out_stride.compact) % 16) == 0 && (out_stride.outer % 4) == 0 && (out_stride.row2 % 4) == 0;
  if (ba.compact != nullptr) ba.stride = 2;  // (a compact destination has room for the even rows only)
  if (!(debug_mode() & 0x2000) && w * w >= 1.17549435e-38f && (double)w * (double)w < 1.0e30) {
    const bool force = (debug_mode() & 0x4000) != 0;
    // The window limit (edt_colwave.hip: window_limit).  fp32 candidates need c_d exact in
    // fp32 up to the limit (w2 = 900: d <= 136): where exactness ends between 64 rows and the limit, the limit
    // is lowered to it (fp32 candidates are ~25 % cheaper than fp64 ones; the tiles in between go to the hulls).
    int T = force ? (int)g.n : window_limit();
    const int exact = edt_lane::brute_exact_prefix(w, T);
    int mode = exact >= T ? 1 : 0;
    // Voxel sizes whose c_d are not all exact: fp32 fma candidates where the caller vouches for a lower bound of the
    // field (g.fmin) and the fp64 sums of the reference are exact on tiles up to c_Tf (brute_f32e_prefix) -- not for a
    // forced tile, whose values are not bounded by c_T.  Failing that, the exact prefix where it is worth a window.
    if (mode == 0 && !force) {
      const int Tf = (g.fmin > 0.0f && !(debug_mode() & 0x2000000)) ? edt_lane::brute_f32e_prefix(w, g.fmin, T) : 0;
      if (Tf >= 64 && Tf > exact) { T = Tf; mode = 2; }
      else if (exact >= 64) { T = exact; mode = 1; }
    }
    // (fp64 candidates walk their far rows one by one: beyond ~190 rows the hull path is the better form for them)
    if (mode == 0 && !force && T > 192) T = 192;
    if (debug_mode() & 0x8000) mode = 0;  // diagnostics: fp64 candidates
    ba.x32 = mode;
    const double cT = (double)(w * w) * (double)T * (double)T;
    float lim = cT < 3.0e38 ? (float)cT : 3.0e38f;
    if ((double)lim > cT) lim = nextafterf(lim, 0.0f);
    uint32_t bits;
    memcpy(&bits, &lim, 4);
    ba.limit_bits = force ? 0x7f800000u : bits;  // (forced: every tile, whatever it holds)
    ba.force = force ? 1 : 0;
    if (T < 1) ba.limit_bits = 0u;
  }
  ba.list_count = list.count;
  ba.list_ids = list.ids;
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_column_pass_wave<CW, BB, XF, SC>)));
  const int64_t tiles_x = ceil_div(g.sx, TC);
  int64_t tiles = tiles_x * g.nouter;
  if (tiles <= 0) return EDT_OK;
  if (!(debug_mode() & 0x800)) tiles = tiles_x * (ceil_div(g.nouter, 8) * 8);  // XCD-aware order
  // 16-byte granules need 16-byte aligned rows; otherwise the tile moves float by float
  const int aligned16 = (g.sx % 4) == 0 && (g.stride % 4) == 0 && (g.outer_stride % 4) == 0 &&
                        (reinterpret_cast<uintptr_t>(F) % 16) == 0 && (scatter == nullptr || scatter_aligned) &&
                        (!XF || reinterpret_cast<uintptr_t>(xf.codes) % 8 == 0);
  if (tiles > 0x7FFFFFFF) { set_error("too many tiles"); return EDT_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((k_column_pass_wave<CW, BB, XF, SC>), dim3((unsigned)tiles), dim3(64 * TC / CW), lds, stream,
                     F, nz, rs, g, w, (int)tiles_x, epi & (3 | kEpiStream), debug_mode(), aligned16, xf, scatter, ba);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

template <int CW, bool BB, bool XF>
static int launch_wave_cbx(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w,
                           int epi, const XFuse &xf, hipStream_t stream, const BandScatter *scatter,
                           bool scatter_aligned, const ColumnOut &out_stride, const TileList &list) {
  // the scattering epilogue (Z-sharded path) is a compile-time variant
  if (scatter != nullptr)
    return launch_wave_cbx_sc<CW, BB, XF, true>(F, nz, rs, g, w, epi, xf, stream, scatter, scatter_aligned, out_stride, list);
  return launch_wave_cbx_sc<CW, BB, XF, false>(F, nz, rs, g, w, epi, xf, stream, nullptr, false, out_stride, list);
}

template <int CW>
int launch_wave_c(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g, float w,
                         int bb, int epi, const XFuse *xf, hipStream_t stream, const BandScatter *scatter,
                         bool sc_al, const ColumnOut &out_stride, const TileList &list) {
  // the border rule and the index form of pass 1 are compile-time variants, the epilogue a run-time one
  const XFuse none = {nullptr, 0, 0.0f, 0};
  if (xf)
    return bb ? launch_wave_cbx<CW, true, true>(F, nz, rs, g, w, epi, *xf, stream, scatter, sc_al, out_stride, list)
              : launch_wave_cbx<CW, false, true>(F, nz, rs, g, w, epi, *xf, stream, scatter, sc_al, out_stride, list);
  return bb ? launch_wave_cbx<CW, true, false>(F, nz, rs, g, w, epi, none, stream, scatter, sc_al, out_stride, list)
            : launch_wave_cbx<CW, false, false>(F, nz, rs, g, w, epi, none, stream, scatter, sc_al, out_stride, list);
}

}  // namespace edt_amd
