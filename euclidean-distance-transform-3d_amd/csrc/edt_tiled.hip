// edt_tiled.hip -- LDS-tiled column pass (passes 2 and 3) for CDNA4 / gfx950.
//
// Work decomposition (one workgroup = one tile):
//   tile    = C adjacent columns (consecutive x, so every global access is a C*4-byte
//             contiguous segment) x the WHOLE scan axis (n rows), staged once in LDS;
//   thread  = (column c, band b): a band is 32 consecutive rows = one bit-word of the run
//             masks.  A 64-lane wavefront therefore covers 64/C bands x C columns.
//   LDS     = fp32 tile [n][C]  +  three bit-planes [n/32][C] (alive hull vertices, run
//             starts, foreground).  With C = 32 every column owns one LDS bank, so the
//             data-dependent row gathers of the hull walk are conflict-free by construction.
//
// Algorithm (exact lower envelope, work-efficient, no divisions, no per-vertex storage):
//   the lower envelope of the parabolas  w2*(p-j)^2 + F[j]  over one label run is the lower
//   convex hull of the points (j, F[j] + w2*j^2).  A hull is just a SUBSET of the rows, so it
//   is stored as one bit per row (the `alive` plane) -- no vertex or breakpoint arrays.
//     phase 0  load:   every thread loads its 32 rows (coalesced) into the LDS tile;
//     phase 1  local:  every thread builds the hull of its own band with a monotone-chain
//                      scan (stack = the set bits of its word, top found with clz);
//     phase 2  merge:  log2(bands) rounds; in each round the thread at a group boundary
//                      joins the hull of the left group with the hull of the right group by
//                      walking the common tangent ("bridge") and clearing the bits in between;
//                      only needed where one label run crosses the boundary;
//     phase 3  eval:   every thread locates the hull vertex owning its first row, then sweeps
//                      its 32 rows forward, evaluating the reference's own expression
//                      fl32(w2*(p-j)^2 + F[j]) (src/edt.hpp:230, :307), the border parabolas
//                      (src/edt.hpp:233-242, :310-311) and the fused toinfinite/sqrt epilogue,
//                      and stores in place.
// Background voxels are neither loaded nor stored (they hold 0 since pass 1 and stay 0).
//
// Arithmetic: rows are < 2^15, so differences and their squares are formed in int32 (one
// full-rate 24-bit multiply), converted once to fp64, and combined with separate fp64
// multiply/add (no contraction) -- value-identical to the reference's
// `w2 * sq(i - v[k]) + ff[v[k]]` and `ff[i] - ff[v[k]] + factor1 * factor2`
// (every product involved is exact in fp64: w2 carries 24 significant bits).
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

namespace edt_amd {

namespace {

// exact (double)(d*d) for |d| < 32768 (24-bit operands, the product fits 31 bits)
__device__ __forceinline__ double sq_i(int d) { return (double)__mul24(d, d); }

// value of parabola j (height Fj) at row p -- the reference's output expression
__device__ __forceinline__ double para(int p, int j, double Fj, double w2) {
  return w2 * sq_i(p - j) + Fj;
}

// numerator of the crossing abscissa of parabolas p < q:  (Fq - Fp) + w2*(q-p)*(q+p)
// (== hull_num of edt_kernels.h: (q-p)*w2*(q+p) is exact in either association)
__device__ __forceinline__ double edge_num(int p, double Fp, int q, double Fq, double w2) {
  return (Fq - Fp) + w2 * (double)__mul24(q - p, q + p);
}

// Highest set bit p with lo <= p < from in a bit-plane column (`words` already points at the
// column; consecutive words are `pitch` apart).  -1 if none.
__device__ __forceinline__ int prev_set(const uint32_t *words, int pitch, int from, int lo) {
  if (from <= lo) return -1;
  int wi = (from - 1) >> 5;
  const int wlo = lo >> 5;
  uint32_t m = words[wi * pitch] & (0xFFFFFFFFu >> (31 - ((from - 1) & 31)));
  while (true) {
    if (wi == wlo) m &= 0xFFFFFFFFu << (lo & 31);
    if (m) return wi * 32 + 31 - __builtin_clz(m);
    if (wi == wlo) return -1;
    --wi;
    m = words[wi * pitch];
  }
}

// Lowest set bit p with after < p <= hi.  -1 if none.
__device__ __forceinline__ int next_set(const uint32_t *words, int pitch, int after, int hi) {
  if (after >= hi) return -1;
  int wi = (after + 1) >> 5;
  const int whi = hi >> 5;
  uint32_t m = words[wi * pitch] & (0xFFFFFFFFu << ((after + 1) & 31));
  while (true) {
    if (wi == whi) m &= 0xFFFFFFFFu >> (31 - (hi & 31));
    if (m) return wi * 32 + __builtin_ctz(m);
    if (wi == whi) return -1;
    ++wi;
    m = words[wi * pitch];
  }
}

// Next hull vertex after j inside the run ending at run_hi, or -1.  `aw` is the calling
// thread's own (post-merge) alive word for rows [row0, row0+32): the common case needs no LDS.
__device__ __forceinline__ int next_vertex(int j, int row0, uint32_t aw, int run_hi,
                                           const uint32_t *acol, int C) {
  const int rj = j - row0;
  int q = -1;
  if (rj >= 0 && rj < 32) {
    const uint32_t m = (rj < 31) ? (aw & (0xFFFFFFFEu << rj)) : 0u;
    if (m) q = row0 + __builtin_ctz(m);
    else if (run_hi > row0 + 31) q = next_set(acol, C, row0 + 31, run_hi);
  } else {
    q = next_set(acol, C, j, run_hi);
  }
  return q > run_hi ? -1 : q;
}

}  // namespace

template <int C, int EPI, bool BB>
__global__ void __launch_bounds__(1024)
k_column_pass_tiled(float *__restrict__ F, const uint32_t *__restrict__ nzbits,
                    const uint32_t *__restrict__ rsbits, AxisGeom g, float w, int tiles_x,
                    int debug_mode) {
  constexpr int epi = EPI;
  constexpr bool bb = BB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NB = (int)blockDim.y;
  const int n = (int)g.n;
  float *tile = reinterpret_cast<float *>(smem);                       // [NB*32][C]
  uint32_t *alive = reinterpret_cast<uint32_t *>(tile + NB * 32 * C);  // [NB][C]
  uint32_t *rsp = alive + NB * C;                                      // [NB][C]

  const int c = (int)threadIdx.x, b = (int)threadIdx.y;
  const int64_t xt = blockIdx.x % tiles_x, o = blockIdx.x / tiles_x;
  const int64_t x = xt * C + c;
  const bool active = x < g.sx;
  const int64_t st = g.stride;
  float *Fcol = F + x + o * g.outer_stride;  // global column
  const double w2 = (double)(w * w);         // fp32 product widened (src/edt.hpp:181, :258)
  float *tcol = tile + c;                    // row r of this column: tcol[r * C]
  uint32_t *acol = alive + c;                // word of band k: acol[k * C]
  const uint32_t *rcol = rsp + c;
  const int row0 = b * 32;

  {
    // ---- phase 0: load ------------------------------------------------------------------
    uint32_t nzword = 0, rsword = 0;
    if (active) {
      const int64_t widx = (o * g.nbands + b) * g.sx + x;
      nzword = nzbits[widx];
      rsword = rsbits[widx];
    }
    {
      float v[32];
      const float *src = Fcol + (int64_t)row0 * st;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        v[r] = 0.0f;
        if ((nzword >> r) & 1u) v[r] = src[(int64_t)r * st];
      }
#pragma unroll
      for (int r = 0; r < 32; ++r) tcol[(row0 + r) * C] = v[r];
    }
    rsp[b * C + c] = rsword;

    if (EDT_DIAG_BITS(debug_mode, 1)) {
      // diagnostics: memory-only variant (tile -> LDS -> store back), the access-pattern floor
      // that the roofline analysis in DESIGN.md compares the full kernel against
      __syncthreads();
      float *dstp = Fcol + (int64_t)row0 * st;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        if ((nzword >> r) & 1u) dstp[(int64_t)r * st] = tcol[(row0 + r) * C];
      }
      return;
    }

  // ---- phase 1: hull of this band (monotone chain; the stack is the set bits of `aw`) ----
  // Written in "predicated, wave-uniform loop" style: every lane steps through the same row
  // index, data-dependent repetition (pops) is a loop on __any(), and per-lane decisions are
  // selects.  This keeps the scalar unit out of the way (no exec-mask bookkeeping per branch).
  uint32_t aw = 0;
  if (EDT_DIAG_BITS(debug_mode, 2)) {  // diagnostics: pretend every foreground row is a hull vertex
    aw = nzword;
    acol[b * C] = aw;
  } else {
    uint32_t seg = 0xFFFFFFFFu;  // rows of the current run segment (from its first row upward)
    int ia = -1, ib = -1;        // second / top vertex of the stack (ia < 0: fewer than two)
    double Fa = 0.0, Fb = 0.0;
    double nab = 0.0;            // edge_num(ia, ib) of the top edge
    double dab = 1.0;            // (double)(ib - ia)
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
      const bool nz = (nzword >> r) & 1u;
      if (!__any(nz)) continue;
      const int row = row0 + r;
      const double Fi = (double)tcol[row * C];
      if ((rsword >> r) & 1u) {  // a run starts here: fresh stack
        ia = -1; ib = -1;
        seg = 0xFFFFFFFFu << r;
      }
      double nbi = edge_num(ib, Fb, row, Fi, w2);
      // pop while the top vertex lies on or above the chord (second, new):
      //   s(ib,row) <= s(ia,ib)  <=>  nbi * (ib-ia) <= nab * (row-ib)   (src/edt.hpp:210, :287)
      bool pop = nz && ia >= 0 && (nbi * dab <= nab * (double)(row - ib));
      while (__any(pop)) {
        if (pop) {
          aw &= ~(1u << (ib - row0));
          ib = ia;
          Fb = Fa;
          nbi = edge_num(ib, Fb, row, Fi, w2);
          const uint32_t below = aw & seg & ((1u << (ib - row0)) - 1u);
          const int top = row0 + 31 - __builtin_clz(below | 1u);
          ia = below ? top : -1;
          Fa = (double)tcol[(below ? top : row) * C];
          nab = edge_num(ia, Fa, ib, Fb, w2);
          dab = (double)(ib - ia);
        }
        pop = nz && ia >= 0 && (nbi * dab <= nab * (double)(row - ib));
      }
      if (nz) {
        aw |= 1u << r;
        ia = ib; Fa = Fb;
        nab = nbi;
        dab = (double)(row - ib);
        ib = row; Fb = Fi;
      }
    }
    acol[b * C] = aw;
  }
  __syncthreads();

  // ---- phase 2: merge hulls across band-group boundaries ------------------------------------
  for (int half = 1; half < NB; half <<= 1) {
    if (EDT_DIAG_BITS(debug_mode, 4)) break;  // diagnostics: no merges
    if (active && (b & (2 * half - 1)) == half) {
      const int R = row0;  // first row of the right group
      // a non-background run crosses the boundary iff row R is foreground and not a run start
      if (R < n && (nzword & 1u) && !(rsword & 1u)) {
        const int glo = (b - half) * 32;
        int ghi = (b + half) * 32;
        if (ghi > n) ghi = n;
        ghi -= 1;
        int Llo = prev_set(rcol, C, R, glo);
        if (Llo < 0) Llo = glo;
        const int nxt = next_set(rcol, C, R, ghi);
        const int Rhi = nxt < 0 ? ghi : nxt - 1;

        int u = R - 1;  // last vertex of the left hull (always alive)
        int v = R;      // first vertex of the right hull (always alive)
        double Fu = (double)tcol[u * C], Fv = (double)tcol[v * C];
        int up = prev_set(acol, C, u, Llo);
        int vn = next_set(acol, C, v, Rhi);
        double Fup = up >= 0 ? (double)tcol[up * C] : 0.0;
        double Fvn = vn >= 0 ? (double)tcol[vn * C] : 0.0;
        while (true) {
          bool moved = false;
          const double nuv = edge_num(u, Fu, v, Fv, w2);
          if (up >= 0 &&
              nuv * (double)(u - up) <= edge_num(up, Fup, u, Fu, w2) * (double)(v - u)) {
            acol[(u >> 5) * C] &= ~(1u << (u & 31));
            u = up; Fu = Fup;
            up = prev_set(acol, C, u, Llo);
            Fup = up >= 0 ? (double)tcol[up * C] : 0.0;
            moved = true;
          } else if (vn >= 0 &&
                     edge_num(v, Fv, vn, Fvn, w2) * (double)(v - u) <= nuv * (double)(vn - v)) {
            acol[(v >> 5) * C] &= ~(1u << (v & 31));
            v = vn; Fv = Fvn;
            vn = next_set(acol, C, v, Rhi);
            Fvn = vn >= 0 ? (double)tcol[vn * C] : 0.0;
            moved = true;
          }
          if (!moved) break;
        }
      }
    }
    __syncthreads();
  }

  // ---- phase 3: evaluate the envelope on this band's rows, in place -------------------------
  aw = acol[b * C];  // this band's vertices after the merges
  if (!active) { nzword = 0; }
  if (EDT_DIAG_BITS(debug_mode, 8)) {  // diagnostics: skip the evaluation, store the tile back
    float *dstp = Fcol + (int64_t)row0 * st;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if ((nzword >> r) & 1u) dstp[(int64_t)r * st] = tcol[(row0 + r) * C];
    }
    nzword = 0;  // nothing left to do for this tile
  }

  // run boundaries outside the band (needed only when a run crosses the band's ends)
  int lo_carry = row0, hi_carry = row0 + 31;
  if ((nzword & 1u) && !(rsword & 1u)) lo_carry = prev_set(rcol, C, row0, 0);
  if (row0 + 31 < n - 1 && (nzword >> 31)) {
    const int nx = next_set(rcol, C, row0 + 31, n - 1);
    hi_carry = nx < 0 ? n - 1 : nx - 1;
  }
  if (hi_carry > n - 1) hi_carry = n - 1;

  int j = row0, jn = -1, run_lo = row0, run_hi = row0;
  double Fj = 0.0, Fjn = 0.0;
  if ((nzword & 1u) && !(rsword & 1u)) {
    // The band begins inside a run that started in an earlier band: find the hull vertex that
    // owns row0 -- start from the last vertex at or before it and walk down the (unimodal)
    // values towards earlier vertices.
    run_lo = lo_carry;
    const uint32_t above = rsword & 0xFFFFFFFEu;
    run_hi = above ? row0 + __builtin_ctz(above) - 1 : hi_carry;
    j = prev_set(acol, C, row0 + 1, run_lo);
    Fj = (double)tcol[j * C];
    double vj = para(row0, j, Fj, w2);
    while (true) {
      const int jp = prev_set(acol, C, j, run_lo);
      if (jp < 0) break;
      const double Fjp = (double)tcol[jp * C];
      const double vp = para(row0, jp, Fjp, w2);
      if (!(vp < vj)) break;
      j = jp; Fj = Fjp; vj = vp;
    }
    jn = next_vertex(j, row0, aw, run_hi, acol, C);
    Fjn = (double)tcol[(jn >= 0 ? jn : row0) * C];
  }

  // Row sweep, wave-uniform control flow only: per-lane decisions are selects, repetition is a
  // loop on __any(), and the one rare case (the next vertex lies in another band) sits behind
  // a single __any() test.  `need` marks lanes whose next-vertex register must be (re)loaded.
  bool need = false;
  float *dst = Fcol + (int64_t)row0 * st;
#pragma unroll 1
  for (int r = 0; r < 32; ++r, dst += st) {
    const bool nz = (nzword >> r) & 1u;
    if (!__any(nz)) continue;
    const int p = row0 + r;
    const bool start = nz && ((rsword >> r) & 1u);  // a run starts at p: first vertex is p
    if (__any(start)) {
      const double Fp = (double)tcol[p * C];
      const uint32_t above = rsword & (0xFFFFFFFEu << r);
      const int hi_here = above ? row0 + __builtin_ctz(above | 0x80000000u) - 1 : hi_carry;
      run_lo = start ? p : run_lo;
      run_hi = start ? hi_here : run_hi;
      j = start ? p : j;
      Fj = start ? Fp : Fj;
      need = need || start;
    }
    double best = para(p, j, Fj, w2);
    while (true) {
      if (__any(need)) {
        // next hull vertex after j: a bit scan of this thread's own alive word ...
        const unsigned rj = (unsigned)(j - row0);
        const uint32_t m = rj < 32u ? (aw & (0xFFFFFFFEu << (rj & 31u))) : 0u;
        const int q = row0 + __builtin_ctz(m | 0x80000000u);
        const bool found = m != 0u && q <= run_hi;
        // ... unless the run continues past this band (or j sits in another band)
        const bool slow = need && !found && (rj < 32u ? (m == 0u && run_hi > row0 + 31) : true);
        int jq = found ? q : -1;
        if (__any(slow)) {
          if (slow) {
            jq = next_set(acol, C, rj < 32u ? row0 + 31 : j, run_hi);
          }
        }
        jn = need ? jq : jn;
        const double Fq = (double)tcol[(jq >= 0 ? jq : p) * C];
        Fjn = need ? Fq : Fjn;
      }
      const double cand = para(p, jn, Fjn, w2);
      const bool adv = nz && jn >= 0 && cand < best;
      if (!__any(adv)) break;
      best = adv ? cand : best;
      j = adv ? jn : j;
      Fj = adv ? Fjn : Fj;
      need = adv;
    }
    need = false;
    // border parabolas of height 0 at run_lo-1 and run_hi+1.  fp32 rounding is monotone, so
    // one narrowing of the fp64 minimum equals the reference's separate narrowings
    // (src/edt.hpp:233-242, :310-311); the nearer border dominates the farther one.
    const bool left = bb || run_lo > 0;
    const bool right = bb || run_hi < n - 1;
    const int dl = left ? p - run_lo + 1 : 0x7FFF;
    const int dr = right ? run_hi - p + 1 : 0x7FFF;
    const int dm = dl < dr ? dl : dr;
    const double border = w2 * sq_i(dm);
    best = (left || right) ? fmin(best, border) : best;
    if (nz) *dst = finish((float)best, epi);
  }
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
bool column_pass_tiled_supported(const AxisGeom &g) {
  // one thread per (column, band), at most 1024 threads and 160 KiB of LDS per tile: 32 columns up to 1024 rows ...
  // 8 columns up to 4096 rows, then 4 / 2 / 1 columns up to 8192 / 16384 / 32736 rows (narrow tiles: 16- / 8- / 4-byte
  // row pieces, poorly coalesced -- still every band of every column has its own thread, where the size-agnostic
  // kernel gives a whole column to one).  Row distances and their products stay below 2^31 (sq_i, edge_num).
  return g.nbands >= 1 && g.nbands <= 1023;
}

template <int C, int EPI, bool BB>
static int launch_tiled_ceb(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                            float w, hipStream_t stream) {
  const int NB = (int)g.nbands;
  const size_t lds = (size_t)NB * C * (32 * sizeof(float) + 2 * sizeof(uint32_t));
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_column_pass_tiled<C, EPI, BB>)));
  const int64_t tiles_x = ceil_div(g.sx, C);
  const int64_t tiles = tiles_x * g.nouter;
  if (tiles <= 0) return EDT_OK;
  if (tiles > 0x7FFFFFFF) { set_error("too many tiles"); return EDT_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((k_column_pass_tiled<C, EPI, BB>), dim3((unsigned)tiles), dim3(C, NB), lds, stream,
                     F, nz, rs, g, w, (int)tiles_x, debug_mode());
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

template <int C>
static int launch_tiled_c(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                          float w, int bb, int epi, hipStream_t stream) {
  // the fused epilogue and the border rule are compile-time variants of the kernel
  switch ((epi & 3) * 2 + (bb ? 1 : 0)) {
    case 0: return launch_tiled_ceb<C, 0, false>(F, nz, rs, g, w, stream);
    case 1: return launch_tiled_ceb<C, 0, true>(F, nz, rs, g, w, stream);
    case 2: return launch_tiled_ceb<C, 1, false>(F, nz, rs, g, w, stream);
    case 3: return launch_tiled_ceb<C, 1, true>(F, nz, rs, g, w, stream);
    case 4: return launch_tiled_ceb<C, 2, false>(F, nz, rs, g, w, stream);
    case 5: return launch_tiled_ceb<C, 2, true>(F, nz, rs, g, w, stream);
    case 6: return launch_tiled_ceb<C, 3, false>(F, nz, rs, g, w, stream);
    default: return launch_tiled_ceb<C, 3, true>(F, nz, rs, g, w, stream);
  }
}

int launch_column_pass_tiled(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                             float w, int bb, int epi, hipStream_t stream) {
  const int64_t NB = g.nbands;
  if (NB * 32 <= 1024) return launch_tiled_c<32>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 16 <= 1024) return launch_tiled_c<16>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 8 <= 1024) return launch_tiled_c<8>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 4 <= 1024) return launch_tiled_c<4>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 2 <= 1024) return launch_tiled_c<2>(F, nz, rs, g, w, bb, epi, stream);
  if (NB < 1024) return launch_tiled_c<1>(F, nz, rs, g, w, bb, epi, stream);
  set_error("axis too long for the tiled column pass");
  return EDT_ERR_UNSUPPORTED;
}

}  // namespace edt_amd
