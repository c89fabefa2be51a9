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
// All fp64 arithmetic is evaluated without contraction, so results are bit-identical to
// the CPU reference (see edt_kernels.h: hull_num).
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

namespace edt_amd {

namespace {

// b (height Fb) lies on or above the chord from a to c  <=>  s(b,c) <= s(a,b): parabola b is
// never the strict minimum and can be dropped (reference pop rule, src/edt.hpp:210, :287).
__device__ __forceinline__ bool dominated(int a, double Fa, int b, double Fb, int c, double Fc,
                                          double w2) {
  const double lhs = hull_num(Fb, Fc, b, c, w2) * (double)(b - a);
  const double rhs = hull_num(Fa, Fb, a, b, w2) * (double)(c - b);
  return lhs <= rhs;
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

}  // namespace

template <int C>
__global__ void __launch_bounds__(1024)
k_column_pass_tiled(float *__restrict__ F, const uint32_t *__restrict__ nzbits,
                    const uint32_t *__restrict__ rsbits, AxisGeom g, float w, int bb, int epi,
                    int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NB = (int)blockDim.y;
  const int n = (int)g.n;
  float *tile = reinterpret_cast<float *>(smem);                  // [NB*32][C]
  uint32_t *alive = reinterpret_cast<uint32_t *>(tile + NB * 32 * C);  // [NB][C]
  uint32_t *rsp = alive + NB * C;                                  // [NB][C]
  uint32_t *nzp = rsp + NB * C;                                    // [NB][C]

  const int c = (int)threadIdx.x, b = (int)threadIdx.y;
  const int64_t xt = blockIdx.x % tiles_x, o = blockIdx.x / tiles_x;
  const int64_t x = xt * C + c;
  const bool active = x < g.sx;
  const int64_t base = x + o * g.outer_stride;
  const int64_t st = g.stride;
  const double w2 = (double)(w * w);  // fp32 product widened (src/edt.hpp:181, :258)

  float *tcol = tile + c;             // element of row r: tcol[r * C]
  uint32_t *acol = alive + c;         // word of band k: acol[k * C]
  const uint32_t *rcol = rsp + c;
  const int row0 = b * 32;

  // ---- phase 0: load ------------------------------------------------------------------
  uint32_t nzword = 0, rsword = 0;
  if (active) {
    const int64_t widx = (o * g.nbands + b) * g.sx + x;
    nzword = nzbits[widx];
    rsword = rsbits[widx];
  }
  {
    float v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      v[r] = 0.0f;
      if ((nzword >> r) & 1u) v[r] = F[base + (int64_t)(row0 + r) * st];
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) tcol[(row0 + r) * C] = v[r];
  }
  rsp[b * C + c] = rsword;
  nzp[b * C + c] = nzword;

  // ---- phase 1: hull of this band (monotone chain; the stack is the set bits of `aw`) ----
  {
    uint32_t aw = 0;
    uint32_t seg = 0xFFFFFFFFu;  // bits of the current run segment (from its first row upward)
    int ia = -1, ib = -1;
    double Fa = 0.0, Fb = 0.0;
    uint32_t todo = nzword;
    while (todo) {
      const int r = __builtin_ctz(todo);
      todo &= todo - 1;
      // a run starts at r, or a background gap lies between the previous vertex and r
      if (((rsword >> r) & 1u) || ib != row0 + r - 1) {
        ia = ib = -1;
        seg = 0xFFFFFFFFu << r;
      }
      const int row = row0 + r;
      const double Fi = (double)tcol[row * C];
      while (ia >= 0 && dominated(ia, Fa, ib, Fb, row, Fi, w2)) {
        aw &= ~(1u << (ib & 31));
        ib = ia;
        Fb = Fa;
        const uint32_t below = aw & seg & ((1u << (ib & 31)) - 1u);
        if (below) {
          ia = row0 + 31 - __builtin_clz(below);
          Fa = (double)tcol[ia * C];
        } else {
          ia = -1;
        }
      }
      aw |= 1u << r;
      ia = ib; Fa = Fb;
      ib = row; Fb = Fi;
    }
    acol[b * C] = aw;
  }
  __syncthreads();

  // ---- phase 2: merge hulls across band-group boundaries ------------------------------------
  for (int half = 1; half < NB; half <<= 1) {
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

        int u = prev_set(acol, C, R, Llo);  // == R-1 (last vertex of the left hull)
        int v = R;                           // first vertex of the right hull
        double Fu = (double)tcol[u * C], Fv = (double)tcol[v * C];
        int up = prev_set(acol, C, u, Llo);
        int vn = next_set(acol, C, v, Rhi);
        double Fup = up >= 0 ? (double)tcol[up * C] : 0.0;
        double Fvn = vn >= 0 ? (double)tcol[vn * C] : 0.0;
        while (true) {
          bool moved = false;
          if (up >= 0 && dominated(up, Fup, u, Fu, v, Fv, w2)) {
            acol[(u >> 5) * C] &= ~(1u << (u & 31));
            u = up; Fu = Fup;
            up = prev_set(acol, C, u, Llo);
            Fup = up >= 0 ? (double)tcol[up * C] : 0.0;
            moved = true;
          }
          if (vn >= 0 && dominated(u, Fu, v, Fv, vn, Fvn, w2)) {
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
  if (!active) return;
  int r = 0;
  while (r < 32) {
    const uint32_t rest = nzword >> r;
    if (rest == 0) break;
    r += __builtin_ctz(rest);  // next foreground row of this band
    const int p0 = row0 + r;
    const int run_lo = ((rsword >> r) & 1u) ? p0 : prev_set(rcol, C, p0, 0);
    const int nxt = next_set(rcol, C, p0, n - 1);
    const int run_hi = nxt < 0 ? n - 1 : nxt - 1;
    const int seg_end = run_hi < row0 + 31 ? run_hi : row0 + 31;
    const bool left = bb || run_lo > 0;
    const bool right = bb || run_hi < n - 1;

    // hull vertex owning p0: start from the last vertex at or before p0, walk down the
    // (unimodal) values
    int j = prev_set(acol, C, p0 + 1, run_lo);
    double Fj = (double)tcol[j * C];
    {
      double vj = w2 * sqd(p0 - j) + Fj;
      while (true) {
        const int jp = prev_set(acol, C, j, run_lo);
        if (jp < 0) break;
        const double Fjp = (double)tcol[jp * C];
        const double vp = w2 * sqd(p0 - jp) + Fjp;
        if (!(vp < vj)) break;
        j = jp; Fj = Fjp; vj = vp;
      }
    }
    int jn = next_set(acol, C, j, run_hi);
    double Fjn = jn >= 0 ? (double)tcol[jn * C] : 0.0;

    for (int p = p0; p <= seg_end; ++p) {
      double best = w2 * sqd(p - j) + Fj;
      while (jn >= 0) {
        const double cand = w2 * sqd(p - jn) + Fjn;
        if (!(cand < best)) break;
        best = cand;
        j = jn; Fj = Fjn;
        jn = next_set(acol, C, j, run_hi);
        Fjn = jn >= 0 ? (double)tcol[jn * C] : 0.0;
      }
      float m = (float)best;
      if (left) m = fminf((float)(w2 * sqd(p - run_lo + 1)), m);
      if (right) m = fminf((float)(w2 * sqd(run_hi - p + 1)), m);
      F[base + (int64_t)p * st] = finish(m, epi);
    }
    r = seg_end - row0 + 1;
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
bool column_pass_tiled_supported(const AxisGeom &g) {
  return g.nbands >= 1 && g.nbands * 8 <= 1024;  // C = 8 is the narrowest tile
}

template <int C>
static int launch_tiled_c(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                          float w, int bb, int epi, hipStream_t stream) {
  const int NB = (int)g.nbands;
  const size_t lds = (size_t)NB * C * (32 * sizeof(float) + 3 * sizeof(uint32_t));
  static bool attr_done = false;  // per C instantiation
  if (!attr_done) {
    EDT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_column_pass_tiled<C>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const int64_t tiles_x = ceil_div(g.sx, C);
  const int64_t tiles = tiles_x * g.nouter;
  if (tiles <= 0) return EDT_OK;
  if (tiles > 0x7FFFFFFF) { set_error("too many tiles"); return EDT_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(k_column_pass_tiled<C>, dim3((unsigned)tiles), dim3(C, NB), lds, stream, F, nz,
                     rs, g, w, bb, epi, (int)tiles_x);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_column_pass_tiled(float *F, const uint32_t *nz, const uint32_t *rs, const AxisGeom &g,
                             float w, int bb, int epi, hipStream_t stream) {
  const int64_t NB = g.nbands;
  if (NB * 32 <= 1024) return launch_tiled_c<32>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 16 <= 1024) return launch_tiled_c<16>(F, nz, rs, g, w, bb, epi, stream);
  if (NB * 8 <= 1024) return launch_tiled_c<8>(F, nz, rs, g, w, bb, epi, stream);
  set_error("axis too long for the tiled column pass");
  return EDT_ERR_UNSUPPORTED;
}

}  // namespace edt_amd
