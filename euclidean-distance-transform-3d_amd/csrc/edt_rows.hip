// edt_rows.hip -- pass 1 (x axis) for CDNA4: one wavefront per group of 32 rows, plus the
// bit-plane transposer that feeds the z pass.
//
// Pass 1 is a label-aware 1-D distance along contiguous rows.  The reference walks each row
// twice with fp32 recurrences (src/edt.hpp:83-118); the result has the closed form
//     d(i) = min( L, R ),   L = T[i-s+1]  (if a boundary exists on the left,  else +inf)
//                           R = T[e-i+1]  (if a boundary exists on the right, else +inf)
//     T[0] = 0, T[k] = fl32(T[k-1] + w)        (the SAME sequential fp32 sums)
// for a voxel i inside the maximal run [s,e] of one non-zero label, and F = fl32(d*d).
// So the row only needs its run boundaries: lanes sit on consecutive x, a wave-wide ballot of
// "label differs from the left neighbour" gives a 64-bit start mask per 64-voxel chunk, and
// s / e follow from clz / ctz on that mask (plus a per-chunk carry for runs that span chunks).
//
// The same sweep emits, per voxel, the three bits the column passes need so that labels are
// read from HBM exactly ONCE by the whole pipeline:
//     nz : label != 0
//     ys : label differs from the voxel at y-1 (run start along y)
//     zs : label differs from the voxel at z-1 (run start along z)
// packed 32 consecutive y per word, layout [z][y/32][x] (coalesced across x for the y pass).
// k_bits_transpose_yz re-packs (nz, zs) into 32 consecutive z per word, layout [y][z/32][x],
// for the z pass.
#include "edt_common.h"
#include "edt_kernels.h"

#pragma clang fp contract(off)

namespace edt_amd {

namespace {
constexpr int kWavesPerBlock = 4;
constexpr int kMaxChunks = 32;  // rows up to 2048 voxels
}  // namespace

template <typename T, bool HAS_Z>
__global__ void __launch_bounds__(kWavesPerBlock * 64, 4)
k_row_pass_tiled(const T *__restrict__ labels, float *__restrict__ out,
                 uint32_t *__restrict__ nz_y, uint32_t *__restrict__ ys_y,
                 uint32_t *__restrict__ zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                 int to_finite, int NC, int64_t nby, int64_t ngroups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS carve: T table [sx+2] | per wave: st[32][NC] u64, fg[32][NC] u64, pre[32][NC], suf[32][NC]
  const int tbl = (int)((sx + 2 + 3) & ~3);
  float *Ttab = reinterpret_cast<float *>(smem);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably wave-uniform
  const int lane = (int)(threadIdx.x & 63);
  unsigned char *wbase = smem + (size_t)tbl * sizeof(float) + (size_t)wave * (32 * NC * 24);
  unsigned long long *st = reinterpret_cast<unsigned long long *>(wbase);          // [32][NC]
  unsigned long long *fg = st + 32 * NC;                                           // [32][NC]
  int *pre = reinterpret_cast<int *>(fg + 32 * NC);                                // [32][NC]
  int *suf = pre + 32 * NC;                                                        // [32][NC]

  // The reference's sequential fp32 sums of the voxel size (src/edt.hpp:97, :113).
  if (threadIdx.x == 0) {
    float acc = 0.0f;
    Ttab[0] = 0.0f;
    for (int k = 1; k <= (int)sx + 1; ++k) {
      acc = acc + w;
      Ttab[k] = acc;
    }
  }
  __syncthreads();

  const int64_t sxy = sx * sy;
  const unsigned long long le_mask = ~0ull >> (63 - lane);  // bits 0..lane

  for (int64_t g0 = (int64_t)blockIdx.x * kWavesPerBlock; g0 < ngroups;
       g0 += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t grp = g0 + wave;
    const bool live = grp < ngroups;
    const int64_t z = live ? grp / nby : 0;
    const int64_t yb = live ? grp % nby : 0;
    const int64_t y0 = yb * 32;
    const int nrows = live ? (int)((sy - y0) < 32 ? (sy - y0) : 32) : 0;

    // ---- phase A: labels -> chunk masks (LDS) and packed bit-words (global) ----------
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
      const int64_t x = (int64_t)c * 64 + lane;
      const bool inb = x < sx;
      // Every load below is unconditional (clamped to a voxel that exists): the left and lower
      // neighbours are plain shifted loads that hit L1/L2, not exchanged through lanes.
      const int64_t xs = inb ? x : sx - 1;
      const int64_t xl = xs > 0 ? xs - 1 : 0;
      const T *base = labels + (z * sy + y0) * sx;       // row y0 of this slice
      const T *lower = (HAS_Z && z > 0) ? base - sxy : base;
      uint32_t nzw = 0, ysw = 0, zsw = 0;
      T above = 0;  // label at (x, y-1, z)
      bool have_above = false;
      if (nrows > 0 && y0 > 0) {
        above = base[xs - sx];
        have_above = true;
      }
      constexpr int kBatch = 16;  // rows loaded back-to-back before any cross-lane work
#pragma unroll 1
      for (int r0 = 0; r0 < nrows; r0 += kBatch) {
        T labv[kBatch], leftv[kBatch], belv[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const int rr = r0 + k < nrows ? r0 + k : nrows - 1;
          const int64_t off = (int64_t)rr * sx;
          labv[k] = base[off + xs];
          leftv[k] = base[off + xl];
          belv[k] = HAS_Z ? lower[off + xs] : labv[k];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const int r = r0 + k;
          if (r < nrows) {  // wave-uniform
            const T lab = labv[k];
            const bool startx = inb && (x == 0 || lab != leftv[k]);
            const unsigned long long stm = __ballot(startx);
            const unsigned long long fgm = __ballot(inb && lab != 0);
            if (lane == 0) {
              st[r * NC + c] = stm;
              fg[r * NC + c] = fgm;
            }
            nzw |= (lab != 0 ? 1u : 0u) << r;
            ysw |= ((!have_above || lab != above) ? 1u : 0u) << r;
            if (HAS_Z) zsw |= ((z == 0 || lab != belv[k]) ? 1u : 0u) << r;
            above = lab;
            have_above = true;
          }
        }
      }
      if (inb && live) {
        const int64_t widx = (z * nby + yb) * sx + x;
        nz_y[widx] = nzw;
        ys_y[widx] = ysw;
        if (HAS_Z) zs_y[widx] = zsw;
      }
    }
    __syncthreads();

    // ---- phase A': per row, carry run boundaries across chunks ---------------------------
    if (lane < 32) {
      const int r = lane;
      if (r < nrows) {
        int last = -1;
        for (int c = 0; c < NC; ++c) {
          pre[r * NC + c] = last;
          const unsigned long long m = st[r * NC + c];
          if (m) last = c * 64 + 63 - __builtin_clzll(m);
        }
      }
    } else {
      const int r = lane - 32;
      if (r < nrows) {
        int nxt = (int)sx;
        for (int c = NC - 1; c >= 0; --c) {
          suf[r * NC + c] = nxt;
          const unsigned long long m = st[r * NC + c];
          if (m) nxt = c * 64 + __builtin_ctzll(m);
        }
      }
    }
    __syncthreads();

    // ---- phase B: distances -----------------------------------------------------------------
#pragma unroll 1
    for (int r = 0; r < nrows; ++r) {
      const int64_t rowbase = (z * sy + (y0 + r)) * sx;
#pragma unroll 2
      for (int c = 0; c < NC; ++c) {
        const int x = c * 64 + lane;
        if (x >= sx) continue;
        const unsigned long long M = st[r * NC + c];
        const bool nz = (fg[r * NC + c] >> lane) & 1ull;
        float f = 0.0f;
        if (nz) {
          const unsigned long long m1 = M & le_mask;
          const int s = m1 ? c * 64 + 63 - __builtin_clzll(m1) : pre[r * NC + c];
          const unsigned long long m2 = M & ~le_mask;
          const int e = (m2 ? c * 64 + __builtin_ctzll(m2) : suf[r * NC + c]) - 1;
          const float L = (s > 0 || bb) ? Ttab[x - s + 1] : INFINITY;
          const float R = (e < (int)sx - 1 || bb) ? Ttab[e - x + 1] : INFINITY;
          const float d = fminf(L, R);
          f = d * d;
          if (to_finite && isinf(f)) f = FLT_MAX;
        }
        out[rowbase + x] = f;
      }
    }
    __syncthreads();  // masks are reused by the next group
  }
}

bool row_pass_tiled_supported(int64_t sx) { return sx >= 1 && sx <= (int64_t)kMaxChunks * 64; }

template <typename T>
static int launch_row_tiled_t(const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                              uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                              int to_finite, hipStream_t stream) {
  const int NC = (int)ceil_div(sx, 64);
  const int64_t nby = ceil_div(sy, kBandRows);
  const int64_t ngroups = nby * sz;
  if (ngroups <= 0) return EDT_OK;
  const size_t lds = (size_t)((sx + 2 + 3) & ~3) * sizeof(float) +
                     (size_t)kWavesPerBlock * 32 * NC * 24;
  static std::atomic<uint64_t> attr_done{0};  // per instantiation, one bit per device
  EDT_HIP_TRY(EDT_LDS_ATTR_ONCE(attr_done, reinterpret_cast<const void *>(&k_row_pass_tiled<T, true>),
                                reinterpret_cast<const void *>(&k_row_pass_tiled<T, false>)));
  int64_t blocks = ceil_div(ngroups, kWavesPerBlock);
  const int64_t resident = 256 * 6;  // persistent grid: the T table is built once per block
  if (blocks > resident) blocks = resident;
  if (zs_y != nullptr)
    hipLaunchKernelGGL((k_row_pass_tiled<T, true>), dim3((unsigned)blocks), dim3(kWavesPerBlock * 64),
                       lds, stream, (const T *)labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb,
                       to_finite, NC, nby, ngroups);
  else
    hipLaunchKernelGGL((k_row_pass_tiled<T, false>), dim3((unsigned)blocks), dim3(kWavesPerBlock * 64),
                       lds, stream, (const T *)labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb,
                       to_finite, NC, nby, ngroups);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

int launch_row_pass_tiled(int dtype, const void *labels, float *out, uint32_t *nz_y, uint32_t *ys_y,
                          uint32_t *zs_y, int64_t sx, int64_t sy, int64_t sz, float w, int bb,
                          int to_finite, hipStream_t stream) {
#define ROW_TILED(T) \
  return launch_row_tiled_t<T>(labels, out, nz_y, ys_y, zs_y, sx, sy, sz, w, bb, to_finite, stream)
  switch (dtype) {
    case EDT_U8: case EDT_BOOL: ROW_TILED(uint8_t);
    case EDT_U16: ROW_TILED(uint16_t);
    case EDT_U32: ROW_TILED(uint32_t);
    case EDT_U64: ROW_TILED(uint64_t);
    case EDT_F32: ROW_TILED(float);
    case EDT_F64: ROW_TILED(double);
    default: set_error("unknown dtype"); return EDT_ERR_BAD_ARG;
  }
#undef ROW_TILED
}

// ---------------------------------------------------------------------------------------
// (nz, zs) words [z][y/32][x]  ->  (nz, rs) words [y][z/32][x].  One thread per
// (x, y-band, z-band): a 32x32 bit-matrix transpose in registers, coalesced across x.
// ---------------------------------------------------------------------------------------
template <int J, uint32_t M>
__device__ __forceinline__ void transpose32_stage(uint32_t (&a)[32]) {
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    if ((k & J) == 0) {
      const uint32_t t = ((a[k] >> J) ^ a[k + J]) & M;  // M: the low J bits of every 2J-bit group
      a[k] ^= t << J;
      a[k + J] ^= t;
    }
  }
}
__device__ __forceinline__ void transpose32(uint32_t (&a)[32]) {
  transpose32_stage<16, 0x0000FFFFu>(a);
  transpose32_stage<8, 0x00FF00FFu>(a);
  transpose32_stage<4, 0x0F0F0F0Fu>(a);
  transpose32_stage<2, 0x33333333u>(a);
  transpose32_stage<1, 0x55555555u>(a);
}

// NZ = false: the run starts alone (round 6: a call whose column passes provably run on the integer kernel never reads a
// foreground plane -- it is neither written by pass X nor transposed)
template <bool NZ>
__global__ void k_bits_transpose_yz(const uint32_t *__restrict__ nz_y,
                                    const uint32_t *__restrict__ zs_y,
                                    uint32_t *__restrict__ nz_z, uint32_t *__restrict__ rs_z,
                                    int64_t sx, int64_t sy, int64_t sz, int64_t nby, int64_t nbz,
                                    int64_t in_zstride) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = sx * nby * nbz;
  if (idx >= total) return;
  const int64_t x = idx % sx;
  const int64_t yb = (idx / sx) % nby;
  const int64_t zb = idx / (sx * nby);
  uint32_t a[32], b[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    const int64_t z = zb * 32 + t;
    a[t] = 0; b[t] = 0;
    if (z < sz) {
      const int64_t w = z * in_zstride + yb * sx + x;
      if constexpr (NZ) a[t] = nz_y[w];
      b[t] = zs_y[w];
    }
  }
  // 32x32 bit-matrix transposes in registers: five butterfly stages of 16 masked swaps each
  // (word t bit r  <->  word r bit t), ~500 operations per plane instead of 32*32 bit extractions
  if constexpr (NZ) transpose32(a);
  transpose32(b);
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const int64_t y = yb * 32 + r;
    if (y >= sy) break;
    const int64_t w = (y * nbz + zb) * sx + x;
    if constexpr (NZ) nz_z[w] = a[r];
    rs_z[w] = b[r];
  }
}

int launch_bits_transpose_yz(const uint32_t *nz_y, const uint32_t *zs_y, uint32_t *nz_z,
                             uint32_t *rs_z, int64_t sx, int64_t sy, int64_t sz,
                             hipStream_t stream, int64_t in_zstride) {
  const int64_t nby = ceil_div(sy, kBandRows), nbz = ceil_div(sz, kBandRows);
  const int64_t total = sx * nby * nbz;
  if (total <= 0) return EDT_OK;
  const int threads = 256;
  if (nz_y != nullptr && nz_z != nullptr)
    hipLaunchKernelGGL(k_bits_transpose_yz<true>, dim3((unsigned)ceil_div(total, threads)), dim3(threads), 0,
                       stream, nz_y, zs_y, nz_z, rs_z, sx, sy, sz, nby, nbz,
                       in_zstride > 0 ? in_zstride : nby * sx);
  else
    hipLaunchKernelGGL(k_bits_transpose_yz<false>, dim3((unsigned)ceil_div(total, threads)), dim3(threads), 0,
                       stream, nz_y, zs_y, nz_z, rs_z, sx, sy, sz, nby, nbz,
                       in_zstride > 0 ? in_zstride : nby * sx);
  EDT_HIP_TRY(hipGetLastError());
  return EDT_OK;
}

}  // namespace edt_amd
