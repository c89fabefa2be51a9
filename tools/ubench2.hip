// tools/ubench2.hip -- issue cost of the instructions the 16-bit integer column kernel (edt_colq16) is made of, next to the
// ones of the fp32 windowed path, on gfx950 (diagnostics, not part of the library).  Every kernel runs ITER x 64 copies of
// one instruction on 8 independent registers per lane; reported: SIMD cycles per wave instruction at 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 256;
#define REP8(S) S(a) S(b) S(c) S(d) S(e) S(f) S(g) S(h)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#define KERNEL32(NAME, ASM)                                                                 \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {               \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = a + 9u, f = a * 5u, g = a ^ 0x33u, \
             h = a + 11u, k = seed | 3u, k2 = seed + 5u;                                    \
    __shared__ uint32_t lds[4096];                                                          \
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;                                       \
    __syncthreads();                                                                        \
    uint32_t la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024;                      \
    for (int i = 0; i < ITER; ++i) {                                                        \
      REP64(ASM)                                                                            \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + la + k2;   \
  }
#define A_MINU32(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MIN3U32(x) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(k2));
#define A_PKADDF32(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xx) : "v"(kk));
#define A_ADDF32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_FMAF32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(k2));
#define A_PKMINU16(x) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKADDU16C(x) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(k));
#define A_PKADDU16S(x) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(x) : "s"(seed));
#define A_PKMULU16(x) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKSHL16(x) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(x));
#define A_PKASHR16(x) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(x));
#define A_BFI(x) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(k), "v"(k2));
#define A_CVTSDWA(x) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x));
#define A_MINU16(x) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_DSREAD(x) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(la));
#define A_DSREAD8(x) asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(x) : "v"(la));
#define A_DSREAD64(x) asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(yy) : "v"(la));
KERNEL32(k_minu32, A_MINU32) KERNEL32(k_min3u32, A_MIN3U32) KERNEL32(k_addf32, A_ADDF32) KERNEL32(k_fmaf32, A_FMAF32)
KERNEL32(k_pkminu16, A_PKMINU16) KERNEL32(k_pkaddu16c, A_PKADDU16C) KERNEL32(k_pkaddu16s, A_PKADDU16S)
KERNEL32(k_pkmulu16, A_PKMULU16) KERNEL32(k_pkshl16, A_PKSHL16) KERNEL32(k_pkashr16, A_PKASHR16) KERNEL32(k_bfi, A_BFI)
KERNEL32(k_cvtsdwa, A_CVTSDWA) KERNEL32(k_minu16, A_MINU16)
#define A_PKMAXF16(x) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKMINF16(x) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKADDF16(x) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MAXF16(x) asm volatile("v_max_f16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MAXF32(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MINF32(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKMAXI16(x) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_PKSUBU16C(x) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(x) : "v"(k));
#define A_MAXU16(x) asm volatile("v_max_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_ADDU16(x) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_SUBU16(x) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MAXU32(x) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_ADDU32(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MIN3F32(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(k2));
#define A_PKFMAF16(x) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(k2));
#define A_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(k2));
KERNEL32(k_pkmaxf16, A_PKMAXF16) KERNEL32(k_pkminf16, A_PKMINF16) KERNEL32(k_pkaddf16, A_PKADDF16) KERNEL32(k_maxf16, A_MAXF16)
KERNEL32(k_maxf32, A_MAXF32) KERNEL32(k_minf32, A_MINF32) KERNEL32(k_pkmaxi16, A_PKMAXI16) KERNEL32(k_pksubu16c, A_PKSUBU16C)
KERNEL32(k_maxu16, A_MAXU16) KERNEL32(k_addu16, A_ADDU16) KERNEL32(k_subu16, A_SUBU16) KERNEL32(k_maxu32, A_MAXU32)
KERNEL32(k_addu32, A_ADDU32) KERNEL32(k_min3f32, A_MIN3F32) KERNEL32(k_pkfmaf16, A_PKFMAF16) KERNEL32(k_andor, A_ANDOR)
// LDS reads: 8 independent loads in flight, one wait per 8
__global__ void __launch_bounds__(256) k_dsread(uint32_t *out, uint32_t seed) {
  __shared__ uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i + seed;
  __syncthreads();
  uint32_t la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048, a, b, c, d, e, f, g, h, acc = 0;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                   "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(la));
      acc += a + b + c + d + e + f + g + h;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the same with a 2-way bank pattern of the q16 tile (16 lanes x 4 bytes contiguous, four groups 512 bytes apart)
__global__ void __launch_bounds__(256) k_dsread_q16(uint32_t *out, uint32_t seed, int gstride) {
  __shared__ uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i + seed;
  __syncthreads();
  uint32_t la = (threadIdx.x & 15) * 4 + ((threadIdx.x >> 4) & 3) * gstride + (threadIdx.x >> 6) * 4096, a, b, c, d, e, f, g, h, acc = 0;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:128\n ds_read_b32 %2, %8 offset:256\n ds_read_b32 %3, %8 offset:384\n"
                   "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1152\n ds_read_b32 %6, %8 offset:1280\n ds_read_b32 %7, %8 offset:1408\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(la));
      acc += a + b + c + d + e + f + g + h;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// v_pk_add_f32 on register pairs
__global__ void __launch_bounds__(256) k_pkaddf32(uint32_t *out, uint32_t seed) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a = {(float)threadIdx.x, 1.0f}, b = a * 3.0f, c = a + 0.5f, d = a + 7.0f, kk = {1.0000001f, 0.5f};
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(kk));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b) : "v"(kk));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c) : "v"(kk));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(kk));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a.x + b.y + c.x + d.y);
}
template <typename K, typename... A>
int run(const char *name, K kern, int blocks_per_cu, uint32_t *out, A... extra) {
  const int blocks = 256 * blocks_per_cu;  // blocks of 4 waves: blocks_per_cu waves per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u, extra...);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u, extra...);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr_per_simd = (double)blocks_per_cu * ITER * 64;
  printf("%-22s %d waves/SIMD %8.3f ms   %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, blocks_per_cu, ms,
         ms * 1e6 / wave_instr_per_simd * 2.4);
  return 0;
}
int main() {
  uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
  for (int occ : {4}) {
    run("v_min_u32", k_minu32, occ, out); run("v_min3_u32", k_min3u32, occ, out); run("v_add_f32", k_addf32, occ, out);
    run("v_fma_f32", k_fmaf32, occ, out); run("v_pk_add_f32", k_pkaddf32, occ, out);
    run("v_pk_min_u16", k_pkminu16, occ, out); run("v_pk_add_u16 clamp", k_pkaddu16c, occ, out);
    run("v_pk_add_u16 clamp sgpr", k_pkaddu16s, occ, out);
    run("v_pk_mul_lo_u16", k_pkmulu16, occ, out); run("v_pk_lshlrev_b16", k_pkshl16, occ, out);
    run("v_pk_ashrrev_i16", k_pkashr16, occ, out); run("v_bfi_b32", k_bfi, occ, out);
    run("v_cvt_f32_u32 sdwa", k_cvtsdwa, occ, out); run("v_min_u16", k_minu16, occ, out);
    run("v_pk_max_f16", k_pkmaxf16, occ, out); run("v_pk_min_f16", k_pkminf16, occ, out); run("v_pk_add_f16", k_pkaddf16, occ, out);
    run("v_max_f16", k_maxf16, occ, out); run("v_max_f32", k_maxf32, occ, out); run("v_min_f32", k_minf32, occ, out);
    run("v_pk_max_i16", k_pkmaxi16, occ, out); run("v_pk_sub_u16 clamp", k_pksubu16c, occ, out); run("v_max_u16", k_maxu16, occ, out);
    run("v_add_u16", k_addu16, occ, out); run("v_sub_u16", k_subu16, occ, out); run("v_max_u32", k_maxu32, occ, out);
    run("v_add_u32", k_addu32, occ, out); run("v_min3_f32", k_min3f32, occ, out); run("v_pk_fma_f16", k_pkfmaf16, occ, out);
    run("v_and_or_b32", k_andor, occ, out);
    run("ds_read_b32 x8 linear", k_dsread, occ, out);
    run("ds_read_b32 q16 g=512", k_dsread_q16, occ, out, 512);
    run("ds_read_b32 q16 g=576", k_dsread_q16, occ, out, 576);
    run("ds_read_b32 q16 g=64", k_dsread_q16, occ, out, 64);
  }
  return 0;
}
