// tools/ubench.hip -- VALU issue-cost microbenchmark for gfx950 (diagnostics, not part of the library).
// Each kernel runs ITER x 64 copies of one instruction on 4 independent registers per lane, with
// 8 waves per SIMD resident, and reports SIMD cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 256;
#define REP4(S) S(a) S(b) S(c) S(d)
#define REP64(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S) REP4(S)
#define KERNEL32(NAME, ASM)                                                                 \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {               \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, k = seed | 3u; \
    for (int i = 0; i < ITER; ++i) {                                                        \
      REP64(ASM)                                                                            \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                             \
  }
#define KERNEL64(NAME, ASM)                                                                 \
  __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {               \
    double a = threadIdx.x + seed, b = a * 3.0 + 1.0, c = a + 0.5, d = a + 7.0, k = 1.0000001; \
    for (int i = 0; i < ITER; ++i) {                                                        \
      REP64(ASM)                                                                            \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);                 \
  }
#define A_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(k));
#define A_FFBH(x) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
#define A_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define A_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define A_CMP(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
#define A_CVTF32I(x) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));
#define A_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k));
#define A_ADDC(x) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : : "vcc");
#define A_DPP(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define D_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define D_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define D_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define D_CVT(x) asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0" : "=&v"(tmp), "+v"(x));
#define D_CMP(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc");
#define D_MOV(x) asm volatile("v_mov_b64 %0, %1" : "+v"(x) : "v"(k));
#define D_MIN(x) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define D_CVTI(x) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(x) : "v"(ki));
KERNEL32(k_add, A_ADD) KERNEL32(k_and, A_AND) KERNEL32(k_lshlor, A_LSHLOR) KERNEL32(k_ffbh, A_FFBH)
KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mullo, A_MULLO) KERNEL32(k_fma32, A_FMA32) KERNEL32(k_cmpsel, A_CMP)
KERNEL32(k_cvtf32i, A_CVTF32I) KERNEL32(k_mov, A_MOV) KERNEL32(k_addc, A_ADDC) KERNEL32(k_dpp, A_DPP)
KERNEL64(k_dadd, D_ADD) KERNEL64(k_dmul, D_MUL) KERNEL64(k_dfma, D_FMA) KERNEL64(k_dcmp, D_CMP)
KERNEL64(k_dmov, D_MOV) KERNEL64(k_dmin, D_MIN)
__global__ void __launch_bounds__(256) k_dcvt(uint32_t *out, uint32_t seed) {
  double a = threadIdx.x + seed, b = a * 3.0 + 1.0, c = a + 0.5, d = a + 7.0; float tmp;
  for (int i = 0; i < ITER; ++i) { REP64(D_CVT) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);
}
__global__ void __launch_bounds__(256) k_dcvti(uint32_t *out, uint32_t seed) {
  double a = threadIdx.x + seed, b = a * 3.0 + 1.0, c = a + 0.5, d = a + 7.0; int ki = seed;
  for (int i = 0; i < ITER; ++i) { REP64(D_CVTI) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);
}
template <typename K>
int run(const char *name, K kern, int instr_per_rep, uint32_t *out) {
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr_per_simd = 8.0 * ITER * 64 * instr_per_rep;  // 8 waves per SIMD
  printf("%-10s %8.3f ms   %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, ms,
         ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  return 0;
}
int main() {
  uint32_t *out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
  run("add_u32", k_add, 1, out); run("and_b32", k_and, 1, out); run("lshl_or", k_lshlor, 1, out);
  run("ffbh", k_ffbh, 1, out); run("mul_u24", k_mul24, 1, out); run("mul_lo", k_mullo, 1, out);
  run("fma_f32", k_fma32, 1, out); run("cmp+sel", k_cmpsel, 2, out); run("cvt_f32_i", k_cvtf32i, 1, out);
  run("mov_b32", k_mov, 1, out); run("addc", k_addc, 1, out); run("mov_dpp", k_dpp, 1, out);
  run("add_f64", k_dadd, 1, out); run("mul_f64", k_dmul, 1, out); run("fma_f64", k_dfma, 1, out);
  run("cmp_f64", k_dcmp, 1, out); run("mov_b64", k_dmov, 1, out); run("min_f64", k_dmin, 1, out);
  run("cvt64<->32", k_dcvt, 2, out); run("cvt_f64_i32", k_dcvti, 1, out);
  return 0;
}
