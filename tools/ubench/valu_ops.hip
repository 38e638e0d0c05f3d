// Microbenchmark: issue cost of the plain VALU instructions the ADM chain is made of, on gfx950.
// pk_vs_plain.hip / fma_rate.hip showed v_fma_f32 at 2 clocks per wave64 and v_pk_fma_f32 at 4; this one asks the same of
// v_mul / v_add / v_max / v_med3 / v_cndmask / v_cvt_f32_ubyte / v_cmp / DPP moves (row and wave shifts) / DPP-fused adds,
// to decide whether packing two coefficients into v_pk_* ops buys anything for the non-FMA part of a kernel.
// Every op runs on 16 independent registers (no dependent-issue stalls); reports clocks per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define OPS(X)                                                                                         \
  X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", "+v"(y[i]) : "v"(a), "v"(b))                            \
  X(1, "v_mul_f32", "v_mul_f32 %0, %0, %1", "+v"(y[i]) : "v"(a))                                         \
  X(2, "v_add_f32", "v_add_f32 %0, %0, %1", "+v"(y[i]) : "v"(b))                                         \
  X(3, "v_max_f32", "v_max_f32 %0, %0, %1", "+v"(y[i]) : "v"(b))                                         \
  X(4, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2", "+v"(y[i]) : "v"(a), "v"(b))                           \
  X(5, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc", "+v"(y[i]) : "v"(a))                      \
  X(6, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %1", "+v"(y[i]) : "v"(u[i]))                            \
  X(7, "v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 vcc, %0, %1", "+v"(y[i]) : "v"(a) : "vcc")                    \
  X(8, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(y[i]) : "v"(u[i])) \
  X(9, "v_mov_b32 dpp wave_shr:1", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "+v"(y[i]) : "v"(u[i])) \
  X(10, "v_add_f32 dpp row_shr:1", "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(y[i]) : "v"(u[i])) \
  X(11, "v_fmac_f32 dpp wave_shl:1", "v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf", "+v"(y[i]) : "v"(u[i]), "v"(a)) \
  X(12, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", "+v"(x[i & 7]) : "v"(av))                             \
  X(13, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", "+v"(x[i & 7]) : "v"(bv))                             \
  X(14, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", "+v"(x[i & 7]) : "v"(av), "v"(bv))                \
  X(15, "v_sub_f32 |a|-b (VOP3)", "v_sub_f32_e64 %0, |%0|, %1", "+v"(y[i]) : "v"(b))                     \
  X(16, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(sel))          \
  X(17, "v_mul_f32 x2 dependent chain", "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1", "+v"(y[i & 3]) : "v"(a)) \
  X(18, "v_cvt_pk... v_exp_f32 (transcendental)", "v_exp_f32 %0, %0", "+v"(y[i]))                       \
  X(19, "v_rcp_f32", "v_rcp_f32 %0, %0", "+v"(y[i]))                                                   \
  X(20, "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(y[i]) : "v"(a), "s"(m64))  \
  X(21, "v_cmp + v_cndmask (vcc) pair", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(y[i]) : "v"(a) : "vcc") \
  X(22, "v_cndmask_b32 vcc, distinct dst", "v_cndmask_b32 %0, %1, %2, vcc", "=v"(y[i]) : "v"(a), "v"(b))    \
  X(23, "v_cvt_f32_u32 sdwa WORD_1", "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "+v"(y[i]) : "v"(u[i])) \
  X(24, "v_mul_f32 x |abs| e64", "v_mul_f32_e64 %0, |%0|, %1", "+v"(y[i]) : "v"(a))                        \
  X(25, "v_fma_f32 distinct regs", "v_fma_f32 %0, %1, %2, %0", "+v"(y[i]) : "v"(y[(i + 5) & 15]), "v"(a))  \
  X(26, "v_readlane + v_writelane", "v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5", "+v"(y[i]) : : "s20") \
  X(27, "ds_bpermute_b32", "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)", "+v"(y[i]) : "v"(u[i]))

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  f2 x[8];
  float y[16];
  unsigned u[16];
  const unsigned sel = 0x0c010c00u;
  const unsigned long long m64 = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
#pragma unroll
  for (int i = 0; i < 16; ++i) { y[i] = threadIdx.x * 0.003f + i; u[i] = threadIdx.x * 2654435761u + i; }
  const f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#define X(id, name, text, ...) if (OP == id) asm volatile(text : __VA_ARGS__);
        OPS(X)
#undef X
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += y[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(float* d, const char* label, double clk_ghz, int per_iter) {
  const int iters = 2000;
  printf("%-40s", label);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wpss[3] = {2, 4, 8};
  for (int wi = 0; wi < 3; ++wi) {
    const int wps = wpss[wi];
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k<OP>), dim3(256 * wps), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // one SIMD runs wps waves, each iters * per_iter instructions
    const double clk = ms * 1e-3 * clk_ghz * 1e9 / ((double)wps * iters * per_iter);
    printf("  w%d %6.3f ms %5.2f clk/instr", wps, ms, clk);
  }
  printf("\n");
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double ghz = khz * 1e-6;
  printf("# clocks per wave64 instruction per SIMD at the reported %0.3f GHz (the sustained clock may be lower); 2000 x 32 per wave\n", ghz);
#define X(id, name, text, ...) run<id>(d, name, ghz, (id == 17 || id == 21 || id == 26) ? 64 : 32);
  OPS(X)
#undef X
  return 0;
}
