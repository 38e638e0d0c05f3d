// Microbenchmark: v_mfma_f32_16x16x16_f16 (K = 16, 2-VGPR operands) against v_mfma_f32_16x16x32_f16 (K = 32, 4-VGPR
// operands) on gfx950: cycles per instruction with 1, 2 and 4 independent accumulators, 1 wave per SIMD and 2.
// Question behind it (vif_march.hip): the second filter pass contracts over (previous block, current block); with K = 16
// each block is its own MFMA (no concatenated operand, no register copies) -- worth it only if the K = 16 form costs half.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int K32, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f4 c[4] = {f4{0, 0, 0, 0}, f4{1, 1, 1, 1}, f4{2, 2, 2, 2}, f4{3, 3, 3, 3}};
  h8 a8, b8; h4 a4, b4;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.01f + i); b8[i] = (_Float16)(1.0f - i * 0.1f); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (K32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a8), "v"(b8));
      else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c[m % NACC]) : "v"(a4), "v"(b4));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int K32, int NACC>
void run(float* d, long long* dc, int wgs_per_cu) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K32, NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long cyc = 0;
  hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  // s_memtime-style counter runs at a fixed 100 MHz on this part: report wall time per MFMA per wave-slot instead
  const double per = ms * 1e6 / (iters * 32.0 * wgs_per_cu);   // ns per MFMA per SIMD (each SIMD runs wgs_per_cu waves)
  printf("%s  acc=%d  waves/SIMD=%d  %.3f ms  %.2f ns per MFMA per SIMD  (~%.1f clk at 2.1 GHz)\n", K32 ? "16x16x32" : "16x16x16",
         NACC, wgs_per_cu, ms, per, per * 2.1);
}

int main() {
  float* d; long long* dc;
  hipMalloc(&d, 256 * 4 * 256 * 4); hipMalloc(&dc, 8);
  for (int w = 1; w <= 2; ++w) {
    run<1, 1>(d, dc, w); run<1, 2>(d, dc, w); run<1, 4>(d, dc, w);
    run<0, 1>(d, dc, w); run<0, 2>(d, dc, w); run<0, 4>(d, dc, w);
  }
  return 0;
}
