// Microbenchmark: what does an f32 MFMA cost the VALU port of its SIMD on gfx950?
// Decides whether a banded-Toeplitz MFMA pass can run BESIDE the packed-FMA passes of the VIF kernel
// (VERDICT r1 item 4(i)).  For each (NV packed FMAs, NM MFMAs) per loop iteration it reports shader cycles per
// iteration per wave at 1, 2, 3, 4 waves per SIMD (s_memtime deltas, median over workgroups), for
//   v_mfma_f32_16x16x4_f32 (32 cyc/SIMD issue) and v_mfma_f32_32x32x2_f32 (64 cyc/SIMD issue),
// plus the bf16 shapes (32x32x16, 16x16x32) that a digit-split vertical pass would use;
// in two arrangements: every wave runs the mixed stream, or half the waves run MFMA only and half VALU only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, int NM, int SHAPE, int SPLIT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b, int iters) {
  f2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 av = {a, a}, bv = {b, b};
  f4 c4[2] = {f4{0, 0, 0, 0}, f4{1, 1, 1, 1}};
  f16v c16[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { c16[0][i] = 0.f; c16[1][i] = 1.f; }
  const float ma = threadIdx.x * 0.01f, mb = 1.0f - threadIdx.x * 0.003f;
  const f4 ba = f4{ma, mb, ma, mb}, bb = f4{mb, ma, mb, ma};  // 8 bf16 per lane (bit patterns do not matter for timing)
  const int wave = threadIdx.x >> 6;
  // SPLIT: waves 0,1 of the workgroup do the MFMAs, waves 2,3 the packed FMAs (twice as many each, same total)
  const bool do_v = SPLIT ? (wave >= 2) : true, do_m = SPLIT ? (wave < 2) : true;
  constexpr int MULT = SPLIT ? 2 : 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < NM * MULT; ++m) {
        // asm volatile keeps the written order (the compiler otherwise folds the FMA recurrences and moves the MFMAs)
        if (SHAPE == 1632) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[m & 1]) : "v"(ba), "v"(bb));
        else if (SHAPE == 3216) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[m & 1]) : "v"(ba), "v"(bb));
        else if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4[m & 1]) : "v"(ma), "v"(mb));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16[m & 1]) : "v"(ma), "v"(mb));
        if (!SPLIT) {
#pragma unroll
          for (int i = 0; i < NV / (NM ? NM : 1); ++i) {
            const int j = (m * (NV / (NM ? NM : 1)) + i) & 7;
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(av), "v"(bv));
          }
        }
      }
    }
    if (do_v && (SPLIT || NM == 0)) {
#pragma unroll
      for (int i = 0; i < NV * MULT; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(av), "v"(bv));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  s += c4[0][0] + c4[1][1];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c16[0][i] + c16[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NM, int SHAPE, int SPLIT>
void run(float* d, unsigned long long* dc, const char* label) {
  const int iters = 4000;
  printf("%-44s", label);
  for (int wps = 1; wps <= 4; ++wps) {  // 256-thread workgroups per CU = waves per SIMD
    const int grid = 256 * wps;
    float ms = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<NV, NM, SHAPE, SPLIT>), dim3(grid), dim3(256), 0, 0, d, dc, 1.0001f, 0.5f, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // cycles of SIMD time per iteration of ONE wave's share: all wps waves of a SIMD run concurrently, so the SIMD
    // spends (median wave cycles / iters) per `wps` wave-iterations
    const double per_iter = (double)h[grid / 2] / iters / wps * (SPLIT ? 2.0 : 1.0);
    printf("  w%d %7.1f cyc (%.2f ms)", wps, per_iter, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("\n");
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  unsigned long long* dc; hipMalloc(&dc, 4096 * sizeof(unsigned long long));
  printf("# SIMD cycles per wave-iteration (median wave s_memtime / iters / waves-per-SIMD); s_memtime ticks at 100 MHz\n");
  printf("# if the numbers look 20x small: scale by shader clock / 100 MHz.  wN = N waves per SIMD\n");
  run<32, 0, 16, 0>(d, dc, "32 pk_fma");
  run<64, 0, 16, 0>(d, dc, "64 pk_fma");
  run<0, 4, 16, 0>(d, dc, "4 mfma16x16x4");
  run<0, 2, 32, 0>(d, dc, "2 mfma32x32x2");
  run<16, 4, 16, 0>(d, dc, "4 mfma16 + 16 pk_fma, one stream");
  run<32, 4, 16, 0>(d, dc, "4 mfma16 + 32 pk_fma, one stream");
  run<64, 4, 16, 0>(d, dc, "4 mfma16 + 64 pk_fma, one stream");
  run<32, 2, 32, 0>(d, dc, "2 mfma32 + 32 pk_fma, one stream");
  run<64, 2, 32, 0>(d, dc, "2 mfma32 + 64 pk_fma, one stream");
  printf("# bf16 MFMA beside packed f32 FMAs\n");
  run<0, 4, 3216, 0>(d, dc, "4 mfma32x32x16_bf16");
  run<0, 8, 1632, 0>(d, dc, "8 mfma16x16x32_bf16");
  run<32, 4, 3216, 0>(d, dc, "4 mfma32_bf16 + 32 pk_fma, one stream");
  run<64, 4, 3216, 0>(d, dc, "4 mfma32_bf16 + 64 pk_fma, one stream");
  run<128, 4, 3216, 0>(d, dc, "4 mfma32_bf16 + 128 pk_fma, one stream");
  run<32, 8, 1632, 0>(d, dc, "8 mfma16_bf16 + 32 pk_fma, one stream");
  run<64, 8, 1632, 0>(d, dc, "8 mfma16_bf16 + 64 pk_fma, one stream");
  run<128, 8, 1632, 0>(d, dc, "8 mfma16_bf16 + 128 pk_fma, one stream");
  run<64, 4, 3216, 1>(d, dc, "4 mfma32_bf16 + 64 pk_fma, split waves");
  run<128, 4, 3216, 1>(d, dc, "4 mfma32_bf16 + 128 pk_fma, split waves");
  run<32, 4, 16, 1>(d, dc, "4 mfma16 + 32 pk_fma, split waves");
  run<64, 4, 16, 1>(d, dc, "4 mfma16 + 64 pk_fma, split waves");
  run<64, 2, 32, 1>(d, dc, "2 mfma32 + 64 pk_fma, split waves");
  return 0;
}
