// Microbenchmark: do packed-f32 VALU ops (v_pk_fma_f32: they contend with MFMA for one pipe, see mfma_coissue.hip)
// run CONCURRENTLY with plain VALU ops (v_fma_f32, v_cndmask, v_cvt, ...) on gfx950 -- inside one wave's stream, and
// across the waves of one SIMD?  If they do, a kernel that is bound by VALU issue can gain by expressing part of its
// FMAs in plain form.  Reports wall time for a fixed amount of work at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: NP packed FMAs per iteration.  MODE 1: NS plain FMAs.  MODE 2: both, interleaved 1 pk : (NS/NP) plain in one
// stream.  MODE 3: split -- waves 0,1 run 2*NP packed, waves 2,3 run 2*NS plain (same total work as MODE 2).
template <int MODE, int NP, int NS>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  f2 x[8];
  float y[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
#pragma unroll
  for (int i = 0; i < 16; ++i) y[i] = threadIdx.x * 0.003f + i;
  const f2 av = {a, a}, bv = {b, b};
  const int wave = threadIdx.x >> 6;
  const bool do_p = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 2);
  const bool do_s = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 2);
  constexpr int MULT = MODE == 3 ? 2 : 1;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(av), "v"(bv));
#pragma unroll
        for (int j = 0; j < NS / NP; ++j)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[(i * (NS / NP) + j) & 15]) : "v"(a), "v"(b));
      }
    } else {
      if (do_p) {
#pragma unroll
        for (int i = 0; i < NP * MULT; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(av), "v"(bv));
      }
      if (do_s) {
#pragma unroll
        for (int i = 0; i < NS * MULT; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i & 15]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NP, int NS>
void run(float* d, const char* label) {
  const int iters = 4000;
  printf("%-52s", label);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; ++wps) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k<MODE, NP, NS>), dim3(256 * wps), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("  w%d %6.3f ms", wps, ms);
  }
  printf("\n");
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  printf("# wall time, 4000 iterations, 256*w workgroups of 256 threads (w waves per SIMD)\n");
  run<0, 32, 0>(d, "32 v_pk_fma_f32");
  run<1, 0, 32>(d, "32 v_fma_f32");
  run<1, 0, 64>(d, "64 v_fma_f32");
  run<2, 32, 32>(d, "32 pk + 32 plain, one stream (1:1)");
  run<2, 32, 64>(d, "32 pk + 64 plain, one stream (1:2)");
  run<3, 32, 32>(d, "32 pk + 32 plain, split waves");
  run<3, 32, 64>(d, "32 pk + 64 plain, split waves");
  return 0;
}
