// Microbenchmark: does v_pk_fma_f32 double FP32 FMA throughput over v_fma_f32 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b, int iters) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b, int iters) {
  float2v x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const float2v av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  for (int wpb = 1; wpb <= 8; wpb *= 2) {           // blocks per CU
    int grid = 256 * wpb;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a); hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double fl = 2.0 * 16 * iters * 256.0 * grid;
      if (rep) printf("scalar v_fma   blocks/CU=%d  %.1f TFLOP/s\n", wpb, fl / ms / 1e9);
      hipEventRecord(a); hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      if (rep) printf("packed v_pk_fma blocks/CU=%d  %.1f TFLOP/s\n", wpb, fl / ms / 1e9);
    }
  }
  return 0;
}
