// Microbenchmark: does the f16 matrix core (v_mfma_f32_16x16x32_f16) run BESIDE non-packed VALU work on gfx950?
// mfma_coissue.hip showed that v_pk_fma_f32 and every MFMA shape share one pipe (their times add).  The scale-0 VIF
// kernel also issues ~1000 plain VALU instructions per wave (converts, integer digit splits, selects, plain FMAs); if
// those overlap with MFMAs of other waves, moving packed-FMA work to the matrix core pays twice.
// For each VALU kind it prints the wall time of: the VALU stream alone, the MFMA stream alone, both in one wave
// (interleaved 1 MFMA : NV/NM VALU), and both split over the waves of a workgroup -- at 2, 3 and 4 waves per SIMD.
// "adds" = mixed ~ alone_v + alone_m; "overlaps" = mixed ~ max(alone_v, alone_m).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { PK_FMA = 0, FMA = 1, AND = 2, CVT16 = 3, PERM = 4, PK_F16 = 5, CNDMASK = 6, MUL_LO = 7 };

template <int KIND>
__device__ __forceinline__ void valu_op(f2& x, unsigned& u, const f2 av, const f2 bv) {
  if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(av), "v"(bv));
  else if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(av.x), "v"(bv.x));
  else if (KIND == AND) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u) : "v"(0xffff00ffu), "v"(0x100u));
  else if (KIND == CVT16) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u) : "v"(x.x));
  else if (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(0x64646464u), "v"(0x04010400u));
  else if (KIND == PK_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u) : "v"(0x3c003c00u), "v"(0u));
  else if (KIND == CNDMASK) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x.x) : "v"(av.x));
  else asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(3u));
}

// MODE 0: VALU only, 1: MFMA only, 2: both in every wave, 3: waves 0,1 MFMA (x2) and waves 2,3 VALU (x2)
template <int KIND, int NV, int NM, int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  f2 x[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i}; u[i] = threadIdx.x * 77u + i; }
  const f2 av = {a, a}, bv = {b, b};
  f4 c4[4] = {f4{0, 0, 0, 0}, f4{1, 1, 1, 1}, f4{2, 2, 2, 2}, f4{3, 3, 3, 3}};
  h8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 0.01f + i); fb[i] = (_Float16)(1.0f - i * 0.1f); }
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && wave < 2);
  const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && wave >= 2);
  constexpr int MULT = MODE == 3 ? 2 : 1;
  constexpr int PER = NV / NM;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[m & 3]) : "v"(fa), "v"(fb));
#pragma unroll
        for (int i = 0; i < PER; ++i) valu_op<KIND>(x[(m * PER + i) & 7], u[(m * PER + i) & 7], av, bv);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM * MULT; ++m)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[m & 3]) : "v"(fa), "v"(fb));
      }
      if (do_v) {
#pragma unroll
        for (int i = 0; i < NV * MULT; ++i) valu_op<KIND>(x[i & 7], u[i & 7], av, bv);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + (float)u[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c4[i][0] + c4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, int NM, int MODE>
float timed(float* d, int wps) {
  const int iters = 2000, grid = 256 * wps;
  float ms = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV, NM, MODE>), dim3(grid), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

template <int KIND, int NV, int NM>
void run(float* d, const char* label) {
  printf("%-34s", label);
  for (int wps = 2; wps <= 4; ++wps) {
    const float v = timed<KIND, NV, NM, 0>(d, wps), m = timed<KIND, NV, NM, 1>(d, wps);
    const float both = timed<KIND, NV, NM, 2>(d, wps), split = timed<KIND, NV, NM, 3>(d, wps);
    printf("  w%d valu %.3f mfma %.3f mixed %.3f split %.3f (sum %.3f)", wps, v, m, both, split, v + m);
  }
  printf("\n");
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  printf("# wall ms, 2000 iterations, 256*w workgroups of 256 threads; per iteration: 8 v_mfma_f32_16x16x32_f16 and NV VALU ops\n");
  run<PK_FMA, 32, 8>(d, "32 v_pk_fma_f32");
  run<FMA, 64, 8>(d, "64 v_fma_f32");
  run<FMA, 32, 8>(d, "32 v_fma_f32");
  run<AND, 64, 8>(d, "64 v_and_or_b32");
  run<CVT16, 64, 8>(d, "64 v_cvt_f16_f32");
  run<PERM, 64, 8>(d, "64 v_perm_b32");
  run<PK_F16, 64, 8>(d, "64 v_pk_fma_f16");
  run<CNDMASK, 64, 8>(d, "64 v_max_f32");
  run<MUL_LO, 32, 8>(d, "32 v_mul_lo_u32");
  return 0;
}
