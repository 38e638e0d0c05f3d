// Microbenchmark: issue rates of the integer instructions the fixed-point kernels lean on (gfx950), relative to
// v_fma_f32.  One wave64 VALU instruction issues per 4 cycles per SIMD at full rate; "quarter rate" ops take 16.
// Prints G-instructions/s per op (chip-wide) -- compare with the v_fma_f32 line.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

#define KERNEL(name, T, INIT, OP)                                                            \
  __global__ __launch_bounds__(256) void name(unsigned* out, unsigned a, unsigned b, int iters) { \
    T x[16];                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = INIT;                              \
    for (int it = 0; it < iters; ++it) {                                                     \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) { OP; }                                 \
    }                                                                                        \
    unsigned long long s = 0;                                                                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s += (unsigned long long)x[i];            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)s ^ (unsigned)(s >> 32);          \
  }

KERNEL(k_fma, float, (float)(threadIdx.x + i), x[i] = fmaf(x[i], __uint_as_float(a), __uint_as_float(b)))
KERNEL(k_udot2, unsigned, threadIdx.x + i,
       x[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, x[i]), __builtin_bit_cast(u16x2, a), b, false))
KERNEL(k_sdot2, int, threadIdx.x + i,
       x[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, x[i]), __builtin_bit_cast(i16x2, a), (int)b, false))
KERNEL(k_sdot2_acc, int, threadIdx.x + i,
       x[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), x[i], false))
KERNEL(k_mul_lo, unsigned, threadIdx.x + i, x[i] = x[i] * a + b)
KERNEL(k_mad24, unsigned, threadIdx.x + i, x[i] = __umul24(x[i], a) + b)
KERNEL(k_mad_u64, unsigned long long, threadIdx.x + i, x[i] = (unsigned long long)(unsigned)x[i] * a + x[i])
KERNEL(k_mul64, unsigned long long, threadIdx.x + i, x[i] = x[i] * (x[i] | a) + b)
KERNEL(k_pk_mul16, unsigned, threadIdx.x + i,
       x[i] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, x[i]) * __builtin_bit_cast(u16x2, a))) + b)
KERNEL(k_fma64, double, (double)(threadIdx.x + i), x[i] = fma(x[i], (double)__uint_as_float(a), (double)__uint_as_float(b)))
KERNEL(k_div64, double, (double)(threadIdx.x + i + 1), x[i] = (double)__uint_as_float(a) / (x[i] + 1.0))

int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 2048 * sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * 8;
  struct { const char* name; void (*fn)(unsigned*, unsigned, unsigned, int); } ks[] = {
      {"v_fma_f32", k_fma}, {"v_dot2_u32_u16", k_udot2}, {"v_dot2_i32_i16 (chain on src)", k_sdot2},
      {"v_dot2c_i32_i16 (accumulate)", k_sdot2_acc}, {"v_mul_lo_u32 + add", k_mul_lo}, {"v_mad_u32_u24", k_mad24},
      {"v_mad_u64_u32", k_mad_u64}, {"64 x 64 multiply + add", k_mul64}, {"v_pk_mul_lo_u16 + add", k_pk_mul16},
      {"v_fma_f64", k_fma64}, {"f64 division (+ add)", k_div64}};
  for (auto& k : ks) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.fn, dim3(grid), dim3(256), 0, 0, d, 0x3f800347u, 0x00030005u, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double ops = 16.0 * iters * 256.0 * grid;
    printf("%-34s %8.1f G source-ops/s  (%.3f ms)\n", k.name, ops / best / 1e6, best);
  }
  return 0;
}
