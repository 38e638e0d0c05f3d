// Do the f16 paths this kernel family relies on keep DENORMAL inputs on gfx950?
//   (1) v_mfma_f32_16x16x32_f16 with a denormal B operand (bits 0x0001 .. 0x03ff = k * 2^-24)
//   (2) v_pk_fma_f16 with a denormal multiplicand: bits(k) * 32768 - c  (u8 / u10 sample -> scaled f16 in one instruction)
// Prints the values the hardware returns next to the exact ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__global__ void k(float* out, float* out2) {
  const int lane = threadIdx.x, n = lane & 15;
  h8 a;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
  const unsigned bits = (unsigned)(n * 67 + 1) & 0x3ffu;           // denormal pattern, differs per column
  const unsigned pair = bits | (bits << 16);
  const h8 b = __builtin_bit_cast(h8, u4v{pair, pair, pair, pair});
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  out[lane] = c[0];                                                // = 32 * bits * 2^-24 when denormals are kept
  const unsigned x = (unsigned)lane * 4u + 1u;                     // a "sample" 1 .. 253
  const h2 v = __builtin_elementwise_fma(__builtin_bit_cast(h2, x | (x << 16)), h2{(_Float16)32768.0f, (_Float16)32768.0f},
                                         h2{(_Float16)-0.25f, (_Float16)-0.25f});
  out2[lane] = (float)v[0];                                        // = (x - 128) * 2^-9
}

int main() {
  float *d, *d2, h[64], h2_[64];
  hipMalloc(&d, 256); hipMalloc(&d2, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d2);
  hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(h2_, d2, 256, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned bits = (unsigned)((l & 15) * 67 + 1) & 0x3ffu;
    const float want = 32.0f * bits * ldexpf(1.0f, -24);
    if (h[l] != want) ++bad1;
    const float want2 = ((float)(l * 4 + 1) - 128.0f) * ldexpf(1.0f, -9);
    if (h2_[l] != want2) ++bad2;
    if (l < 4) printf("lane %d: mfma %.9g (exact %.9g)   pk_fma_f16 %.9g (exact %.9g)\n", l, h[l], want, h2_[l], want2);
  }
  printf("mfma f16 denormal operand: %s (%d of 64 lanes differ)\n", bad1 ? "NOT kept" : "kept exactly", bad1);
  printf("v_pk_fma_f16 denormal multiplicand: %s (%d of 64 lanes differ)\n", bad2 ? "NOT kept" : "kept exactly", bad2);
  return 0;
}
