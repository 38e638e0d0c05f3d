// Shared pieces of the "march" kernel (vif_march.hip; a motion form of it was built in round 3 and dropped): both filter passes of a separable Gaussian on
// v_mfma_f32_16x16x32_f16, the second pass fed straight from the first pass's accumulators.  See vif_march.hip for the scheme.
#pragma once
#include <cmath>
#include <cstring>

#include "pqa_device.h"

namespace pqa {
namespace march {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h8 frag4(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(h8, u4v{a, b, c, d});
}
// two integers k < 2048 (one per 16-bit half) -> two f16 (k - off) * 2^-24, exact; off_bits = off as f16 bits | 0x8000
__device__ __forceinline__ unsigned tiny_minus(unsigned x, unsigned short off_bits) {
  const h2 o = __builtin_bit_cast(h2, (unsigned)off_bits | ((unsigned)off_bits << 16));
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, x) + o);
}
__device__ __forceinline__ f4 mma(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// v - hi for a value v and its f16 rounding hi (one half of a packed pair): exact in f32 (hi shares v's leading bits).
// v_fma_mix_f32 reads the f16 half in place -- no conversion instruction, one VALU op per value.
template <int HALF>
__device__ __forceinline__ float residual(const unsigned hi_pair, const float v) {
  float r;
  if (HALF == 0)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi_pair), "s"(-1.0f), "v"(v));
  else
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi_pair), "s"(-1.0f), "v"(v));
  return r;
}

// Four f32 values v * s + b (s an exact power of two; b removes the mid-grey term of the mean planes, see pass1) -> two
// f16 pieces each: hi = rne(x), lo = rne(x - hi).
// hi / lo: {piece(x0), piece(x1)}, {piece(x2), piece(x3)} -- the element order of an MFMA operand.
template <int HALF>   // which half of the operand vectors (0: dwords 0, 1; 1: dwords 2, 3) receives the pieces
__device__ __forceinline__ void split4(const f4 v, const float s, const float b, u4v& hi, u4v& lo) {
  const f2 xa = __builtin_elementwise_fma(f2{v[0], v[1]}, f2{s, s}, f2{b, b});
  const f2 xb = __builtin_elementwise_fma(f2{v[2], v[3]}, f2{s, s}, f2{b, b});
  const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, h2));
  const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(xb, h2));
  const f2 ra = f2{residual<0>(h0, xa[0]), residual<1>(h0, xa[1])};
  const f2 rb = f2{residual<0>(h1, xb[0]), residual<1>(h1, xb[1])};
  hi[2 * HALF] = h0;
  hi[2 * HALF + 1] = h1;
  lo[2 * HALF] = __builtin_bit_cast(unsigned, __builtin_convertvector(ra, h2));
  lo[2 * HALF + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(rb, h2));
}


// ---- host helpers -----------------------------------------------------------------------------------------------
inline void gaussian(int n, float* out) {   // N taps, sigma = N / 5, normalised in double, stored as float (vif_filter1d_table)
  double v[17], sum = 0.0;
  const double sigma = n / 5.0;
  for (int k = 0; k < n; ++k) {
    const double d = k - n / 2;
    v[k] = exp(-0.5 * d * d / (sigma * sigma));
    sum += v[k];
  }
  for (int k = 0; k < n; ++k) out[k] = (float)(v[k] / sum);
}

// x -> n f16 pieces (round to nearest each time); returns what is left
inline double pieces(double x, int n, uint16_t* out /* stride: one fragment */, size_t stride) {
  double r = x;
  for (int p = 0; p < n; ++p) {
    const _Float16 hh = (_Float16)r;
    uint16_t bits;
    memcpy(&bits, &hh, 2);
    out[(size_t)p * stride] = bits;
    r -= (double)hh;
  }
  return r;
}


}  // namespace march
}  // namespace pqa
