// Per-coefficient pieces of ADM shared by the march kernels (adm_march.hip: one scale per launch; adm_pyramid.hip: scales
// 0 and 1 in one launch): the db2 filter, the DPP wave shifts that stand in for an LDS transposition, and the decouple /
// CSF / masking chain.  Arithmetic: libvmaf adm_tools.c (adm_dwt2_s, adm_decouple_s, adm_csf_s, adm_cm_s) restated in
// oracle/vmaf_oracle.c:271-426; operation order as in adm_scale_kernel (adm.hip), which explains the decouple-as-one-median
// identity.  Every kernel that includes this computes bit-identical coefficient values.
#pragma once
#include "pqa_device.h"

namespace pqa {
namespace admc {

constexpr float kLo0 = 0.482962913144690f, kLo1 = 0.836516303737469f, kLo2 = 0.224143868041857f, kLo3 = -0.129409522550921f;
constexpr float kHi0 = -0.129409522550921f, kHi1 = -0.224143868041857f, kHi2 = 0.836516303737469f, kHi3 = -0.482962913144690f;

__device__ __forceinline__ f2 splat2(float c) { return f2{c, c}; }
// taps accumulate in libvmaf's order: ((c0*s0 + c1*s1) + c2*s2) + c3*s3
__device__ __forceinline__ f2 dwt_lo(f2 t0, f2 t1, f2 t2, f2 t3) {
  return __builtin_elementwise_fma(splat2(kLo3), t3,
                                   __builtin_elementwise_fma(splat2(kLo2), t2, __builtin_elementwise_fma(splat2(kLo1), t1, splat2(kLo0) * t0)));
}
__device__ __forceinline__ f2 dwt_hi(f2 t0, f2 t1, f2 t2, f2 t3) {
  return __builtin_elementwise_fma(splat2(kHi3), t3,
                                   __builtin_elementwise_fma(splat2(kHi2), t2, __builtin_elementwise_fma(splat2(kHi1), t1, splat2(kHi0) * t0)));
}

// lane l <- lane l - 1 (wave_shr:1) / lane l + 1 (wave_shl:1) across the whole wave; the end lanes read 0
__device__ __forceinline__ float from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 from_left(f2 v) { return f2{from_left(v.x), from_left(v.y)}; }
__device__ __forceinline__ f2 from_right(f2 v) { return f2{from_right(v.x), from_right(v.y)}; }

// (the value goes through a by-value float parameter: __builtin_bit_cast applied directly to a vector ELEMENT expression
// compiled to the vector's first element on this toolchain -- both stores of a {ref, dis} pair wrote .x)
__device__ __forceinline__ void store_f32(const float v, const rsrc_t rs, const unsigned voff, const unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// One input row of a lane: columns 2c and 2c + 1, each {ref, dis}
struct Row {
  f2 c0, c1;
};

struct Consts {
  float gain_limit, rf_hv, rf_d, k_hv, k_d;   // k = rf / 30: CSF weight and the masking signal's 1/30 in one constant
};

// What a coefficient leaves behind until the masking sum of the row below exists: the restored coefficients, the centre
// of the masking box
struct Pending {
  float rh, rv, rd, g;
};

// decouple + enhancement-gain limit as one median per orientation, CSF of the additive image: bands {ref, dis} of one
// coefficient -> its restored coefficients (p.rh, p.rv, p.rd) and its masking signal (returned; NOT yet stored in p.g: the
// callers fix the band-edge columns first)
__device__ __forceinline__ float decouple(const f2 bh, const f2 bv, const f2 bd, const Consts& k, Pending& p) {
  const float cos_1deg_sq = 0.99969541350954788f;  // cos(pi/180)^2
  const float oh = bh.x, th = bh.y, ov = bv.x, tv = bv.y, od = bd.x, td = bd.y;
  const float ot_dp = fmaf(ov, tv, oh * th);
  const f2 mag = __builtin_elementwise_fma(bv, bv, bh * bh);   // {|o|^2, |t|^2}
  const float lhs = ot_dp * ot_dp, rhs = cos_1deg_sq * mag.x * mag.y;
  const bool ang = (ot_dp >= 0.0f) && (lhs >= rhs);
  const float m = ang ? k.gain_limit : 1.0f;
  p.rh = __builtin_amdgcn_fmed3f(0.0f, th, oh * m);
  p.rv = __builtin_amdgcn_fmed3f(0.0f, tv, ov * m);
  p.rd = __builtin_amdgcn_fmed3f(0.0f, td, od * m);
  const float ah = th - p.rh, av = tv - p.rv, ad = td - p.rd;
  return fmaf(k.k_d, fabsf(ad), k.k_hv * (fabsf(ah) + fabsf(av)));
}

// denominator: sum |rf o|^3 = rf^3 sum |o|^3 -- the CSF factor is applied to the wave's sum.  den: {h, v, d}
__device__ __forceinline__ void den_accumulate(const f2 bh, const f2 bv, const f2 bd, float* den) {
  const float oh = bh.x, ov = bv.x, od = bd.x;
  den[0] = fmaf(oh * oh, fabsf(oh), den[0]);
  den[1] = fmaf(ov * ov, fabsf(ov), den[1]);
  den[2] = fmaf(od * od, fabsf(od), den[2]);
}

// finish the coefficient whose pending values are p: threshold = 3x3 box of the masking signal + centre.  A pending set of
// zeros contributes exactly 0: x = max(-thr, 0) with thr >= 0.  num: {h, v, d}
__device__ __forceinline__ void finish(const Pending& p, const float s2 /* sums of the row above and of the row itself */,
                                       const float rs_below, const Consts& k, float* num) {
  const float thr = (s2 + rs_below) + p.g;
  const float xh = fmaxf(fmaf(fabsf(p.rh), k.rf_hv, -thr), 0.0f);
  const float xv = fmaxf(fmaf(fabsf(p.rv), k.rf_hv, -thr), 0.0f);
  const float xd = fmaxf(fmaf(fabsf(p.rd), k.rf_d, -thr), 0.0f);
  num[0] = fmaf(xh * xh, xh, num[0]);
  num[1] = fmaf(xv * xv, xv, num[1]);
  num[2] = fmaf(xd * xd, xd, num[2]);
}

}  // namespace admc
}  // namespace pqa
