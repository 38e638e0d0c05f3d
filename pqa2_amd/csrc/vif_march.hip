// VIF scale 0 with BOTH filter passes on the f16 matrix cores, no LDS, no workgroup barriers ("march" kernel).
//
// Arithmetic: libvmaf's float extractor (vif.c compute_vif, vif_tools.c vif_filter1d_s / _sq_s / _xy_s, vif_dec2_s,
// vif_statistic_s) -- the code behind the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated
// in oracle/vmaf_oracle.c.  The separable 17-tap filter commutes, so the HORIZONTAL pass runs first here.
//
// Why.  vif_s0_mfma_kernel (vif.hip) put the vertical pass on the matrix pipe and left the horizontal pass as 340 packed
// FP32 FMAs per wave and tile, 45 % of its VALU issue, with the two passes meeting in LDS (24-27 % bank conflicts on its
// 8-byte accumulator stores, two barriers per tile pair).  Here the second pass is an MFMA as well, and its operand comes
// straight out of the first pass's accumulators:
//
//   pass 1:          D1[row][out col] = sum_k  X[row][in col k] * T1[k][out col]      X = exact integer digit planes (A operand)
//   pass 2:          D2[col][out row] = sum_k  P[col][in row k] * T2[k][out row]      P = D1 split into two f16 pieces
//
// v_mfma_f32_16x16x32_f16 keeps D[m][n] of lane l = (n = l & 15, m = 4 (l >> 4) + i): four ROWS of one column after pass 1.
// The A operand of pass 2 wants, from the same lane, eight K elements of row m = l & 15 -- and K may be enumerated in any
// order as long as the tap matrix T2 uses the same one.  So K group (l >> 4) of pass 2 is DEFINED as {rows 4 (l >> 4) + i of
// the previous 16-row block, rows 4 (l >> 4) + i of the current one}: the lane's own accumulators, converted, ARE its operand.
// No transposition, no LDS, no cross-lane traffic; a wave is independent of every other wave.
//
// A wave owns a 16-column stripe and MARCHES down it in 16-row blocks: load 16 x 32 samples (one 8-byte load per lane and
// image), pass 1, split, pass 2 against the previous block's pieces, statistic on the 16 x 16 outputs, next block.  The
// 32-row window of the vertical filter is the pair (previous block, current block), so the only redundant work is the first
// block of a segment ((L + 1) / L on pass 1 for a segment of L blocks).
//
// Pass 1: the five signals r', d', r'^2, d'^2, r'd' (r' = r - 128) enter EXACTLY, as base-256 digit planes whose 16-bit
// patterns are their own f16 encodings (k * 2^-24); every f32 tap enters as two f16 pieces = 22 bits of it (the third piece,
// at most 2^-22 of the tap, was carried until round 4 -- 7 of 45 MFMAs per block for a difference of 4e-10 in the records,
// three orders of magnitude inside the kernel's and the f32 oracle's own distance from f64: profiles/r06w_tap_pieces_ab.txt);
// products are exact in f32 and the accumulator is f32.
// Pass 2 splits each f32 result v into hi = f16(v), lo = f16(v - hi) (both round-to-nearest: |v - hi - lo| <= 2^-23 |v|, the
// size of ONE f32 rounding) and each tap c * 2^8 into c1 = f16(.), c2 = f16(. - c1); it accumulates hi c1 + hi c2 + lo c1 in
// f32 (the dropped lo c2 is below 2^-24 of the term).  That is an f32-grade evaluation in a different rounding order, not a
// reduced-precision one: tools/sim_split_horizontal.py measured the same distance from an f64 evaluation as an all-f32
// pass in libvmaf's order (8e-8 on the scale-0 numerator), tests/test_gpu_*.py hold it to the same bars as before.
//
// Units.  Pass 2's taps carry 2^8 (so that c2 is a normal f16), the means enter it as natural / 16 and the squares as
// natural: the means come out times 16 and the squares times 256, i.e. everything is consistently in units U = 16: the
// statistic runs on those numbers with sigma_nsq, eps and 1 / sigma_max scaled by 256 (g and both log arguments are
// scale-free), no rescaling instruction anywhere.
//
// The next scale's input (9-tap filter, even rows and columns: vif_dec2_s) rides along: pass 1 puts the eight even output
// columns of ref and of dis into ONE accumulator (tap matrices that are zero on the other image's eight N slots), pass 2
// produces the eight even rows; lanes 0..7 of each 16-lane row then hold four consecutive samples of a half-resolution row
// and store them as one 16-byte write.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "march_common.h"
#include "pqa_device.h"

namespace pqa {

namespace {

using namespace march;

// fragments of the per-lane tap table (each: 64 lanes x 8 f16)
enum : int {
  F_HI = 0,    // 0..2   pass 1, c * 2^19 in pieces (digits that weigh 2^8); the kernel multiplies the first PQA_MARCH_TAP_PIECES
  F_LO = 3,    // 3..5   pass 1, c * 2^11 in pieces (means and low digits)
  F_DR = 6,    // 6..8   pass 1, next scale's taps c' * 2^18, even output columns of REF in N slots 0..7
  F_DD = 9,    // 9..11  the same for DIS in N slots 8..15
  F_V = 12,    // 12,13  pass 2, c * 2^8 in two pieces, K order {previous block rows, current block rows}
  F_VD = 14,   // 14,15  pass 2 of the next scale's input, c' * 2^8, even output rows in N slots 0..7
  F_W = 16,    // 16,17  as F_V with K order {current block rows, previous block rows}
  F_WD = 18,   // 18,19  as F_VD in that order
  F_L9 = 20,   // 20,21  pass 1 of 10-bit clips, c * 2^9 in two pieces (low digits base 1024)
  F_L8 = 22,   // 22,23  pass 1 of 12-bit clips, c * 2^8 in two pieces (low digits base 2048)
  kMarchFrags = 24
};

struct MarchArgs {
  const void* ref;
  const void* dis;
  unsigned pitch_r, pitch_d;               // bytes
  int64_t frame_pitch_r, frame_pitch_d;    // bytes
  int w, h, fold_w, fold_h;
  int aligned;                             // bases and pitches allow one 8-sample load per lane (8 / 16 bytes)
  float gain_limit;
  double* partials;                        // [n_frames][n_part][2]
  int n_part;
  float* dst_ref;
  float* dst_dis;
  unsigned dst_pitch_r, dst_pitch_d;       // floats
  int64_t dst_frame_pitch_r, dst_frame_pitch_d;
  int n_cb, n_cbg, seg_blocks, n_seg, row_blocks;
  float mean_off, dec_off;                 // -128 * sum(17 taps) / 16 and -128 * sum(9 taps): the mid-grey term of the planes filtered as samples
  const uint4* tab;
};

// The pieces of TWO consecutive 16-row blocks after pass 1 (rows 4 (lane >> 4) + i of column (lane & 15)), laid out as the
// pass-2 operands themselves: dwords {0, 1} of each vector belong to the even block, {2, 3} to the odd one.  Plane 5 is
// the next scale's input (N slot (lane & 15) = even column of ref (0..7) / dis (8..15)).
struct Pieces {
  u4v hi[6], lo[6];
};

// vif_statistic_s on two horizontally adjacent pixels, in units U = 16 (see the file comment): accumulates the low-branch
// sums and the three log products.  Same algebra as vif_hstat (vif.hip), which documents each override that drops out.
struct StatAcc {
  f2 num2, den2, pn, qn, pd;
};
__device__ __forceinline__ void stat_pair(StatAcc& s, const f2 mu1, const f2 mu2, const f2 xx, const f2 yy, const f2 xy,
                                          const bool v0, const bool v1, const float gain_limit) {
  const float sigma_nsq = 2.0f * 256.0f, eps = 1.0e-10f * 256.0f, sigma_max_inv = 4.0f / (255.0f * 255.0f * 256.0f);
  const f2 s1 = xx - mu1 * mu1;
  f2 s2 = yy - mu2 * mu2;
  const f2 s12 = xy - mu1 * mu2;
  s2 = f2{fmaxf(s2.x, 0.0f), fmaxf(s2.y, 0.0f)};
  // one compare per pixel ("low" = sigma1_sq < sigma_nsq; every input is a finite filter output, so no NaN case); the
  // branch masks are lane masks in SGPRs, combined with the validity masks by scalar instructions
  const bool lt0 = s1.x < sigma_nsq, lt1 = s1.y < sigma_nsq;
  const bool hx = v0 && !lt0, hy = v1 && !lt1;
  const bool lx = v0 && lt0, ly = v1 && lt1;
  const f2 s1h = f2{hx ? s1.x : 0.0f, hy ? s1.y : 0.0f};
  const f2 gden = s1h + f2{eps, eps};
  const f2 grcp = f2{fast_rcp(gden.x), fast_rcp(gden.y)};
  f2 g = s12 * grcp;
  g = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, gden, s12), grcp, g);
  f2 sv = s2 - g * s12;
  sv = f2{fmaxf(sv.x, eps), fmaxf(sv.y, eps)};
  g = f2{__builtin_amdgcn_fmed3f(g.x, 0.0f, gain_limit), __builtin_amdgcn_fmed3f(g.y, 0.0f, gain_limit)};
  const f2 svn = sv + f2{sigma_nsq, sigma_nsq};
  const f2 narg = __builtin_elementwise_fma(g * g, s1h, svn);
  const f2 darg = __builtin_elementwise_fma(s1h, f2{1.0f / sigma_nsq, 1.0f / sigma_nsq}, f2{1.0f, 1.0f});
  const f2 low = __builtin_elementwise_fma(s2, f2{-sigma_max_inv, -sigma_max_inv}, f2{1.0f, 1.0f});
  s.pn *= narg;
  s.qn *= svn;
  s.pd *= darg;
  const f2 wl = f2{lx ? 1.0f : 0.0f, ly ? 1.0f : 0.0f};
  s.num2 = __builtin_elementwise_fma(wl, low, s.num2);
  s.den2 += wl;
}

// PQA_MARCH_LDS_TABLES: the tap fragments of the next scale's input (used once per block each) stay in LDS and are read
// where they are used instead of occupying 40 VGPRs for the whole march; with them out of the way the kernel fits three
// waves per SIMD (PQA_MARCH_OCC).
#ifndef PQA_MARCH_LDS_TABLES
#define PQA_MARCH_LDS_TABLES 1
#endif
// PQA_MARCH_TAP_PIECES: f16 pieces per pass-1 tap on the means, the high digits and the next scale's input (the low digits
// always took two): 2 = 22 bits of every f32 tap (38 MFMAs per block); 3 = the tap exactly (45; the test partner: VIF chain
// -4.2 %, records 4e-10 apart)
#ifndef PQA_MARCH_TAP_PIECES
#define PQA_MARCH_TAP_PIECES 2
#endif
#ifndef PQA_MARCH_OCC
#define PQA_MARCH_OCC (PQA_MARCH_LDS_TABLES ? 3 : 2)
#endif
// S = uint8_t: 8-bit samples as described at the top.  S = uint16_t: 10-bit samples (libvmaf: x = v / 4 - 128 = (v - 512) / 4):
// a sample (<= 1023 < 2048) is again its own f16 pattern, so the mean planes take the loaded dwords as they are; the squares
// and the cross term of v - 512 (|.| <= 512, products of up to 19 bits) come from 32-bit multiplies and are split into
// base-1024 digits (hi <= 512, lo < 1024); the low digit planes use pieces of c * 2^9.  Every plane then leaves pass 1 as
// 2^-13 (means) or 2^-15 (squares) of its integer sum, and the sample scale (1/4 on the means, 1/16 on the squares) joins
// the split's factor.  12-bit clips: the B12 instance below.
// B12 (with S = uint16_t): 12-bit samples (x = v / 16 - 128 = (v - 2048) / 16).  A sample is no longer its own f16 pattern (k <=
// 2048 only), so the mean planes take the CENTRED sample v - 2048 as a sign-magnitude pattern (|.| <= 2048; the mid-grey
// terms of the means and of the next scale's input are then zero); squares and cross term of v - 2048 (up to 2^22) are split
// into base-2048 digits: the high digit of a square is in [0, 2048], the cross term's in [-2047, 2048] (floor), again a
// sign-magnitude pattern; the low digit planes use pieces of c * 2^8.
template <typename S, bool B12 = false>
__global__ __launch_bounds__(kBlock, PQA_MARCH_OCC) void vif_s0_march_kernel(const MarchArgs a) {
  constexpr bool W16 = sizeof(S) == 2;
  static_assert(W16 || !B12, "12-bit samples are 16-bit elements");
  constexpr int ES = (int)sizeof(S);
  const int tid = threadIdx.x;
#if PQA_MARCH_LDS_TABLES
  // fragments F_DR .. F_DD + 2 (6), F_VD, F_VD + 1, F_WD, F_WD + 1 -> LDS slots 0..9
  __shared__ uint4 stab[10 * 64];
  for (int i = tid; i < 10 * 64; i += kBlock) {
    const int f = i >> 6;
    const int src = f < 6 ? F_DR + f : f < 8 ? F_VD + (f - 6) : F_WD + (f - 8);
    stab[i] = a.tab[src * 64 + (i & 63)];
  }
  __syncthreads();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int id = xcd_remap(blockIdx.x, a.n_cbg * a.n_seg);
  const int cbg = id % a.n_cbg, seg = id / a.n_cbg;
  const int cb = cbg * 4 + wave;   // this wave's 16-column stripe
  const int fr = blockIdx.y;
  double* part = a.partials + ((int64_t)fr * a.n_part + (int64_t)(seg * a.n_cbg + cbg) * 4 + wave) * 2;
  if (cb >= a.n_cb) {   // wave-uniform: the last workgroup of a row of stripes may have idle waves; their slots read 0
    if (lane == 0) { part[0] = 0.0; part[1] = 0.0; }
    return;
  }
  const int x0 = cb * 16;
  const int rb0 = seg * a.seg_blocks;
  const int n_out = min(a.seg_blocks, a.row_blocks - rb0);   // 16-row output blocks of this segment (>= 1)
  const int ys = rb0 * 16;
  const int m = lane & 15, kq = lane >> 4;

  const uint8_t* __restrict__ ref = (const uint8_t*)a.ref + (int64_t)fr * a.frame_pitch_r;   // pitches are in bytes
  const uint8_t* __restrict__ dis = (const uint8_t*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const auto rsrc_r = make_rsrc(ref, (unsigned)a.h * a.pitch_r);
  const auto rsrc_d = make_rsrc(dis, (unsigned)a.h * a.pitch_d);
  // all 32 input columns of the stripe inside the image and 8-byte loads allowed: one load per lane and image
  const bool colfast = a.aligned && x0 - 8 >= 0 && x0 + 24 <= a.w;   // wave-uniform
  const int xin = x0 - 8 + 8 * kq;                                    // this lane's first input column (samples)

  // tap-matrix fragments of this lane
  h8 T[kMarchFrags];
#pragma unroll
  for (int f = 0; f < kMarchFrags; ++f) {
    const bool in_lds = PQA_MARCH_LDS_TABLES && ((f >= F_DR && f < F_DR + 6) || (f >= F_VD && f < F_VD + 2) || (f >= F_WD && f < F_WD + 2));
    const bool third = f == F_HI + 2 || f == F_LO + 2 || f == F_DR + 2 || f == F_DD + 2;
    const bool needed = f < F_L9 || (f < F_L8 ? (W16 && !B12) : B12);   // F_L9: 10-bit clips only; F_L8: 12-bit clips only
    const bool unused = !needed || (PQA_MARCH_TAP_PIECES < 3 && third);
    if (!in_lds && !unused) T[f] = __builtin_bit_cast(h8, a.tab[f * 64 + lane]);
  }
#if PQA_MARCH_LDS_TABLES
  // a fragment kept in LDS, read where it is used.  The lane's byte offset goes through an empty asm once per pass (tab_off,
  // refreshed by TD_REFRESH): the reads of a pass can then be scheduled together and early, but not hoisted out of the
  // march (which would put every fragment back into 4 VGPRs for the whole loop)
  int tab_off = lane * 16;
  const auto lds_frag = [&](const int slot) {
    return __builtin_bit_cast(h8, *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(stab) + tab_off + slot * 1024));
  };
#define TD_REFRESH() asm volatile("" : "+v"(tab_off))
#define TD(f) lds_frag((f) < F_VD ? (f) - F_DR : (f) < F_W ? 6 + (f) - F_VD : 8 + (f) - F_WD)
#else
#define TD_REFRESH()
#define TD(f) T[f]
#endif

  // ---- loads: 8 consecutive samples of row (block row m) per image ----------------------------------------------
  // A block whose 16 rows lie inside the image needs no per-lane row arithmetic at all: the lane part of the address
  // (m * pitch + first column) is fixed for the whole march and the block's first row goes into the load's SCALAR offset.
  const unsigned lane_off_r = (unsigned)m * a.pitch_r + (unsigned)(xin * ES), lane_off_d = (unsigned)m * a.pitch_d + (unsigned)(xin * ES);
  // R / D: the lane's 8 samples, packed as they lie in memory (8 bit: two dwords; 10 bit: four).  The type is exactly as
  // wide as the data: these registers are live across the whole block (they are the prefetch), and two spare dwords per
  // plane were enough to push the 8-bit kernel into spilling its load offsets -- whose reloads then serialised the prefetch.
  using Raw = std::conditional_t<W16, u4v, u2v>;
  const auto load8 = [&](const rsrc_t rs, const unsigned voff, const unsigned soff) -> Raw {
    if constexpr (W16) return __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    else return __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
  };
  const auto load_block = [&](int rb, Raw& R, Raw& D) {
    const int y_first = ys - 8 + 16 * rb;                                 // wave-uniform
    const bool rows_in = y_first >= 0 && y_first + 16 <= a.h;
    if (colfast && rows_in) {
      R = load8(rsrc_r, lane_off_r, (unsigned)y_first * a.pitch_r);
      D = load8(rsrc_d, lane_off_d, (unsigned)y_first * a.pitch_d);
      return;
    }
    const unsigned my = (unsigned)mirror_fold(y_first + m, a.h, a.fold_h);
    if (colfast) {
      R = load8(rsrc_r, my * a.pitch_r + (unsigned)(xin * ES), 0);
      D = load8(rsrc_d, my * a.pitch_d + (unsigned)(xin * ES), 0);
    } else {   // stripes at the left / right image edge (mirrored columns) or unaligned planes: sample by sample
      unsigned r[8], d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned mx = (unsigned)mirror_fold(xin + j, a.w, a.fold_w) * ES;
        if (W16) {
          r[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, my * a.pitch_r + mx, 0, 0);
          d[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, my * a.pitch_d + mx, 0, 0);
        } else {
          r[j] = __builtin_amdgcn_raw_buffer_load_b8(rsrc_r, my * a.pitch_r + mx, 0, 0) & 0xffu;
          d[j] = __builtin_amdgcn_raw_buffer_load_b8(rsrc_d, my * a.pitch_d + mx, 0, 0) & 0xffu;
        }
      }
      if constexpr (W16) {
        R = u4v{r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16)};
        D = u4v{d[0] | (d[1] << 16), d[2] | (d[3] << 16), d[4] | (d[5] << 16), d[6] | (d[7] << 16)};
      } else {
        R = u2v{r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24), r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24)};
        D = u2v{d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24), d[4] | (d[5] << 8) | (d[6] << 16) | (d[7] << 24)};
      }
    }
  };

  // ---- pass 1 + split: one 16 x 32 input block -> this lane's pieces -----------------------------------------------
  const auto pass1 = [&](const Raw R, const Raw D, Pieces& P, auto half) {
    constexpr int H = decltype(half)::value;
    TD_REFRESH();
    const f4 z = f4{0.0f, 0.0f, 0.0f, 0.0f};
    // 16-bit lanes {col 2v, col 2v+1} of this lane's 8 columns.  8 bit: byte -> zero-extended half (one v_perm_b32 each);
    // 10 bit: the loaded dwords as they are.  r16 / d16: the same minus mid-grey, as signed 16-bit integers.
    constexpr short MID = W16 ? (B12 ? 2048 : 512) : 128;
    // two's-complement halves with |.| <= 2048 -> sign-magnitude f16 patterns (+-k * 2^-24)
    const auto sign_mag = [](const unsigned x) -> unsigned {
      const s2v v = __builtin_bit_cast(s2v, x), n = s2v{0, 0} - v;
      return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, n)) | (x & 0x80008000u);
    };
    unsigned ru[4], du[4], r16[4], d16[4], r16s[4], d16s[4], r16t[4], d16t[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if constexpr (W16) {
        ru[v] = R[v];
        du[v] = D[v];
      } else {
        const unsigned sel = (v & 1) ? 0x0c030c02u : 0x0c010c00u;
        ru[v] = __builtin_amdgcn_perm(0u, R[v >> 1], sel);
        du[v] = __builtin_amdgcn_perm(0u, D[v >> 1], sel);
      }
      r16[v] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s2v, ru[v]) - s2v{MID, MID});
      d16[v] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s2v, du[v]) - s2v{MID, MID});
      if constexpr (W16 && !B12) {   // 64 (v - 512) in [-2^15, 2^15): the second factor of every product, see below
        r16s[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, r16[v]) << s2v{6, 6}));
        d16s[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, d16[v]) << s2v{6, 6}));
      }
      if constexpr (B12) {           // 12 bit: 8 (v - 2048) as the second factor and 4 (v - 2048) as the first (x 32 together)
        r16s[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, r16[v]) << s2v{3, 3}));
        d16s[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, d16[v]) << s2v{3, 3}));
        r16t[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, r16[v]) << s2v{2, 2}));
        d16t[v] = __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, d16[v]) << s2v{2, 2}));
        ru[v] = sign_mag(r16[v]);    // the mean planes' operand: the centred sample
        du[v] = sign_mag(d16[v]);
      }
    }
    f4 Dh[5], Dd;
    {  // means and the next scale's input: the SAMPLES themselves (k * 2^-24); the mid-grey term 128 * sum(taps) is a constant
       // of the filter (mirrored borders keep all 17 / 9 taps inside the window) and comes off in the split's fma
      const h8 A = frag4(ru[0], ru[1], ru[2], ru[3]);
      Dh[0] = mma(A, T[F_LO], z); Dd = mma(A, TD(F_DR), z);
      Dh[0] = mma(A, T[F_LO + 1], Dh[0]); Dd = mma(A, TD(F_DR + 1), Dd);
      if (PQA_MARCH_TAP_PIECES == 3) { Dh[0] = mma(A, T[F_LO + 2], Dh[0]); Dd = mma(A, TD(F_DR + 2), Dd); }
    }
    {
      const h8 A = frag4(du[0], du[1], du[2], du[3]);
      Dh[1] = mma(A, T[F_LO], z); Dd = mma(A, TD(F_DD), Dd);
      Dh[1] = mma(A, T[F_LO + 1], Dh[1]); Dd = mma(A, TD(F_DD + 1), Dd);
      if (PQA_MARCH_TAP_PIECES == 3) { Dh[1] = mma(A, T[F_LO + 2], Dh[1]); Dd = mma(A, TD(F_DD + 2), Dd); }
    }
#pragma unroll
    for (int s = 2; s < 5; ++s) {   // r'^2, d'^2, r'd': exact integer products, split into two digits
      // The digits of one plane are extracted right before the MFMAs that consume them -- low digit, its two MFMAs, high
      // digit, its three: the extraction of the next operand then issues under the matrix instructions of the previous one
      // (forming all eight registers first and then five dependent MFMAs back to back cost the 8-bit kernel 10 %).
      unsigned q[4], p0[4], p1[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const unsigned xs = B12 ? (s == 3 ? d16t[v] : r16t[v]) : (s == 3 ? d16[v] : r16[v]), ys = s == 2 ? r16[v] : d16[v];
        if (!W16) {   // 16-bit products; the cross term is signed: + 64 * 256 makes both digits unsigned (one
                      // v_pk_mad_u16), the 64 comes off below
          const s2v x = __builtin_bit_cast(s2v, xs), y = __builtin_bit_cast(s2v, ys);
          q[v] = __builtin_bit_cast(unsigned, s == 4 ? (s2v)(x * y + s2v{0x4000, 0x4000}) : (s2v)(x * y));
        } else {      // |v - 512| <= 512: 32-bit products of the sign-extended halves; the cross term gets 256 * 1024 added
                      // so that its high digit (in [0, 512]) is unsigned like the squares'.  The products are formed times 64
                      // (second factor pre-shifted, |.| <= 2^24): the base-1024 digits of x y then sit at bits 16.. and 6..15
                      // of the register -- halfword-aligned, so that one v_perm_b32 packs the high digits of two pixels and
                      // one v_perm_b32 + one packed shift the low ones (three shift / mask operations each before)
          const unsigned yss = s == 2 ? r16s[v] : d16s[v];
          const int x0_ = (int)(short)(xs & 0xffffu), x1_ = (int)xs >> 16;
          const int y0_ = (int)(short)(yss & 0xffffu), y1_ = (int)yss >> 16;
          const int add = (s == 4 && !B12) ? (256 << 16) : 0;   // (12 bit: the cross term's high digit stays signed)
          p0[v] = (unsigned)(x0_ * y0_ + add);
          p1[v] = (unsigned)(x1_ * y1_ + add);
        }
      }
      unsigned t[4];
      {  // low digit: byte 0 of each 16-bit product / the low 10 bits of each 32-bit product
#pragma unroll
        for (int v = 0; v < 4; ++v)
          t[v] = W16 ? __builtin_bit_cast(unsigned, (us2v)(__builtin_bit_cast(us2v, __builtin_amdgcn_perm(p1[v], p0[v], 0x05040100u)) >> us2v{B12 ? 5 : 6, B12 ? 5 : 6}))
                     : __builtin_amdgcn_perm(0u, q[v], 0x0c020c00u);
        const h8 A = frag4(t[0], t[1], t[2], t[3]);
        Dh[s] = mma(A, T[B12 ? F_L8 : W16 ? F_L9 : F_LO], z);
        Dh[s] = mma(A, T[B12 ? F_L8 + 1 : W16 ? F_L9 + 1 : F_LO + 1], Dh[s]);
      }
      {  // high digit: byte 1 / bits 10.. ; the cross term's offset (64 / 256) comes off in f16, exactly
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          t[v] = W16 ? __builtin_amdgcn_perm(p1[v], p0[v], 0x07060302u) : __builtin_amdgcn_perm(0u, q[v], 0x0c030c01u);
          if (s == 4) t[v] = B12 ? sign_mag(t[v]) : tiny_minus(t[v], W16 ? 0x8100 : 0x8040);
        }
        const h8 A = frag4(t[0], t[1], t[2], t[3]);
        Dh[s] = mma(A, T[F_HI], Dh[s]);
        Dh[s] = mma(A, T[F_HI + 1], Dh[s]);
        if (PQA_MARCH_TAP_PIECES == 3) Dh[s] = mma(A, T[F_HI + 2], Dh[s]);
      }
    }
    // pass 1 leaves every signal of an 8-bit clip times 2^-13 and the next scale's input times 2^-6 (operands k * 2^-24,
    // pieces of c * 2^11, c * 2^19 on the digits that weigh 2^8, c' * 2^18); for 10-bit clips the means again times 2^-13
    // and the squares times 2^-15 (base-1024 digits, low pieces c * 2^9), both still in sample units (x 4 and x 16).
    // Into pass 2: means as natural / 16, squares as natural, the next scale's input as natural.
    // The means were filtered as samples: sum c (x - 128) = sum c x - 128 sum c, with sum c the sum of the taps AS APPLIED (the pieces).
    constexpr float KM = B12 ? 32.0f : W16 ? 128.0f : 512.0f, KS = B12 ? 256.0f : W16 ? 2048.0f : 8192.0f, KD = B12 ? 4.0f : W16 ? 16.0f : 64.0f;
    split4<H>(Dh[0], KM, a.mean_off, P.hi[0], P.lo[0]);
    split4<H>(Dh[1], KM, a.mean_off, P.hi[1], P.lo[1]);
    split4<H>(Dh[2], KS, 0.0f, P.hi[2], P.lo[2]);
    split4<H>(Dh[3], KS, 0.0f, P.hi[3], P.lo[3]);
    split4<H>(Dh[4], KS, 0.0f, P.hi[4], P.lo[4]);
    split4<H>(Dd, KD, a.dec_off, P.hi[5], P.lo[5]);
  };

  double dnum = 0.0, dden = 0.0;
  // validity of this lane's four output columns (wave-uniform per K group, constant over the march)
  bool vcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vcol[i] = x0 + 4 * kq + i < a.w;
  const int ow = a.w >> 1, oh = a.h >> 1;

  // ---- pass 2 + next-scale store + statistic for output rows yo .. yo + 15: the window is the pair of blocks in P;
  // FV / FVD = the tap fragments whose K order matches which half holds the OLDER block
  const auto pass2 = [&](const Pieces& P, const int FV, const int FVD, const int yo) {
    TD_REFRESH();
    const f4 z = f4{0.0f, 0.0f, 0.0f, 0.0f};
    f4 V[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const h8 Ah = __builtin_bit_cast(h8, P.hi[s]), Al = __builtin_bit_cast(h8, P.lo[s]);
      V[s] = mma(Ah, T[FV], z);
      V[s] = mma(Ah, T[FV + 1], V[s]);
      V[s] = mma(Al, T[FV], V[s]);
    }
    f4 Vd;
    {
      const h8 Ah = __builtin_bit_cast(h8, P.hi[5]), Al = __builtin_bit_cast(h8, P.lo[5]);
      const h8 t0 = TD(FVD), t1 = TD(FVD + 1);
      Vd = mma(Ah, t0, z);
      Vd = mma(Ah, t1, Vd);
      Vd = mma(Al, t0, Vd);
    }
    // next scale's input: lanes m < 8 hold 4 consecutive samples of half-resolution row yo / 2 + m
    if (m < 8) {
      const int orow = (yo >> 1) + m, oc = (x0 >> 1) + 4 * (kq & 1);
      if (orow < oh && oc < ow) {
        float* __restrict__ dst = kq < 2 ? a.dst_ref + (int64_t)fr * a.dst_frame_pitch_r + (int64_t)orow * a.dst_pitch_r + oc
                                         : a.dst_dis + (int64_t)fr * a.dst_frame_pitch_d + (int64_t)orow * a.dst_pitch_d + oc;
        const f4 o = Vd * f4{1.0f / 256.0f, 1.0f / 256.0f, 1.0f / 256.0f, 1.0f / 256.0f};
        if (oc + 4 <= ow) {
          *reinterpret_cast<f4*>(dst) = o;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (oc + i < ow) dst[i] = o[i];
        }
      }
    }
    // statistic on this lane's 4 pixels (row yo + m, columns x0 + 4 kq + i)
    const bool vrow = yo + m < a.h;
    StatAcc st{f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{1.0f, 1.0f}, f2{1.0f, 1.0f}, f2{1.0f, 1.0f}};
    stat_pair(st, f2{V[0][0], V[0][1]}, f2{V[1][0], V[1][1]}, f2{V[2][0], V[2][1]}, f2{V[3][0], V[3][1]},
              f2{V[4][0], V[4][1]}, vrow && vcol[0], vrow && vcol[1], a.gain_limit);
    stat_pair(st, f2{V[0][2], V[0][3]}, f2{V[1][2], V[1][3]}, f2{V[2][2], V[2][3]}, f2{V[3][2], V[3][3]},
              f2{V[4][2], V[4][3]}, vrow && vcol[2], vrow && vcol[3], a.gain_limit);
    // the log terms of the lane's four pixels as ONE log per product: in units U = 16 every factor lies in [1, 2^24) (narg <=
    // sigma2_sq + sv_sq + sigma_nsq by Cauchy-Schwarz, each <= 2^22), four of them below 2^96 -- three v_log_f32 instead of six
    const float num = (st.num2.x + st.num2.y) + (fast_log2(st.pn.x * st.pn.y) - fast_log2(st.qn.x * st.qn.y));
    const float den = (st.den2.x + st.den2.y) + fast_log2(st.pd.x * st.pd.y);
    dnum += (double)num;
    dden += (double)den;
  };

  // Blocks alternate between the two halves of the operand vectors: the even blocks live in dwords {0, 1}, the odd ones in
  // {2, 3}, so the pass-2 operand is always the vector as it stands -- four consecutive registers per plane that are never
  // copied.  Which half is the older block flips every step, so the tap fragments exist in both K orders (F_V / F_VD:
  // {older, newer}; F_W / F_WD: {newer, older}).
  Pieces P;
#pragma unroll
  for (int i = 0; i < 6; ++i) { P.hi[i] = u4v{0u, 0u, 0u, 0u}; P.lo[i] = u4v{0u, 0u, 0u, 0u}; }
  const std::integral_constant<int, 0> even{};
  const std::integral_constant<int, 1> odd{};
  Raw Rn, Dn;
  load_block(0, Rn, Dn);
  {
    const Raw Rc = Rn, Dc = Dn;
    load_block(1, Rn, Dn);   // n_out >= 1: block 1 always exists
    pass1(Rc, Dc, P, even);
  }
  for (int rb = 1; rb <= n_out; rb += 2) {
    {   // odd block -> upper half; window (lower half older, upper half newer)
      const Raw Rc = Rn, Dc = Dn;
      if (rb < n_out) load_block(rb + 1, Rn, Dn);   // in flight while this block is computed
      pass1(Rc, Dc, P, odd);
      pass2(P, F_V, F_VD, ys + 16 * (rb - 1));
    }
    if (rb + 1 <= n_out) {   // even block -> lower half; window (lower half newer, upper half older)
      const Raw Rc = Rn, Dc = Dn;
      if (rb + 1 < n_out) load_block(rb + 2, Rn, Dn);
      pass1(Rc, Dc, P, even);
      pass2(P, F_W, F_WD, ys + 16 * rb);
    }
  }
  dnum = wave_sum(dnum);
  dden = wave_sum(dden);
  if (lane == 0) { part[0] = dnum; part[1] = dden; }
}

// The operand encoding leans on f16 denormals surviving v_pk_add_f16 and the MFMA operands.  gfx950 keeps them (measured:
// tools/ubench/f16_denorm.hip), but it is a property of the device and of the kernel's float mode, so every process checks it
// once before the kernel is allowed to run: column n of B holds the pattern k = 67 n + 1 (minus 1 via the packed add), A is
// all ones -> D[.][n] = 32 (k - 1) 2^-24 exactly.  ok[0] counts the lanes that saw that.
__global__ void f16_tiny_probe_kernel(int* ok) {
  const int lane = threadIdx.x, n = lane & 15;
  const unsigned k = ((unsigned)n * 67u + 1u) & 0x3ffu;
  const unsigned t = tiny_minus(k | (k << 16), 0x8001);   // (k - 1) * 2^-24 in both halves
  const h8 b = frag4(t, t, t, t);
  h8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
  const f4 d = mma(a, b, f4{0.0f, 0.0f, 0.0f, 0.0f});
  const float want = 32.0f * (float)(k - 1u) * 5.9604644775390625e-08f;   // 2^-24
  if (d[0] == want && d[3] == want) atomicAdd(ok, 1);
}

// ---- host: the per-lane tap-matrix fragments ------------------------------------------------------------------------
bool build_table(uint16_t* out /* [kMarchFrags][64][8] */) {
  float c17[17], c9[9];
  gaussian(17, c17);
  gaussian(9, c9);
  bool exact = true;
  const size_t stride = 64 * 8;
  for (int lane = 0; lane < 64; ++lane) {
    const int n = lane & 15, kb = lane >> 4;
    for (int j = 0; j < 8; ++j) {
      uint16_t* o = out + (size_t)lane * 8 + j;
      // pass 1 (B operand): K element j of group kb = window column wc; N slot n = output column n, window column n + 8
      const int wc = 8 * kb + j;
      const int t = wc - n;
      const double c = (t >= 0 && t <= 16) ? (double)c17[t] : 0.0;
      if (pieces(c * 524288.0, 3, o + F_HI * stride, stride) != 0.0) exact = false;
      if (pieces(c * 2048.0, 3, o + F_LO * stride, stride) != 0.0) exact = false;
      pieces(c * 512.0, 2, o + F_L9 * stride, stride);   // the 10-bit low digits weigh 2^-10 of the signal: 22 bits of the tap
      pieces(c * 256.0, 2, o + F_L8 * stride, stride);   // the 12-bit low digits (base 2048)
      // next scale: N slot n = even column 2 (n & 7) (window column 2 (n & 7) + 8) of ref (n < 8) or dis (n >= 8)
      const int t9 = wc - (2 * (n & 7) + 4);
      const double cd = (t9 >= 0 && t9 <= 8) ? (double)c9[t9] : 0.0;
      if (pieces(n < 8 ? cd * 262144.0 : 0.0, 3, o + F_DR * stride, stride) != 0.0) exact = false;
      if (pieces(n >= 8 ? cd * 262144.0 : 0.0, 3, o + F_DD * stride, stride) != 0.0) exact = false;
      // pass 2 (B operand): K element j of group kb = window row 4 kb + j of the previous block (j < 4) or
      // 16 + 4 kb + (j - 4) of the current one; N slot n = output row n, window row n + 8
      const int wr = j < 4 ? 4 * kb + j : 16 + 4 * kb + (j - 4);
      const int tv = wr - n;
      pieces((tv >= 0 && tv <= 16) ? (double)c17[tv] * 256.0 : 0.0, 2, o + F_V * stride, stride);
      const int tv9 = wr - (2 * n + 4);   // N slot n < 8 = even output row 2 n
      pieces((n < 8 && tv9 >= 0 && tv9 <= 8) ? (double)c9[tv9] * 256.0 : 0.0, 2, o + F_VD * stride, stride);
      // the same with the register halves swapped: K element j < 4 = current block, j >= 4 = previous block
      const int ws = j < 4 ? 16 + 4 * kb + j : 4 * kb + (j - 4);
      const int tw = ws - n, tw9 = ws - (2 * n + 4);
      pieces((tw >= 0 && tw <= 16) ? (double)c17[tw] * 256.0 : 0.0, 2, o + F_W * stride, stride);
      pieces((n < 8 && tw9 >= 0 && tw9 <= 8) ? (double)c9[tw9] * 256.0 : 0.0, 2, o + F_WD * stride, stride);
    }
  }
  return exact;
}

std::mutex g_tab_mu;
const uint4* g_tab[64] = {};

const uint4* device_tab() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_tab_mu);
  return g_tab[dev];
}

constexpr int kMinSegBlocks = 8;   // a segment repeats one block of pass 1: (L + 1) / L

}  // namespace

int vif_march_table(uint16_t* out, int capacity_halfwords) {
  const int need = kMarchFrags * 64 * 8;
  if (!out || capacity_halfwords < need) return -need;
  return build_table(out) ? kMarchFrags : 0;
}

// How a w x h frame is cut into waves (a function of the geometry only) and what a block costs on the matrix pipe:
// out = {16-column stripes, 16-row blocks, blocks per segment, segments, pass-1 MFMAs per block, pass-2 MFMAs per block}.
void vif_march_shape(int w, int h, int* out) {
  int n_cb, n_cbg, row_blocks, seg_blocks, n_seg;
  n_cb = (w + 15) / 16;
  n_cbg = (n_cb + 3) / 4;
  row_blocks = (h + 15) / 16;
  // segment length: long enough that the repeated first block is cheap, short enough that a launch of a full batch has
  // several waves per SIMD slot to balance.  A function of the GEOMETRY only: a frame's partial sums -- and so its record --
  // must not depend on how many frames share the launch.
  int seg = row_blocks;
#ifndef PQA_MARCH_WAVES_PER_FRAME
#define PQA_MARCH_WAVES_PER_FRAME 768    /* at 2160p; x 97 frames (the automatic batch) = 74 496 waves per launch.  1 536 while the batch
                                            was 32 (swept 6 144 .. 98 304 waves per launch); with 97 frames per launch 34-block segments beat
                                            17-block ones by 2.3 % on the VIF chain (profiles/r06i_launch_shapes_ab.txt) */
#endif
#ifndef PQA_MARCH_WAVES_SCALED
#define PQA_MARCH_WAVES_SCALED 1
#endif
  // PQA_MARCH_WAVES_SCALED: the target follows the frame's pixel count (the automatic batch keeps the BYTES per launch
  // constant, so a launch of smaller frames has as many waves with proportionally fewer per frame; never under 192 waves per
  // frame): 2160p and 1080p march 34-block segments (35 / 34 on pass 1).  With a fixed target 1080p had 8-block segments
  // (9 / 8): VIF chain +7 %, whole path +4.9 % at 1080p, +3 % / +1 % at 720p (profiles/r06c_march_segments_ab.txt)
  const int want_waves = PQA_MARCH_WAVES_SCALED
      ? (int)fmax(192.0, PQA_MARCH_WAVES_PER_FRAME * ((double)w * h) / (3840.0 * 2160.0)) : PQA_MARCH_WAVES_PER_FRAME;
  while (seg > kMinSegBlocks && n_cbg * 4 * ((row_blocks + seg - 1) / seg) < want_waves) seg = (seg + 1) / 2;
  if (seg < kMinSegBlocks) seg = row_blocks < kMinSegBlocks ? row_blocks : kMinSegBlocks;
  seg_blocks = seg;
  n_seg = (row_blocks + seg - 1) / seg;
  out[0] = n_cb; out[1] = row_blocks; out[2] = seg_blocks; out[3] = n_seg;
  out[4] = PQA_MARCH_TAP_PIECES == 3 ? 27 : 20;   // means 2 x P, low digits 3 x 2, high digits 3 x P, next scale's input 2 x P
  out[5] = 18;
}

int vif_march_partials_max(int w, int h) {
  const int n_cbg = ((w + 15) / 16 + 3) / 4, row_blocks = (h + 15) / 16;
  return n_cbg * 4 * ((row_blocks + kMinSegBlocks - 1) / kMinSegBlocks);
}

hipError_t vif_march_prepare() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipSuccess;
  std::lock_guard<std::mutex> lock(g_tab_mu);
  if (g_tab[dev]) return hipSuccess;
  std::vector<uint16_t> h((size_t)kMarchFrags * 64 * 8);
  if (!build_table(h.data()) && PQA_MARCH_TAP_PIECES == 3) return hipSuccess;   // a tap that does not split exactly: the older kernels stay in charge
  void* d = nullptr;
  if ((e = hipMalloc(&d, h.size() * 2)) != hipSuccess) return e;
  if ((e = hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(d);
    return e;
  }
  {  // the probe (see f16_tiny_probe_kernel): without kept denormals scale 0 stays on the VALU kernel, and says so once
    int* ok = nullptr;
    int seen = 0;
    if ((e = hipMalloc((void**)&ok, sizeof(int))) == hipSuccess) {
      e = hipMemset(ok, 0, sizeof(int));
      if (e == hipSuccess) {
        hipLaunchKernelGGL(f16_tiny_probe_kernel, dim3(1), dim3(64), 0, 0, ok);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpy(&seen, ok, sizeof(int), hipMemcpyDeviceToHost);
      (void)hipFree(ok);
    }
    if (e != hipSuccess) {
      (void)hipFree(d);
      return e;
    }
    if (seen != 64) {
      fprintf(stderr, "pqa_vmaf: device %d does not keep f16 denormals (%d of 64 probe lanes exact): VIF scale 0 runs the "
                      "VALU kernel instead of the matrix-core kernel\n", dev, seen);
      (void)hipFree(d);
      return hipSuccess;
    }
  }
  g_tab[dev] = (const uint4*)d;   // lives as long as the process (20 KB per device)
  return hipSuccess;
}

bool launch_vif_s0_march(hipStream_t stream, Elem elem, int bits, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h, float gain_limit,
                         int border101, double* partials, MutPlaneRun next_ref, MutPlaneRun next_dis, int* n_partials,
                         hipError_t* err) {
  MarchArgs a{};
  a.tab = device_tab();
  if (!a.tab) return false;
  if (!((elem == ELEM_U8 && bits == 8) || (elem == ELEM_U16 && (bits == 10 || bits == 12)))) return false;
  const int es = elem == ELEM_U16 ? 2 : 1;
  if (ref.row_pitch * es >= (1ll << 31) || dis.row_pitch * es >= (1ll << 31)) return false;
  a.ref = ref.base; a.dis = dis.base;
  a.pitch_r = (unsigned)(ref.row_pitch * es); a.pitch_d = (unsigned)(dis.row_pitch * es);
  a.frame_pitch_r = ref.frame_pitch * es; a.frame_pitch_d = dis.frame_pitch * es;
  a.w = w; a.h = h;
  a.fold_w = 2 * w - (border101 ? 2 : 1); a.fold_h = 2 * h - (border101 ? 2 : 1);
  a.aligned = (((uintptr_t)ref.base | (uintptr_t)dis.base | (uintptr_t)a.pitch_r | (uintptr_t)a.pitch_d |
                (uintptr_t)a.frame_pitch_r | (uintptr_t)a.frame_pitch_d) & (uintptr_t)(8 * es - 1)) == 0;
  a.gain_limit = gain_limit;
  a.partials = partials;
  a.dst_ref = (float*)next_ref.base; a.dst_dis = (float*)next_dis.base;
  a.dst_pitch_r = (unsigned)next_ref.row_pitch; a.dst_pitch_d = (unsigned)next_dis.row_pitch;
  a.dst_frame_pitch_r = next_ref.frame_pitch; a.dst_frame_pitch_d = next_dis.frame_pitch;
  // 16-byte stores of the half-resolution rows: pitches and bases the library allocates itself satisfy this
  if (((uintptr_t)a.dst_ref | (uintptr_t)a.dst_dis) & 15 || (a.dst_pitch_r | a.dst_pitch_d) & 3 ||
      (a.dst_frame_pitch_r | a.dst_frame_pitch_d) & 3)
    return false;
  {
    float c17[17], c9[9];
    gaussian(17, c17);
    gaussian(9, c9);
    // the taps as pass 1 applies them: the sum of the pieces it multiplies (all of the f32 tap with three, 22 bits with two)
    const auto applied = [](float c, double scale) {
      uint16_t bits[3];
      const double left = pieces((double)c * scale, PQA_MARCH_TAP_PIECES, bits, 1);
      return ((double)c * scale - left) / scale;
    };
    double s17 = 0.0, s9 = 0.0;
    for (int k = 0; k < 17; ++k) s17 += applied(c17[k], 2048.0);
    for (int k = 0; k < 9; ++k) s9 += applied(c9[k], 262144.0);
    // (12-bit clips are filtered as CENTRED samples: nothing to take off)
    a.mean_off = bits == 12 ? 0.0f : (float)(-128.0 * s17 / 16.0);
    a.dec_off = bits == 12 ? 0.0f : (float)(-128.0 * s9);
  }
  {
    int shape[6];
    vif_march_shape(w, h, shape);
    a.n_cb = shape[0]; a.n_cbg = (a.n_cb + 3) / 4; a.row_blocks = shape[1]; a.seg_blocks = shape[2]; a.n_seg = shape[3];
  }
  a.n_part = a.n_cbg * 4 * a.n_seg;
  if (n_partials) *n_partials = a.n_part;
  const dim3 grid(a.n_cbg * a.n_seg, n_frames), block(kBlock);
  if (bits == 12) hipLaunchKernelGGL((vif_s0_march_kernel<uint16_t, true>), grid, block, 0, stream, a);
  else if (elem == ELEM_U16) hipLaunchKernelGGL((vif_s0_march_kernel<uint16_t, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((vif_s0_march_kernel<uint8_t, false>), grid, block, 0, stream, a);
  *err = hipGetLastError();
  return true;
}

}  // namespace pqa
