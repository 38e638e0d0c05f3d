// Fixed-point VIF for gfx950: the arithmetic of libvmaf's integer_vif.c (the extractor `model=version=vmaf_v0.6.1`
// runs, app/vmaf_analyzer.py:377, models/vmaf_v0.6.1.json:31-38), restated in oracle/vmaf_int_oracle.c and matched
// BIT FOR BIT by this kernel (integer work: tests/test_gpu_parity.py::test_vif_fixed_point_is_bit_exact).
//
//   taps: Q16 tables {489,...,7784,...,489} / {1244,...} / {3571,...} / {10904,43728,10904}, sum 65536
//   vertical:   mu = (sum c*x + 2^(s-1)) >> s  (u16; s = bpc at scale 0, 16 deeper)
//               xx = sum c*x*x  (u32 at 8-bit scale 0; u64 >> 2(bpc-8) or >> 16 elsewhere), same for yy, xy
//   horizontal: a_mu = sum c*mu (u32, Q24);  a_xx = sum c*xx (u64) -> (a_xx + 32768) >> 16
//   statistic:  mu^2 via (a*a + 2^31) >> 32, sigma in Q16, sigma_nsq = 2 << 16, log2 from a 2048-step table on the
//               top 16 bits, g = sigma12 / (sigma1_sq + eps) in double; six integer accumulators per frame
//   borders:    reflect-101 on all four edges (pad_top_and_bottom / PADDING_SQ_DATA)
//
// Mapping to the hardware.  Everything 16 x 16 bit goes through v_dot2_u32_u16 (two taps per instruction):
//   * vertical pass (lane <-> column): vertically adjacent samples are packed (row 2m, row 2m+1) into one
//     dword; an output row whose window starts on an even loaded row uses tap pairs (c0,c1),(c2,c3),..,(c16,0),
//     one that starts on an odd row uses (0,c0),(c1,c2),..,(c15,c16) on the SAME packed registers -- no
//     re-packing per alignment.  At 8 bit r*r, d*d, r*d fit 16 bits, so v_pk_mul_lo_u16 squares two rows at once
//     and the second-order sums are dot2 chains too; wider inputs take v_mad_u64_u32.
//   * horizontal pass (lane <-> 4 adjacent outputs of one row): LDS holds eight u16 planes (mu1, mu2, and the
//     low / high halves of xx, yy, xy); horizontally adjacent columns are already dword pairs there, the same
//     even / odd tap-pair trick applies, and a 48-bit sum is (sum c*hi << 16) + sum c*lo exactly.
//   * the log2 table (32768 x u16, built on the host with the expression integer_vif.c uses) is gathered from
//     global memory (L2-resident); 64 KB of LDS for it would halve the occupancy.
// All accumulators are integers, so the two-stage reduction is exact in any order.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

constexpr int kFxTH = kVifTileH;
constexpr int kFxPitch = 264;  // u16 per LDS plane row: 256 columns + 8 so the last group's spare dword stays inside

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct FxTaps {
  uint32_t ev[9], od[9];    // this scale: (c[2i], c[2i+1]) and (c[2i-1], c[2i]) as lo | hi << 16
  uint32_t dev[5], dod[5];  // next scale's taps, same two alignments (fused decimation)
  uint32_t c[17], dc[9];    // plain taps for the 64-bit path / the small horizontal decimation
};

struct FxArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, tiles_x, n_tiles;
  int shift_vp, shift_sq;
  uint32_t add_vp, add_sq;
  double gain_limit;
  const uint16_t* log2_lut;  // entry i <-> round(log2f(32768 + i) * 2048)
  long long* partials;       // [frames][tiles][8]
  uint16_t* dst_ref;
  uint16_t* dst_dis;
  int64_t dst_row_pitch_r, dst_frame_pitch_r, dst_row_pitch_d, dst_frame_pitch_d;
  FxTaps taps;
};

__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), acc, false);
}
__device__ __forceinline__ uint32_t pk_mul16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ int mirror101(int i, int n) {
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  return min(max(i, 0), n - 1);
}

// "best 16 bits" of v (>= 2^16 here): top set bit moved to bit 15 by truncation; x = -(shift)
__device__ __forceinline__ uint32_t best16(uint64_t v, int& x) {
  const int k = 48 - __clzll((long long)v);
  x = -k;
  return (uint32_t)(v >> k);
}

struct FxAcc {
  int num_log = 0, den_log = 0, x = 0, x2 = 0, n_log = 0, den_non_log = 0;
  long long num_non_log = 0;
};

// vif_statistic_8 / vif_statistic_16, the per-pixel part.  Double arithmetic must round like the C oracle
// (compiled with -ffp-contract=off): no fused multiply-add here.
__device__ __forceinline__ void fx_statistic(uint32_t a_mu1, uint32_t a_mu2, uint32_t xx, uint32_t yy, uint32_t xy,
                                             double gain_limit, const uint16_t* __restrict__ lut, FxAcc& A) {
#pragma clang fp contract(off)
  const uint32_t mu1_sq = (uint32_t)(((uint64_t)a_mu1 * a_mu1 + 2147483648ull) >> 32);
  const uint32_t mu2_sq = (uint32_t)(((uint64_t)a_mu2 * a_mu2 + 2147483648ull) >> 32);
  const uint32_t mu1_mu2 = (uint32_t)(((uint64_t)a_mu1 * a_mu2 + 2147483648ull) >> 32);
  const int32_t sigma_nsq = 65536 << 1;
  const int32_t sigma1_sq = (int32_t)(xx - mu1_sq);
  int32_t sigma2_sq = (int32_t)(yy - mu2_sq);
  const int32_t sigma12 = (int32_t)(xy - mu1_mu2);
  sigma2_sq = max(sigma2_sq, 0);
  if (sigma1_sq >= sigma_nsq) {
    int x;
    const uint32_t log_den1 = best16((uint64_t)(uint32_t)(sigma_nsq + sigma1_sq), x);
    A.x += x;
    A.n_log += 1;
    A.den_log += lut[log_den1 - 32768u];
    if (sigma12 > 0 && sigma2_sq > 0) {
      const double eps = 65536 * 1.0e-10;
      double g = (double)sigma12 / ((double)sigma1_sq + eps);
      int32_t sv_sq = (int32_t)((double)sigma2_sq - g * (double)sigma12);
      sv_sq = max(sv_sq, 0);
      g = g < gain_limit ? g : gain_limit;
      const uint32_t numer1 = (uint32_t)sv_sq + (uint32_t)sigma_nsq;
      const long long numer1_tmp = (long long)(g * g * (double)sigma1_sq) + (long long)numer1;
      int x1, x2;
      const uint32_t numlog = best16((uint64_t)numer1_tmp, x1);
      const uint32_t denlog = best16((uint64_t)numer1, x2);
      A.x2 += x2 - x1;
      A.num_log += (int)lut[numlog - 32768u] - (int)lut[denlog - 32768u];
    }
  } else {
    A.num_non_log += sigma2_sq;
    A.den_non_log += 1;
  }
}

template <typename T, int N, int TW, int ND>
__global__ __launch_bounds__(kBlock, 2) void vif_fixed_kernel(const FxArgs a) {
  constexpr int R = N / 2, RD = ND / 2, COLS = TW + N - 1, NIN = kFxTH + N - 1, NPAIR = NIN / 2;
  constexpr int NP = (N + 1) / 2, NPD = (ND + 1) / 2, NG = TW / 4;
  constexpr bool IN8 = sizeof(T) == 1;
  static_assert(COLS <= kBlock && NIN % 2 == 0 && TW % 4 == 0 && COLS + 8 <= kFxPitch, "tile geometry");
  __shared__ __attribute__((aligned(16))) uint16_t P[8][kFxTH][kFxPitch];  // mu1 mu2 xx.lo xx.hi yy.lo yy.hi xy.lo xy.hi
  __shared__ uint32_t SD[ND ? kFxTH / 2 : 1][kBlock];                      // next-scale input, ref | dis << 16
  __shared__ long long red[4][8];

  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int x0 = tx * TW, y0 = ty * kFxTH;
  const int tid = threadIdx.x;

  // ---- 1. vertical pass -----------------------------------------------------------------------
  if (tid < COLS) {
    const int col = tid;
    const unsigned gx = (unsigned)mirror101(x0 - R + col, a.w);
    const unsigned pitch_r = (unsigned)a.row_pitch_r, pitch_d = (unsigned)a.row_pitch_d;
    const auto rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r * (unsigned)sizeof(T));
    const auto rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d * (unsigned)sizeof(T));
    uint32_t r[NIN], d[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      const unsigned gy = (unsigned)mirror101(y0 - R + j, a.h);
      r[j] = buf_load<T>(rsrc_r, gx, gy * pitch_r);
      d[j] = buf_load<T>(rsrc_d, gx, gy * pitch_d);
    }
    uint32_t pr[NPAIR], pd[NPAIR];
#pragma unroll
    for (int m = 0; m < NPAIR; ++m) {
      pr[m] = r[2 * m] | (r[2 * m + 1] << 16);
      pd[m] = d[2 * m] | (d[2 * m + 1] << 16);
    }
#pragma unroll
    for (int o = 0; o < kFxTH; ++o) {
      const int base = o >> 1;                                    // o even: rows o.. ; o odd: rows o-1.. (tap -1 = 0)
      const uint32_t* tp = (o & 1) ? a.taps.od : a.taps.ev;
      uint32_t m1 = a.add_vp, m2 = a.add_vp;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        m1 = dot2(pr[base + i], tp[i], m1);
        m2 = dot2(pd[base + i], tp[i], m2);
      }
      P[0][o][col] = (uint16_t)(m1 >> a.shift_vp);
      P[1][o][col] = (uint16_t)(m2 >> a.shift_vp);
      uint32_t xx, yy, xy;
      if (IN8) {  // squares fit 16 bits and the 17-tap sums fit 32: packed squares, dot2 sums, no shift
        xx = yy = xy = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          xx = dot2(pk_mul16(pr[base + i], pr[base + i]), tp[i], xx);
          yy = dot2(pk_mul16(pd[base + i], pd[base + i]), tp[i], yy);
          xy = dot2(pk_mul16(pr[base + i], pd[base + i]), tp[i], xy);
        }
      } else {
        uint64_t sxx = a.add_sq, syy = a.add_sq, sxy = a.add_sq;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const uint32_t c = a.taps.c[k], rv = r[o + k], dv = d[o + k];
          sxx += (uint64_t)c * (rv * rv);   // samples < 2^16: the squares fit 32 bits
          syy += (uint64_t)c * (dv * dv);
          sxy += (uint64_t)c * (rv * dv);
        }
        xx = (uint32_t)(sxx >> a.shift_sq);
        yy = (uint32_t)(syy >> a.shift_sq);
        xy = (uint32_t)(sxy >> a.shift_sq);
      }
      P[2][o][col] = (uint16_t)xx; P[3][o][col] = (uint16_t)(xx >> 16);
      P[4][o][col] = (uint16_t)yy; P[5][o][col] = (uint16_t)(yy >> 16);
      P[6][o][col] = (uint16_t)xy; P[7][o][col] = (uint16_t)(xy >> 16);
    }
    if (ND) {  // filter1d_8 / filter1d_16 with the next scale's taps, even rows only
#pragma unroll
      for (int q = 0; q < kFxTH / 2; ++q) {
        constexpr int off = R - RD;
        const int start = 2 * q + off;
        const int base = start >> 1;
        const uint32_t* tp = (off & 1) ? a.taps.dod : a.taps.dev;
        uint32_t ar = a.add_vp, ad = a.add_vp;
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
          ar = dot2(pr[base + i], tp[i], ar);
          ad = dot2(pd[base + i], tp[i], ad);
        }
        SD[q][col] = ((ar >> a.shift_vp) & 0xffffu) | ((ad >> a.shift_vp) << 16);
      }
    }
  }
  __syncthreads();

  // ---- 1b. decimation: horizontal pass at even columns -> next scale's u16 planes -----------------
  if (ND) {
    const int ox0 = x0 >> 1, oy0 = y0 >> 1, ow = a.w >> 1, oh = a.h >> 1;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int item = tid + round * kBlock;
      const int oc = item & 127, q = item >> 7;
      const int gx = ox0 + oc, gy = oy0 + q;
      if (oc < TW / 2 && gx < ow && gy < oh) {
        uint32_t ar = 32768u, ad = 32768u;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
          const uint32_t v = SD[q][2 * oc + (R - RD) + k];
          ar += a.taps.dc[k] * (v & 0xffffu);
          ad += a.taps.dc[k] * (v >> 16);
        }
        a.dst_ref[(int64_t)fr * a.dst_frame_pitch_r + (int64_t)gy * a.dst_row_pitch_r + gx] = (uint16_t)(ar >> 16);
        a.dst_dis[(int64_t)fr * a.dst_frame_pitch_d + (int64_t)gy * a.dst_row_pitch_d + gx] = (uint16_t)(ad >> 16);
      }
    }
  }

  // ---- 2. horizontal pass + 3. statistic ----------------------------------------------------------
  FxAcc A;
  constexpr int NDW = NP + 1, NRD = (NDW + 1) / 2;  // dwords a group of 4 outputs needs / 8-byte reads
#pragma unroll 1
  for (int item = tid; item < kFxTH * NG; item += kBlock) {
    const int row = item / NG, g = item - row * NG;
    uint32_t acc[8][4];
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) {
      uint32_t dw[2 * NRD];
      const uint2* p = reinterpret_cast<const uint2*>(&P[pl][row][4 * g]);
#pragma unroll
      for (int q = 0; q < NRD; ++q) {
        const uint2 v = p[q];
        dw[2 * q] = v.x;
        dw[2 * q + 1] = v.y;
      }
      uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        s0 = dot2(dw[i], a.taps.ev[i], s0);      // output 4g:   columns 4g.. (even)
        s1 = dot2(dw[i], a.taps.od[i], s1);      // output 4g+1: columns 4g+1.. (odd; tap -1 = 0 on column 4g)
        s2 = dot2(dw[i + 1], a.taps.ev[i], s2);  // output 4g+2
        s3 = dot2(dw[i + 1], a.taps.od[i], s3);  // output 4g+3
      }
      acc[pl][0] = s0; acc[pl][1] = s1; acc[pl][2] = s2; acc[pl][3] = s3;
    }
    const int gy = y0 + row;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int gx = x0 + 4 * g + t;
      if (gx < a.w && gy < a.h) {
        // a 48-bit sum c*v with v = hi << 16 | lo is (sum c*hi << 16) + sum c*lo; (.. + 32768) >> 16 of it:
        const uint32_t xx = acc[3][t] + ((acc[2][t] + 32768u) >> 16);
        const uint32_t yy = acc[5][t] + ((acc[4][t] + 32768u) >> 16);
        const uint32_t xy = acc[7][t] + ((acc[6][t] + 32768u) >> 16);
        fx_statistic(acc[0][t], acc[1][t], xx, yy, xy, a.gain_limit, a.log2_lut, A);
      }
    }
  }

  // ---- integer block reduction (exact in any order) -------------------------------------------------
  long long v[7] = {A.num_log, A.den_log, A.x, A.x2, A.n_log, A.den_non_log, A.num_non_log};
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    if (i < 6) {
      int s = (int)v[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      v[i] = s;
    } else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) red[tid >> 6][i] = v[i];
  }
  __syncthreads();
  if (tid < 7) {
    long long* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * 8;
    out[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

constexpr uint16_t kFxFilter[4][17] = {
    {489, 935, 1640, 2640, 3896, 5274, 6547, 7455, 7784, 7455, 6547, 5274, 3896, 2640, 1640, 935, 489},
    {1244, 3663, 7925, 12590, 14692, 12590, 7925, 3663, 1244},
    {3571, 16004, 26386, 16004, 3571},
    {10904, 43728, 10904},
};
constexpr int kFxN[4] = {17, 9, 5, 3};

static void fill_pairs(const uint16_t* c, int n, uint32_t* ev, uint32_t* od, int cap) {
  const auto tap = [&](int k) -> uint32_t { return (k >= 0 && k < n) ? c[k] : 0u; };
  for (int i = 0; i < cap; ++i) {
    ev[i] = tap(2 * i) | (tap(2 * i + 1) << 16);
    od[i] = tap(2 * i - 1) | (tap(2 * i) << 16);
  }
}

template <int N, int TW, int ND>
hipError_t launch_fx(hipStream_t stream, Elem elem, const FxArgs& a, int n_frames) {
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  if constexpr (N == 17) {  // only scale 0 ever sees 8-bit samples
    if (elem == ELEM_U8) {
      hipLaunchKernelGGL((vif_fixed_kernel<uint8_t, N, TW, ND>), grid, block, 0, stream, a);
      return hipGetLastError();
    }
  }
  if (elem != ELEM_U16) return hipErrorInvalidValue;
  hipLaunchKernelGGL((vif_fixed_kernel<uint16_t, N, TW, ND>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace

void vif_fixed_log2_table(uint16_t* out32768) {
  // integer_vif.c log_generate(): log2_table[i] = (uint16_t)round(log2f((float)i) * 2048), i in [32767, 65535]
  for (int i = 32768; i < 65536; ++i) out32768[i - 32768] = (uint16_t)round(log2f((float)i) * 2048);
}

hipError_t launch_vif_fixed(hipStream_t stream, int scale, int bit_depth, Elem elem, PlaneRun ref, PlaneRun dis,
                            int n_frames, int w, int h, double gain_limit, const uint16_t* log2_lut,
                            long long* partials, MutPlaneRun next_ref, MutPlaneRun next_dis) {
  if (n_frames <= 0) return hipSuccess;
  FxArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.tiles_x = vif_tiles_x(scale, w);
  a.n_tiles = a.tiles_x * vif_tiles_y(h);
  if (scale == 0) {
    a.shift_vp = bit_depth; a.add_vp = 1u << (bit_depth - 1);
    a.shift_sq = (bit_depth - 8) * 2; a.add_sq = bit_depth == 8 ? 0u : 1u << (a.shift_sq - 1);
  } else {
    a.shift_vp = 16; a.add_vp = 32768u; a.shift_sq = 16; a.add_sq = 32768u;
  }
  a.gain_limit = gain_limit;
  a.log2_lut = log2_lut;
  a.partials = partials;
  a.dst_ref = (uint16_t*)next_ref.base; a.dst_dis = (uint16_t*)next_dis.base;
  a.dst_row_pitch_r = next_ref.row_pitch; a.dst_frame_pitch_r = next_ref.frame_pitch;
  a.dst_row_pitch_d = next_dis.row_pitch; a.dst_frame_pitch_d = next_dis.frame_pitch;
  fill_pairs(kFxFilter[scale], kFxN[scale], a.taps.ev, a.taps.od, 9);
  for (int k = 0; k < 17; ++k) a.taps.c[k] = k < kFxN[scale] ? kFxFilter[scale][k] : 0;
  if (scale < 3) {
    fill_pairs(kFxFilter[scale + 1], kFxN[scale + 1], a.taps.dev, a.taps.dod, 5);
    for (int k = 0; k < 9; ++k) a.taps.dc[k] = k < kFxN[scale + 1] ? kFxFilter[scale + 1][k] : 0;
    if (!a.dst_ref || !a.dst_dis) return hipErrorInvalidValue;
  }
  // inputs: scale 0 reads the caller's samples (u8 at 8 bit, u16 above), deeper scales the u16 planes written here
  if (elem == ELEM_F32 || (scale > 0 && elem != ELEM_U16) || (scale == 0 && (elem == ELEM_U8) != (bit_depth == 8)))
    return hipErrorInvalidValue;
  switch (scale) {
    case 0: return launch_fx<17, 240, 9>(stream, elem, a, n_frames);
    case 1: return launch_fx<9, 248, 5>(stream, elem, a, n_frames);
    case 2: return launch_fx<5, 252, 3>(stream, elem, a, n_frames);
    case 3: return launch_fx<3, 252, 0>(stream, elem, a, n_frames);
  }
  return hipErrorInvalidValue;
}

}  // namespace pqa
