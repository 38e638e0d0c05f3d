// Fixed-point ADM for gfx950: the arithmetic of libvmaf's integer_adm.c (the extractor behind `integer_adm2` /
// `integer_adm_scale0..3` of the default models, app/vmaf_analyzer.py:377), restated in oracle/vmaf_int_oracle.c.
// Integer work: the six accumulators per (frame, scale) this file produces are bit-identical to the restatement's,
// and the scalar epilogue (six cube roots per scale) runs on the host inside pqa_collect with the same libm.
//
//   DWT      Q15 db2 taps; scale 0: pixels -> (sum +- ..) >> bpc, then >> 16 -> int16 bands in Q6 (LL rows have the
//            2^(bpc-1) offset removed); scales 1-3: int32 LL of the previous scale, 64-bit sums, shifts {0,16,16}
//            vertical and {15,16,15} horizontal -> int32 bands in Q21 / Q19 / Q18
//   decouple k = clip(t/o, 0, 1) in Q15 through a 2^30 reciprocal table (15 best bits of o at scales 1-3),
//            r = (k*o + 2^14) >> 15, the 1-degree angle test in float on the 64-bit dot products, enhancement
//            gain limit in double
//   CSF      Q21/Q21/Q23 weights at scale 0 (>> 15/15/17 to Q12), Q32 deeper (>> 28); |.|/30 and |.|/15 by
//            multiplication with 2^17/30 (2^32/30)
//   masking  thr = 8 neighbours of sum_theta |csf(a)|/30 + sum_theta |csf(a)|/15 of the centre;
//            x = |w*r| - thr, x^2 >> 29|30 (30), x^3 >> ceil(log2 w)-4|3 (ceil(log2 w)), row sums >> ceil(log2 h)
//   den      |o|^3 row sums >> ceil(log2 area)-20 at scale 0; (|o|^2 >> 31|30|31) * |o| >> ceil(log2 w) deeper
// Tiling as adm.hip: a 60 x 14 coefficient tile with a one-coefficient halo, vertical DWT with one input column
// per lane into LDS, then one coefficient per lane.  The per-ROW shifts of the accumulations mean a row's sum must
// be complete before it is shifted: tiles emit per-row partial sums (exact integers), finalize.hip adds the tiles
// of a row, applies the row shift and adds the rows.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

constexpr int TW = kAdmTileW, TH = kAdmTileH, GW = TW + 2, GH = TH + 2;
constexpr int VC = 2 * GW + 2, VP = 128;  // vertical-pass columns (126) / LDS pitch
constexpr int SROWS = GH / 2, NIN = 2 * SROWS + 2;
static_assert(VC <= 128 && GH == 16 && GW <= 64 && GH == kAdmFxRows, "one column per lane, 4 rows per wave");

struct AfxArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, ow, oh, tiles_x, n_tiles;
  int left, top, right, bottom;
  int shift_vp, shift_hp;
  int add_vp, add_hp, lo_norm;
  uint32_t i_rf[3];
  int cm_shift_sq[3], cm_shift_sub[3], cm_shift_cub[3];
  int den_shift_sq, den_shift_cub;
  float cos_1deg_sq;
  double gain_limit;
  const int32_t* div_lut;  // [65537]: 2^30 / (i - 32768)
  int32_t* ll_ref;
  int32_t* ll_dis;
  int64_t ll_row_pitch_r, ll_frame_pitch_r, ll_row_pitch_d, ll_frame_pitch_d;
  long long* partials;  // [frames][tiles][GH][6]
};

constexpr int kLo[4] = {15826, 27411, 7345, -4240};
constexpr int kHi[4] = {-4240, -7345, 27411, -15826};

template <typename T> __device__ __forceinline__ int load_sample(rsrc_t r, unsigned x, unsigned row_off);
template <> __device__ __forceinline__ int load_sample<uint8_t>(rsrc_t r, unsigned x, unsigned row_off) {
  return (int)buf_load<uint8_t>(r, x, row_off);
}
template <> __device__ __forceinline__ int load_sample<uint16_t>(rsrc_t r, unsigned x, unsigned row_off) {
  return (int)buf_load<uint16_t>(r, x, row_off);
}
template <> __device__ __forceinline__ int load_sample<int32_t>(rsrc_t r, unsigned x, unsigned row_off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, x * 4u, row_off * 4u, 0);
}

__device__ __forceinline__ long long iabs64(long long v) { return v < 0 ? -v : v; }

// k = clip(t / o, 0, 1) in Q15 through the reciprocal table (adm_decouple / adm_decouple_s123)
template <bool WIDE>
__device__ __forceinline__ int decouple_k(int o, int t, const int32_t* __restrict__ lut) {
  long long tmp;
  if (o == 0) return 32768;
  if (!WIDE) {
    tmp = ((long long)lut[o + 32768] * t + 16384) >> 15;
  } else {
    const int sign = o < 0 ? -1 : 1;
    const unsigned ao = (unsigned)(o < 0 ? -(long long)o : (long long)o);
    int shift = 0;
    int msb = (int)ao;
    if (ao >= 32768u) {  // get_best15_from32: 15 best bits, rounded
      shift = 17 - __clz((int)ao);
      msb = (int)((ao + (1u << (shift - 1))) >> shift);
    }
    tmp = ((long long)lut[msb + 32768] * t * sign + (1ll << (14 + shift))) >> (15 + shift);
  }
  return tmp < 0 ? 0 : (tmp > 32768 ? 32768 : (int)tmp);
}

__device__ __forceinline__ int gain_limit_rst(int rst, int t, double gl) {
#pragma clang fp contract(off)
  if (rst > 0) { const double x = rst * gl; return (int)(x < t ? x : t); }
  if (rst < 0) { const double x = rst * gl; return (int)(x > t ? x : t); }
  return rst;
}

// Wave sums of 24 int64 values per lane (6 quantities x this wave's 4 band rows) as a reduce-scatter: at lane
// distance 32, 16, 8 every lane hands the half of its values it does not keep to its partner, so the number of live
// values halves each step (12 + 6 + 3 exchanges instead of 24 per step); the last three distances reduce 3 values.
// Afterwards lane L holds, in v[0..2], the wave totals of original indices 12*b5 + 6*b4 + 3*b3 + {0,1,2}
// (b5, b4, b3 = bits 5, 4, 3 of L).  Exact integer adds: the order is free.
__device__ __forceinline__ void wave_reduce_scatter24(long long (&v)[24], int lane) {
#pragma unroll
  for (int step = 0; step < 3; ++step) {
    const int m = 32 >> step, n = 12 >> step;
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const long long send = up ? v[i] : v[i + n];
      const long long keep = up ? v[i + n] : v[i];
      v[i] = keep + __shfl_xor(send, m, 64);
    }
  }
#pragma unroll
  for (int m = 4; m > 0; m >>= 1)
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += __shfl_xor(v[i], m, 64);
}

template <typename T, bool WIDE>
__global__ __launch_bounds__(kBlock, 2) void adm_fixed_kernel(const AfxArgs a) {
  __shared__ int V[4][GH][VP];  // vertical DWT: 0 lo(ref) 1 hi(ref) 2 lo(dis) 3 hi(dis)
  __shared__ int F[GH][GW + 2];  // masking signal: sum over orientations of |csf(a)| / 30

  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int cx0 = tx * TW, cy0 = ty * TH;
  const int tid = threadIdx.x;
  // A tile with no coefficient inside the accumulation window [left, right) x [top, bottom) (integer_adm.c crops 10 % on
  // every side: a third of a frame's tiles) contributes zeros to every sum; all it owes is its piece of the approximation
  // band the next scale reads (as in adm.hip).  Workgroup-uniform.
  const bool outside = !(cx0 < a.right && cx0 + TW > a.left && cy0 < a.bottom && cy0 + TH > a.top);
  long long* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * (GH * 6);
  if (outside) {
    if (tid < GH * 6) out[tid] = 0;
    if (!a.ll_ref) return;   // last scale: nothing else to produce
  }
  const unsigned pitch_r = (unsigned)a.row_pitch_r, pitch_d = (unsigned)a.row_pitch_d;
  const rsrc_t rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r * (unsigned)sizeof(T));
  const rsrc_t rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d * (unsigned)sizeof(T));

  // ---- phase 1: vertical DWT ---------------------------------------------------------------------
  {
    const int col = tid & 127;
    const int strip = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (col < VC) {
      const unsigned gx = (unsigned)mirror1(2 * cx0 - 3 + col, a.w);
      int r[NIN], d[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const unsigned gy = (unsigned)mirror1(2 * cy0 - 3 + 2 * SROWS * strip + j, a.h);
        r[j] = load_sample<T>(rsrc_r, gx, gy * pitch_r);
        d[j] = load_sample<T>(rsrc_d, gx, gy * pitch_d);
      }
#pragma unroll
      for (int o = 0; o < SROWS; ++o) {
        const int row = strip * SROWS + o;
        if (!WIDE) {
          int lr = -a.lo_norm, hr = 0, ld = -a.lo_norm, hd = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lr += kLo[k] * r[2 * o + k]; hr += kHi[k] * r[2 * o + k];
            ld += kLo[k] * d[2 * o + k]; hd += kHi[k] * d[2 * o + k];
          }
          V[0][row][col] = (lr + a.add_vp) >> a.shift_vp;
          V[2][row][col] = (ld + a.add_vp) >> a.shift_vp;
          if (!outside) {
            V[1][row][col] = (hr + a.add_vp) >> a.shift_vp;
            V[3][row][col] = (hd + a.add_vp) >> a.shift_vp;
          }
        } else {
          long long lr = 0, hr = 0, ld = 0, hd = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lr += (long long)kLo[k] * r[2 * o + k]; hr += (long long)kHi[k] * r[2 * o + k];
            ld += (long long)kLo[k] * d[2 * o + k]; hd += (long long)kHi[k] * d[2 * o + k];
          }
          V[0][row][col] = (int)((lr + a.add_vp) >> a.shift_vp);
          V[2][row][col] = (int)((ld + a.add_vp) >> a.shift_vp);
          if (!outside) {
            V[1][row][col] = (int)((hr + a.add_vp) >> a.shift_vp);
            V[3][row][col] = (int)((hd + a.add_vp) >> a.shift_vp);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: horizontal DWT, decouple, CSF; denominator cubes -------------------------------------
  const int lcx = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool have = lcx < GW;
  const int lcxs = have ? lcx : 0;
  const int cx = cx0 - 1 + lcxs;
  const bool col_valid = have && cx >= 0 && cx < a.ow;
  const bool col_inner = col_valid && lcxs >= 1 && lcxs <= TW;
  const bool col_win = col_inner && cx >= a.left && cx < a.right;
  int rs[4][3];         // restored coefficients h, v, d of this lane's four rows
  int centre[4];        // sum over orientations of |csf(a)| / 15
  bool win[4];
  long long den[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = wave + 4 * k;
    const int cy = cy0 - 1 + lr;
    if (outside) {   // approximation band only: the low vertical bands through the low-pass taps
      const bool row_valid_o = cy >= 0 && cy < a.oh;
      if (col_inner && lr >= 1 && lr <= TH && row_valid_o) {
        int ll[2];
#pragma unroll
        for (int im = 0; im < 2; ++im) {
          const int2* p = reinterpret_cast<const int2*>(&V[2 * im][lr][2 * lcxs]);
          const int2 v01 = p[0], v23 = p[1];
          const int sv[4] = {v01.x, v01.y, v23.x, v23.y};
          if (!WIDE) {
            int ba = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) ba += kLo[t] * sv[t];
            ll[im] = (short)((ba + a.add_hp) >> a.shift_hp);
          } else {
            long long ba = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) ba += (long long)kLo[t] * sv[t];
            ll[im] = (int)((ba + a.add_hp) >> a.shift_hp);
          }
        }
        a.ll_ref[(int64_t)fr * a.ll_frame_pitch_r + (int64_t)cy * a.ll_row_pitch_r + cx] = ll[0];
        a.ll_dis[(int64_t)fr * a.ll_frame_pitch_d + (int64_t)cy * a.ll_row_pitch_d + cx] = ll[1];
      }
      continue;
    }
    int s[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int2* p = reinterpret_cast<const int2*>(&V[q][lr][2 * lcxs]);
      const int2 v01 = p[0], v23 = p[1];
      s[q][0] = v01.x; s[q][1] = v01.y; s[q][2] = v23.x; s[q][3] = v23.y;
    }
    int band[2][4];  // [ref|dis][a, v, h, d]
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      if (!WIDE) {
        int ba = 0, bv = 0, bh = 0, bd = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ba += kLo[t] * s[2 * im][t]; bv += kHi[t] * s[2 * im][t];
          bh += kLo[t] * s[2 * im + 1][t]; bd += kHi[t] * s[2 * im + 1][t];
        }
        band[im][0] = (short)((ba + a.add_hp) >> a.shift_hp);
        band[im][1] = (short)((bv + a.add_hp) >> a.shift_hp);
        band[im][2] = (short)((bh + a.add_hp) >> a.shift_hp);
        band[im][3] = (short)((bd + a.add_hp) >> a.shift_hp);
      } else {
        long long ba = 0, bv = 0, bh = 0, bd = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ba += (long long)kLo[t] * s[2 * im][t]; bv += (long long)kHi[t] * s[2 * im][t];
          bh += (long long)kLo[t] * s[2 * im + 1][t]; bd += (long long)kHi[t] * s[2 * im + 1][t];
        }
        band[im][0] = (int)((ba + a.add_hp) >> a.shift_hp);
        band[im][1] = (int)((bv + a.add_hp) >> a.shift_hp);
        band[im][2] = (int)((bh + a.add_hp) >> a.shift_hp);
        band[im][3] = (int)((bd + a.add_hp) >> a.shift_hp);
      }
    }
    const bool row_valid = cy >= 0 && cy < a.oh;
    const bool inner = col_inner && lr >= 1 && lr <= TH && row_valid;
    if (a.ll_ref && inner) {
      a.ll_ref[(int64_t)fr * a.ll_frame_pitch_r + (int64_t)cy * a.ll_row_pitch_r + cx] = band[0][0];
      a.ll_dis[(int64_t)fr * a.ll_frame_pitch_d + (int64_t)cy * a.ll_row_pitch_d + cx] = band[1][0];
    }
    const int oh_ = band[0][2], ov = band[0][1], od = band[0][3];
    const int th = band[1][2], tv = band[1][1], td = band[1][3];
    bool angle;
    {
#pragma clang fp contract(off)
      const long long ot_dp = (long long)oh_ * th + (long long)ov * tv;
      const long long o_mag_sq = (long long)oh_ * oh_ + (long long)ov * ov;
      const long long t_mag_sq = (long long)th * th + (long long)tv * tv;
      const float f_dp = (float)ot_dp / 4096.0f;
      angle = (f_dp >= 0.0f) &&
              (f_dp * f_dp >= a.cos_1deg_sq * ((float)o_mag_sq / 4096.0f) * ((float)t_mag_sq / 4096.0f));
    }
    const int o3[3] = {oh_, ov, od}, t3[3] = {th, tv, td};
    int fsum = 0, csum = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int kq = decouple_k<WIDE>(o3[t], t3[t], a.div_lut);
      // k <= 2^15; at scale 0 |o| <= 2^15 as well, so k * o fits 32 bits there
      int rst = WIDE ? (int)(((long long)kq * o3[t] + 16384) >> 15) : (kq * o3[t] + 16384) >> 15;
      if (angle) rst = gain_limit_rst(rst, t3[t], a.gain_limit);
      rs[k][t] = rst;
      const int add = t3[t] - rst;
      if (!WIDE) {
        const int shifts = t == 2 ? 17 : 15, adds = t == 2 ? 65535 : 16384;
        const int dst = (int)a.i_rf[t] * add;
        const int v = (short)((dst + adds) >> shifts);
        const int av = v < 0 ? -v : v;
        fsum += (short)((4369 * av + 2048) >> 12);
        csum += (short)((8738 * av + 2048) >> 12);
      } else {
        const int v = (int)(((long long)a.i_rf[t] * add + (1ll << 27)) >> 28);
        const long long av = iabs64(v);
        fsum += (int)((143165577ll * av + (1ll << 31)) >> 32);
        csum += (int)((286331153ll * av + (1ll << 31)) >> 32);
      }
    }
    centre[k] = csum;
    if (have) F[lr][lcx] = (col_valid && row_valid) ? fsum : 0;
    win[k] = inner && col_win && cy >= a.top && cy < a.bottom;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned long long val = 0;
      if (win[k]) {
        const unsigned long long v = (unsigned long long)iabs64(o3[t]);
        if (!WIDE) {  // |o| <= 2^15: the square fits 32 bits, the cube is one 32 x 32 -> 64 multiply
          const unsigned v32 = (unsigned)v;
          val = (unsigned long long)(v32 * v32) * v32;
        } else {
          const unsigned long long sq = (v * v + (1ull << (a.den_shift_sq - 1))) >> a.den_shift_sq;
          const unsigned long long add_cub = a.den_shift_cub > 0 ? 1ull << (a.den_shift_cub - 1) : 0ull;
          val = (sq * v + add_cub) >> a.den_shift_cub;
        }
      }
      den[k][t] = (long long)val;
    }
  }
  if (outside) return;   // (its partials were cleared at the top)
  __syncthreads();

  // ---- phase 3: contrast masking; per-row sums ------------------------------------------------------
  const int lx0 = have ? min(max(mirror1(cx - 1, a.ow) - (cx0 - 1), 0), GW - 1) : 0;
  const int lx2 = have ? min(max(mirror1(cx + 1, a.ow) - (cx0 - 1), 0), GW - 1) : 0;
  long long sums[24];  // index k * 6 + {num h,v,d, den h,v,d}
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = wave + 4 * k;
    const int cy = cy0 - 1 + lr;
    const int ly0 = min(max(mirror1(cy - 1, a.oh) - (cy0 - 1), 0), GH - 1);
    const int ly2 = min(max(mirror1(cy + 1, a.oh) - (cy0 - 1), 0), GH - 1);
    long long num[3] = {0, 0, 0};
    if (win[k]) {
      // the eight neighbours at their mirrored band positions (adm_cm's 3x3 box without its centre) + the centre term
      long long thr = (long long)F[ly0][lx0] + F[ly0][lcxs] + F[ly0][lx2] + F[lr][lx0] + F[lr][lx2] + F[ly2][lx0] +
                      F[ly2][lcxs] + F[ly2][lx2] + centre[k];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const long long add_cub = a.cm_shift_cub[t] > 0 ? 1ll << (a.cm_shift_cub[t] - 1) : 0ll;
        if (!WIDE) {
          // |r * w| < 2^31, so after the threshold x is a 32-bit quantity: x^2 is one 32 x 32 -> 64 multiply, and
          // x_sq (< 2^34) * x splits into two of them (same integers as the 64-bit forms of the C code)
          const int xw = rs[k][t] * (int)a.i_rf[t];
          const long long xs = (long long)(unsigned)(xw < 0 ? -xw : xw) - (thr << a.cm_shift_sub[t]);
          const unsigned x = xs < 0 ? 0u : (unsigned)xs;
          const unsigned long long x_sq = ((unsigned long long)x * x + (1ull << (a.cm_shift_sq[t] - 1))) >> a.cm_shift_sq[t];
          const unsigned long long cub = (unsigned long long)(unsigned)x_sq * x +
                                         ((unsigned long long)((unsigned)(x_sq >> 32) * x) << 32);
          num[t] = (long long)((cub + (unsigned long long)add_cub) >> a.cm_shift_cub[t]);
        } else {
          long long x = ((long long)a.i_rf[t] * rs[k][t] + (1ll << 27)) >> 28;
          x = iabs64(x) - (thr << a.cm_shift_sub[t]);
          if (x < 0) x = 0;
          const long long x_sq = (x * x + (1ll << (a.cm_shift_sq[t] - 1))) >> a.cm_shift_sq[t];
          num[t] = (x_sq * x + add_cub) >> a.cm_shift_cub[t];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      sums[k * 6 + t] = num[t];
      sums[k * 6 + 3 + t] = den[k][t];
    }
  }
  // all lanes of a wave hold coefficients of the same four band rows: one exact integer sum per row and quantity
  wave_reduce_scatter24(sums, lcx);
  if ((lcx & 7) == 0) {
    const int base = 12 * ((lcx >> 5) & 1) + 6 * ((lcx >> 4) & 1) + 3 * ((lcx >> 3) & 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int idx = base + j, k = idx / 6, q = idx - 6 * k;
      out[(wave + 4 * k) * 6 + q] = sums[j];
    }
  }
}

}  // namespace

// Watson DWT 7/9 model in float, exactly as adm_tools.h dwt_quant_step() evaluates it (view distance 3 H, 1080 lines)
static float fx_dwt_quant_step(int lambda, int theta) {
  static const float a = 0.495f, k = 0.466f, f0 = 0.401f;
  static const float g[4] = {1.501f, 1.0f, 0.534f, 1.0f};
  static const float amp[6][4] = {
      {0.62171f, 0.67234f, 0.72709f, 0.67234f},     {0.34537f, 0.41317f, 0.49428f, 0.41317f},
      {0.18004f, 0.22727f, 0.28688f, 0.22727f},     {0.091401f, 0.11792f, 0.15214f, 0.11792f},
      {0.045943f, 0.059758f, 0.077727f, 0.059758f}, {0.023013f, 0.030018f, 0.039156f, 0.030018f},
  };
  float r = (float)(3.0 * 1080 * M_PI / 180.0);
  float temp = (float)log10(pow(2.0, lambda + 1) * f0 * g[theta] / r);
  float Q = (float)(2.0 * a * pow(10.0, k * temp * temp) / amp[lambda][theta]);
  return Q;
}

static int ceil_log2_minus(double v, int minus) {
  const int s = (int)ceil(log2(v) - minus);
  return s < 0 ? 0 : s;
}

AdmFxScale adm_fixed_scale_params(int scale, int band_w, int band_h) {
  AdmFxScale p{};
  p.scale = scale; p.band_w = band_w; p.band_h = band_h;
  p.left = (int)(band_w * 0.1 - 0.5);
  p.top = (int)(band_h * 0.1 - 0.5);
  p.right = band_w - p.left;
  p.bottom = band_h - p.top;
  p.rf[0] = p.rf[1] = 1.0f / fx_dwt_quant_step(scale, 1);
  p.rf[2] = 1.0f / fx_dwt_quant_step(scale, 2);
  if (scale == 0) {
    p.i_rf[0] = p.i_rf[1] = (uint16_t)((double)p.rf[0] * pow(2.0, 21));
    p.i_rf[2] = (uint16_t)((double)p.rf[2] * pow(2.0, 23));
    p.cm_shift_cub[0] = p.cm_shift_cub[1] = ceil_log2_minus((double)band_w, 4);
    p.cm_shift_cub[2] = ceil_log2_minus((double)band_w, 3);
    p.cm_shift_sq[0] = p.cm_shift_sq[1] = 29; p.cm_shift_sq[2] = 30;
    p.cm_shift_sub[0] = p.cm_shift_sub[1] = 10; p.cm_shift_sub[2] = 12;
    p.cm_final_q[0] = p.cm_final_q[1] = 52; p.cm_final_q[2] = 57;
    const double area = (double)(p.bottom - p.top) * (p.right - p.left);
    p.den_row_shift = ceil_log2_minus(area, 20);
    p.den_final_q = 18 - p.den_row_shift;
  } else {
    static const int fq[3] = {45, 39, 36}, dsq[3] = {31, 30, 31}, dq[3] = {32, 27, 23};
    for (int t = 0; t < 3; ++t) {
      p.i_rf[t] = (uint32_t)((double)p.rf[t] * pow(2.0, 32));
      p.cm_shift_cub[t] = ceil_log2_minus((double)band_w, 0);
      p.cm_shift_sq[t] = 30; p.cm_shift_sub[t] = 0; p.cm_final_q[t] = fq[scale - 1];
    }
    p.den_shift_sq = dsq[scale - 1];
    p.den_shift_cub = ceil_log2_minus((double)band_w, 0);
    p.den_row_shift = ceil_log2_minus((double)band_h, 0);
    p.den_final_q = dq[scale - 1] - p.den_shift_cub - p.den_row_shift;
  }
  p.num_row_shift = ceil_log2_minus((double)band_h, 0);
  return p;
}

void adm_fixed_epilogue(const AdmFxScale& p, const long long acc[6], double* num_out, double* den_out) {
  // adm_cm / adm_csf_den_scale (and their i4_ forms), the part after the integer accumulations
  const float powf_add = powf((float)((p.bottom - p.top) * (p.right - p.left)) / 32.0f, 1.0f / 3.0f);
  float num = 0, den = 0;
  for (int t = 0; t < 3; ++t) {
    const float f_accum =
        (float)((double)acc[t] / pow(2.0, p.cm_final_q[t] - p.cm_shift_cub[t] - p.num_row_shift));
    num += powf(f_accum, 1.0f / 3.0f) + powf_add;
    const double csf = ((double)(unsigned long long)acc[3 + t] / pow(2.0, p.den_final_q)) * pow((double)p.rf[t], 3.0);
    den += powf((float)csf, 1.0f / 3.0f) + powf_add;
  }
  *num_out = (double)num;
  *den_out = (double)den;
}

void adm_fixed_div_table(int32_t* out65537) {
  const int32_t div_Q_factor = 1073741824;  // 2^30
  out65537[32768] = 0;
  for (int i = 1; i <= 32768; ++i) {
    const int32_t recip = div_Q_factor / i;
    out65537[32768 + i] = recip;
    out65537[32768 - i] = 0 - recip;
  }
}

hipError_t launch_adm_fixed(hipStream_t stream, int scale, int bit_depth, Elem elem, PlaneRun ref, PlaneRun dis,
                            int n_frames, int w, int h, double gain_limit, const int32_t* div_lut,
                            MutPlaneRun ll_ref, MutPlaneRun ll_dis, long long* partials) {
  if (n_frames <= 0) return hipSuccess;
  AfxArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.ow = (w + 1) / 2; a.oh = (h + 1) / 2;
  a.tiles_x = adm_tiles_x(a.ow);
  a.n_tiles = a.tiles_x * adm_tiles_y(a.oh);
  const AdmFxScale p = adm_fixed_scale_params(scale, a.ow, a.oh);
  a.left = p.left; a.top = p.top; a.right = p.right; a.bottom = p.bottom;
  if (scale == 0) {
    a.shift_vp = bit_depth; a.add_vp = 1 << (bit_depth - 1);
    a.shift_hp = 16; a.add_hp = 32768;
    a.lo_norm = 46342 * a.add_vp;  // dwt2_db2_coeffs_lo_sum * 2^(bpc-1): range (0..N) -> (-N/2..N/2)
  } else {
    static const int add_vp[3] = {0, 32768, 32768}, add_hp[3] = {16384, 32768, 16384};
    static const int shift_vp[3] = {0, 16, 16}, shift_hp[3] = {15, 16, 15};
    a.shift_vp = shift_vp[scale - 1]; a.add_vp = add_vp[scale - 1];
    a.shift_hp = shift_hp[scale - 1]; a.add_hp = add_hp[scale - 1];
  }
  for (int t = 0; t < 3; ++t) {
    a.i_rf[t] = p.i_rf[t];
    a.cm_shift_sq[t] = p.cm_shift_sq[t]; a.cm_shift_sub[t] = p.cm_shift_sub[t]; a.cm_shift_cub[t] = p.cm_shift_cub[t];
  }
  a.den_shift_sq = p.den_shift_sq; a.den_shift_cub = p.den_shift_cub;
  a.cos_1deg_sq = (float)(cos(1.0 * M_PI / 180.0) * cos(1.0 * M_PI / 180.0));
  a.gain_limit = gain_limit;
  a.div_lut = div_lut;
  a.ll_ref = (int32_t*)ll_ref.base; a.ll_dis = (int32_t*)ll_dis.base;
  a.ll_row_pitch_r = ll_ref.row_pitch; a.ll_frame_pitch_r = ll_ref.frame_pitch;
  a.ll_row_pitch_d = ll_dis.row_pitch; a.ll_frame_pitch_d = ll_dis.frame_pitch;
  a.partials = partials;
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  if (scale == 0) {
    if (elem == ELEM_U8) hipLaunchKernelGGL((adm_fixed_kernel<uint8_t, false>), grid, block, 0, stream, a);
    else if (elem == ELEM_U16) hipLaunchKernelGGL((adm_fixed_kernel<uint16_t, false>), grid, block, 0, stream, a);
    else return hipErrorInvalidValue;
  } else {
    if (elem != ELEM_F32) return hipErrorInvalidValue;  // 4-byte planes: the int32 LL band of the previous scale
    hipLaunchKernelGGL((adm_fixed_kernel<int32_t, true>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace pqa
