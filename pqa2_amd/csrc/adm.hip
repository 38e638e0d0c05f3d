// ADM (db2 DWT -> decouple -> contrast-sensitivity weighting -> contrast masking) for gfx950.
//
// Arithmetic follows libvmaf's float extractor (adm.c compute_adm; adm_tools.c adm_dwt2_s,
// adm_decouple_s, adm_csf_s, adm_csf_den_scale_s, adm_cm_s; adm_tools.h dwt_quant_step) -- the code
// behind the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated in
// oracle/vmaf_oracle.c.
//
// One launch = one scale of a batch of frames, fully fused: the only HBM writes are the two
// approximation bands the next scale needs and six partial doubles per tile.  Reference and distorted
// samples travel together as float2 = {ref, dis}, so every DWT tap is one v_pk_fma_f32 for both images
// (FP32 FMA issues at the same rate packed or not on gfx950: tools/ubench/fma_rate.hip).
//   phase 1  vertical DWT: lane <-> input column, rows addressed through SGPR offsets (buffer_load),
//            20 input rows -> 9 output rows of {lo, hi} x {ref, dis} -> LDS
//   phase 2  horizontal DWT (2 x ds_read_b128 per filter), then per coefficient: approximation band
//            store, decouple, CSF; the masking signal summed over orientations -> LDS, |csf(r)| stays
//            in registers; denominator cube sums accumulate immediately
//   phase 3  threshold = 3x3 box of the masking signal + centre, numerator cubes
// The tile carries a one-coefficient halo (66 x 18 for 64 x 16) so phase 3 never leaves LDS.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

struct AdmArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, ow, oh, tiles_x, n_tiles;
  float inv_scale, gain_limit, rf_hv, rf_d;
  float k_hv, k_d;  // rf / 30: CSF weight and the masking signal's 1/30 in one constant
  int left, top, right, bottom;  // cropped accumulation window in band coordinates
  float* ll_ref;
  float* ll_dis;
  int64_t ll_row_pitch_r, ll_frame_pitch_r, ll_row_pitch_d, ll_frame_pitch_d;
  double* partials;
};

constexpr int TW = kAdmTileW, TH = kAdmTileH, GW = TW + 2, GH = TH + 2, NRP = GH / 2;
constexpr int VC = 2 * GW + 2, VP = 128;  // vertical-pass columns (126) / LDS pitch (float2)
static_assert(VC <= 128 && VC <= VP && GH % 2 == 0 && GW <= 64, "one column per lane, rows in pairs");
constexpr int SROWS = GH / 2;             // output rows per vertical strip (2 strips)
constexpr int NIN = 2 * SROWS + 2;        // input rows per strip
constexpr int NROUND = NRP / 4;           // phase 2/3 rounds: 4 waves x 1 row pair each per round
constexpr int GP = 68;                    // pitch of the masking-signal array

__device__ __forceinline__ f2 splat(float c) { return f2{c, c}; }
__device__ __forceinline__ f2 rcp2(f2 x) { return f2{fast_rcp(x.x), fast_rcp(x.y)}; }
__device__ __forceinline__ f2 clamp01_2(f2 k) {
  return f2{__builtin_amdgcn_fmed3f(k.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(k.y, 0.0f, 1.0f)};
}
__device__ __forceinline__ f2 med3_2(f2 a, f2 b, f2 c) {
  return f2{__builtin_amdgcn_fmed3f(a.x, b.x, c.x), __builtin_amdgcn_fmed3f(a.y, b.y, c.y)};
}

// Layout note.  Everything after the vertical DWT works on float2 = {row 2p, row 2p+1}: two vertically
// adjacent coefficients of the same column.  The vertical pass produces {ref, dis} pairs (one packed FMA
// per tap for both images) and re-pairs them by rows when it stores to LDS, so the horizontal DWT reads
// natural row pairs and the whole decouple / CSF / masking chain runs packed on two coefficients.
// LLONLY: the tile has no coefficient inside the accumulation window (libvmaf crops 10 % on every side: a third of the
// tiles of a frame).  Its six sums are zero by definition; all it owes is its piece of the approximation bands the next
// scale reads: the low band of the vertical pass, two of the eight horizontal filters, no decouple / CSF / masking.
template <typename T, bool LLONLY>
__device__ __forceinline__ void adm_tile(const AdmArgs& a, f2 (*V)[NRP][VP], float (*G)[GP], double* red, int tile, int tx,
                                         int ty, int fr) {
  const float lo0 = 0.482962913144690f, lo1 = 0.836516303737469f, lo2 = 0.224143868041857f,
              lo3 = -0.129409522550921f;
  const float hi0 = -0.129409522550921f, hi1 = -0.224143868041857f, hi2 = 0.836516303737469f,
              hi3 = -0.482962913144690f;

  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int cx0 = tx * TW, cy0 = ty * TH;
  const int tid = threadIdx.x;
  const unsigned pitch_r = (unsigned)a.row_pitch_r, pitch_d = (unsigned)a.row_pitch_d;
  const rsrc_t rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r * (unsigned)sizeof(T));
  const rsrc_t rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d * (unsigned)sizeof(T));

  // ---- phase 1: vertical DWT -------------------------------------------------------------------
  // VC (126) columns x 2 strips of 8 output rows: one pass, every wave busy, rows wave-uniform
  {
    const int col = tid & 127;
    const int strip = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (col < VC) {
      const unsigned gx = (unsigned)mirror1(2 * cx0 - 3 + col, a.w);
      f2 x[NIN];
      // Row offsets are wave-uniform (SGPRs).  A strip whose NIN input rows all lie inside the image -- every strip
      // but the first and last tile rows' -- needs no mirroring: offset = (y0 + j) * pitch with j a literal, 2 scalar
      // ops per row instead of 8 (the mirror's abs / compare / select / clamp plus two multiplies).  The branch is
      // wave-uniform; both arms feed the same DWT code below.
      const int y0 = 2 * cy0 - 3 + 2 * SROWS * strip;
      if (y0 >= 0 && y0 + NIN <= a.h) {
        const unsigned base_r = (unsigned)y0 * pitch_r, base_d = (unsigned)y0 * pitch_d;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const T r = buf_load<T>(rsrc_r, gx, base_r + (unsigned)j * pitch_r);  // row offset rides in an SGPR
          const T d = buf_load<T>(rsrc_d, gx, base_d + (unsigned)j * pitch_d);
          x[j] = PixIO<T>::pair(r, d, a.inv_scale);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const unsigned gy = (unsigned)mirror1(y0 + j, a.h);
          const T r = buf_load<T>(rsrc_r, gx, gy * pitch_r);
          const T d = buf_load<T>(rsrc_d, gx, gy * pitch_d);
          x[j] = PixIO<T>::pair(r, d, a.inv_scale);
        }
      }
#pragma unroll
      for (int p = 0; p < SROWS / 2; ++p) {
        f2 vl[2], vh[2];  // {ref, dis} for rows 2p and 2p+1 of the strip
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int o = 2 * p + e;
          // taps accumulate in libvmaf's order: ((c0*s0 + c1*s1) + c2*s2) + c3*s3
          f2 l = splat(lo0) * x[2 * o];
          l = __builtin_elementwise_fma(splat(lo1), x[2 * o + 1], l);
          l = __builtin_elementwise_fma(splat(lo2), x[2 * o + 2], l);
          l = __builtin_elementwise_fma(splat(lo3), x[2 * o + 3], l);
          vl[e] = l;
          if (!LLONLY) {
            f2 h = splat(hi0) * x[2 * o];
            h = __builtin_elementwise_fma(splat(hi1), x[2 * o + 1], h);
            h = __builtin_elementwise_fma(splat(hi2), x[2 * o + 2], h);
            h = __builtin_elementwise_fma(splat(hi3), x[2 * o + 3], h);
            vh[e] = h;
          }
        }
        const int rp = strip * (SROWS / 2) + p;
        V[0][rp][col] = f2{vl[0].x, vl[1].x};
        V[2][rp][col] = f2{vl[0].y, vl[1].y};
        if (!LLONLY) {
          V[1][rp][col] = f2{vh[0].x, vh[1].x};
          V[3][rp][col] = f2{vh[0].y, vh[1].y};
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: horizontal DWT, decouple, CSF on row pairs ----------------------------------------
  // lane <-> halo'd column (62 of 64 used), wave + 4*round <-> row pair
  const float cos_1deg_sq = 0.99969541350954788f;  // cos(pi/180)^2
  const int lcx = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // row pairs are per wave: row tests stay scalar
  const bool have = lcx < GW;
  // lanes 62, 63 have no column.  They used to read column 0: in their ds_read_b128 service group {36-43, 48-51,
  // 60-63} that address shares banks with lanes 48-51 -> a 2-way conflict on every V read (12.7 % of LDS cycles,
  // profiles/r01f_2160p_sq_counters.txt).  Reading lane 61's address instead is a broadcast: free.
  const int lcxs = have ? lcx : GW - 1;
  const int cx = cx0 - 1 + lcxs;
  const bool col_valid = have && cx >= 0 && cx < a.ow;
  const bool col_inner = col_valid && lcxs >= 1 && lcxs <= TW;
  const bool col_win = col_inner && cx >= a.left && cx < a.right;
  f2 xs[NROUND][3];
  f2 mwin[NROUND];
  f2 den_h = f2{0.0f, 0.0f}, den_v = den_h, den_d = den_h;
#pragma unroll
  for (int k = 0; k < NROUND; ++k) {
    const int rp = wave + 4 * k;
    const int lcyA = 2 * rp, cyA = cy0 - 1 + lcyA, cyB = cyA + 1;
    f2 s[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (LLONLY && (q & 1)) continue;   // the high vertical bands were not produced
      const f4* p = reinterpret_cast<const f4*>(&V[q][rp][2 * lcxs]);
      const f4 v01 = p[0], v23 = p[1];
      s[q][0] = f2{v01.x, v01.y}; s[q][1] = f2{v01.z, v01.w};
      s[q][2] = f2{v23.x, v23.y}; s[q][3] = f2{v23.z, v23.w};
    }
#define PQA_DWT(c0, c1, c2, c3, t)                                                                        \
  __builtin_elementwise_fma(splat(c3), t[3],                                                              \
                            __builtin_elementwise_fma(splat(c2), t[2], __builtin_elementwise_fma(splat(c1), t[1], splat(c0) * t[0])))
    const f2 ra = PQA_DWT(lo0, lo1, lo2, lo3, s[0]);  // reference approximation, rows {A, B}
    const f2 da = PQA_DWT(lo0, lo1, lo2, lo3, s[2]);
    const bool vA = col_valid && cyA >= 0 && cyA < a.oh, vB = col_valid && cyB >= 0 && cyB < a.oh;
    const bool iA = col_inner && lcyA >= 1 && cyA < a.oh;                    // lcyA <= TH always (lcyA <= GH-2)
    const bool iB = col_inner && lcyA + 1 <= TH && cyB >= 0 && cyB < a.oh;   // lcyA + 1 >= 1 always
    if (a.ll_ref) {
      float* lr = a.ll_ref + (int64_t)fr * a.ll_frame_pitch_r;
      float* ld = a.ll_dis + (int64_t)fr * a.ll_frame_pitch_d;
      if (iA) {
        lr[(unsigned)cyA * (unsigned)a.ll_row_pitch_r + (unsigned)cx] = ra.x;
        ld[(unsigned)cyA * (unsigned)a.ll_row_pitch_d + (unsigned)cx] = da.x;
      }
      if (iB) {
        lr[(unsigned)cyB * (unsigned)a.ll_row_pitch_r + (unsigned)cx] = ra.y;
        ld[(unsigned)cyB * (unsigned)a.ll_row_pitch_d + (unsigned)cx] = da.y;
      }
    }
    if (LLONLY) continue;
    const f2 ov = PQA_DWT(hi0, hi1, hi2, hi3, s[0]);  // vertical   (lo-v, hi-h)
    const f2 oh = PQA_DWT(lo0, lo1, lo2, lo3, s[1]);  // horizontal (hi-v, lo-h)
    const f2 od = PQA_DWT(hi0, hi1, hi2, hi3, s[1]);  // diagonal
    const f2 tv = PQA_DWT(hi0, hi1, hi2, hi3, s[2]);
    const f2 th = PQA_DWT(lo0, lo1, lo2, lo3, s[3]);
    const f2 td = PQA_DWT(hi0, hi1, hi2, hi3, s[3]);
#undef PQA_DWT
    // decouple + enhancement-gain limit.  libvmaf computes k = clamp(t / (o + eps), 0, 1), r = k * o and, where the angle
    // between (o_h, o_v) and (t_h, t_v) is below one degree, r = min(r * limit, t) for r > 0 / max(r * limit, t) for r < 0.
    // In exact arithmetic k * o is
    //   t  when t and o have the same sign and |t| <= |o|   (0 <= t/o <= 1)
    //   o  when they have the same sign and |t| >  |o|     (t/o > 1 -> k = 1)
    //   0  when the signs differ or o == 0                 (t/o < 0 -> k = 0; o = 0: k * 0)
    // i.e. the median of {0, t, o}; and the limited value, case by case (limit >= 1), is the median of {0, t, limit * o}:
    //   same sign, |t| <= limit |o|  ->  t      (r was t, or r was o and min(o * limit, t) = t)
    //   same sign, |t| >  limit |o|  ->  limit o
    //   signs differ / o == 0        ->  0      ("r == 0 stays")
    // So r = med3(0, t, o * m) with m = limit under the angle test and 1 otherwise: one select per coefficient, one
    // multiply and ONE v_med3_f32 per orientation, no division.  (Round 1: v_rcp_f32 + Newton step + clamp + multiply for k,
    // then a second median and a select per orientation.)  libvmaf's f32 r = fl(fl(t/o) * o) sits within one ulp of t in
    // the first case; the median returns t itself, which is what the f64 oracle gets too.
    const f2 ot_dp = __builtin_elementwise_fma(ov, tv, oh * th);
    const f2 o_mag_sq = __builtin_elementwise_fma(ov, ov, oh * oh), t_mag_sq = __builtin_elementwise_fma(tv, tv, th * th);
    const f2 lhs = ot_dp * ot_dp, rhs = splat(cos_1deg_sq) * o_mag_sq * t_mag_sq;
    const bool angA = (ot_dp.x >= 0.0f) && (lhs.x >= rhs.x), angB = (ot_dp.y >= 0.0f) && (lhs.y >= rhs.y);
    const f2 z2 = splat(0.0f);
    const f2 m2 = f2{angA ? a.gain_limit : 1.0f, angB ? a.gain_limit : 1.0f};
    const f2 rh = med3_2(z2, th, oh * m2), rv = med3_2(z2, tv, ov * m2), rd = med3_2(z2, td, od * m2);
    // CSF of the additive image; adm_cm_s sums the 3x3 boxes per orientation and then over
    // orientations -- summing over orientations first is the same value up to float rounding
    // (|rf_hv a_h| + |rf_hv a_v| + |rf_d a_d|) / 30 with the constants folded: k_hv (|a_h| + |a_v|) + k_d |a_d|
    const f2 ah = th - rh, av = tv - rv, ad = td - rd;
    const float gA = fmaf(a.k_d, fabsf(ad.x), a.k_hv * (fabsf(ah.x) + fabsf(av.x)));
    const float gB = fmaf(a.k_d, fabsf(ad.y), a.k_hv * (fabsf(ah.y) + fabsf(av.y)));
    const bool wA = iA && col_win && cyA >= a.top && cyA < a.bottom;
    const bool wB = iB && col_win && cyB >= a.top && cyB < a.bottom;
    const f2 mw = f2{wA ? 1.0f : 0.0f, wB ? 1.0f : 0.0f};
    mwin[k] = mw;
    xs[k][0] = rh * splat(a.rf_hv);   // signed; |.| is applied where it is used
    xs[k][1] = rv * splat(a.rf_hv);
    xs[k][2] = rd * splat(a.rf_d);
    // denominator: sum |rf o|^3 = rf^3 sum |o|^3 -- the CSF factor leaves the loop (applied to the tile sum below)
    const f2 qh = mw * (oh * oh), qv = mw * (ov * ov), qd = mw * (od * od);
    den_h = f2{fmaf(qh.x, fabsf(oh.x), den_h.x), fmaf(qh.y, fabsf(oh.y), den_h.y)};
    den_v = f2{fmaf(qv.x, fabsf(ov.x), den_v.x), fmaf(qv.y, fabsf(ov.y), den_v.y)};
    den_d = f2{fmaf(qd.x, fabsf(od.x), den_d.x), fmaf(qd.y, fabsf(od.y), den_d.y)};
    if (have) {
      G[lcyA][lcx] = vA ? gA : 0.0f;
      G[lcyA + 1][lcx] = vB ? gB : 0.0f;
    }
  }
  if (LLONLY) {
    if (tid < 6) a.partials[((int64_t)fr * a.n_tiles + tile) * 6 + tid] = 0.0;
    return;
  }
  __syncthreads();

  // ---- phase 3: contrast masking on row pairs ------------------------------------------------------
  f2 num_h = f2{0.0f, 0.0f}, num_v = num_h, num_d = num_h;
  {
    // band-level mirror of the 3x3 neighbourhood (only bites when the window touches the border)
    // (clamped into the written part of G: the halo columns themselves look one column further out, carry
    // weight 0, and must not pull uninitialised LDS -- possibly NaN bits -- into a 0 * x product)
    // (lanes 62, 63 carry lane 61's cx: the same addresses as lane 61 -> broadcasts, no bank conflict)
    const int lx0 = min(max(mirror1(cx - 1, a.ow) - (cx0 - 1), 0), GW - 1);
    const int lx2 = min(max(mirror1(cx + 1, a.ow) - (cx0 - 1), 0), GW - 1);
#pragma unroll
    for (int k = 0; k < NROUND; ++k) {
      const int rp = wave + 4 * k;
      const int lcyA = 2 * rp, cyA = cy0 - 1 + lcyA;
      // rows A-1, A, B, B+1 of the masking signal, clamped into the tile's halo'd grid (the clamped ones
      // belong to elements that are not inner and carry weight 0)
      int ly[4] = {mirror1(cyA - 1, a.oh) - (cy0 - 1), lcyA, lcyA + 1, mirror1(cyA + 2, a.oh) - (cy0 - 1)};
      ly[0] = min(max(ly[0], 0), GH - 1);
      ly[3] = min(max(ly[3], 0), GH - 1);
      float rs[4], c[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = G[ly[j]][lcxs];
        rs[j] = G[ly[j]][lx0] + m + G[ly[j]][lx2];
        if (j == 1) c[0] = m;
        if (j == 2) c[1] = m;
      }
      // interior: A's lower neighbour is B and B's upper neighbour is A.  At the band's last row the mirror
      // of A+1 is A itself; at the band's first row (B = 0, A above the band) the mirror of B-1 is B+1.
      const float downA = (cyA + 1 < a.oh) ? rs[2] : rs[1];
      const float upB = (cyA >= 0) ? rs[1] : rs[3];
      const f2 thr = f2{(rs[0] + rs[1] + downA) + c[0], (upB + rs[2] + rs[3]) + c[1]};
      const f2 mw = mwin[k];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f2 x = f2{fmaxf(fabsf(xs[k][t].x) - thr.x, 0.0f), fmaxf(fabsf(xs[k][t].y) - thr.y, 0.0f)};
        const f2 x3 = (mw * x) * (x * x);
        if (t == 0) num_h += x3;
        if (t == 1) num_v += x3;
        if (t == 2) num_d += x3;
      }
    }
  }
  const float rf3_hv = a.rf_hv * a.rf_hv * a.rf_hv, rf3_d = a.rf_d * a.rf_d * a.rf_d;
  const float part[6] = {num_h.x + num_h.y, num_v.x + num_v.y, num_d.x + num_d.y,
                         rf3_hv * (den_h.x + den_h.y), rf3_hv * (den_v.x + den_v.y), rf3_d * (den_d.x + den_d.y)};
  double v[6];
  block_sum_f32<6>(part, v, red);
  if (tid == 0) {
    double* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] = v[i];
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock, 4) void adm_scale_kernel(const AdmArgs a) {
  __shared__ f2 V[4][NRP][VP];  // 0 lo(ref) 1 hi(ref) 2 lo(dis) 3 hi(dis); {row 2p, row 2p+1}
  __shared__ float G[GH][GP];   // masking signal: sum over orientations of |csf(a)| / 30
  __shared__ double red[24];
  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const int cx0 = tx * TW, cy0 = ty * TH;
  // no inner coefficient of this tile inside [left, right) x [top, bottom)?  (workgroup-uniform)
  const bool outside = !(cx0 < a.right && cx0 + TW > a.left && cy0 < a.bottom && cy0 + TH > a.top);
  if (outside) {
    if (!a.ll_ref) {   // last scale: nothing to produce at all
      if (threadIdx.x < 6) a.partials[((int64_t)fr * a.n_tiles + tile) * 6 + threadIdx.x] = 0.0;
      return;
    }
    adm_tile<T, true>(a, V, G, red, tile, tx, ty, fr);
  } else {
    adm_tile<T, false>(a, V, G, red, tile, tx, ty, fr);
  }
}

// Watson DWT 7/9 noise-floor model (adm_tools.h dwt_quant_step), Y channel, view distance 3 H,
// 1080-line display.  theta 1 = h/v, theta 2 = d.
static float dwt_quant_step(int lambda, int theta) {
  static const double g[3] = {1.501, 1.0, 0.534};
  static const double amp[4][3] = {{0.62171, 0.67234, 0.72709},
                                   {0.34537, 0.41317, 0.49428},
                                   {0.18004, 0.22727, 0.28688},
                                   {0.091401, 0.11792, 0.15214}};
  const float r = (float)(3.0 * 1080 * M_PI / 180.0);
  const float temp = (float)log10(pow(2.0, lambda + 1) * 0.401 * g[theta] / (double)r);
  return (float)(2.0 * 0.495 * pow(10.0, 0.466 * (double)temp * (double)temp) / amp[lambda][theta]);
}

}  // namespace

float adm_dwt_quant_step(int lambda, int theta) { return dwt_quant_step(lambda, theta); }

hipError_t launch_adm_scale(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                            int w, int h, float inv_scale, float gain_limit, MutPlaneRun ll_ref,
                            MutPlaneRun ll_dis, double* partials, int mode, int* n_partials) {
  if (n_partials) *n_partials = adm_tiles_x((w + 1) / 2) * adm_tiles_y((h + 1) / 2);
  if (n_frames <= 0) return hipSuccess;
  if (mode != ADM_TILED) {
    hipError_t err = hipSuccess;
    if (launch_adm_march(stream, scale, elem, ref, dis, n_frames, w, h, inv_scale, gain_limit, ll_ref, ll_dis, partials,
                         n_partials, &err))
      return err;
  }
  AdmArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.ow = (w + 1) / 2; a.oh = (h + 1) / 2;
  a.tiles_x = adm_tiles_x(a.ow);
  a.n_tiles = a.tiles_x * adm_tiles_y(a.oh);
  a.inv_scale = inv_scale; a.gain_limit = gain_limit;
  a.rf_hv = 1.0f / dwt_quant_step(scale, 1);
  a.rf_d = 1.0f / dwt_quant_step(scale, 2);
  a.k_hv = a.rf_hv / 30.0f;
  a.k_d = a.rf_d / 30.0f;
  const double border = 0.1;  // ADM_BORDER_FACTOR
  a.left = (int)(a.ow * border - 0.5);
  a.top = (int)(a.oh * border - 0.5);
  a.right = a.ow - a.left;
  a.bottom = a.oh - a.top;
  a.ll_ref = (float*)ll_ref.base; a.ll_dis = (float*)ll_dis.base;
  a.ll_row_pitch_r = ll_ref.row_pitch; a.ll_frame_pitch_r = ll_ref.frame_pitch;
  a.ll_row_pitch_d = ll_dis.row_pitch; a.ll_frame_pitch_d = ll_dis.frame_pitch;
  a.partials = partials;
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((adm_scale_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((adm_scale_kernel<uint16_t>), grid, block, 0, stream, a); break;
    case ELEM_F32: hipLaunchKernelGGL((adm_scale_kernel<float>), grid, block, 0, stream, a); break;
  }
  return hipGetLastError();
}

}  // namespace pqa
