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
  int left, top, right, bottom;  // cropped accumulation window in band coordinates
  float* ll_ref;
  float* ll_dis;
  int64_t ll_row_pitch_r, ll_frame_pitch_r, ll_row_pitch_d, ll_frame_pitch_d;
  double* partials;
};

constexpr int TW = kAdmTileW, TH = kAdmTileH, GW = TW + 2, GH = TH + 2;
constexpr int VC = 2 * GW + 2, VP = 128;  // vertical-pass columns (126) / LDS pitch (float2)
static_assert(VC <= 128 && VC <= VP, "one column per lane, two strips per workgroup");
constexpr int SROWS = GH / 2;             // output rows per vertical strip (2 strips)
constexpr int NIN = 2 * SROWS + 2;        // input rows per strip
constexpr int NROUND = 5;                 // phase 2/3 rounds: 4.5 x (64 cols x 4 rows) + halo columns
constexpr int GP = GW + 3;                // pitch of the masking-signal array

__device__ __forceinline__ f2 splat(float c) { return f2{c, c}; }

template <typename T>
__global__ __launch_bounds__(kBlock, 3) void adm_scale_kernel(const AdmArgs a) {
  __shared__ f2 Vlo[GH][VP];    // vertical low-pass  {ref, dis}
  __shared__ f2 Vhi[GH][VP];    // vertical high-pass {ref, dis}
  __shared__ float G[GH][GP];   // masking signal: sum over orientations of |csf(a)| / 30
  __shared__ double red[24];

  const float lo0 = 0.482962913144690f, lo1 = 0.836516303737469f, lo2 = 0.224143868041857f,
              lo3 = -0.129409522550921f;
  const float hi0 = -0.129409522550921f, hi1 = -0.224143868041857f, hi2 = 0.836516303737469f,
              hi3 = -0.482962913144690f;

  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int cx0 = tx * TW, cy0 = ty * TH;
  const int tid = threadIdx.x;
  const unsigned pitch_r = (unsigned)a.row_pitch_r, pitch_d = (unsigned)a.row_pitch_d;
  const rsrc_t rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r * (unsigned)sizeof(T));
  const rsrc_t rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d * (unsigned)sizeof(T));

  // ---- phase 1: vertical DWT -------------------------------------------------------------------
  // items: VC (126) columns x 2 strips of 9 output rows: one pass, every wave busy, rows wave-uniform
  {
    const int col = tid & 127;
    const int strip = __builtin_amdgcn_readfirstlane(tid >> 7);
    if (col < VC) {
    const unsigned gx = (unsigned)mirror1(2 * cx0 - 3 + col, a.w);
    f2 x[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      const unsigned gy = (unsigned)mirror1(2 * cy0 - 3 + 2 * SROWS * strip + j, a.h);
      const T r = buf_load<T>(rsrc_r, gx, gy * pitch_r);  // row offset rides in an SGPR
      const T d = buf_load<T>(rsrc_d, gx, gy * pitch_d);
      x[j] = PixIO<T>::pair(r, d, a.inv_scale);
    }
#pragma unroll
    for (int o = 0; o < SROWS; ++o) {
      const int lr = strip * SROWS + o;
      // taps accumulate in libvmaf's order: ((c0*s0 + c1*s1) + c2*s2) + c3*s3
      f2 vl = splat(lo0) * x[2 * o], vh = splat(hi0) * x[2 * o];
      vl = __builtin_elementwise_fma(splat(lo1), x[2 * o + 1], vl);
      vh = __builtin_elementwise_fma(splat(hi1), x[2 * o + 1], vh);
      vl = __builtin_elementwise_fma(splat(lo2), x[2 * o + 2], vl);
      vh = __builtin_elementwise_fma(splat(hi2), x[2 * o + 2], vh);
      vl = __builtin_elementwise_fma(splat(lo3), x[2 * o + 3], vl);
      vh = __builtin_elementwise_fma(splat(hi3), x[2 * o + 3], vh);
      Vlo[lr][col] = vl;
      Vhi[lr][col] = vh;
    }
    }
  }
  __syncthreads();

  // ---- phase 2: horizontal DWT, decouple, CSF ------------------------------------------------
  // grid of (GW x GH) coefficients incl. halo: rounds 0..3 take columns 1..64 x rows 4r..4r+3, round 4
  // takes rows 16,17 of those columns (threads 0..127) and the two halo columns x 18 rows (threads 128..163)
  const float cos_1deg_sq = 0.99969541350954788f;  // cos(pi/180)^2
  const float eps = 1e-30f;
  float xs[NROUND][3];
  float den_h = 0.0f, den_v = 0.0f, den_d = 0.0f;
  unsigned acc_mask = 0;
#pragma unroll
  for (int k = 0; k < NROUND; ++k) {
    int lcx, lcy;
    bool have = true;
    if (k < 4) {
      lcx = 1 + (tid & 63);
      lcy = 4 * k + (tid >> 6);
      have = lcx <= TW;
    } else if (tid < 128) {
      lcx = 1 + (tid & 63);
      lcy = 16 + (tid >> 6);
      have = lcx <= TW;
    } else {
      const int t = tid - 128;
      have = t < 2 * GH;
      lcx = (t & 1) ? GW - 1 : 0;
      lcy = have ? (t >> 1) : 0;
    }
    // Straight-line code from here: every lane computes its coefficient (positions outside the band or
    // the accumulation window hold finite mirrored data) and validity enters as selects / 0-1 weights.
    // hipcc turns the reference's nested ifs into exec-mask branches otherwise (7 per coefficient).
    if (!have) { lcx = 0; lcy = 0; }
    const int cx = cx0 - 1 + lcx, cy = cy0 - 1 + lcy;
    const bool valid = have && cx >= 0 && cx < a.ow && cy >= 0 && cy < a.oh;
    const f4* pl = reinterpret_cast<const f4*>(&Vlo[lcy][2 * lcx]);
    const f4* ph = reinterpret_cast<const f4*>(&Vhi[lcy][2 * lcx]);
    const f4 l01 = pl[0], l23 = pl[1], h01 = ph[0], h23 = ph[1];
    const f2 l0 = f2{l01.x, l01.y}, l1 = f2{l01.z, l01.w}, l2 = f2{l23.x, l23.y}, l3 = f2{l23.z, l23.w};
    const f2 h0 = f2{h01.x, h01.y}, h1 = f2{h01.z, h01.w}, h2 = f2{h23.x, h23.y}, h3 = f2{h23.z, h23.w};
#define PQA_DWT(c0, c1, c2, c3, s0, s1, s2, s3)                                                        \
  __builtin_elementwise_fma(splat(c3), s3,                                                             \
                            __builtin_elementwise_fma(splat(c2), s2, __builtin_elementwise_fma(splat(c1), s1, splat(c0) * s0)))
    const f2 ba = PQA_DWT(lo0, lo1, lo2, lo3, l0, l1, l2, l3);  // {ref, dis} approximation
    const f2 bv = PQA_DWT(hi0, hi1, hi2, hi3, l0, l1, l2, l3);  // vertical   (lo-v, hi-h)
    const f2 bh = PQA_DWT(lo0, lo1, lo2, lo3, h0, h1, h2, h3);  // horizontal (hi-v, lo-h)
    const f2 bd = PQA_DWT(hi0, hi1, hi2, hi3, h0, h1, h2, h3);  // diagonal
#undef PQA_DWT
    const bool inner = valid && lcx >= 1 && lcx <= TW && lcy >= 1 && lcy <= TH;
    if (inner && a.ll_ref) {
      const unsigned off_r = (unsigned)cy * (unsigned)a.ll_row_pitch_r + (unsigned)cx;
      const unsigned off_d = (unsigned)cy * (unsigned)a.ll_row_pitch_d + (unsigned)cx;
      (a.ll_ref + (int64_t)fr * a.ll_frame_pitch_r)[off_r] = ba.x;
      (a.ll_dis + (int64_t)fr * a.ll_frame_pitch_d)[off_d] = ba.y;
    }
    const float oh = bh.x, ov = bv.x, od = bd.x, th = bh.y, tv = bv.y, td = bd.y;
    // decouple: k = clamp(t / (o + eps), 0, 1) via v_rcp_f32 + one Newton step.  o + eps is either 1e-30
    // (o == 0) or |o| >~ 1e-9 (an f32 DWT of bounded samples cannot produce a smaller non-zero value), so the
    // reciprocal stays finite and no NaN can form.
    const float xh = oh + eps, xv = ov + eps, xd = od + eps;
    const float rch = fast_rcp(xh), rcv = fast_rcp(xv), rcd = fast_rcp(xd);
    float kh = th * rch, kv = tv * rcv, kd = td * rcd;
    kh = __builtin_amdgcn_fmed3f(fmaf(fmaf(-kh, xh, th), rch, kh), 0.0f, 1.0f);
    kv = __builtin_amdgcn_fmed3f(fmaf(fmaf(-kv, xv, tv), rcv, kv), 0.0f, 1.0f);
    kd = __builtin_amdgcn_fmed3f(fmaf(fmaf(-kd, xd, td), rcd, kd), 0.0f, 1.0f);
    float rh = kh * oh, rv = kv * ov, rd = kd * od;
    const float ot_dp = oh * th + ov * tv;
    const float o_mag_sq = oh * oh + ov * ov, t_mag_sq = th * th + tv * tv;
    const bool angle_flag = (ot_dp >= 0.0f) && (ot_dp * ot_dp >= cos_1deg_sq * o_mag_sq * t_mag_sq);
    // enhancement-gain limit under the angle test: r > 0 -> min(r*limit, t); r < 0 -> max(r*limit, t);
    // r == 0 stays.  Because r = clamp(t/o, 0, 1) * o lies between 0 and t and limit >= 1, all three cases
    // are the median of {r, r*limit, t}: one v_med3_f32 (differs from the branchy form only when k*o
    // rounds 1 ulp past t).
    rh = angle_flag ? __builtin_amdgcn_fmed3f(rh, rh * a.gain_limit, th) : rh;
    rv = angle_flag ? __builtin_amdgcn_fmed3f(rv, rv * a.gain_limit, tv) : rv;
    rd = angle_flag ? __builtin_amdgcn_fmed3f(rd, rd * a.gain_limit, td) : rd;
    // CSF of the additive image; adm_cm_s sums the 3x3 boxes per orientation and then over
    // orientations -- summing over orientations first is the same value up to float rounding
    const float g = (1.0f / 30.0f) * (fabsf(a.rf_hv * (th - rh)) + fabsf(a.rf_hv * (tv - rv)) + fabsf(a.rf_d * (td - rd)));
    const bool in_win = inner && cx >= a.left && cx < a.right && cy >= a.top && cy < a.bottom;
    const float mw = in_win ? 1.0f : 0.0f;
    acc_mask |= in_win ? (1u << k) : 0u;
    xs[k][0] = fabsf(rh * a.rf_hv);
    xs[k][1] = fabsf(rv * a.rf_hv);
    xs[k][2] = fabsf(rd * a.rf_d);
    const float vh = fabsf(oh) * a.rf_hv, vv = fabsf(ov) * a.rf_hv, vd = fabsf(od) * a.rf_d;
    den_h = fmaf(mw * vh, vh * vh, den_h);
    den_v = fmaf(mw * vv, vv * vv, den_v);
    den_d = fmaf(mw * vd, vd * vd, den_d);
    if (have) G[lcy][lcx] = valid ? g : 0.0f;
  }
  __syncthreads();

  // ---- phase 3: contrast masking -------------------------------------------------------------
  float num_h = 0.0f, num_v = 0.0f, num_d = 0.0f;
#pragma unroll
  for (int k = 0; k < NROUND - 1; ++k) {  // inner coefficients only live in rounds 0..4; round 4 rows 16 only
    if (acc_mask & (1u << k)) {
      const int lcx = 1 + (tid & 63), lcy = 4 * k + (tid >> 6);
      const int cx = cx0 - 1 + lcx, cy = cy0 - 1 + lcy;
      // band-level mirror of the 3x3 neighbourhood (only bites when the window touches the border)
      const int ly0 = mirror1(cy - 1, a.oh) - (cy0 - 1), ly2 = mirror1(cy + 1, a.oh) - (cy0 - 1);
      const int lx0 = mirror1(cx - 1, a.ow) - (cx0 - 1), lx2 = mirror1(cx + 1, a.ow) - (cx0 - 1);
      const float c = G[lcy][lcx];
      float thr = G[ly0][lx0] + G[ly0][lcx] + G[ly0][lx2];
      thr += G[lcy][lx0] + c + G[lcy][lx2];
      thr += G[ly2][lx0] + G[ly2][lcx] + G[ly2][lx2];
      thr += c;
      float xh = xs[k][0] - thr, xv = xs[k][1] - thr, xd = xs[k][2] - thr;
      xh = fmaxf(xh, 0.0f);
      xv = fmaxf(xv, 0.0f);
      xd = fmaxf(xd, 0.0f);
      num_h += xh * xh * xh;
      num_v += xv * xv * xv;
      num_d += xd * xd * xd;
    }
  }
  if (acc_mask & (1u << 4)) {  // row 16 of the halo'd grid (last inner row), threads 0..63
    const int lcx = 1 + (tid & 63), lcy = 16 + (tid >> 6);
    const int cx = cx0 - 1 + lcx, cy = cy0 - 1 + lcy;
    const int ly0 = mirror1(cy - 1, a.oh) - (cy0 - 1), ly2 = mirror1(cy + 1, a.oh) - (cy0 - 1);
    const int lx0 = mirror1(cx - 1, a.ow) - (cx0 - 1), lx2 = mirror1(cx + 1, a.ow) - (cx0 - 1);
    const float c = G[lcy][lcx];
    float thr = G[ly0][lx0] + G[ly0][lcx] + G[ly0][lx2];
    thr += G[lcy][lx0] + c + G[lcy][lx2];
    thr += G[ly2][lx0] + G[ly2][lcx] + G[ly2][lx2];
    thr += c;
    const float xh = fmaxf(xs[4][0] - thr, 0.0f), xv = fmaxf(xs[4][1] - thr, 0.0f), xd = fmaxf(xs[4][2] - thr, 0.0f);
    num_h += xh * xh * xh;
    num_v += xv * xv * xv;
    num_d += xd * xd * xd;
  }
  const float part[6] = {num_h, num_v, num_d, den_h, den_v, den_d};
  double v[6];
  block_sum_f32<6>(part, v, red);
  if (tid == 0) {
    double* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] = v[i];
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

hipError_t launch_adm_scale(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                            int w, int h, float inv_scale, float gain_limit, MutPlaneRun ll_ref,
                            MutPlaneRun ll_dis, double* partials) {
  if (n_frames <= 0) return hipSuccess;
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
