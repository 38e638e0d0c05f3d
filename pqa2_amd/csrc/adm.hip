// ADM (db2 DWT -> decouple -> contrast-sensitivity weighting -> contrast masking) for gfx950.
//
// Arithmetic follows libvmaf's float extractor (adm.c compute_adm; adm_tools.c adm_dwt2_s,
// adm_decouple_s, adm_csf_s, adm_csf_den_scale_s, adm_cm_s; adm_tools.h dwt_quant_step) -- the code
// behind the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated in
// oracle/vmaf_oracle.c.
//
// One launch = one scale of a batch of frames, fully fused: the only HBM writes are the two
// approximation bands the next scale needs and six partial doubles per tile.
//   phase 1  vertical DWT: lane <-> input column, 14 input rows streamed per thread, lo/hi of ref and
//            dis for 6 output rows -> LDS V[4][18][136]
//   phase 2  horizontal DWT from LDS (2 x ds_read_b64 per array), then per coefficient: approximation
//            band store, decouple, CSF; masking signal F = |csf(a)|/30 -> LDS, |csf(r)| stays in
//            registers; denominator cube sums accumulate immediately
//   phase 3  threshold = sum over orientations of (3x3 box of F + centre) from LDS, numerator cubes
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
constexpr int VC = 2 * TW + 6, VP = 136;  // vertical-pass columns / LDS pitch
constexpr int NITEM = GW * GH, NROUND = (NITEM + kBlock - 1) / kBlock;

__device__ __forceinline__ float clamp01(float k) { return k < 0.0f ? 0.0f : (k > 1.0f ? 1.0f : k); }

template <typename T>
__global__ __launch_bounds__(kBlock) void adm_scale_kernel(const AdmArgs a) {
  __shared__ float V[4][GH][VP];  // vlo_ref, vhi_ref, vlo_dis, vhi_dis
  __shared__ float F[3][GH][GW];  // masking signal per orientation
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

  // ---- phase 1: vertical DWT ---------------------------------------------------------------
  for (int item = tid; item < VC * 3; item += kBlock) {
    const int col = item % VC, strip = item / VC;
    const int gx = mirror(2 * cx0 - 3 + col, a.w);
    float r[14], d[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int gy = mirror(2 * cy0 - 3 + 12 * strip + j, a.h);
      r[j] = PixIO<T>::load(ref + (int64_t)gy * a.row_pitch_r + gx, a.inv_scale);
      d[j] = PixIO<T>::load(dis + (int64_t)gy * a.row_pitch_d + gx, a.inv_scale);
    }
#pragma unroll
    for (int o = 0; o < 6; ++o) {
      const int lr = strip * 6 + o;
      const float r0 = r[2 * o], r1 = r[2 * o + 1], r2 = r[2 * o + 2], r3 = r[2 * o + 3];
      const float d0 = d[2 * o], d1 = d[2 * o + 1], d2 = d[2 * o + 2], d3 = d[2 * o + 3];
      V[0][lr][col] = fmaf(lo3, r3, fmaf(lo2, r2, fmaf(lo1, r1, lo0 * r0)));
      V[1][lr][col] = fmaf(hi3, r3, fmaf(hi2, r2, fmaf(hi1, r1, hi0 * r0)));
      V[2][lr][col] = fmaf(lo3, d3, fmaf(lo2, d2, fmaf(lo1, d1, lo0 * d0)));
      V[3][lr][col] = fmaf(hi3, d3, fmaf(hi2, d2, fmaf(hi1, d1, hi0 * d0)));
    }
  }
  __syncthreads();

  // ---- phase 2: horizontal DWT, decouple, CSF ------------------------------------------------
  const float cos_1deg_sq = 0.99969541350954788f;  // cos(pi/180)^2
  const float eps = 1e-30f;
  float xs[NROUND][3];
  float den_h = 0.0f, den_v = 0.0f, den_d = 0.0f;
  unsigned acc_mask = 0;
#pragma unroll
  for (int k = 0; k < NROUND; ++k) {
    const int item = tid + k * kBlock;
    xs[k][0] = xs[k][1] = xs[k][2] = 0.0f;
    if (item < NITEM) {
      const int lcy = item / GW, lcx = item - lcy * GW;
      const int cx = cx0 - 1 + lcx, cy = cy0 - 1 + lcy;
      const bool valid = cx >= 0 && cx < a.ow && cy >= 0 && cy < a.oh;
      float f_h = 0.0f, f_v = 0.0f, f_d = 0.0f;
      if (valid) {
        float b[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float2 p0 = *reinterpret_cast<const float2*>(&V[q][lcy][2 * lcx]);
          const float2 p1 = *reinterpret_cast<const float2*>(&V[q][lcy][2 * lcx + 2]);
          b[q][0] = p0.x; b[q][1] = p0.y; b[q][2] = p1.x; b[q][3] = p1.y;
        }
#define PQA_LO(s) fmaf(lo3, s[3], fmaf(lo2, s[2], fmaf(lo1, s[1], lo0 * s[0])))
#define PQA_HI(s) fmaf(hi3, s[3], fmaf(hi2, s[2], fmaf(hi1, s[1], hi0 * s[0])))
        const float ra = PQA_LO(b[0]), ov = PQA_HI(b[0]), oh = PQA_LO(b[1]), od = PQA_HI(b[1]);
        const float da = PQA_LO(b[2]), tv = PQA_HI(b[2]), th = PQA_LO(b[3]), td = PQA_HI(b[3]);
#undef PQA_LO
#undef PQA_HI
        const bool inner = lcx >= 1 && lcx <= TW && lcy >= 1 && lcy <= TH;
        if (inner && a.ll_ref) {
          a.ll_ref[(int64_t)fr * a.ll_frame_pitch_r + (int64_t)cy * a.ll_row_pitch_r + cx] = ra;
          a.ll_dis[(int64_t)fr * a.ll_frame_pitch_d + (int64_t)cy * a.ll_row_pitch_d + cx] = da;
        }
        // decouple
        const float kh = clamp01(th / (oh + eps)), kv = clamp01(tv / (ov + eps)), kd = clamp01(td / (od + eps));
        float rh = kh * oh, rv = kv * ov, rd = kd * od;
        const float ot_dp = oh * th + ov * tv;
        const float o_mag_sq = oh * oh + ov * ov, t_mag_sq = th * th + tv * tv;
        const bool angle_flag = (ot_dp >= 0.0f) && (ot_dp * ot_dp >= cos_1deg_sq * o_mag_sq * t_mag_sq);
        if (angle_flag) {
          if (rh > 0.0f) rh = fminf(rh * a.gain_limit, th); else if (rh < 0.0f) rh = fmaxf(rh * a.gain_limit, th);
          if (rv > 0.0f) rv = fminf(rv * a.gain_limit, tv); else if (rv < 0.0f) rv = fmaxf(rv * a.gain_limit, tv);
          if (rd > 0.0f) rd = fminf(rd * a.gain_limit, td); else if (rd < 0.0f) rd = fmaxf(rd * a.gain_limit, td);
        }
        const float ah = th - rh, av = tv - rv, ad = td - rd;
        // CSF of the additive image -> masking signal
        f_h = (1.0f / 30.0f) * fabsf(a.rf_hv * ah);
        f_v = (1.0f / 30.0f) * fabsf(a.rf_hv * av);
        f_d = (1.0f / 30.0f) * fabsf(a.rf_d * ad);
        if (inner && cx >= a.left && cx < a.right && cy >= a.top && cy < a.bottom) {
          acc_mask |= 1u << k;
          xs[k][0] = fabsf(rh * a.rf_hv);
          xs[k][1] = fabsf(rv * a.rf_hv);
          xs[k][2] = fabsf(rd * a.rf_d);
          const float vh = fabsf(oh) * a.rf_hv, vv = fabsf(ov) * a.rf_hv, vd = fabsf(od) * a.rf_d;
          den_h += vh * vh * vh;
          den_v += vv * vv * vv;
          den_d += vd * vd * vd;
        }
      }
      F[0][lcy][lcx] = f_h;
      F[1][lcy][lcx] = f_v;
      F[2][lcy][lcx] = f_d;
    }
  }
  __syncthreads();

  // ---- phase 3: contrast masking -------------------------------------------------------------
  float num_h = 0.0f, num_v = 0.0f, num_d = 0.0f;
#pragma unroll
  for (int k = 0; k < NROUND; ++k) {
    if (acc_mask & (1u << k)) {
      const int item = tid + k * kBlock;
      const int lcy = item / GW, lcx = item - lcy * GW;
      const int cx = cx0 - 1 + lcx, cy = cy0 - 1 + lcy;
      // band-level mirror of the 3x3 neighbourhood (only bites when the window touches the border)
      const int ly[3] = {mirror(cy - 1, a.oh) - (cy0 - 1), lcy, mirror(cy + 1, a.oh) - (cy0 - 1)};
      const int lx[3] = {mirror(cx - 1, a.ow) - (cx0 - 1), lcx, mirror(cx + 1, a.ow) - (cx0 - 1)};
      float thr = 0.0f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float sum1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) sum1 += F[t][ly[i]][lx[j]];
        sum1 += F[t][lcy][lcx];
        thr += sum1;
      }
      float xh = xs[k][0] - thr, xv = xs[k][1] - thr, xd = xs[k][2] - thr;
      xh = xh < 0.0f ? 0.0f : xh;
      xv = xv < 0.0f ? 0.0f : xv;
      xd = xd < 0.0f ? 0.0f : xd;
      num_h += xh * xh * xh;
      num_v += xv * xv * xv;
      num_d += xd * xd * xd;
    }
  }
  double v[6] = {(double)num_h, (double)num_v, (double)num_d, (double)den_h, (double)den_v, (double)den_d};
  block_sum<6>(v, red);
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
