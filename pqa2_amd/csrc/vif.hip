// VIF (4-scale separable-Gaussian pyramid + local mean/variance statistic) for gfx950.
//
// Arithmetic follows libvmaf's float extractor (vif.c compute_vif, vif_tools.c vif_filter1d_s /
// vif_filter1d_sq_s / vif_filter1d_xy_s / vif_dec2_s / vif_statistic_s) -- the code behind the
// reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated in oracle/vmaf_oracle.c.
//
// Kernel shape (one workgroup = 4 waves = one TW x 16 output tile of one frame):
//   1. vertical pass: lane <-> column.  Each thread streams NIN = 8+N-1 input rows of its column
//      straight from HBM/L2 (coalesced row segments, no re-layout), forms r, d, r*r, d*d, r*d and
//      scatters them into 8 x 5 register accumulators in libvmaf's tap order; results go to LDS.
//   2. horizontal pass: lane <-> (row, 8-column segment).  ds_read_b64 with a pitch of 130 floats and
//      segments dealt 4 apart per 16-lane group => conflict-free (bank = 2*row + 32*group).
//      8 x 5 outputs per thread stay in registers and feed the statistic directly.
//   3. statistic + wave-shuffle / LDS block reduction in double -> one (num, den) partial per tile;
//      a fixed-order second stage (finalize.hip) makes 1..8-GPU results bit-identical.
// Nothing but the two partial doubles is written: HBM traffic = the two input planes, once.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {

namespace {

struct Taps {
  float f[17];
};

static Taps gaussian_taps(int n) {
  // N taps, sigma = N/5, normalised in double, stored as float (libvmaf vif_filter1d_table).
  Taps t{};
  double v[17], sum = 0.0;
  const double sigma = n / 5.0;
  for (int k = 0; k < n; ++k) {
    const double d = k - n / 2;
    v[k] = exp(-0.5 * d * d / (sigma * sigma));
    sum += v[k];
  }
  for (int k = 0; k < n; ++k) t.f[k] = (float)(v[k] / sum);
  return t;
}

struct VifStatArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, tiles_x, n_tiles;
  float inv_scale, gain_limit;
  double* partials;
  Taps taps;
};

constexpr int kP = 130;  // LDS row pitch in floats: == 2 (mod 64) -> ds_read_b64 rows land 2 banks apart

template <typename T, int N, int TW>
__global__ __launch_bounds__(kBlock) void vif_stat_kernel(const VifStatArgs a) {
  constexpr int R = N / 2, TH = kVifTileH, COLS = TW + N - 1, NSEG = TW / 8, S = 8, NIN = S + N - 1;
  static_assert(COLS <= 128 && COLS <= kP, "tile too wide");
  static_assert(TW % 8 == 0 && NSEG <= 16, "segment map");
  __shared__ float sv[5][TH][kP];
  __shared__ double red[8];

  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x;

  // ---- 1. vertical pass --------------------------------------------------------------------
  {
    const int col = tid & 127, seg = tid >> 7;
    if (col < COLS) {
      const int gx = mirror(x0 - R + col, a.w);
      float r[NIN], d[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const int gy = mirror(y0 + seg * S - R + j, a.h);
        r[j] = PixIO<T>::load(ref + (int64_t)gy * a.row_pitch_r + gx, a.inv_scale);
        d[j] = PixIO<T>::load(dis + (int64_t)gy * a.row_pitch_d + gx, a.inv_scale);
      }
      float acc[S][5];
#pragma unroll
      for (int o = 0; o < S; ++o)
#pragma unroll
        for (int s = 0; s < 5; ++s) acc[o][s] = 0.0f;
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const float rr = r[j] * r[j], dd = d[j] * d[j], rd = r[j] * d[j];
#pragma unroll
        for (int o = 0; o < S; ++o) {
          const int k = j - o;  // tap index: increases with j, so each output sums taps 0..N-1 in order
          if (k >= 0 && k < N) {
            const float c = a.taps.f[k];
            acc[o][0] = fmaf(c, r[j], acc[o][0]);
            acc[o][1] = fmaf(c, d[j], acc[o][1]);
            acc[o][2] = fmaf(c, rr, acc[o][2]);
            acc[o][3] = fmaf(c, dd, acc[o][3]);
            acc[o][4] = fmaf(c, rd, acc[o][4]);
          }
        }
      }
#pragma unroll
      for (int o = 0; o < S; ++o)
#pragma unroll
        for (int s = 0; s < 5; ++s) sv[s][seg * S + o][col] = acc[o][s];
    }
  }
  __syncthreads();

  // ---- 2. horizontal pass + 3. statistic ------------------------------------------------------
  const int wave = tid >> 6, lane = tid & 63;
  const int row = lane & 15, seg = wave + 4 * (lane >> 4);
  float num = 0.0f, den = 0.0f;
  if (seg < NSEG) {
    constexpr int NREAD = (8 + N - 1 + 1) / 2;  // float2 reads per signal
    float out[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      float in[2 * NREAD];
      const float2* p = reinterpret_cast<const float2*>(&sv[s][row][seg * 8]);
#pragma unroll
      for (int q = 0; q < NREAD; ++q) {
        const float2 v = p[q];
        in[2 * q] = v.x;
        in[2 * q + 1] = v.y;
      }
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) acc = fmaf(a.taps.f[k], in[o + k], acc);
        out[s][o] = acc;
      }
    }
    const int gy = y0 + row;
    const float sigma_nsq = 2.0f, eps = 1.0e-10f, sigma_max_inv = 4.0f / (255.0f * 255.0f);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int gx = x0 + seg * 8 + o;
      const float mu1 = out[0][o], mu2 = out[1][o];
      float sigma1_sq = out[2][o] - mu1 * mu1;
      float sigma2_sq = out[3][o] - mu2 * mu2;
      const float sigma12 = out[4][o] - mu1 * mu2;
      sigma1_sq = fmaxf(sigma1_sq, 0.0f);
      sigma2_sq = fmaxf(sigma2_sq, 0.0f);
      // g = sigma12 / (sigma1_sq + eps): v_rcp_f32 plus one Newton correction -- exact 1.0 when the two
      // are equal (identical frames => vif_scale == 1 exactly, as in libvmaf), 2 FMAs instead of a full div
      const float gden = sigma1_sq + eps, grcp = fast_rcp(gden);
      float g = sigma12 * grcp;
      g = fmaf(fmaf(-g, gden, sigma12), grcp, g);
      float sv_sq = sigma2_sq - g * sigma12;
      if (sigma1_sq < eps) { g = 0.0f; sv_sq = sigma2_sq; sigma1_sq = 0.0f; }
      if (sigma2_sq < eps) { g = 0.0f; sv_sq = 0.0f; }
      if (g < 0.0f) { sv_sq = sigma2_sq; g = 0.0f; }
      sv_sq = fmaxf(sv_sq, eps);
      g = fminf(g, a.gain_limit);
      float num_val = fast_log2(1.0f + (g * g * sigma1_sq) * fast_rcp(sv_sq + sigma_nsq));
      float den_val = fast_log2(1.0f + sigma1_sq * (1.0f / sigma_nsq));
      if (sigma12 < 0.0f) num_val = 0.0f;
      if (sigma1_sq < sigma_nsq) { num_val = 1.0f - sigma2_sq * sigma_max_inv; den_val = 1.0f; }
      if (gx < a.w && gy < a.h) { num += num_val; den += den_val; }
    }
  }
  double v[2] = {(double)num, (double)den};
  block_sum<2>(v, red);
  if (tid == 0) {
    double* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * 2;
    out[0] = v[0];
    out[1] = v[1];
  }
}

// ---- decimation: filter with the destination scale's kernel, keep even samples ----------------
struct VifDecArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, ow, oh, tiles_x, n_tiles;
  float inv_scale;
  float* dst_ref;
  float* dst_dis;
  int64_t dst_row_pitch_r, dst_frame_pitch_r, dst_row_pitch_d, dst_frame_pitch_d;
  Taps taps;
};

constexpr int kDecTW = 64, kDecTH = 16;  // output tile

template <typename T, int N>
__global__ __launch_bounds__(kBlock) void vif_dec_kernel(const VifDecArgs a) {
  constexpr int R = N / 2, COLS = 2 * kDecTW - 2 + N, S = 8, NIN = 2 * S - 2 + N, PD = COLS + 1;
  __shared__ float sv[2][kDecTH][PD];
  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int ox0 = tx * kDecTW, oy0 = ty * kDecTH;
  const int tid = threadIdx.x;

  // vertical pass at even rows only: lane <-> input column
  for (int item = tid; item < COLS * (kDecTH / S); item += kBlock) {
    const int col = item % COLS, seg = item / COLS;
    const int gx = mirror(2 * ox0 - R + col, a.w);
    float acc[S][2];
#pragma unroll
    for (int o = 0; o < S; ++o) acc[o][0] = acc[o][1] = 0.0f;
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      const int gy = mirror(2 * (oy0 + seg * S) - R + j, a.h);
      const float r = PixIO<T>::load(ref + (int64_t)gy * a.row_pitch_r + gx, a.inv_scale);
      const float d = PixIO<T>::load(dis + (int64_t)gy * a.row_pitch_d + gx, a.inv_scale);
#pragma unroll
      for (int o = 0; o < S; ++o) {
        const int k = j - 2 * o;
        if (k >= 0 && k < N) {
          acc[o][0] = fmaf(a.taps.f[k], r, acc[o][0]);
          acc[o][1] = fmaf(a.taps.f[k], d, acc[o][1]);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < S; ++o) {
      sv[0][seg * S + o][col] = acc[o][0];
      sv[1][seg * S + o][col] = acc[o][1];
    }
  }
  __syncthreads();

  // horizontal pass at even columns: lane <-> output column, 4 rows per thread
  const int oc = tid & 63, rg = tid >> 6;
  const int gx = ox0 + oc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = rg * 4 + q, gy = oy0 + row;
    float ar = 0.0f, ad = 0.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      ar = fmaf(a.taps.f[k], sv[0][row][2 * oc + k], ar);
      ad = fmaf(a.taps.f[k], sv[1][row][2 * oc + k], ad);
    }
    if (gx < a.ow && gy < a.oh) {
      a.dst_ref[(int64_t)fr * a.dst_frame_pitch_r + (int64_t)gy * a.dst_row_pitch_r + gx] = ar;
      a.dst_dis[(int64_t)fr * a.dst_frame_pitch_d + (int64_t)gy * a.dst_row_pitch_d + gx] = ad;
    }
  }
}

template <int N, int TW>
hipError_t launch_stat_n(hipStream_t stream, Elem elem, const VifStatArgs& a, int n_frames) {
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((vif_stat_kernel<uint8_t, N, TW>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((vif_stat_kernel<uint16_t, N, TW>), grid, block, 0, stream, a); break;
    case ELEM_F32: hipLaunchKernelGGL((vif_stat_kernel<float, N, TW>), grid, block, 0, stream, a); break;
  }
  return hipGetLastError();
}

template <int N>
hipError_t launch_dec_n(hipStream_t stream, Elem elem, const VifDecArgs& a, int n_frames) {
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((vif_dec_kernel<uint8_t, N>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((vif_dec_kernel<uint16_t, N>), grid, block, 0, stream, a); break;
    case ELEM_F32: hipLaunchKernelGGL((vif_dec_kernel<float, N>), grid, block, 0, stream, a); break;
  }
  return hipGetLastError();
}

constexpr int kVifN[4] = {17, 9, 5, 3};
constexpr int kVifTW[4] = {112, 120, 120, 120};

}  // namespace

int vif_tile_w(int scale) { return kVifTW[scale]; }

hipError_t launch_vif_stat(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                           int w, int h, float inv_scale, float gain_limit, double* partials) {
  if (n_frames <= 0) return hipSuccess;
  VifStatArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.tiles_x = vif_tiles_x(scale, w);
  a.n_tiles = a.tiles_x * vif_tiles_y(h);
  a.inv_scale = inv_scale; a.gain_limit = gain_limit;
  a.partials = partials;
  a.taps = gaussian_taps(kVifN[scale]);
  switch (scale) {
    case 0: return launch_stat_n<17, 112>(stream, elem, a, n_frames);
    case 1: return launch_stat_n<9, 120>(stream, elem, a, n_frames);
    case 2: return launch_stat_n<5, 120>(stream, elem, a, n_frames);
    case 3: return launch_stat_n<3, 120>(stream, elem, a, n_frames);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_vif_decimate(hipStream_t stream, int dst_scale, Elem elem, PlaneRun ref, PlaneRun dis,
                               int n_frames, int w, int h, float inv_scale, MutPlaneRun dst_ref,
                               MutPlaneRun dst_dis) {
  if (n_frames <= 0) return hipSuccess;
  VifDecArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h; a.ow = w / 2; a.oh = h / 2;
  if (a.ow <= 0 || a.oh <= 0) return hipSuccess;
  a.tiles_x = (a.ow + kDecTW - 1) / kDecTW;
  a.n_tiles = a.tiles_x * ((a.oh + kDecTH - 1) / kDecTH);
  a.inv_scale = inv_scale;
  a.dst_ref = (float*)dst_ref.base; a.dst_dis = (float*)dst_dis.base;
  a.dst_row_pitch_r = dst_ref.row_pitch; a.dst_frame_pitch_r = dst_ref.frame_pitch;
  a.dst_row_pitch_d = dst_dis.row_pitch; a.dst_frame_pitch_d = dst_dis.frame_pitch;
  a.taps = gaussian_taps(kVifN[dst_scale]);
  switch (dst_scale) {
    case 1: return launch_dec_n<9>(stream, elem, a, n_frames);
    case 2: return launch_dec_n<5>(stream, elem, a, n_frames);
    case 3: return launch_dec_n<3>(stream, elem, a, n_frames);
  }
  return hipErrorInvalidValue;
}

}  // namespace pqa
