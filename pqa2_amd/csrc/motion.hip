// motion: mean absolute difference of the 5-tap-blurred reference luma of consecutive frames.
//
// Arithmetic follows libvmaf float_motion.c (extract), motion.c (compute_motion / vmaf_image_sad_c)
// and convolution.c (convolution_f32_c_s with FILTER_5_s) -- reached through the reference's
// `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated in oracle/vmaf_oracle.c.
//
// The blur is linear, so SAD(blur(cur), blur(prev)) is evaluated as sum |blur(cur - prev)|: the frame
// difference is formed exactly on the integer samples, one blur instead of two, and no blurred plane
// is ever written to or re-read from HBM (libvmaf keeps a 3-deep ring of blurred planes).  The f32
// rounding of the two orders differs by ~1e-7 relative; a static clip gives exactly 0 either way.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

struct MotionArgs {
  const void* ref;
  int64_t row_pitch, frame_pitch;
  const void* prev0;
  int64_t prev0_row_pitch;
  int w, h, tiles_x, n_tiles;
  float inv_scale;
  double* partials;
  float f[5];
};

constexpr int TW = kMotionTileW, TH = kMotionTileH, R = 2, COLS = TW + 4, P = 130, NSEG = TW / 8, S = 8, NIN = S + 4;

template <typename T>
__global__ __launch_bounds__(kBlock) void motion_kernel(const MotionArgs a) {
  __shared__ float sv[TH][P];
  __shared__ double red[4];
  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const int tid = threadIdx.x;
  const T* __restrict__ cur = (const T*)a.ref + (int64_t)fr * a.frame_pitch;
  const T* __restrict__ prev;
  int64_t prev_pitch;
  if (fr == 0) {
    prev = (const T*)a.prev0;
    prev_pitch = a.prev0_row_pitch;
  } else {
    prev = cur - a.frame_pitch;
    prev_pitch = a.row_pitch;
  }
  if (prev == nullptr) {  // first frame of the clip: motion_0 = 0
    if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = 0.0;
    return;
  }
  const int x0 = tx * TW, y0 = ty * TH;
  {
    const int col = tid & 127, seg = tid >> 7;
    if (col < COLS) {
      const int gx = mirror(x0 - R + col, a.w);
      float acc[S];
#pragma unroll
      for (int o = 0; o < S; ++o) acc[o] = 0.0f;
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const int gy = mirror(y0 + seg * S - R + j, a.h);
        const float d = (PixIO<T>::raw(cur + (int64_t)gy * a.row_pitch + gx) -
                         PixIO<T>::raw(prev + (int64_t)gy * prev_pitch + gx)) * a.inv_scale;
#pragma unroll
        for (int o = 0; o < S; ++o) {
          const int k = j - o;
          if (k >= 0 && k < 5) acc[o] = fmaf(a.f[k], d, acc[o]);
        }
      }
#pragma unroll
      for (int o = 0; o < S; ++o) sv[seg * S + o][col] = acc[o];
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int row = lane & 15, seg = wave + 4 * (lane >> 4);
  float sad = 0.0f;
  if (seg < NSEG) {
    float in[12];
    const float2* p = reinterpret_cast<const float2*>(&sv[row][seg * 8]);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float2 v = p[q];
      in[2 * q] = v.x;
      in[2 * q + 1] = v.y;
    }
    const int gy = y0 + row;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) acc = fmaf(a.f[k], in[o + k], acc);
      if (x0 + seg * 8 + o < a.w && gy < a.h) sad += fabsf(acc);
    }
  }
  double v[1] = {(double)sad};
  block_sum<1>(v, red);
  if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = v[0];
}

}  // namespace

hipError_t launch_motion(hipStream_t stream, Elem elem, PlaneRun ref, const void* prev0, int64_t prev0_row_pitch,
                         int n_frames, int w, int h, float inv_scale, double* partials) {
  if (n_frames <= 0) return hipSuccess;
  MotionArgs a{};
  a.ref = ref.base; a.row_pitch = ref.row_pitch; a.frame_pitch = ref.frame_pitch;
  a.prev0 = prev0; a.prev0_row_pitch = prev0_row_pitch;
  a.w = w; a.h = h;
  a.tiles_x = (w + TW - 1) / TW;
  a.n_tiles = motion_tiles(w, h);
  a.inv_scale = inv_scale;
  a.partials = partials;
  {  // FILTER_5_s: 5 taps, sigma 1.0
    double v[5], sum = 0.0;
    for (int k = 0; k < 5; ++k) { v[k] = exp(-0.5 * (k - 2) * (k - 2)); sum += v[k]; }
    for (int k = 0; k < 5; ++k) a.f[k] = (float)(v[k] / sum);
  }
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((motion_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((motion_kernel<uint16_t>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pqa
