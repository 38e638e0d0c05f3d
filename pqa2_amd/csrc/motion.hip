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
  double* partials;
  f2 vt[6];  // vertical tap pairs {c[k], c[k-1]}, c[-1] = c[5] = 0
  f2 ht[5];  // horizontal splats {c[k], c[k]}
};

constexpr int TW = kMotionTileW, TH = kMotionTileH, R = 2, COLS = TW + 4, NSEG = TW / 4, S = 16, NIN = S + 4;
constexpr int kP2 = 258;  // float2 pitch == 2 (mod 32): conflict-free ds_read_b128
static_assert(COLS <= 256 && NSEG <= 64 && TW % 4 == 0 && TH == S, "one column per thread, one 16-row strip");

// Same packed layout as the VIF kernel: float2 = {row 2p, row 2p+1} of one column; the vertical pass
// uses tap pairs with the input broadcast, the horizontal pass broadcast taps with input pairs.
// Works on the raw integer frame difference; the 2^-(bpc-8) sample scale is applied to the final sum
// (exact: a power of two commutes with every rounding on the way).
template <typename T>
__global__ __launch_bounds__(kBlock) void motion_kernel(const MotionArgs a) {
  __shared__ f2 sv[TH / 2][kP2];
  __shared__ double red[4];
  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const int tid = threadIdx.x;
  const T* __restrict__ cur = (const T*)a.ref + (int64_t)fr * a.frame_pitch;
  const T* __restrict__ prev;
  unsigned pitch_p;
  if (fr == 0) {
    prev = (const T*)a.prev0;
    pitch_p = (unsigned)a.prev0_row_pitch;
  } else {
    prev = cur - a.frame_pitch;
    pitch_p = (unsigned)a.row_pitch;
  }
  if (prev == nullptr) {  // first frame of the clip: motion_0 = 0
    if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = 0.0;
    return;
  }
  const unsigned pitch_c = (unsigned)a.row_pitch;
  const rsrc_t rsrc_c = make_rsrc(cur, (unsigned)a.h * pitch_c * (unsigned)sizeof(T));
  const rsrc_t rsrc_p = make_rsrc(prev, (unsigned)a.h * pitch_p * (unsigned)sizeof(T));
  const int x0 = tx * TW, y0 = ty * TH;
  {
    const int col = tid;  // one column per thread; the 16-row strip is the whole tile: rows are wave-uniform
    if (col < COLS) {
      const unsigned gx = (unsigned)mirror1(x0 - R + col, a.w);
      f2 acc[S / 2];
#pragma unroll
      for (int p = 0; p < S / 2; ++p) acc[p] = f2{0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const unsigned gy = (unsigned)mirror1(y0 - R + j, a.h);
        const int c = (int)buf_load<T>(rsrc_c, gx, gy * pitch_c);
        const int p0 = (int)buf_load<T>(rsrc_p, gx, gy * pitch_p);
        const float d = (float)(c - p0);
#pragma unroll
        for (int p = 0; p < S / 2; ++p) {
          const int k = j - 2 * p;
          if (k >= 0 && k <= 5) acc[p] = __builtin_elementwise_fma(a.vt[k], f2{d, d}, acc[p]);
        }
      }
#pragma unroll
      for (int p = 0; p < S / 2; ++p) sv[p][col] = acc[p];
    }
  }
  __syncthreads();
  // lane l: row pair l & 7, segment wave + 4 * (l >> 3) + 32 * round: the 16-byte chunk index of a lane is
  // (rp + 8 * (j & 1) + const) mod 16, distinct inside every ds_read_b128 lane group
  const int wave = tid >> 6, lane = tid & 63;
  const int rp = lane & 7;
  float sad = 0.0f;
  const int gyA = y0 + 2 * rp;
  const float mA = gyA < a.h ? 1.0f : 0.0f, mB = gyA + 1 < a.h ? 1.0f : 0.0f;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int seg = wave + 4 * (lane >> 3) + 32 * round;
    if (seg < NSEG) {
      f2 in[8];
      const f4* p = reinterpret_cast<const f4*>(&sv[rp][seg * 4]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 v = p[q];
        in[2 * q] = f2{v.x, v.y};
        in[2 * q + 1] = f2{v.z, v.w};
      }
      f2 out[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) out[o] = f2{0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int o = 0; o < 4; ++o) out[o] = __builtin_elementwise_fma(a.ht[k], in[o + k], out[o]);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float mc = (x0 + seg * 4 + o) < a.w ? 1.0f : 0.0f;
        sad = fmaf(mc * mA, fabsf(out[o].x), sad);
        sad = fmaf(mc * mB, fabsf(out[o].y), sad);
      }
    }
  }
  const float part[1] = {sad};
  double v[1];
  block_sum_f32<1>(part, v, red);
  if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = v[0];
}

}  // namespace

hipError_t launch_motion(hipStream_t stream, Elem elem, PlaneRun ref, const void* prev0, int64_t prev0_row_pitch,
                         int n_frames, int w, int h, float inv_scale, double* partials, int mode, int* n_partials) {
  if (n_partials) *n_partials = motion_tiles(w, h);
  if (n_frames <= 0) return hipSuccess;
  if (mode == MOTION_AUTO) {
    hipError_t err = hipSuccess;
    if (launch_motion_march(stream, elem, ref, prev0, prev0_row_pitch, n_frames, w, h, partials, n_partials, &err)) return err;
  }
  MotionArgs a{};
  a.ref = ref.base; a.row_pitch = ref.row_pitch; a.frame_pitch = ref.frame_pitch;
  a.prev0 = prev0; a.prev0_row_pitch = prev0_row_pitch;
  a.w = w; a.h = h;
  a.tiles_x = (w + TW - 1) / TW;
  a.n_tiles = motion_tiles(w, h);
  (void)inv_scale;  // applied by the finalize stage (motion_norm)
  a.partials = partials;
  {  // FILTER_5_s: 5 taps, sigma 1.0
    double v[5], sum = 0.0;
    float f[5];
    for (int k = 0; k < 5; ++k) { v[k] = exp(-0.5 * (k - 2) * (k - 2)); sum += v[k]; }
    for (int k = 0; k < 5; ++k) f[k] = (float)(v[k] / sum);
    for (int k = 0; k <= 5; ++k) a.vt[k] = f2{k < 5 ? f[k] : 0.0f, k > 0 ? f[k - 1] : 0.0f};
    for (int k = 0; k < 5; ++k) a.ht[k] = f2{f[k], f[k]};
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
