// Fixed-point motion for gfx950: the arithmetic of libvmaf's integer_motion.c (the extractor behind
// `integer_motion` / `integer_motion2` of the default models, app/vmaf_analyzer.py:377), restated in
// oracle/vmaf_int_oracle.c and matched bit for bit.
//   blur:  Q16 taps {3571, 16004, 26386, 16004, 3571}; vertical (sum + 2^(bpc-1)) >> bpc -> u16 (Q8 of the 8-bit
//          scale), horizontal (sum + 32768) >> 16 -> u16; borders: index -i -> i, n-1+i -> n-i (edge_16)
//   score: SAD of the blurred planes of frames i and i-1 (u64), then (float)(sad / 256.) / (w * h)
// The rounding after each pass makes the blur non-linear, so -- unlike the f32 kernel -- both frames are blurred;
// nothing is written to HBM (libvmaf keeps a ring of blurred planes).  Same v_dot2_u32_u16 scheme as
// vif_fixed.hip: samples packed in vertically / horizontally adjacent pairs, even and odd window starts served
// by two tap-pair tables on the same packed registers.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct MfxArgs {
  const void* ref;
  int64_t row_pitch, frame_pitch;
  const void* prev0;
  int64_t prev0_row_pitch;
  int w, h, tiles_x, n_tiles;
  int shift;
  uint32_t add;
  unsigned long long* partials;
  uint32_t ev[3], od[3];
};

constexpr int TW = kMotionTileW, TH = kMotionTileH, R = 2, COLS = TW + 4, NG = TW / 4, NIN = TH + 4, PITCH = 264;
static_assert(COLS <= kBlock && NIN % 2 == 0 && COLS + 8 <= PITCH, "tile geometry");

__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), acc, false);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void motion_fixed_kernel(const MfxArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t P[2][TH][PITCH];  // vertically blurred rows: current, previous
  __shared__ unsigned long long red[4];
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
    if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = 0ull;
    return;
  }
  const unsigned pitch_c = (unsigned)a.row_pitch;
  const rsrc_t rsrc_c = make_rsrc(cur, (unsigned)a.h * pitch_c * (unsigned)sizeof(T));
  const rsrc_t rsrc_p = make_rsrc(prev, (unsigned)a.h * pitch_p * (unsigned)sizeof(T));
  const int x0 = tx * TW, y0 = ty * TH;
  if (tid < COLS) {
    const int col = tid;
    const unsigned gx = (unsigned)mirror1(x0 - R + col, a.w);
    uint32_t pc[NIN / 2], pp[NIN / 2];
#pragma unroll
    for (int m = 0; m < NIN / 2; ++m) {
      const unsigned gy0 = (unsigned)mirror1(y0 - R + 2 * m, a.h), gy1 = (unsigned)mirror1(y0 - R + 2 * m + 1, a.h);
      pc[m] = (uint32_t)buf_load<T>(rsrc_c, gx, gy0 * pitch_c) | ((uint32_t)buf_load<T>(rsrc_c, gx, gy1 * pitch_c) << 16);
      pp[m] = (uint32_t)buf_load<T>(rsrc_p, gx, gy0 * pitch_p) | ((uint32_t)buf_load<T>(rsrc_p, gx, gy1 * pitch_p) << 16);
    }
#pragma unroll
    for (int o = 0; o < TH; ++o) {
      const int base = o >> 1;
      const uint32_t* tp = (o & 1) ? a.od : a.ev;
      uint32_t c = a.add, p = a.add;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c = dot2(pc[base + i], tp[i], c);
        p = dot2(pp[base + i], tp[i], p);
      }
      P[0][o][col] = (uint16_t)(c >> a.shift);
      P[1][o][col] = (uint16_t)(p >> a.shift);
    }
  }
  __syncthreads();
  uint32_t sad = 0;
#pragma unroll 1
  for (int item = tid; item < TH * NG; item += kBlock) {
    const int row = item / NG, g = item - row * NG;
    uint32_t out[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const uint2* q = reinterpret_cast<const uint2*>(&P[pl][row][4 * g]);
      const uint2 v0 = q[0], v1 = q[1];
      const uint32_t dw[4] = {v0.x, v0.y, v1.x, v1.y};
      uint32_t s0 = 32768u, s1 = 32768u, s2 = 32768u, s3 = 32768u;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        s0 = dot2(dw[i], a.ev[i], s0);
        s1 = dot2(dw[i], a.od[i], s1);
        s2 = dot2(dw[i + 1], a.ev[i], s2);
        s3 = dot2(dw[i + 1], a.od[i], s3);
      }
      out[pl][0] = s0 >> 16; out[pl][1] = s1 >> 16; out[pl][2] = s2 >> 16; out[pl][3] = s3 >> 16;
    }
    const int gy = y0 + row;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int d = (int)out[0][t] - (int)out[1][t];
      if (x0 + 4 * g + t < a.w && gy < a.h) sad += (uint32_t)(d < 0 ? -d : d);
    }
  }
  unsigned long long v = sad;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

hipError_t launch_motion_fixed(hipStream_t stream, int bit_depth, Elem elem, PlaneRun ref, const void* prev0,
                               int64_t prev0_row_pitch, int n_frames, int w, int h, unsigned long long* partials) {
  if (n_frames <= 0) return hipSuccess;
  MfxArgs a{};
  a.ref = ref.base; a.row_pitch = ref.row_pitch; a.frame_pitch = ref.frame_pitch;
  a.prev0 = prev0; a.prev0_row_pitch = prev0_row_pitch;
  a.w = w; a.h = h;
  a.tiles_x = (w + TW - 1) / TW;
  a.n_tiles = motion_tiles(w, h);
  a.shift = bit_depth; a.add = 1u << (bit_depth - 1);
  a.partials = partials;
  static const uint32_t f[5] = {3571, 16004, 26386, 16004, 3571};
  const auto tap = [&](int k) -> uint32_t { return (k >= 0 && k < 5) ? f[k] : 0u; };
  for (int i = 0; i < 3; ++i) {
    a.ev[i] = tap(2 * i) | (tap(2 * i + 1) << 16);
    a.od[i] = tap(2 * i - 1) | (tap(2 * i) << 16);
  }
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((motion_fixed_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((motion_fixed_kernel<uint16_t>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pqa
