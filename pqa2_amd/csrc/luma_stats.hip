// Per-frame luma statistics for white "bookend" frame detection -- the step right before the scoring
// path in the reference: BookendAligner samples frames and takes np.mean(gray), np.std(gray) and the
// fraction of pixels above a threshold (app/bookend_alignment.py:796-800, 902-904, 998-1020;
// app/reference_analyzer.py:127-144).  Here the three reductions are exact integers per frame
// (sum, sum of squares, count above threshold); mean / std / ratio follow on the host in float64.
// Pure streaming: one 16-byte load per lane per step, v_dot4_u32_u8 for both sums -> HBM-bound.
//
// Gray convention (pqa_set_luma_gray).  The reference's `gray` is cv2.cvtColor(BGR2GRAY) of a frame that cv2.VideoCapture
// has already converted from limited-range YUV to full-range BGR: with BT.601 on both legs the chroma terms cancel and
// gray = clamp(round((Y - 16) * 255 / 219), 0, 255), an 8-bit number whatever the clip's bit depth.  MAP = true applies
// exactly that map per sample before the three reductions (one v_cvt_f32_ubyte + v_fma_f32 + v_cvt_pk_u8_f32 per sample:
// the convert rounds to nearest and saturates to 0..255, the small bias makes the exact .5 cases of 10/12-bit samples
// round up like floor(x + 0.5)); the statistics are then those of the mapped 8-bit gray and the threshold is in its units.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

struct LumaArgs {
  const void* base;
  int64_t row_pitch, frame_pitch;
  int w, h;
  unsigned threshold;
  float map_a, map_b;            // MAP: gray = sat_u8(rne(sample * map_a + map_b))
  unsigned long long* partials;  // [n_frames][kLumaBlocks][3]
};

__device__ __forceinline__ unsigned gray_u8(float sample, float a, float b) {   // one mapped sample in bits 0..7
  return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(sample, a, b), 0u, 0u);
}
// four u8 samples of a dword -> their four gray bytes, same positions
__device__ __forceinline__ unsigned gray4_u8(unsigned x, float a, float b) {
  unsigned g = 0;
  g = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)(x & 0xffu), a, b), 0u, g);
  g = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)((x >> 8) & 0xffu), a, b), 1u, g);
  g = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)((x >> 16) & 0xffu), a, b), 2u, g);
  g = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)(x >> 24), a, b), 3u, g);
  return g;
}

__device__ __forceinline__ unsigned count_gt4(unsigned x, unsigned thr) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) c += ((x >> (8 * i)) & 0xffu) > thr ? 1u : 0u;
  return c;
}

template <typename T, bool MAP>
__global__ __launch_bounds__(kBlock) void luma_stats_kernel(const LumaArgs a) {
  __shared__ unsigned long long red[12];
  const int fr = blockIdx.y;
  const T* __restrict__ p = (const T*)a.base + (int64_t)fr * a.frame_pitch;
  const int tid = threadIdx.x;
  constexpr int VEC = 16 / sizeof(T);
  const bool aligned = ((a.row_pitch * sizeof(T)) % 16 == 0) && ((uintptr_t)p % 16 == 0);
  unsigned long long sum = 0, sq = 0, cnt = 0;
  const auto eat = [&](const uint4 x, unsigned& rs, unsigned& rq, unsigned& rc) {
    const unsigned xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (sizeof(T) == 1) {
        const unsigned g = MAP ? gray4_u8(xs[i], a.map_a, a.map_b) : xs[i];
        rs = __builtin_amdgcn_udot4(g, 0x01010101u, rs, false);
        rq = __builtin_amdgcn_udot4(g, g, rq, false);
        rc += count_gt4(g, a.threshold);
      } else {
        unsigned lo = xs[i] & 0xffffu, hi = xs[i] >> 16;
        if constexpr (MAP) {
          lo = gray_u8((float)lo, a.map_a, a.map_b);
          hi = gray_u8((float)hi, a.map_a, a.map_b);
        }
        sum += lo + hi;
        sq += (unsigned long long)lo * lo + (unsigned long long)hi * hi;
        rc += (lo > a.threshold ? 1u : 0u) + (hi > a.threshold ? 1u : 0u);
      }
    }
  };
  if ((uintptr_t)p % 16 == 0 && a.row_pitch == a.w && ((int64_t)a.w * a.h * (int64_t)sizeof(T)) % 16 == 0) {
    // contiguous plane (no row padding): one flat grid-stride sweep, every lane busy whatever the row length
    // (a 1080p row is 120 16-byte vectors: the row-by-row form below keeps 136 of 256 lanes idle there)
    const int64_t nv = (int64_t)a.w * a.h * (int64_t)sizeof(T) / 16;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    for (int64_t v0 = (int64_t)blockIdx.x * kBlock * 8; v0 < nv; v0 += (int64_t)gridDim.x * kBlock * 8) {
      unsigned rs = 0, rq = 0, rc = 0;   // 8 vectors of 16 bytes per lane: 32-bit partial sums cannot overflow
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t v = v0 + (int64_t)u * kBlock + tid;
        if (v < nv) eat(q[v], rs, rq, rc);
      }
      sum += rs; sq += rq; cnt += rc;
    }
  } else {
    const int wv = aligned ? a.w / VEC : 0;
    for (int y = blockIdx.x; y < a.h; y += gridDim.x) {
      const T* row = p + (int64_t)y * a.row_pitch;
      unsigned rs = 0, rq = 0, rc = 0;  // per-row 32-bit accumulators (8-bit rows cannot overflow them)
      for (int v = tid; v < wv; v += kBlock) eat(reinterpret_cast<const uint4*>(row)[v], rs, rq, rc);
      for (int x = wv * VEC + tid; x < a.w; x += kBlock) {
        unsigned v = row[x];
        if constexpr (MAP) v = gray_u8((float)v, a.map_a, a.map_b);
        sum += v;
        sq += (unsigned long long)v * v;
        rc += v > a.threshold ? 1u : 0u;
      }
      sum += rs; sq += rq; cnt += rc;
    }
  }
  unsigned long long v[3] = {sum, sq, cnt};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    if ((tid & 63) == 0) red[(tid >> 6) * 3 + i] = v[i];
  }
  __syncthreads();
  if (tid < 3) a.partials[((int64_t)fr * gridDim.x + blockIdx.x) * 3 + tid] = (red[tid] + red[3 + tid]) + (red[6 + tid] + red[9 + tid]);
}

__global__ __launch_bounds__(kBlock) void luma_stats_finalize(const unsigned long long* partials, int n_blocks,
                                                               unsigned long long* out) {
  __shared__ unsigned long long red[12];
  const int fr = blockIdx.x, tid = threadIdx.x;
  unsigned long long v[3] = {0, 0, 0};
  for (int b = tid; b < n_blocks; b += kBlock)
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] += partials[((int64_t)fr * n_blocks + b) * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    if ((tid & 63) == 0) red[(tid >> 6) * 3 + i] = v[i];
  }
  __syncthreads();
  if (tid < 3) out[(int64_t)fr * 3 + tid] = (red[tid] + red[3 + tid]) + (red[6 + tid] + red[9 + tid]);
}

}  // namespace

void luma_gray_map(int bit_depth, float* a, float* b) {
  // gray = floor((Y - 16 s) * 255 / (219 s) + 0.5), s = 2^(bpc - 8): as one f32 fma followed by a round-to-nearest-even
  // convert.  Exact .5 cases exist from 10 bit up (Y - 64 = 146 (2k + 1)); the nearest non-tie sits 1 / (876 s / 4) away,
  // so a bias of a quarter of that turns every tie into "up" and moves nothing else (tests/test_bookend.py walks every
  // sample value of 8, 10 and 12 bit through this arithmetic in numpy).
  const double s = (double)(1 << (bit_depth - 8));
  *a = (float)(255.0 / (219.0 * s));
  *b = (float)(-16.0 * 255.0 / 219.0 + 0.25 / (219.0 * s));
}

hipError_t launch_luma_stats(hipStream_t stream, Elem elem, PlaneRun luma, int n_frames, int w, int h,
                             unsigned threshold, int gray_bit_depth, unsigned long long* partials,
                             unsigned long long* out) {
  if (n_frames <= 0) return hipSuccess;
  LumaArgs a{};
  a.base = luma.base; a.row_pitch = luma.row_pitch; a.frame_pitch = luma.frame_pitch;
  a.w = w; a.h = h; a.threshold = threshold; a.partials = partials;
  const bool map = gray_bit_depth > 0;   // 0: statistics of the samples as they are
  if (map) luma_gray_map(gray_bit_depth, &a.map_a, &a.map_b);
  const dim3 grid(kLumaBlocks, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8:
      if (map) hipLaunchKernelGGL((luma_stats_kernel<uint8_t, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((luma_stats_kernel<uint8_t, false>), grid, block, 0, stream, a);
      break;
    case ELEM_U16:
      if (map) hipLaunchKernelGGL((luma_stats_kernel<uint16_t, true>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((luma_stats_kernel<uint16_t, false>), grid, block, 0, stream, a);
      break;
    default: return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(luma_stats_finalize, dim3(n_frames), dim3(kBlock), 0, stream, partials, kLumaBlocks, out);
  return hipGetLastError();
}

}  // namespace pqa
