// PSNR and SSIM side features, as FFmpeg's `psnr` and `ssim` filters define them -- the two extra
// passes the reference runs at app/vmaf_analyzer.py:1027-1034 and :1057-1064 (vf_psnr.c per-plane
// SSE; vf_ssim.c ssim_4x4xn_{8,16}bit / ssim_end1{,x} / ssim_plane).  Restated in oracle/vmaf_oracle.c.
//
// SSE is exact integer arithmetic end to end (u64 partials, fixed-order final sum): bit-exact.
// SSIM window scores are the same float expression on the same integer sums; they are accumulated
// in double.
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

// ---- SSE ----------------------------------------------------------------------------------
struct SseArgs {
  const void* a;
  const void* b;
  int64_t row_pitch_a, frame_pitch_a, row_pitch_b, frame_pitch_b;
  int w, h;
  unsigned long long* partials;
};

__device__ __forceinline__ unsigned sq_diff4_u8(unsigned x, unsigned y) {
  // sum over 4 packed bytes of (x_i - y_i)^2 = x.x + y.y - 2 x.y  (v_dot4_u32_u8, exact)
  const unsigned xx = __builtin_amdgcn_udot4(x, x, 0u, false);
  const unsigned yy = __builtin_amdgcn_udot4(y, y, 0u, false);
  const unsigned xy = __builtin_amdgcn_udot4(x, y, 0u, false);
  return xx + yy - 2u * xy;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void sse_kernel(const SseArgs a) {
  __shared__ unsigned long long red[4];
  const int fr = blockIdx.y;
  const T* __restrict__ pa = (const T*)a.a + (int64_t)fr * a.frame_pitch_a;
  const T* __restrict__ pb = (const T*)a.b + (int64_t)fr * a.frame_pitch_b;
  const int tid = threadIdx.x;
  constexpr int VEC = 16 / sizeof(T);
  const bool aligned = ((a.row_pitch_a * sizeof(T)) % 16 == 0) && ((a.row_pitch_b * sizeof(T)) % 16 == 0) &&
                       ((uintptr_t)pa % 16 == 0) && ((uintptr_t)pb % 16 == 0);
  const int wv = aligned ? (a.w / VEC) : 0;  // 16-byte vectors per row
  unsigned long long sse = 0;
  for (int y = blockIdx.x; y < a.h; y += gridDim.x) {
    const T* ra = pa + (int64_t)y * a.row_pitch_a;
    const T* rb = pb + (int64_t)y * a.row_pitch_b;
    unsigned row_acc = 0;  // <= 3840*65025 fits for 8-bit rows; hbd flushes per vector below
    for (int v = tid; v < wv; v += kBlock) {
      const uint4 x = reinterpret_cast<const uint4*>(ra)[v];
      const uint4 z = reinterpret_cast<const uint4*>(rb)[v];
      if constexpr (sizeof(T) == 1) {
        row_acc += sq_diff4_u8(x.x, z.x) + sq_diff4_u8(x.y, z.y) + sq_diff4_u8(x.z, z.z) + sq_diff4_u8(x.w, z.w);
      } else {
        const unsigned xs[4] = {x.x, x.y, x.z, x.w}, zs[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d0 = (int)(xs[i] & 0xffffu) - (int)(zs[i] & 0xffffu);
          const int d1 = (int)(xs[i] >> 16) - (int)(zs[i] >> 16);
          sse += (unsigned long long)((long long)d0 * d0) + (unsigned long long)((long long)d1 * d1);
        }
      }
    }
    for (int x = wv * VEC + tid; x < a.w; x += kBlock) {
      const long long d = (long long)ra[x] - (long long)rb[x];
      sse += (unsigned long long)(d * d);
    }
    sse += row_acc;
  }
  // wave reduce (u64 via two shuffles) then LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sse += __shfl_down(sse, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = sse;
  __syncthreads();
  if (tid == 0) a.partials[(int64_t)fr * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- SSIM ---------------------------------------------------------------------------------
struct SsimArgs {
  const void* main;
  const void* ref;
  int64_t row_pitch_m, frame_pitch_m, row_pitch_r, frame_pitch_r;
  int bw, bh;  // 4x4 blocks per row / column (w>>2, h>>2)
  int tiles_x, n_tiles;
  int max_value;
  double* partials;
  unsigned long long* sse_partials;  // nullable: per-tile SSE of the blocks this tile owns (ss - 2*s12, exact)
};

constexpr int SBW = kSsimTileBW + 1, SBH = kSsimTileBH + 1;  // block sums needed per tile

template <typename T>
__global__ __launch_bounds__(kBlock) void ssim_kernel(const SsimArgs a) {
  __shared__ unsigned sums[4][SBH][SBW + 1];
  __shared__ double red[4];
  __shared__ unsigned long long redu[4];
  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ pm = (const T*)a.main + (int64_t)fr * a.frame_pitch_m;
  const T* __restrict__ pr = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const int bx0 = tx * kSsimTileBW, by0 = ty * kSsimTileBH;
  const int tid = threadIdx.x;
  const bool aligned4 = sizeof(T) == 1 && (a.row_pitch_m % 4 == 0) && (a.row_pitch_r % 4 == 0) &&
                        ((uintptr_t)pm % 4 == 0) && ((uintptr_t)pr % 4 == 0);

  // A tile needs one block row / column more than it has windows; for the squared error each 4x4 block is
  // owned by exactly one tile (the extra row / column belongs to the neighbour, except at the plane's end).
  const bool last_x = tx == a.tiles_x - 1, last_y = (tile / a.tiles_x) == (a.n_tiles / a.tiles_x) - 1;
  unsigned long long sse = 0;
  for (int item = tid; item < SBW * SBH; item += kBlock) {
    const int ly = item / SBW, lx = item - ly * SBW;
    const int bx = bx0 + lx, by = by0 + ly;
    unsigned s1 = 0, s2 = 0, ss = 0, s12 = 0;
    if (bx < a.bw && by < a.bh) {
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const T* m = pm + (int64_t)(4 * by + y) * a.row_pitch_m + 4 * bx;
        const T* r = pr + (int64_t)(4 * by + y) * a.row_pitch_r + 4 * bx;
        if constexpr (sizeof(T) == 1) {
          unsigned x, z;
          if (aligned4) {
            x = *reinterpret_cast<const unsigned*>(m);
            z = *reinterpret_cast<const unsigned*>(r);
          } else {
            x = m[0] | (m[1] << 8) | (m[2] << 16) | ((unsigned)m[3] << 24);
            z = r[0] | (r[1] << 8) | (r[2] << 16) | ((unsigned)r[3] << 24);
          }
          s1 = __builtin_amdgcn_udot4(x, 0x01010101u, s1, false);
          s2 = __builtin_amdgcn_udot4(z, 0x01010101u, s2, false);
          ss = __builtin_amdgcn_udot4(x, x, ss, false);
          ss = __builtin_amdgcn_udot4(z, z, ss, false);
          s12 = __builtin_amdgcn_udot4(x, z, s12, false);
        } else {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const unsigned p = m[x], q = r[x];
            s1 += p; s2 += q; ss += p * p; ss += q * q; s12 += p * q;
          }
        }
      }
    }
    sums[0][ly][lx] = s1; sums[1][ly][lx] = s2; sums[2][ly][lx] = ss; sums[3][ly][lx] = s12;
    if ((lx < kSsimTileBW || last_x) && (ly < kSsimTileBH || last_y))
      sse += (unsigned long long)ss - 2ull * s12;  // sum (a-b)^2 over the block; zero for blocks outside the plane
  }
  if (a.sse_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sse += __shfl_down(sse, off, 64);
    if ((tid & 63) == 0) redu[tid >> 6] = sse;
  }
  __syncthreads();
  if (a.sse_partials && tid == 0)
    a.sse_partials[(int64_t)fr * a.n_tiles + tile] = (redu[0] + redu[1]) + (redu[2] + redu[3]);

  // one 8x8 window (2x2 blocks) per thread
  const int wx = tid & 31, wy = tid >> 5;
  double val = 0.0;
  if (bx0 + wx < a.bw - 1 && by0 + wy < a.bh - 1) {
    long long s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      s[q] = (long long)sums[q][wy][wx] + sums[q][wy][wx + 1] + sums[q][wy + 1][wx] + sums[q][wy + 1][wx + 1];
    const long long mx = a.max_value;
    const long long c1 = (long long)(.01 * .01 * mx * mx * 64 + .5);
    const long long c2 = (long long)(.03 * .03 * mx * mx * 64 * 63 + .5);
    const long long vars = s[2] * 64 - s[0] * s[0] - s[1] * s[1];
    const long long covar = s[3] * 64 - s[0] * s[1];
    const float f = (float)(2 * s[0] * s[1] + c1) * (float)(2 * covar + c2) /
                    ((float)(s[0] * s[0] + s[1] * s[1] + c1) * (float)(vars + c2));
    val = (double)f;
  }
  double v[1] = {val};
  block_sum<1>(v, red);
  if (tid == 0) a.partials[(int64_t)fr * a.n_tiles + tile] = v[0];
}

}  // namespace

hipError_t launch_sse(hipStream_t stream, Elem elem, PlaneRun pa, PlaneRun pb, int n_frames, int w, int h,
                      unsigned long long* partials) {
  if (n_frames <= 0) return hipSuccess;
  SseArgs a{};
  a.a = pa.base; a.b = pb.base;
  a.row_pitch_a = pa.row_pitch; a.frame_pitch_a = pa.frame_pitch;
  a.row_pitch_b = pb.row_pitch; a.frame_pitch_b = pb.frame_pitch;
  a.w = w; a.h = h;
  a.partials = partials;
  const dim3 grid(kSseBlocksPerPlane, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((sse_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((sse_kernel<uint16_t>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_ssim(hipStream_t stream, Elem elem, PlaneRun pm, PlaneRun pr, int n_frames, int w, int h,
                       int max_value, double* partials, unsigned long long* sse_partials) {
  if (n_frames <= 0) return hipSuccess;
  SsimArgs a{};
  a.main = pm.base; a.ref = pr.base;
  a.row_pitch_m = pm.row_pitch; a.frame_pitch_m = pm.frame_pitch;
  a.row_pitch_r = pr.row_pitch; a.frame_pitch_r = pr.frame_pitch;
  a.bw = w >> 2; a.bh = h >> 2;
  a.n_tiles = ssim_tiles(w, h);
  if (a.n_tiles == 0) return hipSuccess;
  a.tiles_x = ((a.bw - 1) + kSsimTileBW - 1) / kSsimTileBW;
  a.max_value = max_value;
  a.partials = partials;
  a.sse_partials = sse_partials;
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((ssim_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((ssim_kernel<uint16_t>), grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pqa
