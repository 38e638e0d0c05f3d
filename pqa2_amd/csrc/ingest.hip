// Decoder surfaces -> the planes the feature kernels read.
//
// The reference never sees decoded frames: its ffmpeg child decodes and scores in one process
// (app/vmaf_analyzer.py:406-419).  A drop-in that keeps decode on the GPU (VCN through rocDecode,
// which this image does not have) is handed what a hardware decoder writes: NV12 (8 bit: a Y plane and ONE plane of
// interleaved U,V pairs at half resolution) or P010 / P012 (the same with 16-bit little-endian samples whose VALUE sits in
// the UPPER bits, the low 16 - bit_depth bits zero).  pqa_submit_surfaces (pqa_api.hip) consumes such surfaces in place:
//   * an NV12 luma plane is read by the feature kernels where it lies (pitched plane: nothing to do);
//   * interleaved chroma is split into U and V planes (only when the context scores chroma: PSNR / SSIM on all planes);
//   * 16-bit samples are shifted down to the LSB-aligned form the kernels compute on.
// Pure byte work, HBM-bound: 16-byte loads and stores, one thread per 8 output samples, rows and frames in the grid.
#include "ingest.h"
#include "pqa_device.h"

namespace pqa {
namespace {

struct IngestArgs {
  const uint8_t* src;
  int64_t src_row_pitch, src_frame_pitch;   // bytes
  uint8_t* dst0;                            // luma (SHIFT16) or U (DEINT*)
  uint8_t* dst1;                            // V (DEINT*)
  int64_t dst_row_pitch, dst_frame_pitch;   // bytes
  int w, h;                                 // output samples per row / rows (of ONE output plane)
  int shift;                                // right shift of 16-bit samples
};

enum { SHIFT16 = 0, DEINT8 = 1, DEINT16 = 2 };

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// two 16-bit samples per dword, both shifted right (v_pk_lshrrev_b16)
__device__ __forceinline__ unsigned shr16x2(unsigned x, int s) {
  const us2 v = __builtin_bit_cast(us2, x) >> us2{(unsigned short)s, (unsigned short)s};
  return __builtin_bit_cast(unsigned, v);
}

// VEC: every source row starts 16-byte aligned (base and pitches multiples of 16) and so does every destination row (the
// library's own slot layout: 64-byte row pitch).  Only chunks that lie completely inside the row take the 16-byte path.
template <int MODE, bool VEC>
__global__ __launch_bounds__(kBlock) void ingest_kernel(const IngestArgs a) {
  const int chunk = blockIdx.x * kBlock + threadIdx.x;   // 8 output samples per chunk
  const int x = chunk * 8;
  if (x >= a.w) return;
  const int y = blockIdx.y, fr = blockIdx.z;
  const uint8_t* __restrict__ s = a.src + (int64_t)fr * a.src_frame_pitch + (int64_t)y * a.src_row_pitch;
  uint8_t* __restrict__ d0 = a.dst0 + (int64_t)fr * a.dst_frame_pitch + (int64_t)y * a.dst_row_pitch;
  uint8_t* __restrict__ d1 = MODE == SHIFT16 ? nullptr : a.dst1 + (int64_t)fr * a.dst_frame_pitch + (int64_t)y * a.dst_row_pitch;
  const bool full = x + 8 <= a.w;
  if (VEC && full) {
    if (MODE == SHIFT16) {
      u4 v = *reinterpret_cast<const u4*>(s + (int64_t)x * 2);
      v = u4{shr16x2(v.x, a.shift), shr16x2(v.y, a.shift), shr16x2(v.z, a.shift), shr16x2(v.w, a.shift)};
      *reinterpret_cast<u4*>(d0 + (int64_t)x * 2) = v;
    } else if (MODE == DEINT8) {
      const u4 v = *reinterpret_cast<const u4*>(s + (int64_t)x * 2);   // U0 V0 U1 V1 ... U7 V7
      // bytes 0,2 of each dword are U, bytes 1,3 are V: v_perm_b32 gathers four of a kind from two dwords
      const u2 u = u2{__builtin_amdgcn_perm(v.y, v.x, 0x06040200u), __builtin_amdgcn_perm(v.w, v.z, 0x06040200u)};
      const u2 vv = u2{__builtin_amdgcn_perm(v.y, v.x, 0x07050301u), __builtin_amdgcn_perm(v.w, v.z, 0x07050301u)};
      *reinterpret_cast<u2*>(d0 + x) = u;
      *reinterpret_cast<u2*>(d1 + x) = vv;
    } else {
      const u4 p = *reinterpret_cast<const u4*>(s + (int64_t)x * 4);        // U0 V0 U1 V1 U2 V2 U3 V3 (16 bit each)
      const u4 q = *reinterpret_cast<const u4*>(s + (int64_t)x * 4 + 16);   // U4 ... V7
      // low halves are U, high halves V
      const auto lo = [](unsigned a0, unsigned a1) { return __builtin_amdgcn_perm(a1, a0, 0x05040100u); };
      const auto hi = [](unsigned a0, unsigned a1) { return __builtin_amdgcn_perm(a1, a0, 0x07060302u); };
      const u4 u = u4{shr16x2(lo(p.x, p.y), a.shift), shr16x2(lo(p.z, p.w), a.shift), shr16x2(lo(q.x, q.y), a.shift),
                      shr16x2(lo(q.z, q.w), a.shift)};
      const u4 vv = u4{shr16x2(hi(p.x, p.y), a.shift), shr16x2(hi(p.z, p.w), a.shift), shr16x2(hi(q.x, q.y), a.shift),
                       shr16x2(hi(q.z, q.w), a.shift)};
      *reinterpret_cast<u4*>(d0 + (int64_t)x * 2) = u;
      *reinterpret_cast<u4*>(d1 + (int64_t)x * 2) = vv;
    }
    return;
  }
  // row tails, and every chunk of a surface whose rows are not 16-byte aligned: sample by sample
  const int n = full ? 8 : a.w - x;
  for (int i = 0; i < n; ++i) {
    if (MODE == SHIFT16) {
      const unsigned short v = *reinterpret_cast<const unsigned short*>(s + (int64_t)(x + i) * 2);
      *reinterpret_cast<unsigned short*>(d0 + (int64_t)(x + i) * 2) = (unsigned short)(v >> a.shift);
    } else if (MODE == DEINT8) {
      d0[x + i] = s[(int64_t)(x + i) * 2];
      d1[x + i] = s[(int64_t)(x + i) * 2 + 1];
    } else {
      const unsigned short u = *reinterpret_cast<const unsigned short*>(s + (int64_t)(x + i) * 4);
      const unsigned short v = *reinterpret_cast<const unsigned short*>(s + (int64_t)(x + i) * 4 + 2);
      *reinterpret_cast<unsigned short*>(d0 + (int64_t)(x + i) * 2) = (unsigned short)(u >> a.shift);
      *reinterpret_cast<unsigned short*>(d1 + (int64_t)(x + i) * 2) = (unsigned short)(v >> a.shift);
    }
  }
}

template <int MODE>
hipError_t launch(hipStream_t stream, const IngestArgs& a, int n_frames) {
  if (n_frames <= 0 || a.w <= 0 || a.h <= 0) return hipSuccess;
  if (a.h > 65535 || n_frames > 65535) return hipErrorInvalidValue;
  const bool vec = (((uintptr_t)a.src | (uintptr_t)a.src_row_pitch | (uintptr_t)a.src_frame_pitch) & 15) == 0 &&
                   (((uintptr_t)a.dst0 | (uintptr_t)a.dst1 | (uintptr_t)a.dst_row_pitch | (uintptr_t)a.dst_frame_pitch) & 15) == 0;
  const dim3 grid(((a.w + 7) / 8 + kBlock - 1) / kBlock, a.h, n_frames), block(kBlock);
  if (vec) hipLaunchKernelGGL((ingest_kernel<MODE, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((ingest_kernel<MODE, false>), grid, block, 0, stream, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_ingest_shift16(hipStream_t stream, const void* src, int64_t src_row_pitch, int64_t src_frame_pitch,
                                 void* dst, int64_t dst_row_pitch, int64_t dst_frame_pitch, int w, int h, int shift,
                                 int n_frames) {
  IngestArgs a{};
  a.src = (const uint8_t*)src; a.src_row_pitch = src_row_pitch; a.src_frame_pitch = src_frame_pitch;
  a.dst0 = (uint8_t*)dst; a.dst1 = nullptr; a.dst_row_pitch = dst_row_pitch; a.dst_frame_pitch = dst_frame_pitch;
  a.w = w; a.h = h; a.shift = shift;
  return launch<SHIFT16>(stream, a, n_frames);
}

hipError_t launch_ingest_deinterleave(hipStream_t stream, int esize, const void* src_uv, int64_t src_row_pitch,
                                      int64_t src_frame_pitch, void* dst_u, void* dst_v, int64_t dst_row_pitch,
                                      int64_t dst_frame_pitch, int w, int h, int shift, int n_frames) {
  IngestArgs a{};
  a.src = (const uint8_t*)src_uv; a.src_row_pitch = src_row_pitch; a.src_frame_pitch = src_frame_pitch;
  a.dst0 = (uint8_t*)dst_u; a.dst1 = (uint8_t*)dst_v; a.dst_row_pitch = dst_row_pitch; a.dst_frame_pitch = dst_frame_pitch;
  a.w = w; a.h = h; a.shift = shift;
  return esize == 1 ? launch<DEINT8>(stream, a, n_frames) : launch<DEINT16>(stream, a, n_frames);
}

}  // namespace pqa
