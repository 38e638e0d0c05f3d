// Device-side helpers shared by the gfx950 feature kernels (wave64, LDS-staged stencils).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqa {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;  // 4 waves per workgroup everywhere

// libvmaf border rule (vif_tools.c / adm_tools.c / convolution_internal.h): index -i -> i, index n-1+i -> n-i
// (the high edge repeats the edge sample).  The oracle keeps the general folding loop; the kernels use mirror1.

// Buffer addressing: a 128-bit resource (base + size, built from wave-uniform values) with the column
// offset in a VGPR and the row offset in an SGPR.  One plane row costs zero VALU address arithmetic and
// out-of-range offsets read 0 instead of faulting.  Offsets are in ELEMENTS here.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <typename T> __device__ __forceinline__ T buf_load(rsrc_t r, unsigned lane_off, unsigned row_off);
template <> __device__ __forceinline__ uint8_t buf_load<uint8_t>(rsrc_t r, unsigned lane_off, unsigned row_off) {
  return __builtin_amdgcn_raw_buffer_load_b8(r, lane_off, row_off, 0);
}
template <> __device__ __forceinline__ uint16_t buf_load<uint16_t>(rsrc_t r, unsigned lane_off, unsigned row_off) {
  return __builtin_amdgcn_raw_buffer_load_b16(r, lane_off * 2u, row_off * 2u, 0);
}
template <> __device__ __forceinline__ float buf_load<float>(rsrc_t r, unsigned lane_off, unsigned row_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off * 4u, row_off * 4u, 0));
}

// The border rule with a single fold and a clamp.  Exact whenever n > the stencil radius (every plane the
// library accepts: w, h >= 16 at scale 0 and >= 2 at the deepest scale); indices further out belong to
// tile positions beyond the image whose results are masked, they only need to stay in bounds.
__device__ __forceinline__ int mirror1(int i, int n) {
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - i - 1 : i;
  return min(max(i, 0), n - 1);
}
// Same with the high-edge fold point as a parameter: fold = 2n - 1 is mirror1 (vif_tools.c), fold = 2n - 2
// is reflect-101 on both edges, the padding integer_vif.c applies (pad_top_and_bottom / PADDING_SQ_DATA).
__device__ __forceinline__ int mirror_fold(int i, int n, int fold) {
  i = i < 0 ? -i : i;
  i = i >= n ? fold - i : i;
  return min(max(i, 0), n - 1);
}

// {ref, dis} samples -> float2 as libvmaf picture_copy does: v * inv_scale - 128 (inv_scale = 2^-(bpc-8));
// f32 planes (pyramid levels) pass through.
template <typename T> struct PixIO;
template <> struct PixIO<uint8_t> {
  static __device__ __forceinline__ f2 pair(uint8_t r, uint8_t d, float) {
    return f2{(float)r, (float)d} + f2{-128.0f, -128.0f};
  }
};
template <> struct PixIO<uint16_t> {
  static __device__ __forceinline__ f2 pair(uint16_t r, uint16_t d, float inv_scale) {
    return f2{(float)r, (float)d} * f2{inv_scale, inv_scale} + f2{-128.0f, -128.0f};
  }
};
template <> struct PixIO<float> {
  static __device__ __forceinline__ f2 pair(float r, float d, float) { return f2{r, d}; }
};

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give each XCD a
// contiguous run of tile ids so tiles that share halo rows hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Deterministic block sum of NV doubles per thread (256 threads).  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) lds[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
  }
}

// Wave64 sum of one float per lane with DPP adds only (no LDS traffic, 6 VALU): quad swaps, row shifts,
// then row broadcasts; the total lands in lane 63 and is returned wave-uniform.  Fixed order: deterministic.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v = dpp_add<0xb1>(v);         // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e>(v);         // quad_perm:[2,3,0,1]
  v = dpp_add<0x114>(v);        // row_shr:4
  v = dpp_add<0x118>(v);        // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Block sum of NV floats per thread: DPP inside each wave (f32), the four wave totals added in double
// by thread 0.  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum_f32(const float (&in)[NV], double (&out)[NV], double* lds /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float w = wave_sum_f32(in[i]);
    if (lane == 0) lds[wid * NV + i] = (double)w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
  }
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }  // v_log_f32 (base 2)

}  // namespace pqa
