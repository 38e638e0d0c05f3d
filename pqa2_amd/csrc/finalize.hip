// Second-stage reduction: every per-tile partial of a batch -> one 24-slot record per frame.
// One workgroup per (frame, quantity group) walks each partial array in a FIXED order (thread-strided, then the same
// shuffle/LDS tree), so a frame's record does not depend on how frames are split over batches, launches or ranks (by
// construction; checked with 1, 2 and 3 ranks sharing one GPU -- a run on separate GPUs has not been possible yet).
// Record layout: include/pqa_vmaf.h (PQA_REC_*).
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

__device__ double reduce_strided(const double* p, int n, int stride, double* red) {
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < n; i += kBlock) acc[0] += p[(int64_t)i * stride];
  __syncthreads();  // red reuse
  block_sum<1>(acc, red);
  return acc[0];  // valid in thread 0
}

__global__ __launch_bounds__(kBlock) void finalize_kernel(const FinalizeArgs a) {
  __shared__ double red[4];
  __shared__ unsigned long long redu[4];
  const int fr = blockIdx.x;
  const int grp = blockIdx.y;  // 0..3 vif scale, 4..7 adm scale, 8 motion + ssim + sse
  const int row = (int)(((int64_t)a.slot_base + (int64_t)fr * a.slot_step) % a.capacity);
  double* rec = a.records + (int64_t)row * a.record_stride;
  const int tid = threadIdx.x;

  if (a.has_vif && grp < 4 && a.vif_fx_part[grp]) {
    // integer_vif.c epilogue: exact integer sums over the tiles, then the two doubles exactly as the C code
    // forms them (operation by operation, no fused multiply-add)
#pragma clang fp contract(off)
    const int s = grp;
    const long long* p = a.vif_fx_part[s] + (int64_t)fr * a.vif_tiles[s] * kVifFxPartials;
    long long q[7];
    for (int i = 0; i < 7; ++i) {
      long long v = 0;
      for (int t = tid; t < a.vif_tiles[s]; t += kBlock) v += p[(int64_t)t * kVifFxPartials + i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      __syncthreads();
      if ((tid & 63) == 0) redu[tid >> 6] = (unsigned long long)v;
      __syncthreads();
      q[i] = (long long)((redu[0] + redu[1]) + (redu[2] + redu[3]));
    }
    if (tid == 0) {
      const long long num_log = q[0], den_log = q[1], x = q[2], x2 = q[3], n_log = q[4], den_non_log = q[5],
                      num_non_log = q[6];
      rec[0 + s] = (double)num_log / 2048.0 + (double)x2 +
                   ((double)den_non_log - ((double)num_non_log / 16384.0) / 65025.0);
      rec[4 + s] = (double)den_log / 2048.0 - (double)(x + n_log * 17) + (double)den_non_log;
    }
  } else if (a.has_vif && grp < 4) {
    {
      const int s = grp;
      const double* p = a.vif_part[s] + (int64_t)fr * a.vif_tiles[s] * 2;
      const double num = reduce_strided(p, a.vif_tiles[s], 2, red);
      const double den = reduce_strided(p + 1, a.vif_tiles[s], 2, red);
      if (tid == 0) { rec[0 + s] = num; rec[4 + s] = den; }
    }
  }
  if (a.has_adm && grp >= 4 && grp < 8 && a.adm_fx_part[grp - 4]) {
    // fixed-point ADM is finished by adm_fixed_finalize_kernel (24 workgroups per frame instead of 4)
  } else if (a.has_adm && grp >= 4 && grp < 8) {
    {
      const int s = grp - 4;
      const double* p = a.adm_part[s] + (int64_t)fr * a.adm_tiles[s] * 6;
      double q[6];
      for (int i = 0; i < 6; ++i) q[i] = reduce_strided(p + i, a.adm_tiles[s], 6, red);
      if (tid == 0) {
        // adm_cm_s / adm_csf_den_scale_s epilogue: sum over orientations of cbrt(sum) + cbrt(area/32)
        const double area_term = cbrt((double)(a.adm_area[s] / 32.0f));
        rec[8 + s] = cbrt(q[0]) + cbrt(q[1]) + cbrt(q[2]) + 3.0 * area_term;
        rec[12 + s] = cbrt(q[3]) + cbrt(q[4]) + cbrt(q[5]) + 3.0 * area_term;
      }
    }
  }
  if (grp != 8) return;
  if (a.has_motion && a.motion_fx_part) {
    const unsigned long long* p = a.motion_fx_part + (int64_t)fr * a.motion_tiles;
    unsigned long long v = 0;
    for (int i = tid; i < a.motion_tiles; i += kBlock) v += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) redu[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {  // integer_motion.c normalize_and_scale_sad(): (float)(sad / 256.) / (w * h)
      const unsigned long long sad = (redu[0] + redu[1]) + (redu[2] + redu[3]);
      rec[16] = (double)((float)((double)sad / 256.0) / (float)a.motion_wh);
    }
    __syncthreads();
  } else if (a.has_motion) {
    const double sad = reduce_strided(a.motion_part + (int64_t)fr * a.motion_tiles, a.motion_tiles, 1, red);
    if (tid == 0) rec[16] = sad * a.motion_norm;
  }
  for (int pl = 0; pl < a.n_ssim_planes; ++pl) {
    const double s = reduce_strided(a.ssim_part[pl] + (int64_t)fr * a.ssim_tiles[pl], a.ssim_tiles[pl], 1, red);
    if (tid == 0) rec[17 + pl] = s * a.ssim_norm[pl];
  }
  for (int pl = 0; pl < a.n_sse_planes; ++pl) {
    unsigned long long v = 0;
    if (a.sse_use_a[pl]) v += a.sse_part[pl][(int64_t)fr * kSseBlocksPerPlane + tid];
    if (a.sse_part_b[pl]) v += a.sse_part_b[pl][(int64_t)fr * kSseBlocksPerPlane + tid];
    if (a.sse_tile_part[pl]) {
      const unsigned long long* p = a.sse_tile_part[pl] + (int64_t)fr * a.ssim_tiles[pl];
      for (int i = tid; i < a.ssim_tiles[pl]; i += kBlock) v += p[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) redu[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      const unsigned long long sse = (redu[0] + redu[1]) + (redu[2] + redu[3]);
      reinterpret_cast<unsigned long long*>(rec)[20 + pl] = sse;  // integer, bit-cast into the slot
    }
  }
}

// integer_adm.c: a row's sum is complete only across the tiles of that row; then the per-row shift, then the rows.
// One workgroup per (frame, scale, quantity); a thread owns whole rows, so the order of the exact integer adds is free.
__global__ __launch_bounds__(kBlock) void adm_fixed_finalize_kernel(const FinalizeArgs a) {
  __shared__ unsigned long long redu[4];
  const int fr = blockIdx.x, s = blockIdx.y / 6, q = blockIdx.y % 6, tid = threadIdx.x;
  if (!a.adm_fx_part[s]) return;
  const int row = (int)(((int64_t)a.slot_base + (int64_t)fr * a.slot_step) % a.capacity);
  const long long* p = a.adm_fx_part[s] + (int64_t)fr * a.adm_tiles[s] * (kAdmFxRows * 6);
  const int tiles_x = a.adm_fx_tiles_x[s];
  const int shift = q < 3 ? a.adm_fx_num_shift[s] : a.adm_fx_den_shift[s];
  const unsigned long long add = shift > 0 ? 1ull << (shift - 1) : 0ull;
  unsigned long long v = 0;
  for (int r = a.adm_fx_top[s] + tid; r < a.adm_fx_bottom[s]; r += kBlock) {
    const int ty = r / kAdmTileH, lr = r - ty * kAdmTileH + 1;
    unsigned long long rowsum = 0;
    for (int tx = 0; tx < tiles_x; ++tx)
      rowsum += (unsigned long long)p[((int64_t)(ty * tiles_x + tx) * kAdmFxRows + lr) * 6 + q];
    v += (rowsum + add) >> shift;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((tid & 63) == 0) redu[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) a.adm_fx_acc[((int64_t)row * 4 + s) * 6 + q] = (long long)((redu[0] + redu[1]) + (redu[2] + redu[3]));
}

}  // namespace

hipError_t launch_finalize(hipStream_t stream, const FinalizeArgs& args) {
  if (args.n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_kernel, dim3(args.n_frames, 9), dim3(kBlock), 0, stream, args);
  if (args.has_adm && args.adm_fx_part[0])
    hipLaunchKernelGGL(adm_fixed_finalize_kernel, dim3(args.n_frames, 24), dim3(kBlock), 0, stream, args);
  return hipGetLastError();
}

}  // namespace pqa
