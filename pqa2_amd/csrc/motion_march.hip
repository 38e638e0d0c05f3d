// motion as a register-only march: mean absolute difference of the 5-tap-blurred reference luma of consecutive frames,
// no LDS, no workgroup barriers, no vertical halo re-read.
//
// Arithmetic: libvmaf float_motion.c (extract), motion.c (compute_motion / vmaf_image_sad_c), convolution.c
// (convolution_f32_c_s with FILTER_5_s) -- reached through the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419
// -- restated in oracle/vmaf_oracle.c; evaluated as sum |blur(cur - prev)| exactly like motion_kernel (motion.hip), with the
// same operation order per pixel (vertical taps in row order, then horizontal taps left to right): every blurred difference
// is bit-identical to that kernel's, only the order in which they are summed differs.
//
// Why.  motion_kernel stages 20 input rows of 256 columns through LDS for a 252 x 16 tile (25 % of the rows read and
// filtered twice), pays a byte load per sample, lane and image (40 per lane and tile) and a barrier per tile: 5.0 - 5.6 us
// per 2160p frame at 78 % VALU busy.
//
// How (the scheme of adm_march.hip).  A wave owns a stripe of 124 columns and marches DOWN it four rows at a time:
//   lane <-> input columns 2c, 2c + 1 (c = 62 * stripe - 1 + lane; lanes 0 and 63 only feed their neighbours' horizontal
//            taps); one 2-byte / 4-byte load per image and row for 8- / 16-bit samples, row offset in an SGPR;
//   vertical pass in SCATTER form: a new difference row adds c[k] * d to the five output rows it belongs to; an output row
//            is complete two rows later.  With four rows per step a 5-row window touches at most two steps, so what crosses
//            a step is four partial rows -- loop-carried registers, no rolling window to copy;
//   horizontal pass: columns 2c - 2, 2c - 1 and 2c + 2, 2c + 3 are the neighbouring lanes' pairs, fetched with DPP wave
//            shifts (4 per completed row);
//   |.| accumulates per lane and column in f32 for 16 rows, then in double; row validity is a wave-uniform 0 / 1 weight in
//            an SGPR, column validity is applied once to the lane's sums.
// A segment of R rows feeds R + 4 rows ((R + 4) / R instead of 20 / 16).
#include <cmath>

#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

constexpr int kCols = 124;   // output columns per wave (lanes 1..62, two each)

struct MotionMarchArgs {
  const void* ref;
  unsigned pitch;            // elements
  int64_t frame_pitch;       // elements
  const void* prev0;         // reference luma in front of frame 0 of the run (nullable: its motion is 0)
  unsigned prev0_pitch;
  int w, h;
  int aligned, aligned0;     // rows of the clip / of prev0 allow one two-sample load per lane
  int n_stripes, n_sg, seg_rows, n_seg;
  float c[5];
  double* partials;          // [n_frames][n_part]
  int n_part;
};

__device__ __forceinline__ float from_left(float v) {   // lane l <- lane l - 1 across the whole wave
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {  // lane l <- lane l + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// the lane's two samples of one row of one image, as loaded (one register in the fast path) and as floats (exact integers:
// v_cvt_f32_ubyte0 / 1 extract and convert a byte in one instruction)
template <typename T, bool EDGE>
struct PairLoader {
  rsrc_t rs;
  unsigned v0, v1, pitch;
  int h;
  struct Raw { unsigned a, b; };
  __device__ __forceinline__ Raw load(int y) const {
    const unsigned so = (unsigned)mirror1(y, h) * pitch;
    if constexpr (EDGE) {
      return Raw{(unsigned)buf_load<T>(rs, v0, so), (unsigned)buf_load<T>(rs, v1, so)};
    } else if constexpr (sizeof(T) == 1) {
      return Raw{(unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, v0, so, 0), 0u};
    } else {
      return Raw{(unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, v0 * 2u, so * 2u, 0), 0u};
    }
  }
  static __device__ __forceinline__ f2 to_float(const Raw r) {
    if constexpr (EDGE) return f2{(float)r.a, (float)r.b};
    else if constexpr (sizeof(T) == 1) return f2{(float)(r.a & 0xffu), (float)((r.a >> 8) & 0xffu)};
    else return f2{(float)(r.a & 0xffffu), (float)(r.a >> 16)};
  }
};

template <typename T, bool EDGE>
__device__ __forceinline__ double march(const MotionMarchArgs& a, const PairLoader<T, EDGE>& cur, const PairLoader<T, EDGE>& prv,
                                        const int r0, const int r1, const int x_first /* the lane's first column */) {
  const float c0 = a.c[0], c1 = a.c[1], c2 = a.c[2], c3 = a.c[3], c4 = a.c[4];
  // partial vertical sums of the four output rows that are still waiting for input rows ({column 2c, column 2c + 1} each)
  f2 p0 = f2{0.0f, 0.0f}, p1 = p0, p2 = p0, p3 = p0;
  float sad0 = 0.0f, sad1 = 0.0f;
  double dsad0 = 0.0, dsad1 = 0.0;
  const int y_end = min(r1, a.h);
  // differences of the first step's rows, in flight before the loop; every step loads the next step's four rows first
  typename PairLoader<T, EDGE>::Raw cs[4], ps[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { cs[j] = cur.load(r0 - 2 + j); ps[j] = prv.load(r0 - 2 + j); }
  int step = 0;
  for (int y = r0 - 2; y < r1 + 2; y += 4, ++step) {   // this step feeds rows y .. y + 3 and completes rows y - 2 .. y + 1
    f2 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = PairLoader<T, EDGE>::to_float(cs[j]) - PairLoader<T, EDGE>::to_float(ps[j]);   // exact integers
    // (unconditionally: behind the last step the rows are mirrored back into the image and never used -- a conditional
    // prefetch would make the compiler copy sixteen registers at the loop's back edge)
#pragma unroll
    for (int j = 0; j < 4; ++j) { cs[j] = cur.load(y + 4 + j); ps[j] = prv.load(y + 4 + j); }
    // vertical pass, taps in row order exactly as motion_kernel accumulates them (first term: c0 * d + 0)
    const f2 k0 = f2{c0, c0}, k1 = f2{c1, c1}, k2 = f2{c2, c2}, k3 = f2{c3, c3}, k4 = f2{c4, c4};
    f2 o[4];   // completed rows y - 2 .. y + 1
    o[0] = __builtin_elementwise_fma(k4, d[0], p0);
    o[1] = __builtin_elementwise_fma(k4, d[1], __builtin_elementwise_fma(k3, d[0], p1));
    o[2] = __builtin_elementwise_fma(k4, d[2], __builtin_elementwise_fma(k3, d[1], __builtin_elementwise_fma(k2, d[0], p2)));
    o[3] = __builtin_elementwise_fma(k4, d[3], __builtin_elementwise_fma(k3, d[2], __builtin_elementwise_fma(k2, d[1], __builtin_elementwise_fma(k1, d[0], p3))));
    p0 = __builtin_elementwise_fma(k3, d[3], __builtin_elementwise_fma(k2, d[2], __builtin_elementwise_fma(k1, d[1], k0 * d[0])));   // row y + 2
    p1 = __builtin_elementwise_fma(k2, d[3], __builtin_elementwise_fma(k1, d[2], k0 * d[1]));                                              // row y + 3
    p2 = __builtin_elementwise_fma(k1, d[3], k0 * d[2]);                                                                                    // row y + 4
    p3 = k0 * d[3];                                                                                                                         // row y + 5
    // horizontal pass and |.| of the completed rows; a row outside [r0, min(r1, h)) weighs 0 (wave-uniform)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yo = y - 2 + j;
      // 0 / 1 as a float, formed with integer (scalar) operations so that it stays in an SGPR
      const float wrow = __builtin_bit_cast(float, (unsigned)-(int)(yo >= r0 && yo < y_end) & 0x3f800000u);
      const float m0 = o[j].x, m1 = o[j].y;
      const float l0 = from_left(m0), l1 = from_left(m1), q0 = from_right(m0), q1 = from_right(m1);
      const float h0 = fmaf(c4, q0, fmaf(c3, m1, fmaf(c2, m0, fmaf(c1, l1, c0 * l0))));
      const float h1 = fmaf(c4, q1, fmaf(c3, q0, fmaf(c2, m1, fmaf(c1, m0, c0 * l1))));
      sad0 = fmaf(wrow, fabsf(h0), sad0);
      sad1 = fmaf(wrow, fabsf(h1), sad1);
    }
    if ((step & 3) == 3) { dsad0 += (double)sad0; dsad1 += (double)sad1; sad0 = 0.0f; sad1 = 0.0f; }
  }
  dsad0 += (double)sad0;
  dsad1 += (double)sad1;
  // column validity once per wave: lanes 1..62 own their two columns, as far as the image goes
  const int lane = threadIdx.x & 63;
  const bool own = lane >= 1 && lane <= 62;
  return (own && x_first < a.w ? dsad0 : 0.0) + (own && x_first + 1 < a.w ? dsad1 : 0.0);
}

#ifndef PQA_MOTION_SEG_ROWS
#define PQA_MOTION_SEG_ROWS 128  /* multiple of 4; a segment feeds 4 rows more than it owns.  64 while the automatic batch was 32
                                    frames; at 97 frames per launch 128 rows: +4.8 % on the motion chain (profiles/r06i_launch_shapes_ab.txt) */
#endif

template <typename T>
__global__ __launch_bounds__(kBlock) void motion_march_kernel(const MotionMarchArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = xcd_remap(blockIdx.x, a.n_sg * a.n_seg);
  const int sg = id % a.n_sg, seg = id / a.n_sg;
  const int stripe = sg * 4 + wave;
  const int fr = blockIdx.y;
  double* __restrict__ part = a.partials + (int64_t)fr * a.n_part + (int64_t)id * 4 + wave;
  const T* __restrict__ cur = (const T*)a.ref + (int64_t)fr * a.frame_pitch;
  const T* __restrict__ prev = fr == 0 ? (const T*)a.prev0 : cur - a.frame_pitch;
  const unsigned pitch_p = fr == 0 ? a.prev0_pitch : a.pitch;
  if (stripe >= a.n_stripes || prev == nullptr) {   // idle wave of the last group / first frame of the clip: motion_0 = 0
    if (lane == 0) *part = 0.0;
    return;
  }
  const int r0 = seg * a.seg_rows, r1 = min(r0 + a.seg_rows, a.h);
  const int cs = stripe * (kCols / 2);             // first column PAIR of the stripe
  const int c = cs - 1 + lane;                     // this lane's pair: columns 2c, 2c + 1
  const rsrc_t rs_c = make_rsrc(cur, (unsigned)a.h * a.pitch * (unsigned)sizeof(T));
  const rsrc_t rs_p = make_rsrc(prev, (unsigned)a.h * pitch_p * (unsigned)sizeof(T));
  const bool fast = a.aligned && (fr > 0 || a.aligned0) && cs >= 1 && 2 * (cs + kCols / 2) + 1 < a.w;   // wave-uniform
  double v;
  if (fast) {
    const PairLoader<T, false> lc{rs_c, (unsigned)(2 * c), 0u, a.pitch, a.h}, lp{rs_p, (unsigned)(2 * c), 0u, pitch_p, a.h};
    v = march<T, false>(a, lc, lp, r0, r1, 2 * c);
  } else {
    const unsigned v0 = (unsigned)mirror1(2 * c, a.w), v1 = (unsigned)mirror1(2 * c + 1, a.w);
    const PairLoader<T, true> lc{rs_c, v0, v1, a.pitch, a.h}, lp{rs_p, v0, v1, pitch_p, a.h};
    v = march<T, true>(a, lc, lp, r0, r1, 2 * c);
  }
  v = wave_sum(v);
  if (lane == 0) *part = v;
}

}  // namespace

int motion_march_partials(int w, int h) {
  const int n_stripes = (w + kCols - 1) / kCols;
  return ((n_stripes + 3) / 4) * 4 * ((h + PQA_MOTION_SEG_ROWS - 1) / PQA_MOTION_SEG_ROWS);
}

bool launch_motion_march(hipStream_t stream, Elem elem, PlaneRun ref, const void* prev0, int64_t prev0_row_pitch, int n_frames,
                         int w, int h, double* partials, int* n_partials, hipError_t* err) {
  if (elem != ELEM_U8 && elem != ELEM_U16) return false;
  const int es = elem == ELEM_U16 ? 2 : 1;
  if ((int64_t)ref.row_pitch * h * es >= (1ll << 31) || (int64_t)prev0_row_pitch * h * es >= (1ll << 31)) return false;
  MotionMarchArgs a{};
  a.ref = ref.base; a.pitch = (unsigned)ref.row_pitch; a.frame_pitch = ref.frame_pitch;
  a.prev0 = prev0; a.prev0_pitch = (unsigned)prev0_row_pitch;
  a.w = w; a.h = h;
  const uintptr_t two = (uintptr_t)(2 * es - 1);
  a.aligned = ((uintptr_t)ref.base & two) == 0 && ((ref.row_pitch | ref.frame_pitch) & 1) == 0;
  a.aligned0 = ((uintptr_t)prev0 & two) == 0 && (prev0_row_pitch & 1) == 0;
  a.n_stripes = (w + kCols - 1) / kCols;
  a.n_sg = (a.n_stripes + 3) / 4;
  a.seg_rows = PQA_MOTION_SEG_ROWS;
  a.n_seg = (h + a.seg_rows - 1) / a.seg_rows;
  {  // FILTER_5_s: 5 taps, sigma 1.0 (the table motion_kernel uses)
    double v[5], sum = 0.0;
    for (int k = 0; k < 5; ++k) { v[k] = exp(-0.5 * (k - 2) * (k - 2)); sum += v[k]; }
    for (int k = 0; k < 5; ++k) a.c[k] = (float)(v[k] / sum);
  }
  a.partials = partials;
  a.n_part = a.n_sg * 4 * a.n_seg;
  if (n_partials) *n_partials = a.n_part;
  const dim3 grid(a.n_sg * a.n_seg, n_frames), block(kBlock);
  if (elem == ELEM_U16) hipLaunchKernelGGL((motion_march_kernel<uint16_t>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((motion_march_kernel<uint8_t>), grid, block, 0, stream, a);
  *err = hipGetLastError();
  return true;
}

}  // namespace pqa
