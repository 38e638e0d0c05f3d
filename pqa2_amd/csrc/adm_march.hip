// ADM, one scale, as a register-only MARCH: no LDS, no workgroup barriers, no vertical halo.
//
// Arithmetic: libvmaf's float extractor (adm.c compute_adm; adm_tools.c adm_dwt2_s, adm_decouple_s, adm_csf_s,
// adm_csf_den_scale_s, adm_cm_s) -- the code behind the reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 --
// restated in oracle/vmaf_oracle.c:271-426; per coefficient the same operations in the same order as adm_scale_kernel
// (adm.hip), which documents the decouple-as-one-median identity.
//
// Why.  adm_scale_kernel stages a 62 x 16 halo'd grid of coefficients through LDS for 60 x 14 results (18 % recomputed, 1.35 x
// the input read), pays a byte load per sample and lane (36 per lane and tile), two barriers, and about a third of its VALU
// work is LDS stores / reads, address arithmetic, row / column predicates and the mirrored neighbourhood of the masking box.
//
// How.  A wave owns a stripe of 60 coefficient columns (lanes 2..61; lanes 1 and 62 hold the halo columns of the masking box,
// lanes 0 and 63 exist for THEIR horizontal taps only) and marches DOWN the band one coefficient row at a time:
//   lane <-> coefficient column c = 60 * stripe - 2 + lane; it reads ITS two input columns 2c, 2c + 1 of both images with one
//            load per image and row (2 bytes / 4 bytes / 8 bytes for u8 / u16 / f32 planes, row offset in an SGPR);
//   vertical db2 pass in registers over a rolling window of four input rows (two carried, two new per coefficient row), both
//            images packed as {ref, dis} in v_pk_fma_f32, libvmaf's tap order;
//   horizontal pass: the taps at columns 2c - 1 and 2c + 2 are the neighbouring lanes' vertical results, fetched with DPP
//            wave shifts (v_mov_b32 wave_shr:1 / wave_shl:1, 8 per row) -- no LDS transposition;
//   decouple, CSF, masking signal g per coefficient; its 3-wide horizontal sum is two more DPP operations, the 3-tall sum a
//            rolling pair of registers: row i - 1 is finished (threshold, numerator cubes) right after row i's sum exists.
// Rows are partitioned into segments at the rows of libvmaf's accumulation window [top, bottom) (10 % crop): segments
// outside it -- and whole stripes outside [left, right) -- only owe the approximation band the next scale reads (the low
// half of both passes: 12 of the 32 packed FMAs, no decouple, no masking).  A segment inside recomputes ONE row above and
// ONE below (the masking box's neighbours): (L + 2) / L instead of 16 / 14 vertically; horizontally 64 lanes for 60 columns
// as before.  Column masks are applied once per wave (to the lane's sums), row masks do not exist.
// Stripes that touch the left / right image edge (mirrored input columns) and planes whose rows are not aligned for the
// two-sample loads take the EDGE instantiation: one load per sample with mirrored column indices.
#include <type_traits>

#include "adm_chain.h"
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

constexpr int kStripe = 60;   // inner coefficient columns per wave (lanes 2..61)
constexpr int kHalo = 2;      // lanes on each side: one halo COEFFICIENT column, one more that only feeds its horizontal taps

struct AdmMarchArgs {
  const void* ref;
  const void* dis;
  unsigned pitch_r, pitch_d;               // elements
  int64_t frame_pitch_r, frame_pitch_d;    // elements
  int w, h, ow, oh;
  int aligned;                             // rows allow one two-sample load per lane
  int n_stripes, n_sg;                     // stripes, groups of four stripes (one workgroup each)
  int n_frames;
  int left, top, right, bottom;            // accumulation window in band coordinates
  int reg_start[4];                        // row regions [0, top), [top, bottom), [bottom, oh): starts, reg_start[3] = oh
  int seg_rows[3], seg_first[4];           // rows per segment of a region; first segment id of a region, seg_first[3] = total
  float inv_scale, gain_limit, rf_hv, rf_d, k_hv, k_d;
  float* ll_ref;
  float* ll_dis;
  unsigned ll_pitch_r, ll_pitch_d;         // floats
  int64_t ll_frame_pitch_r, ll_frame_pitch_d;
  double* partials;                        // [n_frames][n_part][6]
  int n_part;
};

using namespace admc;

// The same as loaded (before conversion): what a prefetch keeps in registers for one row and BOTH images
template <typename T, bool EDGE> struct RawRow;
template <> struct RawRow<uint8_t, false> { unsigned r, d; };          // two bytes each
template <> struct RawRow<uint16_t, false> { unsigned r, d; };         // two halfwords each
template <> struct RawRow<float, false> { f2 r, d; };
template <typename T> struct RawRow<T, true> { T r0, r1, d0, d1; };
template <> struct RawRow<float, true> { f2 c0, c1; };   // f32 samples need no conversion: paired as {ref, dis} on arrival

template <typename T, bool EDGE>
struct RowLoader {
  rsrc_t rsrc_r, rsrc_d;
  unsigned v0, v1;     // lane offsets (elements) of its two columns; the fast path uses v0 only
  unsigned pitch_r, pitch_d;
  int h;
  float inv_scale;

  __device__ __forceinline__ RawRow<T, EDGE> load(int y /* wave-uniform, any integer */) const {
    const unsigned gy = (unsigned)mirror1(y, h);
    const unsigned so_r = gy * pitch_r, so_d = gy * pitch_d;
    RawRow<T, EDGE> o;
    if constexpr (EDGE && sizeof(T) == 4) {
      o.c0 = f2{buf_load<T>(rsrc_r, v0, so_r), buf_load<T>(rsrc_d, v0, so_d)};
      o.c1 = f2{buf_load<T>(rsrc_r, v1, so_r), buf_load<T>(rsrc_d, v1, so_d)};
    } else if constexpr (EDGE) {
      o.r0 = buf_load<T>(rsrc_r, v0, so_r); o.r1 = buf_load<T>(rsrc_r, v1, so_r);
      o.d0 = buf_load<T>(rsrc_d, v0, so_d); o.d1 = buf_load<T>(rsrc_d, v1, so_d);
    } else if constexpr (sizeof(T) == 1) {
      o.r = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, v0, so_r, 0);
      o.d = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, v0, so_d, 0);
    } else if constexpr (sizeof(T) == 2) {
      o.r = __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, v0 * 2u, so_r * 2u, 0);
      o.d = __builtin_amdgcn_raw_buffer_load_b32(rsrc_d, v0 * 2u, so_d * 2u, 0);
    } else {
      o.r = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, v0 * 4u, so_r * 4u, 0));
      o.d = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_d, v0 * 4u, so_d * 4u, 0));
    }
    return o;
  }
  __device__ __forceinline__ Row convert(const RawRow<T, EDGE>& x) const {
    if constexpr (EDGE && sizeof(T) == 4) {
      return Row{x.c0, x.c1};
    } else if constexpr (EDGE) {
      return Row{PixIO<T>::pair(x.r0, x.d0, inv_scale), PixIO<T>::pair(x.r1, x.d1, inv_scale)};
    } else if constexpr (sizeof(T) == 1) {
      return Row{PixIO<T>::pair((uint8_t)(x.r & 0xffu), (uint8_t)(x.d & 0xffu), inv_scale),
                 PixIO<T>::pair((uint8_t)((x.r >> 8) & 0xffu), (uint8_t)((x.d >> 8) & 0xffu), inv_scale)};
    } else if constexpr (sizeof(T) == 2) {
      return Row{PixIO<T>::pair((uint16_t)(x.r & 0xffffu), (uint16_t)(x.d & 0xffffu), inv_scale),
                 PixIO<T>::pair((uint16_t)(x.r >> 16), (uint16_t)(x.d >> 16), inv_scale)};
    } else {
      return Row{f2{x.r.x, x.d.x}, f2{x.r.y, x.d.y}};
    }
  }
};

template <typename T, bool EDGE>
__device__ __forceinline__ void march_full(const AdmMarchArgs& a, const RowLoader<T, EDGE>& ld, const int lane, const int cs,
                                           const int r0, const int r1, float* __restrict__ ll_r, float* __restrict__ ll_d,
                                           const unsigned ll_voff, double* __restrict__ part) {
  const Consts K{a.gain_limit, a.rf_hv, a.rf_d, a.k_hv, a.k_d};
  const rsrc_t ll_rs_r = make_rsrc(ll_r, ll_r ? (unsigned)a.oh * a.ll_pitch_r * 4u : 0u);
  const rsrc_t ll_rs_d = make_rsrc(ll_d, ll_d ? (unsigned)a.oh * a.ll_pitch_d * 4u : 0u);
  // (last scale: no approximation band is owed; the resource then has zero records and every store is dropped -- cheaper
  // than a branch that splits the horizontal pass into basic blocks)
  // band-level mirror of the masking box's columns: column -1 is column 1, column ow is column ow - 1 (wave-uniform lanes)
  const bool fix_left = EDGE && cs == 0;
  const int lane_m1 = mirror1(-1, a.ow) + kHalo;       // the lane whose column column -1 mirrors to (column 1; 0 when ow == 1)
  const int lane_ow = a.ow - (cs - kHalo);             // the lane that stands for column ow
  const bool fix_right = EDGE && lane_ow <= 63;

  float acc[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // num h, v, d; den h, v, d -- of this lane's column, a few rows
  double dacc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

  // One coefficient row from its four input rows.  OWN: the row belongs to the segment (its approximation band is stored,
  // its denominator terms accumulate); the rows above and below only provide the masking signal.  Leaves the row's pending
  // values in p and returns the 3-wide sum of its masking signal.
  const auto row = [&](const Row& xa, const Row& xb, const Row& xc, const Row& xd, const int i, auto own, Pending& p) -> float {
    constexpr bool OWN = decltype(own)::value;
    const f2 vl0 = dwt_lo(xa.c0, xb.c0, xc.c0, xd.c0), vl1 = dwt_lo(xa.c1, xb.c1, xc.c1, xd.c1);
    const f2 vh0 = dwt_hi(xa.c0, xb.c0, xc.c0, xd.c0), vh1 = dwt_hi(xa.c1, xb.c1, xc.c1, xd.c1);
    const f2 vlm = from_left(vl1), vlp = from_right(vl0), vhm = from_left(vh1), vhp = from_right(vh0);
    const f2 bv = dwt_hi(vlm, vl0, vl1, vlp);   // vertical   (lo-v, hi-h)
    const f2 bh = dwt_lo(vhm, vh0, vh1, vhp);   // horizontal (hi-v, lo-h)
    const f2 bd = dwt_hi(vhm, vh0, vh1, vhp);   // diagonal
    if (OWN) {   // lanes without a column of their own carry an out-of-range offset: the store is dropped
      const f2 ba = dwt_lo(vlm, vl0, vl1, vlp);   // approximation {ref, dis}
      store_f32(ba.x, ll_rs_r, ll_voff, (unsigned)i * a.ll_pitch_r * 4u);
      store_f32(ba.y, ll_rs_d, ll_voff, (unsigned)i * a.ll_pitch_d * 4u);
    }
    float g = decouple(bh, bv, bd, K, p);
    if (EDGE) {
      if (fix_left) {
        const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), lane_m1));
        g = lane == kHalo - 1 ? g1 : g;
      }
      if (fix_right) {
        const float gl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), lane_ow - 1));
        g = lane == lane_ow ? gl : g;
      }
    }
    p.g = g;
    if (OWN) den_accumulate(bh, bv, bd, acc + 3);
    return (from_left(g) + g) + from_right(g);
  };
  const auto finish_row = [&](const Pending& p, const float s2, const float rs_below) { finish(p, s2, rs_below, K, acc); };
  const auto flush = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q) { dacc[q] += (double)acc[q]; acc[q] = 0.0f; }
  };
  const std::integral_constant<bool, true> own{};
  const std::integral_constant<bool, false> guest{};

  // ---- warm-up: the row above the segment (its mirror image at the band's first row) only provides its masking sum
  Row w0, w1, w2, w3;                   // the rolling window: rows 2i - 1, 2i (carried) and 2i + 1, 2i + 2 (new) of row i
  float rs_prev, s2 = 0.0f;
  Pending pa{0.0f, 0.0f, 0.0f, 0.0f}, pb = pa;
  {
    const int wi = mirror1(r0 - 1, a.oh);
    w0 = ld.convert(ld.load(2 * wi - 1)); w1 = ld.convert(ld.load(2 * wi));
    w2 = ld.convert(ld.load(2 * wi + 1)); w3 = ld.convert(ld.load(2 * wi + 2));
    Pending unused;
    rs_prev = row(w0, w1, w2, w3, wi, guest, unused);
    if (wi != r0 - 1) {   // band's first row: the window does not continue from the mirrored warm-up row
      w2 = ld.convert(ld.load(2 * r0 - 1)); w3 = ld.convert(ld.load(2 * r0));
    }
  }
  // ---- march.  One step = row i from the carried rows (c0, c1) and the prefetched ones (converted into n0, n1), then the
  // row above is finished.  Steps alternate between the two halves of the window, of the pending pair and of the prefetch
  // queue, so nothing is ever copied; the loads for rows i + 1 and i + 2 are in flight while row i is computed (a lone wave
  // of a deep scale's small launch would otherwise spend a full memory latency per row).
  RawRow<T, EDGE> qa0 = ld.load(2 * r0 + 1), qa1 = ld.load(2 * r0 + 2), qb0 = ld.load(2 * r0 + 3), qb1 = ld.load(2 * r0 + 4);
  const auto step = [&](const Row& c0, const Row& c1, Row& n0, Row& n1, const Pending& above, Pending& mine,
                        RawRow<T, EDGE>& q0, RawRow<T, EDGE>& q1, const int i, auto is_own) {
    n0 = ld.convert(q0); n1 = ld.convert(q1);
    if (decltype(is_own)::value) { q0 = ld.load(2 * i + 5); q1 = ld.load(2 * i + 6); }   // row i + 2's new rows
    const float rs = row(c0, c1, n0, n1, i, is_own, mine);
    finish_row(above, s2, rs);
    s2 = rs_prev + rs;
    rs_prev = rs;
  };
  int i = r0;
  for (; i + 1 < r1; i += 2) {
    step(w2, w3, w0, w1, pa, pb, qa0, qa1, i, own);
    step(w0, w1, w2, w3, pb, pa, qb0, qb1, i + 1, own);
    if (((i - r0) & 6) == 6) flush();
  }
  if (i < r1) {   // odd row count: one more step, then back to the canonical halves
    step(w2, w3, w0, w1, pa, pb, qa0, qa1, i, own);
    w2 = w0; w3 = w1; pa = pb; qa0 = qb0; qa1 = qb1;
  }
  // ---- the row below the segment again only provides its sum; below the band's last row that sum is the last row's own
  if (r1 < a.oh) {
    step(w2, w3, w0, w1, pa, pb, qa0, qa1, r1, guest);
  } else {
    finish_row(pa, s2, rs_prev);
  }
  flush();
  // column mask once per wave: the lane has a column of its own inside [left, right)
  const int c = cs - kHalo + lane;
  const bool col_win = lane >= kHalo && lane < kHalo + kStripe && c >= a.left && c < a.right;
  const double rf3_hv = (double)(a.rf_hv * a.rf_hv * a.rf_hv), rf3_d = (double)(a.rf_d * a.rf_d * a.rf_d);
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    double v = wave_sum(col_win ? dacc[q] : 0.0);
    if (q >= 3) v *= q == 5 ? rf3_d : rf3_hv;
    if (lane == 0) part[q] = v;
  }
}

// approximation band only (rows / stripes outside the accumulation window): the low half of both passes
template <typename T, bool EDGE>
__device__ __forceinline__ void march_ll(const AdmMarchArgs& a, const RowLoader<T, EDGE>& ld, const int r0, const int r1,
                                         float* __restrict__ ll_r, float* __restrict__ ll_d, const unsigned ll_voff) {
  const rsrc_t ll_rs_r = make_rsrc(ll_r, (unsigned)a.oh * a.ll_pitch_r * 4u);
  const rsrc_t ll_rs_d = make_rsrc(ll_d, (unsigned)a.oh * a.ll_pitch_d * 4u);
  Row w0, w1, w2, w3;
  w2 = ld.convert(ld.load(2 * r0 - 1)); w3 = ld.convert(ld.load(2 * r0));
  RawRow<T, EDGE> qa0 = ld.load(2 * r0 + 1), qa1 = ld.load(2 * r0 + 2), qb0 = ld.load(2 * r0 + 3), qb1 = ld.load(2 * r0 + 4);
  const auto step = [&](const Row& c0, const Row& c1, Row& n0, Row& n1, RawRow<T, EDGE>& q0, RawRow<T, EDGE>& q1, const int i) {
    n0 = ld.convert(q0); n1 = ld.convert(q1);
    q0 = ld.load(2 * i + 5); q1 = ld.load(2 * i + 6);
    const f2 vl0 = dwt_lo(c0.c0, c1.c0, n0.c0, n1.c0), vl1 = dwt_lo(c0.c1, c1.c1, n0.c1, n1.c1);
    const f2 ba = dwt_lo(from_left(vl1), vl0, vl1, from_right(vl0));
    store_f32(ba.x, ll_rs_r, ll_voff, (unsigned)i * a.ll_pitch_r * 4u);
    store_f32(ba.y, ll_rs_d, ll_voff, (unsigned)i * a.ll_pitch_d * 4u);
  };
  int i = r0;
  for (; i + 1 < r1; i += 2) {
    step(w2, w3, w0, w1, qa0, qa1, i);
    step(w0, w1, w2, w3, qb0, qb1, i + 1);
  }
  if (i < r1) step(w2, w3, w0, w1, qa0, qa1, i);
}

#ifndef PQA_ADM_MARCH_OCC
#define PQA_ADM_MARCH_OCC 4
#endif
template <typename T>
__global__ __launch_bounds__(kBlock, PQA_ADM_MARCH_OCC) void adm_march_kernel(const AdmMarchArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Which (frame, segment, stripe group) a workgroup takes: ids as they come (NOT xcd_remap: a contiguous range of ids per
  // XCD is the same rows of every frame, and segments outside the crop window cost a third of those inside) and
  // longest-first -- the segments inside the window of ALL frames, then the approximation-band-only regions below and above
  // (adm_pyramid.hip has the measurements).
  const int n_seg = a.seg_first[3];
  const int sg = blockIdx.x % a.n_sg;
  const int t = blockIdx.x / a.n_sg;
  const int fr = t % a.n_frames, k = t / a.n_frames;
  const int seg = (k + a.seg_first[1]) % n_seg;        // region 1 first, then region 2, then region 0
  const int id = seg * a.n_sg + sg;
  const int stripe = sg * 4 + wave;
  double* __restrict__ part = a.partials + ((int64_t)fr * a.n_part + (int64_t)id * 4 + wave) * 6;
  const int region = seg >= a.seg_first[2] ? 2 : seg >= a.seg_first[1] ? 1 : 0;
  const int r0 = a.reg_start[region] + (seg - a.seg_first[region]) * a.seg_rows[region];
  const int r1 = min(r0 + a.seg_rows[region], a.reg_start[region + 1]);
  const int cs = stripe * kStripe;
  const bool in_win = region == 1 && cs < a.right && cs + kStripe > a.left;   // wave-uniform
  if (stripe >= a.n_stripes || (!in_win && !a.ll_ref)) {   // idle wave of the last group / nothing owed at the last scale
    if (lane < 6) part[lane] = 0.0;
    return;
  }
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  float* __restrict__ ll_r = a.ll_ref ? a.ll_ref + (int64_t)fr * a.ll_frame_pitch_r : nullptr;
  float* __restrict__ ll_d = a.ll_ref ? a.ll_dis + (int64_t)fr * a.ll_frame_pitch_d : nullptr;
  const int c = cs - kHalo + lane;
  // lanes with a band column of their own store it; the others carry an offset no buffer is large enough for
  const unsigned ll_voff = (lane >= kHalo && lane < kHalo + kStripe && c < a.ow) ? (unsigned)c * 4u : 0x80000000u;
  // every input column of the stripe (halo lanes included) inside the image and rows aligned for two-sample loads?
  const bool fast = a.aligned && cs >= kHalo && 2 * (cs + kStripe + kHalo - 1) + 1 < a.w;   // wave-uniform
  const unsigned bytes_r = (unsigned)a.h * a.pitch_r * (unsigned)sizeof(T), bytes_d = (unsigned)a.h * a.pitch_d * (unsigned)sizeof(T);
  if (fast) {
    const RowLoader<T, false> ld{make_rsrc(ref, bytes_r), make_rsrc(dis, bytes_d), (unsigned)(2 * c), 0u, a.pitch_r, a.pitch_d, a.h, a.inv_scale};
    if (in_win) {
      march_full<T, false>(a, ld, lane, cs, r0, r1, ll_r, ll_d, ll_voff, part);
    } else {
      march_ll<T, false>(a, ld, r0, r1, ll_r, ll_d, ll_voff);
      if (lane < 6) part[lane] = 0.0;
    }
  } else {
    const RowLoader<T, true> ld{make_rsrc(ref, bytes_r), make_rsrc(dis, bytes_d), (unsigned)mirror1(2 * c, a.w), (unsigned)mirror1(2 * c + 1, a.w),
                                a.pitch_r, a.pitch_d, a.h, a.inv_scale};
    if (in_win) {
      march_full<T, true>(a, ld, lane, cs, r0, r1, ll_r, ll_d, ll_voff, part);
    } else {
      march_ll<T, true>(a, ld, r0, r1, ll_r, ll_d, ll_voff);
      if (lane < 6) part[lane] = 0.0;
    }
  }
}

// Rows per segment: long enough that the two recomputed rows of a segment inside the window are cheap ((L + 2) / L), short
// enough that a launch has several rounds of waves per SIMD to balance.  Measured on the box (gpurun_out/r04g_*): 32 rows
// everywhere beat shorter segments for the approximation-band-only regions (a wave's prologue -- six dependent row loads --
// weighs more the shorter its march) and shorter segments bought the deep scales 3-8 % of launches that cost 0.5-1.5 us per
// frame; the knobs stay for the next sweep.  Segments are a function of the band's geometry only (a frame's record must not
// depend on the batch it shares a launch with).
#ifndef PQA_ADM_SEG_ROWS
#define PQA_ADM_SEG_ROWS 32      /* longest segment */
#endif
#ifndef PQA_ADM_SEG_MIN
#define PQA_ADM_SEG_MIN 32       /* shortest */
#endif
#ifndef PQA_ADM_WAVES
#define PQA_ADM_WAVES 384        /* waves per frame and region the segment length aims at (between the two bounds) */
#endif

struct Partition {
  int reg_start[4], seg_rows[3], seg_first[4];
};
Partition partition_rows(int oh, int top, int bottom, int n_stripes) {
  Partition p{};
  p.reg_start[0] = 0; p.reg_start[1] = top; p.reg_start[2] = bottom; p.reg_start[3] = oh;
  int first = 0;
  for (int r = 0; r < 3; ++r) {
    const int len = p.reg_start[r + 1] - p.reg_start[r];
    int rows = (len * n_stripes + PQA_ADM_WAVES - 1) / PQA_ADM_WAVES;
    rows = rows < PQA_ADM_SEG_MIN ? PQA_ADM_SEG_MIN : rows > PQA_ADM_SEG_ROWS ? PQA_ADM_SEG_ROWS : rows;
    const int n = (len + rows - 1) / rows;           // 0 for an empty region
    p.seg_rows[r] = n ? (len + n - 1) / n : 1;       // evened out
    p.seg_first[r] = first;
    first += n ? (len + p.seg_rows[r] - 1) / p.seg_rows[r] : 0;
  }
  p.seg_first[3] = first;
  return p;
}

}  // namespace

int adm_march_partials(int band_w, int band_h) {
  const int top = (int)(band_h * 0.1 - 0.5);
  const int n_stripes = (band_w + kStripe - 1) / kStripe;
  const Partition p = partition_rows(band_h, top, band_h - top, n_stripes);
  return ((n_stripes + 3) / 4) * 4 * p.seg_first[3];
}

bool launch_adm_march(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h,
                      float inv_scale, float gain_limit, MutPlaneRun ll_ref, MutPlaneRun ll_dis, double* partials,
                      int* n_partials, hipError_t* err) {
  const int es = elem == ELEM_U8 ? 1 : elem == ELEM_U16 ? 2 : 4;
  // 32-bit buffer offsets: a plane (and an approximation band) must stay below 2 GiB
  if ((int64_t)ref.row_pitch * h * es >= (1ll << 31) || (int64_t)dis.row_pitch * h * es >= (1ll << 31)) return false;
  AdmMarchArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.pitch_r = (unsigned)ref.row_pitch; a.pitch_d = (unsigned)dis.row_pitch;
  a.frame_pitch_r = ref.frame_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.ow = (w + 1) / 2; a.oh = (h + 1) / 2;
  if (ll_ref.base && ((int64_t)ll_ref.row_pitch * a.oh * 4 >= (1ll << 31) || (int64_t)ll_dis.row_pitch * a.oh * 4 >= (1ll << 31))) return false;
  // one load of two samples per lane: every row must start on a multiple of two samples
  const uintptr_t two = (uintptr_t)(2 * es - 1);
  a.aligned = (((uintptr_t)ref.base | (uintptr_t)dis.base) & two) == 0 &&
              ((ref.row_pitch | dis.row_pitch | ref.frame_pitch | dis.frame_pitch) & 1) == 0;
  a.n_stripes = (a.ow + kStripe - 1) / kStripe;
  a.n_sg = (a.n_stripes + 3) / 4;
  const double border = 0.1;  // ADM_BORDER_FACTOR
  a.left = (int)(a.ow * border - 0.5);
  a.top = (int)(a.oh * border - 0.5);
  a.right = a.ow - a.left;
  a.bottom = a.oh - a.top;
  const Partition p = partition_rows(a.oh, a.top, a.bottom, a.n_stripes);
  for (int r = 0; r < 4; ++r) { a.reg_start[r] = p.reg_start[r]; a.seg_first[r] = p.seg_first[r]; }
  for (int r = 0; r < 3; ++r) a.seg_rows[r] = p.seg_rows[r];
  a.inv_scale = inv_scale; a.gain_limit = gain_limit;
  a.rf_hv = 1.0f / adm_dwt_quant_step(scale, 1);
  a.rf_d = 1.0f / adm_dwt_quant_step(scale, 2);
  a.k_hv = a.rf_hv / 30.0f;
  a.k_d = a.rf_d / 30.0f;
  a.ll_ref = (float*)ll_ref.base; a.ll_dis = (float*)ll_dis.base;
  a.ll_pitch_r = (unsigned)ll_ref.row_pitch; a.ll_pitch_d = (unsigned)ll_dis.row_pitch;
  a.ll_frame_pitch_r = ll_ref.frame_pitch; a.ll_frame_pitch_d = ll_dis.frame_pitch;
  a.partials = partials;
  a.n_part = a.n_sg * 4 * p.seg_first[3];
  if (n_partials) *n_partials = a.n_part;
  a.n_frames = n_frames;
  const dim3 grid((unsigned)a.n_sg * p.seg_first[3] * n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((adm_march_kernel<uint8_t>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((adm_march_kernel<uint16_t>), grid, block, 0, stream, a); break;
    case ELEM_F32: hipLaunchKernelGGL((adm_march_kernel<float>), grid, block, 0, stream, a); break;
  }
  *err = hipGetLastError();
  return true;
}

}  // namespace pqa
