// ADM scales 0 AND 1 in one launch: the approximation band of scale 0 never leaves the registers.
//
// Arithmetic: exactly adm_march.hip's (adm_chain.h), which is libvmaf's float extractor (adm.c compute_adm; adm_tools.c)
// restated in oracle/vmaf_oracle.c:271-426; per coefficient bit-identical to the one-scale-per-launch kernels.
//
// Why.  With the march kernel ADM became a memory problem: scale 0 writes its approximation band as two f32 planes (16.6 MB
// per 2160p frame pair) that scale 1 reads back -- 33 of the chain's 60 MB per frame, at the ~4 TB/s these access patterns
// reach 8 of its 15.7 us.  A batch of 32 frames is 531 MB of it: twice the Infinity Cache, so the re-read comes from HBM.
//
// How.  A lane of adm_march_kernel holds two adjacent INPUT columns and produces one coefficient column.  Here a lane holds
// FOUR input columns (one dword of 8-bit samples per image and row) and produces TWO adjacent scale-0 columns 2c, 2c + 1 --
// which are precisely the two input columns a scale-1 lane needs.  So the same lane, with the same neighbours, runs the
// scale-1 march on its own scale-0 results:
//   scale 0, row i    : vertical db2 on 4 columns, horizontal taps 4c - 1 and 4c + 4 from the neighbouring lanes (DPP), two
//                       coefficients A = 2c and B = 2c + 1: approximation pair {A, B} -> a row of the scale-1 window;
//                       decouple / CSF / masking for both (the masking box's columns 2c - 1 and 2c + 2 are the neighbours' B
//                       and A);
//   scale 1, row j    : after scale-0 rows 2j + 1 and 2j + 2: one step of adm_march.hip's loop on the four window rows
//                       2j - 1 .. 2j + 2, its approximation band (the input of scale 2) is the only plane written.
// A wave owns 60 scale-1 columns (lanes 2..61) = 120 scale-0 columns = 240 input columns and a segment of scale-1 rows
// [R0, R1); it computes scale-0 rows 2 R0 - 3 .. 2 R1 + 2 (three above and below for scale 1's window and box) and scale-1
// rows R0 - 1 .. R1.  The decouple / masking chain of a row runs only where some coefficient of it can reach an accumulated
// threshold (wave-uniform tests against libvmaf's 10 % crop); elsewhere a row costs the low half of the filters.
// Taken for frames whose width and height are multiples of 4 and at least 128 (every broadcast format; the band mirrors then
// touch one column / row per scale); everything else runs one scale per launch.
#include <type_traits>

#include "adm_chain.h"
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {
namespace {

using namespace admc;

constexpr int kP1 = 60;   // scale-1 columns per wave (lanes 2..61)
constexpr int kPH = 2;    // lanes on each side that only feed taps / the masking box

struct PyramidArgs {
  const void* ref;
  const void* dis;
  unsigned pitch_r, pitch_d;               // elements
  int64_t frame_pitch_r, frame_pitch_d;    // elements
  int w, h, ow0, oh0, ow1, oh1;
  int aligned;                             // rows allow one four-sample load per lane
  int n_stripes, n_sg, seg_rows, n_seg;    // seg_rows: scale-1 rows per segment
  int n_frames;
  int left0, top0, right0, bottom0, left1, top1, right1, bottom1;
  float inv_scale;
  Consts k0, k1;
  float* ll_ref;                           // approximation band of scale 1 (ow1 x oh1): the input of scale 2
  float* ll_dis;
  unsigned ll_pitch_r, ll_pitch_d;         // floats
  int64_t ll_frame_pitch_r, ll_frame_pitch_d;
  double* part0;                           // [n_frames][n_part][6] per scale
  double* part1;
  int n_part;
};

// four input columns 4c .. 4c + 3 of one row, each {ref, dis}
struct Row4 {
  f2 c[4];
};
template <typename T, bool EDGE> struct Raw4;
template <> struct Raw4<uint8_t, false> { unsigned r, d; };                       // four bytes each
template <> struct Raw4<uint16_t, false> { unsigned r0, r1, d0, d1; };            // four halfwords each
template <typename T> struct Raw4<T, true> { T r[4], d[4]; };

template <typename T, bool EDGE>
struct Loader4 {
  rsrc_t rsrc_r, rsrc_d;
  unsigned v[4];       // lane offsets (elements) of its four columns; the fast path uses v[0] only
  unsigned pitch_r, pitch_d;
  int h;
  float inv_scale;
  __device__ __forceinline__ Raw4<T, EDGE> load(int y /* wave-uniform, any integer */) const {
    const unsigned gy = (unsigned)mirror1(y, h);
    const unsigned so_r = gy * pitch_r, so_d = gy * pitch_d;
    Raw4<T, EDGE> o;
    if constexpr (EDGE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { o.r[k] = buf_load<T>(rsrc_r, v[k], so_r); o.d[k] = buf_load<T>(rsrc_d, v[k], so_d); }
    } else if constexpr (sizeof(T) == 1) {
      o.r = __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, v[0], so_r, 0);
      o.d = __builtin_amdgcn_raw_buffer_load_b32(rsrc_d, v[0], so_d, 0);
    } else {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      const u2 r = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, v[0] * 2u, so_r * 2u, 0));
      const u2 d = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_d, v[0] * 2u, so_d * 2u, 0));
      o.r0 = r[0]; o.r1 = r[1]; o.d0 = d[0]; o.d1 = d[1];
    }
    return o;
  }
  __device__ __forceinline__ Row4 convert(const Raw4<T, EDGE>& x) const {
    Row4 o;
    if constexpr (EDGE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o.c[k] = PixIO<T>::pair(x.r[k], x.d[k], inv_scale);
    } else if constexpr (sizeof(T) == 1) {
      o.c[0] = PixIO<T>::pair((uint8_t)(x.r & 0xffu), (uint8_t)(x.d & 0xffu), inv_scale);
      o.c[1] = PixIO<T>::pair((uint8_t)((x.r >> 8) & 0xffu), (uint8_t)((x.d >> 8) & 0xffu), inv_scale);
      o.c[2] = PixIO<T>::pair((uint8_t)((x.r >> 16) & 0xffu), (uint8_t)((x.d >> 16) & 0xffu), inv_scale);
      o.c[3] = PixIO<T>::pair((uint8_t)(x.r >> 24), (uint8_t)(x.d >> 24), inv_scale);
    } else {
      o.c[0] = PixIO<T>::pair((uint16_t)(x.r0 & 0xffffu), (uint16_t)(x.d0 & 0xffffu), inv_scale);
      o.c[1] = PixIO<T>::pair((uint16_t)(x.r0 >> 16), (uint16_t)(x.d0 >> 16), inv_scale);
      o.c[2] = PixIO<T>::pair((uint16_t)(x.r1 & 0xffffu), (uint16_t)(x.d1 & 0xffffu), inv_scale);
      o.c[3] = PixIO<T>::pair((uint16_t)(x.r1 >> 16), (uint16_t)(x.d1 >> 16), inv_scale);
    }
    return o;
  }
};

// what a scale's masking box carries from row to row for one coefficient column
struct BoxState {
  float s2 = 0.0f, rs_prev = 0.0f;     // sum of the two rows above; the row above
  Pending pend{0.0f, 0.0f, 0.0f, 0.0f};
};

template <typename T, bool EDGE>
__device__ __forceinline__ void pyramid_march(const PyramidArgs& a, const Loader4<T, EDGE>& ld, const int lane, const int stripe,
                                              const int R0, const int R1, const int fr, double* __restrict__ part0,
                                              double* __restrict__ part1, double (*dsum)[kBlock]) {
  const int cs1 = stripe * kP1;                      // first scale-1 column of the stripe
  const int c1 = cs1 - kPH + lane;                   // this lane's scale-1 column; its scale-0 columns are 2 c1, 2 c1 + 1
  const bool inner = lane >= kPH && lane < kPH + kP1;
  // which rows / stripes can reach an accumulated threshold at all (wave-uniform)
  const bool win0 = 2 * cs1 < a.right0 && 2 * (cs1 + kP1) > a.left0;
  const bool win1 = cs1 < a.right1 && cs1 + kP1 > a.left1;
  // band-level mirrors of the columns just outside a band (first / last stripe only): scale-0 column -1 = lane 1's B is
  // column 1 = lane 2's B; scale-0 column ow0 = the A of the lane that stands for scale-1 column ow1 is column ow0 - 1 = its
  // left neighbour's B.  The same two lanes hold the scale-1 window's columns -1 (slot c1) and ow0 (slot c0), and scale 1's
  // own masking columns -1 / ow1.
  const bool first = EDGE && stripe == 0;
  const int lane_e = a.ow1 - (cs1 - kPH);            // the lane that stands for scale-1 column ow1
  const bool last = EDGE && lane_e <= 63;
  const bool at_m1 = first && lane == kPH - 1, at_e = last && lane == lane_e;

  float* __restrict__ ll_r = a.ll_ref + (int64_t)fr * a.ll_frame_pitch_r;
  float* __restrict__ ll_d = a.ll_dis + (int64_t)fr * a.ll_frame_pitch_d;
  const rsrc_t ll_rs_r = make_rsrc(ll_r, (unsigned)a.oh1 * a.ll_pitch_r * 4u);
  const rsrc_t ll_rs_d = make_rsrc(ll_d, (unsigned)a.oh1 * a.ll_pitch_d * 4u);
  const unsigned ll_voff = (inner && c1 < a.ow1) ? (unsigned)c1 * 4u : 0x80000000u;   // others: no buffer is that large

  float accA[6] = {0, 0, 0, 0, 0, 0}, accB[6] = {0, 0, 0, 0, 0, 0}, acc1[6] = {0, 0, 0, 0, 0, 0};   // num h v d, den h v d
  // The per-lane double sums (12 of them = 24 registers) live in LDS, one slot per thread and sum: a flush is a read, an add
  // and a write per sum every eight scale-0 rows, and the kernel fits three waves per SIMD without spilling (187 -> under 168
  // VGPRs; with the sums in registers the compiler spilled 36 dwords at three waves, and two waves measured 5 % slower).
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 12; ++q) dsum[q][tid] = 0.0;
  const float mA = (inner && 2 * c1 >= a.left0 && 2 * c1 < a.right0) ? 1.0f : 0.0f;
  const float mB = (inner && 2 * c1 + 1 >= a.left0 && 2 * c1 + 1 < a.right0) ? 1.0f : 0.0f;
  const auto flush = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      dsum[q][tid] += (double)fmaf(mB, accB[q], mA * accA[q]);
      dsum[6 + q][tid] += (double)acc1[q];
      accA[q] = 0.0f; accB[q] = 0.0f; acc1[q] = 0.0f;
    }
  };

  BoxState bxA, bxB, bx1;
  // ---- scale 0, row i: from the carried rows (xa, xb) and the new ones (xc, xd) -> its approximation pair {A, B}; the
  // chain (when asked for) leaves the row's pending values and finishes row i - 1
  const auto step0 = [&](const Row4& xa, const Row4& xb, const Row4& xc, const Row4& xd, const int i, const bool chain,
                         const bool acc_this, const bool acc_above) -> Row {
    f2 vl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) vl[k] = dwt_lo(xa.c[k], xb.c[k], xc.c[k], xd.c[k]);
    const f2 vlm = from_left(vl[3]), vlp = from_right(vl[0]);
    Row ll{dwt_lo(vlm, vl[0], vl[1], vl[2]), dwt_lo(vl[1], vl[2], vl[3], vlp)};
    float rsA = 0.0f, rsB = 0.0f;
    Pending pA{0.0f, 0.0f, 0.0f, 0.0f}, pB = pA;
    if (chain) {
      f2 vh[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) vh[k] = dwt_hi(xa.c[k], xb.c[k], xc.c[k], xd.c[k]);
      const f2 vhm = from_left(vh[3]), vhp = from_right(vh[0]);
      const f2 bvA = dwt_hi(vlm, vl[0], vl[1], vl[2]), bhA = dwt_lo(vhm, vh[0], vh[1], vh[2]), bdA = dwt_hi(vhm, vh[0], vh[1], vh[2]);
      const f2 bvB = dwt_hi(vl[1], vl[2], vl[3], vlp), bhB = dwt_lo(vh[1], vh[2], vh[3], vhp), bdB = dwt_hi(vh[1], vh[2], vh[3], vhp);
      float gA = decouple(bhA, bvA, bdA, a.k0, pA);
      float gB = decouple(bhB, bvB, bdB, a.k0, pB);
      if (EDGE) {
        if (first) { const float t = from_right(gB); gB = at_m1 ? t : gB; }
        if (last) { const float t = from_left(gB); gA = at_e ? t : gA; }
      }
      pA.g = gA; pB.g = gB;
      if (acc_this) { den_accumulate(bhA, bvA, bdA, accA + 3); den_accumulate(bhB, bvB, bdB, accB + 3); }
      rsA = (from_left(gB) + gA) + gB;
      rsB = (gA + gB) + from_right(gA);
    }
    if (acc_above) { finish(bxA.pend, bxA.s2, rsA, a.k0, accA); finish(bxB.pend, bxB.s2, rsB, a.k0, accB); }
    bxA.s2 = bxA.rs_prev + rsA; bxA.rs_prev = rsA; bxA.pend = pA;
    bxB.s2 = bxB.rs_prev + rsB; bxB.rs_prev = rsB; bxB.pend = pB;
    (void)i;
    return ll;
  };
  // ---- scale 1, row j from its four window rows (adm_march.hip's step)
  const auto step1 = [&](const Row& xa, const Row& xb, const Row& xc, const Row& xd, const int j, const bool chain, const bool own,
                         const bool acc_this, const bool acc_above) {
    const f2 vl0 = dwt_lo(xa.c0, xb.c0, xc.c0, xd.c0), vl1 = dwt_lo(xa.c1, xb.c1, xc.c1, xd.c1);
    const f2 vlm = from_left(vl1), vlp = from_right(vl0);
    if (own) {
      const f2 ba = dwt_lo(vlm, vl0, vl1, vlp);
      store_f32(ba.x, ll_rs_r, ll_voff, (unsigned)j * a.ll_pitch_r * 4u);
      store_f32(ba.y, ll_rs_d, ll_voff, (unsigned)j * a.ll_pitch_d * 4u);
    }
    float rs = 0.0f;
    Pending p{0.0f, 0.0f, 0.0f, 0.0f};
    if (chain) {
      const f2 vh0 = dwt_hi(xa.c0, xb.c0, xc.c0, xd.c0), vh1 = dwt_hi(xa.c1, xb.c1, xc.c1, xd.c1);
      const f2 vhm = from_left(vh1), vhp = from_right(vh0);
      const f2 bv = dwt_hi(vlm, vl0, vl1, vlp), bh = dwt_lo(vhm, vh0, vh1, vhp), bd = dwt_hi(vhm, vh0, vh1, vhp);
      float g = decouple(bh, bv, bd, a.k1, p);
      if (EDGE) {
        if (first) { const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), kPH + 1)); g = at_m1 ? t : g; }
        if (last) { const float t = from_left(g); g = at_e ? t : g; }
      }
      p.g = g;
      if (acc_this) den_accumulate(bh, bv, bd, acc1 + 3);
      rs = (from_left(g) + g) + from_right(g);
    }
    if (acc_above) finish(bx1.pend, bx1.s2, rs, a.k1, acc1);
    bx1.s2 = bx1.rs_prev + rs; bx1.rs_prev = rs; bx1.pend = p;
  };
  // the scale-1 window's view of a scale-0 approximation pair: at the band's first / last column the mirrored neighbour
  const auto band_edges = [&](Row r) -> Row {
    if (EDGE) {
      if (first) { const f2 t = from_right(r.c1); r.c1 = at_m1 ? t : r.c1; }     // column -1 <- column 1
      if (last) { const f2 t = from_left(r.c1); r.c0 = at_e ? t : r.c0; }        // column ow0 <- column ow0 - 1
    }
    return r;
  };

  // ---- the march.  Scale-1 row j needs scale-0 rows 2j - 1 .. 2j + 2; per iteration: scale-0 rows 2j + 1 and 2j + 2, then
  // scale-1 row j.  Scale-0 row i reads input rows 2i - 1 .. 2i + 2: two carried, two new (prefetched one iteration ahead).
  const int j_first = R0 - 1, j_last = R1 < a.oh1 ? R1 : R1 - 1;
  const int i0 = 2 * j_first - 1;                    // first scale-0 row computed: 2 R0 - 3
  Row4 x0, x1, x2, x3;
  x2 = ld.convert(ld.load(2 * i0 - 1)); x3 = ld.convert(ld.load(2 * i0));
  Raw4<T, EDGE> q[4];
  q[0] = ld.load(2 * i0 + 1); q[1] = ld.load(2 * i0 + 2); q[2] = ld.load(2 * i0 + 3); q[3] = ld.load(2 * i0 + 4);
  // prologue: scale-0 rows i0, i0 + 1 (approximation band only) fill the upper half of the first window
  Row w0, w1, w2, w3;
  {
    x0 = ld.convert(q[0]); x1 = ld.convert(q[1]);
    q[0] = ld.load(2 * i0 + 5); q[1] = ld.load(2 * i0 + 6);
    w2 = band_edges(step0(x2, x3, x0, x1, i0, false, false, false));
    x2 = ld.convert(q[2]); x3 = ld.convert(q[3]);
    q[2] = ld.load(2 * i0 + 7); q[3] = ld.load(2 * i0 + 8);
    w3 = band_edges(step0(x0, x1, x2, x3, i0 + 1, false, false, false));
  }
  const auto need0 = [&](int i) { return win0 && i >= a.top0 - 1 && i <= a.bottom0 && i >= 2 * R0 - 1 && i <= 2 * R1; };
  const auto accum0 = [&](int i) { return win0 && i >= a.top0 && i < a.bottom0 && i >= 2 * R0 && i < 2 * R1; };
  const auto need1 = [&](int j) { return win1 && j >= a.top1 - 1 && j <= a.bottom1; };
  const auto accum1 = [&](int j) { return win1 && j >= a.top1 && j < a.bottom1 && j >= R0 && j < R1; };
  for (int j = j_first; j <= j_last; ++j) {
    const int ia = 2 * j + 1, ib = 2 * j + 2;        // the two new scale-0 rows
    w0 = w2; w1 = w3;
    x0 = ld.convert(q[0]); x1 = ld.convert(q[1]);
    q[0] = ld.load(2 * ia + 5); q[1] = ld.load(2 * ia + 6);          // rows 2 (ia + 2) + 1, + 2: next iteration's first step
    w2 = band_edges(step0(x2, x3, x0, x1, ia, need0(ia), accum0(ia), accum0(ia - 1)));
    x2 = ld.convert(q[2]); x3 = ld.convert(q[3]);
    q[2] = ld.load(2 * ib + 5); q[3] = ld.load(2 * ib + 6);
    w3 = band_edges(step0(x0, x1, x2, x3, ib, need0(ib), accum0(ib), accum0(ib - 1)));
    // band-level mirror of the ROWS just outside the scale-0 band (first / last segment of a frame)
    if (j == 0) w0 = w2;                             // row -1 <- row 1
    if (ib == a.oh0) w3 = w2;                        // row oh0 <- row oh0 - 1
    step1(w0, w1, w2, w3, j, need1(j), j >= R0 && j < R1, accum1(j), accum1(j - 1));
    if (((j - j_first) & 3) == 3) flush();
  }
  flush();
  const bool col1 = inner && c1 >= a.left1 && c1 < a.right1;
  const double rf0_hv = (double)(a.k0.rf_hv * a.k0.rf_hv * a.k0.rf_hv), rf0_d = (double)(a.k0.rf_d * a.k0.rf_d * a.k0.rf_d);
  const double rf1_hv = (double)(a.k1.rf_hv * a.k1.rf_hv * a.k1.rf_hv), rf1_d = (double)(a.k1.rf_d * a.k1.rf_d * a.k1.rf_d);
#pragma unroll
  for (int q6 = 0; q6 < 6; ++q6) {
    double v0 = wave_sum(dsum[q6][tid]);             // (the column masks of scale 0 went in at every flush)
    double v1 = wave_sum(col1 ? dsum[6 + q6][tid] : 0.0);
    if (q6 >= 3) { v0 *= q6 == 5 ? rf0_d : rf0_hv; v1 *= q6 == 5 ? rf1_d : rf1_hv; }
    if (lane == 0) { part0[q6] = v0; part1[q6] = v1; }
  }
}

#ifndef PQA_ADM_PYRAMID_OCC
#define PQA_ADM_PYRAMID_OCC 3
#endif
template <typename T>
__global__ __launch_bounds__(kBlock, PQA_ADM_PYRAMID_OCC) void adm_pyramid_kernel(const PyramidArgs a) {
  __shared__ double dsum[12][kBlock];   // per-thread double sums: scale 0 (6), scale 1 (6); no thread reads another's
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Which (frame, segment, stripe group) a workgroup takes.  NOT xcd_remap: that gives every XCD one contiguous range of ids
  // -- the same rows of every frame -- and the rows differ 3 x in cost (inside / outside the crop window): the XCDs that
  // drew the middle of the frame worked while the others idled.  Consecutive ids go to the XCDs round-robin, which deals
  // every XCD the same mix.  And the ORDER is longest-first: a wave lives ~75 us and a launch is under three rounds of
  // them, so whatever is dispatched last decides the tail -- the segments in the middle of the frame (full chain) of ALL
  // frames go first, those at its top and bottom (approximation band only, a third of the cost) last.
#ifdef PQA_ADM_PYRAMID_PLAIN_ORDER
  const int sg = blockIdx.x % a.n_sg, seg = (blockIdx.x / a.n_sg) % a.n_seg, fr = blockIdx.x / (a.n_sg * a.n_seg);
#else
  const int sg = blockIdx.x % a.n_sg;
  const int t = blockIdx.x / a.n_sg;
  const int fr = t % a.n_frames, k = t / a.n_frames;                         // k: rank of the segment, centre-out
  const int seg = (a.n_seg - 1) / 2 + ((k & 1) ? (k + 1) / 2 : -(k / 2));
#endif
  const int id = seg * a.n_sg + sg;
  const int stripe = sg * 4 + wave;
  double* __restrict__ part0 = a.part0 + ((int64_t)fr * a.n_part + (int64_t)id * 4 + wave) * 6;
  double* __restrict__ part1 = a.part1 + ((int64_t)fr * a.n_part + (int64_t)id * 4 + wave) * 6;
  if (stripe >= a.n_stripes) {   // idle wave of the last group
    if (lane < 6) { part0[lane] = 0.0; part1[lane] = 0.0; }
    return;
  }
  const int R0 = seg * a.seg_rows, R1 = min(R0 + a.seg_rows, a.oh1);
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const unsigned bytes_r = (unsigned)a.h * a.pitch_r * (unsigned)sizeof(T), bytes_d = (unsigned)a.h * a.pitch_d * (unsigned)sizeof(T);
  const int cs1 = stripe * kP1, c1 = cs1 - kPH + lane;
  // every input column of the stripe (all 64 lanes x 4) inside the image and rows aligned for four-sample loads?
  const bool fast = a.aligned && cs1 >= kPH && 4 * (cs1 + kP1 + kPH - 1) + 3 < a.w;   // wave-uniform
  if (fast) {
    const Loader4<T, false> ld{make_rsrc(ref, bytes_r), make_rsrc(dis, bytes_d), {(unsigned)(4 * c1), 0u, 0u, 0u}, a.pitch_r, a.pitch_d, a.h, a.inv_scale};
    pyramid_march<T, false>(a, ld, lane, stripe, R0, R1, fr, part0, part1, dsum);
  } else {
    const Loader4<T, true> ld{make_rsrc(ref, bytes_r), make_rsrc(dis, bytes_d),
                              {(unsigned)mirror1(4 * c1, a.w), (unsigned)mirror1(4 * c1 + 1, a.w), (unsigned)mirror1(4 * c1 + 2, a.w), (unsigned)mirror1(4 * c1 + 3, a.w)},
                              a.pitch_r, a.pitch_d, a.h, a.inv_scale};
    pyramid_march<T, true>(a, ld, lane, stripe, R0, R1, fr, part0, part1, dsum);
  }
}

#ifndef PQA_ADM_PYRAMID_SEG_ROWS
#define PQA_ADM_PYRAMID_SEG_ROWS 32   /* scale-1 rows per segment: 2 x 32 + 6 scale-0 rows, 32 + 2 scale-1 rows */
#endif

}  // namespace

bool adm_pyramid_takes(Elem elem, int w, int h) {
  return (elem == ELEM_U8 || elem == ELEM_U16) && w % 4 == 0 && h % 4 == 0 && w >= 128 && h >= 128;
}

int adm_pyramid_partials(int w, int h) {
  const int ow1 = w / 4, oh1 = h / 4;
  const int n_stripes = (ow1 + kP1 - 1) / kP1;
  return ((n_stripes + 3) / 4) * 4 * ((oh1 + PQA_ADM_PYRAMID_SEG_ROWS - 1) / PQA_ADM_PYRAMID_SEG_ROWS);
}

bool launch_adm_pyramid(hipStream_t stream, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h, float inv_scale,
                        float gain_limit, MutPlaneRun ll_ref, MutPlaneRun ll_dis, double* partials0, double* partials1,
                        int* n_partials, hipError_t* err) {
  if (!adm_pyramid_takes(elem, w, h) || !ll_ref.base || !ll_dis.base) return false;
  const int es = elem == ELEM_U8 ? 1 : 2;
  if ((int64_t)ref.row_pitch * h * es >= (1ll << 31) || (int64_t)dis.row_pitch * h * es >= (1ll << 31)) return false;
  PyramidArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.pitch_r = (unsigned)ref.row_pitch; a.pitch_d = (unsigned)dis.row_pitch;
  a.frame_pitch_r = ref.frame_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.ow0 = w / 2; a.oh0 = h / 2; a.ow1 = w / 4; a.oh1 = h / 4;
  if ((int64_t)ll_ref.row_pitch * a.oh1 * 4 >= (1ll << 31) || (int64_t)ll_dis.row_pitch * a.oh1 * 4 >= (1ll << 31)) return false;
  const uintptr_t four = (uintptr_t)(4 * es - 1);
  a.aligned = (((uintptr_t)ref.base | (uintptr_t)dis.base) & four) == 0 &&
              ((ref.row_pitch | dis.row_pitch | ref.frame_pitch | dis.frame_pitch) & 3) == 0;
  a.n_stripes = (a.ow1 + kP1 - 1) / kP1;
  a.n_sg = (a.n_stripes + 3) / 4;
  a.seg_rows = PQA_ADM_PYRAMID_SEG_ROWS;
  a.n_seg = (a.oh1 + a.seg_rows - 1) / a.seg_rows;
  const double border = 0.1;  // ADM_BORDER_FACTOR
  a.left0 = (int)(a.ow0 * border - 0.5); a.top0 = (int)(a.oh0 * border - 0.5);
  a.right0 = a.ow0 - a.left0; a.bottom0 = a.oh0 - a.top0;
  a.left1 = (int)(a.ow1 * border - 0.5); a.top1 = (int)(a.oh1 * border - 0.5);
  a.right1 = a.ow1 - a.left1; a.bottom1 = a.oh1 - a.top1;
  if (a.top0 < 2 || a.top1 < 2) return false;   // the rows beside the band's first / last row must not reach a threshold
  a.inv_scale = inv_scale;
  for (int s = 0; s < 2; ++s) {
    Consts& k = s ? a.k1 : a.k0;
    k.gain_limit = gain_limit;
    k.rf_hv = 1.0f / adm_dwt_quant_step(s, 1);
    k.rf_d = 1.0f / adm_dwt_quant_step(s, 2);
    k.k_hv = k.rf_hv / 30.0f;
    k.k_d = k.rf_d / 30.0f;
  }
  a.ll_ref = (float*)ll_ref.base; a.ll_dis = (float*)ll_dis.base;
  a.ll_pitch_r = (unsigned)ll_ref.row_pitch; a.ll_pitch_d = (unsigned)ll_dis.row_pitch;
  a.ll_frame_pitch_r = ll_ref.frame_pitch; a.ll_frame_pitch_d = ll_dis.frame_pitch;
  a.part0 = partials0; a.part1 = partials1;
  a.n_part = a.n_sg * 4 * a.n_seg;
  if (n_partials) *n_partials = a.n_part;
  a.n_frames = n_frames;
  const dim3 grid((unsigned)a.n_sg * a.n_seg * n_frames), block(kBlock);
  if (elem == ELEM_U16) hipLaunchKernelGGL((adm_pyramid_kernel<uint16_t>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((adm_pyramid_kernel<uint8_t>), grid, block, 0, stream, a);
  *err = hipGetLastError();
  return true;
}

}  // namespace pqa
