// VIF (4-scale separable-Gaussian pyramid + local mean/variance statistic) for gfx950.
//
// Arithmetic follows libvmaf's float extractor (vif.c compute_vif, vif_tools.c vif_filter1d_s /
// vif_filter1d_sq_s / vif_filter1d_xy_s / vif_dec2_s / vif_statistic_s) -- the code behind the
// reference's `libvmaf=` call site, app/vmaf_analyzer.py:373-419 -- restated in oracle/vmaf_oracle.c.
//
// Kernel shape (one workgroup = 4 waves = one TW x 8 output tile of one frame, TW = 240/248/252):
//   1. vertical pass: thread <-> column (TW + N - 1 <= 256 columns).  Each thread streams its column's
//      8 + N - 1 input rows with buffer loads whose row offset is an SGPR (coalesced row segments, zero
//      VALU address math), forms r, d, r*r, d*d, r*d and accumulates 4 row PAIRS x 5 signals with
//      v_pk_fma_f32: acc{2p,2p+1} += {c[k], c[k-1]} * {x, x} (tap pair from SGPRs, input broadcast), in
//      libvmaf's tap order.  The next scale's input (filter with the next kernel, keep even samples) is
//      accumulated from the same rows.  Results go to LDS as float2 = {row 2p, row 2p+1}.
//   2. horizontal pass: lane <-> (row pair, 4-column segment); ds_read_b128 of input pairs, conflict-free
//      by construction (see the lane map below); out{2p,2p+1}[o] += c[k] * in{2p,2p+1}[o+k].
//   3. statistic on the pair, DPP wave sum, one (num, den) double partial per tile; a fixed-order second
//      stage (finalize.hip) keeps a frame's record independent of batch size and rank count.
// HBM traffic: the two input planes once (halo re-reads hit L2) + the half-resolution planes written once.
//
// Which kernel runs where (launch_vif_stat at the bottom): scales 1-3 and scale 0 of 12-bit clips: vif_stat_kernel (VALU);
// scale 0 of 8- and 10-bit clips: vif_s0_march_kernel in vif_march.hip (both filter passes on the f16 matrix cores) -- the
// kernel below that puts only the vertical pass there, vif_s0_mfma_kernel (round 2), remains as its A/B partner
// (PQA_VIF_MFMA=2) and as the fallback when the march kernel cannot take the caller's next-scale planes.
#include <cstring>
#include <mutex>
#include <vector>

#include <cstdio>
#include "kernels.h"
#include "pqa_device.h"

namespace pqa {

namespace {

struct Taps {
  float f[17];
};

static Taps gaussian_taps(int n) {
  // N taps, sigma = N/5, normalised in double, stored as float (libvmaf vif_filter1d_table).
  Taps t{};
  double v[17], sum = 0.0;
  const double sigma = n / 5.0;
  for (int k = 0; k < n; ++k) {
    const double d = k - n / 2;
    v[k] = exp(-0.5 * d * d / (sigma * sigma));
    sum += v[k];
  }
  for (int k = 0; k < n; ++k) t.f[k] = (float)(v[k] / sum);
  return t;
}

// Tap table of the packed vertical pass: ct[k] = {c[k], c[k-1]} with c[-1] = c[N] = 0, so one
// v_pk_fma_f32 with the input broadcast updates the two rows of a row pair in libvmaf's tap order.
struct TapPairs {
  f2 vt[18];  // vertical:   {c[k], c[k-1]}
  f2 ht[17];  // horizontal: {c[k], c[k]}  (ready-made SGPR pairs: no splat moves, no op_sel juggling)
  f2 dt[9];   // next scale's taps {c'[k], c'[k]} for the fused decimation
};

static TapPairs tap_pairs(const Taps& t, int n, const Taps* next, int n_next) {
  TapPairs p{};
  if (next)
    for (int k = 0; k < n_next; ++k) p.dt[k] = f2{next->f[k], next->f[k]};
  for (int k = 0; k <= n; ++k) {
    p.vt[k].x = k < n ? t.f[k] : 0.0f;
    p.vt[k].y = k > 0 ? t.f[k - 1] : 0.0f;
  }
  for (int k = 0; k < n; ++k) p.ht[k] = f2{t.f[k], t.f[k]};
  return p;
}

struct VifStatArgs {
  const void* ref;
  const void* dis;
  int64_t row_pitch_r, frame_pitch_r, row_pitch_d, frame_pitch_d;
  int w, h, tiles_x, n_tiles;
  int fold_w, fold_h;  // high-edge fold points of the border rule (pqa_device.h mirror_fold)
  float inv_scale, gain_limit;
  double* partials;
  // fused decimation (scale s -> s+1): filtered with the NEXT scale's taps, even samples kept
  float* dst_ref;
  float* dst_dis;
  int64_t dst_row_pitch_r, dst_frame_pitch_r, dst_row_pitch_d, dst_frame_pitch_d;
  // Scale 0 of 8-bit clips splits the tile grid between kernels.  A launch either covers the rectangle
  // [tx_lo, tx_hi) x [ty_lo, ty_hi) of a grid of tiles_x x grid_rows units or (EDGE variants) everything outside it.
  // Units are tiles for vif_stat_kernel and vertically adjacent tile PAIRS (2p, 2p+1) for vif_s0_mfma_kernel.
  int tx_lo, tx_hi, ty_lo, ty_hi, grid_rows;
  const uint4* atab;   // MFMA kernel: per-lane tap-matrix fragments (kAtabFrags x 64 lanes x 8 f16)
  int extra_ty;           // >= 0: the last border unit is the pair (extra_ty, extra_ty + 1) of which only the LOWER tile
                          // is new (odd number of tile rows); its upper tile is skipped
  TapPairs taps;
};

// Border unit number e (0 .. n_border) -> unit id (row * tiles_x + column) in the full grid: the rows above and below
// the rectangle in full, then the left / right flanks of the rows beside it.
__device__ __forceinline__ int border_tile(int e, const VifStatArgs& a) {
  const int W = a.tiles_x, top = a.ty_lo * W;
  if (e < top) return e;
  e -= top;
  const int bottom = (a.grid_rows - a.ty_hi) * W;
  if (e < bottom) return a.ty_hi * W + e;
  e -= bottom;
  const int side = W - (a.tx_hi - a.tx_lo);
  const int r = e / side, i = e - r * side;
  return (a.ty_lo + r) * W + (i < a.tx_lo ? i : a.tx_hi + (i - a.tx_lo));
}

constexpr int kP2 = 258;  // LDS row pitch in float2: == 2 (mod 32) -> conflict-free ds_read_b128 (see below)

// FP32 FMA throughput on gfx950 is 64 FLOP/clk/SIMD however it is issued (v_pk_fma_f32 every 4 clocks, plain
// v_fma_f32 every 2 with >= 2 waves per SIMD, f32 MFMA: tools/ubench/pk_vs_plain.hip, mfma_coissue.hip); the packed
// form halves the instruction count.  Everything hot is arranged as float2 = {row 2p, row 2p+1} of one column:
//   vertical pass:   acc{2p,2p+1} += {c[k], c[k-1]} * {x, x}     (tap PAIR from SGPRs, input broadcast)
//   horizontal pass: out{2p,2p+1}[o] += c[k] * in{2p,2p+1}[o+k]  (tap broadcast, input pair from LDS)

// Phases 1b (fused decimation), 2 (horizontal pass) and 3 (statistic + tile partial) of one TW x 8 tile whose
// vertical-pass results are in LDS: sv[signal][row pair][column] = {row 2p, row 2p+1}, sd[even row][column] =
// {ref, dis}.  Shared by the VALU kernel (vif_stat_kernel) and the matrix-core kernel (vif_s0_mfma_kernel).
// FULL: every pixel of the tile lies inside the image (interior tile pairs of the matrix-core kernel): the validity
// masks of edge tiles drop out.
template <int N, int TW, int ND, bool FULL = false>
__device__ __forceinline__ void vif_hstat(const VifStatArgs& a, const f2* sv /* [5][TH/2][kP2] */,
                                          const f2* sd /* [TH/2][kP2] */, double* red, int fr, int tile, int x0, int y0) {
  constexpr int R = N / 2, TH = kVifTileH, NSEG = TW / 4, NRP = TH / 2;
  constexpr int RD = ND / 2;
  const int tid = threadIdx.x;
  fr = __builtin_amdgcn_readfirstlane(fr); tile = __builtin_amdgcn_readfirstlane(tile);   // workgroup-uniform: keep
  x0 = __builtin_amdgcn_readfirstlane(x0); y0 = __builtin_amdgcn_readfirstlane(y0);       // the address math scalar
  // ---- 1b. fused decimation: horizontal pass at even columns, written straight to the next scale --
  if (ND) {
    const int ox0 = x0 >> 1, oy0 = y0 >> 1, ow = a.w >> 1, oh = a.h >> 1;
    // the tile's first output sample as a scalar address, the thread's place inside the tile as a 32-bit offset
    // in BYTES (global_store with an SGPR base and a 32-bit VGPR offset: no 64-bit VALU arithmetic; a tile spans 4 rows)
    float* __restrict__ tr = a.dst_ref + ((int64_t)fr * a.dst_frame_pitch_r + (int64_t)oy0 * a.dst_row_pitch_r + ox0);
    float* __restrict__ td = a.dst_dis + ((int64_t)fr * a.dst_frame_pitch_d + (int64_t)oy0 * a.dst_row_pitch_d + ox0);
    const unsigned rp_r = (unsigned)a.dst_row_pitch_r, rp_d = (unsigned)a.dst_row_pitch_d;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const int item = tid + round * kBlock;
      const int oc = item & 127, orow = item >> 7;  // 128 slots per row, TW/2 of them used
      if (oc < TW / 2 && (FULL || (ox0 + oc < ow && oy0 + orow < oh))) {
        f2 acc = f2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < ND; ++k) acc = __builtin_elementwise_fma(a.taps.dt[k], sd[orow * kP2 + 2 * oc + (R - RD) + k], acc);
        *reinterpret_cast<float*>(reinterpret_cast<char*>(tr) + ((unsigned)orow * rp_r + (unsigned)oc) * 4u) = acc.x;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(td) + ((unsigned)orow * rp_d + (unsigned)oc) * 4u) = acc.y;
      }
    }
  }

  // ---- 2. horizontal pass + 3. statistic --------------------------------------------------------
  // lane l: row pair rp = l & 3, segment (4 columns) seg = 2*(s&3) + (wave&1) + 8*(s>>2) + 32*(wave>>1)
  // with s = l >> 2.  The row-pair pitch is 129 16-byte chunks (== 1 mod 16) and 2*seg == 4*s + const
  // (mod 16), so a lane's chunk index is l + const (mod 16): consecutive lanes hit consecutive chunks,
  // which is conflict-free for every ds_read_b128 lane group.
  const int wave = tid >> 6, lane = tid & 63;
  const int rp = lane & 3, sl = lane >> 2;
  const int seg = 2 * (sl & 3) + (wave & 1) + 8 * (sl >> 2) + 32 * (wave >> 1);
  float num = 0.0f, den = 0.0f;
  if (seg < NSEG) {
    constexpr int NCOL = 4 + N - 1, NREAD = (NCOL + 1) / 2;
    f2 out[5][4];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      f2 in[2 * NREAD];
      const f4* p = reinterpret_cast<const f4*>(&sv[(s * NRP + rp) * kP2 + seg * 4]);
#pragma unroll
      for (int q = 0; q < NREAD; ++q) {
        const f4 v = p[q];
        in[2 * q] = f2{v.x, v.y};
        in[2 * q + 1] = f2{v.z, v.w};
      }
      // tap-major order: the four outputs are independent chains, so consecutive v_pk_fma_f32 never
      // depend on each other (a dependent pair costs an s_nop); each output still sums taps 0..N-1 in order
#pragma unroll
      for (int o = 0; o < 4; ++o) out[s][o] = f2{0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int o = 0; o < 4; ++o) out[s][o] = __builtin_elementwise_fma(a.taps.ht[k], in[o + k], out[s][o]);
      __builtin_amdgcn_sched_barrier(0);  // one signal's 10 ds_read_b128 in flight at a time, not all 50
    }
    const float sigma_nsq = 2.0f, eps = 1.0e-10f, sigma_max_inv = 4.0f / (255.0f * 255.0f);
    // validity as 0/1 weights folded into the accumulation (an fma instead of an add): no branches, and
    // out-of-image positions of edge tiles still hold finite values (mirrored real pixels)
    const int gyA = y0 + 2 * rp;
    const bool vrow[2] = {FULL || gyA < a.h, FULL || gyA + 1 < a.h};
    f2 num2 = f2{0.0f, 0.0f}, den2 = f2{0.0f, 0.0f};
    // the log terms of the sigma1_sq >= sigma_nsq branch are summed as the log of a product: per row of
    // the pair the four columns' arguments (each in [2, 2^16]) are multiplied first, so the thread takes
    // 6 v_log_f32 instead of 16 and needs no reciprocal for num's ratio (log a/b = log a - log b)
    f2 pn = f2{1.0f, 1.0f}, qn = f2{1.0f, 1.0f}, pd = f2{1.0f, 1.0f};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const bool vcol = FULL || (x0 + seg * 4 + o) < a.w;
      // the two rows of the pair go through the statistic together: every add / mul / fma is packed,
      // only max / min / select / rcp / log are per element
      const f2 mu1 = out[0][o], mu2 = out[1][o];
      const f2 s1 = out[2][o] - mu1 * mu1;
      f2 s2 = out[3][o] - mu2 * mu2;
      const f2 s12 = out[4][o] - mu1 * mu2;
      s2 = f2{fmaxf(s2.x, 0.0f), fmaxf(s2.y, 0.0f)};
      // validity and the branch choice: out-of-image positions of edge tiles hold finite values (mirrored real
      // pixels); they and the positions of the low branch contribute a factor 1 to the log products and a weight 0 / 1
      // to the low sums.  ONE select does it for all three products: with sigma1_sq replaced by 0 the arguments
      // become narg = svn (cancels against qn's svn) and darg = 1.  (A select costs the VALU as much as a packed FMA.)
      // libvmaf's MAX(sigma1_sq, 0) needs no instruction either: the log branch has sigma1_sq >= 2, everything else
      // continues with 0.
      const bool vx = vcol && vrow[0], vy = vcol && vrow[1];
      const bool hx = vx && !(s1.x < sigma_nsq), hy = vy && !(s1.y < sigma_nsq);
      const bool lx = vx && (s1.x < sigma_nsq), ly = vy && (s1.y < sigma_nsq);
      const f2 s1h = f2{hx ? s1.x : 0.0f, hy ? s1.y : 0.0f};
      // g = sigma12 / (sigma1_sq + eps): v_rcp_f32 plus one Newton correction -- exact 1.0 when the two
      // are equal (identical frames), 2 FMAs instead of a full IEEE division.  (Off the log branch g is sigma12 / eps:
      // large but finite, clamped below, and multiplied by s1h = 0.)
      const f2 gden = s1h + f2{eps, eps};
      const f2 grcp = f2{fast_rcp(gden.x), fast_rcp(gden.y)};
      f2 g = s12 * grcp;
      g = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, gden, s12), grcp, g);
      f2 sv = s2 - g * s12;
      // vif_statistic_s also has `if (sigma1_sq < eps) {g = 0; sv_sq = sigma2_sq; sigma1_sq = 0}` and
      // `if (g < 0) {sv_sq = sigma2_sq; g = 0}`.  Both are dead for the result: the first implies
      // sigma1_sq < sigma_nsq and the second implies sigma12 < 0, and each of those overrides num/den below.
      // Its third override, `if (sigma2_sq < eps) {g = 0; sv_sq = 0}`, changes nothing measurable either: with
      // sigma2_sq < 1e-10 Cauchy-Schwarz bounds |sigma12| <= sqrt(sigma1_sq * 1e-10), so g^2 sigma1_sq <= 1e-10 next to
      // sv_sq + 2 in the log's argument, and sv_sq = sigma2_sq - g sigma12 lies within 1e-10 of 0 and is raised to eps
      // by the max below in both forms.  (Round 1 kept it as two compares and four selects per pixel pair.)
      sv = f2{fmaxf(sv.x, eps), fmaxf(sv.y, eps)};
      // clamp g to [0, gain_limit] with one v_med3_f32: the upper bound is vif_enhn_gain_limit, the lower
      // bound makes g*g = 0 when sigma12 < 0, i.e. num_val = log2(1) = 0 -- libvmaf's `if (sigma12 < 0) num_val = 0`
      g = f2{__builtin_amdgcn_fmed3f(g.x, 0.0f, a.gain_limit), __builtin_amdgcn_fmed3f(g.y, 0.0f, a.gain_limit)};
      const f2 svn = sv + f2{sigma_nsq, sigma_nsq};
      // num_val = log2(1 + g^2 sigma1_sq / (sv_sq + sigma_nsq)) = log2(narg) - log2(svn)
      const f2 narg = __builtin_elementwise_fma(g * g, s1h, svn);
      const f2 darg = __builtin_elementwise_fma(s1h, f2{1.0f / sigma_nsq, 1.0f / sigma_nsq}, f2{1.0f, 1.0f});
      const f2 low = __builtin_elementwise_fma(s2, f2{-sigma_max_inv, -sigma_max_inv}, f2{1.0f, 1.0f});
      pn *= narg;
      qn *= svn;
      pd *= darg;
      const f2 wl = f2{lx ? 1.0f : 0.0f, ly ? 1.0f : 0.0f};
      num2 = __builtin_elementwise_fma(wl, low, num2);
      den2 += wl;
      __builtin_amdgcn_sched_barrier(0);  // two pixels' worth of temporaries live at a time
    }
    num = (num2.x + num2.y) + ((fast_log2(pn.x) - fast_log2(qn.x)) + (fast_log2(pn.y) - fast_log2(qn.y)));
    den = (den2.x + den2.y) + (fast_log2(pd.x) + fast_log2(pd.y));
  }
  {
    const float part[2] = {num, den};
    double v[2];
    block_sum_f32<2>(part, v, red);
    if (tid == 0) {
      double* out = a.partials + ((int64_t)fr * a.n_tiles + tile) * 2;
      out[0] = v[0];
      out[1] = v[1];
    }
  }
}

template <typename T, int N, int TW, int ND, bool EDGE = false>
__global__ __launch_bounds__(kBlock, 3) void vif_stat_kernel(const VifStatArgs a) {
  constexpr int R = N / 2, TH = kVifTileH, COLS = TW + N - 1, NSEG = TW / 4, S = 8, NIN = S + N - 1;
  constexpr int RD = ND / 2;  // ND = taps of the next scale's filter (0: last scale, no decimation)
  static_assert(ND == 0 || RD <= R, "decimation window must sit inside the statistic window");
  static_assert(COLS <= 256 && COLS <= kP2, "one column per thread");
  static_assert(TW % 4 == 0 && NSEG <= 64 && TH == S, "segment map");
  __shared__ f2 sv[5][TH / 2][kP2];  // [signal][row pair][column] = {row 2p, row 2p+1}
  __shared__ f2 sd[ND ? TH / 2 : 1][ND ? kP2 : 1];  // decimation: [even row][column] = {ref, dis}
  __shared__ double red[12];

  const int tile = EDGE ? border_tile(blockIdx.x, a) : xcd_remap(blockIdx.x, a.n_tiles);
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x;

  // ---- 1. vertical pass: lane <-> column, wave-uniform row addressing ---------------------------
  {
    const int col = tid;   // one column per thread; the 8-row strip is the whole tile: rows are uniform
    constexpr int seg = 0;
    if (col < COLS) {
      const unsigned gx = (unsigned)mirror_fold(x0 - R + col, a.w, a.fold_w);
      const unsigned pitch_r = (unsigned)a.row_pitch_r, pitch_d = (unsigned)a.row_pitch_d;  // planes < 4 G samples
      const auto rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r * (unsigned)sizeof(T));
      const auto rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d * (unsigned)sizeof(T));
      f2 acc[S / 2][5];
#pragma unroll
      for (int p = 0; p < S / 2; ++p)
#pragma unroll
        for (int s = 0; s < 5; ++s) acc[p][s] = f2{0.0f, 0.0f};
      f2 dacc[S / 2];  // next-scale input at the strip's even rows, {ref, dis}
#pragma unroll
      for (int q = 0; q < S / 2; ++q) dacc[q] = f2{0.0f, 0.0f};
      // Input rows stream through in groups of G: the loads of group g+1 are in flight while group g is
      // consumed; sched_barrier keeps hipcc from hoisting all 2*NIN loads (and their r*r, d*d, r*d) to
      // the top, which costs ~120 VGPRs and a wave of occupancy.
#ifndef PQA_VIF_G
#define PQA_VIF_G 12   /* rows per load group: 2 groups at N = 17 (swept 4/6/8/12 on the box) */
#endif
      constexpr int G = PQA_VIF_G, NG = (NIN + G - 1) / G;
      T rn[G], dn[G];
      auto issue = [&](int g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int j = g * G + i;
          if (j < NIN) {
            const unsigned gy = (unsigned)mirror_fold(y0 + seg * S - R + j, a.h, a.fold_h);
            // scalar frame base + 32-bit (row * pitch + column) lane offset -> global_load with an SGPR base
            // buffer_load: lane offset (column) in a VGPR, row offset in an SGPR -> no per-load VALU
            rn[i] = buf_load<T>(rsrc_r, gx, gy * pitch_r);
            dn[i] = buf_load<T>(rsrc_d, gx, gy * pitch_d);
          }
        }
      };
      issue(0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        T rc[G], dc[G];
#pragma unroll
        for (int i = 0; i < G; ++i) { rc[i] = rn[i]; dc[i] = dn[i]; }
        if (g + 1 < NG) issue(g + 1);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int j = g * G + i;
          if (j < NIN) {
            const f2 x = PixIO<T>::pair(rc[i], dc[i], a.inv_scale);  // {r, d} - 128 (one v_pk_add / v_pk_fma)
            const f2 xx = x * x;                                      // {r*r, d*d}   (one v_pk_mul)
            const float r = x.x, d = x.y, rr = xx.x, dd = xx.y, rd = x.x * x.y;
            if (ND) {
#pragma unroll
              for (int q = 0; q < S / 2; ++q) {
                const int kd = j - (2 * q + R - RD);  // even row 2q, tap kd of the next scale's filter
                if (kd >= 0 && kd < ND) dacc[q] = __builtin_elementwise_fma(a.taps.dt[kd], x, dacc[q]);
              }
            }
#pragma unroll
            for (int p = 0; p < S / 2; ++p) {
              const int k = j - 2 * p;  // row 2p takes tap k, row 2p+1 tap k-1: both sum taps in increasing order
              if (k >= 0 && k <= N) {
                const f2 c = a.taps.vt[k];
                acc[p][0] = __builtin_elementwise_fma(c, f2{r, r}, acc[p][0]);
                acc[p][1] = __builtin_elementwise_fma(c, f2{d, d}, acc[p][1]);
                acc[p][2] = __builtin_elementwise_fma(c, f2{rr, rr}, acc[p][2]);
                acc[p][3] = __builtin_elementwise_fma(c, f2{dd, dd}, acc[p][3]);
                acc[p][4] = __builtin_elementwise_fma(c, f2{rd, rd}, acc[p][4]);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int p = 0; p < S / 2; ++p)
#pragma unroll
        for (int s = 0; s < 5; ++s) sv[s][seg * (S / 2) + p][col] = acc[p][s];
      if (ND) {
#pragma unroll
        for (int q = 0; q < S / 2; ++q) sd[seg * (S / 2) + q][col] = dacc[q];
      }
    }
  }
  __syncthreads();

  vif_hstat<N, TW, ND>(a, &sv[0][0][0], &sd[0][0], red, fr, tile, x0, y0);
}


// ================================================================================================================
// Scale 0 of 8-bit and 10-bit clips on the matrix cores: the 17-tap VERTICAL pass as a banded-Toeplitz product with exact inputs.
//
// Why.  The VALU kernel above is bound by VALU issue cycles: FP32 FMA throughput is 64 FLOP/clk/SIMD however it is
// issued (v_pk_fma_f32 every 4 clocks, plain v_fma_f32 every 2, the f32 MFMA forms), FMAs of either kind share their
// pipe with MFMA, and every other VALU instruction (convert, select, max, byte permute, copy) takes the 4 clocks of a
// packed FMA, about half of which an MFMA of another wave can hide (tools/ubench/mfma_coissue.hip, pk_vs_plain.hip,
// mfma_plain_coissue.hip).  Its vertical pass + sample conversion is 42 % of its issue cycles.  The f16 matrix pipe has
// 16x the FMA rate, and at scale 0 the filter INPUTS are small integers, so they can go through it without loss:
//   * the five signals are split into base-256 digit planes of exact integers: r' = r-128, d' = d-128 (|.| <= 128);
//     r'^2, d'^2 in [0, 16384] -> hi, lo with value = 256 hi + lo; r'd' in [-16256, 16384] -> hi = floor(./256) in
//     [-64, 64], lo in [0, 255];
//   * an integer k < 2048 in a 16-bit lane already IS an f16 -- the bit pattern reads as k * 2^-24 (denormals and the
//     first normal binade share one ulp) -- so a digit costs the one byte permute that extracts it and no conversion;
//     signed planes (r', d', the cross term's high digit) take one v_pk_add_f16 of the offset, exact;
//   * every f32 tap c is split into three f16 pieces, exactly (checked on the host): c * 2^19 for the digits that
//     weigh 2^8 (the largest piece is 62 272 < 65 504) and c * 2^11 for the mean planes; the low digit planes use the
//     first two pieces of c * 2^11 (22 bits: their weight is 2^-8 of the signal);
//   * v_mfma_f32_16x16x32_f16: D[16 out rows][16 cols] += A[16][32 input rows] * B[32][16]; f16 x f16 products are
//     exact in f32, the accumulator is f32.  A = the Toeplitz band of tap pieces (per-lane constants from a table),
//     B = a digit plane.  16 output rows need exactly the 32 input rows one instruction holds.
// Every signal comes out as 2^-13 times the f32 convolution with an error below one f32 rounding of libvmaf's own
// tap-by-tap sum; the factor is folded into the horizontal taps (an exact power of two).  The next scale's input
// (9-tap filter, even rows) rides the mean planes' B operands with its own band matrix.
//
// Shape.  One workgroup = two vertically adjacent tiles of the VALU kernel's grid (TW x 16 outputs, 32 input rows).
// Wave w owns input columns 64w..64w+63 in two passes of 32; lane l: n = l & 15 -> columns 2n, 2n+1 (one 16-bit load
// per row: N-blocks b = 0, 1), g = l >> 4 -> input rows 8g..8g+7 (the K group of the B operand).  The output rows
// are permuted in A so that accumulator registers {0,1} of lane group g are rows {2g, 2g+1} of the UPPER tile and
// {2,3} the same rows of the LOWER tile: one ds_write_b64 per signal and N-block stores {row 2p, row 2p+1} of a column
// in the layout vif_hstat reads; the lower tile's half waits in registers while the upper tile runs its
// horizontal pass + statistic, then takes its place in LDS (the LDS holds one 8-row tile: 3 workgroups per CU).
// Only tiles whose 32 x 256 input window lies inside the image take this path (no mirroring: 86 % of the tiles at
// 2160p, 72 % at 1080p); border tiles run vif_stat_kernel<.., EDGE>.  Both write the same per-tile partials.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// fragments in the table: 0-2: c*2^19 pieces (high digits)  3-4, 12: c*2^11 pieces (means; 3-4 also the 8-bit low digits)
// 5-7: c'*2^18 pieces (decimation)  8-9: c*2^9 pieces (10-bit low digits, base 1024)  10-11: unused
constexpr int kAtabFrags = 13;
// an 8-byte LDS store the compiler may not fuse with its neighbour into a 16-byte one (volatile): the two halves come
// from different accumulators, and fusing them means four register copies per store
__device__ __forceinline__ void lds_store_f2(f2* p, f2 v) {
  typedef __attribute__((address_space(3))) volatile f2 lds_f2;
  *(lds_f2*)p = v;
}
__device__ __forceinline__ void lds_store_f32(float* p, float v) {
  typedef __attribute__((address_space(3))) volatile float lds_f32;
  *(lds_f32*)p = v;
}
// v_bfi_b32: bits of `a` where the mask is set, bits of `b` elsewhere
__device__ __forceinline__ unsigned bfi32(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ h8 frag_from(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(h8, u4v{a, b, c, d});
}
// Digits and samples go into the B operand as f16 bit patterns 0x0000 .. 0x07ff, which the f16 format reads as
// k * 2^-24 for every k < 2048 (denormals and the first normal binade share one ulp): an integer in the low bits of a
// 16-bit lane IS its own f16 encoding, no conversion instruction.  gfx950 keeps f16 denormals in v_pk_add_f16 and
// in the MFMA operands (tools/ubench/f16_denorm.hip, profiles/r02j_ubench_f16_denorm.txt); the tests that compare this
// kernel with the VALU kernel on extreme samples would catch a flush.
// two integers k (one per 16-bit half, k < 2048) -> two f16 (k - off) * 2^-24, exact; `off_bits` = off as f16 bits | 0x8000
__device__ __forceinline__ unsigned f16_tiny_minus(unsigned x, unsigned short off_bits) {
  const h2 o = __builtin_bit_cast(h2, (unsigned)off_bits | ((unsigned)off_bits << 16));
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, x) + o);
}

// T = uint8_t: 8-bit samples as described above.  T = uint16_t: 10-bit samples (libvmaf: x = v / 4 - 128 = (v - 512) / 4):
// v - 512 is one exact f16 plane, the squares (<= 2^18) and the cross term split into base-1024 digits (hi <= 256,
// lo < 1024: still below 2048) from 32-bit products; the low planes use pieces of c * 2^9, and with the sample scale
// (1/4 on the means, 1/16 on the squares) every signal comes out as 2^-11 of libvmaf's: again one factor for the
// horizontal taps.  12-bit clips (squares of 22 bits: three digits) stay on the VALU kernel.
template <typename T, bool EDGE>
__global__ __launch_bounds__(kBlock, 3) void vif_s0_mfma_kernel(const VifStatArgs a) {
  constexpr int N = 17, TW = 240, ND = 9, TH = kVifTileH;
  constexpr bool W16 = sizeof(T) == 2;
  constexpr int ES = (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) f2 sv[5][TH / 2][kP2];
  __shared__ __attribute__((aligned(16))) f2 sd[TH / 2][kP2];
  __shared__ double red[12];

  int tx, ty;
  bool skip_upper = false;
  if (EDGE) {   // pairs outside the interior rectangle: rows / columns are mirrored per lane in the loads below
    const int u = border_tile(blockIdx.x, a);
    const int pr = u / a.tiles_x;
    tx = u % a.tiles_x;
    // odd number of tile rows: the last unit row is the pair (extra_ty, extra_ty + 1) whose upper tile already belongs
    // to the pair above it -- only its lower tile is produced here
    skip_upper = a.extra_ty >= 0 && pr == a.grid_rows - 1;
    ty = skip_upper ? a.extra_ty : 2 * pr;
  } else {
    const int n_tx = a.tx_hi - a.tx_lo;
    const int idx = xcd_remap(blockIdx.x, n_tx * (a.ty_hi - a.ty_lo));
    tx = a.tx_lo + idx % n_tx; ty = 2 * (a.ty_lo + idx / n_tx);
  }
  const int fr = blockIdx.y;
  const T* __restrict__ ref = (const T*)a.ref + (int64_t)fr * a.frame_pitch_r;
  const T* __restrict__ dis = (const T*)a.dis + (int64_t)fr * a.frame_pitch_d;
  const int x0 = tx * TW, y0 = ty * TH;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
  const unsigned pitch_r = (unsigned)a.row_pitch_r * ES, pitch_d = (unsigned)a.row_pitch_d * ES;  // bytes
  const auto rsrc_r = make_rsrc(ref, (unsigned)a.h * pitch_r);
  const auto rsrc_d = make_rsrc(dis, (unsigned)a.h * pitch_d);

  // tap-matrix fragments of this lane (A operand: row = lane & 15, K group = lane >> 4)
  // slots 0-2: c * 2^19 (high digits)   3-4: low-digit pieces of this sample type   5-7: c' * 2^18 (decimation)
  // 8..: the pieces of c * 2^11 the mean planes use that slots 3-4 do not already hold
  constexpr int NA = W16 ? 11 : 9;
  constexpr int kSlotFrag[11] = {0, 1, 2, W16 ? 8 : 3, W16 ? 9 : 4, 5, 6, 7, W16 ? 3 : 12, 4, 12};
  constexpr int MU0 = W16 ? 8 : 3, MU1 = W16 ? 9 : 4, MU2 = W16 ? 10 : 8;
  h8 A[NA];
#pragma unroll
  for (int f = 0; f < NA; ++f) A[f] = __builtin_bit_cast(h8, a.atab[kSlotFrag[f] * 64 + lane]);
  f2 park[2][5][2];   // lower tile's rows {2g, 2g+1} of columns 2n, 2n+1 per pass and signal
  f4 park_d[2];    // lower tile's decimation row g: {ref, dis} x 2 columns
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int col0 = 64 * wave + 32 * pass + 2 * n;  // tile column (= LDS column) of N-block 0; N-block 1 is col0 + 1
    unsigned rr_[8], dr_[8];  // rows 8g+j: the two columns (8 bit: in the low 16 bits; 16 bit: low / high half)
    if (!EDGE) {
      const unsigned gx = (unsigned)(x0 - (N / 2) + col0);
      const unsigned gy = (unsigned)(y0 - (N / 2) + 8 * g);
      const unsigned off_r = gy * pitch_r + gx * ES, off_d = gy * pitch_d + gx * ES;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (W16) {
          rr_[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc_r, off_r, (unsigned)j * pitch_r, 0);
          dr_[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc_d, off_d, (unsigned)j * pitch_d, 0);
        } else {
          rr_[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, off_r, (unsigned)j * pitch_r, 0);
          dr_[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, off_d, (unsigned)j * pitch_d, 0);
        }
      }
    } else {
      // the border rule per lane (same folds as the VALU kernel): two loads per row, rows mirrored one by one
      const unsigned gx0 = (unsigned)mirror_fold(x0 - (N / 2) + col0, a.w, a.fold_w) * ES;
      const unsigned gx1 = (unsigned)mirror_fold(x0 - (N / 2) + col0 + 1, a.w, a.fold_w) * ES;
      // most edge pairs sit in the first / last tile COLUMN with all 32 rows inside the image (270 of 298 at 2160p): their
      // rows need no fold -- a scalar offset per row as in the interior kernel, only the two columns are per-lane
      const bool rows_in = y0 - (N / 2) >= 0 && y0 - (N / 2) + 32 <= a.h;   // workgroup-uniform
      if (rows_in) {
        const unsigned gyb = (unsigned)(y0 - (N / 2) + 8 * g);
        const unsigned br0 = gyb * pitch_r + gx0, br1 = gyb * pitch_r + gx1, bd0 = gyb * pitch_d + gx0, bd1 = gyb * pitch_d + gx1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (W16) {
            const unsigned r0 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, br0, (unsigned)j * pitch_r, 0);
            const unsigned r1 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, br1, (unsigned)j * pitch_r, 0);
            const unsigned d0 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, bd0, (unsigned)j * pitch_d, 0);
            const unsigned d1 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, bd1, (unsigned)j * pitch_d, 0);
            rr_[j] = r0 | (r1 << 16);
            dr_[j] = d0 | (d1 << 16);
          } else {
            const unsigned r0 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_r, br0, (unsigned)j * pitch_r, 0) & 0xffu;
            const unsigned r1 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_r, br1, (unsigned)j * pitch_r, 0) & 0xffu;
            const unsigned d0 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_d, bd0, (unsigned)j * pitch_d, 0) & 0xffu;
            const unsigned d1 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_d, bd1, (unsigned)j * pitch_d, 0) & 0xffu;
            rr_[j] = r0 | (r1 << 8);
            dr_[j] = d0 | (d1 << 8);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned gy = (unsigned)mirror_fold(y0 - (N / 2) + 8 * g + j, a.h, a.fold_h);
          if (W16) {
            const unsigned r0 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, gy * pitch_r + gx0, 0, 0);
            const unsigned r1 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_r, gy * pitch_r + gx1, 0, 0);
            const unsigned d0 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, gy * pitch_d + gx0, 0, 0);
            const unsigned d1 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc_d, gy * pitch_d + gx1, 0, 0);
            rr_[j] = r0 | (r1 << 16);
            dr_[j] = d0 | (d1 << 16);
          } else {
            const unsigned r0 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_r, gy * pitch_r + gx0, 0, 0) & 0xffu;
            const unsigned r1 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_r, gy * pitch_r + gx1, 0, 0) & 0xffu;
            const unsigned d0 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_d, gy * pitch_d + gx0, 0, 0) & 0xffu;
            const unsigned d1 = __builtin_amdgcn_raw_buffer_load_b8(rsrc_d, gy * pitch_d + gx1, 0, 0) & 0xffu;
            rr_[j] = r0 | (r1 << 8);
            dr_[j] = d0 | (d1 << 8);
          }
        }
      }
    }
    f4 D[5][2], Dd[2][2];
    // Accumulators start in the FIRST product of each chain (srcC = 0) instead of being cleared one by one.
    const f4 zero4 = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      // 16-bit lanes {row 2v, row 2v+1} of this N-block's column (K order of the B operand: element j = row 8g + j)
      unsigned ru[4], du[4], r16[4], d16[4];
      // 8 bit: {0, byte b of source 0, 0, byte b of source 1};  16 bit: {half b of source 0, half b of source 1}
      const unsigned selb = W16 ? (b ? 0x07060302u : 0x05040100u) : (b ? 0x0c050c01u : 0x0c040c00u);
      constexpr short MID = W16 ? 512 : 128;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        ru[v] = __builtin_amdgcn_perm(rr_[2 * v + 1], rr_[2 * v], selb);
        du[v] = __builtin_amdgcn_perm(dr_[2 * v + 1], dr_[2 * v], selb);
        r16[v] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s2v, ru[v]) - s2v{MID, MID});
        d16[v] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s2v, du[v]) - s2v{MID, MID});
      }
      unsigned t[4];
#define PQA_MMA(Dacc, frag) Dacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[frag], B, Dacc, 0, 0, 0)
#define PQA_MMA0(Dacc, frag, C0) Dacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[frag], B, C0, 0, 0, 0)
      constexpr unsigned short MID_TINY = W16 ? 0x8200 : 0x8080;   // -MID * 2^-24 as f16 bits
      {  // means: r' and d' (sample - mid-grey) * 2^-24, plus the next scale's input from the same operands
#pragma unroll
        for (int v = 0; v < 4; ++v) t[v] = f16_tiny_minus(ru[v], MID_TINY);
        const h8 B = frag_from(t[0], t[1], t[2], t[3]);
        PQA_MMA0(D[0][b], MU0, zero4); PQA_MMA0(Dd[0][b], 5, zero4); PQA_MMA(D[0][b], MU1); PQA_MMA(Dd[0][b], 6); PQA_MMA(D[0][b], MU2); PQA_MMA(Dd[0][b], 7);
      }
      {
#pragma unroll
        for (int v = 0; v < 4; ++v) t[v] = f16_tiny_minus(du[v], MID_TINY);
        const h8 B = frag_from(t[0], t[1], t[2], t[3]);
        PQA_MMA0(D[1][b], MU0, zero4); PQA_MMA0(Dd[1][b], 5, zero4); PQA_MMA(D[1][b], MU1); PQA_MMA(Dd[1][b], 6); PQA_MMA(D[1][b], MU2); PQA_MMA(Dd[1][b], 7);
      }
      // squares and cross term: integer products (exact); their digits are f16 operands as they are (k * 2^-24)
#pragma unroll
      for (int s = 2; s < 5; ++s) {
        if (!W16) {
          unsigned q[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const s2v x = __builtin_bit_cast(s2v, s == 3 ? d16[v] : r16[v]);
            const s2v y = __builtin_bit_cast(s2v, s == 2 ? r16[v] : d16[v]);
            // the cross term is signed: + 64 * 256 makes both digits unsigned (one v_pk_mad_u16), the 64 comes off below
            q[v] = __builtin_bit_cast(unsigned, s == 4 ? (s2v)(x * y + s2v{0x4000, 0x4000}) : (s2v)(x * y));
          }
          {  // low digit: byte 0 of each 16-bit product
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = __builtin_amdgcn_perm(0u, q[v], 0x0c020c00u);
            const h8 B = frag_from(t[0], t[1], t[2], t[3]);
            PQA_MMA0(D[s][b], 3, zero4); PQA_MMA(D[s][b], 4);
          }
          {  // high digit: byte 1
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              t[v] = __builtin_amdgcn_perm(0u, q[v], 0x0c030c01u);
              if (s == 4) t[v] = f16_tiny_minus(t[v], 0x8040);   // - 64 * 2^-24
            }
            const h8 B = frag_from(t[0], t[1], t[2], t[3]);
            PQA_MMA(D[s][b], 0); PQA_MMA(D[s][b], 1); PQA_MMA(D[s][b], 2);
          }
        } else {
          // 10 bit: |v - 512| <= 512, products of 20 bits: 32-bit multiplies on the sign-extended halves; the cross
          // term gets 256 * 1024 added so that its high digit (>> 10, in [0, 512]) is unsigned like the squares'
          unsigned lo_[4], hi_[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const unsigned xs = s == 3 ? d16[v] : r16[v], ys = s == 2 ? r16[v] : d16[v];
            const int x0_ = (int)(short)(xs & 0xffffu), x1_ = (int)xs >> 16;
            const int y0_ = (int)(short)(ys & 0xffffu), y1_ = (int)ys >> 16;
            const int add = s == 4 ? (256 << 10) : 0;
            const unsigned p0 = (unsigned)(x0_ * y0_ + add), p1 = (unsigned)(x1_ * y1_ + add);
            // {lo(p1), lo(p0)} and {hi(p1), hi(p0)} in the 16-bit halves (v_bfi_b32)
            lo_[v] = bfi32(0x03ff0000u, p1 << 16, p0 & 0x3ffu);
            const unsigned hgh = bfi32(0x03ff0000u, p1 << 6, p0 >> 10);
            hi_[v] = s == 4 ? f16_tiny_minus(hgh, 0x8100) : hgh;   // - 256 * 2^-24
          }
          {
            const h8 B = frag_from(lo_[0], lo_[1], lo_[2], lo_[3]);
            PQA_MMA0(D[s][b], 3, zero4); PQA_MMA(D[s][b], 4);
          }
          {
            const h8 B = frag_from(hi_[0], hi_[1], hi_[2], hi_[3]);
            PQA_MMA(D[s][b], 0); PQA_MMA(D[s][b], 1); PQA_MMA(D[s][b], 2);
          }
        }
      }
#undef PQA_MMA
#undef PQA_MMA0
    }
    // upper tile -> LDS (registers 0,1 = rows 2g, 2g+1), lower tile's half parked
#pragma unroll
    // (one 8-byte store per N-block: an accumulator's registers {0,1} are adjacent, the two N-blocks' are not -- a
    // 16-byte store would first copy them together, and a copy costs the VALU as much as a packed FMA)
    for (int s = 0; s < 5; ++s) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        lds_store_f2(&sv[s][g][col0 + b], f2{D[s][b][0], D[s][b][1]});
        park[pass][s][b] = f2{D[s][b][2], D[s][b][3]};
      }
    }
    // (the decimation row's four values sit in four accumulators: four 4-byte stores instead of four copies + one store)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* slot = reinterpret_cast<float*>(&sd[g][col0 + b]);
      lds_store_f32(slot, Dd[0][b][0]);
      lds_store_f32(slot + 1, Dd[1][b][0]);
    }
    park_d[pass] = f4{Dd[0][0][1], Dd[1][0][1], Dd[0][1][1], Dd[1][1][1]};
  }
  if (!skip_upper) {   // workgroup-uniform
    __syncthreads();
    vif_hstat<N, TW, ND, !EDGE>(a, &sv[0][0][0], &sd[0][0], red, fr, ty * a.tiles_x + tx, x0, y0);
    // (vif_hstat ends with a workgroup barrier inside its block sum: every LDS read of the upper tile is done)
  } else {
    __syncthreads();   // the upper tile's LDS stores above must not race with the lower tile's below
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int col0 = 64 * wave + 32 * pass + 2 * n;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      lds_store_f2(&sv[s][g][col0], park[pass][s][0]);
      lds_store_f2(&sv[s][g][col0 + 1], park[pass][s][1]);
    }
    *reinterpret_cast<f4*>(&sd[g][col0]) = park_d[pass];
  }
  __syncthreads();
  vif_hstat<N, TW, ND, !EDGE>(a, &sv[0][0][0], &sd[0][0], red, fr, (ty + 1) * a.tiles_x + tx, x0, y0 + TH);
}

// The operand encoding above leans on f16 denormals surviving v_pk_add_f16 and the MFMA's B operand.  gfx950 keeps them
// (measured), but it is a property of the device and of the kernel's float mode, so every process checks it once before the
// matrix-core kernel is allowed to run: column n of B holds the pattern k = 67 n + 1 (minus 1 via the packed add), A is all
// ones -> D[.][n] = 32 (k - 1) 2^-24 exactly.  ok[0] counts the lanes that saw that.
__global__ void f16_tiny_probe_kernel(int* ok) {
  const int lane = threadIdx.x, n = lane & 15;
  const unsigned k = ((unsigned)n * 67u + 1u) & 0x3ffu;
  const unsigned t = f16_tiny_minus(k | (k << 16), 0x8001);   // (k - 1) * 2^-24 in both halves
  const h8 b = frag_from(t, t, t, t);
  h8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
  const f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
  const float want = 32.0f * (float)(k - 1u) * 5.9604644775390625e-08f;   // 2^-24
  if (d[0] == want && d[3] == want) atomicAdd(ok, 1);
}

// Host: the per-lane A fragments.  Row m = lane & 15 = 4 gg + i of the product is output row 8 (i >> 1) + 2 gg + (i & 1)
// of the 16-row tile pair (see the kernel comment); K index k = 8 (lane >> 4) + j is input row k - 8 relative to the
// pair's first row.  The decimation band puts even output row 2 gg of the upper (i = 0) / lower (i = 1) tile in
// registers 0 / 1 and leaves 2, 3 empty.
static bool build_atab(uint16_t* out /* [kAtabFrags][64][8] */) {
  const Taps c17 = gaussian_taps(17), c9 = gaussian_taps(9);
  bool exact = true;
  for (int lane = 0; lane < 64; ++lane) {
    const int m = lane & 15, gg = m >> 2, i = m & 3, kg = lane >> 4;
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * kg + j;
      const int row = 8 * (i >> 1) + 2 * gg + (i & 1);
      const int t17 = k - row;
      const double c = (t17 >= 0 && t17 <= 16) ? (double)c17.f[t17] : 0.0;
      const int t9 = k - (8 * i + 2 * gg + 4);
      const double cd = (i < 2 && t9 >= 0 && t9 <= 8) ? (double)c9.f[t9] : 0.0;
      const auto pieces = [&](double x, int n_pieces, int first_frag) {
        double r = x;
        for (int p = 0; p < n_pieces; ++p) {
          const _Float16 h = (_Float16)r;
          uint16_t bits;
          memcpy(&bits, &h, 2);
          out[((size_t)(first_frag + p) * 64 + lane) * 8 + j] = bits;
          r -= (double)h;
        }
        return r;
      };
      if (pieces(c * 524288.0, 3, 0) != 0.0) exact = false;
      {  // c * 2^11: three pieces, exact (the last one may be an f16 denormal); the low digit planes use the first two
        double r = pieces(c * 2048.0, 2, 3);
        const _Float16 h = (_Float16)r;
        uint16_t bits;
        memcpy(&bits, &h, 2);
        out[((size_t)12 * 64 + lane) * 8 + j] = bits;
        if (r - (double)h != 0.0) exact = false;
      }
      if (pieces(cd * 262144.0, 3, 5) != 0.0) exact = false;
      pieces(c * 512.0, 2, 8);
      pieces(0.0, 2, 10);
    }
  }
  return exact;
}

// One fragment table per device, uploaded by vif_mfma_prepare() (called from pqa_create: the launchers themselves never
// allocate, so they stay capturable in a hipGraph).  Without a table the launcher keeps the VALU kernel.
static std::mutex g_atab_mu;
static const uint4* g_atab[64] = {};

static const uint4* device_atab() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_atab_mu);
  return g_atab[dev];
}

template <int N, int TW, int ND>
hipError_t launch_stat_n(hipStream_t stream, Elem elem, const VifStatArgs& a, int n_frames) {
  const dim3 grid(a.n_tiles, n_frames), block(kBlock);
  switch (elem) {
    case ELEM_U8: hipLaunchKernelGGL((vif_stat_kernel<uint8_t, N, TW, ND>), grid, block, 0, stream, a); break;
    case ELEM_U16: hipLaunchKernelGGL((vif_stat_kernel<uint16_t, N, TW, ND>), grid, block, 0, stream, a); break;
    case ELEM_F32: hipLaunchKernelGGL((vif_stat_kernel<float, N, TW, ND>), grid, block, 0, stream, a); break;
  }
  return hipGetLastError();
}

// Scale 0 of 8- and 10-bit clips, round-2 kernel: tile pairs through vif_s0_mfma_kernel (vertical pass on the matrix cores,
// horizontal pass on the VALU), pairs on an image edge through its EDGE variant.  Returns false when nothing was launched
// (no table, pitches / bases the two-column loads cannot take): the caller then runs the VALU kernel on every tile.
bool launch_s0_split(hipStream_t stream, bool ten_bit, const VifStatArgs& base, int n_frames, hipError_t* err) {
  constexpr int TW = 240, TH = kVifTileH;
  const int tiles_y = base.n_tiles / base.tiles_x, pair_rows = tiles_y / 2;
  if (pair_rows == 0) return false;
  // the interior kernel loads two columns at a time (16 bits of an 8-bit plane, 32 bits of a 10-bit plane): even element
  // pitches, bases aligned to that (any plane the library packs itself; other caller layouts fall back to the VALU kernel)
  const uintptr_t amask = ten_bit ? 3 : 1;
  if ((base.row_pitch_r | base.row_pitch_d | base.frame_pitch_r | base.frame_pitch_d) & 1) return false;
  if (((uintptr_t)base.ref | (uintptr_t)base.dis) & amask) return false;
  VifStatArgs m = base;
  m.atab = device_atab();
  if (!m.atab) return false;
  // the vertical pass leaves every signal of an 8-bit clip times 2^-13 (operands k * 2^-24, tap pieces c * 2^11, or
  // c * 2^19 on the digits that weigh 2^8) and the next scale's input times 2^-6; for 10-bit clips (x = (v - 512) / 4,
  // digits base 1024) the same operands give 2^-11 and 2^-4 of the scaled signals.  Exact powers of two, folded into
  // the horizontal taps.
  const float sq = ten_bit ? 2048.0f : 8192.0f, dec = ten_bit ? 16.0f : 64.0f;
  for (int k = 0; k < 17; ++k) m.taps.ht[k] = f2{base.taps.ht[k].x * sq, base.taps.ht[k].y * sq};
  for (int k = 0; k < 9; ++k) m.taps.dt[k] = f2{base.taps.dt[k].x * dec, base.taps.dt[k].y * dec};
  // Pair p = tiles (2p, 2p+1), rows 16p .. 16p+15; it is INTERIOR when its 32 x 256 input window (rows 16p-8 .. 16p+23,
  // columns 240tx-8 .. 240tx+247) lies inside the image: those pairs load two columns at a time with scalar row offsets.
  // An odd last tile row is produced by one more (edge) pair that starts one tile higher and skips its upper tile.
  m.extra_ty = (tiles_y & 1) ? tiles_y - 2 : -1;
  m.grid_rows = pair_rows + (tiles_y & 1);
  m.tx_lo = 1;
  m.tx_hi = base.w >= 2 * TW + 8 ? (base.w - (TW + 8)) / TW + 1 : 1;
  m.ty_lo = 1;
  m.ty_hi = base.h >= 5 * TH ? (base.h - 3 * TH) / (2 * TH) + 1 : 1;
  if (m.ty_hi > pair_rows) m.ty_hi = pair_rows;
  if (m.tx_hi <= m.tx_lo || m.ty_hi <= m.ty_lo) m.tx_hi = m.tx_lo = m.ty_hi = m.ty_lo = 0;   // no interior at all
  const int n_int = (m.tx_hi - m.tx_lo) * (m.ty_hi - m.ty_lo);
  const int n_edge = base.tiles_x * m.grid_rows - n_int;
  const dim3 gi(n_int, n_frames), ge(n_edge, n_frames), block(kBlock);
#define PQA_LAUNCH_MFMA(T)                                                                          \
  do {                                                                                              \
    if (n_int > 0) hipLaunchKernelGGL((vif_s0_mfma_kernel<T, false>), gi, block, 0, stream, m);     \
    if ((*err = hipGetLastError()) != hipSuccess) return true;                                      \
    if (n_edge > 0) hipLaunchKernelGGL((vif_s0_mfma_kernel<T, true>), ge, block, 0, stream, m);     \
    *err = hipGetLastError();                                                                       \
  } while (0)
  if (ten_bit) PQA_LAUNCH_MFMA(uint16_t); else PQA_LAUNCH_MFMA(uint8_t);
#undef PQA_LAUNCH_MFMA
  return true;
}

constexpr int kVifN[4] = {17, 9, 5, 3};
constexpr int kVifTW[4] = {240, 248, 252, 252};

}  // namespace

int vif_tile_w(int scale) { return kVifTW[scale]; }

hipError_t vif_mfma_prepare() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipSuccess;   // no table: scale 0 stays on the VALU kernel on such a device
  std::lock_guard<std::mutex> lock(g_atab_mu);
  if (g_atab[dev]) return hipSuccess;
  std::vector<uint16_t> h((size_t)kAtabFrags * 64 * 8);
  if (!build_atab(h.data())) return hipSuccess;  // a tap that does not split exactly into f16 pieces: VALU kernel
  void* d = nullptr;
  if ((e = hipMalloc(&d, h.size() * 2)) != hipSuccess) return e;
  if ((e = hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(d);
    return e;
  }
  {  // the probe (see f16_tiny_probe_kernel): without kept denormals scale 0 stays on the VALU kernel, and says so once
    int* ok = nullptr;
    int seen = 0;
    if ((e = hipMalloc((void**)&ok, sizeof(int))) == hipSuccess) {
      e = hipMemset(ok, 0, sizeof(int));
      if (e == hipSuccess) {
        hipLaunchKernelGGL(f16_tiny_probe_kernel, dim3(1), dim3(64), 0, 0, ok);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpy(&seen, ok, sizeof(int), hipMemcpyDeviceToHost);
      (void)hipFree(ok);
    }
    if (e != hipSuccess) {
      (void)hipFree(d);
      return e;
    }
    if (seen != 64) {
      fprintf(stderr, "pqa_vmaf: device %d does not keep f16 denormals (%d of 64 probe lanes exact): VIF scale 0 runs the "
                      "VALU kernel instead of the matrix-core kernel\n", dev, seen);
      (void)hipFree(d);
      return hipSuccess;
    }
  }
  g_atab[dev] = (const uint4*)d;   // lives as long as the process (8 KB per device)
  return hipSuccess;
}

hipError_t launch_vif_stat(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                           int w, int h, float inv_scale, float gain_limit, int border101, double* partials,
                           MutPlaneRun next_ref, MutPlaneRun next_dis, int s0_mode, int* n_partials) {
  if (n_partials) *n_partials = vif_tiles_x(scale, w) * vif_tiles_y(h);
  if (n_frames <= 0) return hipSuccess;
  // (10 bit is recognised by its sample scale 1/4; 12-bit clips have squares of 22 bits, three digits: tiled kernels)
  if (scale == 0 && (elem == ELEM_U8 || (elem == ELEM_U16 && inv_scale == 0.25f)) && s0_mode == VIF_S0_AUTO && next_ref.base &&
      next_dis.base) {
    hipError_t err = hipSuccess;
    if (launch_vif_s0_march(stream, elem, ref, dis, n_frames, w, h, gain_limit, border101, partials, next_ref, next_dis, n_partials, &err))
      return err;
    if (n_partials) *n_partials = vif_tiles_x(scale, w) * vif_tiles_y(h);
  }
  VifStatArgs a{};
  a.ref = ref.base; a.dis = dis.base;
  a.row_pitch_r = ref.row_pitch; a.frame_pitch_r = ref.frame_pitch;
  a.row_pitch_d = dis.row_pitch; a.frame_pitch_d = dis.frame_pitch;
  a.w = w; a.h = h;
  a.fold_w = 2 * w - (border101 ? 2 : 1); a.fold_h = 2 * h - (border101 ? 2 : 1);
  a.tiles_x = vif_tiles_x(scale, w);
  a.n_tiles = a.tiles_x * vif_tiles_y(h);
  a.inv_scale = inv_scale; a.gain_limit = gain_limit;
  a.partials = partials;
  a.dst_ref = (float*)next_ref.base; a.dst_dis = (float*)next_dis.base;
  a.dst_row_pitch_r = next_ref.row_pitch; a.dst_frame_pitch_r = next_ref.frame_pitch;
  a.dst_row_pitch_d = next_dis.row_pitch; a.dst_frame_pitch_d = next_dis.frame_pitch;
  const Taps cur = gaussian_taps(kVifN[scale]);
  const Taps nxt = scale < 3 ? gaussian_taps(kVifN[scale + 1]) : Taps{};
  a.taps = tap_pairs(cur, kVifN[scale], scale < 3 ? &nxt : nullptr, scale < 3 ? kVifN[scale + 1] : 0);
  if (scale < 3 && (!a.dst_ref || !a.dst_dis)) return hipErrorInvalidValue;
  // scale 0 of 10-bit clips (and of 8-bit clips the march kernel could not take, or with VIF_S0_SPLIT): vertical pass on
  // the matrix cores (10 bit is recognised by its sample scale 1/4; 12-bit clips and every deeper scale run the VALU kernel)
  if (scale == 0 && s0_mode != VIF_S0_VALU && (elem == ELEM_U8 || (elem == ELEM_U16 && inv_scale == 0.25f))) {
    hipError_t err = hipSuccess;
    if (launch_s0_split(stream, elem == ELEM_U16, a, n_frames, &err)) return err;
  }
  switch (scale) {
    case 0: return launch_stat_n<17, 240, 9>(stream, elem, a, n_frames);
    case 1: return launch_stat_n<9, 248, 5>(stream, elem, a, n_frames);
    case 2: return launch_stat_n<5, 252, 3>(stream, elem, a, n_frames);
    case 3: return launch_stat_n<3, 252, 0>(stream, elem, a, n_frames);
  }
  return hipErrorInvalidValue;
}

}  // namespace pqa
