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
// Which kernel runs where (launch_vif_stat at the bottom): scales 1-3: vif_stat_kernel (VALU);
// scale 0 (8-, 10- and 12-bit clips): vif_s0_march_kernel in vif_march.hip (both filter passes on the f16 matrix cores), with
// vif_stat_kernel as its test partner (PQA_VIF_MFMA=0) and its fallback (no tap table, next-scale planes whose pitches its
// 16-byte stores cannot take).  The round-2 kernel that put only the vertical pass on the matrix cores left the build in
// round 4: tools/experiments/vif_s0_mfma_round2.hip.txt.
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
  TapPairs taps;
};

constexpr int kP2 = 258;  // LDS row pitch in float2: == 2 (mod 32) -> conflict-free ds_read_b128 (see below)

// FP32 FMA throughput on gfx950 is 64 FLOP/clk/SIMD however it is issued (v_pk_fma_f32 every 4 clocks, plain
// v_fma_f32 every 2 with >= 2 waves per SIMD, f32 MFMA: tools/ubench/pk_vs_plain.hip, mfma_coissue.hip); the packed
// form halves the instruction count.  Everything hot is arranged as float2 = {row 2p, row 2p+1} of one column:
//   vertical pass:   acc{2p,2p+1} += {c[k], c[k-1]} * {x, x}     (tap PAIR from SGPRs, input broadcast)
//   horizontal pass: out{2p,2p+1}[o] += c[k] * in{2p,2p+1}[o+k]  (tap broadcast, input pair from LDS)

// Phases 1b (fused decimation), 2 (horizontal pass) and 3 (statistic + tile partial) of one TW x 8 tile whose
// vertical-pass results are in LDS: sv[signal][row pair][column] = {row 2p, row 2p+1}, sd[even row][column] =
// {ref, dis}.
template <int N, int TW, int ND>
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
      if (oc < TW / 2 && (ox0 + oc < ow && oy0 + orow < oh)) {
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
    const bool vrow[2] = {gyA < a.h, gyA + 1 < a.h};
    f2 num2 = f2{0.0f, 0.0f}, den2 = f2{0.0f, 0.0f};
    // the log terms of the sigma1_sq >= sigma_nsq branch are summed as the log of a product: per row of
    // the pair the four columns' arguments (each in [2, 2^16]) are multiplied first, so the thread takes
    // 6 v_log_f32 instead of 16 and needs no reciprocal for num's ratio (log a/b = log a - log b)
    f2 pn = f2{1.0f, 1.0f}, qn = f2{1.0f, 1.0f}, pd = f2{1.0f, 1.0f};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const bool vcol = (x0 + seg * 4 + o) < a.w;
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

template <typename T, int N, int TW, int ND>
__global__ __launch_bounds__(kBlock, 3) void vif_stat_kernel(const VifStatArgs a) {
  constexpr int R = N / 2, TH = kVifTileH, COLS = TW + N - 1, NSEG = TW / 4, S = 8, NIN = S + N - 1;
  constexpr int RD = ND / 2;  // ND = taps of the next scale's filter (0: last scale, no decimation)
  static_assert(ND == 0 || RD <= R, "decimation window must sit inside the statistic window");
  static_assert(COLS <= 256 && COLS <= kP2, "one column per thread");
  static_assert(TW % 4 == 0 && NSEG <= 64 && TH == S, "segment map");
  __shared__ f2 sv[5][TH / 2][kP2];  // [signal][row pair][column] = {row 2p, row 2p+1}
  __shared__ f2 sd[ND ? TH / 2 : 1][ND ? kP2 : 1];  // decimation: [even row][column] = {ref, dis}
  __shared__ double red[12];

  const int tile = xcd_remap(blockIdx.x, a.n_tiles);
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

constexpr int kVifN[4] = {17, 9, 5, 3};
constexpr int kVifTW[4] = {240, 248, 252, 252};

}  // namespace

int vif_tile_w(int scale) { return kVifTW[scale]; }

hipError_t launch_vif_stat(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                           int w, int h, float inv_scale, float gain_limit, int border101, double* partials,
                           MutPlaneRun next_ref, MutPlaneRun next_dis, int s0_mode, int* n_partials) {
  if (n_partials) *n_partials = vif_tiles_x(scale, w) * vif_tiles_y(h);
  if (n_frames <= 0) return hipSuccess;
  // (the bit depth of 16-bit elements is recognised by the sample scale: 1/4 = 10 bit, 1/16 = 12 bit)
  const int bits = elem == ELEM_U8 ? 8 : (elem == ELEM_U16 && inv_scale == 0.25f) ? 10 : (elem == ELEM_U16 && inv_scale == 0.0625f) ? 12 : 0;
  if (scale == 0 && bits && s0_mode == VIF_S0_AUTO && next_ref.base && next_dis.base) {
    hipError_t err = hipSuccess;
    if (launch_vif_s0_march(stream, elem, bits, ref, dis, n_frames, w, h, gain_limit, border101, partials, next_ref, next_dis, n_partials, &err))
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
  switch (scale) {
    case 0: return launch_stat_n<17, 240, 9>(stream, elem, a, n_frames);
    case 1: return launch_stat_n<9, 248, 5>(stream, elem, a, n_frames);
    case 2: return launch_stat_n<5, 252, 3>(stream, elem, a, n_frames);
    case 3: return launch_stat_n<3, 252, 0>(stream, elem, a, n_frames);
  }
  return hipErrorInvalidValue;
}

}  // namespace pqa
