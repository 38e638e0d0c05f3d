// Host-side launch interface of the gfx950 feature kernels.  All pointers are device pointers,
// pitches are in ELEMENTS of the plane's sample type.  Every launcher is asynchronous on `stream`
// and allocates nothing (workspaces come from the context), so a caller may capture them in a
// hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqa {

enum Elem : int { ELEM_U8 = 0, ELEM_U16 = 1, ELEM_F32 = 2 };

// A run of equally-shaped planes: frame f lives at base + f * frame_pitch.
struct PlaneRun {
  const void* base;
  int64_t row_pitch;    // elements
  int64_t frame_pitch;  // elements
};

struct MutPlaneRun {
  void* base;
  int64_t row_pitch;
  int64_t frame_pitch;
};

// ---- VIF ------------------------------------------------------------------------------------
// Tile geometry of the statistic kernel at a scale (filter widths 17/9/5/3).
int vif_tile_w(int scale);
constexpr int kVifTileH = 8;
inline int vif_tiles_x(int scale, int w) { return (w + vif_tile_w(scale) - 1) / vif_tile_w(scale); }
inline int vif_tiles_y(int h) { return (h + kVifTileH - 1) / kVifTileH; }

// partials: [n_frames][tiles][2] doubles (num, den), tiles = tiles_x * tiles_y.
// border101: 0 = vif_tools.c border rule (high edge repeated), 1 = integer_vif.c padding (reflect-101).
// For scale < 3 the same launch also produces the next scale's input (fused decimation): the planes are
// filtered with the next scale's kernel and even samples kept -> next_ref / next_dis, (w/2 x h/2) f32.
// *n_partials (nullable) receives the number of (num, den) pairs per frame the launch wrote: vif_tiles_x * vif_tiles_y for
// the tiled kernels, one per wave segment for the march kernel of scale 0 (vif_march.hip); the finalize stage sums that many.
// s0_mode: which kernel scale 0 runs (read once per context from PQA_VIF_MFMA in pqa_create: A/B runs and the tests that
// compare the paths): VIF_S0_AUTO = the march kernel (8-, 10- and 12-bit clips);
// VIF_S0_VALU = VALU only.
enum : int { VIF_S0_VALU = 0, VIF_S0_AUTO = 1 };
hipError_t launch_vif_stat(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                           int w, int h, float inv_scale, float gain_limit, int border101, double* partials,
                           MutPlaneRun next_ref, MutPlaneRun next_dis, int s0_mode = VIF_S0_AUTO, int* n_partials = nullptr);
// Scale 0 (8-, 10-, 12-bit clips: `bits`), both filter passes on the f16 matrix cores (vif_march.hip).  vif_march_prepare uploads its tap
// table once per device (pqa_create does; synchronous, idempotent; a device that does not keep f16 denormals -- probed once,
// the operand encoding leans on them -- gets no table and scale 0 stays on the VALU kernel); launch_vif_s0_march returns false when it cannot take
// the planes (no table, pitches the stores cannot take) and launch_vif_stat then falls back to the tiled kernels.
// vif_march_partials_max: upper bound of *n_partials for a w x h frame (workspace sizing).
hipError_t vif_march_prepare();
int vif_march_partials_max(int w, int h);
// host: the per-lane tap-matrix fragments the march kernel reads ([fragment][lane 0..63][8 f16 bit patterns]) -- no device
// needed; returns the number of fragments, 0 when a tap does not split exactly, -(halfwords needed) when `out` is too small
int vif_march_table(uint16_t* out, int capacity_halfwords);
// {16-column stripes, 16-row blocks, blocks per segment, segments, pass-1 MFMAs per block, pass-2 MFMAs per block} of a w x h frame
void vif_march_shape(int w, int h, int* out6);
bool launch_vif_s0_march(hipStream_t stream, Elem elem, int bits, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h, float gain_limit,
                         int border101, double* partials, MutPlaneRun next_ref, MutPlaneRun next_dis, int* n_partials,
                         hipError_t* err);
// Fixed-point VIF (integer_vif.c arithmetic, vif_fixed.hip): same tiling; partials are [n_frames][tiles][8] int64
// {num_log, den_log, x, x2, n_log, den_non_log, num_non_log, -}; next_ref / next_dis are u16 planes (w/2 x h/2).
// elem: scale 0 reads the caller's samples (ELEM_U8 at 8 bit, ELEM_U16 above), deeper scales ELEM_U16.
constexpr int kVifFxPartials = 8;
void vif_fixed_log2_table(uint16_t* out32768);  // host: entry i = round(log2f(32768 + i) * 2048)
hipError_t launch_vif_fixed(hipStream_t stream, int scale, int bit_depth, Elem elem, PlaneRun ref, PlaneRun dis,
                            int n_frames, int w, int h, double gain_limit, const uint16_t* log2_lut,
                            long long* partials, MutPlaneRun next_ref, MutPlaneRun next_dis);

// ---- ADM ------------------------------------------------------------------------------------
constexpr int kAdmTileW = 60, kAdmTileH = 14;  // 126 input columns (one per lane), 16 halo'd rows = 8 row pairs
inline int adm_tiles_x(int band_w) { return (band_w + kAdmTileW - 1) / kAdmTileW; }
inline int adm_tiles_y(int band_h) { return (band_h + kAdmTileH - 1) / kAdmTileH; }

// One ADM scale: db2 DWT of (w x h) planes -> decouple -> CSF -> contrast masking.
// partials: [n_frames][tiles][6] doubles (num h,v,d cube sums; den h,v,d cube sums).
// ll_ref/ll_dis receive the approximation band (ceil(w/2) x ceil(h/2)) for the next scale
// (base may be null at the last scale).
// mode (read once per context from PQA_ADM_MARCH / PQA_ADM_PYRAMID in pqa_create): ADM_AUTO = the march kernel (adm_march.hip: one partial
// sextet per wave segment), ADM_TILED = the LDS-tiled kernel (adm.hip: one per 60 x 14 tile; A/B partner, and the fallback
// for planes of 2 GiB and more).  *n_partials (nullable) receives the number of sextets per frame the launch wrote.
enum : int { ADM_TILED = 0, ADM_AUTO = 1, ADM_MARCH = 2 };   // ADM_MARCH: the march kernel, one scale per launch (no pyramid)
hipError_t launch_adm_scale(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames,
                            int w, int h, float inv_scale, float gain_limit, MutPlaneRun ll_ref,
                            MutPlaneRun ll_dis, double* partials, int mode = ADM_AUTO, int* n_partials = nullptr);
// The march kernel (adm_march.hip).  adm_march_partials: sextets per frame it writes for a band_w x band_h band (workspace
// sizing; a function of the geometry only).  launch_adm_march returns false when it cannot take the planes.
int adm_march_partials(int band_w, int band_h);
bool launch_adm_march(hipStream_t stream, int scale, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h,
                      float inv_scale, float gain_limit, MutPlaneRun ll_ref, MutPlaneRun ll_dis, double* partials,
                      int* n_partials, hipError_t* err);
// Scales 0 and 1 in ONE launch (adm_pyramid.hip): scale 0's approximation band stays in registers; partials0 / partials1
// receive one sextet per wave segment each (the same count, *n_partials), ll_ref / ll_dis the approximation band of SCALE 1
// (w/4 x h/4: the input of scale 2).  adm_pyramid_takes: u8 / u16 planes whose width and height are multiples of 4 and >= 128;
// launch_adm_pyramid returns false when it cannot take the planes (the caller then runs one scale per launch).
bool adm_pyramid_takes(Elem elem, int w, int h);
int adm_pyramid_partials(int w, int h);
bool launch_adm_pyramid(hipStream_t stream, Elem elem, PlaneRun ref, PlaneRun dis, int n_frames, int w, int h, float inv_scale,
                        float gain_limit, MutPlaneRun ll_ref, MutPlaneRun ll_dis, double* partials0, double* partials1,
                        int* n_partials, hipError_t* err);
float adm_dwt_quant_step(int lambda, int theta);   // Watson model step (adm_tools.h dwt_quant_step), theta 1 = h/v, 2 = d

// Fixed-point ADM (integer_adm.c arithmetic, adm_fixed.hip).  Same tiling as launch_adm_scale; the approximation
// bands handed to the next scale are int32 planes (4-byte elements: pass them on as ELEM_F32-sized runs).
// partials: [n_frames][tiles][kAdmFxRows][6] int64 per-row sums {num h,v,d, den h,v,d} (row 0 and 15 = halo, zero).
constexpr int kAdmFxRows = kAdmTileH + 2;
struct AdmFxScale {           // per-scale constants of the fixed-point path, shared by kernel, finalize and host epilogue
  int scale, band_w, band_h;
  int left, top, right, bottom;
  float rf[3];
  uint32_t i_rf[3];
  int cm_shift_sq[3], cm_shift_sub[3], cm_shift_cub[3], cm_final_q[3];
  int den_shift_sq, den_shift_cub, den_final_q;
  int num_row_shift, den_row_shift;
};
AdmFxScale adm_fixed_scale_params(int scale, int band_w, int band_h);
void adm_fixed_div_table(int32_t* out65537);  // host: div_lookup of integer_adm.c (2^30 / i, odd-symmetric)
// host: {num_scale, den_scale} from the six integer accumulators, as adm_cm / adm_csf_den_scale finish them
void adm_fixed_epilogue(const AdmFxScale& p, const long long acc[6], double* num_out, double* den_out);
hipError_t launch_adm_fixed(hipStream_t stream, int scale, int bit_depth, Elem elem, PlaneRun ref, PlaneRun dis,
                            int n_frames, int w, int h, double gain_limit, const int32_t* div_lut,
                            MutPlaneRun ll_ref, MutPlaneRun ll_dis, long long* partials);

// ---- motion ---------------------------------------------------------------------------------
constexpr int kMotionTileW = 252, kMotionTileH = 16;
inline int motion_tiles(int w, int h) {
  return ((w + kMotionTileW - 1) / kMotionTileW) * ((h + kMotionTileH - 1) / kMotionTileH);
}
// SAD of the 5-tap-blurred reference luma of frame f against frame f-1.  Frame 0 of the run uses
// `prev0` (may be null: then its partials are zero).  partials: [n_frames][tiles] doubles.
// mode (PQA_MOTION_MARCH, read once per context): MOTION_AUTO = the march kernel (motion_march.hip: one partial per wave
// segment), MOTION_TILED = the LDS-tiled kernel (motion.hip: one per 252 x 16 tile; test partner and fallback for planes of
// 2 GiB and more).  *n_partials (nullable) receives the number of partials per frame the launch wrote.
enum : int { MOTION_TILED = 0, MOTION_AUTO = 1 };
hipError_t launch_motion(hipStream_t stream, Elem elem, PlaneRun ref, const void* prev0, int64_t prev0_row_pitch,
                         int n_frames, int w, int h, float inv_scale, double* partials, int mode = MOTION_AUTO,
                         int* n_partials = nullptr);
int motion_march_partials(int w, int h);   // partials per frame of the march kernel (workspace sizing; geometry only)
bool launch_motion_march(hipStream_t stream, Elem elem, PlaneRun ref, const void* prev0, int64_t prev0_row_pitch, int n_frames,
                         int w, int h, double* partials, int* n_partials, hipError_t* err);
// Fixed-point motion (integer_motion.c arithmetic, motion_fixed.hip): partials are [n_frames][tiles] uint64 SADs of
// the Q8 blurred planes.
hipError_t launch_motion_fixed(hipStream_t stream, int bit_depth, Elem elem, PlaneRun ref, const void* prev0,
                               int64_t prev0_row_pitch, int n_frames, int w, int h, unsigned long long* partials);

// ---- PSNR (FFmpeg psnr filter) and SSIM (FFmpeg ssim filter) ----------------------------------
constexpr int kSseBlocksPerPlane = 256;
// partials: [n_frames][kSseBlocksPerPlane] uint64 SSE of the (w x h) rectangle starting at a.base / b.base
hipError_t launch_sse(hipStream_t stream, Elem elem, PlaneRun a, PlaneRun b, int n_frames, int w, int h,
                      unsigned long long* partials);

constexpr int kSsimTileBW = 32, kSsimTileBH = 8;  // tile = 32 x 8 windows (4x4-block grid)
inline int ssim_tiles(int w, int h) {
  const int ww = (w >> 2) - 1, wh = (h >> 2) - 1;
  if (ww < 1 || wh < 1) return 0;
  return ((ww + kSsimTileBW - 1) / kSsimTileBW) * ((wh + kSsimTileBH - 1) / kSsimTileBH);
}
// main = distorted, ref = reference (order of the filter's inputs).  partials: [n_frames][tiles] doubles.
// sse_partials (nullable): [n_frames][tiles] uint64 -- the squared error of the 4*(w>>2) x 4*(h>>2) part of the
// plane falls out of the block sums the SSIM needs anyway (ss - 2*s12), so PSNR costs no second pass; the
// right / bottom remainder strips (w or h not a multiple of 4) are left to launch_sse.
hipError_t launch_ssim(hipStream_t stream, Elem elem, PlaneRun main, PlaneRun ref, int n_frames, int w, int h,
                       int max_value, double* partials, unsigned long long* sse_partials);

// ---- luma statistics (white bookend-frame detection, the step before the scoring path) ----------
constexpr int kLumaBlocks = 128;
// out: [n_frames][3] uint64 = {sum, sum of squares, count(sample > threshold)}; partials: [n_frames][kLumaBlocks][3].
// gray_bit_depth 0: statistics of the samples as they are; 8/10/12: of gray = clamp(round((Y - 16 s) * 255 / (219 s)), 0, 255),
// s = 2^(bpc - 8) -- the limited -> full range expansion behind cv2's BGR2GRAY --, threshold in those 8-bit units.
hipError_t launch_luma_stats(hipStream_t stream, Elem elem, PlaneRun luma, int n_frames, int w, int h,
                             unsigned threshold, int gray_bit_depth, unsigned long long* partials,
                             unsigned long long* out);
void luma_gray_map(int bit_depth, float* a, float* b);   // the f32 (scale, offset) pair the kernel applies

// ---- finalize -------------------------------------------------------------------------------
// Fixed-order reduction of every partial array of a batch into per-frame records.
struct FinalizeArgs {
  int n_frames;
  int has_vif, has_adm, has_motion, n_sse_planes, n_ssim_planes;
  const double* vif_part[4];   int vif_tiles[4];
  const long long* vif_fx_part[4];            // non-null: fixed-point VIF partials (kVifFxPartials int64 per tile)
  const double* adm_part[4];   int adm_tiles[4];   float adm_area[4];  // cropped-window area per scale
  const long long* adm_fx_part[4];            // non-null: fixed-point ADM per-row partials
  int adm_fx_tiles_x[4], adm_fx_top[4], adm_fx_bottom[4], adm_fx_num_shift[4], adm_fx_den_shift[4];
  long long* adm_fx_acc;                      // [capacity][4][6] ring: the six accumulators per frame and scale
  const double* motion_part;   int motion_tiles;   double motion_norm;  // 2^-(bpc-8) / (w*h)
  const unsigned long long* motion_fx_part;   // non-null: fixed-point motion SAD partials
  unsigned motion_wh;                         // w * h, for normalize_and_scale_sad()
  const unsigned long long* sse_part[3];      // [n_frames][kSseBlocksPerPlane]: whole plane, or the right strip
  const unsigned long long* sse_part_b[3];    // nullable: bottom strip
  const unsigned long long* sse_tile_part[3]; // nullable: [n_frames][ssim_tiles] from the SSIM kernel
  int sse_use_a[3];                           // 1 when sse_part holds data for this batch
  const double* ssim_part[3];  int ssim_tiles[3];  double ssim_norm[3]; // 1/(windows)
  double* records;             // [capacity][record_stride] ring
  int slot_base, slot_step, capacity;  // record row of batch frame f = (slot_base + f*slot_step) % capacity
  int record_stride;
};
hipError_t launch_finalize(hipStream_t stream, const FinalizeArgs& args);

}  // namespace pqa
