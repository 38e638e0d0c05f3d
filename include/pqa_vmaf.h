/*
 * pqa_vmaf.h -- C ABI of the MI355X-native VMAF feature engine (libpqa_vmaf.so).
 *
 * Drop-in boundary.  In the reference (yoseph007/PQA2) the whole scoring hot path is three child
 * processes started by VMAFAnalyzer.analyze_videos():
 *     subprocess.Popen([ffmpeg, ..., "-lavfi", "libvmaf=..."])      app/vmaf_analyzer.py:406-419,446
 *     subprocess.run([ffmpeg, ..., "-lavfi", "psnr=stats_file=..."]) app/vmaf_analyzer.py:1027-1045
 *     subprocess.run([ffmpeg, ..., "-lavfi", "ssim=stats_file=..."]) app/vmaf_analyzer.py:1057-1075
 * This library sits where those calls are: the host feeds decoded planes, the library returns one
 * fixed-size feature record per frame, and the host (Python, as in the reference) applies the
 * bundled models/vmaf_*.json SVM, pools, and writes the libvmaf-format JSON / stats files that
 * _parse_vmaf_results (app/vmaf_analyzer.py:628-964) reads back.  INTEGRATION.md shows the binding.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns PQA_OK (0) or a
 * negative pqa_status; pqa_last_error() explains the most recent failure.  The caller owns every
 * buffer it passes in; the library owns all device memory it allocates.  A context is
 * single-threaded (mirrors VMAFAnalyzer._process_lock, app/vmaf_analyzer.py:29,251); the only call
 * legal from another thread is pqa_cancel() (mirrors terminate_analysis, app/vmaf_analyzer.py:139).
 * One context per GPU.
 */
#ifndef PQA_VMAF_H
#define PQA_VMAF_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define PQA_API __attribute__((visibility("default")))
#else
#define PQA_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pqa_status {
  PQA_OK = 0,
  PQA_EINVAL = -1,     /* bad argument / unsupported geometry */
  PQA_EDEVICE = -2,    /* HIP runtime failure (text in pqa_last_error) */
  PQA_ENOMEM = -3,     /* host or device allocation failed */
  PQA_ECANCELLED = -4, /* pqa_cancel() was called; pqa_reset() re-arms the context */
  PQA_ESTATE = -5      /* call sequence error: collecting a frame that was never submitted (or whose record a
                          later frame has replaced), or submitting a frame whose ring slot still holds the
                          UNCOLLECTED record of a different frame (result_capacity too small for the caller's
                          collect cadence).  Nothing is launched or overwritten when this is returned. */
} pqa_status;

/* feature mask: which extractors run per frame */
enum {
  PQA_FEAT_VIF = 1u << 0,    /* libvmaf float VIF, 4 scales        (libvmaf= call site, :377-419) */
  PQA_FEAT_ADM = 1u << 1,    /* libvmaf float ADM, 4 scales        (same call site)               */
  PQA_FEAT_MOTION = 1u << 2, /* libvmaf motion (SAD of blurred ref) (same call site)              */
  PQA_FEAT_PSNR = 1u << 3,   /* FFmpeg psnr filter: per-plane SSE   (:1027-1034)                  */
  PQA_FEAT_SSIM = 1u << 4,   /* FFmpeg ssim filter: per-plane SSIM  (:1057-1064)                  */
  PQA_FEAT_VMAF = PQA_FEAT_VIF | PQA_FEAT_ADM | PQA_FEAT_MOTION,
  PQA_FEAT_ALL = PQA_FEAT_VMAF | PQA_FEAT_PSNR | PQA_FEAT_SSIM
};

/* One record = PQA_RECORD_DOUBLES 8-byte slots per frame. */
enum {
  PQA_REC_VIF_NUM = 0,  /* [4] vif numerator per scale                                        */
  PQA_REC_VIF_DEN = 4,  /* [4] vif denominator per scale     (vif_scale_s = num/den)           */
  PQA_REC_ADM_NUM = 8,  /* [4] adm numerator per scale                                        */
  PQA_REC_ADM_DEN = 12, /* [4] adm denominator per scale     (adm2 = sum num / sum den)        */
  PQA_REC_MOTION = 16,  /*     motion_i (motion2 is a host-side min over neighbours)           */
  PQA_REC_SSIM = 17,    /* [3] FFmpeg ssim Y, U, V                                            */
  PQA_REC_SSE = 20,     /* [3] FFmpeg psnr SSE Y, U, V: uint64 bit-cast into the slot (exact)   */
  PQA_REC_RESERVED = 23,
  PQA_RECORD_DOUBLES = 24
};

typedef struct pqa_config {
  uint32_t struct_size;        /* sizeof(pqa_config), for ABI growth                              */
  int32_t device;              /* HIP device ordinal                                              */
  uint32_t width, height;      /* luma size, both >= 16                                           */
  uint32_t bit_depth;          /* 8, 10 or 12 (samples > 8 bit are little-endian uint16)          */
  uint32_t n_planes;           /* 1 = luma only, 3 = Y,U,V (needed for chroma PSNR/SSIM)          */
  uint32_t chroma_hshift;      /* log2 horizontal chroma subsampling (4:2:0 -> 1)                 */
  uint32_t chroma_vshift;      /* log2 vertical chroma subsampling   (4:2:0 -> 1)                 */
  uint32_t features;           /* PQA_FEAT_* mask                                                 */
  uint32_t max_batch;          /* frames per kernel launch (0 -> auto: ~1.5 GiB of luma, 8..256)   */
  uint32_t result_capacity;    /* records kept on the device, ring indexed by frame_index
                                  (0 -> default 16384)                                            */
  uint32_t n_subsample;        /* libvmaf n_subsample (:379): VIF/ADM on frames i % k == 0 only;
                                  motion on every frame (0/1 -> every frame)                      */
  double vif_enhn_gain_limit;  /* 100.0 default; 1.0 for *neg models (feature_opts_dicts)         */
  double adm_enhn_gain_limit;  /* 100.0 default; 1.0 for *neg models                              */
  uint32_t vif_border;         /* PQA_VIF_BORDER_*: which libvmaf extractor's VIF padding to follow  */
  uint32_t fixed_point;        /* PQA_FIXED_* mask: extractors to run in libvmaf's fixed-point arithmetic
                                  instead of f32 (0 = none: the fast path, the default)              */
} pqa_config;

/* libvmaf has two VIF extractors with different image-border handling.  `model=version=vmaf_v0.6.1`
 * (app/vmaf_analyzer.py:377) names VMAF_integer_feature_vif_* (models/vmaf_v0.6.1.json:31-38), i.e.
 * integer_vif.c, which pads by reflect-101 on all four edges; the vmaf_float_* models name float_vif
 * (vif_tools.c), which repeats the edge sample at the bottom/right edge.  The arithmetic here is f32 in
 * both cases (DESIGN.md "float vs fixed-point" quantifies the residual); this selects the border only.
 * pqa_config.fixed_point goes the whole way per extractor: PQA_FIXED_VIF = integer_vif.c (Q16 taps, Q8 means,
 * 2048-step log2 table, integer accumulators; implies the integer border), PQA_FIXED_MOTION = integer_motion.c
 * (Q8 blurred planes, integer SAD), PQA_FIXED_ADM = integer_adm.c (Q15 db2, int16/int32 bands, reciprocal table,
 * shifted cube accumulators; the six cube roots per scale are taken on the host inside pqa_collect).
 * Bit-identical to oracle/vmaf_int_oracle.c; slower than the f32 path (DESIGN.md section 3 has the numbers). */
enum { PQA_FIXED_VIF = 1, PQA_FIXED_MOTION = 2, PQA_FIXED_ADM = 4, PQA_FIXED_ALL = 7 };
enum {
  PQA_VIF_BORDER_FLOAT = 0,   /* vif_tools.c:   index -i -> i,  n-1+i -> n-i    */
  PQA_VIF_BORDER_INTEGER = 1  /* integer_vif.c: index -i -> i,  n-1+i -> n-1-i  */
};

/* A clip already resident in device memory (HBM): frame f of plane p starts at
 * plane[p] + f * frame_pitch[p]; rows are row_pitch[p] bytes apart.  Pitches are in BYTES. */
typedef struct pqa_device_clip {
  const void* plane[3];
  int64_t row_pitch[3];
  int64_t frame_pitch[3];
} pqa_device_clip;

typedef struct pqa_ctx pqa_ctx;

/* Library / record introspection. */
PQA_API const char* pqa_version(void);
PQA_API int pqa_record_doubles(void);

/* Fill cfg with defaults (8-bit 4:2:0, PQA_FEAT_VMAF, gain limits 100). */
PQA_API void pqa_config_init(pqa_config* cfg, uint32_t width, uint32_t height);

/* Create / destroy.  Replaces process start-up of the ffmpeg child (app/vmaf_analyzer.py:446). */
PQA_API int pqa_create(const pqa_config* cfg, pqa_ctx** out);
PQA_API void pqa_destroy(pqa_ctx* ctx);

/* Run all work of this context on an existing HIP stream (e.g. torch's current stream).
 * NULL restores the context's own stream. */
PQA_API int pqa_set_stream(pqa_ctx* ctx, void* hip_stream);

/* Submit one decoded frame pair from HOST memory.  planes[p] / strides[p] (bytes) for p < n_planes.
 * The library copies into pinned staging, uploads on a copy stream and launches kernels once
 * max_batch frames are pending (or at pqa_collect / pqa_flush).  Frames must arrive in increasing,
 * consecutive frame_index order within a clip; motion of the first submitted frame is 0 unless
 * pqa_set_motion_halo() supplied its predecessor.
 * Replaces one frame's worth of the libvmaf/psnr/ssim filter graph input. */
PQA_API int pqa_submit(pqa_ctx* ctx, int64_t frame_index, const void* const ref_planes[3], const int64_t ref_strides[3],
               const void* const dis_planes[3], const int64_t dis_strides[3]);

/* The same for a frame pair that lies in two FILES as packed planes (rows width * sample-size bytes apart, no padding --
 * raw .yuv and .y4m payloads): plane p of the reference frame starts at byte ref_plane_offsets[p] of ref_fd, likewise for
 * the distorted clip.  The library reads (pread) straight into its pinned staging with its packing threads -- one copy out
 * of the page cache and no page faults, against mapping the file and copying from the mapping.  The descriptors are only
 * read, never closed, and their file positions are not moved.  A short read (truncated file, bad offset) is PQA_EINVAL.
 * This is the shape of the reference's inputs: two files (app/vmaf_analyzer.py:411-419, `-i distorted -i reference`). */
PQA_API int pqa_submit_fd(pqa_ctx* ctx, int64_t frame_index, int ref_fd, const int64_t ref_plane_offsets[3], int dis_fd,
                          const int64_t dis_plane_offsets[3]);

/* n_frames CONSECUTIVE frame pairs of two such files in one call: frame first_index + k has plane p at byte
 * ref_plane_offsets[p] + k * ref_frame_stride of ref_fd (a .y4m payload: stride = frame bytes + 6 for "FRAME\n"; raw .yuv:
 * the frame bytes), likewise for the distorted clip.  Same results as n_frames calls of pqa_submit_fd; inside the call the
 * packing threads read frame k + 1 while frame k is on its way to the device -- one wake-up of the threads per staging half
 * (8 frames) instead of one per frame, which is what lets a 2160p clip approach the PCIe rate.  n_frames may be any number
 * up to result_capacity; a caller that reports progress or polls for cancellation submits in runs of a few frames.
 * A short read anywhere in a run is PQA_EINVAL and none of the frames of the staging half it occurred in is submitted. */
PQA_API int pqa_submit_fd_run(pqa_ctx* ctx, int64_t first_index, int32_t n_frames, int ref_fd,
                              const int64_t ref_plane_offsets[3], int64_t ref_frame_stride, int dis_fd,
                              const int64_t dis_plane_offsets[3], int64_t dis_frame_stride);

/* Submit n_frames consecutive frame pairs that are ALREADY in device memory (no copies).  "Already" includes ordering:
 * the context's kernels run on its own stream (or the one given to pqa_set_stream), so whatever produced the frames must be
 * complete -- or on that same stream -- before this call; the same holds for pqa_submit_surfaces and
 * pqa_luma_stats_device.
 * prev_ref_luma (nullable, device pointer, prev_row_pitch bytes) is the reference luma of frame
 * first_index-1 -- the one-frame halo a frame-sharded rank needs for motion.  When NULL the context
 * continues from the last frame it saw if that was first_index-1, else motion(first_index) = 0. */
PQA_API int pqa_submit_device(pqa_ctx* ctx, int64_t first_index, int32_t n_frames, const pqa_device_clip* ref,
                      const pqa_device_clip* dis, const void* prev_ref_luma, int64_t prev_row_pitch);

/* Decoder surfaces in device memory: what a hardware decoder (VCN through rocDecode / VA-API) writes and what a drop-in
 * that keeps decode on the GPU hands over instead of the planes ffmpeg would feed libvmaf (app/vmaf_analyzer.py:415-416
 * names the two inputs).  NV12: 8-bit Y plane + ONE plane of interleaved U,V pairs at half
 * resolution in both directions.  P01X: the same layout with 16-bit little-endian samples whose value sits in the UPPER
 * bit_depth bits (P010 for a 10-bit context, P012 for a 12-bit one).  Frame f's planes start at luma + f * luma_frame_pitch
 * and chroma + f * chroma_frame_pitch; pitches in BYTES.  chroma may be NULL when the context has n_planes == 1. */
enum { PQA_SURFACE_NV12 = 1, PQA_SURFACE_P01X = 2 };
typedef struct pqa_surface_clip {
  uint32_t struct_size;        /* sizeof(pqa_surface_clip) */
  uint32_t format;             /* PQA_SURFACE_* */
  const void* luma;
  const void* chroma;
  int64_t luma_row_pitch, luma_frame_pitch;
  int64_t chroma_row_pitch, chroma_frame_pitch;
} pqa_surface_clip;

/* Submit n_frames consecutive frame pairs held as decoder surfaces.  An NV12 luma plane is scored where it lies;
 * interleaved chroma is split into planes and 16-bit samples are shifted down on the way (device-side, into the
 * context's own staging; nothing crosses PCIe).  The context must be 4:2:0 (chroma shifts 1, 1) when n_planes == 3.
 * prev_ref (nullable) carries, as its frame 0, the reference frame first_index-1 -- the motion halo of a frame-sharded
 * rank; NULL continues from the last frame the context saw, as pqa_submit_device does.  Same record semantics as the
 * other submit calls. */
PQA_API int pqa_submit_surfaces(pqa_ctx* ctx, int64_t first_index, int32_t n_frames, const pqa_surface_clip* ref,
                                const pqa_surface_clip* dis, const pqa_surface_clip* prev_ref);

/* Host-memory variant of the halo for the pqa_submit path. */
PQA_API int pqa_set_motion_halo(pqa_ctx* ctx, const void* prev_ref_luma_host, int64_t row_stride);

/* Launch whatever pqa_submit has pending (partial batch). */
PQA_API int pqa_flush(pqa_ctx* ctx);

/* Copy `count` records starting at frame first_index into records[count][PQA_RECORD_DOUBLES].  Launches a pending
 * partial batch, then waits only for the batch that produced the youngest requested record (one completion event
 * per batch): batches submitted later keep running while the caller works on these records (SVM, pooling).
 * Frames that were never submitted, or whose record has been replaced, give PQA_ESTATE.  A collected record's ring
 * slot becomes free for frame index + k * result_capacity.  Replaces reading the libvmaf JSON log / stats files. */
PQA_API int pqa_collect(pqa_ctx* ctx, int64_t first_index, int32_t count, double* records);

/* Wait for all submitted work without collecting. */
PQA_API int pqa_sync(pqa_ctx* ctx);

/* Thread-safe: makes every later (and the current, between batches) submit/collect return
 * PQA_ECANCELLED.  Mirrors VMAFAnalyzer.terminate_analysis (app/vmaf_analyzer.py:139-151). */
PQA_API int pqa_cancel(pqa_ctx* ctx);

/* Clear the cancel flag and the motion continuity state (start of a new clip). */
PQA_API int pqa_reset(pqa_ctx* ctx);

/* Text of the most recent failure on this context (ctx == NULL: last pqa_create failure). */
PQA_API const char* pqa_last_error(const pqa_ctx* ctx);

/* Per-frame luma statistics of a device-resident clip: out[n_frames][3] (host) = {sum, sum of squares,
 * count(sample > threshold)}, exact integers.  mean / std / white-pixel ratio follow in float64 on the host.
 * Replaces the cv2 loops np.mean(gray) / np.std(gray) / np.sum(gray > threshold) of the reference's white
 * bookend-frame detection (app/bookend_alignment.py:796-800, 902-904, 998-1020; app/reference_analyzer.py:
 * 127-144) -- the step that runs right before analyze_videos.  n_frames <= max_batch per call is NOT required
 * (the call loops).  Synchronous. */
PQA_API int pqa_luma_stats_device(pqa_ctx* ctx, const void* luma, int64_t row_pitch, int64_t frame_pitch,
                                  int32_t n_frames, uint32_t threshold, uint64_t* out);

/* The same statistics for frames in HOST memory: luma_frames[i] points at frame i's luma plane (rows row_stride bytes
 * apart; the frames need not be contiguous -- a detector samples every k-th frame of a memory-mapped clip).  Frames
 * are packed into pinned staging and uploaded in chunks while the previous chunk's kernel runs.  out[n_frames][3] as
 * above.  This is what pqa2_amd.bookend.detect() drives.  Synchronous. */
PQA_API int pqa_luma_stats(pqa_ctx* ctx, const void* const* luma_frames, int64_t row_stride, int32_t n_frames,
                           uint32_t threshold, uint64_t* out);

/* What "gray" means to the two luma-statistics calls above.  PQA_GRAY_LUMA (default): the luma samples as they are.
 * PQA_GRAY_BT601_FULL: gray = clamp(round((Y - 16 s) * 255 / (219 s)), 0, 255), s = 2^(bit_depth - 8) -- what the
 * reference's cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) sees for a limited-range clip (cv2.VideoCapture has expanded it to
 * full-range BGR; with BT.601 on both legs the chroma terms cancel).  The statistics are then those of that 8-bit gray for
 * every bit depth, and `threshold` is in its units: the absolute constants of the reference's rules (180 / 200 / 220 / 230 /
 * 240, app/bookend_alignment.py:818-852, app/reference_analyzer.py:134) mean what they mean there. */
enum { PQA_GRAY_LUMA = 0, PQA_GRAY_BT601_FULL = 1 };
PQA_API int pqa_set_luma_gray(pqa_ctx* ctx, uint32_t mode);

/* Measurement hooks (bench.py): HIP-event timing of individual kernels on the context's stream.
 * kernel ids: 0..3 vif_stat scale s (each also produces the next scale's planes), 4..6 reserved,
 * 7..10 adm scale s,
 * 11 motion, 12 sse, 13 ssim, 14 finalize.
 * pqa_profile_enable(ctx, 0) stops, (ctx, 1) times every kernel, (ctx, mask << 1) only the kernels whose bit
 * is set in mask (event records between kernels are not free: ~10 % of a step when every kernel is timed). */
enum { PQA_PROF_KERNELS = 15 };
PQA_API int pqa_profile_enable(pqa_ctx* ctx, int on);
PQA_API int pqa_profile_read(pqa_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches, uint64_t* frames);
PQA_API const char* pqa_profile_kernel_name(int kernel_id);

/* Test hook (no device needed): the per-lane tap-matrix fragments of the scale-0 VIF kernel, [fragment][lane 0..63][8 f16 bit
 * patterns], so that a CPU test can replay both matrix passes in numpy against a direct convolution (tests/test_host.py).
 * Returns the number of fragments, or -(halfwords needed) when `out` is too small. */
PQA_API int pqa_debug_vif_march_table(uint16_t* out, int32_t capacity_halfwords);

/* Test / measurement hook (no device needed): how the scale-0 VIF kernel cuts a width x height frame into waves, and what a
 * 16 x 16 block costs on the matrix cores: out6 = {16-column stripes, 16-row blocks, blocks per segment, segments per stripe,
 * first-pass MFMAs per block, second-pass MFMAs per block}.  A wave = one stripe x one segment; a segment repeats one block
 * of the first pass.  bench.py prices `roofline.mfma` with it instead of mirroring the rule. */
PQA_API int pqa_debug_vif_march_shape(uint32_t width, uint32_t height, int32_t* out6);

#ifdef __cplusplus
}
#endif
#endif /* PQA_VMAF_H */
