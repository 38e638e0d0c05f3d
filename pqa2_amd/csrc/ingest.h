// Launchers of ingest.hip (decoder surfaces -> planes).  Kept out of kernels.h: that header is part of the source hash
// that ties profiles/kernel_counters.json to the scoring kernels (bench.py kernel_source_hash).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqa {

// 16-bit samples with the value in the upper bits (P010 / P012 luma) -> LSB-aligned u16 plane.  Pitches in bytes.
hipError_t launch_ingest_shift16(hipStream_t stream, const void* src, int64_t src_row_pitch, int64_t src_frame_pitch,
                                 void* dst, int64_t dst_row_pitch, int64_t dst_frame_pitch, int w, int h, int shift,
                                 int n_frames);
// Interleaved U,V pairs (NV12: esize 1, shift 0; P010 / P012: esize 2, shift 16 - bit depth) -> U plane and V plane of
// w x h samples each (same pitches for both).
hipError_t launch_ingest_deinterleave(hipStream_t stream, int esize, const void* src_uv, int64_t src_row_pitch,
                                      int64_t src_frame_pitch, void* dst_u, void* dst_v, int64_t dst_row_pitch,
                                      int64_t dst_frame_pitch, int w, int h, int shift, int n_frames);

}  // namespace pqa
