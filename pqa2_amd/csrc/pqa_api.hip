// C ABI of libpqa_vmaf.so: context, workspaces, batching, host staging, profiling hooks.
// Declarations and the reference interfaces each entry point replaces: include/pqa_vmaf.h.
#include "../../include/pqa_vmaf.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <unistd.h>

#include "host_pack.h"
#include "host_ring.h"
#include "ingest.h"
#include "kernels.h"

using namespace pqa;
using pqa::host::PackPool;
using pqa::host::PackTask;
using pqa::host::run_pack_task;

namespace {

thread_local std::string g_create_error;

struct Level {
  int w = 0, h = 0;
  int64_t pitch = 0, frame_pitch = 0;  // elements (float)
  float* ref = nullptr;
  float* dis = nullptr;
};

struct Half {
  uint8_t* pinned = nullptr;
  uint8_t* dev = nullptr;
  hipEvent_t copied = nullptr, computed = nullptr;
  bool copied_pending = false, computed_pending = false;
};

struct ProfEv {
  hipEvent_t a, b;
  int id, frames;
};

constexpr int kBatchEvents = 64;  // ring of per-batch completion events
constexpr int kLumaOutFrames = 2048;  // luma statistics kept on the device between host copies

// Staging buffers of the host path outlive their context: pinning 2 x 200 MB (2160p 4:2:0, 8 frames) and releasing it again
// costs ~55 ms per context -- more than scoring a 48-frame 2160p clip -- and the reference's caller makes a fresh analyzer
// per run (app/ui/tabs/analysis_tab.py:588).  One set per process is kept for the next context of the same device and
// size (PQA_STAGING_CACHE=0: off; anything else it holds is released when a differently sized set is parked).
struct StagingSet {
  int device = -1;
  size_t bytes = 0;
  uint8_t* pinned[2] = {nullptr, nullptr};
  uint8_t* dev[2] = {nullptr, nullptr};
};
std::mutex g_staging_mu;
StagingSet g_staging;   // guarded by g_staging_mu; empty when bytes == 0

bool staging_cache_enabled() {
  const char* e = getenv("PQA_STAGING_CACHE");
  return !(e && e[0] == '0');
}

void staging_release(StagingSet& s) {   // caller holds the lock or owns s; the right device must be current
  for (int i = 0; i < 2; ++i) {
    if (s.pinned[i]) hipHostFree(s.pinned[i]);
    if (s.dev[i]) hipFree(s.dev[i]);
    s.pinned[i] = s.dev[i] = nullptr;
  }
  s.bytes = 0;
  s.device = -1;
}

}  // namespace

struct pqa_ctx {
  pqa_config cfg{};
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
  hipStream_t aux[2] = {nullptr, nullptr};  // ADM and motion/PSNR/SSIM chains run beside the VIF chain
  hipEvent_t fork_ev = nullptr, join_ev[2] = {nullptr, nullptr};
  Elem elem = ELEM_U8;
  int esize = 1;
  float inv_scale = 1.0f;
  int pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0};
  int n_planes = 1;
  int B = 8, HB = 8, capacity = 16384, k_sub = 1;  // B: frames per launch; HB: frames per host-staging half
  Level vif_lv[4], adm_lv[4];
  double* vif_part[4] = {};
  long long* vif_fx_part[4] = {};  // fixed-point VIF: int64 partials instead of (num, den) doubles
  uint16_t* vif_lut = nullptr;     // integer_vif.c's log2 table, entries 32768..65535
  bool vif_fixed = false, motion_fixed = false, adm_fixed = false;
  long long* adm_fx_part[4] = {};   // fixed-point ADM: per-row int64 partials
  long long* adm_fx_acc = nullptr;  // [capacity][4][6] ring of accumulators, finished on the host in pqa_collect
  int32_t* adm_div_lut = nullptr;
  AdmFxScale adm_fx[4] = {};
  unsigned long long* motion_fx_part = nullptr;
  int vif_tiles[4] = {};
  int vif_part_cap0 = 0;   // partial pairs per frame the scale-0 buffer holds (tiled kernels or the march kernel)
  double* adm_part[4] = {};
  int adm_tiles[4] = {};
  float adm_area[4] = {};
  double* motion_part = nullptr;
  int motion_tiles_n = 0;
  unsigned long long* sse_part[3] = {};
  unsigned long long* sse_part_b[3] = {};
  unsigned long long* sse_tile_part[3] = {};
  double* ssim_part[3] = {};
  int ssim_tiles_n[3] = {};
  double ssim_norm[3] = {};
  double* records = nullptr;
  unsigned long long* luma_part = nullptr;
  unsigned long long* luma_out = nullptr;
  // host-frame luma statistics (pqa_luma_stats): two pinned + two device halves of LB luma planes
  uint8_t* luma_pinned[2] = {nullptr, nullptr};
  uint8_t* luma_dev[2] = {nullptr, nullptr};
  hipEvent_t luma_copied[2] = {nullptr, nullptr};
  int64_t luma_pitch = 0;
  int LB = 0;
  bool luma_ready = false;
  uint32_t luma_gray = PQA_GRAY_LUMA;
  // motion continuity
  uint8_t* last_luma = nullptr;
  int64_t last_luma_pitch = 0;  // bytes
  int64_t last_index = -1;
  bool have_last = false, halo_armed = false;
  // host staging (pqa_submit path)
  Half half[2];
  int cur_half = 0, pending = 0;
  int64_t pending_first = 0;
  size_t slot_bytes = 0;
  size_t plane_off[2][3] = {};
  int64_t slot_row_pitch[3] = {};
  bool staging_ready = false;
  std::thread pin_thread;            // pins the second staging half while the first one fills (ensure_staging)
  hipError_t pin_err = hipSuccess;   // its result; read after joining it (staging_half_ready)
  uint8_t* surf_dev = nullptr;   // pqa_submit_surfaces: B slots of unpacked planes (device only, allocated on first use)
  std::unique_ptr<PackPool> pack_pool;
  bool pack_pool_tried = false;
  std::vector<PackTask> pack_tasks;
  // record-ring bookkeeping (host side): which frame a slot holds, whether it was collected, and the batch that
  // writes it.  Lets pqa_collect wait for ITS batch only and makes the PQA_ESTATE promises of the header real.
  host::RecordRing ring;             // host_ring.h
  hipEvent_t batch_ev[kBatchEvents] = {};
  uint64_t batch_ev_seq[kBatchEvents] = {};  // sequence number last recorded into each event
  uint64_t batch_seq = 0;            // batches launched so far (the next batch gets batch_seq + 1)
  uint64_t done_seq = 0;             // every batch <= done_seq is known to be complete
  std::atomic<int> cancelled{0};
  std::string err;
  std::vector<void*> allocs;
  // profiling
  int multi_stream = 0;              // 0 one stream; 1 three streams from the start of a batch; 2 three streams behind VIF scale 0
  int vif_s0_mode = VIF_S0_AUTO;   // PQA_VIF_MFMA, read once in pqa_create
  int adm_mode = ADM_AUTO;         // PQA_ADM_MARCH, read once in pqa_create
  int motion_mode = MOTION_AUTO;   // PQA_MOTION_MARCH, read once in pqa_create
  bool trace = false;   // PQA_TRACE=1: synchronise after every launch and name it on stderr (localises a stall)
  bool prof = false;
  uint32_t prof_mask = 0xffffffffu;
  std::vector<ProfEv> evs;
  double prof_ms[PQA_PROF_KERNELS] = {};
  uint64_t prof_n[PQA_PROF_KERNELS] = {}, prof_frames[PQA_PROF_KERNELS] = {};
};

namespace {

int fail(pqa_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail((c), e_ == hipErrorOutOfMemory ? PQA_ENOMEM : PQA_EDEVICE, "%s failed: %s", #expr, \
                  hipGetErrorString(e_));                                                        \
  } while (0)

template <typename T>
int dev_alloc(pqa_ctx* c, T** out, size_t count) {
  void* p = nullptr;
  if (count == 0) count = 1;
  HIPCHK(c, hipMalloc(&p, count * sizeof(T)));
  c->allocs.push_back(p);
  *out = (T*)p;
  return PQA_OK;
}

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

static const char* kProfNames[PQA_PROF_KERNELS] = {
    "vif_stat_s0", "vif_stat_s1", "vif_stat_s2", "vif_stat_s3", "reserved4", "reserved5",
    "reserved6", "adm_scale_s0", "adm_scale_s1", "adm_scale_s2", "adm_scale_s3", "motion", "sse",
    "ssim", "finalize"};

struct ProfScope {
  pqa_ctx* c;
  ProfEv ev{};
  bool on;
  hipStream_t st;
  ProfScope(pqa_ctx* ctx, int id, int frames, hipStream_t stream)
      : c(ctx), on(ctx->prof && ((ctx->prof_mask >> id) & 1u)), st(stream), ev_id(id), ev_frames(frames) {
    if (c->trace) {
      fprintf(stderr, "[pqa trace] %s queued\n", kProfNames[id]);
      fflush(stderr);
    }
    if (!on) return;
    ev.id = id; ev.frames = frames;
    if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
    hipEventRecord(ev.a, st);
  }
  ~ProfScope() {
    if (c->trace) {   // the launches of this scope are queued: wait for them and say so, flushed
      const hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, "[pqa trace] %s done (%d frames)%s%s\n", kProfNames[ev_id], ev_frames, e == hipSuccess ? "" : ": ",
              e == hipSuccess ? "" : hipGetErrorString(e));
      fflush(stderr);
    }
    if (!on) return;
    hipEventRecord(ev.b, st);
    c->evs.push_back(ev);
  }
  int ev_id = 0, ev_frames = 0;
};

void prof_drain(pqa_ctx* c) {
  for (auto& e : c->evs) {
    float ms = 0.0f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->prof_ms[e.id] += ms;
      c->prof_n[e.id] += 1;
      c->prof_frames[e.id] += e.frames;
    }
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  c->evs.clear();
}

// ---- record-ring bookkeeping -----------------------------------------------------------------------
// A slot may be rewritten by the SAME frame index (a re-run) or once its record has been collected; a different
// frame landing on an uncollected record would lose it silently, so that is a call-sequence error.
int check_slots_free(pqa_ctx* c, int64_t first, int n) {
  int64_t f = 0, holder = 0;
  if (!c->ring.can_submit(first, n, &f, &holder))
    return fail(c, PQA_ESTATE,
                "frame %lld would overwrite the uncollected record of frame %lld (result_capacity %d): "
                "collect it first or create the context with a larger result_capacity",
                (long long)f, (long long)holder, c->capacity);
  return PQA_OK;
}

void claim_slots(pqa_ctx* c, int64_t first, int n, uint64_t seq) { c->ring.claim(first, n, seq); }

// Wait until batch `seq` (and, the stream being in order, every earlier one) has written its records.
int wait_batch(pqa_ctx* c, uint64_t seq) {
  if (seq <= c->done_seq) return PQA_OK;
  const int e = (int)(seq % kBatchEvents);
  // the event may meanwhile carry a later batch of the same stream: waiting for that one is still correct
  HIPCHK(c, hipEventSynchronize(c->batch_ev[e]));
  c->done_seq = c->batch_ev_seq[e] > seq ? c->batch_ev_seq[e] : seq;
  return PQA_OK;
}

// ---- the batch: every kernel for n consecutive device-resident frames -------------------------
int process_batch(pqa_ctx* c, int64_t first, int n, const pqa_device_clip* ref, const pqa_device_clip* dis,
                  const void* prev, int64_t prev_pitch_bytes) {
  const uint32_t feat = c->cfg.features;
  const int es = c->esize;
  for (int p = 0; p < c->n_planes; ++p) {
    if (ref->row_pitch[p] % es || dis->row_pitch[p] % es || ref->frame_pitch[p] % es || dis->frame_pitch[p] % es)
      return fail(c, PQA_EINVAL, "plane %d pitch is not a multiple of the sample size", p);
    if (!ref->plane[p] || !dis->plane[p]) return fail(c, PQA_EINVAL, "plane %d pointer is null", p);
  }
  {
    const int rc = check_slots_free(c, first, n);
    if (rc != PQA_OK) return rc;
  }
  const PlaneRun rY{ref->plane[0], ref->row_pitch[0] / es, ref->frame_pitch[0] / es};
  const PlaneRun dY{dis->plane[0], dis->row_pitch[0] / es, dis->frame_pitch[0] / es};
  const int w = c->pw[0], h = c->ph[0];
  hipStream_t st = c->stream;
  // fork: the ADM chain and the motion/PSNR/SSIM kernels are independent of the VIF chain; on their own
  // streams they fill the SIMD time the VALU-bound VIF kernels leave while waiting, and the small deep-scale
  // grids overlap instead of running one after another
  // (with EVERY kernel event-timed -- the breakdown pass of bench.py -- the chains stay on one stream, so that a kernel's
  // figure is its own and not its share of a crowded device; a subset mask, as in the timed region, changes nothing)
  const bool multi = c->multi_stream && !(c->prof && c->prof_mask == 0xffffffffu);
  // mode 2: the other chains start behind VIF scale 0 (the dominant launch runs alone, its figures stay its own) and
  // overlap VIF scales 1-3 only; the fork event is then recorded right after that launch
  hipStream_t st_adm = multi ? c->aux[0] : st, st_misc = multi ? c->aux[1] : st;

  // frames that get spatial features (libvmaf n_subsample: index % k == 0)
  const int k = c->k_sub;
  int e0 = 0, sp_n = n;
  if (k > 1) {
    e0 = (int)((k - first % k) % k);
    sp_n = e0 < n ? (n - e0 + k - 1) / k : 0;
  }
  const bool fork_late = multi && c->multi_stream == 2 && (feat & PQA_FEAT_VIF) && !c->vif_fixed && sp_n > 0;
  if (multi && !fork_late) {
    HIPCHK(c, hipEventRecord(c->fork_ev, st));
    HIPCHK(c, hipStreamWaitEvent(st_adm, c->fork_ev, 0));
    HIPCHK(c, hipStreamWaitEvent(st_misc, c->fork_ev, 0));
  }
  const auto sub = [&](PlaneRun r) {
    return PlaneRun{(const uint8_t*)r.base + (int64_t)e0 * r.frame_pitch * es, r.row_pitch, r.frame_pitch * k};
  };
  const PlaneRun rYs = k > 1 ? sub(rY) : rY, dYs = k > 1 ? sub(dY) : dY;

  // motion: reference luma in front of the batch (caller's halo, the armed halo, or the last frame of the previous batch)
  const void* p0 = prev;
  int64_t p0_pitch = prev_pitch_bytes;
  if (feat & PQA_FEAT_MOTION) {
    if (!p0 && (c->halo_armed || (c->have_last && c->last_index == first - 1))) {
      p0 = c->last_luma;
      p0_pitch = c->last_luma_pitch;
    }
    if (p0 && p0_pitch % es) return fail(c, PQA_EINVAL, "halo pitch is not a multiple of the sample size");
  }
  int vif_np[4] = {c->vif_tiles[0], c->vif_tiles[1], c->vif_tiles[2], c->vif_tiles[3]};   // partial pairs written per frame
  int adm_np[4] = {c->adm_tiles[0], c->adm_tiles[1], c->adm_tiles[2], c->adm_tiles[3]};   // partial sextets written per frame
  int motion_np = c->motion_tiles_n;                                                       // motion partials written per frame

  if ((feat & PQA_FEAT_VIF) && sp_n > 0 && c->vif_fixed) {
    PlaneRun cr = rYs, cd = dYs;
    Elem ce = c->elem;
    int cw = w, ch = h;
    for (int s = 0; s < 4; ++s) {
      MutPlaneRun nr{nullptr, 0, 0}, nd{nullptr, 0, 0};
      if (s < 3) {  // the pyramid buffers hold u16 planes in this mode (same element pitches, half the bytes)
        Level& L = c->vif_lv[s + 1];
        nr = MutPlaneRun{L.ref, L.pitch, L.frame_pitch};
        nd = MutPlaneRun{L.dis, L.pitch, L.frame_pitch};
      }
      {
        ProfScope ps(c, s, sp_n, st);
        HIPCHK(c, launch_vif_fixed(st, s, (int)c->cfg.bit_depth, ce, cr, cd, sp_n, cw, ch, c->cfg.vif_enhn_gain_limit,
                                   c->vif_lut, c->vif_fx_part[s], nr, nd));
      }
      if (s < 3) {
        Level& L = c->vif_lv[s + 1];
        cr = PlaneRun{L.ref, L.pitch, L.frame_pitch};
        cd = PlaneRun{L.dis, L.pitch, L.frame_pitch};
        ce = ELEM_U16;
        cw = L.w; ch = L.h;
      }
    }
  } else if ((feat & PQA_FEAT_VIF) && sp_n > 0) {
    PlaneRun cr = rYs, cd = dYs;
    Elem ce = c->elem;
    int cw = w, ch = h;
    for (int s = 0; s < 4; ++s) {
      MutPlaneRun nr{nullptr, 0, 0}, nd{nullptr, 0, 0};
      if (s < 3) {
        Level& L = c->vif_lv[s + 1];
        nr = MutPlaneRun{L.ref, L.pitch, L.frame_pitch};
        nd = MutPlaneRun{L.dis, L.pitch, L.frame_pitch};
      }
      {
        ProfScope ps(c, s, sp_n, st);
        HIPCHK(c, launch_vif_stat(st, s, ce, cr, cd, sp_n, cw, ch, c->inv_scale,
                                  (float)c->cfg.vif_enhn_gain_limit, c->cfg.vif_border == PQA_VIF_BORDER_INTEGER,
                                  c->vif_part[s], nr, nd, c->vif_s0_mode, &vif_np[s]));
      }
      if (s == 0 && fork_late) {
        HIPCHK(c, hipEventRecord(c->fork_ev, st));
        HIPCHK(c, hipStreamWaitEvent(st_adm, c->fork_ev, 0));
        HIPCHK(c, hipStreamWaitEvent(st_misc, c->fork_ev, 0));
      }
      if (s < 3) {
        Level& L = c->vif_lv[s + 1];
        cr = PlaneRun{L.ref, L.pitch, L.frame_pitch};
        cd = PlaneRun{L.dis, L.pitch, L.frame_pitch};
        ce = ELEM_F32;
        cw = L.w; ch = L.h;
      }
    }
  }
  if ((feat & PQA_FEAT_ADM) && sp_n > 0 && c->adm_fixed) {
    PlaneRun cr = rYs, cd = dYs;
    Elem ce = c->elem;
    int cw = w, ch = h;
    for (int s = 0; s < 4; ++s) {
      MutPlaneRun lr{nullptr, 0, 0}, ld{nullptr, 0, 0};
      if (s < 3) {  // the approximation-band buffers hold int32 planes in this mode (same 4-byte elements)
        Level& L = c->adm_lv[s + 1];
        lr = MutPlaneRun{L.ref, L.pitch, L.frame_pitch};
        ld = MutPlaneRun{L.dis, L.pitch, L.frame_pitch};
      }
      {
        ProfScope ps(c, 7 + s, sp_n, st_adm);
        HIPCHK(c, launch_adm_fixed(st_adm, s, (int)c->cfg.bit_depth, ce, cr, cd, sp_n, cw, ch,
                                   c->cfg.adm_enhn_gain_limit, c->adm_div_lut, lr, ld, c->adm_fx_part[s]));
      }
      if (s < 3) {
        Level& L = c->adm_lv[s + 1];
        cr = PlaneRun{L.ref, L.pitch, L.frame_pitch};
        cd = PlaneRun{L.dis, L.pitch, L.frame_pitch};
        ce = ELEM_F32;
        cw = L.w; ch = L.h;
      }
    }
  } else if ((feat & PQA_FEAT_ADM) && sp_n > 0) {
    PlaneRun cr = rYs, cd = dYs;
    Elem ce = c->elem;
    int cw = w, ch = h;
    int s = 0;
    if (c->adm_mode == ADM_AUTO) {   // scales 0 and 1 in one launch when the geometry allows (adm_pyramid.hip)
      Level& L2 = c->adm_lv[2];
      hipError_t perr = hipSuccess;
      int np = 0;
      ProfScope ps(c, 7, sp_n, st_adm);
      if (launch_adm_pyramid(st_adm, ce, cr, cd, sp_n, cw, ch, c->inv_scale, (float)c->cfg.adm_enhn_gain_limit,
                             MutPlaneRun{L2.ref, L2.pitch, L2.frame_pitch}, MutPlaneRun{L2.dis, L2.pitch, L2.frame_pitch},
                             c->adm_part[0], c->adm_part[1], &np, &perr)) {
        HIPCHK(c, perr);
        adm_np[0] = adm_np[1] = np;
        cr = PlaneRun{L2.ref, L2.pitch, L2.frame_pitch};
        cd = PlaneRun{L2.dis, L2.pitch, L2.frame_pitch};
        ce = ELEM_F32;
        cw = L2.w; ch = L2.h;
        s = 2;
      }
    }
    for (; s < 4; ++s) {
      MutPlaneRun lr{nullptr, 0, 0}, ld{nullptr, 0, 0};
      if (s < 3) {
        Level& L = c->adm_lv[s + 1];
        lr = MutPlaneRun{L.ref, L.pitch, L.frame_pitch};
        ld = MutPlaneRun{L.dis, L.pitch, L.frame_pitch};
      }
      {
        ProfScope ps(c, 7 + s, sp_n, st_adm);
        HIPCHK(c, launch_adm_scale(st_adm, s, ce, cr, cd, sp_n, cw, ch, c->inv_scale,
                                   (float)c->cfg.adm_enhn_gain_limit, lr, ld, c->adm_part[s], c->adm_mode, &adm_np[s]));
      }
      if (s < 3) {
        Level& L = c->adm_lv[s + 1];
        cr = PlaneRun{L.ref, L.pitch, L.frame_pitch};
        cd = PlaneRun{L.dis, L.pitch, L.frame_pitch};
        ce = ELEM_F32;
        cw = L.w; ch = L.h;
      }
    }
  }
  if (feat & PQA_FEAT_MOTION) {
    ProfScope ps(c, 11, n, st_misc);
    if (c->motion_fixed)
      HIPCHK(c, launch_motion_fixed(st_misc, (int)c->cfg.bit_depth, c->elem, rY, p0, p0 ? p0_pitch / es : 0, n, w, h,
                                    c->motion_fx_part));
    else
      HIPCHK(c, launch_motion(st_misc, c->elem, rY, p0, p0 ? p0_pitch / es : 0, n, w, h, c->inv_scale, c->motion_part,
                              c->motion_mode, &motion_np));
  }
  int n_sse = 0, n_ssim = 0;
  bool sse_a[3] = {false, false, false}, sse_b[3] = {false, false, false}, sse_t[3] = {false, false, false};
  const bool both = (feat & PQA_FEAT_PSNR) && (feat & PQA_FEAT_SSIM);
  if (feat & PQA_FEAT_PSNR) {
    n_sse = c->n_planes;
    ProfScope ps(c, 12, n, st_misc);
    for (int p = 0; p < c->n_planes; ++p) {
      const PlaneRun a{dis->plane[p], dis->row_pitch[p] / es, dis->frame_pitch[p] / es};
      const PlaneRun b{ref->plane[p], ref->row_pitch[p] / es, ref->frame_pitch[p] / es};
      const int pw = c->pw[p], ph = c->ph[p];
      if (both && c->ssim_tiles_n[p] > 0) {
        // the SSIM kernel delivers the squared error of the 4-aligned part; only remainder strips are left
        sse_t[p] = true;
        const int w4 = (pw >> 2) << 2, h4 = (ph >> 2) << 2;
        if (w4 < pw) {
          const PlaneRun ar{(const uint8_t*)a.base + (int64_t)w4 * es, a.row_pitch, a.frame_pitch};
          const PlaneRun br{(const uint8_t*)b.base + (int64_t)w4 * es, b.row_pitch, b.frame_pitch};
          HIPCHK(c, launch_sse(st_misc, c->elem, ar, br, n, pw - w4, ph, c->sse_part[p]));
          sse_a[p] = true;
        }
        if (h4 < ph) {
          const PlaneRun ab{(const uint8_t*)a.base + (int64_t)h4 * a.row_pitch * es, a.row_pitch, a.frame_pitch};
          const PlaneRun bb{(const uint8_t*)b.base + (int64_t)h4 * b.row_pitch * es, b.row_pitch, b.frame_pitch};
          HIPCHK(c, launch_sse(st_misc, c->elem, ab, bb, n, w4, ph - h4, c->sse_part_b[p]));
          sse_b[p] = true;
        }
      } else {
        HIPCHK(c, launch_sse(st_misc, c->elem, a, b, n, pw, ph, c->sse_part[p]));
        sse_a[p] = true;
      }
    }
  }
  if (feat & PQA_FEAT_SSIM) {
    n_ssim = c->n_planes;
    ProfScope ps(c, 13, n, st_misc);
    for (int p = 0; p < c->n_planes; ++p) {
      const PlaneRun a{dis->plane[p], dis->row_pitch[p] / es, dis->frame_pitch[p] / es};
      const PlaneRun b{ref->plane[p], ref->row_pitch[p] / es, ref->frame_pitch[p] / es};
      HIPCHK(c, launch_ssim(st_misc, c->elem, a, b, n, c->pw[p], c->ph[p], (1 << c->cfg.bit_depth) - 1, c->ssim_part[p],
                            sse_t[p] ? c->sse_tile_part[p] : nullptr));
    }
  }

  if (multi) {  // join
    HIPCHK(c, hipEventRecord(c->join_ev[0], st_adm));
    HIPCHK(c, hipEventRecord(c->join_ev[1], st_misc));
    HIPCHK(c, hipStreamWaitEvent(st, c->join_ev[0], 0));
    HIPCHK(c, hipStreamWaitEvent(st, c->join_ev[1], 0));
  }

  FinalizeArgs fa{};
  for (int s = 0; s < 4; ++s) {
    fa.vif_part[s] = c->vif_part[s]; fa.vif_tiles[s] = c->vif_fixed ? c->vif_tiles[s] : vif_np[s];
    fa.vif_fx_part[s] = c->vif_fixed ? c->vif_fx_part[s] : nullptr;
    fa.adm_part[s] = c->adm_part[s]; fa.adm_tiles[s] = c->adm_fixed ? c->adm_tiles[s] : adm_np[s]; fa.adm_area[s] = c->adm_area[s];
    fa.adm_fx_part[s] = c->adm_fixed ? c->adm_fx_part[s] : nullptr;
    fa.adm_fx_tiles_x[s] = adm_tiles_x(c->adm_fx[s].band_w);
    fa.adm_fx_top[s] = c->adm_fx[s].top; fa.adm_fx_bottom[s] = c->adm_fx[s].bottom;
    fa.adm_fx_num_shift[s] = c->adm_fx[s].num_row_shift; fa.adm_fx_den_shift[s] = c->adm_fx[s].den_row_shift;
  }
  fa.motion_part = c->motion_part;
  fa.motion_tiles = c->motion_fixed ? c->motion_tiles_n : motion_np;
  fa.motion_norm = (double)c->inv_scale / ((double)w * h);
  fa.motion_fx_part = c->motion_fixed ? c->motion_fx_part : nullptr;
  fa.motion_wh = (unsigned)w * (unsigned)h;
  for (int p = 0; p < 3; ++p) {
    fa.sse_part[p] = c->sse_part[p];
    fa.sse_use_a[p] = sse_a[p] ? 1 : 0;
    fa.sse_part_b[p] = sse_b[p] ? c->sse_part_b[p] : nullptr;
    fa.sse_tile_part[p] = sse_t[p] ? c->sse_tile_part[p] : nullptr;
    fa.ssim_part[p] = c->ssim_part[p]; fa.ssim_tiles[p] = c->ssim_tiles_n[p]; fa.ssim_norm[p] = c->ssim_norm[p];
  }
  fa.adm_fx_acc = c->adm_fx_acc;
  fa.records = c->records;
  fa.record_stride = PQA_RECORD_DOUBLES;
  fa.capacity = c->capacity;
  {
    ProfScope ps(c, 14, n, st);
    if (k > 1) {
      if (sp_n > 0 && (feat & (PQA_FEAT_VIF | PQA_FEAT_ADM))) {
        FinalizeArgs f1 = fa;
        f1.n_frames = sp_n;
        f1.has_vif = !!(feat & PQA_FEAT_VIF); f1.has_adm = !!(feat & PQA_FEAT_ADM);
        f1.slot_base = (int)((first + e0) % c->capacity); f1.slot_step = k;
        HIPCHK(c, launch_finalize(st, f1));
      }
      FinalizeArgs f2 = fa;
      f2.n_frames = n;
      f2.has_motion = !!(feat & PQA_FEAT_MOTION); f2.n_sse_planes = n_sse; f2.n_ssim_planes = n_ssim;
      f2.slot_base = (int)(first % c->capacity); f2.slot_step = 1;
      HIPCHK(c, launch_finalize(st, f2));
    } else {
      fa.n_frames = n;
      fa.has_vif = !!(feat & PQA_FEAT_VIF); fa.has_adm = !!(feat & PQA_FEAT_ADM);
      fa.has_motion = !!(feat & PQA_FEAT_MOTION); fa.n_sse_planes = n_sse; fa.n_ssim_planes = n_ssim;
      fa.slot_base = (int)(first % c->capacity); fa.slot_step = 1;
      HIPCHK(c, launch_finalize(st, fa));
    }
  }
  {  // the records of this batch are complete here: one event per batch, so pqa_collect waits for ITS batch only
    const uint64_t seq = c->batch_seq + 1;
    const int e = (int)(seq % kBatchEvents);
    HIPCHK(c, hipEventRecord(c->batch_ev[e], st));
    c->batch_ev_seq[e] = seq;
    c->batch_seq = seq;
    claim_slots(c, first, n, seq);
  }
  if (feat & PQA_FEAT_MOTION) {
    // keep the last reference luma so the next batch continues the motion chain
    const uint8_t* src = (const uint8_t*)ref->plane[0] + (int64_t)(n - 1) * ref->frame_pitch[0];
    HIPCHK(c, hipMemcpy2DAsync(c->last_luma, c->last_luma_pitch, src, ref->row_pitch[0], (size_t)w * es, h,
                               hipMemcpyDeviceToDevice, st));
    c->have_last = true;
    c->halo_armed = false;
    c->last_index = first + n - 1;
  }
  return PQA_OK;
}

// One frame pair as the library lays it out itself (pinned staging, its device copy, unpacked decoder surfaces): ref
// planes then dis planes, rows padded to 64 bytes, planes to 256.
void slot_layout(pqa_ctx* c) {
  size_t off = 0;
  for (int side = 0; side < 2; ++side)
    for (int p = 0; p < c->n_planes; ++p) {
      c->slot_row_pitch[p] = round_up((int64_t)c->pw[p] * c->esize, 64);
      c->plane_off[side][p] = off;
      off += round_up(c->slot_row_pitch[p] * c->ph[p], 256);
    }
  c->slot_bytes = off;
}

int ensure_staging(pqa_ctx* c) {
  if (c->staging_ready) return PQA_OK;
  slot_layout(c);
  const size_t half_bytes = c->slot_bytes * c->HB;
  bool reused = false;
  if (staging_cache_enabled()) {
    std::lock_guard<std::mutex> g(g_staging_mu);
    if (g_staging.bytes == half_bytes && g_staging.device == c->device) {
      for (int i = 0; i < 2; ++i) {
        c->half[i].pinned = g_staging.pinned[i];
        c->half[i].dev = g_staging.dev[i];
        g_staging.pinned[i] = g_staging.dev[i] = nullptr;
      }
      g_staging.bytes = 0;
      g_staging.device = -1;
      reused = true;
    }
  }
  for (int i = 0; i < 2; ++i) {
    Half& H = c->half[i];
    HIPCHK(c, hipEventCreateWithFlags(&H.copied, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&H.computed, hipEventDisableTiming));
  }
  HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!reused) {
    // Pinning costs ~40 us per MiB (a 2160p 4:2:0 half of 8 frame pairs: 8 ms): only the first half is needed before
    // the first frame can be packed; the second is pinned by a helper thread while the first one fills.
    HIPCHK(c, hipHostMalloc((void**)&c->half[0].pinned, half_bytes, hipHostMallocDefault));
    HIPCHK(c, hipMalloc((void**)&c->half[0].dev, half_bytes));
    const auto pin_second = [c, half_bytes] {
      hipError_t e = hipSetDevice(c->device);
      if (e == hipSuccess) e = hipHostMalloc((void**)&c->half[1].pinned, half_bytes, hipHostMallocDefault);
      if (e == hipSuccess) e = hipMalloc((void**)&c->half[1].dev, half_bytes);
      c->pin_err = e;
    };
    try {
      c->pin_thread = std::thread(pin_second);
    } catch (...) {   // no thread to be had: pin it here
      pin_second();
    }
  }
  c->staging_ready = true;
  return PQA_OK;
}

// Before half i is touched: the helper thread that pins the second half has to be done with it (joining it also makes
// its writes to half[1] visible here).
int staging_half_ready(pqa_ctx* c, int i) {
  if (i == 0) return PQA_OK;
  if (c->pin_thread.joinable()) c->pin_thread.join();
  if (c->pin_err != hipSuccess)
    return fail(c, c->pin_err == hipErrorOutOfMemory ? PQA_ENOMEM : PQA_EDEVICE, "pinning the second staging half failed: %s",
                hipGetErrorString(c->pin_err));
  return PQA_OK;
}

int flush_pending(pqa_ctx* c) {
  if (c->pending == 0) return PQA_OK;
  Half& H = c->half[c->cur_half];
  HIPCHK(c, hipEventRecord(H.copied, c->copy_stream));
  H.copied_pending = true;
  HIPCHK(c, hipStreamWaitEvent(c->stream, H.copied, 0));
  pqa_device_clip r{}, d{};
  for (int p = 0; p < c->n_planes; ++p) {
    r.plane[p] = H.dev + c->plane_off[0][p];
    d.plane[p] = H.dev + c->plane_off[1][p];
    r.row_pitch[p] = d.row_pitch[p] = c->slot_row_pitch[p];
    r.frame_pitch[p] = d.frame_pitch[p] = (int64_t)c->slot_bytes;
  }
  const int n = c->pending;
  int rc = check_slots_free(c, c->pending_first, n);
  if (rc != PQA_OK) return rc;   // PQA_ESTATE: nothing launched, the packed frames stay pending (collect, then flush again)
  c->pending = 0;
  rc = process_batch(c, c->pending_first, n, &r, &d, nullptr, 0);
  if (rc != PQA_OK) return rc;
  HIPCHK(c, hipEventRecord(H.computed, c->stream));
  H.computed_pending = true;
  c->cur_half ^= 1;
  return PQA_OK;
}

void copy_plane_rows(uint8_t* dst, int64_t dst_pitch, const uint8_t* src, int64_t src_pitch, size_t row_bytes, int h) {
  if (dst_pitch == src_pitch) {
    memcpy(dst, src, (size_t)dst_pitch * (h - 1) + row_bytes);
    return;
  }
  for (int y = 0; y < h; ++y) memcpy(dst + (int64_t)y * dst_pitch, src + (int64_t)y * src_pitch, row_bytes);
}

// One frame pair from memory planes or from files into the pinned staging slot, its upload queued on the copy stream.
struct PlaneSrc {
  const uint8_t* ptr;   // memory source (fd < 0): rows `stride` bytes apart
  int64_t stride;
  int fd;               // file source: rows `stride` bytes apart from `off`
  int64_t off;
};

// n consecutive frame pairs (frame k's planes lie k * frame_step[side] bytes after src's; memory sources: n == 1) into
// staging slots and onto the copy stream.  A run is cut into chunks that fit the current staging half; within a chunk the
// helpers pack frame k + 1 while the caller queues the upload of frame k (PackPool::run_frames): one fork / join per
// chunk, the link busy from the first completed frame on.
int submit_run(pqa_ctx* c, int64_t first_index, int n_frames, const PlaneSrc (&src)[2][3], const int64_t (&frame_step)[2]) {
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = PQA_OK;
  int done = 0;
  while (done < n_frames) {
    const int64_t frame_index = first_index + done;
    if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
    if (c->pending > 0 && frame_index != c->pending_first + c->pending) {
      rc = flush_pending(c);  // non-consecutive index starts a new run: the pending frames claim their slots first
      if (rc != PQA_OK) return rc;
    }
    rc = ensure_staging(c);
    if (rc != PQA_OK) return rc;
    int m = n_frames - done;
    if (m > c->HB - c->pending) m = c->HB - c->pending;
    if (m > PackPool::kMaxFrames) m = PackPool::kMaxFrames;
    // before anything is packed (the caller can collect and retry): against the claimed slots, and against the frames still
    // pending in this run (consecutive indices: they collide only when the run is as long as the ring)
    rc = check_slots_free(c, frame_index, m);
    if (rc != PQA_OK) return rc;
    if (c->pending + m > c->capacity)
      return fail(c, PQA_ESTATE, "more pending frames than result_capacity %d", c->capacity);
    rc = staging_half_ready(c, c->cur_half);   // the second half is pinned by a helper thread while the first one fills
    if (rc != PQA_OK) return rc;
    Half& H = c->half[c->cur_half];
    if (c->pending == 0) {
      c->pending_first = frame_index;
      if (H.copied_pending) {  // pinned half still feeding an earlier upload?
        HIPCHK(c, hipEventSynchronize(H.copied));
        H.copied_pending = false;
      }
      if (H.computed_pending) {  // device half still read by an earlier batch?
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, H.computed, 0));
        H.computed_pending = false;
      }
    }
    const int slot0 = c->pending;
    // Pack into the pinned slots in ~1 MiB tasks (a 2160p 4:2:0 pair is 25 MB: one core's memcpy / pread, not PCIe, would
    // bound this path), drained by the caller and a few persistent helpers.
    bool ok = true;
    hipError_t copy_err = hipSuccess;
    const auto upload = [&](int k) {   // frame k of the chunk is complete in its pinned slot: queue its copy
      const size_t off = (size_t)(slot0 + k) * c->slot_bytes;
      copy_err = hipMemcpyAsync(H.dev + off, H.pinned + off, c->slot_bytes, hipMemcpyHostToDevice, c->copy_stream);
      return copy_err == hipSuccess;
    };
    try {  // no exception may cross the C ABI: if the task list or the helpers cannot be made, that is PQA_ENOMEM below
      c->pack_tasks.clear();
      static const unsigned task_bytes = [] { const char* e = getenv("PQA_PACK_TASK_KB"); const int kb = e ? atoi(e) : 0; return (unsigned)(kb > 0 ? kb : 1024) << 10; }();
      for (int k = 0; k < m; ++k) {
        uint8_t* slot = H.pinned + (size_t)(slot0 + k) * c->slot_bytes;
        for (int side = 0; side < 2; ++side)
          for (int p = 0; p < c->n_planes; ++p) {
            const PlaneSrc& ps = src[side][p];
            const int64_t step = (int64_t)(done + k) * frame_step[side];
            const size_t row_bytes = (size_t)c->pw[p] * c->esize;
            int rows_per = (int)(task_bytes / (row_bytes ? row_bytes : 1));
            if (rows_per < 1) rows_per = 1;
            for (int y = 0; y < c->ph[p]; y += rows_per) {
              const int rows = c->ph[p] - y < rows_per ? c->ph[p] - y : rows_per;
              PackTask t{slot + c->plane_off[side][p] + (int64_t)y * c->slot_row_pitch[p],
                         ps.fd < 0 ? ps.ptr + step + (int64_t)y * ps.stride : nullptr, c->slot_row_pitch[p], ps.stride, row_bytes, rows};
              t.fd = ps.fd;
              t.file_off = ps.off + step + (int64_t)y * ps.stride;
              t.frame = k;
              c->pack_tasks.push_back(t);
            }
          }
      }
      if (c->slot_bytes >= (4u << 20) && !c->pack_pool_tried) {
        c->pack_pool_tried = true;
        // threads packing a frame, the caller included.  Default 8: plain host buffers reach the PCIe ceiling with 4, frames
        // read from files (pread out of the page cache) or out of a memory-mapped file scale further (2160p through
        // analyze_videos: 880 frames/s with 4, 1 370 with 8; beyond that the box's noise decides, profiles/r04o_pack_threads.txt)
        const char* e = getenv("PQA_PACK_THREADS");
        int n = e ? atoi(e) : 8;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw) n = hw;
        if (n > 1) {
          try { c->pack_pool.reset(new PackPool(n - 1)); } catch (...) { c->pack_pool.reset(); }   // serial packing then
        }
      }
      if (c->pack_pool && c->slot_bytes >= (4u << 20)) {
        ok = c->pack_pool->run_frames(c->pack_tasks.data(), (int)c->pack_tasks.size(), m, upload);
      } else {
        int k_done = 0;
        for (size_t i = 0; i < c->pack_tasks.size() && ok; ++i) {
          ok = run_pack_task(c->pack_tasks[i]);
          if (ok && (i + 1 == c->pack_tasks.size() || c->pack_tasks[i + 1].frame != c->pack_tasks[i].frame)) ok = upload(k_done++);
        }
      }
    } catch (...) {
      return fail(c, PQA_ENOMEM, "out of host memory while staging frame %lld", (long long)frame_index);
    }
    if (copy_err != hipSuccess) return fail(c, PQA_EDEVICE, "hipMemcpyAsync failed: %s", hipGetErrorString(copy_err));
    if (!ok)   // nothing of this chunk becomes pending: its slots are simply packed again by the next submit
      return fail(c, PQA_EINVAL, "frame %lld%s: short read from a file source (truncated file or bad plane offset)",
                  (long long)frame_index, m > 1 ? " (or one of the frames after it in this run)" : "");
    c->pending += m;
    done += m;
    if (c->pending == c->HB) {
      rc = flush_pending(c);
      if (rc != PQA_OK) return rc;
    }
  }
  return PQA_OK;
}

int submit_one(pqa_ctx* c, int64_t frame_index, const PlaneSrc (&src)[2][3]) {
  const int64_t no_step[2] = {0, 0};
  return submit_run(c, frame_index, 1, src, no_step);
}

}  // namespace

// ================================================================================================
extern "C" {

const char* pqa_version(void) { return "pqa_vmaf 0.2.0 (gfx950; libvmaf-float VIF/ADM/motion, FFmpeg psnr/ssim; VIF scale 0 on the f16 matrix cores)"; }
int pqa_record_doubles(void) { return PQA_RECORD_DOUBLES; }

void pqa_config_init(pqa_config* cfg, uint32_t width, uint32_t height) {
  if (!cfg) return;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof *cfg;
  cfg->width = width; cfg->height = height;
  cfg->bit_depth = 8;
  cfg->n_planes = 1;
  cfg->chroma_hshift = cfg->chroma_vshift = 1;
  cfg->features = PQA_FEAT_VMAF;
  cfg->max_batch = 0;  // auto
  cfg->result_capacity = 16384;
  cfg->n_subsample = 1;
  cfg->vif_enhn_gain_limit = 100.0;
  cfg->adm_enhn_gain_limit = 100.0;
}

int pqa_create(const pqa_config* cfg, pqa_ctx** out) {
  if (!cfg || !out) return fail(nullptr, PQA_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(pqa_config)) return fail(nullptr, PQA_EINVAL, "pqa_config.struct_size mismatch");
  if (cfg->width < 16 || cfg->height < 16 || cfg->width > 16384 || cfg->height > 16384)
    return fail(nullptr, PQA_EINVAL, "unsupported frame size %ux%u (16..16384)", cfg->width, cfg->height);
  if (cfg->bit_depth != 8 && cfg->bit_depth != 10 && cfg->bit_depth != 12)
    return fail(nullptr, PQA_EINVAL, "unsupported bit depth %u (8, 10, 12)", cfg->bit_depth);
  if (cfg->n_planes != 1 && cfg->n_planes != 3) return fail(nullptr, PQA_EINVAL, "n_planes must be 1 or 3");
  if (cfg->chroma_hshift > 2 || cfg->chroma_vshift > 2) return fail(nullptr, PQA_EINVAL, "bad chroma shift");
  if ((cfg->features & ~(uint32_t)PQA_FEAT_ALL) || cfg->features == 0)
    return fail(nullptr, PQA_EINVAL, "bad feature mask 0x%x", cfg->features);
  if (cfg->vif_border > PQA_VIF_BORDER_INTEGER || (cfg->fixed_point & ~(uint32_t)PQA_FIXED_ALL))
    return fail(nullptr, PQA_EINVAL, "bad vif_border %u / fixed_point 0x%x", cfg->vif_border, cfg->fixed_point);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, PQA_EDEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PQA_EINVAL, "device %d out of range (%d)", cfg->device, ndev);

  pqa_ctx* c = new (std::nothrow) pqa_ctx();
  if (!c) return fail(nullptr, PQA_ENOMEM, "out of host memory");
  c->cfg = *cfg;
  c->device = cfg->device;
  if (cfg->max_batch) {
    c->B = (int)cfg->max_batch;
  } else {  // auto: about 1.5 GiB of luma per launch (2160p -> 97, 1080p -> 256): every launch pays a ramp and a tail, and the
            // deep scales are rounds of few, long-lived waves -- 32 -> 96 frames per launch measured +3 % at 2160p
            // (profiles/r06h_batch.txt); the workspace that goes with it is 43.5 MB per 2160p frame, 4.2 GB of the 288
    const int64_t per_frame = 2ll * cfg->width * cfg->height * (cfg->bit_depth > 8 ? 2 : 1);
    c->B = (int)((1536ll << 20) / per_frame);
    if (c->B < 8) c->B = 8;
  }
  if (c->B > 256) c->B = 256;
  c->HB = c->B < 8 ? c->B : 8;  // the host path is PCIe-bound: small pinned halves, same kernels
  c->capacity = cfg->result_capacity ? (int)cfg->result_capacity : 16384;
  if (c->capacity < c->B) c->capacity = c->B;
  c->k_sub = cfg->n_subsample > 1 ? (int)cfg->n_subsample : 1;
  if (c->cfg.vif_enhn_gain_limit <= 0) c->cfg.vif_enhn_gain_limit = 100.0;
  if (c->cfg.adm_enhn_gain_limit <= 0) c->cfg.adm_enhn_gain_limit = 100.0;
  c->elem = cfg->bit_depth > 8 ? ELEM_U16 : ELEM_U8;
  c->esize = cfg->bit_depth > 8 ? 2 : 1;
  c->inv_scale = 1.0f / (float)(1 << (cfg->bit_depth - 8));
  c->n_planes = (int)cfg->n_planes;
  c->pw[0] = (int)cfg->width; c->ph[0] = (int)cfg->height;
  for (int p = 1; p < 3; ++p) {
    c->pw[p] = (int)((cfg->width + (1u << cfg->chroma_hshift) - 1) >> cfg->chroma_hshift);
    c->ph[p] = (int)((cfg->height + (1u << cfg->chroma_vshift) - 1) >> cfg->chroma_vshift);
  }

  auto bail = [&](int rc) {
    g_create_error = c->err;
    pqa_destroy(c);
    return rc;
  };
#define CREATE_TRY(expr) do { int rc_ = (expr); if (rc_ != PQA_OK) return bail(rc_); } while (0)
#define CREATE_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fail(c, e_ == hipErrorOutOfMemory ? PQA_ENOMEM : PQA_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    return bail(e_ == hipErrorOutOfMemory ? PQA_ENOMEM : PQA_EDEVICE); } } while (0)

  CREATE_HIP(hipSetDevice(c->device));
  CREATE_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  {
    // PQA_MULTI_STREAM=1: the ADM chain and the motion / PSNR / SSIM kernels on their own streams beside the VIF chain.
    // With the ADM march kernel (memory-bound at scales 0 / 1) that is worth +2 ... +4 % at 2160p (profiles/r04h_batch_sweep.txt,
    // r04j_prio.txt) -- but a kernel's launch then lasts as long as its share of a crowded device allows (VIF scale 0: 1.23 ms
    // instead of 0.76), so the per-kernel figures stop describing the kernels; stream priorities changed nothing.  Off by default.
    const char* e = getenv("PQA_MULTI_STREAM");
    c->multi_stream = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 0;
    const char* t = getenv("PQA_TRACE");
    c->trace = t && t[0] == '1';
    const char* v = getenv("PQA_VIF_MFMA");   // 0: VALU kernels only (the march kernel's test partner); default: march kernel
    c->vif_s0_mode = (v && v[0] == '0') ? VIF_S0_VALU : VIF_S0_AUTO;
    const char* am = getenv("PQA_ADM_MARCH");  // 0: the LDS-tiled ADM kernel (A/B partner of the march kernel)
    const char* ap = getenv("PQA_ADM_PYRAMID");  // 0: the march kernel one scale per launch (test partner of the pyramid kernel)
    c->adm_mode = (am && am[0] == '0') ? ADM_TILED : (ap && ap[0] == '0') ? ADM_MARCH : ADM_AUTO;
    const char* mm = getenv("PQA_MOTION_MARCH");  // 0: the LDS-tiled motion kernel (test partner of the march kernel)
    c->motion_mode = (mm && mm[0] == '0') ? MOTION_TILED : MOTION_AUTO;
  }
  for (int i = 0; i < 2; ++i) {
    CREATE_HIP(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
    CREATE_HIP(hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
  }
  CREATE_HIP(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
  for (int i = 0; i < kBatchEvents; ++i) CREATE_HIP(hipEventCreateWithFlags(&c->batch_ev[i], hipEventDisableTiming));
  try {
    c->ring.init(c->capacity);
  } catch (...) {
    fail(c, PQA_ENOMEM, "out of host memory");
    return bail(PQA_ENOMEM);
  }

  c->vif_fixed = (cfg->features & PQA_FEAT_VIF) && (cfg->fixed_point & PQA_FIXED_VIF);
  c->motion_fixed = (cfg->features & PQA_FEAT_MOTION) && (cfg->fixed_point & PQA_FIXED_MOTION);
  c->adm_fixed = (cfg->features & PQA_FEAT_ADM) && (cfg->fixed_point & PQA_FIXED_ADM);
  if (c->adm_fixed) {
    std::vector<int32_t> lut(65537);
    adm_fixed_div_table(lut.data());
    CREATE_TRY(dev_alloc(c, &c->adm_div_lut, lut.size()));
    CREATE_HIP(hipMemcpy(c->adm_div_lut, lut.data(), lut.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if ((cfg->features & PQA_FEAT_VIF) && !c->vif_fixed) CREATE_HIP(vif_march_prepare());
  if (c->vif_fixed) {
    std::vector<uint16_t> lut(32768);
    vif_fixed_log2_table(lut.data());
    CREATE_TRY(dev_alloc(c, &c->vif_lut, lut.size()));
    CREATE_HIP(hipMemcpy(c->vif_lut, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  const int w = c->pw[0], h = c->ph[0], B = c->B;
  // VIF pyramid (floor halving) and ADM approximation bands (ceil halving), f32, rows padded to 64 B
  int vw = w, vh = h, aw = w, ah = h;
  for (int s = 0; s < 4; ++s) {
    if (s > 0) {
      vw /= 2; vh /= 2;
      aw = (aw + 1) / 2; ah = (ah + 1) / 2;
      Level& V = c->vif_lv[s];
      V.w = vw; V.h = vh; V.pitch = round_up(vw, 16); V.frame_pitch = V.pitch * vh;
      Level& A = c->adm_lv[s];
      A.w = aw; A.h = ah; A.pitch = round_up(aw, 16); A.frame_pitch = A.pitch * ah;
      if (cfg->features & PQA_FEAT_VIF) {
        CREATE_TRY(dev_alloc(c, &V.ref, (size_t)V.frame_pitch * B));
        CREATE_TRY(dev_alloc(c, &V.dis, (size_t)V.frame_pitch * B));
      }
      if (cfg->features & PQA_FEAT_ADM) {
        CREATE_TRY(dev_alloc(c, &A.ref, (size_t)A.frame_pitch * B));
        CREATE_TRY(dev_alloc(c, &A.dis, (size_t)A.frame_pitch * B));
      }
    }
    if (vw < 2 || vh < 2) { fail(c, PQA_EINVAL, "frame too small for 4 VIF scales"); return bail(PQA_EINVAL); }
    c->vif_tiles[s] = vif_tiles_x(s, vw) * vif_tiles_y(vh);
    if (s == 0) {   // scale 0 may run the march kernel, which writes one partial pair per wave segment
      const int mp = vif_march_partials_max(vw, vh);
      c->vif_part_cap0 = c->vif_tiles[0] > mp ? c->vif_tiles[0] : mp;
    }
    const int bw = (aw + 1) / 2, bh = (ah + 1) / 2;  // band size produced at ADM scale s
    c->adm_tiles[s] = adm_tiles_x(bw) * adm_tiles_y(bh);
    const int left = (int)(bw * 0.1 - 0.5), top = (int)(bh * 0.1 - 0.5);
    c->adm_area[s] = (float)((bh - 2 * top) * (bw - 2 * left));
    if ((cfg->features & PQA_FEAT_VIF) && !c->vif_fixed)
      CREATE_TRY(dev_alloc(c, &c->vif_part[s], (size_t)(s == 0 ? c->vif_part_cap0 : c->vif_tiles[s]) * 2 * B));
    if ((cfg->features & PQA_FEAT_VIF) && c->vif_fixed)
      CREATE_TRY(dev_alloc(c, &c->vif_fx_part[s], (size_t)c->vif_tiles[s] * kVifFxPartials * B));
    c->adm_fx[s] = adm_fixed_scale_params(s, bw, bh);
    if ((cfg->features & PQA_FEAT_ADM) && !c->adm_fixed) {   // one sextet per tile (adm.hip) or per wave segment (adm_march.hip)
      int mp = adm_march_partials(bw, bh);
      if (s < 2 && adm_pyramid_takes(c->elem, w, h)) { const int pp = adm_pyramid_partials(w, h); mp = pp > mp ? pp : mp; }
      CREATE_TRY(dev_alloc(c, &c->adm_part[s], (size_t)(c->adm_tiles[s] > mp ? c->adm_tiles[s] : mp) * 6 * B));
    }
    if ((cfg->features & PQA_FEAT_ADM) && c->adm_fixed)
      CREATE_TRY(dev_alloc(c, &c->adm_fx_part[s], (size_t)c->adm_tiles[s] * kAdmFxRows * 6 * B));
  }
  c->motion_tiles_n = motion_tiles(w, h);
  if (cfg->features & PQA_FEAT_MOTION) {
    {   // one partial per tile (motion.hip) or per wave segment (motion_march.hip)
      const int mp = motion_march_partials(w, h);
      CREATE_TRY(dev_alloc(c, &c->motion_part, (size_t)(c->motion_tiles_n > mp ? c->motion_tiles_n : mp) * B));
    }
    if (c->motion_fixed) CREATE_TRY(dev_alloc(c, &c->motion_fx_part, (size_t)c->motion_tiles_n * B));
    c->last_luma_pitch = round_up((int64_t)w * c->esize, 64);
    CREATE_TRY(dev_alloc(c, &c->last_luma, (size_t)c->last_luma_pitch * h));
  }
  for (int p = 0; p < c->n_planes; ++p) {
    if (cfg->features & PQA_FEAT_PSNR) {
      CREATE_TRY(dev_alloc(c, &c->sse_part[p], (size_t)kSseBlocksPerPlane * B));
      CREATE_TRY(dev_alloc(c, &c->sse_part_b[p], (size_t)kSseBlocksPerPlane * B));
    }
    if (cfg->features & PQA_FEAT_SSIM) {
      c->ssim_tiles_n[p] = ssim_tiles(c->pw[p], c->ph[p]);
      const int ww = (c->pw[p] >> 2) - 1, wh = (c->ph[p] >> 2) - 1;
      c->ssim_norm[p] = (ww > 0 && wh > 0) ? 1.0 / ((double)ww * wh) : 0.0;
      CREATE_TRY(dev_alloc(c, &c->ssim_part[p], (size_t)(c->ssim_tiles_n[p] ? c->ssim_tiles_n[p] : 1) * B));
      CREATE_TRY(dev_alloc(c, &c->sse_tile_part[p], (size_t)(c->ssim_tiles_n[p] ? c->ssim_tiles_n[p] : 1) * B));
    }
  }
  CREATE_TRY(dev_alloc(c, &c->luma_part, (size_t)kLumaBlocks * 3 * B));
  CREATE_TRY(dev_alloc(c, &c->luma_out, (size_t)3 * (B > kLumaOutFrames ? B : kLumaOutFrames)));
  if (c->adm_fixed) {
    CREATE_TRY(dev_alloc(c, &c->adm_fx_acc, (size_t)c->capacity * 24));
    CREATE_HIP(hipMemsetAsync(c->adm_fx_acc, 0, (size_t)c->capacity * 24 * sizeof(long long), c->stream));
  }
  CREATE_TRY(dev_alloc(c, &c->records, (size_t)c->capacity * PQA_RECORD_DOUBLES));
  CREATE_HIP(hipMemsetAsync(c->records, 0, (size_t)c->capacity * PQA_RECORD_DOUBLES * sizeof(double), c->stream));
  CREATE_HIP(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
#undef CREATE_HIP
  *out = c;
  return PQA_OK;
}

void pqa_destroy(pqa_ctx* c) {
  if (!c) return;
  if (c->pin_thread.joinable()) c->pin_thread.join();
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
  for (int i = 0; i < 2; ++i) if (c->aux[i]) hipStreamSynchronize(c->aux[i]);
  prof_drain(c);
  for (void* p : c->allocs) hipFree(p);
  {  // park the staging buffers for the next context (everything using them has been synchronised above)
    StagingSet mine;
    mine.device = c->device;
    mine.bytes = c->slot_bytes * (size_t)c->HB;
    bool complete = c->staging_ready;
    for (int i = 0; i < 2; ++i) {
      mine.pinned[i] = c->half[i].pinned;
      mine.dev[i] = c->half[i].dev;
      complete = complete && mine.pinned[i] && mine.dev[i];
    }
    if (complete && staging_cache_enabled()) {
      std::lock_guard<std::mutex> g(g_staging_mu);
      if (g_staging.bytes) {           // one set per process: the older one goes
        if (g_staging.device != c->device) hipSetDevice(g_staging.device);
        staging_release(g_staging);
        hipSetDevice(c->device);
      }
      g_staging = mine;
    } else {
      staging_release(mine);
    }
  }
  for (int i = 0; i < 2; ++i) {
    Half& H = c->half[i];
    if (H.copied) hipEventDestroy(H.copied);
    if (H.computed) hipEventDestroy(H.computed);
  }
  for (int i = 0; i < 2; ++i) {
    if (c->luma_pinned[i]) hipHostFree(c->luma_pinned[i]);
    if (c->luma_dev[i]) hipFree(c->luma_dev[i]);
    if (c->luma_copied[i]) hipEventDestroy(c->luma_copied[i]);
  }
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  for (int i = 0; i < 2; ++i) {
    if (c->aux[i]) hipStreamDestroy(c->aux[i]);
    if (c->join_ev[i]) hipEventDestroy(c->join_ev[i]);
  }
  if (c->fork_ev) hipEventDestroy(c->fork_ev);
  for (int i = 0; i < kBatchEvents; ++i) if (c->batch_ev[i]) hipEventDestroy(c->batch_ev[i]);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
}

int pqa_set_stream(pqa_ctx* c, void* hip_stream) {
  if (!c) return PQA_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->done_seq = c->batch_seq;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return PQA_OK;
}

int pqa_submit_device(pqa_ctx* c, int64_t first_index, int32_t n_frames, const pqa_device_clip* ref,
                      const pqa_device_clip* dis, const void* prev_ref_luma, int64_t prev_row_pitch) {
  if (!c) return PQA_EINVAL;
  if (!ref || !dis || n_frames < 0 || first_index < 0) return fail(c, PQA_EINVAL, "bad argument");
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  if (n_frames > c->capacity)
    return fail(c, PQA_ESTATE, "%d frames in one call exceed result_capacity %d (records would overwrite each other)",
                n_frames, c->capacity);
  HIPCHK(c, hipSetDevice(c->device));
  int rc = flush_pending(c);
  if (rc != PQA_OK) return rc;
  if ((rc = check_slots_free(c, first_index, n_frames)) != PQA_OK) return rc;   // the whole run, before any batch is launched
  // a run longer than a batch goes down in EQUAL launches (300 frames at batch 32: ten of 30, not nine of 32 and one of 12 --
  // a short launch pays the same ramp and tail for fewer waves); records do not depend on how a run is cut
  const int n_launch = (n_frames + c->B - 1) / c->B;
  const int per = n_launch ? (n_frames + n_launch - 1) / n_launch : 0;
  for (int done = 0; done < n_frames;) {
    if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
    const int n = n_frames - done < per ? n_frames - done : per;
    pqa_device_clip r = *ref, d = *dis;
    for (int p = 0; p < c->n_planes; ++p) {
      r.plane[p] = (const uint8_t*)ref->plane[p] + (int64_t)done * ref->frame_pitch[p];
      d.plane[p] = (const uint8_t*)dis->plane[p] + (int64_t)done * dis->frame_pitch[p];
    }
    const void* prev = nullptr;
    int64_t prev_pitch = 0;
    if (done == 0) {
      prev = prev_ref_luma;
      prev_pitch = prev_row_pitch;
    } else {
      prev = (const uint8_t*)ref->plane[0] + (int64_t)(done - 1) * ref->frame_pitch[0];
      prev_pitch = ref->row_pitch[0];
    }
    rc = process_batch(c, first_index + done, n, &r, &d, prev, prev_pitch);
    if (rc != PQA_OK) return rc;
    done += n;
  }
  return PQA_OK;
}

int pqa_submit_surfaces(pqa_ctx* c, int64_t first_index, int32_t n_frames, const pqa_surface_clip* ref,
                        const pqa_surface_clip* dis, const pqa_surface_clip* prev_ref) {
  if (!c) return PQA_EINVAL;
  if (!ref || !dis || n_frames < 0 || first_index < 0) return fail(c, PQA_EINVAL, "bad argument");
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  if (n_frames > c->capacity)
    return fail(c, PQA_ESTATE, "%d frames in one call exceed result_capacity %d (records would overwrite each other)",
                n_frames, c->capacity);
  const int bpc = (int)c->cfg.bit_depth;
  const uint32_t want = bpc <= 8 ? (uint32_t)PQA_SURFACE_NV12 : (uint32_t)PQA_SURFACE_P01X;
  const pqa_surface_clip* all[3] = {ref, dis, prev_ref};
  for (int i = 0; i < 3; ++i) {
    const pqa_surface_clip* s = all[i];
    if (!s) continue;
    if (s->struct_size != sizeof(pqa_surface_clip)) return fail(c, PQA_EINVAL, "pqa_surface_clip.struct_size mismatch");
    if (s->format != want)
      return fail(c, PQA_EINVAL, "surface format %u does not fit a %d-bit context (NV12 for 8 bit, P01X above)", s->format, bpc);
    if (!s->luma || s->luma_row_pitch < (int64_t)c->pw[0] * c->esize || s->luma_row_pitch % c->esize)
      return fail(c, PQA_EINVAL, "surface luma pointer / pitch");
    if (i < 2 && c->n_planes == 3 &&
        (!s->chroma || s->chroma_row_pitch < (int64_t)c->pw[1] * 2 * c->esize || s->chroma_row_pitch % c->esize))
      return fail(c, PQA_EINVAL, "surface chroma pointer / pitch");
  }
  if (c->n_planes == 3 && (c->cfg.chroma_hshift != 1 || c->cfg.chroma_vshift != 1))
    return fail(c, PQA_EINVAL, "decoder surfaces are 4:2:0: the context's chroma shifts must be 1, 1");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = flush_pending(c);
  if (rc != PQA_OK) return rc;
  if ((rc = check_slots_free(c, first_index, n_frames)) != PQA_OK) return rc;   // before anything is unpacked
  const bool shift_luma = bpc > 8;
  const int shift = 16 - bpc;
  // Unpacked planes go to a device buffer of B slots.  One buffer is enough: unpack and scoring run on the same stream,
  // so the next batch's unpack starts when this batch's kernels are done.
  const bool stage = shift_luma || c->n_planes == 3;
  if (stage && !c->surf_dev) {
    slot_layout(c);
    HIPCHK(c, hipMalloc((void**)&c->surf_dev, c->slot_bytes * (size_t)c->B));
    c->allocs.push_back(c->surf_dev);
  }
  if (prev_ref && (c->cfg.features & PQA_FEAT_MOTION)) {
    if (shift_luma) {   // the halo frame goes where a previous batch would have left it, shifted down
      HIPCHK(c, launch_ingest_shift16(c->stream, prev_ref->luma, prev_ref->luma_row_pitch, 0, c->last_luma, c->last_luma_pitch,
                                      0, c->pw[0], c->ph[0], shift, 1));
      c->halo_armed = true;
    }
  }
  const int n_launch = (n_frames + c->B - 1) / c->B;   // equal launches, as in pqa_submit_device
  const int per = n_launch ? (n_frames + n_launch - 1) / n_launch : 0;
  for (int done = 0; done < n_frames;) {
    if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
    const int n = n_frames - done < per ? n_frames - done : per;
    pqa_device_clip r{}, d{};
    const pqa_surface_clip* side[2] = {ref, dis};
    pqa_device_clip* out[2] = {&r, &d};
    for (int sd = 0; sd < 2; ++sd) {
      const pqa_surface_clip* s = side[sd];
      const uint8_t* luma = (const uint8_t*)s->luma + (int64_t)done * s->luma_frame_pitch;
      if (shift_luma) {
        uint8_t* dst = c->surf_dev + c->plane_off[sd][0];
        HIPCHK(c, launch_ingest_shift16(c->stream, luma, s->luma_row_pitch, s->luma_frame_pitch, dst, c->slot_row_pitch[0],
                                        (int64_t)c->slot_bytes, c->pw[0], c->ph[0], shift, n));
        out[sd]->plane[0] = dst; out[sd]->row_pitch[0] = c->slot_row_pitch[0]; out[sd]->frame_pitch[0] = (int64_t)c->slot_bytes;
      } else {   // NV12 luma: scored in place
        out[sd]->plane[0] = luma; out[sd]->row_pitch[0] = s->luma_row_pitch; out[sd]->frame_pitch[0] = s->luma_frame_pitch;
      }
      if (c->n_planes == 3) {
        const uint8_t* uv = (const uint8_t*)s->chroma + (int64_t)done * s->chroma_frame_pitch;
        uint8_t* du = c->surf_dev + c->plane_off[sd][1];
        uint8_t* dv = c->surf_dev + c->plane_off[sd][2];
        if (c->slot_row_pitch[1] != c->slot_row_pitch[2]) return fail(c, PQA_EINVAL, "internal: U and V staging pitches differ");
        HIPCHK(c, launch_ingest_deinterleave(c->stream, c->esize, uv, s->chroma_row_pitch, s->chroma_frame_pitch, du, dv,
                                             c->slot_row_pitch[1], (int64_t)c->slot_bytes, c->pw[1], c->ph[1],
                                             shift_luma ? shift : 0, n));
        out[sd]->plane[1] = du; out[sd]->plane[2] = dv;
        out[sd]->row_pitch[1] = out[sd]->row_pitch[2] = c->slot_row_pitch[1];
        out[sd]->frame_pitch[1] = out[sd]->frame_pitch[2] = (int64_t)c->slot_bytes;
      }
    }
    // the halo of the first batch: an NV12 luma plane is usable as it is; a shifted one was armed above
    const void* prev = nullptr;
    int64_t prev_pitch = 0;
    if (done == 0 && prev_ref && !shift_luma) { prev = prev_ref->luma; prev_pitch = prev_ref->luma_row_pitch; }
    rc = process_batch(c, first_index + done, n, &r, &d, prev, prev_pitch);
    if (rc != PQA_OK) return rc;
    done += n;
  }
  return PQA_OK;
}

int pqa_submit(pqa_ctx* c, int64_t frame_index, const void* const ref_planes[3], const int64_t ref_strides[3],
               const void* const dis_planes[3], const int64_t dis_strides[3]) {
  if (!c) return PQA_EINVAL;
  if (!ref_planes || !dis_planes || !ref_strides || !dis_strides || frame_index < 0)
    return fail(c, PQA_EINVAL, "bad argument");
  PlaneSrc src[2][3];
  for (int p = 0; p < c->n_planes; ++p) {
    if (!ref_planes[p] || !dis_planes[p]) return fail(c, PQA_EINVAL, "plane %d pointer is null", p);
    const size_t row_bytes = (size_t)c->pw[p] * c->esize;
    if ((size_t)ref_strides[p] < row_bytes || (size_t)dis_strides[p] < row_bytes)
      return fail(c, PQA_EINVAL, "plane %d stride smaller than a row", p);
    src[0][p] = PlaneSrc{(const uint8_t*)ref_planes[p], ref_strides[p], -1, 0};
    src[1][p] = PlaneSrc{(const uint8_t*)dis_planes[p], dis_strides[p], -1, 0};
  }
  return submit_one(c, frame_index, src);
}

int pqa_submit_fd(pqa_ctx* c, int64_t frame_index, int ref_fd, const int64_t ref_plane_offsets[3], int dis_fd,
                  const int64_t dis_plane_offsets[3]) {
  if (!c) return PQA_EINVAL;
  if (ref_fd < 0 || dis_fd < 0 || !ref_plane_offsets || !dis_plane_offsets || frame_index < 0)
    return fail(c, PQA_EINVAL, "bad argument");
  PlaneSrc src[2][3];
  for (int p = 0; p < c->n_planes; ++p) {
    if (ref_plane_offsets[p] < 0 || dis_plane_offsets[p] < 0) return fail(c, PQA_EINVAL, "plane %d file offset is negative", p);
    const int64_t row_bytes = (int64_t)c->pw[p] * c->esize;   // planes lie packed in the file: rows row_bytes apart
    src[0][p] = PlaneSrc{nullptr, row_bytes, ref_fd, ref_plane_offsets[p]};
    src[1][p] = PlaneSrc{nullptr, row_bytes, dis_fd, dis_plane_offsets[p]};
  }
  return submit_one(c, frame_index, src);
}

int pqa_submit_fd_run(pqa_ctx* c, int64_t first_index, int32_t n_frames, int ref_fd, const int64_t ref_plane_offsets[3],
                      int64_t ref_frame_stride, int dis_fd, const int64_t dis_plane_offsets[3], int64_t dis_frame_stride) {
  if (!c) return PQA_EINVAL;
  if (ref_fd < 0 || dis_fd < 0 || !ref_plane_offsets || !dis_plane_offsets || first_index < 0 || n_frames < 0 ||
      ref_frame_stride < 0 || dis_frame_stride < 0)
    return fail(c, PQA_EINVAL, "bad argument");
  if (n_frames > c->capacity)
    return fail(c, PQA_ESTATE, "%d frames in one call exceed result_capacity %d (records would overwrite each other)", n_frames, c->capacity);
  PlaneSrc src[2][3];
  for (int p = 0; p < c->n_planes; ++p) {
    if (ref_plane_offsets[p] < 0 || dis_plane_offsets[p] < 0) return fail(c, PQA_EINVAL, "plane %d file offset is negative", p);
    const int64_t row_bytes = (int64_t)c->pw[p] * c->esize;   // planes lie packed in the file: rows row_bytes apart
    src[0][p] = PlaneSrc{nullptr, row_bytes, ref_fd, ref_plane_offsets[p]};
    src[1][p] = PlaneSrc{nullptr, row_bytes, dis_fd, dis_plane_offsets[p]};
  }
  const int64_t step[2] = {ref_frame_stride, dis_frame_stride};
  return submit_run(c, first_index, n_frames, src, step);
}

int pqa_set_motion_halo(pqa_ctx* c, const void* prev_ref_luma_host, int64_t row_stride) {
  if (!c) return PQA_EINVAL;
  if (!(c->cfg.features & PQA_FEAT_MOTION)) return PQA_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = flush_pending(c);
  if (rc != PQA_OK) return rc;
  if (!prev_ref_luma_host) {
    c->halo_armed = false;
    c->have_last = false;
    return PQA_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy2D(c->last_luma, c->last_luma_pitch, prev_ref_luma_host, row_stride, (size_t)c->pw[0] * c->esize,
                        c->ph[0], hipMemcpyHostToDevice));
  c->halo_armed = true;
  return PQA_OK;
}

int pqa_flush(pqa_ctx* c) {
  if (!c) return PQA_EINVAL;
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  HIPCHK(c, hipSetDevice(c->device));
  return flush_pending(c);
}

int pqa_sync(pqa_ctx* c) {
  if (!c) return PQA_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = c->cancelled.load() ? PQA_OK : flush_pending(c);
  if (rc != PQA_OK) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->done_seq = c->batch_seq;
  prof_drain(c);
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  return PQA_OK;
}

int pqa_collect(pqa_ctx* c, int64_t first_index, int32_t count, double* records) {
  if (!c) return PQA_EINVAL;
  if (count < 0 || first_index < 0 || (count > 0 && !records)) return fail(c, PQA_EINVAL, "bad argument");
  if (count > c->capacity) return fail(c, PQA_ESTATE, "count %d exceeds result_capacity %d", count, c->capacity);
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = flush_pending(c);  // a partial host batch holding requested frames has to be launched first
  if (rc != PQA_OK) return rc;
  // every requested frame must be the one its ring slot holds (never submitted / overwritten -> PQA_ESTATE);
  // then wait for the youngest batch among them only -- later batches keep running under the host's work
  uint64_t need = 0;
  {
    int64_t bad = 0, holder = 0;
    bool never = false;
    if (!c->ring.collectable(first_index, count, &need, &bad, &never, &holder)) {
      if (never) return fail(c, PQA_ESTATE, "frame %lld was never submitted", (long long)bad);
      return fail(c, PQA_ESTATE, "frame %lld was never submitted, or its record was overwritten by frame %lld",
                  (long long)bad, (long long)holder);
    }
  }
  rc = wait_batch(c, need);
  if (rc != PQA_OK) return rc;
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  const size_t rec_bytes = PQA_RECORD_DOUBLES * sizeof(double);
  int64_t row = first_index % c->capacity;
  int done = 0;
  while (done < count) {
    const int n = (int)((c->capacity - row) < (count - done) ? (c->capacity - row) : (count - done));
    HIPCHK(c, hipMemcpy(records + (size_t)done * PQA_RECORD_DOUBLES, c->records + (size_t)row * PQA_RECORD_DOUBLES,
                        n * rec_bytes, hipMemcpyDeviceToHost));
    if (c->adm_fixed) {
      // integer_adm.c's scalar epilogue (six cube roots per scale) on the host, with the libm libvmaf would use:
      // the integer accumulators are exact, so are the resulting adm num / den
      std::vector<long long> acc;
      try {
        acc.resize((size_t)n * 24);
      } catch (...) {
        return fail(c, PQA_ENOMEM, "out of host memory");
      }
      HIPCHK(c, hipMemcpy(acc.data(), c->adm_fx_acc + (size_t)row * 24, acc.size() * sizeof(long long),
                          hipMemcpyDeviceToHost));
      for (int f = 0; f < n; ++f)
        for (int s = 0; s < 4; ++s) {
          double* rec = records + (size_t)(done + f) * PQA_RECORD_DOUBLES;
          adm_fixed_epilogue(c->adm_fx[s], &acc[(size_t)f * 24 + s * 6], &rec[PQA_REC_ADM_NUM + s], &rec[PQA_REC_ADM_DEN + s]);
        }
    }
    done += n;
    row = 0;
  }
  c->ring.mark_collected(first_index, count);
  return PQA_OK;
}

int pqa_luma_stats_device(pqa_ctx* c, const void* luma, int64_t row_pitch, int64_t frame_pitch, int32_t n_frames,
                          uint32_t threshold, uint64_t* out) {
  if (!c) return PQA_EINVAL;
  if (!luma || n_frames < 0 || (n_frames > 0 && !out)) return fail(c, PQA_EINVAL, "bad argument");
  if (row_pitch % c->esize || frame_pitch % c->esize) return fail(c, PQA_EINVAL, "pitch is not a multiple of the sample size");
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  HIPCHK(c, hipSetDevice(c->device));
  // launches of up to B frames each write their results side by side into luma_out (kLumaOutFrames frames); the host
  // copy and the sync happen once per kLumaOutFrames frames, not once per launch
  for (int done = 0; done < n_frames;) {
    const int group = n_frames - done < kLumaOutFrames ? n_frames - done : kLumaOutFrames;
    for (int g0 = 0; g0 < group;) {
      const int n = group - g0 < c->B ? group - g0 : c->B;
      const PlaneRun run{(const uint8_t*)luma + (int64_t)(done + g0) * frame_pitch, row_pitch / c->esize, frame_pitch / c->esize};
      HIPCHK(c, launch_luma_stats(c->stream, c->elem, run, n, c->pw[0], c->ph[0], threshold,
                                  c->luma_gray == PQA_GRAY_BT601_FULL ? (int)c->cfg.bit_depth : 0, c->luma_part,
                                  c->luma_out + (size_t)g0 * 3));
      g0 += n;
    }
    HIPCHK(c, hipMemcpyAsync(out + (size_t)done * 3, c->luma_out, (size_t)group * 3 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    done += group;
  }
  return PQA_OK;
}

int pqa_luma_stats(pqa_ctx* c, const void* const* luma_frames, int64_t row_stride, int32_t n_frames, uint32_t threshold,
                   uint64_t* out) {
  if (!c) return PQA_EINVAL;
  if (n_frames < 0 || (n_frames > 0 && (!luma_frames || !out))) return fail(c, PQA_EINVAL, "bad argument");
  const size_t row_bytes = (size_t)c->pw[0] * c->esize;
  if (n_frames > 0 && (size_t)row_stride < row_bytes) return fail(c, PQA_EINVAL, "stride smaller than a row");
  for (int f = 0; f < n_frames; ++f)   // before anything is queued
    if (!luma_frames[f]) return fail(c, PQA_EINVAL, "frame %d pointer is null", f);
  if (c->cancelled.load()) return fail(c, PQA_ECANCELLED, "cancelled");
  HIPCHK(c, hipSetDevice(c->device));
  const int h = c->ph[0];
  if (!c->luma_ready) {  // lazily: most contexts never detect bookends
    c->luma_pitch = round_up((int64_t)row_bytes, 64);
    c->LB = c->B < 8 ? c->B : 8;
    const size_t half = (size_t)c->luma_pitch * h * c->LB;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      if (!c->luma_pinned[i]) e = hipHostMalloc((void**)&c->luma_pinned[i], half, hipHostMallocDefault);
      if (e == hipSuccess && !c->luma_dev[i]) e = hipMalloc((void**)&c->luma_dev[i], half);
      if (e == hipSuccess && !c->luma_copied[i]) e = hipEventCreateWithFlags(&c->luma_copied[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {   // all six or none: a half-made set must not make every later call fail on a null handle
      for (int i = 0; i < 2; ++i) {
        if (c->luma_pinned[i]) hipHostFree(c->luma_pinned[i]);
        if (c->luma_dev[i]) hipFree(c->luma_dev[i]);
        if (c->luma_copied[i]) hipEventDestroy(c->luma_copied[i]);
        c->luma_pinned[i] = c->luma_dev[i] = nullptr;
        c->luma_copied[i] = nullptr;
      }
      return fail(c, e == hipErrorOutOfMemory ? PQA_ENOMEM : PQA_EDEVICE, "luma staging allocation failed: %s", hipGetErrorString(e));
    }
    c->luma_ready = true;
  }
  const size_t frame_bytes = (size_t)c->luma_pitch * h;
  const int gray_bpc = c->luma_gray == PQA_GRAY_BT601_FULL ? (int)c->cfg.bit_depth : 0;
  // Chunks of LB frames alternate between two pinned / device halves; every chunk's kernel writes its results next to the
  // previous ones in luma_out, and the host gets them in ONE copy + sync per kLumaOutFrames frames: no device-to-host copy
  // sits between the chunks (into the caller's pageable `out` it would block the host until the kernel in front of it is
  // done, and the next chunk could not be packed under that kernel).
  int rc = PQA_OK;
  int chunk = 0;
  for (int done = 0; done < n_frames && rc == PQA_OK;) {
    const int group = n_frames - done < kLumaOutFrames ? n_frames - done : kLumaOutFrames;
    for (int g0 = 0; g0 < group && rc == PQA_OK; ++chunk) {
      if (c->cancelled.load()) { rc = fail(c, PQA_ECANCELLED, "cancelled"); break; }
      const int n = group - g0 < c->LB ? group - g0 : c->LB;
      const int hf = chunk & 1;
      hipError_t e = hipSuccess;
      if (chunk >= 2) e = hipEventSynchronize(c->luma_copied[hf]);   // the upload two chunks ago has left this pinned half
      if (e == hipSuccess) {
        for (int f = 0; f < n; ++f)
          copy_plane_rows(c->luma_pinned[hf] + (size_t)f * frame_bytes, c->luma_pitch, (const uint8_t*)luma_frames[done + g0 + f],
                          row_stride, row_bytes, h);
        e = hipMemcpyAsync(c->luma_dev[hf], c->luma_pinned[hf], (size_t)n * frame_bytes, hipMemcpyHostToDevice, c->stream);
      }
      if (e == hipSuccess) e = hipEventRecord(c->luma_copied[hf], c->stream);
      if (e == hipSuccess) {
        const PlaneRun run{c->luma_dev[hf], c->luma_pitch / c->esize, (int64_t)(frame_bytes / c->esize)};
        e = launch_luma_stats(c->stream, c->elem, run, n, c->pw[0], h, threshold, gray_bpc, c->luma_part,
                              c->luma_out + (size_t)g0 * 3);
      }
      if (e != hipSuccess) { rc = fail(c, PQA_EDEVICE, "luma statistics chunk failed: %s", hipGetErrorString(e)); break; }
      g0 += n;
    }
    if (rc == PQA_OK) {
      const hipError_t e = hipMemcpyAsync(out + (size_t)done * 3, c->luma_out, (size_t)group * 3 * sizeof(uint64_t),
                                          hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) rc = fail(c, PQA_EDEVICE, "luma statistics copy failed: %s", hipGetErrorString(e));
    }
    // always: nothing queued on the stream may still point at the pinned halves or at `out` when this call returns
    const hipError_t es = hipStreamSynchronize(c->stream);
    if (es != hipSuccess && rc == PQA_OK) rc = fail(c, PQA_EDEVICE, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    done += group;
  }
  return rc;
}

int pqa_set_luma_gray(pqa_ctx* c, uint32_t mode) {
  if (!c) return PQA_EINVAL;
  if (mode != PQA_GRAY_LUMA && mode != PQA_GRAY_BT601_FULL) return fail(c, PQA_EINVAL, "bad gray mode %u", mode);
  c->luma_gray = mode;
  return PQA_OK;
}

int pqa_cancel(pqa_ctx* c) {
  if (!c) return PQA_EINVAL;
  c->cancelled.store(1);
  return PQA_OK;
}

int pqa_reset(pqa_ctx* c) {
  if (!c) return PQA_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  c->cancelled.store(0);
  c->done_seq = c->batch_seq;
  c->ring.reset();
  c->pending = 0;
  c->have_last = false;
  c->halo_armed = false;
  c->last_index = -1;
  c->err.clear();
  return PQA_OK;
}

const char* pqa_last_error(const pqa_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int pqa_profile_enable(pqa_ctx* c, int on) {
  if (!c) return PQA_EINVAL;
  prof_drain(c);
  c->prof = on != 0;
  c->prof_mask = (on == 1 || on == 0) ? 0xffffffffu : (uint32_t)on >> 1;  // on = 1: all; on = (mask << 1): subset
  if (on) {
    memset(c->prof_ms, 0, sizeof c->prof_ms);
    memset(c->prof_n, 0, sizeof c->prof_n);
    memset(c->prof_frames, 0, sizeof c->prof_frames);
  }
  return PQA_OK;
}

int pqa_profile_read(pqa_ctx* c, int kernel_id, double* total_ms, uint64_t* launches, uint64_t* frames) {
  if (!c || kernel_id < 0 || kernel_id >= PQA_PROF_KERNELS) return PQA_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  if (total_ms) *total_ms = c->prof_ms[kernel_id];
  if (launches) *launches = c->prof_n[kernel_id];
  if (frames) *frames = c->prof_frames[kernel_id];
  return PQA_OK;
}

int pqa_debug_vif_march_table(uint16_t* out, int32_t capacity_halfwords) { return vif_march_table(out, capacity_halfwords); }
int pqa_debug_vif_march_shape(uint32_t width, uint32_t height, int32_t* out6) {
  if (!out6 || width == 0 || height == 0 || width > 65536 || height > 65536) return PQA_EINVAL;
  int shape[6];
  vif_march_shape((int)width, (int)height, shape);
  for (int i = 0; i < 6; ++i) out6[i] = shape[i];
  return PQA_OK;
}

const char* pqa_profile_kernel_name(int kernel_id) {
  return (kernel_id >= 0 && kernel_id < PQA_PROF_KERNELS) ? kProfNames[kernel_id] : "";
}

}  // extern "C"
