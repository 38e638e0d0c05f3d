// CPU harness for the library's device-free host code (pqa2_amd/csrc/host_pack.h, host_ring.h): built with
// -fsanitize=thread and with -fsanitize=address,undefined by tests/test_host_sanitizers.py and driven through the call
// sequences the GPU tests put the real library through (tests/test_gpu_configs.py PQA_ESTATE cases, the truncated-file cases
// of tests/test_gpu_engine.py, a cancel flag raised from another thread).  Memory / race safety only: no parity claim.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <random>
#include <thread>
#include <vector>

#include "../pqa2_amd/csrc/host_pack.h"
#include "../pqa2_amd/csrc/host_ring.h"

using namespace pqa::host;

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #x); exit(1); } } while (0)

struct Geometry { int w, h, n_planes; };

// the task list pqa_api.hip builds for m frames: rows of every plane of both clips in ~task_bytes pieces
static std::vector<PackTask> make_tasks(const Geometry& g, int m, uint8_t* slots, size_t slot_bytes, const size_t (*plane_off)[3],
                                        const int64_t* slot_pitch, const uint8_t* const mem[2], const int fds[2], int64_t frame_step,
                                        const int64_t* file_plane_off, unsigned task_bytes) {
  std::vector<PackTask> t;
  for (int k = 0; k < m; ++k)
    for (int side = 0; side < 2; ++side)
      for (int p = 0; p < g.n_planes; ++p) {
        const int pw = p ? (g.w + 1) / 2 : g.w, ph = p ? (g.h + 1) / 2 : g.h;
        int rows_per = (int)(task_bytes / (unsigned)pw);
        if (rows_per < 1) rows_per = 1;
        for (int y = 0; y < ph; y += rows_per) {
          const int rows = ph - y < rows_per ? ph - y : rows_per;
          PackTask x{slots + (size_t)k * slot_bytes + plane_off[side][p] + (int64_t)y * slot_pitch[p],
                     mem[side] ? mem[side] + (int64_t)k * frame_step + file_plane_off[p] + (int64_t)y * pw : nullptr, slot_pitch[p], pw,
                     (size_t)pw, rows};
          x.fd = mem[side] ? -1 : fds[side];
          x.file_off = (int64_t)k * frame_step + file_plane_off[p] + (int64_t)y * pw;
          x.frame = k;
          t.push_back(x);
        }
      }
  return t;
}

static void pack_scenarios(const char* dir) {
  const Geometry g{322, 182, 3};   // odd width: staging rows are padded, so every plane is packed row by row
  const int n = 11, HB = 4;
  const int cw = (g.w + 1) / 2, ch = (g.h + 1) / 2;
  const int64_t frame_bytes = (int64_t)g.w * g.h + 2ll * cw * ch;
  const int64_t file_plane_off[3] = {0, (int64_t)g.w * g.h, (int64_t)g.w * g.h + (int64_t)cw * ch};
  const int64_t slot_pitch[3] = {(g.w + 63) / 64 * 64, (cw + 63) / 64 * 64, (cw + 63) / 64 * 64};
  size_t plane_off[2][3], off = 0;
  for (int side = 0; side < 2; ++side)
    for (int p = 0; p < 3; ++p) { plane_off[side][p] = off; off += (size_t)slot_pitch[p] * (p ? ch : g.h); }
  const size_t slot_bytes = off;
  std::mt19937 rng(7);
  std::vector<uint8_t> clip[2];
  char path[2][512];
  int fds[2];
  for (int side = 0; side < 2; ++side) {
    clip[side].resize((size_t)frame_bytes * n);
    for (auto& b : clip[side]) b = (uint8_t)rng();
    snprintf(path[side], sizeof path[side], "%s/harness_%d.yuv", dir, side);
    FILE* f = fopen(path[side], "wb");
    CHECK(f && fwrite(clip[side].data(), 1, clip[side].size(), f) == clip[side].size());
    fclose(f);
    fds[side] = open(path[side], O_RDONLY);
    CHECK(fds[side] >= 0);
  }
  std::vector<uint8_t> slots(slot_bytes * HB);
  const auto verify = [&](int first, int m) {
    for (int k = 0; k < m; ++k)
      for (int side = 0; side < 2; ++side)
        for (int p = 0; p < 3; ++p) {
          const int pw = p ? cw : g.w, ph = p ? ch : g.h;
          for (int y = 0; y < ph; ++y)
            CHECK(!memcmp(slots.data() + (size_t)k * slot_bytes + plane_off[side][p] + (size_t)y * slot_pitch[p],
                          clip[side].data() + (size_t)(first + k) * frame_bytes + file_plane_off[p] + (size_t)y * pw, (size_t)pw));
        }
  };
  std::atomic<int> cancelled{0};
  for (int helpers : {1, 3, 7}) {
    PackPool pool(helpers);
    // (1) memory sources, frame by frame (pqa_submit), plain run()
    for (int i = 0; i < n; ++i) {
      const uint8_t* mem[2] = {clip[0].data() + (size_t)i * frame_bytes, clip[1].data() + (size_t)i * frame_bytes};
      auto t = make_tasks(g, 1, slots.data(), slot_bytes, plane_off, slot_pitch, mem, fds, 0, file_plane_off, 16 << 10);
      CHECK(pool.run(t.data(), (int)t.size()));
      verify(i, 1);
    }
    // (2) file sources in runs (pqa_submit_fd_run): frames are handed over in order, on the caller's thread
    const uint8_t* nomem[2] = {nullptr, nullptr};
    const auto caller = std::this_thread::get_id();
    for (int first = 0; first < n; first += HB) {
      const int m = n - first < HB ? n - first : HB;
      std::fill(slots.begin(), slots.end(), 0);
      const int64_t base[3] = {file_plane_off[0] + first * frame_bytes, file_plane_off[1] + first * frame_bytes, file_plane_off[2] + first * frame_bytes};
      auto t = make_tasks(g, m, slots.data(), slot_bytes, plane_off, slot_pitch, nomem, fds, frame_bytes, base, 24 << 10);
      int next = 0;
      const bool ok = pool.run_frames(t.data(), (int)t.size(), m, [&](int k) {
        CHECK(k == next && std::this_thread::get_id() == caller);
        verify(first + k, 0);   // (the slot of frame k is complete here: checked below for all of them)
        ++next;
        return true;
      });
      CHECK(ok && next == m);
      verify(first, m);
    }
    // (3) a file that ends inside frame 2 of a run of 4: false, and no frame from the incomplete one on is handed over
    {
      char sp[512];
      snprintf(sp, sizeof sp, "%s/harness_short.yuv", dir);
      FILE* f = fopen(sp, "wb");
      CHECK(f && fwrite(clip[0].data(), 1, (size_t)(2 * frame_bytes + 1000), f) == (size_t)(2 * frame_bytes + 1000));
      fclose(f);
      int sfd[2] = {open(sp, O_RDONLY), fds[1]};
      CHECK(sfd[0] >= 0);
      auto t = make_tasks(g, 4, slots.data(), slot_bytes, plane_off, slot_pitch, nomem, sfd, frame_bytes, file_plane_off, 24 << 10);
      std::vector<int> seen;
      CHECK(!pool.run_frames(t.data(), (int)t.size(), 4, [&](int k) { seen.push_back(k); return true; }));
      for (size_t i = 0; i < seen.size(); ++i) CHECK(seen[i] == (int)i && seen[i] < 2);
      // a callback that fails (the upload could not be queued) ends the hand-over too
      auto t2 = make_tasks(g, 3, slots.data(), slot_bytes, plane_off, slot_pitch, nomem, fds, frame_bytes, file_plane_off, 24 << 10);
      int calls = 0;
      CHECK(!pool.run_frames(t2.data(), (int)t2.size(), 3, [&](int) { ++calls; return false; }) && calls == 1);
      close(sfd[0]);
      remove(sp);
    }
    // (4) pqa_cancel from another thread while a submit loop runs: the loop polls the flag between runs, as submit_run does
    {
      cancelled.store(0);
      std::thread other([&] { std::this_thread::sleep_for(std::chrono::milliseconds(2)); cancelled.store(1); });
      int submitted = 0;
      for (int rep = 0; rep < 400 && !cancelled.load(); ++rep) {
        auto t = make_tasks(g, HB, slots.data(), slot_bytes, plane_off, slot_pitch, nomem, fds, frame_bytes, file_plane_off, 24 << 10);
        CHECK(pool.run_frames(t.data(), (int)t.size(), HB, [&](int) { return true; }));
        submitted += HB;
      }
      other.join();
      CHECK(submitted > 0);
    }
  }   // the pool is destroyed while idle here, three times
  for (int side = 0; side < 2; ++side) { close(fds[side]); remove(path[side]); }
}

static void ring_scenarios() {
  RecordRing r;
  r.init(4);
  int64_t f = -1, holder = -1, bad = -1;
  bool never = false;
  uint64_t need = 0;
  CHECK(!r.collectable(0, 1, &need, &bad, &never, &holder) && never && bad == 0);       // nothing was ever submitted
  CHECK(r.can_submit(0, 4, &f, &holder));
  r.claim(0, 2, 1);
  r.claim(2, 2, 2);
  CHECK(!r.can_submit(4, 1, &f, &holder) && f == 4 && holder == 0);                      // would overwrite frame 0's record
  CHECK(r.can_submit(0, 4, &f, &holder));                                                // a re-run of the same frames may
  CHECK(r.collectable(0, 4, &need, &bad, &never, &holder) && need == 2);
  CHECK(r.collectable(0, 2, &need, &bad, &never, &holder) && need == 1);                 // ... waits for ITS batch only
  r.mark_collected(0, 1);
  CHECK(r.can_submit(4, 1, &f, &holder));
  CHECK(!r.can_submit(4, 2, &f, &holder) && f == 5 && holder == 1);
  r.claim(4, 1, 3);                                                                      // wraps onto slot 0
  CHECK(!r.collectable(0, 1, &need, &bad, &never, &holder) && !never && bad == 0 && holder == 4);   // overwritten
  CHECK(r.collectable(4, 1, &need, &bad, &never, &holder) && need == 3);
  CHECK(r.collectable(1, 3, &need, &bad, &never, &holder) && need == 2);
  r.mark_collected(1, 3);
  r.mark_collected(4, 1);
  CHECK(r.can_submit(5, 4, &f, &holder));
  r.claim(5, 4, 4);                                                                      // frames 5..8 on slots 1, 2, 3, 0
  CHECK(r.collectable(5, 4, &need, &bad, &never, &holder) && need == 4);
  CHECK(!r.can_submit(9, 1, &f, &holder) && holder == 5);
  r.reset();
  CHECK(r.can_submit(9, 4, &f, &holder));
  CHECK(!r.collectable(5, 1, &need, &bad, &never, &holder) && never);
  // a long walk: submit in batches of 3, collect two batches late; never a false conflict, never a lost record
  r.init(16);
  uint64_t seq = 0;
  for (int64_t first = 0; first < 3000; first += 3) {
    CHECK(r.can_submit(first, 3, &f, &holder));
    r.claim(first, 3, ++seq);
    if (first >= 6) {
      CHECK(r.collectable(first - 6, 3, &need, &bad, &never, &holder) && need == seq - 2);
      r.mark_collected(first - 6, 3);
    }
  }
}

// proof that the sanitizer under which this binary was built is alive: a real data race / a real heap overflow, on request
static int g_plain = 0;
static void selftest(const char* what) {
  if (!strcmp(what, "race")) {
    std::thread a([] { for (int i = 0; i < 100000; ++i) g_plain = g_plain + 1; });
    std::thread b([] { for (int i = 0; i < 100000; ++i) g_plain = g_plain + 1; });
    a.join(); b.join();
  } else {
    std::vector<uint8_t> v(16);
    volatile uint8_t* p = v.data();
    p[16 + (g_plain & 1)] = 1;
  }
  printf("selftest %s done (%d)\n", what, g_plain);
}

int main(int argc, char** argv) {
  if (argc > 2 && !strcmp(argv[1], "--selftest")) { selftest(argv[2]); return 0; }
  const char* dir = argc > 1 ? argv[1] : "/tmp";
  ring_scenarios();
  pack_scenarios(dir);
  printf("host harness ok\n");
  return 0;
}
