// Host side of the staging path, free of any device call: packing frames -- from memory planes or straight out of files -- into
// (pinned) staging slots with a small persistent thread pool.  Header-only and plain C++17 so that a CPU harness can drive it
// under ThreadSanitizer / AddressSanitizer (tests/host_harness.cpp, pytest -m "not gpu"); pqa_api.hip includes it as is.
#pragma once
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace pqa {
namespace host {

// pqa_submit packs the caller's planes into pinned staging.  One core's memcpy (~10 GB/s) is far below PCIe, so the
// rows of a frame pair are split into ~1 MiB tasks that a few persistent helper threads and the caller drain together.
struct PackTask {
  uint8_t* dst;
  const uint8_t* src;      // memory source (fd < 0) ...
  int64_t dst_pitch, src_pitch;
  size_t row_bytes;
  int rows;
  int fd = -1;             // ... or a file: rows lie src_pitch bytes apart from file_off on (pqa_submit_fd)
  int64_t file_off = 0;
  int frame = 0;           // which frame of a run the task belongs to (PackPool::run_frames)
};

// one task, by whoever takes it.  Returns false on a short read / I/O error of a file source.
inline bool run_pack_task(const PackTask& t) {
  if (t.fd < 0) {
    if (t.dst_pitch == t.src_pitch) {
      memcpy(t.dst, t.src, (size_t)t.dst_pitch * (t.rows - 1) + t.row_bytes);
    } else {
      for (int y = 0; y < t.rows; ++y) memcpy(t.dst + (int64_t)y * t.dst_pitch, t.src + (int64_t)y * t.src_pitch, t.row_bytes);
    }
    return true;
  }
  const auto read_all = [&](uint8_t* dst, size_t len, int64_t off) {
    while (len > 0) {
      const ssize_t got = pread(t.fd, dst, len, (off_t)off);
      if (got < 0 && errno == EINTR) continue;
      if (got <= 0) return false;   // EOF inside a frame or an I/O error
      dst += got; off += got; len -= (size_t)got;
    }
    return true;
  };
  if (t.dst_pitch == t.src_pitch) return read_all(t.dst, (size_t)t.dst_pitch * (t.rows - 1) + t.row_bytes, t.file_off);
  for (int y = 0; y < t.rows; ++y)
    if (!read_all(t.dst + (int64_t)y * t.dst_pitch, t.row_bytes, t.file_off + (int64_t)y * t.src_pitch)) return false;
  return true;
}

class PackPool {
 public:
  static constexpr int kMaxFrames = 64;   // frames of one run_frames call (a staging half holds at most 8)

  explicit PackPool(int helpers) {
    try {
      for (int i = 0; i < helpers; ++i) workers_.emplace_back([this] { loop(); });
    } catch (...) {
      // a std::thread that cannot start (EAGAIN, RLIMIT_NPROC) throws while earlier workers are joinable: unwinding
      // the vector would call std::terminate.  Stop and join what did start, then let the caller fall back to serial.
      shutdown();
      throw;
    }
  }
  ~PackPool() { shutdown(); }
  PackPool(const PackPool&) = delete;
  PackPool& operator=(const PackPool&) = delete;

  bool run(const PackTask* tasks, int n) {   // false: a file source came up short
    return run_frames(tasks, n, 0, [](int) { return true; });
  }

  // The tasks of n_frames consecutive frames (task.frame = 0 .. n_frames - 1, in non-decreasing order) in ONE fork / join:
  // the helpers keep packing frame k + 1 while the CALLER, between its own tasks, hands every completed frame -- in order --
  // to on_frame_done(k) (the upload of that frame: the caller's thread is the only one that talks to the device).  One
  // wake-up of the helpers per run instead of one per frame.  Returns false when a file source came up short or a callback
  // returned false; frames after the first incomplete one are then not handed over.
  template <typename F>
  bool run_frames(const PackTask* tasks, int n, int n_frames, F&& on_frame_done) {
    if (n_frames > kMaxFrames) return false;
    for (int k = 0; k < n_frames; ++k) left_[k].store(0, std::memory_order_relaxed);
    for (int i = 0; i < n; ++i)
      if (n_frames > 0) left_[tasks[i].frame].fetch_add(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> g(m_);
      tasks_ = tasks; n_ = n; track_ = n_frames > 0; next_.store(0); finished_ = 0; failed_.store(false); ++gen_;
    }
    cv_work_.notify_all();
    int issued = 0;
    bool cb_ok = true;
    const auto hand_over = [&] {
      while (cb_ok && issued < n_frames && !failed_.load(std::memory_order_acquire) &&
             left_[issued].load(std::memory_order_acquire) == 0) {
        cb_ok = on_frame_done(issued);
        ++issued;
      }
    };
    for (;;) {   // the caller drains like a helper and looks after completed frames between tasks
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      do_task(tasks_[i]);
      hand_over();
    }
    {
      std::unique_lock<std::mutex> g(m_);
      cv_done_.wait(g, [this] { return finished_ == (int)workers_.size(); });
    }
    hand_over();
    return cb_ok && !failed_.load();
  }

 private:
  void shutdown() noexcept {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto& t : workers_)
      if (t.joinable()) t.join();
    workers_.clear();
  }
  void do_task(const PackTask& t) {
    if (!run_pack_task(t)) failed_.store(true, std::memory_order_release);
    if (track_) left_[t.frame].fetch_sub(1, std::memory_order_acq_rel);
  }
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      do_task(tasks_[i]);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_work_.wait(g, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      drain();
      std::lock_guard<std::mutex> g(m_);
      if (++finished_ == (int)workers_.size()) cv_done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  const PackTask* tasks_ = nullptr;   // tasks_, n_, track_: written under m_ before gen_ moves, read by helpers after they saw it
  int n_ = 0, finished_ = 0;
  bool track_ = false;
  std::atomic<int> next_{0};
  std::atomic<bool> failed_{false};
  std::atomic<int> left_[kMaxFrames];   // tasks of frame k not yet done
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace host
}  // namespace pqa
