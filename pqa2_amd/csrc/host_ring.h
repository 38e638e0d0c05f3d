// Host-side bookkeeping of the record ring, free of any device call (see host_pack.h for why it is a header of its own):
// which frame a slot holds, whether its record was collected, and the batch that writes it.  Lets pqa_collect wait for ITS
// batch only and makes the PQA_ESTATE promises of include/pqa_vmaf.h real.
#pragma once
#include <cstdint>
#include <vector>

namespace pqa {
namespace host {

class RecordRing {
 public:
  enum : uint8_t { kEmpty = 0, kSubmitted = 1, kCollected = 2 };

  void init(int capacity) {
    cap_ = capacity;
    frame_.assign((size_t)capacity, -1);
    state_.assign((size_t)capacity, (uint8_t)kEmpty);
    seq_.assign((size_t)capacity, 0);
  }
  int capacity() const { return cap_; }

  // A slot may be rewritten by the SAME frame index (a re-run) or once its record has been collected; a different frame
  // landing on an uncollected record would lose it silently.  Returns true when frames first .. first + n - 1 may be
  // submitted; otherwise *frame / *holder name the first conflict.
  bool can_submit(int64_t first, int n, int64_t* frame, int64_t* holder) const {
    for (int i = 0; i < n; ++i) {
      const int64_t f = first + i;
      const size_t s = (size_t)(f % cap_);
      if (state_[s] == kSubmitted && frame_[s] != f) {
        if (frame) *frame = f;
        if (holder) *holder = frame_[s];
        return false;
      }
    }
    return true;
  }
  void claim(int64_t first, int n, uint64_t seq) {
    for (int i = 0; i < n; ++i) {
      const size_t s = (size_t)((first + i) % cap_);
      frame_[s] = first + i;
      state_[s] = kSubmitted;
      seq_[s] = seq;
    }
  }
  // What pqa_collect needs to know about frames first .. first + count - 1: all of them submitted (and not overwritten)?
  // Then *need = the latest batch sequence number among them.  Otherwise *bad names the first offending frame and
  // *never says whether it was never submitted (true) or overwritten by another frame (false, *holder = that frame).
  bool collectable(int64_t first, int count, uint64_t* need, int64_t* bad, bool* never, int64_t* holder) const {
    uint64_t n = 0;
    for (int i = 0; i < count; ++i) {
      const int64_t f = first + i;
      const size_t s = (size_t)(f % cap_);
      if (state_[s] == kEmpty || frame_[s] != f) {
        if (bad) *bad = f;
        if (never) *never = state_[s] == kEmpty;
        if (holder) *holder = frame_[s];
        return false;
      }
      if (seq_[s] > n) n = seq_[s];
    }
    if (need) *need = n;
    return true;
  }
  void mark_collected(int64_t first, int count) {
    for (int i = 0; i < count; ++i) state_[(size_t)((first + i) % cap_)] = kCollected;
  }
  void reset() {
    for (auto& s : state_) s = kEmpty;
    for (auto& f : frame_) f = -1;
  }

 private:
  int cap_ = 0;
  std::vector<int64_t> frame_;   // -1: nothing submitted into this slot
  std::vector<uint8_t> state_;
  std::vector<uint64_t> seq_;    // sequence number of the batch whose finalize writes the slot
};

}  // namespace host
}  // namespace pqa
