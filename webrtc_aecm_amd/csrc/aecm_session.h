// Single-stream session behind the reference's public C ABI (include/echo_control_mobile.h).
//
// Host-side restatement of the two content-independent layers above the block DSP:
//   L4  session wrapper  -- far-end jitter buffer, start-up gating, EstBufDelay / DelayComp
//                           (reference aecm/echo_control_mobile.cc:142-408, 534-594)
//   L3  frame adapter    -- 80-sample frames re-blocked to 64-sample blocks, output stuffing
//                           (reference aecm/aecm_core.cc:501-572; ring semantics aecm/ring_buffer.c)
// The blocks themselves (WebRtcAecm_ProcessBlock) run on the GPU through a one-stream BatchEngine.
#ifndef AECM_AMD_SESSION_H_
#define AECM_AMD_SESSION_H_

#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <vector>

#include "aecm_engine.h"

namespace aecm {

// FIFO of int16 samples with the read-pointer semantics of the reference ring buffer
// (aecm/ring_buffer.c:97-211): reads and writes are clamped to what is available, and the read
// pointer can be moved backwards over already-consumed (or never-written, zero) samples.
class SampleRing {
public:
    explicit SampleRing(size_t capacity) : data_(capacity, 0), head_(0), count_(0) {}
    void Reset() { std::fill(data_.begin(), data_.end(), 0); head_ = 0; count_ = 0; }     // WebRtc_InitBuffer :75-82
    size_t available_read() const { return count_; }                                      // :213-223
    size_t available_write() const { return data_.size() - count_; }                      // :225-231
    size_t Write(const int16_t *src, size_t n);                                           // :142-174
    size_t Read(int16_t *dst, size_t n);                                                  // :97-140
    int MoveReadPtr(int n);                                                               // :176-211
private:
    std::vector<int16_t> data_;
    size_t head_, count_;
};

class Session {
public:
    static Session *Create();
    int32_t Init(int32_t samp_freq);
    int32_t BufferFarendError(const int16_t *farend, size_t n) const;
    int32_t BufferFarend(const int16_t *farend, size_t n);
    int32_t Process(const int16_t *near_noisy, const int16_t *near_clean, int16_t *out, size_t n, int16_t ms_in_snd_card_buf);
    int32_t SetConfig(int16_t cng_mode, int16_t echo_mode);
    int32_t InitEchoPath(const void *path, size_t size_bytes);
    int32_t GetEchoPath(void *path, size_t size_bytes);

private:
    Session();
    void EstBufDelay();          // echo_control_mobile.cc:534-573
    void DelayComp();            // echo_control_mobile.cc:575-594

    std::unique_ptr<BatchEngine> engine_;
    // --- AecMobile (echo_control_mobile.cc:42-79) ---
    int samp_freq_ = 0;
    int mult_ = 1;               // core mult (aecm_core.cc:368)
    short buf_size_start_ = 0;
    int known_delay_ = 0;
    int16_t farend_old_[2][80];
    short init_flag_ = 0;
    short counter_ = 0, sum_ = 0, first_val_ = 0, check_buf_size_ctr_ = 0;
    short ms_in_snd_card_buf_ = 0, filt_delay_ = 0;
    int time_for_delay_change_ = 0, ec_startup_ = 0, check_buff_size_ = 0, delay_change_ = 0;
    short last_delay_diff_ = 0;
    SampleRing farend_buf_;      // 50 frames of 80 samples (:31-36,98)
    // --- frame adapter rings (aecm_core.cc:183-205): FRAME_LEN + PART_LEN = 144 samples each ---
    SampleRing far_frames_, near_frames_, clean_frames_, out_frames_;
};

}  // namespace aecm
#endif  // AECM_AMD_SESSION_H_
