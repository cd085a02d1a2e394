// Content-independent machinery of one AECM session, generic over the sample type.
//
// Everything the reference does between the public ABI and WebRtcAecm_ProcessBlock only MOVES
// samples (jitter buffer, start-up gating, delay compensation by re-reading or skipping far-end
// samples, 80 -> 64 re-blocking, output stuffing); none of it looks at their values:
//   L4  session wrapper  reference aecm/echo_control_mobile.cc:142-408, 534-594
//   L3  frame adapter    reference aecm/aecm_core.cc:501-572, ring semantics aecm/ring_buffer.c
// SessionFlow<T> restates that machinery once.  With T = int16_t it drives live audio (the
// single-stream ABI, aecm_session.cpp); with T = int32_t it runs on sample *indices* and yields the
// gather/scatter schedule shared by every stream of a batch that sees the same call pattern
// (aecm_schedule.cpp) -- that is how whole batches of sessions run device-resident.
#ifndef AECM_AMD_SESSION_FLOW_H_
#define AECM_AMD_SESSION_FLOW_H_

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "aecm_state.h"

namespace aecm {

constexpr int kFrameLen = 80;                 // FRAME_LEN (aecm_defines.h:17)
constexpr int kFarBufLen = 256;               // FAR_BUF_LEN = PART_LEN4 (aecm_defines.h:25)
constexpr int kBufSizeFrames = 50;            // BUF_SIZE_FRAMES (echo_control_mobile.cc:29)
constexpr int kSampMsNb = 8;                  // samples per ms, narrowband (:37)
constexpr short kInitCheck = 42;              // (:40)

// Return codes of the ABI (echo_control_mobile.h:23-30), kept numeric here to stay header-only.
constexpr int32_t kErrUnspecified = 12000, kErrUnsupported = 12001, kErrUninitialized = 12002, kErrNullPointer = 12003,
                  kErrBadParameter = 12004, kWarnBadParameter = 12100;

// FIFO with the read-pointer semantics of the reference ring buffer (aecm/ring_buffer.c:97-211):
// reads and writes are clamped to what is available, and the read pointer can be moved backwards
// over already-consumed (or never-written, `fill`) samples.
template <class T>
class SampleRing {
public:
    SampleRing(size_t capacity, T fill) : data_(capacity, fill), fill_(fill), head_(0), count_(0) {}
    void Reset() { std::fill(data_.begin(), data_.end(), fill_); head_ = 0; count_ = 0; }   // WebRtc_InitBuffer :75-82
    size_t available_read() const { return count_; }                                         // :213-223
    size_t available_write() const { return data_.size() - count_; }                         // :225-231
    size_t Write(const T *src, size_t n) {                                                   // :142-174
        n = std::min(n, available_write());
        const size_t cap = data_.size();
        size_t w = (head_ + count_) % cap;
        for (size_t i = 0; i < n; ++i) {
            data_[w] = src[i];
            w = w + 1 == cap ? 0 : w + 1;
        }
        count_ += n;
        return n;
    }
    size_t Read(T *dst, size_t n) {                                                          // :97-140
        n = std::min(n, count_);
        const size_t cap = data_.size();
        for (size_t i = 0; i < n; ++i) {
            dst[i] = data_[head_];
            head_ = head_ + 1 == cap ? 0 : head_ + 1;
        }
        count_ -= n;
        return n;
    }
    int MoveReadPtr(int n) {                                                                 // :176-211
        const int free_elems = (int)available_write();
        const int readable = (int)available_read();
        if (n > readable) n = readable;
        if (n < -free_elems) n = -free_elems;
        const int cap = (int)data_.size();
        int h = ((int)head_ + n) % cap;
        if (h < 0) h += cap;
        head_ = (size_t)h;
        count_ = (size_t)(readable - n);
        return n;
    }
private:
    std::vector<T> data_;
    T fill_;
    size_t head_, count_;
};

template <class T>
class SessionFlow {
public:
    // `zero` is what never-written buffer memory reads as (0 for audio, a "no sample" tag for indices).
    explicit SessionFlow(T zero)
        : zero_(zero), farend_buf_(kBufSizeFrames * kFrameLen, zero), far_frames_(kFrameLen + kBlock, zero),
          near_frames_(kFrameLen + kBlock, zero), clean_frames_(kFrameLen + kBlock, zero),
          out_frames_(kFrameLen + kBlock, zero) {
        Clear();
    }

    bool initialized() const { return init_flag_ == kInitCheck; }
    bool in_startup() const { return ec_startup_ != 0; }

    // WebRtcAecm_Init minus the core (echo_control_mobile.cc:142-191, aecm_core.cc:368-378).
    int32_t Init(int32_t samp_freq) {
        if (samp_freq != 8000 && samp_freq != 16000) return kErrBadParameter;
        samp_freq_ = samp_freq;
        mult_ = samp_freq / 8000;
        far_frames_.Reset(); near_frames_.Reset(); clean_frames_.Reset(); out_frames_.Reset();
        farend_buf_.Reset();
        Clear();
        init_flag_ = kInitCheck;
        delay_change_ = 1;
        check_buff_size_ = 1;
        ec_startup_ = 1;
        return 0;
    }

    int32_t BufferFarendError(const T *farend, size_t n) const {                            // :195-213
        if (farend == nullptr) return kErrNullPointer;
        if (init_flag_ != kInitCheck) return kErrUninitialized;
        if (n != 80 && n != 160) return kErrBadParameter;
        return 0;
    }

    int32_t BufferFarend(const T *farend, size_t n) {                                       // :215-234
        const int32_t err = BufferFarendError(farend, n);
        if (err != 0) return err;
        if (!ec_startup_) DelayComp();
        far_accepted_ = farend_buf_.Write(farend, n);      // a full jitter buffer drops what does not fit (ring_buffer.c:142-150)
        return 0;
    }
    // How many samples of the last BufferFarend call entered the jitter buffer.
    size_t last_far_accepted() const { return far_accepted_; }

    // WebRtcAecm_Process (:236-408).  run_blocks(far, near, clean_or_null, out, n_blocks) must fill
    // out[0 .. n_blocks*64) with the WebRtcAecm_ProcessBlock results of the n_blocks consecutive
    // blocks and return true.  *passthrough is set when the call was served by the start-up copy.
    template <class RunBlocks>
    int32_t Process(const T *near_noisy, const T *near_clean, T *out, size_t n, int16_t ms, RunBlocks &&run_blocks,
                    bool *passthrough = nullptr) {
        int32_t ret = 0;
        if (passthrough) *passthrough = false;
        if (near_noisy == nullptr) return kErrNullPointer;
        if (out == nullptr) return kErrNullPointer;
        if (init_flag_ != kInitCheck) return kErrUninitialized;
        if (n != 80 && n != 160) return kErrBadParameter;
        if (ms < 0) { ms = 0; ret = kWarnBadParameter; }
        else if (ms > 500) { ms = 500; ret = kWarnBadParameter; }
        ms = (int16_t)(ms + 10);
        ms_in_snd_card_buf_ = ms;

        const size_t n_frames = n / kFrameLen;
        const size_t n_blocks_10ms = n_frames / (size_t)mult_;

        if (ec_startup_) {                                                                  // :285-356
            const T *src = near_clean ? near_clean : near_noisy;
            if (out != src) memcpy(out, src, sizeof(T) * n);
            if (passthrough) *passthrough = true;
            const short filled = (short)((short)farend_buf_.available_read() / kFrameLen);
            if (check_buff_size_) {
                check_buf_size_ctr_++;
                if (counter_ == 0) { first_val_ = ms_in_snd_card_buf_; sum_ = 0; }
                const double tol = std::max(0.2 * ms_in_snd_card_buf_, (double)kSampMsNb);
                if (abs(first_val_ - ms_in_snd_card_buf_) < tol) {
                    sum_ = (short)(sum_ + ms_in_snd_card_buf_);
                    counter_++;
                } else {
                    counter_ = 0;
                }
                if (counter_ * n_blocks_10ms >= 6) {
                    buf_size_start_ = (short)std::min<long>((3 * sum_ * mult_) / (counter_ * 40), kBufSizeFrames);
                    check_buff_size_ = 0;
                }
                if (check_buf_size_ctr_ * n_blocks_10ms > 50) {
                    buf_size_start_ = (short)std::min<long>((3 * ms_in_snd_card_buf_ * mult_) / 40, kBufSizeFrames);
                    check_buff_size_ = 0;
                }
            }
            if (!check_buff_size_) {
                if (filled == buf_size_start_) {
                    ec_startup_ = 0;
                } else if (filled > buf_size_start_) {
                    farend_buf_.MoveReadPtr((int)farend_buf_.available_read() - (int)buf_size_start_ * kFrameLen);
                    ec_startup_ = 0;
                }
            }
            return ret;
        }

        // ---- AECM enabled (:358-397) ----
        // Pass 1: everything that does not depend on the DSP output, in the reference's order: pull
        // far frames, run the buffer-delay estimator, re-block far/near into 64-sample blocks.
        T far_blocks[4 * kBlock], near_blocks[4 * kBlock], clean_blocks[4 * kBlock], out_blocks[4 * kBlock];
        int blocks_of_frame[2] = {0, 0};
        int total_blocks = 0;
        for (size_t i = 0; i < n_frames; ++i) {
            T farend[kFrameLen];
            const short filled = (short)((short)farend_buf_.available_read() / kFrameLen);
            if (filled > 0) {
                farend_buf_.Read(farend, kFrameLen);
                memcpy(farend_old_[i], farend, sizeof farend);                              // keep for underruns (:373)
            } else {
                memcpy(farend, farend_old_[i], sizeof farend);                              // replay the last frame (:376-379)
            }
            if ((i == 0 && samp_freq_ == 8000) || (i == 1 && samp_freq_ == 16000)) EstBufDelay();   // :384-387
            // WebRtcAecm_ProcessFrame (aecm_core.cc:501-572).  The core's 256-sample far delay line
            // (:515-516) is a pass-through because the core's knownDelay is 0 for its whole life
            // (aecm_core.cc:372; the wrapper's knownDelay is never forwarded, echo_control_mobile.cc:392).
            far_frames_.Write(farend, kFrameLen);
            near_frames_.Write(near_noisy + kFrameLen * i, kFrameLen);
            if (near_clean) clean_frames_.Write(near_clean + kFrameLen * i, kFrameLen);
            while (far_frames_.available_read() >= (size_t)kBlock) {
                far_frames_.Read(far_blocks + total_blocks * kBlock, kBlock);
                near_frames_.Read(near_blocks + total_blocks * kBlock, kBlock);
                if (near_clean) clean_frames_.Read(clean_blocks + total_blocks * kBlock, kBlock);
                ++total_blocks;
                ++blocks_of_frame[i];
            }
        }
        // The blocks: WebRtcAecm_ProcessBlock x total_blocks.
        if (total_blocks > 0 &&
            !run_blocks(far_blocks, near_blocks, near_clean ? clean_blocks : nullptr, out_blocks, total_blocks))
            return -1;
        // Pass 2: output side of ProcessFrame (aecm_core.cc:554-569), frame by frame.
        int consumed = 0;
        for (size_t i = 0; i < n_frames; ++i) {
            for (int b = 0; b < blocks_of_frame[i]; ++b, ++consumed) out_frames_.Write(out_blocks + consumed * kBlock, kBlock);
            const int size = (int)out_frames_.available_read();
            if (size < kFrameLen) out_frames_.MoveReadPtr(size - kFrameLen);                // stuff with old samples
            out_frames_.Read(out + kFrameLen * i, kFrameLen);
        }
        return ret;
    }

private:
    void Clear() {
        buf_size_start_ = 0; known_delay_ = 0; counter_ = 0; sum_ = 0; first_val_ = 0; check_buf_size_ctr_ = 0;
        ms_in_snd_card_buf_ = 0; filt_delay_ = 0; time_for_delay_change_ = 0; ec_startup_ = 0; check_buff_size_ = 0;
        delay_change_ = 0; last_delay_diff_ = 0;
        for (auto &row : farend_old_) std::fill(row, row + kFrameLen, zero_);
    }

    void EstBufDelay() {                                                                    // :534-573
        short n_samp_far = (short)farend_buf_.available_read();
        short n_samp_snd_card = (short)(ms_in_snd_card_buf_ * kSampMsNb * mult_);
        short delay_new = (short)(n_samp_snd_card - n_samp_far);
        if (delay_new < kFrameLen) {
            farend_buf_.MoveReadPtr(kFrameLen);
            delay_new = (short)(delay_new + kFrameLen);
        }
        filt_delay_ = (short)std::max(0, (8 * filt_delay_ + 2 * delay_new) / 10);
        const short diff = (short)(filt_delay_ - known_delay_);
        if (diff > 224) {
            if (last_delay_diff_ < 96) time_for_delay_change_ = 0;
            else time_for_delay_change_++;
        } else if (diff < 96 && known_delay_ > 0) {
            if (last_delay_diff_ > 224) time_for_delay_change_ = 0;
            else time_for_delay_change_++;
        } else {
            time_for_delay_change_ = 0;
        }
        last_delay_diff_ = diff;
        if (time_for_delay_change_ > 25) known_delay_ = std::max((int)filt_delay_ - 160, 0);
    }

    void DelayComp() {                                                                      // :575-594
        const int n_samp_far = (int)farend_buf_.available_read();
        const int max_stuff = 10 * kFrameLen;
        const int n_samp_snd_card = ms_in_snd_card_buf_ * kSampMsNb * mult_;
        const int delay_new = n_samp_snd_card - n_samp_far;
        if (delay_new > kFarBufLen - kFrameLen * mult_) {
            int n_add = std::max((n_samp_snd_card >> 1) - n_samp_far, kFrameLen);
            n_add = std::min(n_add, max_stuff);
            farend_buf_.MoveReadPtr(-n_add);
            delay_change_ = 1;
        }
    }

    T zero_;
    // --- AecMobile (echo_control_mobile.cc:42-79) ---
    int samp_freq_ = 0;
    int mult_ = 1;               // core mult (aecm_core.cc:368)
    short buf_size_start_ = 0;
    int known_delay_ = 0;
    T farend_old_[2][kFrameLen];
    short init_flag_ = 0;
    short counter_ = 0, sum_ = 0, first_val_ = 0, check_buf_size_ctr_ = 0;
    short ms_in_snd_card_buf_ = 0, filt_delay_ = 0;
    int time_for_delay_change_ = 0, ec_startup_ = 0, check_buff_size_ = 0, delay_change_ = 0;
    short last_delay_diff_ = 0;
    SampleRing<T> farend_buf_;   // 50 frames of 80 samples (:31-36,98)
    // --- frame adapter rings (aecm_core.cc:183-205): FRAME_LEN + PART_LEN = 144 samples each ---
    SampleRing<T> far_frames_, near_frames_, clean_frames_, out_frames_;
    size_t far_accepted_ = 0;
};

// Gather / scatter schedule of a whole recording processed as n_calls x (BufferFarend, Process) of
// `frame` samples with a constant msInSndCardBuf -- identical for every stream of a batch.
struct RecordingSchedule {
    int n_blocks = 0;                 // WebRtcAecm_ProcessBlock calls the session makes
    std::vector<int32_t> far_map;     // [n_blocks*64] far sample index feeding each block sample, -1 = zero
    std::vector<int32_t> near_map;    // [n_blocks*64] likewise for the near end
    std::vector<int32_t> out_map;     // [n_calls*frame]: >= 0 block-output sample, -1 zero, <= -2 near sample -(v+2)
    int32_t first_error = 0;          // first non-zero return code other than the 12100 warning
    bool warned = false;              // some call returned AECM_BAD_PARAMETER_WARNING
};
RecordingSchedule BuildRecordingSchedule(int fs, int frame, int n_calls, int16_t ms_in_snd_card_buf);

}  // namespace aecm
#endif  // AECM_AMD_SESSION_FLOW_H_
