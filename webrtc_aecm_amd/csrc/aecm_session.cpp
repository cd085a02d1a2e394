#include "aecm_session.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/echo_control_mobile.h"

namespace aecm {
namespace {
constexpr int kFrameLen = 80;                 // FRAME_LEN (aecm_defines.h:17)
constexpr int kFarBufLen = 256;               // FAR_BUF_LEN = PART_LEN4 (aecm_defines.h:25)
constexpr int kBufSizeFrames = 50;            // BUF_SIZE_FRAMES (echo_control_mobile.cc:29)
constexpr int kSampMsNb = 8;                  // samples per ms, narrowband (:37)
constexpr short kInitCheck = 42;              // (:40)
}  // namespace

// ---- SampleRing --------------------------------------------------------------------------------

size_t SampleRing::Write(const int16_t *src, size_t n) {
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

size_t SampleRing::Read(int16_t *dst, size_t n) {
    n = std::min(n, count_);
    const size_t cap = data_.size();
    for (size_t i = 0; i < n; ++i) {
        dst[i] = data_[head_];
        head_ = head_ + 1 == cap ? 0 : head_ + 1;
    }
    count_ -= n;
    return n;
}

int SampleRing::MoveReadPtr(int n) {
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

// ---- Session -----------------------------------------------------------------------------------

Session::Session()
    : farend_buf_(kBufSizeFrames * kFrameLen), far_frames_(kFrameLen + kBlock), near_frames_(kFrameLen + kBlock),
      clean_frames_(kFrameLen + kBlock), out_frames_(kFrameLen + kBlock) {
    memset(farend_old_, 0, sizeof farend_old_);
}

Session *Session::Create() {
    const char *dev_env = getenv("AECM_DEVICE");
    const int device = dev_env ? atoi(dev_env) : 0;
    BatchEngine *engine = BatchEngine::Create(1, device);
    if (!engine) return nullptr;                 // no usable GPU: there is no CPU path to fall back to
    Session *s = new Session();
    s->engine_.reset(engine);
    return s;
}

int32_t Session::Init(int32_t samp_freq) {                                     // echo_control_mobile.cc:142-191
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    samp_freq_ = samp_freq;
    mult_ = samp_freq / 8000;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;               // InitCore + default config
    far_frames_.Reset(); near_frames_.Reset(); clean_frames_.Reset(); out_frames_.Reset();   // aecm_core.cc:375-378
    farend_buf_.Reset();
    init_flag_ = kInitCheck;
    delay_change_ = 1;
    sum_ = 0; counter_ = 0; check_buff_size_ = 1; first_val_ = 0;
    ec_startup_ = 1; buf_size_start_ = 0; check_buf_size_ctr_ = 0; filt_delay_ = 0;
    time_for_delay_change_ = 0; known_delay_ = 0; last_delay_diff_ = 0;
    memset(farend_old_, 0, sizeof farend_old_);
    return 0;
}

int32_t Session::BufferFarendError(const int16_t *farend, size_t n) const {    // :195-213
    if (farend == nullptr) return AECM_NULL_POINTER_ERROR;
    if (init_flag_ != kInitCheck) return AECM_UNINITIALIZED_ERROR;
    if (n != 80 && n != 160) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

int32_t Session::BufferFarend(const int16_t *farend, size_t n) {               // :215-234
    const int32_t err = BufferFarendError(farend, n);
    if (err != 0) return err;
    if (!ec_startup_) DelayComp();
    farend_buf_.Write(farend, n);
    return 0;
}

int32_t Session::SetConfig(int16_t cng_mode, int16_t echo_mode) {              // :410-479
    if (init_flag_ != kInitCheck) return AECM_UNINITIALIZED_ERROR;
    // The reference validates and commits cngMode before it looks at echoMode (:421-428).
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (echo_mode < 0 || echo_mode > 4) {
        if (!engine_->SetCngMode(cng_mode, 0, 1)) return AECM_UNSPECIFIED_ERROR;
        return AECM_BAD_PARAMETER_ERROR;
    }
    return engine_->SetConfig(cng_mode, echo_mode, 0, 1) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t Session::InitEchoPath(const void *path, size_t size_bytes) {           // :481-505
    if (path == nullptr) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (init_flag_ != kInitCheck) return AECM_UNINITIALIZED_ERROR;
    int16_t tmp[kBins];
    memcpy(tmp, path, sizeof tmp);
    return engine_->SetEchoPath(0, tmp) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t Session::GetEchoPath(void *path, size_t size_bytes) {                  // :507-532
    if (path == nullptr) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (init_flag_ != kInitCheck) return AECM_UNINITIALIZED_ERROR;
    int16_t tmp[kBins];
    if (!engine_->GetEchoPath(0, tmp)) return AECM_UNSPECIFIED_ERROR;
    memcpy(path, tmp, sizeof tmp);
    return 0;
}

void Session::EstBufDelay() {                                                   // :534-573
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

void Session::DelayComp() {                                                     // :575-594
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

int32_t Session::Process(const int16_t *near_noisy, const int16_t *near_clean, int16_t *out, size_t n,
                         int16_t ms) {                                          // :236-408
    int32_t ret = 0;
    if (near_noisy == nullptr) return AECM_NULL_POINTER_ERROR;
    if (out == nullptr) return AECM_NULL_POINTER_ERROR;
    if (init_flag_ != kInitCheck) return AECM_UNINITIALIZED_ERROR;
    if (n != 80 && n != 160) return AECM_BAD_PARAMETER_ERROR;
    if (ms < 0) { ms = 0; ret = AECM_BAD_PARAMETER_WARNING; }
    else if (ms > 500) { ms = 500; ret = AECM_BAD_PARAMETER_WARNING; }
    ms = (int16_t)(ms + 10);
    ms_in_snd_card_buf_ = ms;

    const size_t n_frames = n / kFrameLen;
    const size_t n_blocks_10ms = n_frames / (size_t)mult_;

    if (ec_startup_) {                                                          // :285-356
        const int16_t *src = near_clean ? near_clean : near_noisy;
        if (out != src) memcpy(out, src, sizeof(int16_t) * n);
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
    int16_t far_blocks[4 * kBlock], near_blocks[4 * kBlock], clean_blocks[4 * kBlock], out_blocks[4 * kBlock];
    int blocks_of_frame[2] = {0, 0};
    int total_blocks = 0;
    for (size_t i = 0; i < n_frames; ++i) {
        int16_t farend[kFrameLen];
        const short filled = (short)((short)farend_buf_.available_read() / kFrameLen);
        if (filled > 0) {
            farend_buf_.Read(farend, kFrameLen);
            memcpy(farend_old_[i], farend, sizeof farend);                      // keep for underruns (:373)
        } else {
            memcpy(farend, farend_old_[i], sizeof farend);                      // replay the last frame (:376-379)
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
    // The blocks: WebRtcAecm_ProcessBlock x total_blocks on the GPU (one launch).
    if (total_blocks > 0) {
        IoView io{far_blocks, near_blocks, near_clean ? clean_blocks : nullptr, out_blocks,
                  (int64_t)total_blocks * kBlock, kBlock};
        if (!engine_->ProcessBlocksHost(io, total_blocks)) return -1;
    }
    // Pass 2: output side of ProcessFrame (aecm_core.cc:554-569), frame by frame.
    int consumed = 0;
    for (size_t i = 0; i < n_frames; ++i) {
        for (int b = 0; b < blocks_of_frame[i]; ++b, ++consumed) out_frames_.Write(out_blocks + consumed * kBlock, kBlock);
        const int size = (int)out_frames_.available_read();
        if (size < kFrameLen) out_frames_.MoveReadPtr(size - kFrameLen);        // stuff with old samples
        out_frames_.Read(out + kFrameLen * i, kFrameLen);
    }
    return ret;
}

}  // namespace aecm
