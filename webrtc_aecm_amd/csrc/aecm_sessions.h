// Streaming batch of sessions: S independent WebRtcAecm_* sessions that all see the same call
// pattern (one BufferFarend + one Process of n samples per tick, same msInSndCardBuf) -- the shape of
// a media server mixing many calls on a 10 ms clock.
//
// The session wrapper and the frame adapter only move samples (aecm_session_flow.h), so ONE
// SessionFlow runs on the host in the index domain (64-bit absolute sample tags) and its decisions are
// applied to all streams on the device: the audio lives in per-stream rings in HBM, each tick is
//   prepare (append far/near to the rings + gather the tick's blocks) -> WebRtcAecm_ProcessBlock x nb
//   -> finish (block outputs into the output ring + assemble the tick's output): three launches, the
//   per-sample source decisions travel as kernel arguments.
#ifndef AECM_AMD_SESSIONS_H_
#define AECM_AMD_SESSIONS_H_

#include <stdint.h>

#include <memory>
#include <vector>

#include "aecm_engine.h"
#include "aecm_session_flow.h"

namespace aecm {

class SessionBatch {
public:
    static SessionBatch *Create(int num_streams, int device_id);
    ~SessionBatch();
    int num_streams() const { return engine_->num_streams(); }
    BatchEngine *engine() { return engine_.get(); }

    int32_t Init(int32_t samp_freq);
    int32_t SetConfig(int16_t cng_mode, int16_t echo_mode);
    // One tick for every session; far/near/clean/out are [S][>= n] with the given stream stride, device
    // or host pointers; clean (WebRtcAecm_Process's nearendClean) may be null.  Returns the code each
    // session's WebRtcAecm_Process would return.
    int32_t Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, int n,
                 int16_t ms, bool host_pointers);

private:
    SessionBatch() : flow_(-1) {}
    static constexpr int64_t kRing = 8192;     // >= 4000 (jitter buffer) + 160 + 144 + stale re-reads; power of two
    std::unique_ptr<BatchEngine> engine_;
    SessionFlow<int64_t> flow_;
    int64_t far_pos_ = 0, near_pos_ = 0, blocks_done_ = 0;
    int16_t *far_ring_ = nullptr, *near_ring_ = nullptr, *out_ring_ = nullptr;   // [S][kRing]
    int16_t *clean_ring_ = nullptr;   // [S][kRing], allocated by the first tick that carries a clean near-end
    int16_t *blk_ = nullptr;          // [4][S][4*64] gathered far / near / clean blocks and block outputs of a tick
    int16_t *io_dev_ = nullptr;       // [4][S][160] staging when the caller passes host pointers
    int device_ = 0;
};

}  // namespace aecm
#endif  // AECM_AMD_SESSIONS_H_
