// Streaming batch of sessions: S independent WebRtcAecm_* sessions ticking together (one BufferFarend + one
// Process of n samples per tick each, or two of 80 for a session flagged so) -- the shape of a media server mixing
// many calls on a 10 ms clock.  msInSndCardBuf and the call flags are per tick for everybody or per session.
//
// The session wrapper and the frame adapter of the reference (echo_control_mobile.cc, aecm_core.cc:501-572) run ON THE
// DEVICE, per session, as position arithmetic on per-session state (aecm_flow_plan.h): a planning kernel with one lane
// per session, then a tick kernel with one wavefront per session (aecm_kernels.h: TickFlowIo).  The audio lives in
// per-session rings in HBM.  The host does nothing per session; sessions share nothing but the tick.
#ifndef AECM_AMD_SESSIONS_H_
#define AECM_AMD_SESSIONS_H_

#include <stdint.h>

#include <memory>

#include "aecm_engine.h"
#include "aecm_kernels.h"

namespace aecm {

class SessionBatch {
public:
    static SessionBatch *Create(int num_streams, int device_id);
    ~SessionBatch();
    int num_streams() const { return engine_->num_streams(); }
    BatchEngine *engine() { return engine_.get(); }

    int32_t Init(int32_t samp_freq);
    int32_t SetConfig(int16_t cng_mode, int16_t echo_mode);
    // Per-session control (a media server recycling one slot while the others keep running), reference
    // echo_control_mobile.cc:142-191 (WebRtcAecm_Init), :410-479 (set_config), :481-532 (Init/GetEchoPath).
    int32_t InitSession(int session);
    int32_t SetConfigSession(int session, int16_t cng_mode, int16_t echo_mode);
    int32_t InitEchoPathSession(int session, const void *path, size_t size_bytes);
    int32_t GetEchoPathSession(int session, void *path, size_t size_bytes);
    // One tick for every session; far/near/clean/out are [S][>= n] with the given stream stride, device
    // or host pointers; clean (WebRtcAecm_Process's nearendClean) may be null.
    //   ms_per_session == nullptr: every session gets `ms`; the return value is the code each session's
    //     WebRtcAecm_Process would return.
    //   ms_per_session != nullptr (host array, S entries): session s gets ms_per_session[s]; codes (host,
    //     S entries, may be null) receives each session's code, the return value is 0 or the first
    //     non-zero code.
    //   flags_per_session (host array, S entries, may be null):
    //     bit 0 (kNoFarend) = this session gets NO WebRtcAecm_BufferFarend call in this tick (far-end underrun: its
    //     Process replays the last far frame, reference echo_control_mobile.cc:369-380); its far row is ignored.
    //     bit 1 (kSplitCalls, 160-sample ticks only) = this session makes TWO BufferFarend + Process call pairs of 80
    //     samples in this tick instead of one of 160 (the reference treats the two cadences differently:
    //     echo_control_mobile.cc:282-283, 384-385).
    //   flags: the same bits for every session when flags_per_session is null.
    //   far may be null when no session makes a BufferFarend call in this tick (kNoFarend for everybody).
    int32_t Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, size_t n,
                 int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, bool host_pointers, int flags = 0);
    // The same tick, enqueued on the object's stream without waiting for it (device pointers, or host memory the device
    // can address: see RegisterHostAudio): the two launches of tick t + 1 may be issued while tick t still runs.
    // wait_event (hipEvent_t, may be null): the tick's kernels wait for it first (the caller's producer of far / near);
    // done_event (hipEvent_t, may be null): recorded behind the tick (the caller's consumer of out waits for it).
    int32_t TickAsync(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, size_t n,
                      int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, void *wait_event,
                      void *done_event, int flags = 0);
    // Far-end bursts: WebRtcAecm_BufferFarend calls that come without a WebRtcAecm_Process (reference
    // echo_control_mobile.cc:215-234) -- session s makes calls_per_session[s] (host array, S entries <= calls; null:
    // everybody makes `calls`) consecutive calls of n samples on far[s][c * n .. + n).  Together with ticks whose sessions are
    // flagged kNoFarend (far may then be null) any interleaving of the two reference calls can be expressed per session.
    // Enqueued on the object's stream like a tick (device pointers; host pointers are staged and the call waits).
    int32_t BufferFarend(const int16_t *far, int64_t stream_stride, size_t n, int32_t calls, const uint8_t *calls_per_session, bool host_pointers,
                         bool wait, void *wait_event, void *done_event);
    int32_t Synchronize();
    static constexpr uint8_t kNoFarend = kFlowNoFarend, kSplitCalls = kFlowSplitCalls;

    // Snapshot of ONE live session (checkpoint; migration into another object, on another GPU): everything the reference
    // keeps per instance -- AecMobile's wrapper members and jitter buffer (echo_control_mobile.cc:42-79), the core's frame
    // buffers (aecm_core.h:41-60) and the core state proper (the block stream's snapshot, aecm_engine.h) -- in a form that
    // does not depend on the object it came from:
    //   header (32 B) | block-stream blob (BatchEngine::kStateBytes) | wrapper state kFlowWords int32 | far ring kRing int16
    //   (indexed by the session's own far-stream positions) | the last kOutTail block-output samples before F_BLK_POS |
    //   the last kNearTail near-end (and clean near-end) samples ticked (the only ones a later block can still ask for:
    //   fewer than one block is ever pending) | framed-far ring | the two replay rows.
    // The near-end rings are indexed by the OBJECT's tick position, so their tails are re-placed at the importing object's.
    // ImportSession validates everything FlowTick / the tick kernel turn into a count or an index (aecm_flow_plan.h:
    // FlowStateDefect) and the block-stream blob like ImportState does; a refused blob changes nothing.  The session's rate
    // must be the object's.  Both calls wait for the ticks enqueued so far.
    static constexpr int kOutTail = 256, kNearTail = 64;
    static constexpr size_t kSessionHeaderBytes = 32;
    static constexpr size_t kSessionBytes = kSessionHeaderBytes + BatchEngine::kStateBytes + kFlowWords * 4 + kFlowFarRing * 2 + kOutTail * 2 +
                                            2 * kNearTail * 2 + kFlowFarFrameRing * 2 + 2 * kFlowFrame * 2;
    int32_t ExportSession(int session, void *buf);
    int32_t ImportSession(int session, const void *buf);

private:
    SessionBatch() {}
    int32_t CheckSession(int session) const;
    bool ResetFlowRows(int first, int count);
    static constexpr int64_t kRing = kFlowFarRing;   // >= 4000 (jitter buffer) + a tick + the window a replayed frame may age in
    std::unique_ptr<BatchEngine> engine_;
    int fs_ = 0;                               // 0: not initialised
    // A tick that failed on the device leaves the rings and the wrapper state out of step: every later call is refused
    // (AECM_UNSPECIFIED_ERROR) until Init.
    bool poisoned_ = false;
    int64_t near_pos_ = 0;                     // near-end samples ticked so far = ring position of the next tick's first one
    int16_t *far_ring_ = nullptr, *near_ring_ = nullptr, *out_ring_ = nullptr;   // [S][kRing]
    int16_t *clean_ring_ = nullptr;            // [S][kRing], allocated by the first tick that carries a clean near-end
    int16_t *io_dev_ = nullptr;                // [4][S][160] staging when the caller passes host pointers
    int32_t *flow_state_ = nullptr;            // [kFlowFieldsUsed][S] wrapper state, field-major
    int32_t *flow_plans_ = nullptr;            // [S][kFlowPlanWords]: this tick's plan of every session
    int16_t *far_frames_ = nullptr;            // [S][kFlowFarFrameRing]
    int16_t *far_old_ = nullptr;               // [S][2 * 80]
    // Per-session msInSndCardBuf / flags of a tick: pinned host memory the planning kernel reads in place.  Two slots used
    // alternately, each guarded by an event recorded behind the planning launch that reads it, so that an asynchronous
    // tick can fill the other slot while the previous tick is still in flight.
    static constexpr int kArgSlots = 2;
    int16_t *ms_host_[kArgSlots] = {nullptr, nullptr}, *ms_dev_[kArgSlots] = {nullptr, nullptr};          // [S] each
    uint8_t *flags_host_[kArgSlots] = {nullptr, nullptr}, *flags_dev_[kArgSlots] = {nullptr, nullptr};    // [S] each
    hipEvent_t slot_read_[kArgSlots] = {nullptr, nullptr};
    bool slot_busy_[kArgSlots] = {false, false};
    int slot_ = 0;
    int32_t Enqueue(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, size_t n,
                    int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, bool host_pointers,
                    void *wait_event, void *done_event, int flags);
    bool EventUsable(void *ev) const;
    int AcquireArgSlot(bool needed, bool *ok);
    int32_t Fail();
    int device_ = 0;
};

}  // namespace aecm
#endif  // AECM_AMD_SESSIONS_H_
