// Streaming batch of sessions: S independent WebRtcAecm_* sessions ticking together (one BufferFarend + one
// Process of n samples per tick each, or two of 80 for a session flagged so) -- the shape of a media server mixing
// many calls on a 10 ms clock.  msInSndCardBuf and the call flags are per tick for everybody or per session.
//
// Default form (flow_mode_): the session wrapper and the frame adapter run ON THE DEVICE, per session, as position
// arithmetic on per-session state (aecm_flow_plan.h): a planning kernel with one lane per session, then a tick kernel
// with one wavefront per session (aecm_kernels.h: TickFlowIo).  The host does nothing per session; sessions share
// nothing but the tick.
//
// Earlier form, kept for A/B and selected by AECM_TICK_MODE=lean|fused|three: sessions with the same msInSndCardBuf /
// flag history share ONE SessionFlow that runs on the host in the index domain (sample tags instead of samples; a
// "flow class") and whose decisions are applied to all its members on the device: the audio lives in per-stream rings
// in HBM, each tick is
//   prepare (append far/near to the rings + gather the tick's blocks) -> WebRtcAecm_ProcessBlock x nb
//   -> finish (block outputs into the output ring + assemble the tick's output)
// as three launches, or fused into one launch (one wavefront per session does all of it; "lean" with the sources as
// runs of ring positions, "fused" with one source code per sample).  With one class the source decisions travel as
// kernel arguments, with several they sit in a device table indexed by the session's class (at most kMaxFlowClasses).
#ifndef AECM_AMD_SESSIONS_H_
#define AECM_AMD_SESSIONS_H_

#include <stdint.h>

#include <memory>
#include <vector>

#include "aecm_engine.h"
#include "aecm_kernels.h"
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
    //   flags_per_session (host array, S entries, may be null): bit 0 (kNoFarend) = this session gets NO
    //     WebRtcAecm_BufferFarend call in this tick (far-end underrun: its Process replays the last far frame,
    //     reference echo_control_mobile.cc:369-380); its far row is ignored.
    int32_t Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, size_t n,
                 int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, bool host_pointers);
    //     bit 1 (kSplitCalls, 160-sample ticks only) = this session makes TWO BufferFarend + Process call pairs of 80
    //     samples in this tick instead of one of 160 (the reference treats the two cadences differently:
    //     echo_control_mobile.cc:282-283, 384-385).
    static constexpr uint8_t kNoFarend = 1, kSplitCalls = 2;
    // 0: the session machinery runs on the device (the default), there are no classes and no limit on distinct histories
    int num_flow_classes() const { return flow_mode_ ? 0 : (int)classes_.size(); }

    static constexpr int kMaxFlowClasses = 1024;

private:
    // Sessions whose msInSndCardBuf history is identical share one SessionFlow (run in the index domain on
    // the host).  A class splits when its members present different values in a tick; classes never merge.
    struct FlowClass {
        SessionFlow<int64_t> flow;
        int64_t blocks_done = 0;
        int64_t far_count = 0;   // far samples its jitter buffer has accepted so far = the next far tag
        int16_t ms = 0;          // this tick's msInSndCardBuf
        bool no_far = false;     // this tick: no BufferFarend call
        bool split_calls = false;   // this tick: two calls of 80 samples instead of one of 160
        int64_t born = -1;       // tick at which InitSession created it (-1: from Init); fresh sessions of one tick share a class
        int32_t members = 0;
        FlowClass() : flow(-1) {}
    };
    SessionBatch() {}
    int32_t Regroup(const int16_t *ms_per_session, int16_t ms_uniform, const uint8_t *flags_per_session);
    void DropEmptyClasses();
    int32_t CheckSession(int session) const;
    int32_t AdvanceClass(FlowClass &c, int n, bool has_clean, TickClassEntry *entry, TickLeanEntry *lean, bool *lean_ok, bool *coded_ok, bool *stale);
    enum TickMode { kTickFlow, kTickLean, kTickFused, kTickThreeLaunch };
    int32_t TickFlow(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride, int n, int16_t ms,
                     const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, bool host_pointers);
    bool ResetFlowRows(int first, int count);
    static TickMode ChooseTickMode(int num_streams);
    static constexpr int64_t kRing = 8192;     // >= 4000 (jitter buffer) + 160 + 144 + stale re-reads; power of two
    std::unique_ptr<BatchEngine> engine_;
    std::vector<FlowClass> classes_;
    std::vector<int32_t> class_of_;            // host copy, [S]
    std::vector<int32_t> last_key_;            // per-session (ms | flags << 16) of the previous per-session tick
    int64_t tick_count_ = 0;
    int fs_ = 0;
    // A tick that failed after the host-side flows advanced leaves them out of step with the device rings:
    // every later call is refused (AECM_UNSPECIFIED_ERROR) until Init.
    bool poisoned_ = false;
    bool class_of_dirty_ = false;
    int64_t near_pos_ = 0;
    int16_t *far_ring_ = nullptr, *near_ring_ = nullptr, *out_ring_ = nullptr;   // [S][kRing]
    int16_t *clean_ring_ = nullptr;   // [S][kRing], allocated by the first tick that carries a clean near-end
    int16_t *blk_ = nullptr;          // [4][S][4*64] gathered far / near / clean blocks and block outputs of a tick
    int16_t *io_dev_ = nullptr;       // [4][S][160] staging when the caller passes host pointers
    int32_t *class_of_dev_ = nullptr, *blocks_per_stream_dev_ = nullptr;          // [S] each
    TickClassEntry *table_dev_ = nullptr, *table_host_ = nullptr;                 // [kMaxFlowClasses], host copy pinned
    TickLeanEntry *lean_dev_ = nullptr, *lean_host_ = nullptr;                    // the same ticks as run descriptions
    // Device-resident session machinery (the default tick form, aecm_flow_plan.h): per-session wrapper state, framed far
    // stream and far-end replay rows in HBM; the classes above are then unused.
    bool flow_mode_ = true;
    int32_t *flow_state_ = nullptr;            // [kFlowFieldsUsed][S]
    int32_t *flow_plans_ = nullptr;            // [S][kFlowPlanWords]: this tick's plan of every session
    int16_t *far_frames_ = nullptr;            // [S][kFlowFarFrameRing]
    int16_t *far_old_ = nullptr;               // [S][2 * 80]
    int16_t *ms_host_ = nullptr, *ms_dev_ = nullptr;          // [S] per-session msInSndCardBuf of the tick: pinned host memory and
    uint8_t *flags_host_ = nullptr, *flags_dev_ = nullptr;    // [S] per-session flags          its address on the device (read in place)
    int device_ = 0;
};

}  // namespace aecm
#endif  // AECM_AMD_SESSIONS_H_
