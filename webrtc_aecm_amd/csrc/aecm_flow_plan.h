// The session wrapper and the frame adapter of ONE session as position arithmetic, for the device.
//
// Between the public ABI and WebRtcAecm_ProcessBlock the reference only MOVES samples (aecm_session_flow.h restates
// that machinery generically; reference aecm/echo_control_mobile.cc:215-408, 534-594, aecm/aecm_core.cc:501-572,
// aecm/ring_buffer.c).  Every one of its buffers is written front to back, so a buffer never has to be simulated
// sample by sample: the jitter buffer is the pair (samples accepted so far, samples consumed so far) over an absolute
// "far stream", the frame rings are the pair (samples framed, samples blocked) over the stream of 80-sample frames,
// the output ring is the pair (block outputs produced, output samples delivered).  A tick of a session is then a
// handful of integer updates -- FlowTick below -- that yield a PLAN: which input samples the jitter buffer accepts
// and where they go, where each 80-sample far frame comes from (the far stream or the replay copy of an underrun),
// how many blocks each frame completes, and where each output frame is read.  On the device a tick is two launches:
// aecm_flow_plan_kernel runs FlowTick with ONE LANE PER SESSION (64 sessions per wavefront; the state is stored field-
// major so that the lanes' accesses coalesce) and leaves each session's plan as kFlowPlanWords words; aecm_tick_flow_kernel
// (one wavefront per session) reads its plan into scalar registers and moves the samples it names around the blocks.
// Nothing about a session lives on the host, so every session may have its own msInSndCardBuf and call pattern.
//
// All positions are uint32 counters that wrap; only differences and positions modulo a power-of-two ring are used.
#ifndef AECM_AMD_FLOW_PLAN_H_
#define AECM_AMD_FLOW_PLAN_H_

#include <stdint.h>

#include <initializer_list>

#if defined(__HIPCC__)
#define AECM_FLOW_HD __host__ __device__ __forceinline__
#else
#define AECM_FLOW_HD inline
#endif

namespace aecm {

// One session's wrapper state: kFlowFieldsUsed int32, stored field-major on the device (state[field * S + session],
// SessionBatch::flow_state_).
enum FlowField : int {
    F_BUF_SIZE_START = 0,       // AecMobile::bufSizeStart            echo_control_mobile.cc:42-79
    F_KNOWN_DELAY,              // knownDelay
    F_COUNTER,                  // counter
    F_SUM,                      // sum
    F_FIRST_VAL,                // firstVal
    F_CHECK_BUF_SIZE_CTR,       // checkBufSizeCtr
    F_MS,                       // msInSndCardBuf (after the +10 of :266)
    F_FILT_DELAY,               // filtDelay
    F_TIME_FOR_DELAY_CHANGE,    // timeForDelayChange
    F_EC_STARTUP,               // ECstartup
    F_CHECK_BUFF_SIZE,          // checkBuffSize
    F_DELAY_CHANGE,             // delayChange
    F_LAST_DELAY_DIFF,          // lastDelayDiff
    F_FAR_RP,                   // far stream: samples the jitter buffer has handed out (its read pointer; moves both ways)
    F_FAR_WP,                   // far stream: samples the jitter buffer has accepted
    F_FRM_POS,                  // frame stream: samples written into the frame rings (far and near alike)
    F_BLK_POS,                  // frame stream: samples consumed as blocks = 64 * blocks processed = output stream written
    F_OUT_RP,                   // output stream: read pointer of the output frame ring (moves back when stuffing)
    F_RUN_POS,                  // frame stream: from this position on the framed far end is ONE run of the far stream ...
    F_RUN_DELTA,                // ... namely frame-stream position x = far-stream position x + F_RUN_DELTA
    F_FF_VALID,                 // the framed-far ring holds the pending frame-stream samples [F_BLK_POS, F_FRM_POS)
    F_OLD_POS0, F_OLD_POS1,     // farendOld[i] (the frame replayed on an underrun) = far stream [F_OLD_POS_i, + 80) ...
    F_OLD_ROW0, F_OLD_ROW1,     // ... unless 1: it has been copied to replay row i (initially: the zeroed row)
    kFlowFieldsUsed,
    kFlowWords = 32
};

constexpr int kFlowFrame = 80;                      // FRAME_LEN
constexpr int kFlowBlock = 64;                      // PART_LEN
constexpr int kFlowJitterCapacity = 50 * 80;        // kBufSizeSamp = BUF_SIZE_FRAMES * FRAME_LEN (:29-36)
constexpr int kFlowFarFrameRing = 256;              // ring of the framed far stream on the device (>= 143 + 80, power of two)
constexpr int kFlowNoFarend = 1, kFlowSplitCalls = 2;   // = SessionBatch::kNoFarend / kSplitCalls
constexpr int kFlowFarRing = 8192;                  // far ring of the device: positions older than this many accepted samples are gone
constexpr int kFlowOldAge = kFlowFarRing - 5 * kFlowFrame;   // a replay frame this far behind the write position moves to its row

struct FlowFarPiece {          // far_in[src, src + count) -> far stream positions [pos, pos + count)
    int32_t src, count;
    uint32_t pos;
};
struct FlowFrame {             // one 80-sample frame of the tick (input samples [80 f, 80 f + 80))
    int32_t active;            // 0: its call was served by the start-up copy (out = clean or noisy near-end, :285-291)
    int32_t far_from_stream;   // 1: far frame = far stream [far_pos, far_pos + 80) (a fresh frame, or the replay of one that is
                               // still in the far ring); 0: replay row old_idx
    uint32_t far_pos;
    int32_t old_idx;
    uint32_t frm_pos;          // where the frame lands in the frame stream
    int32_t n_blocks;          // blocks this frame completes
    uint32_t out_pos;          // output stream position of this frame's 80 output samples
};
struct FlowPlan {
    FlowFarPiece far[2];       // per call
    FlowFrame frame[2];
    int32_t n_calls, n_frames, n_blocks;
    uint32_t blk_pos0;         // frame-stream / output-stream position of the tick's first block
    uint32_t near_base;        // frame-stream position x of the near end sits at near-ring position near_base + x
    // direct: every far sample the tick's blocks consume belongs to one run of the far stream (no replay, re-read or skip
    // since the oldest of them was framed): block sample x is far-stream sample x + far_delta, the blocks fetch from the
    // far ring itself and nothing is framed.  Otherwise the tick's frames go through the framed-far ring, preceded --
    // when the previous ticks were direct -- by the left_count pending samples [blk_pos0, ..) = far stream + left_delta.
    int32_t direct;
    uint32_t far_delta;
    int32_t left_count;
    uint32_t left_delta;
    int32_t spill[2];          // replay frame i is about to leave the far ring: far stream [spill_pos[i], + 80) -> replay row i
    uint32_t spill_pos[2];
};

struct FlowRegs {              // FlowField values in registers
    int32_t v[kFlowFieldsUsed];
};

AECM_FLOW_HD int32_t FlowAsShort(int32_t x) { return (int32_t)(int16_t)x; }
AECM_FLOW_HD int32_t FlowMin(int32_t a, int32_t b) { return a < b ? a : b; }
AECM_FLOW_HD int32_t FlowMax(int32_t a, int32_t b) { return a > b ? a : b; }

// State after WebRtcAecm_Init (echo_control_mobile.cc:142-191): everything 0 except the three start-up flags (and the
// flags that say where the device keeps things).
AECM_FLOW_HD bool FlowFieldStartsAtOne(int f) {
    return f == F_DELAY_CHANGE || f == F_CHECK_BUFF_SIZE || f == F_EC_STARTUP || f == F_FF_VALID || f == F_OLD_ROW0 || f == F_OLD_ROW1;
}
AECM_FLOW_HD void FlowInit(int32_t words[kFlowWords]) {
    for (int i = 0; i < kFlowWords; ++i) words[i] = i < kFlowFieldsUsed && FlowFieldStartsAtOne(i) ? 1 : 0;
}

// Is this wrapper state one FlowTick and the tick kernel may run on (WebRtcAecmSessions_ImportSession)?  Everything they
// turn into a loop count, a lane count or a 4-bit field of the plan: the pending frame-stream samples (< one block: the block
// loop of a frame and the plan's left_count), the jitter buffer's fill, the output ring's fill, the flags; and everything
// that names far-ring samples a later tick will read (replay frames, pending samples of direct ticks) must name samples the
// ring still holds.  Positions themselves are free-running counters: any value is a position.  tests/sim/sim_flow.cpp checks
// that every state a session passes through is accepted.  Returns 0 or the 1-based index of the offending field.
AECM_FLOW_HD int FlowStateDefect(const int32_t v[kFlowWords]) {
    for (int f : {F_EC_STARTUP, F_CHECK_BUFF_SIZE, F_DELAY_CHANGE, F_FF_VALID, F_OLD_ROW0, F_OLD_ROW1})
        if (v[f] != 0 && v[f] != 1) return f + 1;
    if (v[F_BUF_SIZE_START] < 0 || v[F_BUF_SIZE_START] > 50) return F_BUF_SIZE_START + 1;          // BUF_SIZE_FRAMES (:29-36, :322,:332)
    if (v[F_MS] < 0 || v[F_MS] > 510) return F_MS + 1;                                                // clamped to [0, 500], + 10 (:258-266)
    for (int f : {F_KNOWN_DELAY, F_COUNTER, F_SUM, F_FIRST_VAL, F_CHECK_BUF_SIZE_CTR, F_FILT_DELAY, F_LAST_DELAY_DIFF})
        if (v[f] < -32768 || v[f] > 32767) return f + 1;                                              // the reference's short members
    const uint32_t readable = (uint32_t)v[F_FAR_WP] - (uint32_t)v[F_FAR_RP];
    if (readable > (uint32_t)kFlowJitterCapacity) return F_FAR_RP + 1;
    const uint32_t pending = (uint32_t)v[F_FRM_POS] - (uint32_t)v[F_BLK_POS];
    if (pending >= (uint32_t)kFlowBlock) return F_BLK_POS + 1;
    const uint32_t out_fill = (uint32_t)v[F_BLK_POS] - (uint32_t)v[F_OUT_RP];
    if (out_fill > (uint32_t)(kFlowFrame + kFlowBlock)) return F_OUT_RP + 1;                          // the output frame ring holds FRAME_LEN + PART_LEN (aecm_core.cc:204-205)
    // A replay frame that still lives in the far ring only has not been lapped there: FlowTick / FlowBurstBegin move it to
    // its row before the write position gets further than kFlowOldAge + one tick ahead of it.
    for (int i = 0; i < 2; ++i)
        if (!v[F_OLD_ROW0 + i] && (uint32_t)v[F_FAR_WP] - (uint32_t)v[F_OLD_POS0 + i] > (uint32_t)(kFlowOldAge + 2 * kFlowFrame)) return F_OLD_POS0 + i + 1;
    // Pending samples that were left in the far ring by direct ticks (F_FF_VALID == 0) are one run of the far stream that
    // ends before the jitter buffer's read pointer and is as young as anything the jitter buffer can still hand out
    // (the read pointer never falls back by more than the buffer's capacity; a frame and the pending samples lie before it).
    if (!v[F_FF_VALID]) {
        const uint32_t behind = (uint32_t)v[F_FAR_WP] - ((uint32_t)v[F_BLK_POS] + (uint32_t)v[F_RUN_DELTA]);    // write position - far position of the first pending sample
        if (behind < pending || behind > (uint32_t)(kFlowJitterCapacity + 2 * kFlowFrame + kFlowBlock)) return F_RUN_DELTA + 1;
        if ((uint32_t)v[F_BLK_POS] - (uint32_t)v[F_RUN_POS] > (uint32_t)(4 * kFlowBlock)) return F_RUN_POS + 1;  // FlowTick keeps it at the last tick's first block
    }
    for (int f = kFlowFieldsUsed; f < kFlowWords; ++f)
        if (v[f] != 0) return f + 1;
    return 0;
}

// WebRtc_MoveReadPtr of the jitter buffer (ring_buffer.c:176-211): clamped to what is readable / free.
AECM_FLOW_HD void FlowMoveFarReadPtr(FlowRegs &s, int32_t n) {
    const int32_t readable = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
    const int32_t free_elems = kFlowJitterCapacity - readable;
    n = FlowMin(n, readable);
    n = FlowMax(n, -free_elems);
    s.v[F_FAR_RP] = (int32_t)((uint32_t)s.v[F_FAR_RP] + (uint32_t)n);
}

// WebRtcAecm_DelayComp (:575-594).
AECM_FLOW_HD void FlowDelayComp(FlowRegs &s, int mult) {
    const int32_t n_samp_far = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
    const int32_t n_samp_snd_card = s.v[F_MS] * 8 * mult;
    const int32_t delay_new = n_samp_snd_card - n_samp_far;
    if (delay_new > 256 - kFlowFrame * mult) {
        int32_t n_add = FlowMax((n_samp_snd_card >> 1) - n_samp_far, kFlowFrame);
        n_add = FlowMin(n_add, 10 * kFlowFrame);
        FlowMoveFarReadPtr(s, -n_add);
        s.v[F_DELAY_CHANGE] = 1;
    }
}

// One WebRtcAecm_BufferFarend call of len samples (:215-234): delay compensation once the start-up phase is over, then
// the jitter buffer takes what fits -- a full buffer drops the rest (ring_buffer.c:142-150).  Returns the number of
// samples accepted; they land at far-stream positions [F_FAR_WP before the call, + accepted).
AECM_FLOW_HD int32_t FlowFarendCall(FlowRegs &s, int mult, int len) {
    if (!s.v[F_EC_STARTUP]) FlowDelayComp(s, mult);
    const int32_t readable = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
    const int32_t accepted = FlowMin(len, kFlowJitterCapacity - readable);
    s.v[F_FAR_WP] = (int32_t)((uint32_t)s.v[F_FAR_WP] + (uint32_t)accepted);
    return accepted;
}

// WebRtcAecm_BufferFarend calls that come WITHOUT a WebRtcAecm_Process (a far-end burst: the network delivered several
// frames between two ticks; WebRtcAecmSessions_BufferFarend).  Each is FlowFarendCall; the only other thing a burst has to
// mind is the far ring of the device, which is longer than the jitter buffer but finite: a replay frame that lives in the
// ring only (F_OLD_ROWi == 0) moves to its row before the burst's samples can lap it.  FlowBurstBegin decides that from
// an upper bound of what `calls` calls of `len` samples can add (what is free now: delay compensation only ever makes
// the buffer fuller), so the copy can be made before the first sample of the burst is written; a row taking over a
// little early reads the same samples.  The wrapper fields a burst reads / writes: FlowBurstReads / FlowBurstWrites.
struct FlowBurst {
    int32_t spill[2];
    uint32_t spill_pos[2];
};
AECM_FLOW_HD void FlowBurstBegin(FlowRegs &s, int len, int calls, FlowBurst &b) {
    const int32_t readable = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
    const int64_t offered = (int64_t)len * calls;
    const int32_t most = (int32_t)(offered < (int64_t)(kFlowJitterCapacity - readable) ? offered : (int64_t)(kFlowJitterCapacity - readable));
    for (int i = 0; i < 2; ++i) {
        b.spill_pos[i] = (uint32_t)s.v[F_OLD_POS0 + i];
        b.spill[i] = !s.v[F_OLD_ROW0 + i] && (int32_t)((uint32_t)s.v[F_FAR_WP] + (uint32_t)most - (uint32_t)s.v[F_OLD_POS0 + i]) > kFlowOldAge;
        if (b.spill[i]) s.v[F_OLD_ROW0 + i] = 1;
    }
}
template <class Fn>
AECM_FLOW_HD void FlowBurstReads(Fn &&fn) {
    for (int f : {F_FAR_RP, F_FAR_WP, F_MS, F_EC_STARTUP, F_DELAY_CHANGE, F_OLD_POS0, F_OLD_POS1, F_OLD_ROW0, F_OLD_ROW1}) fn(f);
}
template <class Fn>
AECM_FLOW_HD void FlowBurstWrites(Fn &&fn) {
    for (int f : {F_FAR_RP, F_FAR_WP, F_DELAY_CHANGE, F_OLD_ROW0, F_OLD_ROW1}) fn(f);
}

// WebRtcAecm_EstBufDelay (:534-573).
AECM_FLOW_HD void FlowEstBufDelay(FlowRegs &s, int mult) {
    const int32_t n_samp_far = FlowAsShort((int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]));
    const int32_t n_samp_snd_card = FlowAsShort(s.v[F_MS] * 8 * mult);
    int32_t delay_new = FlowAsShort(n_samp_snd_card - n_samp_far);
    if (delay_new < kFlowFrame) {
        FlowMoveFarReadPtr(s, kFlowFrame);
        delay_new = FlowAsShort(delay_new + kFlowFrame);
    }
    const int32_t acc = 8 * s.v[F_FILT_DELAY] + 2 * delay_new;            // C division truncates towards zero; negative -> max(0, .) = 0
    s.v[F_FILT_DELAY] = FlowAsShort(acc <= 0 ? 0 : (int32_t)((uint32_t)acc / 10u));
    const int32_t diff = FlowAsShort(s.v[F_FILT_DELAY] - s.v[F_KNOWN_DELAY]);
    if (diff > 224) {
        s.v[F_TIME_FOR_DELAY_CHANGE] = s.v[F_LAST_DELAY_DIFF] < 96 ? 0 : s.v[F_TIME_FOR_DELAY_CHANGE] + 1;
    } else if (diff < 96 && s.v[F_KNOWN_DELAY] > 0) {
        s.v[F_TIME_FOR_DELAY_CHANGE] = s.v[F_LAST_DELAY_DIFF] > 224 ? 0 : s.v[F_TIME_FOR_DELAY_CHANGE] + 1;
    } else {
        s.v[F_TIME_FOR_DELAY_CHANGE] = 0;
    }
    s.v[F_LAST_DELAY_DIFF] = diff;
    if (s.v[F_TIME_FOR_DELAY_CHANGE] > 25) s.v[F_KNOWN_DELAY] = FlowMax(s.v[F_FILT_DELAY] - 160, 0);
}

// The start-up phase of WebRtcAecm_Process (:285-356), n_frames 80-sample frames in this call.
AECM_FLOW_HD void FlowStartup(FlowRegs &s, int mult, int n_frames) {
    const int32_t n_blocks_10ms = n_frames / mult;
    const int32_t avail = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
    const int32_t filled = FlowAsShort(FlowAsShort(avail) / kFlowFrame);
    const int32_t ms = s.v[F_MS];
    if (s.v[F_CHECK_BUFF_SIZE]) {
        s.v[F_CHECK_BUF_SIZE_CTR] = FlowAsShort(s.v[F_CHECK_BUF_SIZE_CTR] + 1);
        if (s.v[F_COUNTER] == 0) {
            s.v[F_FIRST_VAL] = ms;
            s.v[F_SUM] = 0;
        }
        // |firstVal - ms| < max(0.2 * ms, 8.0) in doubles (:310-312): 0.2 * ms rounds to ms / 5 exactly whenever that is an
        // integer (ms <= 510), so the comparison with an integer is d < 8 || 5 d < ms.
        int32_t d = s.v[F_FIRST_VAL] - ms;
        d = d < 0 ? -d : d;
        if (d < 8 || 5 * d < ms) {
            s.v[F_SUM] = FlowAsShort(s.v[F_SUM] + ms);
            s.v[F_COUNTER] = FlowAsShort(s.v[F_COUNTER] + 1);
        } else {
            s.v[F_COUNTER] = 0;
        }
        // Both comparisons are made in size_t in the reference (short counter * size_t nBlocks10ms, :320,:330): a counter
        // that has wrapped negative -- more than 32 767 start-up calls of 80 samples at 16 kHz, where nBlocks10ms is 0 and
        // neither limit can fire -- is a huge unsigned product there.
        if ((uint64_t)(int64_t)s.v[F_COUNTER] * (uint64_t)n_blocks_10ms >= 6u) {
            s.v[F_BUF_SIZE_START] = FlowAsShort(FlowMin((3 * s.v[F_SUM] * mult) / (s.v[F_COUNTER] * 40), 50));
            s.v[F_CHECK_BUFF_SIZE] = 0;
        }
        if ((uint64_t)(int64_t)s.v[F_CHECK_BUF_SIZE_CTR] * (uint64_t)n_blocks_10ms > 50u) {
            s.v[F_BUF_SIZE_START] = FlowAsShort(FlowMin((3 * ms * mult) / 40, 50));
            s.v[F_CHECK_BUFF_SIZE] = 0;
        }
    }
    if (!s.v[F_CHECK_BUFF_SIZE]) {
        if (filled == s.v[F_BUF_SIZE_START]) {
            s.v[F_EC_STARTUP] = 0;
        } else if (filled > s.v[F_BUF_SIZE_START]) {
            FlowMoveFarReadPtr(s, avail - s.v[F_BUF_SIZE_START] * kFlowFrame);
            s.v[F_EC_STARTUP] = 0;
        }
    }
}

// One tick: [WebRtcAecm_BufferFarend] + WebRtcAecm_Process of n samples (80 or 160), or two such pairs of 80 samples
// (kFlowSplitCalls, n = 160).  ms = the caller's msInSndCardBuf, near_pos = near-ring position of the tick's first
// near-end sample.  Advances s and fills the plan.
AECM_FLOW_HD void FlowTick(FlowRegs &s, int fs, int n, int ms, int flags, uint32_t near_pos, FlowPlan &p) {
    const int mult = fs == 16000 ? 2 : 1;
    const bool split = (flags & kFlowSplitCalls) != 0 && n == 2 * kFlowFrame;
    const int n_calls = split ? 2 : 1;
    const int len = split ? kFlowFrame : n;
    const int frames_per_call = len / kFlowFrame;
    p.n_calls = n_calls;
    p.n_frames = n / kFlowFrame;
    p.n_blocks = 0;
    p.blk_pos0 = (uint32_t)s.v[F_BLK_POS];
    p.near_base = 0;
    p.direct = 0;
    p.far_delta = p.left_delta = 0;
    p.left_count = 0;
    const uint32_t frm_start = (uint32_t)s.v[F_FRM_POS], run_delta_start = (uint32_t)s.v[F_RUN_DELTA];
    // A replay frame that still lives in the far ring only is copied to its row before the ring's write position laps it.
    // The row takes over from the NEXT tick on: a replay in this tick still reads the ring, so nothing in the tick depends
    // on the copy.
    int32_t refreshed[2] = {0, 0};
    for (int i = 0; i < 2; ++i) {
        p.spill_pos[i] = (uint32_t)s.v[F_OLD_POS0 + i];
        p.spill[i] = !s.v[F_OLD_ROW0 + i] && (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_OLD_POS0 + i]) > kFlowOldAge;
    }
    for (int f = 0; f < 2; ++f) p.frame[f] = FlowFrame{0, 0, 0u, 0, 0u, 0, 0u};
    for (int c = 0; c < 2; ++c) p.far[c] = FlowFarPiece{c * len, 0, (uint32_t)s.v[F_FAR_WP]};
    ms = ms < 0 ? 0 : ms > 500 ? 500 : ms;                                                   // :258-265 (the warning is the host's business)
    // Frame slot of frame i of call c: c * frames_per_call + i = c + i for the three shapes there are (1 x 1, 1 x 2,
    // 2 x 1); both loops have constant bounds so that the plan stays in registers on the device.
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 2; ++c) {
        if (c >= n_calls) break;
        // ---- WebRtcAecm_BufferFarend (:215-234) ----
        if (!(flags & kFlowNoFarend)) {
            p.far[c].pos = (uint32_t)s.v[F_FAR_WP];
            p.far[c].count = FlowFarendCall(s, mult, len);
        }
        // ---- WebRtcAecm_Process (:236-408) ----
        s.v[F_MS] = ms + 10;
        if (s.v[F_EC_STARTUP]) {
            FlowStartup(s, mult, frames_per_call);
            continue;
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int i = 0; i < 2; ++i) {
            if (i >= frames_per_call) break;
            FlowFrame &fr = p.frame[(c + i) & 1];
            fr.active = 1;
            fr.old_idx = i;
            const int32_t avail = (int32_t)((uint32_t)s.v[F_FAR_WP] - (uint32_t)s.v[F_FAR_RP]);
            if (FlowAsShort(FlowAsShort(avail) / kFlowFrame) > 0) {                              // :369-375
                fr.far_from_stream = 1;
                fr.far_pos = (uint32_t)s.v[F_FAR_RP];
                s.v[F_OLD_POS0 + i] = (int32_t)fr.far_pos;                                       // farendOld[i] = this frame (:373)
                s.v[F_OLD_ROW0 + i] = 0;
                refreshed[i] = 1;
                s.v[F_FAR_RP] = (int32_t)((uint32_t)s.v[F_FAR_RP] + (uint32_t)kFlowFrame);
            } else if (!s.v[F_OLD_ROW0 + i]) {                                                   // :376-379: replay, still in the far ring
                fr.far_from_stream = 1;
                fr.far_pos = (uint32_t)s.v[F_OLD_POS0 + i];
            }
            if ((i == 0 && fs == 8000) || (i == 1 && fs == 16000)) FlowEstBufDelay(s, mult);   // :384-387
            // WebRtcAecm_ProcessFrame (aecm_core.cc:501-572); the core's far delay line is a pass-through (knownDelay 0)
            fr.frm_pos = (uint32_t)s.v[F_FRM_POS];
            if (fr.far_from_stream) {                       // does this frame continue the run of the far stream the previous one ended?
                const int32_t delta = (int32_t)(fr.far_pos - fr.frm_pos);
                if (delta != s.v[F_RUN_DELTA]) {
                    s.v[F_RUN_DELTA] = delta;
                    s.v[F_RUN_POS] = (int32_t)fr.frm_pos;
                }
            } else {
                s.v[F_RUN_POS] = (int32_t)(fr.frm_pos + (uint32_t)kFlowFrame);   // a replay row is no part of the far stream
            }
            p.near_base = near_pos + (uint32_t)(kFlowFrame * (c + i)) - fr.frm_pos;
            s.v[F_FRM_POS] = (int32_t)(fr.frm_pos + (uint32_t)kFlowFrame);
            int nb = 0;
            while ((int32_t)((uint32_t)s.v[F_FRM_POS] - (uint32_t)s.v[F_BLK_POS]) >= kFlowBlock) {
                s.v[F_BLK_POS] = (int32_t)((uint32_t)s.v[F_BLK_POS] + (uint32_t)kFlowBlock);
                ++nb;
            }
            fr.n_blocks = nb;
            p.n_blocks += nb;
            const int32_t size = (int32_t)((uint32_t)s.v[F_BLK_POS] - (uint32_t)s.v[F_OUT_RP]);
            if (size < kFlowFrame) s.v[F_OUT_RP] = (int32_t)((uint32_t)s.v[F_BLK_POS] - (uint32_t)kFlowFrame);   // stuffing (:559-562)
            fr.out_pos = (uint32_t)s.v[F_OUT_RP];
            s.v[F_OUT_RP] = (int32_t)(fr.out_pos + (uint32_t)kFlowFrame);
        }
    }
    for (int i = 0; i < 2; ++i)
        if (p.spill[i] && !refreshed[i]) s.v[F_OLD_ROW0 + i] = 1;
    if (p.frame[0].active || p.frame[1].active) {          // every active frame completes at least one block
        if ((int32_t)((uint32_t)s.v[F_RUN_POS] - p.blk_pos0) <= 0) {
            p.direct = 1;
            p.far_delta = (uint32_t)s.v[F_RUN_DELTA];
            s.v[F_RUN_POS] = (int32_t)p.blk_pos0;           // any position of the run will do: keep it near (the counters wrap)
            s.v[F_FF_VALID] = 0;
        } else {
            if (!s.v[F_FF_VALID]) {                         // the ticks before were direct: their run covers what is pending
                p.left_count = (int32_t)(frm_start - p.blk_pos0);
                p.left_delta = run_delta_start;
            }
            s.v[F_FF_VALID] = 1;
        }
    }
}

// The plan as the kFlowPlanWords int32 the two kernels exchange.
constexpr int kFlowPlanWords = 16;
enum FlowPlanWord : int {
    P_FAR_COUNTS = 0,     // far[0].count | far[1].count << 16   (far[c].src = c * 80: a second piece only exists for two 80-sample calls)
    P_FAR_POS0, P_FAR_POS1,
    P_BITS,               // frame f at bits [4 f, 4 f + 4): active | far_from_stream << 1 | old_idx << 2 | spill[f] << 3;
                          // n_frames << 8, n_blocks << 12, direct << 16, left_count << 20
    P_FRAME_FAR_POS0, P_FRAME_FAR_POS1, P_FRM_POS0, P_FRM_POS1, P_OUT_POS0, P_OUT_POS1,
    P_BLK_POS0, P_NEAR_BASE, P_FAR_DELTA, P_LEFT_DELTA, P_SPILL_POS0, P_SPILL_POS1
};
AECM_FLOW_HD void FlowPackPlan(const FlowPlan &p, int32_t w[kFlowPlanWords]) {
    w[P_FAR_COUNTS] = p.far[0].count | (p.far[1].count << 16);
    w[P_FAR_POS0] = (int32_t)p.far[0].pos;
    w[P_FAR_POS1] = (int32_t)p.far[1].pos;
    int32_t bits = (p.n_frames << 8) | (p.n_blocks << 12) | (p.direct << 16) | (p.left_count << 20);
    for (int f = 0; f < 2; ++f)
        bits |= (p.frame[f].active | (p.frame[f].far_from_stream << 1) | (p.frame[f].old_idx << 2) | (p.spill[f] << 3)) << (4 * f);
    w[P_BITS] = bits;
    w[P_FRAME_FAR_POS0] = (int32_t)p.frame[0].far_pos;
    w[P_FRAME_FAR_POS1] = (int32_t)p.frame[1].far_pos;
    w[P_FRM_POS0] = (int32_t)p.frame[0].frm_pos;
    w[P_FRM_POS1] = (int32_t)p.frame[1].frm_pos;
    w[P_OUT_POS0] = (int32_t)p.frame[0].out_pos;
    w[P_OUT_POS1] = (int32_t)p.frame[1].out_pos;
    w[P_BLK_POS0] = (int32_t)p.blk_pos0;
    w[P_NEAR_BASE] = (int32_t)p.near_base;
    w[P_FAR_DELTA] = (int32_t)p.far_delta;
    w[P_LEFT_DELTA] = (int32_t)p.left_delta;
    w[P_SPILL_POS0] = (int32_t)p.spill_pos[0];
    w[P_SPILL_POS1] = (int32_t)p.spill_pos[1];
}
// far[1].src is the length of the first call: 80 whenever there is a second piece.  (Per-frame block counts do not travel.)
AECM_FLOW_HD void FlowUnpackPlan(const int32_t w[kFlowPlanWords], FlowPlan &p) {
    p.far[0] = FlowFarPiece{0, w[P_FAR_COUNTS] & 0xffff, (uint32_t)w[P_FAR_POS0]};
    p.far[1] = FlowFarPiece{kFlowFrame, (int32_t)((uint32_t)w[P_FAR_COUNTS] >> 16), (uint32_t)w[P_FAR_POS1]};
    const int32_t bits = w[P_BITS];
    p.n_calls = 0;
    p.n_frames = (bits >> 8) & 15;
    p.n_blocks = (bits >> 12) & 15;
    p.direct = (bits >> 16) & 1;
    p.left_count = (bits >> 20) & 127;
    for (int f = 0; f < 2; ++f) {
        const int32_t b = (bits >> (4 * f)) & 15;
        p.frame[f] = FlowFrame{b & 1, (b >> 1) & 1, (uint32_t)w[P_FRAME_FAR_POS0 + f], (b >> 2) & 1, (uint32_t)w[P_FRM_POS0 + f], 0,
                               (uint32_t)w[P_OUT_POS0 + f]};
        p.spill[f] = (b >> 3) & 1;
        p.spill_pos[f] = (uint32_t)w[P_SPILL_POS0 + f];
    }
    p.blk_pos0 = (uint32_t)w[P_BLK_POS0];
    p.near_base = (uint32_t)w[P_NEAR_BASE];
    p.far_delta = (uint32_t)w[P_FAR_DELTA];
    p.left_delta = (uint32_t)w[P_LEFT_DELTA];
}

}  // namespace aecm
#endif  // AECM_AMD_FLOW_PLAN_H_
