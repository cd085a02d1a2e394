// TEST INFRASTRUCTURE: the device's session machinery (webrtc_aecm_amd/csrc/aecm_flow_plan.h: position arithmetic +
// the sample movements aecm_tick_flow_kernel performs) against the generic restatement of the reference's wrapper
// (aecm_session_flow.h: SessionFlow<T>, itself pinned to the reference by tests/test_sim.py / test_gpu_parity.py),
// both run on sample TAGS instead of samples: every far / near input sample of a session's life gets a unique tag,
// block outputs are tagged by (block, sample), so equality of the tags that reach each block and each output sample
// is equality of the data flow for every possible audio content.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "aecm_flow_plan.h"
#include "aecm_ops.h"
#include "aecm_session_flow.h"

namespace {

using namespace aecm;

constexpr int64_t kRingLen = 8192, kOutTagBase = int64_t(1) << 40, kNone = -1;

struct Rng {
    uint64_t s;
    uint32_t next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(s >> 33);
    }
    int range(int lo, int hi) { return lo + (int)(next() % (uint32_t)(hi - lo + 1)); }
    bool chance(int percent) { return (int)(next() % 100u) < percent; }
};

// What aecm_tick_flow_kernel keeps per session, in the tag domain.
struct DeviceSide {
    FlowRegs regs;
    std::vector<int64_t> far_ring, near_ring, out_ring, far_frames, far_old;
    uint32_t near_pos = 0;
    int64_t blocks_done = 0, direct_ticks = 0, spill_ticks = 0;
    explicit DeviceSide() : far_ring(kRingLen, kNone), near_ring(kRingLen, kNone), out_ring(kRingLen, kNone), far_frames(kFlowFarFrameRing, kNone),
                            far_old(2 * kFlowFrame, kNone) {
        int32_t words[kFlowWords];
        FlowInit(words);
        for (int k = 0; k < kFlowFieldsUsed; ++k) regs.v[k] = words[k];
    }
    // `calls` WebRtcAecm_BufferFarend calls of len samples without a Process, exactly as aecm_buffer_farend_kernel does them:
    // only the fields a burst may read are visible, the spills come first, then call by call.
    void BufferFarend(int fs, int len, int calls, const int64_t *far_in) {
        const int64_t mask = kRingLen - 1;
        FlowRegs r;
        for (int k = 0; k < kFlowFieldsUsed; ++k) r.v[k] = 0x5a5a5a5a;                  // poison what the kernel does not load
        FlowBurstReads([&](int f) { r.v[f] = regs.v[f]; });
        FlowBurst b;
        FlowBurstBegin(r, len, calls, b);
        for (int i = 0; i < 2; ++i)
            if (b.spill[i])
                for (int j = 0; j < kFlowFrame; ++j) far_old[i * kFlowFrame + j] = far_ring[(b.spill_pos[i] + (uint32_t)j) & mask];
        spill_ticks += b.spill[0] + b.spill[1];
        const int mult = fs == 16000 ? 2 : 1;
        for (int c = 0; c < calls; ++c) {
            const uint32_t pos = (uint32_t)r.v[F_FAR_WP];
            const int32_t accepted = FlowFarendCall(r, mult, len);
            for (int j = 0; j < accepted; ++j) far_ring[(pos + (uint32_t)j) & mask] = far_in[c * len + j];
            burst_dropped += len - accepted;
        }
        FlowBurstWrites([&](int f) { regs.v[f] = r.v[f]; });
    }
    int64_t burst_dropped = 0;
    // One tick exactly as the kernel does it: plan, appends, far frames, blocks, output frames.
    void Tick(int fs, int n, int ms, int flags, const int64_t *far_in, const int64_t *near_in, int64_t *out, std::vector<int64_t> *blk_far,
              std::vector<int64_t> *blk_near) {
        const int64_t mask = kRingLen - 1;
        FlowPlan planned, p;
        FlowTick(regs, fs, n, ms, flags, near_pos, planned);
        int32_t words[kFlowPlanWords];                                  // the plan travels between the two kernels in this form
        FlowPackPlan(planned, words);
        FlowUnpackPlan(words, p);
        for (int j = 0; j < n; ++j) {
            for (int c = 0; c < 2; ++c)
                if (j >= p.far[c].src && j < p.far[c].src + p.far[c].count) far_ring[(p.far[c].pos + (uint32_t)(j - p.far[c].src)) & mask] = far_in[j];
            near_ring[(near_pos + (uint32_t)j) & mask] = near_in[j];
        }
        // replay frames about to leave the far ring move to their rows first
        for (int i = 0; i < 2; ++i)
            if (p.spill[i])
                for (int j = 0; j < kFlowFrame; ++j) far_old[i * kFlowFrame + j] = far_ring[(p.spill_pos[i] + (uint32_t)j) & mask];
        spill_ticks += p.spill[0] + p.spill[1];
        direct_ticks += p.direct;
        if (!p.direct) {
            // framing: the pending samples the direct ticks before left in the far ring, then the tick's frames; every
            // load before any store, as in the kernel
            int64_t left[kFlowBlock], frames[2][kFlowFrame];
            for (int j = 0; j < p.left_count; ++j) left[j] = far_ring[(p.blk_pos0 + p.left_delta + (uint32_t)j) & mask];
            for (int f = 0; f < p.n_frames; ++f) {
                const FlowFrame &q = p.frame[f];
                if (!q.active) continue;
                for (int j = 0; j < kFlowFrame; ++j)
                    frames[f][j] = q.far_from_stream ? far_ring[(q.far_pos + (uint32_t)j) & mask] : far_old[q.old_idx * kFlowFrame + j];
            }
            for (int j = 0; j < p.left_count; ++j) far_frames[(p.blk_pos0 + (uint32_t)j) & (kFlowFarFrameRing - 1)] = left[j];
            for (int f = 0; f < p.n_frames; ++f) {
                const FlowFrame &q = p.frame[f];
                if (!q.active) continue;
                for (int j = 0; j < kFlowFrame; ++j) far_frames[(q.frm_pos + (uint32_t)j) & (kFlowFarFrameRing - 1)] = frames[f][j];
            }
        }
        for (int b = 0; b < p.n_blocks; ++b, ++blocks_done)
            for (int t = 0; t < kFlowBlock; ++t) {
                const uint32_t x = p.blk_pos0 + (uint32_t)(b * kFlowBlock + t);
                blk_far->push_back(p.direct ? far_ring[(x + p.far_delta) & mask] : far_frames[x & (kFlowFarFrameRing - 1)]);
                blk_near->push_back(near_ring[(p.near_base + x) & mask]);
                out_ring[x & mask] = kOutTagBase + blocks_done * kFlowBlock + t;
            }
        for (int f = 0; f < p.n_frames; ++f)
            for (int j = 0; j < kFlowFrame; ++j)
                out[f * kFlowFrame + j] = p.frame[f].active ? out_ring[(p.frame[f].out_pos + (uint32_t)j) & mask] : near_in[f * kFlowFrame + j];
        near_pos += (uint32_t)n;
    }
};

}  // namespace

extern "C" {

// Drives one session through n_ticks ticks of a call pattern drawn from `scenario` on both sides.  Returns -1 when every
// block input and every output sample agreed, else the first tick that differed; detail[0] = what differed (1 block
// count, 2 far block tags, 3 near block tags, 4 output, 5 return code), detail[1] = blocks processed in total,
// detail[2] = ticks spent past the start-up phase, detail[3] = ticks in which the jitter buffer dropped far samples,
// detail[4] = ticks whose blocks fetched the far end from the far ring directly, detail[5] = replay frames moved to rows,
// detail[6] = far-end calls made outside ticks (bursts), detail[7] = samples of those the full jitter buffer dropped;
// detail[0] = 100 + field: FlowStateDefect refused a state the session passed through.
int64_t sim_flow_fuzz(uint64_t seed, int fs, int n_ticks, int scenario, uint32_t start_pos, int64_t *detail) {
    Rng rng{seed * 2654435761ull + 12345};
    DeviceSide dev;
    // counters that wrap: start them anywhere (the device state is position arithmetic modulo 2^32)
    dev.regs.v[F_FAR_RP] = dev.regs.v[F_FAR_WP] = (int32_t)start_pos;
    dev.regs.v[F_FRM_POS] = dev.regs.v[F_BLK_POS] = dev.regs.v[F_OUT_RP] = (int32_t)(start_pos * 3u);
    dev.near_pos = start_pos * 7u;
    SessionFlow<int64_t> ref(kNone);
    ref.Init(fs);
    int64_t far_offered = 0, near_offered = 0, ref_blocks = 0;
    int ms_walk = 40;
    for (int i = 0; i < 8; ++i) detail[i] = 0;
    for (int64_t tick = 0; tick < n_ticks; ++tick) {
        int n = fs == 16000 ? 160 : 80, ms = 40, flags = 0;
        switch (scenario) {
            case 0: break;                                                       // the reference CLI's cadence
            case 1: ms = rng.range(-20, 140); if (rng.chance(3)) ms = rng.chance(50) ? -300 : 700; break;
            case 2: if (rng.chance(25)) flags |= kFlowNoFarend; ms = rng.range(20, 60); break;
            case 3: n = 80; ms = rng.range(30, 50); break;                       // 16 kHz in 80-sample calls never leaves start-up: the buffer saturates
            case 4: n = rng.chance(50) ? 80 : 160; ms = rng.range(0, 500); if (rng.chance(30)) flags |= kFlowNoFarend; if (rng.chance(30)) flags |= kFlowSplitCalls; break;
            case 5: ms_walk += rng.range(-6, 6); ms_walk = ms_walk < 0 ? 0 : ms_walk > 500 ? 500 : ms_walk; ms = ms_walk;
                    if (rng.chance(8)) flags |= kFlowNoFarend; if (rng.chance(50)) flags |= kFlowSplitCalls; break;
            case 6: ms = (tick / 200) % 2 ? 480 : 10; if (rng.chance(2)) flags |= kFlowNoFarend; break;   // delay steps: stuffing and skipping
            case 7: n = 160; flags = rng.chance(70) ? kFlowSplitCalls : 0; ms = rng.range(0, 80); if (tick % 50 < 10) flags |= kFlowNoFarend; break;
            case 9: {                                                           // replay slot 1 goes unused for so long that its frame leaves
                const int ph = (int)(tick % 200);                               // the far ring (two 80-sample calls only use slot 0), then both
                n = 160; ms = rng.range(35, 45);                                // slots are replayed in a run of underruns
                if (ph >= 60 && ph < 150) flags |= kFlowSplitCalls;
                if (ph >= 138 && ph < 160) flags |= kFlowNoFarend;
                break;
            }
            case 11:                                                            // 16 kHz in 80-sample calls for more than 32 767 calls: the
                n = tick < 33100 ? 80 : 160;                                    // wrapper's short counters wrap while nBlocks10ms == 0, then
                ms = 40 + (int)(tick % 3);                                      // 160-sample calls compare them as size_t (:320,:330)
                break;
            case 12: case 13: case 14: break;                                   // far-end bursts, below
            default: n = rng.chance(20) ? 80 : 160; ms = rng.range(0, 200); flags = rng.range(0, 3); break;
        }
        if (n != 160) flags &= ~kFlowSplitCalls;
        // far-end bursts: extra WebRtcAecm_BufferFarend calls (of their own size) between two ticks, on both sides
        int extra = 0, extra_len = n;
        switch (scenario) {
            case 12: extra = (int)(tick % 6 == 4) + 2 * (int)(tick % 6 == 5); if (tick % 50 == 49) extra = 30;      // k = 0,1,1,1,2,3 far calls per tick
                     if (tick % 6 == 0) flags |= kFlowNoFarend; ms = rng.range(30, 60); break;                      // + a 30-frame burst every 50 ticks
            case 13: extra = rng.chance(30) ? rng.range(1, 4) : 0; extra_len = rng.chance(50) ? 80 : 160; if (rng.chance(35)) flags |= kFlowNoFarend;
                     if (rng.chance(2)) extra = rng.range(20, 60); ms = rng.range(0, 300); n = rng.chance(30) ? 80 : 160; if (rng.chance(30)) flags |= kFlowSplitCalls;
                     if (n != 160) flags &= ~kFlowSplitCalls; break;
            case 14: { const int ph = (int)(tick % 400);                       // long silences of the far end, then everything that was held back at once:
                     if (ph >= 100 && ph < 160) flags |= kFlowNoFarend;       // the jitter buffer overflows, replay frames are lapped in the far ring
                     if (ph == 160) extra = 60; if (ph == 300) { extra = 255; extra_len = 160; }
                     n = 160; if (ph >= 200 && ph < 290) flags |= kFlowSplitCalls; ms = rng.range(35, 45); break; }
            default: break;
        }
        if (extra > 0) {
            std::vector<int64_t> burst((size_t)extra * extra_len);
            for (size_t j = 0; j < burst.size(); ++j) burst[j] = far_offered + (int64_t)j;
            far_offered += (int64_t)burst.size();
            dev.BufferFarend(fs, extra_len, extra, burst.data());
            for (int c = 0; c < extra; ++c) ref.BufferFarend(burst.data() + (size_t)c * extra_len, (size_t)extra_len);
            detail[6] += extra;
        }
        int64_t far_in[160], near_in[160], out_dev[160], out_ref[160];
        for (int j = 0; j < n; ++j) { far_in[j] = far_offered + j; near_in[j] = (int64_t(1) << 32) + near_offered + j; }
        far_offered += n;
        near_offered += n;
        std::vector<int64_t> dfar, dnear, rfar, rnear;
        const int64_t dropped_before = (int64_t)(uint32_t)dev.regs.v[F_FAR_WP];
        dev.Tick(fs, n, ms, flags, far_in, near_in, out_dev, &dfar, &dnear);
        if (!(flags & kFlowNoFarend) && (uint32_t)dev.regs.v[F_FAR_WP] - (uint32_t)dropped_before != (uint32_t)n) detail[3]++;
        if (!dev.regs.v[F_EC_STARTUP]) detail[2]++;
        // the reference side: the same calls through SessionFlow
        const int n_calls = (flags & kFlowSplitCalls) ? 2 : 1, len = n / n_calls;
        int32_t rc_ref = 0;
        for (int c = 0; c < n_calls; ++c) {
            if (!(flags & kFlowNoFarend)) ref.BufferFarend(far_in + c * len, (size_t)len);
            const int32_t rc = ref.Process(near_in + c * len, nullptr, out_ref + c * len, (size_t)len, (int16_t)ms,
                                           [&](const int64_t *fb, const int64_t *nb, const int64_t *, int64_t *ob, int nblk) {
                                               rfar.insert(rfar.end(), fb, fb + nblk * kFlowBlock);
                                               rnear.insert(rnear.end(), nb, nb + nblk * kFlowBlock);
                                               for (int j = 0; j < nblk * kFlowBlock; ++j) ob[j] = kOutTagBase + ref_blocks * kFlowBlock + j;
                                               ref_blocks += nblk;
                                               return true;
                                           });
            if (rc != 0 && rc_ref == 0) rc_ref = rc;
        }
        const int32_t rc_dev = (ms < 0 || ms > 500) ? kWarnBadParameter : 0;
        detail[1] = ref_blocks;
        detail[4] = dev.direct_ticks;
        detail[5] = dev.spill_ticks;
        detail[7] = dev.burst_dropped;
        int what = 0;
        {   // every state a session passes through is one WebRtcAecmSessions_ImportSession accepts
            int32_t words[kFlowWords] = {0};
            for (int k = 0; k < kFlowFieldsUsed; ++k) words[k] = dev.regs.v[k];
            if (FlowStateDefect(words) != 0) {
                detail[0] = 100 + FlowStateDefect(words);
                return tick;
            }
        }
        if (dfar.size() != rfar.size()) what = 1;
        else if (dfar != rfar) what = 2;
        else if (dnear != rnear) what = 3;
        else if (memcmp(out_dev, out_ref, sizeof(int64_t) * n) != 0) what = 4;
        else if (rc_dev != rc_ref) what = 5;
        if (what) {
            detail[0] = what;
            return tick;
        }
    }
    return -1;
}

// |a - b| < max(0.2 * b, 8.0) in doubles (echo_control_mobile.cc:310-312) against the integer form FlowStartup uses, for
// every msInSndCardBuf pair the wrapper can hold.  Returns the number of disagreements.
int sim_flow_tolerance_check(void) {
    int bad = 0;
    for (int first = 0; first <= 520; ++first)
        for (int ms = 0; ms <= 520; ++ms) {
            const double tol = std::max(0.2 * ms, 8.0);
            const bool ref = abs(first - ms) < tol;
            const int d = first > ms ? first - ms : ms - first;
            const bool ours = d < 8 || 5 * d < ms;
            bad += ref != ours;
        }
    return bad;
}

// divu_by_magic (aecm_ops.h: the NLMS step's division by bin + 1 through a 33-bit reciprocal) against n / d for every
// divisor the kernel uses and dividends around every multiple boundary plus a pseudo-random sweep of [0, 2^31].
// Returns the number of disagreements.
int sim_div_magic_check(void) {
    int bad = 0;
    uint64_t x = 88172645463325252ull;
    for (int d = 1; d <= 65; ++d) {
        int magic, shift;
        div_magic(d, &magic, &shift);
        auto check = [&](uint32_t n) { bad += (uint32_t)divu_by_magic<int>((int)n, magic, shift) != n / (uint32_t)d; };
        for (uint32_t k = 0; k < 64; ++k)
            for (int e = -2; e <= 2; ++e) {
                const int64_t n = (int64_t)k * d * 33554432 / 64 * 64 + e;            // around multiples of d across the range
                if (n >= 0 && n <= 2147483648ll) check((uint32_t)n);
            }
        for (uint32_t n : {0u, 1u, 2147483646u, 2147483647u, 2147483648u}) check(n);
        for (int i = 0; i < 200000; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            check((uint32_t)(x % 2147483649ull));
        }
    }
    return bad;
}

}  // extern "C"
