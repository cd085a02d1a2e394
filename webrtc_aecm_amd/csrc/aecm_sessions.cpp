#include "aecm_sessions.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/echo_control_mobile.h"

namespace aecm {

static_assert(kFlowFarRing == 8192, "the far ring of SessionBatch (kRing) is the one aecm_flow_plan.h ages replay frames against");

#define AECM_HIP_OK(expr) ((expr) == hipSuccess)

SessionBatch *SessionBatch::Create(int num_streams, int device_id) {
    BatchEngine *e = BatchEngine::Create(num_streams, device_id);
    if (!e) return nullptr;
    SessionBatch *b = new SessionBatch();
    b->engine_.reset(e);
    b->device_ = device_id;
    const size_t S = (size_t)num_streams;
    bool ok = AECM_HIP_OK(hipMalloc((void **)&b->far_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->near_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->out_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->blk_, 4 * S * kTickMaxBlockSamples * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->io_dev_, 4 * S * 160 * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->class_of_dev_, S * sizeof(int32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&b->blocks_per_stream_dev_, S * sizeof(int32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&b->table_dev_, kMaxFlowClasses * sizeof(TickClassEntry))) &&
              AECM_HIP_OK(hipHostMalloc((void **)&b->table_host_, kMaxFlowClasses * sizeof(TickClassEntry), hipHostMallocDefault)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->lean_dev_, kMaxFlowClasses * sizeof(TickLeanEntry))) &&
              AECM_HIP_OK(hipHostMalloc((void **)&b->lean_host_, kMaxFlowClasses * sizeof(TickLeanEntry), hipHostMallocDefault));
    b->flow_mode_ = ChooseTickMode(num_streams) == kTickFlow;
    if (ok && b->flow_mode_)
        ok = AECM_HIP_OK(hipMalloc((void **)&b->flow_state_, S * kFlowFieldsUsed * sizeof(int32_t))) &&
             AECM_HIP_OK(hipMalloc((void **)&b->flow_plans_, S * kFlowPlanWords * sizeof(int32_t))) &&
             AECM_HIP_OK(hipMalloc((void **)&b->far_frames_, S * kFlowFarFrameRing * 2)) &&
             AECM_HIP_OK(hipMalloc((void **)&b->far_old_, S * 2 * kFlowFrame * 2)) &&
             // per-session msInSndCardBuf / flags of a tick: pinned host arrays the planning kernel reads in place
             AECM_HIP_OK(hipHostMalloc((void **)&b->ms_host_, S * sizeof(int16_t), hipHostMallocMapped)) &&
             AECM_HIP_OK(hipHostMalloc((void **)&b->flags_host_, S, hipHostMallocMapped)) &&
             AECM_HIP_OK(hipHostGetDevicePointer((void **)&b->ms_dev_, b->ms_host_, 0)) &&
             AECM_HIP_OK(hipHostGetDevicePointer((void **)&b->flags_dev_, b->flags_host_, 0));
    if (!ok) {
        delete b;
        return nullptr;
    }
    return b;
}

SessionBatch::~SessionBatch() {
    (void)hipSetDevice(device_);
    if (engine_) (void)engine_->Synchronize();
    (void)hipFree(far_ring_);
    (void)hipFree(near_ring_);
    (void)hipFree(out_ring_);
    (void)hipFree(clean_ring_);
    (void)hipFree(blk_);
    (void)hipFree(io_dev_);
    (void)hipFree(class_of_dev_);
    (void)hipFree(blocks_per_stream_dev_);
    (void)hipFree(table_dev_);
    if (table_host_) (void)hipHostFree(table_host_);
    (void)hipFree(lean_dev_);
    if (lean_host_) (void)hipHostFree(lean_host_);
    (void)hipFree(flow_state_);
    (void)hipFree(flow_plans_);
    (void)hipFree(far_frames_);
    (void)hipFree(far_old_);
    if (ms_host_) (void)hipHostFree(ms_host_);
    if (flags_host_) (void)hipHostFree(flags_host_);
}

// WebRtcAecm_Init of the wrapper state of sessions [first, first + count) on the device: all zero but the three start-up
// flags (FlowInit), empty frame stream and replay rows.  The rings are cleared by the callers.
bool SessionBatch::ResetFlowRows(int first, int count) {
    hipStream_t st = engine_->stream();
    const size_t S = (size_t)engine_->num_streams();
    bool ok = true;
    for (int f = 0; f < kFlowFieldsUsed; ++f) {                       // field-major: one short run per field
        const int value = FlowFieldStartsAtOne(f) ? 1 : 0;
        ok = ok && AECM_HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(flow_state_ + (size_t)f * S + first), value, (size_t)count, st));
    }
    return ok && AECM_HIP_OK(hipMemsetAsync(far_frames_ + (size_t)first * kFlowFarFrameRing, 0, (size_t)count * kFlowFarFrameRing * 2, st)) &&
           AECM_HIP_OK(hipMemsetAsync(far_old_ + (size_t)first * 2 * kFlowFrame, 0, (size_t)count * 2 * kFlowFrame * 2, st));
}

int32_t SessionBatch::Init(int32_t samp_freq) {
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    const size_t bytes = (size_t)S * kRing * 2;
    if (!AECM_HIP_OK(hipMemsetAsync(far_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(near_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(out_ring_, 0, bytes, engine_->stream())) ||
        (clean_ring_ && !AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, engine_->stream()))))
        return AECM_UNSPECIFIED_ERROR;
    if (flow_mode_ && !ResetFlowRows(0, S)) return AECM_UNSPECIFIED_ERROR;
    near_pos_ = 0;
    tick_count_ = 0;
    fs_ = samp_freq;
    poisoned_ = false;
    classes_.clear();
    classes_.emplace_back();                      // every session starts in one class
    classes_[0].members = S;
    class_of_.assign((size_t)S, 0);
    last_key_.clear();
    class_of_dirty_ = true;
    return classes_[0].flow.Init(samp_freq);
}

int32_t SessionBatch::CheckSession(int session) const {
    if (classes_.empty() || !classes_[0].flow.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    if (session < 0 || session >= engine_->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

void SessionBatch::DropEmptyClasses() {
    bool any_empty = false;
    for (const FlowClass &c : classes_) any_empty = any_empty || c.members <= 0;
    if (!any_empty) return;
    std::vector<int32_t> remap(classes_.size(), -1);
    std::vector<FlowClass> kept;
    for (size_t k = 0; k < classes_.size(); ++k)
        if (classes_[k].members > 0) {
            remap[k] = (int32_t)kept.size();
            kept.push_back(std::move(classes_[k]));
        }
    classes_.swap(kept);
    for (int32_t &c : class_of_) c = remap[c];
    class_of_dirty_ = true;
}

// WebRtcAecm_Init of ONE session (same sampling rate as the batch): fresh core state on the device, fresh
// session flow on the host.  Sessions re-initialised between the same two ticks share one flow class.  The
// session's device rings need no clearing: a fresh flow only ever refers to samples appended after its birth
// (its far / output tags restart at 0), everything older reads as "never written" = 0.
int32_t SessionBatch::InitSession(int session) {
    if (int32_t rc = CheckSession(session)) return rc;
    if (flow_mode_) {
        // fresh core state, fresh wrapper state, and rings that read as never written (a fresh jitter buffer's read
        // pointer can be moved back over never-written memory, the output ring is stuffed from it: ring_buffer.c:75-82)
        hipStream_t st = engine_->stream();
        const size_t row = (size_t)kRing * 2, off = (size_t)session * kRing;
        const bool ok = AECM_HIP_OK(hipSetDevice(device_)) && engine_->InitStreams(session, 1) && ResetFlowRows(session, 1) &&
                        AECM_HIP_OK(hipMemsetAsync(far_ring_ + off, 0, row, st)) && AECM_HIP_OK(hipMemsetAsync(out_ring_ + off, 0, row, st)) &&
                        AECM_HIP_OK(hipStreamSynchronize(st));
        if (!ok) { poisoned_ = true; return AECM_UNSPECIFIED_ERROR; }
        return 0;
    }
    int32_t id = -1;
    for (size_t k = 0; k < classes_.size(); ++k)
        if (classes_[k].born == tick_count_ && classes_[k].far_count == 0 && classes_[k].blocks_done == 0) { id = (int32_t)k; break; }
    if (id < 0) {
        if ((int)classes_.size() >= kMaxFlowClasses) return AECM_UNSUPPORTED_FUNCTION_ERROR;
        classes_.emplace_back();
        id = (int32_t)classes_.size() - 1;
        classes_[id].born = tick_count_;
        if (int32_t rc = classes_[id].flow.Init(fs_)) return rc;
    }
    if (!engine_->InitStreams(session, 1)) { poisoned_ = true; return AECM_UNSPECIFIED_ERROR; }
    const int32_t old = class_of_[(size_t)session];
    if (old != id) {
        classes_[old].members--;
        classes_[id].members++;
        class_of_[(size_t)session] = id;
        class_of_dirty_ = true;
        last_key_.clear();
        DropEmptyClasses();
    }
    return 0;
}

int32_t SessionBatch::SetConfigSession(int session, int16_t cng_mode, int16_t echo_mode) {
    if (int32_t rc = CheckSession(session)) return rc;
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (echo_mode < 0 || echo_mode > 4) {                    // the reference commits cngMode before it rejects echoMode (:421-428)
        if (!engine_->SetCngMode(cng_mode, session, 1)) return AECM_UNSPECIFIED_ERROR;
        return AECM_BAD_PARAMETER_ERROR;
    }
    return engine_->SetConfig(cng_mode, echo_mode, session, 1) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t SessionBatch::InitEchoPathSession(int session, const void *path, size_t size_bytes) {
    if (path == nullptr) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (int32_t rc = CheckSession(session)) return rc;
    int16_t tmp[kBins];
    memcpy(tmp, path, sizeof tmp);
    return engine_->SetEchoPath(session, tmp) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t SessionBatch::GetEchoPathSession(int session, void *path, size_t size_bytes) {
    if (path == nullptr) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (int32_t rc = CheckSession(session)) return rc;
    int16_t tmp[kBins];
    if (!engine_->GetEchoPath(session, tmp)) return AECM_UNSPECIFIED_ERROR;
    memcpy(path, tmp, sizeof tmp);
    return 0;
}

int32_t SessionBatch::SetConfig(int16_t cng_mode, int16_t echo_mode) {
    if (classes_.empty() || !classes_[0].flow.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (echo_mode < 0 || echo_mode > 4) {
        if (!engine_->SetCngMode(cng_mode, 0, -1)) return AECM_UNSPECIFIED_ERROR;
        return AECM_BAD_PARAMETER_ERROR;
    }
    return engine_->SetConfig(cng_mode, echo_mode, 0, -1) ? 0 : AECM_UNSPECIFIED_ERROR;
}

// Which form a tick takes.  Default: the lean one-launch form (run-encoded sources, aecm_kernels.h) whenever the
// tick's sample movements fit its description; AECM_TICK_MODE=lean|fused|three (or the older AECM_TICK_FUSED=1|0)
// forces the coded one-launch or the three-launch form (A/B measurements, tests of those paths).
SessionBatch::TickMode SessionBatch::ChooseTickMode(int num_streams) {
    static const int forced = [] {
        if (const char *m = getenv("AECM_TICK_MODE")) {
            if (!strcmp(m, "flow")) return (int)kTickFlow;
            if (!strcmp(m, "lean")) return (int)kTickLean;
            if (!strcmp(m, "fused")) return (int)kTickFused;
            if (!strcmp(m, "three")) return (int)kTickThreeLaunch;
        }
        if (const char *e = getenv("AECM_TICK_FUSED")) return (int)(e[0] != '0' ? kTickFused : kTickThreeLaunch);
        return -1;
    }();
    (void)num_streams;
    return forced >= 0 ? (TickMode)forced : kTickFlow;
}

// Describe `count` sample tags as runs of consecutive ring positions (aecm_kernels.h: TickRuns).  tag >= 0: a sample
// of `kind`'s ring; tag == -1: zero; tag <= -2 (outputs only): near-ring sample -(tag + 2).  False if more than
// kTickMaxRuns runs are needed.
static bool BuildRuns(const int64_t *tags, int count, int64_t ring_len, int32_t kind_pos, TickRuns *t) {
    memset(t, 0, sizeof *t);
    int n = 0;
    for (int i = 0; i < count;) {
        const int64_t v = tags[i];
        const bool zero = v == -1, near = v <= -2;
        const int64_t base = near ? -v - 2 : v;
        int j = i + 1;
        while (j < count) {
            const int64_t w = tags[j];
            const bool same = zero ? w == -1 : near ? (w <= -2 && -w - 2 == base + (j - i)) : (w >= 0 && w == base + (j - i));
            if (!same) break;
            ++j;
        }
        if (n == kTickMaxRuns) return false;
        t->end[n] = j;
        t->off[n] = zero ? kTickRunZero : (int32_t)((base - i) & (ring_len - 1));
        t->kind[n] = near ? (int32_t)kTickNearRing : kind_pos;
        ++n;
        i = j;
    }
    t->n = n > 0 ? n : 1;
    if (n == 0) { t->end[0] = count; t->off[0] = kTickRunZero; }
    return true;
}

// Give every session the class that matches (its previous class, its msInSndCardBuf and far-end flag of this tick).
int32_t SessionBatch::Regroup(const int16_t *ms_per_session, int16_t ms_uniform, const uint8_t *flags_per_session) {
    const size_t S = class_of_.size();
    std::vector<int32_t> key(S);
    for (size_t s = 0; s < S; ++s) {
        const int32_t ms = (uint16_t)(ms_per_session ? ms_per_session[s] : ms_uniform);
        key[s] = ms | (flags_per_session ? (int32_t)(flags_per_session[s] & (kNoFarend | kSplitCalls)) << 16 : 0);
    }
    if (last_key_ == key) {                                                     // same grouping as last tick
        return 0;
    }
    struct Child { int32_t key; int32_t id; };
    std::vector<std::vector<Child>> children(classes_.size());
    std::vector<FlowClass> next;
    std::vector<int32_t> next_class_of(S);
    for (size_t s = 0; s < S; ++s) {
        const int32_t old = class_of_[s];
        int32_t id = -1;
        for (const Child &c : children[old])
            if (c.key == key[s]) { id = c.id; break; }
        if (id < 0) {
            if ((int)next.size() >= kMaxFlowClasses) return AECM_UNSUPPORTED_FUNCTION_ERROR;
            id = (int32_t)next.size();
            next.push_back(classes_[old]);           // the flow state before this tick
            next.back().ms = (int16_t)(uint16_t)(key[s] & 0xffff);
            next.back().no_far = ((key[s] >> 16) & kNoFarend) != 0;
            next.back().split_calls = ((key[s] >> 16) & kSplitCalls) != 0;
            next.back().members = 0;
            children[old].push_back({key[s], id});
        }
        next[id].members++;
        next_class_of[s] = id;
    }
    classes_.swap(next);
    class_of_.swap(next_class_of);
    last_key_.swap(key);
    class_of_dirty_ = true;
    return 0;
}

// One tick of one class's session machinery in the index domain: where every block sample and every
// output sample comes from, as source codes (aecm_kernels.h).  A near tag is the absolute sample count;
// a far tag counts the samples the class's jitter buffer has ACCEPTED (a saturated buffer drops what does
// not fit, and may then re-read arbitrarily old content for ever: in accepted-sample time that content
// is never more than the buffer's 4000 samples away, so it always sits inside the device ring).
int32_t SessionBatch::AdvanceClass(FlowClass &c, int n, bool has_clean, TickClassEntry *entry, TickLeanEntry *lean, bool *lean_ok,
                                   bool *coded_ok, bool *stale) {
    memset(lean, 0, sizeof *lean);
    lean->far_pos = c.far_count;
    lean->out_pos = c.blocks_done * kBlock;
    lean->n_frames = n / kTickFrame;
    for (int f = 0; f < 2; ++f) { lean->out[f].n = 1; lean->out[f].end[0] = kTickFrame; lean->out[f].off[0] = kTickRunZero; }
    entry->n_block_samples = 0;
    entry->n_far = 0;
    entry->far_pos = c.far_count;
    entry->out_pos = c.blocks_done * kBlock;
    for (int i = 0; i < n; ++i) entry->assemble.out[i] = -1;
    // The tick's calls: one BufferFarend + Process of n samples, or (split_calls, n = 160) two of 80 samples.
    const int n_calls = c.split_calls ? 2 : 1, per_call = n / n_calls;
    const int64_t far_first = c.far_count, out_first = c.blocks_done * kBlock;
    int64_t out_tags[kTickMaxSamples], blk_far[kTickMaxBlockSamples], blk_near[kTickMaxBlockSamples];
    int n_blocks = 0, accepted[2] = {0, 0};
    int32_t rc_all = 0;
    for (int k = 0; k < n_calls; ++k) {
        int64_t far_tags[kTickMaxSamples], near_tags[kTickMaxSamples];
        for (int i = 0; i < per_call; ++i) { far_tags[i] = c.far_count + i; near_tags[i] = near_pos_ + k * per_call + i; }
        if (!c.no_far) {                                   // far-end underrun: no BufferFarend call in this tick
            const int32_t rc = c.flow.BufferFarend(far_tags, (size_t)per_call);
            if (rc != 0) return rc;
            accepted[k] = (int)c.flow.last_far_accepted();
            c.far_count += accepted[k];
        }
        bool passthrough = false;
        const int64_t out_base = out_first + (int64_t)n_blocks * kBlock;
        int got = 0;
        int64_t *out_k = out_tags + k * per_call;
        // the clean near-end is positioned exactly like the noisy one: it shares the near tags
        const int32_t rc = c.flow.Process(near_tags, has_clean ? near_tags : nullptr, out_k, (size_t)per_call, c.ms,
                                          [&](const int64_t *fb, const int64_t *nb, const int64_t *, int64_t *ob, int nblk) {
                                              memcpy(blk_far + n_blocks * kBlock, fb, sizeof(int64_t) * nblk * kBlock);
                                              memcpy(blk_near + n_blocks * kBlock, nb, sizeof(int64_t) * nblk * kBlock);
                                              for (int j = 0; j < nblk * kBlock; ++j) ob[j] = out_base + j;
                                              got = nblk;
                                              return true;
                                          },
                                          &passthrough);
        if (rc != 0 && rc != AECM_BAD_PARAMETER_WARNING) return rc;   // nothing processed; the rings still take the samples
        if (rc != 0 && rc_all == 0) rc_all = rc;
        if (passthrough)
            for (int i = 0; i < per_call; ++i) out_k[i] = -(out_k[i] + 2);
        n_blocks += got;
    }
    // what the far ring takes: the accepted samples of each call (a saturated jitter buffer drops the rest)
    lean->n_far = accepted[0];
    if (n_calls == 2) {
        if (accepted[0] == per_call) lean->n_far += accepted[1];                 // contiguous
        else { lean->far2_src = per_call; lean->far2_cnt = accepted[1]; *coded_ok = false; }   // the coded forms index the input row by tag
    }
    entry->n_far = lean->n_far;
    const int nbs = n_blocks * kBlock;
    const int64_t far_end = c.far_count, near_end = near_pos_ + n, out_end = out_first + nbs;
    auto code = [&](int64_t tag, int64_t first_of_tick, int64_t end, int kind_now, int kind_ring) -> int32_t {
        if (tag < 0) return -1;
        if (end - tag > kRing) *stale = true;           // every tag must still be inside its ring
        if (tag >= first_of_tick) return (int32_t)((kind_now << 28) | (int32_t)(tag - first_of_tick));
        return (int32_t)((kind_ring << 28) | (int32_t)(tag & (kRing - 1)));
    };
    for (int k = 0; k < nbs; ++k) {
        entry->gather.far[k] = code(blk_far[k], far_first, far_end, kTickFromInput, kTickFromRing);
        entry->gather.near[k] = code(blk_near[k], near_pos_, near_end, kTickFromInput, kTickFromRing);
    }
    for (int i = 0; i < n; ++i) {
        const int64_t v = out_tags[i];
        entry->assemble.out[i] = v >= 0    ? code(v, out_first, out_end, kTickFromInput, kTickFromRing)
                                 : v <= -2 ? code(-v - 2, near_pos_, near_end, kTickNearInput, kTickNearRing)
                                           : -1;
    }
    entry->n_block_samples = nbs;
    // the same tick as run descriptions (lean one-launch form)
    lean->n_blocks = n_blocks;
    for (int b = 0; b < n_blocks; ++b)
        if (!BuildRuns(blk_far + b * kBlock, kBlock, kRing, kTickFromRing, &lean->far[b]) ||
            !BuildRuns(blk_near + b * kBlock, kBlock, kRing, kTickFromRing, &lean->near[b]))
            *lean_ok = false;
    for (int f = 0; f < n / kTickFrame; ++f)
        if (!BuildRuns(out_tags + f * kTickFrame, kTickFrame, kRing, kTickFromRing, &lean->out[f])) *lean_ok = false;
    c.blocks_done += n_blocks;
    return rc_all;
}

int32_t SessionBatch::Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride, size_t n_samples,
                           int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes,
                           bool host_pointers) {
    if (far == nullptr || near == nullptr || out == nullptr) return AECM_NULL_POINTER_ERROR;
    if (classes_.empty() || !classes_[0].flow.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (n_samples != 80 && n_samples != 160) return AECM_BAD_PARAMETER_ERROR;       // compared as size_t: 2^32 + 80 is not 80
    if (stride < (int64_t)n_samples) return AECM_BAD_PARAMETER_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    const int n = (int)n_samples;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    if (clean && !clean_ring_) {
        const size_t bytes = (size_t)S * kRing * 2;
        if (!AECM_HIP_OK(hipMalloc((void **)&clean_ring_, bytes)) || !AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, st)))
            return AECM_UNSPECIFIED_ERROR;
    }
    if (flags_per_session && n != 160) {
        uint8_t any = 0;
        for (int s = 0; s < S; ++s) any |= flags_per_session[s];
        if (any & kSplitCalls) return AECM_BAD_PARAMETER_ERROR;                           // two 80-sample calls need 160 samples
    }
    if (flow_mode_) return TickFlow(far, near, clean, out, stride, n, ms, ms_per_session, flags_per_session, codes, host_pointers);
    // 1. which class every session is in for this tick
    if (ms_per_session || flags_per_session) {
        if (int32_t rc = Regroup(ms_per_session, ms, flags_per_session)) return rc;
    } else {
        for (FlowClass &c : classes_) { c.ms = ms; c.no_far = false; c.split_calls = false; }
        last_key_.clear();
    }
    const int n_classes = (int)classes_.size();
    // 2. the session machinery of every class in the index domain (the table is read by the previous tick's
    //    kernels until they finish: every tick ends with a stream synchronisation)
    bool stale = false, lean_ok = true, coded_ok = true;
    std::vector<int32_t> class_rc((size_t)n_classes, 0);
    int32_t first_rc = 0, max_nbs = 0;
    for (int k = 0; k < n_classes; ++k) {
        class_rc[k] = AdvanceClass(classes_[k], n, clean != nullptr, &table_host_[k], &lean_host_[k], &lean_ok, &coded_ok, &stale);
        if (class_rc[k] != 0 && first_rc == 0) first_rc = class_rc[k];
        max_nbs = std::max(max_nbs, table_host_[k].n_block_samples);
    }
    // From here on the host-side flows have advanced: a failure leaves them out of step with the device rings,
    // so it poisons the object (every later call is refused until Init) instead of silently corrupting audio.
    auto fail = [&]() -> int32_t {
        poisoned_ = true;
        (void)hipStreamSynchronize(st);              // async copies from caller / pinned memory may still be in flight
        return AECM_UNSPECIFIED_ERROR;
    };
    if (stale) return fail();
    if (codes)
        for (int s = 0; s < S; ++s) codes[s] = class_rc[class_of_[s]];
    // 3. device side of the tick: prepare -> blocks -> finish
    const int16_t *dfar = far, *dnear = near, *dclean = clean;
    int16_t *dout = out;
    int64_t dstride = stride;
    if (host_pointers) {
        dstride = 160;
        int16_t *f = io_dev_, *d = io_dev_ + (size_t)S * 160, *c = io_dev_ + 3 * (size_t)S * 160;
        if (!AECM_HIP_OK(hipMemcpy2DAsync(f, 320, far, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(hipMemcpy2DAsync(d, 320, near, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            (clean && !AECM_HIP_OK(hipMemcpy2DAsync(c, 320, clean, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st))))
            return fail();
        dfar = f;
        dnear = d;
        dout = io_dev_ + 2 * (size_t)S * 160;
        if (clean) dclean = c;
    }
    // pass-through samples come from the clean near-end when there is one (echo_control_mobile.cc:285-291)
    const int16_t *pass_ring = clean ? clean_ring_ : near_ring_, *pass_in = clean ? dclean : dnear;
    bool ok = true;
    TickMode mode = ChooseTickMode(S);
    if (mode == kTickLean && (!lean_ok || engine_->variant() != kVariantFast)) mode = S < 32768 ? kTickFused : kTickThreeLaunch;
    if (mode != kTickLean && !coded_ok) {
        if (!lean_ok || engine_->variant() != kVariantFast) return fail();     // no form can express this tick (never seen in practice)
        mode = kTickLean;
    }
    if (n_classes > 1) {
        if (class_of_dirty_) {
            ok = AECM_HIP_OK(hipMemcpyAsync(class_of_dev_, class_of_.data(), (size_t)S * sizeof(int32_t), hipMemcpyHostToDevice, st));
            class_of_dirty_ = !ok;
        }
        if (mode == kTickLean)
            ok = ok && AECM_HIP_OK(hipMemcpyAsync(lean_dev_, lean_host_, (size_t)n_classes * sizeof(TickLeanEntry), hipMemcpyHostToDevice, st));
        else
            ok = ok && AECM_HIP_OK(hipMemcpyAsync(table_dev_, table_host_, (size_t)n_classes * sizeof(TickClassEntry),
                                                  hipMemcpyHostToDevice, st));
    }
    if (ok && mode == kTickLean) {
        // one launch per tick, sources as runs of ring positions: the wave appends to its rings, reads its blocks'
        // inputs back from them, writes the block outputs to the output ring and assembles the tick's output
        TickIo tio{dfar, dnear, dclean, dout, dstride, n, far_ring_, near_ring_, clean_ring_, out_ring_, kRing, near_pos_};
        ok = n_classes == 1 ? AECM_HIP_OK(LaunchTickLean(engine_->state_ptrs(), tio, S, nullptr, nullptr, &lean_host_[0], st))
                            : AECM_HIP_OK(LaunchTickLean(engine_->state_ptrs(), tio, S, class_of_dev_, lean_dev_, nullptr, st));
    } else
    if (ok && mode == kTickFused) {
        // one launch per tick: every session's wave appends, runs its blocks through the source codes and
        // assembles its output (wins while the tick is launch- and latency-bound)
        TickIo tio{dfar, dnear, dclean, dout, dstride, n, far_ring_, near_ring_, clean_ring_, out_ring_, kRing, near_pos_};
        ok = n_classes == 1
                 ? AECM_HIP_OK(LaunchTick(engine_->state_ptrs(), tio, S, engine_->variant(), nullptr, nullptr, &table_host_[0], st))
                 : AECM_HIP_OK(LaunchTick(engine_->state_ptrs(), tio, S, engine_->variant(), class_of_dev_, table_dev_, nullptr, st));
    } else if (ok) {
        // three launches: prepare (append + gather into dense block rows) -> blocks -> finish (ring + assemble);
        // the block kernel then runs with its leanest I/O, which wins once the GPU is full
        int16_t *bfar = blk_, *bnear = blk_ + (size_t)S * kTickMaxBlockSamples, *bout = blk_ + 2 * (size_t)S * kTickMaxBlockSamples;
        int16_t *bclean = blk_ + 3 * (size_t)S * kTickMaxBlockSamples;
        if (n_classes == 1) {
            const TickClassEntry &e = table_host_[0];
            const int nbs = e.n_block_samples;
            ok = AECM_HIP_OK(LaunchTickPrepare(dfar, dnear, dclean, dstride, n, e.n_far, far_ring_, near_ring_, clean_ring_, kRing,
                                               e.far_pos, near_pos_, bfar, bnear, bclean, nbs, e.gather, S, st));
            if (ok && nbs > 0) {
                IoView io{bfar, bnear, clean ? bclean : nullptr, bout, nbs, kBlock};
                ok = engine_->ProcessBlocks(io, nbs / kBlock);
            }
            ok = ok && AECM_HIP_OK(LaunchTickFinish(bout, nbs, out_ring_, pass_ring, kRing, e.out_pos, pass_in, dstride, dout, n,
                                                    e.assemble, S, st));
        } else {
            ok = AECM_HIP_OK(LaunchTickPrepareClasses(dfar, dnear, dclean, dstride, n, far_ring_, near_ring_, clean_ring_, kRing,
                                                      near_pos_, bfar, bnear, bclean, class_of_dev_, table_dev_,
                                                      blocks_per_stream_dev_, S, st));
            if (ok && max_nbs > 0) {
                IoView io{bfar, bnear, clean ? bclean : nullptr, bout, kTickMaxBlockSamples, kBlock};
                ok = engine_->ProcessBlocks(io, max_nbs / kBlock, blocks_per_stream_dev_);
            }
            ok = ok && AECM_HIP_OK(LaunchTickFinishClasses(bout, out_ring_, pass_ring, kRing, pass_in, dstride, dout, n,
                                                           class_of_dev_, table_dev_, S, st));
        }
    }
    near_pos_ += n;
    tick_count_ += 1;
    if (!ok) return fail();
    if (host_pointers &&
        !AECM_HIP_OK(hipMemcpy2DAsync(out, stride * 2, dout, 320, (size_t)n * 2, S, hipMemcpyDeviceToHost, st)))
        return fail();
    if (!AECM_HIP_OK(hipStreamSynchronize(st))) return fail();
    return first_rc;
}

// The tick with the session machinery on the device: nothing per session happens on the host beyond handing over the
// tick's msInSndCardBuf / flags.  The return codes need no device either: the only thing a call of an initialised
// session with valid arguments can return is the warning for an out-of-range msInSndCardBuf (:258-265).
int32_t SessionBatch::TickFlow(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride, int n, int16_t ms,
                               const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes, bool host_pointers) {
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    auto fail = [&]() -> int32_t {
        poisoned_ = true;
        (void)hipStreamSynchronize(st);
        return AECM_UNSPECIFIED_ERROR;
    };
    auto code_of = [](int16_t v) -> int32_t { return (v < 0 || v > 500) ? AECM_BAD_PARAMETER_WARNING : 0; };
    int32_t first_rc = 0;
    if (ms_per_session) {
        int lo = ms_per_session[0], hi = lo;                                // one vectorisable pass; the per-session codes
        for (int s = 1; s < S; ++s) {                                       // only need a second one when somebody is out of range
            lo = std::min<int>(lo, ms_per_session[s]);
            hi = std::max<int>(hi, ms_per_session[s]);
        }
        if (lo < 0 || hi > 500) {
            for (int s = 0; s < S; ++s) {
                const int32_t rc = code_of(ms_per_session[s]);
                if (codes) codes[s] = rc;
                if (rc != 0 && first_rc == 0) first_rc = rc;
            }
        } else if (codes) {
            memset(codes, 0, (size_t)S * sizeof(int32_t));
        }
        memcpy(ms_host_, ms_per_session, (size_t)S * sizeof(int16_t));      // pinned + mapped; the previous tick ended with a synchronisation
    } else {
        first_rc = code_of(ms);
        if (codes)
            for (int s = 0; s < S; ++s) codes[s] = first_rc;
    }
    if (flags_per_session) {
        memcpy(flags_host_, flags_per_session, (size_t)S);
    }
    const int16_t *dfar = far, *dnear = near, *dclean = clean;
    int16_t *dout = out;
    int64_t dstride = stride;
    if (host_pointers) {
        dstride = 160;
        int16_t *f = io_dev_, *d = io_dev_ + (size_t)S * 160, *c = io_dev_ + 3 * (size_t)S * 160;
        if (!AECM_HIP_OK(hipMemcpy2DAsync(f, 320, far, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(hipMemcpy2DAsync(d, 320, near, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            (clean && !AECM_HIP_OK(hipMemcpy2DAsync(c, 320, clean, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st))))
            return fail();
        dfar = f;
        dnear = d;
        dout = io_dev_ + 2 * (size_t)S * 160;
        if (clean) dclean = c;
    }
    TickIo tio{dfar, dnear, dclean, dout, dstride, n, far_ring_, near_ring_, clean_ring_, out_ring_, kRing, near_pos_};
    TickFlowIo fio{flow_state_, flow_plans_, far_frames_, far_old_, ms_per_session ? ms_dev_ : nullptr, flags_per_session ? flags_dev_ : nullptr,
                   ms, 0, fs_};
    const bool ok = engine_->variant() == kVariantFast && AECM_HIP_OK(LaunchTickFlow(engine_->state_ptrs(), tio, fio, S, st));
    near_pos_ += n;
    tick_count_ += 1;
    if (!ok) return fail();
    if (host_pointers && !AECM_HIP_OK(hipMemcpy2DAsync(out, stride * 2, dout, 320, (size_t)n * 2, S, hipMemcpyDeviceToHost, st)))
        return fail();
    if (!AECM_HIP_OK(hipStreamSynchronize(st))) return fail();
    return first_rc;
}

}  // namespace aecm
