#include "aecm_sessions.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/echo_control_mobile.h"
#include "aecm_state_check.h"

namespace aecm {

#define AECM_HIP_OK(expr) ((expr) == hipSuccess)

SessionBatch *SessionBatch::Create(int num_streams, int device_id) {
    BatchEngine *e = BatchEngine::Create(num_streams, device_id);
    if (!e) return nullptr;
    SessionBatch *b = new SessionBatch();
    b->engine_.reset(e);
    b->device_ = device_id;
    const size_t S = (size_t)num_streams;
    const bool ok = AECM_HIP_OK(hipMalloc((void **)&b->far_ring_, S * kRing * 2)) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->near_ring_, S * kRing * 2)) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->out_ring_, S * kRing * 2)) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->io_dev_, 4 * S * 160 * 2)) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->flow_state_, S * kFlowFieldsUsed * sizeof(int32_t))) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->flow_plans_, S * kFlowPlanWords * sizeof(int32_t))) &&
                    AECM_HIP_OK(hipMalloc((void **)&b->far_frames_, S * kFlowFarFrameRing * 2)) &&
                    // (the per-session msInSndCardBuf / flags slots: pinned host arrays the planning kernel reads in place, below)
                    AECM_HIP_OK(hipMalloc((void **)&b->far_old_, S * 2 * kFlowFrame * 2));
    bool slots = ok;
    for (int k = 0; k < kArgSlots && slots; ++k)
        slots = AECM_HIP_OK(hipHostMalloc((void **)&b->ms_host_[k], S * sizeof(int16_t), hipHostMallocMapped)) &&
                AECM_HIP_OK(hipHostMalloc((void **)&b->flags_host_[k], S, hipHostMallocMapped)) &&
                AECM_HIP_OK(hipHostGetDevicePointer((void **)&b->ms_dev_[k], b->ms_host_[k], 0)) &&
                AECM_HIP_OK(hipHostGetDevicePointer((void **)&b->flags_dev_[k], b->flags_host_[k], 0)) &&
                AECM_HIP_OK(hipEventCreateWithFlags(&b->slot_read_[k], hipEventDisableTiming));
    if (!slots) {
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
    (void)hipFree(io_dev_);
    (void)hipFree(flow_state_);
    (void)hipFree(flow_plans_);
    (void)hipFree(far_frames_);
    (void)hipFree(far_old_);
    for (int k = 0; k < kArgSlots; ++k) {
        if (ms_host_[k]) (void)hipHostFree(ms_host_[k]);
        if (flags_host_[k]) (void)hipHostFree(flags_host_[k]);
        if (slot_read_[k]) (void)hipEventDestroy(slot_read_[k]);
    }
}

// WebRtcAecm_Init of the wrapper side of sessions [first, first + count): one launch (aecm_kernels.h).
bool SessionBatch::ResetFlowRows(int first, int count) {
    TickIo tio{nullptr, nullptr, nullptr, nullptr, 0, 0, far_ring_, near_ring_, clean_ring_, out_ring_, kRing, 0};
    TickFlowIo fio{flow_state_, flow_plans_, far_frames_, far_old_, nullptr, nullptr, 0, 0, fs_};
    return AECM_HIP_OK(LaunchResetSessions(tio, fio, engine_->num_streams(), first, count, engine_->stream()));
}

int32_t SessionBatch::Init(int32_t samp_freq) {
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    const size_t bytes = (size_t)S * kRing * 2;
    if (!AECM_HIP_OK(hipMemsetAsync(near_ring_, 0, bytes, engine_->stream())) ||
        (clean_ring_ && !AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, engine_->stream()))) || !ResetFlowRows(0, S))
        return AECM_UNSPECIFIED_ERROR;
    near_pos_ = 0;
    fs_ = samp_freq;
    poisoned_ = false;
    return 0;
}

int32_t SessionBatch::CheckSession(int session) const {
    if (fs_ == 0) return AECM_UNINITIALIZED_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    if (session < 0 || session >= engine_->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

// WebRtcAecm_Init of ONE session (same sampling rate as the batch): fresh core state, fresh wrapper state, and rings
// that read as never written (a fresh jitter buffer's read pointer can be moved back over never-written memory, the
// output ring is stuffed from it: ring_buffer.c:75-82).
int32_t SessionBatch::InitSession(int session) {
    if (int32_t rc = CheckSession(session)) return rc;
    // two launches, ordered before the next tick on the object's stream: no synchronisation with the host
    const bool ok = AECM_HIP_OK(hipSetDevice(device_)) && engine_->InitStreams(session, 1) && ResetFlowRows(session, 1);
    if (!ok) { poisoned_ = true; return AECM_UNSPECIFIED_ERROR; }
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
    if (fs_ == 0) return AECM_UNINITIALIZED_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (echo_mode < 0 || echo_mode > 4) {
        if (!engine_->SetCngMode(cng_mode, 0, -1)) return AECM_UNSPECIFIED_ERROR;
        return AECM_BAD_PARAMETER_ERROR;
    }
    return engine_->SetConfig(cng_mode, echo_mode, 0, -1) ? 0 : AECM_UNSPECIFIED_ERROR;
}

// One tick: a planning launch (one lane per session) and a tick launch (one wavefront per session); nothing per session
// happens on the host beyond handing over the tick's msInSndCardBuf / flags.  The return codes need no device either:
// the only thing a call of an initialised session with valid arguments can return is the warning for an out-of-range
// msInSndCardBuf (echo_control_mobile.cc:258-265).
int32_t SessionBatch::Fail() {
    poisoned_ = true;
    (void)hipStreamSynchronize(engine_->stream());      // async copies from caller / pinned memory may still be in flight
    return AECM_UNSPECIFIED_ERROR;
}

// The caller's events are parameters: a stale or foreign handle is refused before anything is touched (a query of a
// live event answers "done" or "not ready", never anything else), and refused without consequences for the sessions.
bool SessionBatch::EventUsable(void *ev) const {
    if (!ev) return true;
    const hipError_t q = hipEventQuery(static_cast<hipEvent_t>(ev));
    if (q == hipSuccess || q == hipErrorNotReady) return true;
    (void)hipGetLastError();
    return false;
}

// The next of the pinned per-session argument slots; when it is going to be written (needed), wait -- normally not at all --
// until the launch that last read it has run.
int SessionBatch::AcquireArgSlot(bool needed, bool *ok) {
    const int slot = slot_;
    slot_ = (slot_ + 1) % kArgSlots;
    *ok = true;
    if (needed && slot_busy_[slot]) {
        *ok = AECM_HIP_OK(hipEventSynchronize(slot_read_[slot]));
        slot_busy_[slot] = false;
    }
    return slot;
}

int32_t SessionBatch::BufferFarend(const int16_t *far, int64_t stride, size_t n_samples, int32_t calls, const uint8_t *calls_per_session,
                                   bool host_pointers, bool wait, void *wait_event, void *done_event) {
    if (far == nullptr) return AECM_NULL_POINTER_ERROR;                                  // the order of WebRtcAecm_GetBufferFarendError (:195-213)
    if (fs_ == 0) return AECM_UNINITIALIZED_ERROR;
    if (n_samples != 80 && n_samples != 160) return AECM_BAD_PARAMETER_ERROR;
    if (calls < 0 || calls > 255 || stride < (int64_t)n_samples * calls) return AECM_BAD_PARAMETER_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams(), n = (int)n_samples;
    if (calls_per_session)
        for (int s = 0; s < S; ++s)
            if (calls_per_session[s] > calls) return AECM_BAD_PARAMETER_ERROR;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    if (!EventUsable(wait_event) || !EventUsable(done_event)) return AECM_BAD_PARAMETER_ERROR;
    hipStream_t st = engine_->stream();
    if (calls > 0) {
        bool slot_ok;
        const int slot = AcquireArgSlot(calls_per_session != nullptr, &slot_ok);
        if (!slot_ok) return Fail();
        if (calls_per_session) memcpy(flags_host_[slot], calls_per_session, (size_t)S);     // pinned + mapped: the kernel reads it in place
        if (wait_event && !AECM_HIP_OK(hipStreamWaitEvent(st, static_cast<hipEvent_t>(wait_event), 0))) {
            (void)hipGetLastError();
            return AECM_BAD_PARAMETER_ERROR;                                              // nothing has been enqueued: no poison
        }
        TickFlowIo fio{flow_state_, flow_plans_, far_frames_, far_old_, nullptr, nullptr, 0, 0, fs_};
        const uint8_t *calls_dev = calls_per_session ? flags_dev_[slot] : nullptr;
        bool ok = true;
        if (!host_pointers) {
            TickIo tio{far, nullptr, nullptr, nullptr, stride, n, far_ring_, nullptr, nullptr, nullptr, kRing, 0};
            ok = AECM_HIP_OK(LaunchBufferFarend(tio, fio, calls_dev, 0, calls, S, st));
        } else {
            // host audio: staged through the ticks' device rows, as many calls per round as they hold
            const int per_round = (4 * 160) / n;
            for (int base = 0; base < calls && ok; base += per_round) {
                const int round = std::min(per_round, calls - base);
                const size_t width = (size_t)round * n * 2;
                ok = AECM_HIP_OK(hipMemcpy2DAsync(io_dev_, width, far + (size_t)base * n, (size_t)stride * 2, width, S, hipMemcpyHostToDevice, st));
                TickIo tio{io_dev_, nullptr, nullptr, nullptr, (int64_t)round * n, n, far_ring_, nullptr, nullptr, nullptr, kRing, 0};
                ok = ok && AECM_HIP_OK(LaunchBufferFarend(tio, fio, calls_dev, base, round, S, st));
            }
        }
        if (!ok) return Fail();
        if (calls_per_session) {
            if (!AECM_HIP_OK(hipEventRecord(slot_read_[slot], st))) return Fail();
            slot_busy_[slot] = true;
        }
    }
    if (done_event && !AECM_HIP_OK(hipEventRecord(static_cast<hipEvent_t>(done_event), st))) return Fail();
    if ((wait || host_pointers) && !AECM_HIP_OK(hipStreamSynchronize(st))) return Fail();
    return 0;
}

int32_t SessionBatch::Enqueue(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride, size_t n_samples,
                              int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes,
                              bool host_pointers, void *wait_event, void *done_event, int flags) {
    if (near == nullptr || out == nullptr) return AECM_NULL_POINTER_ERROR;
    if (far == nullptr) {                 // only a tick in which nobody makes a WebRtcAecm_BufferFarend call needs no far rows
        bool nobody = flags_per_session ? true : (flags & kNoFarend) != 0;
        if (flags_per_session && fs_ != 0)
            for (int s = 0; s < engine_->num_streams() && nobody; ++s) nobody = (flags_per_session[s] & kNoFarend) != 0;
        if (!nobody) return AECM_NULL_POINTER_ERROR;
    }
    if (fs_ == 0) return AECM_UNINITIALIZED_ERROR;
    if (n_samples != 80 && n_samples != 160) return AECM_BAD_PARAMETER_ERROR;       // compared as size_t: 2^32 + 80 is not 80
    if (stride < (int64_t)n_samples) return AECM_BAD_PARAMETER_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    // the tick kernel exists for the fast cross-lane primitives only: a batch switched to the safe variant is told so (and
    // stays usable after switching back) instead of being poisoned
    if (engine_->variant() != kVariantFast) return AECM_UNSUPPORTED_FUNCTION_ERROR;
    const int n = (int)n_samples;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    if (!EventUsable(wait_event) || !EventUsable(done_event)) return AECM_BAD_PARAMETER_ERROR;
    if (n != 160) {
        uint8_t any = flags_per_session ? 0 : (uint8_t)flags;
        if (flags_per_session)
            for (int s = 0; s < S; ++s) any |= flags_per_session[s];
        if (any & kSplitCalls) return AECM_BAD_PARAMETER_ERROR;                           // two 80-sample calls need 160 samples
    }
    if (clean && !clean_ring_) {
        const size_t bytes = (size_t)S * kRing * 2;
        if (!AECM_HIP_OK(hipMalloc((void **)&clean_ring_, bytes))) { clean_ring_ = nullptr; return AECM_UNSPECIFIED_ERROR; }
        if (!AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, st))) {
            (void)hipFree(clean_ring_);
            clean_ring_ = nullptr;
            return Fail();
        }
    }
    auto fail = [&]() -> int32_t { return Fail(); };
    // this tick's argument slot
    bool slot_ok;
    const int slot = AcquireArgSlot(ms_per_session || flags_per_session, &slot_ok);
    if (!slot_ok) return fail();
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
        memcpy(ms_host_[slot], ms_per_session, (size_t)S * sizeof(int16_t));      // pinned + mapped
    } else {
        first_rc = code_of(ms);
        if (codes)
            for (int s = 0; s < S; ++s) codes[s] = first_rc;
    }
    if (flags_per_session) {
        memcpy(flags_host_[slot], flags_per_session, (size_t)S);
    }
    // without far rows (nobody buffers a far frame in this tick) the kernel's far row pointer is never used for a sample that
    // counts; it still has to be an address: the near rows
    if (far == nullptr) far = near;
    const int16_t *dfar = far, *dnear = near, *dclean = clean;
    int16_t *dout = out;
    int64_t dstride = stride;
    // Host audio: staged in device rows of n samples; rows that are dense on the host too travel as one copy each way.
    auto copy_rows = [&](void *dst, size_t dpitch, const void *src, size_t spitch, hipMemcpyKind kind) {
        const size_t width = (size_t)n * 2;
        if (dpitch == width && spitch == width) return AECM_HIP_OK(hipMemcpyAsync(dst, src, width * S, kind, st));
        return AECM_HIP_OK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, S, kind, st));
    };
    if (host_pointers) {
        dstride = n;
        const size_t plane = (size_t)S * 160;
        int16_t *f = io_dev_, *d = io_dev_ + plane, *c = io_dev_ + 3 * plane;
        if (!copy_rows(f, (size_t)n * 2, far, (size_t)stride * 2, hipMemcpyHostToDevice) ||
            !copy_rows(d, (size_t)n * 2, near, (size_t)stride * 2, hipMemcpyHostToDevice) ||
            (clean && !copy_rows(c, (size_t)n * 2, clean, (size_t)stride * 2, hipMemcpyHostToDevice)))
            return fail();
        dfar = f;
        dnear = d;
        dout = io_dev_ + 2 * plane;
        if (clean) dclean = c;
    }
    TickIo tio{dfar, dnear, dclean, dout, dstride, n, far_ring_, near_ring_, clean_ring_, out_ring_, kRing, near_pos_};
    TickFlowIo fio{flow_state_, flow_plans_, far_frames_, far_old_, ms_per_session ? ms_dev_[slot] : nullptr,
                   flags_per_session ? flags_dev_[slot] : nullptr, ms, flags & (kNoFarend | kSplitCalls), fs_};
    if (wait_event && !AECM_HIP_OK(hipStreamWaitEvent(st, static_cast<hipEvent_t>(wait_event), 0))) {
        (void)hipGetLastError();
        return AECM_BAD_PARAMETER_ERROR;      // nothing of this tick has been enqueued (device pointers: no staging copies): no poison
    }
    const bool ok = AECM_HIP_OK(LaunchTickFlow(engine_->state_ptrs(), tio, fio, S, st));
    near_pos_ += n;
    if (!ok) return fail();
    if (ms_per_session || flags_per_session) {            // both launches of the tick are behind this event; only the first reads the slot
        if (!AECM_HIP_OK(hipEventRecord(slot_read_[slot], st))) return fail();
        slot_busy_[slot] = true;
    }
    if (host_pointers && !copy_rows(out, (size_t)stride * 2, dout, (size_t)n * 2, hipMemcpyDeviceToHost)) return fail();
    if (done_event && !AECM_HIP_OK(hipEventRecord(static_cast<hipEvent_t>(done_event), st))) return fail();
    return first_rc;
}

int32_t SessionBatch::Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride, size_t n_samples,
                           int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session, int32_t *codes,
                           bool host_pointers, int flags) {
    const int32_t rc = Enqueue(far, near, clean, out, stride, n_samples, ms, ms_per_session, flags_per_session, codes, host_pointers,
                               nullptr, nullptr, flags);
    if (rc != 0 && rc != AECM_BAD_PARAMETER_WARNING) return rc;            // nothing was enqueued (or the object is poisoned)
    if (!AECM_HIP_OK(hipStreamSynchronize(engine_->stream()))) return Fail();
    return rc;
}

int32_t SessionBatch::TickAsync(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride,
                                size_t n_samples, int16_t ms, const int16_t *ms_per_session, const uint8_t *flags_per_session,
                                int32_t *codes, void *wait_event, void *done_event, int flags) {
    return Enqueue(far, near, clean, out, stride, n_samples, ms, ms_per_session, flags_per_session, codes, false, wait_event, done_event, flags);
}

// ---- session snapshots ---------------------------------------------------------------------------------
namespace {
struct SessionSnapshotHeader {
    uint32_t magic, version, fs, has_clean, flow_words, far_ring, out_tail, near_tail;
};
constexpr uint32_t kSessionMagic = 0x4e534541u;       // "AESN"
constexpr uint32_t kSessionVersion = 1;               // bump with aecm_flow_plan.h's field list or the layout below
static_assert(sizeof(SessionSnapshotHeader) == SessionBatch::kSessionHeaderBytes, "session snapshot header");
struct SessionOffsets {                                // byte offsets of the parts behind the header
    size_t state = SessionBatch::kSessionHeaderBytes, flow = state + BatchEngine::kStateBytes, far = flow + kFlowWords * 4,
           out = far + kFlowFarRing * 2, near = out + SessionBatch::kOutTail * 2, clean = near + SessionBatch::kNearTail * 2,
           frames = clean + SessionBatch::kNearTail * 2, old = frames + kFlowFarFrameRing * 2, end = old + 2 * kFlowFrame * 2;
};
static_assert(SessionOffsets().end == SessionBatch::kSessionBytes, "session snapshot size");
}  // namespace

int32_t SessionBatch::ExportSession(int session, void *buf) {
    if (buf == nullptr) return AECM_NULL_POINTER_ERROR;
    if (int32_t rc = CheckSession(session)) return rc;
    if (!AECM_HIP_OK(hipSetDevice(device_)) || !engine_->Synchronize()) return Fail();
    const int S = engine_->num_streams();
    const SessionOffsets at;
    uint8_t *p = static_cast<uint8_t *>(buf);
    const SessionSnapshotHeader h{kSessionMagic, kSessionVersion, (uint32_t)fs_, clean_ring_ ? 1u : 0u, (uint32_t)kFlowWords, (uint32_t)kRing,
                                  (uint32_t)kOutTail, (uint32_t)kNearTail};
    memcpy(p, &h, sizeof h);
    if (!engine_->ExportState(session, p + at.state)) return Fail();
    int32_t flow[kFlowWords] = {0};
    std::vector<int16_t> row(kRing);
    auto tail = [&](const int16_t *ring_row, uint32_t end_pos, int n, uint8_t *dst) -> bool {      // ring positions [end_pos - n, end_pos)
        if (!AECM_HIP_OK(hipMemcpy(row.data(), ring_row, kRing * 2, hipMemcpyDeviceToHost))) return false;
        int16_t *d = reinterpret_cast<int16_t *>(dst);
        for (int k = 0; k < n; ++k) d[k] = row[(end_pos - (uint32_t)n + (uint32_t)k) & (uint32_t)(kRing - 1)];
        return true;
    };
    bool ok = AECM_HIP_OK(hipMemcpy2D(flow, sizeof(int32_t), flow_state_ + session, (size_t)S * sizeof(int32_t), sizeof(int32_t), kFlowFieldsUsed,
                                      hipMemcpyDeviceToHost));
    memcpy(p + at.flow, flow, sizeof flow);
    ok = ok && AECM_HIP_OK(hipMemcpy(p + at.far, far_ring_ + (size_t)session * kRing, kRing * 2, hipMemcpyDeviceToHost)) &&
         tail(out_ring_ + (size_t)session * kRing, (uint32_t)flow[F_BLK_POS], kOutTail, p + at.out) &&
         tail(near_ring_ + (size_t)session * kRing, (uint32_t)near_pos_, kNearTail, p + at.near);
    if (ok && clean_ring_) ok = tail(clean_ring_ + (size_t)session * kRing, (uint32_t)near_pos_, kNearTail, p + at.clean);
    else memset(p + at.clean, 0, kNearTail * 2);
    ok = ok && AECM_HIP_OK(hipMemcpy(p + at.frames, far_frames_ + (size_t)session * kFlowFarFrameRing, kFlowFarFrameRing * 2, hipMemcpyDeviceToHost)) &&
         AECM_HIP_OK(hipMemcpy(p + at.old, far_old_ + (size_t)session * 2 * kFlowFrame, 2 * kFlowFrame * 2, hipMemcpyDeviceToHost));
    return ok ? 0 : Fail();
}

int32_t SessionBatch::ImportSession(int session, const void *buf) {
    if (buf == nullptr) return AECM_NULL_POINTER_ERROR;
    if (int32_t rc = CheckSession(session)) return rc;
    const SessionOffsets at;
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    SessionSnapshotHeader h;
    memcpy(&h, p, sizeof h);
    if (h.magic != kSessionMagic || h.version != kSessionVersion || h.fs != (uint32_t)fs_ || h.has_clean > 1u || h.flow_words != (uint32_t)kFlowWords ||
        h.far_ring != (uint32_t)kRing || h.out_tail != (uint32_t)kOutTail || h.near_tail != (uint32_t)kNearTail)
        return AECM_BAD_PARAMETER_ERROR;
    int32_t flow[kFlowWords];
    memcpy(flow, p + at.flow, sizeof flow);
    if (FlowStateDefect(flow) != 0) return AECM_BAD_PARAMETER_ERROR;
    SnapshotHeader core;                                     // the core state must be of the session's rate
    memcpy(&core, p + at.state, sizeof core);
    if (!SnapshotHeaderOk(core) || core.fs != h.fs) return AECM_BAD_PARAMETER_ERROR;
    {   // the block-stream blob: the same rules ImportState applies, BEFORE anything is allocated or written (a refused snapshot changes nothing)
        std::vector<uint32_t> vec(kVecWordsPerStream);
        int32_t scal[kNumScal];
        memcpy(vec.data(), p + at.state + BatchEngine::kStateHeaderBytes, kVecWordsPerStream * 4);
        memcpy(scal, p + at.state + BatchEngine::kStateHeaderBytes + kVecWordsPerStream * 4, sizeof scal);
        if (ValidateStateImage(vec.data(), scal, (int)core.fs) != nullptr) return AECM_BAD_PARAMETER_ERROR;
    }
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    if (h.has_clean && !clean_ring_) {                      // as the first tick that carries a clean near end would
        const size_t bytes = (size_t)S * kRing * 2;
        if (!AECM_HIP_OK(hipMalloc((void **)&clean_ring_, bytes))) { clean_ring_ = nullptr; return AECM_UNSPECIFIED_ERROR; }
        if (!AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, st))) {      // a ring that may hold anything is no ring
            (void)hipFree(clean_ring_);
            clean_ring_ = nullptr;
            return Fail();
        }
    }
    // (ImportState validates the blob once more and is the first thing that writes)
    if (const int32_t rc = engine_->ImportState(session, p + at.state)) return rc;
    std::vector<int16_t> row(kRing, 0);
    auto place_tail = [&](int16_t *ring_row, uint32_t end_pos, int n, const uint8_t *src, bool zero_rest) -> bool {
        if (!zero_rest && !AECM_HIP_OK(hipMemcpy(row.data(), ring_row, kRing * 2, hipMemcpyDeviceToHost))) return false;
        if (zero_rest) std::fill(row.begin(), row.end(), (int16_t)0);
        const int16_t *s = reinterpret_cast<const int16_t *>(src);
        for (int k = 0; k < n; ++k) row[(end_pos - (uint32_t)n + (uint32_t)k) & (uint32_t)(kRing - 1)] = s[k];
        return AECM_HIP_OK(hipMemcpy(ring_row, row.data(), kRing * 2, hipMemcpyHostToDevice));
    };
    bool ok = AECM_HIP_OK(hipStreamSynchronize(st)) &&
              AECM_HIP_OK(hipMemcpy2D(flow_state_ + session, (size_t)S * sizeof(int32_t), flow, sizeof(int32_t), sizeof(int32_t), kFlowFieldsUsed,
                                      hipMemcpyHostToDevice)) &&
              AECM_HIP_OK(hipMemcpy(far_ring_ + (size_t)session * kRing, p + at.far, kRing * 2, hipMemcpyHostToDevice)) &&
              place_tail(out_ring_ + (size_t)session * kRing, (uint32_t)flow[F_BLK_POS], kOutTail, p + at.out, true) &&
              place_tail(near_ring_ + (size_t)session * kRing, (uint32_t)near_pos_, kNearTail, p + at.near, false);
    if (ok && clean_ring_) ok = place_tail(clean_ring_ + (size_t)session * kRing, (uint32_t)near_pos_, kNearTail, p + at.clean, false);
    ok = ok && AECM_HIP_OK(hipMemcpy(far_frames_ + (size_t)session * kFlowFarFrameRing, p + at.frames, kFlowFarFrameRing * 2, hipMemcpyHostToDevice)) &&
         AECM_HIP_OK(hipMemcpy(far_old_ + (size_t)session * 2 * kFlowFrame, p + at.old, 2 * kFlowFrame * 2, hipMemcpyHostToDevice));
    return ok ? 0 : Fail();           // a session half written is not a session: the object is poisoned
}

int32_t SessionBatch::Synchronize() {
    if (fs_ == 0) return AECM_UNINITIALIZED_ERROR;
    if (poisoned_) return AECM_UNSPECIFIED_ERROR;
    if (!AECM_HIP_OK(hipSetDevice(device_)) || !AECM_HIP_OK(hipStreamSynchronize(engine_->stream()))) return Fail();
    return 0;
}

}  // namespace aecm
