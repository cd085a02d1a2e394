// extern "C" surface of libaecm_mi355x.so: the reference's session ABI (include/echo_control_mobile.h)
// and the batch extension (include/aecm_batch.h).  Plain pointers and sizes only.
#include <hip/hip_runtime_api.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/aecm_batch.h"
#include "../../include/echo_control_mobile.h"
#include "aecm_engine.h"
#include "aecm_session.h"
#include "aecm_sessions.h"

using aecm::BatchEngine;
using aecm::Session;

struct AecmBatch {
    BatchEngine *engine;
};

struct AecmSessions {
    aecm::SessionBatch *batch;
};

extern "C" {

// ---- session ABI (reference aecm/echo_control_mobile.h:46-202) -----------------------------------

void *WebRtcAecm_Create(void) { return Session::Create(); }

void WebRtcAecm_Free(void *inst) { delete static_cast<Session *>(inst); }

int32_t WebRtcAecm_Init(void *inst, int32_t sampFreq) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->Init(sampFreq);
}

int32_t WebRtcAecm_GetBufferFarendError(void *inst, const int16_t *farend, size_t nrOfSamples) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->BufferFarendError(farend, nrOfSamples);
}

int32_t WebRtcAecm_BufferFarend(void *inst, const int16_t *farend, size_t nrOfSamples) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->BufferFarend(farend, nrOfSamples);
}

int32_t WebRtcAecm_Process(void *inst, const int16_t *nearendNoisy, const int16_t *nearendClean, int16_t *out,
                           size_t nrOfSamples, int16_t msInSndCardBuf) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->Process(nearendNoisy, nearendClean, out, nrOfSamples, msInSndCardBuf);
}

int32_t WebRtcAecm_set_config(void *inst, AecmConfig config) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->SetConfig(config.cngMode, config.echoMode);
}

int32_t WebRtcAecm_InitEchoPath(void *inst, const void *echo_path, size_t size_bytes) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->InitEchoPath(echo_path, size_bytes);
}

int32_t WebRtcAecm_GetEchoPath(void *inst, void *echo_path, size_t size_bytes) {
    if (inst == nullptr) return -1;
    return static_cast<Session *>(inst)->GetEchoPath(echo_path, size_bytes);
}

size_t WebRtcAecm_echo_path_size_bytes(void) { return aecm::kBins * sizeof(int16_t); }

// ---- batch extension -------------------------------------------------------------------------------

AecmBatch *WebRtcAecmBatch_Create(int32_t num_streams, int32_t device_id) {
    BatchEngine *e = BatchEngine::Create(num_streams, device_id);
    if (!e) return nullptr;
    AecmBatch *b = new (std::nothrow) AecmBatch{e};
    if (!b) delete e;
    return b;
}

void WebRtcAecmBatch_Free(AecmBatch *b) {
    if (!b) return;
    delete b->engine;
    delete b;
}

int32_t WebRtcAecmBatch_num_streams(const AecmBatch *b) { return b ? b->engine->num_streams() : -1; }

int32_t WebRtcAecmBatch_Init(AecmBatch *b, int32_t sampFreq) {
    if (!b) return -1;
    if (sampFreq != 8000 && sampFreq != 16000) return AECM_BAD_PARAMETER_ERROR;
    return b->engine->Init(sampFreq) ? 0 : AECM_UNSPECIFIED_ERROR;
}

static int32_t CheckRange(const AecmBatch *b, int32_t first, int32_t *count) {
    if (!b) return -1;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (*count < 0) *count = b->engine->num_streams() - first;
    if (first < 0 || *count < 0 || first + *count > b->engine->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

int32_t WebRtcAecmBatch_set_config(AecmBatch *b, AecmConfig config, int32_t first, int32_t count) {
    if (int32_t rc = CheckRange(b, first, &count)) return rc;
    if (config.cngMode != AecmFalse && config.cngMode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (config.echoMode < 0 || config.echoMode > 4) return AECM_BAD_PARAMETER_ERROR;
    return b->engine->SetConfig(config.cngMode, config.echoMode, first, count) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_Control(AecmBatch *b, int32_t fixed_delay, int32_t nlp_flag, int32_t first, int32_t count) {
    if (int32_t rc = CheckRange(b, first, &count)) return rc;
    // the reference uses fixedDelay unchecked as a far-history offset (aecm_core.cc:157-172, MAX_DELAY = 100 slots)
    // and narrows both arguments to int16_t: validated as the values the core will hold -- anything outside
    // [-32768, kHistory) is refused (a large negative int32 would otherwise wrap into range or beyond it)
    if (fixed_delay < -32768 || fixed_delay >= aecm::kHistory) return AECM_BAD_PARAMETER_ERROR;
    if (nlp_flag < -32768 || nlp_flag > 32767) return AECM_BAD_PARAMETER_ERROR;
    return b->engine->Control(fixed_delay, nlp_flag, first, count) ? 0 : AECM_UNSPECIFIED_ERROR;
}

static int32_t CheckIo(const AecmBatch *b, const void *far_p, const void *near_p, const void *out_p, int32_t num_blocks) {
    if (!b) return -1;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (!far_p || !near_p || !out_p) return AECM_NULL_POINTER_ERROR;
    if (num_blocks < 0) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

int32_t WebRtcAecmBatch_ProcessBlocks(AecmBatch *b, const int16_t *far_dev, const int16_t *near_dev,
                                      const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                      int64_t block_stride, int32_t num_blocks) {
    if (int32_t rc = CheckIo(b, far_dev, near_dev, out_dev, num_blocks)) return rc;
    if (num_blocks == 0) return 0;
    aecm::IoView io{far_dev, near_dev, near_clean_dev, out_dev, stream_stride, block_stride};
    return b->engine->ProcessBlocks(io, num_blocks) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_ProcessBlocksHost(AecmBatch *b, const int16_t *far_host, const int16_t *near_host,
                                          const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                          int64_t block_stride, int32_t num_blocks) {
    if (int32_t rc = CheckIo(b, far_host, near_host, out_host, num_blocks)) return rc;
    if (num_blocks == 0) return 0;
    aecm::IoView io{far_host, near_host, near_clean_host, out_host, stream_stride, block_stride};
    return b->engine->ProcessBlocksHost(io, num_blocks) ? 0 : AECM_UNSPECIFIED_ERROR;
}

static int32_t ProcessRecordings(AecmBatch *b, const int16_t *far_p, const int16_t *near_p, const int16_t *clean_p,
                                 int16_t *out_p, int64_t stream_stride, int32_t samples_per_call, int32_t num_calls, int16_t ms,
                                 bool host) {
    if (int32_t rc = CheckIo(b, far_p, near_p, out_p, num_calls)) return rc;
    if (samples_per_call != 80 && samples_per_call != 160) return AECM_BAD_PARAMETER_ERROR;
    if (stream_stride < (int64_t)samples_per_call * num_calls) return AECM_BAD_PARAMETER_ERROR;
    if (num_calls == 0) return 0;
    int32_t rc = 0;
    if (!b->engine->ProcessRecordings(far_p, near_p, clean_p, out_p, stream_stride, samples_per_call, num_calls, ms, host, &rc))
        return AECM_UNSPECIFIED_ERROR;
    return rc;
}

int32_t WebRtcAecmBatch_ProcessRecordings(AecmBatch *b, const int16_t *far_dev, const int16_t *near_dev,
                                          const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                          int32_t samples_per_call, int32_t num_calls, int16_t msInSndCardBuf) {
    return ProcessRecordings(b, far_dev, near_dev, near_clean_dev, out_dev, stream_stride, samples_per_call, num_calls,
                             msInSndCardBuf, false);
}

int32_t WebRtcAecmBatch_ProcessRecordingsHost(AecmBatch *b, const int16_t *far_host, const int16_t *near_host,
                                              const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                              int32_t samples_per_call, int32_t num_calls, int16_t msInSndCardBuf) {
    return ProcessRecordings(b, far_host, near_host, near_clean_host, out_host, stream_stride, samples_per_call, num_calls,
                             msInSndCardBuf, true);
}

int32_t WebRtcAecmBatch_Synchronize(AecmBatch *b) {
    if (!b) return -1;
    return b->engine->Synchronize() ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_GetLastLaunchMs(AecmBatch *b, float *ms) {
    if (!b) return -1;
    if (!ms) return AECM_NULL_POINTER_ERROR;
    return b->engine->LastLaunchMs(ms) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_GetTimers(AecmBatch *b, double *total_ms, int64_t *launches) {
    if (!b) return -1;
    if (!total_ms || !launches) return AECM_NULL_POINTER_ERROR;
    return b->engine->Timers(total_ms, launches) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_ResetTimers(AecmBatch *b) {
    if (!b) return -1;
    b->engine->ResetTimers();
    return 0;
}

int32_t WebRtcAecmBatch_InitEchoPath(AecmBatch *b, int32_t stream, const void *echo_path, size_t size_bytes) {
    if (!b) return -1;
    if (!echo_path) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != aecm::kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (stream < 0 || stream >= b->engine->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    int16_t tmp[aecm::kBins];
    memcpy(tmp, echo_path, sizeof tmp);
    return b->engine->SetEchoPath(stream, tmp) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_GetEchoPath(AecmBatch *b, int32_t stream, void *echo_path, size_t size_bytes) {
    if (!b) return -1;
    if (!echo_path) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != aecm::kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (stream < 0 || stream >= b->engine->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    int16_t tmp[aecm::kBins];
    if (!b->engine->GetEchoPath(stream, tmp)) return AECM_UNSPECIFIED_ERROR;
    memcpy(echo_path, tmp, sizeof tmp);
    return 0;
}

size_t WebRtcAecmBatch_state_size_bytes(void) { return BatchEngine::kStateBytes; }

static int32_t CheckState(const AecmBatch *b, int32_t stream, const void *state, size_t size_bytes) {
    if (!b) return -1;
    if (!state) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != BatchEngine::kStateBytes) return AECM_BAD_PARAMETER_ERROR;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (stream < 0 || stream >= b->engine->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

int32_t WebRtcAecmBatch_ExportState(AecmBatch *b, int32_t stream, void *state, size_t size_bytes) {
    if (int32_t rc = CheckState(b, stream, state, size_bytes)) return rc;
    return b->engine->ExportState(stream, state) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_ImportState(AecmBatch *b, int32_t stream, const void *state, size_t size_bytes) {
    if (int32_t rc = CheckState(b, stream, state, size_bytes)) return rc;
    return b->engine->ImportState(stream, state);
}

static int32_t CheckStates(const AecmBatch *b, int32_t first, int32_t count, const void *states, size_t size_bytes) {
    if (!b) return -1;
    if (!states) return AECM_NULL_POINTER_ERROR;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (first < 0 || count < 0 || count > b->engine->num_streams() - first) return AECM_BAD_PARAMETER_ERROR;
    if (size_bytes != (size_t)count * BatchEngine::kStateBytes) return AECM_BAD_PARAMETER_ERROR;
    return 0;
}

int32_t WebRtcAecmBatch_ExportStates(AecmBatch *b, int32_t first, int32_t count, void *states_host, size_t size_bytes) {
    if (int32_t rc = CheckStates(b, first, count, states_host, size_bytes)) return rc;
    return b->engine->ExportStates(first, count, states_host, false) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_ImportStates(AecmBatch *b, int32_t first, int32_t count, const void *states_host, size_t size_bytes) {
    if (int32_t rc = CheckStates(b, first, count, states_host, size_bytes)) return rc;
    return b->engine->ImportStates(first, count, states_host, false);
}

int32_t WebRtcAecmBatch_ExportStatesDevice(AecmBatch *b, int32_t first, int32_t count, void *states_dev, size_t size_bytes) {
    if (int32_t rc = CheckStates(b, first, count, states_dev, size_bytes)) return rc;
    if (reinterpret_cast<uintptr_t>(states_dev) & 3) return AECM_BAD_PARAMETER_ERROR;        // the kernels move 32-bit words
    return b->engine->ExportStates(first, count, states_dev, true) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_ImportStatesDevice(AecmBatch *b, int32_t first, int32_t count, const void *states_dev, size_t size_bytes) {
    if (int32_t rc = CheckStates(b, first, count, states_dev, size_bytes)) return rc;
    return b->engine->ImportStates(first, count, states_dev, true);
}

int32_t WebRtcAecmBatch_GetDigest(AecmBatch *b, int32_t stream, uint32_t digest[AECM_BATCH_DIGEST_WORDS]) {
    if (!b) return -1;
    if (!digest) return AECM_NULL_POINTER_ERROR;
    if (!b->engine->initialized()) return AECM_UNINITIALIZED_ERROR;
    if (stream < 0 || stream >= b->engine->num_streams()) return AECM_BAD_PARAMETER_ERROR;
    return b->engine->Digest(stream, digest) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_SetKernelVariant(AecmBatch *b, int32_t variant) {
    if (!b) return -1;
    if (variant != AECM_KERNEL_SAFE && variant != AECM_KERNEL_FAST) return AECM_BAD_PARAMETER_ERROR;
    b->engine->set_variant(variant);
    return 0;
}

int32_t WebRtcAecmBatch_SetLaunchChunking(AecmBatch *b, int32_t chunk_blocks, int32_t min_streams) {
    if (!b) return -1;
    if (chunk_blocks < 0 || chunk_blocks > (1 << 20)) return AECM_BAD_PARAMETER_ERROR;
    b->engine->set_queue_chunk(chunk_blocks, min_streams);
    return 0;
}

int32_t WebRtcAecmBatch_SetLaunchPipelining(AecmBatch *b, int32_t min_streams) {
    if (!b) return -1;
    b->engine->set_pipelined_min_streams(min_streams);
    return 0;
}

int32_t WebRtcAecmBatch_DescribeLaunch(const AecmBatch *b, int32_t num_blocks, int32_t has_clean_input, int32_t *chunk_blocks) {
    if (!b) return -1;
    return b->engine->DescribeLaunch(num_blocks, has_clean_input != 0, chunk_blocks);
}

int32_t WebRtcAecmBatch_DescribeLaunchFor(int32_t num_streams, int32_t compute_units, int32_t num_blocks, int32_t has_clean_input,
                                          int32_t *chunk_blocks) {
    if (num_streams <= 0 || compute_units <= 0 || num_blocks <= 0) return -1;
    const aecm::LaunchDescription d = aecm::DescribeLaunchWith(aecm::DefaultLaunchPolicy(compute_units), aecm::kVariantFast, num_streams, num_blocks,
                                                               has_clean_input != 0);
    if (chunk_blocks) *chunk_blocks = d.form == 2 ? d.chunk_blocks : d.form == 3 ? d.shape : 0;
    return d.form;
}

// ---- the launch policy as one value (aecm_engine.h: LaunchPolicy) --------------------------------------
static void PolicyToAbi(const aecm::LaunchPolicy &p, AecmLaunchPolicy *o) {
    *o = AecmLaunchPolicy{(int32_t)sizeof(AecmLaunchPolicy), p.compute_units, p.queue_chunk_blocks, p.queue_chunk_explicit ? 1 : 0, p.queue_min_streams,
                          p.pipelined_min_streams, p.pipelined_min_blocks, p.pipelined_max_streams, p.resident_waves, p.rotation_stream_limit,
                          p.pipe.tail_waves, p.pipe.front_waves, p.pipe.raw, p.pipe.delay_waves, p.pipe.gain_waves, p.pipe.spread, p.pipe.wgs_per_cu, p.pipe.rot, p.pipe.prio};
}
static aecm::LaunchPolicy PolicyFromAbi(const AecmLaunchPolicy &a) {
    aecm::LaunchPolicy p;
    p.compute_units = a.compute_units;
    p.queue_chunk_blocks = a.queue_chunk_blocks;
    p.queue_chunk_explicit = a.queue_chunk_explicit != 0;
    p.queue_min_streams = a.queue_min_streams;
    p.pipelined_min_streams = a.pipelined_min_streams > 0 ? a.pipelined_min_streams : 0x7fffffff;
    p.pipelined_min_blocks = a.pipelined_min_blocks;
    p.pipelined_max_streams = a.pipelined_max_streams;
    p.resident_waves = a.resident_waves;
    p.rotation_stream_limit = a.rotation_stream_limit;
    p.pipe.tail_waves = a.pipe_tail_waves;
    p.pipe.front_waves = a.pipe_front_waves;
    p.pipe.raw = a.pipe_raw;
    p.pipe.delay_waves = a.pipe_delay_waves;
    p.pipe.gain_waves = a.pipe_gain_waves;
    p.pipe.spread = a.pipe_spread;
    p.pipe.wgs_per_cu = a.pipe_wgs_per_cu;
    p.pipe.rot = a.pipe_rot;
    p.pipe.prio = a.pipe_prio;
    return p;
}

int32_t WebRtcAecmBatch_DefaultLaunchPolicy(int32_t compute_units, AecmLaunchPolicy *policy) {
    if (!policy) return AECM_NULL_POINTER_ERROR;
    if (compute_units <= 0) return AECM_BAD_PARAMETER_ERROR;
    PolicyToAbi(aecm::DefaultLaunchPolicy(compute_units), policy);
    return 0;
}

int32_t WebRtcAecmBatch_GetLaunchPolicy(const AecmBatch *b, AecmLaunchPolicy *policy) {
    if (!b) return -1;
    if (!policy) return AECM_NULL_POINTER_ERROR;
    PolicyToAbi(b->engine->launch_policy(), policy);
    return 0;
}

int32_t WebRtcAecmBatch_SetLaunchPolicy(AecmBatch *b, const AecmLaunchPolicy *policy) {
    if (!b) return -1;
    if (!policy) return AECM_NULL_POINTER_ERROR;
    if (policy->struct_size != (int32_t)sizeof(AecmLaunchPolicy)) return AECM_BAD_PARAMETER_ERROR;
    return b->engine->set_launch_policy(PolicyFromAbi(*policy)) ? 0 : AECM_BAD_PARAMETER_ERROR;
}

static void DescriptionToAbi(const aecm::LaunchDescription &d, AecmLaunchDescription *o) {
    *o = AecmLaunchDescription{d.form, d.chunk_blocks, d.shape, d.workgroups, d.waves_per_workgroup, d.workgroups_per_cu, d.rounds_x1000, d.cu_load_evenness_x1000};
}

int32_t WebRtcAecmBatch_DescribeLaunchDetail(const AecmLaunchPolicy *policy, int32_t compute_units, int32_t num_streams, int32_t num_blocks,
                                             int32_t has_clean_input, AecmLaunchDescription *out) {
    if (!out) return AECM_NULL_POINTER_ERROR;
    if (num_streams <= 0 || num_blocks <= 0) return AECM_BAD_PARAMETER_ERROR;
    aecm::LaunchPolicy p;
    if (policy) {
        if (policy->struct_size != (int32_t)sizeof(AecmLaunchPolicy)) return AECM_BAD_PARAMETER_ERROR;
        p = PolicyFromAbi(*policy);
        if (!aecm::LaunchPolicyValid(p) || p.compute_units <= 0) return AECM_BAD_PARAMETER_ERROR;
    } else {
        if (compute_units <= 0) return AECM_BAD_PARAMETER_ERROR;
        p = aecm::DefaultLaunchPolicy(compute_units);
    }
    DescriptionToAbi(aecm::DescribeLaunchWith(p, aecm::kVariantFast, num_streams, num_blocks, has_clean_input != 0), out);
    return 0;
}

int32_t WebRtcAecmSessions_DescribeTick(int32_t num_sessions, int32_t compute_units, AecmLaunchDescription *out) {
    if (!out) return AECM_NULL_POINTER_ERROR;
    if (num_sessions <= 0 || compute_units <= 0) return AECM_BAD_PARAMETER_ERROR;
    DescriptionToAbi(aecm::DescribeTickLaunch(num_sessions, compute_units), out);
    return 0;
}

int32_t WebRtcAecm_SetDefaultDevice(int32_t device_id) {
    if (device_id < 0) return AECM_BAD_PARAMETER_ERROR;
    Session::SetDefaultDevice(device_id);
    return 0;
}

// ---- streaming batch of sessions ---------------------------------------------------------------------

AecmSessions *WebRtcAecmSessions_Create(int32_t num_streams, int32_t device_id) {
    aecm::SessionBatch *sb = aecm::SessionBatch::Create(num_streams, device_id);
    if (!sb) return nullptr;
    return new AecmSessions{sb};
}

void WebRtcAecmSessions_Free(AecmSessions *s) {
    if (!s) return;
    delete s->batch;
    delete s;
}

int32_t WebRtcAecmSessions_Init(AecmSessions *s, int32_t sampFreq) { return s ? s->batch->Init(sampFreq) : -1; }

int32_t WebRtcAecmSessions_set_config(AecmSessions *s, AecmConfig config) {
    return s ? s->batch->SetConfig(config.cngMode, config.echoMode) : -1;
}

int32_t WebRtcAecmSessions_InitSession(AecmSessions *s, int32_t session) { return s ? s->batch->InitSession(session) : -1; }

size_t WebRtcAecmSessions_session_size_bytes(void) { return aecm::SessionBatch::kSessionBytes; }

int32_t WebRtcAecmSessions_ExportSession(AecmSessions *s, int32_t session, void *snapshot, size_t size_bytes) {
    if (!s) return -1;
    if (!snapshot) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != aecm::SessionBatch::kSessionBytes) return AECM_BAD_PARAMETER_ERROR;
    return s->batch->ExportSession(session, snapshot);
}

int32_t WebRtcAecmSessions_ImportSession(AecmSessions *s, int32_t session, const void *snapshot, size_t size_bytes) {
    if (!s) return -1;
    if (!snapshot) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != aecm::SessionBatch::kSessionBytes) return AECM_BAD_PARAMETER_ERROR;
    return s->batch->ImportSession(session, snapshot);
}

int32_t WebRtcAecmSessions_set_config_session(AecmSessions *s, int32_t session, AecmConfig config) {
    return s ? s->batch->SetConfigSession(session, config.cngMode, config.echoMode) : -1;
}

int32_t WebRtcAecmSessions_InitEchoPath(AecmSessions *s, int32_t session, const void *echo_path, size_t size_bytes) {
    return s ? s->batch->InitEchoPathSession(session, echo_path, size_bytes) : -1;
}

int32_t WebRtcAecmSessions_GetEchoPath(AecmSessions *s, int32_t session, void *echo_path, size_t size_bytes) {
    return s ? s->batch->GetEchoPathSession(session, echo_path, size_bytes) : -1;
}

// nrOfSamples stays a size_t all the way down: 2^32 + 80 must be refused like the reference does, not narrowed to 80.
int32_t WebRtcAecmSessions_Tick(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples,
                                int16_t msInSndCardBuf) {
    if (!s) return -1;
    return s->batch->Tick(far_dev, near_dev, near_clean_dev, out_dev, stream_stride, nrOfSamples, msInSndCardBuf, nullptr,
                          nullptr, nullptr, false);
}

int32_t WebRtcAecmSessions_TickHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                    const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                    size_t nrOfSamples, int16_t msInSndCardBuf) {
    if (!s) return -1;
    return s->batch->Tick(far_host, near_host, near_clean_host, out_host, stream_stride, nrOfSamples, msInSndCardBuf,
                          nullptr, nullptr, nullptr, true);
}

int32_t WebRtcAecmSessions_TickPerSession(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                          const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                          size_t nrOfSamples, const int16_t *msInSndCardBuf_host, int32_t *codes_host) {
    if (!s) return -1;
    if (!msInSndCardBuf_host) return AECM_NULL_POINTER_ERROR;
    return s->batch->Tick(far_dev, near_dev, near_clean_dev, out_dev, stream_stride, nrOfSamples, 0, msInSndCardBuf_host,
                          nullptr, codes_host, false);
}

int32_t WebRtcAecmSessions_TickPerSessionHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                              const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                              size_t nrOfSamples, const int16_t *msInSndCardBuf_host, int32_t *codes_host) {
    if (!s) return -1;
    if (!msInSndCardBuf_host) return AECM_NULL_POINTER_ERROR;
    return s->batch->Tick(far_host, near_host, near_clean_host, out_host, stream_stride, nrOfSamples, 0,
                          msInSndCardBuf_host, nullptr, codes_host, true);
}

int32_t WebRtcAecmSessions_TickFlags(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                     const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples,
                                     const int16_t *msInSndCardBuf_host, const uint8_t *flags_host, int32_t *codes_host) {
    if (!s) return -1;
    if (!msInSndCardBuf_host || !flags_host) return AECM_NULL_POINTER_ERROR;
    return s->batch->Tick(far_dev, near_dev, near_clean_dev, out_dev, stream_stride, nrOfSamples, 0, msInSndCardBuf_host,
                          flags_host, codes_host, false);
}

int32_t WebRtcAecmSessions_TickFlagsHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                         const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                         size_t nrOfSamples, const int16_t *msInSndCardBuf_host, const uint8_t *flags_host,
                                         int32_t *codes_host) {
    if (!s) return -1;
    if (!msInSndCardBuf_host || !flags_host) return AECM_NULL_POINTER_ERROR;
    return s->batch->Tick(far_host, near_host, near_clean_host, out_host, stream_stride, nrOfSamples, 0,
                          msInSndCardBuf_host, flags_host, codes_host, true);
}

int32_t WebRtcAecmSessions_TickAsync(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev, const int16_t *near_clean_dev,
                                     int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples, int16_t msInSndCardBuf,
                                     const int16_t *msInSndCardBuf_host, const uint8_t *flags_host, int32_t *codes_host,
                                     void *wait_hip_event, void *done_hip_event) {
    if (!s) return -1;
    if (flags_host && !msInSndCardBuf_host) return AECM_NULL_POINTER_ERROR;
    return s->batch->TickAsync(far_dev, near_dev, near_clean_dev, out_dev, stream_stride, nrOfSamples, msInSndCardBuf,
                               msInSndCardBuf_host, flags_host, codes_host, wait_hip_event, done_hip_event);
}

int32_t WebRtcAecmSessions_BufferFarend(AecmSessions *s, const int16_t *far_dev, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                        const uint8_t *calls_host) {
    return s ? s->batch->BufferFarend(far_dev, stream_stride, nrOfSamples, calls, calls_host, false, true, nullptr, nullptr) : -1;
}

int32_t WebRtcAecmSessions_BufferFarendHost(AecmSessions *s, const int16_t *far_host, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                            const uint8_t *calls_host) {
    return s ? s->batch->BufferFarend(far_host, stream_stride, nrOfSamples, calls, calls_host, true, true, nullptr, nullptr) : -1;
}

int32_t WebRtcAecmSessions_BufferFarendAsync(AecmSessions *s, const int16_t *far_dev, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                             const uint8_t *calls_host, void *wait_hip_event, void *done_hip_event) {
    return s ? s->batch->BufferFarend(far_dev, stream_stride, nrOfSamples, calls, calls_host, false, false, wait_hip_event, done_hip_event) : -1;
}

int32_t WebRtcAecmSessions_Process(AecmSessions *s, const int16_t *near_dev, const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                   size_t nrOfSamples, int16_t msInSndCardBuf, const int16_t *msInSndCardBuf_host, int32_t *codes_host) {
    if (!s) return -1;
    return s->batch->Tick(nullptr, near_dev, near_clean_dev, out_dev, stream_stride, nrOfSamples, msInSndCardBuf, msInSndCardBuf_host, nullptr,
                          codes_host, false, AECM_SESSION_NO_FAREND);
}

int32_t WebRtcAecmSessions_ProcessHost(AecmSessions *s, const int16_t *near_host, const int16_t *near_clean_host, int16_t *out_host,
                                       int64_t stream_stride, size_t nrOfSamples, int16_t msInSndCardBuf, const int16_t *msInSndCardBuf_host,
                                       int32_t *codes_host) {
    if (!s) return -1;
    return s->batch->Tick(nullptr, near_host, near_clean_host, out_host, stream_stride, nrOfSamples, msInSndCardBuf, msInSndCardBuf_host, nullptr,
                          codes_host, true, AECM_SESSION_NO_FAREND);
}

int32_t WebRtcAecmSessions_Synchronize(AecmSessions *s) { return s ? s->batch->Synchronize() : -1; }

int32_t WebRtcAecmSessions_SetKernelVariant(AecmSessions *s, int32_t variant) {
    if (!s) return -1;
    if (variant != AECM_KERNEL_SAFE && variant != AECM_KERNEL_FAST) return AECM_BAD_PARAMETER_ERROR;
    s->batch->engine()->set_variant(variant);
    return 0;
}

// Caller-owned host audio, pinned and mapped once: the alias returned is a device pointer every *_dev argument accepts.
int32_t WebRtcAecmBatch_RegisterHostBuffer(int32_t device_id, void *host, size_t size_bytes, void **device_alias) {
    if (!host || !device_alias) return AECM_NULL_POINTER_ERROR;
    if (size_bytes == 0) return AECM_BAD_PARAMETER_ERROR;
    if (hipSetDevice(device_id) != hipSuccess) return AECM_BAD_PARAMETER_ERROR;
    const hipError_t e = hipHostRegister(host, size_bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorHostMemoryAlreadyRegistered ? AECM_BAD_PARAMETER_ERROR : AECM_UNSPECIFIED_ERROR;
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || !dev) {
        (void)hipHostUnregister(host);
        return AECM_UNSPECIFIED_ERROR;
    }
    *device_alias = dev;
    return 0;
}

int32_t WebRtcAecmBatch_UnregisterHostBuffer(int32_t device_id, void *host) {
    if (!host) return AECM_NULL_POINTER_ERROR;
    if (hipSetDevice(device_id) != hipSuccess) return AECM_BAD_PARAMETER_ERROR;
    if (hipDeviceSynchronize() != hipSuccess) return AECM_UNSPECIFIED_ERROR;      // nothing may still be reading or writing it
    if (hipHostUnregister(host) != hipSuccess) {
        (void)hipGetLastError();                       // not a registered buffer: the caller's mistake, not a sticky device error
        return AECM_BAD_PARAMETER_ERROR;
    }
    return 0;
}

int32_t WebRtcAecmBatch_SelfTest(int32_t device_id, int32_t exhaustive, uint64_t failures[8]) {
    if (!failures) return AECM_NULL_POINTER_ERROR;
    if (hipSetDevice(device_id) != hipSuccess) return AECM_UNSPECIFIED_ERROR;
    uint64_t *dev = nullptr;
    uint32_t *consts = nullptr;
    std::vector<uint32_t> blob;
    aecm::BuildKernelConstants(&blob);
    if (hipMalloc((void **)&dev, 8 * sizeof(uint64_t)) != hipSuccess) return AECM_UNSPECIFIED_ERROR;
    if (hipMalloc((void **)&consts, blob.size() * 4) != hipSuccess) { (void)hipFree(dev); return AECM_UNSPECIFIED_ERROR; }
    int32_t rc = AECM_UNSPECIFIED_ERROR;
    if (hipMemcpy(consts, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemset(dev, 0, 8 * sizeof(uint64_t)) == hipSuccess && aecm::LaunchSelfTest(dev, exhaustive, consts, nullptr) == hipSuccess &&
        hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(failures, dev, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess)
        rc = 0;
    (void)hipFree(dev);
    (void)hipFree(consts);
    return rc;
}

int32_t WebRtcAecmBatch_GetCheckCounters(int32_t device_id, uint64_t counters[2], int32_t reset) {
    if (!counters) return AECM_NULL_POINTER_ERROR;
    if (hipSetDevice(device_id) != hipSuccess) return AECM_UNSPECIFIED_ERROR;
    const hipError_t e = aecm::ReadCheckCounters(counters, reset != 0);
    if (e == hipErrorNotSupported) return AECM_UNSUPPORTED_FUNCTION_ERROR;
    return e == hipSuccess ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t WebRtcAecmBatch_DebugFft128(int32_t device_id, int16_t *data_host, int32_t *scales_host, int32_t variant,
                                    int32_t kernel_variant, int32_t count) {
    if (!data_host || !scales_host) return AECM_NULL_POINTER_ERROR;
    if (variant < 0 || variant > 2 || count < 0) return AECM_BAD_PARAMETER_ERROR;
    if (count == 0) return 0;
    if (hipSetDevice(device_id) != hipSuccess) return AECM_UNSPECIFIED_ERROR;
    std::vector<uint32_t> blob;
    aecm::BuildKernelConstants(&blob);
    int16_t *data = nullptr;
    int32_t *scales = nullptr;
    uint32_t *consts = nullptr;
    const size_t bytes = (size_t)count * 256 * sizeof(int16_t);
    int32_t rc = AECM_UNSPECIFIED_ERROR;
    if (hipMalloc((void **)&data, bytes) == hipSuccess && hipMalloc((void **)&scales, (size_t)count * 4) == hipSuccess &&
        hipMalloc((void **)&consts, blob.size() * 4) == hipSuccess &&
        hipMemcpy(consts, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(data, data_host, bytes, hipMemcpyHostToDevice) == hipSuccess &&
        aecm::LaunchFft128(data, scales, variant, kernel_variant != 0, count, consts, nullptr) == hipSuccess &&
        hipDeviceSynchronize() == hipSuccess && hipMemcpy(data_host, data, bytes, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(scales_host, scales, (size_t)count * 4, hipMemcpyDeviceToHost) == hipSuccess)
        rc = 0;
    (void)hipFree(data);
    (void)hipFree(scales);
    (void)hipFree(consts);
    return rc;
}

int32_t WebRtcAecmBatch_DeviceInfo(int32_t device_id, char *name, size_t name_len, int32_t *compute_units,
                                   int32_t *clock_khz) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return AECM_UNSPECIFIED_ERROR;
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    return 0;
}

int32_t WebRtcAecmBatch_DevicePciBusId(int32_t device_id, char *bus_id, size_t bus_id_len) {
    if (!bus_id) return AECM_NULL_POINTER_ERROR;
    if (bus_id_len < 13) return AECM_BAD_PARAMETER_ERROR;                     // "0000:00:00.0" + NUL
    return hipDeviceGetPCIBusId(bus_id, (int)bus_id_len, device_id) == hipSuccess ? 0 : AECM_UNSPECIFIED_ERROR;
}

}  // extern "C"
