#include "aecm_session.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../include/echo_control_mobile.h"

namespace aecm {

// The HIP device single sessions are created on (WebRtcAecm_Create has no device argument): process-wide, set through
// WebRtcAecm_SetDefaultDevice (include/aecm_batch.h).
static std::atomic<int> g_default_device{0};
void Session::SetDefaultDevice(int device) { g_default_device.store(device, std::memory_order_relaxed); }
int Session::DefaultDevice() { return g_default_device.load(std::memory_order_relaxed); }

Session *Session::Create() {
    BatchEngine *engine = BatchEngine::Create(1, DefaultDevice());
    if (!engine) return nullptr;                 // no usable GPU: there is no CPU path to fall back to
    Session *s = new Session();
    s->engine_.reset(engine);
    return s;
}

int32_t Session::Init(int32_t samp_freq) {                                     // echo_control_mobile.cc:142-191
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;               // InitCore + default config
    return flow_.Init(samp_freq);
}

int32_t Session::Process(const int16_t *near_noisy, const int16_t *near_clean, int16_t *out, size_t n, int16_t ms) {
    return flow_.Process(near_noisy, near_clean, out, n, ms,
                         [this](const int16_t *far_b, const int16_t *near_b, const int16_t *clean_b, int16_t *out_b, int nb) {
                             IoView io{far_b, near_b, clean_b, out_b, (int64_t)nb * kBlock, kBlock};
                             return engine_->ProcessBlocksHost(io, nb);   // WebRtcAecm_ProcessBlock x nb on the GPU
                         });
}

int32_t Session::SetConfig(int16_t cng_mode, int16_t echo_mode) {              // :410-479
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
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
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
    int16_t tmp[kBins];
    memcpy(tmp, path, sizeof tmp);
    return engine_->SetEchoPath(0, tmp) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t Session::GetEchoPath(void *path, size_t size_bytes) {                  // :507-532
    if (path == nullptr) return AECM_NULL_POINTER_ERROR;
    if (size_bytes != kBins * sizeof(int16_t)) return AECM_BAD_PARAMETER_ERROR;
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
    int16_t tmp[kBins];
    if (!engine_->GetEchoPath(0, tmp)) return AECM_UNSPECIFIED_ERROR;
    memcpy(path, tmp, sizeof tmp);
    return 0;
}

}  // namespace aecm
