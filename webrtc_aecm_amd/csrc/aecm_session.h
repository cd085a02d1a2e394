// Single-stream session behind the reference's public C ABI (include/echo_control_mobile.h):
// SessionFlow<int16_t> (the content-independent session wrapper + frame adapter, aecm_session_flow.h)
// on top of a one-stream BatchEngine that runs the blocks on the GPU.
#ifndef AECM_AMD_SESSION_H_
#define AECM_AMD_SESSION_H_

#include <stddef.h>
#include <stdint.h>

#include <memory>

#include "aecm_engine.h"
#include "aecm_session_flow.h"

namespace aecm {

class Session {
public:
    static Session *Create();
    static void SetDefaultDevice(int device);     // the HIP device of sessions created from now on (process-wide; default 0)
    static int DefaultDevice();
    int32_t Init(int32_t samp_freq);
    int32_t BufferFarendError(const int16_t *farend, size_t n) const { return flow_.BufferFarendError(farend, n); }
    int32_t BufferFarend(const int16_t *farend, size_t n) { return flow_.BufferFarend(farend, n); }
    int32_t Process(const int16_t *near_noisy, const int16_t *near_clean, int16_t *out, size_t n, int16_t ms_in_snd_card_buf);
    int32_t SetConfig(int16_t cng_mode, int16_t echo_mode);
    int32_t InitEchoPath(const void *path, size_t size_bytes);
    int32_t GetEchoPath(void *path, size_t size_bytes);

private:
    Session() : flow_(0) {}
    std::unique_ptr<BatchEngine> engine_;
    SessionFlow<int16_t> flow_;
};

}  // namespace aecm
#endif  // AECM_AMD_SESSION_H_
