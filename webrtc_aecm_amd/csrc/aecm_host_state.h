// Host-side view of the device state image (aecm_state.h): initial image, configuration,
// echo-path import/export and the parity digest.  Pure host C++; used by the engine
// (aecm_engine.hip) and, for the CPU lane simulator, by tests/sim.
#ifndef AECM_AMD_HOST_STATE_H_
#define AECM_AMD_HOST_STATE_H_

#include <stdint.h>

#include <vector>

#include "aecm_state.h"

namespace aecm {

constexpr int kDigestWords = 24;

struct StreamImage {
    std::vector<uint32_t> vec;    // kNumVec * 64
    std::vector<int32_t> scal;    // 64
    StreamImage() : vec(kVecWordsPerStream, 0u), scal(kNumScal, 0) {}
};

// WebRtcAecm_InitCore (reference aecm/aecm_core.cc:358-473) + the wrapper's default
// set_config(cng = 1, echoMode = 3) (aecm/echo_control_mobile.cc:183-188).  The far history is
// all-zero and is cleared separately.  Returns false for an unsupported rate.
bool BuildInitImage(int fs, StreamImage *img);

// The core part of WebRtcAecm_set_config (echo_control_mobile.cc:410-479).  Returns false on a
// parameter outside {0,1} x {0..4}.
bool ApplyConfig(int32_t *scal, int cng_mode, int echo_mode);

// WebRtcAecm_Control (aecm_core.cc:477-482).
void ApplyControl(int32_t *scal, int fixed_delay, int nlp_flag);

// WebRtcAecm_InitEchoPathCore (aecm_core.cc:249-265) / WebRtcAecm_GetEchoPath (echo_control_mobile.cc:507-532).
void SetEchoPath(uint32_t *vec, int32_t *scal, const int16_t path[kBins]);
void GetEchoPath(const uint32_t *vec, const int32_t *scal, int16_t path[kBins]);

// Is this (vec, scal) image one the kernels may run on?  Everything the block kernel uses as an index, a lane number or
// a shift count, and every value range its cheaper-instruction shortcuts rely on (aecm_ops.h: as_i16 / as_nonneg /
// mul24 / checked_shift31 claims that hold for every state the algorithm itself can reach): int16 fields inside int16,
// supGain >= 0, Q domains <= 14, counters and flags in their ranges.  fs: the rate the image claims (mult * 8000).
// Used by WebRtcAecmBatch_ImportState; returns the name of the first offending field, or nullptr when the image is sane.
const char *ValidateStateImage(const uint32_t *vec, const int32_t *scal, int fs);

// The read-only constants blob of the kernels (layout: aecm_state.h, kConstBlobWords words).
void BuildKernelConstants(std::vector<uint32_t> *blob);

// 24-word digest of one stream's state in the oracle's canonical order
// (oracle/aecm_oracle.c:aecm_oracle_digest, include/aecm_batch.h).
void ComputeDigest(const uint32_t *vec, const int32_t *scal, const uint16_t *hist, uint32_t digest[kDigestWords]);

}  // namespace aecm
#endif  // AECM_AMD_HOST_STATE_H_
