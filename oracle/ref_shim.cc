/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED PRODUCT.
 *
 * A small probe that is linked INTO oracle/_ref/libaecm_ref.so next to the unmodified reference
 * objects (see oracle/Makefile).  It contains no reference code: it includes the reference's own
 * headers from /root/reference at build time and only
 *   - writes the core fields WebRtcAecm_set_config writes (echo_control_mobile.cc:424-476), so a
 *     block-level driver can configure a bare AecmCore without the session wrapper, and
 *   - reads the reference's state to produce the same 24-word digest as aecm_oracle_digest(),
 *     which is how tests localise the first diverging state variable.
 */
#include <stdint.h>
#include <string.h>

#include "aecm_core.h"
#include "delay_estimator.h"
#include "echo_control_mobile.h"

namespace {
/* Mirror of the two private structs of delay_estimator_wrapper.cc:25-47 (layout only). */
union SpectrumWord { float f; int32_t i; };
struct WrapFar { SpectrumWord *mean_far_spectrum; int far_spectrum_initialized; int spectrum_size;
                 BinaryDelayEstimatorFarend *binary_farend; };
struct WrapNear { SpectrumWord *mean_near_spectrum; int near_spectrum_initialized; int spectrum_size;
                  BinaryDelayEstimator *binary_handle; };

inline uint32_t fnv_step(uint32_t h, uint32_t w) { return (h ^ w) * 16777619u; }
const uint32_t kFnvInit = 2166136261u;
inline uint32_t pack16(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | (((uint32_t)(uint16_t)hi) << 16); }
}  // namespace

extern "C" {

/* Same mapping as WebRtcAecm_set_config (echo_control_mobile.cc:410-479), applied to a bare core. */
int refshim_core_set_config(AecmCore *core, int cng_mode, int echo_mode) {
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return -1;
    if (echo_mode < 0 || echo_mode > 4) return -1;
    int a = SUPGAIN_ERROR_PARAM_A, b = SUPGAIN_ERROR_PARAM_B, d = SUPGAIN_ERROR_PARAM_D, g = SUPGAIN_DEFAULT;
    if (echo_mode < 3) { int s = 3 - echo_mode; a >>= s; b >>= s; d >>= s; g >>= s; }
    else if (echo_mode == 4) { a <<= 1; b <<= 1; d <<= 1; g <<= 1; }
    core->cngMode = (int16_t)cng_mode;
    core->supGain = (int16_t)g;
    core->supGainOld = (int16_t)g;
    core->supGainErrParamA = (int16_t)a;
    core->supGainErrParamD = (int16_t)d;
    core->supGainErrParamDiffAB = (int16_t)(a - b);
    core->supGainErrParamDiffBD = (int16_t)(b - d);
    return 0;
}

void refshim_core_digest(const AecmCore *c, uint32_t d[24]) {
    const WrapFar *wf = (const WrapFar *)c->delay_estimator_farend;
    const WrapNear *wn = (const WrapNear *)c->delay_estimator;
    const BinaryDelayEstimator *b = wn->binary_handle;
    const BinaryDelayEstimatorFarend *bf = wf->binary_farend;
    uint32_t h;
    d[0] = c->totCount;
    d[1] = c->seed;
    d[2] = pack16(c->startupState, c->far_history_pos);
    d[3] = pack16(c->dfaNoisyQDomain, c->dfaNoisyQDomainOld);
    d[4] = pack16(c->farLogEnergy, c->farEnergyMin);
    d[5] = pack16(c->farEnergyMax, c->farEnergyMaxMin);
    d[6] = pack16(c->farEnergyVAD, c->farEnergyMSE);
    d[7] = pack16(c->currentVADValue, c->vadUpdateCount);
    d[8] = pack16(c->firstVAD, c->mseChannelCount);
    d[9] = (uint32_t)c->mseAdaptOld;
    d[10] = (uint32_t)c->mseStoredOld;
    d[11] = (uint32_t)c->mseThreshold;
    d[12] = pack16(c->supGain, c->supGainOld);
    d[13] = (uint32_t)b->last_delay;
    d[14] = (uint32_t)b->minimum_probability;
    d[15] = (uint32_t)b->last_delay_probability;
    h = kFnvInit; for (int i = 0; i < PART_LEN1; ++i) h = fnv_step(h, pack16(c->channelStored[i], c->channelAdapt16[i])); d[16] = h;
    h = kFnvInit; for (int i = 0; i < PART_LEN1; ++i) h = fnv_step(h, (uint32_t)c->channelAdapt32[i]); d[17] = h;
    h = kFnvInit; for (int i = 0; i < PART_LEN1; ++i) h = fnv_step(h, (uint32_t)c->echoFilt[i]); d[18] = h;
    h = kFnvInit; for (int i = 0; i < PART_LEN1; ++i) h = fnv_step(h, (uint32_t)(uint16_t)c->nearFilt[i]); d[19] = h;
    h = kFnvInit;
    for (int i = 0; i < PART_LEN1; ++i) {
        h = fnv_step(h, (uint32_t)c->noiseEst[i]);
        h = fnv_step(h, pack16(c->noiseEstTooLowCtr[i], c->noiseEstTooHighCtr[i]));
    }
    d[20] = fnv_step(h, (uint32_t)(uint16_t)c->noiseEstCtr);
    h = kFnvInit;
    for (int i = 12; i <= 43; ++i) { h = fnv_step(h, (uint32_t)wf->mean_far_spectrum[i].i); h = fnv_step(h, (uint32_t)wn->mean_near_spectrum[i].i); }
    for (int i = 0; i < MAX_DELAY; ++i) { h = fnv_step(h, bf->binary_far_history[i]); h = fnv_step(h, (uint32_t)b->mean_bit_counts[i]); }
    d[21] = fnv_step(h, pack16(wf->far_spectrum_initialized, wn->near_spectrum_initialized));
    h = kFnvInit;
    // only entries [0, MIN_MSE_COUNT) of the log-energy histories are ever read (aecm_core.cc:943-952): the rest is dead state
    for (int i = 0; i < MIN_MSE_COUNT; ++i) { h = fnv_step(h, pack16(c->nearLogEnergy[i], c->echoAdaptLogEnergy[i])); h = fnv_step(h, (uint32_t)(uint16_t)c->echoStoredLogEnergy[i]); }
    d[22] = h;
    h = kFnvInit;
    for (int i = 0; i < PART_LEN; ++i) { h = fnv_step(h, pack16(c->xBuf[i], c->dBufNoisy[i])); h = fnv_step(h, (uint32_t)(uint16_t)c->outBuf[i]); }
    for (int p = 0; p < MAX_DELAY; ++p) {
        h = fnv_step(h, (uint32_t)c->far_q_domains[p]);
        for (int i = 0; i < PART_LEN1; ++i) h = fnv_step(h, (uint32_t)c->far_history[p * PART_LEN1 + i]);
    }
    d[23] = h;
}

/* C-linkage doors onto the reference's (C++-mangled) core entry points, aecm_core.h:149-239. */
AecmCore *refshim_core_create(void) { return WebRtcAecm_CreateCore(); }
int refshim_core_init(AecmCore *core, int fs) { return WebRtcAecm_InitCore(core, fs); }
void refshim_core_free(AecmCore *core) { WebRtcAecm_FreeCore(core); }
void refshim_core_init_echo_path(AecmCore *core, const int16_t *path) { WebRtcAecm_InitEchoPathCore(core, path); }
void refshim_core_control(AecmCore *core, int delay, int nlp_flag) { WebRtcAecm_Control(core, delay, nlp_flag); }
int refshim_core_process_block(AecmCore *core, const int16_t *far_b, const int16_t *near_b, const int16_t *clean_b,
                               int16_t *out) {
    return WebRtcAecm_ProcessBlock(core, far_b, near_b, clean_b, out);
}

/* Drive n_blocks consecutive blocks of one stream through the reference's ProcessBlock. */
int refshim_core_process_stream(AecmCore *core, const int16_t *far_s, const int16_t *near_s, int16_t *out,
                                size_t n_blocks) {
    for (size_t b = 0; b < n_blocks; ++b) {
        int r = WebRtcAecm_ProcessBlock(core, far_s + b * PART_LEN, near_s + b * PART_LEN, NULL, out + b * PART_LEN);
        if (r) return r;
    }
    return 0;
}

}  // extern "C"
