#include "aecm_host_state.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <initializer_list>

#include "aecm_ops.h"
#include "aecm_state_check.h"
#include "aecm_tables.h"

namespace aecm {
[[noreturn]] void aecm_mul24_range_violation(int a, int b) {
    fprintf(stderr, "aecm: mul24 precondition violated (%d * %d)\n", a, b);
    abort();
}
[[noreturn]] void aecm_i16_range_violation(int v) {
    fprintf(stderr, "aecm: as_i16 precondition violated (%d)\n", v);
    abort();
}
[[noreturn]] void aecm_shift_range_violation(int c) {
    fprintf(stderr, "aecm: shift count outside [-31, 31] (%d)\n", c);
    abort();
}
[[noreturn]] void aecm_nonneg_violation(int v) {
    fprintf(stderr, "aecm: as_nonneg precondition violated (%d)\n", v);
    abort();
}
namespace {

inline uint32_t Pack16(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | (((uint32_t)(uint16_t)hi) << 16); }
inline int Lo16(uint32_t w) { return (int16_t)(w & 0xffff); }
inline uint32_t FnvStep(uint32_t h, uint32_t w) { return (h ^ w) * 16777619u; }
constexpr uint32_t kFnvInit = 2166136261u;
inline int BitRev6(int t) {
    int r = 0;
    for (int b = 0; b < 6; ++b) r |= ((t >> b) & 1) << (5 - b);
    return r;
}

}  // namespace

bool ApplyConfig(int32_t *scal, int cng_mode, int echo_mode) {
    if (cng_mode != 0 && cng_mode != 1) return false;
    if (echo_mode < 0 || echo_mode > 4) return false;
    // SUPGAIN_DEFAULT / ERROR_PARAM_A / _B / _D (reference aecm/aecm_defines.h:62-68), scaled per echoMode
    int g = 256, a = 3072, b = 1536, d = 256;
    if (echo_mode < 3) {
        const int s = 3 - echo_mode;
        g >>= s; a >>= s; b >>= s; d >>= s;
    } else if (echo_mode == 4) {
        g <<= 1; a <<= 1; b <<= 1; d <<= 1;
    }
    scal[S_CNG] = cng_mode;
    scal[S_SUPGAIN] = g;
    scal[S_SUPGAIN_OLD] = g;
    scal[S_SG_A] = a;
    scal[S_SG_D] = d;
    scal[S_SG_DAB] = a - b;
    scal[S_SG_DBD] = b - d;
    return true;
}

void ApplyControl(int32_t *scal, int fixed_delay, int nlp_flag) {
    scal[S_NLP] = (int16_t)nlp_flag;
    scal[S_FIXED_DELAY] = (int16_t)fixed_delay;
}

const char *ValidateStateImage(const uint32_t *vec, const int32_t *scal, int fs) {
    // the rules themselves: aecm_state_check.h (shared with the device-side validation of the bulk import)
    for (int f = 0; f < kNumScalUsed; ++f)
        if (const int d = ScalarFieldDefect(f, scal[f], fs)) return StateDefectName(d);
    for (int t = 0; t < kLanes; ++t)
        if (const int d = LaneWordsDefect(t, vec[V_NEARFILT * kLanes + t], vec[V_NOISE * kLanes + t], vec[V_M01 * kLanes + t]))
            return StateDefectName(d);
    return nullptr;
}

void SetEchoPath(uint32_t *vec, int32_t *scal, const int16_t path[kBins]) {
    for (int t = 0; t < kLanes; ++t) {
        vec[V_CH16 * kLanes + t] = Pack16(path[t], path[t]);
        vec[V_CH32 * kLanes + t] = (uint32_t)(int32_t)path[t] << 16;
    }
    scal[S_B64_CHSTORED] = path[64];
    scal[S_B64_CHADAPT16] = path[64];
    scal[S_B64_CHADAPT32] = (int32_t)((uint32_t)(int32_t)path[64] << 16);
    scal[S_MSE_ADAPT_OLD] = 1000;
    scal[S_MSE_STORED_OLD] = 1000;
    scal[S_MSE_THRESH] = 0x7fffffff;
    scal[S_MSECNT] = 0;
}

void GetEchoPath(const uint32_t *vec, const int32_t *scal, int16_t path[kBins]) {
    for (int t = 0; t < kLanes; ++t) path[t] = (int16_t)Lo16(vec[V_CH16 * kLanes + t]);
    path[64] = (int16_t)scal[S_B64_CHSTORED];
}

bool BuildInitImage(int fs, StreamImage *img) {
    if (fs != 8000 && fs != 16000) return false;
    std::fill(img->vec.begin(), img->vec.end(), 0u);
    std::fill(img->scal.begin(), img->scal.end(), 0);
    uint32_t *vec = img->vec.data();
    int32_t *scal = img->scal.data();
    scal[S_MULT] = fs / 8000;                               // aecm_core.cc:368
    scal[S_SEED] = 666;                                     // :385
    scal[S_HISTPOS] = kHistory;                             // :397
    scal[S_NLP] = 1;                                        // :399
    scal[S_FIXED_DELAY] = -1;                               // :400
    for (int t = 0; t < kLanes; ++t)                        // delay_estimator.cc:490-493
        vec[V_M01 * kLanes + t] = Pack16(20 << 9, t < kSecondPass ? (20 << 9) : 0);
    scal[S_MIN_PROB] = 32 << 9;                             // delay_estimator.cc:494-498
    scal[S_LAST_PROB] = 32 << 9;
    scal[S_LAST_DELAY] = -2;
    SetEchoPath(vec, scal, fs == 8000 ? kAecmChannelStored8k : kAecmChannelStored16k);   // :413-417
    {                                                       // :427-435 pink-ish initial noise floor
        int32_t t32 = kBins * kBins;
        int16_t t16 = kBins;
        int32_t noise[kBins];
        int i = 0;
        for (; i < (kBins >> 1) - 1; ++i) {
            noise[i] = t32 << 8;
            t16--;
            t32 -= (int32_t)((t16 << 1) + 1);
        }
        for (; i < kBins; ++i) noise[i] = t32 << 8;
        for (int t = 0; t < kLanes; ++t) vec[V_NOISE * kLanes + t] = (uint32_t)noise[t];
        scal[S_B64_NOISE] = noise[64];
    }
    scal[S_FE_MIN] = 32767;                                 // :437-445
    scal[S_FE_MAX] = -32768;
    scal[S_FE_VAD] = 1025;
    scal[S_FIRSTVAD] = 1;
    scal[S_CNG] = 1;                                        // :423
    return ApplyConfig(scal, 1, 3);
}

void BuildKernelConstants(std::vector<uint32_t> *blob) {
    blob->assign(kConstBlobWords, 0u);
    uint32_t *lc = blob->data();
    uint32_t a = 1, c = 0;
    for (int t = 0; t < kLanes; ++t) {
        if (t > 0) { a *= 69069u; c = c * 69069u + 1u; }      // A^t, C_t: state after t LCG steps
        lc[LC_LCG_MUL * kLanes + t] = a;
        lc[LC_LCG_ADD * kLanes + t] = c;
        int magic = 0, shift = 0;
        div_magic(t + 1, &magic, &shift);
        lc[LC_DIV_MAGIC * kLanes + t] = (uint32_t)magic;
        lc[LC_DIV_SHIFT * kLanes + t] = (uint32_t)shift;
        const int brev = BitRev6(t);
        lc[LC_HANN_LO * kLanes + t] = (uint32_t)((int32_t)kAecmSqrtHanningQ14[t] << 2);          // << 2: aecm_wave.h, window()
        lc[LC_HANN_HI * kLanes + t] = (uint32_t)((int32_t)kAecmSqrtHanningQ14[64 - t] << 2);
        lc[LC_HANN_SYN_LO * kLanes + t] = (uint32_t)(int32_t)kAecmSqrtHanningQ14[brev];
        lc[LC_HANN_SYN_HI * kLanes + t] = (uint32_t)(int32_t)kAecmSqrtHanningQ14[64 - brev];
        lc[LC_BIN0_REAL * kLanes + t] = t == 0 ? 0x0000ffffu : 0xffffffffu;
        lc[LC_NOT_BIN0 * kLanes + t] = t == 0 ? 0u : 0xffffffffu;
        lc[LC_NLP_AVG_BAND * kLanes + t] = (t >= 4 && t <= 24) ? 0xffffffffu : 0u;
        lc[LC_NLP_LOW_BINS * kLanes + t] = t < 24 ? 0x7fff0000u : 0u;
    }
    // LDS image: packed twiddles (w_re = (wr, -wi), w_im = (wi, wr)) of the inverse transform per
    // [stage][lane]: wr = cos, wi = +sin of entry m << k, m = position & (2^stage - 1) (reference
    // complex_fft.c:412-420; forward: wi = -sin, :296-303); positions are bit-reversed lanes.
    uint32_t *img = blob->data() + kLaneConstRows * kLanes;
    for (int stage = 0; stage < 7; ++stage)
        for (int t = 0; t < kLanes; ++t) {
            const int idx = (BitRev6(t) & ((1 << stage) - 1)) << (6 - stage);
            const int wr = kAecmTwiddleCosQ15[idx], wi = kAecmTwiddleSinQ15[idx];
            uint32_t *e = img + (stage * kLanes + t) * 4;
            e[0] = Pack16(wr, -wi);
            e[1] = Pack16(wi, wr);
            e[2] = Pack16(-wr, wi);                           // per-half negations (|twiddle| <= 32767: no overflow)
            e[3] = Pack16(-wi, -wr);
        }
    // Forward stages 1..6 in the multiply-add form of fft128 (aecm_wave.h): (w_re, w_im, -w_re, -w_im) per
    // lane, and for the even stages the accumulator offsets (s_re, 1 - s_re, s_im, 1 - s_im) with
    // s = sum of the two halves of the packed twiddle.
    uint32_t *ft = img + kLdsTwiddleWords;
    uint32_t *fo = ft + kLdsFwdTwiddleWords;
    for (int stage = 1; stage < 7; ++stage)
        for (int t = 0; t < kLanes; ++t) {
            const int idx = (BitRev6(t) & ((1 << stage) - 1)) << (6 - stage);
            const int wr = kAecmTwiddleCosQ15[idx], wi = -kAecmTwiddleSinQ15[idx];
            uint32_t *e = ft + ((stage - 1) * kLanes + t) * 4;
            e[0] = Pack16(wr, -wi);
            e[1] = Pack16(wi, wr);
            e[2] = Pack16(-wr, wi);
            e[3] = Pack16(-wi, -wr);
            if ((stage & 1) == 0) {
                uint32_t *o = fo + ((stage / 2 - 1) * kLanes + t) * 4;
                const int s_re = wr - wi, s_im = wi + wr;
                o[0] = (uint32_t)s_re;
                o[1] = (uint32_t)(1 - s_re);
                o[2] = (uint32_t)s_im;
                o[3] = (uint32_t)(1 - s_im);
            }
        }
    uint32_t *cs = fo + kLdsFwdOffsetWords;
    for (int i = 0; i < kLdsCosSinWords; ++i) cs[i] = Pack16(kAecmCosQ13[i], kAecmSinQ13[i]);
    uint32_t *hn = cs + kLdsCosSinWords;
    for (int i = 0; i < 65; ++i) hn[i] = (uint32_t)(int32_t)kAecmSqrtHanningQ14[i];
}

void ComputeDigest(const uint32_t *vec, const int32_t *scal, const uint16_t *hist, uint32_t d[kDigestWords]) {
    auto V = [&](int f, int t) { return vec[f * kLanes + t]; };
    uint32_t h;
    d[0] = (uint32_t)scal[S_TOTCOUNT];
    d[1] = (uint32_t)scal[S_SEED];
    d[2] = Pack16(scal[S_STARTUP], scal[S_HISTPOS]);
    d[3] = Pack16(scal[S_DFANOISYQ], scal[S_DFANOISYQ_OLD]);
    d[4] = Pack16(scal[S_FARLOG], scal[S_FE_MIN]);
    d[5] = Pack16(scal[S_FE_MAX], scal[S_FE_MAXMIN]);
    d[6] = Pack16(scal[S_FE_VAD], scal[S_FE_MSE]);
    d[7] = Pack16(scal[S_CURVAD], scal[S_VADCNT]);
    d[8] = Pack16(scal[S_FIRSTVAD], scal[S_MSECNT]);
    d[9] = (uint32_t)scal[S_MSE_ADAPT_OLD];
    d[10] = (uint32_t)scal[S_MSE_STORED_OLD];
    d[11] = (uint32_t)scal[S_MSE_THRESH];
    d[12] = Pack16(scal[S_SUPGAIN], scal[S_SUPGAIN_OLD]);
    d[13] = (uint32_t)scal[S_LAST_DELAY];
    d[14] = (uint32_t)scal[S_MIN_PROB];
    d[15] = (uint32_t)scal[S_LAST_PROB];
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) h = FnvStep(h, V(V_CH16, i));
    d[16] = FnvStep(h, Pack16(scal[S_B64_CHSTORED], scal[S_B64_CHADAPT16]));
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) h = FnvStep(h, V(V_CH32, i));
    d[17] = FnvStep(h, (uint32_t)scal[S_B64_CHADAPT32]);
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) h = FnvStep(h, V(V_ECHOFILT, i));
    d[18] = FnvStep(h, (uint32_t)scal[S_B64_ECHOFILT]);
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) h = FnvStep(h, V(V_NEARFILT, i) & 0xffffu);
    d[19] = FnvStep(h, (uint32_t)(uint16_t)scal[S_B64_NEARFILT]);
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) {
        h = FnvStep(h, V(V_NOISE, i));
        const uint32_t w = V(V_NEARFILT, i);
        h = FnvStep(h, Pack16((w >> 16) & 7, (w >> 19) & 7));
    }
    h = FnvStep(h, (uint32_t)scal[S_B64_NOISE]);
    h = FnvStep(h, Pack16(scal[S_B64_LOWCTR], scal[S_B64_HIGHCTR]));
    d[20] = FnvStep(h, (uint32_t)(uint16_t)scal[S_NOISECTR]);
    h = kFnvInit;
    for (int i = 12; i <= 43; ++i) { h = FnvStep(h, V(V_MEAN, i)); h = FnvStep(h, V(V_MEAN, (i + 32) & 63)); }
    for (int i = 0; i < kHistory; ++i) {
        h = FnvStep(h, i < 64 ? V(V_BH0, i) : V(V_BH1, i - 64));
        h = FnvStep(h, i < 64 ? (V(V_M01, i) & 0xffffu) : (V(V_M01, i - 64) >> 16));
    }
    d[21] = FnvStep(h, Pack16(scal[S_FAR_INIT], scal[S_NEAR_INIT]));
    h = kFnvInit;
    for (int i = 0; i < kLogEntries; ++i) {                  // the entries the algorithm reads (aecm_core.cc:943-952)
        h = FnvStep(h, V(V_BH1, kSecondPass + i));
        h = FnvStep(h, V(V_HQ, kSecondPass + i) >> 16);
    }
    d[22] = h;
    h = kFnvInit;
    for (int i = 0; i < kLanes; ++i) {
        h = FnvStep(h, V(V_XD_OLD, i));
        h = FnvStep(h, V(V_OUTBUF, BitRev6(i)) & 0xffffu);
    }
    for (int p = 0; p < kHistory; ++p) {
        const uint32_t nf = V(V_NEARFILT, p < 64 ? p : p - 64), hq = V(V_HQ, p < 64 ? p : p - 64);
        h = FnvStep(h, p < 64 ? ((nf >> 22) & 31u) : (nf >> 27));                // far_q_domains[p]
        for (int i = 0; i < kLanes; ++i) h = FnvStep(h, (uint32_t)hist[p * kLanes + i]);
        h = FnvStep(h, p < 64 ? (hq & 0xffffu) : (hq >> 16));                    // far_history[p][64]
    }
    d[23] = h;
}

}  // namespace aecm
