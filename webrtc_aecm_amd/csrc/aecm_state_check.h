// Is a stream's state image (aecm_state.h) one the kernels may run on?  ONE statement of the rules, for the host
// (ValidateStateImage, aecm_host_state.cpp: WebRtcAecmBatch_ImportState of a single blob) and for the device (the validation
// pass of the bulk import, aecm_kernels.hip, where thread t of a workgroup checks scalar field t and lane t of the lane words).
//
// Checked: everything the block kernel uses as an index, a lane number or a shift count, and every value range its
// cheaper-instruction shortcuts rely on (aecm_ops.h: as_i16 / as_nonneg / mul24 / checked_shift31 claims that hold for every
// state the algorithm itself can reach): int16 members inside int16, supGain >= 0, Q domains <= 14, counters and flags in
// their ranges.  A blob no run of the algorithm can produce is refused, not run.
#ifndef AECM_AMD_STATE_CHECK_H_
#define AECM_AMD_STATE_CHECK_H_

#include <stdint.h>

#include "aecm_state.h"

#if defined(__HIPCC__)
#define AECM_CHECK_HD __host__ __device__ inline
#else
#define AECM_CHECK_HD inline
#endif

namespace aecm {

enum StateDefect : int {
    kDefectNone = 0, kDefectMult, kDefectHistPos, kDefectLastDelay, kDefectFixedDelay, kDefectStartup, kDefectQDomain, kDefectCng,
    kDefectFlag, kDefectNoiseCtr64, kDefectInt16Member, kDefectSupGain, kDefectSeed, kDefectDelayProbability, kDefectNoise64,
    kDefectFarQDomains, kDefectNoise, kDefectMeanBitCounts, kNumStateDefects
};

AECM_CHECK_HD const char *StateDefectName(int d) {
    switch (d) {
        case kDefectMult: return "mult";
        case kDefectHistPos: return "far_history_pos";
        case kDefectLastDelay: return "last_delay";
        case kDefectFixedDelay: return "fixedDelay";
        case kDefectStartup: return "startupState";
        case kDefectQDomain: return "dfaQDomain";
        case kDefectCng: return "cngMode";
        case kDefectFlag: return "flag";
        case kDefectNoiseCtr64: return "noiseEstCtr[64]";
        case kDefectInt16Member: return "int16 member";
        case kDefectSupGain: return "supGain";
        case kDefectSeed: return "seed";
        case kDefectDelayProbability: return "delay probability";
        case kDefectNoise64: return "noiseEst[64]";
        case kDefectFarQDomains: return "far_q_domains";
        case kDefectNoise: return "noiseEst";
        case kDefectMeanBitCounts: return "mean_bit_counts";
        default: return nullptr;
    }
}

// Scalar field f (ScalField) holding v, in a stream that claims the sampling rate fs.
AECM_CHECK_HD int ScalarFieldDefect(int f, int32_t v, int fs) {
    const auto in = [v](int lo, int hi) { return v >= lo && v <= hi; };
    switch (f) {
        // indices, lanes, shift counts
        case S_MULT: return (fs == 8000 || fs == 16000) && v * 8000 == fs ? 0 : kDefectMult;
        case S_HISTPOS: return in(0, kHistory) ? 0 : kDefectHistPos;
        case S_LAST_DELAY: return in(-2, kHistory - 1) ? 0 : kDefectLastDelay;
        case S_FIXED_DELAY: return in(-32768, kHistory - 1) ? 0 : kDefectFixedDelay;
        case S_STARTUP: return in(0, 2) ? 0 : kDefectStartup;
        // Q domains are norms of a non-negative int16 maximum (WebRtcSpl_NormW16, spl_inl.h:108: at most 14 -- Q 15 would need a
        // negative one); the nearFilt update's range arguments (aecm_wave.h: near_filt_update) are written for |dQ| <= 14
        case S_DFANOISYQ: case S_DFANOISYQ_OLD: case S_DFACLEANQ: case S_DFACLEANQ_OLD: return in(0, 14) ? 0 : kDefectQDomain;
        // flags and small counters
        case S_CNG: return in(0, 1) ? 0 : kDefectCng;
        case S_CURVAD: case S_FIRSTVAD: case S_FAR_INIT: case S_NEAR_INIT: return in(0, 1) ? 0 : kDefectFlag;
        case S_B64_LOWCTR: case S_B64_HIGHCTR: return in(0, 7) ? 0 : kDefectNoiseCtr64;
        // the reference's int16 members (the kernel treats their narrowing casts as the identity)
        case S_FARLOG: case S_FE_MIN: case S_FE_MAX: case S_FE_MAXMIN: case S_FE_VAD: case S_FE_MSE: case S_VADCNT: case S_MSECNT:
        case S_SUPGAIN_OLD: case S_NOISECTR: case S_NLP: case S_SG_A: case S_SG_D: case S_SG_DAB: case S_SG_DBD: case S_B64_CHSTORED:
        case S_B64_CHADAPT16: case S_B64_NEARFILT:
            return in(-32768, 32767) ? 0 : kDefectInt16Member;
        case S_SUPGAIN: return in(0, 32767) ? 0 : kDefectSupGain;            // a smoothed maximum of non-negative targets (aecm_core.cc:1000-1052)
        case S_SEED: return v >= 0 ? 0 : kDefectSeed;                           // the LCG state is 31 bits (spl.cc:129-147)
        case S_MIN_PROB: case S_LAST_PROB: return v >= 0 ? 0 : kDefectDelayProbability;
        case S_B64_NOISE: return v >= 0 ? 0 : kDefectNoise64;
        default: return 0;
    }
}

// Lane t of the lane-vector words V_NEARFILT, V_NOISE, V_M01.
AECM_CHECK_HD int LaneWordsDefect(int t, uint32_t near_filt_word, uint32_t noise_word, uint32_t m01_word) {
    if (((near_filt_word >> 22) & 31u) > 14u || (t < kSecondPass && (near_filt_word >> 27) > 14u)) return kDefectFarQDomains;
    if ((int32_t)noise_word < 0) return kDefectNoise;
    // no slot t + 64 for t >= 36: that half stays 0
    if ((m01_word & 0xffffu) > (32u << 9) || (m01_word >> 16) > (t < kSecondPass ? (32u << 9) : 0u)) return kDefectMeanBitCounts;
    return 0;
}

// The header of a snapshot blob (WebRtcAecmBatch_ExportState(s)): pins the layout the blob was written with.
struct SnapshotHeader {
    uint32_t magic, version, fs, num_vec, num_scal, history, lanes, reserved;
};
constexpr uint32_t kSnapshotMagic = 0x53434541u;      // "AECS"
constexpr uint32_t kStateLayoutVersion = 3;           // bump whenever aecm_state.h's field lists change
constexpr size_t kStateHeaderBytes = sizeof(SnapshotHeader);
constexpr size_t kStateBlobBytes = kStateHeaderBytes + kVecWordsPerStream * 4 + size_t(kNumScal) * 4 + kHistWordsPerStream * 2;
static_assert(kStateHeaderBytes == 32 && kStateBlobBytes % 16 == 0, "snapshot blob layout");
AECM_CHECK_HD bool SnapshotHeaderOk(const SnapshotHeader &h) {
    return h.magic == kSnapshotMagic && h.version == kStateLayoutVersion && h.num_vec == (uint32_t)kNumVec && h.num_scal == (uint32_t)kNumScal &&
           h.history == (uint32_t)kHistory && h.lanes == (uint32_t)kLanes && (h.fs == 8000u || h.fs == 16000u);
}

}  // namespace aecm
#endif  // AECM_AMD_STATE_CHECK_H_
