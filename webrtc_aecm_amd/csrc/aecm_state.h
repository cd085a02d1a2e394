// Device-resident per-stream state layout of the MI355X AECM engine (shared by host and device).
//
// One wavefront (64 lanes) owns one stream.  A stream's state is three HBM regions:
//
//   vec  : uint32_t [S][kNumVec][64]   "lane vectors": word v, lane t  -> one 256-byte line per
//                                       field per wave, loaded/stored once per launch
//   scal : int32_t  [S][64]            wave-uniform scalars (+ everything that belongs to bin 64)
//   hist : uint16_t [S][100][64]       far-spectrum history, bins 0..63 (reference far_history,
//                                       aecm/aecm_core.h:64); one 128-byte row written and (when
//                                       the estimated delay is not 0) one row read per block
//
// Lane t owns frequency bin t (0..63), time samples t and t+64, delay-estimator history slots t and
// t+64 (<100) and log-energy history entry t.  Bin 64 (the real-only Nyquist bin) is wave-uniform and
// lives in scal[].  The field list mirrors the live subset of the reference's AecmCore
// (aecm/aecm_core.h:41-141) and delay estimator (aecm/delay_estimator.h:22-63,
// aecm/delay_estimator_wrapper.cc:25-47); see DESIGN.md section 3 for the mapping table.
#ifndef AECM_AMD_STATE_H_
#define AECM_AMD_STATE_H_

#include <stddef.h>
#include <stdint.h>

namespace aecm {

constexpr int kBlock = 64;      // PART_LEN   (reference aecm/aecm_defines.h:19)
constexpr int kBins = 65;       // PART_LEN1  (aecm_defines.h:22)
constexpr int kHistory = 100;   // MAX_DELAY  (aecm_defines.h:26)
constexpr int kLanes = 64;

// ---- lane-vector words ------------------------------------------------------------------------
// 12 words per lane (3 072 B per stream) + the scalar block = 3 328 B round trip per launch.  Fields whose live part is
// narrower than 64 lanes x 32 bits share a word: the 100-slot delay-estimator arrays spill 36 slots into a second
// lane pass, the three log-energy histories are only ever read in their first 20 entries (reference
// aecm_core.cc:943-952) and live in the unused lanes 36..55 of those second-pass words, the binary-spectrum
// thresholds only exist for bins 12..43 so the far and near sets interleave in one word, and 5-bit / 3-bit
// quantities ride in the spare bits of the nearFilt word.  In registers every field has its own lane vector
// (aecm_wave.h: load_state / store_state do the packing once per launch).
enum VecField : int {
    V_XD_OLD = 0,   // lo16: xBuf[t] (previous far block), hi16: dBufNoisy[t] (previous near block)
    V_OUTBUF,       // lo16: outBuf[bitrev6(t)] (overlap-add tail, kept in IFFT output lane order)
                    // hi16: dBufClean[t] (previous clean-near block; only used with a clean input)
    V_CH16,         // lo16: channelStored[t], hi16: channelAdapt16[t]
    V_CH32,         // channelAdapt32[t]
    V_ECHOFILT,     // echoFilt[t]
    V_NEARFILT,     // bits 0..15 nearFilt[t] | 16..18 noiseEstTooLowCtr[t] | 19..21 noiseEstTooHighCtr[t] (both < 5)
                    // | 22..26 far_q_domains[slot t] | 27..31 far_q_domains[slot t + 64] (t < 36; Q <= 14)
    V_NOISE,        // noiseEst[t]
    V_MEAN,         // lanes 12..43: mean_far_spectrum[t]; the other lanes t: mean_near_spectrum[(t + 32) & 63]
                    // (only bins 12..43 of either exist, delay_estimator_wrapper.cc:92-125)
    V_BH0,          // binary_far_history[t]        (slot 0 = newest)
    V_BH1,          // lanes 0..35: binary_far_history[t + 64]; lanes 36..55: lo16 nearLogEnergy[t - 36], hi16 echoAdaptLogEnergy[t - 36]
    V_M01,          // lo16: mean_bit_counts[t], hi16: mean_bit_counts[t + 64] (t < 36); Q9 values <= 32 << 9
    V_HQ,           // lo16: far_history[slot t][64]; hi16: lanes 0..35 far_history[slot t + 64][64], lanes 36..55 echoStoredLogEnergy[t - 36]
    kNumVec
};
constexpr int kSecondPass = 36;      // MAX_DELAY - 64: live lanes of the second-pass words
constexpr int kLogEntries = 20;      // MIN_MSE_COUNT: entries of the log-energy histories the algorithm reads

// ---- wave-uniform scalars ---------------------------------------------------------------------
enum ScalField : int {
    S_TOTCOUNT = 0, S_SEED, S_STARTUP, S_HISTPOS,
    S_DFANOISYQ, S_DFANOISYQ_OLD, S_DFACLEANQ, S_DFACLEANQ_OLD,
    S_FARLOG, S_FE_MIN, S_FE_MAX, S_FE_MAXMIN, S_FE_VAD, S_FE_MSE,
    S_CURVAD, S_VADCNT, S_FIRSTVAD, S_MSECNT,
    S_MSE_ADAPT_OLD, S_MSE_STORED_OLD, S_MSE_THRESH,
    S_SUPGAIN, S_SUPGAIN_OLD, S_NOISECTR,
    S_FAR_INIT, S_NEAR_INIT, S_MIN_PROB, S_LAST_PROB, S_LAST_DELAY,
    // configuration (written by init / set_config / control)
    S_MULT, S_CNG, S_NLP, S_FIXED_DELAY, S_SG_A, S_SG_D, S_SG_DAB, S_SG_DBD,
    // bin 64
    S_B64_CHSTORED, S_B64_CHADAPT16, S_B64_CHADAPT32, S_B64_ECHOFILT, S_B64_NEARFILT,
    S_B64_NOISE, S_B64_LOWCTR, S_B64_HIGHCTR,
    kNumScalUsed,
    kNumScal = 64
};
static_assert(kNumScalUsed <= kNumScal, "scalar block overflow");

constexpr size_t kVecWordsPerStream = size_t(kNumVec) * kLanes;
constexpr size_t kHistWordsPerStream = size_t(kHistory) * kLanes;   // uint16 units

// Read-only constants blob shared by all streams (built once on the host, aecm_host_state.cpp):
//   rows of per-lane constants, then the image of the kernel's LDS tables.
enum LaneConstRow : int {
    LC_LCG_MUL = 0, LC_LCG_ADD,          // LCG jump-ahead A^t, C_t (reference spl.cc:129-147)
    LC_DIV_MAGIC, LC_DIV_SHIFT,          // reciprocal of (bin index + 1) (aecm_core.cc:904)
    LC_HANN_LO, LC_HANN_HI,              // analysis window hann[t] << 2, hann[64-t] << 2
    LC_HANN_SYN_LO, LC_HANN_SYN_HI,      // synthesis window in IFFT output lane order
    LC_BIN0_REAL,                        // 0x0000ffff in lane 0, all ones elsewhere: and-mask that clears bin 0's imaginary half (aecm_core_c.cc:296)
    LC_NOT_BIN0,                         // 0 in lane 0, all ones elsewhere: bin 0 gets no comfort noise (aecm_core_c.cc:146-147)
    LC_NLP_AVG_BAND,                     // all ones in lanes 4..24, 0 elsewhere: the bins the wideband NLP averages (aecm_core_c.cc:628-636)
    LC_NLP_LOW_BINS,                     // 0x7fff0000 in lanes 0..23, 0 from lane 24 on: or-ed onto the average it exempts the low bins from the clamp (:638-646)
    kLaneConstRows
};
constexpr int kLdsTwiddleWords = 7 * kLanes * 4;       // inverse transform: [stage][lane] (w_re, w_im, -w_re, -w_im)
constexpr int kLdsFwdTwiddleWords = 6 * kLanes * 4;    // forward stages 1..6: (w_re, w_im, -w_re, -w_im) per lane
constexpr int kLdsFwdOffsetWords = 3 * kLanes * 4;     // forward stages 2,4,6: accumulator offsets (see fft128)
constexpr int kLdsCosSinWords = 360;
constexpr int kLdsHannWords = 68;   // 65 entries + pad (16-byte struct alignment)
constexpr int kLdsImageWords = kLdsTwiddleWords + kLdsFwdTwiddleWords + kLdsFwdOffsetWords + kLdsCosSinWords + kLdsHannWords;
constexpr int kConstBlobWords = kLaneConstRows * kLanes + kLdsImageWords;

struct StatePtrs {
    uint32_t *vec;
    int32_t *scal;
    uint16_t *hist;
    const uint32_t *consts;   // kConstBlobWords words
};

// Strided view of the audio I/O of one launch: sample (stream s, block b, i) lives at
// base[s * stream_stride + b * block_stride + i].  Stream-major files are (T*64, 64); a tick-major
// server layout [T][S][64] is (64, S*64).
struct IoView {
    const int16_t *far;
    const int16_t *near;
    const int16_t *near_clean;   // optional (nullptr): the reference's nearendClean input
    int16_t *out;
    int64_t stream_stride;
    int64_t block_stride;
};

}  // namespace aecm
#endif  // AECM_AMD_STATE_H_
