/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED PRODUCT (see aecm_oracle.h).
 *
 * Plain-C restatement of the reference AECM block path.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).  Integer semantics are the
 * reference's as compiled for x86-64 by gcc: two's complement, arithmetic >> on signed values,
 * truncating narrowing conversions, wrap-around unsigned sums.  This file is built with -fwrapv
 * and every variable shift count is masked to 5 bits (what the x86 shifter does), so nothing here
 * depends on undefined behaviour.
 *
 * Pinned against oracle/_ref/libaecm_ref.so (the unmodified reference) and tests/golden/.
 */
#include "aecm_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "aecm_oracle_tables.h"

/* ---- algorithm constants (reference aecm/aecm_defines.h:17-85) ---- */
#define CONV_LEN 512
#define CONV_LEN2 1024
#define FAR_ENERGY_MIN 1025
#define FAR_ENERGY_DIFF 929
#define ENERGY_DEV_TOL 400
#define FAR_ENERGY_VAD_REGION 230
#define MU_MIN 10
#define MU_MAX 1
#define MU_DIFF 9
#define MIN_MSE_COUNT 20
#define MIN_MSE_DIFF 29
#define MSE_RESOLUTION 5
#define RESOLUTION_CHANNEL16 12
#define RESOLUTION_CHANNEL32 28
#define CHANNEL_VAD 16
#define RESOLUTION_SUPGAIN 8
#define SUPGAIN_DEFAULT (1 << RESOLUTION_SUPGAIN)
#define SUPGAIN_ERROR_PARAM_A 3072
#define SUPGAIN_ERROR_PARAM_B 1536
#define SUPGAIN_ERROR_PARAM_D SUPGAIN_DEFAULT
#define SUPGAIN_EPC_DT 200
#define ONE_Q14 (1 << 14)
#define NLP_COMP_LOW 3277
#define NLP_COMP_HIGH ONE_Q14
/* delay estimator constants (aecm/delay_estimator.cc:23-28, delay_estimator.h kMaxBitCountsQ9,
 * delay_estimator_wrapper.cc:50-55) */
#define BAND_FIRST 12
#define BAND_LAST 43
#define MAX_BIT_COUNTS_Q9 (32 << 9)
#define PROB_OFFSET 1024
#define PROB_LOWER_LIMIT 8704
#define PROB_MIN_SPREAD 2816

struct AecmOracle {
    /* configuration written by init / set_config / control */
    int mult;                 /* aecm_core.cc:368 */
    int cng_mode;             /* aecm_core.h:110 */
    int nlp_flag;             /* aecm_core.cc:399 */
    int fixed_delay;          /* aecm_core.cc:400 */
    int16_t sg_err_a, sg_err_d, sg_err_diff_ab, sg_err_diff_bd; /* aecm_core.cc:451-454 */

    /* time-domain carry-over: first halves of xBuf/dBufNoisy/dBufClean, outBuf */
    int16_t x_old[ORC_BLOCK], d_old[ORC_BLOCK], c_old[ORC_BLOCK], out_ovl[ORC_BLOCK];

    /* far spectrum history (aecm_core.h:64-66) */
    uint16_t far_hist[ORC_HISTORY][ORC_BINS];
    int far_hist_q[ORC_HISTORY];
    int far_hist_pos;

    /* delay estimator (delay_estimator_wrapper.cc:25-47, delay_estimator.h:22-63) */
    int32_t mean_far[ORC_BINS], mean_near[ORC_BINS];
    int far_init, near_init;
    uint32_t bin_far_hist[ORC_HISTORY];
    int far_bit_counts[ORC_HISTORY];
    int32_t mean_bit_counts[ORC_HISTORY];
    int32_t minimum_probability;
    int32_t last_delay_probability;
    int last_delay;

    /* channel + filters (aecm_core.h:87-108) */
    int16_t ch_stored[ORC_BINS], ch_adapt16[ORC_BINS];
    int32_t ch_adapt32[ORC_BINS];
    int32_t echo_filt[ORC_BINS];
    int16_t near_filt[ORC_BINS];
    int32_t noise_est[ORC_BINS];
    int noise_low_ctr[ORC_BINS], noise_high_ctr[ORC_BINS];
    int16_t noise_est_ctr;

    /* energies / VAD (aecm_core.h:73-127) */
    int16_t near_log[ORC_LOGBUF], echo_adapt_log[ORC_LOGBUF], echo_stored_log[ORC_LOGBUF];
    int16_t far_log;
    int16_t far_energy_min, far_energy_max, far_energy_maxmin, far_energy_vad, far_energy_mse;
    int current_vad;
    int16_t vad_update_count;
    int first_vad;
    int16_t startup_state, mse_channel_count, sup_gain, sup_gain_old;
    int32_t mse_adapt_old, mse_stored_old, mse_threshold;
    int16_t dfa_clean_q, dfa_clean_q_old, dfa_noisy_q, dfa_noisy_q_old;
    uint32_t tot_count;
    uint32_t seed;

    /* Not state: how often a block took the data-dependent paths the HIP kernel has short forms for (aecm_oracle_get_stats). */
    uint64_t stats[ORC_STATS];
};

/* ------------------------------------------------------------------------------------------ */
/* Fixed-point primitives (reference aecm/spl_inl.h, aecm/signal_processing_library.{h,cc})     */
/* ------------------------------------------------------------------------------------------ */

static int clz32(uint32_t n) { return n == 0 ? 32 : __builtin_clz(n); }     /* spl_inl.h:41-48 */
static int norm_w32(int32_t a) { return a == 0 ? 0 : clz32((uint32_t)(a < 0 ? ~a : a)) - 1; } /* :97 */
static int norm_u32(uint32_t a) { return a == 0 ? 0 : clz32(a); }            /* spl_inl.h:103 */
static int norm_w16(int16_t a) {                                             /* spl_inl.h:109 */
    int32_t a32 = a;
    return a == 0 ? 0 : clz32((uint32_t)(a < 0 ? ~a32 : a32)) - 17;
}
static int32_t add_sat_w32(int32_t a, int32_t b) {                           /* spl_inl.h:70-81 */
    int32_t sum = (int32_t)((uint32_t)a + (uint32_t)b);
    if ((a < 0) == (b < 0) && (a < 0) != (sum < 0)) return sum < 0 ? INT32_MAX : INT32_MIN;
    return sum;
}
static int16_t sat_w16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); } /* :59 */
static int32_t abs_w16(int16_t a) { return a >= 0 ? a : -(int32_t)a; }       /* spl.h:99 (int result) */
static int32_t div_w32_w16(int32_t num, int16_t den) {                      /* spl.cc:116-123 */
    return den != 0 ? num / den : (int32_t)0x7FFFFFFF;
}
static uint32_t div_u32_u16(uint32_t num, uint16_t den) {                   /* spl.cc:107-114 */
    return den != 0 ? num / den : 0xFFFFFFFFu;
}
/* WEBRTC_SPL_SHIFT_W32 (spl.h:128) on a signed operand: left = multiply, right = arithmetic. */
static int32_t shift_w32(int32_t x, int c) {
    return c >= 0 ? (int32_t)((uint32_t)x << (c & 31)) : x >> ((-c) & 31);
}
/* ...and on an unsigned operand: left = wrapping multiply, right = logical (Appendix A T12/T21). */
static uint32_t shift_u32(uint32_t x, int c) { return c >= 0 ? x << (c & 31) : x >> ((-c) & 31); }

/* WebRtcSpl_SqrtFloor (spl.cc:76-105): successive approximation, delta = 2^15 .. 2^0. */
int32_t aecm_oracle_sqrt_floor(int32_t value) {
    int32_t root = 0;
    for (int n = 15; n >= 0; --n) {
        int32_t try1 = root + (1 << n);
        if (value >= (int32_t)((uint32_t)try1 << n)) {
            value -= (int32_t)((uint32_t)try1 << n);
            root |= 2 << n;
        }
    }
    return root >> 1;
}

static int max_abs_w16(const int16_t *v, int n) {                            /* spl.cc:154-174 */
    int maximum = 0;
    for (int i = 0; i < n; ++i) {
        int a = v[i] < 0 ? -(int)v[i] : (int)v[i];
        if (a > maximum) maximum = a;
    }
    return maximum > 32767 ? 32767 : maximum;
}

/* ------------------------------------------------------------------------------------------ */
/* 128-point complex radix-2 transforms, "mode 1" (reference aecm/complex_fft.c:181-491)        */
/* ------------------------------------------------------------------------------------------ */

/* Pairs (i, bitrev7(i)) with i < bitrev7(i): exactly the 56 swaps of the reference's index_7 table
 * (complex_fft.c:152-160), computed once instead of stored. */
static unsigned char g_swap_a[56], g_swap_b[56];
static int g_swap_ready = 0;
static void init_swaps(void) {
    int n = 0;
    for (unsigned i = 0; i < 128; ++i) {
        unsigned r = 0;
        for (int b = 0; b < 7; ++b) r |= ((i >> b) & 1u) << (6 - b);
        if (i < r) { g_swap_a[n] = (unsigned char)i; g_swap_b[n] = (unsigned char)r; ++n; }
    }
    g_swap_ready = 1;
}

/* In place.  Forward (complex_fft.c:241-359): every stage halves the data (net 1/128).
 * Inverse (complex_fft.c:361-491): every stage first looks at max|x| over all 256 int16 values and
 * shifts by 0, 1 or 2; the sum of the shifts is returned through scale_out.
 * The bit-reversal (complex_fft.c:181-209, table index_7 = all pairs (i, bitrev7(i)), i < rev) is
 * part of WebRtcSpl_RealForwardFFT/RealInverseFFT (real_fft.c:67,94) and is done here as well. */
void aecm_oracle_fft128(int16_t re[128], int16_t im[128], int inverse, int *scale_out) {
    if (!g_swap_ready) init_swaps();
    for (int n = 0; n < 56; ++n) {
        const unsigned i = g_swap_a[n], r = g_swap_b[n];
        int16_t t = re[i]; re[i] = re[r]; re[r] = t;
        t = im[i]; im[i] = im[r]; im[r] = t;
    }
    int scale = 0;
    for (int stage = 0; stage < 7; ++stage) {
        const int l = 1 << stage;          /* half-span of this stage's butterflies          */
        const int k = 9 - stage;           /* twiddle stride exponent: j = m << k  (:253,296) */
        int shift = 0;
        int32_t round2 = 8192;
        if (inverse) {                     /* complex_fft.c:382-396 */
            int m = 0;
            for (int i = 0; i < 128; ++i) {
                int a = re[i] < 0 ? -(int)re[i] : (int)re[i];
                int b = im[i] < 0 ? -(int)im[i] : (int)im[i];
                if (a > m) m = a;
                if (b > m) m = b;
            }
            if (m > 32767) m = 32767;
            if (m > 13573) { shift++; scale++; round2 <<= 1; }
            if (m > 27146) { shift++; scale++; round2 <<= 1; }
        }
        for (int m = 0; m < l; ++m) {
            const int r = (m << k) >> 3;   /* all indices used are multiples of 8 */
            const int32_t wr = kOrcTwiddleCosQ15[r];
            const int32_t wi = inverse ? kOrcTwiddleSinQ15[r] : -(int32_t)kOrcTwiddleSinQ15[r];
            for (int i = m; i < 128; i += 2 * l) {
                const int j = i + l;
                /* complex_fft.c:332-350 (forward) / :465-482 (inverse) */
                int32_t tr = (wr * re[j] - wi * im[j] + 1) >> 1;
                int32_t ti = (wr * im[j] + wi * re[j] + 1) >> 1;
                int32_t qr = (int32_t)re[i] * 16384;
                int32_t qi = (int32_t)im[i] * 16384;
                if (!inverse) {
                    re[j] = (int16_t)((qr - tr + 16384) >> 15);
                    im[j] = (int16_t)((qi - ti + 16384) >> 15);
                    re[i] = (int16_t)((qr + tr + 16384) >> 15);
                    im[i] = (int16_t)((qi + ti + 16384) >> 15);
                } else {
                    re[j] = (int16_t)((qr - tr + round2) >> (shift + 14));
                    im[j] = (int16_t)((qi - ti + round2) >> (shift + 14));
                    re[i] = (int16_t)((qr + tr + round2) >> (shift + 14));
                    im[i] = (int16_t)((qi + ti + round2) >> (shift + 14));
                }
            }
        }
    }
    if (scale_out) *scale_out = scale;
}

/* ------------------------------------------------------------------------------------------ */
/* TimeToFrequencyDomain + WindowAndFFT (reference aecm/aecm_core_c.cc:166-191, 261-365)        */
/* ------------------------------------------------------------------------------------------ */

/* x = 128 samples (old half then new half).  Outputs the 65-bin spectrum (bins 0 and 64 purely
 * real), |X| per bin, sum of |X|; returns the dynamic Q (0..14). */
static int time_to_frequency(const int16_t x[128], int16_t fr[ORC_BINS], int16_t fi[ORC_BINS],
                             uint16_t mag[ORC_BINS], uint32_t *mag_sum) {
    int16_t re[128], im[128];
    const int q = norm_w16((int16_t)max_abs_w16(x, 128));        /* aecm_core_c.cc:288-289 */
    for (int i = 0; i < ORC_BLOCK; ++i) {                         /* aecm_core_c.cc:174-182 */
        int16_t s = (int16_t)(x[i] * (1 << q));
        re[i] = (int16_t)((s * kOrcSqrtHanningQ14[i]) >> 14);
        s = (int16_t)(x[i + ORC_BLOCK] * (1 << q));
        re[ORC_BLOCK + i] = (int16_t)((s * kOrcSqrtHanningQ14[ORC_BLOCK - i]) >> 14);
    }
    memset(im, 0, sizeof im);                                     /* real_fft.c:59-65 */
    aecm_oracle_fft128(re, im, 0, NULL);
    for (int i = 0; i < ORC_BLOCK; ++i) {                         /* aecm_core_c.cc:188-190 */
        fr[i] = re[i];
        fi[i] = (int16_t)(-im[i]);
    }
    fr[ORC_BLOCK] = re[ORC_BLOCK];
    fi[0] = 0;                                                    /* aecm_core_c.cc:296-297 */
    fi[ORC_BLOCK] = 0;
    mag[0] = (uint16_t)abs_w16(fr[0]);                            /* :298-302 */
    mag[ORC_BLOCK] = (uint16_t)abs_w16(fr[ORC_BLOCK]);
    uint32_t sum = (uint32_t)mag[0] + (uint32_t)mag[ORC_BLOCK];
    for (int i = 1; i < ORC_BLOCK; ++i) {                         /* :304-362 (no ABS_APPROX) */
        if (fr[i] == 0) {
            mag[i] = (uint16_t)abs_w16(fi[i]);
        } else if (fi[i] == 0) {
            mag[i] = (uint16_t)abs_w16(fr[i]);
        } else {
            int16_t a = (int16_t)abs_w16(fr[i]);
            int16_t b = (int16_t)abs_w16(fi[i]);
            int32_t s = add_sat_w32(a * a, b * b);
            mag[i] = (uint16_t)aecm_oracle_sqrt_floor(s);
        }
        sum += (uint32_t)mag[i];
    }
    *mag_sum = sum;
    return q;
}

/* ------------------------------------------------------------------------------------------ */
/* Delay estimator, integer half only (the float histogram is output-dead: SURVEY.md section 0) */
/* ------------------------------------------------------------------------------------------ */

static void mean_estimator_fix(int32_t new_value, int factor, int32_t *mean) { /* delay_estimator.cc:690 */
    int32_t diff = new_value - *mean;
    if (diff < 0) diff = -((-diff) >> factor);
    else diff = diff >> factor;
    *mean += diff;
}

/* BinarySpectrumFix (delay_estimator_wrapper.cc:92-125). */
static uint32_t binary_spectrum(const uint16_t *spectrum, int32_t *threshold, int q, int *initialized) {
    uint32_t out = 0;
    if (!*initialized) {
        for (int i = BAND_FIRST; i <= BAND_LAST; ++i) {
            if (spectrum[i] > 0) {
                int32_t s15 = ((int32_t)spectrum[i]) << (15 - q);
                threshold[i] = s15 >> 1;
                *initialized = 1;
            }
        }
    }
    for (int i = BAND_FIRST; i <= BAND_LAST; ++i) {
        int32_t s15 = ((int32_t)spectrum[i]) << (15 - q);
        mean_estimator_fix(s15, 6, &threshold[i]);
        if (s15 > threshold[i]) out |= 1u << (i - BAND_FIRST);
    }
    return out;
}

static int popcount32(uint32_t v) { return __builtin_popcount(v); }          /* delay_estimator.cc:44 */

/* WebRtc_AddBinaryFarSpectrum (delay_estimator.cc:369-382). */
static void add_binary_far(AecmOracle *o, uint32_t word) {
    memmove(&o->bin_far_hist[1], &o->bin_far_hist[0], (ORC_HISTORY - 1) * sizeof(uint32_t));
    o->bin_far_hist[0] = word;
    memmove(&o->far_bit_counts[1], &o->far_bit_counts[0], (ORC_HISTORY - 1) * sizeof(int));
    o->far_bit_counts[0] = popcount32(word);
}

/* WebRtc_ProcessBinarySpectrum (delay_estimator.cc:521-664), lookahead 0, robust validation off
 * (aecm_core.cc:218,225). */
static int process_binary(AecmOracle *o, uint32_t near_word) {
    int candidate = -1;
    int32_t best = MAX_BIT_COUNTS_Q9, worst = 0;
    int any_far = 0;
    for (int i = 0; i < ORC_HISTORY; ++i) {
        int32_t bc = popcount32(near_word ^ o->bin_far_hist[i]) << 9;       /* :546,553 */
        if (o->far_bit_counts[i] > 0) {                                     /* :558-563 */
            int shifts = 13 - ((3 * o->far_bit_counts[i]) >> 4);
            mean_estimator_fix(bc, shifts, &o->mean_bit_counts[i]);
            any_far = 1;                                                    /* :623-626 */
        }
    }
    for (int i = 0; i < ORC_HISTORY; ++i) {                                 /* :568-576 */
        if (o->mean_bit_counts[i] < best) { best = o->mean_bit_counts[i]; candidate = i; }
        if (o->mean_bit_counts[i] > worst) worst = o->mean_bit_counts[i];
    }
    int32_t valley = worst - best;
    if (o->minimum_probability > PROB_LOWER_LIMIT && valley > PROB_MIN_SPREAD) { /* :593-606 */
        int32_t thr = best + PROB_OFFSET;
        if (thr < PROB_LOWER_LIMIT) thr = PROB_LOWER_LIMIT;
        if (o->minimum_probability > thr) o->minimum_probability = thr;
    }
    o->last_delay_probability++;                                            /* :609 */
    int valid = (valley > PROB_OFFSET) &&                                   /* :618-620 */
                ((best < o->minimum_probability) || (best < o->last_delay_probability));
    if (any_far && valid) {                                                 /* :643-661 */
        o->last_delay = candidate;
        if (best < o->last_delay_probability) o->last_delay_probability = best;
    }
    return o->last_delay;
}

/* ------------------------------------------------------------------------------------------ */
/* Energies, VAD, step size, channel update, suppression gain (reference aecm/aecm_core.cc)      */
/* ------------------------------------------------------------------------------------------ */

static int16_t asym_filt(int16_t old, int16_t in, int16_t step_pos, int16_t step_neg) { /* :588-605 */
    if ((old == 32767) | (old == -32768)) return in;
    int16_t r = old;
    if (old > in) r = (int16_t)(r - ((old - in) >> step_neg));
    else r = (int16_t)(r + ((in - old) >> step_pos));
    return r;
}

static int16_t log_energy_q8(uint32_t energy, int q) {                      /* :612-628 */
    int16_t v = 7 << 7;
    if (energy > 0) {
        int zeros = norm_u32(energy);
        int16_t frac = (int16_t)(((energy << zeros) & 0x7FFFFFFFu) >> 23);
        v = (int16_t)(v + (((31 - zeros) << 8) + frac - (q << 8)));
    }
    return v;
}

/* WebRtcAecm_CalcEnergies (aecm_core.cc:644-755) incl. CalcLinearEnergiesC (:267-284). */
static void calc_energies(AecmOracle *o, const uint16_t *far_spec, int16_t far_q, uint32_t near_energy,
                          int32_t *echo_est) {
    uint32_t e_adapt = 0, e_stored = 0, e_far = 0;
    int16_t inc_max = 4, dec_max = 11, inc_min = 11, dec_min = 3;

    memmove(o->near_log + 1, o->near_log, sizeof(int16_t) * (ORC_LOGBUF - 1));
    o->near_log[0] = log_energy_q8(near_energy, o->dfa_noisy_q);

    for (int i = 0; i < ORC_BINS; ++i) {
        echo_est[i] = (int32_t)o->ch_stored[i] * (int32_t)far_spec[i];
        e_far += (uint32_t)far_spec[i];
        e_adapt += (uint32_t)((int32_t)o->ch_adapt16[i] * (int32_t)far_spec[i]);
        e_stored += (uint32_t)echo_est[i];
    }

    memmove(o->echo_adapt_log + 1, o->echo_adapt_log, sizeof(int16_t) * (ORC_LOGBUF - 1));
    memmove(o->echo_stored_log + 1, o->echo_stored_log, sizeof(int16_t) * (ORC_LOGBUF - 1));

    o->far_log = log_energy_q8(e_far, far_q);
    o->echo_adapt_log[0] = log_energy_q8(e_adapt, RESOLUTION_CHANNEL16 + far_q);
    o->echo_stored_log[0] = log_energy_q8(e_stored, RESOLUTION_CHANNEL16 + far_q);

    if (o->far_log > FAR_ENERGY_MIN) {                                      /* :692-730 */
        if (o->startup_state == 0) { inc_max = 2; dec_min = 2; inc_min = 8; }
        o->far_energy_min = asym_filt(o->far_energy_min, o->far_log, inc_min, dec_min);
        o->far_energy_max = asym_filt(o->far_energy_max, o->far_log, inc_max, dec_max);
        o->far_energy_maxmin = (int16_t)(o->far_energy_max - o->far_energy_min);

        int16_t t16 = (int16_t)(2560 - o->far_energy_min);
        if (t16 > 0) t16 = (int16_t)((t16 * FAR_ENERGY_VAD_REGION) >> 9);
        else t16 = 0;
        t16 = (int16_t)(t16 + FAR_ENERGY_VAD_REGION);

        if ((o->startup_state == 0) | (o->vad_update_count > 1024)) {
            o->far_energy_vad = (int16_t)(o->far_energy_min + t16);
        } else {
            if (o->far_energy_vad > o->far_log) {
                o->far_energy_vad =
                    (int16_t)(o->far_energy_vad + ((o->far_log + t16 - o->far_energy_vad) >> 6));
                o->vad_update_count = 0;
            } else {
                o->vad_update_count++;
            }
        }
        o->far_energy_mse = (int16_t)(o->far_energy_vad + (1 << 8));
    }

    if (o->far_log > o->far_energy_vad) {                                   /* :733-740 */
        if ((o->startup_state == 0) | (o->far_energy_maxmin > FAR_ENERGY_DIFF)) o->current_vad = 1;
    } else {
        o->current_vad = 0;
    }
    if (o->current_vad && o->first_vad) {                                   /* :741-754 */
        o->first_vad = 0;
        if (o->echo_adapt_log[0] > o->near_log[0]) {
            for (int i = 0; i < ORC_BINS; ++i) o->ch_adapt16[i] >>= 3;
            o->echo_adapt_log[0] = (int16_t)(o->echo_adapt_log[0] - (3 << 8));
            o->first_vad = 1;
        }
    }
}

/* WebRtcAecm_CalcStepSize (aecm_core.cc:767-794). */
static int16_t calc_step_size(const AecmOracle *o) {
    int16_t mu = MU_MAX;
    if (!o->current_vad) {
        mu = 0;
    } else if (o->startup_state > 0) {
        if (o->far_energy_min >= o->far_energy_max) {
            mu = MU_MIN;
        } else {
            int16_t t16 = (int16_t)(o->far_log - o->far_energy_min);
            int32_t t32 = t16 * MU_DIFF;
            t32 = div_w32_w16(t32, o->far_energy_maxmin);
            mu = (int16_t)(MU_MIN - 1 - (int16_t)t32);
        }
        if (mu < MU_MAX) mu = MU_MAX;
    }
    return mu;
}

/* StoreAdaptiveChannelC (aecm_core.cc:286-306). */
static void store_adaptive_channel(AecmOracle *o, const uint16_t *far_spec, int32_t *echo_est) {
    memcpy(o->ch_stored, o->ch_adapt16, sizeof o->ch_stored);
    for (int i = 0; i < ORC_BINS; ++i) echo_est[i] = (int32_t)o->ch_stored[i] * (int32_t)far_spec[i];
}

/* ResetAdaptiveChannelC (aecm_core.cc:308-323). */
static void reset_adaptive_channel(AecmOracle *o) {
    memcpy(o->ch_adapt16, o->ch_stored, sizeof o->ch_adapt16);
    for (int i = 0; i < ORC_BINS; ++i) o->ch_adapt32[i] = (int32_t)((uint32_t)(int32_t)o->ch_stored[i] << 16);
}

/* WebRtcAecm_UpdateChannel (aecm_core.cc:810-986). */
static void update_channel(AecmOracle *o, const uint16_t *far_spec, int16_t far_q, const uint16_t *dfa,
                           int16_t mu, int32_t *echo_est) {
    if (mu) {
        for (int i = 0; i < ORC_BINS; ++i) {
            uint32_t u1, u2;
            int32_t t1, t2;
            int16_t zeros_ch = (int16_t)norm_u32((uint32_t)o->ch_adapt32[i]);
            int16_t zeros_far = (int16_t)norm_u32((uint32_t)far_spec[i]);
            int16_t shift_ch_far, zeros_num, zeros_dfa, xfa_q, dfa_q, t16;
            if (zeros_ch + zeros_far > 31) {                                 /* :836-850 */
                u1 = (uint32_t)o->ch_adapt32[i] * (uint32_t)far_spec[i];
                shift_ch_far = 0;
            } else {
                shift_ch_far = (int16_t)(32 - zeros_ch - zeros_far);
                u1 = (uint32_t)(shift_ch_far >= 32 ? 0 : o->ch_adapt32[i] >> shift_ch_far) *
                     (uint32_t)far_spec[i];
            }
            zeros_num = (int16_t)norm_u32(u1);                               /* :852-867 */
            zeros_dfa = dfa[i] ? (int16_t)norm_u32((uint32_t)dfa[i]) : 32;
            t16 = (int16_t)(zeros_dfa - 2 + o->dfa_noisy_q - RESOLUTION_CHANNEL32 - far_q + shift_ch_far);
            if (zeros_num > t16 + 1) {
                xfa_q = t16;
                dfa_q = (int16_t)(zeros_dfa - 2);
            } else {
                xfa_q = (int16_t)(zeros_num - 2);
                dfa_q = (int16_t)(RESOLUTION_CHANNEL32 + far_q - o->dfa_noisy_q - shift_ch_far + xfa_q);
            }
            u1 = shift_u32(u1, xfa_q);                                       /* :869-872 */
            u2 = shift_u32((uint32_t)dfa[i], dfa_q);
            t1 = (int32_t)u2 - (int32_t)u1;
            zeros_num = (int16_t)norm_w32(t1);
            if (t1 && (far_spec[i] > (CHANNEL_VAD << far_q))) {              /* :873-920 */
                int16_t shift_num, shift2res;
                if (zeros_num + zeros_far > 31) {
                    if (t1 > 0) t2 = (int32_t)((uint32_t)t1 * (uint32_t)far_spec[i]);
                    else t2 = -(int32_t)((uint32_t)(-t1) * (uint32_t)far_spec[i]);
                    shift_num = 0;
                } else {
                    shift_num = (int16_t)(32 - (zeros_num + zeros_far));
                    if (t1 > 0) t2 = (t1 >> shift_num) * (int32_t)far_spec[i];
                    else t2 = -((-t1 >> shift_num) * (int32_t)far_spec[i]);
                }
                t2 = div_w32_w16(t2, (int16_t)(i + 1));
                shift2res = (int16_t)(shift_num + shift_ch_far - xfa_q - mu - ((30 - zeros_far) << 1));
                if (norm_w32(t2) < shift2res) t2 = INT32_MAX;
                else t2 = shift_w32(t2, shift2res);
                o->ch_adapt32[i] = add_sat_w32(o->ch_adapt32[i], t2);
                if (o->ch_adapt32[i] < 0) o->ch_adapt32[i] = 0;
                o->ch_adapt16[i] = (int16_t)(o->ch_adapt32[i] >> 16);
            }
        }
    }

    if ((o->startup_state == 0) & (o->current_vad)) {                        /* :926-929 */
        store_adaptive_channel(o, far_spec, echo_est);
    } else {
        if (o->far_log < o->far_energy_mse) o->mse_channel_count = 0;        /* :931-935 */
        else o->mse_channel_count++;
        if (o->mse_channel_count >= (MIN_MSE_COUNT + 10)) {                  /* :937-983 */
            int32_t mse_stored = 0, mse_adapt = 0;
            for (int i = 0; i < MIN_MSE_COUNT; ++i) {
                int32_t d = (int32_t)o->echo_stored_log[i] - (int32_t)o->near_log[i];
                mse_stored += d >= 0 ? d : -d;
                d = (int32_t)o->echo_adapt_log[i] - (int32_t)o->near_log[i];
                mse_adapt += d >= 0 ? d : -d;
            }
            if (((mse_stored << MSE_RESOLUTION) < (MIN_MSE_DIFF * mse_adapt)) &
                ((o->mse_stored_old << MSE_RESOLUTION) < (MIN_MSE_DIFF * o->mse_adapt_old))) {
                reset_adaptive_channel(o);
            } else if (((MIN_MSE_DIFF * mse_stored) > (mse_adapt << MSE_RESOLUTION)) &
                       (mse_adapt < o->mse_threshold) & (o->mse_adapt_old < o->mse_threshold)) {
                store_adaptive_channel(o, far_spec, echo_est);
                if (o->mse_threshold == INT32_MAX) {
                    o->mse_threshold = mse_adapt + o->mse_adapt_old;
                } else {
                    int scaled = o->mse_threshold * 5 / 8;
                    o->mse_threshold += ((mse_adapt - scaled) * 205) >> 8;
                }
            }
            o->mse_channel_count = 0;
            o->mse_stored_old = mse_stored;
            o->mse_adapt_old = mse_adapt;
        }
    }
}

/* WebRtcAecm_CalcSuppressionGain (aecm_core.cc:1000-1052). */
static int16_t calc_suppression_gain(AecmOracle *o) {
    int16_t sup = SUPGAIN_DEFAULT;
    int16_t t16;
    if (!o->current_vad) {
        sup = 0;
    } else {
        t16 = (int16_t)(o->near_log[0] - o->echo_stored_log[0] - 0 /* ENERGY_DEV_OFFSET */);
        int16_t dE = (int16_t)abs_w16(t16);
        if (dE < ENERGY_DEV_TOL) {
            if (dE < SUPGAIN_EPC_DT) {
                int32_t t32 = o->sg_err_diff_ab * dE;
                t32 += (SUPGAIN_EPC_DT >> 1);
                t16 = (int16_t)div_w32_w16(t32, SUPGAIN_EPC_DT);
                sup = (int16_t)(o->sg_err_a - t16);
            } else {
                int32_t t32 = o->sg_err_diff_bd * (ENERGY_DEV_TOL - dE);
                t32 += ((ENERGY_DEV_TOL - SUPGAIN_EPC_DT) >> 1);
                t16 = (int16_t)div_w32_w16(t32, (ENERGY_DEV_TOL - SUPGAIN_EPC_DT));
                sup = (int16_t)(o->sg_err_d + t16);
            }
        } else {
            sup = o->sg_err_d;
        }
    }
    t16 = sup > o->sup_gain_old ? sup : o->sup_gain_old;
    o->sup_gain_old = sup;
    o->sup_gain = (int16_t)(o->sup_gain + (int16_t)((t16 - o->sup_gain) >> 4));
    return o->sup_gain;
}

/* ------------------------------------------------------------------------------------------ */
/* Comfort noise (reference aecm/aecm_core_c.cc:52-164) + LCG (spl.cc:129-147)                  */
/* ------------------------------------------------------------------------------------------ */

static void comfort_noise(AecmOracle *o, const uint16_t *dfa, int16_t *er, int16_t *ei, const int16_t *lambda) {
    int16_t rnd[ORC_BLOCK], u_re[ORC_BINS], u_im[ORC_BINS], noise16[ORC_BINS];
    const int16_t shift_near_to_noise = (int16_t)(15 - o->dfa_clean_q);     /* :49,66 */
    int16_t min_track;
    if (o->noise_est_ctr < 100) { o->noise_est_ctr++; min_track = 6; }      /* :72-78 */
    else min_track = 9;

    for (int i = 0; i < ORC_BINS; ++i) {                                    /* :81-127 */
        int32_t in = (int32_t)dfa[i] << shift_near_to_noise;
        if (in < o->noise_est[i]) {
            o->noise_low_ctr[i] = 0;
            if (o->noise_est[i] < (1 << min_track)) {
                o->noise_high_ctr[i]++;
                if (o->noise_high_ctr[i] >= 5) { o->noise_est[i]--; o->noise_high_ctr[i] = 0; }
            } else {
                o->noise_est[i] -= (o->noise_est[i] - in) >> min_track;
            }
        } else {
            o->noise_high_ctr[i] = 0;
            if ((o->noise_est[i] >> 19) > 0) {
                o->noise_est[i] >>= 11;
                o->noise_est[i] *= 2049;
            } else if ((o->noise_est[i] >> 11) > 0) {
                o->noise_est[i] *= 2049;
                o->noise_est[i] >>= 11;
            } else {
                o->noise_low_ctr[i]++;
                if (o->noise_low_ctr[i] >= 5) {
                    o->noise_est[i] += (o->noise_est[i] >> 9) + 1;
                    o->noise_low_ctr[i] = 0;
                }
            }
        }
    }
    for (int i = 0; i < ORC_BINS; ++i) {                                    /* :129-140 */
        int32_t t32 = o->noise_est[i] >> shift_near_to_noise;
        if (t32 > 32767) {
            t32 = 32767;
            o->noise_est[i] = t32 << shift_near_to_noise;
        }
        noise16[i] = (int16_t)t32;
        int16_t t16 = (int16_t)(ONE_Q14 - lambda[i]);
        noise16[i] = (int16_t)((t16 * noise16[i]) >> 14);
    }
    for (int i = 0; i < ORC_BLOCK; ++i) {                                   /* spl.cc:129-147 */
        o->seed = (o->seed * 69069u + 1u) & 0x7FFFFFFFu;
        rnd[i] = (int16_t)(o->seed >> 16);
    }
    u_re[0] = 0;                                                            /* :146-158 */
    u_im[0] = 0;
    for (int i = 1; i < ORC_BINS; ++i) {
        int16_t idx = (int16_t)((359 * rnd[i - 1]) >> 15);
        u_re[i] = (int16_t)((noise16[i] * kOrcCosQ13[idx]) >> 13);
        u_im[i] = (int16_t)((-noise16[i] * kOrcSinQ13[idx]) >> 13);
    }
    u_im[ORC_BLOCK] = 0;
    for (int i = 0; i < ORC_BINS; ++i) {                                    /* :160-163 */
        er[i] = sat_w16((int32_t)er[i] + u_re[i]);
        ei[i] = sat_w16((int32_t)ei[i] + u_im[i]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* The block (reference aecm/aecm_core_c.cc:368-711)                                            */
/* ------------------------------------------------------------------------------------------ */

int aecm_oracle_process_block(AecmOracle *o, const int16_t far_blk[ORC_BLOCK],
                              const int16_t near_noisy[ORC_BLOCK], const int16_t *near_clean,
                              int16_t out[ORC_BLOCK]) {
    int16_t xbuf[128], dbuf[128], cbuf[128];
    int16_t far_re[ORC_BINS], far_im[ORC_BINS], dfw_re[ORC_BINS], dfw_im[ORC_BINS];
    int16_t efw_re[ORC_BINS], efw_im[ORC_BINS];
    uint16_t xfa[ORC_BINS], dfa_noisy[ORC_BINS], dfa_clean_buf[ORC_BINS];
    const uint16_t *dfa_clean = dfa_clean_buf;
    uint32_t xfa_sum, dfa_noisy_sum, dfa_clean_sum;
    int32_t echo_est[ORC_BINS];
    int16_t hnl[ORC_BINS];
    int16_t num_pos_coef = 0;

    if (o->startup_state < 2)                                                /* :420-424 */
        o->startup_state = (int16_t)((o->tot_count >= CONV_LEN) + (o->tot_count >= CONV_LEN2));

    memcpy(xbuf, o->x_old, sizeof o->x_old);                                 /* :428-436 */
    memcpy(xbuf + ORC_BLOCK, far_blk, sizeof(int16_t) * ORC_BLOCK);
    memcpy(dbuf, o->d_old, sizeof o->d_old);
    memcpy(dbuf + ORC_BLOCK, near_noisy, sizeof(int16_t) * ORC_BLOCK);
    if (near_clean) {
        memcpy(cbuf, o->c_old, sizeof o->c_old);
        memcpy(cbuf + ORC_BLOCK, near_clean, sizeof(int16_t) * ORC_BLOCK);
    }

    int far_q = time_to_frequency(xbuf, far_re, far_im, xfa, &xfa_sum);      /* :439 */
    (void)xfa_sum;
    int zeros_dbuf_noisy = time_to_frequency(dbuf, dfw_re, dfw_im, dfa_noisy, &dfa_noisy_sum); /* :442 */
    o->dfa_noisy_q_old = o->dfa_noisy_q;
    o->dfa_noisy_q = (int16_t)zeros_dbuf_noisy;
    if (!near_clean) {                                                       /* :449-464 */
        dfa_clean = dfa_noisy;
        o->dfa_clean_q_old = o->dfa_noisy_q_old;
        o->dfa_clean_q = o->dfa_noisy_q;
        dfa_clean_sum = dfa_noisy_sum;
    } else {
        int zc = time_to_frequency(cbuf, dfw_re, dfw_im, dfa_clean_buf, &dfa_clean_sum);
        o->dfa_clean_q_old = o->dfa_clean_q;
        o->dfa_clean_q = (int16_t)zc;
    }
    (void)dfa_clean_sum;

    /* WebRtcAecm_UpdateFarHistory (aecm_core.cc:125-138) */
    o->far_hist_pos++;
    if (o->far_hist_pos >= ORC_HISTORY) o->far_hist_pos = 0;
    o->far_hist_q[o->far_hist_pos] = far_q;
    memcpy(o->far_hist[o->far_hist_pos], xfa, sizeof xfa);

    /* WebRtc_AddFarSpectrumFix (delay_estimator_wrapper.cc:233-263) */
    add_binary_far(o, binary_spectrum(xfa, o->mean_far, far_q, &o->far_init));
    /* WebRtc_DelayEstimatorProcessFix (delay_estimator_wrapper.cc:447-476) */
    int delay = process_binary(o, binary_spectrum(dfa_noisy, o->mean_near, zeros_dbuf_noisy, &o->near_init));
    if (delay == -2) delay = 0;                                              /* :479-483 */
    if (o->fixed_delay >= 0) delay = o->fixed_delay;                         /* :485-488 */

    /* WebRtcAecm_AlignedFarend (aecm_core.cc:157-172) */
    int pos = o->far_hist_pos - delay;
    if (pos < 0) pos += ORC_HISTORY;
    far_q = o->far_hist_q[pos];
    const uint16_t *far_spec = o->far_hist[pos];
    const int16_t zeros_xbuf = (int16_t)far_q;

    calc_energies(o, far_spec, zeros_xbuf, dfa_noisy_sum, echo_est);         /* :498 */
    const int16_t mu = calc_step_size(o);                                    /* :503 */
    o->tot_count++;                                                          /* :506 */
    update_channel(o, far_spec, zeros_xbuf, dfa_noisy, mu, echo_est);        /* :511 */
    const int16_t sup_gain = calc_suppression_gain(o);                       /* :514 */
    o->stats[ORC_STAT_BLOCKS]++;
    o->stats[ORC_STAT_NLMS] += mu != 0;                                      /* aecm_core.cc:830 */
    o->stats[ORC_STAT_GAIN_ZERO] += sup_gain == 0;
    o->stats[ORC_STAT_Q_STEADY] += o->dfa_clean_q <= o->dfa_clean_q_old;
    o->stats[ORC_STAT_DELAYED] += delay != 0;
    o->stats[ORC_STAT_VAD] += o->current_vad != 0;

    for (int i = 0; i < ORC_BINS; ++i) {                                     /* :517-615 */
        uint32_t gained;
        int16_t res_diff, t16a, t16b, q_diff;
        int32_t t32 = echo_est[i] - o->echo_filt[i];
        o->echo_filt[i] += (int32_t)(((int64_t)t32 * 50) >> 8);

        int16_t zeros32 = (int16_t)(norm_w32(o->echo_filt[i]) + 1);
        int16_t zeros16 = (int16_t)(norm_w16(sup_gain) + 1);
        if (zeros32 + zeros16 > 16) {
            gained = (uint32_t)o->echo_filt[i] * (uint32_t)(uint16_t)sup_gain;
            res_diff = 14 - RESOLUTION_CHANNEL16 - RESOLUTION_SUPGAIN;
            res_diff = (int16_t)(res_diff + (o->dfa_clean_q - zeros_xbuf));
        } else {
            t16a = (int16_t)(17 - zeros32 - zeros16);
            res_diff = (int16_t)(14 + t16a - RESOLUTION_CHANNEL16 - RESOLUTION_SUPGAIN);
            res_diff = (int16_t)(res_diff + (o->dfa_clean_q - zeros_xbuf));
            if (zeros32 > t16a) {
                gained = (uint32_t)o->echo_filt[i] * (uint32_t)(uint16_t)(sup_gain >> t16a);
            } else {
                gained = (uint32_t)((o->echo_filt[i] >> t16a) * (int32_t)sup_gain);
            }
        }

        zeros16 = (int16_t)norm_w16(o->near_filt[i]);                        /* :552-579 */
        int16_t dq = (int16_t)(o->dfa_clean_q - o->dfa_clean_q_old);
        if (zeros16 < dq && o->near_filt[i]) {
            t16a = (int16_t)(o->near_filt[i] * (1 << zeros16));
            q_diff = (int16_t)(zeros16 - dq);
            t16b = (int16_t)(dfa_clean[i] >> ((-q_diff) & 31));
        } else {
            t16a = dq < 0 ? (int16_t)(o->near_filt[i] >> ((-dq) & 31))
                          : (int16_t)(o->near_filt[i] * (1 << dq));
            q_diff = 0;
            t16b = (int16_t)dfa_clean[i];
        }
        t32 = (int32_t)(t16b - t16a);
        t16b = (int16_t)(t32 >> 4);
        t16b = (int16_t)(t16b + t16a);
        zeros16 = (int16_t)norm_w16(t16b);
        if ((t16b) & (-q_diff > zeros16)) {                                  /* :572, literally */
            o->near_filt[i] = 32767;
        } else {
            o->near_filt[i] = q_diff < 0 ? (int16_t)(t16b * (1 << -q_diff)) : (int16_t)(t16b >> q_diff);
        }

        if (gained == 0) {                                                   /* :582-611 */
            hnl[i] = ONE_Q14;
        } else if (o->near_filt[i] == 0) {
            hnl[i] = 0;
        } else {
            gained += (uint32_t)(o->near_filt[i] >> 1);
            uint32_t tu = div_u32_u16(gained, (uint16_t)o->near_filt[i]);
            t32 = (int32_t)shift_u32(tu, res_diff);
            if (t32 > ONE_Q14) hnl[i] = 0;
            else if (t32 < 0) hnl[i] = ONE_Q14;
            else {
                hnl[i] = (int16_t)(ONE_Q14 - (int16_t)t32);
                if (hnl[i] < 0) hnl[i] = 0;
            }
        }
        if (hnl[i]) num_pos_coef++;
    }

    if (o->mult == 2) {                                                      /* :618-648 */
        int32_t avg = 0;
        for (int i = 0; i < ORC_BINS; ++i) hnl[i] = (int16_t)((hnl[i] * hnl[i]) >> 14);
        for (int i = 4; i <= 24; ++i) avg += (int32_t)hnl[i];
        avg /= 21;
        for (int i = 24; i < ORC_BINS; ++i)
            if (hnl[i] > (int16_t)avg) hnl[i] = (int16_t)avg;
    }

    if (o->nlp_flag) {                                                       /* :651-686 */
        for (int i = 0; i < ORC_BINS; ++i) {
            if (hnl[i] > NLP_COMP_HIGH) hnl[i] = ONE_Q14;
            else if (hnl[i] < NLP_COMP_LOW) hnl[i] = 0;
            int16_t nlp_gain = num_pos_coef < 3 ? 0 : ONE_Q14;
            if (!((hnl[i] == ONE_Q14) && (nlp_gain == ONE_Q14)))
                hnl[i] = (int16_t)((hnl[i] * nlp_gain) >> 14);
            efw_re[i] = (int16_t)((dfw_re[i] * hnl[i] + 8192) >> 14);
            efw_im[i] = (int16_t)((dfw_im[i] * hnl[i] + 8192) >> 14);
        }
    } else {                                                                 /* :687-700 */
        for (int i = 0; i < ORC_BINS; ++i) {
            efw_re[i] = (int16_t)((dfw_re[i] * hnl[i] + 8192) >> 14);
            efw_im[i] = (int16_t)((dfw_im[i] * hnl[i] + 8192) >> 14);
        }
    }

    if (o->cng_mode == 1) comfort_noise(o, dfa_clean, efw_re, efw_im, hnl);  /* :702-705 */

    /* InverseFFTAndWindow (:193-246) + WebRtcSpl_RealInverseFFT (real_fft.c:74-102) */
    int16_t re[128], im[128];
    for (int i = 0; i <= ORC_BLOCK; ++i) {                                   /* :206-214 */
        re[i] = efw_re[i];
        im[i] = (int16_t)(-efw_im[i]);
    }
    for (int i = ORC_BLOCK + 1; i < 128; ++i) {                              /* real_fft.c:87-91 */
        re[i] = re[128 - i];
        im[i] = (int16_t)(-im[128 - i]);
    }
    int out_cfft = 0;
    aecm_oracle_fft128(re, im, 1, &out_cfft);
    o->stats[ORC_STAT_IFFT_UNSCALED] += out_cfft == 0;                       /* complex_fft.c:382-396: no stage rescaled */
    const int sh = out_cfft - o->dfa_clean_q;
    for (int i = 0; i < ORC_BLOCK; ++i) {                                    /* :218-235 */
        int16_t y = (int16_t)((re[i] * kOrcSqrtHanningQ14[i] + 8192) >> 14);
        int32_t t32 = shift_w32((int32_t)y, sh);
        out[i] = sat_w16(t32 + o->out_ovl[i]);
        t32 = (re[ORC_BLOCK + i] * kOrcSqrtHanningQ14[ORC_BLOCK - i]) >> 14;
        t32 = shift_w32(t32, sh);
        o->out_ovl[i] = sat_w16(t32);
    }
    memcpy(o->x_old, far_blk, sizeof o->x_old);                              /* :239-245 */
    memcpy(o->d_old, near_noisy, sizeof o->d_old);
    if (near_clean) memcpy(o->c_old, near_clean, sizeof o->c_old);
    (void)far_re; (void)far_im;
    return 0;
}

int aecm_oracle_process_stream(AecmOracle *o, const int16_t *far_s, const int16_t *near_s, int16_t *out,
                               size_t n_blocks) {
    for (size_t b = 0; b < n_blocks; ++b) {
        int r = aecm_oracle_process_block(o, far_s + b * ORC_BLOCK, near_s + b * ORC_BLOCK, NULL,
                                          out + b * ORC_BLOCK);
        if (r) return r;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Lifetime / configuration                                                                     */
/* ------------------------------------------------------------------------------------------ */

void aecm_oracle_get_stats(const AecmOracle *o, uint64_t stats[ORC_STATS]) { memcpy(stats, o->stats, sizeof o->stats); }

AecmOracle *aecm_oracle_create(void) { return (AecmOracle *)calloc(1, sizeof(AecmOracle)); }
void aecm_oracle_free(AecmOracle *o) { free(o); }

void aecm_oracle_init_echo_path(AecmOracle *o, const int16_t path[ORC_BINS]) { /* aecm_core.cc:249-265 */
    memcpy(o->ch_stored, path, sizeof o->ch_stored);
    memcpy(o->ch_adapt16, path, sizeof o->ch_adapt16);
    for (int i = 0; i < ORC_BINS; ++i) o->ch_adapt32[i] = (int32_t)((uint32_t)(int32_t)o->ch_adapt16[i] << 16);
    o->mse_adapt_old = 1000;
    o->mse_stored_old = 1000;
    o->mse_threshold = INT32_MAX;
    o->mse_channel_count = 0;
}

void aecm_oracle_get_echo_path(const AecmOracle *o, int16_t path[ORC_BINS]) {  /* echo_control_mobile.cc:530 */
    memcpy(path, o->ch_stored, sizeof o->ch_stored);
}

int aecm_oracle_set_config(AecmOracle *o, int cng_mode, int echo_mode) {      /* echo_control_mobile.cc:410-479 */
    if (cng_mode != 0 && cng_mode != 1) return -1;
    if (echo_mode < 0 || echo_mode > 4) return -1;
    o->cng_mode = cng_mode;
    int a = SUPGAIN_ERROR_PARAM_A, b = SUPGAIN_ERROR_PARAM_B, d = SUPGAIN_ERROR_PARAM_D, g = SUPGAIN_DEFAULT;
    if (echo_mode < 3) { int s = 3 - echo_mode; a >>= s; b >>= s; d >>= s; g >>= s; }
    else if (echo_mode == 4) { a <<= 1; b <<= 1; d <<= 1; g <<= 1; }
    o->sup_gain = (int16_t)g;
    o->sup_gain_old = (int16_t)g;
    o->sg_err_a = (int16_t)a;
    o->sg_err_d = (int16_t)d;
    o->sg_err_diff_ab = (int16_t)(a - b);
    o->sg_err_diff_bd = (int16_t)(b - d);
    return 0;
}

void aecm_oracle_control(AecmOracle *o, int fixed_delay, int nlp_flag) {      /* aecm_core.cc:477-482 */
    o->nlp_flag = (int16_t)nlp_flag;
    o->fixed_delay = (int16_t)fixed_delay;
}

int aecm_oracle_init(AecmOracle *o, int fs) {                                 /* aecm_core.cc:358-473 */
    if (fs != 8000 && fs != 16000) return -1;
    memset(o, 0, sizeof *o);
    o->mult = (int16_t)fs / 8000;
    o->seed = 666;
    o->tot_count = 0;
    /* delay estimator init (delay_estimator.cc:330-334, 483-504; wrapper.cc:208-225, 338-355) */
    for (int i = 0; i < ORC_HISTORY; ++i) o->mean_bit_counts[i] = 20 << 9;
    o->minimum_probability = MAX_BIT_COUNTS_Q9;
    o->last_delay_probability = MAX_BIT_COUNTS_Q9;
    o->last_delay = -2;
    o->far_hist_pos = ORC_HISTORY;
    o->nlp_flag = 1;
    o->fixed_delay = -1;
    aecm_oracle_init_echo_path(o, fs == 8000 ? kOrcChannelStored8k : kOrcChannelStored16k);
    o->noise_est_ctr = 0;
    o->cng_mode = 1;
    {                                                                          /* :427-435 */
        int32_t t32 = ORC_BINS * ORC_BINS;
        int16_t t16 = ORC_BINS;
        int i = 0;
        for (; i < (ORC_BINS >> 1) - 1; ++i) {
            o->noise_est[i] = t32 << 8;
            t16--;
            t32 -= (int32_t)((t16 << 1) + 1);
        }
        for (; i < ORC_BINS; ++i) o->noise_est[i] = t32 << 8;
    }
    o->far_energy_min = 32767;
    o->far_energy_max = -32768;
    o->far_energy_maxmin = 0;
    o->far_energy_vad = FAR_ENERGY_MIN;
    o->far_energy_mse = 0;
    o->current_vad = 0;
    o->vad_update_count = 0;
    o->first_vad = 1;
    o->startup_state = 0;
    return aecm_oracle_set_config(o, 1, 3);  /* echo_control_mobile.cc:183-188 */
}

/* ------------------------------------------------------------------------------------------ */
/* Digest                                                                                       */
/* ------------------------------------------------------------------------------------------ */

static uint32_t fnv_step(uint32_t h, uint32_t w) { return (h ^ w) * 16777619u; }
#define FNV_INIT 2166136261u
#define PACK16(lo, hi) (((uint32_t)(uint16_t)(lo)) | (((uint32_t)(uint16_t)(hi)) << 16))

void aecm_oracle_digest(const AecmOracle *o, uint32_t d[ORC_DIGEST_WORDS]) {
    uint32_t h;
    d[0] = o->tot_count;
    d[1] = o->seed;
    d[2] = PACK16(o->startup_state, o->far_hist_pos);
    d[3] = PACK16(o->dfa_noisy_q, o->dfa_noisy_q_old);
    d[4] = PACK16(o->far_log, o->far_energy_min);
    d[5] = PACK16(o->far_energy_max, o->far_energy_maxmin);
    d[6] = PACK16(o->far_energy_vad, o->far_energy_mse);
    d[7] = PACK16(o->current_vad, o->vad_update_count);
    d[8] = PACK16(o->first_vad, o->mse_channel_count);
    d[9] = (uint32_t)o->mse_adapt_old;
    d[10] = (uint32_t)o->mse_stored_old;
    d[11] = (uint32_t)o->mse_threshold;
    d[12] = PACK16(o->sup_gain, o->sup_gain_old);
    d[13] = (uint32_t)o->last_delay;
    d[14] = (uint32_t)o->minimum_probability;
    d[15] = (uint32_t)o->last_delay_probability;
    h = FNV_INIT; for (int i = 0; i < ORC_BINS; ++i) h = fnv_step(h, PACK16(o->ch_stored[i], o->ch_adapt16[i])); d[16] = h;
    h = FNV_INIT; for (int i = 0; i < ORC_BINS; ++i) h = fnv_step(h, (uint32_t)o->ch_adapt32[i]); d[17] = h;
    h = FNV_INIT; for (int i = 0; i < ORC_BINS; ++i) h = fnv_step(h, (uint32_t)o->echo_filt[i]); d[18] = h;
    h = FNV_INIT; for (int i = 0; i < ORC_BINS; ++i) h = fnv_step(h, (uint32_t)(uint16_t)o->near_filt[i]); d[19] = h;
    h = FNV_INIT;
    for (int i = 0; i < ORC_BINS; ++i) {
        h = fnv_step(h, (uint32_t)o->noise_est[i]);
        h = fnv_step(h, PACK16(o->noise_low_ctr[i], o->noise_high_ctr[i]));
    }
    d[20] = fnv_step(h, (uint32_t)(uint16_t)o->noise_est_ctr);
    h = FNV_INIT;
    for (int i = BAND_FIRST; i <= BAND_LAST; ++i) { h = fnv_step(h, (uint32_t)o->mean_far[i]); h = fnv_step(h, (uint32_t)o->mean_near[i]); }
    for (int i = 0; i < ORC_HISTORY; ++i) { h = fnv_step(h, o->bin_far_hist[i]); h = fnv_step(h, (uint32_t)o->mean_bit_counts[i]); }
    d[21] = fnv_step(h, PACK16(o->far_init, o->near_init));
    h = FNV_INIT;
    /* only entries [0, MIN_MSE_COUNT) of the log-energy histories are ever read (aecm_core.cc:943-952): the rest is dead state */
    for (int i = 0; i < 20; ++i) { h = fnv_step(h, PACK16(o->near_log[i], o->echo_adapt_log[i])); h = fnv_step(h, (uint32_t)(uint16_t)o->echo_stored_log[i]); }
    d[22] = h;
    h = FNV_INIT;
    for (int i = 0; i < ORC_BLOCK; ++i) { h = fnv_step(h, PACK16(o->x_old[i], o->d_old[i])); h = fnv_step(h, (uint32_t)(uint16_t)o->out_ovl[i]); }
    for (int p = 0; p < ORC_HISTORY; ++p) {
        h = fnv_step(h, (uint32_t)o->far_hist_q[p]);
        for (int i = 0; i < ORC_BINS; ++i) h = fnv_step(h, (uint32_t)o->far_hist[p][i]);
    }
    d[23] = h;
}
