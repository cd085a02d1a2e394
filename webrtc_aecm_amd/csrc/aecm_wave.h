// Wave-generic AECM block DSP: one 64-lane wavefront processes one stream, block after block.
//
// This header is the single source of the hot path.  It is written against a small "wave policy"
// W (lane-vector type, cross-lane exchange, reductions, ballots, LDS-table lookups, coalesced
// loads/stores) and is instantiated twice:
//   * webrtc_aecm_amd/csrc/aecm_kernels.hip  with the gfx950 policy (vi = int in a VGPR, DPP /
//     permlane / ds_bpermute cross-lane ops, tables in LDS)  -> the product;
//   * tests/sim/  with a 64-lane CPU simulator policy               -> test infrastructure that
//     lets `pytest -m "not gpu"` check this very code against the oracle without a GPU.
//
// Behavioural spec: WebRtcAecm_ProcessBlock and everything below it (reference
// aecm/aecm_core_c.cc:368-711; each section cites the lines it restates).  All arithmetic is
// Q-format integer and bit-exact; SURVEY.md Appendix A lists the C-semantics traps reproduced here.
//
// Data placement (lane t of 64):
//   time domain   : samples t (old half) and t+64 (new half) of a 128-sample analysis window
//   FFT           : radix-2 DIT on bit-reversed input == lane t starts with points n=t and n=t+64
//                   (positions bitrev7(n) = 2*bitrev6(t) and +1), one butterfly per lane per stage;
//                   between stages the two operands are re-paired by a lane exchange on lane bit
//                   5,4,3,2,1,0 (xor 32,16,8,4,2,1).  After the 7th stage lane t holds points
//                   bitrev6(t) and bitrev6(t)+64.
//   frequency dom.: bin t in lane t (one ds_bpermute after the forward FFT); bin 64 wave-uniform
//   delay estim.  : history slots t and t+64 (<100); mean_far/near thresholds for bins 12..43
//   synthesis     : IFFT output stays in bit-reversed lane order; the overlap buffer is kept in that
//                   order, so no permutation is needed before the coalesced 128-byte store.
#ifndef AECM_AMD_WAVE_H_
#define AECM_AMD_WAVE_H_

#include "aecm_ops.h"
#include "aecm_state.h"

// Profiling builds (-DAECM_MARKERS, device only) drop a comment marker into the ISA at each phase
// boundary, pinned by two live values, so tools can count instructions per phase.  No-op otherwise.
#if defined(AECM_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
#define AECM_PHASE_MARK(id, x, y) asm volatile("; AECM_MARK " #id : "+v"(x), "+v"(y))
#else
#define AECM_PHASE_MARK(id, x, y) ((void)0)
#endif

// Branch layout hints for the inverse transform's per-stage scaling: "no scaling" is the common case
// (the suppressed output is small), so that path should be the fall-through (measured +0.5 %).
// Kernel A/B switches (tools/ab_build.py); the defaults are the shipped forms.
#ifndef AECM_FWD_STAGE1_REAL
#define AECM_FWD_STAGE1_REAL 1        // forward stage 1 of a real signal knows its imaginary inputs are zero
#endif
#ifndef AECM_NOISE_TRACKING_FAST_PATH
#define AECM_NOISE_TRACKING_FAST_PATH 1   // comfort noise: short update when every estimate is >= 2^11 (one wave-uniform test)
#endif
#ifndef AECM_GAIN_ZERO_PATH
#define AECM_GAIN_ZERO_PATH 1             // supGain == 0 (wave-uniform): the spectrum passes through, only filters / estimator / seed move
#endif
#ifndef AECM_GAIN_ZERO_PATH_TICK
#define AECM_GAIN_ZERO_PATH_TICK 1
#endif
#ifndef AECM_NEAR_FILT_STEADY_PATH
#define AECM_NEAR_FILT_STEADY_PATH 1      // Wiener gain: short nearFilt update when the block's Q domain did not rise (wave-uniform)
#endif
#ifndef AECM_NEAR_FILT_STEADY_PATH_TICK
#define AECM_NEAR_FILT_STEADY_PATH_TICK 1      // tick kernel at 7 waves per SIMD, unstructurized uniform regions: 0.2306 ms with it, 0.2346 without
#endif
#ifndef AECM_NOISE_TRACKING_FAST_PATH_TICK
#define AECM_NOISE_TRACKING_FAST_PATH_TICK 1
#endif
#ifndef AECM_WIENER_DIV_FLOAT
#define AECM_WIENER_DIV_FLOAT 1       // the Wiener gain's 32-by-16-bit division in two float steps (W::divu_u32_u16)
#endif
// Joint scaling tests of the inverse transform's stages (see fft128); 0 = one test per stage.
#ifndef AECM_IFFT_GROUPED_SCALE_TESTS
#define AECM_IFFT_GROUPED_SCALE_TESTS 2
#endif
#ifndef AECM_IFFT_GROUPED_SCALE_TESTS_TICK
#define AECM_IFFT_GROUPED_SCALE_TESTS_TICK 2   // tick kernel, ms per tick at 65 536 sessions: 1 0.2306, 2 0.2299
#endif
#ifndef AECM_IFFT_GROUPED_SCALE_TESTS_CLEAN
#define AECM_IFFT_GROUPED_SCALE_TESTS_CLEAN 2
#endif
#if defined(__GNUC__)
#define AECM_UNLIKELY(c) __builtin_expect(!!(c), 0)
#define AECM_LIKELY(c) __builtin_expect(!!(c), 1)
#else
#define AECM_UNLIKELY(c) (c)
#define AECM_LIKELY(c) (c)
#endif

// Census builds (-DAECM_CENSUS_HOTPATH, tools/isa_phase_breakdown.py --hot): the branches a block of the steady state
// never takes (start-up over, far end active with a non-zero delay, no rescaling inside the inverse transform, none of
// the once-per-30-blocks channel bookkeeping) are declared unreachable, so the block loop becomes the straight-line
// hot path and a static count of it approximates the dynamic instruction mix.  Identity in every other build.
#if defined(AECM_CENSUS_HOTPATH) && defined(__HIP_DEVICE_COMPILE__)
#define AECM_STEADY_NEVER(c) (__builtin_expect(!!(c), 0) && (__builtin_unreachable(), true))
#define AECM_STEADY_ALWAYS(c) (__builtin_expect(!!(c), 1) || (__builtin_unreachable(), false))
#else
#define AECM_STEADY_NEVER(c) (c)
#define AECM_STEADY_ALWAYS(c) (c)
#endif

namespace aecm {

// ---- algorithm constants (reference aecm/aecm_defines.h:17-85, delay_estimator.cc:23-28) --------
constexpr int kConvLen = 512, kConvLen2 = 1024;
constexpr int kFarEnergyMin = 1025, kFarEnergyDiff = 929, kEnergyDevTol = 400, kFarEnergyVadRegion = 230;
constexpr int kMuMin = 10, kMuMax = 1, kMuDiff = 9;
constexpr int kMinMseCount = 20, kMinMseDiff = 29, kMseResolution = 5;
constexpr int kResChannel16 = 12, kResChannel32 = 28, kChannelVad = 16, kResSupgain = 8;
constexpr int kSupgainEpcDt = 200, kOneQ14 = 1 << 14, kNlpCompLow = 3277, kNlpCompHigh = kOneQ14;
constexpr int kBandFirst = 12, kBandLast = 43;
constexpr int kMaxBitCountsQ9 = 32 << 9, kProbOffset = 1024, kProbLowerLimit = 8704, kProbMinSpread = 2816;

// Per-bin persistent state; I = lane vector for bins 0..63, int for bin 64.
template <class I>
struct BinState {
    I ch_stored, ch_adapt16, ch_adapt32, echo_filt, near_filt, noise_est, low_ctr, high_ctr;
};

// Wave-uniform persistent state (mirrors the scalar members of the reference's AecmCore).
struct Uniform {
    int tot_count, seed, startup, hist_pos;
    int dfa_noisy_q, dfa_noisy_q_old, dfa_clean_q, dfa_clean_q_old;
    int far_log, fe_min, fe_max, fe_maxmin, fe_vad, fe_mse;
    int cur_vad, vad_cnt, first_vad, mse_cnt;
    int mse_adapt_old, mse_stored_old, mse_thresh;
    int sup_gain, sup_gain_old, noise_ctr;
    int far_init, near_init, min_prob, last_prob, last_delay;
    int mult, cng, nlp, fixed_delay, sg_a, sg_d, sg_dab, sg_dbd;
    // Not persistent: where the newest entry of the three log-energy histories sits in lanes 0..19.  The reference
    // shifts its arrays by one every block (aecm_core.cc:665-669) but only ever looks at the newest entry and at sums
    // over the first 20 (:943-952), so in registers they are 20-slot rings: one v_writelane per history and block
    // instead of a move + a whole-wave shift.  Entry k (0 = newest) is lane (log_pos + k) mod 20; load_state starts at 0
    // and store_state writes the canonical order back.
    int log_pos;
    // Not persistent: set by kernels whose policy lowers the issue priority of a wave that has run ahead of the launch's
    // other waves (wave_gfx950.h: kDynamicPrio; aecm_block_kernels.hip: the pipelined kernel).  0 everywhere else.
    int prio_drop = 0;
};

template <class W, bool kHasClean>
struct BlockEngine {
    using vi = typename W::vi;
    using vb = typename W::vb;

    // Joint scaling tests of the inverse transform (fft128): per kernel family, because the duplicated stage bodies cost
    // registers.
    static constexpr int kIfftGroupedTests = kHasClean ? AECM_IFFT_GROUPED_SCALE_TESTS_CLEAN
                                             : W::kTight ? AECM_IFFT_GROUPED_SCALE_TESTS_TICK : AECM_IFFT_GROUPED_SCALE_TESTS;

    // The comfort-noise estimator's short update (noise_bin<.., kTracking>) duplicates the phase's code: the one fast
    // kernel that then no longer fits its register budget (clean input + rotation: small launches only) keeps the long form.
    static constexpr bool kNoiseTrackingFastPath = AECM_NOISE_TRACKING_FAST_PATH && (W::kTight ? AECM_NOISE_TRACKING_FAST_PATH_TICK != 0 : true) &&
                                                   !(kHasClean && !W::kPhasePriority);

    static constexpr bool kGainZeroPath = AECM_GAIN_ZERO_PATH && (W::kTight ? AECM_GAIN_ZERO_PATH_TICK != 0 : true);
    static constexpr bool kNearFiltSteadyPath = AECM_NEAR_FILT_STEADY_PATH && (W::kTight ? AECM_NEAR_FILT_STEADY_PATH_TICK != 0 : true);

    // Everything a wave keeps in registers across the blocks of one launch.
    struct Regs {
        Uniform u;
        BinState<vi> b;       // bins 0..63
        BinState<int> b64;    // bin 64
        vi x_old, d_old, c_old, out_ovl;
        vi mean, bh0, bh1, m01, hq0, hq1;   // mean: binary-spectrum thresholds, far end in lanes 12..43, near end in the others (binary_spectra)
        vi near_log, adapt_log, stored_log;
        // lane constants
        vi lane, brev;
        vi lc[kLaneConstRows];             // LaneConstRow (aecm_state.h); unused when the policy serves them from a table
        vi table_index;                    // this lane's index into the policy's constant table, renewed every block
        vi k_p;                            // 32770 in a vector register (see fft_stage)
        int lcg_mul64, lcg_add64, bin64_div_magic, bin64_div_shift;
    };

    struct Spectrum {
        vi re, im, mag;       // bins 0..63
        int re64, mag64;      // bin 64 (imaginary part is 0 by construction)
        int q;                // dynamic Q of the block
    };

    // ------------------------------------------------------------------------------------------
    // lane constants
    // ------------------------------------------------------------------------------------------
    static AECM_HD vi bitrev6(vi t) {
        vi r = ((t & 1) << 5) | ((t & 2) << 3) | ((t & 4) << 1) | (lsr(t, 1) & 4) | (lsr(t, 3) & 2) | (lsr(t, 5) & 1);
        return r;
    }

    // LCG jump-ahead: after j steps seed_j = A^j * seed + C_j (mod 2^31); draw j-1 feeds bin j
    // (reference spl.cc:129-147, aecm_core_c.cc:143-150), so lane t needs j = t, bin 64 and the
    // carried-over seed need j = 64.
    static constexpr uint32_t lcg_pow(int j) { uint32_t a = 1; for (int i = 0; i < j; ++i) a *= 69069u; return a; }
    static constexpr uint32_t lcg_inc(int j) { uint32_t c = 0; for (int i = 0; i < j; ++i) c = c * 69069u + 1u; return c; }

    static AECM_HD void init_lane_constants(Regs &r, const uint32_t *consts) {
        r.lane = W::stream_lane_id();
        r.brev = bitrev6(r.lane);
        r.k_p = W::opaque_const(32770);
        r.lcg_mul64 = (int)lcg_pow(64);
        r.lcg_add64 = (int)lcg_inc(64);
        r.bin64_div_magic = (int)4162814457u;            // ceil(2^39 / 65) - 2^32, see div_magic()
        r.bin64_div_shift = 7;
        if (W::kLaneConstsInTable) return;               // device: read from the LDS copy of the blob at each use
        if (W::kPrecomputedConstants) {                  // device: one coalesced load per row
            for (int k = 0; k < kLaneConstRows; ++k) r.lc[k] = W::load_u32(consts + k * kLanes, r.lane);
            return;
        }
        // definition (the host builds the blob from the same formulas; tests compare the two)
        r.lc[LC_HANN_LO] = shl(sext16(W::hann(r.lane)), 2);         // analysis window << 2 (see window()), first half : hann[t]
        r.lc[LC_HANN_HI] = shl(sext16(W::hann(vi(64) - r.lane)), 2);     //                                second half: hann[64-t]
        r.lc[LC_HANN_SYN_LO] = sext16(W::hann(r.brev));             // synthesis window in IFFT output lane order
        r.lc[LC_HANN_SYN_HI] = sext16(W::hann(vi(64) - r.brev));
        vi a = vi(1), c = vi(0);
        int a64 = 1, c64 = 0;
        for (int j = 1; j < 64; ++j) {
            a64 = mul(a64, 69069);
            c64 = add(mul(c64, 69069), 1);
            auto here = (r.lane == vi(j));
            a = sel(here, vi(a64), a);
            c = sel(here, vi(c64), c);
        }
        r.lc[LC_LCG_MUL] = a;
        r.lc[LC_LCG_ADD] = c;
        W::div_magic_lanes(r.lane + 1, r.lc[LC_DIV_MAGIC], r.lc[LC_DIV_SHIFT]);
        r.lc[LC_BIN0_REAL] = sel(r.lane == 0, vi(0xffff), vi(-1));
        r.lc[LC_NOT_BIN0] = sel(r.lane == 0, vi(0), vi(-1));
        r.lc[LC_NLP_AVG_BAND] = sel((r.lane >= 4) & (r.lane <= 24), vi(-1), vi(0));
        r.lc[LC_NLP_LOW_BINS] = sel(r.lane < 24, vi(0x7fff0000), vi(0));
    }

    // The lane id for conditions inside the block loop: on the device a copy the compiler cannot see through, renewed
    // every block (table_index), so that lane masks are compared where they are used (one VALU instruction) instead of
    // being hoisted out of the loop into scalar register pairs that then spill (two v_readlane per use).
    static AECM_HD vi lane_now(const Regs &r) {
        if constexpr (W::kLaneConstsInTable) return r.table_index;
        else return r.lane;
    }
    // Per-lane constant of row ROW (hann rows are stored sign-extended).
    template <int ROW>
    static AECM_HD vi lane_const(const Regs &r) {
        if constexpr (W::kLaneConstsInTable) return W::template table_lane_const<ROW>(r.table_index);
        else return r.lc[ROW];
    }

    // ------------------------------------------------------------------------------------------
    // 128-point radix-2 transforms (reference aecm/complex_fft.c:241-491 "mode 1", with the
    // bit reversal of :181-209 folded into the lane placement)
    // ------------------------------------------------------------------------------------------
    static AECM_HD vi pack(vi re, vi im) { return zext16(re) | shl(im, 16); }
    static AECM_HD vi lo16(vi p) { return sext16(p); }
    static AECM_HD vi hi16(vi p) { return sar(p, 16); }

    // a = packed (re,im) of the butterfly's upper operand (position i), b = lower (position i+l).
    // Returns the sum of the per-stage shifts (inverse only; the reference's return value "scale").
    //
    // One butterfly of the reference (complex_fft.c:332-350 forward, :465-482 inverse):
    //     T  = wr*x_b - wi*y_b + 1          (likewise for the imaginary part)
    //     t  = T >> 1
    //     out = (int16)((x_a * 2^14 +- t + rnd) >> sh)    fwd: rnd = 2^14, sh = 15
    //                                                     inv: rnd = 2^13 << shift, sh = 14 + shift
    // i.e. out = bits [sh+15 : sh] of the 32-bit sum.  Multiplying the sum by 2^(16-sh) moves those
    // bits into the upper half of a 32-bit word, and multiplication is exact modulo 2^32:
    //     Y = (x_a << (30 - sh)) +- ((T >> 1) << (16 - sh)) + 2^15        out = upper half of Y
    // so the four narrowing shifts and the re-packing of a stage collapse into two byte permutes,
    // and T is one v_dot2_i32_i16 on the packed operand with packed twiddles (wr,-wi) / (wi,wr).
    // kRealInput: the imaginary parts of a and b are known to be zero on entry (forward transform of
    // a real signal, real_fft.c:59-65), which lets stage 0 (twiddle = (32767, 0)) skip half its work.
    //
    // Stage S pairs positions differing in bit S; the operands of stage S > 0 are brought together by
    // exchanging on lane bit (6 - S).

    // Forward stage 0 of a real signal: twiddle (32767, 0), imaginary inputs 0, so T_im = 1 and the
    // imaginary outputs are 0.  sh = 15: base = (x_a << 15) + 2^15 is even, so
    // Y+ = base + ((T >> 1) << 1) = (base + T) & ~1 and only the upper half of Y is kept: the product
    // accumulates straight onto base + 1 and bit 0 never matters for Y+.  For Y- = 2*base - Y+ the
    // upper half equals that of Z = 2*base + 1 - acc (acc = base + T): Y- = Z - [acc even], and Z can
    // only be a multiple of 2^16 when acc is odd.
    static AECM_HD void fft_stage0_real(vi &a, vi &b) {
        vi acc = add(mul24(vi(32767), lo16(b)), shl_add(lo16(a), 15, 32769));
        b = lsr(sub(shl_add(a, 16, 65537), acc), 16);
        a = lsr(acc, 16);
    }

    // The same stage fed straight from the analysis window (window()): a, b arrive as the products x * (hann << 2), whose
    // UPPER halves are the windowed samples w_a, w_b (the reference's (x * hann) >> 14 truncated to int16, aecm_core_c.cc:
    // 174-182).  v_mad_i32_i16 multiplies an upper half in place, so neither sample is ever extracted: with
    //     nacc = w_a * (-32768) + (w_b * (-32767) - 32769) = -acc          (acc as in fft_stage0_real, modulo 2^32)
    // the outputs are  a' = upper half of -nacc  and  b' = upper half of (w_a << 16) + 65537 + nacc,  where w_a << 16 is the
    // product with its lower half masked off: two multiply-adds, a mask, a three-operand add, a negation and two shifts for
    // the window's two narrowing shifts, their sign extensions and the whole of stage 0.
    static AECM_HD void fft_stage0_windowed(vi &a, vi &b) {
        const vi nacc = mad16_hi(a, vi(-32768), mad16_hi_uc(b, vi(-32767), -32769));
        b = lsr(add(add(a & (int)0xffff0000, nacc), 65537), 16);
        a = lsr(neg(nacc), 16);
    }

    // Forward stages 1..6 as 4 multiply-adds + 4 dot products + 2 byte permutes.  With K = -32768,
    // x_a * K + c = c - (x_a << 15) is one v_mad_i32_i16 taking either half of the packed operand, so the
    // base B = (x_a << 15) + 2^15 is formed NEGATED; since ~v = -v - 1 and the output is the upper half
    // of a word, computing the complement of the word gives the complement of the output for free:
    //   odd stages (true in, complemented out):
    //     ~(B + 1 + T) = (-32770 - (x_a << 15)) - T,      ~(B - T) = (-32769 - (x_a << 15)) + T
    //   even stages (complemented in a' = ~a, b' = ~b, true out): x = -x' - 1, so B = -(x_a' << 15) and
    //     T = -T' - s with T' the dot product on b' and s the sum of the twiddle's halves:
    //     B + 1 + T = ((1 - s) - (x_a' << 15)) - T',      B - T = (s - (x_a' << 15)) + T'.
    // All exact modulo 2^32.  Six stages: the last one (even) ends in true values.
    // kRealInput and S == 1: both operands are still purely real (stage 0 of a real signal leaves zero imaginary parts),
    // so the imaginary accumulators' bases x_a.im * K + c are just c: a uniform addend of the dot product.
    template <int S, int N, bool kRealInput>
    static AECM_HD void fft_stage_forward(vi (&aa)[N], vi (&bb)[N]) {
        static_assert(S >= 1 && S <= 6, "forward stages 1..6");
        constexpr bool kTrueIn = (S & 1) != 0;
        constexpr bool kImagZero = kRealInput && S == 1 && AECM_FWD_STAGE1_REAL;
        constexpr bool kNeedImB = S != 6;       // bins 65..127: only the real part of bin 64 is used (aecm_core_c.cc:297)
        vi w_re, w_im, nw_re, nw_im;
        W::template fwd_twiddles<S>(w_re, w_im, nw_re, nw_im);
        vi s_re = vi(0), c_re = vi(0), s_im = vi(0), c_im = vi(0);
        if constexpr (!kTrueIn) W::template fwd_offsets<kTrueIn ? 2 : S>(s_re, c_re, s_im, c_im);
        const vi k = vi(-32768);
        W::template exchange_all<6 - S, N>(aa, bb);
        for (int n = 0; n < N; ++n) {
            vi &a = aa[n], &b = bb[n];
            vi p_re, p_im, m_re, m_im = vi(0);
            if constexpr (kTrueIn) {
                p_re = dot2_i16(b, nw_re, mad16_lo_uc(a, k, -32770));
                m_re = dot2_i16(b, w_re, mad16_lo_uc(a, k, -32769));
                if constexpr (kImagZero) {
                    p_im = dot2_i16_uc(b, nw_im, -32770);
                    m_im = dot2_i16_uc(b, w_im, -32769);
                } else {
                    p_im = dot2_i16(b, nw_im, mad16_hi_uc(a, k, -32770));
                    m_im = dot2_i16(b, w_im, mad16_hi_uc(a, k, -32769));
                }
            } else {
                p_re = dot2_i16(b, nw_re, mad16_lo(a, k, c_re));
                m_re = dot2_i16(b, w_re, mad16_lo(a, k, s_re));
                p_im = dot2_i16(b, nw_im, mad16_hi(a, k, c_im));
                if (kNeedImB) m_im = dot2_i16(b, w_im, mad16_hi(a, k, s_im));
            }
            a = pack_hi16(p_re, p_im);
            b = kNeedImB ? pack_hi16(m_re, m_im) : lsr(m_re, 16);
        }
    }

    // The generic stage: every inverse stage (data-dependent scaling, complex_fft.c:382-396) and forward
    // stage 0 of a complex signal.  Returns the sum of the shifts applied (inverse only).
    // max |x| <= T over the 256 int16 of a transform held as packed operands a, b.  For an int16 x, |x| > T  <=>
    // (uint16)(x + T) > 2T (|-32768| counts as above every T), so per packed word: add, unsigned max over a and b,
    // saturating subtract of 2T, and "some half non-zero" is one compare + ballot -- no abs, no unpacking.
    template <int T>
    static AECM_HD bool fft_max_abs_within(const vi &a, const vi &b) {
        constexpr int kT = T * 0x10001, k2T = (int)((unsigned)(2 * T) * 0x10001u);
        const vi over = pk_sub_sat_u16(pk_max_u16(pk_add_u16(a, vi(kT)), pk_add_u16(b, vi(kT))), vi(k2T));
        return W::ballot(over != 0) == 0;
    }
    // How far one unscaled inverse stage can grow the largest magnitude M of a transform (complex_fft.c:465-482 with
    // shift 0): out = (x_a * 2^14 +- t + 2^13) >> 14 with t = (wr x_b - wi y_b + 1) >> 1 and |wr| + |wi| <= L for every
    // twiddle of the table (L = 46342 > sqrt(2) * 32768; tests/test_sim.py checks the table against it), so
    //     |out| <= (2^14 M + (L M + 1) / 2 + 1 + 2^13) / 2^14 + 1 < M (32768 + L) / 32768 + 2.
    // kNoScaleBound<K>: the largest M for which K consecutive stages provably all see max |x| <= 13573, i.e. none of
    // them scales (complex_fft.c:382-396) -- their per-stage tests can be skipped.
    static constexpr int fft_growth(int m) { return (int)(((int64_t)m * (32768 + 46342)) >> 15) + 2; }
    static constexpr int kScaleThreshold1 = 13573, kScaleThreshold2 = 27146;
    template <int K>
    static constexpr int no_scale_bound() {
        int best = 0;
        for (int m = 1; m <= kScaleThreshold1; ++m) {
            int v = m;
            bool ok = true;
            for (int k = 1; k < K; ++k) { v = fft_growth(v); if (v > kScaleThreshold1) { ok = false; break; } }
            if (!ok) break;
            best = m;
        }
        return best;
    }

    // A last stage's real-only output (S == 6: the inverse transform's a and b, the forward transform's b).  The value is the
    // upper half of a 32-bit accumulator.  Forward: moved down (bin 64's real part is read as a 16-bit value).  Inverse:
    // handed on as it is -- the synthesis window multiplies the upper half in place (v_mad_i32_i16 with op_sel), so the
    // shift down and the sign extension after it never happen; whoever wants the int16 takes hi16() of it.
    template <bool kInverse>
    static AECM_HD vi last_real(vi acc) {
        if constexpr (kInverse) return acc;
        else return lsr(acc, 16);
    }
    // kProvenNoScale (inverse only): the caller has shown that this stage's max |x| is <= 13573.
    template <bool kInverse, int S, int N, bool kProvenNoScale = false>
    static AECM_HD int fft_stage_generic(vi (&aa)[N], vi (&bb)[N], const vi &k_p) {
        int scale = 0;
        vi w_re, w_im, nw_re = vi(0), nw_im = vi(0);
        if constexpr (kInverse) W::template inv_twiddles<S>(w_re, w_im, nw_re, nw_im);   // (wr,-wi), (wi,wr) packed, and negated
        else W::template twiddles<S, kInverse>(w_re, w_im);
        // Last stage: the caller only consumes the real parts (inverse: real_fft.c:97-99) resp. bins
        // 0..63 complex and the real part of bin 64 (forward: aecm_core_c.cc:297)
        constexpr bool kNeedImA = !(S == 6 && kInverse), kNeedImB = S != 6;
        if constexpr (S > 0) W::template exchange_all<6 - (S > 0 ? S : 1), N>(aa, bb);
        for (int n = 0; n < N; ++n) {
            vi &a = aa[n], &b = bb[n];
            // Data-dependent scaling of the inverse transform (complex_fft.c:382-396): shift = [max|x| > 13573] +
            // [max|x| > 27146] over all 256 int16 of the transform (|-32768| counts as 32767: above both thresholds
            // either way).  "No scaling" is the usual case (the suppressed output is small) and the fall-through path;
            // the second threshold is only looked at when the first one fired.
            bool rescale = !kInverse;                      // the forward transform always scales by one bit (sh = 15)
            if (kInverse && !kProvenNoScale) rescale = !fft_max_abs_within<kScaleThreshold1>(a, b);
            if (AECM_STEADY_NEVER(AECM_UNLIKELY(rescale))) {
                int shift = 1;
                if (kInverse) {
                    shift = fft_max_abs_within<kScaleThreshold2>(a, b) ? 1 : 2;
                    scale += shift;
                }
                if (shift == 1) {
                    // sh = 15: as in fft_stage0_real, with a complex twiddle
                    vi acc_re = dot2_i16(b, w_re, shl_add(lo16(a), 15, 32769));      // base + T_re
                    vi z_re = sub(shl_add(a, 16, 65537), acc_re);                    // Z = 2*base + 1 - acc
                    vi acc_im = vi(0), z_im = vi(0);
                    if (kNeedImA) acc_im = dot2_i16(b, w_im, shl_add(hi16(a), 15, 32769));
                    if (kNeedImB) z_im = sub((a & (int)0xffff0000) + 65537, acc_im);
                    a = kNeedImA ? pack_hi16(acc_re, acc_im) : last_real<kInverse>(acc_re);
                    b = kNeedImB ? pack_hi16(z_re, z_im) : last_real<kInverse>(z_re);
                } else {
                    // shift == 2, sh = 16 (rare): Y = (x_a << 14) +- (T >> 1) + 2^15
                    vi t_re = sar(dot2_i16(b, w_re, vi(1)), 1);
                    vi base_re = shl(lo16(a), 14) + 32768;
                    if (kNeedImA) {
                        vi t_im = sar(dot2_i16(b, w_im, vi(1)), 1);
                        vi base_im = shl(hi16(a), 14) + 32768;
                        a = pack_hi16(add(base_re, t_re), add(base_im, t_im));
                        b = pack_hi16(sub(base_re, t_re), sub(base_im, t_im));
                    } else {
                        a = last_real<kInverse>(add(base_re, t_re));
                        b = last_real<kInverse>(sub(base_re, t_re));
                    }
                }
            } else {
                // sh = 14, inverse only.  base = (x_a << 16) + 2^15 has 15 zero low bits and (T >> 1) << 2 is 2T with
                // bit 1 cleared, so V = base + 2T equals Y+ except possibly in bit 1, and Z = 2*base + 2 - V equals Y- or
                // Y- + 2 with Y- a multiple of 4: the upper halves are those of Y+ and Y-.  With T = D + 1 (D the dot
                // product) and P = base + 2:   V = P + 2D,   Z = P - 2 - 2D = P + 2(-D - 1),
                // and -D - 1 is the dot product with the negated twiddle and addend -1: five instructions per
                // component (P, two dot products, two shift-adds), no subtraction, no constant moves.
                const vi p_re = shl(a, 16) | k_p;
                const vi v_re = shl_add(dot2_i16_c0(b, w_re), 1, p_re);
                const vi z_re = shl_add(dot2_i16_cm1(b, nw_re), 1, p_re);
                if (kNeedImA) {
                    const vi p_im = (a & (int)0xffff0000) | k_p;
                    const vi v_im = shl_add(dot2_i16_c0(b, w_im), 1, p_im);
                    const vi z_im = shl_add(dot2_i16_cm1(b, nw_im), 1, p_im);
                    a = pack_hi16(v_re, v_im);
                    b = pack_hi16(z_re, z_im);
                } else {
                    a = last_real<kInverse>(v_re);
                    b = last_real<kInverse>(z_re);
                }
            }
        }
        return scale;
    }

    // k_p: the constant 32770 pinned in a VGPR for the whole launch (Regs::k_p; only the inverse stages use it: one
    // register constant serves both their shift-or and their and-or, which take a single scalar operand).
    template <bool kInverse, bool kRealInput, int N, int S, bool kProvenNoScale = false>
    static AECM_HD int fft_stage(vi (&aa)[N], vi (&bb)[N], const vi &k_p) {
        if constexpr (S == 0 && kRealInput && !kInverse) {
            for (int n = 0; n < N; ++n) fft_stage0_real(aa[n], bb[n]);
            return 0;
        } else if constexpr (S > 0 && !kInverse) {
            fft_stage_forward<S, N, kRealInput>(aa, bb);
            return 0;
        } else {
            return fft_stage_generic<kInverse, S, N, kProvenNoScale>(aa, bb, k_p);
        }
    }

    // N independent transforms advance in lockstep (the far-end, near-end and optional clean
    // near-end windows of a block): each stage's twiddles are fetched once and are dead again before
    // the next stage, which keeps the register footprint of the tables at one stage's worth.
    //
    // Inverse transform: instead of testing every stage for scaling (5 instructions each), stages 0..2 are tested
    // together -- max |x| <= no_scale_bound<3>() = 2327 at stage 0 proves that none of them scales -- and so are stages 3
    // and 4 (bound 5621); a group whose joint test fails falls back to the per-stage tests, stages 5 and 6 always test
    // for themselves.  On speech-like data the joint tests pass for ~95 % / ~92 % of the blocks: 4.3 tests per block on
    // average instead of 7.  Bit-exact by construction: a skipped test is one whose outcome is proven.
    // kFirstStage = 1: stage 0 has been done by the caller (fft_stage0_windowed).
    template <bool kInverse, bool kRealInput, int N, int kFirstStage = 0>
    static AECM_HD int fft128(vi (&aa)[N], vi (&bb)[N], const vi &k_p) {
        static_assert(kFirstStage == 0 || (kFirstStage == 1 && !kInverse && kRealInput), "only the windowed forward transform starts at stage 1");
        int scale = 0;
        if constexpr (kInverse && N == 1 && kIfftGroupedTests > 0) {
            static_assert(no_scale_bound<1>() == 13573 && no_scale_bound<2>() == 5621 && no_scale_bound<3>() == 2327, "growth bound");
            if (AECM_STEADY_ALWAYS(AECM_LIKELY(fft_max_abs_within<no_scale_bound<3>()>(aa[0], bb[0])))) {
                fft_stage<kInverse, kRealInput, N, 0, true>(aa, bb, k_p);
                fft_stage<kInverse, kRealInput, N, 1, true>(aa, bb, k_p);
                fft_stage<kInverse, kRealInput, N, 2, true>(aa, bb, k_p);
            } else {
                scale += fft_stage<kInverse, kRealInput, N, 0>(aa, bb, k_p);
                scale += fft_stage<kInverse, kRealInput, N, 1>(aa, bb, k_p);
                scale += fft_stage<kInverse, kRealInput, N, 2>(aa, bb, k_p);
            }
            if (AECM_STEADY_ALWAYS(AECM_LIKELY(fft_max_abs_within<no_scale_bound<2>()>(aa[0], bb[0])))) {
                fft_stage<kInverse, kRealInput, N, 3, true>(aa, bb, k_p);
                fft_stage<kInverse, kRealInput, N, 4, true>(aa, bb, k_p);
            } else {
                scale += fft_stage<kInverse, kRealInput, N, 3>(aa, bb, k_p);
                scale += fft_stage<kInverse, kRealInput, N, 4>(aa, bb, k_p);
            }
            if (kIfftGroupedTests >= 2 && AECM_STEADY_ALWAYS(fft_max_abs_within<no_scale_bound<2>()>(aa[0], bb[0]))) {
                fft_stage<kInverse, kRealInput, N, 5, true>(aa, bb, k_p);
                fft_stage<kInverse, kRealInput, N, 6, true>(aa, bb, k_p);
            } else {
                scale += fft_stage<kInverse, kRealInput, N, 5>(aa, bb, k_p);
                scale += fft_stage<kInverse, kRealInput, N, 6>(aa, bb, k_p);
            }
            return scale;
        }
        if constexpr (kFirstStage == 0) scale += fft_stage<kInverse, kRealInput, N, 0>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 1>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 2>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 3>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 4>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 5>(aa, bb, k_p);
        scale += fft_stage<kInverse, kRealInput, N, 6>(aa, bb, k_p);
        return scale;
    }
    template <bool kInverse, bool kRealInput>
    static AECM_HD int fft128(vi &a, vi &b, const vi &k_p) {
        vi aa[1] = {a}, bb[1] = {b};
        const int scale = fft128<kInverse, kRealInput, 1>(aa, bb, k_p);
        a = aa[0];
        b = bb[0];
        return scale;
    }

    // ------------------------------------------------------------------------------------------
    // TimeToFrequencyDomain + WindowAndFFT (reference aecm/aecm_core_c.cc:166-191, 261-365)
    // ------------------------------------------------------------------------------------------
    // max_abs: max |x| over the 128 samples of the analysis window (the caller reduces the transforms
    // of one block together, see process_block).
    static AECM_HD vi abs_max(vi old_s, vi new_s) { return imax(iabs(old_s), iabs(new_s)); }
    // The analysis window of one signal, as the products stage 0 starts from; returns the dynamic Q.
    static AECM_HD int window(const Regs &r, vi old_s, vi new_s, int max_abs, vi &a, vi &b) {
        // dynamic Q: norm of max |x|, |-32768| clamped to 32767 (:288-289)
        int mx = imin(max_abs, 32767);
        int q = norm_w16(mx);
        // window (:174-182): scale, truncate to int16, multiply by sqrt-Hanning Q14, truncate
        // q = norm16(max |x|) with the maximum clamped to 32767, so x << q fits int16 for every sample of the window
        // (the reference's (int16_t) cast is the identity; for x = -32768 the clamp makes q = 0)
        // The window rows hold hann << 2 (<= 2^16), so the windowed sample (x * hann) >> 14 -- which fits int16: |x << q| <
        // 2^15, hann <= 2^14 -- is the UPPER half of the 32-bit product; stage 0 consumes it there (fft_stage0_windowed)
        a = mul24(as_i16(shl(old_s, q)), lane_const<LC_HANN_LO>(r));
        b = mul24(as_i16(shl(new_s, q)), lane_const<LC_HANN_HI>(r));
        return q;
    }
    // Spectrum of one signal from the forward transform's outputs.
    static AECM_HD void spectrum(const Regs &r, vi a, vi b, int q, Spectrum &sp) {
        // lane t holds X[bitrev6(t)] in a and X[bitrev6(t)+64] in b
        int x64 = W::readlane(b, 0);
        vi x = W::bpermute(a, r.brev);                      // bin t -> lane t
        x = x & lane_const<LC_BIN0_REAL>(r);                // bin 0: imaginary part forced to 0 (:296)
        sp.re = lo16(x);
        sp.im = hi16(pk_neg_i16(x));                        // conjugate (:188-190), int16 wrap
        sp.re64 = sext16(x64);                              // bin 64: imag forced to 0 (:297)
        // magnitudes (:298-362, AECM_WITH_ABS_APPROX off).  The reference special-cases re == 0 /
        // im == 0 (|.| of the other part) and saturates re^2+im^2 at 2^31-1; both are subsumed by an
        // exact floor(sqrt) on the unsigned sum: floor(sqrt(x^2)) == |x|, and the only sum above
        // 2^31-1 is 2^31 (re = im = -32768), whose floor-sqrt 46340 equals that of 2^31-1.
        // (-im)^2 == im^2 also for the wrapped -32768, so the packed bin squares itself.
        vi sq = dot2_i16_c0(x, x);                          // <= 2^31 as unsigned
        sp.mag = W::isqrt31(sq);                            // <= 46340 < 2^16
        sp.mag64 = zext16(iabs(sp.re64));
        sp.q = q;
    }

    // ------------------------------------------------------------------------------------------
    // Delay estimator (reference aecm/delay_estimator_wrapper.cc:92-125, delay_estimator.cc:369-382,
    // 521-664; the float "robust validation" half is disabled and output-dead)
    // ------------------------------------------------------------------------------------------
    // Both binary spectra of a block in one pass.  Only bins 12..43 of either spectrum take part, so the far-end
    // thresholds live in lanes 12..43 of r.mean and the near-end ones in the other 32 lanes (bin b in lane (b + 32) & 63:
    // the layout of the V_MEAN state word); the near-end magnitudes are brought there by one lane rotation (ds_bpermute,
    // off the vector ALU) and every instruction of the threshold update then works on 64 live lanes instead of twice on 32.
    // Returns the far word; near_word by reference.
    static AECM_HD int binary_spectra(Regs &r, vi far_mag, int far_q, vi near_mag, int near_q, int &near_word) {
        Uniform &u = r.u;
        const vb far_lanes = (lane_now(r) >= kBandFirst) & (lane_now(r) <= kBandLast);
        // Q15 (:100-103); each spectrum is shifted by its own Q before the merge (a uniform count), mag <= 46340 so v > 0 <=> mag > 0
        const vi v = sel(far_lanes, shl(far_mag, 15 - far_q), W::bpermute(shl(near_mag, 15 - near_q), (r.lane + 32) & 63));
        if (AECM_STEADY_NEVER(!(u.far_init & u.near_init))) {                                 // delay_estimator_wrapper.cc:106-114, per spectrum
            const vb fresh = (far_lanes & (u.far_init == 0)) | (!far_lanes & (u.near_init == 0));
            const vb seed = fresh & (v > 0);
            r.mean = sel(seed, sar(v, 1), r.mean);
            if (W::ballot(seed & far_lanes) != 0) u.far_init = 1;
            if (W::ballot(seed & !far_lanes) != 0) u.near_init = 1;
        }
        r.mean = mean_step(v, 6, r.mean);
        const uint64_t bits = W::ballot(v > r.mean);
        // bins 12..31 from lanes 44..63, bins 32..43 from lanes 0..11.  Two scalar shifts and an or, each half pinned to the
        // scalar unit: written as one 64-bit expression the compiler picks v_alignbit_b32 and moves both halves of the
        // (wave-uniform) ballot into vector registers for it
        near_word = W::per_block(W::per_block((int)((uint32_t)(bits >> 32) >> kBandFirst)) | W::per_block((int)((uint32_t)bits << (32 - kBandFirst))));
        return (int)(uint32_t)(bits >> kBandFirst);
    }

    static AECM_HD int process_binary(Regs &r, int near_word) {
        Uniform &u = r.u;
        vb valid1 = lane_now(r) < (kHistory - 64);       // compared here (one instruction), not hoisted into a spilled mask
        vb nz0 = r.bh0 != 0, nz1 = r.bh1 != 0;                                     // far_bit_counts > 0
        // :623-626; combined on the scalar side (a ballot of the or-ed conditions makes the compiler turn the mask into
        // an integer per lane and compare it again)
        bool any_far = (W::ballot(nz0) | (W::ballot(nz1) & ((uint64_t(1) << (kHistory - 64)) - 1))) != 0;
        // The 100 means are Q9 values <= 32 << 9 = 2^14 and their update (:550-564, delay_estimator.cc:690-702) never
        // leaves 16 bits: slots t and t + 64 are advanced together as the two halves of one word (r.m01, the layout of the
        // V_M01 state word), with packed 16-bit instructions.  Upper halves of lanes >= 36 (no slot) stay 0: their far
        // bit count is masked to 0, which freezes them.
        const vi fb = popc(r.bh0) | shl(sel(valid1, popc(r.bh1), vi(0)), 16);                     // far_bit_counts
        const vi bc = pk_shl_b16(popc(r.bh0 ^ vi(near_word)) | shl(popc(r.bh1 ^ vi(near_word)), 16), vi(0x00090009));   // Q9
        const vi factor = pk_sub_i16(vi(0x000d000d), pk_lshr_b16(pk_mul_lo_u16(fb, vi(0x00030003)), vi(0x00040004)));   // 13 - (3 fb >> 4), 7..13
        const vi diff = pk_sub_i16(bc, r.m01);                                                    // |.| <= 2^14
        const vi round = pk_ashr_i16(diff, vi(0x000f000f)) & pk_sub_i16(pk_shl_b16(vi(0x00010001), factor), vi(0x00010001));
        const vi step = pk_ashr_i16(pk_add_i16(diff, round), factor);                             // truncating toward zero
        r.m01 = pk_mad_u16(step, pk_nonzero_u16(fb), r.m01);                                                // only where the far word has a bit set (:558)
        // first minimum / maximum over the 100 means (:568-576): (mean << 7 | slot) is a total order
        const vi m0 = zext16(r.m01), m1 = lsr(r.m01, 16);
        vi key0 = shl(m0, 7) | r.lane;
        vi key1 = sel(valid1, shl(m1, 7) | (r.lane + 64), vi(0x7fffffff));
        int kmin, worst;
        W::reduce_min_max(imin(key0, key1), imax(m0, m1), kmin, worst);
        int best = kmin >> 7, candidate = kmin & 127;
        if (best >= kMaxBitCountsQ9) { best = kMaxBitCountsQ9; candidate = -1; }
        worst = imax(0, worst);
        int valley = worst - best;
        if (u.min_prob > kProbLowerLimit && valley > kProbMinSpread) {               // :593-606
            int thr = imax(best + kProbOffset, kProbLowerLimit);
            if (u.min_prob > thr) u.min_prob = thr;
        }
        u.last_prob = add(u.last_prob, 1);                                           // :609
        bool valid = (valley > kProbOffset) & (best < imax(u.min_prob, u.last_prob));           // best < min_prob || best < last_prob
        if (any_far && valid) {                                                      // :643-661
            u.last_delay = candidate;
            if (best < u.last_prob) u.last_prob = best;
        }
        return u.last_delay;
    }

    // ------------------------------------------------------------------------------------------
    // Energies / VAD / step size (reference aecm/aecm_core.cc:588-794)
    // ------------------------------------------------------------------------------------------
    // Scalar helpers, written as straight-line selects: every value here is wave-uniform and lives on the scalar unit,
    // where a branch costs more than the few instructions it skips.
    static AECM_HD int asym_filt(int old, int in, int step_pos, int step_neg) {      // :588-605
        const int d = in - old;                                                       // old, in are int16
        const int up = sext16(old + sar(d, step_pos)), down = sext16(old - sar(-d, step_neg));
        const int filtered = d < 0 ? down : up;                                       // "old > in" is d < 0; d == 0: both are old
        // old == 32767 or old == -32768 (the initial values): (old + 32769) mod 2^16 is 0 or 1 exactly for those two
        return ((old + 32769) & 0xffff) < 2 ? in : filtered;
    }

    static AECM_HD int log_energy_q8(int energy, int q) {                             // :612-628
        // (7 << 7) + ((31 - zeros) << 8) + frac - (q << 8) with frac = bits 30..23 of the normalised energy.  Taking
        // bit 31 (the leading one, 256) along saves the mask; the value lies in [-5760, 9087] for 0 <= q <= 26, so the
        // reference's (int16_t) is the identity.  zeros of 0 is never used (the result is replaced): no zero fix-up of the count.
        const int zeros = clz32_nz(energy);
        const int mant = lsr(shl(energy, zeros), 23);                                  // 256 + frac
        const int v = as_i16(mant - shl(zeros, 8) + (((31 - q) << 8) + (7 << 7) - 256));
        return energy != 0 ? v : (7 << 7);
    }

    // CalcLinearEnergies + CalcEnergies (:267-284, :644-755).  echo_est = channelStored * far.
    // near / near64: the near-end magnitudes, whose sum is the reference's dfaNoisySum (aecm_core_c.cc:446).
    static AECM_HD void calc_energies(Regs &r, vi far, int far64, int far_q, vi near, int near64, vi &echo_est,
                                      int &echo_est64) {
        Uniform &u = r.u;
        echo_est = mul24(r.b.ch_stored, far);
        echo_est64 = mul(r.b64.ch_stored, far64);
        int e_near, e_far, e_adapt, e_stored;            // the four uint32-wrapping sums in one pass
        W::reduce_add4(near, far, mul24(r.b.ch_adapt16, far), echo_est, e_near, e_far, e_adapt, e_stored);
        e_near = add(e_near, near64);
        e_far = add(e_far, far64);
        e_adapt = add(e_adapt, mul(r.b64.ch_adapt16, far64));
        e_stored = add(e_stored, echo_est64);
        // :665-669 as a ring (see Uniform::log_pos); pinned to the scalar unit (with the scalar registers as full as they
        // are the compiler otherwise keeps this counter in a VGPR: three vector instructions and a v_readfirstlane per block)
        {
            const int p = W::per_block(u.log_pos) - 1;
            u.log_pos = W::per_block(p < 0 ? kLogEntries - 1 : p);
        }
        r.near_log = W::writelane(r.near_log, log_energy_q8(e_near, u.dfa_noisy_q), u.log_pos);
        u.far_log = log_energy_q8(e_far, far_q);
        r.adapt_log = W::writelane(r.adapt_log, log_energy_q8(e_adapt, kResChannel16 + far_q), u.log_pos);
        r.stored_log = W::writelane(r.stored_log, log_energy_q8(e_stored, kResChannel16 + far_q), u.log_pos);

        if (AECM_STEADY_ALWAYS(u.far_log > kFarEnergyMin)) {                          // :692-730
            int inc_max = 4, dec_max = 11, inc_min = 11, dec_min = 3;
            if (AECM_STEADY_NEVER(u.startup == 0)) { inc_max = 2; dec_min = 2; inc_min = 8; }
            u.fe_min = asym_filt(u.fe_min, u.far_log, inc_min, dec_min);
            u.fe_max = asym_filt(u.fe_max, u.far_log, inc_max, dec_max);
            u.fe_maxmin = sext16(u.fe_max - u.fe_min);
            int t16 = sext16(2560 - u.fe_min);
            t16 = t16 > 0 ? sext16(sar(t16 * kFarEnergyVadRegion, 9)) : 0;
            t16 = sext16(t16 + kFarEnergyVadRegion);
            if ((u.startup == 0) | (u.vad_cnt > 1024)) {
                u.fe_vad = sext16(u.fe_min + t16);
            } else if (u.fe_vad > u.far_log) {
                u.fe_vad = sext16(u.fe_vad + sar(u.far_log + t16 - u.fe_vad, 6));
                u.vad_cnt = 0;
            } else {
                u.vad_cnt = sext16(u.vad_cnt + 1);
            }
            u.fe_mse = sext16(u.fe_vad + (1 << 8));
        }
        if (u.far_log > u.fe_vad) {                                                   // :733-740
            if ((u.startup == 0) | (u.fe_maxmin > kFarEnergyDiff)) u.cur_vad = 1;
        } else {
            u.cur_vad = 0;
        }
        if (AECM_STEADY_NEVER(u.cur_vad && u.first_vad)) {                            // :741-754
            u.first_vad = 0;
            int adapt0 = W::readlane(r.adapt_log, u.log_pos), near0 = W::readlane(r.near_log, u.log_pos);
            if (adapt0 > near0) {
                r.b.ch_adapt16 = sar(r.b.ch_adapt16, 3);
                r.b64.ch_adapt16 = sar(r.b64.ch_adapt16, 3);
                r.adapt_log = W::writelane(r.adapt_log, sext16(adapt0 - (3 << 8)), u.log_pos);
                u.first_vad = 1;
            }
        }
    }

    static AECM_HD int calc_step_size(const Uniform &u) {                             // :767-794
        int mu = kMuMax;
        if (AECM_STEADY_NEVER(!u.cur_vad)) {
            mu = 0;
        } else if (AECM_STEADY_ALWAYS(u.startup > 0)) {
            if (u.fe_min >= u.fe_max) {
                mu = kMuMin;
            } else {
                // mu = MU_MIN - 1 - (int16)(9 * t16 / farEnergyMaxMin), then clamped to >= MU_MAX = 1: every quotient in
                // [8, 32767] gives 1.  The usual case -- far energy at or above its tracked minimum, a tracked range of
                // at least 9 (so that the quotient of a numerator <= 9 * 32767 stays below 2^15 and the int16 cast is
                // the identity) -- therefore needs no division: three compare-subtract steps on the scalar unit instead
                // of a float-reciprocal division on the vector unit.
                const int t16 = sext16(u.far_log - u.fe_min);
                const int num = t16 * kMuDiff, den = u.fe_maxmin;
                int q;
                if (AECM_LIKELY(num >= 0 && den >= kMuDiff)) {
                    int rem = num;
                    q = 8;
                    if (rem < 8 * den) {
                        q = 0;
                        if (rem >= 4 * den) { q = 4; rem -= 4 * den; }
                        if (rem >= 2 * den) { q += 2; rem -= 2 * den; }
                        // 0 <= rem < 2 den here: +1 iff rem >= den.  As sign arithmetic on a pinned scalar: written as a
                        // comparison, the compiler turns the uniform boolean into an integer on the vector unit.
                        q += 1 + W::pin_uniform(sar(rem - den, 31));
                    }
                } else {
                    q = sext16(divi(num, den));
                }
                mu = sext16(kMuMin - 1 - q);
            }
            if (mu < kMuMax) mu = kMuMax;
        }
        return mu;
    }

    // ------------------------------------------------------------------------------------------
    // Channel update (reference aecm/aecm_core.cc:810-986)
    // ------------------------------------------------------------------------------------------
    // NLMS step of one bin (:831-921).  div_magic_k/div_shift_k: reciprocal of (bin index + 1).
    template <class I>
    static AECM_HD void nlms_bin(BinState<I> &s, I far, I dfa, I div_magic_k, I div_shift_k, int dfa_noisy_q, int far_q,
                                 int mu) {
        I zeros_ch = norm_u32(s.ch_adapt32);                                                   // may be negative right after InitEchoPath
        I zeros_far = norm_u32_nn(far);                                                        // far is a uint16
        I shift_ch_far = imax(I(32) - zeros_ch - zeros_far, I(0));                            // :836-850: 0 when zeros_ch + zeros_far > 31
        // shift_ch_far == 32 only for ch_adapt32 == 0 (norm 0 by convention) and far == 0: the product is 0 either way,
        // and sar() takes its count modulo 32, so the reference's "shift 0 when the norms sum to 0" needs no select
        I u1 = mul(sar(s.ch_adapt32, shift_ch_far), far);
        I zeros_num = norm_u32(u1);                                                           // :852-867
        I zeros_dfa = clz32(dfa);                                                             // NormU32, and 32 for dfa == 0 (:856-860)
        I t16 = as_i16(zeros_dfa - 2 + dfa_noisy_q - kResChannel32 - far_q + shift_ch_far);   // |.| < 128: counts, Q values
        // :861-867  "if (zerosNum > tmp16no1 + 1) { xfaQ = tmp16no1; dfaQ = zerosDfa - 2; } else { xfaQ = zerosNum - 2;
        // dfaQ = RESOLUTION_CHANNEL32 + far_q - dfaNoisyQDomain - shiftChFar + xfaQ; }".  The condition is zerosNum - 2 >= tmp16no1,
        // so xfaQ is the smaller of the two candidates; and tmp16no1 = zerosDfa - 2 - (RESOLUTION_CHANNEL32 + far_q -
        // dfaNoisyQDomain) + shiftChFar makes the first branch's dfaQ the second branch's formula too: no select at all.
        I xfa_q = imin(t16, as_i16(zeros_num - 2));
        I dfa_q = as_i16(I(kResChannel32 + far_q - dfa_noisy_q) - shift_ch_far + xfa_q);
        u1 = shift_u31(u1, xfa_q);                                                            // :869-872; both counts lie in [-28, 30]
        I u2 = shift_u31(dfa, dfa_q);
        I t1 = sub(u2, u1);
        zeros_num = norm_w32_nz(t1);                                                          // t1 == 0: no update, nothing below is used
        auto update = (t1 != 0) & (far > shl(I(kChannelVad), far_q));                         // :873
        I shift_num = imax(I(32) - (zeros_num + zeros_far), I(0));                            // :886-902: 0 when the sum > 31
        // |t1| < 2^30 (both aligned operands keep two bits of headroom, :852-872) and the shift above leaves the product
        // below 2^31, so the reference's sign juggling around its unsigned multiply (:886-902) and its truncating signed
        // division (:904) amount to: magnitude = (|t1| >> shift_num) * far, divide it, give it the sign of t1.
        I sign = sar(t1, 31);
        I mag = as_nonneg(mul(sar(sub(t1 ^ sign, sign), shift_num), far));
        I t2 = divu_by_magic(mag, div_magic_k, div_shift_k);                                  // :904  / (bin + 1)
        t2 = sub(t2 ^ sign, sign);
        I shift2res = as_i16(shift_num + shift_ch_far - xfa_q - mu - shl(I(30) - zeros_far, 1));
        t2 = sel(norm_w32(t2) < shift2res, I(0x7fffffff), shift_i(t2, shift2res));            // :906-912
        I n32 = add_sat32(s.ch_adapt32, t2);                                                  // :913-919
        n32 = sel(n32 < 0, I(0), n32);
        s.ch_adapt32 = sel(update, n32, s.ch_adapt32);
        s.ch_adapt16 = sel(update, sar(n32, 16), s.ch_adapt16);
    }

    static AECM_HD void store_adaptive_channel(Regs &r, vi far, int far64, vi &echo_est, int &echo_est64) {
        r.b.ch_stored = r.b.ch_adapt16;                                                       // :286-306
        r.b64.ch_stored = r.b64.ch_adapt16;
        echo_est = mul24(r.b.ch_stored, far);
        echo_est64 = mul(r.b64.ch_stored, far64);
    }

    static AECM_HD void update_channel(Regs &r, vi far, int far64, int far_q, vi dfa, int dfa64, int mu,
                                       vi &echo_est, int &echo_est64) {
        Uniform &u = r.u;
        if (AECM_STEADY_ALWAYS(mu)) {
            nlms_bin<vi>(r.b, far, dfa, lane_const<LC_DIV_MAGIC>(r), lane_const<LC_DIV_SHIFT>(r), u.dfa_noisy_q, far_q, mu);
            nlms_bin<int>(r.b64, far64, dfa64, r.bin64_div_magic, r.bin64_div_shift, u.dfa_noisy_q, far_q, mu);
        }
        if (AECM_STEADY_NEVER((u.startup == 0) & (u.cur_vad != 0))) {                         // :926-929
            store_adaptive_channel(r, far, far64, echo_est, echo_est64);
        } else {
            if (u.far_log < u.fe_mse) u.mse_cnt = 0;                                          // :931-935
            else u.mse_cnt = sext16(u.mse_cnt + 1);
            if (AECM_STEADY_NEVER(u.mse_cnt >= (kMinMseCount + 10))) {                        // :937-983
                vb first20 = lane_now(r) < kMinMseCount;
                int mse_stored = W::reduce_add(sel(first20, iabs(r.stored_log - r.near_log), vi(0)));
                int mse_adapt = W::reduce_add(sel(first20, iabs(r.adapt_log - r.near_log), vi(0)));
                if (((shl(mse_stored, kMseResolution)) < (kMinMseDiff * mse_adapt)) &
                    ((shl(u.mse_stored_old, kMseResolution)) < mul(kMinMseDiff, u.mse_adapt_old))) {
                    r.b.ch_adapt16 = r.b.ch_stored;                                           // :308-323
                    r.b.ch_adapt32 = shl(r.b.ch_stored, 16);
                    r.b64.ch_adapt16 = r.b64.ch_stored;
                    r.b64.ch_adapt32 = shl(r.b64.ch_stored, 16);
                } else if (((kMinMseDiff * mse_stored) > shl(mse_adapt, kMseResolution)) &
                           (mse_adapt < u.mse_thresh) & (u.mse_adapt_old < u.mse_thresh)) {
                    store_adaptive_channel(r, far, far64, echo_est, echo_est64);
                    if (u.mse_thresh == 0x7fffffff) {
                        u.mse_thresh = add(mse_adapt, u.mse_adapt_old);
                    } else {
                        int scaled = divi(mul(u.mse_thresh, 5), 8);
                        u.mse_thresh = add(u.mse_thresh, sar(mul(sub(mse_adapt, scaled), 205), 8));
                    }
                }
                u.mse_cnt = 0;
                u.mse_stored_old = mse_stored;
                u.mse_adapt_old = mse_adapt;
            }
        }
    }

    // cur_vad and the newest entries of the near-end and stored-channel log-energy histories: what channel_block leaves for it
    static AECM_HD int calc_suppression_gain(Uniform &u, int cur_vad, int near0, int stored0) {      // :1000-1052
        int sup;
        if (AECM_STEADY_NEVER(!cur_vad)) {
            sup = 0;
        } else {
            int dE = sext16(iabs(sext16(near0 - stored0)));
            if (dE < kEnergyDevTol) {
                if (dE < kSupgainEpcDt) {
                    int t32 = u.sg_dab * dE + (kSupgainEpcDt >> 1);
                    sup = sext16(u.sg_a - sext16(divi(t32, kSupgainEpcDt)));
                } else {
                    int t32 = u.sg_dbd * (kEnergyDevTol - dE) + ((kEnergyDevTol - kSupgainEpcDt) >> 1);
                    sup = sext16(u.sg_d + sext16(divi(t32, kEnergyDevTol - kSupgainEpcDt)));
                }
            } else {
                sup = u.sg_d;
            }
        }
        int t16 = sup > u.sup_gain_old ? sup : u.sup_gain_old;
        u.sup_gain_old = sup;
        u.sup_gain = sext16(u.sup_gain + sext16(sar(t16 - u.sup_gain, 4)));
        return u.sup_gain;
    }

    // ------------------------------------------------------------------------------------------
    // Wiener gain of one bin (reference aecm/aecm_core_c.cc:517-615)
    // ------------------------------------------------------------------------------------------
    // kUni: the instantiation for bin 64, whose operands are wave-uniform (on the device I is int for both).
    template <bool kUni, class I>
    static AECM_HD I uniform_hint(I x) {
        if constexpr (kUni) return I(W::pin_uniform((int)x));
        else return x;
    }
    // nearFilt of one bin (reference aecm/aecm_core_c.cc:552-579): the filter is brought to the block's Q domain and moved
    // 1/16 of the way to the near-end magnitude.
    // kQSteady: the caller has established dfaCleanQDomain <= dfaCleanQDomainOld (a wave-uniform fact; 94 % of the blocks of
    // the bench signal, instrumented oracle).  Then "zeros16 < dq" (:554) is false for every bin (a norm is >= 0), qDomainDiff
    // is 0, the old filter is shifted RIGHT by the uniform difference and the saturation test of :572 cannot fire: five
    // instructions instead of twenty-five.
    template <class I, bool kQSteady>
    static AECM_HD void near_filt_update(BinState<I> &s, I dfa_clean, int clean_q, int clean_q_old) {
        const int dqq = sext16(clean_q - clean_q_old);
        if constexpr (kQSteady) {
            const I t_a = as_i16(sar(s.near_filt, neg(dqq)));                                 // dqq in [-14, 0]
            const I t_b = sext16(dfa_clean);                                                  // the reference's (int16_t) of the uint16 magnitude
            s.near_filt = as_i16(as_i16(sar(sub(t_b, t_a), 4)) + t_a);                        // between t_a and t_b: no narrowing can change it
            return;
        }
        // nearFilt == 0 reads as norm 15 here, which no Q-domain step (|dqq| <= 14) exceeds: the reference's
        // "&& nearFilt" needs no test of its own.
        I zn = norm_w16_nz(s.near_filt);
        auto c = zn < dqq;
        I a_else = sext16(shift_i31(s.near_filt, I(dqq)));
        I q_diff = sel(c, zn - dqq, I(0));
        I t_a = sel(c, sext16(shl(s.near_filt, zn)), a_else);
        I t_b = sext16(lsr(dfa_clean, neg(q_diff)));                                          // q_diff == 0 unless c (then < 0)
        t_b = sext16(sext16(sar(sub(t_b, t_a), 4)) + t_a);
        I z2 = norm_w16_nz(t_b);                                                              // t_b == 0 has bit 0 clear: its norm does not matter
        auto weird = (t_b & sel(neg(q_diff) > z2, I(1), I(0))) != 0;                           // :572 literally
        s.near_filt = sel(weird, I(32767), sext16(shl(t_b, neg(q_diff))));                     // q_diff <= 0; a shift by 0 leaves the int16 t_b
    }

    // kGainZero: the caller has established supGain == 0 (wave-uniform).  Whatever the regime of :527-550 the gained echo
    // estimate is then 0 and the gain ONE_Q14 (:582): only the two filters move.
    template <class I, bool kUni = false, bool kQSteady = false, bool kGainZero = false>
    static AECM_HD I wiener_bin(BinState<I> &s, I echo_est, I dfa_clean, int sup_gain, int clean_q, int clean_q_old,
                                int zeros_xbuf) {
        // echoFilt += ((int64)(echoEst - echoFilt) * 50) >> 8 (:523-525): the arithmetic shift of the 64-bit product is
        // the upper word of d * (50 << 24), one multiply-high
        I d = sub(echo_est, s.echo_filt);
        s.echo_filt = add(s.echo_filt, mulhi_i32(d, I(50 << 24)));
        if constexpr (kGainZero) {
            near_filt_update<I, kQSteady>(s, dfa_clean, clean_q, clean_q_old);                // :552-579
            return I(kOneQ14);
        }

        // echoFilt == 0: the product below is 0 whatever the regime and the gain is then ONE_Q14 (:582): its norm is never looked at
        I zeros32 = norm_w32_nz(s.echo_filt) + 1;                                             // :527-550
        int zeros16 = norm_w16(sup_gain) + 1;
        I t16 = I(17) - zeros32 - zeros16;                                                    // :529: the "safe" regime is t16 <= 0
        int dq = clean_q - zeros_xbuf;
        I tpos = imax(t16, I(0));
        I res_diff = as_i16(tpos + (14 - kResChannel16 - kResSupgain + dq));
        // Three regimes (:534,:544,:548), all "low 32 bits of a product": echoFilt * supGain when nothing can overflow
        // (t16 <= 0), else the t16 excess bits are shifted out of supGain (if echoFilt has more headroom than that,
        // zeros32 > t16) or out of echoFilt.  supGain is never negative (it is a smoothed maximum of non-negative targets,
        // aecm_core.cc:1000-1052), so the reference's (uint16_t) casts are the identity and the three regimes are
        // one multiply of two right-shifted operands whose shift counts add up to max(t16, 0).
        I sh_l = sel(zeros32 > t16, I(0), tpos);
        I lhs = sar(s.echo_filt, sh_l);
        I rhs = sar(I(as_nonneg(sup_gain)), tpos - sh_l);
        I gained = mul(lhs, rhs);

        near_filt_update<I, kQSteady>(s, dfa_clean, clean_q, clean_q_old);                    // :552-579

        I g2 = add(gained, sar(s.near_filt, 1));                                              // :582-611
        I quot;                                                                               // WebRtcSpl_DivU32U16 (:584); nearFilt == 0 is overridden below
        if constexpr (kUni || !AECM_WIENER_DIV_FLOAT) quot = divu(g2, zext16(s.near_filt));
        else quot = W::divu_u32_u16(g2, zext16(s.near_filt));
        I t32 = shift_u31(quot, res_diff);                                                    // -20 <= res_diff <= 23
        // :597-611: hnl = ONE_Q14 - t32 clipped to [0, ONE_Q14], ONE_Q14 for a t32 that wrapped negative: ONE_Q14 - clamp(t32)
        I h;
        if constexpr (kUni) {
            // the same value as max(ONE_Q14 - max(t32, 0), 0) (no overflow: t32 >= -2^31), with the difference pinned to a
            // scalar register: fused, the expression is a saturating subtract, which only the vector unit has
            h = imax(uniform_hint<kUni>(I(kOneQ14) - imax(t32, I(0))), I(0));
        } else {
            h = I(kOneQ14) - imax(imin(t32, I(kOneQ14)), I(0));
        }
        return sel(gained == 0, I(kOneQ14), sel(s.near_filt == 0, I(0), h));
    }

    // ------------------------------------------------------------------------------------------
    // Comfort noise of one bin (reference aecm/aecm_core_c.cc:52-164); returns (uReal, uImag) as the UPPER HALVES of p_re, p_im
    // ------------------------------------------------------------------------------------------
    // kTracking: the caller has established that every noise estimate of the block is >= 2^11 (the usual state once the
    // estimator has found a noise floor: 2^11 in its Q15-like domain is 1/16 of an LSB of the spectrum).  Then neither
    // "small estimate" rule (:88-98 the decrement every 5th block below 2^minTrackShift <= 2^9, :117-125 the slow
    // increment below 2^11) can fire, which leaves one select between the tracking step down and the 1/2048 step up.
    // kSilent: the caller has established that the gain of every bin is ONE_Q14: the noise amplitude (ONE_Q14 - hnl) * est is 0
    // whatever the estimate (:142-147), so only the estimator moves and u_re, u_im are left alone.
    template <class I, bool kTracking = false, bool kSilent = false>
    static AECM_HD void noise_bin(BinState<I> &s, I dfa, I hnl, I rnd, I gate, int shift_n, int min_track, I &p_re, I &p_im) {
        I in = shl(dfa, shift_n);                                                             // :81-127
        auto lt = in < s.noise_est;
        I ne;
        if constexpr (kTracking) {
            I ne_lt = sub(s.noise_est, sar(sub(s.noise_est, in), min_track));
            I ne_ge = sel(sar(s.noise_est, 19) > 0, mul24(sar(s.noise_est, 11), I(2049)), sar(mul24(s.noise_est & 0x7ffff, I(2049)), 11));
            ne = sel(lt, ne_lt, ne_ge);
            s.low_ctr = sel(lt, I(0), s.low_ctr);
            s.high_ctr = sel(lt, s.high_ctr, I(0));
        } else {
        auto small = s.noise_est < (1 << min_track);
        I high_inc = s.high_ctr + 1;
        auto dec = high_inc >= 5;
        I ne_lt = sel(small, sel(dec, s.noise_est - 1, s.noise_est), sub(s.noise_est, sar(sub(s.noise_est, in), min_track)));
        I high_lt = sel(small, sel(dec, I(0), high_inc), s.high_ctr);
        auto c19 = sar(s.noise_est, 19) > 0;
        auto c11 = sar(s.noise_est, 11) > 0;
        I low_inc = s.low_ctr + 1;
        auto inc = low_inc >= 5;
        I ne_ge = sel(c19, mul24(sar(s.noise_est, 11), I(2049)),
                      sel(c11, sar(mul24(s.noise_est & 0x7ffff, I(2049)), 11),   // c11 && !c19: noise_est < 2^19
                          sel(inc, s.noise_est + sar(s.noise_est, 9) + 1, s.noise_est)));
        I low_ge = sel(c11, s.low_ctr, sel(inc, I(0), low_inc));              // c19 implies c11
        ne = sel(lt, ne_lt, ne_ge);
        s.low_ctr = sel(lt, I(0), low_ge);
        s.high_ctr = sel(lt, high_lt, I(0));
        }
        I t32 = sar(ne, shift_n);                                                             // :129-140
        auto clamp = t32 > 32767;
        t32 = sel(clamp, I(32767), t32);
        s.noise_est = sel(clamp, shl(I(32767), shift_n), ne);
        if constexpr (kSilent) return;
        I n16 = as_i16(sar(mul24(as_i16(I(kOneQ14) - hnl), as_i16(t32)), 14));               // 0 <= hnl <= 2^14, 0 <= t32 <= 32767
        n16 = n16 & gate;                                                                     // bin 0 gets no comfort noise (:146-147): one mask instead of two selects
        I idx = as_i16(sar(mul24(I(359), rnd), 15));                                            // :150
        // :153-156  uReal = (noise * cos) >> 13, uImag = (-noise * sin) >> 13 (|cos|, |sin| <= 2^13): with the amplitude times 8
        // (< 2^18) they are the upper halves of the two products, where a packed add takes them from (comfort_noise)
        const I n8 = shl(n16, 3);
        p_re = mul24(n8, W::cos360(idx));
        p_im = mul24(opaque_v(neg(n8)), W::sin360(idx));                  // opaque: a plain negation, not one redone in 24 bits
    }

    // ComfortNoise of the block (:52-164, called at :702-705) added to the suppressed spectrum (e = re | im << 16 of bins 0..63;
    // e_re64, e_im64).  kSilent: every gain is ONE_Q14, see noise_bin.
    template <bool kSilent>
    static AECM_HD void comfort_noise(Regs &r, const Spectrum &clean, vi hnl, int hnl64, vi &e, int &e_re64, int &e_im64) {
        Uniform &u = r.u;
        int shift_n = sext16(15 - u.dfa_clean_q);
        int min_track = 9;
        if (AECM_STEADY_NEVER(u.noise_ctr < 100)) { u.noise_ctr = sext16(u.noise_ctr + 1); min_track = 6; }
        // LCG jump-ahead: lane t gets the t-th of this block's 64 draws
        vi rnd = vi(0);
        int s64 = add(mul(r.lcg_mul64, u.seed), r.lcg_add64) & 0x7fffffff;
        if constexpr (!kSilent) {
            vi st = add(mul(lane_const<LC_LCG_MUL>(r), vi(u.seed)), lane_const<LC_LCG_ADD>(r)) & 0x7fffffff;
            rnd = as_i16(lsr(st, 16));                                              // st < 2^31
        }
        int rnd64 = sext16(lsr(s64, 16));
        u.seed = s64;
        vi p_re = vi(0), p_im = vi(0);
        int p_re64 = 0, p_im64 = 0;
        // every estimate at or above 2^11: the short form of the update (see noise_bin)
        const vi gate = lane_const<LC_NOT_BIN0>(r);                                    // 0 in lane 0, all ones elsewhere
        const bool tracking = kNoiseTrackingFastPath && (W::ballot(r.b.noise_est > vi(2047)) == ~0ull) & (r.b64.noise_est > 2047);
        if (AECM_STEADY_ALWAYS(AECM_LIKELY(tracking))) {
            noise_bin<vi, true, kSilent>(r.b, clean.mag, hnl, rnd, gate, shift_n, min_track, p_re, p_im);
            noise_bin<int, true, kSilent>(r.b64, clean.mag64, hnl64, rnd64, -1, shift_n, min_track, p_re64, p_im64);
        } else {
            noise_bin<vi, false, kSilent>(r.b, clean.mag, hnl, rnd, gate, shift_n, min_track, p_re, p_im);
            noise_bin<int, false, kSilent>(r.b64, clean.mag64, hnl64, rnd64, -1, shift_n, min_track, p_re64, p_im64);
        }
        if constexpr (kSilent) return;
        // :160-163  efw = AddSatW16(efw, u), real and imaginary part in one packed saturating add; bin 64: uImag = 0 (:158)
        e = pk_add_sat_i16(e, pack_hi16(p_re, p_im));
        e_re64 = sat16(e_re64 + sar(p_re64, 16));
        e_im64 = sat16(e_im64);
    }

    // ------------------------------------------------------------------------------------------
    // state load / store
    // ------------------------------------------------------------------------------------------
    static AECM_HD void load_state(Regs &r, const uint32_t *vec, const int32_t *scal) {
        auto V = [&](int f) { return W::load_u32(vec + f * kLanes, r.lane); };
        const vi up36 = (r.lane + kSecondPass) & 63;            // lane t reads what lane t + 36 holds
        vi w = V(V_XD_OLD);
        r.x_old = lo16(w); r.d_old = hi16(w);
        w = V(V_OUTBUF);
        r.out_ovl = lo16(w); r.c_old = hi16(w);
        w = V(V_CH16);
        r.b.ch_stored = lo16(w); r.b.ch_adapt16 = hi16(w);
        r.b.ch_adapt32 = V(V_CH32);
        r.b.echo_filt = V(V_ECHOFILT);
        w = V(V_NEARFILT);
        r.b.near_filt = lo16(w); r.b.low_ctr = lsr(w, 16) & 7; r.b.high_ctr = lsr(w, 19) & 7;
        const vi hq = V(V_HQ);
        r.hq0 = zext16(hq) | shl(lsr(w, 22) & 31, 16);           // far_history[slot][64] | far_q << 16 (see process_block)
        r.hq1 = lsr(hq, 16) | shl(lsr(w, 27), 16);               // lanes >= 36 carry log entries here: never read as slots
        r.stored_log = sext16(W::bpermute(lsr(hq, 16), up36));
        r.b.noise_est = V(V_NOISE);
        w = V(V_MEAN);
        r.mean = w;                                              // far-end thresholds in lanes 12..43, near-end ones in the others
        r.bh0 = V(V_BH0);
        w = V(V_BH1);
        r.bh1 = sel(r.lane < kSecondPass, w, vi(0));
        w = W::bpermute(w, up36);
        r.near_log = lo16(w); r.adapt_log = hi16(w);
        w = V(V_M01);
        r.m01 = w;
        const auto row = W::load_scalar_row(scal);
        Uniform &u = r.u;
        u.log_pos = 0;
        u.tot_count = row.get(S_TOTCOUNT); u.seed = row.get(S_SEED); u.startup = row.get(S_STARTUP); u.hist_pos = row.get(S_HISTPOS);
        u.dfa_noisy_q = row.get(S_DFANOISYQ); u.dfa_noisy_q_old = row.get(S_DFANOISYQ_OLD);
        u.dfa_clean_q = row.get(S_DFACLEANQ); u.dfa_clean_q_old = row.get(S_DFACLEANQ_OLD);
        u.far_log = row.get(S_FARLOG); u.fe_min = row.get(S_FE_MIN); u.fe_max = row.get(S_FE_MAX); u.fe_maxmin = row.get(S_FE_MAXMIN);
        u.fe_vad = row.get(S_FE_VAD); u.fe_mse = row.get(S_FE_MSE); u.cur_vad = row.get(S_CURVAD); u.vad_cnt = row.get(S_VADCNT);
        u.first_vad = row.get(S_FIRSTVAD); u.mse_cnt = row.get(S_MSECNT); u.mse_adapt_old = row.get(S_MSE_ADAPT_OLD);
        u.mse_stored_old = row.get(S_MSE_STORED_OLD); u.mse_thresh = row.get(S_MSE_THRESH);
        u.sup_gain = row.get(S_SUPGAIN); u.sup_gain_old = row.get(S_SUPGAIN_OLD); u.noise_ctr = row.get(S_NOISECTR);
        u.far_init = row.get(S_FAR_INIT); u.near_init = row.get(S_NEAR_INIT); u.min_prob = row.get(S_MIN_PROB);
        u.last_prob = row.get(S_LAST_PROB); u.last_delay = row.get(S_LAST_DELAY);
        u.mult = row.get(S_MULT); u.cng = row.get(S_CNG); u.nlp = row.get(S_NLP); u.fixed_delay = row.get(S_FIXED_DELAY);
        u.sg_a = row.get(S_SG_A); u.sg_d = row.get(S_SG_D); u.sg_dab = row.get(S_SG_DAB); u.sg_dbd = row.get(S_SG_DBD);
        BinState<int> &e = r.b64;
        e.ch_stored = row.get(S_B64_CHSTORED); e.ch_adapt16 = row.get(S_B64_CHADAPT16); e.ch_adapt32 = row.get(S_B64_CHADAPT32);
        e.echo_filt = row.get(S_B64_ECHOFILT); e.near_filt = row.get(S_B64_NEARFILT); e.noise_est = row.get(S_B64_NOISE);
        e.low_ctr = row.get(S_B64_LOWCTR); e.high_ctr = row.get(S_B64_HIGHCTR);
    }

    // The previous block's far / near input samples: the only state front_block reads.
    static AECM_HD void load_time_state(const uint32_t *vec, vi lane, vi &x_old, vi &d_old) {
        const vi w = W::load_u32(vec + V_XD_OLD * kLanes, lane);
        x_old = lo16(w);
        d_old = hi16(w);
    }
    static AECM_HD void store_time_state(uint32_t *vec, vi lane, vi x_old, vi d_old) {
        W::store_u32(vec + V_XD_OLD * kLanes, lane, pack(x_old, d_old));
    }

    // The overlap buffer (and the clean input's previous block, which shares its word): the only state tail_block touches.
    static AECM_HD void load_tail_state(const uint32_t *vec, vi lane, vi &out_ovl, vi &c_old) {
        const vi w = W::load_u32(vec + V_OUTBUF * kLanes, lane);
        out_ovl = lo16(w);
        c_old = hi16(w);
    }
    static AECM_HD void store_tail_state(uint32_t *vec, vi lane, vi out_ovl, vi c_old) {
        W::store_u32(vec + V_OUTBUF * kLanes, lane, pack(out_ovl, c_old));
    }

    // The delay estimator's state: all delay_block touches (the thresholds, the bit histories of the far end, the 100 means, five
    // scalars).  V_BH1 carries the second bit-history word in lanes 0..35 only (lanes 36..55 belong to two log-energy histories).
    static AECM_HD void load_delay_state(Regs &r, const uint32_t *vec, const int32_t *scal) {
        auto V = [&](int f) { return W::load_u32(vec + f * kLanes, r.lane); };
        r.mean = V(V_MEAN);
        r.bh0 = V(V_BH0);
        r.bh1 = sel(r.lane < kSecondPass, V(V_BH1), vi(0));
        r.m01 = V(V_M01);
        const auto row = W::load_scalar_row(scal);
        Uniform &u = r.u;
        u.far_init = row.get(S_FAR_INIT); u.near_init = row.get(S_NEAR_INIT); u.min_prob = row.get(S_MIN_PROB);
        u.last_prob = row.get(S_LAST_PROB); u.last_delay = row.get(S_LAST_DELAY);
    }
    static AECM_HD void store_delay_state(const Regs &r, uint32_t *vec, int32_t *scal) {
        auto V = [&](int f, vi w) { W::store_u32(vec + f * kLanes, r.lane, w); };
        V(V_MEAN, r.mean);
        V(V_BH0, r.bh0);
        W::store_u32_if(r.lane < kSecondPass, vec + V_BH1 * kLanes, r.lane, r.bh1);
        V(V_M01, r.m01);
        if (W::is_first_lane()) {
            const Uniform &u = r.u;
            W::store_scalar(scal, S_FAR_INIT, u.far_init); W::store_scalar(scal, S_NEAR_INIT, u.near_init); W::store_scalar(scal, S_MIN_PROB, u.min_prob);
            W::store_scalar(scal, S_LAST_PROB, u.last_prob); W::store_scalar(scal, S_LAST_DELAY, u.last_delay);
        }
    }

    // kTimeState = false: without V_XD_OLD (a wave that only ran back_block does not have it: store_time_state);
    // kTailState = false: without V_OUTBUF (a wave that only ran middle_block: store_tail_state);
    // kDelayState = false: without the delay estimator's state (a wave that was handed the delays: store_delay_state)
    template <bool kTimeState = true, bool kTailState = true, bool kDelayState = true>
    static AECM_HD void store_state(const Regs &r, uint32_t *vec, int32_t *scal) {
        auto V = [&](int f, vi w) { W::store_u32(vec + f * kLanes, r.lane, w); };
        const vb second = r.lane < kSecondPass;                  // live lanes of the second-pass words
        const vb logs = (r.lane >= kSecondPass) & (r.lane < kSecondPass + kLogEntries);
        // lane t (36..55) takes log entry k = t - 36, which lives in ring lane (log_pos + k) mod 20
        const vi ring = (r.lane - kSecondPass) + r.u.log_pos;
        const vi down36 = sel(ring >= kLogEntries, ring - kLogEntries, ring) & 63;
        if constexpr (kTimeState) V(V_XD_OLD, pack(r.x_old, r.d_old));
        if constexpr (kTailState) V(V_OUTBUF, pack(r.out_ovl, r.c_old));
        V(V_CH16, pack(r.b.ch_stored, r.b.ch_adapt16));
        V(V_CH32, r.b.ch_adapt32);
        V(V_ECHOFILT, r.b.echo_filt);
        V(V_NEARFILT, zext16(r.b.near_filt) | shl(r.b.low_ctr & 7, 16) | shl(r.b.high_ctr & 7, 19) | shl(lsr(r.hq0, 16) & 31, 22) |
                          sel(second, shl(lsr(r.hq1, 16), 27), vi(0)));
        V(V_NOISE, r.b.noise_est);
        if constexpr (kDelayState) {
            V(V_MEAN, r.mean);
            V(V_BH0, r.bh0);
            V(V_BH1, sel(second, r.bh1, sel(logs, W::bpermute(pack(r.near_log, r.adapt_log), down36), vi(0))));
            V(V_M01, r.m01);
        } else {
            W::store_u32_if(!second, vec + V_BH1 * kLanes, r.lane, sel(logs, W::bpermute(pack(r.near_log, r.adapt_log), down36), vi(0)));
        }
        V(V_HQ, zext16(r.hq0) | shl(sel(second, r.hq1, sel(logs, W::bpermute(r.stored_log, down36), vi(0))), 16));
        if (W::is_first_lane()) {
            const Uniform &u = r.u;
            W::store_scalar(scal, S_TOTCOUNT, u.tot_count); W::store_scalar(scal, S_SEED, u.seed); W::store_scalar(scal, S_STARTUP, u.startup); W::store_scalar(scal, S_HISTPOS, u.hist_pos);
            W::store_scalar(scal, S_DFANOISYQ, u.dfa_noisy_q); W::store_scalar(scal, S_DFANOISYQ_OLD, u.dfa_noisy_q_old);
            W::store_scalar(scal, S_DFACLEANQ, u.dfa_clean_q); W::store_scalar(scal, S_DFACLEANQ_OLD, u.dfa_clean_q_old);
            W::store_scalar(scal, S_FARLOG, u.far_log); W::store_scalar(scal, S_FE_MIN, u.fe_min); W::store_scalar(scal, S_FE_MAX, u.fe_max); W::store_scalar(scal, S_FE_MAXMIN, u.fe_maxmin);
            W::store_scalar(scal, S_FE_VAD, u.fe_vad); W::store_scalar(scal, S_FE_MSE, u.fe_mse); W::store_scalar(scal, S_CURVAD, u.cur_vad); W::store_scalar(scal, S_VADCNT, u.vad_cnt);
            W::store_scalar(scal, S_FIRSTVAD, u.first_vad); W::store_scalar(scal, S_MSECNT, u.mse_cnt); W::store_scalar(scal, S_MSE_ADAPT_OLD, u.mse_adapt_old);
            W::store_scalar(scal, S_MSE_STORED_OLD, u.mse_stored_old); W::store_scalar(scal, S_MSE_THRESH, u.mse_thresh);
            W::store_scalar(scal, S_SUPGAIN, u.sup_gain); W::store_scalar(scal, S_SUPGAIN_OLD, u.sup_gain_old); W::store_scalar(scal, S_NOISECTR, u.noise_ctr);
            if constexpr (kDelayState) {
                W::store_scalar(scal, S_FAR_INIT, u.far_init); W::store_scalar(scal, S_NEAR_INIT, u.near_init); W::store_scalar(scal, S_MIN_PROB, u.min_prob);
                W::store_scalar(scal, S_LAST_PROB, u.last_prob); W::store_scalar(scal, S_LAST_DELAY, u.last_delay);
            }
            const BinState<int> &e = r.b64;
            W::store_scalar(scal, S_B64_CHSTORED, e.ch_stored); W::store_scalar(scal, S_B64_CHADAPT16, e.ch_adapt16); W::store_scalar(scal, S_B64_CHADAPT32, e.ch_adapt32);
            W::store_scalar(scal, S_B64_ECHOFILT, e.echo_filt); W::store_scalar(scal, S_B64_NEARFILT, e.near_filt); W::store_scalar(scal, S_B64_NOISE, e.noise_est);
            W::store_scalar(scal, S_B64_LOWCTR, e.low_ctr); W::store_scalar(scal, S_B64_HIGHCTR, e.high_ctr);
        }
    }

    // ------------------------------------------------------------------------------------------
    // One block (reference aecm/aecm_core_c.cc:368-711).  far_new/near_new/clean_new: lane t holds
    // sample t of the new 64-sample block (sign-extended).  Returns the output block in IFFT lane
    // order: lane t holds out[bitrev6(t)].
    // ------------------------------------------------------------------------------------------
    // A block comes in two parts.  front_block: TimeToFrequencyDomain of the far-end, near-end and (optional) clean near-end
    // windows (:439, :442, :452) -- a function of the input samples alone (the previous and the new 64 of each signal), not
    // of the adaptive state.  back_block: everything else, which is sequential per stream.  process_block is one after the
    // other in one wave; the kernel for launches smaller than the chip (aecm_block_kernels.hip: aecm_process_pipelined_kernel)
    // runs the two parts in different waves of a workgroup, one block apart.
    static AECM_HD void update_startup(Uniform &u) {
        if (AECM_STEADY_NEVER(u.startup < 2)) u.startup = (gtu(u.tot_count, kConvLen - 1) ? 1 : 0) + (gtu(u.tot_count, kConvLen2 - 1) ? 1 : 0);  // :420-424
    }
    // r: only the lane constants are used (lane, brev, table_index).  The three transforms advance in lockstep.
    static AECM_HD void front_block(const Regs &r, vi x_old, vi far_new, vi d_old, vi near_new, vi c_old, vi clean_new,
                                    Spectrum &xf, Spectrum &df, Spectrum &cf) {
        AECM_PHASE_MARK(0, far_new, near_new);
        W::template phase_priority<1>(r.u.prio_drop);
        {
            constexpr int kSignals = kHasClean ? 3 : 2;
            int max_abs[3], q[3] = {0, 0, 0};
            W::reduce_max2(abs_max(x_old, far_new), abs_max(d_old, near_new), max_abs[0], max_abs[1]);
            vi fa[kSignals], fb[kSignals];
            q[0] = window(r, x_old, far_new, max_abs[0], fa[0], fb[0]);
            q[1] = window(r, d_old, near_new, max_abs[1], fa[1], fb[1]);
            if (kHasClean) {
                max_abs[2] = W::reduce_max(abs_max(c_old, clean_new));
                q[2] = window(r, c_old, clean_new, max_abs[2], fa[kSignals - 1], fb[kSignals - 1]);
            }
            for (int n = 0; n < kSignals; ++n) fft_stage0_windowed(fa[n], fb[n]);
            fft128<false, true, kSignals, 1>(fa, fb, r.k_p);
            spectrum(r, fa[0], fb[0], q[0], xf);
            AECM_PHASE_MARK(1, xf.mag, xf.re);
            W::template phase_priority<2>(r.u.prio_drop);
            spectrum(r, fa[1], fb[1], q[1], df);
            if (kHasClean) spectrum(r, fa[kSignals - 1], fb[kSignals - 1], q[2], cf);
        }
        AECM_PHASE_MARK(2, df.mag, df.re);
    }

    // front_block without the spectra (and without a clean input): the forward transforms' outputs of the far-end (index 0) and
    // the near-end signal and their dynamic Q; spectrum() of each is then somebody else's work (the pipelined kernel's balanced
    // form hands these over instead of the spectra, so that the magnitudes are computed by the waves that have time to spare).
    static AECM_HD void front_transforms(const Regs &r, vi x_old, vi far_new, vi d_old, vi near_new, vi (&fa)[2], vi (&fb)[2], int (&q)[2]) {
        int max_abs[2];
        W::reduce_max2(abs_max(x_old, far_new), abs_max(d_old, near_new), max_abs[0], max_abs[1]);
        q[0] = window(r, x_old, far_new, max_abs[0], fa[0], fb[0]);
        q[1] = window(r, d_old, near_new, max_abs[1], fa[1], fb[1]);
        for (int n = 0; n < 2; ++n) fft_stage0_windowed(fa[n], fb[n]);
        fft128<false, true, 2, 1>(fa, fb, r.k_p);
    }

    static AECM_HD vi process_block(Regs &r, uint16_t *hist, vi far_new, vi near_new, vi clean_new) {
        update_startup(r.u);
        if (W::kLaneConstsInTable) r.table_index = W::table_index_for_this_block();
        Spectrum xf, df, cf;
        front_block(r, r.x_old, far_new, r.d_old, near_new, r.c_old, clean_new, xf, df, cf);
        const vi out = back_block(r, hist, xf, df, cf);
        r.x_old = far_new;                                                            // :239-245
        r.d_old = near_new;
        if (kHasClean) r.c_old = clean_new;
        return out;
    }

    // back_block itself is two parts again.  middle_block: delay estimator ... comfort noise, the sequential heart of a stream;
    // it ends with the residual spectrum packed as the inverse transform's operands.  tail_block: inverse transform, synthesis
    // window and overlap-add (:193-246) -- a function of those operands, the block's Q domain and the overlap buffer alone.
    // The pipelined kernel can run the tails of a workgroup's streams in a wave of their own, one block behind the middles.
    struct TailInput {
        vi a, b;              // lane t: Y[t] and Y[64 - t] (lane 0: Y[64]), packed (re, -im)
        int clean_q;          // dfaCleanQDomain of the block
    };
    // Of xf only mag / mag64 / q are read; cf only with a clean input.
    static AECM_HD vi back_block(Regs &r, uint16_t *hist, const Spectrum &xf, const Spectrum &df, const Spectrum &cf) {
        const TailInput t = middle_block(r, hist, xf, df, cf);
        return tail_block(r, t.a, t.b, t.clean_q);
    }

    // The delay estimator of a block (delay_estimator_wrapper.cc:92-125, 233-263, 447-476; delay_estimator.cc:369-382, 521-664):
    // both binary spectra, the far word into its history, the 100 means, the delay.  It reads the two magnitude spectra and the Q
    // domains of the block and touches nothing but its own state (r.mean, r.bh0, r.bh1, r.m01, far_init / near_init / min_prob /
    // last_prob / last_delay).  (Round 5 tried it in the pipelined kernel's front waves, one block ahead of everything else: slower at
    // every size -- profiles/r05_experiments.md section 1.5.)  Returns last_delay (-2 until the first valid estimate).
    static AECM_HD int delay_block(Regs &r, const Spectrum &xf, const Spectrum &df) {
        int near_word;
        {
            int word = binary_spectra(r, xf.mag, xf.q, df.mag, df.q, near_word);
            int carry = W::readlane(r.bh0, 63);
            r.bh0 = W::shift_up1(r.bh0, word);
            r.bh1 = W::shift_up1(r.bh1, carry);
        }
        AECM_PHASE_MARK(3, r.bh0, r.mean);
        W::template phase_priority<4>(r.u.prio_drop);
        // near binary spectrum -> delay (delay_estimator_wrapper.cc:447-476)
        return process_binary(r, near_word);
    }
    // The delay a block works with (:479-488) and the far-history slot it then fetches (AlignedFarend, aecm_core.cc:157-172), for
    // a block whose UpdateFarHistory left the write position at hist_pos.
    static AECM_HD int effective_delay(const Uniform &u, int last_delay) {
        int delay = last_delay;
        if (delay == -2) delay = 0;                                                   // :479-483
        if (W::per_block(u.fixed_delay) >= 0) delay = u.fixed_delay;                  // :485-488
        return delay;
    }
    static AECM_HD int aligned_slot(int hist_pos, int delay) {
        const int pos = hist_pos - delay;
        return pos < 0 ? pos + kHistory : pos;
    }

    // middle_block is two halves again.  channel_block: far history, (delay estimator,) aligned far end, energies / VAD, step
    // size, channel update (:466-511) -- the part the NEXT block's echo estimate waits for.  gain_block: suppression gain, Wiener
    // gains, NLP, gain product, comfort noise (:514-705), which depend on the channel half through GainInput alone and otherwise
    // only on state of their own (echo / near filters, noise estimate, counters, supGain, the seed): the pipelined kernel's
    // smallest shapes run them in different waves, one block apart.  Both halves work with the block's Q domains: track_q first.
    struct GainInput {
        vi echo_est;          // bins 0..63 (after a possible StoreAdaptiveChannel)
        int echo_est64, far_q;
        int cur_vad, near0, stored0;      // calc_suppression_gain's inputs
    };
    static AECM_HD void track_q(Uniform &u, const Spectrum &df, const Spectrum &cf) {
        u.dfa_noisy_q_old = u.dfa_noisy_q;
        u.dfa_noisy_q = df.q;
        if (kHasClean) {                                                              // :449-464
            u.dfa_clean_q_old = u.dfa_clean_q;
            u.dfa_clean_q = cf.q;
        } else {
            u.dfa_clean_q_old = u.dfa_noisy_q_old;
            u.dfa_clean_q = u.dfa_noisy_q;
        }
    }
    // kDelayGiven: the block's delay estimate (delay_block's result) comes from the caller -- the pipelined kernel's delay waves --
    // and with it, for a delay other than 0, the far-end magnitudes of the aligned block (far_given: what the history row holds).
    template <bool kDelayGiven = false>
    static AECM_HD TailInput middle_block(Regs &r, uint16_t *hist, const Spectrum &xf, const Spectrum &df, const Spectrum &cf, int delay_given = 0,
                                          vi far_given = vi(0)) {
        W::template phase_priority<3>(r.u.prio_drop);
        track_q(r.u, df, cf);
        const GainInput g = channel_block<kDelayGiven>(r, hist, xf, df, delay_given, far_given);
        return gain_block(r, df, cf, g);
    }
    // Of xf only mag / mag64 / q are read, of df mag / mag64 (and, without delay_given, q).
    template <bool kDelayGiven = false>
    static AECM_HD GainInput channel_block(Regs &r, uint16_t *hist, const Spectrum &xf, const Spectrum &df, int delay_given = 0, vi far_given = vi(0)) {
        Uniform &u = r.u;

        // UpdateFarHistory (aecm_core.cc:125-138)
        u.hist_pos = u.hist_pos + 1;
        if (u.hist_pos >= kHistory) u.hist_pos = 0;
        W::store_u16(hist + u.hist_pos * kLanes, r.lane, xf.mag);
        {
            int side = zext16(xf.mag64) | shl(xf.q, 16);
            if (u.hist_pos < 64) r.hq0 = W::writelane(r.hq0, side, u.hist_pos);
            else r.hq1 = W::writelane(r.hq1, side, u.hist_pos - 64);
        }

        // far and near binary spectra, the far word -> history, near binary spectrum -> delay
        int delay = delay_given;
        if constexpr (!kDelayGiven) delay = delay_block(r, xf, df);
        delay = effective_delay(u, delay);

        AECM_PHASE_MARK(4, r.m01, r.mean);
        W::template phase_priority<5>(r.u.prio_drop);
        // AlignedFarend (aecm_core.cc:157-172)
        const int pos = aligned_slot(u.hist_pos, delay);
        int side = pos < 64 ? W::readlane(r.hq0, pos) : W::readlane(r.hq1, pos - 64);
        const int far_q = sar(side, 16);
        const int far64 = zext16(side);
        vi far = xf.mag;
        if (AECM_STEADY_ALWAYS(delay != 0)) far = kDelayGiven ? far_given : W::load_u16(hist + pos * kLanes, r.lane);

        vi echo_est;
        int echo_est64;
        AECM_PHASE_MARK(5, far, r.m01);
        W::template phase_priority<6>(r.u.prio_drop);
        calc_energies(r, far, far64, far_q, df.mag, df.mag64, echo_est, echo_est64);  // :498
        const int mu = calc_step_size(u);                                             // :503
        u.tot_count = add(u.tot_count, 1);                                            // :506
        AECM_PHASE_MARK(6, echo_est, r.near_log);
        W::template phase_priority<7>(r.u.prio_drop);
        update_channel(r, far, far64, far_q, df.mag, df.mag64, mu, echo_est, echo_est64);   // :511
        GainInput g;
        g.echo_est = echo_est; g.echo_est64 = echo_est64; g.far_q = far_q;
        g.cur_vad = u.cur_vad;
        g.near0 = W::readlane(r.near_log, u.log_pos);
        g.stored0 = W::readlane(r.stored_log, u.log_pos);
        AECM_PHASE_MARK(7, r.b.ch_adapt32, g.echo_est);
        return g;
    }
    // r: the Q domains (track_q), echo_filt / near_filt / noise_est / the two counters of every bin, sup_gain(_old), noise_ctr, seed.
    static AECM_HD TailInput gain_block(Regs &r, const Spectrum &df, const Spectrum &cf, const GainInput &g) {
        Uniform &u = r.u;
        const Spectrum &clean = kHasClean ? cf : df;   // "dfw"/"ptrDfaClean" of the reference (T30)
        const vi echo_est = g.echo_est;
        const int echo_est64 = g.echo_est64, far_q = g.far_q;
        W::template phase_priority<8>(r.u.prio_drop);
        const int sup_gain = calc_suppression_gain(u, g.cur_vad, g.near0, g.stored0);  // :514

        vi e;                            // the suppressed spectrum of bins 0..63, packed re | im << 16
        int e_re64, e_im64 = 0;
        const bool q_steady = kNearFiltSteadyPath && u.dfa_clean_q <= u.dfa_clean_q_old;     // see near_filt_update
        if (AECM_STEADY_NEVER(kGainZeroPath && sup_gain == 0)) {
            // No suppression at all: supGain has decayed to 0 (the far end has been silent for a while; 31 % of the blocks of
            // the bench signal, instrumented oracle).  Every gain is ONE_Q14 (see wiener_bin), which its square >> 14, the
            // average of the preferred band, the NLP thresholds (65 non-zero gains) and the rounded product with dfw
            // (:618-686) all leave as it is, and the comfort noise has amplitude 0 (see noise_bin): the spectrum passes through,
            // the filters, the noise estimator and the random seed move as in any other block.
            if (AECM_LIKELY(q_steady)) {
                wiener_bin<vi, false, true, true>(r.b, echo_est, clean.mag, 0, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
                wiener_bin<int, true, true, true>(r.b64, echo_est64, clean.mag64, 0, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
            } else {
                wiener_bin<vi, false, false, true>(r.b, echo_est, clean.mag, 0, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
                wiener_bin<int, true, false, true>(r.b64, echo_est64, clean.mag64, 0, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
            }
            e = pack(clean.re, clean.im); e_re64 = clean.re64;
            if (AECM_STEADY_ALWAYS(W::per_block(u.cng) == 1))
                comfort_noise<true>(r, clean, vi(kOneQ14), kOneQ14, e, e_re64, e_im64);
        } else {
        vi hnl;
        int hnl64;
        if (AECM_STEADY_ALWAYS(AECM_LIKELY(q_steady))) {
            hnl = wiener_bin<vi, false, true>(r.b, echo_est, clean.mag, sup_gain, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
            hnl64 = wiener_bin<int, true, true>(r.b64, echo_est64, clean.mag64, sup_gain, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
        } else {
            hnl = wiener_bin<vi>(r.b, echo_est, clean.mag, sup_gain, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
            hnl64 = wiener_bin<int, true>(r.b64, echo_est64, clean.mag64, sup_gain, u.dfa_clean_q, u.dfa_clean_q_old, far_q);
        }
        const int num_pos = (int)__builtin_popcountll(W::ballot(hnl != 0)) + (hnl64 != 0 ? 1 : 0);   // :612-614

        AECM_PHASE_MARK(8, hnl, r.b.near_filt);
        W::template phase_priority<9>(r.u.prio_drop);
        if (AECM_STEADY_ALWAYS(W::per_block(u.mult) == 2)) {                          // :618-648
            hnl = as_i16(sar(mul24(hnl, hnl), 14));
            hnl64 = sext16(sar(mul(hnl64, hnl64), 14));
            int avg = W::reduce_add(hnl & lane_const<LC_NLP_AVG_BAND>(r));          // bins 4..24
            avg = sext16(divi(avg, 21));
            // bins 24..63 are clamped to the average: 0 <= avg < 2^15, so for the bins below 24 avg | 0x7fff0000 is above every gain
            hnl = imin(hnl, vi(avg) | lane_const<LC_NLP_LOW_BINS>(r));
            if (hnl64 > avg) hnl64 = avg;
        }
        if (AECM_STEADY_ALWAYS(W::per_block(u.nlp))) {                                // :651-686
            // hnl <= ONE_Q14 == NLP_COMP_HIGH always (a Wiener gain, its square >> 14, or a clamp to their average), so the
            // reference's "hnl > NLP_COMP_HIGH -> ONE_Q14" (:655-657) can never fire: only the lower threshold is live
            hnl = sel(as_i16(hnl) < kNlpCompLow, vi(0), hnl);
            hnl64 = hnl64 < kNlpCompLow ? 0 : hnl64;
            if (num_pos < 3) { hnl = vi(0); hnl64 = 0; }
        }
        // (the gain as a value the optimiser cannot see through: knowing that the 24-bit multiplies below only look at its
        // low 24 bits, it otherwise carries the gain shifted left by 8 through the NLP's selects and shifts it back per use)
        hnl = opaque_v(hnl);
        // :680-685  efw = (dfw * hnl + 8192) >> 14 (|dfw| <= 2^15, 0 <= hnl <= 2^14: an int16).  With the gain times 4 the wanted 16 bits
        // are the upper half of dfw * (hnl << 2) + 32768 (|.| <= 2^31, exact in 32 bits), and one byte permute packs both parts
        {
            const vi hnl4 = shl(hnl, 2);
            e = pack_hi16(add(mul24(clean.re, hnl4), 32768), add(mul24(clean.im, hnl4), 32768));
        }
        e_re64 = sext16(sar(mul(clean.re64, hnl64) + 8192, 14));

        AECM_PHASE_MARK(9, e, hnl);
        W::template phase_priority<10>(r.u.prio_drop);
        if (AECM_STEADY_ALWAYS(W::per_block(u.cng) == 1))                             // :702-705
            comfort_noise<false>(r, clean, hnl, hnl64, e, e_re64, e_im64);
        }

        AECM_PHASE_MARK(10, e, r.b.noise_est);
        W::template phase_priority<11>(r.u.prio_drop);
        // InverseFFTAndWindow (:193-246) + RealInverseFFT (real_fft.c:74-102):
        // Y[c] = (re[c], -im[c]) for c <= 64, conj-symmetric extension for c > 64 (T7).  The negated imaginary part (int16 wrap)
        // is the upper half times 0xffff modulo 2^16
        vi y = pk_mul_lo_u16(e, vi((int)0xffff0001));
        vi mirrored = W::bpermute(e, (vi(64) - r.lane) & 63);                          // lane t <- bin 64-t
        int y64 = zext16(e_re64) | shl(sext16(neg(e_im64)), 16);
        TailInput t;
        t.a = y;
        t.b = sel(r.lane == 0, vi(y64), mirrored);
        t.clean_q = u.dfa_clean_q;
        return t;
    }

    // r: the overlap buffer (out_ovl) and the lane constants (lane, table_index, k_p) are used.
    static AECM_HD vi tail_block(Regs &r, vi a, vi b, int clean_q) {
        const int out_cfft = fft128<true, false>(a, b, r.k_p);
        AECM_PHASE_MARK(11, a, b);
        W::template phase_priority<12>(r.u.prio_drop);
        const int sh = out_cfft - clean_q;
        // lane t holds y[bitrev6(t)] (a) and y[bitrev6(t)+64] (b): real parts only, in the UPPER halves (fft_stage_generic: last_real)
        vi first = as_i16(sar(mad16_hi_uc(a, lane_const<LC_HANN_SYN_LO>(r), 8192), 14));                // :219-221
        vi out = sat16(add(shift_i31(first, vi(sh)), r.out_ovl));                     // :222-227; |sh| <= 14
        vi second = sar(mad16_hi_uc(b, lane_const<LC_HANN_SYN_HI>(r), 0), 14);                          // :229-234
        r.out_ovl = sat16(shift_i31(second, vi(sh)));
        AECM_PHASE_MARK(12, out, r.out_ovl);
        W::template phase_priority<13>(r.u.prio_drop);
        return out;
    }

    // ------------------------------------------------------------------------------------------
    // A launch: n_blocks consecutive blocks of one stream, state in registers throughout.  Io says where
    // block b's 64 new samples come from and where its 64 output samples go:
    //   vi far(const Regs &, int b), near(...), clean(...)   lane t -> sample t of block b
    //   void out(const Regs &, int b, vi v)                  lane t holds output sample bitrev6(t) (= r.brev)
    //   void ready()                                         called once, after the state loads have been issued and
    //                                                        before the first sample fetch (a place to wait for the caller's own stores)
    // ------------------------------------------------------------------------------------------
    template <class Io>
    static AECM_HD void run_stream_io(const StatePtrs &st, Io &io, int64_t stream, int n_blocks) {
        Regs r;
        init_lane_constants(r, st.consts);
        uint32_t *vec = st.vec + stream * (int64_t)kVecWordsPerStream;
        int32_t *scal = st.scal + stream * (int64_t)kNumScal;
        uint16_t *hist = st.hist + stream * (int64_t)kHistWordsPerStream;
        load_state(r, vec, scal);
        io.ready();
        vi far_next = io.far(r, 0);
        vi near_next = io.near(r, 0);
        vi clean_next = kHasClean ? io.clean(r, 0) : vi(0);
        W::begin_stream();
        auto step = [&](int blk) __attribute__((always_inline)) {
            vi far_cur = far_next, near_cur = near_next, clean_cur = clean_next;
            if (blk + 1 < n_blocks) {             // prefetch the next block's 3 x 128 bytes
                far_next = io.far(r, blk + 1);
                near_next = io.near(r, blk + 1);
                if (kHasClean) clean_next = io.clean(r, blk + 1);
            }
            W::begin_block(blk, n_blocks);
            vi out = process_block(r, hist, far_cur, near_cur, clean_cur);
            io.out(r, blk, out);
            AECM_PHASE_MARK(13, r.out_ovl, r.x_old);
        };
#if defined(AECM_BLOCK_LOOP_UNROLL2)
        // two blocks per trip: the values a block hands to the next one (prefetched samples, the lane vectors a DPP shift
        // rebuilds in a fresh register) change registers instead of being copied back at the loop's end
        int blk = 0;
        for (; blk + 1 < n_blocks; blk += 2) {
            step(blk);
            step(blk + 1);
        }
        if (blk < n_blocks) step(blk);
#else
        for (int blk = 0; blk < n_blocks; ++blk) step(blk);
#endif
        AECM_PHASE_MARK(14, r.out_ovl, r.x_old);
        store_state(r, vec, scal);
    }

    // The same with the state loads already issued by the caller (init_lane_constants + load_state into r: the tick kernel issues
    // them ahead of its table fill).  A copy of the text above rather than a call from it: the block kernels' code is kept
    // instruction for instruction what the round's profiles were taken on.
    template <class Io>
    static AECM_HD void run_stream_loaded(Regs &r, const StatePtrs &st, Io &io, int64_t stream, int n_blocks) {
        uint32_t *vec = st.vec + stream * (int64_t)kVecWordsPerStream;
        int32_t *scal = st.scal + stream * (int64_t)kNumScal;
        uint16_t *hist = st.hist + stream * (int64_t)kHistWordsPerStream;
        io.ready();
        vi far_next = io.far(r, 0);
        vi near_next = io.near(r, 0);
        vi clean_next = kHasClean ? io.clean(r, 0) : vi(0);
        W::begin_stream();
        auto step = [&](int blk) __attribute__((always_inline)) {
            vi far_cur = far_next, near_cur = near_next, clean_cur = clean_next;
            if (blk + 1 < n_blocks) {             // prefetch the next block's 3 x 128 bytes
                far_next = io.far(r, blk + 1);
                near_next = io.near(r, blk + 1);
                if (kHasClean) clean_next = io.clean(r, blk + 1);
            }
            W::begin_block(blk, n_blocks);
            vi out = process_block(r, hist, far_cur, near_cur, clean_cur);
            io.out(r, blk, out);
            AECM_PHASE_MARK(13, r.out_ovl, r.x_old);
        };
#if defined(AECM_BLOCK_LOOP_UNROLL2)
        // two blocks per trip: the values a block hands to the next one (prefetched samples, the lane vectors a DPP shift
        // rebuilds in a fresh register) change registers instead of being copied back at the loop's end
        int blk = 0;
        for (; blk + 1 < n_blocks; blk += 2) {
            step(blk);
            step(blk + 1);
        }
        if (blk < n_blocks) step(blk);
#else
        for (int blk = 0; blk < n_blocks; ++blk) step(blk);
#endif
        AECM_PHASE_MARK(14, r.out_ovl, r.x_old);
        store_state(r, vec, scal);
    }

    // The strided audio view of the batch interface (aecm_state.h: IoView).
    struct StridedIo {
        const IoView &v;
        int64_t base;
        AECM_HD vi far(const Regs &r, int b) const { return W::load_i16(v.far + base + (int64_t)b * v.block_stride, r.lane); }
        AECM_HD vi near(const Regs &r, int b) const { return W::load_i16(v.near + base + (int64_t)b * v.block_stride, r.lane); }
        AECM_HD vi clean(const Regs &r, int b) const { return W::load_i16(v.near_clean + base + (int64_t)b * v.block_stride, r.lane); }
        AECM_HD void out(const Regs &r, int b, vi val) const { W::store_i16(v.out + base + (int64_t)b * v.block_stride, r.brev, val); }
        AECM_HD void ready() const {}
    };
    static AECM_HD void run_stream(const StatePtrs &st, const IoView &io, int64_t stream, int n_blocks) {
        StridedIo sio{io, stream * io.stream_stride};
        run_stream_io(st, sio, stream, n_blocks);
    }
};

}  // namespace aecm
#endif  // AECM_AMD_WAVE_H_
