// TEST INFRASTRUCTURE -- a 64-lane CPU simulator of the "wave policy" that
// webrtc_aecm_amd/csrc/aecm_wave.h is written against.  It lets the CPU-only test suite run the very
// same block-DSP source that the HIP kernel instantiates and compare it with the oracle.  It is never
// built into, linked with or called by the product library.
#ifndef AECM_TESTS_WAVE_SIM_H_
#define AECM_TESTS_WAVE_SIM_H_

#include <stdint.h>

#include <cmath>

#include "aecm_ops.h"
#include "aecm_tables.h"

namespace aecm {

struct VecB {
    bool v[64];
};
struct VecI {
    int v[64];
    VecI() { for (int i = 0; i < 64; ++i) v[i] = 0; }
    VecI(int s) { for (int i = 0; i < 64; ++i) v[i] = s; }   // NOLINT: implicit broadcast on purpose
};

#define SIM_BIN(NAME, EXPR)                                                   \
    inline VecI NAME(const VecI &a, const VecI &b) {                          \
        VecI r;                                                               \
        for (int i = 0; i < 64; ++i) { int x = a.v[i], y = b.v[i]; r.v[i] = (EXPR); } \
        return r;                                                             \
    }
SIM_BIN(operator+, add(x, y))
SIM_BIN(operator-, sub(x, y))
SIM_BIN(operator*, mul(x, y))
SIM_BIN(operator&, x & y)
SIM_BIN(operator|, x | y)
SIM_BIN(operator^, x ^ y)
SIM_BIN(operator<<, shl(x, y))
SIM_BIN(operator>>, sar(x, y))
SIM_BIN(shl, shl(x, y))
SIM_BIN(sar, sar(x, y))
SIM_BIN(lsr, lsr(x, y))
SIM_BIN(mul, mul(x, y))
SIM_BIN(mul24, mul24(x, y))
SIM_BIN(mulhi_u32, mulhi_u32(x, y))
SIM_BIN(mulhi_i32, mulhi_i32(x, y))
SIM_BIN(add, add(x, y))
SIM_BIN(sub, sub(x, y))
SIM_BIN(imin, imin(x, y))
SIM_BIN(imax, imax(x, y))
SIM_BIN(min_u32, min_u32(x, y))
SIM_BIN(shift_u31, shift_u31(x, y))
SIM_BIN(shift_i31, shift_i31(x, y))
SIM_BIN(divi, divi(x, y))
SIM_BIN(divu, divu(x, y))
SIM_BIN(pack_hi16, pack_hi16(x, y))
SIM_BIN(pk_max_i16, pk_max_i16(x, y))
SIM_BIN(pk_add_i16, pk_add_i16(x, y))
SIM_BIN(pk_sub_i16, pk_sub_i16(x, y))
SIM_BIN(pk_mul_lo_u16, pk_mul_lo_u16(x, y))
SIM_BIN(pk_shl_b16, pk_shl_b16(x, y))
SIM_BIN(pk_lshr_b16, pk_lshr_b16(x, y))
SIM_BIN(pk_ashr_i16, pk_ashr_i16(x, y))
SIM_BIN(pk_min_u16, pk_min_u16(x, y))
SIM_BIN(pk_add_u16, pk_add_u16(x, y))
SIM_BIN(pk_add_sat_i16, pk_add_sat_i16(x, y))
SIM_BIN(pk_max_u16, pk_max_u16(x, y))
SIM_BIN(pk_sub_sat_u16, pk_sub_sat_u16(x, y))
SIM_BIN(dot2_i16_c0, dot2_i16_c0(x, y))
SIM_BIN(dot2_i16_cm1, dot2_i16_cm1(x, y))
#undef SIM_BIN

#define SIM_UN(NAME, EXPR)                                                    \
    inline VecI NAME(const VecI &a) {                                         \
        VecI r;                                                               \
        for (int i = 0; i < 64; ++i) { int x = a.v[i]; r.v[i] = (EXPR); }     \
        return r;                                                             \
    }
SIM_UN(operator~, ~x)
SIM_UN(neg, neg(x))
SIM_UN(sext16, sext16(x))
SIM_UN(as_i16, as_i16(x))
SIM_UN(as_nonneg, as_nonneg(x))
SIM_UN(zext16, zext16(x))
SIM_UN(iabs, iabs(x))
SIM_UN(clz32, clz32(x))
SIM_UN(ffbh_i, ffbh_i(x))
SIM_UN(low_mask, low_mask(x))
SIM_UN(popc, popc(x))
SIM_UN(pk_abs_sat_i16, pk_abs_sat_i16(x))
SIM_UN(pk_neg_i16, pk_neg_i16(x))
SIM_UN(opaque_v, opaque_v(x))
SIM_UN(pk_nonzero_u16, pk_nonzero_u16(x))
SIM_UN(max_halves_i16, max_halves_i16(x))
#undef SIM_UN

#define SIM_CMP(NAME, EXPR)                                                   \
    inline VecB NAME(const VecI &a, const VecI &b) {                          \
        VecB r;                                                               \
        for (int i = 0; i < 64; ++i) { int x = a.v[i], y = b.v[i]; r.v[i] = (EXPR); } \
        return r;                                                             \
    }
SIM_CMP(operator==, x == y)
SIM_CMP(operator!=, x != y)
SIM_CMP(operator<, x < y)
SIM_CMP(operator<=, x <= y)
SIM_CMP(operator>, x > y)
SIM_CMP(operator>=, x >= y)
SIM_CMP(ltu, ltu(x, y))
SIM_CMP(gtu, gtu(x, y))
#undef SIM_CMP

#define SIM_MASK(NAME, EXPR)                                                  \
    inline VecB NAME(const VecB &a, const VecB &b) {                          \
        VecB r;                                                               \
        for (int i = 0; i < 64; ++i) { bool x = a.v[i], y = b.v[i]; r.v[i] = (EXPR); } \
        return r;                                                             \
    }
SIM_MASK(operator&, x && y)
SIM_MASK(operator|, x || y)
SIM_MASK(operator==, x == y)
SIM_MASK(operator!=, x != y)
#undef SIM_MASK
inline VecI shl_add(const VecI &a, int n, int c) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = shl_add(a.v[i], n, c);
    return r;
}
inline VecI shl_add(const VecI &a, int n, const VecI &c) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = shl_add(a.v[i], n, c.v[i]);
    return r;
}
#define SIM_MAD16(NAME)                                                       \
    inline VecI NAME(const VecI &a, const VecI &k, const VecI &c) {           \
        VecI r;                                                               \
        for (int i = 0; i < 64; ++i) r.v[i] = NAME(a.v[i], k.v[i], c.v[i]);   \
        return r;                                                             \
    }                                                                         \
    inline VecI NAME(const VecI &a, const VecI &k, int c) { return NAME(a, k, VecI(c)); }
inline VecI dot2_i16_uc(const VecI &a, const VecI &b, int c) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = dot2_i16_uc(a.v[i], b.v[i], c);
    return r;
}
SIM_MAD16(mad16_lo)
SIM_MAD16(mad16_hi)
SIM_MAD16(mad16_lo_uc)
SIM_MAD16(mad16_hi_uc)
#undef SIM_MAD16
inline VecI pk_mad_u16(const VecI &a, const VecI &b, const VecI &c) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = pk_mad_u16(a.v[i], b.v[i], c.v[i]);
    return r;
}
inline VecI dot2_i16(const VecI &a, const VecI &b, const VecI &c) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = dot2_i16(a.v[i], b.v[i], c.v[i]);
    return r;
}
inline VecB operator!(const VecB &a) { VecB r; for (int i = 0; i < 64; ++i) r.v[i] = !a.v[i]; return r; }
inline VecB operator&(const VecB &a, bool b) { VecB r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] && b; return r; }
inline VecB operator&(bool b, const VecB &a) { return a & b; }

inline VecI sel(const VecB &c, const VecI &a, const VecI &b) {
    VecI r;
    for (int i = 0; i < 64; ++i) r.v[i] = c.v[i] ? a.v[i] : b.v[i];
    return r;
}
inline VecI sel(bool c, const VecI &a, const VecI &b) { return c ? a : b; }

struct SimTables {
    static int hann(int i) { return kAecmSqrtHanningQ14[i]; }
};

// The policy itself.  Cross-lane primitives are written from their mathematical definition, not
// from any GPU instruction, so the simulator is an independent statement of what each one means.
struct SimWave {
    using vi = VecI;
    using vb = VecB;
    static constexpr bool kPrecomputedConstants = false;   // the simulator evaluates the definitions
    static constexpr bool kLaneConstsInTable = false;
    static constexpr bool kTight = false;
    static constexpr bool kPhasePriority = false;
    template <int ROW> static vi table_lane_const(const vi &) { return vi(0); }   // never used (kLaneConstsInTable == false)
    static vi table_index_for_this_block() { return vi(0); }
    static void begin_stream() {}
    static void begin_block(int, int) {}
    template <int PHASE> static void phase_priority(int = 0) {}
    static int pin_uniform(int x) { return x; }                            // device: a uniform value pinned to a scalar register
    static int per_block(int x) { return x; }                              // device: keeps launch-invariant conditions in the loop                                   // device: issue-priority rotation

    static vi lane_id() { vi r; for (int i = 0; i < 64; ++i) r.v[i] = i; return r; }
    static vi stream_lane_id() { return lane_id(); }
    static bool is_first_lane() { return true; }
    static void div_magic_lanes(const vi &d, vi &magic, vi &shift) {
        for (int i = 0; i < 64; ++i) div_magic(d.v[i], &magic.v[i], &shift.v[i]);
    }
    static int uni(int x) { return x; }   // device: v_readfirstlane (value is wave-uniform)

    static vi lut(const int16_t *t, int n, const vi &idx) {
        vi r;
        for (int i = 0; i < 64; ++i) r.v[i] = t[((idx.v[i] % n) + n) % n];
        return r;
    }
    static vi hann(const vi &i) { return lut(kAecmSqrtHanningQ14, 65, i); }
    // Packed twiddles of stage S for this lane's butterfly: w_re = (wr, -wi), w_im = (wi, wr) with
    // wr = cos, wi = -sin (forward) / +sin (inverse) of entry m << k, m = position & (2^S - 1),
    // k = 9 - S of kSinTable1024 (complex_fft.c:296-303, 412-420); positions are bit-reversed lanes.
    template <int S, bool kInverse>
    static void twiddles(vi &w_re, vi &w_im) {
        for (int t = 0; t < 64; ++t) {
            int brev = 0;
            for (int b = 0; b < 6; ++b) brev |= ((t >> b) & 1) << (5 - b);
            const int idx = (brev & ((1 << S) - 1)) << (6 - S);
            const int wr = kAecmTwiddleCosQ15[idx];
            const int wi = kInverse ? kAecmTwiddleSinQ15[idx] : -kAecmTwiddleSinQ15[idx];
            w_re.v[t] = (wr & 0xffff) | (int)((unsigned)(-wi) << 16);
            w_im.v[t] = (wi & 0xffff) | (int)((unsigned)wr << 16);
        }
    }
    // Inverse stages: the twiddles and their per-half negations.
    template <int S>
    static void inv_twiddles(vi &w_re, vi &w_im, vi &nw_re, vi &nw_im) {
        twiddles<S, true>(w_re, w_im);
        for (int t = 0; t < 64; ++t) {
            nw_re.v[t] = ((-sext16(w_re.v[t])) & 0xffff) | (int)((unsigned)(-(w_re.v[t] >> 16)) << 16);
            nw_im.v[t] = ((-sext16(w_im.v[t])) & 0xffff) | (int)((unsigned)(-(w_im.v[t] >> 16)) << 16);
        }
    }
    static vi opaque_const(int k) { return vi(k); }        // device: a constant pinned in a VGPR
    // Forward stages 1..6 in multiply-add form: the twiddles, their negations, and (even stages) the
    // accumulator offsets s and 1 - s, s = sum of the halves of the packed twiddle (see fft128).
    template <int S>
    static void fwd_twiddles(vi &w_re, vi &w_im, vi &nw_re, vi &nw_im) {
        twiddles<S, false>(w_re, w_im);
        for (int t = 0; t < 64; ++t) {
            nw_re.v[t] = ((-sext16(w_re.v[t])) & 0xffff) | (int)((unsigned)(-(w_re.v[t] >> 16)) << 16);
            nw_im.v[t] = ((-sext16(w_im.v[t])) & 0xffff) | (int)((unsigned)(-(w_im.v[t] >> 16)) << 16);
        }
    }
    template <int S>
    static void fwd_offsets(vi &s_re, vi &one_minus_s_re, vi &s_im, vi &one_minus_s_im) {
        vi w_re, w_im;
        twiddles<S, false>(w_re, w_im);
        for (int t = 0; t < 64; ++t) {
            s_re.v[t] = sext16(w_re.v[t]) + (w_re.v[t] >> 16);
            s_im.v[t] = sext16(w_im.v[t]) + (w_im.v[t] >> 16);
            one_minus_s_re.v[t] = 1 - s_re.v[t];
            one_minus_s_im.v[t] = 1 - s_im.v[t];
        }
    }
    static vi cos360(const vi &i) { return lut(kAecmCosQ13, 360, i); }
    static vi sin360(const vi &i) { return lut(kAecmSinQ13, 360, i); }
    static int cos360(int i) { return kAecmCosQ13[i]; }
    static int sin360(int i) { return kAecmSinQ13[i]; }

    // Re-pair FFT operands across lane bit Q: lanes with bit Q clear keep a and receive the
    // partner's a into b; lanes with bit Q set keep b and receive the partner's b into a.
    template <int Q>
    static void exchange(vi &a, vi &b) {
        vi na = a, nb = b;
        for (int i = 0; i < 64; ++i) {
            int p = i ^ (1 << Q);
            if ((i >> Q) & 1) na.v[i] = b.v[p];
            else nb.v[i] = a.v[p];
        }
        a = na;
        b = nb;
    }
    template <int Q, int N>
    static void exchange_all(vi (&aa)[N], vi (&bb)[N]) { for (int n = 0; n < N; ++n) exchange<Q>(aa[n], bb[n]); }
    static vi divu_u32_u16(const vi &n, const vi &d) { return divu(n, d); }    // device: two float steps (wave_gfx950.h)
    static int reduce_max(const vi &v) { int m = v.v[0]; for (int i = 1; i < 64; ++i) m = v.v[i] > m ? v.v[i] : m; return m; }
    static int reduce_min(const vi &v) { int m = v.v[0]; for (int i = 1; i < 64; ++i) m = v.v[i] < m ? v.v[i] : m; return m; }
    static int reduce_add(const vi &v) { int s = 0; for (int i = 0; i < 64; ++i) s = add(s, v.v[i]); return s; }
    // Several independent reductions at once (the GPU merges their butterfly steps).
    static void reduce_max2(const vi &x, const vi &y, int &rx, int &ry) { rx = reduce_max(x); ry = reduce_max(y); }
    static void reduce_min_max(const vi &x, const vi &y, int &min_x, int &max_y) { min_x = reduce_min(x); max_y = reduce_max(y); }
    static void reduce_add4(const vi &x, const vi &y, const vi &z, const vi &w, int &rx, int &ry, int &rz, int &rw) {
        rx = reduce_add(x); ry = reduce_add(y); rz = reduce_add(z); rw = reduce_add(w);
    }
    static uint64_t ballot(const vb &m) { uint64_t r = 0; for (int i = 0; i < 64; ++i) r |= (uint64_t)(m.v[i] ? 1 : 0) << i; return r; }
    static int readlane(const vi &v, int lane) { return v.v[lane & 63]; }
    static vi writelane(const vi &v, int value, int lane) { vi r = v; r.v[lane & 63] = value; return r; }
    static vi bpermute(const vi &v, const vi &src) { vi r; for (int i = 0; i < 64; ++i) r.v[i] = v.v[src.v[i] & 63]; return r; }
    static vi shift_up1(const vi &v, int fill) { vi r; r.v[0] = fill; for (int i = 1; i < 64; ++i) r.v[i] = v.v[i - 1]; return r; }
    static vi isqrt31(const vi &v) {
        vi r;
        for (int i = 0; i < 64; ++i) {
            uint32_t x = (uint32_t)v.v[i];
            uint32_t s = (uint32_t)std::sqrt((double)x);
            while ((uint64_t)s * s > x) --s;
            while ((uint64_t)(s + 1) * (s + 1) <= x) ++s;
            r.v[i] = (int)s;
        }
        return r;
    }
    static int isqrt31(int v) { return isqrt31(vi(v)).v[0]; }

    static vi load_u32(const uint32_t *p, const vi &idx) { vi r; for (int i = 0; i < 64; ++i) r.v[i] = (int)p[idx.v[i]]; return r; }
    static void store_u32(uint32_t *p, const vi &idx, const vi &val) { for (int i = 0; i < 64; ++i) p[idx.v[i]] = (uint32_t)val.v[i]; }
    static void store_u32_if(const VecB &m, uint32_t *p, const vi &idx, const vi &val) { for (int i = 0; i < 64; ++i) if (m.v[i]) p[idx.v[i]] = (uint32_t)val.v[i]; }
    static vi load_i16(const int16_t *p, const vi &idx) { vi r; for (int i = 0; i < 64; ++i) r.v[i] = p[idx.v[i]]; return r; }
    static vi load_u16(const uint16_t *p, const vi &idx) { vi r; for (int i = 0; i < 64; ++i) r.v[i] = p[idx.v[i]]; return r; }
    static void store_i16(int16_t *p, const vi &idx, const vi &val) { for (int i = 0; i < 64; ++i) p[idx.v[i]] = (int16_t)val.v[i]; }
    static void store_u16(uint16_t *p, const vi &idx, const vi &val) { for (int i = 0; i < 64; ++i) p[idx.v[i]] = (uint16_t)val.v[i]; }
    struct ScalarRow {
        const int32_t *p;
        int get(int f) const { return p[f]; }
    };
    static ScalarRow load_scalar_row(const int32_t *scal) { return ScalarRow{scal}; }
    static void store_scalar(int32_t *scal, int f, int v) { scal[f] = v; }
};

}  // namespace aecm
#endif  // AECM_TESTS_WAVE_SIM_H_
