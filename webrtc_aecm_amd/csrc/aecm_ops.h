// Scalar fixed-point primitives with fully defined semantics (two's complement wrap, 5-bit shift
// counts, arithmetic >> on signed) used by the wave-generic block DSP (aecm_wave.h).
//
// On the GPU a "lane vector" is simply an int held in a VGPR, so these scalar overloads ARE the
// vector operations; the CPU lane simulator used by the tests adds 64-wide overloads with the same
// names (tests/sim/wave_sim.h).  They restate the reference's SPL helpers:
//   norm_w32 / norm_u32 / norm_w16  -> aecm/spl_inl.h:97-111
//   add_sat32 / sat16               -> aecm/spl_inl.h:59-85
//   shift_i / shift_u               -> WEBRTC_SPL_SHIFT_W32, aecm/signal_processing_library.h:128
//   divi / divu                     -> WebRtcSpl_DivW32W16 / DivU32U16, aecm/signal_processing_library.cc:107-123
#ifndef AECM_AMD_OPS_H_
#define AECM_AMD_OPS_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AECM_HD __host__ __device__ __forceinline__
#else
#define AECM_HD inline
#endif

namespace aecm {

AECM_HD int shl(int a, int n) { return (int)((unsigned)a << (n & 31)); }
AECM_HD int sar(int a, int n) { return a >> (n & 31); }
AECM_HD int lsr(int a, int n) { return (int)((unsigned)a >> (n & 31)); }
AECM_HD int mul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
AECM_HD int add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
AECM_HD int sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
AECM_HD int neg(int a) { return (int)(0u - (unsigned)a); }
AECM_HD int sext16(int a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe(a, 0, 16);       // keep it one v_bfe_i32 (do not let shifts be re-associated around it)
#else
    return (int)(int16_t)a;
#endif
}
// (a << n) + c, wrapping                                                     -> v_lshl_add_u32
AECM_HD int shl_add(int a, int n, int c) { return add(shl(a, n), c); }
AECM_HD int zext16(int a) { return a & 0xffff; }
AECM_HD int sel(bool c, int a, int b) { return c ? a : b; }
AECM_HD int imin(int a, int b) { return a < b ? a : b; }
AECM_HD int imax(int a, int b) { return a > b ? a : b; }
AECM_HD int iabs(int a) { return a < 0 ? neg(a) : a; }
AECM_HD bool ltu(int a, int b) { return (unsigned)a < (unsigned)b; }
AECM_HD bool gtu(int a, int b) { return (unsigned)a > (unsigned)b; }
AECM_HD int clz32(int a) { return a == 0 ? 32 : __builtin_clz((unsigned)a); }
// Leading zeros of an operand the caller knows to be non-zero -- or whose count it then ignores (any value for 0).
AECM_HD int clz32_nz(int a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_clz((unsigned)a);            // no zero fix-up (s_flbit_i32_b32 / v_ffbh_u32 alone)
#else
    return a == 0 ? 32 : __builtin_clz((unsigned)a);
#endif
}
AECM_HD int popc(int a) { return __builtin_popcount((unsigned)a); }
// Truncating signed / unsigned division with the reference's divide-by-zero results.
AECM_HD int divi(int a, int b) {
    if (b == 0) return 0x7fffffff;
    if (b == -1) return neg(a);
    return a / b;
}
AECM_HD int divu(int a, int b) { return b == 0 ? -1 : (int)((unsigned)a / (unsigned)b); }

// Low 32 bits of a * b where BOTH operands are known to fit in 24 signed bits (-2^23 <= x < 2^23):
// full-rate v_mul_i32_i24 instead of the quarter-rate 32-bit multiply.  The host build checks the
// precondition (the CPU lane simulator runs every test input through it).
#if !defined(__HIP_DEVICE_COMPILE__)
[[noreturn]] void aecm_mul24_range_violation(int a, int b);
#elif defined(AECM_CHECKED)
// Audit build (-DAECM_CHECKED, libaecm_mi355x_checked.so; never the shipped kernel): the device counts violated
// preconditions instead of assuming them, and computes the exact result so that parity still holds.
// [0] mul24 operand outside 24 signed bits, [1] as_i16 argument outside int16.
extern __device__ unsigned long long g_aecm_check_fail[2];
#endif
AECM_HD int mul24(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AECM_CHECKED)
    if (a < -(1 << 23) || a >= (1 << 23) || b < -(1 << 23) || b >= (1 << 23)) {
        atomicAdd(&g_aecm_check_fail[0], 1ull);
        return mul(a, b);
    }
    return __mul24(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    if (a < -(1 << 23) || a >= (1 << 23) || b < -(1 << 23) || b >= (1 << 23)) aecm_mul24_range_violation(a, b);
    return mul(a, b);
#endif
}

// The reference narrows many intermediate values to int16_t.  Where the value provably already lies
// in [-32768, 32767] the narrowing is the identity and costs nothing here; the host build checks the
// claim (the CPU lane simulator runs every test input, including the full-scale fuzz, through it).
#if !defined(__HIP_DEVICE_COMPILE__)
[[noreturn]] void aecm_i16_range_violation(int v);
#endif
AECM_HD int as_i16(int v) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (v < -32768 || v > 32767) aecm_i16_range_violation(v);
#elif defined(AECM_CHECKED)
    if (v < -32768 || v > 32767) {
        atomicAdd(&g_aecm_check_fail[1], 1ull);
        return sext16(v);                                 // what the reference's (int16_t) cast does
    }
#endif
    return v;
}

// A value the surrounding arithmetic proves non-negative (as a signed 32-bit number); the host build checks the claim
// like as_i16, the audit build counts violations on the device (counter 1).
#if !defined(__HIP_DEVICE_COMPILE__)
[[noreturn]] void aecm_nonneg_violation(int v);
#endif
AECM_HD int as_nonneg(int v) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (v < 0) aecm_nonneg_violation(v);
#elif defined(AECM_CHECKED)
    if (v < 0) atomicAdd(&g_aecm_check_fail[1], 1ull);
#endif
    return v;
}

// ---- packed-int16 primitives (a 32-bit word holds lo | hi<<16) -------------------------------------
// Each has an exact portable definition; on gfx950 the same function is a single instruction.
// sext(a.lo)*sext(b.lo) + sext(a.hi)*sext(b.hi) + c  (wrapping)              -> v_dot2_i32_i16
AECM_HD int dot2_i16(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short aecm_short2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(aecm_short2, a), __builtin_bit_cast(aecm_short2, b), c, false);
#else
    return add(add(mul(sext16(a), sext16(b)), mul(sar(a, 16), sar(b, 16))), c);
#endif
}
// The same with the addend fixed to 0 / -1.  On the GPU this is the three-source VOP3P form with an inline constant
// (the compiler would otherwise emit v_mov + the accumulating VOP2 form v_dot2c: one instruction more per product).
#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int dot2_i16_c0(int a, int b) { int r; asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
AECM_HD int dot2_i16_cm1(int a, int b) { int r; asm("v_dot2_i32_i16 %0, %1, %2, -1" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
AECM_HD int dot2_i16_c0(int a, int b) { return dot2_i16(a, b, 0); }
AECM_HD int dot2_i16_cm1(int a, int b) { return dot2_i16(a, b, -1); }
#endif
// The same with a wave-uniform addend (an SGPR operand of the three-source form).
#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int dot2_i16_uc(int a, int b, int c) { int r; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }
#else
AECM_HD int dot2_i16_uc(int a, int b, int c) { return dot2_i16(a, b, c); }
#endif
// sext(a.lo) * sext(k.lo) + c resp. sext(a.hi) * sext(k.lo) + c  (wrapping)    -> v_mad_i32_i16 (op_sel picks the half)
// The _uc forms take a wave-uniform c (kept in an SGPR on the GPU).
#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int mad16_lo(int a, int k, int c) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
AECM_HD int mad16_hi(int a, int k, int c) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
AECM_HD int mad16_lo_uc(int a, int k, int c) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "s"(c)); return r; }
AECM_HD int mad16_hi_uc(int a, int k, int c) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(k), "s"(c)); return r; }
#else
AECM_HD int mad16_lo(int a, int k, int c) { return add(mul(sext16(a), sext16(k)), c); }
AECM_HD int mad16_hi(int a, int k, int c) { return add(mul(sar(a, 16), sext16(k)), c); }
AECM_HD int mad16_lo_uc(int a, int k, int c) { return mad16_lo(a, k, c); }
AECM_HD int mad16_hi_uc(int a, int k, int c) { return mad16_hi(a, k, c); }
#endif
// (upper half of yr) | (upper half of yi) << 16                              -> v_perm_b32
AECM_HD int pack_hi16(int yr, int yi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__builtin_amdgcn_perm((unsigned)yi, (unsigned)yr, 0x07060302u);
#else
    return (int)(((unsigned)yr >> 16) | ((unsigned)yi & 0xffff0000u));
#endif
}
// per half: |x| with |-32768| saturated to 32767                             -> v_pk_sub_i16 clamp + v_pk_max_i16
AECM_HD int pk_abs_sat_i16(int a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short aecm_short2 __attribute__((ext_vector_type(2)));
    aecm_short2 v = __builtin_bit_cast(aecm_short2, a);
    aecm_short2 n = __builtin_elementwise_sub_sat((aecm_short2){0, 0}, v);
    return __builtin_bit_cast(int, __builtin_elementwise_max(v, n));
#else
    int lo = sext16(a), hi = sar(a, 16);
    lo = lo < 0 ? (lo == -32768 ? 32767 : -lo) : lo;
    hi = hi < 0 ? (hi == -32768 ? 32767 : -hi) : hi;
    return (lo & 0xffff) | (int)((unsigned)hi << 16);
#endif
}
// per half: -x, wrapping (-(-32768) == -32768)                               -> v_pk_sub_i16
AECM_HD int pk_neg_i16(int a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short aecm_short2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, (aecm_short2)((aecm_short2){0, 0} - __builtin_bit_cast(aecm_short2, a)));
#else
    return (int)(((0u - (unsigned)a) & 0xffffu) | ((0u - ((unsigned)a & 0xffff0000u)) & 0xffff0000u));
#endif
}
// More per-half (packed 16-bit) arithmetic, each one gfx950 instruction: a + b, a - b (wrapping), a * b (low 16 bits),
// a << n / a >> n with per-half counts (n & 15), arithmetic a >> n, unsigned min.
#if defined(__HIP_DEVICE_COMPILE__)
#define AECM_PK_BINOP(NAME, T, EXPR)                                                         \
    AECM_HD int NAME(int a, int b) {                                                         \
        typedef T aecm_v2 __attribute__((ext_vector_type(2)));                               \
        const aecm_v2 x = __builtin_bit_cast(aecm_v2, a), y = __builtin_bit_cast(aecm_v2, b); \
        return __builtin_bit_cast(int, (aecm_v2)(EXPR));                                     \
    }
AECM_PK_BINOP(pk_add_i16, short, x + y)
AECM_PK_BINOP(pk_sub_i16, short, x - y)
AECM_PK_BINOP(pk_mul_lo_u16, unsigned short, x * y)
AECM_PK_BINOP(pk_shl_b16, unsigned short, x << y)          // counts are < 16 at every call site (the instruction takes them modulo 16)
AECM_PK_BINOP(pk_lshr_b16, unsigned short, x >> y)
AECM_PK_BINOP(pk_ashr_i16, short, x >> y)
AECM_PK_BINOP(pk_min_u16, unsigned short, __builtin_elementwise_min(x, y))
#undef AECM_PK_BINOP
// per half: a * b + c (low 16 bits)                                          -> v_pk_mad_u16
AECM_HD int pk_mad_u16(int a, int b, int c) {
    typedef unsigned short aecm_v2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, (aecm_v2)(__builtin_bit_cast(aecm_v2, a) * __builtin_bit_cast(aecm_v2, b) + __builtin_bit_cast(aecm_v2, c)));
}
#else
AECM_HD int pk_mad_u16(int a, int b, int c) {
    const unsigned lo = ((unsigned)a & 0xffffu) * ((unsigned)b & 0xffffu) + ((unsigned)c & 0xffffu);
    const unsigned hi = ((unsigned)a >> 16) * ((unsigned)b >> 16) + ((unsigned)c >> 16);
    return (int)((lo & 0xffffu) | (hi << 16));
}
#define AECM_PK_BINOP(NAME, LO, HI)                                                          \
    AECM_HD int NAME(int a, int b) {                                                         \
        const int al = sext16(a), ah = sar(a, 16), bl = sext16(b), bh = sar(b, 16);          \
        const unsigned ul = (unsigned)al & 0xffffu, uh = (unsigned)ah & 0xffffu, vl = (unsigned)bl & 0xffffu, vh = (unsigned)bh & 0xffffu; \
        (void)al; (void)ah; (void)bl; (void)bh; (void)ul; (void)uh; (void)vl; (void)vh;      \
        return (int)(((unsigned)(LO) & 0xffffu) | ((unsigned)(HI) << 16));                   \
    }
AECM_PK_BINOP(pk_add_i16, al + bl, ah + bh)
AECM_PK_BINOP(pk_sub_i16, al - bl, ah - bh)
AECM_PK_BINOP(pk_mul_lo_u16, ul * vl, uh * vh)
AECM_PK_BINOP(pk_shl_b16, ul << (vl & 15u), uh << (vh & 15u))
AECM_PK_BINOP(pk_lshr_b16, ul >> (vl & 15u), uh >> (vh & 15u))
AECM_PK_BINOP(pk_ashr_i16, al >> (bl & 15), ah >> (bh & 15))
AECM_PK_BINOP(pk_min_u16, ul < vl ? ul : vl, uh < vh ? uh : vh)
#undef AECM_PK_BINOP
#endif
// per half: 1 where the half is non-zero, else 0.  v_pk_min_u16 with the inline constant 1, as assembly: written as a
// minimum (or a comparison) the compiler turns it into per-half compares, selects and a re-pack -- five instructions.
#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int pk_nonzero_u16(int x) {
    int r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
    return r;
}
#else
AECM_HD int pk_nonzero_u16(int x) { return (((unsigned)x & 0xffffu) != 0 ? 1 : 0) | (((unsigned)x >> 16) != 0 ? 0x10000 : 0); }
#endif
// x, as a value the optimiser cannot see through (it otherwise rewrites packed arithmetic it recognises -- a multiply by a
// 0 / 1 factor -- into per-half compares, selects and a re-pack: six instructions for one)
#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int opaque_v(int x) { asm("" : "+v"(x)); return x; }
#else
AECM_HD int opaque_v(int x) { return x; }
#endif
// per half signed max                                                        -> v_pk_max_i16
AECM_HD int pk_max_i16(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short aecm_short2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(aecm_short2, a), __builtin_bit_cast(aecm_short2, b)));
#else
    int lo = imax(sext16(a), sext16(b)), hi = imax(sar(a, 16), sar(b, 16));
    return (lo & 0xffff) | (int)((unsigned)hi << 16);
#endif
}
// per half, unsigned: a + b wrapping / max / a - b saturating at 0          -> v_pk_add_u16 / v_pk_max_u16 / v_pk_sub_u16 clamp
AECM_HD int pk_add_u16(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short aecm_ushort2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, (aecm_ushort2)(__builtin_bit_cast(aecm_ushort2, a) + __builtin_bit_cast(aecm_ushort2, b)));
#else
    return (int)((((unsigned)a + (unsigned)b) & 0xffffu) | (((unsigned)a & 0xffff0000u) + ((unsigned)b & 0xffff0000u)));
#endif
}
AECM_HD int pk_max_u16(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short aecm_ushort2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(aecm_ushort2, a), __builtin_bit_cast(aecm_ushort2, b)));
#else
    const unsigned al = (unsigned)a & 0xffffu, bl = (unsigned)b & 0xffffu, ah = (unsigned)a >> 16, bh = (unsigned)b >> 16;
    return (int)((al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16));
#endif
}
AECM_HD int pk_sub_sat_u16(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short aecm_ushort2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, __builtin_elementwise_sub_sat(__builtin_bit_cast(aecm_ushort2, a), __builtin_bit_cast(aecm_ushort2, b)));
#else
    const unsigned al = (unsigned)a & 0xffffu, bl = (unsigned)b & 0xffffu, ah = (unsigned)a >> 16, bh = (unsigned)b >> 16;
    return (int)((al > bl ? al - bl : 0u) | ((ah > bh ? ah - bh : 0u) << 16));
#endif
}
// per half, signed: a + b saturating at +-2^15                                -> v_pk_add_i16 clamp
AECM_HD int pk_add_sat_i16(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short aecm_short2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(aecm_short2, a), __builtin_bit_cast(aecm_short2, b)));
#else
    int lo = sext16(a) + sext16(b), hi = sar(a, 16) + sar(b, 16);
    lo = lo > 32767 ? 32767 : lo < -32768 ? -32768 : lo;
    hi = hi > 32767 ? 32767 : hi < -32768 ? -32768 : hi;
    return (lo & 0xffff) | (int)((unsigned)hi << 16);
#endif
}
// max(sext(lo), sext(hi))
AECM_HD int max_halves_i16(int a) { return imax(sext16(a), sar(a, 16)); }

// ---- generic (scalar or lane-vector) helpers built on the overload set above --------------------
// WebRtcSpl_NormU32 / NormW32 / NormW16 (aecm/spl_inl.h:97-111): leading zeros of a (0 for a == 0) /
// redundant sign bits of a (0 for a == 0, 31 resp. 15 for a == -1).
// v_ffbh_i32 / s_flbit_i32 ("leading bits equal to the sign bit", -1 for 0 and -1) has no clang builtin in this
// toolchain; the LLVM intrinsic is reached through its assembler name.  The norms are built on its raw result:
//   norm_w32(a)    = a == 0 ? 0 : min_u32(ffbh_i(a) - 1, 31)        (-1 - 1 wraps to a huge unsigned -> 31 for a == -1)
//   norm_w16(a)    = the same - 16 for a in int16 range
//   norm_u32_nn(a) = max(ffbh_i(a), 0) for an operand known to be >= 0 as a signed number (two instructions)
#if defined(__HIP_DEVICE_COMPILE__)
extern "C" __device__ int aecm_llvm_amdgcn_sffbh(int) __asm("llvm.amdgcn.sffbh.i32");
AECM_HD int ffbh_i(int a) { return aecm_llvm_amdgcn_sffbh(a); }
AECM_HD int norm_u32(int a) { return clz32(a) & 31; }          // clz32(0) == 32: "& 31" is the a == 0 case
#else
AECM_HD int ffbh_i(int a) { return (a == 0 || a == -1) ? -1 : __builtin_clz((unsigned)(a < 0 ? ~a : a)); }
template <class I> AECM_HD I norm_u32(I a) { return sel(a == 0, I(0), clz32(a)); }
#endif
AECM_HD int min_u32(int a, int b) { return (unsigned)a < (unsigned)b ? a : b; }
template <class I> AECM_HD I norm_u32_nn(I a) { return imax(ffbh_i(as_nonneg(a)), I(0)); }
template <class I> AECM_HD I norm_w32(I a) { return sel(a == 0, I(0), min_u32(ffbh_i(a) - 1, I(31))); }
template <class I> AECM_HD I norm_w16(I a) { return sel(a == 0, I(0), min_u32(ffbh_i(a) - 1, I(31)) - 16); }
// norm_w32 / norm_w16 for an operand whose zero case the caller does not care about, or wants to read as "as many
// redundant sign bits as there can be" (result for 0: 31 resp. 15): no zero test
template <class I> AECM_HD I norm_w32_nz(I a) { return min_u32(ffbh_i(a) - 1, I(31)); }
template <class I> AECM_HD I norm_w16_nz(I a) { return min_u32(ffbh_i(a) - 1, I(31)) - 16; }
// High 32 bits of the signed 64-bit product                                  -> v_mul_hi_i32 / s_mul_hi_i32
AECM_HD int mulhi_i32(int a, int b) { return (int)(((int64_t)a * (int64_t)b) >> 32); }
// High 32 bits of the unsigned 64-bit product                                -> v_mul_hi_u32
AECM_HD int mulhi_u32(int a, int b) { return (int)(((uint64_t)(uint32_t)a * (uint64_t)(uint32_t)b) >> 32); }

// floor(n / d) for 0 <= n <= 2^31 and a small divisor 1 <= d <= 2^16 through a precomputed reciprocal: with
// L = ceil(log2 d) and the 33-bit M = ceil(2^(32+L) / d) in [2^32, 2^33),
//   floor(n / d) == floor(n * M / 2^(32+L)) == (n + mulhi_u32(n, M - 2^32)) >> L
// (Granlund-Montgomery with one more bit of reciprocal than the dividend has, exact for every 32-bit n; the sum cannot
// overflow: mulhi_u32(n, .) < n <= 2^31).  d == 1 is no special case: M - 2^32 == 0, L == 0.
AECM_HD void div_magic(int d, int *magic, int *shift) {
    const int L = d <= 1 ? 0 : 32 - clz32(d - 1);                // ceil(log2 d)
    const uint64_t num = (uint64_t)1 << (32 + L);
    const uint64_t m33 = (num + (uint64_t)d - 1) / (uint64_t)d;
    *magic = (int)(uint32_t)(m33 - ((uint64_t)1 << 32));
    *shift = L;
}
template <class I> AECM_HD I divu_by_magic(I n, I magic, I shift) { return lsr(add(n, mulhi_u32(n, magic)), shift); }

#if defined(__HIP_DEVICE_COMPILE__)
AECM_HD int add_sat32(int a, int b) { return __builtin_elementwise_add_sat(a, b); }   // v_add_i32 clamp
#else
template <class I> AECM_HD I add_sat32(I a, I b) {
    I s = add(a, b);
    auto ovf = ((a < 0) == (b < 0)) & ((a < 0) != (s < 0));
    return sel(ovf, sel(s < 0, I(0x7fffffff), I((int)0x80000000)), s);
}
#endif
template <class I> AECM_HD I sat16(I v) { return imax(imin(v, I(32767)), I(-32768)); }
// c >= 0: x * 2^c (wrapping); c < 0: arithmetic / logical right shift by -c.
template <class I, class C> AECM_HD I shift_i(I x, C c) { return sel(c >= 0, shl(x, c), sar(x, neg(c))); }
template <class I, class C> AECM_HD I shift_u(I x, C c) { return sel(c >= 0, shl(x, c), lsr(x, neg(c))); }
// The same for a count known to lie in [-31, 31] (the host build checks the claim; audit build: counter 1): the value
// is placed in the upper half of a 64-bit word and shifted right by 32 - c, one 64-bit shift (same issue cost as a
// 32-bit one on gfx950) instead of two shifts, a negation, a compare and a select.
#if !defined(__HIP_DEVICE_COMPILE__)
[[noreturn]] void aecm_shift_range_violation(int c);
#endif
AECM_HD int checked_shift31(int c) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (c < -31 || c > 31) aecm_shift_range_violation(c);
#elif defined(AECM_CHECKED)
    if (c < -31 || c > 31) atomicAdd(&g_aecm_check_fail[1], 1ull);
#endif
    return c;
}
AECM_HD int shift_u31(int x, int c) { return (int)(uint32_t)(((uint64_t)(uint32_t)x << 32) >> (32 - checked_shift31(c))); }
AECM_HD int shift_i31(int x, int c) { return (int)(uint32_t)(uint64_t)(((int64_t)x << 32) >> (32 - checked_shift31(c))); }

// new = mean + ((new - mean) >> factor) with the shift applied to the magnitude
// (WebRtc_MeanEstimatorFix, aecm/delay_estimator.cc:690-702).
// (1 << width) - 1                                                            -> v_bfm_b32 / s_bfm_b32
AECM_HD int low_mask(int width) { return (int)((1u << (width & 31)) - 1u); }
// The shift truncates toward zero: (diff + (diff < 0 ? 2^factor - 1 : 0)) >> factor.
template <class I, class C> AECM_HD I mean_step(I value, C factor, I mean) {
    I diff = sub(value, mean);
    return add(mean, sar(add(diff, sar(diff, 31) & low_mask(factor)), factor));
}

}  // namespace aecm
#endif  // AECM_AMD_OPS_H_
