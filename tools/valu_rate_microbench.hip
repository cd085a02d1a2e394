// Micro-benchmark behind DESIGN.md's VALU roofline: sustained wave64 issue cost (ns and shader cycles
// per instruction per SIMD) of the integer VALU instruction classes the AECM kernel is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_microbench.hip -o tools/valu_rate_microbench
// Every kernel executes ITERS x 512 copies of one instruction per wave, 8 independent registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define X8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)

#define KERNEL(NAME, ASM8)                                                                     \
    __global__ __launch_bounds__(256) void NAME(int *out, int iters) {                         \
        int a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 5, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 9, a7 = a0 + 13; \
        int b = blockIdx.x + 3, c = threadIdx.x & 7;                                           \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 64; ++r) { asm volatile(ASM8 OPS); }         \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;   \
    }

#define KERNEL_SCLOB(NAME, ASM8)                                                               \
    __global__ __launch_bounds__(256) void NAME(int *out, int iters) {                         \
        int a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 5, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 9, a7 = a0 + 13; \
        int b = blockIdx.x + 3, c = threadIdx.x & 7;                                           \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 64; ++r) { asm volatile(ASM8 OPS : "s20", "s21", "s22", "s23", "scc"); } \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;   \
    }

#define F_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define F_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define F_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define F_SHL(i) "v_lshlrev_b32 %" #i ", %9, %" #i "\n"
#define F_ASHR(i) "v_ashrrev_i32 %" #i ", %9, %" #i "\n"
#define F_MAX(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define F_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define F_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 0, 16\n"
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define F_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define F_MUL24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define F_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define F_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define F_DOT2C(i) "v_dot2c_i32_i16 %" #i ", %8, %9\n"
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define F_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define F_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define F_PKSUBC(i) "v_pk_sub_i16 %" #i ", %8, %" #i " clamp\n"
#define F_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define F_CNDSGPR(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define F_CMP(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define F_CMPS(i) "v_cmp_lt_i32_e64 s[20:21], %" #i ", %8\n"
#define F_READLANE(i) "v_readlane_b32 s20, %" #i ", 63\n"
#define F_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define F_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define F_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define F_ADDCLAMP(i) "v_add_i32 %" #i ", %" #i ", %8 clamp\n"
#define F_SDWA(i) "v_mul_i32_i24_sdwa %" #i ", sext(%" #i "), sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"

// alternating destination/source registers so that consecutive DPP ops are independent
#define DPP_ALT "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %4, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %6, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %3, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %5, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                "v_max_i32_dpp %7, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

KERNEL(k_add, X8(F_ADD))
KERNEL(k_sub, X8(F_SUB))
KERNEL(k_and, X8(F_AND))
KERNEL(k_xor, X8(F_XOR))
KERNEL(k_shl, X8(F_SHL))
KERNEL(k_ashr, X8(F_ASHR))
KERNEL(k_max, X8(F_MAX))
KERNEL(k_minu, X8(F_MINU))
KERNEL(k_bfe, X8(F_BFE))
KERNEL(k_lshladd, X8(F_LSHLADD))
KERNEL(k_add3, X8(F_ADD3))
KERNEL(k_andor, X8(F_ANDOR))
KERNEL(k_mul24, X8(F_MUL24))
KERNEL(k_mad24, X8(F_MAD24))
KERNEL(k_mullo, X8(F_MULLO))
KERNEL(k_mulhi, X8(F_MULHI))
KERNEL(k_dot2c, X8(F_DOT2C))
KERNEL(k_perm, X8(F_PERM))
KERNEL(k_ffbh, X8(F_FFBH))
KERNEL(k_pkmax, X8(F_PKMAX))
KERNEL(k_pksubc, X8(F_PKSUBC))
KERNEL(k_cndvcc, "v_cmp_lt_i32 vcc, %0, %8\n" X8(F_CNDVCC))
KERNEL(k_cndsgpr, "v_cmp_lt_i32_e64 s[20:21], %0, %8\n" X8(F_CNDSGPR))
KERNEL(k_cndvcc_nocmp, X8(F_CNDVCC))
#define F_CNDVCC64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define F_CNDVCC_OTHER(i) "v_cndmask_b32 %" #i ", %9, %8, vcc\n"
KERNEL(k_cndvcc64, "v_cmp_lt_i32 vcc, %0, %8\n" X8(F_CNDVCC64))
KERNEL(k_cndvcc_salu, "s_mov_b64 vcc, 0x5555\n" X8(F_CNDVCC))
KERNEL(k_cndvcc_nodep, "v_cmp_lt_i32 vcc, %0, %8\n" X8(F_CNDVCC_OTHER))
KERNEL(k_cmp_then_adds, "v_cmp_lt_i32 vcc, %0, %8\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %2, %2, %8\n v_cmp_lt_i32 vcc, %3, %8\n v_add_u32 %4, %4, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_add_u32 %5, %5, %8\n")
KERNEL(k_cndvcc_pair, "v_cmp_lt_i32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_i32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                      "v_cmp_lt_i32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_i32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n")
KERNEL(k_cndsgpr_pair, "v_cmp_lt_i32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_lt_i32_e64 s[22:23], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
                       "v_cmp_lt_i32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_lt_i32_e64 s[22:23], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, %9, %0\n v_lshrrev_b32 %1, %9, %1\n v_lshrrev_b32 %2, %9, %2\n v_lshrrev_b32 %3, %9, %3\n v_lshrrev_b32 %4, %9, %4\n v_lshrrev_b32 %5, %9, %5\n v_lshrrev_b32 %6, %9, %6\n v_lshrrev_b32 %7, %9, %7\n")
KERNEL(k_or, "v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n")
KERNEL(k_shl_const, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
KERNEL(k_cmp, X8(F_CMP))
KERNEL(k_cmps, X8(F_CMPS))
KERNEL(k_readlane, X8(F_READLANE))
KERNEL(k_movdpp, X8(F_MOVDPP))
KERNEL(k_maxdpp, DPP_ALT)
KERNEL(k_sqrt, X8(F_SQRT))
KERNEL(k_cvt, X8(F_CVT))
KERNEL(k_bcnt, X8(F_BCNT))
KERNEL(k_addclamp, X8(F_ADDCLAMP))
KERNEL(k_sdwa, X8(F_SDWA))

// ---- second batch: candidates for substitutions, literal operands, mixes -------------------------
#define F_MAD16(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %9\n"
#define F_MAD16HI(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %9 op_sel:[1,0,0,0]\n"
#define F_DOT2(i) "v_dot2_i32_i16 %" #i ", %8, %9, %" #i "\n"
#define F_ADDLIT(i) "v_add_u32 %" #i ", 0x8001, %" #i "\n"
#define F_ANDLIT(i) "v_and_b32 %" #i ", 0xffff8000, %" #i "\n"
#define F_MOVSDWA(i) "v_mov_b32_sdwa %" #i ", %" #i " dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
#define F_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define F_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 15\n"
#define F_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 16, %8\n"
#define F_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define F_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define F_PKADD(i) "v_pk_add_i16 %" #i ", %" #i ", %8\n"
#define F_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define F_SUBREV(i) "v_subrev_u32 %" #i ", %8, %" #i "\n"
#define F_ASHR16(i) "v_ashrrev_i32 %" #i ", 16, %" #i "\n"
#define F_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define F_MAX3(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define F_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define F_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define F_CNDE64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
KERNEL(k_mad16, X8(F_MAD16))
KERNEL(k_mad16hi, X8(F_MAD16HI))
KERNEL(k_dot2, X8(F_DOT2))
KERNEL(k_addlit, X8(F_ADDLIT))
KERNEL(k_andlit, X8(F_ANDLIT))
KERNEL(k_movsdwa, X8(F_MOVSDWA))
KERNEL(k_addsdwa, X8(F_ADDSDWA))
KERNEL(k_alignbit, X8(F_ALIGNBIT))
KERNEL(k_lshlor, X8(F_LSHLOR))
KERNEL(k_bfi, X8(F_BFI))
KERNEL(k_not, X8(F_NOT))
KERNEL(k_pkadd, X8(F_PKADD))
KERNEL(k_madu24, X8(F_MADU24))
KERNEL(k_subrev, X8(F_SUBREV))
KERNEL(k_ashr16, X8(F_ASHR16))
KERNEL(k_med3, X8(F_MED3))
KERNEL(k_max3, X8(F_MAX3))
KERNEL(k_xad, X8(F_XAD))
KERNEL(k_addco, X8(F_ADDCO))
KERNEL(k_cnde64, X8(F_CNDE64))
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                 "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                 "v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %4\n v_permlane16_swap_b32 %5, %6\n v_permlane16_swap_b32 %7, %0\n")
// mixes: do the two classes simply add up?
KERNEL(k_mix_add_shl, "v_add_u32 %0, %0, %8\n v_lshlrev_b32 %1, 3, %1\n v_add_u32 %2, %2, %8\n v_lshlrev_b32 %3, 3, %3\n"
                      "v_add_u32 %4, %4, %8\n v_lshlrev_b32 %5, 3, %5\n v_add_u32 %6, %6, %8\n v_lshlrev_b32 %7, 3, %7\n")
KERNEL(k_mix_add_dot, "v_add_u32 %0, %0, %8\n v_dot2c_i32_i16 %1, %8, %9\n v_add_u32 %2, %2, %8\n v_dot2c_i32_i16 %3, %8, %9\n"
                      "v_add_u32 %4, %4, %8\n v_dot2c_i32_i16 %5, %8, %9\n v_add_u32 %6, %6, %8\n v_dot2c_i32_i16 %7, %8, %9\n")
// dependent chain on ONE register (latency of back-to-back dependent ops from one wave; other waves fill in)
KERNEL(k_dep_add, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n"
                  "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n")
KERNEL(k_dep_shl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n"
                  "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %0, 1, %0\n")
// scalar ALU and a VALU/SALU interleave
KERNEL_SCLOB(k_salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
               "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")
KERNEL_SCLOB(k_valu_salu, "v_lshlrev_b32 %0, 3, %0\n s_add_u32 s20, s20, 1\n v_lshlrev_b32 %1, 3, %1\n s_add_u32 s21, s21, 1\n"
                    "v_lshlrev_b32 %2, 3, %2\n s_add_u32 s22, s22, 1\n v_lshlrev_b32 %3, 3, %3\n s_add_u32 s23, s23, 1\n")

// ---- third batch (round 2): how long must a run of fast-class ops be to keep its ~2.3-cycle rate? -------------
// A = v_add_u32 (fast class), X = v_perm_b32 (4-cycle class), all on independent registers.
#define A_(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define X_(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define M_(i) "v_mov_b32 %" #i ", %8\n"
#define S_(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define D_(i) "v_dot2c_i32_i16 %" #i ", %8, %9\n"
KERNEL(k_run1, A_(0) X_(1) A_(2) X_(3) A_(4) X_(5) A_(6) X_(7))                 // A X A X ...
KERNEL(k_run2, A_(0) A_(1) X_(2) X_(3) A_(4) A_(5) X_(6) X_(7))                 // AA XX AA XX
KERNEL(k_run4, A_(0) A_(1) A_(2) A_(3) X_(4) X_(5) X_(6) X_(7))                 // AAAA XXXX
KERNEL(k_run6, A_(0) A_(1) A_(2) A_(3) A_(4) A_(5) X_(6) X_(7))                 // AAAAAA XX
KERNEL(k_run7, A_(0) A_(1) A_(2) A_(3) A_(4) A_(5) A_(6) X_(7))                 // AAAAAAA X
KERNEL(k_aax, A_(0) A_(1) X_(2) A_(3) A_(4) X_(5) A_(6) A_(7))                  // AAX AAX AA (wraps into AAAAX...)
KERNEL(k_mixfast, A_(0) M_(1) S_(2) A_(3) M_(4) S_(5) A_(6) S_(7))              // different fast-class opcodes back to back
KERNEL(k_dep2, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n" X_(1) "v_add_u32 %2, %2, %8\n v_add_u32 %2, %2, %8\n" X_(3)
               "v_add_u32 %4, %4, %8\n v_add_u32 %4, %4, %8\n" X_(5))           // dependent fast pairs between 4-cycle ops (9 instructions)
KERNEL(k_run2d, A_(0) A_(1) D_(2) D_(3) A_(4) A_(5) D_(6) D_(7))                // AA DD with dot2c
KERNEL(k_e64, "v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n"
              "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")   // fast op, VOP3 encoding
KERNEL_SCLOB(k_addsgpr, "v_add_u32 %0, s20, %0\n v_add_u32 %1, s20, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s20, %3\n"
                        "v_add_u32 %4, s20, %4\n v_add_u32 %5, s20, %5\n v_add_u32 %6, s20, %6\n v_add_u32 %7, s20, %7\n")        // SGPR operand
KERNEL_SCLOB(k_nop, "s_nop 0\n v_perm_b32 %0, %0, %8, %9\n s_nop 0\n v_perm_b32 %1, %1, %8, %9\n s_nop 0\n v_perm_b32 %2, %2, %8, %9\n s_nop 0\n v_perm_b32 %3, %3, %8, %9\n")

// ---- fourth batch: 64-bit shifts (candidates for "shift by a signed count" in one instruction) ------------------
#define KERNEL64(NAME, ASM4)                                                                   \
    __global__ __launch_bounds__(256) void NAME(int *out, int iters) {                         \
        long long a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 5, a3 = a0 + 7;                 \
        int c = (threadIdx.x & 7) + 1;                                                         \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 64; ++r) { asm volatile(ASM4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)); } \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a0 + a1 + a2 + a3);                 \
    }
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n"
                   "v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n")
KERNEL64(k_ashr64, "v_ashrrev_i64 %0, %4, %0\n v_ashrrev_i64 %1, %4, %1\n v_ashrrev_i64 %2, %4, %2\n v_ashrrev_i64 %3, %4, %3\n"
                   "v_ashrrev_i64 %0, %4, %0\n v_ashrrev_i64 %1, %4, %1\n v_ashrrev_i64 %2, %4, %2\n v_ashrrev_i64 %3, %4, %3\n")
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n"
                   "v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n")
KERNEL(k_alignbit_v, "v_alignbit_b32 %0, %0, %8, %9\n v_alignbit_b32 %1, %1, %8, %9\n v_alignbit_b32 %2, %2, %8, %9\n v_alignbit_b32 %3, %3, %8, %9\n"
                     "v_alignbit_b32 %4, %4, %8, %9\n v_alignbit_b32 %5, %5, %8, %9\n v_alignbit_b32 %6, %6, %8, %9\n v_alignbit_b32 %7, %7, %8, %9\n")
KERNEL(k_mulhi_i, "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n")
KERNEL(k_ffbh_i, "v_ffbh_i32 %0, %0\n v_ffbh_i32 %1, %1\n v_ffbh_i32 %2, %2\n v_ffbh_i32 %3, %3\n v_ffbh_i32 %4, %4\n v_ffbh_i32 %5, %5\n v_ffbh_i32 %6, %6\n v_ffbh_i32 %7, %7\n")
KERNEL(k_bfm, "v_bfm_b32 %0, %0, %8\n v_bfm_b32 %1, %1, %8\n v_bfm_b32 %2, %2, %8\n v_bfm_b32 %3, %3, %8\n v_bfm_b32 %4, %4, %8\n v_bfm_b32 %5, %5, %8\n v_bfm_b32 %6, %6, %8\n v_bfm_b32 %7, %7, %8\n")
KERNEL(k_cvtrcp, "v_cvt_f32_u32 %0, %0\n v_rcp_iflag_f32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_rcp_iflag_f32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_rcp_iflag_f32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_rcp_iflag_f32 %3, %3\n")

typedef void (*kern_t)(int *, int);
static void run(const char *name, kern_t fn, int waves_per_simd, double extra_per_8 = 0) {
    int *out;
    const int blocks = 256 * waves_per_simd;   // 256 CUs: waves_per_simd blocks of 4 waves each -> that many waves per SIMD
    hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 100;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 64 * (8 + extra_per_8) * waves_per_simd;
    printf("%-22s w/SIMD %d : %7.3f ms  %.2f ns/inst/SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
    hipFree(out);
}

int main(int argc, char **argv) {
    struct { const char *n; kern_t f; double extra; } t[] = {
        {"v_add_u32", k_add, 0}, {"v_sub_u32", k_sub, 0}, {"v_and_b32", k_and, 0}, {"v_xor_b32", k_xor, 0},
        {"v_lshlrev_b32", k_shl, 0}, {"v_ashrrev_i32", k_ashr, 0}, {"v_max_i32", k_max, 0}, {"v_min_u32", k_minu, 0},
        {"v_bfe_i32", k_bfe, 0}, {"v_lshl_add_u32", k_lshladd, 0}, {"v_add3_u32", k_add3, 0}, {"v_and_or_b32", k_andor, 0},
        {"v_mul_i32_i24", k_mul24, 0}, {"v_mad_i32_i24", k_mad24, 0}, {"v_mul_lo_u32", k_mullo, 0}, {"v_mul_hi_u32", k_mulhi, 0},
        {"v_dot2c_i32_i16", k_dot2c, 0}, {"v_perm_b32", k_perm, 0}, {"v_ffbh_u32", k_ffbh, 0}, {"v_pk_max_i16", k_pkmax, 0},
        {"v_pk_sub_i16 clamp", k_pksubc, 0}, {"v_cndmask_b32 vcc", k_cndvcc, 1}, {"v_cndmask_b32 sgpr", k_cndsgpr, 1},
        {"cndmask vcc (no cmp)", k_cndvcc_nocmp, 0}, {"cmp + 8 cndmask_e64 vcc", k_cndvcc64, 1},
        {"s_mov vcc + 8 cndmask", k_cndvcc_salu, 0}, {"cmp + 8 cndmask (dst!=src)", k_cndvcc_nodep, 1}, {"cmp,add,cnd,add mix", k_cmp_then_adds, 0}, {"cmp+cndmask vcc pairs", k_cndvcc_pair, 0},
        {"cmp+cndmask sgpr pairs", k_cndsgpr_pair, 0}, {"v_lshrrev_b32", k_lshr, 0}, {"v_or_b32", k_or, 0},
        {"v_lshlrev_b32 const", k_shl_const, 0}, {"v_mov_b32", k_mov, 0},
        {"v_cmp_lt_i32 vcc", k_cmp, 0}, {"v_cmp_lt_i32 sgpr", k_cmps, 0}, {"v_readlane_b32", k_readlane, 0},
        {"v_mov_b32_dpp", k_movdpp, 0}, {"v_max_i32_dpp", k_maxdpp, 0}, {"v_sqrt_f32", k_sqrt, 0}, {"v_cvt_f32_u32", k_cvt, 0},
        {"v_bcnt_u32_b32", k_bcnt, 0}, {"v_add_i32 clamp", k_addclamp, 0}, {"v_mul_i32_i24_sdwa", k_sdwa, 0},
    };
    struct { const char *n; kern_t f; double extra; } t2[] = {
        {"v_mad_i32_i16", k_mad16, 0}, {"v_mad_i32_i16 op_sel hi", k_mad16hi, 0}, {"v_dot2_i32_i16 (VOP3P)", k_dot2, 0},
        {"v_add_u32 literal", k_addlit, 0}, {"v_and_b32 literal", k_andlit, 0}, {"v_mov_b32_sdwa", k_movsdwa, 0},
        {"v_add_u32_sdwa", k_addsdwa, 0}, {"v_alignbit_b32", k_alignbit, 0}, {"v_lshl_or_b32", k_lshlor, 0}, {"v_bfi_b32", k_bfi, 0},
        {"v_not_b32", k_not, 0}, {"v_pk_add_i16", k_pkadd, 0}, {"v_mad_u32_u24", k_madu24, 0}, {"v_subrev_u32", k_subrev, 0},
        {"v_ashrrev_i32 16", k_ashr16, 0}, {"v_med3_i32", k_med3, 0}, {"v_max3_i32", k_max3, 0}, {"v_xad_u32", k_xad, 0},
        {"v_add_co_u32", k_addco, 0}, {"v_cndmask_b32_e64 sgpr", k_cnde64, 0}, {"v_permlane32_swap", k_swap32, 0},
        {"v_permlane16_swap", k_swap16, 0}, {"mix add/lshl", k_mix_add_shl, 0}, {"mix add/dot2c", k_mix_add_dot, 0},
        {"dependent v_add chain", k_dep_add, 0}, {"dependent v_lshl chain", k_dep_shl, 0}, {"s_add_u32", k_salu, 0},
        {"v_lshl + s_add interleaved (per pair)", k_valu_salu, -4},
    };
    if (argc > 1 && !strcmp(argv[1], "--batch4")) {
        struct { const char *n; kern_t f; double extra; } t4[] = {
            {"v_lshrrev_b64", k_lshr64, 0}, {"v_ashrrev_i64", k_ashr64, 0}, {"v_lshlrev_b64", k_lshl64, 0},
            {"v_alignbit_b32 (vgpr shift)", k_alignbit_v, 0}, {"v_mul_hi_i32", k_mulhi_i, 0}, {"v_ffbh_i32", k_ffbh_i, 0},
            {"v_bfm_b32", k_bfm, 0}, {"v_cvt_f32_u32 + v_rcp_iflag_f32 (per pair)", k_cvtrcp, -4},
        };
        for (int w : {7})
            for (auto &e : t4) run(e.n, e.f, w, e.extra);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--batch3")) {
        // extra = instructions per unrolled body - 8
        struct { const char *n; kern_t f; double extra; } t3[] = {
            {"A X A X A X A X (run 1)", k_run1, 0}, {"AA XX AA XX (run 2)", k_run2, 0}, {"AAAA XXXX (run 4)", k_run4, 0},
            {"AAAAAA XX (run 6)", k_run6, 0}, {"AAAAAAA X (run 7)", k_run7, 0}, {"AAX AAX AA", k_aax, 0},
            {"add mov ashr mixed fast ops", k_mixfast, 0}, {"dependent AA X x3 (9 instr)", k_dep2, 1},
            {"AA DD (dot2c) run 2", k_run2d, 0}, {"v_add_u32_e64 x8", k_e64, 0}, {"v_add_u32 sgpr operand x8", k_addsgpr, 0},
            {"s_nop 0 + v_perm x4 (per pair)", k_nop, -4},
        };
        for (int w : {2, 7})
            for (auto &e : t3) run(e.n, e.f, w, e.extra);
        return 0;
    }
    const bool second = argc > 1 && !strcmp(argv[1], "--batch2");
    if (!second)
        for (auto &e : t) run(e.n, e.f, 5, e.extra);
    else
        for (int w : {1, 2, 6})
            for (auto &e : t2) run(e.n, e.f, w, e.extra);
    return 0;
}
