// Issue cost of the gfx950 vector instructions the block kernel is made of, relative to v_add_u32.
//
// The block kernel is bound by the vector issue port, so what an instruction costs is the time the port
// is held, not its latency.  Each test fills every SIMD with kWaves waves that run the same straight-line body of 64
// copies of one instruction (8 independent destination registers, so that with several waves per SIMD no dependency is
// ever waited for); the ratio of its time to the v_add_u32 body's is the cost in "plain VALU" units (4 cycles each).
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o webrtc_aecm_amd/_lib/valu_rates && gpurun -- webrtc_aecm_amd/_lib/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define R64(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T)

// one kernel per instruction: BODY(k) is the asm text with destination/source registers picked by k
#define DEFINE_TEST(name, BODY)                                                                          \
    __global__ void __launch_bounds__(64) test_##name(int *out, int iters, int a_in, int b_in) {                       \
        int d[8], a = a_in + (int)threadIdx.x, b = b_in | 1;                                                            \
        unsigned long long w[8];                                                                                       \
        int s0 = 0;                                                                                                    \
        unsigned long long mask = __ballot(a & 1);                                                                     \
        for (int k = 0; k < 8; ++k) { d[k] = a + k; w[k] = (unsigned long long)(a + k) * 0x100000001ull; }              \
        for (int i = 0; i < iters; ++i) { R64(BODY) }                                                                  \
        int acc = s0;                                                                                                  \
        for (int k = 0; k < 8; ++k) acc += d[k] + (int)w[k] + (int)(w[k] >> 32);                                        \
        if (acc == 0x7fffffff) out[threadIdx.x] = acc;                                                                 \
    }

#define V3(op) [](int) {}
#define T_ADD(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(d[k]) : "v"(a));
#define T_MUL_LO(k) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MUL_HI_U(k) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MUL_HI_I(k) asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MUL_U24(k) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MUL_I24(k) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MAD_U24(k) asm volatile("v_mad_u32_u24 %0, %1, %0, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_MAD_I24(k) asm volatile("v_mad_i32_i24 %0, %1, %0, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_MAD_U64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[k]) : "v"(b), "v"(a) : "vcc");
#define T_MUL_HI_U24(k) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(d[k]));
#define T_RCP_IFLAG(k) asm volatile("v_rcp_iflag_f32 %0, %0" : "+v"(d[k]));
#define T_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(d[k]));
#define T_CVT_F32_U32(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(d[k]));
#define T_CVT_U32_F32(k) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(d[k]));
#define T_CVT_FLR(k) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(d[k]));
#define T_MUL_F32(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(d[k]) : "v"(a));
#define T_FMA_F32(k) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_LSHR_B64(k) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w[k]) : "v"(b));
#define T_ASHR_I64(k) asm volatile("v_ashrrev_i64 %0, %1, %0" : "+v"(w[k]) : "v"(b));
#define T_LSHL_B64(k) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w[k]) : "v"(b));
#define T_DOT2(k) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_DOT2C(k) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_PK_ADD(k) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_PK_MUL(k) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_PK_MAD(k) asm volatile("v_pk_mad_u16 %0, %1, %0, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_PK_ASHR(k) asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_PK_MIN(k) asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_MOV_DPP_SHR(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d[k]) : "v"(a));
#define T_ADD_DPP_SHR(k) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d[k]) : "v"(a));
#define T_MOV_DPP_BCAST(k) asm volatile("v_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(d[k]) : "v"(a));
#define T_MOV_DPP_QUAD(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d[k]) : "v"(a));
#define T_MAX_DPP(k) asm volatile("v_max_i32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(d[k]) : "v"(a));
#define T_ADD_SDWA(k) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(d[k]) : "v"(a));
#define T_READLANE(k) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(d[k]));
#define T_READFIRST(k) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(d[k]));
#define T_WRITELANE(k) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(d[k]) : "s"(b_in));
#define T_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "s"(mask));
#define T_PK_SUB_CLAMP(k) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(d[k]) : "v"(b));
#define T_MAD_I32_I16(k) asm volatile("v_mad_i32_i16 %0, %1, %0, %2" : "+v"(d[k]) : "v"(b), "v"(a));
#define T_MIN_I32(k) asm volatile("v_min_i32 %0, %1, %0" : "+v"(d[k]) : "v"(a));
#define T_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(d[k]) : "v"(a));
#define T_LSHRREV(k) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_ADD_SGPR(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(d[k]) : "s"(b_in));
#define T_ADD_LIT(k) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(d[k]));
#define T_CMP_VCC(k) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(d[k]), "v"(a) : "vcc");
#define T_CMP_SGPR(k) asm volatile("v_cmp_lt_i32_e64 s[20:21], %0, %1" : : "v"(d[k]), "v"(a) : "s20", "s21");
#define T_PERM(k) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_BFE_I32(k) asm volatile("v_bfe_i32 %0, %0, 3, 16" : "+v"(d[k]));
#define T_ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(d[k]) : "v"(a));
#define T_MED3(k) asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_MAX3(k) asm volatile("v_max3_i32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_ADD3(k) asm volatile("v_add3_u32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_LSHL_ADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(d[k]) : "v"(a));
#define T_AND_OR(k) asm volatile("v_and_or_b32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_FFBH(k) asm volatile("v_ffbh_u32 %0, %0" : "+v"(d[k]));
#define T_FFBH_I(k) asm volatile("v_ffbh_i32 %0, %0" : "+v"(d[k]));
#define T_BCNT(k) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[k]) : "v"(a));
#define T_ASHR(k) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_LSHLREV(k) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(d[k]) : "v"(b));
#define T_SUB_CO(k) asm volatile("v_sub_co_u32 %0, vcc, %1, %0" : "+v"(d[k]) : "v"(a) : "vcc");
#define T_ADD_I32_CLAMP(k) asm volatile("v_add_i32 %0, %1, %0 clamp" : "+v"(d[k]) : "v"(a));
#define T_ADD_I16_CLAMP(k) asm volatile("v_add_i16 %0, %1, %0 clamp" : "+v"(d[k]) : "v"(a));
#define T_SAD(k) asm volatile("v_sad_u32 %0, %1, %0, %2" : "+v"(d[k]) : "v"(a), "v"(b));
#define T_PERMLANE32(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d[k]), "+v"(d[(k + 1) & 7]));
#define T_PERMLANE16(k) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(d[k]), "+v"(d[(k + 1) & 7]));
#define T_SWIZZLE(k) asm volatile("ds_swizzle_b32 %0, %0 offset:0x8041\n s_waitcnt lgkmcnt(0)" : "+v"(d[k]));
#define T_BPERMUTE(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(d[k]) : "v"(a));
#define T_SWIZZLE_NOWAIT(k) asm volatile("ds_swizzle_b32 %0, %0 offset:0x8041" : "+v"(d[k]));
#define T_BPERMUTE_NOWAIT(k) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(d[k]) : "v"(a));
#define T_SALU(k) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(b_in) : "scc");
#define T_SMUL(k) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(b_in));
#define T_SALU_VALU(k) asm volatile("s_add_u32 %1, %1, %2\n v_add_u32 %0, %3, %0" : "+v"(d[k]), "+s"(s0) : "s"(b_in), "v"(a) : "scc");
#define T_NOP(k) asm volatile("s_nop 0");

#define ALL_TESTS(X) \
    X(ADD) X(MUL_LO) X(MUL_HI_U) X(MUL_HI_I) X(MUL_U24) X(MUL_I24) X(MAD_U24) X(MAD_I24) X(MAD_U64) X(MUL_HI_U24) \
    X(RCP) X(RCP_IFLAG) X(SQRT) X(CVT_F32_U32) X(CVT_U32_F32) X(CVT_FLR) X(MUL_F32) X(FMA_F32) \
    X(LSHR_B64) X(ASHR_I64) X(LSHL_B64) X(DOT2) X(DOT2C) X(PK_ADD) X(PK_MUL) X(PK_MAD) X(PK_ASHR) X(PK_MIN) \
    X(MOV_DPP_SHR) X(ADD_DPP_SHR) X(MOV_DPP_BCAST) X(MOV_DPP_QUAD) X(MAX_DPP) X(ADD_SDWA) \
    X(READLANE) X(READFIRST) X(WRITELANE) X(CNDMASK) X(PK_SUB_CLAMP) X(MAD_I32_I16) X(MIN_I32) X(MOV) X(LSHRREV) X(ADD_SGPR) X(ADD_LIT) X(CMP_VCC) X(CMP_SGPR) X(PERM) X(BFE_I32) X(ALIGNBIT) X(MED3) X(MAX3) X(ADD3) \
    X(LSHL_ADD) X(AND_OR) X(FFBH) X(FFBH_I) X(BCNT) X(ASHR) X(LSHLREV) X(SUB_CO) X(ADD_I32_CLAMP) X(ADD_I16_CLAMP) X(SAD) \
    X(PERMLANE32) X(PERMLANE16) X(SWIZZLE) X(BPERMUTE) X(SWIZZLE_NOWAIT) X(BPERMUTE_NOWAIT) X(SALU) X(SMUL) X(SALU_VALU) X(NOP)

#define X(n) DEFINE_TEST(n, T_##n)
ALL_TESTS(X)
#undef X

typedef void (*Kernel)(int *, int, int, int);
struct Test { const char *name; Kernel k; };

int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 7;
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int waves = cus * 4 * waves_per_simd;
    int *out;
    CHECK(hipMalloc(&out, 64 * sizeof(int)));
    std::vector<Test> tests = {
#define X(n) {#n, test_##n},
        ALL_TESTS(X)
#undef X
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double base = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"clock_mhz\": %d, \"tests\": {\n", prop.name, cus, waves_per_simd, iters,
           prop.clockRate / 1000);
    for (size_t t = 0; t < tests.size(); ++t) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(tests[t].k, dim3(waves), dim3(64), 0, 0, out, iters, 3, 5);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        // per SIMD: waves_per_simd waves x iters x 64 instructions issued one after the other
        const double ns_per_instr = best * 1e6 / ((double)waves_per_simd * iters * 64);
        if (t == 0) base = ns_per_instr;
        printf("  \"%s\": {\"ns\": %.4f, \"vs_add\": %.3f}%s\n", tests[t].name, ns_per_instr, ns_per_instr / base, t + 1 < tests.size() ? "," : "");
    }
    printf("}}\n");
    return 0;
}
