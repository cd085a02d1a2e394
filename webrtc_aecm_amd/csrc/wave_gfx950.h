// gfx950 (MI355X / CDNA4) wave policy for aecm_wave.h: a lane vector is an int in a VGPR, a
// wave-uniform value is an int the compiler keeps in an SGPR, tables live in LDS.
//
// Two variants of the cross-lane primitives:
//   kFast = false : only ds_bpermute-based HIP shuffles (__shfl_xor/__shfl_up) -- semantics are the
//                   documented ones, used as the in-kernel reference for the self test;
//   kFast = true  : DPP row operations (quad_perm / row_ror / row_shl+row_shr with bank masks /
//                   row_half_mirror / row_mirror / wave_shr), v_permlane16_swap / v_permlane32_swap
//                   for the two widest FFT exchanges, v_readlane for the final cross-row combine.
// aecm_selftest_kernel (aecm_kernels.hip) checks every kFast primitive against kFast = false and
// isqrt31 against its definition on the device before any parity claim is made.
#ifndef AECM_AMD_WAVE_GFX950_H_
#define AECM_AMD_WAVE_GFX950_H_

#include <hip/hip_runtime.h>

#include "aecm_ops.h"
#include "aecm_state.h"

namespace aecm {

// LDS tables, filled by the kernel prologue (aecm_kernels.hip).
struct LdsTables {
    int lane_rows[kLaneConstRows][kLanes];   // LaneConstRow: per-lane constants (see lane_const() in aecm_wave.h)
    // Packed twiddles of the inverse transform and their per-half negations, (w_re, w_im, -w_re, -w_im) per
    // [stage][lane]: a stage is one ds_read_b128 at lane*16 + constant offset, no VALU address or packing work.
    int4 twiddle_inv[7][64];
    // Forward stages 1..6 in the multiply-add form of fft128: (w_re, w_im, -w_re, -w_im), one ds_read_b128;
    // stages 2, 4, 6 additionally (s_re, 1 - s_re, s_im, 1 - s_im).
    int4 fwd_twiddle[6][64];
    int4 fwd_offset[3][64];
    int cossin[360];   // lo16: cos Q13, hi16: sin Q13     comfort-noise phase table
    int hann[kLdsHannWords];   // sqrt-Hanning Q14 (65 entries + pad)
};
static_assert(sizeof(LdsTables) == kConstBlobWords * 4, "LDS image layout (aecm_state.h) out of sync");
extern __shared__ LdsTables g_lds[];   // one instance (dynamic LDS)

#ifndef AECM_PRIORITY_ROTATION_MIN_BLOCKS
#define AECM_PRIORITY_ROTATION_MIN_BLOCKS 8
#endif
#ifndef AECM_PRIORITY_ROTATION_PERIOD_LOG2
#define AECM_PRIORITY_ROTATION_PERIOD_LOG2 0   // measured (r3): every block 839, every 4th 834, every 16th 835, never 829 M frames/s
#endif
// Issue priority by phase of the block: entry k = phase k of tools/isa_phase_breakdown.py (the code after marker k - 1 of
// aecm_wave.h; entry 0 is unused): 1-2 forward transforms + magnitudes, 3-5 delay estimator, 6 energies / VAD, 7 NLMS,
// 8 Wiener gain, 9-10 NLP + comfort noise, 11 inverse transform, 12 synthesis, 13 output store.
// Rule: the priority never falls as the block advances -- a wave closer to finishing its block is served first, and the
// forward transforms (dense, independent vector work at the start of a block) of the waves that yield soak up whatever
// issue slots remain.  A wave in the scalar-heavy middle has few vector instructions, each on the critical path between
// scalar ones: serving those first gets it through quickly.  Measured (r3, 65 536 streams, M frames/s, one box per group):
//   per-block rotation 841 | phases 3..10 at 3, transforms at 0: 851 | + inverse transform / synthesis at 1: 875, at 3: 884,
//   + phase 13 at 3: 893 | forward transforms HIGH, middle low: 831 | comfort noise one level below its neighbours: 788 (!) |
//   falling after the middle (3,3,3,3,3,2,2,2,1,1,1): 775 (!)
//   after the round's instruction-count work (second box): all 3 but the transforms 892 | 1,1,1,2,2,2,2,2,3,...: 901 |
//   2,2,2,2,2,3,...: 908 | 1,1,1,2,2,3,... (shipped): 916 | 2,2,2,1,1,2,3,...: 916 | 1,1,1,1,1,3,...: 912
// profiles/r03_experiments.md has the full tables.
#ifndef AECM_PHASE_PRIOS
#define AECM_PHASE_PRIOS 0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 3, 3, 3, 3
#endif
#ifndef AECM_QUAD_EXCHANGE_SWIZZLE
#define AECM_QUAD_EXCHANGE_SWIZZLE 1
#endif
#ifndef AECM_LANE_CONSTS_IN_LDS
#define AECM_LANE_CONSTS_IN_LDS 1
#endif

#define AECM_DPP(old, src, ctrl, row_mask, bank_mask, bound) \
    __builtin_amdgcn_update_dpp((old), (src), (ctrl), (row_mask), (bank_mask), (bound))

enum : int {
    kDppQuadXor1 = 0xB1,       // quad_perm:[1,0,3,2]
    kDppQuadXor2 = 0x4E,       // quad_perm:[2,3,0,1]
    kDppRowShl4 = 0x104,       // dst[i] = src[i+4] within a row of 16
    kDppRowShr4 = 0x114,       // dst[i] = src[i-4]
    kDppRowRor8 = 0x128,       // dst[i] = src[(i+8) mod 16]  == xor 8
    kDppWaveShr1 = 0x138,      // dst[i] = src[i-1] across the whole wave
    kDppRowMirror = 0x140,     // dst[i] = src[15-i]
    kDppRowHalfMirror = 0x141, // dst[i] = src[7-i] within 8
    kDppRowBcast15 = 0x142,    // lane 15 of each row -> every lane of the next row
    kDppRowBcast31 = 0x143     // lane 31 -> every lane of rows 2 and 3
};

// v_writelane_b32 has no clang builtin in this toolchain: the LLVM intrinsic through its assembler name (like v_ffbh_i32, aecm_ops.h)
extern "C" __device__ int aecm_llvm_amdgcn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// kPhasePrio: issue priority by phase of the block (AECM_PHASE_PRIOS) instead of the per-block rotation.  The launcher
// picks it for launches of more waves than the chip holds at once (measured, M frames/s, phase priority vs rotation:
// 65 536 streams 893 vs 841, 16 384 streams 827 vs 799, 8 kHz 878 vs 841, with a clean input 779 vs 746; but 4 096
// streams -- fewer waves than the chip has slots, all running in lock step -- 673 vs 697) and for session ticks (0.267 vs 0.270 ms).
// kTightRegisters: the kernel instantiating this policy has little register headroom (the tick kernel, whose I/O plan lives in
// scalar registers next to the engine's): the code-size-for-registers trades of aecm_wave.h (joint scaling tests of the inverse
// transform, the data-dependent short paths) are switched by their own AECM_*_TICK macros there.  At present all of them
// are on in every kernel family.
// kCoherentState: the stream's state (lane vectors, scalars, far-spectrum history) is handed from wave to wave INSIDE one
// launch (the chunk-queue kernel, aecm_block_kernels.hip): every access to it is a relaxed agent-scope atomic -- on gfx950
// a plain global_load / global_store with the sc1 bit, which is served at the level all eight XCDs share instead of the
// CU's L1 or the XCD's L2 -- and the scalars travel as one lane vector instead of through the (non-coherent) scalar cache.
// kDynamicPrio: the phase priorities can be lowered at run time (phase_priority's `drop`): the pipelined kernel demotes the
// waves of a workgroup that has run ahead of the launch's slowest one (aecm_block_kernels.hip), so that the SIMD's arbiter --
// highest priority first, then the OLDEST wave -- stops compounding the head start of the workgroups dispatched first.
#ifndef AECM_PRIO_DROP_LEVELS
#define AECM_PRIO_DROP_LEVELS 3        // how far a demoted wave's phase priorities fall (3: all the way to 0)
#endif
template <bool kFast, bool kPhasePrio = true, bool kTightRegisters = false, bool kCoherentState = false, bool kDynamicPrio = false>
struct Gfx950Wave {
    static constexpr bool kTight = kTightRegisters;
    static constexpr bool kCoherent = kCoherentState;
    using vi = int;
    using vb = bool;
    static constexpr bool kPrecomputedConstants = true;    // lane constants and LDS tables come from the host-built blob
    // Per-lane constants are read from the LDS copy of the blob where they are used (one ds_read_b32
    // each, off the VALU) instead of occupying 8 VGPRs for the whole launch.
    static constexpr bool kLaneConstsInTable = AECM_LANE_CONSTS_IN_LDS;
    // The index is lane_id() plus a zero the compiler cannot see through, re-made every block, so the
    // reads stay inside the block loop next to their uses instead of being hoisted into registers.
    static __device__ __forceinline__ int table_index_for_this_block() {
        int zero = 0;
        asm volatile("" : "+s"(zero));
        return lane_id() + zero;
    }
    template <int ROW>
    static __device__ __forceinline__ int table_lane_const(int index) { return g_lds[0].lane_rows[ROW][index]; }

    // Called at the top of every block.  The SIMD's arbiter favours its oldest wave, so among waves that
    // started together the oldest runs at solo speed and the youngest gets the leftovers: they finish
    // far apart and the tail of a launch runs at low occupancy (a launch whose waves all fit on the chip
    // at once lost 10-16 % to this).  Rotating the wave's issue priority every block, with a per-workgroup
    // offset, shares the ports evenly over time; neutral for many-round launches.
    // A wave-uniform value the compiler must look at afresh in every block.  Conditions on launch-invariant
    // configuration (mult, nlp, cng, ...) are otherwise hoisted out of the block loop as 64-bit lane masks, and with the
    // scalar registers as full as they are here those masks get spilled to lanes of a VGPR and read back with two
    // v_readlane -- VALU instructions -- per use; re-evaluating the condition is one scalar compare.
    // x, behind an addition of a scalar zero the compiler cannot see through.  For wave-uniform values, where the
    // compiler would otherwise fuse scalar arithmetic into an operation only the vector unit has (a saturating subtract,
    // a boolean turned into an integer through v_cndmask) and pay VALU slots plus a v_readfirstlane for ONE value.  (Not
    // an asm register constraint on x itself: where the compiler happens to hold x in a VGPR that is a hard error.)
    static __device__ __forceinline__ int pin_uniform(int x) {
        int zero = 0;
        asm volatile("" : "+s"(zero));
        return x + zero;
    }
    static __device__ __forceinline__ int per_block(int x) {
        asm volatile("" : "+s"(x));
        return x;
    }
    static constexpr int kPhasePrios[] = {AECM_PHASE_PRIOS, -1};
#if defined(AECM_PRIORITY_ROTATION)
    static constexpr bool kPhasePriority = false;         // A/B: the per-block rotation of rounds 1-2 everywhere
#else
    static constexpr bool kPhasePriority = kPhasePrio;
#endif
    // Called where phase PHASE begins (after marker PHASE - 1 ... the marker ids of aecm_wave.h).
    template <int P>
    static __device__ __forceinline__ void set_prio() {
        if constexpr (P <= 0) __builtin_amdgcn_s_setprio(0);
        else if constexpr (P == 1) __builtin_amdgcn_s_setprio(1);
        else if constexpr (P == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
    // drop (wave-uniform, kDynamicPrio only): non-zero = this wave runs AECM_PRIO_DROP_LEVELS below the table.
    template <int PHASE>
    static __device__ __forceinline__ void phase_priority(int drop = 0) {
        if constexpr (kPhasePriority) {
            constexpr int n = (int)(sizeof(kPhasePrios) / sizeof(int)) - 1;             // entries 0 (unused) .. 13
            static_assert(n == 14 && PHASE >= 1 && PHASE <= 13, "one priority per phase 1..13 (entry 0 is unused)");
            constexpr int now = kPhasePrios[PHASE];
            constexpr int before = PHASE == 1 ? kPhasePrios[n - 1] : kPhasePrios[PHASE - 1];   // phase 13 of the block before; begin_stream() makes that true of the first block too
            if constexpr (now != before) {
                if constexpr (kDynamicPrio) {
                    // set at every change of the table, demoted or not: the flag may have changed since the block before
                    constexpr int low = now > AECM_PRIO_DROP_LEVELS ? now - AECM_PRIO_DROP_LEVELS : 0;
                    if (drop != 0) set_prio<low>();
                    else set_prio<now>();
                } else {
                    set_prio<now>();
                }
            }
        }
    }
    // Once per launch, ahead of the block loop: a wave starts at priority 0, and phase_priority<1> only emits an s_setprio
    // when entry 1 differs from entry 13 (the block before) -- so the first block is put where every later block's phase 1
    // is, whatever table is being tried (in the 2-3 block tick launches the first block is most of the launch).
    static __device__ __forceinline__ void begin_stream() {
        if constexpr (kPhasePriority) {
            constexpr int first = kPhasePrios[1];
            if constexpr (first == 1) __builtin_amdgcn_s_setprio(1);
            else if constexpr (first == 2) __builtin_amdgcn_s_setprio(2);
            else if constexpr (first == 3) __builtin_amdgcn_s_setprio(3);
        }
    }
    static __device__ __forceinline__ void begin_block(int blk, int n_blocks) {
        if (kPhasePriority) return;
        if (per_block(n_blocks) < AECM_PRIORITY_ROTATION_MIN_BLOCKS) return;   // a 2-3 block tick launch is over before shares even out
        // the priority changes every 2^AECM_PRIORITY_ROTATION_PERIOD_LOG2 blocks: the four-way switch below is three
        // branches, paid once per period instead of once per block
        if (AECM_PRIORITY_ROTATION_PERIOD_LOG2 > 0 && (blk & ((1 << AECM_PRIORITY_ROTATION_PERIOD_LOG2) - 1)) != 0) return;
        const unsigned h = (blockIdx.x * 2654435761u) >> 16;
        switch ((((unsigned)blk >> AECM_PRIORITY_ROTATION_PERIOD_LOG2) + h) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    static __device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
    // The lane id the per-stream addresses of run_stream_io are built from.  In the chunk-queue kernel, which runs many
    // streams per wave, a copy the compiler cannot see through: the per-lane halves of those 64-bit addresses are then
    // formed per item instead of being hoisted out of the item loop into registers that do not exist (scratch spills).
    static __device__ __forceinline__ int stream_lane_id() {
        int t = lane_id();
        if constexpr (kCoherent) asm volatile("" : "+v"(t));
        return t;
    }
    static __device__ __forceinline__ bool is_first_lane() { return lane_id() == 0; }
    static __device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
    static __device__ __forceinline__ void div_magic_lanes(int d, int &magic, int &shift) { div_magic(d, &magic, &shift); }

    // ---- tables ----
    static __device__ __forceinline__ int hann(int i) { return g_lds[0].hann[i]; }
    template <int S, bool kInverse>
    static __device__ __forceinline__ void twiddles(int &w_re, int &w_im) {
        if constexpr (kInverse) {
            const int4 w = g_lds[0].twiddle_inv[S][lane_id()];
            w_re = w.x;
            w_im = w.y;
        } else if constexpr (S == 0) {            // W^0 = (32767, 0)
            w_re = 32767;
            w_im = (int)(32767u << 16);
        } else {                                  // forward stages live in the multiply-add table
            const int4 w = g_lds[0].fwd_twiddle[S - 1][lane_id()];
            w_re = w.x;
            w_im = w.y;
        }
    }
    template <int S>
    static __device__ __forceinline__ void inv_twiddles(int &w_re, int &w_im, int &nw_re, int &nw_im) {
        const int4 w = g_lds[0].twiddle_inv[S][lane_id()];
        w_re = w.x; w_im = w.y; nw_re = w.z; nw_im = w.w;
    }
    // A compile-time constant the compiler must keep in a VGPR (three-operand VALU instructions take one scalar
    // or literal operand only; pinning the other constant of an and-or in a register keeps it one instruction).
    static __device__ __forceinline__ int opaque_const(int k) {
        asm("" : "+v"(k));
        return k;
    }
    template <int S>
    static __device__ __forceinline__ void fwd_twiddles(int &w_re, int &w_im, int &nw_re, int &nw_im) {
        const int4 w = g_lds[0].fwd_twiddle[S - 1][lane_id()];
        w_re = w.x; w_im = w.y; nw_re = w.z; nw_im = w.w;
    }
    template <int S>
    static __device__ __forceinline__ void fwd_offsets(int &s_re, int &one_minus_s_re, int &s_im, int &one_minus_s_im) {
        const int4 o = g_lds[0].fwd_offset[S / 2 - 1][lane_id()];
        s_re = o.x; one_minus_s_re = o.y; s_im = o.z; one_minus_s_im = o.w;
    }
#if defined(AECM_PROBE_COSSIN_NO_GATHER)     // counter attribution only (wrong results): the comfort-noise phase gather reads entry = lane instead
    static __device__ __forceinline__ int cos360(int i) { return sext16(g_lds[0].cossin[lane_id() + (i & 0)]); }
    static __device__ __forceinline__ int sin360(int i) { return g_lds[0].cossin[lane_id() + (i & 0)] >> 16; }
#else
    static __device__ __forceinline__ int cos360(int i) { return sext16(g_lds[0].cossin[i]); }
    static __device__ __forceinline__ int sin360(int i) { return g_lds[0].cossin[i] >> 16; }
#endif

    // ---- xor shuffles ----
    template <int M>
    static __device__ __forceinline__ int shfl_xor(int v) {
        if constexpr (!kFast) {
            return __shfl_xor(v, M);
        } else if constexpr (M == 1) {
            return AECM_DPP(0, v, kDppQuadXor1, 0xf, 0xf, true);
        } else if constexpr (M == 2) {
            return AECM_DPP(0, v, kDppQuadXor2, 0xf, 0xf, true);
        } else if constexpr (M == 4) {
            int t = AECM_DPP(v, v, kDppRowShl4, 0xf, 0x5, false);   // banks 0,2 (bit 2 clear) <- lane+4
            return AECM_DPP(t, v, kDppRowShr4, 0xf, 0xa, false);    // banks 1,3 (bit 2 set)   <- lane-4
        } else if constexpr (M == 8) {
            return AECM_DPP(0, v, kDppRowRor8, 0xf, 0xf, true);
        } else {
            return __shfl_xor(v, M);
        }
    }

    // Re-pair FFT operands across lane bit Q (see tests/sim/wave_sim.h for the definition):
    //   lanes with bit Q clear: a stays, b <- partner's a;   lanes with bit Q set: b stays, a <- partner's b.
    template <int Q>
    static __device__ __forceinline__ void exchange(int &a, int &b) {
        if constexpr (kFast && Q == 5) {
            auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
            a = (int)r[0];
            b = (int)r[1];
        } else if constexpr (kFast && Q == 4) {
            auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
            a = (int)r[0];
            b = (int)r[1];
        } else if constexpr (kFast && Q == 3) {
            // bank = lane[3:2]; bit 3 set <=> banks 2,3.  DPP writes only the enabled banks, the rest keep `old`.
            const int na = AECM_DPP(a, b, kDppRowRor8, 0xf, 0xc, false);
            const int nb = AECM_DPP(b, a, kDppRowRor8, 0xf, 0x3, false);
            a = na;
            b = nb;
        } else if constexpr (kFast && Q == 2) {
            const int na = AECM_DPP(a, b, kDppRowShr4, 0xf, 0xa, false);   // banks 1,3 <- b of lane-4
            const int nb = AECM_DPP(b, a, kDppRowShl4, 0xf, 0x5, false);   // banks 0,2 <- a of lane+4
            a = na;
            b = nb;
        } else if constexpr (kFast && AECM_QUAD_EXCHANGE_SWIZZLE) {
            // The two quad stages: DPP row / bank masks cannot tell the lanes of a pair apart, so the DPP form is two moves
            // plus two selects.  ds_swizzle_b32 in quad-permute mode does the moves on the LDS crossbar (no LDS memory, no
            // vector-ALU slot): two selects remain.  The LDS pipe is 25 % busy (profiles/r03_lds_conflicts.md).
            const bool hi = (lane_id() >> Q) & 1;
            const int pb = __builtin_amdgcn_ds_swizzle(b, 0x8000 | (Q == 0 ? kDppQuadXor1 : kDppQuadXor2));
            const int pa = __builtin_amdgcn_ds_swizzle(a, 0x8000 | (Q == 0 ? kDppQuadXor1 : kDppQuadXor2));
            a = hi ? pb : a;
            b = hi ? b : pa;
        } else if constexpr (kFast) {
            const bool hi = (lane_id() >> Q) & 1;
            const int pb = AECM_DPP(0, b, Q == 0 ? kDppQuadXor1 : kDppQuadXor2, 0xf, 0xf, true);
            const int pa = AECM_DPP(0, a, Q == 0 ? kDppQuadXor1 : kDppQuadXor2, 0xf, 0xf, true);
            a = hi ? pb : a;
            b = hi ? b : pa;
        } else {
            const bool hi = (lane_id() >> Q) & 1;
            int send = hi ? a : b;
            int recv = __shfl_xor(send, 1 << Q);
            if (hi) a = recv;
            else b = recv;
        }
    }

    // The same for the N transforms that advance in lock step.  (A v_cndmask_b32_dpp form of the quad stages -- two VALU
    // per transform, mask moves on the scalar unit -- was measured slower and removed: profiles/r03_experiments.md.)
    template <int Q, int N>
    static __device__ __forceinline__ void exchange_all(int (&aa)[N], int (&bb)[N]) {
        for (int n = 0; n < N; ++n) exchange<Q>(aa[n], bb[n]);
    }

    // ---- reductions: every lane of a 16-lane row gets the row result (4 DPP steps), then the rows are
    // chained with row_bcast15 / row_bcast31 into lane 63 and read back with one v_readlane.
    // `identity` is what lanes not written by a masked DPP step see (op(v, identity) == v).
    template <class Op>
    static __device__ __forceinline__ int reduce(int v, int identity, Op op) {
        if constexpr (kFast) {
            // full masks + valid source lanes: `old` is dead, bound_ctrl:1 lets the DPP fold into the op
            v = op(v, AECM_DPP(identity, v, kDppQuadXor1, 0xf, 0xf, true));
            v = op(v, AECM_DPP(identity, v, kDppQuadXor2, 0xf, 0xf, true));
            v = op(v, AECM_DPP(identity, v, kDppRowHalfMirror, 0xf, 0xf, true));
            v = op(v, AECM_DPP(identity, v, kDppRowMirror, 0xf, 0xf, true));
            v = op(v, AECM_DPP(identity, v, kDppRowBcast15, 0xa, 0xf, false));   // rows 1,3 += rows 0,2
            v = op(v, AECM_DPP(identity, v, kDppRowBcast31, 0xc, 0xf, false));   // rows 2,3 += rows 0+1
            return __builtin_amdgcn_readlane(v, 63);
        } else {
            v = op(v, __shfl_xor(v, 1));
            v = op(v, __shfl_xor(v, 2));
            v = op(v, __shfl_xor(v, 4));
            v = op(v, __shfl_xor(v, 8));
            v = op(v, __shfl_xor(v, 16));
            v = op(v, __shfl_xor(v, 32));
            return __builtin_amdgcn_readfirstlane(v);
        }
    }
    static __device__ __forceinline__ int reduce_max(int v) { return reduce(v, (int)0x80000000, [](int a, int b) { return a > b ? a : b; }); }
    static __device__ __forceinline__ int reduce_min(int v) { return reduce(v, 0x7fffffff, [](int a, int b) { return a < b ? a : b; }); }
    static __device__ __forceinline__ int reduce_add(int v) { return reduce(v, 0, [](int a, int b) { return add(a, b); }); }

    // Several independent reductions sharing their butterfly steps.  exchange<5>(x, y) leaves
    // a = [x lanes 0..31 | y lanes 0..31], b = [x lanes 32..63 | y lanes 32..63], so op(a, b) holds x's
    // partials in the lower half of the wave and y's in the upper half; exchange<4> does the same
    // across the 16-lane rows.  Then the rows are reduced in place (4 DPP steps) and read back.
    template <class Op>
    static __device__ __forceinline__ int reduce_row(int v, int identity, Op op) {
        v = op(v, AECM_DPP(identity, v, kDppQuadXor1, 0xf, 0xf, true));
        v = op(v, AECM_DPP(identity, v, kDppQuadXor2, 0xf, 0xf, true));
        v = op(v, AECM_DPP(identity, v, kDppRowHalfMirror, 0xf, 0xf, true));
        return op(v, AECM_DPP(identity, v, kDppRowMirror, 0xf, 0xf, true));
    }
    template <class Op>
    static __device__ __forceinline__ void reduce2(int x, int y, int identity, Op op, int &rx, int &ry) {
        if constexpr (kFast) {
            exchange<5>(x, y);
            int v = reduce_row(op(x, y), identity, op);
            v = op(v, AECM_DPP(identity, v, kDppRowBcast15, 0xa, 0xf, false));   // rows 1,3 += rows 0,2
            rx = __builtin_amdgcn_readlane(v, 31);
            ry = __builtin_amdgcn_readlane(v, 63);
        } else {
            rx = reduce(x, identity, op);
            ry = reduce(y, identity, op);
        }
    }
    static __device__ __forceinline__ void reduce_max2(int x, int y, int &rx, int &ry) {
        reduce2(x, y, (int)0x80000000, [](int a, int b) { return a > b ? a : b; }, rx, ry);
    }
    // min over x and max over y in one pass: max(y) = -min(-y) (y != INT_MIN: callers pass means / keys >= 0)
    static __device__ __forceinline__ void reduce_min_max(int x, int y, int &min_x, int &max_y) {
        int neg_max;
        reduce2(x, neg(y), 0x7fffffff, [](int a, int b) { return a < b ? a : b; }, min_x, neg_max);
        max_y = neg(neg_max);
    }
    static __device__ __forceinline__ void reduce_add4(int x, int y, int z, int w, int &rx, int &ry, int &rz, int &rw) {
        if constexpr (kFast) {
            exchange<5>(x, y);
            exchange<5>(z, w);
            int xy = add(x, y), zw = add(z, w);      // [x | y] and [z | w], 32-lane partials
            exchange<4>(xy, zw);
            // rows: [x, z, y, w], 16-lane partials
            const int v = reduce_row(add(xy, zw), 0, [](int a, int b) { return add(a, b); });
            rx = __builtin_amdgcn_readlane(v, 15);
            rz = __builtin_amdgcn_readlane(v, 31);
            ry = __builtin_amdgcn_readlane(v, 47);
            rw = __builtin_amdgcn_readlane(v, 63);
        } else {
            rx = reduce_add(x); ry = reduce_add(y); rz = reduce_add(z); rw = reduce_add(w);
        }
    }

    static __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
    static __device__ __forceinline__ int readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
    static __device__ __forceinline__ int writelane(int v, int value, int lane) { return aecm_llvm_amdgcn_writelane(value, lane, v); }   // v_writelane_b32 (value and lane wave-uniform)
    static __device__ __forceinline__ int bpermute(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
    static __device__ __forceinline__ int shift_up1(int v, int fill) {
        if constexpr (kFast) {
            // in place (lane 0, which has no source, keeps its own value) and then lane 0 written from the scalar side: the
            // result can live in the register of v -- with `fill` as the old value it needs a register of its own, a move of
            // the fill into it and, for a history carried round the block loop, a copy back
            return writelane(AECM_DPP(v, v, kDppWaveShr1, 0xf, 0xf, false), fill, 0);
        } else {
            int t = __shfl_up(v, 1);
            return lane_id() == 0 ? fill : t;
        }
    }

    // floor(sqrt(x)) for 0 <= x <= 2^31: v_sqrt_f32 is within 1 ulp and float(x) within 2^-24
    // relative, i.e. the computed root is within 0.0083 of the true one; adding 0.02 (>= 0.0156 after
    // rounding at magnitude 2^15) makes the truncation land on floor or floor + 1, so one exact
    // downward correction suffices.  Verified exhaustively on the device by the self test.
    static __device__ __forceinline__ int isqrt31(int x) {
        const unsigned ux = (unsigned)x;
        unsigned r = (unsigned)(__builtin_amdgcn_sqrtf((float)ux) + 0.02f);
        r -= (r * r > ux) ? 1u : 0u;
        return (int)r;
    }

    // floor(n / d) for an unsigned 32-bit n and 1 <= d < 2^16 (the Wiener gain's WebRtcSpl_DivU32U16, reference
    // aecm_core_c.cc:584; for d == 0 the caller never looks at the result) in two float steps, 14 instructions instead of
    // the compiler's 20-instruction 32-by-32 expansion behind a zero test:
    //   r  = v_rcp_f32(d)                     d is exact in float, r within 1 ulp: relative error <= 2^-22 (margin included)
    //   q1 = trunc(float(n) * r)              relative error of the product <= 2^-24 + 2^-22 + 2^-24, so |q1 - n/d| <= 1537
    //   r1 = n - q1 * d                       exact modulo 2^32, and the true value fits: |r1| <= 1537 d < 2^27
    //   q2 = floor(float(r1) * r + 2^-10)     |r1 / d| <= 1538: absolute error < 6.7e-4 < 2^-10, so the biased value lies in
    //                                         (x, x + 2^-9) for x = r1 / d, whose fractional part is at most 1 - 2^-16:
    //                                         q2 is floor(x) or floor(x) + 1
    //   r2 = r1 - q2 * d in [-d, d)           negative exactly when q2 is one too large
    //   n / d = q1 + q2 + (r2 >> 31)
    // v_cvt_u32_f32 saturates (n / d near 2^32 with d == 1).  Checked on the device by the self test (every divisor, edge
    // and random dividends) against the integer division.
    static __device__ __forceinline__ int divu_u32_u16(int n, int d) {
        const float r = __builtin_amdgcn_rcpf((float)(unsigned)d);
        const float t = (float)(unsigned)n * r;
        int q1, q2;
        asm("v_cvt_u32_f32 %0, %1" : "=v"(q1) : "v"(t));
        const int r1 = sub(n, mul(q1, d));
        const float t1 = __builtin_fmaf((float)r1, r, 0x1p-10f);
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(q2) : "v"(t1));
        const int r2 = sub(r1, __mul24(q2, d));
        return add(add(q1, q2), sar(r2, 31));
    }

    // ---- memory ----
    // state words and history rows (see kCoherentState); the audio rows (load_i16 / store_i16) are read-only resp. write-only
    // inside a launch and stay plain
    static __device__ __forceinline__ int load_u32(const uint32_t *p, int idx) {
        if constexpr (kCoherent) return (int)__hip_atomic_load(p + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return (int)p[idx];
    }
    static __device__ __forceinline__ void store_u32(uint32_t *p, int idx, int v) {
        if constexpr (kCoherent) __hip_atomic_store(p + idx, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[idx] = (uint32_t)v;
    }
    static __device__ __forceinline__ void store_u32_if(bool lane_takes_part, uint32_t *p, int idx, int v) {
        if (lane_takes_part) store_u32(p, idx, v);
    }
    static __device__ __forceinline__ int load_i16(const int16_t *p, int idx) { return p[idx]; }
    static __device__ __forceinline__ int load_u16(const uint16_t *p, int idx) {
        if constexpr (kCoherent) return __hip_atomic_load(p + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return p[idx];
    }
    static __device__ __forceinline__ void store_i16(int16_t *p, int idx, int v) { p[idx] = (int16_t)v; }
    static __device__ __forceinline__ void store_u16(uint16_t *p, int idx, int v) {
        if constexpr (kCoherent) __hip_atomic_store(p + idx, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[idx] = (uint16_t)v;
    }
    // The 64 wave-uniform state words of a stream.  Plain: read where they are used (the compiler fetches them with scalar
    // loads).  Coherent: one lane vector (word f in lane f), each word then read out of its lane.
    struct ScalarRow {
        const int32_t *p;
        int v;
        __device__ __forceinline__ int get(int f) const {
            if constexpr (kCoherent) return __builtin_amdgcn_readlane(v, f);
            else return __builtin_amdgcn_readfirstlane(p[f]);
        }
    };
    static __device__ __forceinline__ ScalarRow load_scalar_row(const int32_t *scal) {
        if constexpr (kCoherent) return ScalarRow{scal, (int)__hip_atomic_load(scal + lane_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        else return ScalarRow{scal, 0};
    }
    // called by the wave's first lane only
    static __device__ __forceinline__ void store_scalar(int32_t *scal, int f, int v) {
        if constexpr (kCoherent) __hip_atomic_store(scal + f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else scal[f] = v;
    }
};

}  // namespace aecm
#endif  // AECM_AMD_WAVE_GFX950_H_
