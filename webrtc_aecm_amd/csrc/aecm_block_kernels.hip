// The block kernels of the MI355X AECM engine: aecm_process_kernel, one wavefront per stream.
//
// The whole persistent state of a stream (~40 lane vectors + ~50 scalars) is loaded into registers once, n_blocks blocks
// are processed back to back (WebRtcAecm_ProcessBlock-equivalents, aecm_wave.h), and the state is written back once.  Per
// block a wave touches 3 x 128 B of audio I/O (prefetched one block ahead), writes one 128-byte far-spectrum row and reads
// at most one.  The constant tables (FFT twiddles, comfort-noise cos/sin, sqrt-Hanning) are staged in LDS by the prologue.
// No MFMA: nothing here is a dense contraction.
//
// Compiled with -mllvm -structurizecfg-skip-uniform-regions (build.py: SOURCE_FLAGS).  Every branch of the block loop is
// wave-uniform (scalar conditions); the structurizer otherwise rewrites each if / else into guarded single-entry regions
// joined by flow blocks -- extra scalar mask logic, extra branches and duplicated code that the nested data-dependent
// short paths of aecm_wave.h multiply: 3294 -> 2933 instructions in the kernel, 928 -> 983 M frames/s
// (profiles/r03_experiments.md section 7).  For the same reason this unit is compiled with SimplifyCFG's phi-to-select
// folding and the speculative-execution pass turned off (build.py: KEEP_BRANCHES_FLAGS): a uniform branch that skips a
// block costs nothing, its select form executes both sides on the vector port that bounds the kernel (983 -> 1 010 M,
// section 9).  A translation unit of its own so that it compiles next to the tick kernels' unit (aecm_kernels.hip) and its
// flags can differ from theirs.
#define AECM_TABLE_ATTR __device__
#if defined(AECM_CHECKED)
#define g_aecm_check_fail g_aecm_check_fail_blocks      // device symbols are per translation unit (no relocatable device code)
#endif
#include "aecm_kernel_common.h"

namespace aecm {

#if defined(AECM_CHECKED)
__device__ unsigned long long g_aecm_check_fail_blocks[2];
#endif
// This unit's share of the audit counters (see ReadCheckCounters in aecm_kernels.hip).
hipError_t ReadBlockKernelCheckCounters(uint64_t counters[2], bool reset) {
#if defined(AECM_CHECKED)
    unsigned long long host[2] = {0, 0};
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_aecm_check_fail_blocks), sizeof host);
    if (e != hipSuccess) return e;
    counters[0] = host[0];
    counters[1] = host[1];
    if (reset) {
        const unsigned long long zero[2] = {0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_aecm_check_fail_blocks), zero, sizeof zero);
    }
    return e;
#else
    (void)counters;
    (void)reset;
    return hipErrorNotSupported;
#endif
}

// Occupancy target: the kernel is bound by instruction issue with every wave strictly in order, so
// resident waves are what hides one wave's latencies from the VALU port.  7 waves/SIMD = 72 VGPRs; the
// fast variants need 64 / 68 (no spills).  Measured in round 1: 5 -> 6 -> 7 waves = 609 -> 657 -> 674 M frames/s.
// The hardware's ceiling is 7 as well: a CU's LDS holds seven copies of the 21 KB tables (one per 4-wave workgroup).
#ifndef AECM_WAVES_PER_EU
#if defined(AECM_CHECKED)
#define AECM_WAVES_PER_EU 4       // the audit build's checks need registers; its speed does not matter
#else
#define AECM_WAVES_PER_EU 7
#endif
#endif
#ifndef AECM_MAX_WAVES_PER_EU
#define AECM_MAX_WAVES_PER_EU 8
#endif
// The rotation variants only ever run launches of at most kRotationWavesPerEu waves per SIMD (LaunchProcessBlocks): they
// are built for that occupancy and get the larger register budget (80 VGPRs) that goes with it.
#ifndef AECM_ROTATION_WAVES_PER_EU
#if defined(AECM_CHECKED)
#define AECM_ROTATION_WAVES_PER_EU 4
#else
#define AECM_ROTATION_WAVES_PER_EU 6
#endif
#endif
template <bool kFast, bool kHasClean, bool kPhasePrio = true>
__global__ __launch_bounds__(64 * kWavesPerWorkgroup)
__attribute__((amdgpu_waves_per_eu(kPhasePrio ? AECM_WAVES_PER_EU : AECM_ROTATION_WAVES_PER_EU, AECM_MAX_WAVES_PER_EU)))
void aecm_process_kernel(StatePtrs st, IoView io, int n_streams, int n_blocks, const int32_t *blocks_per_stream) {
    FillLdsTables<64 * kWavesPerWorkgroup>(st.consts);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t stream = (int64_t)blockIdx.x * kWavesPerWorkgroup + wave;
    if (stream >= n_streams) return;
    if (blocks_per_stream) {
        n_blocks = __builtin_amdgcn_readfirstlane(blocks_per_stream[stream]);
        if (n_blocks <= 0) return;
    }
    BlockEngine<Gfx950Wave<kFast, kPhasePrio>, kHasClean>::run_stream(st, io, stream, n_blocks);
}

int RotationStreamLimit(int compute_units) {
    return (compute_units > 0 ? compute_units : 256) * 4 * AECM_ROTATION_WAVES_PER_EU;
}

hipError_t LaunchProcessBlocks(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, int variant,
                               int rotation_stream_limit, hipStream_t stream, const int32_t *blocks_per_stream) {
    if (n_streams <= 0 || n_blocks <= 0) return hipSuccess;
    const dim3 grid((n_streams + kWavesPerWorkgroup - 1) / kWavesPerWorkgroup);
    const dim3 block(64 * kWavesPerWorkgroup);
    const size_t lds = sizeof(LdsTables);
    const bool clean = io.near_clean != nullptr;
    // Issue priority by phase of the block when the launch is more waves than the chip holds at once (they then run in
    // rounds and spread over the phases by themselves), the per-block rotation when every wave of the launch is resident
    // from the start and they would otherwise march in lock step (wave_gfx950.h: kPhasePrio).  The limit belongs to the
    // engine's device (RotationStreamLimit of its CU count, taken once in BatchEngine::Create): nothing cached here.
    const bool phase = n_streams > rotation_stream_limit;
#define AECM_LAUNCH(F, C, P) hipLaunchKernelGGL((aecm_process_kernel<F, C, P>), grid, block, lds, stream, st, io, n_streams, n_blocks, blocks_per_stream)
    if (variant == kVariantFast) {
        if (clean) { if (phase) AECM_LAUNCH(true, true, true); else AECM_LAUNCH(true, true, false); }
        else { if (phase) AECM_LAUNCH(true, false, true); else AECM_LAUNCH(true, false, false); }
    } else {
        if (clean) AECM_LAUNCH(false, true, false);
        else AECM_LAUNCH(false, false, false);
    }
#undef AECM_LAUNCH
    return hipGetLastError();
}

}  // namespace aecm
