// The block kernels of the MI355X AECM engine.  One block DSP (BlockEngine, aecm_wave.h) under three launch forms:
//   aecm_process_kernel            one wavefront per stream for the whole launch (described next)
//   aecm_process_queue_kernel      launches larger than the chip: (chunk, stream) items claimed in order by a resident grid
//   aecm_process_pipelined_kernel  launches the chip holds at once: six waves per four streams, transforms one block ahead
// (aecm_engine.cpp: LaunchBlocks picks by the size of the launch; the forms give identical results.)
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
#include <type_traits>

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

// ---- the chunk-queue form: one launch, (chunk, stream) items claimed in order by resident waves -------------------------
// A launch of S streams x T blocks in the form above gives every wave one stream for all T blocks.  Waves start when a slot
// frees up, so when the last stream has been handed out the resident waves are anywhere between their first and their last
// block, and the launch drains for most of a stream's duration at falling occupancy -- a SIMD below ~5 waves no longer
// fills its issue ports: measured 1.7-2.3 ms of an 83 ms launch at 65 536 streams, the same 2 ms of a 22 ms launch at 16 384
// (profiles/r04_experiments.md section 1).  Here the launch is cut into chunks of C blocks, item i = (chunk i / S, stream
// i % S), and a grid that just fills the chip claims items in order from one counter: all streams advance together, and
// what is left when the counter runs out is at most C blocks per wave -- the drain shrinks by T / C.
//
// A stream's chunk c + 1 is picked up by whichever wave comes next, on any CU of any XCD, so between chunks the state
// travels through memory at agent scope (Gfx950Wave<.., kCoherentState>: sc1 loads and stores, no fences -- a release
// fence at agent scope is a write-back of the XCD's whole L2).  Order: the claimer of item i - S holds it before item i is
// claimed (one counter), so a wave that finds done[stream] < chunk waits for a wave that is running -- no deadlock even
// when the grid is larger than the chip; with S >= the resident waves the wait practically never happens.  The wait is
// bounded anyway: a wave that gives up raises *err (never cleared by a launch; the engine reports it at the next
// synchronisation) and leaves.
// What the hand-over rests on.  This is the write-through publish form of the platform guide (cdna_hip_programming.md section 6,
// Guideline 16 / R1, and MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility: valid forms"):
//   producer: every byte of the payload (state words, scalars, history rows) leaves the wave as an sc1 store -- a relaxed
//             agent-scope __hip_atomic_store of 4 / 2 bytes per lane, the kernel's natural store width, which the guide names as
//             the right form for such epilogues; sc1 stores are written through to memory and DROP the line from the XCD's L2 --
//             then `asm volatile("s_waitcnt vmcnt(0)")` in the publishing wave (inline asm: the compiler cannot drop or move
//             it), then ONE relaxed agent-scope flag store;
//   consumer: relaxed polls of the flag, then sc1 loads of the payload ("sc1 loads may replace the acquire only when the producer
//             stored sc1": they bypass the CU's L1 and are served where the eight XCDs agree).
// One wave produces and one wave consumes a stream's chunk, so "every writing wave drains" is that one s_waitcnt.  No
// buffer_wbl2 / buffer_inv anywhere: a release fence is a write-back of the XCD's whole L2 (the audio output rows of every
// resident wave), an acquire drops the CU's L1 under three other workgroups.  Measured cost of the fenced form
// (-DAECM_QUEUE_FENCES=1: release store / acquire poll on done[]; the same results): 65 536 streams 1 053 -> 1 030 M frames/s,
// 8 192 streams (32-block chunks) 983 -> 899 M.  The build switch stays for a platform whose sc1 semantics differ;
// tests/test_gpu_parity.py::test_chunk_queue_half_a_million_hand_overs and tools/soak_parity.py are the gate either way.
#ifndef AECM_QUEUE_FENCES
#define AECM_QUEUE_FENCES 0
#endif
constexpr int kQueueAcquire = AECM_QUEUE_FENCES ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED;
constexpr int kQueueRelease = AECM_QUEUE_FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED;
constexpr int kQueueCtlWords = 16;          // [0] next item, then done[stream] = chunks of this launch completed
// A wait is for a wave that is running a chunk, so its bound grows with the chunk: 2^20 polls (x (s_sleep 16 = 1 024 cycles +
// one load): a second or two) or 64 polls per block of the chunk, whichever is more (a block takes a wave ~7 us on a full chip:
// ~3 polls) -- a healthy predecessor on a shared or debugged GPU is not mistaken for a hung one.
__device__ __forceinline__ uint32_t QueueMaxPolls(int chunk_blocks) {
    const uint32_t by_chunk = (uint32_t)chunk_blocks << 6;         // chunk_blocks <= 2^20 (BatchEngine::kMaxQueueChunk)
    return by_chunk > (1u << 20) ? by_chunk : (1u << 20);
}

template <bool kHasClean>
__global__ __launch_bounds__(64 * kWavesPerWorkgroup)
__attribute__((amdgpu_waves_per_eu(AECM_WAVES_PER_EU, AECM_MAX_WAVES_PER_EU)))
void aecm_process_queue_kernel(StatePtrs st, IoView io, int n_streams, int n_blocks, int chunk_blocks, int n_chunks, uint32_t *ctl,
                               uint32_t *err) {
    FillLdsTables<64 * kWavesPerWorkgroup>(st.consts);
    using E = BlockEngine<Gfx950Wave<true, true, false, true>, kHasClean>;
    const uint32_t n_items = (uint32_t)n_streams * (uint32_t)n_chunks;
    uint32_t *done = ctl + kQueueCtlWords;
    // Every lane takes part in the queue's few memory operations with the same address and, for the counter, an addend
    // that is 1 in lane 0 only: no lane-divergent control flow anywhere in the item loop.
    const uint32_t one_in_lane0 = (threadIdx.x & 63u) == 0 ? 1u : 0u;
    for (;;) {
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_fetch_add(ctl, one_in_lane0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (item >= n_items) break;
        if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
        const uint32_t chunk = item / (uint32_t)n_streams;
        const uint32_t stream = item - chunk * (uint32_t)n_streams;
        if (chunk != 0) {
            uint32_t polls = 0;
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(done + stream, kQueueAcquire, __HIP_MEMORY_SCOPE_AGENT)) < chunk) {
                __builtin_amdgcn_s_sleep(16);
                if (++polls > QueueMaxPolls(chunk_blocks) ||
                    ((polls & 1023u) == 0 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)) {
                    __hip_atomic_store(err, 1u + stream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every wave leaves at its next claim
                    return;
                }
            }
        }
        asm volatile("" ::: "memory");                       // the state loads stay behind the flag
        const int first = (int)chunk * chunk_blocks;
        const int nb = n_blocks - first < chunk_blocks ? n_blocks - first : chunk_blocks;
        typename E::StridedIo sio{io, (int64_t)stream * io.stream_stride + (int64_t)first * io.block_stride};
        E::run_stream_io(st, sio, (int64_t)stream, nb);
        // every store of the chunk (state, history rows: sc1, written through) has completed before the flag is raised
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(done + stream, chunk + 1u, kQueueRelease, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the pipelined form: launches the chip holds at once ------------------------------------------------------------
// With one wavefront per stream a launch of S <= 4 096 streams puts at most 4 waves on a SIMD, every one of them alternating
// between a stretch of dense vector work (the transforms) and a long scalar-heavy stretch (delay estimator, energies,
// channel bookkeeping, gains): 4 096 streams run at 75 % of the rate the issue ports sustain with 7 waves.  But a block's
// first third -- windows, forward transforms and magnitudes (BlockEngine::front_block, 36 % of its vector instructions) --
// depends on the input samples alone, not on the adaptive state.  Here a workgroup of SIX waves serves FOUR streams:
// waves 0..3 run back_block for one stream each, waves 4 and 5 run front_block for two streams each, one block ahead, and
// hand the spectra over through LDS (two buffers per stream, one barrier per block: while the back waves read the spectra
// of block b the front waves write those of block b + 1).  Per stream the sequential part of a block shrinks to the back
// part, and a SIMD holds 6 waves instead of 4 -- two of them nearly pure vector work that overlaps the others' scalar
// stretches by construction.  4 workgroups per CU (24 waves, 4 x 27 KB of LDS) = 4 096 streams on 256 CUs, all resident.
#ifndef AECM_PIPE_FRONT_PRIO
#define AECM_PIPE_FRONT_PRIO 0        // the front waves' issue priority (the back waves': by phase, 1..3).  Also in the sixteen-wave shape,
                                      // where the front waves are the longest link: 1 / 2 / 3 there cost 10-13 % (1 024 streams 613 -> 549 / 549 / 531)
#endif
constexpr int kPipeStreams = 4;
// Slot rotations: PipeShape::rot (the kernel's slot_of), chosen per shape in PipelinedShapeFor.  There is none per workgroup of a CU by
// default (rot bits 6-7): the hardware does that one itself -- traced (tools/pipe_trace.py, cu_mates), the second and third workgroup
// of a CU each start one SIMD further round the cycle 0, 2, 1, 3 than the one before; a software rotation on top brings the same
// slots back onto the same SIMDs (1 024 streams as 2 x 2 per CU: 620 -> 493 M frames/s).
// kFront front waves per workgroup: 2 (two streams each; the form for launches that fill the chip) or 4 (one stream each: ten-wave
// workgroups with two tail waves, two to a CU -- launches of up to 2 048 streams, where the chip has wave slots to spare and a
// front wave with two streams' transforms is the longest link of the chain).
// A third role (round 5): tail_block -- inverse transform, synthesis window, overlap-add, the output store: 19 % of a block's
// vector instructions and a long chain of LDS table reads and lane exchanges -- in kTail "tail" waves of their own, one block
// BEHIND the middle waves (which then run middle_block only).  kTail = 1: one wave for the workgroup's four streams, seven
// waves per workgroup = 28 per CU with four workgroups, every wave slot of the SIMDs taken; kTail = 2: two waves of two
// streams, eight waves per workgroup, three workgroups per CU (launches of up to 3 072 streams).  The stream's sequential part
// shrinks once more (what a small launch is bound by) and a SIMD gets one more wave of dense vector work to fill its port with.
// A fourth role (round 5, small launches): delay_block -- both binary spectra, the bit histories, the 100 means; a chain of
// reductions and scalar decisions on state of its own, a sixth of the middle wave's instructions -- in kDelay "delay" waves, one
// stream each, one block AHEAD of the middle waves (between them and the front waves): the spectra then live for three steps
// (three slots per stream) and the delay reaches the middle wave through LDS.
// A fifth (smallest launches: at most one workgroup per CU): the back wave's remaining work in its two halves, channel_block
// (far history ... channel update: what the next block's echo estimate waits for) and gain_block (suppression gain ... comfort
// noise: fed by the channel half, otherwise state of its own), the second in kGain "gain" waves one block behind the first.
// Sixteen waves per four streams then: 4 channel, 4 front, 2 tail, 2 delay (two streams each), 4 gain -- the longest link of the
// chain is half of what it was.
constexpr int PipeWaves(int tail_waves, int front_waves = 2, int delay_waves = 0, int gain_waves = 0) {
    return kPipeStreams + front_waves + tail_waves + delay_waves + gain_waves;
}
struct PipeSlot {           // the spectra of one block of one stream on their way from the front to the back wave
    int near_x[kLanes];     // near-end spectrum, bins 0..63: re | im << 16 (im conjugated as the block path uses it)
    int mags[kLanes];       // far-end magnitude | near-end magnitude << 16 (both <= 46 340)
    int scalars[kLanes];    // lanes 0..4: far mag[64], far Q, near re[64], near mag[64], near Q
};
struct PipeRawSlot {        // the same hand-over BEFORE the spectra are formed (kRaw instantiations): the transforms' outputs
    int fa[2][kLanes], fb[2][kLanes];     // BlockEngine::front_transforms: far end [0], near end [1]
    int q[2], pad[2];
};
struct PipeTailSlot {       // the residual spectrum of one block of one stream on its way from the middle to the tail wave
    int a[kLanes], b[kLanes];   // BlockEngine::TailInput
    int clean_q, pad[3];
};
struct PipeGainSlot {       // BlockEngine::GainInput on its way from the channel to the gain wave
    int echo_est[kLanes];
    int echo_est64, far_q, cur_vad, near0, stored0, pad[3];
};
struct PipeGainState {      // the gain wave's part of the stream state on its way to the channel wave, which stores the state (end of the launch)
    int echo_filt[kLanes], near_filt_ctrs[kLanes], noise_est[kLanes];      // (near_filt | low_ctr << 16 | high_ctr << 19: the V_NEARFILT layout)
    int scal[16];
};
template <int kTail, bool kRaw = false, int kDelay = 0, int kGain = 0>
struct PipeShared {
    // [block modulo the ring: 2, + 1 with delay waves, + 1 with gain waves][stream of the workgroup]
    typename std::conditional<kRaw, PipeRawSlot, PipeSlot>::type slots[2 + (kDelay ? 1 : 0) + (kGain ? 1 : 0)][kPipeStreams];
    PipeTailSlot tails[kTail ? 2 : 1][kTail ? kPipeStreams : 1];
    int delays[2][kPipeStreams];          // delay_block's result on its way from the delay to the middle wave ...
    int far_rows[kDelay ? 2 : 1][kDelay ? kPipeStreams : 1][kLanes];      // ... and the far-history row that goes with it (AlignedFarend)
    PipeGainSlot gains[kGain ? 2 : 1][kGain ? kPipeStreams : 1];
    PipeGainState gain_state[kGain ? kPipeStreams : 1];
    int ahead;                            // this workgroup leads the launch's slowest one by more than the allowed lead (balance, below)
    int level;                            // the front waves' base priority for the current group of blocks (balance modes 2, 3)
};
#ifndef AECM_PIPE_TAIL_PRIO
#define AECM_PIPE_TAIL_PRIO 1         // the tail waves' issue priority
#endif
// Which launches get delay waves when the caller does not say (workgroups of four streams the launch makes, CUs of the device): those the
// sixteen-wave shape takes (below).
#ifndef AECM_PIPE_DELAY_DEFAULT
#define AECM_PIPE_DELAY_DEFAULT(n_wg, cus) ((n_wg) <= 2 * (cus) ? kPipeStreams : 0)
#endif
// Workgroups of the sixteen-wave shape a CU takes (experiments: two of them are 32 waves, eight per SIMD -- the kernel's 61 VGPRs allow it)
#ifndef AECM_PIPE_GAIN_WGS_PER_CU
#define AECM_PIPE_GAIN_WGS_PER_CU (PipeWorkgroupsPerCu<2, 4, 2, 4>())
#endif
// Which launches get gain waves when the caller does not say: up to two sixteen-wave workgroups per CU (eight streams per CU).
// Round 5 gave this shape one workgroup per CU only (1 024 streams 492 -> 612 M frames/s with the gain waves, 256 streams 124 -> 155)
// because two of them measured 620 M at 2 048 streams against the ten-wave shape's 740: with 81 SGPRs the two never shared a CU
// (PipeWavesPerEu above), and with every one-stream role of a slot on the same SIMD a CU of five streams ran them at 0.43 instead of
// 0.60 M frames/s each.  Built for eight waves per SIMD and with the roles staggered over the SIMDs (the kernel's slot_of) the shape
// carries a CU's eight streams as well as the ten-wave shape and everything below better (profiles/r06_experiments.md):
// 1 280 streams 482 -> 650, 1 536: 571 -> 688, 1 792: 647 -> 702, 2 048: 738 / 734.
#ifndef AECM_PIPE_GAIN_DEFAULT
#define AECM_PIPE_GAIN_DEFAULT(n_wg, cus) ((n_wg) <= 2 * (cus) ? kPipeStreams : 0)
#endif
#ifndef AECM_PIPE_GAIN_PRIO
#define AECM_PIPE_GAIN_PRIO 3         // the gain waves' issue priority (1 / 2 / 3: 1 024 streams 594 / 607 / 613 M frames/s)
#endif
#ifndef AECM_PIPE_DELAY_PRIO
#define AECM_PIPE_DELAY_PRIO 1        // the delay waves' issue priority
#endif

// Balance.  Every workgroup of a pipelined launch is resident from the start and has the same amount of work, but the SIMD's
// arbiter serves the highest priority first and then its OLDEST wave: the workgroups dispatched first pull ahead, finish
// early, and the launch ends at low occupancy (round 4: 4.71 of the 6 placed waves resident on average, vector port 76 %).
// Progress feedback: blocks are counted in groups of 2^AECM_PIPE_BALANCE_GROUP_LOG2.  At every group boundary the first
// front wave of a workgroup (the "monitor": front waves finish their step early and would otherwise just park at the
// barrier) publishes the workgroup's group count in its own word of progress[] (a write-through store, no atomic), reads
// everybody's words (n_workgroups / 64 loads of 64 lanes, issued at the top of its step and looked at at the end of it) and
// takes their minimum: a workgroup more than AECM_PIPE_BALANCE_LEAD groups ahead of the slowest one runs its back waves with
// lowered phase priorities (Gfx950Wave<.., kDynamicPrio>) for the next group -- they still issue whenever the others leave a
// slot free (a lowered priority is work-conserving, a sleeping wave is not), but no longer win ties.  The flag reaches the
// other waves through LDS behind the barrier every wave executes anyway.  Stale words only make the verdict late.
#ifndef AECM_PIPE_BALANCE
#define AECM_PIPE_BALANCE 2
#endif
#ifndef AECM_PIPE_BALANCE_GROUP_LOG2
#define AECM_PIPE_BALANCE_GROUP_LOG2 4
#endif
#ifndef AECM_PIPE_BALANCE_LEAD
#define AECM_PIPE_BALANCE_LEAD 1
#endif
// AECM_PIPE_BALANCE: 0 off; 1 the back waves of a workgroup that is ahead run demoted (and its front waves at
// AECM_PIPE_FRONT_PRIO instead of .._BEHIND, if those differ); 2 only the front waves' priority follows the flag -- a
// workgroup advances at the pace of its front waves (the back waves run at higher priorities and park at the barrier until
// the spectra of the next block are there: measured 30 % of their time at 4 096 streams), so the front waves are the handle.
// AECM_PIPE_BALANCE_RULE: which workgroups take the LOW front priority: 0 those more than LEAD groups ahead of the slowest,
// 1 those less than LEAD groups behind the fastest (i.e. only the stragglers are raised).  AECM_PIPE_BALANCE 3: like 2 with
// one level per group of lead (proportional instead of on / off).
#ifndef AECM_PIPE_FRONT_PRIO_BEHIND
#if AECM_PIPE_BALANCE == 3
#define AECM_PIPE_FRONT_PRIO_BEHIND 2
#elif AECM_PIPE_BALANCE == 2
#define AECM_PIPE_FRONT_PRIO_BEHIND 1
#else
#define AECM_PIPE_FRONT_PRIO_BEHIND AECM_PIPE_FRONT_PRIO     // the front waves' priority while their workgroup is NOT ahead
#endif
#endif
#ifndef AECM_PIPE_MONITOR_SAMPLE
#define AECM_PIPE_MONITOR_SAMPLE 0
#endif
#ifndef AECM_PIPE_BALANCE_RULE
#define AECM_PIPE_BALANCE_RULE 0
#endif
// A front wave's priority rises by AECM_PIPE_FRONT_SECOND_BOOST while it works on its second stream (the rule of the back
// waves' phase table -- the priority rises with the progress through the step -- applied to the front waves).
#ifndef AECM_PIPE_FRONT_SECOND_BOOST
#define AECM_PIPE_FRONT_SECOND_BOOST 1
#endif
#ifndef AECM_PIPE_FRONT_SECOND_BOOST_BALANCED
#define AECM_PIPE_FRONT_SECOND_BOOST_BALANCED 0
#endif
__device__ __forceinline__ void SetPrioDynamic(int p) {      // s_setprio takes an immediate
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
constexpr int kPipeGroupLog2 = AECM_PIPE_BALANCE_GROUP_LOG2, kPipeGroupMask = (1 << kPipeGroupLog2) - 1;
constexpr int kPipeTraceWaves = 16;       // per-wave records per workgroup in the diagnostics build (the largest workgroup has 14)
constexpr int kPipeMonitorLoads = 10;     // x 64 lanes x 2 halves: launches of up to 1 280 workgroups are balanced, larger ones run as before

// (With delay waves everything behind the front waves is one step later: the delay waves work on block s - 1, the middle
// waves on block s - 2 out of slots[(s - 2) % 3], the tail waves on block s - 3; n_blocks + 3 barriers.)
// Synchronisation: ONE workgroup barrier per step.  In step s the front waves write the spectra of block s into slots[s & 1],
// the middle (back) waves work on block s - 1 out of slots[(s - 1) & 1] and, with tail waves, leave its residual spectrum in
// tails[(s - 1) & 1], which the tail waves turn into output samples in step s + 1.  Every wave of the workgroup executes
// n_blocks + 1 + (kTail ? 1 : 0) barriers whatever its role and whether or not its streams exist (a role's idle steps are the
// bare barriers in front of / behind its block loop).
// (Measured against rings with per-stream counters and no barrier -- pairs of waves that only wait for each other:
// slower, 4 096 streams 783 vs 818 M frames/s.  A wave parked at a barrier costs nothing; a wave polling a counter costs
// issue slots, and at the priority its last phase left it with it starves the wave it is waiting for.)
// Seven-wave workgroups (kTail = 1) are built for EIGHT waves per SIMD: four of them per CU are 28 waves, exactly the 4 x 7 slots
// the usual budget leaves -- and the dispatcher does not find that exact fit (a workgroup's seven waves go 2 + 2 + 2 + 1 over the
// SIMDs): measured, the fourth workgroup of every CU only started when the first had finished.  At 8 slots per SIMD (<= 64
// VGPRs, which the kernel needs anyway, and <= 80 SGPRs) there is room to spare.
#ifndef AECM_PIPE_TAIL1_WAVES_PER_EU
#define AECM_PIPE_TAIL1_WAVES_PER_EU 8
#endif
// kBalance: the progress feedback of the front waves' priority (above) is compiled in; launches that do not use it (fewer than
// four workgroups per CU) take the instantiation without it -- the same kernel with its monitor and its run-time priority
// levels switched off at run time measured 2 % slower there.
// kRaw: the front waves hand over the transforms' raw outputs and the back waves form the spectra (bin order, magnitudes: 40 of a
// stream's 198 front-part vector instructions) themselves.  For the balanced instantiation: with four workgroups per CU a
// workgroup advances at the pace of its front waves while its back waves sit at the barrier for a third of their time.
#ifndef AECM_PIPE_RAW_HANDOVER
#define AECM_PIPE_RAW_HANDOVER 1
#endif
// The occupancy a shape is built for.  Eight waves per SIMD need <= 64 VGPRs AND <= 80 SGPRs (MI355X_MICROARCH.md "Residency": with
// 82-96 SGPRs the hardware admits seven, whatever the compiler's occupancy line says): the sixteen-wave shape fits both without a
// spill, and two of its workgroups on a CU are exactly the 32 wave slots.  (Round 5 measured "two sixteen-wave workgroups per CU" as
// 620 M at 2 048 streams -- with 81 SGPRs they never were on a CU together; they took turns.)
#ifndef AECM_PIPE16_WAVES_PER_EU
#define AECM_PIPE16_WAVES_PER_EU 8
#endif
#ifndef AECM_PIPE10_WAVES_PER_EU
#define AECM_PIPE10_WAVES_PER_EU AECM_WAVES_PER_EU
#endif
#ifndef AECM_PIPE8_WAVES_PER_EU
#define AECM_PIPE8_WAVES_PER_EU AECM_WAVES_PER_EU
#endif
#ifndef AECM_PIPE6_WAVES_PER_EU
#define AECM_PIPE6_WAVES_PER_EU AECM_WAVES_PER_EU
#endif
constexpr int PipeWavesPerEu(int tail_waves, int front_waves, int delay_waves, int gain_waves) {
#if defined(AECM_CHECKED)
    return AECM_WAVES_PER_EU;
#else
    return PipeWaves(tail_waves, front_waves, delay_waves, gain_waves) == 7 ? AECM_PIPE_TAIL1_WAVES_PER_EU
           : gain_waves != 0 ? AECM_PIPE16_WAVES_PER_EU
           : delay_waves != 0 ? AECM_WAVES_PER_EU
           : tail_waves == 0 ? AECM_PIPE6_WAVES_PER_EU
           : front_waves == 4 ? AECM_PIPE10_WAVES_PER_EU : AECM_PIPE8_WAVES_PER_EU;
#endif
}
template <int kTail, bool kBalance, bool kRaw = false, int kFront = 2, int kDelay = 0, int kGain = 0>
__global__ __launch_bounds__(64 * PipeWaves(kTail, kFront, kDelay, kGain))
__attribute__((amdgpu_waves_per_eu(PipeWavesPerEu(kTail, kFront, kDelay, kGain), AECM_MAX_WAVES_PER_EU)))
void aecm_process_pipelined_kernel(StatePtrs st, IoView io, int streams_base, int streams_rem, int n_blocks, uint32_t *progress, int n_workgroups,
                                    int wgs_per_round, int rot, int prio) {
    constexpr int kMode = kBalance ? AECM_PIPE_BALANCE : 0;               // AECM_PIPE_BALANCE's meaning, per instantiation
    constexpr int kFrontBehind = kBalance ? AECM_PIPE_FRONT_PRIO_BEHIND : AECM_PIPE_FRONT_PRIO;
    // the second-stream boost is for the launches without balance: on top of it, it costs (4 096 streams: 862 M frames/s without, 844 with)
    constexpr int kBoost = kBalance ? AECM_PIPE_FRONT_SECOND_BOOST_BALANCED : AECM_PIPE_FRONT_SECOND_BOOST;
#if defined(AECM_PIPE_TRACE)     // diagnostics build: per wave, when it started / ended (100 MHz wall clock) and how long it sat at barriers (shader clocks)
    const uint64_t trace_t0 = wall_clock64(), trace_c0 = clock64();
    uint64_t trace_wait = 0;
#define AECM_PIPE_BARRIER() do { const uint64_t c_ = clock64(); __syncthreads(); trace_wait += clock64() - c_; } while (0)
#else
#define AECM_PIPE_BARRIER() __syncthreads()
#endif
    static_assert(kDelay == 0 || (kPipeStreams % kDelay == 0 && !kRaw), "delay waves: in the shapes with formed spectra");
    static_assert(kGain == 0 || (kGain == kPipeStreams && kDelay != 0), "gain waves: one per stream, with delay waves");
    constexpr int kWaves = PipeWaves(kTail, kFront, kDelay, kGain), kPipeFrontWaves = kFront, kPipeStreamsPerFront = kPipeStreams / kFront;
    constexpr int kLagD = kDelay ? 1 : 0, kLagG = kGain ? 1 : 0;           // steps the delay / gain waves put between the front waves and the rest
    constexpr int kSlots = 2 + kLagD + kLagG;
    PipeShared<kTail, kRaw, kDelay, kGain> &sh = *reinterpret_cast<PipeShared<kTail, kRaw, kDelay, kGain> *>(&g_lds[1]);        // behind the tables
    if (kMode != 0 && threadIdx.x == 0) { sh.ahead = 0; sh.level = kFrontBehind; }
    FillLdsTables<64 * kWaves>(st.consts);                              // ends in a barrier
    using W = Gfx950Wave<true, true, false, false, kMode == 1>;
    using E = BlockEngine<W, false>;
    using EF = BlockEngine<Gfx950Wave<true, false>, false>;               // the front and tail waves keep one priority (no per-phase s_setprio)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // This workgroup's streams (PipeSplit, below): streams_base of them, one more in the first streams_rem workgroups -- the
    // dispatcher deals workgroups out to the CUs in turn, so the CUs' loads differ by at most one stream.  A workgroup with fewer
    // than kPipeStreams streams keeps them in the slots that spread them over the waves that serve two slots each (front, tail,
    // delay waves): 1 -> slot 0; 2 -> slots 0, 2; 3 -> slots 0, 1, 2.  The waves of an empty slot only keep the barriers.
    const int wg = (int)blockIdx.x;
    const int64_t first = (int64_t)wg * streams_base + (wg < streams_rem ? wg : streams_rem);
    const int live_mask = (0xf7510 >> (4 * (streams_base + (wg < streams_rem ? 1 : 0)))) & 0xf;
    const auto slot_live = [&](int k) -> bool { return ((live_mask >> k) & 1) != 0; };
    const auto slot_stream = [&](int k) -> int64_t { return first + __builtin_popcount((unsigned)(live_mask & ((1 << k) - 1))); };
    // Which wave serves which slot.  The hardware deals a workgroup's waves out to the CU's four SIMDs in turn, so with the natural
    // numbering (wave = role's first wave + slot) every one-stream-per-wave role of a slot would sit on the same SIMD: a workgroup
    // with an empty slot leaves one SIMD idle and two workgroups with the same live slots crowd the same SIMDs -- a SIMD's vector
    // port carries the instruction stream of one stream's waves at 0.9 M frames/s and no more (measured: one five-stream CU among
    // four-stream CUs ran its streams at 0.43 instead of 0.60 M frames/s and held the whole launch up).  So the roles are
    // staggered (the front wave of slot k sits one SIMD further than its back wave, the gain wave two) and the workgroups that
    // share a CU -- workgroup i and i + wgs_per_round, ... -- each start one SIMD further.  Any bijection is correct.
    const int wg_rot = wgs_per_round > 0 ? (wg / wgs_per_round) * ((rot >> 6) & 3) : 0;
    const int rot_front = rot & 3, rot_gain = (rot >> 2) & 3, rot_delay = (rot >> 4) & 3, rot_tail = (rot >> 8) & 3;
    int k0 = 0;                                                                              // the first slot of a wave that serves several
    const auto ks = [&](int k) -> int { return (k0 + k) & (kPipeStreams - 1); };
    const auto slot_of = [&](int j, int role_rot) -> int { return (j + role_rot + wg_rot) & (kPipeStreams - 1); };
    static_assert((kPipeStreams & (kPipeStreams - 1)) == 0, "slot rotation");
    if (wave < kPipeStreams) {
        // ---- back (middle) wave: one stream, everything of a block after the forward transforms (and before the inverse one, with tail waves) ----
        typename E::Regs r;
        E::init_lane_constants(r, st.consts);
        const int slot = slot_of(wave, 0);
        const int64_t stream = slot_stream(slot);
        const bool live = slot_live(slot);
        uint32_t *vec = st.vec + stream * (int64_t)kVecWordsPerStream;
        int32_t *scal = st.scal + stream * (int64_t)kNumScal;
        uint16_t *hist = st.hist + stream * (int64_t)kHistWordsPerStream;
        typename E::StridedIo sio{io, stream * io.stream_stride};
        if (live) E::load_state(r, vec, scal);
        r.u.prio_drop = 0;
        W::begin_stream();
        AECM_PIPE_BARRIER();                                              // step 0: the spectra of block 0 are in slots[0]
        if (kDelay != 0) AECM_PIPE_BARRIER();                             // step 1: the delay waves' first
        int slot_idx = 0;                                                 // blk mod kSlots
        for (int blk = 0; blk < n_blocks; ++blk) {                        // step blk + 1 (+ 1 with delay waves)
            // Balance, mode 1 (see above): the monitor wrote the flag at the end of its step blk, which ran next to this wave's
            // block blk - 1 and ended in the barrier this wave has just passed.
            if (kMode == 1 && (blk & kPipeGroupMask) == 0 && blk != 0) r.u.prio_drop = __builtin_amdgcn_readfirstlane(sh.ahead);
            if (live) {
                const int lane = W::lane_id();
                typename E::Spectrum xf, df;
                r.table_index = W::table_index_for_this_block();
                if constexpr (kRaw) {
                    const PipeRawSlot &slot_in = sh.slots[slot_idx][slot];
                    const int fa0 = slot_in.fa[0][lane], fb0 = slot_in.fb[0][lane], fa1 = slot_in.fa[1][lane], fb1 = slot_in.fb[1][lane];
                    const int q0 = __builtin_amdgcn_readfirstlane(slot_in.q[0]), q1 = __builtin_amdgcn_readfirstlane(slot_in.q[1]);
                    E::spectrum(r, fa0, fb0, q0, xf);
                    E::spectrum(r, fa1, fb1, q1, df);
                } else {
                    const PipeSlot &slot_in = sh.slots[slot_idx][slot];
                    const int x = slot_in.near_x[lane], m = slot_in.mags[lane], sc = slot_in.scalars[lane];
                    xf.mag = zext16(m);
                    xf.mag64 = __builtin_amdgcn_readlane(sc, 0);
                    xf.q = __builtin_amdgcn_readlane(sc, 1);
                    xf.re = xf.im = 0;
                    xf.re64 = 0;
                    df.re = sext16(x);
                    df.im = sar(x, 16);
                    df.mag = lsr(m, 16);
                    df.re64 = __builtin_amdgcn_readlane(sc, 2);
                    df.mag64 = __builtin_amdgcn_readlane(sc, 3);
                    df.q = __builtin_amdgcn_readlane(sc, 4);
                }
                E::update_startup(r.u);
                int delay_given = 0, far_given = 0;
                if constexpr (kDelay != 0) {
                    delay_given = __builtin_amdgcn_readfirstlane(sh.delays[blk & 1][slot]);
                    far_given = sh.far_rows[blk & 1][slot][lane];
                }
                if constexpr (kTail != 0) {
                    if constexpr (kGain != 0) {
                        W::template phase_priority<3>(r.u.prio_drop);
                        E::track_q(r.u, df, df);
                        const typename E::GainInput g = E::template channel_block<true>(r, hist, xf, df, delay_given, far_given);
                        PipeGainSlot &gs = sh.gains[blk & 1][slot];
                        gs.echo_est[lane] = g.echo_est;
                        if (lane == 0) { gs.echo_est64 = g.echo_est64; gs.far_q = g.far_q; gs.cur_vad = g.cur_vad; gs.near0 = g.near0; gs.stored0 = g.stored0; }
                    } else {
                        const typename E::TailInput t = E::template middle_block<kDelay != 0>(r, hist, xf, df, df, delay_given, far_given);
                        PipeTailSlot &ts = sh.tails[blk & 1][slot];
                        ts.a[lane] = t.a;
                        ts.b[lane] = t.b;
                        if (lane == 0) ts.clean_q = t.clean_q;
                    }
                } else {
                    const typename E::TailInput t = E::template middle_block<kDelay != 0>(r, hist, xf, df, df, delay_given, far_given);
                    const int out = E::tail_block(r, t.a, t.b, t.clean_q);        // (= back_block)
                    sio.out(r, blk, out);
                }
            }
            slot_idx = slot_idx + 1 == kSlots ? 0 : slot_idx + 1;
            AECM_PIPE_BARRIER();                                          // slots[blk & 1] are free again, block blk + 1 is in the others
        }
        if (kGain != 0) AECM_PIPE_BARRIER();                              // the gain waves' last step
        if (kTail != 0) AECM_PIPE_BARRIER();                              // the tail waves' last step
        if constexpr (kGain != 0) {
            AECM_PIPE_BARRIER();                                          // the gain wave's part of the state is in gain_state
            if (live) {
                const int lane = W::lane_id();
                const PipeGainState &g = sh.gain_state[slot];
                const int nf = g.near_filt_ctrs[lane];
                r.b.echo_filt = g.echo_filt[lane];
                r.b.near_filt = sext16(nf); r.b.low_ctr = lsr(nf, 16) & 7; r.b.high_ctr = lsr(nf, 19) & 7;
                r.b.noise_est = g.noise_est[lane];
                Uniform &u = r.u;
                auto S = [&](int i) { return __builtin_amdgcn_readfirstlane(g.scal[i]); };
                u.seed = S(0); u.sup_gain = S(1); u.sup_gain_old = S(2); u.noise_ctr = S(3);
                r.b64.echo_filt = S(4); r.b64.near_filt = S(5); r.b64.noise_est = S(6); r.b64.low_ctr = S(7); r.b64.high_ctr = S(8);
            }
        }
        if (live) E::template store_state<false, kTail == 0, kDelay == 0>(r, vec, scal);
    } else if (wave < kPipeStreams + kPipeFrontWaves) {
        // ---- front wave: two streams, the transforms of the block after the one their back waves are at ----
        typename EF::Regs r;
        EF::init_lane_constants(r, st.consts);
        int level = kBalance ? kFrontBehind : (prio & 3);                 // this group's base priority (without balance: the launch's, PipeShape::prio)
        SetPrioDynamic(level);
        k0 = slot_of((wave - kPipeStreams) * kPipeStreamsPerFront, rot_front);        // (a wave of two slots: k0 and the one after it, round the ring)
        int x_old[kPipeStreamsPerFront], d_old[kPipeStreamsPerFront], far_next[kPipeStreamsPerFront], near_next[kPipeStreamsPerFront];
        bool live[kPipeStreamsPerFront];
        for (int k = 0; k < kPipeStreamsPerFront; ++k) {
            const int64_t stream = slot_stream(ks(k));
            live[k] = slot_live(ks(k));
            x_old[k] = d_old[k] = far_next[k] = near_next[k] = 0;
            if (live[k]) {
                EF::load_time_state(st.vec + stream * (int64_t)kVecWordsPerStream, r.lane, x_old[k], d_old[k]);
                typename EF::StridedIo sio{io, stream * io.stream_stride};
                far_next[k] = sio.far(r, 0);
                near_next[k] = sio.near(r, 0);
            }
        }
        int slot_idx = 0;                                                 // blk mod kSlots
        for (int blk = 0; blk <= n_blocks; ++blk) {                      // step blk writes block blk (the last step: nothing)
            // Balance: the monitor's step at a group boundary (see above).  pv[] is only ever read under the condition it is
            // loaded under (no initialisation: a register written by a move while a load of an earlier trip may still be
            // pending in the compiler's eyes costs a wait for everything in flight at the top of every trip).
            const bool boundary = kMode != 0 && (blk & kPipeGroupMask) == 0 && blk != 0;
            const bool monitor = boundary && wave == kPipeStreams;
            int pv[kPipeMonitorLoads];
            if (monitor) {
                // a workgroup's word is the 16-bit COMPLEMENT of its group count: the cleared buffer (and the unused half of an
                // odd last word) then reads as "as far ahead as can be", never as the slowest
                const int g = blk >> kPipeGroupLog2;
                __hip_atomic_store(reinterpret_cast<uint16_t *>(progress) + blockIdx.x, (uint16_t)(0xffff - (g < 0xffff ? g : 0xffff)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int lane = W::lane_id(), n_words = (n_workgroups + 1) >> 1;
#if AECM_PIPE_MONITOR_SAMPLE
                // a sample instead of everybody: 128 workgroups (one load), a different 128 at every boundary and in every workgroup.
                // The slow workgroups are a quarter to a half of the launch (the ones dispatched last to each CU), so every sample has some.
                const int n_chunks = (n_words + 63) >> 6, chunk = (int)((blockIdx.x + (unsigned)g * 7u) % (unsigned)n_chunks);
                const int w = lane + 64 * chunk;
                pv[0] = (int)__hip_atomic_load(progress + (w < n_words ? w : n_words - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
#pragma unroll
                for (int i = 0; i < kPipeMonitorLoads; ++i) {
                    const int w = lane + 64 * i;
                    if (64 * i < n_words)
                        pv[i] = (int)__hip_atomic_load(progress + (w < n_words ? w : n_words - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#endif
            }
            if (kMode != 0 && kFrontBehind != AECM_PIPE_FRONT_PRIO && (blk & kPipeGroupMask) == 1 && blk > kPipeGroupMask) {
                level = __builtin_amdgcn_readfirstlane(sh.level);
                if (kBoost == 0) SetPrioDynamic(level);
            }
            if (blk < n_blocks) {
#pragma unroll
                for (int k = 0; k < kPipeStreamsPerFront; ++k) {
                    if (!live[k]) continue;
                    if (kBoost != 0) SetPrioDynamic(level + (k == 0 ? 0 : kBoost));      // folds to immediates without balance
                    const int far_cur = far_next[k], near_cur = near_next[k];
                    if (blk + 1 < n_blocks) {
                        typename EF::StridedIo sio{io, slot_stream(ks(k)) * io.stream_stride};
                        far_next[k] = sio.far(r, blk + 1);
                        near_next[k] = sio.near(r, blk + 1);
                    }
                    r.table_index = Gfx950Wave<true, false>::table_index_for_this_block();
                    const int lane = W::lane_id();
                    if constexpr (kRaw) {
                        int fa[2], fb[2], q[2];
                        EF::front_transforms(r, x_old[k], far_cur, d_old[k], near_cur, fa, fb, q);
                        PipeRawSlot &slot = sh.slots[slot_idx][ks(k)];
                        slot.fa[0][lane] = fa[0]; slot.fb[0][lane] = fb[0];
                        slot.fa[1][lane] = fa[1]; slot.fb[1][lane] = fb[1];
                        if (lane == 0) { slot.q[0] = q[0]; slot.q[1] = q[1]; }
                    } else {
                        typename EF::Spectrum xf, df, cf;
                        EF::front_block(r, x_old[k], far_cur, d_old[k], near_cur, 0, 0, xf, df, cf);
                        PipeSlot &slot = sh.slots[slot_idx][ks(k)];
                        slot.near_x[lane] = (df.re & 0xffff) | (int)((unsigned)df.im << 16);
                        slot.mags[lane] = xf.mag | (int)((unsigned)df.mag << 16);
                        int sc = 0;
                        sc = W::writelane(sc, xf.mag64, 0);
                        sc = W::writelane(sc, xf.q, 1);
                        sc = W::writelane(sc, df.re64, 2);
                        sc = W::writelane(sc, df.mag64, 3);
                        sc = W::writelane(sc, df.q, 4);
                        slot.scalars[lane] = sc;
                    }
                    x_old[k] = far_cur;
                    d_old[k] = near_cur;
                }
            }
            if (monitor) {
                const int n_words = (n_workgroups + 1) >> 1;
                int m = AECM_PIPE_BALANCE_RULE == 0 ? pv[0] : pk_add_u16(pv[0], -1);
#if !AECM_PIPE_MONITOR_SAMPLE
#pragma unroll
                for (int i = 1; i < kPipeMonitorLoads; ++i)
                    if (64 * i < n_words) m = AECM_PIPE_BALANCE_RULE == 0 ? pk_max_u16(m, pv[i]) : pk_min_u16(m, pk_add_u16(pv[i], -1));
#endif
#if AECM_PIPE_BALANCE_RULE == 0
                const int slowest = 0xffff - W::reduce_max(imax(zext16(m), lsr(m, 16)));      // group count of the slowest workgroup that has published
                const int lead = (blk >> kPipeGroupLog2) - slowest;
#if defined(AECM_PIPE_BALANCE_DRY)       // A/B: the monitor runs, nobody is ever demoted (what the mechanism itself costs)
                sh.ahead = lead > 0x7ffffff ? 1 : 0;
#else
                sh.ahead = lead > AECM_PIPE_BALANCE_LEAD ? 1 : 0;
#endif
#else
                // the fastest: the smallest published word (an unpublished or unused half reads 0: minus one it is the largest)
                const int fastest = 0xfffe - W::reduce_min(imin(zext16(m), lsr(m, 16)));
                const int lead = AECM_PIPE_BALANCE_LEAD + 1 - (fastest - (blk >> kPipeGroupLog2));     // (> LEAD <=> less than one group behind the fastest)
                sh.ahead = lead > AECM_PIPE_BALANCE_LEAD ? 1 : 0;
#endif
                if (kMode == 3)      // proportional: one priority level per group of lead beyond the allowed one, from .._BEHIND down to AECM_PIPE_FRONT_PRIO
                    sh.level = imax(kFrontBehind - imax(lead - AECM_PIPE_BALANCE_LEAD, 0), AECM_PIPE_FRONT_PRIO);
                else
                    sh.level = sh.ahead ? AECM_PIPE_FRONT_PRIO : kFrontBehind;
            }
            slot_idx = slot_idx + 1 == kSlots ? 0 : slot_idx + 1;
            AECM_PIPE_BARRIER();
        }
        if (kDelay != 0) AECM_PIPE_BARRIER();                             // the middle waves' last step
        if (kGain != 0) AECM_PIPE_BARRIER();                              // the gain waves' last step
        if (kTail != 0) AECM_PIPE_BARRIER();                              // the tail waves' last step
        if (kGain != 0) AECM_PIPE_BARRIER();                              // (state hand-over of the gain waves)
        for (int k = 0; k < kPipeStreamsPerFront; ++k)
            if (live[k]) EF::store_time_state(st.vec + slot_stream(ks(k)) * (int64_t)kVecWordsPerStream, r.lane, x_old[k], d_old[k]);
    } else if (wave < kPipeStreams + kPipeFrontWaves + kTail) {
        // ---- tail wave: kPipeStreams / kTail streams, inverse transform + synthesis + output of the block BEFORE the one the middle waves are at ----
        constexpr int kPer = kTail ? kPipeStreams / kTail : 1;
        typename EF::Regs r;
        EF::init_lane_constants(r, st.consts);
        r.u.prio_drop = 0;
        SetPrioDynamic((prio >> 2) & 3);
        k0 = slot_of((wave - kPipeStreams - kPipeFrontWaves) * kPer, rot_tail);
        int ovl[kPer], c_old[kPer];
        bool live[kPer];
        for (int k = 0; k < kPer; ++k) {
            const int64_t stream = slot_stream(ks(k));
            live[k] = slot_live(ks(k));
            ovl[k] = c_old[k] = 0;
            if (live[k]) EF::load_tail_state(st.vec + stream * (int64_t)kVecWordsPerStream, r.lane, ovl[k], c_old[k]);
        }
        AECM_PIPE_BARRIER();                                              // steps 0 and 1 (and 2 with delay waves): nothing to do yet
        AECM_PIPE_BARRIER();
        if (kDelay != 0) AECM_PIPE_BARRIER();
        if (kGain != 0) AECM_PIPE_BARRIER();
        for (int blk = 0; blk < n_blocks; ++blk) {                        // step blk + 2 (+ 1 with delay waves, + 1 with gain waves)
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                if (!live[k]) continue;
                const PipeTailSlot &ts = sh.tails[blk & 1][ks(k)];
                const int lane = W::lane_id();
                const int a = ts.a[lane], b = ts.b[lane];
                const int clean_q = __builtin_amdgcn_readfirstlane(ts.clean_q);
                r.table_index = Gfx950Wave<true, false>::table_index_for_this_block();
                r.out_ovl = ovl[k];
                const int out = EF::tail_block(r, a, b, clean_q);
                ovl[k] = r.out_ovl;
                typename EF::StridedIo sio{io, slot_stream(ks(k)) * io.stream_stride};
                sio.out(r, blk, out);
            }
            AECM_PIPE_BARRIER();
        }
        if (kGain != 0) AECM_PIPE_BARRIER();                              // (state hand-over of the gain waves)
        for (int k = 0; k < kPer; ++k)
            if (live[k]) EF::store_tail_state(st.vec + slot_stream(ks(k)) * (int64_t)kVecWordsPerStream, r.lane, ovl[k], c_old[k]);
    } else if constexpr (kDelay != 0) {
        if (wave < kPipeStreams + kPipeFrontWaves + kTail + kDelay) {
            // ---- delay wave: kPipeStreams / kDelay streams, the delay estimator of the block AFTER the one their channel waves are at ----
            constexpr int kPer = kPipeStreams / kDelay;
            typename EF::Regs r;
            EF::init_lane_constants(r, st.consts);
            r.u.prio_drop = 0;
            SetPrioDynamic((prio >> 4) & 3);
            k0 = slot_of((wave - (kPipeStreams + kPipeFrontWaves + kTail)) * kPer, rot_delay);
            // the estimator's state per stream (BlockEngine::load_delay_state's fields), moved into r around each call
            int mean[kPer], bh0[kPer], bh1[kPer], m01[kPer], far_init[kPer], near_init[kPer], min_prob[kPer], last_prob[kPer], last_delay[kPer];
            int hist_pos[kPer], fixed_delay[kPer];                        // the channel wave's u.hist_pos, followed here
            bool live[kPer];
            auto swap_in = [&](int k) {
                r.mean = mean[k]; r.bh0 = bh0[k]; r.bh1 = bh1[k]; r.m01 = m01[k];
                r.u.far_init = far_init[k]; r.u.near_init = near_init[k]; r.u.min_prob = min_prob[k]; r.u.last_prob = last_prob[k];
                r.u.last_delay = last_delay[k]; r.u.fixed_delay = fixed_delay[k];
            };
            auto swap_out = [&](int k) {
                mean[k] = r.mean; bh0[k] = r.bh0; bh1[k] = r.bh1; m01[k] = r.m01;
                far_init[k] = r.u.far_init; near_init[k] = r.u.near_init; min_prob[k] = r.u.min_prob; last_prob[k] = r.u.last_prob;
                last_delay[k] = r.u.last_delay;
            };
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int64_t stream = slot_stream(ks(k));
                live[k] = slot_live(ks(k));
                hist_pos[k] = 0; fixed_delay[k] = -1;
                r.mean = r.bh0 = r.bh1 = r.m01 = 0;
                r.u.far_init = r.u.near_init = r.u.min_prob = r.u.last_prob = r.u.last_delay = 0;
                if (live[k]) {
                    const int32_t *scal = st.scal + stream * (int64_t)kNumScal;
                    EF::load_delay_state(r, st.vec + stream * (int64_t)kVecWordsPerStream, scal);
                    hist_pos[k] = __builtin_amdgcn_readfirstlane(scal[S_HISTPOS]);
                    fixed_delay[k] = __builtin_amdgcn_readfirstlane(scal[S_FIXED_DELAY]);
                }
                swap_out(k);
            }
            AECM_PIPE_BARRIER();                                          // step 0: nothing to do yet
            int slot_idx = 0, slot_before = kSlots - 1;
            for (int blk = 0; blk < n_blocks; ++blk) {                    // step blk + 1
                const int lane = W::lane_id();
                int far[kPer];
                bool fetch[kPer];
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    fetch[k] = false;
                    if (!live[k]) continue;
                    const PipeSlot &slot = sh.slots[slot_idx][ks(k)];
                    const int m = slot.mags[lane], sc = slot.scalars[lane];
                    typename EF::Spectrum xf, df;
                    xf.mag = zext16(m);
                    xf.q = __builtin_amdgcn_readlane(sc, 1);
                    df.mag = lsr(m, 16);
                    df.q = __builtin_amdgcn_readlane(sc, 4);
                    r.table_index = Gfx950Wave<true, false>::table_index_for_this_block();
                    swap_in(k);
                    const int estimate = EF::delay_block(r, xf, df);
                    swap_out(k);
                    if (lane == 0) sh.delays[blk & 1][ks(k)] = estimate;
                    // AlignedFarend for the channel wave: the history row it would fetch next step.  The row of the block before this
                    // one is being written in this very step -- but that block's spectrum is still in its slot; older rows are in
                    // memory (written at least one barrier ago, or by an earlier launch); a delay of 0 is the block's own spectrum,
                    // which the channel wave has.
                    hist_pos[k] = hist_pos[k] + 1 >= kHistory ? 0 : hist_pos[k] + 1;
                    const int delay = EF::effective_delay(r.u, estimate);
                    fetch[k] = delay != 0;
                    if (delay != 0) {
                        const uint16_t *hist = st.hist + slot_stream(ks(k)) * (int64_t)kHistWordsPerStream;
                        if (delay == 1 && blk > 0) far[k] = zext16(sh.slots[slot_before][ks(k)].mags[lane]);
                        else far[k] = Gfx950Wave<true, false>::load_u16(hist + EF::aligned_slot(hist_pos[k], delay) * kLanes, lane);
                    }
                }
#pragma unroll
                for (int k = 0; k < kPer; ++k)                            // (the stores after every stream's fetch is under way)
                    if (fetch[k]) sh.far_rows[blk & 1][ks(k)][lane] = far[k];
                slot_before = slot_idx;
                slot_idx = slot_idx + 1 == kSlots ? 0 : slot_idx + 1;
                AECM_PIPE_BARRIER();
            }
            AECM_PIPE_BARRIER();                                          // the channel waves' last step
            if (kGain != 0) AECM_PIPE_BARRIER();                          // the gain waves' last step
            if (kTail != 0) AECM_PIPE_BARRIER();                          // the tail waves' last step
            if (kGain != 0) AECM_PIPE_BARRIER();                          // (state hand-over of the gain waves)
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                if (!live[k]) continue;
                swap_in(k);
                EF::store_delay_state(r, st.vec + slot_stream(ks(k)) * (int64_t)kVecWordsPerStream, st.scal + slot_stream(ks(k)) * (int64_t)kNumScal);
            }
        } else if constexpr (kGain != 0) {
            // ---- gain wave: one stream, gain_block of the block BEFORE the one its channel wave is at ----
            typename EF::Regs r;
            EF::init_lane_constants(r, st.consts);
            r.u.prio_drop = 0;
            SetPrioDynamic((prio >> 6) & 3);
            const int k = slot_of(wave - (kPipeStreams + kPipeFrontWaves + kTail + kDelay), rot_gain);
            const int64_t stream = slot_stream(k);
            const bool live = slot_live(k);
            if (live) EF::load_state(r, st.vec + stream * (int64_t)kVecWordsPerStream, st.scal + stream * (int64_t)kNumScal);
            AECM_PIPE_BARRIER();                                          // steps 0, 1, 2: nothing to do yet
            AECM_PIPE_BARRIER();
            AECM_PIPE_BARRIER();
            int slot_idx = 0;
            for (int blk = 0; blk < n_blocks; ++blk) {                    // step blk + 3
                if (live) {
                    const int lane = W::lane_id();
                    const PipeSlot &slot = sh.slots[slot_idx][k];
                    const int x = slot.near_x[lane], m = slot.mags[lane], sc = slot.scalars[lane];
                    typename EF::Spectrum df;
                    df.re = sext16(x);
                    df.im = sar(x, 16);
                    df.mag = lsr(m, 16);
                    df.re64 = __builtin_amdgcn_readlane(sc, 2);
                    df.mag64 = __builtin_amdgcn_readlane(sc, 3);
                    df.q = __builtin_amdgcn_readlane(sc, 4);
                    const PipeGainSlot &gs = sh.gains[blk & 1][k];
                    typename EF::GainInput g;
                    g.echo_est = gs.echo_est[lane];
                    g.echo_est64 = __builtin_amdgcn_readfirstlane(gs.echo_est64);
                    g.far_q = __builtin_amdgcn_readfirstlane(gs.far_q);
                    g.cur_vad = __builtin_amdgcn_readfirstlane(gs.cur_vad);
                    g.near0 = __builtin_amdgcn_readfirstlane(gs.near0);
                    g.stored0 = __builtin_amdgcn_readfirstlane(gs.stored0);
                    r.table_index = Gfx950Wave<true, false>::table_index_for_this_block();
                    EF::track_q(r.u, df, df);
                    const typename EF::TailInput t = EF::gain_block(r, df, df, g);
                    PipeTailSlot &ts = sh.tails[blk & 1][k];
                    ts.a[lane] = t.a;
                    ts.b[lane] = t.b;
                    if (lane == 0) ts.clean_q = t.clean_q;
                }
                slot_idx = slot_idx + 1 == kSlots ? 0 : slot_idx + 1;
                AECM_PIPE_BARRIER();
            }
            if (live) {                                                   // this wave's part of the state -> the channel wave (which stores the state)
                const int lane = W::lane_id();
                PipeGainState &g = sh.gain_state[k];
                g.echo_filt[lane] = r.b.echo_filt;
                g.near_filt_ctrs[lane] = zext16(r.b.near_filt) | shl(r.b.low_ctr & 7, 16) | shl(r.b.high_ctr & 7, 19);
                g.noise_est[lane] = r.b.noise_est;
                if (lane == 0) {
                    const Uniform &u = r.u;
                    g.scal[0] = u.seed; g.scal[1] = u.sup_gain; g.scal[2] = u.sup_gain_old; g.scal[3] = u.noise_ctr;
                    g.scal[4] = r.b64.echo_filt; g.scal[5] = r.b64.near_filt; g.scal[6] = r.b64.noise_est; g.scal[7] = r.b64.low_ctr; g.scal[8] = r.b64.high_ctr;
                }
            }
            if (kTail != 0) AECM_PIPE_BARRIER();                          // the tail waves' last step
            AECM_PIPE_BARRIER();                                          // (state hand-over)
        }
    }
#if defined(AECM_PIPE_TRACE)
    if ((threadIdx.x & 63u) == 0) {
        uint64_t *tr = reinterpret_cast<uint64_t *>(progress + 2 * ((n_workgroups + 3) / 4) * 2) + ((size_t)blockIdx.x * kPipeTraceWaves + wave) * 4;
        // where the wave ran: HW_ID (wave slot 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13) above bit 40 of the wait count, XCC_ID above bit 40 of the total
        const uint64_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        tr[0] = trace_t0; tr[1] = wall_clock64(); tr[2] = (trace_wait & ((1ull << 40) - 1)) | ((hw_id & 0xffffff) << 40);
        tr[3] = ((clock64() - trace_c0) & ((1ull << 40) - 1)) | ((xcc_id & 0xf) << 40);
    }
#endif
}
#undef AECM_PIPE_BARRIER

// Streams a pipelined launch keeps resident at once: workgroups per CU by wave slots (4 SIMDs x 7) and by LDS (160 KB).
template <int kTail, int kFront = 2, int kDelay = 0, int kGain = 0>
constexpr int PipeWorkgroupsPerCu() {
    constexpr int by_waves = 4 * PipeWavesPerEu(kTail, kFront, kDelay, kGain) / PipeWaves(kTail, kFront, kDelay, kGain);
    constexpr int by_lds = (int)((160 * 1024) / (sizeof(LdsTables) + (kDelay ? sizeof(PipeShared<kTail, false, kDelay, kGain>) : sizeof(PipeShared<kTail, true>))));
    return by_waves < by_lds ? by_waves : by_lds;
}
// Streams a pipelined launch of this shape keeps resident at once: workgroups per CU by wave slots (4 SIMDs x 7) and by LDS (160 KB).
int PipelinedStreamLimit(int compute_units, int tail_waves, int front_waves, int delay_waves, int gain_waves, int wgs_per_cu) {
    const int per_cu = wgs_per_cu > 0 ? wgs_per_cu
                       : tail_waves == 0 ? PipeWorkgroupsPerCu<0>()
                       : gain_waves != 0 ? AECM_PIPE_GAIN_WGS_PER_CU
                       : delay_waves != 0 ? (front_waves == 4 ? PipeWorkgroupsPerCu<2, 4, 4>() : PipeWorkgroupsPerCu<2, 2, 4>())
                       : front_waves == 4 ? PipeWorkgroupsPerCu<2, 4>() : PipeWorkgroupsPerCu<2>();
    return (compute_units > 0 ? compute_units : 256) * per_cu * kPipeStreams;
}

// The shape of a pipelined launch by its size (measured: profiles/r05_experiments.md section 1.4; M frames/s, 2 048 blocks):
//   up to one workgroup per CU    (1 024 streams on 256 CUs)  two tail waves                                  372 -> 421
//   up to two per CU              (2 048)                     two tail waves, FOUR front waves, raw hand-over  600 -> 740
//   up to three per CU            (3 072)                     two tail waves, raw hand-over                    735 -> 795
//   more (four per CU: 4 096)                                 balance + raw hand-over (launches of >= 128 blocks)  816 -> 863
// tail_waves / front_waves / raw < 0: by this table; otherwise the caller's wish where the shape exists and fits (experiments).
PipeShape PipelinedShapeFor(int n_streams, int n_blocks, int compute_units, const PipeWishes &wishes) {
    const int tail_waves = wishes.tail_waves, front_waves = wishes.front_waves, raw = wishes.raw, delay_waves = wishes.delay_waves, gain_waves = wishes.gain_waves;
    const int spread = wishes.spread, wgs = wishes.wgs_per_cu;
    const int cus = compute_units > 0 ? compute_units : 256;
    const int n_wg = (n_streams + kPipeStreams - 1) / kPipeStreams;
    PipeShape sh{0, 2, false, false, 0, 0, cus, 0, 0, n_wg};
    const int want_tail = tail_waves < 0 ? 2 : tail_waves;
    if (want_tail >= 2 && n_streams <= PipelinedStreamLimit(cus, 2, 2, 0, 0, wgs)) sh.tail_waves = 2;
    const int want_front = front_waves < 0 ? (n_wg > cus ? 4 : 2) : front_waves;
    if (want_front >= 4 && sh.tail_waves == 2 && n_streams <= PipelinedStreamLimit(cus, 2, 4, 0, 0, wgs)) sh.front_waves = 4;
    sh.balance = sh.tail_waves == 0 && AECM_PIPE_BALANCE != 0 && n_blocks >= (8 << kPipeGroupLog2) && n_wg <= 128 * kPipeMonitorLoads && n_wg > 3 * cus;
    const bool want_raw = (raw < 0 ? (AECM_PIPE_RAW_HANDOVER != 0 && (sh.balance || (sh.tail_waves == 2 && n_wg > cus))) : raw != 0);
    // the raw form exists for: the balanced shape, and the two-tail shapes
    sh.raw = want_raw && (sh.balance || sh.tail_waves == 2);
    if (sh.balance && !want_raw) sh.balance = false;          // (no balanced instantiation without the raw hand-over)
    // delay waves: the two-tail shapes with formed spectra
    const int want_delay = delay_waves < 0 ? AECM_PIPE_DELAY_DEFAULT(n_wg, cus) : delay_waves;
    // (Delay waves where the CU is short of issue slots rather than of independent work -- two delay waves next to four front waves
    // at two workgroups per CU, one for the workgroup's four streams at three and four per CU -- measured slower than the shapes
    // above: 2 048 streams 712 vs 740 M frames/s, 3 072 555 vs 796, 4 096 796 vs 859.  Not instantiated.)
    if (want_delay != 0 && sh.tail_waves == 2 && (raw < 0 || raw == 0) && n_streams <= PipelinedStreamLimit(cus, 2, 2, kPipeStreams, 0, wgs)) {
        sh.front_waves = 2;
        sh.delay_waves = kPipeStreams;
        sh.raw = false;
        // gain waves: the sixteen-wave shape (four front waves, two delay waves)
        const int want_gain = gain_waves < 0 ? AECM_PIPE_GAIN_DEFAULT(n_wg, cus) : gain_waves;
        if (want_gain != 0 && (front_waves < 0 || front_waves == 4) && n_streams <= PipelinedStreamLimit(cus, 2, 4, 2, kPipeStreams, wgs)) {
            sh.gain_waves = kPipeStreams;
            sh.front_waves = 4;
            sh.delay_waves = 2;
        }
    }
    // Onto the set of kernels the library carries (LaunchProcessBlocksPipelined): four front waves exist with the raw hand-over (or with
    // delay and gain waves) only -- a wish for them without it gets the raw form rather than a launch error.
    if (sh.front_waves == 4 && sh.delay_waves == 0 && !sh.raw) sh.raw = true;
    // Even load (round 6).  The shape holds per_cu workgroups on a CU; with fewer workgroups of four streams than that on some CUs
    // the launch ends when the fullest CU does (1 536 streams = 384 workgroups: half the CUs carried two, the launch ran slower than
    // 1 024 streams).  Every CU gets its full count of workgroups instead, of three or four (two, one) streams each: the
    // dispatcher deals workgroups out to the CUs in turn, the first n_streams % workgroups of them serve one stream more
    // (the kernel's first lines), so the CUs' loads differ by at most one stream.
    if (spread != 0) {
        const int per_cu = PipelinedStreamLimit(cus, sh.tail_waves, sh.front_waves, sh.delay_waves, sh.gain_waves, wgs) / (cus * kPipeStreams);
        int full = per_cu * cus < n_streams ? per_cu * cus : n_streams;
        if (sh.balance && full > 128 * kPipeMonitorLoads) full = 128 * kPipeMonitorLoads;      // (what the monitor wave reads; n_wg fits: see balance above)
        if (full > sh.workgroups) sh.workgroups = full;
    }
    // The slot rotations (the kernel's slot_of; measured over all sixteen front x gain and front x tail combinations, M frames/s):
    //   sixteen waves, workgroups of one stream (two of them)   front + 1, gain + 2:  256 streams 186 -> 227, 512: 365 -> 441, 768: 507 -> 550
    //   sixteen waves, workgroups of two to four streams        front + 3, gain + 1:  1 024: 626 -> 673, 1 280: 537 -> 650, 1 536: 574 -> 688, 1 792: 641 -> 702
    //   eight waves (front and tail waves of two slots each)    front + 3, tail + 2:  2 304: 634 -> 679, 2 560: 706 -> 709, 2 816: 734 -> 753
    //   ten and six waves                                       none (nothing moved by more than the run-to-run spread)
    if (sh.gain_waves != 0) sh.rot = n_streams < 2 * sh.workgroups ? (1 | (2 << 2)) : (3 | (1 << 2));
    else if (sh.tail_waves == 2 && sh.front_waves == 2 && sh.delay_waves == 0) sh.rot = 3 | (2 << 8);
    else sh.rot = 0;
    if (wishes.rot >= 0) sh.rot = wishes.rot;
    // The roles' issue priorities (the channel / middle / back waves': by phase of the block, 1..3): front | tail << 2 | delay << 4 | gain << 6.
    // Swept like the rotations (profiles/r06_experiments.md section 1.5; M frames/s):
    //   sixteen waves, two to four streams per workgroup: the delay waves at 0 instead of 1    1 280 streams 648 -> 682, 1 536: 682 -> 684, 1 792: 700 -> 708, 2 048: 727 -> 721
    //     (workgroups of one stream keep 1: 512 streams 443 vs 432)
    //   eight waves, workgroups not all full: the tail waves at 0 instead of 1                 2 304 streams 676 -> 685, 2 560: 705 -> 732, 2 816: 751 -> 783 (3 072, all full: 787 at 1, 760 at 0)
    int tail_prio = AECM_PIPE_TAIL_PRIO, delay_prio = AECM_PIPE_DELAY_PRIO;
    if (sh.gain_waves != 0 && n_streams >= 2 * sh.workgroups) delay_prio = 0;
    if (sh.tail_waves == 2 && sh.front_waves == 2 && sh.delay_waves == 0 && n_streams < kPipeStreams * sh.workgroups) tail_prio = 0;
    sh.prio = AECM_PIPE_FRONT_PRIO | (tail_prio << 2) | (delay_prio << 4) | (AECM_PIPE_GAIN_PRIO << 6);
    if (wishes.prio >= 0) sh.prio = wishes.prio;
    return sh;
}

int PipelinedWorkgroupWaves(const PipeShape &shape) { return PipeWaves(shape.tail_waves, shape.front_waves, shape.delay_waves, shape.gain_waves); }
int PipelinedWorkgroupsPerCu(const PipeShape &shape) {
    return PipelinedStreamLimit(1, shape.tail_waves, shape.front_waves, shape.delay_waves, shape.gain_waves) / kPipeStreams;
}

hipError_t LaunchProcessBlocksPipelined(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, const PipeShape &shape, uint32_t *progress,
                                        hipStream_t stream) {
    if (n_streams <= 0 || n_blocks <= 0) return hipSuccess;
    const int n_wg = shape.workgroups;
    if (n_wg <= 0 || (int64_t)n_wg * kPipeStreams < n_streams || n_wg > n_streams) return hipErrorInvalidValue;
    const dim3 grid(n_wg), block(64 * PipeWaves(shape.tail_waves, shape.front_waves, shape.delay_waves, shape.gain_waves));
    const int streams_base = n_streams / n_wg, streams_rem = n_streams % n_wg;
    if (shape.balance) {
        if (!progress || n_wg > 128 * kPipeMonitorLoads) return hipErrorInvalidValue;
        const hipError_t e = hipMemsetAsync(progress, 0, PipelinedControlBytes(n_wg), stream);
        if (e != hipSuccess) return e;
    }
#if !defined(AECM_PIPE_TRACE)
    if (!shape.balance) progress = nullptr;
#endif
#define AECM_LAUNCH_PIPE(T, B, R, F, D, G) hipLaunchKernelGGL((aecm_process_pipelined_kernel<T, B, R, F, D, G>), grid, block, sizeof(LdsTables) + sizeof(PipeShared<T, R, D, G>), \
                                                              stream, st, io, streams_base, streams_rem, n_blocks, progress, (int)grid.x, shape.wgs_per_round, shape.rot, shape.prio)
    // The instantiations the library carries (PipelinedShapeFor only ever asks for these).  One tail wave for four streams (kTail = 1,
    // seven-wave workgroups) measured slower than its neighbours at every size and is not built.
    const int key = shape.gain_waves * 10000 + shape.delay_waves * 1000 + shape.tail_waves * 100 + shape.front_waves * 10 + (shape.raw ? 1 : 0);
    if (shape.balance) { if (key != 21) return hipErrorInvalidValue; AECM_LAUNCH_PIPE(0, true, true, 2, 0, 0); }
    else if (key == 20) AECM_LAUNCH_PIPE(0, false, false, 2, 0, 0);
    else if (key == 220) AECM_LAUNCH_PIPE(2, false, false, 2, 0, 0);
    else if (key == 221) AECM_LAUNCH_PIPE(2, false, true, 2, 0, 0);
    else if (key == 241) AECM_LAUNCH_PIPE(2, false, true, 4, 0, 0);
    else if (key == 4220) AECM_LAUNCH_PIPE(2, false, false, 2, 4, 0);
    else if (key == 42240) AECM_LAUNCH_PIPE(2, false, false, 4, 2, 4);
    else return hipErrorInvalidValue;
#undef AECM_LAUNCH_PIPE
    return hipGetLastError();
}

// The progress words of a pipelined launch: 16 bits per workgroup (cleared by the launch).
size_t PipelinedControlBytes(int n_workgroups) {
    const size_t n_wg = (size_t)n_workgroups;
#if defined(AECM_PIPE_TRACE)
    return 4 * ((n_wg + 3) / 4) * sizeof(uint32_t) + n_wg * kPipeTraceWaves * 4 * sizeof(uint64_t);      // progress halves (padded), then the trace records
#else
    return (n_wg + 2) / 2 * sizeof(uint32_t);
#endif
}
size_t PipelinedTraceOffsetBytes(int n_workgroups) {
    const size_t n_wg = (size_t)n_workgroups;
    return 4 * ((n_wg + 3) / 4) * sizeof(uint32_t);
}

size_t QueueControlBytes(int n_streams) { return ((size_t)kQueueCtlWords + (size_t)n_streams) * sizeof(uint32_t); }

// Whether a launch of this shape takes the chunk-queue kernel (chunk_blocks > 0: the engine's setting).
bool QueueLaunchApplies(int n_streams, int n_blocks, int variant, int chunk_blocks, int min_streams, bool ragged) {
    if (chunk_blocks <= 0 || variant != kVariantFast || ragged) return false;
    if (n_streams <= min_streams || n_blocks < 2 * chunk_blocks) return false;
    const int64_t items = (int64_t)n_streams * ((n_blocks + chunk_blocks - 1) / chunk_blocks);
    return items < (int64_t(1) << 31);
}

hipError_t LaunchProcessBlocksQueued(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, int chunk_blocks,
                                     int resident_waves, uint32_t *ctl, uint32_t *err, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(ctl, 0, QueueControlBytes(n_streams), stream);
    if (e != hipSuccess) return e;
    const int n_chunks = (n_blocks + chunk_blocks - 1) / chunk_blocks;
    const int resident_groups = resident_waves / kWavesPerWorkgroup;
    const int needed = (n_streams + kWavesPerWorkgroup - 1) / kWavesPerWorkgroup;
    const dim3 grid(needed < resident_groups ? needed : resident_groups);
    const dim3 block(64 * kWavesPerWorkgroup);
    const size_t lds = sizeof(LdsTables);
    if (io.near_clean != nullptr)
        hipLaunchKernelGGL((aecm_process_queue_kernel<true>), grid, block, lds, stream, st, io, n_streams, n_blocks, chunk_blocks, n_chunks, ctl, err);
    else
        hipLaunchKernelGGL((aecm_process_queue_kernel<false>), grid, block, lds, stream, st, io, n_streams, n_blocks, chunk_blocks, n_chunks, ctl, err);
    return hipGetLastError();
}

int ResidentWaves(int compute_units) { return (compute_units > 0 ? compute_units : 256) * 4 * AECM_WAVES_PER_EU; }

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
