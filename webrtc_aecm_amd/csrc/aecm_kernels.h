// Launchers of the gfx950 kernels (aecm_kernels.hip).  Host-callable, HIP runtime types only.
#ifndef AECM_AMD_KERNELS_H_
#define AECM_AMD_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "aecm_flow_plan.h"
#include "aecm_state.h"

namespace aecm {

#ifndef AECM_WAVES_PER_WORKGROUP
#define AECM_WAVES_PER_WORKGROUP 4
#endif
constexpr int kWavesPerWorkgroup = AECM_WAVES_PER_WORKGROUP;    // 256 threads: 4 streams share one copy of the LDS tables

enum KernelVariant : int {
    kVariantSafe = 0,   // ds_bpermute shuffles only
    kVariantFast = 1    // DPP / permlane-swap cross-lane primitives (default)
};

// Process n_blocks consecutive blocks of streams [0, n_streams) -- one wavefront per stream.
// blocks_per_stream (device, may be null): stream s processes blocks_per_stream[s] <= n_blocks blocks
// instead (streaming sessions whose flow classes are out of phase).
hipError_t LaunchProcessBlocks(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, int variant,
                               hipStream_t stream, const int32_t *blocks_per_stream = nullptr);

// Replicate one stream image (vec: kNumVec*64 words, scal: 64 words, both on the device) into
// streams [first, first + count) and clear their far-spectrum history.
hipError_t LaunchBroadcastImage(const StatePtrs &st, const uint32_t *image_vec, const int32_t *image_scal,
                                int first, int count, hipStream_t stream);

// Overwrite a few scalar fields (field ids from ScalField) of streams [first, first + count).
hipError_t LaunchPatchScalars(const StatePtrs &st, const int32_t *fields_dev, const int32_t *values_dev, int n_fields,
                              int first, int count, hipStream_t stream);

// Session-schedule gather / scatter (aecm_session_flow.h: RecordingSchedule), all streams at once.
//   dst[s][j] = map[j] >= 0 ? src[s*src_stride + map[j]] : 0            j in [0, n)
hipError_t LaunchGatherByMap(const int16_t *src, int64_t src_stride, const int32_t *map_dev, int64_t n, int16_t *dst,
                             int64_t dst_stride, int n_streams, hipStream_t stream);
//   out[s][j] = v >= 0 ? blocks[s][v] : (v == -1 ? 0 : near[s][-(v+2)])  with v = map[j]
hipError_t LaunchAssembleOutput(const int16_t *blocks, int64_t blocks_stride, const int16_t *near, int64_t near_stride,
                                const int32_t *map_dev, int64_t n, int16_t *out, int64_t out_stride, int n_streams,
                                hipStream_t stream);

// Streaming sessions (aecm_sessions.cpp): per-stream sample rings of `ring_len` (power of two) elements.
// A tick is three launches: prepare -> LaunchProcessBlocks -> finish.  Where each sample comes from is
// decided on the host (SessionFlow in the index domain) and travels as kernel arguments, one int32
// source code per sample: -1 = zero, else (kind << 28) | index.
constexpr int kTickMaxBlockSamples = 256;   // <= 4 blocks per tick
constexpr int kTickMaxSamples = 160;
constexpr int kTickFrame = 80;              // FRAME_LEN: output frames are assembled 80 samples at a time
enum TickSource : int32_t {
    kTickFromInput = 0,      // gather: this tick's far/near input row          assemble: this tick's block outputs
    kTickFromRing = 1,       // gather: the far/near ring                       assemble: the output ring
    kTickNearInput = 2,      //                                                 assemble: this tick's near input (pass-through)
    kTickNearRing = 3        //                                                 assemble: the near ring (pass-through)
};
struct TickGatherCodes { int32_t far[kTickMaxBlockSamples], near[kTickMaxBlockSamples]; };
struct TickAssembleCodes { int32_t out[kTickMaxSamples]; };
// prepare: append the tick's first n_far far and all n near(/clean) samples to the rings (at far_pos /
// near_pos) and gather the tick's nb blocks: bfar/bnear(/bclean)[s][j] for j in [0, nb*64).  The far ring
// only takes what the session's jitter buffer accepted (a saturated one drops the rest), so a far tag is
// the count of ACCEPTED samples and never outlives the ring.  clean_in == nullptr: no clean near-end
// (clean_ring / bclean unused); the clean samples follow the near codes and positions.
hipError_t LaunchTickPrepare(const int16_t *far_in, const int16_t *near_in, const int16_t *clean_in, int64_t in_stride, int n,
                             int n_far, int16_t *far_ring, int16_t *near_ring, int16_t *clean_ring, int64_t ring_len, int64_t far_pos,
                             int64_t near_pos, int16_t *bfar, int16_t *bnear, int16_t *bclean, int n_block_samples,
                             const TickGatherCodes &codes, int n_streams, hipStream_t stream);
// finish: append the nb*64 block outputs to the output ring (at out_pos) and assemble the tick's n
// output samples.
hipError_t LaunchTickFinish(const int16_t *bout, int n_block_samples, int16_t *out_ring, const int16_t *near_ring,
                            int64_t ring_len, int64_t out_pos, const int16_t *near_in, int64_t io_stride, int16_t *out,
                            int n, const TickAssembleCodes &codes, int n_streams, hipStream_t stream);

// Sessions whose msInSndCardBuf histories differ live in different flow classes (aecm_sessions.h): the
// per-sample source codes then come from a device table indexed by the stream's class, and the block
// buffers use a fixed row stride of kTickMaxBlockSamples.
struct TickClassEntry {
    int32_t n_block_samples;     // blocks of this tick * 64
    int32_t n_far;               // how many of the tick's far samples the jitter buffer accepted (the first n_far)
    int64_t far_pos;             // where they go in the far ring
    int64_t out_pos;             // where this class's block outputs go in the output ring
    TickGatherCodes gather;
    TickAssembleCodes assemble;
};
hipError_t LaunchTickPrepareClasses(const int16_t *far_in, const int16_t *near_in, const int16_t *clean_in, int64_t in_stride,
                                    int n, int16_t *far_ring, int16_t *near_ring, int16_t *clean_ring, int64_t ring_len,
                                    int64_t near_pos, int16_t *bfar, int16_t *bnear, int16_t *bclean,
                                    const int32_t *class_of_stream, const TickClassEntry *table, int32_t *blocks_per_stream,
                                    int n_streams, hipStream_t stream);
hipError_t LaunchTickFinishClasses(const int16_t *bout, int16_t *out_ring, const int16_t *pass_ring, int64_t ring_len,
                                   const int16_t *pass_in, int64_t io_stride, int16_t *out, int n,
                                   const int32_t *class_of_stream, const TickClassEntry *table, int n_streams,
                                   hipStream_t stream);

// One tick of S streaming sessions as ONE launch, one wavefront per session: append the tick's far / near
// (/ clean) samples to the rings, run the session's blocks with their inputs fetched through the source
// codes (no intermediate block buffers), write the block outputs to the output ring and assemble the
// tick's n output samples (this tick's block outputs are kept in LDS for that).
//   class_of_stream / table == nullptr: every session uses `single`; else session s uses table[class_of_stream[s]].
struct TickIo {
    const int16_t *far_in, *near_in, *clean_in;   // [S][io_stride]; clean_in may be null
    int16_t *out;                                 // [S][io_stride]
    int64_t io_stride;
    int32_t n;                                    // samples of this tick (80 or 160)
    int16_t *far_ring, *near_ring, *clean_ring, *out_ring;   // [S][ring_len]
    int64_t ring_len, near_pos;
};
hipError_t LaunchTick(const StatePtrs &st, const TickIo &io, int n_streams, int variant, const int32_t *class_of_stream,
                      const TickClassEntry *table, const TickClassEntry *single, hipStream_t stream);

// The lean form of the one-launch tick.  The session machinery moves samples in long runs (a jitter-buffer frame, a
// re-read of old content, a stretch of never-written zeros), so instead of one source code per sample a block's 64
// inputs / an output frame's 80 samples are described by at most kTickMaxRuns runs of consecutive ring positions.
// The wave appends the tick's samples to its rings FIRST, then reads everything back from the rings (one uniform base
// + a per-lane position; workgroup-scope fences order the wave's own stores and loads), so the per-lane work of a
// fetch is "add, mask" for the usual single run and the whole description sits in scalar registers.  A tick whose
// description does not fit (more runs) falls back to the coded forms above.
constexpr int kTickMaxRuns = 4;
constexpr int32_t kTickRunZero = INT32_MIN;   // off value of a run of zeros (never-written buffer memory)
struct TickRuns {
    int32_t n;                       // runs in use (>= 1)
    int32_t end[kTickMaxRuns];       // exclusive end index, within the block / frame, of run k
    int32_t off[kTickMaxRuns];       // sample i of run k sits at ring position (i + off[k]) & (ring_len - 1); kTickRunZero: zeros
    int32_t kind[kTickMaxRuns];      // output frames only: kTickFromRing = output ring, kTickNearRing = near (clean) ring
};
struct TickLeanEntry {
    int32_t n_blocks;                // blocks of this tick (<= 4)
    int32_t n_far;                   // how many of the tick's far samples the jitter buffer accepted (the first n_far)
    int32_t n_frames;                // output frames of 80 samples (1 or 2)
    int32_t far2_src, far2_cnt;      // a second accepted piece of the far row (two 80-sample calls, the first one cut short
                                     // by a full jitter buffer): input samples [far2_src, far2_src + far2_cnt) follow the first n_far
    int32_t reserved;
    int64_t far_pos, out_pos;        // where the accepted far samples / this tick's block outputs go in their rings
    TickRuns far[4], near[4];        // per block
    TickRuns out[2];                 // per output frame
};
hipError_t LaunchTickLean(const StatePtrs &st, const TickIo &io, int n_streams, const int32_t *class_of_stream,
                          const TickLeanEntry *table, const TickLeanEntry *single, hipStream_t stream);

// The device-resident session machinery: the wrapper itself (jitter buffer, start-up gating, delay compensation,
// 80 -> 64 re-blocking, output stuffing) as position arithmetic on per-session state in HBM (aecm_flow_plan.h), so every
// session has its own msInSndCardBuf / call pattern / age and the host does nothing per session.  A tick is two launches:
//   aecm_flow_plan_kernel   one LANE per session: FlowTick on the session's state (field-major, coalesced) -> its plan
//   aecm_tick_flow_kernel   one WAVEFRONT per session: plan into scalar registers, append the tick's samples to the rings,
//                           frame the far end, run the blocks, assemble the output
// Per session in HBM besides TickIo's rings: kFlowFieldsUsed int32 of wrapper state, kFlowPlanWords of plan, a ring of
// kFlowFarFrameRing samples holding the framed far stream, and the two 80-sample replay rows of far-end underruns
// (farendOld, echo_control_mobile.cc:57).
struct TickFlowIo {
    int32_t *state;                    // [kFlowFieldsUsed][S]
    int32_t *plans;                    // [S][kFlowPlanWords]
    int16_t *far_frames;               // [S][kFlowFarFrameRing]
    int16_t *far_old;                  // [S][2 * 80]
    const int16_t *ms_per_session;     // [S] or null: everybody gets `ms`
    const uint8_t *flags_per_session;  // [S] or null: everybody gets `flags` (kFlowNoFarend | kFlowSplitCalls)
    int32_t ms, flags, fs;
};
hipError_t LaunchTickFlow(const StatePtrs &st, const TickIo &io, const TickFlowIo &fio, int n_streams, hipStream_t stream);

// Diagnostics: `count` independent 128-point transforms of the block kernel's fft128, one wavefront
// each, on natural-order data (data[k] = re[128] then im[128] of transform k, in place).  variant:
// 0 forward of real input, 1 forward complex, 2 inverse; scales[k] = the inverse's accumulated scale.
// Outputs nobody consumes in the block path are returned as 0 (forward: im of bins >= 64; inverse: im).
hipError_t LaunchFft128(int16_t *data_dev, int32_t *scales_dev, int variant, int fast, int count, const uint32_t *consts_dev,
                        hipStream_t stream);

// Audit build only (-DAECM_CHECKED): how often a "provably fits" precondition of the block DSP was violated on
// the device since the last reset -- [0] mul24 operands, [1] as_i16 arguments (aecm_ops.h).  Synchronises the
// device.  hipErrorNotSupported in the shipped build, whose kernels carry no checks.
hipError_t ReadCheckCounters(uint64_t counters[2], bool reset);

// Device self test of the wave primitives; counters[0..7] are failure counts (all must be 0):
//  0 shfl_xor, 1 exchange, 2 reduce_max/min/add, 3 shift_up1, 4 bpermute/readlane/writelane, 5 ballot,
//  6 isqrt31 (exhaustive over [0, 2^31) when exhaustive != 0, else 2^24 samples), 7 table upload.
hipError_t LaunchSelfTest(uint64_t *counters_dev, int exhaustive, const uint32_t *consts_dev, hipStream_t stream);

}  // namespace aecm
#endif  // AECM_AMD_KERNELS_H_
