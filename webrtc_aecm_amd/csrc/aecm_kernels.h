// Launchers of the gfx950 kernels (aecm_block_kernels.hip, aecm_kernels.hip).  Host-callable, HIP runtime types only.
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
// instead (batches whose streams have different amounts of audio pending).
// rotation_stream_limit = RotationStreamLimit(CU count of the device the launch runs on): launches of at most that many
// streams are fully resident from the start and take the kernel variants built for that occupancy.
hipError_t LaunchProcessBlocks(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, int variant,
                               int rotation_stream_limit, hipStream_t stream, const int32_t *blocks_per_stream = nullptr);
int RotationStreamLimit(int compute_units);

// The chunk-queue form of the same launch (aecm_block_kernels.hip): items of chunk_blocks blocks claimed in order by a
// grid that just fills the chip.  ctl: QueueControlBytes(n_streams) of device memory owned by the engine (cleared by the
// launch); *err (device, never cleared by a launch) becomes non-zero if a wave gave up waiting for its predecessor.
size_t QueueControlBytes(int n_streams);
bool QueueLaunchApplies(int n_streams, int n_blocks, int variant, int chunk_blocks, int min_streams, bool ragged);
hipError_t LaunchProcessBlocksQueued(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, int chunk_blocks,
                                     int resident_waves, uint32_t *ctl, uint32_t *err, hipStream_t stream);
int ResidentWaves(int compute_units);

// The pipelined form of a launch the chip holds at once (aecm_block_kernels.hip): a workgroup serves four streams with one
// "back" wave per stream and, in waves of their own, the state-independent forward transforms one block ahead (front waves)
// and -- in some shapes -- the inverse transforms one block behind (tail waves).  Fast variant, no clean input, every stream
// the same number of blocks.
struct PipeShape {
    int tail_waves;      // 0 or 2 per workgroup
    int front_waves;     // 2 (two streams each) or 4 (one each; with tail waves only)
    bool raw;            // the front waves hand over the transforms' outputs, the back waves form the spectra
    bool balance;        // progress feedback on the front waves' issue priority (needs `progress`)
    int delay_waves;     // 0, 4 or (with gain waves) 2 per workgroup: the delay estimator in waves of its own, one block ahead of the back waves (with tail waves, not raw)
    int gain_waves;      // 0 or 4 per workgroup: the gain half of the back waves' work in waves of its own, one block behind the channel half (with delay waves)
    int wgs_per_round;   // the device's CUs: workgroups i, i + wgs_per_round, ... are taken to share a CU (the dispatcher deals them out in turn) and start their slots on different SIMDs
    int rot;             // slot rotations: front | gain << 2 | delay << 4 | per workgroup of a CU << 6 | tail << 8 (the kernel's slot_of)
    int prio;            // the roles' issue priorities: front | tail << 2 | delay << 4 | gain << 6 (the channel / middle / back waves': by phase, 1..3)
    int workgroups;      // of the launch: ceil(n_streams / 4) .. n_streams; the streams are dealt out evenly (the first n_streams % workgroups serve one more)
};
// What a caller may wish for the shape instead of leaving it to the launch's size (experiments, tests: AecmLaunchPolicy in
// include/aecm_batch.h carries the same fields); < 0 = by size.  A wish is taken where the shape exists and fits.
struct PipeWishes {
    int tail_waves = -1, front_waves = -1, raw = -1, delay_waves = -1, gain_waves = -1;
    int spread = 1;          // != 0: every CU gets the shape's full count of workgroups, of fewer than four streams each where the launch is short of streams
    int wgs_per_cu = 0;      // > 0: workgroups of the shape a CU takes (0: what the shape is built for)
    int rot = -1;            // >= 0: PipeShape::rot
    int prio = -1;           // >= 0: PipeShape::prio
};
// The shape a launch of this size takes on a device of compute_units CUs.
PipeShape PipelinedShapeFor(int n_streams, int n_blocks, int compute_units, const PipeWishes &wishes = PipeWishes());
int PipelinedStreamLimit(int compute_units, int tail_waves, int front_waves = 2, int delay_waves = 0, int gain_waves = 0, int wgs_per_cu = 0);
// Waves of a workgroup of this shape, and how many such workgroups a CU holds at once (by the wave slots the kernel is built for and the LDS it declares).
int PipelinedWorkgroupWaves(const PipeShape &shape);
int PipelinedWorkgroupsPerCu(const PipeShape &shape);
// progress: PipelinedControlBytes(shape.workgroups) of device memory owned by the engine (cleared by the launch): 16 bits per
// workgroup, by which the workgroups of a balanced launch keep in step.
size_t PipelinedControlBytes(int n_workgroups);
size_t PipelinedTraceOffsetBytes(int n_workgroups);     // diagnostics builds (-DAECM_PIPE_TRACE): where the per-wave records follow the progress words
hipError_t LaunchProcessBlocksPipelined(const StatePtrs &st, const IoView &io, int n_streams, int n_blocks, const PipeShape &shape, uint32_t *progress,
                                        hipStream_t stream);

// Replicate one stream image (vec: kNumVec*64 words, scal: 64 words, both on the device) into
// streams [first, first + count) and clear their far-spectrum history.
hipError_t LaunchBroadcastImage(const StatePtrs &st, const uint32_t *image_vec, const int32_t *image_scal,
                                int first, int count, hipStream_t stream);

// Overwrite a few scalar fields (field ids from ScalField) of streams [first, first + count).  The fields travel as a
// kernel argument: no staging copy, nothing for the host to wait for.
constexpr int kMaxPatchFields = 16;
struct ScalarPatch {
    int32_t n;
    int32_t field[kMaxPatchFields], value[kMaxPatchFields];
};
hipError_t LaunchPatchScalars(const StatePtrs &st, const ScalarPatch &patch, int first, int count, hipStream_t stream);

// Bulk state snapshots of streams [first, first + count): blobs_dev = count blobs of kStateBlobBytes (aecm_state_check.h:
// 32-byte header + lane vectors + scalars + far-spectrum history), anything the device can address.  Validate: result_dev[0]
// (preset to 0xffffffff) becomes the index of the first blob the kernels may not run on, result_dev[1] (preset to 0) gets bit 0
// when a blob of another sampling rate than fs_batch is among them; Scatter writes the blobs into the streams unchecked.
hipError_t LaunchGatherStates(const StatePtrs &st, int first, int count, void *blobs_dev, hipStream_t stream);
hipError_t LaunchValidateStates(const void *blobs_dev, int count, int fs_batch, uint32_t *result_dev, hipStream_t stream);
hipError_t LaunchScatterStates(const StatePtrs &st, int first, int count, const void *blobs_dev, hipStream_t stream);

// Session-schedule gather / scatter (aecm_session_flow.h: RecordingSchedule), all streams at once.
//   dst[s][j] = map[j] >= 0 ? src[s*src_stride + map[j]] : 0            j in [0, n)
hipError_t LaunchGatherByMap(const int16_t *src, int64_t src_stride, const int32_t *map_dev, int64_t n, int16_t *dst,
                             int64_t dst_stride, int n_streams, hipStream_t stream);
//   out[s][j] = v >= 0 ? blocks[s][v] : (v == -1 ? 0 : near[s][-(v+2)])  with v = map[j]
hipError_t LaunchAssembleOutput(const int16_t *blocks, int64_t blocks_stride, const int16_t *near, int64_t near_stride,
                                const int32_t *map_dev, int64_t n, int16_t *out, int64_t out_stride, int n_streams,
                                hipStream_t stream);

// Streaming sessions (aecm_sessions.cpp): per-session sample rings of `ring_len` (power of two) int16 in HBM, indexed by
// wrapping stream positions.
struct TickIo {
    const int16_t *far_in, *near_in, *clean_in;   // [S][io_stride]; clean_in may be null
    int16_t *out;                                 // [S][io_stride]
    int64_t io_stride;
    int32_t n;                                    // samples of this tick (80 or 160)
    int16_t *far_ring, *near_ring, *clean_ring, *out_ring;   // [S][ring_len]
    int64_t ring_len, near_pos;
};

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
int TickWorkgroupWaves();        // sessions per workgroup of the tick kernel
int TickWorkgroupsPerCu();       // workgroups of it a CU holds at once
// Far-end bursts: WebRtcAecm_BufferFarend calls WITHOUT a Process (reference echo_control_mobile.cc:215-234), one wavefront
// per session.  Session s makes clamp(calls_per_session[s] - call_base, 0, max_calls) calls of io.n samples (max_calls each
// when calls_per_session is null) on io.far_in[s][c * io.n .. + io.n): delay compensation when past the start-up phase, then
// what fits into the jitter buffer goes to the far ring.  Of io only far_in / io_stride / n / far_ring / ring_len are used.
hipError_t LaunchBufferFarend(const TickIo &io, const TickFlowIo &fio, const uint8_t *calls_per_session, int call_base, int max_calls, int n_streams,
                              hipStream_t stream);
// WebRtcAecm_Init of the wrapper side of sessions [first, first + count): wrapper state as after Init (aecm_flow_plan.h:
// FlowFieldStartsAtOne), far / output rings, framed-far ring and replay rows reading as never written (zero).  Only the
// ring pointers, ring_len and the state / far_frames / far_old pointers of io / fio are used.
hipError_t LaunchResetSessions(const TickIo &io, const TickFlowIo &fio, int n_streams, int first, int count, hipStream_t stream);

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
// The share of aecm_block_kernels.hip (device symbols are per translation unit); ReadCheckCounters adds it in.  Does not synchronise.
hipError_t ReadBlockKernelCheckCounters(uint64_t counters[2], bool reset);

// Device self test of the wave primitives; counters[0..7] are failure counts (all must be 0):
//  0 shfl_xor, 1 exchange, 2 reduce_max/min/add, 3 shift_up1, 4 bpermute/readlane/writelane, 5 ballot,
//  6 isqrt31 (exhaustive over [0, 2^31) when exhaustive != 0, else 2^24 samples), 7 table upload.
hipError_t LaunchSelfTest(uint64_t *counters_dev, int exhaustive, const uint32_t *consts_dev, hipStream_t stream);

}  // namespace aecm
#endif  // AECM_AMD_KERNELS_H_
