// Host-side owner of the device state of S AECM streams and of the HIP stream they run on.
#ifndef AECM_AMD_ENGINE_H_
#define AECM_AMD_ENGINE_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "aecm_host_state.h"
#include "aecm_kernels.h"
#include "aecm_state.h"

namespace aecm {

constexpr int kDefaultQueueChunk = 128;

// How ProcessBlocks launches are scheduled on a device (results never depend on it): the thresholds between the launch forms and
// the wishes for the pipelined form's shape.  One value type, set through the C ABI (include/aecm_batch.h: AecmLaunchPolicy mirrors
// it field by field); DefaultLaunchPolicy derives it from the device's compute units alone.  The environment is consulted only by
// an -DAECM_EXPERIMENTS build (ApplyEnvironmentWishes).
struct LaunchPolicy {
    int compute_units = 0;
    int queue_chunk_blocks = kDefaultQueueChunk;   // chunk queue: blocks per item; 0 = every launch keeps one wavefront per stream
    bool queue_chunk_explicit = false;             // set by the caller: taken as it is (else quartered while every stream's wave is resident)
    int queue_min_streams = -1;                    // the queue above this many streams; < 0: above pipelined_max_streams
    int pipelined_min_streams = 2;                 // a single stream (the drop-in ABI's 10 ms calls) keeps its one-wave launch; > num_streams: never
    // launches of one or two blocks keep one wave per stream: the pipelined kernel's fill and drain steps (up to five with the sixteen-wave
    // shape) cost more than they save there -- 1 024 streams x 1 / 2 / 3 / 4 blocks: 10.8 / 12.4 / 14.3 / 15.9 us against 8.7 / 11.7 / 15.1 / 18.2 us
    int pipelined_min_blocks = 3;
    int pipelined_max_streams = 0;                 // what the chip holds of the pipelined form's widest shape (16 streams per CU)
    int resident_waves = 0;                        // waves of the one-wave-per-stream kernels the chip holds (28 per CU)
    int rotation_stream_limit = 0;                 // launches of at most this many streams take the kernel variants built for full residency
    PipeWishes pipe;
};
LaunchPolicy DefaultLaunchPolicy(int compute_units);
bool LaunchPolicyValid(const LaunchPolicy &p);
// What a ProcessBlocks launch looks like on the device (WebRtcAecmBatch_DescribeLaunchDetail; capacity planning).
struct LaunchDescription {
    int form = 0;                   // AECM_LAUNCH_*
    int chunk_blocks = 0;           // chunk queue: blocks per item
    int shape = 0;                  // pipelined: the shape bits of WebRtcAecmBatch_DescribeLaunch
    int workgroups = 0, waves_per_workgroup = 0;
    int workgroups_per_cu = 0;      // of this kernel a CU holds at once
    int rounds_x1000 = 0;           // 1000 x workgroups / (CUs x workgroups_per_cu): 1000 = the chip exactly full once; 9140 = nine full rounds and one 14 % full
    int cu_load_evenness_x1000 = 1000;   // pipelined: 1000 x (streams / CUs) / streams on the fullest CU
};
LaunchDescription DescribeLaunchWith(const LaunchPolicy &policy, int variant, int num_streams, int num_blocks, bool has_clean);
LaunchDescription DescribeTickLaunch(int num_sessions, int compute_units);

class BatchEngine {
public:
    // Returns nullptr if the device cannot be used or memory cannot be allocated.
    static BatchEngine *Create(int num_streams, int device_id);
    ~BatchEngine();

    int num_streams() const { return num_streams_; }
    bool initialized() const { return initialized_; }
    hipStream_t stream() const { return stream_; }

    // All methods return a hipError_t-free status: true on success.
    bool Init(int fs);
    // Re-initialise streams [first, first + count) only (same rate as the last Init; default config): WebRtcAecm_InitCore
    // + default set_config for those streams, asynchronous on stream().
    bool InitStreams(int first, int count);
    bool SetConfig(int cng_mode, int echo_mode, int first, int count);
    bool SetCngMode(int cng_mode, int first, int count);
    bool Control(int fixed_delay, int nlp_flag, int first, int count);
    // async; blocks_per_stream_dev (may be null): per-stream block counts <= num_blocks
    bool ProcessBlocks(const IoView &io_dev, int num_blocks, const int32_t *blocks_per_stream_dev = nullptr);
    bool ProcessBlocksRange(const IoView &io_dev, int num_blocks, int first, int count, const int32_t *blocks_per_stream_dev);
    bool ProcessBlocksHost(const IoView &io_host, int num_blocks);     // sync
    // Whole recordings as sessions: every stream is driven like a fresh WebRtcAecm_* session by
    // n_calls x (BufferFarend, Process) of `frame` samples with a constant msInSndCardBuf
    // (aecm_session_flow.h).  far/near/clean/out: [S][>= n_calls*frame], device (or host) pointers; clean
    // (WebRtcAecm_Process's nearendClean) may be null.
    // *rc receives the ABI return code a single session would have produced (0 or 12100).
    bool ProcessRecordings(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stream_stride,
                           int frame, int n_calls, int16_t ms, bool host_pointers, int32_t *rc);
    int fs() const { return fs_; }
    bool Synchronize();
    bool LastLaunchMs(float *ms);
    bool Timers(double *total_ms, int64_t *launches);
    void ResetTimers();
    bool SetEchoPath(int stream, const int16_t path[kBins]);
    bool GetEchoPath(int stream, int16_t path[kBins]);
    bool Digest(int stream, uint32_t digest[kDigestWords]);
    static constexpr size_t kStateHeaderBytes = 32;
    static constexpr size_t kStateBytes = kStateHeaderBytes + kVecWordsPerStream * 4 + kNumScal * 4 + kHistWordsPerStream * 2;
    bool ExportState(int stream, void *buf);
    int32_t ImportState(int stream, const void *buf);       // 0 / AECM_BAD_PARAMETER_ERROR / AECM_UNSPECIFIED_ERROR
    // The same for streams [first, first + count) at once: `states` = count blobs of kStateBytes, host memory or (device =
    // true) anything the device can address -- device memory, or the alias of a registered host buffer.  One gather /
    // scatter launch (+ one copy per chunk of streams for host memory) instead of three blocking copies per stream.
    // ImportStates is all or nothing: every blob is validated like ImportState validates one (on the device for device
    // blobs) before any stream is touched.
    bool ExportStates(int first, int count, void *states, bool device);
    int32_t ImportStates(int first, int count, const void *states, bool device);
    void set_variant(int v) { variant_ = v; }
    // blocks = 0: every launch in the one-stream-per-wave form.  min_streams < 0: launches of more streams than the chip
    // holds waves take the queue form (the default); otherwise launches of more than min_streams streams do (tests).
    void set_queue_chunk(int blocks, int min_streams) {
        policy_.queue_chunk_blocks = blocks < 0 ? 0 : blocks > kMaxQueueChunk ? kMaxQueueChunk : blocks;
        policy_.queue_chunk_explicit = true;
        policy_.queue_min_streams = min_streams;
    }
    static constexpr int kMaxQueueChunk = 1 << 20;
    int DescribeLaunch(int num_blocks, bool has_clean, int *chunk_blocks) const;
    // n <= 0: never.  A threshold set through the ABI is taken as it is: launches of any length from n streams
    void set_pipelined_min_streams(int n) { policy_.pipelined_min_streams = n > 0 ? n : 0x7fffffff; policy_.pipelined_min_blocks = 1; }
    const LaunchPolicy &launch_policy() const { return policy_; }
    bool set_launch_policy(const LaunchPolicy &p);       // false (nothing changed): not a valid policy, or one for another CU count than the device's
    int variant() const { return variant_; }
    const StatePtrs &state_ptrs() const { return st_; }      // for kernels launched by the session batch on stream()

private:
    BatchEngine() = default;
    bool PatchScalars(const int32_t *fields, const int32_t *values, int n, int first, int count);
    bool FlushTimers() { return HarvestTimers(true); }

    int device_ = 0;
    LaunchPolicy policy_;                // DefaultLaunchPolicy of device_'s CU count until the caller sets another
    // The chunk-queue form's control words (grown on first use) and the error word a wave raises when it gives up waiting.
    uint32_t *queue_ctl_ = nullptr, *queue_err_ = nullptr;
    size_t queue_ctl_bytes_ = 0;
    bool queue_unchecked_ = false;       // a queue launch has been enqueued since the error word was last read
    bool launch_failed_ = false;         // a wave of a queue launch gave up: sticky until Init (CheckQueueError)
    bool LaunchBlocks(const StatePtrs &st, const IoView &io, int count, int num_blocks, const int32_t *blocks_per_stream_dev);
    bool CheckQueueError();
    bool Drain();
    bool EnsureLaunchErrorWord();
    bool EnsureLaunchControl(size_t need);
    int trace_streams_ = 0;              // diagnostics builds (-DAECM_PIPE_TRACE) only
    int num_streams_ = 0;
    bool initialized_ = false;
    int variant_ = kVariantFast;
    int fs_ = 0;
    hipStream_t stream_ = nullptr;
    StatePtrs st_{nullptr, nullptr, nullptr, nullptr};
    uint32_t *consts_dev_ = nullptr;     // kernel constants blob (aecm_state.h)
    uint32_t *image_vec_dev_ = nullptr;
    int32_t *image_scal_dev_ = nullptr;
    // Launch timing: a small ring of HIP event pairs recorded around every block-kernel launch on stream_.
    // ProcessBlocks only harvests pairs that have already completed (hipEventQuery) and waits for the oldest
    // one only when the ring is full, so launches queue back to back; the getters harvest everything.
    static constexpr int kTimerSlots = 16;
    hipEvent_t ev_start_[kTimerSlots] = {}, ev_stop_[kTimerSlots] = {};
    int timer_head_ = 0, timer_pending_ = 0;     // oldest pending slot, number of pending slots
    bool HarvestTimers(bool wait_all);
    float last_ms_ = 0.f;
    double total_ms_ = 0.0;
    int64_t launches_ = 0;
    // staging for ProcessBlocksHost
    int16_t *stage_dev_ = nullptr;
    size_t stage_elems_ = 0;
    static constexpr int kHostChunkStreams = 8192;
    bool ProcessBlocksHostPipelined(const IoView &io_host, int num_blocks);
    // Small host calls (the single-session ABI: one to three blocks of one stream): the kernel reads its inputs from and
    // writes its output to a pinned host buffer mapped into the device's address space, so a call is two small host
    // copies, one launch and one synchronisation -- no staging copies, no timing events.
    static constexpr size_t kMappedBytes = 64 * 1024;
    int16_t *mapped_host_ = nullptr, *mapped_dev_ = nullptr;
    bool ProcessBlocksHostMapped(const IoView &io_host, int num_blocks);
    // scratch of ProcessRecordings (grow-only; the streams of a batch are processed in chunks that fit it)
    static constexpr size_t kRecordingScratchBytes = size_t(1) << 30;
    int32_t *rec_maps_ = nullptr;
    size_t rec_maps_elems_ = 0;
    int16_t *rec_scratch_ = nullptr;
    size_t rec_scratch_elems_ = 0;
    bool EnsureRecordingScratch(size_t map_elems, size_t sample_elems);
    bool mixed_rates_ = false;            // ImportState brought in a stream of the other sampling rate
    // staging of the host forms of ExportStates / ImportStates (grow-only, at most kStateStageStreams blobs) + the validation verdict
    static constexpr int kStateStageStreams = 8192;
    uint8_t *state_stage_ = nullptr;
    size_t state_stage_bytes_ = 0;
    uint32_t *state_verdict_ = nullptr;
    bool EnsureStateStage(int streams);
    hipStream_t download_stream_ = nullptr;    // ProcessBlocksHost: downloads overlap the next chunk's uploads
};

}  // namespace aecm
#endif  // AECM_AMD_ENGINE_H_
