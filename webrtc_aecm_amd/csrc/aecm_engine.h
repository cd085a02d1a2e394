// Host-side owner of the device state of S AECM streams and of the HIP stream they run on.
#ifndef AECM_AMD_ENGINE_H_
#define AECM_AMD_ENGINE_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "aecm_host_state.h"
#include "aecm_kernels.h"
#include "aecm_state.h"

namespace aecm {

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
    bool SetConfig(int cng_mode, int echo_mode, int first, int count);
    bool SetCngMode(int cng_mode, int first, int count);
    bool Control(int fixed_delay, int nlp_flag, int first, int count);
    // async; blocks_per_stream_dev (may be null): per-stream block counts <= num_blocks
    bool ProcessBlocks(const IoView &io_dev, int num_blocks, const int32_t *blocks_per_stream_dev = nullptr);
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
    static constexpr size_t kStateBytes = kVecWordsPerStream * 4 + kNumScal * 4 + kHistWordsPerStream * 2;
    bool ExportState(int stream, void *buf);
    bool ImportState(int stream, const void *buf);
    void set_variant(int v) { variant_ = v; }
    int variant() const { return variant_; }
    const StatePtrs &state_ptrs() const { return st_; }      // for kernels launched by the session batch on stream()

private:
    BatchEngine() = default;
    bool PatchScalars(const int32_t *fields, const int32_t *values, int n, int first, int count);
    bool FlushTimers();

    int device_ = 0;
    int num_streams_ = 0;
    bool initialized_ = false;
    int variant_ = kVariantFast;
    int fs_ = 0;
    hipStream_t stream_ = nullptr;
    StatePtrs st_{nullptr, nullptr, nullptr, nullptr};
    uint32_t *consts_dev_ = nullptr;     // kernel constants blob (aecm_state.h)
    uint32_t *image_vec_dev_ = nullptr;
    int32_t *image_scal_dev_ = nullptr;
    int32_t *patch_dev_ = nullptr;       // 2 x 16 ints
    hipEvent_t ev_start_ = nullptr, ev_stop_ = nullptr;
    bool timed_pending_ = false;
    float last_ms_ = 0.f;
    double total_ms_ = 0.0;
    int64_t launches_ = 0;
    // staging for ProcessBlocksHost
    int16_t *stage_dev_ = nullptr;
    size_t stage_elems_ = 0;
    static constexpr int kHostChunkStreams = 8192;
    bool ProcessBlocksHostPipelined(const IoView &io_host, int num_blocks);
    hipStream_t download_stream_ = nullptr;    // ProcessBlocksHost: downloads overlap the next chunk's uploads
};

}  // namespace aecm
#endif  // AECM_AMD_ENGINE_H_
