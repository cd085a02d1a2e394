#include "aecm_engine.h"

#include "aecm_session_flow.h"
#include "aecm_state_check.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace aecm {

#define AECM_HIP_OK(expr) ((expr) == hipSuccess)

// (kDefaultQueueChunk: launches of more streams than the chip holds waves are cut into chunks of that many blocks and scheduled from
// a queue -- aecm_block_kernels.hip: aecm_process_queue_kernel; measured: profiles/r04_experiments.md section 1.)

#if defined(AECM_EXPERIMENTS)
// Experiments build only (tools/ab_build.py adds -DAECM_EXPERIMENTS): the environment's wishes on top of the default policy, so that
// tools/sweep_streams.py and the A/B scripts can try launch forms without a caller that sets a policy.  The shipped library reads no
// environment variable anywhere.
static void ApplyEnvironmentWishes(LaunchPolicy *p) {
    auto num = [](const char *name, int *dst) { if (const char *env = getenv(name)) *dst = atoi(env); };
    if (const char *env = getenv("AECM_QUEUE_CHUNK")) p->queue_chunk_blocks = std::min(std::max(0, atoi(env)), BatchEngine::kMaxQueueChunk);
    num("AECM_QUEUE_MIN_STREAMS", &p->queue_min_streams);
    num("AECM_PIPE_FRONT", &p->pipe.front_waves);
    num("AECM_PIPE_TAIL", &p->pipe.tail_waves);
    num("AECM_PIPE_RAW", &p->pipe.raw);
    num("AECM_PIPE_DELAY", &p->pipe.delay_waves);
    num("AECM_PIPE_GAIN", &p->pipe.gain_waves);
    num("AECM_PIPE_SPREAD", &p->pipe.spread);
    num("AECM_PIPE_WGS", &p->pipe.wgs_per_cu);
    num("AECM_PIPE_ROT", &p->pipe.rot);
    num("AECM_PIPE_PRIO", &p->pipe.prio);
    num("AECM_PIPE_MAX_STREAMS", &p->pipelined_max_streams);
    if (const char *env = getenv("AECM_PIPELINED")) p->pipelined_min_streams = atoi(env) > 0 ? atoi(env) : 0x7fffffff;   // 0: off; n: from n streams
    if (const char *env = getenv("AECM_PIPE_MIN_BLOCKS")) p->pipelined_min_blocks = atoi(env) > 1 ? atoi(env) : 1;
}
#endif

// The launch-form limits of a device with `compute_units` CUs: everything follows from the CU count and from what the kernels are
// built for (waves per SIMD, LDS per workgroup: aecm_block_kernels.hip; tests/test_capi.py checks those against the code object).
LaunchPolicy DefaultLaunchPolicy(int compute_units) {
    LaunchPolicy p;
    p.compute_units = compute_units;
    p.rotation_stream_limit = RotationStreamLimit(compute_units);
    p.resident_waves = ResidentWaves(compute_units);
    p.pipelined_max_streams = PipelinedStreamLimit(compute_units, 0);      // the six-wave shape, four workgroups per CU
#if defined(AECM_EXPERIMENTS)
    ApplyEnvironmentWishes(&p);
#endif
    return p;
}

bool LaunchPolicyValid(const LaunchPolicy &p) {
    const PipeWishes &w = p.pipe;
    auto in = [](int v, std::initializer_list<int> ok) { return std::find(ok.begin(), ok.end(), v) != ok.end(); };
    return p.compute_units >= 0 && p.queue_chunk_blocks >= 0 && p.queue_chunk_blocks <= BatchEngine::kMaxQueueChunk && p.pipelined_min_blocks >= 1 &&
           p.pipelined_max_streams >= 0 && p.pipelined_max_streams <= PipelinedStreamLimit(p.compute_units, 0) && p.resident_waves > 0 &&
           p.resident_waves <= ResidentWaves(p.compute_units) && p.rotation_stream_limit >= 0 && in(w.tail_waves, {-1, 0, 2}) &&
           in(w.front_waves, {-1, 2, 4}) && in(w.raw, {-1, 0, 1}) && in(w.delay_waves, {-1, 0, 2, 4}) && in(w.gain_waves, {-1, 0, 4}) &&
           in(w.spread, {0, 1}) && w.wgs_per_cu >= 0 && w.wgs_per_cu <= 8 && w.rot >= -1 && w.rot < 1024 && w.prio >= -1 && w.prio < 256;
}

BatchEngine *BatchEngine::Create(int num_streams, int device_id) {
    if (num_streams <= 0) return nullptr;
    int n_dev = 0;
    if (!AECM_HIP_OK(hipGetDeviceCount(&n_dev)) || device_id < 0 || device_id >= n_dev) return nullptr;
    if (!AECM_HIP_OK(hipSetDevice(device_id))) return nullptr;
    BatchEngine *e = new BatchEngine();
    e->device_ = device_id;
    e->num_streams_ = num_streams;
    int cus = 0;
    if (!AECM_HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id))) cus = 0;
    e->policy_ = DefaultLaunchPolicy(cus);
    const size_t S = (size_t)num_streams;
    bool ok = AECM_HIP_OK(hipStreamCreateWithFlags(&e->stream_, hipStreamNonBlocking)) &&
              AECM_HIP_OK(hipMalloc((void **)&e->st_.vec, S * kVecWordsPerStream * sizeof(uint32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&e->st_.scal, S * kNumScal * sizeof(int32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&e->st_.hist, S * kHistWordsPerStream * sizeof(uint16_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&e->image_vec_dev_, kVecWordsPerStream * sizeof(uint32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&e->image_scal_dev_, kNumScal * sizeof(int32_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&e->consts_dev_, kConstBlobWords * sizeof(uint32_t))) &&
              true;                                   // the timing events are created on first use (ProcessBlocks)
    if (ok) {
        // the 20 KB constants blob is built once per process (every WebRtcAecm_Create makes an engine of its own)
        static const std::vector<uint32_t> blob = [] { std::vector<uint32_t> b; BuildKernelConstants(&b); return b; }();
        ok = AECM_HIP_OK(hipMemcpy(e->consts_dev_, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        e->st_.consts = e->consts_dev_;
    }
    if (!ok) {
        delete e;
        return nullptr;
    }
    return e;
}

bool BatchEngine::set_launch_policy(const LaunchPolicy &p) {
    if (!LaunchPolicyValid(p) || p.compute_units != policy_.compute_units) return false;
    policy_ = p;
    return true;
}

BatchEngine::~BatchEngine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (int i = 0; i < kTimerSlots; ++i) {
        if (ev_start_[i]) (void)hipEventDestroy(ev_start_[i]);
        if (ev_stop_[i]) (void)hipEventDestroy(ev_stop_[i]);
    }
    (void)hipFree(st_.vec);
    (void)hipFree(st_.scal);
    (void)hipFree(st_.hist);
    (void)hipFree(image_vec_dev_);
    (void)hipFree(image_scal_dev_);
    (void)hipFree(consts_dev_);
    (void)hipFree(queue_ctl_);
    (void)hipFree(queue_err_);
    (void)hipFree(stage_dev_);
    if (mapped_host_) (void)hipHostFree(mapped_host_);
    (void)hipFree(rec_maps_);
    (void)hipFree(rec_scratch_);
    (void)hipFree(state_stage_);
    (void)hipFree(state_verdict_);
    if (download_stream_) (void)hipStreamDestroy(download_stream_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

bool BatchEngine::Init(int fs) {
    StreamImage img;
    if (!BuildInitImage(fs, &img)) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    if (!AECM_HIP_OK(hipMemcpyAsync(image_vec_dev_, img.vec.data(), img.vec.size() * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, stream_)))
        return false;
    if (!AECM_HIP_OK(hipMemcpyAsync(image_scal_dev_, img.scal.data(), img.scal.size() * sizeof(int32_t),
                                    hipMemcpyHostToDevice, stream_)))
        return false;
    if (!AECM_HIP_OK(LaunchBroadcastImage(st_, image_vec_dev_, image_scal_dev_, 0, num_streams_, stream_))) return false;
    // a launch that failed (a wave of a chunk-queue launch gave up waiting: CheckQueueError) left the streams it had not
    // finished in an unknown state; re-initialising all of them is what clears the record
    if (queue_err_ && !AECM_HIP_OK(hipMemsetAsync(queue_err_, 0, sizeof(uint32_t), stream_))) return false;
    queue_unchecked_ = false;
    launch_failed_ = false;
    if (!AECM_HIP_OK(hipStreamSynchronize(stream_))) return false;   // img goes out of scope
    initialized_ = true;
    fs_ = fs;
    mixed_rates_ = false;
    return true;
}

bool BatchEngine::InitStreams(int first, int count) {
    if (!initialized_ || first < 0 || count < 0 || first + count > num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    // image_*_dev_ still hold the initial image of the last Init
    return AECM_HIP_OK(LaunchBroadcastImage(st_, image_vec_dev_, image_scal_dev_, first, count, stream_));
}

// Stream-ordered, no synchronisation: the next launch on the engine's stream sees the new values.
bool BatchEngine::PatchScalars(const int32_t *fields, const int32_t *values, int n, int first, int count) {
    if (n > kMaxPatchFields) return false;
    if (count < 0) count = num_streams_ - first;
    if (first < 0 || count < 0 || first + count > num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    ScalarPatch patch{};
    patch.n = n;
    memcpy(patch.field, fields, n * sizeof(int32_t));
    memcpy(patch.value, values, n * sizeof(int32_t));
    return AECM_HIP_OK(LaunchPatchScalars(st_, patch, first, count, stream_));
}

bool BatchEngine::SetConfig(int cng_mode, int echo_mode, int first, int count) {
    int32_t scal[kNumScal] = {0};
    if (!ApplyConfig(scal, cng_mode, echo_mode)) return false;
    const int32_t fields[7] = {S_CNG, S_SUPGAIN, S_SUPGAIN_OLD, S_SG_A, S_SG_D, S_SG_DAB, S_SG_DBD};
    int32_t values[7];
    for (int i = 0; i < 7; ++i) values[i] = scal[fields[i]];
    return PatchScalars(fields, values, 7, first, count);
}

bool BatchEngine::SetCngMode(int cng_mode, int first, int count) {
    const int32_t fields[1] = {S_CNG};
    const int32_t values[1] = {cng_mode};
    return PatchScalars(fields, values, 1, first, count);
}

bool BatchEngine::Control(int fixed_delay, int nlp_flag, int first, int count) {
    int32_t scal[kNumScal] = {0};
    ApplyControl(scal, fixed_delay, nlp_flag);
    const int32_t fields[2] = {S_NLP, S_FIXED_DELAY};
    const int32_t values[2] = {scal[S_NLP], scal[S_FIXED_DELAY]};
    return PatchScalars(fields, values, 2, first, count);
}

// Collect the durations of finished launches.  wait_all: block until every pending pair has completed
// (getters); otherwise take only what has already finished -- never blocks.
bool BatchEngine::HarvestTimers(bool wait_all) {
    while (timer_pending_ > 0) {
        const int slot = timer_head_;
        if (wait_all) {
            if (!AECM_HIP_OK(hipEventSynchronize(ev_stop_[slot]))) return false;
        } else {
            const hipError_t q = hipEventQuery(ev_stop_[slot]);
            if (q == hipErrorNotReady) return true;
            if (q != hipSuccess) return false;
        }
        float ms = 0.f;
        if (!AECM_HIP_OK(hipEventElapsedTime(&ms, ev_start_[slot], ev_stop_[slot]))) return false;
        last_ms_ = ms;
        total_ms_ += ms;
        launches_ += 1;
        timer_head_ = (timer_head_ + 1) % kTimerSlots;
        --timer_pending_;
    }
    return true;
}

// The word a wave of a chunk-queue or pipelined launch raises when it gives up waiting for another wave (never cleared by a launch).
bool BatchEngine::EnsureLaunchErrorWord() {
    if (queue_err_) return true;
    return AECM_HIP_OK(hipMalloc((void **)&queue_err_, sizeof(uint32_t))) && AECM_HIP_OK(hipMemsetAsync(queue_err_, 0, sizeof(uint32_t), stream_));
}

// One launch of the block kernels over `count` streams (st, io already offset to the first of them), in the chunk-queue form
// when the launch is larger than the chip (see QueueLaunchApplies).
// The control words of a chunk-queue or pipelined launch (one buffer, grown on first use; launches on stream_ are ordered, so
// consecutive launches may share it).
bool BatchEngine::EnsureLaunchControl(size_t need) {
    if (need <= queue_ctl_bytes_) return true;
    if (!AECM_HIP_OK(hipStreamSynchronize(stream_))) return false;       // stream-ordered work may still read the old one
    (void)hipFree(queue_ctl_);
    queue_ctl_ = nullptr;
    queue_ctl_bytes_ = 0;
    if (!AECM_HIP_OK(hipMalloc((void **)&queue_ctl_, need))) return false;
    queue_ctl_bytes_ = need;
    return true;
}

// Which form a launch of `count` streams x num_blocks blocks takes under a policy (host logic, no device): the rules of
// INTEGRATION.md's table.
//   chunk queue   every launch of more streams than the pipelined form takes (or than queue_min_streams, when set) that is at least
//                 two chunks long: above the chip's resident waves because the launch would otherwise end in a long drain, and between
//                 4 096 streams and that too -- every wave is resident there, but the SIMD's arbiter favours its oldest wave, the waves
//                 dispatched first pull ahead, and with items claimed in order those waves simply take more of the work (4 608 streams
//                 728 -> 830 M frames/s, 6 144: 844 -> 930 M; profiles/r04_experiments.md section 4).  Shorter chunks there: a quarter.
//   pipelined     launches the chip holds at once: fast variant, no clean input, every stream the same number of blocks
namespace {
int QueueMinStreams(const LaunchPolicy &p) { return p.queue_min_streams >= 0 ? p.queue_min_streams : p.pipelined_max_streams; }
// (The quarter rule is for the default chunk; a length set by the caller is taken as it is.)
int QueueChunkFor(const LaunchPolicy &p, int count) {
    if (p.queue_chunk_blocks <= 0 || count > p.resident_waves || p.queue_chunk_explicit) return p.queue_chunk_blocks;
    return std::max(8, p.queue_chunk_blocks / 4);
}
bool PipelinedLaunchApplies(const LaunchPolicy &p, int variant, int count, int num_blocks, bool clean, bool ragged) {
    return variant == kVariantFast && !clean && !ragged && count >= p.pipelined_min_streams && count <= p.pipelined_max_streams &&
           num_blocks >= p.pipelined_min_blocks;
}
int PipeShapeBits(const PipeShape &sh) {
    return sh.tail_waves | (sh.balance ? 0x100 : 0) | (sh.front_waves == 4 ? 0x200 : 0) | (sh.raw ? 0x400 : 0) | (sh.delay_waves ? 0x800 : 0) |
           (sh.gain_waves ? 0x1000 : 0);
}
}  // namespace

LaunchDescription DescribeLaunchWith(const LaunchPolicy &p, int variant, int count, int num_blocks, bool has_clean) {
    LaunchDescription d;
    const int cus = p.compute_units > 0 ? p.compute_units : 256;
    auto rounds = [&](LaunchDescription &x) { x.rounds_x1000 = (int)((int64_t)1000 * x.workgroups / ((int64_t)cus * std::max(1, x.workgroups_per_cu))); };
    const int chunk = QueueChunkFor(p, count);
    if (QueueLaunchApplies(count, num_blocks, variant, chunk, QueueMinStreams(p), false)) {
        d.form = 2;
        d.chunk_blocks = chunk;
        d.waves_per_workgroup = kWavesPerWorkgroup;
        d.workgroups_per_cu = p.resident_waves / (cus * kWavesPerWorkgroup);
        d.workgroups = std::min((count + kWavesPerWorkgroup - 1) / kWavesPerWorkgroup, p.resident_waves / kWavesPerWorkgroup);
        rounds(d);
        return d;
    }
    if (PipelinedLaunchApplies(p, variant, count, num_blocks, has_clean, false)) {
        const PipeShape sh = PipelinedShapeFor(count, num_blocks, p.compute_units, p.pipe);
        d.form = 3;
        d.shape = PipeShapeBits(sh);
        d.workgroups = sh.workgroups;
        d.waves_per_workgroup = PipelinedWorkgroupWaves(sh);
        d.workgroups_per_cu = p.pipe.wgs_per_cu > 0 ? p.pipe.wgs_per_cu : PipelinedWorkgroupsPerCu(sh);
        rounds(d);
        // workgroups i, i + CUs, ... share a CU (the dispatcher deals them out in turn); the first count % workgroups serve one stream more
        {
            const int base = count / sh.workgroups, rem = count % sh.workgroups;
            int fullest = 0;
            for (int c = 0; c < std::min(cus, sh.workgroups); ++c) {
                int load = 0;
                for (int w = c; w < sh.workgroups; w += cus) load += base + (w < rem ? 1 : 0);
                fullest = std::max(fullest, load);
            }
            d.cu_load_evenness_x1000 = (int)((int64_t)1000 * count / ((int64_t)std::max(fullest, 1) * std::min(cus, sh.workgroups)));
        }
        return d;
    }
    d.form = variant == kVariantFast && count > p.rotation_stream_limit ? 1 : 0;
    d.waves_per_workgroup = kWavesPerWorkgroup;
    d.workgroups = (count + kWavesPerWorkgroup - 1) / kWavesPerWorkgroup;
    d.workgroups_per_cu = (d.form == 1 ? p.resident_waves : std::max(p.rotation_stream_limit, 1)) / (cus * kWavesPerWorkgroup);
    rounds(d);
    return d;
}

// A tick of a session batch (aecm_kernels.hip: aecm_tick_flow_kernel): one wavefront per session, four per workgroup, seven such
// workgroups per CU -- 65 536 sessions on 256 CUs are 16 384 workgroups on 1 792 places: 9.14 rounds, the last one 14 % full.
LaunchDescription DescribeTickLaunch(int num_sessions, int compute_units) {
    LaunchDescription d;
    const int cus = compute_units > 0 ? compute_units : 256;
    d.form = 0;
    d.waves_per_workgroup = TickWorkgroupWaves();
    d.workgroups = (num_sessions + d.waves_per_workgroup - 1) / d.waves_per_workgroup;
    d.workgroups_per_cu = TickWorkgroupsPerCu();
    d.rounds_x1000 = (int)((int64_t)1000 * d.workgroups / ((int64_t)cus * d.workgroups_per_cu));
    return d;
}

bool BatchEngine::LaunchBlocks(const StatePtrs &st, const IoView &io, int count, int num_blocks, const int32_t *blocks_per_stream_dev) {
    if (launch_failed_) return false;                 // streams half processed by an abandoned launch: nothing runs until Init
    const int chunk = QueueChunkFor(policy_, count);
    if (QueueLaunchApplies(count, num_blocks, variant_, chunk, QueueMinStreams(policy_), blocks_per_stream_dev != nullptr)) {
        if (!EnsureLaunchControl(QueueControlBytes(count))) return false;
        if (!EnsureLaunchErrorWord()) return false;
        queue_unchecked_ = true;
        return AECM_HIP_OK(LaunchProcessBlocksQueued(st, io, count, num_blocks, chunk, policy_.resident_waves, queue_ctl_, queue_err_, stream_));
    }
    if (PipelinedLaunchApplies(policy_, variant_, count, num_blocks, io.near_clean != nullptr, blocks_per_stream_dev != nullptr)) {
        const PipeShape shape = PipelinedShapeFor(count, num_blocks, policy_.compute_units, policy_.pipe);
        bool need_ctl = shape.balance;
#if defined(AECM_PIPE_TRACE)
        need_ctl = true;                      // (the diagnostics build keeps the buffer: its per-wave records live behind the progress words)
        trace_streams_ = shape.workgroups;
#endif
        if (need_ctl && !EnsureLaunchControl(PipelinedControlBytes(shape.workgroups))) return false;
        return AECM_HIP_OK(LaunchProcessBlocksPipelined(st, io, count, num_blocks, shape, need_ctl ? queue_ctl_ : nullptr, stream_));
    }
    return AECM_HIP_OK(LaunchProcessBlocks(st, io, count, num_blocks, variant_, policy_.rotation_stream_limit, stream_, blocks_per_stream_dev));
}

int BatchEngine::DescribeLaunch(int num_blocks, bool has_clean, int *chunk_blocks) const {
    const LaunchDescription d = DescribeLaunchWith(policy_, variant_, num_streams_, num_blocks, has_clean);
    if (chunk_blocks) *chunk_blocks = d.form == 2 ? d.chunk_blocks : d.form == 3 ? d.shape : 0;
    return d.form;
}

// Wait for everything enqueued on stream_; false if a HIP call failed or a wave of a chunk-queue launch gave up waiting.
bool BatchEngine::Drain() {
    if (!AECM_HIP_OK(hipStreamSynchronize(stream_))) return false;
#if defined(AECM_PIPE_TRACE)     // diagnostics build: the last pipelined launch's per-wave records -> $AECM_PIPE_TRACE_FILE (raw uint64 x 4 per wave)
    if (const char *path = getenv("AECM_PIPE_TRACE_FILE")) {
        if (trace_streams_ > 0) {
            const size_t off = PipelinedTraceOffsetBytes(trace_streams_), bytes = PipelinedControlBytes(trace_streams_) - off;
            std::vector<uint8_t> host(bytes);
            if (AECM_HIP_OK(hipMemcpy(host.data(), reinterpret_cast<uint8_t *>(queue_ctl_) + off, bytes, hipMemcpyDeviceToHost))) {
                if (FILE *f = fopen(path, "wb")) { fwrite(host.data(), 1, bytes, f); fclose(f); }
            }
            trace_streams_ = 0;
        }
    }
#endif
    return CheckQueueError();
}

// After a synchronisation of stream_: did a wave of a chunk-queue launch give up waiting (it never should)?  The verdict
// is sticky: an abandoned launch leaves streams half processed, so every later Drain() (ExportState, Digest, GetEchoPath,
// Synchronize ...) and every later launch fails too, until Init() has re-initialised the streams.
bool BatchEngine::CheckQueueError() {
    if (launch_failed_) return false;
    if (!queue_unchecked_) return true;
    uint32_t err = 0;
    if (!AECM_HIP_OK(hipMemcpy(&err, queue_err_, sizeof err, hipMemcpyDeviceToHost))) return false;      // unchecked stays set: asked again next time
    queue_unchecked_ = false;
    if (err != 0) launch_failed_ = true;
    return err == 0;
}

bool BatchEngine::ProcessBlocks(const IoView &io, int num_blocks, const int32_t *blocks_per_stream_dev) {
    return ProcessBlocksRange(io, num_blocks, 0, num_streams_, blocks_per_stream_dev);
}

// Streams [first, first + count); io (and blocks_per_stream_dev) are indexed from `first`.
bool BatchEngine::ProcessBlocksRange(const IoView &io, int num_blocks, int first, int count, const int32_t *blocks_per_stream_dev) {
    if (first < 0 || count < 0 || first + count > num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    if (!HarvestTimers(false)) return false;
    if (timer_pending_ == kTimerSlots) {         // ring full: wait for the oldest launch only
        if (!AECM_HIP_OK(hipEventSynchronize(ev_stop_[timer_head_])) || !HarvestTimers(false)) return false;
    }
    const int slot = (timer_head_ + timer_pending_) % kTimerSlots;
    if (!ev_start_[slot] && (!AECM_HIP_OK(hipEventCreate(&ev_start_[slot])) || !AECM_HIP_OK(hipEventCreate(&ev_stop_[slot])))) return false;
    if (!AECM_HIP_OK(hipEventRecord(ev_start_[slot], stream_))) return false;
    StatePtrs st = st_;
    st.vec += (size_t)first * kVecWordsPerStream;
    st.scal += (size_t)first * kNumScal;
    st.hist += (size_t)first * kHistWordsPerStream;
    if (!LaunchBlocks(st, io, count, num_blocks, blocks_per_stream_dev)) return false;
    if (!AECM_HIP_OK(hipEventRecord(ev_stop_[slot], stream_))) return false;
    ++timer_pending_;
    return true;
}

bool BatchEngine::ProcessBlocksHost(const IoView &io, int num_blocks) {
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    // Dense repack on the device side: [S][T*64] for each of far / near / (clean) / out.
    const size_t per = (size_t)num_streams_ * num_blocks * kBlock;
    const int n_in = io.near_clean ? 3 : 2;
    const size_t need = per * (n_in + 1);
#if defined(AECM_EXPERIMENTS)
    static const bool mapped = [] { const char *e = getenv("AECM_HOST_MAPPED"); return !(e && e[0] == '0'); }();
#else
    constexpr bool mapped = true;
#endif
    if (mapped && need * sizeof(int16_t) <= kMappedBytes) return ProcessBlocksHostMapped(io, num_blocks);
    if (need > stage_elems_) {
        (void)hipFree(stage_dev_);
        stage_dev_ = nullptr;
        stage_elems_ = 0;
        if (!AECM_HIP_OK(hipMalloc((void **)&stage_dev_, need * sizeof(int16_t)))) return false;
        stage_elems_ = need;
    }
    const bool dense = io.block_stride == kBlock && io.stream_stride == (int64_t)num_blocks * kBlock;
#if defined(AECM_EXPERIMENTS)
    static const bool pipeline = [] { const char *e = getenv("AECM_HOST_PIPELINE"); return !(e && e[0] == '0'); }();
#else
    constexpr bool pipeline = true;
#endif
    if (pipeline && dense && num_streams_ >= 2 * kHostChunkStreams) return ProcessBlocksHostPipelined(io, num_blocks);
    std::vector<int16_t> tmp;
    auto upload = [&](const int16_t *src, int16_t *dst) -> bool {
        if (dense) return AECM_HIP_OK(hipMemcpyAsync(dst, src, per * sizeof(int16_t), hipMemcpyHostToDevice, stream_));
        tmp.resize(per);
        for (int s = 0; s < num_streams_; ++s)
            for (int b = 0; b < num_blocks; ++b)
                memcpy(&tmp[((size_t)s * num_blocks + b) * kBlock], src + s * io.stream_stride + b * io.block_stride,
                       kBlock * sizeof(int16_t));
        return AECM_HIP_OK(hipMemcpy(dst, tmp.data(), per * sizeof(int16_t), hipMemcpyHostToDevice));
    };
    IoView dev;
    dev.far = stage_dev_;
    dev.near = stage_dev_ + per;
    dev.near_clean = io.near_clean ? stage_dev_ + 2 * per : nullptr;
    dev.out = stage_dev_ + (size_t)n_in * per;
    dev.stream_stride = (int64_t)num_blocks * kBlock;
    dev.block_stride = kBlock;
    if (!upload(io.far, const_cast<int16_t *>(dev.far))) return false;
    if (!upload(io.near, const_cast<int16_t *>(dev.near))) return false;
    if (io.near_clean && !upload(io.near_clean, const_cast<int16_t *>(dev.near_clean))) return false;
    if (!ProcessBlocks(dev, num_blocks)) return false;
    if (dense) {
        if (!AECM_HIP_OK(hipMemcpyAsync(io.out, dev.out, per * sizeof(int16_t), hipMemcpyDeviceToHost, stream_))) return false;
        return Drain();
    }
    tmp.resize(per);
    if (!AECM_HIP_OK(hipMemcpyAsync(tmp.data(), dev.out, per * sizeof(int16_t), hipMemcpyDeviceToHost, stream_))) return false;
    if (!Drain()) return false;
    for (int s = 0; s < num_streams_; ++s)
        for (int b = 0; b < num_blocks; ++b)
            memcpy(io.out + s * io.stream_stride + b * io.block_stride, &tmp[((size_t)s * num_blocks + b) * kBlock],
                   kBlock * sizeof(int16_t));
    return true;
}

bool BatchEngine::ProcessBlocksHostMapped(const IoView &io, int num_blocks) {
    if (!mapped_host_) {                          // both pointers are published together, only when both calls succeeded
        void *dev = nullptr;
        int16_t *host = nullptr;
        if (!AECM_HIP_OK(hipHostMalloc((void **)&host, kMappedBytes, hipHostMallocMapped))) return false;
        if (!AECM_HIP_OK(hipHostGetDevicePointer(&dev, host, 0)) || !dev) {
            (void)hipHostFree(host);
            return false;
        }
        mapped_host_ = host;
        mapped_dev_ = static_cast<int16_t *>(dev);
    }
    const size_t row = (size_t)num_blocks * kBlock, per = (size_t)num_streams_ * row;
    const int n_in = io.near_clean ? 3 : 2;
    auto rows = [&](const int16_t *src, int16_t *dst, bool to_mapped) {     // src / dst: the caller's side, the mapped side
        for (int s = 0; s < num_streams_; ++s)
            for (int b = 0; b < num_blocks; ++b) {
                const int16_t *user = src + s * io.stream_stride + b * io.block_stride;
                int16_t *ours = dst + (size_t)s * row + (size_t)b * kBlock;
                if (to_mapped) memcpy(ours, user, kBlock * sizeof(int16_t));
                else memcpy(const_cast<int16_t *>(user), ours, kBlock * sizeof(int16_t));
            }
    };
    rows(io.far, mapped_host_, true);
    rows(io.near, mapped_host_ + per, true);
    if (io.near_clean) rows(io.near_clean, mapped_host_ + 2 * per, true);
    IoView dev{mapped_dev_, mapped_dev_ + per, io.near_clean ? mapped_dev_ + 2 * per : nullptr, mapped_dev_ + (size_t)n_in * per,
               (int64_t)row, kBlock};
    if (!LaunchBlocks(st_, dev, num_streams_, num_blocks, nullptr)) return false;
    if (!Drain()) return false;
    rows(io.out, mapped_host_ + (size_t)n_in * per, false);
    return true;
}

// Dense host buffers, many streams: the streams are cut into chunks; a helper thread downloads chunk k
// (pageable-memory copies block their host thread) while this thread uploads and launches chunk k + 1,
// so both directions of the link are busy.
bool BatchEngine::ProcessBlocksHostPipelined(const IoView &io, int num_blocks) {
    const size_t row = (size_t)num_blocks * kBlock;                       // samples per stream
    const int n_in = io.near_clean ? 3 : 2;
    const size_t per = (size_t)num_streams_ * row;
    if (!download_stream_ && !AECM_HIP_OK(hipStreamCreateWithFlags(&download_stream_, hipStreamNonBlocking))) return false;
    if (!FlushTimers()) return false;
    const int n_chunks = (num_streams_ + kHostChunkStreams - 1) / kHostChunkStreams;
    std::vector<hipEvent_t> done(n_chunks, nullptr);
    bool ok = true;
    for (auto &e : done) ok = ok && AECM_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int16_t *dfar = stage_dev_, *dnear = stage_dev_ + per, *dclean = stage_dev_ + 2 * per, *dout = stage_dev_ + (size_t)n_in * per;
    std::atomic<int> launched{0};
    std::atomic<bool> failed{!ok};
    std::thread downloader([&] {
        if (!AECM_HIP_OK(hipSetDevice(device_))) { failed = true; return; }
        for (int k = 0; k < n_chunks; ++k) {
            while (launched.load(std::memory_order_acquire) <= k) {
                if (failed.load()) return;
                std::this_thread::yield();
            }
            const size_t first = (size_t)k * kHostChunkStreams;
            const size_t count = std::min<size_t>(kHostChunkStreams, num_streams_ - first);
            if (!AECM_HIP_OK(hipEventSynchronize(done[k])) ||
                !AECM_HIP_OK(hipMemcpyAsync(io.out + first * row, dout + first * row, count * row * sizeof(int16_t),
                                            hipMemcpyDeviceToHost, download_stream_)) ||
                !AECM_HIP_OK(hipStreamSynchronize(download_stream_))) {
                failed = true;
                return;
            }
        }
    });
    for (int k = 0; k < n_chunks && !failed.load(); ++k) {
        const size_t first = (size_t)k * kHostChunkStreams;
        const size_t count = std::min<size_t>(kHostChunkStreams, num_streams_ - first);
        const size_t off = first * row, bytes = count * row * sizeof(int16_t);
        bool up = AECM_HIP_OK(hipMemcpyAsync(dfar + off, io.far + off, bytes, hipMemcpyHostToDevice, stream_)) &&
                  AECM_HIP_OK(hipMemcpyAsync(dnear + off, io.near + off, bytes, hipMemcpyHostToDevice, stream_)) &&
                  (!io.near_clean || AECM_HIP_OK(hipMemcpyAsync(dclean + off, io.near_clean + off, bytes, hipMemcpyHostToDevice, stream_)));
        StatePtrs st = st_;
        st.vec += first * kVecWordsPerStream;
        st.scal += first * kNumScal;
        st.hist += first * kHistWordsPerStream;
        IoView dev{dfar + off, dnear + off, io.near_clean ? dclean + off : nullptr, dout + off, (int64_t)row, kBlock};
        up = up && LaunchBlocks(st, dev, (int)count, num_blocks, nullptr) &&
             AECM_HIP_OK(hipEventRecord(done[k], stream_));
        if (!up) failed = true;
        launched.store(k + 1, std::memory_order_release);
    }
    if (failed.load()) launched.store(n_chunks, std::memory_order_release);      // let the helper fall out of its wait
    downloader.join();
    ok = !failed.load() && Drain();
    for (auto e : done)
        if (e) (void)hipEventDestroy(e);
    return ok;
}

// Grow-only device scratch shared by the recording batches (no hipMalloc / hipFree per call).
bool BatchEngine::EnsureRecordingScratch(size_t map_elems, size_t sample_elems) {
    if (map_elems > rec_maps_elems_) {
        (void)hipFree(rec_maps_);
        rec_maps_ = nullptr;
        rec_maps_elems_ = 0;
        if (!AECM_HIP_OK(hipMalloc((void **)&rec_maps_, map_elems * sizeof(int32_t)))) return false;
        rec_maps_elems_ = map_elems;
    }
    if (sample_elems > rec_scratch_elems_) {
        (void)hipFree(rec_scratch_);
        rec_scratch_ = nullptr;
        rec_scratch_elems_ = 0;
        if (!AECM_HIP_OK(hipMalloc((void **)&rec_scratch_, sample_elems * sizeof(int16_t)))) return false;
        rec_scratch_elems_ = sample_elems;
    }
    return true;
}

bool BatchEngine::ProcessRecordings(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out,
                                    int64_t stream_stride, int frame, int n_calls, int16_t ms, bool host_pointers, int32_t *rc) {
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    if (mixed_rates_) { *rc = kErrUnsupported; return true; }        // an imported stream runs at another rate than fs_
    const RecordingSchedule sch = BuildRecordingSchedule(fs_, frame, n_calls, ms);
    if (sch.first_error) { *rc = sch.first_error; return true; }
    *rc = sch.warned ? kWarnBadParameter : 0;
    const int64_t n_in = (int64_t)n_calls * frame, n_blk = (int64_t)sch.n_blocks * kBlock;
    const int n_near = clean ? 2 : 1;      // the clean near-end follows the near-end's schedule sample for sample
    // Device scratch per stream: gathered far/near(/clean) blocks + block outputs (+ staged I/O for host pointers).
    // The streams are processed in chunks so that the scratch stays below kRecordingScratchBytes whatever the
    // batch size and recording length (and the 1-D grids of the gather kernels stay below 2^31 workgroups).
    const size_t blk_per_stream = (size_t)(2 + n_near) * (size_t)n_blk;
    const size_t io_per_stream = host_pointers ? (size_t)(2 + n_near) * (size_t)n_in : 0;
    const size_t per_stream = std::max<size_t>(blk_per_stream + io_per_stream, 1);
    const int64_t tiles = std::max<int64_t>((std::max(n_in, n_blk) + 255) / 256, 1);
    int64_t chunk = std::max<int64_t>((int64_t)(kRecordingScratchBytes / (per_stream * sizeof(int16_t))), 1);
    chunk = std::min<int64_t>(chunk, 0x7fffffffll / tiles);
    chunk = std::min<int64_t>(chunk, num_streams_);
    const size_t map_elems = std::max<size_t>((size_t)(2 * n_blk + n_in), 1);
    if (!EnsureRecordingScratch(map_elems, (size_t)chunk * per_stream)) return false;
    int32_t *maps = rec_maps_;
    bool ok = true;
    if (n_blk > 0) {
        ok = AECM_HIP_OK(hipMemcpyAsync(maps, sch.far_map.data(), n_blk * sizeof(int32_t), hipMemcpyHostToDevice, stream_)) &&
             AECM_HIP_OK(hipMemcpyAsync(maps + n_blk, sch.near_map.data(), n_blk * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    }
    if (ok) ok = AECM_HIP_OK(hipMemcpyAsync(maps + 2 * n_blk, sch.out_map.data(), n_in * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    for (int64_t s0 = 0; ok && s0 < num_streams_; s0 += chunk) {
        const size_t C = (size_t)std::min<int64_t>(chunk, num_streams_ - s0);
        int16_t *bfar = rec_scratch_, *bnear = bfar + C * n_blk, *bout = bnear + C * n_blk, *bclean = bout + C * n_blk;
        const int16_t *dfar = far + s0 * stream_stride, *dnear = near + s0 * stream_stride;
        const int16_t *dclean = clean ? clean + s0 * stream_stride : nullptr;
        int16_t *dout = out + s0 * stream_stride;
        int64_t dstride = stream_stride;
        if (host_pointers) {
            int16_t *io = rec_scratch_ + C * blk_per_stream;
            dstride = n_in;
            auto upload = [&](const int16_t *src, int16_t *dst) {
                return AECM_HIP_OK(hipMemcpy2DAsync(dst, n_in * 2, src, stream_stride * 2, n_in * 2, C, hipMemcpyHostToDevice, stream_));
            };
            ok = upload(dfar, io) && upload(dnear, io + C * n_in) && (!clean || upload(dclean, io + 3 * C * n_in));
            dfar = io;
            dnear = io + C * n_in;
            dout = io + 2 * C * n_in;
            if (clean) dclean = io + 3 * C * n_in;
        }
        if (ok && n_blk > 0) {
            ok = AECM_HIP_OK(LaunchGatherByMap(dfar, dstride, maps, n_blk, bfar, n_blk, (int)C, stream_)) &&
                 AECM_HIP_OK(LaunchGatherByMap(dnear, dstride, maps + n_blk, n_blk, bnear, n_blk, (int)C, stream_)) &&
                 (!clean || AECM_HIP_OK(LaunchGatherByMap(dclean, dstride, maps + n_blk, n_blk, bclean, n_blk, (int)C, stream_)));
            if (ok) {
                IoView io{bfar, bnear, clean ? bclean : nullptr, bout, n_blk, kBlock};
                ok = ProcessBlocksRange(io, sch.n_blocks, (int)s0, (int)C, nullptr);
            }
        }
        // pass-through samples of the start-up phase come from the clean near-end when there is one
        // (reference echo_control_mobile.cc:285-291)
        if (ok) ok = AECM_HIP_OK(LaunchAssembleOutput(bout, n_blk, clean ? dclean : dnear, dstride, maps + 2 * n_blk, n_in, dout, dstride,
                                                      (int)C, stream_));
        if (ok && host_pointers)
            ok = AECM_HIP_OK(hipMemcpy2DAsync(out + s0 * stream_stride, stream_stride * 2, dout, n_in * 2, n_in * 2, C,
                                              hipMemcpyDeviceToHost, stream_));
    }
    if (!Drain()) ok = false;      // sch (the maps' host copy) goes out of scope
    return ok;
}

bool BatchEngine::Synchronize() {
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    return Drain();
}

bool BatchEngine::LastLaunchMs(float *ms) {
    if (!FlushTimers()) return false;
    *ms = last_ms_;
    return true;
}

bool BatchEngine::Timers(double *total_ms, int64_t *launches) {
    if (!FlushTimers()) return false;
    *total_ms = total_ms_;
    *launches = launches_;
    return true;
}

void BatchEngine::ResetTimers() {
    (void)FlushTimers();
    total_ms_ = 0.0;
    launches_ = 0;
}

bool BatchEngine::SetEchoPath(int stream, const int16_t path[kBins]) {
    if (stream < 0 || stream >= num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    // WebRtcAecm_InitEchoPathCore touches the two channel words of the lane-vector image and seven scalars: write
    // exactly those (two 256-byte rows + one scalar patch), in stream order after whatever is still running.
    std::vector<uint32_t> vec(kVecWordsPerStream, 0u);
    int32_t scal[kNumScal] = {0};
    aecm::SetEchoPath(vec.data(), scal, path);
    uint32_t *dvec = st_.vec + (size_t)stream * kVecWordsPerStream;
    // `vec` is a local: whatever happens below, no copy from it may still be in flight when this function returns
    bool ok = AECM_HIP_OK(hipMemcpyAsync(dvec + V_CH16 * kLanes, vec.data() + V_CH16 * kLanes, kLanes * sizeof(uint32_t), hipMemcpyHostToDevice, stream_)) &&
              AECM_HIP_OK(hipMemcpyAsync(dvec + V_CH32 * kLanes, vec.data() + V_CH32 * kLanes, kLanes * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    if (ok) {
        const int32_t fields[7] = {S_B64_CHSTORED, S_B64_CHADAPT16, S_B64_CHADAPT32, S_MSE_ADAPT_OLD, S_MSE_STORED_OLD, S_MSE_THRESH, S_MSECNT};
        int32_t values[7];
        for (int i = 0; i < 7; ++i) values[i] = scal[fields[i]];
        ok = PatchScalars(fields, values, 7, stream, 1);
    }
    const bool drained = Drain();
    return ok && drained;
}

bool BatchEngine::GetEchoPath(int stream, int16_t path[kBins]) {
    if (stream < 0 || stream >= num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    std::vector<uint32_t> ch(kLanes);
    std::vector<int32_t> scal(kNumScal);
    if (!Drain()) return false;
    if (!AECM_HIP_OK(hipMemcpy(ch.data(), st_.vec + (size_t)stream * kVecWordsPerStream + V_CH16 * kLanes,
                               kLanes * sizeof(uint32_t), hipMemcpyDeviceToHost)))
        return false;
    if (!AECM_HIP_OK(hipMemcpy(scal.data(), st_.scal + (size_t)stream * kNumScal, kNumScal * sizeof(int32_t),
                               hipMemcpyDeviceToHost)))
        return false;
    for (int t = 0; t < kLanes; ++t) path[t] = (int16_t)(ch[t] & 0xffff);
    path[64] = (int16_t)scal[S_B64_CHSTORED];
    return true;
}

// Snapshot = header + vec + scal + hist (aecm_state_check.h: SnapshotHeader).  The header pins the layout the blob was written
// with, so a blob from another build (different field lists) or a corrupted one is refused instead of being used as addresses.
static_assert(kStateHeaderBytes == BatchEngine::kStateHeaderBytes && kStateBlobBytes == BatchEngine::kStateBytes, "snapshot blob size");

bool BatchEngine::ExportState(int stream, void *buf) {
    if (stream < 0 || stream >= num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_)) || !Drain()) return false;
    uint8_t *p = static_cast<uint8_t *>(buf);
    uint8_t *body = p + kStateHeaderBytes;
    if (!(AECM_HIP_OK(hipMemcpy(body, st_.vec + (size_t)stream * kVecWordsPerStream, kVecWordsPerStream * 4, hipMemcpyDeviceToHost)) &&
          AECM_HIP_OK(hipMemcpy(body + kVecWordsPerStream * 4, st_.scal + (size_t)stream * kNumScal, kNumScal * 4, hipMemcpyDeviceToHost)) &&
          AECM_HIP_OK(hipMemcpy(body + kVecWordsPerStream * 4 + kNumScal * 4, st_.hist + (size_t)stream * kHistWordsPerStream,
                                kHistWordsPerStream * 2, hipMemcpyDeviceToHost))))
        return false;
    int32_t mult = 0;
    memcpy(&mult, body + kVecWordsPerStream * 4 + S_MULT * 4, 4);
    const SnapshotHeader h{kSnapshotMagic, kStateLayoutVersion, (uint32_t)mult * 8000u, (uint32_t)kNumVec, (uint32_t)kNumScal,
                           (uint32_t)kHistory, (uint32_t)kLanes, 0u};
    memcpy(p, &h, sizeof h);
    return true;
}

// 0, kErrBadParameter (not a snapshot of this layout, or index-like fields out of range) or kErrUnspecified (HIP).
int32_t BatchEngine::ImportState(int stream, const void *buf) {
    if (stream < 0 || stream >= num_streams_) return kErrBadParameter;
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    SnapshotHeader h;
    memcpy(&h, p, sizeof h);
    if (!SnapshotHeaderOk(h)) return kErrBadParameter;
    const uint8_t *body = p + kStateHeaderBytes;
    int32_t scal[kNumScal];
    memcpy(scal, body + kVecWordsPerStream * 4, sizeof scal);
    // everything the kernel uses as an index or a shift count, and the value ranges its arithmetic shortcuts rely on
    std::vector<uint32_t> vec(kVecWordsPerStream);
    memcpy(vec.data(), body, kVecWordsPerStream * 4);
    const bool sane = ValidateStateImage(vec.data(), scal, (int)h.fs) == nullptr;
    if (!sane) return kErrBadParameter;
    if (!AECM_HIP_OK(hipSetDevice(device_)) || !Drain()) return kErrUnspecified;
    if (!(AECM_HIP_OK(hipMemcpy(st_.vec + (size_t)stream * kVecWordsPerStream, body, kVecWordsPerStream * 4, hipMemcpyHostToDevice)) &&
          AECM_HIP_OK(hipMemcpy(st_.scal + (size_t)stream * kNumScal, body + kVecWordsPerStream * 4, kNumScal * 4, hipMemcpyHostToDevice)) &&
          AECM_HIP_OK(hipMemcpy(st_.hist + (size_t)stream * kHistWordsPerStream, body + kVecWordsPerStream * 4 + kNumScal * 4,
                                kHistWordsPerStream * 2, hipMemcpyHostToDevice))))
        return kErrUnspecified;
    if ((int)h.fs != fs_) mixed_rates_ = true;       // ProcessRecordings schedules every stream for fs_: refuse until the next Init
    return 0;
}

bool BatchEngine::EnsureStateStage(int streams) {
    const size_t need = (size_t)std::min(streams, kStateStageStreams) * kStateBytes;
    if (need > state_stage_bytes_) {
        if (!AECM_HIP_OK(hipStreamSynchronize(stream_))) return false;
        (void)hipFree(state_stage_);
        state_stage_ = nullptr;
        state_stage_bytes_ = 0;
        if (!AECM_HIP_OK(hipMalloc((void **)&state_stage_, need))) return false;
        state_stage_bytes_ = need;
    }
    return state_verdict_ || AECM_HIP_OK(hipMalloc((void **)&state_verdict_, 2 * sizeof(uint32_t)));
}

bool BatchEngine::ExportStates(int first, int count, void *states, bool device) {
    if (first < 0 || count < 0 || first + count > num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    if (count == 0) return Drain();
    if (device) return AECM_HIP_OK(LaunchGatherStates(st_, first, count, states, stream_)) && Drain();
    if (!EnsureStateStage(count)) return false;
    uint8_t *dst = static_cast<uint8_t *>(states);
    for (int s0 = 0; s0 < count; s0 += kStateStageStreams) {                      // the copy of a chunk is ordered before the next gather
        const int n = std::min(kStateStageStreams, count - s0);
        if (!AECM_HIP_OK(LaunchGatherStates(st_, first + s0, n, state_stage_, stream_)) ||
            !AECM_HIP_OK(hipMemcpyAsync(dst + (size_t)s0 * kStateBytes, state_stage_, (size_t)n * kStateBytes, hipMemcpyDeviceToHost, stream_)))
            return false;
    }
    return Drain();
}

int32_t BatchEngine::ImportStates(int first, int count, const void *states, bool device) {
    if (first < 0 || count < 0 || first + count > num_streams_) return kErrBadParameter;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return kErrUnspecified;
    if (count == 0) return 0;
    bool other_rate = false;
    if (device) {
        // the kernels read the blobs as 32-bit words: a pointer that is not a multiple of 4 would fault on the GPU
        if (reinterpret_cast<uintptr_t>(states) & 3) return kErrBadParameter;
        if (!EnsureStateStage(1)) return kErrUnspecified;
        const uint32_t preset[2] = {0xffffffffu, 0u};
        uint32_t verdict[2] = {0, 0};
        if (!AECM_HIP_OK(hipMemcpyAsync(state_verdict_, preset, sizeof preset, hipMemcpyHostToDevice, stream_)) ||
            !AECM_HIP_OK(LaunchValidateStates(states, count, fs_, state_verdict_, stream_)) ||
            !AECM_HIP_OK(hipMemcpyAsync(verdict, state_verdict_, sizeof verdict, hipMemcpyDeviceToHost, stream_)) || !Drain())
            return kErrUnspecified;
        if (verdict[0] != 0xffffffffu) return kErrBadParameter;
        other_rate = verdict[1] != 0;
        if (!AECM_HIP_OK(LaunchScatterStates(st_, first, count, states, stream_)) || !Drain()) return kErrUnspecified;
    } else {
        const uint8_t *src = static_cast<const uint8_t *>(states);
        for (int s = 0; s < count; ++s) {                                        // all or nothing: every blob before any stream
            const uint8_t *blob = src + (size_t)s * kStateBytes;
            SnapshotHeader h;
            memcpy(&h, blob, sizeof h);
            if (!SnapshotHeaderOk(h)) return kErrBadParameter;
            const uint32_t *vec = reinterpret_cast<const uint32_t *>(blob + kStateHeaderBytes);      // blobs are 16-byte multiples: aligned if the buffer is
            const int32_t *scal = reinterpret_cast<const int32_t *>(blob + kStateHeaderBytes + kVecWordsPerStream * 4);
            uint32_t vec_copy[kVecWordsPerStream];
            int32_t scal_copy[kNumScal];
            if (reinterpret_cast<uintptr_t>(blob) & 3) {
                memcpy(vec_copy, vec, sizeof vec_copy);
                memcpy(scal_copy, scal, sizeof scal_copy);
                vec = vec_copy;
                scal = scal_copy;
            }
            if (ValidateStateImage(vec, scal, (int)h.fs) != nullptr) return kErrBadParameter;
            other_rate = other_rate || (int)h.fs != fs_;
        }
        if (!EnsureStateStage(count)) return kErrUnspecified;
        for (int s0 = 0; s0 < count; s0 += kStateStageStreams) {
            const int n = std::min(kStateStageStreams, count - s0);
            if (!AECM_HIP_OK(hipMemcpyAsync(state_stage_, src + (size_t)s0 * kStateBytes, (size_t)n * kStateBytes, hipMemcpyHostToDevice, stream_)) ||
                !AECM_HIP_OK(LaunchScatterStates(st_, first + s0, n, state_stage_, stream_)))
                return kErrUnspecified;
        }
        if (!Drain()) return kErrUnspecified;
    }
    if (other_rate) mixed_rates_ = true;
    return 0;
}

bool BatchEngine::Digest(int stream, uint32_t digest[kDigestWords]) {
    if (stream < 0 || stream >= num_streams_) return false;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return false;
    std::vector<uint32_t> vec(kVecWordsPerStream);
    std::vector<int32_t> scal(kNumScal);
    std::vector<uint16_t> hist(kHistWordsPerStream);
    if (!Drain()) return false;
    if (!AECM_HIP_OK(hipMemcpy(vec.data(), st_.vec + (size_t)stream * kVecWordsPerStream, vec.size() * sizeof(uint32_t),
                               hipMemcpyDeviceToHost)))
        return false;
    if (!AECM_HIP_OK(hipMemcpy(scal.data(), st_.scal + (size_t)stream * kNumScal, scal.size() * sizeof(int32_t),
                               hipMemcpyDeviceToHost)))
        return false;
    if (!AECM_HIP_OK(hipMemcpy(hist.data(), st_.hist + (size_t)stream * kHistWordsPerStream,
                               hist.size() * sizeof(uint16_t), hipMemcpyDeviceToHost)))
        return false;
    ComputeDigest(vec.data(), scal.data(), hist.data(), digest);
    return true;
}

}  // namespace aecm
