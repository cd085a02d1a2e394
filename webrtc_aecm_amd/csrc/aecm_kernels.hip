// gfx950 kernels of the MI355X AECM engine other than the block kernels (aecm_block_kernels.hip): state maintenance,
// the session schedule's gather / scatter, the streaming-session tick kernels, the device self-test.
#define AECM_TABLE_ATTR __device__
#include <initializer_list>

#include "aecm_kernel_common.h"
#include "aecm_state_check.h"

namespace aecm {

#if defined(AECM_CHECKED)
__device__ unsigned long long g_aecm_check_fail[2];
#endif
// Audit counters of the -DAECM_CHECKED build (aecm_ops.h); hipErrorNotSupported in the shipped build.
hipError_t ReadCheckCounters(uint64_t counters[2], bool reset) {
#if defined(AECM_CHECKED)
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    unsigned long long host[2] = {0, 0};
    e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_aecm_check_fail), sizeof host);
    if (e != hipSuccess) return e;
    uint64_t blocks[2] = {0, 0};
    e = ReadBlockKernelCheckCounters(blocks, reset);       // the block kernels' unit keeps its own pair
    if (e != hipSuccess) return e;
    counters[0] = host[0] + blocks[0];
    counters[1] = host[1] + blocks[1];
    if (reset) {
        const unsigned long long zero[2] = {0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_aecm_check_fail), zero, sizeof zero);
    }
    return e;
#else
    (void)counters;
    (void)reset;
    return hipErrorNotSupported;
#endif
}

// ---- state maintenance ---------------------------------------------------------------------------

__global__ void aecm_broadcast_image_kernel(StatePtrs st, const uint32_t *image_vec, const int32_t *image_scal,
                                            int first, int count) {
    const int64_t s = (int64_t)first + blockIdx.x;
    if (blockIdx.x >= (unsigned)count) return;
    uint32_t *vec = st.vec + s * (int64_t)kVecWordsPerStream;
    for (int i = threadIdx.x; i < (int)kVecWordsPerStream; i += blockDim.x) vec[i] = image_vec[i];
    int32_t *scal = st.scal + s * (int64_t)kNumScal;
    for (int i = threadIdx.x; i < kNumScal; i += blockDim.x) scal[i] = image_scal[i];
    uint32_t *hist = reinterpret_cast<uint32_t *>(st.hist + s * (int64_t)kHistWordsPerStream);
    for (int i = threadIdx.x; i < (int)kHistWordsPerStream / 2; i += blockDim.x) hist[i] = 0u;
}

hipError_t LaunchBroadcastImage(const StatePtrs &st, const uint32_t *image_vec, const int32_t *image_scal, int first,
                                int count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_broadcast_image_kernel, dim3(count), dim3(256), 0, stream, st, image_vec, image_scal, first,
                       count);
    return hipGetLastError();
}

__global__ void aecm_patch_scalars_kernel(StatePtrs st, ScalarPatch patch, int first, int count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)count * patch.n) return;
    const int64_t s = first + i / patch.n;
    const int f = (int)(i % patch.n);
    st.scal[s * (int64_t)kNumScal + patch.field[f]] = patch.value[f];
}

hipError_t LaunchPatchScalars(const StatePtrs &st, const ScalarPatch &patch, int first, int count, hipStream_t stream) {
    if (count <= 0 || patch.n <= 0) return hipSuccess;
    const int64_t total = (int64_t)count * patch.n;
    hipLaunchKernelGGL(aecm_patch_scalars_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, st, patch, first, count);
    return hipGetLastError();
}

// ---- bulk state snapshots (aecm_state_check.h: header + vec + scal + hist per stream) ---------------------------------
// One workgroup per stream; a blob is kStateBlobBytes / 4 words: 8 header words, then the three regions as they lie in HBM.
constexpr int kBlobHeaderWords = (int)(kStateHeaderBytes / 4), kBlobVecWords = (int)kVecWordsPerStream, kBlobScalWords = kNumScal,
              kBlobHistWords = (int)(kHistWordsPerStream / 2), kBlobWords = (int)(kStateBlobBytes / 4);
static_assert(kBlobHeaderWords + kBlobVecWords + kBlobScalWords + kBlobHistWords == kBlobWords, "blob layout");

__global__ __launch_bounds__(256) void aecm_gather_states_kernel(StatePtrs st, int first, uint32_t *blobs) {
    const int64_t s = (int64_t)first + blockIdx.x;
    uint32_t *blob = blobs + (int64_t)blockIdx.x * kBlobWords;
    const uint32_t *vec = st.vec + s * (int64_t)kVecWordsPerStream;
    const uint32_t *scal = reinterpret_cast<const uint32_t *>(st.scal + s * (int64_t)kNumScal);
    const uint32_t *hist = reinterpret_cast<const uint32_t *>(st.hist + s * (int64_t)kHistWordsPerStream);
    if (threadIdx.x < kBlobHeaderWords) {
        const uint32_t header[kBlobHeaderWords] = {kSnapshotMagic, kStateLayoutVersion, scal[S_MULT] * 8000u, (uint32_t)kNumVec, (uint32_t)kNumScal,
                                                   (uint32_t)kHistory, (uint32_t)kLanes, 0u};
        blob[threadIdx.x] = header[threadIdx.x];
    }
    for (int i = threadIdx.x; i < kBlobVecWords; i += 256) blob[kBlobHeaderWords + i] = vec[i];
    if (threadIdx.x < kBlobScalWords) blob[kBlobHeaderWords + kBlobVecWords + threadIdx.x] = scal[threadIdx.x];
    for (int i = threadIdx.x; i < kBlobHistWords; i += 256) blob[kBlobHeaderWords + kBlobVecWords + kBlobScalWords + i] = hist[i];
}

// result[0] = index of the first blob that may not be run on (atomicMin; the caller presets 0xffffffff), result[1] |= 1 when a
// blob of another sampling rate than fs_batch is among them.  64 threads per blob: thread t checks scalar field t and lane t.
__global__ __launch_bounds__(64) void aecm_validate_states_kernel(const uint32_t *blobs, int fs_batch, uint32_t *result) {
    const uint32_t *blob = blobs + (int64_t)blockIdx.x * kBlobWords;
    SnapshotHeader h;
    h.magic = blob[0]; h.version = blob[1]; h.fs = blob[2]; h.num_vec = blob[3]; h.num_scal = blob[4]; h.history = blob[5]; h.lanes = blob[6]; h.reserved = blob[7];
    const int t = threadIdx.x;
    const uint32_t *vec = blob + kBlobHeaderWords, *scal = vec + kBlobVecWords;
    int defect = SnapshotHeaderOk(h) ? 0 : 1;
    if (!defect) defect = ScalarFieldDefect(t, (int32_t)scal[t], (int)h.fs);
    if (!defect) defect = LaneWordsDefect(t, vec[V_NEARFILT * kLanes + t], vec[V_NOISE * kLanes + t], vec[V_M01 * kLanes + t]);
    if (__ballot(defect != 0) != 0 && t == 0) atomicMin(result, blockIdx.x);
    if (t == 0 && !defect && (int)h.fs != fs_batch) atomicOr(result + 1, 1u);
}

__global__ __launch_bounds__(256) void aecm_scatter_states_kernel(StatePtrs st, int first, const uint32_t *blobs) {
    const int64_t s = (int64_t)first + blockIdx.x;
    const uint32_t *blob = blobs + (int64_t)blockIdx.x * kBlobWords + kBlobHeaderWords;
    uint32_t *vec = st.vec + s * (int64_t)kVecWordsPerStream;
    uint32_t *scal = reinterpret_cast<uint32_t *>(st.scal + s * (int64_t)kNumScal);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st.hist + s * (int64_t)kHistWordsPerStream);
    for (int i = threadIdx.x; i < kBlobVecWords; i += 256) vec[i] = blob[i];
    if (threadIdx.x < kBlobScalWords) scal[threadIdx.x] = blob[kBlobVecWords + threadIdx.x];
    for (int i = threadIdx.x; i < kBlobHistWords; i += 256) hist[i] = blob[kBlobVecWords + kBlobScalWords + i];
}

hipError_t LaunchGatherStates(const StatePtrs &st, int first, int count, void *blobs_dev, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_gather_states_kernel, dim3(count), dim3(256), 0, stream, st, first, static_cast<uint32_t *>(blobs_dev));
    return hipGetLastError();
}
hipError_t LaunchValidateStates(const void *blobs_dev, int count, int fs_batch, uint32_t *result_dev, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_validate_states_kernel, dim3(count), dim3(64), 0, stream, static_cast<const uint32_t *>(blobs_dev), fs_batch, result_dev);
    return hipGetLastError();
}
hipError_t LaunchScatterStates(const StatePtrs &st, int first, int count, const void *blobs_dev, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_scatter_states_kernel, dim3(count), dim3(256), 0, stream, st, first, static_cast<const uint32_t *>(blobs_dev));
    return hipGetLastError();
}

// ---- session-schedule gather / scatter ---------------------------------------------------------------

// One workgroup per (stream, 256-sample tile); the stream index is folded into blockIdx.x (no 65 535 grid.y limit).
__global__ void aecm_gather_by_map_kernel(const int16_t *src, int64_t src_stride, const int32_t *map, int64_t n,
                                          int16_t *dst, int64_t dst_stride, unsigned tiles) {
    const int64_t s = blockIdx.x / tiles;
    const int64_t j = (int64_t)(blockIdx.x % tiles) * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t m = map[j];
    dst[s * dst_stride + j] = m >= 0 ? src[s * src_stride + m] : (int16_t)0;
}

// The host side cuts batches so that n_streams * tiles fits a 31-bit grid (aecm_engine.cpp).
hipError_t LaunchGatherByMap(const int16_t *src, int64_t src_stride, const int32_t *map_dev, int64_t n, int16_t *dst,
                             int64_t dst_stride, int n_streams, hipStream_t stream) {
    if (n <= 0 || n_streams <= 0) return hipSuccess;
    const int64_t tiles = (n + 255) / 256;
    if (tiles * n_streams > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(aecm_gather_by_map_kernel, dim3((unsigned)(tiles * n_streams)), dim3(256), 0, stream, src,
                       src_stride, map_dev, n, dst, dst_stride, (unsigned)tiles);
    return hipGetLastError();
}

__global__ void aecm_assemble_output_kernel(const int16_t *blocks, int64_t blocks_stride, const int16_t *near,
                                            int64_t near_stride, const int32_t *map, int64_t n, int16_t *out,
                                            int64_t out_stride, unsigned tiles) {
    const int64_t s = blockIdx.x / tiles;
    const int64_t j = (int64_t)(blockIdx.x % tiles) * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t v = map[j];
    int16_t r = 0;
    if (v >= 0) r = blocks[s * blocks_stride + v];
    else if (v <= -2) r = near[s * near_stride + (-(int64_t)v - 2)];
    out[s * out_stride + j] = r;
}

hipError_t LaunchAssembleOutput(const int16_t *blocks, int64_t blocks_stride, const int16_t *near, int64_t near_stride,
                                const int32_t *map_dev, int64_t n, int16_t *out, int64_t out_stride, int n_streams,
                                hipStream_t stream) {
    if (n <= 0 || n_streams <= 0) return hipSuccess;
    const int64_t tiles = (n + 255) / 256;
    if (tiles * n_streams > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(aecm_assemble_output_kernel, dim3((unsigned)(tiles * n_streams)), dim3(256), 0, stream,
                       blocks, blocks_stride, near, near_stride, map_dev, n, out, out_stride, (unsigned)tiles);
    return hipGetLastError();
}

// ---- device-resident session machinery (aecm_flow_plan.h) ---------------------------------------------
template <bool kHasClean, class Append>
struct TickFlowBlockIo {
    using E = BlockEngine<Gfx950Wave<true, true, true>, kHasClean>;
    using Regs = typename E::Regs;
    const int16_t *far_src;          // the far ring itself (direct ticks) or the framed far stream
    const int16_t *nr, *cr;          // near / clean rings of this session
    int16_t *out_row;                // output stream ring
    int far_mask, mask;
    unsigned far_pos, near_pos, out_pos;      // positions of the tick's first block in far_src / the near rings / the output ring
    const Append &append;            // the tick's samples -> rings
    __device__ __forceinline__ int far(const Regs &r, int b) const { return far_src[(far_pos + b * kBlock + r.lane) & far_mask]; }
    __device__ __forceinline__ int near(const Regs &r, int b) const { return nr[(near_pos + b * kBlock + r.lane) & mask]; }
    __device__ __forceinline__ int clean(const Regs &r, int b) const { return cr[(near_pos + b * kBlock + r.lane) & mask]; }
    __device__ __forceinline__ void out(const Regs &r, int b, int v) const { out_row[(out_pos + b * kBlock + r.brev) & mask] = (int16_t)v; }
    // Called by the engine after it has issued its state loads: the tick's samples go into the rings HERE (8-byte stores
    // that, placed ahead of those loads, would make the compiler fetch the scalar half of the state with vector loads: for
    // all it knows they could alias it), then this wave's fetches below must see this wave's stores.
    __device__ __forceinline__ void ready() const {
        append();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
};

// One lane per session: the tick's session machinery for 64 sessions per wavefront.
__global__ __launch_bounds__(256)
void aecm_flow_plan_kernel(TickFlowIo fio, int n, unsigned near_pos, int n_streams) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_streams) return;
    FlowRegs r;
    for (int k = 0; k < kFlowFieldsUsed; ++k) r.v[k] = fio.state[(size_t)k * n_streams + s];
    const int ms = fio.ms_per_session ? (int)fio.ms_per_session[s] : fio.ms;
    const int flags = fio.flags_per_session ? (int)fio.flags_per_session[s] : fio.flags;
    FlowPlan p;
    FlowTick(r, fio.fs, n, ms, flags, near_pos, p);
    for (int k = 0; k < kFlowFieldsUsed; ++k) fio.state[(size_t)k * n_streams + s] = r.v[k];
    int32_t w[kFlowPlanWords];
    FlowPackPlan(p, w);
    int4 *dst = reinterpret_cast<int4 *>(fio.plans + (size_t)s * kFlowPlanWords);
    for (int q = 0; q < kFlowPlanWords / 4; ++q) dst[q] = make_int4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// Sessions per workgroup of the tick kernel.  A workgroup's waves are placed together, n / 4 per SIMD: with the 7 waves
// per SIMD the kernel is built for (below), 8-wave workgroups can only ever fill 6 of the 7 slots, 4-wave workgroups all
// of them -- at the price of one 20 KB table fill (from L2) per 4 sessions instead of per 8.  Measured, 65 536 sessions,
// ms per tick: 8 sessions 0.2356, 4 sessions 0.2251, 2 sessions 0.3153.  (Round 2, at 8 waves per SIMD, 8 per
// workgroup was the better size: profiles/r02_experiments.md.)
#ifndef AECM_TICK_FLOW_WAVES
#define AECM_TICK_FLOW_WAVES 4
#endif
#ifndef AECM_TICK_EARLY_STATE_LOAD
#define AECM_TICK_EARLY_STATE_LOAD 1      // 65 536 sessions: 0.2218 -> 0.2194 ms per tick.  (Timing probes without the two fences of the
                                          // tick -- samples into the rings before the blocks fetch them, block outputs before the
                                          // output frames read them -- moved nothing: 0.2199 / 0.2193 / 0.2199 / 0.2190 ms.)
#endif
#ifndef AECM_TICK_SPLIT_PLAN_WORDS
#define AECM_TICK_SPLIT_PLAN_WORDS 1
#endif
#ifndef AECM_TICK_FLOW_WAVES_PER_EU
#if defined(AECM_CHECKED)
#define AECM_TICK_FLOW_WAVES_PER_EU 4
#else
// 7, not 8: the scalar budget of 8 waves per SIMD (78) costs this kernel hundreds of scalar spills; at 94 it has 52.  With the
// uniform regions left unstructurized (build.py): 8 waves 0.272 ms per tick of 65 536 sessions, 7 waves 0.234, 6 waves 0.237
// (default pipeline: 0.241 at 8 waves, its best).
#define AECM_TICK_FLOW_WAVES_PER_EU 7
#endif
#endif
constexpr int kTickFlowWaves = AECM_TICK_FLOW_WAVES;
template <bool kHasClean>
__global__ __launch_bounds__(64 * kTickFlowWaves) __attribute__((amdgpu_waves_per_eu(AECM_TICK_FLOW_WAVES_PER_EU, 8)))
void aecm_tick_flow_kernel(StatePtrs st, TickIo io, TickFlowIo fio, int n_streams) {
    const int64_t s = (int64_t)blockIdx.x * kTickFlowWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // 1. the session's plan for this tick into scalar registers; requested before the table fill so that the fill hides
    //    the latency
    int32_t w[kFlowPlanWords];
    {
        const int32_t *pw = fio.plans + (s < n_streams ? s : 0) * kFlowPlanWords;
        for (int k = 0; k < kFlowPlanWords; ++k) w[k] = pw[k];
    }
#if AECM_TICK_EARLY_STATE_LOAD
    // the session's state loads are issued here, ahead of the table fill and its barrier: their latency runs next to the fill's
    using EarlyE = BlockEngine<Gfx950Wave<true, true, true>, kHasClean>;
    typename EarlyE::Regs early_r;
    {
        const int64_t sl = s < n_streams ? s : 0;
        EarlyE::init_lane_constants(early_r, st.consts);
        EarlyE::load_state(early_r, st.vec + sl * (int64_t)kVecWordsPerStream, st.scal + sl * (int64_t)kNumScal);
    }
#endif
    FillLdsTables<64 * kTickFlowWaves>(st.consts);
    if (s >= n_streams) return;
    // The 16 words arrive as one s_load_dwordx16 register tuple; left like that, the register allocator spills and reloads
    // the WHOLE tuple (16 v_writelane / v_readlane) around every use in another basic block.  Passing each word through an
    // empty asm makes them 16 independent scalars that are spilled one by one, and only where needed.
#if AECM_TICK_SPLIT_PLAN_WORDS
    for (int k = 0; k < kFlowPlanWords; ++k) asm("" : "+s"(w[k]));      // not volatile: a volatile asm counts as a memory clobber and turns the engine's scalar state loads into vector loads
#endif
    FlowPlan p;
    FlowUnpackPlan(w, p);
    const int mask = (int)io.ring_len - 1;
    const int n = io.n;
    const int16_t *fin = io.far_in + s * io.io_stride, *nin = io.near_in + s * io.io_stride;
    const int16_t *cin = kHasClean ? io.clean_in + s * io.io_stride : nullptr;
    int16_t *fr = io.far_ring + s * io.ring_len, *nr = io.near_ring + s * io.ring_len;
    int16_t *cr = kHasClean ? io.clean_ring + s * io.ring_len : nullptr;
    int16_t *orow = io.out_ring + s * io.ring_len;
    int16_t *ff = fio.far_frames + s * kFlowFarFrameRing, *old = fio.far_old + s * (2 * kFlowFrame);
    int16_t *out = io.out + s * io.io_stride;
    // 2. the tick's samples into the rings: what the jitter buffer accepted of the far end, all of the near end.  Four
    //    samples (8 bytes) per lane where everything is 8-byte aligned -- the caller's rows, and ring positions that are
    //    multiples of 4 (near positions always are: ticks are 80 or 160 samples; far positions unless a saturated jitter
    //    buffer accepted an odd count; rings are a multiple of 4 long, so a group never straddles the wrap) -- else one
    //    sample per lane.
    typedef short Quad __attribute__((ext_vector_type(4)));
    const bool rows_aligned = ((reinterpret_cast<uintptr_t>(io.far_in) | reinterpret_cast<uintptr_t>(io.near_in) |
                                reinterpret_cast<uintptr_t>(io.out) | (kHasClean ? reinterpret_cast<uintptr_t>(io.clean_in) : 0) |
                                (uintptr_t)(io.io_stride * 2)) & 7) == 0;
    const bool far_aligned = ((p.far[0].pos | p.far[0].count | p.far[1].pos | p.far[1].count) & 3) == 0;
    const auto append = [&]() {
        if (rows_aligned && far_aligned) {
            if (lane < n / 4) {
                const int j = 4 * lane;
                for (int c = 0; c < 2; ++c)
                    if (j >= p.far[c].src && j < p.far[c].src + p.far[c].count)
                        *reinterpret_cast<Quad *>(fr + ((p.far[c].pos + (unsigned)(j - p.far[c].src)) & mask)) = *reinterpret_cast<const Quad *>(fin + j);
                *reinterpret_cast<Quad *>(nr + (((unsigned)io.near_pos + j) & mask)) = *reinterpret_cast<const Quad *>(nin + j);
                if (kHasClean) *reinterpret_cast<Quad *>(cr + (((unsigned)io.near_pos + j) & mask)) = *reinterpret_cast<const Quad *>(cin + j);
            }
        } else {
            for (int j = lane; j < n; j += 64) {
                for (int c = 0; c < 2; ++c)
                    if (j >= p.far[c].src && j < p.far[c].src + p.far[c].count) fr[(p.far[c].pos + (unsigned)(j - p.far[c].src)) & mask] = fin[j];
                nr[((unsigned)io.near_pos + j) & mask] = nin[j];
                if (kHasClean) cr[((unsigned)io.near_pos + j) & mask] = cin[j];
            }
        }
    };
    // 3. the far end of the tick's blocks.  Usually (p.direct) it is one run of the far stream and the blocks fetch it from
    //    the far ring itself.  Otherwise (an underrun replay, a jump of the jitter buffer's read pointer, the first tick
    //    after start-up) the frames are laid out in the framed-far ring first.
    //    No fences here: a conditional fence before the engine's state loads makes the compiler fetch the scalar half of
    //    the state with vector loads.  None is needed either: the row a spill fills is not read before the next tick, and
    //    far samples that arrived in this very tick are taken from the input row instead of the ring.
    if (p.spill[0] | p.spill[1]) {           // rare: a replay frame is about to be lapped in the far ring -> its row
        for (int i = 0; i < 2; ++i) {
            if (!p.spill[i]) continue;
            int16_t *row = old + i * kFlowFrame;
            const int16_t a0 = fr[(p.spill_pos[i] + lane) & mask], a1 = fr[(p.spill_pos[i] + 64 + (lane & 15)) & mask];
            row[lane] = a0;
            if (lane < kFlowFrame - 64) row[64 + lane] = a1;
        }
    }
    if (!p.direct && (p.frame[0].active | p.frame[1].active)) {
        // far stream position q, from the ring -- or, if it was appended in this tick (a nearly empty jitter buffer), from
        // the piece of the input row it came from
        auto far_stream = [&](unsigned q) -> int16_t {
            const int r0 = (int)(q - p.far[0].pos), r1 = (int)(q - p.far[1].pos);
            const bool in0 = r0 >= 0 && r0 < p.far[0].count, in1 = r1 >= 0 && r1 < p.far[1].count;
            const int16_t ring = fr[q & mask], fresh = fin[in0 ? r0 : in1 ? p.far[1].src + r1 : 0];
            return (in0 || in1) ? fresh : ring;
        };
        // every load before any store: what direct ticks left pending in the far ring, then the tick's frames
        int16_t left = 0, v0[2] = {0, 0}, v1[2] = {0, 0};            // frame f: samples lane and 64 + lane (lanes 0..15)
        if (lane < p.left_count) left = fr[(p.blk_pos0 + p.left_delta + lane) & mask];
        for (int f = 0; f < 2; ++f) {
            const FlowFrame &q = p.frame[f];
            if (!q.active) continue;
            const int16_t *row = old + q.old_idx * kFlowFrame;
            v0[f] = q.far_from_stream ? far_stream(q.far_pos + lane) : row[lane];
            if (lane < kFlowFrame - 64) v1[f] = q.far_from_stream ? far_stream(q.far_pos + 64 + lane) : row[64 + lane];
        }
        if (lane < p.left_count) ff[(p.blk_pos0 + lane) & (kFlowFarFrameRing - 1)] = left;
        for (int f = 0; f < 2; ++f) {
            const FlowFrame &q = p.frame[f];
            if (!q.active) continue;
            ff[(q.frm_pos + lane) & (kFlowFarFrameRing - 1)] = v0[f];
            if (lane < kFlowFrame - 64) ff[(q.frm_pos + 64 + lane) & (kFlowFarFrameRing - 1)] = v1[f];
        }
    }
    // 4. the blocks (the fence between the stores above and the blocks' fetches is TickFlowBlockIo::ready)
    const int nb = p.n_blocks;
    if (nb > 0) {
        using Io = TickFlowBlockIo<kHasClean, decltype(append)>;
        Io bio{p.direct ? fr : ff, nr, cr, orow, p.direct ? mask : kFlowFarFrameRing - 1, mask,
               p.direct ? p.blk_pos0 + p.far_delta : p.blk_pos0, p.near_base + p.blk_pos0, p.blk_pos0, append};
#if AECM_TICK_EARLY_STATE_LOAD
        Io::E::run_stream_loaded(early_r, st, bio, s, nb);
#else
        Io::E::run_stream_io(st, bio, s, nb);
#endif
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    } else {
        append();                    // a session still in its start-up phase: no blocks, but its rings take the samples
    }
    // 5. the output frames: block outputs (this tick's or, when stuffing, older ones) or the start-up copy of the
    //    (clean) near end (echo_control_mobile.cc:285-291)
    //    Output positions are multiples of 16 (blocks of 64, frames of 80, stuffing by 16): both frames as groups of four
    //    samples, 20 lanes each, when the caller's rows are aligned.
    const int16_t *pass = kHasClean ? cin : nin;
    if (rows_aligned && ((p.frame[0].out_pos | p.frame[1].out_pos) & 3) == 0) {
        const int f = lane >= kFlowFrame / 4 ? 1 : 0, j = 4 * (lane - f * (kFlowFrame / 4));
        if (lane < p.n_frames * (kFlowFrame / 4)) {
            const Quad from_ring = *reinterpret_cast<const Quad *>(orow + ((p.frame[f].out_pos + j) & mask));
            const Quad from_input = *reinterpret_cast<const Quad *>(pass + f * kFlowFrame + j);
            *reinterpret_cast<Quad *>(out + f * kFlowFrame + j) = p.frame[f].active ? from_ring : from_input;
        }
    } else {
        for (int f = 0; f < 2; ++f) {
            if (f >= p.n_frames) continue;
            for (int j = lane; j < kFlowFrame; j += 64)
                out[f * kFlowFrame + j] = p.frame[f].active ? orow[(p.frame[f].out_pos + j) & mask] : pass[f * kFlowFrame + j];
        }
    }
}

int TickWorkgroupWaves() { return kTickFlowWaves; }
int TickWorkgroupsPerCu() {       // by the wave slots the kernel is built for, and by the LDS its tables take
    const int by_waves = 4 * AECM_TICK_FLOW_WAVES_PER_EU / kTickFlowWaves, by_lds = (int)((160 * 1024) / sizeof(LdsTables));
    return by_waves < by_lds ? by_waves : by_lds;
}

hipError_t LaunchTickFlow(const StatePtrs &st, const TickIo &io, const TickFlowIo &fio, int n_streams, hipStream_t stream) {
    if (n_streams <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_flow_plan_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, fio, io.n, (unsigned)io.near_pos, n_streams);
    const dim3 grid((n_streams + kTickFlowWaves - 1) / kTickFlowWaves), block(64 * kTickFlowWaves);
    const size_t lds = sizeof(LdsTables);
    if (io.clean_in) hipLaunchKernelGGL((aecm_tick_flow_kernel<true>), grid, block, lds, stream, st, io, fio, n_streams);
    else hipLaunchKernelGGL((aecm_tick_flow_kernel<false>), grid, block, lds, stream, st, io, fio, n_streams);
    return hipGetLastError();
}

// WebRtcAecm_BufferFarend calls without a Process (far-end bursts), one wavefront per session: the few wrapper fields a
// burst touches (aecm_flow_plan.h: FlowBurstReads) into scalar registers, replay frames the burst could lap in the far
// ring to their rows first, then call by call FlowFarendCall and the accepted samples into the far ring.  Session s makes
// clamp(calls_per_session[s] - call_base, 0, max_calls) calls (everybody max_calls without the array) on the samples
// far_in[s][c * n .. + n), c = 0, 1, ...
__global__ __launch_bounds__(256)
void aecm_buffer_farend_kernel(TickIo io, TickFlowIo fio, const uint8_t *calls_per_session, int call_base, int max_calls, int n_streams) {
    const int64_t s = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (s >= n_streams) return;
    const int lane = threadIdx.x & 63;
    int calls = max_calls;
    if (calls_per_session) calls = FlowMin(FlowMax((int)calls_per_session[s] - call_base, 0), max_calls);
    calls = __builtin_amdgcn_readfirstlane(calls);
    if (calls <= 0) return;
    FlowRegs r;
    FlowBurstReads([&](int f) { r.v[f] = __builtin_amdgcn_readfirstlane(fio.state[(size_t)f * n_streams + s]); });
    const int mask = (int)io.ring_len - 1, n = io.n, mult = fio.fs == 16000 ? 2 : 1;
    int16_t *fr = io.far_ring + s * io.ring_len, *old = fio.far_old + s * (2 * kFlowFrame);
    const int16_t *fin = io.far_in + s * io.io_stride;
    FlowBurst b;
    FlowBurstBegin(r, n, calls, b);
    if (b.spill[0] | b.spill[1]) {
        for (int i = 0; i < 2; ++i) {
            if (!b.spill[i]) continue;
            int16_t *row = old + i * kFlowFrame;
            const int16_t a0 = fr[(b.spill_pos[i] + lane) & mask], a1 = fr[(b.spill_pos[i] + 64 + (lane & 15)) & mask];
            row[lane] = a0;
            if (lane < kFlowFrame - 64) row[64 + lane] = a1;
        }
        // the rows have their samples (the stores above needed the loaded values) before the burst may overwrite them in the ring
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    for (int c = 0; c < calls; ++c) {
        const unsigned pos = (unsigned)r.v[F_FAR_WP];
        const int accepted = FlowFarendCall(r, mult, n);
        for (int j = lane; j < accepted; j += 64) fr[(pos + (unsigned)j) & mask] = fin[c * n + j];
    }
    if (lane == 0) FlowBurstWrites([&](int f) { fio.state[(size_t)f * n_streams + s] = r.v[f]; });
}

hipError_t LaunchBufferFarend(const TickIo &io, const TickFlowIo &fio, const uint8_t *calls_per_session, int call_base, int max_calls, int n_streams,
                              hipStream_t stream) {
    if (n_streams <= 0 || max_calls <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_buffer_farend_kernel, dim3((n_streams + 3) / 4), dim3(256), 0, stream, io, fio, calls_per_session, call_base, max_calls, n_streams);
    return hipGetLastError();
}

__global__ __launch_bounds__(256)
void aecm_reset_sessions_kernel(TickIo io, TickFlowIo fio, int n_streams, int first) {
    const int64_t s = (int64_t)first + blockIdx.x;
    auto zero = [&](int16_t *row, int64_t n_samples) {                       // rows are 4-byte aligned and even-sized
        uint32_t *w = reinterpret_cast<uint32_t *>(row);
        for (int64_t i = threadIdx.x; i < n_samples / 2; i += 256) w[i] = 0u;
    };
    zero(io.far_ring + s * io.ring_len, io.ring_len);
    zero(io.out_ring + s * io.ring_len, io.ring_len);
    zero(fio.far_frames + s * kFlowFarFrameRing, kFlowFarFrameRing);
    zero(fio.far_old + s * (2 * kFlowFrame), 2 * kFlowFrame);
    if (threadIdx.x < kFlowFieldsUsed) fio.state[(size_t)threadIdx.x * n_streams + s] = FlowFieldStartsAtOne((int)threadIdx.x) ? 1 : 0;
}

hipError_t LaunchResetSessions(const TickIo &io, const TickFlowIo &fio, int n_streams, int first, int count, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(aecm_reset_sessions_kernel, dim3(count), dim3(256), 0, stream, io, fio, n_streams, first);
    return hipGetLastError();
}

// ---- self test of the wave primitives ------------------------------------------------------------

__device__ __forceinline__ unsigned Mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int Q>
__device__ __forceinline__ void TestExchange(int va, int vb, uint64_t *fails) {
    int a0 = va, b0 = vb, a1 = va, b1 = vb;
    Gfx950Wave<false>::exchange<Q>(a0, b0);
    Gfx950Wave<true>::exchange<Q>(a1, b1);
    // definition: lanes with bit Q clear keep a, b <- partner's a; lanes with bit Q set keep b, a <- partner's b
    const int lane = threadIdx.x & 63;
    const int pa = __shfl(va, lane ^ (1 << Q)), pb = __shfl(vb, lane ^ (1 << Q));
    const int ea = ((lane >> Q) & 1) ? pb : va, eb = ((lane >> Q) & 1) ? vb : pa;
    if (a0 != ea || b0 != eb) atomicAdd((unsigned long long *)&fails[1], 1ull);
    if (a1 != ea || b1 != eb) atomicAdd((unsigned long long *)&fails[1], 1ull);
}

// exchange_all (the N-transform form, DPP-select assembly for the quad stages) against N applications of the definition
template <int Q, int N>
__device__ __forceinline__ void TestExchangeAll(int va, int vb, uint64_t *fails) {
    const int lane = threadIdx.x & 63;
    int aa[N], bb[N], ea[N], eb[N];
    for (int n = 0; n < N; ++n) {
        aa[n] = (int)Mix((unsigned)va + 0x9e37u * n);
        bb[n] = (int)Mix((unsigned)vb ^ (0x85ebu * (n + 1)));
        const int pa = __shfl(aa[n], lane ^ (1 << Q)), pb = __shfl(bb[n], lane ^ (1 << Q));
        ea[n] = ((lane >> Q) & 1) ? pb : aa[n];
        eb[n] = ((lane >> Q) & 1) ? bb[n] : pa;
    }
    Gfx950Wave<true>::exchange_all<Q, N>(aa, bb);
    for (int n = 0; n < N; ++n)
        if (aa[n] != ea[n] || bb[n] != eb[n]) atomicAdd((unsigned long long *)&fails[1], 1ull);
}

__global__ __launch_bounds__(256) void aecm_selftest_kernel(uint64_t *fails, int exhaustive, const uint32_t *consts) {
    FillLdsTables<256>(consts);
    using S = Gfx950Wave<false>;
    using F = Gfx950Wave<true>;
    const int lane = threadIdx.x & 63;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    auto bump = [&](int k) { atomicAdd((unsigned long long *)&fails[k], 1ull); };

    if (blockIdx.x < 64) {
        for (int round = 0; round < 16; ++round) {
            const int v = (int)Mix(gid * 977u + round * 131071u + 1u);
            const int w = (int)Mix(gid * 31u + round * 8191u + 7u);
            // 0: xor shuffles against __shfl with an explicit source lane
            if (F::shfl_xor<1>(v) != __shfl(v, lane ^ 1)) bump(0);
            if (F::shfl_xor<2>(v) != __shfl(v, lane ^ 2)) bump(0);
            if (F::shfl_xor<4>(v) != __shfl(v, lane ^ 4)) bump(0);
            if (F::shfl_xor<8>(v) != __shfl(v, lane ^ 8)) bump(0);
            if (F::shfl_xor<16>(v) != __shfl(v, lane ^ 16)) bump(0);
            if (F::shfl_xor<32>(v) != __shfl(v, lane ^ 32)) bump(0);
            // 1: FFT operand exchange on every lane bit
            TestExchange<0>(v, w, fails); TestExchange<1>(v, w, fails); TestExchange<2>(v, w, fails);
            TestExchange<3>(v, w, fails); TestExchange<4>(v, w, fails); TestExchange<5>(v, w, fails);
            TestExchangeAll<0, 1>(v, w, fails); TestExchangeAll<0, 2>(v, w, fails); TestExchangeAll<0, 3>(v, w, fails);
            TestExchangeAll<1, 1>(v, w, fails); TestExchangeAll<1, 2>(v, w, fails); TestExchangeAll<1, 3>(v, w, fails);
            TestExchangeAll<2, 2>(v, w, fails); TestExchangeAll<5, 3>(v, w, fails);
            // 2: reductions against a serial readlane loop
            int mx = (int)0x80000000, mn = 0x7fffffff, sm = 0;
            for (int i = 0; i < 64; ++i) {
                const int x = __shfl(v, i);
                mx = x > mx ? x : mx; mn = x < mn ? x : mn; sm = add(sm, x);
            }
            if (F::reduce_max(v) != mx || S::reduce_max(v) != mx) bump(2);
            if (F::reduce_min(v) != mn || S::reduce_min(v) != mn) bump(2);
            if (F::reduce_add(v) != sm || S::reduce_add(v) != sm) bump(2);
            {   // merged reductions
                int a0, a1, a2, a3, mn2, mx2;
                int sw = 0, mxw = (int)0x80000000, mnv = 0x7fffffff, mxq = (int)0x80000000;
                const int q = (int)(Mix((unsigned)w) >> 1);            // >= 0 (reduce_min_max negates)
                const int z = (int)Mix((unsigned)v ^ 0x51ed27u), u4 = (int)Mix((unsigned)w + 77u);
                int sz = 0, su = 0;
                for (int i = 0; i < 64; ++i) {
                    const int xw = __shfl(w, i), xz = __shfl(z, i), xu = __shfl(u4, i), xq = __shfl(q, i), xv = __shfl(v, i);
                    sw = add(sw, xw); sz = add(sz, xz); su = add(su, xu);
                    mxw = xw > mxw ? xw : mxw; mnv = xv < mnv ? xv : mnv; mxq = xq > mxq ? xq : mxq;
                }
                F::reduce_max2(v, w, a0, a1);
                if (a0 != mx || a1 != mxw) bump(2);
                S::reduce_max2(v, w, a0, a1);
                if (a0 != mx || a1 != mxw) bump(2);
                F::reduce_min_max(v, q, mn2, mx2);
                if (mn2 != mnv || mx2 != mxq) bump(2);
                S::reduce_min_max(v, q, mn2, mx2);
                if (mn2 != mnv || mx2 != mxq) bump(2);
                F::reduce_add4(v, w, z, u4, a0, a1, a2, a3);
                if (a0 != sm || a1 != sw || a2 != sz || a3 != su) bump(2);
                S::reduce_add4(v, w, z, u4, a0, a1, a2, a3);
                if (a0 != sm || a1 != sw || a2 != sz || a3 != su) bump(2);
            }
            // 3: whole-wave shift by one lane with fill
            const int up = __shfl(v, lane == 0 ? 0 : lane - 1);
            const int expect_up = lane == 0 ? 12345 + round : up;
            if (F::shift_up1(v, 12345 + round) != expect_up || S::shift_up1(v, 12345 + round) != expect_up) bump(3);
            // 4: bpermute / readlane / writelane
            const int src = (int)(Mix(gid + round) & 63u);
            if (F::bpermute(v, src) != __shfl(v, src)) bump(4);
            const int sel_lane = (round * 7 + (int)blockIdx.x) & 63;
            if (F::readlane(v, sel_lane) != __shfl(v, sel_lane)) bump(4);
            const int wl = F::writelane(v, 424242, sel_lane);
            if (wl != (lane == sel_lane ? 424242 : v)) bump(4);
            // 4 (cont.): packed-int16 primitives against their portable definitions
            {
                const int x = v, y = w, c = (int)Mix(gid ^ 0x9e3779b9u);
                const int ed = add(add(mul(sext16(x), sext16(y)), mul(sar(x, 16), sar(y, 16))), c);
                if (dot2_i16(x, y, c) != ed) bump(4);
                if (pack_hi16(x, y) != (int)(((unsigned)x >> 16) | ((unsigned)y & 0xffff0000u))) bump(4);
                int lo = sext16(x), hi = sar(x, 16);
                lo = lo < 0 ? (lo == -32768 ? 32767 : -lo) : lo;
                hi = hi < 0 ? (hi == -32768 ? 32767 : -hi) : hi;
                if (pk_abs_sat_i16(x) != ((lo & 0xffff) | (int)((unsigned)hi << 16))) bump(4);
                if (pk_abs_sat_i16((int)0x80008000) != 0x7fff7fff) bump(4);
                // v_mad_i32_i16 with and without op_sel, vector and scalar addend
                const int uc = (int)Mix((unsigned)round * 2654435761u + blockIdx.x);   // wave-uniform
                if (mad16_lo(x, y, c) != add(mul(sext16(x), sext16(y)), c)) bump(4);
                if (mad16_hi(x, y, c) != add(mul(sar(x, 16), sext16(y)), c)) bump(4);
                if (mad16_lo_uc(x, y, uc) != add(mul(sext16(x), sext16(y)), uc)) bump(4);
                if (mad16_hi_uc(x, y, uc) != add(mul(sar(x, 16), sext16(y)), uc)) bump(4);
                if (mad16_lo((int)0x80008000, -32768, 1) != 0x40000001 || mad16_hi((int)0x7fff0000, -32768, -32770) != (int)0xbffffffe) bump(4);
                {   // per-half saturating add (v_pk_add_i16 clamp): random words and words that saturate either way
                    const int xs[3] = {x, (int)0x7fff8000, (int)0x80017ffe}, ys[3] = {y, (int)0x0001ffff, (int)0xfffe0003};
                    for (int k = 0; k < 3; ++k) {
                        int lo = sext16(xs[k]) + sext16(ys[k]), hi = sar(xs[k], 16) + sar(ys[k], 16);
                        lo = lo > 32767 ? 32767 : lo < -32768 ? -32768 : lo;
                        hi = hi > 32767 ? 32767 : hi < -32768 ? -32768 : hi;
                        if (pk_add_sat_i16(xs[k], ys[k]) != ((lo & 0xffff) | (int)((unsigned)hi << 16))) bump(4);
                    }
                }
                const int ml = imax(sext16(x), sext16(y)), mh = imax(sar(x, 16), sar(y, 16));
                if (pk_max_i16(x, y) != ((ml & 0xffff) | (int)((unsigned)mh << 16))) bump(4);
                // per half "non-zero" (v_pk_min_u16 with an inline constant, as assembly); random words and words with an empty half
                const int xz[4] = {x, x & 0xffff, (int)((unsigned)x & 0xffff0000u), 0};
                for (int k = 0; k < 4; ++k)
                    if (pk_nonzero_u16(xz[k]) != ((((unsigned)xz[k] & 0xffffu) != 0 ? 1 : 0) | (((unsigned)xz[k] >> 16) != 0 ? 0x10000 : 0))) bump(4);
            }
            // 5: ballot bit order
            const bool p = (v >> 3) & 1;
            const uint64_t bal = F::ballot(p);
            if (((bal >> lane) & 1ull) != (p ? 1ull : 0ull)) bump(5);
            if (__builtin_popcountll(bal) != F::reduce_add(p ? 1 : 0)) bump(5);
        }
        // 7: LDS tables
        if (threadIdx.x < 64) {
            {   // last stage: lane t uses entry bitrev6(t); stage 0: entry 0 for every lane
                int wre, wim, brev = 0;
                for (int b = 0; b < 6; ++b) brev |= ((lane >> b) & 1) << (5 - b);
                F::template twiddles<6, false>(wre, wim);
                if (sext16(wre) != kAecmTwiddleCosQ15[brev] || sar(wre, 16) != kAecmTwiddleSinQ15[brev]) bump(7);
                if (sext16(wim) != -kAecmTwiddleSinQ15[brev] || sar(wim, 16) != kAecmTwiddleCosQ15[brev]) bump(7);
                F::template twiddles<6, true>(wre, wim);
                if (sext16(wre) != kAecmTwiddleCosQ15[brev] || sar(wre, 16) != -kAecmTwiddleSinQ15[brev]) bump(7);
                F::template twiddles<0, true>(wre, wim);
                if (wre != 32767 || wim != (int)(32767u << 16)) bump(7);
            }
            if (F::hann(lane + 1) != kAecmSqrtHanningQ14[lane + 1]) bump(7);
            for (int i = lane; i < 360; i += 64)
                if (F::cos360(i) != kAecmCosQ13[i] || F::sin360(i) != kAecmSinQ13[i]) bump(7);
        }
    }
    // 6: floor(sqrt) -- exhaustive over [0, 2^31) in grid-stride, or a 2^24-point sample
    const uint64_t total = exhaustive ? (1ull << 31) : (1ull << 24);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < total; i += stride) {
        const unsigned x = exhaustive ? (unsigned)i : (Mix((unsigned)i) & 0x7fffffffu);
        const uint64_t r = (unsigned)F::isqrt31((int)x);
        if (r * r > x || (r + 1) * (r + 1) <= x) bump(6);
    }
    // 6 (cont.): floor(n / d) of the Wiener gain for EVERY divisor 1..65535: dividends around multiples of d (where an
    // off-by-one would show), the extremes, and random ones -- 64 (quick) or 1 024 (exhaustive) dividends per divisor and thread slot
    {
        const int per = exhaustive ? 1024 : 64;
        for (uint64_t i = gid; i < 65535ull * (unsigned)per; i += stride) {
            const unsigned d = 1u + (unsigned)(i % 65535ull), j = (unsigned)(i / 65535ull);
            const unsigned h = Mix((unsigned)i * 2654435761u + 12345u);
            unsigned n;
            switch (j & 7u) {
                case 0: n = h; break;
                case 1: n = (h / d) * d; break;                       // an exact multiple
                case 2: n = (h / d) * d - 1u; break;                  // one below a multiple (wraps to 2^32 - 1 for h < d)
                case 3: n = (h / d) * d + d - 1u; break;
                case 4: n = 0xffffffffu - (h & 0xffffu); break;       // the top of the range
                case 5: n = h & 0xffffu; break;                       // small dividends
                case 6: n = (0xffffffffu / d) * d - (j >> 3 & 1u); break;
                default: n = h >> (h & 31u); break;
            }
            if ((unsigned)F::divu_u32_u16((int)n, (int)d) != n / d) bump(6);
        }
    }
    // edge values
    if (gid == 0) {
        const unsigned edges[7] = {0u, 1u, 0x7fffffffu, 0x7ffea810u /*46340^2*/, 0x7ffea80fu, 0x40000000u, 0x80000000u};
        for (int k = 0; k < 7; ++k) {
            const uint64_t r = (unsigned)F::isqrt31((int)edges[k]);
            if (r * r > edges[k] || (r + 1) * (r + 1) <= edges[k]) bump(6);
        }
    }
}

template <bool kFast>
__global__ __launch_bounds__(256) void aecm_fft128_kernel(int16_t *data, int32_t *scales, int variant, int count,
                                                          const uint32_t *consts) {
    FillLdsTables<256>(consts);
    using E = BlockEngine<Gfx950Wave<kFast>, false>;
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= count) return;
    int16_t *re = data + (size_t)k * 256, *im = re + 128;
    int a = (re[lane] & 0xffff) | (int)((unsigned)(variant == 0 ? 0 : im[lane]) << 16);
    int b = (re[lane + 64] & 0xffff) | (int)((unsigned)(variant == 0 ? 0 : im[lane + 64]) << 16);
    int scale = 0;
    const int kp = Gfx950Wave<kFast>::opaque_const(32770);
    if (variant == 0) scale = E::template fft128<false, true>(a, b, kp);
    else if (variant == 1) scale = E::template fft128<false, false>(a, b, kp);
    else {
        scale = E::template fft128<true, false>(a, b, kp);
        a >>= 16;                                   // the inverse transform hands its real parts on in the upper halves
        b >>= 16;
    }
    int r = 0;
    for (int i = 0; i < 6; ++i) r |= ((lane >> i) & 1) << (5 - i);
    re[r] = (int16_t)a;
    re[r + 64] = (int16_t)b;
    im[r] = variant == 2 ? (int16_t)0 : (int16_t)(a >> 16);
    im[r + 64] = 0;
    if (lane == 0) scales[k] = scale;
}
hipError_t LaunchFft128(int16_t *data_dev, int32_t *scales_dev, int variant, int fast, int count, const uint32_t *consts_dev,
                        hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const dim3 grid((count + 3) / 4), block(256);
    if (fast) hipLaunchKernelGGL(aecm_fft128_kernel<true>, grid, block, sizeof(LdsTables), stream, data_dev, scales_dev, variant, count, consts_dev);
    else hipLaunchKernelGGL(aecm_fft128_kernel<false>, grid, block, sizeof(LdsTables), stream, data_dev, scales_dev, variant, count, consts_dev);
    return hipGetLastError();
}

hipError_t LaunchSelfTest(uint64_t *counters_dev, int exhaustive, const uint32_t *consts_dev, hipStream_t stream) {
    hipLaunchKernelGGL(aecm_selftest_kernel, dim3(exhaustive ? 4096 : 256), dim3(256), sizeof(LdsTables), stream,
                       counters_dev, exhaustive, consts_dev);
    return hipGetLastError();
}

}  // namespace aecm
