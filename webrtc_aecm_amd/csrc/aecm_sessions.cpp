#include "aecm_sessions.h"

#include <string.h>

#include "../../include/echo_control_mobile.h"

namespace aecm {

#define AECM_HIP_OK(expr) ((expr) == hipSuccess)

SessionBatch *SessionBatch::Create(int num_streams, int device_id) {
    BatchEngine *e = BatchEngine::Create(num_streams, device_id);
    if (!e) return nullptr;
    SessionBatch *b = new SessionBatch();
    b->engine_.reset(e);
    b->device_ = device_id;
    const size_t S = (size_t)num_streams;
    bool ok = AECM_HIP_OK(hipMalloc((void **)&b->far_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->near_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->out_ring_, S * kRing * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->blk_, 4 * S * 4 * kBlock * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->io_dev_, 4 * S * 160 * 2));
    if (!ok) {
        delete b;
        return nullptr;
    }
    return b;
}

SessionBatch::~SessionBatch() {
    (void)hipSetDevice(device_);
    if (engine_) (void)engine_->Synchronize();
    (void)hipFree(far_ring_);
    (void)hipFree(near_ring_);
    (void)hipFree(out_ring_);
    (void)hipFree(clean_ring_);
    (void)hipFree(blk_);
    (void)hipFree(io_dev_);
}

int32_t SessionBatch::Init(int32_t samp_freq) {
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;
    const size_t bytes = (size_t)engine_->num_streams() * kRing * 2;
    if (!AECM_HIP_OK(hipMemsetAsync(far_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(near_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(out_ring_, 0, bytes, engine_->stream())) ||
        (clean_ring_ && !AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, engine_->stream()))))
        return AECM_UNSPECIFIED_ERROR;
    far_pos_ = near_pos_ = blocks_done_ = 0;
    return flow_.Init(samp_freq);
}

int32_t SessionBatch::SetConfig(int16_t cng_mode, int16_t echo_mode) {
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (cng_mode != AecmFalse && cng_mode != AecmTrue) return AECM_BAD_PARAMETER_ERROR;
    if (echo_mode < 0 || echo_mode > 4) {
        if (!engine_->SetCngMode(cng_mode, 0, -1)) return AECM_UNSPECIFIED_ERROR;
        return AECM_BAD_PARAMETER_ERROR;
    }
    return engine_->SetConfig(cng_mode, echo_mode, 0, -1) ? 0 : AECM_UNSPECIFIED_ERROR;
}

int32_t SessionBatch::Tick(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride, int n,
                           int16_t ms, bool host_pointers) {
    if (far == nullptr || near == nullptr || out == nullptr) return AECM_NULL_POINTER_ERROR;
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (n != 80 && n != 160) return AECM_BAD_PARAMETER_ERROR;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    if (clean && !clean_ring_) {
        const size_t bytes = (size_t)S * kRing * 2;
        if (!AECM_HIP_OK(hipMalloc((void **)&clean_ring_, bytes)) || !AECM_HIP_OK(hipMemsetAsync(clean_ring_, 0, bytes, st)))
            return AECM_UNSPECIFIED_ERROR;
    }
    const int16_t *dfar = far, *dnear = near, *dclean = clean;
    int16_t *dout = out;
    int64_t dstride = stride;
    if (host_pointers) {
        dstride = 160;
        int16_t *f = io_dev_, *d = io_dev_ + (size_t)S * 160, *c = io_dev_ + 3 * (size_t)S * 160;
        if (!AECM_HIP_OK(hipMemcpy2DAsync(f, 320, far, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(hipMemcpy2DAsync(d, 320, near, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            (clean && !AECM_HIP_OK(hipMemcpy2DAsync(c, 320, clean, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st))))
            return AECM_UNSPECIFIED_ERROR;
        dfar = f;
        dnear = d;
        dout = io_dev_ + 2 * (size_t)S * 160;
        if (clean) dclean = c;
    }
    // 1. the session machinery in the index domain; a tag is the absolute sample count of a far / near
    //    sample, the tick's samples are [far_pos_, far_pos_ + n) and [near_pos_, near_pos_ + n)
    int64_t far_tags[kTickMaxSamples], near_tags[kTickMaxSamples], out_tags[kTickMaxSamples];
    for (int i = 0; i < n; ++i) { far_tags[i] = far_pos_ + i; near_tags[i] = near_pos_ + i; }
    int32_t rc = flow_.BufferFarend(far_tags, (size_t)n);
    if (rc != 0) return rc;
    int64_t blk_far[kTickMaxBlockSamples], blk_near[kTickMaxBlockSamples];
    int n_blocks = 0;
    bool passthrough = false, stale = false;
    const int64_t out_base = blocks_done_ * kBlock;
    // the clean near-end is positioned exactly like the noisy one: it shares the near tags
    rc = flow_.Process(near_tags, clean ? near_tags : nullptr, out_tags, (size_t)n, ms,
                       [&](const int64_t *fb, const int64_t *nb, const int64_t *, int64_t *ob, int nblk) {
                           memcpy(blk_far, fb, sizeof(int64_t) * nblk * kBlock);
                           memcpy(blk_near, nb, sizeof(int64_t) * nblk * kBlock);
                           for (int k = 0; k < nblk * kBlock; ++k) ob[k] = out_base + k;
                           n_blocks = nblk;
                           return true;
                       },
                       &passthrough);
    if (rc != 0 && rc != AECM_BAD_PARAMETER_WARNING) {
        // the far samples were consumed by BufferFarend: keep the rings in step with the flow
        TickGatherCodes none;
        if (!AECM_HIP_OK(LaunchTickPrepare(dfar, dnear, dclean, dstride, n, far_ring_, near_ring_, clean_ring_, kRing, far_pos_,
                                           near_pos_, blk_, blk_, blk_, 0, none, S, st)))
            return AECM_UNSPECIFIED_ERROR;
        far_pos_ += n;
        near_pos_ += n;
        return rc;
    }
    if (passthrough)
        for (int i = 0; i < n; ++i) out_tags[i] = -(out_tags[i] + 2);
    // 2. every sample's source as a code the kernels understand; every tag must still be inside its ring
    const int nbs = n_blocks * kBlock;
    const int64_t far_end = far_pos_ + n, near_end = near_pos_ + n, out_end = out_base + nbs;
    auto code = [&](int64_t tag, int64_t first_of_tick, int64_t end, int kind_now, int kind_ring) -> int32_t {
        if (tag < 0) return -1;
        if (end - tag > kRing) stale = true;
        if (tag >= first_of_tick) return (int32_t)((kind_now << 28) | (int32_t)(tag - first_of_tick));
        return (int32_t)((kind_ring << 28) | (int32_t)(tag & (kRing - 1)));
    };
    TickGatherCodes gather;
    for (int k = 0; k < nbs; ++k) {
        gather.far[k] = code(blk_far[k], far_pos_, far_end, kTickFromInput, kTickFromRing);
        gather.near[k] = code(blk_near[k], near_pos_, near_end, kTickFromInput, kTickFromRing);
    }
    TickAssembleCodes assemble;
    for (int i = 0; i < n; ++i) {
        const int64_t v = out_tags[i];
        assemble.out[i] = v >= 0    ? code(v, out_base, out_end, kTickFromInput, kTickFromRing)
                          : v <= -2 ? code(-v - 2, near_pos_, near_end, kTickNearInput, kTickNearRing)
                                    : -1;
    }
    if (stale) return AECM_UNSPECIFIED_ERROR;
    // 3. device side of the tick: prepare -> blocks -> finish
    int16_t *bfar = blk_, *bnear = blk_ + (size_t)S * kTickMaxBlockSamples, *bout = blk_ + 2 * (size_t)S * kTickMaxBlockSamples;
    int16_t *bclean = blk_ + 3 * (size_t)S * kTickMaxBlockSamples;
    if (!AECM_HIP_OK(LaunchTickPrepare(dfar, dnear, dclean, dstride, n, far_ring_, near_ring_, clean_ring_, kRing, far_pos_,
                                       near_pos_, bfar, bnear, bclean, nbs, gather, S, st)))
        return AECM_UNSPECIFIED_ERROR;
    far_pos_ += n;
    near_pos_ += n;
    if (n_blocks > 0) {
        IoView io{bfar, bnear, clean ? bclean : nullptr, bout, nbs, kBlock};
        if (!engine_->ProcessBlocks(io, n_blocks)) return AECM_UNSPECIFIED_ERROR;
        blocks_done_ += n_blocks;
    }
    // pass-through samples come from the clean near-end when there is one (echo_control_mobile.cc:285-291)
    if (!AECM_HIP_OK(LaunchTickFinish(bout, nbs, out_ring_, clean ? clean_ring_ : near_ring_, kRing, out_base,
                                      clean ? dclean : dnear, dstride, dout, n, assemble, S, st)))
        return AECM_UNSPECIFIED_ERROR;
    if (host_pointers &&
        !AECM_HIP_OK(hipMemcpy2DAsync(out, stride * 2, dout, 320, (size_t)n * 2, S, hipMemcpyDeviceToHost, st)))
        return AECM_UNSPECIFIED_ERROR;
    if (!AECM_HIP_OK(hipStreamSynchronize(st))) return AECM_UNSPECIFIED_ERROR;
    return rc;
}

}  // namespace aecm
