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
              AECM_HIP_OK(hipMalloc((void **)&b->blk_, 3 * S * 4 * kBlock * 2)) &&
              AECM_HIP_OK(hipMalloc((void **)&b->tags_dev_, (256 + 256 + 160) * sizeof(int64_t))) &&
              AECM_HIP_OK(hipMalloc((void **)&b->io_dev_, 3 * S * 160 * 2));
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
    (void)hipFree(blk_);
    (void)hipFree(tags_dev_);
    (void)hipFree(io_dev_);
}

int32_t SessionBatch::Init(int32_t samp_freq) {
    if (samp_freq != 8000 && samp_freq != 16000) return AECM_BAD_PARAMETER_ERROR;
    if (!engine_->Init(samp_freq)) return AECM_UNSPECIFIED_ERROR;
    const size_t bytes = (size_t)engine_->num_streams() * kRing * 2;
    if (!AECM_HIP_OK(hipMemsetAsync(far_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(near_ring_, 0, bytes, engine_->stream())) ||
        !AECM_HIP_OK(hipMemsetAsync(out_ring_, 0, bytes, engine_->stream())))
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

int32_t SessionBatch::Tick(const int16_t *far, const int16_t *near, int16_t *out, int64_t stride, int n, int16_t ms,
                           bool host_pointers) {
    if (far == nullptr || near == nullptr || out == nullptr) return AECM_NULL_POINTER_ERROR;
    if (!flow_.initialized()) return AECM_UNINITIALIZED_ERROR;
    if (n != 80 && n != 160) return AECM_BAD_PARAMETER_ERROR;
    if (!AECM_HIP_OK(hipSetDevice(device_))) return AECM_UNSPECIFIED_ERROR;
    const int S = engine_->num_streams();
    hipStream_t st = engine_->stream();
    const int16_t *dfar = far, *dnear = near;
    int16_t *dout = out;
    int64_t dstride = stride;
    if (host_pointers) {
        dstride = 160;
        int16_t *f = io_dev_, *d = io_dev_ + (size_t)S * 160;
        if (!AECM_HIP_OK(hipMemcpy2DAsync(f, 320, far, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(hipMemcpy2DAsync(d, 320, near, stride * 2, (size_t)n * 2, S, hipMemcpyHostToDevice, st)))
            return AECM_UNSPECIFIED_ERROR;
        dfar = f;
        dnear = d;
        dout = io_dev_ + 2 * (size_t)S * 160;
    }
    // 1. the tick's samples into the rings; their tags are absolute sample counts
    if (!AECM_HIP_OK(LaunchRingAppend(dfar, dstride, n, far_ring_, kRing, far_pos_, S, st)) ||
        !AECM_HIP_OK(LaunchRingAppend(dnear, dstride, n, near_ring_, kRing, near_pos_, S, st)))
        return AECM_UNSPECIFIED_ERROR;
    int64_t far_tags[160], near_tags[160], out_tags[160];
    for (int i = 0; i < n; ++i) { far_tags[i] = far_pos_ + i; near_tags[i] = near_pos_ + i; }
    // 2. the session machinery in the index domain
    int32_t rc = flow_.BufferFarend(far_tags, (size_t)n);
    if (rc != 0) return rc;
    far_pos_ += n;
    int64_t blk_far[256], blk_near[256];
    int n_blocks = 0;
    bool passthrough = false, stale = false;
    rc = flow_.Process(near_tags, nullptr, out_tags, (size_t)n, ms,
                       [&](const int64_t *fb, const int64_t *nb, const int64_t *, int64_t *ob, int nblk) {
                           memcpy(blk_far, fb, sizeof(int64_t) * nblk * kBlock);
                           memcpy(blk_near, nb, sizeof(int64_t) * nblk * kBlock);
                           for (int k = 0; k < nblk * kBlock; ++k) ob[k] = blocks_done_ * kBlock + k;
                           n_blocks = nblk;
                           return true;
                       },
                       &passthrough);
    near_pos_ += n;
    if (rc != 0 && rc != AECM_BAD_PARAMETER_WARNING) return rc;
    if (passthrough)
        for (int i = 0; i < n; ++i) out_tags[i] = -(out_tags[i] + 2);
    // every tag must still be inside its ring
    for (int k = 0; k < n_blocks * kBlock; ++k)
        stale |= (blk_far[k] >= 0 && far_pos_ - blk_far[k] > kRing) || (blk_near[k] >= 0 && near_pos_ - blk_near[k] > kRing);
    const int64_t out_head = (blocks_done_ + n_blocks) * kBlock;
    for (int i = 0; i < n; ++i) {
        const int64_t v = out_tags[i];
        stale |= (v >= 0 && out_head - v > kRing) || (v <= -2 && near_pos_ - (-v - 2) > kRing);
    }
    if (stale) return AECM_UNSPECIFIED_ERROR;
    // 3. device side of the tick
    if (n_blocks > 0) {
        const int64_t nb64 = (int64_t)n_blocks * kBlock;
        int16_t *bfar = blk_, *bnear = blk_ + (size_t)S * 256, *bout = blk_ + 2 * (size_t)S * 256;
        if (!AECM_HIP_OK(hipMemcpyAsync(tags_dev_, blk_far, nb64 * sizeof(int64_t), hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(hipMemcpyAsync(tags_dev_ + 256, blk_near, nb64 * sizeof(int64_t), hipMemcpyHostToDevice, st)) ||
            !AECM_HIP_OK(LaunchRingGather(far_ring_, kRing, tags_dev_, nb64, bfar, nb64, S, st)) ||
            !AECM_HIP_OK(LaunchRingGather(near_ring_, kRing, tags_dev_ + 256, nb64, bnear, nb64, S, st)))
            return AECM_UNSPECIFIED_ERROR;
        IoView io{bfar, bnear, nullptr, bout, nb64, kBlock};
        if (!engine_->ProcessBlocks(io, n_blocks)) return AECM_UNSPECIFIED_ERROR;
        if (!AECM_HIP_OK(LaunchRingAppend(bout, nb64, nb64, out_ring_, kRing, blocks_done_ * kBlock, S, st)))
            return AECM_UNSPECIFIED_ERROR;
        blocks_done_ += n_blocks;
    }
    if (!AECM_HIP_OK(hipMemcpyAsync(tags_dev_ + 512, out_tags, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st)) ||
        !AECM_HIP_OK(LaunchRingAssemble(out_ring_, near_ring_, kRing, tags_dev_ + 512, n, dout, dstride, S, st)))
        return AECM_UNSPECIFIED_ERROR;
    // the tag arrays live on this stack frame: the copies above must have been consumed before returning
    if (host_pointers &&
        !AECM_HIP_OK(hipMemcpy2DAsync(out, stride * 2, dout, 320, (size_t)n * 2, S, hipMemcpyDeviceToHost, st)))
        return AECM_UNSPECIFIED_ERROR;
    if (!AECM_HIP_OK(hipStreamSynchronize(st))) return AECM_UNSPECIFIED_ERROR;
    return rc;
}

}  // namespace aecm
