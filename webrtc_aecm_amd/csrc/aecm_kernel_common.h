// Device-side pieces shared by the translation units that hold kernels (aecm_block_kernels.hip, aecm_kernels.hip).
#ifndef AECM_AMD_KERNEL_COMMON_H_
#define AECM_AMD_KERNEL_COMMON_H_

#include "aecm_kernels.h"
#include "aecm_tables.h"
#include "aecm_wave.h"
#include "wave_gfx950.h"

namespace aecm {

// The LDS tables are an image inside the host-built constants blob: one coalesced copy per workgroup.
// kThreads = the workgroup size, a compile-time constant: the copy loop then needs nothing from the dispatch packet (a
// scalar load whose wait would also hold up every other scalar load in flight at kernel entry).
template <int kThreads>
__device__ __forceinline__ void FillLdsTables(const uint32_t *consts) {
    static_assert(kConstBlobWords % 4 == 0, "the constants blob is copied 16 bytes at a time");
    constexpr int kVecs = kConstBlobWords / 4, kPasses = (kVecs + kThreads - 1) / kThreads;
    const int4 *src = reinterpret_cast<const int4 *>(consts);
    int4 *dst = reinterpret_cast<int4 *>(&g_lds[0]);
    int4 tmp[kPasses];                                   // every load in flight before the first LDS store
    // Indices are clamped instead of predicated (the last vector is then copied by several threads, harmlessly): with
    // predicates the compiler sinks each load next to its store and the passes wait for each other.
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
        const int i = (int)threadIdx.x + k * kThreads;
        tmp[k] = src[i < kVecs ? i : kVecs - 1];
    }
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
        const int i = (int)threadIdx.x + k * kThreads;
        dst[i < kVecs ? i : kVecs - 1] = tmp[k];
    }
    __syncthreads();
}

}  // namespace aecm

#endif  // AECM_AMD_KERNEL_COMMON_H_
