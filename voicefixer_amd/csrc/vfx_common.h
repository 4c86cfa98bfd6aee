// Internal helpers shared by the libvfx_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/vfx_hip.h"

extern std::atomic<uint64_t> g_vfx_launches;

#define VFX_LAUNCHED() (g_vfx_launches.fetch_add(1, std::memory_order_relaxed))

static inline int vfx_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VFX_OK : (int)e;
}

// Zero ``bytes`` (a multiple of 4) of device memory on a stream with a KERNEL (vfx_misc.hip), not hipMemsetAsync: see there.
int vfx_zero_u32(void* p, size_t bytes, hipStream_t s);

static inline bool vfx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float vfx_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ float vfx_post(float v, int act, float slope) {
    switch (act) {
        case VFX_POST_LRELU: return vfx_lrelu(v, slope);
        case VFX_POST_ELU: return v > 0.f ? v : expm1f(v);
        case VFX_POST_TANH: return tanhf(v);
        case VFX_POST_SIGMOID: return 1.f / (1.f + expf(-v));
        case VFX_POST_LRELU_SNAKE: { float u = vfx_lrelu(v, slope); return u + sinf(u); }
        default: return v;
    }
}
