// Internal helpers shared by the libvfx_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/vfx_hip.h"

extern __attribute__((visibility("hidden"))) std::atomic<uint64_t> g_vfx_launches;

#define VFX_LAUNCHED() (g_vfx_launches.fetch_add(1, std::memory_order_relaxed))

static inline int vfx_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VFX_OK : (int)e;
}

// Zero ``bytes`` (a multiple of 4) of device memory on a stream with a KERNEL (vfx_misc.hip), not hipMemsetAsync: see there.
// (internal: hidden visibility, not part of the C ABI)
__attribute__((visibility("hidden"))) int vfx_zero_u32(void* p, size_t bytes, hipStream_t s);

// Development switches (VFX_WINO4_D1, VFX_KC16, ...): environment variables that change WHICH kernel a launch runs.  They are read
// only by -DVFX_DEV builds (`make dev` -> libvfx_hip_dev.so, selected with VFX_LIB=...); the release library reads no
// environment variable at all, and the switch names do not appear in it (tests/test_api_cpu.py greps the .so).
#ifdef VFX_DEV
#include <stdlib.h>
#define VFX_DEV_ENV(name) getenv(name)
#else
#define VFX_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif

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
