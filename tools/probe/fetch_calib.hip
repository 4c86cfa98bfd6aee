// fetch_calib.hip -- calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS WIDTH (tools/pmc_summary.py).
// MI355X_MICROARCH.md (HBM section) calibrates only 16 B/lane loads ("FETCH_SIZE reports exactly 1/2 of the bytes";
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// The convolution kernels of this library read activations with DWORD buffer loads (64 lanes x 4 B = one 256-byte run
// per wave instruction) and store with dword buffer stores, so those are measured here next to the 16-byte forms, on
// buffers far larger than the 256 MB Infinity Cache:
//     hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_calib.hip -o /tmp/fetch_calib
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- /tmp/fetch_calib
//     rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o w -- /tmp/fetch_calib
//     python tools/pmc_summary.py calib out/.../f_counter_collection.csv out/.../w_counter_collection.csv out.json
// Every kernel moves exactly BYTES bytes in the direction its name says (reads are reduced into one float per wave so
// that they cannot be dropped; that store traffic is 1/64 of a percent).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr size_t BYTES = 1ull << 30;   // 1 GiB per kernel, 4x the Infinity Cache

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, 0x7fffffff, 0x00020000);
}

// grid-stride over 64 KB slabs: a wave instruction covers 256 B (b32) or 1 KB (b128) of consecutive addresses
__global__ __launch_bounds__(256) void calib_read_b32(const float* x, float* sink) {
    const __amdgpu_buffer_rsrc_t r = rsrc(x);
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * 65536; base < BYTES; base += (size_t)gridDim.x * 65536) {
        const __amdgpu_buffer_rsrc_t rb = rsrc(reinterpret_cast<const char*>(x) + base);
#pragma unroll 16
        for (int i = 0; i < 64; ++i)
            acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, threadIdx.x * 4, i * 1024, 0));
    }
    (void)r;
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_read_b128(const float* x, float* sink) {
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * 65536; base < BYTES; base += (size_t)gridDim.x * 65536) {
        const __amdgpu_buffer_rsrc_t rb = rsrc(reinterpret_cast<const char*>(x) + base);
#pragma unroll 16
        for (int i = 0; i < 16; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, threadIdx.x * 16, i * 4096, 0);
            acc += __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z) + __uint_as_float(v.w);
        }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_write_b32(float* y) {
    for (size_t base = (size_t)blockIdx.x * 65536; base < BYTES; base += (size_t)gridDim.x * 65536) {
        const __amdgpu_buffer_rsrc_t rb = rsrc(reinterpret_cast<char*>(y) + base);
#pragma unroll 16
        for (int i = 0; i < 64; ++i) __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u + i, rb, threadIdx.x * 4, i * 1024, 0);
    }
}

__global__ __launch_bounds__(256) void calib_write_b128(float* y) {
    for (size_t base = (size_t)blockIdx.x * 65536; base < BYTES; base += (size_t)gridDim.x * 65536) {
        const __amdgpu_buffer_rsrc_t rb = rsrc(reinterpret_cast<char*>(y) + base);
#pragma unroll 16
        for (int i = 0; i < 16; ++i) {
            u32x4 v = {0x3f800000u + i, 1u, 2u, 3u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rb, threadIdx.x * 16, i * 4096, 0);
        }
    }
}

// dword stores as the Winograd kernels issue them: 32 consecutive positions per half-wave, the two half-waves 4 channel
// rows apart (128-byte runs, every 64-byte half-line written by one instruction)
__global__ __launch_bounds__(256) void calib_write_b32_halfwave_rows(float* y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 65536; base < BYTES; base += (size_t)gridDim.x * 65536) {
        const __amdgpu_buffer_rsrc_t rb = rsrc(reinterpret_cast<char*>(y) + base);
        // 64 KB slab = 128 rows of 128 floats; wave w owns rows 32w .. 32w+31; instruction i writes rows (8i' + e, +4)
#pragma unroll 16
        for (int i = 0; i < 64; ++i) {
            const int row = wave * 32 + (i & 15) * 2 + (lane >> 5);
            const int col = (i >> 4) * 32 + (lane & 31);
            __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u + i, rb, (row * 128 + col) * 4, 0, 0);
        }
    }
}

int main() {
    float *x, *y, *sink;
    if (hipMalloc(&x, BYTES) != hipSuccess || hipMalloc(&y, BYTES) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) return 1;
    hipMemset(x, 0, BYTES);
    hipMemset(y, 0, BYTES);
    hipDeviceSynchronize();
    const dim3 grid(2048), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read_b32, grid, block, 0, 0, x, sink);
        hipLaunchKernelGGL(calib_read_b128, grid, block, 0, 0, x, sink);
        hipLaunchKernelGGL(calib_write_b32, grid, block, 0, 0, y);
        hipLaunchKernelGGL(calib_write_b128, grid, block, 0, 0, y);
        hipLaunchKernelGGL(calib_write_b32_halfwave_rows, grid, block, 0, 0, y);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("fetch_calib: 5 kernels x 2 launches, %zu bytes each\n", BYTES);
    return 0;
}
