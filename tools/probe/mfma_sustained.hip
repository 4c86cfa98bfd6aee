// Development probe (standalone, no libvfx): does the chip HOLD its fp32 MFMA rate?  A kernel of nothing but v_mfma_f32_32x32x2_f32
// (six independent accumulators per wave like convwg4_kernel, two workgroups of four waves per CU, random operands) is launched back
// to back for a few seconds; printed per 0.25 s window: TFLOP/s and the shader clock measured inside the kernel (s_memtime against
// s_memrealtime).  With `tools/probe/power_sampler.sh` beside it: socket power and sclk from rocm-smi over the same seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained [seconds] [lds_reads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int LDSB>
__global__ __launch_bounds__(256, 2) void burn(const float* __restrict__ x, float* out, int nsteps, unsigned long long* clk) {
    __shared__ float smem[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) smem[i] = x[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[6];
    for (int k = 0; k < 6; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = x[tid], b = x[tid + 256];
    const float* bp = smem + (tid & 63);
    for (int u = 0; u < nsteps; ++u) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float bb = b;
            if (LDSB) bb = bp[((u * 6 + k) & 63) * 64];
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[k], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < 6; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int ldsb = argc > 2 ? atoi(argv[2]) : 0;
    float *x, *out; unsigned long long* clk;
    CK(hipMalloc(&x, 8 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&clk, 64));
    std::vector<float> h(2 << 20);
    srand(3);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(x, h.data(), 8 << 20, hipMemcpyHostToDevice));
    const int nsteps = 4096, wgs = 512 * 4;          // 4 residency rounds of 2 workgroups per CU: ~5 ms per launch
    const double flop = (double)wgs * 4 * nsteps * 6 * 4096.0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("fp32 MFMA only%s, %d workgroups x %d steps x 6 MFMAs per wave per launch\n", ldsb ? " + one LDS B-fragment read per MFMA" : "", wgs, nsteps);
    double t = 0, win = 0, wflop = 0; int wn = 0;
    while (t < seconds) {
        CK(hipEventRecord(e0));
        if (ldsb) hipLaunchKernelGGL(burn<1>, dim3(wgs), dim3(256), 0, 0, x, out, nsteps, clk);
        else hipLaunchKernelGGL(burn<0>, dim3(wgs), dim3(256), 0, 0, x, out, nsteps, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t += ms * 1e-3; win += ms * 1e-3; wflop += flop; ++wn;
        if (win >= 0.25) {
            unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
            printf("t = %5.2f s   %6.1f TFLOP/s   shader clock %.3f GHz (in-kernel)   %d launches\n", t, wflop / win / 1e12, (double)c[0] / c[1] * 0.1, wn);
            fflush(stdout);
            win = 0; wflop = 0; wn = 0;
        }
    }
    return 0;
}
