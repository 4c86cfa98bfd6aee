// Development probe (standalone, no libvfx): what does each ingredient of the tap-GEMM main loop cost the fp32 MFMA pipe?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Every variant runs the same number of v_mfma_f32_32x32x2_f32 per wave (4 independent accumulators, like a 64x64
// wave tile); reported: TFLOP/s and the shader clock measured inside the kernel (s_memtime vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// MODE bit 0: B fragments from LDS (2 x ds_read_b32 per k-step), bit 1: A vectors from global (2 x 16 B per 4 k-steps),
// bit 2: __syncthreads every 12 steps, bit 3: activation staging traffic (4 x 16-byte global loads + LDS writes per 12 steps)
// MODE bit 4: the staging loads STREAM (every workgroup walks its own 16 KB-per-chunk region of a multi-GB buffer: HBM
// traffic like a real convolution) and every 96 steps the accumulators are stored (64 KB per workgroup, "epilogue")
template <int MODE>
__global__ __launch_bounds__(256, 3) void probe(const float* __restrict__ w, const float* __restrict__ x, float* out,
                                                int nsteps, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) smem[i] = x[(blockIdx.x * 8192 + i) & 0xfffff] + (float)(i & 15) * 0.01f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, 0x7fffffff, 0x00020000);
    const int voff = (lo * 8 + 4 * hi) * 4;
    float4 a0 = make_float4(1.f, 2.f, 3.f, 4.f), a1 = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    float4 n0 = a0, n1 = a1;
    float4 xv[4];
    const float* rowp = smem + hi * 132 + (wave & 1) * 64 + lo;
    const int xbase = ((blockIdx.x * 256 + tid) * 4) * 4;
    for (int u = 0; u < nsteps; ++u) {
        float4 c0 = a0, c1 = a1;
        if (MODE & 2) {
            a0 = n0; a1 = n1;
            const int soff = (u & 63) * 4096;
            u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wr, voff, soff, 0);
            n0 = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            r = __builtin_amdgcn_raw_buffer_load_b128(wr, voff + 1024, soff, 0);
            n1 = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
        if ((MODE & 8) && (u % 12) == 0) {
            if (u) for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(smem + 4096 + 4 * (tid + 256 * j)) = xv[j];
            for (int j = 0; j < 4; ++j) {
                u32x4 r;
                if (MODE & 16) {
                    // chunk c of workgroup g: 16 KB at ((g * 97 + c) % 65536) * 16 KB  (1 GB window, no reuse within a launch)
                    const unsigned cidx = ((unsigned)blockIdx.x * 97u + (unsigned)(u / 12)) & 65535u;
                    r = __builtin_amdgcn_raw_buffer_load_b128(xr, (tid + 256 * j) * 16, (int)(cidx * 16384u), 0);
                } else {
                    r = __builtin_amdgcn_raw_buffer_load_b128(xr, xbase + j * 16 * 256 * 1024, (u / 12 & 31) * 4096 * 4, 0);
                }
                xv[j] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
        }
        const float* xt = rowp + (u & 7) * 264;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float b0, b1;
            if (MODE & 1) { b0 = xt[kk * 264]; b1 = xt[kk * 264 + 32]; }
            else { b0 = c0.x + (float)kk; b1 = c1.y; }
            const float av0 = kk == 0 ? c0.x : (kk == 1 ? c0.y : (kk == 2 ? c0.z : c0.w));
            const float av1 = kk == 0 ? c1.x : (kk == 1 ? c1.y : (kk == 2 ? c1.z : c1.w));
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b1, acc[1][1], 0, 0, 0);
        }
        if ((MODE & 4) && (u % 12) == 11) __syncthreads();
        if ((MODE & 16) && (u % 96) == 95) {
            float* o = out + (((size_t)blockIdx.x * 13 + u / 96) & 8191) * 16384 + wave * 4096 + lane;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) o[((i * 2 + j) * 16 + r) * 64] = acc[i][j][r];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (MODE & 8) s += xv[0].x + xv[1].y + xv[2].z + xv[3].w;
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 7 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int wgs, int nsteps, const float* w, const float* x, float* out, unsigned long long* clk, size_t lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), lds, 0, w, x, out, nsteps, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), lds, 0, w, x, out, nsteps, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flop = (double)wgs * 4 * nsteps * 16 * (2.0 * 32 * 32 * 2);
    printf("%-44s wgs %5d lds %6zu  %8.3f ms  %7.2f TFLOP/s   shader clock %.3f GHz (in-kernel)\n", name, wgs, lds, ms,
           flop / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
}

// Variant loop: step loop unrolled by 3 with three A register sets (A vectors loaded TWO steps ahead, no register
// rotation moves, counted vmcnt) and B fragments prefetched BD k-steps ahead.  Same work as probe<15/31>.
template <int MODE, int BD>
__global__ __launch_bounds__(256, 3) void probe2(const float* __restrict__ w, const float* __restrict__ x, float* out,
                                                 int nsteps, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) smem[i] = x[(blockIdx.x * 8192 + i) & 0xfffff] + (float)(i & 15) * 0.01f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0, 0x7fffffff, 0x00020000);
    const int voff = (lo * 8 + 4 * hi) * 4;
    float4 A[3][2];
    float4 xv[4];
    const float* rowp = smem + hi * 132 + (wave & 1) * 64 + lo;
    auto loadA = [&](float4 (&d)[2], int u) {
        const int soff = (u & 63) * 4096;
        u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wr, voff, soff, 0);
        d[0] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        r = __builtin_amdgcn_raw_buffer_load_b128(wr, voff + 1024, soff, 0);
        d[1] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
    };
    loadA(A[0], 0);
    loadA(A[1], 1);
    float bq[3][2];   // B fragments of k-steps q, q+1, (q+2): ring of 3, static indices thanks to the x3 unroll (12 k-steps)
    auto readB = [&](float (&d)[2], int u, int kk) {
        const float* xt = rowp + (u & 7) * 264 + kk * 264;
        d[0] = xt[0];
        d[1] = xt[32];
    };
    readB(bq[0], 0, 0);
    if (BD == 2) readB(bq[1], 0, 1);
    for (int u = 0; u < nsteps; u += 3) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int uu = u + v;
            loadA(A[(v + 2) % 3], uu + 2);
            if ((MODE & 8) && (uu % 12) == 0) {
                if (uu) for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(smem + 4096 + 4 * (tid + 256 * j)) = xv[j];
                for (int j = 0; j < 4; ++j) {
                    const unsigned cidx = ((unsigned)blockIdx.x * 97u + (unsigned)(uu / 12)) & 65535u;
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xr, (tid + 256 * j) * 16, (int)(cidx * 16384u), 0);
                    xv[j] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int q = v * 4 + kk;                       // k-step index within the unrolled triple (0..11)
                // prefetch the fragments BD k-steps ahead
                const int qn = q + BD;
                const int un = uu + (kk + BD) / 4, kn = (kk + BD) % 4;
                readB(bq[qn % 3], un, kn);
                __builtin_amdgcn_sched_barrier(0);
                const float4 c0 = A[v][0], c1 = A[v][1];
                const float av0 = kk == 0 ? c0.x : (kk == 1 ? c0.y : (kk == 2 ? c0.z : c0.w));
                const float av1 = kk == 0 ? c1.x : (kk == 1 ? c1.y : (kk == 2 ? c1.z : c1.w));
                const float b0 = bq[q % 3][0], b1 = bq[q % 3][1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b1, acc[1][1], 0, 0, 0);
            }
            if ((MODE & 4) && (uu % 12) == 11) __syncthreads();
            if ((MODE & 16) && (uu % 96) == 95) {
                float* o = out + (((size_t)blockIdx.x * 13 + uu / 96) & 8191) * 16384 + wave * 4096 + lane;
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) o[((i * 2 + j) * 16 + r) * 64] = acc[i][j][r];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    s += xv[0].x + xv[1].y + xv[2].z + xv[3].w + A[0][0].x + A[1][1].y + A[2][0].z + bq[0][0] + bq[1][1] + bq[2][0];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 7 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int BD>
static void run2(const char* name, int wgs, int nsteps, const float* w, const float* x, float* out, unsigned long long* clk, size_t lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe2<MODE, BD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe2<MODE, BD>), dim3(wgs), dim3(256), lds, 0, w, x, out, nsteps, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe2<MODE, BD>), dim3(wgs), dim3(256), lds, 0, w, x, out, nsteps, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flop = (double)wgs * 4 * nsteps * 16 * (2.0 * 32 * 32 * 2);
    printf("%-44s wgs %5d lds %6zu  %8.3f ms  %7.2f TFLOP/s   shader clock %.3f GHz (in-kernel)\n", name, wgs, lds, ms,
           flop / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
}

int main(int argc, char** argv) {
    float *w, *x, *out; unsigned long long* clk;
    CK(hipMalloc(&w, 64 << 20)); CK(hipMalloc(&x, 1ull << 30)); CK(hipMalloc(&out, 512ull << 20)); CK(hipMalloc(&clk, 64));
    CK(hipMemset(w, 0, 64 << 20)); CK(hipMemset(x, 0, 1ull << 30));
    const bool randfill = argc > 1;
    if (randfill) {   // uniform [-1, 1) operands instead of zeros (data-dependent power -> clock)
        std::vector<float> h(16 << 20);
        srand(3);
        for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        CK(hipMemcpy(w, h.data(), 64 << 20, hipMemcpyHostToDevice));
        for (int k = 0; k < 16; ++k) CK(hipMemcpy((char*)x + ((size_t)k << 26), h.data(), 64 << 20, hipMemcpyHostToDevice));
        printf("operands: uniform random [-1,1)\n");
    } else printf("operands: zeros\n");
    const int NS = argc > 2 ? atoi(argv[2]) : 1152;  // steps per workgroup lifetime (x 16 MFMA per wave)
    const int rounds = 4 * 1152 / NS;                 // same total work whatever the lifetime
    printf("steps per workgroup %d, %d residency rounds\n", NS, rounds);
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        const size_t lds = per_cu == 4 ? 36 * 1024 : (per_cu == 3 ? 52 * 1024 : (per_cu == 2 ? 76 * 1024 : 150 * 1024));
        const int wgs = 256 * per_cu * rounds;
        printf("---- %d workgroup(s) of 4 waves per CU (LDS-limited)\n", per_cu);
        run<0>("MFMA only", wgs, NS, w, x, out, clk, lds);
        run<1>("+ B fragments from LDS", wgs, NS, w, x, out, clk, lds);
        run<3>("+ A vectors from L2 (buffer_load x2 / step)", wgs, NS, w, x, out, clk, lds);
        run<7>("+ barrier every 12 steps", wgs, NS, w, x, out, clk, lds);
        run<15>("+ activation staging (4 x 16 B / 12 steps)", wgs, NS, w, x, out, clk, lds);
        run<31>("+ streaming HBM reads + 64 KB store / 96 steps", wgs, NS, w, x, out, clk, lds);
        run2<31, 1>("x3 unroll: A 2 steps ahead, B 1 k-step ahead", wgs, NS, w, x, out, clk, lds);
        run2<31, 2>("x3 unroll: A 2 steps ahead, B 2 k-steps ahead", wgs, NS, w, x, out, clk, lds);
    }
    return 0;
}
