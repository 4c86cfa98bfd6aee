// Standalone reproducer attempt for the cross-kernel interference described in DESIGN.md section 6.
// Does NOT link libvfx_hip.  Build + run on an MI355X:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_hazard/repro.hip -o /tmp/pkfma_repro && /tmp/pkfma_repro
//
// Round-1 observation (inside the library): fp32 conv launches whose staging code applies the eval-BatchNorm
// scale/shift with v_pk_fma_f32 returned a wrong tile now and then while conv_x3_kernel (v_mfma_f32_32x32x16_bf16)
// ran on another stream; the same code with scalar v_fma_f32 never did.  This program isolates the two
// ingredients: a VICTIM kernel whose only arithmetic is y = lrelu(x * sc + sh) with (sc, sh) read from an LDS
// table -- once as v_pk_fma_f32 (inline asm), once as two v_fma_f32 -- verified bit-for-bit against the host,
// and a DISTURBER kernel looping on bf16 (or fp32) MFMAs on a second stream, co-resident on every CU.
// Output: mismatching launches per (victim form, disturber) cell.  All zeros = the standalone pair does NOT
// reproduce the library's symptom (the hardware explanation stays unproven).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int C = 64;          // channels (table entries)
constexpr int L = 1 << 16;     // positions per channel

// y[c][l] = lrelu(x[c][l] * sc[c] + sh[c]); one workgroup = 256 threads = one channel x 512 positions
template <bool PACKED>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ sc,
                                              const float* __restrict__ sh, float* __restrict__ y, int reps) {
    __shared__ float tab[2 * C];
    for (int i = threadIdx.x; i < C; i += 256) { tab[2 * i] = sc[i]; tab[2 * i + 1] = sh[i]; }
    __syncthreads();
    const int c = blockIdx.y;
    const int l = (blockIdx.x * 256 + threadIdx.x) * 2;
    const f32x2 v = *reinterpret_cast<const f32x2*>(x + (size_t)c * L + l);
    f32x2 r = v;
    for (int k = 0; k < reps; ++k) {      // the same value every repetition: any deviation is a wrong result
        const float s = tab[2 * c], t = tab[2 * c + 1];
        f32x2 o;
        if constexpr (PACKED) {
            const f32x2 ss = {s, s}, tt = {t, t};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(v), "v"(ss), "v"(tt));
        } else {
            float o0, o1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(o0) : "v"(v.x), "v"(s), "v"(t));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(o1) : "v"(v.y), "v"(s), "v"(t));
            o = {o0, o1};
        }
        o.x = o.x > 0.f ? o.x : o.x * 0.01f;
        o.y = o.y > 0.f ? o.y : o.y * 0.01f;
        if (k == 0) r = o;
        else if (o.x != r.x || o.y != r.y) r = {__builtin_nanf(""), __builtin_nanf("")};  // poison on any deviation
    }
    *reinterpret_cast<f32x2*>(y + (size_t)c * L + l) = r;
}

// KIND 0: v_mfma_f32_32x32x16_bf16 loop, 1: v_mfma_f32_32x32x2_f32 loop, (LDS fragment reads like the x3 kernel)
template <int KIND>
__global__ __launch_bounds__(256, 2) void disturber(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short frag[64 * 8 * 4];
    for (int i = threadIdx.x; i < 64 * 8 * 4; i += 256) frag[i] = (unsigned short)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(frag + ((lane + it) & 63) * 8);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(frag + 64 * 8 + ((lane + 2 * it) & 63) * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        } else {
            const float a = (float)frag[(lane + it) & 2047], b = (float)frag[(lane + 3 * it) & 2047];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 40;
    std::vector<float> hx((size_t)C * L), hsc(C), hsh(C), want((size_t)C * L), got((size_t)C * L);
    srand(7);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 4.f - 2.f;
    for (int c = 0; c < C; ++c) { hsc[c] = 0.8f + 0.4f * rand() / RAND_MAX; hsh[c] = 0.3f * (rand() / (float)RAND_MAX - 0.5f); }
    for (int c = 0; c < C; ++c)
        for (int l = 0; l < L; ++l) {
            float o = fmaf(hx[(size_t)c * L + l], hsc[c], hsh[c]);
            want[(size_t)c * L + l] = o > 0.f ? o : o * 0.01f;
        }
    float *dx, *dsc, *dsh, *dy, *dsink;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dy, hx.size() * 4)); CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4));
    CK(hipMalloc(&dsink, 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, hsc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, hsh.data(), C * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sd;
    CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sd));
    const dim3 vgrid(L / 512, C);
    const char* dname[3] = {"bf16 MFMA 32x32x16", "fp32 MFMA 32x32x2", "none"};
    printf("%-22s %-22s %s\n", "victim", "disturber (2nd stream)", "launches with a wrong result / trials");
    int total_bad = 0;
    for (int packed = 1; packed >= 0; --packed)
        for (int d = 0; d < 3; ++d) {
            int bad = 0;
            for (int t = 0; t < trials; ++t) {
                if (d == 0) hipLaunchKernelGGL(disturber<0>, dim3(512), dim3(256), 0, sd, dsink, 40000);
                if (d == 1) hipLaunchKernelGGL(disturber<1>, dim3(512), dim3(256), 0, sd, dsink, 20000);
                CK(hipMemsetAsync(dy, 0, hx.size() * 4, sv));
                if (packed) hipLaunchKernelGGL(victim<true>, vgrid, dim3(256), 0, sv, dx, dsc, dsh, dy, 64);
                else hipLaunchKernelGGL(victim<false>, vgrid, dim3(256), 0, sv, dx, dsc, dsh, dy, 64);
                CK(hipStreamSynchronize(sv));
                CK(hipMemcpy(got.data(), dy, hx.size() * 4, hipMemcpyDeviceToHost));
                bad += memcmp(got.data(), want.data(), hx.size() * 4) != 0;
                CK(hipStreamSynchronize(sd));
            }
            printf("%-22s %-22s %d / %d\n", packed ? "v_pk_fma_f32" : "2 x v_fma_f32", dname[d], bad, trials);
            total_bad += bad;
        }
    printf(total_bad ? "RESULT: deviations observed (see table)\n"
                     : "RESULT: no deviation in any cell -- the standalone pair does not reproduce the symptom\n");
    return 0;
}
