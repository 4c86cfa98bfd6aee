// vfx_misc.hip -- the HBM-bound glue kernels of the path: Cout==1 convolutions,
// 2x2 average pooling on pitch maps, layout changes around the denoiser, the UNet
// input/output elementwise stages, the vocoder's mel->cond front-end and the
// per-utterance peak rule + centre trim.  All are coalesced along the contiguous axis.
#include "vfx_common.h"

std::atomic<uint64_t> g_vfx_launches{0};

extern "C" int vfx_version(void) { return 100; }
#ifndef VFX_BUILD_ID
#define VFX_BUILD_ID "unstamped"
#endif
extern "C" const char* vfx_build_id(void) { return VFX_BUILD_ID; }
extern "C" uint64_t vfx_launch_count(void) { return g_vfx_launches.load(); }

// --------------------------------------------------------------------------------------
// Cout == 1 convolution: one thread per output position, channels looped, taps unrolled
// --------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const float* __restrict__ x, long long x_bs,
                                                         long long x_cs, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         long long y_bs, int Cin, int L0, int reflect, int post_act,
                                                         int out_mask, const int* __restrict__ rows) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int L = rows ? rows[b] : L0;   // ragged batches: zero / reflect padding at this row's own end
    if (l >= L) return;
    int idx[K];
    bool ok[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        int g = l + t - (K - 1) / 2;
        ok[t] = true;
        if (reflect) {
            if (g < 0) g = -g;
            if (g >= L) g = 2 * (L - 1) - g;
        } else if (g < 0 || g >= L) {
            ok[t] = false;
            g = 0;
        }
        idx[t] = g;
    }
    const float* xb = x + (long long)b * x_bs;
    float acc = bias ? bias[0] : 0.f;
    for (int c = 0; c < Cin; ++c) {
        const float* row = xb + (long long)c * x_cs;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float v = ok[t] ? row[idx[t]] : 0.f;
            acc = fmaf(w[c * K + t], v, acc);
        }
    }
    acc = vfx_post(acc, post_act, 0.f);
    if (out_mask && ((l & out_mask) == out_mask)) acc = 0.f;
    y[(long long)b * y_bs + l] = acc;
}

// The k = 7 instance of the vocoder's last convolution (ReflectionPad1d(3) + Conv1d(64, 1, 7) + Tanh, generator.py:100,145) reads
// 3.7 GB at batch 32 and is HBM-bound by its bytes -- but with one thread per output and seven dword loads per channel it
// issued 448 load instructions per thread and ran at 3.3 TB/s, bound by the texture addresser's instruction rate, not by HBM
// (round 3: 1.265 ms).  Here a thread owns FOUR consecutive outputs l .. l+3 (l a multiple of 4): per channel three aligned
// 16-byte loads deliver x[l-4 .. l+7], of which the 28 products use x[l-3 .. l+6] -- 0.75 load instructions per output and
// channel instead of 7.  Threads whose window leaves the row (the first and the last of a row: reflect / zero padding, ragged row
// ends) take the per-element index path of conv_cout1_kernel for their four outputs.
__global__ __launch_bounds__(256) void conv_cout1x4_k7_kernel(const float* __restrict__ x, long long x_bs, long long x_cs,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, long long y_bs, int Cin, int L0,
                                                              int reflect, int post_act, const int* __restrict__ rows) {
    const int l = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.y;
    const int L = rows ? rows[b] : L0;
    if (l >= L) return;
    const float* xb = x + (long long)b * x_bs;
    const float b0 = bias ? bias[0] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    if (l >= 4 && l + 8 <= L) {
        const float* p = xb + l - 4;
#pragma unroll 4
        for (int c = 0; c < Cin; ++c) {
            const float4 v0 = *reinterpret_cast<const float4*>(p);
            const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
            const float4 v2 = *reinterpret_cast<const float4*>(p + 8);
            p += x_cs;
            const float e[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};   // e[i] = x[l - 4 + i]
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const float wt = w[c * 7 + t];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = fmaf(wt, e[o + t + 1], acc[o]);   // x[l + o + t - 3]
            }
        }
    } else {
        for (int o = 0; o < 4; ++o) {
            if (l + o >= L) break;
            int idx[7];
            bool ok[7];
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                int g = l + o + t - 3;
                ok[t] = true;
                if (reflect) {
                    if (g < 0) g = -g;
                    if (g >= L) g = 2 * (L - 1) - g;
                } else if (g < 0 || g >= L) {
                    ok[t] = false;
                    g = 0;
                }
                idx[t] = g;
            }
            float s = b0;
            for (int c = 0; c < Cin; ++c) {
                const float* row = xb + (long long)c * x_cs;
#pragma unroll
                for (int t = 0; t < 7; ++t) s = fmaf(w[c * 7 + t], ok[t] ? row[idx[t]] : 0.f, s);
            }
            acc[o] = s;
        }
    }
    float* yo = y + (long long)b * y_bs + l;
    if (l + 4 <= L) {
        *reinterpret_cast<float4*>(yo) = make_float4(vfx_post(acc[0], post_act, 0.f), vfx_post(acc[1], post_act, 0.f),
                                                     vfx_post(acc[2], post_act, 0.f), vfx_post(acc[3], post_act, 0.f));
    } else {
        for (int o = 0; o < 4 && l + o < L; ++o) yo[o] = vfx_post(acc[o], post_act, 0.f);
    }
}

extern "C" int vfx_conv1d_cout1_f32(const vfx_tensor* x, const float* w, const float* bias, const vfx_tensor* y,
                                    int B, int Cin, int L, int k, int pad_mode, int post_act, int out_mask_log2,
                                    vfx_stream_t stream) {
    if (!x || !y || !w || B <= 0 || Cin <= 0 || L <= 0 || B > 65535) return VFX_EINVAL;
    if (x->lstride != 1 || y->lstride != 1) return VFX_EALIGN;
    if (pad_mode == VFX_PAD_REFLECT && L < 8) return VFX_EINVAL;
    dim3 grid((L + 255) / 256, B);
    const int mask = out_mask_log2 > 0 ? (1 << out_mask_log2) - 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    static const bool x4_off = VFX_DEV_ENV("VFX_COUT1_X4") && atoi(VFX_DEV_ENV("VFX_COUT1_X4")) == 0;   // development
    const bool vec_ok = !x4_off && mask == 0 && ((x->cstride | x->bstride | y->bstride) & 3) == 0 &&
                        (((uintptr_t)x->ptr | (uintptr_t)y->ptr) & 15) == 0;
    if (k == 7 && vec_ok)
        hipLaunchKernelGGL(conv_cout1x4_k7_kernel, dim3((L + 1023) / 1024, B), dim3(256), 0, s, (const float*)x->ptr, x->bstride,
                           x->cstride, w, bias, (float*)y->ptr, y->bstride, Cin, L, pad_mode == VFX_PAD_REFLECT, post_act,
                           (const int*)x->rows);
    else if (k == 7)
        hipLaunchKernelGGL(conv_cout1_kernel<7>, grid, dim3(256), 0, s, (const float*)x->ptr, x->bstride, x->cstride,
                           w, bias, (float*)y->ptr, y->bstride, Cin, L, pad_mode == VFX_PAD_REFLECT, post_act, mask,
                           (const int*)x->rows);
    else if (k == 1)
        hipLaunchKernelGGL(conv_cout1_kernel<1>, grid, dim3(256), 0, s, (const float*)x->ptr, x->bstride, x->cstride,
                           w, bias, (float*)y->ptr, y->bstride, Cin, L, 0, post_act, mask, (const int*)x->rows);
    else
        return VFX_EINVAL;
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// avg_pool2d(2,2) on pitch maps
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, long long x_bs, long long x_cs,
                                                      float* __restrict__ y, long long y_bs, long long y_cs, int C,
                                                      int Ho, int lp_in) {
    const int Pi = 1 << lp_in, Po = Pi >> 1;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (o >= Ho * Po) return;
    const int oy = o >> (lp_in - 1), ox = o & (Po - 1);
    float v = 0.f;
    if (ox < Po - 1) {  // W_out = floor(W_in/2) = Po-1 valid columns
        const float* p = x + (long long)b * x_bs + (long long)c * x_cs + (long long)(2 * oy) * Pi + 2 * ox;
        const float2 r0 = *reinterpret_cast<const float2*>(p);
        const float2 r1 = *reinterpret_cast<const float2*>(p + Pi);
        v = (r0.x + r0.y + r1.x + r1.y) * 0.25f;
    }
    y[(long long)b * y_bs + (long long)c * y_cs + o] = v;
}

extern "C" int vfx_avgpool2x2_f32(const vfx_tensor* x, const vfx_tensor* y, int B, int C, int H, int pitch_log2,
                                  vfx_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0 || H < 2 || (H & 1) || pitch_log2 < 2 || B > 65535 || C > 65535) return VFX_EINVAL;
    if ((x->cstride & 1) || (x->bstride & 1)) return VFX_EALIGN;
    const int Ho = H / 2, Po = 1 << (pitch_log2 - 1);
    dim3 grid((Ho * Po + 255) / 256, C, B);
    hipLaunchKernelGGL(avgpool_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, x->bstride,
                       x->cstride, (float*)y->ptr, y->bstride, y->cstride, C, Ho, pitch_log2);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// (B,T,C) frame-major -> (B,C,ld) channel-major, 32x32 LDS tile transpose
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tm_to_cm_kernel(const float* __restrict__ src, float* __restrict__ dst, int T,
                                                       int C, long long d_bs, long long d_cs) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? src[((long long)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (t < T && c < C) dst[(long long)b * d_bs + (long long)c * d_cs + t] = tile[tx][r];
    }
}

extern "C" int vfx_tm_to_cm_f32(const float* src, float* dst, int B, int T, int C, int64_t dst_bstride,
                                int64_t dst_cstride, vfx_stream_t stream) {
    if (!src || !dst || B <= 0 || T <= 0 || C <= 0 || B > 65535) return VFX_EINVAL;
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(tm_to_cm_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, T, C,
                       (long long)dst_bstride, (long long)dst_cstride);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// UNet input: U = [log10(max(mel,1e-8)), log10(max(mask*mel,1e-8))] on (Tp, 128-pitch)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unet_input_kernel(const float* __restrict__ mel, const float* __restrict__ mask,
                                                         long long m_bs, long long m_cs, float* __restrict__ u,
                                                         long long u_bs, long long u_cs, int nch, int T0, int Tp,
                                                         const int* __restrict__ t_rows) {
    // block = 32 frames x 128 bins; mask is channel-major so it goes through an LDS transpose
    __shared__ float tile[128][33];
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    const int T = t_rows ? t_rows[b] : T0;   // ragged batches: frames of this row (mel rows are T0 apart); zeros from T on
    {
        const int tx = tid & 31, cy = tid >> 5;  // 32 t x 8 c
        for (int c = cy; c < 128; c += 8) {
            const int t = t0 + tx;
            tile[c][tx] = t < T ? mask[(long long)b * m_bs + (long long)c * m_cs + t] : 0.f;
        }
    }
    __syncthreads();
    const int c = tid & 127;
    float* u0 = u + (long long)b * u_bs;
    float* u1 = u0 + u_cs;
    for (int r = tid >> 7; r < 32; r += 2) {
        const int t = t0 + r;
        if (t >= Tp) break;
        float a0 = 0.f, a1 = 0.f;
        if (t < T && c < 127) {
            const float m = mel[((long long)b * T0 + t) * 128 + c];
            a0 = log10f(fmaxf(m, 1e-8f));
            a1 = log10f(fmaxf(tile[c][r] * m, 1e-8f));
        }
        u0[(long long)t * 128 + c] = a0;
        u1[(long long)t * 128 + c] = a1;
        for (int ch = 2; ch < nch; ++ch) u0[ch * u_cs + (long long)t * 128 + c] = 0.f;  // zero filler channels
    }
}

extern "C" int vfx_unet_input_f32(const float* mel, const vfx_tensor* mask, const vfx_tensor* unet_in, int nch, int B,
                                  int T, int Tp, vfx_stream_t stream) {
    if (!mel || !mask || !unet_in || !unet_in->ptr || nch < 2 || B <= 0 || T <= 0 || Tp < T || (Tp & 63) || B > 65535)
        return VFX_EINVAL;
    if (mask->lstride != 1 || unet_in->lstride != 1) return VFX_EALIGN;
    dim3 grid((Tp + 31) / 32, B);
    hipLaunchKernelGGL(unet_input_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, (const float*)mask->ptr,
                       mask->bstride, mask->cstride, (float*)unet_in->ptr, unet_in->bstride, unet_in->cstride, nch, T,
                       Tp, (const int*)mask->rows);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// logmel = unet_out + x.  The UNet never sees bin 127 (pad column of the pitch layout) and
// emits 0 there, so x[...,127] = log10(max(mask*mel,1e-8)) is recomputed from mel and mask.
__global__ __launch_bounds__(256) void unet_output_kernel(const float* __restrict__ uo, long long uo_bs,
                                                          const float* __restrict__ ui, long long ui_bs,
                                                          long long ui_cs, const float* __restrict__ mel,
                                                          const float* __restrict__ mask, long long m_bs,
                                                          long long m_cs, float* __restrict__ logmel,
                                                          float* __restrict__ den, int T, int Tp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= (long long)T * 128) return;
    const int t = (int)(i >> 7), c = (int)(i & 127);
    float x, o;
    if (c < 127) {
        x = ui[(long long)b * ui_bs + ui_cs + i];
        o = uo[(long long)b * uo_bs + i];
    } else {
        const float m = mel[((long long)b * T + t) * 128 + 127];
        const float k = mask[(long long)b * m_bs + 127 * m_cs + t];
        x = log10f(fmaxf(k * m, 1e-8f));
        o = 0.f;
    }
    const float lm = o + x;
    logmel[(long long)b * T * 128 + i] = lm;
    den[(long long)b * T * 128 + i] = exp10f(fminf(lm, 5.f));
}

extern "C" int vfx_unet_output_f32(const vfx_tensor* unet_out, const vfx_tensor* unet_in, const float* mel,
                                    const vfx_tensor* mask, float* logmel, float* denoised, int B, int T, int Tp,
                                    vfx_stream_t stream) {
    if (!unet_out || !unet_in || !unet_out->ptr || !unet_in->ptr || !mel || !mask || !logmel || !denoised || B <= 0 ||
        T <= 0 || Tp < T || B > 65535)
        return VFX_EINVAL;
    dim3 grid((unsigned)(((long long)T * 128 + 255) / 256), B);
    hipLaunchKernelGGL(unet_output_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)unet_out->ptr,
                       unet_out->bstride, (const float*)unet_in->ptr, unet_in->bstride, unet_in->cstride, mel,
                       (const float*)mask->ptr, mask->bstride, mask->cstride, logmel, denoised, T, Tp);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// vocoder front-end: mel (B,T,128) -> cond (B,128,T') channel-major
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mel_to_cond_kernel(const float* __restrict__ mel, float* __restrict__ cond,
                                                          long long c_bs, long long c_cs, int T0, int Tc0,
                                                          int apply_weight, const int* __restrict__ t_rows) {
    __shared__ float tile[32][129];
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    const int T = t_rows ? t_rows[b] : T0;               // ragged batches: this row's frames (mel rows are T0 apart)
    const int Tc = t_rows ? T + (T & 1) + 4 : Tc0;       // ... and its own tail padding
    {
        const int c = tid & 127;
        // w_k = a*exp(b*k), k = 1..128 in float32 (config.py:296-316: torch.linspace then exp)
        const float kf = (float)(c + 1);
        const float wgt = apply_weight ? 18.8927416350036f * expf(0.0269863588184314f * kf) : 1.0f;
        for (int r = tid >> 7; r < 32; r += 2) {
            const int t = t0 + r;
            float v = -4.0f;
            if (t < T) {
                const float m = fabsf(mel[((long long)b * T0 + t) * 128 + c] / wgt);
                const float S = 20.f * log10f(fmaxf(1e-5f, m)) - 20.f;
                v = fminf(fmaxf(8.f * ((S + 115.f) / 115.f) - 4.f, -4.f), 4.f);
            }
            tile[r][c] = v;
        }
    }
    __syncthreads();
    const int tx = tid & 31;
    for (int c = tid >> 5; c < 128; c += 8) {
        const int t = t0 + tx;
        if (t < Tc) cond[(long long)b * c_bs + (long long)c * c_cs + t] = tile[tx][c];
    }
}

extern "C" int vfx_mel_to_cond_rows_f32(const float* mel, const vfx_tensor* cond, int B, int T, const int32_t* t_rows,
                                        int apply_weight, vfx_stream_t stream);
extern "C" int vfx_mel_to_cond_ex_f32(const float* mel, const vfx_tensor* cond, int B, int T, int apply_weight,
                                      vfx_stream_t stream);
extern "C" int vfx_mel_to_cond_f32(const float* mel, const vfx_tensor* cond, int B, int T, vfx_stream_t stream) {
    return vfx_mel_to_cond_ex_f32(mel, cond, B, T, 1, stream);
}

extern "C" int vfx_mel_to_cond_ex_f32(const float* mel, const vfx_tensor* cond, int B, int T, int apply_weight,
                                      vfx_stream_t stream) {
    return vfx_mel_to_cond_rows_f32(mel, cond, B, T, nullptr, apply_weight, stream);
}

// t_rows (device int32[B], may be NULL): frames of every row; T is then the row pitch of mel (>= every t_rows[b]) and
// row b is written up to its own t_rows[b] + t_rows[b]%2 + 4 frames
extern "C" int vfx_mel_to_cond_rows_f32(const float* mel, const vfx_tensor* cond, int B, int T, const int32_t* t_rows,
                                        int apply_weight, vfx_stream_t stream) {
    if (!mel || !cond || !cond->ptr || B <= 0 || T <= 0 || B > 65535) return VFX_EINVAL;
    if (cond->lstride != 1) return VFX_EALIGN;
    const int Tc = T + (T & 1) + 4;   // (>= every row's own t + t%2 + 4: the even ceiling is monotone)
    dim3 grid((Tc + 31) / 32, B);
    hipLaunchKernelGGL(mel_to_cond_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, (float*)cond->ptr,
                       cond->bstride, cond->cstride, T, Tc, apply_weight, (const int*)t_rows);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// peak rule + centre trim
// --------------------------------------------------------------------------------------
// Zeroing by a KERNEL, not hipMemsetAsync (round 4).  With hipMemsetAsync the peak workspace and the GRU mailboxes became memset
// NODES of a captured HIP graph (Pipeline.enable_graphs), and this sequence then went wrong reproducibly on ROCm 7.0 / torch 2.10:
// capture, two good replays, ONE eager pass of the same shape on the same stream (which issues its own memsets and makes torch
// allocate new segments), and every later replay returned the waveform divided by ~3 -- the peak rule saw a stale workspace
// (tools/dev/graph_retry_diag.py; found by tests/test_api_gpu.py::test_gru_retry_with_graphs_enabled_runs_eager in an order the
// full suite does not run it in).  Memset nodes on their own replay correctly (checked in isolation), so the mechanism is not
// established; with the zeroing as an ordinary kernel node the sequence is bit-exact again, and an eager launch costs the same.
__global__ void zero_u32_kernel(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int vfx_zero_u32(void* p, size_t bytes, hipStream_t s) {
    const size_t n = (bytes + 3) / 4;
    if (n == 0) return VFX_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(zero_u32_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, n);
    return vfx_last_error();
}

__global__ __launch_bounds__(256) void peak_kernel(const float* __restrict__ y, long long y_bs, int Ly0,
                                                   uint32_t* __restrict__ peak, const int* __restrict__ ly_rows) {
    const int b = blockIdx.y;
    const int Ly = ly_rows ? ly_rows[b] : Ly0;   // ragged batches: the peak is taken over this row's own samples
    const float* p = y + (long long)b * y_bs;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < Ly; i += (long long)gridDim.x * 256)
        m = fmaxf(m, fabsf(p[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        atomicMax(peak + b, __float_as_uint(m));  // non-negative floats order like uints
    }
}

__global__ __launch_bounds__(256) void trim_kernel(const float* __restrict__ y, long long y_bs, int start0,
                                                   float* __restrict__ out, long long o_bs, int N0,
                                                   const uint32_t* __restrict__ peak, const int* __restrict__ n_rows,
                                                   int Ly0, const int* __restrict__ ly_rows) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int N = n_rows ? n_rows[b] : N0;                 // per-utterance length (vfx_post_rows_f32)
    const int Ly = ly_rows ? ly_rows[b] : Ly0;
    const int start = n_rows ? (Ly - N) / 2 : start0;      // _trim_center: drop (Ly - N) // 2 samples in front
    if (i >= N) return;
    const float pk = __uint_as_float(peak[b]);
    float v = y[(long long)b * y_bs + start + i];
    if (pk > 1.0f) v = v / pk;
    out[(long long)b * o_bs + i] = v;
}

extern "C" int vfx_peak_f32(const float* y, int64_t y_bstride, int Ly, int B, uint32_t* peak, vfx_stream_t stream) {
    if (!y || !peak || B <= 0 || Ly <= 0 || B > 65535) return VFX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int e = vfx_zero_u32(peak, sizeof(uint32_t) * B, s);
    if (e != VFX_OK) return e;
    int nb = (Ly + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(peak_kernel, dim3(nb, B), dim3(256), 0, s, y, (long long)y_bstride, Ly, peak, (const int*)nullptr);
    VFX_LAUNCHED();
    return vfx_last_error();
}

extern "C" int vfx_post_f32(const float* y, int64_t y_bstride, int Ly, float* out, int64_t out_bstride, int N, int B,
                            uint32_t* peak_ws, vfx_stream_t stream) {
    if (!y || !out || !peak_ws || B <= 0 || N <= 0 || Ly < N || B > 65535) return VFX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int e = vfx_zero_u32(peak_ws, sizeof(uint32_t) * B, s);
    if (e != VFX_OK) return e;
    int nb = (Ly + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(peak_kernel, dim3(nb, B), dim3(256), 0, s, y, (long long)y_bstride, Ly, peak_ws, (const int*)nullptr);
    VFX_LAUNCHED();
    const int d = Ly - N;
    hipLaunchKernelGGL(trim_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, y, (long long)y_bstride, d / 2, out,
                       (long long)out_bstride, N, peak_ws, (const int*)nullptr, Ly, (const int*)nullptr);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// Per-row lengths: row b keeps n_rows[b] <= n_max samples (device int32[B]) of its ly_rows[b] <= Ly vocoder samples
// (ly_rows NULL: every row has Ly); the peak rule looks at the row's own samples only.
extern "C" int vfx_post_rows_f32(const float* y, int64_t y_bstride, int Ly, const int32_t* ly_rows, float* out,
                                 int64_t out_bstride, const int32_t* n_rows, int n_max, int B, uint32_t* peak_ws,
                                 vfx_stream_t stream) {
    if (!y || !out || !peak_ws || !n_rows || B <= 0 || n_max <= 0 || Ly < n_max || B > 65535) return VFX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int e = vfx_zero_u32(peak_ws, sizeof(uint32_t) * B, s);
    if (e != VFX_OK) return e;
    int nb = (Ly + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(peak_kernel, dim3(nb, B), dim3(256), 0, s, y, (long long)y_bstride, Ly, peak_ws,
                       (const int*)ly_rows);
    VFX_LAUNCHED();
    hipLaunchKernelGGL(trim_kernel, dim3((n_max + 255) / 256, B), dim3(256), 0, s, y, (long long)y_bstride, 0, out,
                       (long long)out_bstride, n_max, peak_ws, (const int*)n_rows, Ly, (const int*)ly_rows);
    VFX_LAUNCHED();
    return vfx_last_error();
}
