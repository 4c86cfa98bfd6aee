// vfx_stft.hip -- fused STFT(2048/441, reflect-centred, periodic hann) -> |.| -> 128-bin
// HTK mel.  Replaces voicefixer/base.py:78-85 (_pre) = fDomainHelper.py:81-110 (torchlibrosa
// conv-DFT, 4.2 GMAC per 10 s) + mel_scale.py:63-77 (dense 1025x128 matmul, 1.5 % non-zero).
//
// One workgroup (256 threads) transforms FPW consecutive frames of one utterance.  Per frame:
// the 2048 windowed samples (reflect index map applied on the fly, coalesced global reads,
// L2 absorbs the 4.6x frame overlap) go to LDS, a radix-2 Stockham autosort FFT runs in two
// LDS ping-pong buffers (11 passes x 4 butterflies per thread, twiddles from an LDS copy of
// the host-computed table), magnitudes of bins 0..1024 stay in LDS, and 128 threads apply
// the banded filterbank (<= 55 taps per mel bin).  HBM traffic is the algorithmic minimum
// 4*N + 512*T bytes per utterance; the 1025-bin spectrogram is never
// written out.
#include "vfx_common.h"

#define NFFT 2048
#define HOP 441
#define NMEL 128
#define FPW 4

// Device copies of the constant tables, one set PER DEVICE (a process may drive several GPUs; the normal
// deployment is one process per GPU): indexed by the calling thread's current HIP device.
#define VFX_MAX_DEVICES 64
struct FrontTables {
    float* window = nullptr;
    float2* twiddle = nullptr;
    int* lo = nullptr;     // HTK filterbank of the restorer front-end (vfx_frontend_init)
    int* hi = nullptr;
    int* off = nullptr;
    float* coef = nullptr;
    int nnz = 0;
    int* olo = nullptr;    // slaney filterbank of the Vocoder.oracle front-end (vfx_frontend_init_oracle)
    int* ohi = nullptr;
    int* ooff = nullptr;
    float* ocoef = nullptr;
    int onnz = 0;
};
static FrontTables g_front[VFX_MAX_DEVICES];

static FrontTables* front_tables() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VFX_MAX_DEVICES) return nullptr;
    return &g_front[dev];
}

static hipError_t front_upload(void** dst, const void* src, size_t bytes) {
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
}

extern "C" int vfx_frontend_init(const float* window, const float* twiddle, const int32_t* lo, const int32_t* hi,
                                 const int32_t* off, const float* coef, int nnz) {
    if (!window || !twiddle || !lo || !hi || !off || !coef || nnz <= 0) return VFX_EINVAL;
    FrontTables* ft = front_tables();
    if (!ft) return VFX_EINVAL;
    hipError_t e;
    if ((e = front_upload((void**)&ft->window, window, NFFT * sizeof(float))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->twiddle, twiddle, (NFFT / 2) * sizeof(float2))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->lo, lo, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->hi, hi, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->off, off, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->coef, coef, (size_t)nnz * sizeof(float))) != hipSuccess) return (int)e;
    ft->nnz = nnz;
    return VFX_OK;
}

// Test hook: copy the banded filterbank the kernels actually read back to the host (which = 0: HTK table of
// vfx_stft_mel_f32, 1: slaney table of vfx_stft_mel_oracle_f32).  coef holds up to coef_capacity floats.
extern "C" int vfx_frontend_readback(int which, int32_t* lo, int32_t* hi, int32_t* off, float* coef, int coef_capacity,
                                     int* nnz_out) {
    FrontTables* ft = front_tables();
    if (!ft || !lo || !hi || !off || !coef || !nnz_out || (which != 0 && which != 1)) return VFX_EINVAL;
    const int* dlo = which ? ft->olo : ft->lo;
    const int* dhi = which ? ft->ohi : ft->hi;
    const int* doff = which ? ft->ooff : ft->off;
    const float* dcoef = which ? ft->ocoef : ft->coef;
    const int nnz = which ? ft->onnz : ft->nnz;
    if (!dlo || nnz <= 0) return VFX_EINVAL;  // table not uploaded on this device
    if (coef_capacity < nnz) return VFX_ERANGE;
    hipError_t e;
    if ((e = hipMemcpy(lo, dlo, NMEL * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    if ((e = hipMemcpy(hi, dhi, NMEL * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    if ((e = hipMemcpy(off, doff, NMEL * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    if ((e = hipMemcpy(coef, dcoef, (size_t)nnz * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
    *nnz_out = nnz;
    return VFX_OK;
}

template <bool ORACLE>
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, long long wav_stride, int N0,
                                                       const int* __restrict__ n_rows, int T, float* __restrict__ mel,
                                                       const uint32_t* __restrict__ peak,
                                                       const float* __restrict__ window,
                                                       const float2* __restrict__ twiddle, const int* __restrict__ lo,
                                                       const int* __restrict__ hi, const int* __restrict__ off,
                                                       const float* __restrict__ coef) {
    __shared__ float2 buf0[NFFT];
    __shared__ float2 buf1[NFFT];
    __shared__ float2 tw[NFFT / 2];
    __shared__ float mag[NFFT / 2 + 4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const float* x = wav + (long long)b * wav_stride;
    const int N = n_rows ? n_rows[b] : N0;   // per-utterance sample count (ragged batches); T is then the row PITCH of mel
    const int Trow = n_rows ? 1 + N / HOP : T;

    for (int i = tid; i < NFFT / 2; i += 256) tw[i] = twiddle[i];

    for (int f = 0; f < FPW; ++f) {
        const int t = blockIdx.x * FPW + f;
        if (t >= Trow) break;  // uniform
        // ---- windowed frame.  restore path: reflect index map x_p[n] = x[reflect(HOP*t + n - 1024)];
        // Vocoder.oracle path (librosa >= 0.10 stft): zero padding and the wav / max|wav| pre-scale
        const int base = HOP * t - NFFT / 2;
        const float gain = ORACLE ? 1.0f / __uint_as_float(peak[b]) : 1.0f;
#pragma unroll
        for (int k = 0; k < NFFT / 256; ++k) {
            const int n = tid + 256 * k;
            int g = base + n;
            float v;
            if (ORACLE) {
                v = (g >= 0 && g < N) ? x[g] * gain : 0.f;
            } else {
                if (g < 0) g = -g;
                if (g >= N) g = 2 * (N - 1) - g;
                v = x[g];
            }
            buf0[n] = make_float2(v * window[n], 0.f);
        }
        __syncthreads();
        // ---- Stockham radix-2, 11 passes
        float2* src = buf0;
        float2* dst = buf1;
#pragma unroll 1
        for (int ns_log = 0; ns_log < 11; ++ns_log) {
            const int Ns = 1 << ns_log;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = tid + 256 * k;           // butterfly index, 0..1023
                const int kk = j & (Ns - 1);
                const float2 w = tw[kk << (10 - ns_log)];  // exp(-2 pi i kk / (2 Ns))
                const float2 u = src[j];
                const float2 v0 = src[j + NFFT / 2];
                const float2 v = make_float2(v0.x * w.x - v0.y * w.y, v0.x * w.y + v0.y * w.x);
                const int j0 = ((j - kk) << 1) + kk;
                dst[j0] = make_float2(u.x + v.x, u.y + v.y);
                dst[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
            }
            __syncthreads();
            float2* tmp = src; src = dst; dst = tmp;
        }
        // ---- magnitudes of bins 0..1024: sqrt(clamp(re^2 + im^2, 1e-8))
        for (int i = tid; i <= NFFT / 2; i += 256) {
            const float2 c = src[i];
            mag[i] = ORACLE ? sqrtf(c.x * c.x + c.y * c.y) : sqrtf(fmaxf(c.x * c.x + c.y * c.y, 1e-8f));
        }
        __syncthreads();
        // ---- banded mel
        if (tid < NMEL) {
            const int l = lo[tid], h = hi[tid];
            const float* cf = coef + off[tid];
            float acc = 0.f;
            for (int i = l; i <= h; ++i) acc = fmaf(mag[i], cf[i - l], acc);
            mel[((long long)b * T + t) * NMEL + tid] = acc;
        }
        __syncthreads();
    }
}

extern "C" int vfx_stft_mel_f32(const float* wav, int64_t wav_stride, int B, int N, float* mel, vfx_stream_t stream) {
    if (!wav || !mel || B <= 0 || N < NFFT / 2 + 1 || B > 65535) return VFX_EINVAL;
    const FrontTables* ft = front_tables();
    if (!ft || !ft->window || !ft->lo) return VFX_EINVAL;  // vfx_frontend_init not called on this device
    const int T = 1 + N / HOP;
    dim3 grid((T + FPW - 1) / FPW, B);
    hipLaunchKernelGGL(stft_mel_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav, (long long)wav_stride, N,
                       (const int*)nullptr, T, mel, (const uint32_t*)nullptr, ft->window, ft->twiddle, ft->lo, ft->hi,
                       ft->off, ft->coef);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// The same for a batch whose rows have DIFFERENT sample counts (n_rows: device int32[B], every n_rows[b] >= 1025):
// row b gets its own 1 + n_rows[b]/441 frames, T is the row pitch of mel (>= every row's frame count).
extern "C" int vfx_stft_mel_rows_f32(const float* wav, int64_t wav_stride, int B, const int32_t* n_rows, int T, float* mel,
                                     vfx_stream_t stream) {
    if (!wav || !mel || !n_rows || B <= 0 || T < 3 || B > 65535) return VFX_EINVAL;
    const FrontTables* ft = front_tables();
    if (!ft || !ft->window || !ft->lo) return VFX_EINVAL;
    dim3 grid((T + FPW - 1) / FPW, B);
    hipLaunchKernelGGL(stft_mel_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, wav, (long long)wav_stride, 0,
                       (const int*)n_rows, T, mel, (const uint32_t*)nullptr, ft->window, ft->twiddle, ft->lo, ft->hi, ft->off,
                       ft->coef);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// ---- Vocoder.oracle front-end (voicefixer/vocoder/base.py:61-71): wav / max|wav| -> |librosa.stft|
// (hop 441, zero padding) -> slaney-normalised HTK mel (librosa.filters.mel), tables uploaded once.
extern "C" int vfx_frontend_init_oracle(const int32_t* lo, const int32_t* hi, const int32_t* off, const float* coef,
                                        int nnz) {
    if (!lo || !hi || !off || !coef || nnz <= 0) return VFX_EINVAL;
    FrontTables* ft = front_tables();
    if (!ft) return VFX_EINVAL;
    hipError_t e;
    if ((e = front_upload((void**)&ft->olo, lo, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->ohi, hi, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->ooff, off, NMEL * sizeof(int))) != hipSuccess) return (int)e;
    if ((e = front_upload((void**)&ft->ocoef, coef, (size_t)nnz * sizeof(float))) != hipSuccess) return (int)e;
    ft->onnz = nnz;
    return VFX_OK;
}

extern "C" int vfx_stft_mel_oracle_f32(const float* wav, int64_t wav_stride, int B, int N, const uint32_t* peak,
                                       float* mel, vfx_stream_t stream) {
    if (!wav || !mel || !peak || B <= 0 || N < 1 || B > 65535) return VFX_EINVAL;
    const FrontTables* ft = front_tables();
    if (!ft || !ft->window || !ft->olo) return VFX_EINVAL;  // vfx_frontend_init / vfx_frontend_init_oracle not called
    const int T = 1 + N / HOP;
    dim3 grid((T + FPW - 1) / FPW, B);
    hipLaunchKernelGGL(stft_mel_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, wav, (long long)wav_stride, N,
                       (const int*)nullptr, T, mel, peak, ft->window, ft->twiddle, ft->olo, ft->ohi, ft->ooff, ft->ocoef);
    VFX_LAUNCHED();
    return vfx_last_error();
}


// ======================================================================================
// mode-1 pre-filter: VoiceFixer.remove_higher_frequency (voicefixer/base.py:87-104)
//   S = librosa.stft(wav)            n_fft 2048, hop 512, periodic hann, centre, zero ("constant") padding
//   E[f] = sum_t max(0, log10(|S[f,t]| + 1e-8)) ; cut = first i with sum_{f<=i} E[f] >= 0.95 * sum E
//   S'[f,t] = (f < cut) ? |S| * (S / (|S| + 1e-8)) : 0 ; y = librosa.istft(S')   (length 512 * (N // 512))
// Five small kernels; everything stays on the device (the reference does this in numpy on the host).
// ======================================================================================
#define HF_HOP 512

__device__ __forceinline__ float2* fft2048_lds(float2* buf0, float2* buf1, const float2* tw, int tid) {
    float2* src = buf0;
    float2* dst = buf1;
#pragma unroll 1
    for (int ns_log = 0; ns_log < 11; ++ns_log) {
        const int Ns = 1 << ns_log;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = tid + 256 * k;
            const int kk = j & (Ns - 1);
            const float2 w = tw[kk << (10 - ns_log)];
            const float2 u = src[j];
            const float2 v0 = src[j + NFFT / 2];
            const float2 v = make_float2(v0.x * w.x - v0.y * w.y, v0.x * w.y + v0.y * w.x);
            const int j0 = ((j - kk) << 1) + kk;
            dst[j0] = make_float2(u.x + v.x, u.y + v.y);
            dst[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2* tmp = src; src = dst; dst = tmp;
    }
    return src;
}

// (b, frame) -> complex spectrum S[b][t][0..1024]
__global__ __launch_bounds__(256) void hf_stft_kernel(const float* __restrict__ wav, long long wav_stride, int N, int T,
                                                      float2* __restrict__ S, const float* __restrict__ window,
                                                      const float2* __restrict__ twiddle) {
    __shared__ float2 buf0[NFFT];
    __shared__ float2 buf1[NFFT];
    __shared__ float2 tw[NFFT / 2];
    const int tid = threadIdx.x, t = blockIdx.x, b = blockIdx.y;
    const float* x = wav + (long long)b * wav_stride;
    for (int i = tid; i < NFFT / 2; i += 256) tw[i] = twiddle[i];
    const int base = HF_HOP * t - NFFT / 2;
#pragma unroll
    for (int k = 0; k < NFFT / 256; ++k) {
        const int n = tid + 256 * k;
        const int g = base + n;
        buf0[n] = make_float2((g >= 0 && g < N) ? x[g] * window[n] : 0.f, 0.f);
    }
    __syncthreads();
    const float2* r = fft2048_lds(buf0, buf1, tw, tid);
    float2* out = S + ((long long)b * T + t) * (NFFT / 2 + 1);
    for (int i = tid; i <= NFFT / 2; i += 256) out[i] = r[i];
}

// per bin: E[b][f] = sum_t max(0, log10(|S| + 1e-8)), fixed summation order (deterministic cut-off)
__global__ __launch_bounds__(256) void hf_energy_kernel(const float2* __restrict__ S, int T, float* __restrict__ E) {
    const int f = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (f > NFFT / 2) return;
    const float2* p = S + (long long)b * T * (NFFT / 2 + 1) + f;
    // float64 running sum, rounded once: within 1 ulp of the reference's pairwise float32 np.sum(feature, axis=1)
    // whatever T is (a sequential float32 sum drifts by ~T ulp, enough to move a borderline cut-off bin)
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
        const float2 c = p[(long long)t * (NFFT / 2 + 1)];
        acc += (double)fmaxf(log10f(sqrtf(c.x * c.x + c.y * c.y) + 1e-8f), 0.f);
    }
    E[b * (NFFT / 2 + 1) + f] = (float)acc;
}

// the reference's while loop (base.py:97-101), one thread per utterance
__global__ void hf_cutoff_kernel(const float* __restrict__ E, float ratio, int* __restrict__ cut) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const float* e = E + b * (NFFT / 2 + 1);
    // np.sum(energy_level) is a pairwise float32 sum: a float64 sum rounded once is its nearest stand-in; the
    // running level stays float32 like the reference's `curent_level += energy_level[i + 1]`
    double tot = 0.0;
    for (int f = 0; f <= NFFT / 2; ++f) tot += (double)e[f];
    const float total = (float)tot;
    const float threshold = total * ratio;
    float level = e[0];
    int i = 0;
    while (i < NFFT / 2 && level < threshold) {  // the reference would index past the end at i = 1024
        level += e[i + 1];
        ++i;
    }
    cut[b] = i;
}

// (b, frame): masked spectrum -> inverse real FFT -> windowed frame F[b][t][2048]
__global__ __launch_bounds__(256) void hf_iframe_kernel(const float2* __restrict__ S, int T, const int* __restrict__ cut,
                                                        float* __restrict__ F, const float* __restrict__ window,
                                                        const float2* __restrict__ twiddle) {
    __shared__ float2 buf0[NFFT];
    __shared__ float2 buf1[NFFT];
    __shared__ float2 tw[NFFT / 2];
    const int tid = threadIdx.x, t = blockIdx.x, b = blockIdx.y;
    for (int i = tid; i < NFFT / 2; i += 256) tw[i] = twiddle[i];
    const int c0 = cut[b];
    const float2* sp = S + ((long long)b * T + t) * (NFFT / 2 + 1);
    // Hermitian extension of the masked spectrum, conjugated: IFFT(X) = conj(FFT(conj(X))) / N
#pragma unroll
    for (int k = 0; k < NFFT / 256; ++k) {
        const int n = tid + 256 * k;
        const int f = n <= NFFT / 2 ? n : NFFT - n;
        float2 c = make_float2(0.f, 0.f);
        if (f < c0) {
            const float2 v = sp[f];
            const float m = sqrtf(v.x * v.x + v.y * v.y);
            const float g = m / (m + 1e-8f);  // spec * cos + j spec * sin with cos = re / (mag + EPS)
            c = make_float2(v.x * g, v.y * g);
        }
        // irfft ignores the imaginary part of the DC and Nyquist bins
        if (f == 0 || f == NFFT / 2) c.y = 0.f;
        buf0[n] = make_float2(c.x, n <= NFFT / 2 ? -c.y : c.y);
    }
    __syncthreads();
    const float2* r = fft2048_lds(buf0, buf1, tw, tid);
    float* out = F + ((long long)b * T + t) * NFFT;
    for (int i = tid; i < NFFT; i += 256) out[i] = r[i].x * (1.0f / NFFT) * window[i];
}

// overlap-add of 4 frames per sample + window-sum-square normalisation + centre trim
__global__ __launch_bounds__(256) void hf_ola_kernel(const float* __restrict__ F, int T, const float* __restrict__ window,
                                                     float* __restrict__ out, long long out_stride, int Nout) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= Nout) return;
    const int n = i + NFFT / 2;  // position in the untrimmed signal
    const int thi = n / HF_HOP < T - 1 ? n / HF_HOP : T - 1;
    int tlo = (n - NFFT + HF_HOP) / HF_HOP;
    if (n - NFFT + 1 <= 0) tlo = 0;
    float acc = 0.f, wss = 0.f;
    for (int t = tlo; t <= thi; ++t) {
        const int k = n - t * HF_HOP;
        if (k < 0 || k >= NFFT) continue;
        acc += F[((long long)b * T + t) * NFFT + k];
        wss += window[k] * window[k];
    }
    out[(long long)b * out_stride + i] = wss > 1.1754944e-38f ? acc / wss : acc;
}

extern "C" size_t vfx_hf_workspace_bytes(int B, int N) {
    const size_t T = 1 + (size_t)N / HF_HOP;
    return (size_t)B * T * (NFFT / 2 + 1) * sizeof(float2) + (size_t)B * T * NFFT * sizeof(float) +
           (size_t)B * (NFFT / 2 + 1) * sizeof(float) + (size_t)B * sizeof(int) + 256;
}

extern "C" int vfx_hf_cut_f32(const float* wav, int64_t wav_stride, int B, int N, float* out, int64_t out_stride,
                              float ratio, void* workspace, size_t workspace_bytes, int32_t* cutoff_out,
                              vfx_stream_t stream) {
    if (!wav || !out || !workspace || B <= 0 || N < HF_HOP || B > 65535) return VFX_EINVAL;
    const FrontTables* ft = front_tables();
    if (!ft || !ft->window) return VFX_EINVAL;  // vfx_frontend_init not called on this device
    const float* d_window = ft->window;
    const float2* d_twiddle = ft->twiddle;
    if (workspace_bytes < vfx_hf_workspace_bytes(B, N)) return VFX_ERANGE;
    if (!vfx_aligned16(workspace)) return VFX_EALIGN;
    const int T = 1 + N / HF_HOP;
    const int Nout = HF_HOP * (N / HF_HOP);
    char* ws = (char*)workspace;
    float2* S = (float2*)ws;
    ws += (size_t)B * T * (NFFT / 2 + 1) * sizeof(float2);
    float* F = (float*)ws;
    ws += (size_t)B * T * NFFT * sizeof(float);
    float* E = (float*)ws;
    ws += (size_t)B * (NFFT / 2 + 1) * sizeof(float);
    int* cut = cutoff_out ? cutoff_out : (int*)ws;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hf_stft_kernel, dim3(T, B), dim3(256), 0, s, wav, (long long)wav_stride, N, T, S, d_window,
                       d_twiddle);
    VFX_LAUNCHED();
    hipLaunchKernelGGL(hf_energy_kernel, dim3((NFFT / 2 + 256) / 256, B), dim3(256), 0, s, S, T, E);
    VFX_LAUNCHED();
    hipLaunchKernelGGL(hf_cutoff_kernel, dim3(B), dim3(64), 0, s, E, ratio, cut);
    VFX_LAUNCHED();
    hipLaunchKernelGGL(hf_iframe_kernel, dim3(T, B), dim3(256), 0, s, S, T, cut, F, d_window, d_twiddle);
    VFX_LAUNCHED();
    hipLaunchKernelGGL(hf_ola_kernel, dim3((Nout + 255) / 256, B), dim3(256), 0, s, F, T, d_window, out,
                       (long long)out_stride, Nout);
    VFX_LAUNCHED();
    return vfx_last_error();
}
