// vfx_gru.hip -- recurrent part of one bidirectional GRU layer (hidden 256), PyTorch gate
// order (r, z, n), h0 = 0.  Replaces torch.nn.GRU inside BN_GRU
// (voicefixer/restorer/model.py:22-62): the x-projections W_ih x + b_ih (with the scalar
// BatchNorm folded in) are ONE implicit GEMM done by vfx_conv1d_f32 beforehand; what is left
// is strictly sequential in t:
//     gh = W_hh h + b_hh ; r = s(gi_r+gh_r) ; z = s(gi_z+gh_z) ; n = tanh(gi_n + r*gh_n)
//     h' = (1-z)*n + z*h
// One workgroup owns one (utterance, direction) sequence, so there is NO inter-workgroup
// synchronisation per step.  Thread j owns hidden unit j (rows j, 256+j, 512+j of W_hh).
// W_hh^T (768 KB per direction) does not fit a CU: KREG k-rows live in VGPRs for the whole
// sequence (256 threads = one wave per SIMD = the full 512-VGPR budget), the rest streams
// from L2 every step (coalesced: thread j reads whh_t[k][g*256+j]).  h is double buffered
// in LDS: one barrier per step.
#include "vfx_common.h"

#define GRU_H 256
#define GRU_G 768

template <int KREG>
__global__ __launch_bounds__(256, 1) void gru_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                     const float* __restrict__ bhh, float* __restrict__ out,
                                                     long long o_bs, long long o_cs, int T) {
    __shared__ float hbuf[2][GRU_H];
    const int j = threadIdx.x;
    const int b = blockIdx.x, dir = blockIdx.y;
    const float* W = whh_t + (long long)dir * GRU_H * GRU_G;
    const float* g = gi + (long long)b * T * (2 * GRU_G) + dir * GRU_G;
    float* o = out + (long long)b * o_bs + (long long)(dir * GRU_H + j) * o_cs;
    const float br = bhh[dir * GRU_G + j], bz = bhh[dir * GRU_G + GRU_H + j], bn = bhh[dir * GRU_G + 2 * GRU_H + j];

    float wr[KREG > 0 ? KREG : 1], wz[KREG > 0 ? KREG : 1], wn[KREG > 0 ? KREG : 1];
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        wr[k] = W[k * GRU_G + j];
        wz[k] = W[k * GRU_G + GRU_H + j];
        wn[k] = W[k * GRU_G + 2 * GRU_H + j];
    }

    hbuf[0][j] = 0.f;
    float hj = 0.f;
    __syncthreads();

    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gr = g[(long long)t * (2 * GRU_G) + j];
    float gz = g[(long long)t * (2 * GRU_G) + GRU_H + j];
    float gn = g[(long long)t * (2 * GRU_G) + 2 * GRU_H + j];

    for (int s = 0; s < T; ++s) {
        const float* h = hbuf[s & 1];
        // prefetch next step's x-projection
        const int tn = t + dt;
        float ngr = 0.f, ngz = 0.f, ngn = 0.f;
        if (s + 1 < T) {
            ngr = g[(long long)tn * (2 * GRU_G) + j];
            ngz = g[(long long)tn * (2 * GRU_G) + GRU_H + j];
            ngn = g[(long long)tn * (2 * GRU_G) + 2 * GRU_H + j];
        }
        float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
        for (int k = 0; k < KREG; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(h + k);
            ar = fmaf(wr[k], hv.x, ar); az = fmaf(wz[k], hv.x, az); an = fmaf(wn[k], hv.x, an);
            ar = fmaf(wr[k + 1], hv.y, ar); az = fmaf(wz[k + 1], hv.y, az); an = fmaf(wn[k + 1], hv.y, an);
            ar = fmaf(wr[k + 2], hv.z, ar); az = fmaf(wz[k + 2], hv.z, az); an = fmaf(wn[k + 2], hv.z, an);
            ar = fmaf(wr[k + 3], hv.w, ar); az = fmaf(wz[k + 3], hv.w, az); an = fmaf(wn[k + 3], hv.w, an);
        }
        float ar2 = 0.f, az2 = 0.f, an2 = 0.f;
#pragma unroll 8
        for (int k = KREG; k < GRU_H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(h + k);
            const float* w0 = W + (long long)k * GRU_G + j;
            ar2 = fmaf(w0[0], hv.x, ar2); az2 = fmaf(w0[GRU_H], hv.x, az2); an2 = fmaf(w0[2 * GRU_H], hv.x, an2);
            ar2 = fmaf(w0[GRU_G], hv.y, ar2); az2 = fmaf(w0[GRU_G + GRU_H], hv.y, az2); an2 = fmaf(w0[GRU_G + 2 * GRU_H], hv.y, an2);
            ar2 = fmaf(w0[2 * GRU_G], hv.z, ar2); az2 = fmaf(w0[2 * GRU_G + GRU_H], hv.z, az2); an2 = fmaf(w0[2 * GRU_G + 2 * GRU_H], hv.z, an2);
            ar2 = fmaf(w0[3 * GRU_G], hv.w, ar2); az2 = fmaf(w0[3 * GRU_G + GRU_H], hv.w, az2); an2 = fmaf(w0[3 * GRU_G + 2 * GRU_H], hv.w, an2);
        }
        ar += ar2 + br; az += az2 + bz; an += an2 + bn;
        const float r = 1.f / (1.f + expf(-(gr + ar)));
        const float z = 1.f / (1.f + expf(-(gz + az)));
        const float n = tanhf(gn + r * an);
        hj = (1.f - z) * n + z * hj;
        hbuf[(s + 1) & 1][j] = hj;
        o[t] = hj;
        gr = ngr; gz = ngz; gn = ngn;
        t = tn;
        __syncthreads();
    }
}

extern "C" int vfx_gru_bidir_f32(const float* gi, const float* whh_t, const float* bhh, const vfx_tensor* out, int B,
                                 int T, vfx_stream_t stream) {
    if (!gi || !whh_t || !bhh || !out || !out->ptr || B <= 0 || T <= 0 || B > 65535) return VFX_EINVAL;
    if (out->lstride != 1) return VFX_EALIGN;
    hipLaunchKernelGGL(gru_kernel<112>, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, gi, whh_t, bhh,
                       (float*)out->ptr, out->bstride, out->cstride, T);
    VFX_LAUNCHED();
    return vfx_last_error();
}
