// vfx_gru.hip -- recurrent part of one bidirectional GRU layer (hidden 256), PyTorch gate
// order (r, z, n), h0 = 0.  Replaces torch.nn.GRU inside BN_GRU
// (voicefixer/restorer/model.py:22-62): the x-projections W_ih x + b_ih (with the scalar
// BatchNorm folded in) are ONE implicit GEMM done by vfx_conv1d_f32 beforehand; what is left
// is strictly sequential in t:
//     gh = W_hh h + b_hh ; r = s(gi_r+gh_r) ; z = s(gi_z+gh_z) ; n = tanh(gi_n + r*gh_n)
//     h' = (1-z)*n + z*h
// One workgroup (512 threads) owns one (utterance, direction) sequence, so there is NO
// inter-workgroup synchronisation per step.  Thread (j, half) owns hidden unit j (rows j,
// 256+j, 512+j of W_hh) over half of the k range; the halves are combined through LDS.
// W_hh (768 KB per direction) does not fit one CU, so it is split three ways, fixed for the
// whole sequence:
//     KREG k-rows per half  in VGPRs          (2 waves per SIMD => 256 registers per thread)
//     KLDS k-rows per half  in LDS            (~150 KB)
//     KSTR k-rows per half  streamed from L2  every step, as coalesced 16-byte buffer loads
//                                             through two small register buffers (2 in flight)
// The packed weight layout is produced by voicefixer_amd/packing.py::pack_gru_whh and
// documented in include/vfx_hip.h.
#include "vfx_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GRU_H 256
#define GRU_G 768
#define GRU_KREG 40
#define GRU_KLDS 24
#define GRU_KSTR (128 - GRU_KREG - GRU_KLDS)  // 48
#define GRU_NQ (GRU_KSTR / 4)

// packed sizes (floats) per direction
#define GRU_R_FLOATS (2 * GRU_KREG * GRU_G)
#define GRU_L_FLOATS (2 * GRU_KLDS * GRU_G)
#define GRU_S_FLOATS (2 * GRU_NQ * 3 * GRU_H * 4)

__global__ __launch_bounds__(512, 1) void gru_kernel(const float* __restrict__ gi, const float* __restrict__ wpk,
                                                     const float* __restrict__ bhh, float* __restrict__ out,
                                                     long long o_bs, long long o_cs, int Tpitch,
                                                     const int* __restrict__ t_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                               // [2][KLDS][768]
    float* hbuf = wl + GRU_L_FLOATS;               // [2][256]
    float* part = hbuf + 2 * GRU_H;                // [3][256] partial sums of half 1
    const int tid = threadIdx.x;
    const int j = tid & 255, half = tid >> 8;
    const int b = blockIdx.x, dir = blockIdx.y;
    // ragged batches: this sequence has t_rows[b] frames (the reverse direction starts at its own last frame);
    // Tpitch is the frame pitch of gi
    const int T = t_rows ? t_rows[b] : Tpitch;
    const float* Wd = wpk + (long long)dir * (GRU_R_FLOATS + GRU_L_FLOATS + GRU_S_FLOATS);
    const float* WR = Wd + (long long)half * GRU_KREG * GRU_G;
    const float* WL = Wd + GRU_R_FLOATS;
    const float* g = gi + (long long)b * Tpitch * (2 * GRU_G) + dir * GRU_G;
    float* o = out + (long long)b * o_bs + (long long)(dir * GRU_H + j) * o_cs;

    float wr[GRU_KREG], wz[GRU_KREG], wn[GRU_KREG];
#pragma unroll
    for (int k = 0; k < GRU_KREG; ++k) {
        wr[k] = WR[k * GRU_G + j];
        wz[k] = WR[k * GRU_G + GRU_H + j];
        wn[k] = WR[k * GRU_G + 2 * GRU_H + j];
    }
    for (int i = tid; i < GRU_L_FLOATS; i += 512) wl[i] = WL[i];
    float br = 0.f, bz = 0.f, bn = 0.f;
    if (half == 0) {
        br = bhh[dir * GRU_G + j];
        bz = bhh[dir * GRU_G + GRU_H + j];
        bn = bhh[dir * GRU_G + 2 * GRU_H + j];
    }
    if (tid < 2 * GRU_H) hbuf[tid] = 0.f;
    float hj = 0.f;
    __syncthreads();

    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (half == 0) {
        gr = g[(long long)t * (2 * GRU_G) + j];
        gz = g[(long long)t * (2 * GRU_G) + GRU_H + j];
        gn = g[(long long)t * (2 * GRU_G) + 2 * GRU_H + j];
    }
    const float* wlh = wl + half * GRU_KLDS * GRU_G + j;
    // uniform base (both halves), per-lane voffset = half block + j*16
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wd + GRU_R_FLOATS + GRU_L_FLOATS), (short)0, (int)(GRU_S_FLOATS * 4), 0x00020000);
    const int svoff = (half * GRU_NQ * 3 * GRU_H + j) * 16;

    for (int s = 0; s < T; ++s) {
        const float* h = hbuf + (s & 1) * GRU_H + half * 128;  // this half's k range
        // ---- streamed rows go through two small register buffers (QB quads of k each, NB batches):
        // two batches are in flight; a buffer is refilled right after it has been consumed.
        constexpr int QB = 2, NB = GRU_NQ / QB;
        static_assert(GRU_NQ % (2 * QB) == 0, "streamed rows come in an even number of batches");
        float4 s0[QB][3], s1[QB][3];
        // buffer loads: one VGPR offset + a scalar offset per load, so the streamed loads do not each
        // pin a 64-bit VGPR address across the step loop
#define GRU_LOAD(buf, batch)                                                                     \
    _Pragma("unroll") for (int q = 0; q < QB; ++q) _Pragma("unroll") for (int gg = 0; gg < 3; ++gg) { \
        const u32x4 r_ = __builtin_amdgcn_raw_buffer_load_b128(srd, svoff, ((((batch)*QB + q) * 3 + gg) * GRU_H) * 16, 0); \
        buf[q][gg] = make_float4(__uint_as_float(r_.x), __uint_as_float(r_.y), __uint_as_float(r_.z), __uint_as_float(r_.w)); \
    }
#define GRU_USE(buf, batch)                                                                      \
    _Pragma("unroll") for (int q = 0; q < QB; ++q) {                                             \
        const float4 hv = *reinterpret_cast<const float4*>(h + GRU_KREG + GRU_KLDS + 4 * ((batch)*QB + q)); \
        ar2 = fmaf(buf[q][0].x, hv.x, ar2); az2 = fmaf(buf[q][1].x, hv.x, az2); an2 = fmaf(buf[q][2].x, hv.x, an2); \
        ar2 = fmaf(buf[q][0].y, hv.y, ar2); az2 = fmaf(buf[q][1].y, hv.y, az2); an2 = fmaf(buf[q][2].y, hv.y, an2); \
        ar2 = fmaf(buf[q][0].z, hv.z, ar2); az2 = fmaf(buf[q][1].z, hv.z, az2); an2 = fmaf(buf[q][2].z, hv.z, an2); \
        ar2 = fmaf(buf[q][0].w, hv.w, ar2); az2 = fmaf(buf[q][1].w, hv.w, az2); an2 = fmaf(buf[q][2].w, hv.w, an2); \
    }
        GRU_LOAD(s0, 0)
        GRU_LOAD(s1, 1)
        __builtin_amdgcn_sched_barrier(0);
        // prefetch next step's x-projection
        const int tn = t + dt;
        float ngr = 0.f, ngz = 0.f, ngn = 0.f;
        if (half == 0 && s + 1 < T) {
            ngr = g[(long long)tn * (2 * GRU_G) + j];
            ngz = g[(long long)tn * (2 * GRU_G) + GRU_H + j];
            ngn = g[(long long)tn * (2 * GRU_G) + 2 * GRU_H + j];
        }
        float ar = br, az = bz, an = bn;
        // ---- register-resident rows
#pragma unroll
        for (int k = 0; k < GRU_KREG; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(h + k);
            ar = fmaf(wr[k], hv.x, ar); az = fmaf(wz[k], hv.x, az); an = fmaf(wn[k], hv.x, an);
            ar = fmaf(wr[k + 1], hv.y, ar); az = fmaf(wz[k + 1], hv.y, az); an = fmaf(wn[k + 1], hv.y, an);
            ar = fmaf(wr[k + 2], hv.z, ar); az = fmaf(wz[k + 2], hv.z, az); an = fmaf(wn[k + 2], hv.z, an);
            ar = fmaf(wr[k + 3], hv.w, ar); az = fmaf(wz[k + 3], hv.w, az); an = fmaf(wn[k + 3], hv.w, an);
            if ((k & 7) == 4) __builtin_amdgcn_sched_barrier(0);  // bound how far the h reads are hoisted
        }
        float ar2 = 0.f, az2 = 0.f, an2 = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; i += 2) {
            GRU_USE(s0, i)
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < NB) { GRU_LOAD(s0, i + 2) }
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
                // ---- LDS-resident rows (placed here so they overlap the refill of s0)
#pragma unroll
                for (int k = 0; k < GRU_KLDS; k += 4) {
                    const float4 hv = *reinterpret_cast<const float4*>(h + GRU_KREG + k);
                    const float* w0 = wlh + k * GRU_G;
                    ar2 = fmaf(w0[0], hv.x, ar2); az2 = fmaf(w0[GRU_H], hv.x, az2); an2 = fmaf(w0[2 * GRU_H], hv.x, an2);
                    ar2 = fmaf(w0[GRU_G], hv.y, ar2); az2 = fmaf(w0[GRU_G + GRU_H], hv.y, az2); an2 = fmaf(w0[GRU_G + 2 * GRU_H], hv.y, an2);
                    ar2 = fmaf(w0[2 * GRU_G], hv.z, ar2); az2 = fmaf(w0[2 * GRU_G + GRU_H], hv.z, az2); an2 = fmaf(w0[2 * GRU_G + 2 * GRU_H], hv.z, an2);
                    ar2 = fmaf(w0[3 * GRU_G], hv.w, ar2); az2 = fmaf(w0[3 * GRU_G + GRU_H], hv.w, az2); an2 = fmaf(w0[3 * GRU_G + 2 * GRU_H], hv.w, an2);
                    if ((k & 7) == 4) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            GRU_USE(s1, i + 1)
            __builtin_amdgcn_sched_barrier(0);
            if (i + 3 < NB) { GRU_LOAD(s1, i + 3) }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef GRU_LOAD
#undef GRU_USE
        ar += ar2; az += az2; an += an2;
        if (half == 1) {
            part[j] = ar;
            part[GRU_H + j] = az;
            part[2 * GRU_H + j] = an;
        }
        __syncthreads();
        if (half == 0) {
            ar += part[j];
            az += part[GRU_H + j];
            an += part[2 * GRU_H + j];
            const float r = 1.f / (1.f + expf(-(gr + ar)));
            const float z = 1.f / (1.f + expf(-(gz + az)));
            const float n = tanhf(gn + r * an);
            hj = (1.f - z) * n + z * hj;
            hbuf[((s + 1) & 1) * GRU_H + j] = hj;
            o[t] = hj;
            gr = ngr; gz = ngz; gn = ngn;
        }
        t = tn;
        __syncthreads();
    }
}

extern "C" int vfx_gru_bidir_f32(const float* gi, const float* whh_packed, const float* bhh, const vfx_tensor* out,
                                 int B, int T, vfx_stream_t stream) {
    if (!gi || !whh_packed || !bhh || !out || !out->ptr || B <= 0 || T <= 0 || B > 65535) return VFX_EINVAL;
    if (out->lstride != 1) return VFX_EALIGN;
    if (!vfx_aligned16(whh_packed)) return VFX_EALIGN;
    static unsigned long long attr_set = 0;   // bit d: the attribute is set on device d (one process may drive several)
    int attr_dev = 0;
    if (hipGetDevice(&attr_dev) != hipSuccess) attr_dev = 0;
    const size_t lds = (GRU_L_FLOATS + 2 * GRU_H + 3 * GRU_H) * sizeof(float);
    if (!((attr_set >> (attr_dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set |= 1ull << (attr_dev & 63);
    }
    hipLaunchKernelGGL(gru_kernel, dim3(B, 2), dim3(512), lds, (hipStream_t)stream, gi, whh_packed, bhh,
                       (float*)out->ptr, out->bstride, out->cstride, T, (const int*)out->rows);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// layout constants for the host-side packer (keeps Python and the kernel in sync)
extern "C" void vfx_gru_layout(int* kreg, int* klds, int* kstr) {
    *kreg = GRU_KREG;
    *klds = GRU_KLDS;
    *kstr = GRU_KSTR;
}

// ======================================================================================
// Two-CU variant: each (utterance, direction) sequence is owned by a PAIR of workgroups.
// Workgroup r holds the k-half r of W_hh^T entirely in registers (192 weights per thread,
// 512 threads) -- nothing is streamed from L2 any more, which is what bounds the one-workgroup
// kernel (384 KB per step at the per-CU L2 rate).  Per step each workgroup computes the partial
// gate pre-activations over its k-half for all 768 rows, keeps the 384 that belong to "its" hidden
// units (units 128r .. 128r+127 == its own k range, so h never has to travel) and hands the other
// 384 to its partner through 8-byte {tag = step+1, value} granules written with ONE agent-scope
// relaxed store each (MI355X guide, hand-off recipe R2: the data is the flag, no fence).  The
// receiver polls with relaxed agent-scope loads.  Mailboxes are zeroed by a kernel (vfx_zero_u32: not a memset node, see vfx_misc.hip) before every
// launch; spins are bounded and raise a device flag instead of hanging.
// Residency: partners have adjacent block ids and the host never launches more workgroups than CUs.
// (Round 3 tried the partners w and w + 8 -- workgroup ids are dealt round-robin to the 8 XCDs, so both land on one XCD
// -- expecting the 0.1-0.3 us the guide prices a cross-XCD hand-off above a same-XCD one: 2.20 -> 2.14 us per step for
// one utterance, but 2.22 -> 2.90 us per step at batch 32, where the 16 polling workgroups of an XCD then hit the same
// L2 channels their partners write through; profiles/r03_gru_same_xcd_pairing.txt.  Adjacent ids stay.)
// ======================================================================================
typedef unsigned long long u64;
#define G2_SPIN_LIMIT (1u << 24)

// Thread map (round 3): thread t = (unit index ju = t >> 2 of a 128-unit half, k-quarter kq = t & 3).  Every thread
// holds 2 x 3 x 32 weights: its 32 k-rows of the three gate rows of the PARTNER's unit ju and of its OWN unit ju.  A time
// step runs the partner's rows FIRST (96 FMAs, the four k-quarters of a unit are neighbouring lanes: two DPP quad
// permutes add them, no LDS, no barrier) and sends them off, THEN its own rows -- so the 0.8-1.0 us a granule needs to
// reach the other CU overlaps the second half of the arithmetic instead of following all of it.  One barrier per step
// (the new h); round 2 had thread (row j, k-half), one block of 192 FMAs, an LDS exchange between the k-halves and two
// barriers, and sent after everything was summed: 2.2 us per step.
__device__ __forceinline__ float quad_sum(float v) {
    // sum over the four lanes of a quad: quad_perm [1,0,3,2] then [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    return v;
}

__global__ __launch_bounds__(512, 2) void gru2_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t,
                                                      const float* __restrict__ bhh, float* __restrict__ out,
                                                      long long o_bs, long long o_cs, int Tpitch, u64* __restrict__ mbox,
                                                      int* __restrict__ err, const int* __restrict__ t_rows) {
    __shared__ __attribute__((aligned(16))) float hloc[2][128];
    const int tid = threadIdx.x;
    const int ju = tid >> 2, kq = tid & 3;
    const int wg = blockIdx.x;
    const int r = wg & 1, pair = wg >> 1;
    const int dir = pair & 1, b = pair >> 1;
    const int T = t_rows ? t_rows[b] : Tpitch;   // ragged batches: frames of this sequence (both partners read the same value)
    // k-rows of this thread: hidden units 128 r + 32 kq .. + 31 (this workgroup's own h values)
    const float* W = whh_t + (long long)dir * GRU_H * GRU_G + (long long)(r * 128 + kq * 32) * GRU_G;
    const float* g = gi + (long long)b * Tpitch * (2 * GRU_G) + dir * GRU_G;
    const int own = r * 128 + ju, peer = (r ^ 1) * 128 + ju;   // hidden units whose gate rows this thread works on
    const bool lead = kq == 0;             // the lane of a quad that sends / receives / finalises unit ju
    float* o = out + (long long)b * o_bs + (long long)(dir * GRU_H + own) * o_cs;
    // mailbox of workgroup (pair, r): [slot 2][384] granules, written by the partner
    u64* my_box = mbox + ((long long)pair * 2 + r) * 2 * 384;
    u64* peer_box = mbox + ((long long)pair * 2 + (r ^ 1)) * 2 * 384;

    float wpr[32], wpz[32], wpn[32];       // partner's unit: rows peer, 256 + peer, 512 + peer
    float wor[32], woz[32], won[32];       // own unit
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        wpr[k] = W[k * GRU_G + peer];
        wpz[k] = W[k * GRU_G + GRU_H + peer];
        wpn[k] = W[k * GRU_G + 2 * GRU_H + peer];
        wor[k] = W[k * GRU_G + own];
        woz[k] = W[k * GRU_G + GRU_H + own];
        won[k] = W[k * GRU_G + 2 * GRU_H + own];
    }
    float br = 0.f, bz = 0.f, bn = 0.f, hj = 0.f;
    if (lead) {
        br = bhh[dir * GRU_G + own];
        bz = bhh[dir * GRU_G + GRU_H + own];
        bn = bhh[dir * GRU_G + 2 * GRU_H + own];
    }
    if (tid < 256) hloc[tid >> 7][tid & 127] = 0.f;
    __syncthreads();

    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (lead) {
        gr = g[(long long)t * (2 * GRU_G) + own];
        gz = g[(long long)t * (2 * GRU_G) + GRU_H + own];
        gn = g[(long long)t * (2 * GRU_G) + 2 * GRU_H + own];
    }
    bool failed = false;

    for (int s = 0; s < T; ++s) {
        const float* h = hloc[s & 1] + kq * 32;
        const int tn = t + dt;
        float ngr = 0.f, ngz = 0.f, ngn = 0.f;
        if (lead && s + 1 < T) {           // next step's x-projection
            ngr = g[(long long)tn * (2 * GRU_G) + own];
            ngz = g[(long long)tn * (2 * GRU_G) + GRU_H + own];
            ngn = g[(long long)tn * (2 * GRU_G) + 2 * GRU_H + own];
        }
        // ---- partner's rows first (h is read from LDS four values at a time, again in the second half: keeping all 32
        // in registers across both halves does not fit next to the 192 weights)
        float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 hv = *reinterpret_cast<const float4*>(h + 4 * k);
            ar = fmaf(wpr[4 * k], hv.x, ar); az = fmaf(wpz[4 * k], hv.x, az); an = fmaf(wpn[4 * k], hv.x, an);
            ar = fmaf(wpr[4 * k + 1], hv.y, ar); az = fmaf(wpz[4 * k + 1], hv.y, az); an = fmaf(wpn[4 * k + 1], hv.y, an);
            ar = fmaf(wpr[4 * k + 2], hv.z, ar); az = fmaf(wpz[4 * k + 2], hv.z, az); an = fmaf(wpn[4 * k + 2], hv.z, an);
            ar = fmaf(wpr[4 * k + 3], hv.w, ar); az = fmaf(wpz[4 * k + 3], hv.w, az); an = fmaf(wpn[4 * k + 3], hv.w, an);
            if (k & 1) __builtin_amdgcn_sched_barrier(0);  // bound how far the h reads are hoisted
        }
        ar = quad_sum(ar); az = quad_sum(az); an = quad_sum(an);
        const unsigned tag = (unsigned)(s + 1);
        const int slot = (s & 1) * 384;
        if (lead) {
            // hand the partials of the partner's unit over: one 8-byte agent-scope store per value (the data is the flag)
            u64* dst = peer_box + slot + ju * 3;
            __hip_atomic_store(dst + 0, ((u64)tag << 32) | __float_as_uint(ar), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, ((u64)tag << 32) | __float_as_uint(az), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2, ((u64)tag << 32) | __float_as_uint(an), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);   // (keep the send in front of the second half)
        // ---- own rows, while the granules travel
        float cr = 0.f, cz = 0.f, cn = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 hv = *reinterpret_cast<const float4*>(h + 4 * k);
            cr = fmaf(wor[4 * k], hv.x, cr); cz = fmaf(woz[4 * k], hv.x, cz); cn = fmaf(won[4 * k], hv.x, cn);
            cr = fmaf(wor[4 * k + 1], hv.y, cr); cz = fmaf(woz[4 * k + 1], hv.y, cz); cn = fmaf(won[4 * k + 1], hv.y, cn);
            cr = fmaf(wor[4 * k + 2], hv.z, cr); cz = fmaf(woz[4 * k + 2], hv.z, cz); cn = fmaf(won[4 * k + 2], hv.z, cn);
            cr = fmaf(wor[4 * k + 3], hv.w, cr); cz = fmaf(woz[4 * k + 3], hv.w, cz); cn = fmaf(won[4 * k + 3], hv.w, cn);
            if (k & 1) __builtin_amdgcn_sched_barrier(0);
        }
        cr = quad_sum(cr); cz = quad_sum(cz); cn = quad_sum(cn);
        if (lead) {
            const u64* src = my_box + slot + ju * 3;
            // the three granules are polled TOGETHER: three loads in flight per round trip instead of three dependent ones
            u64 v0 = 0, v1 = 0, v2 = 0;
            unsigned spins = 0;
            for (;;) {
                v0 = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag && (unsigned)(v2 >> 32) == tag) break;
                if (++spins > G2_SPIN_LIMIT) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            cr += __uint_as_float((unsigned)v0) + br;
            cz += __uint_as_float((unsigned)v1) + bz;
            cn += __uint_as_float((unsigned)v2) + bn;
            const float rr = 1.f / (1.f + expf(-(gr + cr)));
            const float zz = 1.f / (1.f + expf(-(gz + cz)));
            const float nn = tanhf(gn + rr * cn);
            hj = (1.f - zz) * nn + zz * hj;
            hloc[(s + 1) & 1][ju] = hj;
            o[t] = hj;
            gr = ngr; gz = ngz; gn = ngn;
        }
        t = tn;
        if (__syncthreads_or(failed ? 1 : 0)) break;  // barrier (the new h is visible) + uniform exit if any lane timed out
    }
    if (failed) atomicExch(err, 1);
}

extern "C" int vfx_gru_bidir2_f32(const float* gi, const float* whh_t, const float* bhh, const vfx_tensor* out, int B,
                                  int T, void* mailbox, size_t mailbox_bytes, int32_t* err_flag, vfx_stream_t stream) {
    if (!gi || !whh_t || !bhh || !out || !out->ptr || !mailbox || !err_flag || B <= 0 || T <= 0) return VFX_EINVAL;
    if (out->lstride != 1) return VFX_EALIGN;
    if (B * 4 > 240) return VFX_ERANGE;  // every workgroup must be resident (one per CU, 256 CUs)
    const size_t need = (size_t)B * 2 * 2 * 2 * 384 * sizeof(u64);
    if (mailbox_bytes < need) return VFX_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    int e = vfx_zero_u32(mailbox, need, s);
    if (e != VFX_OK) return e;
    hipLaunchKernelGGL(gru2_kernel, dim3(B * 4), dim3(512), 0, s, gi, whh_t, bhh, (float*)out->ptr, out->bstride,
                       out->cstride, T, (u64*)mailbox, (int*)err_flag, (const int*)out->rows);
    VFX_LAUNCHED();
    return vfx_last_error();
}
