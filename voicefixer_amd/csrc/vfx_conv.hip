// vfx_conv.hip -- the convolution family of the VoiceFixer path as ONE implicit-GEMM
// kernel on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on MI355X), plus its opt-in
// split-bf16 sibling conv_x3_kernel (VFX_MATH_BF16X3, v_mfma_f32_32x32x16_bf16; further down).
//
// Every convolution on the path (dilated Conv1d k3, reflect-padded Conv1d k7,
// polyphase ConvTranspose1d, Conv2d 3x3/1x1 on pitch maps, ConvTranspose2d 3x3 s2,
// Linear) is a sum over "taps": for output channel n and logical position q
//     acc[n, q] = sum_t sum_c  W[slab_t][c][n] * pre(X[c][q + off_t])
// so the GEMM is  M = Cout (MFMA rows, A = packed weights, n contiguous),
//                 N = positions (MFMA cols, B = activations, l contiguous),
//                 K = Cin x taps.
// No im2col is materialised: an activation tile WITH its halo is staged once per K-chunk
// into LDS and each tap reads it at a shifted column (large dilations use one segment per
// tap instead).  The pre-activation (leaky-ReLU / eval-BatchNorm+ReLU), zero/reflect
// padding and the pad-column mask of 2-D maps are applied while staging; bias, residual,
// activation and the output index map (stride-s interleave of transposed convolutions)
// in the epilogue.
//
// Workgroup = 256 threads = 4 wave64, each wave owns RM x RL accumulators of 32x32
// (16 VGPR each).  LDS is double buffered, the next K-chunk travels global -> VGPR while
// the MFMAs of the current one run, one barrier per chunk; 33 KB of LDS and 127-155 VGPR
// keep three to four workgroups per CU so that some stage while others compute.
// Accumulators start at bias (+ residual); interior tiles (guard bands) run a branch-free
// instance, instantiated per number of activation staging slots.
#include "vfx_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#ifndef VFX_X3_ABL
#define VFX_X3_ABL 0  // development: ablations of the bf16x3 kernel only (1 = no LDS writes in the loop, 2 = no fragment reads / MFMA)
#endif
#ifndef VFX_ABL
#define VFX_ABL 0  // development ablations: 1 = no staging in the K loop, 2 = no barrier, 4 = no MFMA
#endif
#define VFX_MAXPH 8
#define VFX_MAXT 9

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#if VFX_ABL & 8
// development: per-phase cycle totals (wave 0 of every workgroup), read back with vfx_debug_read
__device__ unsigned long long g_dbg[10];
__device__ unsigned long long g_dbgx[6];   // convwg4_kernel: finer prologue split (vfx_debug_read_x)
extern "C" int vfx_debug_read(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 10);
    if (reset) { unsigned long long z[10] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)); }
    return 0;
}
extern "C" int vfx_debug_read_x(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbgx), sizeof(unsigned long long) * 6);
    if (reset) { unsigned long long z[6] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dbgx), z, sizeof(z)); }
    return 0;
}
// per-workgroup timeline of convwg4_kernel: wave 0 of every workgroup appends {HW_ID | XCC_ID << 32, linear workgroup id, s_memtime at
// entry / K loop start / K loop end / exit} to a device buffer the tool hands in (tools/wg4_timeline.py)
__device__ unsigned long long* g_trace_buf;
__device__ unsigned int g_trace_cap, g_trace_n;
extern "C" int vfx_debug_trace(unsigned long long* buf, unsigned int cap_records) {
    const unsigned int z = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &buf, sizeof(buf));
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_cap), &cap_records, sizeof(cap_records));
    hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &z, sizeof(z));
    return 0;
}
extern "C" unsigned int vfx_debug_trace_count(void) {
    unsigned int n = 0;
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), sizeof(n));
    return n;
}
#define DBG_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define DBG_T(v)
#endif

// per-phase tap tables live in device memory (cached per distinct conv geometry): a by-value
// kernarg array indexed per lane makes the compiler hold the whole table in SGPRs.
struct PhaseTab {
    int ntaps, nseg, ooff, pad_;
    int seg_org[VFX_MAXT];  // global origin of segment s relative to q0 (multiple of 4)
    int tap_lds[VFX_MAXT];  // LDS float offset of (tap t, row 0, q-local 0)
    int tap_w[VFX_MAXT];    // weight slab of tap t
};
struct ConvTables {
    PhaseTab ph[VFX_MAXPH];
};

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    const float* pre_scale;
    const float* pre_shift;
    const ConvTables* tab;
    int B, Cin, CinPad, Cout, Lin, Lq, Lout;
    long long x_bs, x_cs, y_bs, y_cs, y_ls, r_bs, r_cs, r_ls;
    int segw;       // LDS row pitch of the activation tile (floats, multiple of 4)
    int xs_floats;  // LDS floats reserved for the activation tile (per buffer)
    int ws_floats;  // LDS floats reserved for the weight tile (per buffer)
    int q_shift, q_mask, o_rs, o_cs;  // out = (q >> q_shift)*o_rs + (q & q_mask)*o_cs + ooff
    int pad_mode, pre_act, post_act;
    float pre_slope, post_slope;
    int in_mask, out_mask;  // pitch-1 (e.g. 127) or 0: positions with (l & mask) == mask are structural zeros
    int tile_lo, tile_hi;   // interior (FAST) tiles along L: [tile_lo, tile_hi)
    int tpw;                // consecutive L-tiles walked by one FAST workgroup
    int x_guard;            // readable elements before every input row (vfx_tensor.guard)
    // split-K (few workgroups, long K: the deep UNet levels at small batch): grid.z = B * ksplit, split ks reduces
    // K-chunks [ks*cpp, (ks+1)*cpp) and writes raw partial sums to ws[ks][b][n][q]; splitk_reduce_kernel adds them
    // in fixed order and applies bias / residual / activation / pad-column zeroing
    int ksplit, cpp, ws_ls;
    float* ws;
    int bl_step;            // tile step along q (= BL, or BL - span for exact-width halo tiles)
    int nxv;                // activation staging slots per thread this launch needs (host copy of the kernel's nxv)
    int res_init;           // 1: plain output map (out = q) -> the residual is loaded into the accumulators up front
    // ---- bf16x3 instance (conv_x3_kernel) only
    const void* w3;         // weights as bf16 hi/lo planes: [slab][Cin/16][plane][k-half][Cout][8]
    const ConvTables* tab3; // tap tables in POSITIONS (no 4-alignment: the x3 staging moves single floats)
    int segw3;              // positions per staged segment
    int xplane3;            // bytes between the hi and the lo plane of the activation tile
    int buf3;               // bytes per LDS buffer (activation planes + weight tile)
    int bl3;                // output positions per tile (< BL when the halo is staged inside the BL columns)
    int xrow3, wrow3;       // ROWS == 3 (3x3 on a pitch map): byte step of one map row in x / of one kernel row in w3
    // ---- convw_kernel (vfx_convw.inc): weights as L2-resident A-operand vectors, deep activation chunks
    const float* wd;        // weights packed [slab][Cin/8][Cout][8 channels in the order 0,2,4,6,1,3,5,7]
    int kcx;                // channels per activation chunk (8, 16 or 32)
    int ntiles_l, xcd_chunk;      // tiles along L; XCD-aware tile order: tiles per XCD (0 = linear order)
    int nph_fold;                 // convw_kernel, transposed convolutions: phases folded into blockIdx.x (0 = phase in blockIdx.y)
    int wg_d;                     // convwg4_kernel (Winograd F(4,3) along the dilated axis): the dilation
    const int* x_rows;            // ragged batches: valid input length of batch item b (device int32[B]) or NULL (= Lin)
    int lq_extra;                 // ... its valid output-position count is x_rows[b] + lq_extra (1 for transposed 1-D convs)
    int stagger, stagger_wgs;     // development: start stagger of the first residency round (s_sleep units, workgroups)
    int w_tap0, w_tapstep, w_slab0, w_slabstep, w_slab_ph, w_ooff0, w_ooff_ph, w_nseg, w_seg0, w_segstep;  // tap geometry (affine)
    // fused ResStack layer: second convolution (k3, dilation 1) and the LDS tile between the two
    const float* wd2;
    const float* bias2;
    int yp;                 // row pitch of the intermediate tile (floats)
    float mid_slope;        // leaky-ReLU between the two convolutions
    int rb_nq1;             // resblk4_kernel: quads of the first (dilated) half per tile = whole blocks of 4d columns x d
};

// Staging slots per thread.  The host picks KC (8 or 4) so that the activation tile never needs
// more than MAXXV float4 per thread; slots beyond the tile are clamped duplicates of its last
// vector, which keeps stage_load() straight-line code (no branch => no serialising waits).
template <int KC>
struct StageCfg {
    static constexpr int MAXXV = 4;              // float4 per thread for the activation tile
    static constexpr int MAXWV = KC == 16 ? 6 : (KC == 8 ? 4 : 5);  // float4 per thread for the weight tile
};

// Per-thread staging state: which float4 of the activation / weight tile this thread moves.
template <int MAXXV, int MAXWV>
struct StageState {
    int x_l[MAXXV];    // global l of element 0 of the vector (multiple of 4, may be < 0)
    int x_kc[MAXXV];   // row (channel within chunk)
    int x_off[MAXXV];  // kc*xcs + l
    int w_off[MAXWV];  // element offset into w for chunk 0
    float4 xv[MAXXV];
    float4 wv[MAXWV];
};

// FAST tiles: every staged vector of every chunk lies inside the row -> unconditional float4
// loads, no per-element range logic, no branches.  Boundary tiles run the general instance.
template <bool FAST, int MAXXV, int MAXWV>
__device__ __forceinline__ void stage_load(StageState<MAXXV, MAXWV>& st, const ConvArgs& a,
                                           const float* __restrict__ xb, int xcs, int c0, int lshift,
                                           __amdgpu_buffer_rsrc_t xrsrc, __amdgpu_buffer_rsrc_t wrsrc, int Lin_row) {
    if constexpr (FAST) {
        // buffer loads: per-slot byte offset in one VGPR (fixed for the whole K loop) + the chunk's offset
        // in an SGPR -> no 64-bit address arithmetic per load
        const int xso = (c0 * xcs + lshift) * 4, wso = c0 * a.Cout * 4;
#pragma unroll
        for (int j = 0; j < MAXXV; ++j) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, st.x_off[j], xso, 0);
            st.xv[j] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
#pragma unroll
        for (int j = 0; j < MAXWV; ++j) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, st.w_off[j], wso, 0);
            st.wv[j] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        }
        return;
    }
    const float* __restrict__ xc = xb + (long long)c0 * xcs + lshift;
    const float* __restrict__ wc = a.w + (long long)c0 * a.Cout;
#pragma unroll
    for (int j = 0; j < MAXXV; ++j) {
        if constexpr (FAST) {
            st.xv[j] = *reinterpret_cast<const float4*>(xc + st.x_off[j]);
        } else {
            const int Lin = Lin_row;
            const bool reflect = a.pad_mode == VFX_PAD_REFLECT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int kc = st.x_kc[j];
            if (c0 + kc < a.Cin) {
                const float* row = xc + (long long)kc * xcs;
                const int l = st.x_l[j];
                if (l >= 0 && l + 3 < Lin) {
                    v = *reinterpret_cast<const float4*>(row + l);
                } else {
                    float e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int g = l + k;
                        if (reflect) {
                            if (g < 0) g = -g;
                            if (g >= Lin) g = 2 * (Lin - 1) - g;
                        }
                        e[k] = (g >= 0 && g < Lin) ? row[g] : 0.f;
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            st.xv[j] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < MAXWV; ++j) st.wv[j] = *reinterpret_cast<const float4*>(wc + st.w_off[j]);
}

// Interior tile of a launch without BatchNorm pre-activation / pad-column mask (every vocoder and
// denoiser layer): the activation is at most a leaky-ReLU, no predicate of any kind is needed.
template <int NTHR, int MAXXV, int MAXWV>
__device__ __forceinline__ void stage_write_plain(StageState<MAXXV, MAXWV>& st, const ConvArgs& a, float* xs, float* ws,
                                                  int nxv, int nwv, int xtotal, int wtotal, int tid) {
    const float slope = a.pre_act == VFX_PRE_LRELU ? a.pre_slope : 1.f;
#pragma unroll
    for (int j = 0; j < MAXXV; ++j) {
        if (j < nxv) {
            float4 v = st.xv[j];
            v.x = v.x > 0.f ? v.x : v.x * slope;
            v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope;
            v.w = v.w > 0.f ? v.w : v.w * slope;
            int i = tid + NTHR * j;
            i = i < xtotal ? i : xtotal - 1;
            *reinterpret_cast<float4*>(xs + 4 * i) = v;
        }
    }
#pragma unroll
    for (int j = 0; j < MAXWV; ++j)
        if (j < nwv) {
            int i = tid + NTHR * j;
            i = i < wtotal ? i : wtotal - 1;
            *reinterpret_cast<float4*>(ws + 4 * i) = st.wv[j];
        }
}

template <bool FAST, int NTHR, int MAXXV, int MAXWV>
__device__ __forceinline__ void stage_write(StageState<MAXXV, MAXWV>& st, const ConvArgs& a, int c0, float* xs,
                                            float* ws, const float* aff, int nxv, int nwv, int xtotal,
                                            int wtotal, int tid, int lshift, bool range_mask, int Lin_row) {
    const int pre_act = a.pre_act;
    const float pre_slope = a.pre_slope;
    const int in_mask = a.in_mask;
#pragma unroll
    for (int j = 0; j < MAXXV; ++j) {
        if (j < nxv) {
            float e[4] = {st.xv[j].x, st.xv[j].y, st.xv[j].z, st.xv[j].w};
            if (pre_act == VFX_PRE_LRELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = vfx_lrelu(e[k], pre_slope);
            } else if (pre_act == VFX_PRE_AFFINE_LRELU) {
                const int c = c0 + st.x_kc[j];
                const float sc = aff[2 * c], sh = aff[2 * c + 1];  // LDS copy of (scale, shift)
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = vfx_lrelu(fmaf(e[k], sc, sh), pre_slope);
                if constexpr (!FAST) {
                    // zero padding applies to the ACTIVATED input
                    const int l = st.x_l[j] + lshift;
                    const bool cbad = c >= a.Cin;
                    if (a.pad_mode != VFX_PAD_REFLECT) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (l + k < 0 || l + k >= Lin_row || cbad) e[k] = 0.f;
                    }
                }
            }
            if constexpr (FAST) {
                // tile reaches into the caller's guard band: whatever was read outside [0, Lin) is padding
                if (range_mask) {
                    const int l = st.x_l[j] + lshift;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (l + k < 0 || l + k >= Lin_row) e[k] = 0.f;
                }
            }
            if (in_mask >= 3) {
                // pitch >= 4 and l % 4 == 0: only the last element of a vector can sit on the pad column
                const int l = st.x_l[j] + lshift;
                if (((l + 3) & in_mask) == in_mask) e[3] = 0.f;
            } else if (in_mask) {
                const int l = st.x_l[j] + lshift;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (((l + k) & in_mask) == in_mask) e[k] = 0.f;
            }
            int i = tid + NTHR * j;
            i = i < xtotal ? i : xtotal - 1;  // duplicates rewrite the tile's last vector with the same value
            *reinterpret_cast<float4*>(xs + 4 * i) = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
#pragma unroll
    for (int j = 0; j < MAXWV; ++j)
        if (j < nwv) {
            int i = tid + NTHR * j;
            i = i < wtotal ? i : wtotal - 1;
            *reinterpret_cast<float4*>(ws + 4 * i) = st.wv[j];
        }
}

// ---- accumulators start at the bias (C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)):
// the loads overlap the first K-chunk; in the epilogue they were 16 exposed load->use round trips per wave
template <int RM, int RL>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[RM][RL], const float* __restrict__ bias, int nbase) {
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = bias ? *reinterpret_cast<const float4*>(bias + nbase + i * 32 + 8 * g)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                acc[i][j][4 * g + 0] = bq.x; acc[i][j][4 * g + 1] = bq.y;
                acc[i][j][4 * g + 2] = bq.z; acc[i][j][4 * g + 3] = bq.w;
            }
        }
}

// ... and, for plain output maps (out = q), at bias + residual: the residual tile of this workgroup is read while
// the first K-chunk is staged instead of in 16 load->use round trips per wave after the K loop.  (In-place
// residual updates stay safe: a workgroup reads and writes only its own tile.)
template <int RM, int RL>
__device__ __forceinline__ void acc_init_residual(f32x16 (&acc)[RM][RL], const float* __restrict__ rb, int rcs,
                                                  int rls, int nbase, int qbase, int qend) {
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        const int q = qbase + j * 32;
        if (q < qend) {
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += rb[(nbase + i * 32 + (r & 3) + 8 * (r >> 2)) * rcs + q * rls];
        }
    }
}

// ---- epilogue of one tile (bias is already in the accumulators)
template <int BM, int BL, int WGM, int WGL>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[BM / WGM / 32][BL / WGL / 32], int q0,
                                              int m0, int b, int wm, int wl, int lo, int hi, int ooff, int qend) {
    constexpr int WMT = BM / WGM, WLT = BL / WGL, RM = WMT / 32, RL = WLT / 32;
    float* __restrict__ yb = a.y + (long long)b * a.y_bs;
    const float* __restrict__ rb = (a.res && !a.res_init) ? a.res + (long long)b * a.r_bs : nullptr;
    const int nbase = m0 + wm * WMT + 4 * hi;
    if (a.post_act <= VFX_POST_LRELU) {
        // hot case (no / leaky-ReLU activation): unrolled, 32-bit offsets from the per-batch base, one
        // 32x32 accumulator tile at a time (sched barriers keep the register footprint of the in-loop
        // epilogue small: it runs while the next tile's first chunks sit in the staging registers)
        const float slope = a.post_act == VFX_POST_LRELU ? a.post_slope : 1.f;
        const int ycs = (int)a.y_cs, yls = (int)a.y_ls, rcs = (int)a.r_cs, rls = (int)a.r_ls;
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int q = q0 + wl * WLT + j * 32 + lo;
            const int out = (q >> a.q_shift) * a.o_rs + (q & a.q_mask) * a.o_cs + ooff;
            const bool ok = q < qend && out >= 0 && out < a.Lout;
            const bool zero = a.out_mask && ((out & a.out_mask) == a.out_mask);
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                if (ok) {
                    const int n0 = nbase + i * 32;
                    const int yo = n0 * ycs + out * yls;
                    const int ro = n0 * rcs + out * rls;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1];
                        float v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                        if (rb) {
                            v0 += rb[ro + (8 * g + 0) * rcs];
                            v1 += rb[ro + (8 * g + 1) * rcs];
                            v2 += rb[ro + (8 * g + 2) * rcs];
                            v3 += rb[ro + (8 * g + 3) * rcs];
                        }
                        v0 = v0 > 0.f ? v0 : v0 * slope; v1 = v1 > 0.f ? v1 : v1 * slope;
                        v2 = v2 > 0.f ? v2 : v2 * slope; v3 = v3 > 0.f ? v3 : v3 * slope;
                        if (zero) { v0 = 0.f; v1 = 0.f; v2 = 0.f; v3 = 0.f; }
                        yb[yo + (8 * g + 0) * ycs] = v0;
                        yb[yo + (8 * g + 1) * ycs] = v1;
                        yb[yo + (8 * g + 2) * ycs] = v2;
                        yb[yo + (8 * g + 3) * ycs] = v3;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            const int q = q0 + wl * WLT + j * 32 + lo;
            const int out = (q >> a.q_shift) * a.o_rs + (q & a.q_mask) * a.o_cs + ooff;
            const bool ok = q < qend && out >= 0 && out < a.Lout;
            const bool zero = a.out_mask && ((out & a.out_mask) == a.out_mask);
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                // transcendental activations (a handful of layers): rolled over the 16 accumulator
                // registers (uniform dynamic index -> v_movrel) to keep the code small
#pragma unroll 1
                for (int r = 0; r < 16; ++r) {
                    const int n = nbase + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (ok) {
                        float v = acc[i][j][r];
                        if (rb) v += rb[(long long)n * a.r_cs + (long long)out * a.r_ls];
                        v = vfx_post(v, a.post_act, a.post_slope);
                        if (zero) v = 0.f;
                        yb[(long long)n * a.y_cs + (long long)out * a.y_ls] = v;
                    }
                }
            }
        }
    }
}

// split-K: raw partial sums of one tile -> workspace [ks][b][Cout][ws_ls]
template <int BM, int BL, int WGM, int WGL>
__device__ __forceinline__ void conv_epilogue_partial(const ConvArgs& a, f32x16 (&acc)[BM / WGM / 32][BL / WGL / 32],
                                                      int q0, int m0, int b, int ks, int wm, int wl, int lo, int hi,
                                                      int qend) {
    constexpr int WMT = BM / WGM, WLT = BL / WGL, RM = WMT / 32, RL = WLT / 32;
    float* __restrict__ wb = a.ws + ((long long)ks * a.B + b) * a.Cout * a.ws_ls;
    const int nbase = m0 + wm * WMT + 4 * hi;
#pragma unroll
    for (int j = 0; j < RL; ++j) {
        const int q = q0 + wl * WLT + j * 32 + lo;
        if (q < qend) {
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wb[(nbase + i * 32 + (r & 3) + 8 * (r >> 2)) * a.ws_ls + q] = acc[i][j][r];
        }
    }
}

// y[b,n,q] = post(bias[n] + res[b,n,q] + sum_ks ws[ks][b][n][q]), splits added in index order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, int ws_ls,
                                                           const float* __restrict__ bias, const float* res,
                                                           long long r_bs, long long r_cs, long long r_ls, float* y,
                                                           long long y_bs, long long y_cs, long long y_ls, int B,
                                                           int Cout, int Lq, int post_act, float post_slope,
                                                           int out_mask) {
    const long long total = (long long)B * Cout * Lq;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int q = (int)(e % Lq);
        const long long bn = e / Lq;
        const int n = (int)(bn % Cout), b = (int)(bn / Cout);
        float v = bias ? bias[n] : 0.f;
        if (res) v += res[b * r_bs + n * r_cs + q * r_ls];
        for (int ks = 0; ks < ksplit; ++ks) v += ws[(((long long)ks * B + b) * Cout + n) * ws_ls + q];
        v = vfx_post(v, post_act, post_slope);
        if (out_mask && ((q & out_mask) == out_mask)) v = 0.f;
        y[b * y_bs + n * y_cs + q * y_ls] = v;
    }
}

// exact i / d for 0 <= i < 2^20, 0 < d < 2^12 without the ~40-instruction integer division
__device__ __forceinline__ int fast_div(int i, int d, float inv) {
    int q = (int)((float)i * inv);
    if (q * d > i) --q;
    if ((q + 1) * d <= i) ++q;
    return q;
}

// NXV: activation staging slots per thread the launch needs (1 .. 4; interior instance only):
// slots are loaded unconditionally, so instantiating the exact count avoids issuing duplicate loads.
// SPLITK: the launch is cut along K (see ConvArgs::ksplit); a separate instantiation so that the split decode and
// the partial-sum epilogue cost the regular instances no registers (they sit exactly at the 4-waves-per-SIMD limit).
template <int BM, int BL, int WGM, int WGL, int KC, bool FAST, int NXV = 4, bool SPLITK = false>
__global__ __launch_bounds__(64 * WGM * WGL, WGM * WGL / 2) void conv_taps_kernel(const ConvArgs a) {
    constexpr int WMT = BM / WGM, WLT = BL / WGL, RM = WMT / 32, RL = WLT / 32;
    constexpr int NTHR = 64 * WGM * WGL;
    constexpr int MAXXV = NXV, MAXWV = StageCfg<KC>::MAXWV;
    static_assert(NXV <= StageCfg<KC>::MAXXV && (FAST || NXV == StageCfg<KC>::MAXXV), "slot count");
    static_assert((WGM * WGL == 4 || WGM * WGL == 8) && RM >= 1 && RL >= 1, "4 or 8 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    DBG_T(t_entry);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave / WGL, wl = wave % WGL;
    // FAST grid covers the interior tiles [tile_lo, tile_hi); the general grid covers the rest
    int tile = blockIdx.x;
    if constexpr (FAST) tile += a.tile_lo;
    else if (tile >= a.tile_lo) tile += a.tile_hi - a.tile_lo;
    const int q0 = tile * a.bl_step;
    const int m0g = blockIdx.y * BM;
    const int ph = m0g / a.Cout;
    const int m0 = m0g - ph * a.Cout;
    const int ksplit = SPLITK ? a.ksplit : 1;
    const int b = SPLITK ? blockIdx.z / ksplit : blockIdx.z;
    const int ks = SPLITK ? blockIdx.z - b * ksplit : 0;
    // ragged batches: every batch item has its own valid length; positions past it are zero padding of the
    // (activated) input, tiles past its last output position have nothing to do
    const int Lin_row = a.x_rows ? __builtin_amdgcn_readfirstlane(a.x_rows[b]) : a.Lin;
    const int Lq_row = a.x_rows ? Lin_row + a.lq_extra : a.Lq;
    if (q0 >= Lq_row && !SPLITK) return;
    const int qend = min(Lq_row, q0 + a.bl_step);
    const PhaseTab* __restrict__ pt = &a.tab->ph[ph];
    // block-uniform table entries: readfirstlane makes the uniformity provable (SGPRs, scalar branches)
    const int nt = __builtin_amdgcn_readfirstlane(pt->ntaps);
    const int nseg = __builtin_amdgcn_readfirstlane(pt->nseg);
    int tap_lds[VFX_MAXT];
#pragma unroll
    for (int t = 0; t < VFX_MAXT; ++t) tap_lds[t] = __builtin_amdgcn_readfirstlane(pt->tap_lds[t]);
    const int segw = a.segw;
    const int sv = segw >> 2;                 // float4 per LDS row
    const int xtotal = nseg * KC * sv;        // float4 in the activation tile
    const int wtotal = nt * KC * (BM / 4);    // float4 in the weight tile
    const int nxv = (xtotal + NTHR - 1) / NTHR;  // staging slots in use (uniform)
    const int nwv = (wtotal + NTHR - 1) / NTHR;

    const float* __restrict__ xb = a.x + (long long)b * a.x_bs;
    const int xcs = (int)a.x_cs;

    // ---- per-thread staging slots (fixed for the whole K loop).  Slot indices past the end of
    // the tile are clamped to its last vector: duplicates load and store the same value, which
    // keeps the hot loop free of per-lane predicates.
    StageState<MAXXV, MAXWV> st;
    const float inv_sv = 1.0f / (float)sv, inv_row = 1.0f / (float)(KC * sv);
#pragma unroll
    for (int j = 0; j < MAXXV; ++j) {
        int i = tid + NTHR * j;
        i = i < xtotal ? i : xtotal - 1;
        const int s = fast_div(i, KC * sv, inv_row);
        const int rem = i - s * (KC * sv);
        const int kc = fast_div(rem, sv, inv_sv);
        const int v = rem - kc * sv;
        st.x_kc[j] = kc;
        st.x_l[j] = q0 + pt->seg_org[s] + 4 * v;
        st.x_off[j] = FAST ? (kc * xcs + st.x_l[j] + a.x_guard) * 4 : kc * xcs + st.x_l[j];
    }
#pragma unroll
    for (int j = 0; j < MAXWV; ++j) {
        int i = tid + NTHR * j;
        i = i < wtotal ? i : wtotal - 1;
        const int t = i / (KC * (BM / 4));
        const int rem = i - t * (KC * (BM / 4));
        const int kc = rem / (BM / 4);
        const int v = rem - kc * (BM / 4);
        st.w_off[j] = ((pt->tap_w[t] * a.CinPad + kc) * a.Cout + m0 + 4 * v) * (FAST ? 4 : 1);
    }

    f32x16 acc[RM][RL];
    acc_init_bias<RM, RL>(acc, SPLITK ? nullptr : a.bias, m0 + wm * WMT + 4 * hi);
    if (!SPLITK && a.res && a.res_init)
        acc_init_residual<RM, RL>(acc, a.res + (long long)b * a.r_bs, (int)a.r_cs, (int)a.r_ls, m0 + wm * WMT + 4 * hi,
                                  q0 + wl * WLT + lo, qend);

    const int ooff = __builtin_amdgcn_readfirstlane(pt->ooff);
    const int a_col = wm * WMT + lo;  // column into the weight tile row
    const int b_col = wl * WLT + lo;  // column into the activation tile row
    const int nchunks = (a.Cin + KC - 1) / KC;
    const int bufstride = a.xs_floats + a.ws_floats;

    // per-channel (scale, shift) of the fused BatchNorm pre-activation: one LDS copy per workgroup
    float* aff = smem + 2 * bufstride;
    if (a.pre_act == VFX_PRE_AFFINE_LRELU) {
        for (int c = tid; c < a.CinPad; c += NTHR) {
            aff[2 * c] = c < a.Cin ? a.pre_scale[c] : 1.f;
            aff[2 * c + 1] = c < a.Cin ? a.pre_shift[c] : 0.f;
        }
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xb) - a.x_guard, (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, 0x7fffffff, 0x00020000);
    // Software pipeline over the K chunks, two steps deep: at the top of step s the registers hold
    // chunk s+1 (loaded a whole step ago, so the wait is free); they are written to the other LDS
    // buffer, the loads of chunk s+2 are issued, and only then the MFMAs of chunk s run.  One barrier
    // per step.  (A multi-tile variant with the epilogue inside this loop was measured SLOWER: its
    // extra live registers cost the third wave per SIMD, which matters more than the saved prologue.)
    const int cbase = SPLITK ? ks * a.cpp : 0;                      // first K-chunk of this split
    const int S = SPLITK ? min(a.cpp, nchunks - cbase) : nchunks;
    int ch1 = cbase + 1, ch2 = cbase + 2;  // chunk held in registers / chunk being loaded at the top of step s
    constexpr int ti1 = 0, ti2 = 0;
    stage_load<FAST>(st, a, xb, xcs, cbase * KC, 0, xrsrc, wrsrc, Lin_row);
    // does any staged vector of this tile leave [0, Lin)?  (uniform; only possible with a guard band)
    bool range_mask = false;
    if constexpr (FAST) {
        for (int sg = 0; sg < nseg; ++sg) {
            const int o = q0 + __builtin_amdgcn_readfirstlane(pt->seg_org[sg]);
            range_mask |= (o < 0) || (o + segw > Lin_row);
        }
    }
    const bool plain = FAST && !range_mask && a.in_mask == 0 && a.pre_act != VFX_PRE_AFFINE_LRELU;
    if (plain) stage_write_plain<NTHR>(st, a, smem, smem + a.xs_floats, nxv, nwv, xtotal, wtotal, tid);
    else stage_write<FAST, NTHR>(st, a, cbase * KC, smem, smem + a.xs_floats, aff, nxv, nwv, xtotal, wtotal, tid, 0, range_mask, Lin_row);
    if (S > 1) stage_load<FAST>(st, a, xb, xcs, (cbase + 1) * KC, 0, xrsrc, wrsrc, Lin_row);
    __syncthreads();
#if VFX_ABL & 8
    unsigned long long d_write = 0, d_load = 0, d_mfma = 0, d_bar = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < S; ++s) {
        const float* xs = smem + (s & 1) * bufstride;
        const float* ws = xs + a.xs_floats;
        DBG_T(t0);
#if !(VFX_ABL & 1)
        if (s + 1 < S) {
            float* nxs = smem + ((s + 1) & 1) * bufstride;
            if (plain) stage_write_plain<NTHR>(st, a, nxs, nxs + a.xs_floats, nxv, nwv, xtotal, wtotal, tid);
            else stage_write<FAST, NTHR>(st, a, ch1 * KC, nxs, nxs + a.xs_floats, aff, nxv, nwv, xtotal, wtotal, tid,
                                         ti1 * BL, range_mask, Lin_row);
#if VFX_ABL & 8
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        }
        DBG_T(t1);
        if (s + 1 < S) {
            if (s + 2 < S) stage_load<FAST>(st, a, xb, xcs, ch2 * KC, ti2 * BL, xrsrc, wrsrc, Lin_row);
        }
#endif
        DBG_T(t2);
#pragma unroll
        for (int t = 0; t < VFX_MAXT; ++t) {
            if ((t < nt) && !(VFX_ABL & 4)) {
                const float* xt = xs + tap_lds[t] + b_col;
                const float* wt = ws + t * (KC * BM) + a_col;
                // fragments of k-step kk+1 are read from LDS before the MFMAs of k-step kk are issued
                float af[2][RM], bf[2][RL];
#pragma unroll
                for (int i = 0; i < RM; ++i) af[0][i] = wt[hi * BM + i * 32];
#pragma unroll
                for (int j = 0; j < RL; ++j) bf[0][j] = xt[hi * segw + j * 32];
#pragma unroll
                for (int kk = 0; kk < KC / 2; ++kk) {
                    if (kk + 1 < KC / 2) {
#pragma unroll
                        for (int i = 0; i < RM; ++i) af[(kk + 1) & 1][i] = wt[(2 * kk + 2 + hi) * BM + i * 32];
#pragma unroll
                        for (int j = 0; j < RL; ++j) bf[(kk + 1) & 1][j] = xt[(2 * kk + 2 + hi) * segw + j * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the next step's LDS reads ahead of these MFMAs
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int j = 0; j < RL; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
                }
            }
        }
#if VFX_ABL & 8
        asm volatile("s_nop 0" ::: "memory");
#endif
        DBG_T(t3);
#if !(VFX_ABL & 2)
        __syncthreads();
#endif
        DBG_T(t4);
#if VFX_ABL & 8
        d_write += t1 - t0; d_load += t2 - t1; d_mfma += t3 - t2; d_bar += t4 - t3;
#endif
        ++ch1;
        ++ch2;
    }
#if VFX_ABL & 8
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (SPLITK) conv_epilogue_partial<BM, BL, WGM, WGL>(a, acc, q0, m0, b, ks, wm, wl, lo, hi, qend);
    else conv_epilogue<BM, BL, WGM, WGL>(a, acc, q0, m0, b, wm, wl, lo, hi, ooff, qend);
#if VFX_ABL & 8
    if (tid == 0 && FAST) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_dbg[0], d_write); atomicAdd(&g_dbg[1], d_load); atomicAdd(&g_dbg[2], d_mfma);
        atomicAdd(&g_dbg[3], d_bar); atomicAdd(&g_dbg[4], t_loop - t_start); atomicAdd(&g_dbg[5], t_end - t_loop);
        atomicAdd(&g_dbg[6], 1ull); atomicAdd(&g_dbg[7], (unsigned long long)S);
        atomicAdd(&g_dbg[8], t_start - t_entry);
    }
#endif
}


// --------------------------------------------------------------------------------------
// bf16x3 instance: the same tap-GEMM with every fp32 operand split into two bf16 terms
// (x = xh + xl, |x - xh - xl| <= 2^-17 |x|) and the product evaluated as xh*wh + xh*wl + xl*wh on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped xl*wl term is 2^-18 relative).
// Three bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16 channels.
// Measured end-to-end waveform error vs the fp32 path: 2e-6 RMS, the same size as the difference
// between two fp32 summation orders.  Opt-in (vfx_act.math).
//
// LDS image per K-chunk of 16 channels: activations as [plane][k-half][position][8 ch] bf16 (one ds_read_b128
// per lane yields the 8 consecutive k of an MFMA B operand; the 32 lanes of a k-half read 512 contiguous
// bytes: no bank conflicts), weights as [tap][plane][k-half][row][8 k] in exactly the global packing order
// (straight copy).  A staging unit is (position, 8-channel half): 8 single-float buffer loads that differ
// only in their scalar offset, activation + split in registers, two 16-byte LDS writes.
// --------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: halo tile staged in exactly BL columns (tile = BL - span output positions: no extra staging slot,
//         for small spans); 1: halo tile of BL + span columns (one extra slot); 2: one BL-column segment per tap.
// ROWS 3: a 3x3 convolution on a pitch map runs as the 3-tap (dx) case over "virtual" K-chunks (channel chunk c,
//         kernel row dy): chunk (c, dy) reads the activations one map row up/down (a scalar offset) and the weight
//         slabs of kernel row dy -- same LDS footprint and staging as the 1-D k3 case instead of 9 taps at once.
//         With NT = 1 the same mechanism splits a widely dilated 1-D k3 (taps at -d, 0, +d: no shared halo) into
//         three single-tap K-steps per channel chunk: 33 KB of LDS per workgroup instead of 98 KB, three
//         workgroups per CU instead of one.
template <int BM, int BL, int WGM, int WGL, int NT, int MODE, int ROWS>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(const ConvArgs a) {
    constexpr int WMT = BM / WGM, WLT = BL / WGL, RM = WMT / 32, RL = WLT / 32;
    constexpr int NTHR = 256;
    constexpr bool SEG = MODE == 2;
    constexpr int XDEPTH = SEG ? 1 : 2;  // activation prefetch distance in steps (= register sets)
    constexpr int NXS = SEG ? (NT * BL * 2) / NTHR : (BL * 2) / NTHR + MODE;  // (position, k-half) units per thread
    constexpr int WVEC = NT * BM * 4;                     // 16-byte vectors in the weight tile
    constexpr int NWS = (WVEC + NTHR - 1) / NTHR;         // ... per thread (clamped duplicates when not a multiple)
    static_assert(WGM * WGL == 4 && (BL * 2) % NTHR == 0 && (ROWS == 1 || (ROWS == 3 && (NT == 3 || NT == 1) && MODE == 0)),
                  "tile/thread mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    DBG_T(t_entry);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 31, hi = lane >> 5;
    const int wm = wave / WGL, wl = wave % WGL;
    const int tile = blockIdx.x + a.tile_lo;
    const int q0 = tile * a.bl3;
    const int m0g = blockIdx.y * BM;
    const int ph = m0g / a.Cout;
    const int m0 = m0g - ph * a.Cout;
    const int b = blockIdx.z;
    const PhaseTab* __restrict__ pt = &a.tab3->ph[ph];
    int tap_lds[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tap_lds[t] = __builtin_amdgcn_readfirstlane(pt->tap_lds[t]);
    const int segw = a.segw3;
    const int nseg = SEG ? NT : 1;
    const int npos = nseg * segw;
    const int xunits = npos * 2;
    const int xcs = (int)a.x_cs;
    const float* __restrict__ xb = a.x + (long long)b * a.x_bs;
    const int nchunks = a.Cin >> 4;
    const int nsteps = nchunks * ROWS;  // K steps: (channel chunk, kernel row)

    // ---- per-thread staging slots, fixed for the whole K loop.  Units are enumerated k-half major so that
    // consecutive lanes hold consecutive positions: coalesced loads, contiguous (conflict-free) LDS writes.
    int x_voff[NXS];   // byte offset of (channel 8*kh, position) from the row base; channel i of the half adds a scalar
    int x_l[NXS];      // global position (range / pad-column masks)
    int x_lds[NXS];    // byte offset of the unit's hi vector in the activation tile
    const float inv_sw = 1.0f / (float)segw;
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
        int u = tid + NTHR * j;
        u = u < xunits ? u : xunits - 1;  // clamped duplicates rewrite the same value
        const int kh = u >= npos ? 1 : 0;
        const int p = u - kh * npos;
        const int sg = SEG ? fast_div(p, segw, inv_sw) : 0;
        const int pos = p - sg * segw;
        const int l = q0 + pt->seg_org[sg] + pos;
        x_l[j] = l;
        x_voff[j] = (kh * 8 * xcs + l + a.x_guard) * 4;
        x_lds[j] = u * 16;
    }
    int w_voff[NWS];
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
        int v = tid + NTHR * j;             // LDS order [tap][plane][k-half][row], 16 bytes each
        v = v < WVEC ? v : WVEC - 1;
        const int t = v / (BM * 4);
        const int rem = v - t * (BM * 4);   // = (plane*2 + kh)*BM + row
        const int pk = rem / BM;
        const int row = rem - pk * BM;
        w_voff[j] = (((pt->tap_w[t] * nchunks * 4 + pk) * a.Cout) + m0 + row) * 16;
    }
    const int w_cstep = 4 * a.Cout * 16;  // bytes per 16-channel chunk in the packed weights

    f32x16 acc[RM][RL];
    acc_init_bias<RM, RL>(acc, a.bias, m0 + wm * WMT + 4 * hi);
    if (a.res && a.res_init) {
        const int qe0 = q0 + a.bl3;
        acc_init_residual<RM, RL>(acc, a.res + (long long)b * a.r_bs, (int)a.r_cs, (int)a.r_ls, m0 + wm * WMT + 4 * hi,
                                  q0 + wl * WLT + lo, qe0 < a.Lq ? qe0 : a.Lq);
    }

    const int ooff = __builtin_amdgcn_readfirstlane(pt->ooff);
    // (ROWS 3: the base sits one map row lower so that the scalar row offset dy * xrow3 is never negative --
    // buffer addressing treats the scalar offset as unsigned)
    const __amdgpu_buffer_rsrc_t xrsrc_real = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(xb) - a.x_guard - (ROWS == 3 ? (a.xrow3 >> 2) : 0), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc_real =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w3), (short)0, 0x7fffffff, 0x00020000);
    // loads of chunks past the end stay in the instruction stream (see the loop) but go through a zero-length
    // buffer resource: they return 0 without touching memory
    const __amdgpu_buffer_rsrc_t nullrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, 0, 0x00020000);

    bool range_mask = false;
#pragma unroll
    for (int sg = 0; sg < (SEG ? NT : 1); ++sg) {
        const int o = q0 + __builtin_amdgcn_readfirstlane(pt->seg_org[sg]);
        const int rowspan = ROWS == 3 ? (a.xrow3 >> 2) : 0;
        range_mask |= (o - rowspan < 0) || (o + segw + rowspan > a.Lin);
    }
    const bool masked = range_mask || a.in_mask != 0;
    const float slope = a.pre_act != VFX_PRE_NONE ? a.pre_slope : 1.f;
    const int xplane = a.xplane3, bufb = a.buf3;
    const int wbase = 2 * xplane;  // weight tile follows the two activation planes

    // Prefetch distances: activations (HBM latency) two steps -- two register sets, the loads of chunk s+3 are
    // issued in step s and consumed in step s+2; weights (L2 hits) one step -- one register set, loaded BEFORE the
    // activations of the same step so that waiting for them (in-order vmcnt) never waits for younger activation
    // loads.  Per-tap segments (MODE 2) need three slots per set and keep one set.
    const int xcs4 = __builtin_amdgcn_readfirstlane(xcs * 4);
    float xr[XDEPTH][NXS][8];
    u32x4 wr[NWS];
    // step v -> (channel chunk, kernel row)
    auto chunk_of = [&](int v) { return ROWS == 3 ? (int)(((unsigned)v * 0xAAABu) >> 17) : v; };  // v / 3, v < 2^15
    auto load_x = [&](int v, int set) {
        const __amdgpu_buffer_rsrc_t xrsrc = v < nsteps ? xrsrc_real : nullrsrc;
        const int c = chunk_of(v);
        const int xso = __builtin_amdgcn_readfirstlane(c * 16 * xcs * 4 + (ROWS == 3 ? (v - 3 * c) * a.xrow3 : 0));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int so = xso + i * xcs4;  // scalar: one s_add per channel row
#pragma unroll
            for (int j = 0; j < NXS; ++j)
                xr[set][j][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, x_voff[j], so, 0));
        }
    };
    auto load_w = [&](int v) {
        const __amdgpu_buffer_rsrc_t wrsrc = v < nsteps ? wrsrc_real : nullrsrc;
        const int c = chunk_of(v);
        const int wso = __builtin_amdgcn_readfirstlane(c * w_cstep + (ROWS == 3 ? (v - 3 * c) * a.wrow3 : 0));
#pragma unroll
        for (int j = 0; j < NWS; ++j) wr[j] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff[j], wso, 0);
    };
    const bool affine = a.pre_act == VFX_PRE_AFFINE_LRELU;
    const float4* aff4 = reinterpret_cast<const float4*>(smem3 + 2 * bufb);  // [Cin] x (scale, shift), two per float4
    auto write_chunk = [&](unsigned char* buf, int set, int v) {
        const int c = chunk_of(v);
        const int roff = ROWS == 3 ? (v - 3 * c - 1) * (a.xrow3 >> 2) : 0;  // this step's row offset in positions
#pragma unroll
        for (int j = 0; j < NXS; ++j) {
            u32x4 h4, l4;
            const int kh = (tid + NTHR * j) >= npos ? 1 : 0;  // (clamped duplicates sit in k-half 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v0 = xr[set][j][2 * i], v1 = xr[set][j][2 * i + 1];
                if (affine) {  // eval-BatchNorm: (scale, shift) of channels 16c + 8kh + 2i, +1 from the LDS copy
                    const float4 ss = aff4[c * 8 + kh * 4 + i];
                    v0 = fmaf(v0, ss.x, ss.y);
                    v1 = fmaf(v1, ss.z, ss.w);
                }
                v0 = v0 > 0.f ? v0 : v0 * slope;
                v1 = v1 > 0.f ? v1 : v1 * slope;
                const f32x2 v = {v0, v1};
                const bf16x2 h = __builtin_convertvector(v, bf16x2);
                const f32x2 r = v - __builtin_convertvector(h, f32x2);
                const bf16x2 l = __builtin_convertvector(r, bf16x2);
                h4[i] = __builtin_bit_cast(unsigned, h);
                l4[i] = __builtin_bit_cast(unsigned, l);
            }
            if (masked) {
                const int l = x_l[j] + roff;
                const bool z = (range_mask && (l < 0 || l >= a.Lin)) || (a.in_mask && ((l & a.in_mask) == a.in_mask));
                if (z) { h4 = u32x4{0, 0, 0, 0}; l4 = u32x4{0, 0, 0, 0}; }
            }
            *reinterpret_cast<u32x4*>(buf + x_lds[j]) = h4;
            *reinterpret_cast<u32x4*>(buf + xplane + x_lds[j]) = l4;
            __builtin_amdgcn_sched_barrier(0);  // one slot's conversion temporaries at a time
        }
#pragma unroll
        for (int j = 0; j < NWS; ++j) {
            int v = tid + NTHR * j;
            if (WVEC % NTHR) v = v < WVEC ? v : WVEC - 1;  // duplicates rewrite the tile's last vector
            *reinterpret_cast<u32x4*>(buf + wbase + v * 16) = wr[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // fragment addresses: a lane's 8 consecutive k sit in one 16-byte granule; granules of one k-half are
    // contiguous over rows / positions, so every 16-lane group of a ds_read_b128 covers 256 contiguous bytes
    const int a_off = (hi * BM + wm * WMT + lo) * 16;    // inside one (tap, plane) weight tile of BM*32 bytes
    const int b_off = (hi * npos + wl * WLT + lo) * 16;  // inside one activation plane

    // one tap's fragments live at a time: the other wave of the SIMD covers the LDS latency (a second fragment
    // set, i.e. reading tap t+1 during tap t's MFMAs, was measured: no gain)
    auto mfma_chunk = [&](const unsigned char* cur) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8 ah[RM], al[RM], bh[RL], bl[RL];
            const unsigned char* wt = cur + wbase + t * (2 * BM * 32) + a_off;
            const unsigned char* xt = cur + tap_lds[t] * 16 + b_off;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(wt + i * 512);
                al[i] = *reinterpret_cast<const bf16x8*>(wt + BM * 32 + i * 512);
            }
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(xt + j * 512);
                bl[j] = *reinterpret_cast<const bf16x8*>(xt + xplane + j * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // nsteps is even (host: Cin % 32 == 0).  Every load below is unconditional: a branch around a load makes
    // the compiler merge "in flight" with "not in flight" at the join and wait for everything (vmcnt(0)), which
    // would undo the prefetch distance.  Steps past the end are read through the zero-length resource.
    load_w(0);
    load_x(0, 0);
    if (affine) {
        float2* aff = reinterpret_cast<float2*>(smem3 + 2 * bufb);
        for (int ch = tid; ch < a.Cin; ch += NTHR) aff[ch] = make_float2(a.pre_scale[ch], a.pre_shift[ch]);
        __syncthreads();
    }
    write_chunk(smem3, 0, 0);
    load_w(1);
    load_x(1, 0);
    if constexpr (XDEPTH == 2) load_x(2, 1);
    __syncthreads();
#if VFX_ABL & 8
    unsigned long long d_write = 0, d_load = 0, d_mfma = 0, d_bar = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    // step s computes K-step s from LDS buffer s&1; activation set (s&1 or 0) and the weight set hold step s+1.
    // The last pair runs the same body.  (A peeled tail made LLVM rotate the loop around the common prefix,
    // which cost 64 accumulator moves per iteration and ~40 spilled registers; nounroll keeps it from peeling.)
#pragma nounroll
    for (int s = 0; s < nsteps; s += 2) {
        DBG_T(t0);
#if !(VFX_ABL & 1) && !(VFX_X3_ABL & 1)
        write_chunk(smem3 + bufb, 0, s + 1);
#if VFX_ABL & 8
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        DBG_T(t1);
        load_w(s + 2);
        load_x(s + (XDEPTH == 2 ? 3 : 2), 0);
#else
        DBG_T(t1);
#endif
        DBG_T(t2);
#if !(VFX_ABL & 4) && !(VFX_X3_ABL & 2)
        mfma_chunk(smem3);
#endif
#if VFX_ABL & 8
        asm volatile("s_nop 0" ::: "memory");
#endif
        DBG_T(t3);
#if !(VFX_ABL & 2)
        __syncthreads();
#endif
        DBG_T(t4);
#if !(VFX_ABL & 1) && !(VFX_X3_ABL & 1)
        if (s + 2 < nsteps) write_chunk(smem3, XDEPTH - 1, s + 2);  // (no load inside the branch)
#if VFX_ABL & 8
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        DBG_T(t5);
        load_w(s + 3);
        load_x(s + (XDEPTH == 2 ? 4 : 3), XDEPTH - 1);
#else
        DBG_T(t5);
#endif
        DBG_T(t6);
#if !(VFX_ABL & 4) && !(VFX_X3_ABL & 2)
        mfma_chunk(smem3 + bufb);
#endif
#if VFX_ABL & 8
        asm volatile("s_nop 0" ::: "memory");
#endif
        DBG_T(t7);
#if !(VFX_ABL & 2)
        __syncthreads();
#endif
        DBG_T(t8);
#if VFX_ABL & 8
        d_write += (t1 - t0) + (t5 - t4); d_load += (t2 - t1) + (t6 - t5);
        d_mfma += (t3 - t2) + (t7 - t6); d_bar += (t4 - t3) + (t8 - t7);
#endif
    }
#if VFX_ABL & 8
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    const int qe = q0 + a.bl3;
    conv_epilogue<BM, BL, WGM, WGL>(a, acc, q0, m0, b, wm, wl, lo, hi, ooff, qe < a.Lq ? qe : a.Lq);
#if VFX_ABL & 8
    if (tid == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_dbg[0], d_write); atomicAdd(&g_dbg[1], d_load); atomicAdd(&g_dbg[2], d_mfma);
        atomicAdd(&g_dbg[3], d_bar); atomicAdd(&g_dbg[4], t_loop - t_start); atomicAdd(&g_dbg[5], t_end - t_loop);
        atomicAdd(&g_dbg[6], 1ull); atomicAdd(&g_dbg[7], (unsigned long long)nsteps);
        atomicAdd(&g_dbg[8], t_start - t_entry);
    }
#endif
}

// --------------------------------------------------------------------------------------
// host side: tap tables, tile choice, launch
// --------------------------------------------------------------------------------------
struct TapSpec {
    int off;
    int slab;
};
struct PhaseSpec {
    int ntaps;
    TapSpec taps[VFX_MAXT];
    int ooff;
};

struct TileCfg {
    int BM, BL;
    float util;
};
static const TileCfg kTiles[] = {
    {128, 128, 1.00f}, {64, 256, 1.00f}, {128, 64, 0.90f},
    {32, 256, 0.90f},  {64, 64, 0.75f},  {32, 128, 0.75f},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

// Device-resident copies of tap tables, one per (device, distinct geometry) (a few dozen per model).
// First use of a geometry does a synchronous hipMalloc+hipMemcpy (warm-up); afterwards the
// lookup is a host-side map hit, so steady-state launches stay asynchronous/capturable.
static const ConvTables* device_tables(const ConvTables& tb) {
    static std::mutex mu;
    static std::map<std::string, const ConvTables*> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::string key(reinterpret_cast<const char*>(&dev), sizeof(dev));  // device memory: one copy per device
    key.append(reinterpret_cast<const char*>(&tb), sizeof(tb));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    {
        static const bool dump = VFX_DEV_ENV("VFX_DEBUG_ARGS") && atoi(VFX_DEV_ENV("VFX_DEBUG_ARGS")) != 0;  // development
        if (dump) {
            fprintf(stderr, "conv table:");
            const unsigned* w = reinterpret_cast<const unsigned*>(&tb);
            for (size_t i = 0; i < sizeof(PhaseTab) / 4; ++i) fprintf(stderr, " %08x", w[i]);
            fprintf(stderr, "\n");
        }
    }
    ConvTables* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(ConvTables)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, &tb, sizeof(ConvTables), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    cache.emplace(std::move(key), d);
    return d;
}

static thread_local int g_last_tile = 0;
extern "C" int vfx_last_conv_tile(void) { return g_last_tile; }

static inline int floor4(int v) { return v >= 0 ? (v & ~3) : -(((-v) + 3) & ~3); }

template <int BM, int BL, int WGM, int WGL, int KC, bool FAST, int NXV = 4, bool SPLITK = false>
static int launch_one(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static unsigned long long attr_set = 0;   // bit d: the attribute is set on device d (one process may drive several)
    int attr_dev = 0;
    if (hipGetDevice(&attr_dev) != hipSuccess) attr_dev = 0;
    auto kern = conv_taps_kernel<BM, BL, WGM, WGL, KC, FAST, NXV, SPLITK>;
    if (!((attr_set >> (attr_dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set |= 1ull << (attr_dev & 63);
    }
    {
        static const bool dump = VFX_DEV_ENV("VFX_DEBUG_ARGS") && atoi(VFX_DEV_ENV("VFX_DEBUG_ARGS")) != 0;  // development
        if (dump) {
            fprintf(stderr, "conv_taps<%d,%d,%d,%d,%d,%d,%d,%d> grid %u %u %u lds %zu args:", BM, BL, WGM, WGL, KC, (int)FAST,
                    NXV, (int)SPLITK, grid.x, grid.y, grid.z, lds);
            const unsigned* w = reinterpret_cast<const unsigned*>(&a);
            for (size_t i = 0; i < sizeof(ConvArgs) / 4; ++i) fprintf(stderr, " %08x", w[i]);
            fprintf(stderr, "\n");
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGL), lds, s, a);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// interior tiles go to the FAST instance, the (few) boundary tiles to the general one
template <int BM, int BL, int WGM, int WGL, int KC>
static int launch_cfg(const ConvArgs& a, int ntiles, int gy, int gz, size_t lds, hipStream_t s) {
    const int nfast = a.tile_hi - a.tile_lo;
    int rc = VFX_OK;
    if (a.ksplit > 1) {  // split-K launches (few, small): the generic-slot-count instances
        if (nfast > 0) rc = launch_one<BM, BL, WGM, WGL, KC, true, 4, true>(a, dim3(nfast, gy, gz), lds, s);
        if (rc == VFX_OK && ntiles - nfast > 0)
            rc = launch_one<BM, BL, WGM, WGL, KC, false, 4, true>(a, dim3(ntiles - nfast, gy, gz), lds, s);
        return rc;
    }
    if (nfast > 0) {
        const dim3 grid((nfast + a.tpw - 1) / a.tpw, gy, gz);
        if constexpr (WGM * WGL == 4) {
            rc = a.nxv <= 1   ? launch_one<BM, BL, WGM, WGL, KC, true, 1>(a, grid, lds, s)
                 : a.nxv == 2 ? launch_one<BM, BL, WGM, WGL, KC, true, 2>(a, grid, lds, s)
                 : a.nxv == 3 ? launch_one<BM, BL, WGM, WGL, KC, true, 3>(a, grid, lds, s)
                              : launch_one<BM, BL, WGM, WGL, KC, true, 4>(a, grid, lds, s);
        } else {
            rc = launch_one<BM, BL, WGM, WGL, KC, true, 4>(a, grid, lds, s);
        }
    }
    if (rc == VFX_OK && ntiles - nfast > 0)
        rc = launch_one<BM, BL, WGM, WGL, KC, false>(a, dim3(ntiles - nfast, gy, gz), lds, s);
    return rc;
}

static int fill_segments(ConvArgs& a, ConvTables& tb, int nphase, const PhaseSpec* phs, int BL, int KC,
                         bool exact = false) {
    // decide halo vs per-tap segments per phase; compute LDS pitch and offsets
    int segw = 0;
    bool halo[VFX_MAXPH];
    for (int p = 0; p < nphase; ++p) {
        int mn = phs[p].taps[0].off, mx = mn;
        for (int t = 1; t < phs[p].ntaps; ++t) {
            mn = phs[p].taps[t].off < mn ? phs[p].taps[t].off : mn;
            mx = phs[p].taps[t].off > mx ? phs[p].taps[t].off : mx;
        }
        const int span = mx - mn;
        halo[p] = (long long)BL + span + 3 <= (long long)phs[p].ntaps * (BL + 3);
        int need;
        if (halo[p]) {
            need = exact ? BL : (mn - floor4(mn)) + BL + span;
        } else {
            need = exact ? BL : BL + 3;
        }
        need = (need + 3) & ~3;
        segw = need > segw ? need : segw;
    }
    a.segw = segw;
    int xs = 0;
    for (int p = 0; p < nphase; ++p) {
        PhaseTab& T = tb.ph[p];
        T.ntaps = phs[p].ntaps;
        T.ooff = phs[p].ooff;
        if (halo[p]) {
            int mn = phs[p].taps[0].off;
            for (int t = 1; t < phs[p].ntaps; ++t) mn = phs[p].taps[t].off < mn ? phs[p].taps[t].off : mn;
            const int org = exact ? mn : floor4(mn);  // exact: unaligned origin (interior instance, buffer loads)
            T.nseg = 1;
            T.seg_org[0] = org;
            for (int t = 0; t < phs[p].ntaps; ++t) {
                T.tap_lds[t] = phs[p].taps[t].off - org;
                T.tap_w[t] = phs[p].taps[t].slab;
            }
        } else {
            T.nseg = phs[p].ntaps;
            for (int t = 0; t < phs[p].ntaps; ++t) {
                const int org = exact ? phs[p].taps[t].off : floor4(phs[p].taps[t].off);
                T.seg_org[t] = org;
                T.tap_lds[t] = t * KC * segw + (phs[p].taps[t].off - org);
                T.tap_w[t] = phs[p].taps[t].slab;
            }
        }
        const int fl = T.nseg * KC * segw;
        xs = fl > xs ? fl : xs;
    }
    a.xs_floats = xs;
    return VFX_OK;
}



// split-K workspace: one growing device buffer per stream (allocated during warm-up; a launch that fits reuses it)
static float* splitk_workspace(hipStream_t s, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> pool;  // (device, stream): the null stream exists per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = pool[std::make_pair(dev, s)];
    if (e.second < bytes) {
        // A block that was handed out once is never freed or moved: a captured HIP graph (Pipeline.enable_graphs)
        // bakes the workspace pointer of its capture stream into its kernel nodes, and torch recycles stream handles
        // (round-robin pool of 32), so a later, larger request on the same handle must not invalidate that pointer.
        // Outgrown blocks are retired, not released (split-K only serves launches of <= 192 workgroups: the blocks are
        // a few MB, and their number is bounded by the distinct sizes a process asks for).
        static std::vector<float*> retired;
        if (e.first) retired.push_back(e.first);
        e.first = nullptr;
        e.second = 0;
        const size_t want = bytes + bytes / 2;
        if (hipMalloc((void**)&e.first, want) != hipSuccess) { e.first = nullptr; return nullptr; }
        e.second = want;
    }
    return e.first;
}

// ---- bf16x3 launch path (opt-in per launch, vfx_act.math == VFX_MATH_BF16X3).  Returns VFX_ENOTSUP when the
// geometry is outside what conv_x3_kernel covers; the caller then runs the fp32 kernel.
#define VFX_ENOTSUP (-100)

#include "vfx_convw.inc"
#include "vfx_convwg.inc"
#include "vfx_convwg4p.inc"
#include "vfx_convwg4x.inc"
#include "vfx_convtw.inc"
#include "vfx_convwg2d.inc"
#include "vfx_resblk4.inc"

template <int BM, int BL, int WGM, int WGL, int NT, int MODE, int ROWS = 1>
static int launch_x3_one(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static unsigned long long attr_set = 0;   // bit d: the attribute is set on device d (one process may drive several)
    int attr_dev = 0;
    if (hipGetDevice(&attr_dev) != hipSuccess) attr_dev = 0;
    auto kern = conv_x3_kernel<BM, BL, WGM, WGL, NT, MODE, ROWS>;
    if (!((attr_set >> (attr_dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set |= 1ull << (attr_dev & 63);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    VFX_LAUNCHED();
    return vfx_last_error();
}

template <int BM, int BL, int WGM, int WGL>
static int launch_x3_tile(const ConvArgs& a, int nt, int mode, int rows, dim3 grid, size_t lds, hipStream_t s) {
    if (rows == 3 && nt == 3) return launch_x3_one<BM, BL, WGM, WGL, 3, 0, 3>(a, grid, lds, s);
    if (rows == 3) return launch_x3_one<BM, BL, WGM, WGL, 1, 0, 3>(a, grid, lds, s);
#define VFX_X3(NT_)                                                                                  \
    if (nt == NT_)                                                                                   \
        return mode == 2   ? launch_x3_one<BM, BL, WGM, WGL, NT_, 2>(a, grid, lds, s)                \
               : mode == 1 ? launch_x3_one<BM, BL, WGM, WGL, NT_, 1>(a, grid, lds, s)                \
                           : launch_x3_one<BM, BL, WGM, WGL, NT_, 0>(a, grid, lds, s);
    VFX_X3(1)
    VFX_X3(2)
    VFX_X3(3)
#undef VFX_X3
    return VFX_ENOTSUP;
}

static int try_launch_x3(ConvArgs a, const vfx_tensor* x, int nphase, const PhaseSpec* phs_in, const void* w3,
                         hipStream_t stream) {
    const int Cin = a.Cin, Cout = a.Cout, B = a.B, Lq = a.Lq, Lin = a.Lin;
    if (!w3 || !vfx_aligned16(w3) || Cin % 32 != 0 || Cout % 32 != 0) return VFX_ENOTSUP;
    if (a.pad_mode == VFX_PAD_REFLECT || a.x_rows) return VFX_ENOTSUP;  // (no per-row lengths in the bf16x3 kernel)
    // (raw buffer resource, 32-bit byte offsets within one batch item: see launch_conv)
    if (((long long)a.CinPad * x->cstride + Lin + 2 * x->guard) * 4 >= (1ll << 31) - (1ll << 20)) return VFX_ENOTSUP;
    // 3x3 on a pitch map (9 taps (ky-1)*P + (kx-1), slab ky*3+kx) -> 3 kernel rows x the 3-tap dx case
    int rows = 1, P = 0;
    PhaseSpec row_spec;
    const PhaseSpec* phs = phs_in;
    if (nphase == 1 && phs_in[0].ntaps == 9) {
        P = phs_in[0].taps[3].off - phs_in[0].taps[0].off;
        for (int t = 0; t < 9; ++t)
            if (phs_in[0].taps[t].off != (t / 3 - 1) * P + (t % 3 - 1) || phs_in[0].taps[t].slab != t) return VFX_ENOTSUP;
        if (P < 4) return VFX_ENOTSUP;
        rows = 3;
        row_spec.ntaps = 3;
        row_spec.ooff = phs_in[0].ooff;
        for (int t = 0; t < 3; ++t) { row_spec.taps[t].off = t - 1; row_spec.taps[t].slab = t; }
        phs = &row_spec;
    }
    int wrow_slabs = 3;  // slabs per "row" step
    static const bool tapsplit_all = VFX_DEV_ENV("VFX_X3_TAPSPLIT") && atoi(VFX_DEV_ENV("VFX_X3_TAPSPLIT")) != 0;  // development
    if (rows == 1 && nphase == 1 && phs[0].ntaps == 3 && phs[0].taps[0].slab == 0 && phs[0].taps[1].slab == 1 &&
        phs[0].taps[2].slab == 2 && phs[0].taps[1].off == 0 && phs[0].taps[0].off == -phs[0].taps[2].off &&
        (phs[0].taps[2].off > 60 || (tapsplit_all && phs[0].taps[2].off > 0))) {
        // widely dilated k3: one tap per K-step (see the kernel comment)
        rows = 3;
        P = phs[0].taps[2].off;
        wrow_slabs = 1;
        row_spec.ntaps = 1;
        row_spec.ooff = phs[0].ooff;
        row_spec.taps[0].off = 0;
        row_spec.taps[0].slab = 0;
        phs = &row_spec;
    }
    const int nt = phs[0].ntaps;
    if (nt < 1 || nt > 3) return VFX_ENOTSUP;
    int span = 0, seg_lo = 0x7fffffff, seg_hi = -0x7fffffff;
    for (int p = 0; p < nphase; ++p) {
        if (phs[p].ntaps != nt) return VFX_ENOTSUP;
        int mn = phs[p].taps[0].off, mx = mn;
        for (int t = 1; t < nt; ++t) {
            mn = phs[p].taps[t].off < mn ? phs[p].taps[t].off : mn;
            mx = phs[p].taps[t].off > mx ? phs[p].taps[t].off : mx;
        }
        span = mx - mn > span ? mx - mn : span;
        seg_lo = mn < seg_lo ? mn : seg_lo;
        seg_hi = mx > seg_hi ? mx : seg_hi;
    }
    // staging mode: 0 = halo inside the BL staged columns (tile of BL - span outputs, small spans),
    // 1 = BL + span columns (one more staging slot), 2 = one segment per tap (large dilations)
    const int mode = span <= 8 ? 0 : (span <= 120 ? 1 : 2);
    const bool seg = mode == 2;
    int BM, BL;
    if (Cout % 128 == 0) { BM = 128; BL = 128; }
    else if (Cout % 64 == 0) {
        static const bool bl128 = VFX_DEV_ENV("VFX_X3_BL128") && atoi(VFX_DEV_ENV("VFX_X3_BL128")) != 0;  // development
        if (!seg && Lq >= 4096 && !bl128) { BM = 64; BL = 256; }
        else { BM = 64; BL = 128; }
    } else {
        if (seg) return VFX_ENOTSUP;
        BM = 32; BL = 256;
    }
    const int segw = mode == 1 ? BL + span : BL;
    const int bl_eff = mode == 0 ? BL - span : BL;
    ConvTables tb;
    std::memset(&tb, 0, sizeof(tb));
    for (int p = 0; p < nphase; ++p) {
        PhaseTab& T = tb.ph[p];
        T.ntaps = nt;
        T.ooff = phs[p].ooff;
        int mn = phs[p].taps[0].off;
        for (int t = 1; t < nt; ++t) mn = phs[p].taps[t].off < mn ? phs[p].taps[t].off : mn;
        T.nseg = seg ? nt : 1;
        for (int t = 0; t < nt; ++t) {
            if (seg) { T.seg_org[t] = phs[p].taps[t].off; T.tap_lds[t] = t * segw; }
            else { T.seg_org[0] = mn; T.tap_lds[t] = phs[p].taps[t].off - mn; }
            T.tap_w[t] = phs[p].taps[t].slab;
        }
    }
    // every tile must be interior with respect to the guard band (loads inside [-guard, Lin + guard))
    const int ntiles = (Lq + bl_eff - 1) / bl_eff;
    const long long g = a.x_guard;
    const int rowspan = rows == 3 ? P : 0;
    if (-(long long)seg_lo + rowspan > g) return VFX_ENOTSUP;
    if ((long long)(ntiles - 1) * bl_eff + seg_hi + BL + rowspan > (long long)Lin + g) return VFX_ENOTSUP;
    a.tab3 = device_tables(tb);
    if (!a.tab3) return VFX_EINVAL;
    const int nchunks = Cin / 16;
    a.w3 = w3;
    a.segw3 = segw;
    a.bl3 = bl_eff;
    a.xrow3 = P * 4;
    a.wrow3 = wrow_slabs * nchunks * (4 * Cout * 16);  // one kernel row = 3 slabs (one tap = 1 slab) of [Cin/16] chunks
    a.xplane3 = (seg ? nt : 1) * segw * 32;
    a.buf3 = 2 * a.xplane3 + nt * 2 * BM * 32;
    a.tile_lo = 0;
    a.tile_hi = ntiles;
    const size_t lds = 2ull * a.buf3 + (a.pre_act == VFX_PRE_AFFINE_LRELU ? (size_t)Cin * 8 : 0);
    if (lds > 160 * 1024) return VFX_ENOTSUP;
    // few workgroups and a long K (deep UNet levels, first upsampling stage at small batch): the fp32 kernel
    // (three to four workgroups per CU, split-K for plain maps) is faster than this instance running a long
    // serial K loop on a fraction of the chip (measured at batch 1: 23.5 -> 20.4 ms per 10 s utterance)
    if ((long long)ntiles * (nphase * Cout / BM) * B <= 96 && Cin * (rows == 3 && nt == 3 ? 9 : nt) >= 1024)
        return VFX_ENOTSUP;
    const dim3 grid(ntiles, nphase * Cout / BM, B);
    g_last_tile = BM * 100000 + BL * 100 + 16;
    (void)x;
    if (BM == 128) return launch_x3_tile<128, 128, 2, 2>(a, nt, mode, rows, grid, lds, stream);
    if (BM == 32) {
        if (rows == 3 && nt == 3) return launch_x3_one<32, 256, 1, 4, 3, 0, 3>(a, grid, lds, stream);
        if (rows == 3) return VFX_ENOTSUP;
        if (nt == 1 && mode == 0) return launch_x3_one<32, 256, 1, 4, 1, 0, 1>(a, grid, lds, stream);
        return VFX_ENOTSUP;
    }
    if (BL == 256) return launch_x3_tile<64, 256, 1, 4>(a, nt, mode, rows, grid, lds, stream);
    return launch_x3_tile<64, 128, 1, 4>(a, nt, mode, rows, grid, lds, stream);
}

static int launch_conv(const vfx_tensor* x, const float* w, const float* bias, const vfx_tensor* res,
                       const vfx_tensor* y, int B, int Cin, int Cout, int Lin, int Lq, int Lout,
                       int nphase, const PhaseSpec* phs, int q_shift, int q_mask, int o_rs, int o_cs,
                       int pad_mode, const vfx_act* act, int in_mask, int out_mask, hipStream_t stream) {
    if (!x || !y || !w || !x->ptr || !y->ptr || B <= 0 || Cin <= 0 || Cout <= 0 || Lin <= 0 || Lq <= 0)
        return VFX_EINVAL;
    if (Cout % 32 != 0 || nphase < 1 || nphase > VFX_MAXPH) return VFX_EINVAL;
    if (x->lstride != 1 || (x->cstride & 3) || (x->bstride & 3) || !vfx_aligned16(x->ptr) || !vfx_aligned16(w))
        return VFX_EALIGN;
    if (pad_mode == VFX_PAD_REFLECT && Lin < 8) return VFX_EINVAL;
    if (B > 65535) return VFX_EINVAL;

    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = (const float*)x->ptr;
    a.w = w;
    a.bias = bias;
    a.res = res ? (const float*)res->ptr : nullptr;
    a.y = (float*)y->ptr;
    a.B = B; a.Cin = Cin; a.CinPad = (Cin + 7) & ~7; a.Cout = Cout;
    a.Lin = Lin; a.Lq = Lq; a.Lout = Lout;
    a.x_bs = x->bstride; a.x_cs = x->cstride;
    a.x_guard = (x->guard > 0 && pad_mode != VFX_PAD_REFLECT) ? (int)x->guard : 0;
    a.x_rows = x->rows;
    a.lq_extra = Lq - Lin;
    a.y_bs = y->bstride; a.y_cs = y->cstride; a.y_ls = y->lstride;
    if (res) { a.r_bs = res->bstride; a.r_cs = res->cstride; a.r_ls = res->lstride; }
    a.q_shift = q_shift; a.q_mask = q_mask; a.o_rs = o_rs; a.o_cs = o_cs;
    a.res_init = res && nphase == 1 && q_shift == 31 && o_rs == 0 && o_cs == 1 && phs[0].ooff == 0 && Lq <= Lout &&
                 (long long)Cout * res->cstride + (long long)Lq * res->lstride < (1ll << 31);
    a.pad_mode = pad_mode;
    a.pre_act = act ? act->pre_act : VFX_PRE_NONE;
    a.pre_slope = act ? act->pre_slope : 0.f;
    a.pre_scale = act ? act->pre_scale : nullptr;
    a.pre_shift = act ? act->pre_shift : nullptr;
    a.post_act = act ? act->post_act : VFX_POST_NONE;
    a.post_slope = act ? act->post_slope : 0.f;
    if (a.pre_act == VFX_PRE_AFFINE_LRELU && (!a.pre_scale || !a.pre_shift)) return VFX_EINVAL;
    a.in_mask = in_mask; a.out_mask = out_mask;

    int maxnt = 0;
    for (int p = 0; p < nphase; ++p) maxnt = phs[p].ntaps > maxnt ? phs[p].ntaps : maxnt;
    if (maxnt < 1 || maxnt > VFX_MAXT) return VFX_EINVAL;
    if (act && act->math == VFX_MATH_BF16X3) {
        const int rc3 = try_launch_x3(a, x, nphase, phs, act->w_x3, stream);
        if (rc3 != VFX_ENOTSUP) return rc3;
    }
    if (act && act->w_wino4) {
        int rc4 = VFX_ENOTSUP;
        if (nphase == 1 && phs[0].ntaps == 9 && in_mask > 0) {
            const int P = in_mask + 1;
            bool ok = true;
            for (int t = 0; t < 9; ++t) ok &= phs[0].taps[t].off == (t / 3 - 1) * P + (t % 3 - 1) && phs[0].taps[t].slab == t;
            if (ok) rc4 = try_launch_convwg4s_2d(a, x, P, act->w_wino4, stream);
        } else {
            rc4 = try_launch_convwg4(a, x, nphase, phs, act->w_wino4, stream);
        }
        if (rc4 != VFX_ENOTSUP) return rc4;
    }
    if (act && act->w_direct) {
        int rcw = VFX_ENOTSUP;
        if (nphase == 1 && phs[0].ntaps == 9 && in_mask > 0) {
            // 3x3 on a pitch map: taps (ky-1)*P + (kx-1), slab ky*3+kx
            const int P = in_mask + 1;
            bool ok = true;
            for (int t = 0; t < 9; ++t) ok &= phs[0].taps[t].off == (t / 3 - 1) * P + (t % 3 - 1) && phs[0].taps[t].slab == t;
            if (ok) rcw = try_launch_convw_2d(a, x, P, act->w_direct, stream);
        } else {
            rcw = try_launch_convw(a, x, nphase, phs, act->w_direct, stream);
        }
        if (rcw != VFX_ENOTSUP) return rcw;
    }
    // tile choice: maximise (tile efficiency) x (tail efficiency along L) x (wave quantisation)
    int best = -1;
    float best_score = -1.f;
    for (int i = 0; i < kNumTiles; ++i) {
        const TileCfg& t = kTiles[i];
        if (Cout % t.BM) continue;
        const long long lt = (Lq + t.BL - 1) / t.BL;
        const long long nwg = lt * (long long)(nphase * Cout / t.BM) * B;
        const float tail = (float)Lq / (float)(lt * t.BL);
        const long long slots = 512;  // 2 workgroups per CU x 256 CUs
        const float quant = (float)nwg / (float)(((nwg + slots - 1) / slots) * slots);
        const float score = t.util * tail * quant;
        if (score > best_score * 1.02f) { best_score = score; best = i; }
    }
    if (best < 0) return VFX_EINVAL;
    const TileCfg& tc = kTiles[best];

    // K-chunk depth: 8 channels per chunk unless the staged tiles would need more than 4 float4 per
    // thread (activations) / the weight tile more than its slot budget -> 4 channels per chunk
    // development switch: VFX_WAVES8=1 runs the two big tiles with 8 waves (wave tile 32x64)
    static const bool waves8_env = VFX_DEV_ENV("VFX_WAVES8") && atoi(VFX_DEV_ENV("VFX_WAVES8")) != 0;
    const bool waves8 = waves8_env && ((tc.BM == 128 && tc.BL == 128) || (tc.BM == 64 && tc.BL == 256));
    const int nthr = waves8 ? 512 : 256;
    // development switch: VFX_KC16=1 lets 3-tap (or fewer) launches on the 128x128 tile use 16-channel chunks
    static const bool kc16_env = VFX_DEV_ENV("VFX_KC16") && atoi(VFX_DEV_ENV("VFX_KC16")) != 0;
    // Exact-width halo tiles: when the taps of every phase span only a few positions (dilation <= 3, transposed
    // convolutions, k = 1) the halo is staged INSIDE the BL columns and a tile produces BL - span outputs.  The
    // activation tile then needs exactly one staging slot per thread (half the loads, conversions and LDS writes
    // of the BL + halo layout) for <= 6 % idle MFMA columns.  Needs unaligned 16-byte loads, i.e. the interior
    // instance on every tile (guard band), and no pad-column mask (its fast path assumes aligned vectors).
    // Launches whose phases all stage one segment per tap (wide dilations) get exact BL-wide segments the same
    // way (no alignment slack: 3 slots instead of 4), at no cost in columns.
    int maxspan = 0;
    bool all_segments = true;
    for (int p = 0; p < nphase; ++p) {
        int mn = phs[p].taps[0].off, mx = mn;
        for (int t = 1; t < phs[p].ntaps; ++t) {
            mn = phs[p].taps[t].off < mn ? phs[p].taps[t].off : mn;
            mx = phs[p].taps[t].off > mx ? phs[p].taps[t].off : mx;
        }
        maxspan = mx - mn > maxspan ? mx - mn : maxspan;
        all_segments &= !((long long)tc.BL + (mx - mn) + 3 <= (long long)phs[p].ntaps * (tc.BL + 3));
    }
    static const int exact_off = VFX_DEV_ENV("VFX_NO_EXACT") ? atoi(VFX_DEV_ENV("VFX_NO_EXACT")) : 0;  // development: 1 all, 2 segments
    bool exact = exact_off != 1 && x->guard > 0 && pad_mode != VFX_PAD_REFLECT && in_mask == 0 && !waves8 &&
                 ((maxspan > 0 && maxspan <= 8) || (all_segments && exact_off != 2));
    ConvTables tb;
    int KC, ntiles;
    size_t lds;
    for (;;) {
        KC = (kc16_env && maxnt <= 3 && tc.BM == 128 && tc.BL == 128 && !waves8 && Cin % 16 == 0) ? 16 : 8;
        for (;;) {
            std::memset(&tb, 0, sizeof(tb));
            int rc = fill_segments(a, tb, nphase, phs, tc.BL, KC, exact);
            if (rc) return rc;
            int maxseg = 0;
            for (int p = 0; p < nphase; ++p) maxseg = tb.ph[p].nseg > maxseg ? tb.ph[p].nseg : maxseg;
            const bool xfit = (long long)maxseg * KC * (a.segw / 4) <= 4 * nthr;
            a.nxv = (int)(((long long)maxseg * KC * (a.segw / 4) + nthr - 1) / nthr);
            const bool wfit = (long long)maxnt * KC * tc.BM <= (KC == 16 ? 6 : (KC == 8 ? 4 : 5)) * 4 * nthr;
            if (xfit && wfit) break;
            if (KC == 4) return VFX_ERANGE;
            KC = KC == 16 ? 8 : 4;
        }
        a.ws_floats = maxnt * KC * tc.BM;
        lds = (2ull * (a.xs_floats + a.ws_floats) + 2ull * a.CinPad) * sizeof(float);
        {
            // development knob: VFX_LDS_MIN_KB pads the request to limit workgroups per CU (occupancy studies)
            static const long pad_kb = VFX_DEV_ENV("VFX_LDS_MIN_KB") ? atol(VFX_DEV_ENV("VFX_LDS_MIN_KB")) : 0;
            if (pad_kb > 0 && lds < (size_t)pad_kb * 1024) lds = (size_t)pad_kb * 1024;
        }
        if (lds > 160 * 1024) return VFX_ERANGE;

        // interior tiles: all staged vectors of all phases inside [0, Lin) and no channel tail
        a.bl_step = (exact && !all_segments) ? tc.BL - maxspan : tc.BL;
        ntiles = (Lq + a.bl_step - 1) / a.bl_step;
        int seg_lo = 0x7fffffff, seg_hi = -0x7fffffff;
        for (int p = 0; p < nphase; ++p)
            for (int sg = 0; sg < tb.ph[p].nseg; ++sg) {
                seg_lo = tb.ph[p].seg_org[sg] < seg_lo ? tb.ph[p].seg_org[sg] : seg_lo;
                seg_hi = tb.ph[p].seg_org[sg] > seg_hi ? tb.ph[p].seg_org[sg] : seg_hi;
            }
        // loads are safe inside [-guard, Lin + guard); values outside [0, Lin) are masked in the kernel
        const long long g = (x->guard > 0 && pad_mode != VFX_PAD_REFLECT) ? x->guard : 0;
        const long long need_lo = -(long long)seg_lo - g;  // q0 >= need_lo
        int tlo = need_lo > 0 ? (int)((need_lo + a.bl_step - 1) / a.bl_step) : 0;
        const long long room = (long long)Lin + g - seg_hi - a.segw;
        int thi = room >= 0 ? (int)(room / a.bl_step) + 1 : 0;
        if (thi > ntiles) thi = ntiles;
        if (tlo > thi || Cin % KC != 0) { tlo = 0; thi = 0; }
        // the interior instance reads one batch item through a raw buffer resource with 32-bit byte offsets
        // (num_records 2^31 - 1: anything beyond reads as zero): rows of a batch item spanning 2 GB or more (a
        // 64-channel stage of > 3 minutes) run the general, pointer-addressed instance on every tile
        if (((long long)a.CinPad * x->cstride + Lin + 2 * g) * 4 >= (1ll << 31) - (1ll << 20)) { tlo = 0; thi = 0; }
        // ragged batches: a tile that is interior for the longest row holds the END of a shorter one, and only the
        // general instance mirrors there (the interior one can only zero what lies outside the row)
        if (a.x_rows && pad_mode == VFX_PAD_REFLECT) { tlo = 0; thi = 0; }
        a.tile_lo = tlo;
        a.tile_hi = thi;
        if (exact && !(tlo == 0 && thi == ntiles)) { exact = false; continue; }  // boundary tiles: aligned layout
        break;
    }
    a.tab = device_tables(tb);
    if (!a.tab) return VFX_EINVAL;
    g_last_tile = tc.BM * 100000 + tc.BL * 100 + KC;
    // split-K: a launch with few workgroups and a long K (the deep UNet levels at small batch: 48 workgroups x 864
    // serial K-chunks) is cut along K so that the whole chip works on it; plain output maps only
    a.ksplit = 1;
    {
        static const bool splitk_off = VFX_DEV_ENV("VFX_NO_SPLITK") && atoi(VFX_DEV_ENV("VFX_NO_SPLITK")) != 0;  // development
        const int nchunks = (Cin + KC - 1) / KC;
        const long long nwg = (long long)ntiles * (nphase * Cout / tc.BM) * B;
        const bool plain_map = nphase == 1 && q_shift == 31 && o_rs == 0 && o_cs == 1 && phs[0].ooff == 0 && Lq <= Lout;
        if (!splitk_off && plain_map && nwg <= 192 && nchunks >= 32) {
            int want = (int)(512 / nwg);
            if (want > 8) want = 8;
            if (want > nchunks / 16) want = nchunks / 16;
            if (want >= 2 && (long long)B * want <= 65535) {
                a.cpp = (nchunks + want - 1) / want;
                a.ksplit = (nchunks + a.cpp - 1) / a.cpp;
                a.ws_ls = Lq;
                a.ws = splitk_workspace(stream, (size_t)a.ksplit * B * Cout * Lq * sizeof(float));
                if (!a.ws) a.ksplit = 1;
            }
        }
    }
    {
        // tiles per FAST workgroup: as many as keeps >= ~1024 workgroups in flight, at most 8
        const long long nwg1 = (long long)(a.tile_hi - a.tile_lo) * (nphase * Cout / tc.BM) * B;
        (void)nwg1;
        a.tpw = 1;  // multi-tile workgroups measured slower (register pressure), see the kernel comment
    }
    const int gy = nphase * Cout / tc.BM;
    const int gz = B * a.ksplit;
    int rc = VFX_EINVAL;
    if (KC == 16) rc = launch_cfg<128, 128, 2, 2, 16>(a, ntiles, gy, gz, lds, stream);
#define VFX_CASE(BM_, BL_, WGM_, WGL_)                                                          \
    else if (tc.BM == BM_ && tc.BL == BL_)                                                      \
        rc = KC == 8 ? launch_cfg<BM_, BL_, WGM_, WGL_, 8>(a, ntiles, gy, gz, lds, stream)      \
                     : launch_cfg<BM_, BL_, WGM_, WGL_, 4>(a, ntiles, gy, gz, lds, stream);
    else if (waves8 && tc.BM == 128 && tc.BL == 128)
        rc = KC == 8 ? launch_cfg<128, 128, 4, 2, 8>(a, ntiles, gy, gz, lds, stream)
                     : launch_cfg<128, 128, 4, 2, 4>(a, ntiles, gy, gz, lds, stream);
    else if (waves8 && tc.BM == 64 && tc.BL == 256)
        rc = KC == 8 ? launch_cfg<64, 256, 2, 4, 8>(a, ntiles, gy, gz, lds, stream)
                     : launch_cfg<64, 256, 2, 4, 4>(a, ntiles, gy, gz, lds, stream);
    VFX_CASE(128, 128, 2, 2)
    VFX_CASE(64, 256, 1, 4)
    VFX_CASE(128, 64, 4, 1)
    VFX_CASE(32, 256, 1, 4)
    VFX_CASE(64, 64, 2, 2)
    VFX_CASE(32, 128, 1, 4)
#undef VFX_CASE
    if (rc != VFX_OK || a.ksplit == 1) return rc;
    const long long total = (long long)B * Cout * Lq;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a.ws, a.ksplit, a.ws_ls, a.bias, a.res,
                       a.r_bs, a.r_cs, a.r_ls, a.y, a.y_bs, a.y_cs, a.y_ls, B, Cout, Lq, a.post_act, a.post_slope,
                       a.out_mask);
    VFX_LAUNCHED();
    return vfx_last_error();
}

// --------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------
extern "C" int vfx_conv1d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                              const vfx_tensor* res, const vfx_tensor* y, int B, int Cin, int Cout,
                              int L, int k, int dilation, int pad_mode, const vfx_act* act,
                              vfx_stream_t stream) {
    if (k < 1 || k > VFX_MAXT || !(k & 1) || dilation < 1) return VFX_EINVAL;
    PhaseSpec ph;
    ph.ntaps = k;
    ph.ooff = 0;
    for (int t = 0; t < k; ++t) {
        ph.taps[t].off = (t - (k - 1) / 2) * dilation;
        ph.taps[t].slab = t;
    }
    return launch_conv(x, w_packed, bias, res, y, B, Cin, Cout, L, L, L, 1, &ph, 31, 0x7fffffff, 0, 1,
                       pad_mode, act, 0, 0, (hipStream_t)stream);
}

extern "C" int vfx_convtr1d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                                const vfx_tensor* y, int B, int Cin, int Cout, int Lin, int stride,
                                const vfx_act* act, vfx_stream_t stream) {
    // out[o] = x[q]*w[r] + x[q-1]*w[r+s],  u = o + pad = q*s + r   
    if (stride < 1 || stride > VFX_MAXPH) return VFX_EINVAL;
    if (act && act->w_wino4) {                    // the Winograd F(3,2) form (vfx_convtw.inc): 2/3 of the direct sum's MFMAs
        const int rc = try_launch_convtw(x, act->w_wino4, bias, y, B, Cin, Cout, Lin, stride, act, (hipStream_t)stream);
        if (rc != VFX_ENOTSUP) return rc;
    }
    const int pad = stride / 2 + stride % 2;
    PhaseSpec ph[VFX_MAXPH];
    for (int r = 0; r < stride; ++r) {
        ph[r].ntaps = 2;
        ph[r].taps[0].off = 0;
        ph[r].taps[0].slab = r;
        ph[r].taps[1].off = -1;
        ph[r].taps[1].slab = r + stride;
        ph[r].ooff = r - pad;
    }
    return launch_conv(x, w_packed, bias, nullptr, y, B, Cin, Cout, Lin, Lin + 1, stride * Lin, stride, ph,
                       31, 0x7fffffff, 0, stride, VFX_PAD_ZERO, act, 0, 0, (hipStream_t)stream);
}

extern "C" int vfx_conv2d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                              const vfx_tensor* res, const vfx_tensor* y, int B, int Cin, int Cout, int H,
                              int pitch_log2, int ksize, const vfx_act* act, vfx_stream_t stream) {
    if ((ksize != 1 && ksize != 3) || pitch_log2 < 1 || pitch_log2 > 12 || H < 1) return VFX_EINVAL;
    const int P = 1 << pitch_log2;
    PhaseSpec ph;
    ph.ooff = 0;
    ph.ntaps = ksize * ksize;
    for (int ky = 0; ky < ksize; ++ky)
        for (int kx = 0; kx < ksize; ++kx) {
            ph.taps[ky * ksize + kx].off = (ky - ksize / 2) * P + (kx - ksize / 2);
            ph.taps[ky * ksize + kx].slab = ky * ksize + kx;
        }
    const int L = H * P;
    return launch_conv(x, w_packed, bias, res, y, B, Cin, Cout, L, L, L, 1, &ph, 31, 0x7fffffff, 0, 1,
                       VFX_PAD_ZERO, act, P - 1, P - 1, (hipStream_t)stream);
}

extern "C" int vfx_convtr2d_3x3s2_f32(const vfx_tensor* x, const float* w_packed, const vfx_tensor* y, int B,
                                      int Cin, int Cout, int h, int in_pitch_log2, const vfx_act* act,
                                      vfx_stream_t stream) {
    // oy = 2a + py, ox = 2b + px; even parity has taps k=0 (i = a) and k=2 (i = a-1), odd has k=1.
    if (in_pitch_log2 < 1 || in_pitch_log2 > 11 || h < 1) return VFX_EINVAL;
    const int Pi = 1 << in_pitch_log2, Po = 2 * Pi;
    PhaseSpec ph[4];
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            PhaseSpec& p = ph[py * 2 + px];
            p.ntaps = 0;
            p.ooff = py * Po + px;
            const int nky = py == 0 ? 2 : 1, nkx = px == 0 ? 2 : 1;
            for (int iy = 0; iy < nky; ++iy)
                for (int ix = 0; ix < nkx; ++ix) {
                    const int ky = py == 0 ? 2 * iy : 1, kx = px == 0 ? 2 * ix : 1;
                    const int dy = py == 0 ? iy : 0, dx = px == 0 ? ix : 0;  // input shift (a-dy, b-dx)
                    p.taps[p.ntaps].off = -(dy * Pi + dx);
                    p.taps[p.ntaps].slab = ky * 3 + kx;
                    ++p.ntaps;
                }
        }
    const int Lin = h * Pi;
    // q = a*Pi + b  ->  out = a*(2*Po) + b*2 + ooff
    return launch_conv(x, w_packed, nullptr, nullptr, y, B, Cin, Cout, Lin, Lin, 2 * h * Po, 4, ph,
                       in_pitch_log2, Pi - 1, 2 * Po, 2, VFX_PAD_ZERO, act, Pi - 1, Po - 1,
                       (hipStream_t)stream);
}
