/*
 * vfx_hip.h -- C ABI of libvfx_hip.so, the MI355X (gfx950) kernel library behind
 * voicefixer_amd.VoiceFixer.restore()/restore_inmem() and Vocoder.forward()/oracle().
 *
 * The reference (haoheliu/voicefixer) has no FFI: its hot path is a chain of stock torch
 * operators.  Each entry point below replaces the torch operator calls named in its
 * comment (paths relative to the reference root).  What a reference maintainer would bind
 * is shown in INTEGRATION.md (ctypes stubs).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's
 *     data_ptr()), float32 unless stated, never retained after the call returns;
 *   - activations are channel-major: element (b, c, l) of a vfx_tensor lives at
 *     ptr[b*bstride + c*cstride + l*lstride] (strides in ELEMENTS). Inputs of the conv
 *     family need lstride == 1, cstride % 4 == 0 and a 16-byte aligned ptr;
 *   - 2-D feature maps (B,C,H,W) are stored flattened with a power-of-two row pitch
 *     P = W+1: element (h, w) at l = h*P + w; column P-1 is a structural zero that serves
 *     as left/right zero padding for 3x3 convolutions (the UNet works on W = 127, 63, ...
 *     1 mel bins, so P = 128, 64, ... 2);
 *   - all launches go to `stream` (a hipStream_t passed as void*), no internal
 *     synchronisation, no allocation; thread-safe for distinct streams;
 *   - return 0 on success, a negative VFX_E* code for bad arguments, or a positive
 *     hipError_t from the launch.  Nothing throws.
 */
#ifndef VFX_HIP_H
#define VFX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vfx_stream_t; /* hipStream_t */

#define VFX_OK 0
#define VFX_EINVAL (-1)   /* bad argument / unsupported shape */
#define VFX_EALIGN (-2)   /* pointer or stride alignment */
#define VFX_ERANGE (-3)   /* tile bookkeeping overflow (tap span too large for the tile) */

typedef struct {
    void* ptr;
    int64_t bstride, cstride, lstride; /* element strides */
    int64_t guard; /* INPUT tensors of the conv family: number of elements that may be READ (contents
                      arbitrary, they are masked) before l = 0 and after l = L-1 of every (b, c) row.
                      With a guard >= the largest tap offset + 8 every tile takes the branch-free
                      interior kernel; 0 is always legal (boundary tiles then run the general kernel). */
    const int32_t* rows; /* ragged batches (may be NULL): device int32[B], the VALID length of batch item b of this
                      tensor (<= the L / H*P argument of the call, which is then the row capacity).  As an INPUT of the
                      conv family, positions >= rows[b] are zero padding of the activated input (reflect padding
                      mirrors at rows[b]); outputs past a row's valid length are unspecified.  Lets utterances of
                      different lengths share one launch with results identical to separate launches. */
} vfx_tensor;

/* pre-activation applied to the INPUT while it is staged into LDS */
#define VFX_PRE_NONE 0
#define VFX_PRE_LRELU 1         /* leaky_relu(x, pre_slope) */
#define VFX_PRE_AFFINE_LRELU 2  /* leaky_relu(x*pre_scale[c] + pre_shift[c], pre_slope): eval BatchNorm + (leaky)ReLU */
/* post-activation applied in the epilogue, after bias and residual */
#define VFX_POST_NONE 0
#define VFX_POST_LRELU 1        /* leaky_relu(., post_slope) */
#define VFX_POST_ELU 2
#define VFX_POST_TANH 3
#define VFX_POST_SIGMOID 4
#define VFX_POST_LRELU_SNAKE 5  /* v = leaky_relu(., post_slope); v + sin(v)  (UpsampleNet prologue fused upstream) */

/* arithmetic of the contraction (per launch) */
#define VFX_MATH_F32 0     /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation (default) */
#define VFX_MATH_BF16X3 1  /* opt-in: every fp32 operand split into two bf16 terms, x*w evaluated as
                              xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
                              (relative product error <= 2^-16); needs w_x3 and an input guard band
                              (vfx_tensor.guard >= largest tap offset + tile width), falls back to fp32
                              per launch for geometries the bf16x3 kernel does not cover (Cin % 32,
                              reflect padding, more than 3 taps other than 3x3, mixed tap counts) */

typedef struct {
    int pre_act;
    float pre_slope;
    const float* pre_scale; /* [Cin] for VFX_PRE_AFFINE_LRELU */
    const float* pre_shift; /* [Cin] */
    int post_act;
    float post_slope;
    int math;               /* VFX_MATH_* */
    const void* w_x3;       /* VFX_MATH_BF16X3: the packed weights as bf16 (hi, lo) planes, layout
                               [slab][Cin/16][plane][k-half][Cout][8] (voicefixer_amd/packing.py::pack_x3) */
    const float* w_direct;  /* optional (may be NULL): the same fp32 weights packed [slab][CinPad/8][Cout][8], the 8
                               channels of a group in the order 0,2,4,6,1,3,5,7 (packing.py::pack_direct).  Lets 1-D
                               launches with 2-3 taps, zero padding, a guard band and no BatchNorm pre-activation run
                               on convw_kernel (weights read from L2 as MFMA A-operand vectors, activations staged
                               16-32 channels deep); bit-identical results are NOT implied (other summation order
                               within fp32 rounding).  Everything else, and small launches, use w_packed. */
    const float* w_wino4;   /* optional (may be NULL), k = 3 stride-1 Conv1d: the Winograd F(4,3) transform of the three tap
                               slabs, six slabs U = G w packed like w_direct (packing.py::pack_wino4).  Large launches (>=
                               512 workgroups) with Cin % 32 == 0, Cout % 64 == 0, zero padding and no BatchNorm
                               pre-activation then run on convwg4_kernel: four outputs a dilation apart share six
                               products instead of twelve (2x fewer fp32 MFMAs than the direct sum; rounding ~3x the
                               direct sum's, ~1e-6 relative, relative to the LARGEST operand of a quad's six-tap window:
                               profiles/r03_winograd_error_sweep.txt).  For vfx_conv2d_f32 with ksize 3: the transform
                               along the kernel's ROW axis for each kernel column, 18 slabs kx*6 + plane
                               (packing.py::pack_wino4_2d); launches with Cout % 64 == 0 and Cin % 32 == 0, or Cout % 32
                               == 0 and Cin % 16 == 0, on maps of pitch <= 64 / 128 run on convwg4s_kernel (four
                               vertically adjacent outputs share 18 products instead of 36).  For vfx_convtr1d_f32: the
                               Winograd F(3,2) transform along the INPUT axis for every output phase r of the polyphase
                               form (out[q s + r - pad] = w[r + s] x[q - 1] + w[r] x[q]), 4 s slabs 4 r + plane
                               (packing.py::pack_wino32_tr); launches of >= 512 workgroups with Cin % 32 == 0, Cout % 64
                               == 0 and no activation run on convtw_kernel: three consecutive q of a phase share four
                               products instead of six (transform constants 1 and 1/2: rounding as the direct sum's). */
} vfx_act;

#define VFX_PAD_ZERO 0
#define VFX_PAD_REFLECT 1

int vfx_version(void);

/* 16 hex digits: sha256 over every source file this library was built from (csrc/Makefile: BUILD_ID).  Measurement
 * records (bench.py, profiles/ PMC summaries) carry it so that a profile can be tied to the library that produced it. */
const char* vfx_build_id(void);

/* Number of hipLaunchKernel calls issued by this library in this process (test hook:
 * proves the HIP path, not a fallback, produced a result). */
uint64_t vfx_launch_count(void);

/* Kernel family chosen by the most recent conv-family launch of the calling thread, encoded
 * BM*100000 + BL*100 + code (bench.py's roofline bookkeeping maps it to the template instances a profiler shows):
 *   code 4 / 8        conv_taps_kernel<BM,BL,*,*,KC=code,...>   (first-generation kernel: small launches, reflect
 *                     padding, 1x1 / transposed 2-D convolutions, split-K)
 *   code 16           conv_x3_kernel<BM,BL,...>                  (opt-in bf16x3 arithmetic)
 *   code 51 / 52 / 54 convw_kernel<BM,BL,*,*,NT=2|3,*,false>     (1-D, activation chunks of 8 / 16 / 32 channels)
 *   code 59           convw_kernel<BM,BL,*,*,NT=9,*,false>       (3x3 on a pitch map)
 *   code 61 / 62 / 64 convw_kernel<BM,BL,*,*,3,*,true>           (vfx_resblock_f32: direct sums, chunks of 8 / 16 / 32 channels)
 *   code 71 / 72 / 74 convw_kernel<BM,BL,*,*,3,*,2>              (vfx_resblock_f32 with w2_wino: second half as Winograd F(2,3))
 *   code 91 / 92 / 94 convw_kernel<BM,BL,*,*,3,*,3>              (vfx_resblock_f32 with w2_wino4: second half as Winograd F(4,3))
 *   code 96           resblk4_kernel                             (vfx_resblock_f32 with w1_wino4 + w2_wino4: both halves F(4,3))
 *   code 80           convwg4_kernel<..>                          (Winograd F(4,3), 1-D), BL = output positions
 *   code 81           convwg4p_kernel<..>                         (the same tile on persistent workgroups: the two long launches of a
 *                     ResStack layer, tap / staging / weight pipeline running across tiles)
 *   code 88           convwg4s_kernel<..>                         (Winograd F(4,3), 3x3 on a pitch map, kernel columns share one tile) */
int vfx_last_conv_tile(void);

/* ---- convolution family: implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------
 * Weights are pre-packed on the host as [slab][CinPad][Cout] (Cout contiguous), where a
 * slab is one kernel tap, CinPad = Cin rounded up to 8 (zero filled).  See
 * voicefixer_amd/packing.py.  bias may be NULL; res may be NULL.                      */

/* y[b,n,l] = post( bias[n] + res[b,n,l] + sum_{c,t} w[t][c][n] * pre(x[b,c,l+(t-(k-1)/2)*dil]) )
 * Replaces torch conv1d (+ leaky_relu / ELU / tanh / ReflectionPad1d around it) in
 * voicefixer/vocoder/model/generator.py:33-54,74-76,95-99 and modules.py:549-576
 * (ResStack), and torch linear in voicefixer/restorer/model.py:69-99 (k = 1).
 * Cout % 32 == 0; for Cout == 1 use vfx_conv1d_cout1_f32. */
int vfx_conv1d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                   const vfx_tensor* res, const vfx_tensor* y, int B, int Cin, int Cout, int L,
                   int k, int dilation, int pad_mode, const vfx_act* act, vfx_stream_t stream);

/* One whole ResStack layer, fused (voicefixer/vocoder/model/modules.py:592-609, one iteration of the loop):
 *   y[b,n,l] = post( x[b,n,l] + bias2[n] + conv_k3_d1( lrelu_s( bias1 + conv_k3_dil( lrelu_s(x) ) ) )[b,n,l] )
 * for C = 64 or 128 channels.  The intermediate tensor lives in LDS (one tile of columns per workgroup), never in HBM.
 * x needs a guard band >= dilation + 264 (vfx_tensor.guard) unless resblk4_kernel takes the launch; y must NOT alias x
 * (a tile reads the input columns of its neighbours).  post_act as vfx_act.post_act.
 * Weights: both convolutions in the w_direct layout of vfx_act (required), and optionally their Winograd transforms --
 * the library picks the cheapest arithmetic the operands allow, results differ by fp32 rounding only:
 *   w2_wino   F(2,3) of the SECOND convolution (U0 = w0, U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w2, four slabs packed like
 *             w_direct, packing.py::pack_wino): its dilation-1 half forms 4 products per output pair instead of 6 on the LDS
 *             tile (C = 64 with the 256-column tile, C = 128);
 *   w2_wino4  F(4,3) of the second convolution (six slabs U = G w as vfx_act.w_wino4, packing.py::pack_wino4), C = 64 with
 *             16-byte aligned rows of x and y: 6 products per output QUAD instead of 12 (63 quads per 256-column tile; residual
 *             and stores move as 16-byte vectors);
 *   w1_wino4  F(4,3) of the FIRST (dilated) convolution, C = 64 together with w2_wino4 and a dilation whose blocks of 4 d
 *             positions fill a 256-column tile (d <= 32 with >= 48 of 64 quad columns used: the stage's d = 1, 3, 9, 27): BOTH
 *             halves form 6 products per four outputs (resblk4_kernel: the dilated half along the dilated axis, its output
 *             transform into the LDS tile; x needs no guard band there). */
typedef struct {
    const float* w1_direct;  /* required */
    const float* bias1;      /* may be NULL */
    const float* w2_direct;  /* required */
    const float* bias2;      /* may be NULL */
    const float* w2_wino;    /* optional */
    const float* w2_wino4;   /* optional */
    const float* w1_wino4;   /* optional */
} vfx_resblock_w;
int vfx_resblock_f32(const vfx_tensor* x, const vfx_tensor* y, const vfx_resblock_w* w, int B, int C, int L,
                     int dilation, float slope, int post_act, float post_slope, vfx_stream_t stream);

/* ConvTranspose1d(Cin, Cout, kernel 2s, stride s, padding s/2 + s%2, output_padding s%2):
 * Lin -> s*Lin.  Polyphase: s phases x 2 taps.  w_packed = [2s][CinPad][Cout], slab k is
 * kernel tap k.  Replaces voicefixer/vocoder/model/modules.py:449-459,519 (UpsampleNet.layer). */
int vfx_convtr1d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                     const vfx_tensor* y, int B, int Cin, int Cout, int Lin, int stride,
                     const vfx_act* act, vfx_stream_t stream);

/* Conv2d ksize x ksize (ksize 1 or 3, padding ksize/2, stride 1) on pitch-P maps, P = 1<<pitch_log2,
 * W = P-1.  w_packed = [ksize*ksize][CinPad][Cout], slab ky*ksize+kx.  Output pad column is
 * written as zero.  Replaces voicefixer/restorer/modules.py:18-26,33-41,47-53 (ConvBlockRes). */
int vfx_conv2d_f32(const vfx_tensor* x, const float* w_packed, const float* bias,
                   const vfx_tensor* res, const vfx_tensor* y, int B, int Cin, int Cout, int H,
                   int pitch_log2, int ksize, const vfx_act* act, vfx_stream_t stream);

/* ConvTranspose2d 3x3 stride 2 padding 0 followed by the reference's prune of the last
 * output row: (h, w=P-1) -> (2h, 2w+1) on pitch 2P.  w_packed = [9][CinPad][Cout], slab
 * ky*3+kx.  Replaces voicefixer/restorer/modules.py:112-121,141-150 (DecoderBlockRes). */
int vfx_convtr2d_3x3s2_f32(const vfx_tensor* x, const float* w_packed, const vfx_tensor* y,
                           int B, int Cin, int Cout, int h, int in_pitch_log2,
                           const vfx_act* act, vfx_stream_t stream);

/* Cout == 1 convolution (channel reduction, HBM-bound): y[b,0,l] = post(bias + sum w[c][t] * x[b,c,l+t-(k-1)/2]),
 * w = [Cin][k].  Replaces the final ReflectionPad1d(3)+Conv1d(64,1,7)+Tanh
 * (generator.py:95-99) and UNet after_conv2 1x1 (restorer/model_kqq_bn.py:119-126,174). */
int vfx_conv1d_cout1_f32(const vfx_tensor* x, const float* w, const float* bias,
                         const vfx_tensor* y, int B, int Cin, int L, int k, int pad_mode,
                         int post_act, int out_mask_log2, vfx_stream_t stream);

/* avg_pool2d(2,2) (floor) on pitch maps: (H, P) -> (H/2, P/2).
 * Replaces voicefixer/restorer/modules.py:103. */
int vfx_avgpool2x2_f32(const vfx_tensor* x, const vfx_tensor* y, int B, int C, int H,
                       int pitch_log2, vfx_stream_t stream);

/* ---- analysis front-end ---------------------------------------------------------- */

/* One-time upload of the tables used by vfx_stft_mel_f32: periodic hann window [2048],
 * twiddles exp(-2*pi*i*m/2048) m<1024 as interleaved (re,im) [2048], and the banded mel
 * filterbank: lo[128], hi[128] (inclusive bin range), off[128] (start into coef), coef[nnz].
 * All HOST pointers; the library keeps device copies. */
int vfx_frontend_init(const float* window, const float* twiddle, const int32_t* lo,
                      const int32_t* hi, const int32_t* off, const float* coef, int nnz);

/* Test hook: copy the banded filterbank the kernels read back to HOST buffers lo/hi/off[128], coef[coef_capacity]
 * (which = 0: the HTK table of vfx_stft_mel_f32, 1: the slaney table of vfx_stft_mel_oracle_f32); *nnz_out = number
 * of coefficients.  Lets a test assert the mel bin indexing bit-exactly on what the device holds (the reference's
 * matrix is voicefixer/tools/mel_scale.py:147-238).  Tables are kept per device (the caller's current device). */
int vfx_frontend_readback(int which, int32_t* lo, int32_t* hi, int32_t* off, float* coef, int coef_capacity,
                          int* nnz_out);

/* wav [B][N] (row stride wav_stride) -> mel [B][T][128], T = 1 + N/441.  Reflect-padded,
 * centred STFT (n_fft 2048, hop 441), |.| with the 1e-8 power clamp, banded HTK mel.
 * The 1025-bin spectrogram never leaves LDS.  N >= 1025.
 * Replaces voicefixer/base.py:78-85 (_pre): fDomainHelper.py:81-110 + mel_scale.py:63-77. */
int vfx_stft_mel_f32(const float* wav, int64_t wav_stride, int B, int N, float* mel,
                     vfx_stream_t stream);

/* The same for rows of DIFFERENT sample counts n_rows[b] (device int32[B], each >= 1025): row b gets its own
 * 1 + n_rows[b]/441 frames, T is the row pitch of mel (>= every row's frame count).  Pair with vfx_post_rows_f32. */
int vfx_stft_mel_rows_f32(const float* wav, int64_t wav_stride, int B, const int32_t* n_rows, int T, float* mel,
                          vfx_stream_t stream);

/* Vocoder.oracle front-end on the device (voicefixer/vocoder/base.py:61-71): peak[b] = max|wav[b]|
 * (vfx_peak_f32, float bits in a uint32), then |librosa.stft(wav/peak)| (n_fft 2048, hop 441, zero
 * "constant" padding, no clamp) and the slaney-normalised HTK filterbank of librosa.filters.mel,
 * uploaded once in banded form by vfx_frontend_init_oracle -> mel [B][T][128].  Follow with
 * vfx_mel_to_cond_ex_f32(apply_weight = 0). */
int vfx_frontend_init_oracle(const int32_t* lo, const int32_t* hi, const int32_t* off, const float* coef,
                             int nnz);
int vfx_peak_f32(const float* y, int64_t y_bstride, int Ly, int B, uint32_t* peak, vfx_stream_t stream);
int vfx_stft_mel_oracle_f32(const float* wav, int64_t wav_stride, int B, int N, const uint32_t* peak,
                            float* mel, vfx_stream_t stream);

/* mode-1 pre-filter, VoiceFixer.remove_higher_frequency (voicefixer/base.py:87-104): STFT 2048/512
 * (periodic hann, centred, zero padding) -> per-bin clipped log10 energy -> cut-off bin at `ratio`
 * of the cumulative energy -> bins >= cut-off zeroed -> ISTFT (overlap-add, window-sum-square
 * normalisation).  wav [B][N] -> out [B][512*(N/512)].  workspace: vfx_hf_workspace_bytes(B, N)
 * device bytes, 16-byte aligned.  cutoff_out (device, B int32, may be NULL) receives the cut-off
 * bin index of every utterance.  Needs vfx_frontend_init (window, twiddles). */
size_t vfx_hf_workspace_bytes(int B, int N);
int vfx_hf_cut_f32(const float* wav, int64_t wav_stride, int B, int N, float* out, int64_t out_stride,
                   float ratio, void* workspace, size_t workspace_bytes, int32_t* cutoff_out,
                   vfx_stream_t stream);

/* (B,T,128) frame-major <-> channel-major (B,128,ld) transposes used around the denoiser. */
int vfx_tm_to_cm_f32(const float* src, float* dst, int B, int T, int C, int64_t dst_bstride,
                     int64_t dst_cstride, vfx_stream_t stream);

/* clean = mask*mel; x = log10(max(clean,1e-8)); U = [log10(max(mel,1e-8)), x] written as the
 * UNet input (B,nch>=2,Tp*128) pitch map (channels >= 2 are zero filler so the first conv has no
 * channel tail) with bin 127 and rows >= T zero.  mask is channel-major
 * (B,128,*).  Ragged batches: mask->rows[b] = frames of row b (T is then the frame pitch of mel; rows from
 * mask->rows[b] on are zero).  Replaces voicefixer/restorer/model.py:105-108 + model_kqq_bn.py:145-151. */
int vfx_unet_input_f32(const float* mel, const vfx_tensor* mask, const vfx_tensor* unet_in, int nch,
                       int B, int T, int Tp, vfx_stream_t stream);

/* logmel = unet_out + x, x = log10(max(mask*mel,1e-8)); the UNet never sees mel bin 127 and
 * emits 0 there (model_kqq_bn.py:151,177), so logmel[...,127] = x[...,127] is recomputed from
 * mel and mask.  denoised = 10^min(logmel,5).  Both outputs are (B,T,128) frame-major.
 * Replaces restorer/model.py:112 + voicefixer/base.py:125 (from_log). */
int vfx_unet_output_f32(const vfx_tensor* unet_out, const vfx_tensor* unet_in, const float* mel,
                        const vfx_tensor* mask, float* logmel, float* denoised, int B, int T,
                        int Tp, vfx_stream_t stream);

/* One bidirectional GRU layer, hidden 256, PyTorch gate order (r,z,n), h0 = 0: the recurrent
 * part.  gi = x-projections incl. b_ih, frame-major (B,T,1536) = [fwd r,z,n | bwd r,z,n].
 * whh_packed = W_hh of both directions in the layout of voicefixer_amd/packing.py::pack_gru_whh
 * (per direction: register-resident rows, LDS-resident rows, L2-streamed rows; the split is
 * reported by vfx_gru_layout).  bhh = [2][768].  out is channel-major (B,512,*): fwd in channels
 * 0..255, bwd in 256..511.  Ragged batches: out->rows[b] = frames of sequence b (T is then the frame pitch of gi; the
 * reverse direction starts at frame rows[b]-1, frames >= rows[b] of out are not written); same for vfx_gru_bidir2_f32.
 * Replaces the recurrent part of torch.nn.GRU in voicefixer/restorer/model.py:37-44,57-62. */
void vfx_gru_layout(int* kreg, int* klds, int* kstr);
int vfx_gru_bidir_f32(const float* gi, const float* whh_packed, const float* bhh,
                      const vfx_tensor* out, int B, int T, vfx_stream_t stream);

/* Same recurrence with every sequence owned by a PAIR of workgroups (one CU each) that keep W_hh
 * entirely in registers and exchange partial gate sums through tagged 8-byte granules: ~2x faster
 * per step, needs 4*B <= 240 workgroups to be co-resident (split larger batches).  whh_t = plain
 * [2][256][768] (W_hh transposed per direction); mailbox = B*24576 device bytes (zeroed by the call);
 * err_flag = device int32, set to 1 if a partner never answered (bounded spin, never hangs). */
int vfx_gru_bidir2_f32(const float* gi, const float* whh_t, const float* bhh, const vfx_tensor* out,
                       int B, int T, void* mailbox, size_t mailbox_bytes, int32_t* err_flag,
                       vfx_stream_t stream);

/* ---- synthesis front/back -------------------------------------------------------- */

/* mel (B,T,128) linear -> cond (B,128,T') channel-major, T' = T + T%2 + 4, tail = -4.0:
 * m/=18.8927416350036*exp(0.0269863588184314*k); S=20log10(max(1e-5,|m|))-20;
 * c=clip(8(S+115)/115-4,-4,4).  Replaces voicefixer/vocoder/base.py:51-54 +
 * model/util.py:8-36,69-80 + config.py:310-316. */
int vfx_mel_to_cond_f32(const float* mel, const vfx_tensor* cond, int B, int T,
                        vfx_stream_t stream);
/* same with the division by the analytic mel weights optional (0 for the Vocoder.oracle path,
 * whose mel is already slaney-normalised: vocoder/base.py:72 has no weight division) */
int vfx_mel_to_cond_ex_f32(const float* mel, const vfx_tensor* cond, int B, int T, int apply_weight,
                           vfx_stream_t stream);
/* ragged batches: t_rows (device int32[B], may be NULL) = frames of every row, T = row pitch of mel; row b is written
 * up to its own t_rows[b] + t_rows[b]%2 + 4 frames */
int vfx_mel_to_cond_rows_f32(const float* mel, const vfx_tensor* cond, int B, int T, const int32_t* t_rows,
                             int apply_weight, vfx_stream_t stream);

/* Per-utterance peak rule + centre trim (voicefixer/base.py:131-135, _trim_center :63-76):
 * peak[b] = max|y[b,:]|; out[b, 0:N] = y[b, d/2 : d/2+N] * (peak>1 ? 1/peak : 1), d = Ly-N.
 * peak_ws is a caller-provided device buffer of B uint32 words. */
int vfx_post_f32(const float* y, int64_t y_bstride, int Ly, float* out, int64_t out_bstride,
                 int N, int B, uint32_t* peak_ws, vfx_stream_t stream);

/* The same with per-row lengths (ragged batches): row b keeps n_rows[b] samples (device int32[B], every
 * n_rows[b] <= n_max) of its ly_rows[b] <= Ly vocoder samples (ly_rows NULL: every row has Ly), d = ly_rows[b] - n_rows[b];
 * the peak is taken over the row's own samples.  Columns >= n_rows[b] of out are left untouched. */
int vfx_post_rows_f32(const float* y, int64_t y_bstride, int Ly, const int32_t* ly_rows, float* out,
                      int64_t out_bstride, const int32_t* n_rows, int n_max, int B, uint32_t* peak_ws,
                      vfx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VFX_HIP_H */
