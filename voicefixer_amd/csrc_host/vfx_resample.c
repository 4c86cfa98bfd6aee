/* vfx_resample.c -- polyphase rate conversion for the folder driver's decode workers (host side, plain C).
 *
 * The reference resamples whatever it loads to 44.1 kHz inside librosa.load (voicefixer/base.py:47-49: soxr's "HQ"
 * converter, native code).  audio_io.resample_hq designs the filter (Kaiser-windowed sinc to the same recipe) and, until
 * this file existed, applied it with scipy's upfirdn: 168x real time per thread for 48 kHz input and barely more on eight
 * threads -- an eighth of what one MI355X restores.  Here the same sum
 *
 *     y[m] = sum_k g[c + m*down - k*up] * x[k],   g = up * h (L taps, odd), c = (L - 1) / 2,   m < ny
 *
 * is evaluated per output sample as ONE dot product of the input with the taps of its phase (phase = (c + m*down) mod up),
 * the taps of every phase stored contiguously and reversed so that both operands are read ascending; sixteen partial
 * sums let the compiler vectorise without reassociating, the AVX2+FMA clone is picked at load time where the CPU has it.
 * Called through ctypes without the interpreter lock.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vfx_audio.h"

#define NACC 16

__attribute__((target_clones("avx2,fma", "default")))
static float dot_f32(const float* a, const float* b, long n) {
    float acc[NACC];
    for (int l = 0; l < NACC; ++l) acc[l] = 0.f;
    long j = 0;
    for (; j + NACC <= n; j += NACC)
        for (int l = 0; l < NACC; ++l) acc[l] += a[j + l] * b[j + l];
    float s = 0.f;
    for (; j < n; ++j) s += a[j] * b[j];
    for (int l = 0; l < NACC; ++l) s += acc[l];
    return s;
}

int vfx_resample_poly_f32(const float* x, unsigned long long n, const float* g, int L, int up, int down,
                          float* y, unsigned long long ny) {
    if (!x || !g || !y || L < 1 || !(L & 1) || up < 1 || down < 1) return VFX_FLAC_EINVAL;
    const long c = (L - 1) / 2;
    const long J = (L + up - 1) / up;                 /* taps per phase (the last ones of some phases are zero fill) */
    /* bank[p][i] = g[p + (J - 1 - i) * up]: the taps of phase p, reversed */
    float* bank = (float*)calloc((size_t)up * (size_t)J, sizeof(float));
    if (!bank) return VFX_FLAC_ENOMEM;
    for (long p = 0; p < up; ++p)
        for (long j = 0; j < J; ++j) {
            const long t = p + j * up;
            bank[p * J + (J - 1 - j)] = t < L ? g[t] : 0.f;
        }
    for (unsigned long long m = 0; m < ny; ++m) {
        const long long pos = (long long)c + (long long)m * down;   /* tap index that meets x[0] */
        const long long kmax = pos / up;                            /* newest input sample under the filter */
        const long p = (long)(pos - kmax * up);
        /* taps j = 0 .. J-1 meet x[kmax - j]; reversed: bank[p][i] meets x[kmax - J + 1 + i] */
        long long lo = kmax - J + 1;                                /* input index under bank[p][0] */
        long i0 = 0, i1 = J;
        if (lo < 0) { i0 = (long)(-lo); }
        if (lo + J > (long long)n) { i1 = (long)((long long)n - lo); }
        y[m] = i1 > i0 ? dot_f32(bank + p * J + i0, x + (lo + i0), i1 - i0) : 0.f;
    }
    free(bank);
    return 0;
}
