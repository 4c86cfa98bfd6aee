/* vfx_audio.h -- C ABI of libvfx_audio.so: the native FLAC frame codec behind voicefixer_amd/flac.py and the polyphase
 * resampler behind voicefixer_amd/audio_io.py (host side).
 *
 * Replaces, for the folder driver's decode / encode workers, what the reference reaches through
 * librosa.load (voicefixer/base.py:47-49) and soundfile.write (voicefixer/tools/wav.py:36-37): libsndfile's FLAC
 * codec.  Plain pointers and sizes, no Python or torch types; every call is re-entrant and touches only its arguments
 * (ctypes releases the interpreter lock around it, so a thread pool decodes / encodes files in parallel).
 * The container level (metadata blocks, STREAMINFO, the MD5 of the decoded audio) stays in flac.py.
 */
#ifndef VFX_AUDIO_H
#define VFX_AUDIO_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
    VFX_FLAC_OK = 0,
    VFX_FLAC_EINVAL = 1,     /* bad argument */
    VFX_FLAC_ESYNC = 2,      /* lost frame sync */
    VFX_FLAC_ECRC8 = 3,      /* frame header CRC-8 mismatch */
    VFX_FLAC_ECRC16 = 4,     /* frame CRC-16 mismatch */
    VFX_FLAC_ERESERVED = 5,  /* reserved / invalid field in the stream */
    VFX_FLAC_EOVERRUN = 6,   /* a code runs past the end of the data */
    VFX_FLAC_ECHANNELS = 7,  /* channel count changes mid-stream */
    VFX_FLAC_ECAPACITY = 8,  /* output buffer too small */
    VFX_FLAC_ENOMEM = 9
};

int vfx_audio_version(void);

/* Decode the audio frames of a FLAC stream.  data / len: the whole file; first_frame: byte offset of the first frame
 * (after the metadata blocks); nch / bps0: channels and bits per sample from STREAMINFO; out: interleaved samples,
 * room for cap_samples per channel.  *decoded = samples per channel written; on an error *err_byte = offset of the
 * frame it occurred in.  verify != 0 checks the CRC-8 of every frame header and the CRC-16 of every frame. */
int vfx_flac_decode_frames(const unsigned char* data, unsigned long long len, unsigned long long first_frame,
                           int nch, int bps0, int* out, unsigned long long cap_samples,
                           unsigned long long* decoded, int verify, unsigned long long* err_byte);

/* Encode n interleaved samples per channel as FLAC frames (FIXED order-2 prediction + one Rice partition, VERBATIM
 * where that would be longer; independent channels; frames of `blocksize` samples numbered from 0): byte for byte the
 * frames flac.py's encoder writes.  Returns the number of bytes written to out (capacity cap), or -(error code);
 * *min_frame / *max_frame = the smallest / largest frame in bytes (STREAMINFO fields). */
long long vfx_flac_encode_frames(const int* pcm, unsigned long long n, int nch, int bps, int blocksize,
                                 unsigned char* out, unsigned long long cap, unsigned* min_frame, unsigned* max_frame);

/* Polyphase rate conversion (what librosa.load(sr=44100) does inside the reference, voicefixer/base.py:47-49):
 *   y[m] = sum_k g[c + m*down - k*up] * x[k],  c = (L - 1) / 2,  m < ny
 * g = the L-tap (L odd) zero-phase low-pass at the common rate up * fs_in, already scaled by up; x has n samples (zero
 * outside).  This is scipy.signal.resample_poly(x, up, down, window=g / up) in float32 with one dot product per output
 * sample.  Returns 0 or VFX_FLAC_EINVAL / VFX_FLAC_ENOMEM. */
int vfx_resample_poly_f32(const float* x, unsigned long long n, const float* g, int L, int up, int down,
                          float* y, unsigned long long ny);

#ifdef __cplusplus
}
#endif
#endif
