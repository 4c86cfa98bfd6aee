// Second probe (round 6): a dwordx4 buffer load whose voffset is -4 (0xFFFFFFFC) with the row selected through SOFFSET: which dwords come back?
//   hipcc --offload-arch=gfx950 -O2 oob_clip2.hip -o oob_clip2 && ./oob_clip2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* buf, float* out, int nrec_bytes, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, (short)0, nrec_bytes, 0x00020000);
    const int lane = threadIdx.x;
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (3 * lane - 1) * 4, soff, 0);
    out[4 * lane + 0] = __uint_as_float(r.x); out[4 * lane + 1] = __uint_as_float(r.y);
    out[4 * lane + 2] = __uint_as_float(r.z); out[4 * lane + 3] = __uint_as_float(r.w);
}
int main() {
    float *buf, *out, h[64], o[32];
    hipMalloc(&buf, 64 * 4); hipMalloc(&out, 32 * 4);
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice);
    for (int soff : {0, 32 * 4}) {
        hipMemset(out, 0, 32 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, buf, out, 10 * 4, soff);     // a "row" of 10 floats at buf + soff
        hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        printf("row at float %d, 10 floats long; lane L loads floats 3L-1 .. 3L+2 of it:\n", soff / 4);
        for (int l = 0; l < 4; ++l) printf("  lane %d: %g %g %g %g\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    }
    return 0;
}
