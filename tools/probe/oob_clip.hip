// Does a raw-buffer dwordx4 store / load that STRADDLES num_records get clipped per dword on gfx950?   hipcc --offload-arch=gfx950 -O2 oob_clip.hip -o oob_clip && ./oob_clip
// (round 6: convwg4x_kernel's trickle would like to store the quad that straddles a row's end with the row length as num_records.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* buf, float* out, int nrec_bytes) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, (short)0, nrec_bytes, 0x00020000);
    const int lane = threadIdx.x;
    u32x4 v;
    v.x = __float_as_uint(100.f + 4 * lane); v.y = __float_as_uint(101.f + 4 * lane);
    v.z = __float_as_uint(102.f + 4 * lane); v.w = __float_as_uint(103.f + 4 * lane);
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0);
    out[4 * lane + 0] = __uint_as_float(r.x); out[4 * lane + 1] = __uint_as_float(r.y);
    out[4 * lane + 2] = __uint_as_float(r.z); out[4 * lane + 3] = __uint_as_float(r.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 0);
}
int main() {
    float *buf, *out, h[32], o[32];
    hipMalloc(&buf, 32 * 4); hipMalloc(&out, 32 * 4);
    for (int nrec : {40, 44, 36}) {
        for (int i = 0; i < 32; ++i) h[i] = (float)i;
        hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice);
        hipMemset(out, 0, 32 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, buf, out, nrec);
        hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        printf("num_records = %d bytes (%d floats)\n  loaded :", nrec, nrec / 4);
        for (int i = 0; i < 16; ++i) printf(" %g", o[i]);
        printf("\n  memory :");
        for (int i = 0; i < 16; ++i) printf(" %g", h[i]);
        printf("\n");
    }
    return 0;
}
