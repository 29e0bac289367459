// Probe of ds_read_b64_tr_b16 semantics on gfx950: each lane supplies an 8-byte-aligned LDS address; print what it gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr;   // in elements
    if (mode == 0) addr = l * 4;                                   // contiguous: lane l reads elements 4l..4l+3
    else if (mode == 1) addr = (l & 15) / 4 * 64 + (l & 3) * 4 + (l >> 4) * 256;   // rows of 64 elements (128 B)
    else addr = ((l & 15) / 4) * 100 + (l & 3) * 4 + (l >> 4) * 1000;              // arbitrary row stride 100 elems
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + addr));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
