// Probe: does __builtin_amdgcn_global_load_lds with 16 bytes per lane work on this toolchain / gfx950, and where do the bytes land?
// Each lane passes its own global address; the LDS destination is the (wave-uniform) base + lane * 16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out, const int* perm) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned* g = src + perm[lane] * 4;                 // lane reads 16 B from a permuted global slot
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + 64 * 4), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);                             // vmcnt(0): the DMA counts on vmcnt
    __syncthreads();
    for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(64 * 4);
    for (int i = 0; i < 64 * 4; ++i) h[i] = 1000 * (i / 4) + (i % 4);
    std::vector<int> perm(64);
    for (int i = 0; i < 64; ++i) perm[i] = (i * 7 + 3) % 64;
    unsigned *d, *o; int* p;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 4 * 2 * 4); hipMalloc(&p, 64 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(p, perm.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, p);
    std::vector<unsigned> r(64 * 4 * 2);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, untouched = 0;
    for (int i = 0; i < 64 * 4; ++i) if (r[i] != 0xdeadbeefu) ++untouched;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j)
            if (r[64 * 4 + lane * 4 + j] != (unsigned)(1000 * perm[lane] + j)) ++bad;
    printf("first half modified: %d words; second half mismatches vs lane-ordered expectation: %d of 256\n", untouched, bad);
    printf("lane 0: %u %u %u %u (expect %d..)  lane 1: %u ..\n", r[256], r[257], r[258], r[259], 1000 * perm[0], r[260]);
    return 0;
}
