// lds_canary.hip -- does a kernel of ANOTHER process write into this process's LDS?  Workgroups of 256 threads fill their LDS allocation
// (28.8 KB by default: what the RAM column / row-inverse kernels use, five workgroups per CU) with an address pattern, hold it for ~0.3 ms
// while re-reading it, and report every word that changed: offset, value found, how many workgroups saw a change.
//   ./lds_canary.bin [seconds] [lds_bytes]        (run beside scripts/r6/aggressor.py processes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
struct Rec { unsigned off, val, block, iter; };
__global__ __launch_bounds__(256) void canary(Rec* recs, unsigned* nrec, unsigned* nbad_wg, int words, int iters) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0xC0DE0000u ^ (unsigned)i * 2654435761u;
    __syncthreads();
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = lds[i], want = 0xC0DE0000u ^ (unsigned)i * 2654435761u;
            if (v != want) {
                const unsigned k = atomicAdd(nrec, 1u);
                if (k < 4096) recs[k] = Rec{(unsigned)i * 4, v, blockIdx.x, (unsigned)it};
                lds[i] = want;
                bad = 1;
            }
        }
        __builtin_amdgcn_s_sleep(32);
        __syncthreads();
    }
    if (threadIdx.x == 0 && bad) atomicAdd(nbad_wg, 1u);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 20.0;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 28800;
    Rec* d_recs; unsigned *d_n, *d_bw;
    hipMalloc(&d_recs, 4096 * sizeof(Rec)); hipMalloc(&d_n, 4); hipMalloc(&d_bw, 4);
    hipMemset(d_n, 0, 4); hipMemset(d_bw, 0, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&canary), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(canary, dim3(256 * 5), dim3(256), lds_bytes, 0, d_recs, d_n, d_bw, lds_bytes / 4, 40);
        hipDeviceSynchronize();
        launches += 20;
    }
    unsigned n, bw;
    hipMemcpy(&n, d_n, 4, hipMemcpyDeviceToHost); hipMemcpy(&bw, d_bw, 4, hipMemcpyDeviceToHost);
    static Rec h[4096];
    hipMemcpy(h, d_recs, sizeof(h), hipMemcpyDeviceToHost);
    printf("lds_canary: %ld launches x 1280 workgroups of %d bytes: %u changed words in %u workgroups\n", launches, lds_bytes, n, bw);
    for (unsigned i = 0; i < n && i < 48; ++i) printf("  block %5u iter %2u offset %6u (0x%05x): found 0x%08x (as float %g)\n", h[i].block, h[i].iter, h[i].off, h[i].off, h[i].val, *(float*)&h[i].val);
    return 0;
}
