// barrier_canary.hip -- does a workgroup barrier still hold when workgroups of ANOTHER kernel share the CU?  Every round each thread writes
// the round number into its own LDS word, passes a barrier, reads the word of a thread of ANOTHER wave and of its own wave, passes a second
// barrier.  A read that returns an older round = the barrier (or the LDS write it orders) did not hold.  Three access widths (4 / 8 / 16
// bytes: ds_write_b32 / _b64 / _b128) because the failing RAM kernels use 8-byte accesses.
//   ./barrier_canary.bin [seconds] [lds_bytes]      (run beside scripts/r6/aggressor.py processes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
template <typename V>
__global__ __launch_bounds__(256) void canary(unsigned* stats, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    V* lds = reinterpret_cast<V*>(smem);
    const int tid = threadIdx.x;
    unsigned stale = 0, wrong = 0;
    for (int r = 1; r <= rounds; ++r) {
        V w;
        unsigned* wp = reinterpret_cast<unsigned*>(&w);
        for (unsigned i = 0; i < sizeof(V) / 4; ++i) wp[i] = (unsigned)r * 1024u + tid;
        // a Stockham-like pattern: 7 words per thread, strided
        for (int k = 0; k < 7; ++k) lds[(tid + 256 * k)] = w;
        __syncthreads();
        for (int k = 0; k < 7; ++k) {
            const int src = (tid * 37 + 64 + 11 * k) & 255;          // another thread, mostly another wave
            const V got = lds[src + 256 * ((k + 3) % 7)];
            const unsigned* gp = reinterpret_cast<const unsigned*>(&got);
            for (unsigned i = 0; i < sizeof(V) / 4; ++i) {
                const unsigned want = (unsigned)r * 1024u + src;
                if (gp[i] != want) { if (gp[i] == want - 1024u) ++stale; else ++wrong; }
            }
        }
        __syncthreads();
    }
    if (stale) atomicAdd(&stats[0], stale);
    if (wrong) atomicAdd(&stats[1], wrong);
    if (stale | wrong) atomicAdd(&stats[2], 1u);
}
template <typename V>
static void run(const char* name, double secs, int lds_bytes, unsigned* d) {
    hipMemset(d, 0, 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&canary<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(canary<V>, dim3(256 * 5), dim3(256), lds_bytes, 0, d, 60);
        hipDeviceSynchronize();
        launches += 20;
    }
    unsigned h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("barrier_canary %-6s %ld launches x 1280 workgroups x 60 rounds: %u stale reads (previous round), %u other wrong reads, %u threads affected\n", name, launches, h[0], h[1], h[2]);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 12.0;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 28800;
    unsigned* d; hipMalloc(&d, 16);
    run<unsigned>("b32", secs / 3, lds_bytes, d);
    run<uint2>("b64", secs / 3, lds_bytes, d);
    run<uint4>("b128", secs / 3, lds_bytes, d);
    return 0;
}
