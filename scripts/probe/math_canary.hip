// math_canary.hip -- do the transcendental / division instructions of a wave still return the right bits while kernels of another
// process share the SIMD?  Every thread evaluates sqrtf, IEEE division, 1/sqrt, fp64 division TWICE on the same operand (the second
// evaluation behind an opaque barrier for the optimiser) and compares the bits; fma chains as the control group.
//   ./math_canary.bin [seconds]        (run beside scripts/r6/aggressor.py processes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
__device__ __forceinline__ float launder(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double launderd(double v) { asm volatile("" : "+v"(v)); return v; }
__global__ __launch_bounds__(256) void canary(unsigned* stats, int rounds, unsigned seed) {
    unsigned bad[5] = {0, 0, 0, 0, 0};
    unsigned s = seed ^ (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        s = s * 1664525u + 1013904223u;
        const float x = 1.0f + (float)(s >> 8) * (1.0f / 16777216.0f) * 1000.0f, a = 3.0f + (float)(s & 255u);
        const float x2 = launder(x), a2 = launder(a);
        const float q1 = sqrtf(x), q2 = sqrtf(x2);
        const float d1 = __fdiv_rn(a, x), d2 = __fdiv_rn(a2, x2);
        const float i1 = 1.0f / sqrtf(x + 1e-5f), i2 = 1.0f / sqrtf(x2 + 1e-5f);
        const double e1 = (double)a / (double)x, e2 = launderd((double)a2) / launderd((double)x2);
        const float f1 = __builtin_fmaf(a, x, q1), f2 = __builtin_fmaf(a2, x2, q2);
        bad[0] += __float_as_uint(q1) != __float_as_uint(q2);
        bad[1] += __float_as_uint(d1) != __float_as_uint(d2);
        bad[2] += __float_as_uint(i1) != __float_as_uint(i2);
        bad[3] += __double_as_longlong(e1) != __double_as_longlong(e2);
        bad[4] += __float_as_uint(f1) != __float_as_uint(f2);
    }
    for (int k = 0; k < 5; ++k) if (bad[k]) atomicAdd(&stats[k], bad[k]);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 12.0;
    unsigned* d; hipMalloc(&d, 32); hipMemset(d, 0, 32);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(canary, dim3(256 * 8), dim3(256), 0, 0, d, 200, (unsigned)launches + k);
        hipDeviceSynchronize();
        launches += 20;
    }
    unsigned h[5];
    hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("math_canary: %ld launches x 2048 workgroups x 256 threads x 200 rounds: mismatching pairs  sqrtf %u  fdiv %u  rsqrt %u  f64 div %u  fma (control) %u\n",
           launches, h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
