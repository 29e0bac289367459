// Micro-benchmark (not part of the library): how should 256 workgroups combine a [9][64][64] fp32 block each?
//   0: plain stores into a private partial per workgroup (what rd_wgrad does) -- 37.7 MB written
//   1: agent-scope float atomics into ONE buffer
//   2: agent-scope float atomics into 8 buffers, one per XCD (HW_REG_XCC_ID)
//   3: workgroup-scope float atomics into the XCD's buffer
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_partials atomic_partials.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
constexpr int N = 9 * 64 * 64;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int reps) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    float* dst = out + (size_t)(MODE == 0 ? blockIdx.x : (MODE == 1 ? 0 : xcc)) * N;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < N; i += 256) {
            const float v = 1.0f + (float)(i & 7);
            if (MODE == 0) dst[i] = v;
            else if (MODE == 3) __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}
int main() {
    float* d;
    hipMalloc(&d, (size_t)256 * N * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int it = 0; it < 3; ++it) {
            hipMemset(d, 0, (size_t)256 * N * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, 1);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, 1);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, 1);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, 1);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h((size_t)8 * N);
            hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
            double tot = 0; for (int s = 0; s < (mode == 1 ? 1 : 8); ++s) for (int i = 0; i < 64; ++i) tot += h[(size_t)s * N + i];
            // expected over the first 64 elements: mode 1: 256 * sum(1..8)*8 = 256*288; modes 2,3: same total over the 8 slots
            printf("mode %d: %.1f us   check(sum of first 64 elements over slots) = %.0f (expect %d for modes 1-3)\n", mode, ms * 1e3, tot, 256 * 288);
        }
    }
    return 0;
}
