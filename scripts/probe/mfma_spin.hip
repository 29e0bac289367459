// mfma_spin.hip -- load generators for the packed-fp32 / op_sel hunt: kernels that leave room on a CU (256 threads, no LDS, few registers)
// and do ONE thing in a loop.   ./mfma_spin.bin <mfma|valu|pkswap|dpp|lds|trans> [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f2;
template <int KIND>
__global__ __launch_bounds__(256) void spin(float* sink, int iters) {
    __shared__ float sh[256];
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float v = threadIdx.x * 1e-3f + 1.f, w = 0.5f;
    f2 p = {v, w}, q = {w, v};
    const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        if constexpr (KIND == 6) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc4, 0, 0, 0);
        if constexpr (KIND == 1) { v = __builtin_fmaf(v, 1.0001f, w); w = __builtin_fmaf(w, 0.9999f, v); }
        if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(q));
        if constexpr (KIND == 3) { v += __shfl_xor(v, 1, 64); }
        if constexpr (KIND == 4) { sh[threadIdx.x] = v; __syncthreads(); v += sh[(threadIdx.x + 65) & 255]; __syncthreads(); }
        if constexpr (KIND == 5) { v = sqrtf(v) + 1.0f / (w + 2.f); }
    }
    float s = v + w + p.x + p.y + acc4[0];
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) sink[0] = s;
}
int main(int argc, char** argv) {
    const char* kind = argc > 1 ? argv[1] : "mfma";
    const double secs = argc > 2 ? atof(argv[2]) : 20.0;
    float* d; hipMalloc(&d, 16);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 10; ++k) {
            if (!strcmp(kind, "mfma")) hipLaunchKernelGGL(spin<0>, dim3(1024), dim3(256), 0, 0, d, 20000);
            else if (!strcmp(kind, "mfma16")) hipLaunchKernelGGL(spin<6>, dim3(1024), dim3(256), 0, 0, d, 20000);
            else if (!strcmp(kind, "valu")) hipLaunchKernelGGL(spin<1>, dim3(1024), dim3(256), 0, 0, d, 200000);
            else if (!strcmp(kind, "pkswap")) hipLaunchKernelGGL(spin<2>, dim3(1024), dim3(256), 0, 0, d, 200000);
            else if (!strcmp(kind, "dpp")) hipLaunchKernelGGL(spin<3>, dim3(1024), dim3(256), 0, 0, d, 50000);
            else if (!strcmp(kind, "lds")) hipLaunchKernelGGL(spin<4>, dim3(1024), dim3(256), 0, 0, d, 20000);
            else hipLaunchKernelGGL(spin<5>, dim3(1024), dim3(256), 0, 0, d, 50000);
        }
        hipDeviceSynchronize();
        n += 10;
    }
    printf("mfma_spin %s: %ld launches\n", kind, n);
    return 0;
}
