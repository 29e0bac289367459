// box.hip -- what THIS box's GPU sustains on two fixed micro-kernels: a streaming copy (16 B per lane, read + write) and a dependent-free
// loop of v_mfma_f32_32x32x16_bf16.  bench.py prints both beside the step rate (`box`, `value_normalised`): boxes of the pool differ by
// +-3 % in clocks / HBM (round 5: more than a round's gain), and a line that carries its own normaliser can be compared across boxes.
// Measurement infrastructure only: nothing on the training path calls it.
#include "common.h"
#include "../../include/ramdsir.h"

namespace {

__global__ __launch_bounds__(256) void box_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// 8 waves per CU (two per SIMD), four independent accumulators per wave: the matrix pipe never waits for a result
__global__ __launch_bounds__(512, 1) void box_mfma_kernel(float* sink, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const unsigned seed = 0x3f803f80u + (threadIdx.x & 3) * 0x00200020u;          // bf16 pairs of 1.0 .. 1.75
    const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(seed, seed, seed, seed));
    const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u));   // 2^-7: sums stay finite
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.678f) sink[0] = s;                      // keeps the loop alive; never true
}

// stamps of the shader-clock counter (s_memtime) and of the constant 100 MHz counter, ONE PAIR PER COMPUTE UNIT (the shader-clock
// counters of different XCDs are not aligned: a difference is only meaningful between two stamps taken on the same unit).  4096 small
// workgroups reach every unit; each writes the slot of the unit it runs on: out[unit][0..1], unit = XCC_ID << 8 | SE_ID << 5 | SH_ID << 4
// | CU_ID (2048 slots).  Two launches around a stretch of work on one stream give its AVERAGE SHADER CLOCK per unit (bench.py
// `box.shader_ghz`: is the step clock-throttled?)
__global__ __launch_bounds__(64) void box_clock_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned unit = ((xcc & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        out[2 * unit + 0] = __builtin_readcyclecounter();
        out[2 * unit + 1] = wall_clock64();
    }
}

}  // namespace

extern "C" int rd_box_probe(int which, void* a, void* b, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (which == 0) {                                      // copy n bytes from a to b (n % 16 == 0)
        if (!a || !b || n < 16 || n % 16) return -1;
        rd_launch(box_copy_kernel, dim3(rd_num_cus() * 8), dim3(256), 0, st, (const uint4*)a, (uint4*)b, (size_t)(n / 16));
    } else if (which == 1) {                               // n iterations of 4 MFMAs per wave, 8 waves per CU; a: 4 bytes of scratch
        if (!a || n < 1 || n > (1ll << 30)) return -1;
        rd_launch(box_mfma_kernel, dim3(rd_num_cus()), dim3(512), 0, st, (float*)a, (int)n);
    } else if (which == 2) {                               // a: 2048 x two 8-byte counters (shader clock, 100 MHz clock), one pair per unit
        if (!a) return -1;
        rd_launch(box_clock_kernel, dim3(4096), dim3(64), 0, st, (unsigned long long*)a);
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}
