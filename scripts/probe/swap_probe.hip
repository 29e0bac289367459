// swap_probe.hip -- stand-alone reproducer for the v_permlane32_swap wrong-result bug (profiles/r05_determinism.txt): does the
// instruction, issued behind a v_mfma_f32_32x32x16_bf16 the way the conv epilogues issued it, ever return something else than the
// two-shuffle regroup of the same registers?
//
//   hipcc --offload-arch=gfx950 -O3 -o swap_probe swap_probe.hip
//   ./swap_probe [launches] [iterations per launch]              (three at once: the "shared GPU" condition of round 5)
//
// Per iteration every wave forms one 32 x 32 x 16 product of lane- and iteration-dependent bf16 operands, regroups its accumulator
// with eight v_permlane32_swap (exactly the epilogue's pairs: rows 8v+j with 8v+4+j) and, from the SAME registers, with two
// ds_bpermute exchanges per pair; a lane whose two results differ raises a flag, and the wave counts the 16-lane groups that hold a
// flag (ballot).  Kernel shapes: 1 x 512 threads per CU (160 KB of LDS claimed) and 2 x 256 threads per CU (80 KB each) -- the
// two-workgroups-per-CU shape is the one that failed alone on the GPU.  Modes: 0 = MFMA -> swap back to back; 1 = the epilogue's
// neighbourhood as well (a global load in flight, LDS reads between the swaps, a global store behind them, one barrier per iteration);
// 2 = the EXACT instruction sequence the compiler emitted in conv_small_fwd_kernel's epilogue (four v_mov, four v_permlane32_swap back
// to back, then VALU reads of the swapped registers at once -- no LDS wait between the swap and its first consumer, unlike modes 0/1,
// where the reference shuffles put an s_waitcnt in front of every comparison), pinned with inline assembly; the reference is formed
// BEFORE the sequence from copies of the registers.
// Counts are of WAVE-LEVEL swap instructions.  Exit status 0 whatever is found; the numbers go to stdout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void swap_kernel(unsigned long long* counts, const uint4* in, uint4* out, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_tab = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 1024; i += THREADS) s_tab[i] = (float)(i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const unsigned wid = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    unsigned long long bad_groups = 0, bad_swaps = 0, swaps = 0;
    uint4 pend = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 1) pend = in[(hash32(wid + it) & 0xffff) * 64 + lane];          // a load in flight across the product
        // operands: small integers in bf16 (exact products and sums), different in every lane and iteration
        unsigned r = hash32(seed ^ (wid * 0x9e3779b9u) ^ (unsigned)(it * 64 + lane));
        unsigned au[4], bu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r = hash32(r + e);
            const unsigned lo = 0x3f80u + ((r & 3) << 5), hi = 0x3f80u + (((r >> 2) & 3) << 5);        // 1.0, 1.25, 1.5, 1.75
            au[e] = lo | (hi << 16);
            const unsigned lo2 = 0x3f80u + (((r >> 4) & 3) << 5), hi2 = 0x3f80u + (((r >> 6) & 3) << 5);
            bu[e] = lo2 | (hi2 << 16);
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(au[0], au[1], au[2], au[3])),
                                                      __builtin_bit_cast(bf16x8, make_uint4(bu[0], bu[1], bu[2], bu[3])), acc, 0, 0, 0);
        unsigned flag = 0;
        float sum = 0.f;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                unsigned a[4], b[4], r0[4], r1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = __float_as_uint(acc[8 * v + j]);
                    b[j] = __float_as_uint(acc[8 * v + 4 + j]);
                    const unsigned oa = __shfl_xor(a[j], 32, 64), ob = __shfl_xor(b[j], 32, 64);
                    r0[j] = h == 0 ? a[j] : ob;
                    r1[j] = h == 0 ? oa : b[j];
                }
                unsigned t0, t1, t2, t3, e0, e1, e2, e3;
                float s0, s1;
                asm volatile(
                    "s_waitcnt lgkmcnt(0)\n"
                    "v_mov_b32 %[t0], %[b0]\n"
                    "v_mov_b32 %[t1], %[b1]\n"
                    "v_mov_b32 %[t2], %[b2]\n"
                    "v_mov_b32 %[t3], %[b3]\n"
                    "v_permlane32_swap_b32 %[a0], %[t0]\n"
                    "v_permlane32_swap_b32 %[a1], %[t1]\n"
                    "v_permlane32_swap_b32 %[a2], %[t2]\n"
                    "v_permlane32_swap_b32 %[a3], %[t3]\n"
                    "v_mov_b32 %[e0], %[a1]\n"
                    "v_mov_b32 %[e1], %[a0]\n"
                    "v_mov_b32 %[e2], %[a3]\n"
                    "v_mov_b32 %[e3], %[a2]\n"
                    "v_add_f32 %[s0], %[t0], %[t1]\n"
                    "v_add_f32 %[s1], %[t2], %[t3]\n"
                    : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
                      [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [s0] "=&v"(s0), [s1] "=&v"(s1)
                    : [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]));
                flag |= (a[0] != r0[0] || a[1] != r0[1] || a[2] != r0[2] || a[3] != r0[3]) ? 1u : 0u;        // the registers, read later
                flag |= (t0 != r1[0] || t1 != r1[1] || t2 != r1[2] || t3 != r1[3]) ? 1u : 0u;
                flag |= (e0 != r0[1] || e1 != r0[0] || e2 != r0[3] || e3 != r0[2]) ? 2u : 0u;                // the reads right behind the swaps
                flag |= (s0 != __uint_as_float(r1[0]) + __uint_as_float(r1[1]) || s1 != __uint_as_float(r1[2]) + __uint_as_float(r1[3])) ? 2u : 0u;
            }
        } else
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a = __float_as_uint(acc[8 * v + j]), b = __float_as_uint(acc[8 * v + 4 + j]);
                const auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                const unsigned oa = __shfl_xor(a, 32, 64), ob = __shfl_xor(b, 32, 64);
                const unsigned r0 = h == 0 ? a : ob, r1 = h == 0 ? oa : b;
                flag |= (s[0] != r0 || s[1] != r1) ? 1u : 0u;
                if constexpr (MODE == 1) sum += s_tab[(lane * 4 + j + 8 * v) & 1023] + __uint_as_float(s[0]) + __uint_as_float(s[1]);
            }
        const unsigned long long m = __ballot(flag != 0);
        swaps += 8;
        if (m) {
            bad_swaps += 1;
            for (int g = 0; g < 4; ++g) bad_groups += ((m >> (16 * g)) & 0xffffull) ? 1 : 0;
        }
        if constexpr (MODE == 1) {
            out[(wid & 0xfff) * 64 + lane] = make_uint4(__float_as_uint(sum), pend.x, pend.y ^ pend.z, pend.w);
            __syncthreads();
        }
    }
    if (lane == 0) {
        atomicAdd(&counts[0], swaps);
        atomicAdd(&counts[1], bad_swaps);                    // iterations (of 8 swaps) with at least one wrong lane
        atomicAdd(&counts[2], bad_groups);                   // 16-lane groups holding a wrong lane
    }
}

template <int THREADS, int MODE>
static void run(const char* name, int launches, int iters, int cus, unsigned long long* d_counts, uint4* d_in, uint4* d_out) {
    const int wg_per_cu = 512 / THREADS;
    const size_t lds = (size_t)160 * 1024 / wg_per_cu - 256;        // claims the CU's LDS: exactly wg_per_cu workgroups resident
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&swap_kernel<THREADS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long tot[3] = {0, 0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int l = 0; l < launches; ++l) {
        hipMemsetAsync(d_counts, 0, 3 * sizeof(unsigned long long), 0);
        hipLaunchKernelGGL((swap_kernel<THREADS, MODE>), dim3(cus * wg_per_cu), dim3(THREADS), lds, 0, d_counts, d_in, d_out, iters, 0x1234u + l);
        unsigned long long c[3];
        hipMemcpy(c, d_counts, sizeof(c), hipMemcpyDeviceToHost);
        for (int i = 0; i < 3; ++i) tot[i] += c[i];
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s swaps %14llu  iterations with a wrong lane %10llu  wrong 16-lane groups %10llu  (%d launches, %.0f ms)  err %d\n", name, tot[0],
           tot[1], tot[2], launches, ms, (int)hipGetLastError());
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_counts;
    uint4 *d_in, *d_out;
    hipMalloc(&d_counts, 3 * sizeof(unsigned long long));
    hipMalloc(&d_in, (size_t)65536 * 64 * sizeof(uint4));
    hipMalloc(&d_out, (size_t)4096 * 64 * sizeof(uint4));
    hipMemset(d_in, 1, (size_t)65536 * 64 * sizeof(uint4));
    printf("swap_probe on %s (%d CUs): %d launches x %d iterations per shape\n", prop.name, cus, launches, iters);
    run<512, 0>("1 x 512 threads per CU, MFMA -> swap", launches, iters, cus, d_counts, d_in, d_out);
    run<256, 0>("2 x 256 threads per CU, MFMA -> swap", launches, iters, cus, d_counts, d_in, d_out);
    run<512, 1>("1 x 512 threads per CU, epilogue neighbourhood", launches, iters, cus, d_counts, d_in, d_out);
    run<256, 1>("2 x 256 threads per CU, epilogue neighbourhood", launches, iters, cus, d_counts, d_in, d_out);
    run<512, 2>("1 x 512 threads per CU, the kernel's sequence", launches, iters, cus, d_counts, d_in, d_out);
    run<256, 2>("2 x 256 threads per CU, the kernel's sequence", launches, iters, cus, d_counts, d_in, d_out);
    return 0;
}
