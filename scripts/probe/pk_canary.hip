// pk_canary.hip -- do the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) of a wave still return the right bits while
// workgroups of another kernel share the CU?  The Stockham RAM kernels failed only when built WITH these instructions (profiles/
// r06_ram_coresidency.txt).  Every thread evaluates each packed operation and the same two scalar operations (v_fma_f32 / v_mul_f32 /
// v_add_f32: the same IEEE results) on operands that change every round, directly and through an LDS round trip as a butterfly would.
//   ./pk_canary.bin [seconds] [lds_bytes]        (run beside scripts/r6/aggressor.py processes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(2))) float f2;
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the modifier forms complex butterflies compile to: negated halves, swapped halves
__device__ __forceinline__ f2 pk_add_neg(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // a - b
__device__ __forceinline__ f2 pk_add_neghi(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }          // (a.x + b.x, a.y - b.y)
__device__ __forceinline__ f2 pk_mul_swap(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (a.x * b.y, a.y * b.x)
__device__ __forceinline__ f2 pk_fma_swapneg(f2 a, f2 b, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }   // (a.x * b.y - c.x, a.y * b.x + c.y)
__device__ __forceinline__ f2 pk_mul_bcast(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }     // (a.x * b.x, a.y * b.x)
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bool ne(f2 p, float x, float y) { return __float_as_uint(p.x) != __float_as_uint(x) || __float_as_uint(p.y) != __float_as_uint(y); }
struct Sample { float ax, ay, bx, by, rx, ry; };
__device__ Sample g_samples[64];
__device__ unsigned g_nsamples;
__global__ __launch_bounds__(256) void canary(unsigned* stats, int rounds, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f2* lds = reinterpret_cast<f2*>(smem);
    const int tid = threadIdx.x;
    unsigned bad[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned s = seed ^ (blockIdx.x * 256 + tid) * 2654435761u;
    for (int r = 0; r < rounds; ++r) {
        s = s * 1664525u + 1013904223u;
        const f2 a = {1.0f + (float)(s >> 8) * 5.9604645e-8f, -2.0f + (float)(s & 0xffffu) * 1.52e-5f};
        s = s * 1664525u + 1013904223u;
        const f2 b = {0.5f + (float)(s >> 9) * 1.19e-7f, 1.5f - (float)(s & 0xfffu) * 2.4e-4f};
        s = s * 1664525u + 1013904223u;
        const f2 c = {(float)(s >> 10) * 2.3e-7f - 0.3f, 0.25f + (float)(s & 0xffu) * 3.9e-3f};
        const f2 p1 = pk_fma(a, b, c), p2 = pk_mul(a, b), p3 = pk_add(a, c);
        bad[0] += ne(p1, s_fma(a.x, b.x, c.x), s_fma(a.y, b.y, c.y));
        bad[1] += ne(p2, s_mul(a.x, b.x), s_mul(a.y, b.y));
        bad[2] += ne(p3, s_add(a.x, c.x), s_add(a.y, c.y));
        bad[4] += ne(pk_add_neg(a, c), s_add(a.x, -c.x), s_add(a.y, -c.y));
        bad[5] += ne(pk_add_neghi(a, c), s_add(a.x, c.x), s_add(a.y, -c.y));
        {
            const f2 sw = pk_mul_swap(a, b);
            if (ne(sw, s_mul(a.x, b.y), s_mul(a.y, b.x))) {
                ++bad[6];
                const unsigned k = atomicAdd(&g_nsamples, 1u);
                if (k < 64) g_samples[k] = Sample{a.x, a.y, b.x, b.y, sw.x, sw.y};
            }
        }
        bad[7] += ne(pk_fma_swapneg(a, b, c), s_fma(a.x, b.y, -c.x), s_fma(a.y, b.x, c.y));
        bad[8] += ne(pk_mul_bcast(a, b), s_mul(a.x, b.x), s_mul(a.y, b.x));
        // a butterfly-like round trip: packed results through LDS to another thread and back into packed operations
        lds[tid] = p1;
        lds[256 + tid] = p2;
        __syncthreads();
        const int o = (tid * 5 + 64) & 255;
        const f2 u = lds[o], v = lds[256 + o];
        const f2 w = pk_fma(u, v, p3);
        bad[3] += ne(w, s_fma(u.x, v.x, p3.x), s_fma(u.y, v.y, p3.y));
        __syncthreads();
    }
    for (int k = 0; k < 9; ++k) if (bad[k]) atomicAdd(&stats[k], bad[k]);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 12.0;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 28800;
    unsigned* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&canary), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(canary, dim3(256 * 5), dim3(256), lds_bytes, 0, d, 100, (unsigned)launches + k);
        hipDeviceSynchronize();
        launches += 20;
    }
    unsigned h[9];
    hipMemcpy(h, d, 36, hipMemcpyDeviceToHost);
    printf("pk_canary: %ld launches x 1280 workgroups x 256 threads x 100 rounds: packed != scalar   pk_fma %u  pk_mul %u  pk_add %u  pk_fma behind LDS %u"
           "  add neg %u  add neg_hi %u  mul swapped %u  fma swapped+neg %u  mul broadcast %u\n", launches, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    static Sample hs[64];
    unsigned ns = 0;
    hipMemcpyFromSymbol(&ns, HIP_SYMBOL(g_nsamples), 4);
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_samples), sizeof(hs));
    for (unsigned i = 0; i < ns && i < 12; ++i) {
        const Sample& q = hs[i];
        printf("  mul swapped: a = (%.9g, %.9g) b = (%.9g, %.9g): got (%.9g, %.9g), want (%.9g, %.9g); unswapped product (%.9g, %.9g); (ax*bx, ay*bx)=(%.9g, %.9g); (ax*by, ay*by)=(%.9g,%.9g)\n", q.ax, q.ay, q.bx, q.by, q.rx, q.ry,
               q.ax * q.by, q.ay * q.bx, q.ax * q.bx, q.ay * q.by, q.ax * q.bx, q.ay * q.bx, q.ax * q.by, q.ay * q.by);
    }
    return 0;
}
