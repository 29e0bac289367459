// simd_probe.hip -- which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits [5:4]; the dispatcher's choice.)
// wgrad_sym_kernel pairs the two waves of a SIMD in opposite phase order and needs to know.   ./simd_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ char smem[];
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hwid;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 8 * 4);
    static unsigned h[4096 * 8];
    for (int lds : {0, 150 * 1024}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, dim3(512), dim3(512), lds, 0, d);
        hipMemcpy(h, d, 512 * 8 * 4, hipMemcpyDeviceToHost);
        int hist[16] = {0};
        printf("dynamic LDS %d KB: SIMD of waves 0..7 in the first workgroups:\n", lds / 1024);
        for (int b = 0; b < 512; ++b) {
            int cnt[4] = {0, 0, 0, 0};
            bool mod4 = true;
            for (int w = 0; w < 8; ++w) { const int s = (h[b * 8 + w] >> 4) & 3; cnt[s]++; if (s != ((h[b * 8] >> 4) + w) % 4 && w < 4) mod4 = false; }
            bool same = true;
            for (int w = 0; w < 4; ++w) if (((h[b * 8 + w] >> 4) & 3) != ((h[b * 8 + w + 4] >> 4) & 3)) same = false;
            hist[(cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2 ? 1 : 0) + (same ? 2 : 0)]++;
            if (b < 6) { for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
        }
        printf("  of 512 workgroups: two waves per SIMD in %d; waves w and w+4 on the same SIMD in %d\n", hist[1] + hist[3], hist[2] + hist[3]);
    }
    return 0;
}
