// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per ACCESS WIDTH
// (MI355X_MICROARCH.md: "FETCH_SIZE reports half of 16-B/lane streaming reads; other widths and WRITE_SIZE are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Every kernel touches each byte of a 1 GiB buffer exactly once (4x the 256 MB
// Infinity Cache: nothing is served from a cache), so true bytes = 2^30 per kernel (copy: 2^30 read + 2^30 written).
//   hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib
//   rocprofv3 -i ../pmc_hbm.txt --kernel-trace --output-format csv -d out -o p -- ./pmc_calib       (scripts/pmc_calib/run.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr size_t BYTES = 1ull << 30;

template <typename V> __global__ void calib_read(const V* __restrict__ src, unsigned* out, size_t n) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = src[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// 12 bytes per lane as the RAM row pass reads its uint8 pixels: 4 pixels x 3 channels = three dword loads at stride 12
__global__ void calib_read12(const unsigned* __restrict__ src, unsigned* out, size_t n12) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n12; i += (size_t)gridDim.x * blockDim.x)
        acc ^= src[3 * i] ^ src[3 * i + 1] ^ src[3 * i + 2];
    if (acc == 0x12345678u) out[0] = acc;
}
// 8-byte fp64 pairs at a 16-byte pitch's first half: what bn_finalize / the folded prologue read from the statistic slots -- a strided
// read of 16 of every 64 bytes
__global__ void calib_read16_of_64(const uint4* __restrict__ src, unsigned* out, size_t n64) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[4 * i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename V> __global__ void calib_write(V* dst, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v;
        unsigned* w = reinterpret_cast<unsigned*>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) w[k] = seed + (unsigned)i;
        dst[i] = v;
    }
}
__global__ void calib_copy16(const uint4* __restrict__ src, uint4* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// bf16 NHWC halo-tile pattern of the conv loaders: 16 B per lane, rows of 34 pixels x 64 B with a 2-pixel overlap between neighbouring
// tiles (each byte of the tensor is requested 1.06x: the counter should see the true bytes once if the overlap hits in L2)
__global__ void calib_read_tiles(const uint4* __restrict__ src, unsigned* out, int H, int W, int N) {
    // tensor [N][H][W][32 bf16 = 4 x 16 B]; tile 8 x 32 outputs -> halo 10 x 34
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    unsigned acc = 0;
    for (int it = threadIdx.x; it < 10 * 34 * 4; it += blockDim.x) {
        const int s = it & 3, pix = it >> 2, py = pix / 34, px = pix - py * 34;
        const int y = min(max(ty * 8 - 1 + py, 0), H - 1), x = min(max(tx * 32 - 1 + px, 0), W - 1);
        const uint4 v = src[(((size_t)n * H + y) * W + x) * 4 + s];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    void *a, *b;
    unsigned* out;
    CK(hipMalloc(&a, BYTES));
    CK(hipMalloc(&b, BYTES));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, BYTES));
    CK(hipMemset(b, 2, BYTES));
    CK(hipDeviceSynchronize());
    const int G = 256 * 16, T = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read<uint4>, dim3(G), dim3(T), 0, 0, (const uint4*)a, out, BYTES / 16);
        hipLaunchKernelGGL(calib_read<uint2>, dim3(G), dim3(T), 0, 0, (const uint2*)a, out, BYTES / 8);
        hipLaunchKernelGGL(calib_read<unsigned>, dim3(G), dim3(T), 0, 0, (const unsigned*)a, out, BYTES / 4);
        hipLaunchKernelGGL(calib_read12, dim3(G), dim3(T), 0, 0, (const unsigned*)a, out, BYTES / 12);
        hipLaunchKernelGGL(calib_read16_of_64, dim3(G), dim3(T), 0, 0, (const uint4*)a, out, BYTES / 64);
        hipLaunchKernelGGL(calib_write<uint4>, dim3(G), dim3(T), 0, 0, (uint4*)b, BYTES / 16, 7u + rep);
        hipLaunchKernelGGL(calib_write<uint2>, dim3(G), dim3(T), 0, 0, (uint2*)b, BYTES / 8, 8u + rep);
        hipLaunchKernelGGL(calib_write<unsigned>, dim3(G), dim3(T), 0, 0, (unsigned*)b, BYTES / 4, 9u + rep);
        hipLaunchKernelGGL(calib_copy16, dim3(G), dim3(T), 0, 0, (const uint4*)a, (uint4*)b, BYTES / 16);
        // 64 images of 512 x 512 x 32 bf16 = 1 GiB
        hipLaunchKernelGGL(calib_read_tiles, dim3(512 / 32, 512 / 8, 64), dim3(256), 0, 0, (const uint4*)a, out, 512, 512, 64);
        CK(hipDeviceSynchronize());
    }
    printf("pmc_calib done: %zu bytes per kernel\n", BYTES);
    return 0;
}
