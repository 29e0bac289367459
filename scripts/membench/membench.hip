// membench.hip -- read-bandwidth probes for the HBM-bound conv kernels (standalone; hipcc --offload-arch=gfx950 membench.hip -o membench).
// Question: what limits a loader of 4 waves per CU with 24 x 1 KB loads in flight each (conv_small_fwd_ws_kernel) to ~2.8 TB/s when
// torch's elementwise kernels stream at 5-6 TB/s?  Variants: waves per CU, loads in flight per wave, linear vs tile-row addressing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// linear: every workgroup streams a contiguous slice; a wave reads U x 1 KB per round, all in flight before the first use
template <int U>
__global__ void linear_kernel(const uint4* __restrict__ src, unsigned* out, size_t n16, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t per_wg = n16 / gridDim.x;
    const uint4* base = src + (size_t)blockIdx.x * per_wg;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = base[((size_t)(r * nw + wave) * U + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// pipelined linear: NSET register sets of U loads, consumed one set per iteration (the loader's structure)
template <int U, int NSET>
__global__ __launch_bounds__(512) void pipe_kernel(const uint4* __restrict__ src, unsigned* out, size_t n16, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t per_wg = n16 / gridDim.x;
    const uint4* base = src + (size_t)blockIdx.x * per_wg;
    unsigned acc = 0;
    uint4 v[NSET][U];
    auto issue = [&](uint4 (&r)[U], int j) {
        const int jj = j < rounds ? j : rounds - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = base[((size_t)(jj * nw + wave) * U + u) * 64 + lane];
    };
#pragma unroll
    for (int k = 0; k < NSET; ++k) issue(v[k], k);
    for (int r0 = 0; r0 < rounds; r0 += NSET) {
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[k][u].x ^ v[k][u].y ^ v[k][u].z ^ v[k][u].w;
            issue(v[k], r0 + k + NSET);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// tile rows: an image of H x W pixels x 64 B; a workgroup walks tiles of 10 rows x 34 pixels (halo tile of an 8 x 32 output tile),
// consecutive tiles along x; 256 threads: thread -> (pixel = (tid + 256 b) / 4, 16-byte slot = tid & 3), 6 items
template <int NSET, int MAP = 0>
__global__ __launch_bounds__(256) void tile_kernel(const uint4* __restrict__ src, unsigned* out, int H, int W, int tiles_per_wg, int interleave) {
    const int tid = threadIdx.x;
    const int tiles_x = (W + 31) / 32, ntiles = tiles_x * ((H + 7) / 8);
    const int n = blockIdx.z;
    const uint4* img = src + (size_t)n * H * W * 4;
    int py[6], px[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        int pix = (tid + 256 * b) >> 2;
        if (MAP == 1) pix = (tid >> 6) * 85 + b * 16 + ((tid & 63) >> 2);       // a wave reads 85 consecutive halo pixels (2.5 rows)
        if (pix > 339) pix = 339;
        py[b] = pix / 34;
        px[b] = pix - py[b] * 34;
    }
    const int nt_all = interleave ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x
                                  : min(ntiles, ((int)blockIdx.x + 1) * tiles_per_wg) - min(ntiles, (int)blockIdx.x * tiles_per_wg);
    unsigned acc = 0;
    uint4 v[NSET][6];
    auto issue = [&](uint4 (&r)[6], int j) {
        const int jj = j < nt_all ? j : nt_all - 1;
        const int t = interleave ? blockIdx.x + jj * gridDim.x : blockIdx.x * tiles_per_wg + jj;
        const int y0 = (t / tiles_x) * 8 - 1, x0 = (t % tiles_x) * 32 - 1;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int y = min(max(y0 + py[b], 0), H - 1), x = min(max(x0 + px[b], 0), W - 1);
            r[b] = img[(size_t)(y * W + x) * 4 + (tid & 3)];
        }
    };
#pragma unroll
    for (int k = 0; k < NSET; ++k) issue(v[k], k);
    for (int r0 = 0; r0 < nt_all; r0 += NSET) {
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
#pragma unroll
            for (int u = 0; u < 6; ++u) acc ^= v[k][u].x ^ v[k][u].y ^ v[k][u].z ^ v[k][u].w;
            issue(v[k], r0 + k + NSET);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the loader inside a two-role workgroup: 512 threads, waves 4-7 load (tile_kernel's body), waves 0-3 only meet them at the per-tile
// barrier; WORK: the idle role spins on VALU work between barriers (what MFMA waves would do)
template <int NSET, int WORK>
__global__ __launch_bounds__(512) void role_kernel(const uint4* __restrict__ src, unsigned* out, int H, int W, int tiles_per_wg) {
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int tid = threadIdx.x & 255;
    const int tiles_x = (W + 31) / 32, ntiles = tiles_x * ((H + 7) / 8);
    const int n = blockIdx.z;
    const uint4* img = src + (size_t)n * H * W * 4;
    const int nt_all = min(ntiles, ((int)blockIdx.x + 1) * tiles_per_wg) - min(ntiles, (int)blockIdx.x * tiles_per_wg);
    unsigned acc = 0;
    if (role == 1) {
        int py[6], px[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            int pix = (tid + 256 * b) >> 2;
            if (pix > 339) pix = 339;
            py[b] = pix / 34;
            px[b] = pix - py[b] * 34;
        }
        uint4 v[NSET][6];
        auto issue = [&](uint4 (&r)[6], int j) {
            const int jj = j < nt_all ? j : nt_all - 1;
            const int t = blockIdx.x * tiles_per_wg + jj;
            const int y0 = (t / tiles_x) * 8 - 1, x0 = (t % tiles_x) * 32 - 1;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int y = min(max(y0 + py[b], 0), H - 1), x = min(max(x0 + px[b], 0), W - 1);
                r[b] = img[(size_t)(y * W + x) * 4 + (tid & 3)];
            }
        };
#pragma unroll
        for (int k = 0; k < NSET; ++k) issue(v[k], k);
        for (int r0 = 0; r0 < nt_all; r0 += NSET) {
#pragma unroll
            for (int k = 0; k < NSET; ++k) {
#pragma unroll
                for (int u = 0; u < 6; ++u) acc ^= v[k][u].x ^ v[k][u].y ^ v[k][u].z ^ v[k][u].w;
                issue(v[k], r0 + k + NSET);
                if (r0 + k < nt_all) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    } else {
        float f = (float)tid;
        for (int it = 0; it < nt_all; ++it) {
            for (int w = 0; w < WORK; ++w) f = f * 1.0001f + 0.5f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        acc = __float_as_uint(f);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static double time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const int N = 8, H = 400, W = 400;
    const size_t bytes = (size_t)N * H * W * 64;                // 82 MB: one 32-channel bf16 layer
    // rotate over several buffers so that the 256 MB infinity cache does not serve the re-runs
    const int NBUF = 6;
    std::vector<uint4*> bufs(NBUF);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    unsigned* out;
    CK(hipMalloc(&out, 64));
    int rot = 0;
    auto next = [&]() { rot = (rot + 1) % NBUF; return bufs[rot]; };
    const size_t n16 = bytes / 16;
    printf("82 MB per launch, %d buffers in rotation\n", NBUF);
#define LIN(U, WG, TH) do { \
        const int rounds = (int)(n16 / (WG) / ((TH) / 64) / (U) / 64); \
        double us = time_us([&]() { hipLaunchKernelGGL((linear_kernel<U>), dim3(WG), dim3(TH), 0, 0, next(), out, n16, rounds); }); \
        printf("linear   U=%2d  %5d wgs x %4d thr: %7.1f us  %5.2f TB/s\n", U, WG, TH, us, (double)rounds * (WG) * ((TH) / 64) * (U) * 1024 / us / 1e6); \
    } while (0)
    LIN(4, 256, 256);
    LIN(8, 256, 256);
    LIN(24, 256, 256);
    LIN(4, 256, 1024);
    LIN(8, 256, 1024);
    LIN(4, 2048, 256);
    LIN(4, 8192, 256);
    LIN(1, 8192, 256);
#define PIPE(U, NSET, WG, TH) do { \
        const int rounds = (int)(n16 / (WG) / ((TH) / 64) / (U) / 64); \
        double us = time_us([&]() { hipLaunchKernelGGL((pipe_kernel<U, NSET>), dim3(WG), dim3(TH), 0, 0, next(), out, n16, rounds); }); \
        printf("pipe     U=%2d x %d sets  %5d wgs x %4d thr: %7.1f us  %5.2f TB/s\n", U, NSET, WG, TH, us, (double)rounds * (WG) * ((TH) / 64) * (U) * 1024 / us / 1e6); \
    } while (0)
    PIPE(6, 4, 256, 256);
    PIPE(6, 8, 256, 256);
    PIPE(6, 4, 256, 512);
    PIPE(6, 4, 512, 256);
    PIPE(6, 4, 1024, 256);
    PIPE(6, 2, 2048, 256);
#define TILE(NSET, TPW, IL) TILEM(NSET, 0, TPW, IL)
#define TILEM(NSET, MAP, TPW, IL) do { \
        const int ntiles = 13 * 50, gx = (ntiles + (TPW) - 1) / (TPW); \
        double us = time_us([&]() { hipLaunchKernelGGL((tile_kernel<NSET, MAP>), dim3(gx, 1, N), dim3(256), 0, 0, next(), out, H, W, TPW, IL); }); \
        printf("tilerows map %d, %d sets  %4d wgs (%2d tiles each%s): %7.1f us  %5.2f TB/s of tensor bytes\n", MAP, NSET, gx * N, TPW, (IL) ? ", interleaved" : "", us, (double)bytes / us / 1e6); \
    } while (0)
    TILE(4, 21, 0);
    TILE(4, 21, 1);
    TILE(8, 21, 0);
    TILE(4, 11, 0);
    TILE(4, 6, 0);
    TILE(2, 3, 0);
    TILE(2, 1, 0);
    TILE(1, 21, 0);
    TILE(2, 21, 0);
    TILE(3, 21, 0);
    TILEM(1, 1, 21, 0);
    TILEM(2, 1, 21, 0);
    TILEM(4, 1, 21, 0);
    TILEM(2, 1, 3, 0);
#define ROLE(NSET, WORK, TPW) do { \
        const int ntiles = 13 * 50, gx = (ntiles + (TPW) - 1) / (TPW); \
        double us = time_us([&]() { hipLaunchKernelGGL((role_kernel<NSET, WORK>), dim3(gx, 1, N), dim3(512), 0, 0, next(), out, H, W, TPW); }); \
        printf("two roles, %d sets, idle role spins %4d FMAs per tile  %4d wgs: %7.1f us  %5.2f TB/s of tensor bytes\n", NSET, WORK, gx * N, us, (double)bytes / us / 1e6); \
    } while (0)
    ROLE(4, 0, 21);
    ROLE(2, 0, 21);
    ROLE(4, 200, 21);
    ROLE(4, 1000, 21);
    for (auto b : bufs) printf("buffer %p\n", (void*)b);
    return 0;
}
