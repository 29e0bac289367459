// ram_dft.hip -- the ROW pass of Random Amplitude Mixup as a PRUNED DFT ON THE MATRIX CORES (uint8 images).
//
// The mix needs |kx| <= b of each row spectrum, b = floor(0.1 * side) (fundus.py:26): 41 of 201 bins at 400 x 400.  An FFT computes
// all of them on the vector units -- about 900 VALU instructions per image row however the stages are organised (ram_fft.hip's Stockham
// kernel and a version with two register stages per wave both measured 18-19 us for the 16 x 400 rows x 3 channels of a batch of 8
// pairs) -- while the kept bins alone are a small matrix product
//     X[kx][row] = sum_x  T[kx][x] * img[row][x],        T = (cos, -sin)(2 pi kx x / W),
// 1.3 GFLOP per batch, a microsecond of v_mfma_f32_32x32x16_bf16.  fp32 accuracy on bf16 operands by splitting: a coefficient is the
// sum of THREE bf16 terms (hi + mid + lo = 24 mantissa bits; the table is built once per geometry in fp64 by rd_ram_dft_tables), a
// uint8 pixel is exact in one bf16, so three products per coefficient accumulate in the fp32 accumulator what an fp32 FFT would round
// log2(W) times.  A zero channel gives an exactly zero spectrum (the |F_src| == 0 branch of the reference, fundus.py:48), as the
// real-input FFT did.  13 us per batch: what is left is not arithmetic (an empty launch of this grid measures 5 us, the products 2.5)
// but one memory round trip, the reduction and the write-back of the 6.8 MB of row spectra.
// Measured and NOT kept (docs/experiments.md, round 5): the same pass for fp32 pixels (three bf16 terms per pixel, six products:
// 22 us, the FFT is faster) and the row-inverse pass as a matrix product with the output epilogue fused (26 us against 21).
#include "common.h"
#include "ram_dft.h"
#include "../../include/ramdsir.h"

namespace {

__device__ __forceinline__ float bf16_round(float v) { return __uint_as_float(bf16_bits(v) << 16); }

// ---- coefficient tables (built once per geometry)
// row forward: A operand of (tile mt, k-step ks, term t): lane l -> output row m = l % 32 (bin kx = 16 mt + m / 2, part m % 2: cos / -sin),
// pixels x = 16 ks + 8 (l / 32) + e, e < 8
__global__ void ram_dft_row_fwd_table_kernel(uint4* tab, int W, int ntile, int nks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntile * nks * 64) return;
    const int lane = i & 63, ks = (i >> 6) % nks, mt = (i >> 6) / nks;
    const int m = lane & 31, kg = lane >> 5, kx = 16 * mt + (m >> 1), part = m & 1;
    unsigned short t[3][8];
    for (int e = 0; e < 8; ++e) {
        const int x = 16 * ks + 8 * kg + e;
        double s, c;
        sincospi(2.0 * (double)((kx * x) % W) / (double)W, &s, &c);
        const double v = part ? -s : c;
        const float h = bf16_round((float)v);
        const float md = bf16_round((float)(v - (double)h));
        const float lo = bf16_round((float)(v - (double)h - (double)md));
        t[0][e] = (unsigned short)(__float_as_uint(h) >> 16);
        t[1][e] = (unsigned short)(__float_as_uint(md) >> 16);
        t[2][e] = (unsigned short)(__float_as_uint(lo) >> 16);
    }
    for (int k = 0; k < 3; ++k)
        tab[((size_t)(mt * nks + ks) * 3 + k) * 64 + lane] =
            make_uint4(t[k][0] | (t[k][1] << 16), t[k][2] | (t[k][3] << 16), t[k][4] | (t[k][5] << 16), t[k][6] | (t[k][7] << 16));
}

struct RowFwdArgs {
    const void* src; const void* trg; float2* rowspec; const uint4* tab;
    int B, nimg, H, W, KP, ntile, nblk, nks;
};

__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A: a workgroup = (image, block of 32 rows, tile of 16 bins): C[m = (bin, re / im)][n = row] over k = the row's pixels, three
// channels.  Its four waves take every fourth 16-pixel k-step each -- all of a wave's loads (pixels and coefficients) are in flight
// at once, one memory round trip per wave -- and their partial accumulators are added in a fixed order through LDS.
// RF_NJ: k-steps per wave and pass = ceil(W / 64) for the usual sides (one pass); more passes beyond
template <int RF_NJ>
__global__ __launch_bounds__(256) void ram_row_dft_kernel(const RowFwdArgs a) {
    __shared__ float4 red[4][12][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x;
    const int mt = task % a.ntile, blk = (task / a.ntile) % a.nblk, n = task / (a.ntile * a.nblk);
    const int r = lane & 31, kg = lane >> 5, H = a.H, W = a.W;
    const int y = blk * 32 + r;
    const bool yok = y < H;
    const size_t roff = ((size_t)(n < a.B ? n : n - a.B) * H + (yok ? y : H - 1)) * W * 3 + 24 * kg;
    const void* base = n < a.B ? a.src : a.trg;
    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const uint4* ta = a.tab + (size_t)mt * a.nks * 3 * 64 + lane;
    for (int ks0 = wave; ks0 < a.nks; ks0 += 4 * RF_NJ) {
        {
            uint2 dw[RF_NJ][3];
            uint4 tw[RF_NJ][3];
#pragma unroll
            for (int j = 0; j < RF_NJ; ++j) {
                const int ks = min(ks0 + 4 * j, a.nks - 1);
                const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(base) + roff + 48 * ks);
                dw[j][0] = p[0]; dw[j][1] = p[1]; dw[j][2] = p[2];
#pragma unroll
                for (int t = 0; t < 3; ++t) tw[j][t] = ta[(ks * 3 + t) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);               // every load of the pass is requested before the first product waits for one
#pragma unroll
            for (int j = 0; j < RF_NJ; ++j) {
                const unsigned ok = (yok && ks0 + 4 * j < a.nks) ? 0xffffffffu : 0u;
                const unsigned d[6] = {dw[j][0].x & ok, dw[j][0].y & ok, dw[j][1].x & ok, dw[j][1].y & ok, dw[j][2].x & ok, dw[j][2].y & ok};
                float f[24];                                 // pixels 16 ks + 8 kg .. + 8, interleaved channels
#pragma unroll
                for (int i = 0; i < 24; ++i) f[i] = (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint4 x = make_uint4(Slot<bf16_t>::pk2(f[c], f[3 + c]), Slot<bf16_t>::pk2(f[6 + c], f[9 + c]),
                                               Slot<bf16_t>::pk2(f[12 + c], f[15 + c]), Slot<bf16_t>::pk2(f[18 + c], f[21 + c]));
                    acc[c] = mfma(tw[j][0], x, acc[c]);
                    acc[c] = mfma(tw[j][1], x, acc[c]);
                    acc[c] = mfma(tw[j][2], x, acc[c]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            red[wave][c * 4 + qd][lane] = make_float4(acc[c][4 * qd], acc[c][4 * qd + 1], acc[c][4 * qd + 2], acc[c][4 * qd + 3]);
    __syncthreads();
    // C layout: column n = lane % 32 (the row), rows m = 8 q + 4 (lane / 32) + i: bins 4 q + 2 kg, + 1 as (re, im, re, im); wave w adds and
    // stores quads 3 w .. 3 w + 2 of the twelve (channel, q)
    if (!yok) return;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int s_ = wave * 3 + s3, c = s_ >> 2, qd = s_ & 3;
        const float4 v0 = red[0][s_][lane], v1 = red[1][s_][lane], v2 = red[2][s_][lane], v3 = red[3][s_][lane];
        const int kx = 16 * mt + 4 * qd + 2 * kg;
        if (kx < a.KP)
            *reinterpret_cast<float4*>(a.rowspec + (((size_t)n * 3 + c) * H + y) * a.KP + kx) =
                make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w));
    }
}

}  // namespace

int ram_dft_row_fwd(const void* src, const void* trg, int B, int nimg, int H, int W, int b, int KP, float2* rowspec,
                    const void* tables, hipStream_t st) {
    const RamDftGeom g = ram_dft_geom(H, W, b);
    const RamDftLayout l = ram_dft_layout(g);
    RowFwdArgs a;
    a.src = src; a.trg = trg; a.rowspec = rowspec;
    a.tab = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tables) + l.row_fwd);
    a.B = B; a.nimg = nimg; a.H = H; a.W = W; a.KP = KP; a.ntile = g.ntile; a.nblk = (H + 31) / 32; a.nks = g.nks_w;
    const int tasks = nimg * a.nblk * a.ntile;
    const int nj = (a.nks + 3) / 4;
#define RD_RF(NJ) rd_launch((ram_row_dft_kernel<NJ>), dim3(tasks), dim3(256), 0, st, a)
    if (nj <= 4) RD_RF(4);
    else if (nj <= 6) RD_RF(6);
    else if (nj == 7) RD_RF(7);
    else RD_RF(8);
#undef RD_RF
    return (int)hipGetLastError();
}

extern "C" {

int64_t rd_ram_dft_tables_bytes(int H, int W, int b) {
    if (!ram_dft_ok(H, W, b)) return 0;
    return (int64_t)ram_dft_layout(ram_dft_geom(H, W, b)).total;
}

int rd_ram_dft_tables(void* tables, int H, int W, int b, void* stream) {
    if (!tables || !ram_dft_ok(H, W, b)) return -1;
    const RamDftGeom g = ram_dft_geom(H, W, b);
    const RamDftLayout l = ram_dft_layout(g);
    hipStream_t st = (hipStream_t)stream;
    const int n = g.ntile * g.nks_w * 64;
    rd_launch(ram_dft_row_fwd_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(reinterpret_cast<char*>(tables) + l.row_fwd), W, g.ntile, g.nks_w);
    return (int)hipGetLastError();
}

}  // extern "C"
