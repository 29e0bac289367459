// conv_big.hip -- direct (im2col-free) 3x3 / 1x1 convolution on gfx950 MFMA: the generic multi-chunk forward / dgrad kernel.
//
// Replaces the ATen/cuDNN conv2d forward/dgrad/wgrad the reference reaches through nn.Conv2d
// (code/networks/unet.py:37-43,81-88,124-131,281,307) together with everything that sits between two
// convs in the reference graph -- BatchNorm apply, ReLU/LeakyReLU, MaxPool2d(2), bilinear x2,
// torch.cat -- which is folded into the tile loader (forward) or the epilogue (backward).
//
// Implicit GEMM, M = pixels, N = output channels, K = taps x input channels:
//   workgroup  = 256 threads = 4 wave64, output tile 8 rows x 32 columns of one image x NT channels
//   wave w     = tile rows 2w, 2w+1  -> two 32-pixel M-blocks; NB = NT/32 N-blocks
//   LDS        = halo tile (10 x 34 pixels) x 64 B of channels (16 fp32 / 32 bf16 per chunk),
//                16-byte slots XOR-swizzled by (pixel>>2)&3 so that ds_read_b128 of 16 consecutive
//                pixels is bank-conflict free; the weight chunk [tap][n][64 B] swizzled the same way
//   MFMA       = v_mfma_f32_32x32x16_bf16 (bf16) / v_mfma_f32_32x32x2_f32 (fp32, bit-exact fmaf chain)

#include "conv_device.h"
#include "conv_epilogue.h"
#include "conv_dispatch.h"
#include "conv_pf.h"

namespace {

// ------------------------------------------------------------------------------------ conv kernel
template <typename T, int TAPS, int NB>
__global__ __launch_bounds__(256, 2) void conv_kernel(const rd_conv_t p, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                                // BatchNorm finalize folded into this launch (bn_fin.h)
    constexpr int S = Slot<T>::N;
    constexpr int CK = 4 * S;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NT = NB * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [PH*PW][4]
    uint4* s_w = s_in + PH * PW * 4;                       // [TAPS][NT][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW;
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int x0 = (bx % tiles_x) * TW, y0 = (bx / tiles_x) * TH;
    const int n0 = by * NT;
    const int n = bz;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int H = p.H, W = p.W;
    const int slot = (bx + 7 * bz) % rd_stat_nslots(p.stat_slots);

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const T* wbase = reinterpret_cast<const T*>(p.w);
    for (int c0 = 0; c0 < p.CinPad; c0 += CK) {
        __syncthreads();
        {
            const int s = tid & 3;                         // idx & 3 is constant per thread (stride 256)
            SlotCtx<T> ctx;
            slot_ctx<T>(ctx, p.src, p.nsrc, p.Cin, g, c0 + s * S);
            auto map = [&](int idx, int& y, int& x) -> bool {
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                y = y0 - HALO + py;
                x = x0 - HALO + px;
                return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            };
            auto store = [&](int idx, const uint4& u) {
                const int pix = idx >> 2;
                s_in[pix * 4 + (s ^ ((pix >> 2) & 3))] = u;
            };
            tile_fill<T, 256, false>(p.src, ctx, n, H, W, tid, PH * PW * 4, map, store);
        }
        {
            constexpr int WTOT = TAPS * NT * 4, WIT = (WTOT + 255) / 256;
            uint4 wr[WIT];
#pragma unroll
            for (int b = 0; b < WIT; ++b) {
                const int idx = tid + b * 256;
                const int s = idx & 3, rec = idx >> 2;
                const int nn = rec % NT, tap = rec / NT;
                wr[b] = ld16(wbase + ((size_t)((c0 / CK) * TAPS + min(tap, TAPS - 1)) * p.CoutPad + n0 + nn) * CK + s * S);
            }
#pragma unroll
            for (int b = 0; b < WIT; ++b) {
                const int idx = tid + b * 256;
                const int s = idx & 3, rec = idx >> 2;
                const int nn = rec % NT;
                if (idx < WTOT) s_w[rec * 4 + (s ^ ((nn >> 2) & 3))] = wr[b];
            }
        }
        __syncthreads();
        conv_mma_chunk<T, TAPS, NB>(s_in, s_w, wave, li, h, acc);
    }

    conv_epilogue<T, NB>(p, acc, smem, tid, n, g, y0, x0, n0, slot);
}

template <typename T, int TAPS, int NB>
int launch_conv(const rd_conv_t& p, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    const size_t lds = conv_pf_lds<TAPS, NB>(p);
    dim3 grid(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH), p.CoutPad / (NB * 32), p.N);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, TAPS, NB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        attr_set = true;
    }
    if constexpr (sizeof(T) == 2 && TAPS == 9 && NB == 2) {
        static const bool pp_off = rd_switch("RD_CONV_PP_OFF", 0) != 0;
        if (!pp_off) {
            const int rc = rd_conv_pp_dispatch(p, st);
            if (rc != RD_CONV_PP_NA) return rc;
        }
    }
    if constexpr (sizeof(T) == 2) {
        static const bool lean_off = rd_switch("RD_CONV_PF_LEAN_OFF", 0) != 0;
        if (!lean_off && rd_switch("RD_CONV_PF_OFF", 0) == 0) {
            const int rc = rd_conv_pf_lean_dispatch(p, NB == 2, st);      // register epilogues (conv_lean.hip)
            if (rc != RD_CONV_PP_NA) return rc;
        }
    }
    if constexpr (sizeof(T) == 2) {
        static const bool pf_off = rd_switch("RD_CONV_PF_OFF", 0) != 0;
        const int nq = conv_pf_kind(p);
        if (!pf_off && nq && lds <= (size_t)72 * 1024) {
            static bool attr_pf = false;
            if (!attr_pf) {
                const int lds_max = 72 * 1024;                  // two workgroups per CU; covers CinPad <= 1024
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                attr_pf = true;
            }
            if (nq == 1) rd_launch((conv_pf_kernel<T, TAPS, NB, 1>), grid, dim3(256), lds, st, p, rdfin::current());
            else rd_launch((conv_pf_kernel<T, TAPS, NB, 2>), grid, dim3(256), lds, st, p, rdfin::current());
            return (int)hipGetLastError();
        }
    }
    rd_launch((conv_kernel<T, TAPS, NB>), grid, dim3(256), lds, st, p, rdfin::current());
    return (int)hipGetLastError();
}

}  // namespace

static bool conv_big_nb2(const rd_conv_t& p, int dtype) {
    bool nb2 = (p.CoutPad % 64) == 0;
    // small grids (the 25x25 / 50x50 levels): 32-channel tiles double the number of workgroups
    static const int nb1_below = rd_switch("RD_CONV_NB1_BELOW", 300);
    const int wgs64 = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.N * (p.CoutPad / 64);
    if (nb2 && dtype == RD_BF16 && wgs64 < nb1_below) nb2 = false;
    return nb2;
}

bool rd_conv_big_takes_ws(const rd_conv_t& p, int dtype) {
    static const bool pp_off = rd_switch("RD_CONV_PP_OFF", 0) != 0;
    return dtype == RD_BF16 && p.taps == 9 && conv_big_nb2(p, dtype) && !pp_off && rd_conv_ws_takes(p);
}

bool rd_conv_big_stores_sources(const rd_conv_t& p, int dtype) {
    return rd_conv_big_takes_ws(p, dtype) && rd_conv_ws_stores_sources(p);
}

int rd_conv_big_dispatch(const rd_conv_t& p, int dtype, hipStream_t st) {
    const bool nb2 = conv_big_nb2(p, dtype);
    if (dtype == RD_BF16) {
        if (p.taps == 9) return nb2 ? launch_conv<bf16_t, 9, 2>(p, st) : launch_conv<bf16_t, 9, 1>(p, st);
        return nb2 ? launch_conv<bf16_t, 1, 2>(p, st) : launch_conv<bf16_t, 1, 1>(p, st);
    }
    if (p.taps == 9) return nb2 ? launch_conv<float, 9, 2>(p, st) : launch_conv<float, 9, 1>(p, st);
    return nb2 ? launch_conv<float, 1, 2>(p, st) : launch_conv<float, 1, 1>(p, st);
}

