// wgrad.hip -- weight-gradient kernels (K = pixels): the generic one (fp32; bf16 with ragged channel counts), the bf16 kernels
// on [pixel][channel] LDS tiles read through ds_read_b64_tr_b16 (32x32x16 MFMA blocks, and 16x16x32 blocks for <= 16 output
// channels); deterministic two-stage split reduction.
#include "conv_device.h"
#include "conv_dispatch.h"
#include "wgrad_tr.h"

#ifndef RD_WG_SYM_DEFAULT
#define RD_WG_SYM_DEFAULT 1          // (A/B builds: -DRD_WG_SYM_DEFAULT=0 keeps the 64 x 64-block kernels for every layer)
#endif

namespace {

// ------------------------------------------------------------------------------------ wgrad kernel
// dW[tap][n][c] = sum over pixels of dz[p][n] * a[p + tap][c];  M = n (Cout), N = c (Cin), K = pixels.
// A workgroup owns a (MB*32) x (NB*32) x TAPS block of dW and walks pixel tiles with stride gridDim.x;
// its 4 waves are MB*NB output blocks x KS = 4/(MB*NB) pixel-row splits.  Each wave stores its partial
// block; a second kernel reduces the splits in a fixed order (deterministic, no atomics).
template <typename T, int TAPS, int MB, int NB>
__global__ __launch_bounds__(256) void wgrad_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                               // BatchNorm-backward finalize folded into this launch (bn_fin.h)
    constexpr int S = Slot<T>::N;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int CA = NB * 32, CZ = MB * 32;
    constexpr int KS = 4 / (MB * NB);
    constexpr int ROWS = TH / KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* s_a = reinterpret_cast<T*>(smem);                   // [PH*PW][CA]
    T* s_z = s_a + PH * PW * CA;                           // [TH*TW][CZ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int kq = wave / (MB * NB), blk = wave % (MB * NB);
    const int mb = blk / NB, nb = blk % NB;
    const int nbase = blockIdx.y * CZ, cbase = blockIdx.z * CA;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    SlotCtx<T> ctx_a, ctx_z;
    int g_ctx = -1;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const int g = group_of(gm, n);
        __syncthreads();
        if (g != g_ctx) {                                  // BN coefficients depend on the image's group only
            slot_ctx<T>(ctx_a, p.a, p.na, p.Cin, g, cbase + (tid % (CA / S)) * S);
            slot_ctx<T>(ctx_z, &p.dz, 1, p.Cout, g, nbase + (tid % (CZ / S)) * S);
            g_ctx = g;
        }
        {
            const int s = tid % (CA / S);                  // constant per thread: 256 % (CA/S) == 0
            const SlotCtx<T>& ctx = ctx_a;
            auto map = [&](int idx, int& y, int& x) -> bool {
                const int pix = idx / (CA / S);
                const int py = pix / PW, px = pix - py * PW;
                y = y0 - HALO + py;
                x = x0 - HALO + px;
                return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            };
            auto store = [&](int idx, const uint4& u) {
                const int pix = idx / (CA / S);
                *reinterpret_cast<uint4*>(s_a + pix * CA + s * S) = u;
            };
            tile_fill<T>(p.a, ctx, n, H, W, tid, PH * PW * (CA / S), map, store);
        }
        {
            const int s = tid % (CZ / S);
            const SlotCtx<T>& ctx = ctx_z;
            auto map = [&](int idx, int& y, int& x) -> bool {
                const int pix = idx / (CZ / S);
                const int py = pix / TW, px = pix - py * TW;
                y = y0 + py;
                x = x0 + px;
                return y < H && x < W;
            };
            auto store = [&](int idx, const uint4& u) {
                const int pix = idx / (CZ / S);
                *reinterpret_cast<uint4*>(s_z + pix * CZ + s * S) = u;
            };
            tile_fill<T>(&p.dz, ctx, n, H, W, tid, TH * TW * (CZ / S), map, store);
        }
        __syncthreads();
        if constexpr (sizeof(T) == 2) {
            const unsigned short* za = reinterpret_cast<const unsigned short*>(s_z) + mb * 32 + li;
            const unsigned short* aa = reinterpret_cast<const unsigned short*>(s_a) + nb * 32 + li;
            for (int rr = 0; rr < ROWS; ++rr) {
                const int row = kq + rr * KS;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int px0 = ks * 16 + 8 * h;                 // this lane-half's 8 pixels (k = 8h+e)
                    unsigned zp[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned lo = za[(row * TW + px0 + 2 * e) * CZ];
                        const unsigned hi = za[(row * TW + px0 + 2 * e + 1) * CZ];
                        zp[e] = lo | (hi << 16);
                    }
                    const bf16x8 afrag = __builtin_bit_cast(bf16x8, make_uint4(zp[0], zp[1], zp[2], zp[3]));
#pragma unroll
                    for (int kh = 0; kh < (TAPS == 9 ? 3 : 1); ++kh) {
                        const int base = (row + kh) * PW + px0;     // halo coords: input pixel = output pixel + tap
                        if constexpr (TAPS == 9) {
                            unsigned v[10];
#pragma unroll
                            for (int e = 0; e < 10; ++e) v[e] = aa[(base + e) * CA];
                            unsigned P[5], Q[4];
#pragma unroll
                            for (int e = 0; e < 5; ++e) P[e] = v[2 * e] | (v[2 * e + 1] << 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) Q[e] = v[2 * e + 1] | (v[2 * e + 2] << 16);
                            acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                afrag, __builtin_bit_cast(bf16x8, make_uint4(P[0], P[1], P[2], P[3])), acc[kh * 3 + 0], 0, 0, 0);
                            acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                afrag, __builtin_bit_cast(bf16x8, make_uint4(Q[0], Q[1], Q[2], Q[3])), acc[kh * 3 + 1], 0, 0, 0);
                            acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                afrag, __builtin_bit_cast(bf16x8, make_uint4(P[1], P[2], P[3], P[4])), acc[kh * 3 + 2], 0, 0, 0);
                        } else {
                            unsigned P[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                P[e] = (unsigned)aa[(base + 2 * e) * CA] | ((unsigned)aa[(base + 2 * e + 1) * CA] << 16);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                afrag, __builtin_bit_cast(bf16x8, make_uint4(P[0], P[1], P[2], P[3])), acc[0], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
            const float* za = reinterpret_cast<const float*>(s_z) + mb * 32 + li;
            const float* aa = reinterpret_cast<const float*>(s_a) + nb * 32 + li;
            for (int rr = 0; rr < ROWS; ++rr) {
                const int row = kq + rr * KS;
#pragma unroll 4
                for (int s2 = 0; s2 < 16; ++s2) {
                    const int px = 2 * s2 + h;                       // k = h
                    const float av = za[(row * TW + px) * CZ];
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
                        const float bv = aa[((row + kh) * PW + px + kw) * CA];
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
    }
    // waves that split the pixel rows of the tiles (kq > 0) fold their accumulators into wave kq == 0
    // through LDS, one tap per round; then one partial block per workgroup: partial[split][tap][n][c]
    if constexpr (KS > 1) {
        float* s_acc = reinterpret_cast<float*>(smem);     // [(KS-1)][MB*NB][16][64]
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            __syncthreads();
            if (kq > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_acc[(((kq - 1) * (MB * NB) + blk) * 16 + r) * 64 + lane] = acc[tap][r];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int k2 = 0; k2 < KS - 1; ++k2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r] += s_acc[((k2 * (MB * NB) + blk) * 16 + r) * 64 + lane];
            }
        }
    }
    if (kq == 0) {
        float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = nbase + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ccol = cbase + nb * 32 + li;
                out[((size_t)tap * CoutPadW + nrow) * CinPadW + ccol] = acc[tap][r];
            }
    }
}

// ------------------------------------------------------------------------------------ bf16 wgrad, hardware transpose read
// Same blocks / waves / MFMAs / split-K as wgrad_t_kernel, but the LDS tiles keep the memory layout [pixel][channel]:
// the fill is ONE ds_write_b128 per 16-byte channel slot (lanes run slot-fastest: 64-128 contiguous bytes per pixel
// from HBM and into LDS) instead of eight ds_write_b16 + eight extractions, and the K = pixels fragments of both
// MFMA operands come from gfx950's transposing LDS read: in a 16-lane group lane i points ds_read_b64_tr_b16 at
// pixel (i/4), channels 4*(i%4)..+3 of a 4-pixel x 16-channel block and receives channel i of the four pixels
// (probe: scripts/probe/tr_probe.hip).  The three horizontal taps of a kernel row share three such reads of 4
// pixels each (12 >= 8+2), the odd tap is 4 v_alignbit as before.  Plain sources only (PlainSrc), tile-ahead prefetch.
// ROW PITCH AND SWIZZLE.  The LDS serves a ds_read_b64_tr_b16 in two 32-lane halves over 64 banks (bank = (a/4) mod 64: 256 B), and the
// 32 lanes of a half are the 16-lane groups of channel sub-blocks 0-15 and 16-31 of the SAME four pixels: 4 pixel rows x 64 contiguous
// bytes.  They are conflict-free only if the four rows tile the 256 B exactly.  Rounds 1-3 used pitches 144 B / 96 B (reasoned per
// 16-lane group: 4 x 32 B): rows 2 and 3 then overlapped rows 0 and 1 by half -- every transpose read took 3 LDS cycles instead of 2
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33 in wgrad_ws_kernel, profiles/r04_sq_counters.txt).  Now: 32-channel tiles at pitch
// 64 B (no padding: four pixels = 256 contiguous bytes); 64-channel tiles at pitch 128 B with the two 64-byte halves of a pixel
// SWAPPED where bit 1 of the pixel index is set -- of four consecutive pixels (any start) two then put the wanted half at 0 / 128 and
// two at 64 / 192 (mod 256).  A padded pitch of 192 B is conflict-free too but costs 32 KB more LDS per workgroup, and that cost
// 0.03 ms of step time (other lanes' workgroups no longer fit beside it; docs/experiments.md) for no gain of the kernel alone.
template <int TAPS, int MB, int NB, int NQZ, bool PF>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                               // BatchNorm-backward finalize folded into this launch (bn_fin.h)
    typedef bf16_t T;
    constexpr int S = 8;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NPIX = PH * PW;
    constexpr int CA = NB * 32, CZ = MB * 32;
    constexpr int PA = tr_pitch(CA), PZ = tr_pitch(CZ);     // bytes per pixel row
    constexpr int KS = 4 / (MB * NB);
    constexpr int ROWS = TH / KS;
    constexpr int NSA = CA / S, NSZ = CZ / S;               // channel slots per pixel
    constexpr int NITA = (NPIX * NSA + 255) / 256, NITZ = (TH * TW * NSZ) / 256;
    constexpr int KW = (TAPS == 9) ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_a = smem;                                       // [NPIX + 4][PA]
    char* s_z = smem + (NPIX + 4) * PA;                     // [TH*TW][PZ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int kq = wave / (MB * NB), blk = wave % (MB * NB);
    const int mb = blk / NB, nb = blk % NB;
    const int nbase = blockIdx.y * CZ, cbase = blockIdx.z * CA;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto coords = [&](int tile, int& n, int& y0, int& x0) {
        n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        y0 = (trem / tiles_x) * TH;
        x0 = (trem % tiles_x) * TW;
    };
    // ---- fill mapping: thread -> one channel slot of `a` and one of `dz` (slot fastest), pixels strided
    const int sla = tid % NSA, slz = tid % NSZ;
    const int ca_abs = cbase + sla * S, cz_abs = nbase + slz * S;
    const int sia = (p.na == 1 || ca_abs < p.a[0].C) ? 0 : 1;
    const rd_src_t sda = select_src(p.a, sia);
    const int ca = ca_abs - (sia ? p.a[0].C : 0);
    const bool live_a = ca_abs < p.Cin, live_z = cz_abs < p.Cout;
    PlainSrc<T> psa, psz;
    plain_src_init<T>(psa, sda, live_a ? ca : 0);
    plain_src_init<T>(psz, p.dz, live_z ? cz_abs : 0);
    ItemGeom<NITA> iga;
    ItemGeom<NITZ> igz;
#pragma unroll
    for (int b = 0; b < NITA; ++b) {
        const int pix = tid / NSA + (256 / NSA) * b, py = pix / PW, px = pix - py * PW;
        iga.py[b] = (short)py;
        iga.px[b] = (short)px;
        iga.lds[b] = pix < NPIX ? tr_off<CA>(pix, sla) : -1;
    }
#pragma unroll
    for (int b = 0; b < NITZ; ++b) {
        const int pix = tid / NSZ + (256 / NSZ) * b;
        igz.py[b] = (short)(pix / TW);
        igz.px[b] = (short)(pix % TW);
        igz.lds[b] = tr_off<CZ>(pix, slz);
    }
    uint4 raw_a[NITA][1], raw_z[NITZ][NQZ];
    auto issue = [&](int tile) {
        if constexpr (PF) {
            int n, y0, x0;
            coords(tile, n, y0, x0);
            if (live_a) pfu_issue<T, NITA>(raw_a, psa, iga, n, H, W, y0 - HALO, x0 - HALO);
            if (live_z) pfu_issue<T, NITZ>(raw_z, psz, igz, n, H, W, y0, x0);
        }
    };
    SlotCtx<T> ctx_a, ctx_z;                                // PF == false: generic (pooled / interpolated) sources, synchronous fill
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < ((NPIX + 4) * PA + TH * TW * PZ) / 16; i += 256) z4[i] = make_uint4(0, 0, 0, 0);
    }
    int g_ctx = -1;
    if ((int)blockIdx.x < total_tiles) issue(blockIdx.x);

    // ---- fragment addressing: 16-lane group gq = lane >> 4 -> 16-channel sub-block (gq & 1), K half (gq >> 1)
    const int i16 = lane & 15, gq = lane >> 4;
    const int pix0 = (gq >> 1) * 8 + (i16 >> 2), sub = ((gq & 1) * 16 + (i16 & 3) * 4) * 2;
    const int zoff = tr_frag<CZ>(pix0, mb, sub, 0);
    constexpr bool ROWFLIP = (PW % 4) == 2;                 // a halo row of odd index starts an odd multiple of 2 pixels further on
    const int aoff0 = tr_frag<CA>(pix0, nb, sub, 0), aoff1 = tr_frag<CA>(pix0, nb, sub, ROWFLIP ? 1 : 0);

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const int g = group_of(gm, n);
        if (g != g_ctx) {
            if constexpr (PF) {
                if (live_a) plain_src_coef<T>(psa, sda, g, ca);
                if (live_z) plain_src_coef<T>(psz, p.dz, g, cz_abs);
            } else {
                slot_ctx<T>(ctx_a, p.a, p.na, p.Cin, g, ca_abs);
                slot_ctx<T>(ctx_z, &p.dz, 1, p.Cout, g, cz_abs);
            }
            g_ctx = g;
        }
        __syncthreads();
        if constexpr (PF) {
            if (live_a)
                pfu_consume<T, NITA, 1>(raw_a, psa, iga, H, W, y0 - HALO, x0 - HALO,
                                        [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_a + l) = u; });
            if (live_z)
                pfu_consume<T, NITZ, NQZ>(raw_z, psz, igz, H, W, y0, x0,
                                          [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_z + l) = u; });
        } else {
            // idx = pixel * slots + slot; idx % slots is this thread's slot for every idx it visits (256 % slots == 0)
            tile_fill<T, 256, false>(p.a, ctx_a, n, H, W, tid, NPIX * NSA,
                [&](int idx, int& y, int& x) -> bool {
                    const int pix = idx / NSA, py = pix / PW, px = pix - py * PW;
                    y = y0 - HALO + py;
                    x = x0 - HALO + px;
                    return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                },
                [&](int idx, const uint4& u) { *reinterpret_cast<uint4*>(s_a + tr_off<CA>(idx / NSA, sla)) = u; });
            tile_fill<T, 256, false>(&p.dz, ctx_z, n, H, W, tid, TH * TW * NSZ,
                [&](int idx, int& y, int& x) -> bool {
                    const int pix = idx / NSZ;
                    y = y0 + pix / TW;
                    x = x0 + pix % TW;
                    return y < H && x < W;
                },
                [&](int idx, const uint4& u) { *reinterpret_cast<uint4*>(s_z + tr_off<CZ>(idx / NSZ, slz)) = u; });
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) issue(tile + gridDim.x);
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = kq + rr * KS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* zp = s_z + zoff + (row * TW + ks * 16) * PZ;
                const uint2 z0 = lds_tr(zp), z1 = lds_tr(zp + 4 * PZ);
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, make_uint4(z0.x, z0.y, z1.x, z1.y));
#pragma unroll
                for (int kh = 0; kh < KW; ++kh) {
                    const char* ap = s_a + (((row + kh) & 1) ? aoff1 : aoff0) + ((row + kh) * PW + ks * 16) * PA;     // halo coords: input = output + tap
                    const uint2 a0 = lds_tr(ap), a1 = lds_tr(ap + 4 * PA);
                    if constexpr (TAPS == 9) {
                        const uint2 a2 = lds_tr(ap + 8 * PA);
                        const uint4 dq = make_uint4(a0.x, a0.y, a1.x, a1.y);
                        const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(a0.y, a0.x, 16), __builtin_amdgcn_alignbit(a1.x, a0.y, 16),
                                                    __builtin_amdgcn_alignbit(a1.y, a1.x, 16), __builtin_amdgcn_alignbit(a2.x, a1.y, 16));
                        const uint4 m2 = make_uint4(a0.y, a1.x, a1.y, a2.x);
                        acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                        acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                        acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
                    } else {
                        const uint4 dq = make_uint4(a0.x, a0.y, a1.x, a1.y);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[0], 0, 0, 0);
                    }
                }
            }
        }
    }
    if constexpr (KS > 1) {
        float* s_acc = reinterpret_cast<float*>(smem);     // [(KS-1)][MB*NB][16][64]
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            __syncthreads();
            if (kq > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_acc[(((kq - 1) * (MB * NB) + blk) * 16 + r) * 64 + lane] = acc[tap][r];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int k2 = 0; k2 < KS - 1; ++k2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r] += s_acc[((k2 * (MB * NB) + blk) * 16 + r) * 64 + lane];
            }
        }
    }
    if (kq == 0) {
        float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = nbase + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ccol = cbase + nb * 32 + li;
                out[((size_t)tap * CoutPadW + nrow) * CinPadW + ccol] = acc[tap][r];
            }
    }
}

// Warp-specialised twin of wgrad_tr_kernel<9, 2, 2, NQZ, true> (3x3 taps, 64 x 64 channel blocks, plain sources): 512
// threads = two waves per SIMD with different jobs.  Waves 0-3 own one 32 x 32 quadrant each (144 accumulator registers) and
// do nothing but transpose reads + MFMAs; waves 4-7 fill the OTHER LDS buffer with the next tile (BN affine + activation of
// `a`, the dz operand(s)) and keep the tile after that in flight.  The single-role kernel runs fill and MFMAs back to back
// at one wave per SIMD (SQ counters: 47% of the wave cycles issuing, 20% issue-stalled: profiles/r02_sq_counters.txt).
// LDS tiles are 4 rows x 32 pixels (half of the single-role kernel's) so that two buffers fit: 2 x 47 KB.
// XP: timing experiments of the debug build, compile-time so that they do not change the code around them (RD_WGWS_EXP; wrong results
// when set): 1 no MFMA phase, 4 loader does not transform / write LDS
#ifdef RD_DEBUG_SWITCHES
// debug build: shader-clock stamps of workgroup (5, 0, 0) of the launches whose total tile count equals wg_trace_key
// (rd_debug_wg_trace; scripts/wg_trace.py): [role][iteration][event]
__device__ unsigned long long wg_trace[2][64][4];
__device__ int wg_trace_key;
#define WG_T(role, it, ev) do { \
        if (wg_trace_key == total_tiles && blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 255) == 0 && (it) < 64) \
            wg_trace[role][it][ev] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define WG_T(role, it, ev) do { } while (0)
#endif
template <int NQZ, int XP>
__global__ __launch_bounds__(512, 1) void wgrad_ws_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles, const rdfin::FinArg fa) {
    // BatchNorm-backward finalize folded into this launch (bn_fin.h).  Beside a gradient launch that owns it (the usual case) this
    // kernel derives P / Q / R of ITS 64 gradient channels straight into its coefficient table below -- no stores, no read-back; as
    // the owner (no gradient launch: not a geometry of this kernel in the network) it takes the global path
    const bool fin_tab = fa.which >= 0 && !(fa.flags & RD_FIN_OWNER);
    if (!fin_tab) rdfin::prologue(fa);
    typedef bf16_t T;
    constexpr int S = 8, TAPS = 9;
    constexpr int THW = 4;
    constexpr int PH = THW + 2, PW = TW + 2, NPIX = PH * PW;
    constexpr int CA = 64, CZ = 64, PA = tr_pitch(64), PZ = tr_pitch(64);
    constexpr int NSA = CA / S, NSZ = CZ / S;
    // XP & 8 (timing build): five `a` items per thread instead of seven = what an LDS row ring per vertical strip would leave (only the
    // four new halo rows per tile)
    constexpr int NITA = (XP & 8) ? 5 : (NPIX * NSA + 255) / 256, NITZ = (THW * TW * NSZ) / 256;
    // each buffer ends in a 4 KB dummy record (16 bytes per loader thread): items of dead lanes (channel slots beyond Cin / Cout, the
    // last item's pixels beyond the tile) are written THERE, so that the loader's fill has no divergent branch per item
    constexpr int A_BYTES = (NPIX + 4) * PA, Z_BYTES = THW * TW * PZ, DUMMY = 256 * 16, BUF = A_BYTES + Z_BYTES + DUMMY;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_coef = reinterpret_cast<float*>(smem + 2 * BUF);           // [G][a, z][sc, sh, q][64]: the loader's coefficients

    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // 0: MFMA waves, 1: loader waves
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int nbase = blockIdx.y * CZ, cbase = blockIdx.z * CA;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + THW - 1) / THW;
    const int H = p.H, W = p.W;
    auto coords = [&](int tile, int& n, int& y0, int& x0) {
        n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        y0 = (trem / tiles_x) * THW;
        x0 = (trem % tiles_x) * TW;
    };
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < 2 * BUF / 16; i += 512) z4[i] = make_uint4(0, 0, 0, 0);
        // BatchNorm coefficients of this workgroup's 64 input / 64 gradient channels, every image group: read from global memory
        // HERE, once -- a coefficient load inside the tile loop (when the group changes) would be a conditional vector-memory load,
        // and with one of those anywhere in the loop the compiler can no longer count the loads in flight (see the loader below)
        for (int i = threadIdx.x; i < p.G * 2 * 64; i += 512) {
            const int c = i & 63, which = (i >> 6) & 1, g = i >> 7;
            const int c_abs = (which ? nbase : cbase) + c;
            const int si = (which || p.na == 1 || c_abs < p.a[0].C) ? 0 : 1;
            const rd_src_t sd = which ? p.dz : select_src(p.a, si);
            const int cc = c_abs - ((!which && si) ? p.a[0].C : 0);
            const bool live = c_abs < (which ? p.Cout : p.Cin);
            const bool raw = sd.mode == RD_SRC_RAW || !live, bwd = sd.mode == RD_SRC_BNBWD && live;
            const int gs = sd.g_fixed >= 0 ? sd.g_fixed : g;
            float* row = s_coef + (size_t)((g * 2 + which) * 3) * 64 + c;
            if (which && fin_tab && bwd) {
                float P, Q, R;
                rdfin::bwd_pair(fa, gs, cc, P, Q, R);
                row[0] = P;
                row[64] = R;
                row[128] = Q;
                continue;
            }
            row[0] = raw ? 1.f : sd.scale[gs * sd.C + cc];
            row[64] = raw ? 0.f : sd.shift[gs * sd.C + cc];
            row[128] = bwd ? sd.q[gs * sd.C + cc] : 0.f;
        }
    }
    __syncthreads();

    if (role == 1) {
        // =============================================================================== loader waves
        // BRANCH-FREE vector-memory traffic: the hardware counts a wave's outstanding loads in order (vmcnt) and the compiler can
        // only wait for "all but the N youngest" when N is the same on every path that reaches the wait -- with ONE conditional
        // load in the loop (a dead channel slot, the last tiles, a coefficient reload, a run-time experiment switch) every wait
        // becomes vmcnt(0): the set requested a moment ago is waited for together with the one that is needed, and the second
        // register set buys nothing.  So every thread always loads: channel slots beyond Cin / Cout read slot 0 and skip the LDS
        // write, tiles past the end are GHOSTS (pixel (0, 0) of image 0: one cache line).
        const GroupMap gm = make_gm(p.gstart, p.G);
        const int sla = tid % NSA, slz = tid % NSZ;
        const int ca_abs = cbase + sla * S, cz_abs = nbase + slz * S;
        const int sia = (p.na == 1 || ca_abs < p.a[0].C) ? 0 : 1;
        const rd_src_t sda = select_src(p.a, sia);
        const int ca = ca_abs - (sia ? p.a[0].C : 0);
        const bool live_a = ca_abs < p.Cin, live_z = cz_abs < p.Cout;
        PlainSrc<T> psa, psz;
        plain_src_init<T>(psa, sda, live_a ? ca : 0);
        plain_src_init<T>(psz, p.dz, live_z ? cz_abs : 0);
        ItemGeom<NITA> iga;
        ItemGeom<NITZ> igz;
#pragma unroll
        for (int b = 0; b < NITA; ++b) {
            const int pix = tid / NSA + (256 / NSA) * b, py = pix / PW, px = pix - py * PW;
            iga.py[b] = (short)py;
            iga.px[b] = (short)px;
            iga.lds[b] = (pix < NPIX && live_a) ? tr_off<CA>(pix, sla) : A_BYTES + Z_BYTES + tid * 16;     // (relative to s_a)
        }
#pragma unroll
        for (int b = 0; b < NITZ; ++b) {
            const int pix = tid / NSZ + (256 / NSZ) * b;
            igz.py[b] = (short)(pix / TW);
            igz.px[b] = (short)(pix % TW);
            igz.lds[b] = live_z ? tr_off<CZ>(pix, slz) : Z_BYTES + tid * 16;                               // (relative to s_z)
        }
        // TWO register sets: the tiles after the next are in flight while the next one is transformed.  (With one set the
        // request went out right before the barrier and was consumed right behind it: as the loader is the slower role the
        // barrier wait is ~0 and every tile paid a full memory latency -- timing builds, RD_WGWS_EXP: 153 us with, 108 us
        // without the loads on dec.convu3.conv3.)
        uint4 raw_aA[NITA][1], raw_zA[NITZ][NQZ], raw_aB[NITA][1], raw_zB[NITZ][NQZ];
        int offa[NITA], offz[NITZ];                             // (py W + px) C of every item, once (conv_device.h pfu_issue_pre)
        pfu_item_offsets<T, NITA>(offa, psa, iga, W);
        pfu_item_offsets<T, NITZ>(offz, psz, igz, W);
        auto issue = [&](uint4 (&ra)[NITA][1], uint4 (&rz)[NITZ][NQZ], int tile) {
            int n, y0, x0;
            coords(tile, n, y0, x0);
            const bool ghost = tile >= total_tiles;           // a ghost reads one line of image 0 (fill() zeroes its buffer)
            n = ghost ? 0 : n;
            const bool border = x0 + TW + 1 > W;                // the halo tile hangs over the right edge
            pfu_issue_pre<T, NITA>(ra, psa, iga, offa, n, H, W, y0 - 1, x0 - 1, ghost, border);
            pfu_issue_pre<T, NITZ>(rz, psz, igz, offz, n, H, W, y0, x0, ghost, border);
        };
        const bool z_raw = p.dz.mode == RD_SRC_RAW;             // a stored dz / dlogits: copied, not transformed
        const bool a_raw = p.a[0].mode == RD_SRC_RAW && (p.na == 1 || p.a[1].mode == RD_SRC_RAW);
        int g_ctx = -1;
        const int stride = gridDim.x;
        auto fill = [&](uint4 (&ra)[NITA][1], uint4 (&rz)[NITZ][NQZ], int tile, int it) {
            WG_T(1, it, 0);
            int n, y0, x0;
            coords(tile, n, y0, x0);
            const bool ghost = tile >= total_tiles;           // zeros into the buffer nobody reads any more
            n = ghost ? 0 : n;
            y0 = ghost ? -(1 << 20) : y0;
            x0 = ghost ? -(1 << 20) : x0;
            const int g = group_of(gm, n);
            if (g != g_ctx) {                                 // LDS reads (lgkmcnt), not vector-memory loads
                const float* ra_c = s_coef + (size_t)(g * 2 * 3) * 64 + sla * S;
                const float* rz_c = s_coef + (size_t)((g * 2 + 1) * 3) * 64 + slz * S;
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    psa.sc[e] = ra_c[e];
                    psa.sh[e] = ra_c[64 + e];
                    psa.q[e] = ra_c[128 + e];
                    psz.sc[e] = rz_c[e];
                    psz.sh[e] = rz_c[64 + e];
                    psz.q[e] = rz_c[128 + e];
                }
                g_ctx = g;
            }
            char* s_a = smem + (it & 1) * BUF;
            char* s_z = s_a + A_BYTES;
            if constexpr (!(XP & 4)) {
                if (a_raw)             // stored operand (rd_src_t.out of the forward launch): copied, like a stored dz
                    pfu_consume<T, NITA, 1, true, 2>(ra, psa, iga, H, W, y0 - 1, x0 - 1,
                                                     [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_a + l) = u; });
                else
                    pfu_consume<T, NITA, 1, true>(ra, psa, iga, H, W, y0 - 1, x0 - 1,
                                                  [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_a + l) = u; });
                if (z_raw && NQZ == 1) {
#pragma unroll
                    for (int b = 0; b < NITZ; ++b) {
                        const int y = y0 + igz.py[b], x = x0 + igz.px[b];
                        *reinterpret_cast<uint4*>(s_z + igz.lds[b]) = (y < H && x < W) ? rz[b][0] : make_uint4(0, 0, 0, 0);
                    }
                } else {
                    pfu_consume<T, NITZ, NQZ, true>(rz, psz, igz, H, W, y0, x0,
                                              [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_z + l) = u; });
                }
            } else {
                unsigned acc = 0;                               // wait for the set, touch nothing else
#pragma unroll
                for (int b = 0; b < NITA; ++b) acc ^= ra[b][0].x ^ ra[b][0].y ^ ra[b][0].z ^ ra[b][0].w;
#pragma unroll
                for (int b = 0; b < NITZ; ++b) acc ^= rz[b][0].x ^ rz[b][0].y ^ rz[b][0].z ^ rz[b][0].w;
                if (acc == 0x12345678u) *reinterpret_cast<unsigned*>(s_a) = acc;
            }
            WG_T(1, it, 1);
            issue(ra, rz, tile + 2 * stride);                 // a ghost past the end
            WG_T(1, it, 2);
        };
        const int t0 = blockIdx.x;
        issue(raw_aA, raw_zA, t0);
        issue(raw_aB, raw_zB, t0 + stride);
        int it = 0;
        for (int tile = t0; tile < total_tiles; tile += 2 * stride, it += 2) {
            fill(raw_aA, raw_zA, tile, it);
            __syncthreads();               // tile `it` is in its buffer; the MFMA waves are done with the other one
            WG_T(1, it, 3);
            fill(raw_aB, raw_zB, tile + stride, it + 1);
            if (tile + stride < total_tiles) __syncthreads();
            WG_T(1, it + 1, 3);
        }
        return;
    }

    // =================================================================================== MFMA waves
    const int mb = wave >> 1, nb = wave & 1;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // fragment addressing: 16-lane group gq = lane >> 4 -> 16-channel sub-block (gq & 1), K half (gq >> 1)
    const int i16 = lane & 15, gq = lane >> 4;
    const int pix0 = (gq >> 1) * 8 + (i16 >> 2), sub = ((gq & 1) * 16 + (i16 & 3) * 4) * 2;
    const int zoff = tr_frag<CZ>(pix0, mb, sub, 0);
    static_assert(PW % 4 == 2, "halo rows of odd index flip the half swizzle (tr_frag)");
    const int aoff_r[2] = {tr_frag<CA>(pix0, nb, sub, 0), tr_frag<CA>(pix0, nb, sub, 1)};
    int it = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x, ++it) {
        WG_T(0, it, 0);
        __syncthreads();
        WG_T(0, it, 1);
        if constexpr ((XP & 1) != 0) continue;
        const char* s_a = smem + (it & 1) * BUF;
        const char* s_z = s_a + A_BYTES;
        // The LDS pipe, not the matrix pipe, bounds this kernel (SQ counters: LDS active 88 % of the cycles, a third of that bank
        // conflicts), so every fragment is read ONCE per tile: the dz fragments of the four output rows up front (32 registers), then
        // the six halo rows of `a` in turn, each used by the up to three (output row, kernel row) pairs it belongs to --
        // 16 + 36 transpose reads per wave and tile instead of 16 + 72.
        uint4 zf[THW][2];
#pragma unroll
        for (int row = 0; row < THW; ++row)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* zp = s_z + zoff + (row * TW + ks * 16) * PZ;
                const uint2 z0 = lds_tr(zp), z1 = lds_tr(zp + 4 * PZ);
                zf[row][ks] = make_uint4(z0.x, z0.y, z1.x, z1.y);
            }
#pragma unroll
        for (int r = 0; r < THW + 2; ++r) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* ap = s_a + aoff_r[r & 1] + (r * PW + ks * 16) * PA;       // halo coords: input row = output row + kernel row
                const uint2 a0 = lds_tr(ap), a1 = lds_tr(ap + 4 * PA), a2 = lds_tr(ap + 8 * PA);
                const uint4 dq = make_uint4(a0.x, a0.y, a1.x, a1.y);
                const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(a0.y, a0.x, 16), __builtin_amdgcn_alignbit(a1.x, a0.y, 16),
                                            __builtin_amdgcn_alignbit(a1.y, a1.x, 16), __builtin_amdgcn_alignbit(a2.x, a1.y, 16));
                const uint4 m2 = make_uint4(a0.y, a1.x, a1.y, a2.x);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int row = r - kh;
                    if (row < 0 || row >= THW) continue;
                    const bf16x8 afrag = __builtin_bit_cast(bf16x8, zf[row][ks]);
                    acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                    acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                    acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
                }
            }
        }
        WG_T(0, it, 2);
    }
    WG_T(0, 63, 0);
    const int li = lane & 31, h = lane >> 5;
    float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nrow = nbase + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ccol = cbase + nb * 32 + li;
            out[((size_t)tap * CoutPadW + nrow) * CinPadW + ccol] = acc[tap][r];
        }
    WG_T(0, 63, 1);
}

// 16-channel twin of wgrad_tr_kernel (v_mfma_f32_16x16x32_bf16, K = the 32 pixels of a tile row, one 16x16 block,
// the 4 waves split the 8 rows): 32 bytes of channels per pixel, pitch 48 B (disjoint 8-bank spans for 4 pixel rows).
template <int TAPS, int NQZ>
__global__ __launch_bounds__(256, 3) void wgrad_c16_tr_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                               // BatchNorm-backward finalize folded into this launch (bn_fin.h)
    typedef bf16_t T;
    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    constexpr int S = 8;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NPIX = PH * PW;
    constexpr int PA = 48, PZ = 48;                         // bytes per pixel row (16 channels + pad)
    constexpr int KS = 4, ROWS = TH / KS;
    constexpr int NITA = (NPIX * 2 + 255) / 256, NITZ = (TH * TW * 2) / 256;
    constexpr int KW = (TAPS == 9) ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_a = smem;                                       // [NPIX + 4][PA]
    char* s_z = smem + (NPIX + 4) * PA;                     // [TH*TW][PZ]

    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);

    f32x4v acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    auto coords = [&](int tile, int& n, int& y0, int& x0) {
        n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        y0 = (trem / tiles_x) * TH;
        x0 = (trem % tiles_x) * TW;
    };
    const int sl = tid & 1;                                 // this thread's channel slot (of 2) in both tiles
    // grid (splits, Cout blocks of 16, Cin blocks of 16): this workgroup's 16 x 16 block of every tap
    const int ca_abs = blockIdx.z * 16 + sl * S, cz_abs = blockIdx.y * 16 + sl * S;
    const int sia = (p.na == 1 || ca_abs < p.a[0].C) ? 0 : 1;
    const rd_src_t sda = select_src(p.a, sia);
    const int ca = ca_abs - (sia ? p.a[0].C : 0);
    const bool live_a = ca_abs < p.Cin, live_z = cz_abs < p.Cout;
    PlainSrc<T> psa, psz;
    plain_src_init<T>(psa, sda, live_a ? ca : 0);
    plain_src_init<T>(psz, p.dz, live_z ? cz_abs : 0);
    ItemGeom<NITA> iga;
    ItemGeom<NITZ> igz;
#pragma unroll
    for (int b = 0; b < NITA; ++b) {
        const int pix = (tid >> 1) + 128 * b, py = pix / PW, px = pix - py * PW;
        iga.py[b] = (short)py;
        iga.px[b] = (short)px;
        iga.lds[b] = pix < NPIX ? pix * PA + sl * 16 : -1;
    }
#pragma unroll
    for (int b = 0; b < NITZ; ++b) {
        const int pix = (tid >> 1) + 128 * b;
        igz.py[b] = (short)(pix / TW);
        igz.px[b] = (short)(pix % TW);
        igz.lds[b] = pix * PZ + sl * 16;
    }
    uint4 raw_a[NITA][1], raw_z[NITZ][NQZ];
    auto issue = [&](int tile) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        if (live_a) pfu_issue<T, NITA>(raw_a, psa, iga, n, H, W, y0 - HALO, x0 - HALO);
        if (live_z) pfu_issue<T, NITZ>(raw_z, psz, igz, n, H, W, y0, x0);
    };
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < ((NPIX + 4) * PA + TH * TW * PZ) / 16; i += 256) z4[i] = make_uint4(0, 0, 0, 0);
    }
    int g_ctx = -1;
    if ((int)blockIdx.x < total_tiles) issue(blockIdx.x);
    // fragment addressing: 16-lane group kg = K block (8 pixels), lane li = channel
    const int zoff = (kg * 8 + (li >> 2)) * PZ + (li & 3) * 8;
    const int aoff = (kg * 8 + (li >> 2)) * PA + (li & 3) * 8;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const int g = group_of(gm, n);
        if (g != g_ctx) {
            if (live_a) plain_src_coef<T>(psa, sda, g, ca);
            if (live_z) plain_src_coef<T>(psz, p.dz, g, cz_abs);
            g_ctx = g;
        }
        __syncthreads();
        if (live_a)
            pfu_consume<T, NITA, 1>(raw_a, psa, iga, H, W, y0 - HALO, x0 - HALO,
                                    [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_a + l) = u; });
        if (live_z)
            pfu_consume<T, NITZ, NQZ>(raw_z, psz, igz, H, W, y0, x0,
                                      [&](int l, const uint4& u) { *reinterpret_cast<uint4*>(s_z + l) = u; });
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) issue(tile + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = kq + rr * KS;
            const char* zp = s_z + zoff + (row * TW) * PZ;
            const uint2 z0 = lds_tr(zp), z1 = lds_tr(zp + 4 * PZ);
            const bf16x8 afrag = __builtin_bit_cast(bf16x8, make_uint4(z0.x, z0.y, z1.x, z1.y));
#pragma unroll
            for (int kh = 0; kh < KW; ++kh) {
                const char* ap = s_a + aoff + ((row + kh) * PW) * PA;
                const uint2 a0 = lds_tr(ap), a1 = lds_tr(ap + 4 * PA);
                const uint4 dq = make_uint4(a0.x, a0.y, a1.x, a1.y);
                if constexpr (TAPS == 9) {
                    const uint2 a2 = lds_tr(ap + 8 * PA);
                    const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(a0.y, a0.x, 16), __builtin_amdgcn_alignbit(a1.x, a0.y, 16),
                                                __builtin_amdgcn_alignbit(a1.y, a1.x, 16), __builtin_amdgcn_alignbit(a2.x, a1.y, 16));
                    const uint4 m2 = make_uint4(a0.y, a1.x, a1.y, a2.x);
                    acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                    acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                    acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[0], 0, 0, 0);
                }
            }
        }
    }
    {
        float* s_acc = reinterpret_cast<float*>(smem);     // [(KS-1)][TAPS][4][64]
        __syncthreads();
        if (kq > 0) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[(((kq - 1) * TAPS + tap) * 4 + r) * 64 + lane] = acc[tap][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int k2 = 0; k2 < KS - 1; ++k2)
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tap][r] += s_acc[((k2 * TAPS + tap) * 4 + r) * 64 + lane];
        }
    }
    if (kq == 0) {
        // D layout of the 16x16 MFMA: column (N, cin) = lane&15, row (M, cout) = 4*(lane>>4) + r
        float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((size_t)tap * CoutPadW + blockIdx.y * 16 + 4 * kg + r) * CinPadW + blockIdx.z * 16 + li] = acc[tap][r];
    }
}

// Split reduction.  (A float4-coalesced, chunked variant with atomics into a zeroed dW measured 320 us/step SLOWER over
// the 40 launches -- the extra memset launch and fewer blocks in flight cost more than the strided reads, which are L2 hits
// on partials written microseconds earlier: profiles/README.md round 2.)
// block = 32 outputs x 8 split lanes: each thread sums every 8th split, LDS folds the 8 lanes in a fixed order
template <int OB>                                         // outputs per block; 256/OB threads share one output's splits
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* dW, int nsplit, int taps, int Cout,
                                                           int Cin, int CoutPadW, int CinPadW, float beta) {
    constexpr int QL = 256 / OB;
    __shared__ float s[QL][OB];
    const int total = taps * Cout * Cin;
    const int o = threadIdx.x % OB, ql = threadIdx.x / OB;
    const size_t stride = (size_t)taps * CoutPadW * CinPadW;
    for (int base = blockIdx.x * OB; base < total; base += gridDim.x * OB) {
        const int i = base + o;
        float acc = 0.f;
        int c = 0, n = 0, tap = 0;
        if (i < total) {
            c = i % Cin; n = (i / Cin) % Cout; tap = i / (Cin * Cout);
            const float* src = partial + ((size_t)tap * CoutPadW + n) * CinPadW + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four loads in flight per thread
            int k = ql;
            for (; k + 3 * QL < nsplit; k += 4 * QL) {
                a0 += src[(size_t)k * stride];
                a1 += src[(size_t)(k + QL) * stride];
                a2 += src[(size_t)(k + 2 * QL) * stride];
                a3 += src[(size_t)(k + 3 * QL) * stride];
            }
            for (; k < nsplit; k += QL) a0 += src[(size_t)k * stride];
            acc = (a0 + a1) + (a2 + a3);
        }
        s[ql][o] = acc;
        __syncthreads();
        if (ql == 0 && i < total) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < QL; ++q) v += s[q][o];
            float* d = dW + ((size_t)n * Cin + c) * taps + tap;
            *d = (beta != 0.f ? beta * *d : 0.f) + v;
        }
        __syncthreads();
    }
}

// every source a plain per-pixel read (no pooling / on-the-fly upsampling) of whole 16-byte channel slots
bool wgrad_pf_ok(const rd_wgrad_t& p) {
    for (int i = 0; i < p.na; ++i) {
        const int m = p.a[i].mode;
        if (!(m == RD_SRC_RAW || m == RD_SRC_AFF || m == RD_SRC_AFFACT) || p.a[i].C % 8) return false;
    }
    return (p.dz.mode == RD_SRC_RAW || p.dz.mode == RD_SRC_BNBWD) && p.dz.C % 8 == 0;
}

// every source made of whole 16-byte channel slots (any read mode)
bool wgrad_slots_ok(const rd_wgrad_t& p) {
    for (int i = 0; i < p.na; ++i)
        if (p.a[i].C % 8) return false;
    return p.dz.C % 8 == 0;
}

struct WgradGeom {
    int MB, NB, KS, CoutPadW, CinPadW, gx, total_tiles, nsplit;
    bool c16;
    bool sym;              // wgrad_sym_kernel: 128 gradient channels x 64 input channels per workgroup
};

template <typename T>
WgradGeom wgrad_geom(const rd_wgrad_t& p) {
    WgradGeom g;
    g.c16 = false;
    g.sym = false;
    // 16 x 16 blocks (v_mfma_f32_16x16x32_bf16, 3 workgroups/CU): the 16-channel layers, and -- a grid of such blocks,
    // transpose-read kernel only -- the layers with <= 16 output channels on 32 input channels (dec.out1 32->2,
    // dec.convu1.conv2 32->16, rec.convu1.conv1 32->16), which would leave half of every 32 x 32 block's rows empty
    if (sizeof(T) == 2 && p.Cout <= 16 && p.Cin <= (rd_switch("RD_WG_C16_GRID", 1) ? 32 : 16) && wgrad_pf_ok(p) && rd_switch("RD_WG_TR_OFF", 0) == 0) {
        g.c16 = true;
        g.MB = g.NB = 1;                                   // blocks per WORKGROUP
        g.KS = 4;
        g.CoutPadW = 16;
        g.CinPadW = (p.Cin + 15) / 16 * 16;
        g.total_tiles = p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
        // exactly the resident set (no second, partially filled round) at 3 workgroups/CU
        // (cu_limit does not apply: these layers are the 400x400 / 200x200 ones, HBM-bound, and want every CU's load path)
        int gx = 3 * rd_num_cus() / (g.CinPadW / 16);
        if (gx > g.total_tiles) gx = g.total_tiles;
        g.gx = gx < 1 ? 1 : gx;
        g.nsplit = g.gx;
        return g;
    }
    const int cout32 = (p.Cout + 31) / 32, cin32 = (p.Cin + 31) / 32;
    // fp32 keeps 32x32 blocks (LDS budget); bf16 uses 64-wide tiles where the layer has them
    g.MB = (sizeof(T) == 2 && cout32 % 2 == 0 && rd_switch("RD_WG_MB_MAX", 2) >= 2) ? 2 : 1;
    g.NB = (sizeof(T) == 2 && cin32 % 2 == 0 && rd_switch("RD_WG_NB_MAX", 2) >= 2) ? 2 : 1;
    g.KS = 4 / (g.MB * g.NB);
    g.CoutPadW = cout32 * 32;
    g.CinPadW = cin32 * 32;
    g.total_tiles = p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    int pairs = (g.CoutPadW / (g.MB * 32)) * (g.CinPadW / (g.NB * 32));
    // 3x3 layers with whole 128-channel blocks of gradient channels on plain sources: the symmetric eight-wave kernel, half as many
    // workgroups per pixel split (each owns 128 x 64 channels), so twice the splits on the same compute-unit budget
    if (sizeof(T) == 2 && p.taps == 9 && g.MB == 2 && g.NB == 2 && g.CoutPadW % 128 == 0 && wgrad_pf_ok(p) && p.G * (2 * 64 + 3 * 128) * 4 <= 16384 && (p.na == 1 || p.a[0].C % 64 == 0) &&
        rd_switch("RD_WG_WS", 1) && rd_switch("RD_WG_SYM", RD_WG_SYM_DEFAULT) && rd_switch("RD_WG_TR_OFF", 0) == 0) {
        g.sym = true;
        pairs /= 2;
    }
    // these kernels hold 144 accumulator registers per lane -> one workgroup per CU is resident: launching more
    // workgroups than CUs only multiplies the partial-sum traffic (147 KB per workgroup for a 64x64 tile)
    // cu_limit: a weight gradient launched on a side stream beside the dgrad chain takes only part of the GPU (its persistent
    // workgroups cannot share a CU with the chain's kernels, so a full-width launch makes the chain wait: tuning.py side_cus)
    const int slots = p.cu_limit > 0 ? p.cu_limit : rd_switch("RD_WG_SLOTS", rd_num_cus());
    int gx = (slots + pairs - 1) / pairs;
    if (gx > g.total_tiles) gx = g.total_tiles;
    if (gx < 1) gx = 1;
    g.gx = gx;
    g.nsplit = gx;
    return g;
}

template <typename T, int TAPS, int MB, int NB>
int launch_wgrad(const rd_wgrad_t& p, const WgradGeom& g, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    size_t lds = (size_t)(PH * PW * NB * 32 + TH * TW * MB * 32) * sizeof(T);
    const size_t lds_red = (size_t)(4 / (MB * NB) - 1) * (MB * NB) * 16 * 64 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    dim3 grid(g.gx, g.CoutPadW / (MB * 32), g.CinPadW / (NB * 32));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, TAPS, MB, NB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    rd_launch((wgrad_kernel<T, TAPS, MB, NB>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    return (int)hipGetLastError();
}

template <int TAPS>
int launch_wgrad_c16(const rd_wgrad_t& p, const WgradGeom& g, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    dim3 grid(g.gx, g.CoutPadW / 16, g.CinPadW / 16);
    size_t lds = (size_t)((TH + 2 * HALO) * (TW + 2 * HALO) + 4) * 48 + (size_t)TH * TW * 48;
    const size_t lds_red = (size_t)3 * TAPS * 4 * 64 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    if (p.dz.mode == RD_SRC_BNBWD)
        rd_launch((wgrad_c16_tr_kernel<TAPS, 2>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    else
        rd_launch((wgrad_c16_tr_kernel<TAPS, 1>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    return (int)hipGetLastError();
}

template <int TAPS, int MB, int NB>
int launch_wgrad_tr(const rd_wgrad_t& p, const WgradGeom& g, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    size_t lds = (size_t)(PH * PW + 4) * tr_pitch(NB * 32) + (size_t)TH * TW * tr_pitch(MB * 32);
    const size_t lds_red = (size_t)(4 / (MB * NB) - 1) * (MB * NB) * 16 * 64 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    dim3 grid(g.gx, g.CoutPadW / (MB * 32), g.CinPadW / (NB * 32));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tr_kernel<TAPS, MB, NB, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tr_kernel<TAPS, MB, NB, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tr_kernel<TAPS, MB, NB, 1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if constexpr (TAPS == 9 && MB == 2 && NB == 2) {
        if (g.sym) return rd_wgrad_sym_launch(p, g.gx, g.CoutPadW, g.CinPadW, st);
        // 64 x 64 blocks, plain sources: the warp-specialised kernel (4-row LDS tiles: twice the tile count)
        static const int ws = rd_switch("RD_WG_WS", 1);
        if (ws && wgrad_pf_ok(p)) {
            const int ws_lds = 2 * ((6 * PW + 4) * tr_pitch(64) + 4 * TW * tr_pitch(64) + 256 * 16) + p.G * 2 * 3 * 64 * (int)sizeof(float);
            const int tiles_ws = p.N * ((p.H + 3) / 4) * ((p.W + TW - 1) / TW);
#define RD_WGWS_LAUNCH(NQZ, XP) do { \
                static int attr_lds = 0; \
                if (attr_lds < ws_lds) { \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_kernel<NQZ, XP>), hipFuncAttributeMaxDynamicSharedMemorySize, ws_lds); \
                    attr_lds = ws_lds; \
                } \
                rd_launch((wgrad_ws_kernel<NQZ, XP>), grid, dim3(512), ws_lds, st, p, g.CoutPadW, g.CinPadW, tiles_ws, rdfin::current()); \
                return (int)hipGetLastError(); \
            } while (0)
#ifdef RD_DEBUG_SWITCHES
            if (p.dz.mode == RD_SRC_BNBWD && rd_switch("RD_WGWS_EXP", 0) == 8) RD_WGWS_LAUNCH(2, 8);
            if (p.dz.mode == RD_SRC_BNBWD && rd_switch("RD_WGWS_EXP", 0) == 9) RD_WGWS_LAUNCH(2, 1);
            if (p.dz.mode == RD_SRC_BNBWD && rd_switch("RD_WGWS_EXP", 0) == 4) RD_WGWS_LAUNCH(1, 4);     // one operand, no transform: "both operands stored"

            if (p.dz.mode != RD_SRC_BNBWD) {
                switch (rd_switch("RD_WGWS_EXP", 0)) {
                case 1: RD_WGWS_LAUNCH(1, 1);
                case 4: RD_WGWS_LAUNCH(1, 4);
                case 5: RD_WGWS_LAUNCH(1, 5);
                default: break;
                }
            }
#endif
            if (p.dz.mode == RD_SRC_BNBWD) RD_WGWS_LAUNCH(2, 0);
            RD_WGWS_LAUNCH(1, 0);
#undef RD_WGWS_LAUNCH
        }
    }
    if (!wgrad_pf_ok(p))
        rd_launch((wgrad_tr_kernel<TAPS, MB, NB, 1, false>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    else if (p.dz.mode == RD_SRC_BNBWD)
        rd_launch((wgrad_tr_kernel<TAPS, MB, NB, 2, true>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    else
        rd_launch((wgrad_tr_kernel<TAPS, MB, NB, 1, true>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles, rdfin::current());
    return (int)hipGetLastError();
}

template <typename T>
int dispatch_wgrad(const rd_wgrad_t& p, hipStream_t st) {
    const WgradGeom g = wgrad_geom<T>(p);
    int e;
    if (g.c16) {
        e = p.taps == 9 ? launch_wgrad_c16<9>(p, g, st) : launch_wgrad_c16<1>(p, g, st);
    } else if (sizeof(T) == 2 && wgrad_slots_ok(p) && rd_switch("RD_WG_TR_OFF", 0) == 0) {
        // bf16, every source made of whole 16-byte channel slots: [pixel][channel] LDS tiles + ds_read_b64_tr_b16
#define RD_WGT(TAPS_)                                                                \
    if (g.MB == 2 && g.NB == 2) e = launch_wgrad_tr<TAPS_, 2, 2>(p, g, st);          \
    else if (g.MB == 2) e = launch_wgrad_tr<TAPS_, 2, 1>(p, g, st);                  \
    else if (g.NB == 2) e = launch_wgrad_tr<TAPS_, 1, 2>(p, g, st);                  \
    else e = launch_wgrad_tr<TAPS_, 1, 1>(p, g, st);
        if (p.taps == 9) { RD_WGT(9) } else { RD_WGT(1) }
#undef RD_WGT
    } else {
        // fp32, and bf16 with ragged channel counts (e.g. a 3-channel image that is not padded to a slot)
#define RD_WG(TAPS_)                                                                 \
    if (g.MB == 2 && g.NB == 2) e = launch_wgrad<T, TAPS_, 2, 2>(p, g, st);          \
    else if (g.MB == 2) e = launch_wgrad<T, TAPS_, 2, 1>(p, g, st);                  \
    else if (g.NB == 2) e = launch_wgrad<T, TAPS_, 1, 2>(p, g, st);                  \
    else e = launch_wgrad<T, TAPS_, 1, 1>(p, g, st);
        if (p.taps == 9) { RD_WG(9) } else { RD_WG(1) }
#undef RD_WG
    }
    if (e) return e;
    if (rd_switch("RD_WG_NO_REDUCE", 0)) return 0;          // debug build only: timing experiment (what the split reductions cost the step)
    return rd_wgrad_reduce_launch(p.partial, p.dW, g.nsplit, p.taps, p.Cout, p.Cin, g.CoutPadW, g.CinPadW, p.beta, st);
}

}  // namespace

int rd_wgrad_reduce_launch(const float* partial, float* dW, int nsplit, int taps, int Cout, int Cin, int CoutPadW, int CinPadW, float beta,
                           hipStream_t st) {
    const int total = taps * Cout * Cin;
    // many splits of a small filter (the 16/32-channel layers): 8 outputs x 32 split lanes per block
    if (nsplit >= 128 && total <= 16384) {
        rd_launch(wgrad_reduce_kernel<8>, dim3((total + 7) / 8), dim3(256), 0, st, partial, dW, nsplit, taps, Cout, Cin, CoutPadW, CinPadW, beta);
    } else {
        int blocks = (total + 31) / 32;
        if (blocks > 8192) blocks = 8192;
        rd_launch(wgrad_reduce_kernel<32>, dim3(blocks), dim3(256), 0, st, partial, dW, nsplit, taps, Cout, Cin, CoutPadW, CinPadW, beta);
    }
    return (int)hipGetLastError();
}

int rd_wgrad_dispatch(const rd_wgrad_t& p, int dtype, hipStream_t st) {
    return dtype == RD_BF16 ? dispatch_wgrad<bf16_t>(p, st) : dispatch_wgrad<float>(p, st);
}

int64_t rd_wgrad_ws_bytes(const rd_wgrad_t& p, int dtype) {
    const WgradGeom g = dtype == RD_BF16 ? wgrad_geom<bf16_t>(p) : wgrad_geom<float>(p);
    return (int64_t)g.nsplit * p.taps * g.CoutPadW * g.CinPadW * (int64_t)sizeof(float);
}


#ifdef RD_DEBUG_SWITCHES
extern "C" int rd_debug_wg_trace(int key, unsigned long long* out) {       // debug library only: arm (out == null) or read 2 x 64 x 4 stamps
    if (!out) return (int)hipMemcpyToSymbol(HIP_SYMBOL(wg_trace_key), &key, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wg_trace), sizeof(unsigned long long) * 2 * 64 * 4);
}
#endif
