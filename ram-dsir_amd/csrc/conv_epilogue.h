// conv_epilogue.h -- the output side shared by the multi-chunk conv kernels (conv_big.hip, conv_pp.hip): accumulators ->
// LDS staging -> 16-byte NHWC stores (forward: + bias, BatchNorm batch statistics) or the gradient destinations
// (backward: activation mask, pool scatter, skip accumulation, BN-backward sums).
#pragma once
#include "conv_device.h"

namespace {

// ------------------------------------------------------------------------------------ epilogue (shared)
template <typename T, int NB>
__device__ __forceinline__ void conv_epilogue(const rd_conv_t& p, f32x16 (&acc)[2][NB], char* smem, int tid, int n, int g, int y0, int x0,
                                              int n0, int slot) {
    constexpr int S = Slot<T>::N;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    // ---------------------------------------------------------------- epilogue
    // C/D layout of the 32x32 MFMA: column (N, channel) = lane&31, row (M, pixel) = (r&3)+8*(r>>2)+4*(lane>>5).
    // Each 32-channel block is staged through LDS as fp32 [256 pixels][32 ch] so that the global side runs
    // on 16-byte slots (coalesced stores; vector reads of z / old gradients in the backward epilogues).
    constexpr int SL = 32 / S;                             // slots per 32 channels
    float* s_out = reinterpret_cast<float*>(smem);         // [TH*TW][32]
    double* s_red = reinterpret_cast<double*>(s_out + TH * TW * 32);   // [32][2]
    T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __syncthreads();
        const int cb = n0 + nb * 32;
        if (tid < 64) s_red[tid] = 0.0;
        {
            const int cch = cb + li;
            const bool cok = cch < p.Cout;
            const float bsv = (p.emode == 0 && cok && p.bias) ? p.bias[cch] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int y = y0 + wave * 2 + mb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = acc[mb][nb][r] + bsv;
                    s_out[((wave * 2 + mb) * TW + col) * 32 + li] = v;
                    // BatchNorm sums are those of the result WITHOUT the bias (ramdsir.h, RD_STAT_SLOTS): finalize adds it back
                    if (cok && y < H && x0 + col < W) { const float a0 = acc[mb][nb][r]; s1 += a0; s2 += a0 * a0; }
                }
            }
            __syncthreads();
            if (p.emode == 0 && p.stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0 && cok) {
                    atomicAdd(&s_red[li * 2 + 0], (double)s1);
                    atomicAdd(&s_red[li * 2 + 1], (double)s2);
                }
            }
        }
        const int sl = tid % SL;                           // constant per thread: 256 % SL == 0
        const int c = cb + sl * S;
        float b1[S], b2[S];
#pragma unroll
        for (int e = 0; e < S; ++e) b1[e] = b2[e] = 0.f;
        const int di = (p.emode == 1 && c >= p.c_split) ? 1 : 0;
        const rd_dst_t d = select_dst(p, di);
        const int cd = c - (di ? p.c_split : 0);
        float dsc[S], dsh[S];
        {
            const int gd = d.g_fixed >= 0 ? d.g_fixed : g;
#pragma unroll
            for (int e = 0; e < S; ++e) {
                const bool ok = p.emode == 1 && c < p.Cout && d.kind != RD_DST_NONE && d.scale && (cd + e < d.Cd);
                dsc[e] = ok ? d.scale[gd * d.Cd + cd + e] : 1.f;
                dsh[e] = ok ? d.shift[gd * d.Cd + cd + e] : 0.f;
            }
        }
        if (c < p.Cout) {
            for (int idx = tid; idx < TH * TW * SL; idx += 256) {
                const int pix = idx / SL;
                const int y = y0 + pix / TW, x = x0 + pix % TW;
                if (y >= H || x >= W) continue;
                float v[S];
#pragma unroll
                for (int e = 0; e < S; e += 4) {
                    const float4 f = *reinterpret_cast<const float4*>(s_out + pix * 32 + sl * S + e);
                    v[e] = f.x; v[e + 1] = f.y; v[e + 2] = f.z; v[e + 3] = f.w;
                }
                if (p.emode == 0)
                    store_vec<T>(out + ((size_t)(n * H + y) * W + x) * p.Cout + c, v, p.Cout - c, (p.Cout % S) == 0);
                else if (d.kind != RD_DST_NONE)
                    grad_item<T>(d, g, n, y, x, H, W, cd, v, dsc, dsh, b1, b2);
            }
        }
        // (wave-uniform condition: emode is a launch constant; lanes without a live destination add zeros)
        if (p.emode == 1) flush_bstats<S, SL>(s_red, lane, sl, b1, b2);
        __syncthreads();
        if (tid < 32 && cb + tid < p.Cout) {
            if (p.emode == 0) {
                if (p.stats) {
                    const size_t so = (((size_t)g * RD_STAT_SLOTS + slot) * p.Cout + cb + tid) * 2;
                    atomicAdd(&p.stats[so + 0], s_red[tid * 2 + 0]);
                    atomicAdd(&p.stats[so + 1], s_red[tid * 2 + 1]);
                }
            } else {
                const int cch = cb + tid;
                const int dj = cch >= p.c_split ? 1 : 0;
                const rd_dst_t dd = select_dst(p, dj);
                if (dd.kind != RD_DST_NONE && dd.bstats) {
                    const int cdd = cch - (dj ? p.c_split : 0);
                    const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
                    const size_t so = (((size_t)gd * RD_STAT_SLOTS + slot) * dd.Cd + cdd) * 2;
                    atomicAdd(&dd.bstats[so + 0], s_red[tid * 2 + 0]);
                    atomicAdd(&dd.bstats[so + 1], s_red[tid * 2 + 1]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ register epilogue (bf16)
// For accumulators produced with the MFMA roles swapped (A = weights, B = pixels): row (r&3) + 8*(r>>2) + 4*h of block nb
// is an output channel, the column (lane & 31) a pixel of tile row wave*2 + mb.  rd_half_swap regroups a lane's 16
// channels of a block into two vectors of 8 contiguous channels (lanes h=0: 16v..16v+7, lanes h=1: 16v+8..16v+15), which
// are stored / combined with the destination tensors as whole 16-byte NHWC slots.  ~1/3 of the instructions of the
// LDS-staged conv_epilogue above.  EP 1 = forward (+bias, BatchNorm sums), EP 2 = gradient into plain destinations.
// s_epi: [NT] bias (EP 1) or [2][NT] producer scale / shift of the destination channels (EP 2), staged by the caller.
#ifndef PF_T
#define PF_T(ev) do { } while (0)
#endif
template <typename T, int NB, int EP, int TS = 0>
__device__ __forceinline__ void conv_epilogue_lean(const rd_conv_t& p, f32x16 (&acc)[2][NB], double* s_red, const float* s_epi,
                                                   int tid, int n, int g, int y0, int x0, int n0, int slot) {
    static_assert(sizeof(T) == 2 && (EP == 1 || EP == 2), "bf16, forward or plain-gradient");
    constexpr int S = 8, NT = NB * 32;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    if (tid < NT * 2) s_red[tid] = 0.0;
    __syncthreads();
    PF_T(7);
    float sa[NB][2][S], sb[NB][2][S];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < S; ++e) sa[nb][v][e] = sb[nb][v][e] = 0.f;
    // gradient launches: EVERY z / g vector of this lane is requested before the first one is used.  Written as one loop (load,
    // use, store, next vector) the compiler has to assume that a store may alias the next load and waits for vmcnt(0) eight times --
    // the stores count too -- i.e. sixteen dependent memory round trips per tile, most of a workgroup's life (dec.convu2.conv3
    // dgrad: 29 us per workgroup against 15 us for the forward launch of the same shape).  Lanes without a live destination read
    // the first weight vector instead (always mapped), so that no load is conditional.
    uint4 zq[2][NB][2], gq[2][NB][2];
    if constexpr (EP == 2) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            int pr, pc;
            bool live;
            tile_pixel<TS>(wave * 2 + mb, li, pr, pc, live);
            const int y = y0 + pr, x = x0 + pc;
            const bool valid = live && y < H && x < W;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int c = n0 + nb * 32 + 16 * v;
                    const int di = c >= p.c_split ? 1 : 0;
                    const rd_dst_t d = select_dst(p, di);
                    const bool ok = d.kind != RD_DST_NONE && valid;
                    const size_t idx = ((size_t)((n + d.n_off) * H + y) * W + x) * d.Cd + (c - (di ? p.c_split : 0)) + 8 * h;
                    const T* dummy = reinterpret_cast<const T*>(p.w);
                    zq[mb][nb][v] = ld16((ok && d.z) ? reinterpret_cast<const T*>(d.z) + idx : dummy);
                    gq[mb][nb][v] = ld16((ok && d.accumulate) ? reinterpret_cast<const T*>(d.g) + idx : dummy);
                }
        }
    }
    PF_T(8);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int pr, pc;
        bool live;
        tile_pixel<TS>(wave * 2 + mb, li, pr, pc, live);
        const int y = y0 + pr, x = x0 + pc;
        const bool valid = live && y < H && x < W;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int cl = nb * 32 + 16 * v + 8 * h;             // first channel of this lane's vector within the tile
                float o[S];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = __float_as_uint(acc[mb][nb][8 * v + j]);
                    const unsigned b = __float_as_uint(acc[mb][nb][8 * v + 4 + j]);
                    const HalfSwap r = rd_half_swap(a, b, h);
                    o[j] = __uint_as_float(r.r0);
                    o[4 + j] = __uint_as_float(r.r1);
                }
                if constexpr (EP == 1) {
                    if (!valid) continue;
                    const float4 b0 = *reinterpret_cast<const float4*>(s_epi + cl), b1 = *reinterpret_cast<const float4*>(s_epi + cl + 4);
                    const float bias[S] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        sa[nb][v][e] += o[e];                     // sums exclude the bias (ramdsir.h, RD_STAT_SLOTS)
                        sb[nb][v][e] += o[e] * o[e];
                        o[e] += bias[e];
                    }
                    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + ((size_t)(n * H + y) * W + x) * p.Cout + n0 + cl) = Slot<T>::pack(o);
                } else {
                    const int c = n0 + nb * 32 + 16 * v;             // c_split % 16 == 0: one destination per vector pair
                    const int di = c >= p.c_split ? 1 : 0;
                    const rd_dst_t d = select_dst(p, di);
                    if (d.kind == RD_DST_NONE || !valid) continue;
                    const size_t idx = ((size_t)((n + d.n_off) * H + y) * W + x) * d.Cd + (c - (di ? p.c_split : 0)) + 8 * h;
                    const uint4 zero4 = make_uint4(0, 0, 0, 0);
                    const uint4 zu = d.z ? zq[mb][nb][v] : zero4, gu = d.accumulate ? gq[mb][nb][v] : zero4;
                    float z[S], gw[S];
                    Slot<T>::unpack(zu, z);
                    Slot<T>::unpack(gu, gw);
                    // producer scale / shift of the vector's 8 channels as four 16-byte LDS reads; the activation gradient as ONE select
                    // per element: `lo` is the factor where bn(z) <= 0 -- the slope, or 1 for a destination without a mask (a
                    // wave-uniform choice made per vector, not a branch per element)
                    const float4 c0 = *reinterpret_cast<const float4*>(s_epi + cl), c1 = *reinterpret_cast<const float4*>(s_epi + cl + 4);
                    const float4 h0 = *reinterpret_cast<const float4*>(s_epi + NT + cl), h1 = *reinterpret_cast<const float4*>(s_epi + NT + cl + 4);
                    const float sc[S] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    const float sh[S] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    const float lo = (d.act && d.z) ? d.slope : 1.f;
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        const float m = (z[e] * sc[e] + sh[e]) > 0.f ? 1.f : lo;
                        const float gn = o[e] * m;
                        sa[nb][v][e] += gn;
                        sb[nb][v][e] += gn * z[e];
                        gw[e] += gn;
                    }
                    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(d.g) + idx) = Slot<T>::pack(gw);
                    if (mb == 0 && nb == 0 && v == 0) PF_T(11);
                }
            }
    }
    PF_T(9);
    // per-channel sums over the 32 pixel lanes of each half-wave (half_wave_sums), then NB LDS atomics per lane
    {
        float r[NB * 32];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    r[((nb * 2 + v) * S + e) * 2 + 0] = sa[nb][v][e];
                    r[((nb * 2 + v) * S + e) * 2 + 1] = sb[nb][v][e];
                }
        half_wave_sums<NB * 32>(r, li);
        const int base = half_wave_sum_index(li) * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = base + j, c = idx >> 1;                       // c = (nb * 2 + v) * 8 + e
            const int cl = (c >> 4) * 32 + ((c >> 3) & 1) * 16 + 8 * h + (c & 7);
            atomicAdd(&s_red[cl * 2 + (idx & 1)], (double)r[j]);
        }
    }
    PF_T(10);
    __syncthreads();
    if (tid < NT) {
        const int cch = n0 + tid;
        if constexpr (EP == 1) {
            if (p.stats) {
                const size_t so = (((size_t)g * RD_STAT_SLOTS + slot) * p.Cout + cch) * 2;
                atomicAdd(&p.stats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&p.stats[so + 1], s_red[tid * 2 + 1]);
            }
        } else {
            const int dj = cch >= p.c_split ? 1 : 0;
            const rd_dst_t dd = select_dst(p, dj);
            if (dd.kind != RD_DST_NONE && dd.bstats) {
                const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
                const size_t so = (((size_t)gd * RD_STAT_SLOTS + slot) * dd.Cd + cch - (dj ? p.c_split : 0)) * 2;
                atomicAdd(&dd.bstats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&dd.bstats[so + 1], s_red[tid * 2 + 1]);
            }
        }
    }
}

}  // namespace
