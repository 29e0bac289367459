// conv_pf.h -- the chunk-pipelined multi-chunk conv kernel (conv_pf_kernel) and its MFMA phase, shared by conv_big.hip
// (LDS-staged epilogue) and conv_lean.hip (register epilogues), so that the two sets of instantiations build in parallel.
#pragma once
#include "conv_device.h"
#include "conv_epilogue.h"

namespace {

// MFMA phase of one input-channel chunk: s_in [halo pixel][4 slots], s_w [tap][n][4 slots], both XOR-swizzled
// SWAP: A = weights, B = pixels (accumulator rows = output channels: the register epilogues)
template <typename T, int TAPS, int NB, bool SWAP = false, int TS = 0>
__device__ __forceinline__ void conv_mma_chunk(const uint4* s_in, const uint4* s_w, int wave, int li, int h, f32x16 (&acc)[2][NB]) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PW = TileGeo<TS>::LdsPitch(HALO);        // LDS row pitch of the halo tile (conv_device.h)
    constexpr int NT = NB * 32;
    int pr[2], pc[2];                                      // this lane's tile pixel of its two M-blocks
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        bool live;
        tile_pixel<TS>(wave * 2 + mb, li, pr[mb], pc[mb], live);
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int pix = (pr[mb] + kh) * PW + pc[mb] + kw;
            const uint4* a_rec = s_in + pix * 4;
            const int a_sw = (pix >> 2) & 3;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int nn = nb * 32 + li;
                if constexpr (SWAP) Mma<T>::chunk(s_w + (tap * NT + nn) * 4, (nn >> 2) & 3, a_rec, a_sw, h, acc[mb][nb]);
                else Mma<T>::chunk(a_rec, a_sw, s_w + (tap * NT + nn) * 4, (nn >> 2) & 3, h, acc[mb][nb]);
            }
        }
    }
}

// 0: a source needs the generic loader (pooled / on-the-fly upsampled / ragged channel tail);
// 1: plain single-operand sources (forward); 2: BN-backward or raw gradient source (dgrad)
int conv_pf_kind(const rd_conv_t& p) {
    constexpr int S = 8;
    bool two = false;
    for (int i = 0; i < p.nsrc; ++i) {
        const int m = p.src[i].mode;
        if (!(m == RD_SRC_RAW || m == RD_SRC_AFF || m == RD_SRC_AFFACT || m == RD_SRC_BNBWD) || p.src[i].C % S) return 0;
        two = two || m == RD_SRC_BNBWD;
    }
    if (p.Cin % S) {                                        // a padded narrow source: only its first slot is live
        if (p.nsrc != 1 || p.Cin > S) return 0;
    }
    return two ? 2 : 1;
}

#ifndef PF_T
#define PF_T(ev) do { } while (0)          // conv_lean.hip (debug build) records shader-clock stamps of a few workgroups here
#endif
// ------------------------------------------------------------------------------------ chunk-pipelined kernel
// Same tile / LDS / MFMA layout as conv_kernel, for launches whose sources are all plain per-pixel reads of
// whole 16-byte channel slots (PlainSrc; the host checks).  The raw input vectors and the weight vectors of
// chunk c+1 are fetched into registers right after chunk c's LDS fill, so their L2/HBM latency runs under
// chunk c's MFMA phase instead of in front of chunk c+1's (the generic kernel is ~55% stalled on exactly that).
// NQ = 2: the dgrad form, two operands per item (g and z of the BN backward; a raw dz aliases z to g, q = 0).
// EP: 0 = LDS-staged epilogue (conv_epilogue), 1 / 2 = register epilogue forward / plain gradient (conv_epilogue_lean)
// TS: output-tile shape (conv_device.h TileGeo; the LDS-staged epilogue EP 0 knows shape 0 only)
template <typename T, int TAPS, int NB, int NQ, int EP = 0, int TS = 0>
__global__ __launch_bounds__(256, 2) void conv_pf_kernel(const rd_conv_t p, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                                // BatchNorm finalize folded into this launch (bn_fin.h)
    static_assert(TS == 0 || EP != 0, "tile shapes other than 8x32 need a register epilogue");
    constexpr int S = Slot<T>::N;
    constexpr int CK = 4 * S;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int TH = TileGeo<TS>::H, TW = TileGeo<TS>::W;     // (shadow the 8 x 32 globals)
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int PWL = TileGeo<TS>::LdsPitch(HALO);       // row pitch of the halo tile IN LDS (>= PW; conv_device.h TileGeo)
    constexpr int NT = NB * 32;
    constexpr int NIT = (PH * PW + 63) / 64;               // halo pixels per thread (fixed channel slot tid & 3)
    constexpr int WTOT = TAPS * NT * 4, WIT = (WTOT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [PH*PWL][4]
    uint4* s_w = s_in + PH * PWL * 4;                      // [TAPS][NT][4]

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

    const int s = tid & 3;
    // per-item geometry is recomputed where needed (a few integer ops) rather than kept live across the MFMA phase
    auto geom = [&](ItemGeom<NIT>& ig) {
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int pix = (tid >> 2) + 64 * b, py = pix / PW, px = pix - py * PW;
            ig.py[b] = (short)py;
            ig.px[b] = (short)px;
            const int lp = py * PWL + px;                  // the pixel's LDS record
            ig.lds[b] = pix < PH * PW ? lp * 4 + (s ^ ((lp >> 2) & 3)) : -1;
        }
    };
    const T* wbase = reinterpret_cast<const T*>(p.w);
    uint4 raw[NIT][NQ], wr[WIT];
    // this thread's channel slot of chunk c0 -> source + channel inside it (also recomputed, not kept)
    auto slot_src = [&](int c0, rd_src_t& sd, int& cc) -> bool {
        const int c = c0 + s * S;
        const int si = (p.nsrc == 1 || c < p.src[0].C) ? 0 : 1;
        sd = select_src(p.src, si);
        const bool live = c < p.Cin;
        cc = live ? c - (si ? p.src[0].C : 0) : 0;
        return live;
    };
    auto issue = [&](int c0) {
        rd_src_t sd;
        int cc;
        if (slot_src(c0, sd, cc)) {
            PlainSrc<T> ps;
            ItemGeom<NIT> ig;
            geom(ig);
            plain_src_init<T>(ps, sd, cc);
            pfu_issue<T, NIT>(raw, ps, ig, n, H, W, y0 - HALO, x0 - HALO);
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT, tap = rec / NT;
            wr[b] = ld16(wbase + ((size_t)((c0 / CK) * TAPS + min(tap, TAPS - 1)) * p.CoutPad + n0 + nn) * CK + sw * S);
        }
    };
    PF_T(0);
    issue(0);
    // BN coefficient rows of this image's group, all input channels, staged once: the per-chunk fill then reads them
    // from LDS (~100 cycles) instead of from L2 right in front of the transform (an exposed ~1 us per chunk)
    float* s_coef = reinterpret_cast<float*>(s_w + TAPS * NT * 4);        // [3][CinPad]: sc, sh, q
    float* s_epi = s_coef + 3 * p.CinPad;                                  // EP != 0: [2][NT] bias | producer scale, shift
    if constexpr (EP != 0) {
        if (tid < NT) {
            const int cch = n0 + tid;
            if constexpr (EP == 1) {
                s_epi[tid] = (p.bias && cch < p.Cout) ? p.bias[cch] : 0.f;
            } else {
                const int dj = cch >= p.c_split ? 1 : 0;
                const rd_dst_t dd = select_dst(p, dj);
                const int cdd = cch - (dj ? p.c_split : 0);
                const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
                const bool ok = cch < p.Cout && dd.kind != RD_DST_NONE && dd.scale && cdd < dd.Cd;
                s_epi[tid] = ok ? dd.scale[gd * dd.Cd + cdd] : 1.f;
                s_epi[NT + tid] = ok ? dd.shift[gd * dd.Cd + cdd] : 0.f;
            }
        }
    }
    for (int c = tid; c < p.CinPad; c += 256) {
        const int si = (p.nsrc == 1 || c < p.src[0].C) ? 0 : 1;
        const rd_src_t sd = select_src(p.src, si);
        const int cc = c - (si ? p.src[0].C : 0);
        const bool live = c < p.Cin, raw = sd.mode == RD_SRC_RAW, bwd = sd.mode == RD_SRC_BNBWD;
        const int gg = sd.g_fixed >= 0 ? sd.g_fixed : g;
        s_coef[c] = (live && !raw) ? sd.scale[gg * sd.C + cc] : 1.f;
        s_coef[p.CinPad + c] = (live && !raw) ? sd.shift[gg * sd.C + cc] : 0.f;
        s_coef[2 * p.CinPad + c] = (live && bwd) ? sd.q[gg * sd.C + cc] : 0.f;
    }
    for (int c0 = 0; c0 < p.CinPad; c0 += CK) {
        __syncthreads();
        {
            rd_src_t sd;
            int cc;
            ItemGeom<NIT> ig;
            geom(ig);
            if (slot_src(c0, sd, cc)) {
                PlainSrc<T> ps;
                plain_src_init<T>(ps, sd, cc);
                {
                    const float* cp = s_coef + c0 + s * S;
#pragma unroll
                    for (int e = 0; e < S; e += 4) {
                        const float4 a = *reinterpret_cast<const float4*>(cp + e);
                        const float4 b = *reinterpret_cast<const float4*>(cp + p.CinPad + e);
                        ps.sc[e] = a.x; ps.sc[e + 1] = a.y; ps.sc[e + 2] = a.z; ps.sc[e + 3] = a.w;
                        ps.sh[e] = b.x; ps.sh[e + 1] = b.y; ps.sh[e + 2] = b.z; ps.sh[e + 3] = b.w;
                        if constexpr (NQ == 2) {
                            const float4 d = *reinterpret_cast<const float4*>(cp + 2 * p.CinPad + e);
                            ps.q[e] = d.x; ps.q[e + 1] = d.y; ps.q[e + 2] = d.z; ps.q[e + 3] = d.w;
                        } else {
                            ps.q[e] = ps.q[e + 1] = ps.q[e + 2] = ps.q[e + 3] = 0.f;
                        }
                    }
                }
                pfu_consume<T, NIT>(raw, ps, ig, H, W, y0 - HALO, x0 - HALO, [&](int l, const uint4& u) { s_in[l] = u; });
                PF_T(1 + 3 * (c0 / CK));
            } else {
#pragma unroll
                for (int b = 0; b < NIT; ++b)
                    if (ig.lds[b] >= 0) s_in[ig.lds[b]] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT;
            if (idx < WTOT) s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = wr[b];
        }
        __syncthreads();
        PF_T(2 + 3 * (c0 / CK));
        if (c0 + CK < p.CinPad) issue(c0 + CK);
        conv_mma_chunk<T, TAPS, NB, EP != 0, TS>(s_in, s_w, wave, li, h, acc);
        PF_T(3 + 3 * (c0 / CK));
    }
    __syncthreads();
    PF_T(13);
    if constexpr (EP == 0) conv_epilogue<T, NB>(p, acc, smem, tid, n, g, y0, x0, n0, slot);
    else conv_epilogue_lean<T, NB, EP, TS>(p, acc, reinterpret_cast<double*>(smem), s_epi, tid, n, g, y0, x0, n0, slot);
    PF_T(14);
}


// dynamic LDS of conv_pf_kernel: halo tile + weight chunk + BN coefficient rows + (register epilogues) bias / producer rows
template <int TAPS, int NB, int TS = 0>
inline size_t conv_pf_lds(const rd_conv_t& p) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int TH = TileGeo<TS>::H, TW = TileGeo<TS>::W;
    constexpr int PH = TH + 2 * HALO, PW = TileGeo<TS>::LdsPitch(HALO);
    size_t lds = (size_t)(PH * PW * 4 + TAPS * NB * 32 * 4) * sizeof(uint4) + (size_t)3 * p.CinPad * sizeof(float) + (size_t)2 * NB * 32 * sizeof(float);
    const size_t lds_epi = (size_t)TH * TW * 32 * sizeof(float) + 64 * sizeof(double);
    return lds < lds_epi ? lds_epi : lds;
}

}  // namespace
