// conv_small.hip -- persistent small-channel conv kernels (Cin <= one 64-byte chunk, Cout <= 32): every 400x400 /
// 200x200 layer of the U-Net, forward and dgrad.  See the comment above conv_small_kernel.
#include "conv_device.h"
#include "conv_dispatch.h"

namespace {

// SRCG: sources may need the generic (synchronous) loader: max-pool / upsample / odd channel counts.
// EPI : 0 forward, 1 gradient with plain full-slot destinations only (lean path), 3 plain + upsample-side,
//       4 plain + max-pool, 2 anything.
template <typename T, int TAPS, bool SRCG, int EPI>
__global__ __launch_bounds__(256, 2) void conv_small_kernel(const rd_conv_t p, int tiles_per_wg, const rdfin::FinArg fa) {
    constexpr int S = Slot<T>::N;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NPIX = PH * PW, NT = 32;
    constexpr int NIT = (NPIX * 4 + 255) / 256;
    constexpr int NV = 16 / S;                             // output vectors of S contiguous channels per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [NPIX][4]
    uint4* s_w = s_in + NPIX * 4;                          // [TAPS][32][4]
    double* s_red = reinterpret_cast<double*>(s_w + TAPS * NT * 4);  // [32][2], fp64: see conv_device.h flush_bstats
    float* s_dsc = reinterpret_cast<float*>(s_red + 64);                             // [32] producer scale of the gradient destinations
    float* s_dsh = s_dsc + 32;                             // [32] producer shift
    float* s_fin = s_dsh + 32;                             // [rdfin::FIN_LDS_FLOATS] coefficient table of a folded finalize (bn_fin.h)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % rd_stat_nslots(p.stat_slots);

    // ---- LDS prologue: zero the input tile once (channel slots beyond Cin stay zero for every tile), packed
    //      weights once per workgroup, destination BN coefficients of the gradient epilogues
    for (int i = tid; i < NPIX * 4; i += 256) s_in[i] = make_uint4(0, 0, 0, 0);
    {
        const T* wbase = reinterpret_cast<const T*>(p.w);
        constexpr int WTOT = TAPS * NT * 4, WIT = (WTOT + 255) / 256;
        uint4 wr[WIT];
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT, tap = rec / NT;
            wr[b] = ld16(wbase + ((size_t)(min(tap, TAPS - 1) * (p.w_tap_rows ? p.w_tap_rows : p.CoutPad) + nn) * p.CinPad + sw * S));
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT;
            if (idx < WTOT) s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = wr[b];
        }
    }
    if (tid < 64) s_red[tid] = 0.0;
    if constexpr (EPI > 0) {
        if (tid < 32) {
            const int dj = tid >= p.c_split ? 1 : 0;
            const rd_dst_t dd = select_dst(p, dj);
            const int cdd = tid - (dj ? p.c_split : 0);
            const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
            const bool ok = tid < p.Cout && dd.kind != RD_DST_NONE && dd.scale && cdd < dd.Cd;
            s_dsc[tid] = ok ? dd.scale[gd * dd.Cd + cdd] : 1.f;
            s_dsh[tid] = ok ? dd.shift[gd * dd.Cd + cdd] : 0.f;
        }
    }

    // ---- loader geometry: items = (halo pixel, LIVE channel slot).  16-channel bf16 layers have 2 live slots of
    //      4: all 256 threads work on them (instead of half the lanes idling on zero slots)
    int nsl = 4;
    if constexpr (!SRCG) {
        const int nl = (p.Cin + S - 1) / S;
        nsl = nl <= 1 ? 1 : (nl <= 2 ? 2 : 4);
    }
    const int nsh = nsl == 1 ? 0 : (nsl == 2 ? 1 : 2);
    const int sslot = tid & (nsl - 1);
    const int nit = (NPIX * nsl + 255) >> 8;
    ItemGeom<NIT> ig;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int pixi = (tid + b * 256) >> nsh;
        const int pix = min(pixi, NPIX - 1);
        ig.py[b] = (short)(pix / PW);
        ig.px[b] = (short)(pix - (pix / PW) * PW);
        ig.lds[b] = pixi < NPIX ? pix * 4 + (sslot ^ ((pix >> 2) & 3)) : -1;
    }
    // BatchNorm finalize of the producer folded into this launch (bn_fin.h), behind the weight loads: the coefficients reach slot_ctx
    // through LDS (src[] points there)
    rd_src_t src[2];
    rdfin::conv_prologue_lds(p, fa, s_fin, rdfin::FIN_LDS_FLOATS, src);
    SlotCtx<T> ctx;
    slot_ctx<T>(ctx, src, p.nsrc, p.Cin, g, sslot * S);
    const rd_src_t ssrc = select_src(src, ctx.si > 0 ? 1 : 0);
    const bool live_slot = ctx.si >= 0;
    bool pre = true;
    if constexpr (SRCG)
        pre = live_slot && (ssrc.C % S) == 0 && (ssrc.C - ctx.c) >= S &&
              (ssrc.mode == RD_SRC_RAW || ssrc.mode == RD_SRC_AFF || ssrc.mode == RD_SRC_AFFACT || ssrc.mode == RD_SRC_BNBWD);
    int nks = 2;
    if constexpr (sizeof(T) == 2) nks = p.Cin <= 16 ? 1 : 2;

    // ---- epilogue: after the (weights x pixels) MFMA a lane owns ONE pixel (column lane&31) and 16 output
    //      channels; they are regrouped into NV vectors of S contiguous channels (bf16: one half-wave exchange, rd_half_swap
    //      per pair of registers) so that all global traffic of the epilogue is 16-byte, straight from registers
    int cbv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) cbv[v] = (S == 8) ? 16 * v + 8 * h : 8 * v + 4 * h;
    float sa[NV][S], sb[NV][S], bs[NV][S];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < S; ++e) {
            sa[v][e] = sb[v][e] = 0.f;
            bs[v][e] = (EPI == 0 && p.bias && cbv[v] + e < p.Cout) ? p.bias[cbv[v] + e] : 0.f;
        }
    T* out = reinterpret_cast<T*>(p.out);

    uint4 raw[NIT][2];
    __syncthreads();                                       // zero-fill / weights / coefficients published
    if (pre && live_slot && t_begin < t_end)
        pf_issue<T, NIT>(raw, ssrc, ctx, ig, n, H, W, (t_begin / tiles_x) * TH - HALO, (t_begin % tiles_x) * TW - HALO, nit);

    for (int t = t_begin; t < t_end; ++t) {
        const int x0 = (t % tiles_x) * TW, y0 = (t / tiles_x) * TH;
        if (pre) {
            if (live_slot) pf_consume<T, NIT>(raw, ssrc, ctx, ig, H, W, y0 - HALO, x0 - HALO, s_in, nit);
        } else if constexpr (SRCG) {
            auto map = [&](int idx, int& y, int& x) -> bool {
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                y = y0 - HALO + py;
                x = x0 - HALO + px;
                return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            };
            auto store = [&](int idx, const uint4& u) {
                const int pix = idx >> 2;
                s_in[pix * 4 + (sslot ^ ((pix >> 2) & 3))] = u;
            };
            if (live_slot) tile_fill<T>(src, ctx, n, H, W, tid, NPIX * 4, map, store);
        }
        __syncthreads();
        if (pre && live_slot && t + 1 < t_end)
            pf_issue<T, NIT>(raw, ssrc, ctx, ig, n, H, W, ((t + 1) / tiles_x) * TH - HALO, ((t + 1) % tiles_x) * TW - HALO, nit);

        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int pix = (wave * 2 + mb + kh) * PW + li + kw;
                // A = packed weights (M = output channel), B = pixels (N = pixel of the tile row)
                Mma<T>::chunk(s_w + (tap * NT + li) * 4, (li >> 2) & 3, s_in + pix * 4, (pix >> 2) & 3, h, acc[mb], nks);
            }
        }
        // plain gradient destinations of the 1x1 kernel: the producer tensors / old gradients of all 2 x NV vectors are requested
        // together, ahead of the first use (conv_device.h grad_plain_issue).  The 3x3 kernel has no registers left for the requests:
        // with them it spills (52-116 bytes per lane) and the row-block launches of dec.convu1.conv1's dgrad go from 58 + 54 to
        // 81 + 71 us, so it keeps the vector-by-vector grad_plain.
        GradPlainReq gq[2][NV];
        auto issue_row = [&](int mb) {
            const int y = y0 + wave * 2 + mb, x = x0 + li;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cb = cbv[v];
                const int di = cb >= p.c_split ? 1 : 0;
                const rd_dst_t d = select_dst(p, di);
                grad_plain_issue<T>(gq[mb][v], d, y < H && x < W && cb < p.Cout && d.kind != RD_DST_NONE, n, y, x, H, W,
                                    cb - (di ? p.c_split : 0), p.w);
            }
        };
        if constexpr (EPI == 1 && TAPS == 1) {
            issue_row(0);
            issue_row(1);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int y = y0 + wave * 2 + mb, x = x0 + li;
            const bool valid = y < H && x < W;
            float vec[NV][S];
            if constexpr (S == 8) {
                // accumulator rows: channel (r&3) + 8*(r>>2) + 4*h.  Groups q = r>>2: pair (2v, 2v+1) -> lanes h=0
                // end up with channels 16v..16v+7, lanes h=1 with 16v+8..16v+15
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned a = __float_as_uint(acc[mb][8 * v + j]);
                        const unsigned b = __float_as_uint(acc[mb][8 * v + 4 + j]);
                        const HalfSwap r = rd_half_swap(a, b, h);
                        vec[v][j] = __uint_as_float(r.r0);
                        vec[v][4 + j] = __uint_as_float(r.r1);
                    }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) vec[v][j] = acc[mb][4 * v + j];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cb = cbv[v];
                if (!(valid && cb < p.Cout)) continue;
                if constexpr (EPI == 0) {
                    float o[S];
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        o[e] = vec[v][e] + bs[v][e];
                        sa[v][e] += vec[v][e];                    // sums exclude the bias (ramdsir.h, RD_STAT_SLOTS)
                        sb[v][e] += vec[v][e] * vec[v][e];
                    }
                    store_vec<T>(out + ((size_t)(n * H + y) * W + x) * p.Cout + cb, o, p.Cout - cb, (p.Cout % S) == 0);
                } else {
                    const int di = cb >= p.c_split ? 1 : 0;
                    const rd_dst_t d = select_dst(p, di);
                    if (d.kind == RD_DST_NONE) continue;
                    const int cd = cb - (di ? p.c_split : 0);
                    float dsc[S], dsh[S];
#pragma unroll
                    for (int e = 0; e < S; e += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(s_dsc + cb + e), b4 = *reinterpret_cast<const float4*>(s_dsh + cb + e);
                        dsc[e] = a4.x; dsc[e + 1] = a4.y; dsc[e + 2] = a4.z; dsc[e + 3] = a4.w;
                        dsh[e] = b4.x; dsh[e + 1] = b4.y; dsh[e + 2] = b4.z; dsh[e + 3] = b4.w;
                    }
                    if constexpr (EPI == 1 && TAPS == 1) {
                        grad_plain_finish<T>(gq[mb][v], d, vec[v], dsc, dsh, sa[v], sb[v]);
                    } else if constexpr (EPI == 1) {
                        grad_plain<T>(d, n, y, x, H, W, cd, vec[v], dsc, dsh, sa[v], sb[v]);
                    } else {
                        constexpr int KM = EPI == 3 ? 5 : (EPI == 4 ? 3 : 7);
                        grad_item<T, KM>(d, g, n, y, x, H, W, cd, vec[v], dsc, dsh, sa[v], sb[v]);
                    }
                }
            }
        }
        __syncthreads();                                   // the next tile rewrites s_in
    }

    // ---- flush the BN sums of all tiles of this workgroup: lanes of a half-wave hold different pixels of the
    //      same channels -> summed over the 32 lanes (conv_device.h half_wave_sums), one LDS atomic per lane
    flush_half_wave_sums16<S, NV>(s_red, sa, sb, li, h);
    __syncthreads();
    if (tid < 32 && tid < p.Cout) {
        if constexpr (EPI == 0) {
            if (p.stats) {
                const size_t so = (((size_t)g * RD_STAT_SLOTS + slot) * p.Cout + tid) * 2;
                atomicAdd(&p.stats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&p.stats[so + 1], s_red[tid * 2 + 1]);
            }
        } else {
            const int dj = tid >= p.c_split ? 1 : 0;
            const rd_dst_t dd = select_dst(p, dj);
            if (dd.kind != RD_DST_NONE && dd.bstats) {
                const int cdd = tid - (dj ? p.c_split : 0);
                const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
                const size_t so = (((size_t)gd * RD_STAT_SLOTS + slot) * dd.Cd + cdd) * 2;
                atomicAdd(&dd.bstats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&dd.bstats[so + 1], s_red[tid * 2 + 1]);
            }
        }
    }
}

// Variant with the accumulators staged through LDS ([256 pixels][32 ch] fp32): used where the register epilogue
// of conv_small_kernel would spill (max-pool / upsample-side destinations, generic sources).
template <typename T, int TAPS, bool SRCG, int EPI>
__global__ __launch_bounds__(256, 2) void conv_small_stage_kernel(const rd_conv_t p, int tiles_per_wg, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                                // BatchNorm finalize folded into this launch (bn_fin.h)
    constexpr int S = Slot<T>::N;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NT = 32, SL = 32 / S;
    constexpr int TOTAL = PH * PW * 4, NIT = (TOTAL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [PH*PW][4]
    uint4* s_w = s_in + PH * PW * 4;                       // [TAPS][32][4]
    float* s_out = reinterpret_cast<float*>(s_w + TAPS * NT * 4);   // [TH*TW][32]
    double* s_red = reinterpret_cast<double*>(s_out + TH * TW * 32);   // [32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % rd_stat_nslots(p.stat_slots);

    // ---- packed weights: once per workgroup
    {
        const T* wbase = reinterpret_cast<const T*>(p.w);
        constexpr int WTOT = TAPS * NT * 4, WIT = (WTOT + 255) / 256;
        uint4 wr[WIT];
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT, tap = rec / NT;
            wr[b] = ld16(wbase + ((size_t)(min(tap, TAPS - 1) * (p.w_tap_rows ? p.w_tap_rows : p.CoutPad) + nn) * p.CinPad + sw * S));
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT;
            if (idx < WTOT) s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = wr[b];
        }
    }
    if (tid < 64) s_red[tid] = 0.0;

    // ---- loader constants of this thread (its channel slot never changes)
    const int sslot = tid & 3;
    SlotCtx<T> ctx;
    slot_ctx<T>(ctx, p.src, p.nsrc, p.Cin, g, sslot * S);
    const rd_src_t ssrc = select_src(p.src, ctx.si > 0 ? 1 : 0);
    // !SRCG: the host guarantees simple modes and full slots; a slot beyond Cin prefetches zeros (mode RAW)
    bool pre = true;
    if constexpr (SRCG)
        pre = ctx.si >= 0 && (ssrc.C % S) == 0 && (ssrc.C - ctx.c) >= S &&
              (ssrc.mode == RD_SRC_RAW || ssrc.mode == RD_SRC_AFF || ssrc.mode == RD_SRC_AFFACT || ssrc.mode == RD_SRC_BNBWD);
    const bool live_slot = ctx.si >= 0;

    // ---- epilogue constants of this thread
    const int sl = tid % SL;
    const int c = sl * S;
    const int di = (p.emode == 1 && c >= p.c_split) ? 1 : 0;
    const rd_dst_t d = select_dst(p, di);
    const int cd = c - (di ? p.c_split : 0);
    float dsc[S], dsh[S], b1[S], b2[S];
    {
        const int gd = d.g_fixed >= 0 ? d.g_fixed : g;
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const bool ok = p.emode == 1 && c < p.Cout && d.kind != RD_DST_NONE && d.scale && (cd + e < d.Cd);
            dsc[e] = ok ? d.scale[gd * d.Cd + cd + e] : 1.f;
            dsh[e] = ok ? d.shift[gd * d.Cd + cd + e] : 0.f;
            b1[e] = b2[e] = 0.f;
        }
    }
    const bool cok = li < p.Cout;
    const float bsv = (p.emode == 0 && cok && p.bias) ? p.bias[li] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    T* out = reinterpret_cast<T*>(p.out);

    ItemGeom<NIT> ig;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        const int idx = tid + b * 256;
        const int pix = min(idx, TOTAL - 1) >> 2;
        ig.py[b] = (short)(pix / PW);
        ig.px[b] = (short)(pix - (pix / PW) * PW);
        ig.lds[b] = idx < TOTAL ? pix * 4 + (sslot ^ ((pix >> 2) & 3)) : -1;
    }
    uint4 raw[NIT][2];
    if (pre && live_slot && t_begin < t_end)
        pf_issue<T, NIT>(raw, ssrc, ctx, ig, n, H, W, (t_begin / tiles_x) * TH - HALO, (t_begin % tiles_x) * TW - HALO);

    for (int t = t_begin; t < t_end; ++t) {
        const int x0 = (t % tiles_x) * TW, y0 = (t / tiles_x) * TH;
        if (pre) {
            if (live_slot) {
                pf_consume<T, NIT>(raw, ssrc, ctx, ig, H, W, y0 - HALO, x0 - HALO, s_in);
            } else {
                for (int idx = tid; idx < TOTAL; idx += 256) {
                    const int pix = idx >> 2;
                    s_in[pix * 4 + (sslot ^ ((pix >> 2) & 3))] = make_uint4(0, 0, 0, 0);
                }
            }
        } else if constexpr (SRCG) {
            auto map = [&](int idx, int& y, int& x) -> bool {
                const int pix = idx >> 2;
                const int py = pix / PW, px = pix - py * PW;
                y = y0 - HALO + py;
                x = x0 - HALO + px;
                return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            };
            auto store = [&](int idx, const uint4& u) {
                const int pix = idx >> 2;
                s_in[pix * 4 + (sslot ^ ((pix >> 2) & 3))] = u;
            };
            tile_fill<T>(p.src, ctx, n, H, W, tid, TOTAL, map, store);
        }
        __syncthreads();
        if (pre && live_slot && t + 1 < t_end)
            pf_issue<T, NIT>(raw, ssrc, ctx, ig, n, H, W, ((t + 1) / tiles_x) * TH - HALO, ((t + 1) % tiles_x) * TW - HALO);

        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int pix = (wave * 2 + mb + kh) * PW + li + kw;
                Mma<T>::chunk(s_in + pix * 4, (pix >> 2) & 3, s_w + (tap * NT + li) * 4, (li >> 2) & 3, h, acc[mb]);
            }
        }
        // stage the 32-channel block (s_out is a separate LDS region: no barrier needed before writing it)
        const bool interior = (x0 + TW <= W) && (y0 + TH <= H);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int y = y0 + wave * 2 + mb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = acc[mb][r] + bsv;
                s_out[((wave * 2 + mb) * TW + col) * 32 + li] = v;
                if constexpr (EPI == 0) {
                    if (interior || (y < H && x0 + col < W)) { s1 += acc[mb][r]; s2 += acc[mb][r] * acc[mb][r]; }   // bias-free sums
                }
            }
        }
        __syncthreads();
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
                if constexpr (EPI == 0) {
                    store_vec<T>(out + ((size_t)(n * H + y) * W + x) * p.Cout + c, v, p.Cout - c, (p.Cout % S) == 0);
                } else if constexpr (EPI == 1) {
                    if (d.kind != RD_DST_NONE) grad_plain<T>(d, n, y, x, H, W, cd, v, dsc, dsh, b1, b2);
                } else {
                    constexpr int KM = EPI == 3 ? 5 : (EPI == 4 ? 3 : 7);
                    if (d.kind != RD_DST_NONE) grad_item<T, KM>(d, g, n, y, x, H, W, cd, v, dsc, dsh, b1, b2);
                }
            }
        }
        __syncthreads();                                   // s_in / s_out are rewritten by the next tile
    }

    // ---- flush the BN sums of all tiles of this workgroup
    if constexpr (EPI == 0) {
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0 && cok) {
                atomicAdd(&s_red[li * 2 + 0], (double)s1);
                atomicAdd(&s_red[li * 2 + 1], (double)s2);
            }
        }
    } else {
        flush_bstats<S, SL>(s_red, lane, sl, b1, b2);
    }
    __syncthreads();
    if (tid < 32 && tid < p.Cout) {
        if (p.emode == 0) {
            if (p.stats) {
                const size_t so = (((size_t)g * RD_STAT_SLOTS + slot) * p.Cout + tid) * 2;
                atomicAdd(&p.stats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&p.stats[so + 1], s_red[tid * 2 + 1]);
            }
        } else {
            const int dj = tid >= p.c_split ? 1 : 0;
            const rd_dst_t dd = select_dst(p, dj);
            if (dd.kind != RD_DST_NONE && dd.bstats) {
                const int cdd = tid - (dj ? p.c_split : 0);
                const int gd = dd.g_fixed >= 0 ? dd.g_fixed : g;
                const size_t so = (((size_t)gd * RD_STAT_SLOTS + slot) * dd.Cd + cdd) * 2;
                atomicAdd(&dd.bstats[so + 0], s_red[tid * 2 + 0]);
                atomicAdd(&dd.bstats[so + 1], s_red[tid * 2 + 1]);
            }
        }
    }
}

template <typename T, int TAPS, bool SRCG, int EPI>
int launch_conv_small(const rd_conv_t& p, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr bool REG_EPI = !SRCG && EPI <= 1;            // register epilogue where it stays spill-free
    const size_t lds = (size_t)(PH * PW * 4 + TAPS * 32 * 4) * sizeof(uint4) +
                       (REG_EPI ? (size_t)64 * sizeof(double) + 64 * sizeof(float) + rdfin::FIN_LDS_FLOATS * sizeof(float)
                                : (size_t)TH * TW * 32 * sizeof(float) + 64 * sizeof(double));
    const int ntiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    static int tpw_env = -1;
    if (tpw_env < 0) tpw_env = rd_switch("RD_TPW", 0);
    // tiles per workgroup: the launch runs in rounds of `slots` resident workgroups (2 per CU); pick the
    // count whose last round is (nearly) full -- e.g. 16 images x 650 tiles: 4 tiles/workgroup is 6 rounds
    // x 4 tile-times, 21 tiles/workgroup is 1 round x 21 -- charging half a tile-time per workgroup prologue
    int tpw = tpw_env;
    if (tpw <= 0) {
        const bool limited = p.cu_limit > 0 && p.cu_limit < rd_num_cus();
        const long slots = 2L * (limited ? p.cu_limit : rd_num_cus());
        double best = 1e30;
        // a limited launch (side lane: ramdsir.h cu_limit) must fit ONE round of its budget, however many tiles that takes
        const int tmin = limited ? (int)(((long)ntiles * p.N + slots - 1) / slots) : 1;
        for (int t = tmin < 1 ? 1 : tmin; (t <= 32 || limited) && t <= ntiles; ++t) {
            const long wgs = (long)((ntiles + t - 1) / t) * p.N;
            const double cost = (double)((wgs + slots - 1) / slots) * (t + 0.5);
            if (cost < best - 1e-9) { best = cost; tpw = t; }
            if (limited) break;                             // the smallest count that fits is the one
        }
        if (tpw <= 0) tpw = ntiles;
    }
    dim3 grid((ntiles + tpw - 1) / tpw, 1, p.N);
    static bool attr_set = false;
    if constexpr (REG_EPI) {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<T, TAPS, SRCG, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        rd_launch((conv_small_kernel<T, TAPS, SRCG, EPI>), grid, dim3(256), lds, st, p, tpw, rdfin::current());
    } else {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_stage_kernel<T, TAPS, SRCG, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        rd_launch((conv_small_stage_kernel<T, TAPS, SRCG, EPI>), grid, dim3(256), lds, st, p, tpw, rdfin::current());
    }
    return (int)hipGetLastError();
}

template <typename T, int TAPS>
int dispatch_conv_small(const rd_conv_t& p, hipStream_t st) {
    constexpr int S = Slot<T>::N;
    bool srcg = false;
    for (int i = 0; i < p.nsrc; ++i) {
        const int m = p.src[i].mode;
        if (!(m == RD_SRC_RAW || m == RD_SRC_AFF || m == RD_SRC_AFFACT || m == RD_SRC_BNBWD) || (p.src[i].C % S)) srcg = true;
    }
    int epi = 0;
    if (p.emode == 1) {
        epi = 1;
        bool pool = false, upy = false, narrow = false;
        for (int i = 0; i < 2; ++i) {
            const rd_dst_t& d = p.dst[i];
            if (d.kind == RD_DST_NONE) continue;
            pool |= d.kind == RD_DST_POOL;
            upy |= d.kind == RD_DST_UPY;
            narrow |= (d.Cd % S) != 0;
        }
        if (pool && upy) epi = 2;
        else if (upy) epi = 3;
        else if (pool) epi = 4;
        else if (narrow) epi = 2;
    } else if (p.Cout % S) {
        epi = 0;                                           // store_vec handles narrow outputs
    }
#define RD_CS(SG, EP) return launch_conv_small<T, TAPS, SG, EP>(p, st)
    if (!srcg) {
        if (epi == 0) RD_CS(false, 0);
        if (epi == 1) RD_CS(false, 1);
        if (epi == 3) RD_CS(false, 3);
        if (epi == 4) RD_CS(false, 4);
        RD_CS(false, 2);
    }
    if (epi == 0) RD_CS(true, 0);
    if (epi == 1) RD_CS(true, 1);
    RD_CS(true, 2);
#undef RD_CS
}

}  // namespace

int rd_conv_small_dispatch(const rd_conv_t& p, int dtype, hipStream_t st) {
    if (dtype == RD_BF16 && p.taps == 9 && p.emode == 0) {                // forward launches: conv_small_fwd_kernel where it applies
        const int r = rd_conv_small_fwd_dispatch(p, dtype, st);
        if (r != RD_CONV_PP_NA) return r;
    }
    if (dtype == RD_BF16) return p.taps == 9 ? dispatch_conv_small<bf16_t, 9>(p, st) : dispatch_conv_small<bf16_t, 1>(p, st);
    return p.taps == 9 ? dispatch_conv_small<float, 9>(p, st) : dispatch_conv_small<float, 1>(p, st);
}

