// conv_device.h -- device-side building blocks shared by the conv / wgrad translation units: tile loaders with the
// fused BN / activation / pool / upsample transforms, MFMA atoms, gradient epilogues, register prefetch helpers.
#pragma once
#include "common.h"
#include "../../include/ramdsir.h"
#include "bn_fin.h"
#include <stdlib.h>

namespace {


constexpr int TH = 8, TW = 32;

// Output-tile shapes of the 256-pixel conv tiles.  A workgroup's four waves own 8 M-blocks of 32 pixels; shape 0 maps M-block
// m, lane li to tile pixel (row m, column li) of an 8 x 32 rectangle.  On images whose side is not a multiple of 32 that leaves
// 22-39 % of the MFMA lanes on pixels outside the image (100 -> 4 x 32 columns, 50 -> 2 x 32, 25 -> 32).  Shape 1 flattens the
// 256 lanes over a 10 x 25 rectangle (pixel j = 32 m + li -> row j / 25, column j % 25; 250 of 256 lanes live): sides 25, 50, 100
// and 200 are covered exactly in x and within 0-20 % in y, with a halo tile (12 x 27) that is no larger than shape 0's (10 x 34).
// LdsPitch: row pitch (pixels) of the halo tile in LDS.  The MFMA fragment reads (ds_read_b128, serviced in 16-lane groups over a
// 256-byte bank row = 16 pixel entries of one channel slot) are conflict-free when the 16 lanes of a group read pixels that are
// distinct mod 16.  Shape 0: lane li reads pixel base + li of one tile row -- consecutive.  Shape 1: the 32 lanes of an M-block
// straddle one or two row breaks, and with the natural pitch 27 the lanes behind a break land 2 entries further on, on the entries
// of two other lanes of their group (2-way conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.20-0.31 in the round-3 counters of
// every kernel that ran shape 1).  Pitch 41 = 25 + 16 makes the pixel index of lane j equal to j + 16 * row: distinct mod 16 again.
template <int TS> struct TileGeo {
    static constexpr int H = 8, W = 32;
    static constexpr int LdsPitch(int halo) { return W + 2 * halo; }
};
template <> struct TileGeo<1> {
    static constexpr int H = 10, W = 25;
    static constexpr int LdsPitch(int halo) { return halo ? W + 16 : W; }
};

template <int TS>
__device__ __forceinline__ void tile_pixel(int mblock, int li, int& r, int& c, bool& live) {
    if constexpr (TS == 0) {
        r = mblock; c = li; live = true;
    } else {
        const int j = mblock * 32 + li;
        r = j / TileGeo<TS>::W;
        c = j - r * TileGeo<TS>::W;
        live = j < TileGeo<TS>::H * TileGeo<TS>::W;
    }
}

// share of a launch's MFMA lanes that land on image pixels under a tile shape (host side: picks the shape)
inline double tile_efficiency(int H, int W, int th, int tw) {
    const long tiles = (long)((H + th - 1) / th) * ((W + tw - 1) / tw);
    return (double)H * W / (double)(tiles * 256);
}

// dz = P*g + Q*z + R of the BatchNorm backward (and sc*x + q*x2 + sh of the unified plain-source form) with the two fused
// multiply-adds written out: left to the compiler, `a*b + c*d + e` contracts as fma(a, b, fma(c, d, e)) in one kernel and as
// fma(c, d, fma(a, b, e)) in another (it depends on what the surrounding code lets the vectoriser do), the results differ in the
// last fp32 bit, and one element in ~10^5 then rounds to the other bf16 -- enough to break "the fused backward's input gradient is
// bit-identical to rd_conv's" (tests/test_gpu_fused_bwd.py) on an unlucky case.
__device__ __forceinline__ float bn_bwd_value(float sc, float x, float q, float x2, float sh) {
    return __builtin_fmaf(sc, x, __builtin_fmaf(q, x2, sh));
}

// ------------------------------------------------------------------------------------ tile loader
template <typename T>
__device__ __forceinline__ void load_vec(const T* p, int cvalid, bool vec_ok, float* f) {
    constexpr int S = Slot<T>::N;
    if (vec_ok && cvalid >= S) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        Slot<T>::unpack(u, f);
    } else {
#pragma unroll
        for (int e = 0; e < S; ++e) f[e] = (e < cvalid) ? to_f<T>(p[e]) : 0.f;
    }
}

// Per-thread, per-chunk constants of the tile loader: which source the thread's channel slot belongs to
// and that slot's BN coefficients (each thread keeps ONE slot index for a whole chunk, so these are loaded
// once per chunk instead of once per pixel).
template <typename T>
struct SlotCtx {
    static constexpr int S = Slot<T>::N;
    int si;        // source index, -1: channel slot beyond Cin (zeros)
    int c;         // channel within the source
    float sc[S], sh[S], q[S];
};

// the geometry part of slot_ctx alone: which source channel c belongs to (no coefficient loads)
template <typename T>
__device__ __forceinline__ void slot_geom(SlotCtx<T>& k, const rd_src_t* src, int nsrc, int Cin, int c) {
    k.si = -1;
    k.c = 0;
    if (c >= Cin) return;
    k.si = (nsrc == 1 || c < src[0].C) ? 0 : 1;
    k.c = c - (k.si ? src[0].C : 0);
}

template <typename T>
__device__ __forceinline__ void slot_ctx(SlotCtx<T>& k, const rd_src_t* src, int nsrc, int Cin, int g_img, int c) {
    constexpr int S = Slot<T>::N;
    k.si = -1;
    k.c = 0;
    if (c >= Cin) return;
    k.si = (nsrc == 1 || c < src[0].C) ? 0 : 1;
    k.c = c - (k.si ? src[0].C : 0);
    // field-by-field select keeps the kernarg struct out of scratch
    const int C = k.si ? src[1].C : src[0].C;
    const int mode = k.si ? src[1].mode : src[0].mode;
    const int gf = k.si ? src[1].g_fixed : src[0].g_fixed;
    const float* scp = k.si ? src[1].scale : src[0].scale;
    const float* shp = k.si ? src[1].shift : src[0].shift;
    const float* qp = k.si ? src[1].q : src[0].q;
    const int cvalid = C - k.c;
    const int g = gf >= 0 ? gf : g_img;
#pragma unroll
    for (int e = 0; e < S; ++e) {
        const bool ok = (e < cvalid) && mode != RD_SRC_RAW;
        k.sc[e] = ok ? scp[g * C + k.c + e] : 0.f;
        k.sh[e] = ok ? shp[g * C + k.c + e] : 0.f;
        k.q[e] = (ok && mode == RD_SRC_BNBWD) ? qp[g * C + k.c + e] : 0.f;
    }
}

// One 16-byte slot of conv-input pixel (n, y, x) of source s, transformed.  (y, x) are in the conv's
// H x W frame and inside the image.
template <typename T>
__device__ __forceinline__ void load_slot(const rd_src_t& s, const SlotCtx<T>& k, int n, int y, int x, int H, int W, float* v) {
    constexpr int S = Slot<T>::N;
    const int C = s.C, c = k.c;
    const int cvalid = C - c;
    const bool vec_ok = (C % S) == 0;
    const T* base = reinterpret_cast<const T*>(s.ptr);
    n += s.n_off;
    switch (s.mode) {
    case RD_SRC_RAW: {
        load_vec<T>(base + ((size_t)(n * H + y) * W + x) * C + c, cvalid, vec_ok, v);
    } break;
    case RD_SRC_AFF: {
        load_vec<T>(base + ((size_t)(n * H + y) * W + x) * C + c, cvalid, vec_ok, v);
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = v[e] * k.sc[e] + k.sh[e];
    } break;
    case RD_SRC_AFFACT: {
        load_vec<T>(base + ((size_t)(n * H + y) * W + x) * C + c, cvalid, vec_ok, v);
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = act_fn(v[e] * k.sc[e] + k.sh[e], s.slope);
    } break;
    case RD_SRC_POOL: {
        const int Hs = 2 * H, Ws = 2 * W;
        float t[4][S];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            load_vec<T>(base + ((size_t)(n * Hs + 2 * y + (j >> 1)) * Ws + 2 * x + (j & 1)) * C + c, cvalid, vec_ok, t[j]);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            float m = act_fn(t[0][e] * k.sc[e] + k.sh[e], s.slope);
#pragma unroll
            for (int j = 1; j < 4; ++j) m = fmaxf(m, act_fn(t[j][e] * k.sc[e] + k.sh[e], s.slope));
            v[e] = m;
        }
    } break;
    case RD_SRC_UP: {
        const int Hs = H >> 1, Ws = W >> 1;
        int y0, y1, x0, x1;
        float ly, lx;
        up2_coord(y, Hs, y0, y1, ly);
        up2_coord(x, Ws, x0, x1, lx);
        float t00[S], t01[S], t10[S], t11[S];
        load_vec<T>(base + ((size_t)(n * Hs + y0) * Ws + x0) * C + c, cvalid, vec_ok, t00);
        load_vec<T>(base + ((size_t)(n * Hs + y0) * Ws + x1) * C + c, cvalid, vec_ok, t01);
        load_vec<T>(base + ((size_t)(n * Hs + y1) * Ws + x0) * C + c, cvalid, vec_ok, t10);
        load_vec<T>(base + ((size_t)(n * Hs + y1) * Ws + x1) * C + c, cvalid, vec_ok, t11);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            float top = t00[e] + lx * (t01[e] - t00[e]);
            float bot = t10[e] + lx * (t11[e] - t10[e]);
            float u = top + ly * (bot - top);
            v[e] = act_fn(u * k.sc[e] + k.sh[e], s.slope);
        }
    } break;
    case RD_SRC_BNBWD: {
        const T* zb = reinterpret_cast<const T*>(s.ptr2);
        const size_t off = ((size_t)(n * H + y) * W + x) * C + c;
        float gz[S], zz[S];
        load_vec<T>(base + off, cvalid, vec_ok, gz);
        load_vec<T>(zb + off, cvalid, vec_ok, zz);
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = bn_bwd_value(k.sc[e], gz[e], k.q[e], zz[e], k.sh[e]);
    } break;
    default:
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < S; ++e)
        if (e >= cvalid) v[e] = 0.f;
}

// slot of conv-input pixel (n,y,x) for the thread's channel slot; zero outside the image / channels
template <typename T>
__device__ __forceinline__ uint4 gather_slot(const rd_src_t* src, const SlotCtx<T>& k, int n, int y, int x, int H, int W) {
    constexpr int S = Slot<T>::N;
    float v[S];
#pragma unroll
    for (int e = 0; e < S; ++e) v[e] = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && k.si >= 0) {
        if (k.si == 0)
            load_slot<T>(src[0], k, n, y, x, H, W, v);
        else
            load_slot<T>(src[1], k, n, y, x, H, W, v);
    }
    return Slot<T>::pack(v);
}

// ------------------------------------------------------------------------------------ loads-first tile fill
// The tile loaders keep many independent 16-byte loads in flight per thread (the layers are HBM-bound):
// phase A issues the loads of a whole batch of items, phase B transforms and stores them to LDS.  The
// source mode is resolved OUTSIDE the item loops so that each loop body is straight-line code.
__device__ __forceinline__ rd_src_t select_src(const rd_src_t* src, int si) {
    rd_src_t s;
    s.ptr = si ? src[1].ptr : src[0].ptr;
    s.ptr2 = si ? src[1].ptr2 : src[0].ptr2;
    s.scale = si ? src[1].scale : src[0].scale;
    s.shift = si ? src[1].shift : src[0].shift;
    s.q = si ? src[1].q : src[0].q;
    s.out = si ? src[1].out : src[0].out;
    s.mode = si ? src[1].mode : src[0].mode;
    s.C = si ? src[1].C : src[0].C;
    s.slope = si ? src[1].slope : src[0].slope;
    s.n_off = si ? src[1].n_off : src[0].n_off;
    s.g_fixed = si ? src[1].g_fixed : src[0].g_fixed;
    s.fin_flags = 0;
    s.fin = nullptr;
    return s;
}

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename T, int MODE, int BATCH, int STRIDE, bool BF, typename MapFn, typename StoreFn>
__device__ __forceinline__ void tile_fill_mode(const rd_src_t& s, const SlotCtx<T>& k, int n, int H, int W, int tid, int total,
                                               MapFn map, StoreFn store) {
    constexpr int S = Slot<T>::N;
    constexpr int NQ = (MODE == RD_SRC_POOL || MODE == RD_SRC_UP) ? 4 : (MODE == RD_SRC_BNBWD ? 2 : 1);
    const int C = s.C;
    const T* base = reinterpret_cast<const T*>(s.ptr) + k.c;
    const T* base2 = reinterpret_cast<const T*>(s.ptr2) + k.c;
    const int nn = n + s.n_off;
    for (int idx0 = tid; idx0 < total; idx0 += STRIDE * BATCH) {
        // Phase A is branch-free on purpose: a load inside `if (inside image)` makes the compiler wait for it at
        // the join, i.e. one exposed memory latency per item.  Out-of-range items load a clamped (valid) address
        // and are zeroed in phase B.
        uint4 raw[BATCH][NQ];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int idx = min(idx0 + b * STRIDE, total - 1);
            int y = 0, x = 0;
            const bool in0 = map(idx, y, x);
            y = min(max(y, 0), H - 1);
            x = min(max(x, 0), W - 1);
            if constexpr (!BF) {
                // small images (25x25, 50x50 under 8x32 tiles): a large share of the items is outside the image;
                // there the skipped loads are worth more than the exposed latency
#pragma unroll
                for (int q = 0; q < NQ; ++q) raw[b][q] = make_uint4(0, 0, 0, 0);
                if (!(in0 && idx0 + b * STRIDE < total)) continue;
            }
            if constexpr (MODE == RD_SRC_RAW || MODE == RD_SRC_AFF || MODE == RD_SRC_AFFACT) {
                raw[b][0] = ld16(base + ((size_t)(nn * H + y) * W + x) * C);
            } else if constexpr (MODE == RD_SRC_BNBWD) {
                const size_t off = ((size_t)(nn * H + y) * W + x) * C;
                raw[b][0] = ld16(base + off);
                raw[b][1] = ld16(base2 + off);
            } else if constexpr (MODE == RD_SRC_POOL) {
                const int Ws = 2 * W;
                const T* p00 = base + ((size_t)(nn * 2 * H + 2 * y) * Ws + 2 * x) * C;
                raw[b][0] = ld16(p00);
                raw[b][1] = ld16(p00 + C);
                raw[b][2] = ld16(p00 + (size_t)Ws * C);
                raw[b][3] = ld16(p00 + (size_t)Ws * C + C);
            } else {  // UP
                const int Hs = H >> 1, Ws = W >> 1;
                int y0, y1, x0, x1;
                float ly, lx;
                up2_coord(y, Hs, y0, y1, ly);
                up2_coord(x, Ws, x0, x1, lx);
                const T* pn = base + (size_t)nn * Hs * Ws * C;
                raw[b][0] = ld16(pn + ((size_t)y0 * Ws + x0) * C);
                raw[b][1] = ld16(pn + ((size_t)y0 * Ws + x1) * C);
                raw[b][2] = ld16(pn + ((size_t)y1 * Ws + x0) * C);
                raw[b][3] = ld16(pn + ((size_t)y1 * Ws + x1) * C);
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int idx = idx0 + b * STRIDE;
            if (idx >= total) continue;
            int y = 0, x = 0;
            const bool in = map(idx, y, x);
            float v[S];
#pragma unroll
            for (int e = 0; e < S; ++e) v[e] = 0.f;
            if (in) {
                if constexpr (MODE == RD_SRC_RAW) {
                    store(idx, raw[b][0]);
                    continue;
                } else if constexpr (MODE == RD_SRC_AFF) {
                    Slot<T>::unpack(raw[b][0], v);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = v[e] * k.sc[e] + k.sh[e];
                } else if constexpr (MODE == RD_SRC_AFFACT) {
                    Slot<T>::unpack(raw[b][0], v);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = act_fn(v[e] * k.sc[e] + k.sh[e], s.slope);
                } else if constexpr (MODE == RD_SRC_BNBWD) {
                    float zz[S];
                    Slot<T>::unpack(raw[b][0], v);
                    Slot<T>::unpack(raw[b][1], zz);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = bn_bwd_value(k.sc[e], v[e], k.q[e], zz[e], k.sh[e]);
                } else if constexpr (MODE == RD_SRC_POOL) {
                    float t[S];
                    Slot<T>::unpack(raw[b][0], v);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = act_fn(v[e] * k.sc[e] + k.sh[e], s.slope);
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        Slot<T>::unpack(raw[b][j], t);
#pragma unroll
                        for (int e = 0; e < S; ++e) v[e] = fmaxf(v[e], act_fn(t[e] * k.sc[e] + k.sh[e], s.slope));
                    }
                } else {  // UP
                    const int Hs = H >> 1, Ws = W >> 1;
                    int y0, y1, x0, x1;
                    float ly, lx;
                    up2_coord(y, Hs, y0, y1, ly);
                    up2_coord(x, Ws, x0, x1, lx);
                    float t00[S], t01[S], t10[S], t11[S];
                    Slot<T>::unpack(raw[b][0], t00);
                    Slot<T>::unpack(raw[b][1], t01);
                    Slot<T>::unpack(raw[b][2], t10);
                    Slot<T>::unpack(raw[b][3], t11);
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        const float top = t00[e] + lx * (t01[e] - t00[e]);
                        const float bot = t10[e] + lx * (t11[e] - t10[e]);
                        const float u = top + ly * (bot - top);
                        v[e] = act_fn(u * k.sc[e] + k.sh[e], s.slope);
                    }
                }
            }
            store(idx, Slot<T>::pack(v));
        }
    }
}

// fills `total` items (item idx -> (pixel, this thread's channel slot)); map(idx, y, x) gives the conv-frame
// pixel and whether it lies inside the image; store(idx, u) writes the 16-byte slot to LDS.
template <typename T, int STRIDE = 256, bool BF = true, typename MapFn, typename StoreFn>
__device__ __forceinline__ void tile_fill(const rd_src_t* src, const SlotCtx<T>& k, int n, int H, int W, int tid, int total,
                                          MapFn map, StoreFn store) {
    constexpr int S = Slot<T>::N;
    if (k.si < 0) {
        for (int idx = tid; idx < total; idx += STRIDE) store(idx, make_uint4(0, 0, 0, 0));
        return;
    }
    const rd_src_t s = select_src(src, k.si);
    const bool fast = (s.C % S) == 0 && (s.C - k.c) >= S;
    if (!fast && s.mode == RD_SRC_RAW && s.C <= 4 && k.c == 0) {
        // narrow raw tensors (3-channel image, 2/3-class dlogits): element loads, still loads-first
        const T* base = reinterpret_cast<const T*>(s.ptr);
        const int C = s.C, nn = n + s.n_off;
        constexpr int NB_ = 6;
        for (int idx0 = tid; idx0 < total; idx0 += STRIDE * NB_) {
            T e[NB_][4];
#pragma unroll
            for (int b = 0; b < NB_; ++b) {
                const int idx = idx0 + b * STRIDE;
                int y = 0, x = 0;
                const bool in = idx < total && map(idx, y, x);
#pragma unroll
                for (int j = 0; j < 4; ++j) e[b][j] = from_f<T>(0.f);
                if (in) {
                    const T* pp = base + ((size_t)(nn * H + y) * W + x) * C;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < C) e[b][j] = pp[j];
                }
            }
#pragma unroll
            for (int b = 0; b < NB_; ++b) {
                const int idx = idx0 + b * STRIDE;
                if (idx >= total) continue;
                float v[S];
#pragma unroll
                for (int j = 0; j < S; ++j) v[j] = j < 4 ? to_f<T>(e[b][j]) : 0.f;
                store(idx, Slot<T>::pack(v));
            }
        }
        return;
    }
    if (!fast) {                                           // other odd channel counts: generic per-item path
        for (int idx = tid; idx < total; idx += STRIDE) {
            int y = 0, x = 0;
            float v[S];
#pragma unroll
            for (int e = 0; e < S; ++e) v[e] = 0.f;
            if (map(idx, y, x)) load_slot<T>(s, k, n, y, x, H, W, v);
            store(idx, Slot<T>::pack(v));
        }
        return;
    }
    switch (s.mode) {
    case RD_SRC_RAW: tile_fill_mode<T, RD_SRC_RAW, 6, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    case RD_SRC_AFF: tile_fill_mode<T, RD_SRC_AFF, 6, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    case RD_SRC_AFFACT: tile_fill_mode<T, RD_SRC_AFFACT, 6, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    case RD_SRC_BNBWD: tile_fill_mode<T, RD_SRC_BNBWD, 3, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    case RD_SRC_POOL: tile_fill_mode<T, RD_SRC_POOL, 2, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    default: tile_fill_mode<T, RD_SRC_UP, 2, STRIDE, BF>(s, k, n, H, W, tid, total, map, store); break;
    }
}

__device__ __forceinline__ GroupMap make_gm(const int32_t* gstart, int G) {
    GroupMap gm;
    gm.G = G;
#pragma unroll
    for (int i = 0; i <= RD_MAX_GROUPS; ++i) gm.gs[i] = gstart[i];
    return gm;
}

// XCD-aware block order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own
// L2), so neighbouring tiles -- which share halo rows / columns, and the output-channel blocks of one pixel tile, which
// share the whole input tile -- would land on 8 different L2s and each fetch the shared data from the fabric again.
// Remap the linear id so that XCD k works on the k-th contiguous eighth of the grid (x = pixel tile fastest).
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = lin & 7, idx = lin >> 3, q = total >> 3, r = total & 7;
    const int l2 = xcd * q + min(xcd, r) + idx;
    bx = l2 % gx;
    by = (l2 / gx) % gy;
    bz = l2 / (gx * gy);
}

// ------------------------------------------------------------------------------------ MFMA atoms
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one 64-byte chunk = 32 channels = 2 k-steps of 16; lane-half h owns 8 channels per k-step
    static __device__ __forceinline__ void chunk(const uint4* a_rec, int a_sw, const uint4* b_rec, int b_sw, int h,
                                                 f32x16& acc, int nks = 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks >= nks) break;                          // wave-uniform: <=16 input channels need one k-step
            const int slot = ks * 2 + h;
            uint4 au = a_rec[slot ^ a_sw];
            uint4 bu = b_rec[slot ^ b_sw];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), __builtin_bit_cast(bf16x8, bu),
                                                          acc, 0, 0, 0);
        }
    }
};
template <> struct Mma<float> {
    // one 64-byte chunk = 16 channels; lane-half h owns channels 8h..8h+7 (k order is a permutation,
    // identical for A and B); 8 x v_mfma_f32_32x32x2_f32
    static __device__ __forceinline__ void chunk(const uint4* a_rec, int a_sw, const uint4* b_rec, int b_sw, int h,
                                                 f32x16& acc, int nks = 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int slot = 2 * h + q;
            uint4 au = a_rec[slot ^ a_sw];
            uint4 bu = b_rec[slot ^ b_sw];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au.x), __uint_as_float(bu.x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au.y), __uint_as_float(bu.y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au.z), __uint_as_float(bu.z), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(au.w), __uint_as_float(bu.w), acc, 0, 0, 0);
        }
    }
};

// ------------------------------------------------------------------------------------ gradient epilogue
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float* v, int cvalid, bool vec_ok) {
    constexpr int S = Slot<T>::N;
    if (vec_ok && cvalid >= S) {
        *reinterpret_cast<uint4*>(p) = Slot<T>::pack(v);
    } else {
#pragma unroll
        for (int e = 0; e < S; ++e)
            if (e < cvalid) p[e] = from_f<T>(v[e]);
    }
}

// S channels (one slot) of the gradient w.r.t. a conv input pixel -> gradient w.r.t. the producer's BN
// output: activation mask, max-pool scatter or upsample-side mask; b1 += g, b2 += g*z per channel.
template <typename T, int KMASK = 7>
__device__ __forceinline__ void grad_item(const rd_dst_t& d, int g_img, int n, int y, int x, int H, int W, int cd,
                                          const float* da, const float* sc, const float* sh, float* b1, float* b2) {
    constexpr int S = Slot<T>::N;
    T* gp = reinterpret_cast<T*>(d.g);
    const T* zp = reinterpret_cast<const T*>(d.z);
    const int Cd = d.Cd;
    const int cvalid = Cd - cd;
    const bool vec_ok = (Cd % S) == 0;
    n += d.n_off;
    if ((KMASK & 1) && d.kind == RD_DST_PLAIN) {
        const size_t idx = ((size_t)(n * H + y) * W + x) * Cd + cd;
        float z[S], gw[S];
#pragma unroll
        for (int e = 0; e < S; ++e) z[e] = 0.f;
        if (zp) load_vec<T>(zp + idx, cvalid, vec_ok, z);
        if (d.accumulate) load_vec<T>(gp + idx, cvalid, vec_ok, gw);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const float m = (d.act && zp) ? act_grad(z[e] * sc[e] + sh[e], d.slope) : 1.f;
            const float gn = da[e] * m;
            b1[e] += gn;
            b2[e] += gn * z[e];
            gw[e] = d.accumulate ? gw[e] + gn : gn;
        }
        store_vec<T>(gp + idx, gw, cvalid, vec_ok);
    } else if ((KMASK & 2) && d.kind == RD_DST_POOL) {
        const int Hd = 2 * H, Wd = 2 * W;
        float zz[4][S], best[S];
        int arg[S];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = ((size_t)(n * Hd + 2 * y + (k >> 1)) * Wd + 2 * x + (k & 1)) * Cd + cd;
            load_vec<T>(zp + idx, cvalid, vec_ok, zz[k]);
#pragma unroll
            for (int e = 0; e < S; ++e) {
                const float a = act_fn(zz[k][e] * sc[e] + sh[e], d.slope);
                if (k == 0 || a > best[e]) { best[e] = a; arg[e] = k; }      // first max wins (ATen max_pool2d)
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = ((size_t)(n * Hd + 2 * y + (k >> 1)) * Wd + 2 * x + (k & 1)) * Cd + cd;
            float gw[S];
            if (d.accumulate) load_vec<T>(gp + idx, cvalid, vec_ok, gw);
#pragma unroll
            for (int e = 0; e < S; ++e) {
                float gn = 0.f;
                if (arg[e] == k) {
                    gn = da[e] * (d.act ? act_grad(zz[k][e] * sc[e] + sh[e], d.slope) : 1.f);
                    b1[e] += gn;
                    b2[e] += gn * zz[k][e];
                }
                gw[e] = d.accumulate ? gw[e] + gn : gn;
            }
            store_vec<T>(gp + idx, gw, cvalid, vec_ok);
        }
    } else if ((KMASK & 4) && d.kind == RD_DST_UPY) {
        const int Hs = H >> 1, Ws = W >> 1;
        int yy0, yy1, xx0, xx1;
        float ly, lx;
        up2_coord(y, Hs, yy0, yy1, ly);
        up2_coord(x, Ws, xx0, xx1, lx);
        float t00[S], t01[S], t10[S], t11[S], gw[S];
        load_vec<T>(zp + ((size_t)(n * Hs + yy0) * Ws + xx0) * Cd + cd, cvalid, vec_ok, t00);
        load_vec<T>(zp + ((size_t)(n * Hs + yy0) * Ws + xx1) * Cd + cd, cvalid, vec_ok, t01);
        load_vec<T>(zp + ((size_t)(n * Hs + yy1) * Ws + xx0) * Cd + cd, cvalid, vec_ok, t10);
        load_vec<T>(zp + ((size_t)(n * Hs + yy1) * Ws + xx1) * Cd + cd, cvalid, vec_ok, t11);
        const size_t idx = ((size_t)(n * H + y) * W + x) * Cd + cd;
        if (d.accumulate) load_vec<T>(gp + idx, cvalid, vec_ok, gw);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const float top = t00[e] + lx * (t01[e] - t00[e]), bot = t10[e] + lx * (t11[e] - t10[e]);
            const float u = top + ly * (bot - top);
            const float m = d.act ? act_grad(u * sc[e] + sh[e], d.slope) : 1.f;
            const float gn = da[e] * m;
            b1[e] += gn;
            b2[e] += gn * u;
            gw[e] = d.accumulate ? gw[e] + gn : gn;
        }
        store_vec<T>(gp + idx, gw, cvalid, vec_ok);
    }
}

// plain destination, full 16-byte slots: g = da * act'(z*sc+sh) (+ old g); b1 += g, b2 += g*z
template <typename T>
__device__ __forceinline__ void grad_plain(const rd_dst_t& d, int n, int y, int x, int H, int W, int cd, const float* da,
                                           const float* sc, const float* sh, float* b1, float* b2) {
    constexpr int S = Slot<T>::N;
    T* gp = reinterpret_cast<T*>(d.g);
    const T* zp = reinterpret_cast<const T*>(d.z);
    const size_t idx = ((size_t)((n + d.n_off) * H + y) * W + x) * d.Cd + cd;
    float z[S], gw[S];
    uint4 zu = make_uint4(0, 0, 0, 0), gu = make_uint4(0, 0, 0, 0);
    if (zp) zu = ld16(zp + idx);
    if (d.accumulate) gu = ld16(gp + idx);
    Slot<T>::unpack(zu, z);
    Slot<T>::unpack(gu, gw);
    const float lo = (d.act && zp) ? d.slope : 1.f;         // one select per element, no branch (see grad_plain_finish)
#pragma unroll
    for (int e = 0; e < S; ++e) {
        const float m = (z[e] * sc[e] + sh[e]) > 0.f ? 1.f : lo;
        const float gn = da[e] * m;
        b1[e] += gn;
        b2[e] += gn * z[e];
        gw[e] += gn;
    }
    *reinterpret_cast<uint4*>(gp + idx) = Slot<T>::pack(gw);
}

// grad_plain in two halves, so that a kernel can REQUEST the producer tensor (and the old gradient) of all its vectors before it uses
// the first one.  A loop of grad_plain calls runs as load, wait, compute, store, wait (the compiler must assume that the store
// aliases the next load, and stores count in vmcnt): two dependent memory round trips per vector.  Lanes without a live destination
// read `dummy` (any mapped 16 bytes), so that no load is conditional.
struct GradPlainReq {
    uint4 zu, gu;
    size_t idx;
    bool ok;
};
template <typename T>
__device__ __forceinline__ void grad_plain_issue(GradPlainReq& q, const rd_dst_t& d, bool ok, int n, int y, int x, int H, int W, int cd, const void* dummy) {
    q.ok = ok;
    q.idx = ((size_t)((n + d.n_off) * H + y) * W + x) * d.Cd + cd;
    const T* dm = reinterpret_cast<const T*>(dummy);
    q.zu = ld16((ok && d.z) ? reinterpret_cast<const T*>(d.z) + q.idx : dm);
    q.gu = ld16((ok && d.accumulate) ? reinterpret_cast<const T*>(d.g) + q.idx : dm);
}
template <typename T>
__device__ __forceinline__ void grad_plain_finish(const GradPlainReq& q, const rd_dst_t& d, const float* da, const float* sc, const float* sh,
                                                  float* b1, float* b2) {
    constexpr int S = Slot<T>::N;
    if (!q.ok) return;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    float z[S], gw[S];
    Slot<T>::unpack(d.z ? q.zu : zero4, z);
    Slot<T>::unpack(d.accumulate ? q.gu : zero4, gw);
    const float lo = (d.act && d.z) ? d.slope : 1.f;        // the factor where bn(z) <= 0; 1 for a destination without a mask
#pragma unroll
    for (int e = 0; e < S; ++e) {
        const float m = (z[e] * sc[e] + sh[e]) > 0.f ? 1.f : lo;
        const float gn = da[e] * m;
        b1[e] += gn;
        b2[e] += gn * z[e];
        gw[e] += gn;
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(d.g) + q.idx) = Slot<T>::pack(gw);
}

// The regroup of an accumulator between the two half-waves: r0 = (a of lanes 0-31 | b of lanes 0-31 in lanes 32-63), r1 = (a of lanes 32-63
// in lanes 0-31 | b of lanes 32-63) -- one v_permlane32_swap.  h = lane / 32.
// HISTORY (rounds 5-6).  Round 5 found wrong pixels in conv_small_fwd_kernel under GPU sharing (whole 16-lane groups holding exactly the
// bias), blamed this instruction and replaced it by shuffles in that kernel; round 6 replaced it everywhere -- and then found the real
// cause (profiles/r06_pk_opsel_erratum.txt): not the swap, but the PACKED fp32 add the SLP vectorizer formed right behind it for the bias
// (`v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`).  A packed fp32 instruction whose op_sel makes the low result read the HIGH dword of
// a source reads 0 for it while a wave of another kernel executes MFMAs on the same SIMD (scripts/probe/pk_canary.hip beside
// mfma_spin.hip: 0 alone, 5.7e8 wrong products in 8 s beside an MFMA loop) -- bias + 0 = the bias.  The shuffle form merely changed the
// register allocation so that the vectorizer no longer paired the adds.  Evidence for the instruction: the stand-alone probe (7.9e10 swaps,
// profiles/r06_swap_probe.txt) and round 5's own reproducer under three builds: swap + vectorizer 578-610 of 2 500 repetitions wrong in each
// of three processes; swap WITHOUT the vectorizer 0 of 7 500; shuffles without 0 of 7 500 -- and the swap is 1.5 % of the step faster
// (4.05 vs 4.11 ms).  The library is built with -fno-slp-vectorize -fno-vectorize and contains no packed fp32 instruction
// (tests/test_cpu_host.py disassembles it).  -DRD_NO_PERMLANE32_SWAP builds the exchange form (one 32-lane ds_bpermute per pair).
struct HalfSwap { unsigned r0, r1; };
__device__ __forceinline__ HalfSwap rd_half_swap(unsigned a, unsigned b, int h) {
    HalfSwap r;
#ifndef RD_NO_PERMLANE32_SWAP
    const auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    r.r0 = s[0];
    r.r1 = s[1];
#else
    const unsigned got = __shfl_xor(h == 0 ? b : a, 32, 64);   // the lower half-wave needs its partner's a and gives its b; the upper the other way round
    r.r0 = h == 0 ? a : got;
    r.r1 = h == 0 ? got : b;
#endif
    return r;
}

// ------------------------------------------------------------------------------------ half-wave sums of many values
// The BatchNorm sums of a workgroup: every lane holds NTOT (32 or 64) partial sums -- one pixel column, all channels -- and each has
// to be summed over the 32 lanes of its half-wave.  A butterfly per VALUE (five __shfl_xor steps each, the form all kernels had) is
// 5 x NTOT ds_bpermute round trips through the LDS crossbar, and written per channel with the LDS atomic behind it they run as
// NTOT/2 dependent chains: 17 000 cycles (8 us) per workgroup (scripts/pf_trace.py), more than a third of a conv_pf_kernel
// workgroup's life.  Here every step HALVES the set instead: lanes whose bit `STEP` is clear keep the lower half of the values and
// hand the upper half to their partner, and the other way round, so that NTOT - NTOT/32 exchanges do the whole job, the first two
// steps (48 of 62 exchanges) as DPP quad permutes inside the VALU.  Afterwards lane li of each half-wave holds, in r[0 .. NTOT/32),
// the complete sums of the values with original index half_wave_sum_index(li) * (NTOT / 32) + j.
// The order of the fp32 additions is fixed, so the result is the same from run to run.
template <int MASK>
__device__ __forceinline__ float lane_xor_f(float v) {
    if constexpr (MASK == 1) {
        const int q = __builtin_bit_cast(int, v);
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(q, q, 0xB1, 0xF, 0xF, true));     // quad_perm [1, 0, 3, 2]
    } else if constexpr (MASK == 2) {
        const int q = __builtin_bit_cast(int, v);
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(q, q, 0x4E, 0xF, 0xF, true));     // quad_perm [2, 3, 0, 1]
    } else {
        return __shfl_xor(v, MASK, 64);
    }
}
template <int NTOT, int N = NTOT, int STEP = 0>
__device__ __forceinline__ void half_wave_sums(float (&r)[NTOT], int li) {
    static_assert(NTOT == 32 || NTOT == 64, "32 or 64 values per lane");
    if constexpr (STEP < 5) {
        const bool up = ((li >> STEP) & 1) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const float keep = up ? r[i + N / 2] : r[i];
            const float send = up ? r[i] : r[i + N / 2];
            r[i] = keep + lane_xor_f<(1 << STEP)>(send);
        }
        half_wave_sums<NTOT, N / 2, STEP + 1>(r, li);
    }
}
__device__ __forceinline__ int half_wave_sum_index(int li) { return (int)(__brev((unsigned)li) >> 27); }
// the same exchange over the lane bits MASK, 2 MASK, .. 32 (lanes that differ in the low bits own different channels): N values -> 1
template <int NTOT, int N, int MASK>
__device__ __forceinline__ void lane_group_sums(float (&r)[NTOT], int lane) {
    if constexpr (N > 1) {
        static_assert(MASK <= 32, "more values than lanes to spread them over");
        const bool up = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const float keep = up ? r[i + N / 2] : r[i];
            const float send = up ? r[i] : r[i + N / 2];
            r[i] = keep + lane_xor_f<MASK>(send);
        }
        lane_group_sums<NTOT, N / 2, MASK * 2>(r, lane);
    }
}
// the common case: NV vectors of S channels per lane (NV * S = 16; channel of (v, e) = 2 S v + S h + e), both statistics -> 32 values,
// one LDS atomic per lane into the workgroup's fp64 sums s_red[channel][2]
template <int S, int NV>
__device__ __forceinline__ void flush_half_wave_sums16(double* s_red, const float (&sa)[NV][S], const float (&sb)[NV][S], int li, int h) {
    static_assert(NV * S == 16, "16 channels per lane");
    float r[32];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < S; ++e) {
            r[(v * S + e) * 2 + 0] = sa[v][e];
            r[(v * S + e) * 2 + 1] = sb[v][e];
        }
    half_wave_sums<32>(r, li);
    const int idx = half_wave_sum_index(li), c = idx >> 1;
    const int ch = 2 * S * (c / S) + S * h + (c % S);
    atomicAdd(&s_red[ch * 2 + (idx & 1)], (double)r[0]);
}

// The same sums WITHOUT atomics: wave `wave` leaves its 64 values (channel, statistic) in s_part[wave][64]; the caller adds the waves in
// index order behind a barrier.  LDS atomics arrive in a timing-dependent order, and an fp64 sum is only independent of the order while
// the terms span less than 2^29 in magnitude -- per-wave sums over a few rows do not always (scripts/conv_repeat_stress.py).
template <int S, int NV>
__device__ __forceinline__ void store_half_wave_sums16(double* s_part, int wave, const float (&sa)[NV][S], const float (&sb)[NV][S], int li, int h) {
    static_assert(NV * S == 16, "16 channels per lane");
    float r[32];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < S; ++e) {
            r[(v * S + e) * 2 + 0] = sa[v][e];
            r[(v * S + e) * 2 + 1] = sb[v][e];
        }
    half_wave_sums<32>(r, li);
    const int idx = half_wave_sum_index(li), c = idx >> 1;
    const int ch = 2 * S * (c / S) + S * h + (c % S);
    s_part[wave * 64 + ch * 2 + (idx & 1)] = (double)r[0];
}

// sum b1/b2 over the lanes of a wave that own the same channel slot (lane % SL), then one LDS atomic per
// wave and channel instead of one per thread (64-way same-address contention otherwise)
template <int S, int SL>
// The per-workgroup sums of the BatchNorm statistics are accumulated in FP64 in LDS (and in fp64 slots in global memory): the waves of a
// workgroup arrive in a timing-dependent order, an fp32 accumulator would round differently from run to run, and every statistic is
// amplified through the remaining layers (DESIGN.md Numerics).  A sum of fp32 values in fp64 is exact as long as the terms span fewer
// than 2^(53-24) in magnitude and count, i.e. independent of the order: the step is reproducible run to run (scripts/repro_check.py).
__device__ __forceinline__ void flush_bstats(double* s_red, int lane, int sl, float* b1, float* b2) {
    // 2 S values per lane, summed over the 64 / SL lanes with the same channel slot by the halving exchange of half_wave_sums (lane
    // bits log2(SL) .. 5: 2 S - 1 exchanges instead of 2 S per step), then ONE LDS atomic per lane
    constexpr int N = 2 * S;
    static_assert(N * SL == 64, "one value per lane after the last step");
    float r[N];
#pragma unroll
    for (int e = 0; e < S; ++e) {
        r[e * 2 + 0] = b1[e];
        r[e * 2 + 1] = b2[e];
    }
    lane_group_sums<N, N, SL>(r, lane);
    constexpr int LOGN = (N == 16) ? 4 : 3;
    const int idx = (int)(__brev((unsigned)(lane / SL)) >> (32 - LOGN));
    atomicAdd(&s_red[(sl * S + (idx >> 1)) * 2 + (idx & 1)], (double)r[0]);
}

// picks dst[0] or dst[1] field by field (lane-varying di): keeps the kernarg struct out of scratch
__device__ __forceinline__ rd_dst_t select_dst(const rd_conv_t& p, int di) {
    rd_dst_t d;
    d.g = di ? p.dst[1].g : p.dst[0].g;
    d.z = di ? p.dst[1].z : p.dst[0].z;
    d.scale = di ? p.dst[1].scale : p.dst[0].scale;
    d.shift = di ? p.dst[1].shift : p.dst[0].shift;
    d.bstats = di ? p.dst[1].bstats : p.dst[0].bstats;
    d.kind = di ? p.dst[1].kind : p.dst[0].kind;
    d.act = di ? p.dst[1].act : p.dst[0].act;
    d.accumulate = di ? p.dst[1].accumulate : p.dst[0].accumulate;
    d.Cd = di ? p.dst[1].Cd : p.dst[0].Cd;
    d.slope = di ? p.dst[1].slope : p.dst[0].slope;
    d.n_off = di ? p.dst[1].n_off : p.dst[0].n_off;
    d.g_fixed = di ? p.dst[1].g_fixed : p.dst[0].g_fixed;
    d.pad_ = 0;
    return d;
}

// ------------------------------------------------------------------------------------ small-channel persistent kernel
// Cin <= one 64-byte chunk and Cout <= 32 (every 400x400 / 200x200 layer of the U-Net, forward and dgrad):
// these layers are HBM-bound, so the kernel is built around keeping loads in flight.  A workgroup walks
// `tiles_per_wg` consecutive 8x32 tiles of one image; the packed weights stay in LDS for all of them; the
// raw 16-byte slots of tile t+1 are requested into registers BEFORE the MFMAs and the epilogue of tile t
// and are transformed / written to LDS afterwards (register double buffering).  BN sums are kept in
// registers across the tiles and flushed with one set of atomics per workgroup.
// per-thread item geometry of the halo tile (constant for the whole kernel: hoisted out of the tile loop)
template <int NIT>
struct ItemGeom {
    short py[NIT], px[NIT];
    int lds[NIT];          // slot index in s_in, -1: item does not exist
};

template <typename T, int NIT, int NQ>
__device__ __forceinline__ void pf_issue(uint4 (&raw)[NIT][NQ], const rd_src_t& s, const SlotCtx<T>& k, const ItemGeom<NIT>& ig, int n,
                                         int H, int W, int yh, int xh, int nit = NIT) {
    // branch-free (clamped addresses): conditional loads would be waited for one by one (see tile_fill_mode);
    // 32-bit element offsets from an image base keep the address math off the 64-bit VALU path
    const int C = s.C;
    const T* base = reinterpret_cast<const T*>(s.ptr) + k.c + (size_t)(n + s.n_off) * H * W * C;
    const T* base2 = reinterpret_cast<const T*>(s.ptr2) + k.c + (size_t)(n + s.n_off) * H * W * C;
    unsigned off[NIT];
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        if (b >= nit) break;                               // wave-uniform: 16-channel layers have half the items
        const int y = min(max(yh + ig.py[b], 0), H - 1), x = min(max(xh + ig.px[b], 0), W - 1);
        off[b] = (unsigned)((y * W + x) * C);
        raw[b][0] = ld16(base + off[b]);
    }
    if constexpr (NQ == 2) {
        if (s.mode == RD_SRC_BNBWD) {
#pragma unroll
            for (int b = 0; b < NIT; ++b) {
                if (b >= nit) break;
                raw[b][1] = ld16(base2 + off[b]);
            }
        }
    }
}

template <typename T, int NIT, int NQ, typename StoreFn>
__device__ __forceinline__ void pf_consume_fn(const uint4 (&raw)[NIT][NQ], const rd_src_t& s, const SlotCtx<T>& k, const ItemGeom<NIT>& ig,
                                              int H, int W, int yh, int xh, StoreFn store, int nit = NIT) {
    constexpr int S = Slot<T>::N;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        if (b >= nit) break;
        if (ig.lds[b] < 0) continue;
        const int y = yh + ig.py[b], x = xh + ig.px[b];
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        uint4 u = in ? raw[b][0] : make_uint4(0, 0, 0, 0);
        if (s.mode != RD_SRC_RAW) {
            float v[S];
            Slot<T>::unpack(raw[b][0], v);
            if (s.mode == RD_SRC_AFF) {
#pragma unroll
                for (int e = 0; e < S; ++e) v[e] = v[e] * k.sc[e] + k.sh[e];
            } else if (s.mode == RD_SRC_AFFACT) {
#pragma unroll
                for (int e = 0; e < S; ++e) v[e] = act_fn(v[e] * k.sc[e] + k.sh[e], s.slope);
            } else if constexpr (NQ == 2) {
                float zz[S];
                Slot<T>::unpack(raw[b][1], zz);
#pragma unroll
                for (int e = 0; e < S; ++e) v[e] = bn_bwd_value(k.sc[e], v[e], k.q[e], zz[e], k.sh[e]);
            }
            u = in ? Slot<T>::pack(v) : make_uint4(0, 0, 0, 0);
        }
        store(ig.lds[b], u);
    }
}

template <typename T, int NIT>
__device__ __forceinline__ void pf_consume(const uint4 (&raw)[NIT][2], const rd_src_t& s, const SlotCtx<T>& k, const ItemGeom<NIT>& ig,
                                           int H, int W, int yh, int xh, uint4* s_in, int nit = NIT) {
    pf_consume_fn<T, NIT, 2>(raw, s, k, ig, H, W, yh, xh, [&](int l, const uint4& u) { s_in[l] = u; }, nit);
}


// ------------------------------------------------------------------------------------------------
// Plain per-pixel sources in ONE branch-free form (the prefetching loaders):
//     v = act( sc*x + q*x2 + sh ; slope )        act(v; 1) = v
//   RAW: sc=1 q=0 sh=0 slope=1 | AFF: slope=1 | AFFACT: slope | BNBWD: x2 = z (ptr2), slope=1
// Loaders instantiated with NQ=2 always fetch x2; a single-operand source then aliases x2 to x with q=0.
// Only whole 16-byte channel slots (C % S == 0).
template <typename T>
struct PlainSrc {
    static constexpr int S = Slot<T>::N;
    const T* p0;
    const T* p1;
    int C, n_off;
    float slope;
    float sc[S], sh[S], q[S];
};

template <typename T>
__device__ __forceinline__ void plain_src_init(PlainSrc<T>& k, const rd_src_t& s, int c) {
    k.p0 = reinterpret_cast<const T*>(s.ptr) + c;
    k.p1 = s.mode == RD_SRC_BNBWD ? reinterpret_cast<const T*>(s.ptr2) + c : k.p0;
    k.C = s.C;
    k.n_off = s.n_off;
    k.slope = s.mode == RD_SRC_AFFACT ? s.slope : 1.f;
}

template <typename T>
__device__ __forceinline__ void plain_src_coef(PlainSrc<T>& k, const rd_src_t& s, int g_img, int c) {
    constexpr int S = Slot<T>::N;
    const int g = s.g_fixed >= 0 ? s.g_fixed : g_img;
    const bool raw = s.mode == RD_SRC_RAW, bwd = s.mode == RD_SRC_BNBWD;
#pragma unroll
    for (int e = 0; e < S; ++e) {
        k.sc[e] = raw ? 1.f : s.scale[g * s.C + c + e];
        k.sh[e] = raw ? 0.f : s.shift[g * s.C + c + e];
        k.q[e] = bwd ? s.q[g * s.C + c + e] : 0.f;
    }
}

template <typename T, int NIT, int NQ>
__device__ __forceinline__ void pfu_issue(uint4 (&raw)[NIT][NQ], const PlainSrc<T>& k, const ItemGeom<NIT>& ig, int n,
                                          int H, int W, int yh, int xh, int nit = NIT) {
    const size_t img = (size_t)(n + k.n_off) * H * W * k.C;
    const T* b0 = k.p0 + img;
    const T* b1 = k.p1 + img;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        if (b >= nit) break;
        const int y = min(max(yh + ig.py[b], 0), H - 1), x = min(max(xh + ig.px[b], 0), W - 1);
        const unsigned off = (unsigned)((y * W + x) * k.C);
        raw[b][0] = ld16(b0 + off);
        if constexpr (NQ == 2) raw[b][1] = ld16(b1 + off);
    }
}

// pfu_issue for a loader whose items and source are fixed for the whole kernel: the element offset of every item inside an image
// ((py W + px) C) is computed ONCE; a request is then the add of the tile's (wave-uniform, scalar) offset, the move of items
// outside the image to the clamped pixel by full-rate 24-bit multiplies, one v_med3 that keeps the address inside the image, and the
// pointer add -- no quarter-rate v_mul_lo_u32 per item as in pfu_issue.
template <typename T, int NIT>
__device__ __forceinline__ void pfu_item_offsets(int (&ioff)[NIT], const PlainSrc<T>& k, const ItemGeom<NIT>& ig, int W) {
#pragma unroll
    for (int b = 0; b < NIT; ++b) ioff[b] = ((int)ig.py[b] * W + (int)ig.px[b]) * k.C;
}
template <typename T, int NIT, int NQ>
__device__ __forceinline__ void pfu_issue_pre(uint4 (&raw)[NIT][NQ], const PlainSrc<T>& k, const ItemGeom<NIT>& ig, const int (&ioff)[NIT], int n,
                                              int H, int W, int yh, int xh, bool ghost, bool border) {
    const size_t img = (size_t)(n + k.n_off) * H * W * k.C;
    const T* b0 = k.p0 + img;
    const T* b1 = k.p1 + img;
    // a ghost tile (past the end of a workgroup's list: requested so that the loop stays branch-free) reads pixel (0, 0) with
    // every item -- one cache line, not a tile's worth of traffic
    const int toff = ghost ? -(1 << 28) : (yh * W + xh) * k.C, last = (H * W - 1) * k.C;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        // items beyond the RIGHT edge move to the last pixel of their row (a full-rate 24-bit multiply of the distance): left where
        // they were, the 28 columns that a 100-wide image's last tile column hangs over the edge fetched the NEXT rows' lines instead
        // of re-reading one -- +9 % FETCH_SIZE on the weight-gradient family (PMC).  The single halo row / column beyond the other
        // edges reads a neighbouring row's pixels (clamping those too costs the 50 x 50 / 100 x 100 layers 5 % of their time).
        int adj = 0;
        if (border) {                                         // wave-uniform: only tiles that hang over the right edge pay for it
            const int x = xh + (int)ig.px[b];
            adj = __mul24(min(x, W - 1) - x, k.C);
        }
        const int off = min(max(toff + ioff[b] + adj, 0), last);
        raw[b][0] = ld16(b0 + off);
        if constexpr (NQ == 2) raw[b][1] = ld16(b1 + off);
        // the loads stay in ITEM order (and two calls in call order): pfu_consume uses them in that order, and the compiler's wait
        // counts are only as good as the worst order on any path -- left alone it issued the prologue's two sets newest-first, so the
        // loop's first use waited for vmcnt(0), i.e. for the set requested one tile ago as well
        __builtin_amdgcn_sched_barrier(0);
    }
}

// NOSKIP: every item has an LDS destination (the caller points dead lanes at a dummy record): no divergent branch per item
// KIND 0: the general form above; 1: no activation (slope 1 known at compile time: the BatchNorm-backward source P g + Q z + R);
// 2: the 16 bytes go to LDS as they are (raw tensors)
template <typename T, int NIT, int NQ, bool NOSKIP = false, int KIND = 0, typename StoreFn>
__device__ __forceinline__ void pfu_consume(const uint4 (&raw)[NIT][NQ], const PlainSrc<T>& k, const ItemGeom<NIT>& ig,
                                            int H, int W, int yh, int xh, StoreFn store, int nit = NIT) {
    constexpr int S = Slot<T>::N;
#pragma unroll
    for (int b = 0; b < NIT; ++b) {
        if (b >= nit) break;
        if constexpr (!NOSKIP) {
            if (ig.lds[b] < 0) continue;
        }
        const int y = yh + ig.py[b], x = xh + ig.px[b];
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        if constexpr (KIND == 2) {
            unsigned keep = in ? 0xffffffffu : 0u;
            asm("" : "+v"(keep));
            uint4 u = raw[b][0];
            u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
            store(ig.lds[b], u);
            continue;
        }
        float v[S];
        Slot<T>::unpack(raw[b][0], v);
        if constexpr (NQ == 2) {
            float zz[S];
            Slot<T>::unpack(raw[b][1], zz);
#pragma unroll
            for (int e = 0; e < S; ++e) {
                const float t = bn_bwd_value(k.sc[e], v[e], k.q[e], zz[e], k.sh[e]);
                v[e] = KIND == 1 ? t : act_fn(t, k.slope);
            }
        } else {
#pragma unroll
            for (int e = 0; e < S; ++e) {
                const float t = k.sc[e] * v[e] + k.sh[e];
                v[e] = KIND == 1 ? t : act_fn(t, k.slope);
            }
        }
        // pixels outside the image become zeros by an AND with a per-lane mask, and the mask is laundered through an empty asm:
        // written as a select, the compiler turns it into a BRANCH around the whole transform (one s_cbranch_execz per item) and,
        // because the loaded registers are then consumed inside a conditional block, waits for them with vmcnt(0) -- i.e. also for
        // the set that was requested for the NEXT tile (ISA of wgrad_ws_kernel's loader; scripts/wg_trace.py)
        unsigned keep = in ? 0xffffffffu : 0u;
        asm("" : "+v"(keep));
        uint4 u = Slot<T>::pack(v);
        u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
        store(ig.lds[b], u);
    }
}

}  // namespace

