// conv.hip -- direct (im2col-free) 3x3 / 1x1 convolution on gfx950 MFMA: forward, dgrad, wgrad.
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
#include "common.h"
#include "../../include/ramdsir.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 32;

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
        for (int e = 0; e < S; ++e) v[e] = k.sc[e] * gz[e] + k.q[e] * zz[e] + k.sh[e];
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
    s.mode = si ? src[1].mode : src[0].mode;
    s.C = si ? src[1].C : src[0].C;
    s.slope = si ? src[1].slope : src[0].slope;
    s.n_off = si ? src[1].n_off : src[0].n_off;
    s.g_fixed = si ? src[1].g_fixed : src[0].g_fixed;
    s.pad_ = 0;
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
                    for (int e = 0; e < S; ++e) v[e] = k.sc[e] * v[e] + k.q[e] * zz[e] + k.sh[e];
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
#pragma unroll
    for (int e = 0; e < S; ++e) {
        const float m = (d.act && zp) ? act_grad(z[e] * sc[e] + sh[e], d.slope) : 1.f;
        const float gn = da[e] * m;
        b1[e] += gn;
        b2[e] += gn * z[e];
        gw[e] += gn;
    }
    *reinterpret_cast<uint4*>(gp + idx) = Slot<T>::pack(gw);
}

// sum b1/b2 over the lanes of a wave that own the same channel slot (lane % SL), then one LDS atomic per
// wave and channel instead of one per thread (64-way same-address contention otherwise)
template <int S, int SL>
__device__ __forceinline__ void flush_bstats(float* s_red, int lane, int sl, float* b1, float* b2) {
#pragma unroll
    for (int e = 0; e < S; ++e) {
#pragma unroll
        for (int o = SL; o < 64; o <<= 1) {
            b1[e] += __shfl_xor(b1[e], o, 64);
            b2[e] += __shfl_xor(b2[e], o, 64);
        }
    }
    if (lane < SL) {
#pragma unroll
        for (int e = 0; e < S; ++e) {
            atomicAdd(&s_red[(sl * S + e) * 2 + 0], b1[e]);
            atomicAdd(&s_red[(sl * S + e) * 2 + 1], b2[e]);
        }
    }
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

// ------------------------------------------------------------------------------------ conv kernel
template <typename T, int TAPS, int NB>
__global__ __launch_bounds__(256, 2) void conv_kernel(const rd_conv_t p) {
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
    const int x0 = (blockIdx.x % tiles_x) * TW, y0 = (blockIdx.x / tiles_x) * TH;
    const int n0 = blockIdx.y * NT;
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int H = p.H, W = p.W;
    const int slot = (blockIdx.x + 7 * blockIdx.z) % RD_STAT_SLOTS;

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
                wr[b] = ld16(wbase + ((size_t)(min(tap, TAPS - 1) * p.CoutPad + n0 + nn) * p.CinPad + c0 + s * S));
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
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int pix = (wave * 2 + mb + kh) * PW + li + kw;
                const uint4* a_rec = s_in + pix * 4;
                const int a_sw = (pix >> 2) & 3;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int nn = nb * 32 + li;
                    Mma<T>::chunk(a_rec, a_sw, s_w + (tap * NT + nn) * 4, (nn >> 2) & 3, h, acc[mb][nb]);
                }
            }
        }
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of the 32x32 MFMA: column (N, channel) = lane&31, row (M, pixel) = (r&3)+8*(r>>2)+4*(lane>>5).
    // Each 32-channel block is staged through LDS as fp32 [256 pixels][32 ch] so that the global side runs
    // on 16-byte slots (coalesced stores; vector reads of z / old gradients in the backward epilogues).
    constexpr int SL = 32 / S;                             // slots per 32 channels
    float* s_out = reinterpret_cast<float*>(smem);         // [TH*TW][32]
    float* s_red = s_out + TH * TW * 32;                   // [32][2]
    T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __syncthreads();
        const int cb = n0 + nb * 32;
        if (tid < 64) s_red[tid] = 0.f;
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
                    if (cok && y < H && x0 + col < W) { s1 += v; s2 += v * v; }
                }
            }
            __syncthreads();
            if (p.emode == 0 && p.stats) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0 && cok) {
                    atomicAdd(&s_red[li * 2 + 0], s1);
                    atomicAdd(&s_red[li * 2 + 1], s2);
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

template <typename T, int NIT>
__device__ __forceinline__ void pf_issue(uint4 (&raw)[NIT][2], const rd_src_t& s, const SlotCtx<T>& k, const ItemGeom<NIT>& ig, int n,
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
    if (s.mode == RD_SRC_BNBWD) {
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            if (b >= nit) break;
            raw[b][1] = ld16(base2 + off[b]);
        }
    }
}

template <typename T, int NIT>
__device__ __forceinline__ void pf_consume(const uint4 (&raw)[NIT][2], const rd_src_t& s, const SlotCtx<T>& k, const ItemGeom<NIT>& ig,
                                           int H, int W, int yh, int xh, uint4* s_in, int nit = NIT) {
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
            } else {
                float zz[S];
                Slot<T>::unpack(raw[b][1], zz);
#pragma unroll
                for (int e = 0; e < S; ++e) v[e] = k.sc[e] * v[e] + k.q[e] * zz[e] + k.sh[e];
            }
            u = in ? Slot<T>::pack(v) : make_uint4(0, 0, 0, 0);
        }
        s_in[ig.lds[b]] = u;
    }
}

// SRCG: sources may need the generic (synchronous) loader: max-pool / upsample / odd channel counts.
// EPI : 0 forward, 1 gradient with plain full-slot destinations only (lean path), 3 plain + upsample-side,
//       4 plain + max-pool, 2 anything.
template <typename T, int TAPS, bool SRCG, int EPI>
__global__ __launch_bounds__(256, 2) void conv_small_kernel(const rd_conv_t p, int tiles_per_wg) {
    constexpr int S = Slot<T>::N;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NPIX = PH * PW, NT = 32;
    constexpr int NIT = (NPIX * 4 + 255) / 256;
    constexpr int NV = 16 / S;                             // output vectors of S contiguous channels per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [NPIX][4]
    uint4* s_w = s_in + NPIX * 4;                          // [TAPS][32][4]
    float* s_red = reinterpret_cast<float*>(s_w + TAPS * NT * 4);   // [32][2]
    float* s_dsc = s_red + 64;                             // [32] producer scale of the gradient destinations
    float* s_dsh = s_dsc + 32;                             // [32] producer shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % RD_STAT_SLOTS;

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
            wr[b] = ld16(wbase + ((size_t)(min(tap, TAPS - 1) * p.CoutPad + nn) * p.CinPad + sw * S));
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT;
            if (idx < WTOT) s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = wr[b];
        }
    }
    if (tid < 64) s_red[tid] = 0.f;
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
    SlotCtx<T> ctx;
    slot_ctx<T>(ctx, p.src, p.nsrc, p.Cin, g, sslot * S);
    const rd_src_t ssrc = select_src(p.src, ctx.si > 0 ? 1 : 0);
    const bool live_slot = ctx.si >= 0;
    bool pre = true;
    if constexpr (SRCG)
        pre = live_slot && (ssrc.C % S) == 0 && (ssrc.C - ctx.c) >= S &&
              (ssrc.mode == RD_SRC_RAW || ssrc.mode == RD_SRC_AFF || ssrc.mode == RD_SRC_AFFACT || ssrc.mode == RD_SRC_BNBWD);
    int nks = 2;
    if constexpr (sizeof(T) == 2) nks = p.Cin <= 16 ? 1 : 2;

    // ---- epilogue: after the (weights x pixels) MFMA a lane owns ONE pixel (column lane&31) and 16 output
    //      channels; they are regrouped into NV vectors of S contiguous channels (bf16: one v_permlane32_swap
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
            if (live_slot) tile_fill<T>(p.src, ctx, n, H, W, tid, NPIX * 4, map, store);
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
                        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                        vec[v][j] = __uint_as_float(r[0]);
                        vec[v][4 + j] = __uint_as_float(r[1]);
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
                        sa[v][e] += o[e];
                        sb[v][e] += o[e] * o[e];
                    }
                    store_vec<T>(out + ((size_t)(n * H + y) * W + x) * p.Cout + cb, o, p.Cout - cb, (p.Cout % S) == 0);
                } else {
                    const int di = cb >= p.c_split ? 1 : 0;
                    const rd_dst_t d = select_dst(p, di);
                    if (d.kind == RD_DST_NONE) continue;
                    const int cd = cb - (di ? p.c_split : 0);
                    float dsc[S], dsh[S];
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        dsc[e] = s_dsc[cb + e];
                        dsh[e] = s_dsh[cb + e];
                    }
                    if constexpr (EPI == 1) {
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
    //      same channels -> xor-reduce over the 32 lanes, one LDS atomic per wave half and channel
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < S; ++e) {
            float a = sa[v][e], b = sb[v][e];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (li == 0 && cbv[v] + e < 32) {
                atomicAdd(&s_red[(cbv[v] + e) * 2 + 0], a);
                atomicAdd(&s_red[(cbv[v] + e) * 2 + 1], b);
            }
        }
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
__global__ __launch_bounds__(256, 2) void conv_small_stage_kernel(const rd_conv_t p, int tiles_per_wg) {
    constexpr int S = Slot<T>::N;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int NT = 32, SL = 32 / S;
    constexpr int TOTAL = PH * PW * 4, NIT = (TOTAL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_in = reinterpret_cast<uint4*>(smem);          // [PH*PW][4]
    uint4* s_w = s_in + PH * PW * 4;                       // [TAPS][32][4]
    float* s_out = reinterpret_cast<float*>(s_w + TAPS * NT * 4);   // [TH*TW][32]
    float* s_red = s_out + TH * TW * 32;                   // [32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % RD_STAT_SLOTS;

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
            wr[b] = ld16(wbase + ((size_t)(min(tap, TAPS - 1) * p.CoutPad + nn) * p.CinPad + sw * S));
        }
#pragma unroll
        for (int b = 0; b < WIT; ++b) {
            const int idx = tid + b * 256;
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT;
            if (idx < WTOT) s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = wr[b];
        }
    }
    if (tid < 64) s_red[tid] = 0.f;

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
                    if (interior || (y < H && x0 + col < W)) { s1 += v; s2 += v * v; }
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
                atomicAdd(&s_red[li * 2 + 0], s1);
                atomicAdd(&s_red[li * 2 + 1], s2);
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
                       (REG_EPI ? (size_t)(64 + 64) * sizeof(float) : (size_t)(TH * TW * 32 + 64) * sizeof(float));
    const int ntiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    static int tpw_env = -1;
    if (tpw_env < 0) { const char* e = getenv("RD_TPW"); tpw_env = e ? atoi(e) : 0; }
    int tpw = tpw_env > 0 ? tpw_env : 4;
    while (tpw_env <= 0 && tpw > 1 && (long)((ntiles + tpw - 1) / tpw) * p.N < 1536) tpw >>= 1;     // keep >= ~3 workgroups per CU-slot
    dim3 grid((ntiles + tpw - 1) / tpw, 1, p.N);
    static bool attr_set = false;
    if constexpr (REG_EPI) {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<T, TAPS, SRCG, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv_small_kernel<T, TAPS, SRCG, EPI>), grid, dim3(256), lds, st, p, tpw);
    } else {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_stage_kernel<T, TAPS, SRCG, EPI>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv_small_stage_kernel<T, TAPS, SRCG, EPI>), grid, dim3(256), lds, st, p, tpw);
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

template <typename T, int TAPS, int NB>
int launch_conv(const rd_conv_t& p, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    size_t lds = (size_t)(PH * PW * 4 + TAPS * NB * 32 * 4) * sizeof(uint4);
    const size_t lds_epi = (size_t)(TH * TW * 32 + 64) * sizeof(float);
    if (lds < lds_epi) lds = lds_epi;
    dim3 grid(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH), p.CoutPad / (NB * 32), p.N);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, TAPS, NB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_kernel<T, TAPS, NB>), grid, dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

template <typename T>
int dispatch_conv(const rd_conv_t& p, hipStream_t st) {
    constexpr int CK = 4 * Slot<T>::N;
    if (p.CinPad == CK && p.CoutPad == 32)                 // one K chunk, one N block: the HBM-bound layers
        return p.taps == 9 ? dispatch_conv_small<T, 9>(p, st) : dispatch_conv_small<T, 1>(p, st);
    const bool nb2 = (p.CoutPad % 64) == 0;
    if (p.taps == 9) return nb2 ? launch_conv<T, 9, 2>(p, st) : launch_conv<T, 9, 1>(p, st);
    return nb2 ? launch_conv<T, 1, 2>(p, st) : launch_conv<T, 1, 1>(p, st);
}

// ------------------------------------------------------------------------------------ wgrad kernel
// dW[tap][n][c] = sum over pixels of dz[p][n] * a[p + tap][c];  M = n (Cout), N = c (Cin), K = pixels.
// A workgroup owns a (MB*32) x (NB*32) x TAPS block of dW and walks pixel tiles with stride gridDim.x;
// its 4 waves are MB*NB output blocks x KS = 4/(MB*NB) pixel-row splits.  Each wave stores its partial
// block; a second kernel reduces the splits in a fixed order (deterministic, no atomics).
template <typename T, int TAPS, int MB, int NB>
__global__ __launch_bounds__(256) void wgrad_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles) {
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

// ------------------------------------------------------------------------------------ bf16 wgrad, transposed LDS tiles
// Same decomposition as wgrad_kernel, but the tiles are stored [channel][pixel] so that the K (pixel) run of
// both MFMA operands is contiguous: the dz fragment is one ds_read_b128, the three horizontal taps of a row
// come from ONE 10-pixel window (ds_read_b128 + ds_read_b32; the odd tap is 4 v_alignbit), instead of
// 38 ds_read_u16 + packing per k-step.  Channel rows are padded to 16*odd bytes mod 256 so the 16-lane
// groups of a b128 read hit 64 distinct banks.  Each wave fills whole channel slots (lanes run over
// pixels), so the transposing ds_write_b16 of a wave land on consecutive pixels of one row.
template <int TAPS, int MB, int NB>
__global__ __launch_bounds__(256) void wgrad_t_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles) {
    typedef bf16_t T;
    constexpr int S = 8;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int PWL = (TAPS == 9) ? 40 : 32;             // LDS row pitch in pixels (16-byte aligned windows)
    constexpr int AP = PH * PWL + 8;                        // elements per channel row: 816 B (3x3) / 528 B (1x1)
    constexpr int ZP = TH * TW + 8;                         // 528 B
    constexpr int CA = NB * 32, CZ = MB * 32;
    constexpr int KS = 4 / (MB * NB);
    constexpr int ROWS = TH / KS;
    constexpr int NSA = CA / S, NSZ = CZ / S;               // channel slots per tile
    constexpr int SPA = (NSA + 3) / 4, SPZ = (NSZ + 3) / 4; // slots per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* s_a = reinterpret_cast<unsigned short*>(smem);     // [CA][AP]
    unsigned short* s_z = s_a + CA * AP;                                // [CZ][ZP]

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

    SlotCtx<T> ctx_a[SPA], ctx_z[SPZ];
    int g_ctx = -1;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const int g = group_of(gm, n);
        if (g != g_ctx) {
#pragma unroll
            for (int q = 0; q < SPA; ++q) slot_ctx<T>(ctx_a[q], p.a, p.na, p.Cin, g, cbase + (wave + 4 * q) * S);
#pragma unroll
            for (int q = 0; q < SPZ; ++q) slot_ctx<T>(ctx_z[q], &p.dz, 1, p.Cout, g, nbase + (wave + 4 * q) * S);
            g_ctx = g;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SPA; ++q) {
            const int sl = wave + 4 * q;
            if (sl < NSA) {
                auto map = [&](int pix, int& y, int& x) -> bool {
                    const int py = pix / PW, px = pix - py * PW;
                    y = y0 - HALO + py;
                    x = x0 - HALO + px;
                    return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                };
                auto store = [&](int pix, const uint4& u) {
                    const int py = pix / PW, px = pix - py * PW;
                    unsigned short* d = s_a + (sl * S) * AP + py * PWL + px;
                    d[0 * AP] = (unsigned short)(u.x & 0xffff); d[1 * AP] = (unsigned short)(u.x >> 16);
                    d[2 * AP] = (unsigned short)(u.y & 0xffff); d[3 * AP] = (unsigned short)(u.y >> 16);
                    d[4 * AP] = (unsigned short)(u.z & 0xffff); d[5 * AP] = (unsigned short)(u.z >> 16);
                    d[6 * AP] = (unsigned short)(u.w & 0xffff); d[7 * AP] = (unsigned short)(u.w >> 16);
                };
                tile_fill<T, 64>(p.a, ctx_a[q], n, H, W, lane, PH * PW, map, store);
            }
        }
#pragma unroll
        for (int q = 0; q < SPZ; ++q) {
            const int sl = wave + 4 * q;
            if (sl < NSZ) {
                auto map = [&](int pix, int& y, int& x) -> bool {
                    const int py = pix / TW, px = pix - py * TW;
                    y = y0 + py;
                    x = x0 + px;
                    return y < H && x < W;
                };
                auto store = [&](int pix, const uint4& u) {
                    unsigned short* d = s_z + (sl * S) * ZP + pix;
                    d[0 * ZP] = (unsigned short)(u.x & 0xffff); d[1 * ZP] = (unsigned short)(u.x >> 16);
                    d[2 * ZP] = (unsigned short)(u.y & 0xffff); d[3 * ZP] = (unsigned short)(u.y >> 16);
                    d[4 * ZP] = (unsigned short)(u.z & 0xffff); d[5 * ZP] = (unsigned short)(u.z >> 16);
                    d[6 * ZP] = (unsigned short)(u.w & 0xffff); d[7 * ZP] = (unsigned short)(u.w >> 16);
                };
                tile_fill<T, 64>(&p.dz, ctx_z[q], n, H, W, lane, TH * TW, map, store);
            }
        }
        __syncthreads();
        const unsigned short* zr = s_z + (mb * 32 + li) * ZP;
        const unsigned short* ar = s_a + (nb * 32 + li) * AP;
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = kq + rr * KS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int px0 = ks * 16 + 8 * h;                  // this lane-half's 8 pixels (k = 8h+e)
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(zr + row * TW + px0));
#pragma unroll
                for (int kh = 0; kh < (TAPS == 9 ? 3 : 1); ++kh) {
                    const unsigned short* wp = ar + (row + kh) * PWL + px0;     // halo coords: input = output + tap
                    const uint4 dq = *reinterpret_cast<const uint4*>(wp);
                    if constexpr (TAPS == 9) {
                        const unsigned d4 = *reinterpret_cast<const unsigned*>(wp + 8);
                        const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(dq.y, dq.x, 16), __builtin_amdgcn_alignbit(dq.z, dq.y, 16),
                                                    __builtin_amdgcn_alignbit(dq.w, dq.z, 16), __builtin_amdgcn_alignbit(d4, dq.w, 16));
                        const uint4 m2 = make_uint4(dq.y, dq.z, dq.w, d4);
                        acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                        acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                        acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
                    } else {
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

// ------------------------------------------------------------------------------------ bf16 wgrad, <= 32 channels
// The HBM-bound layers (16/32 channels at 400x400 / 200x200): v_mfma_f32_16x16x32_bf16 with K = the 32 pixels
// of one tile row, 16x16 channel blocks (no padding of 16-channel layers to 32), 4 accumulator VGPRs per tap
// (36 for a 3x3) instead of 144 -> 4 workgroups per CU keep enough loads in flight.  MB x NB 16-channel
// blocks per workgroup; the 4 waves are MB*NB blocks x KS = 4/(MB*NB) row splits.
template <int TAPS, int MB, int NB>
__global__ __launch_bounds__(256, 3) void wgrad_c16_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles) {
    typedef bf16_t T;
    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    constexpr int S = 8;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr int PWL = (TAPS == 9) ? 40 : 32;
    constexpr int AP = PH * PWL + 8, ZP = TH * TW + 8;
    constexpr int CA = NB * 16, CZ = MB * 16;
    constexpr int KS = 4 / (MB * NB);
    constexpr int ROWS = TH / KS;
    constexpr int NSA = CA / S, NSZ = CZ / S, NJOB = NSA + NSZ;      // fill jobs: one 8-channel slot each
    constexpr int JPW = (NJOB + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* s_a = reinterpret_cast<unsigned short*>(smem);     // [CA][AP]
    unsigned short* s_z = s_a + CA * AP;                                // [CZ][ZP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int kq = wave / (MB * NB), blk = wave % (MB * NB);
    const int mb = blk / NB, nb = blk % NB;
    const int nbase = blockIdx.y * CZ, cbase = blockIdx.z * CA;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);

    f32x4v acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    SlotCtx<T> ctx[JPW];
    int g_ctx = -1;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        const int g = group_of(gm, n);
        if (g != g_ctx) {
#pragma unroll
            for (int q = 0; q < JPW; ++q) {
                const int job = wave + 4 * q;
                if (job < NSA) slot_ctx<T>(ctx[q], p.a, p.na, p.Cin, g, cbase + job * S);
                else if (job < NJOB) slot_ctx<T>(ctx[q], &p.dz, 1, p.Cout, g, nbase + (job - NSA) * S);
            }
            g_ctx = g;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < JPW; ++q) {
            const int job = wave + 4 * q;
            if (job < NSA) {
                auto map = [&](int pix, int& y, int& x) -> bool {
                    const int py = pix / PW, px = pix - py * PW;
                    y = y0 - HALO + py;
                    x = x0 - HALO + px;
                    return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                };
                auto store = [&](int pix, const uint4& u) {
                    const int py = pix / PW, px = pix - py * PW;
                    unsigned short* d = s_a + (job * S) * AP + py * PWL + px;
                    d[0 * AP] = (unsigned short)(u.x & 0xffff); d[1 * AP] = (unsigned short)(u.x >> 16);
                    d[2 * AP] = (unsigned short)(u.y & 0xffff); d[3 * AP] = (unsigned short)(u.y >> 16);
                    d[4 * AP] = (unsigned short)(u.z & 0xffff); d[5 * AP] = (unsigned short)(u.z >> 16);
                    d[6 * AP] = (unsigned short)(u.w & 0xffff); d[7 * AP] = (unsigned short)(u.w >> 16);
                };
                tile_fill<T, 64>(p.a, ctx[q], n, H, W, lane, PH * PW, map, store);
            } else if (job < NJOB) {
                const int sl = job - NSA;
                auto map = [&](int pix, int& y, int& x) -> bool {
                    const int py = pix / TW, px = pix - py * TW;
                    y = y0 + py;
                    x = x0 + px;
                    return y < H && x < W;
                };
                auto store = [&](int pix, const uint4& u) {
                    unsigned short* d = s_z + (sl * S) * ZP + pix;
                    d[0 * ZP] = (unsigned short)(u.x & 0xffff); d[1 * ZP] = (unsigned short)(u.x >> 16);
                    d[2 * ZP] = (unsigned short)(u.y & 0xffff); d[3 * ZP] = (unsigned short)(u.y >> 16);
                    d[4 * ZP] = (unsigned short)(u.z & 0xffff); d[5 * ZP] = (unsigned short)(u.z >> 16);
                    d[6 * ZP] = (unsigned short)(u.w & 0xffff); d[7 * ZP] = (unsigned short)(u.w >> 16);
                };
                tile_fill<T, 64>(&p.dz, ctx[q], n, H, W, lane, TH * TW, map, store);
            }
        }
        __syncthreads();
        const unsigned short* zr = s_z + (mb * 16 + li) * ZP + kg * 8;
        const unsigned short* ar = s_a + (nb * 16 + li) * AP + kg * 8;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int row = kq + rr * KS;
            const bf16x8 afrag = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(zr + row * TW));
#pragma unroll
            for (int kh = 0; kh < (TAPS == 9 ? 3 : 1); ++kh) {
                const unsigned short* wp = ar + (row + kh) * PWL;
                const uint4 dq = *reinterpret_cast<const uint4*>(wp);
                if constexpr (TAPS == 9) {
                    const unsigned d4 = *reinterpret_cast<const unsigned*>(wp + 8);
                    const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(dq.y, dq.x, 16), __builtin_amdgcn_alignbit(dq.z, dq.y, 16),
                                                __builtin_amdgcn_alignbit(dq.w, dq.z, 16), __builtin_amdgcn_alignbit(d4, dq.w, 16));
                    const uint4 m2 = make_uint4(dq.y, dq.z, dq.w, d4);
                    acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                    acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                    acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[0], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (KS > 1) {
        float* s_acc = reinterpret_cast<float*>(smem);     // [(KS-1)][MB*NB][TAPS][4][64]
        __syncthreads();
        if (kq > 0) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[((((kq - 1) * (MB * NB) + blk) * TAPS + tap) * 4 + r) * 64 + lane] = acc[tap][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int k2 = 0; k2 < KS - 1; ++k2)
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tap][r] += s_acc[(((k2 * (MB * NB) + blk) * TAPS + tap) * 4 + r) * 64 + lane];
        }
    }
    if (kq == 0) {
        // D layout of the 16x16 MFMA: column (N, cin) = lane&15, row (M, cout) = 4*(lane>>4) + r
        float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nrow = nbase + mb * 16 + 4 * kg + r;
                const int ccol = cbase + nb * 16 + li;
                out[((size_t)tap * CoutPadW + nrow) * CinPadW + ccol] = acc[tap][r];
            }
    }
}

// block = 32 outputs x 8 split lanes: each thread sums every 8th split, LDS folds the 8 lanes in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* dW, int nsplit, int taps, int Cout,
                                                           int Cin, int CoutPadW, int CinPadW, float beta) {
    __shared__ float s[8][32];
    const int total = taps * Cout * Cin;
    const int o = threadIdx.x & 31, ql = threadIdx.x >> 5;
    const size_t stride = (size_t)taps * CoutPadW * CinPadW;
    for (int base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
        const int i = base + o;
        float acc = 0.f;
        int c = 0, n = 0, tap = 0;
        if (i < total) {
            c = i % Cin; n = (i / Cin) % Cout; tap = i / (Cin * Cout);
            const float* src = partial + ((size_t)tap * CoutPadW + n) * CinPadW + c;
            for (int k = ql; k < nsplit; k += 8) acc += src[k * stride];
        }
        s[ql][o] = acc;
        __syncthreads();
        if (ql == 0 && i < total) {
            const float v = ((s[0][o] + s[1][o]) + (s[2][o] + s[3][o])) + ((s[4][o] + s[5][o]) + (s[6][o] + s[7][o]));
            float* d = dW + ((size_t)n * Cin + c) * taps + tap;
            *d = (beta != 0.f ? beta * *d : 0.f) + v;
        }
        __syncthreads();
    }
}

struct WgradGeom {
    int MB, NB, KS, CoutPadW, CinPadW, gx, total_tiles, nsplit;
    bool c16;
};

template <typename T>
WgradGeom wgrad_geom(const rd_wgrad_t& p) {
    WgradGeom g;
    g.c16 = false;
    // 16-channel blocks only where the kernel stays spill-free at 3 workgroups/CU (one block per workgroup)
    if (sizeof(T) == 2 && p.Cout <= 16 && p.Cin <= 16) {
        g.c16 = true;
        g.MB = p.Cout > 16 ? 2 : 1;
        g.NB = p.Cin > 16 ? 2 : 1;
        g.KS = 4 / (g.MB * g.NB);
        g.CoutPadW = g.MB * 16;
        g.CinPadW = g.NB * 16;
        g.total_tiles = p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
        int gx = 1024;                                      // 4 workgroups per CU
        if (gx > g.total_tiles) gx = g.total_tiles;
        g.gx = gx < 1 ? 1 : gx;
        g.nsplit = g.gx;
        return g;
    }
    const int cout32 = (p.Cout + 31) / 32, cin32 = (p.Cin + 31) / 32;
    // fp32 keeps 32x32 blocks (LDS budget); bf16 uses 64-wide tiles where the layer has them
    g.MB = (sizeof(T) == 2 && cout32 % 2 == 0) ? 2 : 1;
    g.NB = (sizeof(T) == 2 && cin32 % 2 == 0) ? 2 : 1;
    g.KS = 4 / (g.MB * g.NB);
    g.CoutPadW = cout32 * 32;
    g.CinPadW = cin32 * 32;
    g.total_tiles = p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const int pairs = (g.CoutPadW / (g.MB * 32)) * (g.CinPadW / (g.NB * 32));
    // these kernels hold 144 accumulator registers per lane -> one workgroup per CU is resident: launching more
    // workgroups than CUs only multiplies the partial-sum traffic (147 KB per workgroup for a 64x64 tile)
    int gx = (256 + pairs - 1) / pairs;
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
    hipLaunchKernelGGL((wgrad_kernel<T, TAPS, MB, NB>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles);
    return (int)hipGetLastError();
}

template <int TAPS, int MB, int NB>
int launch_wgrad_c16(const rd_wgrad_t& p, const WgradGeom& g, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO;
    constexpr int PWL = (TAPS == 9) ? 40 : 32;
    size_t lds = (size_t)(NB * 16 * (PH * PWL + 8) + MB * 16 * (TH * TW + 8)) * 2;
    const size_t lds_red = (size_t)(4 / (MB * NB) - 1) * (MB * NB) * TAPS * 4 * 64 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    dim3 grid(g.gx, g.CoutPadW / (MB * 16), g.CinPadW / (NB * 16));
    hipLaunchKernelGGL((wgrad_c16_kernel<TAPS, MB, NB>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles);
    return (int)hipGetLastError();
}

template <int TAPS, int MB, int NB>
int launch_wgrad_t(const rd_wgrad_t& p, const WgradGeom& g, hipStream_t st) {
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO;
    constexpr int PWL = (TAPS == 9) ? 40 : 32;
    size_t lds = (size_t)(NB * 32 * (PH * PWL + 8) + MB * 32 * (TH * TW + 8)) * 2;
    const size_t lds_red = (size_t)(4 / (MB * NB) - 1) * (MB * NB) * 16 * 64 * sizeof(float);
    if (lds < lds_red) lds = lds_red;
    dim3 grid(g.gx, g.CoutPadW / (MB * 32), g.CinPadW / (NB * 32));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_t_kernel<TAPS, MB, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_t_kernel<TAPS, MB, NB>), grid, dim3(256), lds, st, p, g.CoutPadW, g.CinPadW, g.total_tiles);
    return (int)hipGetLastError();
}

template <typename T>
int dispatch_wgrad(const rd_wgrad_t& p, hipStream_t st) {
    const WgradGeom g = wgrad_geom<T>(p);
    int e;
    if (g.c16) {
#define RD_WGC(TAPS_)                                                                \
    if (g.MB == 2 && g.NB == 2) e = launch_wgrad_c16<TAPS_, 2, 2>(p, g, st);         \
    else if (g.MB == 2) e = launch_wgrad_c16<TAPS_, 2, 1>(p, g, st);                 \
    else if (g.NB == 2) e = launch_wgrad_c16<TAPS_, 1, 2>(p, g, st);                 \
    else e = launch_wgrad_c16<TAPS_, 1, 1>(p, g, st);
        if (p.taps == 9) { RD_WGC(9) } else { RD_WGC(1) }
#undef RD_WGC
    } else if constexpr (sizeof(T) == 2) {
#define RD_WGT(TAPS_)                                                                \
    if (g.MB == 2 && g.NB == 2) e = launch_wgrad_t<TAPS_, 2, 2>(p, g, st);           \
    else if (g.MB == 2) e = launch_wgrad_t<TAPS_, 2, 1>(p, g, st);                   \
    else if (g.NB == 2) e = launch_wgrad_t<TAPS_, 1, 2>(p, g, st);                   \
    else e = launch_wgrad_t<TAPS_, 1, 1>(p, g, st);
        if (p.taps == 9) { RD_WGT(9) } else { RD_WGT(1) }
#undef RD_WGT
    } else {
#define RD_WG(TAPS_)                                                                 \
    if (g.MB == 2 && g.NB == 2) e = launch_wgrad<T, TAPS_, 2, 2>(p, g, st);          \
    else if (g.MB == 2) e = launch_wgrad<T, TAPS_, 2, 1>(p, g, st);                  \
    else if (g.NB == 2) e = launch_wgrad<T, TAPS_, 1, 2>(p, g, st);                  \
    else e = launch_wgrad<T, TAPS_, 1, 1>(p, g, st);
        if (p.taps == 9) { RD_WG(9) } else { RD_WG(1) }
#undef RD_WG
    }
    if (e) return e;
    const int total = p.taps * p.Cout * p.Cin;
    int blocks = (total + 31) / 32;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.partial, p.dW, g.nsplit, p.taps, p.Cout, p.Cin,
                       g.CoutPadW, g.CinPadW, p.beta);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_weights_kernel(const float* w, T* out, int Cout, int Cin, int taps, int transpose, int RowPad, int ColPad) {
    // forward:   out[tap][n<RowPad(Cout)][c<ColPad(Cin)]   = w[n][c][tap]
    // transpose: out[tap'][c<RowPad(Cin)][n<ColPad(Cout)]  = w[n][c][taps-1-tap']
    const int total = taps * RowPad * ColPad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % ColPad, row = (i / ColPad) % RowPad, tap = i / (ColPad * RowPad);
        float v = 0.f;
        if (!transpose) {
            if (row < Cout && col < Cin) v = w[((size_t)row * Cin + col) * taps + tap];
        } else {
            if (row < Cin && col < Cout) v = w[((size_t)col * Cin + row) * taps + (taps - 1 - tap)];
        }
        out[i] = from_f<T>(v);
    }
}

// all convs of the network in one launch: entry e owns packed elements [start_e, start_{e+1})
template <typename T>
__global__ void pack_weights_batched_kernel(const float* params, T* packed, const rd_pack_entry_t* tab, int n_entries,
                                            int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_entries - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].start <= i) lo = mid; else hi = mid - 1;
        }
        const rd_pack_entry_t e = tab[lo];
        const int j = (int)(i - e.start);
        const int col = j % e.ColPad, row = (j / e.ColPad) % e.RowPad, tap = j / (e.ColPad * e.RowPad);
        const float* w = params + e.src_off;
        float v = 0.f;
        if (!e.transpose) {
            if (row < e.Cout && col < e.Cin) v = w[((size_t)row * e.Cin + col) * e.taps + tap];
        } else {
            if (row < e.Cin && col < e.Cout) v = w[((size_t)col * e.Cin + row) * e.taps + (e.taps - 1 - tap)];
        }
        packed[e.dst_off + j] = from_f<T>(v);
    }
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

extern "C" {

int64_t rd_packed_elems(int Cout, int Cin, int taps, int transpose, int dtype) {
    const int ck = dtype == RD_BF16 ? 32 : 16;
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    return (int64_t)taps * round_up(rows, 32) * round_up(cols, ck);
}

int rd_pack_weights(const float* w_oihw, void* packed, int Cout, int Cin, int taps, int transpose, int dtype, void* stream) {
    const int ck = dtype == RD_BF16 ? 32 : 16;
    const int rows = transpose ? Cin : Cout, cols = transpose ? Cout : Cin;
    const int RowPad = round_up(rows, 32), ColPad = round_up(cols, ck);
    const int total = taps * RowPad * ColPad;
    int blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RD_BF16)
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w_oihw, (bf16_t*)packed, Cout, Cin, taps,
                           transpose, RowPad, ColPad);
    else
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, w_oihw, (float*)packed, Cout, Cin, taps,
                           transpose, RowPad, ColPad);
    return (int)hipGetLastError();
}

int rd_pack_weights_batched(const float* params, void* packed, const rd_pack_entry_t* table_dev, int n_entries, int64_t total,
                            int dtype, void* stream) {
    if (n_entries < 1 || total < 1) return -1;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RD_BF16)
        hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, st, params, (bf16_t*)packed, table_dev,
                           n_entries, total);
    else
        hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3((int)blocks), dim3(256), 0, st, params, (float*)packed, table_dev,
                           n_entries, total);
    return (int)hipGetLastError();
}

int rd_conv(const rd_conv_t* p, int dtype, void* stream) {
    if (!p || (p->taps != 9 && p->taps != 1) || p->G < 1 || p->G > RD_MAX_GROUPS || p->nsrc < 1 || p->nsrc > 2) return -1;
    const int ck = dtype == RD_BF16 ? 32 : 16;
    if (p->CinPad % ck || p->CoutPad % 32 || p->CinPad < p->Cin || p->CoutPad < p->Cout) return -2;
    hipStream_t st = (hipStream_t)stream;
    return dtype == RD_BF16 ? dispatch_conv<bf16_t>(*p, st) : dispatch_conv<float>(*p, st);
}

int64_t rd_wgrad_workspace(const rd_wgrad_t* p, int dtype) {
    const WgradGeom g = dtype == RD_BF16 ? wgrad_geom<bf16_t>(*p) : wgrad_geom<float>(*p);
    return (int64_t)g.nsplit * p->taps * g.CoutPadW * g.CinPadW * (int64_t)sizeof(float);
}

int rd_wgrad(const rd_wgrad_t* p, int dtype, void* stream) {
    if (!p || (p->taps != 9 && p->taps != 1) || p->G < 1 || p->G > RD_MAX_GROUPS || p->na < 1 || p->na > 2) return -1;
    hipStream_t st = (hipStream_t)stream;
    return dtype == RD_BF16 ? dispatch_wgrad<bf16_t>(*p, st) : dispatch_wgrad<float>(*p, st);
}

}  // extern "C"
