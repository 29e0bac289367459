// conv_small_fwd.hip -- forward kernel of the small-channel 3x3 convs (<= 32 channels in and out, bf16): the 400x400 / 200x200
// levels, where a launch moves 80-330 MB for 2-40 GFLOP.
//
// conv_small_kernel (conv_small.hip) runs these as 2 x 256-thread workgroups per CU: the raw vectors of tile t+1 are requested,
// tile t is multiplied and stored, a barrier, tile t+1 is transformed into the ONE LDS tile, a barrier -- 2.5-3.1 TB/s of
// algorithmic traffic where plain streaming kernels reach 5-6 TB/s on this part (scripts/hbm_ceiling.py, scripts/membench).
// What the timing builds of this file showed (XP switches below, scripts/sw_exp.sh; 32 -> 32 channels at 400x400, 8 images, 60 us):
//   * the halo-tile read pattern itself is not the problem: 4 waves per CU with two register sets in flight stream it at 4.4 TB/s
//     of tensor bytes (membench "tilerows"); more sets do not help, more waves are not needed;
//   * a wave is an in-order instruction stream at ~6 cycles per dependent VALU instruction: with four MFMA waves per CU (one per
//     SIMD) the 36 MFMAs of a tile + the LDS reads in front of each of them take 2.2 us, the epilogue 0.6, its stores 0.8 --
//     3 us per tile and CU whatever the loader does (a warp-specialised version of this kernel, loader waves beside MFMA waves
//     as in conv_small_bwd_fused_kernel, measured 65-70 us: its loader waves idle two thirds of the time).
// So this kernel keeps ALL EIGHT waves of a 512-thread workgroup busy with the same work and removes the LDS reads it can:
//   * one workgroup per CU walks the 8 x 32 tiles of its image; wave w owns tile row w: 18 MFMAs (v_mfma_f32_32x32x16_bf16,
//     weights x pixels roles) on 18 pixel fragments read from LDS (3 halo rows x 3 columns x 2 k-steps) -- the WEIGHTS live in
//     registers for the whole launch (18 x 16 B per lane, read once from L2), so LDS traffic per tile halves;
//   * every thread is also a loader: SW_NSET register sets of raw 16-byte vectors are in flight (tiles t+2 ..), the set of tile
//     t+1 is transformed (BatchNorm affine + activation of the producer, bf16 pack) into the other of TWO LDS tile buffers;
//   * register epilogue (permlane regroup, BatchNorm sums, bias, 16-byte NHWC stores), ONE barrier per tile that waits for LDS
//     traffic only (s_waitcnt lgkmcnt(0) + s_barrier: nobody waits for loads in flight);
//   * the loader is BRANCH-FREE per item and per tile: vmcnt counts outstanding loads in order and the compiler can only wait for
//     "all but the N youngest" when N is the same on every path -- one conditional load anywhere in the loop turns every wait into
//     vmcnt(0).  Tiles past the workgroup's range are GHOSTS (every lane reads pixel (0, 0): one cache line; zeros go to the buffer
//     nobody reads any more), items past the halo tile and channel slots past Cin land in a dummy LDS record.
#include <type_traits>
#include "conv_device.h"
#include "conv_dispatch.h"

namespace {

constexpr int SW_PH = TH + 2, SW_PW = TW + 2, SW_NPIX = SW_PH * SW_PW;      // 10 x 34 halo tile
constexpr int SW_BUF_BYTES = SW_NPIX * 64 + 64;                             // + one dummy record (items that do not exist)
constexpr int SW_FIN = 2 * SW_BUF_BYTES + 64 * 8 + 32 * 4;                   // coefficient table of a folded BatchNorm finalize (bn_fin.h)
constexpr int SW_LDS = SW_FIN + rdfin::FIN_LDS_FLOATS * 4;
constexpr int SW_NSET = 3;                                                 // 2: the same step (4.85 ms), 5 for the <=16-channel inputs: 4.91

__device__ uint4 sw_trash[1024];
#ifdef RD_DEBUG_SWITCHES
__device__ int sw_jitter_flag;
__device__ __forceinline__ bool sw_jitter_on() { return *reinterpret_cast<volatile int*>(&sw_jitter_flag) != 0; }
#endif                         // where the stores of lanes without an output pixel go

__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XP: timing experiments, compile-time so that they do not change the code around them (debug build only, RD_SW_EXP; results are
// wrong when set): 1 no MFMAs / fragment reads, 2 no stores, 4 no transform / LDS writes, 16 no epilogue
// NSL: live 16-byte channel slots of the input: 1, 2 or 4.  OUTV: 16-byte output vectors per lane: 2 (Cout 32), 1 (Cout 16), 3 (Cout <= 4:
// the network's output layers, four unconditional scalar stores per lane), 0 (any other Cout, element-wise conditional stores)
// RD_SW_XP_ALL (compile-time, -DRD_SW_XP_ALL=bits: timing builds of EVERY instantiation, scripts/ab_layers_lib.sh; results wrong when set)
#ifndef RD_SW_XP_ALL
#define RD_SW_XP_ALL 0
#endif
#ifndef RD_SW_NWV_DEFAULT
#define RD_SW_NWV_DEFAULT 4
#endif
// NWV: waves per workgroup.  8: one 512-thread workgroup per CU, wave w owns tile row w (inputs of 32 channels: two rows per wave do not fit
// their 72 registers of weights).  4 (inputs of <= 16 channels, the default there): TWO independent 256-thread workgroups per CU, wave w owns
// rows w and w + 4 -- the phases of a tile (requests, transform, products, epilogue) are serial inside a workgroup (one barrier per tile),
// two workgroups interleave theirs without one: 58 -> 45 us for the 16 -> 16 launches at 400 x 400, the step 4.19 -> 4.12 ms.
// (When first measured this shape cost the step its run-to-run repeatability on some boxes; the reason was neither the shape nor the regroup
// of the epilogue but a packed fp32 add with op_sel behind it: conv_device.h rd_half_swap, profiles/r06_pk_opsel_erratum.txt.)
template <int NSL, int OUTV, int XP_, int NWV>
__global__ __launch_bounds__(64 * NWV, 8 / NWV) void conv_small_fwd_kernel(const rd_conv_t p, int tiles_per_wg, const rdfin::FinArg fa) {
    typedef bf16_t T;
    constexpr int XP = XP_ | RD_SW_XP_ALL;
    constexpr int S = 8, NV = 2;
    constexpr int NKS = NSL <= 2 ? 1 : 2;                // k-steps of 16 channels
    constexpr int NSH = NSL == 1 ? 0 : (NSL == 2 ? 1 : 2);
    constexpr int NT = 64 * NWV, RPW = TH / NWV;         // threads; tile rows per wave
    constexpr int NITV = (SW_NPIX * NSL + NT - 1) / NT;  // loader items (halo pixel, live channel slot) per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_inb = smem;                                                      // 2 x [NPIX + 1][4 slots x 16 B]
    double* s_red = reinterpret_cast<double*>(smem + 2 * SW_BUF_BYTES);      // [32][2], fp64 (conv_device.h flush_bstats)
    float* s_bias = reinterpret_cast<float*>(s_red + 64);                    // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;           // wave = tile row
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_first = blockIdx.x * tiles_per_wg;
    const int nt = min(ntiles, t_first + tiles_per_wg) - t_first;
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % rd_stat_nslots(p.stat_slots);

    // ---- prologue: zero the tile buffers (channel slots beyond Cin stay zero), bias table, this lane's weight fragments
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < 2 * SW_BUF_BYTES / 16; i += NT) z4[i] = make_uint4(0, 0, 0, 0);
        if (tid < 32) s_bias[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        if (tid < 64) s_red[tid] = 0.0;
    }
    // A operand of (tap, k-step): output channel li, input channels (2 ks + h) * 8 .. + 8
    uint4 wreg[9][NKS];
    {
        const T* wbase = reinterpret_cast<const T*>(p.w) + (size_t)li * p.CinPad + h * S;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wreg[tap][ks] = ld16(wbase + (size_t)tap * p.CoutPad * p.CinPad + ks * 2 * S);
    }

    // ---- loader constants.  The geometry first (which source and channel this thread's slot is): the BatchNorm coefficients sc / sh are
    //      filled in below, BEHIND the first tile requests -- with a folded finalize (bn_fin.h) they come out of this launch's own prologue,
    //      whose memory round trip then runs beside the tiles'
    const int sslot = tid & (NSL - 1);
    SlotCtx<T> ctx;
    slot_geom<T>(ctx, p.src, p.nsrc, p.Cin, sslot * S);
    const bool live_slot = ctx.si >= 0;
    const rd_src_t ssrc = select_src(p.src, ctx.si > 0 ? 1 : 0);
    const bool rawm = ssrc.mode == RD_SRC_RAW;
    const float slope = ssrc.mode == RD_SRC_AFFACT ? ssrc.slope : 1.f;
    float sc[S], sh[S];
    // wave-uniform transform kind: 0 copy (raw tensors), 1 affine + ReLU (every BatchNorm + ReLU producer), 2 general
    const int kind = __builtin_amdgcn_ballot_w64(!(rawm || !live_slot)) == 0 ? 0
                   : (__builtin_amdgcn_ballot_w64(live_slot && !(ssrc.mode == RD_SRC_AFFACT && ssrc.slope == 0.f)) == 0 ? 1 : 2);
    int iyx[NITV], ilds[NITV];
    unsigned ioff[NITV];                                     // element offset of the item inside a halo tile that lies in the image
    const int C = ssrc.C;
#pragma unroll
    for (int b = 0; b < NITV; ++b) {
        const int pixi = (tid + b * NT) >> NSH;
        const int pix = min(pixi, SW_NPIX - 1);
        const int py = pix / SW_PW, px = pix - py * SW_PW;
        iyx[b] = (py << 16) | px;
        ilds[b] = (pixi < SW_NPIX && live_slot) ? pix * 4 + (sslot ^ ((pix >> 2) & 3)) : SW_NPIX * 4 + (tid & 3);
        ioff[b] = (unsigned)((py * W + px) * C);
    }
    const T* base = reinterpret_cast<const T*>(ssrc.ptr) + ctx.c + (size_t)(n + ssrc.n_off) * H * W * C;

    // position of a tile stream (wave-uniform): tile coordinates advanced by one tile per iteration; past the workgroup's range the
    // stream is a ghost
    struct Pos { int tx, ty, j; };
    auto pos_at = [&](int j) {
        Pos q;
        const int t = t_first + j;
        q.ty = t / tiles_x;
        q.tx = t - q.ty * tiles_x;
        q.j = j;
        return q;
    };
    auto advance = [&](Pos& q) {
        q.j += 1;
        q.tx += 1;
        if (q.tx == tiles_x) { q.tx = 0; q.ty += 1; }
    };
    // the halo tile lies inside the image: no clamping, no zero fill (about 80 % of the tiles of a 400 x 400 image)
    auto interior = [&](const Pos& q) {
        const int yh = q.ty * TH - 1, xh = q.tx * TW - 1;
        return q.j < nt && yh >= 0 && xh >= 0 && yh + SW_PH <= H && xh + SW_PW <= W;
    };

    uint4 raw[SW_NSET][NITV];                                // set j % SW_NSET carries this workgroup's tile j
    // both branches issue the same loads into the same registers in the same order (see the header comment)
    auto issue = [&](uint4 (&r)[NITV], const Pos& q) {
        const bool ghost = q.j >= nt;
        const int yh = ghost ? -(1 << 20) : q.ty * TH - 1, xh = ghost ? -(1 << 20) : q.tx * TW - 1;
        if (interior(q)) {
            const unsigned toff = (unsigned)((yh * W + xh) * C);
#pragma unroll
            for (int b = 0; b < NITV; ++b) r[b] = ld16(base + (toff + ioff[b]));
        } else {
#pragma unroll
            for (int b = 0; b < NITV; ++b) {
                const int y = min(max(yh + (iyx[b] >> 16), 0), H - 1), x = min(max(xh + (iyx[b] & 0xffff), 0), W - 1);
                r[b] = ld16(base + (unsigned)((y * W + x) * C));
            }
        }
    };
    auto transform = [&](auto kind_c, uint4 u) {
        constexpr int KIND = decltype(kind_c)::value;
        if constexpr (KIND == 0) return u;
        float v[S];
        Slot<T>::unpack(u, v);
        if constexpr (KIND == 1) {
            // BatchNorm + ReLU producer: affine in fp32, round to bf16, ReLU on the packed pair (v_pk_max_i16 against 0: a negative
            // bf16 is a negative int16; rounding keeps the sign, so this is bf16(max(y, 0)) bit for bit)
#pragma unroll
            for (int e = 0; e < S; ++e) v[e] = v[e] * sc[e] + sh[e];
            u = Slot<T>::pack(v);
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u.x) : "v"(u.x));
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u.y) : "v"(u.y));
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u.z) : "v"(u.z));
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(u.w) : "v"(u.w));
            return u;
        }
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = act_fn(v[e] * sc[e] + sh[e], slope);
        return Slot<T>::pack(v);
    };
    auto consume_k = [&](auto kind_c, const uint4 (&r)[NITV], const Pos& q) {
        uint4* s_in = reinterpret_cast<uint4*>(s_inb + (q.j & 1) * SW_BUF_BYTES);
        if constexpr ((XP & 4) != 0) {                       // wait for the set, touch nothing else
            unsigned acc = 0;
#pragma unroll
            for (int b = 0; b < NITV; ++b) acc ^= r[b].x ^ r[b].y ^ r[b].z ^ r[b].w;
            if (acc == 0x12345678u) s_in[SW_NPIX * 4] = make_uint4(acc, 0, 0, 0);
            return;
        }
        if (interior(q)) {
#pragma unroll
            for (int b = 0; b < NITV; ++b) s_in[ilds[b]] = transform(kind_c, r[b]);
        } else {
            const bool ghost = q.j >= nt;
            const int yh = ghost ? -(1 << 20) : q.ty * TH - 1, xh = ghost ? -(1 << 20) : q.tx * TW - 1;
#pragma unroll
            for (int b = 0; b < NITV; ++b) {
                const int y = yh + (iyx[b] >> 16), x = xh + (iyx[b] & 0xffff);
                const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && live_slot;
                uint4 u = transform(kind_c, r[b]);
                unsigned keep = in ? 0xffffffffu : 0u;       // a mask, not a select: no branch around the transform (conv_device.h pfu_consume)
                asm("" : "+v"(keep));
                u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
                s_in[ilds[b]] = u;
            }
        }
    };

    auto consume = [&](const uint4 (&r)[NITV], const Pos& q) {       // one wave-uniform branch per tile, not per item
        if (kind == 1) consume_k(std::integral_constant<int, 1>(), r, q);
        else if (kind == 0) consume_k(std::integral_constant<int, 0>(), r, q);
        else consume_k(std::integral_constant<int, 2>(), r, q);
    };

    // ---- compute constants: this lane's pixel column li of tile row `wave`; fragment (kh, kw, ks) of the row = halo pixel
    //      (wave + kh, li + kw), 16-byte slot 2 ks + h
    int foff[RPW][3][3];
#pragma unroll
    for (int rw = 0; rw < RPW; ++rw)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int pix = (wave + NWV * rw + kh) * SW_PW + li + kw;
                foff[rw][kh][kw] = (pix * 4 + (h ^ ((pix >> 2) & 3))) * 16;  // slot h; slot 2 + h = this ^ 32 bytes
            }
    float sa[NV][S], sb[NV][S];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < S; ++e) sa[v][e] = sb[v][e] = 0.f;
    T* out = reinterpret_cast<T*>(p.out) + (size_t)n * H * W * p.Cout;
    const unsigned ooff = (unsigned)((wave * W + li) * p.Cout + 8 * h);       // this lane's output vector 0 inside a tile
    const bool vec_ok = (p.Cout % S) == 0;
    const bool two = __builtin_amdgcn_readfirstlane((int)(p.Cout > 16));     // <= 16 output channels: one vector per lane

    auto compute_row = [&](const Pos& q, const int rw) {
        const int x0 = q.tx * TW, y0 = q.ty * TH;
        const int trow = wave + NWV * rw;                    // this wave's tile row
        const char* s_in = s_inb + (q.j & 1) * SW_BUF_BYTES;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (!(XP & 1)) {
            if constexpr (NKS == 1) {
                // <= 16 input channels: all nine fragments requested before the first MFMA (36 registers this instantiation has): one LDS
                // round trip per tile instead of one per kernel row
                uint4 f[3][3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) f[kh][kw] = *reinterpret_cast<const uint4*>(s_in + foff[rw][kh][kw]);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[kh * 3 + kw][0]),
                                                                      __builtin_bit_cast(bf16x8, f[kh][kw]), acc, 0, 0, 0);
            } else {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                uint4 f[3][NKS];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) f[kw][ks] = *reinterpret_cast<const uint4*>(s_in + (foff[rw][kh][kw] ^ (ks * 32)));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[kh * 3 + kw][ks]),
                                                                      __builtin_bit_cast(bf16x8, f[kw][ks]), acc, 0, 0, 0);
            }
            }
        }
        if constexpr ((XP & 16) != 0) return;
        const bool live = q.j < nt;                          // ghost iterations run the same code with every lane off
        const bool full = live && y0 + TH <= H && x0 + TW <= W;               // wave-uniform: every pixel of the tile exists
        const bool valid = live && y0 + trow < H && x0 + li < W;
        const unsigned toff = (unsigned)((y0 * W + x0) * p.Cout);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if ((OUTV == 1 || OUTV == 3) && v == 1) continue;
            if (OUTV == 0 && v == 1 && !two) continue;
            const int cb = 16 * v + 8 * h;
            float vec[S], o[S];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ua = __float_as_uint(acc[8 * v + j]);
                const unsigned ub = __float_as_uint(acc[8 * v + 4 + j]);
                // (conv_device.h rd_half_swap: the wrong pixels of round 5 were the op_sel'd packed bias add behind this regroup, not the regroup)
                const HalfSwap rs = rd_half_swap(ua, ub, h);
                vec[j] = __uint_as_float(rs.r0);
                vec[4 + j] = __uint_as_float(rs.r1);
            }
            const float4 b0 = *reinterpret_cast<const float4*>(s_bias + cb), b1 = *reinterpret_cast<const float4*>(s_bias + cb + 4);
            const float bs[S] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < S; ++e) o[e] = vec[e] + bs[e];
            T* dst = out + (toff + ooff + (unsigned)(NWV * rw * W * p.Cout) + 16 * v);
            if constexpr (OUTV == 3) {
                // <= 4 output channels (the out1 convs: 2 classes, 3 colours): four scalar stores per lane, unconditional like the
                // vector stores below -- lanes of the upper half-wave, pixels outside the image and channels beyond Cout write to the
                // trash record
                const bool st = valid && h == 0;
                T* tr = reinterpret_cast<T*>(&sw_trash[tid]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T* qd = (st && e < p.Cout) ? dst + e : tr + e;
                    if constexpr (!(XP & 2)) *qd = from_f<T>(o[e]);
                }
                if (st) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sa[v][e] += vec[e];
                        sb[v][e] += vec[e] * vec[e];
                    }
                }
            } else if constexpr (OUTV != 0) {
                // the store is UNCONDITIONAL on every path (lanes outside the image / ghost tiles write to a trash record in device
                // memory): stores count in vmcnt like loads do, and a store that exists on one path only makes every later wait
                // conservative
                if (full) {
                    if constexpr (!(XP & 2)) *reinterpret_cast<uint4*>(dst) = Slot<T>::pack(o);
#pragma unroll
                    for (int e = 0; e < S; ++e) {
                        sa[v][e] += vec[e];                    // sums exclude the bias (ramdsir.h, RD_STAT_SLOTS)
                        sb[v][e] += vec[e] * vec[e];
                    }
                } else {
                    if constexpr (!(XP & 2)) {
                        uint4* qd = valid ? reinterpret_cast<uint4*>(dst) : &sw_trash[tid];
                        *qd = Slot<T>::pack(o);
                    }
                    if (valid) {
#pragma unroll
                        for (int e = 0; e < S; ++e) {
                            sa[v][e] += vec[e];
                            sb[v][e] += vec[e] * vec[e];
                        }
                    }
                }
            } else {
                if (!(valid && cb < p.Cout)) continue;
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    sa[v][e] += vec[e];
                    sb[v][e] += vec[e] * vec[e];
                }
                if constexpr (!(XP & 2)) store_vec<T>(dst, o, p.Cout - cb, vec_ok);
            }
        }
    };
    auto compute = [&](const Pos& q) {
#pragma unroll
        for (int rw = 0; rw < RPW; ++rw) compute_row(q, rw);
    };

    // ---- pipeline: sets 0 .. NSET-1 requested, tile 0 transformed, its set re-used for tile NSET
    Pos qc = pos_at(0), ql = pos_at(0), qi = pos_at(0);      // streams: compute (tile it), transform (it + 1), request (it + 1 + NSET)
    // The prologue issues the stores of an iteration too (to the trash record): the compiler's wait counts at the top of the loop are
    // the minimum over the paths that reach it, and a prologue without stores would make the loop wait as if its own stores did not
    // exist -- i.e. for the loads requested only ONE iteration earlier.
    auto fake_stores = [&]() {
#pragma unroll
        for (int rw = 0; rw < RPW; ++rw)
        if constexpr (OUTV == 3 && !(XP & 2)) {
            T* tr = reinterpret_cast<T*>(&sw_trash[512 + tid]);
#pragma unroll
            for (int e = 0; e < 4; ++e) tr[e] = from_f<T>(0.f);
        } else if constexpr (OUTV != 0 && !(XP & 2)) {
#pragma unroll
            for (int v = 0; v < (OUTV == 2 ? 2 : 1); ++v) sw_trash[v * 512 + tid] = make_uint4(0, 0, 0, 0);
        }
    };
#pragma unroll
    for (int k = 0; k < SW_NSET; ++k) { issue(raw[k], qi); advance(qi); if (k) fake_stores(); }
    {
        // BatchNorm finalize of the producer folded into this launch: the coefficients reach slot_ctx through LDS (src[] points there)
        rd_src_t src[2];
        rdfin::conv_prologue_lds(p, fa, reinterpret_cast<float*>(smem + SW_FIN), rdfin::FIN_LDS_FLOATS, src);
        slot_ctx<T>(ctx, src, p.nsrc, p.Cin, g, sslot * S);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            sc[e] = rawm ? 1.f : ctx.sc[e];
            sh[e] = rawm ? 0.f : ctx.sh[e];
        }
    }
    __syncthreads();                                         // zero-fill done before the first transformed tile lands
    consume(raw[0], ql);
    advance(ql);
    issue(raw[0], qi);
    advance(qi);
    fake_stores();
    sw_barrier();                                            // tile 0 is in buffer 0
    // iteration `it`: tile it+1 -> the other buffer, tile it+1+SW_NSET requested into its set, tile `it` multiplied and stored
    // The two waves of a SIMD (w and w + 4) run the two halves of an iteration in OPPOSITE order: one transforms tile it+1 (VALU, LDS
    // writes) while the other multiplies tile it (the matrix pipe), then they swap.  With the same order in all eight waves both
    // waves of a SIMD reach their 18 dependent MFMAs together and one of them waits out the other's 1152 cycles.  Two loops, not a
    // branch inside one: each is straight-line code, so the compiler still counts the loads and stores in flight (see above).
    if (XP & 32 ? true : wave < NWV / 2) {
        for (int it0 = 0; it0 < nt; it0 += SW_NSET) {
#pragma unroll
            for (int k = 0; k < SW_NSET; ++k) {
                consume(raw[(k + 1) % SW_NSET], ql);
                advance(ql);
                issue(raw[(k + 1) % SW_NSET], qi);
                advance(qi);
                compute(qc);
                advance(qc);
                sw_barrier();
            }
        }
    } else {
        for (int it0 = 0; it0 < nt; it0 += SW_NSET) {
#pragma unroll
            for (int k = 0; k < SW_NSET; ++k) {
                compute(qc);
                advance(qc);
                consume(raw[(k + 1) % SW_NSET], ql);
                advance(ql);
                issue(raw[(k + 1) % SW_NSET], qi);
                advance(qi);
                sw_barrier();
            }
        }
    }

    // ---- BatchNorm sums of all tiles: summed over the 32 lanes of a half-wave (conv_device.h half_wave_sums), one LDS atomic per lane,
    //      one global set per workgroup
    __syncthreads();                                         // the tile buffers are free: per-wave sums land there, added in wave order
    double* s_part = reinterpret_cast<double*>(smem);        // [NWV][64]
    store_half_wave_sums16<S, NV>(s_part, wave, sa, sb, li, h);
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += s_part[w * 64 + tid];
        s_red[tid] = t;
    }
    __syncthreads();
#ifdef RD_DEBUG_SWITCHES
    // debug library, RD_SW_JITTER=1: a pseudo-random wait in front of the slot atomics, so that the ORDER in which the workgroups of a launch
    // arrive at a slot differs from run to run (scripts/determinism_probe.py: does any result depend on that order?)
    if (tiles_per_wg < 0 || sw_jitter_on()) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned long long w = (t0 * 2654435761ull >> 7) & 4095;
        while (__builtin_readcyclecounter() - t0 < w) __builtin_amdgcn_s_sleep(1);
    }
#endif
    if (tid < 32 && tid < p.Cout && p.stats) {
        const size_t so = (((size_t)g * RD_STAT_SLOTS + slot) * p.Cout + tid) * 2;
        atomicAdd(&p.stats[so + 0], s_red[tid * 2 + 0]);
        atomicAdd(&p.stats[so + 1], s_red[tid * 2 + 1]);
    }
}

inline bool sw_aligned16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- host side
// forward launches of conv_small's class with whole 16-byte channel slots and single-operand sources; RD_CONV_PP_NA otherwise
int rd_conv_small_fwd_dispatch(const rd_conv_t& p, int dtype, hipStream_t st) {
    static const int on = rd_switch("RD_CONV_SMALL_FWD", 1);
    if (!on || dtype != RD_BF16 || p.taps != 9 || p.emode != 0 || p.CinPad != 32 || p.CoutPad != 32 || p.Cout > 32) return RD_CONV_PP_NA;
    for (int i = 0; i < p.nsrc; ++i) {
        const rd_src_t& s = p.src[i];
        if (!(s.mode == RD_SRC_RAW || s.mode == RD_SRC_AFF || s.mode == RD_SRC_AFFACT) || s.C % 8 || !sw_aligned16(s.ptr)) return RD_CONV_PP_NA;
    }
    if (!sw_aligned16(p.w) || ((p.Cout % 8) == 0 && !sw_aligned16(p.out))) return RD_CONV_PP_NA;
    if ((size_t)p.N * p.H * p.W * 32 >= (1ull << 31)) return RD_CONV_PP_NA;        // 32-bit element offsets inside the kernel
    const int ntiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    const bool limited = p.cu_limit > 0 && p.cu_limit < rd_num_cus();
    const int nl_ = (p.Cin + 7) / 8;
    static const int nwv_sw = rd_switch("RD_SW_NWV", RD_SW_NWV_DEFAULT);             // 8: one 512-thread workgroup per CU; 4: two of 256
    // two rows per wave need a second row accumulator, twice the fragment offsets and loader items: the 32-channel-input instantiations
    // (72 registers of weights) spill 200-470 bytes per lane that way and stay on one workgroup per CU
    static const int nwv_mask = rd_switch("RD_SW_NWV_MASK", 15);              // debug: 1 Cout 16, 2 Cout 32, 4 Cout <= 4 / other, 8 inputs of <= 8 channels
    const int cls = (nl_ <= 1 ? 8 : (p.Cout == 16 ? 1 : (p.Cout == 32 ? 2 : 4)));
    const int nwv = (nwv_sw == 4 && nl_ <= 2 && (nwv_mask & cls)) ? 4 : 8;
    const long slots = (long)(limited ? p.cu_limit : rd_num_cus()) * (8 / nwv);
    // tiles per workgroup: whole rounds of resident workgroups; a workgroup pays about two tile-times of pipeline fill + drain
    int tpw = 0;
    double best = 1e30;
    const int tmin = limited ? (int)(((long)ntiles * p.N + slots - 1) / slots) : 1;
    for (int t = tmin < 1 ? 1 : tmin; (t <= 64 || limited) && t <= ntiles; ++t) {
        const long wgs = (long)((ntiles + t - 1) / t) * p.N;
        const double cost = (double)((wgs + slots - 1) / slots) * (t + 2.0);
        if (cost < best - 1e-9) { best = cost; tpw = t; }
        if (limited) break;
    }
    if (tpw <= 0) tpw = ntiles;
    tpw = (tpw + SW_NSET - 1) / SW_NSET * SW_NSET;          // the tile loop is unrolled by the register sets: whole groups, fewer ghost tiles
    static const int tpw_forced = rd_switch("RD_SW_TPW", 0);          // debug build: the tests force multi-tile workgroups on small images
    if (tpw_forced > 0) tpw = tpw_forced < ntiles ? tpw_forced : ntiles;
    dim3 grid((ntiles + tpw - 1) / tpw, 1, p.N);
    const int nl = (p.Cin + 7) / 8;
#define RD_SW_LAUNCH(NSL, OUTV, XP) do { \
        static bool attr_set = false; \
        if (!attr_set) { \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_fwd_kernel<NSL, OUTV, XP, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_fwd_kernel<NSL, OUTV, XP, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS); \
            attr_set = true; \
        } \
        if (nwv == 4) rd_launch((conv_small_fwd_kernel<NSL, OUTV, XP, 4>), grid, dim3(256), SW_LDS, st, p, tpw, rdfin::current()); \
        else rd_launch((conv_small_fwd_kernel<NSL, OUTV, XP, 8>), grid, dim3(512), SW_LDS, st, p, tpw, rdfin::current()); \
        return (int)hipGetLastError(); \
    } while (0)
#ifdef RD_DEBUG_SWITCHES
    if (nl > 2 && p.Cout == 32) {                           // timing experiments on the 32 -> 32 kernel
        switch (rd_switch("RD_SW_EXP", 0)) {
        case 1: RD_SW_LAUNCH(4, 2, 1);
        case 2: RD_SW_LAUNCH(4, 2, 2);
        case 3: RD_SW_LAUNCH(4, 2, 3);
        case 4: RD_SW_LAUNCH(4, 2, 4);
        case 5: RD_SW_LAUNCH(4, 2, 5);
        case 6: RD_SW_LAUNCH(4, 2, 6);
        case 18: RD_SW_LAUNCH(4, 2, 18);
        case 22: RD_SW_LAUNCH(4, 2, 22);
        case 7: RD_SW_LAUNCH(4, 2, 7);
        case 19: RD_SW_LAUNCH(4, 2, 19);
        case 23: RD_SW_LAUNCH(4, 2, 23);
        case 32: RD_SW_LAUNCH(4, 2, 32);
        default: break;
        }
    }
#endif
#ifdef RD_DEBUG_SWITCHES
    {
        static bool jit_set = false;
        if (!jit_set) {
            const int j = rd_switch("RD_SW_JITTER", 0);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(sw_jitter_flag), &j, sizeof(int));
            jit_set = true;
        }
    }
#endif
    static const int narrow_on = rd_switch("RD_SW_NARROW", 1);
    const int outv = p.Cout == 32 ? 2 : (p.Cout == 16 ? 1 : ((p.Cout <= 4 && narrow_on) ? 3 : 0));
#define RD_SW_OUT(NSL) do { if (outv == 2) RD_SW_LAUNCH(NSL, 2, 0); if (outv == 1) RD_SW_LAUNCH(NSL, 1, 0); \
                            if (outv == 3) RD_SW_LAUNCH(NSL, 3, 0); RD_SW_LAUNCH(NSL, 0, 0); } while (0)
    if (nl <= 1) RD_SW_OUT(1);
    if (nl <= 2) RD_SW_OUT(2);
    RD_SW_OUT(4);
#undef RD_SW_OUT
#undef RD_SW_LAUNCH
}
