// wgrad_sym.hip -- weight gradient of the 3x3 layers with whole 128-channel blocks of gradient channels (bf16, plain sources): 128 x 64
// channel blocks of dW per workgroup on eight symmetric waves.  Dispatched by wgrad.hip (wgrad_geom: `sym`), reduced by its
// wgrad_reduce_kernel.  cuDNN's wgrad behind /root/reference/code/networks/unet.py:37-43,81-88 (ConvD / ConvU convs, backward).
#include "conv_device.h"
#include "conv_dispatch.h"
#include "wgrad_tr.h"

#ifndef RD_SYM_ORDER_EXPR
#define RD_SYM_ORDER_EXPR (__builtin_amdgcn_readfirstlane(slot) & 1)
#endif
namespace {

// ------------------------------------------------------------------------------------ 128-wide blocks, eight symmetric waves (round 6)
// wgrad_ws_kernel is its loader waves: one wave per SIMD runs the operand transform (BatchNorm affine + activation of `a`, P g + Q z + R
// of the gradient pair) as one dependent VALU chain, ~515 instructions per 64 x 64-channel tile at 6.5 cycles each, and the layers with
// >= 128 channels repeat that transform for every 64 x 64 block of dW -- 2-4 times per operand tile (traffic 1.92 x the algorithmic
// bytes, 0.06-0.10 of the MFMA roof on AI-789 layers: VERDICT round 5).  Here a workgroup owns 128 gradient channels x 64 input
// channels x 9 taps: the 64-channel `a` tile is transformed ONCE per 128 output channels, and all eight waves are equal -- each owns
// one 32 x 32 x 9 block (144 accumulator registers, as before) AND an eighth of the loader's items.  The two waves of a SIMD run the
// two halves of an iteration in OPPOSITE order (`order` = wave / 4):
//     order 0:  products of tile t (LDS buffer t & 1)     | transform tile t + 1 into the other buffer, request tile t + 2 | barrier
//     order 1:  transform tile t + 1, request tile t + 2  | products of tile t                                            | barrier
// so that on every SIMD one wave's transform chain runs beside the other wave's MFMAs -- what the warp-specialised kernel bought with
// dedicated waves, without idling the loader's registers during the products or the MFMA wave's during the transform.  Per thread and
// tile: 4 `a` items + 4 gradient items (pairs) = ~390 VALU for 72 MFMAs (the ws loader: 515 for 72).  One register set: the request for
// tile t + 2 goes out at the end of the transform of t + 1 and has a whole product phase to land.  Coefficients live in an LDS table
// ([G][a: scale, shift | z: P, R, Q]) and are read per tile (10 ds_read_b128), not held in 40 registers.
// LDS: 2 x (26.0 KB `a` + 32 KB gradient (two 64-channel sub-tiles, each in wgrad_ws_kernel's swizzled layout) + 8 KB dummy records).
template <int NQZ>
__global__ __launch_bounds__(512, 1) void wgrad_sym_kernel(const rd_wgrad_t p, int CoutPadW, int CinPadW, int total_tiles, const rdfin::FinArg fa) {
    const bool fin_tab = fa.which >= 0 && !(fa.flags & RD_FIN_OWNER);
    if (!fin_tab) rdfin::prologue(fa);
    typedef bf16_t T;
    constexpr int S = 8, TAPS = 9;
    constexpr int THW = 4;
    constexpr int PH = THW + 2, PW = TW + 2, NPIX = PH * PW;
    constexpr int PA = tr_pitch(64), PZ = tr_pitch(64);
    constexpr int A_BYTES = (NPIX + 4) * PA, ZSUB = THW * TW * PZ, Z_BYTES = 2 * ZSUB, DUMMY = 512 * 16, BUF = A_BYTES + Z_BYTES + DUMMY;
    constexpr int CROW = 2 * 64 + 3 * 128;                    // floats of the coefficient table per image group
    static_assert(PW == 34 && TW == 32, "item geometry below");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_coef = reinterpret_cast<float*>(smem + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* s_simd = reinterpret_cast<int*>(s_coef + p.G * CROW);        // [4]: waves counted per SIMD (below)
    const int nbase = blockIdx.y * 128, cbase = blockIdx.z * 64;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + THW - 1) / THW;
    const int H = p.H, W = p.W;
    auto coords = [&](int tile, int& n, int& y0, int& x0) {
        n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        y0 = (trem / tiles_x) * THW;
        x0 = (trem % tiles_x) * TW;
    };
    // the workgroup's 64 input channels lie inside ONE source (the host checks a[0].C % 64 == 0 for concatenated inputs): base pointers
    // stay wave-uniform (scalar registers) and a load is `scalar base + 32-bit lane offset`
    const int sia = (p.na == 1 || cbase < p.a[0].C) ? 0 : 1;
    const rd_src_t sda = select_src(p.a, sia);
    const int ca0 = cbase - (sia ? p.a[0].C : 0);             // first channel of the block inside its source
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < 2 * BUF / 16; i += 512) z4[i] = make_uint4(0, 0, 0, 0);
        if (tid < 4) s_simd[tid] = 0;
        // coefficient table, every image group, read from global memory HERE and nowhere in the loop (a conditional vector-memory load
        // anywhere in the loop costs the compiler its count of the loads in flight: wgrad_ws_kernel)
        for (int i = tid; i < p.G * 192; i += 512) {
            const int g = i / 192, r = i - g * 192;
            const bool which = r >= 64;                       // false: input channel r, true: gradient channel r - 64
            const int c = which ? r - 64 : r;
            const rd_src_t sd = which ? p.dz : sda;
            const int cc = which ? nbase + c : ca0 + c;
            const bool live = which ? (nbase + c < p.Cout) : (cbase + c < p.Cin);
            const bool raw = sd.mode == RD_SRC_RAW || !live, bwd = sd.mode == RD_SRC_BNBWD && live;
            const int gs = sd.g_fixed >= 0 ? sd.g_fixed : g;
            float* row = s_coef + g * CROW;
            if (which) {
                float P, Q, R;
                if (fin_tab && bwd) {
                    rdfin::bwd_pair(fa, gs, cc, P, Q, R);
                } else {
                    P = raw ? 1.f : sd.scale[gs * sd.C + cc];
                    R = raw ? 0.f : sd.shift[gs * sd.C + cc];
                    Q = bwd ? sd.q[gs * sd.C + cc] : 0.f;
                }
                row[128 + c] = P;
                row[256 + c] = R;
                row[384 + c] = Q;
            } else {
                row[c] = raw ? 1.f : sd.scale[gs * sd.C + cc];
                row[64 + c] = raw ? 0.f : sd.shift[gs * sd.C + cc];
            }
        }
    }

    // ---- loader part of every thread.  Item geometry is LINEAR in the item index, so that a thread keeps a base and the strides are
    // wave-uniform (the 144 accumulator registers leave no room for per-item tables):
    //   `a` (6 x 34 halo pixels x 8 slots): slot tid % 8, pixel lane q = tid / 8; items 0-2: rows 2b + (q >= 34), column q mod 34;
    //        item 3: the four columns 30..33 of the odd rows that are left (q < 12)
    //   gradient (4 x 32 pixels x 16 slots): slot tid % 16, column tid / 16, item b = row b
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int sla = tid & 7, qa = tid >> 3, slz = tid & 15, qz = tid >> 4;
    const int hi = qa >= 34 ? 1 : 0, pxa = qa - 34 * hi;
    const bool has3 = qa < 12;
    const int py3 = 1 + 2 * (qa >> 2), px3 = 30 + (qa & 3);
    const bool live_a = cbase + sla * S < p.Cin, live_z = nbase + slz * S < p.Cout;
    const int Ca = sda.C, Cz = p.dz.C;
    const int ca = live_a ? ca0 + sla * S : 0, cz = live_z ? nbase + slz * S : 0;
    const int dummy_a = A_BYTES + Z_BYTES + tid * 16, dummy_z = Z_BYTES + tid * 16;               // (relative to s_a / s_z)
    const int lds_a0 = live_a ? tr_off<64>(hi * PW + pxa, sla) : dummy_a, lds_a_str = live_a ? 2 * PW * PA : 0;
    const int lds_a3 = (live_a && has3) ? tr_off<64>(py3 * PW + px3, sla) : dummy_a;
    const int lds_z0 = live_z ? (slz >> 3) * ZSUB + tr_off<64>(qz, slz & 7) : dummy_z, lds_z_str = live_z ? TW * PZ : 0;
    const int ioff_a0 = (hi * W + pxa) * Ca, ioff_a3 = (py3 * W + px3) * Ca, ioff_z0 = qz * Cz;
    const char* const pa0 = reinterpret_cast<const char*>(sda.ptr);
    const char* const pz0 = reinterpret_cast<const char*>(p.dz.ptr);
    const char* const pz1 = p.dz.mode == RD_SRC_BNBWD ? reinterpret_cast<const char*>(p.dz.ptr2) : pz0;
    const float slope_a = sda.mode == RD_SRC_AFFACT ? sda.slope : 1.f;
    const bool z_raw = p.dz.mode == RD_SRC_RAW, a_raw = sda.mode == RD_SRC_RAW;
    const int last_a = (H * W - 1) * Ca, last_z = (H * W - 1) * Cz;

    uint4 raw_a[4], raw_z[4][NQZ];
    // Register budget (two waves per SIMD: 256): 144 accumulators + ~20 of loader state leave ~90.  The `a` items (16 registers) are
    // requested ONE TILE AHEAD and stay in flight across the product phase; the gradient items (32 registers with the pair) are
    // requested at the head of the transform phase and consumed at its end, behind the `a` items -- their latency is exposed to THIS
    // wave, but the other wave of the SIMD is in its product phase then (the opposite order), so the matrix pipe does not idle.
    auto issue_a = [&](int tile) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const bool ghost = tile >= total_tiles;               // past the end: one cache line of image 0, so that the loop stays branch-free
        n = ghost ? 0 : n;
        const bool border = x0 + TW + 1 > W;                  // wave-uniform: the halo tile hangs over the right edge
        const char* ba = pa0 + (size_t)(n + sda.n_off) * H * W * Ca * sizeof(T);
        const int toff_a = ghost ? -(1 << 28) : ((y0 - 1) * W + x0 - 1) * Ca;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // items beyond the right edge re-read the last pixel of their row (pfu_issue_pre: otherwise they fetch the NEXT rows' lines)
            const int px = b < 3 ? pxa : px3;
            int off = toff_a + (b < 3 ? ioff_a0 + b * 2 * W * Ca : ioff_a3);
            if (border) {
                const int x = x0 - 1 + px;
                off += __mul24(min(x, W - 1) - x, Ca);
            }
            off = min(max(off, 0), last_a) + ca;
            raw_a[b] = ld16(ba + (unsigned)(off * (int)sizeof(T)));
            __builtin_amdgcn_sched_barrier(0);                // loads stay in item order: the consumer's wait counts depend on it
        }
    };
    auto issue_z = [&](int tile) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const bool ghost = tile >= total_tiles;
        n = ghost ? 0 : n;
        const bool border = x0 + TW + 1 > W;
        const char* bz0 = pz0 + (size_t)(n + p.dz.n_off) * H * W * Cz * sizeof(T);
        const char* bz1 = pz1 + (size_t)(n + p.dz.n_off) * H * W * Cz * sizeof(T);
        const int toff_z = ghost ? -(1 << 28) : (y0 * W + x0) * Cz;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int off = toff_z + ioff_z0 + b * W * Cz;
            if (border) {
                const int x = x0 + qz;
                off += __mul24(min(x, W - 1) - x, Cz);
            }
            off = min(max(off, 0), last_z) + cz;
            raw_z[b][0] = ld16(bz0 + (unsigned)(off * (int)sizeof(T)));
            if constexpr (NQZ == 2) raw_z[b][1] = ld16(bz1 + (unsigned)(off * (int)sizeof(T)));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int stride = gridDim.x;
    // tile -> LDS buffer it & 1 (its `a` items are in registers, its gradient items are requested here); the next tile's `a` items
    auto fill = [&](int tile, int it) {
        int n, y0, x0;
        coords(tile, n, y0, x0);
        const bool ghost = tile >= total_tiles;               // zeros into the buffer nobody reads any more
        n = ghost ? 0 : n;
        y0 = ghost ? -(1 << 20) : y0;
        x0 = ghost ? -(1 << 20) : x0;
        const int g = group_of(gm, n);
        char* s_a = smem + (it & 1) * BUF;
        char* s_z = s_a + A_BYTES;
        {
            float sc[S], sh[S];
            const float4* rc = reinterpret_cast<const float4*>(s_coef + g * CROW + sla * S);
            const float4 c0 = rc[0], c1 = rc[1], h0 = rc[16], h1 = rc[17];
            sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
            sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int y = y0 - 1 + (b < 3 ? 2 * b + hi : py3), x = x0 - 1 + (b < 3 ? pxa : px3);
                const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                // zeros outside the image by an AND with a laundered mask, not a select (pfu_consume: a select becomes a branch around
                // the transform and the loads behind it are then waited for with vmcnt(0))
                unsigned keep = in ? 0xffffffffu : 0u;
                asm("" : "+v"(keep));
                uint4 u = raw_a[b];
                if (!a_raw) {                                 // (uniform)
                    float v[S];
                    Slot<T>::unpack(raw_a[b], v);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = act_fn(sc[e] * v[e] + sh[e], slope_a);
                    u = Slot<T>::pack(v);
                }
                u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
                *reinterpret_cast<uint4*>(s_a + (b < 3 ? lds_a0 + b * lds_a_str : lds_a3)) = u;
                __builtin_amdgcn_sched_barrier(0);            // one item's temporaries at a time (register budget: see products)
            }
        }
        {
            float P[S], R[S], Q[S];
            const float4* rc = reinterpret_cast<const float4*>(s_coef + g * CROW + 128 + slz * S);
            const float4 c0 = rc[0], c1 = rc[1], h0 = rc[32], h1 = rc[33], q0 = rc[64], q1 = rc[65];
            P[0] = c0.x; P[1] = c0.y; P[2] = c0.z; P[3] = c0.w; P[4] = c1.x; P[5] = c1.y; P[6] = c1.z; P[7] = c1.w;
            R[0] = h0.x; R[1] = h0.y; R[2] = h0.z; R[3] = h0.w; R[4] = h1.x; R[5] = h1.y; R[6] = h1.z; R[7] = h1.w;
            Q[0] = q0.x; Q[1] = q0.y; Q[2] = q0.z; Q[3] = q0.w; Q[4] = q1.x; Q[5] = q1.y; Q[6] = q1.z; Q[7] = q1.w;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int y = y0 + b, x = x0 + qz;
                const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                unsigned keep = in ? 0xffffffffu : 0u;
                asm("" : "+v"(keep));
                uint4 u = raw_z[b][0];
                if (!(z_raw && NQZ == 1)) {
                    float v[S];
                    Slot<T>::unpack(raw_z[b][0], v);
                    if constexpr (NQZ == 2) {
                        float zz[S];
                        Slot<T>::unpack(raw_z[b][1], zz);
#pragma unroll
                        for (int e = 0; e < S; ++e) v[e] = bn_bwd_value(P[e], v[e], Q[e], zz[e], R[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < S; ++e) v[e] = P[e] * v[e] + R[e];
                    }
                    u = Slot<T>::pack(v);
                }
                u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
                *reinterpret_cast<uint4*>(s_z + lds_z0 + b * lds_z_str) = u;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        issue_a(tile + stride);
        issue_z(tile + stride);
    };

    // ---- MFMA part of every wave: gradient block mbz (of 4), input block nb (of 2)
    const int mbz = wave & 3, nb = wave >> 2;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int i16 = lane & 15, gq = lane >> 4;
    const int pix0 = (gq >> 1) * 8 + (i16 >> 2), sub = ((gq & 1) * 16 + (i16 & 3) * 4) * 2;
    const int zoff = (mbz >> 1) * ZSUB + tr_frag<64>(pix0, mbz & 1, sub, 0);
    static_assert(PW % 4 == 2, "halo rows of odd index flip the half swizzle (tr_frag)");
    const int aoff_r[2] = {tr_frag<64>(pix0, nb, sub, 0), tr_frag<64>(pix0, nb, sub, 1)};
    auto products = [&](int it) {
        const char* s_a = smem + (it & 1) * BUF;
        const char* s_z = s_a + A_BYTES;
        // 12 steps = (K half ks) x (halo row r).  The fragments of step i + 1 are requested in front of the MFMAs of step i and a
        // scheduling barrier closes every step: left alone the scheduler hoists ALL transpose reads of a tile to the top (to hide their
        // latency) and holds ~80 fragment registers -- with 144 accumulators, the register set in flight for the next tile and two
        // waves per SIMD that spills.  Gradient fragments: at most three output rows of one K half are live.
        uint4 zf[THW];
        uint2 c0, c1, c2;
        auto rd_a = [&](int step, uint2& a0, uint2& a1, uint2& a2) {
            const int ks = step / (THW + 2), r = step % (THW + 2);
            const char* ap = s_a + aoff_r[r & 1] + (r * PW + ks * 16) * PA;           // halo coords: input row = output row + kernel row
            a0 = lds_tr(ap);
            a1 = lds_tr(ap + 4 * PA);
            a2 = lds_tr(ap + 8 * PA);
        };
        auto rd_z = [&](int step) {
            const int ks = step / (THW + 2), r = step % (THW + 2);
            if (r < THW) {
                const char* zp = s_z + zoff + (r * TW + ks * 16) * PZ;
                const uint2 z0 = lds_tr(zp), z1 = lds_tr(zp + 4 * PZ);
                zf[r] = make_uint4(z0.x, z0.y, z1.x, z1.y);
            }
        };
        rd_z(0);
        rd_a(0, c0, c1, c2);
#pragma unroll
        for (int step = 0; step < 2 * (THW + 2); ++step) {
            const int r = step % (THW + 2);
            uint2 n0 = c0, n1 = c1, n2 = c2;
            if (step + 1 < 2 * (THW + 2)) {
                if ((step + 1) % (THW + 2) != 0) rd_z(step + 1);        // (a new K half overwrites the rows of the old one: behind its last use)
                rd_a(step + 1, n0, n1, n2);
            }
            const uint4 dq = make_uint4(c0.x, c0.y, c1.x, c1.y);
            const uint4 m1 = make_uint4(__builtin_amdgcn_alignbit(c0.y, c0.x, 16), __builtin_amdgcn_alignbit(c1.x, c0.y, 16),
                                        __builtin_amdgcn_alignbit(c1.y, c1.x, 16), __builtin_amdgcn_alignbit(c2.x, c1.y, 16));
            const uint4 m2 = make_uint4(c0.y, c1.x, c1.y, c2.x);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int row = r - kh;
                if (row < 0 || row >= THW) continue;
                const bf16x8 afrag = __builtin_bit_cast(bf16x8, zf[row]);
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, dq), acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m1), acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, __builtin_bit_cast(bf16x8, m2), acc[kh * 3 + 2], 0, 0, 0);
            }
            if ((step + 1) % (THW + 2) == 0 && step + 1 < 2 * (THW + 2)) rd_z(step + 1);
            c0 = n0; c1 = n1; c2 = n2;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int t0 = blockIdx.x;
    issue_a(t0);
    issue_z(t0);
    __syncthreads();                                          // the cleared buffers and the coefficient table
    // WHICH half of an iteration a wave runs first must alternate between the waves that share a SIMD -- and which waves do is the
    // dispatcher's choice, not a function of the wave index (measured: with order = wave / 4 the kernel ran at the speed of NO overlap,
    // 9 500 cycles per tile for 4 608 of MFMA issue).  So every wave reads the SIMD it sits on from the hardware id register and takes
    // the next number of that SIMD; the order is a scheduling hint only (any assignment computes the same sums: every wave passes the
    // same barriers, the loader items and the MFMA blocks follow the thread index).
    int order;
    {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int simd = (hwid >> 4) & 3;
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&s_simd[simd], 1);
        order = RD_SYM_ORDER_EXPR;
    }
    fill(t0, 0);                                              // tile 0 -> buffer 0; tile 1's `a` items requested
    __syncthreads();
    // ONE loop body for both orders (two copies of the product phase in the two arms of a branch cost the register allocator its
    // accumulators: 300 spilled registers): the order-1 waves run one transform ahead and meet the others at the barrier from the
    // middle of the body --   order 0:  P(t) F(t+1) |      order 1:  F(t+1) ; P(t) | F(t+2) ; P(t+1) | ...        (| = barrier)
    // Every wave passes the same number of barriers; between two of them an order-0 wave runs P then F, an order-1 wave F then P.
    const int ahead = 1 + order;
    if (order) fill(t0 + stride, 1);
    int it = 0;
    for (int tile = t0; tile < total_tiles; tile += stride, ++it) {
        products(it);
        __builtin_amdgcn_sched_barrier(0);
        if (order) __syncthreads();                           // buffer (it + 1) & 1 complete, buffer it & 1 free
        fill(tile + ahead * stride, it + ahead);
        if (!order) __syncthreads();
    }
    const int li = lane & 31, h = lane >> 5;
    float* out = p.partial + (size_t)blockIdx.x * TAPS * CoutPadW * CinPadW;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nrow = nbase + mbz * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ccol = cbase + nb * 32 + li;
            out[((size_t)tap * CoutPadW + nrow) * CinPadW + ccol] = acc[tap][r];
        }
}

}  // namespace

int rd_wgrad_sym_launch(const rd_wgrad_t& p, int gx, int CoutPadW, int CinPadW, hipStream_t st) {
    constexpr int PW = TW + 2;
    const int sym_lds = 2 * ((6 * PW + 4) * tr_pitch(64) + 2 * 4 * TW * tr_pitch(64) + 512 * 16) + p.G * (2 * 64 + 3 * 128) * (int)sizeof(float) + 16;
    const int tiles_sym = p.N * ((p.H + 3) / 4) * ((p.W + TW - 1) / TW);
    const dim3 sgrid(gx, CoutPadW / 128, CinPadW / 64);
    static int sym_attr = 0;
    if (sym_attr < sym_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_sym_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, sym_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_sym_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, sym_lds);
        sym_attr = sym_lds;
    }
    if (p.dz.mode == RD_SRC_BNBWD)
        rd_launch((wgrad_sym_kernel<2>), sgrid, dim3(512), sym_lds, st, p, CoutPadW, CinPadW, tiles_sym, rdfin::current());
    else
        rd_launch((wgrad_sym_kernel<1>), sgrid, dim3(512), sym_lds, st, p, CoutPadW, CinPadW, tiles_sym, rdfin::current());
    return (int)hipGetLastError();
}
