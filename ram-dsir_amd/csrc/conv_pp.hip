// conv_pp.hip -- persistent, software-pipelined 3x3 convolution for the >=64-channel layers (bf16, 64 output channels
// per tile, plain per-pixel sources).  Same implicit GEMM, tile (8 x 32 pixels x 64 channels) and MFMA
// (v_mfma_f32_32x32x16_bf16) as conv_pf_kernel (conv_pf.h); what changes is WHEN things happen:
//
//   * ONE workgroup per CU, persistent over a contiguous range of tiles (conv_pp_kernel: output-channel block fastest,
//     so the blocks of one pixel tile re-read it from this XCD's L2; conv_ws_kernel: channel block slowest, so the
//     BatchNorm sums of a block stay in registers across its tiles);
//   * the (tile, 32-channel chunk) pairs of that range form one stream of STEPS.  While the MFMAs of step s run out
//     of LDS buffer s&1, the same waves transform the raw vectors of step s+1 (BN affine + activation, bf16 pack)
//     into buffer (s+1)&1 and request the vectors of step s+2 from HBM/L2 -- item by item, placed BETWEEN the MFMA
//     groups of the nine taps, so the VALU / LDS-store / global-load issue slots ride in the shadow of the matrix
//     pipe (a 32x32x16 MFMA occupies it for 32 cycles; an in-order wave can issue ~5 other instructions meanwhile);
//   * MFMA operand fragments of tap t+1 are read from LDS before the MFMAs of tap t are issued (register double
//     buffer), one barrier per step, and the pipeline runs ACROSS tiles: the first chunk of the next tile is
//     already in LDS when the epilogue of this one starts.
//
// conv_pf_kernel runs the same work as fill -> barrier -> MFMA -> barrier with two workgroups per CU (fill VALU and MFMA
// of one wave never overlap, LDS reads are waited for right before their MFMA, every tile starts with an exposed HBM
// round trip).  Two kernels live here: conv_pp_kernel (4 waves, every wave does everything, LDS-staged epilogue) and
// conv_ws_kernel (8 waves, MFMA waves + loader waves, register epilogues); rd_conv_pp_dispatch says which launch takes which.
#include "conv_device.h"
#include "conv_epilogue.h"
#include "conv_dispatch.h"

namespace {

constexpr int PP_PH = TH + 2, PP_PW = TW + 2, PP_NPIX = PP_PH * PP_PW;     // 10 x 34 halo tile
constexpr int PP_NT = 64, PP_NIT = 6, PP_WIT = 9;
// LDS operand tiles are kept as four PLANES, one per 16-byte channel slot of the 32-channel chunk:
//   input  [slot][halo pixel]      weights [slot][tap * 64 + n]          (16 B per entry)
// so that (i) consecutive lanes of a ds_read_b128 touch consecutive 16-byte entries (conflict free without an XOR
// swizzle) and (ii) every fragment address is ONE per-lane base + a compile-time immediate (tap / row / k-step),
// i.e. no address arithmetic inside the MFMA stream.  +32 B per plane staggers the planes by 8 banks for the
// slot-fastest ds_write_b128 of the fill.
constexpr int PP_PLANE_A = (PP_NPIX + 4) * 16 + 32;                        // + 4 dummy pixels (stores of non-items)
constexpr int PP_PLANE_W = 9 * PP_NT * 16 + 32;
constexpr int PP_IN_BYTES = 4 * PP_PLANE_A, PP_W_BYTES = 4 * PP_PLANE_W;
constexpr int PP_W0 = 2 * PP_IN_BYTES;
constexpr int PP_EPI = PP_W0 + 2 * PP_W_BYTES;
constexpr int PP_EPI_BYTES = TH * TW * 32 * 4 + 64 * 8;
constexpr int PP_TAB = PP_EPI + PP_EPI_BYTES;
constexpr int PP_MAX_CHUNKS = 16;                                          // CinPad <= 512
constexpr int PP_LDS = PP_TAB + PP_MAX_CHUNKS * 4 * 48;                    // 154368 B of the CU's 160 KB

// pointers that come out of the LDS slot table have lost their address space: say "global" so that the loads are
// global_load (vmcnt only) and not flat_load (which also ties up lgkmcnt, the counter the LDS fragment reads wait on)
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16g(const void* q) {
    const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t*)(uintptr_t)q;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ldf4g(const void* q) {
    const f32x4_t v = *(const __attribute__((address_space(1))) f32x4_t*)(uintptr_t)q;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16g(void* q, const uint4& u) {
    u32x4_t v;
    v.x = u.x; v.y = u.y; v.z = u.z; v.w = u.w;
    *(__attribute__((address_space(1))) u32x4_t*)(uintptr_t)q = v;
}
#else
__device__ __forceinline__ void st16g(void* q, const uint4& u) { *reinterpret_cast<uint4*>(q) = u; }
__device__ __forceinline__ uint4 ld16g(const void* q) { return *reinterpret_cast<const uint4*>(q); }
__device__ __forceinline__ float4 ldf4g(const void* q) { return *reinterpret_cast<const float4*>(q); }
#endif

struct PpStage {            // wave-uniform position in the step stream
    int tile, c;
    int n, y0, x0, n0, g, txy;
};

struct PpGeo {
    int ncb, tiles_x, tiles_xy, nch, tile_end;
    int cb_slow, n_img;     // cb_slow: channel block is the slowest tile coordinate (register-resident BN sums)
    int th, tw;             // pixel tile (conv_device.h TileGeo: 8 x 32, or 10 x 25 flattened onto the 256 MFMA rows)
};

// what a thread needs to know about its 16-byte channel slot of chunk c (one row per (chunk, slot), built once in LDS
// from the kernel-argument descriptors so that the step loop never touches them)
struct __attribute__((aligned(16))) PpSlot {
    const bf16_t* ptr;      // source tensor + channel offset
    const float* scale;     // coefficient rows + channel offset (a valid dummy for RAW sources)
    const float* shift;
    int C, n_off;
    float slope;            // activation slope, 1 = none
    int g_fixed;
    int flags;              // 1: live (channel < Cin), 2: RAW (coefficients 1 / 0)
    int pad_;
};
static_assert(sizeof(PpSlot) == 48, "PpSlot is read as three 16-byte LDS vectors");

__device__ __forceinline__ void pp_decode(PpStage& s, const PpGeo& q, const GroupMap& gm) {
    const int per_cb = q.n_img * q.tiles_xy;
    const int cb = q.cb_slow ? s.tile / per_cb : s.tile % q.ncb;
    const int r = q.cb_slow ? s.tile % per_cb : s.tile / q.ncb;
    s.txy = r % q.tiles_xy;
    s.n = r / q.tiles_xy;
    s.y0 = (s.txy / q.tiles_x) * q.th;
    s.x0 = (s.txy % q.tiles_x) * q.tw;
    s.n0 = cb * PP_NT;
    s.g = group_of(gm, s.n);
}

// next step of the stream; the last step repeats (its loads stay valid, its results are never used)
__device__ __forceinline__ void pp_advance(PpStage& s, const PpGeo& q, const GroupMap& gm) {
    if (s.c + 1 < q.nch) {
        s.c += 1;
    } else if (s.tile + 1 < q.tile_end) {
        s.c = 0;
        s.tile += 1;
        pp_decode(s, q, gm);
    }
}

// conv_ws_kernel's tile order (cb_slow: channel block, then image, then tile row, then tile column): the next tile is one carry chain
// away.  The division form above is ~150 dependent scalar instructions per call -- three signed divisions by run-time values, a
// 16-way group search, reciprocals parked in VGPR lanes -- and both roles ran it at EVERY tile end, in front of the barrier, with
// nothing to overlap it: 1 000-2 000 cycles of a 12 000 (forward) / 20 000 (gradient) cycle tile (profiles/r06_ws_trace.txt, stamps
// "advance" and "shift").  pp_decode remains for the first tile of a workgroup.
__device__ __forceinline__ void ws_advance(PpStage& s, const PpGeo& q, const GroupMap& gm) {
    if (s.c + 1 < q.nch) {
        s.c += 1;
    } else if (s.tile + 1 < q.tile_end) {
        s.c = 0;
        s.tile += 1;
        s.txy += 1;
        s.x0 += q.tw;
        if (s.x0 >= q.tiles_x * q.tw) {
            s.x0 = 0;
            s.y0 += q.th;
            if (s.txy == q.tiles_xy) {
                s.txy = 0;
                s.y0 = 0;
                s.n += 1;
                if (s.n == q.n_img) {
                    s.n = 0;
                    s.n0 += PP_NT;
                }
                s.g = group_of(gm, s.n);
            }
        }
    }
}

// The epilogue is the LDS-staged one of conv_epilogue.h (any destination kind).  Launches that qualify for a register
// epilogue (forward, gradient into plain tensors) go to conv_ws_kernel below instead.
__global__ __launch_bounds__(256, 1) void conv_pp_kernel(const rd_conv_t p, int tiles_total, const rdfin::FinArg fa) {
    rdfin::prologue(fa);                                // BatchNorm finalize folded into this launch (bn_fin.h)
    typedef bf16_t T;
    constexpr int S = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);
    PpGeo q;
    q.ncb = p.CoutPad / PP_NT;
    q.tiles_x = (W + TW - 1) / TW;
    q.tiles_xy = q.tiles_x * ((H + TH - 1) / TH);
    q.nch = p.CinPad / 32;
    q.cb_slow = 0;
    q.n_img = p.N;
    q.th = TH;
    q.tw = TW;
    const int tile_begin = (int)((long long)blockIdx.x * tiles_total / gridDim.x);
    q.tile_end = (int)((long long)(blockIdx.x + 1) * tiles_total / gridDim.x);
    const int nsteps = (q.tile_end - tile_begin) * q.nch;
    if (nsteps <= 0) return;

    // ---- slot table
    if (tid < q.nch * 4) {
        const int c = (tid >> 2) * 32 + (tid & 3) * S;
        const int si = (p.nsrc == 1 || c < p.src[0].C) ? 0 : 1;
        const rd_src_t sd = select_src(p.src, si);
        const bool live = c < p.Cin, rawm = sd.mode == RD_SRC_RAW;
        const int cc = live ? c - (si ? p.src[0].C : 0) : 0;
        PpSlot e;
        e.ptr = reinterpret_cast<const T*>(sd.ptr) + cc;
        e.scale = rawm ? reinterpret_cast<const float*>(p.w) : sd.scale + cc;
        e.shift = rawm ? reinterpret_cast<const float*>(p.w) : sd.shift + cc;
        e.C = sd.C;
        e.n_off = sd.n_off;
        e.slope = sd.mode == RD_SRC_AFFACT ? sd.slope : 1.f;
        e.g_fixed = rawm ? 0 : sd.g_fixed;
        e.flags = (live ? 1 : 0) | (rawm ? 2 : 0);
        e.pad_ = 0;
        reinterpret_cast<PpSlot*>(smem + PP_TAB)[tid] = e;
    }
    __syncthreads();

    // ---- per-thread constants of the fill: channel slot sw of a chunk, halo items, LDS byte addresses
    const int sw = tid & 3, nn_w = tid >> 2;
    int it_yx[PP_NIT], it_lds[PP_NIT];
#pragma unroll
    for (int b = 0; b < PP_NIT; ++b) {
        const int pix = (tid >> 2) + 64 * b;
        const int py = pix / PP_PW;
        it_yx[b] = (py << 16) | (pix - py * PP_PW);
        it_lds[b] = sw * PP_PLANE_A + (pix < PP_NPIX ? pix : PP_NPIX) * 16;
    }
    const int w_lds = PP_W0 + sw * PP_PLANE_W + nn_w * 16;                          // + tap * 1024 + parity * PP_W_BYTES
    const T* wbase = reinterpret_cast<const T*>(p.w) + nn_w * 32 + sw * S;      // packed [chunk][tap][CoutPad][32]
    const int CoutPad = p.CoutPad;
    // MFMA fragment bases: lane (li, h) reads entry li (+ immediate) of plane ks*2 + h
    const int a_base = h * PP_PLANE_A + (wave * 2 * PP_PW + li) * 16;
    const int b_base = PP_W0 + h * PP_PLANE_W + li * 16;

    // ---- pipeline registers
    uint4 raw[PP_NIT] = {}, wr[PP_WIT] = {};
    float scC[S], shC[S], scL[S], shL[S];
    float slopeC = 1.f, slopeL = 1.f;
    bool liveC = false, liveL = false;
    const T* ldbase = nullptr;          // stage L: this thread's channel slot of image n, pixel (0,0)
    int ldC = 0;

    PpStage L, Cs, M;
    L.tile = tile_begin;
    L.c = 0;
    pp_decode(L, q, gm);

    // stage L set-up: source / channel of this thread's slot, BN coefficients (consumed one step later)
    auto begin_issue = [&]() {
        const PpSlot e = reinterpret_cast<const PpSlot*>(smem + PP_TAB)[L.c * 4 + sw];
        liveL = (e.flags & 1) != 0;
        const bool rawm = (e.flags & 2) != 0;
        ldC = e.C;
        ldbase = e.ptr + (size_t)(L.n + e.n_off) * H * W * e.C;
        slopeL = e.slope;
        const int g = e.g_fixed >= 0 ? e.g_fixed : L.g;
        const float* sp = e.scale + (rawm ? 0 : g * e.C);
        const float* hp = e.shift + (rawm ? 0 : g * e.C);
        const float4 a0 = ldf4g(sp), a1 = ldf4g(sp + 4), b0 = ldf4g(hp), b1 = ldf4g(hp + 4);
        scL[0] = a0.x; scL[1] = a0.y; scL[2] = a0.z; scL[3] = a0.w; scL[4] = a1.x; scL[5] = a1.y; scL[6] = a1.z; scL[7] = a1.w;
        shL[0] = b0.x; shL[1] = b0.y; shL[2] = b0.z; shL[3] = b0.w; shL[4] = b1.x; shL[5] = b1.y; shL[6] = b1.z; shL[7] = b1.w;
#pragma unroll
        for (int e2 = 0; e2 < S; ++e2) {
            scL[e2] = rawm ? 1.f : scL[e2];
            shL[e2] = rawm ? 0.f : shL[e2];
        }
    };
    auto issue_item = [&](int b) {
        const int y = min(max(L.y0 - 1 + (it_yx[b] >> 16), 0), H - 1), x = min(max(L.x0 - 1 + (it_yx[b] & 0xffff), 0), W - 1);
        raw[b] = ld16g(ldbase + (unsigned)((y * W + x) * ldC));
    };
    auto issue_w = [&](int t) {
        wr[t] = ld16(wbase + ((size_t)(L.c * 9 + t) * CoutPad + L.n0) * 32);
    };
    auto consume_item = [&](int b, int par) {
        const int y = Cs.y0 - 1 + (it_yx[b] >> 16), x = Cs.x0 - 1 + (it_yx[b] & 0xffff);
        const bool in = liveC & ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
        float v[S];
        Slot<T>::unpack(raw[b], v);
#pragma unroll
        for (int e = 0; e < S; ++e) v[e] = act_fn(scC[e] * v[e] + shC[e], slopeC);
        const uint4 u = Slot<T>::pack(v);
        *reinterpret_cast<uint4*>(smem + par * PP_IN_BYTES + it_lds[b]) = in ? u : make_uint4(0, 0, 0, 0);
    };
    auto consume_w = [&](int t, int par) {
        *reinterpret_cast<uint4*>(smem + par * PP_W_BYTES + w_lds + t * (PP_NT * 16)) = wr[t];
    };
    auto shift_stages = [&]() {          // C <- L, L <- next step
        Cs = L;
        liveC = liveL;
        slopeC = slopeL;
#pragma unroll
        for (int e = 0; e < S; ++e) { scC[e] = scL[e]; shC[e] = shL[e]; }
        pp_advance(L, q, gm);
    };

    // ---- MFMA operand fragments: A = 32 pixels x 16 channels, B = 32 output channels x 16 channels
    struct Frag { uint4 a[2][2], b[2][2]; };
    Frag fr[2];
    auto load_frags = [&](int tap, int par, Frag& f) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const char* s_a = smem + par * PP_IN_BYTES + a_base;
        const char* s_b = smem + par * PP_W_BYTES + b_base;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                f.a[mb][ks] = *reinterpret_cast<const uint4*>(s_a + ks * 2 * PP_PLANE_A + ((mb + kh) * PP_PW + kw) * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                f.b[nb][ks] = *reinterpret_cast<const uint4*>(s_b + ks * 2 * PP_PLANE_W + (tap * PP_NT + nb * 32) * 16);
    };

    f32x16 acc[2][2];
    // ---- prologue: step 0 into buffer 0, step 1 requested
    begin_issue();
#pragma unroll
    for (int b = 0; b < PP_NIT; ++b) issue_item(b);
#pragma unroll
    for (int t = 0; t < PP_WIT; ++t) issue_w(t);
    shift_stages();                      // C = step 0, L = step 1
    begin_issue();
#pragma unroll
    for (int b = 0; b < PP_NIT; ++b) { consume_item(b, 0); issue_item(b); }
#pragma unroll
    for (int t = 0; t < PP_WIT; ++t) { consume_w(t, 0); issue_w(t); }
    M = Cs;
    shift_stages();                      // C = step 1, L = step 2
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const int par = s & 1, nxt = par ^ 1;
        if (M.c == 0) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        }
        load_frags(0, par, fr[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_frags(t + 1, par, fr[(t + 1) & 1]);
            const Frag& f = fr[t & 1];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[mb][ks]),
                                                                              __builtin_bit_cast(bf16x8, f.b[nb][ks]), acc[mb][nb], 0, 0, 0);
            // fill of step s+1 / requests of step s+2, one piece per tap
            if (t == 0) begin_issue();
            if (t >= 3) {
                consume_item(t - 3, nxt);
                issue_item(t - 3);
            }
            consume_w(t, nxt);
            issue_w(t);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (M.c == q.nch - 1) {
            const int slot = (M.txy + 7 * M.n) % rd_stat_nslots(p.stat_slots);
            conv_epilogue<T, 2>(p, acc, smem + PP_EPI, tid, M.n, M.g, M.y0, M.x0, M.n0, slot);
        }
        M = Cs;
        shift_stages();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------ warp-specialised variant
// conv_pp_kernel above is INSTRUCTION-ISSUE bound: one wave per SIMD carries ~13 non-MFMA instructions per MFMA (fill
// transform, LDS fragment reads, address arithmetic), of which an in-order wave hides ~5 behind a 32-cycle
// v_mfma_f32_32x32x16_bf16 (SQ counters, profiles/README.md round 2: 51% of the wave cycles issuing, 18% issue-stalled,
// matrix pipe 26% busy).  Here the same step stream is split over TWO waves per SIMD with different jobs:
//   waves 0-3 (MFMA waves):   LDS fragment reads + the 72 MFMAs of a step + the register epilogue  (~1.5 instr / MFMA)
//   waves 4-7 (loader waves): global loads of step s+2, BN / activation transform + LDS fill of step s+1, weights
// so that the VALU work of the fill issues from another wave while the matrix pipe runs.  Same LDS planes, same tile
// order and the same arithmetic as conv_pp_kernel (bit-identical results); one barrier per step for all eight waves.
// MODE 1 = forward, MODE 2 = gradient into plain destinations (the register epilogues; nothing else is launched here).
// The loader's per-item cost is what bounds a step (s_memtime traces, scripts/ws_trace.py), so it is kept lean: halo
// offsets and bounds are computed once per TILE, the affine runs as v_pk_fma_f32, ReLU as v_pk_max_i16 on the packed
// bf16 pair (identical to ReLU before the rounding), raw sources are copied without touching the VALU.
// LDS map of conv_ws_kernel: two input buffers of four planes [slot][halo pixel at row pitch PL], two weight buffers (the planes of
// conv_pp_kernel), TAB (MODE 1: bias [CoutPad]; MODE 2: [G][2][CoutPad] producer scale | shift), the channel-slot table.
// Shape 1 (10 x 25 tiles) stores its 12 x 27 halo tile at pitch 41 (conv_device.h TileGeo::LdsPitch: the fragment reads of an MFMA
// row block that straddles tile rows stay conflict-free); the room comes from the staged-epilogue area this kernel never had a use for.
template <int TS> struct WsLds {
    static constexpr int PL = TileGeo<TS>::LdsPitch(1);
    static constexpr int NENT = (TileGeo<TS>::H + 2) * PL;                 // entries per plane (+ 4 dummies: stores of non-items)
    static constexpr int PLANE_A = (NENT + 4) * 16 + 32;
    static constexpr int IN_BYTES = 4 * PLANE_A;
    static constexpr int W0 = 2 * IN_BYTES;
    static constexpr int TAB = W0 + 2 * PP_W_BYTES;
    static constexpr int TAB_BYTES = 16 * 1024;
    static constexpr int SLOTS = TAB + TAB_BYTES;
    static constexpr int SLOTS2 = SLOTS + PP_MAX_CHUNKS * 4 * 48;         // BWD2: second operand + Q row per (chunk, slot), 16 B each
    static constexpr int SLOTS3 = SLOTS2 + PP_MAX_CHUNKS * 4 * 16;        // SOUT: where the transformed source is written as well, 8 B each
    static constexpr int LDS = SLOTS3 + PP_MAX_CHUNKS * 4 * 8;
};
static_assert(WsLds<0>::LDS <= 160 * 1024 && WsLds<1>::LDS <= 160 * 1024, "conv_ws_kernel: LDS map exceeds the CU");
#ifdef RD_DEBUG_SWITCHES
__device__ unsigned long long ws_trace[2][64][4];          // [role][step][event] shader-clock stamps of workgroup 5 (debug build)
#define WS_T(role, s, ev) do { if (trace_ && blockIdx.x == 5 && tid == 0 && (s) < 63) ws_trace[role][s][ev] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long ws_wg[256][2];               // [workgroup][entry, exit] on the 100 MHz clock: who starts late when the lanes share the device
__device__ unsigned long long ws_fine[2][16][16];         // [role][step][event]: stamps INSIDE a step (they cost a scalar-memory wait each)
#define WS_F(role, s, ev) do { if (trace_ && blockIdx.x == 5 && tid == 0 && (s) < 16) ws_fine[role][s][ev] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_T(role, s, ev) do { } while (0)
#define WS_F(role, s, ev) do { } while (0)
#endif
#if defined(RD_DEBUG_SWITCHES) && defined(RD_WS_EXP)         // timing experiments (results wrong): -DRD_WS_EXP + RD_CONV_WS_EXP=bits
#define WS_EXP(bit) ((exp_ & (bit)) != 0)
#else
#define WS_EXP(bit) false
#endif

// BWD2 (MODE 2 only): the source is a BatchNorm-backward pair (g, z) and the loader forms dz = P g + Q z + R while it stages the tile
// -- two loads and two fused multiply-adds per vector on waves that otherwise wait at the barrier for half of every tile -- instead of
// reading a dz tensor that a separate rd_bn_apply launch stored (round 4: 15 elementwise launches and 0.86 GB of traffic per step)
// SOUT (rd_src_t.out): the loader waves also WRITE what they stage -- act(bn(z)) of a forward launch, dz of a gradient launch -- to a
// tensor of the source's shape, interior pixels of every tile once (first output-channel block only): the weight gradient of the same
// layer then reads both of its operands as stored tensors and its loader, which is that kernel's pole, only copies (wgrad.hip).  The
// stores are unconditional (items outside the tile interior / the image, sources without an `out`: a per-lane trash record) so that
// the loader's vector-memory operations stay countable.
__device__ uint4 ws_trash[256];
template <int MODE, int TS, bool BWD2 = false, bool SOUT = false>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(const rd_conv_t p, int tiles_total, const rdfin::FinArg fa) {
    // BatchNorm finalize folded into this launch (bn_fin.h), global path: the loader waves fetch the coefficients with global loads
    // every step (ldf4g below: not flat loads, which would tie up the LDS counter), so the vectors must be in memory
#ifdef RD_DEBUG_SWITCHES
    const unsigned long long t_entry_ = __builtin_readcyclecounter();
    const unsigned long long w_entry_ = wall_clock64();
#endif
    rdfin::prologue(fa);
#ifdef RD_DEBUG_SWITCHES
    const unsigned long long t_fin_ = __builtin_readcyclecounter();
#endif
    typedef bf16_t T;
    constexpr int S = 8;
    static_assert(MODE == 1 || MODE == 2, "register epilogues only");
    static_assert(!BWD2 || MODE == 2, "two-operand sources exist in the gradient mode only");
    // TS 1: 10 x 25 pixel tiles, flattened row-major onto the 256 MFMA rows (250 live); the 12 x 27 halo tile fits the planes of
    // the 10 x 34 one, only the per-lane pixel of an MFMA row and the halo row pitch change
    constexpr int THt = TileGeo<TS>::H, TWt = TileGeo<TS>::W, PWt = TWt + 2, NPIXt = (THt + 2) * PWt;
    static_assert(64 * PP_NIT >= NPIXt, "one halo item per thread and pass");
    // this kernel's own LDS map (WsLds<TS>): the planes hold the halo tile at the conflict-free row pitch of its shape
    typedef WsLds<TS> LM;
    constexpr int PLt = LM::PL, WS_PLANE_A = LM::PLANE_A, WS_IN_BYTES = LM::IN_BYTES, WS_W0 = LM::W0, WS_TAB = LM::TAB, WS_SLOTS = LM::SLOTS;
    constexpr int WS_SLOTS2 = LM::SLOTS2, WS_SLOTS3 = LM::SLOTS3;
    // who stages the weight chunk of the next step: the MFMA waves in the forward mode (they have ~1.5 issue slots per MFMA
    // gap to spare and 36 registers of headroom; the loader's nine weight vectors cost 1000-3000 cycles per step on top of
    // its ~3000 for the halo items against ~3300 of MFMAs: traces in profiles/r02_ws_trace.txt), the loader waves in the
    // gradient mode (its MFMA waves are at the register limit, and its loader mostly copies raw dz vectors)
    constexpr bool WMFMA = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // 0: MFMA waves, 1: loader waves
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const GroupMap gm = make_gm(p.gstart, p.G);
#ifdef RD_DEBUG_SWITCHES
    const int exp_ = tiles_total >> 26;                    // WS_EXP bits
    const bool trace_ = ((tiles_total >> 25) & 1) != 0;    // this launch records the s_memtime trace
    tiles_total &= (1 << 25) - 1;
#endif
    PpGeo q;
    q.ncb = p.CoutPad / PP_NT;
    q.tiles_x = (W + TWt - 1) / TWt;
    q.tiles_xy = q.tiles_x * ((H + THt - 1) / THt);
    q.nch = p.CinPad / 32;
    q.cb_slow = 1;
    q.n_img = p.N;
    q.th = THt;
    q.tw = TWt;
    const int tile_begin = (int)((long long)blockIdx.x * tiles_total / gridDim.x);
    q.tile_end = (int)((long long)(blockIdx.x + 1) * tiles_total / gridDim.x);
    const int nsteps = (q.tile_end - tile_begin) * q.nch;
    if (nsteps <= 0) return;

    // ---- tables: channel-slot descriptors of every chunk, bias / producer coefficients of every output channel
    if ((int)threadIdx.x < q.nch * 4) {
        const int t = threadIdx.x;
        const int c = (t >> 2) * 32 + (t & 3) * S;
        const int si = (p.nsrc == 1 || c < p.src[0].C) ? 0 : 1;
        const rd_src_t sd = select_src(p.src, si);
        const bool live = c < p.Cin, rawm = sd.mode == RD_SRC_RAW;
        const int cc = live ? c - (si ? p.src[0].C : 0) : 0;
        PpSlot e;
        e.ptr = reinterpret_cast<const T*>(sd.ptr) + cc;
        e.scale = rawm ? reinterpret_cast<const float*>(p.w) : sd.scale + cc;
        e.shift = rawm ? reinterpret_cast<const float*>(p.w) : sd.shift + cc;
        e.C = sd.C;
        e.n_off = sd.n_off;
        e.slope = sd.mode == RD_SRC_AFFACT ? sd.slope : 1.f;
        e.g_fixed = rawm ? 0 : sd.g_fixed;
        e.flags = (live ? 1 : 0) | (rawm ? 2 : 0);
        e.pad_ = 0;
        reinterpret_cast<PpSlot*>(smem + WS_SLOTS)[t] = e;
        if constexpr (BWD2) {
            const bool bwd = sd.mode == RD_SRC_BNBWD;
            struct { const T* ptr2; const float* q; } e2;
            e2.ptr2 = bwd ? reinterpret_cast<const T*>(sd.ptr2) + cc : e.ptr;                 // a plain source aliases its own tensor, Q = 0
            e2.q = bwd ? sd.q + cc : nullptr;
            *reinterpret_cast<decltype(e2)*>(smem + WS_SLOTS2 + t * 16) = e2;
        }
        if constexpr (SOUT) *reinterpret_cast<T**>(smem + WS_SLOTS3 + t * 8) = (live && sd.out) ? reinterpret_cast<T*>(sd.out) + cc : nullptr;
    }
    {
        float* tab = reinterpret_cast<float*>(smem + WS_TAB);
        if constexpr (MODE == 1) {
            for (int c = threadIdx.x; c < p.CoutPad; c += 512) tab[c] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
        } else {
            for (int i = threadIdx.x; i < p.G * p.CoutPad; i += 512) {
                const int g = i / p.CoutPad, c = i - g * p.CoutPad;
                const int di = c >= p.c_split ? 1 : 0;
                const rd_dst_t d = select_dst(p, di);
                const int cd = c - (di ? p.c_split : 0);
                const bool ok = c < p.Cout && d.kind != RD_DST_NONE && d.scale != nullptr && cd < d.Cd;
                const int gd = d.g_fixed >= 0 ? d.g_fixed : g;
                tab[(g * 2 + 0) * p.CoutPad + c] = ok ? d.scale[gd * d.Cd + cd] : 1.f;
                tab[(g * 2 + 1) * p.CoutPad + c] = ok ? d.shift[gd * d.Cd + cd] : 0.f;
            }
        }
    }
    __syncthreads();

    if (role == 1) {
        // =============================================================================== loader waves
        // (s_setprio 1 for this half, the younger one of the workgroup: gradient launches +6 %, the step 3.998 -> 4.017 ms; for the
        // MFMA half: no change -- docs/experiments.md, round 6)
        const int sw = tid & 3, nn_w = tid >> 2;
        int it_yx[PP_NIT], it_lds[PP_NIT];
#pragma unroll
        for (int b = 0; b < PP_NIT; ++b) {
            const int pix = (tid >> 2) + 64 * b;
            const int py = pix / PWt;
            it_yx[b] = (py << 16) | (pix - py * PWt);
            it_lds[b] = sw * WS_PLANE_A + (pix < NPIXt ? py * PLt + (pix - py * PWt) : LM::NENT) * 16;
        }
        const int w_lds = WS_W0 + sw * PP_PLANE_W + nn_w * 16;
        const T* wbase = reinterpret_cast<const T*>(p.w) + nn_w * 32 + sw * S;
        const size_t w_tap = (size_t)p.CoutPad * 32;            // elements between the taps of one chunk
        uint4 raw[PP_NIT] = {}, wr[PP_WIT] = {};
        uint4 raw2[BWD2 ? PP_NIT : 1] = {};                  // BWD2: the second operand (z) of every halo item
        float scC[S], shC[S], scL[S], shL[S];
        float qC[BWD2 ? S : 1] = {}, qL[BWD2 ? S : 1] = {};
        float slopeC = 1.f, slopeL = 1.f;
        bool copyC = false, copyL = false;   // raw source without activation: the 16 bytes go to LDS as they are
        const T* ldbase = nullptr;
        const T* ldbase2 = nullptr;
        int ldC = 0;
        int offL[PP_NIT];                    // stage L's tile: pixel index (clamped into the image) of each halo item
        int inL = 0, inC = 0;                // bit b: halo item b lies inside the image (and this slot carries a channel)
        // SOUT: stage C's pixel indices / channel stride / destination, and which of this thread's items are tile-interior pixels
        int offC[SOUT ? PP_NIT : 1] = {};
        int stCs = 0, intr = 0;
        T* stL = nullptr;
        T* stC = nullptr;
        if constexpr (SOUT) {
#pragma unroll
            for (int b = 0; b < PP_NIT; ++b) {
                const int py = it_yx[b] >> 16, px = it_yx[b] & 0xffff;
                intr |= (py >= 1 && py <= THt && px >= 1 && px <= TWt) ? (1 << b) : 0;
            }
        }
        PpStage L, Cs;
        L.tile = tile_begin;
        L.c = 0;
        pp_decode(L, q, gm);
        auto tile_geom = [&]() {
            inL = 0;
#pragma unroll
            for (int b = 0; b < PP_NIT; ++b) {
                const int y = L.y0 - 1 + (it_yx[b] >> 16), x = L.x0 - 1 + (it_yx[b] & 0xffff);
                inL |= (((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W)) ? (1 << b) : 0;
                offL[b] = min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
            }
        };
        int maskL = 0;
        auto begin_issue = [&]() {
            const PpSlot e = reinterpret_cast<const PpSlot*>(smem + WS_SLOTS)[L.c * 4 + sw];
            const bool live = (e.flags & 1) != 0, rawm = (e.flags & 2) != 0;
            maskL = live ? inL : 0;
            ldC = e.C;
            ldbase = e.ptr + (size_t)(L.n + e.n_off) * H * W * e.C;
            slopeL = e.slope;
            copyL = rawm && e.slope == 1.f;
            if constexpr (BWD2) {
                struct E2 { const T* ptr2; const float* q; };
                const E2 e2 = *reinterpret_cast<const E2*>(smem + WS_SLOTS2 + (L.c * 4 + sw) * 16);
                ldbase2 = e2.ptr2 + (size_t)(L.n + e.n_off) * H * W * e.C;
                const bool hasq = e2.q != nullptr;
                const float* qp = hasq ? e2.q + (e.g_fixed >= 0 ? e.g_fixed : L.g) * e.C : reinterpret_cast<const float*>(p.w);
                const float4 q0 = ldf4g(qp), q1 = ldf4g(qp + 4);
                qL[0] = q0.x; qL[1] = q0.y; qL[2] = q0.z; qL[3] = q0.w; qL[4] = q1.x; qL[5] = q1.y; qL[6] = q1.z; qL[7] = q1.w;
#pragma unroll
                for (int e3 = 0; e3 < S; ++e3) qL[e3] = hasq ? qL[e3] : 0.f;
                copyL = copyL && !hasq;
            }
            if constexpr (SOUT) {
                T* o = *reinterpret_cast<T* const*>(smem + WS_SLOTS3 + (L.c * 4 + sw) * 8);
                stL = (o != nullptr && L.n0 == 0) ? o + (size_t)(L.n + e.n_off) * H * W * e.C : nullptr;
            }
            const int g = e.g_fixed >= 0 ? e.g_fixed : L.g;
            const float* sp = e.scale + (rawm ? 0 : g * e.C);
            const float* hp = e.shift + (rawm ? 0 : g * e.C);
            const float4 a0 = ldf4g(sp), a1 = ldf4g(sp + 4), b0 = ldf4g(hp), b1 = ldf4g(hp + 4);
            scL[0] = a0.x; scL[1] = a0.y; scL[2] = a0.z; scL[3] = a0.w; scL[4] = a1.x; scL[5] = a1.y; scL[6] = a1.z; scL[7] = a1.w;
            shL[0] = b0.x; shL[1] = b0.y; shL[2] = b0.z; shL[3] = b0.w; shL[4] = b1.x; shL[5] = b1.y; shL[6] = b1.z; shL[7] = b1.w;
#pragma unroll
            for (int e2 = 0; e2 < S; ++e2) {
                scL[e2] = rawm ? 1.f : scL[e2];
                shL[e2] = rawm ? 0.f : shL[e2];
            }
        };
        auto issue_item = [&](int b) {
            if (!WS_EXP(2)) raw[b] = ld16g(ldbase + (size_t)(unsigned)offL[b] * (unsigned)ldC);
            if constexpr (BWD2) raw2[b] = ld16g(ldbase2 + (size_t)(unsigned)offL[b] * (unsigned)ldC);
        };
        const T* wptr = nullptr;
        auto begin_w = [&]() { wptr = wbase + ((size_t)(L.c * 9) * p.CoutPad + L.n0) * 32; };
        auto issue_w = [&](int t) {
            if (!WS_EXP(1)) wr[t] = ld16(wptr + t * w_tap);
        };
        int maskC = 0;
        // XFORM is wave-uniform (a ballot over the wave's slots): false = every slot of this step is a plain copy
        auto consume_item = [&](int b, int par, bool xform) {
            uint4 u = raw[b];
            if (xform) {
                float v[S];
                Slot<T>::unpack(u, v);
                if constexpr (BWD2) {
                    float z2[S];
                    Slot<T>::unpack(raw2[b], z2);
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = bn_bwd_value(scC[e], v[e], qC[e], z2[e], shC[e]);      // dz = P g + Q z + R
                } else {
#pragma unroll
                    for (int e = 0; e < S; ++e) v[e] = act_fn(scC[e] * v[e] + shC[e], slopeC);
                }
                u = Slot<T>::pack(v);
            }
            const bool in = ((maskC >> b) & 1) != 0;
            *reinterpret_cast<uint4*>(smem + par * WS_IN_BYTES + it_lds[b]) = in ? u : make_uint4(0, 0, 0, 0);
            if constexpr (SOUT) {
                const bool st = in && ((intr >> b) & 1) != 0 && stC != nullptr;
                T* dp = st ? stC + (size_t)(unsigned)offC[b] * (unsigned)stCs : reinterpret_cast<T*>(ws_trash + tid);
                st16g(dp, u);
            }
        };
        auto consume_w = [&](int t, int par) {
            *reinterpret_cast<uint4*>(smem + par * PP_W_BYTES + w_lds + t * (PP_NT * 16)) = wr[t];
        };
        auto shift_stages = [&]() {
            Cs = L;
            maskC = maskL;
            slopeC = slopeL;
            copyC = copyL;
#pragma unroll
            for (int e = 0; e < S; ++e) { scC[e] = scL[e]; shC[e] = shL[e]; }
            if constexpr (BWD2) {
#pragma unroll
                for (int e = 0; e < S; ++e) qC[e] = qL[e];
            }
            if constexpr (SOUT) {
                stC = stL;
                stCs = ldC;
#pragma unroll
                for (int b = 0; b < PP_NIT; ++b) offC[b] = offL[b];
            }
            const int tile0 = L.tile;
            ws_advance(L, q, gm);
            if (L.tile != tile0) tile_geom();
        };
        // prologue: step 0 into buffer 0, step 1 requested
        tile_geom();
        begin_issue();
        begin_w();
#pragma unroll
        for (int b = 0; b < PP_NIT; ++b) issue_item(b);
        if constexpr (!WMFMA) {
#pragma unroll
            for (int t = 0; t < PP_WIT; ++t) wr[t] = ld16(wptr + t * w_tap);
        }
        shift_stages();
        begin_issue();
        begin_w();
#pragma unroll
        for (int b = 0; b < PP_NIT; ++b) { consume_item(b, 0, true); issue_item(b); }
        if constexpr (!WMFMA) {
#pragma unroll
            for (int t = 0; t < PP_WIT; ++t) { consume_w(t, 0); wr[t] = ld16(wptr + t * w_tap); }
        }
        shift_stages();
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int nxt = (s & 1) ^ 1;
            if (WS_EXP(8)) { __syncthreads(); continue; }
            WS_T(1, s, 0);
            WS_F(1, s, 0);
            begin_issue();
            begin_w();
            WS_F(1, s, 1);
            if (__builtin_amdgcn_ballot_w64(!copyC) != 0) {
#pragma unroll
                for (int b = 0; b < PP_NIT; ++b) { consume_item(b, nxt, true); issue_item(b); WS_F(1, s, 2 + b); }
            } else {
#pragma unroll
                for (int b = 0; b < PP_NIT; ++b) { consume_item(b, nxt, false); issue_item(b); }
            }
            WS_T(1, s, 1);
            if constexpr (!WMFMA) {
#pragma unroll
                for (int t = 0; t < PP_WIT; ++t) { consume_w(t, nxt); issue_w(t); }
            }
            WS_T(1, s, 2);
            WS_F(1, s, 8);
            shift_stages();
            WS_F(1, s, 9);
            __syncthreads();
            WS_T(1, s, 3);
            WS_F(1, s, 10);
        }
        return;
    }

    // =================================================================================== MFMA waves
    // this lane's pixel of the wave's two MFMA row blocks: tile row / column, LDS offset of its halo origin
    int px_r[2], px_c[2], a_off[2];
    bool px_live[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        tile_pixel<TS>(wave * 2 + mb, li, px_r[mb], px_c[mb], px_live[mb]);
        a_off[mb] = h * WS_PLANE_A + (px_r[mb] * PLt + px_c[mb]) * 16;
    }
    const int b_base = WS_W0 + h * PP_PLANE_W + li * 16;
    const float* tab = reinterpret_cast<const float*>(smem + WS_TAB);
    PpStage M;
    M.tile = tile_begin;
    M.c = 0;
    pp_decode(M, q, gm);
    // WMFMA: this wave's share of the weight chunk (thread -> row nn_w, channel slot sw, all nine taps), one step ahead in LDS
    uint4 wq[PP_WIT] = {};
    PpStage Lw = M;                                        // the step whose weights are in flight
    const int sw_w = tid & 3, nn_w = tid >> 2;
    const int w_lds = WS_W0 + sw_w * PP_PLANE_W + nn_w * 16;
    const T* wbase = reinterpret_cast<const T*>(p.w) + nn_w * 32 + sw_w * S;
    const size_t w_tap = (size_t)p.CoutPad * 32;
    auto w_issue = [&](int t) { wq[t] = ld16(wbase + ((size_t)(Lw.c * 9) * p.CoutPad + Lw.n0) * 32 + t * w_tap); };
    auto w_store = [&](int t, int par) { *reinterpret_cast<uint4*>(smem + par * PP_W_BYTES + w_lds + t * (PP_NT * 16)) = wq[t]; };
    if constexpr (WMFMA) {
#pragma unroll
        for (int t = 0; t < PP_WIT; ++t) w_issue(t);       // step 0
#pragma unroll
        for (int t = 0; t < PP_WIT; ++t) w_store(t, 0);    // step 0 -> buffer 0
        ws_advance(Lw, q, gm);                             // Lw = step 1: what the first loop iteration requests AND stores
    }

    // operand fragments of one (tap, k-step) group: A = 2 x (32 pixels x 16 channels), B = 2 x (32 outputs x 16 channels);
    // the next group is read while the four MFMAs of this one run (register double buffer at group granularity)
    struct Frag { uint4 a[2], b[2]; };
    Frag fr[2];
    auto load_group = [&](int grp, int par, Frag& f) {
        const int tap = grp >> 1, ks = grp & 1;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const char* s_a = smem + par * WS_IN_BYTES + ks * 2 * WS_PLANE_A;
        const char* s_b = smem + par * PP_W_BYTES + b_base + ks * 2 * PP_PLANE_W;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) f.a[mb] = *reinterpret_cast<const uint4*>(s_a + a_off[mb] + (kh * PLt + kw) * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) f.b[nb] = *reinterpret_cast<const uint4*>(s_b + (tap * PP_NT + nb * 32) * 16);
    };

    f32x16 acc[2][2];
    // per-lane BatchNorm partial sums of the current (channel block, group) run; a lane owns channels
    // n0 + nb*32 + 16v + 8h + e of pixel column li
    float sa[2][2][S], sb[2][2][S];
    int cur_n0 = -1, cur_g = -1;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int e = 0; e < S; ++e) sa[nb][v][e] = sb[nb][v][e] = 0.f;
    auto flush_stats = [&]() {
        if (cur_n0 < 0) return;
        const int slot = blockIdx.x % rd_stat_nslots(p.stat_slots);
        // sums over the 32 pixel lanes of each half-wave (conv_device.h half_wave_sums): afterwards every lane holds BOTH statistics
        // of one channel and adds them to the global fp64 slot itself
        float r[64];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    r[((nb * 2 + v) * S + e) * 2 + 0] = sa[nb][v][e];
                    r[((nb * 2 + v) * S + e) * 2 + 1] = sb[nb][v][e];
                    sa[nb][v][e] = sb[nb][v][e] = 0.f;
                }
        half_wave_sums<64>(r, li);
        const int cq = half_wave_sum_index(li);                            // (nb * 2 + v) * 8 + e
        const int nbq = cq >> 4, vq = (cq >> 3) & 1, eq = cq & 7;
        if constexpr (MODE == 1) {
            if (p.stats) {
                const size_t so = (((size_t)cur_g * RD_STAT_SLOTS + slot) * p.Cout + cur_n0 + nbq * 32 + 16 * vq + 8 * h + eq) * 2;
                atomicAdd(&p.stats[so + 0], (double)r[0]);
                atomicAdd(&p.stats[so + 1], (double)r[1]);
            }
        } else {
            const int c = cur_n0 + nbq * 32 + 16 * vq;
            const int di = c >= p.c_split ? 1 : 0;
            const rd_dst_t d = select_dst(p, di);
            if (d.kind != RD_DST_NONE && d.bstats) {
                const int gd = d.g_fixed >= 0 ? d.g_fixed : cur_g;
                const size_t so = (((size_t)gd * RD_STAT_SLOTS + slot) * d.Cd + c - (di ? p.c_split : 0) + 8 * h + eq) * 2;
                atomicAdd(&d.bstats[so + 0], (double)r[0]);
                atomicAdd(&d.bstats[so + 1], (double)r[1]);
            }
        }
    };
    // MODE 2: the producer's raw tensor of this tile (activation mask, sum g*z), requested at the start of its last K step;
    // the old gradient of an accumulating destination (skip connections: few launches) is read in the epilogue itself --
    // prefetching it too costs 16 more registers, which spill at two waves per SIMD
    uint4 zq[2][2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v) zq[mb][nb][v] = make_uint4(0, 0, 0, 0);
    // Where a lane's vectors of the destination tensors sit: 32-BIT byte offsets from the (wave-uniform) tensor base, one multiply of
    // 24-bit operands each (rd_conv_ws_takes: the images of a destination number < 2^24 pixels and < 4 GB).  Written with size_t
    // indices, each of the sixteen addresses of a tile -- eight producer-tensor requests, eight gradient stores -- was ~35
    // instructions with two 64-bit multiply-adds and two quarter-rate 32-bit multiplies: ~2 000 cycles in front of the last K step's
    // MFMAs and ~2 000 more inside the epilogue of a 20 000-cycle tile (profiles/r06_ws_trace.txt, stamp "pref+g0").  The gradient
    // and the producer tensor of a destination share shape and index, so the epilogue reuses the prefetch's offsets.
    unsigned zoff[2][2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v) zoff[mb][nb][v] = 0u;
    // The eight requests are UNCONDITIONAL (a vector without a producer tensor reads a dummy line): written as `if (d.z) zq = load`, every
    // load sat in its own conditional block and the compiler put `s_waitcnt vmcnt(0)` in front of each one -- eight dependent HBM round
    // trips in front of the last K step's MFMAs (scripts/ws_trace.py on dec.convu2.conv3's gradient: 9 000-10 700 cycles for that step
    // against 3 300 for the others)
    auto dgrad_prefetch = [&]() {
        const char* dummy = reinterpret_cast<const char*>(p.w);
        unsigned pixl[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int y = min(M.y0 + px_r[mb], H - 1), x = min(M.x0 + px_c[mb], W - 1);
            pixl[mb] = __umul24((unsigned)y, (unsigned)W) + (unsigned)x;
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int c0 = M.n0 + nb * 32 + 16 * v;
                const int di = c0 >= p.c_split ? 1 : 0;
                const rd_dst_t d = select_dst(p, di);
                const bool has = d.kind != RD_DST_NONE && d.z != nullptr;
                const unsigned nbase = (unsigned)((M.n + d.n_off) * H * W);
                const unsigned cd = (unsigned)(c0 - (di ? p.c_split : 0)) + 8u * h;
                const char* zb = has ? reinterpret_cast<const char*>(d.z) : dummy;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const unsigned off = (__umul24(nbase + pixl[mb], (unsigned)d.Cd) + cd) * (unsigned)sizeof(T);
                    zoff[mb][nb][v] = off;
                    zq[mb][nb][v] = ld16(zb + (has ? off : 0u));
                }
            }
    };
    // one accumulator block's 16 channels of a lane's pixel as two 8-channel NHWC vectors
    auto regroup = [&](const f32x16& a, int v, float* o) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const HalfSwap r = rd_half_swap(__float_as_uint(a[8 * v + j]), __float_as_uint(a[8 * v + 4 + j]), h);
            o[j] = __uint_as_float(r.r0);
            o[4 + j] = __uint_as_float(r.r1);
        }
    };

#ifdef RD_DEBUG_SWITCHES
    if (trace_ && blockIdx.x == 5 && tid == 0) {
        ws_trace[0][63][0] = __builtin_readcyclecounter(); ws_trace[0][63][1] = wall_clock64();
        ws_fine[0][15][0] = t_entry_; ws_fine[0][15][1] = t_fin_; ws_fine[0][15][2] = ws_trace[0][63][0];      // kernel entry, folded finalize done, tables built
    }
#endif
    __syncthreads();                     // buffer 0 filled (the loader waves' prologue)
#ifdef RD_DEBUG_SWITCHES
    if (trace_ && blockIdx.x == 5 && tid == 0) ws_fine[0][15][3] = __builtin_readcyclecounter();                   // first buffer filled
#endif
    // the accumulators are cleared BEHIND each tile's epilogue (and here for the first tile), not at the head of a tile: the compiler then
    // sees them dead across the epilogue and regroups them in place instead of copying 64 registers per tile first
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    for (int s = 0; s < nsteps; ++s) {
        const int par = s & 1;
        WS_T(0, s, 0);
        WS_F(0, s, 0);
        if constexpr (MODE == 2) {
            if (M.c == q.nch - 1) dgrad_prefetch();
        }
        load_group(0, par, fr[0]);
        WS_F(0, s, 1);
#pragma unroll
        for (int grp = 0; grp < 18; ++grp) {
            if constexpr (WMFMA) {
                // weights of step s+1: requested one tap per group in the first half of the step, written to the other buffer in
                // the second half (an L2 hit has ~9 groups = 1 600 cycles to land).  Round 3 carried them in registers ACROSS the
                // step boundary (requested in step s, stored in step s+1): the loop's back edge then needed all nine vectors in
                // their loop-header registers -- `s_waitcnt vmcnt(8) .. vmcnt(0)` + 18 v_mov_b64 behind every barrier, i.e. every
                // step waited for the loads it had just issued (ISA of conv_ws_kernel<1, *>)
                if (grp < PP_WIT) w_issue(grp);
                else w_store(grp - PP_WIT, par ^ 1);
            }
            if (grp + 1 < 18 && !WS_EXP(64)) load_group(grp + 1, par, fr[(grp + 1) & 1]);
            const Frag& f = fr[grp & 1];
            if (!WS_EXP(4)) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[nb]),
                                                                              __builtin_bit_cast(bf16x8, f.a[mb]), acc[mb][nb], 0, 0, 0);
            }
            // Keep the fragment double buffer REAL: without these the compiler sinks the four ds_read_b128 of group grp+1 below the group
            // boundary, next to their MFMAs (sched_barrier does not order memory operations at instruction selection), and every group
            // waits lgkmcnt(0) twice for reads it has just issued -- 3 100-3 400 cycles per step against 2 304 of MFMA issue.  The
            // memory clobber holds the reads in their group; the group barriers interleave them with its four MFMAs (ISA: W[l1] M r W[l1] M r L M r
            // M r): each read (and the weight load / store of the group) issues in the shadow of a running MFMA instead of all of them
            // behind the fourth one -- forward launches another -1.5...-3.5 % alone, gradient launches -1...-5 %
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x220, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 5) WS_F(0, s, 2);
            if (grp == 11) WS_F(0, s, 3);
        }
        WS_T(0, s, 1);
        WS_F(0, s, 4);
        if (M.c == q.nch - 1) {
            if (M.n0 != cur_n0 || M.g != cur_g) {
                flush_stats();
                cur_n0 = M.n0;
                cur_g = M.g;
            }
            int py[2], pxx[2];
            bool pin[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                py[mb] = M.y0 + px_r[mb];
                pxx[mb] = M.x0 + px_c[mb];
                pin[mb] = px_live[mb] && py[mb] < H && pxx[mb] < W;
            }
            WS_F(0, s, 5);
            if constexpr (MODE == 1) {
                // 32-bit byte offsets from the uniform tensor base (see zoff above): one per tile row of the wave, the eight stores differ by
                // an immediate
                char* outb = reinterpret_cast<char*>(p.out);
                unsigned ob[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    ob[mb] = (__umul24((unsigned)(M.n * H * W) + __umul24((unsigned)py[mb], (unsigned)W) + (unsigned)pxx[mb], (unsigned)p.Cout) +
                              (unsigned)(M.n0 + 8 * h)) * (unsigned)sizeof(T);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        // the bias vector of these 8 channels is read once for both tile rows of the wave
                        const float* bp = tab + M.n0 + nb * 32 + 16 * v + 8 * h;
                        const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                        const float bs[S] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            float o[S];
                            regroup(acc[mb][nb], v, o);
                            if (pin[mb]) {
#pragma unroll
                                for (int e = 0; e < S; ++e) {
                                    if (!WS_EXP(32)) {
                                        sa[nb][v][e] += o[e];          // sums exclude the bias (ramdsir.h, RD_STAT_SLOTS)
                                        sb[nb][v][e] += o[e] * o[e];
                                    }
                                    o[e] += bs[e];
                                }
                                if (!WS_EXP(16)) *reinterpret_cast<uint4*>(outb + ob[mb] + (nb * 32 + 16 * v) * (int)sizeof(T)) = Slot<T>::pack(o);
                            }
                        }
                        WS_F(0, s, 6 + nb * 2 + v);
                    }
            } else {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int di = (M.n0 + nb * 32 + 16 * v) >= p.c_split ? 1 : 0;
                        const rd_dst_t d = select_dst(p, di);
                        if (d.kind == RD_DST_NONE) continue;
                        const float* cp = tab + (size_t)(M.g * 2) * p.CoutPad + M.n0 + nb * 32 + 16 * v + 8 * h;
                        const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                        const float4 d0 = *reinterpret_cast<const float4*>(cp + p.CoutPad), d1 = *reinterpret_cast<const float4*>(cp + p.CoutPad + 4);
                        const float psc[S] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const float psh[S] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                        const float lo = (d.act && d.z) ? d.slope : 1.f;       // one select per element (conv_device.h grad_plain_finish)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            float da[S];
                            regroup(acc[mb][nb], v, da);
                            if (!pin[mb]) continue;
                            // no load in this loop (destinations that accumulate go to conv_pf_kernel: rd_conv_pp_dispatch): with the old
                            // gradient read here (`d.accumulate ? ld16(gp) : 0`), every vector waited vmcnt(0) -- i.e. for the previous
                            // vector's STORE to be acknowledged -- before its own arithmetic: eight serialized round trips per tile
                            // (6 500-11 000 cycles of epilogue against the forward mode's 2 000-3 000)
                            T* gp = reinterpret_cast<T*>(reinterpret_cast<char*>(d.g) + zoff[mb][nb][v]);   // a live pixel: the prefetch's clamp was the identity
                            float z[S], gw[S];
                            Slot<T>::unpack(zq[mb][nb][v], z);
#pragma unroll
                            for (int e = 0; e < S; ++e) {
                                const float m = (z[e] * psc[e] + psh[e]) > 0.f ? 1.f : lo;
                                const float gn = da[e] * m;
                                sa[nb][v][e] += gn;
                                sb[nb][v][e] += gn * (d.z ? z[e] : 0.f);
                                gw[e] = gn;
                            }
                            *reinterpret_cast<uint4*>(gp) = Slot<T>::pack(gw);
                        }
                        WS_F(0, s, 6 + nb * 2 + v);
                    }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
            WS_F(0, s, 10);
        }
        WS_T(0, s, 2);
        ws_advance(M, q, gm);
        if constexpr (WMFMA) ws_advance(Lw, q, gm);
        WS_F(0, s, 11);
        __syncthreads();
        WS_T(0, s, 3);
        WS_F(0, s, 12);
    }
#ifdef RD_DEBUG_SWITCHES
    if (trace_ && blockIdx.x == 5 && tid == 0) { ws_trace[0][63][2] = __builtin_readcyclecounter(); ws_trace[0][63][3] = wall_clock64(); }
#endif
    flush_stats();
#ifdef RD_DEBUG_SWITCHES
    if (trace_ && tid == 0 && blockIdx.x < 256) { ws_wg[blockIdx.x][0] = w_entry_; ws_wg[blockIdx.x][1] = wall_clock64(); }
#endif
}

// plain per-pixel single-operand sources made of whole 16-byte channel slots (the fill above is branch-free)
// 0: not for these kernels; 1: single-operand sources; 2: a BatchNorm-backward source (conv_ws_kernel<2, *, true> only)
int pp_sources_kind(const rd_conv_t& p) {
    int kind = 1;
    for (int i = 0; i < p.nsrc; ++i) {
        const int m = p.src[i].mode;
        if (!(m == RD_SRC_RAW || m == RD_SRC_AFF || m == RD_SRC_AFFACT || m == RD_SRC_BNBWD) || p.src[i].C % 8) return 0;
        if (m != RD_SRC_RAW && (((uintptr_t)p.src[i].scale | (uintptr_t)p.src[i].shift) & 15)) return 0;   // float4 coefficient reads
        if (m == RD_SRC_BNBWD) {
            if (p.emode != 1 || (((uintptr_t)p.src[i].q | (uintptr_t)p.src[i].ptr2) & 15)) return 0;
            kind = 2;
        }
    }
    return p.Cin % 8 == 0 ? kind : 0;
}

}  // namespace

#ifdef RD_DEBUG_SWITCHES
extern "C" int rd_debug_ws_wg(unsigned long long* out) {                // debug library only: 256 x {entry, exit} of the last traced launch
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_wg), sizeof(unsigned long long) * 256 * 2);
}
extern "C" int rd_debug_ws_fine(unsigned long long* out) {              // debug library only: 2 x 16 x 16 stamps inside the steps
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_fine), sizeof(unsigned long long) * 2 * 16 * 16);
}
extern "C" int rd_debug_ws_trace(unsigned long long* out) {             // debug library only: 2 x 64 x 4 stamps
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ws_trace), sizeof(unsigned long long) * 2 * 64 * 4);
}
#endif

// does this launch run on conv_ws_kernel?  ONE definition for the dispatch below and for rd_conv_honours_src_out (conv_api.hip), which
// tells the host whether a launch will write rd_src_t.out (only this kernel does)
bool rd_conv_ws_takes(const rd_conv_t& p) {
    const int skind = pp_sources_kind(p);
    if (p.taps != 9 || p.CoutPad % PP_NT || p.CinPad > 32 * PP_MAX_CHUNKS || !skind) return false;
    static const int ws = rd_switch("RD_CONV_WS", 3), ws_min2 = rd_switch("RD_CONV_WS_MIN2", 0);
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.N * (p.CoutPad / PP_NT);
    const int mode = rd_conv_lean_mode(p, PP_NT);
    const size_t tab = (size_t)(mode == 1 ? 1 : 2 * p.G) * p.CoutPad * sizeof(float);
    bool accumulates = false;                                  // gradient launches that add to an existing gradient stay with conv_pf_kernel
    if (mode == 2)
        for (int i = 0; i < 2; ++i) accumulates = accumulates || (p.dst[i].kind != RD_DST_NONE && p.dst[i].accumulate);
    if (skind == 2 && mode != 2) return false;
    // the epilogues address their destinations with 32-bit byte offsets built from 24-bit multiplies (conv_ws_kernel, zoff / ob)
    auto fits32 = [&](long long images, long long C) {
        const long long px = images * p.H * p.W;
        return px < (1ll << 24) && px * C * (long long)sizeof(bf16_t) < (1ll << 32);
    };
    if (mode == 1 && !fits32(p.N, p.Cout)) return false;
    if (mode == 2)
        for (int i = 0; i < 2; ++i)
            if (p.dst[i].kind != RD_DST_NONE && !fits32((long long)p.N + p.dst[i].n_off, p.dst[i].Cd)) return false;
    return mode && (ws & mode) && (mode == 1 || (tiles >= ws_min2 && !accumulates)) && tab <= (size_t)WsLds<0>::TAB_BYTES;
}

// does this launch also WRITE the rd_src_t.out tensors it is given?  Only the SOUT instantiations of conv_ws_kernel do: forward launches
// (mode 1) and gradient launches on a BatchNorm-backward source (mode 2, two-operand loader).  A gradient launch on plain sources runs
// conv_ws_kernel<2, TS>, which never stores what it stages.  ONE definition for the dispatch below and for rd_conv_honours_src_out.
bool rd_conv_ws_stores_sources(const rd_conv_t& p) {
    if (!rd_conv_ws_takes(p)) return false;
    return rd_conv_lean_mode(p, PP_NT) == 1 || pp_sources_kind(p) == 2;
}

int rd_conv_pp_dispatch(const rd_conv_t& p, hipStream_t st) {
    const int skind = pp_sources_kind(p);
    if (p.taps != 9 || p.CoutPad % PP_NT || p.CinPad > 32 * PP_MAX_CHUNKS || !skind) return RD_CONV_PP_NA;
    static int n_cu = 0;
    if (!n_cu) {
        n_cu = rd_num_cus();
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<0>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<0>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<1>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<1>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<0>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<1>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<1, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<0>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<1, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<1>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<0>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<2, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WsLds<1>::LDS);
    }
    const int tiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.N * (p.CoutPad / PP_NT);
    const int cus = (p.cu_limit > 0 && p.cu_limit < n_cu) ? p.cu_limit : n_cu;     // a side lane's budget (ramdsir.h)
    const int grid = tiles < cus ? tiles : cus;
    // Launches that qualify for a register epilogue (conv_dispatch.h: 1 forward, 2 gradient into plain tensors) take the
    // warp-specialised kernel.  Layer timings (scripts/ab_layers.sh, us, previous kernel -> conv_ws_kernel):
    //   forward:   256->256 @50x50 102 -> 90, 128->128 @100x100 99 -> 85, 64->64 @200x200 96 -> 84, 128->64 @100x100 62 -> 54,
    //              64->64 @100x100 43 -> 38: every forward launch gains;
    //   gradient:  64->64 @200x200 153 -> 137, 128->128 @100x100 129 -> 122, but 128->128 @50x50 40 -> 52, 64->64 @100x100
    //              55 -> 62: with few tiles per CU the two-workgroup conv_pf_kernel streams the materialised dz better, so
    //              gradient launches need >= RD_CONV_WS_MIN2 tiles (round 2 default: 1536 = 6 per CU).
    //   Round 3, in the STEP (scripts/sweep_opts.sh, alternating): threshold 1536 5.06-5.07 ms, 800: 5.19, 400: 5.22, never: 5.01-5.02 --
    //   a gradient launch that takes whole CUs (512 threads, 154 KB LDS) keeps the weight-gradient lane's kernels off them for its
    //   whole duration; the two-workgroup conv_pf_kernel shares.  Default: never for gradients (the kernel mode stays, tests force it).
    //   Round 4: the gradient mode's MFMA waves had their eight producer-tensor loads and their eight stores each behind a vmcnt(0)
    //   (scripts/ws_trace.py: last K step 9 000-10 700 cycles, epilogue 6 500-11 000); with both fixed a tile takes ~15 000 cycles
    //   instead of ~24 000 and every >= 64-channel gradient launch is 15-20 % faster here than on conv_pf_kernel (dec.convu2.conv3
    //   103 -> 87 us, dec.convu3.conv3 85 -> 70, dec.convu4.conv3 84 -> 69, enc.convd3.* 40 -> 33).  In the step it needs a compute-unit
    //   budget (tuning.py dgrad_cus = 160: the persistent workgroups leave 96 CUs to the other lanes): 4.48 -> 4.43 ms
    //   (scripts/sweep_ws2.sh, alternating; 4.48 with the whole GPU, 4.51 with 128).  Default: every non-accumulating gradient launch.
    static const int ws = rd_switch("RD_CONV_WS", 3), ws_min2 = rd_switch("RD_CONV_WS_MIN2", 0);
    const int mode = rd_conv_lean_mode(p, PP_NT);
    const size_t tab = (size_t)(mode == 1 ? 1 : 2 * p.G) * p.CoutPad * sizeof(float);
    bool accumulates = false;                                  // gradient launches that add to an existing gradient stay with conv_pf_kernel
    if (mode == 2)
        for (int i = 0; i < 2; ++i) accumulates = accumulates || (p.dst[i].kind != RD_DST_NONE && p.dst[i].accumulate);
    const bool takes = rd_conv_ws_takes(p);
    if (skind == 2 && !takes) return RD_CONV_PP_NA;            // a two-operand source runs on conv_ws_kernel<2, *, true> or not here at all
    if (takes) {
        // debug build only: RD_CONV_WS_EXP = timing experiments (-DRD_WS_EXP), RD_CONV_WS_TRACE_MIN = launches with at least
        // that many tiles record the s_memtime trace (scripts/ws_trace.py); both ride in the high bits of the tile count
        static const int ws_exp = rd_switch("RD_CONV_WS_EXP", 0), ws_trace_min = rd_switch("RD_CONV_WS_TRACE_MIN", 1 << 30);
        // 10 x 25 tiles where they put more of the MFMA rows on image pixels (sides 50 / 100 / 200: fewer tiles per CU).  The twelve
        // forward launches of the step that come here, alone (scripts/layer_bench.py, RD_CONV_WS_FLAT 0 -> 1): 574 -> 513 us summed
        // (dec.convu4.conv3 81 -> 66, dec.convu3.conv3 78 -> 65, dec.convu2.conv3 84 -> 73, the 50-pixel levels 32-37 -> 29-33);
        // step 4.97 -> 4.92 ms (scripts/sweep_opts.sh, three alternating pairs)
        static const int ws_flat = rd_switch("RD_CONV_WS_FLAT", 1);
        const int tiles1 = ((p.W + TileGeo<1>::W - 1) / TileGeo<1>::W) * ((p.H + TileGeo<1>::H - 1) / TileGeo<1>::H) * p.N * (p.CoutPad / PP_NT);
        const bool flat = ws_flat && tiles1 * 1.04 < tiles;
        const int nt = flat ? tiles1 : tiles, grid_ws = nt < cus ? nt : cus;
        // RD_CONV_WS_TRACE_MODE (0 any / 1 forward / 2 gradient) and RD_CONV_WS_TRACE_CIN (0 any) narrow the traced launches inside a whole step
        static const int ws_trace_mode = rd_switch("RD_CONV_WS_TRACE_MODE", 0), ws_trace_cin = rd_switch("RD_CONV_WS_TRACE_CIN", 0);
        const bool traced = tiles >= ws_trace_min && (!ws_trace_mode || ws_trace_mode == mode) && (!ws_trace_cin || ws_trace_cin == p.Cin);
        const int arg = nt | (ws_exp << 26) | (traced ? 1 << 25 : 0);
        bool sout = false;                                     // a source asks for its staged values to be stored as well (ramdsir.h)
        for (int i = 0; i < p.nsrc; ++i) sout = sout || p.src[i].out != nullptr;
        sout = sout && rd_conv_ws_stores_sources(p);
        if (mode == 1 && sout) {
            if (flat) rd_launch((conv_ws_kernel<1, 1, false, true>), dim3(grid_ws), dim3(512), WsLds<1>::LDS, st, p, arg, rdfin::current());
            else rd_launch((conv_ws_kernel<1, 0, false, true>), dim3(grid_ws), dim3(512), WsLds<0>::LDS, st, p, arg, rdfin::current());
        } else if (mode == 1) {
            if (flat) rd_launch((conv_ws_kernel<1, 1>), dim3(grid_ws), dim3(512), WsLds<1>::LDS, st, p, arg, rdfin::current());
            else rd_launch((conv_ws_kernel<1, 0>), dim3(grid_ws), dim3(512), WsLds<0>::LDS, st, p, arg, rdfin::current());
        } else if (skind == 2 && sout) {
            if (flat) rd_launch((conv_ws_kernel<2, 1, true, true>), dim3(grid_ws), dim3(512), WsLds<1>::LDS, st, p, arg, rdfin::current());
            else rd_launch((conv_ws_kernel<2, 0, true, true>), dim3(grid_ws), dim3(512), WsLds<0>::LDS, st, p, arg, rdfin::current());
        } else if (skind == 2) {
            if (flat) rd_launch((conv_ws_kernel<2, 1, true>), dim3(grid_ws), dim3(512), WsLds<1>::LDS, st, p, arg, rdfin::current());
            else rd_launch((conv_ws_kernel<2, 0, true>), dim3(grid_ws), dim3(512), WsLds<0>::LDS, st, p, arg, rdfin::current());
        } else {
            if (flat) rd_launch((conv_ws_kernel<2, 1>), dim3(grid_ws), dim3(512), WsLds<1>::LDS, st, p, arg, rdfin::current());
            else rd_launch((conv_ws_kernel<2, 0>), dim3(grid_ws), dim3(512), WsLds<0>::LDS, st, p, arg, rdfin::current());
        }
        return (int)hipGetLastError();
    }
    // conv_pp_kernel (LDS-staged epilogue, any destination): measured (gpurun_out/lb_pp*.txt) 1.06-1.2x over conv_pf_kernel
    // with at most one tile per CU -- the 25x25 level -- and 0.7-0.9x on longer tile ranges, where its un-overlapped
    // epilogue (2100 VALU per tile and wave at one workgroup per CU) costs more than the pipelined K loop gains.
    static const bool pp_all = rd_switch("RD_CONV_PP_ALL", 0) != 0;
    if (tiles > n_cu && !pp_all) return RD_CONV_PP_NA;
    rd_launch(conv_pp_kernel, dim3(grid), dim3(256), PP_LDS, st, p, tiles, rdfin::current());
    return (int)hipGetLastError();
}
