// conv_fused.hip -- backward of a small-channel 3x3 conv in ONE launch: the input gradient (dgrad, with the activation mask,
// the skip accumulation and the BatchNorm-backward sums of conv_small_kernel's gradient epilogue) AND the weight gradient.
//
// Why: on the 400x400 / 200x200 layers (<= 32 channels in and out) both halves of the backward are HBM-bound and read the
// same tensors -- dgrad: g, z (the BatchNorm backward folded into the read) and the producer's raw tensor z_prev (mask, BN sums);
// wgrad: g, z and act(bn(z_prev)).  Separately that is six tensor reads per layer; fused it is three, and the weight-gradient
// MFMAs run on matrix pipes that the HBM-bound dgrad leaves idle (scripts/ablate_step.py: the <= 32-channel weight gradients
// cost the step 0.74 ms although they sit on a side stream -- they compete with the dgrad chain for the same bandwidth).
//
// Structure (bf16 only): one 512-thread workgroup per CU walks consecutive 8x32 tiles of one image.
//   waves 4-7  LOADER: the halo tile of dz = P*g + Q*z + R (or a stored dz / dlogits) of tile t+1 goes raw into registers while
//              tile t is being used, is transformed and written to one of THREE LDS tile buffers [pixel][32 ch] (16-byte slots
//              XOR-swizzled by pixel, as conv_small_kernel's);
//   waves 0-3  COMPUTE, per iteration t:
//              1. request the epilogue operands of tile t (the forward input's raw tensor, the old gradient when accumulating),
//              2. dgrad MFMAs of tile t (weights x pixels roles, v_mfma_f32_32x32x16_bf16; weights resident in LDS),
//              3. weight-gradient MFMAs of tile t-1: dW[tap][co][ci] += sum over the tile's pixels q of a[q][ci] * dz[q - tap][co]
//                 -- K = pixels, so both operands come through the transposing LDS read ds_read_b64_tr_b16 from the [pixel][channel]
//                 tiles: dz from the halo tile the dgrad used (still intact: three buffers), a from the tile the epilogue of
//                 iteration t-1 wrote; v_mfma_f32_16x16x32_bf16, one 16x16 block x 9 taps per wave (36 accumulator registers;
//                 layers with fewer blocks than waves split the tile rows instead),
//              4. register epilogue of tile t: mask, skip accumulation, BN-backward sums, 16-byte NHWC stores -- and
//                 a = act(bn(z_prev)) (the forward input, needed for the mask anyway) written to the `a` tile for step 3 of t+1;
//              one barrier per iteration for both roles.
//   At the end every workgroup stores its dW block sums [9][CoutPad16][CinPad16] fp32; wgrad_reduce_kernel (wgrad.hip) sums the
//   workgroups in a fixed order (deterministic), as for the stand-alone weight-gradient kernels.
#include "conv_device.h"
#include "conv_dispatch.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short fz_s4;
typedef __attribute__((address_space(3))) fz_s4 fz_lds_s4;
__device__ __forceinline__ uint2 fz_tr(const char* p) {
    const fz_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fz_lds_s4*)(p));
    return __builtin_bit_cast(uint2, v);
}

// Workgroup barrier of the tile loop.  __syncthreads() is a fence + barrier: hipcc drains vmcnt(0) in front of it, i.e. every wave
// would wait at every tile for the prefetches it has just issued (the loader's next two tiles, the compute waves' epilogue
// operands and stores) -- one exposed memory latency per tile, which is what bounds a one-workgroup-per-CU kernel.  Only the LDS
// traffic has to be complete at this barrier; loads into registers are waited for by the compiler at their first use.
__device__ __forceinline__ void fz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FusedWg {
    rd_src_t a[2];       // the conv's forward input (== the dgrad destinations' producer tensors), as rd_wgrad_t describes it
    float* partial;      // [workgroups][9][CoutPadW][CinPadW]
    int CoutPadW, CinPadW;
};

constexpr int FZ_PH = TH + 2, FZ_PW = TW + 2, FZ_NPIX = FZ_PH * FZ_PW;      // 10 x 34 halo tile
// one dz tile buffer + 256 dummy records: loader items that do not exist (past the halo tile, channel slots past Cin) still store
// somewhere, so that the loader has no branch per item (conv_small_fwd.hip's header comment: one conditional vector-memory
// instruction anywhere in the loop turns every wait of the compiler into vmcnt(0))
constexpr int FZ_IN_BYTES = FZ_NPIX * 64 + 256 * 16;
__device__ uint4 fz_trash[256];                          // where the gradient stores of lanes without a destination pixel go
constexpr int FZ_A_BYTES = TH * TW * 64;                                     // one `a` tile buffer
constexpr int FZ_W_BYTES = 9 * 32 * 64;
constexpr int FZ_FIN = 3 * FZ_IN_BYTES + 2 * FZ_A_BYTES + FZ_W_BYTES + 64 * 8 + (32 + 32) * 4;   // coefficient table of a folded finalize (bn_fin.h)
constexpr int FZ_LDS = FZ_FIN + rdfin::FIN_LDS_FLOATS * 4;

// NSL: live 16-byte channel slots of dz (1, 2 or 4: compile-time item count); NQ: 2 = BatchNorm-backward source (g and z), 1 = a stored
// dz / dlogits (copied as it is).
// FZ_XP: timing experiments (compile with -DRD_FZ_EXP=bits; results are wrong when set): 1 no weight-gradient phase, 2 no dgrad MFMAs,
// 4 no gradient stores, 8 loader issues no global loads, 16 no epilogue-operand loads, 32 no `a` tile writes
#ifdef RD_FZ_EXP
#define FZ_XP(bit) (((RD_FZ_EXP) & (bit)) != 0)
#else
#define FZ_XP(bit) false
#endif
template <int NSL, int NQ>
__global__ __launch_bounds__(512, 1) void conv_small_bwd_fused_kernel(const rd_conv_t p, const FusedWg w, int tiles_per_wg, const rdfin::FinArg fa) {
    typedef bf16_t T;
    constexpr int S = 8, NT = 32, NV = 2;
    constexpr int NSH = NSL == 1 ? 0 : (NSL == 2 ? 1 : 2);
    constexpr int NIT = (FZ_NPIX * NSL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_inb = smem;                                                      // 3 x [NPIX][4 slots x 16 B]
    char* s_ab = smem + 3 * FZ_IN_BYTES;                                     // 2 x [256][4 x 16 B]
    uint4* s_w = reinterpret_cast<uint4*>(s_ab + 2 * FZ_A_BYTES);            // [9][32][4]
    double* s_red = reinterpret_cast<double*>(reinterpret_cast<char*>(s_w) + FZ_W_BYTES);   // [32][2], fp64 (conv_device.h flush_bstats)
    float* s_dsc = reinterpret_cast<float*>(s_red + 64);                                              // [32] forward-input BN scale (1 for raw tensors)
    float* s_dsh = s_dsc + 32;                                               // [32] shift

    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, ntiles = tiles_x * ((H + TH - 1) / TH);
    const int t_begin = blockIdx.x * tiles_per_wg;
    const int t_end = min(ntiles, t_begin + tiles_per_wg);
    const int nt = t_end - t_begin;
    const int NI = (nt + 2) & ~1;                          // iterations (barriers) of the tile loop, both roles
    const int n = blockIdx.z;
    const GroupMap gm = make_gm(p.gstart, p.G);
    const int g = group_of(gm, n);
    const int slot = (blockIdx.x + 7 * blockIdx.z) % rd_stat_nslots(p.stat_slots);

    // ---- LDS prologue (all 512 threads): zero the tile buffers (channel slots beyond the live ones stay zero), the packed
    //      dgrad weights, the per-channel coefficients of the forward input
    {
        uint4* z4 = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < (3 * FZ_IN_BYTES + 2 * FZ_A_BYTES) / 16; i += 512) z4[i] = make_uint4(0, 0, 0, 0);
        const T* wbase = reinterpret_cast<const T*>(p.w);
        for (int idx = threadIdx.x; idx < 9 * NT * 4; idx += 512) {
            const int sw = idx & 3, rec = idx >> 2;
            const int nn = rec % NT, tap = rec / NT;
            s_w[rec * 4 + (sw ^ ((nn >> 2) & 3))] = ld16(wbase + ((size_t)(tap * p.CoutPad + nn) * p.CinPad + sw * S));
        }
        if (threadIdx.x < 64) s_red[threadIdx.x] = 0.0;
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            const int dj = c >= p.c_split ? 1 : 0;
            const rd_src_t as = select_src(w.a, dj);
            const int cdd = c - (dj ? p.c_split : 0);
            const int ga = as.g_fixed >= 0 ? as.g_fixed : g;
            const bool live = c < p.Cout && cdd < as.C;
            const bool aff = live && as.mode != RD_SRC_RAW;
            s_dsc[c] = aff ? as.scale[ga * as.C + cdd] : 1.f;
            s_dsh[c] = aff ? as.shift[ga * as.C + cdd] : 0.f;
        }
    }
    // BatchNorm-backward finalize of this layer folded into the launch (bn_fin.h): P / Q / R reach the loader waves through LDS (src[])
    rd_src_t src[2];
    rdfin::conv_prologue_lds(p, fa, reinterpret_cast<float*>(smem + FZ_FIN), rdfin::FIN_LDS_FLOATS, src);
    __syncthreads();

    if (role == 1) {
        // =============================================================================== loader + weight-gradient waves
        const int sslot = tid & (NSL - 1);
        const bool live_slot = sslot * S < p.Cin;
        const rd_src_t ssrc = select_src(src, 0);
        PlainSrc<T> ps;
        plain_src_init<T>(ps, ssrc, live_slot ? sslot * S : 0);
        plain_src_coef<T>(ps, ssrc, g, live_slot ? sslot * S : 0);
        ItemGeom<NIT> ig;
#pragma unroll
        for (int b = 0; b < NIT; ++b) {
            const int pixi = (tid + b * 256) >> NSH;
            const int pix = min(pixi, FZ_NPIX - 1);
            ig.py[b] = (short)(pix / FZ_PW);
            ig.px[b] = (short)(pix - (pix / FZ_PW) * FZ_PW);
            // slots swizzled by the halo COLUMN; items that do not exist land in this thread's dummy record behind the tile
            ig.lds[b] = (pixi < FZ_NPIX && live_slot) ? pix * 4 + (sslot ^ ((ig.px[b] >> 2) & 3)) : FZ_NPIX * 4 + tid;
        }
        int ioff[NIT];
        pfu_item_offsets<T, NIT>(ioff, ps, ig, W);

        // ---- weight-gradient roles: 16x16 blocks (cob, cib); nb blocks over these 4 waves, a block's waves split the tile rows
        const int nbi = w.CinPadW >> 4, nb = (w.CoutPadW >> 4) * nbi;          // nb in {1, 2, 4}
        const int blk = wave % nb, kq = wave / nb, wpb = 4 / nb;
        const int cob = blk / nbi, cib = blk % nbi;
        typedef __attribute__((ext_vector_type(4))) float f32x4v;
        f32x4v accw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) accw[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // fragment addressing of the transposing reads: 16-lane group kg = 8-pixel K block, lane i16 -> pixel kg*8 + (i16>>2),
        // 4 channels (i16 & 3) of the wave's 16-channel block
        const int i16 = lane & 15, kg = lane >> 4;
        const int cpx = kg * 8 + (i16 >> 2);                                    // pixel column of this lane's first read
        const int zs0 = cob * 2 + ((i16 & 3) >> 1), as0 = cib * 2 + ((i16 & 3) >> 1), sub = (i16 & 1) * 8;
        // dz halo tile: slot swizzle ((c >> 2) & 3) of the halo column c (a function of the column only, so that these offsets are
        // per-lane constants + a row term; 16 consecutive columns of a row still spread over all banks for the dgrad's ds_read_b128)
        int zoff[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = cpx + 4 * j;
            zoff[j] = c * 64 + ((zs0 ^ ((c >> 2) & 3)) << 4) + sub;
        }
        // `a` tile [8 rows][32 px] x 64 B, slot swizzle x(c) = ((c>>1)&3) ^ (((c>>3)&1)<<1) of the pixel column c (written by the epilogue)
        int aoff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = cpx + 4 * j;
            aoff[j] = c * 64 + ((as0 ^ (((c >> 1) & 3) ^ (((c >> 3) & 1) << 1))) << 4) + sub;
        }

        // two register sets: tiles t+2 and t+3 are in flight while tile t+1 is transformed.  EVERY request and EVERY transform happens
        // on every path: tiles past the workgroup's range are ghosts (all items read pixel (0, 0), zeros go to a buffer nobody reads
        // any more), so that the compiler can count the loads in flight and wait for the older set only (vmcnt(NIT * NQ))
        uint4 rawA[NIT][NQ], rawB[NIT][NQ];
        constexpr int CK_ = NQ == 1 ? 2 : 1;               // pfu_consume KIND: copy / affine without activation
        auto request = [&](uint4 (&raw)[NIT][NQ], int j) {
            const bool ghost = j >= nt;
            const int t = t_begin + j;
            const int yh = (t / tiles_x) * TH - 1, xh = (t % tiles_x) * TW - 1;
            if (!FZ_XP(8)) pfu_issue_pre<T, NIT, NQ>(raw, ps, ig, ioff, n, H, W, yh, xh, ghost, !ghost && xh + FZ_PW > W);
        };
        auto transform = [&](const uint4 (&raw)[NIT][NQ], int j) {
            const bool ghost = j >= nt;
            const int t = t_begin + j;
            const int yh = ghost ? -(1 << 20) : (t / tiles_x) * TH - 1, xh = ghost ? -(1 << 20) : (t % tiles_x) * TW - 1;
            uint4* dstb = reinterpret_cast<uint4*>(s_inb + (j % 3) * FZ_IN_BYTES);
            pfu_consume<T, NIT, NQ, true, CK_>(raw, ps, ig, H, W, yh, xh, [&](int l, const uint4& u) { dstb[l] = u; });
        };
        // set A carries the even tiles of this workgroup, set B the odd ones
        request(rawA, 0);
        request(rawB, 1);
        transform(rawA, 0);
        request(rawA, 2);
        __syncthreads();                                   // tile 0 is in buffer 0
        // iteration `it`: tile it+1 goes into buffer (it+1) % 3, tile it+3 is requested into the register set just emptied, and the
        // weight gradient of tile it-1 is accumulated from the `a` tile the epilogue of iteration it-1 wrote and that tile's dz
        // halo buffer (still intact: three buffers) -- K = the 32 pixels of a tile row per MFMA
        auto step = [&](uint4 (&raw)[NIT][NQ], int it) {
            transform(raw, it + 1);
            request(raw, it + 3);
            if (it >= 1 && it <= nt && !FZ_XP(1)) {
                const char* s_a = s_ab + ((it - 1) & 1) * FZ_A_BYTES;
                const char* s_z = s_inb + ((it - 1) % 3) * FZ_IN_BYTES;
                for (int r = kq; r < TH; r += wpb) {
                    const uint2 a0 = fz_tr(s_a + aoff[0] + r * (TW * 64)), a1 = fz_tr(s_a + aoff[1] + r * (TW * 64));
                    const bf16x8 bfrag = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int rr = r + kh;                               // halo row; halo column = tile column + kw
                        uint2 d[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) d[j] = fz_tr(s_z + zoff[j] + rr * (FZ_PW * 64));
                        const uint4 f0 = make_uint4(d[0].x, d[0].y, d[1].x, d[1].y);
                        const uint4 f1 = make_uint4(__builtin_amdgcn_alignbit(d[0].y, d[0].x, 16), __builtin_amdgcn_alignbit(d[1].x, d[0].y, 16),
                                                    __builtin_amdgcn_alignbit(d[1].y, d[1].x, 16), __builtin_amdgcn_alignbit(d[2].x, d[1].y, 16));
                        const uint4 f2 = make_uint4(d[0].y, d[1].x, d[1].y, d[2].x);
                        accw[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f0), bfrag, accw[kh * 3 + 0], 0, 0, 0);
                        accw[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f1), bfrag, accw[kh * 3 + 1], 0, 0, 0);
                        accw[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f2), bfrag, accw[kh * 3 + 2], 0, 0, 0);
                    }
                }
            }
        };
        // both roles run NI = nt + 1 iterations rounded up to an even count (the compute waves' last one or two are empty), so that the
        // pair below has no conditional half: a request that exists on one path only would make every wait conservative
        for (int it = 0; it < NI; it += 2) {
            step(rawB, it);
            fz_barrier();
            step(rawA, it + 1);
            fz_barrier();
        }
        // ---- weight-gradient blocks: waves that split a block's rows (kq > 0) hand their sums to kq == 0 through LDS
        float* s_acc = reinterpret_cast<float*>(smem);     // [(wpb-1)][nb][9*4][64]: the tile buffers are dead now
        if (kq > 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[(((kq - 1) * nb + blk) * 36 + t * 4 + r) * 64 + lane] = accw[t][r];
        }
        __syncthreads();
        if (kq == 0) {
            for (int k2 = 0; k2 < wpb - 1; ++k2)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accw[t][r] += s_acc[((k2 * nb + blk) * 36 + t * 4 + r) * 64 + lane];
            // D of the 16x16 MFMA: column (N = ci) = lane & 15, row (M = co) = 4 * (lane >> 4) + r; accumulator index t is the
            // tap of the HALO offset (kh', kw'): the forward tap is the flipped one, 8 - t
            float* out = w.partial + (size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 9 * w.CoutPadW * w.CinPadW;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[((size_t)(8 - t) * w.CoutPadW + cob * 16 + 4 * kg + r) * w.CinPadW + cib * 16 + i16] = accw[t][r];
        }
    } else {
        // =============================================================================== dgrad waves
        constexpr int nks = NSL <= 2 ? 1 : 2;
        // ---- epilogue constants (conv_small_kernel's register epilogue: lane owns pixel li of a tile row and channels
        //      16v + 8h .. +7 after the permlane regroup)
        int cbv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cbv[v] = 16 * v + 8 * h;
        float sa[NV][S], sb[NV][S];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < S; ++e) sa[v][e] = sb[v][e] = 0.f;
        // per (v): destination / forward-input tensor of the lane's channel vector
        const T* ap[NV];
        T* gp[NV];
        int Cd[NV], cd[NV];
        bool live[NV], accum[NV], masked[NV], relu[NV];
        float slp[NV];                                       // activation slope of a = act(.) (1: none): one destination per vector
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cb = cbv[v];
            const int di = cb >= p.c_split ? 1 : 0;
            const rd_dst_t d = select_dst(p, di);
            const rd_src_t as = select_src(w.a, di);
            cd[v] = cb - (di ? p.c_split : 0);
            Cd[v] = d.Cd;
            live[v] = cb < p.Cout && d.kind == RD_DST_PLAIN;
            accum[v] = live[v] && d.accumulate;
            masked[v] = live[v] && d.act && d.z;
            slp[v] = (live[v] && as.mode == RD_SRC_AFFACT) ? as.slope : 1.f;
            relu[v] = masked[v] && as.mode == RD_SRC_AFFACT && as.slope == 0.f && d.slope == 0.f;
            ap[v] = reinterpret_cast<const T*>(as.ptr) + (size_t)(n + as.n_off) * H * W * as.C + cd[v];
            gp[v] = reinterpret_cast<T*>(d.g) + (size_t)(n + d.n_off) * H * W * d.Cd + cd[v];
        }
        // layers with <= 16 input channels of the forward conv: the second channel vector does not exist (wave-uniform)
        const bool two = __builtin_amdgcn_readfirstlane((int)(p.Cout > 16));
        __syncthreads();                                   // tile 0 is in buffer 0

        uint4 araw[2][NV];
        // epilogue operands (the forward input's raw tensor) of a tile: requested one iteration ahead, right after the previous
        // tile's epilogue has consumed the registers (clamped addresses: branch-free loads; invalid pixels are masked later)
        // unconditional like the loader's: a tile past the range (ghost) and a vector without a destination read one valid line
        const T* dummy = reinterpret_cast<const T*>(p.w);
        auto request = [&](int j) {
            if (FZ_XP(16)) return;
            const bool ghost = j >= nt;
            const int t = t_begin + j;
            const int xx0 = (t % tiles_x) * TW, yy0 = (t / tiles_x) * TH;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int y = min(yy0 + wave * 2 + mb, H - 1), x = min(xx0 + li, W - 1);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const T* q = (live[v] && !ghost) ? ap[v] + (unsigned)((y * W + x) * Cd[v]) : dummy;
                    araw[mb][v] = ld16(q);
                }
            }
        };
        request(0);
        for (int it = 0; it < NI; ++it) {
            const int t = t_begin + it;
            const int x0 = (t % tiles_x) * TW, y0 = (t / tiles_x) * TH;
            if (it < nt) {
                const uint4* s_in = reinterpret_cast<const uint4*>(s_inb + (it % 3) * FZ_IN_BYTES);
                char* s_aw = s_ab + (it & 1) * FZ_A_BYTES;
                // dgrad MFMAs of the wave's two tile rows (weights x pixels roles), INTERLEAVED: a row has one accumulator block, so
                // its 18 MFMAs are one dependent chain (64 cycles each instead of 32); two chains side by side hide that
                f32x16 accs[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accs[mb][r] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (FZ_XP(2)) break;
                    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int pix = (wave * 2 + mb + kh) * FZ_PW + li + kw;
                        Mma<T>::chunk(s_w + (tap * NT + li) * 4, (li >> 2) & 3, s_in + pix * 4, ((li + kw) >> 2) & 3, h, accs[mb], nks);
                    }
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const f32x16& acc = accs[mb];
                    // register epilogue of the row
                    const int y = y0 + wave * 2 + mb, x = x0 + li;
                    const bool valid = y < H && x < W;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        if (v == 1 && !two) {                // same vector-memory instructions on both paths (see the loader)
                            if (!FZ_XP(4)) fz_trash[tid] = make_uint4(0, 0, 0, 0);
                            continue;
                        }
                        const int cb = cbv[v];
                        float vec[S], xr[S], go[S], av[S];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned ua = __float_as_uint(acc[8 * v + j]);
                            const unsigned ub = __float_as_uint(acc[8 * v + 4 + j]);
                            const HalfSwap rs = rd_half_swap(ua, ub, h);
                            vec[j] = __uint_as_float(rs.r0);
                            vec[4 + j] = __uint_as_float(rs.r1);
                        }
                        Slot<T>::unpack(araw[mb][v], xr);
                        const bool on = valid && live[v];
                        float dsc[S], dsh[S];                // producer scale / shift of the vector's channels: four 16-byte LDS reads
#pragma unroll
                        for (int e = 0; e < S; e += 4) {
                            const float4 a4 = *reinterpret_cast<const float4*>(s_dsc + cb + e), b4 = *reinterpret_cast<const float4*>(s_dsh + cb + e);
                            dsc[e] = a4.x; dsc[e + 1] = a4.y; dsc[e + 2] = a4.z; dsc[e + 3] = a4.w;
                            dsh[e] = b4.x; dsh[e + 1] = b4.y; dsh[e + 2] = b4.z; dsh[e + 3] = b4.w;
                        }
                        if (relu[v]) {
                            // BN + ReLU producer (every layer but ConvD.bn1 / raw inputs): a = max(y, 0), g = y > 0 ? da : 0
#pragma unroll
                            for (int e = 0; e < S; ++e) {
                                const float yv = xr[e] * dsc[e] + dsh[e];
                                av[e] = fmaxf(yv, 0.f);
                                const float gn = (on && yv > 0.f) ? vec[e] : 0.f;
                                sa[v][e] += gn;
                                sb[v][e] += gn * xr[e];
                                go[e] = gn;
                            }
                        } else {
                            const float lo = masked[v] ? p.dst[0].slope : 1.f;
#pragma unroll
                            for (int e = 0; e < S; ++e) {
                                const float yv = xr[e] * dsc[e] + dsh[e];
                                av[e] = act_fn(yv, slp[v]);
                                const float m = yv > 0.f ? 1.f : lo;
                                const float gn = on ? vec[e] * m : 0.f;
                                sa[v][e] += gn;
                                sb[v][e] += gn * xr[e];
                                go[e] = gn;
                            }
                        }
                        if (accum[v]) {                      // the old gradient of a second consumer (rare among these layers: not prefetched)
                            float gold[S];
                            Slot<T>::unpack(on ? ld16(gp[v] + (unsigned)((y * W + x) * Cd[v])) : make_uint4(0, 0, 0, 0), gold);
#pragma unroll
                            for (int e = 0; e < S; ++e) go[e] += gold[e];
                        }
                        // unconditional store (lanes without a destination pixel write to the trash record): stores count in vmcnt too
                        if (!FZ_XP(4)) {
                            uint4* qd = on ? reinterpret_cast<uint4*>(gp[v] + (unsigned)((y * W + x) * Cd[v])) : &fz_trash[tid];
                            *qd = Slot<T>::pack(go);
                        }
                        // the forward input of this pixel for the weight gradient (zero outside the image / beyond the channels)
                        const int c = li, sl = 2 * v + h;
                        const uint4 au = on ? Slot<T>::pack(av) : make_uint4(0, 0, 0, 0);
                        if (!FZ_XP(32)) *reinterpret_cast<uint4*>(s_aw + ((wave * 2 + mb) * TW + c) * 64 +
                                                  ((sl ^ (((c >> 1) & 3) ^ (((c >> 3) & 1) << 1))) << 4)) = au;
                    }
                }
                request(it + 1);
            }
            fz_barrier();
        }

        // ---- BN-backward sums of all tiles: summed over the 32 lanes of a half-wave (conv_device.h half_wave_sums), one LDS atomic per
        //      lane, one global set per workgroup
        flush_half_wave_sums16<S, NV>(s_red, sa, sb, li, h);
        __syncthreads();                                   // (the other role's cross-wave dW barrier)
        if (tid < 32 && tid < p.Cout) {
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

inline bool aligned16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

template <int NSL, int NQ>
int fused_launch_one(dim3 grid, hipStream_t st, const rd_conv_t& p, const FusedWg& fw, int tpw) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_bwd_fused_kernel<NSL, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS);
        attr_set = true;
    }
    rd_launch((conv_small_bwd_fused_kernel<NSL, NQ>), grid, dim3(512), FZ_LDS, st, p, fw, tpw, rdfin::current());
    return (int)hipGetLastError();
}
int fused_launch(int nsl, int nq, dim3 grid, hipStream_t st, const rd_conv_t& p, const FusedWg& fw, int tpw) {
    if (nq == 2) return nsl == 1 ? fused_launch_one<1, 2>(grid, st, p, fw, tpw) : nsl == 2 ? fused_launch_one<2, 2>(grid, st, p, fw, tpw) : fused_launch_one<4, 2>(grid, st, p, fw, tpw);
    return nsl == 1 ? fused_launch_one<1, 1>(grid, st, p, fw, tpw) : nsl == 2 ? fused_launch_one<2, 1>(grid, st, p, fw, tpw) : fused_launch_one<4, 1>(grid, st, p, fw, tpw);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- host side
bool rd_bwd_fused_ok(const rd_conv_t& p, const rd_wgrad_t& w, int dtype) {
    if (dtype != RD_BF16 || rd_switch("RD_FUSED_BWD", 1) == 0) return false;
    if (p.taps != 9 || w.taps != 9 || p.emode != 1 || p.nsrc != 1) return false;
    if (p.w_tap_rows) return false;                        // a launch over a row block of a wider pack: conv_small_kernel only
    if (p.CinPad != 32 || p.CoutPad != 32 || p.Cin > 32 || p.Cout > 32) return false;
    const rd_src_t& s = p.src[0];
    if (!(s.mode == RD_SRC_RAW || s.mode == RD_SRC_BNBWD) || s.C % 8 || s.C < p.Cin || !aligned16(s.ptr)) return false;
    if (s.mode == RD_SRC_BNBWD && !aligned16(s.ptr2)) return false;
    // the same tensor as the weight gradient's dz operand, the same geometry
    if (w.dz.ptr != s.ptr || w.dz.ptr2 != s.ptr2 || w.dz.mode != s.mode || w.dz.C != s.C) return false;
    if (w.N != p.N || w.H != p.H || w.W != p.W || w.Cout != p.Cin || w.Cin != p.Cout || w.G != p.G) return false;
    if (p.c_split % 8) return false;
    const int nd = p.c_split < p.Cout ? 2 : 1;
    if (w.na != nd) return false;
    float slope0 = p.dst[0].slope;
    for (int i = 0; i < nd; ++i) {
        const rd_dst_t& d = p.dst[i];
        const rd_src_t& a = w.a[i];
        const int width = i == 0 ? (nd == 2 ? p.c_split : p.Cout) : p.Cout - p.c_split;
        if (d.kind != RD_DST_PLAIN || d.Cd % 8 || d.Cd != width || !aligned16(d.g)) return false;
        if (!(a.mode == RD_SRC_RAW || a.mode == RD_SRC_AFF || a.mode == RD_SRC_AFFACT) || a.C != d.Cd || !aligned16(a.ptr)) return false;
        if (a.n_off != d.n_off || a.g_fixed != d.g_fixed) return false;
        if (d.z && d.z != a.ptr) return false;             // the mask is taken from the tensor the forward conv read
        if (d.z && (a.scale != d.scale || a.shift != d.shift)) return false;
        if (d.act && d.z && a.mode != RD_SRC_AFFACT) return false;
        if (d.act && d.z && (d.slope != slope0 || a.slope != d.slope)) return false;
        if (!d.z && a.mode != RD_SRC_RAW) return false;
    }
    if ((size_t)p.N * p.H * p.W * 32 >= (1ull << 31)) return false;     // 32-bit element offsets inside the kernel
    return true;
}

static int fused_tiles_per_wg(const rd_conv_t& p, int& gx) {
    const int ntiles = ((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH);
    const bool limited = p.cu_limit > 0 && p.cu_limit < rd_num_cus();
    const long slots = limited ? p.cu_limit : rd_num_cus();               // one 512-thread workgroup per CU
    int tpw = 0;
    double best = 1e30;
    const int tmin = limited ? (int)(((long)ntiles * p.N + slots - 1) / slots) : 1;
    for (int t = tmin < 1 ? 1 : tmin; (t <= 64 || limited) && t <= ntiles; ++t) {
        const long wgs = (long)((ntiles + t - 1) / t) * p.N;
        // whole rounds of resident workgroups; a workgroup pays about one tile-time of pipeline fill + drain
        const double cost = (double)((wgs + slots - 1) / slots) * (t + 1.0);
        if (cost < best - 1e-9) { best = cost; tpw = t; }
        if (limited) break;
    }
    if (tpw <= 0) tpw = ntiles;
    gx = (ntiles + tpw - 1) / tpw;
    return tpw;
}

int64_t rd_bwd_fused_ws_bytes(const rd_conv_t& p, const rd_wgrad_t& w) {
    int gx;
    fused_tiles_per_wg(p, gx);
    const int cop = (w.Cout + 15) / 16 * 16, cip = (w.Cin + 15) / 16 * 16;
    return (int64_t)gx * p.N * 9 * cop * cip * (int64_t)sizeof(float);
}

int rd_bwd_fused_dispatch(const rd_conv_t& p, const rd_wgrad_t& w, hipStream_t st) {
    int gx;
    const int tpw = fused_tiles_per_wg(p, gx);
    FusedWg fw;
    fw.a[0] = w.a[0];
    fw.a[1] = w.a[1];
    fw.partial = w.partial;
    fw.CoutPadW = (w.Cout + 15) / 16 * 16;
    fw.CinPadW = (w.Cin + 15) / 16 * 16;
    dim3 grid(gx, 1, p.N);
    const int nl = (p.Cin + 7) / 8, nsl = nl <= 1 ? 1 : (nl <= 2 ? 2 : 4);
    const int nq = p.src[0].mode == RD_SRC_BNBWD ? 2 : 1;
    return fused_launch(nsl, nq, grid, st, p, fw, tpw);
}

// the workgroups' dW block sums -> dW (fixed order: deterministic); its own entry point so that the host can put it on the
// weight-gradient lane, off the dgrad chain
int rd_bwd_fused_reduce_dispatch(const rd_conv_t& p, const rd_wgrad_t& w, hipStream_t st) {
    int gx;
    fused_tiles_per_wg(p, gx);
    return rd_wgrad_reduce_launch(w.partial, w.dW, gx * p.N, 9, w.Cout, w.Cin, (w.Cout + 15) / 16 * 16, (w.Cin + 15) / 16 * 16, w.beta, st);
}
