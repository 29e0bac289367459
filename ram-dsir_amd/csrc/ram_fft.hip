// ram_fft.hip -- Random Amplitude Mixup (RAM) on the GPU: batched 2-D real FFT -> low-frequency amplitude
// lerp with a partner image -> inverse FFT -> clip -> normalise, written straight into the network's
// NHWC input batch.  Replaces the numpy trio extract_amp_spectrum / low_freq_mutate_np /
// source_to_target_freq (code/dataset/fundus.py:13-61, prostate.py:10-62) and the call sites
// fundus.py:211-225 / prostate.py:186-188, which run on the CPU in DataLoader workers in the reference.
//
// The amplitude lerp keeps the source phase, so it is a real gain on the source spectrum inside the
// centred (2b+1)^2 window and the identity elsewhere:   out = src + IFFT2( D ),
//     D[k] = (1-lam) * (|F_trg[k]|/|F_src[k]| - 1) * F_src[k]      for |ky|<=b, |kx|<=b     (0 elsewhere)
// (|F_src|==0: D = (1-lam)|F_trg|, the reference's angle()==0 case).  Only window bins are ever needed:
//   A  row pass     one workgroup per PAIR of image rows: Stockham FFTs (LDS ping-pong) of their 3 channels,
//                   keep bins kx = 0..b                                             -> rowspec[img][c][y][kx]
//   B  column pass  one workgroup per (sample, channel, 4 bins kx): FFT of the src and trg columns, mix on
//                   the 2b+1 bins, inverse FFT of the zero-padded columns             -> colout[n][c][y][kx]
//   C  row inverse  one workgroup per pair of output rows: Hermitian row spectrum (2b+1 bins) -> real row,
//                   + src, clip, scale, store both network inputs (img, img_freq) as NHWC dtype.
// The sides 256 / 384 / 400 / 512 (the reference's Fundus crop, Prostate slices, BASELINE.json's two synthetic
// shapes) run compile-time radix plans (4,4,4,4 / 4,4,4,2,3 / 4,4,5,5 / 4,4,4,4,2): every quotient, stride and
// twiddle step is a constant; any other side that factors into 2, 3, 5 runs the run-time plan.  Image rows are
// read as 16-byte vectors (fp32, or uint8 as decoded PNGs are: 1 byte per value), spectrum panels with the bin
// index fastest in both directions, the outputs as whole 16-byte channel slots.
// Standalone entry points for the reference's free functions: rd_ram_amp (|fft2|, full spectrum), rd_ram_mutate (the
// window lerp on two amplitude arrays), and rd_ram_mix with `trg_amp` (source_to_target_freq given an amplitude array).
#include "common.h"
#include "../../include/ramdsir.h"
#include "ram_dft.h"

// THE STOCKHAM KERNELS (row pass, column pass, row inverse, column amplitude) CLAIM THE WHOLE LDS OF THEIR COMPUTE UNIT (round 6).  The
// column and row-inverse kernels of the training path need 28.8 KB, and with that five of their
// workgroups -- or workgroups of OTHER kernels -- share a CU.  Beside certain kernels of the step (conv_small_kernel forward, the 32-wide
// conv_kernel, the small weight-gradient kernels; on another stream of the same process or in another process) they then produced slightly
// wrong results: single elements of a transform off by a few per cent, i.e. whole columns / rows of the mixed image off by a bf16 ulp or
// two -- in 44 % of the PIPELINED steps of one process (RAM beside the encoder backward: scripts/r6/pipelined_x_check.py) and in 2-6 % of
// the steps under three-process load, never alone, never beside innocent kernels (profiles/r06_ram_coresidency.txt: LDS contents,
// barriers, sqrt / division and the MFMA regroup all hold under the same load; padding the buffers by 16 KB on both sides does not help;
// one workgroup per CU does).  ROOT CAUSE (found later the same round, profiles/r06_pk_opsel_erratum.txt): the SLP vectorizer compiled the
// complex butterflies to packed fp32 instructions with swapped halves (op_sel), and such an instruction reads 0 for the swapped operand while a
// wave of another kernel executes MFMAs on the same SIMD.  The library is built without the vectorizers now (no packed fp32 in the device code);
// the whole-CU claim stays as the second, independent measure -- it costs 3 us and also lowered the kernels' measured traffic.
// RD_RAM_LDS_EXCLUSIVE=0 / RD_RAM_LDS_PAD builds are for the experiments only.
#ifndef RD_RAM_LDS_EXCLUSIVE
#define RD_RAM_LDS_EXCLUSIVE 1
#endif
#ifndef RD_RAM_LDS_PAD
#define RD_RAM_LDS_PAD 0            // experiment builds: bytes of unused LDS in front of and behind the buffers
#endif
constexpr size_t RAM_CU_LDS = 160 * 1024;
inline size_t ram_lds_request(size_t need) {
    need += 2 * RD_RAM_LDS_PAD;
    return (RD_RAM_LDS_EXCLUSIVE && need < RAM_CU_LDS) ? RAM_CU_LDS : need;
}

namespace {

struct FftPlan {
    int N, nst;
    int radix[12];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// s * i * a   (s = -1 forward, +1 inverse)
__device__ __forceinline__ float2 imul(float2 a, float s) { return make_float2(-s * a.y, s * a.x); }

// radix-R DFT of v[0..R) in place (v[m] <- X_m); sgn = -1 forward, +1 inverse
template <int R>
__device__ __forceinline__ void butterfly(float2* v, float sgn) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = imul(csub(v[1], v[3]), sgn);
        v[0] = cadd(t0, t2);
        v[1] = cadd(t1, t3);
        v[2] = csub(t0, t2);
        v[3] = csub(t1, t3);
    } else if constexpr (R == 3) {
        const float2 t1 = cadd(v[1], v[2]);
        const float2 t2 = csub(v[0], cscale(t1, 0.5f));
        const float2 t3 = imul(cscale(csub(v[1], v[2]), 0.86602540378443864676f), sgn);
        v[0] = cadd(v[0], t1);
        v[1] = cadd(t2, t3);
        v[2] = csub(t2, t3);
    } else {  // R == 5
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
        const float2 m1 = cadd(v[0], cadd(cscale(a1, c1), cscale(a2, c2)));
        const float2 m2 = cadd(v[0], cadd(cscale(a1, c2), cscale(a2, c1)));
        const float2 n1 = imul(cadd(cscale(b1, s1), cscale(b2, s2)), sgn);
        const float2 n2 = imul(csub(cscale(b1, s2), cscale(b2, s1)), sgn);
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = cadd(m1, n1);
        v[4] = csub(m1, n1);
        v[2] = cadd(m2, n2);
        v[3] = csub(m2, n2);
    }
}

// One Stockham autosort stage of `batch` independent length-N transforms in LDS, all constants known at compile time.
// Transform t ping-pongs between buf[(t*2+0)*N ..] and buf[(t*2+1)*N ..].  tw[TWS*k] = (cos 2 pi k/N, -sin 2 pi k/N)
// (TWS = 2: the table of the twice-as-long transform, shared with the real-input post-pass).
template <int R, bool INV, int N, int NS, int TWS>
__device__ __forceinline__ void stage_fixed(float2* buf, int cur, int batch, const float2* tw, int tid, int nthreads) {
    constexpr int nb = N / R, tstep = N / (NS * R);
    constexpr float sgn = INV ? 1.f : -1.f;
    for (int jj = tid; jj < batch * nb; jj += nthreads) {
        const int t = jj / nb, j = jj - t * nb;
        const int q = j / NS, k = j - q * NS;
        const float2* a = buf + (t * 2 + cur) * N;
        float2* o = buf + (t * 2 + (cur ^ 1)) * N + q * NS * R + k;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = a[j + r * nb];
        if constexpr (NS > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                float2 w = tw[TWS * r * k * tstep];
                if (INV) w.y = -w.y;
                v[r] = cmul(v[r], w);
            }
        }
        butterfly<R>(v, sgn);
#pragma unroll
        for (int r = 0; r < R; ++r) o[r * NS] = v[r];
    }
    __syncthreads();
}

// compile-time plan = the radix list; returns which half holds the result
template <int N, bool INV, int TWS, int NS, int CUR>
__device__ __forceinline__ int fft_chain(float2*, int, const float2*, int, int) {
    static_assert(NS == N, "the radices must multiply to N");
    return CUR;
}
template <int N, bool INV, int TWS, int NS, int CUR, int R, int... Rest>
__device__ __forceinline__ int fft_chain(float2* buf, int batch, const float2* tw, int tid, int nthreads) {
    stage_fixed<R, INV, N, NS, TWS>(buf, CUR, batch, tw, tid, nthreads);
    return fft_chain<N, INV, TWS, NS * R, CUR ^ 1, Rest...>(buf, batch, tw, tid, nthreads);
}

template <int N, bool INV, int TWS = 1>
__device__ __forceinline__ int fft_fixed(float2* buf, int batch, const float2* tw, int tid, int nthreads) {
    if constexpr (N == 128) return fft_chain<N, INV, TWS, 1, 0, 4, 4, 4, 2>(buf, batch, tw, tid, nthreads);
    else if constexpr (N == 192) return fft_chain<N, INV, TWS, 1, 0, 4, 4, 4, 3>(buf, batch, tw, tid, nthreads);
    else if constexpr (N == 200) return fft_chain<N, INV, TWS, 1, 0, 4, 2, 5, 5>(buf, batch, tw, tid, nthreads);
    else if constexpr (N == 256) return fft_chain<N, INV, TWS, 1, 0, 4, 4, 4, 4>(buf, batch, tw, tid, nthreads);
    else if constexpr (N == 384) return fft_chain<N, INV, TWS, 1, 0, 4, 4, 4, 2, 3>(buf, batch, tw, tid, nthreads);
    else if constexpr (N == 400) return fft_chain<N, INV, TWS, 1, 0, 4, 4, 5, 5>(buf, batch, tw, tid, nthreads);
    else {
        static_assert(N == 512, "compile-time plans: 128, 192, 200, 256, 384, 400, 512");
        return fft_chain<N, INV, TWS, 1, 0, 4, 4, 4, 4, 2>(buf, batch, tw, tid, nthreads);
    }
}

// run-time plan (any side that factors into 2, 3, 5): same stages, quotients by float reciprocals (exact for these
// ranges: a < 2^12, divisor <= 1024)
__device__ int fft_runtime(float2* buf, int batch, const FftPlan& pl, const float2* tw, bool inverse, int tid, int nthreads) {
    const int N = pl.N;
    const float sgn = inverse ? 1.f : -1.f;
    int Ns = 1, cur = 0;
    for (int st = 0; st < pl.nst; ++st) {
        const int R = pl.radix[st];
        const int nb = N / R;
        const int tstep = N / (Ns * R);
        const float inv_nb = 1.0f / (float)nb, inv_Ns = 1.0f / (float)Ns;
        for (int jj = tid; jj < batch * nb; jj += nthreads) {
            const int t = (int)(((float)jj + 0.5f) * inv_nb), j = jj - t * nb;
            const float2* a = buf + (size_t)(t * 2 + cur) * N;
            const int q = (int)(((float)j + 0.5f) * inv_Ns);
            const int k = j - q * Ns;
            float2* o = buf + (size_t)(t * 2 + (cur ^ 1)) * N + q * Ns * R + k;
            float2 v[5];
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r < R) {
                    v[r] = a[j + r * nb];
                    if (r > 0 && k > 0) {
                        float2 w = tw[r * k * tstep];
                        if (inverse) w.y = -w.y;
                        v[r] = cmul(v[r], w);
                    }
                }
            if (R == 2) butterfly<2>(v, sgn);
            else if (R == 4) butterfly<4>(v, sgn);
            else if (R == 3) butterfly<3>(v, sgn);
            else butterfly<5>(v, sgn);
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r < R) o[r * Ns] = v[r];
        }
        __syncthreads();
        cur ^= 1;
        Ns *= R;
    }
    return cur;
}

template <int NFIX>
__device__ __forceinline__ int fft_any(float2* buf, int batch, const FftPlan& pl, const float2* tw, bool inverse, int tid, int nthreads) {
    if constexpr (NFIX == 0) return fft_runtime(buf, batch, pl, tw, inverse, tid, nthreads);
    else return inverse ? fft_fixed<NFIX, true>(buf, batch, tw, tid, nthreads) : fft_fixed<NFIX, false>(buf, batch, tw, tid, nthreads);
}

struct RamArgs {
    const void* src; const void* trg; const float* trg_amp; const float* lam;
    void* out_img; void* out_freq;
    float2* rowspec; float2* colout;
    const float2* tw_w; const float2* tw_h;
    float* amp_out;
    int B, H, W, b, cs, KP, nkeep;       // nkeep bins kept per row (b+1, or W/2+1 for the amplitude entry point); KP = nkeep padded to 4
    int nimg_rows, nch, src_u8, planar;  // row pass: images to transform, channels per image (3; 1 = planar planes)
    float clip_lo, clip_hi, scale, div, offset;
    FftPlan pw, ph;
};

// value i (= x*3 + c) of an interleaved image row
__device__ __forceinline__ void put3(float2* buf, int W, int i, float v) {
    const int x = i / 3, c = i - 3 * x;
    buf[c * 2 * W + x] = make_float2(v, 0.f);
}
// A: grid (ceil(H/ROWS), images).  An all-zero channel must give an EXACTLY zero spectrum, because the mix divides by
// |F_src| (the reference's angle()==0 branch, fundus.py:48 on a constant-zero plane): never pack two different signals into
// one complex transform.  Compile-time sides use the real-input form -- the W reals of ONE channel as the W/2 complex points
// z[n] = x[2n] + i x[2n+1] (transform c spans 2 halves x W/2 float2 = 2 W floats: channel c's pixel x is float c*2W + x), a
// W/2-point transform, then X[k] = E[k] + W_N^k O[k] with E, O = (Z[k] +- conj Z[W/2-k]) / 2 (/ i) for the kept bins only --
// which is zero for zero input and halves the butterflies and LDS traffic; the run-time plan transforms (x, 0).
template <int NW, int ROWS, int NT>
__global__ __launch_bounds__(NT) void ram_row_fwd_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    constexpr bool REAL = NW != 0;
    const int W = a.W, H = a.H, nch = a.nch;
    const int M = REAL ? W / 2 : W;                         // points per transform
    float2* buf = reinterpret_cast<float2*>(smem_);        // [ROWS * nch transforms][2][M], then the twiddle table [W]
    const int y0 = blockIdx.x * ROWS, n = blockIdx.y;
    float2* s_tw = buf + ROWS * nch * 2 * M;               // twiddles out of LDS: a butterfly does not wait on L2 for them
    for (int i = threadIdx.x; i < W; i += blockDim.x) s_tw[i] = a.tw_w[i];
    const int rows = min(ROWS, H - y0);
    for (int r = 0; r < rows; ++r) {
        float2* rb = buf + (size_t)r * nch * 2 * M;
        float* rf = reinterpret_cast<float*>(rb);
        const int y = y0 + r;
        if (a.planar) {
            const float* p = reinterpret_cast<const float*>(a.src) + ((size_t)n * H + y) * W;
            for (int x = threadIdx.x; x < W; x += blockDim.x) {
                if constexpr (REAL) rf[x] = p[x];
                else rb[x] = make_float2(p[x], 0.f);
            }
            continue;
        }
        const size_t row = ((size_t)(n < a.B ? n : n - a.B) * H + y) * W * 3;
        const void* base = n < a.B ? a.src : a.trg;
        if constexpr (REAL) {
            // 4 pixels per thread (W % 4 == 0 for every compile-time side): 12 bytes / three float4 in, one 16-byte LDS
            // store per channel out -- no per-value index arithmetic
            for (int q = threadIdx.x; q < W / 4; q += blockDim.x) {
                float v[12];
                if (a.src_u8) {
                    const unsigned* p32 = reinterpret_cast<const unsigned*>(reinterpret_cast<const uint8_t*>(base) + row) + 3 * q;
                    const unsigned w0 = p32[0], w1 = p32[1], w2 = p32[2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (float)((w0 >> (8 * e)) & 0xffu);
                        v[4 + e] = (float)((w1 >> (8 * e)) & 0xffu);
                        v[8 + e] = (float)((w2 >> (8 * e)) & 0xffu);
                    }
                } else {
                    const float4* p4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + row) + 3 * q;
                    const float4 f0 = p4[0], f1 = p4[1], f2 = p4[2];
                    v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
                    v[8] = f2.x; v[9] = f2.y; v[10] = f2.z; v[11] = f2.w;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)                  // v[3*j + c] = pixel 4q+j, channel c
                    *reinterpret_cast<float4*>(rf + c * 2 * W + 4 * q) = make_float4(v[c], v[3 + c], v[6 + c], v[9 + c]);
            }
        } else if (a.src_u8) {
            const uint8_t* p = reinterpret_cast<const uint8_t*>(base) + row;
            for (int i = threadIdx.x; i < 3 * W; i += blockDim.x) put3(rb, W, i, (float)p[i]);
        } else {
            const float* p = reinterpret_cast<const float*>(base) + row;
            for (int i = threadIdx.x; i < 3 * W; i += blockDim.x) put3(rb, W, i, p[i]);
        }
    }
    __syncthreads();
    const int nk = a.nkeep, KP = a.KP;
    if constexpr (REAL) {
        const int cur = fft_fixed<NW / 2, false, 2>(buf, rows * nch, s_tw, threadIdx.x, blockDim.x);
        for (int i = threadIdx.x; i < rows * nch * nk; i += blockDim.x) {
            const int t = i / nk, kx = i - t * nk;          // t = r * nch + c;  kx <= b < M
            const int r = t / nch, c = t - r * nch;
            const float2* Z = buf + (size_t)(t * 2 + cur) * M;
            const float2 zk = Z[kx], zm = Z[kx == 0 ? 0 : M - kx];
            const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));        // (Z[k] + conj Z[M-k]) / 2
            const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));       // (Z[k] - conj Z[M-k]) / (2i)
            a.rowspec[(((size_t)n * nch + c) * H + y0 + r) * KP + kx] = cadd(E, cmul(s_tw[kx], O));
        }
    } else {
        const int cur = fft_runtime(buf, rows * nch, a.pw, s_tw, false, threadIdx.x, blockDim.x);
        for (int i = threadIdx.x; i < rows * nch * nk; i += blockDim.x) {
            const int t = i / nk, kx = i - t * nk;          // t = r * nch + c
            const int r = t / nch, c = t - r * nch;
            a.rowspec[(((size_t)n * nch + c) * H + y0 + r) * KP + kx] = buf[(size_t)(t * 2 + cur) * W + kx];
        }
    }
}

// B: grid (ceil((b+1)/KT), 3, B): KT bins per workgroup.  LDS transforms: [0,KT) the src columns, [KT,2KT) the trg
// columns (forward), then [KT,2KT) again as the inverse inputs.
template <int NH, int KT, int NT>
__global__ __launch_bounds__(NT) void ram_col_mix_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_ + RD_RAM_LDS_PAD);        // [2*KT][2][H], then the twiddle table [H]
    const int kx0 = blockIdx.x * KT, c = blockIdx.y, n = blockIdx.z, H = a.H, KP = a.KP, b = a.b;
    const int nk = min(KT, a.nkeep - kx0);
    float2* s_tw = buf + 2 * KT * 2 * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) s_tw[i] = a.tw_h[i];
    const bool have_trg = a.trg_amp == nullptr;
    const float2* cs = a.rowspec + ((size_t)(n * 3 + c) * H) * KP + kx0;
    const float2* ct = a.rowspec + ((size_t)((n + a.B) * 3 + c) * H) * KP + kx0;
    for (int i = threadIdx.x; i < H * KT; i += blockDim.x) {
        const int y = i / KT, k = i - y * KT;               // bins fastest: KT*8 contiguous bytes per y
        const bool ok = k < nk;
        buf[(size_t)(k * 2) * H + y] = ok ? cs[(size_t)y * KP + k] : make_float2(0.f, 0.f);
        if (have_trg) buf[(size_t)((KT + k) * 2) * H + y] = ok ? ct[(size_t)y * KP + k] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int cur = fft_any<NH>(buf, have_trg ? 2 * KT : KT, a.ph, s_tw, false, threadIdx.x, blockDim.x);
    const float lam = a.lam[n];
    const int nwin = 2 * b + 1;
    constexpr int MAXI = (KT * 1025 + NT - 1) / NT;          // window items per thread (H <= 1024)
    float2 dv[MAXI];
    int di[MAXI];
#pragma unroll
    for (int m = 0; m < MAXI; ++m) {
        const int i = threadIdx.x + m * NT;
        di[m] = -1;
        if (i < KT * nwin) {
            const int k = i / nwin, ky = i - k * nwin - b;
            const int idx = (ky + H) % H;
            if (k < nk) {
                const float2 Fs = buf[(size_t)(k * 2 + cur) * H + idx];
                float At;
                if (have_trg) {
                    const float2 Ft = buf[(size_t)((KT + k) * 2 + cur) * H + idx];
                    At = sqrtf(Ft.x * Ft.x + Ft.y * Ft.y);
                } else {
                    At = a.trg_amp[(((size_t)n * 3 + c) * H + idx) * a.W + kx0 + k];
                }
                const float As = sqrtf(Fs.x * Fs.x + Fs.y * Fs.y);
                dv[m] = As > 0.f ? cscale(Fs, (1.f - lam) * (At / As - 1.f)) : make_float2((1.f - lam) * At, 0.f);
                di[m] = k * H + idx;
            }
        }
    }
    __syncthreads();
    float2* ib = buf + (size_t)KT * 2 * H;                  // the trg block becomes the inverse transforms
    for (int i = threadIdx.x; i < KT * H; i += blockDim.x) ib[(size_t)(i / H) * 2 * H + (i % H)] = make_float2(0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MAXI; ++m)
        if (di[m] >= 0) ib[(size_t)(di[m] / H) * 2 * H + (di[m] % H)] = dv[m];
    __syncthreads();
    const int c2 = fft_any<NH>(ib, KT, a.ph, s_tw, true, threadIdx.x, blockDim.x);
    float2* out = a.colout + ((size_t)(n * 3 + c) * H) * KP + kx0;
    for (int i = threadIdx.x; i < H * KT; i += blockDim.x) {
        const int y = i / KT, k = i - y * KT;
        if (k < nk) out[(size_t)y * KP + k] = ib[(size_t)(k * 2 + c2) * H + y];
    }
}

// amplitude of the full spectrum (extract_amp_spectrum, fundus.py:13-19): grid (W/2+1, planes); column kx and its mirror
__global__ __launch_bounds__(256) void ram_col_amp_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_);        // [1][2][H], then the twiddle table [H]
    const int kx = blockIdx.x, pl = blockIdx.y, H = a.H, W = a.W, KP = a.KP;
    float2* s_tw = buf + 2 * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        s_tw[i] = a.tw_h[i];
        buf[i] = a.rowspec[((size_t)pl * H + i) * KP + kx];
    }
    __syncthreads();
    const int cur = fft_runtime(buf, 1, a.ph, s_tw, false, threadIdx.x, blockDim.x);
    const bool mirror = kx > 0 && kx != W - kx;             // F[-ky][-kx] = conj F[ky][kx]
    for (int ky = threadIdx.x; ky < H; ky += blockDim.x) {
        const float2 F = buf[cur * H + ky];
        const float A = sqrtf(F.x * F.x + F.y * F.y);
        a.amp_out[((size_t)pl * H + ky) * W + kx] = A;
        if (mirror) a.amp_out[((size_t)pl * H + (H - ky) % H) * W + W - kx] = A;
    }
}

// C: grid (ceil(H/ROWS), B)
template <typename T, int NW, int ROWS, int NT>
__global__ __launch_bounds__(NT) void ram_row_inv_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_ + RD_RAM_LDS_PAD);        // [ROWS * 2 transforms][2][W], then the twiddle table [W]
    const int y0 = blockIdx.x * ROWS, n = blockIdx.y, W = a.W, H = a.H, nb1 = a.b + 1, KP = a.KP;
    const int rows = min(ROWS, H - y0);
    float2* s_tw = buf + ROWS * 2 * 2 * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) s_tw[x] = a.tw_w[x];
    for (int i = threadIdx.x; i < rows * 2 * W; i += blockDim.x) buf[(size_t)(i / W) * 2 * W + (i % W)] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = threadIdx.x; i < rows * nb1; i += blockDim.x) {
        const int r = i / nb1, kx = i - r * nb1;
        const size_t o = ((size_t)(n * 3) * H + y0 + r) * KP + kx, cstep = (size_t)H * KP;
        const float2 R0 = a.colout[o], R1 = a.colout[o + cstep], R2 = a.colout[o + 2 * cstep];
        float2* z01 = buf + (size_t)(r * 2) * 2 * W;
        float2* z2 = buf + (size_t)(r * 2 + 1) * 2 * W;
        z01[kx] = make_float2(R0.x - R1.y, R0.y + R1.x);                 // R0 + i R1
        z2[kx] = R2;
        if (kx > 0) {
            z01[W - kx] = make_float2(R0.x + R1.y, -R0.y + R1.x);        // conj(R0) + i conj(R1)
            z2[W - kx] = make_float2(R2.x, -R2.y);
        }
    }
    __syncthreads();
    const int cur = fft_any<NW>(buf, rows * 2, a.pw, s_tw, true, threadIdx.x, blockDim.x);
    const float inv = 1.0f / ((float)H * (float)W);
    constexpr int SLOT = 16 / (int)sizeof(T);               // elements of one 16-byte channel slot
    for (int i = threadIdx.x; i < rows * W; i += blockDim.x) {
        const int r = i / W, x = i - r * W;
        const float2 v01 = buf[(size_t)((r * 2) * 2 + cur) * W + x];
        const float corr[3] = {v01.x * inv, v01.y * inv, buf[(size_t)((r * 2 + 1) * 2 + cur) * W + x].x * inv};
        const size_t pix = ((size_t)n * H + y0 + r) * W + x;
        float s[3];
        if (a.src_u8) {
            const uint8_t* sp = reinterpret_cast<const uint8_t*>(a.src) + pix * 3;
            s[0] = (float)sp[0]; s[1] = (float)sp[1]; s[2] = (float)sp[2];
        } else {
            const float* sp = reinterpret_cast<const float*>(a.src) + pix * 3;
            s[0] = sp[0]; s[1] = sp[1]; s[2] = sp[2];
        }
        float oi[SLOT], of[SLOT];
#pragma unroll
        for (int c = 0; c < SLOT; ++c) oi[c] = of[c] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float f = fminf(fmaxf(s[c] + corr[c], a.clip_lo), a.clip_hi);
            // div != 0: x / div + offset, the reference's `img /= 127.5; img -= 1.0` bit for bit (fundus.py:217-218)
            oi[c] = a.div != 0.f ? __fdiv_rn(s[c], a.div) + a.offset : s[c] * a.scale + a.offset;
            of[c] = a.div != 0.f ? __fdiv_rn(f, a.div) + a.offset : f * a.scale + a.offset;
        }
        T* pi = reinterpret_cast<T*>(a.out_img) + pix * a.cs;
        T* pf = reinterpret_cast<T*>(a.out_freq) + pix * a.cs;
        if (a.cs == SLOT) {                                  // one 16-byte slot per pixel: channels 3.. are written as zeros
            *reinterpret_cast<uint4*>(pi) = Slot<T>::pack(oi);
            *reinterpret_cast<uint4*>(pf) = Slot<T>::pack(of);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pi[c] = from_f<T>(oi[c]);
                pf[c] = from_f<T>(of[c]);
            }
        }
    }
}

// low_freq_mutate_np (fundus.py:21-39) on two amplitude arrays [C][H][W]: lerp inside the centred window
__global__ void ram_mutate_kernel(const float* as, const float* at, float* out, int C, int H, int W, int b, float lam) {
    const size_t total = (size_t)C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int ky = y < (H + 1) / 2 ? y : y - H, kx = x < (W + 1) / 2 ? x : x - W;   // fftfreq order
        const bool in = ky >= -b && ky <= b && kx >= -b && kx <= b;
        out[i] = in ? as[i] * lam + at[i] * (1.f - lam) : as[i];
    }
}

bool make_plan(int N, FftPlan& p) {
    p.N = N;
    p.nst = 0;
    int n = N;
    while (n % 4 == 0) { p.radix[p.nst++] = 4; n /= 4; }
    while (n % 2 == 0) { p.radix[p.nst++] = 2; n /= 2; }
    while (n % 3 == 0) { p.radix[p.nst++] = 3; n /= 3; }
    while (n % 5 == 0) { p.radix[p.nst++] = 5; n /= 5; }
    return n == 1 && p.nst <= 12 && N <= 1024;
}

int pad4(int v) { return (v + 3) / 4 * 4; }

template <int NW, int ROWS, int NT>
void launch_row_fwd_r(const RamArgs& a, int nimg, hipStream_t st) {
    const size_t lds = ram_lds_request((size_t)(ROWS * a.nch * 2 * (NW ? a.W / 2 : a.W) + a.W) * sizeof(float2));
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ram_row_fwd_kernel<NW, ROWS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    rd_launch((ram_row_fwd_kernel<NW, ROWS, NT>), dim3((a.H + ROWS - 1) / ROWS, nimg), dim3(NT), lds, st, a);
}
// one workgroup per CU (ram_lds_request): 1024 threads on 8 image rows; the run-time plan (any side) keeps one row and 256 threads
template <int NW>
void launch_row_fwd(const RamArgs& a, int nimg, hipStream_t st) {
    if (NW == 0) return launch_row_fwd_r<NW, 1, 256>(a, nimg, st);
    const int rows = rd_switch("RD_RAM_ROWS_FWD", RD_RAM_LDS_EXCLUSIVE ? 8 : 1);
    if (RD_RAM_LDS_EXCLUSIVE && rows == 8) return launch_row_fwd_r<NW, 8, 1024>(a, nimg, st);
    if (rows == 1) launch_row_fwd_r<NW, 1, 256>(a, nimg, st);
    else launch_row_fwd_r<NW, 2, 256>(a, nimg, st);
}

template <int NH, int KT, int NT>
int launch_col_mix_k(const RamArgs& a, hipStream_t st) {
    const size_t lds = ram_lds_request((size_t)(2 * KT * 2 + 1) * a.H * sizeof(float2));
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ram_col_mix_kernel<NH, KT, NT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    rd_launch((ram_col_mix_kernel<NH, KT, NT>), dim3((a.nkeep + KT - 1) / KT, 3, a.B), dim3(NT), lds, st, a);
    return 0;
}
// One workgroup per CU (ram_lds_request): 1024 threads on 4 bins (264 workgroups at 8 x 400 x 400: one round of the 256 CUs) instead of five
// workgroups of 256 threads on 2 bins each
template <int NH>
int launch_col_mix(const RamArgs& a, hipStream_t st) {
    if (NH == 0) return launch_col_mix_k<NH, 1, 256>(a, st);
    const int kt = rd_switch("RD_RAM_KT", RD_RAM_LDS_EXCLUSIVE ? 4 : 2);
    if (RD_RAM_LDS_EXCLUSIVE && kt == 4) return launch_col_mix_k<NH, 4, 1024>(a, st);
    if (kt == 1) return launch_col_mix_k<NH, 1, 256>(a, st);
    if (kt == 2) return launch_col_mix_k<NH, 2, 256>(a, st);
    return launch_col_mix_k<NH, 4, 256>(a, st);
}

template <typename T, int NW, int ROWS, int NT>
void launch_row_inv_r(const RamArgs& a, hipStream_t st) {
    const size_t lds = ram_lds_request((size_t)(ROWS * 2 * 2 + 1) * a.W * sizeof(float2));
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ram_row_inv_kernel<T, NW, ROWS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    rd_launch((ram_row_inv_kernel<T, NW, ROWS, NT>), dim3((a.H + ROWS - 1) / ROWS, a.B), dim3(NT), lds, st, a);
}
// One workgroup per CU: 1024 threads on 8 output rows (400 workgroups at 8 x 400 x 400) instead of five workgroups of 256 threads on 2 rows
template <typename T, int NW>
void launch_row_inv(const RamArgs& a, hipStream_t st) {
    if (NW == 0) return launch_row_inv_r<T, NW, 1, 256>(a, st);
    const int rows = rd_switch("RD_RAM_ROWS_INV", RD_RAM_LDS_EXCLUSIVE ? 8 : 2);
    if (RD_RAM_LDS_EXCLUSIVE && rows == 8) return launch_row_inv_r<T, NW, 8, 1024>(a, st);
    if (rows == 1) launch_row_inv_r<T, NW, 1, 256>(a, st);
    else if (rows == 4) launch_row_inv_r<T, NW, 4, 256>(a, st);
    else launch_row_inv_r<T, NW, 2, 256>(a, st);
}

#define RD_BY_SIDE(side, CALL)            \
    switch (side) {                       \
        case 256: CALL(256); break;       \
        case 384: CALL(384); break;       \
        case 400: CALL(400); break;       \
        case 512: CALL(512); break;       \
        default: CALL(0); break;          \
    }

}  // namespace

extern "C" {

int64_t rd_ram_workspace(int B, int H, int W, int b) {
    // rowspec [2B][3][H][KP] + colout [B][3][H][KP] complex64, KP = (b+1) padded to 4
    return (int64_t)3 * B * 3 * H * pad4(b + 1) * (int64_t)sizeof(float2);
}

int rd_ram_mix(const rd_ram_t* p, int dtype, void* stream) {
    if (!p || p->C != 3) return -1;
    RamArgs a;
    if (!make_plan(p->W, a.pw) || !make_plan(p->H, a.ph)) return -2;   // sizes must factor into 2,3,5
    if (p->b < 0 || 2 * p->b + 1 > p->H || 2 * p->b + 1 > p->W) return -3;
    if (!p->src || (!p->trg && !p->trg_amp) || !p->lam || !p->out_img || !p->out_freq || !p->workspace) return -4;
    a.src = p->src; a.trg = p->trg; a.trg_amp = p->trg_amp; a.lam = p->lam;
    a.out_img = p->out_img; a.out_freq = p->out_freq;
    a.B = p->B; a.H = p->H; a.W = p->W; a.b = p->b;
    a.nkeep = p->b + 1;
    a.KP = pad4(a.nkeep);
    a.rowspec = reinterpret_cast<float2*>(p->workspace);
    a.colout = a.rowspec + (size_t)2 * p->B * 3 * p->H * a.KP;
    a.tw_w = reinterpret_cast<const float2*>(p->tw_w);
    a.tw_h = reinterpret_cast<const float2*>(p->tw_h);
    a.amp_out = nullptr;
    a.nch = 3; a.planar = 0; a.src_u8 = p->src_u8 ? 1 : 0;
    a.cs = p->out_cstride > 0 ? p->out_cstride : 3;
    if (a.cs < 3) return -1;
    a.clip_lo = p->clip_lo; a.clip_hi = p->clip_hi; a.scale = p->scale; a.div = p->div; a.offset = p->offset;
    hipStream_t st = (hipStream_t)stream;
    const int nimg = p->trg_amp ? p->B : 2 * p->B;          // the partner's spectrum is only needed when it is given as an image
    // pass A: the kept bins of uint8 images as a matrix product when the caller has provided the coefficient tables (ram_dft.hip: 13 us
    // against 19 for the FFTs at 8 x 400 x 400; fp32 pixels need three bf16 terms each and measured slower, 22 us), else whole-row FFTs
    static const int dft_on = rd_switch("RD_RAM_DFT", 1);
    int e = 0;
    if (dft_on && a.src_u8 && p->dft_tables != nullptr && ram_dft_ok(p->H, p->W, p->b)) {
        e = ram_dft_row_fwd(a.src, a.trg, a.B, nimg, a.H, a.W, a.b, a.KP, a.rowspec, p->dft_tables, st);
        if (e) return e;
    } else {
#define RD_ROWF(S_) launch_row_fwd<S_>(a, nimg, st)
        RD_BY_SIDE(p->W, RD_ROWF)
#undef RD_ROWF
    }
#define RD_COLM(S_) e = launch_col_mix<S_>(a, st)
    RD_BY_SIDE(p->H, RD_COLM)
#undef RD_COLM
    if (e) return e;
    if (dtype == RD_BF16) {
#define RD_ROWI(S_) launch_row_inv<bf16_t, S_>(a, st)
        RD_BY_SIDE(p->W, RD_ROWI)
#undef RD_ROWI
    } else {
#define RD_ROWI(S_) launch_row_inv<float, S_>(a, st)
        RD_BY_SIDE(p->W, RD_ROWI)
#undef RD_ROWI
    }
    return (int)hipGetLastError();
}

int64_t rd_ram_amp_workspace(int C, int H, int W) { return (int64_t)C * H * pad4(W / 2 + 1) * (int64_t)sizeof(float2); }

int rd_ram_amp(const float* img_chw, float* amp_chw, int C, int H, int W, void* workspace, const float* tw_w, const float* tw_h,
               void* stream) {
    if (!img_chw || !amp_chw || !workspace || C < 1) return -1;
    RamArgs a = {};
    if (!make_plan(W, a.pw) || !make_plan(H, a.ph)) return -2;
    a.src = img_chw;
    a.B = C; a.H = H; a.W = W;
    a.nkeep = W / 2 + 1;
    a.KP = pad4(a.nkeep);
    a.rowspec = reinterpret_cast<float2*>(workspace);
    a.tw_w = reinterpret_cast<const float2*>(tw_w);
    a.tw_h = reinterpret_cast<const float2*>(tw_h);
    a.amp_out = amp_chw;
    a.nch = 1; a.planar = 1;
    hipStream_t st = (hipStream_t)stream;
    launch_row_fwd<0>(a, C, st);
    {
        const size_t lds = ram_lds_request((size_t)3 * H * sizeof(float2));
        static bool attr = false;
        if (!attr && lds > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ram_col_amp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        rd_launch(ram_col_amp_kernel, dim3(a.nkeep, C), dim3(256), lds, st, a);
    }
    return (int)hipGetLastError();
}

int rd_ram_mutate(const float* amp_src, const float* amp_trg, float* out, int C, int H, int W, int b, float lam, void* stream) {
    if (!amp_src || !amp_trg || !out || b < 0 || 2 * b + 1 > H || 2 * b + 1 > W) return -1;
    const size_t total = (size_t)C * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    rd_launch(ram_mutate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, amp_src, amp_trg, out, C, H, W, b, lam);
    return (int)hipGetLastError();
}

}  // extern "C"
