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
//   A  row pass     one workgroup per image row: Stockham FFT (radix 4/2/3/5, LDS ping-pong) of the
//                   3 channels, keep bins kx = 0..b                                  -> rowspec[img][c][kx][y]
//   B  column pass  one workgroup per (sample, channel, kx): FFT of the src and trg columns, mix on the
//                   2b+1 bins, inverse FFT of the zero-padded column                   -> colout[n][c][kx][y]
//   C  row inverse  one workgroup per output row: Hermitian row spectrum (2b+1 bins) -> real row, + src,
//                   clip, scale, store both network inputs (img, img_freq) as NHWC dtype.
// All loads of image rows / spectrum columns are contiguous; butterflies run out of LDS.
#include "common.h"
#include "../../include/ramdsir.h"

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

// `batch` independent length-N Stockham autosort FFTs in LDS.  Transform t ping-pongs between
// buf[(t*2+0)*N ..] and buf[(t*2+1)*N ..]; returns which half holds the result (same for every t).
// tw[k] = (cos 2*pi*k/N, -sin 2*pi*k/N): the fp64-accurate table from the host, copied into LDS by the caller.
__device__ int fft_stockham(float2* buf, int batch, const FftPlan& pl, const float2* tw, bool inverse, int tid, int nthreads) {
    const int N = pl.N;
    const float sgn = inverse ? 1.f : -1.f;
    int Ns = 1, cur = 0;
    for (int st = 0; st < pl.nst; ++st) {
        const int R = pl.radix[st];
        const int nb = N / R;
        const int tstep = N / (Ns * R);
        // integer quotients by the run-time stage constants through float reciprocals: exact for these ranges
        // (a < 2^12, divisor <= 1024: |rounding| ~ 1e-4 of the 0.5/divisor margin), a fraction of the cost of a division
        const float inv_nb = 1.0f / (float)nb, inv_Ns = 1.0f / (float)Ns;
        for (int jj = tid; jj < batch * nb; jj += nthreads) {
            const int t = (int)(((float)jj + 0.5f) * inv_nb), j = jj - t * nb;
            const float2* a = buf + (size_t)(t * 2 + cur) * N;
            float2* o = buf + (size_t)(t * 2 + (cur ^ 1)) * N;
            const int q = (int)(((float)j + 0.5f) * inv_Ns);
            const int k = j - q * Ns;
            const int j0 = q * Ns * R + k;
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
            if (R == 2) {
                o[j0] = cadd(v[0], v[1]);
                o[j0 + Ns] = csub(v[0], v[1]);
            } else if (R == 4) {
                const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = imul(csub(v[1], v[3]), sgn);
                o[j0] = cadd(t0, t2);
                o[j0 + Ns] = cadd(t1, t3);
                o[j0 + 2 * Ns] = csub(t0, t2);
                o[j0 + 3 * Ns] = csub(t1, t3);
            } else if (R == 3) {
                const float2 t1 = cadd(v[1], v[2]);
                const float2 t2 = csub(v[0], cscale(t1, 0.5f));
                const float2 t3 = imul(cscale(csub(v[1], v[2]), 0.86602540378443864676f), sgn);
                o[j0] = cadd(v[0], t1);
                o[j0 + Ns] = cadd(t2, t3);
                o[j0 + 2 * Ns] = csub(t2, t3);
            } else {  // R == 5
                const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
                const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
                const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
                const float2 m1 = cadd(v[0], cadd(cscale(a1, c1), cscale(a2, c2)));
                const float2 m2 = cadd(v[0], cadd(cscale(a1, c2), cscale(a2, c1)));
                const float2 n1 = imul(cadd(cscale(b1, s1), cscale(b2, s2)), sgn);
                const float2 n2 = imul(csub(cscale(b1, s2), cscale(b2, s1)), sgn);
                o[j0] = cadd(v[0], cadd(a1, a2));
                o[j0 + Ns] = cadd(m1, n1);
                o[j0 + 4 * Ns] = csub(m1, n1);
                o[j0 + 2 * Ns] = cadd(m2, n2);
                o[j0 + 3 * Ns] = csub(m2, n2);
            }
        }
        __syncthreads();
        cur ^= 1;
        Ns *= R;
    }
    return cur;
}

struct RamArgs {
    const float* src; const float* trg; const float* lam;
    void* out_img; void* out_freq;
    float2* rowspec; float2* colout;
    const float2* tw_w; const float2* tw_h;
    int B, H, W, b, cs;
    float clip_lo, clip_hi, scale, offset;
    FftPlan pw, ph;
};

// A: grid (H, 2B)  images [0,B) = src batch, [B,2B) = trg batch.  One transform per channel (zero imaginary
// part) rather than a packed pair: an all-zero channel must give an EXACTLY zero spectrum, because the mix
// divides by |F_src| (the reference's angle()==0 branch, fundus.py:48 on a constant-zero plane).
__global__ __launch_bounds__(256) void ram_row_fwd_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_);        // [3 transforms][2][W], then the twiddle table [W]
    const int y = blockIdx.x, n = blockIdx.y, W = a.W, H = a.H;
    float2* s_tw = buf + 6 * W;                            // twiddles out of LDS: a butterfly no longer waits on L2 for them
    for (int i = threadIdx.x; i < W; i += blockDim.x) s_tw[i] = a.tw_w[i];
    const float* img = (n < a.B ? a.src + (size_t)n * H * W * 3 : a.trg + (size_t)(n - a.B) * H * W * 3) + (size_t)y * W * 3;
    for (int i = threadIdx.x; i < 3 * W; i += blockDim.x) {
        const int x = i / 3, c = i - 3 * x;
        buf[(size_t)c * 2 * W + x] = make_float2(img[i], 0.f);
    }
    __syncthreads();
    const int cur = fft_stockham(buf, 3, a.pw, s_tw, false, threadIdx.x, blockDim.x);
    const int nb1 = a.b + 1;
    for (int i = threadIdx.x; i < 3 * nb1; i += blockDim.x) {
        const int c = i / nb1, kx = i - c * nb1;
        a.rowspec[((size_t)(n * 3 + c) * nb1 + kx) * H + y] = buf[(size_t)(c * 2 + cur) * W + kx];
    }
}

// B: grid (b+1, 3, B)
__global__ __launch_bounds__(256) void ram_col_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_);        // [2 transforms][2][H], then the twiddle table [H]
    const int kx = blockIdx.x, c = blockIdx.y, n = blockIdx.z, H = a.H, nb1 = a.b + 1;
    float2* s_tw = buf + 4 * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) s_tw[i] = a.tw_h[i];
    const float2* cs = a.rowspec + ((size_t)(n * 3 + c) * nb1 + kx) * H;
    const float2* ct = a.rowspec + ((size_t)((n + a.B) * 3 + c) * nb1 + kx) * H;
    for (int y = threadIdx.x; y < H; y += blockDim.x) {
        buf[y] = cs[y];
        buf[2 * H + y] = ct[y];
    }
    __syncthreads();
    const int cur = fft_stockham(buf, 2, a.ph, s_tw, false, threadIdx.x, blockDim.x);
    const float2* fs = buf + cur * H;
    const float2* ft = buf + (2 + cur) * H;
    float2* din = buf + (cur ^ 1) * H;                      // free half of transform 0 becomes the inverse input
    const float lam = a.lam[n];
    for (int y = threadIdx.x; y < H; y += blockDim.x) din[y] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.b + 1; i += blockDim.x) {
        const int ky = i - a.b;
        const int idx = (ky + H) % H;
        const float2 Fs = fs[idx], Ft = ft[idx];
        const float As = sqrtf(Fs.x * Fs.x + Fs.y * Fs.y), At = sqrtf(Ft.x * Ft.x + Ft.y * Ft.y);
        din[idx] = As > 0.f ? cscale(Fs, (1.f - lam) * (At / As - 1.f)) : make_float2((1.f - lam) * At, 0.f);
    }
    __syncthreads();
    // inverse FFT of din (it sits in half cur^1 of transform 0: run a batch of 1 starting from that half)
    float2* ibuf = buf;                                    // transform 0's two halves
    // fft_stockham starts from half 0: if din is in half 1, swap roles by offsetting through a copy-free trick:
    // copy din to half 0 when needed (H <= 1024 elements, one pass)
    if ((cur ^ 1) == 1) {
        for (int y = threadIdx.x; y < H; y += blockDim.x) ibuf[y] = din[y];
        __syncthreads();
    }
    const int c2 = fft_stockham(ibuf, 1, a.ph, s_tw, true, threadIdx.x, blockDim.x);
    const float2* yv = ibuf + c2 * H;
    float2* out = a.colout + ((size_t)(n * 3 + c) * nb1 + kx) * H;
    for (int y = threadIdx.x; y < H; y += blockDim.x) out[y] = yv[y];
}

// C: grid (H, B)
template <typename T>
__global__ __launch_bounds__(256) void ram_row_inv_kernel(const RamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    float2* buf = reinterpret_cast<float2*>(smem_);        // [2 transforms][2][W], then the twiddle table [W]
    const int y = blockIdx.x, n = blockIdx.y, W = a.W, H = a.H, nb1 = a.b + 1;
    float2* s_tw = buf + 4 * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        buf[x] = make_float2(0.f, 0.f);
        buf[2 * W + x] = make_float2(0.f, 0.f);
        s_tw[x] = a.tw_w[x];
    }
    __syncthreads();
    for (int kx = threadIdx.x; kx < nb1; kx += blockDim.x) {
        const float2 R0 = a.colout[((size_t)(n * 3 + 0) * nb1 + kx) * H + y];
        const float2 R1 = a.colout[((size_t)(n * 3 + 1) * nb1 + kx) * H + y];
        const float2 R2 = a.colout[((size_t)(n * 3 + 2) * nb1 + kx) * H + y];
        buf[kx] = make_float2(R0.x - R1.y, R0.y + R1.x);                 // R0 + i R1
        buf[2 * W + kx] = R2;
        if (kx > 0) {
            buf[W - kx] = make_float2(R0.x + R1.y, -R0.y + R1.x);        // conj(R0) + i conj(R1)
            buf[2 * W + W - kx] = make_float2(R2.x, -R2.y);
        }
    }
    __syncthreads();
    const int cur = fft_stockham(buf, 2, a.pw, s_tw, true, threadIdx.x, blockDim.x);
    const float2* r01 = buf + cur * W;
    const float2* r2 = buf + (2 + cur) * W;
    const float inv = 1.0f / ((float)H * (float)W);
    const float* srow = a.src + ((size_t)n * H + y) * W * 3;
    T* oi = reinterpret_cast<T*>(a.out_img) + ((size_t)n * H + y) * W * a.cs;
    T* of = reinterpret_cast<T*>(a.out_freq) + ((size_t)n * H + y) * W * a.cs;
    for (int i = threadIdx.x; i < 3 * W; i += blockDim.x) {
        const int x = i / 3, c = i - 3 * x;
        const float corr = (c == 0 ? r01[x].x : (c == 1 ? r01[x].y : r2[x].x)) * inv;
        const float s = srow[i];
        float f = s + corr;
        f = fminf(fmaxf(f, a.clip_lo), a.clip_hi);
        oi[x * a.cs + c] = from_f<T>(s * a.scale + a.offset);
        of[x * a.cs + c] = from_f<T>(f * a.scale + a.offset);
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
    return n == 1 && p.nst <= 12;
}

}  // namespace

extern "C" {

int64_t rd_ram_workspace(int B, int H, int W, int b) {
    // rowspec [2B][3][b+1][H] + colout [B][3][b+1][H] complex64
    return (int64_t)3 * B * 3 * (b + 1) * H * (int64_t)sizeof(float2);
}

int rd_ram_mix(const rd_ram_t* p, int dtype, void* stream) {
    if (!p || p->C != 3) return -1;
    RamArgs a;
    if (!make_plan(p->W, a.pw) || !make_plan(p->H, a.ph)) return -2;   // sizes must factor into 2,3,5
    if (p->b < 0 || 2 * p->b + 1 > p->H || 2 * p->b + 1 > p->W) return -3;
    a.src = p->src; a.trg = p->trg; a.lam = p->lam;
    a.out_img = p->out_img; a.out_freq = p->out_freq;
    a.rowspec = reinterpret_cast<float2*>(p->workspace);
    a.colout = a.rowspec + (size_t)2 * p->B * 3 * (p->b + 1) * p->H;
    a.tw_w = reinterpret_cast<const float2*>(p->tw_w);
    a.tw_h = reinterpret_cast<const float2*>(p->tw_h);
    a.B = p->B; a.H = p->H; a.W = p->W; a.b = p->b;
    a.cs = p->out_cstride > 0 ? p->out_cstride : 3;
    if (a.cs < 3) return -1;
    a.clip_lo = p->clip_lo; a.clip_hi = p->clip_hi; a.scale = p->scale; a.offset = p->offset;
    hipStream_t st = (hipStream_t)stream;
    const size_t lw = (size_t)5 * p->W * sizeof(float2), lh = (size_t)5 * p->H * sizeof(float2);      // + twiddle table
    hipLaunchKernelGGL(ram_row_fwd_kernel, dim3(p->H, 2 * p->B), dim3(256), (size_t)7 * p->W * sizeof(float2), st, a);
    hipLaunchKernelGGL(ram_col_kernel, dim3(p->b + 1, 3, p->B), dim3(256), lh, st, a);
    if (dtype == RD_BF16) hipLaunchKernelGGL(ram_row_inv_kernel<bf16_t>, dim3(p->H, p->B), dim3(256), lw, st, a);
    else hipLaunchKernelGGL(ram_row_inv_kernel<float>, dim3(p->H, p->B), dim3(256), lw, st, a);
    return (int)hipGetLastError();
}

}  // extern "C"
