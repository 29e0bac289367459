// loss.hip -- fused segmentation / consistency / restoration losses (forward + dlogits) and Adam.
//
// Reference sites: code/train.py:246-259,265-283 (fundus), :412-451 (prostate), KD :85-88,
// code/utils/losses.py:8-33 (dice_loss, dice_loss_multi), torch.nn.BCELoss / CrossEntropyLoss /
// MSELoss / KLDivLoss semantics, torch.optim.Adam + poly LR code/train.py:573-576,289-293.
#include "common.h"
#include "../../include/ramdsir.h"

namespace {

constexpr int NS_MAX = 24;
constexpr int KMAX = 4;
constexpr int SL_U = 5;                  // pixels per thread whose loads are in flight together (seg-loss kernels, fundus)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// The fundus loss kernels are bound by their transcendental arithmetic (2.6 M (pixel, class) pairs x ~300 instructions of IEEE expf /
// logf / log1pf / division): the hardware exp2 / log2 / rcp (1 ulp) behind these cost ~10x less; the sums they feed are means over
// 2.6 M terms, far below the 1e-4 the loss values are held to (tests/test_gpu_ops.py::test_seg_loss_*, test_gpu_step.py)
__device__ __forceinline__ float fsigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float flog_(float x) { return __logf(x); }
__device__ __forceinline__ float fdiv_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// layout of the reduced sums
//  fundus  : 0 bce1 1 bce2 2 I1 3 Z1 4 I2 5 Z2 6 Y 7 cons
//  prostate: 0 ce1 1 ce2 2 cons, then for class i=1..K-1 at 3+5(i-1): I1 Z1 I2 Z2 Y
template <typename T>
__global__ __launch_bounds__(256) void seg_loss_sums_kernel(const rd_seg_loss_t p, int NS) {
    const int HW = p.H * p.W, K = p.K;
    const int npix = p.B * HW;
    const T* lg = reinterpret_cast<const T*>(p.logits);
    float acc[NS_MAX];
#pragma unroll
    for (int i = 0; i < NS_MAX; ++i) acc[i] = 0.f;
    // fundus: hardware transcendentals (above: 38.7 -> 19.1 us), and the loads of SL_U pixels issued before any arithmetic (clamped
    // index for the tail: branch-free loads, masked accumulation; on its own that changed nothing -- the kernel was ALU-bound)
    if (p.kind == 0 && K <= 2) {
        const float* mk = reinterpret_cast<const float*>(p.target);
        const int stride = gridDim.x * blockDim.x;
        for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < npix; i0 += stride * SL_U) {
            float tt[SL_U][2], a1[SL_U][2], a2[SL_U][2];
            bool ok[SL_U];
#pragma unroll
            for (int u = 0; u < SL_U; ++u) {
                const int i = i0 + u * stride;
                ok[u] = i < npix;
                const int ic = ok[u] ? i : npix - 1;
                const int n = ic / HW, pix = ic - n * HW;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int kk = k < K ? k : 0;
                    tt[u][k] = mk[((size_t)n * K + kk) * HW + pix];
                    a1[u][k] = to_f<T>(lg[(size_t)ic * K + kk]);
                    a2[u][k] = to_f<T>(lg[((size_t)(n + p.B) * HW + pix) * K + kk]);
                }
            }
#pragma unroll
            for (int u = 0; u < SL_U; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k >= K) continue;
                    const float t = tt[u][k];
                    const float p1 = fsigmoid_(a1[u][k]), p2 = fsigmoid_(a2[u][k]);
                    const float lp1 = flog_(p1), lp2 = flog_(p2);
                    acc[0] -= t * fmaxf(lp1, -100.f) + (1.f - t) * fmaxf(flog_(1.f - p1), -100.f);
                    acc[1] -= t * fmaxf(lp2, -100.f) + (1.f - t) * fmaxf(flog_(1.f - p2), -100.f);
                    acc[2] += p1 * t; acc[3] += p1 * p1;
                    acc[4] += p2 * t; acc[5] += p2 * p2;
                    acc[6] += t * t;
                    if (p.consistency == 1) acc[7] += (p1 - p2) * (lp1 - lp2);
                    else if (p.consistency == 2) acc[7] += (p2 - p1) * (p2 - p1);
                }
            }
        }
    } else
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int n = i / HW, pix = i - n * HW;
        const T* l1p = lg + (size_t)i * K;
        const T* l2p = lg + ((size_t)(n + p.B) * HW + pix) * K;
        if (p.kind == 0) {
            const float* mk = reinterpret_cast<const float*>(p.target);
            for (int k = 0; k < K; ++k) {
                const float t = mk[((size_t)n * K + k) * HW + pix];
                const float p1 = sigmoidf_(to_f<T>(l1p[k])), p2 = sigmoidf_(to_f<T>(l2p[k]));
                acc[0] -= t * fmaxf(logf(p1), -100.f) + (1.f - t) * fmaxf(log1pf(-p1), -100.f);
                acc[1] -= t * fmaxf(logf(p2), -100.f) + (1.f - t) * fmaxf(log1pf(-p2), -100.f);
                acc[2] += p1 * t; acc[3] += p1 * p1;
                acc[4] += p2 * t; acc[5] += p2 * p2;
                acc[6] += t * t;
                if (p.consistency == 1) acc[7] += (p1 - p2) * (logf(p1) - logf(p2));
                else if (p.consistency == 2) acc[7] += (p2 - p1) * (p2 - p1);
            }
        } else {
            const int64_t* tg = reinterpret_cast<const int64_t*>(p.target);
            const int t = (int)tg[i];
            float a1[KMAX], a2[KMAX], m1 = -3.0e38f, m2 = -3.0e38f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) { a1[k] = to_f<T>(l1p[k]); a2[k] = to_f<T>(l2p[k]); m1 = fmaxf(m1, a1[k]); m2 = fmaxf(m2, a2[k]); }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) { a1[k] = expf(a1[k] - m1); a2[k] = expf(a2[k] - m2); s1 += a1[k]; s2 += a2[k]; }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float p1 = a1[k] / s1, p2 = a2[k] / s2;
                    if (k == t) { acc[0] -= logf(p1); acc[1] -= logf(p2); }
                    if (p.consistency == 1) acc[2] += (p1 - p2) * (logf(p1) - logf(p2));
                    else if (p.consistency == 2) acc[2] += (p2 - p1) * (p2 - p1);
                    if (k >= 1) {
                        const float tk = (t == k) ? 1.f : 0.f;
                        float* a = acc + 3 + 5 * (k - 1);
                        a[0] += p1 * tk; a[1] += p1 * p1; a[2] += p2 * tk; a[3] += p2 * p2; a[4] += tk;
                    }
                }
        }
    }
    __shared__ float s[4][NS_MAX];
#pragma unroll
    for (int j = 0; j < NS_MAX; ++j) {
        const float v = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS_MAX)
        p.partial[(size_t)blockIdx.x * NS_MAX + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

// the per-block partial sums -> the loss terms (fp64, fixed order) + the sums the gradient pass needs
__global__ __launch_bounds__(1024) void seg_loss_final_kernel(const rd_seg_loss_t p, int nblocks) {
    __shared__ double sd[NS_MAX];
    __shared__ double sw[32][32];
    // thread (g = tid >> 5, j = tid & 31): sum j over the blocks g, g + 32, ... -- all loads independent (one round trip; the per-sum
    // loops of round 1 made 16 dependent trips), then sum j's 32 partial sums in a fixed order
    {
        const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
        double v = 0.0;
        if (j < NS_MAX)
            for (int b = g; b < nblocks; b += 32) v += (double)p.partial[(size_t)b * NS_MAX + j];
        sw[g][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS_MAX) {
        double s = 0.0;
        for (int g = 0; g < 32; ++g) s += sw[g][threadIdx.x];
        sd[threadIdx.x] = s;
        p.partial[(size_t)nblocks * NS_MAX + threadIdx.x] = (float)s;       // sums for the gradient pass
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double eps = 1e-5;
        double seg1, seg2, d1, d2, cons;
        if (p.kind == 0) {
            const double nel = (double)p.B * p.K * p.H * p.W;
            seg1 = sd[0] / nel; seg2 = sd[1] / nel; cons = sd[7] / nel;
            d1 = 1.0 - (2 * sd[2] + eps) / (sd[3] + sd[6] + eps);
            d2 = 1.0 - (2 * sd[4] + eps) / (sd[5] + sd[6] + eps);
        } else {
            const double npx = (double)p.B * p.H * p.W;
            seg1 = sd[0] / npx; seg2 = sd[1] / npx; cons = sd[2] / (npx * p.K);
            d1 = d2 = 0.0;
            for (int k = 1; k < p.K; ++k) {
                const double* a = sd + 3 + 5 * (k - 1);
                d1 += 1.0 - (2 * a[0] + eps) / (a[1] + a[4] + eps);
                d2 += 1.0 - (2 * a[2] + eps) / (a[3] + a[4] + eps);
            }
            d1 /= (p.K - 1); d2 /= (p.K - 1);
        }
        if (p.consistency == 0) cons = 0.0;
        p.losses_out[0] = (float)seg1; p.losses_out[1] = (float)d1; p.losses_out[2] = (float)seg2; p.losses_out[3] = (float)d2;
        p.losses_out[4] = (float)cons;
        p.losses_out[5] = (float)(seg1 + seg2 + d1 + d2 + p.cons_weight * cons);
        p.losses_out[6] = 0.f; p.losses_out[7] = 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_grad_kernel(const rd_seg_loss_t p, int nblocks) {
    const int HW = p.H * p.W, K = p.K, Ks = p.dlogits_cstride > 0 ? p.dlogits_cstride : p.K;
    const int npix = p.B * HW;
    const T* lg = reinterpret_cast<const T*>(p.logits);
    T* dl = reinterpret_cast<T*>(p.dlogits);
    const float* sm = p.partial + (size_t)nblocks * NS_MAX;
    const float eps = 1e-5f;
    const float w = p.cons_weight;
    if (p.kind == 0 && K <= 2) {
        // fundus: as in the sums kernel, the loads of SL_U pixels go out before any arithmetic
        const float nel = (float)p.B * K * HW;
        const float N1 = 2.f * sm[2] + eps, D1 = sm[3] + sm[6] + eps;
        const float N2 = 2.f * sm[4] + eps, D2 = sm[5] + sm[6] + eps;
        const float rnel = 1.f / nel, rD1 = 1.f / (D1 * D1), rD2 = 1.f / (D2 * D2);
        const float* mk = reinterpret_cast<const float*>(p.target);
        const int stride = gridDim.x * blockDim.x;
        for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < npix; i0 += stride * SL_U) {
            float tt[SL_U][2], a1[SL_U][2], a2[SL_U][2];
#pragma unroll
            for (int u = 0; u < SL_U; ++u) {
                const int i = i0 + u * stride;
                const int ic = i < npix ? i : npix - 1;
                const int n = ic / HW, pix = ic - n * HW;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int kk = k < K ? k : 0;
                    tt[u][k] = mk[((size_t)n * K + kk) * HW + pix];
                    a1[u][k] = to_f<T>(lg[(size_t)ic * K + kk]);
                    a2[u][k] = to_f<T>(lg[((size_t)(n + p.B) * HW + pix) * K + kk]);
                }
            }
#pragma unroll
            for (int u = 0; u < SL_U; ++u) {
                const int i = i0 + u * stride;
                if (i >= npix) continue;
                const int n = i / HW, pix = i - n * HW;
                const size_t e1 = (size_t)i * Ks, e2 = ((size_t)(n + p.B) * HW + pix) * Ks;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k >= K) continue;
                    const float t = tt[u][k];
                    const float p1 = fsigmoid_(a1[u][k]), p2 = fsigmoid_(a2[u][k]);
                    float g1 = fdiv_(p1 - t, fmaxf((1.f - p1) * p1, 1e-12f)) * rnel;    // ATen binary_cross_entropy_backward
                    float g2 = fdiv_(p2 - t, fmaxf((1.f - p2) * p2, 1e-12f)) * rnel;
                    g1 += -(2.f * t * D1 - N1 * 2.f * p1) * rD1;
                    g2 += -(2.f * t * D2 - N2 * 2.f * p2) * rD2;
                    if (p.consistency == 1) {
                        const float dlg = flog_(p1) - flog_(p2);
                        g1 += w * (dlg + fdiv_(p1 - p2, p1)) * rnel;
                        g2 += w * (-dlg - fdiv_(p1 - p2, p2)) * rnel;
                    } else if (p.consistency == 2) {
                        g1 += w * (-2.f * (p2 - p1)) * rnel;
                        g2 += w * (2.f * (p2 - p1)) * rnel;
                    }
                    dl[e1 + k] = from_f<T>(g1 * p1 * (1.f - p1));
                    dl[e2 + k] = from_f<T>(g2 * p2 * (1.f - p2));
                }
            }
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int n = i / HW, pix = i - n * HW;
        const size_t o1 = (size_t)i * K, o2 = ((size_t)(n + p.B) * HW + pix) * K;
        const size_t e1 = (size_t)i * Ks, e2 = ((size_t)(n + p.B) * HW + pix) * Ks;
        if (p.kind == 0) {
            const float nel = (float)p.B * K * HW;
            const float N1 = 2.f * sm[2] + eps, D1 = sm[3] + sm[6] + eps;
            const float N2 = 2.f * sm[4] + eps, D2 = sm[5] + sm[6] + eps;
            const float* mk = reinterpret_cast<const float*>(p.target);
            for (int k = 0; k < K; ++k) {
                const float t = mk[((size_t)n * K + k) * HW + pix];
                const float p1 = sigmoidf_(to_f<T>(lg[o1 + k])), p2 = sigmoidf_(to_f<T>(lg[o2 + k]));
                float g1 = (p1 - t) / fmaxf((1.f - p1) * p1, 1e-12f) / nel;         // ATen binary_cross_entropy_backward
                float g2 = (p2 - t) / fmaxf((1.f - p2) * p2, 1e-12f) / nel;
                g1 += -(2.f * t * D1 - N1 * 2.f * p1) / (D1 * D1);
                g2 += -(2.f * t * D2 - N2 * 2.f * p2) / (D2 * D2);
                if (p.consistency == 1) {
                    const float dlg = logf(p1) - logf(p2);
                    g1 += w * (dlg + (p1 - p2) / p1) / nel;
                    g2 += w * (-dlg - (p1 - p2) / p2) / nel;
                } else if (p.consistency == 2) {
                    g1 += w * (-2.f * (p2 - p1)) / nel;
                    g2 += w * (2.f * (p2 - p1)) / nel;
                }
                dl[e1 + k] = from_f<T>(g1 * p1 * (1.f - p1));
                dl[e2 + k] = from_f<T>(g2 * p2 * (1.f - p2));
            }
        } else {
            const float npx = (float)p.B * HW, nel = npx * K;
            const int64_t* tg = reinterpret_cast<const int64_t*>(p.target);
            const int t = (int)tg[i];
            float a1[KMAX], a2[KMAX], q1[KMAX], q2[KMAX], m1 = -3.0e38f, m2 = -3.0e38f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) { a1[k] = to_f<T>(lg[o1 + k]); a2[k] = to_f<T>(lg[o2 + k]); m1 = fmaxf(m1, a1[k]); m2 = fmaxf(m2, a2[k]); }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) { a1[k] = expf(a1[k] - m1); a2[k] = expf(a2[k] - m2); s1 += a1[k]; s2 += a2[k]; }
            float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float p1 = a1[k] / s1, p2 = a2[k] / s2;
                    a1[k] = p1; a2[k] = p2;
                    float g1 = 0.f, g2 = 0.f;
                    if (k >= 1) {
                        const float* a = sm + 3 + 5 * (k - 1);
                        const float tk = (t == k) ? 1.f : 0.f;
                        const float Na = 2.f * a[0] + eps, Da = a[1] + a[4] + eps;
                        const float Nb = 2.f * a[2] + eps, Db = a[3] + a[4] + eps;
                        g1 += -(2.f * tk * Da - Na * 2.f * p1) / (Da * Da) / (float)(K - 1);
                        g2 += -(2.f * tk * Db - Nb * 2.f * p2) / (Db * Db) / (float)(K - 1);
                    }
                    if (p.consistency == 1) {
                        const float dlg = logf(p1) - logf(p2);
                        g1 += w * (dlg + (p1 - p2) / p1) / nel;
                        g2 += w * (-dlg - (p1 - p2) / p2) / nel;
                    } else if (p.consistency == 2) {
                        g1 += w * (-2.f * (p2 - p1)) / nel;
                        g2 += w * (2.f * (p2 - p1)) / nel;
                    }
                    q1[k] = g1; q2[k] = g2;
                    dot1 += p1 * g1; dot2 += p2 * g2;
                }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float tk = (t == k) ? 1.f : 0.f;
                    dl[e1 + k] = from_f<T>(a1[k] * (q1[k] - dot1) + (a1[k] - tk) / npx);
                    dl[e2 + k] = from_f<T>(a2[k] * (q2[k] - dot2) + (a2[k] - tk) / npx);
                }
        }
    }
}

// ----------------------------------------------------------------------------------- restoration loss
template <typename T>
__global__ __launch_bounds__(256) void rec_loss_kernel(const T* lg, const T* tgt, T* dl, float* partial, int per_img,
                                                       GroupMap gm, float lambda_rec, int C, int Ts, int Ds) {
    const int n = blockIdx.y, g = group_of(gm, n);
    const float cnt = (float)(gm.gs[g + 1] - gm.gs[g]) * (float)per_img;
    const size_t base = (size_t)n * per_img;
    float acc = 0.f;
    // tanh through the hardware exp2 / rcp (as the fundus seg-loss kernels, above): 1 - 2 / (exp(2x) + 1), exact limits at +-inf
    const float gscale = lambda_rec * 2.f / cnt;
    const float rC = 1.f / (float)C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_img; i += gridDim.x * blockDim.x) {
        int pix = (int)((float)i * rC);                    // i / C without the integer-division sequence (per_img < 2^24), corrected
        pix += (i - pix * C >= C) ? 1 : 0;
        pix -= (i - pix * C < 0) ? 1 : 0;
        const int c = i - pix * C;
        const size_t px = (size_t)n * (per_img / C) + pix;
        const float x = to_f<T>(lg[base + i]);
        const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);
        const float d = r - to_f<T>(tgt[px * Ts + c]);
        acc += d * d;
        dl[px * Ds + c] = from_f<T>(gscale * d * (1.f - r * r));
    }
    __shared__ float s[4];
    const float v = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[n * gridDim.x + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// one wave per group: lanes stride over the per-block partial sums (fixed order -> deterministic), fp64 wave reduce
__global__ __launch_bounds__(64 * RD_MAX_GROUPS) void rec_loss_final_kernel(const float* partial, float* mse_out, int bx, int per_img, GroupMap gm) {
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (g >= gm.G) return;
    const int lo = gm.gs[g] * bx, hi = gm.gs[g + 1] * bx;
    double s = 0.0;
    for (int i = lo + lane; i < hi; i += 64) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) mse_out[g] = (float)(s / ((double)(gm.gs[g + 1] - gm.gs[g]) * per_img));
}

// ----------------------------------------------------------------------------------- Adam
__global__ void adam_prepare_kernel(const rd_adam_t p) {
    const int it = *p.iter;
    // LR in force at iteration `it` was written after iteration it-1 from ITS iter_num (train.py:289)
    double lr = p.base_lr;
    if (it > 0) lr = (double)p.base_lr * pow(1.0 - (double)(it - 1) / (double)p.total_iters, 0.9);
    const double t = (double)(it + 1);
    p.hyper_out[0] = (float)lr;
    p.hyper_out[1] = (float)(1.0 - pow((double)p.beta1, t));
    p.hyper_out[2] = (float)(1.0 - pow((double)p.beta2, t));
    p.hyper_out[3] = (float)it;
    *p.iter = it + 1;
}

__global__ __launch_bounds__(256) void adam_update_kernel(const rd_adam_t p) {
    const float lr = p.hyper_out[0], bc1 = p.hyper_out[1], bc2 = p.hyper_out[2];
    const float sq_bc2 = sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = p.grad[i];
        const float m = p.beta1 * p.exp_avg[i] + (1.f - p.beta1) * g;
        const float v = p.beta2 * p.exp_avg_sq[i] + (1.f - p.beta2) * g * g;
        p.exp_avg[i] = m;
        p.exp_avg_sq[i] = v;
        const float step = (i < p.n_half_lr ? 0.5f * lr : lr) / bc1;
        const float denom = sqrtf(v) / sq_bc2 + p.eps;
        p.param[i] -= step * (m / denom);
    }
}

// rd_zero: every range in one launch (16-byte stores; the last bytes of a range whose size is not a multiple of 16 one by one)
constexpr int ZERO_MAX_RANGES = 8;
struct ZeroArgs {
    void* p[ZERO_MAX_RANGES];
    unsigned long long bytes[ZERO_MAX_RANGES];
    int n;
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(const ZeroArgs a) {
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int r = 0; r < a.n; ++r) {
        uint4* q = reinterpret_cast<uint4*>(a.p[r]);
        const size_t nv = a.bytes[r] / 16;
        for (size_t i = t0; i < nv; i += stride) q[i] = make_uint4(0, 0, 0, 0);
        if (blockIdx.x == 0) {
            unsigned char* b = reinterpret_cast<unsigned char*>(a.p[r]);
            for (size_t i = nv * 16 + threadIdx.x; i < a.bytes[r]; i += 256) b[i] = 0;
        }
    }
}

GroupMap host_gm2(int G, const int32_t* gs) {
    GroupMap gm;
    gm.G = G;
    for (int i = 0; i <= RD_MAX_GROUPS; ++i) gm.gs[i] = (gs && i <= G) ? gs[i] : 0;
    return gm;
}

int seg_blocks(const rd_seg_loss_t* p) {
    const int npix = p->B * p->H * p->W;
    int b = (npix + 255) / 256;
    if (b > 1024) b = 1024;
    return b < 1 ? 1 : b;
}

int rec_bx(int per_img) {
    int b = (per_img + 1023) / 1024;
    if (b > 256) b = 256;
    return b < 1 ? 1 : b;
}

}  // namespace

extern "C" {

int64_t rd_seg_loss_workspace(const rd_seg_loss_t* p) { return (int64_t)(seg_blocks(p) + 1) * NS_MAX * sizeof(float); }

int rd_seg_loss(const rd_seg_loss_t* p, int dtype, void* stream) {
    if (!p || p->K < 1 || p->K > KMAX || (p->kind == 1 && p->K < 2)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int nb = seg_blocks(p);
    const int NS = p->kind == 0 ? 8 : 3 + 5 * (p->K - 1);
    if (dtype == RD_BF16) rd_launch(seg_loss_sums_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, *p, NS);
    else rd_launch(seg_loss_sums_kernel<float>, dim3(nb), dim3(256), 0, st, *p, NS);
    rd_launch(seg_loss_final_kernel, dim3(1), dim3(1024), 0, st, *p, nb);
    if (p->dlogits) {
        if (dtype == RD_BF16) rd_launch(seg_loss_grad_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, *p, nb);
        else rd_launch(seg_loss_grad_kernel<float>, dim3(nb), dim3(256), 0, st, *p, nb);
    }
    return (int)hipGetLastError();
}

int64_t rd_rec_loss_workspace(int B, int H, int W, int C) { return (int64_t)B * rec_bx(H * W * C) * sizeof(float); }

int rd_rec_loss(const void* lg, const void* tgt, void* dl, float* mse_out, float* partial_ws, int B, int H, int W, int C,
                int target_cstride, int dlogits_cstride, int G, const int32_t* gstart_host, float lambda_rec, int dtype, void* stream) {
    if (G < 1 || G > RD_MAX_GROUPS) return -1;
    const int Ts = target_cstride > 0 ? target_cstride : C, Ds = dlogits_cstride > 0 ? dlogits_cstride : C;
    if (Ts < C || Ds < C) return -1;
    hipStream_t st = (hipStream_t)stream;
    // rec_loss_kernel splits an element index into (pixel, channel) with a float reciprocal + one correction step: exact while the
    // index stays below 2^24 (a 2048 x 2048 x 3 image is 12.6 M); larger images are refused rather than indexed wrongly
    if ((long long)H * W * C >= (1ll << 24)) return -1;
    const int per_img = H * W * C;
    const int bx = rec_bx(per_img);
    const GroupMap gm = host_gm2(G, gstart_host);
    if (dtype == RD_BF16)
        rd_launch(rec_loss_kernel<bf16_t>, dim3(bx, B), dim3(256), 0, st, (const bf16_t*)lg, (const bf16_t*)tgt, (bf16_t*)dl,
                           partial_ws, per_img, gm, lambda_rec, C, Ts, Ds);
    else
        rd_launch(rec_loss_kernel<float>, dim3(bx, B), dim3(256), 0, st, (const float*)lg, (const float*)tgt, (float*)dl,
                           partial_ws, per_img, gm, lambda_rec, C, Ts, Ds);
    rd_launch(rec_loss_final_kernel, dim3(1), dim3(64 * G), 0, st, partial_ws, mse_out, bx, per_img, gm);
    return (int)hipGetLastError();
}

int rd_adam_step(const rd_adam_t* p, void* stream) {
    if (!p || p->n < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    rd_launch(adam_prepare_kernel, dim3(1), dim3(1), 0, st, *p);
    int64_t nb = (p->n + 255) / 256;
    if (nb > 4096) nb = 4096;
    rd_launch(adam_update_kernel, dim3((int)nb), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}

// optimizer.zero_grad() (train.py:285,454) + the per-step reset of the BatchNorm sum buffers: n device ranges set to zero bytes on
// `stream`.  Up to ZERO_MAX_RANGES 16-byte-aligned ranges go in ONE kernel launch (the range table travels in the kernel arguments); as
// one hipMemsetAsync per range the step began with three fill kernels of the runtime back to back, 5 us each for 10 MB, on the
// critical lane (scripts/r6/lane_gaps.py).  More ranges, or an unaligned pointer: the runtime's fills as before.
int rd_zero(void* const* ptrs_host, const int64_t* bytes_host, int n, void* stream) {
    if (n < 0 || (n > 0 && (!ptrs_host || !bytes_host))) return -1;
    ZeroArgs a;
    a.n = 0;
    bool one_launch = true;
    for (int i = 0; i < n && one_launch; ++i) {
        if (!ptrs_host[i] || bytes_host[i] <= 0) continue;
        if (a.n == ZERO_MAX_RANGES || ((uintptr_t)ptrs_host[i] & 15)) { one_launch = false; break; }
        a.p[a.n] = ptrs_host[i];
        a.bytes[a.n] = (unsigned long long)bytes_host[i];
        ++a.n;
    }
    if (one_launch) {
        if (a.n == 0) return 0;
        rd_launch(zero_ranges_kernel, dim3(rd_num_cus() * 8), dim3(256), 0, (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
    for (int i = 0; i < n; ++i) {
        if (!ptrs_host[i] || bytes_host[i] <= 0) continue;
        RD_CHECK(hipMemsetAsync(ptrs_host[i], 0, (size_t)bytes_host[i], (hipStream_t)stream));
    }
    return 0;
}

}  // extern "C"
