// bn.hip -- BatchNorm finalize (forward / backward), upsample-side statistics and backward, and the
// NCHW fp32 <-> NHWC layout kernels at the module boundary.
//
// Reference sites: nn.BatchNorm2d via normalization('bn') code/networks/unet.py:17-28,
// DomainSpecificBatchNorm2d code/networks/dsbn.py:24-27, nn.Upsample(bilinear, x2) unet.py:84,127.
#include "common.h"
#include "../../include/ramdsir.h"
#include "bn_fin.h"

namespace {

// One workgroup per channel, one wave per BN group: lane k sums slot k of the RD_STAT_SLOTS partial sums.  Every global
// load a group needs (statistics, gamma/beta, running statistics) is issued before the first dependent instruction,
// so a launch costs one memory round trip; the old form (one wave walking the groups, loads behind the reduction)
// paid 2 x G of them, on the critical path of the step 76 times.  Groups that share one BatchNorm (the two passes of
// the seg network) update its running statistics in group order, exactly as two consecutive module calls would:
// thread 0 replays the momentum recursion from the per-group results in LDS.
__global__ __launch_bounds__(64 * RD_MAX_GROUPS) void bn_finalize_fwd_kernel(const rd_bn_fwd_t p) {
    __shared__ float s_mean[RD_MAX_GROUPS], s_unb[RD_MAX_GROUPS], s_rm[RD_MAX_GROUPS], s_rv[RD_MAX_GROUPS];
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the counters thread 0 of channel 0 bumps at the very end are requested NOW, beside the statistics loads: read there they would be a
    // second memory round trip at the tail of the one workgroup every launch waits for (38 launches: 302 -> 270 us alone)
    long long nbt_v[RD_MAX_GROUPS];
    if (c == 0 && threadIdx.x == 0 && p.training) {
#pragma unroll
        for (int i = 0; i < RD_MAX_GROUPS; ++i) nbt_v[i] = (i < p.G && p.num_batches_tracked[i]) ? *p.num_batches_tracked[i] : 0;
    }
    double s1 = 0.0, s2 = 0.0;
    if (p.training) {
        for (int k = lane; k < RD_STAT_SLOTS; k += 64) {
            s1 += p.stats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 0];
            s2 += p.stats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 1];
        }
    }
    const float cbias = p.conv_bias ? p.conv_bias[c] : 0.f;      // the sums are those of (conv result - bias)
    const float gam = p.gamma[g][c], bet = p.beta[g][c];
    const bool has_run = p.running_mean[g] != nullptr;
    float rm = 0.f, rv = 1.f;
    if (has_run) {
        rm = p.running_mean[g][c];
        rv = p.running_var[g][c];
    }
    float mean, invstd, unb = 0.f;
    if (p.training) {
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        const rdfin::FwdStat r = rdfin::fwd_stat(s1, s2, p.count[g], cbias, p.eps);     // shared with the folded finalize (bn_fin.h)
        mean = r.mean;
        invstd = r.invstd;
        unb = r.unb;
    } else {
        mean = rm;
        invstd = 1.0f / sqrtf(rv + p.eps);
    }
    if (lane == 0) {
        rdfin::FwdStat r;
        r.mean = mean; r.invstd = invstd; r.unb = unb;
        float sc, sh;
        rdfin::fwd_coef(gam, bet, r, sc, sh);
        p.scale[g * p.C + c] = sc;
        p.shift[g * p.C + c] = sh;
        p.mean[g * p.C + c] = mean;
        p.invstd[g * p.C + c] = invstd;
        s_mean[g] = mean; s_unb[g] = unb; s_rm[g] = rm; s_rv[g] = rv;
    }
    if (!p.training) return;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int i = 0; i < p.G; ++i) {
        if (!p.running_mean[i]) continue;
        const float nm = rdfin::momentum_step(s_rm[i], s_mean[i], p.momentum);
        const float nv = rdfin::momentum_step(s_rv[i], s_unb[i], p.momentum);
        p.running_mean[i][c] = nm;
        p.running_var[i][c] = nv;
        for (int j = i + 1; j < p.G; ++j)                  // a later group on the same BatchNorm continues from here
            if (p.running_mean[j] == p.running_mean[i]) { s_rm[j] = nm; s_rv[j] = nv; }
    }
    if (c == 0) {
        for (int i = 0; i < p.G; ++i) {
            if (!p.num_batches_tracked[i]) continue;
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && p.num_batches_tracked[j] != p.num_batches_tracked[i];
            if (!first) continue;
            int n = 0;
            for (int j = i; j < p.G; ++j) n += p.num_batches_tracked[j] == p.num_batches_tracked[i] ? 1 : 0;
            long long v = 0;
#pragma unroll
            for (int j = 0; j < RD_MAX_GROUPS; ++j) v = j == i ? nbt_v[j] : v;       // static indexing: nbt_v stays in registers
            *p.num_batches_tracked[i] = v + n;
        }
    }
}

__global__ __launch_bounds__(64 * RD_MAX_GROUPS) void bn_finalize_bwd_kernel(const rd_bn_bwd_t p) {
    __shared__ float s_s1[RD_MAX_GROUPS], s_s2[RD_MAX_GROUPS];
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s1d = 0.0, sgzd = 0.0;
    for (int k = lane; k < RD_STAT_SLOTS; k += 64) {
        s1d += p.bstats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 0];
        sgzd += p.bstats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 1];
    }
    const float mu = p.mean[g * p.C + c], is = p.invstd[g * p.C + c], gam = p.gamma[g][c];
    s1d = wave_sum_d(s1d);
    sgzd = wave_sum_d(sgzd);
    if (lane == 0) {
        const rdfin::BwdStat r = rdfin::bwd_stat(s1d, sgzd, mu, is);                     // shared with the folded finalize (bn_fin.h)
        float P, Q, R;
        rdfin::bwd_coef(gam, is, mu, r, p.count[g], P, Q, R);
        p.P[g * p.C + c] = P;
        p.Q[g * p.C + c] = Q;
        p.R[g * p.C + c] = R;
        s_s1[g] = r.s1; s_s2[g] = r.s2;
    }
    __syncthreads();
    // dgamma / dbeta: groups on one BatchNorm add into the same element, in group order.  Lane i handles group i (the first group of
    // every distinct pointer writes): the two read-modify-writes of a lane have both loads in flight together -- thread 0 walking the
    // groups made two dependent memory round trips at the tail of every workgroup
    if ((int)threadIdx.x >= p.G) return;
    const int i = threadIdx.x;
    float* dgp = p.dgamma[i];
    float* dbp = p.dbeta[i];
    bool fg = dgp != nullptr, fb = dbp != nullptr;
    for (int j = 0; j < i; ++j) {
        fg = fg && p.dgamma[j] != dgp;
        fb = fb && p.dbeta[j] != dbp;
    }
    float a = fg ? dgp[c] : 0.f, b = fb ? dbp[c] : 0.f;
    for (int j = i; j < p.G; ++j) {
        if (p.dgamma[j] == dgp) a += s_s2[j];
        if (p.dbeta[j] == dbp) b += s_s1[j];
    }
    if (fg) dgp[c] = a;
    if (fb) dbp[c] = b;
}

// ------------------------------------------------------------------------------------ GroupNorm(1, C)
// nn.GroupNorm(1, planes) (code/networks/unet.py:20-21: ONE group = all channels of a sample): every image is its own statistics
// group g (the launch lists are built with gstart = 0, 1, ..., N), and what differs from BatchNorm is only which sums are pooled:
// mean / variance over (C, H, W) of the image.  Same descriptor, same outputs as the BatchNorm finalize kernels (per-(group, channel)
// scale / shift, resp. the P, Q, R of dz = P*g + Q*z + R), so every conv / weight-gradient kernel is unchanged.
// (nn.InstanceNorm2d = BatchNorm statistics per (image, channel) without affine or running statistics: the BatchNorm kernels with one
// group per image, gamma = 1, beta = 0 and null running-statistics pointers.)
__device__ __forceinline__ double block_sum_d(double v, double* s_tmp) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += s_tmp[i];      // fixed order
    return r;
}

__global__ __launch_bounds__(256) void gn_finalize_fwd_kernel(const rd_bn_fwd_t p) {
    __shared__ double s_tmp[4];
    const int g = blockIdx.x;
    const double cnt = (double)p.count[g];                  // pixels per channel of this image
    double S1 = 0.0, S2 = 0.0;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < RD_STAT_SLOTS; ++k) {
            s1 += p.stats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 0];
            s2 += p.stats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 1];
        }
        const double b = p.conv_bias ? (double)p.conv_bias[c] : 0.0;   // the sums are those of (conv result - bias)
        S1 += s1 + cnt * b;
        S2 += s2 + 2.0 * b * s1 + cnt * b * b;
    }
    S1 = block_sum_d(S1, s_tmp);
    S2 = block_sum_d(S2, s_tmp);
    const double n = cnt * (double)p.C;
    const double m = S1 / n;
    double vard = S2 / n - m * m;
    if (vard < 0.0) vard = 0.0;
    const float mean = (float)m, invstd = 1.0f / sqrtf((float)vard + p.eps);
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const float sc = p.gamma[g][c] * invstd;
        p.scale[g * p.C + c] = sc;
        p.shift[g * p.C + c] = p.beta[g][c] - mean * sc;
        p.mean[g * p.C + c] = mean;
        p.invstd[g * p.C + c] = invstd;
    }
}

// one workgroup walks the images in order, so that dgamma / dbeta (shared by all images) accumulate in a fixed order
__global__ __launch_bounds__(256) void gn_finalize_bwd_kernel(const rd_bn_bwd_t p) {
    __shared__ double s_tmp[4];
    constexpr int CPT = 4;                                  // channels per thread: C <= 1024
    float dg[CPT], db[CPT], dcb[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) dg[j] = db[j] = dcb[j] = 0.f;
    for (int g = 0; g < p.G; ++g) {
        const float mu = p.mean[g * p.C], is = p.invstd[g * p.C];          // the same for every channel of the image
        double b1[CPT], b2[CPT];
        double A = 0.0, B = 0.0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = threadIdx.x + j * blockDim.x;
            b1[j] = b2[j] = 0.0;
            if (c < p.C) {
                for (int k = 0; k < RD_STAT_SLOTS; ++k) {
                    b1[j] += p.bstats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 0];
                    b2[j] += p.bstats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 1];
                }
                const double gam = (double)p.gamma[g][c];
                b2[j] = (double)is * (b2[j] - (double)mu * b1[j]);     // sum over the pixels of g * zhat, channel c
                A += gam * b1[j];
                B += gam * b2[j];
            }
        }
        A = block_sum_d(A, s_tmp);
        B = block_sum_d(B, s_tmp);
        const double n = (double)p.count[g] * (double)p.C;
        const float Q = (float)(-(double)is * (double)is * B / n);
        const float R = (float)(-(double)is * A / n) - Q * mu;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = threadIdx.x + j * blockDim.x;
            if (c < p.C) {
                const float P = p.gamma[g][c] * is;
                p.P[g * p.C + c] = P;
                p.Q[g * p.C + c] = Q;
                p.R[g * p.C + c] = R;
                dg[j] += (float)b2[j];
                db[j] += (float)b1[j];
                if (p.dbias) {                                  // sum over the pixels of dz = P g + Q z + R
                    double sz = 0.0;
                    for (int k = 0; k < RD_STAT_SLOTS; ++k) sz += p.fstats[((size_t)(g * RD_STAT_SLOTS + k) * p.C + c) * 2 + 0];
                    if (p.conv_bias) sz += (double)p.count[g] * (double)p.conv_bias[c];
                    dcb[j] += (float)((double)P * b1[j] + (double)Q * sz + (double)R * (double)p.count[g]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * blockDim.x;
        if (c < p.C) {
            if (p.dgamma[0]) p.dgamma[0][c] += dg[j];
            if (p.dbeta[0]) p.dbeta[0][c] += db[j];
            if (p.dbias) p.dbias[c] += dcb[j];
        }
    }
}

template <typename T>
__device__ __forceinline__ void ldv(const T* p, float* f) {
    Slot<T>::unpack(*reinterpret_cast<const uint4*>(p), f);
}

// Shifted sums of one thread (pivot k = the first value it saw, count n) -> plain sums in fp64:
//   sum x = s1 + n k,   sum x^2 = s2 + 2 k s1 + n k^2      (|x - k| ~ sigma, so s1, s2 are accurate in fp32)
__device__ __forceinline__ void unshift(float s1, float s2, float k, int n, double& S1, double& S2) {
    const double kd = (double)k, nd = (double)n;
    S1 = (double)s1 + nd * kd;
    S2 = (double)s2 + 2.0 * kd * (double)s1 + nd * kd * kd;
}

// lanes with the same channel slot (lane % SL; SL a power of two <= 32) are summed inside the wave, one LDS atomic per wave
// and channel, then one global fp64 atomic per workgroup and channel into the workgroup's statistics slot
template <int S>
__device__ __forceinline__ void reduce_stats_d(double (&A1)[S], double (&A2)[S], double* s_red, int SL, int sl, double* stats_gc, int C) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < S; ++e) {
        for (int o = SL; o < 64; o <<= 1) {
            A1[e] += __shfl_xor(A1[e], o, 64);
            A2[e] += __shfl_xor(A2[e], o, 64);
        }
    }
    if (lane < SL) {
#pragma unroll
        for (int e = 0; e < S; ++e) {
            atomicAdd(&s_red[(sl * S + e) * 2 + 0], (double)A1[e]);
            atomicAdd(&s_red[(sl * S + e) * 2 + 1], (double)A2[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&stats_gc[i], s_red[i]);
}

// sum / sum of squares of the virtual upsampled tensor; grid (blocks, N)
template <typename T>
__global__ __launch_bounds__(256) void up_stats_kernel(const T* t, double* stats, T* y_out, int h, int w, int C, GroupMap gm, int nslots) {
    // One item = the 2 x 2 hi-res pixels between four lo-res pixels (i..i+1, j..j+1; i, j from -1: the image border
    // clamps): those four outputs read the SAME four lo-res vectors with row / column weights {0.25, 0.75}, so an item
    // costs 4 loads and 12 flops per channel for 4 outputs (a pixel-per-item loop needs 16 and 24).  Per output the
    // arithmetic is unchanged -- top = t00 + lx (t01 - t00), bot likewise, u = top + ly (bot - top) (ATen's order) -- so
    // the results are bit-identical to interpolating every pixel on its own.
    constexpr int S = Slot<T>::N;
    extern __shared__ double s_redd[];                     // [C][2]
    const int n = blockIdx.y, g = group_of(gm, n);
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_redd[i] = 0.0;
    __syncthreads();
    const int SL = C / S;
    const int H = 2 * h, W = 2 * w;
    const int bw = w + 1, items = (h + 1) * bw * SL;
    const int sl = threadIdx.x % SL;                       // constant per thread (256 % SL == 0, stride % SL == 0)
    float a1[S], a2[S], piv[S];
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < S; ++e) a1[e] = a2[e] = piv[e] = 0.f;
    const T* b = t + (size_t)n * h * w * C + sl * S;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < items; idx += gridDim.x * blockDim.x) {
        const int blk = idx / SL;
        const int bi = blk / bw, i = bi - 1, j = blk - bi * bw - 1;
        const int r0 = max(i, 0), r1 = min(i + 1, h - 1), c0 = max(j, 0), c1 = min(j + 1, w - 1);
        float t00[S], t01[S], t10[S], t11[S];
        ldv<T>(b + ((size_t)r0 * w + c0) * C, t00);
        ldv<T>(b + ((size_t)r0 * w + c1) * C, t01);
        ldv<T>(b + ((size_t)r1 * w + c0) * C, t10);
        ldv<T>(b + ((size_t)r1 * w + c1) * C, t11);
        // the weights of up2_coord: hi-res 2i+1 sits at lo-res i + 0.25, 2i+2 at i + 0.75; at a clamped border both
        // lo-res vectors are the same one and any weight returns it exactly
        float u[2][2][S];
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const float dt = t01[e] - t00[e], db = t11[e] - t10[e];
            const float tl = t00[e] + 0.25f * dt, tr = t00[e] + 0.75f * dt;
            const float bl = t10[e] + 0.25f * db, br = t10[e] + 0.75f * db;
            const float dl = bl - tl, dr = br - tr;
            u[0][0][e] = tl + 0.25f * dl; u[1][0][e] = tl + 0.75f * dl;
            u[0][1][e] = tr + 0.25f * dr; u[1][1][e] = tr + 0.75f * dr;
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int Y = 2 * i + 1 + dy;
            if ((unsigned)Y >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int X = 2 * j + 1 + dx;
                if ((unsigned)X >= (unsigned)W) continue;
                float* uu = u[dy][dx];
                if (y_out) {
                    const uint4 pk = Slot<T>::pack(uu);
                    *reinterpret_cast<uint4*>(y_out + ((size_t)n * H * W + (size_t)Y * W + X) * C + sl * S) = pk;
                    Slot<T>::unpack(pk, uu);
                }
                if (cnt == 0) {
#pragma unroll
                    for (int e = 0; e < S; ++e) piv[e] = uu[e];
                }
                ++cnt;
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    const float d = uu[e] - piv[e];
                    a1[e] += d;
                    a2[e] += d * d;
                }
            }
        }
    }
    double A1[S], A2[S];
#pragma unroll
    for (int e = 0; e < S; ++e) unshift(a1[e], a2[e], piv[e], cnt, A1[e], A2[e]);
    const int slot = (blockIdx.x + 7 * blockIdx.y) % nslots;
    reduce_stats_d<S>(A1, A2, s_redd, SL, sl, stats + ((size_t)g * RD_STAT_SLOTS + slot) * C * 2, C);
}

// sum / sum of squares of a plain NHWC tensor (standalone BatchNorm2d / DSBN forward); grid (blocks, N)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* x, double* stats, int HW, int C, GroupMap gm) {
    extern __shared__ double s_redd[];                     // [C][2]
    const int n = blockIdx.y, g = group_of(gm, n);
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_redd[i] = 0.0;
    __syncthreads();
    // generic channel counts (a standalone BN may have any C): one thread per (pixel, channel) element, channel fixed per
    // thread when blockDim % C == 0, else recomputed -- keep it simple: thread t owns channel t % C of pixels t / C + k*P
    const int tpb = blockDim.x / C * C;                    // threads that take part (multiple of C)
    const int c = threadIdx.x % C;
    float a1 = 0.f, a2 = 0.f, piv = 0.f;
    int cnt = 0;
    if ((int)threadIdx.x < tpb) {
        const int ppb = tpb / C;                           // pixels per block per pass
        for (int pix = blockIdx.x * ppb + threadIdx.x / C; pix < HW; pix += gridDim.x * ppb) {
            const float v = to_f<T>(x[((size_t)n * HW + pix) * C + c]);
            if (cnt == 0) piv = v;
            ++cnt;
            const float d = v - piv;
            a1 += d;
            a2 += d * d;
        }
    }
    double A1, A2;
    unshift(a1, a2, piv, cnt, A1, A2);
    atomicAdd(&s_redd[c * 2 + 0], A1);
    atomicAdd(&s_redd[c * 2 + 1], A2);
    __syncthreads();
    const int slot = (blockIdx.x + 7 * blockIdx.y) % RD_STAT_SLOTS;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x)
        atomicAdd(&stats[((size_t)g * RD_STAT_SLOTS + slot) * C * 2 + i], s_redd[i]);
}

// ------------------------------------------------------------------------------------ 2x2 max-pool, materialised
// p = maxpool2(act(z*scale+shift)) stored once (1/4 of the pixels), so that the conv behind nn.MaxPool2d(2) (unet.py:45,56)
// reads a plain tensor on the prefetching tile loaders in forward, dgrad and wgrad, instead of gathering 4 taps per element
// in each of them.  grid (blocks, N); one item = one pooled pixel x one 16-byte channel slot.
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* z, const float* scale, const float* shift, float slope, T* out,
                                                       int Ho, int Wo, int C, GroupMap gm, const rdfin::FinArg fa) {
    constexpr int S = Slot<T>::N;
    extern __shared__ float s_cf[];                        // [2][C]
    rdfin::prologue(fa);                                   // the producer's BatchNorm finalize folded into this launch (bn_fin.h)
    const int n = blockIdx.y, g = group_of(gm, n);
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        s_cf[i] = scale ? scale[g * C + i] : 1.f;
        s_cf[C + i] = shift ? shift[g * C + i] : 0.f;
    }
    __syncthreads();
    const int SL = C / S, W = 2 * Wo;
    const int items = Ho * Wo * SL;
    const T* zb = z + (size_t)n * 4 * Ho * Wo * C;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < items; idx += gridDim.x * blockDim.x) {
        const int sl = idx % SL, pix = idx / SL;
        const int y = pix / Wo, x = pix - y * Wo;
        float v[4][S], best[S];
#pragma unroll
        for (int k = 0; k < 4; ++k) ldv<T>(zb + ((size_t)(2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * C + sl * S, v[k]);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const float sc = s_cf[sl * S + e], sh = s_cf[C + sl * S + e];
            best[e] = act_fn(v[0][e] * sc + sh, slope);
#pragma unroll
            for (int k = 1; k < 4; ++k) best[e] = fmaxf(best[e], act_fn(v[k][e] * sc + sh, slope));
        }
        *reinterpret_cast<uint4*>(out + ((size_t)n * Ho * Wo + pix) * C + sl * S) = Slot<T>::pack(best);
    }
}

// backward: the gradient gp w.r.t. the pooled tensor goes to the FIRST maximum of each window (ATen max_pool2d), times the
// activation's derivative there; the other three positions get 0 (or keep what a skip connection wrote: accumulate); the
// BatchNorm-backward sums (sum g, sum g*z) of the scattered part are added to the producer's statistics slots.
// ACC (the old gradient is read and added to) is a template parameter: as a run-time flag, the four conditional loads made every wait
// of the loop a vmcnt(0) -- the z loads were waited for one by one
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* gp, const T* z, const float* scale, const float* shift, float slope,
                                                       int act, T* gout, double* bstats, int Ho, int Wo, int C,
                                                       GroupMap gm, int nslots) {
    constexpr int S = Slot<T>::N;
    extern __shared__ double s_redd[];                     // [C][2], then [2][C] floats of coefficients
    float* s_cf = reinterpret_cast<float*>(s_redd + 2 * C);
    const int n = blockIdx.y, g = group_of(gm, n);
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_redd[i] = 0.0;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        s_cf[i] = scale ? scale[g * C + i] : 1.f;
        s_cf[C + i] = shift ? shift[g * C + i] : 0.f;
    }
    __syncthreads();
    const int SL = C / S, W = 2 * Wo;
    const int items = Ho * Wo * SL;
    const int sl = threadIdx.x % SL;                       // constant per thread (256 % SL == 0, stride % SL == 0)
    const T* zb = z + (size_t)n * 4 * Ho * Wo * C + sl * S;
    T* gb = gout + (size_t)n * 4 * Ho * Wo * C + sl * S;
    float b1[S], b2[S], sc[S], sh[S];
#pragma unroll
    for (int e = 0; e < S; ++e) {
        b1[e] = b2[e] = 0.f;
        sc[e] = s_cf[sl * S + e];
        sh[e] = s_cf[C + sl * S + e];
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < items; idx += gridDim.x * blockDim.x) {
        const int pix = idx / SL;
        const int y = pix / Wo, x = pix - y * Wo;
        float zz[4][S], gw[4][S], da[S];
        ldv<T>(gp + ((size_t)n * Ho * Wo + pix) * C + sl * S, da);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = ((size_t)(2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * C;
            ldv<T>(zb + o, zz[k]);
            if constexpr (ACC) ldv<T>(gb + o, gw[k]);
        }
#pragma unroll
        for (int e = 0; e < S; ++e) {
            float best = act_fn(zz[0][e] * sc[e] + sh[e], slope);
            int arg = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float a = act_fn(zz[k][e] * sc[e] + sh[e], slope);
                if (a > best) { best = a; arg = k; }      // first max wins
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gn = 0.f;
                if (arg == k) {
                    gn = da[e] * (act ? act_grad(zz[k][e] * sc[e] + sh[e], slope) : 1.f);
                    b1[e] += gn;
                    b2[e] += gn * zz[k][e];
                }
                gw[k][e] = ACC ? gw[k][e] + gn : gn;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4*>(gb + ((size_t)(2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * C) = Slot<T>::pack(gw[k]);
    }
    if (!bstats) return;
    double A1[S], A2[S];
#pragma unroll
    for (int e = 0; e < S; ++e) { A1[e] = (double)b1[e]; A2[e] = (double)b2[e]; }
    const int slot = (blockIdx.x + 7 * blockIdx.y) % nslots;
    reduce_stats_d<S>(A1, A2, s_redd, SL, sl, bstats + ((size_t)g * RD_STAT_SLOTS + slot) * C * 2, C);
}

// weights of the (up to 4) hi-res rows 2y-1..2y+2 on lo-res row y, and the 3 coefficients of row y of U^T U
__device__ __forceinline__ void up_adjoint_1d(int y, int h, float* wy, float* A) {
    A[0] = A[1] = A[2] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int Y = 2 * y - 1 + j;
        wy[j] = 0.f;
        if (Y < 0 || Y >= 2 * h) continue;
        int i0, i1;
        float lam;
        up2_coord(Y, h, i0, i1, lam);
        const float wgt = (i0 == y ? 1.f - lam : 0.f) + (i1 == y ? lam : 0.f);
        wy[j] = wgt;
        if (wgt != 0.f) {
            A[i0 - y + 1] += wgt * (1.f - lam);
            A[i1 - y + 1] += wgt * lam;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void up_bwd_kernel(const T* g2, const T* t, T* dt, const float* P, const float* Q,
                                                     const float* R, int h, int w, int C, GroupMap gm, const rdfin::FinArg fa) {
    constexpr int S = Slot<T>::N;
    rdfin::prologue(fa);                                   // the BatchNorm-backward finalize folded into this launch (bn_fin.h)
    const int n = blockIdx.y, gi = group_of(gm, n);
    const int SL = C / S;
    const int items = h * w * SL;
    const int H = 2 * h, W = 2 * w;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < items; idx += gridDim.x * blockDim.x) {
        const int sl = idx % SL, pix = idx / SL;
        const int y = pix / w, x = pix - y * w;
        float wy[4], wx[4], Ay[3], Ax[3];
        up_adjoint_1d(y, h, wy, Ay);
        up_adjoint_1d(x, w, wx, Ax);
        float G[S], TU[S];
#pragma unroll
        for (int e = 0; e < S; ++e) G[e] = TU[e] = 0.f;
        const T* gb = g2 + (size_t)n * H * W * C + sl * S;
        const T* tb = t + (size_t)n * h * w * C + sl * S;
        // branch-free: out-of-image taps have weight exactly 0 (up_adjoint_1d) and read a clamped, valid address, so the
        // 16 + 9 loads of an item are issued back to back instead of one per conditional block
        uint4 gq[4][4], tq[3][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int Yc = min(max(2 * y - 1 + j, 0), H - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int Xc = min(max(2 * x - 1 + k, 0), W - 1);
                gq[j][k] = *reinterpret_cast<const uint4*>(gb + ((size_t)Yc * W + Xc) * C);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int yc = min(max(y + a - 1, 0), h - 1);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int xc = min(max(x + b - 1, 0), w - 1);
                tq[a][b] = *reinterpret_cast<const uint4*>(tb + ((size_t)yc * w + xc) * C);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wgt = wy[j] * wx[k];
                float v[S];
                Slot<T>::unpack(gq[j][k], v);
#pragma unroll
                for (int e = 0; e < S; ++e) G[e] += wgt * v[e];
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float cf = Ay[a] * Ax[b];
                float v[S];
                Slot<T>::unpack(tq[a][b], v);
#pragma unroll
                for (int e = 0; e < S; ++e) TU[e] += cf * v[e];
            }
        const float Wsum = (wy[0] + wy[1] + wy[2] + wy[3]) * (wx[0] + wx[1] + wx[2] + wx[3]);
        float o[S];
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const int c = gi * C + sl * S + e;
            o[e] = P[c] * G[e] + Q[c] * TU[e] + R[c] * Wsum;
        }
        *reinterpret_cast<uint4*>(dt + ((size_t)(n * h + y) * w + x) * C + sl * S) = Slot<T>::pack(o);
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int N, int C, int H, int W, int Cs) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const size_t pix = i / C;
        const int xw = pix % W, yh = (pix / W) % H, n = pix / ((size_t)W * H);
        y[pix * Cs + c] = from_f<T>(x[(((size_t)n * C + c) * H + yh) * W + xw]);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* z, float* y, const float* scale, const float* shift, int act, float slope,
                                    int N, int C, int H, int W, GroupMap gm) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xw = i % W, yh = (i / W) % H, c = (i / ((size_t)W * H)) % C, n = i / ((size_t)W * H * C);
        float v = to_f<T>(z[(((size_t)n * H + yh) * W + xw) * C + c]);
        if (scale) {
            const int g = group_of(gm, n);
            v = v * scale[g * C + c] + shift[g * C + c];
            if (act) v = act_fn(v, slope);
        }
        y[i] = v;
    }
}

// one block = 64 consecutive pixels of one image; per-channel partial sums through LDS
template <typename T>
__global__ __launch_bounds__(256) void grad_in_kernel(const float* dy, const T* z, T* gout, const float* scale,
                                                      const float* shift, double* bstats, int act, float slope,
                                                      int accumulate, int C, int H, int W, GroupMap gm) {
    extern __shared__ double s_red[];                      // [C][2]
    const int n = blockIdx.y, g = group_of(gm, n);
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_red[i] = 0.0;
    __syncthreads();
    const int HW = H * W;
    const int p0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * C; i += blockDim.x) {
        const int c = i / 64, pp = p0 + (i % 64);          // consecutive lanes -> consecutive pixels of a NCHW plane
        float gn = 0.f, zz = 0.f;
        if (pp < HW) {
            const size_t zi = ((size_t)n * HW + pp) * C + c;
            zz = z ? to_f<T>(z[zi]) : 0.f;
            float m = 1.f;
            if (act && scale) m = act_grad(zz * scale[g * C + c] + shift[g * C + c], slope);
            gn = dy[((size_t)n * C + c) * HW + pp] * m;
            gout[zi] = from_f<T>(accumulate ? to_f<T>(gout[zi]) + gn : gn);
        }
        float a = wave_sum(gn), b = wave_sum(gn * zz);     // 64 lanes share channel c
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_red[c * 2 + 0], (double)a);
            atomicAdd(&s_red[c * 2 + 1], (double)b);
        }
    }
    __syncthreads();
    if (bstats) {
        const int slot = (blockIdx.x + 7 * blockIdx.y) % RD_STAT_SLOTS;
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&bstats[((size_t)g * RD_STAT_SLOTS + slot) * C * 2 + i], s_red[i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* partial, int64_t npix, int C, int Cs) {
    // C <= 8; partial[block][C]
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) acc[c] += to_f<T>(x[i * Cs + c]);
    __shared__ float s[4][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < C) partial[blockIdx.x * 8 + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

// one wave per channel; lanes stride over the per-block partials
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* partial, float* out, int nblocks, int C, float beta) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b * 8 + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
}

// out = act(a*x + b*x2 + c; slope) over NHWC, one 16-byte channel slot per thread-item
template <typename T, bool TWO>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, const T* x2, T* out, const float* a, const float* b, const float* c,
                                                       float slope, int HW, int C, GroupMap gm) {
    constexpr int S = Slot<T>::N;
    const int n = blockIdx.y, g = group_of(gm, n);
    const int SL = C / S;
    const size_t base = (size_t)n * HW * C;
    const int items = HW * SL;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < items; idx += gridDim.x * blockDim.x) {
        const int sl = idx % SL;
        float ca[S], cb[S], cc[S];
#pragma unroll
        for (int e = 0; e < S; ++e) {
            ca[e] = a[g * C + sl * S + e];
            cc[e] = c[g * C + sl * S + e];
            cb[e] = TWO ? b[g * C + sl * S + e] : 0.f;
        }
        float v[S], w[S];
        const size_t off = base + (size_t)idx * S;
        Slot<T>::unpack(*reinterpret_cast<const uint4*>(x + off), v);
        if constexpr (TWO) {
            Slot<T>::unpack(*reinterpret_cast<const uint4*>(x2 + off), w);
#pragma unroll
            for (int e = 0; e < S; ++e) v[e] = act_fn(ca[e] * v[e] + cb[e] * w[e] + cc[e], slope);
        } else {
#pragma unroll
            for (int e = 0; e < S; ++e) v[e] = act_fn(ca[e] * v[e] + cc[e], slope);
        }
        *reinterpret_cast<uint4*>(out + off) = Slot<T>::pack(v);
    }
}

GroupMap host_gm(int G, const int32_t* gs) {
    GroupMap gm;
    gm.G = G;
    for (int i = 0; i <= RD_MAX_GROUPS; ++i) gm.gs[i] = (gs && i <= G) ? gs[i] : 0;
    return gm;
}

int grid_for(size_t total, int per_block = 256, int cap = 4096) {
    size_t b = (total + per_block - 1) / per_block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int rd_bn_finalize_fwd(const rd_bn_fwd_t* p, void* stream) {
    if (!p || p->G < 1 || p->G > RD_MAX_GROUPS) return -1;
    rd_launch(bn_finalize_fwd_kernel, dim3(p->C), dim3(64 * p->G), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

int rd_bn_finalize_bwd(const rd_bn_bwd_t* p, void* stream) {
    if (!p || p->G < 1 || p->G > RD_MAX_GROUPS) return -1;
    rd_launch(bn_finalize_bwd_kernel, dim3(p->C), dim3(64 * p->G), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

int rd_gn_finalize_fwd(const rd_bn_fwd_t* p, void* stream) {
    if (!p || p->G < 1 || p->G > RD_MAX_GROUPS || p->C < 1) return -1;
    rd_launch(gn_finalize_fwd_kernel, dim3(p->G), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

int rd_gn_finalize_bwd(const rd_bn_bwd_t* p, void* stream) {
    if (!p || p->G < 1 || p->G > RD_MAX_GROUPS || p->C < 1 || p->C > 1024) return -1;
    rd_launch(gn_finalize_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

int rd_up_stats(const void* t, double* stats, void* y_out, int N, int h, int w, int C, int G, const int32_t* gstart_host, int dtype,
                int stat_slots, void* stream) {
    const int S = dtype == RD_BF16 ? 8 : 4;
    if (C % S || 256 % (C / S) || stat_slots < 0 || stat_slots > RD_STAT_SLOTS) return -2;
    const int nslots = stat_slots > 0 ? stat_slots : RD_STAT_SLOTS;
    const GroupMap gm = host_gm(G, gstart_host);
    const int items = (h + 1) * (w + 1) * (C / S);                     // 2 x 2 output pixels per item
    static const int us_per = rd_switch("RD_UPS_PER", 2);
    int bx = grid_for(items, 256 * us_per, 4096);
    dim3 grid(bx, N);
    if (dtype == RD_BF16)
        rd_launch(up_stats_kernel<bf16_t>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream,
                           (const bf16_t*)t, stats, (bf16_t*)y_out, h, w, C, gm, nslots);
    else
        rd_launch(up_stats_kernel<float>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream,
                           (const float*)t, stats, (float*)y_out, h, w, C, gm, nslots);
    return (int)hipGetLastError();
}

int rd_bn_stats(const void* x, double* stats, int N, int H, int W, int C, int G, const int32_t* gstart_host, int dtype, void* stream) {
    if (!x || !stats || C < 1 || C > 256) return -2;
    if (G < 1 || G > RD_MAX_GROUPS) return -1;
    const GroupMap gm = host_gm(G, gstart_host);
    const int ppb = 256 / C;                                   // pixels per block per pass
    const int bx = grid_for(H * W, ppb * 16, 2048);
    dim3 grid(bx, N);
    if (dtype == RD_BF16)
        rd_launch(bn_stats_kernel<bf16_t>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream, (const bf16_t*)x, stats, H * W, C, gm);
    else
        rd_launch(bn_stats_kernel<float>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream, (const float*)x, stats, H * W, C, gm);
    return (int)hipGetLastError();
}

int rd_pool_fwd(const void* z, const float* scale, const float* shift, float slope, void* out, int N, int Ho, int Wo, int C, int G,
                const int32_t* gstart_host, int dtype, const rd_bn_fwd_t* fin, int fin_flags, void* stream) {
    const int S = dtype == RD_BF16 ? 8 : 4;
    if (!z || !out || C % S || C > 1024) return -2;
    if (G < 1 || G > RD_MAX_GROUPS) return -1;
    rdfin::FinArg fa;
    rdfin::no_arg(fa);
    if (fin) {
        rd_src_t s;
        memset(&s, 0, sizeof(s));
        s.fin = fin;
        s.fin_flags = fin_flags;
        s.mode = RD_SRC_AFFACT;
        s.C = C;
        if (!scale || !shift || fin->scale != scale || fin->shift != shift || rdfin::make_arg(fa, &s, 1)) return -3;
    }
    const GroupMap gm = host_gm(G, gstart_host);
    dim3 grid(grid_for((size_t)Ho * Wo * (C / S), 256 * 4, 2048), N);
    if (dtype == RD_BF16)
        rd_launch(pool_fwd_kernel<bf16_t>, grid, dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, (const bf16_t*)z, scale, shift,
                           slope, (bf16_t*)out, Ho, Wo, C, gm, fa);
    else
        rd_launch(pool_fwd_kernel<float>, grid, dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, (const float*)z, scale, shift,
                           slope, (float*)out, Ho, Wo, C, gm, fa);
    return (int)hipGetLastError();
}

int rd_pool_bwd(const void* gp, const void* z, const float* scale, const float* shift, float slope, int act, void* g, int accumulate,
                double* bstats, int N, int Ho, int Wo, int C, int G, const int32_t* gstart_host, int dtype, int stat_slots, void* stream) {
    const int S = dtype == RD_BF16 ? 8 : 4;
    if (!gp || !z || !g || C % S || C > 1024 || 256 % (C / S) || stat_slots < 0 || stat_slots > RD_STAT_SLOTS) return -2;
    const int nslots = stat_slots > 0 ? stat_slots : RD_STAT_SLOTS;
    if (G < 1 || G > RD_MAX_GROUPS) return -1;
    const GroupMap gm = host_gm(G, gstart_host);
    dim3 grid(grid_for((size_t)Ho * Wo * (C / S), 256 * 4, 2048), N);
    const size_t lds = 2 * C * sizeof(double) + 2 * C * sizeof(float);
    if (dtype == RD_BF16) {
        if (accumulate)
            rd_launch((pool_bwd_kernel<bf16_t, true>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)gp, (const bf16_t*)z, scale,
                               shift, slope, act, (bf16_t*)g, bstats, Ho, Wo, C, gm, nslots);
        else
            rd_launch((pool_bwd_kernel<bf16_t, false>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)gp, (const bf16_t*)z, scale,
                               shift, slope, act, (bf16_t*)g, bstats, Ho, Wo, C, gm, nslots);
    } else {
        if (accumulate)
            rd_launch((pool_bwd_kernel<float, true>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)gp, (const float*)z, scale, shift,
                               slope, act, (float*)g, bstats, Ho, Wo, C, gm, nslots);
        else
            rd_launch((pool_bwd_kernel<float, false>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)gp, (const float*)z, scale, shift,
                               slope, act, (float*)g, bstats, Ho, Wo, C, gm, nslots);
    }
    return (int)hipGetLastError();
}

int rd_bn_apply(const void* x, const void* x2, void* out, const float* a, const float* b, const float* c, float slope, int N, int H,
                int W, int C, int G, const int32_t* gstart_host, int dtype, void* stream) {
    const int S = dtype == RD_BF16 ? 8 : 4;
    if (C % S || !x || !out || !a || !c || ((x2 != nullptr) != (b != nullptr))) return -2;
    if (G < 1 || G > RD_MAX_GROUPS) return -1;
    const GroupMap gm = host_gm(G, gstart_host);
    const int items = H * W * (C / S);
    dim3 grid(grid_for(items, 256 * 4, 1024), N);
    hipStream_t st = (hipStream_t)stream;
#define RD_APPLY(T_, TWO_) rd_launch((bn_apply_kernel<T_, TWO_>), grid, dim3(256), 0, st, (const T_*)x, (const T_*)x2, (T_*)out, \
                                              a, b, c, slope, H * W, C, gm)
    if (dtype == RD_BF16) { if (x2) RD_APPLY(bf16_t, true); else RD_APPLY(bf16_t, false); }
    else { if (x2) RD_APPLY(float, true); else RD_APPLY(float, false); }
#undef RD_APPLY
    return (int)hipGetLastError();
}

int rd_up_bwd(const void* g, const void* t, void* dt, const float* P, const float* Q, const float* R, int N, int h, int w,
              int C, int G, const int32_t* gstart_host, int dtype, const rd_bn_bwd_t* fin, int fin_flags, void* stream) {
    const int S = dtype == RD_BF16 ? 8 : 4;
    if (C % S) return -2;
    rdfin::FinArg fa;
    rdfin::no_arg(fa);
    if (fin) {
        rd_src_t s;
        memset(&s, 0, sizeof(s));
        s.fin = fin;
        s.fin_flags = fin_flags;
        s.mode = RD_SRC_BNBWD;
        s.C = C;
        if (fin->P != P || fin->Q != Q || fin->R != R || rdfin::make_arg(fa, &s, 1)) return -3;
    }
    const GroupMap gm = host_gm(G, gstart_host);
    dim3 grid(grid_for((size_t)h * w * (C / S), 256, 1024), N);
    if (dtype == RD_BF16)
        rd_launch(up_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)t,
                           (bf16_t*)dt, P, Q, R, h, w, C, gm, fa);
    else
        rd_launch(up_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)t,
                           (float*)dt, P, Q, R, h, w, C, gm, fa);
    return (int)hipGetLastError();
}

int rd_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int cstride, int dtype, void* stream) {
    const int Cs = cstride > 0 ? cstride : C;
    if (Cs < C) return -1;
    const size_t total = (size_t)N * C * H * W;
    if (dtype == RD_BF16)
        rd_launch(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, N, C, H, W, Cs);
    else
        rd_launch(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (float*)y, N, C, H, W, Cs);
    return (int)hipGetLastError();
}

int rd_nhwc_to_nchw(const void* z, float* y, const float* scale, const float* shift, int act, float slope, int N, int C,
                    int H, int W, int G, const int32_t* gstart_host, int dtype, void* stream) {
    const size_t total = (size_t)N * C * H * W;
    const GroupMap gm = host_gm(G, gstart_host);
    if (dtype == RD_BF16)
        rd_launch(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, y,
                           scale, shift, act, slope, N, C, H, W, gm);
    else
        rd_launch(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)z, y,
                           scale, shift, act, slope, N, C, H, W, gm);
    return (int)hipGetLastError();
}

int rd_grad_in(const float* dy, const void* z, void* g, const float* scale, const float* shift, double* bstats, int act,
               float slope, int accumulate, int N, int C, int H, int W, int G, const int32_t* gstart_host, int dtype,
               void* stream) {
    const GroupMap gm = host_gm(G, gstart_host);
    dim3 grid((H * W + 63) / 64, N);
    if (dtype == RD_BF16)
        rd_launch(grad_in_kernel<bf16_t>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream, dy, (const bf16_t*)z,
                           (bf16_t*)g, scale, shift, bstats, act, slope, accumulate, C, H, W, gm);
    else
        rd_launch(grad_in_kernel<float>, grid, dim3(256), 2 * C * sizeof(double), (hipStream_t)stream, dy, (const float*)z,
                           (float*)g, scale, shift, bstats, act, slope, accumulate, C, H, W, gm);
    return (int)hipGetLastError();
}

int rd_colsum(const void* x, float* out, float* partial_ws, int64_t npix, int C, int cstride, float beta, int dtype, void* stream) {
    const int Cs = cstride > 0 ? cstride : C;
    if (C > 8) return -2;
    const int nb = grid_for((size_t)npix, 256 * 4, 1024);
    if (dtype == RD_BF16)
        rd_launch(colsum_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, partial_ws, npix, C, Cs);
    else
        rd_launch(colsum_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, partial_ws, npix, C, Cs);
    rd_launch(colsum_final_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial_ws, out, nb, C, beta);
    return (int)hipGetLastError();
}

}  // extern "C"
