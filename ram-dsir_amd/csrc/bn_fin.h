// bn_fin.h -- BatchNorm finalize FOLDED into the consumer (round 5): the launch that first reads a layer's BatchNorm coefficients
// derives them from the producer's statistic slots in its own prologue, instead of a one-workgroup-per-channel finalize launch
// between every two dependent convs (52 launches of ~7 us on the critical lane of the training step, 11 % of it).
//
// Contract (include/ramdsir.h, rd_src_t.fin): `fin` points to the rd_bn_fwd_t (forward sources: scale / shift) or rd_bn_bwd_t
// (RD_SRC_BNBWD sources: P / Q / R) that the explicit finalize launch would have been given (a HOST struct, copied into the kernel
// arguments by the entry point: FinArg below).  EVERY workgroup of the
// launch computes ALL coefficients of the layer and writes them to the descriptor's output vectors -- identical values from every
// workgroup, so the concurrent writes are benign -- then a workgroup barrier (workgroup-scope release / acquire: the waves of a
// workgroup share the CU's vector L1, and every later read of the vectors, by this workgroup or any other, sees L2 lines that already
// hold the final values), and the kernel body reads the vectors exactly as it did when a finalize launch had written them.  Consumers
// ordered behind this launch (weight gradients, the gradient epilogues, skip connections) read the vectors without a `fin`.
// The workgroup with the highest block index of an RD_FIN_OWNER launch also does what must happen exactly once: saved mean / invstd,
// running statistics (momentum recursion in group order for groups that share one BatchNorm), num_batches_tracked; dgamma / dbeta.
//
// Numerics: the formulas and fp32 / fp64 roles are those of bn_finalize_fwd_kernel / bn_finalize_bwd_kernel (bn.hip) verbatim; the
// slot sums are fp64 sums of fp32 terms -- exact, hence order-independent, WHILE the terms of one sum span fewer than 2^29 in
// magnitude x count (conv_device.h flush_bstats; DESIGN.md "run-to-run reproducibility": true of every tensor of the network, not of
// adversarial data) -- so a folded finalize is bit-identical to the explicit launch.  With 8 copies instead of 64 (RD_STAT_SLOTS_FOLD)
// each copy takes 8 x as many atomics; the persistent kernels add once per workgroup, which is what keeps that cheap.
#pragma once
#include <string.h>
#include "common.h"
#include "../../include/ramdsir.h"

namespace rdfin {

__device__ __forceinline__ bool last_block() {
    return blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1;
}

// sum of the (sum, sum2) pairs of group g, channel c over the first `ns` slots (ns a multiple of 8: 8 loads in flight at a time)
__device__ __forceinline__ void slot_sums(const double* __restrict__ st, int g, int c, int C, int ns, double& s1, double& s2) {
    typedef __attribute__((ext_vector_type(2))) double d2;
    const d2* b = reinterpret_cast<const d2*>(st) + (size_t)g * RD_STAT_SLOTS * C + c;
    s1 = 0.0;
    s2 = 0.0;
    for (int k0 = 0; k0 < ns; k0 += 8) {
        d2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b[(size_t)(k0 + k) * C];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1 += v[k].x; s2 += v[k].y; }
    }
}

// The arithmetic below is shared with the explicit finalize kernels (bn.hip) and compiled with floating-point contraction OFF: whether
// a * b + c becomes one fma or two roundings otherwise depends on the code around the expression, i.e. on the kernel it was inlined into,
// and one ulp in a BatchNorm coefficient is amplified to 1e-3 in the gradients by the ReLU decisions behind it (numerics).
struct FwdStat { float mean, invstd, unb; };

__device__ __forceinline__ FwdStat fwd_stat(double s1, double s2, float count, float cbias, float eps) {
#pragma clang fp contract(off)
    const double cnt = (double)count;
    const double m0 = s1 / cnt;                                // mean of the bias-free result
    double vard = s2 / cnt - m0 * m0;
    if (vard < 0.0) vard = 0.0;
    const float var = (float)vard;
    FwdStat r;
    r.mean = (float)(m0 + (double)cbias);
    r.invstd = 1.0f / sqrtf(var + eps);
    r.unb = cnt > 1.0 ? (float)(vard * cnt / (cnt - 1.0)) : var;
    return r;
}

__device__ __forceinline__ void fwd_coef(float gam, float bet, const FwdStat& r, float& sc, float& sh) {
#pragma clang fp contract(off)
    sc = gam * r.invstd;
    sh = bet - r.mean * sc;
}

// one step of the running-statistics recursion: (1 - momentum) * old + momentum * batch value
__device__ __forceinline__ float momentum_step(float old, float x, float mo) {
#pragma clang fp contract(off)
    return (1.f - mo) * old + mo * x;
}

// ---- the descriptor of a folded finalize travels BY VALUE in the kernel arguments (the host entry points copy it from rd_src_t.fin):
// its fields arrive with the kernel's other arguments instead of costing a dependent memory round trip of their own
struct FinArg {
    union { rd_bn_fwd_t f; rd_bn_bwd_t b; } d;
    int32_t which;        // -1: none; conv launches: index of the source that carries it; weight gradients: 0 (dz)
    int32_t flags;        // RD_FIN_*
    int32_t is_bwd;       // d.b is valid (RD_SRC_BNBWD source), else d.f
    int32_t pad_;
};
constexpr int FIN_MAX_G = 8;                                   // groups of a folded finalize (static selects below); the host checks

// (group, channel) of pair i without an integer division
__device__ __forceinline__ void pair_of(int i, int C, int& g, int& c) {
    g = 0;
    c = i;
    while (c >= C) { c -= C; ++g; }
}

// forward: scale = gamma * invstd, shift = beta - mean * scale for every (group, channel); owner: mean / invstd / running statistics
// lds: NULL, or a table of 2 * G * C floats in LDS that receives the coefficients as well: [0][g][c] scale, [1][g][c] shift
// xp: timing experiments of the debug build (scripts/fin_ubench.py; results are wrong when set)
__device__ __forceinline__ void fwd(const FinArg& fa, float* lds = nullptr, bool all_write = true) {
    const rd_bn_fwd_t& f = fa.d.f;
    const int C = f.C, G = f.G, ns = f.nslots > 0 ? f.nslots : RD_STAT_SLOTS;
    const double* st = f.stats;
    const bool own = (fa.flags & RD_FIN_OWNER) != 0 && last_block();
    // a flat thread index: with a 2-D / 3-D block threads that share threadIdx.x would otherwise race on the owner's read-modify-writes
    const int nthr = blockDim.x * blockDim.y * blockDim.z;
    const int flat = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    for (int i = flat; i < G * C; i += nthr) {
        int g, c;
        pair_of(i, C, g, c);
        double s1, s2;
        slot_sums(st, g, c, C, ns, s1, s2);
        // per-group fields by static index + select: the descriptor lives in the kernel-argument segment, a statically indexed field
        // is a scalar load issued with the kernel's other arguments
        const float* gp = f.gamma[0];
        const float* bp = f.beta[0];
        float cnt = f.count[0];
#pragma unroll
        for (int j = 1; j < FIN_MAX_G; ++j)
            if (g == j) { gp = f.gamma[j]; bp = f.beta[j]; cnt = f.count[j]; }
        const float cbias = f.conv_bias ? f.conv_bias[c] : 0.f;
        const float gam = gp[c], bet = bp[c];
        // the owner's head thread of a chain of groups on one BatchNorm (the two passes of the seg network) also needs the NEXT group's
        // sums: requested here, beside its own, so that the common case (chains of one or two) costs one memory round trip
        bool head = false;
        int nxt = -1;
        float rm = 0.f, rv = 1.f;
        double t1 = 0.0, t2 = 0.0;
        if (own && f.running_mean[g]) {
            head = true;
            for (int j = 0; j < g; ++j) head = head && f.running_mean[j] != f.running_mean[g];
            if (head) {
                for (int j = G - 1; j > g; --j)
                    if (f.running_mean[j] == f.running_mean[g]) nxt = j;
                rm = f.running_mean[g][c];
                rv = f.running_var[g][c];
                if (nxt >= 0) slot_sums(st, nxt, c, C, ns, t1, t2);
            }
        }
        const FwdStat r = fwd_stat(s1, s2, cnt, cbias, f.eps);
        float sc, sh;
        fwd_coef(gam, bet, r, sc, sh);
        if (lds) {
            lds[i] = sc;
            lds[G * C + i] = sh;
        }
        if (all_write || own) {                                // (LDS delivery: one workgroup of the owner launch writes for later launches)
            f.scale[i] = sc;
            f.shift[i] = sh;
        }
        if (own) {
            f.mean[i] = r.mean;
            f.invstd[i] = r.invstd;
        }
        if (head) {
            const float mo = f.momentum;
            float nm = momentum_step(rm, r.mean, mo);
            float nv = momentum_step(rv, r.unb, mo);
            int j = nxt;
            while (j >= 0) {                                   // later groups on the same BatchNorm continue from here, in group order
                const FwdStat rj = fwd_stat(t1, t2, f.count[j], cbias, f.eps);
                nm = momentum_step(nm, rj.mean, mo);
                nv = momentum_step(nv, rj.unb, mo);
                int k = -1;
                for (int q = G - 1; q > j; --q)
                    if (f.running_mean[q] == f.running_mean[g]) k = q;
                j = k;
                if (j >= 0) slot_sums(st, j, c, C, ns, t1, t2);
            }
            f.running_mean[g][c] = nm;
            f.running_var[g][c] = nv;
        }
        if (own && i == 0) {
            for (int a = 0; a < G; ++a) {
                long long* q = reinterpret_cast<long long*>(f.num_batches_tracked[a]);
                if (!q) continue;
                bool first = true;
                for (int j = 0; j < a; ++j) first = first && reinterpret_cast<long long*>(f.num_batches_tracked[j]) != q;
                if (!first) continue;
                int n = 0;
                for (int j = a; j < G; ++j) n += reinterpret_cast<long long*>(f.num_batches_tracked[j]) == q ? 1 : 0;
                *q += n;
            }
        }
    }
}

struct BwdStat { float s1, s2; };

__device__ __forceinline__ BwdStat bwd_stat(double s1d, double sgzd, float mu, float is) {
#pragma clang fp contract(off)
    BwdStat r;
    r.s1 = (float)s1d;
    r.s2 = is * (float)(sgzd - (double)mu * s1d);              // sum g * zhat (the subtraction cancels when |mean| >> sigma: fp64)
    return r;
}

__device__ __forceinline__ void bwd_coef(float gam, float is, float mu, const BwdStat& r, float cnt, float& P, float& Q, float& R) {
#pragma clang fp contract(off)
    P = gam * is;
    Q = -gam * is * is * r.s2 / cnt;
    R = -P * r.s1 / cnt - Q * mu;
}

// P, Q, R of ONE (group, channel) pair: for kernels that need only their own channels and keep them in a table of their own (the
// weight gradients: a workgroup covers 64 of the layer's channels); no stores, no ownership
__device__ __forceinline__ void bwd_pair(const FinArg& fa, int g, int c, float& P, float& Q, float& R) {
    const rd_bn_bwd_t& q = fa.d.b;
    const int C = q.C, ns = q.nslots > 0 ? q.nslots : RD_STAT_SLOTS;
    double s1d, sgzd;
    slot_sums(q.bstats, g, c, C, ns, s1d, sgzd);
    const float* gp = q.gamma[0];
    float cnt = q.count[0];
#pragma unroll
    for (int j = 1; j < FIN_MAX_G; ++j)
        if (g == j) { gp = q.gamma[j]; cnt = q.count[j]; }
    const float mu = q.mean[g * C + c], is = q.invstd[g * C + c], gam = gp[c];
    const BwdStat r = bwd_stat(s1d, sgzd, mu, is);
    bwd_coef(gam, is, mu, r, cnt, P, Q, R);
}

// backward: dz = P g + Q z + R coefficients for every (group, channel); owner: dgamma += sum g zhat, dbeta += sum g, in group order
// lds: NULL, or 3 * G * C floats: [0] P, [1] R, [2] Q -- the order of rd_src_t's scale, shift, q for an RD_SRC_BNBWD source
__device__ __forceinline__ void bwd(const FinArg& fa, float* lds = nullptr, bool all_write = true) {
    const rd_bn_bwd_t& q = fa.d.b;
    const int C = q.C, G = q.G, ns = q.nslots > 0 ? q.nslots : RD_STAT_SLOTS;
    const double* st = q.bstats;
    const bool own = (fa.flags & RD_FIN_OWNER) != 0 && last_block();
    // a flat thread index: with a 2-D / 3-D block threads that share threadIdx.x would otherwise race on the owner's read-modify-writes
    const int nthr = blockDim.x * blockDim.y * blockDim.z;
    const int flat = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    for (int i = flat; i < G * C; i += nthr) {
        int g, c;
        pair_of(i, C, g, c);
        double s1d, sgzd;
        slot_sums(st, g, c, C, ns, s1d, sgzd);
        const float* gp = q.gamma[0];
        float cnt = q.count[0];
#pragma unroll
        for (int j = 1; j < FIN_MAX_G; ++j)
            if (g == j) { gp = q.gamma[j]; cnt = q.count[j]; }
        const float mu = q.mean[i], is = q.invstd[i], gam = gp[c];
        float* dgp = own ? q.dgamma[g] : nullptr;
        float* dbp = own ? q.dbeta[g] : nullptr;
        bool hg = dgp != nullptr, hb = dbp != nullptr;
        for (int j = 0; j < g; ++j) {
            hg = hg && q.dgamma[j] != dgp;
            hb = hb && q.dbeta[j] != dbp;
        }
        float a = hg ? dgp[c] : 0.f, b = hb ? dbp[c] : 0.f;
        int nxt = -1;
        double u1 = 0.0, u2 = 0.0;
        float mu2 = 0.f, is2 = 1.f;
        if (hg || hb) {
            for (int j = G - 1; j > g; --j)
                if ((hg && q.dgamma[j] == dgp) || (hb && q.dbeta[j] == dbp)) nxt = j;
            if (nxt >= 0) {
                slot_sums(st, nxt, c, C, ns, u1, u2);
                mu2 = q.mean[nxt * C + c];
                is2 = q.invstd[nxt * C + c];
            }
        }
        const BwdStat r = bwd_stat(s1d, sgzd, mu, is);
        float P, Q, R;
        bwd_coef(gam, is, mu, r, cnt, P, Q, R);
        if (lds) {
            lds[i] = P;
            lds[G * C + i] = R;
            lds[2 * G * C + i] = Q;
        }
        if (all_write || own) {
            q.P[i] = P;
            q.Q[i] = Q;
            q.R[i] = R;
        }
        if (hg || hb) {
            a += r.s2;
            b += r.s1;
            int j = nxt;
            while (j >= 0) {
                const BwdStat rj = bwd_stat(u1, u2, mu2, is2);
                if (hg && q.dgamma[j] == dgp) a += rj.s2;
                if (hb && q.dbeta[j] == dbp) b += rj.s1;
                int k = -1;
                for (int t = G - 1; t > j; --t)
                    if ((hg && q.dgamma[t] == dgp) || (hb && q.dbeta[t] == dbp)) k = t;
                j = k;
                if (j >= 0) {
                    slot_sums(st, j, c, C, ns, u1, u2);
                    mu2 = q.mean[j * C + c];
                    is2 = q.invstd[j * C + c];
                }
            }
            if (hg) dgp[c] = a;
            if (hb) dbp[c] = b;
        }
    }
}

// barrier that orders LDS traffic only: nobody waits for the global stores of the coefficients (their readers are later launches)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// floats of LDS a kernel sets aside for the coefficient table of a folded finalize (3 vectors of G * C <= 1024 entries)
constexpr int FIN_LDS_FLOATS = 3 * 1024;

__device__ __forceinline__ void run(const FinArg& fa, float* lds) {
    if (fa.is_bwd) bwd(fa, lds, lds == nullptr);
    else fwd(fa, lds, lds == nullptr);
}

// Prologue, global path: every workgroup writes the coefficient vectors, waits for its stores (full barrier) and the kernel body reads
// them back from memory, unchanged.  For the kernels that are not on the critical lane of the step.
__device__ __forceinline__ void prologue(const FinArg& fa) {
    if (fa.which < 0) return;                                  // uniform over the launch
    run(fa, nullptr);
    __syncthreads();
}

// Prologue with the coefficients delivered THROUGH LDS: src[] receives copies of the launch's source descriptors in which the folded
// source's scale / shift / q point into `lds` (flat pointers: the readers -- slot_ctx, plain_src_coef, the per-chunk tables -- load
// through them unchanged), so the kernel does not wait for its own global stores and read them back.  The kernel must read
// coefficients through src[], not through p.src[], and must not reuse the table's LDS while they are still read.  Falls back to the
// global path when the table does not fit.
__device__ __forceinline__ void conv_prologue_lds(const rd_conv_t& p, const FinArg& fa, float* lds, int cap_floats, rd_src_t (&src)[2]) {
    src[0] = p.src[0];
    src[1] = p.src[1];
    if (fa.which < 0) return;
    const int GC = fa.is_bwd ? fa.d.b.G * fa.d.b.C : fa.d.f.G * fa.d.f.C;
    const bool fits = (fa.is_bwd ? 3 : 2) * GC <= cap_floats;
    run(fa, fits ? lds : nullptr);
    if (!fits) {
        __syncthreads();
        return;
    }
    lds_barrier();
    if (fa.which == 0) {
        src[0].scale = lds;
        src[0].shift = lds + GC;
        if (fa.is_bwd) src[0].q = lds + 2 * GC;
    } else {
        src[1].scale = lds;
        src[1].shift = lds + GC;
        if (fa.is_bwd) src[1].q = lds + 2 * GC;
    }
}


// ---- host side: the by-value argument of a launch, built by the entry points from the descriptors' `fin` pointers
inline int make_arg(FinArg& a, const rd_src_t* srcs, int n) {
    memset(&a, 0, sizeof(a));
    a.which = -1;
    for (int i = 0; i < n; ++i) {
        if (!srcs[i].fin) continue;
        if (a.which >= 0) return -3;                           // one folded finalize per launch
        a.which = i;
        a.flags = srcs[i].fin_flags;
        a.is_bwd = srcs[i].mode == RD_SRC_BNBWD ? 1 : 0;
        if (srcs[i].mode == RD_SRC_RAW) return -3;
        if (a.is_bwd) memcpy(&a.d.b, srcs[i].fin, sizeof(rd_bn_bwd_t));
        else memcpy(&a.d.f, srcs[i].fin, sizeof(rd_bn_fwd_t));
        const int G = a.is_bwd ? a.d.b.G : a.d.f.G, C = a.is_bwd ? a.d.b.C : a.d.f.C;
        if (G < 1 || G > FIN_MAX_G || C < 1 || C != srcs[i].C) return -3;
        // slot_sums reads the copies in blocks of eight: a count that is not a whole number of blocks would run into the next group's
        // copies (or past the buffer).  The producer's stat_slots must not exceed this count, or part of its sums is never read (ramdsir.h)
        const int ns = a.is_bwd ? a.d.b.nslots : a.d.f.nslots;
        if (ns < 0 || ns > RD_STAT_SLOTS || ns % 8) return -3;
        if (!a.is_bwd && !a.d.f.training) return -3;           // eval mode has no batch statistics to finalize
    }
    return 0;
}
inline void no_arg(FinArg& a) {
    memset(&a, 0, sizeof(a));
    a.which = -1;
}
FinArg& current();                                             // this thread's argument, set by the entry point in progress (conv_api.hip)

}  // namespace rdfin

// statistic slots a launch adds into: rd_conv_t.stat_slots (0 = all RD_STAT_SLOTS)
__device__ __forceinline__ int rd_stat_nslots(int stat_slots) { return stat_slots > 0 ? stat_slots : RD_STAT_SLOTS; }
