// conv_api.hip -- extern "C" entry points of the convolution family (rd_conv, rd_wgrad, weight packing).
#include "common.h"
#include "../../include/ramdsir.h"
#include "conv_dispatch.h"
#include "bn_fin.h"

namespace {

// ------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_weights_kernel(const float* w, T* out, int Cout, int Cin, int taps, int transpose, int RowPad, int ColPad) {
    // K-chunk-major: the CK = 64 B / sizeof(T) columns one K step of the conv kernels consumes are contiguous for every
    // (tap, row), and all taps x rows of a chunk follow each other -> a workgroup's weight fetch for one chunk is one
    // contiguous block (it was 64-byte pieces 2*ColPad bytes apart, the dominant L2 traffic of the 25x25 / 50x50 levels)
    // forward:   out[col/CK][tap][n<RowPad(Cout)][col%CK]   = w[n][col][tap]
    // transpose: out[col/CK][tap'][c<RowPad(Cin)][col%CK]   = w[col][c][taps-1-tap']
    constexpr int CK = 64 / (int)sizeof(T);
    const int total = taps * RowPad * ColPad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cc = i % CK, row = (i / CK) % RowPad, tap = (i / (CK * RowPad)) % taps, col = (i / (CK * RowPad * taps)) * CK + cc;
        float v = 0.f;
        if (!transpose) {
            if (row < Cout && col < Cin) v = w[((size_t)row * Cin + col) * taps + tap];
        } else {
            if (row < Cin && col < Cout) v = w[((size_t)col * Cin + row) * taps + (taps - 1 - tap)];
        }
        out[i] = from_f<T>(v);
    }
}

// all convs of the network in one launch: entry e owns packed elements [start_e, start_{e+1}).  A thread produces one
// 16-byte vector (V = 8 bf16 / 4 fp32 consecutive input channels of one (chunk, tap, row): CK is a multiple of V, so a
// vector never straddles rows or entries): one table search and one store per V elements
template <typename T>
__global__ void pack_weights_batched_kernel(const float* params, T* packed, const rd_pack_entry_t* tab, int n_entries,
                                            int64_t total) {
    constexpr int CK = 64 / (int)sizeof(T), V = 16 / (int)sizeof(T);
    const int64_t nvec = total / V;
    // the entry starts in LDS: the binary search is 7-8 DEPENDENT reads per vector, and from global memory that chain was the kernel
    // (30 us for 9 M elements; `start` fits 32 bits: the packed arena is < 2^31 elements)
    extern __shared__ int s_start[];
    for (int k = threadIdx.x; k < n_entries; k += blockDim.x) s_start[k] = (int)tab[k].start;
    __syncthreads();
    for (int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iv < nvec; iv += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = iv * V;
        int lo = 0, hi = n_entries - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int64_t)s_start[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const rd_pack_entry_t e = tab[lo];
        const int j = (int)(i - e.start);
        const int cc = j % CK, row = (j / CK) % e.RowPad, tap = (j / (CK * e.RowPad)) % e.taps, col0 = (j / (CK * e.RowPad * e.taps)) * CK + cc;
        const float* w = params + e.src_off;
        float v[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int col = col0 + k;
            v[k] = 0.f;
            if (!e.transpose) {
                if (row < e.Cout && col < e.Cin) v[k] = w[((size_t)row * e.Cin + col) * e.taps + tap];
            } else {
                if (row < e.Cin && col < e.Cout) v[k] = w[((size_t)col * e.Cin + row) * e.taps + (e.taps - 1 - tap)];
            }
        }
        *reinterpret_cast<uint4*>(packed + e.dst_off + j) = Slot<T>::pack(v);
    }
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

thread_local rdfin::FinArg t_fin_arg;                          // the folded-finalize argument of the entry point in progress on this thread

}  // namespace

rdfin::FinArg& rdfin::current() { return t_fin_arg; }


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
        rd_launch(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, w_oihw, (bf16_t*)packed, Cout, Cin, taps,
                           transpose, RowPad, ColPad);
    else
        rd_launch(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, w_oihw, (float*)packed, Cout, Cin, taps,
                           transpose, RowPad, ColPad);
    return (int)hipGetLastError();
}

int rd_pack_weights_batched(const float* params, void* packed, const rd_pack_entry_t* table_dev, int n_entries, int64_t total,
                            int dtype, void* stream) {
    if (n_entries < 1 || total < 1) return -1;
    if (total % 8) return -2;                                   // entries are whole 64-byte K chunks
    if (total >= (1ll << 31) || n_entries > 8192) return -2;    // 32-bit entry starts in LDS
    int64_t blocks = (total / (dtype == RD_BF16 ? 8 : 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RD_BF16)
        rd_launch(pack_weights_batched_kernel<bf16_t>, dim3((int)blocks), dim3(256), n_entries * sizeof(int), st, params, (bf16_t*)packed, table_dev,
                           n_entries, total);
    else
        rd_launch(pack_weights_batched_kernel<float>, dim3((int)blocks), dim3(256), n_entries * sizeof(int), st, params, (float*)packed, table_dev,
                           n_entries, total);
    return (int)hipGetLastError();
}

int rd_conv(const rd_conv_t* p, int dtype, void* stream) {
    if (!p || (p->taps != 9 && p->taps != 1) || p->G < 1 || p->G > RD_MAX_GROUPS || p->nsrc < 1 || p->nsrc > 2) return -1;
    const int ck = dtype == RD_BF16 ? 32 : 16;
    if (p->CinPad % ck || p->CoutPad % 32 || p->CinPad < p->Cin || p->CoutPad < p->Cout) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (p->stat_slots < 0 || p->stat_slots > RD_STAT_SLOTS) return -2;               // the kernels add into slot (index mod stat_slots) of [RD_STAT_SLOTS] copies
    if (rdfin::make_arg(rdfin::current(), p->src, p->nsrc)) return -3;                // rd_src_t.fin: copied into the kernel arguments
    if (p->w_tap_rows && !(p->CinPad == ck && p->CoutPad == 32 && p->emode == 1 && p->w_tap_rows >= p->CoutPad)) return -2;   // ramdsir.h
    if (p->CinPad == ck && p->CoutPad == 32) return rd_conv_small_dispatch(*p, dtype, st);   // one K chunk, one N block
    return rd_conv_big_dispatch(*p, dtype, st);
}

int rd_conv_honours_src_out(const rd_conv_t* p, int dtype) {
    if (!p || (p->taps != 9 && p->taps != 1) || p->G < 1 || p->G > RD_MAX_GROUPS || p->nsrc < 1 || p->nsrc > 2) return 0;
    const int ck = dtype == RD_BF16 ? 32 : 16;
    if (p->CinPad % ck || p->CoutPad % 32 || p->CinPad < p->Cin || p->CoutPad < p->Cout || p->w_tap_rows) return 0;
    if (p->CinPad == ck && p->CoutPad == 32) return 0;           // the small-channel kernels
    return rd_conv_big_stores_sources(*p, dtype) ? 1 : 0;
}

int64_t rd_wgrad_workspace(const rd_wgrad_t* p, int dtype) {
    return rd_wgrad_ws_bytes(*p, dtype);
}

int rd_wgrad(const rd_wgrad_t* p, int dtype, void* stream) {
    if (!p || (p->taps != 9 && p->taps != 1) || p->G < 1 || p->G > RD_MAX_GROUPS || p->na < 1 || p->na > 2) return -1;
    if (p->a[0].fin || (p->na > 1 && p->a[1].fin)) return -3;                         // only dz may carry a folded finalize (ramdsir.h)
    if (rdfin::make_arg(rdfin::current(), &p->dz, 1)) return -3;
    return rd_wgrad_dispatch(*p, dtype, (hipStream_t)stream);
}

// ---- backward of a small-channel 3x3 conv in one launch (conv_fused.hip)
int rd_conv_bwd_fused_ok(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype) {
    if (!dgrad || !wgrad || dgrad->G < 1 || dgrad->G > RD_MAX_GROUPS) return 0;
    return rd_bwd_fused_ok(*dgrad, *wgrad, dtype) ? 1 : 0;
}

int64_t rd_conv_bwd_fused_workspace(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype) {
    if (!rd_conv_bwd_fused_ok(dgrad, wgrad, dtype)) return 0;
    return rd_bwd_fused_ws_bytes(*dgrad, *wgrad);
}

int rd_conv_bwd_fused(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype, void* stream) {
    if (!rd_conv_bwd_fused_ok(dgrad, wgrad, dtype) || !wgrad->partial || !wgrad->dW) return -1;
    if (dgrad->stat_slots < 0 || dgrad->stat_slots > RD_STAT_SLOTS) return -2;
    if (rdfin::make_arg(rdfin::current(), dgrad->src, dgrad->nsrc)) return -3;          // the gradient descriptor's dz carries it; wgrad->dz.fin is ignored
    return rd_bwd_fused_dispatch(*dgrad, *wgrad, (hipStream_t)stream);
}

int rd_conv_bwd_fused_reduce(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype, void* stream) {
    if (!rd_conv_bwd_fused_ok(dgrad, wgrad, dtype) || !wgrad->partial || !wgrad->dW) return -1;
    return rd_bwd_fused_reduce_dispatch(*dgrad, *wgrad, (hipStream_t)stream);
}

}  // extern "C"

