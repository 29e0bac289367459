// conv_lean.hip -- conv_pf_kernel (conv_pf.h) instantiated with the register epilogues of conv_epilogue.h:
// forward launches (EP 1) and gradient launches whose destinations are all plain tensors (EP 2), bf16, whole
// 32- / 64-channel output tiles.  Everything else stays with the LDS-staged epilogue in conv_big.hip.
#include "conv_device.h"
#ifdef RD_DEBUG_SWITCHES
// debug build: shader-clock stamps of five workgroups (one per round of resident workgroups at 2560 tiles) of the launches whose
// grid.x equals pf_trace_key (rd_debug_pf_trace; scripts/pf_trace.py)
__device__ unsigned long long pf_trace[8][16];
__device__ int pf_trace_key;
#define PF_T(ev) do { \
        if (pf_trace_key == (int)gridDim.x && threadIdx.x == 0 && blockIdx.y == 0 && (ev) < 15) { \
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z; \
            if (lin % 600 == 7 && lin / 600 < 8) { pf_trace[lin / 600][ev] = __builtin_readcyclecounter(); if ((ev) == 0) pf_trace[lin / 600][15] = wall_clock64(); } \
        } \
    } while (0)
#endif
#include "conv_epilogue.h"
#include "conv_dispatch.h"
#include "conv_pf.h"

namespace {

template <int TAPS, int NB, int TS>
int launch_lean(const rd_conv_t& p, int ep, int nq, hipStream_t st) {
    typedef bf16_t T;
    const size_t lds = conv_pf_lds<TAPS, NB, TS>(p);
    if (lds > (size_t)72 * 1024) return RD_CONV_PP_NA;
    constexpr int THt = TileGeo<TS>::H, TWt = TileGeo<TS>::W;
    dim3 grid(((p.W + TWt - 1) / TWt) * ((p.H + THt - 1) / THt), p.CoutPad / (NB * 32), p.N);
    static bool attr = false;
    if (!attr) {
        const int lds_max = 72 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 1, 1, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 1, 2, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 2, 2, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        attr = true;
    }
    if (ep == 1) rd_launch((conv_pf_kernel<T, TAPS, NB, 1, 1, TS>), grid, dim3(256), lds, st, p, rdfin::current());
    else if (nq == 1) rd_launch((conv_pf_kernel<T, TAPS, NB, 1, 2, TS>), grid, dim3(256), lds, st, p, rdfin::current());
    else rd_launch((conv_pf_kernel<T, TAPS, NB, 2, 2, TS>), grid, dim3(256), lds, st, p, rdfin::current());
    return (int)hipGetLastError();
}

// 10 x 25 tiles where they put clearly more of the MFMA lanes on image pixels (sides 25 / 50 / 100 / 200: 0.61-0.89 -> 0.81-0.98).
// conv_pf_kernel launches run in rounds of the resident workgroups (two per CU), so ALONE fewer tiles only help when they remove
// a round (scripts/layer_bench.py, RD_CONV_FLAT_TILES 0 / 1: dec.convu4.conv3 dgrad 104 -> 95 us, enc.convd3.* 52-56 -> 48-52,
// dec.convu3.conv1 72 -> 68, others +-5 %: the sum over the family is unchanged) -- but in the STEP every workgroup not launched
// is CU time the other two lanes get: 5.07-5.08 ms with 8 x 32 only, 5.06 with the whole-round rule (mode 2), 5.03 with mode 1
// (scripts/sweep_opts.sh, alternating).  The persistent conv_ws_kernel has the same choice (conv_pp.hip, RD_CONV_WS_FLAT).
template <int NB>
inline bool flat_tiles(const rd_conv_t& p) {
    const int mode = rd_switch("RD_CONV_FLAT_TILES", 1);       // 0 never, 1 whenever more lanes are live, 2 only when a round goes away
    if (mode == 0) return false;
    const double e1 = tile_efficiency(p.H, p.W, TileGeo<1>::H, TileGeo<1>::W), e0 = tile_efficiency(p.H, p.W, TH, TW);
    if (e1 <= 1.04 * e0) return false;
    if (mode == 1) return true;
    const long nblk = p.CoutPad / (NB * 32), slots = 2L * rd_num_cus();
    const long w0 = (long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.N * nblk;
    const long w1 = (long)((p.W + TileGeo<1>::W - 1) / TileGeo<1>::W) * ((p.H + TileGeo<1>::H - 1) / TileGeo<1>::H) * p.N * nblk;
    return (w1 + slots - 1) / slots < (w0 + slots - 1) / slots;
}

}  // namespace

int rd_conv_pf_lean_dispatch(const rd_conv_t& p, bool nb2, hipStream_t st) {
    const int nq = conv_pf_kind(p);
    if (!nq) return RD_CONV_PP_NA;
    const int ep = rd_conv_lean_mode(p, nb2 ? 64 : 32);
    if (!ep || (ep == 1 && nq != 1)) return RD_CONV_PP_NA;
    if (p.taps == 9) {
        if (nb2 ? flat_tiles<2>(p) : flat_tiles<1>(p)) return nb2 ? launch_lean<9, 2, 1>(p, ep, nq, st) : launch_lean<9, 1, 1>(p, ep, nq, st);
        return nb2 ? launch_lean<9, 2, 0>(p, ep, nq, st) : launch_lean<9, 1, 0>(p, ep, nq, st);
    }
    return nb2 ? launch_lean<1, 2, 0>(p, ep, nq, st) : launch_lean<1, 1, 0>(p, ep, nq, st);
}

#ifdef RD_DEBUG_SWITCHES
extern "C" int rd_debug_pf_trace(int key, unsigned long long* out) {       // debug library only: arm (out == null) or read 8 x 16 stamps
    if (!out) return (int)hipMemcpyToSymbol(HIP_SYMBOL(pf_trace_key), &key, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_trace), sizeof(unsigned long long) * 8 * 16);
}
#endif
