// conv_lean.hip -- conv_pf_kernel (conv_pf.h) instantiated with the register epilogues of conv_epilogue.h:
// forward launches (EP 1) and gradient launches whose destinations are all plain tensors (EP 2), bf16, whole
// 32- / 64-channel output tiles.  Everything else stays with the LDS-staged epilogue in conv_big.hip.
#include "conv_device.h"
#include "conv_epilogue.h"
#include "conv_dispatch.h"
#include "conv_pf.h"

namespace {

template <int TAPS, int NB>
int launch_lean(const rd_conv_t& p, int ep, int nq, hipStream_t st) {
    typedef bf16_t T;
    const size_t lds = conv_pf_lds<TAPS, NB>(p);
    if (lds > (size_t)72 * 1024) return RD_CONV_PP_NA;
    dim3 grid(((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH), p.CoutPad / (NB * 32), p.N);
    static bool attr = false;
    if (!attr) {
        const int lds_max = 72 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pf_kernel<T, TAPS, NB, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        attr = true;
    }
    if (ep == 1) hipLaunchKernelGGL((conv_pf_kernel<T, TAPS, NB, 1, 1>), grid, dim3(256), lds, st, p);
    else if (nq == 1) hipLaunchKernelGGL((conv_pf_kernel<T, TAPS, NB, 1, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_pf_kernel<T, TAPS, NB, 2, 2>), grid, dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

}  // namespace

int rd_conv_pf_lean_dispatch(const rd_conv_t& p, bool nb2, hipStream_t st) {
    const int nq = conv_pf_kind(p);
    if (!nq) return RD_CONV_PP_NA;
    const int ep = rd_conv_lean_mode(p, nb2 ? 64 : 32);
    if (!ep || (ep == 1 && nq != 1)) return RD_CONV_PP_NA;
    if (p.taps == 9) return nb2 ? launch_lean<9, 2>(p, ep, nq, st) : launch_lean<9, 1>(p, ep, nq, st);
    return nb2 ? launch_lean<1, 2>(p, ep, nq, st) : launch_lean<1, 1>(p, ep, nq, st);
}
