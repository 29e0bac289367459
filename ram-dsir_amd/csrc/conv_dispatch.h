// conv_dispatch.h -- host-side entry points of the conv / wgrad translation units (C++ linkage, internal).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ramdsir.h"
int rd_conv_small_dispatch(const rd_conv_t& p, int dtype, hipStream_t st);   // Cin <= one chunk, Cout <= 32
int rd_conv_big_dispatch(const rd_conv_t& p, int dtype, hipStream_t st);     // everything else
int rd_wgrad_dispatch(const rd_wgrad_t& p, int dtype, hipStream_t st);
// wgrad_sym.hip: the 128 x 64-block kernel, grid (gx pixel splits, CoutPadW / 128, CinPadW / 64); partials in wgrad.hip's layout
int rd_wgrad_sym_launch(const rd_wgrad_t& p, int gx, int CoutPadW, int CinPadW, hipStream_t st);
int64_t rd_wgrad_ws_bytes(const rd_wgrad_t& p, int dtype);
// split reduction of per-workgroup weight-gradient sums partial[nsplit][taps][CoutPadW][CinPadW] -> dW (wgrad.hip)
int rd_wgrad_reduce_launch(const float* partial, float* dW, int nsplit, int taps, int Cout, int Cin, int CoutPadW, int CinPadW, float beta,
                           hipStream_t st);
// fused dgrad + weight gradient of the small-channel 3x3 convs (conv_fused.hip)
bool rd_bwd_fused_ok(const rd_conv_t& dgrad, const rd_wgrad_t& wgrad, int dtype);
int64_t rd_bwd_fused_ws_bytes(const rd_conv_t& dgrad, const rd_wgrad_t& wgrad);
int rd_bwd_fused_dispatch(const rd_conv_t& dgrad, const rd_wgrad_t& wgrad, hipStream_t st);
int rd_bwd_fused_reduce_dispatch(const rd_conv_t& dgrad, const rd_wgrad_t& wgrad, hipStream_t st);
// persistent software-pipelined 3x3 kernel for bf16 launches with CoutPad % 64 == 0 and plain sources (conv_pp.hip);
// RD_CONV_PP_NA when the launch does not qualify (the caller falls back to conv_big's kernels)
constexpr int RD_CONV_PP_NA = -1000;
// forward kernel of the small-channel 3x3 convs with whole channel slots (conv_small_fwd.hip); RD_CONV_PP_NA when the launch does not qualify
int rd_conv_small_fwd_dispatch(const rd_conv_t& p, int dtype, hipStream_t st);
int rd_conv_pp_dispatch(const rd_conv_t& p, hipStream_t st);
bool rd_conv_ws_takes(const rd_conv_t& p);           // conv_pp.hip: the launch runs on conv_ws_kernel (the kernel that writes rd_src_t.out)
bool rd_conv_big_takes_ws(const rd_conv_t& p, int dtype);   // conv_big.hip: ... after conv_big's own routing (64-wide tiles, bf16, 3x3)
bool rd_conv_ws_stores_sources(const rd_conv_t& p);  // conv_pp.hip: ... on an instantiation that WRITES rd_src_t.out (dispatch and query share it)
bool rd_conv_big_stores_sources(const rd_conv_t& p, int dtype);

// Register ("lean") epilogues: accumulators leave as 16-byte NHWC vectors straight from registers (MFMA roles swapped:
// weights x pixels, rd_half_swap regroup) instead of through the LDS-staged epilogue of conv_epilogue.h.
// 0 = not eligible, 1 = forward, 2 = gradient with plain destinations only.  NT = output channels per workgroup.
inline int rd_conv_lean_mode(const rd_conv_t& p, int NT) {
    if (p.Cout != p.CoutPad || p.Cout % NT) return 0;
    if (p.emode == 0) return (((uintptr_t)p.out) & 15) == 0 ? 1 : 0;
    if (p.c_split % 16) return 0;
    bool any = false;
    for (int i = 0; i < 2; ++i) {
        const rd_dst_t& d = p.dst[i];
        if (i == 1 && p.c_split >= p.Cout) break;              // dst[1] unused
        if (d.kind == RD_DST_NONE) continue;
        any = true;
        const int width = i == 0 ? (p.c_split < p.Cout ? p.c_split : p.Cout) : p.Cout - p.c_split;
        if (d.kind != RD_DST_PLAIN || d.Cd % 8 || d.Cd < width || (((uintptr_t)d.g | (uintptr_t)d.z) & 15)) return 0;
        if (d.scale && (((uintptr_t)d.scale | (uintptr_t)d.shift) & 3)) return 0;
    }
    return any ? 2 : 0;
}
// conv_pf_kernel with a register epilogue (conv_lean.hip); RD_CONV_PP_NA when the launch does not qualify
int rd_conv_pf_lean_dispatch(const rd_conv_t& p, bool nb2, hipStream_t st);
