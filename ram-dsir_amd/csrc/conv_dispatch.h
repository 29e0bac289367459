// conv_dispatch.h -- host-side entry points of the conv / wgrad translation units (C++ linkage, internal).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ramdsir.h"
int rd_conv_small_dispatch(const rd_conv_t& p, int dtype, hipStream_t st);   // Cin <= one chunk, Cout <= 32
int rd_conv_big_dispatch(const rd_conv_t& p, int dtype, hipStream_t st);     // everything else
int rd_wgrad_dispatch(const rd_wgrad_t& p, int dtype, hipStream_t st);
int64_t rd_wgrad_ws_bytes(const rd_wgrad_t& p, int dtype);
// persistent software-pipelined 3x3 kernel for bf16 launches with CoutPad % 64 == 0 and plain sources (conv_pp.hip);
// RD_CONV_PP_NA when the launch does not qualify (the caller falls back to conv_big's kernels)
constexpr int RD_CONV_PP_NA = -1000;
int rd_conv_pp_dispatch(const rd_conv_t& p, hipStream_t st);
