// wgrad_tr.h -- [pixel][channel] LDS tile layout of the transpose-read weight-gradient kernels (wgrad.hip, wgrad_sym.hip): row pitch,
// half swizzle and the ds_read_b64_tr_b16 wrapper.  The layout is explained at wgrad_tr_kernel (wgrad.hip, "ROW PITCH AND SWIZZLE").
#pragma once
#include "conv_device.h"

namespace {

constexpr int tr_pitch(int channels) { return channels == 32 ? 64 : 128; }
// byte offset of 16-byte channel slot `slot` of pixel `pix` in a [pixel][C channels] tile
template <int C>
__device__ __forceinline__ int tr_off(int pix, int slot) {
    if constexpr (C == 32) return pix * 64 + slot * 16;
    else return pix * 128 + ((((slot >> 2) ^ (pix >> 1)) & 1) << 6) + (slot & 3) * 16;
}
// a lane's fragment base: pixel pix0 of the tile, 32-channel block blk, `sub` bytes into the block; flip = 1 where the pixels actually
// read sit an odd multiple of 2 further on (halo rows of odd index at a row length of 34: +34 r)
template <int C>
__device__ __forceinline__ int tr_frag(int pix0, int blk, int sub, int flip) {
    if constexpr (C == 32) return pix0 * 64 + sub;
    else return pix0 * 128 + (((blk ^ (pix0 >> 1) ^ flip) & 1) << 6) + sub;
}
typedef __attribute__((ext_vector_type(4))) short tr_s4;
typedef __attribute__((address_space(3))) tr_s4 tr_lds_s4;
__device__ __forceinline__ uint2 lds_tr(const char* p) {
    const tr_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_s4*)(p));
    return __builtin_bit_cast(uint2, v);
}


}  // namespace
