// common.h -- device-side helpers shared by every kernel file (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <utility>
#include <stdint.h>
#include <stdio.h>

// ---- every kernel launch of the library goes through rd_launch.  A lane fork of the launch list (runlist.hip) normally costs the
// producing stream an event RECORD -- a barrier packet of its own between two dependent kernels: +3.0 us per fork on a chain of
// dependent kernels, +5.5 us with the waiting stream's first kernel starting beside it (scripts/probe/ext_event.hip), ~40 forks in a
// training step.  When the launch list knows that a fork follows a launch it sets rd_tls_stop_event, and the launch binds that event
// to the kernel's OWN dispatch packet (hipExtLaunchKernelGGL stopEvent: +0.06 us): the waiting stream's hipStreamWaitEvent then needs
// no record.  An entry point with several launches binds the event to each in turn; the last binding is the one a later wait sees.
extern thread_local hipEvent_t rd_tls_stop_event;           // runlist.hip; null outside a launch the list has marked
extern thread_local int rd_tls_stop_used;                   // set when a launch has bound the event
template <typename F, typename... A>
inline void rd_launch(F kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, A&&... args) {
    if (hipEvent_t e = rd_tls_stop_event) {
        rd_tls_stop_used = 1;
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)shmem, st, nullptr, e, 0, std::forward<A>(args)...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, std::forward<A>(args)...);
    }
}

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RD_MAX_GROUPS 16

#define RD_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            fprintf(stderr, "[ramdsir] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,         \
                    hipGetErrorString(_e));                                                 \
            return (int)_e;                                                                 \
        }                                                                                   \
    } while (0)

// ---------------------------------------------------------------------------------- element types
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }   // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ unsigned bf16_bits(float v) {
    bf16_t b = (bf16_t)v;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}

// A "slot" is 16 bytes of one pixel's channel vector: 4 fp32 or 8 bf16 channels.
template <typename T> struct Slot;
template <> struct Slot<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
        f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Slot<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    static __device__ __forceinline__ unsigned pk2(float a, float b) {       // one v_cvt_pk_bf16_f32 (RNE)
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        f2 v = {a, b};
        b2 r = __builtin_convertvector(v, b2);
        return __builtin_bit_cast(unsigned, r);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pk2(f[0], f[1]), pk2(f[2], f[3]), pk2(f[4], f[5]), pk2(f[6], f[7]));
    }
};

// ReLU (slope 0) / LeakyReLU (0 < slope < 1) / identity (slope 1): max(v, v*slope), two VALU ops, no select
__device__ __forceinline__ float act_fn(float v, float slope) { return fmaxf(v, v * slope); }
__device__ __forceinline__ float act_grad(float v, float slope) { return v > 0.f ? 1.f : slope; }

// group of image n given group start offsets gs[0..G] (gs[G] == N)
struct GroupMap {
    int G;
    int gs[RD_MAX_GROUPS + 1];
};
__device__ __forceinline__ int group_of(const GroupMap& gm, int n) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < RD_MAX_GROUPS; ++i)
        if (i < gm.G && n >= gm.gs[i]) g = i;
    return g;
}

// bilinear x2, align_corners=False (torch upsample_bilinear2d, scale 0.5):  src = (dst+0.5)/2-0.5 clamped at 0
__device__ __forceinline__ void up2_coord(int d, int n_src, int& i0, int& i1, float& lam) {
    float s = ((float)d + 0.5f) * 0.5f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < n_src - 1 ? 1 : 0);
    lam = s - (float)i0;
}

// block-wide sum of one float per thread (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Dispatch overrides (experiments, the forced-dispatch tests) exist only in the DEBUG build of the library
// (`make debug` -> libramdsir_hip_dbg.so, -DRD_DEBUG_SWITCHES, loaded with RAMDSIR_DEBUG_LIB=1): there rd_switch reads the
// environment variable; in the product library it IS the default -- no environment lookups, the branches fold away.
#ifdef RD_DEBUG_SWITCHES
#include <stdlib.h>
inline int rd_switch(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#else
constexpr int rd_switch(const char*, int dflt) { return dflt; }
#endif

// compute units of the current device (256 on MI355X), queried once
inline int rd_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        else
            n = 256;
    }
    return n;
}
