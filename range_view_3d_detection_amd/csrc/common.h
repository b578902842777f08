// Shared device/host helpers for librv3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rv3d.h"

typedef uint16_t bf16_t;  // storage type of bf16 tensors
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---------------------------------------------------------------------------------------
// error reporting (thread-local string, int status) -- the contract of include/rv3d.h
// ---------------------------------------------------------------------------------------
void rv_set_error(const char* fmt, ...);
#define RV_FAIL(...)               \
    do {                           \
        rv_set_error(__VA_ARGS__); \
        return 1;                  \
    } while (0)
#define RV_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) RV_FAIL(__VA_ARGS__); \
    } while (0)
#define RV_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) RV_FAIL("%s: %s", name, hipGetErrorString(e__));    \
    } while (0)

static inline int rv_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int rv_pad32(int c) { return (c + 31) & ~31; }

// ---------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// Opaque pass-through: consumers of `v` cannot be scheduled above this point.  Used on prefetch registers so that
// hipcc does not hoist the unpack/transform of freshly loaded data (and with it an s_waitcnt vmcnt(0)) in front of
// the MFMA section the loads are supposed to overlap with.
__device__ __forceinline__ void pin_here(u32x4& v) { asm volatile("" : "+v"(v)); }

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
