// Shared device/host helpers for librv3d_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rv3d.h"

// 16-bit OPERAND type of the MFMA kernels.  The library is built twice from the same sources (csrc/Makefile):
//   librv3d_hip.so      operands bf16 -- training (the reference trains in `precision: bf16-mixed`, conf/trainer/train.yaml:14)
//   librv3d_hip_f16.so  operands fp16 (-DRV_OPERAND_F16) -- inference under `torch.autocast(dtype=torch.float16)`, which is what
//                       the reference evaluates in (nn/arch/detector.py:329-340, conf/model/range_view.yaml:26 eval_precision: 16)
// Same tiles, same staging, same schedules: only the element conversions below and the MFMA opcode differ
// (v_mfma_f32_16x16x32_f16 issues at the bf16 rate).  The `bf16` in identifiers (bf16_t, f2bf, bf16x8 ...) reads "the
// 16-bit operand type" in the fp16 build.
typedef uint16_t bf16_t;  // storage type of the 16-bit tensors
#ifdef RV_OPERAND_F16
typedef _Float16 rv_elem_t;
#define RV_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 rv_elem_t;
#define RV_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(8))) rv_elem_t bf16x8;
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
// compute units of the CURRENT device (a read-only cache of a device property, one slot per device ordinal: a process that drives
// several devices gets each one's own count; relaxed atomics: two threads racing on the first call store the same value).  256 = an
// MI355X when no device is visible: the host-side planning entry points also run on GPU-less build boxes.  Note that hipGetDevice
// initialises the HIP runtime: the planning calls (rv_tap_launch_info, rv_tap_stats_rows, rv_tap_wgrad_workspace_bytes ...) are not
// fork-safe "pure" functions (include/rv3d.h says so).
static inline int rv_cu_count() {
    static std::atomic<int> cache[32];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < 32) {
        n = cache[dev].load(std::memory_order_relaxed);
        if (n > 0) return n;
    }
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev < 32) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}
// grid of a persistent launch: one workgroup per CU, a multiple of the 8 XCDs (the blockIdx -> XCD deals assume it), never empty
static inline int rv_persistent_grid() {
    const int g = rv_cu_count() & ~7;
    return g < 8 ? 8 : g;
}

// ---------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------
#ifdef RV_OPERAND_F16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float bf_lo(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)packed); }
__device__ __forceinline__ float bf_hi(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed >> 16)); }
// round-to-nearest-even (v_cvt_f16_f32); values beyond 65504 become infinities, as under torch.autocast(float16)
__device__ __forceinline__ bf16_t f2bf(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(bf16_t, h);
}
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
#endif
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
