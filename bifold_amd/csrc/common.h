// Shared device/host helpers for libbifold_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bifold_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define BF_WAVE 64

// ---- host-side error plumbing -------------------------------------------
void bf_set_error(int hip_code, const char* fmt, ...);

#define BF_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            bf_set_error(0, __VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

#define BF_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            bf_set_error((int)e__, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return BF_ELAUNCH;                                                    \
        }                                                                         \
    } while (0)

static inline bool bf_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int bf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t bf_dtype_size(int dt) { return dt == BF_BF16 ? 2 : 4; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements <-> float4, for fp32 and bf16 storage
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ float4 load(const float* p) { return *(const float4*)p; }
    static __device__ __forceinline__ void store(float* p, float4 v) { *(float4*)p = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ float4 load(const bf16_t* p) {
        bf16x4 r = *(const bf16x4*)p;
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    static __device__ __forceinline__ void store(bf16_t* p, float4 v) {
        bf16x4 r;
        r[0] = (bf16_t)v.x; r[1] = (bf16_t)v.y; r[2] = (bf16_t)v.z; r[3] = (bf16_t)v.w;
        *(bf16x4*)p = r;
    }
};

__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float erf_gelu_f(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float erf_gelu_grad_f(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
