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
typedef _Float16 f16_t;                                            // IEEE binary16: CLIP's native dtype, BF_F16
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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

// Zero `n` floats in stream order with a kernel of this library instead of hipMemsetAsync: the runtime's fill leaves the
// device idle for ~12 us between the two kernels it sits between (profiles/r06_step_gaps.txt: eight of them on the
// backward's dependent chain); an ordinary launch does not.  (A template so that every translation unit may hold it.)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void bf_zero_f32_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
static inline hipError_t bf_zero_f32_async(float* p, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bf_zero_f32_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
    return hipGetLastError();
}

static inline bool bf_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int bf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t bf_dtype_size(int dt) { return (dt == BF_BF16 || dt == BF_F16) ? 2 : 4; }
static inline bool bf_is_half(int dt) { return dt == BF_BF16 || dt == BF_F16; }       // a 16-bit MFMA operand type

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
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// The two 16-bit operand types share every byte image (LDS tiles, fragments, stores); what differs is the MFMA
// opcode, the transposing LDS read's builtin and the fp32 <-> 16-bit conversion.  H16<H> names the vector types.
template <typename H> struct H16;
template <> struct H16<bf16_t> { typedef bf16x4 x4; typedef bf16x8 x8; };
template <> struct H16<f16_t> { typedef f16x4 x4; typedef f16x8 x8; };

__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// ds_read_b64_tr_b16: the hardware-transposed LDS read (the instruction does not look at the element type)
template <typename H> __device__ __forceinline__ typename H16<H>::x4 lds_read_tr16(const void* lds_addr);
template <> __device__ __forceinline__ bf16x4 lds_read_tr16<bf16_t>(const void* lds_addr) {
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)lds_addr);
}
template <> __device__ __forceinline__ f16x4 lds_read_tr16<f16_t>(const void* lds_addr) {
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)lds_addr));
}

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

template <> struct Vec4<f16_t> {
    static __device__ __forceinline__ float4 load(const f16_t* p) {
        f16x4 r = *(const f16x4*)p;
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    static __device__ __forceinline__ void store(f16_t* p, float4 v) {
        f16x4 r;
        r[0] = (f16_t)v.x; r[1] = (f16_t)v.y; r[2] = (f16_t)v.z; r[3] = (f16_t)v.w;
        *(f16x4*)p = r;
    }
};

__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + expf(-1.702f * x)); }
__device__ __forceinline__ float erf_gelu_f(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float erf_gelu_grad_f(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- fast variants for the 16-bit paths (bf16 and fp16) -----------------------------------------------------
// The accurate erff/expf above cost ~40 VALU instructions per element; in a GEMM epilogue
// over 1.4e8 elements that is >150 us with the MFMA pipe idle.  16-bit outputs carry 2^-9 (bf16) / 2^-11 (fp16)
// relative rounding, so the Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7: one v_exp, one
// v_rcp, six FMAs) is three orders of magnitude below what the output can resolve.  The fp32
// ("exact") mode keeps the accurate functions.
__device__ __forceinline__ float fast_erfc_abs(float az, float& e2) {   // erfc(|z|), e2 = exp(-z^2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    e2 = __expf(-az * az);
    return p * t * e2;
}
__device__ __forceinline__ float erf_gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    float e2;
    const float ec = fast_erfc_abs(fabsf(z), e2);
    const float erf = copysignf(1.0f - ec, z);
    return 0.5f * x * (1.0f + erf);
}
__device__ __forceinline__ float erf_gelu_grad_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    float e2;                                                   // exp(-x^2 / 2)
    const float ec = fast_erfc_abs(fabsf(z), e2);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - ec, z));
    return fmaf(x * 0.39894228040143267794f, e2, cdf);
}
// gelu(x) and gelu'(x) from ONE erfc / exp evaluation (the forward epilogue that saves gelu' for the backward)
__device__ __forceinline__ void erf_gelu_both_fast(float x, float& g, float& gp) {
    const float z = x * 0.70710678118654752440f;
    float e2;                                                   // exp(-x^2 / 2)
    const float ec = fast_erfc_abs(fabsf(z), e2);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - ec, z));
    g = x * cdf;
    gp = fmaf(x * 0.39894228040143267794f, e2, cdf);
}
__device__ __forceinline__ void erf_gelu_both_f(float x, float& g, float& gp) {
    g = erf_gelu_f(x);
    gp = erf_gelu_grad_f(x);
}
__device__ __forceinline__ float quick_gelu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}

__device__ __forceinline__ float tanh_gelu_f(float x) {          // gelu_pytorch_tanh (HF SigLIP MLP)
    return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float tanh_gelu_fast(float x) {       // tanh(u) = 1 - 2 / (1 + e^{2u})
    const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}

// d/dx [0.5 x (1 + tanh u)],  u = c (x + 0.044715 x^3):  0.5 (1 + t) + 0.5 x (1 - t^2) c (1 + 0.134145 x^2)
__device__ __forceinline__ float tanh_gelu_grad_f(float x) {
    const float t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * x * x);
}
__device__ __forceinline__ float tanh_gelu_grad_fast(float x) {
    const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    return fmaf(0.5f * x * (1.0f - t * t), 0.7978845608028654f * fmaf(0.134145f * x, x, 1.0f), 0.5f * (1.0f + t));
}

template <typename TD> struct ActMath;      // TD = compute dtype of the call
template <> struct ActMath<float> {
    static __device__ __forceinline__ float quick(float x) { return quick_gelu_f(x); }
    static __device__ __forceinline__ float erf(float x) { return erf_gelu_f(x); }
    static __device__ __forceinline__ float erf_grad(float x) { return erf_gelu_grad_f(x); }
    static __device__ __forceinline__ void erf_both(float x, float& g, float& gp) { erf_gelu_both_f(x, g, gp); }
    static __device__ __forceinline__ float tanh_gelu(float x) { return tanh_gelu_f(x); }
    static __device__ __forceinline__ float tanh_grad(float x) { return tanh_gelu_grad_f(x); }
};
template <> struct ActMath<bf16_t> {
    static __device__ __forceinline__ float quick(float x) { return quick_gelu_fast(x); }
    static __device__ __forceinline__ float erf(float x) { return erf_gelu_fast(x); }
    static __device__ __forceinline__ float erf_grad(float x) { return erf_gelu_grad_fast(x); }
    static __device__ __forceinline__ void erf_both(float x, float& g, float& gp) { erf_gelu_both_fast(x, g, gp); }
    static __device__ __forceinline__ float tanh_gelu(float x) { return tanh_gelu_fast(x); }
    static __device__ __forceinline__ float tanh_grad(float x) { return tanh_gelu_grad_fast(x); }
};
template <> struct ActMath<f16_t> : ActMath<bf16_t> {};          // same fast forms: 1.5e-7 absolute is below 2^-11 too

// "Opt a kernel in to more than 64 KB of dynamic LDS" is a per-DEVICE attribute of the function: the guards that make the
// call once (and again when a larger size is asked for) are kept per device, so one process may drive several devices
// (include/bifold_hip.h).  A device index beyond the table simply sets the attribute on every call.
#define BF_MAX_DEVICES 16
struct BfPerDevice {
    size_t v[BF_MAX_DEVICES];
    size_t spill;
    size_t& cur() {
        int d = 0;
        (void)hipGetDevice(&d);
        if (d < 0 || d >= BF_MAX_DEVICES) { spill = 0; return spill; }
        return v[d];
    }
};

// ---- replayable dropout masks ---------------------------------------------------------------------------------------
// The keep decision of element i of a tensor is a pure function of (seed, i).  One splitmix64 value serves FOUR consecutive
// elements (i >> 2 is hashed; element i & 3 takes 15 of its bits from the 16-bit field i & 3): keep iff those 15 bits
// >= p * 2^15.  A lane that holds four (or eight) consecutive 16-bit elements therefore pays one (two) hashes and turns
// them into AND-masks with packed 16-bit arithmetic (dropout_mask4): the adapters' dropout inside the LoRA kernels
// (lora.hip) costs a quarter of what one hash per element did (round 6; the hash, not HBM, bounded bf_dropout: 225 us for
// 283 M elements).  The backward (and a CPU checker, through bf_dropout_mask) regenerates the mask instead of storing it.
// Used by bf_dropout / bf_dropout_rows (the fusion blocks' and the embeddings' nn.Dropout), the bf_lora_* kernels (PEFT's
// lora_dropout) and bf_attention_dropout_fwd / _bwd (nn.Dropout on the attention probabilities: element index
// ((b H + h) L + i) L + j).
__device__ __forceinline__ unsigned long long dropout_hash(unsigned long long q, unsigned long long seed) {
    unsigned long long z = q + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool dropout_keep(unsigned long long idx, unsigned long long seed, unsigned thr) {
    const unsigned long long z = dropout_hash(idx >> 2, seed);
    return (((unsigned)(z >> (16 * (unsigned)(idx & 3))) & 0xFFFFu) >> 1) >= thr;
}
// AND-masks (0xFFFF per kept element) for the four 16-bit elements idx .. idx + 3 (idx % 4 == 0) as two packed words;
// thr_m1 = dropout_threshold_packed(p).  field >> 1 >= thr  <=>  (thr - 1) - (field >> 1) < 0.
typedef short bf_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 dropout_mask4(unsigned long long idx, unsigned long long seed, unsigned thr_m1) {
    const unsigned long long z = dropout_hash(idx >> 2, seed);
    const bf_i16x2 t = __builtin_bit_cast(bf_i16x2, thr_m1);
    const bf_i16x2 f0 = __builtin_bit_cast(bf_i16x2, ((unsigned)z >> 1) & 0x7FFF7FFFu);
    const bf_i16x2 f1 = __builtin_bit_cast(bf_i16x2, ((unsigned)(z >> 32) >> 1) & 0x7FFF7FFFu);
    uint2 m;
    m.x = __builtin_bit_cast(unsigned, (bf_i16x2)((t - f0) >> 15));
    m.y = __builtin_bit_cast(unsigned, (bf_i16x2)((t - f1) >> 15));
    return m;
}
static inline unsigned dropout_threshold(float p) { return (unsigned)((double)p * 32768.0 + 0.5); }
// (thr - 1) in both halves of a word, for dropout_mask4; thr = 0 (p < 2^-16) keeps everything.
static inline unsigned dropout_threshold_packed(float p) {
    const unsigned t = (dropout_threshold(p) - 1u) & 0xFFFFu;
    return t | (t << 16);
}
