// fp8 (OCP e4m3) forward path of the FROZEN towers (BASELINE configs[4]: "fp8 MFMA weights / activations"):
//   * bf_layernorm_fwd_fp8   LayerNorm whose output goes straight to e4m3 with ONE scale per token row
//   * bf_quantize_rows_fp8   weights [N, K] -> e4m3 with one scale per output channel (done once per load)
//   * bf_gemm_fp8            C = act((sa[m] sb[n]) * sum_k a8[m,k] b8[n,k] + bias[n]) in bf16 / fp32, on
//                            v_mfma_scale_f32_16x16x128_f8f6f4 (block scales fixed at 2^0: the scaled instruction is
//                            the one that runs fp8 at twice the bf16 rate; the real scales are per row / per
//                            channel and are applied to the fp32 accumulator in the epilogue)
// Row / channel scales keep an outlier token or channel from costing every other one its mantissa: LayerNorm
// outputs have |max| / rms ~ 4-10 per row, i.e. 2-3 of e4m3's 8 exponent steps.
// Only GEMMs whose A operand comes out of a LayerNorm use this path (QKV in-projection, MLP fc1: 53 % of the
// towers' FLOPs); an operand produced in bf16 by another kernel would have to be re-read to be quantised, which
// costs more HBM bytes than fp8 saves on this HBM-bound path.
#include "common.h"

#include <math.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) void* lds_void_ptr8;

#define FP8_MAX 448.0f

int bf_gemm_fp8_p8_try(const void* a8, const float* sa, const void* b8, const float* sb, void* c, int c_dtype,
                       const float* bias, int act, int m, int n, int k, int lda, int ldb, int ldc, bf_stream_t stream);

namespace {

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = fminf(FP8_MAX, fmaxf(-FP8_MAX, a)); b = fminf(FP8_MAX, fmaxf(-FP8_MAX, b));
    c = fminf(FP8_MAX, fmaxf(-FP8_MAX, c)); d = fminf(FP8_MAX, fmaxf(-FP8_MAX, d));
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (unsigned)p;
}

// ------------------------------------------------------------------ LayerNorm -> e4m3 + row scale
// One wave per row, NC float4 chunks per lane (cols <= NC * 256).
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_fp8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, uint8_t* __restrict__ y,
                                                         float* __restrict__ scale, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = cols >> 2;
    const float* xr = x + (size_t)row * cols;
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        v[i] = c < nvec ? *(const float4*)(xr + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 g = *(const float4*)(gamma + c * 4), bb = *(const float4*)(beta + c * 4);
            v[i].x = (v[i].x - mean) * rstd * g.x + bb.x; v[i].y = (v[i].y - mean) * rstd * g.y + bb.y;
            v[i].z = (v[i].z - mean) * rstd * g.z + bb.z; v[i].w = (v[i].w - mean) * rstd * g.w + bb.w;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    uint8_t* yr = y + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) *(unsigned*)(yr + c * 4) = pack4_e4m3(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
    if (lane == 0) scale[row] = sc;
}

// ------------------------------------------------------------------ rows of a matrix -> e4m3 + row scale
template <typename T>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const T* __restrict__ w, uint8_t* __restrict__ q,
                                                            float* __restrict__ scale, int rows, int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* wr = w + (size_t)row * ld;
    const int nvec = cols >> 2;
    float amax = 0.f;
    for (int c = lane; c < nvec; c += 64) {
        const float4 v = Vec4<T>::load(wr + c * 4);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / sc;
    uint8_t* qr = q + (size_t)row * cols;
    for (int c = lane; c < nvec; c += 64) {
        const float4 v = Vec4<T>::load(wr + c * 4);
        *(unsigned*)(qr + c * 4) = pack4_e4m3(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
    if (lane == 0) scale[row] = sc;
}

// ------------------------------------------------------------------ the GEMM
struct Fp8Dims { int m, n, k, lda, ldb, tiles_m, tiles_n; };
struct Fp8Epi {
    void* c; const float* sa; const float* sb; const float* bias; int ldc; int act; int c_f32;
};

__device__ __forceinline__ int xcd_remap8(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// 128 rows x 128 BYTES per operand tile: the byte geometry of the bf16 BK = 64 tiles (gemm.hip, dma_tile), so
// the same source-side XOR swizzle applies: chunk c (16 B) of row r sits in slot c ^ (r & 7).
__device__ __forceinline__ void dma_tile8(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int r0, int k0, int wave,
                                          int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i;
        const int row = pc * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (row & 7);
        const unsigned voff = (unsigned)((size_t)(r0 + row) * ld + k0 + ch * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr8)(lds_tile + pc * 1024), 16, voff, 0, 0, 0);
    }
}
// 16 rows x 128 k: lane l holds row (l & 15), the 32 bytes of k-block (l >> 4)
__device__ __forceinline__ i32x8 fp8_frag(const char* lds, int rbase, int lane) {
    const int row = rbase + (lane & 15);
    const int kb = lane >> 4;
    const uint4 lo = *(const uint4*)(lds + row * 128 + (((2 * kb) ^ (row & 7)) << 4));
    const uint4 hi = *(const uint4*)(lds + row * 128 + (((2 * kb + 1) ^ (row & 7)) << 4));
    i32x8 r;
    r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
    r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
    return r;
}

__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                          Fp8Dims d, Fp8Epi ep, unsigned a_bytes, unsigned b_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A 16 KB | B 16 KB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = d.tiles_m * d.tiles_n;
    const int t = xcd_remap8(blockIdx.x, tiles);
    // 16 row tiles per group, row index fastest: the group's A panels and the column's W panel stay in the XCD's L2
    const int per_group = 16 * d.tiles_n;
    const int grp = t / per_group, first_m = grp * 16;
    const int gsz = min(16, d.tiles_m - first_m);
    const int loc = t - grp * per_group;
    const int m0 = (first_m + loc % gsz) * 128, n0 = (loc / gsz) * 128;
    const int nk = d.k >> 7;

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

    f32x4 acc[4][4];          // acc[j][i]: rows n (j tile), cols m (i tile): the MFMA roles are swapped
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    dma_tile8(ra, smem, d.lda, m0, 0, wave, lane);
    dma_tile8(rb, smem + 16384, d.ldb, n0, 0, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int unit = 0x7F7F7F7F;                                  // e8m0 2^0 in every byte: no block scaling
    for (int kt = 0; kt < nk; ++kt) {
        const char* la = smem + (kt & 1) * 32768;
        const char* lb = la + 16384;
        if (kt + 1 < nk) {
            char* na = smem + ((kt + 1) & 1) * 32768;
            dma_tile8(ra, na, d.lda, m0, (kt + 1) * 128, wave, lane);
            dma_tile8(rb, na + 16384, d.ldb, n0, (kt + 1) * 128, wave, lane);
        }
        i32x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = fp8_frag(la, wm * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = fp8_frag(lb, wn * 64 + j * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[j][i], 0, 0, 0, unit, 0, unit);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // D' layout: col = lane & 15 -> m, row = (lane >> 4) * 4 + reg -> n: four consecutive n per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= d.m) continue;
        const float sa = ep.sa[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= d.n) continue;
            const float4 sb = *(const float4*)(ep.sb + n);
            float4 v = make_float4(acc[j][i][0] * (sa * sb.x), acc[j][i][1] * (sa * sb.y), acc[j][i][2] * (sa * sb.z),
                                   acc[j][i][3] * (sa * sb.w));
            if (ep.bias) {
                const float4 b = *(const float4*)(ep.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (ep.act == BF_ACT_QUICK_GELU) {
                v.x = quick_gelu_fast(v.x); v.y = quick_gelu_fast(v.y); v.z = quick_gelu_fast(v.z); v.w = quick_gelu_fast(v.w);
            } else if (ep.act == BF_ACT_TANH_GELU) {
                v.x = tanh_gelu_fast(v.x); v.y = tanh_gelu_fast(v.y); v.z = tanh_gelu_fast(v.z); v.w = tanh_gelu_fast(v.w);
            } else if (ep.act == BF_ACT_ERF_GELU) {
                v.x = erf_gelu_fast(v.x); v.y = erf_gelu_fast(v.y); v.z = erf_gelu_fast(v.z); v.w = erf_gelu_fast(v.w);
            }
            if (ep.c_f32) Vec4<float>::store((float*)ep.c + (size_t)m * ep.ldc + n, v);
            else Vec4<bf16_t>::store((bf16_t*)ep.c + (size_t)m * ep.ldc + n, v);
        }
    }
}

}  // namespace

extern "C" int bf_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_fp8, float* row_scale,
                                    int rows, int cols, float eps, bf_stream_t stream) {
    BF_REQUIRE(x && gamma && beta && y_fp8 && row_scale, BF_EINVAL, "bf_layernorm_fwd_fp8: null pointer");
    BF_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && cols <= 1024, BF_EINVAL,
               "bf_layernorm_fwd_fp8: cols must be a multiple of 4, at most 1024 (got %d)", cols);
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(gamma) && bf_aligned16(beta) && bf_aligned16(y_fp8), BF_EALIGN,
               "bf_layernorm_fwd_fp8: unaligned pointer");
    dim3 grid(bf_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN8(NCV) hipLaunchKernelGGL(ln_fwd_fp8_kernel<NCV>, grid, block, 0, s, x, gamma, beta, (uint8_t*)y_fp8, row_scale, \
                                    rows, cols, eps)
    if (cols <= 256) LN8(1); else if (cols <= 512) LN8(2); else if (cols <= 768) LN8(3); else LN8(4);
#undef LN8
    BF_CHECK_LAUNCH("bf_layernorm_fwd_fp8");
    return BF_OK;
}

extern "C" int bf_quantize_rows_fp8(const void* w, int w_dtype, void* q_fp8, float* row_scale, int rows, int cols, int ld,
                                    bf_stream_t stream) {
    BF_REQUIRE(w && q_fp8 && row_scale && rows > 0 && cols > 0 && (cols & 3) == 0 && ld >= cols && (ld & 3) == 0, BF_EINVAL,
               "bf_quantize_rows_fp8: bad arguments");
    BF_REQUIRE(bf_aligned16(w) && bf_aligned16(q_fp8), BF_EALIGN, "bf_quantize_rows_fp8: unaligned pointer");
    dim3 grid(bf_cdiv(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (w_dtype == BF_F32)
        hipLaunchKernelGGL(quantize_rows_kernel<float>, grid, block, 0, s, (const float*)w, (uint8_t*)q_fp8, row_scale, rows,
                           cols, ld);
    else if (w_dtype == BF_BF16)
        hipLaunchKernelGGL(quantize_rows_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)w, (uint8_t*)q_fp8, row_scale,
                           rows, cols, ld);
    else BF_REQUIRE(false, BF_EINVAL, "bf_quantize_rows_fp8: bad dtype %d", w_dtype);
    BF_CHECK_LAUNCH("bf_quantize_rows_fp8");
    return BF_OK;
}

extern "C" int bf_gemm_fp8(const void* a_fp8, const float* a_scale, const void* b_fp8, const float* b_scale, void* c,
                           int c_dtype, const float* bias, int act, int m, int n, int k, int lda, int ldb, int ldc,
                           bf_stream_t stream) {
    BF_REQUIRE(a_fp8 && a_scale && b_fp8 && b_scale && c, BF_EINVAL, "bf_gemm_fp8: null pointer");
    BF_REQUIRE(m > 0 && n > 0 && k > 0 && (k % 128) == 0, BF_EINVAL, "bf_gemm_fp8: k must be a positive multiple of 128 (got %d)", k);
    BF_REQUIRE((n & 3) == 0 && (ldc & 3) == 0 && ldc >= n && lda >= k && ldb >= k && (lda & 15) == 0 && (ldb & 15) == 0,
               BF_EINVAL, "bf_gemm_fp8: n / ldc must be multiples of 4, lda / ldb multiples of 16");
    BF_REQUIRE(c_dtype == BF_F32 || c_dtype == BF_BF16, BF_EINVAL, "bf_gemm_fp8: bad c_dtype");
    BF_REQUIRE(act == BF_ACT_NONE || act == BF_ACT_QUICK_GELU || act == BF_ACT_ERF_GELU || act == BF_ACT_TANH_GELU,
               BF_EINVAL, "bf_gemm_fp8: activation %d is not available on the forward-only fp8 path", act);
    BF_REQUIRE(bf_aligned16(a_fp8) && bf_aligned16(b_fp8) && bf_aligned16(c) && bf_aligned16(b_scale) &&
                   (!bias || bf_aligned16(bias)), BF_EALIGN, "bf_gemm_fp8: unaligned pointer");
    const size_t a_ext = (size_t)(m - 1) * lda + k, b_ext = (size_t)(n - 1) * ldb + k;
    BF_REQUIRE(a_ext < (1ull << 32) && b_ext < (1ull << 32), BF_EINVAL, "bf_gemm_fp8: operand larger than 4 GiB");
    {   // large-M problems: the persistent 256 x 256 kernel (gemm.hip)
        const int rc = bf_gemm_fp8_p8_try(a_fp8, a_scale, b_fp8, b_scale, c, c_dtype, bias, act, m, n, k, lda, ldb, ldc, stream);
        if (rc <= 0) return rc;
    }
    Fp8Dims d{m, n, k, lda, ldb, bf_cdiv(m, 128), bf_cdiv(n, 128)};
    Fp8Epi ep{c, a_scale, b_scale, bias, ldc, act, c_dtype == BF_F32};
    const long nblk = (long)d.tiles_m * d.tiles_n;
    BF_REQUIRE(nblk < (1L << 31), BF_EINVAL, "bf_gemm_fp8: grid too large");
    static bool lds_ok = false;
    if (!lds_ok) {
        (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        lds_ok = true;
    }
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)nblk), dim3(256), 65536, (hipStream_t)stream, (const uint8_t*)a_fp8,
                       (const uint8_t*)b_fp8, d, ep, (unsigned)a_ext, (unsigned)b_ext);
    BF_CHECK_LAUNCH("bf_gemm_fp8");
    return BF_OK;
}
