// LayerNorm fwd/bwd, column sums, casts and the token/patch plumbing kernels.
// All of these are HBM-bound: one wavefront (64 lanes) per row, 16-byte accesses,
// wave-shuffle reductions, fp32 math.
#include "common.h"
#include "tuning.h"

#include <type_traits>

#define LN_MAXC 8   // float4 chunks per lane kept in registers -> cols <= 2048

// ------------------------------------------------------------------ LayerNorm fwd
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     TY* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int cols,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = cols >> 2;
    const TX* xr = x + (size_t)row * cols;
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] = Vec4<TX>::load(xr + c * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    TY* yr = y + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 g = *(const float4*)(gamma + c * 4);
            const float4 b = *(const float4*)(beta + c * 4);
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + b.x;
            o.y = (v[i].y - mu) * rs * g.y + b.y;
            o.z = (v[i].z - mu) * rs * g.z + b.z;
            o.w = (v[i].w - mu) * rs * g.w + b.w;
            Vec4<TY>::store(yr + c * 4, o);
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// Exact-width variant (cols == NC * 256): no per-chunk bounds tests, and each wave carries R rows
// at once so 2 * NC 16-byte loads are in flight per lane before the first reduction (the one-row
// kernel above is latency-bound at ~3.9 TB/s on the 50 k x 768 encoder activations).
template <typename TX, typename TY, int NC, int R>
__global__ __launch_bounds__(256) void ln_fwd_rows_kernel(const TX* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          TY* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd, int rows, float eps) {
    constexpr int cols = NC * 256;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    float4 v[R][NC];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(row0 + r, rows - 1);
        const TX* xr = x + (size_t)row * cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) v[r][i] = Vec4<TX>::load(xr + (lane + i * 64) * 4);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o, 64);
    }
    float mu[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = s[r] / (float)cols;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float a = v[r][i].x - mu[r], b = v[r][i].y - mu[r], c = v[r][i].z - mu[r],
                        d = v[r][i].w - mu[r];
            q[r] += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o, 64);
    }
    float4 g[NC], bt[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        g[i] = *(const float4*)(gamma + (lane + i * 64) * 4);
        bt[i] = *(const float4*)(beta + (lane + i * 64) * 4);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= rows) break;
        const float rs = rsqrtf(q[r] / (float)cols + eps);
        TY* yr = y + (size_t)(row0 + r) * cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float4 o;
            o.x = (v[r][i].x - mu[r]) * rs * g[i].x + bt[i].x;
            o.y = (v[r][i].y - mu[r]) * rs * g[i].y + bt[i].y;
            o.z = (v[r][i].z - mu[r]) * rs * g[i].z + bt[i].z;
            o.w = (v[r][i].w - mu[r]) * rs * g[i].w + bt[i].w;
            Vec4<TY>::store(yr + (lane + i * 64) * 4, o);
        }
        if (lane == 0) {
            if (mean) mean[row0 + r] = mu[r];
            if (rstd) rstd[row0 + r] = rs;
        }
    }
}

extern "C" int bf_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                                void* y, int y_dtype, float* mean, float* rstd, int rows, int cols,
                                float eps, bf_stream_t stream) {
    BF_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && cols <= LN_MAXC * 256, BF_EINVAL,
               "bf_layernorm_fwd: need cols %% 4 == 0 and cols <= %d (got rows=%d cols=%d)",
               LN_MAXC * 256, rows, cols);
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(y) && bf_aligned16(gamma) && bf_aligned16(beta),
               BF_EALIGN, "bf_layernorm_fwd: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bf_cdiv(rows, 4)), block(256);
    const int ln_rows = bf_tuning().ln_rows;
    if (x_dtype == BF_F32 && (cols == 512 || cols == 768 || cols == 1024) && ln_rows > 1 &&
        (bf_is_half(y_dtype) || y_dtype == BF_F32)) {
#define LN_ROWS(TY, NC, R)                                                                       \
    hipLaunchKernelGGL((ln_fwd_rows_kernel<float, TY, NC, R>), dim3(bf_cdiv(rows, 4 * R)), block, 0, \
                       s, (const float*)x, gamma, beta, (TY*)y, mean, rstd, rows, eps)
#define LN_ROWS_NC(TY, R)                                                                       \
    do {                                                                                        \
        if (cols == 512) LN_ROWS(TY, 2, R); else if (cols == 768) LN_ROWS(TY, 3, R);            \
        else LN_ROWS(TY, 4, R);                                                                 \
    } while (0)
        if (y_dtype == BF_BF16) { if (ln_rows >= 4) LN_ROWS_NC(bf16_t, 4); else LN_ROWS_NC(bf16_t, 2); }
        else if (y_dtype == BF_F16) { if (ln_rows >= 4) LN_ROWS_NC(f16_t, 4); else LN_ROWS_NC(f16_t, 2); }
        else { if (ln_rows >= 4) LN_ROWS_NC(float, 4); else LN_ROWS_NC(float, 2); }
#undef LN_ROWS_NC
#undef LN_ROWS
        BF_CHECK_LAUNCH("bf_layernorm_fwd");
        return BF_OK;
    }
#define LN_LAUNCH(TX, TY)                                                                       \
    hipLaunchKernelGGL((ln_fwd_kernel<TX, TY>), grid, block, 0, s, (const TX*)x, gamma, beta,  \
                       (TY*)y, mean, rstd, rows, cols, eps)
    if (x_dtype == BF_F32 && y_dtype == BF_F32) LN_LAUNCH(float, float);
    else if (x_dtype == BF_F32 && y_dtype == BF_BF16) LN_LAUNCH(float, bf16_t);
    else if (x_dtype == BF_BF16 && y_dtype == BF_BF16) LN_LAUNCH(bf16_t, bf16_t);
    else if (x_dtype == BF_BF16 && y_dtype == BF_F32) LN_LAUNCH(bf16_t, float);
    else if (x_dtype == BF_F32 && y_dtype == BF_F16) LN_LAUNCH(float, f16_t);
    else if (x_dtype == BF_F16 && y_dtype == BF_F16) LN_LAUNCH(f16_t, f16_t);
    else if (x_dtype == BF_F16 && y_dtype == BF_F32) LN_LAUNCH(f16_t, float);
    else BF_REQUIRE(false, BF_EINVAL, "bf_layernorm_fwd: bad dtype codes %d/%d", x_dtype, y_dtype);
#undef LN_LAUNCH
    BF_CHECK_LAUNCH("bf_layernorm_fwd");
    return BF_OK;
}

// ------------------------------------------------------------------ residual add + LayerNorm
// x_out = x + delta ; y = LN(x_out).  `delta` is the bf16 output of the branch's last GEMM (out-proj / MLP fc2, bias
// included): with the add done HERE the GEMM runs its plain-bf16 epilogue (no 256 KB residual tile to wait for at item
// start, half the store bytes) and this kernel, which reads the residual stream anyway, adds 2 B/element of traffic.
// Two rows per wave like ln_fwd_rows_kernel; cols == NC * 256.
template <typename TY, int NC, typename TDL = bf16_t, bool HAS2 = false>
__global__ __launch_bounds__(256) void add_ln_rows_kernel(const float* x, const TDL* __restrict__ delta,
                                                          const TDL* __restrict__ delta2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* x_out, TY* __restrict__ y,      // x_out may BE x (the in-place stream update): no restrict on the pair
                                                          float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                          float eps) {
    constexpr int cols = NC * 256;
    constexpr int R = 2;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    float4 v[R][NC];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(row0 + r, rows - 1);
        const float* xr = x + (size_t)row * cols;
        const TDL* dr = delta + (size_t)row * cols;
        // second branch (the frozen towers' lazily written stream, bf_add2_layernorm_fwd): (x + delta) + delta2, the
        // order in which two separate passes would have added them; all three loads of a row go out together
        const TDL* dr2 = HAS2 ? delta2 + (size_t)row * cols : nullptr;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float4 a = *(const float4*)(xr + (lane + i * 64) * 4);
            const float4 d = Vec4<TDL>::load(dr + (lane + i * 64) * 4);
            if constexpr (HAS2) {
                const float4 e = Vec4<TDL>::load(dr2 + (lane + i * 64) * 4);
                v[r][i] = make_float4((a.x + d.x) + e.x, (a.y + d.y) + e.y, (a.z + d.z) + e.z, (a.w + d.w) + e.w);
            } else {
                v[r][i] = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o, 64);
    }
    float mu[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = s[r] / (float)cols;
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float a = v[r][i].x - mu[r], b = v[r][i].y - mu[r], c = v[r][i].z - mu[r], d = v[r][i].w - mu[r];
            q[r] += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o, 64);
    }
    float4 g[NC], bt[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        g[i] = *(const float4*)(gamma + (lane + i * 64) * 4);
        bt[i] = *(const float4*)(beta + (lane + i * 64) * 4);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= rows) break;
        const float rs = rsqrtf(q[r] / (float)cols + eps);
        float* xo = x_out ? x_out + (size_t)(row0 + r) * cols : nullptr;      // no x_out: only y = LN(x + delta) is wanted
        TY* yr = y + (size_t)(row0 + r) * cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (xo) *(float4*)(xo + (lane + i * 64) * 4) = v[r][i];
            float4 o;
            o.x = (v[r][i].x - mu[r]) * rs * g[i].x + bt[i].x;
            o.y = (v[r][i].y - mu[r]) * rs * g[i].y + bt[i].y;
            o.z = (v[r][i].z - mu[r]) * rs * g[i].z + bt[i].z;
            o.w = (v[r][i].w - mu[r]) * rs * g[i].w + bt[i].w;
            Vec4<TY>::store(yr + (lane + i * 64) * 4, o);
        }
        if (lane == 0) {
            if (mean) mean[row0 + r] = mu[r];
            if (rstd) rstd[row0 + r] = rs;
        }
    }
}

extern "C" int bf_add_layernorm_fwd(const float* x, const void* delta, int delta_dtype, const float* gamma, const float* beta,
                                    float* x_out, void* y, int y_dtype, float* mean, float* rstd, int rows, int cols,
                                    float eps, bf_stream_t stream) {
    BF_REQUIRE(x_out, BF_EINVAL, "bf_add_layernorm_fwd: null pointer");
    return bf_add2_layernorm_fwd(x, delta, nullptr, delta_dtype, gamma, beta, x_out, y, y_dtype, mean, rstd, rows, cols, eps,
                                 stream);
}

extern "C" int bf_add2_layernorm_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype,
                                     const float* gamma, const float* beta, float* x_out, void* y, int y_dtype,
                                     float* mean, float* rstd, int rows, int cols, float eps, bf_stream_t stream) {
    BF_REQUIRE(x && delta && gamma && beta && y, BF_EINVAL, "bf_add_layernorm_fwd: null pointer");
    BF_REQUIRE(!delta2 || bf_aligned16(delta2), BF_EALIGN, "bf_add2_layernorm_fwd: unaligned delta2");
    BF_REQUIRE(rows > 0 && (cols == 256 || cols == 512 || cols == 768 || cols == 1024), BF_EINVAL,
               "bf_add_layernorm_fwd: cols must be 256, 512, 768 or 1024 (got %d)", cols);
    BF_REQUIRE(bf_is_half(y_dtype) || y_dtype == BF_F32, BF_EINVAL, "bf_add_layernorm_fwd: bad y dtype");
    BF_REQUIRE(bf_is_half(delta_dtype) && (y_dtype == BF_F32 || y_dtype == delta_dtype), BF_EINVAL,
               "bf_add_layernorm_fwd: delta is a 16-bit tensor and a 16-bit y has its type (delta %d, y %d)", delta_dtype, y_dtype);
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(delta) && (!x_out || bf_aligned16(x_out)) && bf_aligned16(y) &&
                   bf_aligned16(gamma) && bf_aligned16(beta), BF_EALIGN, "bf_add_layernorm_fwd: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bf_cdiv(rows, 8)), block(256);
#define ADDLN2(TY, NCV, TDL, H2) hipLaunchKernelGGL((add_ln_rows_kernel<TY, NCV, TDL, H2>), grid, block, 0, s, x, (const TDL*)delta, \
                                                   (const TDL*)delta2, gamma, beta, x_out, (TY*)y, mean, rstd, rows, eps)
#define ADDLN(TY, NCV, TDL) do { if (delta2) ADDLN2(TY, NCV, TDL, true); else ADDLN2(TY, NCV, TDL, false); } while (0)
#define ADDLN_NC(TY, TDL)                                                              \
    do {                                                                               \
        if (cols == 256) ADDLN(TY, 1, TDL); else if (cols == 512) ADDLN(TY, 2, TDL);   \
        else if (cols == 768) ADDLN(TY, 3, TDL); else ADDLN(TY, 4, TDL);               \
    } while (0)
    if (y_dtype == BF_BF16) ADDLN_NC(bf16_t, bf16_t);
    else if (y_dtype == BF_F16) ADDLN_NC(f16_t, f16_t);
    else if (delta_dtype == BF_F16) ADDLN_NC(float, f16_t);
    else ADDLN_NC(float, bf16_t);
#undef ADDLN_NC
#undef ADDLN
#undef ADDLN2
    BF_CHECK_LAUNCH("bf_add_layernorm_fwd");
    return BF_OK;
}

// ------------------------------------------------------------------ LayerNorm bwd
#define LN_BWD_ROWS_PER_WAVE 8
extern "C" int bf_layernorm_bwd_parts(int rows) {
    int p = bf_cdiv(rows, 4 * LN_BWD_ROWS_PER_WAVE);
    return p < 1 ? 1 : (p > 1024 ? 1024 : p);
}

// NC = float4 chunks per lane (cols <= NC*256): sized per call so that the five per-lane
// accumulator arrays do not cost occupancy at the common widths (512 -> 2, 768 -> 3).
// R16 (the 16-bit gradient stream, fp16 mode): the residual gradient arrives as a TC tensor (`dres` is then a TC*) and
// leaves ONLY as dx_cdt -- no fp32 copy is read or written: 360 MB instead of 576 per call at the fusion's shape.
template <typename TDY, typename TC, int NC, bool R16 = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dres,
    float* __restrict__ dx, TC* __restrict__ dx_cdt, float* __restrict__ dgamma_part,
    float* __restrict__ dbeta_part, float* __restrict__ dres_sum_part, float* __restrict__ dx_sum_part,
    int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [2][4][cols]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = cols >> 2;
    float4 dg[NC], db[NC], gm[NC], sr[NC], sx[NC];
    const bool want_sums = dres_sum_part != nullptr;     // bias gradients of the two residual GEMMs
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        dg[i] = make_float4(0, 0, 0, 0);
        db[i] = make_float4(0, 0, 0, 0);
        sr[i] = make_float4(0, 0, 0, 0);
        sx[i] = make_float4(0, 0, 0, 0);
        const int c = lane + i * 64;
        gm[i] = (c < nvec) ? *(const float4*)(gamma + c * 4) : make_float4(0, 0, 0, 0);
    }
    const float inv_cols = 1.0f / (float)cols;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const size_t off = (size_t)row * cols;
        float4 xh[NC], g[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                const float4 xv = *(const float4*)(x + off + c * 4);
                const float4 d = Vec4<TDY>::load(dy + off + c * 4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs,
                                    (xv.w - mu) * rs);
                g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y;
                dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
                db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) * inv_cols;
        const float m2 = wave_sum(s2) * inv_cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                float4 o;
                o.x = rs * (g[i].x - m1 - xh[i].x * m2);
                o.y = rs * (g[i].y - m1 - xh[i].y * m2);
                o.z = rs * (g[i].z - m1 - xh[i].z * m2);
                o.w = rs * (g[i].w - m1 - xh[i].w * m2);
                if (dres) {
                    float4 r;
                    if constexpr (R16) r = Vec4<TC>::load((const TC*)dres + off + c * 4);
                    else r = *(const float4*)(dres + off + c * 4);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                    if (want_sums) { sr[i].x += r.x; sr[i].y += r.y; sr[i].z += r.z; sr[i].w += r.w; }
                }
                if (want_sums) { sx[i].x += o.x; sx[i].y += o.y; sx[i].z += o.z; sx[i].w += o.w; }
                if constexpr (!R16) *(float4*)(dx + off + c * 4) = o;
                if (dx_cdt) Vec4<TC>::store(dx_cdt + off + c * 4, o);
            }
        }
    }
    // combine the 4 waves of the block, write one partial row per block
    float* rg = red;
    float* rb = red + 4 * cols;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            *(float4*)(rg + wave * cols + c * 4) = dg[i];
            *(float4*)(rb + wave * cols + c * 4) = db[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        dgamma_part[(size_t)blockIdx.x * cols + c] =
            (rg[c] + rg[cols + c]) + (rg[2 * cols + c] + rg[3 * cols + c]);
        dbeta_part[(size_t)blockIdx.x * cols + c] =
            (rb[c] + rb[cols + c]) + (rb[2 * cols + c] + rb[3 * cols + c]);
    }
    if (want_sums) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                *(float4*)(rg + wave * cols + c * 4) = sr[i];
                *(float4*)(rb + wave * cols + c * 4) = sx[i];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) {
            dres_sum_part[(size_t)blockIdx.x * cols + c] =
                (rg[c] + rg[cols + c]) + (rg[2 * cols + c] + rg[3 * cols + c]);
            dx_sum_part[(size_t)blockIdx.x * cols + c] =
                (rb[c] + rb[cols + c]) + (rb[2 * cols + c] + rb[3 * cols + c]);
        }
    }
}

// The 16-bit gradient stream's backward (R16 above) with EIGHT columns per lane: the three 16-bit tensors of a row (dy, the
// incoming residual gradient, the outgoing one) then move as 16-byte pieces like the fp32 x does -- with four columns per
// lane they were 8-byte pieces, half-width transactions, and the kernel sat at 4.4 TB/s where its fp32 sibling reaches 6.1.
// Same arithmetic, same per-column partial sums (the order of a row's two wave reductions differs from the 4-column form
// only in which lane holds which column).
template <typename TC, int NC8>
__global__ __launch_bounds__(256) void ln_bwd16_kernel(
    const TC* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const TC* __restrict__ dres,
    TC* __restrict__ dx_cdt, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
    float* __restrict__ dres_sum_part, float* __restrict__ dx_sum_part, int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [2][4][cols]
    typedef typename H16<TC>::x8 h8;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nv8 = cols >> 3;
    float dg[NC8][8], db[NC8][8], gm[NC8][8], sr[NC8][8], sx[NC8][8];
    const bool want_sums = dres_sum_part != nullptr;
#pragma unroll
    for (int i = 0; i < NC8; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dg[i][e] = 0.f; db[i][e] = 0.f; sr[i][e] = 0.f; sx[i][e] = 0.f;
            gm[i][e] = c < nv8 ? gamma[c * 8 + e] : 0.f;
        }
    }
    const float inv_cols = 1.0f / (float)cols;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const size_t off = (size_t)row * cols;
        float xh[NC8][8], g[NC8][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC8; ++i) {
            const int c = lane + i * 64;
            if (c < nv8) {
                const float4 x0 = *(const float4*)(x + off + c * 8), x1 = *(const float4*)(x + off + c * 8 + 4);
                const h8 d8 = *(const h8*)(dy + off + c * 8);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)d8[e];
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = d * gm[i][e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    dg[i][e] += d * xh[i][e];
                    db[i][e] += d;
                }
            }
        }
        const float m1 = wave_sum(s1) * inv_cols;
        const float m2 = wave_sum(s2) * inv_cols;
#pragma unroll
        for (int i = 0; i < NC8; ++i) {
            const int c = lane + i * 64;
            if (c < nv8) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] - m1 - xh[i][e] * m2);
                if (dres) {
                    const h8 r8 = *(const h8*)(dres + off + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float r = (float)r8[e];
                        o[e] += r;
                        if (want_sums) sr[i][e] += r;
                    }
                }
                h8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (want_sums) sx[i][e] += o[e];
                    o8[e] = (TC)o[e];
                }
                *(h8*)(dx_cdt + off + c * 8) = o8;
            }
        }
    }
    float* rg = red;
    float* rb = red + 4 * cols;
    auto put = [&](float* dst, const float (*v)[8]) {
#pragma unroll
        for (int i = 0; i < NC8; ++i) {
            const int c = lane + i * 64;
            if (c < nv8) {
                *(float4*)(dst + wave * cols + c * 8) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                *(float4*)(dst + wave * cols + c * 8 + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
            }
        }
    };
    put(rg, dg); put(rb, db);
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        dgamma_part[(size_t)blockIdx.x * cols + c] = (rg[c] + rg[cols + c]) + (rg[2 * cols + c] + rg[3 * cols + c]);
        dbeta_part[(size_t)blockIdx.x * cols + c] = (rb[c] + rb[cols + c]) + (rb[2 * cols + c] + rb[3 * cols + c]);
    }
    if (want_sums) {
        __syncthreads();
        put(rg, sr); put(rb, sx);
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) {
            dres_sum_part[(size_t)blockIdx.x * cols + c] = (rg[c] + rg[cols + c]) + (rg[2 * cols + c] + rg[3 * cols + c]);
            dx_sum_part[(size_t)blockIdx.x * cols + c] = (rb[c] + rb[cols + c]) + (rb[2 * cols + c] + rb[3 * cols + c]);
        }
    }
}

extern "C" int bf_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_cdt, int cdt_dtype, float* dgamma_part, float* dbeta_part,
                                float* dres_sum_part, float* dx_sum_part, int rows, int cols,
                                bf_stream_t stream) {
    BF_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && cols <= LN_MAXC * 256, BF_EINVAL,
               "bf_layernorm_bwd: need cols %% 4 == 0 and cols <= %d (got rows=%d cols=%d)",
               LN_MAXC * 256, rows, cols);
    BF_REQUIRE((dres_sum_part == nullptr) == (dx_sum_part == nullptr) && (!dres_sum_part || dres), BF_EINVAL,
               "bf_layernorm_bwd: dres_sum_part and dx_sum_part come together and need dres");
    BF_REQUIRE(bf_aligned16(dy) && bf_aligned16(x) && (!dx || bf_aligned16(dx)) && bf_aligned16(gamma) &&
                   (!dres || bf_aligned16(dres)) && (!dx_cdt || bf_aligned16(dx_cdt)),
               BF_EALIGN, "bf_layernorm_bwd: pointers must be 16-byte aligned");
    // dx == NULL: the 16-bit gradient stream -- dres (if any) is a cdt_dtype tensor, only dx_cdt is written
    const bool r16 = dx == nullptr;
    BF_REQUIRE(!r16 || (dx_cdt && cdt_dtype != BF_F32 && dy_dtype == cdt_dtype), BF_EINVAL,
               "bf_layernorm_bwd: without an fp32 dx the call needs a 16-bit dx_cdt of dy's type");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bf_layernorm_bwd_parts(rows)), block(256);
    const size_t lds = (size_t)8 * cols * sizeof(float);
#define LNB_LAUNCH_NC(TDY, TC, NCV)                                                               \
    do {                                                                                          \
        if (r16 && sizeof(TC) == 2 && sizeof(TDY) == 2)                                           \
            hipLaunchKernelGGL((ln_bwd_kernel<TDY, TC, NCV, (sizeof(TC) == 2 && sizeof(TDY) == 2)>), grid, block, lds, s, \
                               (const TDY*)dy, x, gamma, mean, rstd, dres, dx, (TC*)dx_cdt, dgamma_part, dbeta_part, \
                               dres_sum_part, dx_sum_part, rows, cols);                           \
        else                                                                                      \
            hipLaunchKernelGGL((ln_bwd_kernel<TDY, TC, NCV>), grid, block, lds, s, (const TDY*)dy, x, gamma, \
                               mean, rstd, dres, dx, (TC*)dx_cdt, dgamma_part, dbeta_part, dres_sum_part,   \
                               dx_sum_part, rows, cols);                                          \
    } while (0)
#define LNB_LAUNCH(TDY, TC)                                                                       \
    do {                                                                                          \
        if (cols <= 512) LNB_LAUNCH_NC(TDY, TC, 2);                                               \
        else if (cols <= 768) LNB_LAUNCH_NC(TDY, TC, 3);                                          \
        else if (cols <= 1024) LNB_LAUNCH_NC(TDY, TC, 4);                                         \
        else LNB_LAUNCH_NC(TDY, TC, 8);                                                           \
    } while (0)
    const int cd = dx_cdt ? cdt_dtype : BF_F32;
    if (r16 && (cols & 7) == 0 && cols <= 1024 && bf_tuning().ln_bwd16_wide) {      // eight columns per lane (above)
#define LNB16(TC, N8)                                                                             \
    hipLaunchKernelGGL((ln_bwd16_kernel<TC, N8>), grid, block, lds, s, (const TC*)dy, x, gamma, mean, rstd, \
                       (const TC*)dres, (TC*)dx_cdt, dgamma_part, dbeta_part, dres_sum_part, dx_sum_part, rows, cols)
        if (cd == BF_F16) { if (cols <= 512) LNB16(f16_t, 1); else LNB16(f16_t, 2); }
        else { if (cols <= 512) LNB16(bf16_t, 1); else LNB16(bf16_t, 2); }
#undef LNB16
        BF_CHECK_LAUNCH("bf_layernorm_bwd");
        return BF_OK;
    }
    if (dy_dtype == BF_F32 && cd == BF_F32) LNB_LAUNCH(float, float);
    else if (dy_dtype == BF_F32 && cd == BF_BF16) LNB_LAUNCH(float, bf16_t);
    else if (dy_dtype == BF_BF16 && cd == BF_BF16) LNB_LAUNCH(bf16_t, bf16_t);
    else if (dy_dtype == BF_BF16 && cd == BF_F32) LNB_LAUNCH(bf16_t, float);
    else if (dy_dtype == BF_F32 && cd == BF_F16) LNB_LAUNCH(float, f16_t);
    else if (dy_dtype == BF_F16 && cd == BF_F16) LNB_LAUNCH(f16_t, f16_t);
    else if (dy_dtype == BF_F16 && cd == BF_F32) LNB_LAUNCH(f16_t, float);
    else BF_REQUIRE(false, BF_EINVAL, "bf_layernorm_bwd: bad dtype codes");
#undef LNB_LAUNCH
#undef LNB_LAUNCH_NC
    BF_CHECK_LAUNCH("bf_layernorm_bwd");
    return BF_OK;
}

// ------------------------------------------------------------------ column sums
// grid (col tiles of 256, row chunks); block 256 = 64 float4 column groups x 4 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x,
                                                     float* __restrict__ out, int rows, int cols,
                                                     int ld, int rows_per_block) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const bool live = c4 * 4 < cols;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float4 acc = make_float4(0, 0, 0, 0);
    if (live) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const float4 v = Vec4<T>::load(x + (size_t)r * ld + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
        const float4 a = red[0][lane], b = red[1][lane], c = red[2][lane], d = red[3][lane];
        float* o = out + c4 * 4;
        atomicAdd(o + 0, (a.x + b.x) + (c.x + d.x));
        atomicAdd(o + 1, (a.y + b.y) + (c.y + d.y));
        atomicAdd(o + 2, (a.z + b.z) + (c.z + d.z));
        atomicAdd(o + 3, (a.w + b.w) + (c.w + d.w));
    }
}

extern "C" int bf_colsum(const void* x, int x_dtype, float* out, int rows, int cols, int ld,
                         int accumulate, bf_stream_t stream) {
    BF_REQUIRE(rows > 0 && cols > 0 && (cols & 3) == 0 && (ld & 3) == 0 && ld >= cols, BF_EINVAL,
               "bf_colsum: need cols %% 4 == 0, ld %% 4 == 0 (rows=%d cols=%d ld=%d)", rows, cols, ld);
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(out), BF_EALIGN, "bf_colsum: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = bf_zero_f32_async(out, (size_t)cols, s);
        BF_REQUIRE(e == hipSuccess, BF_ELAUNCH, "bf_colsum: zeroing the sums failed");
    }
    const int col_tiles = bf_cdiv(cols, 256);
    int row_chunks = bf_cdiv(2048, col_tiles);
    int rpb = bf_cdiv(rows, row_chunks);
    if (rpb < 16) rpb = 16;
    row_chunks = bf_cdiv(rows, rpb);
    dim3 grid(col_tiles, row_chunks), block(256);
    if (x_dtype == BF_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)x, out, rows, cols, ld, rpb);
    else if (x_dtype == BF_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, out, rows, cols, ld, rpb);
    else if (x_dtype == BF_F16)
        hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, out, rows, cols, ld, rpb);
    else BF_REQUIRE(false, BF_EINVAL, "bf_colsum: bad dtype %d", x_dtype);
    BF_CHECK_LAUNCH("bf_colsum");
    return BF_OK;
}

// Column sums of up to four stacked partial arrays [n_arrays][nparts][cols] in ONE launch, written
// (not accumulated) to separate outputs: LayerNorm backward's dgamma / dbeta / dres-sum / dx-sum
// partials cost two to four bf_colsum launches plus their memsets otherwise (eight ~5-10 us launches).
// Block = 16 columns x 64 row groups (rows strided over the groups, LDS tree at the end): with 64-column blocks a
// 512-wide problem had 8 x n_arrays workgroups on 256 CUs, each thread adding 64 dependent loads -- 21 us of
// latency per call, sixteen calls per step.
__global__ __launch_bounds__(1024) void colsum_multi_kernel(const float* __restrict__ parts, int nparts, int cols,
                                                            float* __restrict__ o0, float* __restrict__ o1,
                                                            float* __restrict__ o2, float* __restrict__ o3) {
    __shared__ float red[64][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, a = blockIdx.y;
    const int c = blockIdx.x * 16 + cl;
    const float* src = parts + (size_t)a * nparts * cols;
    float acc0 = 0.f, acc1 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 64 < nparts; r += 128) {
            acc0 += src[(size_t)r * cols + c];
            acc1 += src[(size_t)(r + 64) * cols + c];
        }
        if (r < nparts) acc0 += src[(size_t)r * cols + c];
    }
    red[rg][cl] = acc0 + acc1;
    __syncthreads();
    if (threadIdx.x < 16 && c < cols) {
        float t = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) t += red[i][threadIdx.x];
        float* out = a == 0 ? o0 : a == 1 ? o1 : a == 2 ? o2 : o3;
        out[c] = t;
    }
}

extern "C" int bf_colsum_multi(const float* parts, int n_arrays, int nparts, int cols, float* out0, float* out1,
                               float* out2, float* out3, bf_stream_t stream) {
    BF_REQUIRE(parts && n_arrays >= 1 && n_arrays <= 4 && nparts > 0 && cols > 0, BF_EINVAL,
               "bf_colsum_multi: bad arguments (n_arrays=%d nparts=%d cols=%d)", n_arrays, nparts, cols);
    float* outs[4] = {out0, out1, out2, out3};
    for (int i = 0; i < n_arrays; ++i) BF_REQUIRE(outs[i] != nullptr, BF_EINVAL, "bf_colsum_multi: out%d is NULL", i);
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(bf_cdiv(cols, 16), n_arrays), dim3(1024), 0, (hipStream_t)stream,
                       parts, nparts, cols, out0, out1, out2, out3);
    BF_CHECK_LAUNCH("bf_colsum_multi");
    return BF_OK;
}

// ------------------------------------------------------------------ cast
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst,
                                                   size_t n4, size_t count) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        Vec4<TD>::store(dst + i * 4, Vec4<TS>::load(src + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (count & 3))
        dst[n4 * 4 + threadIdx.x] = from_f32<TD>(to_f32<TS>(src[n4 * 4 + threadIdx.x]));
}

extern "C" int bf_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t count,
                       bf_stream_t stream) {
    BF_REQUIRE(count > 0, BF_EINVAL, "bf_cast: count must be > 0");
    BF_REQUIRE(bf_aligned16(src) && bf_aligned16(dst), BF_EALIGN, "bf_cast: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(256);
    if (src_dtype == BF_F32 && dst_dtype == BF_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, s, (const float*)src, (bf16_t*)dst, n4, count);
    else if (src_dtype == BF_BF16 && dst_dtype == BF_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)src, (float*)dst, n4, count);
    else if (src_dtype == BF_F32 && dst_dtype == BF_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, s, (const float*)src, (float*)dst, n4, count);
    else if (src_dtype == BF_BF16 && dst_dtype == BF_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, n4, count);
    else if (src_dtype == BF_F32 && dst_dtype == BF_F16)
        hipLaunchKernelGGL((cast_kernel<float, f16_t>), grid, block, 0, s, (const float*)src, (f16_t*)dst, n4, count);
    else if (src_dtype == BF_F16 && dst_dtype == BF_F32)
        hipLaunchKernelGGL((cast_kernel<f16_t, float>), grid, block, 0, s, (const f16_t*)src, (float*)dst, n4, count);
    else if (src_dtype == BF_F16 && dst_dtype == BF_F16)
        hipLaunchKernelGGL((cast_kernel<f16_t, f16_t>), grid, block, 0, s, (const f16_t*)src, (f16_t*)dst, n4, count);
    else if (src_dtype == BF_BF16 && dst_dtype == BF_F16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, f16_t>), grid, block, 0, s, (const bf16_t*)src, (f16_t*)dst, n4, count);
    else if (src_dtype == BF_F16 && dst_dtype == BF_BF16)
        hipLaunchKernelGGL((cast_kernel<f16_t, bf16_t>), grid, block, 0, s, (const f16_t*)src, (bf16_t*)dst, n4, count);
    else BF_REQUIRE(false, BF_EINVAL, "bf_cast: bad dtype codes");
    BF_CHECK_LAUNCH("bf_cast");
    return BF_OK;
}

// ------------------------------------------------------------------ patchify (im2col)
// One thread per 4 consecutive kx of one (patch, c, ky): reads 16 B of the image row,
// writes 4 elements of the column matrix.  Consecutive threads walk kx, ky, c, patch.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       T* __restrict__ cols, int batch, int size,
                                                       int patch) {
    const int g = size / patch;
    const int kq = patch >> 2;                          // float4 per kernel row
    const size_t per_patch = (size_t)3 * patch * kq;
    const size_t total = (size_t)batch * g * g * per_patch;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pidx = i / per_patch;
        int r = (int)(i - pidx * per_patch);
        const int c = r / (patch * kq);
        r -= c * patch * kq;
        const int ky = r / kq, kx4 = r - ky * kq;
        const int b = (int)(pidx / (g * g));
        const int pp = (int)(pidx - (size_t)b * g * g);
        const int py = pp / g, px = pp - py * g;
        const float4 v = *(const float4*)(img + (((size_t)b * 3 + c) * size + (py * patch + ky)) * size +
                                          px * patch + kx4 * 4);
        Vec4<T>::store(cols + pidx * ((size_t)3 * patch * patch) + ((size_t)c * patch + ky) * patch + kx4 * 4, v);
    }
}

// Any even patch size (ViT-L/14: 14-pixel rows are only 8-byte aligned) and a padded row length:
// one thread per 2 consecutive kx; columns [3 p^2, ld) are written as zeros so that the GEMM's K
// can be a multiple of its tile (588 -> 640 for patch 14).
template <typename T>
__global__ __launch_bounds__(256) void patchify_ld_kernel(const float* __restrict__ img, T* __restrict__ cols,
                                                          int batch, int size, int patch, int ld) {
    const int g = size / patch;
    const int half = ld >> 1;                           // pairs per output row
    const int kreal = 3 * patch * patch;
    const size_t total = (size_t)batch * g * g * half;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pidx = i / half;
        const int col = (int)(i - pidx * half) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (col < kreal) {
            const int c = col / (patch * patch);
            const int r = col - c * patch * patch;
            const int ky = r / patch, kx = r - ky * patch;
            const int b = (int)(pidx / (g * g));
            const int pp = (int)(pidx - (size_t)b * g * g);
            const int py = pp / g, px = pp - py * g;
            v = *(const float2*)(img + (((size_t)b * 3 + c) * size + (py * patch + ky)) * size + px * patch + kx);
        }
        T* dst = cols + pidx * (size_t)ld + col;
        dst[0] = (T)v.x;
        dst[1] = (T)v.y;
    }
}

extern "C" int bf_patchify_ld(const float* img, void* cols, int cdt, int batch, int size, int patch, int ld,
                              bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && patch > 0 && (patch & 1) == 0 && size % patch == 0, BF_EINVAL,
               "bf_patchify_ld: need an even patch and size %% patch == 0 (size=%d patch=%d)", size, patch);
    BF_REQUIRE(ld >= 3 * patch * patch && (ld & 7) == 0, BF_EINVAL,
               "bf_patchify_ld: ld must be a multiple of 8 and >= 3 patch^2 (ld=%d)", ld);
    BF_REQUIRE(bf_aligned16(img) && bf_aligned16(cols), BF_EALIGN, "bf_patchify_ld: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)batch * (size / patch) * (size / patch) * (ld / 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    dim3 grid((unsigned)blocks), block(256);
    if (cdt == BF_F32)
        hipLaunchKernelGGL(patchify_ld_kernel<float>, grid, block, 0, s, img, (float*)cols, batch, size, patch, ld);
    else if (cdt == BF_BF16)
        hipLaunchKernelGGL(patchify_ld_kernel<bf16_t>, grid, block, 0, s, img, (bf16_t*)cols, batch, size, patch, ld);
    else if (cdt == BF_F16)
        hipLaunchKernelGGL(patchify_ld_kernel<f16_t>, grid, block, 0, s, img, (f16_t*)cols, batch, size, patch, ld);
    else BF_REQUIRE(false, BF_EINVAL, "bf_patchify_ld: bad dtype %d", cdt);
    BF_CHECK_LAUNCH("bf_patchify_ld");
    return BF_OK;
}

extern "C" int bf_patchify(const float* img, void* cols, int cdt, int batch, int size, int patch,
                           bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && patch > 0 && (patch & 3) == 0 && size % patch == 0, BF_EINVAL,
               "bf_patchify: need patch %% 4 == 0 and size %% patch == 0 (size=%d patch=%d)", size, patch);
    BF_REQUIRE(bf_aligned16(img) && bf_aligned16(cols), BF_EALIGN, "bf_patchify: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)batch * 3 * size * size / 4;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    if (cdt == BF_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, s, img, (float*)cols, batch, size, patch);
    else if (cdt == BF_BF16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, block, 0, s, img, (bf16_t*)cols, batch, size, patch);
    else if (cdt == BF_F16)
        hipLaunchKernelGGL(patchify_kernel<f16_t>, grid, block, 0, s, img, (f16_t*)cols, batch, size, patch);
    else BF_REQUIRE(false, BF_EINVAL, "bf_patchify: bad dtype %d", cdt);
    BF_CHECK_LAUNCH("bf_patchify");
    return BF_OK;
}

// ------------------------------------------------------------------ vision assemble + ln_pre
__global__ __launch_bounds__(256) void vision_assemble_kernel(
    const float* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ x,
    int batch, int n_patch, int width, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int L = n_patch + 1;
    if (row >= batch * L) return;
    const int b = row / L, t = row - b * L;
    const float* src = (t == 0) ? cls : patches + ((size_t)b * n_patch + (t - 1)) * width;
    const float* pr = pos + (size_t)t * width;
    const int nvec = width >> 2;
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 a = *(const float4*)(src + c * 4);
            const float4 p = *(const float4*)(pr + c * 4);
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mu = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float a = v[i].x - mu, bb = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)width + eps);
    float* xr = x + (size_t)row * width;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 g = *(const float4*)(gamma + c * 4);
            const float4 be = *(const float4*)(beta + c * 4);
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + be.x;
            o.y = (v[i].y - mu) * rs * g.y + be.y;
            o.z = (v[i].z - mu) * rs * g.z + be.z;
            o.w = (v[i].w - mu) * rs * g.w + be.w;
            *(float4*)(xr + c * 4) = o;
        }
    }
}

extern "C" int bf_vision_assemble(const float* patches, const float* cls, const float* pos,
                                  const float* gamma, const float* beta, float* x, int batch,
                                  int n_patch, int width, float eps, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_patch > 0 && (width & 3) == 0 && width <= LN_MAXC * 256, BF_EINVAL,
               "bf_vision_assemble: bad shape (n_patch=%d width=%d)", n_patch, width);
    BF_REQUIRE(bf_aligned16(patches) && bf_aligned16(cls) && bf_aligned16(pos) && bf_aligned16(x),
               BF_EALIGN, "bf_vision_assemble: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(bf_cdiv((long)batch * (n_patch + 1), 4)), block(256);
    hipLaunchKernelGGL(vision_assemble_kernel, grid, block, 0, s, patches, cls, pos, gamma, beta, x,
                       batch, n_patch, width, eps);
    BF_CHECK_LAUNCH("bf_vision_assemble");
    return BF_OK;
}

// ------------------------------------------------------------------ K1: class token + ln_pre + first ln_1
// The patch-embed GEMM's epilogue has already written  patches + pos[1 + p]  into rows b*(n+1) + 1 + p of x
// (row remap + modulo-indexed residual, the machinery of K8).  This pass finishes the token assembly in place:
// row b*(n+1) becomes cls + pos[0], every row goes through ln_pre (fp32, written back: it is the residual stream),
// and the SAME registers go through block 0's ln_1, whose output (the first QKV GEMM's operand) is written in the
// compute dtype -- one read of the stream instead of three (assembly, ln_pre, ln_1; models/clip/model.py:300-309,
// :206-207).
template <typename TY>
__global__ __launch_bounds__(256) void vision_lnpre_ln1_kernel(
    float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos0,
    const float* __restrict__ g0, const float* __restrict__ b0, const float* __restrict__ g1,
    const float* __restrict__ b1, TY* __restrict__ h, int rows, int seq, int width, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool is_cls = (row % seq) == 0;
    float* xr = x + (size_t)row * width;
    const int nvec = width >> 2;
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            if (is_cls) {
                const float4 a = *(const float4*)(cls + c * 4);
                const float4 p = *(const float4*)(pos0 + c * 4);
                v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            } else {
                v[i] = *(const float4*)(xr + c * 4);
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    // ln_pre
    float mu = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float a = v[i].x - mu, bb = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    float rs = rsqrtf(wave_sum(q) / (float)width + eps);
    s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 g = *(const float4*)(g0 + c * 4);
            const float4 be = *(const float4*)(b0 + c * 4);
            v[i].x = (v[i].x - mu) * rs * g.x + be.x;
            v[i].y = (v[i].y - mu) * rs * g.y + be.y;
            v[i].z = (v[i].z - mu) * rs * g.z + be.z;
            v[i].w = (v[i].w - mu) * rs * g.w + be.w;
            *(float4*)(xr + c * 4) = v[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    // ln_1 of the first block on the values just written
    mu = wave_sum(s) / (float)width;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float a = v[i].x - mu, bb = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    rs = rsqrtf(wave_sum(q) / (float)width + eps);
    TY* hr = h + (size_t)row * width;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 g = *(const float4*)(g1 + c * 4);
            const float4 be = *(const float4*)(b1 + c * 4);
            float4 o;
            o.x = (v[i].x - mu) * rs * g.x + be.x;
            o.y = (v[i].y - mu) * rs * g.y + be.y;
            o.z = (v[i].z - mu) * rs * g.z + be.z;
            o.w = (v[i].w - mu) * rs * g.w + be.w;
            Vec4<TY>::store(hr + c * 4, o);
        }
    }
}

extern "C" int bf_vision_lnpre_ln1(float* x, const float* cls, const float* pos0, const float* g0, const float* b0,
                                   const float* g1, const float* b1, void* h, int h_dtype, int batch, int n_patch,
                                   int width, float eps, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_patch > 0 && (width & 3) == 0 && width <= LN_MAXC * 256, BF_EINVAL,
               "bf_vision_lnpre_ln1: bad shape (n_patch=%d width=%d)", n_patch, width);
    BF_REQUIRE(h_dtype == BF_F32 || bf_is_half(h_dtype), BF_EINVAL, "bf_vision_lnpre_ln1: bad output dtype %d", h_dtype);
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(cls) && bf_aligned16(pos0) && bf_aligned16(h) && bf_aligned16(g0) &&
               bf_aligned16(b0) && bf_aligned16(g1) && bf_aligned16(b1), BF_EALIGN, "bf_vision_lnpre_ln1: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rows = batch * (n_patch + 1);
    dim3 grid(bf_cdiv((long)rows, 4)), block(256);
    if (h_dtype == BF_BF16)
        hipLaunchKernelGGL(vision_lnpre_ln1_kernel<bf16_t>, grid, block, 0, s, x, cls, pos0, g0, b0, g1, b1, (bf16_t*)h,
                           rows, n_patch + 1, width, eps);
    else if (h_dtype == BF_F16)
        hipLaunchKernelGGL(vision_lnpre_ln1_kernel<f16_t>, grid, block, 0, s, x, cls, pos0, g0, b0, g1, b1, (f16_t*)h,
                           rows, n_patch + 1, width, eps);
    else
        hipLaunchKernelGGL(vision_lnpre_ln1_kernel<float>, grid, block, 0, s, x, cls, pos0, g0, b0, g1, b1, (float*)h,
                           rows, n_patch + 1, width, eps);
    BF_CHECK_LAUNCH("bf_vision_lnpre_ln1");
    return BF_OK;
}

// ------------------------------------------------------------------ token embedding
__global__ __launch_bounds__(256) void embed_tokens_kernel(
    const int32_t* __restrict__ tokens, const float* __restrict__ table,
    const float* __restrict__ pos, float* __restrict__ x, int rows, int ctx, int width, int vocab,
    int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int id = tokens[row];
    if (id < 0 || id >= vocab) {
        if (lane == 0 && err_flag) atomicExch(err_flag, 1);
        id = 0;
    }
    const int t = row % ctx;
    const float* e = table + (size_t)id * width;
    const float* p = pos + (size_t)t * width;
    float* o = x + (size_t)row * width;
    for (int c = lane; c < (width >> 2); c += 64) {
        const float4 a = *(const float4*)(e + c * 4);
        const float4 b = *(const float4*)(p + c * 4);
        *(float4*)(o + c * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

extern "C" int bf_embed_tokens(const int32_t* tokens, const float* table, const float* pos, float* x,
                               int batch, int ctx, int width, int vocab, int* err_flag,
                               bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && ctx > 0 && (width & 3) == 0 && vocab > 0, BF_EINVAL,
               "bf_embed_tokens: bad shape (ctx=%d width=%d vocab=%d)", ctx, width, vocab);
    BF_REQUIRE(bf_aligned16(table) && bf_aligned16(pos) && bf_aligned16(x), BF_EALIGN,
               "bf_embed_tokens: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rows = batch * ctx;
    dim3 grid(bf_cdiv(rows, 4)), block(256);
    hipLaunchKernelGGL(embed_tokens_kernel, grid, block, 0, s, tokens, table, pos, x, rows, ctx, width,
                       vocab, err_flag);
    BF_CHECK_LAUNCH("bf_embed_tokens");
    return BF_OK;
}

// ------------------------------------------------------------------ fusion input assembly
__global__ __launch_bounds__(256) void fusion_assemble_kernel(
    const float* __restrict__ text_feat, const float* __restrict__ rgb_feat,
    const float* __restrict__ text_token, const float* __restrict__ text_pos,
    const float* __restrict__ rgb_pos, const float* __restrict__ type_emb, float* __restrict__ x,
    int batch, int n_text, int n_rgb, int dim) {
    const int L = 1 + n_text + n_rgb;
    const int nvec = dim >> 2;
    const size_t total = (size_t)batch * L * nvec;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % nvec);
        const size_t row = i / nvec;
        const int b = (int)(row / L), t = (int)(row - (size_t)b * L);
        const float* src;
        const float* ps;
        const float* ty;
        if (t == 0) {
            src = text_token; ps = text_pos; ty = type_emb;
        } else if (t <= n_text) {
            src = text_feat + ((size_t)b * n_text + (t - 1)) * dim;
            ps = text_pos + (size_t)t * dim; ty = type_emb;
        } else {
            const int p = t - 1 - n_text;
            src = rgb_feat + ((size_t)b * n_rgb + p) * dim;
            ps = rgb_pos + (size_t)p * dim; ty = type_emb + dim;
        }
        const float4 a = *(const float4*)(src + c * 4);
        const float4 p4 = *(const float4*)(ps + c * 4);
        const float4 t4 = *(const float4*)(ty + c * 4);
        // reference order: (feat + pos) + type   (rgb_clip.py:55,64 then fusion.py:52)
        *(float4*)(x + row * dim + c * 4) = make_float4((a.x + p4.x) + t4.x, (a.y + p4.y) + t4.y,
                                                        (a.z + p4.z) + t4.z, (a.w + p4.w) + t4.w);
    }
}

extern "C" int bf_fusion_assemble(const float* text_feat, const float* rgb_feat,
                                  const float* text_token, const float* text_pos,
                                  const float* rgb_pos, const float* type_emb, float* x, int batch,
                                  int n_text, int n_rgb, int dim, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_text > 0 && n_rgb > 0 && (dim & 3) == 0, BF_EINVAL,
               "bf_fusion_assemble: bad shape");
    BF_REQUIRE(bf_aligned16(text_feat) && bf_aligned16(rgb_feat) && bf_aligned16(text_token) &&
                   bf_aligned16(text_pos) && bf_aligned16(rgb_pos) && bf_aligned16(type_emb) &&
                   bf_aligned16(x), BF_EALIGN, "bf_fusion_assemble: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)batch * (1 + n_text + n_rgb) * (dim / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fusion_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, s, text_feat,
                       rgb_feat, text_token, text_pos, rgb_pos, type_emb, x, batch, n_text, n_rgb, dim);
    BF_CHECK_LAUNCH("bf_fusion_assemble");
    return BF_OK;
}

// backward, stage 1: one thread per (token, float4 column): sum dX over the batch, and
// write the rgb rows out in the compute dtype.
template <typename TC, typename TI = float>
__global__ __launch_bounds__(256) void fusion_assemble_bwd_kernel(
    const TI* __restrict__ dx, TC* __restrict__ d_rgb_feat, float* __restrict__ d_text_pos,
    float* __restrict__ d_rgb_pos, int batch, int n_text, int n_rgb, int dim, int per_chunk) {
    // grid.y cuts the batch into chunks: one workgroup column per chunk, partial sums by fp32 atomics into the
    // zeroed per-token gradients (one thread looping over all 256 samples left 138 workgroups on the chip
    // streaming 144 MB at 0.9 TB/s)
    const int L = 1 + n_text + n_rgb;
    const int nvec = dim >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L * nvec) return;
    const int t = i / nvec, c = i - t * nvec;
    const int b0 = blockIdx.y * per_chunk, b1 = min(batch, b0 + per_chunk);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int b = b0; b < b1; ++b) {
        const float4 v = Vec4<TI>::load(dx + ((size_t)b * L + t) * dim + c * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        if (t > n_text)
            Vec4<TC>::store(d_rgb_feat + ((size_t)b * n_rgb + (t - 1 - n_text)) * dim + c * 4, v);
    }
    float* o = (t <= n_text) ? d_text_pos + (size_t)t * dim + c * 4
                             : d_rgb_pos + (size_t)(t - 1 - n_text) * dim + c * 4;
    atomicAdd(o + 0, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
}
// stage 2: type-embedding gradients (sum the per-token sums over each modality) and text_token.
// Block = 16 columns x 16 row groups.
__global__ __launch_bounds__(256) void fusion_assemble_bwd2_kernel(
    const float* __restrict__ d_text_pos, const float* __restrict__ d_rgb_pos,
    float* __restrict__ d_text_token, float* __restrict__ d_type, int n_text, int n_rgb, int dim) {
    __shared__ float r0[16][17], r1[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < dim) {
        for (int t = rg; t <= n_text; t += 16) s0 += d_text_pos[(size_t)t * dim + c];
        for (int p = rg; p < n_rgb; p += 16) s1 += d_rgb_pos[(size_t)p * dim + c];
    }
    r0[rg][cl] = s0; r1[rg][cl] = s1;
    __syncthreads();
    if (threadIdx.x < 16 && c < dim) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a += r0[i][threadIdx.x]; b += r1[i][threadIdx.x]; }
        d_type[c] = a;
        d_type[dim + c] = b;
        d_text_token[c] = d_text_pos[c];
    }
}

extern "C" int bf_fusion_assemble_bwd(const void* dx, int dx_dtype, void* d_rgb_feat, int cdt,
                                      float* d_text_token, float* d_text_pos, float* d_rgb_pos,
                                      float* d_type, int batch, int n_text, int n_rgb, int dim,
                                      bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_text > 0 && n_rgb > 0 && (dim & 3) == 0, BF_EINVAL,
               "bf_fusion_assemble_bwd: bad shape");
    BF_REQUIRE(bf_aligned16(dx) && bf_aligned16(d_rgb_feat) && bf_aligned16(d_text_pos) &&
                   bf_aligned16(d_rgb_pos), BF_EALIGN, "bf_fusion_assemble_bwd: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int L = 1 + n_text + n_rgb;
    BF_REQUIRE(bf_zero_f32_async(d_text_pos, (size_t)(1 + n_text) * dim, s) == hipSuccess &&
                   bf_zero_f32_async(d_rgb_pos, (size_t)n_rgb * dim, s) == hipSuccess,
               BF_ELAUNCH, "bf_fusion_assemble_bwd: zeroing the position gradients failed");
    int chunks = batch >= 64 ? 16 : (batch >= 8 ? 4 : 1);
    const int per_chunk = bf_cdiv(batch, chunks);
    chunks = bf_cdiv(batch, per_chunk);
    dim3 grid(bf_cdiv((long)L * (dim / 4), 256), chunks), block(256);
    // dx: fp32, or (the 16-bit gradient stream of the fp16 mode) a tensor of the compute dtype
    BF_REQUIRE(dx_dtype == BF_F32 || (dx_dtype == BF_F16 && cdt == BF_F16), BF_EINVAL,
               "bf_fusion_assemble_bwd: dx is fp32, or fp16 together with an fp16 d_rgb_feat (got %d / %d)", dx_dtype, cdt);
    if (dx_dtype == BF_F16)
        hipLaunchKernelGGL((fusion_assemble_bwd_kernel<f16_t, f16_t>), grid, block, 0, s, (const f16_t*)dx,
                           (f16_t*)d_rgb_feat, d_text_pos, d_rgb_pos, batch, n_text, n_rgb, dim, per_chunk);
    else if (cdt == BF_F32)
        hipLaunchKernelGGL(fusion_assemble_bwd_kernel<float>, grid, block, 0, s, (const float*)dx, (float*)d_rgb_feat,
                           d_text_pos, d_rgb_pos, batch, n_text, n_rgb, dim, per_chunk);
    else if (cdt == BF_BF16)
        hipLaunchKernelGGL(fusion_assemble_bwd_kernel<bf16_t>, grid, block, 0, s, (const float*)dx, (bf16_t*)d_rgb_feat,
                           d_text_pos, d_rgb_pos, batch, n_text, n_rgb, dim, per_chunk);
    else if (cdt == BF_F16)
        hipLaunchKernelGGL(fusion_assemble_bwd_kernel<f16_t>, grid, block, 0, s, (const float*)dx, (f16_t*)d_rgb_feat,
                           d_text_pos, d_rgb_pos, batch, n_text, n_rgb, dim, per_chunk);
    else BF_REQUIRE(false, BF_EINVAL, "bf_fusion_assemble_bwd: bad dtype %d", cdt);
    BF_CHECK_LAUNCH("bf_fusion_assemble_bwd/1");
    hipLaunchKernelGGL(fusion_assemble_bwd2_kernel, dim3(bf_cdiv(dim, 16)), block, 0, s, d_text_pos,
                       d_rgb_pos, d_text_token, d_type, n_text, n_rgb, dim);
    BF_CHECK_LAUNCH("bf_fusion_assemble_bwd/2");
    return BF_OK;
}

// ------------------------------------------------------------------ generic fusion-input segment
__global__ __launch_bounds__(256) void segment_assemble_kernel(
    const float* __restrict__ src, long src_bstride, const float* __restrict__ lead,
    const float* __restrict__ pos, const float* __restrict__ type_vec, float* __restrict__ x, int batch,
    int n, int seq_total, int row_off, int dim) {
    const int nvec = dim >> 2;
    const size_t total = (size_t)batch * (n + 1) * nvec;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % nvec);
        const size_t row = i / nvec;
        const int b = (int)(row / (n + 1)), t = (int)(row - (size_t)b * (n + 1));
        const float* s = (t == 0) ? lead : src + (size_t)b * src_bstride + (size_t)(t - 1) * dim;
        float4 v = *(const float4*)(s + c * 4);
        if (pos) {
            const float4 p = *(const float4*)(pos + (size_t)t * dim + c * 4);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const float4 ty = *(const float4*)(type_vec + c * 4);
        v.x += ty.x; v.y += ty.y; v.z += ty.z; v.w += ty.w;
        *(float4*)(x + ((size_t)b * seq_total + row_off + t) * dim + c * 4) = v;
    }
}

extern "C" int bf_segment_assemble(const float* src, long src_bstride, const float* lead, const float* pos,
                                   const float* type_vec, float* x, int batch, int n, int seq_total,
                                   int row_off, int dim, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n > 0 && (dim & 3) == 0 && row_off >= 0 && row_off + n + 1 <= seq_total, BF_EINVAL,
               "bf_segment_assemble: bad shape (n=%d row_off=%d seq_total=%d dim=%d)", n, row_off, seq_total, dim);
    BF_REQUIRE(bf_aligned16(src) && bf_aligned16(lead) && bf_aligned16(type_vec) && bf_aligned16(x) &&
                   (!pos || bf_aligned16(pos)) && (src_bstride & 3) == 0, BF_EALIGN,
               "bf_segment_assemble: unaligned pointer or stride");
    const size_t total = (size_t)batch * (n + 1) * (dim / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(segment_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       src_bstride, lead, pos, type_vec, x, batch, n, seq_total, row_off, dim);
    BF_CHECK_LAUNCH("bf_segment_assemble");
    return BF_OK;
}

__global__ __launch_bounds__(256) void segment_bwd1_kernel(const float* __restrict__ dx, float* __restrict__ tmp,
                                                           float* __restrict__ d_pos, int batch, int n,
                                                           int seq_total, int row_off, int dim) {
    const int nvec = dim >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (n + 1) * nvec) return;
    const int t = i / nvec, c = i - t * nvec;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < batch; ++b) {
        const float4 v = *(const float4*)(dx + ((size_t)b * seq_total + row_off + t) * dim + c * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(tmp + (size_t)t * dim + c * 4) = acc;
    if (d_pos) *(float4*)(d_pos + (size_t)t * dim + c * 4) = acc;
}
__global__ __launch_bounds__(256) void segment_bwd2_kernel(const float* __restrict__ tmp, float* __restrict__ d_lead,
                                                           float* __restrict__ d_type, int n, int dim) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= dim) return;
    float s = 0.f;
    for (int t = 0; t <= n; ++t) s += tmp[(size_t)t * dim + c];
    d_type[c] += s;
    d_lead[c] += tmp[c];
}

extern "C" int bf_segment_assemble_bwd(const float* dx, float* d_lead, float* d_pos, float* d_type, float* tmp_ws,
                                       int batch, int n, int seq_total, int row_off, int dim, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n > 0 && (dim & 3) == 0 && row_off >= 0 && row_off + n + 1 <= seq_total, BF_EINVAL,
               "bf_segment_assemble_bwd: bad shape");
    BF_REQUIRE(bf_aligned16(dx) && bf_aligned16(tmp_ws) && (!d_pos || bf_aligned16(d_pos)) && d_lead && d_type,
               BF_EALIGN, "bf_segment_assemble_bwd: unaligned or missing pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(segment_bwd1_kernel, dim3(bf_cdiv((long)(n + 1) * (dim / 4), 256)), dim3(256), 0, s, dx,
                       tmp_ws, d_pos, batch, n, seq_total, row_off, dim);
    BF_CHECK_LAUNCH("bf_segment_assemble_bwd/1");
    hipLaunchKernelGGL(segment_bwd2_kernel, dim3(bf_cdiv(dim, 256)), dim3(256), 0, s, tmp_ws, d_lead, d_type, n, dim);
    BF_CHECK_LAUNCH("bf_segment_assemble_bwd/2");
    return BF_OK;
}

// ----------------------------------------------------------------------------------------
// Dropout with a REPLAYABLE mask (LoRA's `lora_dropout` on the adapter input, reference
// bifold/models/siglip.py:33-40 -> peft LoraLayer: lora_B(lora_A(dropout(x)))).  The keep decision of
// element i is a pure function of (seed, i) (common.h: splitmix64 of i >> 2, 15 bits >= p * 2^15), so the backward pass
// (and a CPU oracle, through bf_dropout_mask) regenerates the mask instead of storing it.
// ----------------------------------------------------------------------------------------
// (dropout_keep / dropout_threshold: common.h -- the attention kernels with dropout on the probabilities share them)
template <typename T, bool ACC>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n4,
                                                      unsigned thr, float inv_keep, unsigned long long seed) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = Vec4<T>::load(x + i * 4);
        float* vp = (float*)&v;
        float4 o = ACC ? Vec4<T>::load(y + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* op = (float*)&o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            op[e] += dropout_keep(i * 4 + e, seed, thr) ? vp[e] * inv_keep : 0.f;
        Vec4<T>::store(y + i * 4, o);
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, size_t n, unsigned thr,
                                                           unsigned long long seed) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        mask[i] = dropout_keep(i, seed, thr) ? 1 : 0;
}


extern "C" int bf_dropout(const void* x, void* y, int dtype, size_t count, float p, unsigned long long seed,
                          int accumulate, bf_stream_t stream) {
    BF_REQUIRE(x && y && count > 0 && (count & 3) == 0, BF_EINVAL, "bf_dropout: count must be a positive multiple of 4");
    BF_REQUIRE(p >= 0.f && p < 1.f, BF_EINVAL, "bf_dropout: p must be in [0, 1) (got %f)", (double)p);
    BF_REQUIRE(dtype == BF_F32 || bf_is_half(dtype), BF_EINVAL, "bf_dropout: bad dtype");
    BF_REQUIRE(bf_aligned16(x) && bf_aligned16(y), BF_EALIGN, "bf_dropout: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const unsigned thr = dropout_threshold(p);
    const float inv_keep = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
#define BF_DROP(T, ACC) hipLaunchKernelGGL((dropout_kernel<T, ACC>), dim3((unsigned)blocks), dim3(256), 0, s, \
                                           (const T*)x, (T*)y, n4, thr, inv_keep, seed)
    if (dtype == BF_F32) { if (accumulate) BF_DROP(float, true); else BF_DROP(float, false); }
    else if (dtype == BF_F16) { if (accumulate) BF_DROP(f16_t, true); else BF_DROP(f16_t, false); }
    else { if (accumulate) BF_DROP(bf16_t, true); else BF_DROP(bf16_t, false); }
#undef BF_DROP
    BF_CHECK_LAUNCH("bf_dropout");
    return BF_OK;
}

extern "C" int bf_dropout_mask(uint8_t* mask, size_t count, float p, unsigned long long seed, bf_stream_t stream) {
    BF_REQUIRE(mask && count > 0 && p >= 0.f && p < 1.f, BF_EINVAL, "bf_dropout_mask: bad args");
    size_t blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, count,
                       dropout_threshold(p), seed);
    BF_CHECK_LAUNCH("bf_dropout_mask");
    return BF_OK;
}

// Dropout on a ROW RANGE of every group of a [batch, group_rows, cols] tensor, in place, plus an optional per-column
// vector added AFTER the mask: the embeddings' nn.Dropout of RGBOnly (ref:bifold/models/rgb_clip.py:39-42,56,65:
// `x = dropout(x + pos)`) applied to the text / image rows of the assembled fusion input, followed by fusion.py:52's
// `inp += token_type_embeddings(mod)`.  Mask index of element (b, r, c) of the range: (b * nrows + r) * cols + c, i.e. the
// flat index of the modality's own [batch, nrows, cols] tensor (what bf_dropout_mask replays).  p = 0: only the add.
template <typename T>
__global__ __launch_bounds__(256) void dropout_rows_kernel(T* __restrict__ x, const float* __restrict__ add, int group_rows,
                                                           int row0, int nrows, int cols, size_t total4, unsigned thr,
                                                           float inv_keep, unsigned long long seed) {
    const int c4n = cols >> 2;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t row = i / c4n;                            // row of the [batch * nrows] range
        const int c = (int)(i - row * c4n) * 4;
        const size_t b = row / nrows, r = row - b * nrows;
        T* p = x + ((b * group_rows + row0 + r) * (size_t)cols + c);
        float4 v = Vec4<T>::load(p);
        float* vp = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (thr) vp[e] = dropout_keep(row * cols + c + e, seed, thr) ? vp[e] * inv_keep : 0.f;
            if (add) vp[e] += add[c + e];
        }
        Vec4<T>::store(p, v);
    }
}

extern "C" int bf_dropout_rows(void* x, int dtype, const float* add, int batch, int group_rows, int row0, int nrows, int cols,
                               float p, unsigned long long seed, bf_stream_t stream) {
    BF_REQUIRE(x && batch > 0 && group_rows > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= group_rows && cols > 0 &&
               (cols & 3) == 0, BF_EINVAL, "bf_dropout_rows: bad shape");
    BF_REQUIRE(p >= 0.f && p < 1.f, BF_EINVAL, "bf_dropout_rows: p must be in [0, 1) (got %f)", (double)p);
    BF_REQUIRE(dtype == BF_F32 || bf_is_half(dtype), BF_EINVAL, "bf_dropout_rows: bad dtype");
    BF_REQUIRE(bf_aligned16(x), BF_EALIGN, "bf_dropout_rows: unaligned pointer");
    const size_t total4 = (size_t)batch * nrows * (cols >> 2);
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const unsigned thr = dropout_threshold(p);
    const float inv_keep = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
#define DROWS(T) hipLaunchKernelGGL((dropout_rows_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)x, add, group_rows, \
                                    row0, nrows, cols, total4, thr, inv_keep, seed)
    if (dtype == BF_F32) DROWS(float); else if (dtype == BF_F16) DROWS(f16_t); else DROWS(bf16_t);
#undef DROWS
    BF_CHECK_LAUNCH("bf_dropout_rows");
    return BF_OK;
}
