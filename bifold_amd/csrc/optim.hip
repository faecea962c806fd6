// Fused Adam / AdamW over the flat fp32 parameter arena, gradient sum-of-squares and the
// clip coefficient.  HBM-bound: 4 fp32 reads + 3 fp32 writes (+ 1 bf16 write) per element.
#include "common.h"

#include <math.h>

template <bool SHADOW, typename TS>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   TS* __restrict__ shadow, size_t n4, float lr,
                                                   float beta1, float beta2, float eps, float wd,
                                                   int decoupled, float step_size, float sqrt_bc2,
                                                   const float* __restrict__ gscale,
                                                   const unsigned* __restrict__ skip_flag,
                                                   const float* __restrict__ skipped_dev, float skipped_known,
                                                   int step) {
    // Guarded step (fp16 mode's dynamic loss scale): a non-zero device flag -- non-finite gradient elements counted by
    // bf_scale_check during this backward, MAX-reduced over the ranks -- makes the whole launch a no-op for p / m / v
    // and the 16-bit copy, like torch.cuda.amp.GradScaler.step skipping optimizer.step(), without a host read.
    if (skip_flag && skip_flag[0] != 0u) return;
    // A skipped step does not count for Adam's bias corrections.  The host computed step_size / sqrt_bc2 for
    // `step` = its own count minus the skips it KNEW of; skips since then (a device counter) are taken off here.
    if (skipped_dev) {
        const float extra = skipped_dev[0] - skipped_known;
        if (extra > 0.5f) {
            const double t = fmax(1.0, (double)step - (double)extra);
            step_size = (float)((double)lr / (1.0 - pow((double)beta1, t)));
            sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, t));
        }
    }
    const float gs = gscale ? gscale[0] : 1.0f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pv = ((const float4*)p)[i];
        float4 gv = ((const float4*)g)[i];
        float4 mv = ((const float4*)m)[i];
        float4 vv = ((const float4*)v)[i];
        float* pp = (float*)&pv; float* gp = (float*)&gv; float* mp = (float*)&mv; float* vp = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float grad = gp[e] * gs;
            float par = pp[e];
            if (wd != 0.f) {
                if (decoupled) par *= (1.0f - lr * wd);
                else grad = fmaf(wd, par, grad);
            }
            // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
            const float m1 = mp[e] + (grad - mp[e]) * (1.0f - beta1);
            const float v1 = vp[e] * beta2 + (1.0f - beta2) * grad * grad;
            const float denom = sqrtf(v1) / sqrt_bc2 + eps;
            par -= step_size * (m1 / denom);
            pp[e] = par; mp[e] = m1; vp[e] = v1;
        }
        ((float4*)p)[i] = pv;
        ((float4*)m)[i] = mv;
        ((float4*)v)[i] = vv;
        // the 16-bit copy is the rounding of the STORED fp32 master (what a cast of the arena gives): without the fence
        // the compiler folds the last fma and the conversion into v_fma_mix*_f16, i.e. one rounding of the exact result
        asm volatile("" : "+v"(pv.x), "+v"(pv.y), "+v"(pv.z), "+v"(pv.w));
        if (SHADOW) Vec4<TS>::store(shadow + i * 4, pv);
    }
}

extern "C" int bf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow,
                            int shadow_dtype, size_t count, int step, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int decoupled, const float* grad_scale_dev,
                            const unsigned* skip_flag_dev, const float* skipped_dev, float skipped_known,
                            bf_stream_t stream) {
    BF_REQUIRE(count > 0 && (count & 3) == 0, BF_EINVAL,
               "bf_adam_step: the arena length must be a multiple of 4 (got %zu)", count);
    BF_REQUIRE(step >= 1, BF_EINVAL, "bf_adam_step: step is 1-based (got %d)", step);
    BF_REQUIRE(!shadow || bf_is_half(shadow_dtype), BF_EINVAL, "bf_adam_step: the shadow copy is BF_BF16 or BF_F16 (got %d)",
               shadow_dtype);
    BF_REQUIRE(bf_aligned16(param) && bf_aligned16(grad) && bf_aligned16(exp_avg) && bf_aligned16(exp_avg_sq) &&
                   (!shadow || bf_aligned16(shadow)), BF_EALIGN, "bf_adam_step: unaligned pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (shadow && shadow_dtype == BF_F16)
        hipLaunchKernelGGL((adam_kernel<true, f16_t>), grid, block, 0, s, param, grad, exp_avg, exp_avg_sq, (f16_t*)shadow,
                           n4, lr, beta1, beta2, eps, weight_decay, decoupled, step_size, sqrt_bc2, grad_scale_dev,
                           skip_flag_dev, skipped_dev, skipped_known, step);
    else if (shadow)
        hipLaunchKernelGGL((adam_kernel<true, bf16_t>), grid, block, 0, s, param, grad, exp_avg, exp_avg_sq, (bf16_t*)shadow,
                           n4, lr, beta1, beta2, eps, weight_decay, decoupled, step_size, sqrt_bc2, grad_scale_dev,
                           skip_flag_dev, skipped_dev, skipped_known, step);
    else
        hipLaunchKernelGGL((adam_kernel<false, bf16_t>), grid, block, 0, s, param, grad, exp_avg, exp_avg_sq, (bf16_t*)nullptr,
                           n4, lr, beta1, beta2, eps, weight_decay, decoupled, step_size, sqrt_bc2, grad_scale_dev,
                           skip_flag_dev, skipped_dev, skipped_known, step);
    BF_CHECK_LAUNCH("bf_adam_step");
    return BF_OK;
}

#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n4, size_t count,
                                                    float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)x)[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const float t = x[n4 * 4 + threadIdx.x];
        acc += t * t;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial, int n,
                                                           float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int bf_sumsq(const float* x, size_t count, float* partial_ws, float* out, bf_stream_t stream) {
    BF_REQUIRE(count > 0, BF_EINVAL, "bf_sumsq: count must be > 0");
    BF_REQUIRE(bf_aligned16(x), BF_EALIGN, "bf_sumsq: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n4, count, partial_ws);
    BF_CHECK_LAUNCH("bf_sumsq/1");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial_ws, (int)blocks, out);
    BF_CHECK_LAUNCH("bf_sumsq/2");
    return BF_OK;
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ out) {
    const float norm = sqrtf(sumsq[0]);
    out[0] = fminf(1.0f, max_norm / (norm + 1e-6f));
}
extern "C" int bf_clip_coef(const float* sumsq, float max_norm, float* out, bf_stream_t stream) {
    BF_REQUIRE(sumsq && out && max_norm > 0.f, BF_EINVAL, "bf_clip_coef: bad args");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, out);
    BF_CHECK_LAUNCH("bf_clip_coef");
    return BF_OK;
}

// torch.nn.utils.clip_grad_norm_ over the flat gradient arena, entirely on the device:
// ws[0..1023] partial sums, ws[1024] = total norm, ws[1025] = clip coefficient.
__global__ __launch_bounds__(256) void clip_finish_kernel(float* __restrict__ ws, int n, float max_norm,
                                                          float* __restrict__ norm_out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += ws[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        ws[SUMSQ_BLOCKS] = norm;
        ws[SUMSQ_BLOCKS + 1] = fminf(1.0f, max_norm / (norm + 1e-6f));
        norm_out[0] = norm;
    }
}
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ x, size_t n4, size_t count,
                                                            const float* __restrict__ coef) {
    const float c = coef[0];
    if (c >= 1.0f) return;                       // torch multiplies by the clamped 1.0: the identity
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ((float4*)x)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        ((float4*)x)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) x[n4 * 4 + threadIdx.x] *= c;
}

extern "C" int bf_clip_grad_norm(float* grad, size_t count, float max_norm, float* ws, float* norm_out,
                                 bf_stream_t stream) {
    BF_REQUIRE(grad && ws && norm_out && count > 0 && max_norm > 0.f, BF_EINVAL, "bf_clip_grad_norm: bad args");
    BF_REQUIRE(bf_aligned16(grad), BF_EALIGN, "bf_clip_grad_norm: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad, n4, count, ws);
    BF_CHECK_LAUNCH("bf_clip_grad_norm/1");
    hipLaunchKernelGGL(clip_finish_kernel, dim3(1), dim3(256), 0, s, ws, (int)blocks, max_norm, norm_out);
    BF_CHECK_LAUNCH("bf_clip_grad_norm/2");
    size_t sblocks = (n4 + 255) / 256;
    if (sblocks > 4096) sblocks = 4096;
    if (sblocks < 1) sblocks = 1;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)sblocks), dim3(256), 0, s, grad, n4, count,
                       ws + SUMSQ_BLOCKS + 1);
    BF_CHECK_LAUNCH("bf_clip_grad_norm/3");
    return BF_OK;
}

// x *= scale in place, counting the non-finite results: the two ends of the fp16 mode's loss scaling (the gradient
// entering the tail's backward is multiplied by S, every slice of the gradient arena by 1 / S when it completes).
__global__ __launch_bounds__(256) void scale_check_kernel(float* __restrict__ x, size_t n4, size_t count, float c,
                                                          const float* __restrict__ c_dev,
                                                          unsigned* __restrict__ nonfinite) {
    if (c_dev) c = c_dev[0];
    unsigned bad = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ((float4*)x)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        ((float4*)x)[i] = v;
        bad += !isfinite(v.x) + !isfinite(v.y) + !isfinite(v.z) + !isfinite(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const float t = x[n4 * 4 + threadIdx.x] * c;
        x[n4 * 4 + threadIdx.x] = t;
        bad += !isfinite(t);
    }
    if (nonfinite && __any(bad != 0)) {
        bad = (unsigned)wave_sum((float)bad);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(nonfinite, bad);
    }
}

extern "C" int bf_scale_check(float* x, size_t count, float scale, unsigned* nonfinite_count, bf_stream_t stream) {
    BF_REQUIRE(x && count > 0, BF_EINVAL, "bf_scale_check: bad args");
    BF_REQUIRE(bf_aligned16(x), BF_EALIGN, "bf_scale_check: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, count, scale,
                       (const float*)nullptr, nonfinite_count);
    BF_CHECK_LAUNCH("bf_scale_check");
    return BF_OK;
}

extern "C" int bf_scale_check_dev(float* x, size_t count, const float* scale_dev, unsigned* nonfinite_count,
                                  bf_stream_t stream) {
    BF_REQUIRE(x && count > 0 && scale_dev, BF_EINVAL, "bf_scale_check_dev: bad args");
    BF_REQUIRE(bf_aligned16(x), BF_EALIGN, "bf_scale_check_dev: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, count, 1.0f,
                       scale_dev, nonfinite_count);
    BF_CHECK_LAUNCH("bf_scale_check_dev");
    return BF_OK;
}

// End of a guarded step: the loss scale backs off after an overflow and grows back (never past `max_scale`) after
// `growth_interval` clean steps; the per-step flag is folded into the running totals and cleared.  One thread.
//   state[0] = S, [1] = 1 / S, [2] = clean steps since the last change, [3] = steps skipped so far,
//   state[4] = non-finite gradient elements seen so far (diagnostics), [5] = overflows at S == min_scale
__global__ void gradscale_update_kernel(float* __restrict__ st, unsigned* __restrict__ flag, float backoff,
                                        float growth, int growth_interval, float min_scale, float max_scale) {
    const unsigned bad = flag[0];
    float S = st[0];
    if (bad) {
        if (S <= min_scale) st[5] += 1.0f;
        S = fmaxf(S * backoff, min_scale);
        st[2] = 0.0f;
        st[3] += 1.0f;
        st[4] += (float)bad;
    } else {
        const float clean = st[2] + 1.0f;
        if (clean >= (float)growth_interval && S < max_scale) {
            S = fminf(S * growth, max_scale);
            st[2] = 0.0f;
        } else {
            st[2] = clean;
        }
    }
    st[0] = S;
    st[1] = 1.0f / S;
    flag[0] = 0u;
}

extern "C" int bf_gradscale_update(float* state, unsigned* nonfinite_flag, float backoff, float growth,
                                   int growth_interval, float min_scale, float max_scale, bf_stream_t stream) {
    BF_REQUIRE(state && nonfinite_flag, BF_EINVAL, "bf_gradscale_update: null state / flag");
    BF_REQUIRE(backoff > 0.f && backoff < 1.f && growth >= 1.f && growth_interval >= 1 && min_scale > 0.f &&
                   max_scale >= min_scale, BF_EINVAL, "bf_gradscale_update: bad schedule (backoff %g growth %g every %d, "
               "scale in [%g, %g])", backoff, growth, growth_interval, min_scale, max_scale);
    hipLaunchKernelGGL(gradscale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, nonfinite_flag, backoff,
                       growth, growth_interval, min_scale, max_scale);
    BF_CHECK_LAUNCH("bf_gradscale_update");
    return BF_OK;
}
