// Heatmap decoder (collapsed affine form) + sigmoid + BCE, forward and backward.
//
// ConvDecoder (reference models/modules.py:33-74) is five 1x1 convs with no activation,
// interleaved with four bilinear x2 upsamples.  Both commute: the convs fold into one
// vector w_eff[D] and scalar b_eff, the upsamples into one separable operator U [S, g]
// whose rows sum to 1.  logits = U (tok . w_eff + b_eff) U^T.  The reference materialises
// ~60 MB of fp32 intermediates per sample and head; this form reads the tokens once and
// writes the heatmap once.  Everything here is HBM-bound and fp32.
#include "common.h"

struct Ptr5 { const float* p[5]; };
struct MPtr5 { float* p[5]; };

// out[c] = sum_k r[k] * W[k*cols + c]   (r in LDS), all threads of the block
// (the k range is split over blockDim.x / cols thread groups and summed through LDS `part`, so
// that all 1024 threads stream W with coalesced, independent loads: these kernels are single
// workgroups on tiny matrices and purely latency-bound)
__device__ __forceinline__ void vecmat(const float* r, const float* __restrict__ W, int rows, int cols,
                                       float* out_lds, float* out_glb, float* part) {
    const int groups = max(1, (int)blockDim.x / cols);
    const int gidx = threadIdx.x / cols, c = threadIdx.x - gidx * cols;
    if (gidx < groups) {
        const int per = (rows + groups - 1) / groups;
        const int k0 = gidx * per, k1 = min(rows, k0 + per);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = k0;
        for (; k + 3 < k1; k += 4) {
            a0 = fmaf(r[k], W[(size_t)k * cols + c], a0);
            a1 = fmaf(r[k + 1], W[(size_t)(k + 1) * cols + c], a1);
            a2 = fmaf(r[k + 2], W[(size_t)(k + 2) * cols + c], a2);
            a3 = fmaf(r[k + 3], W[(size_t)(k + 3) * cols + c], a3);
        }
        for (; k < k1; ++k) a0 = fmaf(r[k], W[(size_t)k * cols + c], a0);
        part[gidx * cols + c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < cols; cc += blockDim.x) {
        float acc = 0.f;
        for (int gi = 0; gi < groups; ++gi) acc += part[gi * cols + cc];
        if (out_lds) out_lds[cc] = acc;
        if (out_glb) out_glb[cc] = acc;
    }
}
__device__ __forceinline__ float block_dot(const float* a, const float* __restrict__ b, int n, float* red) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(a[i], b[i], s);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(1024) void collapse_fwd_kernel(Ptr5 w, Ptr5 b, float* __restrict__ w_eff,
                                                           float* __restrict__ b_eff,
                                                           float* __restrict__ rows_ws, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C1 = D / 2, C3 = D / 4;
    float* r2 = sm;             // [C1]
    float* r3 = r2 + C1;        // [C1]
    float* r4 = r3 + C1;        // [C3]
    float* r5 = r4 + C3;        // [C3]
    float* red = r5 + C3;       // [16]
    float* part = red + 16;     // [1024]
    for (int i = threadIdx.x; i < C3; i += blockDim.x) { r5[i] = w.p[4][i]; rows_ws[2 * C1 + C3 + i] = w.p[4][i]; }
    __syncthreads();
    vecmat(r5, w.p[3], C3, C3, r4, rows_ws + 2 * C1, part);
    __syncthreads();
    vecmat(r4, w.p[2], C3, C1, r3, rows_ws + C1, part);
    __syncthreads();
    vecmat(r3, w.p[1], C1, C1, r2, rows_ws, part);
    __syncthreads();
    vecmat(r2, w.p[0], C1, D, nullptr, w_eff, part);
    const float t4 = block_dot(r5, b.p[3], C3, red);
    const float t3 = block_dot(r4, b.p[2], C3, red);
    const float t2 = block_dot(r3, b.p[1], C1, red);
    const float t1 = block_dot(r2, b.p[0], C1, red);
    if (threadIdx.x == 0) b_eff[0] = b.p[4][0] + t4 + t3 + t2 + t1;
}

extern "C" int bf_decoder_collapse_fwd(const float* const* w, const float* const* b, float* w_eff,
                                       float* b_eff, float* rows_ws, int dim, bf_stream_t stream) {
    BF_REQUIRE(w && b && dim > 0 && (dim & 3) == 0, BF_EINVAL, "bf_decoder_collapse_fwd: bad dim %d", dim);
    Ptr5 pw, pb;
    for (int i = 0; i < 5; ++i) { pw.p[i] = w[i]; pb.p[i] = b[i]; }
    BF_REQUIRE(dim <= 1024, BF_EINVAL, "bf_decoder_collapse_fwd: dim must be <= 1024");
    const size_t lds = (size_t)(dim + dim / 2 + 16 + 1024) * sizeof(float);
    hipLaunchKernelGGL(collapse_fwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, pw, pb, w_eff,
                       b_eff, rows_ws, dim);
    BF_CHECK_LAUNCH("bf_decoder_collapse_fwd");
    return BF_OK;
}

// dr_prev[k] = sum_c W[k][c] * g[c] + gb * bias[k]  (wave per row);  dW[k][c] = r[k] * g[c]
__device__ __forceinline__ void back_step(const float* __restrict__ W, const float* __restrict__ bias,
                                          const float* r_glb, const float* g_lds, float gb, int rows,
                                          int cols, float* dW, float* dbias, float* dr_lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int k = wave; k < rows; k += nwaves) {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s = fmaf(W[(size_t)k * cols + c], g_lds[c], s);
        s = wave_sum(s);
        if (lane == 0) {
            dr_lds[k] = s + gb * bias[k];
            dbias[k] = gb * r_glb[k];
        }
    }
    for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
        const int k = i / cols, c = i - k * cols;
        dW[i] = r_glb[k] * g_lds[c];
    }
}

__global__ __launch_bounds__(1024) void collapse_bwd_kernel(Ptr5 w, Ptr5 b, const float* __restrict__ rows_ws,
                                                           const float* __restrict__ g_w_eff,
                                                           const float* __restrict__ g_b_eff, MPtr5 dw,
                                                           MPtr5 db, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C1 = D / 2, C3 = D / 4;
    float* ga = sm;          // [D]
    float* gbuf = ga + D;    // [D]
    const float gb = g_b_eff[0];
    for (int i = threadIdx.x; i < D; i += blockDim.x) ga[i] = g_w_eff[i];
    __syncthreads();
    const float* r2 = rows_ws;
    const float* r3 = rows_ws + C1;
    const float* r4 = rows_ws + 2 * C1;
    const float* r5 = rows_ws + 2 * C1 + C3;
    back_step(w.p[0], b.p[0], r2, ga, gb, C1, D, dw.p[0], db.p[0], gbuf);     // -> dr2 in gbuf
    __syncthreads();
    back_step(w.p[1], b.p[1], r3, gbuf, gb, C1, C1, dw.p[1], db.p[1], ga);    // -> dr3 in ga
    __syncthreads();
    back_step(w.p[2], b.p[2], r4, ga, gb, C3, C1, dw.p[2], db.p[2], gbuf);    // -> dr4 in gbuf
    __syncthreads();
    back_step(w.p[3], b.p[3], r5, gbuf, gb, C3, C3, dw.p[3], db.p[3], ga);    // -> dr5 in ga
    __syncthreads();
    for (int i = threadIdx.x; i < C3; i += blockDim.x) dw.p[4][i] = ga[i];
    if (threadIdx.x == 0) db.p[4][0] = gb;
}

extern "C" int bf_decoder_collapse_bwd(const float* const* w, const float* const* b, const float* rows_ws,
                                       const float* g_w_eff, const float* g_b_eff, float* const* dw,
                                       float* const* db, int dim, bf_stream_t stream) {
    BF_REQUIRE(w && b && dw && db && dim > 0 && (dim & 3) == 0, BF_EINVAL, "bf_decoder_collapse_bwd: bad args");
    Ptr5 pw, pb;
    MPtr5 pdw, pdb;
    for (int i = 0; i < 5; ++i) { pw.p[i] = w[i]; pb.p[i] = b[i]; pdw.p[i] = dw[i]; pdb.p[i] = db[i]; }
    const size_t lds = (size_t)2 * dim * sizeof(float);
    hipLaunchKernelGGL(collapse_bwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, pw, pb, rows_ws,
                       g_w_eff, g_b_eff, pdw, pdb, dim);
    BF_CHECK_LAUNCH("bf_decoder_collapse_bwd");
    return BF_OK;
}

// ------------------------------------------------------------------ token projection
#define DEC_MAXH 4
#define DEC_MAXC 4        // float4 chunks per lane -> dim <= 1024

__global__ __launch_bounds__(256) void decoder_project_kernel(const float* __restrict__ tok,
                                                              const float* __restrict__ w_eff,
                                                              const float* __restrict__ b_eff,
                                                              float* __restrict__ low, int batch,
                                                              int n_heads, int n_patch, int dim,
                                                              long tok_bstride) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= batch * n_patch) return;
    const int b = t / n_patch, p = t - b * n_patch;
    const float* row = tok + (size_t)b * tok_bstride + (size_t)(1 + p) * dim;
    const int nvec = dim >> 2;
    float acc[DEC_MAXH];
#pragma unroll
    for (int h = 0; h < DEC_MAXH; ++h) acc[h] = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const float4 x = *(const float4*)(row + c * 4);
#pragma unroll
            for (int h = 0; h < DEC_MAXH; ++h)
                if (h < n_heads) {
                    const float4 w = *(const float4*)(w_eff + (size_t)h * dim + c * 4);
                    acc[h] += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
                }
        }
    }
#pragma unroll
    for (int h = 0; h < DEC_MAXH; ++h)
        if (h < n_heads) {
            const float s = wave_sum(acc[h]);
            if (lane == 0) low[((size_t)h * batch + b) * n_patch + p] = s + b_eff[h];
        }
}

extern "C" int bf_decoder_project(const float* tok, long tok_bstride, const float* w_eff, const float* b_eff,
                                  float* low, int batch, int n_heads, int n_patch, int dim, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_heads >= 1 && n_heads <= DEC_MAXH && n_patch > 0 && (dim & 3) == 0 &&
                   dim <= DEC_MAXC * 256, BF_EINVAL, "bf_decoder_project: bad shape (heads=%d dim=%d)", n_heads, dim);
    BF_REQUIRE(bf_aligned16(tok) && bf_aligned16(w_eff) && (tok_bstride & 3) == 0, BF_EALIGN,
               "bf_decoder_project: unaligned pointer or stride");
    hipLaunchKernelGGL(decoder_project_kernel, dim3(bf_cdiv((long)batch * n_patch, 4)), dim3(256), 0,
                       (hipStream_t)stream, tok, w_eff, b_eff, low, batch, n_heads, n_patch, dim, tok_bstride);
    BF_CHECK_LAUNCH("bf_decoder_project");
    return BF_OK;
}

// ------------------------------------------------------------------ upsample + sigmoid
#define DEC_MAXG 24
__global__ __launch_bounds__(256) void upsample_sigmoid_kernel(const float* __restrict__ low,
                                                               const float* __restrict__ U,
                                                               float* __restrict__ heat, int g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = g * 16;
    float* Us = sm;               // [S][g]
    float* lw = Us + S * g;       // [g][g]
    float* T = lw + ((g * g + 3) & ~3);   // [g][S]   T[i][x] = sum_j low[i][j] U[x][j]  (16-byte aligned for every g)
    const size_t bh = blockIdx.x;
    for (int i = threadIdx.x; i < S * g; i += 256) Us[i] = U[i];
    for (int i = threadIdx.x; i < g * g; i += 256) lw[i] = low[bh * g * g + i];
    __syncthreads();
    // U[x][j] is non-zero only for j within one coarse cell of x / 16 (four composed bilinear x2 steps reach less than
    // one coarse sample either side; the backward kernel below relies on the same band): three terms per sum, not g.
    for (int idx = threadIdx.x; idx < g * S; idx += 256) {
        const int i = idx / S, x = idx - i * S;
        const int c = x >> 4;
        float a = 0.f;
#pragma unroll
        for (int k = -1; k <= 1; ++k) {                       // fixed trip count: all six LDS reads go out together
            const int j = c + k, jc = min(max(j, 0), g - 1);
            float l = lw[i * g + jc], u = Us[x * g + jc];
            asm volatile("" : "+v"(l), "+v"(u));
            a = j == jc ? fmaf(l, u, a) : a;
        }
        T[idx] = a;
    }
    __syncthreads();
    float* dst = heat + bh * (size_t)S * S;
    const int S4 = S >> 2;
    const int yh = S >> 1, y_lo = blockIdx.y * yh;           // two workgroups per map: the upper and the lower half
    const int step_y = 256 / S4, step_x = 256 - step_y * S4;  // the walk idx += 256 without a division per step
    int yy = threadIdx.x / S4, xq = threadIdx.x - yy * S4;
    for (; yy < yh; yy += step_y, xq += step_x) {
        if (xq >= S4) { xq -= S4; ++yy; if (yy >= yh) break; }
        const int x4 = xq * 4, y = y_lo + yy;
        const int c = y >> 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = -1; k <= 1; ++k) {
            const int i = c + k, ic = min(max(i, 0), g - 1);
            float u = Us[y * g + ic];
            float4 t = *(const float4*)(T + ic * S + x4);
            asm volatile("" : "+v"(u), "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));   // reads unconditional, then selects
            const bool in = i == ic;
            a.x = in ? fmaf(u, t.x, a.x) : a.x; a.y = in ? fmaf(u, t.y, a.y) : a.y;
            a.z = in ? fmaf(u, t.z, a.z) : a.z; a.w = in ? fmaf(u, t.w, a.w) : a.w;
        }
        // v_exp_f32 + v_rcp_f32 (1 ulp each; the argument's rounding adds |logit| * 6e-8 relative): the accurate expf
        // and the IEEE division were 90 of this loop's 200 instructions per four pixels, and the loop was VALU-bound
        a.x = __builtin_amdgcn_rcpf(1.0f + __expf(-a.x)); a.y = __builtin_amdgcn_rcpf(1.0f + __expf(-a.y));
        a.z = __builtin_amdgcn_rcpf(1.0f + __expf(-a.z)); a.w = __builtin_amdgcn_rcpf(1.0f + __expf(-a.w));
        *(float4*)(dst + (size_t)y * S + x4) = a;
    }
}

extern "C" int bf_decoder_upsample_sigmoid(const float* low, const float* U, float* heat, int n_maps,
                                           int grid, bf_stream_t stream) {
    BF_REQUIRE(n_maps > 0 && grid > 0 && grid <= DEC_MAXG, BF_EINVAL,
               "bf_decoder_upsample_sigmoid: grid must be in 1..%d (got %d)", DEC_MAXG, grid);
    BF_REQUIRE(bf_aligned16(heat), BF_EALIGN, "bf_decoder_upsample_sigmoid: unaligned heat");
    const int S = grid * 16;
    const size_t lds = (size_t)(S * grid + ((grid * grid + 3) & ~3) + grid * S) * sizeof(float);
    hipLaunchKernelGGL(upsample_sigmoid_kernel, dim3(n_maps, 2), dim3(256), lds, (hipStream_t)stream,
                       low, U, heat, grid);
    BF_CHECK_LAUNCH("bf_decoder_upsample_sigmoid");
    return BF_OK;
}

// ------------------------------------------------------------------ BCE forward
extern "C" int bf_bce_parts(int batch) {
    int p = batch * 8;
    return p > 1024 ? 1024 : (p < 1 ? 1 : p);
}

__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ heat,
                                                      const float* __restrict__ target,
                                                      float* __restrict__ loss_part, int batch, int pixels,
                                                      long hb, long hh, long tb, long th) {
    __shared__ float red[4];
    const int h = blockIdx.y;
    const int p4n = pixels >> 2;
    const size_t total = (size_t)batch * p4n;
    float acc = 0.f;
    // the grid-stride walk i += gridDim.x * 256 kept as (sample, quad) with one division up front: a 64-bit division
    // per step cost more than the two logarithms
    const size_t first = blockIdx.x * (size_t)256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    int b = (int)(first / p4n), q = (int)(first - (size_t)b * p4n);
    const int step_b = (int)(step / p4n), step_q = (int)(step - (size_t)step_b * p4n);
    for (size_t i = first; i < total; i += step, b += step_b, q += step_q) {
        if (q >= p4n) { q -= p4n; ++b; }
        const int p = q * 4;
        const float4 pv = *(const float4*)(heat + b * hb + h * hh + p);
        const float4 tv = *(const float4*)(target + b * tb + h * th + p);
        const float pp[4] = {pv.x, pv.y, pv.z, pv.w};
        const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lp = fmaxf(logf(pp[e]), -100.0f);
            const float l1 = fmaxf(logf(1.0f - pp[e]), -100.0f);
            acc -= tt[e] * lp + (1.0f - tt[e]) * l1;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[(size_t)h * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int bf_bce_fwd(const float* heat, const float* target, float* loss_part, int batch, int n_heads,
                          int pixels, long heat_bstride, long heat_hstride, long tgt_bstride,
                          long tgt_hstride, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_heads > 0 && pixels > 0 && (pixels & 3) == 0, BF_EINVAL, "bf_bce_fwd: bad shape");
    BF_REQUIRE((heat_bstride & 3) == 0 && (heat_hstride & 3) == 0 && (tgt_bstride & 3) == 0 &&
                   (tgt_hstride & 3) == 0 && bf_aligned16(heat) && bf_aligned16(target),
               BF_EALIGN, "bf_bce_fwd: strides must be multiples of 4 elements, pointers 16-byte aligned");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(bf_bce_parts(batch), n_heads), dim3(256), 0, (hipStream_t)stream,
                       heat, target, loss_part, batch, pixels, heat_bstride, heat_hstride, tgt_bstride,
                       tgt_hstride);
    BF_CHECK_LAUNCH("bf_bce_fwd");
    return BF_OK;
}

__global__ __launch_bounds__(64) void bce_finish_kernel(const float* __restrict__ loss_part,
                                                        float* __restrict__ loss, int n_part, float inv_count) {
    const int h = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 64) acc += loss_part[(size_t)h * n_part + i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[h] = acc * inv_count;
}

extern "C" int bf_bce_finish(const float* loss_part, float* loss, int n_heads, int n_part, float inv_count,
                             bf_stream_t stream) {
    BF_REQUIRE(n_heads > 0 && n_part > 0, BF_EINVAL, "bf_bce_finish: bad shape");
    hipLaunchKernelGGL(bce_finish_kernel, dim3(n_heads), dim3(64), 0, (hipStream_t)stream, loss_part, loss,
                       n_part, inv_count);
    BF_CHECK_LAUNCH("bf_bce_finish");
    return BF_OK;
}

// ------------------------------------------------------------------ BCE backward (elementwise)
// d_heat = gscale[0] * inv_count * (p - t) / max(p (1 - p), 1e-12)     (torch's BCELoss backward)
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ heat,
                                                      const float* __restrict__ target,
                                                      const float* __restrict__ gscale,
                                                      float* __restrict__ d_heat, size_t n4, float inv_count) {
    const float gs = gscale[0] * inv_count;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 p = ((const float4*)heat)[i];
        const float4 t = ((const float4*)target)[i];
        float4 d;
        d.x = gs * (p.x - t.x) / fmaxf(p.x * (1.0f - p.x), 1e-12f);
        d.y = gs * (p.y - t.y) / fmaxf(p.y * (1.0f - p.y), 1e-12f);
        d.z = gs * (p.z - t.z) / fmaxf(p.z * (1.0f - p.z), 1e-12f);
        d.w = gs * (p.w - t.w) / fmaxf(p.w * (1.0f - p.w), 1e-12f);
        ((float4*)d_heat)[i] = d;
    }
}

extern "C" int bf_bce_bwd(const float* heat, const float* target, const float* gscale, float* d_heat,
                          size_t count, float inv_count, bf_stream_t stream) {
    BF_REQUIRE(count > 0 && (count & 3) == 0, BF_EINVAL, "bf_bce_bwd: count must be a positive multiple of 4");
    BF_REQUIRE(bf_aligned16(heat) && bf_aligned16(target) && bf_aligned16(d_heat), BF_EALIGN,
               "bf_bce_bwd: unaligned pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, heat, target,
                       gscale, d_heat, n4, inv_count);
    BF_CHECK_LAUNCH("bf_bce_bwd");
    return BF_OK;
}

// ------------------------------------------------------------------ sigmoid o upsample backward
// per map: dlogit = d_heat * p (1 - p);  g_low = U^T dlogit U      (adjoint of the separable upsample)
#define UB_ROWS 16
#define UB_PF ((UB_ROWS * DEC_MAXG * 4 + 255) / 256)      // float4 per thread and operand of one 16-row chunk
__global__ __launch_bounds__(256) void upsample_sigmoid_bwd_kernel(const float* __restrict__ heat,
                                                                   const float* __restrict__ d_heat,
                                                                   const float* __restrict__ U,
                                                                   float* __restrict__ g_low, int g) {
    // per map: dl = d_heat * p(1-p);  R = dl U  (S x g);  g_low = U^T R  (g x g).
    // Rows are processed in chunks of 16 staged in LDS; thread (y, j) owns R[y][j] of the chunk
    // and thread (i, j) owns g_low[i][j], so there are no cross-lane reductions at all.
    // Two workgroups per map (blockIdx.y): each owns half of g_low's rows i and walks only the fine rows y those rows
    // can see -- U[y][i] is zero outside (i - 1) * 16 <= y < (i + 2) * 16 -- so every entry still has ONE owner that adds
    // its terms in ascending y (the skipped terms are exact zeros: same bits as the one-workgroup walk over all rows).
    // The next chunk's heat / d_heat rows are requested before the current chunk is reduced.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S = g * 16;
    float* Us = sm;                       // [S][g]
    float* dl = Us + S * g;               // [UB_ROWS][S]
    float* Rc = dl + UB_ROWS * S;         // [UB_ROWS][g]
    const size_t bh = blockIdx.x;
    const int gh = (g + 1) >> 1;
    const int i_lo = blockIdx.y * gh, i_hi = min(g, i_lo + gh);
    if (i_lo >= i_hi) return;             // g == 1: nothing for the second workgroup (uniform: before any barrier)
    const int y_beg = max(0, (i_lo - 1) * 16), y_end = min(S, (i_hi + 1) * 16);
    const float* hp = heat + bh * (size_t)S * S;
    const float* dp = d_heat + bh * (size_t)S * S;
    const int S4 = S >> 2, chunk4 = UB_ROWS * S4;
    int off[UB_PF];                       // element offset inside a chunk: the same in LDS and (row stride S) in memory
#pragma unroll
    for (int k = 0; k < UB_PF; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int y = idx / S4;
        off[k] = idx < chunk4 ? y * S + (idx - y * S4) * 4 : -1;
    }
    float4 pp[UB_PF], dd[UB_PF];
    auto fetch = [&](int y0) {
#pragma unroll
        for (int k = 0; k < UB_PF; ++k)
            if (off[k] >= 0) {
                pp[k] = *(const float4*)(hp + (size_t)y0 * S + off[k]);
                dd[k] = *(const float4*)(dp + (size_t)y0 * S + off[k]);
            }
    };
    fetch(y_beg);
    for (int i = threadIdx.x; i < S * g; i += 256) Us[i] = U[i];
    const int n_own = (i_hi - i_lo) * g;  // <= 12 * 24 entries over 256 threads
    float gacc[2] = {0.f, 0.f};
    for (int y0 = y_beg; y0 < y_end; y0 += UB_ROWS) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UB_PF; ++k)
            if (off[k] >= 0) {
                const float4 p = pp[k], d = dd[k];
                float4 o;
                o.x = d.x * (p.x * (1.0f - p.x)); o.y = d.y * (p.y * (1.0f - p.y));
                o.z = d.z * (p.z * (1.0f - p.z)); o.w = d.w * (p.w * (1.0f - p.w));
                *(float4*)(dl + off[k]) = o;
            }
        if (y0 + UB_ROWS < y_end) fetch(y0 + UB_ROWS);
        __syncthreads();
        for (int idx = threadIdx.x; idx < UB_ROWS * g; idx += 256) {
            const int y = idx / g, j = idx - y * g;
            // U[x][j] is non-zero only for x within one coarse cell of j (bilinear support)
            const int x_lo = max(0, (j - 1) * 16), x_hi = min(S, (j + 2) * 16);
            float a = 0.f;
#pragma unroll 8
            for (int x = x_lo; x < x_hi; ++x) a = fmaf(dl[y * S + x], Us[x * g + j], a);
            Rc[idx] = a;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ij = threadIdx.x + e * 256;
            if (ij < n_own) {
                const int i = i_lo + ij / g, j = ij % g;
                float a = gacc[e];
#pragma unroll
                for (int y = 0; y < UB_ROWS; ++y) a = fmaf(Us[(y0 + y) * g + i], Rc[y * g + j], a);
                gacc[e] = a;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ij = threadIdx.x + e * 256;
        if (ij < n_own) g_low[bh * g * g + i_lo * g + ij] = gacc[e];
    }
}

extern "C" int bf_decoder_upsample_sigmoid_bwd(const float* heat, const float* d_heat, const float* U,
                                               float* g_low, int n_maps, int grid, bf_stream_t stream) {
    BF_REQUIRE(n_maps > 0 && grid > 0 && grid <= DEC_MAXG, BF_EINVAL,
               "bf_decoder_upsample_sigmoid_bwd: grid must be in 1..%d", DEC_MAXG);
    const int S = grid * 16;
    const size_t lds = (size_t)(S * grid + UB_ROWS * S + UB_ROWS * grid) * sizeof(float);
    hipLaunchKernelGGL(upsample_sigmoid_bwd_kernel, dim3(n_maps, 2), dim3(256), lds, (hipStream_t)stream, heat,
                       d_heat, U, g_low, grid);
    BF_CHECK_LAUNCH("bf_decoder_upsample_sigmoid_bwd");
    return BF_OK;
}

// ------------------------------------------------------------------ token projection backward
#define PB_BLOCKS 1024
extern "C" size_t bf_decoder_project_bwd_ws(int batch, int n_heads, int dim) {
    (void)batch;
    return (size_t)PB_BLOCKS * n_heads * (dim + 4);
}

// stage 1: wave per token; d_tok rows written; per-block partial G / gb
// MH / MC: compile-time bounds on heads and on float4 chunks per lane (2 / 2 at the bench shape: 16 accumulator registers
// instead of 64, which is what lets four of these workgroups share a CU)
template <int MH, int MC, typename TD = float>
__global__ __launch_bounds__(256) void decoder_project_bwd_kernel(
    const float* __restrict__ tok, const float* __restrict__ w_eff, const float* __restrict__ g_low,
    TD* __restrict__ d_tok, float* __restrict__ part, int batch, int n_heads, int n_patch, int dim,
    long tok_bstride, long dtok_bstride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][n_heads][dim+4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = dim >> 2;
    const int L = n_patch + 1;
    float4 G[MH][MC];
    float gb[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        gb[h] = 0.f;
#pragma unroll
        for (int i = 0; i < MC; ++i) G[h][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int rows = batch * L;
    // two rows of the wave's walk per trip, both rows' loads requested before either is used (the walk's order, and with
    // it the order of the sums in G / gb, is unchanged)
    for (int row0 = blockIdx.x * 4 + wave; row0 < rows; row0 += gridDim.x * 8) {
        float4 x[2][MC];
        float gl[2][MH];
        TD* dr[2];
        int kind[2];                       // 0: no row, 1: the class token's row (zeros), 2: a patch row
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = row0 + r * gridDim.x * 4;
            kind[r] = 0;
            if (row >= rows) continue;
            const int b = row / L, t = row - b * L;
            dr[r] = d_tok + (size_t)b * dtok_bstride + (size_t)t * dim;
            kind[r] = t == 0 ? 1 : 2;
            if (t == 0) continue;
#pragma unroll
            for (int h = 0; h < MH; ++h)
                gl[r][h] = h < n_heads ? g_low[((size_t)h * batch + b) * n_patch + (t - 1)] : 0.f;
            const float* xr = tok + (size_t)b * tok_bstride + (size_t)t * dim;
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int c = lane + i * 64;
                if (c < nvec) x[r][i] = *(const float4*)(xr + c * 4);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (kind[r] == 1) {
                for (int c = lane; c < nvec; c += 64) Vec4<TD>::store(dr[r] + c * 4, make_float4(0.f, 0.f, 0.f, 0.f));
            } else if (kind[r] == 2) {
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    const int c = lane + i * 64;
                    if (c < nvec) {
                        const float4 xv = x[r][i];
                        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int h = 0; h < MH; ++h)
                            if (h < n_heads) {
                                const float4 w = *(const float4*)(w_eff + (size_t)h * dim + c * 4);
                                const float gh = gl[r][h];
                                d.x = fmaf(gh, w.x, d.x); d.y = fmaf(gh, w.y, d.y);
                                d.z = fmaf(gh, w.z, d.z); d.w = fmaf(gh, w.w, d.w);
                                G[h][i].x = fmaf(gh, xv.x, G[h][i].x); G[h][i].y = fmaf(gh, xv.y, G[h][i].y);
                                G[h][i].z = fmaf(gh, xv.z, G[h][i].z); G[h][i].w = fmaf(gh, xv.w, G[h][i].w);
                            }
                        Vec4<TD>::store(dr[r] + c * 4, d);
                    }
                }
#pragma unroll
                for (int h = 0; h < MH; ++h) gb[h] += gl[r][h];
            }
        }
    }
    const int stride = dim + 4;
#pragma unroll
    for (int h = 0; h < MH; ++h)
        if (h < n_heads) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int c = lane + i * 64;
                if (c < nvec) *(float4*)(sm + ((size_t)wave * n_heads + h) * stride + c * 4) = G[h][i];
            }
            if (lane == 0) sm[((size_t)wave * n_heads + h) * stride + dim] = gb[h];
        }
    __syncthreads();
    const int per = n_heads * stride;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int c = i % stride;
        if (c <= dim)
            part[(size_t)blockIdx.x * per + i] = (sm[i] + sm[per + i]) + (sm[2 * per + i] + sm[3 * per + i]);
    }
}
// stage 2: 64 outputs x 16 slices of the partials per workgroup; the slices are added in slice order (deterministic)
__global__ __launch_bounds__(1024) void decoder_project_bwd2_kernel(const float* __restrict__ part,
                                                                    float* __restrict__ g_w_eff,
                                                                    float* __restrict__ g_b_eff, int nblocks,
                                                                    int n_heads, int dim) {
    __shared__ float red[16][64];
    const int stride = dim + 4, per = n_heads * stride;
    const int o = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float acc = 0.f;
    if (i < per) {
#pragma unroll 4
        for (int k = ks; k < nblocks; k += 16) acc += part[(size_t)k * per + i];
    }
    red[ks][o] = acc;
    __syncthreads();
    if (ks != 0 || i >= per) return;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][o];
    const int h = i / stride, c = i - h * stride;
    if (c < dim) g_w_eff[(size_t)h * dim + c] = t;
    else if (c == dim) g_b_eff[h] = t;
}

extern "C" int bf_decoder_project_bwd(const float* tok, long tok_bstride, const float* w_eff,
                                      const float* g_low, void* d_tok, int dtok_dtype, long dtok_bstride, float* g_w_eff,
                                      float* g_b_eff, float* part_ws, int batch, int n_heads, int n_patch,
                                      int dim, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && n_heads >= 1 && n_heads <= DEC_MAXH && n_patch > 0 && (dim & 3) == 0 &&
                   dim <= DEC_MAXC * 256, BF_EINVAL, "bf_decoder_project_bwd: bad shape");
    BF_REQUIRE(bf_aligned16(tok) && bf_aligned16(w_eff) && bf_aligned16(d_tok) && bf_aligned16(part_ws) &&
                   (tok_bstride & 3) == 0 && (dtok_bstride & 3) == 0,
               BF_EALIGN, "bf_decoder_project_bwd: unaligned pointer or stride");
    hipStream_t s = (hipStream_t)stream;
    const int rows = batch * (n_patch + 1);
    int nblocks = bf_cdiv(rows, 4);
    if (nblocks > PB_BLOCKS) nblocks = PB_BLOCKS;
    const size_t lds = (size_t)4 * n_heads * (dim + 4) * sizeof(float);
    BF_REQUIRE(dtok_dtype == BF_F32 || dtok_dtype == BF_F16, BF_EINVAL,
               "bf_decoder_project_bwd: d_tok is fp32 or (the fp16 mode's 16-bit gradient stream) fp16, got %d", dtok_dtype);
#define PB_LAUNCH(MH, MC)                                                                                       \
    do {                                                                                                        \
        if (dtok_dtype == BF_F16)                                                                               \
            hipLaunchKernelGGL((decoder_project_bwd_kernel<MH, MC, f16_t>), dim3(nblocks), dim3(256), lds, s, tok, w_eff, \
                               g_low, (f16_t*)d_tok, part_ws, batch, n_heads, n_patch, dim, tok_bstride, dtok_bstride); \
        else                                                                                                    \
            hipLaunchKernelGGL((decoder_project_bwd_kernel<MH, MC>), dim3(nblocks), dim3(256), lds, s, tok, w_eff, g_low, \
                               (float*)d_tok, part_ws, batch, n_heads, n_patch, dim, tok_bstride, dtok_bstride); \
    } while (0)
    const bool few_heads = n_heads <= 2, narrow = dim <= 512;
    if (few_heads && narrow) PB_LAUNCH(2, 2);
    else if (few_heads) PB_LAUNCH(2, DEC_MAXC);
    else if (narrow) PB_LAUNCH(DEC_MAXH, 2);
    else PB_LAUNCH(DEC_MAXH, DEC_MAXC);
#undef PB_LAUNCH
    BF_CHECK_LAUNCH("bf_decoder_project_bwd/1");
    hipLaunchKernelGGL(decoder_project_bwd2_kernel, dim3(bf_cdiv(n_heads * (dim + 4), 64)), dim3(1024), 0, s,
                       part_ws, g_w_eff, g_b_eff, nblocks, n_heads, dim);
    BF_CHECK_LAUNCH("bf_decoder_project_bwd/2");
    return BF_OK;
}

// ================================================================== the mask head's product and the mask losses
// PickPlaceConvDecoder with compute_mask=True (reference models/pickplace.py:66-69,76-98,107-119): a further collapsed
// ConvDecoder reads the image tokens the fusion model was HANDED (bf_decoder_project / bf_decoder_upsample_sigmoid, one
// head), and every pick heatmap leaves as sigmoid(pick) * mask_heatmap.  The product and its backward are elementwise.
__global__ __launch_bounds__(256) void mask_mul_fwd_kernel(const float* __restrict__ s, const float* __restrict__ m,
                                                           float* __restrict__ out, int n_heads, size_t per4) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per4; i += (size_t)gridDim.x * 256) {
        const float4 mv = ((const float4*)m)[i];
        for (int h = 0; h < n_heads; ++h) {
            const float4 sv = ((const float4*)s)[h * per4 + i];
            ((float4*)out)[h * per4 + i] = make_float4(sv.x * mv.x, sv.y * mv.y, sv.z * mv.z, sv.w * mv.w);
        }
    }
}

extern "C" int bf_mask_mul_fwd(const float* s, const float* mask, float* out, int n_heads, size_t per_head,
                               bf_stream_t stream) {
    BF_REQUIRE(n_heads >= 1 && per_head > 0 && (per_head & 3) == 0, BF_EINVAL,
               "bf_mask_mul_fwd: per_head must be a positive multiple of 4");
    BF_REQUIRE(bf_aligned16(s) && bf_aligned16(mask) && bf_aligned16(out), BF_EALIGN, "bf_mask_mul_fwd: unaligned pointer");
    const size_t per4 = per_head / 4;
    size_t blocks = (per4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mask_mul_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, mask, out,
                       n_heads, per4);
    BF_CHECK_LAUNCH("bf_mask_mul_fwd");
    return BF_OK;
}

// d_s[h] = d_out[h] * m;   d_m = d_mask_in + sum_h d_out[h] * s[h]   (the sum only when the mask is not detached;
// d_out[h] NULL = a head the loss ignores; d_m NULL = nobody wants it)
struct MaskMulPtrs { const float* d_out[DEC_MAXH]; };
__global__ __launch_bounds__(256) void mask_mul_bwd_kernel(MaskMulPtrs dp, const float* __restrict__ s,
                                                           const float* __restrict__ m,
                                                           const float* __restrict__ d_mask_in,
                                                           float* __restrict__ d_s, float* __restrict__ d_m,
                                                           int n_heads, size_t per4, int through) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per4; i += (size_t)gridDim.x * 256) {
        const float4 mv = ((const float4*)m)[i];
        float4 dm = d_mask_in ? ((const float4*)d_mask_in)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int h = 0; h < n_heads; ++h) {
            float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dp.d_out[h]) go = ((const float4*)dp.d_out[h])[i];
            ((float4*)d_s)[h * per4 + i] = make_float4(go.x * mv.x, go.y * mv.y, go.z * mv.z, go.w * mv.w);
            if (through && dp.d_out[h]) {
                const float4 sv = ((const float4*)s)[h * per4 + i];
                dm.x = fmaf(go.x, sv.x, dm.x); dm.y = fmaf(go.y, sv.y, dm.y);
                dm.z = fmaf(go.z, sv.z, dm.z); dm.w = fmaf(go.w, sv.w, dm.w);
            }
        }
        if (d_m) ((float4*)d_m)[i] = dm;
    }
}

extern "C" int bf_mask_mul_bwd(const float* const* d_out, const float* s, const float* mask, const float* d_mask_in,
                               float* d_s, float* d_mask, int n_heads, size_t per_head, int detach,
                               bf_stream_t stream) {
    BF_REQUIRE(n_heads >= 1 && n_heads <= DEC_MAXH && per_head > 0 && (per_head & 3) == 0, BF_EINVAL,
               "bf_mask_mul_bwd: 1..%d heads, per_head a positive multiple of 4", DEC_MAXH);
    BF_REQUIRE(d_out && s && mask && d_s, BF_EINVAL, "bf_mask_mul_bwd: NULL operand");
    MaskMulPtrs dp;
    for (int h = 0; h < DEC_MAXH; ++h) {
        dp.d_out[h] = h < n_heads ? d_out[h] : nullptr;
        BF_REQUIRE(bf_aligned16(dp.d_out[h]), BF_EALIGN, "bf_mask_mul_bwd: unaligned d_out[%d]", h);
    }
    BF_REQUIRE(bf_aligned16(s) && bf_aligned16(mask) && bf_aligned16(d_mask_in) && bf_aligned16(d_s) &&
                   bf_aligned16(d_mask), BF_EALIGN, "bf_mask_mul_bwd: unaligned pointer");
    const size_t per4 = per_head / 4;
    size_t blocks = (per4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mask_mul_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dp, s, mask,
                       d_mask_in, d_s, d_mask, n_heads, per4, detach ? 0 : 1);
    BF_CHECK_LAUNCH("bf_mask_mul_bwd");
    return BF_OK;
}

// dst[b, i] += src[b, i], i < row_elems: the mask head's token gradient joins the gradient of the fusion INPUT's image
// rows (fp32, or the fp16 gradient stream of fp16 mode)
template <typename TD>
__global__ __launch_bounds__(256) void rows_accumulate_kernel(const float* __restrict__ src, TD* __restrict__ dst,
                                                              long src_bstride, long dst_bstride, int batch,
                                                              long row4) {
    const size_t total = (size_t)batch * row4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / row4, q = i - b * row4;
        const float4 a = *(const float4*)(src + b * src_bstride + q * 4);
        TD* d = dst + b * dst_bstride + q * 4;
        const float4 c = Vec4<TD>::load(d);
        Vec4<TD>::store(d, make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w));
    }
}

extern "C" int bf_rows_accumulate(const float* src, long src_bstride, void* dst, int dst_dtype, long dst_bstride,
                                  int batch, long row_elems, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && row_elems > 0 && (row_elems & 3) == 0, BF_EINVAL, "bf_rows_accumulate: bad shape");
    BF_REQUIRE(dst_dtype == BF_F32 || dst_dtype == BF_F16, BF_EINVAL, "bf_rows_accumulate: dst is fp32 or fp16, got %d",
               dst_dtype);
    BF_REQUIRE(bf_aligned16(src) && (src_bstride & 3) == 0 && (dst_bstride & 3) == 0 &&
                   (((uintptr_t)dst) & (dst_dtype == BF_F32 ? 15u : 7u)) == 0,
               BF_EALIGN, "bf_rows_accumulate: unaligned pointer or stride");
    const long row4 = row_elems / 4;
    size_t blocks = ((size_t)batch * row4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dst_dtype == BF_F16)
        hipLaunchKernelGGL((rows_accumulate_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                           (f16_t*)dst, src_bstride, dst_bstride, batch, row4);
    else
        hipLaunchKernelGGL((rows_accumulate_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                           (float*)dst, src_bstride, dst_bstride, batch, row4);
    BF_CHECK_LAUNCH("bf_rows_accumulate");
    return BF_OK;
}

// ------------------------------------------------------------------ DiceLoss (reference losses/__init__.py:126-136)
// per sample b: A = 2 sum(p t) + 1, D = sum p + sum t + 1, loss = sum_b (1 - A / D).  stats[b] = {sum p t, sum p, sum t, 0}
__global__ __launch_bounds__(256) void dice_stats_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                         float* __restrict__ stats, int pixels) {
    __shared__ float red[4][3];
    const int b = blockIdx.x;
    const float4* pp = (const float4*)(p + (size_t)b * pixels);
    const float4* tp = (const float4*)(t + (size_t)b * pixels);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < (pixels >> 2); i += 256) {
        const float4 a = pp[i], c = tp[i];
        s0 += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
        s1 += (a.x + a.y) + (a.z + a.w);
        s2 += (c.x + c.y) + (c.z + c.w);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0; red[threadIdx.x >> 6][1] = s1; red[threadIdx.x >> 6][2] = s2; }
    __syncthreads();
    if (threadIdx.x < 3)
        stats[(size_t)b * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 3) stats[(size_t)b * 4 + 3] = 0.f;
}
__global__ __launch_bounds__(64) void dice_finish_kernel(const float* __restrict__ stats, float* __restrict__ loss, int batch) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < batch; b += 64) {
        const float A = 2.0f * stats[(size_t)b * 4] + 1.0f, D = stats[(size_t)b * 4 + 1] + stats[(size_t)b * 4 + 2] + 1.0f;
        acc += 1.0f - A / D;
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[0] = acc;
}

extern "C" int bf_dice_fwd(const float* prob, const float* target, float* stats, float* loss, int batch, int pixels,
                           bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && pixels > 0 && (pixels & 3) == 0, BF_EINVAL, "bf_dice_fwd: bad shape");
    BF_REQUIRE(bf_aligned16(prob) && bf_aligned16(target) && stats && loss, BF_EALIGN, "bf_dice_fwd: unaligned / NULL pointer");
    hipLaunchKernelGGL(dice_stats_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, prob, target, stats, pixels);
    BF_CHECK_LAUNCH("bf_dice_fwd/1");
    hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, loss, batch);
    BF_CHECK_LAUNCH("bf_dice_fwd/2");
    return BF_OK;
}

// d loss / d p[b, j] = -(2 t_j D_b - A_b) / D_b^2
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ t, const float* __restrict__ stats,
                                                       const float* __restrict__ gscale, float* __restrict__ d_p,
                                                       int pixels) {
    const int b = blockIdx.y;
    const float A = 2.0f * stats[(size_t)b * 4] + 1.0f, D = stats[(size_t)b * 4 + 1] + stats[(size_t)b * 4 + 2] + 1.0f;
    const float g = gscale[0], c1 = -2.0f * g / D, c0 = g * A / (D * D);
    const float4* tp = (const float4*)(t + (size_t)b * pixels);
    float4* dp = (float4*)(d_p + (size_t)b * pixels);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (pixels >> 2); i += gridDim.x * 256) {
        const float4 c = tp[i];
        dp[i] = make_float4(fmaf(c1, c.x, c0), fmaf(c1, c.y, c0), fmaf(c1, c.z, c0), fmaf(c1, c.w, c0));
    }
}

extern "C" int bf_dice_bwd(const float* target, const float* stats, const float* gscale, float* d_prob, int batch,
                           int pixels, bf_stream_t stream) {
    BF_REQUIRE(batch > 0 && pixels > 0 && (pixels & 3) == 0, BF_EINVAL, "bf_dice_bwd: bad shape");
    BF_REQUIRE(bf_aligned16(target) && bf_aligned16(d_prob) && stats && gscale, BF_EALIGN,
               "bf_dice_bwd: unaligned / NULL pointer");
    int bx = bf_cdiv(pixels >> 2, 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(bx, batch), dim3(256), 0, (hipStream_t)stream, target, stats, gscale, d_prob,
                       pixels);
    BF_CHECK_LAUNCH("bf_dice_bwd");
    return BF_OK;
}

// ------------------------------------------------------------------ FocalLoss (reference losses/__init__.py:139-155)
// f = alpha_t * ce * (1 - p_t)^gamma,  ce = BCE with torch's -100 clamp of each log,  p_t = p t + (1 - p)(1 - t),
// alpha_t = alpha t + (1 - alpha)(1 - t) (alpha < 0: no weighting);  loss = sum f / rows  (`.mean(1).sum()` on [B, S, S])
__device__ __forceinline__ float focal_pow(float q, float gamma) {
    return gamma == 2.0f ? q * q : (gamma == 1.0f ? q : (gamma == 0.0f ? 1.0f : powf(q, gamma)));
}
__device__ __forceinline__ float focal_term(float p, float t, float alpha, float gamma) {
    const float ce = -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.0f));
    const float q = 1.0f - (p * t + (1.0f - p) * (1.0f - t));
    float f = ce * focal_pow(q, gamma);
    if (alpha >= 0.0f) f *= alpha * t + (1.0f - alpha) * (1.0f - t);
    return f;
}
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                        float* __restrict__ part, size_t n4, float alpha, float gamma) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = ((const float4*)p)[i], c = ((const float4*)t)[i];
        acc += (focal_term(a.x, c.x, alpha, gamma) + focal_term(a.y, c.y, alpha, gamma)) +
               (focal_term(a.z, c.z, alpha, gamma) + focal_term(a.w, c.w, alpha, gamma));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int bf_focal_fwd(const float* prob, const float* target, float* loss_part, float* loss, size_t count,
                            int n_part, float inv_rows, float alpha, float gamma, bf_stream_t stream) {
    BF_REQUIRE(count > 0 && (count & 3) == 0 && n_part >= 1 && n_part <= 1024, BF_EINVAL, "bf_focal_fwd: bad shape");
    BF_REQUIRE(bf_aligned16(prob) && bf_aligned16(target) && loss_part && loss, BF_EALIGN,
               "bf_focal_fwd: unaligned / NULL pointer");
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(n_part), dim3(256), 0, (hipStream_t)stream, prob, target, loss_part,
                       count / 4, alpha, gamma);
    BF_CHECK_LAUNCH("bf_focal_fwd");
    return bf_bce_finish(loss_part, loss, 1, n_part, inv_rows, stream);
}

// d f / d p = alpha_t [ ce' q^gamma + ce gamma q^(gamma-1) (1 - 2 t) ],  ce' = (p - t) / max(p (1 - p), 1e-12) (torch's
// binary_cross_entropy backward),  q = 1 - p_t,  d q / d p = 1 - 2 t
__device__ __forceinline__ float focal_grad(float p, float t, float alpha, float gamma) {
    const float ce = -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.0f));
    const float dce = (p - t) / fmaxf(p * (1.0f - p), 1e-12f);
    const float q = 1.0f - (p * t + (1.0f - p) * (1.0f - t));
    const float dq = gamma == 0.0f ? 0.0f : gamma * focal_pow(q, gamma - 1.0f);
    float d = dce * focal_pow(q, gamma) + ce * dq * (1.0f - 2.0f * t);
    if (alpha >= 0.0f) d *= alpha * t + (1.0f - alpha) * (1.0f - t);
    return d;
}
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                        const float* __restrict__ gscale, float* __restrict__ d_p,
                                                        size_t n4, float inv_rows, float alpha, float gamma) {
    const float gs = gscale[0] * inv_rows;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = ((const float4*)p)[i], c = ((const float4*)t)[i];
        ((float4*)d_p)[i] = make_float4(gs * focal_grad(a.x, c.x, alpha, gamma), gs * focal_grad(a.y, c.y, alpha, gamma),
                                        gs * focal_grad(a.z, c.z, alpha, gamma), gs * focal_grad(a.w, c.w, alpha, gamma));
    }
}

extern "C" int bf_focal_bwd(const float* prob, const float* target, const float* gscale, float* d_prob, size_t count,
                            float inv_rows, float alpha, float gamma, bf_stream_t stream) {
    BF_REQUIRE(count > 0 && (count & 3) == 0, BF_EINVAL, "bf_focal_bwd: count must be a positive multiple of 4");
    BF_REQUIRE(bf_aligned16(prob) && bf_aligned16(target) && bf_aligned16(d_prob) && gscale, BF_EALIGN,
               "bf_focal_bwd: unaligned / NULL pointer");
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prob, target, gscale,
                       d_prob, n4, inv_rows, alpha, gamma);
    BF_CHECK_LAUNCH("bf_focal_bwd");
    return BF_OK;
}
