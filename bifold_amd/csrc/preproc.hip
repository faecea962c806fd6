// Input side of the step on the device: what the reference's Processor does on DataLoader
// workers with torchvision (bifold/data/processor.py:45-95, 302-342) -- Resize(bicubic, antialias) + CenterCrop +
// scale + Normalize of camera frames, and the spatial augmentation's affine warp.  HBM-bound byte work.
#include "common.h"

#include <math.h>

namespace {

// Keys' cubic with a = -0.5: the kernel of PIL / torch's antialiased bicubic resize.
__device__ __forceinline__ float cubic_aa(float x) {
    x = fabsf(x);
    if (x < 1.0f) return ((1.5f * x - 2.5f) * x) * x + 1.0f;
    if (x < 2.0f) return (((-0.5f * x + 2.5f) * x - 4.0f) * x + 2.0f);
    return 0.0f;
}

// The taps of output index `o` along one axis (PIL's precompute_coeffs / torch's _compute_indices_weights_aa):
// centre = (o + 0.5) * scale, support = 2 * max(scale, 1); taps [lo, lo + n) get
// cubic((j + lo - centre + 0.5) / max(scale, 1)) normalised to sum 1.
struct Taps { int lo, n; float inv_fs, centre; };
__device__ __forceinline__ Taps taps_for(int o, float scale, int in_size) {
    const float fs = fmaxf(scale, 1.0f);
    const float support = 2.0f * fs;
    const float centre = ((float)o + 0.5f) * scale;
    int lo = (int)(centre - support + 0.5f);
    if (lo < 0) lo = 0;
    int hi = (int)(centre + support + 0.5f);
    if (hi > in_size) hi = in_size;
    Taps t;
    t.lo = lo; t.n = hi - lo; t.inv_fs = 1.0f / fs; t.centre = centre;
    return t;
}
__device__ __forceinline__ float tap_weight(const Taps& t, int j) {
    return cubic_aa(((float)(j + t.lo) - t.centre + 0.5f) * t.inv_fs);
}
__device__ __forceinline__ float round_u8(float v) { return fminf(255.0f, fmaxf(0.0f, rintf(v))); }

// pass 1: rows stay, columns are filtered.  raw [B, H, W, 3] u8 (+ mask [B, H, W]: masked-out pixels read as 77,
// processor.py:255-258) -> tmp [B, H, wo, 3] u8 holding output columns [x0, x0 + wo) of the resized width.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ raw, const float* __restrict__ mask,
                                                       uint8_t* __restrict__ tmp, int batch, int H, int W, int wo, int x0,
                                                       float scale) {
    const size_t total = (size_t)batch * H * wo;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % wo);
        const size_t row = i / wo;                     // b * H + y
        const Taps t = taps_for(x0 + x, scale, W);
        const uint8_t* src = raw + row * (size_t)W * 3;
        const float* mrow = mask ? mask + row * (size_t)W : nullptr;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
        for (int j = 0; j < t.n; ++j) {
            const float w = tap_weight(t, j);
            const uint8_t* p = src + (size_t)(t.lo + j) * 3;
            float r = p[0], g = p[1], b = p[2];
            if (mrow) {
                const float mk = mrow[t.lo + j];
                r = floorf(r * mk + (1.0f - mk) * 77.0f);
                g = floorf(g * mk + (1.0f - mk) * 77.0f);
                b = floorf(b * mk + (1.0f - mk) * 77.0f);
            }
            a0 = fmaf(w, r, a0); a1 = fmaf(w, g, a1); a2 = fmaf(w, b, a2); ws += w;
        }
        const float inv = 1.0f / ws;
        uint8_t* d = tmp + i * 3;
        d[0] = (uint8_t)round_u8(a0 * inv); d[1] = (uint8_t)round_u8(a1 * inv); d[2] = (uint8_t)round_u8(a2 * inv);
    }
}

// pass 2: columns stay, rows are filtered; crop rows [y0, y0 + S); round to u8 like the reference's uint8 image,
// then v2.ToDtype(scale=True) and v2.Normalize; HWC -> CHW.
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, int batch,
                                                       int H, int S, int y0, float scale, float m0, float m1, float m2,
                                                       float is0, float is1, float is2) {
    const size_t pixels = (size_t)S * S;
    const size_t total = (size_t)batch * pixels;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / pixels;
        const int p = (int)(i - b * pixels), y = p / S, x = p - y * S;
        const Taps t = taps_for(y0 + y, scale, H);
        const uint8_t* src = tmp + (b * H * (size_t)S + x) * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ws = 0.f;
        for (int j = 0; j < t.n; ++j) {
            const float w = tap_weight(t, j);
            const uint8_t* q = src + (size_t)(t.lo + j) * S * 3;
            a0 = fmaf(w, (float)q[0], a0); a1 = fmaf(w, (float)q[1], a1); a2 = fmaf(w, (float)q[2], a2); ws += w;
        }
        const float inv = 1.0f / ws;
        float* o = out + b * 3 * pixels + p;
        o[0] = (round_u8(a0 * inv) / 255.0f - m0) * is0;
        o[pixels] = (round_u8(a1 * inv) / 255.0f - m1) * is1;
        o[2 * pixels] = (round_u8(a2 * inv) / 255.0f - m2) * is2;
    }
}

// v2.functional.affine(img, angle, translate=[dx, dy], scale=1, shear=0) with its defaults: nearest
// interpolation, zero fill, rotation about the image centre (processor.py:333-341).  params [N, 3] = (angle in
// degrees, dx, dy) per image; every channel of image n shares them.
__global__ __launch_bounds__(256) void affine_nearest_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ params, int n_img, int chans,
                                                             int H, int W) {
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)n_img * chans * plane;
    const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / (chans * plane);
        const int p = (int)(i % plane), y = p / W, x = p - y * W;
        const float ang = params[img * 3] * 0.017453292519943295f;
        const float tx = params[img * 3 + 1], ty = params[img * 3 + 2];
        const float cs = cosf(ang), sn = sinf(ang);
        // inverse map (output -> source), centred coordinates: [cos sin; -sin cos] (p - t)
        const float m2 = -cs * tx - sn * ty, m5 = sn * tx - cs * ty;
        const float xb = (float)x - cx, yb = (float)y - cy;
        const float xs = (cs * xb + sn * yb + m2) + cx;
        const float ys = (-sn * xb + cs * yb + m5) + cy;
        const int ix = (int)rintf(xs), iy = (int)rintf(ys);      // grid_sample(nearest): round half to even
        float v = 0.0f;
        if (ix >= 0 && ix < W && iy >= 0 && iy < H) v = in[(i - p) + (size_t)iy * W + ix];
        out[i] = v;
    }
}

}  // namespace

extern "C" int bf_resize_normalize_rgb(const uint8_t* raw, const float* mask, uint8_t* tmp_ws, float* out, int batch,
                                       int height, int width, int size, const float* mean3, const float* std3,
                                       bf_stream_t stream) {
    BF_REQUIRE(raw && tmp_ws && out && mean3 && std3, BF_EINVAL, "bf_resize_normalize_rgb: null pointer");
    BF_REQUIRE(batch >= 0 && height > 0 && width > 0 && size > 0 && height <= 16384 && width <= 16384 && size <= 16384,
               BF_EINVAL, "bf_resize_normalize_rgb: bad shape");
    BF_REQUIRE(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, BF_EINVAL, "bf_resize_normalize_rgb: std must be positive");
    if (batch == 0) return BF_OK;
    // v2.Resize(size): the SHORTER side becomes `size`, the other int(size * long / short); then CenterCrop(size)
    const int short_side = height < width ? height : width;
    const int new_h = height <= width ? size : (int)((long long)size * height / width);
    const int new_w = width <= height ? size : (int)((long long)size * width / height);
    (void)short_side;
    const int y0 = (int)lrintf((float)(new_h - size) / 2.0f), x0 = (int)lrintf((float)(new_w - size) / 2.0f);
    const float sx = (float)width / (float)new_w, sy = (float)height / (float)new_h;
    hipStream_t s = (hipStream_t)stream;
    size_t t1 = (size_t)batch * height * size, b1 = (t1 + 255) / 256;
    if (b1 > 16384) b1 = 16384;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)b1), dim3(256), 0, s, raw, mask, tmp_ws, batch, height, width, size,
                       x0, sx);
    BF_CHECK_LAUNCH("bf_resize_normalize_rgb/h");
    size_t t2 = (size_t)batch * size * size, b2 = (t2 + 255) / 256;
    if (b2 > 16384) b2 = 16384;
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)b2), dim3(256), 0, s, tmp_ws, out, batch, height, size, y0, sy,
                       mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2]);
    BF_CHECK_LAUNCH("bf_resize_normalize_rgb/v");
    return BF_OK;
}

extern "C" int bf_affine_nearest(const float* in, float* out, const float* params, int n_images, int channels,
                                 int height, int width, bf_stream_t stream) {
    BF_REQUIRE(in && out && params && in != out, BF_EINVAL, "bf_affine_nearest: null or aliased pointers");
    BF_REQUIRE(n_images >= 0 && channels > 0 && height > 0 && width > 0, BF_EINVAL, "bf_affine_nearest: bad shape");
    if (n_images == 0) return BF_OK;
    const size_t total = (size_t)n_images * channels * height * width;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(affine_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, params,
                       n_images, channels, height, width);
    BF_CHECK_LAUNCH("bf_affine_nearest");
    return BF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Depth frames (the rest of it): the reference's depth transform chain on DataLoader workers
// (bifold/data/processor.py:59-78) = DepthScale (+ one random shift per frame, data/depth_augmentations.py:35-43),
// MaskDepth (x mask, :46-51), [torchvision Resize / CenterCrop: the identity at the model's own resolution, which is
// what this path takes], ToDtype(float32), TruncatedDepthStandardization (:6-17): sort the frame, drop the lowest and
// highest `thresh` share, subtract the mean of the rest and divide by sqrt(its unbiased variance + 1e-6).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void depth_prepare_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int batch, int pixels) {
    const size_t total = (size_t)batch * pixels;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float v = depth[i];
        if (shift) v += shift[i / pixels];
        if (mask) v *= mask[i];
        out[i] = v;
    }
}

// order-preserving map of fp32 bit patterns onto unsigned integers (-0 < +0 like the bit patterns sort; torch.sort
// treats them as equal, which cannot change a mean or a variance)
__device__ __forceinline__ unsigned depth_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float depth_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One workgroup per frame.  The two cut VALUES (ranks lo and hi - 1 of the sorted frame) come from a radix select --
// four passes of 8 bits over the frame, both ranks in the same pass -- instead of a sort: the trimmed statistics need
// the values between the cuts and HOW MANY copies of each cut value fall inside [lo, hi) (masked depth frames are
// mostly exact zeros, so the cuts usually sit inside a run of ties).  Sums in fp64, so the result does not depend on
// the summation order.
#define DEPTH_WG 1024
// (x and y carry no __restrict__: the header documents in-place use, x == y, which is what bifold_amd/hip/ops.py does)
__global__ __launch_bounds__(DEPTH_WG) void depth_standardize_kernel(const float* x, float* y,
                                                                     float* __restrict__ stats, int n, int lo, int hi) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel_prefix[2], sel_rank[2];
    __shared__ unsigned long long cnt[4];
    __shared__ double red[DEPTH_WG / 64];
    __shared__ double bc[2];
    const float* xf = x + (size_t)blockIdx.x * n;
    float* yf = y + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    if (tid == 0) { sel_prefix[0] = sel_prefix[1] = 0u; sel_rank[0] = (unsigned)lo; sel_rank[1] = (unsigned)(hi - 1); }
    unsigned maskbits = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) { hist[0][tid] = 0u; hist[1][tid] = 0u; }
        __syncthreads();
        const unsigned p0 = sel_prefix[0], p1 = sel_prefix[1];
        for (int i = tid; i < n; i += DEPTH_WG) {
            const unsigned k = depth_key(xf[i]);
            const unsigned b = (k >> shift) & 255u;
            if ((k & maskbits) == p0) atomicAdd(&hist[0][b], 1u);
            if ((k & maskbits) == p1) atomicAdd(&hist[1][b], 1u);
        }
        __syncthreads();
        if (tid < 2) {                       // one thread per rank walks its 256 bins
            unsigned r = sel_rank[tid], b = 0;
            for (; b < 255u; ++b) {
                const unsigned c = hist[tid][b];
                if (r < c) break;
                r -= c;
            }
            sel_rank[tid] = r;
            sel_prefix[tid] |= b << shift;
        }
        maskbits |= 255u << shift;
        __syncthreads();
    }
    const unsigned k_lo = sel_prefix[0], k_hi = sel_prefix[1];
    const float v_lo = depth_unkey(k_lo), v_hi = depth_unkey(k_hi);
    // counts below / at the cuts, sum of the values strictly between them
    if (tid < 4) cnt[tid] = 0ull;
    __syncthreads();
    unsigned lt_lo = 0, eq_lo = 0, lt_hi = 0;
    double s = 0.0;
    for (int i = tid; i < n; i += DEPTH_WG) {
        const float v = xf[i];
        const unsigned k = depth_key(v);
        lt_lo += k < k_lo; eq_lo += k == k_lo; lt_hi += k < k_hi;
        if (k > k_lo && k < k_hi) s += (double)v;
    }
    atomicAdd(&cnt[0], (unsigned long long)lt_lo);
    atomicAdd(&cnt[1], (unsigned long long)eq_lo);
    atomicAdd(&cnt[2], (unsigned long long)lt_hi);
    auto block_sum = [&](double v) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double t = 0.0;
        for (int w = 0; w < DEPTH_WG / 64; ++w) t += red[w];
        return t;
    };
    const double mid = block_sum(s);
    const long long nin = (long long)hi - lo;
    long long c_lo, c_hi;
    if (k_lo == k_hi) { c_lo = nin; c_hi = 0; }
    else {
        const long long le_lo = (long long)cnt[0] + (long long)cnt[1];
        c_lo = (le_lo < hi ? le_lo : hi) - lo;
        c_hi = (long long)hi - (long long)cnt[2];
    }
    const double mean = (mid + (double)c_lo * (double)v_lo + (double)c_hi * (double)v_hi) / (double)nin;
    double q = 0.0;
    for (int i = tid; i < n; i += DEPTH_WG) {
        const float v = xf[i];
        const unsigned k = depth_key(v);
        if (k > k_lo && k < k_hi) { const double d = (double)v - mean; q += d * d; }
    }
    double ss = block_sum(q);
    ss += (double)c_lo * ((double)v_lo - mean) * ((double)v_lo - mean) + (double)c_hi * ((double)v_hi - mean) * ((double)v_hi - mean);
    const double var = nin > 1 ? ss / (double)(nin - 1) : 0.0;          // torch.var: unbiased
    if (tid == 0) { bc[0] = mean; bc[1] = 1.0 / sqrt(var + 1e-6); }
    __syncthreads();
    const float mf = (float)bc[0], rf = (float)bc[1];
    for (int i = tid; i < n; i += DEPTH_WG) yf[i] = (xf[i] - mf) * rf;
    if (stats && tid == 0) { stats[blockIdx.x * 2] = mf; stats[blockIdx.x * 2 + 1] = (float)var; }
}

}  // namespace

extern "C" int bf_depth_prepare(const float* depth, const float* mask, const float* shift, float* out, int batch,
                                int pixels, bf_stream_t stream) {
    BF_REQUIRE(depth && out, BF_EINVAL, "bf_depth_prepare: null pointer");
    BF_REQUIRE(batch >= 0 && pixels > 0, BF_EINVAL, "bf_depth_prepare: bad shape");
    if (batch == 0) return BF_OK;
    const size_t total = (size_t)batch * pixels;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(depth_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, depth, mask, shift,
                       out, batch, pixels);
    BF_CHECK_LAUNCH("bf_depth_prepare");
    return BF_OK;
}

extern "C" int bf_depth_standardize(const float* x, float* y, float* stats, int batch, int pixels, int lo, int hi,
                                    bf_stream_t stream) {
    BF_REQUIRE(x && y, BF_EINVAL, "bf_depth_standardize: null pointer");
    BF_REQUIRE(batch >= 0 && pixels > 0 && lo >= 0 && hi <= pixels && lo < hi, BF_EINVAL,
               "bf_depth_standardize: need 0 <= lo < hi <= pixels (%d %d %d)", lo, hi, pixels);
    if (batch == 0) return BF_OK;
    hipLaunchKernelGGL(depth_standardize_kernel, dim3((unsigned)batch), dim3(DEPTH_WG), 0, (hipStream_t)stream, x, y, stats,
                       pixels, lo, hi);
    BF_CHECK_LAUNCH("bf_depth_standardize");
    return BF_OK;
}
