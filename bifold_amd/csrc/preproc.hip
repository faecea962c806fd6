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
