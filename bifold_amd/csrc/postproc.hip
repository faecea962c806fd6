// Either side of the training step: heatmap -> pixel action on the way out
// (models/utils.py:60-83) and pixel -> Gaussian target heatmap on the way in
// (data/processor.py:405-426).  One workgroup per map; both are latency-, not bandwidth-bound
// (a 224x224 map is 200 KB), so the point is keeping them on the device: no per-sample
// D2H copy + numpy loop between the network and the action, no H2D of 4 x 200 KB targets
// per sample.
#include "common.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int PP_THREADS = 1024;
constexpr int PP_WAVES = PP_THREADS / 64;

// numpy argmax order: NaN beats everything, then the larger value, then the lower index.
// (-inf, INT_MAX) is the identity: any real element beats it.
__device__ inline bool arg_better(float av, int ai, float bv, int bi) {
    const bool an = av != av, bn = bv != bv;
    if (an || bn) return an && (!bn || ai < bi);
    return av > bv || (av == bv && ai < bi);
}

// (distance, index) minimum, lower index on ties (np.argwhere is row-major, argmin takes the first).
__device__ inline bool near_better(int ad, int ai, int bd, int bi) {
    return ad < bd || (ad == bd && ai < bi);
}

__global__ __launch_bounds__(PP_THREADS) void heatmap_pick_kernel(const float* __restrict__ heat,
                                                                  const float* __restrict__ mask,
                                                                  int* __restrict__ xy,
                                                                  float* __restrict__ conf, int size) {
    __shared__ float s_val[PP_WAVES];
    __shared__ int s_idx[PP_WAVES];
    __shared__ int s_dist[PP_WAVES];
    __shared__ int s_best;
    const int map = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pixels = size * size;
    const float* h = heat + (size_t)map * pixels;

    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < pixels; i += PP_THREADS) {
        const float v = h[i];
        if (arg_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < PP_WAVES; ++w)
            if (arg_better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        s_best = bi;
    }
    __syncthreads();
    int best = s_best;

    if (mask) {
        const float* m = mask + (size_t)map * pixels;
        const int u = best / size, v = best - u * size;
        int bd = 0x7fffffff, ni = 0x7fffffff;
        for (int i = tid; i < pixels; i += PP_THREADS) {
            if (m[i] != 0.0f) {
                const int r = i / size, c = i - r * size;
                const int d = (r - u) * (r - u) + (c - v) * (c - v);
                if (near_better(d, i, bd, ni)) { bd = d; ni = i; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int od = __shfl_xor(bd, off);
            const int oi = __shfl_xor(ni, off);
            if (near_better(od, oi, bd, ni)) { bd = od; ni = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_dist[wave] = bd; s_idx[wave] = ni; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < PP_WAVES; ++w)
                if (near_better(s_dist[w], s_idx[w], bd, ni)) { bd = s_dist[w]; ni = s_idx[w]; }
            s_best = ni;
        }
        __syncthreads();
        best = s_best;
    }
    if (tid == 0) {
        if (best == 0x7fffffff) {            // empty mask: the reference raises; the host checks for -2
            xy[map * 2 + 0] = -2; xy[map * 2 + 1] = -2; conf[map] = 0.0f;
        } else {
            const int r = best / size;
            xy[map * 2 + 0] = best - r * size;     // x = column
            xy[map * 2 + 1] = r;                   // y = row
            conf[map] = h[best];
        }
    }
}

__device__ inline double block_max(double v, double* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double r = s_red[0];
    for (int w = 1; w < PP_WAVES; ++w) r = fmax(r, s_red[w]);
    return r;
}

// fp64 like the reference's numpy (the maps are cast to fp32 at the end, processor.py:439).
__global__ __launch_bounds__(PP_THREADS) void gaussian_targets_kernel(const float* __restrict__ points,
                                                                      const int* __restrict__ counts,
                                                                      float* __restrict__ heat, int max_points,
                                                                      int size, double sigma, int strategy) {
    __shared__ double s_red[PP_WAVES];
    __shared__ double s_cx[BF_MAX_TARGET_POINTS], s_cy[BF_MAX_TARGET_POINTS];
    const int map = blockIdx.x, tid = threadIdx.x;
    const int pixels = size * size;
    float* out = heat + (size_t)map * pixels;
    int n = counts ? counts[map] : max_points;
    n = n < 0 ? 0 : (n > max_points ? max_points : n);
    if (n == 0) {
        for (int i = tid; i < pixels; i += PP_THREADS) out[i] = 0.0f;
        return;
    }
    if (tid < n) {       // python round() / np.round: half to even == rint in the default mode
        s_cx[tid] = rint((double)points[((size_t)map * max_points + tid) * 2 + 0]);
        s_cy[tid] = rint((double)points[((size_t)map * max_points + tid) * 2 + 1]);
    }
    __syncthreads();
    if (n > 1 && strategy == BF_TARGETS_FIT) {
        // scipy.stats.multivariate_normal(mean(points), cov(points, ddof = 1)).pdf on the pixel grid (processor.py:427-436):
        // the UNROUNDED points, an unnormalised density (no division by the maximum).  exp(-1/2 d^T S^-1 d) / (2 pi sqrt|S|)
        // in fp64; scipy's eigen-decomposition form differs from the closed 2 x 2 inverse in the last fp64 ulp only.
        // numpy semantics of the reference's two calls: the density's centre is np.mean of the fp32 points -- an fp32
        // sequential sum / n -- while np.cov promotes to fp64 and centres with its own fp64 mean
        float fx = 0.0f, fy = 0.0f;
        double ax = 0.0, ay = 0.0;
        for (int p = 0; p < n; ++p) {
            const float px = points[((size_t)map * max_points + p) * 2 + 0], py = points[((size_t)map * max_points + p) * 2 + 1];
            fx = __fadd_rn(fx, px); fy = __fadd_rn(fy, py);
            ax += (double)px; ay += (double)py;
        }
        const double mx = (double)__fdiv_rn(fx, (float)n), my = (double)__fdiv_rn(fy, (float)n);
        ax /= (double)n; ay /= (double)n;
        double sxx = 0.0, sxy = 0.0, syy = 0.0;
        for (int p = 0; p < n; ++p) {
            const double dx = (double)points[((size_t)map * max_points + p) * 2 + 0] - ax;
            const double dy = (double)points[((size_t)map * max_points + p) * 2 + 1] - ay;
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
        sxx /= (double)(n - 1); sxy /= (double)(n - 1); syy /= (double)(n - 1);
        const double det = sxx * syy - sxy * sxy;
        // a singular covariance (two points, collinear points) makes scipy raise: NaN here, so it is seen
        const double inv_det = det > 0.0 ? 1.0 / det : nan("");
        const double ixx = syy * inv_det, ixy = -sxy * inv_det, iyy = sxx * inv_det;
        const double norm = 1.0 / (2.0 * M_PI * sqrt(det));
        for (int i = tid; i < pixels; i += PP_THREADS) {
            const int r = i / size, c = i - r * size;
            const double dx = (double)c - mx, dy = (double)r - my;
            out[i] = (float)(exp(-0.5 * (dx * dx * ixx + 2.0 * dx * dy * ixy + dy * dy * iyy)) * norm);
        }
        return;
    }
    if (n == 1 || strategy == BF_TARGETS_FIRST) {
        const double cx = s_cx[0], cy = s_cy[0];
        for (int i = tid; i < pixels; i += PP_THREADS) {
            const int r = i / size, c = i - r * size;
            const double d = ((double)c - cx) * ((double)c - cx) + ((double)r - cy) * ((double)r - cy);
            out[i] = (float)exp(-d / (2.0 * sigma * sigma));
        }
        return;
    }
    // gmm: sum of normalised Gaussians, divided by its maximum
    const double norm = (2.0 * M_PI) * (sigma * sigma);
    double local_max = 0.0;
    for (int i = tid; i < pixels; i += PP_THREADS) {
        const int r = i / size, c = i - r * size;
        double acc = 0.0;
        for (int p = 0; p < n; ++p) {
            const double dx = (double)c - s_cx[p], dy = (double)r - s_cy[p];
            acc += exp(-(dx * dx + dy * dy) / (2.0 * (sigma * sigma))) / norm;
        }
        local_max = fmax(local_max, acc);
    }
    const double mx = block_max(local_max, s_red);
    for (int i = tid; i < pixels; i += PP_THREADS) {
        const int r = i / size, c = i - r * size;
        double acc = 0.0;
        for (int p = 0; p < n; ++p) {
            const double dx = (double)c - s_cx[p], dy = (double)r - s_cy[p];
            acc += exp(-(dx * dx + dy * dy) / (2.0 * (sigma * sigma))) / norm;
        }
        // max == 0 (every point far outside the map): the reference asserts; emit NaN so it is seen
        out[i] = (float)(acc / mx);
    }
}

// raw camera frames -> the network's input: mask fill (value 77, processor.py:257), uint8 -> [0,1]
// (v2.ToDtype(scale=True)), per-channel (x - mean) / std (v2.Normalize), HWC -> CHW.  For frames that
// already have the model's resolution (v2.Resize / CenterCrop are then the identity); resizing is not built.
__global__ __launch_bounds__(256) void preprocess_rgb_kernel(const uint8_t* __restrict__ raw, const float* __restrict__ mask,
                                                             float* __restrict__ out, int batch, int size, float m0,
                                                             float m1, float m2, float is0, float is1, float is2) {
    const size_t pixels = (size_t)size * size;
    const size_t total = (size_t)batch * pixels;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / pixels, p = i - b * pixels;
        const uint8_t* px = raw + i * 3;
        float r = px[0], g = px[1], bl = px[2];
        if (mask) {
            // (rgb * mask + (1 - mask) * 77).astype(uint8): the mask is 0/1 in every shipped dataset; a
            // fractional value truncates like numpy's cast
            const float mk = mask[i];
            r = floorf(r * mk + (1.0f - mk) * 77.0f);
            g = floorf(g * mk + (1.0f - mk) * 77.0f);
            bl = floorf(bl * mk + (1.0f - mk) * 77.0f);
        }
        float* o = out + b * 3 * pixels + p;
        o[0] = (r / 255.0f - m0) * is0;
        o[pixels] = (g / 255.0f - m1) * is1;
        o[2 * pixels] = (bl / 255.0f - m2) * is2;
    }
}

}  // namespace

extern "C" int bf_preprocess_rgb(const uint8_t* raw, const float* mask, float* out, int batch, int size,
                                 const float* mean3, const float* std3, bf_stream_t stream) {
    BF_REQUIRE(raw && out && mean3 && std3, BF_EINVAL, "bf_preprocess_rgb: null pointer");
    BF_REQUIRE(batch >= 0 && size > 0 && size <= 32768, BF_EINVAL, "bf_preprocess_rgb: bad shape");
    BF_REQUIRE(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, BF_EINVAL, "bf_preprocess_rgb: std must be positive");
    if (batch == 0) return BF_OK;
    const size_t total = (size_t)batch * size * size;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(preprocess_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, mask, out,
                       batch, size, mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2]);
    BF_CHECK_LAUNCH("bf_preprocess_rgb");
    return BF_OK;
}

extern "C" int bf_heatmap_pick(const float* heat, const float* mask, int32_t* xy, float* conf, int n_maps,
                               int size, bf_stream_t stream) {
    BF_REQUIRE(heat && xy && conf, BF_EINVAL, "bf_heatmap_pick: null pointer");
    BF_REQUIRE(n_maps >= 0 && size > 0 && size <= 32768 && (long long)size * size < 0x7fffffffLL, BF_EINVAL,
               "bf_heatmap_pick: bad shape (n_maps=%d size=%d)", n_maps, size);
    if (n_maps == 0) return BF_OK;
    hipLaunchKernelGGL(heatmap_pick_kernel, dim3(n_maps), dim3(PP_THREADS), 0, (hipStream_t)stream, heat, mask,
                       (int*)xy, conf, size);
    BF_CHECK_LAUNCH("bf_heatmap_pick");
    return BF_OK;
}

extern "C" int bf_gaussian_targets(const float* points, const int32_t* counts, float* heat, int n_maps,
                                   int max_points, int size, double sigma, int strategy, bf_stream_t stream) {
    BF_REQUIRE(points && heat, BF_EINVAL, "bf_gaussian_targets: null pointer");
    BF_REQUIRE(n_maps >= 0 && size > 0 && size <= 32768, BF_EINVAL,
               "bf_gaussian_targets: bad shape (n_maps=%d size=%d)", n_maps, size);
    BF_REQUIRE(max_points >= 1 && max_points <= BF_MAX_TARGET_POINTS, BF_EINVAL,
               "bf_gaussian_targets: max_points must be in [1, %d] (got %d)", BF_MAX_TARGET_POINTS, max_points);
    BF_REQUIRE(sigma > 0.0, BF_EINVAL, "bf_gaussian_targets: sigma must be positive");
    BF_REQUIRE(strategy == BF_TARGETS_FIRST || strategy == BF_TARGETS_GMM || strategy == BF_TARGETS_FIT, BF_EINVAL,
               "bf_gaussian_targets: unknown strategy %d (first=0, gmm=1, fit=2)", strategy);
    if (n_maps == 0) return BF_OK;
    hipLaunchKernelGGL(gaussian_targets_kernel, dim3(n_maps), dim3(PP_THREADS), 0, (hipStream_t)stream, points,
                       (const int*)counts, heat, max_points, size, sigma, strategy);
    BF_CHECK_LAUNCH("bf_gaussian_targets");
    return BF_OK;
}
