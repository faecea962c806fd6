// The LoRA adapter branch of an adapted Linear, UNMERGED (training with lora_dropout > 0: PEFT cannot fold
// W + (alpha / r) B A into W because the adapters see dropout(x) and the base weight sees x;
// reference bifold/models/siglip.py:33-40 -> peft LoraLayer.forward: result + lora_B(lora_A(dropout(x))) * scaling,
// conf/model/siglip_sequential.yaml:10-14: r = 8, lora_alpha = 32, lora_dropout = 0.01, target_modules [q_proj, v_proj]).
//
// Rounds 3 - 5 ran this branch as generic GEMMs around bf_dropout: per adapted projection a masked copy of the
// [rows, width] activation, an N = r GEMM, a K = r GEMM, and in the backward two more of each plus the masked copy again.
// An N = 8 product has no tile the persistent MFMA GEMM can use (it fell to the round-1 fallback kernel: 580 us for
// 368 640 x 768 x 8) and every pass re-read or re-wrote a [rows, width] tensor: 4.7 ms per vision layer at configs[2].
// These kernels do the branch at the traffic it needs -- each wide tensor is read once per pass, all adapted projections
// of a layer in one launch, masks regenerated in registers (common.h: one hash per four elements):
//
//   bf_lora_down   u[rows, S r]  = alpha [ (D_s . X_s) Wt_s^T ]_s         X_s = x + col_off[s]; Wt [S r, width]
//                  forward:  X = h1 (one read for all s), Wt = A, D_s the masks, alpha = 1 / (1 - p)
//                  backward: X_s = the dY slice of projection s, Wt = scaling B^T, no mask   (du = scaling dY B)
//   bf_lora_up     y[rows, off_s + c] += alpha D_s . (u_s Wt_s)
//                  forward:  y = the packed QKV output, off_s = the projection's column block, Wt = scaling B^T, no mask
//                  backward: y = dh1 (all s into the same columns), Wt = A, D_s the masks, alpha = 1 / (1 - p)
//   bf_lora_tn     out_s = alpha U_s^T (D_s . X_s)      [r, width]  (or transposed: [width, r])
//                  dA_s = du_s^T (D_s . h1) / (1 - p);  dB_s = scaling dY_s^T u_s (transposed, no mask)
//
// bf_lora_down is an MFMA kernel (rows on the accumulator's columns: the reduction runs along the contiguous dimension, so
// operand fragments come straight from global memory); bf_lora_up / bf_lora_tn are vector-unit kernels: a lane owns four
// columns, the 4-byte u values of a row are wave-uniform (scalar loads), 8 FMAs per element and projection.  All three
// are bounded by HBM or by the mask hash, not by the FMA count.
#include "common.h"

#ifndef LORA_KU
#define LORA_KU 2            // bf_lora_down: k-steps (16 columns each) fetched before the first is used
#endif
#ifndef LORA_UN
#define LORA_UN 2            // bf_lora_up / bf_lora_tn: rows fetched before the first is used
#endif

namespace {

struct LoraDev {                       // device copy of bf_lora_slots' per-slot fields
    int col_off[3];
    unsigned long long seed[3];
};

template <typename HT> __device__ __forceinline__ float4 half4_to_f32(uint2 w);
template <> __device__ __forceinline__ float4 half4_to_f32<bf16_t>(uint2 w) {
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xFFFF0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xFFFF0000u));
}
template <> __device__ __forceinline__ float4 half4_to_f32<f16_t>(uint2 w) {
    const f16x4 h = __builtin_bit_cast(f16x4, w);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
// four columns of a lane as two packed pairs (v_pk_fma_f32: two FMAs per issue slot)
struct Quad { f32x2 lo, hi; };
__device__ __forceinline__ Quad quad_zero() { Quad q; q.lo = (f32x2){0.f, 0.f}; q.hi = (f32x2){0.f, 0.f}; return q; }
__device__ __forceinline__ Quad quad_of(float4 v, float a) { Quad q; q.lo = (f32x2){v.x * a, v.y * a}; q.hi = (f32x2){v.z * a, v.w * a}; return q; }
__device__ __forceinline__ void quad_fma(Quad& acc, float u, const Quad& x) {
    const f32x2 u2 = {u, u};
    acc.lo = __builtin_elementwise_fma(u2, x.lo, acc.lo);
    acc.hi = __builtin_elementwise_fma(u2, x.hi, acc.hi);
}
// zero the columns whose 16-bit AND-mask (dropout_mask4) is clear
__device__ __forceinline__ void quad_mask(Quad& q, uint2 m) {
    q.lo[0] = __uint_as_float(__float_as_uint(q.lo[0]) & (unsigned)((int)(m.x << 16) >> 31));
    q.lo[1] = __uint_as_float(__float_as_uint(q.lo[1]) & (unsigned)((int)m.x >> 31));
    q.hi[0] = __uint_as_float(__float_as_uint(q.hi[0]) & (unsigned)((int)(m.y << 16) >> 31));
    q.hi[1] = __uint_as_float(__float_as_uint(q.hi[1]) & (unsigned)((int)m.y >> 31));
}
__device__ __forceinline__ float4 quad_f4(const Quad& q) { return make_float4(q.lo[0], q.lo[1], q.hi[0], q.hi[1]); }

template <typename HT> __device__ __forceinline__ uint2 f32_to_half4(float4 v) {
    typename H16<HT>::x4 h;
    h[0] = (HT)v.x; h[1] = (HT)v.y; h[2] = (HT)v.z; h[3] = (HT)v.w;
    return __builtin_bit_cast(uint2, h);
}

// ----------------------------------------------------------------------------------------------------------------------
// bf_lora_down.  One wave = 32 rows.  D[i][j] = sum_k Wt'[i][k] X^T[k][j]: i = adapter column (S r <= 32, the rest zero),
// j = row, k = 16 columns per MFMA.  Lane (j = lane % 32, hh = lane / 32) loads the 16 bytes X[row j][k0 + 8 hh ..] and
// reads Wt'[i = lane % 32][k0 + 8 hh ..] from LDS; for projection s only the lanes whose i belongs to s read Wt, the others
// read a zero row, so that one accumulator holds every projection's r columns side by side although each sees its own
// masked (or its own slice of) X.
// ----------------------------------------------------------------------------------------------------------------------
template <typename HT, int NS, bool MASK, bool SHARED>
__global__ __launch_bounds__(256) void lora_down_kernel(const HT* __restrict__ x, int ldx, const HT* __restrict__ wt,
                                                        float* __restrict__ u, HT* __restrict__ u_half, int rows, int width,
                                                        int rank, LoraDev sl, unsigned thr_m1, float alpha) {
    typedef typename H16<HT>::x8 x8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nsr = NS * rank;
    const int lrow = width * 2 + 16;                              // +16 bytes: consecutive rows start 4 banks apart
    {   // Wt and a zero row -> LDS
        const int pieces = width / 8;                             // 16-byte pieces per row
        for (int t = threadIdx.x; t < (nsr + 1) * pieces; t += 256) {
            const int r = t / pieces, c = t - r * pieces;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nsr) v = *(const uint4*)(wt + (size_t)r * width + c * 8);
            *(uint4*)(lds + r * lrow + c * 16) = v;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 32;
    if (row0 >= rows) return;
    const int row = min(row0 + j, rows - 1);                      // rows past the end re-read the last one; never stored
    const HT* xr = x + (size_t)row * ldx + 8 * hh;
    const unsigned long long idx0 = (unsigned long long)row * (unsigned)width + (unsigned)(8 * hh);
    int woff[NS];                                                 // this lane's Wt row for projection s (or the zero row)
#pragma unroll
    for (int s = 0; s < NS; ++s) woff[s] = ((j / rank == s) ? j : nsr) * lrow + 16 * hh;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int NL = SHARED ? 1 : NS;                           // distinct wide operands
    auto fetch = [&](uint4 (&raw)[NL], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NL; ++s) raw[s] = *(const uint4*)(xr + sl.col_off[s] + ks * 16);
    };
    auto kstep = [&](const uint4 (&raw)[NL], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            uint4 v = raw[SHARED ? 0 : s];
            if (MASK) {
                const uint2 m0 = dropout_mask4(idx0 + (unsigned)(ks * 16), sl.seed[s], thr_m1);
                const uint2 m1 = dropout_mask4(idx0 + (unsigned)(ks * 16 + 4), sl.seed[s], thr_m1);
                v.x &= m0.x; v.y &= m0.y; v.z &= m1.x; v.w &= m1.y;
            }
            const x8 wf = *(const x8*)(lds + woff[s] + ks * 32);
            acc = mfma_32x32x16(wf, __builtin_bit_cast(x8, v), acc);
        }
    };
    const int ksteps = width / 16;
    int ks = 0;
    for (; ks + LORA_KU <= ksteps; ks += LORA_KU) {               // LORA_KU steps of loads in flight before the first use
        uint4 raw[LORA_KU][NL];
#pragma unroll
        for (int t = 0; t < LORA_KU; ++t) fetch(raw[t], ks + t);
#pragma unroll
        for (int t = 0; t < LORA_KU; ++t) kstep(raw[t], ks + t);
    }
    for (; ks < ksteps; ++ks) {
        uint4 raw[NL];
        fetch(raw, ks);
        kstep(raw, ks);
    }
    if (row0 + j < rows) {
        float* ur = u + (size_t)row * nsr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i0 = 8 * q + 4 * hh;
            if (i0 < nsr) {
                const float4 v = make_float4(acc[4 * q] * alpha, acc[4 * q + 1] * alpha, acc[4 * q + 2] * alpha, acc[4 * q + 3] * alpha);
                *(float4*)(ur + i0) = v;
                if (u_half) *(uint2*)(u_half + (size_t)row * nsr + i0) = f32_to_half4<HT>(v);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// bf_lora_tn: partial sums of one chunk of rows -> ws[chunk][s r + j][width]; lora_tn_reduce_kernel adds the chunks in a
// fixed order (bit-for-bit reproducible: tests compare recompute policies bitwise) and writes alpha * sum to out_s.
// ----------------------------------------------------------------------------------------------------------------------
template <typename HT, int R, int NS, bool MASK, bool SHARED>
__global__ __launch_bounds__(256) void lora_tn_kernel(const HT* __restrict__ x, int ldx, const float* __restrict__ u,
                                                      float* __restrict__ ws, int rows, int width, int rows_per_chunk,
                                                      LoraDev sl, unsigned thr_m1) {
    const int c = threadIdx.x * 4;
    if (c >= width) return;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    Quad acc[NS][R];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < R; ++jj) acc[s][jj] = quad_zero();
    constexpr int NL = SHARED ? 1 : NS;
    for (int row = r0; row < r1; row += LORA_UN) {                 // LORA_UN rows of loads in flight before the first use
        uint2 raw[LORA_UN][NL];
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            const HT* xr = x + (size_t)min(row + t, r1 - 1) * ldx + c;
#pragma unroll
            for (int s = 0; s < NL; ++s) raw[t][s] = *(const uint2*)(xr + sl.col_off[s]);
        }
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            if (row + t >= r1) break;                              // uniform
            const float* ur = u + (size_t)(row + t) * (NS * R);    // wave-uniform: scalar loads
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint2 w = raw[t][SHARED ? 0 : s];
                if (MASK) {
                    const uint2 m = dropout_mask4((unsigned long long)(row + t) * (unsigned)width + (unsigned)c, sl.seed[s], thr_m1);
                    w.x &= m.x; w.y &= m.y;
                }
                const Quad xv = quad_of(half4_to_f32<HT>(w), 1.f);
#pragma unroll
                for (int jj = 0; jj < R; ++jj) quad_fma(acc[s][jj], ur[s * R + jj], xv);
            }
        }
    }
    float* wp = ws + (size_t)blockIdx.x * (NS * R) * width + c;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < R; ++jj) *(float4*)(wp + (size_t)(s * R + jj) * width) = quad_f4(acc[s][jj]);
}

struct LoraOut { float* p[3]; };

// 32 consecutive elements x 8 interleaved subsets of the chunks per workgroup (a thread per element alone leaves the sum of
// 1 536 chunks to 48 workgroups: 68 us); the eight partial sums are added in a fixed order.
__global__ __launch_bounds__(256) void lora_tn_reduce_kernel(const float* __restrict__ ws, LoraOut out, int chunks, int nsr,
                                                             int rank, int width, int transposed, float alpha) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + e;                             // element (sj, c) of the [nsr, width] sum
    const int total = nsr * width;
    const size_t stride = (size_t)total;
    float a0 = 0.f, a1 = 0.f;                                      // two chains
    if (t < total) {
        int k = g;
        for (; k + 8 < chunks; k += 16) {
            a0 += ws[(size_t)k * stride + t];
            a1 += ws[(size_t)(k + 8) * stride + t];
        }
        if (k < chunks) a0 += ws[(size_t)k * stride + t];
    }
    part[g][e] = a0 + a1;
    __syncthreads();
    if (g == 0 && t < total) {
        const float v = alpha * (((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) +
                                 ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e])));
        const int sj = t / width, c = t - sj * width;
        const int s = sj / rank, jj = sj - s * rank;
        out.p[s][transposed ? (size_t)c * rank + jj : (size_t)jj * width + c] = v;
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// bf_lora_up: a lane keeps its four columns of every Wt row in registers (S r x 4 floats) and walks a chunk of rows.
// ----------------------------------------------------------------------------------------------------------------------
template <typename HT, int R, int NS, bool MASK, bool SAME>
__global__ __launch_bounds__(256) void lora_up_kernel(const float* __restrict__ u, const HT* __restrict__ wt, HT* __restrict__ y,
                                                      int ldy, int rows, int width, int rows_per_chunk, LoraDev sl,
                                                      unsigned thr_m1, float alpha) {
    const int c = threadIdx.x * 4;
    if (c >= width) return;
    Quad w[NS][R];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < R; ++jj)
            w[s][jj] = quad_of(half4_to_f32<HT>(*(const uint2*)(wt + (size_t)(s * R + jj) * width + c)), alpha);
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    constexpr int NL = SAME ? 1 : NS;
    for (int row = r0; row < r1; row += LORA_UN) {
        uint2 old[LORA_UN][NL];
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            const HT* yr = y + (size_t)min(row + t, r1 - 1) * ldy + c;
#pragma unroll
            for (int s = 0; s < NL; ++s) old[t][s] = *(const uint2*)(yr + sl.col_off[s]);
        }
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            if (row + t >= r1) break;                              // uniform
            const float* ur = u + (size_t)(row + t) * (NS * R);    // wave-uniform: scalar loads
            HT* yr = y + (size_t)(row + t) * ldy + c;
            Quad sum = quad_zero();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                Quad v = quad_zero();
#pragma unroll
                for (int jj = 0; jj < R; ++jj) quad_fma(v, ur[s * R + jj], w[s][jj]);
                if (MASK)
                    quad_mask(v, dropout_mask4((unsigned long long)(row + t) * (unsigned)width + (unsigned)c, sl.seed[s], thr_m1));
                if (SAME) {
                    sum.lo += v.lo; sum.hi += v.hi;
                } else {
                    const Quad o = quad_of(half4_to_f32<HT>(old[t][s]), 1.f);
                    v.lo += o.lo; v.hi += o.hi;
                    *(uint2*)(yr + sl.col_off[s]) = f32_to_half4<HT>(quad_f4(v));
                }
            }
            if (SAME) {
                const Quad o = quad_of(half4_to_f32<HT>(old[t][0]), 1.f);
                sum.lo += o.lo; sum.hi += o.hi;
                *(uint2*)(yr + sl.col_off[0]) = f32_to_half4<HT>(quad_f4(sum));
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// bf_lora_bwd_input = bf_lora_tn (masked, all slots on x) + bf_lora_up (masked, all slots into y) in one walk over the
// rows: both need mask_s at the same (row, column), both multiply by the same du row, and the hash is what either costs
// (377 + 357 us as two launches at 368 640 x 768).  ws / reduce: as bf_lora_tn.
// ----------------------------------------------------------------------------------------------------------------------
template <typename HT, int R, int NS, bool MASK>
__global__ __launch_bounds__(256) void lora_bwd_input_kernel(const HT* __restrict__ x, int ldx, const float* __restrict__ du,
                                                             const HT* __restrict__ wt, HT* __restrict__ y, int ldy,
                                                             float* __restrict__ ws, int rows, int width, int rows_per_chunk,
                                                             LoraDev sl, unsigned thr_m1, float alpha) {
    const int c = threadIdx.x * 4;
    if (c >= width) return;
    Quad w[NS][R], acc[NS][R];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            w[s][jj] = quad_of(half4_to_f32<HT>(*(const uint2*)(wt + (size_t)(s * R + jj) * width + c)), alpha);
            acc[s][jj] = quad_zero();
        }
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    for (int row = r0; row < r1; row += LORA_UN) {
        uint2 xin[LORA_UN], old[LORA_UN];
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            const size_t rr = (size_t)min(row + t, r1 - 1);
            xin[t] = *(const uint2*)(x + rr * ldx + c);
            old[t] = *(const uint2*)(y + rr * ldy + c);
        }
#pragma unroll
        for (int t = 0; t < LORA_UN; ++t) {
            if (row + t >= r1) break;                              // uniform
            const float* ur = du + (size_t)(row + t) * (NS * R);   // wave-uniform: scalar loads
            Quad sum = quad_zero();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint2 xm = xin[t];
                uint2 m = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                if (MASK) {
                    m = dropout_mask4((unsigned long long)(row + t) * (unsigned)width + (unsigned)c, sl.seed[s], thr_m1);
                    xm.x &= m.x; xm.y &= m.y;
                }
                const Quad xv = quad_of(half4_to_f32<HT>(xm), 1.f);
                Quad v = quad_zero();
#pragma unroll
                for (int jj = 0; jj < R; ++jj) {
                    const float uj = ur[s * R + jj];
                    quad_fma(acc[s][jj], uj, xv);
                    quad_fma(v, uj, w[s][jj]);
                }
                if (MASK) quad_mask(v, m);
                sum.lo += v.lo; sum.hi += v.hi;
            }
            const Quad o = quad_of(half4_to_f32<HT>(old[t]), 1.f);
            sum.lo += o.lo; sum.hi += o.hi;
            *(uint2*)(y + (size_t)(row + t) * ldy + c) = f32_to_half4<HT>(quad_f4(sum));
        }
    }
    float* wp = ws + (size_t)blockIdx.x * (NS * R) * width + c;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int jj = 0; jj < R; ++jj) *(float4*)(wp + (size_t)(s * R + jj) * width) = quad_f4(acc[s][jj]);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
int lora_check(const char* who, const bf_lora_slots* sl, int rows, int width, int cdt, int ld, bool wide_cols) {
    BF_REQUIRE(sl && sl->n_slots >= 1 && sl->n_slots <= 3, BF_EINVAL, "%s: n_slots must be 1, 2 or 3", who);
    BF_REQUIRE(sl->rank == 8 || sl->rank == 16, BF_EINVAL, "%s: rank %d (8 and 16 are built)", who, sl->rank);
    BF_REQUIRE(sl->n_slots * sl->rank <= 32, BF_EINVAL, "%s: n_slots * rank = %d > 32", who, sl->n_slots * sl->rank);
    BF_REQUIRE(bf_is_half(cdt), BF_EINVAL, "%s: 16-bit storage only (the fp32 mode runs the adapters as GEMMs)", who);
    BF_REQUIRE(rows > 0 && width >= 16 && width % 16 == 0 && width <= 4096, BF_EINVAL,
               "%s: rows > 0, width a multiple of 16 up to 4096 (got %d x %d)", who, rows, width);
    BF_REQUIRE(sl->p >= 0.f && sl->p < 1.f, BF_EINVAL, "%s: p must be in [0, 1) (got %f)", who, (double)sl->p);
    BF_REQUIRE(ld >= width && ld % 8 == 0, BF_EINVAL, "%s: row stride %d (needs >= width, multiple of 8)", who, ld);
    for (int s = 0; s < sl->n_slots; ++s)
        BF_REQUIRE(sl->col_off[s] >= 0 && sl->col_off[s] % 8 == 0 && (!wide_cols || sl->col_off[s] + width <= ld), BF_EINVAL,
                   "%s: col_off[%d] = %d does not fit a row of %d", who, s, sl->col_off[s], ld);
    return BF_OK;
}
LoraDev lora_dev(const bf_lora_slots* sl) {
    LoraDev d;
    for (int s = 0; s < 3; ++s) { d.col_off[s] = s < sl->n_slots ? sl->col_off[s] : 0; d.seed[s] = s < sl->n_slots ? sl->seed[s] : 0; }
    return d;
}
bool lora_same_cols(const bf_lora_slots* sl) {
    for (int s = 1; s < sl->n_slots; ++s) if (sl->col_off[s] != sl->col_off[0]) return false;
    return true;
}
// rows per chunk of the two vector-unit kernels: about six workgroups per CU, at least 32 rows each
int lora_rows_per_chunk(int rows) {
    int rpc = (rows + 1535) / 1536;
    if (rpc < 32) rpc = 32;
    return (rpc + 3) & ~3;
}

}  // namespace

#define LORA_SWITCH_NS(NSV, BODY) \
    switch (NSV) { case 1: { constexpr int NS = 1; BODY; } break; case 2: { constexpr int NS = 2; BODY; } break; \
                   default: { constexpr int NS = 3; BODY; } break; }

extern "C" int bf_lora_down(const void* x, int ldx, const void* wt, float* u, void* u_half, int rows, int width,
                            const bf_lora_slots* slots, float alpha, int cdt, bf_stream_t stream) {
    if (int rc = lora_check("bf_lora_down", slots, rows, width, cdt, ldx, true)) return rc;
    BF_REQUIRE(x && wt && u && bf_aligned16(x) && bf_aligned16(wt) && bf_aligned16(u) && bf_aligned16(u_half), BF_EALIGN,
               "bf_lora_down: null or unaligned pointer");
    const int nsr = slots->n_slots * slots->rank;
    const size_t lds = (size_t)(nsr + 1) * (width * 2 + 16);
    BF_REQUIRE(lds <= 64 * 1024, BF_EINVAL, "bf_lora_down: width %d needs %zu bytes of LDS", width, lds);
    const LoraDev d = lora_dev(slots);
    const bool mask = slots->p > 0.f && dropout_threshold(slots->p) > 0, shared = lora_same_cols(slots);
    const unsigned thr = dropout_threshold_packed(slots->p);
    const dim3 grid(bf_cdiv(rows, 128)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HT, M_, S_) hipLaunchKernelGGL((lora_down_kernel<HT, NS, M_, S_>), grid, block, lds, s, (const HT*)x, ldx, \
                                              (const HT*)wt, u, (HT*)u_half, rows, width, slots->rank, d, thr, alpha)
#define BY_FLAGS(HT) do { if (mask) { if (shared) LAUNCH(HT, true, true); else LAUNCH(HT, true, false); } \
                          else { if (shared) LAUNCH(HT, false, true); else LAUNCH(HT, false, false); } } while (0)
    LORA_SWITCH_NS(slots->n_slots, if (cdt == BF_F16) BY_FLAGS(f16_t); else BY_FLAGS(bf16_t));
#undef BY_FLAGS
#undef LAUNCH
    BF_CHECK_LAUNCH("bf_lora_down");
    return BF_OK;
}

extern "C" int bf_lora_up(const float* u, const void* wt, void* y, int ldy, int rows, int width, const bf_lora_slots* slots,
                          float alpha, int cdt, bf_stream_t stream) {
    if (int rc = lora_check("bf_lora_up", slots, rows, width, cdt, ldy, true)) return rc;
    BF_REQUIRE(u && wt && y && bf_aligned16(u) && bf_aligned16(wt) && bf_aligned16(y), BF_EALIGN, "bf_lora_up: null or unaligned pointer");
    const LoraDev d = lora_dev(slots);
    const bool mask = slots->p > 0.f && dropout_threshold(slots->p) > 0, same = lora_same_cols(slots);
    for (int a = 0; !same && a < slots->n_slots; ++a)
        for (int b = a + 1; b < slots->n_slots; ++b)
            BF_REQUIRE(abs(slots->col_off[a] - slots->col_off[b]) >= width, BF_EINVAL,
                       "bf_lora_up: column blocks %d and %d overlap without being the same", a, b);
    const unsigned thr = dropout_threshold_packed(slots->p);
    const int rpc = 128;
    const dim3 grid(bf_cdiv(rows, rpc)), block(((width / 4) + 63) & ~63);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HT, R_, M_, S_) hipLaunchKernelGGL((lora_up_kernel<HT, R_, NS, M_, S_>), grid, block, 0, s, u, (const HT*)wt, \
                                                  (HT*)y, ldy, rows, width, rpc, d, thr, alpha)
#define BY_FLAGS(HT, R_) do { if (mask) { if (same) LAUNCH(HT, R_, true, true); else LAUNCH(HT, R_, true, false); } \
                              else { if (same) LAUNCH(HT, R_, false, true); else LAUNCH(HT, R_, false, false); } } while (0)
#define BY_RANK(HT) do { if (slots->rank == 8) BY_FLAGS(HT, 8); else if constexpr (NS <= 2) BY_FLAGS(HT, 16); } while (0)
    LORA_SWITCH_NS(slots->n_slots, if (cdt == BF_F16) BY_RANK(f16_t); else BY_RANK(bf16_t));
#undef BY_RANK
#undef BY_FLAGS
#undef LAUNCH
    BF_CHECK_LAUNCH("bf_lora_up");
    return BF_OK;
}

extern "C" size_t bf_lora_tn_workspace_elems(int rows, int width, const bf_lora_slots* slots) {
    if (!slots || rows <= 0 || width <= 0) return 0;
    const int rpc = lora_rows_per_chunk(rows);
    return (size_t)bf_cdiv(rows, rpc) * slots->n_slots * slots->rank * width;
}

extern "C" int bf_lora_tn(const void* x, int ldx, const float* u, float* const* out, int transposed, float* ws, size_t ws_elems,
                          int rows, int width, const bf_lora_slots* slots, float alpha, int cdt, bf_stream_t stream) {
    if (int rc = lora_check("bf_lora_tn", slots, rows, width, cdt, ldx, true)) return rc;
    BF_REQUIRE(x && u && out && ws && bf_aligned16(x) && bf_aligned16(u) && bf_aligned16(ws), BF_EALIGN, "bf_lora_tn: null or unaligned pointer");
    BF_REQUIRE(ws_elems >= bf_lora_tn_workspace_elems(rows, width, slots), BF_EWORKSPACE, "bf_lora_tn: workspace of %zu elements, %zu needed",
               ws_elems, bf_lora_tn_workspace_elems(rows, width, slots));
    LoraOut o;
    for (int s = 0; s < 3; ++s) {
        o.p[s] = s < slots->n_slots ? out[s] : nullptr;
        BF_REQUIRE(s >= slots->n_slots || o.p[s], BF_EINVAL, "bf_lora_tn: out[%d] is null", s);
    }
    const LoraDev d = lora_dev(slots);
    const bool mask = slots->p > 0.f && dropout_threshold(slots->p) > 0, shared = lora_same_cols(slots);
    const unsigned thr = dropout_threshold_packed(slots->p);
    const int rpc = lora_rows_per_chunk(rows), chunks = bf_cdiv(rows, rpc);
    const dim3 grid(chunks), block(((width / 4) + 63) & ~63);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HT, R_, M_, S_) hipLaunchKernelGGL((lora_tn_kernel<HT, R_, NS, M_, S_>), grid, block, 0, s, (const HT*)x, ldx, u, ws, \
                                                  rows, width, rpc, d, thr)
#define BY_FLAGS(HT, R_) do { if (mask) { if (shared) LAUNCH(HT, R_, true, true); else LAUNCH(HT, R_, true, false); } \
                              else { if (shared) LAUNCH(HT, R_, false, true); else LAUNCH(HT, R_, false, false); } } while (0)
#define BY_RANK(HT) do { if (slots->rank == 8) BY_FLAGS(HT, 8); else if constexpr (NS <= 2) BY_FLAGS(HT, 16); } while (0)
    LORA_SWITCH_NS(slots->n_slots, if (cdt == BF_F16) BY_RANK(f16_t); else BY_RANK(bf16_t));
#undef BY_RANK
#undef BY_FLAGS
#undef LAUNCH
    BF_CHECK_LAUNCH("bf_lora_tn");
    const int nsr = slots->n_slots * slots->rank;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3(bf_cdiv((long)nsr * width, 32)), dim3(256), 0, s, ws, o, chunks, nsr, slots->rank,
                       width, transposed, alpha);
    BF_CHECK_LAUNCH("bf_lora_tn/reduce");
    return BF_OK;
}

extern "C" int bf_lora_bwd_input(const void* x, int ldx, const float* du, const void* wt, void* y, int ldy, float* const* out,
                                 float* ws, size_t ws_elems, int rows, int width, const bf_lora_slots* slots, float alpha, int cdt,
                                 bf_stream_t stream) {
    if (int rc = lora_check("bf_lora_bwd_input", slots, rows, width, cdt, ldx < ldy ? ldx : ldy, false)) return rc;
    BF_REQUIRE(slots->n_slots * slots->rank <= 16, BF_EINVAL, "bf_lora_bwd_input: n_slots * rank = %d > 16 (use bf_lora_tn + bf_lora_up)",
               slots->n_slots * slots->rank);
    BF_REQUIRE(x && du && wt && y && out && ws && bf_aligned16(x) && bf_aligned16(du) && bf_aligned16(wt) && bf_aligned16(y) &&
               bf_aligned16(ws), BF_EALIGN, "bf_lora_bwd_input: null or unaligned pointer");
    BF_REQUIRE(ws_elems >= bf_lora_tn_workspace_elems(rows, width, slots), BF_EWORKSPACE,
               "bf_lora_bwd_input: workspace of %zu elements, %zu needed", ws_elems, bf_lora_tn_workspace_elems(rows, width, slots));
    LoraOut o;
    for (int s = 0; s < 3; ++s) {
        o.p[s] = s < slots->n_slots ? out[s] : nullptr;
        BF_REQUIRE(s >= slots->n_slots || o.p[s], BF_EINVAL, "bf_lora_bwd_input: out[%d] is null", s);
    }
    const LoraDev d = lora_dev(slots);
    const bool mask = slots->p > 0.f && dropout_threshold(slots->p) > 0;
    const unsigned thr = dropout_threshold_packed(slots->p);
    const int rpc = lora_rows_per_chunk(rows), chunks = bf_cdiv(rows, rpc);
    const dim3 grid(chunks), block(((width / 4) + 63) & ~63);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(HT, R_, NS_, M_) hipLaunchKernelGGL((lora_bwd_input_kernel<HT, R_, NS_, M_>), grid, block, 0, s, (const HT*)x, ldx, du, \
                                                   (const HT*)wt, (HT*)y, ldy, ws, rows, width, rpc, d, thr, alpha)
#define BY_MASK(HT, R_, NS_) do { if (mask) LAUNCH(HT, R_, NS_, true); else LAUNCH(HT, R_, NS_, false); } while (0)
#define BY_SHAPE(HT) do { if (slots->rank == 16) BY_MASK(HT, 16, 1); else if (slots->n_slots == 1) BY_MASK(HT, 8, 1); else BY_MASK(HT, 8, 2); } while (0)
    if (cdt == BF_F16) BY_SHAPE(f16_t); else BY_SHAPE(bf16_t);
#undef BY_SHAPE
#undef BY_MASK
#undef LAUNCH
    BF_CHECK_LAUNCH("bf_lora_bwd_input");
    const int nsr = slots->n_slots * slots->rank;
    hipLaunchKernelGGL(lora_tn_reduce_kernel, dim3(bf_cdiv((long)nsr * width, 32)), dim3(256), 0, s, ws, o, chunks, nsr, slots->rank,
                       width, 0, alpha);
    BF_CHECK_LAUNCH("bf_lora_bwd_input/reduce");
    return BF_OK;
}
