// Multi-head attention on the packed QKV buffer [B, L, 3, H, D].
//
//  * attn_fwd_mfma_kernel (bf16): flash-style, one wave per 32 query rows, K / V^T tiles of
//    64 keys staged in LDS and shared by up to 8 waves.  Scores are computed "swapped"
//    (S^T = K Q^T with v_mfma_f32_32x32x16_bf16) so that a query's scores sit in ONE lane
//    column: the row max / row sum are in-register reductions plus a single exchange with
//    lane^32, and the bf16-packed P^T accumulator registers are directly the B operand of
//    O^T += V^T P^T (no LDS round trip for P).
//  * attn_fwd_simple_kernel / attn_bwd_* (fp32 math, fp32 or bf16 I/O): one thread per
//    query (or key) row with K/V (or Q/dO) tiles broadcast from LDS.  These are the exact
//    (parity) mode, and the first-correct backward for the bf16 mode.
#include "common.h"
#include <type_traits>
#include "tuning.h"
#include "attn_common.h"

#include <stdlib.h>

#define KT 64                                 // keys per LDS tile
#ifndef ATTN_Q_PAIRS
#define ATTN_Q_PAIRS false                    // attn_bwd_q_kernel: two key blocks in flight per wave (measured neutral: 363 vs 365 us the pair, step equal)
#endif
#define IMG_U 5                               // resident kernels: image-load requests a thread keeps in flight (x2 operands)

// Forward kernels: the reference point of the online softmax without a row maximum per block (attn_fwd_res_kernel).  Measured
// per head width (profiles/r05_attention_forward_diet.txt): it pays at d = 48 (2 373 x 48 masked: 604 - 620 -> 578 us; 262 x 48:
// 31.2 -> 29.5 us) and COSTS at d = 32 and d = 64 (the branch between the exponentials and the P V products splits the block
// the compiler schedules across, and the second copy of the score code takes 8 - 22 registers: a workgroup per CU at 576 x 64).
template <int D> struct FwdNoMax { static constexpr bool value = (D == 48); };

// LDS images of the resident split pair at d = 32: 64-byte rows WITHOUT the 16-byte pad (36.9 KB instead of 46 KB per
// (batch, head) at 275 tokens: four workgroups per CU instead of three), the 16-byte chunk index XORed with bits 2-3 of
// the row so that the 16 rows a ds_read_b128 group touches fall on distinct banks (rows r and r + 4 share a bank group
// at a 64-byte stride).  Other head widths keep the padded rows (96-byte rows have no power-of-two chunk count).
template <int D> struct ResImg {
    static constexpr bool SWZ = (D == 32);
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int ROW_DP = SWZ ? DP * 2 : DP * 2 + 16;      // rows holding DP columns (the q kernel's K)
    static constexpr int ROW_D = SWZ ? D * 2 : D * 2 + 16;         // rows holding D columns
    // byte offset of 16-byte chunk `c` (+ `in` bytes inside it) of row `row` in an image of `rowb`-byte rows
    static __device__ __forceinline__ int at(int row, int c, int rowb, int in = 0) {
        return SWZ ? row * rowb + ((c ^ ((row >> 2) & 3)) << 4) + in : row * rowb + (c << 4) + in;
    }
};

// Image load of the resident kernels: `total` (row, 16-byte chunk) items of TWO operands, item idx = row * CH + chunk,
// thread t takes items t, t + nthreads, ...  All requests of a thread are in flight before its first LDS write (as a
// plain load-wait-write loop the 4.5 rounds of a 275 x 32 image were 4.5 dependent trips to memory: 8 of the 28 us a
// workgroup lived, tools/attn_probe.py), in batches of at most IMG_U rounds; only the last round is predicated.  `ld`
// and `wr` get (row, chunk, a, b); with a workgroup-uniform base pointer and a 32-bit byte offset inside `ld` the
// loads take the scalar-base form (the 64-bit per-lane address arithmetic was ~20 instructions per request, a fifth of
// the forward kernel's vector instructions).
typedef unsigned img_u32x4 __attribute__((ext_vector_type(4)));
struct ImgPair { img_u32x4 a, b; };          // native vectors, passed by value: references to HIP's uint4 kept the batch in scratch
template <int N, int CH, bool PRED_LAST, typename LoadFn, typename WriteFn>
__device__ __forceinline__ void img_rounds(int first, int nthreads, int total, LoadFn&& ld, WriteFn&& wr) {
    img_u32x4 va[N], vb[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int idx = first + u * nthreads;
        va[u] = (img_u32x4){0u, 0u, 0u, 0u}; vb[u] = va[u];
        if (!(PRED_LAST && u == N - 1) || idx < total) { const ImgPair pr = ld(idx / CH, idx % CH); va[u] = pr.a; vb[u] = pr.b; }
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int idx = first + u * nthreads;
        if (!(PRED_LAST && u == N - 1) || idx < total) wr(idx / CH, idx % CH, va[u], vb[u]);
    }
}
template <int CH, typename LoadFn, typename WriteFn>
__device__ __forceinline__ void img_load(int tid, int nthreads, int total, LoadFn&& ld, WriteFn&& wr) {
    int rounds = (total + nthreads - 1) / nthreads;         // workgroup-uniform
    int first = tid;
    while (rounds > IMG_U) {
        img_rounds<IMG_U, CH, false>(first, nthreads, total, ld, wr);
        first += IMG_U * nthreads; rounds -= IMG_U;
    }
    switch (rounds) {
        case 5: img_rounds<5, CH, true>(first, nthreads, total, ld, wr); break;
        case 4: img_rounds<4, CH, true>(first, nthreads, total, ld, wr); break;
        case 3: img_rounds<3, CH, true>(first, nthreads, total, ld, wr); break;
        case 2: img_rounds<2, CH, true>(first, nthreads, total, ld, wr); break;
        case 1: img_rounds<1, CH, true>(first, nthreads, total, ld, wr); break;
        default: break;
    }
}

// =====================================================================================
//                      simple kernels (fp32 math; T = storage dtype)
// =====================================================================================
template <typename T, int D>
__device__ __forceinline__ void load_row(float (&dst)[D], const T* src) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const float4 v = Vec4<T>::load(src + d);
        dst[d] = v.x; dst[d + 1] = v.y; dst[d + 2] = v.z; dst[d + 3] = v.w;
    }
}

// stage rows [j0, j0+KT) of operand `which` (and optionally a second operand) into LDS fp32
template <typename T, int D>
__device__ __forceinline__ void stage_rows(float* lds, const T* __restrict__ base, int b, int h,
                                           int which, int j0, int L, int H, int nthreads) {
    for (int idx = threadIdx.x; idx < KT * (D / 4); idx += nthreads) {
        const int j = idx / (D / 4), c = idx - j * (D / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j0 + j < L) v = Vec4<T>::load(base + qkv_off(b, j0 + j, which, h, L, H, D) + c * 4);
        *(float4*)(lds + j * D + c * 4) = v;
    }
}

// DROP: nn.Dropout on the probabilities (ref:bifold/models/transformer.py:36, `attn = self.dropout(attn)` AFTER the softmax):
// the normaliser l sums the probabilities as they are, the P V product takes keep(seed, ((b H + h) L + i) L + j) ? p / (1 - rate) : 0.
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256) void attn_fwd_simple_kernel(
    const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
    const int32_t* __restrict__ row_mask, int L, int H, float scale, int mask_mode,
    unsigned drop_thr = 0, float inv_keep = 1.f, unsigned long long seed = 0) {
    __shared__ __attribute__((aligned(16))) float ks[KT * D];
    __shared__ __attribute__((aligned(16))) float vs[KT * D];
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int i = blockIdx.y * 256 + threadIdx.x;
    const bool live = i < L;
    float q[D], o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (live) load_row<T, D>(q, qkv + qkv_off(b, i, 0, h, L, H, D));
    const bool row_masked = live && mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + i] == 0;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < L; j0 += KT) {
        __syncthreads();
        stage_rows<T, D>(ks, qkv, b, h, 1, j0, L, H, 256);
        stage_rows<T, D>(vs, qkv, b, h, 2, j0, L, H, 256);
        __syncthreads();
        if (!live) continue;
        const int jn = min(KT, L - j0);
        for (int j = 0; j < jn; ++j) {
            if (mask_mode == BF_MASK_CAUSAL && j0 + j > i) break;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s = fmaf(q[d], ks[j * D + d], s);
            s = row_masked ? -100000.0f : s * scale;
            if (s > m) {
                const float a = expf(m - s);
                l *= a;
#pragma unroll
                for (int d = 0; d < D; ++d) o[d] *= a;
                m = s;
            }
            const float p = expf(s - m);
            l += p;
            float pd = p;
            if (DROP) pd = dropout_keep((((unsigned long long)bh * L + i) * L) + (unsigned)(j0 + j), seed, drop_thr) ? p * inv_keep : 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = fmaf(pd, vs[j * D + d], o[d]);
        }
    }
    if (live) {
        const float inv = 1.0f / l;
        T* op = out + ((size_t)b * L + i) * (size_t)H * D + (size_t)h * D;
#pragma unroll
        for (int d = 0; d < D; d += 4)
            Vec4<T>::store(op + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv));
        if (lse) lse[((size_t)b * H + h) * L + i] = m + logf(l);
    }
}

// dQ (thread per query) + delta = rowsum(dO * O)
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ row_mask, T* __restrict__ dqkv,
    float* __restrict__ delta_ws, int L, int H, float scale, int mask_mode,
    unsigned drop_thr = 0, float inv_keep = 1.f, unsigned long long seed = 0) {
    __shared__ __attribute__((aligned(16))) float ks[KT * D];
    __shared__ __attribute__((aligned(16))) float vs[KT * D];
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int i = blockIdx.y * 256 + threadIdx.x;
    const bool live = i < L;
    float q[D], dO[D], dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = 0.f; dO[d] = 0.f; dq[d] = 0.f; }
    float delta = 0.f, lse_i = 0.f;
    if (live) {
        load_row<T, D>(q, qkv + qkv_off(b, i, 0, h, L, H, D));
        const size_t oo = ((size_t)b * L + i) * (size_t)H * D + (size_t)h * D;
        load_row<T, D>(dO, dout + oo);
        float ov[D];
        load_row<T, D>(ov, out + oo);
#pragma unroll
        for (int d = 0; d < D; ++d) delta = fmaf(dO[d], ov[d], delta);
        lse_i = lse[((size_t)b * H + h) * L + i];
        delta_ws[((size_t)b * H + h) * L + i] = delta;
    }
    const bool row_masked = live && mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + i] == 0;
    for (int j0 = 0; j0 < L; j0 += KT) {
        __syncthreads();
        stage_rows<T, D>(ks, qkv, b, h, 1, j0, L, H, 256);
        stage_rows<T, D>(vs, qkv, b, h, 2, j0, L, H, 256);
        __syncthreads();
        if (!live || row_masked) continue;       // masked rows: scores are constants, no dQ
        const int jn = min(KT, L - j0);
        for (int j = 0; j < jn; ++j) {
            if (mask_mode == BF_MASK_CAUSAL && j0 + j > i) break;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                s = fmaf(q[d], ks[j * D + d], s);
                dp = fmaf(dO[d], vs[j * D + d], dp);
            }
            const float p = expf(s * scale - lse_i);
            // with dropout the gradient w.r.t. the probability is the masked, rescaled dP; delta = rowsum(dO . O) is unchanged
            if (DROP) dp = dropout_keep((((unsigned long long)bh * L + i) * L) + (unsigned)(j0 + j), seed, drop_thr) ? dp * inv_keep : 0.f;
            const float ds = p * (dp - delta) * scale;
#pragma unroll
            for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, ks[j * D + d], dq[d]);
        }
    }
    if (live) {
        T* dst = dqkv + qkv_off(b, i, 0, h, L, H, D);
#pragma unroll
        for (int d = 0; d < D; d += 4)
            Vec4<T>::store(dst + d, make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]));
    }
}

// dK, dV (thread per key)
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta_ws, const int32_t* __restrict__ row_mask, T* __restrict__ dqkv,
    int L, int H, float scale, int mask_mode,
    unsigned drop_thr = 0, float inv_keep = 1.f, unsigned long long seed = 0) {
    __shared__ __attribute__((aligned(16))) float qs[KT * D];
    __shared__ __attribute__((aligned(16))) float dos[KT * D];
    __shared__ float lses[KT], dels[KT];
    __shared__ int msk[KT];
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int j = blockIdx.y * 256 + threadIdx.x;
    const bool live = j < L;
    const float inv_len = 1.0f / (float)L;
    float k[D], v[D], dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (live) {
        load_row<T, D>(k, qkv + qkv_off(b, j, 1, h, L, H, D));
        load_row<T, D>(v, qkv + qkv_off(b, j, 2, h, L, H, D));
    }
    for (int i0 = 0; i0 < L; i0 += KT) {
        __syncthreads();
        stage_rows<T, D>(qs, qkv, b, h, 0, i0, L, H, 256);
        for (int idx = threadIdx.x; idx < KT * (D / 4); idx += 256) {
            const int r = idx / (D / 4), c = idx - r * (D / 4);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 + r < L)
                t = Vec4<T>::load(dout + ((size_t)b * L + i0 + r) * (size_t)H * D + (size_t)h * D + c * 4);
            *(float4*)(dos + r * D + c * 4) = t;
        }
        if (threadIdx.x < KT) {
            const int i = i0 + threadIdx.x;
            lses[threadIdx.x] = i < L ? lse[((size_t)b * H + h) * L + i] : 0.f;
            dels[threadIdx.x] = i < L ? delta_ws[((size_t)b * H + h) * L + i] : 0.f;
            msk[threadIdx.x] = (i < L && mask_mode == BF_MASK_QUERY_ROW) ? (row_mask[(size_t)b * L + i] == 0) : 0;
        }
        __syncthreads();
        if (!live) continue;
        const int in = min(KT, L - i0);
        for (int r = 0; r < in; ++r) {
            if (mask_mode == BF_MASK_CAUSAL && j > i0 + r) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                s = fmaf(qs[r * D + d], k[d], s);
                dp = fmaf(dos[r * D + d], v[d], dp);
            }
            const bool rm = msk[r] != 0;
            // a masked query row is exactly uniform over the L keys (all scores equal -1e5);
            // exp(-1e5 - lse) would lose ~2^-7 absolute precision at that magnitude
            const float p = rm ? inv_len : expf(s * scale - lses[r]);
            float pd = p;
            if (DROP) {
                const bool keep = dropout_keep((((unsigned long long)bh * L + (unsigned)(i0 + r)) * L) + (unsigned)j, seed, drop_thr);
                pd = keep ? p * inv_keep : 0.f;
                dp = keep ? dp * inv_keep : 0.f;
            }
            const float ds = rm ? 0.f : p * (dp - dels[r]) * scale;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                dv[d] = fmaf(pd, dos[r * D + d], dv[d]);
                dk[d] = fmaf(ds, qs[r * D + d], dk[d]);
            }
        }
    }
    if (live) {
        T* dkp = dqkv + qkv_off(b, j, 1, h, L, H, D);
        T* dvp = dqkv + qkv_off(b, j, 2, h, L, H, D);
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            Vec4<T>::store(dkp + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]));
            Vec4<T>::store(dvp + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
        }
    }
}

// =====================================================================================
//                               bf16 MFMA flash forward
// =====================================================================================
// LDS: K tile  [KT keys][D] bf16, row stride D*2+16 bytes (conflict-free b128 rows)
//      V^T tile [DP][KT keys'] bf16, row stride KT*2+16 bytes; DP = D rounded up to 32;
//      within every 16-key group the key index has bits 2 and 3 swapped, which is the k
//      order of an accumulator tile reused as an MFMA operand (guide §3).
template <int D, int MASK, typename HT>
__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(
    const HT* __restrict__ qkv, HT* __restrict__ out, float* __restrict__ lse,
    const int32_t* __restrict__ row_mask, int L, int H, float scale, int waves_per_block) {
    constexpr int mask_mode = MASK;
    constexpr bool NOMAX = FwdNoMax<D>::value;
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = D / 16;                // k-steps of Q K^T
    constexpr int DB = DP / 32;               // 32-row blocks of O^T
    constexpr int KROW = D * 2 + 16;          // bytes
    constexpr int VROW = KT * 2 + 16;         // bytes
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KROW];
    __shared__ __attribute__((aligned(16))) char v_lds[DP * VROW];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int q0 = (blockIdx.y * waves_per_block + wave) * 32;   // first query of this wave
    const int nthreads = blockDim.x;
    const int qi = q0 + r;                                        // this lane's query
    const bool q_live = qi < L;

    // Q fragments (B operand): B[k = 8hh + j][col r] = Q[qi][16 ks + 8 hh + j]
    typename H16<HT>::x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (q_live) qf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, qi, 0, h, L, H, D) + ks * 16 + hh * 8);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (HT)0.f;
        }
    }
    const bool row_masked = q_live && mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + qi] == 0;

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sl2 = scale * 1.44269504088896340736f;           // scores in log2 units

    // zero the V^T pad rows once (D = 48 -> rows 48..63)
    if (DP != D) {
        for (int idx = threadIdx.x; idx < (DP - D) * (VROW / 4); idx += nthreads)
            ((uint32_t*)(v_lds + D * VROW))[idx] = 0u;
    }
    // causal: keys beyond the block's last query are never needed
    const int q_hi = min(L, (int)(blockIdx.y + 1) * waves_per_block * 32);
    const int key_end = (mask_mode == BF_MASK_CAUSAL) ? q_hi : L;

    // (Round 5: the next tile's K / V rows requested into registers under the current tile's MFMAs -- built, measured, removed:
    // 16 more registers take the kernel from two eight-wave workgroups per CU to one, and the second workgroup was what hid the
    // loads: 2 373 x 48 masked 604 -> 773 - 800 us, 576 x 64 177 -> 207 us, profiles/r05_attention_forward_diet.txt.)
    for (int j0 = 0; j0 < key_end; j0 += KT) {
        __syncthreads();
        // ---- stage K (row-major) and V (transposed, permuted) -------------------------
        for (int idx = threadIdx.x; idx < KT * (D / 8); idx += nthreads) {
            const int j = idx / (D / 8), c = idx - j * (D / 8);
            uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (j0 + j < L) {
                kv = *(const uint4*)(qkv + qkv_off(b, j0 + j, 1, h, L, H, D) + c * 8);
                vv = *(const uint4*)(qkv + qkv_off(b, j0 + j, 2, h, L, H, D) + c * 8);
            }
            *(uint4*)(k_lds + j * KROW + c * 16) = kv;
            const int pos = (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1);   // swap bits 2 and 3
            const uint16_t* ve = (const uint16_t*)&vv;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(uint16_t*)(v_lds + (c * 8 + e) * VROW + pos * 2) = ve[e];
        }
        __syncthreads();

        // ---- S^T = K Q^T for the two 32-key blocks of this tile, masks -------------------------
        auto scores = [&](f32x16 (&sacc)[2]) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const typename H16<HT>::x8 kf = *(const typename H16<HT>::x8*)(k_lds + (kb * 32 + r) * KROW + ks * 32 + hh * 16);
                    sacc[kb] = mfma_32x32x16(kf, qf[ks], sacc[kb]);
                }
            }
            // the key-tail and causal tests run only on tiles that can contain a masked key
            const bool edge = (j0 + KT > L) || (mask_mode == BF_MASK_CAUSAL && j0 + KT - 1 > q0);
            if (mask_mode == BF_MASK_QUERY_ROW && row_masked) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[kb][e] = -100000.0f / scale;   // all scores equal
            }
            if (edge) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = j0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (key >= L || (mask_mode == BF_MASK_CAUSAL && key > qi)) sacc[kb][e] = -INFINITY;
                    }
            }
        };
        // p = exp2(s * sl2 - mref) in place (log2 domain; one FMA with the scale and the reference folded, one v_exp,
        // one add per score, in packed pairs); returns this lane's partial row sum
        auto probs = [&](f32x16 (&sacc)[2], float mref) -> float {
            f32x2 psum2 = {0.f, 0.f};
            const f32x2 sl2v = {sl2, sl2}, mneg = {-mref, -mref};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 t = {sacc[kb][e], sacc[kb][e + 1]};
                    t = __builtin_elementwise_fma(t, sl2v, mneg);
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    sacc[kb][e] = p2[0];
                    sacc[kb][e + 1] = p2[1];
                    psum2 += p2;
                }
            return psum2[0] + psum2[1];
        };
        // reference point without a row maximum per tile: see attn_fwd_res_kernel (round 5)
        f32x16 sacc[2];
        scores(sacc);
        if constexpr (NOMAX) {
            float ps = 0.f;
            bool classic = j0 == 0;
            if (!classic) {
                ps = probs(sacc, (m == -INFINITY) ? 0.f : m);
                classic = __builtin_amdgcn_ballot_w64(!(ps < 16384.0f)) != 0;
                if (classic) scores(sacc);
            }
            if (classic) {
                float tmax = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[kb][e]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;       // sl2 > 0: max commutes with the scale
                const float m_new = fmaxf(m, tmax);
                const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m - m_to);
                l *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
                m = m_new;
                ps = probs(sacc, m_to);
            }
            l += ps;                      // this lane's half of the row sum: the halves meet after the loop
        } else {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[kb][e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;       // sl2 > 0: max commutes with the scale
            const float m_new = fmaxf(m, tmax);
            // lazy reference point (see attn_fwd_res_kernel): m moves, and l / the accumulators are rescaled, only when a
            // row has risen more than 2^6 above it
            if (__builtin_amdgcn_ballot_w64((m_new - m > 6.0f) || (m == -INFINITY)) != 0) {
                const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m - m_to);
                l *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
                m = m_new;
            }
            const float m_use = (m == -INFINITY) ? 0.f : m;
            // pairs of scores go through packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per
            // instruction); the softmax, not the MFMA, bounds this kernel (28 VALU instructions per MFMA)
            f32x2 psum2 = {0.f, 0.f};
            const f32x2 sl2v = {sl2, sl2}, mneg = {-m_use, -m_use};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 t = {sacc[kb][e], sacc[kb][e + 1]};
                    t = __builtin_elementwise_fma(t, sl2v, mneg);
                    f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    sacc[kb][e] = p2[0];
                    sacc[kb][e + 1] = p2[1];
                    psum2 += p2;
                }
            float psum = psum2[0] + psum2[1];
            psum += __shfl_xor(psum, 32, 64);
            l += psum;
        }

        // ---- O^T += V^T P^T : P^T accumulator registers are the B operand ---------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                typename H16<HT>::x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (HT)sacc[kb][8 * s + e];
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const typename H16<HT>::x8 vf =
                        *(const typename H16<HT>::x8*)(v_lds + (db * 32 + r) * VROW + (kb * 32 + s * 16 + hh * 8) * 2);
                    oacc[db] = mfma_32x32x16(vf, pf, oacc[db]);
                }
            }
    }

    if constexpr (NOMAX) l += __shfl_xor(l, 32, 64);   // the two half-rows' partial sums
    if (q_live) {
        const float inv = 1.0f / l;
        HT* op = out + ((size_t)b * L + qi) * (size_t)H * D + (size_t)h * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (HT)(oacc[db][4 * g + e] * inv);
                    *(typename H16<HT>::x4*)(op + d0) = o4;
                }
            }
        if (lse && hh == 0) lse[((size_t)b * H + h) * L + qi] = (m + log2f(l)) * 0.69314718055994530942f;
    }
}

// =====================================================================================
//        bf16 MFMA forward, K / V of the whole (batch, head) resident in LDS
// =====================================================================================
// The tiled kernel above restages K and V^T for every 64-key tile: two barriers per tile and
// eight 2-byte LDS writes per thread to transpose V.  For the sequences of this model family
// (77 / 197 / 275 tokens) the whole K and V of one (batch, head) fit LDS (<= 74 KB), so they
// are loaded ONCE, row-major, and every wave then walks all key tiles without a barrier;
// V^T fragments come from the row-major image by ds_read_b64_tr_b16 (the recipe of the
// backward kernel's dO^T / Q^T operands).  Same arithmetic, same order as the tiled kernel.
template <int D, int MASK, int NKB, typename HT>
__global__ __launch_bounds__(1024) void attn_fwd_res_kernel(
    const HT* __restrict__ qkv, HT* __restrict__ out, float* __restrict__ lse,
    const int32_t* __restrict__ row_mask, int L, int H, float scale, int Lp) {
    constexpr int mask_mode = MASK;
    constexpr bool NOMAX = FwdNoMax<D>::value;
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = D / 16;
    constexpr int DB = DP / 32;
    constexpr int KROW = D * 2 + 16;          // bytes
    // (Round 3 streamed the packed QKV buffer with the non-temporal hint; since the residual adds ride in the LayerNorm passes
    // nothing re-reads what it would have spared, the buffer was written by the GEMM just before and is best found in the
    // caches, and the hint cost 0.7 % of the step: removed in round 6.)
    auto ld16 = [&](const HT* p) -> uint4 { return *(const uint4*)p; };
    constexpr int VROW = DP * 2 + 16;         // bytes (pad columns of D = 48 are zero)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + (size_t)Lp * KROW;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int q0 = wave * 32;
    const int qi = q0 + r;
    const bool q_live = qi < L;
    const int nthreads = blockDim.x;

    const unsigned row_bytes = 3u * (unsigned)H * D * 2u;                                 // one token of the packed QKV buffer
    const char* k_src = (const char*)(qkv + qkv_off(b, 0, 1, h, L, H, D));              // workgroup-uniform bases
    const char* v_src = (const char*)(qkv + qkv_off(b, 0, 2, h, L, H, D));
    if (DP == D) {
        constexpr int CH = D / 8;
        // the zero rows [L, Lp) first (nobody else writes them), then the L live rows without a per-item row test
        for (int idx = threadIdx.x; idx < (Lp - L) * CH; idx += nthreads) {
            const int j = L + idx / CH, c = idx % CH;
            *(uint4*)(k_lds + j * KROW + c * 16) = make_uint4(0, 0, 0, 0);
            *(uint4*)(v_lds + j * VROW + c * 16) = make_uint4(0, 0, 0, 0);
        }
        img_load<CH>(threadIdx.x, nthreads, L * CH,
                     [&](int j, int c) -> ImgPair {
                         const unsigned off = (unsigned)j * row_bytes + (unsigned)c * 16u;
                         return ImgPair{*(const img_u32x4*)(k_src + off), *(const img_u32x4*)(v_src + off)};
                     },
                     [&](int j, int c, img_u32x4 a, img_u32x4 bb) {
                         *(img_u32x4*)(k_lds + j * KROW + c * 16) = a;
                         *(img_u32x4*)(v_lds + j * VROW + c * 16) = bb;
                     });
    } else {
        constexpr int CH = DP / 8;              // d = 48: V rows carry zero pad columns
        const int total = Lp * CH;
        for (int base = threadIdx.x; base < total; base += IMG_U * nthreads) {
            uint4 kv[IMG_U], vv[IMG_U];
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) {
                const int idx = base + u * nthreads;
                const int j = idx / CH, c = idx - j * CH;
                kv[u] = make_uint4(0, 0, 0, 0); vv[u] = make_uint4(0, 0, 0, 0);
                if (idx < total && j < L && c < D / 8) {
                    kv[u] = ld16(qkv + qkv_off(b, j, 1, h, L, H, D) + c * 8);
                    vv[u] = ld16(qkv + qkv_off(b, j, 2, h, L, H, D) + c * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) {
                const int idx = base + u * nthreads;
                const int j = idx / CH, c = idx - j * CH;
                if (idx < total) {
                    if (c < D / 8) *(uint4*)(k_lds + j * KROW + c * 16) = kv[u];
                    *(uint4*)(v_lds + j * VROW + c * 16) = vv[u];
                }
            }
        }
    }

    typename H16<HT>::x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (q_live) {
            const uint4 qq = ld16((const HT*)(k_src - (size_t)H * D * 2 + ((unsigned)qi * row_bytes + (unsigned)(ks * 32 + hh * 16))));
            qf[ks] = *(const typename H16<HT>::x8*)&qq;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (HT)0.f;
        }
    }
    const bool row_masked = q_live && mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + qi] == 0;

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sl2 = scale * 1.44269504088896340736f;
    const int key_end = (mask_mode == BF_MASK_CAUSAL) ? min(L, q0 + 32) : L;
    __syncthreads();

    // Masked scores of the NKB blocks at j0 (raw: the scale rides in the exponent's FMA)
    auto scores = [&](f32x16 (&sacc)[NKB], int j0) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const typename H16<HT>::x8 kf = *(const typename H16<HT>::x8*)(k_lds + (j0 + kb * 32 + r) * KROW + ks * 32 + hh * 16);
                sacc[kb] = mfma_32x32x16(kf, qf[ks], sacc[kb]);
            }
        }
        const bool edge = (j0 + NKB * 32 > L) || (mask_mode == BF_MASK_CAUSAL && j0 + NKB * 32 - 1 > q0);
        if (mask_mode == BF_MASK_QUERY_ROW && row_masked) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[kb][e] = -100000.0f / scale;
        }
        if (edge) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = j0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= L || (mask_mode == BF_MASK_CAUSAL && key > qi)) sacc[kb][e] = -INFINITY;
                }
        }
    };
    // p = exp2(s * sl2 - mref) IN PLACE; returns this lane's partial row sum (pairs of scores go through packed fp32 ops)
    auto probs = [&](f32x16 (&sacc)[NKB], float mref) -> float {
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 sl2v = {sl2, sl2}, mneg = {-mref, -mref};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                f32x2 t = {sacc[kb][e], sacc[kb][e + 1]};
                t = __builtin_elementwise_fma(t, sl2v, mneg);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                sacc[kb][e] = p2[0];
                sacc[kb][e + 1] = p2[1];
                psum2 += p2;
            }
        return psum2[0] + psum2[1];
    };

    // NKB 32-key halves per step: 2 (a 64-key tile) or 1 -- a sequence whose last 64-key tile would be more than half
    // empty (197 = 6 x 32 + 5, 275 = 8 x 32 + 19) walks 32-key steps: 12 % / 10 % fewer exponentials
    for (int j0 = 0; j0 < key_end; j0 += NKB * 32) {
        // Reference point WITHOUT a row maximum per block (round 5).  The block's probabilities are taken, in place,
        // against the reference point m as it stands, p = exp2(s * sl2 - m); the only thing a maximum protects against is
        // p leaving the 16-bit range, and the lane's own partial row sum (which the kernel needs anyway) bounds every p of
        // the lane: while it stays below 2^14 nothing can have overflowed (fp16: 65 504).  The first block of a wave, and
        // any block in which some lane's partial sum reaches 2^14 (or is not a number), takes the classic path: scores
        // AGAIN (the probabilities overwrote them; two or four MFMAs on fragments still in LDS), row maximum, m moved,
        // l and the output accumulators rescaled, p recomputed -- after the first block that is rare (a row would have
        // to rise 14 / log2(e) = 9.7 above its first block's maximum).  Per block this removes 8 v_max3, the half-wave
        // exchange of the maximum with its LDS round trip and the bookkeeping around the lazy test, a quarter of the
        // loop's vector instructions, at the register count of before (a second register set for the probabilities cost
        // a workgroup per CU and 20 % of the kernel: profiles/r05_attention_forward_diet.txt).  l is kept PER LANE (the
        // lane's 16 of the row's 32 keys per block); the two halves of a row meet once, after the loop.
        f32x16 sacc[NKB];
        scores(sacc, j0);
        if constexpr (NOMAX) {
            float ps = 0.f;
            bool classic = j0 == 0;
            if (!classic) {
                ps = probs(sacc, (m == -INFINITY) ? 0.f : m);
                classic = __builtin_amdgcn_ballot_w64(!(ps < 16384.0f)) != 0;
                if (classic) scores(sacc, j0);
            }
            if (classic) {
                float tmax = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[kb][e]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
                const float m_new = fmaxf(m, tmax);
                const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m - m_to);      // m = -inf (first block): 0, and l / oacc are 0
                l *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
                m = m_new;
                ps = probs(sacc, m_to);
            }
            l += ps;
        } else {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[kb][e]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
            // LAZY reference point: the running maximum m is moved -- and l / the output accumulators rescaled -- only when
            // some row of the wave has risen more than 2^6 above it (or has none yet); otherwise the probabilities of this
            // block are taken against the old m (p <= 64: harmless in fp32 sums and in both 16-bit formats) and the 8-16
            // packed multiplies of the rescale, its exponential and the bookkeeping are skipped: after the first block
            // that is almost every block.  The normalisation at the end uses the same m, so the result is unchanged.
            const float m_new = fmaxf(m, tmax);
            if (__builtin_amdgcn_ballot_w64((m_new - m > 6.0f) || (m == -INFINITY)) != 0) {
                const float m_to = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m - m_to);
                l *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
                m = m_new;
            }
            const float m_use = (m == -INFINITY) ? 0.f : m;
            // pairs of scores go through packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per
            // instruction); the softmax, not the MFMA, bounds this kernel (28 VALU instructions per MFMA)
            f32x2 psum2 = {0.f, 0.f};
            const f32x2 sl2v = {sl2, sl2}, mneg = {-m_use, -m_use};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 t = {sacc[kb][e], sacc[kb][e + 1]};
                    t = __builtin_elementwise_fma(t, sl2v, mneg);
                    f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    sacc[kb][e] = p2[0];
                    sacc[kb][e + 1] = p2[1];
                    psum2 += p2;
                }
            float psum = psum2[0] + psum2[1];
            psum += __shfl_xor(psum, 32, 64);
            l += psum;
        }

#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                typename H16<HT>::x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (HT)sacc[kb][8 * s + e];
                const int row0 = j0 + kb * 32 + 16 * s + 4 * hh + q4;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                    const typename H16<HT>::x8 vf = tr_pair<HT>(v_lds + row0 * VROW + cb, v_lds + (row0 + 8) * VROW + cb);
                    oacc[db] = mfma_32x32x16(vf, pf, oacc[db]);
                }
            }
    }

    // The accumulator layout gives a lane 8-byte pieces of ONE row: stored directly, every store
    // instruction scatters 64 pieces over 32 rows (measured: 30 of the kernel's 110 us at
    // 197 x 64).  Turn the wave's 32 x D tile through its own rows of the (now idle) K image
    // instead, so a lane stores 16 bytes and 8 lanes cover a whole row segment.
    __syncthreads();                                   // every wave is done with K
    char* ot = k_lds + (size_t)q0 * KROW;
    if constexpr (NOMAX) l += __shfl_xor(l, 32, 64);   // the two half-rows' partial sums
    {
        const float inv = q_live ? 1.0f / l : 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (HT)(oacc[db][4 * g + e] * inv);
                    *(typename H16<HT>::x4*)(ot + r * KROW + d0 * 2) = o4;
                }
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = D / 8;                          // 16-byte chunks per output row
#pragma unroll
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int row = idx / CPR, c = idx - row * CPR;
        if (q0 + row < L)
            *(uint4*)(out + ((size_t)b * L + q0 + row) * (size_t)H * D + (size_t)h * D + c * 8) =
                *(const uint4*)(ot + row * KROW + c * 16);
    }
    if (q_live && lse && hh == 0) lse[((size_t)b * H + h) * L + qi] = (m + log2f(l)) * 0.69314718055994530942f;
}

// Transposed fragment reads as inline assembly (attn_bwd_fused32p_kernel): the compiler's wait-count pass puts
// `s_waitcnt vmcnt(0)` in front of a ds_read_b64_tr_b16 builtin while a hand-counted global load is in flight.
__device__ __forceinline__ void tr_issue(uint2& lo, uint2& hi, const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) const char* lds_cp;
    const unsigned a0 = (unsigned)(size_t)(lds_cp)p0, a1 = (unsigned)(size_t)(lds_cp)p1;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1));
}

// =====================================================================================
//                               bf16 MFMA backward
// =====================================================================================
// One workgroup per (batch, head); wave w owns the 32 keys [32w, 32w+32) and keeps dK^T and
// dV^T of those keys in accumulators for the whole kernel, so dK/dV need no sum across waves.
// Per step t it visits query block qb = (w + t) mod nw (staggered, so no two waves add into
// the same dQ rows in a step; one barrier per step):
//   S  = Q_qb K_w^T  and  dP = dO_qb V_w^T  with the KEY on the MFMA lane; their accumulators
//   start at -lse/scale and -delta (row constants), so p = exp2(c S) and dS = p dP need no
//   further reductions, and P / dS (packed to bf16) ARE the B operands of
//   dV^T += dO^T P and dK^T += Q^T dS (sums over the accumulator's ROW index, guide §3).
//   Only dS crosses LDS, once, transposed, for dQ += dS K_w (sum over the lane index).
// Q, dO, K tiles of the whole sequence live in LDS as bf16 rows of D*2+16 bytes; the
// transposed operands (Q^T, dO^T, K as B) are read with ds_read_b64_tr_b16.
template <int D>
struct BwdCfg {
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int KS = D / 16;
    static constexpr int DB = DP / 32;
    static constexpr int ROW = D * 2 + 16;
    static constexpr int MAXW = (D == 32) ? 16 : 8;
    static size_t lds_bytes(int nw) {
        const size_t lp = (size_t)nw * 32;
        return 3 * (lp * ROW + 64) + lp * DP * 4 + (size_t)nw * 32 * 64 + 3 * lp * 4;
    }
};

// Generalisation to sequences that do not fit one workgroup: blockIdx.y selects a chunk of
// 32*nw keys (wave w owns 32 of them); the queries are streamed through LDS in tiles of 32*nw
// rows.  dK/dV stay in registers across all query tiles.  Each tile's dQ is accumulated in LDS
// and flushed when the tile is done: straight to dqkv when there is a single key chunk,
// otherwise by fp32 atomics into dq_ws [B, L, H*D] (cast to bf16 by attn_dq_cast_kernel).
template <int D, int MASK, typename HT>
__global__ __launch_bounds__(BwdCfg<D>::MAXW * 64) void attn_bwd_mfma_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    float* __restrict__ dq_ws, int L, int H, float scale) {
    constexpr int mask_mode = MASK;
    using Cf = BwdCfg<D>;
    constexpr int DP = Cf::DP, KS = Cf::KS, DB = Cf::DB, ROW = Cf::ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = blockDim.x >> 6;
    const int Lp = nw * 32;                       // keys per chunk = queries per tile
    char* q_lds = smem;
    char* do_lds = q_lds + (size_t)Lp * ROW + 64;
    char* k_lds = do_lds + (size_t)Lp * ROW + 64;
    float* dq_lds = (float*)(k_lds + (size_t)Lp * ROW + 64);
    char* ds_img = (char*)(dq_lds + (size_t)Lp * DP);
    float* lse_s = (float*)(ds_img + (size_t)nw * 32 * 64);
    float* del_s = lse_s + Lp;
    int* msk_s = (int*)(del_s + Lp);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int key0 = blockIdx.y * Lp;             // first key of this workgroup's chunk
    const int n_chunks = gridDim.y;
    const int n_qtiles = (L + Lp - 1) / Lp;
    const int nthreads = blockDim.x;
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;

    // ---- this chunk's K rows (B operand of dQ) and the wave's K / V fragments ---------------
    for (int idx = threadIdx.x; idx < Lp * (D / 8); idx += nthreads) {
        const int i = idx / (D / 8), c = idx - i * (D / 8);
        uint4 kv = make_uint4(0, 0, 0, 0);
        if (key0 + i < L) kv = *(const uint4*)(qkv + qkv_off(b, key0 + i, 1, h, L, H, D) + c * 8);
        *(uint4*)(k_lds + i * ROW + c * 16) = kv;
    }
    const int key = key0 + w * 32 + r;
    typename H16<HT>::x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (key < L) {
            kf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, key, 1, h, L, H, D) + ks * 16 + hh * 8);
            vf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, key, 2, h, L, H, D) + ks * 16 + hh * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[ks][e] = (HT)0.f; vf[ks][e] = (HT)0.f; }
        }
    }
    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[db][e] = 0.f; dvacc[db][e] = 0.f; }
    const float inv_len = 1.0f / (float)L;
    // keys beyond L have zero K/V rows but a non-zero exp(-lse): kill the whole lane once
    const float lane_on = key < L ? 1.0f : 0.0f;
    char* my_img = ds_img + w * (32 * 64);

    for (int qt = 0; qt < n_qtiles; ++qt) {
        const int qbase = qt * Lp;                // first query of this tile
        // causal: tiles entirely before this key chunk contribute nothing
        if (mask_mode == BF_MASK_CAUSAL && qbase + Lp - 1 < key0) continue;
        __syncthreads();                          // previous tile fully consumed / flushed
        // ---- stage the tile: Q, dO rows (zero beyond L), lse, delta, row mask; clear dQ ------
        for (int idx = threadIdx.x; idx < Lp * (D / 8); idx += nthreads) {
            const int i = idx / (D / 8), c = idx - i * (D / 8);
            uint4 qv = make_uint4(0, 0, 0, 0), dv = qv;
            if (qbase + i < L) {
                qv = *(const uint4*)(qkv + qkv_off(b, qbase + i, 0, h, L, H, D) + c * 8);
                dv = *(const uint4*)(dout + ((size_t)b * L + qbase + i) * (size_t)H * D + (size_t)h * D + c * 8);
            }
            *(uint4*)(q_lds + i * ROW + c * 16) = qv;
            *(uint4*)(do_lds + i * ROW + c * 16) = dv;
        }
        for (int i = threadIdx.x; i < Lp; i += nthreads) {
            float dl = 0.f, ls = 1e30f;
            int mk = 0;
            const int qi = qbase + i;
            if (qi < L) {
                const size_t oo = ((size_t)b * L + qi) * (size_t)H * D + (size_t)h * D;
#pragma unroll
                for (int c = 0; c < D / 8; ++c) {
                    const typename H16<HT>::x8 ov = *(const typename H16<HT>::x8*)(out + oo + c * 8);
                    const typename H16<HT>::x8 gv = *(const typename H16<HT>::x8*)(dout + oo + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl = fmaf((float)ov[e], (float)gv[e], dl);
                }
                ls = lse[((size_t)b * H + h) * L + qi];
                mk = (mask_mode == BF_MASK_QUERY_ROW) ? (row_mask[(size_t)b * L + qi] == 0) : 0;
            }
            lse_s[i] = ls; del_s[i] = dl; msk_s[i] = mk;
        }
        for (int idx = threadIdx.x; idx < Lp * DP; idx += nthreads) dq_lds[idx] = 0.f;
        __syncthreads();

        for (int t = 0; t < nw; ++t) {
            int qb = w + t;
            if (qb >= nw) qb -= nw;
            const int qrow0 = qb * 32;            // row inside the tile
            // causal: a query block entirely before this wave's key block contributes nothing
            const bool skip = ((mask_mode == BF_MASK_CAUSAL) && (qbase + qrow0 + 31 < key0 + w * 32)) ||
                              (qbase + qrow0 >= L);
            if (!skip) {
                // ---- S and dP with row constants as the initial accumulators ------------------
                f32x16 sacc, pacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *(const float4*)(lse_s + qrow0 + 8 * g + 4 * hh);
                    const float4 d4 = *(const float4*)(del_s + qrow0 + 8 * g + 4 * hh);
                    sacc[4 * g + 0] = -l4.x * inv_scale; sacc[4 * g + 1] = -l4.y * inv_scale;
                    sacc[4 * g + 2] = -l4.z * inv_scale; sacc[4 * g + 3] = -l4.w * inv_scale;
                    pacc[4 * g + 0] = -d4.x; pacc[4 * g + 1] = -d4.y; pacc[4 * g + 2] = -d4.z; pacc[4 * g + 3] = -d4.w;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const typename H16<HT>::x8 qa = *(const typename H16<HT>::x8*)(q_lds + (qrow0 + r) * ROW + ks * 32 + hh * 16);
                    const typename H16<HT>::x8 da = *(const typename H16<HT>::x8*)(do_lds + (qrow0 + r) * ROW + ks * 32 + hh * 16);
                    sacc = mfma_32x32x16(qa, kf[ks], sacc);
                    pacc = mfma_32x32x16(da, vf[ks], pacc);
                }
                // ---- P and dS (lane = key column, register = query row) -----------------------
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float p = __builtin_amdgcn_exp2f(sl2 * sacc[e]) * lane_on;
                    float ds = p * pacc[e] * scale;
                    if (mask_mode != BF_MASK_NONE) {
                        const int ql = qrow0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (mask_mode == BF_MASK_QUERY_ROW && msk_s[ql]) { p = inv_len * lane_on; ds = 0.f; }
                        if (mask_mode == BF_MASK_CAUSAL && key > qbase + ql) { p = 0.f; ds = 0.f; }
                    }
                    sacc[e] = p;
                    pacc[e] = ds;
                }
                typename H16<HT>::x8 pb[2], dsb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pb[s][e] = (HT)sacc[8 * s + e]; dsb[s][e] = (HT)pacc[8 * s + e]; }
                // dS image for dQ: [key column][query row], 64-byte rows
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typename H16<HT>::x4 v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = (HT)pacc[4 * g + e];
                    *(typename H16<HT>::x4*)(my_img + r * 64 + (8 * g + 4 * hh) * 2) = v4;
                }
                // ---- dV^T += dO^T P ;  dK^T += Q^T dS   (A operands by transposed LDS reads) ----
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int row0 = qrow0 + 16 * s + 4 * hh + q4;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                        const typename H16<HT>::x8 dot = tr_pair<HT>(do_lds + row0 * ROW + cb, do_lds + (row0 + 8) * ROW + cb);
                        const typename H16<HT>::x8 qtt = tr_pair<HT>(q_lds + row0 * ROW + cb, q_lds + (row0 + 8) * ROW + cb);
                        dvacc[db] = mfma_32x32x16(dot, pb[s], dvacc[db]);
                        dkacc[db] = mfma_32x32x16(qtt, dsb[s], dkacc[db]);
                    }
                }
                // ---- dQ_qb += dS K_w : A = dS^T image (transposed read), B = K rows (transposed read)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    f32x16 dq;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dq[e] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int krow = 16 * s + 8 * hh + q4;          // key within the block (natural k order)
                        const int cq = (16 * (g16 & 1) + 4 * p4) * 2;   // query column in the image
                        const typename H16<HT>::x8 a = tr_pair<HT>(my_img + krow * 64 + cq, my_img + (krow + 4) * 64 + cq);
                        const int cd = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                        const char* kr = k_lds + (w * 32 + krow) * ROW + cd;
                        const typename H16<HT>::x8 kb2 = tr_pair<HT>(kr, kr + 4 * ROW);
                        dq = mfma_32x32x16(a, kb2, dq);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ql = qrow0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        dq_lds[ql * DP + db * 32 + r] += dq[e];
                    }
                }
            }
            __syncthreads();
        }
        // ---- flush this tile's dQ ------------------------------------------------------------
        const int rows = min(Lp, L - qbase);
        if (n_chunks == 1) {
            for (int idx = threadIdx.x; idx < rows * (D / 8); idx += nthreads) {
                const int i = idx / (D / 8), c = idx - i * (D / 8);
                typename H16<HT>::x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (HT)dq_lds[i * DP + c * 8 + e];
                *(typename H16<HT>::x8*)(dqkv + qkv_off(b, qbase + i, 0, h, L, H, D) + c * 8) = o8;
            }
        } else {
            for (int idx = threadIdx.x; idx < rows * D; idx += nthreads) {
                const int i = idx / D, c = idx - i * D;
                atomicAdd(dq_ws + ((size_t)b * L + qbase + i) * (size_t)H * D + (size_t)h * D + c, dq_lds[i * DP + c]);
            }
        }
    }

    // ---- epilogue: dK, dV from the accumulators (lane = key, 4 consecutive d per store) -------
    if (key < L) {
        HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
        HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 k4, v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { k4[e] = (HT)dkacc[db][4 * g + e]; v4[e] = (HT)dvacc[db][4 * g + e]; }
                    *(typename H16<HT>::x4*)(dkp + d0) = k4;
                    *(typename H16<HT>::x4*)(dvp + d0) = v4;
                }
            }
    }
}

// =====================================================================================
//      bf16 MFMA backward, split in two barrier-free kernels (whole sequence in LDS)
// =====================================================================================
// The single-kernel backward above keeps dK/dV per wave and reduces dQ across waves through
// LDS: a read-add-write chain of 16 x DB LDS accesses and a workgroup barrier per step, 128 KB
// of LDS (one workgroup per CU, 2.25 waves per SIMD) -- measured ~6 k cycles per 32x32 block
// against ~1.5 k of instructions.  For sequences whose K+V (or Q+dO) fit LDS the work is split:
//   attn_bwd_q_kernel  : QUERY on the lane, like the forward kernel.  One wave per 32 queries,
//       K and V of the (batch, head) resident in LDS, recomputes S^T = K Q^T and
//       dP^T = V dO^T per key block, dS^T = P^T (dP^T - delta) scale stays in the accumulator
//       registers and is the B operand of dQ^T += K^T dS^T (K^T by transposed LDS reads).
//       Also writes delta = rowsum(dO * O) for the second kernel.  No barrier after the load.
//   attn_bwd_kv_kernel : KEY on the lane, the dK / dV half of the kernel above without the dQ
//       part: Q and dO of the (batch, head) resident in LDS, no dS image, no dQ accumulation,
//       no per-step barrier; 46 KB of LDS at L = 275, d = 32 (two to three workgroups per CU).
// S and dP are computed twice (14 instead of 10 MFMAs per block, exp twice); both kernels read
// their operands once more from HBM.  Same arithmetic per element as the single kernel.
#ifdef ATTN_PROBE   // in-kernel phase stamps of the split backward (tools/attn_probe.py); not part of the product build
#define ATTN_PROBE_WGS 16384
__device__ unsigned long long attn_probe[ATTN_PROBE_WGS * 8];
extern "C" int bf_debug_attn_probe(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_probe), sizeof(attn_probe)) == hipSuccess ? 0 : 1;
}
#define ATTN_STAMP(slot)                                                                                   \
    do {                                                                                                   \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                          \
        if (threadIdx.x == 0 && wg_ < ATTN_PROBE_WGS) attn_probe[wg_ * 8 + (slot)] = wall_clock64();       \
    } while (0)
#else
#define ATTN_STAMP(slot) do {} while (0)
#endif

template <int D, int MASK, typename HT, bool PAIRS = ATTN_Q_PAIRS>
__global__ __launch_bounds__(512) void attn_bwd_q_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    float* __restrict__ delta_ws, int L, int H, float scale, int Lp) {
    constexpr int mask_mode = MASK;
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = D / 16;
    constexpr int DB = DP / 32;
    using RI = ResImg<D>;
    constexpr int KROW = RI::ROW_DP;          // K rows hold DP columns (zero beyond D) for the transposed reads
    constexpr int VROW = RI::ROW_D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // K and V images interleaved by 32-row block (K block i, V block i, K block i + 1, ...): V sits a compile-time
    // 32 * KROW bytes behind K, and a block starts at a multiple of 32 rows, which the swizzle does not see -- the
    // key loop's addresses are one scalar block base plus lane constants (they were 12 vector instructions per block)
    constexpr int BLK = 32 * (KROW + VROW), V_OFF = 32 * KROW;
    char* k_lds = smem;
    auto k_at = [&](int j, int c) -> char* { return k_lds + (j >> 5) * BLK + RI::at(j & 31, c, KROW); };
    auto v_at = [&](int j, int c) -> char* { return k_lds + (j >> 5) * BLK + V_OFF + RI::at(j & 31, c, VROW); };

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    // blockIdx.y: the 32-query blocks of a (batch, head) are spread over several workgroups of
    // ~5 waves, each with its own K / V image: 9-wave workgroups left 7 of a CU's 16 wave slots idle
    // Query blocks of this wave: blockIdx.y spreads the 32-query blocks of a (batch, head) over several workgroups
    // (each with its own K / V image); inside a workgroup wave slot s walks blocks s, s + waves, s + 2 waves, ...  With
    // four-wave workgroups every SIMD holds one wave of each workgroup (five-wave workgroups put two waves on the
    // SIMD the dispatcher starts with: measured 1.5 workgroups alive per CU and that SIMD doing twice the work), and
    // the slot is rotated by the workgroup id so the wave with the extra block is not always on the same SIMD.
    const int wpb = blockDim.x >> 6;
    const int slot = (wave + (int)(blockIdx.x % (unsigned)wpb)) % wpb;
    const int blk_stride = gridDim.y * wpb;
    const int nblk = (L + 31) >> 5;
    const int nthreads = blockDim.x;
    ATTN_STAMP(0);

    // Operands of a 32-query block: the lane's 16-byte pieces of its Q, dO and O rows (delta = rowsum(dO * O) from the
    // pieces the two half-waves hold), lse, the row mask.  Requested ONE BLOCK AHEAD: read at the top of each block
    // they were a trip to memory of 3.6 us per block in front of nine key steps of 4.3 us (tools/attn_probe.py).
    typename H16<HT>::x8 q_nx[KS], do_nx[KS], o_nx[KS];
    float lse_nx = 1e30f;
    bool mask_nx = false;
    // workgroup-uniform base pointers + 32-bit byte offsets: the loads take the scalar-base form
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u, orow_bytes = (unsigned)H * D * 2u;
    const char* q_src = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));
    const char* k_src = q_src + orow_bytes;
    const char* v_src = k_src + orow_bytes;
    const char* do_src = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
    const char* o_src = (const char*)(out + (size_t)b * L * H * D + (size_t)h * D);
    const float* lse_src = lse + ((size_t)b * H + h) * L;
    auto fetch_block = [&](int bk) {
        const int qn = bk * 32 + r;
        lse_nx = 1e30f; mask_nx = false;
        if (bk < nblk && qn < L) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned in_row = (unsigned)(ks * 32 + hh * 16);
                q_nx[ks] = *(const typename H16<HT>::x8*)(q_src + ((unsigned)qn * row_bytes + in_row));
                do_nx[ks] = *(const typename H16<HT>::x8*)(do_src + ((unsigned)qn * orow_bytes + in_row));
                o_nx[ks] = *(const typename H16<HT>::x8*)(o_src + ((unsigned)qn * orow_bytes + in_row));
            }
            lse_nx = lse_src[qn];
            mask_nx = mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + qn] == 0;
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { q_nx[ks][e] = (HT)0.f; do_nx[ks][e] = (HT)0.f; o_nx[ks][e] = (HT)0.f; }
        }
    };
    const int blk_first = blockIdx.y * wpb + slot;
    fetch_block(blk_first);
    int ka_off[KS], va_off[KS], kt_off[DB][2];         // lane constants of the key loop's LDS addresses
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { ka_off[ks] = RI::at(r, ks * 2 + hh, KROW); va_off[ks] = RI::at(r, ks * 2 + hh, VROW); }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
        kt_off[db][0] = RI::at(4 * hh + q4, cb >> 4, KROW, cb & 15);
        kt_off[db][1] = RI::at(4 * hh + q4 + 8, cb >> 4, KROW, cb & 15);
    }

    // image load (img_load: every request in flight before the first LDS write, scalar bases)
    if (DP == D) {
        constexpr int CH = D / 8;
        for (int idx = threadIdx.x; idx < (Lp - L) * CH; idx += nthreads) {       // zero rows [L, Lp)
            const int j = L + idx / CH, c = idx % CH;
            *(uint4*)k_at(j, c) = make_uint4(0, 0, 0, 0);
            *(uint4*)v_at(j, c) = make_uint4(0, 0, 0, 0);
        }
        img_load<CH>(threadIdx.x, nthreads, L * CH,
                     [&](int j, int c) -> ImgPair {
                         const unsigned off = (unsigned)j * row_bytes + (unsigned)c * 16u;
                         return ImgPair{*(const img_u32x4*)(k_src + off), *(const img_u32x4*)(v_src + off)};
                     },
                     [&](int j, int c, img_u32x4 a, img_u32x4 bb) {
                         *(img_u32x4*)k_at(j, c) = a;
                         *(img_u32x4*)v_at(j, c) = bb;
                     });
    } else {
        constexpr int CH = DP / 8;
        const int total = Lp * CH;
        for (int base = threadIdx.x; base < total; base += IMG_U * nthreads) {
            uint4 kv[IMG_U], vv[IMG_U];
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) {
                const int idx = base + u * nthreads;
                const int j = idx / CH, c = idx - j * CH;
                kv[u] = make_uint4(0, 0, 0, 0); vv[u] = make_uint4(0, 0, 0, 0);
                if (idx < total && j < L && c < D / 8) {
                    kv[u] = *(const uint4*)(k_src + ((unsigned)j * row_bytes + (unsigned)c * 16u));
                    vv[u] = *(const uint4*)(v_src + ((unsigned)j * row_bytes + (unsigned)c * 16u));
                }
            }
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) {
                const int idx = base + u * nthreads;
                const int j = idx / CH, c = idx - j * CH;
                if (idx < total) {
                    *(uint4*)k_at(j, c) = kv[u];
                    if (c < D / 8) *(uint4*)v_at(j, c) = vv[u];
                }
            }
        }
    }

    ATTN_STAMP(1);
    __syncthreads();
    ATTN_STAMP(2);
#ifdef ATTN_PROBE
    unsigned long long probe_pro = 0, probe_key = 0, probe_nblk = 0;
#endif
    for (int blk = blk_first; blk < nblk; blk += blk_stride) {
#ifdef ATTN_PROBE
    const unsigned long long probe_t0 = wall_clock64();
#endif
    const int q0 = blk * 32;
    const int qi = q0 + r;
    const bool q_live = qi < L;
    // this block's operands were requested a block ago (the first: before the image load)
    typename H16<HT>::x8 qf[KS], dof[KS];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = q_nx[ks]; dof[ks] = do_nx[ks];
#pragma unroll
        for (int e = 0; e < 8; ++e) delta = fmaf((float)o_nx[ks][e], (float)do_nx[ks][e], delta);
    }
    delta += __shfl_xor(delta, 32, 64);          // the two half-waves hold the two halves of a row
    const float lse_q = lse_nx;
    const bool row_masked = mask_nx;
    if (q_live && hh == 0) (delta_ws + ((size_t)b * H + h) * L)[qi] = delta;
    fetch_block(blk + blk_stride);                // the next block's operands travel under this block's key loop
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;
    const float s_init = -lse_q * inv_scale;

    f32x16 dqacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqacc[db][e] = 0.f;
    const int key_end = (mask_mode == BF_MASK_CAUSAL) ? min(L, q0 + 32) : L;
#ifdef ATTN_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long probe_t1 = wall_clock64();
#endif

    // One key block in three pieces, so that TWO blocks can be in flight per wave: the S / dP products of the second
    // run on the matrix pipe while the vector pipe turns the first into dS (with 3 waves per SIMD a wave that works
    // through one dependent chain at a time left the SIMD idle 44 % of the kernel: SQ_ACTIVE_INST_ANY, tools/attn_pmc.sh)
    auto scores = [&](int j0, f32x16& sacc, f32x16& pacc) {
        const char* kblk = k_lds + (j0 >> 5) * BLK;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = s_init; pacc[e] = -delta; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const typename H16<HT>::x8 kf = *(const typename H16<HT>::x8*)(kblk + ka_off[ks]);
            const typename H16<HT>::x8 vf = *(const typename H16<HT>::x8*)(kblk + V_OFF + (KROW == VROW ? ka_off[ks] : va_off[ks]));
            sacc = mfma_32x32x16(kf, qf[ks], sacc);
            pacc = mfma_32x32x16(vf, dof[ks], pacc);
        }
    };
    // dS^T (rows = keys, lane = query); masked scores carry no gradient.  (The factor `scale` of dS is applied ONCE, to
    // dQ at the end.  The edge test is wave-uniform and per block: inside the element loop it was a scalar branch per
    // pair of scores.)
    auto to_ds = [&](int j0, f32x16& sacc, const f32x16& pacc, auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const f32x2 sl2v = {sl2, sl2};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {                   // packed fp32: two scores per VALU instruction
            const f32x2 t = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
            const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            const f32x2 ds2 = p2 * (f32x2){pacc[e], pacc[e + 1]};
            sacc[e] = ds2[0];
            sacc[e + 1] = ds2[1];
        }
        if (EDGE) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = j0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (key >= L || (mask_mode == BF_MASK_CAUSAL && key > qi)) sacc[e] = 0.f;
            }
        }
        if (mask_mode == BF_MASK_QUERY_ROW && row_masked) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        }
    };
    auto add_dq = [&](int j0, const f32x16& sacc) {
        const char* kblk = k_lds + (j0 >> 5) * BLK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            typename H16<HT>::x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (HT)sacc[8 * s + e];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const typename H16<HT>::x8 kt = tr_pair<HT>(kblk + kt_off[db][0] + s * 16 * KROW, kblk + kt_off[db][1] + s * 16 * KROW);
                dqacc[db] = mfma_32x32x16(kt, pf, dqacc[db]);
            }
        }
    };
    // blocks [0, clean_end) need no edge handling: whole inside the sequence and (causal) strictly below the diagonal
    const int clean_end = (mask_mode == BF_MASK_CAUSAL) ? min(L & ~31, q0) : (L & ~31);
    int j0 = 0;
    if (PAIRS && D == 32) {              // (wider heads: 64 more registers would cost a wave per SIMD)
        for (; j0 + 64 <= clean_end; j0 += 64) {
            f32x16 sa, pa, sb, pb;
            scores(j0, sa, pa);
            scores(j0 + 32, sb, pb);
            to_ds(j0, sa, pa, std::false_type{});
            add_dq(j0, sa);
            to_ds(j0 + 32, sb, pb, std::false_type{});
            add_dq(j0 + 32, sb);
        }
    }
    for (; j0 < key_end; j0 += 32) {
        f32x16 sacc, pacc;
        scores(j0, sacc, pacc);
        if (j0 + 32 <= clean_end) to_ds(j0, sacc, pacc, std::false_type{});
        else to_ds(j0, sacc, pacc, std::true_type{});
        add_dq(j0, sacc);
    }

#ifdef ATTN_PROBE
    {
        const unsigned long long probe_t2 = wall_clock64();
        probe_pro += probe_t1 - probe_t0; probe_key += probe_t2 - probe_t1; probe_nblk += 1;
    }
#endif
    if (q_live) {
        HT* dq = dqkv + qkv_off(b, qi, 0, h, L, H, D);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (HT)(dqacc[db][4 * g + e] * scale);
                    *(typename H16<HT>::x4*)(dq + d0) = o4;
                }
            }
    }
    }   // blocks of this wave
    ATTN_STAMP(3);
#ifdef ATTN_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(4);
    {   // slots 5-7: wave 0's summed 100 MHz ticks in the per-block prologue (operand loads until they have arrived), the
        // key loops and the dQ stores
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && wg_ < ATTN_PROBE_WGS) {
            attn_probe[wg_ * 8 + 5] = probe_pro; attn_probe[wg_ * 8 + 6] = probe_key; attn_probe[wg_ * 8 + 7] = probe_nblk;
        }
    }
#endif
}

// WPE: waves per SIMD the register allocation is held to.  4 (d = 32 only) = 128 registers, so that FOUR four-wave
// workgroups share a CU like the unpadded LDS images allow (the free allocation takes 137 - 149 registers: three).
template <int D, int MASK, typename HT, int WPE = 1>
__global__ __launch_bounds__(512, WPE) void attn_bwd_kv_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta_ws, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    int L, int H, float scale) {
    constexpr int mask_mode = MASK;
    using Cf = BwdCfg<D>;
    using RI = ResImg<D>;
    constexpr int KS = Cf::KS, DB = Cf::DB, ROW = RI::ROW_D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = (L + 31) >> 5;                 // 32-query blocks of the sequence (all resident in LDS)
    const int Lp = nw * 32;
    // Q and dO images interleaved by 32-row block (Q block i, dO block i, Q block i + 1, ...): the dO operand of a
    // block sits a compile-time 32 * ROW bytes behind the Q operand, so one address register serves both
    constexpr int BLK = 2 * 32 * ROW, DO_OFF = 32 * ROW;
    char* q_lds = smem;
    float* lse_s = (float*)(smem + (size_t)Lp * 2 * ROW + 128);
    float* del_s = lse_s + Lp;
    int* msk_s = (int*)(del_s + Lp);

    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;                   // key blocks of this wave: see attn_bwd_q_kernel
    // (the wave index through readfirstlane: the block loops below then run on scalar registers -- as a per-lane value
    // the query-block counter, its wrap-around and every LDS base were vector arithmetic, 20 instructions per block)
    const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = (wave_u + (int)(blockIdx.x % (unsigned)wpb)) % wpb;
    const int blk_stride = gridDim.y * wpb;
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int nthreads = blockDim.x;
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;

    constexpr bool PF = (WPE != 4);     // the 128-register build has no room for the 16 registers of a block in flight
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u, orow_bytes = (unsigned)H * D * 2u;
    const char* q_src = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));       // workgroup-uniform bases
    const char* k_src = q_src + orow_bytes;
    const char* v_src = k_src + orow_bytes;
    const char* do_src = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
    typename H16<HT>::x8 k_nx[KS], v_nx[KS];
    auto fetch_keys = [&](int wn) {
        const int kn = wn * 32 + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (wn < nw && kn < L) {
                const unsigned off = (unsigned)kn * row_bytes + (unsigned)(ks * 32 + hh * 16);
                k_nx[ks] = *(const typename H16<HT>::x8*)(k_src + off);
                v_nx[ks] = *(const typename H16<HT>::x8*)(v_src + off);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { k_nx[ks][e] = (HT)0.f; v_nx[ks][e] = (HT)0.f; }
            }
        }
    };
    const int w_first = blockIdx.y * wpb + slot;
    fetch_keys(w_first);
    {   // image load (img_load: every request in flight before the first LDS write, scalar bases)
        constexpr int CH = D / 8;
        for (int idx = threadIdx.x; idx < (Lp - L) * CH; idx += nthreads) {       // zero rows [L, Lp)
            const int i = L + idx / CH, c = idx % CH;
            char* dst = q_lds + (i >> 5) * BLK + RI::at(i & 31, c, ROW);
            *(uint4*)dst = make_uint4(0, 0, 0, 0);
            *(uint4*)(dst + DO_OFF) = make_uint4(0, 0, 0, 0);
        }
        img_load<CH>(threadIdx.x, nthreads, L * CH,
                     [&](int i, int c) -> ImgPair {
                         return ImgPair{*(const img_u32x4*)(q_src + ((unsigned)i * row_bytes + (unsigned)c * 16u)),
                                        *(const img_u32x4*)(do_src + ((unsigned)i * orow_bytes + (unsigned)c * 16u))};
                     },
                     [&](int i, int c, img_u32x4 a, img_u32x4 bb) {
                         char* dst = q_lds + (i >> 5) * BLK + RI::at(i & 31, c, ROW);
                         *(img_u32x4*)dst = a;
                         *(img_u32x4*)(dst + DO_OFF) = bb;
                     });
    }
    for (int i = threadIdx.x; i < Lp; i += nthreads) {
        float dl = 0.f, ls = 1e30f;
        int mk = 0;
        if (i < L) {
            ls = lse[((size_t)b * H + h) * L + i];
            dl = delta_ws[((size_t)b * H + h) * L + i];
            mk = (mask_mode == BF_MASK_QUERY_ROW) ? (row_mask[(size_t)b * L + i] == 0) : 0;
        }
        // stored as the accumulators' initial values (-lse / scale, -delta): the block loop loads them straight into the
        // MFMA's C registers (they were 24 sign flips and 8 packed multiplies per block)
        lse_s[i] = -ls * inv_scale; del_s[i] = -dl; msk_s[i] = mk;
    }
    // Lane-constant byte offsets inside a 32-row block of the Q / dO images: a block starts at a multiple of 32 rows,
    // which the swizzle (bits 2-3 of the row) does not see, so the per-block address is one scalar base plus these
    // (recomputed per block they were another 25 vector instructions).
    // (rows 16 apart swizzle alike, so the second half of a block is the first plus 16 * ROW: an instruction offset)
    int a_off[KS], t_off[DB][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = RI::at(r, ks * 2 + hh, ROW);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
        t_off[db][0] = RI::at(4 * hh + q4, cb >> 4, ROW, cb & 15);
        t_off[db][1] = RI::at(4 * hh + q4 + 8, cb >> 4, ROW, cb & 15);
    }
    __syncthreads();
    // K / V fragments of a key block, requested one block ahead (the first: before the image load) -- see attn_bwd_q_kernel
    for (int w = w_first; w < nw; w += blk_stride) {
    const int key = w * 32 + r;
    typename H16<HT>::x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = k_nx[ks]; vf[ks] = v_nx[ks]; }
    if (PF) fetch_keys(w + blk_stride);
    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[db][e] = 0.f; dvacc[db][e] = 0.f; }
    const float inv_len = 1.0f / (float)L;
    const float lane_on = key < L ? 1.0f : 0.0f;

    for (int t = 0; t < nw; ++t) {
        int qb = w + t;                           // start at the diagonal: spreads the LDS rows touched per step
        if (qb >= nw) qb -= nw;
        const int qrow0 = qb * 32;
        if (qrow0 >= L || ((mask_mode == BF_MASK_CAUSAL) && (qrow0 + 31 < w * 32))) continue;
        f32x16 sacc, pacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *(const float4*)(lse_s + qrow0 + 8 * g + 4 * hh);
            const float4 d4 = *(const float4*)(del_s + qrow0 + 8 * g + 4 * hh);
            sacc[4 * g + 0] = l4.x; sacc[4 * g + 1] = l4.y; sacc[4 * g + 2] = l4.z; sacc[4 * g + 3] = l4.w;
            pacc[4 * g + 0] = d4.x; pacc[4 * g + 1] = d4.y; pacc[4 * g + 2] = d4.z; pacc[4 * g + 3] = d4.w;
        }
        const char* q_blk = q_lds + qb * BLK;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const char* pa = q_blk + a_off[ks];
            const typename H16<HT>::x8 qa = *(const typename H16<HT>::x8*)pa;
            const typename H16<HT>::x8 da = *(const typename H16<HT>::x8*)(pa + DO_OFF);
            sacc = mfma_32x32x16(qa, kf[ks], sacc);
            pacc = mfma_32x32x16(da, vf[ks], pacc);
        }
        // Two multiplies of the round-2 kernel are gone from this loop: the dead-key mask (a key >= L pollutes only its
        // own lane's dK / dV columns, which are never stored) and the factor `scale` of dS (applied once, to dK at the end)
        const f32x2 sl2v = {sl2, sl2};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {                   // packed fp32: two scores per VALU instruction
            const f32x2 t = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
            f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            f32x2 ds2 = p2 * (f32x2){pacc[e], pacc[e + 1]};
            if (mask_mode != BF_MASK_NONE) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ql = qrow0 + ((e + u) & 3) + 8 * ((e + u) >> 2) + 4 * hh;
                    if (mask_mode == BF_MASK_QUERY_ROW && msk_s[ql]) { p2[u] = inv_len * lane_on; ds2[u] = 0.f; }
                    if (mask_mode == BF_MASK_CAUSAL && key > ql) { p2[u] = 0.f; ds2[u] = 0.f; }
                }
            }
            sacc[e] = p2[0]; sacc[e + 1] = p2[1];
            pacc[e] = ds2[0]; pacc[e + 1] = ds2[1];
        }
        typename H16<HT>::x8 pb[2], dsb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) { pb[s][e] = (HT)sacc[8 * s + e]; dsb[s][e] = (HT)pacc[8 * s + e]; }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* t0 = q_blk + t_off[db][0] + s * 16 * ROW;
                const char* t1 = q_blk + t_off[db][1] + s * 16 * ROW;
                const typename H16<HT>::x8 dot = tr_pair<HT>(t0 + DO_OFF, t1 + DO_OFF);
                const typename H16<HT>::x8 qtt = tr_pair<HT>(t0, t1);
                dvacc[db] = mfma_32x32x16(dot, pb[s], dvacc[db]);
                dkacc[db] = mfma_32x32x16(qtt, dsb[s], dkacc[db]);
            }
        }
    }

    if (key < L) {
        HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
        HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 k4, v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { k4[e] = (HT)(dkacc[db][4 * g + e] * scale); v4[e] = (HT)dvacc[db][4 * g + e]; }
                    *(typename H16<HT>::x4*)(dkp + d0) = k4;
                    *(typename H16<HT>::x4*)(dvp + d0) = v4;
                }
            }
    }
    if (!PF) fetch_keys(w + blk_stride);
    }   // key blocks of this wave
}

// =====================================================================================
//   d = 32, unmasked, <= 288 tokens: the backward in ONE pass (S, dP and the exponentials once)
// =====================================================================================
// The split pair above computes S, dP and P twice (14 MFMAs and 2 x 16 exponentials per 32 x 32 block) and loads two
// operand images per (batch, head).  On this SIMD the matrix and the vector work of a kernel ADD (tools/micro/valu_rate.hip:
// a key step of 6 MFMAs and 56 vector instructions takes the sum of the two at 1, 2, 3 and 4 waves per SIMD alike), so the
// lever is the instruction count: one pass is 10 MFMAs and 16 exponentials per block.  What the pair avoided is the
// reduction of dQ over key blocks that live in different waves; here it needs no reduction at all:
//   * one workgroup of three KEY waves and one HELPER wave per (batch, head).  Key wave w owns key blocks w, w + 3, w + 6
//     (275 tokens = 9 blocks: balanced) with their V fragments and dK^T / dV^T accumulators in registers for the whole
//     kernel (key on the lane: S / dP with the row constants as initial accumulators; P and dS are the B operands of
//     dV^T += dO^T P and dK^T += Q^T dS);
//   * the waves walk the query blocks together, one key block per step.  In a step a key wave leaves dS of its block,
//     converted and transposed, in a 2-KB LDS image (two buffers per wave; 8-byte chunks swizzled); after the step's barrier
//     the helper wave adds dS K of the three images, in wave order, into ITS accumulator of the query block's dQ (A = an
//     image read transposed, B = K rows from a swizzled image) while the key waves are in the next step;
//   * after three steps the helper scales, converts and stores the finished dQ block through its own 2 KB of LDS (whole
//     64-byte row segments).  One accumulator, fixed order: bit-reproducible.  One workgroup barrier per step.
// LDS: Q / dO blocks interleaved 36.9 KB + K image 18.4 KB + dS images 12 KB + row constants 2.3 KB + 2 KB = 72 KB: two
// workgroups per CU, 8 waves = the two per SIMD that 218 registers (96 accumulators + 24 fragments + a block's working
// set) allow.
struct Fused32 {
    static constexpr int ROW = 64, BLK = 2 * 32 * ROW, DO_OFF = 32 * ROW;
    static size_t lds_bytes(int nw) {
        const size_t lp = (size_t)nw * 32;
        return lp * 2 * ROW + lp * ROW + 2 * lp * 4 + 2 * 3 * 2048 + 2048;
    }
};
template <typename HT>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused32_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout, const float* __restrict__ lse,
    HT* __restrict__ dqkv, int L, int H, float scale) {
    constexpr int D = 32, KS = 2, ROW = Fused32::ROW, BLK = Fused32::BLK, DO_OFF = Fused32::DO_OFF, NKB = 3;
    using RI = ResImg<32>;
    using X8 = typename H16<HT>::x8;
    using X4 = typename H16<HT>::x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = (L + 31) >> 5, Lp = nw * 32;
    char* q_lds = smem;                                    // Q block i, dO block i, Q block i + 1, ...
    char* k_img = q_lds + (size_t)Lp * 2 * ROW;            // K rows, swizzled (B operand of dQ)
    float* lse_s = (float*)(k_img + (size_t)Lp * ROW);     // -lse / scale
    float* del_s = lse_s + Lp;                             // -delta
    char* ds_img = (char*)(del_s + Lp);                    // 2 buffers x 3 key waves: dS^T [key][query], 64-byte rows, 8-byte chunks swizzled
    char* dq_x = ds_img + 2 * 3 * 2048;                    // the helper's 2 KB: a finished dQ block on its way to row-contiguous stores

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int nthreads = blockDim.x;
    // wave 3 is a HELPER: it owns no keys; it takes part in the loads and then sums and stores the three partial dQ blocks of
    // every query block while the three key waves are already in the next one (the sum was 10 % of a key wave's time, and
    // the wave whose turn it was kept the other two at the barrier).  8 waves per CU = the two per SIMD 218 registers allow.
    const bool helper = w == 3;
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;

#ifdef ATTN_PROBE
    if (threadIdx.x == 0 && blockIdx.x < ATTN_PROBE_WGS) attn_probe[blockIdx.x * 8 + 0] = wall_clock64();
#endif
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u, orow_bytes = (unsigned)H * D * 2u;
    const char* q_src = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));       // workgroup-uniform bases
    const char* k_src = q_src + orow_bytes;
    const char* v_src = k_src + orow_bytes;
    const char* do_src = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
    const char* o_src = (const char*)(out + (size_t)b * L * H * D + (size_t)h * D);
    const float* lse_src = lse + ((size_t)b * H + h) * L;

    // ---- everything that comes from memory, requested at once ------------------------------------------------------
    X8 vf[NKB][KS];                 // (the K fragments come from the K image: 24 registers the block loop has no room for)
#pragma unroll
    for (int i = 0; i < NKB; ++i) {
        const int key = (w + 3 * i) * 32 + r;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (key < L && !helper) {
                const unsigned off = (unsigned)key * row_bytes + (unsigned)(ks * 32 + hh * 16);
                vf[i][ks] = *(const X8*)(v_src + off);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) vf[i][ks][e] = (HT)0.f;
            }
        }
    }
    // lse of rows t and t + 256 (a register each until the images are written)
    float lse_r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = (int)threadIdx.x + u * 256;
        lse_r[u] = i < L ? lse_src[i] : 1e30f;
    }
    {
        constexpr int CH = D / 8, U = 5;                  // 275 rows x 4 chunks over 256 threads = 4.3 rounds: ONE batch
        for (int idx = threadIdx.x; idx < (Lp - L) * CH; idx += nthreads) {       // zero rows [L, Lp)
            const int i = L + idx / CH, c = idx % CH;
            char* dst = q_lds + (i >> 5) * BLK + RI::at(i & 31, c, ROW);
            *(uint4*)dst = make_uint4(0, 0, 0, 0);
            *(uint4*)(dst + DO_OFF) = make_uint4(0, 0, 0, 0);
            *(uint4*)(k_img + RI::at(i, c, ROW)) = make_uint4(0, 0, 0, 0);
        }
        // item = (row, 16-byte chunk): Q, dO, K chunks into their images, and the O chunk for delta = rowsum(dO * O) -- the
        // four chunks of a row sit in four neighbouring lanes, summed by two lane exchanges
        const int total = L * CH;
        for (int base = threadIdx.x; base < total; base += U * nthreads) {
            img_u32x4 vq[U], vd[U], vk[U], vo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * nthreads;
                const int i = idx / CH, c = idx % CH;
                vq[u] = (img_u32x4){0u, 0u, 0u, 0u}; vd[u] = vq[u]; vk[u] = vq[u]; vo[u] = vq[u];
                if (idx < total) {
                    const unsigned off = (unsigned)i * row_bytes + (unsigned)c * 16u, ooff = (unsigned)i * orow_bytes + (unsigned)c * 16u;
                    vq[u] = *(const img_u32x4*)(q_src + off);
                    vk[u] = *(const img_u32x4*)(k_src + off);
                    vd[u] = *(const img_u32x4*)(do_src + ooff);
                    vo[u] = *(const img_u32x4*)(o_src + ooff);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * nthreads;
                const int i = idx / CH, c = idx % CH;
                const X8 gv = *(const X8*)&vd[u], ov = *(const X8*)&vo[u];
                float dl = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dl = fmaf((float)ov[e], (float)gv[e], dl);
                dl += __shfl_xor(dl, 1, 64);
                dl += __shfl_xor(dl, 2, 64);
                if (idx < total) {
                    char* dst = q_lds + (i >> 5) * BLK + RI::at(i & 31, c, ROW);
                    *(img_u32x4*)dst = vq[u];
                    *(img_u32x4*)(dst + DO_OFF) = vd[u];
                    *(img_u32x4*)(k_img + RI::at(i, c, ROW)) = vk[u];
                    if (c == 0) del_s[i] = -dl;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = (int)threadIdx.x + u * 256;
        if (i < Lp) {
            lse_s[i] = -lse_r[u] * inv_scale;
            if (i >= L) del_s[i] = 0.f;
        }
    }
    // ---- lane constants ------------------------------------------------------------------------------------------------
    int a_off[KS], t_off[2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = RI::at(r, ks * 2 + hh, ROW);
    {
        const int cb = (16 * (g16 & 1) + 4 * p4) * 2;
        t_off[0] = RI::at(4 * hh + q4, cb >> 4, ROW, cb & 15);
        t_off[1] = RI::at(4 * hh + q4 + 8, cb >> 4, ROW, cb & 15);
    }
    // dS^T image: row = key (64 bytes = 32 queries), 8-byte chunk c of row k at chunk c ^ ((k >> 1) & 7): the 16 lanes a
    // ds_write_b64 group serves (16 consecutive keys) then fall on distinct bank pairs
    int dsw_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) dsw_off[g] = r * 64 + (((2 * g + hh) ^ ((r >> 1) & 7)) << 3);
    int dsr_off[2][2], kb_off[2];       // A operand of dQ: the image read transposed; B operand: K rows read transposed
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int krow = 16 * s2 + 8 * hh + q4 + 4 * u;
            dsr_off[s2][u] = krow * 64 + (((4 * (g16 & 1) + p4) ^ ((krow >> 1) & 7)) << 3);
        }
    {
        const int cd = (16 * (g16 & 1) + 4 * p4) * 2;
        kb_off[0] = RI::at(8 * hh + q4, cd >> 4, ROW, cd & 15);
        kb_off[1] = RI::at(8 * hh + q4 + 4, cd >> 4, ROW, cd & 15);
    }
    f32x16 dkacc[NKB], dvacc[NKB];
#pragma unroll
    for (int i = 0; i < NKB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[i][e] = 0.f; dvacc[i][e] = 0.f; }
    __syncthreads();

#ifdef ATTN_PROBE
    unsigned long long pr_compute = 0, pr_barrier = 0, pr_sum = 0;
    const unsigned pr_wg = blockIdx.x;
    if (threadIdx.x == 0 && pr_wg < ATTN_PROBE_WGS) attn_probe[pr_wg * 8 + 1] = wall_clock64();
#endif
    int step = 0;                                       // key-block steps so far: image buffer = step & 1
    f32x16 dq;                                          // helper: dQ of the current query block, over all nine key blocks
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[e] = 0.f;
    for (int qb = 0; qb < nw; ++qb) {
        const char* q_blk = q_lds + qb * BLK;
        const float* lrow = lse_s + qb * 32 + 4 * hh;
        const float* drow = del_s + qb * 32 + 4 * hh;
#pragma unroll
        for (int i = 0; i < NKB; ++i, ++step) {
            char* img_buf = ds_img + (step & 1) * (3 * 2048);
#ifdef ATTN_PROBE
            const unsigned long long pr_t0 = wall_clock64();
#endif
            const int kb = w + 3 * i;
            if (kb < nw && !helper) {
                char* my_img = img_buf + w * 2048;
                f32x16 sacc, pacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *(const float4*)(lrow + 8 * g);
                    const float4 d4 = *(const float4*)(drow + 8 * g);
                    sacc[4 * g + 0] = l4.x; sacc[4 * g + 1] = l4.y; sacc[4 * g + 2] = l4.z; sacc[4 * g + 3] = l4.w;
                    pacc[4 * g + 0] = d4.x; pacc[4 * g + 1] = d4.y; pacc[4 * g + 2] = d4.z; pacc[4 * g + 3] = d4.w;
                }
                const char* krows = k_img + kb * 32 * ROW;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const char* pa = q_blk + a_off[ks];
                    const X8 qa = *(const X8*)pa;
                    const X8 da = *(const X8*)(pa + DO_OFF);
                    const X8 kfr = *(const X8*)(krows + a_off[ks]);
                    sacc = mfma_32x32x16(qa, kfr, sacc);
                    pacc = mfma_32x32x16(da, vf[i][ks], pacc);
                }
                const f32x2 sl2v = {sl2, sl2};
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 t = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    const f32x2 ds2 = p2 * (f32x2){pacc[e], pacc[e + 1]};
                    sacc[e] = p2[0]; sacc[e + 1] = p2[1];
                    pacc[e] = ds2[0]; pacc[e + 1] = ds2[1];
                }
                // a key beyond L has zero K / V rows but exp(-lse) != 0: its dS column would reach dQ (dK / dV of
                // that lane are never stored).  Only the last key block can hold such lanes.
                if (kb * 32 + 32 > L) {
                    const bool dead = kb * 32 + r >= L;
#pragma unroll
                    for (int e = 0; e < 16; ++e) pacc[e] = dead ? 0.f : pacc[e];
                }
                X8 pb[2], dsb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pb[s][e] = (HT)sacc[8 * s + e]; dsb[s][e] = (HT)pacc[8 * s + e]; }
                // dS^T image for the helper's dQ (the converted values: 4 consecutive queries of this lane's key per 8-byte store)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    X4 v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = dsb[g >> 1][4 * (g & 1) + e];
                    *(X4*)(my_img + dsw_off[g]) = v4;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* t0 = q_blk + t_off[0] + s * 16 * ROW;
                    const char* t1 = q_blk + t_off[1] + s * 16 * ROW;
                    const X8 dot = tr_pair<HT>(t0 + DO_OFF, t1 + DO_OFF);
                    const X8 qtt = tr_pair<HT>(t0, t1);
                    dvacc[i] = mfma_32x32x16(dot, pb[s], dvacc[i]);
                    dkacc[i] = mfma_32x32x16(qtt, dsb[s], dkacc[i]);
                }
            }
#ifdef ATTN_PROBE
            const unsigned long long pr_t1 = wall_clock64();
#endif
            __syncthreads();        // the three images of this step are written; the helper is done with the other buffer
#ifdef ATTN_PROBE
            pr_compute += pr_t1 - pr_t0; pr_barrier += wall_clock64() - pr_t1;
#endif
            if (helper) {           // dQ_qb += dS K of the three key blocks of this step, in wave order: A = an image read
                                    // transposed (sum over the key = lane index), B = K rows read transposed
#pragma unroll
                for (int wv = 0; wv < 3; ++wv) {
                    const int kbw = wv + 3 * i;
                    if (kbw < nw) {
                        const char* im = img_buf + wv * 2048;
                        const char* krows = k_img + kbw * 32 * ROW;
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const X8 a = tr_pair<HT>(im + dsr_off[s][0], im + dsr_off[s][1]);
                            const X8 kb2 = tr_pair<HT>(krows + kb_off[0] + s * 16 * ROW, krows + kb_off[1] + s * 16 * ROW);
                            dq = mfma_32x32x16(a, kb2, dq);
                        }
                    }
                }
            }
        }
        if (helper) {               // the finished dQ block: through the helper's own 2 KB (a wave's LDS traffic is in order)
            HT* hs = (HT*)dq_x;     // so that a lane stores 32 contiguous bytes of a row
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ql = (e & 3) + 8 * (e >> 2) + 4 * hh;
                hs[ql * 32 + r] = (HT)(dq[e] * scale);
                dq[e] = 0.f;
            }
            const int row = lane >> 1, half = lane & 1;
            const int qi = qb * 32 + row;
            const X8 lo = *(const X8*)(hs + row * 32 + half * 16), hi = *(const X8*)(hs + row * 32 + half * 16 + 8);
            if (qi < L) {
                HT* dst = dqkv + qkv_off(b, qi, 0, h, L, H, D) + half * 16;
                *(X8*)dst = lo;
                *(X8*)(dst + 8) = hi;
            }
        }
    }

#ifdef ATTN_PROBE
    if (threadIdx.x == 0 && pr_wg < ATTN_PROBE_WGS) {
        attn_probe[pr_wg * 8 + 2] = pr_compute; attn_probe[pr_wg * 8 + 3] = pr_barrier; attn_probe[pr_wg * 8 + 4] = wall_clock64();
    }
#endif
    // ---- dK, dV of this wave's key blocks -------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NKB; ++i) {
        const int key = (w + 3 * i) * 32 + r;
        if (key < L && !helper) {
            HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
            HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * hh;
                X4 k4, v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { k4[e] = (HT)(dkacc[i][4 * g + e] * scale); v4[e] = (HT)dvacc[i][4 * g + e]; }
                *(X4*)(dkp + d0) = k4;
                *(X4*)(dvp + d0) = v4;
            }
        }
    }
}

// =====================================================================================
//   d = 32, unmasked, <= 288 tokens: the one-pass backward as a PERSISTENT kernel (round 5)
// =====================================================================================
// attn_bwd_fused32_kernel above is a chain of latencies: a key wave's step is LDS reads -> 4 MFMAs -> 16 exponentials ->
// LDS write -> transposed reads -> 4 MFMAs -> barrier, 1 200 cycles for 256 cycles of matrix and ~ 270 of vector work,
// with three key blocks per wave in registers (218 registers: 1.5 waves per SIMD) and a load phase of 6.9 us in front
// of 17 us of steps; two workgroups per CU take turns.  Measured on the way here (profiles/r05_pipe_overlap.txt): the
// matrix pipe and the vector ALU of a SIMD DO overlap -- between waves and inside a hand-placed stream -- so what the
// kernel lacks is waves to overlap, and loads under its compute.  This form (nine-block sequences: 257 - 288 tokens):
//   * ONE workgroup of TWELVE waves per CU, walking (batch, head) items.  Waves 0 - 8 are KEY waves with ONE key block
//     each (dK^T / dV^T: 32 accumulator registers instead of 96; K and V fragments of the block in registers for the item):
//     an item is 9 steps of 9 concurrent blocks instead of 27 steps of 3, 2 - 3 key waves per SIMD to overlap.
//   * wave 9 sums dQ: after the barrier that ends step j it adds K^T dS^T of the nine images of query block j (18 MFMAs into
//     one accumulator, key blocks in order: bit-reproducible) while the key waves are in step j + 1.  Its K^T fragments
//     stay in 72 registers for the item; the image reads go out as assembly in batches ahead of the MFMAs (the compiler
//     serialised read - wait - multiply per key block: 1.0 us per query block for 0.27 us of MFMAs); the lane is the query,
//     so a lane stores its own row of dQ.
//   * waves 10 - 11 are LOADERS: the NEXT item's operands arrive under this item's steps.  A query block's Q / dO rows are
//     read in exactly one step, so the slot of block j is free after step j: the loaders write the next item's block j
//     there during step j + 1 (its rows were requested two steps earlier: 21 registers per set, two sets in flight),
//     together with its row constants -lse / scale and -delta = -rowsum(dO * O) (the O rows go through the loaders'
//     registers only) into the second constants buffer, the block's K rows into the second K image and its V rows into
//     the one V image (the key waves take their V fragments at the top of an item).
//   * the requests are inline-assembly loads with hand-counted waits, in straight-line code: hipcc drains vmcnt(0) at a
//     loop header for every load it knows to be pending (a full trip to memory per step), and with requests inside
//     branches it copied the registers of loads in flight (garbage in one test); tools/check_async_loads.py checks the ISA
//     for any use of such a register before its wait.  No LDS-DMA: the compiler puts vmcnt(0) in front of LDS reads behind one.
//   * the roles are separate code paths with the SAME barrier sequence (nine per item + one between items): as one loop
//     with role tests inside, the register allocator kept every role's state alive at once (168 registers, 62 spilled).
//   * the support waves run at raised priority: as the youngest waves of the workgroup they lost the issue arbitration to
//     the key waves on their SIMDs, and every barrier waited for them (sum 1.0 -> 0.74 us per step, 0.66 us uncontended).
// Measured (tools/attn_probe_fusedp.py, B = 256): a step is 0.58 us of key-wave work + 0.48 us at the barrier (the summing
// wave: one SIMD carries its 18 MFMAs beside two key waves' 16), 2 us between items; 259 - 264 us against 282 - 285 us for
// the kernel above on the same box.  Splitting the sum over two waves and steps balanced the SIMDs but took the loader to
// two rows per thread and the kernel to 165 registers with spills in the loader's path: not kept.
// LDS: Q / dO blocks 36.9 KB + 2 K images 36.9 KB + V image 18.4 KB + 2 x row constants 4.6 KB + 3 x 9 dS images 55.3 KB = 152 KB.
// Global loads the compiler's wait-count pass does not see (inline assembly): the persistent kernel below keeps requests in
// flight ACROSS its steps, where hipcc either drains vmcnt(0) at a loop header for every pending load it knows of or, with the
// steps written out, still waits a full trip to memory per step (2.4 us per step measured).  The caller waits by hand, with an
// asm statement that names the registers (so that no use can be scheduled before it); the code between request and wait is
// straight-line (no loop-carried copies of registers whose loads are in flight) -- checked in the ISA.
__device__ __forceinline__ img_u32x4 gload16_async(const void* p) {
    img_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float gload4_async(const void* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
struct Fused32P {
    static constexpr int ROW = 64, BLK = 2 * 32 * ROW, DO_OFF = 32 * ROW;
    static constexpr int NKW = 9, NWAVES = 12;                 // key waves; waves
    static size_t lds_bytes(int nw) {
        const size_t lp = (size_t)nw * 32;
        return lp * 2 * ROW + 3 * lp * ROW + 2 * (2 * lp * 4) + 3 * (size_t)nw * 2048;
    }
};
// Workgroup barrier that waits for this wave's LDS traffic ONLY.  __syncthreads() is a workgroup fence: it also drains
// vmcnt(0), i.e. every global store the wave has issued -- the summing wave's dQ rows of the step before, the key waves'
// dK / dV rows at the end of an item -- and every other wave of the workgroup then stands at the barrier for the length
// of a store's round trip although nothing in the workgroup reads those stores.  (The loaders keep __syncthreads: their
// global loads are inline assembly the compiler does not count, their deposits are LDS writes.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename HT>
__global__ __launch_bounds__(Fused32P::NWAVES * 64) void attn_bwd_fused32p_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout, const float* __restrict__ lse,
    HT* __restrict__ dqkv, int L, int H, float scale, int n_items) {
    constexpr int D = 32, KS = 2, ROW = Fused32P::ROW, BLK = Fused32P::BLK, DO_OFF = Fused32P::DO_OFF;
    constexpr int NT = Fused32P::NWAVES * 64;
    using RI = ResImg<32>;
    using X8 = typename H16<HT>::x8;
    using X4 = typename H16<HT>::x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = (L + 31) >> 5, Lp = nw * 32;
    char* q_lds = smem;                                    // Q block i, dO block i, Q block i + 1, ...
    char* k_img0 = q_lds + (size_t)Lp * 2 * ROW;           // two K images (rows swizzled): the current item's, the next one's
    const int k_img_bytes = Lp * ROW;
    char* v_img = k_img0 + 2 * (size_t)k_img_bytes;        // ONE V image: the key waves take their fragments at the top of an item
    float* cst = (float*)(v_img + (size_t)k_img_bytes);    // 2 x { -lse / scale [Lp], -delta [Lp] }
    char* ds_img = (char*)(cst + 4 * Lp);                  // 3 buffers x nw key blocks: dS^T [key][query], 64-byte rows, 8-byte chunks swizzled

    const int lane = threadIdx.x & 63;
    // ROLE of this wave (w: 0 - 8 key blocks, 9 the dQ sum, 10 - 11 loaders).  Waves p and p + 4 of a workgroup share a SIMD
    // (tools/micro/pipe_overlap.hip), so with role = wave id the summing wave -- the longest role of a step, 0.83 us against
    // the key waves' 0.58 (profiles/r05_attn_bwd_fused32p_probe.txt) -- sat on SIMD 1 beside TWO key waves, and every
    // barrier waited for it.  So SIMD 1 goes to the three SUPPORT waves (physical waves 1, 5, 9: sum, loader, loader) and three
    // key waves to each of the other SIMDs (222 -> 218 us isolated, step 31.905 -> 31.785 ms; roles in wave order: removed).
    const int wp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = (wp & 3) == 1 ? 9 + (wp >> 2) : (wp >> 2) * 3 + ((wp & 3) == 0 ? 0 : (wp & 3) - 1);
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const float inv_scale = 1.0f / scale;
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u, orow_bytes = (unsigned)H * D * 2u;
    const int G = gridDim.x;
    const img_u32x4 z4 = {0u, 0u, 0u, 0u};

    int v = blockIdx.x;
    if (v >= n_items) return;
#ifdef ATTN_PROBE   // stamps of one wave per role (100 MHz wall clock): tools/attn_probe_fusedp.py
    const bool pr_on = lane == 0 && blockIdx.x < ATTN_PROBE_WGS;
    unsigned long long* pr = attn_probe + (size_t)blockIdx.x * 16;     // 16 slots per workgroup (<= 256 workgroups)
    unsigned long long pr_a = 0, pr_b = 0, pr_c = 0;
    if (pr_on && w == 0) pr[0] = wall_clock64();
#endif

    // ---- the first item: everything from memory at once, by all twelve waves -----------------------------------------------
    {
        const int bh = paired_bh<D>(v, n_items), b = bh / H, h = bh - b * H;
        const char* q_src = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));
        const char* k_src = q_src + orow_bytes;
        const char* do_src = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
        const char* o_src = (const char*)(out + (size_t)b * L * H * D + (size_t)h * D);
        const char* v_src = k_src + orow_bytes;
        constexpr int U = 2;                                  // 288 rows x 4 chunks over 768 threads: 1.5 rounds
        img_u32x4 vq[U], vd[U], vk[U], vo[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = (int)threadIdx.x + u * NT;
            const int i = idx >> 2, c = idx & 3;
            vq[u] = z4; vd[u] = z4; vk[u] = z4; vo[u] = z4; vv[u] = z4;
            if (i < L) {
                const unsigned off = (unsigned)i * row_bytes + (unsigned)c * 16u, ooff = (unsigned)i * orow_bytes + (unsigned)c * 16u;
                vq[u] = *(const img_u32x4*)(q_src + off);
                vk[u] = *(const img_u32x4*)(k_src + off);
                vv[u] = *(const img_u32x4*)(v_src + off);
                vd[u] = *(const img_u32x4*)(do_src + ooff);
                vo[u] = *(const img_u32x4*)(o_src + ooff);
            }
        }
        const float ls0 = (int)threadIdx.x < L ? lse[((size_t)b * H + h) * L + threadIdx.x] : 1e30f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = (int)threadIdx.x + u * NT;
            const int i = idx >> 2, c = idx & 3;
            const X8 gv = *(const X8*)&vd[u], ov = *(const X8*)&vo[u];
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf((float)ov[e], (float)gv[e], dl);
            dl += __shfl_xor(dl, 1, 64);
            dl += __shfl_xor(dl, 2, 64);
            if (i < Lp) {
                char* dst = q_lds + (i >> 5) * BLK + RI::at(i & 31, c, ROW);
                *(img_u32x4*)dst = vq[u];
                *(img_u32x4*)(dst + DO_OFF) = vd[u];
                *(img_u32x4*)(k_img0 + RI::at(i, c, ROW)) = vk[u];
                *(img_u32x4*)(v_img + RI::at(i, c, ROW)) = vv[u];
                if (c == 0) cst[Lp + i] = -dl;
            }
        }
        if ((int)threadIdx.x < Lp) cst[threadIdx.x] = -ls0 * inv_scale;
    }
    __syncthreads();

    if (w < nw) {
        // =========================== KEY wave: key block w ======================================================================
        const float sl2 = scale * 1.44269504088896340736f;
        int a_off[KS], t_off[2], dsw_off[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_off[ks] = RI::at(r, ks * 2 + hh, ROW);
        {
            const int cb = (16 * (g16 & 1) + 4 * p4) * 2;
            t_off[0] = RI::at(4 * hh + q4, cb >> 4, ROW, cb & 15);
            t_off[1] = RI::at(4 * hh + q4 + 8, cb >> 4, ROW, cb & 15);
        }
        // dS^T image: row = key (64 bytes = 32 queries), 8-byte chunk c of row k at chunk c ^ ((k >> 1) & 7)
#pragma unroll
        for (int g = 0; g < 4; ++g) dsw_off[g] = r * 64 + (((2 * g + hh) ^ ((r >> 1) & 7)) << 3);
        const int key = w * 32 + r;
        const bool last_kb = w * 32 + 32 > L, dead = key >= L;
        X8 vf[KS], kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            vf[ks] = *(const X8*)(v_img + w * 32 * ROW + a_off[ks]);
            kf[ks] = *(const X8*)(k_img0 + w * 32 * ROW + a_off[ks]);
        }
        int cur = 0;
        for (;;) {
            const int vn = v + G;
            const bool has_next = vn < n_items;
            const int bh = paired_bh<D>(v, n_items), b = bh / H, h = bh - b * H;
            f32x16 dkacc, dvacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { dkacc[e] = 0.f; dvacc[e] = 0.f; }
            const float* lse_c = cst + cur * 2 * Lp;
            const float* del_c = lse_c + Lp;
            for (int j = 0; j < nw; ++j) {
#ifdef ATTN_PROBE
                const unsigned long long pr_t0 = wall_clock64();
#endif
                const char* q_blk = q_lds + j * BLK;
                const float* lrowp = lse_c + j * 32 + 4 * hh;
                const float* drowp = del_c + j * 32 + 4 * hh;
                char* my_img = ds_img + ((j % 3) * nw + w) * 2048;
                f32x16 sacc, pacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *(const float4*)(lrowp + 8 * g);
                    const float4 d4 = *(const float4*)(drowp + 8 * g);
                    sacc[4 * g + 0] = l4.x; sacc[4 * g + 1] = l4.y; sacc[4 * g + 2] = l4.z; sacc[4 * g + 3] = l4.w;
                    pacc[4 * g + 0] = d4.x; pacc[4 * g + 1] = d4.y; pacc[4 * g + 2] = d4.z; pacc[4 * g + 3] = d4.w;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const char* pa = q_blk + a_off[ks];
                    const X8 qa = *(const X8*)pa;
                    const X8 da = *(const X8*)(pa + DO_OFF);
                    sacc = mfma_32x32x16(qa, kf[ks], sacc);
                    pacc = mfma_32x32x16(da, vf[ks], pacc);
                }
                const f32x2 sl2v = {sl2, sl2};
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 t = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    const f32x2 ds2 = p2 * (f32x2){pacc[e], pacc[e + 1]};
                    sacc[e] = p2[0]; sacc[e + 1] = p2[1];
                    pacc[e] = ds2[0]; pacc[e + 1] = ds2[1];
                }
                // a key beyond L has zero K / V rows but exp(-lse) != 0: its dS column would reach dQ (only the last key block)
                if (last_kb) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) pacc[e] = dead ? 0.f : pacc[e];
                }
                X8 pb[2], dsb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pb[s][e] = (HT)sacc[8 * s + e]; dsb[s][e] = (HT)pacc[8 * s + e]; }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    X4 v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = dsb[g >> 1][4 * (g & 1) + e];
                    *(X4*)(my_img + dsw_off[g]) = v4;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* t0 = q_blk + t_off[0] + s * 16 * ROW;
                    const char* t1 = q_blk + t_off[1] + s * 16 * ROW;
                    const X8 dot = tr_pair<HT>(t0 + DO_OFF, t1 + DO_OFF);
                    const X8 qtt = tr_pair<HT>(t0, t1);
                    dvacc = mfma_32x32x16(dot, pb[s], dvacc);
                    dkacc = mfma_32x32x16(qtt, dsb[s], dkacc);
                }
#ifdef ATTN_PROBE
                const unsigned long long pr_t1 = wall_clock64();
#endif
                lds_barrier();
#ifdef ATTN_PROBE
                pr_a += pr_t1 - pr_t0; pr_b += wall_clock64() - pr_t1;
#endif
            }
#ifdef ATTN_PROBE
            const unsigned long long pr_t2 = wall_clock64();
#endif
            if (!dead) {
                HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
                HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * hh;
                    X4 k4, v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { k4[e] = (HT)(dkacc[4 * g + e] * scale); v4[e] = (HT)dvacc[4 * g + e]; }
                    *(X4*)(dkp + d0) = k4;
                    *(X4*)(dvp + d0) = v4;
                }
            }
            if (!has_next) break;
            lds_barrier();                                // between items: the next K image, Q / dO blocks and constants are in place
            cur ^= 1;
            v = vn;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vf[ks] = *(const X8*)(v_img + w * 32 * ROW + a_off[ks]);
                kf[ks] = *(const X8*)(k_img0 + cur * k_img_bytes + w * 32 * ROW + a_off[ks]);
            }
#ifdef ATTN_PROBE
            pr_c += wall_clock64() - pr_t2;
#endif
        }
#ifdef ATTN_PROBE
        if (pr_on && w == 0) { pr[2] = pr_a; pr[3] = pr_b; pr[7] = pr_c; pr[4] = wall_clock64(); }
        if (pr_on && w == 8) { pr[11] = pr_a; pr[12] = pr_b; }
#endif
    } else if (w == 9) {
        // =========================== wave 9: dQ of query block j while the key waves are in step j + 1 ==========================
        // (Two such waves on alternating query blocks, each spreading a block's eighteen MFMAs over two steps so that no SIMD
        // carries them all, were built: with the third wave loading two rows per thread the kernel needed 165 registers, the
        // compiler spilled in the loader's path -- registers with loads in flight -- and one test saw garbage.  One summing wave.)
        // The youngest waves of the workgroup lose the issue arbitration on their SIMD to the two key waves they share it with
        // (oldest first), and every step's barrier waits for them: static priority for the three support waves.
        __builtin_amdgcn_s_setprio(3);
        int dsr_off[2][2], kb_off[2];       // B operand of dQ^T: a dS^T image read transposed; A operand: K rows read transposed
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int krow = 16 * s2 + 8 * hh + q4 + 4 * u;
                dsr_off[s2][u] = krow * 64 + (((4 * (g16 & 1) + p4) ^ ((krow >> 1) & 7)) << 3);
            }
        {
            const int cd = (16 * (g16 & 1) + 4 * p4) * 2;
            kb_off[0] = RI::at(8 * hh + q4, cd >> 4, ROW, cd & 15);
            kb_off[1] = RI::at(8 * hh + q4 + 4, cd >> 4, ROW, cd & 15);
        }
        int cur = 0;
        for (;;) {
            const int vn = v + G;
            const bool has_next = vn < n_items;
            const int bh = paired_bh<D>(v, n_items), b = bh / H, h = bh - b * H;
            const char* k_img = k_img0 + cur * k_img_bytes;
            // K^T fragments of ALL key blocks, once per item (72 registers this wave has to spare): half of the sum's LDS reads,
            // and with the blocks unrolled -- a block beyond the sequence multiplies a ZERO fragment with the last live image --
            // no branch stands between the reads and the MFMAs, so the compiler requests the images ahead.
            X8 kfr[Fused32P::NKW][2];
#pragma unroll
            for (int wv = 0; wv < Fused32P::NKW; ++wv) {
                const char* krows = k_img + (wv < nw ? wv : 0) * 32 * ROW;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    kfr[wv][s] = tr_pair<HT>(krows + kb_off[0] + s * 16 * ROW, krows + kb_off[1] + s * 16 * ROW);
                    if (wv >= nw) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) kfr[wv][s][e] = (HT)0.f;
                    }
                }
            }
            // dQ^T [d][query] = sum over keys of K^T [d][key] dS^T [key][query]: the lane is the QUERY, so a lane holds 16
            // columns of one row of dQ and stores them itself (the two half-waves swap 8-byte pieces: two 16-byte stores
            // per lane, no LDS turn).  Key blocks in order into one accumulator: bit-reproducible.
            // The image reads go out as inline assembly in batches of three key blocks, the next batch before the MFMAs of the
            // current one: written with the builtin, the compiler requested a block's four fragments, waited, multiplied and
            // only then requested the next block's into the SAME registers -- nine LDS round trips per sum, 1.0 us per query
            // block for 0.27 us of MFMAs (tools/attn_probe_fusedp.py).
            f32x16 dq;
            auto issue2 = [&](uint2 (&f)[2][2][2], const char* img_buf, int b2, int nblk) {   // key blocks 2 b2, 2 b2 + 1
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i >= nblk) continue;
                    const int wv = 2 * b2 + i;
                    const char* im = img_buf + (wv < nw ? wv : nw - 1) * 2048;
#pragma unroll
                    for (int s = 0; s < 2; ++s) tr_issue(f[i][s][0], f[i][s][1], im + dsr_off[s][0], im + dsr_off[s][1]);
                }
            };
            auto mul2 = [&](uint2 (&f)[2][2][2], int b2, int nblk) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i >= nblk) continue;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const uint4 vq = make_uint4(f[i][s][0].x, f[i][s][0].y, f[i][s][1].x, f[i][s][1].y);
                        dq = mfma_32x32x16(kfr[2 * b2 + i][s], *(const X8*)&vq, dq);
                    }
                }
            };
#define BF_LGKM_WAIT2(N, f)                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]), "+v"(f[1][0][0]), \
                 "+v"(f[1][0][1]), "+v"(f[1][1][0]), "+v"(f[1][1][1]))
#define BF_LGKM_WAIT1(N, f)                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0][0][0]), "+v"(f[0][0][1]), "+v"(f[0][1][0]), "+v"(f[0][1][1]))
            static_assert(Fused32P::NKW == 9, "the batches below are written for nine key blocks");
            auto half1 = [&](int jq) {
#pragma unroll
                for (int e = 0; e < 16; ++e) dq[e] = 0.f;
                const char* img_buf = ds_img + (size_t)(jq % 3) * nw * 2048;
                uint2 fa[2][2][2], fb[2][2][2];
                issue2(fa, img_buf, 0, 2);
                issue2(fb, img_buf, 1, 2);
                BF_LGKM_WAIT2(8, fa); mul2(fa, 0, 2); issue2(fa, img_buf, 2, 2);
                BF_LGKM_WAIT2(8, fb); mul2(fb, 1, 2); issue2(fb, img_buf, 3, 2);
                BF_LGKM_WAIT2(8, fa); mul2(fa, 2, 2); issue2(fa, img_buf, 4, 1);
                BF_LGKM_WAIT2(4, fb); mul2(fb, 3, 2);
                BF_LGKM_WAIT1(0, fa); mul2(fa, 4, 1);
            };
#undef BF_LGKM_WAIT2
#undef BF_LGKM_WAIT1
            auto half2 = [&](int jq) {
                const int qi = jq * 32 + r;
                uint2 own[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    X4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (HT)(dq[4 * g + e] * scale);
                    own[g] = *(const uint2*)&o4;
                }
                uint4 st[2];                                // lane (r, 0) owns d 0-7 and 16-23, lane (r, 1) d 8-15 and 24-31
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const unsigned ax = own[2 * k].x, ay = own[2 * k].y, bx = own[2 * k + 1].x, by = own[2 * k + 1].y;
                    const unsigned gx = (unsigned)__shfl_xor((int)(hh ? ax : bx), 32, 64);
                    const unsigned gy = (unsigned)__shfl_xor((int)(hh ? ay : by), 32, 64);
                    st[k].x = hh ? gx : ax; st[k].y = hh ? gy : ay;
                    st[k].z = hh ? bx : gx; st[k].w = hh ? by : gy;
                }
                if (qi < L) {
                    HT* dst = dqkv + qkv_off(b, qi, 0, h, L, H, D) + hh * 8;
                    *(uint4*)dst = st[0];
                    *(uint4*)(dst + 16) = st[1];
                }
            };
            for (int j = 0; j < nw; ++j) {
#ifdef ATTN_PROBE
                const unsigned long long pr_t0 = wall_clock64();
#endif
                if (j >= 1) { half1(j - 1); half2(j - 1); }
#ifdef ATTN_PROBE
                const unsigned long long pr_t1 = wall_clock64();
                pr_a += pr_t1 - pr_t0;
#endif
                lds_barrier();
#ifdef ATTN_PROBE
                pr_c += wall_clock64() - pr_t1;
#endif
            }
#ifdef ATTN_PROBE
            const unsigned long long pr_t3 = wall_clock64();
#endif
            half1(nw - 1); half2(nw - 1);
#ifdef ATTN_PROBE
            pr_b += wall_clock64() - pr_t3;
#endif
            if (!has_next) break;
            lds_barrier();
            cur ^= 1;
            v = vn;
        }
#ifdef ATTN_PROBE
        if (pr_on) { pr[5] = pr_a; pr[6] = pr_b; pr[8] = pr_c; }
#endif
    } else if (w >= 10) {
        // =========================== waves 10, 11: the next item's rows under this item's steps ================================
        // Plain loads, and the nine steps written out: hipcc drains vmcnt(0) at a loop header for every load pending at its
        // entry (the requests would be waited for one step after they are made), and asm loads hidden from its wait-count pass
        // were copied between registers before they had landed.  Straight-line code gets exact counts.
        __builtin_amdgcn_s_setprio(3);
        const int lt = (w - 10) * 64 + lane, lrow = lt >> 2, lc = lt & 3;           // (row, chunk) of a 32-row block
        struct LSet { img_u32x4 q[1], d[1], o[1], k[1], vv[1]; float ls[1]; };
        int cur = 0;
        for (;;) {
            const int vn = v + G;
            const bool has_next = vn < n_items;
            const int bhn = paired_bh<D>(has_next ? vn : v, n_items), bn = bhn / H, hn = bhn - bn * H;
            const char* q_src = (const char*)(qkv + qkv_off(bn, 0, 0, hn, L, H, D));
            const char* k_src = q_src + orow_bytes;
            const char* v_src = k_src + orow_bytes;
            const char* do_src = (const char*)(dout + (size_t)bn * L * H * D + (size_t)hn * D);
            const char* o_src = (const char*)(out + (size_t)bn * L * H * D + (size_t)hn * D);
            const float* ls_src = lse + ((size_t)bn * H + hn) * L;
            char* k_next = k_img0 + (cur ^ 1) * k_img_bytes;
            float* c_next = cst + (cur ^ 1) * 2 * Lp;
            // rows of the next item's block m (a row beyond the sequence asks for the last live row and is zeroed on deposit):
            // six requests per set, so that the waits can count
            auto request = [&](LSet& st, int m) {
#pragma unroll
                for (int u = 0; u < 1; ++u) {
                    const int i = m * 32 + lrow, ic = i < L ? i : L - 1;
                    const unsigned off = (unsigned)ic * row_bytes + (unsigned)lc * 16u, ooff = (unsigned)ic * orow_bytes + (unsigned)lc * 16u;
                    st.q[u] = gload16_async(q_src + off);
                    st.k[u] = gload16_async(k_src + off);
                    st.vv[u] = gload16_async(v_src + off);
                    st.d[u] = gload16_async(do_src + ooff);
                    st.o[u] = gload16_async(o_src + ooff);
                    st.ls[u] = gload4_async(ls_src + ic);
                }
            };
            // ... into the slot block m of THIS item has left; K and constants into the other buffers; V into the one V image
            // (the key waves took their fragments at the top of the item)
            auto deposit = [&](LSet& st, int m, bool newer, bool live) {   // NEWER: the other set's six requests are younger and may stay in flight
                if (newer) asm volatile("s_waitcnt vmcnt(6)" : "+v"(st.q[0]), "+v"(st.k[0]), "+v"(st.vv[0]), "+v"(st.d[0]), "+v"(st.o[0]), "+v"(st.ls[0]));
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(st.q[0]), "+v"(st.k[0]), "+v"(st.vv[0]), "+v"(st.d[0]), "+v"(st.o[0]), "+v"(st.ls[0]));
                if (!live) return;
#pragma unroll
                for (int u = 0; u < 1; ++u) {
                    const int row = lrow, i = m * 32 + row;
                    if (i >= L) { st.q[u] = z4; st.k[u] = z4; st.vv[u] = z4; st.d[u] = z4; st.o[u] = z4; st.ls[u] = 1e30f; }
                    const X8 gv = *(const X8*)&st.d[u], ov = *(const X8*)&st.o[u];
                    float dl = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl = fmaf((float)ov[e], (float)gv[e], dl);
                    dl += __shfl_xor(dl, 1, 64);
                    dl += __shfl_xor(dl, 2, 64);
                    char* dst = q_lds + m * BLK + RI::at(row, lc, ROW);
                    *(img_u32x4*)dst = st.q[u];
                    *(img_u32x4*)(dst + DO_OFF) = st.d[u];
                    *(img_u32x4*)(k_next + RI::at(i, lc, ROW)) = st.k[u];
                    *(img_u32x4*)(v_img + RI::at(i, lc, ROW)) = st.vv[u];
                    if (lc == 0) {
                        c_next[i] = -st.ls[u] * inv_scale;
                        c_next[Lp + i] = -dl;
                    }
                }
            };
            // Straight-line, unconditional requests and waits (the host sends only nine-block sequences here): a request inside
            // a branch makes the registers of a load in flight the operands of a merge, and the compiler COPIES them right
            // behind the load (tools/check_async_loads.py looks for exactly that in the ISA).  Without a next item the
            // requests re-read this item's rows and nothing is deposited.
            // (Three sets in flight -- a block requested three steps before its deposit -- measured: 216 - 221 us against
            // 218 with two: the loader does not hold the barriers up.  profiles/r05_attn_bwd_fused32p_probe.txt: it stands at
            // them for 590 of a step's 970 ns.)
            LSet set0, set1;
            request(set0, 0);
            // step j: block j - 1 (requested in step j - 2) into its slot, block j + 1 requested; the sets alternate
#ifdef ATTN_PROBE
            const unsigned long long pr_t0 = wall_clock64();
#endif
#ifdef ATTN_PROBE
#define BF_LOADER_SYNC { const unsigned long long pt_ = wall_clock64(); __syncthreads(); pr_b += wall_clock64() - pt_; }
#else
#define BF_LOADER_SYNC __syncthreads();
#endif
#define BF_LOADER_STEP(J, ST)                                        \
            if ((J) >= 1) deposit(ST, (J) - 1, true, has_next);      \
            if ((J) + 1 < Fused32P::NKW) request(ST, (J) + 1);       \
            BF_LOADER_SYNC
            BF_LOADER_STEP(0, set1) BF_LOADER_STEP(1, set0) BF_LOADER_STEP(2, set1) BF_LOADER_STEP(3, set0) BF_LOADER_STEP(4, set1)
            BF_LOADER_STEP(5, set0) BF_LOADER_STEP(6, set1) BF_LOADER_STEP(7, set0) BF_LOADER_STEP(8, set1)
#undef BF_LOADER_STEP
#undef BF_LOADER_SYNC
            deposit(set0, Fused32P::NKW - 1, false, has_next);
#ifdef ATTN_PROBE
            pr_a += wall_clock64() - pr_t0;
#endif
            if (!has_next) break;
            __syncthreads();
            cur ^= 1;
            v = vn;
        }
#ifdef ATTN_PROBE
        if (pr_on && w == 10) { pr[1] = pr_a; pr[10] = pr_b; }
#endif
    } else {
        // key waves beyond the sequence's blocks (nw < 9): only the barriers
        for (;;) {
            const int vn = v + G;
            for (int j = 0; j < nw; ++j) __syncthreads();
            if (vn >= n_items) break;
            __syncthreads();
            v = vn;
        }
    }
}

// =====================================================================================
//   The split pair for sequences that do NOT fit LDS (SigLIP towers at 384 px: 576 tokens;
//   the sequential fusion: 642 - 2 950 tokens): same two kernels, the resident operand streamed
// =====================================================================================
// attn_bwd_mfma_kernel handles these lengths with key chunks across workgroups, query tiles
// restaged per chunk, one barrier per 32-query step and dQ reduced through LDS and fp32 atomics:
// 218 TFLOP/s at 2 373 x 48.  Here the wave keeps ITS 32 queries (or keys) in registers for the
// whole kernel, exactly as in attn_bwd_q_kernel / attn_bwd_kv_kernel, and the other operand walks
// through LDS in chunks of CH rows: two barriers per CHUNK, no dQ reduction, no atomics, no dq_ws.
// Geometry of the streamed pair (measured, tools/attn_bench.py --family siglip: 128-row chunks and
// 4-wave workgroups 1.85 / 0.47 ms at 2 373 x 48 / 576 x 64; 256 rows 1.99 / 0.52; 8 waves 1.91 / 0.53).
// With the next chunk's rows held in registers the kernels need 190 - 245 registers, so a CU runs 8 waves as
// two workgroups of 4 (one stages while the other computes).  Compile-time so that the prefetch lives in a
// fixed set of registers.
template <int D>
struct StreamCfg {
    static constexpr int CH = 128;                          // rows per chunk
    static constexpr int WPB = 4;                           // waves per workgroup
    static constexpr int NT = WPB * 64;
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int NPF_Q = CH * (DP / 8) / NT;        // 16-byte pieces per thread and operand (q kernel: K, V)
    static constexpr int NPF_KV = (CH * (D / 8) + NT - 1) / NT;   // (kv kernel: Q, dO)
};

template <int D, int MASK, typename HT>
__global__ __launch_bounds__(StreamCfg<D>::NT, 2) void attn_bwd_q_stream_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    float* __restrict__ delta_ws, int L, int H, float scale) {
    constexpr int mask_mode = MASK;
    using SC = StreamCfg<D>;
    constexpr int CH = SC::CH, NPF = SC::NPF_Q, NT = SC::NT;
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = D / 16;
    constexpr int DB = DP / 32;
    constexpr int KROW = DP * 2 + 16;         // K rows padded to DP columns (zero) for the transposed reads
    constexpr int VROW = D * 2 + 16;
    constexpr int CPR = DP / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + (size_t)CH * KROW;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int nwaves = SC::WPB;
    const int q0 = (blockIdx.y * nwaves + wave) * 32;
    const int qi = q0 + r;
    const bool q_live = qi < L;
    const bool wave_live = q0 < L;

    // The NEXT chunk's K / V rows travel in registers while the current chunk is being consumed:
    // unconditional loads from clamped addresses (the compiler then counts vmcnt instead of draining
    // it), zeros substituted when the registers are written to LDS.
    uint4 pk[NPF], pv[NPF];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int j = min(c0 + idx / CPR, L - 1), c = min(idx % CPR, D / 8 - 1);
            pk[i] = *(const uint4*)(qkv + qkv_off(b, j, 1, h, L, H, D) + c * 8);
            pv[i] = *(const uint4*)(qkv + qkv_off(b, j, 2, h, L, H, D) + c * 8);
        }
    };
    auto stash = [&](int c0) {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int jl = idx / CPR, c = idx - jl * CPR;
            uint4 kq = pk[i], vq = pv[i];
            if (!(c0 + jl < L && c < D / 8)) { kq = z4; vq = z4; }
            *(uint4*)(k_lds + jl * KROW + c * 16) = kq;
            if (c < D / 8) *(uint4*)(v_lds + jl * VROW + c * 16) = vq;
        }
    };

    typename H16<HT>::x8 qf[KS], dof[KS];
    float delta = 0.f, lse_q = 1e30f;
    bool row_masked = false;
    if (q_live) {
        const size_t oo = ((size_t)b * L + qi) * (size_t)H * D + (size_t)h * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, qi, 0, h, L, H, D) + ks * 16 + hh * 8);
            dof[ks] = *(const typename H16<HT>::x8*)(dout + oo + ks * 16 + hh * 8);
        }
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            const typename H16<HT>::x8 ov = *(const typename H16<HT>::x8*)(out + oo + c * 8);
            const typename H16<HT>::x8 gv = *(const typename H16<HT>::x8*)(dout + oo + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta = fmaf((float)ov[e], (float)gv[e], delta);
        }
        lse_q = lse[((size_t)b * H + h) * L + qi];
        row_masked = mask_mode == BF_MASK_QUERY_ROW && row_mask[(size_t)b * L + qi] == 0;
        if (hh == 0) delta_ws[((size_t)b * H + h) * L + qi] = delta;
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { qf[ks][e] = (HT)0.f; dof[ks][e] = (HT)0.f; }
    }
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;
    const float s_init = -lse_q * inv_scale;

    f32x16 dqacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqacc[db][e] = 0.f;
    const int key_end = (mask_mode == BF_MASK_CAUSAL) ? min(L, q0 + 32) : L;
    // query-row mask: a masked row's scores are the constant -1e5 (transformer.py:35), so nothing flows to
    // Q or K through it; when all 32 rows of the wave are masked (whole context frames are) the wave only
    // keeps the barriers company
    const bool wave_masked = mask_mode == BF_MASK_QUERY_ROW && __all(row_masked || !q_live);
    // keys any wave of this workgroup needs (uniform: every wave walks the same chunks)
    const int wg_key_end = (mask_mode == BF_MASK_CAUSAL) ? min(L, (int)(blockIdx.y + 1) * nwaves * 32) : L;

    // a workgroup whose 128 queries are all masked (inside a masked context frame) streams nothing
    const bool wg_idle = mask_mode == BF_MASK_QUERY_ROW && __syncthreads_and(wave_masked || !wave_live);
    if (!wg_idle) fetch(0);
    for (int c0 = 0; c0 < wg_key_end && !wg_idle; c0 += CH) {
        __syncthreads();                          // the previous chunk is consumed
        stash(c0);
        __syncthreads();
        if (c0 + CH < wg_key_end) fetch(c0 + CH); // in flight under this chunk's MFMAs
        if (!wave_live || wave_masked) continue;  // masked query rows receive no gradient: dQ stays 0
        const int kend = min(key_end, c0 + CH);
        for (int j0 = c0; j0 < kend; j0 += 32) {
            const int jl = j0 - c0;
            f32x16 sacc, pacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[e] = s_init; pacc[e] = -delta; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const typename H16<HT>::x8 kf = *(const typename H16<HT>::x8*)(k_lds + (jl + r) * KROW + ks * 32 + hh * 16);
                const typename H16<HT>::x8 vf = *(const typename H16<HT>::x8*)(v_lds + (jl + r) * VROW + ks * 32 + hh * 16);
                sacc = mfma_32x32x16(kf, qf[ks], sacc);
                pacc = mfma_32x32x16(vf, dof[ks], pacc);
            }
            const bool edge = (j0 + 32 > L) || (mask_mode == BF_MASK_CAUSAL && j0 + 31 > q0);
            const f32x2 sl2v = {sl2, sl2}, scalev = {scale, scale};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 t = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
                const f32x2 p2 = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                const f32x2 ds2 = p2 * ((f32x2){pacc[e], pacc[e + 1]} * scalev);
                sacc[e] = ds2[0];
                sacc[e + 1] = ds2[1];
            }
            if (edge) {                                   // one branch per block (see attn_bwd_q_kernel)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = j0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= L || (mask_mode == BF_MASK_CAUSAL && key > qi)) sacc[e] = 0.f;
                }
            }
            if (mask_mode == BF_MASK_QUERY_ROW && row_masked) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                typename H16<HT>::x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (HT)sacc[8 * s + e];
                const int row0 = jl + 16 * s + 4 * hh + q4;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                    const typename H16<HT>::x8 kt = tr_pair<HT>(k_lds + row0 * KROW + cb, k_lds + (row0 + 8) * KROW + cb);
                    dqacc[db] = mfma_32x32x16(kt, pf, dqacc[db]);
                }
            }
        }
    }

    if (q_live) {
        HT* dq = dqkv + qkv_off(b, qi, 0, h, L, H, D);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (HT)dqacc[db][4 * g + e];
                    *(typename H16<HT>::x4*)(dq + d0) = o4;
                }
            }
    }
}

template <int D, int MASK, typename HT>
__global__ __launch_bounds__(StreamCfg<D>::NT, 2) void attn_bwd_kv_stream_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta_ws, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    int L, int H, float scale) {
    constexpr int mask_mode = MASK;
    using Cf = BwdCfg<D>;
    using SC = StreamCfg<D>;
    constexpr int CH = SC::CH, NPF = SC::NPF_KV, NT = SC::NT;
    constexpr int KS = Cf::KS, DB = Cf::DB, ROW = Cf::ROW;
    constexpr int CPR = D / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_lds = smem;
    char* do_lds = q_lds + (size_t)CH * ROW + 64;
    float* lse_s = (float*)(do_lds + (size_t)CH * ROW + 64);
    float* del_s = lse_s + CH;
    int* msk_s = (int*)(del_s + CH);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = SC::WPB;
    const int w = blockIdx.y * nwaves + wave;     // this wave's 32-key block
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;
    const bool wave_live = w * 32 < L;
    constexpr int nblk = CH >> 5;                 // 32-query blocks per chunk

    // next chunk's Q / dO rows (+ lse, delta, mask of its first CH <= NT rows) in registers, as in the q kernel
    static_assert(CH <= NT, "one lse / delta / mask entry per thread");
    uint4 pq[NPF], pd[NPF];
    float p_ls = 1e30f, p_dl = 0.f;
    int p_mk = 0;
    auto fetch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = min((int)threadIdx.x + i * NT, CH * CPR - 1);
            const int il = idx / CPR, c = idx - il * CPR;
            const int i_g = min(c0 + il, L - 1);
            pq[i] = *(const uint4*)(qkv + qkv_off(b, i_g, 0, h, L, H, D) + c * 8);
            pd[i] = *(const uint4*)(dout + ((size_t)b * L + i_g) * (size_t)H * D + (size_t)h * D + c * 8);
        }
        const int i_g = min(c0 + (int)threadIdx.x, L - 1);
        p_ls = lse[((size_t)b * H + h) * L + i_g];
        p_dl = delta_ws[((size_t)b * H + h) * L + i_g];
        if (mask_mode == BF_MASK_QUERY_ROW) p_mk = row_mask[(size_t)b * L + i_g] == 0;
    };
    auto stash = [&](int c0) {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int il = idx / CPR, c = idx - il * CPR;
            uint4 qv = pq[i], dv = pd[i];
            if (!(c0 + il < L)) { qv = z4; dv = z4; }
            if (idx < CH * CPR) {
                *(uint4*)(q_lds + il * ROW + c * 16) = qv;
                *(uint4*)(do_lds + il * ROW + c * 16) = dv;
            }
        }
        if ((int)threadIdx.x < CH) {
            const bool live = c0 + (int)threadIdx.x < L;
            lse_s[threadIdx.x] = live ? p_ls : 1e30f;
            del_s[threadIdx.x] = live ? p_dl : 0.f;
            msk_s[threadIdx.x] = live ? p_mk : 0;
        }
    };

    const int key = w * 32 + r;
    typename H16<HT>::x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (key < L) {
            kf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, key, 1, h, L, H, D) + ks * 16 + hh * 8);
            vf[ks] = *(const typename H16<HT>::x8*)(qkv + qkv_off(b, key, 2, h, L, H, D) + ks * 16 + hh * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { kf[ks][e] = (HT)0.f; vf[ks][e] = (HT)0.f; }
        }
    }
    f32x16 dkacc[DB], dvacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[db][e] = 0.f; dvacc[db][e] = 0.f; }
    const float inv_len = 1.0f / (float)L;
    const float lane_on = key < L ? 1.0f : 0.0f;
    const int wg_first_key = (int)blockIdx.y * nwaves * 32;

    // causal: query chunks before the workgroup's first key see none of its keys
    const int c_first = (mask_mode == BF_MASK_CAUSAL) ? (wg_first_key / CH) * CH : 0;
    fetch(c_first);
    for (int c0 = c_first; c0 < L; c0 += CH) {
        __syncthreads();                          // the previous chunk is consumed
        stash(c0);
        __syncthreads();
        if (c0 + CH < L) fetch(c0 + CH);          // in flight under this chunk's MFMAs
        if (!wave_live) continue;

        for (int t = 0; t < nblk; ++t) {
            int qb = wave + t;                    // staggered start: spreads the LDS rows touched per step
            if (qb >= nblk) qb -= nblk;
            if (qb >= nblk) qb -= nblk;
            const int lq0 = qb * 32;              // row inside the chunk
            const int qrow0 = c0 + lq0;           // query index
            if (qrow0 >= L || ((mask_mode == BF_MASK_CAUSAL) && (qrow0 + 31 < w * 32))) continue;
            if (mask_mode == BF_MASK_QUERY_ROW) {
                // all 32 query rows masked (rows beyond L count as masked: their dO rows are zero): P = 1/L for
                // every (query, key), dS = 0 -> only dV^T += dO^T P is left, no scores, no exp, no dK
                const int qrow = qrow0 + r;
                if (__all(msk_s[lq0 + r] != 0 || qrow >= L)) {
                    typename H16<HT>::x8 pconst;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pconst[e] = (HT)(inv_len * lane_on);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int row0 = lq0 + 16 * s + 4 * hh + q4;
#pragma unroll
                        for (int db = 0; db < DB; ++db) {
                            const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                            const typename H16<HT>::x8 dot = tr_pair<HT>(do_lds + row0 * ROW + cb, do_lds + (row0 + 8) * ROW + cb);
                            dvacc[db] = mfma_32x32x16(dot, pconst, dvacc[db]);
                        }
                    }
                    continue;
                }
            }
            f32x16 sacc, pacc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l4 = *(const float4*)(lse_s + lq0 + 8 * g + 4 * hh);
                const float4 d4 = *(const float4*)(del_s + lq0 + 8 * g + 4 * hh);
                sacc[4 * g + 0] = -l4.x * inv_scale; sacc[4 * g + 1] = -l4.y * inv_scale;
                sacc[4 * g + 2] = -l4.z * inv_scale; sacc[4 * g + 3] = -l4.w * inv_scale;
                pacc[4 * g + 0] = -d4.x; pacc[4 * g + 1] = -d4.y; pacc[4 * g + 2] = -d4.z; pacc[4 * g + 3] = -d4.w;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const typename H16<HT>::x8 qa = *(const typename H16<HT>::x8*)(q_lds + (lq0 + r) * ROW + ks * 32 + hh * 16);
                const typename H16<HT>::x8 da = *(const typename H16<HT>::x8*)(do_lds + (lq0 + r) * ROW + ks * 32 + hh * 16);
                sacc = mfma_32x32x16(qa, kf[ks], sacc);
                pacc = mfma_32x32x16(da, vf[ks], pacc);
            }
            const f32x2 sl2v = {sl2, sl2}, onv = {lane_on, lane_on}, scalev = {scale, scale};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 tt = (f32x2){sacc[e], sacc[e + 1]} * sl2v;
                f32x2 p2 = (f32x2){__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])} * onv;
                f32x2 ds2 = p2 * ((f32x2){pacc[e], pacc[e + 1]} * scalev);
                if (mask_mode != BF_MASK_NONE) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int lql = lq0 + ((e + u) & 3) + 8 * ((e + u) >> 2) + 4 * hh;
                        if (mask_mode == BF_MASK_QUERY_ROW && msk_s[lql]) { p2[u] = inv_len * lane_on; ds2[u] = 0.f; }
                        if (mask_mode == BF_MASK_CAUSAL && key > c0 + lql) { p2[u] = 0.f; ds2[u] = 0.f; }
                    }
                }
                sacc[e] = p2[0]; sacc[e + 1] = p2[1];
                pacc[e] = ds2[0]; pacc[e + 1] = ds2[1];
            }
            typename H16<HT>::x8 pb[2], dsb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { pb[s][e] = (HT)sacc[8 * s + e]; dsb[s][e] = (HT)pacc[8 * s + e]; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row0 = lq0 + 16 * s + 4 * hh + q4;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int cb = (db * 32 + 16 * (g16 & 1) + 4 * p4) * 2;
                    const typename H16<HT>::x8 dot = tr_pair<HT>(do_lds + row0 * ROW + cb, do_lds + (row0 + 8) * ROW + cb);
                    const typename H16<HT>::x8 qtt = tr_pair<HT>(q_lds + row0 * ROW + cb, q_lds + (row0 + 8) * ROW + cb);
                    dvacc[db] = mfma_32x32x16(dot, pb[s], dvacc[db]);
                    dkacc[db] = mfma_32x32x16(qtt, dsb[s], dkacc[db]);
                }
            }
        }
    }

    if (key < L) {
        HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
        HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    typename H16<HT>::x4 k4, v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { k4[e] = (HT)dkacc[db][4 * g + e]; v4[e] = (HT)dvacc[db][4 * g + e]; }
                    *(typename H16<HT>::x4*)(dkp + d0) = k4;
                    *(typename H16<HT>::x4*)(dvp + d0) = v4;
                }
            }
    }
}

// =====================================================================================
//                                        host
// =====================================================================================
static int attn_check(const char* who, const void* qkv, int batch, int seq, int heads, int head_dim,
                      int mask_mode, const int32_t* row_mask, int cdt) {
    BF_REQUIRE(batch > 0 && seq > 0 && heads > 0, BF_EINVAL, "%s: bad shape", who);
    BF_REQUIRE(head_dim == 32 || head_dim == 48 || head_dim == 64, BF_EINVAL,
               "%s: head_dim must be 32, 48 or 64 (got %d)", who, head_dim);
    BF_REQUIRE(mask_mode >= BF_MASK_NONE && mask_mode <= BF_MASK_QUERY_ROW, BF_EINVAL, "%s: bad mask_mode", who);
    BF_REQUIRE(mask_mode != BF_MASK_QUERY_ROW || row_mask, BF_EINVAL, "%s: row_mask is NULL", who);
    BF_REQUIRE(cdt == BF_F32 || bf_is_half(cdt), BF_EINVAL, "%s: bad cdt", who);
    BF_REQUIRE(bf_aligned16(qkv), BF_EALIGN, "%s: unaligned qkv", who);
    BF_REQUIRE((long)batch * heads < 65536L * 32768L, BF_EINVAL, "%s: batch*heads too large", who);
    return BF_OK;
}

// Which kernel takes which call (16-bit modes; every choice below is the winner of a measured A/B, DESIGN.md section 6 and
// docs/ -- the losing variants and their knobs were removed in round 6):
//   forward   K + V of the sequence fit 80 KB of LDS (<= 512 tokens)      attn_fwd_res_kernel (32-key steps when the last 64-key tile
//                                                                          would be more than half empty)
//             longer, no causal mask                                       attn_fwd_pipe_kernel (attention_pipe.hip)
//             longer, causal                                               attn_fwd_mfma_kernel
//   backward  d = 32, no mask, 3 - 9 blocks                                attn_bwd_fused32p_kernel (nine blocks, >= 2 items per CU) / attn_bwd_fused32_kernel
//             d = 32 / 48 whose images fit LDS, not causal under 128 tokens  attn_bwd_q_kernel + attn_bwd_kv_kernel
//             d = 48 / 64, >= 24 blocks, no causal mask                    attn_bwd_q_pipe_kernel + attn_bwd_kv_pipe_kernel (attention_pipe.hip)
//             everything else beyond one workgroup                         attn_bwd_q_stream_kernel + attn_bwd_kv_stream_kernel
//             short causal sequences (the CLIP text tower would, were it trained)   attn_bwd_mfma_kernel, one chunk
// fp32 (the exact mode): the VALU kernels.
extern "C" int bf_attention_fwd(const void* qkv, void* out, float* lse, const int32_t* row_mask,
                                int batch, int seq, int heads, int head_dim, float scale, int mask_mode,
                                int cdt, bf_stream_t stream) {
    int rc = attn_check("bf_attention_fwd", qkv, batch, seq, heads, head_dim, mask_mode, row_mask, cdt);
    if (rc) return rc;
    BF_REQUIRE(bf_aligned16(out), BF_EALIGN, "bf_attention_fwd: unaligned out");
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = cdt == BF_F16;
#define BY_HALF(M, ...) do { if (f16) M(__VA_ARGS__, f16_t); else M(__VA_ARGS__, bf16_t); } while (0)
#define BY_MASK(M, DD) do { if (mask_mode == BF_MASK_NONE) BY_HALF(M, DD, BF_MASK_NONE); else if (mask_mode == BF_MASK_CAUSAL) BY_HALF(M, DD, BF_MASK_CAUSAL); else BY_HALF(M, DD, BF_MASK_QUERY_ROW); } while (0)
#define BY_DIM(M) do { if (head_dim == 32) BY_MASK(M, 32); else if (head_dim == 48) BY_MASK(M, 48); else BY_MASK(M, 64); } while (0)
    if (bf_is_half(cdt)) {
        const int nq32 = bf_cdiv(seq, 32);
        const int lp = bf_cdiv(seq, 64) * 64;
        const int dp = (head_dim + 31) / 32 * 32;
        const size_t res_lds = (size_t)lp * ((head_dim * 2 + 16) + (dp * 2 + 16));
        if (nq32 <= 16 && res_lds <= 80 * 1024) {
            dim3 rgrid(batch * heads), rblock(nq32 * 64);
            // a sequence whose last 64-key tile would be more than half empty walks 32-key steps
            const bool half_steps = mask_mode != BF_MASK_CAUSAL && (seq % 64) != 0 && (seq % 64) <= 32;
#define FWD_RES(DD, MM, HT)                                                                        \
    do {                                                                                          \
        static BfPerDevice lds_set_dev; size_t& lds_set = lds_set_dev.cur();                      \
        if (res_lds > lds_set) {                                                                  \
            (void)hipFuncSetAttribute((const void*)attn_fwd_res_kernel<DD, MM, 2, HT>,            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)res_lds);  \
            (void)hipFuncSetAttribute((const void*)attn_fwd_res_kernel<DD, MM, 1, HT>,            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)res_lds);  \
            lds_set = res_lds;                                                                    \
        }                                                                                         \
        if (half_steps)                                                                           \
            hipLaunchKernelGGL((attn_fwd_res_kernel<DD, MM, 1, HT>), rgrid, rblock, res_lds, s, (const HT*)qkv, \
                               (HT*)out, lse, row_mask, seq, heads, scale, lp);                \
        else                                                                                      \
            hipLaunchKernelGGL((attn_fwd_res_kernel<DD, MM, 2, HT>), rgrid, rblock, res_lds, s, (const HT*)qkv, \
                               (HT*)out, lse, row_mask, seq, heads, scale, lp);                \
    } while (0)
            BY_DIM(FWD_RES);
#undef FWD_RES
            BF_CHECK_LAUNCH("bf_attention_fwd");
            return BF_OK;
        }
        // beyond LDS: the software-pipelined kernel (attention_pipe.hip) unless the mask is causal
        if (bf_tuning().attn_pipe &&
            bf_attn_fwd_pipe_launch(qkv, out, lse, row_mask, batch, seq, heads, head_dim, scale, mask_mode, cdt, s) == BF_OK) {
            BF_CHECK_LAUNCH("bf_attention_fwd");
            return BF_OK;
        }
        const int chunks = bf_cdiv(nq32, 8);
        const int wpb = bf_cdiv(nq32, chunks);
        dim3 grid(batch * heads, chunks), block(wpb * 64);
#define FWD_MFMA(DD, MM, HT)                                                                      \
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<DD, MM, HT>), grid, block, 0, s, (const HT*)qkv,     \
                       (HT*)out, lse, row_mask, seq, heads, scale, wpb)
        BY_DIM(FWD_MFMA);
#undef FWD_MFMA
    } else {
        dim3 grid(batch * heads, bf_cdiv(seq, 256)), block(256);
#define FWD_SIMPLE(DD)                                                                            \
    hipLaunchKernelGGL((attn_fwd_simple_kernel<float, DD>), grid, block, 0, s, (const float*)qkv,  \
                       (float*)out, lse, row_mask, seq, heads, scale, mask_mode)
        if (head_dim == 32) FWD_SIMPLE(32);
        else if (head_dim == 48) FWD_SIMPLE(48);
        else FWD_SIMPLE(64);
#undef FWD_SIMPLE
    }
    BF_CHECK_LAUNCH("bf_attention_fwd");
    return BF_OK;
}

template <typename T>
static int attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse,
                           const int32_t* row_mask, void* dqkv, float* delta_ws, int batch, int seq,
                           int heads, int head_dim, float scale, int mask_mode, hipStream_t s) {
    dim3 grid(batch * heads, bf_cdiv(seq, 256)), block(256);
#define BWD(DD)                                                                                     \
    do {                                                                                            \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DD>), grid, block, 0, s, (const T*)qkv, (const T*)out, \
                           (const T*)dout, lse, row_mask, (T*)dqkv, delta_ws, seq, heads, scale, mask_mode); \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DD>), grid, block, 0, s, (const T*)qkv, (const T*)dout, \
                           lse, delta_ws, row_mask, (T*)dqkv, seq, heads, scale, mask_mode);         \
    } while (0)
    if (head_dim == 32) BWD(32);
    else if (head_dim == 48) BWD(48);
    else BWD(64);
#undef BWD
    return 0;
}

static size_t bwd_lds_of(int head_dim, int nw) {
    return head_dim == 32 ? BwdCfg<32>::lds_bytes(nw) : head_dim == 48 ? BwdCfg<48>::lds_bytes(nw)
                                                                      : BwdCfg<64>::lds_bytes(nw);
}

extern "C" int bf_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                const int32_t* row_mask, void* dqkv, float* delta_ws, int batch,
                                int seq, int heads, int head_dim, float scale, int mask_mode, int cdt,
                                bf_stream_t stream) {
    int rc = attn_check("bf_attention_bwd", qkv, batch, seq, heads, head_dim, mask_mode, row_mask, cdt);
    if (rc) return rc;
    BF_REQUIRE(lse && delta_ws, BF_EINVAL, "bf_attention_bwd: lse and delta_ws are required");
    BF_REQUIRE(bf_aligned16(out) && bf_aligned16(dout) && bf_aligned16(dqkv), BF_EALIGN,
               "bf_attention_bwd: unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = cdt == BF_F16, half = bf_is_half(cdt);
    if (!half) {
        if (cdt == BF_F32)
            attn_bwd_launch<float>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, head_dim, scale, mask_mode, s);
        BF_CHECK_LAUNCH("bf_attention_bwd");
        return BF_OK;
    }
    const int nq32 = bf_cdiv(seq, 32), lp = nq32 * 32;
    const int dp = (head_dim + 31) / 32 * 32;
    // ---- d = 32 without a mask, 3 - 9 blocks (the 275-token fusion of config A): the one-pass kernels
    if (head_dim == 32 && mask_mode == BF_MASK_NONE && nq32 >= 3 && nq32 <= 9) {
        // the persistent form (one twelve-wave workgroup per CU walking items) for nine-block sequences when there are items to
        // walk: BIFOLD_ATTN_BWD_FUSED_P = 1: from two items per CU (the default); 2: always (tests); 0: never
        int n_cu = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 8) n_cu = 256;
        const int fp_env = bf_tuning().attn_bwd_fused_p;
        if (nq32 == Fused32P::NKW && (fp_env == 2 || (fp_env == 1 && batch * heads >= 2 * n_cu))) {
            const size_t lds_p = Fused32P::lds_bytes(nq32);
            const int grid_p = batch * heads < n_cu ? batch * heads : n_cu;
#define BWD_FUSED_P(HT)                                                                                  \
    do {                                                                                                 \
        static BfPerDevice set_p_dev; size_t& set_p = set_p_dev.cur();                                   \
        if (lds_p > set_p) {                                                                             \
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused32p_kernel<HT>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);           \
            set_p = lds_p;                                                                               \
        }                                                                                                \
        hipLaunchKernelGGL((attn_bwd_fused32p_kernel<HT>), dim3(grid_p), dim3(Fused32P::NWAVES * 64), lds_p, s, \
                           (const HT*)qkv, (const HT*)out, (const HT*)dout, lse, (HT*)dqkv, seq, heads, scale, \
                           batch * heads);                                                               \
    } while (0)
            if (f16) BWD_FUSED_P(f16_t); else BWD_FUSED_P(bf16_t);
#undef BWD_FUSED_P
            BF_CHECK_LAUNCH("bf_attention_bwd");
            return BF_OK;
        }
        const size_t lds_f = Fused32::lds_bytes(nq32);
        dim3 grid(batch * heads), block(256);
#define BWD_FUSED(HT)                                                                                    \
    do {                                                                                                 \
        static BfPerDevice set_f_dev; size_t& set_f = set_f_dev.cur();                                   \
        if (lds_f > set_f) {                                                                             \
            (void)hipFuncSetAttribute((const void*)attn_bwd_fused32_kernel<HT>,                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);           \
            set_f = lds_f;                                                                               \
        }                                                                                                \
        hipLaunchKernelGGL((attn_bwd_fused32_kernel<HT>), grid, block, lds_f, s, (const HT*)qkv,         \
                           (const HT*)out, (const HT*)dout, lse, (HT*)dqkv, seq, heads, scale);          \
    } while (0)
        if (f16) BWD_FUSED(f16_t); else BWD_FUSED(bf16_t);
#undef BWD_FUSED
        BF_CHECK_LAUNCH("bf_attention_bwd");
        return BF_OK;
    }
#define BY_HALF(M, ...) do { if (f16) M(__VA_ARGS__, f16_t); else M(__VA_ARGS__, bf16_t); } while (0)
#define BY_MASK(M, DD) do { if (mask_mode == BF_MASK_NONE) BY_HALF(M, DD, BF_MASK_NONE); else if (mask_mode == BF_MASK_CAUSAL) BY_HALF(M, DD, BF_MASK_CAUSAL); else BY_HALF(M, DD, BF_MASK_QUERY_ROW); } while (0)
#define BY_DIM(M) do { if (head_dim == 32) BY_MASK(M, 32); else if (head_dim == 48) BY_MASK(M, 48); else BY_MASK(M, 64); } while (0)
    // ---- sequences whose K + V / Q + dO fit LDS, d = 32 / 48: two barrier-free kernels (dQ with the query on the lane, then
    // dK / dV with the key on the lane), ONE workgroup per (batch, head): three waves x three blocks in the q kernel, four
    // waves in the kv kernel (d = 32: its 128-register build, four workgroups per CU).  d = 64 goes to the streamed pair
    // (300 x 64: 99 vs 134 us resident); short causal sequences to the single kernel (72 vs 116 us at 77 tokens).
    {
        const int pad = head_dim == 32 ? 0 : 16, row = head_dim * 2 + pad;          // ResImg<D>: d = 32 rows are unpadded
        const size_t lds_q = (size_t)lp * ((dp * 2 + pad) + row);
        const size_t lds_kv = 2 * ((size_t)lp * row + 64) + 3 * (size_t)lp * 4;
        if (head_dim != 64 && nq32 <= 16 && lds_q <= 96 * 1024 && lds_kv <= 96 * 1024 &&
            !(mask_mode == BF_MASK_CAUSAL && seq < 128)) {
            const int wpb = nq32 < 3 ? nq32 : 3, wpb_kv = nq32 < 4 ? nq32 : 4;
            dim3 grid(batch * heads, 1), block(wpb * 64);
            dim3 grid_kv(batch * heads, 1), block_kv(wpb_kv * 64);
#define BWD_SPLIT(DD, MM, HT)                                                                       \
    do {                                                                                            \
        static BfPerDevice set_q_dev, set_kv_dev; size_t &set_q = set_q_dev.cur(), &set_kv = set_kv_dev.cur(); \
        if (lds_q > set_q) {                                                                        \
            (void)hipFuncSetAttribute((const void*)attn_bwd_q_kernel<DD, MM, HT>,                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);      \
            set_q = lds_q;                                                                          \
        }                                                                                           \
        if (lds_kv > set_kv) {                                                                      \
            (void)hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<DD, MM, HT, (DD == 32 ? 4 : 1)>, \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);     \
            set_kv = lds_kv;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL((attn_bwd_q_kernel<DD, MM, HT>), grid, block, lds_q, s, (const HT*)qkv,  \
                           (const HT*)out, (const HT*)dout, lse, row_mask, (HT*)dqkv,               \
                           delta_ws, seq, heads, scale, lp);                                        \
        hipLaunchKernelGGL((attn_bwd_kv_kernel<DD, MM, HT, (DD == 32 ? 4 : 1)>), grid_kv, block_kv, lds_kv, s, (const HT*)qkv, \
                           (const HT*)dout, lse, (const float*)delta_ws, row_mask,                  \
                           (HT*)dqkv, seq, heads, scale);                                           \
    } while (0)
            BY_DIM(BWD_SPLIT);
#undef BWD_SPLIT
            BF_CHECK_LAUNCH("bf_attention_bwd");
            return BF_OK;
        }
    }
    // ---- d = 48 / 64 without a causal mask from 24 blocks: the software-pipelined pair (attention_pipe.hip); at 576 x 64 the
    // streamed pair below is faster (364 vs 523 us: three-workgroup grids with idle waves)
    if (bf_tuning().attn_pipe && nq32 >= 24 &&
        bf_attn_bwd_pipe_launch(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, head_dim, scale,
                                mask_mode, cdt, s) == BF_OK) {
        BF_CHECK_LAUNCH("bf_attention_bwd");
        return BF_OK;
    }
    // ---- everything else beyond one workgroup: the same pair with the resident operand streamed through LDS in chunks
    if (head_dim != 32 || nq32 >= 9) {
        // rows per chunk, waves per workgroup: d = 32 fits 128 registers (16 waves per CU as 2 x 8);
        // d = 48 / 64 needs ~190, so 8 waves per CU as 2 workgroups x 4 (one stages while one computes)
        const int CH = head_dim == 32 ? StreamCfg<32>::CH : StreamCfg<64>::CH;       // see StreamCfg
        const int wpb = head_dim == 32 ? StreamCfg<32>::WPB : StreamCfg<64>::WPB;
        const int row = head_dim * 2 + 16;
        const size_t lds_q = (size_t)CH * ((dp * 2 + 16) + row);
        const size_t lds_kv = 2 * ((size_t)CH * row + 64) + 3 * (size_t)CH * 4;
        dim3 grid(batch * heads, bf_cdiv(nq32, wpb)), block(wpb * 64);
#define BWD_STREAM(DD, MM, HT)                                                                      \
    do {                                                                                            \
        static BfPerDevice set_q_dev, set_kv_dev; size_t &set_q = set_q_dev.cur(), &set_kv = set_kv_dev.cur(); \
        if (lds_q > set_q) {                                                                        \
            (void)hipFuncSetAttribute((const void*)attn_bwd_q_stream_kernel<DD, MM, HT>,            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);      \
            set_q = lds_q;                                                                          \
        }                                                                                           \
        if (lds_kv > set_kv) {                                                                      \
            (void)hipFuncSetAttribute((const void*)attn_bwd_kv_stream_kernel<DD, MM, HT>,           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);     \
            set_kv = lds_kv;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL((attn_bwd_q_stream_kernel<DD, MM, HT>), grid, block, lds_q, s, (const HT*)qkv, \
                           (const HT*)out, (const HT*)dout, lse, row_mask, (HT*)dqkv,               \
                           delta_ws, seq, heads, scale);                                            \
        hipLaunchKernelGGL((attn_bwd_kv_stream_kernel<DD, MM, HT>), grid, block, lds_kv, s, (const HT*)qkv, \
                           (const HT*)dout, lse, (const float*)delta_ws, row_mask,                  \
                           (HT*)dqkv, seq, heads, scale);                                           \
    } while (0)
        BY_DIM(BWD_STREAM);
#undef BWD_STREAM
        BF_CHECK_LAUNCH("bf_attention_bwd");
        return BF_OK;
    }
    // ---- what is left is d = 32 under 9 blocks that the resident pair did not take (causal, under 128 tokens): the single
    // kernel, whole sequence in one workgroup
    {
        const int nw = nq32;
        const size_t lds = bwd_lds_of(head_dim, nw);
        BF_REQUIRE(nw <= 16 && lds <= 160 * 1024, BF_EINVAL, "bf_attention_bwd: no kernel for this shape");
        dim3 grid(batch * heads, 1), block(nw * 64);
#define BWD_MFMA(DD, MM, HT)                                                                        \
    do {                                                                                            \
        static BfPerDevice lds_set_dev; size_t& lds_set = lds_set_dev.cur();                        \
        if (lds > lds_set) {                                                                        \
            (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<DD, MM, HT>,                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
            lds_set = lds;                                                                          \
        }                                                                                           \
        hipLaunchKernelGGL((attn_bwd_mfma_kernel<DD, MM, HT>), grid, block, lds, s, (const HT*)qkv, \
                           (const HT*)out, (const HT*)dout, lse, row_mask, (HT*)dqkv,               \
                           (float*)nullptr, seq, heads, scale);                                     \
    } while (0)
        BY_DIM(BWD_MFMA);
#undef BWD_MFMA
    }
#undef BY_DIM
#undef BY_MASK
#undef BY_HALF
    BF_CHECK_LAUNCH("bf_attention_bwd");
    return BF_OK;
}

// =====================================================================================
//   nn.Dropout on the attention probabilities (ref:bifold/models/transformer.py:18,36; p > 0 only in training)
// =====================================================================================
// No shipped configuration sets `dropout` above 0 (ref:bifold/conf/model/*.yaml), so this is the correct-first form: the
// VALU kernels (fp32 math, any storage type) with the replayable mask of common.h on P.  The flash kernels above have no
// dropout; callers reach these entry points only when p > 0.
template <typename T>
static void attn_dropout_fwd_launch(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq,
                                    int heads, int head_dim, float scale, int mask_mode, unsigned thr, float inv_keep,
                                    unsigned long long seed, hipStream_t s) {
    dim3 grid(batch * heads, bf_cdiv(seq, 256)), block(256);
#define FWD_DROP(DD)                                                                                       \
    hipLaunchKernelGGL((attn_fwd_simple_kernel<T, DD, true>), grid, block, 0, s, (const T*)qkv, (T*)out, lse, row_mask, \
                       seq, heads, scale, mask_mode, thr, inv_keep, seed)
    if (head_dim == 32) FWD_DROP(32); else if (head_dim == 48) FWD_DROP(48); else FWD_DROP(64);
#undef FWD_DROP
}
template <typename T>
static void attn_dropout_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse,
                                    const int32_t* row_mask, void* dqkv, float* delta_ws, int batch, int seq, int heads,
                                    int head_dim, float scale, int mask_mode, unsigned thr, float inv_keep,
                                    unsigned long long seed, hipStream_t s) {
    dim3 grid(batch * heads, bf_cdiv(seq, 256)), block(256);
#define BWD_DROP(DD)                                                                                        \
    do {                                                                                                    \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DD, true>), grid, block, 0, s, (const T*)qkv, (const T*)out, \
                           (const T*)dout, lse, row_mask, (T*)dqkv, delta_ws, seq, heads, scale, mask_mode, thr, inv_keep, seed); \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DD, true>), grid, block, 0, s, (const T*)qkv, (const T*)dout, \
                           lse, delta_ws, row_mask, (T*)dqkv, seq, heads, scale, mask_mode, thr, inv_keep, seed); \
    } while (0)
    if (head_dim == 32) BWD_DROP(32); else if (head_dim == 48) BWD_DROP(48); else BWD_DROP(64);
#undef BWD_DROP
}

extern "C" int bf_attention_dropout_fwd(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq,
                                        int heads, int head_dim, float scale, int mask_mode, int cdt, float p,
                                        unsigned long long seed, bf_stream_t stream) {
    int rc = attn_check("bf_attention_dropout_fwd", qkv, batch, seq, heads, head_dim, mask_mode, row_mask, cdt);
    if (rc) return rc;
    BF_REQUIRE(bf_aligned16(out), BF_EALIGN, "bf_attention_dropout_fwd: unaligned out");
    BF_REQUIRE(p > 0.f && p < 1.f, BF_EINVAL, "bf_attention_dropout_fwd: p must be in (0, 1) (got %f)", (double)p);
    BF_REQUIRE((unsigned long long)batch * heads * seq * seq < (1ull << 62), BF_EINVAL, "bf_attention_dropout_fwd: too many scores");
    const unsigned thr = dropout_threshold(p);
    const float inv_keep = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
    if (cdt == BF_F32) attn_dropout_fwd_launch<float>(qkv, out, lse, row_mask, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    else if (cdt == BF_F16) attn_dropout_fwd_launch<f16_t>(qkv, out, lse, row_mask, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    else attn_dropout_fwd_launch<bf16_t>(qkv, out, lse, row_mask, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    BF_CHECK_LAUNCH("bf_attention_dropout_fwd");
    return BF_OK;
}

extern "C" int bf_attention_dropout_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                        const int32_t* row_mask, void* dqkv, float* delta_ws, int batch, int seq, int heads,
                                        int head_dim, float scale, int mask_mode, int cdt, float p, unsigned long long seed,
                                        bf_stream_t stream) {
    int rc = attn_check("bf_attention_dropout_bwd", qkv, batch, seq, heads, head_dim, mask_mode, row_mask, cdt);
    if (rc) return rc;
    BF_REQUIRE(lse && delta_ws, BF_EINVAL, "bf_attention_dropout_bwd: lse and delta_ws are required");
    BF_REQUIRE(bf_aligned16(out) && bf_aligned16(dout) && bf_aligned16(dqkv), BF_EALIGN, "bf_attention_dropout_bwd: unaligned pointer");
    BF_REQUIRE(p > 0.f && p < 1.f, BF_EINVAL, "bf_attention_dropout_bwd: p must be in (0, 1) (got %f)", (double)p);
    const unsigned thr = dropout_threshold(p);
    const float inv_keep = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
    if (cdt == BF_F32) attn_dropout_bwd_launch<float>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    else if (cdt == BF_F16) attn_dropout_bwd_launch<f16_t>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    else attn_dropout_bwd_launch<bf16_t>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, head_dim, scale, mask_mode, thr, inv_keep, seed, s);
    BF_CHECK_LAUNCH("bf_attention_dropout_bwd");
    return BF_OK;
}
