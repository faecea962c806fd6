// GEMM with fused epilogue for gfx950.
//
//   C = epilogue(alpha * op(A) * op(B))        (see include/bifold_hip.h)
//
// Two kernels share one LDS-staged, fully coalesced epilogue:
//   * gemm_bf16_kernel : bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//       128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
//       K-contiguous operands ([M,K] / [N,K]) sit in LDS as 128-byte rows with an
//       XOR-swizzled 16-byte chunk index and are read with ds_read_b128;
//       K-strided operands ([K,M] / [K,N]) keep their memory order in LDS and are
//       read with ds_read_b64_tr_b16 (the hardware transposed read), so dX = dY W and
//       dW = dY^T X need no transposed copies in HBM.
//   * gemm_f32_kernel  : fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
//       128x128x16 tile.  This is the parity ("exact") mode.
// Register-staged double buffering: the global loads of tile t+1 are issued before the
// MFMAs of tile t and written to the other LDS buffer afterwards.
// Block -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a contiguous
// range of output tiles with the N index fastest, so the blocks that share an A row
// panel run on one XCD back to back.
#include "common.h"
#include "tuning.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

struct EpiParams {
    void* c;
    void* c2;
    void* pre;
    const void* aux;
    const float* bias;
    const float* residual;
    int m, n;
    int ldc, ld_aux, ld_res;
    int resid_mod;
    int rows_in, rows_out, row_off;
    int act;
    int accumulate;
    float alpha;
    float* colsum;      // fused column sums of the stored result (atomicAdd into a zeroed [n]) or null
    const void* lr_u;   // low-rank addend (persistent kernel, EPI 5): u [m, lr_k], w [n, lr_k], 16-bit, dense
    const void* lr_w;
    int lr_k;
    int store_policy;   // persistent kernel: bit 0 / 1 / 2 = non-temporal (`nt`) output stores in the plain-bf16 /
                        // general / GELU-grad epilogue.  Output tiles are written once and read by a LATER kernel;
                        // stored plainly they push the operand panels out of the XCD's L2 (fusion QKV: 237 MB of
                        // operand fetches per launch against 74 MB algorithmic, 186 MB with nt)
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// 16 bytes to global memory, optionally with the streaming hint
__device__ __forceinline__ void store16(void* p, uint4 q, bool nt) {
    if (nt) {
        const u32x4_t v = {q.x, q.y, q.z, q.w};
        __builtin_nontemporal_store(v, (u32x4_t*)p);
    } else {
        *(uint4*)p = q;
    }
}

// type of the optional 16-bit copy of an fp32 result (`c2`): the call's 16-bit operand type (exact mode: unused)
template <typename TD> using Half2Of = std::conditional_t<std::is_same<TD, float>::value, bf16_t, TD>;

template <typename TC, typename TD>
__device__ __forceinline__ void epilogue_store4(const EpiParams& ep, int m, int n, float4 v) {
    v.x *= ep.alpha; v.y *= ep.alpha; v.z *= ep.alpha; v.w *= ep.alpha;
    if (ep.bias) {
        const float4 b = *(const float4*)(ep.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    size_t orow = (size_t)m;
    if (ep.rows_in > 0) orow = (size_t)(m / ep.rows_in) * ep.rows_out + ep.row_off + (m % ep.rows_in);
    if (ep.act == BF_ACT_QUICK_GELU || ep.act == BF_ACT_ERF_GELU || ep.act == BF_ACT_TANH_GELU) {
        if (ep.pre) Vec4<TD>::store((TD*)ep.pre + orow * ep.ldc + n, v);
        using AM = ActMath<TD>;
        if (ep.act == BF_ACT_QUICK_GELU) {
            v.x = AM::quick(v.x); v.y = AM::quick(v.y); v.z = AM::quick(v.z); v.w = AM::quick(v.w);
        } else if (ep.act == BF_ACT_TANH_GELU) {
            v.x = AM::tanh_gelu(v.x); v.y = AM::tanh_gelu(v.y); v.z = AM::tanh_gelu(v.z); v.w = AM::tanh_gelu(v.w);
        } else {
            v.x = AM::erf(v.x); v.y = AM::erf(v.y); v.z = AM::erf(v.z); v.w = AM::erf(v.w);
        }
    } else if (ep.act == BF_ACT_ERF_GELU_SAVEGRAD) {
        using AM = ActMath<TD>;
        float4 gp;
        AM::erf_both(v.x, v.x, gp.x); AM::erf_both(v.y, v.y, gp.y);
        AM::erf_both(v.z, v.z, gp.z); AM::erf_both(v.w, v.w, gp.w);
        Vec4<TD>::store((TD*)ep.pre + orow * ep.ldc + n, gp);
    } else if (ep.act == BF_ACT_MUL_AUX) {
        const float4 a = Vec4<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n);
        v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
    } else if (ep.act == BF_ACT_ERF_GELU_GRAD) {
        using AM = ActMath<TD>;
        const float4 a = Vec4<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n);
        v.x *= AM::erf_grad(a.x); v.y *= AM::erf_grad(a.y);
        v.z *= AM::erf_grad(a.z); v.w *= AM::erf_grad(a.w);
    } else if (ep.act == BF_ACT_TANH_GELU_GRAD) {
        using AM = ActMath<TD>;
        const float4 a = Vec4<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n);
        v.x *= AM::tanh_grad(a.x); v.y *= AM::tanh_grad(a.y);
        v.z *= AM::tanh_grad(a.z); v.w *= AM::tanh_grad(a.w);
    }
    if (ep.residual) {
        const size_t rr = ep.resid_mod > 0 ? (size_t)(m % ep.resid_mod) : orow;
        const float4 r = *(const float4*)(ep.residual + rr * ep.ld_res + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    TC* cp = (TC*)ep.c + orow * ep.ldc + n;
    if (ep.accumulate) {
        const float4 o = Vec4<TC>::load(cp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    Vec4<TC>::store(cp, v);
    if (ep.c2) Vec4<Half2Of<TD>>::store((Half2Of<TD>*)ep.c2 + orow * ep.ldc + n, v);
}

// epilogue_store4 with the lane's 4 bias values held in registers (see epilogue_store8 on why the bias is not loaded
// per row) and the persistent kernel's store policy: the fp32 outputs of that kernel leave as ONE 16-byte store per lane,
// 16 lanes per row -- a store instruction is 4 rows x 256 contiguous bytes.
template <typename TC, typename TD>
__device__ __forceinline__ void epilogue_store4b(const EpiParams& ep, int m, int n, float4 v, float4 b) {
    v.x = fmaf(v.x, ep.alpha, b.x); v.y = fmaf(v.y, ep.alpha, b.y);
    v.z = fmaf(v.z, ep.alpha, b.z); v.w = fmaf(v.w, ep.alpha, b.w);
    size_t orow = (size_t)m;
    if (ep.rows_in > 0) orow = (size_t)(m / ep.rows_in) * ep.rows_out + ep.row_off + (m % ep.rows_in);
    if (ep.act == BF_ACT_QUICK_GELU || ep.act == BF_ACT_ERF_GELU || ep.act == BF_ACT_TANH_GELU) {
        if (ep.pre) Vec4<TD>::store((TD*)ep.pre + orow * ep.ldc + n, v);
        using AM = ActMath<TD>;
        if (ep.act == BF_ACT_QUICK_GELU) {
            v.x = AM::quick(v.x); v.y = AM::quick(v.y); v.z = AM::quick(v.z); v.w = AM::quick(v.w);
        } else if (ep.act == BF_ACT_TANH_GELU) {
            v.x = AM::tanh_gelu(v.x); v.y = AM::tanh_gelu(v.y); v.z = AM::tanh_gelu(v.z); v.w = AM::tanh_gelu(v.w);
        } else {
            v.x = AM::erf(v.x); v.y = AM::erf(v.y); v.z = AM::erf(v.z); v.w = AM::erf(v.w);
        }
    } else if (ep.act == BF_ACT_ERF_GELU_SAVEGRAD) {
        using AM = ActMath<TD>;
        float4 gp;
        AM::erf_both(v.x, v.x, gp.x); AM::erf_both(v.y, v.y, gp.y);
        AM::erf_both(v.z, v.z, gp.z); AM::erf_both(v.w, v.w, gp.w);
        Vec4<TD>::store((TD*)ep.pre + orow * ep.ldc + n, gp);
    } else if (ep.act == BF_ACT_MUL_AUX || ep.act == BF_ACT_ERF_GELU_GRAD || ep.act == BF_ACT_TANH_GELU_GRAD) {
        using AM = ActMath<TD>;
        const float4 a = Vec4<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n);
        if (ep.act == BF_ACT_MUL_AUX) { v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w; }
        else if (ep.act == BF_ACT_ERF_GELU_GRAD) {
            v.x *= AM::erf_grad(a.x); v.y *= AM::erf_grad(a.y); v.z *= AM::erf_grad(a.z); v.w *= AM::erf_grad(a.w);
        } else {
            v.x *= AM::tanh_grad(a.x); v.y *= AM::tanh_grad(a.y); v.z *= AM::tanh_grad(a.z); v.w *= AM::tanh_grad(a.w);
        }
    }
    if (ep.residual) {
        const size_t rr = ep.resid_mod > 0 ? (size_t)(m % ep.resid_mod) : orow;
        const float4 r = *(const float4*)(ep.residual + rr * ep.ld_res + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    TC* cp = (TC*)ep.c + orow * ep.ldc + n;
    if (ep.accumulate) {
        const float4 o = Vec4<TC>::load(cp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if constexpr (std::is_same<TC, float>::value) {
        const f32x4 q = {v.x, v.y, v.z, v.w};
        if (ep.store_policy & 2) __builtin_nontemporal_store(q, (f32x4*)cp); else *(f32x4*)cp = q;
    } else {
        Vec4<TC>::store(cp, v);
    }
    if (ep.c2) Vec4<Half2Of<TD>>::store((Half2Of<TD>*)ep.c2 + orow * ep.ldc + n, v);
}

// 8 consecutive columns of one output row.  A store instruction costs the issuing wave the
// same whatever its width (measured on gfx950: ~300 cycles per store and wave when every CU
// stores at once, 512-byte and 1-KB stores alike), so the epilogue moves 16 bytes per lane per
// instruction wherever the output type allows it.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void store_nt(float* p, const float (&v)[8], bool nt) {
        float4 a = make_float4(v[0], v[1], v[2], v[3]), b = make_float4(v[4], v[5], v[6], v[7]);
        store16(p, *(const uint4*)&a, nt);
        store16(p + 4, *(const uint4*)&b, nt);
    }
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void store_nt(bf16_t* p, const float (&v)[8], bool nt) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
        store16(p, *(const uint4*)&r, nt);
    }
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const bf16x8 r = *(const bf16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
        *(bf16x8*)p = r;
    }
};

template <> struct Vec8<f16_t> {
    static __device__ __forceinline__ void store_nt(f16_t* p, const float (&v)[8], bool nt) {
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16_t)v[e];
        store16(p, *(const uint4*)&r, nt);
    }
    static __device__ __forceinline__ void load(const f16_t* p, float (&v)[8]) {
        const f16x8 r = *(const f16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float (&v)[8]) {
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16_t)v[e];
        *(f16x8*)p = r;
    }
};

// bias8: this lane's 8 bias values, loaded ONCE per tile by the caller (zeros without a bias).  A load inside
// the per-row loop makes every iteration wait -- through the in-order vmcnt -- for the previous iteration's
// stores as well: 16 exposed store latencies per tile.
template <typename TC, typename TD>
__device__ __forceinline__ void epilogue_store8(const EpiParams& ep, int m, int n, float (&v)[8],
                                                const float (&bias8)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], ep.alpha, bias8[e]);
    size_t orow = (size_t)m;
    if (ep.rows_in > 0) orow = (size_t)(m / ep.rows_in) * ep.rows_out + ep.row_off + (m % ep.rows_in);
    if (ep.act == BF_ACT_QUICK_GELU || ep.act == BF_ACT_ERF_GELU || ep.act == BF_ACT_TANH_GELU) {
        if (ep.pre) Vec8<TD>::store_nt((TD*)ep.pre + orow * ep.ldc + n, v, ep.store_policy & 2);
        using AM = ActMath<TD>;
        if (ep.act == BF_ACT_QUICK_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = AM::quick(v[e]);
        } else if (ep.act == BF_ACT_TANH_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = AM::tanh_gelu(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = AM::erf(v[e]);
        }
    } else if (ep.act == BF_ACT_ERF_GELU_SAVEGRAD) {
        using AM = ActMath<TD>;
        float gp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) AM::erf_both(v[e], v[e], gp[e]);
        Vec8<TD>::store_nt((TD*)ep.pre + orow * ep.ldc + n, gp, ep.store_policy & 2);
    } else if (ep.act == BF_ACT_MUL_AUX) {
        float a[8];
        Vec8<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a[e];
    } else if (ep.act == BF_ACT_ERF_GELU_GRAD) {
        using AM = ActMath<TD>;
        float a[8];
        Vec8<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= AM::erf_grad(a[e]);
    } else if (ep.act == BF_ACT_TANH_GELU_GRAD) {
        using AM = ActMath<TD>;
        float a[8];
        Vec8<TD>::load((const TD*)ep.aux + (size_t)m * ep.ld_aux + n, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= AM::tanh_grad(a[e]);
    }
    if (ep.residual) {
        const size_t rr = ep.resid_mod > 0 ? (size_t)(m % ep.resid_mod) : orow;
        float r[8];
        Vec8<float>::load(ep.residual + rr * ep.ld_res + n, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    TC* cp = (TC*)ep.c + orow * ep.ldc + n;
    if (ep.accumulate) {
        float o[8];
        Vec8<TC>::load(cp, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += o[e];
    }
    Vec8<TC>::store_nt(cp, v, ep.store_policy & 2);
    if (ep.c2) Vec8<Half2Of<TD>>::store_nt((Half2Of<TD>*)ep.c2 + orow * ep.ldc + n, v, ep.store_policy & 2);
}

#define CT_LD 132   // fp32 staging tile leading dimension (128 + 4 pad)

// Drain the 128x128 fp32 tile staged in LDS with coalesced 16-byte accesses.
template <typename TC, typename TD>
__device__ __forceinline__ void tile_epilogue(const float* ct, const EpiParams& ep, int m0, int n0,
                                              float* slab /* split-K partial or null */) {
    for (int idx = threadIdx.x; idx < 128 * 32; idx += 256) {
        const int r = idx >> 5, c4 = idx & 31;
        const int m = m0 + r, n = n0 + c4 * 4;
        if (m < ep.m && n < ep.n) {
            const float4 v = *(const float4*)(ct + r * CT_LD + c4 * 4);
            if (slab) *(float4*)(slab + (size_t)m * ep.n + n) = v;
            else epilogue_store4<TC, TD>(ep, m, n, v);
        }
    }
}

// Tile order inside an XCD's contiguous range: GROUP_M row tiles form a group; inside a
// group the ROW tile index runs fastest, then the column tile.  The ~64 blocks co-resident
// on an XCD then cover a GROUP_M x (64/GROUP_M) patch of output tiles, so their A row panels
// and B column panels together fit the XCD's 4 MB L2 (with the column index fastest they
// span every column tile, i.e. the whole weight matrix, which thrashes L2 for N >= 2048).
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * group_m;
    const int gsz = min(group_m, tiles_m - first_m);
    const int loc = t - grp * per_group;
    tm = first_m + loc % gsz;
    tn = loc / gsz;
}

// XCD-aware bijective remap of a 1-D block id (guide T1, non-multiple-of-8 safe).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// =====================================================================================
//                                   bf16 MFMA kernel
// =====================================================================================
struct GemmDims {
    int m, n, k;        // problem
    int lda, ldb;
    int tiles_m, tiles_n;
    int k_per_split;    // multiple of the K tile
    int group_m;
    // persistent BK = 64 kernel: the tiles of the under-filled last round, split along K inside the launch
    int tail_tiles;     // 0: off; else the LAST tail_tiles tiles (in item order) are cut into tail_parts K chunks
    int tail_parts;     // S: chunks per tail tile (every chunk non-empty)
    int tail_kt;        // K tiles (of 64) per chunk
    float* tail_ws;     // [tail_tiles][S][8 waves][32][64 lanes] f32x4: the chunks' accumulators, as the registers hold them
};

// ---- global -> register tile loads (4 x 16 B per thread per operand) -------------------
// KC: element (r,k) at p[r*ld + k];  KS: element (r,k) at p[k*ld + r].
template <bool KS, typename H>
__device__ __forceinline__ void bf16_tile_load(uint4 (&reg)[4], const H* __restrict__ p, int ld,
                                               int r0, int rtot, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!KS) {
            const int row = q >> 3, ch = q & 7;
            const int r = r0 + row, k = k0 + ch * 8;
            if (r < rtot && k < kend) v = *(const uint4*)(p + (size_t)r * ld + k);
        } else {
            const int krow = q >> 4, ch = q & 15;
            const int k = k0 + krow, r = r0 + ch * 8;
            if (k < kend && r < rtot) v = *(const uint4*)(p + (size_t)k * ld + r);
        }
        reg[i] = v;
    }
}
__device__ __forceinline__ int ks_swz(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }

template <bool KS>
__device__ __forceinline__ void bf16_tile_store(const uint4 (&reg)[4], char* lds) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i;
        int off;
        if (!KS) {
            const int row = q >> 3, ch = q & 7;
            off = row * 128 + ((ch ^ (row & 7)) << 4);
        } else {
            const int krow = q >> 4, ch = q & 15;
            off = krow * 256 + ((ch ^ ks_swz(krow)) << 4);
        }
        *(uint4*)(lds + off) = reg[i];
    }
}

// ---- LDS -> MFMA fragment (16 rows x 32 k), rows [rbase, rbase+16), k-substep s -------
template <bool KS, typename H>
__device__ __forceinline__ typename H16<H>::x8 bf16_frag(const char* lds, int rbase, int s, int lane) {
    typedef typename H16<H>::x4 H4;
    typedef typename H16<H>::x8 H8;
    if (!KS) {
        const int row = rbase + (lane & 15);
        const int ch = s * 4 + (lane >> 4);
        return *(const H8*)(lds + row * 128 + ((ch ^ (row & 7)) << 4));
    } else {
        const int gl = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int kk = s * 32 + gl * 8 + q;
        const int rc = (rbase >> 3) + (p >> 1);
        const int sw = ks_swz(kk);                       // identical for kk and kk+4
        const int o0 = kk * 256 + ((rc ^ sw) << 4) + ((p & 1) << 3);
        const H4 lo = lds_read_tr16<H>(lds + o0);
        const H4 hi = lds_read_tr16<H>(lds + o0 + 4 * 256);
        H8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

template <bool A_KS, bool B_KS, typename TC, typename H>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const H* __restrict__ A,
                                                           const H* __restrict__ B,
                                                           GemmDims d, EpiParams ep,
                                                           float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][A 16 KB | B 16 KB] x 2 = 64 KB, reused as the fp32 staging tile (66 KB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles = d.tiles_m * d.tiles_n;
    const int z = blockIdx.x / tiles;
    const int t = xcd_remap(blockIdx.x - z * tiles, tiles);
    const int tm = t / d.tiles_n, tn = t - tm * d.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = z * d.k_per_split;
    const int kend = min(d.k, kbeg + d.k_per_split);
    const int nk = (kend - kbeg + 63) >> 6;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    bf16_tile_load<A_KS>(ra, A, d.lda, m0, d.m, kbeg, kend);
    bf16_tile_load<B_KS>(rb, B, d.ldb, n0, d.n, kbeg, kend);
    bf16_tile_store<A_KS>(ra, smem);
    bf16_tile_store<B_KS>(rb, smem + 16384);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const char* la = smem + (kt & 1) * 32768;
        const char* lb = la + 16384;
        const bool more = kt + 1 < nk;
        if (more) {
            bf16_tile_load<A_KS>(ra, A, d.lda, m0, d.m, kbeg + (kt + 1) * 64, kend);
            bf16_tile_load<B_KS>(rb, B, d.ldb, n0, d.n, kbeg + (kt + 1) * 64, kend);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            typename H16<H>::x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = bf16_frag<A_KS, H>(la, wm * 64 + i * 16, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = bf16_frag<B_KS, H>(lb, wn * 64 + j * 16, s, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma_16x16x32(af[i], bfr[j], acc[i][j]);
        }
        if (more) {
            char* na = smem + ((kt + 1) & 1) * 32768;
            bf16_tile_store<A_KS>(ra, na);
            bf16_tile_store<B_KS>(rb, na + 16384);
        }
        __syncthreads();
    }

    // stage accumulators: D layout col = lane&15, row = (lane>>4)*4 + reg
    float* ct = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * CT_LD + wn * 64 + j * 16 + (lane & 15)] =
                    acc[i][j][r];
    __syncthreads();
    float* slab = slabs ? slabs + (size_t)z * ep.m * ep.n : nullptr;
    tile_epilogue<TC, H>(ct, ep, m0, n0, slab);
}

// =====================================================================================
//                        bf16 MFMA kernel, LDS-DMA staged (main path)
// =====================================================================================
// Same tile geometry and LDS images as gemm_bf16_kernel, but operand tiles go HBM -> LDS
// with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write: the ds_write_b128
// staging pass alone kept the LDS port busier than the MFMAs).  An LDS-DMA wave-instruction
// writes 1 KB LINEARLY (M0 base + lane*16), so the XOR swizzle is applied to the per-lane
// SOURCE address: lane l of a piece fetches the chunk that belongs in slot l of the swizzled
// image.  Buffer descriptors carry the exact byte extent of A and B, so rows past the end of
// a matrix read as zeros (M/N tails, and the token tail of dW = dY^T X).
// Epilogue: the MFMA operand roles are swapped (D'[n][m] = sum_k B[n][k] A[m][k]) so each
// lane owns 4 CONSECUTIVE output columns of one row: bias / activation / residual / store
// are 16-byte accesses straight from the accumulators, with no LDS staging and no barrier.
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <bool KS>
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int r0,
                                         int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = wave * 4 + i;                       // 1 KB piece of the 16 KB tile
        unsigned voff;
        if (!KS) {
            const int row = pc * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ (row & 7);
            voff = (unsigned)(((size_t)(r0 + row) * ld + k0 + ch * 8) * 2);
        } else {
            const int krow = pc * 4 + (lane >> 4);
            const int ch = (lane & 15) ^ ks_swz(krow);
            voff = (unsigned)(((size_t)(k0 + krow) * ld + r0 + ch * 8) * 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(lds_tile + pc * 1024), 16, voff, 0, 0, 0);
    }
}

template <bool A_KS, bool B_KS, typename TC, typename H>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dma_kernel(const H* __restrict__ A,
                                                               const H* __restrict__ B,
                                                               GemmDims d, EpiParams ep,
                                                               float* __restrict__ slabs,
                                                               unsigned a_bytes, unsigned b_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A 16 KB | B 16 KB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles = d.tiles_m * d.tiles_n;
    const int z = blockIdx.x / tiles;
    const int t = xcd_remap(blockIdx.x - z * tiles, tiles);
    int tm, tn;
    tile_coords(t, d.tiles_m, d.tiles_n, d.group_m, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = z * d.k_per_split;
    const int kend = min(d.k, kbeg + d.k_per_split);
    const int nk = (kend - kbeg + 63) >> 6;

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

    f32x4 acc[4][4];          // acc[j][i]: rows n (j tile), cols m (i tile)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    dma_tile<A_KS>(ra, smem, d.lda, m0, kbeg, wave, lane);
    dma_tile<B_KS>(rb, smem + 16384, d.ldb, n0, kbeg, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const char* la = smem + (kt & 1) * 32768;
        const char* lb = la + 16384;
        if (kt + 1 < nk) {
            char* na = smem + ((kt + 1) & 1) * 32768;
            dma_tile<A_KS>(ra, na, d.lda, m0, kbeg + (kt + 1) * 64, wave, lane);
            dma_tile<B_KS>(rb, na + 16384, d.ldb, n0, kbeg + (kt + 1) * 64, wave, lane);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            typename H16<H>::x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = bf16_frag<A_KS, H>(la, wm * 64 + i * 16, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = bf16_frag<B_KS, H>(lb, wn * 64 + j * 16, s, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = mfma_16x16x32(bfr[j], af[i], acc[j][i]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // D' layout: col = lane&15 -> m, row = (lane>>4)*4 + reg -> n : 4 consecutive n per lane
    float* slab = slabs ? slabs + (size_t)z * ep.m * ep.n : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m < ep.m && n < ep.n) {
                const float4 v = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
                if (slab) *(float4*)(slab + (size_t)m * ep.n + n) = v;
                else epilogue_store4<TC, H>(ep, m, n, v);
            }
        }
    }
}

// =====================================================================================
//                      BK = 32 LDS images (used by gemm_bf16_k32_kernel)
// =====================================================================================
// K-contiguous tiles are 64-byte rows; chunk c of row r sits in slot c ^ f(r), f = (-(r>>2))&3,
// which is conflict-free for ds_read_b128's 16-lane groups; K-strided tiles keep the 256-byte
// k-row image of the BK = 64 kernels (32 k-rows per 128-wide sub-tile).

__device__ __forceinline__ int kc32_swz(int row) { return (-(row >> 2)) & 3; }

template <bool KS>
__device__ __forceinline__ void dma_pieces32(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int r0,
                                             int k0, int first_piece, int n_pieces, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < n_pieces) {
            const int pc = first_piece + i;
            unsigned voff;
            if (!KS) {                                     // 16 rows x 64 B per piece
                const int row = pc * 16 + (lane >> 2);
                const int ch = (lane & 3) ^ kc32_swz(row);
                voff = (unsigned)(((size_t)(r0 + row) * ld + k0 + ch * 8) * 2);
            } else {                                       // 4 k-rows x 256 B per piece, 8 pieces per sub-tile
                const int sub = pc >> 3;
                const int krow = (pc & 7) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ ks_swz(krow);
                voff = (unsigned)(((size_t)(k0 + krow) * ld + r0 + sub * 128 + ch * 8) * 2);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(lds_tile + pc * 1024), 16, voff, 0, 0, 0);
        }
    }
}

// Byte offset (into the operand's buffer) that lane `lane` fetches for 1 KB piece `pc` of a BK = 32 tile whose first
// row / k is (r0, k0): the per-lane part of dma_pieces32's address, computed ONCE per work item by the persistent kernel.
// Advancing K by one stage then adds a wave-uniform number of bytes (32 k * 2 B, or 32 k-rows * ld * 2 B).
template <bool KS>
__device__ __forceinline__ unsigned dma_piece_offset32(int ld, int r0, int k0, int pc, int lane) {
    if (!KS) {
        const int row = pc * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ kc32_swz(row);
        return (unsigned)(((size_t)(r0 + row) * ld + k0 + ch * 8) * 2);
    } else {
        const int sub = pc >> 3;
        const int krow = (pc & 7) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ ks_swz(krow);
        return (unsigned)(((size_t)(k0 + krow) * ld + r0 + sub * 128 + ch * 8) * 2);
    }
}

template <bool KS, typename H>
__device__ __forceinline__ typename H16<H>::x8 bf16_frag32(const char* lds, int rbase, int lane) {
    typedef typename H16<H>::x4 H4;
    typedef typename H16<H>::x8 H8;
    if (!KS) {
        const int row = rbase + (lane & 15);
        const int ch = lane >> 4;
        return *(const H8*)(lds + row * 64 + ((ch ^ kc32_swz(row)) << 4));
    } else {
        const int gl = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int kk = gl * 8 + q;
        const int rc = ((rbase & 127) >> 3) + (p >> 1);
        const char* base = lds + (rbase >> 7) * 8192 + kk * 256 + ((rc ^ ks_swz(kk)) << 4) + ((p & 1) << 3);
        const H4 lo = lds_read_tr16<H>(base);
        const H4 hi = lds_read_tr16<H>(base + 4 * 256);
        H8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// =====================================================================================
//     bf16 MFMA kernel, 128x128x32 tile, 2 stages (32 KB LDS) -> up to 4 blocks per CU
// =====================================================================================
// In-step timings show the main loops at 620-690 TF/s and the time going to epilogue-heavy
// GEMMs (fp32 residual read + write, dual outputs, GELU' aux reads): their HBM traffic is a
// serial phase per tile.  More co-resident workgroups overlap one tile's epilogue with other
// tiles' MFMAs; halving BK halves the LDS footprint and doubles the blocks per CU.
// WT = 16-row MFMA tiles per wave along M: 4 -> 128x128 block (4 blocks/CU), 2 -> 64x128 block
// (24 KB LDS, 6 blocks/CU).  The small tile is for narrow-N GEMMs whose 128x128 tiling fills the
// chip's workgroup slots only ~70 % in its last round (e.g. 70400x512: 2200 tiles on 1024 slots).
template <bool A_KS, bool B_KS, typename TC, int WT, typename H>
__global__ __launch_bounds__(256, WT == 4 ? 4 : 6) void gemm_bf16_k32_kernel(
    const H* __restrict__ A, const H* __restrict__ B, GemmDims d, EpiParams ep,
    float* __restrict__ slabs, unsigned a_bytes, unsigned b_bytes) {
    constexpr int BM = WT * 32;                    // 2 waves along M
    constexpr int A_BYTES = BM * 64;               // one BK=32 stage of A
    constexpr int STAGE = A_BYTES + 8192;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A | B 8 KB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = d.tiles_m * d.tiles_n;
    const int z = blockIdx.x / tiles;
    const int t = xcd_remap(blockIdx.x - z * tiles, tiles);
    int tm, tn;
    tile_coords(t, d.tiles_m, d.tiles_n, d.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * 128;
    const int kbeg = z * d.k_per_split;
    const int kend = min(d.k, kbeg + d.k_per_split);
    const int nk = (kend - kbeg + 31) >> 5;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);

    f32x4 acc[4][WT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage_tile = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        dma_pieces32<A_KS>(ra, st, d.lda, m0, kbeg + kt * 32, wave * (WT / 2), WT / 2, lane);
        dma_pieces32<B_KS>(rb, st + A_BYTES, d.ldb, n0, kbeg + kt * 32, wave * 2, 2, lane);
    };
    stage_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* la = smem + (kt & 1) * STAGE;
        const char* lb = la + A_BYTES;
        if (kt + 1 < nk) stage_tile(kt + 1);
        typename H16<H>::x8 af[WT], bfr[4];
#pragma unroll
        for (int i = 0; i < WT; ++i) af[i] = bf16_frag32<A_KS, H>(la, wm * (WT * 16) + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = bf16_frag32<B_KS, H>(lb, wn * 64 + j * 16, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < WT; ++i)
                acc[j][i] = mfma_16x16x32(bfr[j], af[i], acc[j][i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float* slab = slabs ? slabs + (size_t)z * ep.m * ep.n : nullptr;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int m = m0 + wm * (WT * 16) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m < ep.m && n < ep.n) {
                const float4 v = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
                if (slab) *(float4*)(slab + (size_t)m * ep.n + n) = v;
                else epilogue_store4<TC, H>(ep, m, n, v);
            }
        }
    }
}

// =====================================================================================
//   bf16 MFMA kernel, PERSISTENT 256x256 tile, 8 waves in two staggered groups ("p8")
// =====================================================================================
// The 128x128 kernels above top out where their structure does (every K step ends in a
// vmcnt(0) + barrier; overlap comes only from co-resident workgroups).  This kernel follows
// the other recipe: ONE workgroup per CU that keeps its LDS-DMA prefetch in flight across
// barriers.
//   * 8 waves = 2 (M) x 4 (N), wave tile 128x64 (128 accumulator registers); the two M-groups
//     sit pairwise on the four SIMDs and run one barrier apart, so while one group issues
//     its LDS reads / DMA pieces the other issues MFMAs (s_setprio raises the MFMA side).
//   * K advances in BK = 32 stages through a 4-slot LDS ring (4 x (A 16 KB | B 16 KB) =
//     128 KB); stage s+2 is issued while stage s is consumed; each wave waits for its own
//     pieces of stage s+1 with a COUNTED vmcnt one phase before anyone reads them, and the
//     barrier in between publishes them (a slot is rewritten 3 stages after its last read).
//   * two phases of 16 MFMAs per stage: phase a reads B (4) + A rows 0-63 (4) fragments and
//     issues the A pieces of stage s+2, phase b reads A rows 64-127 and issues the B pieces.
//   * persistent over (split, tile) work items: the next item's first two stages are issued
//     BEFORE the epilogue of the current one, so the epilogue's loads / stores overlap the
//     DMA flight, and the stores drain under the next item's main loop.
#define P8_STAGE 32768
#ifndef P8_ABL_PIECES      // timing ablation only: DMA pieces per wave and operand stage (2 = the real kernel)
#define P8_ABL_PIECES 2
#endif

#define P8_TAIL_MAX 128            // tail tiles a launch may split (arrival counters in its ticket set)
struct P8Item { int m0, n0, kbeg, nk, z, rev, p, part, slot; };   // part >= 0: K chunk `part` of tail tile `slot`

// A GROUP of deep dW = dY^T X problems that share K (the four weight gradients of one transformer layer) handed to ONE
// launch of the persistent kernel: their output tiles are numbered consecutively, an item is (split, tile) over the union,
// and a single K split of the union fills the CUs -- 48 tiles x 5 splits for the 8 x 512 fusion layer, where the four
// GEMMs on their own need 16 / 16 / 64 / 21 splits (and as many fp32 partial slabs each) to fill 256 CUs.
#define P8_MAX_GROUP 8
struct P8Group {
    const void* a[P8_MAX_GROUP];
    const void* b[P8_MAX_GROUP];
    float* slab[P8_MAX_GROUP];        // [split][m][n] fp32 partials of problem p
    int m[P8_MAX_GROUP], n[P8_MAX_GROUP], lda[P8_MAX_GROUP], ldb[P8_MAX_GROUP];
    int tiles_n[P8_MAX_GROUP];
    int tile_base[P8_MAX_GROUP + 1];  // first tile of problem p in the union; [count] = all tiles
    unsigned a_bytes[P8_MAX_GROUP], b_bytes[P8_MAX_GROUP];
    int count;
};

#ifdef P8_PROBE
__device__ unsigned long long p8_probe[16];
__device__ unsigned long long p8_probe_tail[64 * 8];      // per workgroup < 64 that ran a K chunk: see bf_debug_p8_probe_tail
__device__ unsigned long long p8_probe_sec[8];      // BK64 loop: load 0 / MFMA 0 / load 1 / MFMA 1 cycles of wave 0 and of wave 4
extern "C" int bf_debug_p8_probe(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(p8_probe), sizeof(p8_probe)) == hipSuccess ? 0 : 1;
}
extern "C" int bf_debug_p8_probe_tail(unsigned long long* out) {   // [64][8]: main loop, dump + drain, arrival, read-back, epilogue (cycles); part, parts, arrived
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(p8_probe_tail), sizeof(p8_probe_tail)) == hipSuccess ? 0 : 1;
}
extern "C" int bf_debug_p8_probe_sec(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(p8_probe_sec), sizeof(p8_probe_sec)) == hipSuccess ? 0 : 1;
}
#endif

// EPI: which epilogue this instantiation carries (the host evaluates the same predicates): 0 = general
// (any output type, residual / aux / copies / row remap / split-K slabs), 1 = plain bf16 (alpha, bias,
// activation, optional pre-activation copy), 2 = bf16 GELU-grad.  One kernel with all three paths sat at
// the 256-register limit and spilled differently with every edit.
#ifndef P9_PHASES
#define P9_PHASES 2                   // BK = 64 main loop: MFMA blocks per K tile (4 x 16 MFMAs, or 2 x 32 with half the barriers)
#endif
#ifndef P8_SOFF
#define P8_SOFF 1                     // persistent GEMM: K advance of K-contiguous operands in the DMA's scalar offset, fragment offsets hoisted (0: round 2's loop)
#endif
template <bool A_KS, bool B_KS, typename TC, int EPI, typename H, bool GROUPED = false, int BK = 32>
__device__ __forceinline__ void p8_body(
    const H* __restrict__ A, const H* __restrict__ B, const GemmDims& d, const EpiParams& ep,
    float* __restrict__ slabs, unsigned a_bytes, unsigned b_bytes, int n_items, int mode_bits,
    unsigned* __restrict__ tk, const P8Group* __restrict__ grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 x (A 16 KB | B 16 KB) + 16 B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles = GROUPED ? grp->tile_base[grp->count] : d.tiles_m * d.tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    // operand descriptors and leading dimensions: fixed for a plain launch, re-bound per work item in a grouped one
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    int lda = d.lda, ldb = d.ldb;
    auto bind = [&](const P8Item& it) {
        if constexpr (GROUPED) {
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)grp->a[it.p], 0, grp->a_bytes[it.p], 0x00020000);
            rb = __builtin_amdgcn_make_buffer_rsrc((void*)grp->b[it.p], 0, grp->b_bytes[it.p], 0x00020000);
            lda = grp->lda[it.p];
            ldb = grp->ldb[it.p];
        }
    };

    // Work distribution.  Items (split, tile) are cut into one contiguous range per XCD (blocks with
    // equal bid % 8 share an XCD and therefore an L2: neighbouring tiles, the same A / W panels).  Inside
    // a range the workgroups DRAW items from a ticket counter in global memory instead of owning a
    // fixed stride: a CU that is not available (held by a collective's resident kernel, or by the other
    // tower's kernels on the side stream) then simply draws nothing, instead of leaving its share
    // for a second pass -- with the fixed stride 8 busy CUs cost +44 % (tools/gemm_disturb.py).
    // tk[0..7]: tickets per XCD lane, tk[8]: workgroups finished (the last one zeroes the set for the
    // next launch on this stream).  The draw for the next item is issued by one thread a few stages before
    // the end of the main loop and published through LDS at the item boundary.
    const int NX = G < 8 ? G : 8;
    const int xl = bid % NX;
    const int per_xcd = max(1, G / NX);
    const int rev_rounds = mode_bits & 1;              // bit 0: alternate the K direction per round of tickets
    const bool deep = (mode_bits & 2) != 0;            // bit 1: prefetch distance 3 stages instead of 2 (see the main loop)
    // ranges proportional to the workgroups a lane has (G need not be a multiple of 8): with exactly one
    // item per workgroup no lane may hold more items than workgroups
    auto lane_share = [&](long long n, int x) { return (int)(n * (x * (G / NX) + min(x, G % NX)) / G); };
    // Tail split (BK = 64, round 4): the tiles of an under-filled last round -- the last d.tail_tiles in item order -- are
    // cut into d.tail_parts K chunks, so that round takes 1 / parts of a tile's time; the workgroup whose chunk arrives
    // last sums the chunks (in chunk order: reproducible) and runs the epilogue.  Every XCD lane gets its share of the
    // whole tiles first and then its share of the tail tiles, chunk index fastest.
    const bool tail_on = BK == 64 && !GROUPED && d.tail_tiles > 0;
    const int n_whole = tail_on ? n_items - d.tail_tiles * d.tail_parts : n_items;
    auto lane_start = [&](int x) { return lane_share(n_whole, x); };
    const int x_base = lane_start(xl);
    const int x_whole = lane_start(xl + 1) - x_base;
    const int t_base = tail_on ? lane_share(d.tail_tiles, xl) : 0;
    const int x_cnt = x_whole + (tail_on ? (lane_share(d.tail_tiles, xl + 1) - t_base) * d.tail_parts : 0);
    volatile unsigned* bcast = (volatile unsigned*)(smem + 4 * P8_STAGE);
    auto decode = [&](unsigned ticket, P8Item& w) -> bool {
        if (ticket >= (unsigned)x_cnt) return false;
        w.part = -1; w.slot = 0;
        if (tail_on && (int)ticket >= x_whole) {
            const int u = (int)ticket - x_whole;
            w.slot = t_base + u / d.tail_parts;
            w.part = u - (u / d.tail_parts) * d.tail_parts;
            int tm, tn;
            tile_coords(n_whole + w.slot, d.tiles_m, d.tiles_n, d.group_m, tm, tn);
            w.z = 0; w.p = 0;
            w.m0 = tm * 256; w.n0 = tn * 256;
            w.kbeg = w.part * d.tail_kt * 64;
            w.nk = min(d.tail_kt, ((d.k + 63) >> 6) - w.part * d.tail_kt);
            w.rev = 0;
            return true;
        }
        const int item = x_base + (int)ticket;
        w.z = item / tiles;
        int tm, tn;
        w.p = 0;
        if constexpr (GROUPED) {
            const int t = item - w.z * tiles;
            int p = 0;
            while (p + 1 < grp->count && t >= grp->tile_base[p + 1]) ++p;
            const int loc = t - grp->tile_base[p];
            w.p = p;
            tm = loc / grp->tiles_n[p];
            tn = loc - tm * grp->tiles_n[p];
        } else {
            tile_coords(item - w.z * tiles, d.tiles_m, d.tiles_n, d.group_m, tm, tn);
        }
        w.m0 = tm * 256; w.n0 = tn * 256;
        w.kbeg = w.z * d.k_per_split;
        w.nk = BK == 64 ? (min(d.k, w.kbeg + d.k_per_split) - w.kbeg + 63) >> 6
                        : (min(d.k, w.kbeg + d.k_per_split) - w.kbeg + 31) >> 5;
        // Every other ROUND of an XCD's tickets walks K backwards: the panels two consecutive rounds share (the W
        // panels of the next row group, the A panels of the next column tiles) are then first touched where the
        // previous round left them in the L2, instead of at the end that was evicted longest ago.  The direction is
        // a function of the ticket (not of which workgroup drew it), so a tile's summation order is reproducible.
        w.rev = rev_rounds ? (int)((ticket / (unsigned)per_xcd) & 1u) : 0;
        return true;
    };
    auto finish = [&]() {              // every workgroup exactly once, after its last draw
        if (threadIdx.x == 0 && atomicAdd(tk + 8, 1u) == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) tk[i] = 0u;
        }
    };
    auto issue_a_any = [&](const P8Item& w, int s) {       // any item: full address computation (item boundaries)
        const int ks = w.rev ? w.nk - 1 - s : s;
        dma_pieces32<A_KS>(ra, smem + (s & 3) * P8_STAGE, lda, w.m0, w.kbeg + ks * 32, wave * 2, P8_ABL_PIECES, lane);
    };
    auto issue_b_any = [&](const P8Item& w, int s) {
        const int ks = w.rev ? w.nk - 1 - s : s;
        dma_pieces32<B_KS>(rb, smem + (s & 3) * P8_STAGE + 16384, ldb, w.n0, w.kbeg + ks * 32, wave * 2, P8_ABL_PIECES, lane);
    };

    // In-loop DMA issue.  A piece's address used to be recomputed per piece and stage -- a 32-bit multiply (quarter
    // rate), two more VALU operations and SGPR-spill reloads in front of every buffer_load -- and VALU work of the
    // loading wave group comes out of the MFMA issue slots of its SIMD partner: the timing ablation without the DMA
    // issue ran 1 205 cycles per stage against 1 450 with it.  The per-lane part is now computed once per work item
    // (4 registers), a stage adds one wave-uniform term: one v_add per piece.  The K advance stays in the VGPR offset
    // (not the SGPR offset), so the descriptor's range check still zero-fills rows past the end of a matrix.
    unsigned va[2], vb[2];
    auto set_bases = [&](const P8Item& it) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            va[p] = dma_piece_offset32<A_KS>(lda, it.m0, it.kbeg, wave * 2 + p, lane);
            vb[p] = dma_piece_offset32<B_KS>(ldb, it.n0, it.kbeg, wave * 2 + p, lane);
        }
        asm volatile("" : "+v"(va[0]), "+v"(va[1]), "+v"(vb[0]), "+v"(vb[1]));   // live in registers, not rematerialised
    };
    // (round 3: for a K-CONTIGUOUS operand the K advance goes into the instruction's SCALAR offset -- no vector add per
    // piece.  The descriptor's range check sees the per-lane offset, which alone says whether the row exists, so rows past
    // the end of the matrix still read as zeros; a K-STRIDED operand keeps the vector add: its token tail is cut off by
    // the check on the sum.  With the fragment offsets below this takes the main loop from ~17 to ~5 vector instructions
    // per wave and stage; on this SIMD they come straight out of the MFMA time.)
    auto issue_a = [&](const P8Item& w, int s) {
        const int ks = w.rev ? w.nk - 1 - s : s;
        const unsigned delta = A_KS ? (unsigned)ks * 64u * (unsigned)lda : (unsigned)ks * 64u;
        char* dst = smem + (s & 3) * P8_STAGE + wave * 2048;
#pragma unroll
        for (int p = 0; p < P8_ABL_PIECES; ++p) {
            if (A_KS || !P8_SOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(dst + p * 1024), 16, va[p] + delta, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(dst + p * 1024), 16, va[p], delta, 0, 0);
        }
    };
    auto issue_b = [&](const P8Item& w, int s) {
        const int ks = w.rev ? w.nk - 1 - s : s;
        const unsigned delta = B_KS ? (unsigned)ks * 64u * (unsigned)ldb : (unsigned)ks * 64u;
        char* dst = smem + (s & 3) * P8_STAGE + 16384 + wave * 2048;
#pragma unroll
        for (int p = 0; p < P8_ABL_PIECES; ++p) {
            if (B_KS || !P8_SOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(dst + p * 1024), 16, vb[p] + delta, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(dst + p * 1024), 16, vb[p], delta, 0, 0);
        }
    };
    // lane-constant byte offsets of this wave's first A / B fragment inside a stage (K-contiguous operands: the other
    // fragments are 16 rows = 1 024 bytes apart, which the swizzle does not see): a stage's addresses are its scalar
    // slot base plus these
    unsigned fa_off = 0, fb_off = 0;
    if (!A_KS) { const int row = wm * 128 + (lane & 15); fa_off = (unsigned)(row * 64 + (((lane >> 4) ^ kc32_swz(row)) << 4)); }
    if (!B_KS) { const int row = wn * 64 + (lane & 15); fb_off = (unsigned)(16384 + row * 64 + (((lane >> 4) ^ kc32_swz(row)) << 4)); }
    asm volatile("" : "+v"(fa_off), "+v"(fb_off));

    // ---- BK = 64 main loop ("full lines", K-contiguous operands only; round 4) -----------------------------------
    // The BK = 32 stages above are 64-byte LDS rows, so a DMA piece fetches 16 rows x HALF a 128-byte line: twice the
    // requests per byte on the L2 path, which this loop loads to a third of its peak.  Here a K tile is 64 deep: a row
    // is one whole line (8 chunks of 16 B; chunk c of row r sits at position c ^ ((r >> 1) & 7): conflict-free for
    // ds_read_b128's 16-lane groups), a piece is 8 rows x 128 B, and the ring is TWO 64 KB tiles (A 32 KB | B 32 KB),
    // refilled by REGION as soon as a region's last reader is two phases past it:
    //   phase 0 reads A0 (rows 0-63 of the wave's 128) + B0 (cols 0-31 of its 64) and multiplies A0 x B0,
    //   phase 1 reads B1 -> A0 x B1,   phase 2 reads A1 -> A1 x B1,   phase 3 reads nothing -> A1 x B0 (B0 kept);
    //   phase 0 issues B1 of tile t+1, phase 1 A1 of t+1, phase 2 B0 of t+2, phase 3 A0 of t+2 (2 pieces per wave
    //   each): every region has 5-6 phases of flight time out of two buffers, and 8 pieces per wave are in flight.
    // A wave waits (counted vmcnt) for its own pieces of the region that is read in the NEXT phase; the two barriers
    // in between publish them to both wave groups (which run one barrier apart, as in the BK = 32 loop).
    unsigned va6[2][2] = {{0, 0}, {0, 0}}, vb6[2][2] = {{0, 0}, {0, 0}};   // per-lane byte offsets: [region half][piece]
    unsigned fa6[2] = {0, 0}, fb6[2] = {0, 0};                           // fragment offsets of k block 0 / 1 inside a tile
    auto a_rel6 = [&](int h, int p) { return (wave >> 2) * 128 + h * 64 + (wave & 3) * 16 + p * 8; };
    auto b_rel6 = [&](int h, int p) { return (wave >> 1) * 64 + h * 32 + (wave & 1) * 16 + p * 8; };
    // A piece is 8 rows x 128 B: lane l fetches row rel + (l >> 3), chunk (l & 7) ^ ((row >> 1) & 7).  rel is a multiple of
    // 8 whose bit 3 is the piece index p, so the chunk is (l & 7) ^ (l >> 4) ^ (p << 2): the per-lane part of an offset is
    // one of TWO values per operand, computed once per kernel (one 32-bit multiply each); an item adds a wave-uniform
    // term.  (Until this was split up, every item computed eight offsets with a 64-bit multiply each, and eight more in
    // front of the next item's first DMA: 1 750 cycles per item.)
    unsigned la6[2] = {0, 0}, lb6[2] = {0, 0};
    [[maybe_unused]] unsigned fbk6[2][2] = {{0, 0}, {0, 0}};             // K-strided B: fragment offsets of tile (g, j)
    if constexpr (BK == 64) {
        const unsigned ch0 = (unsigned)((lane & 7) ^ (lane >> 4));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            la6[p] = (unsigned)(lane >> 3) * (unsigned)lda * 2u + ((ch0 ^ (unsigned)(p << 2)) << 4);
            if constexpr (!B_KS) {
                lb6[p] = (unsigned)(lane >> 3) * (unsigned)ldb * 2u + ((ch0 ^ (unsigned)(p << 2)) << 4);
            } else {
                // K-strided B (dX = dY W, W stored [K][N]): a piece is 4 k-rows x 256 B of a 128-column sub-tile in the
                // k-row image of the BK = 32 loop (chunk c of k-row k at c ^ ks_swz(k)); the wave's pieces start at
                // k-rows ((2 wave) & 7) * 4 + 4 p of a 32-deep half, so ks_swz sees (lane >> 4) and wave & 1 only
                const unsigned ch = (unsigned)((lane & 15) ^ ((lane >> 4) << 1) ^ ((wave & 1) << 3));
                lb6[p] = (unsigned)(lane >> 4) * (unsigned)ldb * 2u + (ch << 4);
            }
        }
        asm volatile("" : "+v"(la6[0]), "+v"(la6[1]), "+v"(lb6[0]), "+v"(lb6[1]));
    }
    // wave-uniform byte offset of relative row `rel` of an item's tile (the row part stays in the VECTOR offset of the
    // DMA, where the descriptor's range check sees it: rows past the end of the matrix read as zeros)
    auto a_base6 = [&](const P8Item& it, int rel) { return ((unsigned)(it.m0 + rel) * (unsigned)lda + (unsigned)it.kbeg) * 2u; };
    auto b_base6 = [&](const P8Item& it, int rel) { return ((unsigned)(it.n0 + rel) * (unsigned)ldb + (unsigned)it.kbeg) * 2u; };
    // K-strided B: first k-row of the wave's piece p of k half h inside a tile, and the wave-uniform part of its offset
    auto bk_row6 = [&](int h, int p) { return h * 32 + ((2 * wave) & 7) * 4 + 4 * p; };
    auto bk_base6 = [&](const P8Item& it, int h, int p) {
        return ((unsigned)(it.kbeg + bk_row6(h, p)) * (unsigned)ldb + (unsigned)(it.n0 + (wave >> 2) * 128)) * 2u;
    };
    auto set_bases6 = [&](const P8Item& it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                va6[h][p] = la6[p] + a_base6(it, a_rel6(h, p));
                vb6[h][p] = lb6[p] + (B_KS ? bk_base6(it, h, p) : b_base6(it, b_rel6(h, p)));
            }
        asm volatile("" : "+v"(va6[0][0]), "+v"(va6[0][1]), "+v"(va6[1][0]), "+v"(va6[1][1]),
                          "+v"(vb6[0][0]), "+v"(vb6[0][1]), "+v"(vb6[1][0]), "+v"(vb6[1][1]));
    };
    // region (operand, half) of K tile tt of item `it` -> ring buffer tt & 1; `any`: addresses from scratch (item boundary)
    auto issue6 = [&](const P8Item& it, int tt, int is_b, int h, bool any) {
        const int ks = it.rev ? it.nk - 1 - tt : tt;
        if (B_KS && is_b) {
            // the K advance of a K-strided operand stays in the VECTOR offset (the descriptor's range check must see it)
            const unsigned delta = (unsigned)ks * 128u * (unsigned)ldb;
            char* dst = smem + (tt & 1) * 65536 + 32768 + (wave >> 2) * 16384 + bk_row6(h, 0) * 256;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned v = (any ? lb6[p] + bk_base6(it, h, p) : vb6[h][p]) + delta;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(dst + p * 1024), 16, v, 0, 0, 0);
            }
            return;
        }
        const unsigned delta = (unsigned)ks * 128u;
        char* dst = smem + (tt & 1) * 65536 + (is_b ? 32768 + b_rel6(h, 0) * 128 : a_rel6(h, 0) * 128);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned v;
            if (any) v = is_b ? lb6[p] + b_base6(it, b_rel6(h, p)) : la6[p] + a_base6(it, a_rel6(h, p));
            else v = is_b ? vb6[h][p] : va6[h][p];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(is_b ? rb : ra, (lds_void_ptr)(dst + p * 1024), 16, v, delta, 0, 0);
        }
    };
    auto issue_tile6_any = [&](const P8Item& it, int tt) {
        issue6(it, tt, 1, 0, true); issue6(it, tt, 0, 0, true); issue6(it, tt, 1, 1, true); issue6(it, tt, 0, 1, true);
    };
    if constexpr (BK == 64) {
        static_assert(BK != 64 || (!A_KS && !GROUPED), "the BK = 64 loop takes a K-contiguous A");
        const int sw = (lane >> 1) & 7;                    // (row >> 1) & 7 of the lane's fragment row (bases are multiples of 16)
        fa6[0] = (unsigned)(wm * 16384 + (lane & 15) * 128 + (((lane >> 4) ^ sw) << 4));
        fb6[0] = (unsigned)(32768 + wn * 8192 + (lane & 15) * 128 + (((lane >> 4) ^ sw) << 4));
        fa6[1] = fa6[0] ^ 64u;                             // chunk 4 + c: position bit 2 flipped
        fb6[1] = fb6[0] ^ 64u;
        asm volatile("" : "+v"(fa6[0]), "+v"(fa6[1]), "+v"(fb6[0]), "+v"(fb6[1]));
        if constexpr (B_KS) {
            // transposed reads (ds_read_b64_tr_b16) of the k-row image: lane -> k-row gl * 8 + q (+ 4 for the upper half of
            // the fragment), 8-byte piece p of the 16-column tile; tile (g, j) of the wave's 64 columns XORs its chunk
            // index (an even number) into bits 5-7, which the k-row term never touches
            const int gl = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
            const int kk0 = gl * 8 + q;
            const unsigned lanepart = (unsigned)(kk0 * 256 + ((((pp >> 1)) ^ ks_swz(kk0)) << 4) + ((pp & 1) << 3));
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned c0 = (unsigned)((wn & 1) * 8 + g * 4 + j * 2);
                    fbk6[g][j] = 32768u + (unsigned)(wn >> 1) * 16384u + (lanepart ^ (c0 << 4));
                }
            asm volatile("" : "+v"(fbk6[0][0]), "+v"(fbk6[0][1]), "+v"(fbk6[1][0]), "+v"(fbk6[1][1]));
        }
    }

    P8Item w;
    [[maybe_unused]] int round = 0;                     // items this workgroup has started (probe / bookkeeping only)
    unsigned pend = 0;                 // thread 0: the ticket drawn for the next item
    // (A start skew for the write-heavy epilogues -- workgroups of an XCD starting in phases so that a round's output
    // stores do not leave in one burst -- measured slower on every shape, 3.12 -> 3.15 - 3.27 ms: late starters cost more
    // than the spread-out bursts give back at 7 rounds per launch.  Removed in round 6.)
    if (threadIdx.x == 0) bcast[0] = atomicAdd(tk + xl, 1u);
    __syncthreads();
    const unsigned first = __builtin_amdgcn_readfirstlane(bcast[0]);   // wave-uniform: keeps the item in SGPRs
    unsigned nxt = 0;
    if (!decode(first, w)) { finish(); return; }
    bind(w);
    if constexpr (BK == 64) {
        issue_tile6_any(w, 0);
    } else {
        issue_a_any(w, 0); issue_b_any(w, 0);
        if (w.nk > 1) { issue_a_any(w, 1); issue_b_any(w, 1); }
    }

    while (true) {
        if constexpr (BK == 64) set_bases6(w); else set_bases(w);
        f32x4 acc[4][8];          // acc[j][i]: n tile j (16 cols), m tile i (16 rows); swapped roles
        // A plain residual (C = A B^T + bias + R, the transformer's skip connections) is loaded
        // INTO the accumulators here: 32 loads in flight at once, their latency under the wait for
        // the first stages, instead of 16 dependent load -> add -> store rounds in the epilogue
        // (measured: 55k cycles per tile for the epilogue of a residual GEMM).
        const bool res_init = EPI == 0 && ep.residual && !slabs && ep.act == BF_ACT_NONE && ep.alpha == 1.0f &&
                              ep.resid_mod == 0 && ep.rows_in == 0 && w.part < 0;     // (a K chunk adds it in its epilogue, once)
        if (res_init) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = w.m0 + wm * 128 + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = w.n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    if (m < ep.m && n < ep.n) acc[j][i] = *(const f32x4*)(ep.residual + (size_t)m * ep.ld_res + n);
                    else acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

#ifdef P8_PROBE
        const unsigned long long pcw = clock64();
#endif
        // (Round 5: a COUNTED wait here -- vmcnt(16): the first K tile's pieces were requested before the previous item's
        // sixteen epilogue stores and vmcnt retires in order -- and at tile 0's first in-loop wait, so that the stores stay
        // in flight until block 1 of the first K tile, was built and measured: 3.048 / 3.022 ms against 3.055 / 3.039 ms over
        // the twelve NT shapes, step 31.58 against 31.46 ms.  The 1 200 cycles at this barrier are not the wave's own wait
        // for its stores: every CU is in its epilogue at once, and the round of tiles ends when the write burst has left.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stages 0/1 (and the previous item's stores)
        __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
        if (bid == 0 && threadIdx.x == 0 && round == 2) p8_probe[6] = clock64() - pcw;
#endif
        // Prefetch distance 3 (`deep`): stage 2 goes out here -- after the barrier, because its slot was an epilogue
        // staging area of the previous item until every wave passed it -- and stage s + 3 in phase b of stage s: its
        // slot, (s - 1) & 3, was last read two barrier intervals earlier by the lagging group.  The ring stays 4 slots
        // (the LDS is full); what changes is that three stages instead of two are in flight, i.e. ~50 % more time for
        // a load that misses the L2 to arrive before a wave waits for it.
        if constexpr (BK != 64) {
            if (deep && w.nk > 2) { issue_a(w, 2); issue_b(w, 2); }
        }
        if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one slot behind
        const int nk = w.nk;
#ifdef P8_PROBE
        const unsigned long long pc0 = clock64(), pw0 = wall_clock64();
#endif
        // The next item is drawn LATE (8 stages before the end of this one: enough for the atomic to return
        // under the remaining MFMAs): a workgroup that drew two items up front would take work from
        // one that has not started yet, and with one item per CU (the split dW GEMMs) half the CUs idled.
        const int draw_stage = nk > 8 ? nk - 8 : 0;
        // (the address goes through an opaque per-lane zero: with a provably uniform address hipcc's atomic
        // optimiser rewrites the draw into a wave-aggregated atomic and waits vmcnt(0) for it on the spot,
        // which drained the DMA ring once per item)
        unsigned lane_zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
        if constexpr (BK == 64) {
            typedef typename H16<H>::x8 F8;
            const int draw_tile = nk > 4 ? nk - 4 : 0;
#if P9_PHASES == 2
            // Two MFMA blocks of 32 per K tile (half the barriers of the four-block loop below):
            //   block 0 reads A0 + B0 + B1 (16 fragments) -> A0 x B0, A0 x B1;  block 1 reads A1 (8) -> A1 x B1, A1 x B0.
            //   block 0 issues A1 of tile t+1 (its region was last read in block 1 of tile t-1), block 1 issues B0, B1,
            //   A0 of tile t+2 (last read in block 0 of this tile): every region is refilled one block after its last
            //   read, which is legal because a wave retires its reads (lgkmcnt(0)) BEFORE the barrier that ends its
            //   load section -- the other group, one barrier behind, has then retired the same block's reads too before
            //   anyone reaches the next load section.  8 pieces per wave stay in flight; a region has 2-3 blocks
            //   (2 600 - 3 900 cycles) between issue and first read.
            for (int t = 0; t < nk; ++t) {
                const char* tb = smem + (t & 1) * 65536;
                const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
                if (t == draw_tile && threadIdx.x == 0) pend = atomicAdd(tk + xl + lane_zero, 1u);
                if (t == 0 && n1) { issue6(w, 1, 1, 0, false); issue6(w, 1, 1, 1, false); issue6(w, 1, 0, 0, false); }
                F8 fa[2][4], fb0[2][2], fb1[2][2];
#ifdef P8_PROBE
                // section stamps (lgkmcnt is 0 at each of them: after a barrier that every wave reached with its reads
                // retired): load 0 | MFMA 0 | load 1 | MFMA 1, summed over the item's K tiles
                const unsigned long long q0 = clock64();
#endif
                // ---------------- block 0
                auto bfrag = [&](int g, int kb, int j) -> F8 {
                    if constexpr (!B_KS) {
                        return *(const F8*)(tb + fb6[kb] + g * 4096 + j * 2048);
                    } else {
                        typedef typename H16<H>::x4 F4;
                        const F4 lo = lds_read_tr16<H>(tb + fbk6[g][j] + kb * 8192);
                        const F4 hi = lds_read_tr16<H>(tb + fbk6[g][j] + kb * 8192 + 1024);
                        F8 r;
                        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
                        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
                        return r;
                    }
                };
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb0[kb][j] = bfrag(0, kb, j);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kb][i] = *(const F8*)(tb + fa6[kb] + i * 2048);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb1[kb][j] = bfrag(1, kb, j);
                if (n1) {
                    issue6(w, t + 1, 0, 1, false);                       // A1 of tile t + 1
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // A1 of this tile landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
                const unsigned long long q1 = clock64();
#endif
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = mfma_16x16x32(fb0[kb][j], fa[kb][i], acc[j][i]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2 + j][i] = mfma_16x16x32(fb1[kb][j], fa[kb][i], acc[2 + j][i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
                const unsigned long long q2 = clock64();
#endif
                // ---------------- block 1
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kb][i] = *(const F8*)(tb + fa6[kb] + 8192 + i * 2048);
                if (n2) {
                    issue6(w, t + 2, 1, 0, false); issue6(w, t + 2, 1, 1, false); issue6(w, t + 2, 0, 0, false);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // B0, B1, A0 of tile t + 1 landed
                } else if (n1) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
                const unsigned long long q3 = clock64();
#endif
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2 + j][4 + i] = mfma_16x16x32(fb1[kb][j], fa[kb][i], acc[2 + j][4 + i]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][4 + i] = mfma_16x16x32(fb0[kb][j], fa[kb][i], acc[j][4 + i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
                if (bid == 0 && (threadIdx.x & 255) == 0 && round == 1) {      // wave 0 (group 0) and wave 4 (group 1)
                    const int o = 16 + (threadIdx.x >> 8) * 4;
                    const unsigned long long q4 = clock64();
                    if (t == 0) { p8_probe_sec[o - 16] = 0; p8_probe_sec[o - 15] = 0; p8_probe_sec[o - 14] = 0; p8_probe_sec[o - 13] = 0; }
                    p8_probe_sec[o - 16] += q1 - q0; p8_probe_sec[o - 15] += q2 - q1;
                    p8_probe_sec[o - 14] += q3 - q2; p8_probe_sec[o - 13] += q4 - q3;
                }
#endif
            }
#else
            static_assert(!B_KS, "the four-block variant of the BK = 64 loop was kept for K-contiguous operands only");
            for (int t = 0; t < nk; ++t) {
                const char* tb = smem + (t & 1) * 65536;
                const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
                if (t == draw_tile && threadIdx.x == 0) pend = atomicAdd(tk + xl + lane_zero, 1u);
                // B0, A0 of tile 1: what phases 2 / 3 of "tile -1" would have issued.  (Inside the loop: in front of it
                // hipcc flushes vmcnt(0) in the loop preheader -- it cannot see that the asm wait above covered the
                // residual loads into the accumulators -- and the four pieces would be waited for on the spot.)
                if (t == 0 && n1) { issue6(w, 1, 1, 0, false); issue6(w, 1, 0, 0, false); }
                F8 fa[2][4], fb0[2][2], fb1[2][2];     // [k block][16-row tile]
                // ---------------- phase 0: A0 x B0
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb0[kb][j] = *(const F8*)(tb + fb6[kb] + j * 2048);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kb][i] = *(const F8*)(tb + fa6[kb] + i * 2048);
                if (n1) {
                    issue6(w, t + 1, 1, 1, false);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // B1 of this tile landed
                } else {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = mfma_16x16x32(fb0[kb][j], fa[kb][i], acc[j][i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- phase 1: A0 x B1
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb1[kb][j] = *(const F8*)(tb + fb6[kb] + 4096 + j * 2048);
                if (n1) {
                    issue6(w, t + 1, 0, 1, false);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A1 of this tile landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2 + j][i] = mfma_16x16x32(fb1[kb][j], fa[kb][i], acc[2 + j][i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- phase 2: A1 x B1
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kb][i] = *(const F8*)(tb + fa6[kb] + 8192 + i * 2048);
                if (n2) issue6(w, t + 2, 1, 0, false);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[2 + j][4 + i] = mfma_16x16x32(fb1[kb][j], fa[kb][i], acc[2 + j][4 + i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- phase 3: A1 x B0 (no reads)
                if (n2) {
                    issue6(w, t + 2, 0, 0, false);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A0, B0 of tile t + 1 landed
                } else if (n1) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][4 + i] = mfma_16x16x32(fb0[kb][j], fa[kb][i], acc[j][4 + i]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            }
#endif
        } else
        for (int s = 0; s < nk; ++s) {
            const char* la = smem + (s & 3) * P8_STAGE;
            const char* lb = la + 16384;
            const bool more = s + 2 < nk;
            if (s == draw_stage && threadIdx.x == 0) pend = atomicAdd(tk + xl + lane_zero, 1u);
            // ---------------- phase a
            typename H16<H>::x8 bfr[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[j] = (B_KS || !P8_SOFF) ? bf16_frag32<B_KS, H>(lb, wn * 64 + j * 16, lane)
                                            : *(const typename H16<H>::x8*)(la + fb_off + j * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = (A_KS || !P8_SOFF) ? bf16_frag32<A_KS, H>(la, wm * 128 + i * 16, lane)
                                           : *(const typename H16<H>::x8*)(la + fa_off + i * 1024);
            if (more && !deep) issue_a(w, s + 2);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = mfma_16x16x32(bfr[j], af[i], acc[j][i]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase b
            typename H16<H>::x8 ag[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ag[i] = (A_KS || !P8_SOFF) ? bf16_frag32<A_KS, H>(la, wm * 128 + 64 + i * 16, lane)
                                           : *(const typename H16<H>::x8*)(la + fa_off + 4096 + i * 1024);
            if (!deep) {
                if (more) {
                    issue_b(w, s + 2);
#if P8_ABL_PIECES == 2
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage s+1 landed (s+2 in flight)
#elif P8_ABL_PIECES == 1
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else if (s + 3 < nk) {
                issue_a(w, s + 3); issue_b(w, s + 3);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stage s+1 landed (s+2, s+3 in flight)
            } else if (more) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // s+2 (the last stage) may still be in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][4 + i] = mfma_16x16x32(bfr[j], ag[i], acc[j][4 + i]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // realign the groups
#ifdef P8_PROBE
        const unsigned long long pc1 = clock64(), pw1 = wall_clock64();
#endif

        // (the epilogue's per-lane values are derived from an OPAQUE copy of the lane id made here, so that the compiler
        // cannot hoist them above the main loop, where every register is spoken for: round 4, BK = 64 loop)
        int lane_ep = lane;
        asm volatile("" : "+v"(lane_ep));
        // The plain 16-bit epilogue's bias values are requested HERE, in front of the next item's DMA pieces: vmcnt
        // retires in order, so loaded after them the first use of the bias waited for the whole first K tile of the
        // next item to arrive (~900 cycles per item in the probe).
        [[maybe_unused]] float4 bj[4];
        if constexpr (EPI == 1 || EPI == 3 || EPI == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = w.n0 + wn * 64 + j * 16 + (lane_ep >> 4) * 4;
                bj[j] = (ep.bias && n < ep.n) ? *(const float4*)(ep.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // EPI 5 = EPI 1 + a low-rank addend (bf_gemm_args::lr_u / lr_w: the LoRA branch of the projection): one more
        // K-step of MFMAs whose operand fragments come straight from global memory in the fragment layout (row = lane % 16,
        // eight k values per 16-lane group; groups past lr_k supply zeros).  Issued here, in front of the next item's DMA
        // pieces like the bias, so that their wait is not queued behind a whole K tile.  A tail tile split along K adds
        // it in its first chunk only.
        if constexpr (EPI == 5) {
            if (w.part <= 0) {
                typedef typename H16<H>::x8 H8;
                const H* lu = (const H*)ep.lr_u;
                const H* lw = (const H*)ep.lr_w;
                const int kq = (lane_ep >> 4) * 8;
                const bool kin = kq < ep.lr_k;
                H8 uf[8], wf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = w.n0 + wn * 64 + j * 16 + (lane_ep & 15);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (kin && n < ep.n) v = *(const uint4*)(lw + (size_t)n * ep.lr_k + kq);
                    wf[j] = __builtin_bit_cast(H8, v);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = w.m0 + wm * 128 + i * 16 + (lane_ep & 15);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (kin && m < ep.m) v = *(const uint4*)(lu + (size_t)m * ep.lr_k + kq);
                    uf[i] = __builtin_bit_cast(H8, v);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[j][i] = mfma_16x16x32(wf[j], uf[i], acc[j][i]);
            }
        }
        // next item's first stages go out before this item's epilogue
        P8Item nx;
        ++round;
        if (threadIdx.x == 0) bcast[0] = pend;             // publish the draw made during the main loop
        __builtin_amdgcn_s_barrier();
        nxt = __builtin_amdgcn_readfirstlane(bcast[0]);
        const bool has_next = decode(nxt, nx);
        if (has_next) {
            bind(nx);
            if constexpr (BK == 64) {
                issue_tile6_any(nx, 0);
            } else {
                issue_a_any(nx, 0); issue_b_any(nx, 0);
                if (nx.nk > 1) { issue_a_any(nx, 1); issue_b_any(nx, 1); }
            }
        }
#ifdef P8_PROBE
        const unsigned long long pc2 = clock64();
#endif

        // ---- K chunk of a tail tile: hand the accumulators over; the LAST chunk to arrive sums them and stores the tile
        bool run_epilogue = true;
        if constexpr (BK == 64 && !GROUPED) {
            if (w.part >= 0) {
                // The hand-off is the guide's counter form with write-through payload (MI355X_MICROARCH.md, "Valid
                // forms"): every 16-byte store of the accumulators is `sc1`, every storing wave drains its stores
                // (vmcnt(0)), a workgroup barrier, then ONE lane's agent-scope add on the tile's arrival counter; the
                // workgroup whose add returns parts - 1 came last -- it learns so from the returned value -- and reads
                // every chunk (its own too: a fixed summation order) with `sc1` loads behind a workgroup barrier that
                // the adding wave joins after its add has returned.  No fence, no poll.  The image is the register
                // file as it stands ([wave][register][lane] x 16 B): every store instruction writes 1 KB contiguous.
                const int S = d.tail_parts;
#ifdef P8_PROBE
                const unsigned long long pt0 = clock64();
#endif
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)d.tail_ws, 0, (unsigned)d.tail_tiles * (unsigned)S * 262144u, 0x00020000);
                const unsigned own = ((unsigned)(w.slot * S + w.part) * 8u + (unsigned)wave) * 32768u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[j][i]), rw,
                                                                  (unsigned)lane * 16u, own + (unsigned)(j * 8 + i) * 1024u, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
                const unsigned long long pt1 = clock64();
#endif
                if (threadIdx.x == 0)
                    bcast[1] = __hip_atomic_fetch_add(tk + 16 + w.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_barrier();
                const unsigned arrived = __builtin_amdgcn_readfirstlane(bcast[1]);
                run_epilogue = arrived == (unsigned)S - 1u;
#ifdef P8_PROBE
                if (bid < 64 && threadIdx.x == 0) {
                    unsigned long long* pt = p8_probe_tail + bid * 8;
                    pt[0] = pc1 - pc0; pt[1] = pt1 - pt0; pt[2] = clock64() - pt1; pt[3] = 0; pt[4] = 0;
                    pt[5] = (unsigned long long)w.part; pt[6] = (unsigned long long)S; pt[7] = arrived;
                }
                const unsigned long long pt2 = clock64();
#endif
                if (run_epilogue) {
                    if (threadIdx.x == 0) tk[16 + w.slot] = 0u;          // at rest again for this set's next launch
                    // Read-back, 32 KB per wave and chunk.  This workgroup's own accumulators are in the workspace like
                    // everyone's, so the registers are free to receive: rows i < 4 first, chunk 0 straight into them and chunk
                    // 1 into the registers of rows i >= 4 in the same breath (32 loads per lane in flight), then rows i >= 4
                    // with chunk 0 straight in and the other chunks through eight spare registers.  Eight loads at a time
                    // for everything made the read-back 4 S dependent trips to memory -- as long as the K tiles the split
                    // had saved.  Every element is still summed in chunk order (same bits).
                    auto part_ld = [&](int z, int r) -> f32x4 {
                        const unsigned src = ((unsigned)(w.slot * S + z) * 8u + (unsigned)wave) * 32768u;
                        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             rw, (unsigned)lane * 16u, src + (unsigned)r * 1024u, 16));
                    };
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = part_ld(0, j * 8 + i);
#pragma unroll 1
                    for (int z = 1; z < S; ++z) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[j][4 + i] = part_ld(z, j * 8 + i);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[j][i] += acc[j][4 + i];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][4 + i] = part_ld(0, j * 8 + 4 + i);
#pragma unroll 1
                    for (int z = 1; z < S; ++z) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            f32x4 t8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) t8[q] = part_ld(z, (jj * 2 + (q >> 2)) * 8 + 4 + (q & 3));
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[jj * 2 + (q >> 2)][4 + (q & 3)] += t8[q];
                        }
                    }
#ifdef P8_PROBE
                    if (bid < 64 && threadIdx.x == 0) p8_probe_tail[bid * 8 + 3] = clock64() - pt2;
#endif
                }
            }
        }
        if (run_epilogue) {
        // Epilogue through LDS: in the MFMA layout a wave-instruction's 64 lanes own 16 rows x 4
        // separate 16-byte groups, i.e. 64 scattered requests per store (measured: ~780 cycles per
        // store instruction and wave, the epilogue as long as a K = 768 main loop).  Each wave
        // turns its 128x64 tile around in 32-row chunks through a private 8 KB of ring slots 2-3
        // (slots 0-1 are receiving the next item), so that 16 adjacent lanes cover 64 consecutive
        // columns of one row: 4 whole 128/256-byte row segments per store instruction.
        // output extent of this item: the launch's (plain) or its problem's (grouped: fp32 partial slabs only)
        const int epm = GROUPED ? grp->m[w.p] : ep.m, epn = GROUPED ? grp->n[w.p] : ep.n;
        float* slab = GROUPED ? grp->slab[w.p] + (size_t)w.z * epm * epn
                              : (slabs ? slabs + (size_t)w.z * ep.m * ep.n : nullptr);
        char* stg = smem + 2 * P8_STAGE + wave * 8192;
        EpiParams epi = ep;
        if (res_init) epi.residual = nullptr;              // already in the accumulators
        // ONE copy of the epilogue code: fully unrolled (32 copies of every activation / residual
        // variant, ~100 KB of instructions) it thrashed the 64 KB instruction cache and cost
        // ~780 cycles per store -- as much as a whole K = 768 main loop per tile.
        // Plain bf16 outputs (alpha / bias / activation only): do the math in the MFMA layout, turn
        // the tile around as bf16 (half the LDS bytes: ds_write runs at ~80 B/clk per CU) in two
        // 64-row halves, and store 8 columns per lane_ep.
        constexpr bool fast16 = EPI == 1 || EPI == 3 || EPI == 5;
        if constexpr (fast16) {
            using AM = ActMath<H>;
            // staging offsets of the lane's four 8-byte pieces of a row (one per 16-column tile j), hoisted: the XOR
            // swizzle does not see the 16-row block (ii), which rides in the instruction's immediate offset
            unsigned wo4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr0 = lane_ep & 15, cc4 = j * 4 + (lane_ep >> 4);
                wo4[j] = (unsigned)(lr0 * 128 + (((cc4 >> 1) ^ ((lr0 >> 1) & 7)) << 4) + ((cc4 & 1) << 3));
            }
            // pass 0 writes the pre-activation copy the backward pass needs (ep.pre), pass 1 the result
            auto half_tile = [&](auto h_, auto pass_) {
                constexpr int HALF = decltype(h_)::value;
                constexpr int PASS = decltype(pass_)::value;
                H* dst = PASS == 0 ? (H*)ep.pre : (H*)ep.c;
#ifdef P8_PROBE
                const unsigned long long ph0 = clock64();
#endif
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 a = acc[j][HALF * 4 + ii];
                        float v[4] = {fmaf(a[0], ep.alpha, bj[j].x), fmaf(a[1], ep.alpha, bj[j].y),
                                      fmaf(a[2], ep.alpha, bj[j].z), fmaf(a[3], ep.alpha, bj[j].w)};
                        if (PASS == 1) {
                            if (ep.act == BF_ACT_QUICK_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = AM::quick(v[e]);
                            } else if (ep.act == BF_ACT_TANH_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = AM::tanh_gelu(v[e]);
                            } else if (ep.act == BF_ACT_ERF_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = AM::erf(v[e]);
                            }
                        }
                        typename H16<H>::x4 pk;
                        pk[0] = (H)v[0]; pk[1] = (H)v[1]; pk[2] = (H)v[2]; pk[3] = (H)v[3];
                        *(typename H16<H>::x4*)(stg + wo4[j] + ii * 2048) = pk;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#ifdef P8_PROBE
                const unsigned long long ph1 = clock64();
#endif
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int lr = t * 8 + (lane_ep >> 3), c16 = lane_ep & 7;
                    const uint4 q = *(const uint4*)(stg + lr * 128 + ((c16 ^ ((lr >> 1) & 7)) << 4));
                    const int m = w.m0 + wm * 128 + HALF * 64 + lr;
                    const int n = w.n0 + wn * 64 + c16 * 8;
                    if (m < ep.m && n < ep.n) store16(dst + (size_t)m * ep.ldc + n, q, ep.store_policy & 1);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#ifdef P8_PROBE
                if (bid == 0 && threadIdx.x == 0 && round == 1) {     // wave 0: conversion + LDS writes | LDS reads + store issue
                    p8_probe[8 + HALF * 2] = ph1 - ph0;
                    p8_probe[9 + HALF * 2] = clock64() - ph1;
                }
#endif
            };
            // (Round 4 also built these stores WITHOUT the trip through LDS -- v_permlane16_swap pairs, 16 rows x 64 bytes
            // per store instruction -- and measured 3.17 - 3.20 ms against 3.09 through LDS over the twelve NT shapes: a store
            // of 16 rows x 64 B costs more than the LDS turn that makes it 8 rows x 128 B saves.  Removed in round 6.)
            const bool want_pre = ep.pre && (ep.act == BF_ACT_QUICK_GELU || ep.act == BF_ACT_ERF_GELU ||
                                             ep.act == BF_ACT_TANH_GELU);
            if constexpr (EPI == 3) {
                // BF_ACT_ERF_GELU_SAVEGRAD: gelu and gelu' of a QUARTER tile (32 rows) from ONE erf evaluation per
                // element; the two results go to the two halves of the wave's 8 KB staging buffer at once (no
                // registers held across the turn-around: the half-tile version spilled), then to `c` and `pre`.
                auto quarter_pair = [&](auto q_) {
                    constexpr int Q = decltype(q_)::value;          // rows [32 Q, 32 Q + 32) of the wave's 128
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 a = acc[j][Q * 2 + i2];
                            const float v[4] = {fmaf(a[0], ep.alpha, bj[j].x), fmaf(a[1], ep.alpha, bj[j].y),
                                                fmaf(a[2], ep.alpha, bj[j].z), fmaf(a[3], ep.alpha, bj[j].w)};
                            typename H16<H>::x4 pk, gk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float gv, gp;
                                AM::erf_both(v[e], gv, gp);
                                pk[e] = (H)gv;
                                gk[e] = (H)gp;
                            }
                            *(typename H16<H>::x4*)(stg + wo4[j] + i2 * 2048) = pk;
                            *(typename H16<H>::x4*)(stg + 4096 + wo4[j] + i2 * 2048) = gk;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int lr = t * 8 + (lane_ep >> 3), c16 = lane_ep & 7;
                        const int off = lr * 128 + ((c16 ^ ((lr >> 1) & 7)) << 4);
                        const uint4 qa = *(const uint4*)(stg + off);
                        const uint4 qg = *(const uint4*)(stg + 4096 + off);
                        const int m = w.m0 + wm * 128 + Q * 32 + lr;
                        const int n = w.n0 + wn * 64 + c16 * 8;
                        if (m < ep.m && n < ep.n) {
                            store16((H*)ep.c + (size_t)m * ep.ldc + n, qa, ep.store_policy & 1);
                            store16((H*)ep.pre + (size_t)m * ep.ldc + n, qg, ep.store_policy & 1);
                        }
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                };
                quarter_pair(std::integral_constant<int, 0>{});
                quarter_pair(std::integral_constant<int, 1>{});
                quarter_pair(std::integral_constant<int, 2>{});
                quarter_pair(std::integral_constant<int, 3>{});
            } else {
            if (want_pre) {
                half_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                half_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            }
            half_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            half_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            }
        } else {
        auto stage_chunk = [&](auto cc_) {
            constexpr int C = decltype(cc_)::value;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int lr = ii * 16 + (lane_ep & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = j * 4 + (lane_ep >> 4);
                    *(f32x4*)(stg + lr * 256 + ((cc ^ (lr & 15)) << 4)) = acc[j][2 * C + ii];
                }
            }
        };
        // dX = (dY W) * gelu'(pre) with a bf16 result (the MLP's input gradient): all 16 rows of
        // `pre` this lane_ep needs are requested at once (64 registers of raw bf16), so the tile pays
        // one load latency instead of sixteen; lean body, fully unrolled.
        constexpr bool fastgrad = EPI == 2;
        if constexpr (fastgrad) {
            using AM = ActMath<H>;
            uint4 apre[16];
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const int m = w.m0 + wm * 128 + ct * 8 + (lane_ep >> 3);
                const int n = w.n0 + wn * 64 + (lane_ep & 7) * 8;
                apre[ct] = (m < ep.m && n < ep.n) ? *(const uint4*)((const H*)ep.aux + (size_t)m * ep.ld_aux + n)
                                                  : make_uint4(0, 0, 0, 0);
            }
            float cs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] = 0.f;
            auto grad_chunk = [&](auto cc_) {
                constexpr int C = decltype(cc_)::value;
                stage_chunk(cc_);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int lr = t * 8 + (lane_ep >> 3), c2 = (lane_ep & 7) * 2;
                    const f32x4 q0 = *(const f32x4*)(stg + lr * 256 + ((c2 ^ (lr & 15)) << 4));
                    const f32x4 q1 = *(const f32x4*)(stg + lr * 256 + (((c2 + 1) ^ (lr & 15)) << 4));
                    const int m = w.m0 + wm * 128 + C * 32 + lr;
                    const int n = w.n0 + wn * 64 + c2 * 4;
                    const uint4 raw = apre[C * 4 + t];
                    const typename H16<H>::x8 a8 = *(const typename H16<H>::x8*)&raw;
                    typename H16<H>::x8 o8;
                    float p8[8];                           // the products in fp32: what the bias-gradient sums add
                    if (ep.act == BF_ACT_MUL_AUX) {        // aux already IS gelu'(pre) (forward: ..._SAVEGRAD)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            p8[e] = q0[e] * ep.alpha * (float)a8[e];
                            p8[4 + e] = q1[e] * ep.alpha * (float)a8[4 + e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            p8[e] = q0[e] * ep.alpha * AM::erf_grad((float)a8[e]);
                            p8[4 + e] = q1[e] * ep.alpha * AM::erf_grad((float)a8[4 + e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (H)p8[e];
                    if (m < ep.m && n < ep.n) {
                        store16((H*)ep.c + (size_t)m * ep.ldc + n, *(const uint4*)&o8, ep.store_policy & 4);
                        // Column sums of the fp32 products (round 5; until then of the STORED 16-bit values, converted
                        // back: eight conversions per row segment, half of what the sums cost this epilogue).  The bias
                        // gradient is the sum of the unrounded terms -- what an fp32 reduction of the GEMM's result gives.
#pragma unroll
                        for (int e = 0; e < 8; ++e) cs[e] += p8[e];
                    }
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            };
            grad_chunk(std::integral_constant<int, 0>{});
            grad_chunk(std::integral_constant<int, 1>{});
            grad_chunk(std::integral_constant<int, 2>{});
            grad_chunk(std::integral_constant<int, 3>{});
            if (ep.colsum) {        // 128 rows of this wave -> one atomic per column
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    cs[e] += __shfl_xor(cs[e], 8, 64);
                    cs[e] += __shfl_xor(cs[e], 16, 64);
                    cs[e] += __shfl_xor(cs[e], 32, 64);
                }
                const int n = w.n0 + wn * 64 + (lane_ep & 7) * 8;
                if (lane_ep < 8 && n < ep.n) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(ep.colsum + n + e, cs[e]);
                }
            }
        } else {
        if constexpr (std::is_same<TC, float>::value) {
        // fp32 results (and fp32 split-K slabs): 16 bytes per lane are FOUR columns, so 16 lanes cover a row's 64 columns
        // and a store instruction is 4 rows x 256 contiguous bytes.  (Until round 4 a lane took 8 columns as two
        // 16-byte stores: every store instruction wrote the even or the odd 16-byte pieces of 8 rows -- half of every
        // line it touched, twice the write requests per byte.)
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            const int n = w.n0 + wn * 64 + (lane_ep & 15) * 4;
            if (ep.bias && !slab && n < epn) b4 = *(const float4*)(ep.bias + n);
        }
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            switch (c) {
                case 0: stage_chunk(std::integral_constant<int, 0>{}); break;
                case 1: stage_chunk(std::integral_constant<int, 1>{}); break;
                case 2: stage_chunk(std::integral_constant<int, 2>{}); break;
                default: stage_chunk(std::integral_constant<int, 3>{}); break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int lr = t * 4 + (lane_ep >> 4), cc = lane_ep & 15;
                const f32x4 q = *(const f32x4*)(stg + lr * 256 + ((cc ^ (lr & 15)) << 4));
                const int m = w.m0 + wm * 128 + c * 32 + lr;
                const int n = w.n0 + wn * 64 + cc * 4;
                if (m < epm && n < epn) {
                    if (slab) *(f32x4*)(slab + (size_t)m * epn + n) = q;
                    else epilogue_store4b<TC, H>(epi, m, n, make_float4(q[0], q[1], q[2], q[3]), b4);
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        } else {
        float b8[8];                   // the lane_ep's 8 columns are the same for every row of the tile
        {
            const int n = w.n0 + wn * 64 + (lane_ep & 7) * 8;
            if (ep.bias && !slab && n < epn) Vec8<float>::load(ep.bias + n, b8);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) b8[e] = 0.f;
            }
        }
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            switch (c) {
                case 0: stage_chunk(std::integral_constant<int, 0>{}); break;
                case 1: stage_chunk(std::integral_constant<int, 1>{}); break;
                case 2: stage_chunk(std::integral_constant<int, 2>{}); break;
                default: stage_chunk(std::integral_constant<int, 3>{}); break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int lr = t * 8 + (lane_ep >> 3), c2 = (lane_ep & 7) * 2;
                const f32x4 q0 = *(const f32x4*)(stg + lr * 256 + ((c2 ^ (lr & 15)) << 4));
                const f32x4 q1 = *(const f32x4*)(stg + lr * 256 + (((c2 + 1) ^ (lr & 15)) << 4));
                const int m = w.m0 + wm * 128 + c * 32 + lr;
                const int n = w.n0 + wn * 64 + c2 * 4;
                if (m < epm && n < epn) {
                    if (slab) {
                        float* sp = slab + (size_t)m * epn + n;
                        *(f32x4*)sp = q0;
                        *(f32x4*)(sp + 4) = q1;
                    } else {
                        float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                        epilogue_store8<TC, H>(epi, m, n, v, b8);
                    }
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        }   // 16-bit general outputs
        }
        }   // general epilogue
        }   // run_epilogue
#ifdef P8_PROBE
        if (bid == 0 && threadIdx.x == 0 && round == 1) {
            p8_probe[0] = pc1 - pc0; p8_probe[1] = pw1 - pw0; p8_probe[2] = nk;
            p8_probe[3] = clock64() - pc1; p8_probe[4] = wall_clock64() - pw1; p8_probe[5] = pc2 - pc1;
        }
        if constexpr (BK == 64 && !GROUPED)
            if (w.part >= 0 && bid < 64 && threadIdx.x == 0) p8_probe_tail[bid * 8 + 4] = clock64() - pc1;   // main loop end -> item end
#endif
        if (!has_next) break;
        w = nx;
    }
    finish();
}

template <bool A_KS, bool B_KS, typename TC, int EPI, typename H, int BK = 32>
__global__ __launch_bounds__(512, 1) void gemm_bf16_p8_kernel(
    const H* __restrict__ A, const H* __restrict__ B, GemmDims d, EpiParams ep,
    float* __restrict__ slabs, unsigned a_bytes, unsigned b_bytes, int n_items, int mode_bits,
    unsigned* __restrict__ tk) {
    p8_body<A_KS, B_KS, TC, EPI, H, false, BK>(A, B, d, ep, slabs, a_bytes, b_bytes, n_items, mode_bits, tk, nullptr);
}

// The grouped dW launch: TN layout, fp32 partial slabs (general epilogue), every problem's geometry in `grp`.
template <typename H>
__global__ __launch_bounds__(512, 1) void gemm_p8_grouped_tn_kernel(P8Group grp, GemmDims d, EpiParams ep, int n_items,
                                                                    int mode_bits, unsigned* __restrict__ tk) {
    p8_body<true, true, float, 0, H, true>((const H*)grp.a[0], (const H*)grp.b[0], d, ep, grp.slab[0], grp.a_bytes[0],
                                           grp.b_bytes[0], n_items, mode_bits, tk, &grp);
}

// =====================================================================================
//   e4m3 sibling of the persistent kernel (frozen towers, BASELINE configs[4]): "p8f"
// =====================================================================================
// Same workgroup geometry, ticket draw, LDS byte images and staggered wave groups as gemm_bf16_p8_kernel, with one-byte
// operands: a 32 KB stage (256 rows x 64 B of A | B) holds K = 64, and v_mfma_scale_f32_16x16x128_f8f6f4 (twice the
// bf16 rate, block scales fixed at 2^0) consumes K = 128, i.e. a PAIR of stages: lanes 0-31 of a fragment read the even
// stage's slot, lanes 32-63 the odd stage's.  The four ring slots therefore hold two stage pairs: pair d is consumed
// from slots {2d, 2d+1} & 3 while pair d+1 lands in the other two.
//   load block a : B fragments (4) + A fragments of rows 0-63 (4) of pair d, then ALL 8 DMA pieces of pair d+1
//                  (their slots were last read in load block b of pair d-1, whose reads are retired -- lgkmcnt(0) --
//                  before that block's barrier: the one-phase restaging rule);
//   MFMA block a : 16 MFMAs of K = 128 (32 cycles each);
//   load block b : A fragments of rows 64-127, lgkmcnt(0), then vmcnt(0): every wave's pieces of pair d+1 have landed
//                  before the barrier both wave groups pass in front of pair d+1's first read;
//   MFMA block b : 16 MFMAs.
// Per K the loop moves half the bytes of the bf16 kernel through HBM / L2 / LDS and issues a quarter of its MFMA
// cycles, which is what counts at the socket's power cap.  Epilogue: the plain-bf16 one (scales
// sa[m] * sb[n] on the fp32 accumulator, bias, activation, LDS turn-around, 16-byte row-segment stores).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

struct Fp8P8Epi {
    bf16_t* c; const float* sa; const float* sb; const float* bias;
    int m, n, ldc, act, store_policy;
};

__device__ __forceinline__ unsigned dma_piece_offset_fp8(int ld, int r0, int k0, int pc, int lane) {
    const int row = pc * 16 + (lane >> 2);                 // 16 rows x 64 B per 1 KB piece, as dma_pieces32<false>
    const int ch = (lane & 3) ^ kc32_swz(row);
    return (unsigned)((size_t)(r0 + row) * ld + k0 + ch * 16);
}

// 16 rows x K = 128: lane l holds row (l & 15) and the 32 bytes of k-block (l >> 4): k-blocks 0-1 live in the even
// stage's 64-byte row, 2-3 in the odd stage's
__device__ __forceinline__ i32x8_t fp8_frag_pair(const char* even, const char* odd, int rbase, int lane) {
    const int row = rbase + (lane & 15);
    const int kb = lane >> 4;
    const char* base = ((kb & 2) ? odd : even) + row * 64;
    const int c0 = (kb & 1) * 2;
    const uint4 lo = *(const uint4*)(base + (((c0) ^ kc32_swz(row)) << 4));
    const uint4 hi = *(const uint4*)(base + (((c0 + 1) ^ kc32_swz(row)) << 4));
    i32x8_t r;
    r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
    r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
    return r;
}

__global__ __launch_bounds__(512, 1) void gemm_fp8_p8_kernel(
    const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, GemmDims d, Fp8P8Epi ep,
    unsigned a_bytes, unsigned b_bytes, int n_items, int rev_rounds, unsigned* __restrict__ tk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 x (A 16 KB | B 16 KB) + 16 B
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x, bid = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, b_bytes, 0x00020000);
    const int NX = G < 8 ? G : 8;
    const int xl = bid % NX;
    const int per_xcd = max(1, G / NX);
    auto lane_start = [&](int x) { return (int)((long long)n_items * (x * (G / NX) + min(x, G % NX)) / G); };
    const int x_base = lane_start(xl);
    const int x_cnt = lane_start(xl + 1) - x_base;
    volatile unsigned* bcast = (volatile unsigned*)(smem + 4 * P8_STAGE);
    const int npairs = d.k >> 7;                          // K = 128 per stage pair
    auto decode = [&](unsigned ticket, P8Item& w) -> bool {
        if (ticket >= (unsigned)x_cnt) return false;
        int tm, tn;
        tile_coords(x_base + (int)ticket, d.tiles_m, d.tiles_n, d.group_m, tm, tn);
        w.m0 = tm * 256; w.n0 = tn * 256; w.kbeg = 0; w.nk = npairs; w.z = 0;
        w.rev = rev_rounds ? (int)((ticket / (unsigned)per_xcd) & 1u) : 0;
        return true;
    };
    auto finish = [&]() {
        if (threadIdx.x == 0 && atomicAdd(tk + 8, 1u) == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) tk[i] = 0u;
        }
    };
    unsigned va[2], vb[2];
    auto set_bases = [&](const P8Item& it) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            va[p] = dma_piece_offset_fp8(d.lda, it.m0, 0, wave * 2 + p, lane);
            vb[p] = dma_piece_offset_fp8(d.ldb, it.n0, 0, wave * 2 + p, lane);
        }
        asm volatile("" : "+v"(va[0]), "+v"(va[1]), "+v"(vb[0]), "+v"(vb[1]));
    };
    // pair `pr` of item `w`: its even / odd stages go to slots (2 pr) & 3 and (2 pr + 1) & 3
    auto issue_pair = [&](const P8Item& w, int pr, const unsigned* pa, const unsigned* pb) {
        const int kp = w.rev ? w.nk - 1 - pr : pr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned delta = (unsigned)(kp * 128 + h * 64);
            char* dst = smem + ((2 * pr + h) & 3) * P8_STAGE + wave * 2048;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(dst + p * 1024), 16, pa[p] + delta, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(dst + 16384 + p * 1024), 16, pb[p] + delta, 0, 0, 0);
            }
        }
    };

    // Fragment addressing: a fragment is 16 rows x K = 128; lane l holds row (l & 15) and k-block (l >> 4) = two 16-byte
    // chunks of that row in the even (k-blocks 0-1) or odd (2-3) stage's slot.  A fragment's first row is a multiple of
    // 16, so the row swizzle kc32_swz depends on (l & 15) only: FOUR per-lane offsets (A / B x low / high chunk) serve
    // every fragment, which differ by compile-time constants (the per-fragment address arithmetic cost 35 spilled
    // registers and VALU work in the load blocks).
    const int frow = lane & 15, fkb = lane >> 4;
    const int fz = kc32_swz(frow), fc0 = (fkb & 1) * 2;
    const unsigned f_lo = (unsigned)(frow * 64 + (((fc0) ^ fz) << 4)), f_hi = (unsigned)(frow * 64 + (((fc0 + 1) ^ fz) << 4));
    const unsigned a_lo = f_lo + (unsigned)(wm * 128 * 64), a_hi = f_hi + (unsigned)(wm * 128 * 64);
    const unsigned b_lo = f_lo + 16384u + (unsigned)(wn * 64 * 64), b_hi = f_hi + 16384u + (unsigned)(wn * 64 * 64);
    auto frag = [&](const char* sel, unsigned lo_off, unsigned hi_off, int row16) -> i32x8_t {
        const uint4 lo = *(const uint4*)(sel + lo_off + row16 * 1024);
        const uint4 hi = *(const uint4*)(sel + hi_off + row16 * 1024);
        i32x8_t r;
        r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
        r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
        return r;
    };

    P8Item w;
    [[maybe_unused]] int round = 0;
    unsigned pend = 0;
    if (threadIdx.x == 0) bcast[0] = atomicAdd(tk + xl, 1u);
    __syncthreads();
    const unsigned first = __builtin_amdgcn_readfirstlane(bcast[0]);
    if (!decode(first, w)) { finish(); return; }
    set_bases(w);
    issue_pair(w, 0, va, vb);
    const int unit = 0x7F7F7F7F;                          // e8m0 2^0 in every byte: no block scaling

    while (true) {
        f32x4 acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef P8_PROBE
        const unsigned long long pcw = clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pair 0 (and the previous item's stores)
        __builtin_amdgcn_s_barrier();
#ifdef P8_PROBE
        if (bid == 0 && threadIdx.x == 0 && round == 2) p8_probe[6] = clock64() - pcw;
#endif
        if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind
        const int nk = w.nk;
#ifdef P8_PROBE
        const unsigned long long pc0 = clock64(), pw0 = wall_clock64();
#endif
        const int draw_pair = nk > 3 ? nk - 3 : 0;
        unsigned lane_zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
        for (int pr = 0; pr < nk; ++pr) {
            const char* ev = smem + ((2 * pr) & 3) * P8_STAGE;
            const char* od = smem + ((2 * pr + 1) & 3) * P8_STAGE;
            // ---------------- load block a
            const char* sel = (fkb & 2) ? od : ev;              // this lane's k-blocks live in the even or the odd stage
            i32x8_t bfr[4], af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = frag(sel, b_lo, b_hi, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag(sel, a_lo, a_hi, i);
            if (pr + 1 < nk) issue_pair(w, pr + 1, va, vb);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[j][i], 0, 0, 0, unit, 0, unit);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- load block b
            i32x8_t ag[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ag[i] = frag(sel, a_lo, a_hi, 4 + i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this pair's slots are free once the barrier is passed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // pair pr + 1 has landed
            // the next item is drawn here, right behind the wait: the atomic has a whole pair to return before the next one
            if (pr == draw_pair && threadIdx.x == 0) pend = atomicAdd(tk + xl + lane_zero, 1u);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][4 + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], ag[i], acc[j][4 + i], 0, 0, 0, unit, 0, unit);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        // this wave's 8 x 16 token scales for the epilogue: requested here, behind the last MFMA block, so that they
        // travel under the realign / publish barriers and the next item's prologue issue (one dependent global load per
        // 16 rows inside the store loop cost 2 k cycles each; held through the whole main loop they slowed the loop)
        float sa8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int mrow = w.m0 + wm * 128 + i * 16 + (lane & 15);
            sa8[i] = mrow < ep.m ? ep.sa[mrow] : 0.f;
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // realign the groups
#ifdef P8_PROBE
        const unsigned long long pc1 = clock64(), pw1 = wall_clock64();
        ++round;
#endif

        // next item's first pair goes out before this item's epilogue -- into slots 0-1; the epilogue stages through
        // slots 2-3 (an odd number of pairs ends in slots 0-1... so the LAST pair's slots are never the staging area's:
        // with nk even the last pair sat in slots 2-3 and every wave has passed the barrier behind its last read)
        P8Item nx;
        if (threadIdx.x == 0) bcast[0] = pend;
        __builtin_amdgcn_s_barrier();
        const unsigned nxt = __builtin_amdgcn_readfirstlane(bcast[0]);
        const bool has_next = decode(nxt, nx);
        unsigned na[2], nb[2];
        if (has_next) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                na[p] = dma_piece_offset_fp8(d.lda, nx.m0, 0, wave * 2 + p, lane);
                nb[p] = dma_piece_offset_fp8(d.ldb, nx.n0, 0, wave * 2 + p, lane);
            }
            issue_pair(nx, 0, na, nb);
        }

        // epilogue: scales, bias, activation in the MFMA layout; bf16 turn-around through this wave's 8 KB of slots 2-3
        char* stg = smem + 2 * P8_STAGE + wave * 8192;
        // channel scales first, in place (their registers are dead before the bias ones are live: the epilogue with both
        // sets held next to the 128 accumulators spilled 35 registers and took 29 k cycles per tile instead of 9 k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = w.n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            const float4 sj = n < ep.n ? *(const float4*)(ep.sb + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[j][i][0] *= sj.x; acc[j][i][1] *= sj.y; acc[j][i][2] *= sj.z; acc[j][i][3] *= sj.w;
            }
        }
        float4 bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = w.n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            bj[j] = (ep.bias && n < ep.n) ? *(const float4*)(ep.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        using AM = ActMath<bf16_t>;
        auto half_tile = [&](auto h_) {
            constexpr int H = decltype(h_)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int lr = ii * 16 + (lane & 15);
                const float sa = sa8[H * 4 + ii];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = acc[j][H * 4 + ii];
                    float v[4] = {fmaf(a[0], sa, bj[j].x), fmaf(a[1], sa, bj[j].y), fmaf(a[2], sa, bj[j].z),
                                  fmaf(a[3], sa, bj[j].w)};
                    if (ep.act == BF_ACT_QUICK_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = AM::quick(v[e]);
                    } else if (ep.act == BF_ACT_TANH_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = AM::tanh_gelu(v[e]);
                    } else if (ep.act == BF_ACT_ERF_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = AM::erf(v[e]);
                    }
                    bf16x4 pk;
                    pk[0] = (bf16_t)v[0]; pk[1] = (bf16_t)v[1]; pk[2] = (bf16_t)v[2]; pk[3] = (bf16_t)v[3];
                    const int cc4 = j * 4 + (lane >> 4);
                    *(bf16x4*)(stg + lr * 128 + (((cc4 >> 1) ^ ((lr >> 1) & 7)) << 4) + ((cc4 & 1) << 3)) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int lr = t * 8 + (lane >> 3), c16 = lane & 7;
                const uint4 q = *(const uint4*)(stg + lr * 128 + ((c16 ^ ((lr >> 1) & 7)) << 4));
                const int m = w.m0 + wm * 128 + H * 64 + lr;
                const int n = w.n0 + wn * 64 + c16 * 8;
                if (m < ep.m && n < ep.n) store16(ep.c + (size_t)m * ep.ldc + n, q, ep.store_policy & 1);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        half_tile(std::integral_constant<int, 0>{});
        half_tile(std::integral_constant<int, 1>{});
#ifdef P8_PROBE
        if (bid == 0 && threadIdx.x == 0 && round == 1) {
            p8_probe[0] = pc1 - pc0; p8_probe[1] = pw1 - pw0; p8_probe[2] = nk;
            p8_probe[3] = clock64() - pc1; p8_probe[4] = wall_clock64() - pw1; p8_probe[5] = 0;
        }
#endif
        if (!has_next) break;
        w = nx;
        va[0] = na[0]; va[1] = na[1]; vb[0] = nb[0]; vb[1] = nb[1];
    }
    finish();
}

// =====================================================================================
//                                   exact fp32 MFMA kernel
// =====================================================================================
#define F32_LD 132  // [16][128+4] k-major operand tiles

struct GemmDimsF {
    int m, n, k;
    int lda, ldb;
    int tiles_m, tiles_n;
    int k_per_split;
};

// KC: element (r,k) at p[r*ld+k]; tile = 128 r x 16 k, thread loads 2 float4 along k.
// KS: element (r,k) at p[k*ld+r]; tile = 16 k x 128 r, thread loads 2 float4 along r.
template <bool KS>
__device__ __forceinline__ void f32_tile_load(float4 (&reg)[2], const float* __restrict__ p, int ld,
                                              int r0, int rtot, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!KS) {
            const int row = q >> 2, kc = q & 3;
            const int r = r0 + row, k = k0 + kc * 4;
            if (r < rtot) {
                const float* src = p + (size_t)r * ld + k;
                if (k + 3 < kend) v = *(const float4*)src;
                else {
                    if (k < kend) v.x = src[0];
                    if (k + 1 < kend) v.y = src[1];
                    if (k + 2 < kend) v.z = src[2];
                }
            }
        } else {
            const int krow = q >> 5, rc = q & 31;
            const int k = k0 + krow, r = r0 + rc * 4;
            if (k < kend) {
                const float* src = p + (size_t)k * ld + r;
                if (r + 3 < rtot) v = *(const float4*)src;
                else {
                    if (r < rtot) v.x = src[0];
                    if (r + 1 < rtot) v.y = src[1];
                    if (r + 2 < rtot) v.z = src[2];
                }
            }
        }
        reg[i] = v;
    }
}
template <bool KS>
__device__ __forceinline__ void f32_tile_store(const float4 (&reg)[2], float* lds) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i;
        if (!KS) {
            const int row = q >> 2, kc = q & 3;
            lds[(kc * 4 + 0) * F32_LD + row] = reg[i].x;
            lds[(kc * 4 + 1) * F32_LD + row] = reg[i].y;
            lds[(kc * 4 + 2) * F32_LD + row] = reg[i].z;
            lds[(kc * 4 + 3) * F32_LD + row] = reg[i].w;
        } else {
            const int krow = q >> 5, rc = q & 31;
            *(float4*)(lds + krow * F32_LD + rc * 4) = reg[i];
        }
    }
}

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A,
                                                          const float* __restrict__ B, GemmDimsF d,
                                                          EpiParams ep, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* fs = (float*)smem;   // [buf][A 16*132 | B 16*132]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = d.tiles_m * d.tiles_n;
    const int z = blockIdx.x / tiles;
    const int t = xcd_remap(blockIdx.x - z * tiles, tiles);
    const int tm = t / d.tiles_n, tn = t - tm * d.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = z * d.k_per_split;
    const int kend = min(d.k, kbeg + d.k_per_split);
    const int nk = (kend - kbeg + 15) >> 4;
    const int TS = 16 * F32_LD;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2], rb[2];
    f32_tile_load<A_KS>(ra, A, d.lda, m0, d.m, kbeg, kend);
    f32_tile_load<B_KS>(rb, B, d.ldb, n0, d.n, kbeg, kend);
    f32_tile_store<A_KS>(ra, fs);
    f32_tile_store<B_KS>(rb, fs + TS);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const float* la = fs + (kt & 1) * 2 * TS;
        const float* lb = la + TS;
        const bool more = kt + 1 < nk;
        if (more) {
            f32_tile_load<A_KS>(ra, A, d.lda, m0, d.m, kbeg + (kt + 1) * 16, kend);
            f32_tile_load<B_KS>(rb, B, d.ldb, n0, d.n, kbeg + (kt + 1) * 16, kend);
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const int krow = kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = la[krow * F32_LD + wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = lb[krow * F32_LD + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* na = fs + ((kt + 1) & 1) * 2 * TS;
            f32_tile_store<A_KS>(ra, na);
            f32_tile_store<B_KS>(rb, na + TS);
        }
        __syncthreads();
    }
    // D layout 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* ct = fs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CT_LD + wn * 64 +
                   j * 32 + (lane & 31)] = acc[i][j][r];
    __syncthreads();
    float* slab = slabs ? slabs + (size_t)z * ep.m * ep.n : nullptr;
    tile_epilogue<float, float>(ct, ep, m0, n0, slab);
}

// exact-mode variant with a bf16 C is not needed: in exact mode every tensor is fp32.

// =====================================================================================
//                                split-K reduction + epilogue
// =====================================================================================
template <typename TC, typename TD>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs,
                                                            int split, EpiParams ep) {
    const int n4 = ep.n >> 2;
    const size_t total = (size_t)ep.m * n4;
    const size_t slab = (size_t)ep.m * ep.n;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
        float4 acc = *(const float4*)(slabs + (size_t)m * ep.n + n);
        for (int z = 1; z < split; ++z) {
            const float4 v = *(const float4*)(slabs + z * slab + (size_t)m * ep.n + n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        epilogue_store4<TC, TD>(ep, m, n, acc);
    }
}

// Reduction of a grouped launch's partial slabs: c_p[m, n] = sum_z slab_p[z][m][n] for every problem p (blockIdx.y),
// summed in split order, so the result does not depend on which workgroup produced which partial.
struct GroupedReduce {
    const float* slab[P8_MAX_GROUP];
    float* c[P8_MAX_GROUP];
    int m[P8_MAX_GROUP], n[P8_MAX_GROUP], ldc[P8_MAX_GROUP];
    int split;
    float alpha;
};
__global__ __launch_bounds__(256) void grouped_reduce_kernel(GroupedReduce r) {
    const int p = blockIdx.y;
    const int m_ = r.m[p], n_ = r.n[p], ldc = r.ldc[p];
    const float* __restrict__ sl = r.slab[p];
    float* __restrict__ c = r.c[p];
    const int n4 = n_ >> 2;
    const size_t total = (size_t)m_ * n4, slab = (size_t)m_ * n_;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
        float4 acc = *(const float4*)(sl + (size_t)m * n_ + n);
        for (int z = 1; z < r.split; ++z) {
            const float4 v = *(const float4*)(sl + z * slab + (size_t)m * n_ + n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc.x *= r.alpha; acc.y *= r.alpha; acc.z *= r.alpha; acc.w *= r.alpha;
        *(float4*)(c + (size_t)m * ldc + n) = acc;
    }
}

// =====================================================================================
//                                        host
// =====================================================================================
static thread_local int last_variant = -1;

extern "C" int bf_gemm_last_variant(void) { return last_variant; }

extern "C" int bf_gemm_set_variant(int variant) {       // atomic: may be called while other threads run GEMMs
    bf_tuning().gemm_pipe.store(variant < 0 ? -1 : variant);
    return BF_OK;
}

static int n_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 8) n_cu = 256;
    }
    return n_cu;
}

// The persistent 256x256 kernel hands out whole tiles; when the last round of tiles fills less than ~half of the CUs
// (N = 768: 591 tiles = 2.3 rounds) and K is deep, the tiles of that round are cut into K chunks that are ordinary work
// items of the SAME launch (p8_body's tail split): no extra launches, no slab pass, the tile's last chunk to arrive
// reduces and stores.  (Round 2's form -- the tail rows as a split-K launch of their own -- was removed in round 6.)
struct P8InKernelTail { int tiles, parts, kt; size_t bytes; };
static bool p8_inkernel_tail_plan(const bf_gemm_args* g, P8InKernelTail* plan) {
    const BfTuning& T = bf_tuning();
    if (!T.p8_bk64) return false;
    if (!bf_is_half(g->cdt) || g->trans_a || (g->b_kn && !T.p8_bk64_nn) || g->split_k > 1 || (g->k % 64) != 0 ||
        (g->n % 8) != 0 || (g->ldc % 8) != 0 || g->m < T.p8_min_m)
        return false;
    const int pipe_env = T.gemm_pipe.load();
    if (pipe_env >= 0 && pipe_env != 9) return false;
    const int n_cu = n_cus();
    const long items = (long)bf_cdiv(g->m, 256) * bf_cdiv(g->n, 256);
    const int min_tiles = T.p8_min_tiles >= 0 ? T.p8_min_tiles : n_cu / 2;
    if (items <= n_cu || items < min_tiles) return false;
    const long rem = items % n_cu;
    if (rem == 0 || rem > (long)(n_cu * 0.55) || rem > P8_TAIL_MAX) return false;
    const int nkt = g->k / 64;
    // A chunk costs a 256 KB write-through hand-off on top of its K tiles (and the last arriver reads them all back): it
    // pays only when a chunk is deep enough -- measured round 4 (tools/gemm_bench.py, the split against whole tiles with
    // chunks of any depth): K = 3072 in 3 chunks -5 %, K = 2048 in 6 chunks +-1 %, K = 512 / 768 in 2-8 chunks +12...19 %.
    int parts = (int)(n_cu / rem);
    if (parts > 8) parts = 8;
    if (parts > nkt / T.p8_tail_min_kt) parts = nkt / T.p8_tail_min_kt;
    if (parts < 2) return false;
    const int kt = bf_cdiv(nkt, parts);
    parts = bf_cdiv(nkt, kt);
    if (parts < 2) return false;
    plan->tiles = (int)rem; plan->parts = parts; plan->kt = kt;
    plan->bytes = (size_t)rem * parts * 262144;
    return plan->bytes < (1ull << 32);
}

extern "C" size_t bf_gemm_tail_workspace_bytes(const bf_gemm_args* g) {
    if (!g) return 0;
    P8InKernelTail ik;
    return p8_inkernel_tail_plan(g, &ik) ? ik.bytes : 0;
}

static int gemm_impl(const bf_gemm_args* g, bf_stream_t stream, int force_variant);

// Whether gemm_impl's choices below put this call on the persistent kernel's BK = 64 NT loop with the plain 16-bit
// epilogue -- the one instantiation that carries a low-rank addend (EPI 5).  Same predicates, no side effects.
static bool gemm_lowrank_ok(const bf_gemm_args* g) {
    if (!g || !bf_is_half(g->cdt) || g->c_dtype != g->cdt || g->trans_a || g->b_kn) return false;
    if (g->lr_k != 16 && g->lr_k != 32) return false;
    if (g->split_k > 1 || g->residual || g->c2 || g->accumulate || g->rows_in != 0 || g->aux) return false;
    if (g->act == BF_ACT_ERF_GELU_SAVEGRAD || g->act == BF_ACT_ERF_GELU_GRAD || g->act == BF_ACT_TANH_GELU_GRAD ||
        g->act == BF_ACT_MUL_AUX) return false;
    if (g->m <= 0 || g->n <= 0 || g->k <= 0 || (g->k % 64) || (g->n % 8) || (g->ldc % 8)) return false;
    const BfTuning& T = bf_tuning();
    const int pipe = T.gemm_pipe.load();
    if (!(pipe < 0 || pipe == 9) || !T.p8_bk64) return false;     // (8 forces the BK = 32 loop)
    const int min_tiles = T.p8_min_tiles >= 0 ? T.p8_min_tiles : n_cus() / 2;
    if (pipe < 0 && !(g->m >= T.p8_min_m && (long)bf_cdiv(g->m, 256) * bf_cdiv(g->n, 256) >= min_tiles)) return false;
    const size_t a_ext = ((size_t)(g->m - 1) * g->lda + g->k) * 2, b_ext = ((size_t)(g->n - 1) * g->ldb + g->k) * 2;
    return a_ext + 512ull * (size_t)g->lda < (1ull << 32) && b_ext + 512ull * (size_t)g->ldb < (1ull << 32);
}
extern "C" int bf_gemm_lowrank_ok(const bf_gemm_args* g) { return gemm_lowrank_ok(g) ? 1 : 0; }

// Ticket sets of the persistent kernel: 9 counters per launch (8 XCD lanes + finished workgroups), zero at
// rest (the last workgroup of a launch zeroes its set).  Launches on one stream run in order, so a per-stream
// ring of sets can never be shared by two running kernels; concurrent streams get rings of their own.
#define P8_TK_STREAMS 16
#define P8_TK_RING 64
#define P8_TK_WORDS 160            // per set: 9 ticket / exit counters (padded to 16) + 128 tail-tile arrival counters (+ pad)
__device__ unsigned p8_ticket_sets[P8_TK_STREAMS * P8_TK_RING * P8_TK_WORDS];
// One owner table per device (a `__device__` array has one copy per device); a 17th distinct stream on a device takes
// the least recently used slot after a device synchronise (the old owner's launches may still read their sets).
#define P8_TK_DEVICES 16
struct P8TicketDev {
    unsigned* base;
    hipStream_t owners[P8_TK_STREAMS];
    unsigned next[P8_TK_STREAMS];
    unsigned long long last_use[P8_TK_STREAMS];
    int n_owner;
};
static unsigned* p8_ticket_set(hipStream_t s) {
    static P8TicketDev devs[P8_TK_DEVICES];
    static unsigned long long clock = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= P8_TK_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    P8TicketDev& t = devs[dev];
    if (!t.base && hipGetSymbolAddress((void**)&t.base, HIP_SYMBOL(p8_ticket_sets)) != hipSuccess) return nullptr;
    int slot = -1;
    for (int i = 0; i < t.n_owner; ++i) if (t.owners[i] == s) { slot = i; break; }
    if (slot < 0) {
        if (t.n_owner < P8_TK_STREAMS) {
            slot = t.n_owner++;
        } else {
            slot = 0;
            for (int i = 1; i < P8_TK_STREAMS; ++i) if (t.last_use[i] < t.last_use[slot]) slot = i;
            if (hipDeviceSynchronize() != hipSuccess) return nullptr;     // rare: only when a slot changes hands
        }
        t.owners[slot] = s;
        t.next[slot] = 0;
    }
    t.last_use[slot] = ++clock;
    const unsigned set = t.next[slot]++ % P8_TK_RING;
    return t.base + ((size_t)slot * P8_TK_RING + set) * P8_TK_WORDS;
}

// Launcher of the persistent e4m3 kernel for bf_gemm_fp8 (gemm_fp8.hip): returns BF_OK when it took the call,
// 1 when the shape is not its (the caller then runs its 128 x 128 kernel).
int bf_gemm_fp8_p8_try(const void* a8, const float* sa, const void* b8, const float* sb, void* c, int c_dtype,
                       const float* bias, int act, int m, int n, int k, int lda, int ldb, int ldc, bf_stream_t stream) {
    const BfTuning& T = bf_tuning();
    const int on = T.fp8_p8;
    const int n_cu = n_cus();
    const long tiles = (long)bf_cdiv(m, 256) * bf_cdiv(n, 256);
    if (!on || c_dtype != BF_BF16 || m < 16384 || tiles < n_cu / 2 || (k % 128) != 0 || (n % 8) != 0 || (ldc % 8) != 0 ||
        tiles >= (1L << 30))
        return 1;
    hipStream_t s = (hipStream_t)stream;
    unsigned* tk8 = p8_ticket_set(s);
    if (!tk8) return 1;
    const int p8_group = T.p8_group_m, rev_env = T.p8_revk, pol = T.p8_nt;
    GemmDims d{m, n, k, lda, ldb, bf_cdiv(m, 256), bf_cdiv(n, 256), k, p8_group};
    Fp8P8Epi ep{(bf16_t*)c, sa, sb, bias, m, n, ldc, act, pol};
    const size_t a_ext = (size_t)(m - 1) * lda + k, b_ext = (size_t)(n - 1) * ldb + k;
    static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();
    if (!lds_ok) {
        (void)hipFuncSetAttribute((const void*)gemm_fp8_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P8_STAGE + 16);
        lds_ok = true;
    }
    const unsigned g8 = (unsigned)(n_cu & ~7);
    hipLaunchKernelGGL(gemm_fp8_p8_kernel, dim3(g8), dim3(512), 4 * P8_STAGE + 16, s, (const uint8_t*)a8, (const uint8_t*)b8, d,
                       ep, (unsigned)a_ext, (unsigned)b_ext, (int)tiles, rev_env, tk8);
    BF_CHECK_LAUNCH("bf_gemm_fp8");
    return BF_OK;
}

// ---- grouped deep dW = dY^T X (the weight gradients of one transformer layer in one launch) ------------------------
struct GroupedPlan { int split, kps; long tiles; size_t bytes; };
static bool grouped_tn_plan(const bf_gemm_tn_problem* p, int count, int k, int cdt, GroupedPlan* plan) {
    if (!p || count < 1 || count > P8_MAX_GROUP || !bf_is_half(cdt) || k < 16384 || (k % 32) != 0) return false;
    if (!bf_tuning().dw_grouped) return false;
    const int pipe_env = bf_tuning().gemm_pipe.load();
    if (pipe_env >= 0 && pipe_env != 8 && pipe_env != 9) return false;
    long tiles = 0;
    size_t elems = 0;
    for (int i = 0; i < count; ++i) {
        const bf_gemm_tn_problem& q = p[i];
        if (q.m <= 0 || q.n <= 0 || (q.m & 7) || (q.n & 7) || (q.ld_dy & 7) || (q.ld_x & 7) || (q.ldc & 3) ||
            q.ld_dy < q.m || q.ld_x < q.n || q.ldc < q.n || !q.dy || !q.x || !q.c || !bf_aligned16(q.dy) ||
            !bf_aligned16(q.x) || !bf_aligned16(q.c))
            return false;
        if (((size_t)(k - 1) * q.ld_dy + q.m) * 2 >= (1ull << 32) || ((size_t)(k - 1) * q.ld_x + q.n) * 2 >= (1ull << 32))
            return false;
        tiles += (long)bf_cdiv(q.m, 256) * bf_cdiv(q.n, 256);
        elems += (size_t)q.m * q.n;
    }
    const int n_cu = n_cus();
    if (tiles > n_cu / 2) return false;                 // enough tiles to fill the chip without a K split: plain GEMMs
    int split = (int)(n_cu / tiles);
    if (split > 64) split = 64;
    int kps = ((bf_cdiv(k, split) + 63) / 64) * 64;
    split = bf_cdiv(k, kps);
    if (split < 2) return false;
    plan->split = split; plan->kps = kps; plan->tiles = tiles;
    plan->bytes = (size_t)split * elems * sizeof(float);
    return true;
}

extern "C" size_t bf_gemm_tn_grouped_workspace_bytes(const bf_gemm_tn_problem* p, int count, int k, int cdt) {
    GroupedPlan plan;
    return grouped_tn_plan(p, count, k, cdt, &plan) ? plan.bytes : 0;
}

extern "C" int bf_gemm_tn_grouped(const bf_gemm_tn_problem* p, int count, int k, int cdt, float alpha, void* workspace,
                                  size_t workspace_bytes, bf_stream_t stream) {
    GroupedPlan plan;
    BF_REQUIRE(grouped_tn_plan(p, count, k, cdt, &plan), BF_EINVAL,
               "bf_gemm_tn_grouped: this group does not take the grouped kernel (bf_gemm_tn_grouped_workspace_bytes "
               "returns 0 for it: run the problems through bf_gemm one by one)");
    BF_REQUIRE(workspace && workspace_bytes >= plan.bytes && bf_aligned16(workspace), BF_EWORKSPACE,
               "bf_gemm_tn_grouped: needs %zu workspace bytes, got %zu", plan.bytes, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    unsigned* tk8 = p8_ticket_set(s);
    BF_REQUIRE(tk8 != nullptr, BF_ELAUNCH, "bf_gemm_tn_grouped: no ticket ring for this stream");
    P8Group grp;
    GroupedReduce red;
    memset(&grp, 0, sizeof(grp));
    memset(&red, 0, sizeof(red));
    float* ws = (float*)workspace;
    int base = 0, max_mn4 = 0;
    for (int i = 0; i < count; ++i) {
        const bf_gemm_tn_problem& q = p[i];
        grp.a[i] = q.dy; grp.b[i] = q.x; grp.slab[i] = ws;
        grp.m[i] = q.m; grp.n[i] = q.n; grp.lda[i] = q.ld_dy; grp.ldb[i] = q.ld_x;
        grp.tiles_n[i] = bf_cdiv(q.n, 256);
        grp.tile_base[i] = base;
        base += bf_cdiv(q.m, 256) * bf_cdiv(q.n, 256);
        grp.a_bytes[i] = (unsigned)(((size_t)(k - 1) * q.ld_dy + q.m) * 2);
        grp.b_bytes[i] = (unsigned)(((size_t)(k - 1) * q.ld_x + q.n) * 2);
        red.slab[i] = ws; red.c[i] = q.c; red.m[i] = q.m; red.n[i] = q.n; red.ldc[i] = q.ldc;
        ws += (size_t)plan.split * q.m * q.n;
        if (q.m * (q.n / 4) > max_mn4) max_mn4 = q.m * (q.n / 4);
    }
    grp.tile_base[count] = base;
    grp.count = count;
    red.split = plan.split; red.alpha = alpha;
    GemmDims d{0, 0, k, 0, 0, 0, 0, plan.kps, 1};
    EpiParams ep;
    memset(&ep, 0, sizeof(ep));
    ep.alpha = 1.0f;
    const int skew = bf_tuning().p8_revk | 2;                       // deep TN: prefetch distance 3 (both operands stream from HBM)
    const long items = plan.tiles * plan.split;
    const unsigned g8 = (unsigned)(n_cus() & ~7);
    last_variant = 8;
    if (cdt == BF_F16) {
        static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();
        if (!lds_ok) {
            (void)hipFuncSetAttribute((const void*)gemm_p8_grouped_tn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * P8_STAGE + 16);
            lds_ok = true;
        }
        hipLaunchKernelGGL(gemm_p8_grouped_tn_kernel<f16_t>, dim3(g8), dim3(512), 4 * P8_STAGE + 16, s, grp, d, ep, (int)items,
                           skew, tk8);
    } else {
        static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();
        if (!lds_ok) {
            (void)hipFuncSetAttribute((const void*)gemm_p8_grouped_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * P8_STAGE + 16);
            lds_ok = true;
        }
        hipLaunchKernelGGL(gemm_p8_grouped_tn_kernel<bf16_t>, dim3(g8), dim3(512), 4 * P8_STAGE + 16, s, grp, d, ep, (int)items,
                           skew, tk8);
    }
    BF_CHECK_LAUNCH("bf_gemm_tn_grouped");
    int rblocks = bf_cdiv(max_mn4, 256);
    if (rblocks > 1024) rblocks = 1024;
    hipLaunchKernelGGL(grouped_reduce_kernel, dim3((unsigned)rblocks, (unsigned)count), dim3(256), 0, s, red);
    BF_CHECK_LAUNCH("bf_gemm_tn_grouped/reduce");
    return BF_OK;
}

extern "C" int bf_gemm(const bf_gemm_args* g, bf_stream_t stream) {
    BF_REQUIRE(g != nullptr, BF_EINVAL, "bf_gemm: null args");
    return gemm_impl(g, stream, -1);
}

static int gemm_impl(const bf_gemm_args* g, bf_stream_t stream, int force_variant) {
    BF_REQUIRE(g != nullptr, BF_EINVAL, "bf_gemm: null args");
    BF_REQUIRE(g->m > 0 && g->n > 0 && g->k > 0, BF_EINVAL, "bf_gemm: m, n, k must be > 0 (%d %d %d)",
               g->m, g->n, g->k);
    BF_REQUIRE(g->cdt == BF_F32 || bf_is_half(g->cdt), BF_EINVAL, "bf_gemm: bad cdt %d", g->cdt);
    BF_REQUIRE(g->c_dtype == BF_F32 || g->c_dtype == g->cdt, BF_EINVAL,
               "bf_gemm: c_dtype must be BF_F32 or the compute dtype (cdt %d, c_dtype %d)", g->cdt, g->c_dtype);
    BF_REQUIRE(!(g->cdt == BF_F32 && g->c_dtype != BF_F32), BF_EINVAL,
               "bf_gemm: fp32 operands produce fp32 output only");
    BF_REQUIRE(!(g->accumulate && g->split_k > 1 && g->c_dtype != BF_F32), BF_EINVAL,
               "bf_gemm: accumulate into a bf16 C cannot be combined with split_k");
    BF_REQUIRE(g->act >= BF_ACT_NONE && g->act <= BF_ACT_MUL_AUX, BF_EINVAL, "bf_gemm: bad act");
    BF_REQUIRE((g->act != BF_ACT_ERF_GELU_GRAD && g->act != BF_ACT_TANH_GELU_GRAD && g->act != BF_ACT_MUL_AUX) || g->aux,
               BF_EINVAL, "bf_gemm: the GELU_GRAD / MUL_AUX activations need aux");
    BF_REQUIRE(g->act != BF_ACT_ERF_GELU_SAVEGRAD || g->pre, BF_EINVAL, "bf_gemm: BF_ACT_ERF_GELU_SAVEGRAD needs `pre`");
    const bool half = bf_is_half(g->cdt);
    const int vec = half ? 8 : 4;
    BF_REQUIRE((g->n & 3) == 0 && (g->ldc & 3) == 0 && g->ldc >= g->n, BF_EINVAL,
               "bf_gemm: n and ldc must be multiples of 4 (n=%d ldc=%d)", g->n, g->ldc);
    BF_REQUIRE(g->lda % vec == 0 && g->ldb % vec == 0, BF_EALIGN,
               "bf_gemm: lda/ldb must be multiples of %d elements (lda=%d ldb=%d)", vec, g->lda, g->ldb);
    if (half) {
        // 16-byte chunks along the contiguous dimension must not straddle the extent
        BF_REQUIRE(g->trans_a ? (g->m & 7) == 0 : (g->k & 7) == 0, BF_EINVAL,
                   "bf_gemm(bf16): contiguous extent of A must be a multiple of 8");
        BF_REQUIRE(g->b_kn ? (g->n & 7) == 0 : (g->k & 7) == 0, BF_EINVAL,
                   "bf_gemm(bf16): contiguous extent of B must be a multiple of 8");
    }
    BF_REQUIRE(g->lda >= (g->trans_a ? g->m : g->k) && g->ldb >= (g->b_kn ? g->n : g->k), BF_EINVAL,
               "bf_gemm: leading dimension smaller than the row length");
    BF_REQUIRE(bf_aligned16(g->a) && bf_aligned16(g->b) && bf_aligned16(g->c) &&
                   (!g->bias || bf_aligned16(g->bias)) && (!g->residual || bf_aligned16(g->residual)) &&
                   (!g->pre || bf_aligned16(g->pre)) && (!g->aux || bf_aligned16(g->aux)) &&
                   (!g->c2 || bf_aligned16(g->c2)),
               BF_EALIGN, "bf_gemm: pointers must be 16-byte aligned");
    BF_REQUIRE(!g->residual || ((g->ld_res & 3) == 0 && g->ld_res >= g->n), BF_EINVAL,
               "bf_gemm: bad ld_res %d", g->ld_res);
    BF_REQUIRE(!g->aux || ((g->ld_aux & 3) == 0 && g->ld_aux >= g->n), BF_EINVAL, "bf_gemm: bad ld_aux");
    BF_REQUIRE(g->rows_in >= 0 && (g->rows_in == 0 || (g->rows_out >= g->rows_in + g->row_off && g->row_off >= 0)),
               BF_EINVAL, "bf_gemm: bad row remap (%d -> %d + %d)", g->rows_in, g->rows_out, g->row_off);

    const int ktile = half ? 64 : 16;
    int split = g->split_k > 1 ? g->split_k : 1;
    int kps = ((bf_cdiv(g->k, split) + ktile - 1) / ktile) * ktile;
    split = bf_cdiv(g->k, kps);
    float* slabs = nullptr;
    if (split > 1) {
        const size_t need = (size_t)split * g->m * g->n * sizeof(float);
        BF_REQUIRE(g->workspace && g->workspace_bytes >= need, BF_EWORKSPACE,
                   "bf_gemm: split_k=%d needs %zu workspace bytes, got %zu", split, need, g->workspace_bytes);
        BF_REQUIRE(bf_aligned16(g->workspace), BF_EALIGN, "bf_gemm: unaligned workspace");
        slabs = (float*)g->workspace;
    }

    EpiParams ep;
    ep.c = g->c; ep.c2 = g->c2; ep.pre = g->pre; ep.aux = g->aux; ep.bias = g->bias;
    ep.residual = g->residual; ep.m = g->m; ep.n = g->n; ep.ldc = g->ldc; ep.ld_aux = g->ld_aux;
    ep.ld_res = g->ld_res; ep.resid_mod = g->resid_mod; ep.rows_in = g->rows_in;
    ep.rows_out = g->rows_out; ep.row_off = g->row_off; ep.act = g->act;
    ep.accumulate = g->accumulate; ep.alpha = g->alpha;
    ep.colsum = nullptr;
    ep.lr_u = g->lr_u; ep.lr_w = g->lr_w; ep.lr_k = g->lr_k;
    BF_REQUIRE(!g->lr_u || (g->lr_w && bf_aligned16(g->lr_u) && bf_aligned16(g->lr_w)), BF_EALIGN,
               "bf_gemm: lr_u needs lr_w, both 16-byte aligned");
    BF_REQUIRE(!g->lr_u || gemm_lowrank_ok(g), BF_EINVAL,
               "bf_gemm: this call cannot carry a low-rank addend (bf_gemm_lowrank_ok)");
    const BfTuning& T = bf_tuning();
    ep.store_policy = T.p8_nt;      // bit mask, see EpiParams::store_policy (measured in-step: DESIGN.md §6)
    bool colsum_fused = false;
    BF_REQUIRE(!g->colsum_out || (g->rows_in == 0 && bf_aligned16(g->colsum_out)), BF_EINVAL,
               "bf_gemm: colsum_out needs rows_in == 0 and a 16-byte aligned pointer");

    hipStream_t s = (hipStream_t)stream;
    const int tiles_m = bf_cdiv(g->m, 128), tiles_n = bf_cdiv(g->n, 128);
    const long nblk = (long)tiles_m * tiles_n * split;
    BF_REQUIRE(nblk < (1L << 31), BF_EINVAL, "bf_gemm: grid too large");
    dim3 grid((unsigned)nblk), block(256);
    const size_t lds = (size_t)128 * CT_LD * sizeof(float);   // 67,584 B >= operand buffers

    if (half) {
        const bool f16 = g->cdt == BF_F16;
        const int group_m_env = T.gemm_group_m;
        GemmDims d{g->m, g->n, g->k, g->lda, g->ldb, tiles_m, tiles_n, kps, group_m_env};
        // every launch macro below ends in ..._H(..., H): H = the 16-bit operand type of the call; the output type is
        // float (f32out) or H
#define BY_HALF(M, ...) do { if (f16) M(__VA_ARGS__, f16_t); else M(__VA_ARGS__, bf16_t); } while (0)
#define LAUNCH_BF16_T(AKS, BKS, TC, H)                                                             \
    do {                                                                                          \
        static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();  /* > 64 KB of dynamic LDS must be opted into, once */        \
        if (!lds_ok) {                                                                            \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AKS, BKS, TC, H>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            lds_ok = true;                                                                        \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_bf16_kernel<AKS, BKS, TC, H>), grid, block, lds, s, (const H*)g->a, (const H*)g->b, d, ep, slabs); \
    } while (0)
#define LAUNCH_BF16_H(AKS, BKS, H) do { if (f32out) LAUNCH_BF16_T(AKS, BKS, float, H); else LAUNCH_BF16_T(AKS, BKS, H, H); } while (0)
#define LAUNCH_BF16(AKS, BKS) BY_HALF(LAUNCH_BF16_H, AKS, BKS)
        const bool f32out = (g->c_dtype == BF_F32) || slabs;   // slabs path ignores TC
        // LDS-DMA kernel: K-contiguous operands need whole 64-wide K slabs (no in-row tail),
        // and the byte extents must fit the 32-bit buffer descriptor.
        const size_t a_ext = ((size_t)((g->trans_a ? g->k : g->m) - 1) * g->lda + (g->trans_a ? g->m : g->k)) * 2;
        const size_t b_ext = ((size_t)((g->b_kn ? g->k : g->n) - 1) * g->ldb + (g->b_kn ? g->n : g->k)) * 2;
        const bool dma_ok = ((g->trans_a && g->b_kn) || (g->k % 64) == 0) && a_ext < (1ull << 32) &&
                            b_ext < (1ull << 32);
        const int pipe_env = T.gemm_pipe.load();
        // Kernel choice (measured in-step at B = 256): the BK = 32 kernel runs 4
        // workgroups per CU, which overlaps the epilogue traffic of one tile with the MFMAs of the
        // others and wins on every NN / TN shape and on short-K NT; long-K, large NT GEMMs
        // (K >= 2048 with >= 2e7 outputs) keep the BK = 64 kernel (half the barriers per FLOP).
        // BIFOLD_GEMM_PIPE forces a variant: 0 = BK64, 4 = BK32.  (Measured and dropped, see
        // 256x128 and 256x256 blocks, 128x64 wave tiles, 3/4-deep rings with counted
        // vmcnt -- every one lost to "as many resident workgroups as possible".)
        const bool long_nt = !g->trans_a && !g->b_kn && g->k >= 2048 && (double)g->m * g->n >= 2.0e7;
        // The persistent 256x256 kernel wins in-step on the large-M NT GEMMs (vision / fusion
        // token GEMMs, every epilogue flavour); the text tower's (M = 19.7k, on the side stream
        // next to the vision tower) and the NN / TN shapes stay with the multi-workgroup kernels.
        const int p8_min_m = T.p8_min_m;
        const int p8_min_tiles = T.p8_min_tiles >= 0 ? T.p8_min_tiles : n_cus() / 2;   // at least this many 256x256 tiles (default: one per two CUs)
        const bool p8_nt = !g->trans_a && !g->b_kn && g->m >= p8_min_m &&
                           (long)bf_cdiv(g->m, 256) * bf_cdiv(g->n, 256) >= p8_min_tiles;
        // deep dW = dY^T X GEMMs whose (tile, split) items fill the CUs in exactly one round
        const long items256 = (long)bf_cdiv(g->m, 256) * bf_cdiv(g->n, 256) * split;
        const bool p8_tn = g->trans_a && g->b_kn && split > 1 && items256 > 192 && items256 <= 256;
        // dX = dY W (NN): like NT (with the spill-free instantiations the 512-wide ones win too, in spite of
        // their 2.1 rounds of tiles)
        const bool p8_nn = !g->trans_a && g->b_kn && g->m >= p8_min_m &&
                           (long)bf_cdiv(g->m, 256) * bf_cdiv(g->n, 256) >= p8_min_tiles;
        int variant = force_variant >= 0 ? force_variant
                                         : (pipe_env >= 0 ? pipe_env : ((p8_nt || p8_tn || p8_nn) ? 8 : (long_nt ? 0 : 4)));
        // the persistent kernel draws its items from a per-stream ring of ticket sets; a process that uses
        // more streams than there are rings runs the extra streams' GEMMs on the other kernels
        // variant 9 = the persistent kernel with the BK = 64 full-line main loop for NT (K-contiguous operands, K a
        // multiple of 64): the automatic choice for those shapes
        bool bk64 = T.p8_bk64 != 0;
        if (variant == 9 || (force_variant == 8 && pipe_env == 9)) { variant = 8; bk64 = true; }
        else if (pipe_env == 8) bk64 = false;
        bk64 = bk64 && !g->trans_a && (!g->b_kn || T.p8_bk64_nn) && (g->k % 64) == 0 && (kps % 64) == 0 &&
               a_ext + 512ull * (size_t)g->lda < (1ull << 32) && b_ext + 512ull * (size_t)g->ldb < (1ull << 32);   // 32-bit offsets of the tile rows past M / N
        unsigned* tk8 = variant == 8 ? p8_ticket_set(s) : nullptr;
        if (variant == 8 && !tk8) variant = long_nt ? 0 : 4;
        BF_REQUIRE(!g->lr_u || variant == 8, BF_EINVAL, "bf_gemm: low-rank addend on a stream without a ticket ring (variant %d)", variant);
        last_variant = 1;   // register-staged fallback unless one of the LDS-DMA kernels below takes it
        if (dma_ok && variant == 8 && (g->k % 32) == 0 && (kps % 32) == 0 && (g->n % 8) == 0 && (g->ldc % 8) == 0 &&
            (!g->aux || (g->ld_aux % 8) == 0)) {
            last_variant = 8;
            // column sums fused into the bf16 GELU-grad epilogue (same predicate as the kernel's `fastgrad`)
            if (g->colsum_out && !slabs && !f32out && (g->act == BF_ACT_ERF_GELU_GRAD || g->act == BF_ACT_MUL_AUX) && !g->bias && !g->residual &&
                !g->c2 && !g->pre && !g->accumulate && g->rows_in == 0) {   // == the EPI 2 instantiation below
                BF_REQUIRE(bf_zero_f32_async(g->colsum_out, (size_t)g->n, s) == hipSuccess, BF_ELAUNCH,
                           "bf_gemm: zeroing colsum_out failed");
                ep.colsum = g->colsum_out;
                colsum_fused = true;
            }
            const unsigned ab = (unsigned)a_ext, bb = (unsigned)b_ext;
            const int n_cu = n_cus();
            GemmDims d8 = d;
            d8.tiles_m = bf_cdiv(g->m, 256); d8.tiles_n = bf_cdiv(g->n, 256);
            d8.group_m = T.p8_group_m;
            long items = (long)d8.tiles_m * d8.tiles_n * split;
            BF_REQUIRE(items < (1L << 30), BF_EINVAL, "bf_gemm: too many tiles");
            P8InKernelTail ik;
            if (bk64 && split == 1 && !slabs && g->workspace && bf_aligned16(g->workspace) && p8_inkernel_tail_plan(g, &ik) &&
                g->workspace_bytes >= ik.bytes) {
                d8.tail_tiles = ik.tiles; d8.tail_parts = ik.parts; d8.tail_kt = ik.kt;
                d8.tail_ws = (float*)g->workspace;
                items += (long)ik.tiles * (ik.parts - 1);
            }
            unsigned g8 = (unsigned)(n_cu & ~7);      // every CU draws; workgroups that find no ticket exit
            if (T.p8_grid >= 1 && (unsigned)T.p8_grid < g8) g8 = (unsigned)T.p8_grid;
            // bit 0: alternate the K direction per round of tickets (operand fetches 338 -> 239 MB on the vision QKV
            // GEMM, no change in time: the re-fetches were Infinity-Cache hits); bit 1: prefetch distance 3, which
            // pays only where both operands stream from HBM, i.e. the deep dW = dY^T X GEMMs (+5 %; NT / NN -3..5 %)
            const int rev_env = T.p8_revk, deep_env = T.p8_deep;
            const int deep = deep_env >= 0 ? (deep_env ? 1 : 0) : (g->trans_a && g->b_kn ? 1 : 0);
            int skew = rev_env | (deep << 1);
            // which epilogue instantiation (the predicates of the kernel's EPI paths)
            int epi = 0;
            if (!f32out && !slabs && !g->residual && !g->c2 && !g->accumulate && g->rows_in == 0) {
                if (!g->aux && g->act == BF_ACT_ERF_GELU_SAVEGRAD) epi = 3;
                else if (!g->aux && g->act != BF_ACT_ERF_GELU_GRAD && g->act != BF_ACT_TANH_GELU_GRAD && g->act != BF_ACT_MUL_AUX) epi = 1;
                else if ((g->act == BF_ACT_ERF_GELU_GRAD || g->act == BF_ACT_MUL_AUX) && !g->bias && !g->pre) epi = 2;
            }
            if (g->lr_u) {
                BF_REQUIRE(epi == 1 && bk64 && !g->trans_a && !g->b_kn, BF_EINVAL,
                           "bf_gemm: the low-rank addend did not reach the kernel that carries it (epi %d, bk64 %d)", epi, (int)bk64);
                epi = 5;
            }
#define LAUNCH_P8E(AKS, BKS, TC, EPI, H, BKV)                                                       \
    do {                                                                                          \
        static bool lds_ok[P8_TK_DEVICES];  /* > 64 KB of dynamic LDS must be opted into, once per device */ \
        int dev_ = 0;                                                                             \
        (void)hipGetDevice(&dev_);                                                                \
        if (dev_ < 0 || dev_ >= P8_TK_DEVICES || !lds_ok[dev_]) {                                 \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_p8_kernel<AKS, BKS, TC, EPI, H, BKV>, \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * P8_STAGE + 16);  \
            if (dev_ >= 0 && dev_ < P8_TK_DEVICES) lds_ok[dev_] = true;                           \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_bf16_p8_kernel<AKS, BKS, TC, EPI, H, BKV>), dim3(g8), dim3(512), 4 * P8_STAGE + 16, s, (const H*)g->a, (const H*)g->b, d8, ep, slabs, ab, bb, (int)items, skew, tk8); \
    } while (0)
#define LAUNCH_P8_HB(AKS, BKS, H, BKV)                                                              \
    do {                                                                                          \
        if (f32out) LAUNCH_P8E(AKS, BKS, float, 0, H, BKV);                                       \
        else if (epi == 1) LAUNCH_P8E(AKS, BKS, H, 1, H, BKV);                                    \
        else if (epi == 2) LAUNCH_P8E(AKS, BKS, H, 2, H, BKV);                                    \
        else if (epi == 3) LAUNCH_P8E(AKS, BKS, H, 3, H, BKV);                                    \
        else LAUNCH_P8E(AKS, BKS, H, 0, H, BKV);                                                  \
    } while (0)
#define LAUNCH_P8_H(AKS, BKS, H) LAUNCH_P8_HB(AKS, BKS, H, 32)
#define LAUNCH_P8_H64(AKS, BKS, H) LAUNCH_P8_HB(AKS, BKS, H, 64)
#define LAUNCH_P8(AKS, BKS) BY_HALF(LAUNCH_P8_H, AKS, BKS)
            if (epi == 5) {                                   // (one instantiation per operand type: NT, BK = 64)
                if (f16) LAUNCH_P8E(false, false, f16_t, 5, f16_t, 64); else LAUNCH_P8E(false, false, bf16_t, 5, bf16_t, 64);
            }
            else if (!g->trans_a && !g->b_kn) {
                if (bk64) BY_HALF(LAUNCH_P8_H64, false, false); else LAUNCH_P8(false, false);
            }
            else if (!g->trans_a && g->b_kn) {
                if (bk64) BY_HALF(LAUNCH_P8_H64, false, true); else LAUNCH_P8(false, true);
            }
            else if (g->trans_a && g->b_kn) LAUNCH_P8(true, true);
            else LAUNCH_P8(true, false);
#undef LAUNCH_P8E
#undef LAUNCH_P8_HB
#undef LAUNCH_P8_H
#undef LAUNCH_P8_H64
#undef LAUNCH_P8
        } else if (dma_ok && (variant == 4 || (variant == 8 && !long_nt))) {
            last_variant = 4;
            const unsigned ab = (unsigned)a_ext, bb = (unsigned)b_ext;
            // (64-row tiles fill the chip's workgroup slots better on some shapes and still lost -- more L2 -> LDS bytes per
            // FLOP -- removed in round 6)
#define LAUNCH_K32_T(AKS, BKS, TC, H)                                                                \
    hipLaunchKernelGGL((gemm_bf16_k32_kernel<AKS, BKS, TC, 4, H>), grid, block, 32768, s, (const H*)g->a, (const H*)g->b, d, ep, slabs, ab, bb)
#define LAUNCH_K32_H(AKS, BKS, H) do { if (f32out) LAUNCH_K32_T(AKS, BKS, float, H); else LAUNCH_K32_T(AKS, BKS, H, H); } while (0)
#define LAUNCH_K32(AKS, BKS) BY_HALF(LAUNCH_K32_H, AKS, BKS)
            if (!g->trans_a && !g->b_kn) LAUNCH_K32(false, false);
            else if (!g->trans_a && g->b_kn) LAUNCH_K32(false, true);
            else if (g->trans_a && g->b_kn) LAUNCH_K32(true, true);
            else LAUNCH_K32(true, false);
#undef LAUNCH_K32_T
#undef LAUNCH_K32_H
#undef LAUNCH_K32
        } else if (dma_ok) {
            last_variant = 0;
            const size_t lds2 = 65536;
            const unsigned ab = (unsigned)a_ext, bb = (unsigned)b_ext;
#define LAUNCH_DMA_T(AKS, BKS, TC, H)                                                                \
    do {                                                                                          \
        static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();                                                               \
        if (!lds_ok) {                                                                            \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<AKS, BKS, TC, H>,         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);     \
            lds_ok = true;                                                                        \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<AKS, BKS, TC, H>), grid, block, lds2, s, (const H*)g->a, (const H*)g->b, d, ep, slabs, ab, bb); \
    } while (0)
#define LAUNCH_DMA_H(AKS, BKS, H) do { if (f32out) LAUNCH_DMA_T(AKS, BKS, float, H); else LAUNCH_DMA_T(AKS, BKS, H, H); } while (0)
#define LAUNCH_DMA(AKS, BKS) BY_HALF(LAUNCH_DMA_H, AKS, BKS)
            if (!g->trans_a && !g->b_kn) LAUNCH_DMA(false, false);
            else if (!g->trans_a && g->b_kn) LAUNCH_DMA(false, true);
            else if (g->trans_a && g->b_kn) LAUNCH_DMA(true, true);
            else LAUNCH_DMA(true, false);
#undef LAUNCH_DMA_T
#undef LAUNCH_DMA_H
#undef LAUNCH_DMA
        } else
        if (!g->trans_a && !g->b_kn) LAUNCH_BF16(false, false);
        else if (!g->trans_a && g->b_kn) LAUNCH_BF16(false, true);
        else if (g->trans_a && g->b_kn) LAUNCH_BF16(true, true);
        else LAUNCH_BF16(true, false);
#undef LAUNCH_BF16_T
#undef LAUNCH_BF16_H
#undef LAUNCH_BF16
#undef BY_HALF
    } else {
        last_variant = -1;
        GemmDimsF d{g->m, g->n, g->k, g->lda, g->ldb, tiles_m, tiles_n, kps};
        const float* A = (const float*)g->a;
        const float* B = (const float*)g->b;
#define LAUNCH_F32(AKS, BKS)                                                                       \
    do {                                                                                          \
        static BfPerDevice lds_ok_dev; size_t& lds_ok = lds_ok_dev.cur();                                                               \
        if (!lds_ok) {                                                                            \
            (void)hipFuncSetAttribute((const void*)gemm_f32_kernel<AKS, BKS>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            lds_ok = true;                                                                        \
        }                                                                                         \
        hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS>), grid, block, lds, s, A, B, d, ep, slabs);  \
    } while (0)
        if (!g->trans_a && !g->b_kn) LAUNCH_F32(false, false);
        else if (!g->trans_a && g->b_kn) LAUNCH_F32(false, true);
        else if (g->trans_a && g->b_kn) LAUNCH_F32(true, true);
        else LAUNCH_F32(true, false);
#undef LAUNCH_F32
    }
    BF_CHECK_LAUNCH("bf_gemm");

    if (slabs) {
        const size_t total = (size_t)g->m * (g->n / 4);
        size_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        dim3 rgrid((unsigned)blocks);
        if (g->cdt == BF_BF16) {
            if (g->c_dtype == BF_F32)
                hipLaunchKernelGGL((splitk_reduce_kernel<float, bf16_t>), rgrid, block, 0, s, slabs, split, ep);
            else
                hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t, bf16_t>), rgrid, block, 0, s, slabs, split, ep);
        } else if (g->cdt == BF_F16) {
            if (g->c_dtype == BF_F32)
                hipLaunchKernelGGL((splitk_reduce_kernel<float, f16_t>), rgrid, block, 0, s, slabs, split, ep);
            else
                hipLaunchKernelGGL((splitk_reduce_kernel<f16_t, f16_t>), rgrid, block, 0, s, slabs, split, ep);
        } else {
            hipLaunchKernelGGL((splitk_reduce_kernel<float, float>), rgrid, block, 0, s, slabs, split, ep);
        }
        BF_CHECK_LAUNCH("bf_gemm/splitk_reduce");
    }
    if (g->colsum_out && !colsum_fused)
        return bf_colsum(g->c, g->c_dtype, g->colsum_out, g->m, g->n, g->ldc, 0, stream);
    return BF_OK;
}
