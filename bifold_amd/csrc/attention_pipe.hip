// Software-pipelined attention kernels for sequences whose K / V do not fit LDS (round 6): the SigLIP towers at 384 px
// (576 tokens, d = 64), ViT-L/14 @ 336 (577 tokens, d = 64), the sequential fusion transformer (642 - 2 950 tokens,
// d = 48, query-row mask).  Replaces F.scaled-dot-product of ref:bifold/models/transformer.py:28-41 (fusion) and the
// attention inside HF SigLIP / CLIP's nn.MultiheadAttention (ref:bifold/models/siglip.py:56-70,102-104,
// ref:bifold/models/clip/model.py:214) on the packed QKV buffer [B, L, 3, H, D].
//
// Why a new body.  The tiled forward of rounds 1 - 5 (attn_fwd_mfma_kernel) runs produce-then-consume: a wave issues the
// S = K Q^T MFMAs of a key tile, then ~30 vector instructions per MFMA of softmax, then the P V MFMAs, between two
// barriers per tile with the K / V^T staging (eight 2-byte LDS writes per thread to transpose V) in the open.  On a
// CDNA4 SIMD the matrix and vector pipes of ONE wave's stream only overlap when independent work of both kinds is
// adjacent in program order (profiles/r05_pipe_overlap.txt: 544 -> 372 cycles per key step hand-placed).  Here that
// adjacency is built into the algorithm:
//
//  * a wave owns TWO 32-query blocks a and b (64 queries, 4 waves = 256 queries per workgroup) and walks the 32-key
//    blocks j with the two blocks half a step apart.  Per key block two regions, each one basic block:
//        R1(j):  softmax of S(a, j)          beside the MFMAs of   S(b, j) = K_j Q_b^T   and   O_b += V_{j-1}^T P(b, j-1)
//        R2(j):  softmax of S(b, j)          beside the MFMAs of   O_a += V_j^T P(a, j)  and   S(a, j+1) = K_{j+1} Q_a^T
//    so every region holds 7 (d = 48) or 8 (d = 64) MFMAs that do not depend on the region's vector work, and the
//    compiler's scheduler (with sched_group_barrier hints) interleaves them;
//  * the softmax reference point of a block is FIXED after the first key tile (its row maximum): p = exp2(s * c - m)
//    with no row maximum, no rescale and no half-wave exchange in the loop.  The only hazard is p leaving the 16-bit
//    range, and a lane's partial row sum (needed anyway) bounds every p of the lane: a region ends with ONE test of
//    that sum against 2^14, and the rare block that fails it re-reads its scores, moves m to the true maximum,
//    rescales l and O and recomputes p (the "fixup" path, exercised by test_attention_fwd_rows_that_keep_rising);
//  * query-row mask (ref:bifold/models/transformer.py:33-35: masked QUERY rows get the constant score -1e5, i.e. a
//    uniform softmax): the masked rows' Q fragments are zeroed once, which gives exactly that uniform softmax with no
//    mask code in the loop; their lse is written as the reference's (-1e5 + ln L);
//  * K and V tiles (64 keys) live row-major in a three-slot LDS ring, V^T fragments come from ds_read_b64_tr_b16; the
//    next tiles travel in registers under the current tile's regions (K two tiles ahead, V one: the pipeline reads
//    K one block ahead and V one block behind) and are written at the tile's end: ONE barrier per 64 keys;
//  * the key tail (L % 64) is masked in a peeled copy of the last tile only.
#include "attn_common.h"
#include "tuning.h"

#include <type_traits>

template <int D, int WPB> struct PipeCfg {
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int KS = D / 16, DB = DP / 32;
    static constexpr int KROW = D * 2 + 16;                 // bytes: K rows (conflict-free ds_read_b128 of 32 rows)
    static constexpr int VROW = DP * 2 + 16;                // bytes: V rows, DP columns (pad columns of d = 48 are zero)
    static constexpr int CPR = D / 8;                       // 16-byte chunks of a K / V row
    static constexpr int NT = WPB * 64;
    static constexpr int PIECES = 64 * CPR;                 // chunks of one 64-row tile of one operand
    static constexpr int NPF = (PIECES + NT - 1) / NT;      // ... per thread
    static constexpr int KTILE = 64 * KROW, VTILE = 64 * VROW, SLOT = KTILE + VTILE;
    static constexpr int LDS = 3 * SLOT;
};

typedef unsigned pipe_u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's uint4 struct keeps arrays of it in scratch

// LLVM SchedGroupMask bits
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_DSR 0x100
#define SG_TRANS 0x400

template <int D, int MASK, int WPB, typename HT>
__global__ __launch_bounds__(WPB * 64, 2) void attn_fwd_pipe_kernel(
    const HT* __restrict__ qkv, HT* __restrict__ out, float* __restrict__ lse,
    const int32_t* __restrict__ row_mask, int L, int H, float scale) {
    using C = PipeCfg<D, WPB>;
    using x8 = typename H16<HT>::x8;
    using x4 = typename H16<HT>::x4;
    constexpr int KS = C::KS, DB = C::DB, KROW = C::KROW, VROW = C::VROW, CPR = C::CPR, NT = C::NT, NPF = C::NPF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int q0 = (blockIdx.y * WPB + wave) * 64;             // first query of this wave (block a; block b = q0 + 32)
    const bool wave_live = q0 < L;
    const int ntiles = (L + 63) >> 6;
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u;       // one token of the packed QKV buffer
    const char* kbase = (const char*)(qkv + qkv_off(b, 0, 1, h, L, H, D));
    const char* vbase = (const char*)(qkv + qkv_off(b, 0, 2, h, L, H, D));

    // ---- tile staging: NPF 16-byte pieces per thread and operand -----------------------------------------------
    // loads are unconditional from clamped rows (the compiler then counts vmcnt instead of draining it); rows beyond L
    // hold a copy of row L - 1: finite, and their scores are masked in the last tile
    auto fetch = [&](pipe_u32x4 (&p)[NPF], const char* base, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = min((int)threadIdx.x + i * NT, C::PIECES - 1);
            const int jl = idx / CPR, c = idx - jl * CPR;
            const int j = min(t * 64 + jl, L - 1);
            p[i] = *(const pipe_u32x4*)(base + (size_t)((unsigned)j * row_bytes + (unsigned)c * 16u));
        }
    };
    auto stash = [&](const pipe_u32x4 (&p)[NPF], char* tile, int rowb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int jl = idx / CPR, c = idx - jl * CPR;
            if ((i + 1) * NT <= C::PIECES || idx < C::PIECES) *(pipe_u32x4*)(tile + jl * rowb + c * 16) = p[i];
        }
    };

    // prologue: K(0), K(1), V(0); V pad columns (d = 48) and the whole V part of slot 2 (read once as "V(-1)" with P = 0)
    {
        pipe_u32x4 pk0[NPF], pk1[NPF], pv0[NPF];
        fetch(pk0, kbase, 0);
        fetch(pk1, kbase, 1);
        fetch(pv0, vbase, 0);
        for (int idx = threadIdx.x; idx < C::VTILE / 16; idx += NT) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            *(uint4*)(smem + 2 * C::SLOT + C::KTILE + idx * 16) = z;
            if (C::DP != D) {
                *(uint4*)(smem + 0 * C::SLOT + C::KTILE + idx * 16) = z;
                *(uint4*)(smem + 1 * C::SLOT + C::KTILE + idx * 16) = z;
            }
        }
        __syncthreads();                                          // the zero fill before V(0)'s rows land in slot 0
        stash(pk0, smem, KROW);
        stash(pk1, smem + C::SLOT, KROW);
        stash(pv0, smem + C::KTILE, VROW);
    }

    // ---- this wave's queries ------------------------------------------------------------------------------------
    x8 qa[KS], qb[KS];
    bool masked_a = false, masked_b = false;
    {
        const int ia = q0 + r, ib = q0 + 32 + r;
        const bool live_a = ia < L, live_b = ib < L;
        if (MASK == BF_MASK_QUERY_ROW) {
            masked_a = live_a && row_mask[(size_t)b * L + ia] == 0;
            masked_b = live_b && row_mask[(size_t)b * L + ib] == 0;
        }
        const char* qbase = kbase - (size_t)H * D * 2;            // operand 0 of the same token row
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            uint4 va = z, vb = z;
            if (live_a && !masked_a) va = *(const uint4*)(qbase + (size_t)((unsigned)ia * row_bytes + (unsigned)(ks * 32 + hh * 16)));
            if (live_b && !masked_b) vb = *(const uint4*)(qbase + (size_t)((unsigned)ib * row_bytes + (unsigned)(ks * 32 + hh * 16)));
            qa[ks] = *(const x8*)&va;
            qb[ks] = *(const x8*)&vb;
        }
    }
    const float sl2 = scale * 1.44269504088896340736f;           // scores in log2 units
    const int koff = r * KROW + hh * 16;                          // lane-constant parts of the fragment addresses
    const int voff = (4 * hh + q4) * VROW + (16 * (g16 & 1) + 4 * p4) * 2;

    f32x16 oa[DB], ob[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oa[db][e] = 0.f; ob[db][e] = 0.f; }
    float ma = 0.f, mb = 0.f, la = 0.f, lb = 0.f;                // reference points (log2 units) and per-LANE row sums
    f32x16 sa, sb;                                                // scores / probabilities of the block in flight
    x8 pa[2], pb[2];                                              // packed probabilities: the B operands of P V
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { pa[s][e] = (HT)0.f; pb[s][e] = (HT)0.f; }

    // S = K_blk Q^T (keys on the accumulator's rows, the lane's query on its column)
    auto s_block = [&](f32x16& acc, const char* kblk, const x8 (&q)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const x8 kf = *(const x8*)(kblk + koff + ks * 32);
            acc = mfma_32x32x16(kf, q[ks], acc);
        }
    };
    // O^T += V_blk^T P^T
    auto pv_block = [&](f32x16 (&o)[DB], const char* vblk, const x8 (&pp)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* p0 = vblk + voff + (16 * s) * VROW + db * 64;
                const x8 vf = tr_pair<HT>(p0, p0 + 8 * VROW);
                o[db] = mfma_32x32x16(vf, pp[s], o[db]);
            }
    };
    // p = exp2(s * sl2 - m) in place, packed into pp; returns the lane's partial row sum.  EDGE: keys >= L give p = 0.
    auto probs = [&](f32x16& s, float m, x8 (&pp)[2], int key0, bool edge) __attribute__((always_inline)) -> float {
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float t = fmaf(s[e], sl2, -m);
            if (edge) {
                const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (key >= L) t = -INFINITY;
            }
            const float p = __builtin_amdgcn_exp2f(t);
            s[e] = p;
            if (e & 1) ps1 += p; else ps0 += p;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) pp[u][e] = (HT)s[8 * u + e];
        return ps0 + ps1;
    };
    // the rare block whose probabilities left the safe range: scores again, true row maximum, m moved, l / O rescaled
    auto fixup = [&](f32x16& s, const char* kblk, const x8 (&q)[KS], float& m, float& l, f32x16 (&o)[DB], x8 (&pp)[2],
                     int key0, bool edge) __attribute__((always_inline)) -> float {
        s_block(s, kblk, q);
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (!edge || key < L) tmax = fmaxf(tmax, s[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;          // sl2 > 0: max commutes with the scale
        const float m_new = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        l *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
        m = m_new;
        return probs(s, m, pp, key0, edge);
    };
    auto unsafe = [&](float ps) __attribute__((always_inline)) -> bool { return __builtin_amdgcn_ballot_w64(!(ps < 16384.0f)) != 0; };

    __syncthreads();                                              // K(0), K(1), V(0) are in LDS

    if (wave_live) {
        // reference points: the row maxima over the first key tile (64 keys; L >= 64 is the launcher's precondition)
        float ta = -INFINITY, tb = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s_block(sa, smem + kb * 32 * KROW, qa);
            s_block(sb, smem + kb * 32 * KROW, qb);
#pragma unroll
            for (int e = 0; e < 16; ++e) { ta = fmaxf(ta, sa[e]); tb = fmaxf(tb, sb[e]); }
        }
        ma = fmaxf(ta, __shfl_xor(ta, 32, 64)) * sl2;
        mb = fmaxf(tb, __shfl_xor(tb, 32, 64)) * sl2;
        s_block(sa, smem, qa);                                // the pipeline's head: S(a, 0)
    }

    // One key tile = two key blocks = four regions.  EDGE is the peeled last tile (key tail masked).
    // Ring slots as byte offsets that ROTATE (cur = tile t, nxt = tile t + 1, prv = tile t - 1 = where tile t + 2 lands):
    // a tile's fragment addresses are then one vector add per operand and immediate offsets.
    int s_cur = 0, s_nxt = C::SLOT, s_prv = 2 * C::SLOT;
    auto tile_body = [&](int t, const bool edge) __attribute__((always_inline)) {
        // the next tiles, in flight under this tile's regions.  Unconditional: beyond the last tile the rows clamp to
        // L - 1 and the slots they land in are only read for products nobody uses (S(a, j + 1) of the last block)
        pipe_u32x4 nk[NPF], nv[NPF];
        fetch(nk, kbase, t + 2);
        fetch(nv, vbase, t + 1);
        if (wave_live) {
            const char* kt = smem + s_cur;
            const char* vt = smem + s_cur + C::KTILE;
            const char* kn = smem + s_nxt;
            const char* vp = smem + s_prv + C::KTILE;              // V of tile t - 1 (zeros, with P = 0, at t = 0)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key0 = t * 64 + kb * 32;
                const char* kcur = kt + kb * 32 * KROW;
                const char* vcur = vt + kb * 32 * VROW;
                const char* vprev = kb ? vt : vp + 32 * VROW;      // V of block j - 1
                const char* knext = kb ? kn : kt + 32 * KROW;      // K of block j + 1
                // ---- R1: softmax(a, j)  ||  S(b, j), O_b += V_{j-1}^T P(b, j-1)
                __builtin_amdgcn_sched_barrier(0);
                s_block(sb, kcur, qb);
                pv_block(ob, vprev, pb);
                float ps = probs(sa, ma, pa, key0, edge);
                __builtin_amdgcn_sched_barrier(0);
                if (unsafe(ps)) ps = fixup(sa, kcur, qa, ma, la, oa, pa, key0, edge);
                la += ps;
                // ---- R2: softmax(b, j)  ||  O_a += V_j^T P(a, j), S(a, j+1)
                __builtin_amdgcn_sched_barrier(0);
                pv_block(oa, vcur, pa);
                s_block(sa, knext, qa);
                ps = probs(sb, mb, pb, key0, edge);
                __builtin_amdgcn_sched_barrier(0);
                if (unsafe(ps)) ps = fixup(sb, kcur, qb, mb, lb, ob, pb, key0, edge);
                lb += ps;
            }
        }
        stash(nk, smem + s_prv, KROW);                             // K(t + 2) over K(t - 1)
        stash(nv, smem + s_nxt + C::KTILE, VROW);                  // V(t + 1) over V(t - 2)
        __syncthreads();
        const int s_old = s_cur;
        s_cur = s_nxt; s_nxt = s_prv; s_prv = s_old;
    };
    for (int t = 0; t < ntiles - 1; ++t) tile_body(t, false);
    tile_body(ntiles - 1, (L & 63) != 0);

    // ---- drain: O_b += V_last^T P(b, last); normalise; turn the tile through LDS; store whole row segments --------
    // (the ring is idle: the loop's last barrier is behind every wave)
    char* ot = smem + wave * (64 * KROW);                          // this wave's 64 output rows, KROW-byte rows
    static_assert(WPB * 64 * KROW <= C::LDS, "output turn fits the ring");
    // The pending P V of block b needs V of the last tile (after the rotation: slot s_prv), which nothing has
    // overwritten (V(t + 1) of the last iteration went to another slot); the output turn below reuses the ring, so the
    // product comes first, then a barrier, then the turn.
    if (wave_live) pv_block(ob, smem + s_prv + C::KTILE + 32 * VROW, pb);
    __syncthreads();
    if (wave_live) {
        la += __shfl_xor(la, 32, 64);                             // the two half-rows' partial sums
        lb += __shfl_xor(lb, 32, 64);
        const float inv_a = 1.0f / la, inv_b = 1.0f / lb;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;
                if (d0 < D) {
                    x4 a4, b4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a4[e] = (HT)(oa[db][4 * g + e] * inv_a); b4[e] = (HT)(ob[db][4 * g + e] * inv_b); }
                    *(x4*)(ot + r * KROW + d0 * 2) = a4;
                    *(x4*)(ot + (32 + r) * KROW + d0 * 2) = b4;
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int idx = lane; idx < 64 * CPR; idx += 64) {
            const int row = idx / CPR, c = idx - row * CPR;
            if (q0 + row < L)
                *(uint4*)(out + ((size_t)b * L + q0 + row) * (size_t)H * D + (size_t)h * D + c * 8) =
                    *(const uint4*)(ot + row * KROW + c * 16);
        }
        if (lse && hh == 0) {
            const float ln2 = 0.69314718055994530942f;
            // masked query rows: the reference's scores are the constant -1e5 for every key
            const float mask_m = -100000.0f * 1.44269504088896340736f;
            if (q0 + r < L) lse[((size_t)b * H + h) * L + q0 + r] = ((masked_a ? mask_m : ma) + log2f(la)) * ln2;
            if (q0 + 32 + r < L) lse[((size_t)b * H + h) * L + q0 + 32 + r] = ((masked_b ? mask_m : mb) + log2f(lb)) * ln2;
        }
    }
}

// =====================================================================================
//                                        host
// =====================================================================================
template <int D, int MASK, int WPB, typename HT>
static void fwd_pipe_launch_one(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq,
                                int heads, float scale, hipStream_t s) {
    using C = PipeCfg<D, WPB>;
    static BfPerDevice lds_set_dev;
    size_t& lds_set = lds_set_dev.cur();
    if ((size_t)C::LDS > lds_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel<D, MASK, WPB, HT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        lds_set = C::LDS;
    }
    const int n64 = bf_cdiv(seq, 64);
    dim3 grid(batch * heads, bf_cdiv(n64, WPB)), block(WPB * 64);
    hipLaunchKernelGGL((attn_fwd_pipe_kernel<D, MASK, WPB, HT>), grid, block, C::LDS, s, (const HT*)qkv, (HT*)out, lse,
                       row_mask, seq, heads, scale);
}

int bf_attn_fwd_pipe_launch(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq, int heads,
                            int head_dim, float scale, int mask_mode, int cdt, hipStream_t s) {
    if (!bf_is_half(cdt) || seq < 64 || mask_mode == BF_MASK_CAUSAL || (head_dim != 48 && head_dim != 64)) return 1;
    if ((size_t)seq * 3u * heads * head_dim * 2u >= (1ull << 32)) return 1;   // 32-bit row offsets inside one batch element
    const int n64 = bf_cdiv(seq, 64);
    // waves per workgroup: 4, or 3 when that leaves fewer idle waves (576 tokens = 9 x 64 = three workgroups of three)
    const int wpb = (bf_cdiv(n64, 3) * 3 < bf_cdiv(n64, 4) * 4) ? 3 : 4;
    const bool f16 = cdt == BF_F16;
#define PIPE4(DD, MM, WW)                                                                                            \
    do {                                                                                                             \
        if (f16) fwd_pipe_launch_one<DD, MM, WW, f16_t>(qkv, out, lse, row_mask, batch, seq, heads, scale, s);       \
        else fwd_pipe_launch_one<DD, MM, WW, bf16_t>(qkv, out, lse, row_mask, batch, seq, heads, scale, s);          \
    } while (0)
#define PIPE3(DD, MM) do { if (wpb == 3) PIPE4(DD, MM, 3); else PIPE4(DD, MM, 4); } while (0)
#define PIPE2(DD) do { if (mask_mode == BF_MASK_NONE) PIPE3(DD, BF_MASK_NONE); else PIPE3(DD, BF_MASK_QUERY_ROW); } while (0)
    if (head_dim == 48) PIPE2(48); else PIPE2(64);
#undef PIPE2
#undef PIPE3
#undef PIPE4
    return BF_OK;
}
