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

#define PIPE_MAXBLK 1024                                     // 32-row query blocks per sequence (L <= 32 768): the mask lists's capacity

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
    static constexpr int RING = 3 * SLOT;
    static constexpr int LDS = RING + 2 * PIPE_MAXBLK * 2;   // + the live / masked query-block lists (query-row mask)
};

typedef unsigned pipe_u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's uint4 struct keeps arrays of it in scratch

// LLVM SchedGroupMask bits
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_DSR 0x100
#define SG_TRANS 0x400

// Build-time forms of a region (A/B'd on the GPU; the defaults are what won):
#ifndef PIPE_SCHED
#define PIPE_SCHED 0        // 1: sched_group_barrier pipeline -- every LDS read first, a run of vector work in front of the first MFMA
#endif
#ifndef PIPE_PACK_IN
#define PIPE_PACK_IN 0      // 1: the fp32 -> 16-bit conversions of P inside the region (beside the MFMAs) instead of behind its test
#endif
#ifndef PIPE_LEAD_V
#define PIPE_LEAD_V 10      // PIPE_SCHED 1: vector / transcendental instructions ahead of the first MFMA
#endif
#ifndef PIPE_LEAD_T
#define PIPE_LEAD_T 4
#endif
#ifndef PIPE_BWD_FENCE
#define PIPE_BWD_FENCE 0    // backward kernels: 1 = scheduling fences between a chunk's four regions; 0 = the chunk body is one scheduling region (measured: 14.3 -> 13.97 ms the pair at 2 950 x 48)
#endif
#if PIPE_BWD_FENCE
#define BWD_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BWD_FENCE() do {} while (0)
#endif

// the scheduling pipeline of one region: NDS LDS reads, NM MFMAs
template <int NM, int NDS>
__device__ __forceinline__ void region_hints() {
#if PIPE_SCHED == 1
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 2, 0);
    __builtin_amdgcn_sched_group_barrier(SG_DSR, NDS, 0);
    __builtin_amdgcn_sched_group_barrier(SG_VALU, PIPE_LEAD_V, 0);
    __builtin_amdgcn_sched_group_barrier(SG_TRANS, PIPE_LEAD_T, 0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(SG_VALU, 4, 0);
        __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
    }
#endif
}


// Query-row mask: which 32-row query blocks are masked ENTIRELY (every row that exists is masked)?  Every workgroup of a
// (batch, head) builds the same two lists in LDS, ascending (deterministic): `blist` the blocks with a live row, `mlist` the
// entirely masked ones.  counts[0 / 1] = their lengths.  NT threads; ends with a barrier.
template <int NT>
__device__ __forceinline__ void build_block_lists(const int32_t* __restrict__ mask_b, int L, int nblk, unsigned short* blist,
                                                  unsigned short* mlist, int* counts, int (*wave_cnt)[NT / 64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += NT) {
        const int q = base + threadIdx.x;
        int any_bits = 0;                                    // 32 independent loads (a short-circuit loop is 32 trips to memory)
        if (q < nblk) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = q * 32 + i;
                const int v = mask_b[min(row, L - 1)];
                any_bits |= row < L ? v : 0;
            }
        }
        const bool any_live = any_bits != 0;
        const unsigned long long live_bits = __builtin_amdgcn_ballot_w64(q < nblk && any_live);
        const unsigned long long mask_bits = __builtin_amdgcn_ballot_w64(q < nblk && !any_live);
        if (base == 0 && threadIdx.x == 0) { counts[0] = 0; counts[1] = 0; }
        if (lane == 0) { wave_cnt[0][wave] = __popcll(live_bits); wave_cnt[1][wave] = __popcll(mask_bits); }
        __syncthreads();
        int off_l = counts[0], off_m = counts[1];
        for (int w = 0; w < wave; ++w) { off_l += wave_cnt[0][w]; off_m += wave_cnt[1][w]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (q < nblk) {
            if (any_live) blist[off_l + __popcll(live_bits & below)] = (unsigned short)q;
            else mlist[off_m + __popcll(mask_bits & below)] = (unsigned short)q;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tl = 0, tm = 0;
            for (int w = 0; w < NT / 64; ++w) { tl += wave_cnt[0][w]; tm += wave_cnt[1][w]; }
            counts[0] += tl; counts[1] += tm;
        }
        __syncthreads();
    }
}

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
    const int ntiles = (L + 63) >> 6;
    const int nblk = (L + 31) >> 5;
    // This wave's two 32-query blocks.  Without a mask: blocks 2 w and 2 w + 1 of the sequence.  Under the query-row mask
    // the blocks that are masked ENTIRELY all have the same output row (a uniform softmax: the mean of V) -- siglip_sequential
    // masks whole context frames, two of four on average -- so the waves walk the list of LIVE blocks plus ONE
    // representative masked block, whose wave copies its row to every other masked block at the end.
    unsigned short* blist = (unsigned short*)(smem + C::RING);
    unsigned short* mlist = blist + PIPE_MAXBLK;
    __shared__ int counts[2];
    __shared__ int wave_cnt[2][WPB];
    int n_live = nblk, n_masked = 0;
    if (MASK == BF_MASK_QUERY_ROW) {
        build_block_lists<NT>(row_mask + (size_t)b * L, L, nblk, blist, mlist, counts, wave_cnt);
        n_live = counts[0];
        n_masked = counts[1];
    }
    const int n_work = n_live + (n_masked > 0 ? 1 : 0);
    if ((int)blockIdx.y * WPB * 2 >= n_work) return;              // nothing for this workgroup (uniform, before any barrier below)
    const int li0 = (blockIdx.y * WPB + wave) * 2;
    const bool wave_live = li0 < n_work;
    auto work_block = [&](int li) __attribute__((always_inline)) -> int {     // list entry -> 32-row block, -1: none
        if (li >= n_work) return -1;
        if (MASK != BF_MASK_QUERY_ROW) return li;
        return li < n_live ? (int)blist[li] : (int)mlist[0];
    };
    const int blk_a = work_block(li0), blk_b = work_block(li0 + 1);
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
        const int ia = blk_a * 32 + r, ib = blk_b * 32 + r;
        const bool live_a = blk_a >= 0 && ia < L, live_b = blk_b >= 0 && ib < L;
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
#if PIPE_PACK_IN
                asm volatile("" :: "v"(pa[0]), "v"(pa[1]));
#endif
                region_hints<KS + 2 * DB, KS + 4 * DB>();
                __builtin_amdgcn_sched_barrier(0);
                if (unsafe(ps)) ps = fixup(sa, kcur, qa, ma, la, oa, pa, key0, edge);
                la += ps;
                // ---- R2: softmax(b, j)  ||  O_a += V_j^T P(a, j), S(a, j+1)
                __builtin_amdgcn_sched_barrier(0);
                pv_block(oa, vcur, pa);
                s_block(sa, knext, qa);
                ps = probs(sb, mb, pb, key0, edge);
#if PIPE_PACK_IN
                asm volatile("" :: "v"(pb[0]), "v"(pb[1]));
#endif
                region_hints<KS + 2 * DB, KS + 4 * DB>();
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
    static_assert(WPB * 64 * KROW <= C::RING, "output turn fits the ring");
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
            const int blk = row < 32 ? blk_a : blk_b;
            const int grow = blk * 32 + (row & 31);
            if (blk >= 0 && grow < L)
                *(uint4*)(out + ((size_t)b * L + grow) * (size_t)H * D + (size_t)h * D + c * 8) =
                    *(const uint4*)(ot + row * KROW + c * 16);
        }
        const float ln2 = 0.69314718055994530942f;
        // masked query rows: the reference's scores are the constant -1e5 for every key
        const float mask_m = -100000.0f * 1.44269504088896340736f;
        const float lse_a = ((masked_a ? mask_m : ma) + log2f(la)) * ln2, lse_b = ((masked_b ? mask_m : mb) + log2f(lb)) * ln2;
        if (lse && hh == 0) {
            const int ia = blk_a * 32 + r, ib = blk_b * 32 + r;
            if (blk_a >= 0 && ia < L) lse[((size_t)b * H + h) * L + ia] = lse_a;
            if (blk_b >= 0 && ib < L) lse[((size_t)b * H + h) * L + ib] = lse_b;
        }
        // the representative masked block (list entry n_live): its row goes to every other entirely masked block
        if (MASK == BF_MASK_QUERY_ROW && n_masked > 1 && (li0 == n_live || li0 + 1 == n_live)) {
            const int rep = li0 == n_live ? 0 : 32;                    // row of `ot` holding the block's first row (all rows are equal)
            const float lse_m = __shfl(li0 == n_live ? lse_a : lse_b, 0, 64);
            for (int m = 1; m < n_masked; ++m) {
                const int blk = (int)mlist[m];
#pragma unroll
                for (int idx = lane; idx < 32 * CPR; idx += 64) {
                    const int row = idx / CPR, c = idx - row * CPR;
                    const int grow = blk * 32 + row;
                    if (grow < L)
                        *(uint4*)(out + ((size_t)b * L + grow) * (size_t)H * D + (size_t)h * D + c * 8) =
                            *(const uint4*)(ot + rep * KROW + c * 16);
                }
                if (lse && lane < 32 && blk * 32 + lane < L) lse[((size_t)b * H + h) * L + blk * 32 + lane] = lse_m;
            }
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
    if (!bf_is_half(cdt) || seq < 64 || seq > 32 * PIPE_MAXBLK || mask_mode == BF_MASK_CAUSAL) return 1;
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
    if (head_dim == 32) PIPE2(32); else if (head_dim == 48) PIPE2(48); else PIPE2(64);
#undef PIPE2
#undef PIPE3
#undef PIPE4
    return BF_OK;
}

// =====================================================================================
//   Backward for the same sequences: dK / dV with the KEY on the lane, pipelined (round 6)
// =====================================================================================
// The streamed pair of rounds 2 - 5 (attn_bwd_kv_stream_kernel, attention.hip) spends, per 32 x 32 block and wave, 14 MFMAs
// and ~40 vector cycles per score strictly one after the other at two waves per SIMD: 12.2 ms per launch at 2 950 x 48,
// B = 128 (18 % of the MFMA peak).  This kernel restates it the way the forward above is stated:
//
//  * a wave owns TWO 32-key blocks a and b (64 keys; dK^T / dV^T of both in 128 accumulator registers) and runs ONE wave
//    per SIMD with the whole register file; the 32-query blocks j walk through a four-slot LDS ring, 64 rows per slot;
//  * per query block two regions, each one basic block with vector work of one key block beside MFMAs of both:
//        R1(j):  p = exp2(c S'), dS = p dP'  of (a, j)    beside   dV^T += dO_{j-1}^T P,  dK^T += Q_{j-1}^T dS   of a AND b
//        R2(j):  the same of (b, j)                      beside   S' = Q_{j+1} K^T - lse / c,  dP' = dO_{j+1} V^T - delta   of a AND b
//    the Q / dO row fragments, the row constants and the transposed fragments are read ONCE for both key blocks;
//  * the row constants -lse / scale and -delta are the initial accumulators of S' and dP' (ds_read straight into them);
//    the factor `scale` of dS is applied once, to dK, at the end; keys beyond L need no mask (their columns are not stored);
//  * query-row mask (ref:bifold/models/transformer.py:33-35): a masked query row has P = 1 / L for every key and dS = 0.
//    Inside a live block that is a Q row of zeros and the row constant -ln(L) / scale in LDS: no mask code in the loop.
//    Query blocks that are masked ENTIRELY (siglip_sequential masks whole context frames: two of four on average) are
//    not walked at all: their only contribution is dV[key] += (1 / L) sum_q dO[q], the same vector for every key -- the
//    workgroup sums those dO rows once, in a fixed order, and every wave adds the vector to its dV at the end.
template <int D> struct KvPipeCfg {
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int KS = D / 16, DB = DP / 32;
    static constexpr int ROW = D * 2 + 16;                   // bytes per Q / dO row (the last transposed fragment of d = 48 reads into the next row: finite, unused)
    static constexpr int CPR = D / 8;
    static constexpr int CH = 64;                            // query rows per ring slot (two blocks)
    static constexpr int NT = 256;
    static constexpr int PIECES = CH * CPR;
    static constexpr int NPF = (PIECES + NT - 1) / NT;
    static constexpr int IMG = CH * ROW + 64;                // one operand's rows (+ the overhang of the last row's transposed reads)
    static constexpr int SLOT = 2 * IMG + 2 * CH * 4;        // Q rows, dO rows, -lse / scale, -delta
    static constexpr int NSLOT = 4;
    static constexpr int MAXBLK = PIPE_MAXBLK;                      // 32-row blocks per sequence (L <= 32 768)
    static constexpr int LISTS = 2 * MAXBLK * 2 + 64 * 4;    // live / masked block lists (uint16) + the masked dO sum
    static constexpr int LDS = NSLOT * SLOT + LISTS;
};

template <int D, int MASK, typename HT>
__global__ __launch_bounds__(256, 1) void attn_bwd_kv_pipe_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ delta_ws, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    int L, int H, float scale) {
    using C = KvPipeCfg<D>;
    using x8 = typename H16<HT>::x8;
    using x4 = typename H16<HT>::x4;
    constexpr int KS = C::KS, DB = C::DB, ROW = C::ROW, CPR = C::CPR, NT = C::NT, NPF = C::NPF, CH = C::CH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* blist = (unsigned short*)(smem + C::NSLOT * C::SLOT);          // live query blocks, ascending
    unsigned short* mlist = blist + C::MAXBLK;                                       // entirely masked query blocks
    float* msum = (float*)(mlist + C::MAXBLK);                                       // sum of their dO rows [DP]
    __shared__ int counts[2];                                                        // n_live, n_masked
    __shared__ int wave_cnt[2][4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int k0 = (blockIdx.y * 4 + wave) * 64;                 // this wave's keys: block a = k0 .., block b = k0 + 32 ..
    const bool wave_live = k0 < L;
    const int nblk = (L + 31) >> 5;
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u;         // one token of the packed QKV buffer
    const unsigned orow_bytes = (unsigned)H * D * 2u;             // ... of out / dout
    const char* qbase = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));
    const char* dobase = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
    const float* lse_b = lse + ((size_t)b * H + h) * L;
    const float* del_b = delta_ws + ((size_t)b * H + h) * L;
    const int32_t* mask_b = MASK == BF_MASK_QUERY_ROW ? row_mask + (size_t)b * L : nullptr;
    const float inv_scale = 1.0f / scale;
    const float sl2 = scale * 1.44269504088896340736f;
    const float inv_len = 1.0f / (float)L;

    // ---- which query blocks are walked ------------------------------------------------------------------------
    int n_live = nblk, n_masked = 0;
    if (MASK == BF_MASK_QUERY_ROW) {
        // block q is "masked" when every row of it that exists is masked; lists in ascending order (deterministic)
        build_block_lists<NT>(mask_b, L, nblk, blist, mlist, counts, wave_cnt);
        n_live = counts[0];
        n_masked = counts[1];
        // sum of the dO rows of the masked blocks: thread (row-in-block, chunk) adds its chunk over the blocks in list
        // order, the 32 partial rows are then added in row order -- one fixed order, bit-reproducible
        float part[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) part[e] = 0.f;
        const int ri = threadIdx.x / CPR, ci = threadIdx.x - ri * CPR;
        if (ri < 32) {
            for (int m0 = 0; m0 < n_masked; m0 += 8) {            // eight rows in flight, added in list order
                x8 v[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = min(m0 + u, n_masked - 1);
                    const int row = (int)mlist[m] * 32 + ri;
                    ok[u] = m0 + u < n_masked && row < L;
                    v[u] = *(const x8*)(dobase + (size_t)((unsigned)min(row, L - 1) * orow_bytes + (unsigned)ci * 16u));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) part[e] += ok[u] ? (float)v[u][e] : 0.f;
            }
        }
        float* red = (float*)smem;                                 // [32 rows][DP] partial sums (the ring is not in use yet)
        if (ri < 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[ri * C::DP + ci * 8 + e] = part[e];
        }
        __syncthreads();
        if (threadIdx.x < C::DP) {
            float t = 0.f;
            if (threadIdx.x < D)
                for (int i = 0; i < 32; ++i) t += red[i * C::DP + threadIdx.x];
            msum[threadIdx.x] = t * inv_len;
        }
        __syncthreads();
    }
    const int nchunks = max(1, (n_live + 1) >> 1);

    // ---- ring staging: chunk c = live blocks 2c and 2c + 1 ---------------------------------------------------
    // piece idx of a chunk: row jl = idx / CPR (0 .. 63), 16-byte chunk idx % CPR, of Q and of dO; rows that do not exist
    // (beyond L, or a block index beyond the list) are zero rows with p = 0
    auto chunk_row = [&](int c, int jl) __attribute__((always_inline)) -> int {
        const int li = 2 * c + (jl >> 5);
        if (li >= n_live) return -1;
        const int blk = MASK == BF_MASK_QUERY_ROW ? (int)blist[li] : li;
        const int row = blk * 32 + (jl & 31);
        return row < L ? row : -1;
    };
    // What a fetch leaves in registers is RAW: every select that depends on a loaded value (zero rows, masked rows, the row
    // constants) happens when the registers are written to LDS, a chunk later -- a select at fetch time makes the wave wait
    // for its loads on the spot, and the ring's whole point is that they fly under the regions (measured: 4x).
    struct Staged { pipe_u32x4 q[NPF], g[NPF]; int row[NPF], mk[NPF]; float lse, del; int crow, cmk; };
    auto fetch = [&](Staged& st, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = min((int)threadIdx.x + i * NT, C::PIECES - 1);
            const int jl = idx / CPR, cc = idx - jl * CPR;
            const int row = chunk_row(c, jl);
            const int rc = max(row, 0);
            st.q[i] = *(const pipe_u32x4*)(qbase + (size_t)((unsigned)rc * row_bytes + (unsigned)cc * 16u));
            st.g[i] = *(const pipe_u32x4*)(dobase + (size_t)((unsigned)rc * orow_bytes + (unsigned)cc * 16u));
            st.row[i] = row;
            st.mk[i] = 1;
            if (MASK == BF_MASK_QUERY_ROW) st.mk[i] = mask_b[rc];
        }
        const int crow = chunk_row(c, min((int)threadIdx.x, CH - 1));
        const int crc = max(crow, 0);
        st.crow = crow;
        st.lse = lse_b[crc];
        st.del = del_b[crc];
        st.cmk = 1;
        if (MASK == BF_MASK_QUERY_ROW) st.cmk = mask_b[crc];
    };
    auto stash = [&](const Staged& st, char* slot) __attribute__((always_inline)) {
        const pipe_u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int jl = idx / CPR, cc = idx - jl * CPR;
            if ((i + 1) * NT <= C::PIECES || idx < C::PIECES) {
                const bool dead = st.row[i] < 0;
                // a masked row inside a live block: Q = 0 (uniform P with the row constant below, nothing into dK)
                *(pipe_u32x4*)(slot + jl * ROW + cc * 16) = (dead || st.mk[i] == 0) ? z : st.q[i];
                *(pipe_u32x4*)(slot + C::IMG + jl * ROW + cc * 16) = dead ? z : st.g[i];
            }
        }
        if (threadIdx.x < CH) {
            float cs = -1e30f, cd = 0.f;                           // a row that does not exist: p = 0, dS = 0
            if (st.crow >= 0) {
                cs = st.cmk == 0 ? -__logf((float)L) * inv_scale : -st.lse * inv_scale;
                cd = -st.del;
            }
            ((float*)(slot + 2 * C::IMG))[threadIdx.x] = cs;
            ((float*)(slot + 2 * C::IMG))[CH + threadIdx.x] = cd;
        }
    };
    {
        Staged s0, s1;
        fetch(s0, 0);
        fetch(s1, 1);
        // slot 3 stands in for "chunk -1" (read once with P = dS = 0): zero rows
        for (int idx = threadIdx.x; idx < 2 * C::IMG / 16; idx += NT)
            *(pipe_u32x4*)(smem + 3 * C::SLOT + idx * 16) = (pipe_u32x4){0u, 0u, 0u, 0u};
        stash(s0, smem);
        stash(s1, smem + C::SLOT);
    }

    // ---- this wave's keys ------------------------------------------------------------------------------------
    x8 kfa[KS], vfa[KS], kfb[KS], vfb[KS];
    {
        const int ka = k0 + r, kb = k0 + 32 + r;
        const pipe_u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pipe_u32x4 a = z, b2 = z, c2 = z, d2 = z;
            const unsigned off = (unsigned)(ks * 32 + hh * 16);
            if (ka < L) {
                a = *(const pipe_u32x4*)(qbase + (size_t)H * D * 2 + (size_t)((unsigned)ka * row_bytes + off));
                b2 = *(const pipe_u32x4*)(qbase + (size_t)H * D * 4 + (size_t)((unsigned)ka * row_bytes + off));
            }
            if (kb < L) {
                c2 = *(const pipe_u32x4*)(qbase + (size_t)H * D * 2 + (size_t)((unsigned)kb * row_bytes + off));
                d2 = *(const pipe_u32x4*)(qbase + (size_t)H * D * 4 + (size_t)((unsigned)kb * row_bytes + off));
            }
            kfa[ks] = *(const x8*)&a; vfa[ks] = *(const x8*)&b2;
            kfb[ks] = *(const x8*)&c2; vfb[ks] = *(const x8*)&d2;
        }
    }
    f32x16 dka[DB], dva[DB], dkb[DB], dvb[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dka[db][e] = 0.f; dva[db][e] = 0.f; dkb[db][e] = 0.f; dvb[db][e] = 0.f; }

    const int aoff = r * ROW + hh * 16;                                        // row fragments of a 32-row block
    const int toff = (4 * hh + q4) * ROW + (16 * (g16 & 1) + 4 * p4) * 2;      // transposed fragments
    const int coff = hh * 16;                                                  // row constants: floats 4 hh + 8 g ..

    // S' = Q K^T - lse / scale and dP' = dO V^T - delta of BOTH key blocks for the 32 query rows at `blk` (rows of a slot)
    auto a_phase = [&](const char* slot, int lq0, f32x16& sA, f32x16& pA, f32x16& sB, f32x16& pB) __attribute__((always_inline)) {
        const char* cs = slot + 2 * C::IMG + lq0 * 4 + coff;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s4 = *(const f32x4*)(cs + g * 32);
            const f32x4 d4 = *(const f32x4*)(cs + CH * 4 + g * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sA[4 * g + i] = s4[i]; sB[4 * g + i] = s4[i]; pA[4 * g + i] = d4[i]; pB[4 * g + i] = d4[i]; }
        }
        const char* qr = slot + lq0 * ROW + aoff;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const x8 qa = *(const x8*)(qr + ks * 32);
            const x8 da = *(const x8*)(qr + C::IMG + ks * 32);
            sA = mfma_32x32x16(qa, kfa[ks], sA);
            sB = mfma_32x32x16(qa, kfb[ks], sB);
            pA = mfma_32x32x16(da, vfa[ks], pA);
            pB = mfma_32x32x16(da, vfb[ks], pB);
        }
    };
    // dV^T += dO^T P and dK^T += Q^T dS of both key blocks for the query rows at lq0
    auto b_phase = [&](const char* slot, int lq0, const x8 (&ppA)[2], const x8 (&dsA)[2], const x8 (&ppB)[2], const x8 (&dsB)[2])
        __attribute__((always_inline)) {
        const char* tr = slot + lq0 * ROW + toff;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* p0 = tr + (16 * s) * ROW + db * 64;
                const x8 qtt = tr_pair<HT>(p0, p0 + 8 * ROW);
                const x8 dot = tr_pair<HT>(p0 + C::IMG, p0 + C::IMG + 8 * ROW);
                dva[db] = mfma_32x32x16(dot, ppA[s], dva[db]);
                dvb[db] = mfma_32x32x16(dot, ppB[s], dvb[db]);
                dka[db] = mfma_32x32x16(qtt, dsA[s], dka[db]);
                dkb[db] = mfma_32x32x16(qtt, dsB[s], dkb[db]);
            }
    };
    // p = exp2(S' c), dS = p dP', both packed for the products above
    auto v_phase = [&](const f32x16& sX, const f32x16& pX, x8 (&pp)[2], x8 (&ds)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p = __builtin_amdgcn_exp2f(sX[8 * u + e] * sl2);
                pp[u][e] = (HT)p;
                ds[u][e] = (HT)(p * pX[8 * u + e]);
            }
    };

    f32x16 sA, pA, sB0, pB0, sB1, pB1;
    x8 ppA0[2], dsA0[2], ppA1[2], dsA1[2], ppB[2], dsB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ppA1[u][e] = (HT)0.f; dsA1[u][e] = (HT)0.f; ppB[u][e] = (HT)0.f; dsB[u][e] = (HT)0.f; }

    __syncthreads();                                              // chunks 0, 1 and the zero slot are in LDS
    if (wave_live) a_phase(smem, 0, sA, pA, sB0, pB0);            // the pipeline's head: block 0

    for (int c = 0; c < nchunks; ++c) {
        Staged nx;
        fetch(nx, c + 2);                                          // in flight under this chunk's regions
        if (wave_live) {
            const char* cur = smem + (c & 3) * C::SLOT;
            const char* prv = smem + ((c + 3) & 3) * C::SLOT;
            const char* nxt = smem + ((c + 1) & 3) * C::SLOT;
            // ---- block 2c
            BWD_FENCE();
            v_phase(sA, pA, ppA0, dsA0);
            b_phase(prv, 32, ppA1, dsA1, ppB, dsB);                // block 2c - 1
            BWD_FENCE();
            v_phase(sB0, pB0, ppB, dsB);
            a_phase(cur, 32, sA, pA, sB1, pB1);                    // block 2c + 1
            BWD_FENCE();
            // ---- block 2c + 1
            v_phase(sA, pA, ppA1, dsA1);
            b_phase(cur, 0, ppA0, dsA0, ppB, dsB);                 // block 2c
            BWD_FENCE();
            v_phase(sB1, pB1, ppB, dsB);
            a_phase(nxt, 0, sA, pA, sB0, pB0);                     // block 2c + 2
            BWD_FENCE();
        }
        stash(nx, smem + ((c + 2) & 3) * C::SLOT);
        __syncthreads();
    }
    // drain: the products of the last block (2 nchunks - 1), whose slot nothing has overwritten
    if (wave_live) b_phase(smem + ((nchunks - 1) & 3) * C::SLOT, 32, ppA1, dsA1, ppB, dsB);

    if (wave_live) {
        float madd[DB][16];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) madd[db][e] = 0.f;
        if (MASK == BF_MASK_QUERY_ROW && n_masked > 0) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 m4 = *(const f32x4*)(msum + db * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) madd[db][4 * g + i] = m4[i];
                }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int key = k0 + 32 * half + r;
            if (key < L) {
                HT* dkp = dqkv + qkv_off(b, key, 1, h, L, H, D);
                HT* dvp = dqkv + qkv_off(b, key, 2, h, L, H, D);
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d0 = db * 32 + 8 * g + 4 * hh;
                        if (d0 < D) {
                            x4 k4, v4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float dk = half ? dkb[db][4 * g + e] : dka[db][4 * g + e];
                                const float dv = half ? dvb[db][4 * g + e] : dva[db][4 * g + e];
                                k4[e] = (HT)(dk * scale);
                                v4[e] = (HT)(dv + madd[db][4 * g + e]);
                            }
                            *(x4*)(dkp + d0) = k4;
                            *(x4*)(dvp + d0) = v4;
                        }
                    }
            }
        }
    }
}

// =====================================================================================
//   ... and dQ with the QUERY on the lane, pipelined the same way; also writes delta = rowsum(dO . O) for the kernel above
// =====================================================================================
// A wave owns TWO 32-query blocks a and b (taken from the list of LIVE blocks under the query-row mask: entirely masked
// blocks receive dQ = 0, written by a zero-fill at the top), K / V walk through a four-slot LDS ring, 64 keys per slot.
//        R1(j):  p = exp2(c S - lse c'), dS = p (dP - delta)  of (a, j)   beside   dQ^T += K_{j-1}^T dS^T   of a AND b
//        R2(j):  the same of (b, j)                                     beside   S = K_{j+1} Q^T,  dP - delta = V_{j+1} dO^T - delta   of a AND b
// The query is on the lane, so -delta is ONE value per lane: it enters as the C operand of dP's first MFMA from a constant
// 16-register vector (no moves in the loop) and -lse rides in the exponent's FMA.  Masked rows inside a live block and rows
// beyond L have Q = dO = 0, delta = 0 and an exponent constant of -1e30: p = 0, dS = 0.  Keys beyond L are zero rows of K:
// whatever dS they get multiplies zeros.
template <int D> struct QPipeCfg {
    static constexpr int DP = (D + 31) / 32 * 32;
    static constexpr int KS = D / 16, DB = DP / 32;
    static constexpr int KROW = DP * 2 + 16;                 // K rows carry DP columns (pad columns zero) for the transposed reads
    static constexpr int VROW = D * 2 + 16;
    static constexpr int CPR = D / 8;
    static constexpr int CH = 64;
    static constexpr int NT = 256;
    static constexpr int PIECES = CH * CPR;
    static constexpr int NPF = (PIECES + NT - 1) / NT;
    static constexpr int KIMG = CH * KROW + 64, VIMG = CH * VROW;
    static constexpr int SLOT = KIMG + VIMG;
    static constexpr int NSLOT = 4;
    static constexpr int MAXBLK = PIPE_MAXBLK;
    static constexpr int LISTS = 2 * MAXBLK * 2;
    static constexpr int LDS = NSLOT * SLOT + LISTS;
};

template <int D, int MASK, typename HT>
__global__ __launch_bounds__(256, 1) void attn_bwd_q_pipe_kernel(
    const HT* __restrict__ qkv, const HT* __restrict__ out, const HT* __restrict__ dout,
    const float* __restrict__ lse, const int32_t* __restrict__ row_mask, HT* __restrict__ dqkv,
    float* __restrict__ delta_ws, int L, int H, float scale) {
    using C = QPipeCfg<D>;
    using x8 = typename H16<HT>::x8;
    using x4 = typename H16<HT>::x4;
    constexpr int KS = C::KS, DB = C::DB, KROW = C::KROW, VROW = C::VROW, CPR = C::CPR, NT = C::NT, NPF = C::NPF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* blist = (unsigned short*)(smem + C::NSLOT * C::SLOT);
    unsigned short* mlist = blist + C::MAXBLK;
    __shared__ int counts[2];
    __shared__ int wave_cnt[2][4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int bh = paired_bh<D>(blockIdx.x, gridDim.x), b = bh / H, h = bh - b * H;
    const int nblk = (L + 31) >> 5;
    const unsigned row_bytes = 3u * (unsigned)H * D * 2u;
    const unsigned orow_bytes = (unsigned)H * D * 2u;
    const char* qbase = (const char*)(qkv + qkv_off(b, 0, 0, h, L, H, D));
    const char* kbase = qbase + (size_t)H * D * 2;
    const char* vbase = qbase + (size_t)H * D * 4;
    const char* obase = (const char*)(out + (size_t)b * L * H * D + (size_t)h * D);
    const char* dobase = (const char*)(dout + (size_t)b * L * H * D + (size_t)h * D);
    char* dqbase = (char*)(dqkv + qkv_off(b, 0, 0, h, L, H, D));
    const int32_t* mask_b = MASK == BF_MASK_QUERY_ROW ? row_mask + (size_t)b * L : nullptr;
    const float sl2 = scale * 1.44269504088896340736f;

    int n_live = nblk, n_masked = 0;
    if (MASK == BF_MASK_QUERY_ROW) {                              // the same lists as the kv kernel builds
        build_block_lists<NT>(mask_b, L, nblk, blist, mlist, counts, wave_cnt);
        n_live = counts[0];
        n_masked = counts[1];
        // dQ of the entirely masked blocks is zero: the workgroups of this (batch, head) share the list
        for (int m = blockIdx.y; m < n_masked; m += gridDim.y) {
            const int ri = threadIdx.x / CPR, ci = threadIdx.x - ri * CPR;
            const int row = (int)mlist[m] * 32 + ri;
            if (ri < 32 && row < L)
                *(pipe_u32x4*)(dqbase + (size_t)((unsigned)row * row_bytes + (unsigned)ci * 16u)) = (pipe_u32x4){0u, 0u, 0u, 0u};
        }
    }
    if ((int)blockIdx.y * 8 >= n_live) return;                    // no live block for this workgroup (uniform)

    const int li0 = (blockIdx.y * 4 + wave) * 2;
    const bool wave_live = li0 < n_live;
    const int nchunks = (L + 63) >> 6;

    // ---- ring staging: chunk c = keys 64 c .. 64 c + 63, K (DP columns, pad zero) and V rows ------------------------
    struct Staged { pipe_u32x4 k[NPF], v[NPF]; };
    auto fetch = [&](Staged& st, int c) __attribute__((always_inline)) {      // raw rows (clamped): zeros are substituted at stash time
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = min((int)threadIdx.x + i * NT, C::PIECES - 1);
            const int jl = idx / CPR, cc = idx - jl * CPR;
            const int rc = min(c * 64 + jl, L - 1);
            st.k[i] = *(const pipe_u32x4*)(kbase + (size_t)((unsigned)rc * row_bytes + (unsigned)cc * 16u));
            st.v[i] = *(const pipe_u32x4*)(vbase + (size_t)((unsigned)rc * row_bytes + (unsigned)cc * 16u));
        }
    };
    auto stash = [&](const Staged& st, char* slot, int c) __attribute__((always_inline)) {
        const pipe_u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int jl = idx / CPR, cc = idx - jl * CPR;
            if ((i + 1) * NT <= C::PIECES || idx < C::PIECES) {
                const bool live = c * 64 + jl < L;                 // keys beyond L: zero rows
                *(pipe_u32x4*)(slot + jl * KROW + cc * 16) = live ? st.k[i] : z;
                *(pipe_u32x4*)(slot + C::KIMG + jl * VROW + cc * 16) = live ? st.v[i] : z;
            }
        }
    };
    {
        Staged s0, s1;
        fetch(s0, 0);
        fetch(s1, 1);
        // every image zero once (pad columns of d = 48; slot 3 is read once as "chunk -1" with dS = 0)
        for (int idx = threadIdx.x; idx < C::NSLOT * C::SLOT / 16; idx += NT)
            *(pipe_u32x4*)(smem + idx * 16) = (pipe_u32x4){0u, 0u, 0u, 0u};
        __syncthreads();
        stash(s0, smem, 0);
        stash(s1, smem + C::SLOT, 1);
    }

    // ---- this wave's queries -------------------------------------------------------------------------------------
    x8 qfa[KS], dofa[KS], qfb[KS], dofb[KS];
    float cla = -1e30f, clb = -1e30f;
    f32x16 dia, dib;                                             // -delta of the lane's query in all 16 elements
    int rowa = -1, rowb = -1;
    {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int li = li0 + half;
            int row = -1;
            if (li < n_live) {
                const int blk = MASK == BF_MASK_QUERY_ROW ? (int)blist[li] : li;
                row = blk * 32 + r;
                if (row >= L) row = -1;
            }
            float delta = 0.f, cl = -1e30f;
            bool on = row >= 0;
            x8 qf[KS], dof[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { qf[ks][e] = (HT)0.f; dof[ks][e] = (HT)0.f; }
            if (row >= 0) {
#pragma unroll
                for (int c = 0; c < D / 8; ++c) {
                    const x8 ov = *(const x8*)(obase + (size_t)((unsigned)row * orow_bytes + (unsigned)c * 16u));
                    const x8 gv = *(const x8*)(dobase + (size_t)((unsigned)row * orow_bytes + (unsigned)c * 16u));
#pragma unroll
                    for (int e = 0; e < 8; ++e) delta = fmaf((float)ov[e], (float)gv[e], delta);
                }
                if (hh == 0) delta_ws[((size_t)b * H + h) * L + row] = delta;
                if (MASK == BF_MASK_QUERY_ROW && mask_b[row] == 0) on = false;     // masked row inside a live block: dQ = 0
                if (on) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        qf[ks] = *(const x8*)(qbase + (size_t)((unsigned)row * row_bytes + (unsigned)(ks * 32 + hh * 16)));
                        dof[ks] = *(const x8*)(dobase + (size_t)((unsigned)row * orow_bytes + (unsigned)(ks * 32 + hh * 16)));
                    }
                    cl = -lse[((size_t)b * H + h) * L + row] * 1.44269504088896340736f;
                }
            }
            const float nd = on ? -delta : 0.f;
            if (half == 0) {
                rowa = row; cla = cl;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { qfa[ks] = qf[ks]; dofa[ks] = dof[ks]; }
#pragma unroll
                for (int e = 0; e < 16; ++e) dia[e] = nd;
            } else {
                rowb = row; clb = cl;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { qfb[ks] = qf[ks]; dofb[ks] = dof[ks]; }
#pragma unroll
                for (int e = 0; e < 16; ++e) dib[e] = nd;
            }
        }
    }
    f32x16 dqa[DB], dqb[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dqa[db][e] = 0.f; dqb[db][e] = 0.f; }

    const int aoff_k = r * KROW + hh * 16, aoff_v = r * VROW + hh * 16;
    const int toff = (4 * hh + q4) * KROW + (16 * (g16 & 1) + 4 * p4) * 2;

    auto a_phase = [&](const char* slot, int jl, f32x16& sA, f32x16& pA, f32x16& sB, f32x16& pB) __attribute__((always_inline)) {
        const char* kr = slot + jl * KROW + aoff_k;
        const char* vr = slot + C::KIMG + jl * VROW + aoff_v;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sA[e] = 0.f; sB[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const x8 kf = *(const x8*)(kr + ks * 32);
            const x8 vf = *(const x8*)(vr + ks * 32);
            sA = mfma_32x32x16(kf, qfa[ks], sA);
            sB = mfma_32x32x16(kf, qfb[ks], sB);
            pA = mfma_32x32x16(vf, dofa[ks], ks == 0 ? dia : pA);
            pB = mfma_32x32x16(vf, dofb[ks], ks == 0 ? dib : pB);
        }
    };
    auto c_phase = [&](const char* slot, int jl, const x8 (&dsA)[2], const x8 (&dsB)[2]) __attribute__((always_inline)) {
        const char* tr = slot + jl * KROW + toff;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* p0 = tr + (16 * s) * KROW + db * 64;
                const x8 kt = tr_pair<HT>(p0, p0 + 8 * KROW);
                dqa[db] = mfma_32x32x16(kt, dsA[s], dqa[db]);
                dqb[db] = mfma_32x32x16(kt, dsB[s], dqb[db]);
            }
    };
    auto v_phase = [&](const f32x16& sX, const f32x16& pX, float cl, x8 (&ds)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sX[8 * u + e], sl2, cl));
                ds[u][e] = (HT)(p * pX[8 * u + e]);
            }
    };

    f32x16 sA, pA, sB0, pB0, sB1, pB1;
    x8 dsA0[2], dsA1[2], dsB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dsA1[u][e] = (HT)0.f; dsB[u][e] = (HT)0.f; }

    __syncthreads();
    if (wave_live) a_phase(smem, 0, sA, pA, sB0, pB0);

    for (int c = 0; c < nchunks; ++c) {
        Staged nx;
        fetch(nx, c + 2);
        if (wave_live) {
            const char* cur = smem + (c & 3) * C::SLOT;
            const char* prv = smem + ((c + 3) & 3) * C::SLOT;
            const char* nxt = smem + ((c + 1) & 3) * C::SLOT;
            BWD_FENCE();
            v_phase(sA, pA, cla, dsA0);
            c_phase(prv, 32, dsA1, dsB);                           // key block 2c - 1
            BWD_FENCE();
            v_phase(sB0, pB0, clb, dsB);
            a_phase(cur, 32, sA, pA, sB1, pB1);                    // key block 2c + 1
            BWD_FENCE();
            v_phase(sA, pA, cla, dsA1);
            c_phase(cur, 0, dsA0, dsB);                            // key block 2c
            BWD_FENCE();
            v_phase(sB1, pB1, clb, dsB);
            a_phase(nxt, 0, sA, pA, sB0, pB0);                     // key block 2c + 2
            BWD_FENCE();
        }
        stash(nx, smem + ((c + 2) & 3) * C::SLOT, c + 2);
        __syncthreads();
    }
    if (wave_live) c_phase(smem + ((nchunks - 1) & 3) * C::SLOT, 32, dsA1, dsB);

    if (wave_live) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = half ? rowb : rowa;
            if (row >= 0) {
                HT* dq = (HT*)(dqbase + (size_t)((unsigned)row * row_bytes));
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d0 = db * 32 + 8 * g + 4 * hh;
                        if (d0 < D) {
                            x4 o4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o4[e] = (HT)((half ? dqb[db][4 * g + e] : dqa[db][4 * g + e]) * scale);
                            *(x4*)(dq + d0) = o4;
                        }
                    }
            }
        }
    }
}

template <int D, int MASK, typename HT>
static void bwd_pipe_launch_one(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* row_mask,
                                void* dqkv, float* delta_ws, int batch, int seq, int heads, float scale, hipStream_t s) {
    static BfPerDevice set_dev;
    size_t& set = set_dev.cur();
    if (set == 0) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_q_pipe_kernel<D, MASK, HT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, QPipeCfg<D>::LDS);
        (void)hipFuncSetAttribute((const void*)attn_bwd_kv_pipe_kernel<D, MASK, HT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, KvPipeCfg<D>::LDS);
        set = 1;
    }
    const int nblk = bf_cdiv(seq, 32);
    hipLaunchKernelGGL((attn_bwd_q_pipe_kernel<D, MASK, HT>), dim3(batch * heads, bf_cdiv(nblk, 8)), dim3(256), QPipeCfg<D>::LDS, s,
                       (const HT*)qkv, (const HT*)out, (const HT*)dout, lse, row_mask, (HT*)dqkv, delta_ws, seq, heads, scale);
    hipLaunchKernelGGL((attn_bwd_kv_pipe_kernel<D, MASK, HT>), dim3(batch * heads, bf_cdiv(seq, 256)), dim3(256), KvPipeCfg<D>::LDS, s,
                       (const HT*)qkv, (const HT*)dout, lse, (const float*)delta_ws, row_mask, (HT*)dqkv, seq, heads, scale);
}

int bf_attn_bwd_pipe_launch(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* row_mask,
                            void* dqkv, float* delta_ws, int batch, int seq, int heads, int head_dim, float scale,
                            int mask_mode, int cdt, hipStream_t s) {
    if (!bf_is_half(cdt) || mask_mode == BF_MASK_CAUSAL || (head_dim != 48 && head_dim != 64)) return 1;
    if (seq < 64 || seq > 32 * KvPipeCfg<64>::MAXBLK) return 1;
    if ((size_t)seq * 3u * heads * head_dim * 2u >= (1ull << 32)) return 1;   // 32-bit row offsets inside one batch element
    const bool f16 = cdt == BF_F16;
#define BPIPE3(DD, MM)                                                                                                   \
    do {                                                                                                                 \
        if (f16) bwd_pipe_launch_one<DD, MM, f16_t>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, scale, s);   \
        else bwd_pipe_launch_one<DD, MM, bf16_t>(qkv, out, dout, lse, row_mask, dqkv, delta_ws, batch, seq, heads, scale, s);      \
    } while (0)
#define BPIPE2(DD) do { if (mask_mode == BF_MASK_NONE) BPIPE3(DD, BF_MASK_NONE); else BPIPE3(DD, BF_MASK_QUERY_ROW); } while (0)
    if (head_dim == 48) BPIPE2(48); else BPIPE2(64);
#undef BPIPE2
#undef BPIPE3
    return BF_OK;
}
