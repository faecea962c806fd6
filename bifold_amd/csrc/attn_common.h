// Helpers shared by attention.hip (resident / one-pass kernels) and attention_pipe.hip (software-pipelined kernels for
// sequences beyond LDS): indexing of the packed QKV buffer [B, L, 3, H, D], the XCD-aware (batch, head) order, the
// transposing fragment read.
#pragma once
#include "common.h"

// Workgroup -> (batch, head).  Consecutive workgroup ids go to consecutive XCDs (8 private L2s).  With
// d = 32 a head's slice of a token row is 64 bytes, half a 128-byte line: the other half belongs to the
// neighbouring head, whose workgroup runs on ANOTHER XCD and fetches the same line from HBM again
// (PMC: 432 MB read per forward launch for 216 MB of Q / K / V, 885 - 1 015 MB per backward kernel for
// 360).  Pairing heads 2p and 2p+1 on one XCD, in adjacent dispatch slots, lets the second hit in L2.
// d = 48: four heads span exactly three lines (384 bytes), so groups of four.  d = 64: a head is a line.
template <int D>
__device__ __forceinline__ int paired_bh(int i, int n) {
    constexpr int G = D == 32 ? 2 : (D == 48 ? 4 : 1);
    if (G == 1 || (n % (8 * G))) return i;
    const int xcd = i & 7, s = i >> 3, t = s % G;
    return ((s / G) * 8 + xcd) * G + t;
}

__device__ __forceinline__ size_t qkv_off(int b, int i, int which, int h, int L, int H, int D) {
    return (((size_t)b * L + i) * 3 + which) * (size_t)H * D + (size_t)h * D;
}

// two ds_read_b64_tr_b16 -> one 8-element MFMA fragment (rows r0 .. r0 + 3 and r0 + 8 .. r0 + 11 of a row-major image)
template <typename HT>
__device__ __forceinline__ typename H16<HT>::x8 tr_pair(const char* p0, const char* p1) {
    const typename H16<HT>::x4 lo = lds_read_tr16<HT>(p0);
    const typename H16<HT>::x4 hi = lds_read_tr16<HT>(p1);
    typename H16<HT>::x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// attention_pipe.hip: launchers called by bf_attention_fwd / bf_attention_bwd (attention.hip) after their argument checks.
// Each returns BF_OK, or BF_EUNSUPPORTED-like 1 when the shape is not one it takes (the caller then uses its own kernels).
int bf_attn_fwd_pipe_launch(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq, int heads,
                            int head_dim, float scale, int mask_mode, int cdt, hipStream_t s);
int bf_attn_bwd_pipe_launch(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* row_mask,
                            void* dqkv, float* delta_ws, int batch, int seq, int heads, int head_dim, float scale,
                            int mask_mode, int cdt, hipStream_t s);
