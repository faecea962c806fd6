// Every tuning knob of libbifold_hip in ONE place.  The knobs select kernels / geometries only (results are the same up
// to summation order); they are read from the environment exactly once per process, on first use, under std::call_once
// (thread-safe), and never change afterwards -- except `gemm_pipe`, which bf_gemm_set_variant may set at run time and is
// an atomic.  Defaults are the values the in-step A/B measurements of DESIGN.md §6 settled on.
#pragma once
#include <atomic>

struct BfTuning {
    // ---- GEMM (gemm.hip)
    std::atomic<int> gemm_pipe;   // BIFOLD_GEMM_PIPE        -1 automatic; 0 = 128x128x64, 4 = 128x128x32, 8 = persistent 256x256
    int gemm_group_m;             // BIFOLD_GEMM_GROUP_M     row tiles per group of the 128x128 kernels' tile order (16)
    int gemm_no_dma;              // BIFOLD_GEMM_NO_DMA      1: register-staged fallback kernel only (debug)
    int gemm_bm;                  // BIFOLD_GEMM_BM          64: 64-row tiles in the BK = 32 kernel (opt-in), -1 off
    int p8_min_m;                 // BIFOLD_P8_MIN_M         persistent kernel from this many rows (16384)
    int p8_min_tiles;             // BIFOLD_P8_MIN_TILES     ... and this many 256x256 tiles (-1: half the CUs)
    int p8_group_m;               // BIFOLD_P8_GROUP_M       row tiles per group of the persistent kernel's item order (5)
    int p8_grid;                  // BIFOLD_P8_GRID          cap on the persistent grid (0: every CU)
    int p8_revk;                  // BIFOLD_P8_REVK          alternate the K direction per round of tickets (1)
    int p8_bk64;                  // BIFOLD_P8_BK64          BK = 64 full-line main loop for the persistent NT GEMMs (1)
    int p8_bk64_nn;               // BIFOLD_P8_BK64_NN       ... and for the NN GEMMs (dX = dY W: K-strided B read transposed) (1)
    int p8_deep;                  // BIFOLD_P8_DEEP          prefetch distance 3: -1 = for TN only, 0 / 1 = never / always
    int p8_nt;                    // BIFOLD_P8_NT            bit mask of non-temporal output stores (5)
    int p8_tail_mode;             // BIFOLD_P8_TAIL_MODE     under-filled last round: 2 = K chunks inside the launch (BK = 64 kernels), 1 = round 2's separate split-K launches, 0 = whole tiles (2)
    int p8_tail_min_kt;           // BIFOLD_P8_TAIL_MIN_KT   ... a K chunk of that split is at least this many K tiles of 64 deep (12)
    int p8_skew_phases, p8_skew_us10;   // BIFOLD_P8_SKEW     "phases,tenths of a microsecond per phase": start skew of the write-heavy persistent GEMMs (off)
    int p8_no_tail;               // BIFOLD_P8_NO_TAIL       1: never cut an under-filled last round off as a split-K tail
    int p8_tail_k_nt, p8_tail_k_nn;   // BIFOLD_P8_TAIL_K   "nt,nn": K thresholds of that tail split (2048, off)
    int fp8_p8;                   // BIFOLD_FP8_P8           persistent e4m3 kernel (1)
    int dw_grouped;               // BIFOLD_DW_GROUPED       grouped weight-gradient launch (1)
    // ---- attention (attention.hip)
    int attn_fwd_tiled;           // BIFOLD_ATTN_FWD_TILED   1: always the tiled forward (0)
    int attn_fwd_pipe;            // BIFOLD_ATTN_FWD_PIPE    software-pipelined forward for sequences beyond LDS, d = 48 / 64, no causal mask (attention_pipe.hip) (1)
    int attn_fwd_pipe_min_seq;    // BIFOLD_ATTN_FWD_PIPE_MIN_SEQ   ... also for sequences that fit LDS, from this length (1 << 30: never)
    int attn_fwd_pd;              // BIFOLD_ATTN_FWD_PD      persistent double-buffered d = 32 forward (0: measured no faster than the resident kernel)
    int attn_nt;                  // BIFOLD_ATTN_NT          non-temporal QKV loads in the resident forward (0: they paid while the next GEMM's epilogue re-read an fp32 residual tile -- round 3 -- and cost 0.7 % of the step since it does not, round 4)
    int attn_fwd_kt32;            // BIFOLD_ATTN_FWD_KT32    0 / 1 / 2: 64-key tiles / 32-key steps / by the last tile's fill (2)
    int attn_bwd_nw;              // BIFOLD_ATTN_BWD_NW      cap on the waves of the single backward kernel (0: none)
    int attn_bwd_single;          // BIFOLD_ATTN_BWD_SINGLE  1: never the split pair (0)
    int attn_bwd_resident_wide;   // BIFOLD_ATTN_BWD_RESIDENT_WIDE   1: d = 64 on the resident pair too, -1: d = 32 only (0)
    int attn_stream_min_q32;      // BIFOLD_ATTN_STREAM_MIN_Q32      streamed pair from this many 32-row blocks (1)
    int attn_stream_d32;          // BIFOLD_ATTN_STREAM_D32  1: d = 32 on the streamed pair (0)
    int attn_bwd_wpb;             // BIFOLD_ATTN_BWD_WPB     waves per workgroup of the resident pair's q kernel (3: 275 tokens = 9 blocks, three per wave)
    int attn_bwd_wpb_kv;          // BIFOLD_ATTN_BWD_WPB_KV  ... of its kv kernel (4)
    int attn_bwd_fused;           // BIFOLD_ATTN_BWD_FUSED   one-pass backward for d = 32 without a mask, <= 288 tokens (1)
    int attn_bwd_fused_p;         // BIFOLD_ATTN_BWD_FUSED_P 0 / 1 / 2: the persistent twelve-wave form of that kernel never / from two items per CU / always (1)
    int attn_bwd_fused_p_map;     // BIFOLD_ATTN_BWD_FUSED_P_MAP   1: that kernel's three support waves (dQ sum, two loaders) share ONE SIMD, three key waves on each of the others; 0: roles in wave order (1)
    int attn_bwd_onepass;         // BIFOLD_ATTN_BWD_ONEPASS 1: the one-pass backward for d = 48 / 64 without a causal mask (0: correct, and 10 - 20 % SLOWER than the streamed pair on every shape measured -- attention.hip)
    int attn_onepass_min_q32;     // BIFOLD_ATTN_ONEPASS_MIN_Q32   ... from this many 32-row blocks (24: at 576 x 64 the streamed pair is faster, 364 vs 523 us: three-workgroup grids with idle waves)
    int attn_lds_pad;             // BIFOLD_ATTN_LDS_PAD     extra dynamic LDS bytes for the resident pair: occupancy experiments (0)
    int attn_bwd_split;           // BIFOLD_ATTN_BWD_SPLIT   workgroups per (batch, head) of the resident pair (1)
    int attn_kv_tight;            // BIFOLD_ATTN_KV_TIGHT    128-register build of the d = 32 kv kernel (1)
    int attn_bwd_stream;          // BIFOLD_ATTN_BWD_STREAM  streamed pair for long sequences (1)
    int attn_bwd_pipe;            // BIFOLD_ATTN_BWD_PIPE    software-pipelined pair for d = 48 / 64 without a causal mask, in place of the streamed pair (attention_pipe.hip) (1)
    int attn_bwd_pipe_min_q32;    // BIFOLD_ATTN_BWD_PIPE_MIN_Q32   ... from this many 32-row blocks (24: at 576 x 64 the streamed pair is faster, 364 vs 523 us: three-workgroup grids with idle waves)
    // ---- LayerNorm (norm_elementwise.hip)
    int ln_rows;                  // BIFOLD_LN_ROWS          rows per wave of the exact-width forward (2)
    int ln_bwd16_wide;            // BIFOLD_LN_BWD16_WIDE    16-bit gradient stream: eight columns per lane in the LayerNorm backward (1)
};

BfTuning& bf_tuning();
