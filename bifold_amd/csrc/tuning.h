// Every knob of libbifold_hip in ONE place -- and since round 6 there are four.  Rounds 1 - 5 grew 45 BIFOLD_* variables
// here, each the A/B switch of a kernel variant or a geometry; every one of them was decided by a measurement on the GPU
// (docs/DESIGN_round*.md keep the numbers), and a variable whose other value lost is not a feature
// but an untested code path.  What lost was removed from the library together with its variable; what won is a constant
// below, with the measurement it came from.  The four that remain select behaviour a user or a test has a reason to
// choose; they are read from the environment exactly once per process, on first use, under std::call_once (thread-safe)
// -- except `gemm_pipe`, which bf_gemm_set_variant may set at run time and is an atomic.  Results do not depend on them
// (up to summation order).  The GPU suite runs the config A fixture under every non-default value
// (tests/test_model_gpu.py::test_tiny_fixture_under_every_library_knob).
#pragma once
#include <atomic>

struct BfTuning {
    // ---- the knobs
    std::atomic<int> gemm_pipe;   // BIFOLD_GEMM_PIPE          -1 automatic; 0 = 128x128x64, 4 = 128x128x32, 8 / 9 = persistent 256x256 (BK 32 / BK 64): forces a GEMM family (tests: the `gemm_variant` fixture through bf_gemm_set_variant)
    int p8_grid;                  // BIFOLD_P8_GRID            cap on the persistent GEMMs' grid (0: every CU).  bifold/dist.py lowers it by BIFOLD_P8_RESERVE_CUS while a process group is alive, so that RCCL's kernels find CUs (tests/test_dist_cpu.py)
    int attn_pipe;                // BIFOLD_ATTN_PIPE          1: the software-pipelined attention kernels of round 6 for sequences beyond LDS (attention_pipe.hip); 0: the tiled forward / streamed backward pair of rounds 1 - 5 (kept for causal long sequences and d = 32, so always built and tested)
    int attn_bwd_fused_p;         // BIFOLD_ATTN_BWD_FUSED_P   0 / 1 / 2: the persistent twelve-wave one-pass backward (d = 32, nine blocks) never / from two items per CU / always (tests force 2 to reach it with few items)

    // ---- settled by measurement: constants (the former BIFOLD_* variable in brackets)
    static constexpr int gemm_group_m = 16;      // [GEMM_GROUP_M] row tiles per group of the 128x128 kernels' tile order
    static constexpr int p8_min_m = 16384;       // [P8_MIN_M] persistent kernel from this many rows ...
    static constexpr int p8_min_tiles = -1;      // [P8_MIN_TILES] ... and this many 256x256 tiles (-1: one per two CUs)
    static constexpr int p8_group_m = 5;         // [P8_GROUP_M] row tiles per group of the persistent kernel's item order: 5 -> 32.22 ms, 6 -> 32.26, 3 -> 32.29, 7 -> 32.38, 4 -> 32.42 (round 4, one box)
    static constexpr int p8_revk = 1;            // [P8_REVK] alternate the K direction per round of tickets
    static constexpr int p8_bk64 = 1;            // [P8_BK64] BK = 64 whole-line main loop for the persistent NT GEMMs: 7 234 / 7 255 -> 7 423 / 7 386 samples/s (round 4)
    static constexpr int p8_bk64_nn = 1;         // [P8_BK64_NN] ... and the NN GEMMs (dX = dY W: K-strided B read transposed)
    static constexpr int p8_deep = -1;           // [P8_DEEP] prefetch distance 3 for TN only
    static constexpr int p8_nt = 5;              // [P8_NT] non-temporal output stores: plain 16-bit and GELU-grad outputs (round 2: 1 -> 5 +0.3 %)
    static constexpr int p8_tail_min_kt = 12;    // [P8_TAIL_MIN_KT] an under-filled last round is cut into K chunks of at least this many 64-deep K tiles (K = 3072 in 3 chunks -5 %, K = 512 / 768 +12 ... 19 %)
    static constexpr int fp8_p8 = 1;             // [FP8_P8] persistent e4m3 kernel
    static constexpr int dw_grouped = 1;         // [DW_GROUPED] the weight-gradient GEMMs of a layer in one grouped launch
    static constexpr int ln_rows = 2;            // [LN_ROWS] rows per wave of the exact-width LayerNorm forward (5.4 - 5.7 vs 4.5 - 4.7 TB/s)
    static constexpr int ln_bwd16_wide = 1;      // [LN_BWD16_WIDE] 16-bit gradient stream: eight columns per lane in the LayerNorm backward
};

BfTuning& bf_tuning();
