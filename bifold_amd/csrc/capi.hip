// Error plumbing and ABI version for libbifold_hip.
#include "common.h"

#include "tuning.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

static thread_local char g_err[512] = "";
static thread_local int g_hip_err = 0;

void bf_set_error(int hip_code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_hip_err = hip_code;
}

// ---- tuning knobs: one struct, read once (tuning.h) -----------------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
BfTuning& bf_tuning() {
    static BfTuning t;
    static std::once_flag once;
    std::call_once(once, []() {
        t.gemm_pipe.store(env_int("BIFOLD_GEMM_PIPE", -1));
        t.gemm_group_m = env_int("BIFOLD_GEMM_GROUP_M", 16); if (t.gemm_group_m < 1) t.gemm_group_m = 1;
        t.gemm_no_dma = getenv("BIFOLD_GEMM_NO_DMA") ? 1 : 0;
        t.gemm_bm = env_int("BIFOLD_GEMM_BM", -1);
        t.p8_min_m = env_int("BIFOLD_P8_MIN_M", 16384);
        t.p8_min_tiles = env_int("BIFOLD_P8_MIN_TILES", -1);
        // in-step A/B: round 2 (BK 32 loop): 4 / 2 +0.5 %, 16 -1.7 % against 8; round 4 (BK 64 loop, three runs each on one
        // box): 5 -> 32.22 ms, 6 -> 32.26, 3 -> 32.29, 7 -> 32.38, 4 -> 32.42 (8 and 1: 33.1 against 32.9 for 4 on another box):
        // five A panels of 256 rows are 2 - 2.4 MB of a 4 MB L2, and an odd count does not stride the panels by a power of two
        t.p8_group_m = env_int("BIFOLD_P8_GROUP_M", 5); if (t.p8_group_m < 1) t.p8_group_m = 1;
        t.p8_grid = env_int("BIFOLD_P8_GRID", 0);
        t.p8_revk = env_int("BIFOLD_P8_REVK", 1) ? 1 : 0;
        t.p8_bk64 = env_int("BIFOLD_P8_BK64", 1);         // round 4 same-box A/B of the step: 7 234 / 7 255 -> 7 423 / 7 386 samples/s
        t.p8_bk64_nn = env_int("BIFOLD_P8_BK64_NN", 1);
        t.p8_deep = env_int("BIFOLD_P8_DEEP", -1);
        t.p8_nt = env_int("BIFOLD_P8_NT", 5) & 7;          // plain-16-bit and GELU-grad outputs non-temporal (round 2: 1 -> 5 +0.3 %)
        t.p8_tail_mode = env_int("BIFOLD_P8_TAIL_MODE", 2);
        t.p8_tail_min_kt = env_int("BIFOLD_P8_TAIL_MIN_KT", 12); if (t.p8_tail_min_kt < 1) t.p8_tail_min_kt = 1;
        t.p8_skew_phases = 0; t.p8_skew_us10 = 0;
        if (const char* e = getenv("BIFOLD_P8_SKEW")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d,%d", &a, &b) == 2 && a > 1 && a < 16 && b > 0 && b < 2000) { t.p8_skew_phases = a; t.p8_skew_us10 = b; }
        }
        t.p8_no_tail = getenv("BIFOLD_P8_NO_TAIL") ? 1 : 0;
        t.p8_tail_k_nt = 2048; t.p8_tail_k_nn = 1 << 30;   // in-step A/B (round 2): NN tails at K >= 1536 / 512: +-0.1 %; NT at K >= 768: -2 %
        if (const char* e = getenv("BIFOLD_P8_TAIL_K")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { t.p8_tail_k_nt = a; t.p8_tail_k_nn = b; }
        }
        t.fp8_p8 = env_int("BIFOLD_FP8_P8", 1);
        t.dw_grouped = env_int("BIFOLD_DW_GROUPED", 1);
        t.attn_fwd_tiled = env_int("BIFOLD_ATTN_FWD_TILED", 0) ? 1 : 0;
        t.attn_fwd_pd = env_int("BIFOLD_ATTN_FWD_PD", 0);
        t.attn_fwd_pipe = env_int("BIFOLD_ATTN_FWD_PIPE", 1);
        t.attn_fwd_pipe_min_seq = env_int("BIFOLD_ATTN_FWD_PIPE_MIN_SEQ", 1 << 30);
        t.attn_nt = env_int("BIFOLD_ATTN_NT", 0);
        t.attn_fwd_kt32 = env_int("BIFOLD_ATTN_FWD_KT32", 2);
        t.attn_bwd_nw = env_int("BIFOLD_ATTN_BWD_NW", 0);
        t.attn_bwd_single = env_int("BIFOLD_ATTN_BWD_SINGLE", 0) ? 1 : 0;
        t.attn_bwd_resident_wide = env_int("BIFOLD_ATTN_BWD_RESIDENT_WIDE", 0);
        t.attn_stream_min_q32 = env_int("BIFOLD_ATTN_STREAM_MIN_Q32", 1);
        t.attn_stream_d32 = env_int("BIFOLD_ATTN_STREAM_D32", 0);
        t.attn_bwd_wpb = env_int("BIFOLD_ATTN_BWD_WPB", 3); if (t.attn_bwd_wpb < 1) t.attn_bwd_wpb = 1;
        t.attn_bwd_wpb_kv = env_int("BIFOLD_ATTN_BWD_WPB_KV", 4); if (t.attn_bwd_wpb_kv < 1) t.attn_bwd_wpb_kv = 1;
        t.attn_bwd_fused = env_int("BIFOLD_ATTN_BWD_FUSED", 1);
        t.attn_bwd_fused_p = env_int("BIFOLD_ATTN_BWD_FUSED_P", 1);
        t.attn_bwd_fused_p_map = env_int("BIFOLD_ATTN_BWD_FUSED_P_MAP", 1);
        t.attn_bwd_onepass = env_int("BIFOLD_ATTN_BWD_ONEPASS", 0);
        t.attn_onepass_min_q32 = env_int("BIFOLD_ATTN_ONEPASS_MIN_Q32", 17);
        t.attn_lds_pad = env_int("BIFOLD_ATTN_LDS_PAD", 0); if (t.attn_lds_pad < 0) t.attn_lds_pad = 0;
        t.attn_bwd_split = env_int("BIFOLD_ATTN_BWD_SPLIT", 1); if (t.attn_bwd_split < 1) t.attn_bwd_split = 1;
        t.attn_kv_tight = env_int("BIFOLD_ATTN_KV_TIGHT", 1);
        t.attn_bwd_stream = env_int("BIFOLD_ATTN_BWD_STREAM", 1);
        t.attn_bwd_pipe = env_int("BIFOLD_ATTN_BWD_PIPE", 1);
        t.attn_bwd_pipe_min_q32 = env_int("BIFOLD_ATTN_BWD_PIPE_MIN_Q32", 24);
        t.ln_rows = env_int("BIFOLD_LN_ROWS", 2);
        t.ln_bwd16_wide = env_int("BIFOLD_LN_BWD16_WIDE", 1);
    });
    return t;
}

extern "C" int bf_abi_version(void) { return BF_ABI_VERSION; }
extern "C" const char* bf_last_error(void) { return g_err; }
extern "C" int bf_last_hip_error(void) { return g_hip_err; }

// Test hook: `blocks` workgroups that each hold a whole CU's worth of LDS (a stand-in for a
// collective's long-running kernels) and spin for `cycles` shader clocks, then exit.  Used by
// tools/gemm_disturb.py to measure how the persistent GEMM behaves when some CUs are taken.
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, int* sink) {
    extern __shared__ char hold[];
    hold[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(16);
    if (sink && hold[(threadIdx.x + 1) & 255] == 77 && cycles < 0) sink[0] = 1;
}

extern "C" int bf_debug_spin(int blocks, long long cycles, bf_stream_t stream) {
    BF_REQUIRE(blocks > 0 && blocks <= 256 && cycles >= 0 && cycles < (1LL << 34), BF_EINVAL, "bf_debug_spin: bad arguments");
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 150 * 1024, (hipStream_t)stream, cycles, (int*)nullptr);
    BF_CHECK_LAUNCH("bf_debug_spin");
    return BF_OK;
}
