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
        t.p8_grid = env_int("BIFOLD_P8_GRID", 0);
        t.attn_pipe = env_int("BIFOLD_ATTN_PIPE", 1) ? 1 : 0;
        t.attn_bwd_fused_p = env_int("BIFOLD_ATTN_BWD_FUSED_P", 1);
    });
    return t;
}

extern "C" int bf_abi_version(void) { return BF_ABI_VERSION; }
extern "C" const char* bf_last_error(void) { return g_err; }
extern "C" int bf_last_hip_error(void) { return g_hip_err; }
// what a binding's mirror of a struct must measure (tests/test_host_logic.py compares hip/lib.py's ctypes structures)
extern "C" size_t bf_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(bf_gemm_args);
        case 1: return sizeof(bf_gemm_tn_problem);
        case 2: return sizeof(bf_lora_slots);
        default: return 0;
    }
}

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
