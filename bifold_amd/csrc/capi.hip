// Error plumbing and ABI version for libbifold_hip.
#include "common.h"

#include <string.h>

static thread_local char g_err[512] = "";
static thread_local int g_hip_err = 0;

void bf_set_error(int hip_code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_hip_err = hip_code;
}

extern "C" int bf_abi_version(void) { return BF_ABI_VERSION; }
extern "C" const char* bf_last_error(void) { return g_err; }
extern "C" int bf_last_hip_error(void) { return g_hip_err; }
