// Thread-local last-error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ase_hip.h"

static thread_local char g_err[512] = "";

void ase_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ase_hip_last_error(void) { return g_err; }
extern "C" int ase_hip_abi_version(void) { return ASE_HIP_ABI_VERSION; }
