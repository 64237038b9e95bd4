// Error plumbing shared by every entry point of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pssr_mi355.h"

static thread_local char g_err[512] = "";

void pssr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pssr_last_error(void) { return g_err; }
extern "C" int pssr_abi_version(void) { return 1; }
