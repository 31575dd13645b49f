// Error reporting and version entry points of the C ABI (include/vda.h).
#include "vda_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void vda_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vda_last_error(void) { return g_err; }
extern "C" int vda_abi_version(void) { return 4; }
