// error.hip — last-error text + ABI version for libglowtts_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace glowtts {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace glowtts

extern "C" const char *glowtts_last_error(void) { return glowtts::g_err; }
extern "C" int glowtts_abi_version(void) { return 1; }
