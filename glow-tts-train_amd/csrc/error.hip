// error.hip — last-error text + ABI version for libglowtts_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include "common.hpp"

namespace glowtts {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int env_knob(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}
}  // namespace glowtts

extern "C" const char *glowtts_last_error(void) { return glowtts::g_err; }
extern "C" int glowtts_abi_version(void) { return 1; }
