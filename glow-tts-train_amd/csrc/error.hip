// error.hip — last-error text + ABI version for libglowtts_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.hpp"

namespace glowtts {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// name (GLOWTTS_ + this) and default of every tuning switch, in the order of `enum Knob` (common.hpp); -1 = "unset / automatic"
struct KnobDef { const char *name; int dflt; };
static const KnobDef kKnobs[K_COUNT] = {
    {"CONV_ROW_ADJ", 1}, {"CONV32_1X1", 1}, {"WRW1_PIPE", 1}, {"WRW1_MULTI", 1}, {"WRW1_CUS", -1}, {"WRW1_XCD", 1}, {"WN_FUSED", 0},
    {"WRW_BATCH", 1}, {"WRW5_BSPLIT", 1}, {"WRW_TR", -1}, {"WRW_TR_MT", 4}, {"WRW_TR_NG", 2}, {"WRW_TR_NG_SPLITS", 1},
    {"WRW_TR_PRIO", 2}, {"WRW_TR3", 1}, {"WRW_TR3_MT", 2}, {"MAS_WAVES", 1}, {"WRW5_CUS", -1}, {"WINO", 1},
#ifdef GLOWTTS_TRACE
    {"BND_EXP", 0}, {"WRW1_EXP", 0}, {"CONV_EXP", 0},
#endif
};
static std::atomic<int> g_knob[K_COUNT];
static std::once_flag g_knob_once;

static void knobs_from_environment() {
    for (int k = 0; k < K_COUNT; ++k) {
        char name[64];
        snprintf(name, sizeof(name), "GLOWTTS_%s", kKnobs[k].name);
        const char *e = getenv(name);
        g_knob[k].store((e && e[0]) ? atoi(e) : kKnobs[k].dflt, std::memory_order_relaxed);
    }
}

int knob(Knob k) {
    std::call_once(g_knob_once, knobs_from_environment);
    return g_knob[k].load(std::memory_order_relaxed);
}

static int knob_index(const char *name) {
    if (!name) return -1;
    if (strncmp(name, "GLOWTTS_", 8) == 0) name += 8;
    for (int k = 0; k < K_COUNT; ++k)
        if (strcmp(name, kKnobs[k].name) == 0) return k;
    return -1;
}
}  // namespace glowtts

extern "C" int glowtts_set_knob(const char *name, int value) {
    const int k = glowtts::knob_index(name);
    GLOWTTS_CHECK_ARG(k >= 0, "glowtts_set_knob: no tuning switch named %s (include/glowtts_hip.h lists them)", name ? name : "(null)");
    (void)glowtts::knob((glowtts::Knob)k);              // the environment is read first, so that it cannot overwrite this value later
    glowtts::g_knob[k].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int glowtts_get_knob(const char *name, int *value) {
    const int k = glowtts::knob_index(name);
    GLOWTTS_CHECK_ARG(k >= 0 && value, "glowtts_get_knob: no tuning switch named %s (include/glowtts_hip.h lists them)", name ? name : "(null)");
    *value = glowtts::knob((glowtts::Knob)k);
    return 0;
}

extern "C" const char *glowtts_last_error(void) { return glowtts::g_err; }
extern "C" int glowtts_abi_version(void) { return 1; }
