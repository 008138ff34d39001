// Error plumbing + misc entry points of the C ABI (include/sr_hip.h).
#include "common.h"
#include <string>

static thread_local std::string g_last_error;

void sr_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char* sr_last_error(void) { return g_last_error.c_str(); }
extern "C" int sr_version(void) { return 1; }
extern "C" int sr_max_topk(void) { return SR_MAX_TOPK; }
