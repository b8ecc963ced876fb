// Library-level C ABI: version, target, thread-local error string.
#include "llmie_common.h"

#include <cstring>

namespace llmie {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace llmie

extern "C" int llmie_abi_version(void) { return LLMIE_ABI_VERSION; }
extern "C" const char *llmie_last_error(void) { return llmie::g_err; }
extern "C" const char *llmie_target_arch(void) { return "gfx950"; }
