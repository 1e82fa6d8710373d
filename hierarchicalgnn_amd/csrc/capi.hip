// Error reporting and version entry points of the C ABI (include/hgnn_hip.h).
#include "common.h"
#include <cstring>

namespace hgnn {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace hgnn

extern "C" int hgnn_abi_version(void) { return HGNN_ABI_VERSION; }
extern "C" const char* hgnn_last_error(void) { return hgnn::g_err; }
extern "C" int hgnn_sizeof_plan(void) { return (int)sizeof(hgnn_plan); }
extern "C" int hgnn_sizeof_mlp_desc(void) { return (int)sizeof(hgnn_mlp_desc); }
