// Error reporting for the C-ABI (see include/biomedkg_hip.h).
#include "common.h"

#include <cstdio>
#include <cstring>

namespace bkg {
static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}
void set_error_msg(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }
}  // namespace bkg

extern "C" const char* bkg_last_error() { return bkg::g_err; }
extern "C" int bkg_abi_version() { return 1; }
