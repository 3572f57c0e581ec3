// Error reporting + version of the C-ABI (include/graspbal.h).
#include <stdio.h>
#include <string.h>

#include "gb_common.h"

namespace gb {
static thread_local char g_err[256] = "";

void set_last_error(const char *what, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
}
void clear_last_error() { g_err[0] = 0; }
}  // namespace gb

extern "C" int gb_abi_version(void) { return GB_ABI_VERSION; }
extern "C" const char *gb_last_error(void) { return gb::g_err; }
