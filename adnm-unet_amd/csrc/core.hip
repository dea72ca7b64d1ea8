// Error reporting and ABI version for libadnm_hip.
#include "adnm_common.h"

static thread_local char g_err[512] = "";

void adnm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adnm_last_error(void) { return g_err; }
extern "C" int adnm_abi_version(void) { return 1; }
