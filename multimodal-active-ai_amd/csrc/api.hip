// Error reporting and library identity for the C ABI (include/maai_hip.h).
#include "common.h"
#include "maai_internal.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void maai_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* maai_last_error(void) { return g_err; }
extern "C" int maai_abi_version(void) { return MAAI_ABI_VERSION; }
extern "C" int maai_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}
