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

// ---- kernel-name notes (see common.h) ----
#include <cxxabi.h>
#include <stdlib.h>
static int g_names_on = 0;
static thread_local char g_kname[256] = "";
static thread_local const void* g_kcache_ptr[64];
static thread_local char g_kcache_name[64][256];
static thread_local int g_kcache_n = 0;
extern "C" void maai_note_kernel(const void* fn) {
  if (!g_names_on) return;
  for (int i = 0; i < g_kcache_n; ++i)
    if (g_kcache_ptr[i] == fn) {
      strcpy(g_kname, g_kcache_name[i]);
      return;
    }
  g_kname[0] = 0;
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  (void)hipGetLastError();
  if (!mangled) return;
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  const char* b = (status == 0 && dem) ? dem : mangled;
  if (strncmp(b, "void ", 5) == 0) b += 5;
  // drop the argument list: the last '(' at template depth 0
  size_t n = strlen(b), cut = n;
  int depth = 0;
  for (size_t i = 0; i < n; ++i) {
    if (b[i] == '<') ++depth;
    else if (b[i] == '>') --depth;
    else if (b[i] == '(' && depth == 0) { cut = i; break; }
  }
  if (cut > sizeof(g_kname) - 1) cut = sizeof(g_kname) - 1;
  memcpy(g_kname, b, cut);
  g_kname[cut] = 0;
  if (dem) free(dem);
  if (g_kcache_n < 64) {
    g_kcache_ptr[g_kcache_n] = fn;
    strcpy(g_kcache_name[g_kcache_n], g_kname);
    ++g_kcache_n;
  }
}
extern "C" int maai_kernel_names(int on) {
  const int was = g_names_on;
  g_names_on = on ? 1 : 0;
  g_kname[0] = 0;
  return was;
}
extern "C" const char* maai_last_kernel_name(void) { return g_kname; }
