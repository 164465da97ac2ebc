// Error plumbing and device probe of libsdtrain_hip.so (no global mutable state besides the thread-local message).
#include <stdarg.h>
#include <stdio.h>

#include "sdt_common.h"

static thread_local char g_err[512] = "";

void sdt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* sdt_last_error(void) { return g_err; }
int sdt_abi_version(void) { return 1; }
int sdt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
