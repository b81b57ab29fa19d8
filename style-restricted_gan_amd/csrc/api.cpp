// Error plumbing + ABI version of libsrgan_hip.so.
#include <cstdarg>
#include <cstdio>
#include "srgan_hip.h"

namespace srgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace srgan

extern "C" int srgan_abi_version(void) { return SRGAN_ABI_VERSION; }
extern "C" const char* srgan_last_error(void) { return srgan::g_err; }
