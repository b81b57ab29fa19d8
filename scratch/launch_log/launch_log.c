/* Measurement tooling (GPU box): LD_PRELOAD pass-through logger of every kernel launch of the process -- ours and ATen's.
 * Writes "<seq> <mangled name> gx gy gz bx by bz dynLDS stream" to $SRGAN_LAUNCH_LOG, flushed per line, then forwards to the
 * real runtime.  With AMD_SERIALIZE_KERNEL=3 the last line of the log is the dispatch that was executing when a queue aborts
 * (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT under rocprofv3 --pmc).  Also checks every descriptor against the
 * AQL limits itself and prints "BAD" lines to stderr.
 *   gcc -shared -fPIC -O1 -o launch_log.so launch_log.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned x, y, z; } dim3_t;
typedef int (*launch_fn)(const void*, dim3_t, dim3_t, void**, size_t, void*);
typedef void (*reg_fn)(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*);

/* libamdhip64 arrives as a dependency of a dlopen(RTLD_LOCAL)-ed Python extension: RTLD_NEXT does not see it.  Find the copy the
 * process has mapped. */
static void* real_sym(const char* name) {
  void* f = dlvsym(RTLD_NEXT, name, "hip_4.2");
  if (!f) f = dlsym(RTLD_NEXT, name);
  static const char* const libs[] = {"libamdhip64.so.7", "libamdhip64.so", "libamdhip64.so.6"};
  for (unsigned i = 0; !f && i < sizeof(libs) / sizeof(libs[0]); ++i) {
    void* h = dlopen(libs[i], RTLD_NOW | RTLD_NOLOAD);
    if (h) f = dlsym(h, name);
  }
  return f;
}

static const void** g_host = NULL;
static const char** g_name = NULL;
static int g_n = 0, g_cap = 0;
static FILE* g_log = NULL;
static unsigned long g_seq = 0;

void __hipRegisterFunction(void** modules, const void* hostFunction, char* deviceFunction, const char* deviceName,
                           unsigned threadLimit, void* tid, void* bid, void* blockDim, void* gridDim, int* wSize) {
  static reg_fn real = NULL;
  if (!real) real = (reg_fn)real_sym("__hipRegisterFunction");
  if (g_n == g_cap) {
    g_cap = g_cap ? 2 * g_cap : 8192;
    g_host = (const void**)realloc((void*)g_host, sizeof(void*) * (size_t)g_cap);
    g_name = (const char**)realloc((void*)g_name, sizeof(char*) * (size_t)g_cap);
  }
  g_host[g_n] = hostFunction; g_name[g_n] = strdup(deviceName); ++g_n;
  if (real) real(modules, hostFunction, deviceFunction, deviceName, threadLimit, tid, bid, blockDim, gridDim, wSize);
}

int hipLaunchKernel(const void* func, dim3_t grid, dim3_t block, void** args, size_t shmem, void* stream) {
  static launch_fn real = NULL;
  if (!real) real = (launch_fn)real_sym("hipLaunchKernel");
  if (!real) { fprintf(stderr, "launch_log: the HIP runtime's hipLaunchKernel was not found\n"); abort(); }
  if (!g_log) { const char* p = getenv("SRGAN_LAUNCH_LOG"); g_log = p ? fopen(p, "w") : stderr; }
  const char* name = "?";
  for (int i = g_n - 1; i >= 0; --i) if (g_host[i] == func) { name = g_name[i]; break; }
  unsigned long long tx = (unsigned long long)grid.x * block.x, ty = (unsigned long long)grid.y * block.y,
                     tz = (unsigned long long)grid.z * block.z, th = (unsigned long long)block.x * block.y * block.z;
  if (!grid.x || !grid.y || !grid.z || !th || th > 1024 || tx >> 32 || ty >> 32 || tz >> 32 || shmem > 160 * 1024)
    fprintf(stderr, "BAD launch %s grid %u %u %u block %u %u %u lds %zu\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
  fprintf(g_log, "%lu %.120s %u %u %u %u %u %u %zu %p\n", g_seq++, name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem, stream);
  fflush(g_log);
  return real(func, grid, block, args, shmem, stream);
}
