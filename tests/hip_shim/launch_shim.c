/* Test infrastructure (CPU only): a stand-in for the 14 HIP runtime entry points libsrgan_hip.so imports, preloaded in front of
 * libamdhip64 by tests/test_launch_plan_cpu.py.  It executes nothing: every kernel launch is written to $SRGAN_SHIM_LOG as
 *   <mangled kernel name> gx gy gz bx by bz dynamic_lds
 * so that the grids the host code of csrc/ computes for every BASELINE geometry can be checked against the AQL / gfx950 limits
 * without a GPU.  Never linked into the product. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned x, y, z; } dim3_t;

static const void** g_host = NULL;     /* every library of the process registers here (torch's too): grown on demand */
static const char** g_name = NULL;
static int g_nfuncs = 0, g_cap = 0;
static FILE* g_log = NULL;

static __thread dim3_t t_grid, t_block;
static __thread size_t t_shmem;
static __thread void* t_stream;

static FILE* logf_(void) {
  if (!g_log) {
    const char* p = getenv("SRGAN_SHIM_LOG");
    g_log = p ? fopen(p, "w") : stdout;
  }
  return g_log;
}

void** __hipRegisterFatBinary(const void* data) { static void* handle[4]; (void)data; return handle; }
void __hipUnregisterFatBinary(void** m) { (void)m; if (g_log && g_log != stdout) fflush(g_log); }
void __hipRegisterFunction(void** modules, const void* hostFunction, char* deviceFunction, const char* deviceName,
                           unsigned threadLimit, void* tid, void* bid, void* blockDim, void* gridDim, int* wSize) {
  (void)modules; (void)deviceFunction; (void)threadLimit; (void)tid; (void)bid; (void)blockDim; (void)gridDim; (void)wSize;
  if (g_nfuncs == g_cap) {
    g_cap = g_cap ? 2 * g_cap : 4096;
    g_host = (const void**)realloc((void*)g_host, sizeof(void*) * (size_t)g_cap);
    g_name = (const char**)realloc((void*)g_name, sizeof(char*) * (size_t)g_cap);
  }
  g_host[g_nfuncs] = hostFunction; g_name[g_nfuncs] = strdup(deviceName); ++g_nfuncs;
}
int __hipPushCallConfiguration(dim3_t grid, dim3_t block, size_t shmem, void* stream) {
  t_grid = grid; t_block = block; t_shmem = shmem; t_stream = stream; return 0;
}
int __hipPopCallConfiguration(dim3_t* grid, dim3_t* block, size_t* shmem, void** stream) {
  *grid = t_grid; *block = t_block; *shmem = t_shmem; *stream = t_stream; return 0;
}
int hipLaunchKernel(const void* func, dim3_t grid, dim3_t block, void** args, size_t shmem, void* stream) {
  (void)args; (void)stream;
  const char* name = "?";
  for (int i = g_nfuncs - 1; i >= 0; --i) if (g_host[i] == func) { name = g_name[i]; break; }
  fprintf(logf_(), "%s %u %u %u %u %u %u %zu\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
  return 0;
}
int hipGetLastError(void) { return 0; }
const char* hipGetErrorString(int e) { (void)e; return "launch shim"; }
int hipGetDevice(int* d) { *d = 0; return 0; }
int hipDeviceGetAttribute(int* v, int attr, int dev) { (void)attr; (void)dev; *v = 256; return 0; }   /* only the CU count is asked */
int hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void* f, int bs, size_t dyn) { (void)f; (void)bs; (void)dyn; *n = 1; return 0; }
int hipEventCreate(void** e) { *e = NULL; return 1; }      /* "no events": the launch timer stays off */
int hipEventRecord(void* e, void* s) { (void)e; (void)s; return 0; }
int hipEventElapsedTime(float* ms, void* a, void* b) { (void)a; (void)b; *ms = 0.f; return 0; }
/* the shim marks a line in the log (which entry point the following launches belong to) */
void srgan_shim_mark(const char* what) { fprintf(logf_(), "# %s\n", what); }
