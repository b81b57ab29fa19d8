// prints the SIMD each wave of a 512-thread workgroup lands on (gfx950), 1 block per CU because of the LDS size
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
  __shared__ float big[30000];
  big[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (int)id + (big[5] > 1e9f ? 1 : 0);
}
int main() {
  int* d; hipMalloc(&d, 4 * 8 * sizeof(int));
  k<<<4, 512>>>(d);
  int h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { for (int w = 0; w < 8; ++w) printf("b%d w%d: wave_id %u simd %u cu %u | ", b, w, h[b*8+w] & 15, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15); printf("\n"); }
  return 0;
}
