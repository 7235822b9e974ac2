// what v_permlane32_swap does to (u, u): prints lanes 0, 31, 32, 63 of both results
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k2(unsigned* p) {
  unsigned a = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k2<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i : {0, 31, 32, 63}) printf("lane %d: r0 %u r1 %u\n", i, h[i], h[64 + i]);
  return 0;
}
