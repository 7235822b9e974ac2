// Does hipExtAnyOrderLaunch let two small kernels of ONE stream overlap on gfx950?  (hip_ext.h: "not supported on GFX9xx")
// Pairs of 10-us spin kernels, 16 workgroups each: time per pair with and without the flag on the second launch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
  hipStream_t s;
  (void)hipStreamCreate(&s);
  const long long cyc = 1000;  // wall_clock64 runs at 100 MHz: 10 us
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2000; ++i) {
        hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, s, cyc, (int*)nullptr);
        hipExtLaunchKernelGGL(spin, dim3(16), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, cyc, (int*)nullptr);
      }
      (void)hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
      if (rep) printf("second launch %s: %.1f us per pair\n", flag ? "hipExtAnyOrderLaunch" : "in order", us);
    }
  }
  return 0;
}
