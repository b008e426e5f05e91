// does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx" for the module API)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
__global__ void spin(long long ticks, long long *ts) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) ts[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) ts[1] = wall_clock64();
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  long long *ts; CK(hipMalloc((void **)&ts, 64 * 8));
  long long h[16];
  for (int rep = 0; rep < 3; rep++) {
    for (int flags = 0; flags < 2; flags++) {
      CK(hipMemset(ts, 0, 64 * 8));
      // A (50 us, normal), B (300 us, normal), C (20 us, flags), D (20 us, normal)
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, 5000LL, ts + 0);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, 30000LL, ts + 2);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, 2000LL, ts + 4);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, 2000LL, ts + 6);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, ts, 8 * 8, hipMemcpyDeviceToHost));
      const double u = 0.01;   // us per tick at 100 MHz
      printf("flags=%d  A %.1f..%.1f  B %.1f..%.1f  C %.1f..%.1f  D %.1f..%.1f   (C %s B)\n", flags, 0.0, (h[1] - h[0]) * u, (h[2] - h[0]) * u, (h[3] - h[0]) * u,
             (h[4] - h[0]) * u, (h[5] - h[0]) * u, (h[6] - h[0]) * u, (h[7] - h[0]) * u, h[4] < h[3] ? "OVERLAPS" : "after");
    }
  }
  return 0;
}
