// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on this GPU / runtime?
// Two one-wave kernels that each spin for a fixed time: launched in order they take twice that time, overlapped once.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void Spin(unsigned long long ticks, unsigned *out)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  unsigned n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) n++;
  if (out) out[blockIdx.x] = n;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
  unsigned *d = nullptr;
  CK(hipMalloc(&d, 4096));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long ticks = 20000;   // 200 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d);
      if (mode == 0) hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d + 1);
      if (mode == 1) hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d + 1);
      if (mode == 2) { hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d + 1);
                       hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks / 4, d + 2); }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "in order" : mode == 1 ? "second launch any-order" : "any-order, then an ordered 50 us kernel", ms * 1e3);
    }
  }
  return 0;
}
