// Micro-benchmark behind the host-pointer path of ray_scene_intersection: what pinning a caller's buffer in place
// costs, what a copy through pinned staging costs with 1..16 copy threads, and what the DMA itself sustains.
// Build: hipcc -O2 host_xfer.cpp -o host_xfer -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  const size_t bytes = size_t(2073600) * 64;   // one 1080p ray table
  char *user = static_cast<char *>(malloc(bytes)), *user2 = static_cast<char *>(malloc(bytes));
  memset(user, 1, bytes); memset(user2, 2, bytes);
  void *dev; (void)hipMalloc(&dev, bytes);
  void *pinned; (void)hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); (void)hipMemcpy(dev, user, bytes, hipMemcpyHostToDevice); double t1 = now();
    (void)hipMemcpy(user2, dev, bytes, hipMemcpyDeviceToHost); double t2 = now();
    printf("pageable hipMemcpy: H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); hipError_t e = hipHostRegister(user, bytes, hipHostRegisterDefault); double t1 = now();
    (void)hipMemcpy(dev, user, bytes, hipMemcpyHostToDevice); double t2 = now();
    (void)hipHostUnregister(user); double t3 = now();
    printf("register in place: register %.2f ms (%s), H2D %.2f ms (%.1f GB/s), unregister %.2f ms\n", (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); (void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice); double t1 = now();
    (void)hipMemcpy(pinned, dev, bytes, hipMemcpyDeviceToHost); double t2 = now();
    printf("pinned hipMemcpy: H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
  }
  for (int threads : {1, 2, 4, 8, 16}) {
    for (int dir = 0; dir < 2; ++dir) {
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
          th.emplace_back([&, t] {
            size_t lo = bytes * t / threads, hi = bytes * (t + 1) / threads;
            if (dir == 0) memcpy(static_cast<char *>(pinned) + lo, user + lo, hi - lo);
            else memcpy(user2 + lo, static_cast<char *>(pinned) + lo, hi - lo);
          });
        for (auto &x : th) x.join();
        best = std::min(best, now() - t0);
      }
      printf("%2d threads memcpy %s: %.2f ms (%.1f GB/s)\n", threads, dir ? "pinned->user" : "user->pinned", best * 1e3, bytes / best / 1e9);
    }
  }
  return 0;
}
