// Micro-benchmark: cost of divergent 16-byte-per-lane gathers (global_load_dwordx4) from a table that
// fits L1 / L2 / Infinity Cache, as a function of active lanes per wave and of how many lanes share a
// cache line.  Reports ns per wave-instruction per CU and the implied lane-loads per ns per CU.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) Gather(const float4 *table, uint32_t mask_records, float *out, int iters,
                                              int active, int share)
{
  const uint32_t lane = threadIdx.x & 63u;
  if (static_cast<int>(lane) >= active) return;          // exec-masked lanes: the wave issues with fewer lanes
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) / share * 2654435761u + 12345u;   // `share` neighbouring lanes share a seed
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s = s * 1664525u + 1013904223u;
      const float4 v = table[(s >> 8) & mask_records];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 200;
  const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves per CU
  float *out;
  (void)hipMalloc(&out, sizeof(float) * 256 * blocks);
  for (size_t bytes : {size_t(16) << 10, size_t(2) << 20, size_t(8) << 20, size_t(64) << 20}) {
    const uint32_t records = static_cast<uint32_t>(bytes / 16);
    float4 *table;
    (void)hipMalloc(&table, bytes);
    (void)hipMemset(table, 0, bytes);
    for (int share : {1, 4})
      for (int active : {64, 32, 16, 8}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(Gather, dim3(blocks), dim3(256), 0, 0, table, records - 1, out, iters, active, share);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double wave_instrs_per_cu = 32.0 * iters * 8;
        std::printf("table %6zu KB share %d active %2d: %.3f ms -> %.2f ns per wave-load per CU, %.2f lane-loads/ns/CU, %.1f GB/s/CU\n",
                    bytes >> 10, share, active, ms, ms * 1e6 / wave_instrs_per_cu, wave_instrs_per_cu * active / (ms * 1e6),
                    wave_instrs_per_cu * active * 16 / (ms * 1e6));
      }
    (void)hipFree(table);
  }
  return 0;
}
