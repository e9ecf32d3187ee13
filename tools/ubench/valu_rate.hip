// Micro-benchmark: wave64 VALU issue cost per instruction kind on one SIMD, by resident waves.
// Every variant is a block of 8 independent instructions repeated; inline asm pins the opcode.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(64) Chain(float *out, int iters)
{
  float a[8];
  for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
  float m = 1.0001f, c = 0.5f;
  unsigned long long mask = 0x5555555555555555ull;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 1) {
#define OP(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 2) {
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "s"(mask));
        REP8(OP)
#undef OP
      } else if (MODE == 3) {
#define OP(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 4) {
#define OP(k) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 5) {
#define OP(k) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[k]) : "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 6) {
#define OP(k) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c) : "vcc");
        REP8(OP)
#undef OP
      } else if (MODE == 7) {
#define OP(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 8) {
#define OP(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[k & 6]) : "v"(*(double*)&m));
        REP8(OP)
#undef OP
      }
    }
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void Run(const char *name, int cus, int per)
{
  const int iters = 1000;
  for (int waves_per_simd : {1, 2, 8}) {
    const int blocks = cus * 4 * waves_per_simd;
    float *out;
    hipMalloc(&out, sizeof(float) * 64 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(Chain<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double instrs = double(iters) * 64 * per;  // wave-instructions per wave
    std::printf("%-22s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instr per SIMD\n", name, waves_per_simd, ms,
                ms * 1e6 / (instrs * waves_per_simd));
    hipFree(out);
  }
}

int main()
{
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::printf("clock %d kHz, %d CUs\n", prop.clockRate, cus);
  Run<0>("v_fma_f32", cus, 1);
  Run<7>("v_mul_f32", cus, 1);
  Run<1>("v_max_f32", cus, 1);
  Run<3>("v_max3_f32", cus, 1);
  Run<2>("v_cndmask_b32 (sgpr)", cus, 1);
  Run<6>("v_cmp+v_cndmask (vcc)", cus, 2);
  Run<4>("v_cvt_f32_u32_sdwa", cus, 1);
  Run<5>("v_alignbit_b32", cus, 1);
  Run<8>("v_pk_fma_f32", cus, 1);
  return 0;
}
