// Micro-benchmark: wave64 VALU issue cost per instruction kind on one SIMD at 1, 2, 4 and 6 resident waves.
// Every variant is a block of 8 independent instructions repeated; inline asm pins the opcode.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(64) Chain(float *out, int iters)
{
  float a[8];
  for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
  float m = 1.0001f, c = 0.5f;
  unsigned sel = 0x03020504u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define ASM1(str) asm volatile(str : "+v"(a[k]) : "v"(m), "v"(c), "v"(sel));
      if (MODE == 0) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 1) {
#define OP(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 2) {
#define OP(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 3) {
#define OP(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 4) {
#define OP(k) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[k]), "v"(m) : "vcc");
        REP8(OP)
#undef OP
      } else if (MODE == 5) {
#define OP(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 6) {
#define OP(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 7) {
#define OP(k) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 8) {
#define OP(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 9) {
#define OP(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 10) {
#define OP(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 11) {
#define OP(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 12) {
#define OP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 13) {
#define OP(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 14) {
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(m) : );
        REP8(OP)
#undef OP
      } else if (MODE == 15) {
#define OP(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 16) {
#define OP(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 17) {
#define OP(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(sel), "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 18) {
#define OP(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[k & 6]) : "v"(*(double*)&m));
        REP8(OP)
#undef OP
      } else if (MODE == 19) {
#define OP(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 20) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_perm_b32 %0, %0, %1, %3" : "+v"(a[k]) : "v"(m), "v"(c), "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 21) {
#define OP(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 22) {
#define OP(k) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 23) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        REP8(OP)
#undef OP
      } else if (MODE == 24) {
#define OP(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(sel), "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 25) {
#define OP(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 26) {
#define OP(k) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 27) {
#define OP(k) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 28) {
#define OP(k) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 29) {
#define OP(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 30) {
#define OP(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 31) {
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a[k]) : "v"(m));
        REP8(OP)
#undef OP
      } else if (MODE == 32) {
#define OP(k) asm volatile("v_cmp_le_f32 s[12:13], %0, %1" : : "v"(a[k]), "v"(m) : "s12","s13");
        REP8(OP)
#undef OP
      } else if (MODE == 33) {
#define OP(k) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 34) {
#define OP(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 35) {
#define OP(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 36) {
#define OP(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 37) {
#define OP(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[k]) : "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 38) {
#define OP(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_and_or_b32 %0, %0, %3, %3" : "+v"(a[k]) : "v"(m), "v"(c), "v"(sel));
        REP8(OP)
#undef OP
      } else if (MODE == 39) {
#define OP(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        REP8(OP)
#undef OP
      } else if (MODE == 40) {
#define OP(k) asm volatile("v_med3_f32 %0, %0, %1, %2\n\tv_perm_b32 %0, %0, %1, %3" : "+v"(a[k]) : "v"(m), "v"(c), "v"(sel));
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
  std::printf("%-24s", name);
  for (int waves_per_simd : {1, 2, 4, 6}) {
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
    std::printf("  w%d: %6.3f ns", waves_per_simd, ms * 1e6 / (instrs * waves_per_simd));
    hipFree(out);
  }
  std::printf("   (per wave-instr per SIMD)\n");
}

int main()
{
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::printf("clock %d kHz, %d CUs\n", prop.clockRate, cus);
  Run<0>("v_fma_f32", cus, 1);
  Run<16>("v_fmac_f32", cus, 1);
  Run<10>("v_mul_f32", cus, 1);
  Run<22>("v_mul_legacy_f32", cus, 1);
  Run<3>("v_add_f32", cus, 1);
  Run<15>("v_sub_f32", cus, 1);
  Run<11>("v_max_f32", cus, 1);
  Run<2>("v_min3_f32", cus, 1);
  Run<19>("v_med3_f32", cus, 1);
  Run<1>("v_perm_b32", cus, 1);
  Run<4>("v_cmp_le_f32 vcc", cus, 1);
  Run<14>("v_cndmask_b32 vcc", cus, 1);
  Run<5>("v_lshl_add_u32", cus, 1);
  Run<21>("v_add_u32", cus, 1);
  Run<9>("v_and_b32", cus, 1);
  Run<8>("v_mov_b32", cus, 1);
  Run<17>("v_mad_u32_u24", cus, 1);
  Run<6>("v_cvt_f32_ubyte1", cus, 1);
  Run<7>("v_fma_mix_f32", cus, 1);
  Run<12>("v_rcp_f32", cus, 1);
  Run<13>("v_div_fixup_f32", cus, 1);
  Run<18>("v_pk_mul_f32", cus, 1);
  Run<20>("fma+perm pair", cus, 2);
  Run<23>("fma+min3 pair", cus, 2);
  Run<24>("v_and_or_b32", cus, 1);
  Run<25>("v_or_b32", cus, 1);
  Run<26>("v_lshrrev_b32", cus, 1);
  Run<27>("v_bfe_u32", cus, 1);
  Run<28>("v_alignbit_b32", cus, 1);
  Run<29>("v_xor_b32", cus, 1);
  Run<30>("v_min_u32", cus, 1);
  Run<31>("v_cndmask_b32 sgpr", cus, 1);
  Run<32>("v_cmp_le_f32 sgpr", cus, 1);
  Run<33>("v_lshl_or_b32", cus, 1);
  Run<34>("v_add3_u32", cus, 1);
  Run<35>("v_cvt_f32_u32", cus, 1);
  Run<36>("v_sub_u32", cus, 1);
  Run<37>("v_mul_u32_u24", cus, 1);
  Run<38>("fma+and_or pair", cus, 2);
  Run<39>("v_mov_b32_dpp", cus, 1);
  Run<40>("med3+perm pair", cus, 2);
  return 0;
}
