// Micro-benchmark for the split-operand product (kv_gramh.hpp): issue cost of the VALU instructions of its generation phase and how
// far they overlap with v_mfma_f32_32x32x16_f16 issued by ANOTHER wave of the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o split_rates split_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP 8192

// role 0: VALU op chain (OP)   role 1: MFMA 32x32x16 f16 (4 independent accumulators)
// OP 0 v_exp_f32  1 v_fma_mix_f32  2 v_cvt_pkrtz_f16_f32  3 v_pk_fma_f32  4 v_fma_f32  5 the generation mix (16 exp, 16 mix, 16 cvt)
template <int OP>
__device__ __forceinline__ float valu_loop(float seed) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  uint32_t p = 0x3c003c00u;
  for (int it = 0; it < REP; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[u]));
      if (OP == 1) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[u]) : "v"(p));
      if (OP == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x[u]) : "v"(x[(u + 1) & 7]));
      if (OP == 3) { typedef float f2 __attribute__((ext_vector_type(2))); }
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[u]) : "v"(seed));
    }
    if (OP == 3) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[0]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[4]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[6]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[0]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[4]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[6]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[0]) : "v"(*(double*)&x[2]));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[4]) : "v"(*(double*)&x[2]));
    }
    if (OP == 5) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[u]));
        asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[u]) : "v"(p));
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x[u]) : "v"(x[(u + 1) & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  return s;
}

__device__ __forceinline__ float mfma_loop(float seed, int reps) {
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  f16x8 h8;
  for (int e = 0; e < 8; ++e) h8[e] = (_Float16)(seed + e);
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8, h8, c[u & 3], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  return s;
}

// roles: 0 = every wave VALU, 1 = every wave MFMA, 2 = waves 0-3 of the 512-thread block VALU, waves 4-7 MFMA (one wave of each per SIMD)
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, float seed, int roles, int prio) {
  const bool valu = roles == 0 || (roles == 2 && (threadIdx.x >> 8) == 0);
  float s;
  if (valu) {
    s = valu_loop<OP>(seed + threadIdx.x * 1e-3f);
  } else {
    if (prio) __builtin_amdgcn_s_setprio(1);
    s = mfma_loop(seed, REP);
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* out, double ops_per_it) {
  for (int roles = 0; roles <= 3; ++roles) {
    const int r = roles == 3 ? 2 : roles, prio = roles == 3;
    const int blocks = 256, threads = r == 2 ? 512 : 256;
    if (roles == 1 && OP != 0) continue;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, r, prio);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, r, prio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;
    if (r == 0) {
      printf("%-28s alone: %8.3f ms  %6.2f cycles per instruction (@2.4 GHz, one wave per SIMD)\n", name, ms, cyc / (REP * ops_per_it));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, 1.0001f, r, prio);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-28s alone: %8.3f ms  %6.2f cycles per instruction (@2.4 GHz, two waves per SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / (2 * REP * ops_per_it));
    }
    if (r == 1) printf("%-28s alone: %8.3f ms  %6.2f cycles per MFMA\n", "v_mfma_f32_32x32x16_f16", ms, cyc / (REP * 8.0));
    if (r == 2) printf("%-28s + MFMA wave on the same SIMD (prio %d): %8.3f ms (sum of the two alone = serial, max = full overlap)\n", name, prio, ms);
  }
}

// one wave per SIMD, ONE instruction stream: MFMA followed by NV independent VALU instructions (v_fma_f32), repeated
template <int NV>
__global__ __launch_bounds__(256) void kmix(float* out, float seed) {
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  f16x8 h8;
  for (int e = 0; e < 8; ++e) h8[e] = (_Float16)(seed + e);
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x * 1e-3f;
  for (int it = 0; it < REP; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8, h8, c[u & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[v & 7]) : "v"(seed));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV>
void runmix(float* out) {
  for (int threads = 256; threads <= 512; threads += 256) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kmix<NV>, dim3(256), dim3(256), 0, 0, out, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmix<NV>, dim3(256 * (threads / 256)), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one stream: MFMA + %d v_fma_f32, %d wave(s)/SIMD: %8.3f ms  %6.2f cycles per MFMA slot (@2.4 GHz)\n", NV, threads / 256, ms,
           ms * 1e-3 * 2.4e9 / (REP * 8.0 * (threads / 256)));
  }
}

int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  run<0>("v_exp_f32", out, 8);
  run<1>("v_fma_mix_f32", out, 8);
  run<2>("v_cvt_pkrtz_f16_f32", out, 8);
  run<3>("v_pk_fma_f32", out, 8);
  run<4>("v_fma_f32", out, 8);
  run<5>("mix exp+fma_mix+cvt_pkrtz", out, 24);
  runmix<0>(out); runmix<2>(out); runmix<4>(out); runmix<6>(out); runmix<8>(out); runmix<12>(out);
  return 0;
}
