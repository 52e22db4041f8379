// Micro-benchmark: do float64 VALU instructions overlap with float64 MFMAs on gfx950, or do the two share the SIMD's float64 FMA datapath?
// (v_mfma_f64_16x16x4_f64 takes 64 cycles for 1024 multiply-adds = 16 per cycle and SIMD -- exactly the float64 rate of the vector ALU, and the
// chip's float64 matrix and vector peaks are the same number.)  The float64 K*V kernel (csrc/kv_f64.hpp) generates one covariance value per lane
// with ~25 v_*_f64 instructions between groups of MFMAs: if the two pipes overlapped it would be bound by the larger of the two, if they share
// the datapath by their SUM -- which decides what its roofline is.
// Per loop iteration of one wave: M MFMAs (independent accumulators) and/or V dependent-free v_fma_f64 (or v_fma_f32), in ONE instruction stream;
// and the two-wave form: MFMA waves and VALU waves side by side on the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_valu_overlap mfma_f64_valu_overlap.hip ; prints ms per case (1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define REP 4096

// MODE bit 0: 8 MFMAs per iteration; bit 1: 32 v_fma_f64; bit 2: 32 v_fma_f32; bit 3: role by wave (even waves MFMA, odd waves the VALU part)
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, double seed) {
  const double a = seed + threadIdx.x * 1e-3, b = seed * 0.5;
  const bool by_wave = MODE & 8;
  const int wave = threadIdx.x >> 6;
  const bool do_m = (MODE & 1) && (!by_wave || (wave & 4) == 0);     // (512 threads = 8 waves = 2 per SIMD: waves 0-3 / 4-7 land on SIMDs 0-3 each)
  const bool do_v = (MODE & 6) && (!by_wave || (wave & 4) != 0);
  (void)do_m; (void)do_v;
  f64x4 c[8];
  double d[32];
  float f[32];
  for (int i = 0; i < 8; ++i) c[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < 32; ++i) { d[i] = seed * i; f[i] = (float)seed * i; }
  const double m1 = 1.0 - 1e-9 * seed, a1 = 1e-9 * seed;
  const float m1f = (float)m1, a1f = (float)a1;
  if (by_wave) {
    // role by wave: waves 0-3 of the 8 issue the MFMAs, waves 4-7 the VALU part (each SIMD holds one of either)
    if (do_m) {
      for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
      }
    } else if (do_v) {
      for (int i = 0; i < REP; ++i) {
        if (MODE & 2) {
#pragma unroll
          for (int q = 0; q < 32; ++q) d[q] = __builtin_fma(d[q], m1, a1);
        }
        if (MODE & 4) {
#pragma unroll
          for (int q = 0; q < 32; ++q) f[q] = __builtin_fmaf(f[q], m1f, a1f);
        }
      }
    }
  } else {
    for (int i = 0; i < REP; ++i) {
      if (MODE & 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
          if (MODE & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) d[4 * u + q] = __builtin_fma(d[4 * u + q], m1, a1);
          }
          if (MODE & 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[4 * u + q] = __builtin_fmaf(f[4 * u + q], m1f, a1f);
          }
        }
      } else {
        if (MODE & 2) {
#pragma unroll
          for (int q = 0; q < 32; ++q) d[q] = __builtin_fma(d[q], m1, a1);
        }
        if (MODE & 4) {
#pragma unroll
          for (int q = 0; q < 32; ++q) f[q] = __builtin_fmaf(f[q], m1f, a1f);
        }
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  for (int i = 0; i < 32; ++i) s += d[i] + f[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(const char* name, double* out, int threads) {
  int cus = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, 1.0001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, 1.0001);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s threads/CU=%d  %8.3f ms  (%6.1f cycles per iteration @2.4GHz)\n", name, threads, ms, ms * 1e-3 * 2.4e9 / REP);
  return ms;
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * 8 * 2);
  for (int threads = 256; threads <= 512; threads *= 2) {
    run<1>("8 MFMA f64 16x16x4", out, threads);
    run<2>("32 v_fma_f64", out, threads);
    run<4>("32 v_fma_f32", out, threads);
    run<3>("8 MFMA f64 + 32 v_fma_f64, one instruction stream", out, threads);
    run<5>("8 MFMA f64 + 32 v_fma_f32, one instruction stream", out, threads);
  }
  run<1 | 8>("MFMA waves alone (4 of 8 waves active)", out, 512);
  run<2 | 8>("v_fma_f64 waves alone (4 of 8 waves active)", out, 512);
  run<1 | 2 | 8>("MFMA waves + v_fma_f64 waves side by side on each SIMD", out, 512);
  run<1 | 4 | 8>("MFMA waves + v_fma_f32 waves side by side on each SIMD", out, 512);
  return 0;
}
