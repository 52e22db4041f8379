// Sustained rate of v_mfma_f32_32x32x16_f16 under the chip's power cap: a pure register-resident MFMA loop (four independent accumulators per
// wave, WAVES waves per SIMD on every CU) run for several seconds, achieved TFLOP/s printed per second.  Run beside
// `rocm-smi --showclocks --showpower` (scripts/r5_s2.sh) to read the shader clock the chip holds under this load: the 2.5 PFLOP/s dense f16 peak of
// MI355X_MICROARCH.md is quoted at 2.4 GHz, which a kernel that keeps the f16 matrix pipe busy does not see.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_sustained mfma_f16_sustained.hip ; usage: mfma_f16_sustained [seconds] [waves per SIMD 1..4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP 4096

__global__ __launch_bounds__(256) void k(float* out, float seed) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + 1e-3f * threadIdx.x + e); b[e] = (_Float16)(0.5f * seed - e); }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u], 0, 0, 0);
  }
  float s = 0.f;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += c[u][r];
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int waves = argc > 2 ? atoi(argv[2]) : 2;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, 4);
  const int blocks = cus * waves;   // 256 threads = 4 waves = one per SIMD; `waves` workgroups per CU
  const double flop = 2.0 * 32 * 32 * 16 * 4.0 * REP * 4.0 * blocks;   // per launch: 4 MFMAs x REP per wave, 4 waves per block
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  double last = 0.0;
  long launches = 0, last_l = 0;
  while (true) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    launches += 50;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (t - last >= 1.0) {
      printf("{\"t\": %.2f, \"waves_per_simd\": %d, \"tflops_f16_mfma\": %.1f}\n", t, waves, flop * (launches - last_l) / (t - last) / 1e12);
      fflush(stdout);
      last = t; last_l = launches;
    }
    if (t >= secs) break;
  }
  return 0;
}
