// Micro-benchmark: the float64 matrix-pipe rate of gfx950, MEASURED (the local microarchitecture guide lists no float64 MFMA peak).
// v_mfma_f64_16x16x4_f64 (1024 MAC) and v_mfma_f64_4x4x4_4B_f64, eight independent accumulators per wave, one / two / four waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip ; prints cycles per MFMA per SIMD and TFLOP/s for the whole chip.
// The clock is not assumed: the elapsed time is converted with the shader clock that s_memrealtime / wall clock imply is NOT available here, so
// both "cycles @ 2.4 GHz" and the clock-independent TFLOP/s are printed; scripts read the latter.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define REP 8192

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
  const double a = seed + threadIdx.x * 1e-3, b = seed * 0.5;
  f64x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < REP; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[u], 0, 0, 0);
      else {
        double d1 = c[u][0];
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
        c[u][0] = d1;
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* out, double macs_per_instr) {
  int cus = 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double simds = 4.0 * cus;
    const double per = ms * 1e-3 * 2.4e9 / ((double)wps * REP * 8);
    const double tf = 2.0 * macs_per_instr * 8.0 * REP * wps * simds / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD=%d %8.3f ms  %6.2f cycles/MFMA/SIMD @2.4GHz  %7.2f TFLOP/s (f64, %d CUs)\n", name, wps, ms, per, tf, cus);
  }
}

int main() {
  double* out; hipMalloc(&out, 256 * 4 * 256 * 8 * 2);
  run<0>("v_mfma_f64_16x16x4_f64", out, 1024);
  run<1>("v_mfma_f64_4x4x4_4B_f64", out, 64 * 4);
  return 0;
}
