// Micro-benchmark: issue cost of the small MFMA shapes considered for the 9..32-column contraction tile (gfx950), and a
// layout check of v_mfma_f32_4x4x1_16B_f32 (A: lane 4b+a, B: lane 4b+n, D: lane 4b+n reg a).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rates mfma_rates.hip ; prints cycles per MFMA per SIMD (one wave/SIMD and two).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define REP 16384

// MODE 0: 4x4x1 f32 (8 independent accumulators)   1: 16x16x4 f32   2: 16x16x1_4B f32   3: 32x32x2 f32
// MODE 4: 16x16x32 f16   5: 32x32x16 f16   6: 16x16x16 f16 (legacy)   7: 4x4x1 with one v_exp_f32 per 3 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  const float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = (f32x4)(0.f);
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
  f16x8 h8;
  f16x4 h4;
  for (int e = 0; e < 8; ++e) h8[e] = (_Float16)(a + e);
  for (int e = 0; e < 4; ++e) h4[e] = (_Float16)(a + e);
  float x0 = a, x1 = a + 1, x2 = a + 2;
  for (int i = 0; i < REP; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c4[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4[u], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c4[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[u], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c16[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_16x16x1f32(b, a, c16[u], 0, 0, 0);
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c16[u], 0, 0, 0);
    } else if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8, h8, c4[u], 0, 0, 0);
    } else if (MODE == 5) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8, h8, c16[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8, h8, c16[u], 0, 0, 0);
    } else if (MODE == 6) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c4[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, h4, c4[u], 0, 0, 0);
    } else if (MODE == 7) {  // the planned inner loop mix: 1 exp feeding 3 MFMAs, x 3 (9 MFMAs, 3 exps); 8 "instr" counted = MFMAs only below
      asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
      c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x0, c4[0], 0, 0, 0);
      c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, x0, c4[1], 0, 0, 0);
      c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x0, c4[2], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0" : "+v"(x1));
      c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x1, c4[3], 0, 0, 0);
      c4[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, x1, c4[4], 0, 0, 0);
      c4[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x1, c4[5], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0" : "+v"(x2));
      c4[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, x2, c4[6], 0, 0, 0);
      c4[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, x2, c4[7], 0, 0, 0);
    }
  }
  float s = x0 + x1 + x2;
  for (int i = 0; i < 8; ++i) s += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c16[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, double macs_per_instr) {
  for (int wps = 1; wps <= 2; ++wps) {  // waves per SIMD
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e-3 * 2.4e9 / ((double)wps * REP * 8);
    double tf = 2.0 * macs_per_instr * 8.0 * REP * wps * 1024.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD=%d %8.3f ms  %6.2f cycles/MFMA/SIMD @2.4GHz  %7.1f TFLOP/s\n", name, wps, ms, per, tf);
  }
}

// layout check: D[a][n] of block b must equal sum over the single k of A_b[a] * B_b[n]
__global__ void layout4x4(float* out) {
  const int l = threadIdx.x;
  const float a = 1.0f + l;          // A_b[a] with b = l >> 2, a = l & 3   (if the assumed layout holds)
  const float b = 100.0f + 3.0f * l; // B_b[n] with b = l >> 2, n = l & 3
  f32x4 c = (f32x4)(0.f);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

int main() {
  float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
  run<0>("v_mfma_f32_4x4x1_16B_f32", out, 256);
  run<1>("v_mfma_f32_16x16x4_f32", out, 1024);
  run<2>("v_mfma_f32_16x16x1_4B_f32", out, 1024);
  run<3>("v_mfma_f32_32x32x2_f32", out, 2048);
  run<4>("v_mfma_f32_16x16x32_f16", out, 8192);
  run<5>("v_mfma_f32_32x32x16_f16", out, 16384);
  run<6>("v_mfma_f32_16x16x16_f16", out, 4096);
  run<7>("4x4x1 x8 + 3 v_exp_f32", out, 256);
  hipLaunchKernelGGL(layout4x4, dim3(1), dim3(64), 0, 0, out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l >> 2, n = l & 3;
      const float want = (1.0f + (4 * blk + r)) * (100.0f + 3.0f * (4 * blk + n));  // D_blk[a = r][n] held by lane 4 blk + n
      if (h[l * 4 + r] != want) { if (bad < 8) printf("layout mismatch lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); ++bad; }
    }
  printf("4x4x1 layout check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  return 0;
}
