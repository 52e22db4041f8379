// Micro-benchmark: issue rates of the VALU instructions the small-t generation loop is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 65536
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const f32x2 c = {seed, seed * 0.5f};
  for (int i = 0; i < REP; ++i) {
    if (MODE == 0) {  // 8 v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                   "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
    } else if (MODE == 1) {  // 8 v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 2) {  // 4 v_pk_fma_f32 (8 flop-lanes... 8 results)
      asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                   "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
    } else if (MODE == 3) {  // 4 exp interleaved with 4 fma
      asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %5\n"
                   "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
    } else if (MODE == 4) {  // 2 exp + 6 fma
      asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n"
                   "v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
    } else if (MODE == 5) {  // 8 v_sqrt_f32
      asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                   "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 6) {  // 2 exp + 3 pk_fma (the packed generation mix per 2 pairs, D = 3 incl. sub folded: approx)
      asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %2, %2, %6, %2\n v_pk_fma_f32 %3, %3, %6, %3\n"
                   "v_exp_f32 %1, %1\n v_pk_fma_f32 %4, %4, %6, %4\n v_pk_fma_f32 %5, %5, %6, %5\n"
                   : "+v"(a0), "+v"(a1), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
    } else if (MODE == 7) {  // 4 v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(const char* name, int ninstr, float* out) {
  // 256 CUs x 4 SIMDs; 2 blocks of 256 threads per CU -> 2 waves per SIMD
  const int blocks = 256 * 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 waves x REP x ninstr wave-instructions
  double per = ms * 1e-3 * 2.4e9 / (2.0 * REP * ninstr);
  printf("%-28s %8.3f ms  %6.2f cycles/wave-instr @2.4GHz\n", name, ms, per);
}
int main() {
  float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
  run<0>("v_fma_f32 x8", 8, out);
  run<1>("v_exp_f32 x8", 8, out);
  run<2>("v_pk_fma_f32 x8", 8, out);
  run<7>("v_pk_add_f32 x4", 4, out);
  run<5>("v_sqrt_f32 x8", 8, out);
  run<3>("4 exp + 4 fma", 8, out);
  run<4>("2 exp + 6 fma", 8, out);
  run<6>("2 exp + 4 pk_fma", 6, out);
  return 0;
}
