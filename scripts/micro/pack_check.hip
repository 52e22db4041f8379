// Compares gram_pack_a (LDS image) with gram_pack_a_lane (registers) slot by slot for given points.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../gpytorch_amd/csrc/gram_f16.hpp"
using namespace gpamd;
constexpr int D = 3;
__global__ void k(const float* X, int m, float* outA, float* outL) {  // out: [m][16] slot values as float
  __shared__ __attribute__((aligned(16))) _Float16 Xh[128 * 16];
  const int tid = threadIdx.x, j0 = blockIdx.x * 128;
  if (tid < 128) {
    const int j = j0 + tid;
    float z[4] = {0, 0, 0, 0};
    if (j < m) { z[0] = X[j * 4]; z[1] = X[j * 4 + 1]; z[2] = X[j * 4 + 2]; }
    gram_pack_a<D>(z, j < m, Xh, tid, 128);
  }
  __syncthreads();
  // lanes: tid -> row (tid & 127), h = tid >> 7
  const int row = tid & 127, h = tid >> 7;
  const int j = j0 + row;
  float z[4] = {0, 0, 0, 0};
  if (j < m) { z[0] = X[j * 4]; z[1] = X[j * 4 + 1]; z[2] = X[j * 4 + 2]; }
  f16x8 aq[1];
  gram_pack_a_lane<D>(z, j < m, h, aq);
  if (j < m)
    for (int e = 0; e < 8; ++e) {
      outA[j * 16 + 8 * h + e] = (float)Xh[row * 16 + 8 * h + e];
      outL[j * 16 + 8 * h + e] = (float)aq[0][e];
    }
}
int main() {
  const int m = 1300;
  float* hX = (float*)malloc(m * 16);
  srand(3);
  for (int i = 0; i < m * 4; ++i) hX[i] = (i % 4 == 3) ? 0.f : (float)((rand() / (double)RAND_MAX * 2 - 1) * 2.0);
  // the point that misbehaved
  hX[1229 * 4 + 0] = -0.9970036149024963f; hX[1229 * 4 + 1] = 0.7151787281036377f; hX[1229 * 4 + 2] = 1.5579217672348022f;
  float *dX, *dA, *dL;
  hipMalloc(&dX, m * 16); hipMalloc(&dA, m * 64); hipMalloc(&dL, m * 64);
  hipMemcpy(dX, hX, m * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((m + 127) / 128), dim3(256), 0, 0, dX, m, dA, dL);
  float* hA = (float*)malloc(m * 64); float* hL = (float*)malloc(m * 64);
  hipMemcpy(hA, dA, m * 64, hipMemcpyDeviceToHost); hipMemcpy(hL, dL, m * 64, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < m; ++j)
    for (int s = 0; s < 16; ++s)
      if (hA[j * 16 + s] != hL[j * 16 + s]) {
        if (bad < 12) printf("row %d slot %d: lds %.9g lane %.9g  (z = %.9g %.9g %.9g)\n", j, s, hA[j * 16 + s], hL[j * 16 + s], hX[j * 4], hX[j * 4 + 1], hX[j * 4 + 2]);
        ++bad;
      }
  printf("mismatching slots: %d\n", bad);
  printf("row 1229 lds :"); for (int s = 0; s < 16; ++s) printf(" %.6g", hA[1229 * 16 + s]); printf("\n");
  return 0;
}
