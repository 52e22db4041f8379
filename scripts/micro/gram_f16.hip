// Micro-test: squared distances of a 32 x 32 block from ONE v_mfma_f32_32x32x16_f16 on hi/lo-split augmented
// coordinates (D = 3), checked against float64; includes tiny coordinates (f16-subnormal lo parts).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

__global__ void k(const float* zj, const float* zi, float* S) {  // zj, zi: [32][3]; S[j][i]
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  float a[3], b[3];
  float nj = 0.f, ni = 0.f;
  for (int q = 0; q < 3; ++q) {
    a[q] = zj[l31 * 3 + q];
    b[q] = zi[l31 * 3 + q];
    nj += a[q] * a[q];
    ni += b[q] * b[q];
  }
  f16x8 A, B;
  _Float16 ah, al, bh, bl;
  if (h == 0) {
    for (int q = 0; q < 2; ++q) {
      split(a[q], ah, al);
      split(-2.f * b[q], bh, bl);
      A[4 * q + 0] = ah; B[4 * q + 0] = bh;
      A[4 * q + 1] = ah; B[4 * q + 1] = bl;
      A[4 * q + 2] = al; B[4 * q + 2] = bh;
      A[4 * q + 3] = al; B[4 * q + 3] = bl;
    }
  } else {
    split(a[2], ah, al);
    split(-2.f * b[2], bh, bl);
    A[0] = ah; B[0] = bh;
    A[1] = ah; B[1] = bl;
    A[2] = al; B[2] = bh;
    A[3] = al; B[3] = bl;
    _Float16 nh, nl;
    split(nj, nh, nl);
    A[4] = nh; B[4] = (_Float16)1.f;
    A[5] = nl; B[5] = (_Float16)1.f;
    split(ni, nh, nl);
    A[6] = (_Float16)1.f; B[6] = nh;
    A[7] = (_Float16)1.f; B[7] = nl;
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // j
    S[row * 32 + l31] = c[r];                  // [j][i]
  }
}

int main() {
  float hzj[96], hzi[96], hS[1024];
  double worst = 0, worst_small = 0;
  for (int trial = 0; trial < 3; ++trial) {
    srand(1 + trial);
    double scale = trial == 0 ? 5.6 : (trial == 1 ? 0.2 : 1e-3);
    for (int i = 0; i < 96; ++i) {
      hzj[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * scale * 0.577);
      hzi[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * scale * 0.577);
    }
    if (trial == 1) { hzj[0] = 3.0f; hzi[5] = -3.0f; hzj[7] = 1e-5f; hzi[9] = 3e-6f; }  // mixed magnitudes
    float *dj, *di, *dS;
    hipMalloc(&dj, 384); hipMalloc(&di, 384); hipMalloc(&dS, 4096);
    hipMemcpy(dj, hzj, 384, hipMemcpyHostToDevice);
    hipMemcpy(di, hzi, 384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dj, di, dS);
    hipMemcpy(hS, dS, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int j = 0; j < 32; ++j)
      for (int i = 0; i < 32; ++i) {
        double ref = 0;
        for (int q = 0; q < 3; ++q) { double d = (double)hzj[j * 3 + q] - (double)hzi[i * 3 + q]; ref += d * d; }
        double e = fabs(hS[j * 32 + i] - ref);
        if (e > maxerr) maxerr = e;
        if (ref > maxref) maxref = ref;
      }
    printf("trial %d scale %.3g: max |S - ref| = %.3e   (max S = %.3g)\n", trial, scale, maxerr, maxref);
  }
  return 0;
}
