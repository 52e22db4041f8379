// Small right-hand-side count (t <= 8) fused covariance MVM with the squared distances generated on the matrix
// pipe (Gram form, see kv_gram.hpp) and the t products carried on the VALU.
//
// kv_valu.hpp spends 2*D + 1 VALU instructions + one quarter-rate v_exp_f32 per (i, j) pair and is bound by that
// (5.1e12 pairs/s measured at D = 3, the non-packed VALU roofline).  Here a 32 x 32 block of squared distances costs
// ceil((D+2)/2) MFMAs (the matrix pipe is otherwise idle), and the VALU only does  k = f(S)  and  acc += k * v_j :
// ~2 + T instructions per pair.  Same accuracy policy as kv_gram.hpp (host selects it only when max |z|^2 <= 32,
// never for Matern nu = 1/2).  Used by the predictive-mean solve and the Lanczos steps
// (gpytorch/models/exact_prediction_strategies.py:271,286), which are sequences of t = 1 products.
#pragma once
#include "kv_mfma.hpp"

namespace gpamd {

constexpr int KGV_BN = 512;  // j-tile
constexpr int KGV_NI = 4;    // 32-row tiles per wave
constexpr int KGV_BM = 4 * KGV_NI * 32;

template <int KIND, int D, int T>
__global__ __launch_bounds__(256) void kv_gramv_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KA = (D + 2 + 1) / 2, LDA = 2 * KA + 1;
  constexpr int BN = KGV_BN, NI = KGV_NI;
  __shared__ __attribute__((aligned(16))) float Vs[T * BN];
  __shared__ float Xa[BN * LDA];

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * KGV_BM + wave * (NI * 32);

  float bq[NI][KA];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 32 + l31, a.n - 1);
    float z[DP];
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP + 4 * q);
      z[4 * q + 0] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
    }
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) nn = __builtin_fmaf(z[k], z[k], nn);
#pragma unroll
    for (int q = 0; q < KA; ++q) {
      const int k0 = 2 * q, k1 = 2 * q + 1;
      const float v0 = (k0 < D) ? -2.f * z[k0 < D ? k0 : 0] : (k0 == D ? 1.f : (k0 == D + 1 ? nn : 0.f));
      const float v1 = (k1 < D) ? -2.f * z[k1 < D ? k1 : 0] : (k1 == D ? 1.f : (k1 == D + 1 ? nn : 0.f));
      bq[ni][q] = h ? v1 : v0;
    }
  }
  float acc[NI][T];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int c = 0; c < T; ++c) acc[ni][c] = 0.f;

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    for (int jj = tid; jj < BN; jj += 256) {
      const int j = j0 + jj;
      float z[DP];
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * q);
        z[4 * q + 0] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
      }
      float nn = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) nn = __builtin_fmaf(z[k], z[k], nn);
#pragma unroll
      for (int k = 0; k < 2 * KA; ++k)
        Xa[jj * LDA + k] = (k < D) ? z[k < D ? k : 0] : (k == D ? nn : (k == D + 1 ? (j < jend ? 1.f : 0.f) : 0.f));
#pragma unroll
      for (int c = 0; c < T; ++c) Vs[c * BN + jj] = (c < a.t && j < jend) ? a.Vt[(int64_t)c * a.ldv + j] : 0.f;
    }
    __syncthreads();

#pragma unroll 1
    for (int jb = 0; jb < BN; jb += 32) {
      float aq[KA];
#pragma unroll
      for (int q = 0; q < KA; ++q) aq[q] = Xa[(jb + l31) * LDA + 2 * q + h];
      f32x4 vv[T][4];  // v_c[j] for this half-wave's 16 rows: 4 groups of 4 consecutive j
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) vv[c][g] = *reinterpret_cast<const f32x4*>(&Vs[c * BN + jb + 8 * g + 4 * h]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x16 kk;
#pragma unroll
        for (int r = 0; r < 16; ++r) kk[r] = 0.f;
#pragma unroll
        for (int q = 0; q < KA; ++q) kk = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q], bq[ni][q], kk, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sv = kk[r];
          if constexpr (KIND != KIND_RBF) sv = __builtin_amdgcn_fmed3f(sv, 0.f, 3.0e38f);
          const float kv = cov_from_sq<KIND>(sv);
#pragma unroll
          for (int c = 0; c < T; ++c) acc[ni][c] = __builtin_fmaf(kv, vv[c][r >> 2][r & 3], acc[ni][c]);
        }
      }
    }
  }

  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 32 + l31;
#pragma unroll
    for (int c = 0; c < T; ++c) {
      const float tot = acc[ni][c] + __shfl_xor(acc[ni][c], 32, 64);
      if (h == 0 && i < a.n && c < a.t) Pout[(int64_t)c * a.ldo + i] = tot;
    }
  }
}

}  // namespace gpamd
