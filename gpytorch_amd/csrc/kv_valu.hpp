// Small right-hand-side count (t <= 8) fused covariance MVM on the vector ALU.
//
// With fewer than ~16 columns the contraction cannot feed the matrix pipe: the path is bound by
// kernel GENERATION (one v_exp_f32 [+ v_sqrt_f32] and 2*DP+1 VALU ops per (i, j) pair), so each lane
// owns RPT rows, every staged x_j / v_j is an LDS broadcast read shared by the whole wave, and the
// t products ride on v_fmac.  Used for the predictive-mean solve (t = 1,
// gpytorch/models/exact_prediction_strategies.py:286), Lanczos steps (t = 1, :271) and
// K_*X @ mean_cache (:396).
//
// Same probe-major layout and split-j partial-slab convention as kv_mfma.hpp.
#pragma once
#include "common.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

constexpr int KVV_BN = 256;   // j-tile
constexpr int KVV_RPT = 2;    // rows per thread
constexpr int KVV_BM = 256 * KVV_RPT;

template <int KIND, int D, int T>
__global__ __launch_bounds__(256) void kv_valu_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4;
  constexpr int BN = KVV_BN, RPT = KVV_RPT, DQ = DP / 4;
  __shared__ __attribute__((aligned(16))) float Xs[BN * DP];
  __shared__ __attribute__((aligned(16))) float Vs[BN * T];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);

  float xi[RPT][DP];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int i = min(rb * KVV_BM + r * 256 + tid, a.n - 1);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP + 4 * q);
      xi[r][4 * q + 0] = v[0]; xi[r][4 * q + 1] = v[1]; xi[r][4 * q + 2] = v[2]; xi[r][4 * q + 3] = v[3];
    }
  }
  float acc[RPT][T];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int c = 0; c < T; ++c) acc[r][c] = 0.f;

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    // stage x_j (BN points) and v_j (BN x T, interleaved [j][c])
    for (int idx = tid; idx < BN * DQ; idx += 256) {
      int j = j0 + idx / DQ;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * (idx % DQ));
      *reinterpret_cast<f32x4*>(&Xs[4 * idx]) = v;
    }
#pragma unroll
    for (int c = 0; c < T; ++c) {
      int j = j0 + tid;
      float v = 0.f;
      if (c < a.t && j < jend) v = a.Vt[(int64_t)c * a.ldv + j];
      Vs[tid * T + c] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < BN; ++jj) {
      float xj[DP];
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&Xs[jj * DP + 4 * q]);
        xj[4 * q + 0] = v[0]; xj[4 * q + 1] = v[1]; xj[4 * q + 2] = v[2]; xj[4 * q + 3] = v[3];
      }
      float vj[T];
#pragma unroll
      for (int c = 0; c < T; ++c) vj[c] = Vs[jj * T + c];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          float df = xi[r][k] - xj[k];
          sq = __builtin_fmaf(df, df, sq);
        }
        float kv = cov_from_sq<KIND>(sq);
#pragma unroll
        for (int c = 0; c < T; ++c) acc[r][c] = __builtin_fmaf(kv, vj[c], acc[r][c]);
      }
    }
  }
  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int i = rb * KVV_BM + r * 256 + tid;
    if (i < a.n) {
#pragma unroll
      for (int c = 0; c < T; ++c)
        if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[r][c];
    }
  }
}

}  // namespace gpamd
