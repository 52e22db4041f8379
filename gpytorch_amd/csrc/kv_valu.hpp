// Small right-hand-side count (t <= 8) fused covariance MVM on the vector ALU.
//
// With fewer than ~16 columns the contraction cannot feed the matrix pipe: the path is bound by
// kernel GENERATION (2*D + T VALU ops and one v_exp_f32 [+ v_sqrt_f32] per (i, j) pair), so the whole pair
// pipeline is written on PACKED f32 math: each lane owns 2*RP rows held as RP float2 row pairs, x_j / v_j are
// LDS broadcast reads shared by the whole wave, and differences, squares and the t products issue as
// v_pk_add_f32 / v_pk_fma_f32 (two pairs per instruction; the matrix pipe is idle here, so packed VALU runs at
// its full rate).  Instruction count per pair at D = 3, T = 1: 4.75 (was 7.9 unpacked).
// Used for the predictive-mean solve (t = 1, gpytorch/models/exact_prediction_strategies.py:286), Lanczos
// steps (t = 1, :271) and K_*X @ mean_cache (:396) -- the cold posterior is a sequence of ~450 such products.
//
// Same probe-major layout and split-j partial-slab convention as kv_mfma.hpp.
#pragma once
#include "common.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KVV_BN = 256;   // j-tile
constexpr int KVV_RP = 2;     // float2 row pairs per thread (4 rows)
constexpr int KVV_BM = 256 * 2 * KVV_RP;

// two covariance values at once; transcendentals stay scalar (no packed form), the polynomial part packs
template <int KIND>
__device__ __forceinline__ f32x2 cov_from_sq2(f32x2 s, float p = 0.f) {
  if constexpr (KIND == KIND_RQ) {
    f32x2 o;
    o.x = cov_from_sq<KIND_RQ>(s.x, p);
    o.y = cov_from_sq<KIND_RQ>(s.y, p);
    return o;
  } else if constexpr (KIND == KIND_RBF) {
    f32x2 o;
    o.x = __builtin_amdgcn_exp2f(-s.x);
    o.y = __builtin_amdgcn_exp2f(-s.y);
    return o;
  } else {
    f32x2 r;
    r.x = __builtin_amdgcn_sqrtf(s.x);
    r.y = __builtin_amdgcn_sqrtf(s.y);
    const f32x2 rl = r * (-LOG2E);
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(rl.x);
    e.y = __builtin_amdgcn_exp2f(rl.y);
    if constexpr (KIND == KIND_MATERN12) return e;
    if constexpr (KIND == KIND_MATERN32) return (r + 1.0f) * e;
    return __builtin_elementwise_fma(s, (f32x2)(1.0f / 3.0f), r + 1.0f) * e;
  }
}

template <int KIND, int D, int T>
__global__ __launch_bounds__(256) void kv_valu_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4;
  constexpr int BN = KVV_BN, RP = KVV_RP, DQ = DP / 4;
  __shared__ __attribute__((aligned(16))) float Xs[BN * DP];
  __shared__ __attribute__((aligned(16))) float Vs[BN * T];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);

  f32x2 xi[RP][D];  // row pair p = rows (2p, 2p+1) * 256 + tid of this block
#pragma unroll
  for (int p = 0; p < RP; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = min(rb * KVV_BM + (2 * p + h) * 256 + tid, a.n - 1);
#pragma unroll
      for (int k = 0; k < D; ++k) xi[p][k][h] = a.X1[(int64_t)i * DP + k];
    }
  f32x2 acc[RP][T];
#pragma unroll
  for (int p = 0; p < RP; ++p)
#pragma unroll
    for (int c = 0; c < T; ++c) acc[p][c] = (f32x2)(0.f);

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
      for (int p = 0; p < RP; ++p) {
        f32x2 sq = (f32x2)(0.f);
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const f32x2 df = xi[p][k] - (f32x2)(xj[k]);
          sq = __builtin_elementwise_fma(df, df, sq);
        }
        const f32x2 kv = cov_from_sq2<KIND>(sq, a.kparam);
#pragma unroll
        for (int c = 0; c < T; ++c) acc[p][c] = __builtin_elementwise_fma(kv, (f32x2)(vj[c]), acc[p][c]);
      }
    }
  }
  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int p = 0; p < RP; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = rb * KVV_BM + (2 * p + h) * 256 + tid;
      if (i < a.n) {
#pragma unroll
        for (int c = 0; c < T; ++c)
          if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[p][c][h];
      }
    }
}

}  // namespace gpamd
