// Fused bilinear derivative:  G = sum_{i,j} W_ij * dK_ij/d(theta),   W = L^T R  (never formed in HBM)
//
// Replaces LinearOperator._bilinear_derivative(left_vecs, right_vecs) on the kernel operator
// (third-party linear_operator) and the dense kernel backward it drives:
// gpytorch/functions/rbf_covariance.py:26-29 and matern_covariance.py:53-56 multiply an n x n
// grad_output (= left @ right^T) with a SAVED n x n dK/dl; the chunked variant is
// gpytorch/lazy/lazy_evaluated_kernel_tensor.py:69-104.
//
// Here each wave forms 32x32 tiles of W^T on the matrix pipe,
//     D[j][i] = sum_c R[c][j] * L[c][i]       (A = R tile, B = L tile; both read straight from the
//                                              probe-major [t][ld] vectors, 128-B coalesced)
// and consumes them in registers: every lane evaluates k and dk/ds for its 16 (j, i) pairs and
// accumulates
//     G[0]     += W_ij * k(s_ij)                      (-> d/d outputscale)
//     G[1 + q] += W_ij * dk/ds(s_ij) * (z_iq - z_jq)^2   (-> d/d lengthscale_q; host applies -2*theta/l_q)
// Algorithmic work: 2 n m t flop on MFMA -- one K*V-equivalent for ALL hyper-parameters (ARD included),
// against (1 + d) K*V-equivalents for a per-parameter contraction.
#pragma once
#include "common.hpp"

namespace gpamd {

struct GradArgs {
  const float* X1;  // [n][DP]
  const float* X2;  // [m][DP]
  const float* Lt;  // [t][ldl]  left vectors  (index i, with X1)
  const float* Rt;  // [t][ldr]  right vectors (index j, with X2)
  int64_t ldl, ldr;
  int n, m, t;
  int S, jchunk, nrb;  // j split, chunk length (multiple of 64), row blocks of 128
  double* part;        // [nrb*S][1 + DP]
  const int* tiles = nullptr;   // far-pair culling (kv_cull.hpp): per unit, the starts of the 64-row j steps it visits (nullptr: all of them)
  int tpc1 = 0;
};

template <int KIND, int DP, int ISO>
__global__ __launch_bounds__(256) void kv_grad_kernel(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  constexpr int DQ = DP / 4;
  const int th = (a.t + 1) / 2;  // MFMA k-steps (2 probe columns each)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  float* Ls = dyn + (size_t)wave * (2 * th) * 32;              // [2*th][32] this wave's L tile
  float* Xj = dyn + (size_t)4 * (2 * th) * 32 + wave * 64 * DP;  // [64][DP] two j tiles
  __shared__ double red[4][1 + DP];

  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int i0 = rb * 128 + wave * 32;
  const int i = i0 + l31;

  // stage this wave's L tile: Ls[c][ic] = L[c][i0 + ic]  (zero beyond t / n)
  for (int c = h; c < 2 * th; c += 2) {
    float v = 0.f;
    if (c < a.t && i < a.n) v = a.Lt[(int64_t)c * a.ldl + i];
    Ls[c * 32 + l31] = v;
  }
  float xi[DP];
  {
    const int ic = min(i, a.n - 1);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)ic * DP + 4 * q);
      xi[4 * q + 0] = v[0]; xi[4 * q + 1] = v[1]; xi[4 * q + 2] = v[2]; xi[4 * q + 3] = v[3];
    }
  }
  double g[1 + DP];
#pragma unroll
  for (int q = 0; q <= DP; ++q) g[q] = 0.0;
  __builtin_amdgcn_wave_barrier();

  const int* tl = a.tiles ? a.tiles + (int64_t)unit * a.tpc1 : nullptr;
  for (int tk = 0, j0; (j0 = tile_start<64>(tl, jbeg, tk)) < jend; ++tk) {
    // stage x_j for the two 32-wide j tiles (wave-private LDS: in-order DS ops, no block barrier)
    {
      const int j = min(j0 + lane, a.m - 1);
#pragma unroll
      for (int q = 0; q < DQ; ++q)
        *reinterpret_cast<f32x4*>(&Xj[lane * DP + 4 * q]) = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * q);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int ja = j0 + l31, jb = j0 + 32 + l31;
    const bool va = ja < jend, vb = jb < jend;
#pragma unroll 4
    for (int c2 = 0; c2 < th; ++c2) {
      const int c = 2 * c2 + h;
      const bool vc = c < a.t;
      const float* rrow = a.Rt + (int64_t)c * a.ldr;
      float ra = (vc && va) ? rrow[ja] : 0.f;
      float rbv = (vc && vb) ? rrow[jb] : 0.f;
      float lb = Ls[c * 32 + l31];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra, lb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rbv, lb, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    float f[1 + DP];
#pragma unroll
    for (int q = 0; q <= DP; ++q) f[q] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jr = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float w = half ? acc1[r] : acc0[r];
        float df2[DP];
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Xj[jr * DP + 4 * q]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float df = xi[4 * q + e] - v[e];
            if constexpr (ISO) {
              sq = __builtin_fmaf(df, df, sq);
            } else {
              df2[4 * q + e] = df * df;
              sq += df2[4 * q + e];
            }
          }
        }
        const float kv = cov_from_sq<KIND>(sq);
        const float dk = dcov_dsq<KIND>(sq);
        f[0] = __builtin_fmaf(w, kv, f[0]);
        const float wd = w * dk;
        if constexpr (ISO) {
          // single lengthscale: sum_q (z_iq - z_jq)^2 = s, one fma instead of DP multiplies + DP fmas
          f[1] = __builtin_fmaf(wd, sq, f[1]);
        } else {
#pragma unroll
          for (int q = 0; q < DP; ++q) f[1 + q] = __builtin_fmaf(wd, df2[q], f[1 + q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q <= DP; ++q) g[q] += (double)f[q];
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int q = 0; q <= DP; ++q) {
    double v = wave_sum(g[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (tid <= DP) a.part[(int64_t)unit * (1 + DP) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// out[q] = sum_u part[u][q]   (1 block of 256 threads; fixed order -> reproducible)
__global__ __launch_bounds__(256) void grad_finalize_kernel(const double* __restrict__ part, int units, int nq,
                                                           float* __restrict__ out) {
  __shared__ double sm[4];
  for (int q = 0; q < nq; ++q) {
    double acc = 0.0;
    for (int u = threadIdx.x; u < units; u += 256) acc += part[(int64_t)u * nq + q];
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) out[q] = (float)acc;
    __syncthreads();
  }
}

}  // namespace gpamd
