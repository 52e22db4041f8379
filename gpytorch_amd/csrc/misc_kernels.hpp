// Point preparation, explicit kernel rows / dense tiles / diagonals, and the pivoted-Cholesky
// step kernels (fp32).
#pragma once
#include "common.hpp"

namespace gpamd {

// Xp[i][k] = (X[i][k] - shift[k]) * coef / ls[k]   for k < d,  0 for d <= k < DP
// (the x1.div(lengthscale) of gpytorch/kernels/rbf_kernel.py:78-79 / keops/rbf_kernel.py:45-46 and the
// mean-centring of gpytorch/kernels/matern_kernel.py:94-97, folded with the exp2 / sqrt(2 nu) constants)
__global__ void prep_points_kernel(const float* __restrict__ X, int n, int d, int64_t ldx,
                                   const float* __restrict__ ls, int nls, const float* __restrict__ shift,
                                   float coef, float* __restrict__ Xp, int DP) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * DP) return;
  int i = idx / DP, k = idx - (int64_t)i * DP;
  float v = 0.f;
  if (k < d) {
    float l = ls[nls == 1 ? 0 : k];
    float sh = shift ? shift[k] : 0.f;
    v = (X[(int64_t)i * ldx + k] - sh) * (coef / l);
  }
  Xp[idx] = v;
}

template <int KIND>
__device__ __forceinline__ float cov_pair(const float* __restrict__ a, const float* __restrict__ b, int DP, float kparam = 0.f) {
  float sq = 0.f;
  for (int k = 0; k < DP; ++k) {
    float df = a[k] - b[k];
    sq = __builtin_fmaf(df, df, sq);
  }
  return cov_from_sq<KIND>(sq, kparam);
}

// out[r][j] = scale * k(X1p[rows[r]], X2p[j])   -- explicit rows (LinearOperator._getitem row fetch used by
// pivoted Cholesky: linear_operator functions/_pivoted_cholesky.py)
template <int KIND>
__global__ void kernel_rows_kernel(const float* __restrict__ X1p, const int64_t* __restrict__ rows, int nrows,
                                   const float* __restrict__ X2p, int m, int DP, const float* __restrict__ scale,
                                   float* __restrict__ out, int64_t ldo, float kparam) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (j >= m) return;
  const float* a = X1p + rows[r] * DP;
  float kv = cov_pair<KIND>(a, X2p + (int64_t)j * DP, DP, kparam);
  out[(int64_t)r * ldo + j] = (scale ? *scale : 1.f) * kv;
}

// out[i][j] = scale * k(X1p[i], X2p[j])   (to_dense; small problems, Cholesky fallback, tests)
template <int KIND>
__global__ void kernel_dense_kernel(const float* __restrict__ X1p, int n, const float* __restrict__ X2p, int m, int DP,
                                    const float* __restrict__ scale, float* __restrict__ out, int64_t ldo, float kparam) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= m || i >= n) return;
  float kv = cov_pair<KIND>(X1p + (int64_t)i * DP, X2p + (int64_t)j * DP, DP, kparam);
  out[(int64_t)i * ldo + j] = (scale ? *scale : 1.f) * kv;
}

// out[i] = scale * k(X1p[i], X2p[i])
template <int KIND>
__global__ void kernel_diag_kernel(const float* __restrict__ X1p, const float* __restrict__ X2p, int n, int DP,
                                   const float* __restrict__ scale, float* __restrict__ out, float kparam) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = (scale ? *scale : 1.f) * cov_pair<KIND>(X1p + (int64_t)i * DP, X2p + (int64_t)i * DP, DP, kparam);
}

// ---------------------------------------------------------------------------------------------
// Pivoted Cholesky (wrapper gpytorch/__init__.py:146-173).  Data stay in the original
// index order (L is rank x n row-major, written exactly like the reference's scatter by pi);
// pivoted entries are parked at -inf in `dwork` so the arg-max skips them.  The reference's arg-max
// runs over the PERMUTED order pi[m:], so exact ties (very common in fp32: every point far from all
// pivots still has d_i == theta) resolve to the lowest POSITION in the current permutation; `perm` /
// `pos` track that permutation (one swap per step) so ties resolve identically here.
struct PcState {
  float* dwork;      // [n] running Schur-complement diagonal; -inf once pivoted
  float* L;          // [rank][ldl]
  int64_t ldl;
  int n, rank;
  int64_t* pivots;   // [rank]
  float* scal;       // [4]: orig_error, current error (l1 of live diag / orig), pivot value
  int* ctl;          // [2]: m (steps done), stop flag
  int* perm;         // [n] perm[position] = index
  int* pos;          // [n] pos[index] = position
  float tol;
  float kparam;      // covariance shape parameter (RQ: alpha)
};

// one workgroup of 1024 threads: arg-max of the live diagonal (+ its l1 norm -> error test)
__global__ __launch_bounds__(1024) void pc_pivot_kernel(PcState st, int m) {
  __shared__ float sval[16];
  __shared__ int sidx[16];
  __shared__ float ssum[16];
  if (st.ctl[1]) return;
  const int tid = threadIdx.x;
  float best = -INFINITY, sum = 0.f;
  int bi = 0x7fffffff;  // best = (value, position in the permutation); ties -> lowest position
  for (int i = tid; i < st.n; i += 1024) {
    float v = st.dwork[i];
    if (v > -INFINITY) {
      sum += fabsf(v);
      int ps = st.pos[i];
      if (v > best || (v == best && ps < bi)) { best = v; bi = ps; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) { sval[tid >> 6] = best; sidx[tid >> 6] = bi; ssum[tid >> 6] = sum; }
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) {
      tot += ssum[w];
      if (sval[w] > best || (sval[w] == best && sidx[w] < bi)) { best = sval[w]; bi = sidx[w]; }
    }
    if (m == 0) st.scal[0] = best;  // orig_error = max diag
    float err = tot / st.scal[0];
    st.scal[1] = err;
    // reference loop condition: m == 0 or (m < max_iter and max(errors) > tol)
    if (m > 0 && !(err > st.tol)) { st.ctl[1] = 1; return; }
    // swap positions m and bi of the permutation (reference: pi_m <-> pi_i)
    const int p = st.perm[bi], q = st.perm[m];
    st.perm[m] = p; st.perm[bi] = q;
    st.pos[p] = m; st.pos[q] = bi;
    st.pivots[m] = p;
    st.scal[2] = best;
    st.ctl[0] = m + 1;
  }
}

__global__ void pc_init_perm_kernel(int* perm, int* pos, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { perm[i] = i; pos[i] = i; }
}

// L[m][i] = (K[p][i] - sum_{q<m} L[q][p] L[q][i]) / sqrt(d_p) for live i; L[m][p] = sqrt(d_p); d[i] -= L[m][i]^2
template <int KIND>
__global__ __launch_bounds__(256) void pc_update_kernel(PcState st, int m, const float* __restrict__ Xp, int DP,
                                                        const float* __restrict__ scale) {
  __shared__ float lp[128];  // L[q][p], q < m  (rank <= 128 per launch design)
  if (st.ctl[1] || st.ctl[0] != m + 1) return;
  const int64_t p = st.pivots[m];
  for (int q = threadIdx.x; q < m; q += blockDim.x) lp[q] = st.L[(int64_t)q * st.ldl + p];
  __syncthreads();
  const float piv = sqrtf(st.scal[2]);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.n) return;
  float d = st.dwork[i];
  float out = 0.f;
  if (i == p) {
    out = piv;
    st.dwork[i] = -INFINITY;
  } else if (d > -INFINITY) {
    float v = (scale ? *scale : 1.f) * cov_pair<KIND>(Xp + p * DP, Xp + (int64_t)i * DP, DP, st.kparam);
    for (int q = 0; q < m; ++q) v -= lp[q] * st.L[(int64_t)q * st.ldl + i];
    v = v / piv;
    out = v;
    st.dwork[i] = d - v * v;
  }
  st.L[(int64_t)m * st.ldl + i] = out;
}

}  // namespace gpamd
