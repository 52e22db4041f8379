// Device-resident modified batched CG (mBCG) vector kernels -- probe-major layout.
//
// Replaces the per-iteration chain of ~15 separate torch launches in
// linear_operator.utils.linear_cg (third-party; algorithm restated in oracle/linear_cg.py and
// SURVEY.md A.2; reached from gpytorch/distributions/multivariate_normal.py:249 and
// gpytorch/models/exact_prediction_strategies.py:286,444) by 4 fused launches per iteration
//   kv_mfma (partials) -> cg_reduce_q -> cg_update_xr -> cg_update_d (+ cg_check)
// with every per-column scalar (alpha, beta, rho, ||r||, convergence masks) living on the device:
// the host enqueues iterations without synchronising and polls a single `done` word.
//
// All vectors are [t][ld] ("one probe / right-hand side per row", contiguous over the n data
// points), so each column's inner products are plain contiguous reductions: float4 loads,
// wave-shuffle (DPP) reduction, one partial per workgroup, summed in a fixed order by the consumer
// (bitwise run-to-run reproducible; no atomics).
#pragma once
#include "common.hpp"

namespace gpamd {

constexpr int CG_MAXNB = 256;  // max workgroups (partials) per column

template <typename T>
struct V4 {
  T v[4];
};

template <typename T>
__device__ __forceinline__ V4<T> ld4(const T* __restrict__ p, int64_t i, int n) {
  V4<T> r;
  if (i + 4 <= n) {
    if constexpr (sizeof(T) == 4) {
      f32x4 x = *reinterpret_cast<const f32x4*>(p + i);
      r.v[0] = x[0]; r.v[1] = x[1]; r.v[2] = x[2]; r.v[3] = x[3];
    } else {
      const double2* q = reinterpret_cast<const double2*>(p + i);
      double2 a = q[0], b = q[1];
      r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = (i + e < n) ? p[i + e] : T(0);
  }
  return r;
}

template <typename T>
__device__ __forceinline__ void st4(T* __restrict__ p, int64_t i, int n, const V4<T>& r) {
  if (i + 4 <= n) {
    if constexpr (sizeof(T) == 4) {
      f32x4 x = {r.v[0], r.v[1], r.v[2], r.v[3]};
      *reinterpret_cast<f32x4*>(p + i) = x;
    } else {
      double2* q = reinterpret_cast<double2*>(p + i);
      q[0] = make_double2(r.v[0], r.v[1]);
      q[1] = make_double2(r.v[2], r.v[3]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) p[i + e] = r.v[e];
  }
}

// sum of the NB (<=256) per-workgroup partials of one column; blockDim.x == 256
template <typename T>
__device__ __forceinline__ T sum_partials(const T* __restrict__ part, int nb, T* smem4) {
  T v = (threadIdx.x < nb) ? part[threadIdx.x] : T(0);
  return block_sum_256(v, smem4);
}

// ---------------------------------------------------------------------------------------------
// generic column dot: part[c][b] = sum_i A[c][i] * B[c][i]
template <typename T>
__global__ __launch_bounds__(256) void coldot_kernel(const T* __restrict__ A, const T* __restrict__ B, int64_t ld,
                                                     int n, T* __restrict__ part, const int* __restrict__ done) {
  __shared__ T sm[4];
  if (done && *done) return;
  const int c = blockIdx.y, nb = gridDim.x;
  const T* a = A + (int64_t)c * ld;
  const T* b = B + (int64_t)c * ld;
  T acc = 0;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)nb * 256) {
    V4<T> x = ld4(a, i, n), y = ld4(b, i, n);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += x.v[e] * y.v[e];
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
}

// finish a column reduction: out[c] = sum_b part[c][b]   (grid = t blocks)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partials_kernel(const T* __restrict__ part, int nb, T* __restrict__ out) {
  __shared__ T sm[4];
  T s = sum_partials(part + (int64_t)blockIdx.x * CG_MAXNB, nb, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// Out[c][i] = scale * sum_s P[s][c][i] + (dscale + dvec[i]) * Vd[c][i]   (the K_hat = theta*K + noise epilogue:
// gpytorch/kernels/scale_kernel.py:117-118 and gpytorch/likelihoods/gaussian_likelihood.py:117-121; dvec is the
// heteroskedastic / per-task noise diagonal of FixedNoiseGaussianLikelihood and MultitaskGaussianLikelihood)
// optionally also part[c][b] = sum_i Dv[c][i] * Out[c][i]  (CG denominator d^T K_hat d).
template <typename T, bool WITH_DOT>
__global__ __launch_bounds__(256) void kv_reduce_kernel(const T* __restrict__ P, int S, int64_t pstride, int64_t ldp,
                                                        const T* __restrict__ scale, const T* __restrict__ dscale,
                                                        const T* __restrict__ dvec, const T* __restrict__ Vd,
                                                        int64_t ldd, T* __restrict__ Out,
                                                        int64_t ldo, int n, T* __restrict__ part,
                                                        const int* __restrict__ done) {
  __shared__ T sm[4];
  if (done && *done) return;
  const int c = blockIdx.y, nb = gridDim.x;
  const T sc = scale ? *scale : T(1);
  const T ds = dscale ? *dscale : T(0);
  T acc = 0;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)nb * 256) {
    V4<T> q = ld4(P + (int64_t)c * ldp, i, n);
    for (int s = 1; s < S; ++s) {
      V4<T> p = ld4(P + (int64_t)s * pstride + (int64_t)c * ldp, i, n);
#pragma unroll
      for (int e = 0; e < 4; ++e) q.v[e] += p.v[e];
    }
    V4<T> d = {{T(0), T(0), T(0), T(0)}};
    if (Vd) d = ld4(Vd + (int64_t)c * ldd, i, n);
    V4<T> dv = {{T(0), T(0), T(0), T(0)}};
    if (dvec) dv = ld4(dvec, i, n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q.v[e] *= sc;
      if (Vd) q.v[e] += (ds + dv.v[e]) * d.v[e];
      if (WITH_DOT) acc += d.v[e] * q.v[e];
    }
    st4(Out + (int64_t)c * ldo, i, n, q);
  }
  if constexpr (WITH_DOT) {
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) part[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// mBCG state (device arrays of length t unless noted)
template <typename T>
struct CgState {
  T* X; T* R; T* D; T* Q; T* Z;   // [t][ld]; Z == R when no preconditioner
  int64_t ld;
  int n, t, nb;
  T* bnorm;        // [t] column norms of the rhs (1 for zero columns)
  int* zero_rhs;   // [t]
  int* converged;  // [t]
  T* rho;          // [2][t] double-buffered r^T z
  T* rnorm;        // [t]
  T* part_a;       // [t][CG_MAXNB]  d^T q partials / rhs norm partials
  T* part_rz;      // [t][CG_MAXNB]
  T* part_rr;      // [t][CG_MAXNB]
  T* alpha_hist;   // [hist_len][t]
  T* beta_hist;    // [hist_len][t]
  int hist_len;
  T* stats;        // [4]: sum of residual norms, column count, (spare), (spare)
  int* done;       // [2]: done flag (1 converged, 2 NaN), iterations performed
  T eps, stop_updating_after;
};

// R = B / ||B||, X = 0, D = R (when no preconditioner), partials of R.R.   part_a holds ||B||^2 partials.
template <typename T>
__global__ __launch_bounds__(256) void cg_init_kernel(CgState<T> st, const T* __restrict__ B, int64_t ldb, int copy_d) {
  __shared__ T sm[4];
  const int c = blockIdx.y;
  T nrm2 = sum_partials(st.part_a + (int64_t)c * CG_MAXNB, st.nb, sm);
  T bn = sqrt(nrm2);
  const bool zero = bn < st.eps;
  if (zero) bn = T(1);
  T acc = 0;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < st.n; i += 4 * (int64_t)st.nb * 256) {
    V4<T> b = ld4(B + (int64_t)c * ldb, i, st.n);
    V4<T> z;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b.v[e] = b.v[e] / bn;
      acc += b.v[e] * b.v[e];
      z.v[e] = T(0);
    }
    st4(st.R + (int64_t)c * st.ld, i, st.n, b);
    st4(st.X + (int64_t)c * st.ld, i, st.n, z);
    if (copy_d) st4(st.D + (int64_t)c * st.ld, i, st.n, b);
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    st.part_rr[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
    if (copy_d) st.part_rz[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
    if (blockIdx.x == 0) {
      st.bnorm[c] = bn;
      st.zero_rhs[c] = zero ? 1 : 0;
    }
  }
}

// rho[0][c] = sum part_rz ; rnorm/converged from part_rr    (grid = t blocks)
template <typename T>
__global__ __launch_bounds__(256) void cg_begin_kernel(CgState<T> st) {
  __shared__ T sm[4];
  const int c = blockIdx.x;
  T rz = sum_partials(st.part_rz + (int64_t)c * CG_MAXNB, st.nb, sm);
  T rr = sum_partials(st.part_rr + (int64_t)c * CG_MAXNB, st.nb, sm);
  if (threadIdx.x == 0) {
    st.rho[c] = rz;
    T rn = sqrt(rr);
    st.rnorm[c] = rn;
    st.converged[c] = rn < st.stop_updating_after ? 1 : 0;
    if (c == 0) { st.done[0] = 0; st.done[1] = 0; }
  }
}

// alpha = rho / (d^T q) with the reference's masks; X += alpha D; R -= alpha Q; partials of R.R
template <typename T>
__global__ __launch_bounds__(256) void cg_update_xr_kernel(CgState<T> st, int k, int same_z) {
  __shared__ T sm[4];
  if (*st.done) return;
  if (k < 0) k = st.done[1];   // replayed from a captured graph: the iteration index lives on the device (cg_stop advances it)
  const int c = blockIdx.y;
  T den = sum_partials(st.part_a + (int64_t)c * CG_MAXNB, st.nb, sm);
  const bool bad = den < st.eps;
  if (bad) den = T(1);
  T alpha = st.rho[(k & 1) * st.t + c] / den;
  if (bad || st.converged[c]) alpha = T(0);
  T acc = 0;
  T* X = st.X + (int64_t)c * st.ld;
  T* R = st.R + (int64_t)c * st.ld;
  const T* D = st.D + (int64_t)c * st.ld;
  const T* Q = st.Q + (int64_t)c * st.ld;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < st.n; i += 4 * (int64_t)st.nb * 256) {
    V4<T> x = ld4(X, i, st.n), r = ld4(R, i, st.n), d = ld4(D, i, st.n), q = ld4(Q, i, st.n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.v[e] = r.v[e] - alpha * q.v[e];
      x.v[e] = x.v[e] + alpha * d.v[e];
      acc += r.v[e] * r.v[e];
    }
    st4(X, i, st.n, x);
    st4(R, i, st.n, r);
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    st.part_rr[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
    if (same_z) st.part_rz[(int64_t)c * CG_MAXNB + blockIdx.x] = acc;
    if (blockIdx.x == 0 && k < st.hist_len) st.alpha_hist[(int64_t)k * st.t + c] = alpha;
  }
}

// beta = rho_new / rho_old (masked); D = Z + beta D; publishes rho_new, ||r||, convergence mask
template <typename T>
__global__ __launch_bounds__(256) void cg_update_d_kernel(CgState<T> st, int k) {
  __shared__ T sm[4];
  if (*st.done) return;
  if (k < 0) k = st.done[1];
  const int c = blockIdx.y;
  T rho_new = sum_partials(st.part_rz + (int64_t)c * CG_MAXNB, st.nb, sm);
  T rho_old = st.rho[(k & 1) * st.t + c];
  const bool bad = rho_old < st.eps;
  if (bad) rho_old = T(1);
  T beta = rho_new / rho_old;
  if (bad) beta = T(0);
  T* D = st.D + (int64_t)c * st.ld;
  const T* Z = st.Z + (int64_t)c * st.ld;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < st.n; i += 4 * (int64_t)st.nb * 256) {
    V4<T> d = ld4(D, i, st.n), z = ld4(Z, i, st.n);
#pragma unroll
    for (int e = 0; e < 4; ++e) d.v[e] = z.v[e] + beta * d.v[e];
    st4(D, i, st.n, d);
  }
  if (blockIdx.x == 0) {
    T rr = sum_partials(st.part_rr + (int64_t)c * CG_MAXNB, st.nb, sm);
    if (threadIdx.x == 0) {
      st.rho[((k + 1) & 1) * st.t + c] = rho_new;
      T rn = st.zero_rhs[c] ? T(0) : sqrt(rr);
      st.rnorm[c] = rn;
      st.converged[c] = rn < st.stop_updating_after ? 1 : 0;
      if (k < st.hist_len) st.beta_hist[(int64_t)k * st.t + c] = beta;
    }
  }
}

// stats[0] = sum_c rnorm[c], stats[1] = t     (1 block; all-reduced over ranks by the host when sharded)
template <typename T>
__global__ __launch_bounds__(256) void cg_stats_kernel(CgState<T> st) {
  __shared__ T sm[4];
  if (*st.done) return;
  T acc = 0;
  for (int c = threadIdx.x; c < st.t; c += 256) acc += st.rnorm[c];
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    st.stats[0] = acc;
    st.stats[1] = T(st.t);
  }
}

// the reference's stopping rule (linear_cg: k >= min(10, max_iter-1) and mean residual < tol and
// not (n_tridiag and k < min(n_tridiag_iter, max_iter-1)))
template <typename T>
__global__ void cg_stop_kernel(CgState<T> st, int k, int min_iter, int tridiag_floor, T tol) {
  if (threadIdx.x != 0 || *st.done) return;
  if (k < 0) k = st.done[1];
  T mean = st.stats[0] / st.stats[1];
  if (!(mean == mean)) {
    st.done[0] = 2;
    st.done[1] = k + 1;
    return;
  }
  st.done[1] = k + 1;
  if (k >= min_iter && mean < tol && !(k < tridiag_floor)) st.done[0] = 1;
}

// X *= ||B||
template <typename T>
__global__ __launch_bounds__(256) void cg_finish_kernel(CgState<T> st) {
  const int c = blockIdx.y;
  const T bn = st.bnorm[c];
  T* X = st.X + (int64_t)c * st.ld;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < st.n; i += 4 * (int64_t)st.nb * 256) {
    V4<T> x = ld4(X, i, st.n);
#pragma unroll
    for (int e = 0; e < 4; ++e) x.v[e] *= bn;
    st4(X, i, st.n, x);
  }
}

}  // namespace gpamd
