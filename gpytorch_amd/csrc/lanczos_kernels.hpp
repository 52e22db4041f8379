// Lanczos vector kernels (probe-major rows, float32): the O(n k) work of one Lanczos step with full re-orthogonalisation.
//
// Replaces the torch GEMV / norm chain of linear_operator.utils.lanczos.lanczos_tridiag (third-party; algorithm restated in
// oracle/lanczos.py and SURVEY.md A.7; reached from gpytorch/models/exact_prediction_strategies.py:202,234-238,271) by five
// kernels whose scalars (alpha, beta, the projection coefficients, the "needs another pass" flag) stay on the device:
//   residual : r = w - beta_prev q_prev
//   project  : partial sums of  c_m = <Q[m], r>,  m < k      (one pass over the basis; wave-shuffle + LDS reduction)
//   coef     : c_m = sum of partials (fixed order), flag |= any |c_m| > tol     (all-reduced by the host when row-sharded)
//   subtract : r -= sum_m c_m Q[m]   and partial sums of |r|^2       (one pass over the basis)
//   normalize: beta = sqrt(|r|^2), q_next = r / beta, stop |= beta < 1e-6
// Reductions: float4 loads, wave-shuffle sums, one partial per workgroup, summed in a fixed order -> bitwise reproducible.
#pragma once
#include "cg_kernels.hpp"

namespace gpamd {

constexpr int LZ_MAXNB = 128;   // workgroups (partials) per reduction
constexpr int LZ_MAXK = 512;    // basis vectors per project / subtract launch

__global__ __launch_bounds__(256) void lz_residual_kernel(const float* __restrict__ w, const float* __restrict__ qprev,
                                                          const float* __restrict__ beta_prev, float* __restrict__ r, int n) {
  const float b = (qprev && beta_prev) ? *beta_prev : 0.f;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)gridDim.x * 256) {
    V4<float> x = ld4(w, i, n);
    if (qprev) {
      V4<float> p = ld4(qprev, i, n);
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[e] -= b * p.v[e];
    }
    st4(r, i, n, x);
  }
}

// part[m][blockIdx.x] = sum over this block's slice of Q[m][i] * r[i]      (grid.x = nb <= LZ_MAXNB, k <= LZ_MAXK)
__global__ __launch_bounds__(256) void lz_project_kernel(const float* __restrict__ Q, int64_t ldq, int k, const float* __restrict__ r,
                                                         int n, float* __restrict__ part) {
  __shared__ float wsum[4][LZ_MAXK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = 0; m < k; ++m) {
    const float* q = Q + (int64_t)m * ldq;
    float acc = 0.f;
    for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)gridDim.x * 256) {
      V4<float> x = ld4(q, i, n), y = ld4(r, i, n);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += x.v[e] * y.v[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) wsum[wave][m] = acc;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < k; m += 256)
    part[(int64_t)m * LZ_MAXNB + blockIdx.x] = wsum[0][m] + wsum[1][m] + wsum[2][m] + wsum[3][m];
}

// coef[m] = sum_b part[m][b]; flag[0] |= any |coef| > tol  (tol < 0: no flag).  One block per coefficient.
__global__ __launch_bounds__(64) void lz_coef_kernel(const float* __restrict__ part, int nb, float tol, float* __restrict__ coef,
                                                     int* __restrict__ flag) {
  const int m = blockIdx.x;
  float v = 0.f;
  for (int b = threadIdx.x; b < nb; b += 64) v += part[(int64_t)m * LZ_MAXNB + b];
  v = wave_sum(v);
  if (threadIdx.x == 0) {
    coef[m] = v;
    if (tol >= 0.f && fabsf(v) > tol) atomicOr(flag, 1);
  }
}

// r -= sum_m coef[m] Q[m];  part_rr[blockIdx.x] = this block's share of |r|^2
__global__ __launch_bounds__(256) void lz_subtract_kernel(const float* __restrict__ Q, int64_t ldq, int k, const float* __restrict__ coef,
                                                          float* __restrict__ r, int n, float* __restrict__ part_rr) {
  __shared__ float cs[LZ_MAXK];
  __shared__ float sm[4];
  for (int m = threadIdx.x; m < k; m += 256) cs[m] = coef[m];
  __syncthreads();
  float acc = 0.f;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)gridDim.x * 256) {
    V4<float> x = ld4(r, i, n);
    for (int m = 0; m < k; ++m) {
      V4<float> q = ld4(Q + (int64_t)m * ldq, i, n);
      const float c = cs[m];
#pragma unroll
      for (int e = 0; e < 4; ++e) x.v[e] -= c * q.v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += x.v[e] * x.v[e];
    st4(r, i, n, x);
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = acc;
}

// out = r / sqrt(rr[0]);  norm_out[0] = sqrt(rr[0]);  stop[0] |= norm < tiny       (rr: one finished sum, see lz_coef_kernel)
__global__ __launch_bounds__(256) void lz_normalize_kernel(const float* __restrict__ r, int n, const float* __restrict__ rr,
                                                           float* __restrict__ out, float* __restrict__ norm_out, float tiny,
                                                           int* __restrict__ stop) {
  const float nrm = sqrtf(fmaxf(*rr, 0.f));
  const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
  for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (int64_t)gridDim.x * 256) {
    V4<float> x = ld4(r, i, n);
#pragma unroll
    for (int e = 0; e < 4; ++e) x.v[e] *= inv;
    st4(out, i, n, x);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (norm_out) *norm_out = nrm;
    if (stop && nrm < tiny) atomicOr(stop, 1);
  }
}

}  // namespace gpamd

namespace gpamd {

// ---------------------------------------------------------------------------------------------------------------------------
// Preconditioner coefficients in mixed precision:  W[c][m] = sum_i R[c][i] * Q1[m][i]   (R float32 [t][ldr], Q1 float64 [k][ldq],
// float64 accumulation).  P^-1 R = (R - W Q1) / s2 needs the leading components of R to cancel to ~6 digits (linear_cg.py,
// Preconditioner.apply_), hence float64; rocBLAS' float64 GEMM for these tall-skinny shapes takes 80 ms for some (t, k)
// (profiles/r02_s7_precond_apply_rocblas_f64_shapes.txt: k = 15), 4x a whole K*V at n = 500 000.
// Work split: block b owns a slice of i; threads form a 16 x 16 grid over (column, basis vector) with a CT x MT register tile each;
// 32-element chunks of R and Q1 are staged in LDS.  Partials [b][t][k] are summed in a fixed order by pc_coef_sum_kernel.
constexpr int PC_CHUNK = 32;
constexpr int PC_MT = 8;      // 128 basis rows per tile (blockIdx.y of pc_coef_kernel / the tile loop of pc_apply_kernel)

// TQ / TR: types of the basis rows and of the projected rows (double / float: the preconditioner's Q1 against float32 residuals; float / float: the Lanczos
// basis of the block recurrences below and the Gram matrix of the pivoted-Cholesky factor; double / double: the second Cholesky-QR pass of the preconditioner).  blockIdx.y selects a
// tile of 128 basis rows, so k is unbounded for callers that launch a second grid dimension (the preconditioner launches one: k <= 128).
// MT: basis rows per thread (16 MT per blockIdx.y tile).  8 for long slices; 2 for SMALL n (round 6): at n = 36 584 the 8-row form is 143 workgroups of eight
// load -> barrier -> compute rounds each (59 us for 80 Mflop, 23 applies per protein-shaped closure) -- a quarter of the tile gives four times the
// workgroups with a quarter of the staging per round.  Every W[c][m] is the same sum in the same order whatever MT: bitwise equal results.
template <int CT, typename TQ = double, typename TR = float, int MT = 8>  // t <= 16 * CT
__global__ __launch_bounds__(256) void pc_coef_kernel(const TR* __restrict__ R, int64_t ldr, int t, const TQ* __restrict__ Q,
                                                      int64_t ldq, int k, int n, int slice, double* __restrict__ part) {
  __shared__ TR Rs[16 * CT][PC_CHUNK + 1];
  __shared__ TQ Qs[16 * MT][PC_CHUNK + 1];
  const int tid = threadIdx.x, tc = tid >> 4, tm = tid & 15;
  const int i0 = blockIdx.x * slice, i1 = min(n, i0 + slice);
  const int m0 = blockIdx.y * (16 * MT), kt = min(16 * MT, k - m0);   // this block's basis rows [m0, m0 + kt)
  double acc[CT][MT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = 0.0;
  for (int ib = i0; ib < i1; ib += PC_CHUNK) {
    __syncthreads();
    for (int e = tid; e < 16 * CT * PC_CHUNK; e += 256) {
      const int c = e / PC_CHUNK, ii = e % PC_CHUNK;
      Rs[c][ii] = (c < t && ib + ii < i1) ? R[(int64_t)c * ldr + ib + ii] : TR(0);
    }
    for (int e = tid; e < 16 * MT * PC_CHUNK; e += 256) {
      const int m = e / PC_CHUNK, ii = e % PC_CHUNK;
      Qs[m][ii] = (m < kt && ib + ii < i1) ? Q[(int64_t)(m0 + m) * ldq + ib + ii] : TQ(0);
    }
    __syncthreads();
#pragma unroll 4
    for (int ii = 0; ii < PC_CHUNK; ++ii) {
      double rv[CT], qv[MT];
#pragma unroll
      for (int a = 0; a < CT; ++a) rv[a] = (double)Rs[tc + 16 * a][ii];
#pragma unroll
      for (int b = 0; b < MT; ++b) qv[b] = (double)Qs[tm + 16 * b][ii];
#pragma unroll
      for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = fma(rv[a], qv[b], acc[a][b]);
    }
  }
  double* out = part + (int64_t)blockIdx.x * t * k;
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int c = tc + 16 * a, m = tm + 16 * b;
      if (c < t && m < kt) out[(int64_t)c * k + m0 + m] = acc[a][b];
    }
}

// W[e] = sum_b part[b][e]     (e < t * k).  PC_SUM_LANES consecutive lanes share an element: lane q takes the partials b = q, q + LANES, .. in order and the
// running sums are combined in a fixed order (bitwise reproducible); the loads of a wave cover 8 consecutive elements per b (64 contiguous
// bytes).  Round 6: with one thread walking all nb <= 256 partials of its element the kernel took 35 us whatever the size -- 1.3 ms of a
// protein-shaped closure's 23 preconditioner applies (profiles/r05_s6_workload_protein_kernel_stats.csv); a quarter of the dependent loads now.
constexpr int PC_SUM_LANES = 8;   // lanes per element (4 until the slices went down to 128 elements: up to 256 partials per element, 32 dependent loads per lane now)
__global__ void pc_coef_sum_kernel(const double* __restrict__ part, int nb, int tk, double* __restrict__ W) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = gid / PC_SUM_LANES, q = gid % PC_SUM_LANES;
  double s = 0.0;
  if (e < tk)
    for (int b = q; b < nb; b += PC_SUM_LANES) s += part[(int64_t)b * tk + e];
  // (all 64 lanes take part in the shuffles; lanes of elements past the end carry zeros)
#pragma unroll
  for (int w = 1; w < PC_SUM_LANES; w <<= 1) {
    const double o = __shfl_xor(s, w, 64);
    s = (q & w) ? o + s : s + o;        // both lanes of a pair form the same sum: lower lane's value first
  }
  if (e < tk && q == 0) W[e] = s;
}

// Second half of the preconditioner apply, fused:  Out[c][i] = (R[c][i] - sum_m W[c][m] Q1[m][i]) / sigma2  with the subtraction and
// the division in float64 (the cancellation that float32 cannot carry), R / Out float32, Q1 / W float64.  One workgroup = 256
// consecutive i for PA_CT columns: W of the column group sits in LDS (broadcast reads), every Q1 element is read once per column
// group and stays in a register across its PA_CT multiply-adds.  Replaces float64 copy of R + addmm + div + copy-back (four torch
// kernels and three [t][n] float64 temporaries).
constexpr int PA_CT = 16;
__global__ __launch_bounds__(256) void pc_apply_kernel(const float* __restrict__ R, int64_t ldr, int t, const double* __restrict__ Q,
                                                       int64_t ldq, int k, int n, const double* __restrict__ W,
                                                       const float* __restrict__ sigma2, float* __restrict__ Out, int64_t ldo) {
  __shared__ double Ws[PA_CT * 16 * PC_MT];   // [c][m] of the current 128-row tile of Q1 (round 6: any k, tile by tile; was k <= 128)
  const int c0 = blockIdx.y * PA_CT;
  const int nc = min(PA_CT, t - c0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  double acc[PA_CT];
#pragma unroll
  for (int c = 0; c < PA_CT; ++c) acc[c] = 0.0;
  for (int m0 = 0; m0 < k; m0 += 16 * PC_MT) {
    const int kt = min(16 * PC_MT, k - m0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * kt; e += 256) Ws[(e / kt) * (16 * PC_MT) + (e % kt)] = W[(int64_t)(c0 + e / kt) * k + m0 + (e % kt)];
    __syncthreads();
    if (i < n) {
      // four basis rows per round (eight measured the same at small n and 5 % slower at n = 500 000, 65 columns): their loads are issued together (round 6 -- one load, then its 16 multiply-adds, row after row, was a chain of
      // k memory latencies per wave: 30 us at n = 36 584 where one wave per SIMD has nothing to hide them behind); the sums keep their order
      int m = 0;
      for (; m + 4 <= kt; m += 4) {
        double q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = Q[(int64_t)(m0 + m + u) * ldq + i];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < PA_CT; ++c) acc[c] = fma(Ws[c * (16 * PC_MT) + m + u], q[u], acc[c]);
      }
      for (; m < kt; ++m) {
        const double q = Q[(int64_t)(m0 + m) * ldq + i];
#pragma unroll
        for (int c = 0; c < PA_CT; ++c) acc[c] = fma(Ws[c * (16 * PC_MT) + m], q, acc[c]);
      }
    }
  }
  if (i >= n) return;
  const double inv = 1.0 / (double)sigma2[0];
#pragma unroll
  for (int c = 0; c < PA_CT; ++c)
    if (c < nc) Out[(int64_t)(c0 + c) * ldo + i] = (float)(((double)R[(int64_t)(c0 + c) * ldr + i] - acc[c]) * inv);
}

// ---- BLOCK Lanczos (lanczos.py block_lanczos_steps: the LOVE cache on a block Krylov space; b <= 32 rows per block, any number k of basis rows) ----
// project : W[c][m] = <R[c], Q[m]>      = pc_coef_kernel<1, float> over k tiles of 128 (double accumulation, partials summed in a fixed order)
// subtract: R[c][i] -= sum_m W[c][m] Q[m][i]      (in place, double arithmetic; W of a 128-row tile of the basis in LDS, every Q element read once)
// transform: R[r][i] = sum_c M[r][c] R[c][i]      (in place; M: b x b double -- the inverse Cholesky factor of the block's Gram matrix: Cholesky-QR)
constexpr int LZB_MAXB = 32;   // (16 until round 6: the auto block size at n >= 262 144 is 32)
__global__ __launch_bounds__(256) void lzb_subtract_kernel(const float* __restrict__ Q, int64_t ldq, int k, const double* __restrict__ W, float* R,
                                                          int64_t ldr, int b, int n) {
  __shared__ double Ws[LZB_MAXB * 128];   // [c][m] of the current basis tile
  const int i = blockIdx.x * 256 + threadIdx.x;
  double acc[LZB_MAXB];
#pragma unroll
  for (int c = 0; c < LZB_MAXB; ++c) acc[c] = 0.0;
  for (int m0 = 0; m0 < k; m0 += 128) {
    const int kt = min(128, k - m0);
    __syncthreads();
    for (int e = threadIdx.x; e < b * kt; e += 256) Ws[(e / kt) * 128 + (e % kt)] = W[(int64_t)(e / kt) * k + m0 + (e % kt)];
    for (int e = threadIdx.x; e < (LZB_MAXB - b) * 128; e += 256) Ws[b * 128 + e] = 0.0;
    __syncthreads();
    if (i < n) {
      for (int m = 0; m < kt; ++m) {
        const double q = (double)Q[(int64_t)(m0 + m) * ldq + i];
#pragma unroll
        for (int c = 0; c < LZB_MAXB; ++c) acc[c] = fma(Ws[c * 128 + m], q, acc[c]);
      }
    }
  }
  if (i < n) {
#pragma unroll
    for (int c = 0; c < LZB_MAXB; ++c)
      if (c < b) R[(int64_t)c * ldr + i] = (float)((double)R[(int64_t)c * ldr + i] - acc[c]);
  }
}

__global__ __launch_bounds__(256) void lzb_transform_kernel(const double* __restrict__ M, float* R, int64_t ldr, int b, int n) {
  __shared__ double Ms[LZB_MAXB * LZB_MAXB];
  for (int e = threadIdx.x; e < b * b; e += 256) Ms[e] = M[e];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v[LZB_MAXB];
#pragma unroll
  for (int c = 0; c < LZB_MAXB; ++c) v[c] = c < b ? (double)R[(int64_t)c * ldr + i] : 0.0;
#pragma unroll
  for (int r = 0; r < LZB_MAXB; ++r) {
    if (r < b) {
      double o = 0.0;
#pragma unroll
      for (int c = 0; c < LZB_MAXB; ++c)
        if (c < b) o = fma(Ms[r * b + c], v[c], o);
      R[(int64_t)r * ldr + i] = (float)o;
    }
  }
}

// ---- multi-shift MINRES (contour-integral quadrature, gpytorch/__init__.py:252-278 -> linear_operator.utils.minres): the vector part of one
// iteration for ALL shifts in one pass.  Per (shift q, column c):  d = (v_c - delta d1 - eps d2) / gamma  (written over d2: the caller swaps
// the roles of the two direction buffers),  x += tau d.  v: [t][ld]; d1, d2, x: [Q][t][ld]; coef: [4][Q][t] = delta | eps | 1 / gamma | tau
// (device, produced by the Givens recurrences on [Q, t] scalars).  HBM-bound: 4 reads + 2 writes of Q t n floats, one launch instead of the
// ~8 elementwise torch passes (and as many [Q, t, n] temporaries) it replaces.
__global__ __launch_bounds__(256) void msminres_update_kernel(const float* __restrict__ v, const float* __restrict__ d1, float* __restrict__ d2,
                                                             float* __restrict__ x, const float* __restrict__ coef, int Q, int t, int n,
                                                             int64_t ld) {
  const int c = blockIdx.y, q = blockIdx.z;
  const int qt = Q * t, k = q * t + c;
  const float delta = coef[k], eps = coef[qt + k], ginv = coef[2 * qt + k], tau = coef[3 * qt + k];
  const float* vc = v + (int64_t)c * ld;
  const int64_t off = (int64_t)k * ld;
  for (int i = 4 * (blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * 256 * gridDim.x) {
    if (i + 4 <= n) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(vc + i);
      const f32x4 a = *reinterpret_cast<const f32x4*>(d1 + off + i);
      const f32x4 b = *reinterpret_cast<const f32x4*>(d2 + off + i);
      f32x4 xx = *reinterpret_cast<const f32x4*>(x + off + i);
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d[e] = (vv[e] - delta * a[e] - eps * b[e]) * ginv;
        xx[e] = __builtin_fmaf(tau, d[e], xx[e]);
      }
      *reinterpret_cast<f32x4*>(d2 + off + i) = d;
      *reinterpret_cast<f32x4*>(x + off + i) = xx;
    } else {
      for (int e = 0; i + e < n; ++e) {
        const float d = (vc[i + e] - delta * d1[off + i + e] - eps * d2[off + i + e]) * ginv;
        d2[off + i + e] = d;
        x[off + i + e] = __builtin_fmaf(tau, d, x[off + i + e]);
      }
    }
  }
}

}  // namespace gpamd
