// float64 entry points (include/gpamd.h, "float64" section).
//
// The reference honours float64 inputs (gpytorch/test/*: gradcheck and double-precision model tests).  This file
// provides the float64 building blocks of the same path: point preparation, explicit kernel entries (rows / dense
// tiles / diagonal, float64 exp / sqrt), the column inner products, the K_hat epilogue and the device-resident mBCG
// vector kernels (cg_kernels.hpp is templated on the scalar type).  The fused float32 MFMA K*V kernels have no
// float64 twin yet: in float64 the product K @ V is formed from HIP-generated dense row blocks of K times V with
// rocBLAS DGEMM (gpytorch_amd/backend.py::kv_chunked -- the reference's own chunked strategy,
// gpytorch/lazy/lazy_evaluated_kernel_tensor.py:245-275, run on the device), which is HBM-bound (8 n m bytes per product).
#include "../../include/gpamd.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "cg_kernels.hpp"
#include "kv_f64.hpp"

using namespace gpamd;
namespace gpamd {
extern thread_local char g_err[512];
}

namespace {

int fail64(int code, const char* msg) {
  snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s", msg);
  return code;
}
int launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
double prep_coef64(int kind, double kparam) {
  switch (kind) {
    case GPAMD_RQ: return 1.0 / sqrt(2.0 * kparam);   // (1 + |x - x'|^2 / (2 alpha l^2))^-alpha = (1 + |z - z'|^2)^-alpha
    case GPAMD_RBF: return sqrt(0.5 * 1.4426950408889634);
    case GPAMD_MATERN12: return 1.0;
    case GPAMD_MATERN32: return sqrt(3.0);
    case GPAMD_MATERN52: return sqrt(5.0);
  }
  return 0.0;
}
unsigned col_blocks64(int n) {
  long nb = ((long)n + 1023) / 1024;
  if (nb < 1) nb = 1;
  if (nb > CG_MAXNB) nb = CG_MAXNB;
  return (unsigned)nb;
}

__global__ void prep_points_f64_kernel(const double* __restrict__ X, int n, int d, int64_t ldx, const double* __restrict__ ls,
                                       int nls, const double* __restrict__ shift, double coef, double* __restrict__ Xp, int DP) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * DP) return;
  int i = idx / DP, k = idx - (int64_t)i * DP;
  double v = 0.0;
  if (k < d) v = (X[(int64_t)i * ldx + k] - (shift ? shift[k] : 0.0)) * (coef / ls[nls == 1 ? 0 : k]);
  Xp[idx] = v;
}

template <int KIND>
__device__ __forceinline__ double cov_pair64(const double* __restrict__ a, const double* __restrict__ b, int DP, double p) {
  double sq = 0.0;
  for (int k = 0; k < DP; ++k) {
    double df = a[k] - b[k];
    sq = fma(df, df, sq);
  }
  return cov_from_sq_f64<KIND>(sq, p);
}

// out[r][j] = scale * k(X1p[row(r)], X2p[j]);  rows == nullptr: row(r) = r0 + r  (dense row block)
template <int KIND>
__global__ void kernel_rows_f64_kernel(const double* __restrict__ X1p, const int64_t* __restrict__ rows, int64_t r0,
                                       const double* __restrict__ X2p, int m, int DP, const double* __restrict__ scale,
                                       double* __restrict__ out, int64_t ldo, double p) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (j >= m) return;
  const int64_t i = rows ? rows[r] : r0 + r;
  out[(int64_t)r * ldo + j] = (scale ? *scale : 1.0) * cov_pair64<KIND>(X1p + i * DP, X2p + (int64_t)j * DP, DP, p);
}

template <int KIND>
__global__ void kernel_diag_f64_kernel(const double* __restrict__ X1p, const double* __restrict__ X2p, int n, int DP,
                                       const double* __restrict__ scale, double* __restrict__ out, double p) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = (scale ? *scale : 1.0) * cov_pair64<KIND>(X1p + (int64_t)i * DP, X2p + (int64_t)i * DP, DP, p);
}

// Generic-path bilinear derivative, one row block:  W[r][j] (r < nb, j < m) holds left^T right on entry and
// W * dk/ds on exit (s = squared prepared distance);  acc[0] += sum_rj W[r][j] * k(x1[r0 + r], x2[j]) and, for a family with a
// shape parameter p (RQ: alpha), acc[1] += sum_rj W[r][j] * dk/dp at fixed s.
// The per-dimension sums  sum_rj (W dk/ds)_rj (z_rq - z_jq)^2  are then three GEMM-shaped reductions on the host side
// (backend.py::kv_grad_generic) -- no per-dimension register arrays, so any input dimension works.
template <int KIND, typename T>
__global__ __launch_bounds__(256) void grad_block_kernel(const T* __restrict__ X1p, int64_t r0, const T* __restrict__ X2p, int m,
                                                         int DP, T* __restrict__ W, int64_t ldw, double* __restrict__ acc, T p) {
  __shared__ double red[4];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  double part = 0.0, ppart = 0.0;
  if (j < m) {
    const T* a = X1p + (r0 + r) * DP;
    const T* b = X2p + (int64_t)j * DP;
    T sq = 0;
    for (int k = 0; k < DP; ++k) {
      T df = a[k] - b[k];
      sq += df * df;
    }
    const T w = W[(int64_t)r * ldw + j];
    const T kv = cov_any<KIND>(sq, p);
    part = (double)w * (double)kv;
    if constexpr (KIND == KIND_RQ) ppart = -(double)w * (double)kv * log1p((double)sq);   // k = (1 + s)^-p  ->  dk/dp = -k ln(1 + s)
    W[(int64_t)r * ldw + j] = w * dcov_any<KIND>(sq, p);
  }
  part = block_sum_256(part, red);
  if (threadIdx.x == 0 && part != 0.0) atomicAdd(acc, part);
  if constexpr (KIND == KIND_RQ) {
    __syncthreads();
    ppart = block_sum_256(ppart, red);
    if (threadIdx.x == 0 && ppart != 0.0) atomicAdd(acc + 1, ppart);
  }
}

template <typename T>
int grad_block_launch(int kind, double kparam, const T* X1p, int64_t row0, int nrows, const T* X2p, int m, int dp, T* W, int64_t ldw,
                      double* acc, void* stream) {
  if (nrows <= 0 || m <= 0 || nrows > 65535 || dp <= 0 || ldw < m) return fail64(GPAMD_EINVAL, "kernel_grad_block: bad shape (1 <= nrows <= 65535)");
  dim3 grid((m + 255) / 256, nrows);
  switch (kind) {
#define GB(KE, KK) \
  case KE: hipLaunchKernelGGL((grad_block_kernel<KK, T>), grid, dim3(256), 0, (hipStream_t)stream, X1p, row0, X2p, m, dp, W, ldw, acc, (T)kparam); break;
    GB(GPAMD_RBF, KIND_RBF) GB(GPAMD_MATERN12, KIND_MATERN12) GB(GPAMD_MATERN32, KIND_MATERN32) GB(GPAMD_MATERN52, KIND_MATERN52) GB(GPAMD_RQ, KIND_RQ)
#undef GB
    default: return fail64(GPAMD_EINVAL, "unknown kind");
  }
  return launch_ok("kernel_grad_block");
}

#define KIND_SWITCH64(kind, CALL)                                              \
  switch (kind) {                                                              \
    case GPAMD_RBF: { constexpr int KK = KIND_RBF; CALL; } break;              \
    case GPAMD_MATERN12: { constexpr int KK = KIND_MATERN12; CALL; } break;    \
    case GPAMD_MATERN32: { constexpr int KK = KIND_MATERN32; CALL; } break;    \
    case GPAMD_MATERN52: { constexpr int KK = KIND_MATERN52; CALL; } break;    \
    case GPAMD_RQ: { constexpr int KK = KIND_RQ; CALL; } break;                \
    default: return fail64(GPAMD_EINVAL, "unknown kind");                      \
  }

}  // namespace

namespace {

// column tiles of 16: 1 (t <= 16), 4 (t <= 64), 5 (t <= 80); wider right-hand sides go in groups of 80
int kv64_ct_for(int t) { return t <= 16 ? 1 : (t <= 64 ? 4 : 5); }
int kv64_group(int t, int g0) { return (t - g0) <= 80 ? (t - g0) : 80; }
int kv64_bm(int ct, int dp) { return 4 * kv64_ni_for(ct, dp) * 16; }

template <int KIND, int DP>
const void* kv64_ptr_ct(int ct) {
  switch (ct) {
    case 1: return reinterpret_cast<const void*>(&kv_f64_kernel<KIND, DP, 1>);
    case 4: return reinterpret_cast<const void*>(&kv_f64_kernel<KIND, DP, 4>);
    case 5: return reinterpret_cast<const void*>(&kv_f64_kernel<KIND, DP, 5>);
  }
  return nullptr;
}
template <int KIND>
const void* kv64_ptr_dp(int dp, int ct) {
  switch (dp) {
    case 4: return kv64_ptr_ct<KIND, 4>(ct);
    case 8: return kv64_ptr_ct<KIND, 8>(ct);
    case 12: return kv64_ptr_ct<KIND, 12>(ct);
    case 16: return kv64_ptr_ct<KIND, 16>(ct);
  }
  return nullptr;
}
// t <= 4: the VALU-contraction kernel (kv_f64.hpp kv_f64v_kernel), one or four accumulator columns
template <int KIND>
const void* kv64v_ptr_dp(int dp, int tv) {
#define L(DPV) \
  case DPV: return tv == 1 ? reinterpret_cast<const void*>(&kv_f64v_kernel<KIND, DPV, 1>) : reinterpret_cast<const void*>(&kv_f64v_kernel<KIND, DPV, 4>);
  switch (dp) { L(4) L(8) L(12) L(16) }
#undef L
  return nullptr;
}
const void* kv64v_ptr(int kind, int dp, int tv) {
  switch (kind) {
    case GPAMD_RBF: return kv64v_ptr_dp<KIND_RBF>(dp, tv);
    case GPAMD_MATERN12: return kv64v_ptr_dp<KIND_MATERN12>(dp, tv);
    case GPAMD_MATERN32: return kv64v_ptr_dp<KIND_MATERN32>(dp, tv);
    case GPAMD_MATERN52: return kv64v_ptr_dp<KIND_MATERN52>(dp, tv);
    case GPAMD_RQ: return kv64v_ptr_dp<KIND_RQ>(dp, tv);
  }
  return nullptr;
}
const void* kv64_ptr(int kind, int dp, int ct) {
  switch (kind) {
    case GPAMD_RBF: return kv64_ptr_dp<KIND_RBF>(dp, ct);
    case GPAMD_MATERN12: return kv64_ptr_dp<KIND_MATERN12>(dp, ct);
    case GPAMD_MATERN32: return kv64_ptr_dp<KIND_MATERN32>(dp, ct);
    case GPAMD_MATERN52: return kv64_ptr_dp<KIND_MATERN52>(dp, ct);
    case GPAMD_RQ: return kv64_ptr_dp<KIND_RQ>(dp, ct);
  }
  return nullptr;
}

}  // namespace

struct gpamd_cg64 {
  CgState<double> st;
};

extern "C" {

int gpamd_prep_points_f64(int kind, double kparam, const double* X, int n, int d, int64_t ldx, const double* ls, int nls,
                          const double* shift, double* Xp, int dp, void* stream) {
  if (kind < 0 || kind > GPAMD_RQ || n <= 0 || d <= 0 || dp < d || (nls != 1 && nls != d)) return fail64(GPAMD_EINVAL, "prep_points_f64: bad shape");
  if (kind == GPAMD_RQ && !(kparam > 0.0)) return fail64(GPAMD_EINVAL, "prep_points_f64: the rational-quadratic shape parameter alpha must be positive");
  long total = (long)n * dp;
  hipLaunchKernelGGL(prep_points_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, n, d,
                     ldx, ls, nls, shift, prep_coef64(kind, kparam), Xp, dp);
  return launch_ok("prep_points_f64");
}

int gpamd_kernel_rows_f64(int kind, double kparam, const double* X1p, const int64_t* rows, int64_t row0, int nrows, const double* X2p, int m,
                          int dp, const double* scale, double* out, int64_t ldo, void* stream) {
  if (nrows <= 0 || m <= 0 || nrows > 65535) return fail64(GPAMD_EINVAL, "kernel_rows_f64: bad shape (1 <= nrows <= 65535)");
  dim3 grid((m + 255) / 256, nrows);
  KIND_SWITCH64(kind, hipLaunchKernelGGL((kernel_rows_f64_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, rows, row0,
                                         X2p, m, dp, scale, out, ldo, kparam));
  return launch_ok("kernel_rows_f64");
}

int gpamd_kernel_diag_f64(int kind, double kparam, const double* X1p, const double* X2p, int n, int dp, const double* scale, double* out,
                          void* stream) {
  if (n <= 0) return fail64(GPAMD_EINVAL, "kernel_diag_f64: bad shape");
  KIND_SWITCH64(kind, hipLaunchKernelGGL((kernel_diag_f64_kernel<KK>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                         X1p, X2p, n, dp, scale, out, kparam));
  return launch_ok("kernel_diag_f64");
}

int gpamd_kernel_grad_block_f32(int kind, double kparam, const float* X1p, int64_t row0, int nrows, const float* X2p, int m, int dp, float* W,
                                int64_t ldw, double* acc, void* stream) {
  return grad_block_launch<float>(kind, kparam, X1p, row0, nrows, X2p, m, dp, W, ldw, acc, stream);
}
int gpamd_kernel_grad_block_f64(int kind, double kparam, const double* X1p, int64_t row0, int nrows, const double* X2p, int m, int dp, double* W,
                                int64_t ldw, double* acc, void* stream) {
  return grad_block_launch<double>(kind, kparam, X1p, row0, nrows, X2p, m, dp, W, ldw, acc, stream);
}

int gpamd_kv_plan_f64(int n, int m, int dp, int t, int64_t ldo, int* S, int* jchunk, int64_t* workspace_doubles) {
  if (n <= 0 || m <= 0 || t <= 0 || !S || !jchunk || !workspace_doubles || ldo < n) return fail64(GPAMD_EINVAL, "kv_plan_f64: bad arguments");
  if (dp != 4 && dp != 8 && dp != 12 && dp != 16) return fail64(GPAMD_EUNSUPPORTED, "kv_f64: fused float64 kernel needs d <= 16 (generic path otherwise)");
  const int ct = kv64_ct_for(t > 80 ? 80 : t);
  const int nrb = (n + kv64_bm(ct, dp) - 1) / kv64_bm(ct, dp);
  int s = (3 * 3 * 256 + nrb - 1) / nrb;  // ~3 rounds of 3 resident workgroups on 256 CUs
  const int smax = m / (4 * KV64_BN) > 0 ? m / (4 * KV64_BN) : 1;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  const int jc = ((m + s - 1) / s + KV64_BN - 1) / KV64_BN * KV64_BN;
  *jchunk = jc;
  *S = (m + jc - 1) / jc;
  *workspace_doubles = (int64_t)(*S) * t * ldo;
  return 0;
}

int gpamd_kv_partials_f64(int kind, double kparam, const double* X1p, int n, const double* X2p, int m, int dp, const double* Vt, int64_t ldv,
                          int t, double* P, int64_t ldo, int S, int jchunk, const int* done, void* stream) {
  if (kind < 0 || kind > GPAMD_RQ || n <= 0 || m <= 0 || t <= 0 || S <= 0 || jchunk <= 0 || jchunk % KV64_BN || ldv < m || ldo < n)
    return fail64(GPAMD_EINVAL, "kv_partials_f64: bad shape");
  if (dp != 4 && dp != 8 && dp != 12 && dp != 16) return fail64(GPAMD_EUNSUPPORTED, "kv_f64: fused float64 kernel needs d <= 16 (generic path otherwise)");
  for (int g0 = 0; g0 < t;) {
    const int tg = kv64_group(t, g0);
    const int ct = kv64_ct_for(tg);
    KvArgs64 a;
    a.X1 = X1p; a.X2 = X2p;
    a.Vt = Vt + (int64_t)g0 * ldv;
    a.P = P + (int64_t)g0 * ldo;
    a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)t * ldo;
    a.n = n; a.m = m; a.t = tg;
    a.S = S; a.jchunk = jchunk;
    a.nrb = (n + kv64_bm(ct, dp) - 1) / kv64_bm(ct, dp);
    a.done = done;
    a.kparam = kparam;
    const void* fn = tg <= 4 ? kv64v_ptr(kind, dp, tg == 1 ? 1 : 4) : kv64_ptr(kind, dp, ct);   // (same row block as ct = 1: the plan does not change)
    if (!fn) return fail64(GPAMD_EUNSUPPORTED, "kv_f64: no kernel variant");
    void* kargs[] = {(void*)&a};
    (void)hipLaunchKernel(fn, dim3((unsigned)a.nrb * (unsigned)S), dim3(256), kargs, 0, (hipStream_t)stream);
    g0 += tg;
  }
  return launch_ok("kv_partials_f64");
}

int gpamd_coldot_f64(const double* A, const double* B, int64_t ld, int n, int t, double* out, double* scratch, void* stream) {
  if (n <= 0 || t <= 0 || ld % 4) return fail64(GPAMD_EINVAL, "coldot_f64: bad shape");
  unsigned nb = col_blocks64(n);
  hipLaunchKernelGGL((coldot_kernel<double>), dim3(nb, t), dim3(256), 0, (hipStream_t)stream, A, B, ld, n, scratch, (const int*)nullptr);
  hipLaunchKernelGGL((colsum_partials_kernel<double>), dim3(t), dim3(256), 0, (hipStream_t)stream, scratch, (int)nb, out);
  return launch_ok("coldot_f64");
}

int gpamd_kv_reduce_f64(const double* P, int S, int64_t ldp, int t, int n, const double* scale, const double* dscale,
                        const double* dvec, const double* Vd, int64_t ldd, double* Out, int64_t ldo, const int* done,
                        void* stream) {
  if (S <= 0 || t <= 0 || n <= 0 || ldp % 4 || ldo % 4 || (Vd && ldd % 4)) return fail64(GPAMD_EINVAL, "kv_reduce_f64: bad shape");
  hipLaunchKernelGGL((kv_reduce_kernel<double, false>), dim3(col_blocks64(n), t), dim3(256), 0, (hipStream_t)stream, P, S,
                     (int64_t)t * ldp, ldp, scale, dscale, dvec, Vd, ldd, Out, ldo, n, (double*)nullptr, done);
  return launch_ok("kv_reduce_f64");
}

int64_t gpamd_cg64_fscratch_elems(int t, int hist_len) {
  return (int64_t)4 * t + 4 + (int64_t)2 * hist_len * t + (int64_t)3 * t * CG_MAXNB;  // same layout as the float32 solver
}

gpamd_cg64_t* gpamd_cg64_create(int n, int t, int64_t ld, double* X, double* R, double* D, double* Q, double* Z,
                                double* fscratch, int* iscratch, int hist_len, double eps, double stop_updating_after) {
  if (n <= 0 || t <= 0 || ld % 4 || ld < n || hist_len < 0) {
    fail64(GPAMD_EINVAL, "cg64_create: bad shape");
    return nullptr;
  }
  gpamd_cg64* h = new gpamd_cg64;
  CgState<double>& s = h->st;
  s.X = X; s.R = R; s.D = D; s.Q = Q; s.Z = Z;
  s.ld = ld; s.n = n; s.t = t; s.nb = (int)col_blocks64(n);
  double* f = fscratch;
  s.bnorm = f; f += t;
  s.rnorm = f; f += t;
  s.rho = f; f += 2 * t;
  s.stats = f; f += 4;
  s.alpha_hist = f; f += (int64_t)hist_len * t;
  s.beta_hist = f; f += (int64_t)hist_len * t;
  s.part_a = f; f += (int64_t)t * CG_MAXNB;
  s.part_rz = f; f += (int64_t)t * CG_MAXNB;
  s.part_rr = f;
  s.hist_len = hist_len;
  s.zero_rhs = iscratch;
  s.converged = iscratch + t;
  s.done = iscratch + 2 * t;
  s.eps = eps;
  s.stop_updating_after = stop_updating_after;
  return h;
}
void gpamd_cg64_destroy(gpamd_cg64_t* h) { delete h; }

int gpamd_cg64_init(gpamd_cg64_t* h, const double* B, int64_t ldb, int have_precond, void* stream) {
  if (!h || ldb % 4) return fail64(GPAMD_EINVAL, "cg64_init: bad arguments");
  CgState<double>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(s.nb, s.t);
  (void)hipMemsetAsync(s.done, 0, 2 * sizeof(int), st);
  hipLaunchKernelGGL((coldot_kernel<double>), grid, dim3(256), 0, st, B, B, ldb, s.n, s.part_a, (const int*)nullptr);
  hipLaunchKernelGGL((cg_init_kernel<double>), grid, dim3(256), 0, st, s, B, ldb, have_precond ? 0 : 1);
  if (!have_precond) hipLaunchKernelGGL((cg_begin_kernel<double>), dim3(s.t), dim3(256), 0, st, s);
  return launch_ok("cg64_init");
}

int gpamd_cg64_begin(gpamd_cg64_t* h, void* stream) {
  if (!h) return fail64(GPAMD_EINVAL, "cg64_begin: null handle");
  CgState<double>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((coldot_kernel<double>), dim3(s.nb, s.t), dim3(256), 0, st, s.R, s.Z, s.ld, s.n, s.part_rz, (const int*)nullptr);
  hipLaunchKernelGGL((cg_begin_kernel<double>), dim3(s.t), dim3(256), 0, st, s);
  return launch_ok("cg64_begin");
}

int gpamd_cg64_reduce_q(gpamd_cg64_t* h, const double* P, int S, int64_t ldp, const double* scale, const double* dscale,
                        const double* dvec, void* stream) {
  if (!h || S <= 0 || ldp % 4) return fail64(GPAMD_EINVAL, "cg64_reduce_q: bad arguments");
  CgState<double>& s = h->st;
  hipLaunchKernelGGL((kv_reduce_kernel<double, true>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, P, S, (int64_t)s.t * ldp,
                     ldp, scale, dscale, dvec, s.D, s.ld, s.Q, s.ld, s.n, s.part_a, s.done);
  return launch_ok("cg64_reduce_q");
}

int gpamd_cg64_update_xr(gpamd_cg64_t* h, int k, void* stream) {
  if (!h) return fail64(GPAMD_EINVAL, "cg64_update_xr: null handle");
  CgState<double>& s = h->st;
  hipLaunchKernelGGL((cg_update_xr_kernel<double>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s, k, s.Z == s.R ? 1 : 0);
  return launch_ok("cg64_update_xr");
}

int gpamd_cg64_update_d(gpamd_cg64_t* h, int k, void* stream) {
  if (!h) return fail64(GPAMD_EINVAL, "cg64_update_d: null handle");
  CgState<double>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  if (s.Z != s.R)
    hipLaunchKernelGGL((coldot_kernel<double>), dim3(s.nb, s.t), dim3(256), 0, st, s.R, s.Z, s.ld, s.n, s.part_rz, (const int*)s.done);
  hipLaunchKernelGGL((cg_update_d_kernel<double>), dim3(s.nb, s.t), dim3(256), 0, st, s, k);
  hipLaunchKernelGGL((cg_stats_kernel<double>), dim3(1), dim3(256), 0, st, s);
  return launch_ok("cg64_update_d");
}

int gpamd_cg64_stop(gpamd_cg64_t* h, int k, int min_iter, int tridiag_floor, double tol, void* stream) {
  if (!h) return fail64(GPAMD_EINVAL, "cg64_stop: null handle");
  hipLaunchKernelGGL((cg_stop_kernel<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, h->st, k, min_iter, tridiag_floor, tol);
  return launch_ok("cg64_stop");
}

int gpamd_cg64_finish(gpamd_cg64_t* h, void* stream) {
  if (!h) return fail64(GPAMD_EINVAL, "cg64_finish: null handle");
  CgState<double>& s = h->st;
  hipLaunchKernelGGL((cg_finish_kernel<double>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s);
  return launch_ok("cg64_finish");
}

}  // extern "C"
