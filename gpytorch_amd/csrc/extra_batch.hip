// Batches of SMALL independent Gaussian processes (the reference's batch mode, gpytorch/kernels/kernel.py:163-208 batch_shape,
// test/examples/test_batch_gp_regression.py): one launch generates the dense covariance matrices of ALL members, one launch reduces the
// bilinear derivative of all of them.  A member below settings.max_cholesky_size is factorised (batched Cholesky through torch), so per
// member the fused K*V kernels never run -- what is left is launch-bound: the member loop costs ~50 launches per member and
// evaluation, this file makes the count independent of the batch size.  blockIdx.z = member; member g owns rows [g n, (g + 1) n) of
// the prepared points, its own output scale, shape parameter and diagonal shift.
#include "../../include/gpamd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>

#include "common.hpp"

using namespace gpamd;
namespace gpamd {
extern thread_local char g_err[512];
}

namespace {

int failb(int code, const char* msg) {
  snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s", msg);
  return code;
}
int launch_okb(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

constexpr int GB_MAXDP = 16;   // prepared dimensions of the fused float32 path
constexpr int GB_ROWS = 32;    // rows of W one workgroup of the derivative kernel walks

// out[g][i][j] = scale[g] * k(X1p[g][i], X2p[g][j]) + (i == j ? dadd[g] : 0)
template <int KIND>
__global__ __launch_bounds__(256) void kernel_dense_batched_kernel(const float* __restrict__ X1p, int n, const float* __restrict__ X2p, int m,
                                                                   int DP, const float* __restrict__ kparam, const float* __restrict__ scale,
                                                                   const float* __restrict__ dadd, float* __restrict__ out, int64_t ldo) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y, g = blockIdx.z;
  if (j >= m) return;
  const float* a = X1p + ((int64_t)g * n + i) * DP;
  const float* b = X2p + ((int64_t)g * m + j) * DP;
  float sq = 0.f;
  for (int k = 0; k < DP; ++k) {
    const float df = a[k] - b[k];
    sq = __builtin_fmaf(df, df, sq);
  }
  float kv = (scale ? scale[g] : 1.f) * cov_from_sq<KIND>(sq, kparam ? kparam[g] : 0.f);
  if (dadd && i == j) kv += dadd[g];
  out[((int64_t)g * n + i) * ldo + j] = kv;
}

// G[g][0] += sum_ij W k,  G[g][1 + q] += sum_ij W dk/ds (z_iq - z_jq)^2,  G[g][1 + DP] += sum_ij W dk/dp   (W = W[g], z = prepared points of g)
template <int KIND>
__global__ __launch_bounds__(256) void kernel_grad_batched_kernel(const float* __restrict__ X1p, int n, const float* __restrict__ X2p, int m,
                                                                  int DP, const float* __restrict__ kparam, const float* __restrict__ W,
                                                                  int64_t ldw, double* __restrict__ G) {
  __shared__ double red[4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.z;
  const int r0 = blockIdx.y * GB_ROWS;
  const int r1 = min(n, r0 + GB_ROWS);
  const float p = kparam ? kparam[g] : 0.f;
  float zj[GB_MAXDP], aq[GB_MAXDP];
  float a0 = 0.f, ap = 0.f;
#pragma unroll
  for (int k = 0; k < GB_MAXDP; ++k) {
    aq[k] = 0.f;
    zj[k] = (k < DP && j < m) ? X2p[((int64_t)g * m + j) * DP + k] : 0.f;
  }
  if (j < m) {
    for (int r = r0; r < r1; ++r) {
      const float* a = X1p + ((int64_t)g * n + r) * DP;   // the same address in every lane: scalar loads
      float df2[GB_MAXDP];
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < GB_MAXDP; ++k) {
        const float df = k < DP ? a[k] - zj[k] : 0.f;
        df2[k] = df * df;
        sq += df2[k];
      }
      const float w = W[((int64_t)g * n + r) * ldw + j];
      const float kv = cov_from_sq<KIND>(sq, p);
      const float wdk = w * dcov_dsq<KIND>(sq, p);
      a0 = __builtin_fmaf(w, kv, a0);
      if constexpr (KIND == KIND_RQ) ap -= w * kv * 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + sq);   // dk/dp = -k ln(1 + s); v_log_f32 = log2
#pragma unroll
      for (int k = 0; k < GB_MAXDP; ++k) aq[k] = __builtin_fmaf(wdk, df2[k], aq[k]);
    }
  }
  double* Gg = G + (int64_t)g * (2 + DP);
  double s = block_sum_256((double)a0, red);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(Gg, s);
#pragma unroll
  for (int k = 0; k < GB_MAXDP; ++k) {
    if (k < DP) {   // (uniform)
      __syncthreads();
      s = block_sum_256((double)aq[k], red);
      if (threadIdx.x == 0 && s != 0.0) atomicAdd(Gg + 1 + k, s);
    }
  }
  if constexpr (KIND == KIND_RQ) {
    __syncthreads();
    s = block_sum_256((double)ap, red);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(Gg + 1 + DP, s);
  }
}

#define KIND_SWITCHB(kind, CALL)                                               \
  switch (kind) {                                                              \
    case GPAMD_RBF: { constexpr int KK = KIND_RBF; CALL; } break;              \
    case GPAMD_MATERN12: { constexpr int KK = KIND_MATERN12; CALL; } break;    \
    case GPAMD_MATERN32: { constexpr int KK = KIND_MATERN32; CALL; } break;    \
    case GPAMD_MATERN52: { constexpr int KK = KIND_MATERN52; CALL; } break;    \
    case GPAMD_RQ: { constexpr int KK = KIND_RQ; CALL; } break;                \
    default: return failb(GPAMD_EINVAL, "unknown kind");                       \
  }

}  // namespace

extern "C" {

int gpamd_kernel_dense_batched_f32(int kind, const float* kparam, const float* X1p, int n, const float* X2p, int m, int dp, int b,
                                   const float* scale, const float* dadd, float* out, int64_t ldo, void* stream) {
  if (n <= 0 || m <= 0 || b <= 0 || dp <= 0 || ldo < m) return failb(GPAMD_EINVAL, "kernel_dense_batched: bad shape");
  if (n > 65535 || b > 65535) return failb(GPAMD_EUNSUPPORTED, "kernel_dense_batched: n, b <= 65535");
  if (kind == GPAMD_RQ && !kparam) return failb(GPAMD_EINVAL, "kernel_dense_batched: the rational-quadratic family needs kparam[b]");
  dim3 grid((m + 255) / 256, n, b);
  KIND_SWITCHB(kind, hipLaunchKernelGGL((kernel_dense_batched_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, n, X2p, m, dp,
                                        kparam, scale, dadd, out, ldo));
  return launch_okb("kernel_dense_batched");
}

int gpamd_kernel_grad_batched_f32(int kind, const float* kparam, const float* X1p, int n, const float* X2p, int m, int dp, int b,
                                  const float* W, int64_t ldw, double* G, void* stream) {
  if (n <= 0 || m <= 0 || b <= 0 || dp <= 0 || dp > GB_MAXDP || ldw < m || !G) return failb(GPAMD_EINVAL, "kernel_grad_batched: bad shape (dp <= 16)");
  if ((n + GB_ROWS - 1) / GB_ROWS > 65535 || b > 65535) return failb(GPAMD_EUNSUPPORTED, "kernel_grad_batched: n <= 2097120, b <= 65535");
  if (kind == GPAMD_RQ && !kparam) return failb(GPAMD_EINVAL, "kernel_grad_batched: the rational-quadratic family needs kparam[b]");
  (void)hipMemsetAsync(G, 0, sizeof(double) * (size_t)b * (2 + dp), (hipStream_t)stream);
  dim3 grid((m + 255) / 256, (n + GB_ROWS - 1) / GB_ROWS, b);
  KIND_SWITCHB(kind, hipLaunchKernelGGL((kernel_grad_batched_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, n, X2p, m, dp,
                                        kparam, W, ldw, G));
  return launch_okb("kernel_grad_batched");
}

}  // extern "C"
