// extern "C" entry points of the Lanczos vector kernels (lanczos_kernels.hpp); see include/gpamd.h.
#include "../../include/gpamd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>

#include "lanczos_kernels.hpp"

using namespace gpamd;
namespace gpamd {
extern thread_local char g_err[512];
}

namespace {
int lz_fail(const char* msg) {
  snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s", msg);
  return GPAMD_EINVAL;
}
int lz_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
unsigned lz_blocks(int n) {
  long nb = ((long)n + 1023) / 1024;
  if (nb < 1) nb = 1;
  if (nb > LZ_MAXNB) nb = LZ_MAXNB;
  return (unsigned)nb;
}
}  // namespace

extern "C" {

int gpamd_lanczos_num_partials(int n) { return n > 0 ? (int)lz_blocks(n) : 0; }
int gpamd_lanczos_partial_stride(void) { return LZ_MAXNB; }

int gpamd_lanczos_residual_f32(const float* w, const float* q_prev, const float* beta_prev, float* r, int n, void* stream) {
  if (!w || !r || n <= 0) return lz_fail("lanczos_residual: bad arguments");
  hipLaunchKernelGGL(lz_residual_kernel, dim3(lz_blocks(n)), dim3(256), 0, (hipStream_t)stream, w, q_prev, beta_prev, r, n);
  return lz_check("lanczos_residual");
}

int gpamd_lanczos_project_f32(const float* Q, int64_t ldq, int k, const float* r, int n, float* part, void* stream) {
  if (!Q || !r || !part || n <= 0 || k <= 0 || k > LZ_MAXK || ldq < n || ldq % 4) return lz_fail("lanczos_project: bad arguments (k <= 512, ldq % 4 == 0)");
  hipLaunchKernelGGL(lz_project_kernel, dim3(lz_blocks(n)), dim3(256), 0, (hipStream_t)stream, Q, ldq, k, r, n, part);
  return lz_check("lanczos_project");
}

int gpamd_lanczos_coef_f32(const float* part, int k, int nb, float tol, float* coef, int* flag, void* stream) {
  if (!part || !coef || k <= 0 || nb <= 0 || nb > LZ_MAXNB || (tol >= 0.f && !flag)) return lz_fail("lanczos_coef: bad arguments");
  hipLaunchKernelGGL(lz_coef_kernel, dim3(k), dim3(64), 0, (hipStream_t)stream, part, nb, tol, coef, flag);
  return lz_check("lanczos_coef");
}

int gpamd_lanczos_subtract_f32(const float* Q, int64_t ldq, int k, const float* coef, float* r, int n, float* part_rr, void* stream) {
  if (!Q || !r || !coef || !part_rr || n <= 0 || k <= 0 || k > LZ_MAXK || ldq < n || ldq % 4) return lz_fail("lanczos_subtract: bad arguments");
  hipLaunchKernelGGL(lz_subtract_kernel, dim3(lz_blocks(n)), dim3(256), 0, (hipStream_t)stream, Q, ldq, k, coef, r, n, part_rr);
  return lz_check("lanczos_subtract");
}

int gpamd_lanczos_normalize_f32(const float* r, int n, const float* rr, float* out, float* norm_out, float tiny, int* stop, void* stream) {
  if (!r || !rr || !out || n <= 0) return lz_fail("lanczos_normalize: bad arguments");
  hipLaunchKernelGGL(lz_normalize_kernel, dim3(lz_blocks(n)), dim3(256), 0, (hipStream_t)stream, r, n, rr, out, norm_out, tiny, stop);
  return lz_check("lanczos_normalize");
}

// ---- preconditioner coefficients W = R Q1^T in mixed precision (lanczos_kernels.hpp: pc_coef_kernel) ----
// slices of the row index (one workgroup and one partial each): >= 128 elements per workgroup, at most 256 slices.  (Round 6: 256 elements until then --
// at n = 36 584 each of 143 workgroups ran eight load -> barrier -> compute rounds, latency end to end; five rounds on 229 workgroups now.)
static long pc_slices(int n) {
  long nb = ((long)n + 127) / 128;
  return nb > 256 ? 256 : (nb < 1 ? 1 : nb);
}
constexpr long PC_SMALL_GRID = 1L << 30;   // (was 384 = 1.5 workgroups per CU in the 128-row-tile form) the 32-row-tile form for EVERY <= 16-column projection: 1.6 - 1.9x at n = 36 584 .. 500 000, ranks 15 / 100 (profiles/r06_s31_precond_apply_timing.json): 256 workgroups of 62 latency-bound rounds at n = 500 000 were no better filled than 143 of eight
int64_t gpamd_precond_coef_workspace_doubles(int n, int t, int k) {
  if (n <= 0 || t <= 0 || k <= 0) return 0;
  return (int64_t)pc_slices(n) * t * k;
}

int gpamd_precond_coef_f32f64(const float* R, int64_t ldr, int t, const double* Q, int64_t ldq, int k, int n, double* W,
                              double* workspace, int64_t workspace_doubles, void* stream) {
  if (!R || !Q || !W || !workspace || n <= 0 || t <= 0 || k <= 0 || ldr < n || ldq < n) return lz_fail("precond_coef: bad arguments");
  if (k > 512) return lz_fail("precond_coef: rank > 512");
  const unsigned ktiles = (unsigned)((k + 16 * PC_MT - 1) / (16 * PC_MT));   // 128 basis rows per blockIdx.y
  long nb = pc_slices(n);
  if (workspace_doubles < (int64_t)nb * t * k) return GPAMD_EWORKSPACE;
  const int slice = (int)(((long)n + nb - 1) / nb + PC_CHUNK - 1) / PC_CHUNK * PC_CHUNK;
  nb = ((long)n + slice - 1) / slice;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < t; c0 += 80) {   // column groups of <= 80 (16 x 5 register tile)
    const int tg = t - c0 < 80 ? t - c0 : 80;
    double* part = workspace;             // reused per group: the sum kernel of a group runs before the next group's partials
    if (tg <= 16 && nb * ktiles < PC_SMALL_GRID)
      hipLaunchKernelGGL((pc_coef_kernel<1, double, float, 2>), dim3((unsigned)nb, (unsigned)((k + 31) / 32)), dim3(256), 0, st, R + (int64_t)c0 * ldr, ldr, tg, Q, ldq, k, n, slice, part);
    else if (tg <= 16)
      hipLaunchKernelGGL((pc_coef_kernel<1>), dim3((unsigned)nb, ktiles), dim3(256), 0, st, R + (int64_t)c0 * ldr, ldr, tg, Q, ldq, k, n, slice, part);
    else
      hipLaunchKernelGGL((pc_coef_kernel<5>), dim3((unsigned)nb, ktiles), dim3(256), 0, st, R + (int64_t)c0 * ldr, ldr, tg, Q, ldq, k, n, slice, part);
    hipLaunchKernelGGL(pc_coef_sum_kernel, dim3((PC_SUM_LANES * tg * k + 255) / 256), dim3(256), 0, st, (const double*)part, (int)nb, tg * k, W + (int64_t)c0 * k);
  }
  return lz_check("precond_coef");
}

int gpamd_precond_apply_f32f64(const float* R, int64_t ldr, int t, const double* Q, int64_t ldq, int k, int n, const double* W,
                               const float* sigma2, float* Out, int64_t ldo, void* stream) {
  if (!R || !Q || !W || !sigma2 || !Out || n <= 0 || t <= 0 || k <= 0 || ldr < n || ldq < n || ldo < n)
    return lz_fail("precond_apply: bad arguments");
  if (k > 512) return lz_fail("precond_apply: rank > 512");
  hipLaunchKernelGGL(pc_apply_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)((t + PA_CT - 1) / PA_CT)), dim3(256), 0, (hipStream_t)stream,
                     R, ldr, t, Q, ldq, k, n, W, sigma2, Out, ldo);
  return lz_check("precond_apply");
}

// ---- block Lanczos vector work (lanczos_kernels.hpp: pc_coef_kernel<1, float>, lzb_subtract_kernel, lzb_transform_kernel) ----
extern "C++" {
namespace {
// W[c][m] = <R[c], Q[m]> for ANY number b of rows R, in column groups of 16, and any number k of basis rows (k tiles of 128 over blockIdx.y)
template <typename TQ, typename TR>
int block_project(const TQ* Q, int64_t ldq, int k, const TR* R, int64_t ldr, int b, int n, double* W, double* workspace, int64_t workspace_doubles,
                  hipStream_t st) {
  long nb = ((long)n + 255) / 256;
  if (nb > 256) nb = 256;
  const int bg = b < 16 ? b : 16;
  if (workspace_doubles < (int64_t)nb * bg * k) return GPAMD_EWORKSPACE;
  const int slice = (int)(((long)n + nb - 1) / nb + PC_CHUNK - 1) / PC_CHUNK * PC_CHUNK;
  nb = ((long)n + slice - 1) / slice;
  for (int c0 = 0; c0 < b; c0 += 16) {
    const int tg = b - c0 < 16 ? b - c0 : 16;
    if (nb * ((k + 16 * PC_MT - 1) / (16 * PC_MT)) < PC_SMALL_GRID)   // few workgroups: 32-row basis tiles (lanczos_kernels.hpp, MT = 2; bitwise the same sums)
      hipLaunchKernelGGL((pc_coef_kernel<1, TQ, TR, 2>), dim3((unsigned)nb, (unsigned)((k + 31) / 32)), dim3(256), 0, st, R + (int64_t)c0 * ldr, ldr, tg, Q, ldq, k, n, slice,
                         workspace);
    else
      hipLaunchKernelGGL((pc_coef_kernel<1, TQ, TR>), dim3((unsigned)nb, (unsigned)((k + 16 * PC_MT - 1) / (16 * PC_MT))), dim3(256), 0, st, R + (int64_t)c0 * ldr, ldr, tg,
                         Q, ldq, k, n, slice, workspace);
    hipLaunchKernelGGL(pc_coef_sum_kernel, dim3((PC_SUM_LANES * tg * k + 255) / 256), dim3(256), 0, st, (const double*)workspace, (int)nb, tg * k, W + (int64_t)c0 * k);
  }
  return lz_check("block_project");
}
}  // namespace
}  // extern "C++"

int gpamd_block_project_f32(const float* Q, int64_t ldq, int k, const float* R, int64_t ldr, int b, int n, double* W, double* workspace,
                            int64_t workspace_doubles, void* stream) {
  if (!Q || !R || !W || !workspace || n <= 0 || b <= 0 || k <= 0 || ldq < n || ldr < n) return lz_fail("block_project: bad arguments");
  return block_project<float, float>(Q, ldq, k, R, ldr, b, n, W, workspace, workspace_doubles, (hipStream_t)stream);
}

int gpamd_block_project_f64(const double* Q, int64_t ldq, int k, const double* R, int64_t ldr, int b, int n, double* W, double* workspace,
                            int64_t workspace_doubles, void* stream) {
  if (!Q || !R || !W || !workspace || n <= 0 || b <= 0 || k <= 0 || ldq < n || ldr < n) return lz_fail("block_project: bad arguments");
  return block_project<double, double>(Q, ldq, k, R, ldr, b, n, W, workspace, workspace_doubles, (hipStream_t)stream);
}

int gpamd_block_subtract_f32(const float* Q, int64_t ldq, int k, const double* W, float* R, int64_t ldr, int b, int n, void* stream) {
  if (!Q || !R || !W || n <= 0 || b <= 0 || b > LZB_MAXB || k <= 0 || ldq < n || ldr < n) return lz_fail("block_subtract: bad arguments (b <= 32)");
  hipLaunchKernelGGL(lzb_subtract_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Q, ldq, k, W, R, ldr, b, n);
  return lz_check("block_subtract");
}

int gpamd_block_transform_f32(const double* M, float* R, int64_t ldr, int b, int n, void* stream) {
  if (!M || !R || n <= 0 || b <= 0 || b > LZB_MAXB || ldr < n) return lz_fail("block_transform: bad arguments (b <= 32)");
  hipLaunchKernelGGL(lzb_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, R, ldr, b, n);
  return lz_check("block_transform");
}

int gpamd_msminres_update_f32(const float* v, const float* d1, float* d2, float* x, const float* coef, int Q, int t, int n, int64_t ld,
                              void* stream) {
  if (!v || !d1 || !d2 || !x || !coef || Q <= 0 || t <= 0 || n <= 0 || ld < n || ld % 4 || Q > 65535 || t > 65535)
    return lz_fail("msminres_update: bad arguments (ld % 4 == 0, ld >= n)");
  hipLaunchKernelGGL(msminres_update_kernel, dim3(lz_blocks(n), t, Q), dim3(256), 0, (hipStream_t)stream, v, d1, d2, x, coef, Q, t, n, ld);
  return lz_check("msminres_update");
}

}  // extern "C"
