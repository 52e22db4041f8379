// extern "C" entry points for the fused bilinear-derivative (hyper-parameter gradient) kernel.
#include "../../include/gpamd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>

#include "kv_cull.hpp"
#include "kv_grad.hpp"

using namespace gpamd;
namespace gpamd {
extern thread_local char g_err[512];
}

namespace {
constexpr int GRAD_TGROUP = 128;  // probe columns per launch

int grad_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

void grad_plan(int n, int m, int* S, int* jchunk, int* nrb) {
  *nrb = (n + 127) / 128;
  const int slots = grad_num_cus() * 2;
  int smax = m / 256;
  if (smax < 1) smax = 1;
  if (smax > 64) smax = 64;
  int best = 1;
  double best_eff = -1;
  for (int s = 1; s <= smax; ++s) {
    int jc = ((m + s - 1) / s + 63) / 64 * 64;
    int se = (m + jc - 1) / jc;
    long units = (long)(*nrb) * se;
    long rounds = (units + slots - 1) / slots;
    double eff = (double)units / (double)(rounds * slots);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    if (eff >= 0.92) { best = s; break; }
  }
  int jc = ((m + best - 1) / best + 63) / 64 * 64;
  *jchunk = jc;
  *S = (m + jc - 1) / jc;
}

template <int KIND, int ISO>
int launch_grad(int dp, const GradArgs& a, unsigned grid, size_t lds, hipStream_t st) {
#define L(DPV)                                                                                                  \
  case DPV: {                                                                                                   \
    auto kfn = kv_grad_kernel<KIND, DPV, ISO>;                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, a);                                                 \
    return 0;                                                                                                   \
  }
  switch (dp) { L(4) L(8) L(12) L(16) L(20) L(24) L(32) }
#undef L
  return -2;
}
}  // namespace

extern "C" {

int64_t gpamd_kv_grad_workspace_doubles(int n, int m, int t, int dp) {
  int S, jc, nrb;
  if (n <= 0 || m <= 0 || t <= 0) return 0;
  grad_plan(n, m, &S, &jc, &nrb);
  int groups = (t + GRAD_TGROUP - 1) / GRAD_TGROUP;
  return (int64_t)groups * nrb * S * (1 + dp);
}

int gpamd_kv_grad_f32(int kind, const float* X1p, int n, const float* X2p, int m, int dp, const float* Lt, int64_t ldl,
                      const float* Rt, int64_t ldr, int t, int iso, float* out, double* workspace,
                      int64_t workspace_doubles, void* stream) {
  return gpamd_kv_grad_far_f32(kind, X1p, n, X2p, m, dp, Lt, ldl, Rt, ldr, t, iso, out, workspace, workspace_doubles, stream, nullptr, nullptr, nullptr,
                               nullptr, 0.f, nullptr, 0);
}

int64_t gpamd_kv_grad_far_workspace_ints(int n, int m) {
  if (n <= 0 || m <= 0) return 0;
  int S, jc, nrb;
  grad_plan(n, m, &S, &jc, &nrb);
  return (int64_t)nrb * S * (jc / 64 + 1);
}

int gpamd_kv_grad_far_f32(int kind, const float* X1p, int n, const float* X2p, int m, int dp, const float* Lt, int64_t ldl,
                          const float* Rt, int64_t ldr, int t, int iso, float* out, double* workspace,
                          int64_t workspace_doubles, void* stream, const float* row_centres, const float* row_radii, const float* tile_centres,
                          const float* tile_radii, float sq_cutoff, int* tile_workspace, int64_t tile_workspace_ints) {
  if (kind < 0 || kind > 3 || n <= 0 || m <= 0 || t <= 0 || ldl < n || ldr < m) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad: bad arguments");
    return GPAMD_EINVAL;
  }
  if (dp != 4 && dp != 8 && dp != 12 && dp != 16 && dp != 20 && dp != 24 && dp != 32) return GPAMD_EUNSUPPORTED;
  int S, jc, nrb;
  grad_plan(n, m, &S, &jc, &nrb);
  const int groups = (t + GRAD_TGROUP - 1) / GRAD_TGROUP;
  const int64_t units = (int64_t)nrb * S;
  if (workspace_doubles < groups * units * (1 + dp)) return GPAMD_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool cull = sq_cutoff > 0.f;   // far-pair culling (include/gpamd.h gpamd_kv_partials_far_f32): one tile list per unit, shared by the column groups
  if (cull) {
    if (!row_centres || !row_radii || !tile_centres || !tile_radii || !tile_workspace || tile_workspace_ints < units * (jc / 64 + 1)) {
      snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad: far-pair culling needs the four bounding-sphere arrays and gpamd_kv_grad_far_workspace_ints ints");
      return GPAMD_EINVAL;
    }
    CullArgs c;
    c.rc = row_centres; c.rr = row_radii; c.tc = tile_centres; c.tr = tile_radii;
    c.tiles = tile_workspace; c.tpc1 = jc / 64 + 1;
    c.n = n; c.m = m; c.dp = dp; c.bm = 128; c.bn = 64; c.nrb = nrb; c.jchunk = jc;
    c.sq_cut = sq_cutoff; c.done = nullptr;
    hipLaunchKernelGGL(cull_list_kernel<0>, dim3((unsigned)units), dim3(64), 0, st, c);
  }
  for (int g = 0; g < groups; ++g) {
    const int c0 = g * GRAD_TGROUP;
    const int tg = (t - c0) < GRAD_TGROUP ? (t - c0) : GRAD_TGROUP;
    GradArgs a;
    a.X1 = X1p; a.X2 = X2p;
    a.Lt = Lt + (int64_t)c0 * ldl;
    a.Rt = Rt + (int64_t)c0 * ldr;
    a.ldl = ldl; a.ldr = ldr;
    a.n = n; a.m = m; a.t = tg;
    a.S = S; a.jchunk = jc; a.nrb = nrb;
    a.part = workspace + (int64_t)g * units * (1 + dp);
    if (cull) { a.tiles = tile_workspace; a.tpc1 = jc / 64 + 1; }
    const int th = (tg + 1) / 2;
    const size_t lds = ((size_t)4 * 2 * th * 32 + (size_t)4 * 64 * dp) * sizeof(float);
    int rc = -2;
    switch (kind) {
      case GPAMD_RBF: rc = iso ? launch_grad<KIND_RBF, 1>(dp, a, (unsigned)units, lds, st) : launch_grad<KIND_RBF, 0>(dp, a, (unsigned)units, lds, st); break;
      case GPAMD_MATERN12: rc = iso ? launch_grad<KIND_MATERN12, 1>(dp, a, (unsigned)units, lds, st) : launch_grad<KIND_MATERN12, 0>(dp, a, (unsigned)units, lds, st); break;
      case GPAMD_MATERN32: rc = iso ? launch_grad<KIND_MATERN32, 1>(dp, a, (unsigned)units, lds, st) : launch_grad<KIND_MATERN32, 0>(dp, a, (unsigned)units, lds, st); break;
      case GPAMD_MATERN52: rc = iso ? launch_grad<KIND_MATERN52, 1>(dp, a, (unsigned)units, lds, st) : launch_grad<KIND_MATERN52, 0>(dp, a, (unsigned)units, lds, st); break;
    }
    if (rc) return GPAMD_EUNSUPPORTED;
  }
  hipLaunchKernelGGL(grad_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, (int)(groups * units), 1 + dp, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // extern "C"
