// extern "C" entry points of the Gram-form fused bilinear-derivative kernel with input gradients (kv_grad2.hpp).
// A kvm_* translation unit: compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (the W^T and distance tiles are consumed by the
// VALU straight from the MFMA destination registers).
#include "../../include/gpamd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "cg_kernels.hpp"
#include "kv_cull.hpp"
#include "kv_grad2.hpp"

using namespace gpamd;
namespace gpamd {
extern thread_local char g_err[512];
int grad2_launch_split(int kind, int dk, int mode, const Grad2Args& a, unsigned grid, hipStream_t st);   // kvm_grad3.hip
}

namespace {
// Columns per launch: up to G2_MAXT = 66 (33 MFMA k-steps -> 36 padded -> LDS rows of 76 floats: 61 KB, two workgroups per CU).
// Smaller groups raise the occupancy (34 columns: 36 KB, four workgroups per CU) but regenerate the pairs once per group, and that
// loses: measured at n = 500 000, t = 65 (profiles/r02_s5_grad_timing_maxcols{66,34,18}.json) 349 / 396 / 494 ms for 66 / 34 / 18.
// GPAMD_GRAD2_MAXCOLS overrides (tuning only).
int g2_maxcols() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("GPAMD_GRAD2_MAXCOLS");
    v = e ? atoi(e) : G2_MAXT;
    if (v < 4) v = 4;
    if (v > G2_MAXT) v = G2_MAXT;
    v &= ~1;
  }
  return v;
}

int g2_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// column groups: whole groups of (maxcols - 2), the last one may take up to maxcols (so 65 = 32 + 33, 66 = 32 + 34);
// split-operand contraction: up to WS_CP = 80 columns per launch (five k-steps of 16 column slots)
int g2_take(int rem, bool split = false) {
  if (split) return rem > WS_CP ? WS_CP : rem;
  return rem > g2_maxcols() ? g2_maxcols() - 2 : rem;
}
int g2_groups(int t, bool split = false) {
  int g = 0;
  for (int rem = t; rem > 0; ++g) rem -= g2_take(rem, split);
  return g;
}
int64_t pad_to(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

void g2_plan(int n, int m, int* S, int* jchunk, int* nrb) {
  *nrb = (n + 127) / 128;
  const int slots = g2_num_cus() * 2;
  int smax = m / 512;
  if (smax < 1) smax = 1;
  if (smax > 64) smax = 64;
  int best = 1;
  double best_eff = -1;
  for (int s = 1; s <= smax; ++s) {
    const int jc = ((m + s - 1) / s + G2_BN - 1) / G2_BN * G2_BN;
    const int se = (m + jc - 1) / jc;
    const long units = (long)(*nrb) * se;
    const long rounds = (units + slots - 1) / slots;
    const double eff = (double)units / (double)(rounds * slots);
    if (eff > best_eff + 0.01) { best_eff = eff; best = s; }
  }
  const int jc = ((m + best - 1) / best + G2_BN - 1) / G2_BN * G2_BN;
  *jchunk = jc;
  *S = (m + jc - 1) / jc;
}

int kdims(int d) { return d <= 6 ? d : (d <= 8 ? 8 : (d <= 10 ? 10 : (d <= 12 ? 12 : (d <= 16 ? 16 : (d <= 20 ? 20 : (d <= 24 ? 24 : 32)))))); }

template <int KIND, int D>
size_t g2_lds(int rs, int mode) {
  constexpr int KH = GramF16<D>::KH;
  constexpr int GZ = (1 + 2 * D + 3) / 4;
  return (size_t)(4 * 32 + G2_BN) * rs * 4 + (size_t)KH * G2_BN * 16 * 2 + (mode ? (size_t)4 * GZ * (G2_BN + 4) * 4 : 0);
}

template <int KIND, int D>
int launch_d(int mode, const Grad2Args& a, unsigned grid, hipStream_t st) {
  const size_t lds = g2_lds<KIND, D>(a.rs, mode);
  if (mode == 0) {
    auto kfn = kv_grad2_kernel<KIND, D, 0>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, a);
  } else if constexpr (D > 16) {
    return -2;   // per-dimension sums / input gradients beyond 16 dimensions: the caller's row-block path (backend.kv_grad_generic)
  } else {
    auto kfn = kv_grad2_kernel<KIND, D, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, a);
  }
  return 0;
}

template <int KIND>
int launch_kind(int dk, int mode, const Grad2Args& a, unsigned grid, hipStream_t st) {
  switch (dk) {
    case 1: return launch_d<KIND, 1>(mode, a, grid, st);
    case 2: return launch_d<KIND, 2>(mode, a, grid, st);
    case 3: return launch_d<KIND, 3>(mode, a, grid, st);
    case 4: return launch_d<KIND, 4>(mode, a, grid, st);
    case 5: return launch_d<KIND, 5>(mode, a, grid, st);
    case 6: return launch_d<KIND, 6>(mode, a, grid, st);
    case 8: return launch_d<KIND, 8>(mode, a, grid, st);
    case 10: return launch_d<KIND, 10>(mode, a, grid, st);
    case 12: return launch_d<KIND, 12>(mode, a, grid, st);
    case 16: return launch_d<KIND, 16>(mode, a, grid, st);
    case 20: return launch_d<KIND, 20>(mode, a, grid, st);
    case 24: return launch_d<KIND, 24>(mode, a, grid, st);
    case 32: return launch_d<KIND, 32>(mode, a, grid, st);
  }
  return -2;
}
}  // namespace

extern "C" {

int64_t gpamd_kv_grad2_workspace_doubles(int n, int m, int t, int d) {
  if (n <= 0 || m <= 0 || t <= 0 || d < 1 || d > 32) return 0;
  int S, jc, nrb;
  g2_plan(n, m, &S, &jc, &nrb);
  const int dp = (kdims(d) + 3) / 4 * 4;
  return (int64_t)g2_groups(t) * nrb * S * (2 + dp);
}

int64_t gpamd_kv_grad2_xworkspace_floats(int n, int m, int t, int d) {
  if (n <= 0 || m <= 0 || t <= 0 || d < 1 || d > 32) return 0;
  int S, jc, nrb;
  g2_plan(n, m, &S, &jc, &nrb);
  const int dp = (kdims(d) + 3) / 4 * 4;
  return (int64_t)g2_groups(t) * S * dp * ((n + 3) / 4 * 4);
}

int64_t gpamd_kv_grad2_split_workspace_floats(int n, int m) {
  if (n <= 0 || m <= 0) return 0;
  // hi + lo planes of L (rows padded to 128) and R (rows padded to 64), WS_CP f16 each; column maxima (2 x WS_CP) and scales (2 WS_CP + 1)
  return (pad_to(n, 128) + pad_to(m, G2_BN)) * WS_CP + 4 * WS_CP + 8;
}

int gpamd_kv_grad2_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Lt, int64_t ldl,
                       const float* Rt, int64_t ldr, int t, int iso, float* out, float* Gz1t, int64_t ldg, double* workspace,
                       int64_t workspace_doubles, float* xworkspace, int64_t xworkspace_floats, int flags, float* sworkspace,
                       int64_t sworkspace_floats, void* stream) {
  return gpamd_kv_grad2_far_f32(kind, kparam, X1p, n, X2p, m, d, X1c, Lt, ldl, Rt, ldr, t, iso, out, Gz1t, ldg, workspace, workspace_doubles, xworkspace,
                                xworkspace_floats, flags, sworkspace, sworkspace_floats, stream, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0);
}

int64_t gpamd_kv_grad2_far_workspace_ints(int n, int m) {
  if (n <= 0 || m <= 0) return 0;
  int S, jc, nrb;
  g2_plan(n, m, &S, &jc, &nrb);
  return (int64_t)nrb * S * (jc / G2_BN + 1);
}

int gpamd_kv_grad2_far_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Lt, int64_t ldl,
                           const float* Rt, int64_t ldr, int t, int iso, float* out, float* Gz1t, int64_t ldg, double* workspace,
                           int64_t workspace_doubles, float* xworkspace, int64_t xworkspace_floats, int flags, float* sworkspace,
                           int64_t sworkspace_floats, void* stream, const float* row_centres, const float* row_radii, const float* tile_centres,
                           const float* tile_radii, float sq_cutoff, int* tile_workspace, int64_t tile_workspace_ints) {
  if (n <= 0 || m <= 0 || t <= 0 || ldl < n || ldr < m || d < 1 || d > 32) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad2: bad arguments");
    return GPAMD_EINVAL;
  }
  if (kind != GPAMD_RBF && kind != GPAMD_MATERN32 && kind != GPAMD_MATERN52 && kind != GPAMD_RQ) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad2: Gram-form generation needs RBF / Matern 3/2 / Matern 5/2 (use gpamd_kv_grad_f32)");
    return GPAMD_EUNSUPPORTED;
  }
  const int dk = kdims(d), dp = (dk + 3) / 4 * 4;   // row stride of the prepared clouds = the kernel's DP (25 .. 28 dimensions share the D = 32 kernels: stride 32)
  int S, jc, nrb;
  g2_plan(n, m, &S, &jc, &nrb);
  const bool split = (flags & GPAMD_KV_SPLIT) != 0;
  const int groups = g2_groups(t, split);
  const int64_t units = (int64_t)nrb * S;
  const int64_t ldx = (n + 3) / 4 * 4;
  if (workspace_doubles < groups * units * (2 + dp)) return GPAMD_EWORKSPACE;
  if (Gz1t && (xworkspace_floats < (int64_t)groups * S * dp * ldx || ldg < n || ldg % 4)) return GPAMD_EWORKSPACE;
  if (split && (!sworkspace || sworkspace_floats < gpamd_kv_grad2_split_workspace_floats(n, m) ||
                (reinterpret_cast<uintptr_t>(sworkspace) & 15))) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad2: the split-operand contraction needs a 16-byte aligned sworkspace of gpamd_kv_grad2_split_workspace_floats");
    return GPAMD_EWORKSPACE;
  }
  const int mode = (iso && !Gz1t) ? 0 : 1;
  hipStream_t st = (hipStream_t)stream;
  // far-pair culling (include/gpamd.h gpamd_kv_partials_far_f32): one list of surviving 64-row j steps per (128-row block, j chunk) unit, shared by
  // every column group of this call
  const bool cull = sq_cutoff > 0.f;
  if (cull) {
    if (!row_centres || !row_radii || !tile_centres || !tile_radii || !tile_workspace || tile_workspace_ints < units * (jc / G2_BN + 1)) {
      snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad2: far-pair culling needs the four bounding-sphere arrays and gpamd_kv_grad2_far_workspace_ints ints");
      return GPAMD_EINVAL;
    }
    CullArgs c;
    c.rc = row_centres; c.rr = row_radii; c.tc = tile_centres; c.tr = tile_radii;
    c.tiles = tile_workspace; c.tpc1 = jc / G2_BN + 1;
    c.n = n; c.m = m; c.dp = dp; c.bm = 128; c.bn = G2_BN; c.nrb = nrb; c.jchunk = jc;
    c.sq_cut = sq_cutoff; c.done = nullptr;
    hipLaunchKernelGGL(cull_list_kernel<0>, dim3((unsigned)units), dim3(64), 0, st, c);
  }
  // split-operand workspace: [Lh | Ll | Rh | Rl | colmax L | colmax R | scales]
  const int64_t npad = pad_to(n, 128), mpad = pad_to(m, G2_BN);
  _Float16* Lh = reinterpret_cast<_Float16*>(sworkspace);
  _Float16* Ll = Lh + npad * WS_CP;
  _Float16* Rh = Ll + npad * WS_CP;
  _Float16* Rl = Rh + mpad * WS_CP;
  unsigned* cmax = reinterpret_cast<unsigned*>(Rl + mpad * WS_CP);
  float* scales = reinterpret_cast<float*>(cmax + 2 * WS_CP);
  int c0 = 0;
  for (int g = 0; g < groups; ++g) {
    const int rem = t - c0;
    const int tg = g2_take(rem, split);
    Grad2Args a;
    a.X1 = X1p; a.X2 = X2p; a.Xc = X1c;
    a.Lt = Lt + (int64_t)c0 * ldl;
    a.Rt = Rt + (int64_t)c0 * ldr;
    a.ldl = ldl; a.ldr = ldr;
    a.n = n; a.m = m; a.t = tg;
    a.S = S; a.jchunk = jc; a.nrb = nrb;
    a.th4 = ((tg + 1) / 2 + 3) / 4 * 4;
    a.rs = 2 * a.th4 + 4;   // = 4 * odd: eight lanes' 16-byte reads at this row stride cover all 32 banks exactly once
    a.part = workspace + (int64_t)g * units * (2 + dp);
    a.kparam = kparam;
    a.Px = Gz1t ? xworkspace + (int64_t)g * S * dp * ldx : nullptr;
    a.ldx = ldx;
    a.pxstride = (int64_t)dp * ldx;
    if (cull) { a.tiles = tile_workspace; a.tpc1 = jc / G2_BN + 1; }
    int rc = -2;
    if (split) {
      // pre-pass (kv_wsplit.hpp): column maxima of both blocks, scales with a constant product, the four planes
      (void)hipMemsetAsync(cmax, 0, sizeof(unsigned) * 2 * WS_CP, st);
      unsigned nbn = (unsigned)((n + 4095) / 4096), nbm = (unsigned)((m + 4095) / 4096);
      if (nbn > 64) nbn = 64;
      if (nbm > 64) nbm = 64;
      hipLaunchKernelGGL(wsplit_colmax_kernel<0>, dim3(nbn, tg), dim3(256), 0, st, a.Lt, ldl, n, cmax);
      hipLaunchKernelGGL(wsplit_colmax_kernel<0>, dim3(nbm, tg), dim3(256), 0, st, a.Rt, ldr, m, cmax + WS_CP);
      hipLaunchKernelGGL(wsplit_scales_kernel<0>, dim3(1), dim3(128), 0, st, (const unsigned*)cmax, (const unsigned*)(cmax + WS_CP), tg, scales);
      hipLaunchKernelGGL(wsplit_planes_kernel<0>, dim3((unsigned)((npad + 255) / 256), WS_CP / 8), dim3(256), 0, st, a.Lt, ldl, n, (int)npad, tg,
                         (const float*)scales, Lh, Ll, 1);
      hipLaunchKernelGGL(wsplit_planes_kernel<0>, dim3((unsigned)((mpad + 255) / 256), WS_CP / 8), dim3(256), 0, st, a.Rt, ldr, m, (int)mpad, tg,
                         (const float*)(scales + WS_CP), Rh, Rl, 2);
      a.Lh = Lh; a.Ll = Ll; a.Rh = Rh; a.Rl = Rl;
      a.wscale = scales + 2 * WS_CP;
      const int kid = kind == GPAMD_RBF ? KIND_RBF : (kind == GPAMD_MATERN32 ? KIND_MATERN32 : (kind == GPAMD_MATERN52 ? KIND_MATERN52 : KIND_RQ));
      rc = grad2_launch_split(kid, dk, mode, a, (unsigned)units, st);
    } else {
      switch (kind) {
        case GPAMD_RBF: rc = launch_kind<KIND_RBF>(dk, mode, a, (unsigned)units, st); break;
        case GPAMD_MATERN32: rc = launch_kind<KIND_MATERN32>(dk, mode, a, (unsigned)units, st); break;
        case GPAMD_MATERN52: rc = launch_kind<KIND_MATERN52>(dk, mode, a, (unsigned)units, st); break;
        case GPAMD_RQ: rc = launch_kind<KIND_RQ>(dk, mode, a, (unsigned)units, st); break;
      }
    }
    if (rc) return GPAMD_EUNSUPPORTED;
    c0 += tg;
  }
  // hyper-parameter sums: out[0] = sum W k, out[1 + q] = per-dimension sums (mode 0: out[1] = the single-lengthscale sum),
  // out[1 + dp] = sum W dk/dp at fixed s (shape parameter: RQ alpha; 0 for the other families)
  hipLaunchKernelGGL(grad2_finalize_kernel<0>, dim3(1), dim3(256), 0, st, workspace, (int)(groups * units), 2 + dp, out);
  if (Gz1t) {
    long nb = ((long)n + 1023) / 1024;
    if (nb > CG_MAXNB) nb = CG_MAXNB;
    hipLaunchKernelGGL((kv_reduce_kernel<float, false>), dim3((unsigned)nb, d), dim3(256), 0, st, (const float*)xworkspace, groups * S,
                       (int64_t)dp * ldx, ldx, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (int64_t)0, Gz1t, ldg, n, (float*)nullptr, (const int*)nullptr);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(gpamd::g_err, sizeof(gpamd::g_err), "kv_grad2: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // extern "C"
