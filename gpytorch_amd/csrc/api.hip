// extern "C" entry points of libgpamd.so (see include/gpamd.h for the contract and the reference
// interfaces each one replaces).  gfx950 only; no torch types, no CPU fallbacks.
#include "../../include/gpamd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cg_kernels.hpp"
#include "kv_dispatch.hpp"
#include "kv_cull.hpp"
#include "kv_valu.hpp"
#include "kv_gramv.hpp"
#include "kv_gram4.hpp"
#include "kv_gram16.hpp"
#include "kv_vsplit.hpp"
#include "kv_directh.hpp"
#include "misc_kernels.hpp"

using namespace gpamd;

namespace gpamd {
thread_local char g_err[512] = "";  // shared by every translation unit of the library (gpamd_last_error)
}

namespace {

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

float prep_coef(int kind, float kparam) {
  switch (kind) {
    case GPAMD_RBF: return sqrtf(0.5f * 1.4426950408889634f);  // exp(-0.5 s) = exp2(-(0.5 log2 e) s)
    case GPAMD_MATERN12: return 1.0f;
    case GPAMD_MATERN32: return sqrtf(3.0f);
    case GPAMD_MATERN52: return sqrtf(5.0f);
    case GPAMD_RQ: return 1.0f / sqrtf(2.0f * kparam);   // (1 + |x - x'|^2 / (2 alpha l^2))^-alpha = (1 + |z - z'|^2)^-alpha
  }
  return 0.f;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// column-tile variant for t columns: valu (t <= 8) or mfma CT/EX
struct KvVariant {
  bool valu;
  int tpad;    // valu: 1,2,4,8
  int ct, ex;  // mfma
  int g4;      // > 0: kv_gram4 with this many column groups of four (2, 3, 6); 16 / 17: kv_gram16 without / with EX
  bool split;  // kv_gramh: contraction of hi/lo-split operands on the f16 matrix pipe (ct, ex as for mfma)
  int ni;      // kv_gramh: 32-row tiles per wave
  bool gram;   // the selected kernel forms the squared distances by the quadratic expansion (false: direct differences)
  bool direct; // with split: kv_directh (direct differences + split contraction; ct, ex as for mfma, ct <= 2)
  int bm;      // rows per workgroup
  int bn;      // j tile
};

bool gram_ok(int kind, int flags);

// gram: the Gram-form kernels apply.  Column-count ladder (measured at n = 500 000, profiles/r02_s*_kv_small_t*.json):
//   1..4   kv_gramv (VALU contraction)            5..8   kv_gram4, two column groups of four (4x4x1 MFMA)
//   9..16  kv_gram16 (16-column 16x16x1_4B tile)  17..24 kv_gram4, six groups     25..  kv_gram (32-column tiles)
// flags (tuning / A-B only): GPAMD_KV_WIDE restores the older selection (VALU contraction up to 16 columns, the 32-column
// tile above), GPAMD_KV_G4 sends 9..12 columns to kv_gram4 with three groups
// small: few output rows (n < KGH_SMALL_N) -- the split kernels then take ONE 32-row tile per wave (128 rows per workgroup) so that
// the launch still spreads over the chip
// GPAMD_KV_BLOCK128 (the caller bounds the radius of 128-row blocks only): the split kernels take one row tile per wave (128 rows per workgroup, the
// `small` geometry), every other column count leaves the Gram form (its kernels centre 256 / 512-row blocks) for the direct-difference kernels
KvVariant pick_variant(int t, bool gram, int flags = 0, bool light = false, bool small = false, int dk = 16) {   // dk: kernel dims (light: dk <= 3 and RBF)  // t <= 129 handled per launch group; light: RBF, d <= 3
  KvVariant v{};
  const bool wide = flags & GPAMD_KV_WIDE;
  // GPAMD_KV_SPLIT_FEW: groups of fewer than five columns stay on the split-operand kernels too (a 32-column tile mostly empty: slower than the
  // few-column kernels when every tile is visited, faster when the caller culls far tiles -- only the split kernels walk tile lists)
  const int min_cols = (flags & GPAMD_KV_SPLIT_FEW) ? 1 : KGH_MIN_COLS;
  if (flags & GPAMD_KV_BLOCK128) {
    if (gram && (flags & GPAMD_KV_SPLIT) && t >= min_cols && t <= KGH_GROUP + 1) small = true;
    else gram = false;
  }
  v.gram = gram;
  if (gram && (flags & GPAMD_KV_SPLIT) && t >= min_cols && t <= KGH_GROUP + 1) {
    v.split = true;
    v.ex = (t % 32 == 1 && t > 1) ? 1 : 0;
    v.ct = (t - v.ex + 31) / 32;
    // four row tiles per wave with two column tiles ("lean", kv_gramh.hpp) where they fit: one Gram MFMA per block (dk <= 3) and, with the extra
    // column, the light generation of the RBF only (the other families would spill 7..15 registers there)
    v.ni = small ? 1 : kgh_ni(v.ct, (v.ex && !light && dk <= 16) ? 16 : dk);
    v.bm = kgh_bm(v.ni);
    v.bn = KGH_BN;
  } else if (!gram && (flags & GPAMD_KV_SPLIT) && t >= min_cols && t <= KDH_COLS + 1 && dk <= KDH_MAX_DIM) {
    // direct differences (clouds / rows outside the policy of the quadratic expansion, Matern nu = 1/2) with the contraction on the f16 matrix pipe
    // (kv_directh.hpp): the VALU keeps the generation only -- 9.8 instead of ~17 VALU instructions per pair at d = 3, eleven columns
    v.split = true;
    v.direct = true;
    v.ex = (t % 32 == 1 && t > 1) ? 1 : 0;
    v.ct = (t - v.ex + 31) / 32;
    v.ni = kdh_ni(small, v.ct, dk);
    v.bm = kdh_bm(v.ni);
    v.bn = KGH_BN;
  } else if (gram && !wide && t >= 5 && t <= 24 && dk <= 16) {   // (beyond 16 dimensions: no 4-column / 16-column tile kernels -- kv_gramv up to 16 columns, the 32-column tile above)
    if (t <= 8) v.g4 = 2;
    else if (t <= 12 && (flags & GPAMD_KV_G4)) v.g4 = 3;
    else if (t <= 16) v.g4 = 16;   // kv_gram16
    else if (t == 17) v.g4 = 17;   // kv_gram16 + the extra VALU column (16 probes + y)
    else v.g4 = 6;
    // RBF in <= 3 dimensions, 9..12 columns: generation is one v_exp_f32 per pair and three column groups on 4x4x1 beat the
    // 16-column tile by 7 % (64.0 vs 68.7 ms at n = 500 000, t = 11); every heavier generation (Matern: + v_sqrt_f32, d > 3:
    // more Gram MFMAs) does not issue under the 8-cycle MFMAs and is faster on the 16-column tile
    if (v.g4 == 16 && t <= 12 && light) v.g4 = 3;
    v.bm = v.g4 >= 16 ? KG16_BM : kg4_bm(v.g4);
    v.bn = v.g4 >= 16 ? KG16_BN : KG4_BN;
  } else if (t <= 8 || (gram && t <= 16) || (t <= 16 && dk <= 4)) {   // (direct differences, 9 .. 16 columns: kv_valu<T = 16> up to 4 dimensions -- 33 -> 27 ms per
    // product on the road3d shape (d = 3); at d = 10 the 32-column matrix-pipe tile wins, 0.83 against 1.16 ms on the protein shape's wide rows)
    v.valu = true;
    v.tpad = t <= 1 ? 1 : (t <= 2 ? 2 : (t <= 4 ? 4 : (t <= 8 ? 8 : 16)));
    v.bm = KVV_BM;
    v.bn = KVV_BN;
  } else {
    v.valu = false;
    if (t % 32 == 1 && t > 1) {
      v.ct = (t - 1) / 32;
      v.ex = 1;
    } else {
      v.ct = (t + 31) / 32;
      v.ex = 0;
    }
    v.bm = kv_bm_for_ct(v.ct, dk);
    v.bn = KV_BN;
  }
  return v;
}

constexpr int KV_GROUP = 128;  // columns per launch group (CT = 4); a trailing 129th column rides as EX

bool split_on(int kind, int flags) { return gram_ok(kind, flags) && (flags & GPAMD_KV_SPLIT); }
// direct differences + split contraction (kv_directh.hpp): SPLIT without an applicable GRAM, up to KDH_MAX_DIM dimensions
bool dsplit_on(int kind, int flags, int d) { return !gram_ok(kind, flags) && (flags & GPAMD_KV_SPLIT) && kv_kernel_dims(d) <= KDH_MAX_DIM; }

// split [0, t) into launch groups of <= cap (+1) columns: cap = 128, or 64 for the split-operand kernels
int group_cols(int t, int g0, int cap = KV_GROUP) {
  int rem = t - g0;
  if (rem <= cap + 1) return rem;  // includes the cap + EX case
  return cap;
}
int group_cap(int kind, int flags, int d) { return split_on(kind, flags) ? KGH_GROUP : (dsplit_on(kind, flags, d) ? KDH_COLS : KV_GROUP); }

// Split-operand launches keep, behind the S partial slabs of the workspace: column maxima | column multipliers | the two f16
// planes of every launch group (32 ct rows of ldh positions each).  Offsets in floats, all multiples of 4.
struct SplitLayout {
  int64_t base, colmax, colmul, planes, total;   // total: floats behind `base`
  int64_t ldh;
  int rows;                                      // plane rows over all split groups
};
SplitLayout split_layout(int kind, int flags, int m, int d, int t, int S, int64_t ldo) {
  SplitLayout L{};
  L.base = ((int64_t)S * t * ldo + 3) / 4 * 4;
  L.ldh = ((int64_t)m + KGH_BN - 1) / KGH_BN * KGH_BN;
  if (!split_on(kind, flags) && !dsplit_on(kind, flags, d)) return L;
  const int cap = group_cap(kind, flags, d);
  int rows = 0;
  for (int g0 = 0; g0 < t;) {
    const int tg = group_cols(t, g0, cap);
    KvVariant v = pick_variant(tg, gram_ok(kind, flags), flags, false, false, kv_kernel_dims(d));   // (row tiling does not change the plane rows)
    if (v.split) rows += 32 * v.ct;
    g0 += tg;
  }
  L.rows = rows;
  if (!rows) return L;
  const int64_t ncol = (2 * (int64_t)t + 64 + 3) / 4 * 4;   // 32 ct + 1 multiplier slots per group (<= t + 32 groups + 1)
  L.colmax = 0;
  L.colmul = ncol;
  L.planes = 2 * ncol;
  L.total = 2 * ncol + (int64_t)rows * L.ldh;    // two f16 planes = rows * ldh floats
  return L;
}

int kernel_dims(int d) { return kv_kernel_dims(d); }  // kernels exist for these valid-dimension counts; other d use the next one

const void* family_ptr(int kind, int mode, int d, int v, int ex, int ni = 0) {
  if (mode == KV_MODE_GRAMH) {
    switch (kind) {
      case GPAMD_RBF: return kvh_kernel_ptr_rbf(d, v, ex, ni);
      case GPAMD_MATERN32: return kvh_kernel_ptr_matern32(d, v, ex, ni);
      case GPAMD_MATERN52: return kvh_kernel_ptr_matern52(d, v, ex, ni);
      case GPAMD_RQ: return kvh_kernel_ptr_rq(d, v, ex, ni);
    }
    return nullptr;
  }
  if (mode == KV_MODE_DIRECTH) {
    switch (kind) {
      case GPAMD_RBF: return kvd_kernel_ptr_rbf(d, ni, v, ex);
      case GPAMD_MATERN12: return kvd_kernel_ptr_matern12(d, ni, v, ex);
      case GPAMD_MATERN32: return kvd_kernel_ptr_matern32(d, ni, v, ex);
      case GPAMD_MATERN52: return kvd_kernel_ptr_matern52(d, ni, v, ex);
      case GPAMD_RQ: return kvd_kernel_ptr_rq(d, ni, v, ex);
    }
    return nullptr;
  }
  if (mode == KV_MODE_GRAM4) {
    switch (kind) {
      case GPAMD_RBF: return kvm_kernel_ptr_rbf(d, v);
      case GPAMD_MATERN32: return kvm_kernel_ptr_matern32(d, v);
      case GPAMD_MATERN52: return kvm_kernel_ptr_matern52(d, v);
      case GPAMD_RQ: return kvm_kernel_ptr_rq(d, v);
    }
    return nullptr;
  }
  if (mode == KV_MODE_GRAMV) {
    switch (kind) {
      case GPAMD_RBF: return kvs_kernel_ptr_rbf(d, v);
      case GPAMD_MATERN32: return kvs_kernel_ptr_matern32(d, v);
      case GPAMD_MATERN52: return kvs_kernel_ptr_matern52(d, v);
      case GPAMD_RQ: return kvs_kernel_ptr_rq(d, v);
    }
    return nullptr;
  }
  switch (kind) {
    case GPAMD_RBF: return kv_kernel_ptr_rbf(mode, d, v, ex);
    case GPAMD_MATERN12: return kv_kernel_ptr_matern12(mode, d, v, ex);
    case GPAMD_MATERN32: return kv_kernel_ptr_matern32(mode, d, v, ex);
    case GPAMD_MATERN52: return kv_kernel_ptr_matern52(mode, d, v, ex);
    case GPAMD_RQ: return kv_kernel_ptr_rq(mode, d, v, ex);
  }
  return nullptr;
}

bool gram_ok(int kind, int flags) { return (flags & GPAMD_KV_GRAM) && kind != GPAMD_MATERN12; }

int kv_mode(int kind, int flags, int d, const KvVariant& v) {
  const bool gram = v.gram;   // (= gram_ok(kind, flags) unless GPAMD_KV_BLOCK128 sent this column group to the direct-difference kernels)
  if (v.split) return v.direct ? KV_MODE_DIRECTH : KV_MODE_GRAMH;
  if (v.g4) return KV_MODE_GRAM4;
  if (v.valu) return gram ? KV_MODE_GRAMV : KV_MODE_VALU;
  return gram ? KV_MODE_GRAM : KV_MODE_MFMA;
}

// tile geometry of the selected kernel (the small-t Gram variant uses its own row block / j tile)
void variant_geometry(int mode, KvVariant* v) {
  if (mode == KV_MODE_GRAMV) {
    v->bm = KGV_BM;
    v->bn = KGV_BN;
  }
}

int variant_key(const KvVariant& v) { return v.g4 ? v.g4 : (v.valu ? v.tpad : v.ct); }

// resident workgroups per CU of the selected kernel (runtime occupancy query; static table without a device)
int wg_per_cu(int kind, int mode, int dk, const KvVariant& v) {
  const void* fn = family_ptr(kind, mode, dk, variant_key(v), v.ex, v.ni);
  int nb = 0;
  if (fn && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0) == hipSuccess && nb > 0) return nb;
  (void)hipGetLastError();
  if (v.split) return 3;
  if (v.g4) return 2;
  if (v.valu) return 4;
  if (mode == KV_MODE_GRAM) return 3;
  return v.ct <= 2 ? 3 : 2;
}

void plan_split(int kind, int n, int m, int d, int t, int flags, int* S, int* jchunk) {
  // Grid = (row blocks) x (S chunks of the contracted index).  All units cost the same, so the launch
  // runs in ceil(units / slots) rounds of resident workgroups: pick the S whose last round is nearly
  // full (efficiency = units / (rounds * slots)), keeping every chunk >= 16 LDS tiles (per-unit prologue
  // and partial-slab write < 1 %) and preferring the smallest S among near-ties (less slab traffic).
  const int cap = group_cap(kind, flags, d);
  KvVariant v = pick_variant(group_cols(t, 0, cap), gram_ok(kind, flags), flags, kind == GPAMD_RBF && d <= 3, n < KGH_SMALL_N, kernel_dims(d));
  const int mode = kv_mode(kind, flags, d, v);
  variant_geometry(mode, &v);
  const int nrb = (n + v.bm - 1) / v.bm;
  const long slots = (long)num_cus() * wg_per_cu(kind, mode, kernel_dims(d), v);
  const int min_chunk = 16 * v.bn;
  int smax = m / min_chunk;
  if (smax < 1) smax = (m >= 4 * v.bn) ? m / (4 * v.bn) : 1;  // small problems: favour parallelism
  // few output rows against many contracted ones (the wide-row launch of a block-centred product, a test set against the training set): 16-tile chunks
  // would leave most of the chip without a workgroup -- 72 workgroups for 9 000 x 36 584 pairs on the protein-shaped workload, 1.16 ms where the pairs
  // are worth 0.16 (profiles/r05_s4_workload_protein_kernel_stats.csv) -- so chunks go down to 4 tiles until one round of workgroups exists
  if ((long)nrb * smax < slots && m >= 8 * v.bn) {
    const long want = (slots + nrb - 1) / nrb;
    const int cap4 = m / (4 * v.bn);
    smax = (int)(want < cap4 ? want : cap4);
    if (smax < 1) smax = 1;
  }
  if (smax > 48) smax = 48;
  int best_s = 1;
  double best = -1.0;
  for (int s = 1; s <= smax; ++s) {
    const int jc = ((m + s - 1) / s + v.bn - 1) / v.bn * v.bn;
    const int se = (m + jc - 1) / jc;
    if (se != s) continue;  // rounding collapsed this split onto a smaller one
    const long units = (long)nrb * se;
    const long rounds = (units + slots - 1) / slots;
    // the last chunk is shorter than the others: count its units at their true relative cost
    const double last = (double)(m - (long)(se - 1) * jc) / (double)jc;
    const double work = (double)nrb * ((se - 1) + last);
    const double eff = work / ((double)rounds * (double)slots);
    if (eff > best + 0.01) {
      best = eff;
      best_s = s;
    }
  }
  const int jc = ((m + best_s - 1) / best_s + v.bn - 1) / v.bn * v.bn;
  *jchunk = jc;
  *S = (m + jc - 1) / jc;
}

unsigned col_blocks(int n) {
  long nb = ((long)n + 1023) / 1024;
  if (nb < 1) nb = 1;
  if (nb > CG_MAXNB) nb = CG_MAXNB;
  return (unsigned)nb;
}

}  // namespace

extern "C" {

int gpamd_abi_version(void) { return GPAMD_ABI_VERSION; }

const char* gpamd_last_error(void) { return g_err; }

int gpamd_prep_points_f32(int kind, float kparam, const float* X, int n, int d, int64_t ldx, const float* ls, int nls,
                          const float* shift, float* Xp, int dp, void* stream) {
  if (kind < 0 || kind > GPAMD_RQ) return fail(GPAMD_EINVAL, "prep_points: unknown kind");
  if (n <= 0 || d <= 0 || dp < d || dp % 4 || (nls != 1 && nls != d)) return fail(GPAMD_EINVAL, "prep_points: bad shape");
  if (!aligned16(Xp)) return fail(GPAMD_EINVAL, "prep_points: Xp must be 16-byte aligned");
  long total = (long)n * dp;
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(prep_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, n, d, ldx, ls, nls, shift,
                     prep_coef(kind, kparam), Xp, dp);
  return check_launch("prep_points");
}

int gpamd_kv_plan(int kind, int n, int m, int d, int t, int flags, int64_t ldo, int* S_host, int* jchunk_host,
                  int64_t* workspace_floats_host) {
  if (kind < 0 || kind > GPAMD_RQ || n <= 0 || m <= 0 || t <= 0 || d < 1 || d > KV_MAX_DIM) return fail(GPAMD_EINVAL, "kv_plan: bad shape");
  int S, jc;
  plan_split(kind, n, m, d, t, flags, &S, &jc);
  if (S_host) *S_host = S;
  if (jchunk_host) *jchunk_host = jc;
  if (workspace_floats_host) {
    const SplitLayout L = split_layout(kind, flags, m, d, t, S, ldo);
    *workspace_floats_host = L.total ? L.base + L.total : (int64_t)S * t * ldo;
  }
  return 0;
}

int gpamd_kv_partials_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt,
                          int64_t ldv, int t, float* P, int64_t ldo, int S, int jchunk, int flags, const int* done,
                          void* stream) {
  return gpamd_kv_partials_far_f32(kind, kparam, X1p, n, X2p, m, d, X1c, Vt, ldv, t, P, ldo, S, jchunk, flags, done, stream, nullptr, nullptr, nullptr,
                                   nullptr, 0.f, nullptr, 0);
}

int64_t gpamd_kv_far_workspace_ints(int n, int S, int jchunk) {
  if (n <= 0 || S <= 0 || jchunk <= 0) return 0;
  return (int64_t)((n + 127) / 128) * S * (jchunk / 128 + 1);   // (the smallest row block of any kernel is 128 rows)
}

int gpamd_kv_partials_far_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt,
                              int64_t ldv, int t, float* P, int64_t ldo, int S, int jchunk, int flags, const int* done, void* stream,
                              const float* row_centres, const float* row_radii, const float* tile_centres, const float* tile_radii, float sq_cutoff,
                              int* tile_ws, int64_t tile_ws_ints) {
  if (kind < 0 || kind > GPAMD_RQ) return fail(GPAMD_EINVAL, "kv: unknown kind");
  const bool cull = sq_cutoff > 0.f && row_centres && row_radii && tile_centres && tile_radii && tile_ws;
  if (sq_cutoff > 0.f && !cull) return fail(GPAMD_EINVAL, "kv: far-pair culling needs all four bounding-sphere arrays and the tile-list workspace");
  if (cull && (jchunk % 128 || S <= 0 || tile_ws_ints < gpamd_kv_far_workspace_ints(n, S, jchunk)))
    return fail(GPAMD_EINVAL, "kv: far-pair culling needs jchunk % 128 == 0 (gpamd_kv_plan) and gpamd_kv_far_workspace_ints(n, S, jchunk) ints of workspace");
  if (n <= 0 || m <= 0 || t <= 0 || S <= 0) return fail(GPAMD_EINVAL, "kv: bad shape");
  if (d < 1 || d > KV_MAX_DIM) return fail(GPAMD_EUNSUPPORTED, "kv: input dimension must be in 1..32");
  // kernels are instantiated for D in {1,2,3,4,5,6,8,10,12,16,20,24,32} valid dimensions; other d use the next one
  // (same padded stride, the extra coordinates are the zeros written by prep_points)
  const int dk = kernel_dims(d);
  if (ldv % 4 || ldv < m || ldo < n) return fail(GPAMD_EINVAL, "kv: leading dimensions must be >= extent and ldv % 4 == 0");
  if (!aligned16(Vt) || !aligned16(X1p) || !aligned16(X2p) || !aligned16(X1c)) return fail(GPAMD_EINVAL, "kv: buffers must be 16-byte aligned");
  if (jchunk % 4 || (int64_t)jchunk * S < m) return fail(GPAMD_EINVAL, "kv: jchunk*S must cover m and jchunk % 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int cap = group_cap(kind, flags, d);
  const SplitLayout L = split_layout(kind, flags, m, d, t, S, ldo);
  float* xbase = P + L.base;
  if (L.total) {
    if (!aligned16(P) || ldo % 4) return fail(GPAMD_EINVAL, "kv: the split-operand path needs a 16-byte aligned workspace and ldo % 4 == 0");
    if (jchunk % KGH_BN) return fail(GPAMD_EINVAL, "kv: the split-operand path needs jchunk % 128 == 0 (use gpamd_kv_plan)");
    (void)hipMemsetAsync(xbase + L.colmax, 0, sizeof(float) * (size_t)L.colmul, st);
  }
  int64_t prow = 0;   // plane rows used so far
  int mulslot = 0;    // multiplier slots used so far
  for (int g0 = 0; g0 < t;) {
    const int tg = group_cols(t, g0, cap);
    KvVariant v = pick_variant(tg, gram_ok(kind, flags), flags, kind == GPAMD_RBF && d <= 3, n < KGH_SMALL_N, dk);
    const int mode = kv_mode(kind, flags, d, v);
    variant_geometry(mode, &v);
    KvhArgs ka;
    KvArgs& a = ka.a;
    a.X1 = X1p; a.X2 = X2p;
    a.Vt = Vt + (int64_t)g0 * ldv;
    a.P = P + (int64_t)g0 * ldo;
    a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)t * ldo;
    a.n = n; a.m = m; a.t = tg;
    a.S = S; a.jchunk = jchunk;
    a.nrb = (n + v.bm - 1) / v.bm;
    a.done = done;
    a.kparam = kparam;
    a.Xc = X1c;
    if (cull && v.split) {
      // far-pair culling (split-operand kernels only, kv_mfma.hpp): this group's tile lists, built for ITS row block on the same stream
      CullArgs c;
      c.rc = row_centres; c.rr = row_radii; c.tc = tile_centres; c.tr = tile_radii;
      c.tiles = tile_ws; c.tpc1 = jchunk / 128 + 1;
      c.n = n; c.m = m; c.dp = (dk + 3) / 4 * 4; c.bm = v.bm; c.bn = v.bn; c.nrb = a.nrb; c.jchunk = jchunk;
      c.sq_cut = sq_cutoff; c.done = done;
      hipLaunchKernelGGL(cull_list_kernel<0>, dim3((unsigned)a.nrb * (unsigned)S), dim3(64), 0, st, c);
      a.tiles = tile_ws; a.tpc1 = c.tpc1;
    }
    unsigned grid = (unsigned)a.nrb * (unsigned)S;
    const void* fn = family_ptr(kind, mode, dk, variant_key(v), v.ex, v.ni);
    if (!fn) return fail(GPAMD_EUNSUPPORTED, "kv: no kernel variant for this shape");
    if (v.split) {
      // pre-pass: per-column scale + the two f16 planes of this group's matrix-pipe columns (the extra column stays f32)
      const int tc = 32 * v.ct, tm = tg - v.ex;
      unsigned* colmax = reinterpret_cast<unsigned*>(xbase + L.colmax) + mulslot;
      float* colmul = xbase + L.colmul + mulslot;
      _Float16* vh = reinterpret_cast<_Float16*>(xbase + L.planes) + 2 * prow * L.ldh;
      _Float16* vl = vh + (int64_t)tc * L.ldh;
      unsigned nbm = (unsigned)((m + 4095) / 4096);
      if (nbm > 64) nbm = 64;
      hipLaunchKernelGGL(vsplit_colmax_kernel, dim3(nbm, tm), dim3(256), 0, st, a.Vt, ldv, m, colmax, done);
      hipLaunchKernelGGL(vsplit_kernel, dim3((unsigned)((L.ldh / 8 + 255) / 256), tc), dim3(256), 0, st, a.Vt, ldv, m, tm, vh, vl,
                         L.ldh, (const unsigned*)colmax, colmul, done);
      ka.Vh = vh; ka.Vl = vl; ka.ldh = L.ldh; ka.colmul = colmul;
      prow += tc;
      mulslot += tc + 1;
      void* kargs[] = {(void*)&ka};
      (void)hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, 0, st);
    } else {
      void* kargs[] = {(void*)&a};
      (void)hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, 0, st);
    }
    int rc = check_launch("kv_partials");
    if (rc) return rc;
    g0 += tg;
  }
  return 0;
}

int gpamd_kv_reduce_f32(const float* P, int S, int64_t ldp, int t, int n, const float* scale, const float* dscale,
                        const float* dvec, const float* Vd, int64_t ldd, float* Out, int64_t ldo, const int* done,
                        void* stream) {
  if (S <= 0 || t <= 0 || n <= 0) return fail(GPAMD_EINVAL, "kv_reduce: bad shape");
  if (ldp % 4 || ldo % 4 || (Vd && ldd % 4)) return fail(GPAMD_EINVAL, "kv_reduce: leading dimensions must be multiples of 4");
  dim3 grid(col_blocks(n), t);
  hipLaunchKernelGGL((kv_reduce_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, P, S, (int64_t)t * ldp, ldp,
                     scale, dscale, dvec, Vd, ldd, Out, ldo, n, (float*)nullptr, done);
  return check_launch("kv_reduce");
}

int gpamd_kv_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt, int64_t ldv,
                 int t, const float* scale, const float* dscale, const float* Vd, int64_t ldd, float* Out,
                 int64_t ldo, float* workspace, int64_t workspace_floats, int flags, void* stream) {
  int S, jc;
  if (kind < 0 || kind > GPAMD_RQ || n <= 0 || m <= 0 || t <= 0 || d < 1 || d > KV_MAX_DIM) return fail(GPAMD_EINVAL, "kv: bad shape");
  plan_split(kind, n, m, d, t, flags, &S, &jc);
  const int64_t ldp = (n + 3) / 4 * 4;
  const SplitLayout L = split_layout(kind, flags, m, d, t, S, ldp);
  if (workspace_floats < (L.total ? L.base + L.total : (int64_t)S * t * ldp))
    return fail(GPAMD_EWORKSPACE, "kv: workspace too small (use gpamd_kv_plan with ldo = round_up(n,4) and the same flags)");
  int rc = gpamd_kv_partials_f32(kind, kparam, X1p, n, X2p, m, d, X1c, Vt, ldv, t, workspace, ldp, S, jc, flags, nullptr, stream);
  if (rc) return rc;
  return gpamd_kv_reduce_f32(workspace, S, ldp, t, n, scale, dscale, nullptr, Vd, ldd, Out, ldo, nullptr, stream);
}

#define KIND_SWITCH(kind, CALL)                     \
  switch (kind) {                                   \
    case GPAMD_RBF: { constexpr int KK = KIND_RBF; CALL; } break;       \
    case GPAMD_MATERN12: { constexpr int KK = KIND_MATERN12; CALL; } break; \
    case GPAMD_MATERN32: { constexpr int KK = KIND_MATERN32; CALL; } break; \
    case GPAMD_MATERN52: { constexpr int KK = KIND_MATERN52; CALL; } break; \
    case GPAMD_RQ: { constexpr int KK = KIND_RQ; CALL; } break; \
    default: return fail(GPAMD_EINVAL, "unknown kind"); \
  }

int gpamd_kernel_rows_f32(int kind, float kparam, const float* X1p, const int64_t* rows, int nrows, const float* X2p, int m, int dp,
                          const float* scale, float* out, int64_t ldo, void* stream) {
  if (nrows <= 0 || m <= 0) return fail(GPAMD_EINVAL, "kernel_rows: bad shape");
  dim3 grid((m + 255) / 256, nrows);
  KIND_SWITCH(kind, hipLaunchKernelGGL((kernel_rows_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, rows, nrows,
                                       X2p, m, dp, scale, out, ldo, kparam));
  return check_launch("kernel_rows");
}

int gpamd_kernel_dense_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int dp, const float* scale,
                           float* out, int64_t ldo, void* stream) {
  if (n <= 0 || m <= 0) return fail(GPAMD_EINVAL, "kernel_dense: bad shape");
  if (n > 65535) return fail(GPAMD_EUNSUPPORTED, "kernel_dense: n > 65535 (materialising K is what this library avoids)");
  dim3 grid((m + 255) / 256, n);
  KIND_SWITCH(kind, hipLaunchKernelGGL((kernel_dense_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, n, X2p, m,
                                       dp, scale, out, ldo, kparam));
  return check_launch("kernel_dense");
}

int gpamd_kernel_diag_f32(int kind, float kparam, const float* X1p, const float* X2p, int n, int dp, const float* scale, float* out,
                          void* stream) {
  if (n <= 0) return fail(GPAMD_EINVAL, "kernel_diag: bad shape");
  dim3 grid((n + 255) / 256);
  KIND_SWITCH(kind, hipLaunchKernelGGL((kernel_diag_kernel<KK>), grid, dim3(256), 0, (hipStream_t)stream, X1p, X2p, n, dp,
                                       scale, out, kparam));
  return check_launch("kernel_diag");
}

int gpamd_coldot_f32(const float* A, const float* B, int64_t ld, int n, int t, float* out, float* scratch,
                     void* stream) {
  if (n <= 0 || t <= 0 || ld % 4) return fail(GPAMD_EINVAL, "coldot: bad shape");
  unsigned nb = col_blocks(n);
  hipLaunchKernelGGL((coldot_kernel<float>), dim3(nb, t), dim3(256), 0, (hipStream_t)stream, A, B, ld, n, scratch,
                     (const int*)nullptr);
  hipLaunchKernelGGL((colsum_partials_kernel<float>), dim3(t), dim3(256), 0, (hipStream_t)stream, scratch, (int)nb, out);
  return check_launch("coldot");
}

// ------------------------------------------------------------------------------------------- mBCG
struct gpamd_cg {
  CgState<float> st;
};

int64_t gpamd_cg_fscratch_elems(int t, int hist_len) {
  // bnorm t | rnorm t | rho 2t | stats 4 | alpha_hist h*t | beta_hist h*t | 3 partial arrays t*256
  return (int64_t)4 * t + 4 + (int64_t)2 * hist_len * t + (int64_t)3 * t * CG_MAXNB;
}
int64_t gpamd_cg_iscratch_elems(int t) { return (int64_t)2 * t + 2; }

int gpamd_cg_layout(int t, int hist_len, int64_t* o) {
  if (!o) return fail(GPAMD_EINVAL, "cg_layout: null output");
  o[0] = 0;                                   // bnorm
  o[1] = t;                                   // rnorm
  o[2] = (int64_t)4 * t + 4;                  // alpha_hist
  o[3] = (int64_t)4 * t + 4 + (int64_t)hist_len * t;  // beta_hist
  o[4] = (int64_t)4 * t;                      // stats
  return 0;
}

gpamd_cg_t* gpamd_cg_create_f32(int n, int t, int64_t ld, float* X, float* R, float* D, float* Q, float* Z,
                                float* fscratch, int* iscratch, int hist_len, float eps, float stop_updating_after) {
  if (n <= 0 || t <= 0 || ld % 4 || ld < n || hist_len < 0) {
    fail(GPAMD_EINVAL, "cg_create: bad shape");
    return nullptr;
  }
  gpamd_cg* h = new gpamd_cg;
  CgState<float>& s = h->st;
  s.X = X; s.R = R; s.D = D; s.Q = Q; s.Z = Z;
  s.ld = ld; s.n = n; s.t = t; s.nb = (int)col_blocks(n);
  float* f = fscratch;
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
void gpamd_cg_destroy(gpamd_cg_t* h) { delete h; }
const int* gpamd_cg_done_ptr(const gpamd_cg_t* h) { return h ? h->st.done : nullptr; }

int gpamd_cg_init_f32(gpamd_cg_t* h, const float* B, int64_t ldb, int have_precond, void* stream) {
  if (!h || ldb % 4) return fail(GPAMD_EINVAL, "cg_init: bad arguments");
  CgState<float>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(s.nb, s.t);
  (void)hipMemsetAsync(s.done, 0, 2 * sizeof(int), st);
  hipLaunchKernelGGL((coldot_kernel<float>), grid, dim3(256), 0, st, B, B, ldb, s.n, s.part_a, (const int*)nullptr);
  hipLaunchKernelGGL((cg_init_kernel<float>), grid, dim3(256), 0, st, s, B, ldb, have_precond ? 0 : 1);
  if (!have_precond) hipLaunchKernelGGL((cg_begin_kernel<float>), dim3(s.t), dim3(256), 0, st, s);
  return check_launch("cg_init");
}

// ---- row-sharded solves: gpamd_cg_init_f32 split at the points where the host all-reduces the partial sums -------------
int gpamd_cg_partials_layout(int n, int t, int hist_len, int64_t* offs, int* stride, int* nb) {
  if (!offs || !stride || !nb || n <= 0 || t <= 0 || hist_len < 0) return fail(GPAMD_EINVAL, "cg_partials_layout: bad arguments");
  const int64_t base = (int64_t)4 * t + 4 + (int64_t)2 * hist_len * t;
  offs[0] = base;                               // part_a : d^T q (and ||b||^2 during init)
  offs[1] = base + (int64_t)t * CG_MAXNB;       // part_rz: r^T z
  offs[2] = base + (int64_t)2 * t * CG_MAXNB;   // part_rr: r^T r
  *stride = CG_MAXNB;
  *nb = (int)col_blocks(n);
  return 0;
}

int gpamd_cg_init_norms_f32(gpamd_cg_t* h, const float* B, int64_t ldb, void* stream) {
  if (!h || ldb % 4) return fail(GPAMD_EINVAL, "cg_init_norms: bad arguments");
  CgState<float>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(s.done, 0, 2 * sizeof(int), st);
  hipLaunchKernelGGL((coldot_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, st, B, B, ldb, s.n, s.part_a, (const int*)nullptr);
  return check_launch("cg_init_norms");
}

int gpamd_cg_init_apply_f32(gpamd_cg_t* h, const float* B, int64_t ldb, int copy_d, void* stream) {
  if (!h || ldb % 4) return fail(GPAMD_EINVAL, "cg_init_apply: bad arguments");
  CgState<float>& s = h->st;
  hipLaunchKernelGGL((cg_init_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s, B, ldb, copy_d ? 1 : 0);
  return check_launch("cg_init_apply");
}

int gpamd_cg_begin_apply_f32(gpamd_cg_t* h, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_begin_apply: null handle");
  hipLaunchKernelGGL((cg_begin_kernel<float>), dim3(h->st.t), dim3(256), 0, (hipStream_t)stream, h->st);
  return check_launch("cg_begin_apply");
}

int gpamd_cg_dot_rz_f32(gpamd_cg_t* h, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_dot_rz: null handle");
  CgState<float>& s = h->st;
  hipLaunchKernelGGL((coldot_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s.R, s.Z, s.ld, s.n, s.part_rz,
                     (const int*)s.done);
  return check_launch("cg_dot_rz");
}

int gpamd_cg_update_d_apply_f32(gpamd_cg_t* h, int k, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_update_d_apply: null handle");
  CgState<float>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((cg_update_d_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, st, s, k);
  hipLaunchKernelGGL((cg_stats_kernel<float>), dim3(1), dim3(256), 0, st, s);
  return check_launch("cg_update_d_apply");
}

int gpamd_cg_begin_f32(gpamd_cg_t* h, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_begin: null handle");
  CgState<float>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((coldot_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, st, s.R, s.Z, s.ld, s.n, s.part_rz,
                     (const int*)nullptr);
  hipLaunchKernelGGL((cg_begin_kernel<float>), dim3(s.t), dim3(256), 0, st, s);
  return check_launch("cg_begin");
}

int gpamd_cg_reduce_q_f32(gpamd_cg_t* h, const float* P, int S, int64_t ldp, const float* scale, const float* dscale,
                          const float* dvec, void* stream) {
  if (!h || S <= 0 || ldp % 4) return fail(GPAMD_EINVAL, "cg_reduce_q: bad arguments");
  CgState<float>& s = h->st;
  hipLaunchKernelGGL((kv_reduce_kernel<float, true>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, P, S,
                     (int64_t)s.t * ldp, ldp, scale, dscale, dvec, s.D, s.ld, s.Q, s.ld, s.n, s.part_a, s.done);
  return check_launch("cg_reduce_q");
}

int gpamd_cg_update_xr_f32(gpamd_cg_t* h, int k, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_update_xr: null handle");
  CgState<float>& s = h->st;
  hipLaunchKernelGGL((cg_update_xr_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s, k,
                     s.Z == s.R ? 1 : 0);
  return check_launch("cg_update_xr");
}

int gpamd_cg_update_d_f32(gpamd_cg_t* h, int k, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_update_d: null handle");
  CgState<float>& s = h->st;
  hipStream_t st = (hipStream_t)stream;
  if (s.Z != s.R)
    hipLaunchKernelGGL((coldot_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, st, s.R, s.Z, s.ld, s.n, s.part_rz,
                       (const int*)s.done);
  hipLaunchKernelGGL((cg_update_d_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, st, s, k);
  hipLaunchKernelGGL((cg_stats_kernel<float>), dim3(1), dim3(256), 0, st, s);
  return check_launch("cg_update_d");
}

int gpamd_cg_stop_f32(gpamd_cg_t* h, int k, int min_iter, int tridiag_floor, float tol, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_stop: null handle");
  hipLaunchKernelGGL((cg_stop_kernel<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, h->st, k, min_iter,
                     tridiag_floor, tol);
  return check_launch("cg_stop");
}

int gpamd_cg_finish_f32(gpamd_cg_t* h, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_finish: null handle");
  CgState<float>& s = h->st;
  hipLaunchKernelGGL((cg_finish_kernel<float>), dim3(s.nb, s.t), dim3(256), 0, (hipStream_t)stream, s);
  return check_launch("cg_finish");
}

// ----------------------------------------------------------------------------- pivoted Cholesky
int gpamd_pivoted_cholesky_f32(int kind, float kparam, const float* Xp, int n, int dp, const float* scale, int rank, float tol,
                               float* L, int64_t ldl, int64_t* pivots, float* fwork, int* iwork, void* stream) {
  if (n <= 0 || rank <= 0 || ldl < n) return fail(GPAMD_EINVAL, "pivoted_cholesky: bad shape");
  if (rank > PC_MAX_RANK) return fail(GPAMD_EUNSUPPORTED, "pivoted_cholesky: rank > 512");
  if (rank > n) rank = n;
  hipStream_t st = (hipStream_t)stream;
  PcState s;
  s.dwork = fwork;
  s.scal = fwork + n;
  s.L = L; s.ldl = ldl; s.n = n; s.rank = rank;
  s.pivots = pivots;
  s.ctl = iwork;
  s.perm = iwork + 2;
  s.pos = iwork + 2 + n;
  s.tol = tol;
  s.kparam = kparam;
  (void)hipMemsetAsync(iwork, 0, 2 * sizeof(int), st);
  const int nb = (n + 255) / 256;
  if (n >= 8) {
    // one launch per pivot step: the update of step m and the arg-max for step m + 1 in one kernel (misc_kernels.hpp, "last block decides");
    // the partials live in the former permutation region of iwork (4 nb + 1 <= n words for n >= 8)
    PcPartials pp;
    pp.nb = nb;
    pp.ppos = s.perm;
    pp.pidx = s.perm + nb;
    pp.pval = reinterpret_cast<float*>(s.perm + 2 * nb);
    pp.psum = reinterpret_cast<float*>(s.perm + 3 * nb);
    pp.counter = reinterpret_cast<unsigned*>(s.perm + 4 * nb);
    (void)hipMemsetAsync(pp.counter, 0, sizeof(unsigned), st);
    KIND_SWITCH(kind, hipLaunchKernelGGL((pc_first_kernel<KK>), dim3(nb), dim3(256), 0, st, s, pp, Xp, dp, scale));
    for (int m = 0; m < rank; ++m)
      KIND_SWITCH(kind, hipLaunchKernelGGL((pc_step_kernel<KK>), dim3(nb), dim3(256), 0, st, s, pp, m, Xp, dp, scale));
    return check_launch("pivoted_cholesky");
  }
  hipLaunchKernelGGL(pc_init_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s.perm, s.pos, n);
  // diagonal of the noise-free kernel matrix: scale * k(0)
  KIND_SWITCH(kind, hipLaunchKernelGGL((kernel_diag_kernel<KK>), dim3((n + 255) / 256), dim3(256), 0, st, Xp, Xp, n, dp,
                                       scale, s.dwork, kparam));
  for (int m = 0; m < rank; ++m) {
    hipLaunchKernelGGL(pc_pivot_kernel, dim3(1), dim3(1024), 0, st, s, m);
    KIND_SWITCH(kind, hipLaunchKernelGGL((pc_update_kernel<KK>), dim3((n + 255) / 256), dim3(256), 0, st, s, m, Xp, dp,
                                         scale));
  }
  return check_launch("pivoted_cholesky");
}

// ----------------------------------------------------------------------------- RCCL-communicator variants (multi-GPU hosts without torch)
// librccl is resolved lazily (dlopen / dlsym) so that the library keeps loading on single-GPU hosts without it.
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn rccl_allreduce() {
  // function-local static with a lambda initialiser: initialised exactly once, thread-safe by the language rules (no "tried" flag that a
  // second thread could observe set while the pointer is still null)
  static const nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
    void* hdl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hdl) hdl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return hdl ? reinterpret_cast<nccl_allreduce_fn>(dlsym(hdl, "ncclAllReduce")) : nullptr;
  }();
  return fn;
}
}  // namespace

int gpamd_allreduce_sum_f32(float* buf, int64_t count, void* rccl_comm, void* stream) {
  if (!buf || count <= 0 || !rccl_comm) return fail(GPAMD_EINVAL, "allreduce_sum: bad arguments");
  nccl_allreduce_fn fn = rccl_allreduce();
  if (!fn) return fail(GPAMD_EUNSUPPORTED, "allreduce_sum: librccl.so (ncclAllReduce) not found");
  const int rc = fn(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, rccl_comm, (hipStream_t)stream);
  if (rc != 0) return fail(GPAMD_EINVAL, "allreduce_sum: ncclAllReduce failed");
  return 0;
}

int gpamd_cg_stop_comm_f32(gpamd_cg_t* h, int k, int min_iter, int tridiag_floor, float tol, void* rccl_comm, void* stream) {
  if (!h) return fail(GPAMD_EINVAL, "cg_stop_comm: null handle");
  if (rccl_comm) {
    const int rc = gpamd_allreduce_sum_f32(h->st.stats, 2, rccl_comm, stream);
    if (rc) return rc;
  }
  return gpamd_cg_stop_f32(h, k, min_iter, tridiag_floor, tol, stream);
}

}  // extern "C"
