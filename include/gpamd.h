/* gpamd.h -- C ABI of the MI355X-native BBMM hot path (libgpamd.so, gfx950 only).
 *
 * The reference (cornellius-gp/gpytorch + the third-party linear_operator package) has NO native
 * boundary: its hot path is Python calling torch.  These entry points are what a ctypes / cffi /
 * torch-extension binding for that path binds instead; each one cites the reference interface it
 * replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; no ownership transfer;
 *   - `stream` is a hipStream_t (NULL = default stream); all calls are asynchronous on it;
 *   - return value: 0 on success, a positive hipError_t, or a negative GPAMD_E* argument error;
 *     gpamd_last_error() returns a thread-local message for the last non-zero return;
 *   - vectors are "probe-major": a block of t right-hand sides is float[t][ld], ld >= n, ld % 4 == 0,
 *     base 16-byte aligned (row c = probe c, contiguous over the n data points);
 *   - point clouds are first converted by gpamd_prep_points_f32 into float[n][dp], dp = 4*ceil(d/4),
 *     pre-scaled so that every kernel evaluates k = f(|zi - zj|^2) only.
 */
#ifndef GPAMD_H
#define GPAMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPAMD_ABI_VERSION 5

/* covariance families: gpytorch/kernels/rbf_kernel.py:68-85, matern_kernel.py:85-110 (nu = 1/2, 3/2, 5/2) */
enum { GPAMD_RBF = 0, GPAMD_MATERN12 = 1, GPAMD_MATERN32 = 2, GPAMD_MATERN52 = 3,
       GPAMD_RQ = 4 /* rational quadratic (gpytorch/kernels/rq_kernel.py:60-74): k = (1 + |z - z'|^2)^-alpha on points prepared as
                       x / (l sqrt(2 alpha)); float32 and float64, any input dimension */ };
/* `kparam`: shape parameter of the parametrised covariance families (RQ: alpha > 0; ignored by the others), an EXPLICIT argument of
 * every entry point that evaluates the covariance or prepares points for it (ABI version 2: the library holds no per-thread kernel
 * state, so operators with different alpha may interleave freely on one thread -- AdditiveKernel(RQ, RQ)).  ABI version 3: the
 * float64 / generic entry points take it too (`double kparam`), so the family runs on every path.  ABI version 4 (additive): the block-Lanczos
 * vector entry points gpamd_block_{project,subtract,transform}_f32, and fused float32 kernels for input dimensions up to 32 (was 16).
 * ABI version 5 (additive): gpamd_kv_partials_far_f32 / gpamd_kv_far_workspace_ints -- the same product with far-pair tile culling --, its
 * derivative twins gpamd_kv_grad2_far_f32 / gpamd_kv_grad_far_f32, and the flag GPAMD_KV_SPLIT_FEW. */

enum { GPAMD_EINVAL = -1, GPAMD_EUNSUPPORTED = -2, GPAMD_EWORKSPACE = -3 };

/* kv flags.  GPAMD_KV_GRAM: the caller asserts max |z|^2 <= 32 over both prepared clouds (after centring) -- or passes block centres X1c, see gpamd_kv_partials_f32 --, so
 * the squared distances may be formed by the quadratic expansion on the matrix pipe (kv_gram.hpp; the expansion
 * the reference itself uses, gpytorch/kernels/kernel.py:26-49) with <= 2e-5 relative error in K (worst case at the limit; typically 5e-6).  Ignored for
 * Matern nu = 1/2. */
enum {
  GPAMD_KV_GRAM = 1,
  GPAMD_KV_WIDE = 2, /* tuning / A-B only: with GRAM, the pre-round-2 selection (no kv_gram4 / kv_gram16) */
  GPAMD_KV_G4 = 4,   /* tuning / A-B only: with GRAM, 9..12 columns on kv_gram4 instead of kv_gram16 */
  /* with GRAM, >= 5 columns: the contraction K * V runs on the f16 matrix pipe at f32 accuracy -- both operands split exactly
   * into f16 hi + lo parts (21-22 significant bits, power-of-two column scaling, f32 accumulation), three
   * v_mfma_f32_32x32x16_f16 in place of eight v_mfma_f32_32x32x2_f32 (kv_gramh.hpp).  The f16 planes of V live behind the
   * partial slabs of the workspace: size it with gpamd_kv_plan called with the same flags and pass the plan's jchunk (a multiple of
   * 128); the workspace must be 16-byte aligned and ldo a multiple of 4. */
  GPAMD_KV_SPLIT = 8,
  /* with GRAM and block centres X1c: the caller bounds the radius of 128-row aligned blocks only (not of the 256 / 512-row blocks the larger
   * row tilings centre): column groups of 5..65 columns with SPLIT run the split kernels at one row tile per wave (128 rows per workgroup),
   * every other column group takes the direct-difference kernels. */
  GPAMD_KV_BLOCK128 = 16,
  /* with SPLIT (ABI version 5): column groups of FEWER than five columns run on the split-operand kernels as well (one 32-column tile, mostly empty)
   * instead of the few-column kernels.  Slower when every tile is visited (the default selection stands for that reason); set by callers of
   * gpamd_kv_partials_far_f32, whose tile lists only the split-operand kernels walk: a one-column product of a numerically sparse K then skips the far
   * tiles too.  gpamd_kv_plan must be called with the same flags (the workspace holds the f16 planes of these groups). */
  GPAMD_KV_SPLIT_FEW = 32
};

int gpamd_abi_version(void);
const char* gpamd_last_error(void);

/* x / lengthscale (+ Matern mean-centring) -- gpytorch/kernels/rbf_kernel.py:78-79,
 * gpytorch/kernels/keops/rbf_kernel.py:45-46, keops/matern_kernel.py:69-71.
 * ls: nls = 1 (isotropic) or d (ARD) lengthscales; shift: d-vector or NULL. */
int gpamd_prep_points_f32(int kind, float kparam, const float* X, int n, int d, int64_t ldx, const float* ls, int nls,
                          const float* shift, float* Xp, int dp, void* stream);

/* Launch plan for one fused K*V: split count S and j-chunk so the grid is a whole number of chip fills of the
 * kernel variant that (kind, d, t, flags) selects.  Outputs on the host. workspace_floats = S * t * ldo (the partial slabs),
 * plus the split operand planes when flags carries GPAMD_KV_SPLIT. */
int gpamd_kv_plan(int kind, int n, int m, int d, int t, int flags, int64_t ldo, int* S_host, int* jchunk_host,
                  int64_t* workspace_floats_host);

/* P[s] = k(X1p, X2p[chunk s]) * Vt[:, chunk s]  for s < S -- the matrix-free K @ V of
 * KernelLinearOperator._matmul (gpytorch/kernels/keops/rbf_kernel.py:44-55) and
 * LazyEvaluatedKernelTensor._matmul (gpytorch/lazy/lazy_evaluated_kernel_tensor.py:245-275).
 * d: input dimension (1..32; the prepared clouds have stride dp = 4*ceil(d/4), 25..32 dimensions stride 32).
 * P: float[S][t][ldo].  done: optional device int; non-zero turns the launch into a no-op.
 * X1c (with GPAMD_KV_GRAM; NULL otherwise): float[ceil(n / 128)][dp], the centre of every 128-row chunk of X1p.  The Gram expansion is
 *   then taken relative to the centre of the workgroup's row block (squared distances are translation invariant), so its cancellation
 *   error scales with the BLOCK radius instead of the cloud radius -- the reference's Gram-trick distance has no scale limit
 *   (gpytorch/kernels/kernel.py:26-49).  The caller sorts the rows of X1p along a space-filling curve so that blocks are compact and
 *   asserts: max over 128 / 256 / 512-row aligned blocks (GPAMD_KV_BLOCK128: 128-row blocks) of |z - centre|^2 <= 32 and
 *   (max |z1| + max |z2|)^2 <= 2.5e7 (RQ: 60000 -- the split norm of a contracted point saturates at 60000, f16 range; every family but the
 *   heavy-tailed RQ is zero to f32 precision long before, gram_f16.hpp).  With X1c = NULL the GPAMD_KV_GRAM contract is the cloud-centred one (max |z|^2 <= 32).
 *   These are HARD preconditions of the flag, not hints: the kernels do not verify them (a check would cost a pass over both clouds per launch) and
 *   outside them entries of K lose accuracy silently.  A caller that cannot guarantee them passes flags without GPAMD_KV_GRAM (direct differences: no
 *   scale limit).  The Python host decides per product in backend.gram_mode / SortedView (tests/test_gpu_recenter.py: clouds at 0.8 of the extent limit). */
int gpamd_kv_partials_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt,
                          int64_t ldv, int t, float* P, int64_t ldo, int S, int jchunk, int flags, const int* done,
                          void* stream);

/* The same product with FAR-PAIR TILE CULLING (opt-in; the reference evaluates every pair -- gpytorch/kernels/keops/rbf_kernel.py:44-55 leaves the
 * reduction over all j to KeOps -- and so does gpamd_kv_partials_f32).  For kernels whose mass sits within a few lengthscales (short lengthscales on
 * a wide cloud: the reference's 3droad workload starts at lengthscale 0.05 on z-scored inputs, examples/02_Scalable_Exact_GPs/KeOps_GP_Regression.ipynb)
 * most 128-point tiles of the contracted cloud hold no covariance above f32 resolution for a given block of output rows.  The caller orders BOTH
 * clouds along a space-filling curve (rows of X1p and of X2p, with Vt's columns in X2p's order) and passes bounding spheres:
 *   row_centres  float[ceil(n / 128)][dp], row_radii float[ceil(n / 128)]:  every 128-row aligned chunk of X1p lies within row_radii[q] of row_centres[q];
 *   tile_centres float[ceil(m / 128)][dp], tile_radii float[ceil(m / 128)]: the same for the 128-point aligned tiles of X2p;
 *   sq_cutoff: a squared distance in PREPARED coordinates (gpamd_prep_points_f32) beyond which the caller accepts k = 0.
 * A workgroup (128 .. 512 output rows: the union of its chunks' spheres) skips every tile of its j chunk whose sphere is farther than sqrt(sq_cutoff)
 * from its own; a skipped tile is neither loaded nor generated.  With sq_cutoff = s(eps), the squared distance at which the family's k falls to
 * eps, every dropped entry of K is <= eps, i.e. |(K V)_ic - culled| <= eps * sum_j |V_jc|.  sq_cutoff <= 0: exactly gpamd_kv_partials_f32 (the
 * other far arguments are ignored).  jchunk must be a multiple of 128 (gpamd_kv_plan's always is).  The spheres are taken on trust like the
 * GPAMD_KV_GRAM preconditions.  A small kernel builds, per column group, the list of surviving tiles of every (row block, j chunk) unit into
 * tile_workspace on the same stream; the product kernels walk the list.  Culled: the column groups that run on the split-operand kernels
 * (GPAMD_KV_SPLIT: kv_gramh.hpp, kv_directh.hpp -- the library's default contraction; groups of fewer than five columns only with
 * GPAMD_KV_SPLIT_FEW); every other group (fp32-MFMA contraction, the few-column kernels) evaluates every tile whatever is passed here -- exact,
 * only not faster.  All other arguments as gpamd_kv_partials_f32. */
int gpamd_kv_partials_far_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt,
                              int64_t ldv, int t, float* P, int64_t ldo, int S, int jchunk, int flags, const int* done, void* stream,
                              const float* row_centres, const float* row_radii, const float* tile_centres, const float* tile_radii, float sq_cutoff,
                              int* tile_workspace, int64_t tile_workspace_ints);
/* ints of tile_workspace for a launch plan (S, jchunk) over n output rows: one list of (jchunk / 128 + 1) tile starts per (row block, j chunk). */
int64_t gpamd_kv_far_workspace_ints(int n, int S, int jchunk);

/* Out = scale * sum_s P[s] + (dscale + dvec) .* Vd  (scale/dscale device scalars, NULL = 1 / 0; dvec: optional
 * float[n] diagonal; Vd may be NULL): ScaleKernel.forward (gpytorch/kernels/scale_kernel.py:117-118) and the
 * + noise of _GaussianLikelihoodBase.marginal (gpytorch/likelihoods/gaussian_likelihood.py:117-121; dvec carries
 * FixedNoiseGaussianLikelihood's heteroskedastic diagonal, gaussian_likelihood.py:245-362). */
int gpamd_kv_reduce_f32(const float* P, int S, int64_t ldp, int t, int n, const float* scale, const float* dscale,
                        const float* dvec, const float* Vd, int64_t ldd, float* Out, int64_t ldo, const int* done,
                        void* stream);

/* One-call  Out = scale * K(X1p, X2p) Vt + dscale * Vd  using caller workspace (>= plan's workspace_floats). */
int gpamd_kv_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Vt, int64_t ldv,
                 int t, const float* scale, const float* dscale, const float* Vd, int64_t ldd, float* Out,
                 int64_t ldo, float* workspace, int64_t workspace_floats, int flags, void* stream);

/* Explicit entries (LinearOperator._getitem / _diagonal / to_dense on a kernel operator):
 * rows: out[r][j] = scale*k(X1p[rows[r]], X2p[j]);  dense: out[i][j] (row-major, ldo);  diag: out[i]. */
int gpamd_kernel_rows_f32(int kind, float kparam, const float* X1p, const int64_t* rows, int nrows, const float* X2p, int m, int dp,
                          const float* scale, float* out, int64_t ldo, void* stream);
int gpamd_kernel_dense_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int dp, const float* scale,
                           float* out, int64_t ldo, void* stream);
int gpamd_kernel_diag_f32(int kind, float kparam, const float* X1p, const float* X2p, int n, int dp, const float* scale, float* out,
                          void* stream);

/* out[c] = sum_i A[c][i] * B[c][i]  (per-column inner products; scratch: float[t*256]) */
int gpamd_coldot_f32(const float* A, const float* B, int64_t ld, int n, int t, float* out, float* scratch,
                     void* stream);

/* ---- modified batched CG (linear_operator.utils.linear_cg; called from
 * gpytorch/distributions/multivariate_normal.py:249 and gpytorch/models/exact_prediction_strategies.py:286,444).
 * The handle only stores pointers into caller-owned device buffers. ---- */
typedef struct gpamd_cg gpamd_cg_t;

/* scratch sizes (in elements) for a t-column solve keeping hist_len iterations of (alpha, beta) */
int64_t gpamd_cg_fscratch_elems(int t, int hist_len);
int64_t gpamd_cg_iscratch_elems(int t);
/* offsets (in elements) of the host-visible pieces inside fscratch: bnorm, rnorm, alpha_hist, beta_hist, stats */
int gpamd_cg_layout(int t, int hist_len, int64_t* offsets5_host);

/* X, R, D, Q: float[t][ld].  Z: float[t][ld] preconditioned residual, or == R without a preconditioner.
 * iscratch holds [zero_rhs(t) | converged(t) | done(2)]; done = {flag, iterations}. */
gpamd_cg_t* gpamd_cg_create_f32(int n, int t, int64_t ld, float* X, float* R, float* D, float* Q, float* Z,
                                float* fscratch, int* iscratch, int hist_len, float eps, float stop_updating_after);
void gpamd_cg_destroy(gpamd_cg_t* h);
const int* gpamd_cg_done_ptr(const gpamd_cg_t* h);

/* R = B/|B|, X = 0 (and D = R, rho = r.r when have_precond == 0) */
int gpamd_cg_init_f32(gpamd_cg_t* h, const float* B, int64_t ldb, int have_precond, void* stream);
/* with a preconditioner: after the caller wrote Z = P^-1 R and D = Z: rho = r.z */
int gpamd_cg_begin_f32(gpamd_cg_t* h, void* stream);
/* Row-sharded solves (each rank owns a contiguous block of rows of K_hat; used for the small-t solves
 * -- predictive-mean CG, Lanczos -- where probe-column sharding has nothing to split).  The solver's inner products
 * become sums over ranks: the three per-column partial arrays live in fscratch at offs[0..2] (d^T q, r^T z, r^T r), each
 * [t][stride] with nb used entries per column; between the producing and the consuming call the host sums a column's
 * partials, all-reduces the t sums (RCCL), writes them to entry 0 and zeroes the rest.  gpamd_cg_init_f32 is split at
 * those points: init_norms (-> offs[0]) | init_apply (-> offs[2], and offs[1] when copy_d) | begin_apply. */
int gpamd_cg_partials_layout(int n, int t, int hist_len, int64_t* offs, int* stride, int* nb);
int gpamd_cg_init_norms_f32(gpamd_cg_t* h, const float* B, int64_t ldb, void* stream);
int gpamd_cg_init_apply_f32(gpamd_cg_t* h, const float* B, int64_t ldb, int copy_d, void* stream);
int gpamd_cg_begin_apply_f32(gpamd_cg_t* h, void* stream);
/* preconditioned row-sharded solves: gpamd_cg_begin_f32 / gpamd_cg_update_d_f32 split at the r^T z partial sums --
 * dot_rz writes the per-workgroup partials of r^T z (layout: gpamd_cg_partials_layout, offs[1]); the host all-reduces them;
 * begin_apply / update_d_apply consume them. */
int gpamd_cg_dot_rz_f32(gpamd_cg_t* h, void* stream);
int gpamd_cg_update_d_apply_f32(gpamd_cg_t* h, int k, void* stream);
/* Q = scale * sum_s P[s] + (dscale + dvec) .* D, and d.q partials */
int gpamd_cg_reduce_q_f32(gpamd_cg_t* h, const float* P, int S, int64_t ldp, const float* scale, const float* dscale,
                          const float* dvec, void* stream);
/* alpha; X += alpha D; R -= alpha Q.   k (here, in update_d and in stop): the iteration index, or -1 = take it from the device
 * (the count cg_stop maintains) -- the form to record into a hipGraph that is then replayed once per iteration. */
int gpamd_cg_update_xr_f32(gpamd_cg_t* h, int k, void* stream);
/* (caller applies Z = P^-1 R here when preconditioned)  beta; D = Z + beta D; residual statistics -> stats */
int gpamd_cg_update_d_f32(gpamd_cg_t* h, int k, void* stream);
/* stopping rule on stats (all-reduce stats[0:2] across ranks first when probe columns are sharded) */
int gpamd_cg_stop_f32(gpamd_cg_t* h, int k, int min_iter, int tridiag_floor, float tol, void* stream);
/* ---- RCCL-communicator variants ("RCCL communicator handle passed in for the multi-GPU variants"; they replace the peer
 * copies of gpytorch/kernels/multi_device_kernel.py:49-92).  rccl_comm: an ncclComm_t of the caller (one rank per GPU); stream-ordered on
 * `stream`, no host round trip.  librccl.so is resolved on first use (dlopen), so single-GPU hosts need not have it.
 *   gpamd_cg_stop_comm_f32: the stopping rule of a PROBE-SHARDED solve -- all-reduces the solver's two residual statistics (sum of column
 *     residual norms, column count) over the communicator, then applies the rule: the ONLY per-iteration collective of the design.
 *   gpamd_allreduce_sum_f32: in-place sum of `count` device floats (SLQ partial sums, packed hyper-parameter gradients, the k x t
 *     preconditioner coefficients of a row-sharded apply).
 * The Python host of this repository drives the same collectives through torch.distributed (backend "nccl" = RCCL), which owns its
 * communicators; these entry points serve hosts that hold an ncclComm_t themselves. ---- */
int gpamd_cg_stop_comm_f32(gpamd_cg_t* h, int k, int min_iter, int tridiag_floor, float tol, void* rccl_comm, void* stream);
int gpamd_allreduce_sum_f32(float* buf, int64_t count, void* rccl_comm, void* stream);
/* X *= |B| */
int gpamd_cg_finish_f32(gpamd_cg_t* h, void* stream);

/* ---- pivoted Cholesky of the NOISE-FREE kernel matrix (LinearOperator.pivoted_cholesky; wrapper
 * gpytorch/__init__.py:146-173; consumer AddedDiagLinearOperator._preconditioner). L: float[rank][ldl]
 * (zero-filled by the caller), rank <= 512; fwork: float[n + 4]; iwork: int[2 + 2n]; pivots: int64[rank].
 * Runs `rank` (pivot, update) steps without host synchronisation; steps after the error tolerance is
 * met are no-ops.  On completion iwork[0] = number of columns produced. ---- */
int gpamd_pivoted_cholesky_f32(int kind, float kparam, const float* Xp, int n, int dp, const float* scale, int rank, float tol,
                               float* L, int64_t ldl, int64_t* pivots, float* fwork, int* iwork, void* stream);

/* ---- Lanczos tridiagonalisation with full re-orthogonalisation: the vector work of one step (float32 vectors of length n,
 * basis Q [k][ldq] probe-major).  Replaces the torch GEMV / norm chain of linear_operator.utils.lanczos.lanczos_tridiag
 * (reached from gpytorch/models/exact_prediction_strategies.py:202,234-238,271); the operator product w = K_hat q is
 * gpamd_kv_partials_f32 + gpamd_kv_reduce_f32 with t = 1.  All scalars stay on the device; partial sums are
 * [k][gpamd_lanczos_partial_stride()] floats with gpamd_lanczos_num_partials(n) valid entries per row, finished in a fixed order
 * by gpamd_lanczos_coef_f32 (row-sharded callers all-reduce `coef` between coef and subtract / normalize).
 *   residual : r = w - beta_prev[0] * q_prev            (q_prev == NULL: r = w)
 *   project  : part[m][b] = partial <Q[m], r>, m < k <= 512
 *   coef     : coef[m] = sum_b part[m][b]; tol >= 0: flag[0] |= (|coef[m]| > tol)
 *   subtract : r -= sum_m coef[m] Q[m]; part_rr[b] = partial |r|^2       (finish with gpamd_lanczos_coef_f32(part_rr, 1, ...))
 *   normalize: out = r / sqrt(rr[0]); norm_out[0] = sqrt(rr[0]); stop[0] |= (norm < tiny) ---- */
int gpamd_lanczos_num_partials(int n);
int gpamd_lanczos_partial_stride(void);
int gpamd_lanczos_residual_f32(const float* w, const float* q_prev, const float* beta_prev, float* r, int n, void* stream);
int gpamd_lanczos_project_f32(const float* Q, int64_t ldq, int k, const float* r, int n, float* part, void* stream);
int gpamd_lanczos_coef_f32(const float* part, int k, int nb, float tol, float* coef, int* flag, void* stream);
int gpamd_lanczos_subtract_f32(const float* Q, int64_t ldq, int k, const float* coef, float* r, int n, float* part_rr, void* stream);
int gpamd_lanczos_normalize_f32(const float* r, int n, const float* rr, float* out, float* norm_out, float tiny, int* stop, void* stream);

/* ---- BLOCK Lanczos with full re-orthogonalisation: the vector work of one step for a block of b <= 32 probe-major rows R [b][ldr] against a basis
 * Q [k][ldq] of ANY length k (float32 vectors, float64 accumulation).  Behind LinearOperator.root_inv_decomposition (the LOVE covar_cache,
 * gpytorch/models/exact_prediction_strategies.py:267-272) and the multi-vector form of gpytorch.root_inv_decomposition (gpytorch/__init__.py:190-216): the b
 * vectors ride through ONE b-column gpamd_kv_partials_f32 per step instead of b single-column products.
 *   project  : W[c][m] = <R[c], Q[m]>                  W: double [b][k]; ANY b (column groups of 16); workspace: double[gpamd_precond_coef_workspace_doubles(n, min(b, 16), k)].
 *              With R = Q it is the Gram matrix of tall-skinny rows -- the two Gram products of the preconditioner's Cholesky-QR
 *              (AddedDiagLinearOperator._preconditioner's QR of [L; sigma I]; rocBLAS' float64 GEMM takes 35 ms for the 15 x 217 437 x 15 shape); _f64: both
 *              operands double (the second Cholesky-QR pass)
 *   subtract : R[c] -= sum_m W[c][m] Q[m]              (in place)
 *   transform: R[r] = sum_c M[r][c] R[c]               (in place; M: double [b][b], e.g. the inverse Cholesky factor of R R^T: Cholesky-QR) ---- */
int gpamd_block_project_f32(const float* Q, int64_t ldq, int k, const float* R, int64_t ldr, int b, int n, double* W, double* workspace,
                            int64_t workspace_doubles, void* stream);
int gpamd_block_project_f64(const double* Q, int64_t ldq, int k, const double* R, int64_t ldr, int b, int n, double* W, double* workspace,
                            int64_t workspace_doubles, void* stream);
int gpamd_block_subtract_f32(const float* Q, int64_t ldq, int k, const double* W, float* R, int64_t ldr, int b, int n, void* stream);
int gpamd_block_transform_f32(const double* M, float* R, int64_t ldr, int b, int n, void* stream);

/* ---- multi-shift MINRES (contour-integral quadrature: gpytorch.sqrt_inv_matmul, gpytorch/__init__.py:252-278 ->
 * linear_operator.utils.minres; consumer variational/ciq_variational_strategy.py:217): the vector part of ONE iteration for all Q shifts:
 * d = (v_c - delta d1 - eps d2) / gamma written over d2 (the caller swaps the two direction buffers), x += tau d.
 * v: [t][ld]; d1, d2, x: [Q][t][ld]; coef: float[4][Q][t] = delta | eps | 1 / gamma | tau (device). ---- */
int gpamd_msminres_update_f32(const float* v, const float* d1, float* d2, float* x, const float* coef, int Q, int t, int n, int64_t ld,
                              void* stream);

/* ---- pivoted-Cholesky preconditioner apply, first half:  W[c][m] = sum_i R[c][i] * Q1[m][i]  with R float32 [t][ldr] (the CG
 * residuals), Q1 float64 [k][ldq] (k <= 512), W float64 [t][k], float64 accumulation -- the k x t coefficients of
 * AddedDiagLinearOperator._preconditioner's closure  P^-1 R = (R - Q1 Q1^T R) / sigma^2, whose cancellation float32 cannot carry
 * (gpytorch_amd/linear_cg.py).  workspace: double[gpamd_precond_coef_workspace_doubles(n, t, k)]. ---- */
int64_t gpamd_precond_coef_workspace_doubles(int n, int t, int k);
int gpamd_precond_coef_f32f64(const float* R, int64_t ldr, int t, const double* Q, int64_t ldq, int k, int n, double* W,
                              double* workspace, int64_t workspace_doubles, void* stream);
/* second half, fused:  Out[c][i] = (R[c][i] - sum_m W[c][m] Q1[m][i]) / sigma2[0]  (float64 arithmetic, float32 in / out; Out may
 * alias nothing but itself -- R and Out are different buffers in the mBCG state).  sigma2: one float on the device. */
int gpamd_precond_apply_f32f64(const float* R, int64_t ldr, int t, const double* Q, int64_t ldq, int k, int n, const double* W,
                               const float* sigma2, float* Out, int64_t ldo, void* stream);

/* ---- fused bilinear derivative: out[0] = sum_ij W_ij k_ij, out[1+q] = sum_ij W_ij dk/ds_ij (z_iq - z_jq)^2,
 * W = Lt^T Rt (never formed).  Replaces LinearOperator._bilinear_derivative on the kernel operator and the
 * dense backward of gpytorch/functions/rbf_covariance.py:26-29 / matern_covariance.py:53-56 (chunked variant:
 * gpytorch/lazy/lazy_evaluated_kernel_tensor.py:69-104).  out: float[1 + dp]; workspace: double[>= the query].
 * iso != 0 (single lengthscale): out[1] = sum_ij W_ij dk/ds_ij s_ij and out[2..] = 0 (cheaper epilogue). ---- */
int64_t gpamd_kv_grad_workspace_doubles(int n, int m, int t, int dp);
int gpamd_kv_grad_f32(int kind, const float* X1p, int n, const float* X2p, int m, int dp, const float* Lt, int64_t ldl,
                      const float* Rt, int64_t ldr, int t, int iso, float* out, double* workspace,
                      int64_t workspace_doubles, void* stream);

/* ---- the same derivative with Gram-form generation (squared distances on the matrix pipe, kv_grad2.hpp) and, optionally,
 * the gradient with respect to the PREPARED left points:  Gz1t[q][i] = sum_j W_ij dk/ds_ij * 2 (z_iq - z_jq)  (probe-major
 * [d][ldg]; the caller applies dz/dx = coef / lengthscale_q and theta -- the input gradients the KeOps precedent provides,
 * gpytorch/test/base_keops_test_case.py:105-132; the reference's dense Functions refuse them, rbf_covariance.py:9-10).
 * RBF / Matern 3/2 / Matern 5/2 / RQ only, accurate while max |z|^2 <= 32 or, with block centres X1c (as for gpamd_kv_partials_f32), while the
 * block radius^2 <= 8 (the host's policy for every Gram-form kernel); d = valid
 * dimensions (points are [n][round_up(d,4)]).  out: float[2 + round_up(d,4)]: [0 .. dp] as gpamd_kv_grad_f32, [1 + dp] = sum_ij W_ij dk/dp_ij at
 * fixed s for the family's shape parameter (RQ alpha; 0 otherwise).  Gz1t == NULL: hyper-parameters only (xworkspace unused). ---- */
int64_t gpamd_kv_grad2_workspace_doubles(int n, int m, int t, int d);
int64_t gpamd_kv_grad2_xworkspace_floats(int n, int m, int t, int d);
/* flags & GPAMD_KV_SPLIT: the W = L^T R contraction runs on the f16 matrix pipe at f32 accuracy (hi/lo-split planes of both vector blocks,
 * per-column power-of-two scales with a constant product; kv_wsplit.hpp) -- 15 v_mfma_f32_32x32x16_f16 per 32 x 32 tile instead of 33
 * v_mfma_f32_32x32x2_f32 at 65 columns.  Needs sworkspace (16-byte aligned, gpamd_kv_grad2_split_workspace_floats floats); the plain
 * workspaces sized for the same t cover either path. */
int64_t gpamd_kv_grad2_split_workspace_floats(int n, int m);
int gpamd_kv_grad2_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Lt, int64_t ldl,
                       const float* Rt, int64_t ldr, int t, int iso, float* out, float* Gz1t, int64_t ldg, double* workspace,
                       int64_t workspace_doubles, float* xworkspace, int64_t xworkspace_floats, int flags, float* sworkspace,
                       int64_t sworkspace_floats, void* stream);

/* The two derivative kernels with FAR-PAIR TILE CULLING (ABI version 5; contract, spheres and bound as gpamd_kv_partials_far_f32: both clouds and their
 * vector blocks in curve order, bounding spheres of the 128-point chunks, sq_cutoff the squared prepared distance beyond which k -- and with it dk/ds --
 * is accepted as zero; sq_cutoff <= 0: exactly the un-culled entry points).  A (128-row block, j chunk) unit skips the 64-row j steps whose 128-point
 * tile lies farther than sqrt(sq_cutoff) from its rows: neither the W tile nor the covariance derivative of a skipped step is formed.  The sums are
 * order-free, so the caller need not take anything back to the original order (Gz1t comes out in X1p's order, as always).
 * tile_workspace: gpamd_kv_grad2_far_workspace_ints(n, m) / gpamd_kv_grad_far_workspace_ints(n, m) ints. */
int64_t gpamd_kv_grad2_far_workspace_ints(int n, int m);
int gpamd_kv_grad2_far_f32(int kind, float kparam, const float* X1p, int n, const float* X2p, int m, int d, const float* X1c, const float* Lt, int64_t ldl,
                           const float* Rt, int64_t ldr, int t, int iso, float* out, float* Gz1t, int64_t ldg, double* workspace,
                           int64_t workspace_doubles, float* xworkspace, int64_t xworkspace_floats, int flags, float* sworkspace,
                           int64_t sworkspace_floats, void* stream, const float* row_centres, const float* row_radii, const float* tile_centres,
                           const float* tile_radii, float sq_cutoff, int* tile_workspace, int64_t tile_workspace_ints);
int64_t gpamd_kv_grad_far_workspace_ints(int n, int m);
int gpamd_kv_grad_far_f32(int kind, const float* X1p, int n, const float* X2p, int m, int dp, const float* Lt, int64_t ldl,
                          const float* Rt, int64_t ldr, int t, int iso, float* out, double* workspace,
                          int64_t workspace_doubles, void* stream, const float* row_centres, const float* row_radii, const float* tile_centres,
                          const float* tile_radii, float sq_cutoff, int* tile_workspace, int64_t tile_workspace_ints);

/* ---- batches of SMALL independent GPs (gpytorch/kernels/kernel.py:163-208 batch_shape; test/examples/test_batch_gp_regression.py).
 * Members below settings.max_cholesky_size are factorised, not iterated: what the member loop costs there is launches.  These two
 * entry points make the launch count independent of the batch size (blockIdx.z = member).  b members of n (resp. m) prepared points
 * each, stored back to back: X1p [b][n][dp], X2p [b][m][dp] (gpamd_prep_points_f32 per member, or the same arithmetic by the
 * caller); kparam [b] (RQ; NULL otherwise), scale [b] or NULL, dadd [b] or NULL (added on the diagonal i == j: K_hat in one pass).
 * out [b][n][ldo]:  out[g][i][j] = scale[g] k(x1_gi, x2_gj) + [i == j] dadd[g]   -- replaces the batched `Kernel.forward` +
 * `to_dense()` + `add_diagonal` of a batch-mode ExactGP (kernels/kernel.py:318-330, likelihoods/gaussian_likelihood.py:117-121). */
int gpamd_kernel_dense_batched_f32(int kind, const float* kparam, const float* X1p, int n, const float* X2p, int m, int dp, int b,
                                   const float* scale, const float* dadd, float* out, int64_t ldo, void* stream);
/* Batched bilinear derivative (the kernel backward of every member in one launch; functions/rbf_covariance.py:26-29,
 * matern_covariance.py:53-56): W [b][n][ldw] = d loss / d K_g;  G [b][2 + dp] (device doubles, zeroed here):
 * G[g][0] = sum W k,  G[g][1 + q] = sum W dk/ds (z_iq - z_jq)^2,  G[g][1 + dp] = sum W dk/dkparam at fixed s (RQ; 0 otherwise). */
int gpamd_kernel_grad_batched_f32(int kind, const float* kparam, const float* X1p, int n, const float* X2p, int m, int dp, int b,
                                  const float* W, int64_t ldw, double* G, void* stream);

/* ---- float64 (the reference honours float64 inputs).  Same conventions with double buffers.  The fused MFMA K*V
 * kernels are float32-only; in float64 K @ V is formed from dense row blocks of K generated by
 * gpamd_kernel_rows_f64 (rows == NULL: the block [row0, row0 + nrows)) times V with a library DGEMM by the caller,
 * then fed to the same device-resident mBCG through gpamd_cg64_reduce_q (S = 1). ---- */
int gpamd_prep_points_f64(int kind, double kparam, const double* X, int n, int d, int64_t ldx, const double* ls, int nls,
                          const double* shift, double* Xp, int dp, void* stream);
int gpamd_kernel_rows_f64(int kind, double kparam, const double* X1p, const int64_t* rows, int64_t row0, int nrows, const double* X2p, int m,
                          int dp, const double* scale, double* out, int64_t ldo, void* stream);
int gpamd_kernel_diag_f64(int kind, double kparam, const double* X1p, const double* X2p, int n, int dp, const double* scale, double* out,
                          void* stream);
/* Fused float64 K*V (kv_f64.hpp: float64 generation on the VALU, contraction on v_mfma_f64_16x16x4_f64 -- column groups of <= 4 columns on the
 * VALU: on gfx950 a float64 MFMA runs at the VALU's float64 rate and a 16-column tile would be mostly empty; d <= 16, else
 * GPAMD_EUNSUPPORTED and the caller uses the row-block path).  Same partial-slab convention as gpamd_kv_partials_f32:
 * P [S][t][ldo], to be summed / scaled by gpamd_kv_reduce_f64 or gpamd_cg64_reduce_q.  Replaces
 * KernelLinearOperator._matmul / LazyEvaluatedKernelTensor._matmul (lazy_evaluated_kernel_tensor.py:245-275) for
 * float64 models. */
int gpamd_kv_plan_f64(int n, int m, int dp, int t, int64_t ldo, int* S, int* jchunk, int64_t* workspace_doubles);
int gpamd_kv_partials_f64(int kind, double kparam, const double* X1p, int n, const double* X2p, int m, int dp, const double* Vt, int64_t ldv,
                          int t, double* P, int64_t ldo, int S, int jchunk, const int* done, void* stream);
/* Generic-path bilinear derivative, one row block (any dp; replaces, for float64 / d > 16, what gpamd_kv_grad_f32 does
 * fused; reference: the kernel backward, gpytorch/functions/rbf_covariance.py:26-29, matern_covariance.py:53-56).
 * W [nrows, ldw]: left^T right on entry, W * dk/ds on exit (s = squared prepared distance);
 * acc[0] += sum W * k and, for a family with a shape parameter (RQ), acc[1] += sum W * dk/dkparam at fixed s
 * (acc: TWO device doubles, zeroed by the caller). */
int gpamd_kernel_grad_block_f32(int kind, double kparam, const float* X1p, int64_t row0, int nrows, const float* X2p, int m, int dp, float* W,
                                int64_t ldw, double* acc, void* stream);
int gpamd_kernel_grad_block_f64(int kind, double kparam, const double* X1p, int64_t row0, int nrows, const double* X2p, int m, int dp, double* W,
                                int64_t ldw, double* acc, void* stream);
int gpamd_coldot_f64(const double* A, const double* B, int64_t ld, int n, int t, double* out, double* scratch, void* stream);
int gpamd_kv_reduce_f64(const double* P, int S, int64_t ldp, int t, int n, const double* scale, const double* dscale,
                        const double* dvec, const double* Vd, int64_t ldd, double* Out, int64_t ldo, const int* done,
                        void* stream);
typedef struct gpamd_cg64 gpamd_cg64_t;
int64_t gpamd_cg64_fscratch_elems(int t, int hist_len); /* layout / iscratch: as the float32 solver */
gpamd_cg64_t* gpamd_cg64_create(int n, int t, int64_t ld, double* X, double* R, double* D, double* Q, double* Z,
                                double* fscratch, int* iscratch, int hist_len, double eps, double stop_updating_after);
void gpamd_cg64_destroy(gpamd_cg64_t* h);
int gpamd_cg64_init(gpamd_cg64_t* h, const double* B, int64_t ldb, int have_precond, void* stream);
int gpamd_cg64_begin(gpamd_cg64_t* h, void* stream);
int gpamd_cg64_reduce_q(gpamd_cg64_t* h, const double* P, int S, int64_t ldp, const double* scale, const double* dscale,
                        const double* dvec, void* stream);
int gpamd_cg64_update_xr(gpamd_cg64_t* h, int k, void* stream);
int gpamd_cg64_update_d(gpamd_cg64_t* h, int k, void* stream);
int gpamd_cg64_stop(gpamd_cg64_t* h, int k, int min_iter, int tridiag_floor, double tol, void* stream);
int gpamd_cg64_finish(gpamd_cg64_t* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPAMD_H */
