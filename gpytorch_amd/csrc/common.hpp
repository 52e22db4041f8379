// Shared device helpers for the gfx950 BBMM kernels (wave = 64 lanes, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpamd {

// Covariance families.  The point clouds handed to every kernel are PRE-SCALED by prep_points():
//   RBF     : z = x * sqrt(0.5*log2(e)) / l        ->  k = exp2(-|zi-zj|^2)        (= exp(-0.5 |xi-xj|^2 / l^2))
//   Matern  : z = (x - shift) * sqrt(2 nu) / l     ->  r = |zi-zj|,  k = poly_nu(r) * exp(-r)
// which are the formulas of gpytorch/functions/rbf_covariance.py:14-19 and
// gpytorch/functions/matern_covariance.py:18-50 with the pairwise distance taken directly
// (gpytorch/kernels/keops/rbf_kernel.py:12-15, keops/matern_kernel.py:13-30) instead of through the
// Gram trick of kernels/kernel.py:26-49.
//   RQ      : z = x / (l sqrt(2 alpha))            ->  k = (1 + |zi-zj|^2)^(-alpha)         (gpytorch/kernels/rq_kernel.py:60-74)
//             alpha travels as the runtime shape parameter `p` of the functors below (KvArgs::kparam)
enum Kind : int { KIND_RBF = 0, KIND_MATERN12 = 1, KIND_MATERN32 = 2, KIND_MATERN52 = 3, KIND_RQ = 4 };

constexpr float LOG2E = 1.4426950408889634f;

// Far-pair tile culling (kv_cull.hpp): start of the k-th tile of the contracted cloud a (row block, j chunk) unit visits -- tl: the unit's list of
// surviving tile starts (ascending, terminated by an entry >= the chunk end) or nullptr = every tile of the chunk, jbeg + k * BN.
template <int BN>
__device__ __forceinline__ int tile_start(const int* tl, int jbeg, int k) { return tl ? tl[k] : jbeg + k * BN; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ float cov_from_sq(float s, float p = 0.f) {
  // s = squared distance between pre-scaled points; p = shape parameter (RQ: alpha)
  if constexpr (KIND == KIND_RBF) {
    return __builtin_amdgcn_exp2f(-s);
  } else if constexpr (KIND == KIND_RQ) {
    return __builtin_amdgcn_exp2f(-p * __builtin_amdgcn_logf(1.0f + s));   // v_log_f32 = log2
  } else {
    float r = __builtin_amdgcn_sqrtf(s);
    float e = __builtin_amdgcn_exp2f(-r * LOG2E);
    if constexpr (KIND == KIND_MATERN12) return e;
    if constexpr (KIND == KIND_MATERN32) return (1.0f + r) * e;
    return __builtin_fmaf(s, 1.0f / 3.0f, 1.0f + r) * e;
  }
}

// The same for TWO squared distances at once (Gram-form kernels: S comes from the matrix pipe, possibly a few 1e-6 below zero).  Everything
// but the transcendentals runs on the packed-f32 pipe (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two elements per instruction at the issue
// cost of one), and the clamp of S is a |.| source modifier of v_sqrt_f32 instead of a v_med3_f32 per element: Matern-5/2 2 VALU + 2 transcendental
// instructions per element instead of 5 + 2 (round 4; the split kernel at C3's shape was VALU-bound: 7 VALU + 2 transcendental per element against
// 13-15 MFMAs per 32 x 32 block).  `shift` is added to the exponent (kv_gramh.hpp generates 2^12 K).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__device__ __forceinline__ f32x2 cov_pair_from_sq(f32x2 s, float p = 0.f, float shift = 0.f) {
  if constexpr (KIND == KIND_RBF) {
    return (f32x2){__builtin_amdgcn_exp2f(shift - s[0]), __builtin_amdgcn_exp2f(shift - s[1])};
  } else if constexpr (KIND == KIND_RQ) {
    const f32x2 u = s + 1.0f;   // (a slightly negative s only moves 1 + s by 1e-6)
    const f32x2 l2 = {__builtin_amdgcn_logf(u[0]), __builtin_amdgcn_logf(u[1])};
    const f32x2 t = __builtin_elementwise_fma(l2, (f32x2)(-p), (f32x2)(shift));
    return (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  } else {
    const f32x2 r = {__builtin_amdgcn_sqrtf(__builtin_fabsf(s[0])), __builtin_amdgcn_sqrtf(__builtin_fabsf(s[1]))};
    const f32x2 t = __builtin_elementwise_fma(r, (f32x2)(-LOG2E), (f32x2)(shift));
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    if constexpr (KIND == KIND_MATERN12) return e;
    if constexpr (KIND == KIND_MATERN32) return (r + 1.0f) * e;
    return __builtin_elementwise_fma(s, (f32x2)(1.0f / 3.0f), r + 1.0f) * e;
  }
}

// N pairs at once, STAGE BY STAGE (all square roots, then all exponent multiplies, then all exponentials, ...): written pair by pair the compiler keeps
// the source order -- sqrt, its dependent multiply, exp, its dependent multiply, every instruction waiting on the one before it with an s_nop
// between (133 s_nop per 64 elements in kv_gramv<Matern-5/2, T = 1>, round 6) -- although the N pairs are independent.  Same arithmetic per element as
// cov_pair_from_sq: bitwise the same values.
template <int KIND, int N>
__device__ __forceinline__ void cov_pairs_from_sq(const f32x2 (&s)[N], float p, f32x2 (&k)[N]) {
  if constexpr (KIND == KIND_RBF) {
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = (f32x2){__builtin_amdgcn_exp2f(-s[i][0]), __builtin_amdgcn_exp2f(-s[i][1])};
  } else if constexpr (KIND == KIND_RQ) {
    f32x2 l2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const f32x2 u = s[i] + 1.0f;
      l2[i] = (f32x2){__builtin_amdgcn_logf(u[0]), __builtin_amdgcn_logf(u[1])};
    }
#pragma unroll
    for (int i = 0; i < N; ++i) l2[i] = __builtin_elementwise_fma(l2[i], (f32x2)(-p), (f32x2)(0.f));
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = (f32x2){__builtin_amdgcn_exp2f(l2[i][0]), __builtin_amdgcn_exp2f(l2[i][1])};
  } else {
    f32x2 r[N], t[N], e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = (f32x2){__builtin_amdgcn_sqrtf(__builtin_fabsf(s[i][0])), __builtin_amdgcn_sqrtf(__builtin_fabsf(s[i][1]))};
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = __builtin_elementwise_fma(r[i], (f32x2)(-LOG2E), (f32x2)(0.f));
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = (f32x2){__builtin_amdgcn_exp2f(t[i][0]), __builtin_amdgcn_exp2f(t[i][1])};
    if constexpr (KIND == KIND_MATERN12) {
#pragma unroll
      for (int i = 0; i < N; ++i) k[i] = e[i];
    } else if constexpr (KIND == KIND_MATERN32) {
#pragma unroll
      for (int i = 0; i < N; ++i) k[i] = (r[i] + 1.0f) * e[i];
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = __builtin_elementwise_fma(s[i], (f32x2)(1.0f / 3.0f), r[i] + 1.0f);
#pragma unroll
      for (int i = 0; i < N; ++i) k[i] = r[i] * e[i];
    }
  }
}

// d k / d s (derivative wrt the squared scaled distance), used by the gradient kernels:
//   RBF: k = exp2(-s) = exp(-s ln2)      -> dk/ds = -ln2 * k
//   Matern: with r = sqrt(s): dk/ds = k'(r) / (2 r)
//     nu=1/2: k' = -e^{-r}            -> dk/ds = -e^{-r} / (2r)      (singular at r=0; guarded)
//     nu=3/2: k' = -r e^{-r}          -> dk/ds = -e^{-r} / 2
//     nu=5/2: k' = -(r + r^2)/3 e^{-r}-> dk/ds = -(1 + r) e^{-r} / 6
//   RQ: k = (1+s)^-alpha                 -> dk/ds = -alpha (1+s)^(-alpha-1)
template <int KIND>
__device__ __forceinline__ float dcov_dsq(float s, float p = 0.f) {
  if constexpr (KIND == KIND_RBF) {
    return -0.6931471805599453f * __builtin_amdgcn_exp2f(-s);
  } else if constexpr (KIND == KIND_RQ) {
    return -p * __builtin_amdgcn_exp2f(-(p + 1.0f) * __builtin_amdgcn_logf(1.0f + s));
  } else {
    float r = __builtin_amdgcn_sqrtf(s);
    float e = __builtin_amdgcn_exp2f(-r * LOG2E);
    if constexpr (KIND == KIND_MATERN12) return r > 1e-15f ? -0.5f * e / r : 0.0f;
    if constexpr (KIND == KIND_MATERN32) return -0.5f * e;
    return -(1.0f + r) * e * (1.0f / 6.0f);
  }
}

// ---- float64 exp2 / exp / sqrt of the float64 generation (kv_f64.hpp, the generic row-block kernels).  The float64 kernels are bound by the VALU
// work per covariance value, which issues ALONGSIDE the float64 MFMAs, not under them (profiles/r04_s10_kv_f64_pmc.json: ~5 cycles per VALU
// instruction + 64 per MFMA, additive): the library's exp2 / exp / sqrt carry ~35 / ~35 / ~17 instructions each (special cases, range scaling for
// denormal and huge arguments).  Here the argument is known to be <= 0 (resp. >= 0) and far inside the exponent range, so:
//   exp2(x), x <= 0: n = rint(x), f = x - n in [-1/2, 1/2] (exact), 2^f by a degree-11 polynomial (Chebyshev-node interpolant, 2.0e-17 relative in
//       exact arithmetic, coefficients from mpmath at 60 digits; c0 = 1 exactly so that k(x, x) = 1), v_ldexp_f64 -- 17 instructions;
//   exp(x),  x <= 0: n = rint(x log2 e), f = x - n ln2 with ln2 = hi + lo (hi has 11 trailing zero bits: n hi exact for |n| < 2048), e^f likewise (1.7e-17);
//   sqrt(s), s >= 0: v_rsq_f64 seed, one Goldschmidt step and two residual corrections (the compiler's own expansion without its range scaling).
// No clamp of the argument: v_cvt_i32_f64 saturates (written as an instruction -- the C++ conversion of an out-of-range double is undefined) and
// v_ldexp_f64 with an exponent below -1100 returns zero, so arbitrarily distant pairs give k = 0; NaN arguments stay NaN.  Results within 2-3e-16
// relative of the correctly rounded values (tests/test_gpu_generic.py compares the entries with float64 torch at 1e-12).
__device__ __forceinline__ int cvt_i32_sat_f64(double n) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(n));
  return r;
}
__device__ __forceinline__ double exp2_nonpos_f64(double x) {
  const double n = __builtin_rint(x);
  const double f = x - n;
  double p = 4.4558179083360645e-10;
  p = __builtin_fma(p, f, 7.074194297288521e-09);
  p = __builtin_fma(p, f, 1.0178057087733941e-07);
  p = __builtin_fma(p, f, 1.3215432535912375e-06);
  p = __builtin_fma(p, f, 1.5252733841556773e-05);
  p = __builtin_fma(p, f, 0.00015403530463724353);
  p = __builtin_fma(p, f, 0.001333355814640647);
  p = __builtin_fma(p, f, 0.009618129107587256);
  p = __builtin_fma(p, f, 0.055504108664821625);
  p = __builtin_fma(p, f, 0.24022650695910158);
  p = __builtin_fma(p, f, 0.6931471805599453);
  p = __builtin_fma(p, f, 1.0);
  return __builtin_ldexp(p, cvt_i32_sat_f64(n));
}
__device__ __forceinline__ double exp_nonpos_f64(double x) {
  x = x < -800.0 ? -800.0 : x;   // (n ln2_hi is exact for |n| < 2048 only; e^-800 is zero; a compare-select, not v_max: NaN stays NaN)
  const double n = __builtin_rint(x * 1.4426950408889634);
  double f = __builtin_fma(n, -0x1.62e42fefa3800p-1, x);
  f = __builtin_fma(n, -5.497923018708371e-14, f);
  double p = 2.511003761756205e-08;
  p = __builtin_fma(p, f, 2.763263965412376e-07);
  p = __builtin_fma(p, f, 2.7557240918547625e-06);
  p = __builtin_fma(p, f, 2.480148548228773e-05);
  p = __builtin_fma(p, f, 0.00019841269890047143);
  p = __builtin_fma(p, f, 0.0013888888952314812);
  p = __builtin_fma(p, f, 0.008333333333319601);
  p = __builtin_fma(p, f, 0.04166666666648809);
  p = __builtin_fma(p, f, 0.1666666666666668);
  p = __builtin_fma(p, f, 0.5000000000000019);
  p = __builtin_fma(p, f, 1.0);
  p = __builtin_fma(p, f, 1.0);
  return __builtin_ldexp(p, cvt_i32_sat_f64(n));
}
__device__ __forceinline__ double sqrt_nonneg_f64(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, s);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, s);
  g = __builtin_fma(d, h, g);
  return s > 0.0 ? g : s;          // s = 0: the seed is infinite; NaN stays NaN
}

template <int KIND>
__device__ __forceinline__ double cov_from_sq_f64(double s, double p = 0.0) {
  if constexpr (KIND == KIND_RBF) {
    return exp2_nonpos_f64(-s);
  } else if constexpr (KIND == KIND_RQ) {
    return pow(1.0 + s, -p);
  } else {
    double r = sqrt_nonneg_f64(s);
    double e = exp_nonpos_f64(-r);
    if constexpr (KIND == KIND_MATERN12) return e;
    if constexpr (KIND == KIND_MATERN32) return (1.0 + r) * e;
    return (1.0 + r + s * (1.0 / 3.0)) * e;
  }
}

template <int KIND>
__device__ __forceinline__ double dcov_dsq_f64(double s, double p = 0.0) {
  if constexpr (KIND == KIND_RBF) {
    return -0.6931471805599453 * exp2_nonpos_f64(-s);
  } else if constexpr (KIND == KIND_RQ) {
    return -p * pow(1.0 + s, -p - 1.0);
  } else {
    double r = sqrt_nonneg_f64(s);
    double e = exp_nonpos_f64(-r);
    if constexpr (KIND == KIND_MATERN12) return r > 1e-150 ? -0.5 * e / r : 0.0;
    if constexpr (KIND == KIND_MATERN32) return -0.5 * e;
    return -(1.0 + r) * e * (1.0 / 6.0);
  }
}

// scalar-type dispatch for the kernels templated on T (generic path)
template <int KIND> __device__ __forceinline__ float cov_any(float s, float p = 0.f) { return cov_from_sq<KIND>(s, p); }
template <int KIND> __device__ __forceinline__ double cov_any(double s, double p = 0.0) { return cov_from_sq_f64<KIND>(s, p); }
template <int KIND> __device__ __forceinline__ float dcov_any(float s, float p = 0.f) { return dcov_dsq<KIND>(s, p); }
template <int KIND> __device__ __forceinline__ double dcov_any(double s, double p = 0.0) { return dcov_dsq_f64<KIND>(s, p); }

// Fence between a group of MFMAs whose results land in VGPRs (translation units built with -mllvm -amdgpu-mfma-vgpr-form=1: kvs_*, kvm_*)
// and the first VALU instruction that reads them.  An MFMA -> VALU read-after-write is NOT interlocked by the hardware; the compiler
// inserts the wait states of its hazard table (s_nop 11 for v_mfma_f32_32x32x16_f16), which is NOT enough on gfx950 when the consumer
// follows immediately: kv_gramv_kernel<Matern, D = 3, T = 1> -- v_med3_f32 on the result right after the s_nop -- returned stale values
// in the lanes of the last passes (output rows 16..31 of a tile; 4 % errors on a quarter of the rows, different from run to run;
// profiles/r03_s6_gramv_mfma_vgpr_hazard.txt: wrong with the flag, right with the default AGPR destination -- whose v_accvgpr_read IS
// interlocked -- and right, and 7 % faster, with this fence).  RBF instantiations happened to have enough independent instructions in
// between.  The fence pins the MFMA group (sched_barrier) and idles 32 further wait states once per 32-row j block.
//
// Round 4: the fence must carry DATA dependencies.  `sched_barrier` only binds the machine scheduler; the MFMA builtins are pure
// functions of their operands, so the IR-level passes that run before it (code sinking, instruction combining) are free to move an MFMA
// BELOW a fence that merely sits between it and its reader in the source -- and did: in kv_grad2_kernel<RBF, 3, 0, WSPLIT> the last W MFMA
// and the second Gram MFMA of every j step were emitted AFTER the wait states, back on the compiler's hazard table.  (That was found while
// chasing the 1e-3 .. 4e-3 deviation of the split backward at n = 500 000 and is NOT what caused it: builds with and without the ties give
// bitwise the same sums there, profiles/r04_s1_*, r04_s2_*; the deviation was the round-down accumulation of the f16 matrix pipe, kv_wsplit.hpp.
// It is still exactly the class of defect that produced the 4 % errors above, so the fence closes it by construction.)
// `mfma_result_fence(regs...)` ties every result register to the wait states: an empty
// volatile asm with a "+v" operand BEFORE the s_nops (the producing MFMA must precede it), the same AFTER them (every reader must follow it);
// volatile asms keep their program order among themselves.  The zero-argument form remains for the epilogues of the AGPR-destination
// kernels (kv_gram, kv_gramh, kv_mfma, kv_f64), whose v_accvgpr_read IS interlocked.
template <typename T>
__device__ __forceinline__ void mfma_tie(T& r) { asm volatile("" : "+v"(r)); }
template <typename T, int N>
__device__ __forceinline__ void mfma_tie(T (&r)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(r[k]));
}
template <typename T, int N, int M>
__device__ __forceinline__ void mfma_tie(T (&r)[N][M]) {
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int q = 0; q < M; ++q) asm volatile("" : "+v"(r[k][q]));
}
__device__ __forceinline__ void mfma_result_fence() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15");
  __builtin_amdgcn_sched_barrier(0);
}
template <typename... R>
__device__ __forceinline__ void mfma_result_fence(R&... regs) {
#ifdef GPAMD_FENCE_NO_TIES   // A/B builds only (scripts/grad_at_size_diag.py): the round-3 fence without the data dependencies
  mfma_result_fence();
#else
  __builtin_amdgcn_sched_barrier(0);
  (mfma_tie(regs), ...);
  asm volatile("s_nop 15\n\ts_nop 15");
  (mfma_tie(regs), ...);
  __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* smem4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

}  // namespace gpamd
