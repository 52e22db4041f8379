// Fused covariance-generation + contraction:  P[s] = k(X1, X2[chunk s]) * V[chunk s]   (fp32 MFMA)
//
// Replaces, for the matrix-free path, the reference's
//   KernelLinearOperator._matmul = covar_func(x1, x2) @ rhs  (gpytorch/kernels/keops/rbf_kernel.py:44-55)
//   LazyEvaluatedKernelTensor._matmul (chunked)              (gpytorch/lazy/lazy_evaluated_kernel_tensor.py:245-275)
//   DenseLinearOperator._matmul = torch.matmul(K, V) on the materialised n x n kernel matrix.
// K is never formed: every lane generates exactly the K(x_j, x_i) element its MFMA B-operand slot
// needs, so kernel generation (VALU + transcendental pipe) runs in the shadow of the matrix pipe.
//
// Layout (all fp32, "probe-major"): V and P hold one probe / right-hand-side vector per ROW
//   Vt[c * ldv + j], c < t, j < m          P[s][c * ldo + i], c < t, i < n
// so that the contraction is  P^T[c][i] = sum_j Vt[c][j] K[j][i]:
//   MFMA A (32 x 2) = Vt tile   : lane l supplies A[c = l&31][k = l>>5]      (read from LDS, ds_read_b128)
//   MFMA B (2 x 32) = K tile    : lane l supplies B[k = l>>5][i = l&31]      (computed in registers)
//   MFMA D (32 x 32)            : D[c = (r&3) + 8(r>>2) + 4(l>>5)][i = l&31] (128-B coalesced row stores)
// v_mfma_f32_32x32x2_f32 is bit-exact fp32 (an fmaf chain), 64 cycles/SIMD, 157.3 TFLOP/s chip peak.
//
// Work decomposition: one workgroup = 4 waves (one per SIMD) owns BM = 4*NI*32 rows i and one chunk
// of j ("split-j", S chunks) so the grid can be sized to a whole number of chip-fills; the S partial
// slabs are summed by the consumer kernel (kv_reduce / cg_reduce_q), which is a <1% HBM-bound pass.
// Columns: CT tiles of 32 probe columns on the matrix pipe, plus EX (0/1) extra column carried on the
// VALU (k * e_j fmac on the already-generated K element) -- the [Z | y] right-hand side of the MLL is
// 64+1 columns and would otherwise waste a third 32-wide tile.
#pragma once
#include "common.hpp"

namespace gpamd {

struct KvArgs {
  const float* X1;   // [n][DP] prepared points (rows of the output)
  const float* X2;   // [m][DP] prepared points (contracted index)
  const float* Vt;   // [t][ldv]
  float* P;          // [S][t][ldo] partial outputs
  int64_t ldv, ldo, pstride;
  int n, m, t;
  int S, jchunk, nrb;  // split count, j-chunk length (multiple of BN), row-block count
  const int* done;     // optional device flag: non-zero -> the launch is a no-op (converged CG)
  float kparam;        // shape parameter of the covariance family (RQ: alpha); 0 otherwise
  const float* Xc = nullptr;   // Gram-form kernels: optional [ceil(n / 128)][DP] chunk centres of X1 (gram_f16.hpp) or nullptr
  // Far-pair tile culling (opt-in, settings.far_pair_cutoff; off = the reference's arithmetic: every pair is evaluated).  tiles != nullptr: the
  // workgroup of unit u = s * nrb + rb visits only the tiles listed in tiles[u * tpc1 ..] (tile starts j0, ascending, terminated by an entry
  // >= jend) instead of jbeg, jbeg + BN, ...: cull_list_kernel (below) builds the lists from the bounding spheres of both clouds right before the
  // product.  The tile loops carry ONE scalar pointer for it -- the sphere test itself lives in the list kernel: held in the product kernels it
  // cost the split kernels 8-16 VGPRs (spills at two waves per SIMD) and the few-column kernels a resident wave, culling or not.
  // Read by the split-operand kernels only (kv_gramh.hpp, kv_directh.hpp: >= 5 columns on the library's default contraction): their look-ahead
  // staging already asks "which tile next".  The fp32-MFMA kernels and the few-column kernels visit every tile whatever is passed here -- reading
  // j0 from a list broke the strength reduction of kv_gram's staging addresses (300 bytes of scratch per lane at three waves per SIMD), and any
  // edit of kv_gramv's loop moved the allocator's occupancy choice for a dozen instantiations (3 -> 1 waves at Matern-5/2 d = 10, four columns).
  const int* tiles = nullptr;
  int tpc1 = 0;                // entries per unit: tiles per j chunk + 1 (the terminator)
};

// start of the k-th tile a workgroup visits (tl: its list or nullptr = every tile of the chunk)
template <int BN>
__device__ __forceinline__ int kv_tile_at(const int* tl, int jbeg, int k) { return tile_start<BN>(tl, jbeg, k); }
__device__ __forceinline__ const int* kv_tile_list(const KvArgs& a, int unit) { return a.tiles ? a.tiles + (int64_t)unit * a.tpc1 : nullptr; }

constexpr int KV_BN = 128;           // j-tile staged in LDS per iteration
constexpr int KV_LDT = KV_BN + 4;    // padded LDS row (keeps 16-B alignment, conflict-free b128 reads)

template <int KIND, int D, int CT, int NI, int EX>
__global__ __launch_bounds__(256) void kv_mfma_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4;  // storage stride; only the D valid dimensions are evaluated
  constexpr int BN = KV_BN, LDT = KV_LDT, TC = 32 * CT;
  constexpr int DQ = DP / 4;
  __shared__ __attribute__((aligned(16))) float smem[TC * LDT + BN * DP + BN];
  float* Vs = smem;
  float* Xs = smem + TC * LDT;
  float* Es = Xs + BN * DP;

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 32) + wave * (NI * 32);

  // own points x_i (both half-waves hold the same 32 rows)
  float xi[NI][DP];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int i = min(ibase + ni * 32 + l31, a.n - 1);
    const f32x4* p = reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = p[q];
      xi[ni][4 * q + 0] = v[0]; xi[ni][4 * q + 1] = v[1]; xi[ni][4 * q + 2] = v[2]; xi[ni][4 * q + 3] = v[3];
    }
  }

  f32x16 acc[NI][CT];
  float eacc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    eacc[ni] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ct][r] = 0.f;
  }

  // staging: global -> registers -> LDS between the two barriers of a tile.  Measured (profiles/
  // r01_s4_kv_tune_variants.jsonl): prefetching the next tile into registers across the MFMA phase costs
  // ~30 VGPRs and one resident wave per SIMD; three waves/SIMD with synchronous staging is faster (+3 %).
  constexpr int VQ = TC * (BN / 4) / 256;  // float4 per thread for the V tile (= 8*CT)
  constexpr int VCH = VQ < 8 ? VQ : 8;     // staged in chunks of <= 8 float4 (32 VGPRs) per thread
  constexpr int XQ = (BN * DQ + 255) / 256;

  auto stage_tile = [&](int j0) {
#pragma unroll
    for (int r0 = 0; r0 < VQ; r0 += VCH) {
      f32x4 vreg[VCH];
#pragma unroll
      for (int rr = 0; rr < VCH; ++rr) {
        const int idx = tid + 256 * (r0 + rr);
        const int c = idx / (BN / 4), q = idx % (BN / 4);
        const int j = j0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < VQ && c < a.t) {
          const float* src = a.Vt + (int64_t)c * a.ldv + j;
          if (j + 4 <= jend) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < jend) v[e] = src[e];
          }
        }
        vreg[rr] = v;
      }
#pragma unroll
      for (int rr = 0; rr < VCH; ++rr) {
        const int idx = tid + 256 * (r0 + rr);
        const int c = idx / (BN / 4), q = idx % (BN / 4);
        if (r0 + rr < VQ) *reinterpret_cast<f32x4*>(&Vs[c * LDT + 4 * q]) = vreg[rr];
      }
    }
#pragma unroll
    for (int r = 0; r < XQ; ++r) {
      const int idx = tid + 256 * r;
      if (idx < BN * DQ) {
        const int j = j0 + idx / DQ;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * (idx % DQ));
        *reinterpret_cast<f32x4*>(&Xs[4 * idx]) = v;
      }
    }
    if constexpr (EX) {
      if (tid < BN / 4) {
        const int j = j0 + 4 * tid;
        const float* src = a.Vt + (int64_t)TC * a.ldv + j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j + 4 <= jend) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < jend) v[e] = src[e];
        }
        *reinterpret_cast<f32x4*>(&Es[4 * tid]) = v;
      }
    }
  };

  auto keval = [&](const float (&x)[DP], int jrow) -> float {
    float sq;
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&Xs[jrow * DP + 4 * q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * q + e < D) {
          float df = x[4 * q + e] - v[e];
          sq = (4 * q + e == 0) ? df * df : __builtin_fmaf(df, df, sq);
        }
      }
    }
    return cov_from_sq<KIND>(sq, a.kparam);
  };

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();  // previous tile fully consumed
    stage_tile(j0);
    __syncthreads();

    // K elements are generated one MFMA step ahead (software pipeline) so the VALU chain of step s+1
    // issues in the shadow of the 4 x 64-cycle MFMAs of step s
    float kvn[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) kvn[ni] = keval(xi[ni], 4 * h);

#pragma unroll 2
    for (int g = 0; g < BN / 8; ++g) {
      const int jl = 8 * g + 4 * h;  // this half-wave's 4 consecutive j
      f32x4 av[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(&Vs[(ct * 32 + l31) * LDT + jl]);
      f32x4 ev;
      if constexpr (EX) ev = *reinterpret_cast<const f32x4*>(&Es[jl]);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float kv[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) kv[ni] = kvn[ni];
        // next step's row (wraps harmlessly to the tile start after the last step)
        const int jn = ((st == 3) ? (8 * (g + 1) + 4 * h) : (jl + st + 1)) & (BN - 1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) kvn[ni] = keval(xi[ni], jn);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct][st], kv[ni], acc[ni][ct], 0, 0, 0);
          if constexpr (EX) eacc[ni] = __builtin_fmaf(kv[ni], ev[st], eacc[ni]);
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  // epilogue: D[c][i] -> P[s][c][i]
  mfma_result_fence();   // the accumulators of the last contraction MFMAs are read next (common.hpp; once per workgroup)
  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 32 + l31;
    if (i < a.n) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[ni][ct][r];
        }
    }
    if constexpr (EX) {
      float tot = eacc[ni] + __shfl_xor(eacc[ni], 32, 64);
      if (h == 0 && i < a.n) Pout[(int64_t)TC * a.ldo + i] = tot;
    }
  }
}

}  // namespace gpamd
