// Fused float64 K*V: covariance generated in registers (float64 exp / sqrt on the VALU), contraction on the float64
// matrix pipe (v_mfma_f64_16x16x4_f64), K never formed -- the float64 twin of kv_mfma.hpp for models whose inputs are
// double (the reference honours the input dtype; its gradcheck-style tests run in float64).
//
// Same probe-major layout and split-j partial-slab convention as the float32 kernels.  MFMA operands (guide,
// "Fragment layout"): A (16 x 4) lane l -> A[m = l&15][k = l>>4]; B (4 x 16) lane l -> B[k = l>>4][n = l&15];
// D (16 x 16, 4 doubles per lane) -> D[row = (l>>4) + 4 r][col = l&15].  With out^T[c][i] = sum_j Vt[c][j] K[j][i]:
//   A = Vt tile (read from LDS),  B = K tile: lane (k, n) GENERATES K[j0 + k][i0 + n] -- one covariance evaluation per
//   lane per group of CT MFMAs (the same B operand serves the CT column tiles), so generation (~150 VALU cycles per
//   wave: float64 exp is ~35 instructions) hides under 4-5 MFMAs of 64 cycles when t >= 49, and bounds the product
//   for few columns.
#pragma once
#include "common.hpp"

namespace gpamd {

struct KvArgs64 {
  const double* X1;  // [n][DP]
  const double* X2;  // [m][DP]
  const double* Vt;  // [t][ldv]
  double* P;         // [S][t][ldo]
  int64_t ldv, ldo, pstride;
  int n, m, t;
  int S, jchunk, nrb;
  const int* done;
  double kparam;     // shape parameter of the family (RQ: alpha)
};

typedef double f64x4v __attribute__((ext_vector_type(4)));

constexpr int KV64_BN = 64;            // j tile
constexpr int KV64_LDV = KV64_BN + 1;  // padded LDS row (doubles): stride 130 words -> the 16 lanes of an A-operand read hit 32 distinct banks (BN + 2: 2-way conflicts, 38 % of the LDS cycles, profiles/r04_s10_kv_f64_pmc.json)
// row tiles per wave: accumulators NI * CT * 8 VGPRs + the wave's own points NI * DP * 2 VGPRs (d > 8: two tiles, or the points alone take 128+)
constexpr int kv64_ni_for(int ct, int dp) { return (ct >= 4 || dp > 8) ? 2 : 4; }

template <int KIND, int DP, int CT>
__global__ __launch_bounds__(256) void kv_f64_kernel(KvArgs64 a) {
  constexpr int NI = kv64_ni_for(CT, DP);
  constexpr int BN = KV64_BN, LDV = KV64_LDV, TC = 16 * CT;
  __shared__ __attribute__((aligned(16))) double Vs[TC * LDV];
  __shared__ __attribute__((aligned(16))) double Xs[BN * DP];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 16) + wave * (NI * 16);

  double zi[NI][DP];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 16 + l15, a.n - 1);
#pragma unroll
    for (int k = 0; k < DP; ++k) zi[ni][k] = a.X1[(int64_t)i * DP + k];
  }
  f64x4v acc[NI][CT];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ni][ct] = (f64x4v){0.0, 0.0, 0.0, 0.0};

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    for (int idx = tid; idx < TC * BN; idx += 256) {
      const int c = idx / BN, jj = idx - c * BN;
      const int j = j0 + jj;
      Vs[c * LDV + jj] = (c < a.t && j < jend) ? a.Vt[(int64_t)c * a.ldv + j] : 0.0;
    }
    for (int idx = tid; idx < BN * DP; idx += 256) {
      const int jj = idx / DP;
      // rows beyond jend: any finite point (their V entries are zero)
      Xs[idx] = a.X2[(int64_t)min(j0 + jj, a.m - 1) * DP + (idx - jj * DP)];
    }
    __syncthreads();
#pragma unroll 2
    for (int jb = 0; jb < BN; jb += 4) {
      double zj[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) zj[k] = Xs[(jb + kq) * DP + k];
      double av[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) av[ct] = Vs[(ct * 16 + l15) * LDV + jb + kq];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        double sq = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double df = zi[ni][k] - zj[k];
          sq = fma(df, df, sq);
        }
        const double kv = cov_from_sq_f64<KIND>(sq, a.kparam);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ni][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct], kv, acc[ni][ct], 0, 0, 0);
      }
    }
  }
  mfma_result_fence();   // the accumulators of the last contraction MFMAs are read next (common.hpp; once per workgroup)
  double* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 16 + l15;
    if (i < a.n) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = ct * 16 + kq + 4 * r;
          if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[ni][ct][r];
        }
    }
  }
}

// Few right-hand sides (t <= 4): the contraction on the vector ALU.  On gfx950 the float64 MFMA runs at the rate of the float64 vector FMA (64 cycles
// for the 1024 multiply-adds of a 16 x 16 x 4 tile = 16 per cycle and SIMD, the VALU's own float64 rate) and its time ADDS to the generation's VALU time
// instead of hiding it (profiles/r04_s10_kv_f64_pmc.json: t = 1 costs 210 cycles per covariance value and wave = 29 VALU instructions x 5 + one 64-cycle
// MFMA; scripts/micro/mfma_f64_valu_overlap.hip measures the two together).  A 16-column tile carrying one column therefore wastes 60 of its 64 cycles:
// with TV <= 4 columns each lane multiplies its covariance value into TV accumulators (TV v_fma_f64 of 4 cycles) and the four k-groups of a row are
// summed once at the end with two lane exchanges.  Same tile geometry, staging and partial-slab convention as kv_f64_kernel<CT = 1>.
template <int KIND, int DP, int TV>
__global__ __launch_bounds__(256) void kv_f64v_kernel(KvArgs64 a) {
  constexpr int NI = kv64_ni_for(1, DP);
  constexpr int BN = KV64_BN;
  __shared__ __attribute__((aligned(16))) double Vs[TV * BN];
  __shared__ __attribute__((aligned(16))) double Xs[BN * DP];
  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 16) + wave * (NI * 16);

  double zi[NI][DP];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 16 + l15, a.n - 1);
#pragma unroll
    for (int k = 0; k < DP; ++k) zi[ni][k] = a.X1[(int64_t)i * DP + k];
  }
  double acc[NI][TV];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int c = 0; c < TV; ++c) acc[ni][c] = 0.0;

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    for (int idx = tid; idx < TV * BN; idx += 256) {
      const int c = idx / BN, jj = idx - c * BN;
      const int j = j0 + jj;
      Vs[idx] = (c < a.t && j < jend) ? a.Vt[(int64_t)c * a.ldv + j] : 0.0;
    }
    for (int idx = tid; idx < BN * DP; idx += 256) {
      const int jj = idx / DP;
      Xs[idx] = a.X2[(int64_t)min(j0 + jj, a.m - 1) * DP + (idx - jj * DP)];   // rows beyond jend: any finite point (their V entries are zero)
    }
    __syncthreads();
#pragma unroll 2
    for (int jb = 0; jb < BN; jb += 4) {
      double zj[DP], v[TV];
#pragma unroll
      for (int k = 0; k < DP; ++k) zj[k] = Xs[(jb + kq) * DP + k];
#pragma unroll
      for (int c = 0; c < TV; ++c) v[c] = Vs[c * BN + jb + kq];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        double sq = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double df = zi[ni][k] - zj[k];
          sq = fma(df, df, sq);
        }
        const double kv = cov_from_sq_f64<KIND>(sq, a.kparam);
#pragma unroll
        for (int c = 0; c < TV; ++c) acc[ni][c] = fma(v[c], kv, acc[ni][c]);
      }
    }
  }
  double* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 16 + l15;
#pragma unroll
    for (int c = 0; c < TV; ++c) {
      double t0 = acc[ni][c];
      t0 += __shfl_xor(t0, 16, 64);
      t0 += __shfl_xor(t0, 32, 64);
      if (kq == 0 && i < a.n && c < a.t) Pout[(int64_t)c * a.ldo + i] = t0;
    }
  }
}

}  // namespace gpamd
