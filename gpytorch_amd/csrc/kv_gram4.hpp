// Fused covariance MVM for 3 <= t <= 32 right-hand sides: Gram-form generation (gram_f16.hpp) + contraction on the matrix
// pipe in COLUMN GROUPS OF FOUR (v_mfma_f32_4x4x1_16B_f32).
//
// Why this shape.  The reference's DEFAULT marginal-log-likelihood evaluation solves with num_trace_samples = 10 probes
// + y = 11 columns (gpytorch/settings.py num_trace_samples; linear_operator _probe_vectors_and_norms), the posterior
// covariance path with a handful.  The 32-column tile of kv_gram.hpp is then a third full; kv_gramv.hpp's VALU contraction
// pays 0.73 issue slots per (pair, column).  All f32-input MFMAs run at the same 64 flop/clk/SIMD, so the tile that wastes
// least is the smallest: 4x4x1_16B does 16 independent 4x4 outer products per 8-cycle instruction, and the block structure
// matches the result layout of the 32x32x16 Gram MFMA exactly:
//
//   lane l = (h = l>>5, i = l&31) holds, after the Gram MFMA, S[j(r,h)][i] for r = 0..15, j(r,h) = (r&3) + 8(r>>2) + 4h;
//   MFMA block b = l>>2 is therefore "points 4(b&7)..+3 of this 32-row tile, j half h = b>>3":
//     B_b[n]   = K[j(r,h)][4(b&7) + n]      = THIS lane's own k = f(S[r])                    (no cross-lane movement)
//     A_b[a]   = V[4g + a][j(r,h)]           for column group g: lane l supplies column 4g + (l&3)  (LDS, float4 along j)
//     D_b[a][n] -> lane l, register a        = partial (over its j half) of P[4g + a][i]
//   one instruction per (r, g): 4 columns x 32 points x 2 j.  The two j halves are added once in the epilogue (one
//   __shfl_xor 32), as in kv_gramv.hpp.
//
// Cost per 32x32 block of pairs and wave: KH Gram MFMAs (32 cycles each), 16 v_exp_f32 per lane, 16*G contraction MFMAs of
// 8 cycles (G = ceil(t/4)): t = 11 -> 384 + 32 matrix-pipe cycles per 1024 pairs against 1024 + 32 on the 32-column tile.
// The A operands depend on (r, g, l&3, h) only -- NOT on the row tile -- so one ds_read_b128 per (4 r, g) serves all NI row
// tiles of the wave (a v_mfma cannot take a broadcast operand; re-reading per row tile would saturate the LDS pipe).
//
// Same accuracy policy as kv_gram.hpp (host selects it only when max |z|^2 <= 32, never for Matern nu = 1/2); compiled with
// -mllvm -amdgpu-mfma-vgpr-form=1 (kvm_<family>.hip): the Gram results feed v_exp_f32 directly.
#pragma once
#include "gram_f16.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

constexpr int KG4_BN = 256;                      // j tile staged in LDS
constexpr int KG4_LDT = KG4_BN + 4;              // padded LDS row of the V tile (16-B aligned)
constexpr int kg4_ni(int g) { return g <= 4 ? 4 : 2; }   // 32-row tiles per wave: 16*NI + 4*NI*G accumulator + distance registers
inline int kg4_bm(int g) { return 4 * kg4_ni(g) * 32; }

template <int KIND, int D, int G>
__global__ __launch_bounds__(256) void kv_gram4_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = KG4_BN, LDT = KG4_LDT, NI = kg4_ni(G), T = 4 * G;
  __shared__ __attribute__((aligned(16))) float Vs[T * LDT];           // [c][j]
  __shared__ __attribute__((aligned(16))) _Float16 Xh[KH * BN * 16];   // [kh][j][16] split augmented x_j rows

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5, l3 = lane & 3;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 32) + wave * (NI * 32);
  float cz[DP];   // centre of this workgroup's row block (zero unless the host passed chunk centres: gram_f16.hpp)
  load_center<DP>(a.Xc, ibase - wave * (NI * 32), 4 * NI * 32, a.n, cz);

  f16x8 bq[NI][KH];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 32 + l31, a.n - 1);
    float z[DP];
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP + 4 * q);
      z[4 * q + 0] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
    }
    sub_center<DP>(z, cz);
    gram_pack_b<D>(z, h, bq[ni]);
  }
  f32x4 acc[NI][G];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[ni][g] = (f32x4)(0.f);

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    // V tile: T rows x BN/4 float4 = G float4 per thread, coalesced along j; rows >= t and j >= jend are zero
#pragma unroll
    for (int rr = 0; rr < G; ++rr) {
      const int idx = tid + 256 * rr;
      const int c = idx / (BN / 4), q = idx % (BN / 4);
      const int j = j0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < a.t) {
        const float* src = a.Vt + (int64_t)c * a.ldv + j;
        if (j + 4 <= jend) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < jend) v[e] = src[e];
        }
      }
      *reinterpret_cast<f32x4*>(&Vs[c * LDT + 4 * q]) = v;
    }
    {  // split augmented x_j rows: one contracted point per thread (BN == 256)
      const int j = j0 + tid;
      float z[DP];
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * q);
        z[4 * q + 0] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
      }
      if (j < jend) sub_center<DP>(z, cz);
      gram_pack_a<D>(z, j < jend, Xh, tid, BN);
    }
    __syncthreads();

#pragma unroll 1
    for (int jb = 0; jb < BN; jb += 32) {
      f32x16 kk[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) kk[ni][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const f16x8 aq = *reinterpret_cast<const f16x8*>(&Xh[gram_a_off(kh, jb + l31, h, BN)]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) kk[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, bq[ni][kh], kk[ni], 0, 0, 0);
      }
      mfma_result_fence(kk);   // VGPR-destination MFMA results are read by the VALU next (gram_f16.hpp)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jl = jb + 8 * q + 4 * h;   // this half-wave's 4 consecutive j rows of register group q
        f32x4 av[G];
#pragma unroll
        for (int g = 0; g < G; ++g) av[g] = *reinterpret_cast<const f32x4*>(&Vs[(4 * g + l3) * LDT + jl]);
#pragma unroll
        for (int e = 0; e < 4; e += 2)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            // pair form (common.hpp cov_pair_from_sq): packed-f32 arithmetic around the transcendentals
            const f32x2 kv = cov_pair_from_sq<KIND>((f32x2){kk[ni][4 * q + e], kk[ni][4 * q + e + 1]}, a.kparam);
#pragma unroll
            for (int g = 0; g < G; ++g) {
              acc[ni][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[g][e], kv[0], acc[ni][g], 0, 0, 0);
              acc[ni][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[g][e + 1], kv[1], acc[ni][g], 0, 0, 0);
            }
          }
      }
    }
  }

  mfma_result_fence(acc);   // the accumulators of the last contraction MFMAs are read next
  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 32 + l31;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * g + r;
        const float part = acc[ni][g][r];
        const float tot = part + __shfl_xor(part, 32, 64);
        if (h == 0 && i < a.n && c < a.t) Pout[(int64_t)c * a.ldo + i] = tot;
      }
  }
}

}  // namespace gpamd
