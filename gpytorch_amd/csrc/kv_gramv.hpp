// Small right-hand-side count (t <= 8) fused covariance MVM with the squared distances generated on the matrix
// pipe (hi/lo-split f16 Gram form, gram_f16.hpp) and the t products carried on the VALU.
//
// kv_valu.hpp spends 2*D packed + T VALU instructions and one v_exp_f32 per (i, j) pair and is bound by that
// (6.4e12 pairs/s measured at D = 3, T = 1).  Here a 32 x 32 block of squared distances costs GramF16<D>::KH MFMAs of
// 32 cycles on the otherwise idle matrix pipe (2 cycles per wave of pairs at D = 3), and the VALU only does
// k = f(S)  and  acc += k * v_j.  Same accuracy policy as kv_gram.hpp (the host selects it only when max |z|^2 <= 32,
// never for Matern nu = 1/2).  Used by the predictive-mean solve and the Lanczos steps
// (gpytorch/models/exact_prediction_strategies.py:271,286) -- sequences of t = 1 products -- and by K_*X @ mean_cache.
//
// Lane (l31, h) of a wave owns output row i = l31 of each of its NI row tiles and, per 32-row j block, the 16 j rows
// (r&3) + 8(r>>2) + 4h of the MFMA result layout; the two half-waves are summed once in the epilogue.
#pragma once
#include "gram_f16.hpp"
#include "kv_mfma.hpp"
#include "kv_valu.hpp"

namespace gpamd {

constexpr int KGV_BN = 256;  // j-tile
constexpr int KGV_NI = 4;    // 32-row tiles per wave
constexpr int KGV_BM = 4 * KGV_NI * 32;

template <int KIND, int D, int T>
__global__ __launch_bounds__(256) void kv_gramv_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = KGV_BN, NI = KGV_NI;
  constexpr int TP = T >= 2 ? T / 2 : 1;  // packed column pairs
  __shared__ __attribute__((aligned(16))) float Vs[BN * T];            // [j][c]
  __shared__ __attribute__((aligned(16))) _Float16 Xh[KH * BN * 16];   // [kh][j][16]

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * KGV_BM + wave * (NI * 32);
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
  f32x2 acc2[NI][TP];    // T >= 2: column pairs on packed math
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
    for (int c = 0; c < TP; ++c) acc2[ni][c] = (f32x2)(0.f);
  }

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    {  // stage: one contracted point per thread (BN == 256)
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
#pragma unroll
      for (int c = 0; c < T; ++c) Vs[tid * T + c] = (c < a.t && j < jend) ? a.Vt[(int64_t)c * a.ldv + j] : 0.f;
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
      mfma_result_fence(kk);   // data-dependent fence (common.hpp): the Gram MFMAs before, every reader after the wait states
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = jb + 8 * g + 4 * h;  // this half-wave's 4 consecutive j rows of group g (LDS broadcast reads)
        if constexpr (T == 1) {
          // one column: two j rows per packed multiply-add (v_pk_fma_f32: the loop is bound by VALU issue -- one v_exp_f32 per pair plus
          // the multiply-add; pairing halves the latter); the two partial sums are added in the epilogue
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(&Vs[jl]);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 vv = {v4[e], v4[e + 1]};
            // (pair form: packed-f32 arithmetic around the transcendentals, |S| as a source modifier of v_sqrt_f32 -- common.hpp; the NI row tiles'
            // pairs stage by stage: cov_pairs_from_sq)
            f32x2 sq[NI], kv[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) sq[ni] = (f32x2){kk[ni][4 * g + e], kk[ni][4 * g + e + 1]};
            cov_pairs_from_sq<KIND, NI>(sq, a.kparam, kv);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc2[ni][0] = __builtin_elementwise_fma(kv[ni], vv, acc2[ni][0]);
          }
        } else if constexpr (T == 2 && KIND != KIND_RBF) {
          // TWO columns (four and more keep the scalar form: with two more operand pairs live the allocator gave up a resident wave at d = 10), families with more than one transcendental per pair (Matern: v_sqrt + v_exp + a polynomial; RQ: v_log + v_exp): two
          // j rows per call of the PAIR form, whose arithmetic around the transcendentals is packed (common.hpp cov_pair_from_sq; round 6 -- until then
          // only the one-column variant used it: 36 -> ~28 issue cycles per pair at two columns, the fused solve + Lanczos product of the prediction caches)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            f32x2 vp0[TP], vp1[TP];
#pragma unroll
            for (int c = 0; c < TP; ++c) {
              vp0[c] = *reinterpret_cast<const f32x2*>(&Vs[(jl + e) * T + 2 * c]);
              vp1[c] = *reinterpret_cast<const f32x2*>(&Vs[(jl + e + 1) * T + 2 * c]);
            }
            f32x2 sq[NI], kv[NI];   // the NI row tiles' pairs stage by stage (common.hpp cov_pairs_from_sq)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) sq[ni] = (f32x2){kk[ni][4 * g + e], kk[ni][4 * g + e + 1]};
            cov_pairs_from_sq<KIND, NI>(sq, a.kparam, kv);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
              for (int c = 0; c < TP; ++c) {
                acc2[ni][c] = __builtin_elementwise_fma((f32x2)(kv[ni][0]), vp0[c], acc2[ni][c]);
                acc2[ni][c] = __builtin_elementwise_fma((f32x2)(kv[ni][1]), vp1[c], acc2[ni][c]);
              }
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2 vp[TP];
#pragma unroll
            for (int c = 0; c < TP; ++c) vp[c] = *reinterpret_cast<const f32x2*>(&Vs[(jl + e) * T + 2 * c]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              float sv = kk[ni][4 * g + e];
              if constexpr (KIND != KIND_RBF) sv = __builtin_fabsf(sv);   // (|S|: a source modifier of v_sqrt_f32 / harmless for 1 + s; no v_med3_f32)
              const f32x2 kv = (f32x2)(cov_from_sq<KIND>(sv, a.kparam));
#pragma unroll
              for (int c = 0; c < TP; ++c) acc2[ni][c] = __builtin_elementwise_fma(kv, vp[c], acc2[ni][c]);
            }
          }
        }
      }
    }
  }

  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 32 + l31;
#pragma unroll
    for (int c = 0; c < T; ++c) {
      const float part = T == 1 ? acc2[ni][0][0] + acc2[ni][0][1] : acc2[ni][c / 2][c & 1];
      const float tot = part + __shfl_xor(part, 32, 64);
      if (h == 0 && i < a.n && c < a.t) Pout[(int64_t)c * a.ldo + i] = tot;
    }
  }
}

}  // namespace gpamd
