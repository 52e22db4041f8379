// Fused covariance MVM for 9 <= t <= 17 right-hand sides (the 17th -- 16 probes + y -- rides on the VALU, EX = 1): Gram-form generation (gram_f16.hpp) + contraction on a
// 16-COLUMN matrix-pipe tile (v_mfma_f32_16x16x1_4B_f32).
//
// The reference's DEFAULT marginal-log-likelihood evaluation is an 11-column solve (num_trace_samples = 10 probes + y).
// Measured issue costs on gfx950 (scripts/micro/mfma_rates.hip, profiles/r02_s2_mfma_rates.txt): all f32-input MFMAs run
// near 32 MAC/cycle/SIMD EXCEPT the smallest one -- 32x32x2: 64.7 cycles / 2048 MAC, 16x16x1_4B: 32.3 / 1024,
// 16x16x4: 33.2 / 1024, 4x4x1_16B: 10.2 / 256 (25 MAC/cycle, and VALU work does not issue under it).  So the tile that wastes
// least for 9..16 columns is 16 wide, and of the two 16-wide forms the FOUR-BLOCK one fits the result layout of the 32x32x16
// Gram MFMA without any cross-lane movement:
//
//   after the Gram MFMA lane l = (h = l>>5, i = l&31) holds S[j(r,h)][i], j(r,h) = (r&3) + 8(r>>2) + 4h, r = 0..15;
//   16x16x1_4B block b = l>>4 is therefore "points 16(b&1)..+15 of this 32-row tile, j half h = b>>1":
//     B_b[n]    = K[j(r,h)][16(b&1) + n]   = THIS lane's own k = f(S[r])
//     A_b[m]    = V[m][j(r,h)]              lane l supplies column m = l&15 (LDS, one ds_read_b128 per 4 r)
//     D_b[m][n] -> lane l, registers 4b'..4b'+3 of block b' (every lane holds all four blocks): rows m = 4(l>>4) + reg
//   one instruction per r: 16 columns x 32 points x 2 j.  The two j halves of a point are blocks b and b+2 of the SAME lane:
//   the epilogue adds registers, no shuffle.
//
// Cost per 32x32 block of pairs and wave: KH Gram MFMAs (37 cycles), 16 v_exp_f32 per lane, 16 contraction MFMAs of 32
// cycles, under which the generation VALU work of the other resident waves issues (unlike under 4x4x1).
// Same accuracy policy as kv_gram.hpp; compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (kvm_<family>.hip).
#pragma once
#include "gram_f16.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

constexpr int KG16_BN = 256;             // j tile staged in LDS
constexpr int KG16_LDT = KG16_BN + 4;    // padded LDS row of the V tile
constexpr int KG16_NI = 4;               // 32-row tiles per wave
constexpr int KG16_BM = 4 * KG16_NI * 32;

template <int KIND, int D, int EX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((GramF16<D>::KH == 1 && !EX) ? 3 : 2, (GramF16<D>::KH == 1 && !EX) ? 3 : 2))) void kv_gram16_kernel(KvArgs a) {
  // (three waves per SIMD only where that fits WITHOUT spilling: the extra-column variants needed ~50 scratch registers under the 168-register
  // cap, and the Matern instantiation of exactly that variant returned wrong rows on a full chip -- tests/test_gpu_kv.py regression sweep)
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = KG16_BN, LDT = KG16_LDT, NI = KG16_NI, T = 16;
  __shared__ __attribute__((aligned(16))) float Vs[T * LDT];           // [c][j]
  __shared__ __attribute__((aligned(16))) _Float16 Xh[KH * BN * 16];   // [kh][j][16]
  __shared__ __attribute__((aligned(16))) float Es[EX ? BN : 4];       // 17th column (EX): rides on the VALU

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * KG16_BM + wave * (NI * 32);
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
  f32x16 acc[NI];
  float eacc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    eacc[ni] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  }

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    // V tile: 16 rows x BN/4 float4 = 4 float4 per thread, coalesced along j; rows >= t and j >= jend are zero
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
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
      if constexpr (EX) Es[tid] = j < jend ? a.Vt[(int64_t)T * a.ldv + j] : 0.f;
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
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {   // pair form (common.hpp cov_pair_from_sq): packed-f32 arithmetic around the transcendentals
          const f32x2 k2 = cov_pair_from_sq<KIND>((f32x2){kk[ni][r], kk[ni][r + 1]}, a.kparam);
          kk[ni][r] = k2[0];
          kk[ni][r + 1] = k2[1];
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&Vs[l15 * LDT + jb + 8 * q + 4 * h]);
        f32x4 ev;
        if constexpr (EX) ev = *reinterpret_cast<const f32x4*>(&Es[jb + 8 * q + 4 * h]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x1f32(av[e], kk[ni][4 * q + e], acc[ni], 0, 0, 0);
            if constexpr (EX) eacc[ni] = __builtin_fmaf(kk[ni][4 * q + e], ev[e], eacc[ni]);
          }
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  // D_b[m][n]: registers 4b..4b+3 hold rows m = 4(lane>>4) + reg of block b; block b = (j half b>>1, point half b&1)
  mfma_result_fence(acc);   // the accumulators of the last contraction MFMAs are read next
  float* Pout = a.P + (int64_t)s * a.pstride;
  const int mrow = 4 * (lane >> 4);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      const int i = ibase + ni * 32 + 16 * ih + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = mrow + r;
        const float tot = acc[ni][4 * ih + r] + acc[ni][4 * (2 + ih) + r];
        if (i < a.n && c < a.t) Pout[(int64_t)c * a.ldo + i] = tot;
      }
    }
  if constexpr (EX) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int i = ibase + ni * 32 + l31;
      const float tot = eacc[ni] + __shfl_xor(eacc[ni], 32, 64);
      if (h == 0 && i < a.n) Pout[(int64_t)T * a.ldo + i] = tot;
    }
  }
}

}  // namespace gpamd
