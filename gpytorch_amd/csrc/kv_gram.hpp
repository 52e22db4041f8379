// Fused K*V with the squared-distance tile ALSO generated on the matrix pipe ("Gram form").
//
// Same contract, layout and work decomposition as kv_mfma.hpp.  Difference: instead of every lane
// evaluating |z_i - z_j|^2 with 2*D VALU ops per pair, a 32x32 block of squared distances is produced
// by KA = ceil((D+2)/2) extra MFMAs from augmented coordinates -- exactly the quadratic expansion of the
// reference's sq_dist (gpytorch/kernels/kernel.py:26-49:  [-2x, |x|^2, 1] . [x', 1, |x'|^2]^T ):
//     S[j][i] = sum_k Aaug[j][k] * Baug[k][i],   Aaug[j] = [z_j, |z_j|^2, 1, 0..],  Baug[:,i] = [-2 z_i, 1, |z_i|^2, 0..]
// The MFMA D layout hands lane (h, i) the 16 values S[(r&3)+8(r>>2)+4h][i], r = 0..15 -- which is precisely
// the set of K elements that lane must feed as the B operand of the 16 contraction steps of that 32-row
// j block (step r pairs rows (r&3)+8(r>>2) and +4), so NO cross-lane movement is needed: the VALU only
// applies k = f(max(S,0)) (one v_exp_f32 for RBF) in place.  VALU work per K element drops from
// 2D+1 (+1) to 2 (+1) instructions; cost: KA extra MFMAs per 16*CT useful ones (9 % at D=3, CT=2).
//
// Numerics: S carries the cancellation error of the quadratic expansion, <= ~8 eps * (|z_i|+|z_j|)^2 in
// fp32 (the reference's fp32 path has the same error; it mean-centres for this reason).  The host only
// selects this kernel when max |z|^2 (after centring) <= 32, i.e. a relative error <= 2e-5 in K (worst case, tests/test_gram_split_cpu.py; typically 5e-6), and never
// for Matern nu = 1/2 (k = exp(-sqrt(s)) is not Lipschitz in s at 0).  Otherwise kv_mfma.hpp is used.
//
// Tried and measured without gain on MI355X (profiles/r01_s13_kv_tune_gram_seq_prefetch.jsonl): row tiles one after
// the other with occupancy forced to three waves/SIMD (+1 %, needs spills), register prefetch of the next tile
// across the MFMA phase (0 %), both together (-4 %).  The structure sits at ~122 TFLOP/s; the same loop without
// any generation reaches 140 (profiles/r01_s4_kv_tune_variants.jsonl).
// Since session s17 the Gram block itself runs at the f16 MFMA rate on hi/lo-split coordinates (gram_f16.hpp):
// GramF16<D>::KH instructions of 32 cycles instead of KA of 64.
#pragma once
#include "gram_f16.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

// Register budget: 64 accumulators (CT <= 2) fit three waves per SIMD (<= 168 unified registers) only if the
// allocator is told to: left alone it parks the 16 distance registers in AGPRs (v_accvgpr_read before every v_exp)
// and lands at 180.
template <int KIND, int D, int CT, int NI, int EX, int SAFE = 0>   // SAFE = 1: hazard stress builds only (tune/tune_hazard.hip): the full mfma_result_fence after the Gram MFMAs
// (CT = 1 runs four row tiles per wave: 64 accumulators + 64 distance registers do not fit 168 registers -- 6..44 spilled, and with the extra
// column the Matern-3/2 / RQ instantiations returned wrong rows on a full chip, tests/test_gpu_kv.py regression sweep -> two waves there)
// (beyond 16 dimensions the split x_i operands -- NI * KH * 4 registers, KH = 4 .. 7 -- take the room of the third wave: two waves per SIMD)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((CT == 2 || (CT == 1 && NI < 4)) && D <= 16) ? 3 : 2, ((CT == 2 || (CT == 1 && NI < 4)) && D <= 16) ? 3 : 2)))
void kv_gram_kernel(KvArgs a) {
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;    // 32x32x16 f16 MFMAs per 32x32 block of squared distances
  static_assert(NI >= 2, "the MFMA -> VALU ordering below relies on at least two row tiles per wave");
  constexpr int BN = KV_BN, LDT = KV_LDT, TC = 32 * CT;
  __shared__ __attribute__((aligned(16))) float smem[TC * LDT + BN + KH * BN * 8];
  float* Vs = smem;
  float* Es = smem + TC * LDT;          // 16-B aligned (TC*LDT*4 is a multiple of 16)
  _Float16* Xh = reinterpret_cast<_Float16*>(Es + BN);   // [KH][BN][16] split augmented x_j rows

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 32) + wave * (NI * 32);
  float cz[DP];   // centre of this workgroup's row block (zero unless the host passed chunk centres: gram_f16.hpp)
  load_center<DP>(a.Xc, ibase - wave * (NI * 32), 4 * NI * 32, a.n, cz);

  // B operands of the Gram MFMAs (this lane's output row, k-group h), kept in registers for the whole kernel
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

  constexpr int VQ = TC * (BN / 4) / 256;
  constexpr int VCH = VQ < 8 ? VQ : 8;

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
    // split augmented x_j rows (rows beyond jend: all zero -> S = 0, k = 1; their V entries are zero so they
    // contribute nothing)
    if (tid < BN) {
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

  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    stage_tile(j0);
    __syncthreads();

#pragma unroll 1
    for (int jb = 0; jb < BN; jb += 32) {
      // ---- squared distances of the 32 x (NI*32) block on the matrix pipe, then k = f(S) in place ----
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
      // With the register cap above the distance blocks live in VGPRs and the VALU reads them next: an MFMA -> VALU read-after-write is not
      // interlocked (common.hpp, mfma_result_fence).  Here no idle fence is needed -- the ORDER is pinned instead: all NI (>= 2) Gram MFMAs
      // first, then the tiles are converted one after the other.  A second MFMA cannot issue while the first occupies the matrix pipe, so
      // tile 0 is complete when the last MFMA has issued, and tile ni is read only after the >= 16 transcendental instructions of every
      // earlier tile (>= 128 cycles against the 32 + pipeline cycles of an 8-pass MFMA).
      // Round 5: the distance from a Gram MFMA to the first read of ITS registers no longer rests on the toolchain's table (12 wait states: the
      // very distance measured insufficient in kv_gramv): 8 explicit wait states more behind the whole group (20 in all), tied to the result registers
      // so that no MFMA can sink below them -- 8 more idle cycles per 32-row block against the ~2500 of its contraction (the static audit's bar for
      // this kernel is now 20, tests/test_isa_hazard_cpu.py; the on-device stress test compares bitwise with the fully fenced build)
      __builtin_amdgcn_sched_barrier(0);
      // (SAFE: 0 = the product; 1 = the full fence; 2 = round 4's form, no explicit wait states; 3 = the product's wait states without the second scheduling
      // barrier -- 1 .. 3 exist in the tune library only: the stress test's reference and the A/B of what the 8 wait states cost, scripts/kv_gram_fence_ab.py)
      if constexpr (SAFE == 1) {
        mfma_result_fence(kk);
      } else if constexpr (SAFE == 0 || SAFE == 3) {
        mfma_tie(kk);
        asm volatile("s_nop 7");   // (+ the toolchain's own 12 behind it: the hazard recogniser does not count wait states inside inline asm)
        mfma_tie(kk);
        if constexpr (SAFE == 0) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          // RBF: a slightly negative S (cancellation) only makes k = 2^-S exceed 1 by <= 1e-5 -- no clamp needed; Matern takes sqrt(|S|).
          // Two elements per call: the arithmetic around the transcendentals runs on the packed-f32 pipe (common.hpp cov_pair_from_sq)
          const f32x2 k2 = cov_pair_from_sq<KIND>((f32x2){kk[ni][r], kk[ni][r + 1]}, a.kparam);
          kk[ni][r] = k2[0];
          kk[ni][r + 1] = k2[1];
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- contraction: 16 steps (4 groups of 4), step r pairs rows (r&3)+8(r>>2) and +4 ----
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = jb + 8 * g + 4 * h;
        f32x4 av[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(&Vs[(ct * 32 + l31) * LDT + jl]);
        f32x4 ev;
        if constexpr (EX) ev = *reinterpret_cast<const f32x4*>(&Es[jl]);
#ifndef GPAMD_NO_SETPRIO   // A/B: 240.7 ms with, 245.0-246.0 ms without (n = 500 000, 65 columns; profiles/r02_s23_setprio_ab_fp32_kernel.txt)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct][st], kk[ni][4 * g + st], acc[ni][ct], 0, 0, 0);
            if constexpr (EX) eacc[ni] = __builtin_fmaf(kk[ni][4 * g + st], ev[st], eacc[ni]);
          }
        }
#ifndef GPAMD_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
  }

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
