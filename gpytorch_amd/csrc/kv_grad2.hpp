// Fused bilinear derivative, Gram-form generation, workgroup-shared right tile, input gradients.
//
//   G_theta = sum_ij W_ij dK_ij/dtheta,   Gx1[i] = sum_j W_ij dK_ij/dx1_i,   W = L^T R (never formed in HBM)
//
// Same contract as kv_grad.hpp (LinearOperator._bilinear_derivative on the kernel operator + the dense kernel backward
// of gpytorch/functions/rbf_covariance.py:26-29, matern_covariance.py:53-56; input gradients as the KeOps precedent
// provides them, gpytorch/test/base_keops_test_case.py:105-132).  What changed against kv_grad.hpp (measured 0.37 of the
// fp32 MFMA peak: every wave fetched its own copy of the R tile from L2, one dword per MFMA, latency-bound):
//   * the R tile of a 64-row j step is staged ONCE per workgroup in LDS, transposed to Rs[j][k-half][c/2] so that a lane
//     fetches the A operands of FOUR MFMA steps with one ds_read_b128 (the L tile, per wave, likewise: B operands);
//   * squared distances come from the split-f16 Gram MFMA (gram_f16.hpp) instead of 3 D VALU ops per pair -- its result
//     layout S[j(r,h)][i] is the layout of the W^T tile, so k, dk/ds and W meet in the same register slot;
//   * per-dimension sums never touch (z_i - z_j) per pair: with A = W o dk/ds,
//         sum_ij A_ij (z_iq - z_jq)^2 = sum_i [ z_iq^2 rs_i - 2 z_iq u_iq + v_iq ],
//         rs_i = sum_j A_ij,  u_iq = sum_j A_ij z_jq,  v_iq = sum_j A_ij z_jq^2
//     and [rs | u | v] = A^T-contraction against the columns [1 | z_j | z_j^2] -- the K*V contraction of kv_gram4.hpp with A
//     in the place of K (v_mfma_f32_4x4x1, column groups of four).  The same rs / u give the input gradient for free:
//         dS_ij/dz_iq = 2 (z_iq - z_jq)   ->   Gz1[i][q] = 2 (z_iq rs_i - u_iq)
//     (the host applies dz/dx = coef / l_q and theta).  The combination is carried in float64 per j tile.
//   * single-lengthscale hyper-gradients without input gradients (MODE 0) need only sum A_ij S_ij: one fma per pair.
// Host policy as for kv_gram.hpp: max |z|^2 <= 32, never Matern nu = 1/2 (kv_grad.hpp remains the fallback).
//
// WSPLIT = 1 (the library default, flag GPAMD_KV_SPLIT): the W^T = R L^T tiles run on the f16 matrix pipe at f32 accuracy, exactly like the
// contraction of kv_gramh.hpp -- both vector blocks are split into hi/lo f16 planes by a pre-pass (kv_wsplit.hpp: row-major [row][80],
// per-column power-of-two scales with a constant product) and a 32 x 32 tile costs 5 k-steps x 3 products = 15 v_mfma_f32_32x32x16_f16
// (480 cycles) in place of 33 v_mfma_f32_32x32x2_f32 (2112 cycles) at 65 columns.  The L operands of a wave's 32 rows stay in registers
// for the whole kernel (40 VGPRs); only the R planes of a 64-row j step go through LDS.
#pragma once
#include "gram_f16.hpp"
#include "kv_wsplit.hpp"

namespace gpamd {

struct Grad2Args {
  const float* X1;  // [n][DP] prepared
  const float* X2;  // [m][DP]
  const float* Xc = nullptr;  // optional [ceil(n / 128)][DP] chunk centres of X1 (block-centred Gram expansion, gram_f16.hpp) or nullptr
  const float* Lt;  // [t][ldl] left vectors (index i, with X1)
  const float* Rt;  // [t][ldr] right vectors (index j, with X2)
  int64_t ldl, ldr;
  int n, m, t;
  int S, jchunk, nrb;  // j split, chunk length (multiple of 64), row blocks of 128
  int th4;             // MFMA k-steps (2 columns each) rounded up to a multiple of 4
  int rs;              // LDS row stride (floats) of the L / R tiles: >= 2*th4, == 4 mod 32
  double* part;        // [nrb*S][2 + DP]   (hyper-parameter partial sums; last entry: shape-parameter sum, RQ)
  float kparam;        // covariance shape parameter (RQ: alpha)
  float* Px;           // optional [S][DP][ldx] partial slabs of Gz1 (probe-major: one coordinate per row) or nullptr
  int64_t ldx, pxstride;
  // WSPLIT: hi / lo planes of the scaled vector blocks, row-major [rows_pad][WS_CP] (zero rows beyond n / m), and the device scalar that
  // restores W (kv_wsplit.hpp)
  const _Float16* Lh = nullptr;
  const _Float16* Ll = nullptr;
  const _Float16* Rh = nullptr;
  const _Float16* Rl = nullptr;
  const float* wscale = nullptr;
  const int* tiles = nullptr;   // far-pair culling (kv_cull.hpp): per unit, the starts of the 64-row j steps it visits (nullptr: all of them)
  int tpc1 = 0;
};

constexpr int G2_CPL = WS_CP + 8;   // LDS row stride (f16) of the R planes: 176 B -> conflict-free ds_read_b128

constexpr int G2_BN = 64;    // j rows per step (two 32-row MFMA tiles)
constexpr int G2_MAXT = 66;  // columns per launch (host: kvm_grad2.hip)

// k(s), dk/ds and -- for families with a shape parameter -- dk/dp at fixed s (RQ: k = (1+s)^-p -> dk/dp = -k ln(1+s))
template <int KIND>
__device__ __forceinline__ void cov_and_dcov(float s, float p, float& k, float& dk, float& dp) {
  dp = 0.f;
  if constexpr (KIND == KIND_RBF) {
    k = __builtin_amdgcn_exp2f(-s);
    dk = -0.6931471805599453f * k;
  } else if constexpr (KIND == KIND_RQ) {
    const float l2 = __builtin_amdgcn_logf(1.0f + s);           // log2(1 + s)
    k = __builtin_amdgcn_exp2f(-p * l2);
    dk = -p * k / (1.0f + s);
    dp = -0.6931471805599453f * l2 * k;
  } else {
    const float r = __builtin_amdgcn_sqrtf(s);
    const float e = __builtin_amdgcn_exp2f(-r * LOG2E);
    if constexpr (KIND == KIND_MATERN32) {
      k = (1.0f + r) * e;
      dk = -0.5f * e;
    } else {  // Matern 5/2
      k = __builtin_fmaf(s, 1.0f / 3.0f, 1.0f + r) * e;
      dk = -(1.0f + r) * e * (1.0f / 6.0f);
    }
  }
}

// MODE 0: one lengthscale, no input gradients (VALU: sum A S).   MODE 1: per-dimension sums + optional input gradients.
// (per-dimension mode of the split form beyond 6 dimensions: its registers -- 40 for the L planes, 16-36 for the [1 | z | z^2] accumulators
// and operands -- do not fit two waves per SIMD; G2_WAVES = 1 lets the allocator use the whole file instead of spilling 44-182 VGPRs)
template <int D, int MODE, int WSPLIT>
constexpr int g2_waves() { return (WSPLIT && MODE == 1 && D > 6) ? 1 : 2; }

template <int KIND, int D, int MODE, int WSPLIT = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(g2_waves<D, MODE, WSPLIT>(), 2))) void kv_grad2_kernel(Grad2Args a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = G2_BN;
  constexpr int NZ = 1 + 2 * D, GZ = (NZ + 3) / 4, LDZ = BN + 4;
  constexpr int NK = WS_CP / 16, CPL = G2_CPL;                      // WSPLIT: k-steps of 16 columns, LDS row stride of the R planes
  const int RS = WSPLIT ? 0 : a.rs, TH4 = a.th4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5, l3 = lane & 3;
  float* Ls = dyn + (size_t)wave * 32 * RS;                       // [32 i][RS]: this wave's L tile, [h][c2] within a row
  float* Rs = dyn + (size_t)4 * 32 * RS;                          // [BN j][RS]
  _Float16* Rsh = reinterpret_cast<_Float16*>(dyn);               // WSPLIT: [2 planes][BN j][CPL]
  _Float16* Xh = WSPLIT ? Rsh + 2 * BN * CPL : reinterpret_cast<_Float16*>(Rs + (size_t)BN * RS);  // [KH][BN][16]
  float* Zs = reinterpret_cast<float*>(Xh + KH * BN * 16);        // [4*GZ][LDZ]  (MODE 1)
  __shared__ double red[4][2 + DP];

  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int i0 = rb * 128 + wave * 32;
  const int i = i0 + l31;

  // zero the padded tiles once (columns >= t and the k-step padding stay zero for the whole kernel)
  if constexpr (!WSPLIT)
    for (int e = tid; e < (4 * 32 + BN) * RS; e += 256) dyn[e] = 0.f;
  if constexpr (MODE == 1)
    for (int e = tid; e < 4 * GZ * LDZ; e += 256) Zs[e] = 0.f;
  __syncthreads();
  f16x8 lbh[WSPLIT ? NK : 1], lbl[WSPLIT ? NK : 1];   // WSPLIT: B operands (this lane's row i, k-group h) of all k-steps, kept for the whole kernel
  if constexpr (WSPLIT) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {   // (the planes are zero-padded to whole 128-row blocks: no bounds check)
      lbh[ks] = *reinterpret_cast<const f16x8*>(a.Lh + (int64_t)i * WS_CP + 16 * ks + 8 * h);
      lbl[ks] = *reinterpret_cast<const f16x8*>(a.Ll + (int64_t)i * WS_CP + 16 * ks + 8 * h);
    }
  } else {
    // this wave's L tile: Ls[ic][h_c][c2] = L[c][i0 + ic], c = 2 c2 + h_c   (128-B coalesced rows of Lt)
    for (int c = h; c < a.t; c += 2) {
      float v = 0.f;
      if (i < a.n) v = a.Lt[(int64_t)c * a.ldl + i];
      Ls[l31 * RS + (c & 1) * TH4 + (c >> 1)] = v;
    }
  }
  float zi[DP];
  {
    const int ic = min(i, a.n - 1);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)ic * DP + 4 * q);
      zi[4 * q + 0] = v[0]; zi[4 * q + 1] = v[1]; zi[4 * q + 2] = v[2]; zi[4 * q + 3] = v[3];
    }
  }
  float cz[DP];   // centre of this workgroup's 128 rows (zero unless the host passed chunk centres): every use of z_i / z_j below is
  load_center<DP>(a.Xc, rb * 128, 128, a.n, cz);   // translation invariant, so the centred coordinates serve the sums as well
  sub_center<DP>(zi, cz);
  f16x8 bq[KH];
  gram_pack_b<D>(zi, h, bq);
  double g[2 + DP];
#pragma unroll
  for (int q = 0; q <= DP + 1; ++q) g[q] = 0.0;
  float gx[DP];
#pragma unroll
  for (int q = 0; q < DP; ++q) gx[q] = 0.f;

  // Staging is software-pipelined through registers: the global loads of step k+1 (R tile: <= 5 float4 per thread, x_j:
  // one point per thread of the first wave) are issued BEFORE the MFMA phase of step k and land in LDS after it -- with two
  // workgroups per CU nothing else would hide the ~2 us L2 latency of a synchronous stage (first build: 391 ms vs 242 ms for
  // the K*V of equal flops, profiles/r02_s3_grad_timing_first.json).
  constexpr int NR = WSPLIT ? (2 * BN * (WS_CP / 8)) / 256 : (G2_MAXT * (BN / 4) + 255) / 256;   // WSPLIT: 16-byte chunks of both planes
  f32x4 rreg[NR];
  float zreg[DP];
  auto fetch = [&](int j0) {
    if constexpr (WSPLIT) {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {   // chunk q: plane q / 640, row (q % 640) / 10, 16-byte column chunk q % 10 (planes zero-padded to whole tiles)
        const int q = tid + 256 * rr;
        const int pl = q / (BN * (WS_CP / 8)), rem = q % (BN * (WS_CP / 8));
        const int row = rem / (WS_CP / 8), ck = rem % (WS_CP / 8);
        const _Float16* src = (pl ? a.Rl : a.Rh) + (int64_t)(j0 + row) * WS_CP + 8 * ck;
        rreg[rr] = *reinterpret_cast<const f32x4*>(src);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const int idx = tid + 256 * rr;
        const int c = idx / (BN / 4), q = idx % (BN / 4);
        const int j = j0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < a.t) {
          const float* src = a.Rt + (int64_t)c * a.ldr + j;
          if (j + 4 <= jend && ((a.ldr & 3) == 0)) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < jend) v[e] = src[e];
          }
        }
        rreg[rr] = v;
      }
    }
    if (tid < BN) {
      const int j = j0 + tid;
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * q);
        zreg[4 * q + 0] = v[0]; zreg[4 * q + 1] = v[1]; zreg[4 * q + 2] = v[2]; zreg[4 * q + 3] = v[3];
      }
    }
  };
  const int* tl = a.tiles ? a.tiles + (int64_t)unit * a.tpc1 : nullptr;   // far-pair culling: the look-ahead fetch takes the next SURVIVING step
  const int jfirst = tile_start<BN>(tl, jbeg, 0);
  if (jfirst < jend) fetch(jfirst);

  for (int j0 = jfirst, jn, tk = 1; j0 < jend; j0 = jn, ++tk) {
    jn = tile_start<BN>(tl, jbeg, tk);
    __syncthreads();
    // ---- registers -> LDS: the R tile transposed to Rs[j][k-half][c/2], the split augmented x_j rows, the [1 | z | z^2] columns
    if constexpr (WSPLIT) {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const int q = tid + 256 * rr;
        const int pl = q / (BN * (WS_CP / 8)), rem = q % (BN * (WS_CP / 8));
        const int row = rem / (WS_CP / 8), ck = rem % (WS_CP / 8);
        *reinterpret_cast<f32x4*>(&Rsh[(pl * BN + row) * CPL + 8 * ck]) = rreg[rr];
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const int idx = tid + 256 * rr;
        const int c = idx / (BN / 4), q = idx % (BN / 4);
        if (c < a.t) {
          float* dst = Rs + (size_t)(4 * q) * RS + (c & 1) * TH4 + (c >> 1);
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[(size_t)e * RS] = rreg[rr][e];
        }
      }
    }
    if (tid < BN) {
      const int j = j0 + tid;
      if (j < jend) sub_center<DP>(zreg, cz);
      gram_pack_a<D>(zreg, j < jend, Xh, tid, BN);
      if constexpr (MODE == 1) {
        // WSPLIT: the W tile of this row's 32-row block carries the block sign s_J (kv_wsplit.hpp); the columns it is contracted against take it back
        const float sj = WSPLIT ? ws_blocksign(j >> 5) : 1.f;
        Zs[tid] = j < jend ? sj : 0.f;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          Zs[(1 + q) * LDZ + tid] = zreg[q] * sj;
          Zs[(1 + D + q) * LDZ + tid] = zreg[q] * zreg[q] * sj;
        }
      }
    }
    __syncthreads();
    if (jn < jend) fetch(jn);

    // ---- W^T tiles of the two 32-row j blocks on the matrix pipe (A = R, B = L; 4 k-steps per LDS read)
    f32x16 w0, w1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { w0[r] = 0.f; w1[r] = 0.f; }
    const float* lrow = Ls + l31 * RS + h * TH4;
    const float* r0row = Rs + (size_t)l31 * RS + h * TH4;
    const float* r1row = Rs + (size_t)(32 + l31) * RS + h * TH4;
    if constexpr (WSPLIT) {
      // A = R planes (rows j of the two 32-row halves), B = L planes (registers); the two small products first, then the leading one
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const f16x8 ah0 = *reinterpret_cast<const f16x8*>(&Rsh[(l31) * CPL + 16 * ks + 8 * h]);
        const f16x8 al0 = *reinterpret_cast<const f16x8*>(&Rsh[(BN + l31) * CPL + 16 * ks + 8 * h]);
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(&Rsh[(32 + l31) * CPL + 16 * ks + 8 * h]);
        const f16x8 al1 = *reinterpret_cast<const f16x8*>(&Rsh[(BN + 32 + l31) * CPL + 16 * ks + 8 * h]);
        w0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, lbl[ks], w0, 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, lbl[ks], w1, 0, 0, 0);
        w0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, lbh[ks], w0, 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, lbh[ks], w1, 0, 0, 0);
        w0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, lbh[ks], w0, 0, 0, 0);
        w1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, lbh[ks], w1, 0, 0, 0);
      }
    } else {
      // (fetching the operands of the next four k-steps ahead of the eight MFMAs was measured SLOWER: 368 vs 349 ms,
      // profiles/r02_s6_grad_timing_operand_prefetch_slower.json -- the second resident wave already covers the LDS latency)
      for (int c4 = 0; c4 < TH4; c4 += 4) {
        const f32x4 lb = *reinterpret_cast<const f32x4*>(lrow + c4);
        const f32x4 ra = *reinterpret_cast<const f32x4*>(r0row + c4);
        const f32x4 rbv = *reinterpret_cast<const f32x4*>(r1row + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[e], lb[e], w0, 0, 0, 0);
          w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rbv[e], lb[e], w1, 0, 0, 0);
        }
      }
    }
    // ---- squared distances of the same two blocks (Gram form)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const f16x8 a0 = *reinterpret_cast<const f16x8*>(&Xh[gram_a_off(kh, l31, h, BN)]);
      const f16x8 a1 = *reinterpret_cast<const f16x8*>(&Xh[gram_a_off(kh, 32 + l31, h, BN)]);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq[kh], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq[kh], s1, 0, 0, 0);
    }
    mfma_result_fence(w0, w1, s0, s1);   // VGPR-destination MFMA results are read by the VALU next: data-dependent fence (common.hpp)
    // ---- consume: k, dk/ds, A = W dk/ds
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
    f32x4 zacc[GZ];
#pragma unroll
    for (int gz = 0; gz < GZ; ++gz) zacc[gz] = (f32x4)(0.f);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float fh0 = 0.f, fh1 = 0.f, fh2 = 0.f;   // this 32-row j block alone: WSPLIT un-flips its block sign below
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 zv[GZ];
        if constexpr (MODE == 1) {
          const int jl = half * 32 + 8 * q4 + 4 * h;
#pragma unroll
          for (int gz = 0; gz < GZ; ++gz) zv[gz] = *reinterpret_cast<const f32x4*>(&Zs[(4 * gz + l3) * LDZ + jl]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e;
          float sv = half ? s1[r] : s0[r];
          const float w = half ? w1[r] : w0[r];
          float av;
          if constexpr (KIND == KIND_RBF) {
            // dk/ds = -ln2 k: accumulate with A' = W k and apply -ln2 once per step (RBF_DK); a slightly negative S (cancellation)
            // needs no clamp for 2^-S.  3 VALU ops + v_exp_f32 per pair instead of 6.
            av = w * __builtin_amdgcn_exp2f(-sv);
            fh0 += av;
          } else {
            sv = __builtin_amdgcn_fmed3f(sv, 0.f, 3.0e38f);
            float kv, dk, dpar;
            cov_and_dcov<KIND>(sv, a.kparam, kv, dk, dpar);
            fh0 = __builtin_fmaf(w, kv, fh0);
            if constexpr (KIND == KIND_RQ) fh2 = __builtin_fmaf(w, dpar, fh2);
            av = w * dk;
          }
          if constexpr (MODE == 0) {
            fh1 = __builtin_fmaf(av, sv, fh1);
          } else {
#pragma unroll
            for (int gz = 0; gz < GZ; ++gz) zacc[gz] = __builtin_amdgcn_mfma_f32_4x4x1f32(zv[gz][e], av, zacc[gz], 0, 0, 0);
          }
        }
      }
      const float sh = WSPLIT ? ws_blocksign((j0 >> 5) + half) : 1.f;   // wave-uniform
      f0 = __builtin_fmaf(sh, fh0, f0);
      f1 = __builtin_fmaf(sh, fh1, f1);
      f2 = __builtin_fmaf(sh, fh2, f2);
    }
    if constexpr (MODE == 1) mfma_result_fence(zacc);   // zacc (4x4x1 MFMA results in VGPRs) is read next
    constexpr float RBF_DK = KIND == KIND_RBF ? -0.6931471805599453f : 1.0f;   // see the consume loop
    g[0] += (double)f0;
    g[1 + DP] += (double)f2;
    if constexpr (MODE == 0) {
      g[1] += (double)(RBF_DK * f1);
    } else {
      // columns: 0 -> rs, 1 + q -> u_q, 1 + D + q -> v_q   (this lane's j half; the combination is linear in them)
      const float rs = RBF_DK * zacc[0][0];
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const float u = RBF_DK * zacc[(1 + q) / 4][(1 + q) % 4];
        const float v = RBF_DK * zacc[(1 + D + q) / 4][(1 + D + q) % 4];
        const double zq = (double)zi[q];
        g[1 + q] += zq * zq * (double)rs - 2.0 * zq * (double)u + (double)v;
        gx[q] += 2.f * (zi[q] * rs - u);
      }
    }
  }

  if constexpr (WSPLIT) {
    // this lane's row entered the planes as s_i L_i (kv_wsplit.hpp, ws_rowsign): every sum above is linear in W = L^T R, take the sign back
    const float sg = ws_rowsign(i);
#pragma unroll
    for (int q = 0; q <= DP + 1; ++q) g[q] *= (double)sg;
#pragma unroll
    for (int q = 0; q < DP; ++q) gx[q] *= sg;
  }
#pragma unroll
  for (int q = 0; q <= DP + 1; ++q) {
    double v = wave_sum(g[q]);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  const double wsc = WSPLIT ? (double)*a.wscale : 1.0;   // WSPLIT: the planes carry scaled vectors (kv_wsplit.hpp)
  if (tid <= DP + 1) a.part[(int64_t)unit * (2 + DP) + tid] = wsc * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
  if constexpr (MODE == 1) {
    if (a.Px) {
      float* Pout = a.Px + (int64_t)s * a.pxstride;
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const float tot = gx[q] + __shfl_xor(gx[q], 32, 64);
        if (h == 0 && i < a.n) Pout[(int64_t)q * a.ldx + i] = tot * (float)wsc;
      }
    }
  }
}

// out[q] = sum_u part[u][q]   (1 block of 256 threads; fixed order -> reproducible)
template <int UNUSED>
__global__ __launch_bounds__(256) void grad2_finalize_kernel(const double* __restrict__ part, int units, int nq, float* __restrict__ out) {
  __shared__ double sm[4];
  for (int q = 0; q < nq; ++q) {
    double acc = 0.0;
    for (int u = threadIdx.x; u < units; u += 256) acc += part[(int64_t)u * nq + q];
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) out[q] = (float)acc;
    __syncthreads();
  }
}

}  // namespace gpamd
