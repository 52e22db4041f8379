// Fused K*V, Gram-form generation (gram_f16.hpp), software-pipelined: the K block of step b+1 is generated UNDER the
// contraction MFMAs of step b, and the V tile of the next 128 j arrives by DMA into the other LDS buffer.
//
// kv_gram.hpp runs, per 32-row j block and wave: Gram MFMAs -> 32 v_exp -> 64 contraction MFMAs, and per 128-j tile:
// barrier, global -> registers -> LDS, barrier.  Three waves per SIMD overlap those phases only statistically; the
// matrix pipe idles 13.5 % of the time (profiles/r01_s17).  Here ONE wave keeps the pipe fed:
//   * two K blocks live in registers (kk[0/1], 32 + 32 VGPRs): while the 64 contraction MFMAs of block b issue
//     (one per 64 cycles), the same wave interleaves the split of the x_j rows of block b+1, its Gram MFMA(s) and its
//     32 v_exp -- independent instructions that issue between MFMAs;
//   * the A operands (V) of MFMA group g+1 are read from LDS before the MFMAs of group g issue;
//   * V tiles and the raw x_j rows travel global -> LDS by global_load_lds_dwordx4 (no staging registers, no ordinary
//     global loads in the loop, hence no compiler-inserted vmcnt(0) stalls); ONE barrier per 128-j tile.
// Two workgroups per CU (2 x 33 KB V buffers + 3 x-row buffers), two waves per SIMD, <= 256 registers per lane.
//
// LDS image of a V tile: row c (probe column) = 128 floats, NO padding (the DMA writes wave-uniform base + lane*16 B);
// bank conflicts of the A-operand reads (32 lanes = 32 rows, same 16-B chunk) are avoided by an XOR swizzle applied on
// the SOURCE side: chunk q of row c is stored at chunk position q ^ (c & 31).  The EX column is row TC (c & 31 = 0:
// unswizzled, read as a broadcast).  Rows >= t are copies of row t-1 (never stored).  x_j rows of tile T+2 are fetched
// when tile T starts, so the first block of tile T+1 can be generated before the barrier that publishes V(T+1).
// The last, partial tile of the contracted range is staged synchronously with zero fill.  D <= 4 (one 16-byte row).
#pragma once
#include "../gram_f16.hpp"
#include "../kv_mfma.hpp"

namespace gpamd {

template <int KIND, int D, int CT, int NI, int EX>
__global__ __launch_bounds__(256) void kv_gram2_kernel(KvArgs a) {
  static_assert(D <= 4, "one float4 per point");
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = KV_BN, TC = 32 * CT;
  constexpr int NR = TC + (EX ? 2 : 0);        // rows of the V image (EX column + one duplicate row keep NR even)
  constexpr int VF = NR * BN;                  // floats per V buffer
  constexpr int XF = BN * 4;                   // floats per x-row buffer
  static_assert(BN == 128, "the swizzle below assumes 32 chunks of 16 B per row");
  __shared__ __attribute__((aligned(16))) float smem[2 * VF + 3 * XF];
  float* const Xr = smem + 2 * VF;

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 32) + wave * (NI * 32);
  const int ntile = (jend - jbeg + BN - 1) / BN;
  const int nblk = (jend - jbeg + 31) / 32;

  f16x8 bq[NI][KH];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 32 + l31, a.n - 1);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * 4);
    float z[4] = {v[0], v[1], v[2], v[3]};
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

  // ---- staging -------------------------------------------------------------------------------------------------
  auto dma_v = [&](int tile) {  // full tiles only
    float* V = smem + (tile & 1) * VF;
    const int j0 = jbeg + tile * BN;
#pragma unroll
    for (int rp = 0; rp < (NR / 2 + 3) / 4; ++rp) {
      const int pair = wave + 4 * rp;
      if (pair < NR / 2) {
        const int c = 2 * pair + h;
        const int col = min(c, a.t - 1);
        const int q = l31 ^ (c & 31);
        const float* src = a.Vt + (int64_t)col * a.ldv + j0 + 4 * q;
        __builtin_amdgcn_global_load_lds((const void*)src, (void __attribute__((address_space(3)))*)(V + pair * 2 * BN), 16, 0, 0);
      }
    }
  };
  auto sync_v = [&](int tile) {  // partial (last) tile: plain loads, zero fill
    float* V = smem + (tile & 1) * VF;
    const int j0 = jbeg + tile * BN;
    for (int idx = tid; idx < NR * (BN / 4); idx += 256) {
      const int c = idx / (BN / 4), q = idx % (BN / 4);
      const int col = min(c, a.t - 1);
      const int j = j0 + 4 * q;
      const float* src = a.Vt + (int64_t)col * a.ldv + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j + 4 <= jend) {
        v = *reinterpret_cast<const f32x4*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (j + e < jend) v[e] = src[e];
      }
      *reinterpret_cast<f32x4*>(&V[c * BN + ((q ^ (c & 31)) << 2)]) = v;
    }
  };
  auto stage_v = [&](int tile) {
    if (tile >= ntile) return;
    if (jbeg + (tile + 1) * BN <= jend) dma_v(tile); else sync_v(tile);
  };
  auto dma_x = [&](int tile) {  // 128 raw x_j rows (2 KB): waves 0 and 1, one instruction each; rows clamped to m-1
    if (tile >= ntile || wave >= 2) return;
    const int j = min(jbeg + tile * BN + wave * 64 + lane, a.m - 1);
    __builtin_amdgcn_global_load_lds((const void*)(a.X2 + (int64_t)j * 4),
                                     (void __attribute__((address_space(3)))*)(Xr + (tile % 3) * XF + wave * 256), 16, 0, 0);
  };

  // ---- generation of one K block (32 j rows x NI*32 outputs): squared distances into kk (k = f(S) applied later) -----
  auto gen_sq = [&](int blk, f32x16* kk) {
    const int tile = blk >> 2;
    const int jl = (blk & 3) * 32 + l31;
    const f32x4 v = *reinterpret_cast<const f32x4*>(&Xr[(tile % 3) * XF + jl * 4]);
    const bool valid = jbeg + blk * 32 + l31 < jend;
    float z[4] = {valid ? v[0] : 0.f, valid ? v[1] : 0.f, valid ? v[2] : 0.f, valid ? v[3] : 0.f};
    f16x8 aq[KH];
    gram_pack_a_lane<D>(z, valid, h, aq);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) kk[ni][r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) kk[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[kh], bq[ni][kh], kk[ni], 0, 0, 0);
  };
  auto cov_inplace = [&](f32x16* kk, int e) {  // element e of the NI*16 values of a block
    const int ni = e >> 4, r = e & 15;
    float sv = kk[ni][r];
    if constexpr (KIND != KIND_RBF) sv = __builtin_amdgcn_fmed3f(sv, 0.f, 3.0e38f);
    kk[ni][r] = cov_from_sq<KIND>(sv, a.kparam);
  };

  auto read_av = [&](const float* V, int jb, int g, f32x4* av, f32x4& ev) {
    const int q = (jb >> 2) + 2 * g + h;  // 16-B chunk of this half-wave's 4 consecutive j rows
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(&V[(ct * 32 + l31) * BN + ((q ^ l31) << 2)]);
    if constexpr (EX) ev = *reinterpret_cast<const f32x4*>(&V[TC * BN + (q << 2)]);
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  dma_x(0);
  dma_x(1);
  stage_v(0);
  __syncthreads();
  f32x16 kka[NI], kkb[NI];
  gen_sq(0, kka);
#pragma unroll
  for (int e = 0; e < NI * 16; ++e) cov_inplace(kka, e);

  // ---- main loop: two blocks per iteration so the K double buffer is addressed statically --------------------------
  auto step = [&](int blk, f32x16* kcur, f32x16* knext) {
    const int tile = blk >> 2;
    if ((blk & 3) == 0) {
      // tile start: V(tile) and x(tile+1) have landed and are visible after the barrier; everyone is done with
      // V(tile-1), whose buffer receives V(tile+1)
      if (blk > 0) __syncthreads();
      stage_v(tile + 1);
      dma_x(tile + 2);
    }
    const float* V = smem + (tile & 1) * VF;
    const int jb = (blk & 3) * 32;
    f32x4 av[2][CT], ev[2];
    read_av(V, jb, 0, av[0], ev[0]);
    // squared distances of the NEXT block (the last iteration regenerates its own block: no branch in the pipeline)
    gen_sq(min(blk + 1, nblk - 1), knext);
    constexpr int EPS = NI * 16 / 16;  // k = f(S) evaluations of the next block folded into each of the 16 MFMA steps
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) read_av(V, jb, g + 1, av[(g + 1) & 1], ev[(g + 1) & 1]);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][ct][st], kcur[ni][4 * g + st], acc[ni][ct], 0, 0, 0);
          if constexpr (EX) eacc[ni] = __builtin_fmaf(kcur[ni][4 * g + st], ev[g & 1][st], eacc[ni]);
        }
#pragma unroll
        for (int e = 0; e < EPS; ++e) cov_inplace(knext, (4 * g + st) * EPS + e);
        // pin the interleaving: (at the start of a group: the LDS reads of the NEXT group,) NI*CT MFMAs, then this
        // step's share of transcendental / VALU work
        if (st == 0 && g < 3) __builtin_amdgcn_sched_group_barrier(0x100, CT + EX, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NI * CT, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, EPS * (KIND == KIND_RBF ? 1 : 2), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, EPS * (KIND == KIND_RBF ? 0 : 5) + (EX ? NI : 0), 0);
      }
    }
  };
#pragma unroll 1
  for (int blk = 0; blk < nblk; blk += 2) {
    step(blk, kka, kkb);
    if (blk + 1 < nblk) step(blk + 1, kkb, kka);
  }

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
