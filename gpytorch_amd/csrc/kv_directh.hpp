// Fused covariance MVM for clouds OUTSIDE the accuracy policy of the quadratic expansion: squared distances by DIRECT differences on the
// packed-f32 vector pipe, contraction of hi/lo-split operands on the f16 matrix pipe (the contraction half of kv_gramh.hpp).
//
// Why.  Until round 5 every product whose rows could not be block-centred (backend.gram_mode == 0: short lengthscales on curve-like or very
// sparse clouds, Matern nu = 1/2; and the WIDE rows of a block-centred product) ran on kernels that carry the contraction on the VALU
// (kv_valu.hpp, <= 16 columns: D + T/2 packed instructions per pair) or on v_mfma_f32_32x32x2_f32 (kv_mfma.hpp, 1024 matrix-pipe cycles per
// 32 x 32 block and 32-column tile).  With the default ten probe vectors + y that is 11 multiply-adds per pair against ~6 instructions of
// generation: the contraction is two thirds of the VALU work (road3d-shaped workload: 23-28 ms per product, 87 % of a training iteration,
// profiles/r05_s4_workload_road3d_kernel_stats.csv).  The split contraction moves it to three v_mfma_f32_32x32x16_f16 per 16 contracted rows
// (192 matrix-pipe cycles per block) at the price of 2 VALU instructions per pair for the hi/lo split of K.
//
// Layout.  Exactly kv_gramh.hpp's: lane (h, i = l31) of a wave owns the 16 pairs (j(r, h), i), r = 0..15, j(r, h) = (r & 3) + 8 (r >> 2) + 4 h, of a
// 32 x 32 block, i.e. the eight B-operand slots of contraction MFMA mf = r >> 3; the V planes, the pre-pass (kv_vsplit.hpp), the column
// multipliers and the partial-slab convention are shared with that kernel (KvhArgs).  What differs is where S comes from: the x_j rows of a tile
// are staged TRANSPOSED in LDS (Xf[k][j], float), so that one ds_read_b128 per dimension returns the four CONSECUTIVE rows j(4 q .. 4 q + 3, h)
// of a quad -- adjacent register pairs = the two operands of v_pk_add_f32 / v_pk_fma_f32 -- and all lanes of a half-wave read the same address
// (broadcast, conflict-free).  Per pair of elements: D packed subtractions + D packed multiply-adds, then cov_pair_from_sq and the split as in
// kv_gramh.hpp.  One or two 32-column tiles + an optional extra column on the VALU (round 6: K is generated ONCE for up to 65 columns; until then
// 33-65 columns went in groups of 32 and regenerated K for each).
//
// Software pipeline: as kv_gramh.hpp -- the B operands of step s + 1 are generated between the MFMAs of step s (sched_barrier-pinned slices); the
// x_j rows are staged one tile ahead (double-buffered), the V planes per tile.  The loop is VALU-bound (D = 3 Matern-5/2: ~110 VALU instructions
// against 6 MFMAs = 192 cycles per block), so the MFMAs ride for free; what the kernel buys is the VALU work it no longer does.
#pragma once
#include "kv_gramh.hpp"

namespace gpamd {

constexpr int KDH_MAX_DIM = 10;   // instantiated for D in {1,2,3,4,5,6,8,10}: beyond, the per-half x_j registers (8 D) no longer fit next to the operands
constexpr int KDH_COLS = 64;      // columns per launch group (+ 1 extra VALU column): one or two 32-column tiles, K generated ONCE for all of them
// 32-row tiles per wave: two, or one for few output rows -- and with TWO column tiles beyond four dimensions, where 64 accumulators next to the
// 8 D per-half x_j registers of two row tiles would not fit two waves per SIMD
constexpr int kdh_ni(bool small, int ct = 1, int dk = 1) { return (small || (ct == 2 && dk > 4)) ? 1 : 2; }
inline int kdh_bm(int ni) { return 4 * ni * 32; }

template <int KIND, int D, int NI, int CT = 1, int EX = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void kv_directh_kernel(KvhArgs ka) {
  constexpr int NW = 4, NT = 64 * NW;
  const KvArgs& a = ka.a;
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int BN = KGH_BN, LDH = KGH_LDH, TC = 32 * CT;
  constexpr int XFS = D * BN;   // floats of one x_j buffer
  __shared__ __attribute__((aligned(16))) _Float16 Vhs[TC * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Vls[TC * LDH];
  __shared__ __attribute__((aligned(16))) float Xf[2 * XFS];   // [buf][k][j]
  __shared__ __attribute__((aligned(16))) float Es[EX ? 2 * BN : 4];   // [buf][j] extra column (f32, carried on the VALU as in kv_gramh.hpp)

  if (a.done && *a.done) return;
  float negone;   // -1.0f the optimiser cannot see through (gen_b, kv_gramh.hpp)
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(negone));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;                  // multiple of BN
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (NW * NI * 32) + wave * (NI * 32);

  float zi[NI][D];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = min(ibase + ni * 32 + l31, a.n - 1);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < D) zi[ni][4 * q + e] = v[e];
    }
  }

  f32x16 acc[NI][CT];
  f32x2 eacc2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    eacc2[ni] = (f32x2)(0.f);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ct][r] = 0.f;
  }

  constexpr int VQ = TC * (BN / 8) / NT;   // 16-byte chunks per thread and plane (= 2 CT)

  // x_j rows of the tile starting at j0 -> buffer `buf`, transposed (rows beyond jend: zero -> a finite k against V = 0)
  float xz[DP];
  f32x4 xe = {0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int j0) {
    if (tid < BN) {
      const int j = j0 + tid;
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP + 4 * q);
        xz[4 * q + 0] = v[0]; xz[4 * q + 1] = v[1]; xz[4 * q + 2] = v[2]; xz[4 * q + 3] = v[3];
      }
    }
    if constexpr (EX) {
      if (tid >= BN && tid < BN + BN / 4) {
        const int j = j0 + 4 * (tid - BN);
        const float* src = a.Vt + (int64_t)TC * a.ldv + j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j + 4 <= jend) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < jend) v[e] = src[e];
        }
        xe = v;
      }
    }
  };
  auto store_x = [&](int buf) {
    if (tid < BN) {
#pragma unroll
      for (int k = 0; k < D; ++k) Xf[buf * XFS + k * BN + tid] = xz[k];
    }
    if constexpr (EX) {
      if (tid >= BN && tid < BN + BN / 4) *reinterpret_cast<f32x4*>(&Es[buf * BN + 4 * (tid - BN)]) = xe;
    }
  };
  // extra column: the rows of half mf of block jb this lane pairs with -- ev[q] = rows jb + 16 mf + 8 q + 4 h .. + 3 (the quads of load_zq)
  auto load_ev = [&](int buf, int jb, int mf, f32x4* ev) {
    if constexpr (EX) {
      ev[0] = *reinterpret_cast<const f32x4*>(&Es[buf * BN + jb + 16 * mf + 4 * h]);
      ev[1] = *reinterpret_cast<const f32x4*>(&Es[buf * BN + jb + 16 * mf + 8 + 4 * h]);
    }
  };

  // the two quads of rows a lane needs for half mf of block jb: rows jb + 16 mf + 8 q + 4 h .. + 3, one 16-byte read per dimension and quad
  auto load_zq = [&](int buf, int jb, int mf, f32x4 (*zq)[D]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int k = 0; k < D; ++k) zq[q][k] = *reinterpret_cast<const f32x4*>(&Xf[buf * XFS + k * BN + jb + 16 * mf + 8 * q + 4 * h]);
  };
  // Generation of elements r = 8 mf + 2 p, + 1 in two halves (kv_gramh.hpp):  gen_a: squared distances of the pair (2 D packed instructions), K = f(S)
  // for both, packed hi word;  gen_b: lo = K - hi, packed lo word
  auto gen_a = [&](const f32x4 (*zq)[D], const f32x4* ev, int p, int ni, f32x2& kv, u32x4& bh) {
    const int q = p >> 1, e0 = 2 * (p & 1);
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const f32x2 df = (f32x2){zi[ni][k], zi[ni][k]} - (f32x2){zq[q][k][e0], zq[q][k][e0 + 1]};
      s2 = __builtin_elementwise_fma(df, df, s2);
    }
    kv = cov_pair_from_sq<KIND>(s2, a.kparam, (float)KGH_KSHIFT);
    if constexpr (EX) eacc2[ni] = __builtin_elementwise_fma(kv, (f32x2){ev[q][e0], ev[q][e0 + 1]}, eacc2[ni]);
    bh[p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(kv[0], kv[1]));
  };
  auto gen_b = [&](const f32x2& kv, int p, uint32_t hiw, u32x4& bl) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 hv = __builtin_bit_cast(f16x2, hiw);
    const float l0 = __builtin_fmaf((float)hv[0], negone, kv[0]);
    const float l1 = __builtin_fmaf((float)hv[1], negone, kv[1]);
    bl[p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(l0, l1));
  };
  auto gen_half = [&](int buf, int jb, int mf, int ni, u32x4& bh, u32x4& bl) {
    f32x4 zq[2][D], ev[2];
    load_zq(buf, jb, mf, zq);
    load_ev(buf, jb, mf, ev);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x2 kv;
      gen_a(zq, ev, p, ni, kv, bh);
      gen_b(kv, p, bh[p], bl);
    }
  };

  // prologue: x rows of the first tile, B operands of its first step
  const int* tl = kv_tile_list(a, unit);   // far-pair tile culling (kv_mfma.hpp): the look-ahead staging takes the next SURVIVING tile; no list: every tile
  const int jfirst = kv_tile_at<BN>(tl, jbeg, 0);
  load_x(jfirst);
  store_x(0);
  __syncthreads();
  u32x4 bh[2], bl[2];
  gen_half(0, 0, 0, 0, bh[0], bl[0]);
  gen_half(0, 0, 1, 0, bh[1], bl[1]);

  u32x4 pvh[VQ], pvl[VQ];
  auto load_v = [&](int j0) {
    const int64_t jc = min((int64_t)j0, ka.ldh - BN);   // past the chunk end: any in-bounds tile (never consumed)
#pragma unroll
    for (int rr = 0; rr < VQ; ++rr) {
      const int idx = tid + NT * rr;
      const int c = idx / (BN / 8), q = idx % (BN / 8);
      const int64_t off = (int64_t)c * ka.ldh + jc + 8 * q;
      pvh[rr] = *reinterpret_cast<const u32x4*>(ka.Vh + off);
      pvl[rr] = *reinterpret_cast<const u32x4*>(ka.Vl + off);
    }
  };
  int buf = 0;
  for (int j0 = jfirst, jn, tk = 1; j0 < jend; j0 = jn, buf ^= 1, ++tk) {
    jn = kv_tile_at<BN>(tl, jbeg, tk);
    __syncthreads();   // every wave is done with the V planes of the previous tile and with Xf[buf ^ 1]
    {
      load_v(j0);
      load_x(jn);        // past the end of the chunk: zero rows (the last step's look-ahead generation must stay finite)
      store_x(buf ^ 1);
#pragma unroll
      for (int rr = 0; rr < VQ; ++rr) {
        const int idx = tid + NT * rr;
        const int c = idx / (BN / 8), q = idx % (BN / 8);
        *reinterpret_cast<u32x4*>(&Vhs[c * LDH + 8 * q]) = pvh[rr];
        *reinterpret_cast<u32x4*>(&Vls[c * LDH + 8 * q]) = pvl[rr];
      }
    }
    __syncthreads();

    auto load_a = [&](int jb, f16x8 (*ah)[CT], f16x8 (*al)[CT]) {
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int o = (ct * 32 + l31) * LDH + jb + 16 * mf + 8 * h;
          ah[mf][ct] = *reinterpret_cast<const f16x8*>(&Vhs[o]);
          al[mf][ct] = *reinterpret_cast<const f16x8*>(&Vls[o]);
        }
    };
    f16x8 ah[2][CT], al[2][CT];
    load_a(0, ah, al);
#pragma unroll 2
    for (int jb = 0; jb < BN; jb += 32) {
      f16x8 ahn[2][CT], aln[2][CT];
      load_a((jb + 32) & (BN - 1), ahn, aln);   // unconditional (after the last block: a harmless re-read of block 0), the loop body stays branch-free
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        // next step: (jb, ni + 1), or the first row tile of the next block -- of the NEXT LDS tile after the last block (its x rows are already
        // staged; past the end of the chunk they are zero rows and the result is never used)
        const bool wrap = (ni == NI - 1);
        const int nin = wrap ? 0 : ni + 1;
        const int jbn = wrap ? ((jb + 32) & (BN - 1)) : jb;
        const int bufn = (wrap && jb == BN - 32) ? (buf ^ 1) : buf;
        u32x4 bhn[2], bln[2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          const f16x8 bhv = __builtin_bit_cast(f16x8, bh[mf]);
          const f16x8 blv = __builtin_bit_cast(f16x8, bl[mf]);
          f32x4 zq[2][D], ev[2];
          load_zq(bufn, jbn, mf, zq);   // x_j rows of this half of the NEXT step: in flight under the first MFMA
          load_ev(bufn, jbn, mf, ev);
          f32x2 kv[4];
#pragma unroll
          for (int q = 0; q < 3 * CT; ++q) {
            // the two small terms first, then the leading one; consecutive instructions alternate accumulators
            const int ct = q % CT, term = q / CT;
            acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[mf][ct] : ah[mf][ct], term == 1 ? blv : bhv, acc[ni][ct], 0, 0, 0);
            // half-chunk u = 2 p + (0: gen_a, 1: gen_b); this MFMA's share: [u0, u1):
            //   CT = 1:  a0 b0 a1 | b1 a2 b2 | a3 b3          CT = 2:  a0 | b0 a1 | b1 | a2 | b2 a3 | b3     (kv_gramh.hpp)
            constexpr int U6[7] = {0, 1, 3, 4, 5, 7, 8}, U3[4] = {0, 3, 6, 8};
            const int u0 = CT == 1 ? U3[q] : U6[q], u1 = CT == 1 ? U3[q + 1] : U6[q + 1];
#pragma unroll
            for (int u = u0; u < u1; ++u) {
              if ((u & 1) == 0) gen_a(zq, ev, u >> 1, nin, kv[u >> 1], bhn[mf]);
              else gen_b(kv[u >> 1], u >> 1, bhn[mf][u >> 1], bln[mf]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        bh[0] = bhn[0]; bh[1] = bhn[1]; bl[0] = bln[0]; bl[1] = bln[1];
      }
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { ah[mf][ct] = ahn[mf][ct]; al[mf][ct] = aln[mf][ct]; }
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
          const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[ni][ct][r] * ka.colmul[c];
        }
    }
    if constexpr (EX) {
      const float part = eacc2[ni][0] + eacc2[ni][1];
      const float tot = part + __shfl_xor(part, 32, 64);
      if (h == 0 && i < a.n) Pout[(int64_t)TC * a.ldo + i] = tot * ka.colmul[TC];
    }
  }
}

}  // namespace gpamd
