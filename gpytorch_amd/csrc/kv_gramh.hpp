// Fused covariance MVM with BOTH matrix products on the f16 matrix pipe at f32 accuracy: Gram-form generation
// (gram_f16.hpp) + contraction of hi/lo-SPLIT operands.
//
// Why.  On CDNA4 v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16 rate (157 TFLOP/s against 2.5 PFLOP/s dense), and the
// 64-column product of kv_gram.hpp already sits at 0.86 of that fp32 peak: the f32 matrix pipe is the wall.  The same
// trick that moved the squared distances to the f16 pipe applies to the contraction.  Every f32 operand is written
//     K = Kh + Kl,  V = Vh + Vl        (hi = f16(x), lo = f16(x - hi): 21-22 significant bits, EXACT f16 x f16 products,
//                                        f32 accumulation inside the MFMA)
//     K V  ~=  Kh Vh + Kh Vl + Kl Vh                                   (the dropped Kl Vl term is <= 2^-21 relative)
// i.e. three v_mfma_f32_32x32x16_f16 (8 x the f32 MAC rate each) replace eight v_mfma_f32_32x32x2_f32 per 16 contracted
// rows: 2.7 x less matrix-pipe time per column tile.  K itself is only known to ~5e-6 relative (the cancellation error of
// the quadratic expansion, gram_f16.hpp), V enters with 2^-22, so the product is as accurate as the fp32-MFMA kernel's
// (measured against the float64 oracle in tests/test_gpu_kv_split.py, same 2e-5 bound).
//
// Range.  f16 holds 6e-8 .. 65504, so both operands are scaled by powers of two (exact):
//   * K is generated as 2^KSHIFT K (for the RBF the shift is folded into the |z_i|^2 slot of the Gram operands, for the
//     other families into the exp2 argument: free).  K <= 1 -> hi <= 4096; entries below 2^-26 lose relative (not absolute)
//     precision.  hi is rounded toward zero (one v_cvt_pkrtz per TWO elements), lo = K - hi >= 0 comes from one
//     v_fma_mix_f32 per element, then a second v_cvt_pkrtz: 2 VALU slots per pair beside the v_exp_f32.
//   * V is split ONCE per product by vsplit_kernel (an HBM-bound pre-pass, ~1 % of the launch): column c is scaled so that
//     max |V_c| lands in [2^13, 2^14), split with round-to-nearest, and stored as two f16 planes in the k-slot order of the
//     MFMA A operand (below), so that staging is a plain 16-byte copy and the operand read one ds_read_b128.
//   * the epilogue multiplies column c by colmul[c] = 2^-KSHIFT / scale_c (exact).
//
// Operand layout.  The Gram MFMA leaves lane (h, i) with S[j(r,h)][i], r = 0..15, j(r,h) = (r&3) + 8(r>>2) + 4h.  For the
// contraction  P[c][i] += sum_j V[c][j] K[j][i]  as D[m=c][n=i] = A[m=c][k] B[k][n=i]  with 16 k-slots per instruction:
//   MFMA mf (0/1) of a 32-row block takes registers r = 8 mf + e, e = 0..7, of this lane as B[k = 8h + e][i]
//     (no cross-lane movement: the lane's own 8 K elements, packed to f16)
//   and needs A[c][k = 8h + e] = V[c][j0 + 16 mf + (e&3) + 8(e>>2) + 4h]: the V planes store row c with position
//     16 g + 8h + e  holding  j = 16 g + (e&3) + 8(e>>2) + 4h   -> the operand is 16 contiguous bytes.
// The optional extra column (EX: the "+ y" of [probes | y]) is carried on the VALU in f32 as in kv_gram.hpp.
#pragma once
#include "gram_f16.hpp"
#include "kv_mfma.hpp"

namespace gpamd {

constexpr int KGH_BN = 128;             // j tile staged in LDS
constexpr int KGH_LDH = KGH_BN + 8;     // f16 row stride of the V planes in LDS (272 B: conflict-free ds_read_b128)
constexpr int KGH_KSHIFT = 12;          // K generated as 2^12 K
constexpr int KGH_VEXP = 14;            // max |V_c| * scale_c in [2^13, 2^14)
constexpr int KGH_GROUP = 64;           // columns per launch group (+ 1 extra VALU column)
constexpr int KGH_SMALL_N = 16384;      // fewer output rows: one row tile per wave (NI = 1), four times as many workgroups
constexpr int KGH_MIN_COLS = 5;         // fewer columns: the VALU-contraction kernel (kv_gramv.hpp) wins
constexpr int kgh_ni(int ct, int d = 16) { return d > 16 ? 2 : ((ct == 1 || d <= 3) ? 4 : 2); }   // 32-row tiles per wave: 16 NI CT accumulators (d: kernel dims)
inline int kgh_bm(int ni) { return 4 * ni * 32; }

struct KvhArgs {
  KvArgs a;               // Vt: the f32 columns (only the extra column is read from it)
  const _Float16* Vh;     // [32 CT][ldh] split planes, permuted k-slot order, zero beyond m and beyond t
  const _Float16* Vl;
  int64_t ldh;
  const float* colmul;    // [32 CT (+1)] per-column output multiplier
};

// covariance from the squared distance, times 2^KSHIFT (RBF: the shift is already inside s)
template <int KIND>
__device__ __forceinline__ float cov_scaled(float s, float p) {
  constexpr float KS = (float)KGH_KSHIFT;
  if constexpr (KIND == KIND_RBF) {
    return __builtin_amdgcn_exp2f(-s);
  } else if constexpr (KIND == KIND_RQ) {
    return __builtin_amdgcn_exp2f(__builtin_fmaf(-p, __builtin_amdgcn_logf(1.0f + s), KS));
  } else {
    float r = __builtin_amdgcn_sqrtf(s);
    float e = __builtin_amdgcn_exp2f(__builtin_fmaf(-r, LOG2E, KS));
    if constexpr (KIND == KIND_MATERN12) return e;
    if constexpr (KIND == KIND_MATERN32) return (1.0f + r) * e;
    return __builtin_fmaf(s, 1.0f / 3.0f, 1.0f + r) * e;
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Software pipeline.  One "step" = one 32x32 block of pairs of one row tile: generation (KH Gram MFMAs, then per element
// v_exp_f32 + the hi/lo split: ~4 VALU instructions) and contraction (6 CT MFMAs of 8 passes).  A wave issues in order, and a
// second MFMA cannot issue while the first occupies the pipe, so MFMAs written back to back stall the wave for 32 cycles each
// with the VALU idle.  The loop therefore generates the B operands of step s+1 WHILE the MFMAs of step s run: the source is
// written in that order -- one MFMA, then its share of the next step's VALU work -- and sched_barrier(0) pins it.  To keep the pipeline
// full across LDS tiles the split x_j rows (and the extra column) are staged ONE TILE AHEAD (double-buffered): the last step
// of a tile generates the first block of the next one.
// (The ablation and geometry variants this loop was measured through -- no generation / no MFMAs / staged once / no barriers, one or three
// waves per SIMD, eight waves per workgroup, register prefetch of the next tile, a deeper Gram look-ahead -- live in tune/kv_gramh_ablate.hpp
// and libgpamd_tune.so; DESIGN.md 3.1b has the numbers.)
// (beyond 20 dimensions: KH >= 5 Gram MFMAs per block -- the split x_i / x_j operands no longer fit 256 registers next to 64 accumulators, and the two
// Xh buffers push the LDS image past half a CU: ONE wave per SIMD with the whole register file instead of 46 .. 94 spilled registers)
template <int D>
constexpr int kgh_waves() { return D > 20 ? 1 : 2; }
template <int KIND, int D, int CT, int NI, int EX, int SAFE = 0>   // SAFE = 1 / 2: tune library only (tune/tune_hazard.hip): every Gram result behind the full mfma_result_fence / round 5's form
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kgh_waves<D>(), kgh_waves<D>())))
void kv_gramh_kernel(KvhArgs ka) {
  constexpr int NW = 4, NT = 64 * NW;   // four waves per workgroup: row block = NW * NI * 32 rows sharing one staged V tile
  const KvArgs& a = ka.a;
  constexpr int DP = (D + 3) / 4 * 4, DQ = DP / 4;
  constexpr int KH = GramF16<D>::KH;
  constexpr int BN = KGH_BN, LDH = KGH_LDH, TC = 32 * CT;
  constexpr int XHS = KH * BN * 16;     // f16 elements of one Xh buffer
  constexpr bool PF = NI * CT <= 4 && KH <= 2;   // A operands of block jb + 32 fetched during block jb (16 CT more registers)
  __shared__ __attribute__((aligned(16))) _Float16 Vhs[TC * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Vls[TC * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Xh[2 * XHS];        // [buf][kh][j][16] split augmented x_j rows
  __shared__ __attribute__((aligned(16))) float Es[EX ? 2 * BN : 4];   // [buf][j] extra column

  if (a.done && *a.done) return;
  float negone;   // -1.0f the optimiser cannot see through (gen_b)
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(negone));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;                  // multiple of BN
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (NW * NI * 32) + wave * (NI * 32);
  float cz[DP];   // centre of this workgroup's row block (zero unless the host passed chunk centres: gram_f16.hpp)
  load_center<DP>(a.Xc, ibase - wave * (NI * 32), NW * NI * 32, a.n, cz);

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
    gram_pack_b<D>(z, h, bq[ni], KIND == KIND_RBF ? (float)KGH_KSHIFT : 0.f);
  }

  f32x16 acc[NI][CT];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ct][r] = 0.f;
  }

  constexpr int VQ = TC * (BN / 8) / NT;   // 16-byte chunks per thread and plane (= 2 CT with four waves)

  // split x_j rows + extra column of the tile starting at j0 -> buffer `buf` (rows beyond jend: zero -> k = 2^KSHIFT, V = 0).
  // Two halves: global loads into registers (load_x), then the split rows into LDS (store_x) between the barriers.
  float xz[DP];
  f32x4 xe = {0.f, 0.f, 0.f, 0.f};
  bool xvalid = false;
  auto load_x = [&](int j0) {
    if (tid < BN) {
      const int j = j0 + tid;
      xvalid = j < jend;
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
      if (xvalid) sub_center<DP>(xz, cz);
      gram_pack_a<D>(xz, xvalid, Xh + buf * XHS, tid, BN);
    }
    if constexpr (EX) {
      if (tid >= BN && tid < BN + BN / 4) *reinterpret_cast<f32x4*>(&Es[buf * BN + 4 * (tid - BN)]) = xe;
    }
  };
  auto stage_x = [&](int j0, int buf) {
    load_x(j0);
    store_x(buf);
  };

  auto load_aq = [&](int buf, int jb, f16x8* aq) {
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) aq[kh] = *reinterpret_cast<const f16x8*>(&Xh[buf * XHS + gram_a_off(kh, jb + l31, h, BN)]);
  };
  auto gram = [&](const f16x8* aq, int ni) -> f32x16 {
    f32x16 kk;
#pragma unroll
    for (int r = 0; r < 16; ++r) kk[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) kk = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[kh], bq[ni][kh], kk, 0, 0, 0);
    // The product reads kk on the VALU one contraction MFMA later, behind the toolchain's 12 wait states -- the very distance DESIGN 3.1d measured
    // insufficient in kv_gramv.  Round 6: 8 explicit wait states more, TIED to the result registers (no MFMA can sink below them): 20 in all, as
    // kv_gram_kernel carries since round 5.  Measured against the alternatives on one box (profiles/r06_s4_kv_gramh_fence_ab.json,
    // r06_s5_*): the FULL fence (32 wait states) costs 1.0 - 1.4 % (95.1 -> 96.4 ms at the headline split shape, 129.6 -> 131.0 ms at C3's) --
    // above the 1 % it was allowed --, so the product takes the 8; every variant is bitwise equal to the others on a full chip.
    // (SAFE: 0 = the product; 1 = the full fence -- the stress test's reference; 2 = round 5's form, the toolchain's table only: tune library, A/B)
    if constexpr (SAFE == 1) {
      mfma_result_fence(kk);
    } else if constexpr (SAFE == 0) {
      mfma_tie(kk);
      asm volatile("s_nop 7");   // (+ the toolchain's own 12 behind it: the hazard recogniser does not count wait states inside inline asm)
      mfma_tie(kk);
    }
    return kk;
  };
  // Generation of elements r = 8 mf + 2 p, + 1 of a step in two halves of three VALU instructions each:
  //   gen_a: K = f(S) for both (2 v_exp_f32), packed hi word (v_cvt_pkrtz)      gen_b: lo = K - hi (2 v_fma_mix), packed lo word
  // the extra column's two multiply-adds ride in gen_a as one v_pk_fma_f32.
  f32x2 eacc2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) eacc2[ni] = (f32x2)(0.f);
  auto gen_a = [&](const f32x16& kk, int mf, int p, const f32x4* ev, int ni, f32x2& kv, u32x4& bh) {
    // both elements at once: packed-f32 arithmetic around the transcendentals (common.hpp cov_pair_from_sq); RBF: the 2^KSHIFT scale is already
    // inside S (gram_pack_b's nshift), the other families add it to the exponent
    kv = cov_pair_from_sq<KIND>((f32x2){kk[8 * mf + 2 * p], kk[8 * mf + 2 * p + 1]}, a.kparam, KIND == KIND_RBF ? 0.f : (float)KGH_KSHIFT);
    if constexpr (EX) {
      // rows j(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 8 mf + 2 p + e: ev[p >> 1] holds rows 16 mf + 8 (p >> 1) + 4 h .. + 3
      const f32x2 e2 = {ev[p >> 1][2 * (p & 1)], ev[p >> 1][2 * (p & 1) + 1]};
      eacc2[ni] = __builtin_elementwise_fma(kv, e2, eacc2[ni]);
    }
    bh[p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(kv[0], kv[1]));
  };
  auto gen_b = [&](const f32x2& kv, int p, uint32_t hiw, u32x4& bl) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 hv = __builtin_bit_cast(f16x2, hiw);
    // fmaf((float)h, -1, k) with an OPAQUE -1 (an SGPR the optimiser cannot see through, else it rewrites the fma as k - h and
    // the conversion becomes its own instruction): with f32 denormals flushed (kvh_*.hip are built with
    // -fgpu-flush-denormals-to-zero) the f16 -> f32 extension folds into ONE v_fma_mix_f32 per element.  (hiw comes BY VALUE:
    // __builtin_bit_cast on element p of a `const u32x4&` parameter was compiled as element 0 for every p.  Writing the packed lo
    // word with v_fma_mixlo_f16 / v_fma_mixhi_f16 instead -- one instruction fewer per pair -- measured slower, 90 vs 83 ms.)
    const float l0 = __builtin_fmaf((float)hv[0], negone, kv[0]);
    const float l1 = __builtin_fmaf((float)hv[1], negone, kv[1]);
    bl[p] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(l0, l1));
  };
  auto load_ev = [&](int buf, int jb, int mf, f32x4* ev) {
    if constexpr (EX) {
      ev[0] = *reinterpret_cast<const f32x4*>(&Es[buf * BN + jb + 16 * mf + 4 * h]);
      ev[1] = *reinterpret_cast<const f32x4*>(&Es[buf * BN + jb + 16 * mf + 8 + 4 * h]);
    }
  };
  auto finish_half = [&](const f32x16& kk, int mf, int buf, int jb, int ni, u32x4& bh, u32x4& bl) {
    f32x4 ev[2];
    load_ev(buf, jb, mf, ev);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x2 kv;
      gen_a(kk, mf, p, ev, ni, kv, bh);
      gen_b(kv, p, bh[p], bl);
    }
  };

  // prologue: x rows of the first tile, B operands of its first step.  (Far-pair culling, kv_mfma.hpp: the tile sequence comes from this unit's
  // list -- the look-ahead staging below takes the NEXT SURVIVING tile; no list: jbeg, jbeg + BN, ...)
  const int* tl = kv_tile_list(a, unit);
  const int jfirst = kv_tile_at<BN>(tl, jbeg, 0);
  stage_x(jfirst, 0);
  __syncthreads();
  u32x4 bh[2], bl[2];
  // LEAN (NI * CT > 4: four row tiles per wave and two column tiles, d <= 3): operands of a half are fetched just before its MFMAs
  // instead of per block -- 128 accumulator registers leave no room for the whole block's operands
  constexpr bool LEAN = NI * CT > 4;
  {
    f16x8 aq0[KH];
    load_aq(0, 0, aq0);
    const f32x16 kk = gram(aq0, 0);
    finish_half(kk, 0, 0, 0, 0, bh[0], bl[0]);
    finish_half(kk, 1, 0, 0, 0, bh[1], bl[1]);
  }

  // V planes of one tile: global -> registers (-> LDS between the barriers)
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
    __syncthreads();   // every wave is done with the V planes of the previous tile and with Xh[buf ^ 1]
    {
      load_v(j0);
      load_x(jn);        // past the end of the chunk: zero rows and a zero extra column (the last step's look-ahead generation
                         // must stay finite and add nothing to the extra column)
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

    // A operands of a block (V planes): 16 contiguous bytes per (16-row half, column tile, plane).  Block jb + 32's are
    // fetched during block jb (LDS latency off the critical path); only the first block of a tile waits for them.
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
    f16x8 ah[2][CT], al[2][CT], aqc[KH];
    if constexpr (!LEAN) load_a(0, ah, al);
    load_aq(buf, 0, aqc);
#pragma unroll 2
    for (int jb = 0; jb < BN; jb += 32) {
      // x rows of the next block (of the next tile after the last block: staged one tile ahead)
      f16x8 aqn[KH], ahn[2][CT], aln[2][CT];
      load_aq(jb == BN - 32 ? (buf ^ 1) : buf, (jb + 32) & (BN - 1), aqn);
      if constexpr (PF) load_a((jb + 32) & (BN - 1), ahn, aln);   // unconditional (after the last block: a harmless re-read of block 0); branches in
                                                // this loop body let the optimiser sink the look-ahead generation out of its slots
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        // next step: (jb, ni + 1), or the first row tile of the next block -- of the NEXT LDS tile after the last block (its x
        // rows are already staged; past the end of the chunk they are zero rows and the result is never used)
        const bool wrap = (ni == NI - 1);
        const int nin = wrap ? 0 : ni + 1;
        const int jbn = wrap ? ((jb + 32) & (BN - 1)) : jb;
        const int bufn = (wrap && jb == BN - 32) ? (buf ^ 1) : buf;
        f32x4 ev[2][2];
        if constexpr (!LEAN) {
          load_ev(bufn, jbn, 0, ev[0]);
          load_ev(bufn, jbn, 1, ev[1]);
        }
        const f32x16 kkn = gram(wrap ? aqn : aqc, nin);
        u32x4 bhn[2], bln[2];
        __builtin_amdgcn_sched_barrier(0);
        // contraction of this step, each MFMA followed by its share of the next step's generation; sched_barrier(0) pins the
        // source order (left alone the scheduler groups the MFMAs, and the wave stalls 32 cycles on each with the VALU idle).
        // Eight half-chunks (gen_a / gen_b of four pairs) over the 3 CT MFMAs of a half:
        //   CT = 2:  a0 | b0 a1 | b1 | a2 | b2 a3 | b3          CT = 1:  a0 b0 a1 | b1 a2 b2 | a3 b3
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          const f16x8 bhv = __builtin_bit_cast(f16x8, bh[mf]);
          const f16x8 blv = __builtin_bit_cast(f16x8, bl[mf]);
          f32x2 kv[4];
          if constexpr (LEAN) {
            load_ev(bufn, jbn, mf, ev[mf]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const int o = (ct * 32 + l31) * LDH + jb + 16 * mf + 8 * h;
              ah[mf][ct] = *reinterpret_cast<const f16x8*>(&Vhs[o]);
              al[mf][ct] = *reinterpret_cast<const f16x8*>(&Vls[o]);
            }
          }
#pragma unroll
          for (int q = 0; q < 3 * CT; ++q) {
            // the two small terms first, then the leading one; consecutive instructions alternate accumulators
            const int ct = q % CT, term = q / CT;
            acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[mf][ct] : ah[mf][ct], term == 1 ? blv : bhv, acc[ni][ct], 0, 0, 0);
            // half-chunk u = 2 p + (0: gen_a, 1: gen_b); this MFMA's share: [u0, u1)
            constexpr int U6[7] = {0, 1, 3, 4, 5, 7, 8}, U3[4] = {0, 3, 6, 8};
            const int u0 = CT == 1 ? U3[q] : U6[q], u1 = CT == 1 ? U3[q + 1] : U6[q + 1];
#pragma unroll
            for (int u = u0; u < u1; ++u) {
              if ((u & 1) == 0) gen_a(kkn, mf, u >> 1, ev[mf], nin, kv[u >> 1], bhn[mf]);
              else gen_b(kv[u >> 1], u >> 1, bhn[mf][u >> 1], bln[mf]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        bh[0] = bhn[0]; bh[1] = bhn[1]; bl[0] = bln[0]; bl[1] = bln[1];
      }
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) aqc[kh] = aqn[kh];
      if constexpr (PF) {
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) { ah[mf][ct] = ahn[mf][ct]; al[mf][ct] = aln[mf][ct]; }
      } else if constexpr (!LEAN) {
        load_a((jb + 32) & (BN - 1), ah, al);
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
