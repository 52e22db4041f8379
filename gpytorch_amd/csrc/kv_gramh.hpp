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
// waves per SIMD, eight waves per workgroup, register prefetch of the next tile, a deeper Gram look-ahead -- are `if constexpr` branches of THIS
// body (kv_gramh_body.inc, constants ABL / NW / OCC; the product sets ABL = 0, NW = 4, and its ISA is byte for byte what it was before the
// branches moved in: round 6, diff of the -S output of kvh_*.hip).  tune/kv_gramh_ablate.hpp includes the same body for libgpamd_tune.so, so the
// two cannot drift; DESIGN.md 3.1b has the numbers.)
// ABL: 0 the product; 1 no generation VALU; 2 no contraction MFMAs (the generated operands XOR-folded into the output: a live sink); 3 V planes /
// x rows staged once (no global loads / LDS writes per tile; barriers stay); 4 as 3 and no barriers; 5 A operands of block 0 for every block;
// 6 register prefetch of the next tile's planes; 7 one wave per SIMD (the wrapper's occupancy attribute); 8 no sched_barrier pinning; 10 Gram MFMA
// one step further ahead (NI = 2).  NW: waves per workgroup.  OCC: resident waves per SIMD the wrapper holds the allocator to.
// (beyond 20 dimensions: KH >= 5 Gram MFMAs per block -- the split x_i / x_j operands no longer fit 256 registers next to 64 accumulators, and the two
// Xh buffers push the LDS image past half a CU: ONE wave per SIMD with the whole register file instead of 46 .. 94 spilled registers)
template <int D>
constexpr int kgh_waves() { return D > 20 ? 1 : 2; }
template <int KIND, int D, int CT, int NI, int EX, int SAFE = 0>   // SAFE = 1 / 2: tune library only (tune/tune_hazard.hip): every Gram result behind the full mfma_result_fence / round 5's form
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kgh_waves<D>(), kgh_waves<D>())))
void kv_gramh_kernel(KvhArgs ka) {
  constexpr int ABL = 0, NW = 4, OCC = kgh_waves<D>();   // the product: no ablation, four waves per workgroup (tune/kv_gramh_ablate.hpp sets the others)
#include "kv_gramh_body.inc"
}

}  // namespace gpamd
