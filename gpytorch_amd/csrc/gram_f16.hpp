// Squared distances on the matrix pipe at the f16 rate, at f32 accuracy: hi/lo-split augmented coordinates.
//
// kv_gram.hpp's first form fed the quadratic expansion  S_ji = |z_j|^2 + |z_i|^2 - 2 z_j.z_i  to
// v_mfma_f32_32x32x2_f32: ceil((D+2)/2) instructions of 64 cycles per 32x32 block -- 9 % (D = 3) to 19 % (D = 10) of
// the matrix-pipe time of a 64-column product.  Here every f32 operand x is split exactly as  x = hi + lo,
// hi = f16(x), lo = f16(x - hi)  (22 significant bits), and the expansion becomes a sum of f16 x f16 products
//     z_j.(-2 z_i) = sum_q  ah bh + ah bl + al bh (+ al bl)          (a = z_j, b = -2 z_i)
//     |z_j|^2 * 1 + 1 * |z_i|^2  with both norms split the same way
// i.e. NS = NT*D + 4 "slots" (NT = 4 product terms per dimension, 3 for D >= 12 where al*bl is dropped), which ONE
// v_mfma_f32_32x32x16_f16 (16 slots, 32 cycles, products exact, f32 accumulate) evaluates for D <= 3, two for D <= 7,
// three for D <= 11: 1/6 to 1/4 of the f32 form's matrix-pipe time.  The C/D layout of the 32x32 MFMAs does not depend
// on the input type, so the result registers are still exactly this lane's B operands of the contraction steps.
//
// Accuracy (scripts/micro/gram_f16.hip, checked against float64 on the device): max |S - S_exact| = 6.8e-6 at
// max |z|^2 = 32 (relative error 4.7e-6 in K = 2^-S), 6e-8 absolute floor near S = 0 (f16 subnormal spacing of the split
// norms).  Worst case at the host's selection limit max |z|^2 = 32: 2^-22 (|z_i| + |z_j|)^2 ln 2 ~ 2e-5 relative in K
// (CPU emulation of this slot scheme: tests/test_gram_split_cpu.py).
#pragma once
#include "common.hpp"

namespace gpamd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int D>
struct GramF16 {
  static constexpr int NT = (D <= 11) ? 4 : 3;   // product terms per dimension
  static constexpr int NS = NT * D + 4;          // slots: products + |z_j|^2 (hi, lo) + |z_i|^2 (hi, lo)
  static constexpr int KH = (NS + 15) / 16;      // 32x32x16 MFMAs per 32x32 block
};

// hi + lo == v to 22 bits REQUIRES lo to be computed from the very hi that is stored.  The optimiser is free to
// re-derive "(_Float16)v" from a recomputed v (e.g. |z|^2 contracted differently in two places, 1 ulp apart): when v
// sits on an f16 rounding tie the two copies land on different neighbours and hi + lo is off by a whole f16 ulp
// (found by scripts/micro/pack_check.hip: one point in 1300, 2e-3 error in S).  The empty asm pins hi's float image.
__device__ __forceinline__ void f16_split(float v, _Float16& hi, _Float16& lo) {
  float hf = (float)(_Float16)v;
  asm volatile("" : "+v"(hf));
  hi = (_Float16)hf;  // exact: hf is an f16 value
  lo = (_Float16)(v - hf);
}

// (mfma_result_fence(): common.hpp)

// Block-centred expansion (GPAMD "recentre" mode).  The cancellation error of the quadratic expansion is ~2^-22 (|z_i| + |z_j|)^2: with
// the cloud centred as a whole it grows with the CLOUD radius (host limit max |z|^2 <= 32).  Squared distances are translation
// invariant, so a workgroup may subtract ANY common point c from both operands before the hi/lo split; with the rows sorted along a
// space-filling curve a block of output rows is compact, c = the centre of the block, |z_i - c| <= block radius, and
// |z_j - c| <= |z_j - z_i| + radius: the error is ~2^-22 (sqrt(S) + 2 radius)^2, i.e. small wherever the covariance is not (far pairs:
// k ~ 0 for RBF / Matern, relative error 2^-22 in S for the heavy-tailed RQ).  Xc: [ceil(n / 128)][DP] chunk centres of X1 or nullptr.
template <int DP>
__device__ __forceinline__ void load_center(const float* __restrict__ Xc, int row0, int rows, int n, float* cz) {
  // centre of the row block [row0, row0 + rows): the mean of the centres of its 128-row chunks (rows = 128, 256 or 512; chunk indices
  // are clamped to the last chunk, so a ragged last block still gets a point inside it).  The host evaluates the SAME formula when it
  // bounds the block radii (backend.SortedView).
#pragma unroll
  for (int k = 0; k < DP; ++k) cz[k] = 0.f;
  if (Xc) {
    const int last = (n - 1) >> 7, nch = rows >> 7;
    for (int c = 0; c < nch; ++c) {
      const int ch = min((row0 >> 7) + c, last);
#pragma unroll
      for (int q = 0; q < DP / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Xc + (int64_t)ch * DP + 4 * q);
        cz[4 * q + 0] += v[0]; cz[4 * q + 1] += v[1]; cz[4 * q + 2] += v[2]; cz[4 * q + 3] += v[3];
      }
    }
    const float inv = 1.0f / (float)nch;
#pragma unroll
    for (int k = 0; k < DP; ++k) cz[k] *= inv;
  }
}
template <int DP>
__device__ __forceinline__ void sub_center(float* z, const float* cz) {
#pragma unroll
  for (int k = 0; k < DP; ++k) z[k] -= cz[k];
}

// |z_j - c|^2 of a contracted point far from the row block (block-centred mode: up to the extent of the cloud) must stay inside the f16 range of
// its hi part.  It saturates at 60000: for an output row within the block policy (|z_i - c|^2 <= 32) the expansion then gives
// S >= 60000 - 2 sqrt(32) |z_j - c| >= 2000 as long as |z_j - c| <= 5000 (the host's extent limit, backend.GRAM_MAX_EXTENT_SQ) -- exp2(-2000) and
// exp(-sqrt(2000)) poly are zero in f32 like the true values, so the saturation is invisible for RBF / Matern; the heavy-tailed RQ keeps the unsaturated range.
__device__ __forceinline__ float gram_norm_clamp(float nn) { return __builtin_fminf(nn, 60000.f); }

// A side (contracted points x_j): slot s of a row with coordinates split into zh/zl and |z|^2 into nh/nl
template <int D>
__device__ __forceinline__ _Float16 gram_slot_a(int s, const _Float16* zh, const _Float16* zl, _Float16 nh, _Float16 nl,
                                                _Float16 valid) {
  constexpr int NT = GramF16<D>::NT;
  if (s < NT * D) return (s % NT) < 2 ? zh[s / NT] : zl[s / NT];
  const int u = s - NT * D;
  return u == 0 ? nh : (u == 1 ? nl : (u < 4 ? valid : (_Float16)0.f));
}

// B side (output points x_i): coordinates are those of -2 z_i
template <int D>
__device__ __forceinline__ _Float16 gram_slot_b(int s, const _Float16* bh, const _Float16* bl, _Float16 nh, _Float16 nl) {
  constexpr int NT = GramF16<D>::NT;
  if (s < NT * D) return ((s % NT) & 1) == 0 ? bh[s / NT] : bl[s / NT];
  const int u = s - NT * D;
  return u < 2 ? (_Float16)1.f : (u == 2 ? nh : (u == 3 ? nl : (_Float16)0.f));
}

// B operands of one output row for this lane's k-group h (elements 8h .. 8h+7 of every 16-slot group)
template <int D>
__device__ __forceinline__ void gram_pack_b(const float* z, int h, f16x8* out /* [KH] */, float nshift = 0.f) {
  constexpr int KH = GramF16<D>::KH;
  _Float16 bh[D], bl[D], nh, nl;
  float nn = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    nn = __builtin_fmaf(z[k], z[k], nn);
    f16_split(-2.f * z[k], bh[k], bl[k]);
  }
  f16_split(nn - nshift, nh, nl);   // nshift: S comes out as S - nshift (kv_gramh.hpp folds its 2^KSHIFT scale of K here)
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 v0 = gram_slot_b<D>(kh * 16 + e, bh, bl, nh, nl);
      const _Float16 v1 = gram_slot_b<D>(kh * 16 + 8 + e, bh, bl, nh, nl);
      out[kh][e] = h ? v1 : v0;
    }
}

// A operands of one contracted row for this lane's k-group h, in registers (kv_gram2.hpp: every wave splits its own rows)
template <int D>
__device__ __forceinline__ void gram_pack_a_lane(const float* z, bool valid, int h, f16x8* out /* [KH] */) {
  constexpr int KH = GramF16<D>::KH;
  _Float16 zh[D], zl[D], nh, nl;
  float nn = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    nn = __builtin_fmaf(z[k], z[k], nn);
    f16_split(z[k], zh[k], zl[k]);
  }
  f16_split(gram_norm_clamp(nn), nh, nl);
  const _Float16 one = valid ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 v0 = gram_slot_a<D>(kh * 16 + e, zh, zl, nh, nl, one);
      const _Float16 v1 = gram_slot_a<D>(kh * 16 + 8 + e, zh, zl, nh, nl, one);
      out[kh][e] = h ? v1 : v0;
    }
}

// A operands of one contracted row in LDS: planes Xh[kh][k-half][row][8] -- the 16-byte operand of lane (row, h) of k-step kh sits at
// gram_a_off(kh, row, h, BN).  Consecutive rows are 16 bytes apart, so the staging stores of consecutive threads AND the operand reads of
// consecutive lanes are both contiguous (conflict-free ds_write_b128 / ds_read_b128).  Rounds 1-3 used [kh][row][16] (row stride 32 bytes): the
// reads were contiguous over the wave, but the two 16-byte stores of a staging thread hit every other 16-byte slot -- 2-way bank conflicts that
// showed as 21 % / 17 % of the LDS cycles of the few-column kernels (profiles/r04_r4s11_kv_pmc_split_t1.json), whose tiles carry little else.
#ifdef GPAMD_XH_OLD_LAYOUT   // A/B builds only (scripts/kv_layout_ab.py): the [kh][row][16] image of rounds 1-3
__device__ __forceinline__ constexpr int gram_a_off(int kh, int row, int h, int BN) { return (kh * BN + row) * 16 + 8 * h; }
#else
__device__ __forceinline__ constexpr int gram_a_off(int kh, int row, int h, int BN) { return ((kh * 2 + h) * BN + row) * 8; }
#endif

template <int D>
__device__ __forceinline__ void gram_pack_a(const float* z, bool valid, _Float16* Xh, int row, int BN) {
  constexpr int KH = GramF16<D>::KH;
  _Float16 zh[D], zl[D], nh, nl;
  float nn = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    nn = __builtin_fmaf(z[k], z[k], nn);
    f16_split(z[k], zh[k], zl[k]);
  }
  f16_split(gram_norm_clamp(nn), nh, nl);
  const _Float16 one = valid ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gram_slot_a<D>(kh * 16 + 8 * hh + e, zh, zl, nh, nl, one);
      *reinterpret_cast<f16x8*>(&Xh[gram_a_off(kh, row, hh, BN)]) = v;
    }
}

}  // namespace gpamd
