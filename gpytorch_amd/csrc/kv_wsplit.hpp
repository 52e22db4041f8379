// Pre-pass of the split-operand bilinear derivative (kv_grad2.hpp, WSPLIT): the left / right vector blocks L [t][n], R [t][m] as
// hi/lo f16 planes in ROW-major order [row][CP] (CP = 80 column slots), so that a lane's MFMA operand -- eight consecutive columns of
// one row -- is one 16-byte load.  W = sum_c L_c R_c^T mixes the columns, so the per-column power-of-two scales a_c (L), b_c (R)
// must have a CONSTANT product 2^K: K is set by the column with the largest max|L_c| max|R_c| (both its operands land at ~2^12), every
// other column is scaled DOWN by the ratio, half of it on each side -- columns that contribute nothing underflow harmlessly.
// HBM-bound: reads (n + m) t floats twice, writes (n + m) 80 f16 pairs.
#pragma once
#include "gram_f16.hpp"

namespace gpamd {

constexpr int WS_CP = 80;      // column slots per plane row (five MFMA k-steps of 16)
constexpr int WS_TARGET = 12;  // operands of the dominant column are scaled to max ~2^12

// Row-sign randomisation (round 4).  The f16 matrix pipe does NOT round its f32 accumulation to nearest: aligning the addends of a k-step
// against the largest one drops their low bits by truncation in two's complement, i.e. toward -infinity, whatever the sign of the value --
// every W_ij comes out as W_ij - b_ij with b_ij ~ 1e-8 |W_ij| >= 0 (measured: the same 0.9 .. 1.5e-8 of sum |W dK| on every kind of signed
// vectors, at every n; absent from the fp32 MFMA, absent from a float64 emulation of the split arithmetic with round-to-nearest, and
// reproduced by emulating round-down: tests/test_split_contraction_cpu.py).  In K*V products that bias is 1e-8 of |K| |V| and harmless.  In the
// bilinear derivative n^2 terms of random sign sum to an O(n) result: the bias does not cancel while the sum does, so the RELATIVE error of
// the gradient grew like n -- 2e-2 of the data-fit term at n = 100 000 (profiles/r04_s5_*).  Cure at no cost in the main loop: row i of the
// left block enters the planes multiplied by a pseudo-random sign s_i; the kernel accumulates s_i W_ij - b_ij and every lane, which owns one
// row i for the whole kernel, multiplies its partial sums by s_i at the very end.  The bias becomes -s_i b_ij: a random walk over the n rows
// instead of a coherent sum, 1 / sqrt(n) of its former size (measured after the change: as accurate as the fp32 contraction).
__host__ __device__ __forceinline__ float ws_rowsign(int i) {
  unsigned h = (unsigned)i * 2654435761u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return (h & 0x10000u) ? -1.0f : 1.0f;
}
// ... and 32-row block J of the RIGHT block by an independent sign s_J (a second hash stream): the bias becomes -s_i s_J b_ij, a random walk over
// n * n / 32 (row, block) pairs.  The kernel un-flips per half step (its two 32-row j blocks have wave-uniform signs) and, for the per-dimension
// sums, by staging the [1 | z_j | z_j^2] columns of the A-contraction with the sign already applied.
__host__ __device__ __forceinline__ float ws_blocksign(int J) { return ws_rowsign((int)(0x40000000u | (unsigned)J)); }

// colmax[c] = bits of max_j |V[c][j]| (non-negative floats order like unsigned integers); zeroed by the caller.  (Templates: the header is
// included by more than one translation unit.)
template <int UNUSED>
__global__ __launch_bounds__(256) void wsplit_colmax_kernel(const float* __restrict__ Vt, int64_t ldv, int m, unsigned* __restrict__ colmax) {
  const int c = blockIdx.y;
  const float* src = Vt + (int64_t)c * ldv;
  float mx = 0.f;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += 256 * gridDim.x) mx = fmaxf(mx, fabsf(src[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&colmax[c], __float_as_uint(mx));
}

// scales[0..CP): a_c, scales[CP..2CP): b_c, scales[2CP] = 2^-K (the factor that restores W).  One block of 128 threads.
template <int UNUSED>
__global__ __launch_bounds__(128) void wsplit_scales_kernel(const unsigned* __restrict__ maxL, const unsigned* __restrict__ maxR, int t,
                                                            float* __restrict__ scales) {
  __shared__ int esum[WS_CP];
  __shared__ int emax;
  const int c = threadIdx.x;
  int eL = -200, eR = -200;
  if (c < t) {
    const float l = __uint_as_float(maxL[c]), r = __uint_as_float(maxR[c]);
    if (l > 0.f && l < 3.0e38f) (void)frexpf(l, &eL);
    if (r > 0.f && r < 3.0e38f) (void)frexpf(r, &eR);
  }
  if (c < WS_CP) esum[c] = (c < t && eL > -200 && eR > -200) ? eL + eR : -100000;
  __syncthreads();
  if (c == 0) {
    int mx = -100000;
    for (int q = 0; q < WS_CP; ++q) mx = max(mx, esum[q]);
    emax = mx;
  }
  __syncthreads();
  if (c < WS_CP) {
    float a = 0.f, b = 0.f;
    if (esum[c] > -100000) {
      const int delta = emax - esum[c];                    // >= 0
      int sa = WS_TARGET - eL - delta / 2, sb = WS_TARGET - eR - (delta - delta / 2);
      sa = sa > 120 ? 120 : (sa < -120 ? -120 : sa);
      sb = sb > 120 ? 120 : (sb < -120 ? -120 : sb);
      a = ldexpf(1.0f, sa);
      b = ldexpf(1.0f, sb);
    }
    scales[c] = a;
    scales[WS_CP + c] = b;
  }
  if (c == 0) scales[2 * WS_CP] = emax > -100000 ? ldexpf(1.0f, -(2 * WS_TARGET - emax)) : 0.f;
}

// planes[row][c] = split(scale_c * Vt[c][row]) for row < rows, c < t; zero elsewhere (rows up to rows_pad, columns up to CP).
// One thread per (row, chunk of 8 columns): consecutive threads take consecutive rows -> every one of the 8 loads is coalesced.
template <int UNUSED>
__global__ __launch_bounds__(256) void wsplit_planes_kernel(const float* __restrict__ Vt, int64_t ldv, int rows, int rows_pad, int t,
                                                            const float* __restrict__ scale, _Float16* __restrict__ Ph,
                                                            _Float16* __restrict__ Pl, int flip) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int ck = blockIdx.y;   // chunk of 8 columns
  if (row >= rows_pad) return;
  const float sgn = flip == 1 ? ws_rowsign(row) : (flip == 2 ? ws_blocksign(row >> 5) : 1.0f);   // left block: row signs; right block: 32-row block signs
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * ck + e;
    float v = 0.f;
    if (row < rows && c < t) v = Vt[(int64_t)c * ldv + row] * scale[c] * sgn;
    _Float16 x, y;
    f16_split(v, x, y);
    hi[e] = x;
    lo[e] = y;
  }
  *reinterpret_cast<f16x8*>(Ph + (int64_t)row * WS_CP + 8 * ck) = hi;
  *reinterpret_cast<f16x8*>(Pl + (int64_t)row * WS_CP + 8 * ck) = lo;
}

}  // namespace gpamd
