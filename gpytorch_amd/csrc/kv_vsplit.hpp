// Pre-pass of the split-operand product (kv_gramh.hpp): per-column power-of-two scale and the two f16 planes of V in the
// k-slot order of the MFMA A operand.  HBM-bound: reads n t floats twice, writes n t floats once.  Included by api.hip only.
#pragma once
#include "kv_gramh.hpp"

namespace gpamd {

// ---- pre-pass: column maxima and the split planes ----
// colmax[c] = bits of max_j |V[c][j]|  (non-negative floats order like unsigned integers); zeroed by the caller
__global__ __launch_bounds__(256) void vsplit_colmax_kernel(const float* Vt, int64_t ldv, int m, unsigned* colmax, const int* done) {
  if (done && *done) return;
  const int c = blockIdx.y;
  const float* src = Vt + (int64_t)c * ldv;
  float mx = 0.f;
  for (int j = 4 * (blockIdx.x * 256 + threadIdx.x); j < m; j += 4 * 256 * gridDim.x) {
    if (j + 4 <= m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + j);
      mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    } else {
      for (int e = 0; j + e < m; ++e) mx = fmaxf(mx, fabsf(src[j + e]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&colmax[c], __float_as_uint(mx));
}

// planes[c][16 g + 8 h + e] = split(scale_c * V[c][16 g + (e&3) + 8 (e>>2) + 4 h]); rows c >= t and positions >= m are zero.
// One thread per 8 positions (one 16-byte store per plane); grid.y = 32 CT rows, colmul has one more entry for the extra column.
__global__ __launch_bounds__(256) void vsplit_kernel(const float* Vt, int64_t ldv, int m, int t, _Float16* Vh, _Float16* Vl,
                                                     int64_t ldh, const unsigned* colmax, float* colmul, const int* done) {
  if (done && *done) return;
  const int c = blockIdx.y;
  const int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x;   // chunk of 8 positions
  if (8 * ch >= ldh) return;
  float scale = 1.0f;
  if (c < t) {
    const float mx = __uint_as_float(colmax[c]);
    if (mx > 0.f && mx < 3.0e38f) {
      int ex;
      (void)frexpf(mx, &ex);                    // mx = f 2^ex, f in [0.5, 1)
      int sh = KGH_VEXP - ex;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      scale = ldexpf(1.0f, sh);
    }
    if (ch == 0) colmul[c] = ldexpf(1.0f, -KGH_KSHIFT) / scale;
  }
  if (ch == 0 && c == 0) colmul[gridDim.y] = ldexpf(1.0f, -KGH_KSHIFT);   // the extra (f32, unscaled) column
  const int g = (int)(ch >> 1), hh = (int)(ch & 1);
  const int ja = 16 * g + 4 * hh, jb = ja + 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = (e < 4 ? ja : jb) + (e & 3);
    v[e] = (c < t && j < m) ? Vt[(int64_t)c * ldv + j] * scale : 0.f;
  }
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 a, b;
    f16_split(v[e], a, b);
    hi[e] = a;
    lo[e] = b;
  }
  *reinterpret_cast<f16x8*>(Vh + (int64_t)c * ldh + 8 * ch) = hi;
  *reinterpret_cast<f16x8*>(Vl + (int64_t)c * ldh + 8 * ch) = lo;
}


}  // namespace gpamd
