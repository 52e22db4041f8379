// Far-pair tile culling: the list kernel (settings.far_pair_cutoff / gpamd_kv_partials_far_f32, gpamd_kv_grad2_far_f32, gpamd_kv_grad_far_f32).
// Included by the translation units that hold those entry points (a template, so that each may instantiate it).
#pragma once
#include "common.hpp"

namespace gpamd {

// Tile lists for far-pair culling: one wave per unit (s, rb).  (bn = 64: the derivative kernels' j step -- every step takes the sphere of the
// 128-point tile it lies in.)  Row block rb = rows [rb * bm, + bm) of X1 (bm a multiple of 128): its sphere has
// centre = the mean of its 128-row chunk centres (what load_center forms; chunk indices clamped to the last chunk) and radius
// max_q |c_q - centre| + r_q.  A tile of bn contracted points (bn = 128 or 256: one or two 128-point spheres) survives when any of its spheres
// comes within sqrt(sq_cut) of the row block's; 64 tiles per step, a ballot compacts the survivors in order.
struct CullArgs {
  const float* rc; const float* rr;   // [ceil(n / 128)][dp], [ceil(n / 128)]: chunk spheres of X1
  const float* tc; const float* tr;   // [ceil(m / 128)][dp], [ceil(m / 128)]: tile spheres of X2
  int* tiles; int tpc1;
  int n, m, dp, bm, bn, nrb, jchunk;
  float sq_cut;
  const int* done;
};
constexpr int CULL_MAX_DP = 32;
template <int UNUSED>
__global__ __launch_bounds__(64) void cull_list_kernel(CullArgs a) {
  if (a.done && *a.done) return;
  const int unit = blockIdx.x, lane = threadIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk, jend = min(a.m, jbeg + a.jchunk);
  float c[CULL_MAX_DP];
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < CULL_MAX_DP; ++k) c[k] = 0.f;
  const int last = (a.n - 1) >> 7, nch = max(a.bm >> 7, 1), ch0 = (rb * a.bm) >> 7;
  for (int q = 0; q < nch; ++q) {
    const int ch = min(ch0 + q, last);
#pragma unroll
    for (int k = 0; k < CULL_MAX_DP; ++k)
      if (k < a.dp) c[k] += a.rc[(int64_t)ch * a.dp + k];
  }
  const float inv = 1.0f / (float)nch;
#pragma unroll
  for (int k = 0; k < CULL_MAX_DP; ++k) c[k] *= inv;
  for (int q = 0; q < nch; ++q) {
    const int ch = min(ch0 + q, last);
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < CULL_MAX_DP; ++k)
      if (k < a.dp) {
        const float d = a.rc[(int64_t)ch * a.dp + k] - c[k];
        d2 = __builtin_fmaf(d, d, d2);
      }
    r = __builtin_fmaxf(r, __builtin_sqrtf(d2) * 1.000001f + a.rr[ch]);
  }
  int* out = a.tiles + (int64_t)unit * a.tpc1;
  const int nc = a.bn >= 128 ? a.bn >> 7 : 1, lastc = (a.m - 1) >> 7;
  int cnt = 0;
  for (int j0 = jbeg; j0 < jend; j0 += 64 * a.bn) {
    const int j = j0 + lane * a.bn;
    bool keep = false;
    if (j < jend) {
      for (int q = 0; q < nc; ++q) {
        const int ch = min((j >> 7) + q, lastc);
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < CULL_MAX_DP; ++k)
          if (k < a.dp) {
            const float d = a.tc[(int64_t)ch * a.dp + k] - c[k];
            d2 = __builtin_fmaf(d, d, d2);
          }
        const float gap = __builtin_sqrtf(d2) * 0.999999f - r - a.tr[ch];
        keep = keep || !(gap > 0.f && gap * gap > a.sq_cut);   // (NaN spheres: kept)
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) out[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = j;
    cnt += __popcll(mask);
  }
  if (lane == 0) out[cnt] = jbeg + a.jchunk;   // terminator: >= jend, and small enough that the look-ahead staging's j0 + tid cannot overflow
}

}  // namespace gpamd
