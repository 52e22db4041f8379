// libgpamd_tune.so: ablation builds of the split-operand kernel (kv_gramh.hpp, template parameter ABL) -- which part of the
// loop bounds it?  RBF, d = 3, 64 columns, planes supplied by the caller (any finite content: the timing does not depend on it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kv_gramh_ablate.hpp"

using namespace gpamd;

// abl >= 100: geometry variants of the product loop (ABL = 0): 100 + 10 * code, code 1 = eight waves per workgroup (NW = 8), 2 = three
// waves per SIMD with one row tile per wave (NI = 1, no operand look-ahead), 3 = both NW = 8 and four row tiles (LEAN)
extern "C" int gpamd_tune_kv_gramh_rbf3(int abl, int ni, int ex, const float* X1p, int n, const float* X2p, int m, const float* Vt, int64_t ldv,
                                        const void* Vh, const void* Vl, int64_t ldh, const float* colmul, float* P, int64_t ldo, int S,
                                        int jchunk, void* stream) {
  KvhArgs ka{};   // (no chunk centres, no tile lists)
  KvArgs& a = ka.a;
  a.X1 = X1p; a.X2 = X2p; a.Vt = Vt; a.P = P;
  a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)(64 + ex) * ldo;
  a.n = n; a.m = m; a.t = 64 + ex; a.S = S; a.jchunk = jchunk; a.done = nullptr; a.kparam = 0.f;
  const int nw = (abl == 110 || abl == 130) ? 8 : 4;
  a.nrb = (n + 32 * nw * ni - 1) / (32 * nw * ni);
  ka.Vh = (const _Float16*)Vh; ka.Vl = (const _Float16*)Vl; ka.ldh = ldh; ka.colmul = colmul;
  const dim3 grid((unsigned)a.nrb * S), block(64 * nw);
#define L(A, N) if (abl == A && ni == N && !ex) hipLaunchKernelGGL((kv_gramh_ablate_kernel<KIND_RBF, 3, 2, N, 0, A>), grid, block, 0, (hipStream_t)stream, ka);
#define LX(A, N) if (abl == A && ni == N && ex) hipLaunchKernelGGL((kv_gramh_ablate_kernel<KIND_RBF, 3, 2, N, 1, A>), grid, block, 0, (hipStream_t)stream, ka);
  L(0, 2) L(1, 2) L(2, 2) L(3, 2) L(4, 2) L(5, 2) L(6, 2) L(7, 2) L(8, 2) L(10, 2) L(0, 4) LX(0, 2) LX(0, 4)
#define G(A, N, E, NWV, OCCV) if (abl == A && ni == N && ex == E) hipLaunchKernelGGL((kv_gramh_ablate_kernel<KIND_RBF, 3, 2, N, E, 0, NWV, OCCV>), grid, block, 0, (hipStream_t)stream, ka);
  G(110, 2, 0, 8, 2) G(110, 2, 1, 8, 2) G(120, 1, 0, 4, 3) G(120, 1, 1, 4, 3) G(130, 4, 0, 8, 2) G(130, 4, 1, 8, 2) G(110, 1, 0, 8, 2) G(110, 1, 1, 8, 2)
#undef G
#undef L
#undef LX
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
