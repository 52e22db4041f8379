// libgpamd_tune.so: ablation builds of the split-operand kernel (kv_gramh.hpp, template parameter ABL) -- which part of the
// loop bounds it?  RBF, d = 3, 64 columns, planes supplied by the caller (any finite content: the timing does not depend on it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kv_gramh.hpp"

using namespace gpamd;

extern "C" int gpamd_tune_kv_gramh_rbf3(int abl, int ni, int ex, const float* X1p, int n, const float* X2p, int m, const float* Vt, int64_t ldv,
                                        const void* Vh, const void* Vl, int64_t ldh, const float* colmul, float* P, int64_t ldo, int S,
                                        int jchunk, void* stream) {
  KvhArgs ka;
  KvArgs& a = ka.a;
  a.X1 = X1p; a.X2 = X2p; a.Vt = Vt; a.P = P;
  a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)(64 + ex) * ldo;
  a.n = n; a.m = m; a.t = 64 + ex; a.S = S; a.jchunk = jchunk; a.done = nullptr; a.kparam = 0.f;
  a.nrb = (n + 128 * ni - 1) / (128 * ni);
  ka.Vh = (const _Float16*)Vh; ka.Vl = (const _Float16*)Vl; ka.ldh = ldh; ka.colmul = colmul;
  const dim3 grid((unsigned)a.nrb * S), block(256);
#define L(A, N) if (abl == A && ni == N && !ex) hipLaunchKernelGGL((kv_gramh_kernel<KIND_RBF, 3, 2, N, 0, A>), grid, block, 0, (hipStream_t)stream, ka);
#define LX(A, N) if (abl == A && ni == N && ex) hipLaunchKernelGGL((kv_gramh_kernel<KIND_RBF, 3, 2, N, 1, A>), grid, block, 0, (hipStream_t)stream, ka);
  L(0, 2) L(1, 2) L(3, 2) L(4, 2) L(6, 2) L(7, 2) L(8, 2) L(10, 2) L(0, 4) LX(0, 2) LX(0, 4)
#undef L
#undef LX
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
