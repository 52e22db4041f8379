// libgpamd_tune.so: hazard stress builds of the two kernels that carry > 95 % of all time.  The product kernels (kv_gram.hpp, kv_gramh.hpp) read the
// VGPR results of their Gram MFMAs on the VALU at a distance that is the toolchain's table plus one intervening MFMA (kv_gramh) / 20 explicit wait
// states (kv_gram); the SAFE = 1 instantiations of the SAME templates put the full data-dependent fence (32 wait states, common.hpp) behind every Gram
// MFMA.  tests/test_gpu_hazard_stress.py launches both on one full chip, many times, and compares the partial slabs BITWISE: a stale read would differ
// from run to run (that is how the kv_gramv hazard showed).  The whole library is re-instantiated in namespace gpamd_hz so that nothing
// here can collide with (or be bound to) the product library's symbols.
#define gpamd gpamd_hz
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../kv_gram.hpp"
#include "../kv_gramh.hpp"

using namespace gpamd_hz;

// which: 0 = kv_gram_kernel (fp32 contraction; Vh / Vl / colmul unused), 1 = kv_gramh_kernel (split contraction)
// variant: index into the lists below;  safe: 0 = the product's code path, 1 = fully fenced, 2 = the toolchain's table only (kv_gram: round 4's form, kv_gramh: round 5's); kv_gram only: 3 = the
// product's wait states without the second scheduling barrier (timing A/B, scripts/kv_gram_fence_ab.py)
extern "C" int gpamd_tune_hazard_launch(int which, int variant, int safe, const float* X1p, int n, const float* X2p, int m, const float* Vt, int64_t ldv, int t,
                                        const void* Vh, const void* Vl, int64_t ldh, const float* colmul, float* P, int64_t ldo, int S, int jchunk, void* stream) {
  KvhArgs ka;
  KvArgs& a = ka.a;
  a.X1 = X1p; a.X2 = X2p; a.Vt = Vt; a.P = P;
  a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)t * ldo;
  a.n = n; a.m = m; a.t = t; a.S = S; a.jchunk = jchunk; a.done = nullptr; a.kparam = 0.f; a.Xc = nullptr;
  ka.Vh = (const _Float16*)Vh; ka.Vl = (const _Float16*)Vl; ka.ldh = ldh; ka.colmul = colmul;
  hipStream_t st = (hipStream_t)stream;
  int ni = 0;
#define GRAM(V, K, D, CT, NI, EX)                                                                                             \
  if (which == 0 && variant == V) {                                                                                            \
    ni = NI; a.nrb = (n + 128 * NI - 1) / (128 * NI);                                                                          \
    if (safe == 1) hipLaunchKernelGGL((kv_gram_kernel<K, D, CT, NI, EX, 1>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, a);  \
    else if (safe == 2) hipLaunchKernelGGL((kv_gram_kernel<K, D, CT, NI, EX, 2>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, a); \
    else if (safe == 3) hipLaunchKernelGGL((kv_gram_kernel<K, D, CT, NI, EX, 3>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((kv_gram_kernel<K, D, CT, NI, EX, 0>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, a);            \
  }
#define GRAMH(V, K, D, CT, NI, EX)                                                                                            \
  if (which == 1 && variant == V) {                                                                                            \
    ni = NI; a.nrb = (n + 128 * NI - 1) / (128 * NI);                                                                          \
    if (safe == 1) hipLaunchKernelGGL((kv_gramh_kernel<K, D, CT, NI, EX, 1>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, ka);     \
    else if (safe == 2) hipLaunchKernelGGL((kv_gramh_kernel<K, D, CT, NI, EX, 2>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, ka); \
    else hipLaunchKernelGGL((kv_gramh_kernel<K, D, CT, NI, EX, 0>), dim3((unsigned)a.nrb * S), dim3(256), 0, st, ka);          \
  }
  // the instantiations the round-4 audit named (D = 1: ONE Gram MFMA per block, the shortest distance), the headline / default kernels, one Matern
  GRAM(0, KIND_RBF, 1, 1, 4, 0) GRAM(1, KIND_MATERN32, 1, 1, 4, 0) GRAM(2, KIND_RBF, 3, 2, 2, 1) GRAM(3, KIND_MATERN52, 10, 2, 2, 1)
  GRAMH(0, KIND_MATERN32, 1, 2, 2, 0) GRAMH(1, KIND_RBF, 1, 1, 4, 0) GRAMH(2, KIND_RBF, 3, 2, 4, 1) GRAMH(3, KIND_MATERN52, 10, 2, 2, 1)
#undef GRAM
#undef GRAMH
  if (!ni) return -2;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
