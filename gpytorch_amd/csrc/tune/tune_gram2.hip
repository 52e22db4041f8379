// libgpamd_tune.so: the software-pipelined, DMA-staged Gram kernel (kv_gram2.hpp) -- measured SLOWER than the product
// kernel (profiles/r01_s18_dma_pipelined_vs_sync.json) and therefore kept out of libgpamd.so; scripts/async_check.py
// A/Bs it against the product kernel through this entry point.  RBF, d <= 3, 33 <= t <= 65.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kv_gram2.hpp"

using namespace gpamd;

extern "C" int gpamd_tune_kv_gram2_rbf3(const float* X1p, int n, const float* X2p, int m, const float* Vt, int64_t ldv, int t,
                                        float* P, int64_t ldo, int S, int jchunk, void* stream) {
  if (t < 33 || t > 65) return 3;
  KvArgs a;
  a.X1 = X1p; a.X2 = X2p; a.Vt = Vt; a.P = P;
  a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)t * ldo;
  a.n = n; a.m = m; a.t = t; a.S = S; a.jchunk = jchunk; a.done = nullptr;
  a.nrb = (n + 255) / 256;
  const dim3 grid((unsigned)a.nrb * S), block(256);
  if (t == 65)
    hipLaunchKernelGGL((kv_gram2_kernel<KIND_RBF, 3, 2, 2, 1>), grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((kv_gram2_kernel<KIND_RBF, 3, 2, 2, 0>), grid, block, 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
