// libgpamd_tune.so (NOT part of the product library; `make tune`).  Tuning-only entry point: launches a numbered structural variant of the fused K*V kernel
// (RBF, d <= 3 padded to 4, t = 32*CT + 1).  See kv_mfma_tune.hpp and scripts/kv_tune.py.
#include "../../../include/gpamd.h"

#include <hip/hip_runtime.h>

#include "kv_mfma_tune.hpp"
#include "../kv_gram.hpp"

using namespace gpamd;

namespace {
struct VariantDesc {
  int ni, bn;
  const char* name;
};
//                         CT NI EX BN  PIPE PRIO GRAM MINW STG
#define VARIANTS(X)                       \
  X(0, 2, 2, 1, 128, 0, 0, 0, 1, 0)          \
  X(1, 2, 2, 1, 128, 1, 0, 0, 1, 0)          \
  X(2, 2, 2, 1, 128, 0, 1, 0, 1, 0)          \
  X(3, 2, 2, 1, 128, 0, 0, 1, 1, 0)          \
  X(4, 2, 2, 1, 256, 0, 0, 0, 1, 0)          \
  X(5, 2, 4, 1, 128, 0, 0, 0, 1, 0)          \
  X(6, 2, 4, 1, 128, 1, 0, 0, 1, 0)          \
  X(7, 2, 2, 1, 128, 1, 0, 1, 1, 0)          \
  X(8, 2, 4, 1, 128, 1, 0, 1, 1, 0)          \
  X(9, 2, 2, 1, 128, 0, 0, 0, 2, 0)          \
  X(10, 2, 4, 1, 256, 0, 0, 1, 1, 0)         \
  X(11, 2, 2, 1, 128, 1, 1, 1, 1, 0)         \
  X(12, 2, 2, 1, 128, 0, 0, 1, 2, 0)         \
  X(13, 2, 3, 1, 128, 0, 0, 1, 1, 0)     \
  X(14, 2, 2, 1, 128, 0, 0, 0, 1, 1)     \
  X(15, 2, 2, 1, 128, 0, 0, 1, 1, 1)     \
  X(16, 2, 2, 1, 128, 1, 0, 1, 1, 1)     \
  X(17, 2, 2, 1, 256, 0, 0, 1, 1, 1)     \
  X(18, 2, 3, 1, 128, 0, 0, 1, 1, 1)     \
  X(19, 2, 2, 1, 128, 0, 0, 2, 1, 0)     \
  X(20, 2, 2, 1, 128, 0, 0, 3, 1, 0)     \
  X(21, 2, 2, 1, 128, 0, 0, 2, 1, 1)     \
  X(22, 2, 2, 1, 128, 0, 0, 3, 1, 1)     \
  X(23, 2, 2, 1, 128, 1, 1, 0, 1, 1)     \
  X(24, 2, 2, 1, 128, 1, 0, 0, 1, 1)     \
  X(25, 2, 4, 1, 128, 0, 0, 3, 1, 1)
}  // namespace

extern "C" {

int gpamd_kv_variant_count(void) { return 27; }

int gpamd_kv_variant_info(int variant, int* bm_host, int* bn_host) {
  switch (variant) {
#define X(ID, CT, NI, EX, BN, PIPE, PRIO, GRAM, MINW, STG) \
  case ID:                                            \
    *bm_host = 4 * NI * 32;                           \
    *bn_host = BN;                                    \
    return 0;
    VARIANTS(X)
#undef X
    case 26:
      *bm_host = 256;
      *bn_host = 128;
      return 0;
  }
  return GPAMD_EINVAL;
}

int gpamd_kv_partials_variant_f32(int variant, const float* X1p, int n, const float* X2p, int m, const float* Vt,
                                  int64_t ldv, int t, float* P, int64_t ldo, int S, int jchunk, void* stream) {
  if (t != 65) return GPAMD_EUNSUPPORTED;
  KvArgs a;
  a.X1 = X1p; a.X2 = X2p; a.Vt = Vt; a.P = P;
  a.ldv = ldv; a.ldo = ldo; a.pstride = (int64_t)t * ldo;
  a.n = n; a.m = m; a.t = t; a.S = S; a.jchunk = jchunk; a.done = nullptr;
  switch (variant) {
#define X(ID, CT, NI, EX, BN, PIPE, PRIO, GRAM, MINW, STG)                                                        \
  case ID: {                                                                                                      \
    a.nrb = (n + 4 * NI * 32 - 1) / (4 * NI * 32);                                                                \
    hipLaunchKernelGGL((kv_mfma_tune_kernel<CT, NI, EX, BN, PIPE, PRIO, GRAM, MINW, STG>), dim3((unsigned)a.nrb * S), \
                       dim3(256), 0, (hipStream_t)stream, a);                                                     \
    break;                                                                                                        \
  }
    VARIANTS(X)
#undef X
    case 26:  // Gram-form generation (product kernel), two distance tiles live
      a.nrb = (n + 255) / 256;
      hipLaunchKernelGGL((kv_gram_kernel<KIND_RBF, 3, 2, 2, 1>), dim3((unsigned)a.nrb * S), dim3(256), 0, (hipStream_t)stream, a);
      break;
    default:
      return GPAMD_EINVAL;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
