// Per-covariance-family launchers (one translation unit per family so the build parallelises).
#pragma once
#include "kv_mfma.hpp"

namespace gpamd {

// MFMA variants: CT 32-column tiles (1..4) + EX extra VALU column; NI row tiles per wave by CT.
// NI*CT*16 accumulator registers: 64 (CT <= 2: three waves resident per SIMD) .. 128 (CT = 4: two)
constexpr int kv_ni_for_ct(int ct) { return ct == 1 ? 4 : 2; }
inline int kv_bm_for_ct(int ct) { return 4 * kv_ni_for_ct(ct) * 32; }

#define GPAMD_DECL_FAMILY(NAME)                                                                          \
  int launch_kv_mfma_##NAME(int dp, int ct, int ex, const KvArgs& a, unsigned grid, hipStream_t stream); \
  int launch_kv_gram_##NAME(int dp, int ct, int ex, const KvArgs& a, unsigned grid, hipStream_t stream); \
  int launch_kv_valu_##NAME(int dp, int tpad, const KvArgs& a, unsigned grid, hipStream_t stream);
GPAMD_DECL_FAMILY(rbf)
GPAMD_DECL_FAMILY(matern12)
GPAMD_DECL_FAMILY(matern32)
GPAMD_DECL_FAMILY(matern52)
#undef GPAMD_DECL_FAMILY

}  // namespace gpamd
