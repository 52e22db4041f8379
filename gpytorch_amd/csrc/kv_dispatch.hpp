// Per-covariance-family kernel lookup (one translation unit per family so the build parallelises).
#pragma once
#include "kv_mfma.hpp"

namespace gpamd {

enum { KV_MODE_MFMA = 0, KV_MODE_GRAM = 1, KV_MODE_VALU = 2, KV_MODE_GRAMV = 3, KV_MODE_GRAM4 = 4, KV_MODE_GRAMH = 5, KV_MODE_DIRECTH = 6 };

// MFMA / Gram variants: CT 32-column tiles (1..4) + EX extra VALU column; NI row tiles per wave by CT.
// NI*CT*16 accumulator registers: 64 (CT <= 2) .. 128 (CT = 4)
// (beyond 16 dimensions a wave's own points -- NI rows of D coordinates or KH split operands each -- leave no room for four row tiles: two everywhere)
constexpr int KV_MAX_DIM = 32;          // fused float32 kernels exist for 1 .. 32 input dimensions (ABI version 4; 16 before)
constexpr int kv_ni_for_ct(int ct, int dk = 16) { return (ct == 1 && dk <= 16) ? 4 : 2; }
inline int kv_bm_for_ct(int ct, int dk = 16) { return 4 * kv_ni_for_ct(ct, dk) * 32; }
// kernels exist for these valid-dimension counts; other d use the next one (the extra coordinates are the zeros written by prep_points)
constexpr int kv_kernel_dims(int d) {
  return d <= 6 ? d : (d <= 8 ? 8 : (d <= 10 ? 10 : (d <= 12 ? 12 : (d <= 16 ? 16 : (d <= 20 ? 20 : (d <= 24 ? 24 : 32))))));
}

const void* kv_kernel_ptr_rbf(int mode, int d, int v, int ex);
const void* kv_kernel_ptr_matern12(int mode, int d, int v, int ex);
const void* kv_kernel_ptr_matern32(int mode, int d, int v, int ex);
const void* kv_kernel_ptr_matern52(int mode, int d, int v, int ex);
const void* kv_kernel_ptr_rq(int mode, int d, int v, int ex);

// small-t Gram-form kernels (kvs_<family>.hip; none for Matern nu = 1/2)
const void* kvs_kernel_ptr_rbf(int d, int tpad);
const void* kvs_kernel_ptr_matern32(int d, int tpad);
const void* kvs_kernel_ptr_matern52(int d, int tpad);
const void* kvs_kernel_ptr_rq(int d, int tpad);

// 3..32-column Gram-form kernels, contraction in column groups of four on v_mfma_f32_4x4x1 (kvm_<family>.hip)
const void* kvm_kernel_ptr_rbf(int d, int groups);
const void* kvm_kernel_ptr_matern32(int d, int groups);
const void* kvm_kernel_ptr_matern52(int d, int groups);
const void* kvm_kernel_ptr_rq(int d, int groups);

// direct differences + split contraction (kvd_<family>.hip, kv_directh.hpp): d in {1,2,3,4,5,6,8,10}, ni = 1, 2 row tiles per wave, ct = 1, 2 column tiles, ex
const void* kvd_kernel_ptr_rbf(int d, int ni, int ct, int ex);
const void* kvd_kernel_ptr_matern12(int d, int ni, int ct, int ex);
const void* kvd_kernel_ptr_matern32(int d, int ni, int ct, int ex);
const void* kvd_kernel_ptr_matern52(int d, int ni, int ct, int ex);
const void* kvd_kernel_ptr_rq(int d, int ni, int ct, int ex);

// split-operand kernels: generation and contraction on the f16 matrix pipe (kvh_<family>.hip); ct = 1, 2
const void* kvh_kernel_ptr_rbf(int d, int ct, int ex, int ni);
const void* kvh_kernel_ptr_matern32(int d, int ct, int ex, int ni);
const void* kvh_kernel_ptr_matern52(int d, int ct, int ex, int ni);
const void* kvh_kernel_ptr_rq(int d, int ct, int ex, int ni);

}  // namespace gpamd
