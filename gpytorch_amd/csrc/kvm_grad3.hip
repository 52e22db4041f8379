// The split-operand instantiations of the fused bilinear-derivative kernel (kv_grad2.hpp, WSPLIT = 1) in a translation unit of their own
// (build parallelism); launched by kvm_grad2.hip.  A kvm_* unit: compiled with -mllvm -amdgpu-mfma-vgpr-form=1.
#include <hip/hip_runtime.h>

#include "kv_grad2.hpp"

namespace gpamd {

size_t grad2_split_lds(int kh, int d, int mode) {
  const int gz = (1 + 2 * d + 3) / 4;
  return (size_t)2 * G2_BN * G2_CPL * 2 + (size_t)kh * G2_BN * 16 * 2 + (mode ? (size_t)4 * gz * (G2_BN + 4) * 4 : 0);
}

namespace {
template <int KIND, int D>
int launch_d(int mode, const Grad2Args& a, unsigned grid, hipStream_t st) {
  const size_t lds = grad2_split_lds(GramF16<D>::KH, D, mode);
  if (mode == 0) {
    auto kfn = kv_grad2_kernel<KIND, D, 0, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, a);
  } else if constexpr (D > 16) {
    return -2;   // per-dimension sums / input gradients beyond 16 dimensions: the caller's row-block path (backend.kv_grad_generic)
  } else {
    auto kfn = kv_grad2_kernel<KIND, D, 1, 1>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, a);
  }
  return 0;
}

template <int KIND>
int launch_kind(int dk, int mode, const Grad2Args& a, unsigned grid, hipStream_t st) {
  switch (dk) {
    case 1: return launch_d<KIND, 1>(mode, a, grid, st);
    case 2: return launch_d<KIND, 2>(mode, a, grid, st);
    case 3: return launch_d<KIND, 3>(mode, a, grid, st);
    case 4: return launch_d<KIND, 4>(mode, a, grid, st);
    case 5: return launch_d<KIND, 5>(mode, a, grid, st);
    case 6: return launch_d<KIND, 6>(mode, a, grid, st);
    case 8: return launch_d<KIND, 8>(mode, a, grid, st);
    case 10: return launch_d<KIND, 10>(mode, a, grid, st);
    case 12: return launch_d<KIND, 12>(mode, a, grid, st);
    case 16: return launch_d<KIND, 16>(mode, a, grid, st);
    case 20: return launch_d<KIND, 20>(mode, a, grid, st);
    case 24: return launch_d<KIND, 24>(mode, a, grid, st);
    case 32: return launch_d<KIND, 32>(mode, a, grid, st);
  }
  return -2;
}
}  // namespace

// kind: KIND_* ; dk: kernel dims; returns 0 or -2 (no instantiation)
int grad2_launch_split(int kind, int dk, int mode, const Grad2Args& a, unsigned grid, hipStream_t st) {
  switch (kind) {
    case KIND_RBF: return launch_kind<KIND_RBF>(dk, mode, a, grid, st);
    case KIND_MATERN32: return launch_kind<KIND_MATERN32>(dk, mode, a, grid, st);
    case KIND_MATERN52: return launch_kind<KIND_MATERN52>(dk, mode, a, grid, st);
    case KIND_RQ: return launch_kind<KIND_RQ>(dk, mode, a, grid, st);
  }
  return -2;
}

}  // namespace gpamd
