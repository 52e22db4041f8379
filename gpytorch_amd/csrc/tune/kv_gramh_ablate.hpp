// gpytorch_amd -- MI355X-native BBMM hot path.
// libgpamd_tune.so only: the ablation (ABL) and geometry (NW waves per workgroup, OCC resident waves per SIMD) builds of the split-operand
// kernel.  Until round 6 this file was a hand-synced COPY of ../kv_gramh.hpp as of round 3 (it had missed the packed pair generation, the tile
// lists of the far-pair culling and round 6's tied wait states); now the loop body is ../kv_gramh_body.inc, the very text the product kernel
// includes, whose `if constexpr (ABL ...)` branches hold every variant: what scripts/kgh_ablate.py and scripts/kgh_clock_power.py measure IS
// the product's loop.  ABL values: see ../kv_gramh.hpp.
#pragma once
#include "../kv_gramh.hpp"

namespace gpamd {

template <int KIND, int D, int CT, int NI, int EX, int ABL_ = 0, int NW_ = 4, int OCC_ = 2>
__global__ __launch_bounds__(64 * NW_) __attribute__((amdgpu_waves_per_eu(ABL_ == 7 ? 1 : OCC_, ABL_ == 7 ? 1 : OCC_)))
void kv_gramh_ablate_kernel(KvhArgs ka) {
  constexpr int SAFE = 0, ABL = ABL_, NW = NW_, OCC = (ABL_ == 7 ? 1 : OCC_);
#include "../kv_gramh_body.inc"
}

}  // namespace gpamd
