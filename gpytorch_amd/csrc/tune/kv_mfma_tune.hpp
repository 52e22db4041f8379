// Tuning variants of the fused K*V MFMA kernel (RBF, DP = 4 only): same contract as kv_mfma.hpp,
// extra compile-time knobs.  Used by scripts/kv_tune.py through gpamd_kv_partials_variant_f32 to A/B
// structural choices on the GPU inside ONE process (cdna_hip_programming.md rule 24); winners are
// folded back into kv_mfma.hpp.  Not on the product path.
//
//   NI    row tiles (32 rows) per wave            BNV   j-tile staged in LDS
//   PIPE  software-pipeline K generation one MFMA step ahead
//   PRIO  s_setprio(1) around the MFMA cluster    GRAM  Gram-trick distance (4 fma) instead of differences
//   MINW  __launch_bounds__ min waves per SIMD     STG   1: synchronous staging (no register prefetch -> fewer VGPRs)
#pragma once
#include "../kv_mfma.hpp"

namespace gpamd {

template <int CT, int NI, int EX, int BNV, int PIPE, int PRIO, int GRAM, int MINW, int STG = 0>
__global__ __launch_bounds__(256, MINW) void kv_mfma_tune_kernel(KvArgs a) {
  constexpr int DP = 4, BN = BNV, LDT = BNV + 4, TC = 32 * CT;
  __shared__ __attribute__((aligned(16))) float smem[TC * LDT + BN * DP + BN];
  float* Vs = smem;
  float* Xs = smem + TC * LDT;
  float* Es = Xs + BN * DP;

  if (a.done && *a.done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int unit = blockIdx.x;
  const int s = unit / a.nrb, rb = unit - s * a.nrb;
  const int jbeg = s * a.jchunk;
  const int jend = min(a.m, jbeg + a.jchunk);
  const int ibase = rb * (4 * NI * 32) + wave * (NI * 32);

  float xi[NI][DP];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int i = min(ibase + ni * 32 + l31, a.n - 1);
    f32x4 v = *reinterpret_cast<const f32x4*>(a.X1 + (int64_t)i * DP);
    if constexpr (GRAM == 1) {
      // s = |xi|^2 + dot4([xi, 1], [-2 xj, |xj|^2]); slot 3 of xi holds |xi|^2 (used as the fma seed)
      xi[ni][0] = v[0]; xi[ni][1] = v[1]; xi[ni][2] = v[2];
      xi[ni][3] = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    } else {
      xi[ni][0] = v[0]; xi[ni][1] = v[1]; xi[ni][2] = v[2]; xi[ni][3] = v[3];
    }
  }

  f32x16 acc[NI][CT];
  float eacc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    eacc[ni] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ct][r] = 0.f;
  }

  constexpr int VQ = TC * (BN / 4) / 256;
  constexpr int XQ = (BN + 255) / 256;
  f32x4 vreg[VQ];
  f32x4 xreg[XQ];
  f32x4 ereg;

  auto stage_load = [&](int j0) {
#pragma unroll
    for (int r = 0; r < VQ; ++r) {
      int idx = tid + 256 * r;
      int c = idx / (BN / 4), q = idx % (BN / 4);
      int j = j0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < a.t) {
        const float* src = a.Vt + (int64_t)c * a.ldv + j;
        if (j + 4 <= jend) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < jend) v[e] = src[e];
        }
      }
      vreg[r] = v;
    }
#pragma unroll
    for (int r = 0; r < XQ; ++r) {
      int idx = tid + 256 * r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < BN) {
        int j = j0 + idx;
        if (j < jend) v = *reinterpret_cast<const f32x4*>(a.X2 + (int64_t)j * DP);
        if constexpr (GRAM == 1) {
          float nn = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
          v[0] *= -2.f; v[1] *= -2.f; v[2] *= -2.f; v[3] = nn;
        }
      }
      xreg[r] = v;
    }
    if constexpr (EX) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (tid < BN / 4) {
        int j = j0 + 4 * tid;
        const float* src = a.Vt + (int64_t)TC * a.ldv + j;
        if (j + 4 <= jend) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < jend) v[e] = src[e];
        }
      }
      ereg = v;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int r = 0; r < VQ; ++r) {
      int idx = tid + 256 * r;
      int c = idx / (BN / 4), q = idx % (BN / 4);
      *reinterpret_cast<f32x4*>(&Vs[c * LDT + 4 * q]) = vreg[r];
    }
#pragma unroll
    for (int r = 0; r < XQ; ++r) {
      int idx = tid + 256 * r;
      if (idx < BN) *reinterpret_cast<f32x4*>(&Xs[4 * idx]) = xreg[r];
    }
    if constexpr (EX) {
      if (tid < BN / 4) *reinterpret_cast<f32x4*>(&Es[4 * tid]) = ereg;
    }
  };

  auto keval = [&](const float (&x)[DP], const f32x4& xj) -> float {
    if constexpr (GRAM == 2) {  // ablation: no kernel generation (one dependent VALU op)
      return x[0] + xj[0];
    } else if constexpr (GRAM == 3) {  // ablation: no generation, no use of x_j
      return x[0];
    } else if constexpr (GRAM == 1) {
      float sq = __builtin_fmaf(x[0], xj[0], x[3]);
      sq = __builtin_fmaf(x[1], xj[1], sq);
      sq = __builtin_fmaf(x[2], xj[2], sq);
      sq = sq + xj[3];
      return __builtin_amdgcn_exp2f(-fmaxf(sq, 0.f));
    } else {
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float df = x[k] - xj[k];
        sq = __builtin_fmaf(df, df, sq);
      }
      return __builtin_amdgcn_exp2f(-sq);
    }
  };

  if (STG == 0 && jbeg < jend) stage_load(jbeg);
  for (int j0 = jbeg; j0 < jend; j0 += BN) {
    __syncthreads();
    if constexpr (STG == 1) stage_load(j0);
    stage_write();
    __syncthreads();
    if constexpr (STG == 0) {
      if (j0 + BN < jend) stage_load(j0 + BN);
    }

    float kvn[NI];  // PIPE: K elements of the NEXT step
    if constexpr (PIPE) {
      f32x4 xj = *reinterpret_cast<const f32x4*>(&Xs[(4 * h) * DP]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) kvn[ni] = keval(xi[ni], xj);
    }
#pragma unroll 2
    for (int g = 0; g < BN / 8; ++g) {
      const int jl = 8 * g + 4 * h;
      f32x4 av[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) av[ct] = *reinterpret_cast<const f32x4*>(&Vs[(ct * 32 + l31) * LDT + jl]);
      f32x4 ev;
      if constexpr (EX) ev = *reinterpret_cast<const f32x4*>(&Es[jl]);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float kv[NI];
        if constexpr (PIPE) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) kv[ni] = kvn[ni];
          // next step's x_j (wraps harmlessly into the first rows of the tile at the very end)
          int jn = jl + st + 1;
          if (st == 3) jn = 8 * (g + 1) + 4 * h;
          jn = jn & (BN - 1);
          f32x4 xj = *reinterpret_cast<const f32x4*>(&Xs[jn * DP]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) kvn[ni] = keval(xi[ni], xj);
        } else {
          f32x4 xj = *reinterpret_cast<const f32x4*>(&Xs[(jl + st) * DP]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) kv[ni] = keval(xi[ni], xj);
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[ni][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct][st], kv[ni], acc[ni][ct], 0, 0, 0);
          if constexpr (EX) eacc[ni] = __builtin_fmaf(kv[ni], ev[st], eacc[ni]);
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  float* Pout = a.P + (int64_t)s * a.pstride;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int i = ibase + ni * 32 + l31;
    if (i < a.n) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (c < a.t) Pout[(int64_t)c * a.ldo + i] = acc[ni][ct][r];
        }
    }
    if constexpr (EX) {
      float tot = eacc[ni] + __shfl_xor(eacc[ni], 32, 64);
      if (h == 0 && i < a.n) Pout[(int64_t)TC * a.ldo + i] = tot;
    }
  }
}

}  // namespace gpamd
