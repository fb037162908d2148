// devmath.h -- device-only scalar helpers (include after <hip/hip_runtime.h>)
#pragma once
#include <type_traits>

// The kernels of the panel stream (urgent updates, diagonal bloks, panel solves) share their CUs with a bulk k_update
// workgroup; their waves are dependent chains, the bulk waves are MFMA-bound: the instruction arbiter is told so.
#ifndef PASTIX_AMD_NO_PRIO
#define PANEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define PANEL_PRIO() ((void)0)
#endif

namespace pastix_amd {

// Latency-critical scalar math of the diagonal-blok kernels: hardware estimate (v_rcp_f64 / v_rsq_f64)
// refined by two Newton steps in FMA arithmetic (<= 2 ulp), instead of the ~40-instruction correctly
// rounded software sequences hipcc emits for 1.0/x and sqrt(x): the per-column chain of the diagonal
// factorization is serial, so instruction latency, not throughput, is what a dependency level waits for.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
// returns sqrt(x) and 1/sqrt(x) for x > 0
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& rs) {
  double y = __builtin_amdgcn_rsq(x);
  double h = 0.5 * y;
  y = __builtin_fma(h, __builtin_fma(-x * y, y, 1.0), y);
  h = 0.5 * y;
  y = __builtin_fma(h, __builtin_fma(-x * y, y, 1.0), y);
  double r = x * y;
  r = __builtin_fma(__builtin_fma(-r, r, x), 0.5 * y, r);   // one correction of the root
  s = r;
  rs = y;
}

// Broadcast of lane K of every row of 16 lanes to the lanes of that row (DPP row_newbcast; gfx90a+ also on 64-bit
// operands: one v_mov_b64_dpp).  The tile kernels keep row (lane & 15) of a 16 x 16 tile on every lane, i.e. the four
// rows of a wave hold copies, so this is "entry of tile row K" for every lane -- what two v_readlane + a wait state into
// a scalar pair did before, without leaving the vector pipe.  Source lanes must be ACTIVE (a disabled source lane leaves
// the destination unchanged): callers keep the whole wave in the computation and predicate only their stores.
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false);
}
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, false);
}
// compile-time loop over [B, E): f(std::integral_constant<int, i>)
template <int B, int E, class F>
__device__ __forceinline__ void unroll_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    unroll_for<B + 1, E>(f);
  }
}

}  // namespace pastix_amd
