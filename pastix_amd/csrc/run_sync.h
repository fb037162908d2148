// run_sync.h -- device-only: waiting and publishing inside the run launches (k_run_update, k_run_panel).
//
// Visibility (MI355X_MICROARCH, "inter-workgroup visibility", valid forms): a producer stores its bytes write-through
// (agent-scope stores, `sc1`), every storing wave waits for its stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a
// barrier, ONE lane stores the flag (`sc1`); a consumer polls the flag with agent-scope loads, runs ONE agent-scope
// acquire (invalidates its CU's vector L1), waits for it, meets its workgroup at a barrier and then reads with plain
// loads.  A wait is bounded in TIME: after `limit` ticks of the 100 MHz clock the waiter raises RUN_STUCK and goes on;
// every other waiter sees the flag within a few hundred polls and goes on too, the host returns PASTIX_AMD_ERR_DEVICE --
// a wrong assumption fails the factorization, it cannot hang the device.
#pragma once
#include <hip/hip_runtime.h>

#include "plan.h"

namespace pastix_amd {

__device__ __forceinline__ int run_ld(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void run_st(int32_t* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until *p >= need (one lane per flag; other lanes of the wave may poll other flags)
__device__ __forceinline__ void run_poll(const int32_t* p, const int need, int32_t* stuck, const long long limit) {
  if (run_ld(p) >= need) return;
  const long long t0 = wall_clock64();
  int it = 0;
  while (run_ld(p) < need) {
    ++it;
    if ((it & 127) == 0) {
      if (run_ld(stuck)) return;
      if (wall_clock64() - t0 > limit) { run_st(stuck, 1); return; }
    }
    if (it < 32) __builtin_amdgcn_s_sleep(2);
    else __builtin_amdgcn_s_sleep(20);
  }
}
// consumer side, after the polls of the workgroup's polling wave: acquire + wait (the caller's barrier follows)
__device__ __forceinline__ void run_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// producer side: every wave calls this after its last store of handed-off bytes, then the workgroup's barrier, then
// one lane's run_st of the flag
__device__ __forceinline__ void run_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// write-through store / plain store of a panel entry
template <bool COH>
__device__ __forceinline__ void pst(double* p, const double v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

}  // namespace pastix_amd
