// run_sync.h -- device-only: the ready queues and the hand-off rules of the run launches (k_run_update, k_run_diag_*).
//
// Scheduling (plan.h RunInfo): every task has a counter of inputs that do not exist yet; a finished task decrements its
// consumers' counters and pushes the ones that reach zero into a ready ring (one for the tickets, one for the diagonal
// tasks; every task is pushed exactly once, so a ring never wraps: push = reserve the next slot with an atomic add on the
// tail, store the task there; pop = reserve the next slot with an atomic add on the head, wait for its task).  A
// workgroup never waits for a particular task: it runs the next one to become ready.
// Visibility (MI355X_MICROARCH, "inter-workgroup visibility", valid forms): a producer stores its bytes write-through
// (agent-scope stores, `sc1`), every storing wave waits for its stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a
// barrier, then its decrements and pushes (agent-scope atomics); a consumer that popped a task runs ONE agent-scope
// acquire (invalidates its CU's vector L1), waits for it, meets its workgroup at a barrier and then reads with plain
// loads.  Waiting is bounded in TIME: a workgroup that finds nothing to pop for `limit` ticks of the 100 MHz clock raises
// RUN_STUCK and leaves; every other one sees the flag within a few hundred polls and leaves too, the host returns
// PASTIX_AMD_ERR_DEVICE -- a wrong assumption fails the factorization, it cannot hang the device.
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
// take the NEXT slot of a ring of n slots (a wait-free reservation: an atomic add on the head, ~90 per microsecond on one
// word where a compare-and-swap loop of 500 contenders managed a fraction of one) and wait for the task that is -- or will
// be -- pushed there: the i-th reservation runs the i-th task to become ready.  Invariant: every task is pushed exactly once,
// only holders of a reservation wait, and a workgroup keeps looping until the ring is used up, so every push is consumed by a
// workgroup that already waits for that slot or by one that reserves it later; a slot of the chip is idle only while fewer
// tasks are ready than workgroups are waiting.  Returns -1 when the ring is used up or the run is stuck.
// `limit` > 0 bounds the wait (100 MHz ticks; on expiry RUN_STUCK is raised), 0 = no bound of its own.
// `maxwait` > 0: at most that many workgroups wait at a time -- one that would be the next leaves for good instead (returns
// -1 without a reservation; the launch has min(tasks, 2 x CUs) workgroups, so the others still serve every push).  Round 4's
// mitigation of the stop of the TWO-kernel form (a second persistent kernel on another queue; the one-kernel
// form of real LLt / LDLt has not stopped in 14 000 soaked factorizations): workgroups that never end leave no room on the chip.
// `go` != nullptr (the resident diagonal workers): the wait is timed only once *go is set, i.e. once the tickets' kernel
// runs; until then only an absolute bound of 60 s applies (a tool that serializes kernel launches never starts that kernel
// beside this one: the workers must not spin for ever).
__device__ __forceinline__ int run_pop(const int32_t* ring, int32_t* head, const int n, int32_t* stuck, const long long limit,
                                       const int maxwait = 0, const int32_t* go = nullptr) {
  if (maxwait > 0) {
    const int waiting = run_ld(head) - run_ld(head + (RUN_TAIL - RUN_HEAD));
    if (waiting >= maxwait) return -1;
  }
  const int h = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (h >= n) return -1;
  const int32_t* slot = ring + (int64_t)h * RUN_SLOT;
  int v = run_ld(slot);
  if (v >= 0) return v;
  long long t0 = wall_clock64();
  const long long tstart = t0;
  int it = 0;
  while ((v = run_ld(slot)) < 0) {
    ++it;
    // (now and then a read-modify-write, which executes at the coherence point whatever this XCD's L2 holds of the line)
    if ((it & 63) == 0) {
      if ((v = __hip_atomic_fetch_or(const_cast<int32_t*>(slot), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= 0) break;
      if (run_ld(stuck)) return -1;
      const long long now = wall_clock64();
      bool never_started = false;
      if (go && !run_ld(go)) {                        // (the tickets' kernel has not started: no clock yet)
        t0 = now;
        if (now - tstart < 6000000000LL) continue;
        never_started = true;                         // 60 s and still no tickets' kernel: give up (the note says how long)
        t0 = tstart;
      }
      if (never_started || (limit > 0 && now - t0 > limit)) {
        // (the first one to give up leaves a note for the host: which slot it waited for, how long)
        if (__hip_atomic_exchange(stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          run_st(stuck + 2, h);
          run_st(stuck + 3, n);
          run_st(stuck + 4, (int)((now - t0) / 100000));            // ms
          run_st(stuck + 5, (int)(limit / 100000));
          run_st(stuck + 6, it);
        }
        return -1;
      }
    }
    // fast at first (a chain task is picked up within a microsecond), slower the longer nothing comes
    if (it < 16) __builtin_amdgcn_s_sleep(4);
    else if (it < 4096) __builtin_amdgcn_s_sleep(32);
    else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
  }
  return v;
}
__device__ __forceinline__ void run_push(int32_t* ring, int32_t* tail, const int task) {
  const int pos = __hip_atomic_fetch_add(tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)__hip_atomic_exchange(ring + (int64_t)pos * RUN_SLOT, task, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// an input of ticket c exists now
__device__ __forceinline__ void run_dec_ticket(const RunCtl& rc, const RunInfo* __restrict__ info, const int c) {
  (void)info;
  if (__hip_atomic_fetch_add(rc.cnt + c, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1)
    run_push(rc.q, rc.ctl + RUN_TAIL, c);
}
__device__ __forceinline__ void run_dec_diag(const RunCtl& rc, const int d) {
  if (__hip_atomic_fetch_add(rc.cnt + rc.nticket + d, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) {
    if (rc.onek) run_push(rc.q, rc.ctl + RUN_TAIL, rc.nticket + d);
    else run_push(rc.qd, rc.ctl + RUN_TAIL + 64, d);
  }
}
// consumer side, after the pop: acquire + wait (the caller's barrier follows)
__device__ __forceinline__ void run_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// producer side: every wave calls this after its last store of handed-off bytes, then the workgroup's barrier, then the
// decrements
__device__ __forceinline__ void run_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// write-through store / plain store of a panel entry
template <bool COH>
__device__ __forceinline__ void pst(double* p, const double v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// a load that must see what another wave of this workgroup has stored write-through since (bodies that re-read their own
// stores from memory: the write-through store does not refresh the CU's vector L1)
template <bool COH>
__device__ __forceinline__ double pld(const double* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

}  // namespace pastix_amd
