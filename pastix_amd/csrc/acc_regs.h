// acc_regs.h -- device-only: the accumulation registers a[0:63] by NAME (inline assembly), outside the compiler's register
// allocation.  k_update / k_run_update keep their 128 x 128 tile there (kernels_update.hip: why); the panel-solve
// tickets of the run launch, in which the update path's accumulators are idle, PARK their resident tiles
// there (trsm_parked).  Wait states the compiler cannot insert for code it does not
// see (cdna_hip_programming.md 5.7 item 2) are in the strings.  tests/test_kernel_audit.py requires that no instruction of
// the compiler's touches an AGPR in the files that use these.
#pragma once
#include <hip/hip_runtime.h>

namespace pastix_amd {

// ---- accumulators in a[0:63] ---------------------------------------------------------------------
#define PA_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define PA_ACC_CLOBBER                                                                                           \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", PA_A8(1), PA_A8(2), PA_A8(3), PA_A8(4), PA_A8(5), \
      "a60", "a61", "a62", "a63"

// sub-tile T = 4 mi + ni:  acc[T] += an (MFMA "A": target columns) x bm (MFMA "B": target rows)
template <int T>
__device__ __forceinline__ void acc_mfma(const double an, const double bm) {
  asm volatile("v_mfma_f64_16x16x4_f64 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(an), "v"(bm), "n"(8 * T), "n"(8 * T + 7)
               : PA_ACC_CLOBBER);
}
#define PA_Z4(i) "v_accvgpr_write_b32 a" #i "0, 0\n\tv_accvgpr_write_b32 a" #i "1, 0\n\tv_accvgpr_write_b32 a" #i "2, 0\n\tv_accvgpr_write_b32 a" #i "3, 0\n\t" \
                 "v_accvgpr_write_b32 a" #i "4, 0\n\tv_accvgpr_write_b32 a" #i "5, 0\n\tv_accvgpr_write_b32 a" #i "6, 0\n\tv_accvgpr_write_b32 a" #i "7, 0\n\t" \
                 "v_accvgpr_write_b32 a" #i "8, 0\n\tv_accvgpr_write_b32 a" #i "9, 0\n\t"
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\t"
               "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"
               "v_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\t" PA_Z4(1) PA_Z4(2) PA_Z4(3) PA_Z4(4) PA_Z4(5)
               "v_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\t"
               "s_nop 7" ::: PA_ACC_CLOBBER);          // (v_accvgpr_write -> MFMA SrcC)
}
// every MFMA has retired its D before anything but an MFMA reads the accumulators (16-pass DGEMM: 19 states)
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 7" ::: PA_ACC_CLOBBER); }
// register q (0..3) of sub-tile T: rows l15 of band mi, column g + 4 q of band ni
template <int T, int Q>
__device__ __forceinline__ double acc_read() {
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_accvgpr_read_b32 %1, a[%c3]"
               : "=v"(lo), "=v"(hi)
               : "n"(8 * T + 2 * Q), "n"(8 * T + 2 * Q + 1)
               : PA_ACC_CLOBBER);
  return __hiloint2double(hi, lo);
}

// Wait states are ours on both sides of an asm statement (cdna_hip_programming.md 5.7 item 2): FRESH = the value was just
// produced by an MFMA of the compiler's (its D must have retired before a VALU move reads it: the nops lead the string);
// acc_read_m = the value feeds an MFMA of the compiler's next (VALU write -> MFMA operand: the nops end the string).
template <int T, int Q, bool FRESH = false>
__device__ __forceinline__ void acc_write(const double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (FRESH)
    asm volatile("s_nop 15\n\ts_nop 7\n\tv_accvgpr_write_b32 a[%c2], %0\n\tv_accvgpr_write_b32 a[%c3], %1" ::"v"(lo), "v"(hi),
                 "n"(8 * T + 2 * Q), "n"(8 * T + 2 * Q + 1)
                 : PA_ACC_CLOBBER);
  else
    asm volatile("v_accvgpr_write_b32 a[%c2], %0\n\tv_accvgpr_write_b32 a[%c3], %1" ::"v"(lo), "v"(hi), "n"(8 * T + 2 * Q),
                 "n"(8 * T + 2 * Q + 1)
                 : PA_ACC_CLOBBER);
}
template <int T, int Q>
__device__ __forceinline__ double acc_read_m() {
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_accvgpr_read_b32 %1, a[%c3]\n\ts_nop 3"
               : "=v"(lo), "=v"(hi)
               : "n"(8 * T + 2 * Q), "n"(8 * T + 2 * Q + 1)
               : PA_ACC_CLOBBER);
  return __hiloint2double(hi, lo);
}
}  // namespace pastix_amd
