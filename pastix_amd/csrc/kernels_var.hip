// kernels_var.hip -- diagonal-block and panel-solve kernels of the LDLt (sy) and LU (ge) variants.
// Same structure as k_diag_llt / k_trsm_llt in kernels.hip; the update kernel k_update is shared by all
// variants (the plan tags every piece with the arena of its A, B and C operands).
//
//   LDLt  factor_diag  = PASTIX_sytrf_block   compute_diag.c:262-307  (unit L below, D on the diagonal)
//         factor_trsm1d = kernel_trsm (sy)     compute_trsm.c:84-113   (TRSM "R","L","T","U" -> L*D, kept in the
//                                              second arena for the updates; then columns scaled by 1/d)
//   LU    factor_diag  = PASTIX_getrf_block + DimTrans   compute_diag.c:486-532,564-567 (no row pivoting)
//         factor_trsm1d = kernel_trsm (ge)     compute_trsm.c:58-67    (L_off = A U_d^-1 ; U_off^T = A' L_d^-T unit)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "plan.h"
#include "devmath.h"
#include "run_sync.h"
#include "diag_body.h"

namespace pastix_amd {


__global__ __launch_bounds__(512, 4) void k_diag_lu(double* __restrict__ L, double* __restrict__ U,
                                                 const PanelTask* __restrict__ tasks, double* __restrict__ dinv_ws,
                                                 double critere, long long* __restrict__ nbpivot) {
  PANEL_PRIO();
  __shared__ DiagLuLds S;
  const PanelTask tk = tasks[blockIdx.x];
  diag_lu_body<false>(S, L, U, tk, dinv_ws, critere, nbpivot, threadIdx.x);
}
// the run's diagonal kernel for LU (see k_run_diag, kernels.hip): resident workgroups popping ready diagonal tasks
__global__ __launch_bounds__(512, 4) void k_run_diag_lu(double* __restrict__ L, double* __restrict__ U, const RunD* __restrict__ rd,
                                                     const RunInfo* __restrict__ info, double* __restrict__ dinv_ws,
                                                     const double critere, long long* __restrict__ nbpivot, const RunCtl rc,
                                                     int* __restrict__ resident, const long long limit) {
  PANEL_PRIO();
  __shared__ DiagLuLds S;
  __shared__ int s_task;
  const int tid = threadIdx.x;
  if (tid == 0) __hip_atomic_fetch_add(resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (;;) {
    if (tid == 0) {
      const int v = run_pop(rc.qd, rc.ctl + RUN_HEAD + 64, rc.nd, rc.ctl + RUN_STUCK, 3 * limit, 0, rc.ctl + RUN_GO);   // (see k_run_diag)
      s_task = v;
      if (v >= 0) run_acquire();
    }
    __syncthreads();
    const int di = s_task;
    if (di < 0) break;
    const RunD d = rd[di];
    long long tp = 0;
    if (rc.prof && tid == 0) tp = wall_clock64();
    int ltid = threadIdx.x;
    asm volatile("" : "+v"(ltid));
    diag_lu_body<true>(S, L, U, d.pt, dinv_ws, critere, nbpivot, ltid);
    run_drain();
    __syncthreads();
    if (tid < 64) {
      for (int i = tid; i < d.tn; i += 64) run_dec_ticket(rc, info, d.t0 + i);
      if (rc.prof && tid == 0) {
        long long* pr = rc.prof + 4 * ((int64_t)rc.nticket + di);
        pr[0] = tp; pr[1] = tp; pr[2] = wall_clock64();
      }
    }
    __syncthreads();
  }
}
void launch_run_diag_lu(hipStream_t sd, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                        double critere, long long* nbpivot, const RunCtl& rc, int* resident, long long limit) {
  hipLaunchKernelGGL(k_run_diag_lu, dim3((unsigned)gd), dim3(512), 0, sd, ar.p[0], ar.p[1], rd, info, dinv, critere, nbpivot, rc,
                     resident, limit);
}


// ------------------------------------------------------------------------------------------------
// generalized panel solve  X^T[ct] = Tinv[ct] (A^T[ct] - sum_{p<ct} T[ct,p] X^T[p])   (see k_trsm_llt)
//   MODE 1 (LDLt):  T = unit-lower L_d, in/out = L arena; writes L*D to the U arena and L = (L*D) D^-1 to L
//   MODE 2 (LU, L): T(i,c) = U_d[c][i] (transposed access of the factored blok), in/out = L arena
//   MODE 3 (LU, U): T = unit-lower L_d (from the L arena), in/out = U arena
// ------------------------------------------------------------------------------------------------
template <int NT, int MODE>
__global__ __launch_bounds__(256, NT == 8 ? 4 : 1) void k_trsm_var(double* __restrict__ L, double* __restrict__ U,
                                                  const TrsmTask* __restrict__ tasks,
                                                  const double* __restrict__ dinv_ws) {
  PANEL_PRIO();
  const TrsmTask tk = tasks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int ld = tk.stride, w = tk.width;
  const int nbk = (w + 15) >> 4;
  const int rloc = wave * 16 + l15;
  if (wave * 16 >= tk.nrows) return;
  const bool rvalid = rloc < tk.nrows;
  double* X = (MODE == 3 ? U : L) + tk.off + tk.row0;
  double* Xp = X + rloc;
  const double* Xpc = X + min(rloc, tk.nrows - 1);
  const double* Td = L + tk.off;                       // factored diagonal blok (always in the L arena)
  const double* Ti = dinv_ws + tk.dinv_off + (MODE == 3 ? (int64_t)nbk * 256 : 0);

  d4 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const double v = Xpc[(int64_t)min(col, w - 1) * ld];
      acc[ct][q] = (rvalid && col < w) ? v : 0.0;
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
    if (ct < nbk) {
      const int li = ct * 16 + l15;
      const int lic = min(li, w - 1);
#pragma unroll
      for (int p = 0; p < ct; p++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int lc = p * 16 + g + 4 * q;
          const double tv = (MODE == 2) ? Td[lc + (int64_t)lic * ld] : Td[lic + (int64_t)lc * ld];
          const double a = (li < w) ? -tv : 0.0;
          acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[p][q], acc[ct], 0, 0, 0);
        }
      }
      d4 t = d4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double a = Ti[ct * 256 + l15 + 16 * (g + 4 * q)];
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[ct][q], t, 0, 0, 0);
      }
      acc[ct] = t;
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      if (MODE == 1) {
        const double dv = Td[min(col, w - 1) * (int64_t)(ld + 1)];
        if (rvalid && col < w) {
          (U + tk.off + tk.row0 + rloc)[(int64_t)col * ld] = acc[ct][q];           // L*D (compute_trsm.c:108-109)
          Xp[(int64_t)col * ld] = acc[ct][q] * fast_rcp(dv);                      // L   (:110)
        }
      } else {
        if (rvalid && col < w) Xp[(int64_t)col * ld] = acc[ct][q];
      }
    }
  }
}

void launch_diag_ldlt_w(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                        long long* nbpivot);      // kernels.hip: LDS-resident blok, MFMA trailing update (w <= 128)
void launch_diag_ldlt(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                      long long* nbpivot, int maxw) {
  (void)maxw;                                     // (cblks are at most 128 columns wide: api.cpp build_split)
  if (n <= 0) return;
  launch_diag_ldlt_w(s, L, tasks, n, dinv, critere, nbpivot);
}

void launch_diag_lu(hipStream_t s, double* L, double* U, const PanelTask* tasks, int64_t n, double* dinv,
                    double critere, long long* nbpivot) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_diag_lu, dim3((unsigned)n), dim3(512), 0, s, L, U, tasks, dinv, critere, nbpivot);
}

template <int MODE>
static void launch_trsm_mode(hipStream_t s, double* L, double* U, const TrsmTask* tasks, int64_t n,
                             const double* dinv, int maxw) {
  (void)maxw;
  hipLaunchKernelGGL((k_trsm_var<8, MODE>), dim3((unsigned)n), dim3(256), 0, s, L, U, tasks, dinv);
}

void launch_trsm_ldlt(hipStream_t s, double* L, double* U, const TrsmTask* tasks, int64_t n, const double* dinv,
                      int maxw) {
  if (n <= 0) return;
  launch_trsm_mode<1>(s, L, U, tasks, n, dinv, maxw);
}

void launch_trsm_lu(hipStream_t s, double* L, double* U, const TrsmTask* tasks, int64_t n, const double* dinv,
                    int maxw) {
  if (n <= 0) return;
  launch_trsm_mode<2>(s, L, U, tasks, n, dinv, maxw);
  launch_trsm_mode<3>(s, L, U, tasks, n, dinv, maxw);
}

}  // namespace pastix_amd
