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

namespace pastix_amd {

typedef double d4 __attribute__((ext_vector_type(4)));

// forward substitution for the inverse of a 16x16 lower-triangular tile held in LDS (identity padding
// beyond nb); thread c computes column c.  lower(i,p) is read through the functor.
template <bool COH = false, class F>
__device__ __forceinline__ void tile_inverse(F lower, bool unit, int nb, int c, double (*Ti)[17], double* dst) {
  for (int i = 0; i < 16; i++) {
    double x;
    if (i >= nb || c >= nb) x = (i == c) ? 1.0 : 0.0;
    else if (i < c) x = 0.0;
    else {
      double s = (i == c) ? 1.0 : 0.0;
      for (int p = c; p < i; p++) s -= lower(i, p) * Ti[p][c];
      x = unit ? s : s / lower(i, i);
    }
    Ti[i][c] = x;
  }
  for (int i = 0; i < 16; i++) pst<COH>(&dst[i + 16 * c], Ti[i][c]);
}

// ------------------------------------------------------------------------------------------------
// LU diagonal blok (full w x w square in the L arena; transposed copy into the U arena at the end)
// workspace: [nbk blocks: inverse of (U tile)^T, lower non-unit][nbk blocks: inverse of the unit L tile]
// ------------------------------------------------------------------------------------------------
struct DiagLuLds {
  double Ts[16][17];
  double Lo[16][17];    // tile after getrf: unit L strictly below, U on and above the diagonal
  double Ti[16][17];
  double Xs[16][244];   // L rows below the tile   Xs[p][r] = L[r][p]
  double Ys[16][244];   // U columns right of tile Ys[p][c] = U[p][c]
};
// (256 threads; COH: every store of the blok is write-through -- the run launch hands it to other workgroups)
template <bool COH>
__device__ __forceinline__ void diag_lu_body(DiagLuLds& S, double* __restrict__ L, double* __restrict__ U, const PanelTask& tk,
                                             double* __restrict__ dinv_ws, const double critere,
                                             long long* __restrict__ nbpivot, const int tid) {
  auto& Ts = S.Ts; auto& Lo = S.Lo; auto& Ti = S.Ti; auto& Xs = S.Xs; auto& Ys = S.Ys;
  double* A = L + tk.off;
  const int ld = tk.stride, w = tk.width;
  const int ti = tid & 15, tc = tid >> 4;
  const int nbk = (w + 15) >> 4;
  int npiv = 0;
  for (int kb = 0; kb < w; kb += 16) {
    const int nb = min(16, w - kb), rem = w - kb - nb;
    if (ti < nb && tc < nb) Ts[ti][tc] = pld<COH>(&A[(kb + ti) + (int64_t)(kb + tc) * ld]);
    for (int j = 0; j < nb; j++) {                       // PASTIX_getrf, compute_diag.c:432-469
      __syncthreads();
      double d = Ts[j][j];
      if (fabs(d) < critere) { d = critere; if (tid == 0) npiv++; }
      const double inv = fast_rcp(d);
      if (ti < nb && tc < nb) {
        if (ti == j && tc >= j) Lo[j][tc] = (tc == j) ? d : Ts[j][tc];          // row j of U
        else if (tc == j && ti > j) Lo[ti][j] = Ts[ti][j] * inv;                // column j of L
        else if (ti > j && tc > j) Ts[ti][tc] -= (Ts[ti][j] * inv) * Ts[j][tc]; // GER
      }
    }
    __syncthreads();
    if (ti < nb && tc < nb) pst<COH>(&A[(kb + ti) + (int64_t)(kb + tc) * ld], Lo[ti][tc]);
    if (tid < 16) {
      // inverse of (U tile)^T : lower, non-unit, element (i,p) = U[p][i]
      tile_inverse<COH>([&](int i, int p) { return Lo[p][i]; }, false, nb, tid, Ti,
                   dinv_ws + tk.dinv_off + (int64_t)(kb >> 4) * 256);
    } else if (tid - 16 < rem) {
      // rows below: X = A U_T^-1  (TRSM inside getrf_block's panel, compute_diag.c:496-499 via getrf on m rows)
      const int rr = tid - 16;
      double* ap = A + (kb + nb + rr) + (int64_t)kb * ld;
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; c++) x[c] = pld<COH>(&ap[(int64_t)min(c, nb - 1) * ld]);
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nb) {
          double s = x[c];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < c) s -= x[p] * Lo[p][c];
          x[c] = s * fast_rcp(Lo[c][c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        Xs[c][rr] = (c < nb) ? x[c] : 0.0;
        if (c < nb) pst<COH>(&ap[(int64_t)c * ld], x[c]);
      }
    }
    __syncthreads();
    if (tid < 16) {
      // inverse of the unit-lower L tile
      tile_inverse<COH>([&](int i, int p) { return Lo[i][p]; }, true, nb, tid, Ti,
                   dinv_ws + tk.dinv_off + (int64_t)(nbk + (kb >> 4)) * 256);
    } else if (tid - 16 < rem) {
      // columns right of the tile: Y = L_T^-1 B  (TRSM "L","L","N","U", compute_diag.c:505-508)
      const int cc = tid - 16;
      double* bp = A + kb + (int64_t)(kb + nb + cc) * ld;
      double y[16];
#pragma unroll
      for (int r = 0; r < 16; r++) y[r] = pld<COH>(&bp[min(r, nb - 1)]);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        if (r < nb) {
          double s = y[r];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < r) s -= Lo[r][p] * y[p];
          y[r] = s;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        Ys[r][cc] = (r < nb) ? y[r] : 0.0;
        if (r < nb) pst<COH>(&bp[r], y[r]);
      }
    }
    __syncthreads();
    if (rem > 0) {                                       // A22 -= L21 U12 (full square, compute_diag.c:510-511)
      const int nt = (rem + 3) >> 2;
      double* Cb = A + (kb + nb) + (int64_t)(kb + nb) * ld;
      for (int id = tid; id < nt * nt; id += 256) {
        const int tr = id % nt, tcc = id / nt;
        double c[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) c[a][b] = 0.0;
        for (int p = 0; p < nb; p++) {
          double xa[4], xb[4];
#pragma unroll
          for (int a = 0; a < 4; a++) {
            xa[a] = Xs[p][min(4 * tr + a, 243)];
            xb[a] = Ys[p][min(4 * tcc + a, 243)];
          }
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) c[a][b] += xa[a] * xb[b];
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int a = 0; a < 4; a++) {
            const int r = 4 * tr + a, cc = 4 * tcc + b;
            if (r < rem && cc < rem) pst<COH>(&Cb[r + (int64_t)cc * ld], pld<COH>(&Cb[r + (int64_t)cc * ld]) - c[a][b]);
          }
      }
    }
    __syncthreads();
  }
  // DimTrans (compute_diag.c:521-532, :564-567): U arena diagonal blok = transpose of the factored blok
  double* Ud = U + tk.off;
  for (int id = tid; id < w * w; id += 256) {
    const int a = id % w, b = id / w;
    pst<COH>(&Ud[b + (int64_t)a * ld], pld<COH>(&A[a + (int64_t)b * ld]));
  }
  if (tid == 0 && npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
}
__global__ __launch_bounds__(256) void k_diag_lu(double* __restrict__ L, double* __restrict__ U,
                                                 const PanelTask* __restrict__ tasks, double* __restrict__ dinv_ws,
                                                 double critere, long long* __restrict__ nbpivot) {
  PANEL_PRIO();
  __shared__ DiagLuLds S;
  const PanelTask tk = tasks[blockIdx.x];
  diag_lu_body<false>(S, L, U, tk, dinv_ws, critere, nbpivot, threadIdx.x);
}
// the run's diagonal kernel for LU (see k_run_diag, kernels.hip): resident workgroups popping ready diagonal tasks
__global__ __launch_bounds__(256) void k_run_diag_lu(double* __restrict__ L, double* __restrict__ U, const RunD* __restrict__ rd,
                                                     const RunInfo* __restrict__ info, double* __restrict__ dinv_ws,
                                                     const double critere, long long* __restrict__ nbpivot, const RunCtl rc,
                                                     int* __restrict__ resident, const long long limit) {
  PANEL_PRIO();
  __shared__ DiagLuLds S;
  __shared__ int s_task;
  const int tid = threadIdx.x;
  if (tid == 0) __hip_atomic_fetch_add(resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  (void)limit;
  for (;;) {
    if (tid == 0) {
      const int v = run_pop(rc.qd, rc.ctl + RUN_HEAD + 64, rc.nd, rc.ctl + RUN_STUCK, 0);
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
  hipLaunchKernelGGL(k_run_diag_lu, dim3((unsigned)gd), dim3(256), 0, sd, ar.p[0], ar.p[1], rd, info, dinv, critere, nbpivot, rc,
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
  hipLaunchKernelGGL(k_diag_lu, dim3((unsigned)n), dim3(256), 0, s, L, U, tasks, dinv, critere, nbpivot);
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
