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

// ------------------------------------------------------------------------------------------------
// LU diagonal blok (full w x w square in the L arena; transposed copy into the U arena), w <= 128 (wider cblks are re-cut).
// workspace: [nbk blocks: inverse of (U tile)^T, lower non-unit][nbk blocks: inverse of the unit L tile]
//
// Round 4: on the MFMA pipe, organised like the complex LDLt blok (kernels_z.hip diag_zsy_body): the square does not fit
// LDS beside a k_update workgroup, so its 16 x 16 tiles are resident in REGISTERS in the accumulator layout of
// v_mfma_f64_16x16x4 (lane (l15, g), register q = entry (row l15, column g + 4q)), as TWO triangles: plane 0 = the tiles
// (bi >= bj) of A, plane 1 = the same tiles of A^T -- the upper triangle, transposed -- so that a row of U is a COLUMN of
// plane 1 and sits where an MFMA operand is read.  Five tile pairs (80 VGPRs) per wave on waves 1-7; wave 0 carries the
// chain.  Per 16-column step t:
//   (S1) wave 0: PASTIX_getrf without row pivoting (compute_diag.c:432-469) on the tile T and its transpose, both held
//        negated: a(r, c) -= l(r, j) u(j, c) is one MFMA on each plane with the same two operands swapped (column j of T
//        scaled by 1 / d, column j of T^T); the transposed inverses of the unit-lower L tile and of (U tile)^T for the
//        panel solve ride along (two more MFMAs per column).  Meanwhile waves 1-7 finish the previous step's trailing update.
//   (S2) rows below the tile: L21 = A21 U11^-1 (plane 0) and columns right of it, transposed: U12^T = A12^T L11^-T
//        (plane 1), 16 rows per wave and plane: 4 MFMAs with the tile's inverse (compute_diag.c:496-508).
//   (S3a) the trailing update A22 -= L21 U12 (compute_diag.c:510-511) of the next column band, handed on through LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lu_readlane(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
struct DiagLuLds {
  static constexpr int XR = 112;
  double Ts[2][16][17];        // the step's diagonal tile and its transpose [plane][row][column]
  double Wl[2][256];           // plane 0: inv(U tile), plane 1: inv(L tile), as W[k * 16 + i] = inverse(i, k) of the plane's lower tile
  double Ps[2][16][XR];        // rows below the tile (plane 0) / columns right of it, transposed (plane 1), unsolved
  double Xs[2][16][XR];        // the same, solved: L21 [column][row] and U12^T [row of U][column of U]
};
// (512 threads; COH: every store of the blok is write-through -- the run launch hands it to other workgroups)
template <bool COH>
__device__ __forceinline__ void diag_lu_body(DiagLuLds& S, double* __restrict__ L, double* __restrict__ U, const PanelTask& tk,
                                             double* __restrict__ dinv_ws, const double critere,
                                             long long* __restrict__ nbpivot, const int tid) {
  constexpr int NS = 5;
  double* A = L + tk.off;
  double* Ud = U + tk.off;                                   // DimTrans (compute_diag.c:521-532, :564-567): Ud(b, a) = A(a, b)
  const int ld = tk.stride, w = tk.width;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int nbt = (w + 15) >> 4;
  // tile pairs (bi >= bj) but (0, 0), column by column (see diag_zsy_body): wave 1 + id % 7 holds pair id in slot id / 7
  d4 Cl[NS], Cu[NS];
  int tbi[NS], tbj[NS];
  if (wave > 0) {
#pragma unroll
    for (int sl = 0; sl < NS; sl++) {
      int id = (wave - 1) + 7 * sl, bj = 0, cnt = 7;
      if (id >= 14) { id -= 14; bj = 2; cnt = 6; while (id >= cnt) { id -= cnt; bj++; cnt--; } }
      else if (id >= 7) { id -= 7; bj = 1; }
      const int bi = (bj < 2 ? 1 : bj) + id;
      const bool on = bi < nbt;
      tbi[sl] = on ? bi : -1;
      tbj[sl] = bj;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * bi + l15, col = 16 * bj + g + 4 * q;
        const bool v = on && row < w && col < w;
        const int rc = min(row, w - 1), cc = min(col, w - 1);
        const double lo = pld<COH>(&A[rc + (int64_t)cc * ld]), up = pld<COH>(&A[cc + (int64_t)rc * ld]);
        Cl[sl][q] = v ? lo : 0.0;                            // A(row, col)
        Cu[sl][q] = v ? up : 0.0;                            // A(col, row) = A^T(row, col)
      }
      if (on && bj == 0) {                                   // what lies below / right of the first tile goes to LDS at once
#pragma unroll
        for (int q = 0; q < 4; q++) {
          S.Ps[0][g + 4 * q][16 * (bi - 1) + l15] = Cl[sl][q];
          S.Ps[1][g + 4 * q][16 * (bi - 1) + l15] = Cu[sl][q];
        }
      }
    }
  }
  // the trailing update of one resident pair with the solved rows / columns of step tp: A(r, c) -= L21(r, :) U12(:, c)
  auto update = [&](d4& cl, d4& cu, const int bi, const int bj, const int tp) {
    const int ro = 16 * (bi - tp - 1) + l15, co = 16 * (bj - tp - 1) + l15;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + g;
      const double xr = S.Xs[0][k][ro], xc = S.Xs[0][k][co];                 // L21(r, k), L21(c, k)
      const double yr = S.Xs[1][k][ro], yc = S.Xs[1][k][co];                 // U12(k, r), U12(k, c)
      cl = __builtin_amdgcn_mfma_f64_16x16x4f64(-yc, xr, cl, 0, 0, 0);        // A(r, c)   -= L21(r, k) U12(k, c)
      cu = __builtin_amdgcn_mfma_f64_16x16x4f64(-xc, yr, cu, 0, 0, 0);        // A^T(r, c) -= L21(c, k) U12(k, r)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (wave == 0) {
    int npiv = 0;
    const double cmin = fmax(critere, 2.2250738585072014e-308);
    for (int t = 0; t < nbt; t++) {
      const int kb = 16 * t, nb = min(16, w - kb);
      __syncthreads();                                       // (A) Ts / Ps hold column band t
      d4 Sl, Su, Vl, Vu;                                     // T, T^T and the two inverses, negated (see diag_llt_body)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        double lo, up;
        if (t == 0) {
          const int rc = min(l15, w - 1), cc = min(c, w - 1);
          lo = pld<COH>(&A[rc + (int64_t)cc * ld]); up = pld<COH>(&A[cc + (int64_t)rc * ld]);
          if (l15 >= w || c >= w) { lo = 0.0; up = 0.0; }
        } else {
          lo = S.Ts[0][l15][c]; up = S.Ts[1][l15][c];
        }
        Sl[q] = -lo;
        Su[q] = -up;
        Vl[q] = (c == l15) ? -1.0 : 0.0;
        Vu[q] = (c == l15) ? -1.0 : 0.0;
      }
      unroll_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 2, gj = j & 3;
        if (j < nb) {
          const bool ing = (g == gj), below = ing && l15 > j;
          double d = -lu_readlane(Sl[qj], j + 16 * gj);
          double y = __builtin_amdgcn_rcp(d);
          if (__builtin_expect(!(fabs(d) >= cmin), 0)) {       // |d| < critere (or NaN): compute_diag.c:440-444
            if (fabs(d) < critere) { d = critere; npiv++; }
            y = __builtin_amdgcn_rcp(d);
            Sl[qj] = (ing && l15 == j) ? -d : Sl[qj];
            Su[qj] = (ing && l15 == j) ? -d : Su[qj];
          }
          y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
          y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
          const double sl = Sl[qj] * y;                        // -l(:, j) = -a(:, j) / d
          const double lm = below ? sl : 0.0, um = below ? Su[qj] : 0.0;    // -l(r, j), -u(j, c): below the diagonal of their planes
          Sl[qj] = below ? sl : Sl[qj];                        // column j of L; the diagonal keeps -d, plane 1 keeps row j of U
          if (j < 15) {
            Sl = __builtin_amdgcn_mfma_f64_16x16x4f64(um, lm, Sl, 0, 0, 0);     // a(r, c)   -= l(r, j) u(j, c)
            Su = __builtin_amdgcn_mfma_f64_16x16x4f64(lm, um, Su, 0, 0, 0);     // a^T(r, c) -= l(c, j) u(j, r)
          }
          __builtin_amdgcn_sched_barrier(0);
          const double vl = ing ? Vl[qj] : 0.0;                // unit lower: the inverse's column is not scaled
          const double vu = ing ? Vu[qj] * y : 0.0;            // (U tile)^T: column j of the inverse over m(j, j) = d
          Vu[qj] = ing ? vu : Vu[qj];
          if (j < 15) {
            Vl = __builtin_amdgcn_mfma_f64_16x16x4f64(lm, vl, Vl, 0, 0, 0);     // W(:, i) -= W(:, j) l(i, j)
            Vu = __builtin_amdgcn_mfma_f64_16x16x4f64(um, vu, Vu, 0, 0, 0);     // W(:, i) -= W(:, j) u(j, i)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      double* dstu = dinv_ws + tk.dinv_off + (int64_t)t * 256;
      double* dstl = dinv_ws + tk.dinv_off + (int64_t)(nbt + t) * 256;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        if (l15 < nb && c < nb) {
          if (c < l15) {                                       // L below the diagonal
            pst<COH>(&A[(kb + l15) + (int64_t)(kb + c) * ld], -Sl[q]);
            pst<COH>(&Ud[(kb + c) + (int64_t)(kb + l15) * ld], -Sl[q]);
          }
          if (c <= l15) {                                      // U on and above it: plane 1 entry (l15, c) = u(c, l15)
            pst<COH>(&A[(kb + c) + (int64_t)(kb + l15) * ld], -Su[q]);
            pst<COH>(&Ud[(kb + l15) + (int64_t)(kb + c) * ld], -Su[q]);
          }
        }
        S.Wl[0][l15 * 16 + c] = -Vu[q];
        S.Wl[1][l15 * 16 + c] = -Vl[q];
        pst<COH>(&dstu[c + 16 * l15], -Vu[q]);
        pst<COH>(&dstl[c + 16 * l15], -Vl[q]);
      }
      __syncthreads();                                       // (B)
      __syncthreads();                                       // (C)
    }
    if (lane == 0 && npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
    return;
  }
  for (int t = 0; t < nbt; t++) {
    const int kb = 16 * t, nb = min(16, w - kb), rem = w - kb - nb;
    __syncthreads();                                         // (A)
    if (t > 0) {
#pragma unroll
      for (int sl = 0; sl < NS; sl++)
        if (tbi[sl] >= 0 && tbj[sl] > t) update(Cl[sl], Cu[sl], tbi[sl], tbj[sl], t - 1);     // (S3b) of step t - 1
    }
    __syncthreads();                                         // (B) the tile is factorized
    if ((wave - 1) * 16 < rem) {
      // (S2) block wave - 1 of the rows below (plane 0) and of the columns to the right (plane 1)
      const int ro = 16 * (wave - 1) + l15;
      d4 X = {0.0, 0.0, 0.0, 0.0}, Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const int k = 4 * ks + g;
        X = __builtin_amdgcn_mfma_f64_16x16x4f64(S.Wl[0][k * 16 + l15], S.Ps[0][k][ro], X, 0, 0, 0);
        Y = __builtin_amdgcn_mfma_f64_16x16x4f64(S.Wl[1][k * 16 + l15], S.Ps[1][k][ro], Y, 0, 0, 0);
      }
      const int rr = kb + nb + ro;                           // row of L21 / column of U12 in the blok
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        S.Xs[0][c][ro] = X[q];
        S.Xs[1][c][ro] = Y[q];
        if (c < nb && ro < rem) {
          pst<COH>(&A[rr + (int64_t)(kb + c) * ld], X[q]);     // L21(rr, kb + c)
          pst<COH>(&Ud[(kb + c) + (int64_t)rr * ld], X[q]);
          pst<COH>(&A[(kb + c) + (int64_t)rr * ld], Y[q]);     // U12(kb + c, rr)
          pst<COH>(&Ud[rr + (int64_t)(kb + c) * ld], Y[q]);
        }
      }
    }
    __syncthreads();                                         // (C) solved
    if (t + 1 < nbt) {
#pragma unroll
      for (int sl = 0; sl < NS; sl++)
        if (tbi[sl] >= 0 && tbj[sl] == t + 1) {                // (S3a) column band t + 1, then the hand-over
          update(Cl[sl], Cu[sl], tbi[sl], tbj[sl], t);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int c = g + 4 * q;
            if (tbi[sl] == t + 1) { S.Ts[0][l15][c] = Cl[sl][q]; S.Ts[1][l15][c] = Cu[sl][q]; }
            else { S.Ps[0][c][16 * (tbi[sl] - t - 2) + l15] = Cl[sl][q]; S.Ps[1][c][16 * (tbi[sl] - t - 2) + l15] = Cu[sl][q]; }
          }
        }
    }
  }
}
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
      const int v = run_pop(rc.qd, rc.ctl + RUN_HEAD + 64, rc.nd, rc.ctl + RUN_STUCK, 3 * limit);   // (see k_run_diag)
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
