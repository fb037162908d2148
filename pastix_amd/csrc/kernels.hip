// kernels.hip -- gfx950 (MI355X / CDNA4) device kernels of the sopalin numerical factorization.
//
// Three kernels replace the reference's per-cblk CPU step compute_1d (sopalin_compute.c:747-863); k_diag and k_trsm
// (real LLt; LDLt's diagonal kernel) and the triangular solves are in this file, k_update in kernels_update.hip:
//   k_diag   : factor_diag  (compute_diag.c:538-605)  blocked LLt of the w x w diagonal blok with the
//              static-pivot clamp (compute_diag.c:133-137), plus the 16x16 diagonal-block inverses
//              the panel solve uses;
//   k_trsm   : factor_trsm1d (compute_trsm.c:128-171) panel solve X = A L^-T held entirely in MFMA
//              accumulator registers (v_mfma_f64_16x16x4_f64), one wave per 16 panel rows;
//   k_update : compute_1dgemm = compute_contrib_compact + add_contrib_local
//              (sopalin_compute.c:865-1032, :270-374, :391-598) fused: every workgroup owns one
//              128x128 tile of a target panel and accumulates all contributions ("pieces") that the
//              plan scheduled into this launch in MFMA accumulators, then subtracts them from the
//              tile once.  Tile ownership replaces mutex_blok[] and makes the result deterministic.
//
// MFMA f64 16x16x4 lane maps (measured on gfx950, tools/probe_mfma_f64.hip):
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15];
//   C/D: lane l, register q holds D[i = (l>>4) + 4q][j = l&15].
// The update kernel feeds the target COLUMN index as MFMA "i" and the target ROW index as MFMA "j" so
// that each accumulator register maps to 16 consecutive rows of the column-major panel (128-byte
// segments for the read-modify-write of the tile).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "plan.h"
#include "devmath.h"
#include "run_sync.h"
#include "diag_body.h"

namespace pastix_amd {

__global__ __launch_bounds__(512, 4) void k_diag_llt_w(double* __restrict__ L, const PanelTask* __restrict__ tasks,
                                                    double* __restrict__ dinv_ws, double critere,
                                                    long long* __restrict__ nbpivot, int* __restrict__ errflag) {
  __shared__ double D[DIAG_LDS_DOUBLES];
  __shared__ double Wl[256];
  PANEL_PRIO();
  const PanelTask tk = tasks[blockIdx.x];
  diag_llt_body<false>(D, Wl, L, tk, dinv_ws, critere, nbpivot, errflag, threadIdx.x);
}

__global__ __launch_bounds__(512, 4) void k_diag_ldlt_w(double* __restrict__ L, const PanelTask* __restrict__ tasks,
                                                     double* __restrict__ dinv_ws, double critere,
                                                     long long* __restrict__ nbpivot) {
  __shared__ double D[DIAG_LDS_DOUBLES];
  __shared__ double S[320];
  PANEL_PRIO();
  const PanelTask tk = tasks[blockIdx.x];
  diag_ldlt_body<false>(D, S, L, tk, dinv_ws, critere, nbpivot, threadIdx.x);
}

void launch_diag_ldlt_w(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                        long long* nbpivot) {
  hipLaunchKernelGGL(k_diag_ldlt_w, dim3((unsigned)n), dim3(512), 0, s, L, tasks, dinv, critere, nbpivot);
}

// ------------------------------------------------------------------------------------------------
// k_trsm : X = A * L_d^-T for 64 panel rows per workgroup (one wave per 16 rows), X^T tiles live in
// MFMA accumulators: X^T[ct] = Tinv[ct] * (A^T[ct] - sum_{p<ct} L[ct,p] X^T[p]).  The accumulator of
// tile p (register q = rows g+4q of X^T[p]) is used directly as the B operand of k-step q; the A
// operand supplies the matching column g+4q of L[ct,p], so no lane shuffles or LDS are needed.
// ------------------------------------------------------------------------------------------------
// (a wave solves 16 panel rows: 4 waves = the 64 rows of a TrsmTask, 8 waves = the 128 rows of a run task; COH: the
// solved rows are stored write-through, run_sync.h)
template <int NT, bool COH>
__device__ __forceinline__ void trsm_llt_body(double* __restrict__ L, const TrsmTask& tk, const double* __restrict__ dinv_ws) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int ld = tk.stride, w = tk.width;
  const int nbk = (w + 15) >> 4;
  const int rloc = wave * 16 + l15;
  if (wave * 16 >= tk.nrows) return;
  const bool rvalid = rloc < tk.nrows;
  double* Ap = L + tk.off + tk.row0 + rloc;          // panel row of this lane
  const double* Apc = L + tk.off + tk.row0 + min(rloc, tk.nrows - 1);   // clamped: always readable
  const double* Ld = L + tk.off;                     // diagonal blok (factored)
  const double* Ti = dinv_ws + tk.dinv_off;

  d4 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const double v = Apc[(int64_t)min(col, w - 1) * ld];
      acc[ct][q] = (rvalid && col < w) ? v : 0.0;
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
    if (ct < nbk) {
      const int li = ct * 16 + l15;                  // row of L supplied by this lane
      const int lic = min(li, w - 1);
#pragma unroll
      for (int p = 0; p < ct; p++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int lc = p * 16 + g + 4 * q;
          const double lv = Ld[lic + (int64_t)lc * ld];
          const double a = (li < w) ? -lv : 0.0;
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
      if (rvalid && col < w) pst<COH>(&Ap[(int64_t)col * ld], acc[ct][q]);
    }
  }
}
template <int NT>
__global__ __launch_bounds__(256, NT == 8 ? 4 : 1) void k_trsm_llt(double* __restrict__ L, const TrsmTask* __restrict__ tasks,
                                                  const double* __restrict__ dinv_ws) {
  PANEL_PRIO();
  const TrsmTask tk = tasks[blockIdx.x];
  trsm_llt_body<NT, false>(L, tk, dinv_ws);
}

// ---- the run's diagonal kernel (real LLt) ---------------------------------------------------------------------------
// The diagonal-blok tasks of the run's levels (plan.h RunD) on a few RESIDENT workgroups, started before the run's
// launch and alive until the last diagonal task is taken: each pops ready diagonal tasks from their ring (a task is ready
// when the last update of its diagonal tile has counted down its counter), factorizes the blok exactly as k_diag_llt_w
// does, stores it write-through and counts down the cblk's panel-solve tickets.  (Diagonal tasks are not tickets of
// k_run_update because their code needs 128 VGPRs beside that kernel's 64 accumulation registers; a workgroup that is
// resident never waits for a slot behind tickets that wait for it.  The host checks `resident` before it launches
// k_run_update.)
// FT: 0 LLt, 1 LDLt
template <int FT>
__global__ __launch_bounds__(512, 4) void k_run_diag(double* __restrict__ L, const RunD* __restrict__ rd,
                                                     const RunInfo* __restrict__ info, double* __restrict__ dinv_ws,
                                                     const double critere, long long* __restrict__ nbpivot,
                                                     int* __restrict__ errflag, const RunCtl rc,
                                                     int* __restrict__ resident, const long long limit) {
  __shared__ double D[DIAG_LDS_DOUBLES];
  __shared__ double Ri[320];                         // (LLt: the tile's inverse; LDLt: diagonal, reciprocals, inverse)
  __shared__ int s_task;
  PANEL_PRIO();
  const int tid = threadIdx.x;
  if (tid == 0) __hip_atomic_fetch_add(resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (;;) {
    if (tid == 0) {
      // (bounded like the tickets' wait, three times as long: a diagonal worker legitimately finds nothing while the levels
      // below the run are factorized; but a tool that serializes kernel launches -- rocprofv3 --pmc does -- never starts the
      // tickets' kernel beside this one, and the workers must not spin for ever: RUN_STUCK, everybody leaves, ERR_DEVICE)
      const int v = run_pop(rc.qd, rc.ctl + RUN_HEAD + 64, rc.nd, rc.ctl + RUN_STUCK, 3 * limit, 0, rc.ctl + RUN_GO);
      s_task = v;
      if (v >= 0) run_acquire();
    }
    __syncthreads();
    const int di = s_task;
    if (di < 0) break;                               // every diagonal task is taken (or the run is stuck)
    const RunD d = rd[di];
    long long tp = 0;
    if (rc.prof && tid == 0) tp = wall_clock64();
    // (the thread index is laundered per task: otherwise everything the body derives from it is hoisted out of this loop
    // and kept in registers across it -- 59 spilled VGPRs instead of the 17 of the same body in k_diag_llt_w)
    int ltid = threadIdx.x;
    asm volatile("" : "+v"(ltid));
    if constexpr (FT == 0) diag_llt_body<true>(D, Ri, L, d.pt, dinv_ws, critere, nbpivot, errflag, ltid);
    else diag_ldlt_body<true>(D, Ri, L, d.pt, dinv_ws, critere, nbpivot, ltid);
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

// ------------------------------------------------------------------------------------------------
// coefficient fill: scatter (destination, value) pairs  (Csc2solv_cblk, csc_intern_solve.c:65-132)
// ------------------------------------------------------------------------------------------------
__global__ void k_scatter(double* __restrict__ dst, const int64_t* __restrict__ idx,
                          const double* __restrict__ val, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[idx[i]] = val[i];
}

// The same solve for cblks of at most 128 columns on four waves: wave q keeps the 32 columns it will process in
// registers (one memory round trip for the whole blok, all loads in flight at once), the waves take turns on the
// chain and hand x over through LDS.
template <int MODE, int NR>
__global__ __launch_bounds__(256) void k_solve_diag_q(const double* __restrict__ L,
                                                      const SolveTask* __restrict__ tasks,
                                                      double* __restrict__ x, int64_t ldx, int unit) {
  __shared__ double xs[NR][128];
  const SolveTask tk = tasks[blockIdx.x];
  const double* A = L + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t rcl[2];
  double rinv[2], a[32][2];
#pragma unroll
  for (int j = 0; j < 2; j++) rcl[j] = min(lane + 64 * j, w - 1);
  if constexpr (MODE == 1) {            // (see k_solve_diag_q1: column-wise reads turned through LDS)
    extern __shared__ double S[];
    const int ldl = w | 1;
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int64_t c = min(wave + 4 * i, w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = A[rcl[j] + c * ld];
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int c = wave + 4 * i;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int r = lane + 64 * j;
        if (c < w && r >= c && r < w) S[r + c * ldl] = a[i][j];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int g = 32 * wave + i;
      const int64_t c = min(max(MODE == 0 ? g : w - 1 - g, 0), w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = A[rcl[j] + c * ld];
    }
  }
#pragma unroll
  for (int j = 0; j < 2; j++) rinv[j] = unit ? 1.0 : 1.0 / A[rcl[j] + rcl[j] * ld];
  for (int i = tid; i < NR * 128; i += 256) {
    const int q = i >> 7, r = i & 127;
    if (r < w) xs[q][r] = x[q * ldx + tk.fcol + r];
  }
  __syncthreads();
  if constexpr (MODE == 1) {
    extern __shared__ double S[];
    const int ldl = w | 1;
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int c = min(max(w - 1 - (32 * wave + i), 0), w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = S[c + (int)rcl[j] * ldl];
    }
  }
  // systolic: at step t wave q runs its 32 columns of the chain for right-hand side t - q.  One copy of the chain
  // per wave (Q a compile-time constant: column numbers, slots and lane tests fold)
  const bool rw[2] = {lane < w, lane + 64 < w};
  auto chain = [&](auto Q, int k) {
    constexpr int q = decltype(Q)::value;
    // laundered lane id and width: without them the compiler hoists every step's lane masks and column numbers
    // out of the t loop (they do not depend on t) and spills up to 1600 SGPRs
    int lane = threadIdx.x, w = tk.width;
    asm volatile("" : "+v"(lane), "+s"(w));
    lane &= 63;
    double xr[2];
#pragma unroll
    for (int j = 0; j < 2; j++) xr[j] = xs[k][rcl[j]];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int g = 32 * q + i;
      if (g < w) {                                     // (no break: the 32 steps must stay unrolled)
        const int c = MODE == 0 ? g : w - 1 - g;
        const int slot = c >> 6, src = c & 63;
        const double v = slot ? xr[1] * rinv[1] : xr[0] * rinv[0];
        const double xc = readlane_f64(v, src);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int r = lane + 64 * j;
          const bool upd = MODE == 0 ? (r > c && rw[j]) : (r < c);
          const double nx = upd ? __builtin_fma(-a[i][j], xc, xr[j]) : xr[j];
          xr[j] = (r == c) ? xc : nx;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
      if (rw[j]) xs[k][lane + 64 * j] = xr[j];
  };
  for (int t = 0; t < 3 + NR; t++) {
    const int k = t - wave;
    if (k >= 0 && k < NR && 32 * wave < w) {
      if (wave == 0) chain(std::integral_constant<int, 0>{}, k);
      else if (wave == 1) chain(std::integral_constant<int, 1>{}, k);
      else if (wave == 2) chain(std::integral_constant<int, 2>{}, k);
      else chain(std::integral_constant<int, 3>{}, k);
    }
    __syncthreads();
  }
  for (int i = tid; i < NR * 128; i += 256) {
    const int q = i >> 7, r = i & 127;
    if (r < w) x[q * ldx + tk.fcol + r] = xs[q][r];
  }
}

template <int MODE, class T>
__global__ __launch_bounds__(256) void k_solve_diag_q1(const T* __restrict__ L,
                                                      const SolveTask* __restrict__ tasks,
                                                      double* __restrict__ x, int unit) {
  __shared__ double xs[128];
  const SolveTask tk = tasks[blockIdx.x];
  const T* A = L + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t rcl[2];
  double rinv[2], a[32][2];
#pragma unroll
  for (int j = 0; j < 2; j++) rcl[j] = min(lane + 64 * j, w - 1);
  if constexpr (MODE == 1) {
    // L^T: row c of L is needed with the lanes along the columns.  Read the blok column-wise (coalesced) and
    // turn it through LDS (dynamic: lw x (lw|1) doubles for the widest cblk of the level; odd leading dimension)
    // instead of 64 scattered 8-byte reads per instruction
    extern __shared__ double S[];
    const int ldl = w | 1;
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int64_t c = min(wave + 4 * i, w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = A[rcl[j] + c * ld];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) rinv[j] = unit ? 1.0 : 1.0 / A[rcl[j] + rcl[j] * ld];
    if (tid < w) xs[tid] = x[tk.fcol + tid];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int c = wave + 4 * i;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int r = lane + 64 * j;
        if (c < w && r >= c && r < w) S[r + c * ldl] = a[i][j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int c = min(max(w - 1 - (32 * wave + i), 0), w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = S[c + (int)rcl[j] * ldl];     // (entries with r > c are never used)
    }
  } else {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int g = 32 * wave + i;
      const int64_t c = min(max(MODE == 0 ? g : w - 1 - g, 0), w - 1);
#pragma unroll
      for (int j = 0; j < 2; j++) a[i][j] = A[rcl[j] + c * ld];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) rinv[j] = unit ? 1.0 : 1.0 / A[rcl[j] + rcl[j] * ld];
    if (tid < w) xs[tid] = x[tk.fcol + tid];
    __syncthreads();
  }
  for (int q = 0; q < 4; q++) {
    if (wave == q && 32 * q < w) {
      double xr[2];
#pragma unroll
      for (int j = 0; j < 2; j++) xr[j] = xs[rcl[j]];
#pragma unroll
      for (int i = 0; i < 32; i++) {
        const int g = 32 * q + i;
        if (g >= w) break;
        const int c = MODE == 0 ? g : w - 1 - g;
        const int slot = c >> 6, src = c & 63;
        const double v = slot ? xr[1] * rinv[1] : xr[0] * rinv[0];
        const double xc = readlane_f64(v, src);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int r = lane + 64 * j;
          const bool upd = MODE == 0 ? (r > c && r < w) : (r < c);
          const double nx = upd ? __builtin_fma(-a[i][j], xc, xr[j]) : xr[j];
          xr[j] = (r == c) ? xc : nx;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; j++)
        if (lane + 64 * j < w) xs[lane + 64 * j] = xr[j];
    }
    __syncthreads();
  }
  if (tid < w) x[tk.fcol + tid] = xs[tid];
}

// panel row -> global row of every cblk, tabulated once (the solves' gather / scatter index)
__global__ __launch_bounds__(256) void k_solve_rowidx(const SolveTask* __restrict__ tasks,
                                                      const int64_t* __restrict__ roff,
                                                      const DevBlok* __restrict__ bl, int32_t* __restrict__ ridx) {
  const SolveTask tk = tasks[blockIdx.x];
  int32_t* out = ridx + roff[blockIdx.x];
  for (int b = tk.fblok + (threadIdx.x >> 5); b < tk.lblok; b += 8) {
    const DevBlok bk = bl[b];
    for (int i = threadIdx.x & 31; i <= bk.lrow - bk.frow; i += 32) out[bk.coefind + i] = bk.frow + i;
  }
}
void launch_solve_rowidx(hipStream_t s, const SolveTask* tasks, int64_t ntask, const int64_t* roff,
                         const DevBlok* bl, int32_t* ridx) {
  if (ntask > 0) hipLaunchKernelGGL(k_solve_rowidx, dim3((unsigned)ntask), dim3(256), 0, s, tasks, roff, bl, ridx);
}

// forward, step 2 / backward, step 1 on chunks of panel rows, NR right-hand sides per pass over the panel (x is
// n x NR, leading dimension ldx): lane = row, the four waves split the columns in groups (independent coalesced
// loads in flight, > 2 workgroups per CU on the tall top panels).
template <int NR, class T>
__global__ __launch_bounds__(256) void k_solve_off_fwd64(const T* __restrict__ L,
                                                         const SolveChunk* __restrict__ chunks,
                                                         const int32_t* __restrict__ ridx, double* __restrict__ x,
                                                         int64_t ldx) {
  __shared__ double xs[NR][MAXW];
  __shared__ double part[4][NR][64];
  const SolveChunk ck = chunks[blockIdx.x];
  const T* A = L + ck.off;
  const int ld = ck.stride, w = ck.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NR * MAXW; i += 256) {
    const int q = i / MAXW, c = i - q * MAXW;
    // (the columns beyond the cblk's width are multiplied by zeros below: they must not hold what another kernel left in
    // LDS -- integer keys read as doubles are NaNs, and 0 x NaN poisoned the right-hand sides of a multi-vector solve)
    xs[q][c] = c < w ? x[q * ldx + ck.fcol + c] : 0.0;
  }
  __syncthreads();
  const int p = ck.row0 + lane;
  const T* Ap = A + min(p, ld - 1);
  double sacc[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) sacc[k] = 0.0;
  for (int c0 = wave * 32; c0 < w; c0 += 128) {
    double a[32];
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = Ap[(int64_t)min(c0 + i, w - 1) * ld];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const double m = (c0 + i < w) ? a[i] : 0.0;
#pragma unroll
      for (int k = 0; k < NR; k++) sacc[k] = __builtin_fma(m, xs[k][min(c0 + i, MAXW - 1)], sacc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NR; k++) part[wave][k][lane] = sacc[k];
  __syncthreads();
  if (wave == 0 && lane < ck.nrows) {
    const int64_t gr = ridx[ck.roff + p];
#pragma unroll
    for (int k = 0; k < NR; k++)
      unsafeAtomicAdd(&x[k * ldx + gr], -(part[0][k][lane] + part[1][k][lane] + part[2][k][lane] + part[3][k][lane]));
  }
}

// backward: the sum over the 64 rows of a wave for G = 32/NR columns at a time is a *transposed butterfly*: at every
// step a lane hands half of its partial sums to its partner and keeps the other half (G-1 shuffles for G columns
// instead of 6G), then the 64/G lanes that hold the same column finish with plain steps.
template <int NR, class T>
__global__ __launch_bounds__(256) void k_solve_off_bwd64(const T* __restrict__ L,
                                                         const SolveChunk* __restrict__ chunks,
                                                         const int32_t* __restrict__ ridx, double* __restrict__ x,
                                                         int64_t ldx) {
  constexpr int G = 32 / NR;                 // columns per group
  constexpr int LPC = 64 / G;                // lanes that end with the same column
  const SolveChunk ck = chunks[blockIdx.x];
  const T* A = L + ck.off;
  const int ld = ck.stride, w = ck.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave * G >= w) return;
  const int p = ck.row0 + lane;
  for (int c0 = wave * G; c0 < w; c0 += 4 * G) {
    double acc[NR][G];
#pragma unroll
    for (int k = 0; k < NR; k++)
#pragma unroll
      for (int i = 0; i < G; i++) acc[k][i] = 0.0;
    for (int rb = 0; rb < ck.nrows; rb += 64) {          // the chunk's rows, 64 at a time
      const int pp = min(p + rb, ld - 1);
      const bool rv = lane + rb < ck.nrows;
      const int64_t gr = ridx[ck.roff + pp];
      double xr[NR];
#pragma unroll
      for (int k = 0; k < NR; k++) xr[k] = rv ? x[k * ldx + gr] : 0.0;
      const T* Ap = A + pp;
      double a[G];
#pragma unroll
      for (int i = 0; i < G; i++) a[i] = Ap[(int64_t)min(c0 + i, w - 1) * ld];
#pragma unroll
      for (int i = 0; i < G; i++)
#pragma unroll
        for (int k = 0; k < NR; k++) acc[k][i] = __builtin_fma(a[i], xr[k], acc[k][i]);
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
#pragma unroll
      for (int n = G, d = 32; n > 1; n >>= 1, d >>= 1) {
        const int half = n >> 1;
        const bool up = (lane & d) != 0;
#pragma unroll
        for (int i = 0; i < half; i++) {
          const double send = up ? acc[k][i] : acc[k][i + half];
          const double keep = up ? acc[k][i + half] : acc[k][i];
          acc[k][i] = keep + __shfl_xor(send, d);
        }
      }
#pragma unroll
      for (int d = LPC >> 1; d >= 1; d >>= 1) acc[k][0] += __shfl_xor(acc[k][0], d);
    }
    const int c = c0 + ((lane / LPC) & (G - 1));
    if ((lane & (LPC - 1)) == 0 && c < w) {
#pragma unroll
      for (int k = 0; k < NR; k++) unsafeAtomicAdd(&x[k * ldx + ck.fcol + c], -acc[k][0]);
    }
  }
}

// LDLt: x_k := D_k^-1 x_k between the forward and the backward sweep
template <class T>
__global__ __launch_bounds__(256) void k_solve_dscale(const T* __restrict__ L,
                                                      const SolveTask* __restrict__ tasks, double* __restrict__ x) {
  const SolveTask tk = tasks[blockIdx.x];
  const T* A = L + tk.off;
  for (int c = threadIdx.x; c < tk.width; c += 256) x[tk.fcol + c] /= A[c + (int64_t)c * tk.stride];
}

// ------------------------------------------------------------------------------------------------
// Thin levels of the solve (a handful of cblks per level: the separator chains at the top of the tree -- 200^3: 317
// levels of one cblk, 157 of two): a level there is pure dependent latency -- a 128-step substitution chain in the
// diagonal blok (7-12 us), a panel kernel of a few MB, two kernel boundaries -- and there are hundreds of them per
// sweep.  For these cblks the inverses of the diagonal bloks are formed once per factorization (k_solve_inv), the
// substitution becomes a 128 x 128 matrix-vector product every workgroup of the level does for itself, and a level is
// ONE launch per sweep:
//   forward : every workgroup of cblk k reads b_k, forms x_k = L_kk^-1 b_k, then subtracts its 64 panel rows times
//             x_k from the rows below; the workgroup that draws the last ticket of the cblk -- by then every other
//             one has read b_k -- writes x_k in place;
//   backward: every workgroup subtracts its 256 panel rows' part of L_panel^T x_below from b_k (atomics), fences, draws
//             a ticket; the last one of the cblk reads the complete b_k back and writes x_k = L_kk^-T b_k.
// The reference's counterpart is the per-cblk TRSV + GEMV of up_down_smp (updo.c:114-1608).
// ------------------------------------------------------------------------------------------------
constexpr int INVLD = 128;          // leading dimension of a stored inverse

// One workgroup per (thin cblk, which): the inverse M of a lower-triangular w x w matrix T, by rows:
// M[i, :] = (e_i - sum_{k<i} T[i,k] M[k, :]) / T[i,i]; thread j owns column j, M lives in dynamic LDS (w x (w|1)).
//   which 0: T = the lower triangle of the diagonal blok in arena A (unit diagonal if `unit`);   out = M      [r + c ld]
//   which 1: the same T,                                                                        out = M^T    [c + r ld]
//   which 2: T = U^T, U the upper triangle of the blok (LU: non-unit),                          out = M^T = U^-1 stored
//            as [c + r ld] of M, i.e. x = U^-1 b is the row-wise product the backward kernel does
template <class TE>
__global__ __launch_bounds__(128) void k_solve_inv(const TE* __restrict__ A, const SolveTask* __restrict__ tasks,
                                                   const int32_t* __restrict__ thin_tasks, double* __restrict__ inv,
                                                   int which, int unit) {
  extern __shared__ double S[];                     // M: [i + j * ldl], then the row buffer
  const SolveTask tk = tasks[thin_tasks[blockIdx.x]];
  const TE* T = A + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width, j = threadIdx.x, ldl = w | 1;
  double* row = S + (size_t)ldl * w;
  for (int i = 0; i < w; i++) {
    // row i of T: T[i, k], k <= i
    if (j <= i && j < w) row[j] = which == 2 ? T[j + (int64_t)i * ld] : T[i + (int64_t)j * ld];
    __syncthreads();
    if (j < w) {
      double sum = 0.0;
      for (int k = j; k < i; k++) sum = __builtin_fma(row[k], S[k + j * ldl], sum);
      const double d = unit ? 1.0 : row[i];
      S[i + j * ldl] = j <= i ? ((i == j ? 1.0 : 0.0) - sum) / d : 0.0;
    }
    __syncthreads();
  }
  double* out = inv + (int64_t)tk.thin * INVLD * INVLD;
  for (int i = 0; i < w; i++)
    if (j < w) out[which == 0 ? i + j * INVLD : j + i * INVLD] = S[i + j * ldl];      // M, or M^T
}

// y = M x for the w x w matrix M stored [r + c INVLD] (zeros outside its triangle and beyond w), 256 threads: thread t ->
// row t & 127 and one half of the 128 columns, 32 loads in flight at a time; xs (zero beyond w), ys: LDS vectors of
// 128, tmp of 256
__device__ __forceinline__ void inv_apply(const double* __restrict__ M, const double* xs, double* ys, double* tmp, int tid) {
  const int r = tid & 127, h = tid >> 7;
  const double* Mr = M + r + (int64_t)(64 * h) * INVLD;
  double acc = 0.0;
#pragma unroll
  for (int c0 = 0; c0 < 64; c0 += 32) {
    double m[32];
#pragma unroll
    for (int i = 0; i < 32; i++) m[i] = Mr[(int64_t)(c0 + i) * INVLD];
#pragma unroll
    for (int i = 0; i < 32; i++) acc = __builtin_fma(m[i], xs[64 * h + c0 + i], acc);
  }
  tmp[tid] = acc;
  __syncthreads();
  if (tid < 128) ys[tid] = tmp[tid] + tmp[tid + 128];
  __syncthreads();
}

// The diagonal-blok solve of the levels below the thin ones (tens of thousands of cblks per level), one WAVE per cblk, four
// per workgroup: lane = row, the row's entries in registers (coalesced column reads),
// x on the lanes, the pivot travels by v_readlane.  MODE 0: forward, lower (unit: unit diagonal).  MODE 2: backward,
// upper (LU).  MODE 1: backward with L^T (LLt / LDLt) -- the lane that holds row j holds L[j][c] for every c < j, so
// x_c = (b_c - sum_{j > c} L[j][c] x_j) / d_c is a sum over the lanes: a wave reduction per column instead of a
// transposed copy of the blok.  k_solve_diag_q1 (four waves and an LDS hand-over per cblk) took 0.75 / 1.75 ms per
// sweep for the 42 k leaf cblks of 200^3, whose diagonal bloks are 0.76 GB.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// RP = rows per lane: 1 for cblks of at most 64 columns, 2 up to 128 (rows lane and lane + 64).
template <int MODE, int RP, class T>
__global__ __launch_bounds__(256) void k_solve_diag_n64(const T* __restrict__ L, const SolveTask* __restrict__ tasks,
                                                       int64_t ntask, double* __restrict__ x, int unit) {
  const int lane = threadIdx.x & 63;
  const int64_t ti = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ti >= ntask) return;
  const SolveTask tk = tasks[ti];
  const T* A = L + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width;
  int64_t rc[RP];
  double rinv[RP], xr[RP];
#pragma unroll
  for (int j = 0; j < RP; j++) {
    rc[j] = min(lane + 64 * j, w - 1);
    rinv[j] = unit ? 1.0 : 1.0 / (double)A[rc[j] + rc[j] * ld];
    xr[j] = lane + 64 * j < w ? x[tk.fcol + lane + 64 * j] : 0.0;
  }
  // (32 columns at a time: 64 registers per row instead of 128 -- the waves of a SIMD hide the later round trips)
#pragma unroll
  for (int h = 0; h < 2 * RP; h++) {
    if (32 * h < w) {                                // (wave-uniform)
      double a[RP][32];
#pragma unroll
      for (int i = 0; i < 32; i++) {
        const int g = 32 * h + i;
        const int64_t c = min(max(MODE == 0 ? g : w - 1 - g, 0), w - 1);
#pragma unroll
        for (int j = 0; j < RP; j++) a[j][i] = A[rc[j] + c * ld];
      }
#pragma unroll
      for (int i = 0; i < 32; i++) {
        const int g = 32 * h + i;
        if (g < w) {                                 // (wave-uniform)
          const int c = MODE == 0 ? g : w - 1 - g;
          const int slot = c >> 6, src = c & 63;
          if (MODE == 1) {
            double pr = 0.0;
#pragma unroll
            for (int j = 0; j < RP; j++) {
              const int r = lane + 64 * j;
              pr += (r > c && r < w) ? a[j][i] * xr[j] : 0.0;
            }
            const double sm = wave_sum_f64(pr);
#pragma unroll
            for (int j = 0; j < RP; j++)
              if (lane + 64 * j == c) xr[j] = (xr[j] - sm) * rinv[j];
          } else {
            const double v = (RP == 2 && slot) ? xr[RP - 1] * rinv[RP - 1] : xr[0] * rinv[0];
            const double xc = readlane_f64(v, src);
#pragma unroll
            for (int j = 0; j < RP; j++) {
              const int r = lane + 64 * j;
              const bool upd = MODE == 0 ? (r > c && r < w) : (r < c);
              xr[j] = r == c ? xc : upd ? __builtin_fma(-a[j][i], xc, xr[j]) : xr[j];
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RP; j++)
    if (lane + 64 * j < w) x[tk.fcol + lane + 64 * j] = xr[j];
}

// ---- runs of consecutive thin levels in ONE launch ------------------------------------------------------------------
// The workgroups of a run are listed level after level in sweep order and synchronise cblk by cblk.  Forward: the
// workgroups of cblk k start on b_k when the last chunk that contributes to it inside the run has counted itself
// (cnt[k] reaching expect[k] raises k's flag); a chunk counts itself at every thin cblk its rows face once its
// contributions are acknowledged.  Backward: a chunk waits for the flags of the cblks its rows face, raised by the
// cblk's solver -- its workgroup without rows, which waits for the cblk's chunks on a ticket.  A workgroup only ever waits for workgroups of earlier levels -- smaller
// block indices, which the dispatcher has started before it (a 1-D grid is dispatched in index order on every XCD) --
// so the lowest unfinished workgroup always runs: no deadlock.
// What it gains over a launch per level: no drain / launch / ramp per level (753 levels at 200^3); the waiting
// workgroups hold the first part of their panel rows and of the inverse in registers; and the far rows of a tall panel
// no longer stand between two consecutive diagonal solves -- only the chunk that faces the next cblk does.
// Visibility without fences (MI355X_MICROARCH, inter-workgroup visibility): contributions are agent-scope atomics,
// performed at the memory side; a workgroup waits for its own to be acknowledged (vmcnt(0)) before it counts itself;
// flags and right-hand sides are read with agent-scope (sc1) loads that pass L1 and the XCD's L2, solutions are written
// through with agent-scope stores.  Polled words are never the counted ones (increments do not queue behind polls), a
// flag has FLAG_REP copies 256 B apart and a workgroup polls copy blockIdx % FLAG_REP.  The poll is bounded: after
// SPIN_LIMIT polls (seconds) the workgroup raises *stuck and goes on, the host returns PASTIX_AMD_ERR_DEVICE -- a wrong
// assumption fails the solve, it cannot hang the device.
constexpr int SPIN_LIMIT = 1 << 22;
constexpr int FLAG_PAD = 64;       // ints between two polled words
constexpr int FLAG_REP = 8;
__device__ __forceinline__ void thin_poll(const int* flag, const int t, int* stuck) {
  const int* f = flag + ((int64_t)t * FLAG_REP + blockIdx.x % FLAG_REP) * FLAG_PAD;
  int it = 0;
  while (!__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (++it > SPIN_LIMIT) { __hip_atomic_store(stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    // (once ANY workgroup has given up the solve has failed: the others leave at once instead of spinning out their own
    // limits one after the other down the chain of levels)
    if ((it & 1023) == 0 && __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (it < 64) __builtin_amdgcn_s_sleep(8);                  // (~0.2 us, later ~1.3 us between polls)
    else __builtin_amdgcn_s_sleep(48);
  }
}
__device__ __forceinline__ void thin_raise(int* flag, const int t) {
#pragma unroll
  for (int r = 0; r < FLAG_REP; r++)
    __hip_atomic_store(flag + ((int64_t)t * FLAG_REP + r) * FLAG_PAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// forward: 256 panel rows per workgroup (one per thread); every workgroup applies the inverse for itself
template <class T>
__global__ __launch_bounds__(256) void k_solve_thin_fwd(const T* __restrict__ L, const SolveChunk* __restrict__ chunks,
                                                        const int32_t* __restrict__ ridx, const double* __restrict__ inv,
                                                        int* __restrict__ ticket, const int32_t* __restrict__ tgt,
                                                        const int32_t* __restrict__ expect, int* __restrict__ cnt,
                                                        int* __restrict__ flag, int* __restrict__ stuck,
                                                        double* __restrict__ x) {
  __shared__ double xs[128], ys[128], tmp[256];
  __shared__ int last;
  const SolveChunk ck = chunks[blockIdx.x];
  const int w = ck.width, tid = threadIdx.x;
  // the thread's panel row, first 32 columns: in flight while the levels in front finish
  const int ld = ck.stride;
  const bool rowv = tid < ck.nrows;
  const int p = ck.row0 + min(tid, max(ck.nrows - 1, 0));
  const T* Ap = L + ck.off + p;
  double a0[32];
#pragma unroll
  for (int i = 0; i < 32; i++) a0[i] = rowv ? Ap[(int64_t)min(i, w - 1) * ld] : 0.0;
  const int32_t gr = rowv ? ridx[ck.roff + p] : 0;
  // ... and the thread's part of the inverse
  const double* Mr = inv + (int64_t)ck.thin * INVLD * INVLD + (tid & 127) + (int64_t)(64 * (tid >> 7)) * INVLD;
  double m0[32], m1[32];
#pragma unroll
  for (int i = 0; i < 32; i++) m0[i] = Mr[(int64_t)i * INVLD];
#pragma unroll
  for (int i = 0; i < 32; i++) m1[i] = Mr[(int64_t)(32 + i) * INVLD];
  if (ck.wait) {
    if (tid == 0) thin_poll(flag, ck.thin, stuck);
    __syncthreads();
  }
  if (tid < 128) xs[tid] = tid < w ? __hip_atomic_load(&x[ck.fcol + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  __syncthreads();
  // (every read of b_k by this workgroup is complete: the ticket may be drawn)
  if (tid == 0) last = atomicAdd(&ticket[ck.thin], 1) == ck.nwg - 1;
  {
    const int h = tid >> 7;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i++) acc = __builtin_fma(m0[i], xs[64 * h + i], acc);
#pragma unroll
    for (int i = 0; i < 32; i++) acc = __builtin_fma(m1[i], xs[64 * h + 32 + i], acc);
    tmp[tid] = acc;
    __syncthreads();
    if (tid < 128) ys[tid] = tmp[tid] + tmp[tid + 128];
    __syncthreads();
  }
  // (read again by the backward sweep only, another launch; written through so that no line of x is ever dirty in an L2)
  if (last && tid < w) __hip_atomic_store(&x[ck.fcol + tid], ys[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rowv) {
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i++) sacc = __builtin_fma((i < w) ? a0[i] : 0.0, ys[i], sacc);
    for (int c0 = 32; c0 < w; c0 += 32) {
      double a[32];
#pragma unroll
      for (int i = 0; i < 32; i++) a[i] = Ap[(int64_t)min(c0 + i, w - 1) * ld];
#pragma unroll
      for (int i = 0; i < 32; i++) sacc = __builtin_fma((c0 + i < w) ? a[i] : 0.0, ys[min(c0 + i, 127)], sacc);
    }
    unsafeAtomicAdd(&x[gr], -sacc);
  }
  if (ck.tn > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < ck.tn; i += 256) {
      const int t = tgt[ck.tptr + i];
      if (atomicAdd(&cnt[t], 1) == expect[t] - 1) thin_raise(flag, t);
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void k_solve_thin_bwd(const T* __restrict__ B, const SolveChunk* __restrict__ chunks,
                                                        const int32_t* __restrict__ ridx, const double* __restrict__ invT,
                                                        int* __restrict__ ticket, const int32_t* __restrict__ tgt,
                                                        int* __restrict__ flag, int* __restrict__ stuck,
                                                        double* __restrict__ x) {
  __shared__ double xs[128], tmp[256];
  const SolveChunk ck = chunks[blockIdx.x];
  const int w = ck.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (ck.nrows == 0) {
    // The cblk's SOLVER: its workgroup without panel rows (listed behind the cblk's chunks).  It takes its part of the
    // inverse into registers at once, waits until every chunk of the cblk has counted itself on the ticket (their
    // contributions to b_k are acknowledged by then), and applies the inverse: no load stands between the last
    // contribution and the solution but the right-hand side itself.
    const int h = tid >> 7;
    const double* Mr = invT + (int64_t)ck.thin * INVLD * INVLD + (tid & 127) + (int64_t)(64 * h) * INVLD;
    double m[64];
#pragma unroll
    for (int i = 0; i < 64; i++) m[i] = Mr[(int64_t)i * INVLD];
    if (tid == 0 && ck.nwg > 1) {
      int it = 0;
      while (__hip_atomic_load(&ticket[ck.thin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ck.nwg - 1) {
        if (++it > SPIN_LIMIT) { __hip_atomic_store(stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if ((it & 1023) == 0 && __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (it < 64) __builtin_amdgcn_s_sleep(8);
        else __builtin_amdgcn_s_sleep(48);
      }
    }
    __syncthreads();
    if (tid < 128) xs[tid] = tid < w ? __hip_atomic_load(&x[ck.fcol + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 64; i++) acc = __builtin_fma(m[i], xs[64 * h + i], acc);
    tmp[tid] = acc;
    __syncthreads();
    // (written through to memory: the next levels' workgroups, on any XCD, read it in this launch)
    if (tid < w) __hip_atomic_store(&x[ck.fcol + tid], tmp[tid] + tmp[tid + 128], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) thin_raise(flag, ck.thin);
    return;
  }
  const T* A = B + ck.off;
  const int ld = ck.stride;
  const int p = ck.row0 + lane;
  const int c0 = wave * 32;                                // the wave's 32 columns (w <= 128)
  const bool work = ck.nrows > 0 && c0 < w;
  // the first 64 rows of the wave's columns and the chunk's row indices: in flight while the levels in front finish
  double a[32];
  int32_t gr[4];
  if (work) {
    const T* Ap = A + min(p, ld - 1);
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = Ap[(int64_t)min(c0 + i, w - 1) * ld];
#pragma unroll
    for (int q = 0; q < 4; q++) gr[q] = ridx[ck.roff + min(p + 64 * q, ld - 1)];
  }
  if (ck.tn > 0) {
    for (int i = tid; i < ck.tn; i += 256) thin_poll(flag, tgt[ck.tptr + i], stuck);
    __syncthreads();
  }
  if (work) {
    double xrow[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double xv = __hip_atomic_load(&x[gr[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xrow[q] = lane + 64 * q < ck.nrows ? xv : 0.0;
    }
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = a[i] * xrow[0];
#pragma unroll
    for (int q = 1; q < 4; q++) {                          // the chunk's other rows, 64 at a time
      if (64 * q >= ck.nrows) break;
      const T* Ap = A + min(p + 64 * q, ld - 1);
#pragma unroll
      for (int i = 0; i < 32; i++) a[i] = Ap[(int64_t)min(c0 + i, w - 1) * ld];
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(a[i], xrow[q], acc[i]);
    }
    // transposed butterfly: 32 column sums over the 64 lanes (as k_solve_off_bwd64)
#pragma unroll
    for (int n = 32, d = 32; n > 1; n >>= 1, d >>= 1) {
      const int half = n >> 1;
      const bool up = (lane & d) != 0;
#pragma unroll
      for (int i = 0; i < half; i++) {
        const double send = up ? acc[i] : acc[i + half];
        const double keep = up ? acc[i + half] : acc[i];
        acc[i] = keep + __shfl_xor(send, d);
      }
    }
    acc[0] += __shfl_xor(acc[0], 1);
    const int c = c0 + ((lane >> 1) & 31);
    if (!(lane & 1) && c < w) unsafeAtomicAdd(&x[ck.fcol + c], -acc[0]);
  }
  // the cblk's ticket: every contribution of this chunk has been acknowledged
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) atomicAdd(&ticket[ck.thin], 1);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
// fan-in receive: dst[rows[r] + c*ldd] += src[r + c*nrows]  (recv_handle_fanin, sopalin_sendrecv.c:384-404: the owner
// ADDS the aggregated block; here the block is the sender's compact shadow panel)
__global__ void k_fanin_add(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src,
                            const int32_t* __restrict__ rows, int64_t nrows, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int64_t c = i / nrows, r = i - c * nrows;
    dst[rows[r] + c * ldd] += src[i];
  }
}
void launch_fanin_add(hipStream_t s, double* dst, int64_t ldd, const double* src, const int32_t* rows, int64_t nrows,
                      int64_t ncols) {
  const int64_t total = nrows * ncols;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_fanin_add, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, s, dst,
                     ldd, src, rows, nrows, total);
}

// (cblks are at most 128 columns wide: wider ones are re-cut before planning, api.cpp build_split)
void launch_diag_llt(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                     long long* nbpivot, int* errflag, int maxw) {
  (void)maxw;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_diag_llt_w, dim3((unsigned)n), dim3(512), 0, s, L, tasks, dinv, critere, nbpivot, errflag);
}

void launch_trsm_llt(hipStream_t s, double* L, const TrsmTask* tasks, int64_t n, const double* dinv, int maxw) {
  (void)maxw;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_trsm_llt<8>, dim3((unsigned)n), dim3(256), 0, s, L, tasks, dinv);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device
// ordinal).  A failure is reported and the launch skipped (the error surfaces through hipGetLastError in the caller).
static bool dyn_lds_attr_once(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> g(mu);
  if (done.count({fn, dev})) return true;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    fprintf(stderr, "pastix_amd: hipFuncSetAttribute(dynamic LDS %d B) failed on device %d: %s\n", bytes, dev, hipGetErrorString(e));
    return false;
  }
  done.insert({fn, dev});
  return true;
}

// fwd: L (unit for LDLt/LU).  bwd: LLt/LDLt gather through the L arena, LU through the U arena (U^T panels).
// (cblks are at most 128 columns wide: wider ones are re-cut before planning, api.cpp build_split)
// T: the panels' element type -- double, or float for the factors of the single-precision engine (the vectors stay double)
// nwide: the first nwide tasks of the level are wider than 64 columns (the level's cblks are listed widest first)
template <int MODE, int NR, class T>
static void launch_solve_diag(hipStream_t s, const T* L, const SolveTask* tasks, int64_t ntask, int64_t nwide, double* x,
                              int64_t ldx, int unit, int lvlw) {
  // MODE 1 turns the blok through dynamic LDS sized for the widest cblk of the level
  const size_t smem = MODE == 1 ? (size_t)lvlw * (lvlw | 1) * sizeof(double) : 0;
  if constexpr (NR == 1 || !std::is_same<T, double>::value) {        // one right-hand side: the copy without the systolic loop
    nwide = std::max<int64_t>(0, std::min(nwide, ntask));
    // cblks wider than 64 columns: a wave per cblk where the level has enough of them to fill the chip that way (the leaf
    // levels: 11 k at 200^3), else four waves per cblk (a 128-step chain on one wave is the slower one when it is exposed)
    const bool wide_by_wave = nwide >= 2048;
    if (MODE == 1 && nwide > 0 && !wide_by_wave && !dyn_lds_attr_once((const void*)k_solve_diag_q1<MODE, T>, 128 * 129 * 8)) return;
    for (int k = 0; k < NR; k++) {
      if (nwide > 0 && wide_by_wave)
        hipLaunchKernelGGL((k_solve_diag_n64<MODE, 2, T>), dim3((unsigned)((nwide + 3) / 4)), dim3(256), 0, s, L, tasks, nwide,
                           x + k * ldx, unit);
      else if (nwide > 0)
        hipLaunchKernelGGL((k_solve_diag_q1<MODE, T>), dim3((unsigned)nwide), dim3(256), smem, s, L, tasks, x + k * ldx, unit);
      if (ntask > nwide)
        hipLaunchKernelGGL((k_solve_diag_n64<MODE, 1, T>), dim3((unsigned)((ntask - nwide + 3) / 4)), dim3(256), 0, s, L,
                           tasks + nwide, ntask - nwide, x + k * ldx, unit);
    }
  } else {
    (void)nwide;
    if (MODE == 1 && !dyn_lds_attr_once((const void*)k_solve_diag_q<MODE, NR>, 128 * 129 * 8)) return;
    hipLaunchKernelGGL((k_solve_diag_q<MODE, NR>), dim3((unsigned)ntask), dim3(256), smem, s, L, tasks, x, ldx, unit);
  }
}
// one level of the forward (fwd) or backward sweep for NR right-hand sides (x: n x NR, leading dimension ldx).
// chunks: the 64-row list forward, the 256-row list backward.
template <int NR, class T>
static void solve_level(hipStream_t s, bool fwd, int factotype, const T* L, const T* U,
                        const SolveTask* tasks, int64_t ntask, int64_t nwide, const SolveChunk* chunks, int64_t nchunk,
                        const int32_t* ridx, double* x, int64_t ldx, int lvlw) {
  const int unit = factotype != PASTIX_AMD_FACT_LLT;
  const dim3 gc((unsigned)nchunk);
  if (fwd) {
    if (ntask > 0) launch_solve_diag<0, NR, T>(s, L, tasks, ntask, nwide, x, ldx, unit, lvlw);
    if (nchunk > 0) hipLaunchKernelGGL((k_solve_off_fwd64<NR, T>), gc, dim3(256), 0, s, L, chunks, ridx, x, ldx);
  } else {
    const T* B = factotype == PASTIX_AMD_FACT_LU ? U : L;
    const int mode = factotype == PASTIX_AMD_FACT_LLT ? 0 : factotype == PASTIX_AMD_FACT_LDLT ? 1 : 2;
    if (nchunk > 0) hipLaunchKernelGGL((k_solve_off_bwd64<NR, T>), gc, dim3(256), 0, s, B, chunks, ridx, x, ldx);
    if (ntask > 0) {
      if (mode == 2) launch_solve_diag<2, NR, T>(s, L, tasks, ntask, nwide, x, ldx, 0, lvlw);
      else launch_solve_diag<1, NR, T>(s, L, tasks, ntask, nwide, x, ldx, mode == 1, lvlw);
    }
  }
}
void launch_solve_level(hipStream_t s, bool fwd, int factotype, const double* L, const double* U,
                        const SolveTask* tasks, int64_t ntask, int64_t nwide, const SolveChunk* chunks, int64_t nchunk,
                        const DevBlok* bl, const int32_t* ridx, double* x, int64_t ldx, int nr, int maxw, int lvlw) {
  (void)bl;
  (void)maxw;
  if (nr == 4) solve_level<4, double>(s, fwd, factotype, L, U, tasks, ntask, nwide, chunks, nchunk, ridx, x, ldx, lvlw);
  else if (nr == 2) solve_level<2, double>(s, fwd, factotype, L, U, tasks, ntask, nwide, chunks, nchunk, ridx, x, ldx, lvlw);
  else solve_level<1, double>(s, fwd, factotype, L, U, tasks, ntask, nwide, chunks, nchunk, ridx, x, ldx, lvlw);
}
// the factors of the single-precision engine: float panels, double vectors, one right-hand side per call
void launch_solve_level_s(hipStream_t s, bool fwd, int factotype, const float* L, const float* U, const SolveTask* tasks,
                          int64_t ntask, int64_t nwide, const SolveChunk* chunks, int64_t nchunk, const int32_t* ridx, double* x,
                          int lvlw) {
  solve_level<1, float>(s, fwd, factotype, L, U, tasks, ntask, nwide, chunks, nchunk, ridx, x, 0, lvlw);
}

void launch_solve_dscale(hipStream_t s, const double* L, const SolveTask* tasks, int64_t ntask, double* x) {
  if (ntask > 0) hipLaunchKernelGGL(k_solve_dscale<double>, dim3((unsigned)ntask), dim3(256), 0, s, L, tasks, x);
}

__global__ void k_fill_const(double* __restrict__ dst, int64_t n, double v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = v;
}
void launch_run_diag_lu(hipStream_t sd, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                        double critere, long long* nbpivot, const RunCtl& rc, int* resident, long long limit);   // kernels_var.hip
void launch_run_diag_z(hipStream_t sd, bool herm, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                       double critere, long long* nbpivot, const RunCtl& rc, int* resident, long long limit);                      // kernels_z.hip
void launch_run_panel(hipStream_t sd, int factotype, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                      double critere, long long* nbpivot, int* errflag, const RunCtl& rc, int* resident, long long limit) {
  if (gd <= 0) return;
  if (ar.p[2])                                     // complex double (split planes): LDLt / LDLh
    launch_run_diag_z(sd, factotype == PASTIX_AMD_FACT_LDLH, ar, rd, info, gd, dinv, critere, nbpivot, rc, resident, limit);
  else if (factotype == PASTIX_AMD_FACT_LLT)
    hipLaunchKernelGGL(k_run_diag<0>, dim3((unsigned)gd), dim3(512), 0, sd, ar.p[0], rd, info, dinv, critere, nbpivot, errflag, rc,
                       resident, limit);
  else if (factotype == PASTIX_AMD_FACT_LDLT)
    hipLaunchKernelGGL(k_run_diag<1>, dim3((unsigned)gd), dim3(512), 0, sd, ar.p[0], rd, info, dinv, critere, nbpivot, errflag, rc,
                       resident, limit);
  else
    launch_run_diag_lu(sd, ar, rd, info, gd, dinv, critere, nbpivot, rc, resident, limit);
}

void launch_fill_const(hipStream_t s, double* dst, int64_t n, double v) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_const, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, s, dst, n, v);
}

void launch_scatter(hipStream_t s, double* dst, const int64_t* idx, const double* val, int64_t n) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_scatter, dim3((unsigned)blocks), dim3(256), 0, s, dst, idx, val, n);
}

// thin levels (see k_solve_inv): the inverses of `n` diagonal bloks; A: the arena the triangle is read from (f32: floats)
void launch_solve_inv(hipStream_t s, const void* A, bool f32, const SolveTask* tasks, const int32_t* thin_tasks, int64_t n,
                      double* inv, int which, int unit) {
  if (n <= 0) return;
  const int bytes = (128 * 129 + 128) * (int)sizeof(double);
  if (f32) {
    if (!dyn_lds_attr_once((const void*)k_solve_inv<float>, bytes)) return;
    hipLaunchKernelGGL(k_solve_inv<float>, dim3((unsigned)n), dim3(128), bytes, s, (const float*)A, tasks, thin_tasks, inv, which, unit);
  } else {
    if (!dyn_lds_attr_once((const void*)k_solve_inv<double>, bytes)) return;
    hipLaunchKernelGGL(k_solve_inv<double>, dim3((unsigned)n), dim3(128), bytes, s, (const double*)A, tasks, thin_tasks, inv, which, unit);
  }
}
// one run of thin levels (chunks: its workgroups in sweep order); ticket / cnt / flag zero before the launch
void launch_solve_thin(hipStream_t s, bool fwd, const void* P, bool f32, const SolveChunk* chunks, int64_t nchunk,
                       const int32_t* ridx, const double* inv, int* ticket, const int32_t* tgt, const int32_t* expect, int* cnt,
                       int* flag, int* stuck, double* x) {
  if (nchunk <= 0) return;
  const dim3 g((unsigned)nchunk), b(256);
  if (f32) {
    if (fwd) hipLaunchKernelGGL(k_solve_thin_fwd<float>, g, b, 0, s, (const float*)P, chunks, ridx, inv, ticket, tgt, expect, cnt, flag, stuck, x);
    else hipLaunchKernelGGL(k_solve_thin_bwd<float>, g, b, 0, s, (const float*)P, chunks, ridx, inv, ticket, tgt, flag, stuck, x);
  } else {
    if (fwd) hipLaunchKernelGGL(k_solve_thin_fwd<double>, g, b, 0, s, (const double*)P, chunks, ridx, inv, ticket, tgt, expect, cnt, flag, stuck, x);
    else hipLaunchKernelGGL(k_solve_thin_bwd<double>, g, b, 0, s, (const double*)P, chunks, ridx, inv, ticket, tgt, flag, stuck, x);
  }
}
void launch_solve_dscale_s(hipStream_t s, const float* L, const SolveTask* tasks, int64_t ntask, double* x) {
  if (ntask > 0) hipLaunchKernelGGL(k_solve_dscale<float>, dim3((unsigned)ntask), dim3(256), 0, s, L, tasks, x);
}

}  // namespace pastix_amd
