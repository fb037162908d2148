// kernels_z.hip -- complex double LDLt.  HERM = false: complex-SYMMETRIC (the reference's z `sy` variant: SYR
// is x x^T, TRSM/GEMM use "T", no conjugation anywhere: sopalin_compute.h:549-562, compute_diag.c:223-307,
// compute_trsm.c:84-113).  HERM = true: Hermitian LDL^H (z `he`: zher with the real part of d and a real
// diagonal, TRSM "C", GEMM "N","C": compute_diag.c:326-410, compute_trsm.c:91-95).  Panels are kept as split planes on the device (arena 0/2 = Re/Im of L,
// arena 1/3 = Re/Im of L*D), so the update kernel k_update stays a real fp64 MFMA kernel: every complex
// piece is four real pieces (plan.cpp).  Only the diagonal-blok kernel and the panel solve need complex
// arithmetic; they mirror k_diag_ldlt / k_trsm_var<.,1> of kernels_var.hip.
#include <hip/hip_runtime.h>

#include "plan.h"
#include "devmath.h"
#include "run_sync.h"
#include "diag_body.h"

namespace pastix_amd {


struct cz {
  double re, im;
};
__device__ __forceinline__ cz cmul(cz a, cz b) { return cz{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cz csub(cz a, cz b) { return cz{a.re - b.re, a.im - b.im}; }
template <bool C>
__device__ __forceinline__ cz cj(cz a) { return C ? cz{a.re, -a.im} : a; }
// 1/a on the latency-critical chain of the diagonal-blok kernels: Smith's scaling with the Newton-refined hardware
// reciprocal (devmath.h, <= 2 ulp) instead of three correctly rounded software divisions
__device__ __forceinline__ cz cinv_fast(cz a) {
  if (fabs(a.re) >= fabs(a.im)) {
    const double r = a.im * fast_rcp(a.re), d = fast_rcp(a.re + a.im * r);
    return cz{d, -r * d};
  }
  const double r = a.re * fast_rcp(a.im), d = fast_rcp(a.re * r + a.im);
  return cz{r * d, -d};
}
__device__ __forceinline__ cz cinv(cz a) {
  // 1/a, scaled (Smith) to stay finite for large/small |a|
  if (fabs(a.re) >= fabs(a.im)) {
    const double r = a.im / a.re, d = a.re + a.im * r;
    return cz{1.0 / d, -r / d};
  }
  const double r = a.re / a.im, d = a.re * r + a.im;
  return cz{r / d, -1.0 / d};
}

// Panel solve Y = A L_d^-T (unit lower, complex symmetric), L = Y D^-1; Y^T tiles (re, im) live in MFMA
// accumulators; complex products are four real MFMAs.  w <= 16 NT.
template <int NT, bool HERM>
__global__ __launch_bounds__(256) void k_trsm_zsy(const Arenas ar, const TrsmTask* __restrict__ tasks,
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
  const int64_t xo = tk.off + tk.row0 + min(rloc, tk.nrows - 1);
  const double* Tr = ar.p[0] + tk.off;
  const double* Tim = ar.p[2] + tk.off;
  const double* Ti = dinv_ws + tk.dinv_off;

  d4 yr[NT], yi[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const int64_t o = xo + (int64_t)min(col, w - 1) * ld;
      const double vr = ar.p[0][o], vi = ar.p[2][o];
      yr[ct][q] = (rvalid && col < w) ? vr : 0.0;
      yi[ct][q] = (rvalid && col < w) ? vi : 0.0;
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
          const int64_t o = lic + (int64_t)lc * ld;
          const double tr = (li < w) ? Tr[o] : 0.0, tim = (li < w) ? (HERM ? -Tim[o] : Tim[o]) : 0.0;
          // y[ct] -= t * y[p]
          yr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr, yr[p][q], yr[ct], 0, 0, 0);
          yr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(tim, yi[p][q], yr[ct], 0, 0, 0);
          yi[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr, yi[p][q], yi[ct], 0, 0, 0);
          yi[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tim, yr[p][q], yi[ct], 0, 0, 0);
        }
      }
      d4 nr = d4{0, 0, 0, 0}, ni = d4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double ar_ = Ti[ct * 512 + l15 + 16 * (g + 4 * q)];
        const double ai0 = Ti[ct * 512 + 256 + l15 + 16 * (g + 4 * q)];
        const double ai_ = HERM ? -ai0 : ai0;                  // (conj L)^-1 = conj(L^-1)
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yr[ct][q], nr, 0, 0, 0);
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai_, yi[ct][q], nr, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yi[ct][q], ni, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ai_, yr[ct][q], ni, 0, 0, 0);
      }
      yr[ct] = nr;
      yi[ct] = ni;
    }
  }
  const int64_t so = tk.off + tk.row0 + rloc;
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const int64_t dd = (int64_t)min(col, w - 1) * (ld + 1);
      const cz dinv = cinv(cz{Tr[dd], Tim[dd]});
      if (rvalid && col < w) {
        const int64_t o = so + (int64_t)col * ld;
        const cz y = cz{yr[ct][q], yi[ct][q]};
        const cz l = cmul(y, dinv);
        ar.p[1][o] = y.re;                 // L*D (compute_trsm.c:108-109)
        ar.p[3][o] = y.im;
        ar.p[0][o] = l.re;                 // L   (:110)
        ar.p[2][o] = l.im;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// complex LU without row pivoting (z `ge`: PASTIX_getrf_block + DimTrans, compute_diag.c:432-532,564-567;
// no conjugation anywhere, "N"/"T" only).  Mirrors k_diag_lu / k_trsm_var<.,2|3> of kernels_var.hip on split
// planes: arena 0/2 = Re/Im of L (and of the factored diagonal blok), arena 1/3 = Re/Im of the U panel (stored
// transposed, as a lower panel).
// ------------------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void ztile_inverse(F M, bool unit, int nb, int c, cz (*Ti)[17], double* dst) {
  // column c of the inverse of the lower-triangular 16x16 tile M (identity-padded beyond nb)
  for (int i = 0; i < 16; i++) {
    cz x;
    if (i >= nb || c >= nb) x = cz{(i == c) ? 1.0 : 0.0, 0.0};
    else if (i < c) x = cz{0.0, 0.0};
    else {
      cz s = cz{(i == c) ? 1.0 : 0.0, 0.0};
      for (int p = c; p < i; p++) s = csub(s, cmul(M(i, p), Ti[p][c]));
      x = unit ? s : cmul(s, cinv(M(i, i)));
    }
    Ti[i][c] = x;
  }
  for (int i = 0; i < 16; i++) { dst[i + 16 * c] = Ti[i][c].re; dst[256 + i + 16 * c] = Ti[i][c].im; }
}

template <int XR>
__global__ __launch_bounds__(256) void k_diag_zlu(const Arenas ar, const PanelTask* __restrict__ tasks,
                                                  double* __restrict__ dinv_ws, double critere,
                                                  long long* __restrict__ nbpivot) {
  PANEL_PRIO();
  __shared__ cz Ts[16][17];
  __shared__ cz Lo[16][17];
  __shared__ cz Ti[16][17];
  __shared__ cz Xs[16][XR];    // L rows below the tile   Xs[p][r] = L[r][p]
  __shared__ cz Ys[16][XR];    // U columns right of tile Ys[p][c] = U[p][c]
  const PanelTask tk = tasks[blockIdx.x];
  double* Ar = ar.p[0] + tk.off;
  double* Ai = ar.p[2] + tk.off;
  const int ld = tk.stride, w = tk.width;
  const int tid = threadIdx.x, ti = tid & 15, tc = tid >> 4;
  const int nbk = (w + 15) >> 4;
  int npiv = 0;
  for (int kb = 0; kb < w; kb += 16) {
    const int nb = min(16, w - kb), rem = w - kb - nb;
    if (ti < nb && tc < nb) {
      const int64_t o = (kb + ti) + (int64_t)(kb + tc) * ld;
      Ts[ti][tc] = cz{Ar[o], Ai[o]};
    }
    for (int j = 0; j < nb; j++) {                       // PASTIX_getrf, compute_diag.c:432-469
      __syncthreads();
      cz d = Ts[j][j];
      if (hypot(d.re, d.im) < critere) { d = cz{critere, 0.0}; if (tid == 0) npiv++; }
      const cz inv = cinv(d);
      if (ti < nb && tc < nb) {
        if (ti == j && tc >= j) Lo[j][tc] = (tc == j) ? d : Ts[j][tc];                         // row j of U
        else if (tc == j && ti > j) Lo[ti][j] = cmul(Ts[ti][j], inv);                           // column j of L
        else if (ti > j && tc > j) Ts[ti][tc] = csub(Ts[ti][tc], cmul(cmul(Ts[ti][j], inv), Ts[j][tc]));   // GERU
      }
    }
    __syncthreads();
    if (ti < nb && tc < nb) {
      const int64_t o = (kb + ti) + (int64_t)(kb + tc) * ld;
      Ar[o] = Lo[ti][tc].re;
      Ai[o] = Lo[ti][tc].im;
    }
    if (tid < 16) {
      // inverse of (U tile)^T : lower, non-unit, element (i,p) = U[p][i]
      ztile_inverse([&](int i, int p) { return Lo[p][i]; }, false, nb, tid, Ti,
                    dinv_ws + tk.dinv_off + (int64_t)(kb >> 4) * 512);
    } else if (tid - 16 < rem) {
      // rows below: X = A U_T^-1
      const int rr = tid - 16;
      const int64_t o0 = (kb + nb + rr) + (int64_t)kb * ld;
      cz x[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int64_t o = o0 + (int64_t)min(c, nb - 1) * ld;
        x[c] = cz{Ar[o], Ai[o]};
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nb) {
          cz s = x[c];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < c) s = csub(s, cmul(x[p], Lo[p][c]));
          x[c] = cmul(s, cinv(Lo[c][c]));
        }
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        Xs[c][rr] = (c < nb) ? x[c] : cz{0.0, 0.0};
        if (c < nb) {
          Ar[o0 + (int64_t)c * ld] = x[c].re;
          Ai[o0 + (int64_t)c * ld] = x[c].im;
        }
      }
    }
    __syncthreads();
    if (tid < 16) {
      // inverse of the unit-lower L tile
      ztile_inverse([&](int i, int p) { return Lo[i][p]; }, true, nb, tid, Ti,
                    dinv_ws + tk.dinv_off + (int64_t)(nbk + (kb >> 4)) * 512);
    } else if (tid - 16 < rem) {
      // columns right of the tile: Y = L_T^-1 B  (TRSM "L","L","N","U", compute_diag.c:505-508)
      const int cc = tid - 16;
      const int64_t o0 = kb + (int64_t)(kb + nb + cc) * ld;
      cz y[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int64_t o = o0 + min(r, nb - 1);
        y[r] = cz{Ar[o], Ai[o]};
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        if (r < nb) {
          cz s = y[r];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < r) s = csub(s, cmul(Lo[r][p], y[p]));
          y[r] = s;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        Ys[r][cc] = (r < nb) ? y[r] : cz{0.0, 0.0};
        if (r < nb) {
          Ar[o0 + r] = y[r].re;
          Ai[o0 + r] = y[r].im;
        }
      }
    }
    __syncthreads();
    if (rem > 0) {                                       // A22 -= L21 U12 (full square, compute_diag.c:510-511)
      const int nt = (rem + 1) >> 1;
      const int64_t ob = (kb + nb) + (int64_t)(kb + nb) * ld;
      for (int id = tid; id < nt * nt; id += 256) {
        const int tr = id % nt, tcc = id / nt;
        cz c[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++) c[a][b] = cz{0.0, 0.0};
        for (int p = 0; p < nb; p++) {
          cz xa[2], xb[2];
#pragma unroll
          for (int a = 0; a < 2; a++) {
            xa[a] = Xs[p][min(2 * tr + a, XR - 1)];
            xb[a] = Ys[p][min(2 * tcc + a, XR - 1)];
          }
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
              const cz m = cmul(xa[a], xb[b]);
              c[a][b].re += m.re;
              c[a][b].im += m.im;
            }
        }
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int a = 0; a < 2; a++) {
            const int r = 2 * tr + a, cc = 2 * tcc + b;
            if (r < rem && cc < rem) {
              const int64_t o = ob + r + (int64_t)cc * ld;
              Ar[o] -= c[a][b].re;
              Ai[o] -= c[a][b].im;
            }
          }
      }
    }
    __syncthreads();
  }
  // DimTrans (compute_diag.c:521-532, :564-567): U arena diagonal blok = transpose of the factored blok
  double* Ur = ar.p[1] + tk.off;
  double* Ui = ar.p[3] + tk.off;
  for (int id = tid; id < w * w; id += 256) {
    const int a = id % w, b = id / w;
    Ur[b + (int64_t)a * ld] = Ar[a + (int64_t)b * ld];
    Ui[b + (int64_t)a * ld] = Ai[a + (int64_t)b * ld];
  }
  if (tid == 0 && npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
}

// MODE 2: L panel, T(i,c) = U_d[c][i];  MODE 3: U panel, T = unit-lower L_d.  See k_trsm_var.
template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_trsm_zlu(const Arenas ar, const TrsmTask* __restrict__ tasks,
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
  double* Xr = ar.p[MODE == 3 ? 1 : 0];
  double* Xi = ar.p[MODE == 3 ? 3 : 2];
  const int64_t xo = tk.off + tk.row0 + min(rloc, tk.nrows - 1);
  const double* Tr = ar.p[0] + tk.off;                 // factored diagonal blok (always in the L arenas)
  const double* Tim = ar.p[2] + tk.off;
  const double* Ti = dinv_ws + tk.dinv_off + (MODE == 3 ? (int64_t)nbk * 512 : 0);

  d4 yr[NT], yi[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const int64_t o = xo + (int64_t)min(col, w - 1) * ld;
      const double vr = Xr[o], vi = Xi[o];
      yr[ct][q] = (rvalid && col < w) ? vr : 0.0;
      yi[ct][q] = (rvalid && col < w) ? vi : 0.0;
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
          const int64_t o = (MODE == 2) ? lc + (int64_t)lic * ld : lic + (int64_t)lc * ld;
          const double tr = (li < w) ? Tr[o] : 0.0, tim = (li < w) ? Tim[o] : 0.0;
          yr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr, yr[p][q], yr[ct], 0, 0, 0);
          yr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(tim, yi[p][q], yr[ct], 0, 0, 0);
          yi[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr, yi[p][q], yi[ct], 0, 0, 0);
          yi[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tim, yr[p][q], yi[ct], 0, 0, 0);
        }
      }
      d4 nr = d4{0, 0, 0, 0}, ni = d4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double ar_ = Ti[ct * 512 + l15 + 16 * (g + 4 * q)];
        const double ai_ = Ti[ct * 512 + 256 + l15 + 16 * (g + 4 * q)];
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yr[ct][q], nr, 0, 0, 0);
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai_, yi[ct][q], nr, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yi[ct][q], ni, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ai_, yr[ct][q], ni, 0, 0, 0);
      }
      yr[ct] = nr;
      yi[ct] = ni;
    }
  }
  const int64_t so = tk.off + tk.row0 + rloc;
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      if (rvalid && col < w) {
        Xr[so + (int64_t)col * ld] = yr[ct][q];
        Xi[so + (int64_t)col * ld] = yi[ct][q];
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_zsolve_dscale(const double* __restrict__ Lr, const double* __restrict__ Li,
                                                       const SolveTask* __restrict__ tasks, double* __restrict__ xr,
                                                       double* __restrict__ xi) {
  const SolveTask tk = tasks[blockIdx.x];
  for (int c = threadIdx.x; c < tk.width; c += 256) {
    const int64_t dd = tk.off + c + (int64_t)c * tk.stride;
    const cz v = cmul(cz{xr[tk.fcol + c], xi[tk.fcol + c]}, cinv(cz{Lr[dd], Li[dd]}));
    xr[tk.fcol + c] = v.re;
    xi[tk.fcol + c] = v.im;
  }
}

// ---- second generation (the complex counterparts of k_solve_diag_q1 / k_solve_off_fwd64 / k_solve_off_bwd64 of
// kernels.hip; see there for the organisation) ----
__device__ __forceinline__ double z_readlane(double v, int srclane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// MODE 0: forward, unit lower.  MODE 1: backward with unit L^T (CONJ: L^H).  MODE 2: backward with the upper
// triangle (non-unit, LU).  cblks of at most 128 columns.
template <int MODE, bool CONJ>
__global__ __launch_bounds__(256) void k_zsolve_diag_q1(const double* __restrict__ Lr, const double* __restrict__ Li,
                                                        const SolveTask* __restrict__ tasks,
                                                        double* __restrict__ xre, double* __restrict__ xim) {
  __shared__ double xs[2][128];
  const SolveTask tk = tasks[blockIdx.x];
  const double* Ar = Lr + tk.off;
  const double* Ai = Li + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t rcl[2];
  double ar[32][2], ai[32][2];
  cz rinv[2];
#pragma unroll
  for (int j = 0; j < 2; j++) rcl[j] = min(lane + 64 * j, w - 1);
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int g = 32 * wave + i;
    const int64_t c = min(max(MODE == 0 ? g : w - 1 - g, 0), w - 1);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int64_t o = MODE == 1 ? c + rcl[j] * ld : rcl[j] + c * ld;
      ar[i][j] = Ar[o];
      ai[i][j] = CONJ ? -Ai[o] : Ai[o];
    }
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int64_t dd = rcl[j] + rcl[j] * ld;
    rinv[j] = MODE == 2 ? cinv(cz{Ar[dd], Ai[dd]}) : cz{1.0, 0.0};
  }
  if (tid < w) { xs[0][tid] = xre[tk.fcol + tid]; xs[1][tid] = xim[tk.fcol + tid]; }
  __syncthreads();
  for (int q = 0; q < 4; q++) {
    if (wave == q && 32 * q < w) {
      cz x[2];
#pragma unroll
      for (int j = 0; j < 2; j++) x[j] = cz{xs[0][rcl[j]], xs[1][rcl[j]]};
#pragma unroll
      for (int i = 0; i < 32; i++) {
        const int g = 32 * q + i;
        if (g >= w) break;
        const int c = MODE == 0 ? g : w - 1 - g;
        const int slot = c >> 6, src = c & 63;
        const cz v = slot ? cmul(x[1], rinv[1]) : cmul(x[0], rinv[0]);
        const cz xc = cz{z_readlane(v.re, src), z_readlane(v.im, src)};
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int r = lane + 64 * j;
          const bool upd = MODE == 0 ? (r > c && r < w) : (r < c);
          const cz nx = upd ? csub(x[j], cmul(cz{ar[i][j], ai[i][j]}, xc)) : x[j];
          x[j] = (r == c) ? xc : nx;
        }
      }
#pragma unroll
      for (int j = 0; j < 2; j++)
        if (lane + 64 * j < w) { xs[0][lane + 64 * j] = x[j].re; xs[1][lane + 64 * j] = x[j].im; }
    }
    __syncthreads();
  }
  if (tid < w) { xre[tk.fcol + tid] = xs[0][tid]; xim[tk.fcol + tid] = xs[1][tid]; }
}

// 64 panel rows per workgroup, lane = row, the four waves split the columns in groups of 32
__global__ __launch_bounds__(256) void k_zsolve_off_fwd64(const double* __restrict__ Lr, const double* __restrict__ Li,
                                                          const SolveChunk* __restrict__ chunks,
                                                          const int32_t* __restrict__ ridx, double* __restrict__ xre,
                                                          double* __restrict__ xim) {
  __shared__ double xs[2][MAXW];
  __shared__ double part[2][4][64];
  const SolveChunk ck = chunks[blockIdx.x];
  const int ld = ck.stride, w = ck.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < w; c += 256) { xs[0][c] = xre[ck.fcol + c]; xs[1][c] = xim[ck.fcol + c]; }
  __syncthreads();
  const int p = ck.row0 + lane;
  const double* Ar = Lr + ck.off + min(p, ld - 1);
  const double* Ai = Li + ck.off + min(p, ld - 1);
  double sr = 0.0, si = 0.0;
  for (int c0 = wave * 32; c0 < w; c0 += 128) {
    double a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int64_t o = (int64_t)min(c0 + i, w - 1) * ld;
      a[i] = Ar[o];
      b[i] = Ai[o];
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const bool v = c0 + i < w;
      const double yr = v ? xs[0][min(c0 + i, MAXW - 1)] : 0.0, yi = v ? xs[1][min(c0 + i, MAXW - 1)] : 0.0;
      sr = __builtin_fma(a[i], yr, __builtin_fma(-b[i], yi, sr));
      si = __builtin_fma(a[i], yi, __builtin_fma(b[i], yr, si));
    }
  }
  part[0][wave][lane] = sr;
  part[1][wave][lane] = si;
  __syncthreads();
  if (wave == 0 && lane < ck.nrows) {
    const int64_t gr = ridx[ck.roff + p];
    unsafeAtomicAdd(&xre[gr], -(part[0][0][lane] + part[0][1][lane] + part[0][2][lane] + part[0][3][lane]));
    unsafeAtomicAdd(&xim[gr], -(part[1][0][lane] + part[1][1][lane] + part[1][2][lane] + part[1][3][lane]));
  }
}

// up to 256 rows per workgroup; per wave 16 columns at a time through the transposed butterfly (both planes)
template <bool CONJ>
__global__ __launch_bounds__(256) void k_zsolve_off_bwd64(const double* __restrict__ Br, const double* __restrict__ Bi,
                                                          const SolveChunk* __restrict__ chunks,
                                                          const int32_t* __restrict__ ridx, double* __restrict__ xre,
                                                          double* __restrict__ xim) {
  constexpr int G = 16, LPC = 64 / G;
  const SolveChunk ck = chunks[blockIdx.x];
  const int ld = ck.stride, w = ck.width, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave * G >= w) return;
  const int p = ck.row0 + lane;
  for (int c0 = wave * G; c0 < w; c0 += 4 * G) {
    double accr[G], acci[G];
#pragma unroll
    for (int i = 0; i < G; i++) accr[i] = acci[i] = 0.0;
    for (int rb = 0; rb < ck.nrows; rb += 64) {
      const int pp = min(p + rb, ld - 1);
      const bool rv = lane + rb < ck.nrows;
      const int64_t gr = ridx[ck.roff + pp];
      const double yr = rv ? xre[gr] : 0.0, yi = rv ? xim[gr] : 0.0;
      const double* Ar = Br + ck.off + pp;
      const double* Ai = Bi + ck.off + pp;
      double a[G], b[G];
#pragma unroll
      for (int i = 0; i < G; i++) {
        const int64_t o = (int64_t)min(c0 + i, w - 1) * ld;
        a[i] = Ar[o];
        b[i] = CONJ ? -Ai[o] : Ai[o];
      }
#pragma unroll
      for (int i = 0; i < G; i++) {
        accr[i] = __builtin_fma(a[i], yr, __builtin_fma(-b[i], yi, accr[i]));
        acci[i] = __builtin_fma(a[i], yi, __builtin_fma(b[i], yr, acci[i]));
      }
    }
#pragma unroll
    for (int n = G, d = 32; n > 1; n >>= 1, d >>= 1) {
      const int half = n >> 1;
      const bool up = (lane & d) != 0;
#pragma unroll
      for (int i = 0; i < half; i++) {
        const double sr = up ? accr[i] : accr[i + half], kr = up ? accr[i + half] : accr[i];
        const double si = up ? acci[i] : acci[i + half], ki = up ? acci[i + half] : acci[i];
        accr[i] = kr + __shfl_xor(sr, d);
        acci[i] = ki + __shfl_xor(si, d);
      }
    }
#pragma unroll
    for (int d = LPC >> 1; d >= 1; d >>= 1) {
      accr[0] += __shfl_xor(accr[0], d);
      acci[0] += __shfl_xor(acci[0], d);
    }
    const int c = c0 + ((lane / LPC) & (G - 1));
    if ((lane & (LPC - 1)) == 0 && c < w) {
      unsafeAtomicAdd(&xre[ck.fcol + c], -accr[0]);
      unsafeAtomicAdd(&xim[ck.fcol + c], -acci[0]);
    }
  }
}

// fwd: unit L.  bwd: LDLt/LDLh gather through the L planes (transposed / conjugate-transposed), LU through the U planes.
void launch_zsolve_level(hipStream_t s, bool fwd, int factotype, const Arenas& ar, const SolveTask* tasks, int64_t ntask,
                         const SolveChunk* chunks, int64_t nchunk, const DevBlok* bl, const int32_t* ridx, double* xr,
                         double* xi, int maxw) {
  (void)bl;
  (void)maxw;                              // (cblks are at most 128 columns wide: api.cpp build_split)
  const dim3 b(256), gt((unsigned)ntask), gc((unsigned)nchunk);
  const bool lu = factotype == PASTIX_AMD_FACT_LU, herm = factotype == PASTIX_AMD_FACT_LDLH;
  if (fwd) {
    if (ntask > 0) hipLaunchKernelGGL((k_zsolve_diag_q1<0, false>), gt, b, 0, s, ar.p[0], ar.p[2], tasks, xr, xi);
    if (nchunk > 0) hipLaunchKernelGGL(k_zsolve_off_fwd64, gc, b, 0, s, ar.p[0], ar.p[2], chunks, ridx, xr, xi);
    return;
  }
  const double* Br = lu ? ar.p[1] : ar.p[0];
  const double* Bi = lu ? ar.p[3] : ar.p[2];
  if (nchunk > 0) {
    if (herm) hipLaunchKernelGGL(k_zsolve_off_bwd64<true>, gc, b, 0, s, Br, Bi, chunks, ridx, xr, xi);
    else hipLaunchKernelGGL(k_zsolve_off_bwd64<false>, gc, b, 0, s, Br, Bi, chunks, ridx, xr, xi);
  }
  if (ntask > 0) {
    if (lu) hipLaunchKernelGGL((k_zsolve_diag_q1<2, false>), gt, b, 0, s, ar.p[0], ar.p[2], tasks, xr, xi);
    else if (herm) hipLaunchKernelGGL((k_zsolve_diag_q1<1, true>), gt, b, 0, s, ar.p[0], ar.p[2], tasks, xr, xi);
    else hipLaunchKernelGGL((k_zsolve_diag_q1<1, false>), gt, b, 0, s, ar.p[0], ar.p[2], tasks, xr, xi);
  }
}
void launch_zsolve_dscale(hipStream_t s, const Arenas& ar, const SolveTask* tasks, int64_t ntask, double* xr, double* xi) {
  if (ntask > 0) hipLaunchKernelGGL(k_zsolve_dscale, dim3((unsigned)ntask), dim3(256), 0, s, ar.p[0], ar.p[2], tasks, xr, xi);
}

// interleaved complex <-> split planes
__global__ void k_split(const double* __restrict__ z, double* __restrict__ re, double* __restrict__ im, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { re[i] = z[2 * i]; im[i] = z[2 * i + 1]; }
}
__global__ void k_merge(double* __restrict__ z, const double* __restrict__ re, const double* __restrict__ im, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { z[2 * i] = re[i]; z[2 * i + 1] = im[i]; }
}

template <bool HERM>
__global__ __launch_bounds__(512, 4) void k_diag_zsy_w(const Arenas ar, const PanelTask* __restrict__ tasks,
                                                    double* __restrict__ dinv_ws, double critere,
                                                    long long* __restrict__ nbpivot) {
  PANEL_PRIO();
  __shared__ DiagZLds S;
  const PanelTask tk = tasks[blockIdx.x];
  diag_zsy_body<HERM, false>(S, ar, tk, dinv_ws, critere, nbpivot, threadIdx.x);
}
// the run's diagonal kernel for complex LDLt / LDLh (see k_run_diag, kernels.hip): resident workgroups popping ready tasks
template <bool HERM>
__global__ __launch_bounds__(512, 4) void k_run_diag_z(const Arenas ar, const RunD* __restrict__ rd, const RunInfo* __restrict__ info,
                                                       double* __restrict__ dinv_ws, const double critere,
                                                       long long* __restrict__ nbpivot, const RunCtl rc, int* __restrict__ resident,
                                                       const long long limit) {
  PANEL_PRIO();
  __shared__ DiagZLds S;
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
    int ltid = threadIdx.x;                           // (laundered per task, as in k_run_diag)
    asm volatile("" : "+v"(ltid));
    diag_zsy_body<HERM, true>(S, ar, d.pt, dinv_ws, critere, nbpivot, ltid);
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
void launch_run_diag_z(hipStream_t sd, bool herm, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                       double critere, long long* nbpivot, const RunCtl& rc, int* resident, long long limit) {
  if (gd <= 0) return;
  if (herm) hipLaunchKernelGGL((k_run_diag_z<true>), dim3((unsigned)gd), dim3(512), 0, sd, ar, rd, info, dinv, critere, nbpivot, rc, resident, limit);
  else hipLaunchKernelGGL((k_run_diag_z<false>), dim3((unsigned)gd), dim3(512), 0, sd, ar, rd, info, dinv, critere, nbpivot, rc, resident, limit);
}

// (cblks are at most 128 columns wide -- api.cpp build_split --: one kernel per role)
void launch_diag_zsy(hipStream_t s, bool herm, const Arenas& ar, const PanelTask* tasks, int64_t n, double* dinv,
                     double critere, long long* nbpivot, int maxw) {
  (void)maxw;
  if (n <= 0) return;
  const dim3 g((unsigned)n);
  if (herm) hipLaunchKernelGGL((k_diag_zsy_w<true>), g, dim3(512), 0, s, ar, tasks, dinv, critere, nbpivot);
  else hipLaunchKernelGGL((k_diag_zsy_w<false>), g, dim3(512), 0, s, ar, tasks, dinv, critere, nbpivot);
}
void launch_trsm_zsy(hipStream_t s, bool herm, const Arenas& ar, const TrsmTask* tasks, int64_t n, const double* dinv,
                     int maxw) {
  (void)maxw;
  if (n <= 0) return;
  const dim3 g((unsigned)n), b(256);
  if (herm) hipLaunchKernelGGL((k_trsm_zsy<8, true>), g, b, 0, s, ar, tasks, dinv);
  else hipLaunchKernelGGL((k_trsm_zsy<8, false>), g, b, 0, s, ar, tasks, dinv);
}
void launch_diag_zlu(hipStream_t s, const Arenas& ar, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                     long long* nbpivot, int maxw) {
  (void)maxw;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_diag_zlu<116>, dim3((unsigned)n), dim3(256), 0, s, ar, tasks, dinv, critere, nbpivot);
}
void launch_trsm_zlu(hipStream_t s, const Arenas& ar, const TrsmTask* tasks, int64_t n, const double* dinv, int maxw) {
  (void)maxw;
  if (n <= 0) return;
  const dim3 g((unsigned)n), b(256);
  hipLaunchKernelGGL((k_trsm_zlu<8, 2>), g, b, 0, s, ar, tasks, dinv);
  hipLaunchKernelGGL((k_trsm_zlu<8, 3>), g, b, 0, s, ar, tasks, dinv);
}
void launch_split(hipStream_t s, const double* z, double* re, double* im, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_split, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, z, re, im, n);
}
void launch_merge(hipStream_t s, double* z, const double* re, const double* im, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_merge, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, z, re, im, n);
}

}  // namespace pastix_amd
