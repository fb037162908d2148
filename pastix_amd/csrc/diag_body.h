// diag_body.h -- device-only: the diagonal-blok factorizations (real LLt / LDLt, w <= 128, packed lower triangle resident in
// LDS) as device functions: the bodies of k_diag_llt_w / k_diag_ldlt_w (kernels.hip) and of the diagonal tickets of the run
// launch (kernels_update.hip k_run_update).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.h"
#include "devmath.h"
#include "run_sync.h"

namespace pastix_amd {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// k_diag_llt_w : the diagonal-blok factorization for w <= 128 with the blok resident in LDS (the global-memory
// version above spends ~170 us per 128-wide blok on dependent L2 accesses; this is the latency-critical kernel
// of every level).  Per 16-column block step:
//   (A) wave 0 factors the 16x16 tile in registers, row i in lane i, columns broadcast with v_readlane
//       (no barrier per column), and keeps the rows for (B');
//   (B) waves 1-3: thread-per-row solve of the rows below against the tile (in LDS);
//   (B') wave 0, meanwhile: the tile's inverse for k_trsm, again from registers;
//   (C) all waves: trailing update of the resident blok on the MFMA pipe, 16x16 tiles of the lower part.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// D: lower triangle of the blok, packed by columns (66 KB of LDS): with the 132 KB of a full square the workgroup could
// only start on an EMPTY CU, i.e. never while a k_update launch of the other stream keeps the chip full; this
// size fits beside one k_update workgroup.  Entries outside the w x w part are zero.  Ri: reciprocals of the tile's
// diagonal (two tiles: wave 1 inverts the previous tile while wave 0 factorizes the next one).
// COH: the results are stored write-through (the run launch, run_sync.h).
constexpr int DIAG_LDS_DOUBLES = 128 * 129 / 2;
template <bool COH>
__device__ __forceinline__ void diag_llt_body(double* __restrict__ D, double* __restrict__ Wl, double* __restrict__ L,
                                              const PanelTask& tk, double* __restrict__ dinv_ws, const double critere,
                                              long long* __restrict__ nbpivot, int* __restrict__ errflag, const int tid) {
#define DP(c, r) D[(c) * 128 - (((c) * ((c) + 1)) >> 1) + (r)]
  double* A = L + tk.off;
  const int ld = tk.stride, w = tk.width;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  {
    // blok -> LDS: thread = (row r, column parity); 16 columns per pass, loads issued before the stores.  (16, not 32:
    // the kernel must stay within 128 VGPRs -- a workgroup on the panel stream only gets a slot beside a running bulk
    // launch if its waves fit the 128-register holes a retiring k_update workgroup leaves)
    // 512 threads (eight waves: with 66 KB of LDS two such workgroups per CU are four waves per SIMD, which is what lets
    // the compiler honour the 128-register bound)
    const int r = tid & 127, ch = tid >> 7;               // ch = 0..3
    const int w16 = (w + 15) & ~15;                       // (nothing reads LDS rows or columns beyond the last 16-band of w:
                                                          // narrow cblks -- the leaf levels have thousands -- skip the rest)
    for (int c0 = 0; c0 < w16; c0 += 64) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {                     // loads from clamped addresses
        // (every element is DEFINED in every pass: a partly written array became a 32-register value carried around the
        // ticket loop of k_run_update, spilled at every pop and reloaded in the update tickets' epilogue)
        v[q] = 0.0;
        if (c0 + 4 * q < w16) {
          const int c = min(c0 + ch + 4 * q, w - 1);
          v[q] = A[min(r, w - 1) + (int64_t)c * ld];
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        if (c0 + 4 * q >= w16) break;
        const int c = c0 + ch + 4 * q;
        if (c <= r && r < w16) DP(c, r) = (r < w) ? v[q] : 0.0;
      }
    }
  }
#ifdef DIAG_PROFILE
  long long st[6] = {0, 0, 0, 0, 0, 0};
  long long t_prev = __builtin_readcyclecounter();
#define STAMP(i) { __syncthreads(); long long t_now = __builtin_readcyclecounter(); st[i] += t_now - t_prev; t_prev = t_now; }
#else
#define STAMP(i)
#endif
  STAMP(0)
  int npiv = 0;
  bool bad = false;
  const double cmin = fmax(critere, 2.2250738585072014e-308);   // pivots >= cmin take the short path
  // One 16 x 16 tile of the trailing update A22 -= X X^T (SYRK "L","N", compute_diag.c:197-200), X = columns kb .. kb+15:
  // MFMA "i" = column, "j" = row as in k_update; rows beyond w are zeros.
  auto syrk_tile = [&](const int kb, const int rb, const int cb) {
    d4 c;
    const int row = rb + l15;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = cb + g + 4 * q;
      c[q] = (row >= col) ? DP(col, row) : 0.0;              // (diagonal tiles: the upper part is not stored)
    }
    double xc[4], xr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      xc[ks] = DP(kb + 4 * ks + g, cb + l15);
      xr[ks] = DP(kb + 4 * ks + g, rb + l15);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-xc[ks], xr[ks], c, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = cb + g + 4 * q;
      if (row >= col) DP(col, row) = c[q];
    }
  };
  // Per 16-column step kb:
  //   (S1) wave 0 factorizes the 16 x 16 tile in registers, held as the MFMA accumulator holds it -- lane (l15, g),
  //        register q = entry (row l15, column g + 4q) -- so that the rank-1 update of column j, a(r, c) -= l(r, j) l(c, j),
  //        is ONE v_mfma_f64_16x16x4: both operands are column j itself, which sits in register j / 4 of lane group j % 4,
  //        exactly where k-slice j % 4 of an operand is read (the other three slices are zeros).  No barrier, no LDS and no
  //        scalar broadcast but the pivot's inside the tile (PASTIX_potrf, compute_diag.c:124-153: the same products
  //        subtracted in the same order).  Beside it, off the chain, the transpose W of the tile's inverse (W = I; column
  //        j scaled; W(:, i) -= W(:, j) l(i, j)): a second MFMA per column.  W goes to k_trsm AND to LDS for (S2).
  //        Meanwhile waves 1-7 finish the PREVIOUS step's trailing update (S3b: the tiles right of its first column band).
  //   (S2) the rows below the tile, 16 per wave: X = A21 W (TRSM "R","L","T","N", compute_diag.c:191-195, as the product
  //        with the tile's inverse -- what k_trsm does with every blok below): 4 MFMAs.
  //   (S3a) the first column band of the trailing update (what the next tile and its rows need), one tile per wave.
  for (int kb = 0; kb < w; kb += 16) {
    const int nb = min(16, w - kb), rem = w - kb - nb;
    __syncthreads();
    if (wave == 0) {
      // (the wave issues in order and every instruction of a column sits between two pivots: the tile and the inverse are
      // held NEGATED -- S = -T, V = -W -- so that the operands of the accumulating MFMA, -l(:, j) = S(:, j) / sqrt(d), need no
      // second, negated copy, and the diagonal lane of the scaled column is -sqrt(d) itself)
      d4 S, V;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        S[q] = (g + 4 * q <= l15) ? -DP(kb + g + 4 * q, kb + l15) : 0.0;
        V[q] = (g + 4 * q == l15) ? -1.0 : 0.0;
      }
      unroll_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 2, gj = j & 3;
        if (j < nb) {
          // the chain of a column: pivot -> 1/sqrt -> scaled column -> MFMA; the inverse's column is issued behind the MFMA
          const bool ing = (g == gj);
          double d = -readlane_f64(S[qj], j + 16 * gj);
          double y = __builtin_amdgcn_rsq(d);
          if (__builtin_expect(!(d >= cmin), 0)) {             // |d| < critere, d <= 0 or NaN: compute_diag.c:133-137
            if (fabs(d) < critere) { d = critere; npiv++; }
            if (!(d > 0.0)) bad = true;
            y = __builtin_amdgcn_rsq(d);
            S[qj] = (ing && l15 == j) ? -d : S[qj];
          }
          y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
          y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
          const double sm = S[qj] * y;                         // -l(:, j); on the diagonal lane -d / sqrt(d)
          const double xm = (ing && l15 > j) ? sm : 0.0;       // below the diagonal, as the MFMA operand (both sides)
          S[qj] = (ing && l15 >= j) ? sm : S[qj];              // (x(c) = 0 for c <= j: the MFMA leaves column j and the rows
          if (j < 15) S = __builtin_amdgcn_mfma_f64_16x16x4f64(xm, xm, S, 0, 0, 0);       // above it alone)
          __builtin_amdgcn_sched_barrier(0);
          const double vm = ing ? V[qj] * y : 0.0;
          V[qj] = ing ? vm : V[qj];
          if (j < 15) V = __builtin_amdgcn_mfma_f64_16x16x4f64(xm, vm, V, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      double* dst = dinv_ws + tk.dinv_off + (int64_t)(kb >> 4) * 256;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        if (c <= l15 && c < nb) DP(kb + c, kb + l15) = -S[q];
        Wl[l15 * 16 + c] = -V[q];                              // W(k = l15, i = c) = inv(tile)(i, k)
        pst<COH>(&dst[c + 16 * l15], -V[q]);
      }
    } else if (kb > 0) {
      const int remp = w - kb, nbd = (remp + 15) >> 4;       // (S3b) of step kb - 16: bands bj >= 1
      const int ntile = nbd * (nbd - 1) / 2;
      for (int t = wave - 1; t < ntile; t += 7) {
        int bj = 1, rest = t;                                  // t -> (bi >= bj >= 1): column band bj holds nbd - bj tiles
        while (rest >= nbd - bj) { rest -= nbd - bj; bj++; }
        syrk_tile(kb - 16, kb + (bj + rest) * 16, kb + bj * 16);
      }
    }
    STAMP(1)
    __syncthreads();
    if (wave * 16 < rem) {
      const int rbase = kb + 16 + 16 * wave;
      d4 X = {0.0, 0.0, 0.0, 0.0};
      double wa[4], ar[4];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        wa[ks] = Wl[(4 * ks + g) * 16 + l15];                  // inv(tile)(c = l15, k = 4 ks + g)
        ar[ks] = DP(kb + 4 * ks + g, rbase + l15);             // A21(r = l15, k)
      }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[ks], ar[ks], X, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (g + 4 * q < nb) DP(kb + g + 4 * q, rbase + l15) = X[q];
    }
    STAMP(2)
    __syncthreads();
    if (wave * 16 < rem) syrk_tile(kb, kb + nb + wave * 16, kb + nb);   // (S3a)
    STAMP(3)
  }
  STAMP(4)
  __syncthreads();
  {
    const int r = tid & 127, ch = tid >> 7;
    for (int c = ch; c < w; c += 4)
      if (r < w && r >= c) pst<COH>(&A[r + (int64_t)c * ld], DP(c, r));
  }
  STAMP(5)
#ifdef DIAG_PROFILE
  if (tid == 0) for (int i = 0; i < 6; i++) dinv_ws[8192 + i] = (double)st[i];
#endif
  if (wave == 0 && lane == 0) {
    if (npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
    if (bad) atomicOr(errflag, 1);
  }
#undef DP
}

// k_diag_ldlt_w : the same organisation for the LDLt diagonal blok (PASTIX_sytrf_block, compute_diag.c:262-307), w <= 128:
// packed lower triangle resident in LDS, the 16 x 16 tile factorized by wave 0 in registers (unit L, D on the diagonal,
// static-pivot clamp and the count of positive pivots for IPARM_INERTIA), rows below solved thread-per-row, trailing
// update (L D) L^T on the MFMA pipe with L D formed on the fly from L and the tile's diagonal.
// (D: the packed lower triangle in LDS as in diag_llt_body; Ri: reciprocals of the tile's diagonal, Dd: the diagonal itself;
// COH: results stored write-through for the run launch)
template <bool COH>
__device__ __forceinline__ void diag_ldlt_body(double* __restrict__ D, double* __restrict__ S,
                                               double* __restrict__ L, const PanelTask& tk, double* __restrict__ dinv_ws,
                                               const double critere, long long* __restrict__ nbpivot, const int tid) {
#define DP(c, r) D[(c) * 128 - (((c) * ((c) + 1)) >> 1) + (r)]
  double* A = L + tk.off;
  const int ld = tk.stride, w = tk.width;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  {
    // blok -> LDS: thread = (row r, column parity); 16 columns per pass, loads issued before the stores.  (16, not 32:
    // the kernel must stay within 128 VGPRs -- a workgroup on the panel stream only gets a slot beside a running bulk
    // launch if its waves fit the 128-register holes a retiring k_update workgroup leaves)
    // 512 threads (eight waves: with 66 KB of LDS two such workgroups per CU are four waves per SIMD, which is what lets
    // the compiler honour the 128-register bound)
    const int r = tid & 127, ch = tid >> 7;               // ch = 0..3
    const int w16 = (w + 15) & ~15;                       // (nothing reads LDS rows or columns beyond the last 16-band of w:
                                                          // narrow cblks -- the leaf levels have thousands -- skip the rest)
    for (int c0 = 0; c0 < w16; c0 += 64) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {                     // loads from clamped addresses
        // (every element is DEFINED in every pass: a partly written array became a 32-register value carried around the
        // ticket loop of k_run_update, spilled at every pop and reloaded in the update tickets' epilogue)
        v[q] = 0.0;
        if (c0 + 4 * q < w16) {
          const int c = min(c0 + ch + 4 * q, w - 1);
          v[q] = A[min(r, w - 1) + (int64_t)c * ld];
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        if (c0 + 4 * q >= w16) break;
        const int c = c0 + ch + 4 * q;
        if (c <= r && r < w16) DP(c, r) = (r < w) ? v[q] : 0.0;
      }
    }
  }
  int npiv = 0, npos = 0;
  const double cmin = fmax(critere, 2.2250738585072014e-308);   // pivots >= cmin take the short path
  // LDS scratch S: the diagonal of the tile (two halves: steps alternate), its reciprocals (ditto), the tile's inverse
  double* const Dd = S, * const Ri = S + 32, * const Wl = S + 64;
  // one 16 x 16 tile of A22 -= (L D) L^T (GEMM with the L D copy, compute_diag.c:299-304); MFMA "i" = column, "j" = row as
  // in k_update; rows beyond w are zeros; L D is L times the tile's diagonal
  auto gemm_tile = [&](const int kb, const int rb, const int cb) {
    const double* Dk = Dd + ((kb >> 4) & 1) * 16;
    d4 c;
    const int row = rb + l15;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = cb + g + 4 * q;
      c[q] = (row >= col) ? DP(col, row) : 0.0;              // (diagonal tiles: the upper part is not stored)
    }
    double xc[4], xr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      xc[ks] = DP(kb + 4 * ks + g, cb + l15);
      xr[ks] = DP(kb + 4 * ks + g, rb + l15) * Dk[4 * ks + g];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-xc[ks], xr[ks], c, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = cb + g + 4 * q;
      if (row >= col) DP(col, row) = c[q];
    }
  };
  // The organisation of diag_llt_body: (S1) wave 0 factorizes the tile in the accumulator layout, one MFMA per column for
  // a(r, c) -= (L D)(r, j) L(c, j) (PASTIX_sytrf, compute_diag.c:223-242) and one for the transposed inverse of the unit
  // lower tile, while waves 1-7 finish the previous step's trailing update; (S2) rows below: (L D) = A21 W, L = (L D) / d
  // (compute_diag.c:284-298); (S3a) the first column band of the trailing update.  The previous step's diagonal is still
  // read by its (S3b) while wave 0 produces the next one: Dd / Ri alternate between two halves.
  for (int kb = 0; kb < w; kb += 16) {
    const int nb = min(16, w - kb), rem = w - kb - nb;
    __syncthreads();
    if (wave == 0) {
      d4 S, V;                                                 // the tile and the inverse, negated (see diag_llt_body)
      double* const Dk = Dd + ((kb >> 4) & 1) * 16, * const Rk = Ri + ((kb >> 4) & 1) * 16;
      if (lane < 16) { Dk[lane] = 1.0; Rk[lane] = 1.0; }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        S[q] = (g + 4 * q <= l15) ? -DP(kb + g + 4 * q, kb + l15) : 0.0;
        V[q] = (g + 4 * q == l15) ? -1.0 : 0.0;
      }
      unroll_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 2, gj = j & 3;
        if (j < nb) {
          const bool ing = (g == gj), below = ing && l15 > j;
          double d = -readlane_f64(S[qj], j + 16 * gj);
          double y = __builtin_amdgcn_rcp(d);
          if (__builtin_expect(!(d >= cmin), 0)) {             // |d| < critere or d <= 0 (or NaN)
            if (fabs(d) < critere) { d = critere; npiv++; }
            if (d > 0.0) npos++;
            y = __builtin_amdgcn_rcp(d);
            S[qj] = (ing && l15 == j) ? -d : S[qj];
          } else {
            npos++;                                            // inertia (sopalin3d.c:1144-1160)
          }
          y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
          y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
          const double sy = S[qj] * y;                         // -L(:, j): (L D)(i, j) / d
          const double am = below ? sy : 0.0, bm = below ? S[qj] : 0.0;   // -L(c, j) and -(L D)(r, j): a(r, c) -= (L D)(r, j) L(c, j)
          S[qj] = below ? sy : S[qj];                          // unit L below the diagonal, -d stays on it
          if (j < 15) S = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm, S, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (lane == 0) { Dk[j] = d; Rk[j] = y; }
          const double vm = ing ? V[qj] : 0.0;                 // (unit diagonal: the inverse's column is not scaled)
          if (j < 15) V = __builtin_amdgcn_mfma_f64_16x16x4f64(am, vm, V, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      double* dst = dinv_ws + tk.dinv_off + (int64_t)(kb >> 4) * 256;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        if (c <= l15 && c < nb) DP(kb + c, kb + l15) = -S[q];
        Wl[l15 * 16 + c] = -V[q];
        pst<COH>(&dst[c + 16 * l15], -V[q]);
      }
    } else if (kb > 0) {
      const int remp = w - kb, nbd = (remp + 15) >> 4;       // (S3b) of step kb - 16: bands bj >= 1
      const int ntile = nbd * (nbd - 1) / 2;
      for (int t = wave - 1; t < ntile; t += 7) {
        int bj = 1, rest = t;
        while (rest >= nbd - bj) { rest -= nbd - bj; bj++; }
        gemm_tile(kb - 16, kb + (bj + rest) * 16, kb + bj * 16);
      }
    }
    __syncthreads();
    if (wave * 16 < rem) {
      const int rbase = kb + 16 + 16 * wave;
      d4 X = {0.0, 0.0, 0.0, 0.0};
      double wa[4], ar[4];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        wa[ks] = Wl[(4 * ks + g) * 16 + l15];
        ar[ks] = DP(kb + 4 * ks + g, rbase + l15);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[ks], ar[ks], X, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (g + 4 * q < nb) DP(kb + g + 4 * q, rbase + l15) = X[q] * Ri[((kb >> 4) & 1) * 16 + g + 4 * q];   // L = (L D) / d
    }
    __syncthreads();
    if (wave * 16 < rem) gemm_tile(kb, kb + nb + wave * 16, kb + nb);   // (S3a)
  }
  __syncthreads();
  {
    const int r = tid & 127, ch = tid >> 7;
    for (int c = ch; c < w; c += 4)
      if (r < w && r >= c) pst<COH>(&A[r + (int64_t)c * ld], DP(c, r));
  }
  if (wave == 0 && lane == 0) {
    if (npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
    if (npos) atomicAdd((unsigned long long*)nbpivot + 1, (unsigned long long)npos);
  }
}
#undef DP

// ---- a wave's NS resident 16 x 16 tile pairs (diag_lu_body: L / U^T planes; diag_zsy_body: Re / Im planes), in VGPRs --------
// (Round 5 also tried these bodies as tickets of the run launch, inside k_run_update's 64 VGPRs with four of the five pairs
// PARKED in the accumulation registers a[0:63]: every test passed and one LU factorization in ten at 48^3 had wrong entries
// in the last columns of a diagonal blok -- DESIGN.md 9.  LU and complex bloks keep their resident kernel.)
template <int NS>
struct TilePairs {
  d4 p[2][NS];
};
template <int NS, int SL, int PL>
__device__ __forceinline__ void tp_mfma(TilePairs<NS>& t, const double a, const double b) {
  d4& c = t.p[PL][SL];
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
template <int NS, int SL, int PL, int Q>
__device__ __forceinline__ double tp_get(const TilePairs<NS>& t) { return t.p[PL][SL][Q]; }
template <int NS, int SL, int PL, int Q>
__device__ __forceinline__ void tp_set(TilePairs<NS>& t, const double v) { t.p[PL][SL][Q] = v; }

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
  TilePairs<NS> tp;                                    // plane 0: the tile of A, plane 1: the tile of A^T
  int tbi[NS], tbj[NS];
  if (wave > 0) {
    unroll_for<0, NS>([&](auto SLc) {
      constexpr int sl = decltype(SLc)::value;
      int id = (wave - 1) + 7 * sl, bj = 0, cnt = 7;
      if (id >= 14) { id -= 14; bj = 2; cnt = 6; while (id >= cnt) { id -= cnt; bj++; cnt--; } }
      else if (id >= 7) { id -= 7; bj = 1; }
      const int bi = (bj < 2 ? 1 : bj) + id;
      const bool on = bi < nbt;
      tbi[sl] = on ? bi : -1;
      tbj[sl] = bj;
      double lo[4], up[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * bi + l15, col = 16 * bj + g + 4 * q;
        const bool v = on && row < w && col < w;
        const int rc = min(row, w - 1), cc = min(col, w - 1);
        const double a0 = pld<COH>(&A[rc + (int64_t)cc * ld]), a1 = pld<COH>(&A[cc + (int64_t)rc * ld]);
        lo[q] = v ? a0 : 0.0;                                // A(row, col)
        up[q] = v ? a1 : 0.0;                                // A(col, row) = A^T(row, col)
      }
      unroll_for<0, 4>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        tp_set<NS, sl, 0, q>(tp, lo[q]);
        tp_set<NS, sl, 1, q>(tp, up[q]);
      });
      if (on && bj == 0) {                                   // what lies below / right of the first tile goes to LDS at once
#pragma unroll
        for (int q = 0; q < 4; q++) {
          S.Ps[0][g + 4 * q][16 * (bi - 1) + l15] = lo[q];
          S.Ps[1][g + 4 * q][16 * (bi - 1) + l15] = up[q];
        }
      }
    });
  }
  // the trailing update of resident pair SL with the solved rows / columns of step tp_: A(r, c) -= L21(r, :) U12(:, c)
  auto update = [&](auto SLc, const int bi, const int bj, const int tp_) {
    constexpr int sl = decltype(SLc)::value;
    const int ro = 16 * (bi - tp_ - 1) + l15, co = 16 * (bj - tp_ - 1) + l15;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + g;
      const double xr = S.Xs[0][k][ro], xc = S.Xs[0][k][co];                 // L21(r, k), L21(c, k)
      const double yr = S.Xs[1][k][ro], yc = S.Xs[1][k][co];                 // U12(k, r), U12(k, c)
      tp_mfma<NS, sl, 0>(tp, -yc, xr);                                  // A(r, c)   -= L21(r, k) U12(k, c)
      tp_mfma<NS, sl, 1>(tp, -xc, yr);                                  // A^T(r, c) -= L21(c, k) U12(k, r)
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
      unroll_for<0, NS>([&](auto SLc) {
        constexpr int sl = decltype(SLc)::value;
        if (tbi[sl] >= 0 && tbj[sl] > t) update(SLc, tbi[sl], tbj[sl], t - 1);                 // (S3b) of step t - 1
      });
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
      unroll_for<0, NS>([&](auto SLc) {
        constexpr int sl = decltype(SLc)::value;
        if (tbi[sl] >= 0 && tbj[sl] == t + 1) {                // (S3a) column band t + 1, then the hand-over
          update(SLc, tbi[sl], tbj[sl], t);
          unroll_for<0, 4>([&](auto Qc) {
            constexpr int q = decltype(Qc)::value;
            const int c = g + 4 * q;
            const double v0 = tp_get<NS, sl, 0, q>(tp), v1 = tp_get<NS, sl, 1, q>(tp);
            if (tbi[sl] == t + 1) { S.Ts[0][l15][c] = v0; S.Ts[1][l15][c] = v1; }
            else { S.Ps[0][c][16 * (tbi[sl] - t - 2) + l15] = v0; S.Ps[1][c][16 * (tbi[sl] - t - 2) + l15] = v1; }
          });
        }
      });
    }
  }
}
// ------------------------------------------------------------------------------------------------
// k_diag_zsy_w (round 4): the complex diagonal blok on the MFMA pipe.  The blok does not fit LDS twice (Re / Im planes:
// 132 KB, and the workgroup must fit beside a k_update workgroup), so its 16 x 16 tiles live in REGISTERS, in the
// accumulator layout of v_mfma_f64_16x16x4 -- lane (l15, g), register q = entry (row l15, column g + 4q) -- five tiles
// (Re + Im: 80 VGPRs) per wave on waves 1-7; wave 0 carries the dependency chain.  Per 16-column step t:
//   (S1) wave 0 factorizes the diagonal tile as diag_ldlt_body does (kernels.hip): tile and inverse held negated, one
//        complex rank-1 update = four MFMAs whose operands are column j itself, the pivot's reciprocal and the scaled
//        column on the vector pipe in between (PASTIX_sytrf / hetrf, compute_diag.c:223-242, :326-345); the transposed
//        inverse of the unit-lower tile rides along (four more MFMAs per column).  Meanwhile waves 1-7 apply the PREVIOUS
//        step's trailing update to their tiles right of the next column band (S3b).
//   (S2) rows below the tile, one 16-row block per wave: (L D) = A21 inv(tile)^T|^H (16 MFMAs), L = (L D) / d
//        (TRSM "R","L","T"|"C","U" + the scaling, compute_diag.c:284-298), to global memory and to LDS.
//   (S3a) the trailing update of the next column band (one tile per wave), handed to wave 0 / the next (S2) through LDS.
// LDS: the step's diagonal tile, its inverse, d and 1/d, the unsolved and the solved rows below (66.8 KB).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double zreadlane(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

struct DiagZLds {
  static constexpr int XR = 112;
  double Ts[2][16][17];        // the step's diagonal tile [plane][row][column]
  double Wl[2][256];           // inv(tile)^T [plane][k * 16 + i]
  double Dd[2][2][16];         // the tile's diagonal [step parity][plane][column] (read by the next step's S3b too)
  double Rr[2][16];            // its reciprocals [plane][column]
  double Ps[2][16][XR];        // rows below the tile, unsolved [plane][column][row]
  double Xs[2][16][XR];        // rows below the tile, solved: L [plane][column][row]
};
// (COH: results stored write-through -- the run launch hands the blok to other workgroups, run_sync.h)
template <bool HERM, bool COH>
__device__ __forceinline__ void diag_zsy_body(DiagZLds& S, const Arenas& ar, const PanelTask& tk, double* __restrict__ dinv_ws,
                                              const double critere, long long* __restrict__ nbpivot, const int tid) {
  constexpr int NS = 5;
  double* Ar = ar.p[0] + tk.off;
  double* Ai = ar.p[2] + tk.off;
  const int ld = tk.stride, w = tk.width;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int nbt = (w + 15) >> 4;                             // 16-column steps
  typedef double d4v __attribute__((ext_vector_type(4)));
  // tiles (bi >= bj) but (0, 0), column by column: ids 0-6 (1..7, 0), 7-13 (1..7, 1), 14-19 (2..7, 2), ... 34 (7, 7);
  // wave 1 + id % 7 holds tile id in slot id / 7 -- every column band is spread over the waves
  TilePairs<NS> tp;                                    // plane 0: Re, plane 1: Im (diag_lu_body: how they are held)
  int tbi[NS], tbj[NS];
  if (wave > 0) {
    unroll_for<0, NS>([&](auto SLc) {
      constexpr int sl = decltype(SLc)::value;
      int id = (wave - 1) + 7 * sl, bj = 0, cnt = 7;
      if (id >= 14) { id -= 14; bj = 2; cnt = 6; while (id >= cnt) { id -= cnt; bj++; cnt--; } }
      else if (id >= 7) { id -= 7; bj = 1; }
      const int bi = (bj < 2 ? 1 : bj) + id;
      const bool on = bi < nbt;
      tbi[sl] = on ? bi : -1;
      tbj[sl] = bj;
      double vr[4], vi[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * bi + l15, col = 16 * bj + g + 4 * q;
        const bool v = on && row < w && col < w && row >= col;
        const int64_t o = (int64_t)min(row, w - 1) + (int64_t)min(col, w - 1) * ld;
        const double re = Ar[o], im = Ai[o];
        vr[q] = v ? re : 0.0;
        vi[q] = v ? im : 0.0;
      }
      unroll_for<0, 4>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        tp_set<NS, sl, 0, q>(tp, vr[q]);
        tp_set<NS, sl, 1, q>(tp, vi[q]);
      });
      if (on && bj == 0) {                                   // the rows below the first tile go to LDS at once
#pragma unroll
        for (int q = 0; q < 4; q++) {
          S.Ps[0][g + 4 * q][16 * (bi - 1) + l15] = vr[q];
          S.Ps[1][g + 4 * q][16 * (bi - 1) + l15] = vi[q];
        }
      }
    });
  }
  // the trailing update of resident tile SL with the solved rows of step tp_ (in Xs): C -= (L D)(rows) L(cols)^T|^H
  auto update = [&](auto SLc, const int bi, const int bj, const int tp_) {
    constexpr int sl = decltype(SLc)::value;
    const int ro = 16 * (bi - tp_ - 1) + l15, co = 16 * (bj - tp_ - 1) + l15;
    const double (*Dk)[16] = S.Dd[tp_ & 1];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int k = 4 * ks + g;
      const double lr = S.Xs[0][k][co], li = S.Xs[1][k][co];                 // L(c, k)
      const double xr = S.Xs[0][k][ro], xi = S.Xs[1][k][ro];                 // L(r, k)
      const double dr = Dk[0][k], di = Dk[1][k];
      if (!HERM) {
        const double ldr = xr * dr - xi * di, ldi = xr * di + xi * dr;       // (L D)(r, k)
        tp_mfma<NS, sl, 0>(tp, -lr, ldr);
        tp_mfma<NS, sl, 0>(tp, li, ldi);
        tp_mfma<NS, sl, 1>(tp, -li, ldr);
        tp_mfma<NS, sl, 1>(tp, -lr, ldi);
      } else {
        const double ldr = xr * dr, ldi = xi * dr;                           // L(r, k) Re d;  times conj(L(c, k))
        tp_mfma<NS, sl, 0>(tp, -lr, ldr);
        tp_mfma<NS, sl, 0>(tp, -li, ldi);
        tp_mfma<NS, sl, 1>(tp, -lr, ldi);
        tp_mfma<NS, sl, 1>(tp, li, ldr);
      }
      __builtin_amdgcn_sched_barrier(0);                     // (one k-slice's operands at a time: 80 VGPRs are resident)
    }
  };
  int npiv = 0;
  const double c2 = critere * critere, cmin2 = fmax(c2, 2.2250738585072014e-308);
  // (the two roles are separate loops with the same barriers: the resident tiles are not live in wave 0's code)
  if (wave == 0) {
    for (int t = 0; t < nbt; t++) {
      const int kb = 16 * t, nb = min(16, w - kb);
      __syncthreads();                                       // (A) Ts / Ps hold column band t
      d4v Sr, Si, Vr, Vi;                                    // the tile and the inverse, negated (see diag_llt_body)
      if (lane < 16) { S.Dd[t & 1][0][lane] = 1.0; S.Dd[t & 1][1][lane] = 0.0; S.Rr[0][lane] = 1.0; S.Rr[1][lane] = 0.0; }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        double re, im;
        if (t == 0) {
          const int64_t o = (int64_t)min(l15, w - 1) + (int64_t)min(c, w - 1) * ld;
          re = Ar[o]; im = Ai[o];
          if (l15 >= w || c >= w) { re = 0.0; im = 0.0; }
        } else {
          re = S.Ts[0][l15][c]; im = S.Ts[1][l15][c];
        }
        Sr[q] = (c <= l15) ? -re : 0.0;
        Si[q] = (c <= l15) ? -im : 0.0;
        Vr[q] = (c == l15) ? -1.0 : 0.0;
        Vi[q] = 0.0;
      }
      unroll_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 2, gj = j & 3;
        if (j < nb) {
          const bool ing = (g == gj), below = ing && l15 > j;
          double dr = -zreadlane(Sr[qj], j + 16 * gj);
          double di = HERM ? 0.0 : -zreadlane(Si[qj], j + 16 * gj);
          double m = HERM ? dr * dr : __builtin_fma(dr, dr, di * di);
          double y = __builtin_amdgcn_rcp(m);
          if (__builtin_expect(!(m >= cmin2), 0)) {            // |d| < critere (or NaN)
            if (m < c2) { dr = critere; di = 0.0; npiv++; }
            m = __builtin_fma(dr, dr, di * di);
            y = __builtin_amdgcn_rcp(m);
            Sr[qj] = (ing && l15 == j) ? -dr : Sr[qj];
            Si[qj] = (ing && l15 == j) ? -di : Si[qj];
          }
          y = __builtin_fma(__builtin_fma(-m, y, 1.0), y, y);
          y = __builtin_fma(__builtin_fma(-m, y, 1.0), y, y);
          const double ivr = dr * y, ivi = -di * y;            // 1 / d
          // -L(:, j) = S(:, j) / d
          const double syr = HERM ? Sr[qj] * ivr : __builtin_fma(Sr[qj], ivr, -Si[qj] * ivi);
          const double syi = HERM ? Si[qj] * ivr : __builtin_fma(Sr[qj], ivi, Si[qj] * ivr);
          const double amr = below ? syr : 0.0, ami = below ? syi : 0.0, nami = -ami;
          // sy: a(r, c) -= (L D)(r, j) L(c, j), with -(L D)(:, j) = S(:, j);  he: a(r, c) -= (L(r, j) Re d) conj(L(c, j))
          const double bmr = below ? (HERM ? syr * dr : Sr[qj]) : 0.0, bmi = below ? (HERM ? syi * dr : Si[qj]) : 0.0;
          Sr[qj] = below ? syr : Sr[qj];
          Si[qj] = below ? syi : ((HERM && ing && l15 == j) ? 0.0 : Si[qj]);
          if (j < 15) {
            if (!HERM) {
              Sr = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, bmr, Sr, 0, 0, 0);
              Si = __builtin_amdgcn_mfma_f64_16x16x4f64(ami, bmr, Si, 0, 0, 0);
              Sr = __builtin_amdgcn_mfma_f64_16x16x4f64(nami, bmi, Sr, 0, 0, 0);
              Si = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, bmi, Si, 0, 0, 0);
            } else {
              Sr = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, bmr, Sr, 0, 0, 0);
              Si = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, bmi, Si, 0, 0, 0);
              Sr = __builtin_amdgcn_mfma_f64_16x16x4f64(ami, bmi, Sr, 0, 0, 0);
              Si = __builtin_amdgcn_mfma_f64_16x16x4f64(nami, bmr, Si, 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (lane == 0) { S.Dd[t & 1][0][j] = dr; S.Dd[t & 1][1][j] = di; S.Rr[0][j] = ivr; S.Rr[1][j] = ivi; }
          const double vmr = ing ? Vr[qj] : 0.0, vmi = ing ? Vi[qj] : 0.0;   // (unit diagonal: the inverse's column is not scaled)
          if (j < 15) {
            Vr = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, vmr, Vr, 0, 0, 0);
            Vi = __builtin_amdgcn_mfma_f64_16x16x4f64(amr, vmi, Vi, 0, 0, 0);
            Vr = __builtin_amdgcn_mfma_f64_16x16x4f64(nami, vmi, Vr, 0, 0, 0);
            Vi = __builtin_amdgcn_mfma_f64_16x16x4f64(ami, vmr, Vi, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      double* dst = dinv_ws + tk.dinv_off + (int64_t)t * 512;   // [256 re][256 im] per block
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        if (c <= l15 && l15 < nb) {
          const int64_t o = (kb + l15) + (int64_t)(kb + c) * ld;
          pst<COH>(&Ar[o], -Sr[q]);
          pst<COH>(&Ai[o], -Si[q]);
        }
        S.Wl[0][l15 * 16 + c] = -Vr[q];                      // W(k = l15, i = c) = inv(tile)(i, k)
        S.Wl[1][l15 * 16 + c] = -Vi[q];
        pst<COH>(&dst[c + 16 * l15], -Vr[q]);
        pst<COH>(&dst[256 + c + 16 * l15], -Vi[q]);
      }
      __syncthreads();                                       // (B)
      __syncthreads();                                       // (C)
    }
    if (lane == 0 && npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
    return;
  }
  for (int t = 0; t < nbt; t++) {
    const int kb = 16 * t, nb = min(16, w - kb), rem = w - kb - nb;
    __syncthreads();                                         // (A) Ts / Ps hold column band t
    if (t > 0) {
      // (S3b) of step t - 1: the tiles right of column band t
      unroll_for<0, NS>([&](auto SLc) {
        constexpr int sl = decltype(SLc)::value;
        if (tbi[sl] >= 0 && tbj[sl] > t) update(SLc, tbi[sl], tbj[sl], t - 1);
      });
    }
    __syncthreads();                                         // (B) the tile is factorized
    if ((wave - 1) * 16 < rem) {
      // (S2) row block wave - 1 below the tile
      const int ro = 16 * (wave - 1) + l15;
      d4v Yr = {0.0, 0.0, 0.0, 0.0}, Yi = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const int k = 4 * ks + g;
        const double wr = S.Wl[0][k * 16 + l15], wi = HERM ? -S.Wl[1][k * 16 + l15] : S.Wl[1][k * 16 + l15];   // inv(tile)(c = l15, k)
        const double pr = S.Ps[0][k][ro], pi = S.Ps[1][k][ro];                                            // A21(r, k)
        Yr = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, pr, Yr, 0, 0, 0);
        Yi = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, pi, Yi, 0, 0, 0);
        Yr = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, pi, Yr, 0, 0, 0);
        Yi = __builtin_amdgcn_mfma_f64_16x16x4f64(wi, pr, Yi, 0, 0, 0);
      }
      const int64_t o0 = (kb + nb + ro) + (int64_t)kb * ld;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = g + 4 * q;
        const double ivr = S.Rr[0][c], ivi = S.Rr[1][c];
        const double xr = __builtin_fma(Yr[q], ivr, -Yi[q] * ivi), xi = __builtin_fma(Yr[q], ivi, Yi[q] * ivr);   // L = (L D) / d
        S.Xs[0][c][ro] = xr;
        S.Xs[1][c][ro] = xi;
        if (c < nb && ro < rem) {
          pst<COH>(&Ar[o0 + (int64_t)c * ld], xr);
          pst<COH>(&Ai[o0 + (int64_t)c * ld], xi);
        }
      }
    }
    __syncthreads();                                         // (C) the rows below are solved
    if (t + 1 < nbt) {
      // (S3a) column band t + 1: update, then hand over -- the diagonal tile to wave 0, the others to the next (S2)
      unroll_for<0, NS>([&](auto SLc) {
        constexpr int sl = decltype(SLc)::value;
        if (tbi[sl] >= 0 && tbj[sl] == t + 1) {
          update(SLc, tbi[sl], tbj[sl], t);
          unroll_for<0, 4>([&](auto Qc) {
            constexpr int q = decltype(Qc)::value;
            const int c = g + 4 * q;
            const double v0 = tp_get<NS, sl, 0, q>(tp), v1 = tp_get<NS, sl, 1, q>(tp);
            if (tbi[sl] == t + 1) { S.Ts[0][l15][c] = v0; S.Ts[1][l15][c] = v1; }
            else { S.Ps[0][c][16 * (tbi[sl] - t - 2) + l15] = v0; S.Ps[1][c][16 * (tbi[sl] - t - 2) + l15] = v1; }
          });
        }
      });
    }
  }
}

}  // namespace pastix_amd
