/*
 * sopalin_oracle_impl.h -- TEST INFRASTRUCTURE ONLY.  Included by sopalin_oracle.c once per
 * arithmetic (T = double, T = double complex).  Plain-C restatement of the reference's CPU
 * sopalin path (PaStiX 5.2.2.16); every function cites the reference file:line it follows.
 * The reference reaches its flops through Fortran BLAS; here they are written as plain loops
 * with the same operand roles, so results agree with the reference to rounding, not bitwise.
 *
 * Panels: all cblk panels are concatenated in cblk order in one array; cblk k starts at
 * poff[k] = sum_{c<k} stride(c)*width(c); element (row r of blok b, column j) of cblk k is
 * at poff[k] + coefind(b) + (r-frownum(b)) + j*stride(k)   (SURVEY 8a row a1,
 * src/blend/src/solver.h:94-117).
 *
 * Requires: T, NAME(x), CONJ(x), ABS(x), SQRT(x), SYR_FULL (1 in complex builds)
 */

/* ---- PASTIX_potrf: unblocked LLt, compute_diag.c:124-153 ------------------------------ */
static void NAME(potrf)(T *A, i64 n, i64 ld, i64 *nbpivot, double critere)
{
  i64 k, i, j;
  for (k = 0; k < n; k++) {
    T *d = A + k * (ld + 1);
    if (ABS(*d) < critere) { *d = (T)critere; (*nbpivot)++; }      /* :133-137 */
    *d = SQRT(*d);                                                /* :140-142 */
    { T inv = (T)1.0 / *d;                                        /* SCAL :150 */
      for (i = 1; i < n - k; i++) d[i] *= inv; }
    /* SYR "L" :151.  In complex builds SOPALIN_SYR is GER(x, x^T): the FULL square is updated,
     * no conjugation (sopalin_compute.h:549-562); in real builds dsyr "L" touches the lower part */
    for (j = 1; j < n - k; j++) {
      T xj = d[j];
      T *c = d + j * ld;
      for (i = SYR_FULL ? 1 : j; i < n - k; i++) c[i] -= d[i] * xj;
    }
  }
}

/* X(m x n) := X * L^{-T}, L lower n x n non-unit ("R","L","T","N"), compute_diag.c:191-195,
 * compute_trsm.c:76-85.  unit=1 -> "U" (unit diagonal).  conj=1 -> "C". */
static void NAME(trsm_rlt)(i64 m, i64 n, const T *L, i64 ldl, T *X, i64 ldx, int unit, int cj)
{
  i64 c, p, r;
  for (c = 0; c < n; c++) {
    T *xc = X + c * ldx;
    for (p = 0; p < c; p++) {
      T l = L[c + p * ldl];
      const T *xp = X + p * ldx;
      if (cj) l = CONJ(l);
      for (r = 0; r < m; r++) xc[r] -= xp[r] * l;
    }
    if (!unit) {
      T d = L[c + c * ldl];
      T inv;
      if (cj) d = CONJ(d);
      inv = (T)1.0 / d;
      for (r = 0; r < m; r++) xc[r] *= inv;
    }
  }
}

/* PASTIX_potrf_block (nb = MAXSIZEOFBLOCKS = 64), compute_diag.c:46,171-203 */
static void NAME(potrf_block)(T *A, i64 n, i64 ld, i64 *nbpivot, double critere)
{
  i64 k, nblk = (n + 63) / 64;
  for (k = 0; k < nblk; k++) {
    i64 bs = (n - k * 64 < 64) ? n - k * 64 : 64;
    T *t = A + (k * 64) * (ld + 1), *t1 = t + bs, *t2 = t1 + ld * bs;
    NAME(potrf)(t, bs, ld, nbpivot, critere);
    if (k * 64 + bs < n) {
      i64 ms = n - (k * 64 + bs), i, j, p;
      NAME(trsm_rlt)(ms, bs, t, ld, t1, ld, 0, 0);                 /* :191-195 */
      /* SYRK "L","N": C -= A A^T on the lower triangle (:197-200); "T" even in complex
       * (sopalin_compute.h SOPALIN_SYRK) */
      for (j = 0; j < ms; j++)
        for (p = 0; p < bs; p++) {
          T a = t1[j + p * ld];
          for (i = j; i < ms; i++) t2[i + j * ld] -= t1[i + p * ld] * a;
        }
    }
  }
}

/* PASTIX_sytrf / PASTIX_hetrf: unblocked LDLt / LDLh, compute_diag.c:223-242, :326-345 */
static void NAME(sytrf)(T *A, i64 n, i64 ld, i64 *nbpivot, double critere, int herm)
{
  i64 k, i, j;
  for (k = 0; k < n; k++) {
    T *d = A + k * (ld + 1);
    if (ABS(*d) < critere) { *d = (T)critere; (*nbpivot)++; }
    { T inv = (T)1.0 / *d;
      for (i = 1; i < n - k; i++) d[i] *= inv; }
    /* SYR/HER "L" with alpha = -d_k : A -= d_k x x^T (or x x^H).  Complex symmetric: GER, full
     * square (sopalin_compute.h:549-562); Hermitian: zher "L" */
    for (j = 1; j < n - k; j++) {
      T xj = herm ? CONJ(d[j]) : d[j];
      T *c = d + j * ld;
      T s = (*d) * xj;
      for (i = (SYR_FULL && !herm) ? 1 : j; i < n - k; i++) c[i] -= d[i] * s;
    }
  }
}

/* PASTIX_sytrf_block / _hetrf_block, compute_diag.c:262-307, :365-410; tmp4 = workspace */
static void NAME(sytrf_block)(T *A, i64 n, i64 ld, i64 *nbpivot, double critere, T *tmp4, int herm)
{
  i64 k, nblk = (n + 63) / 64;
  for (k = 0; k < nblk; k++) {
    i64 bs = (n - k * 64 < 64) ? n - k * 64 : 64;
    T *t = A + (k * 64) * (ld + 1), *t1 = t + bs, *t2 = t1 + ld * bs;
    NAME(sytrf)(t, bs, ld, nbpivot, critere, herm);
    if (k * 64 + bs < n) {
      i64 ms = n - (k * 64 + bs), i, j, p, col;
      NAME(trsm_rlt)(ms, bs, t, ld, t1, ld, 1, herm);              /* "R","L","T|C","U" */
      for (col = 0; col < bs; col++) {                            /* :289-298 */
        T alpha = (T)1.0 / t[col * (ld + 1)];
        for (i = 0; i < ms; i++) { tmp4[i + col * ms] = t1[i + col * ld]; t1[i + col * ld] *= alpha; }
      }
      /* GEMM "N","T|C": A22 -= (L D) L^T|H, full ms x ms block (:299-304) */
      for (j = 0; j < ms; j++)
        for (p = 0; p < bs; p++) {
          T b = t1[j + p * ld];
          if (herm) b = CONJ(b);
          for (i = 0; i < ms; i++) t2[i + j * ld] -= tmp4[i + p * ms] * b;
        }
    }
  }
}

/* PASTIX_getrf: unblocked LU without row pivoting, compute_diag.c:432-469 */
static void NAME(getrf)(T *A, i64 m, i64 n, i64 ld, i64 *nbpivot, double critere)
{
  i64 j, i, l, mn = m < n ? m : n;
  for (j = 0; j < mn; j++) {
    T *d = A + j * (ld + 1);
    if (ABS(*d) < critere) { *d = (T)critere; (*nbpivot)++; }
    { T inv = (T)1.0 / *d;
      for (i = 1; i < m - j; i++) d[i] *= inv; }
    if (j + 1 < mn)
      for (l = 1; l < n - j; l++) {                               /* GER :452-454 */
        T u = d[l * ld];
        T *c = d + l * ld;
        for (i = 1; i < m - j; i++) c[i] -= d[i] * u;
      }
  }
  { T *d = A + (n - 1) * (ld + 1);                                /* :458-466 */
    if (ABS(*d) < critere) { *d = (T)critere; (*nbpivot)++; } }
}

/* PASTIX_getrf_block, compute_diag.c:486-518 */
static void NAME(getrf_block)(T *A, i64 rows, i64 cols, i64 ld, i64 *nbpivot, double critere)
{
  i64 k, nblk = (cols + 63) / 64;
  for (k = 0; k < nblk; k++) {
    i64 bs = (cols - k * 64 < 64) ? cols - k * 64 : 64;
    T *t = A + (k * 64) * (ld + 1), *t1 = t + bs, *t2 = t + ld * bs, *t3 = t + (ld + 1) * bs;
    NAME(getrf)(t, rows - k * 64, bs, ld, nbpivot, critere);
    if (k * 64 + bs < cols) {
      i64 ms = rows - k * 64 - bs, i, j, p;
      /* TRSM "L","L","N","U": U12 := L11^{-1} U12 (:505-508) */
      for (j = 0; j < ms; j++) {
        T *u = t2 + j * ld;
        for (p = 0; p < bs; p++) {
          T up = u[p];
          for (i = p + 1; i < bs; i++) u[i] -= t[i + p * ld] * up;
        }
      }
      /* GEMM "N","N": A22 -= L21 U12 (:510-511) */
      for (j = 0; j < ms; j++)
        for (p = 0; p < bs; p++) {
          T u = t2[p + j * ld];
          for (i = 0; i < ms; i++) t3[i + j * ld] -= t1[i + p * ld] * u;
        }
    }
  }
}

/* X(m x n) := X * U^{-1}, U upper n x n ("R","U","N",unit?), compute_trsm.c:62-66 */
static void NAME(trsm_run)(i64 m, i64 n, const T *U, i64 ldu, T *X, i64 ldx, int unit)
{
  i64 c, p, r;
  for (c = 0; c < n; c++) {
    T *xc = X + c * ldx;
    for (p = 0; p < c; p++) {
      T u = U[p + c * ldu];
      const T *xp = X + p * ldx;
      for (r = 0; r < m; r++) xc[r] -= xp[r] * u;
    }
    if (!unit) {
      T inv = (T)1.0 / U[c + c * ldu];
      for (r = 0; r < m; r++) xc[r] *= inv;
    }
  }
}

/*
 * The whole factorization: sequential loop over cblks in index order (a valid order of
 * the reference's static task list: every contribution goes to a higher-numbered cblk),
 * each step = compute_1d (sopalin_compute.c:747-863):
 *   factor_diag (compute_diag.c:538-605) -> factor_trsm1d (compute_trsm.c:128-171) ->
 *   for each off-diagonal blok i: compute_1dgemm (sopalin_compute.c:865-1032) =
 *   compute_contrib_compact (:270-374) + add_contrib_local (:391-598) for every j >= i.
 * facto: 0 LLt (po), 1 LDLt (sy), 2 LU (ge), 3 LDLh (he)   (api.h:381-384)
 * cblk4[4k..] = fcolnum,lcolnum,bloknum,stride (cblknbr+1 entries); blok4[4b..] =
 * frownum,lrownum,cblknum,coefind.  L,U: concatenated panels.  Returns 0, or -1 on a
 * non-finite pivot.
 */
int NAME(sopalin)(int facto, i64 cblknbr, const i64 *cblk4, const i64 *blok4,
                  T *L, T *U, double critere, i64 *nbpivot_out)
{
  i64 k, nbpivot = 0, coefmax = 0, maxpanel = 0;
  i64 *poff = (i64 *)malloc((cblknbr + 1) * sizeof(i64));
  T *w1, *w2;
  int herm = (facto == 3);
  poff[0] = 0;
  for (k = 0; k < cblknbr; k++) {
    i64 w = cblk4[4 * k + 1] - cblk4[4 * k] + 1, s = cblk4[4 * k + 3];
    i64 b, fb = cblk4[4 * k + 2], lb = cblk4[4 * (k + 1) + 2], rem = s - w;
    poff[k + 1] = poff[k] + s * w;
    if (s * w > maxpanel) maxpanel = s * w;
    for (b = fb + 1; b < lb; b++) {
      i64 h = blok4[4 * b + 1] - blok4[4 * b] + 1;
      if (rem * h > coefmax) coefmax = rem * h;
      rem -= h;
    }
  }
  /* maxbloktab1/2 (sopalin_init.c:1062-1065); sized generously so that the degenerate
   * single-cblk LDLt case (SURVEY 8c hazard i) cannot overflow here */
  w1 = (T *)malloc((size_t)(coefmax + maxpanel + 1) * sizeof(T));
  w2 = (T *)malloc((size_t)(coefmax + 1) * sizeof(T));

  for (k = 0; k < cblknbr; k++) {
    i64 fcol = cblk4[4 * k], w = cblk4[4 * k + 1] - fcol + 1, s = cblk4[4 * k + 3];
    i64 fb = cblk4[4 * k + 2], lb = cblk4[4 * (k + 1) + 2];
    T *Lk = L + poff[k], *Uk = U ? U + poff[k] : NULL;
    i64 dimb = s - w, i;
    (void)fcol;

    /* ---- factor_diag ---- */
    if (facto == 0) NAME(potrf_block)(Lk, w, s, &nbpivot, critere);
    else if (facto == 2) {
      i64 a, b;
      NAME(getrf_block)(Lk, w, w, s, &nbpivot, critere);
      for (a = 0; a < w; a++)                                     /* DimTrans :521-532,:564-567 */
        for (b = 0; b < w; b++) Uk[a * s + b] = Lk[b * s + a];
    } else NAME(sytrf_block)(Lk, w, s, &nbpivot, critere, w1, herm);

    /* ---- factor_trsm1d ---- */
    if (fb + 1 < lb) {
      if (facto == 0) NAME(trsm_rlt)(dimb, w, Lk, s, Lk + w, s, 0, 0);
      else if (facto == 2) {
        NAME(trsm_run)(dimb, w, Lk, s, Lk + w, s, 0);              /* "R","U","N","N" on dL */
        NAME(trsm_run)(dimb, w, Uk, s, Uk + w, s, 1);              /* "R","U","N","U" on dU */
      } else {
        i64 c, r;
        NAME(trsm_rlt)(dimb, w, Lk, s, Lk + w, s, 1, herm);        /* gives L*D */
        for (c = 0; c < w; c++) {                                 /* compute_trsm.c:101-113 */
          T alpha = (T)1.0 / Lk[c + c * s];
          for (r = 0; r < dimb; r++) { w1[r + c * dimb] = Lk[w + r + c * s]; Lk[w + r + c * s] *= alpha; }
        }
      }
    }

    /* ---- updates ---- */
    for (i = fb + 1; i < lb; i++) {
      i64 ci = blok4[4 * i + 3];                 /* coefind(i): row offset of blok i in panel */
      i64 dimi = s - ci;                         /* rows from blok i to the bottom */
      i64 dimj = blok4[4 * i + 1] - blok4[4 * i] + 1;
      i64 t = blok4[4 * i + 2];                  /* facing cblk */
      i64 tf = cblk4[4 * t], ts = cblk4[4 * t + 3];
      i64 tfb = cblk4[4 * t + 2], tlb = cblk4[4 * (t + 1) + 2];
      T *Lt = L + poff[t], *Ut = U ? U + poff[t] : NULL;
      i64 r, c, p, j, b3;
      /* compute_contrib_compact: gc = w2 (ld dimi), gb = second result for LU */
      T *gc = w2, *gu = w1 + maxpanel;
      for (c = 0; c < dimj; c++) for (r = 0; r < dimi; r++) gc[r + c * dimi] = 0;
      if (facto == 2) for (c = 0; c < dimj; c++) for (r = 0; r < dimi; r++) gu[r + c * dimi] = 0;
      for (p = 0; p < w; p++)
        for (c = 0; c < dimj; c++) {
          if (facto == 0) {                      /* GEMM "N","C": A_{i:} A_i^H (:326-332) */
            T b = CONJ(Lk[ci + c + p * s]);
            for (r = 0; r < dimi; r++) gc[r + c * dimi] += Lk[ci + r + p * s] * b;
          } else if (facto == 2) {               /* L U^T and U L^T (:312-324) */
            T bu = Uk[ci + c + p * s], bl = Lk[ci + c + p * s];
            for (r = 0; r < dimi; r++) {
              gc[r + c * dimi] += Lk[ci + r + p * s] * bu;
              gu[r + c * dimi] += Uk[ci + r + p * s] * bl;
            }
          } else {                               /* L (L D)^T|H with the saved copy (:356-371) */
            T b = w1[(ci - w) + c + p * dimb];
            if (herm) b = CONJ(b);
            for (r = 0; r < dimi; r++) gc[r + c * dimi] += Lk[ci + r + p * s] * b;
          }
        }
      /* add_contrib_local for every blok j >= i (sopalin_compute.c:911-1010, :391-598) */
      b3 = tfb;
      for (j = i; j < lb; j++) {
        i64 fj = blok4[4 * j], lj = blok4[4 * j + 1];
        i64 hj = lj - fj + 1;
        i64 step = blok4[4 * j + 3] - ci;        /* rows between blok i and blok j (:450-452) */
        i64 coloff = blok4[4 * i] - tf;          /* frownum(b1) - fcolnum(cbl) */
        while (!(fj >= blok4[4 * b3] && lj <= blok4[4 * b3 + 1])) {  /* :938-945 */
          b3++;
          if (b3 >= tlb) { free(poff); free(w1); free(w2); return -2; }
        }
        {
          i64 ga = blok4[4 * b3 + 3] + coloff * ts + (fj - blok4[4 * b3]);   /* :427-429 */
          for (c = 0; c < dimj; c++)
            for (r = 0; r < hj; r++) Lt[ga + r + c * ts] -= gc[step + r + c * dimi];
          if (facto == 2) {
            if (b3 != tfb) {
              for (c = 0; c < dimj; c++)
                for (r = 0; r < hj; r++) Ut[ga + r + c * ts] -= gu[step + r + c * dimi];
            } else if (j != i) {                 /* transposed into coeftab (:430-435,:572-575) */
              i64 ga2 = blok4[4 * b3 + 3] + coloff + (fj - blok4[4 * b3]) * ts;
              for (c = 0; c < dimj; c++)
                for (r = 0; r < hj; r++) Lt[ga2 + c + r * ts] -= gu[step + r + c * dimi];
            }
          }
        }
      }
    }
  }
  free(poff); free(w1); free(w2);
  if (nbpivot_out) *nbpivot_out = nbpivot;
  return 0;
}

/*
 * Coefficient fill: CoefMatrix_Init (coefinit.c:283-296) + Csc2solv_cblk
 * (csc_intern_solve.c:65-132) applied to the permuted matrix that CscOrdistrib
 * (csc_intern_build.c) builds: entry A(i,j) of the user matrix goes to column perm[j],
 * row perm[i]; a symmetric (lower-only) input is mirrored so that every column holds both
 * triangles; only rows >= fcolnum of the cblk are kept (:88-89).  LU: the value of the
 * transposed entry goes to ucoeftab for rows outside the diagonal blok (:110-116).
 * colptr/rows 1-based ("Fortran numbering", pastix.h:159-166); perm 0-based old->new.
 */
int NAME(fill)(int facto, int sym, i64 n, const i64 *colptr, const i64 *rows, const T *vals,
               const i64 *perm, i64 cblknbr, const i64 *cblk4, const i64 *blok4, T *L, T *U)
{
  i64 k, j, p;
  i64 *col2cblk = (i64 *)malloc(n * sizeof(i64));
  i64 *poff = (i64 *)malloc((cblknbr + 1) * sizeof(i64));
  int herm = (facto == 3);
  poff[0] = 0;
  for (k = 0; k < cblknbr; k++) {
    i64 w = cblk4[4 * k + 1] - cblk4[4 * k] + 1, s = cblk4[4 * k + 3];
    poff[k + 1] = poff[k] + s * w;
    for (j = cblk4[4 * k]; j <= cblk4[4 * k + 1]; j++) col2cblk[j] = k;
  }
  for (p = 0; p < poff[cblknbr]; p++) { L[p] = 0; if (U) U[p] = 0; }
  for (j = 0; j < n; j++)
    for (p = colptr[j] - 1; p < colptr[j + 1] - 1; p++) {
      i64 i = rows[p] - 1;
      int pass, npass = (sym && i != j) ? 2 : 1;
      for (pass = 0; pass < npass; pass++) {
        /* pass 0: entry (i,j) with value v; pass 1: mirrored entry (j,i) */
        i64 pr = perm[pass ? j : i], pc = perm[pass ? i : j];
        T v = vals[p];
        i64 kc, b, fb, lb;
        if (pass && herm) v = CONJ(v);
        /* L side: column pc, row pr if pr >= fcolnum(cblk(pc)) */
        kc = col2cblk[pc];
        if (pr >= cblk4[4 * kc]) {
          fb = cblk4[4 * kc + 2]; lb = cblk4[4 * (kc + 1) + 2];
          for (b = fb; b < lb && (blok4[4 * b + 1] < pr || blok4[4 * b] > pr); b++) ;
          if (b < lb)
            L[poff[kc] + blok4[4 * b + 3] + (pr - blok4[4 * b]) + (pc - cblk4[4 * kc]) * cblk4[4 * kc + 3]] = v;
        }
        /* U side (LU only): the transposed entry (pc,pr) lands in column pr's panel at row pc,
         * off-diagonal bloks only */
        if (U && facto == 2) {
          kc = col2cblk[pr];
          if (pc >= cblk4[4 * kc]) {
            fb = cblk4[4 * kc + 2]; lb = cblk4[4 * (kc + 1) + 2];
            for (b = fb; b < lb && (blok4[4 * b + 1] < pc || blok4[4 * b] > pc); b++) ;
            if (b < lb && b != fb)
              U[poff[kc] + blok4[4 * b + 3] + (pc - blok4[4 * b]) + (pr - cblk4[4 * kc]) * cblk4[4 * kc + 3]] = v;
          }
        }
      }
    }
  free(col2cblk); free(poff);
  return 0;
}

/*
 * "Fake factorisation" fill: CoefMatrix_Init with IPARM_FILL_MATRIX = API_YES (coefinit.c:343-443), the reference's
 * structure-only benchmark mode -- no CSC: every panel entry of coeftab is 1 (:411), of ucoeftab 2 (:413), the diagonal
 * of every diagonal blok is gnodenbr^2 (:425-428: "on s'assure que la matrice est diagonale dominante"), and for LU the
 * strictly upper part of coeftab's diagonal blok is a copy of ucoeftab's strictly lower part, i.e. 2 (:431-441).  The
 * pivot threshold of such a run is (gnodenbr^2 + gnodenbr) sqrt(eps) (sopalin3d.c:597-598).
 */
int NAME(fill_fake)(int facto, i64 gnodenbr, i64 cblknbr, const i64 *cblk4, T *L, T *U)
{
  i64 k, p, off = 0, r, c;
  for (k = 0; k < cblknbr; k++) {
    i64 w = cblk4[4 * k + 1] - cblk4[4 * k] + 1, s = cblk4[4 * k + 3];
    for (p = 0; p < s * w; p++) { L[off + p] = 1; if (U && facto == 2) U[off + p] = 2; }
    for (c = 0; c < w; c++) {
      L[off + c + c * s] = (T)((double)gnodenbr * (double)gnodenbr);
      if (U && facto == 2)
        for (r = c + 1; r < w; r++) L[off + c + r * s] = U[off + r + c * s];
    }
    off += s * w;
  }
  return 0;
}

/*
 * Triangular solves on the factored panels (context only: restates the data flow of
 * up_down_smp, updo.c:114, for a single right-hand side in permuted numbering):
 * forward L y = b (unit diagonal for LDLt/LU... LU uses L with the U diagonal, see below),
 * diagonal (LDLt), backward.  Used by tests for end-to-end residuals.
 *   LLt : L L^T x = b
 *   LDLt: L D L^T x = b   (unit-lower L, D on the diagonal of the diagonal bloks)
 *   LU  : coeftab holds L (unit lower, strictly below the diagonal of the diagonal blok) and
 *         U's diagonal+upper part inside the diagonal blok; ucoeftab holds U^T off-diagonal
 *         panels, already scaled (U^T unit-diagonal convention of kernel_trsm: L carries the
 *         pivots: L_off = A U_d^{-1} non-unit, U_off = A' L_d^{-T} unit).
 */
int NAME(solve)(int facto, i64 cblknbr, const i64 *cblk4, const i64 *blok4,
                const T *L, const T *U, T *x)
{
  i64 k;
  i64 *poff = (i64 *)malloc((cblknbr + 1) * sizeof(i64));
  int herm = (facto == 3);
  poff[0] = 0;
  for (k = 0; k < cblknbr; k++)
    poff[k + 1] = poff[k] + cblk4[4 * k + 3] * (cblk4[4 * k + 1] - cblk4[4 * k] + 1);
  /* forward */
  for (k = 0; k < cblknbr; k++) {
    i64 fc = cblk4[4 * k], w = cblk4[4 * k + 1] - fc + 1, s = cblk4[4 * k + 3];
    i64 fb = cblk4[4 * k + 2], lb = cblk4[4 * (k + 1) + 2], b, c, r;
    const T *Lk = L + poff[k];
    for (c = 0; c < w; c++) {
      if (facto == 0) x[fc + c] /= Lk[c + c * s];
      /* LU: L is the getrf "L" part: unit lower inside the diagonal blok */
      for (r = c + 1; r < w; r++) x[fc + r] -= Lk[r + c * s] * x[fc + c];
    }
    for (b = fb + 1; b < lb; b++) {
      i64 fr = blok4[4 * b], h = blok4[4 * b + 1] - fr + 1, ci = blok4[4 * b + 3];
      for (c = 0; c < w; c++)
        for (r = 0; r < h; r++) x[fr + r] -= Lk[ci + r + c * s] * x[fc + c];
    }
  }
  /* diagonal */
  if (facto == 1 || facto == 3)
    for (k = 0; k < cblknbr; k++) {
      i64 fc = cblk4[4 * k], w = cblk4[4 * k + 1] - fc + 1, s = cblk4[4 * k + 3], c;
      for (c = 0; c < w; c++) x[fc + c] /= L[poff[k] + c + c * s];
    }
  /* backward */
  for (k = cblknbr - 1; k >= 0; k--) {
    i64 fc = cblk4[4 * k], w = cblk4[4 * k + 1] - fc + 1, s = cblk4[4 * k + 3];
    i64 fb = cblk4[4 * k + 2], lb = cblk4[4 * (k + 1) + 2], b, c, r;
    const T *Lk = L + poff[k];
    const T *Bk = (facto == 2) ? U + poff[k] : Lk;     /* LU: rows of U = columns of U^T panels */
    for (b = fb + 1; b < lb; b++) {
      i64 fr = blok4[4 * b], h = blok4[4 * b + 1] - fr + 1, ci = blok4[4 * b + 3];
      for (c = 0; c < w; c++) {
        T acc = 0;
        for (r = 0; r < h; r++) {
          T l = Bk[ci + r + c * s];
          if (herm) l = CONJ(l);
          acc += l * x[fr + r];
        }
        x[fc + c] -= acc;
      }
    }
    for (c = w - 1; c >= 0; c--) {
      T acc = x[fc + c];
      for (r = c + 1; r < w; r++) {
        /* LLt/LDLt: L^T ; LU: U upper part stored in the diagonal blok of coeftab */
        T l = (facto == 2) ? Lk[c + r * s] : Lk[r + c * s];
        if (herm) l = CONJ(l);
        acc -= l * x[fc + r];
      }
      if (facto == 0 || facto == 2) acc /= Lk[c + c * s];
      x[fc + c] = acc;
    }
  }
  free(poff);
  return 0;
}
