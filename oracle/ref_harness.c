/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (oracle side).
 *
 * Driver around the REAL reference (PaStiX 5.2.2.16, compiled from /root/reference by
 * oracle/build_ref.sh, never copied).  It links against the reference objects and
 * includes the reference's internal headers only to READ pastix_data->solvmatr
 * (src/sopalin/src/pastixstr.h:47-111, src/blend/src/solver.h:94-168).
 *
 * Two uses:
 *   dump:  run ordering..analysis, then the reference's own fill (pastix_fillin_csc
 *          pastix.c:3136 + Csc2solv_cblk csc_intern_solve.c:65) and the reference's own
 *          numerical factorization (pastix_task_sopalin pastix.c:3439), and write the
 *          CSC, permutation, SolverMatrix layout, pre-factor panels, factored panels,
 *          and scalars to a raw binary file (packed to .npz by tests/golden/make_golden.py).
 *   time:  same pipeline without dumping; prints one JSON line with DPARM_FACT_FLOPS /
 *          DPARM_FACT_TIME (bench.py's cpu_baseline kind="reference").
 *   cmp:   (the *_amd build) ONE analysis, then the numerical factorization twice on the same SolverMatrix: the
 *          reference's CPU engine, whose factors are kept, and the MI355X engine through integration/sopalin_amd_stub.h;
 *          prints max |L_gpu - L_ref| / max |L_ref| over the entries that belong to the factor (and the same for U),
 *          the static-pivot counts and inertias of both -- parity at sizes no fixture file can hold.
 *
 * usage: ref_harness {dump|time|amd|cmp} {lap3d|lap1d|rlap3d|mtx} ARG {llt|ldlt|lu|ldlh} THREADS OUT [minbs maxbs]
 *        (rlap3d = 3-D 7-point pattern with deterministic pseudo-random values:
 *         SPD for llt/ldlt, unsymmetric diagonally dominant for lu)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#ifdef TYPE_COMPLEX
#include <complex.h>
#endif

#include "common_pastix.h"
#include "nompi.h"
#include "sopalin_define.h"
#include "dof.h"
#include "ftgt.h"
#include "symbol.h"
#include "csc.h"
#include "updown.h"
#include "queue.h"
#include "bulles.h"
#include "solver.h"
#include "sopalin_thread.h"
#include "stack.h"
#include "sopalin3d.h"
#include "order.h"
#include "pastixstr.h"
typedef struct pastix_data_t pastix_data_t;

/* printf is compiled to a no-op inside TUs that include common_pastix.h (:727) */
#define OUT(...) fprintf(stdout, __VA_ARGS__)

void pastix(pastix_data_t **pastix_data, MPI_Comm pastix_comm,
            PASTIX_INT n, PASTIX_INT *colptr, PASTIX_INT *row,
            PASTIX_FLOAT *avals, PASTIX_INT *perm, PASTIX_INT *invp, PASTIX_FLOAT *b, PASTIX_INT rhs,
            PASTIX_INT *iparm, double *dparm);
int pastix_fillin_csc(pastix_data_t *pastix_data, MPI_Comm pastix_comm, PASTIX_INT n,
                      PASTIX_INT *colptr, PASTIX_INT *row, PASTIX_FLOAT *avals,
                      PASTIX_FLOAT *b, PASTIX_INT nrhs, PASTIX_INT *loc2glob);
void Csc2solv_cblk(const CscMatrix *cscmtx, SolverMatrix *datacode,
                   PASTIX_FLOAT *trandcsc, PASTIX_INT itercblk);

/* ------------------------------------------------------------------ */
/* deterministic pseudo-random numbers (LCG), uniform in [0,1)          */
static uint64_t lcg_state = 88172645463325252ULL;
static double lcg(void)
{
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}

/* ------------------------------------------------------------------ */
/* geometric nested dissection of an nx*ny*nz grid:
 * split the longest axis at its midpoint, recurse on both halves, number the
 * separator plane last; boxes of <= 8 nodes are numbered lexicographically. */
static long nd_next;
/* REF_ORDER_CONTIG=1 (timing runs only; the golden fixtures were dumped without it): the nodes of a separator plane
 * are numbered in the order in which their neighbours on the low side were numbered instead of lexicographically, so
 * that every descendant box touches a CONTIGUOUS range of the separator -- the same nested dissection, far fewer and
 * taller bloks (80^3: 1.0e5 instead of 2.9e5).  It is an input of pastix() (API_ORDER_PERSONAL), not a change to it. */
static int nd_contig = 0;
static PASTIX_INT *nd_perm = NULL;   /* old -> new, filled as numbers are handed out */
typedef struct { long key, id; } nd_pair_t;
static int nd_cmp(const void *a, const void *b)
{
  const nd_pair_t *x = a, *y = b;
  return x->key < y->key ? -1 : x->key > y->key ? 1 : 0;
}
static void nd_put(PASTIX_INT *invp, long id) { nd_perm[id] = nd_next; invp[nd_next++] = id; }
static void nd_sep(PASTIX_INT *invp, nd_pair_t *sp, long cnt)
{
  long i;
  if (nd_contig) qsort(sp, cnt, sizeof(nd_pair_t), nd_cmp);
  for (i = 0; i < cnt; i++) nd_put(invp, sp[i].id);
  free(sp);
}
static void nd_rec(int x0, int x1, int y0, int y1, int z0, int z1,
                   int NX, int NY, PASTIX_INT *invp /* new->old */)
{
  int dx = x1 - x0, dy = y1 - y0, dz = z1 - z0;
  long cnt = (long)dx * dy * dz;
  int x, y, z;
  long q = 0;
  nd_pair_t *sp;
#define ND_ID(x, y, z) ((x) + (long)NX * ((y) + (long)NY * (z)))
  if (cnt <= 0) return;
  if (cnt <= 8) {
    for (z = z0; z < z1; z++) for (y = y0; y < y1; y++) for (x = x0; x < x1; x++) nd_put(invp, ND_ID(x, y, z));
    return;
  }
  if (dx >= dy && dx >= dz) {
    int m = x0 + dx / 2;
    nd_rec(x0, m, y0, y1, z0, z1, NX, NY, invp);
    nd_rec(m + 1, x1, y0, y1, z0, z1, NX, NY, invp);
    sp = malloc((size_t)dy * dz * sizeof(nd_pair_t));
    for (z = z0; z < z1; z++) for (y = y0; y < y1; y++) {
      sp[q].id = ND_ID(m, y, z);
      sp[q].key = m > x0 ? nd_perm[ND_ID(m - 1, y, z)] : sp[q].id;
      q++;
    }
    nd_sep(invp, sp, q);
  } else if (dy >= dz) {
    int m = y0 + dy / 2;
    nd_rec(x0, x1, y0, m, z0, z1, NX, NY, invp);
    nd_rec(x0, x1, m + 1, y1, z0, z1, NX, NY, invp);
    sp = malloc((size_t)dx * dz * sizeof(nd_pair_t));
    for (z = z0; z < z1; z++) for (x = x0; x < x1; x++) {
      sp[q].id = ND_ID(x, m, z);
      sp[q].key = m > y0 ? nd_perm[ND_ID(x, m - 1, z)] : sp[q].id;
      q++;
    }
    nd_sep(invp, sp, q);
  } else {
    int m = z0 + dz / 2;
    nd_rec(x0, x1, y0, y1, z0, m, NX, NY, invp);
    nd_rec(x0, x1, y0, y1, m + 1, z1, NX, NY, invp);
    sp = malloc((size_t)dx * dy * sizeof(nd_pair_t));
    for (y = y0; y < y1; y++) for (x = x0; x < x1; x++) {
      sp[q].id = ND_ID(x, y, m);
      sp[q].key = m > z0 ? nd_perm[ND_ID(x, y, m - 1)] : sp[q].id;
      q++;
    }
    nd_sep(invp, sp, q);
  }
#undef ND_ID
}

/* ------------------------------------------------------------------ */
typedef struct {
  PASTIX_INT n, nnz;
  PASTIX_INT *colptr, *rows;   /* 1-based */
  PASTIX_FLOAT *vals;
  int sym;                     /* 1: lower triangle only */
} csc_t;

/* 7-point stencil on N^3 (or N x 1 x 1 for 1-D).  full=0: lower triangle (sym),
 * full=1: both triangles.  rnd=0: Laplacian values (diag D, off -1),
 * rnd=1: pseudo-random values (symmetric if !full, unsymmetric if full), diagonally dominant. */
static void gen_stencil(csc_t *A, int NX, int NY, int NZ, int full, int rnd, double diagv)
{
  long n = (long)NX * NY * NZ, i, nnz = 0;
  int x, y, z, k;
  static const int dxs[6] = {-1, 1, 0, 0, 0, 0};
  static const int dys[6] = {0, 0, -1, 1, 0, 0};
  static const int dzs[6] = {0, 0, 0, 0, -1, 1};
  A->n = n; A->sym = !full;
  A->colptr = malloc((n + 1) * sizeof(PASTIX_INT));
  A->rows = malloc(7 * n * sizeof(PASTIX_INT));
  A->vals = malloc(7 * n * sizeof(PASTIX_FLOAT));
  /* off-diagonal value table so that symmetric draws are consistent: val(i,j) from hash */
  for (i = 0; i < n; i++) {
    long nb[7]; int cnt = 0;
    x = i % NX; y = (i / NX) % NY; z = i / ((long)NX * NY);
    A->colptr[i] = nnz + 1;
    nb[cnt++] = i;
    for (k = 0; k < 6; k++) {
      int xx = x + dxs[k], yy = y + dys[k], zz = z + dzs[k];
      long j;
      if (xx < 0 || xx >= NX || yy < 0 || yy >= NY || zz < 0 || zz >= NZ) continue;
      j = xx + (long)NX * (yy + (long)NY * zz);
      if (!full && j < i) continue;
      nb[cnt++] = j;
    }
    /* sort rows ascending */
    { int a, b; for (a = 1; a < cnt; a++) { long v = nb[a]; for (b = a - 1; b >= 0 && nb[b] > v; b--) nb[b + 1] = nb[b]; nb[b + 1] = v; } }
    for (k = 0; k < cnt; k++) {
      long j = nb[k];
      double v;
      if (j == i) v = diagv;
      else if (!rnd) v = -1.0;
      else {
        /* deterministic per-(unordered or ordered) pair value in (-1.5,-0.5) */
        uint64_t a = (uint64_t)(full ? i : (i < j ? i : j)), b = (uint64_t)(full ? j : (i < j ? j : i));
        uint64_t h = (a * 0x9E3779B97F4A7C15ULL) ^ (b * 0xC2B2AE3D27D4EB4FULL + 0x165667B19E3779F9ULL);
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
        v = -0.5 - (double)(h >> 11) / 9007199254740992.0;
      }
      A->rows[nnz] = j + 1;
#ifdef TYPE_COMPLEX
      A->vals[nnz] = v + ((j == i || !rnd) ? 0.0 : 0.25 * v) * I;
#else
      A->vals[nnz] = v;
#endif
      nnz++;
    }
  }
  A->colptr[n] = nnz + 1;
  A->nnz = nnz;
  if (rnd) { /* diagonal = 1 + sum |offdiag| of the full row/col + small random */
    double *s = calloc(n, sizeof(double));
    long j;
    for (j = 0; j < n; j++)
      for (i = A->colptr[j] - 1; i < A->colptr[j + 1] - 1; i++) {
        long r = A->rows[i] - 1;
        if (r == j) continue;
#ifdef TYPE_COMPLEX
        s[j] += cabs(A->vals[i]); if (!full) s[r] += cabs(A->vals[i]);
#else
        s[j] += fabs(A->vals[i]); if (!full) s[r] += fabs(A->vals[i]);
#endif
      }
    for (j = 0; j < n; j++)
      for (i = A->colptr[j] - 1; i < A->colptr[j + 1] - 1; i++)
        if (A->rows[i] - 1 == j) A->vals[i] = s[j] + 1.0 + lcg();
    free(s);
  }
}

/* MatrixMarket coordinate (complex|real) (symmetric|general), 1-based */
static int read_mtx(csc_t *A, const char *fn)
{
  FILE *f = fopen(fn, "r");
  char line[1024];
  int cplx, sym;
  long M, N, NZ, k;
  long *ri, *ci; double *vr, *vi;
  if (!f) return -1;
  if (!fgets(line, sizeof line, f)) return -1;
  cplx = strstr(line, "complex") != NULL;
  sym = strstr(line, "symmetric") != NULL;
  do { if (!fgets(line, sizeof line, f)) return -1; } while (line[0] == '%');
  sscanf(line, "%ld %ld %ld", &M, &N, &NZ);
  ri = malloc(NZ * sizeof(long)); ci = malloc(NZ * sizeof(long));
  vr = malloc(NZ * sizeof(double)); vi = calloc(NZ, sizeof(double));
  for (k = 0; k < NZ; k++) {
    if (cplx) { if (fscanf(f, "%ld %ld %lf %lf", &ri[k], &ci[k], &vr[k], &vi[k]) != 4) return -1; }
    else      { if (fscanf(f, "%ld %ld %lf", &ri[k], &ci[k], &vr[k]) != 3) return -1; }
    if (sym && ri[k] < ci[k]) { long t = ri[k]; ri[k] = ci[k]; ci[k] = t; }
  }
  fclose(f);
  A->n = N; A->nnz = NZ; A->sym = sym;
  A->colptr = calloc(N + 2, sizeof(PASTIX_INT));
  A->rows = malloc(NZ * sizeof(PASTIX_INT));
  A->vals = malloc(NZ * sizeof(PASTIX_FLOAT));
  for (k = 0; k < NZ; k++) A->colptr[ci[k]]++;
  { long acc = 1, j; for (j = 1; j <= N + 1; j++) { long c = A->colptr[j]; A->colptr[j] = acc; acc += c; } }
  /* colptr[j] (1..N) now start (1-based) of column j; shift to 0-index array */
  { long *pos = malloc((N + 2) * sizeof(long)), j;
    for (j = 1; j <= N; j++) pos[j] = A->colptr[j] - 1;
    for (k = 0; k < NZ; k++) {
      long p = pos[ci[k]]++;
      A->rows[p] = ri[k];
#ifdef TYPE_COMPLEX
      A->vals[p] = vr[k] + vi[k] * I;
#else
      A->vals[p] = vr[k];
#endif
    }
    for (j = 0; j < N; j++) A->colptr[j] = A->colptr[j + 1];
    A->colptr[N] = NZ + 1;
    /* sort each column by row */
    for (j = 0; j < N; j++) {
      long a, b, s = A->colptr[j] - 1, e = A->colptr[j + 1] - 1;
      for (a = s + 1; a < e; a++) {
        PASTIX_INT r = A->rows[a]; PASTIX_FLOAT v = A->vals[a];
        for (b = a - 1; b >= s && A->rows[b] > r; b--) { A->rows[b + 1] = A->rows[b]; A->vals[b + 1] = A->vals[b]; }
        A->rows[b + 1] = r; A->vals[b + 1] = v;
      }
    }
    free(pos);
  }
  free(ri); free(ci); free(vr); free(vi);
  return 0;
}

/* ------------------------------------------------------------------ */
static void w64(FILE *f, int64_t v) { fwrite(&v, 8, 1, f); }
static void wd(FILE *f, double v) { fwrite(&v, 8, 1, f); }
static void wints(FILE *f, const PASTIX_INT *p, long cnt, long add)
{
  long i;
  for (i = 0; i < cnt; i++) w64(f, (int64_t)p[i] + add);
}

/* set by oracle/ref_amd_sopalin3d.c (the *_amd build): factorizations handed to the MI355X engine */
int pastix_amd_hook_calls = 0;
int pastix_amd_hook_last_rc = 0;
double pastix_amd_hook_wall = 0;        /* wall time of the last numerical-factorization call at the hooked launch: the GPU
                                           engine's whole replacement (plan + host <-> device + factorization) or the CPU engine's threads */

int main(int argc, char **argv)
{
  pastix_data_t *pd = NULL;
  PASTIX_INT iparm[IPARM_SIZE];
  double dparm[DPARM_SIZE];
  csc_t A;
  PASTIX_INT *perm, *invp;
  PASTIX_FLOAT *b, *bsave;
  int dump, facto, nthr, N = 0;
  const char *kind, *out;
  long i, n;
  SolverMatrix *sm;
  double critere, resid = -1;
  int minbs = -1, maxbs = -1;

  if (argc < 7) {
    fprintf(stderr, "usage: %s {dump|time|amd|cmp} {lap3d|lap1d|rlap3d|mtx} ARG {llt|ldlt|lu|ldlh} THREADS OUT [minbs maxbs]\n", argv[0]);
    return 2;
  }
  const int cmp = !strcmp(argv[1], "cmp");
  dump = !strcmp(argv[1], "dump");
  /* mode "amd" (the *_amd build): like "time", with the numerical factorization on the MI355X engine */
  if (!strcmp(argv[1], "amd")) setenv("PASTIX_AMD_ENGINE", "1", 1);
  kind = argv[2];
  facto = !strcmp(argv[4], "llt") ? API_FACT_LLT : !strcmp(argv[4], "ldlt") ? API_FACT_LDLT
        : !strcmp(argv[4], "lu") ? API_FACT_LU : API_FACT_LDLH;
  nthr = atoi(argv[5]);
  out = argv[6];
  if (argc >= 9) { minbs = atoi(argv[7]); maxbs = atoi(argv[8]); }

  if (!strcmp(kind, "lap3d") || !strcmp(kind, "rlap3d")) {
    N = atoi(argv[3]);
    gen_stencil(&A, N, N, N, facto == API_FACT_LU, kind[0] == 'r', 6.0);
  } else if (!strcmp(kind, "lap1d")) {
    /* the reference's own -lap generator: diag 2, sub-diagonal -1 (laplacian.c:151-181) */
    N = atoi(argv[3]);
    gen_stencil(&A, N, 1, 1, facto == API_FACT_LU, 0, 2.0);
  } else {
    if (read_mtx(&A, argv[3])) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
    if (facto == API_FACT_LU && A.sym) { fprintf(stderr, "LU needs a general matrix\n"); return 2; }
  }
  n = A.n;

  /* ordering: geometric ND for grids, identity otherwise (1-based, same base as CSC) */
  perm = malloc(n * sizeof(PASTIX_INT));
  invp = malloc(n * sizeof(PASTIX_INT));
  if (kind[0] == 'l' || kind[0] == 'r') {
    nd_next = 0;
    nd_perm = perm;
    nd_contig = getenv("REF_ORDER_CONTIG") != NULL;
    if (!strcmp(kind, "lap1d")) nd_rec(0, N, 0, 1, 0, 1, N, 1, invp);
    else nd_rec(0, N, 0, N, 0, N, N, N, invp);
    for (i = 0; i < n; i++) perm[invp[i]] = i;
  } else {
    for (i = 0; i < n; i++) perm[i] = invp[i] = i;
  }
  for (i = 0; i < n; i++) { perm[i] += 1; invp[i] += 1; }

  /* rhs: b[i] = rand()/RAND_MAX with srand(1) */
  b = malloc(n * sizeof(PASTIX_FLOAT));
  bsave = malloc(n * sizeof(PASTIX_FLOAT));
  srand(1);
  for (i = 0; i < n; i++) bsave[i] = b[i] = (double)rand() / RAND_MAX;

  /* (1) defaults */
  iparm[IPARM_MODIFY_PARAMETER] = API_NO;
  pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
  /* (2) parameters (SURVEY 8c harness sequence) */
  iparm[IPARM_THREAD_NBR] = nthr;
  iparm[IPARM_SYM] = A.sym ? (facto == API_FACT_LDLH ? API_SYM_HER : API_SYM_YES) : API_SYM_NO;
  iparm[IPARM_FACTORIZATION] = facto;
  iparm[IPARM_MATRIX_VERIFICATION] = API_NO;
  iparm[IPARM_ORDERING] = API_ORDER_PERSONAL;
  iparm[IPARM_LEVEL_OF_FILL] = -1;
  iparm[IPARM_BINDTHRD] = API_BIND_NO;
  iparm[IPARM_VERBOSE] = API_VERBOSE_NOT;
  iparm[IPARM_RHS_MAKING] = API_RHS_B;
  iparm[IPARM_ITERMAX] = 0;
  if (minbs > 0) { iparm[IPARM_MIN_BLOCKSIZE] = minbs; iparm[IPARM_MAX_BLOCKSIZE] = maxbs; }
  /* REF_FAKE=1: the reference's structure-only "fake factorisation" (IPARM_FILL_MATRIX = API_YES, coefinit.c:343-443):
   * the CSC values are not used by the numerical step */
  const int fake = getenv("REF_FAKE") != NULL;
  if (fake) iparm[IPARM_FILL_MATRIX] = API_YES;
  iparm[IPARM_START_TASK] = API_TASK_ORDERING;
  iparm[IPARM_END_TASK] = API_TASK_ANALYSE;
  pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
  sm = &pd->solvmatr;

  FILE *f = NULL;
  if (dump) {
    long k, coefnbr = 0;
    f = fopen(out, "wb");
    if (!f) { perror(out); return 2; }
    for (k = 0; k < sm->cblknbr; k++)
      coefnbr += (long)sm->cblktab[k].stride * (sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1);
    fwrite("PSTXFIX1", 8, 1, f);
    w64(f, n); w64(f, A.nnz); w64(f, facto);
#ifdef TYPE_COMPLEX
    w64(f, 1);
#else
    w64(f, 0);
#endif
    w64(f, sm->cblknbr); w64(f, sm->bloknbr); w64(f, coefnbr); w64(f, sm->coefmax);
    w64(f, A.sym); w64(f, nthr); w64(f, iparm[IPARM_MIN_BLOCKSIZE]); w64(f, iparm[IPARM_MAX_BLOCKSIZE]);
    wints(f, A.colptr, n + 1, 0);
    wints(f, A.rows, A.nnz, 0);
    fwrite(A.vals, sizeof(PASTIX_FLOAT), A.nnz, f);
    /* the ordering actually used by the analysis (kass re-permutes inside amalgamated
     * supernodes): pastix_data->ordemesh, 0-based after the symbolic step */
    wints(f, pd->ordemesh.permtab, n, 0);
    wints(f, pd->ordemesh.peritab, n, 0);
    for (k = 0; k <= sm->cblknbr; k++) {
      w64(f, sm->cblktab[k].fcolnum); w64(f, sm->cblktab[k].lcolnum);
      w64(f, sm->cblktab[k].bloknum); w64(f, sm->cblktab[k].stride);
    }
    for (k = 0; k < sm->bloknbr; k++) {
      w64(f, sm->bloktab[k].frownum); w64(f, sm->bloktab[k].lrownum);
      w64(f, sm->bloktab[k].cblknum); w64(f, sm->bloktab[k].coefind);
    }
    /* task order of thread 0.. (ttsktab) as cblk numbers */
    w64(f, sm->tasknbr);
    for (k = 0; k < sm->tasknbr; k++) {
      w64(f, sm->tasktab[k].cblknum); w64(f, sm->tasktab[k].prionum); w64(f, sm->tasktab[k].ctrbcnt);
    }

    if (fake) {
      /* input panels of a fake run: what CoefMatrix_Init's else-branch produces inside the sopalin threads (all 1 / all
       * 2, gnodenbr^2 on the diagonals, LU: upper part of coeftab's diagonal bloks = 2); the pin of this case is the
       * factor dump below, which comes from the reference's own fill + factorization */
      critere = dparm[DPARM_EPSILON_MAGN_CTRL] < 0 ? -dparm[DPARM_EPSILON_MAGN_CTRL]
              : ((double)n * (double)n + (double)n) * sqrt(dparm[DPARM_EPSILON_MAGN_CTRL]);
      w64(f, 0);                                   /* no internal CSC: zero entries, n zero column counts */
      for (k = 0; k < n; k++) w64(f, 0);
      for (k = 0; k < sm->cblknbr; k++) {
        long w = sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1, sd = sm->cblktab[k].stride, sz = sd * w, c, r;
        PASTIX_FLOAT *tl = malloc(sz * sizeof(PASTIX_FLOAT)), *tu = malloc(sz * sizeof(PASTIX_FLOAT));
        for (c = 0; c < sz; c++) { tl[c] = 1; tu[c] = 2; }
        for (c = 0; c < w; c++) {
          tl[c + c * sd] = (PASTIX_FLOAT)((double)n * (double)n);
          if (facto == API_FACT_LU) for (r = c + 1; r < w; r++) tl[c + r * sd] = tu[r + c * sd];
        }
        fwrite(tl, sizeof(PASTIX_FLOAT), sz, f);
        if (facto == API_FACT_LU) fwrite(tu, sizeof(PASTIX_FLOAT), sz, f);
        free(tl); free(tu);
      }
    } else {
    /* pre-factor panels through the reference's own fill code */
    pastix_fillin_csc(pd, pd->pastix_comm, n, A.colptr, A.rows, A.vals, b, 1, NULL);
    pd->cscInternFilled = API_YES;
    {
      /* critere = ||A||_1 * sqrt(eps)  (sopalin3d.c:586-606, csc_intern_compute.c:120) */
      const CscMatrix *csc = &pd->cscmtx;
      double themax = 0;
      PASTIX_INT ib, ic, iv;
      for (ib = 0; ib < CSC_FNBR(csc); ib++)
        for (ic = 0; ic < CSC_COLNBR(csc, ib); ic++) {
          double s = 0;
          for (iv = CSC_COL(csc, ib, ic); iv < CSC_COL(csc, ib, ic + 1); iv++)
            s += ABS_FLOAT(CSC_VAL(csc, iv));
          if (s > themax) themax = s;
        }
      critere = dparm[DPARM_EPSILON_MAGN_CTRL] < 0 ? -dparm[DPARM_EPSILON_MAGN_CTRL]
              : themax * sqrt(dparm[DPARM_EPSILON_MAGN_CTRL]);
      /* internal (permuted, symmetrized) CSC, for pinning the restated fill */
      {
        long nnzi = 0;
        for (ib = 0; ib < CSC_FNBR(csc); ib++)
          for (ic = 0; ic < CSC_COLNBR(csc, ib); ic++)
            nnzi += CSC_COL(csc, ib, ic + 1) - CSC_COL(csc, ib, ic);
        w64(f, nnzi);
        /* per global column: count then (row,val) */
        for (ib = 0; ib < CSC_FNBR(csc); ib++)
          for (ic = 0; ic < CSC_COLNBR(csc, ib); ic++) {
            w64(f, CSC_COL(csc, ib, ic + 1) - CSC_COL(csc, ib, ic));
          }
        for (ib = 0; ib < CSC_FNBR(csc); ib++)
          for (ic = 0; ic < CSC_COLNBR(csc, ib); ic++)
            for (iv = CSC_COL(csc, ib, ic); iv < CSC_COL(csc, ib, ic + 1); iv++)
              w64(f, CSC_ROW(csc, iv));
        for (ib = 0; ib < CSC_FNBR(csc); ib++)
          for (ic = 0; ic < CSC_COLNBR(csc, ib); ic++)
            for (iv = CSC_COL(csc, ib, ic); iv < CSC_COL(csc, ib, ic + 1); iv++)
              fwrite(&CSC_VAL(csc, iv), sizeof(PASTIX_FLOAT), 1, f);
      }
    }
    for (k = 0; k < sm->cblknbr; k++) {
      long sz = (long)sm->cblktab[k].stride * (sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1);
      PASTIX_FLOAT *sv = sm->cblktab[k].coeftab, *su = sm->cblktab[k].ucoeftab;
      PASTIX_FLOAT *tl = calloc(sz, sizeof(PASTIX_FLOAT)), *tu = calloc(sz, sizeof(PASTIX_FLOAT));
      sm->cblktab[k].coeftab = tl; sm->cblktab[k].ucoeftab = tu;
      Csc2solv_cblk(&pd->cscmtx, sm, facto == API_FACT_LU ? pd->sopar.transcsc : NULL, k);
      fwrite(tl, sizeof(PASTIX_FLOAT), sz, f);
      if (facto == API_FACT_LU) fwrite(tu, sizeof(PASTIX_FLOAT), sz, f);
      sm->cblktab[k].coeftab = sv; sm->cblktab[k].ucoeftab = su;
      free(tl); free(tu);
    }
    }
  }

  /* (3) the reference's numerical factorization */
  iparm[IPARM_START_TASK] = API_TASK_NUMFACT;
  iparm[IPARM_END_TASK] = API_TASK_NUMFACT;
  pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
  if (cmp) {
    /* keep the CPU engine's factors, factorize again through the stub (the reference re-fills the panels from its
     * internal CSC at every numerical factorization), compare entry by entry.  Compared: `po` / `sy` / `he` -- the lower
     * triangle of every diagonal blok and all rows below (the strict upper triangle is not part of the factor and holds
     * by-products of the rectangular scatter, sopalin_compute.c:427-452); `ge` -- coeftab and ucoeftab in full. */
    long k, c, r;
    const long nc = sm->cblknbr;
    PASTIX_FLOAT **refL = malloc(nc * sizeof(*refL)), **refU = malloc(nc * sizeof(*refU));
    const double t_ref = dparm[DPARM_FACT_TIME];
    const double wall_ref = pastix_amd_hook_wall;
    const long npiv_ref = iparm[IPARM_STATIC_PIVOTING], inertia_ref = iparm[IPARM_INERTIA];
    double maxL = 0, maxU = 0, dL = 0, dU = 0;
    long worst_k = -1;
    for (k = 0; k < nc; k++) {
      const size_t sz = (size_t)sm->cblktab[k].stride * (sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1);
      refL[k] = malloc(sz * sizeof(PASTIX_FLOAT));
      memcpy(refL[k], sm->cblktab[k].coeftab, sz * sizeof(PASTIX_FLOAT));
      refU[k] = NULL;
      if (facto == API_FACT_LU) {
        refU[k] = malloc(sz * sizeof(PASTIX_FLOAT));
        memcpy(refU[k], sm->cblktab[k].ucoeftab, sz * sizeof(PASTIX_FLOAT));
      }
    }
    setenv("PASTIX_AMD_ENGINE", "1", 1);
    iparm[IPARM_START_TASK] = API_TASK_NUMFACT;      /* (pastix() advances START_TASK past the step it has run) */
    iparm[IPARM_END_TASK] = API_TASK_NUMFACT;
    pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
    sm = &pd->solvmatr;
    for (k = 0; k < nc; k++) {
      const long w = sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1, sd = sm->cblktab[k].stride;
      const PASTIX_FLOAT *gl = sm->cblktab[k].coeftab, *gu = sm->cblktab[k].ucoeftab;
      for (c = 0; c < w; c++)
        for (r = (facto == API_FACT_LU ? 0 : c); r < sd; r++) {
          const double a = ABS_FLOAT(refL[k][r + c * sd]), d = ABS_FLOAT(gl[r + c * sd] - refL[k][r + c * sd]);
          if (a > maxL) maxL = a;
          if (!(d <= dL)) { dL = d; worst_k = k; }               /* (a NaN counts) */
        }
      if (facto == API_FACT_LU)
        for (c = 0; c < w; c++)
          for (r = 0; r < sd; r++) {
            const double a = ABS_FLOAT(refU[k][r + c * sd]), d = ABS_FLOAT(gu[r + c * sd] - refU[k][r + c * sd]);
            if (a > maxU) maxU = a;
            if (!(d <= dU)) dU = d;
          }
      free(refL[k]); free(refU[k]);
    }
    free(refL); free(refU);
    OUT("{\"cmp\": 1, \"maxabs_L\": %.6e, \"maxdiff_L\": %.6e, \"rel_L\": %.3e, \"maxabs_U\": %.6e, \"maxdiff_U\": %.6e, "
        "\"rel_U\": %.3e, \"worst_cblk\": %ld, \"static_pivots_ref\": %ld, \"static_pivots_gpu\": %ld, \"inertia_ref\": %ld, "
        "\"inertia_gpu\": %ld, \"time_ref\": %.6f, \"time_gpu\": %.6f, \"wall_ref\": %.6f, \"wall_gpu\": %.6f, \"gpu_engine_calls\": %d, \"gpu_engine_rc\": %d}\n",
        maxL, dL, dL / (maxL > 0 ? maxL : 1), maxU, dU, dU / (maxU > 0 ? maxU : 1), worst_k, npiv_ref,
        (long)iparm[IPARM_STATIC_PIVOTING], inertia_ref, (long)iparm[IPARM_INERTIA], t_ref, dparm[DPARM_FACT_TIME],
        wall_ref, pastix_amd_hook_wall, pastix_amd_hook_calls, pastix_amd_hook_last_rc);
  }
  {
    double flops = dparm[DPARM_FACT_FLOPS], t = dparm[DPARM_FACT_TIME];
    long nnzl = iparm[IPARM_NNZEROS], npiv = iparm[IPARM_STATIC_PIVOTING];
    if (dump) {
      long k;
      for (k = 0; k < sm->cblknbr; k++) {
        long sz = (long)sm->cblktab[k].stride * (sm->cblktab[k].lcolnum - sm->cblktab[k].fcolnum + 1);
        fwrite(sm->cblktab[k].coeftab, sizeof(PASTIX_FLOAT), sz, f);
        if (facto == API_FACT_LU) fwrite(sm->cblktab[k].ucoeftab, sizeof(PASTIX_FLOAT), sz, f);
      }
    }
    /* (4) solve with the reference (no refinement) for an end-to-end residual */
    iparm[IPARM_START_TASK] = API_TASK_SOLVE;
    iparm[IPARM_END_TASK] = API_TASK_SOLVE;
    pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
    {
      /* r = A x - b */
      PASTIX_FLOAT *r = calloc(n, sizeof(PASTIX_FLOAT));
      double nr = 0, nb = 0;
      long j, p;
      for (j = 0; j < n; j++)
        for (p = A.colptr[j] - 1; p < A.colptr[j + 1] - 1; p++) {
          long ii = A.rows[p] - 1;
          r[ii] += A.vals[p] * b[j];
#ifdef TYPE_COMPLEX
          if (A.sym && ii != j) r[j] += (facto == API_FACT_LDLH ? conj(A.vals[p]) : A.vals[p]) * b[ii];
#else
          if (A.sym && ii != j) r[j] += A.vals[p] * b[ii];
#endif
        }
      for (j = 0; j < n; j++) {
        double d = ABS_FLOAT(r[j] - bsave[j]), e = ABS_FLOAT(bsave[j]);
        nr += d * d; nb += e * e;
      }
      resid = sqrt(nr / nb);
      free(r);
    }
    if (dump) {
      wd(f, critere); wd(f, flops); wd(f, t); wd(f, resid);
      w64(f, npiv); w64(f, nnzl); w64(f, iparm[IPARM_INERTIA]);
      fwrite(bsave, sizeof(PASTIX_FLOAT), n, f);
      fwrite(b, sizeof(PASTIX_FLOAT), n, f);
      fclose(f);
    }
    OUT("{\"kind\": \"%s\", \"arg\": \"%s\", \"facto\": \"%s\", \"n\": %ld, \"threads\": %d, "
        "\"cblknbr\": %ld, \"bloknbr\": %ld, \"nnzl\": %ld, \"flops\": %.6e, \"time\": %.6f, "
        "\"gflops\": %.3f, \"static_pivots\": %ld, \"residual\": %.3e, \"gpu_engine_calls\": %d, "
        "\"gpu_engine_rc\": %d, \"inertia\": %ld, \"wall_sopalin\": %.6f}\n",
        kind, argv[3], argv[4], n, nthr, (long)sm->cblknbr, (long)sm->bloknbr, nnzl, flops, t,
        flops / t * 1e-9, npiv, resid, pastix_amd_hook_calls, pastix_amd_hook_last_rc, (long)iparm[IPARM_INERTIA], pastix_amd_hook_wall);
  }
  iparm[IPARM_START_TASK] = API_TASK_CLEAN;
  iparm[IPARM_END_TASK] = API_TASK_CLEAN;
  pastix(&pd, 0, n, A.colptr, A.rows, A.vals, perm, invp, b, 1, iparm, dparm);
  return 0;
}
