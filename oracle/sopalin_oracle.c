/*
 * sopalin_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, no BLAS) of the reference's numerical factorization path
 * (PaStiX 5.2.2.16 sopalin: compute_1d and what it calls).  It is the CHECKER for the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (pastix_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  The reference ships no golden vectors for this path;
 * the oracle is pinned against outputs of the reference itself, compiled here from
 * /root/reference by oracle/build_ref.sh (oracle/_ref/ref_harness_*) and committed as
 * tests/golden/ (.npz files) by tests/golden/make_golden.py (tests/test_oracle_golden.py).
 *
 * Exports (C ABI, i64 = int64_t everywhere):
 *   oracle_{d,z}sopalin / oracle_{d,z}fill / oracle_{d,z}fill_fake / oracle_{d,z}solve   (sopalin_oracle_impl.h)
 *   oracle_fact_flops                                            (below)
 */
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <complex.h>

typedef int64_t i64;

#define T double
#define NAME(x) oracle_d##x
#define CONJ(x) (x)
#define ABS(x) fabs(x)
#define SQRT(x) sqrt(x)
#define SYR_FULL 0
#include "sopalin_oracle_impl.h"
#undef SYR_FULL
#undef T
#undef NAME
#undef CONJ
#undef ABS
#undef SQRT

#define T double complex
#define NAME(x) oracle_z##x
#define CONJ(x) conj(x)
#define ABS(x) cabs(x)
#define SQRT(x) csqrt(x)
#define SYR_FULL 1
#include "sopalin_oracle_impl.h"
#undef SYR_FULL
#undef T
#undef NAME
#undef CONJ
#undef ABS
#undef SQRT

/*
 * DPARM_FACT_FLOPS as the reference defines it: symbCost (blend_symbol_cost.c:52-88) with
 * flops_dpotrf/zpotrf/dgetrf (:282-430) and the LAPACK-style macros of flops.h:
 *   FMULS_POTRF(n)=n(((1/6)n+1/2)n+1/3)  FADDS_POTRF(n)=n(((1/6)n)n-1/6)    (flops.h:116-117)
 *   FLOPS_DTRSM(Right, M, N): FMULS_TRSM = FADDS_TRSM = FMULS_TRMM = 0.5*M*N*(N+1)
 *          (flops.h:91-100: FADDS_TRSM is defined as FMULS_TRMM)  -> M*N*(N+1) in real
 *   GEMM(m,n,k): mul = add = m n k                                          (flops.h:74-75)
 *   FMULS_GETRF(m,n), FADDS_GETRF(m,n)                                       (flops.h:103-108)
 * Real: mul + add; complex: 6 mul + 2 add                                   (flops.h:211-214)
 * The reference's per-cblk LLt count (blend_symbol_cost.c:382-430): POTRF(N) + TRSM(M,N) +
 * sum over off-diagonal bloks of GEMM(M_rem, h, N) with M_rem = rows from this blok down.
 * LU (:282-330): GETRF(N,N) + 2*TRSM(M,N) + 2*sum GEMM.  LDLt uses the LLt count (:78-86).
 */
static double fmuls_potrf(double n) { return n * (((1. / 6.) * n + 0.5) * n + (1. / 3.)); }
static double fadds_potrf(double n) { return n * (((1. / 6.) * n) * n - (1. / 6.)); }
static double fmuls_getrf(double m, double n)
{
  return (m < n) ? 0.5 * m * (m * (n - (1. / 3.) * m - 1.) + n) + (2. / 3.) * m
                 : 0.5 * n * (n * (m - (1. / 3.) * n - 1.) + m) + (2. / 3.) * n;
}
static double fadds_getrf(double m, double n)
{
  return (m < n) ? 0.5 * m * (m * (n - (1. / 3.) * m) - n) + (1. / 6.) * m
                 : 0.5 * n * (n * (m - (1. / 3.) * n) - m) + (1. / 6.) * n;
}
double oracle_fact_flops(int facto, int is_complex, i64 cblknbr, const i64 *cblk4, const i64 *blok4)
{
  double muls = 0, adds = 0;
  i64 k, b;
  for (k = 0; k < cblknbr; k++) {
    double N = (double)(cblk4[4 * k + 1] - cblk4[4 * k] + 1);
    double M = (double)cblk4[4 * k + 3] - N;
    double rem = M, fac = (facto == 2) ? 2.0 : 1.0;
    if (facto == 2) { muls += fmuls_getrf(N, N); adds += fadds_getrf(N, N); }
    else { muls += fmuls_potrf(N); adds += fadds_potrf(N); }
    muls += fac * 0.5 * M * N * (N + 1.); adds += fac * 0.5 * M * N * (N + 1.); /* FADDS_TRSM == FMULS_TRMM, flops.h:99-100 */
    for (b = cblk4[4 * k + 2] + 1; b < cblk4[4 * (k + 1) + 2]; b++) {
      double h = (double)(blok4[4 * b + 1] - blok4[4 * b] + 1);
      muls += fac * rem * h * N; adds += fac * rem * h * N;
      rem -= h;
    }
  }
  return is_complex ? 6. * muls + 2. * adds : muls + adds;
}
