/*
 * sopalin_amd_stub.h -- what a PaStiX 5.2.2.16 maintainer adds to src/sopalin/src/sopalin3d.c to run the numerical
 * factorization on an MI355X through libpastix_amd.so.  Included from sopalin3d.c (after its own
 * includes), i.e. compiled once per arithmetic and factorization like the file itself; it uses nothing but the
 * reference's own accessor macros (sopalin_acces.h:31-254: SYMB_*, SOLV_*), types (Sopalin_Data_t, SopalinParam,
 * sopalin3d.h:90-330) and build macros (CHOL_SOPALIN / SOPALIN_LU / HERMITIAN / TYPE_COMPLEX / PREC_DOUBLE,
 * sopalin_define.h:453-465).
 *
 * API_CALL(sopalin_amd)(sopalin_data) replaces the call
 *     sopalin_launch_thread(..., API_CALL(sopalin_smp), sopalin_data, ...)        (sopalin3d.c:1411-1416)
 * of sopalin_thread(): same inputs (critere computed by init_struct_sopalin, sopalin3d.c:586-606; the panels are
 * allocated and filled by the reference's threads inside that call -- sopalin_init_smp, sopalin_init.c:1196-1211 -- and
 * the stub keeps exactly that part), same outputs (factors in place in
 * coeftab / ucoeftab for updo.c; sopar->diagchange -> IPARM_STATIC_PIVOTING; DPARM_FACT_TIME; IPARM_INERTIA,
 * sopalin3d.c:1119-1160).  Returns 0, or a PASTIX_AMD_ERR_* code on which the caller falls back to the CPU engine.
 */
#ifndef SOPALIN_AMD_STUB_H
#define SOPALIN_AMD_STUB_H

#include <stdlib.h>

#include "pastix_amd.h"

/* The host part of sopalin_smp that must still run, once per computing thread: sopalin_init_smp allocates and fills the
 * thread's panels (CoefMatrix_Allocate + CoefMatrix_Init, sopalin_init.c:1196-1211 -- with their barriers across the
 * SOLV_THRDNBR threads and NUMA-local first touch, exactly as for the CPU engine) and sopalin_clean_smp releases the
 * thread's work buffers (sopalin3d.c:712 and :1082 bracket the task loop the GPU replaces). */
static void *API_CALL(sopalin_amd_init_smp)(void *arg)
{
  sopthread_data_t *argument     = (sopthread_data_t *)arg;
  Sopalin_Data_t   *sopalin_data = (Sopalin_Data_t *)(argument->data);
  SolverMatrix     *datacode     = sopalin_data->datacode;
  PASTIX_INT        me           = argument->me;
  int               init         = INIT_COMPUTE;

  if (THREAD_FUNNELED_OFF) init = init | INIT_SEND;
  if (THREAD_COMM_OFF)     init = init | INIT_RECV;
  sopalin_init_smp(sopalin_data, me, 1, init);
  SYNCHRO_THREAD;
  sopalin_clean_smp(sopalin_data, me);
  return NULL;
}

static int API_CALL(sopalin_amd)(Sopalin_Data_t *sopalin_data)
{
  SolverMatrix *datacode = sopalin_data->datacode;
  SopalinParam *sopar    = sopalin_data->sopar;
  PASTIX_INT    k, b;
  pastix_amd_layout_t   lay;
  pastix_amd_options_t  opts;
  pastix_amd_stats_t    st;
  pastix_amd_cblk_t    *cb;
  pastix_amd_blok_t    *bl;
  void                **ctab, **utab;
  int                   rc;

  if (SOLV_PROCNBR != 1)                       /* MPI runs keep the CPU engine (the multi-GPU driver is bound apart) */
    return PASTIX_AMD_ERR_UNSUPPORTED;
  /* panels: allocated and filled by the reference's own threads, as sopalin_smp starts (see above) */
  sopalin_launch_thread(sopalin_data, SOLV_PROCNUM, SOLV_PROCNBR, datacode->btree,
                        sopar->iparm[IPARM_VERBOSE],
                        SOLV_THRDNBR, API_CALL(sopalin_amd_init_smp), sopalin_data,
                        0, NULL, NULL, 0, NULL, NULL);
  cb   = (pastix_amd_cblk_t *)malloc((SYMB_CBLKNBR + 1) * sizeof(*cb));
  bl   = (pastix_amd_blok_t *)malloc((SYMB_BLOKNBR > 0 ? SYMB_BLOKNBR : 1) * sizeof(*bl));
  ctab = (void **)malloc(SYMB_CBLKNBR * sizeof(void *));
  utab = (void **)malloc(SYMB_CBLKNBR * sizeof(void *));
  if (!cb || !bl || !ctab || !utab) { free(cb); free(bl); free(ctab); free(utab); return PASTIX_AMD_ERR_ALLOC; }

  for (k = 0; k <= SYMB_CBLKNBR; k++) {
    cb[k].fcolnum = SYMB_FCOLNUM(k);
    cb[k].lcolnum = SYMB_LCOLNUM(k);
    cb[k].bloknum = SYMB_BLOKNUM(k);
    cb[k].stride  = (k < SYMB_CBLKNBR) ? SOLV_STRIDE(k) : 0;
  }
  for (b = 0; b < SYMB_BLOKNBR; b++) {
    bl[b].frownum = SYMB_FROWNUM(b);
    bl[b].lrownum = SYMB_LROWNUM(b);
    bl[b].cblknum = SYMB_CBLKNUM(b);
    bl[b].coefind = SOLV_COEFIND(b);
  }
  for (k = 0; k < SYMB_CBLKNBR; k++) {
    ctab[k] = (void *)SOLV_COEFTAB(k);
    utab[k] = (void *)SOLV_UCOEFTAB(k);
  }
  lay.cblknbr = SYMB_CBLKNBR;
  lay.bloknbr = SYMB_BLOKNBR;
  lay.cblktab = cb;
  lay.bloktab = bl;
  memset(&opts, 0, sizeof(opts));
  memset(&st, 0, sizeof(st));
  opts.schur = (sopar->schur == API_YES);      /* compute_1d skips the last diagonal block, sopalin_compute.c:767-772 */

  /* one entry point per (arithmetic x factorization) build of this file */
#if defined(TYPE_COMPLEX) && defined(PREC_DOUBLE)
#  if defined(CHOL_SOPALIN) && defined(SOPALIN_LU)
  rc = pastix_amd_z_ge_sopalin(&lay, ctab, utab, sopalin_data->critere, &opts, &st);
#  elif defined(CHOL_SOPALIN)
  rc = PASTIX_AMD_ERR_UNSUPPORTED;             /* complex `po`: see INTEGRATION.md */
#  elif defined(HERMITIAN)
  rc = pastix_amd_z_he_sopalin(&lay, ctab, sopalin_data->critere, &opts, &st);
#  else
  rc = pastix_amd_z_sy_sopalin(&lay, ctab, sopalin_data->critere, &opts, &st);
#  endif
#elif defined(TYPE_COMPLEX)
#  if defined(CHOL_SOPALIN) && defined(SOPALIN_LU)
  rc = pastix_amd_c_ge_sopalin(&lay, ctab, utab, sopalin_data->critere, &opts, &st);
#  elif defined(CHOL_SOPALIN)
  rc = PASTIX_AMD_ERR_UNSUPPORTED;
#  elif defined(HERMITIAN)
  rc = pastix_amd_c_he_sopalin(&lay, ctab, sopalin_data->critere, &opts, &st);
#  else
  rc = pastix_amd_c_sy_sopalin(&lay, ctab, sopalin_data->critere, &opts, &st);
#  endif
#elif defined(PREC_DOUBLE)
#  if defined(CHOL_SOPALIN) && defined(SOPALIN_LU)
  rc = pastix_amd_d_ge_sopalin(&lay, (double *const *)ctab, (double *const *)utab, sopalin_data->critere, &opts, &st);
#  elif defined(CHOL_SOPALIN)
  rc = pastix_amd_d_po_sopalin(&lay, (double *const *)ctab, sopalin_data->critere, &opts, &st);
#  else                                        /* real `he` is `sy` */
  rc = pastix_amd_d_sy_sopalin(&lay, (double *const *)ctab, sopalin_data->critere, &opts, &st);
#  endif
#else
#  if defined(CHOL_SOPALIN) && defined(SOPALIN_LU)
  rc = pastix_amd_s_ge_sopalin(&lay, (float *const *)ctab, (float *const *)utab, sopalin_data->critere, &opts, &st);
#  elif defined(CHOL_SOPALIN)
  rc = pastix_amd_s_po_sopalin(&lay, (float *const *)ctab, sopalin_data->critere, &opts, &st);
#  else
  rc = pastix_amd_s_sy_sopalin(&lay, (float *const *)ctab, sopalin_data->critere, &opts, &st);
#  endif
#endif

  if (rc == PASTIX_AMD_OK || rc == PASTIX_AMD_ERR_NUMERIC) {
    sopar->diagchange             = (PASTIX_INT)st.nbpivot;   /* -> IPARM_STATIC_PIVOTING (pastix.c:3853)   */
    sopar->dparm[DPARM_FACT_TIME] = st.fact_time;             /* sopalin3d.c:1125-1132                      */
    sopar->iparm[IPARM_INERTIA]   = (PASTIX_INT)st.inertia;   /* -1 unless real LDLt (sopalin3d.c:1144-1160) */
  }
  free(cb); free(bl); free(ctab); free(utab);
  return rc;
}

#endif /* SOPALIN_AMD_STUB_H */
