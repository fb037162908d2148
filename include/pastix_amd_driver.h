/*
 * pastix_amd_driver.h -- pastix()-signature stand-in (host side above the C ABI).
 *
 * Mirrors  void pastix(pastix_data_t **pastix_data, MPI_Comm pastix_comm, PASTIX_INT n,
 *                      PASTIX_INT *colptr, PASTIX_INT *row, PASTIX_FLOAT *avals, PASTIX_INT *perm,
 *                      PASTIX_INT *invp, PASTIX_FLOAT *b, PASTIX_INT rhs, PASTIX_INT *iparm, double *dparm)
 * (src/sopalin/src/pastix.h:219-222) with PASTIX_INT = int64, PASTIX_FLOAT = double, MPI_Comm = int
 * (nompi.h).  iparm/dparm slot numbers and enum values are those of src/common/src/api.h:124-260.
 */
#ifndef PASTIX_AMD_DRIVER_H
#define PASTIX_AMD_DRIVER_H
#include "pastix_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { PASTIX_AMD_IPARM_SIZE = 128, PASTIX_AMD_DPARM_SIZE = 64 };   /* api.h:196,233 */
/* IPARM_ACCESS (api.h:124-197) -- the slots this driver reads or writes */
enum {
  IPARM_MODIFY_PARAMETER = 0, IPARM_START_TASK = 1, IPARM_END_TASK = 2, IPARM_VERBOSE = 3, IPARM_DOF_NBR = 4,
  IPARM_ITERMAX = 5, IPARM_MATRIX_VERIFICATION = 6, IPARM_NBITER = 10, IPARM_AMALGAMATION_LEVEL = 13,
  IPARM_ORDERING = 14, IPARM_STATIC_PIVOTING = 20, IPARM_NNZEROS = 22, IPARM_BASEVAL = 24,
  IPARM_MIN_BLOCKSIZE = 25, IPARM_MAX_BLOCKSIZE = 26, IPARM_SCHUR = 27, IPARM_FACTORIZATION = 30, IPARM_THREAD_NBR = 34,
  IPARM_LEVEL_OF_FILL = 36, IPARM_RHS_MAKING = 38, IPARM_REFINEMENT = 39, IPARM_SYM = 40, IPARM_GMRES_IM = 44,
  IPARM_FILL_MATRIX = 53, IPARM_INERTIA = 54,
  IPARM_ESP_NBTASKS = 55, IPARM_FLOAT = 61, IPARM_ERROR_NUMBER = 63, IPARM_CUDA_NBR = 64
};
/* DPARM_ACCESS (api.h:219-234) */
enum {
  DPARM_FILL_IN = 1, DPARM_EPSILON_REFINEMENT = 5, DPARM_RELATIVE_ERROR = 6, DPARM_SCALED_RESIDUAL = 7,
  DPARM_EPSILON_MAGN_CTRL = 10, DPARM_ANALYZE_TIME = 18, DPARM_FACT_TIME = 20, DPARM_SOLV_TIME = 21,
  DPARM_FACT_FLOPS = 22
};
/* API_TASK (api.h:253-260), API_BOOLEAN (:469-470), API_SYM (:400-402), API_ORDER (:504-509) */
enum { API_TASK_INIT = 0, API_TASK_ORDERING = 1, API_TASK_SYMBFACT = 2, API_TASK_ANALYSE = 3,
       API_TASK_NUMFACT = 4, API_TASK_SOLVE = 5, API_TASK_REFINE = 6, API_TASK_CLEAN = 7 };
enum { API_NO = 0, API_YES = 1 };
enum { API_SYM_YES = 0, API_SYM_NO = 1, API_SYM_HER = 2 };
/* refinement modes, IPARM_REFINEMENT (api.h:353-365) */
enum { API_RAF_GMRES = 0, API_RAF_GRAD = 1, API_RAF_PIVOT = 2, API_RAF_BICGSTAB = 3 };
enum { API_ORDER_SCOTCH = 0, API_ORDER_METIS = 1, API_ORDER_PERSONAL = 2, API_ORDER_LOAD = 3 };

typedef struct pastix_amd_data_s pastix_amd_data_t;

/* avals / b: PASTIX_FLOAT arrays of the arithmetic iparm[IPARM_FLOAT] names (api.h:522-525) -- `double` (API_REALDOUBLE, the
 * D_pastix build), interleaved `double complex` (API_COMPLEXDOUBLE, Z_pastix), `float` (API_REALSINGLE, S_pastix: the
 * factorization runs on the native fp32 engine; norm, solve and refinement keep double vectors inside and the result is
 * rounded to float) or interleaved `float complex` (API_COMPLEXSINGLE, C_pastix: no native complex-single engine -- the
 * values are widened and the fp64 engine computes). */
void pastix_amd_pastix(pastix_amd_data_t **pastix_data, int pastix_comm, pastix_amd_int_t n,
                       pastix_amd_int_t *colptr, pastix_amd_int_t *row, void *avals, pastix_amd_int_t *perm,
                       pastix_amd_int_t *invp, void *b, pastix_amd_int_t rhs, pastix_amd_int_t *iparm,
                       double *dparm);
/* Schur mode (iparm[IPARM_SCHUR] = API_YES): pastix_setSchurUnknownList (pastix.c:6200-6215; list in the CSC's base,
 * call it before the ordering task) isolates the unknowns at the end of the ordering as ONE cblk that is updated
 * but not factorized; pastix_getSchur (pastix.c:6434-6470) copies that cblk's panel: nschur x nschur, column-major,
 * in the order of the final permutation (lower triangle for LLt / LDLt, the whole square for LU).  The SOLVE and
 * REFINE tasks are not available in Schur mode. */
int pastix_amd_set_schur_unknown_list(pastix_amd_data_t **pastix_data, pastix_amd_int_t n, const pastix_amd_int_t *list);
int pastix_amd_get_schur(pastix_amd_data_t *pastix_data, void *schur);
/* extension: tell the ordering step that the matrix is an nx*ny*nz 7-point grid (geometric ND);
 * without it and without API_ORDER_PERSONAL the natural order is used (no Scotch/METIS here). */
int pastix_amd_set_grid(pastix_amd_data_t **pastix_data, pastix_amd_int_t nx, pastix_amd_int_t ny,
                        pastix_amd_int_t nz);
/* the device plan behind a pastix_data (NULL before API_TASK_ANALYSE) */
pastix_amd_plan_t *pastix_amd_data_plan(pastix_amd_data_t *pastix_data);

#ifdef __cplusplus
}
#endif
#endif
