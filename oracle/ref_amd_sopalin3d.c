/*
 * TEST INFRASTRUCTURE (oracle/): the reference as the CALLER of the MI355X engine.
 *
 * This translation unit IS the reference's src/sopalin/src/sopalin3d.c (included from /root/reference, untouched,
 * through the include path) with one name redirected: its calls of sopalin_launch_thread() (sopalin3d.c:1411 and the
 * fused variants below it) go to AMD_HOOK, defined at the end of this file, which hands the numerical factorization --
 * and only it -- to integration/sopalin_amd_stub.h when PASTIX_AMD_ENGINE is set in the environment and runs the
 * reference's own launcher otherwise (solve-only and refinement threads always do).  It replaces the four
 * sopalin3d_{po,ge,sy,he}.o objects in oracle/_ref/ref_harness_*_amd (oracle/build_ref.sh, -DAMD_HOOK=<unique name per
 * variant>); everything else of that binary is the reference: pastix(), ordering, kass, blend, CoefMatrix_Init, updo.
 * Nothing of the reference is copied into the repository.
 */
/* The reference's own headers first: common_pastix.h brings redefine_functions.h (:453 `#define sopalin_launch_thread
 * PASTIX_PREFIX(sopalin_launch_thread)`), sopalin_thread.h declares the real launcher (:73-77).  Both are include-guarded,
 * so sopalin3d.c's own #includes of them below are no-ops and the redirection set up in between stays in force. */
#include "common_pastix.h"
#include "sopalin_thread.h"
#undef sopalin_launch_thread
#define sopalin_launch_thread AMD_HOOK
void AMD_HOOK(void *sopalin_data,
              PASTIX_INT procnum, PASTIX_INT procnbr, void *ptr, PASTIX_INT verbose,
              PASTIX_INT calc_thrdnbr, void * (*calc_routine)(void *), void *calc_data,
              PASTIX_INT comm_thrdnbr, void * (*comm_routine)(void *), void *comm_data,
              PASTIX_INT ooc_thrdnbr,  void * (*ooc_routine) (void *), void *ooc_data);
#include "sopalin3d.c"
#undef sopalin_launch_thread
#define sopalin_launch_thread PASTIX_PREFIX(sopalin_launch_thread)      /* as redefine_functions.h had it */

#include "../integration/sopalin_amd_stub.h"

extern int pastix_amd_hook_calls;       /* ref_harness.c: how many factorizations went to the GPU engine */
extern int pastix_amd_hook_last_rc;
extern double pastix_amd_hook_wall;     /* wall time of the last factorization at this launch, either engine */
#include <sys/time.h>
static double amd_hook_now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }

void AMD_HOOK(void *sopalin_data,
              PASTIX_INT procnum, PASTIX_INT procnbr, void *ptr, PASTIX_INT verbose,
              PASTIX_INT calc_thrdnbr, void * (*calc_routine)(void *), void *calc_data,
              PASTIX_INT comm_thrdnbr, void * (*comm_routine)(void *), void *comm_data,
              PASTIX_INT ooc_thrdnbr,  void * (*ooc_routine) (void *), void *ooc_data)
{
  const double t0 = amd_hook_now();
  if (calc_routine == API_CALL(sopalin_smp) && getenv("PASTIX_AMD_ENGINE") != NULL) {
    int rc = API_CALL(sopalin_amd)((Sopalin_Data_t *)calc_data);
    pastix_amd_hook_last_rc = rc;
    if (rc == PASTIX_AMD_OK) { pastix_amd_hook_calls++; pastix_amd_hook_wall = amd_hook_now() - t0; return; }
    fprintf(stderr, "pastix_amd engine returned %d: falling back to the CPU engine\n", rc);
  }
  sopalin_launch_thread(sopalin_data, procnum, procnbr, ptr, verbose, calc_thrdnbr, calc_routine, calc_data,
                        comm_thrdnbr, comm_routine, comm_data, ooc_thrdnbr, ooc_routine, ooc_data);
  if (calc_routine == API_CALL(sopalin_smp)) pastix_amd_hook_wall = amd_hook_now() - t0;
}
