/*
 * pastix_amd.h -- C ABI of the MI355X-native sopalin (numerical factorization) path.
 *
 * Drop-in boundary.  The reference's numerical factorization is entered through
 *     void {po,ge,sy,he}_sopalin_thread(SolverMatrix *m, SopalinParam *sopar)
 * (src/sopalin/src/sopalin3d.h:381-467, body sopalin3d.c:1388-1422, sole caller
 * pastix_task_sopalin, src/sopalin/src/pastix.c:3561-3575).  A PaStiX maintainer binds the
 * functions below at exactly that call site: the layout structs are the
 * read-only subset of SolverMatrix / SolverCblk / SolverBlok the path reads
 * (src/blend/src/solver.h:94-168), with 64-bit indices (200^3 needs coefnbr > 2^31).
 *
 * All functions return 0 on success or a negative PASTIX_AMD_ERR_* code; nothing aborts.
 * Plain pointers and sizes only: no C++/torch types cross this boundary.
 */
#ifndef PASTIX_AMD_H
#define PASTIX_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t pastix_amd_int_t;

/* IPARM_FACTORIZATION values (src/common/src/api.h:381-384) */
enum { PASTIX_AMD_FACT_LLT = 0, PASTIX_AMD_FACT_LDLT = 1, PASTIX_AMD_FACT_LU = 2, PASTIX_AMD_FACT_LDLH = 3 };
/* IPARM_FLOAT values (api.h:522-525): built: REALDOUBLE (LLt, LDLt, LU) and COMPLEXDOUBLE (symmetric LDLt,
 * Hermitian LDLh, LU) */
enum { PASTIX_AMD_REALSINGLE = 0, PASTIX_AMD_REALDOUBLE = 1, PASTIX_AMD_COMPLEXSINGLE = 2, PASTIX_AMD_COMPLEXDOUBLE = 3 };

enum {
  PASTIX_AMD_OK = 0,
  PASTIX_AMD_ERR_BADPARAMETER = -1,   /* api.h BADPARAMETER_ERR */
  PASTIX_AMD_ERR_ALLOC = -2,          /* out of host/device memory */
  PASTIX_AMD_ERR_DEVICE = -3,         /* HIP runtime error / no gfx950 device */
  PASTIX_AMD_ERR_NUMERIC = -4,        /* non-finite pivot met (reference would produce NaNs, compute_diag.c:143) */
  PASTIX_AMD_ERR_UNSUPPORTED = -5,    /* precision/factorization variant not built yet */
  PASTIX_AMD_ERR_LAYOUT = -6,         /* layout violates the SolverMatrix invariants (solver_check.c) */
  PASTIX_AMD_ERR_TIMEOUT = -7         /* multi-GPU driver: the streams did not drain within PASTIX_AMD_DIST_TIMEOUT seconds
                                         (a peer is missing or the two ends of a channel disagree); channels aborted */
};

/* SolverCblk subset (solver.h:94-107). cblktab has cblknbr+1 entries (last: bloknum = bloknbr). */
typedef struct pastix_amd_cblk_s {
  pastix_amd_int_t fcolnum;   /* first column (0-based, inclusive) */
  pastix_amd_int_t lcolnum;   /* last column (inclusive) */
  pastix_amd_int_t bloknum;   /* first blok = the diagonal blok */
  pastix_amd_int_t stride;    /* panel leading dimension = sum of blok heights */
} pastix_amd_cblk_t;

/* SolverBlok subset (solver.h:111-117) */
typedef struct pastix_amd_blok_s {
  pastix_amd_int_t frownum;   /* first row (inclusive) */
  pastix_amd_int_t lrownum;   /* last row (inclusive) */
  pastix_amd_int_t cblknum;   /* facing cblk */
  pastix_amd_int_t coefind;   /* row offset of the blok inside the panel */
} pastix_amd_blok_t;

typedef struct pastix_amd_layout_s {
  pastix_amd_int_t cblknbr;
  pastix_amd_int_t bloknbr;
  const pastix_amd_cblk_t *cblktab;   /* [cblknbr+1] */
  const pastix_amd_blok_t *bloktab;   /* [bloknbr]   */
} pastix_amd_layout_t;

/* Tunables of the device engine (all have defaults when the struct is zeroed). */
typedef struct pastix_amd_options_s {
  int device;            /* HIP device ordinal */
  int lookahead;         /* chunk size of the update schedule: contributions into one 128x128 target
                            tile are applied in groups whose accumulated inner dimension reaches this
                            value; 1 = every source separately (right-looking), huge = once per tile
                            (left-looking); <=0 = default (512, 1024 from 1e12 flop, 2048 above 5e13 flop) */
  int verbose;
  int external_arena;    /* 1: do not allocate the panel arena; the caller provides device memory with
                            pastix_amd_plan_set_arena (e.g. a torch tensor used with torch.distributed) */
  int schur;             /* 1: IPARM_SCHUR semantics of compute_1d (sopalin_compute.c:767-772): the last cblk is not
                            factorized; on return its panel holds the Schur complement (what pastix_getSchur reads).
                            That cblk is never re-cut, whatever its width.  Solves are not available on such a plan. */
  int quadrant_min;      /* leaf-side launches: a slot whose tasks of small pieces number at least this many has them cut
                            into 64x64 quadrant tasks for k_update_small (plan.cpp); <= 0 = default (1024) */
  int quadrant_fill_pct; /* ... a task qualifies when its pieces fill less than this many percent of the 128x128x16
                            chunks k_update would run for them; <= 0 = default (25) */
  int run_schedule;      /* the thin levels at the top of the elimination tree (at most run_max_cblks cblks each) as ONE
                            dependency-driven launch -- tasks gated by tile counters, the way the reference's tasks wait
                            for TASK_CTRBCNT (sopalin3d.c:790-1025) -- instead of launches per level: 0 = default (on for
                            real double LLt / LDLt / LU and complex double LDLt / LDLh on one GPU up to 2e14 flop),
                            1 = on wherever it is built, whatever the size, -1 = off (the level-by-level schedule) */
  int run_max_cblks;     /* <= 0 = default (32) */
  int run_t_workers;     /* unused (kept for the layout of the struct): panel solves are tickets of the run launch */
  int run_d_workers;     /* resident workgroups of the run's diagonal-blok kernel (they pop ready diagonal tasks);
                            <= 0 = default (8) */
  int gather_min;        /* a target tile that ONE source cblk reaches with at least this many rectangles (fragmented
                            layouts: blend on separators numbered across their low-side neighbours) gets them as one
                            GATHERED piece -- consecutive source rows, scattered landing, gathered while they are staged
                            (plan.cpp); 0 = default (from 2 rectangles, on layouts whose off-diagonal bloks per (cblk, facing
                            cblk) pair average 1.5 or more), > 0 = from this many on any layout, -1 = never */
  int reserved[4];
} pastix_amd_options_t;

/* Statistics of a plan / a factorization. */
typedef struct pastix_amd_stats_s {
  double fact_flops;       /* DPARM_FACT_FLOPS definition (blend_symbol_cost.c:52-88) on this layout */
  double fact_time;        /* seconds, first kernel launch -> last kernel done (DPARM_FACT_TIME semantics) */
  double update_time;      /* seconds during which at least one launch of the update (GEMM+scatter) kernel was in
                              flight, from HIP events (the launches of the two streams may overlap) */
  double h2d_time, d2h_time;
  pastix_amd_int_t nbpivot;  /* static pivots (IPARM_STATIC_PIVOTING) */
  pastix_amd_int_t coefnbr;  /* panel elements (one of L/U) */
  pastix_amd_int_t nlevels, ntasks, npieces, nupdate_launches;
  pastix_amd_int_t inertia;  /* IPARM_INERTIA: positive D entries (real LDLt), -1 otherwise */
  double update_flops;     /* 2*m*n*k summed over the update pieces of this plan */
  double local_flops;      /* fact_flops restricted to the cblks this plan owns (== fact_flops on one GPU) */
  double update_bytes;     /* algorithmic bytes of the update kernel: 8k(m+n) per piece + 16*tm*tn per task */
  double full_flops;       /* part of update_flops carried by full 128x128 pieces */
  double update_time_sum;  /* sum of the durations of the bulk update launches (kernel k_update<.,0>), nupdate_launches
                              of them, carrying update_flops - urgent_flops */
  double urgent_flops;     /* part of update_flops done by the urgent launches of the two-stream driver (kernel
                              k_update<.,1>; 0 with one stream) */
  double urgent_time_sum;  /* sum of their durations */
  pastix_amd_int_t nurgent_launches;
  double solve_time;         /* s, device time of the last pastix_amd_solve call's sweeps (no host transfers) */
  double nquadrant_tasks;    /* tasks of the plan that are 64x64 quadrant tasks (run by k_update_small) */
  double run_time;           /* s, duration of the run launch (k_run_update: the update and panel-solve tasks of the thin levels
                                in one dependency-driven launch, opts.run_schedule); 0 when the level-by-level schedule ran.
                                It is one of the nupdate_launches and part of update_time_sum */
  double run_flops;          /* update flops carried by that launch (part of update_flops) */
  pastix_amd_int_t run_tickets, run_first_level;   /* its tasks; the first level it covers (-1: none) */
  /* the one-shot entry points ({po,sy,he,ge}_sopalin): what the caller of the drop-in pays besides fact_time */
  double plan_time;          /* s, analysis of the layout + device tables + arenas (0 when the cached plan of the previous call
                                with the same layout, factorization and options was reused) */
  double total_time;         /* s, wall time of the whole call: plan_time + h2d_time + factorization + d2h_time */
} pastix_amd_stats_t;

typedef struct pastix_amd_plan_s pastix_amd_plan_t;

/* ---- one-shot drop-ins for {po,ge,sy}_sopalin_thread (double real) -------------------------
 * coeftab[k] / ucoeftab[k]: host panel of cblk k, column-major stride(k) x width(k), already filled
 * (CoefMatrix_Init, coefinit.c:283-296); factorized in place, as the reference leaves them for
 * updo.c.  critere: pivot threshold (sopalin3d.c:586-606).  nbpivot -> sopar->diagchange.
 * The buffers are overwritten WHILE the call runs (from 2e12 flop on, the panels of the lower levels are copied back beside
 * the factorization of the upper ones); during the call the device holds one more copy of the panels than the plan keeps
 * between calls.  h2d_time / d2h_time of the stats: the copies in and out, d2h_time including the part
 * that ran beside the factorization. */
int pastix_amd_d_po_sopalin(const pastix_amd_layout_t *layout, double *const *coeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_d_sy_sopalin(const pastix_amd_layout_t *layout, double *const *coeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_d_ge_sopalin(const pastix_amd_layout_t *layout, double *const *coeftab, double *const *ucoeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);

/* complex double: coeftab[k] / ucoeftab[k] are the reference's interleaved `double complex` panels.
 *   z_sy = Z_sy_sopalin_thread, complex-SYMMETRIC LDLt (no conjugation: sopalin_compute.h:549-562)
 *   z_he = Z_he_sopalin_thread, Hermitian LDL^H (-DHERMITIAN: compute_diag.c:326-410, compute_trsm.c:91-95)
 *   z_ge = Z_ge_sopalin_thread, LU with static pivoting (compute_diag.c:432-532)
 * The complex `po` variant of the reference mixes symmetric and Hermitian BLAS calls (csqrt/geru/TRSM "T" in
 * compute_diag.c:124-203 with zherk and GEMM "N","C" in sopalin_compute.c:326-332); it is not offered:
 * plan_create(LLT, COMPLEXDOUBLE) returns PASTIX_AMD_ERR_UNSUPPORTED. */
int pastix_amd_z_sy_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_z_he_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_z_ge_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, void *const *ucoeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);

/* single precision (S_ / C_ {po,sy,he,ge}_sopalin_thread, sopalin3d.h:381-467 under -DPREC_SIMPLE): coeftab[k] are
 * `float` / interleaved `float complex` panels.
 *   s_*: the native fp32 engine -- float arenas on the device, fp32 MFMA kernels (kernels_f32.hip); nothing is widened.
 *        The staged API takes it too: pastix_amd_plan_create(..., PASTIX_AMD_REALSINGLE, ...) with float panels / CSC
 *        values (the vectors of pastix_amd_solve stay double, so that refinement recovers double accuracy).
 *   c_*: complex single is NOT native: the panels are widened to double complex on the host, factorized by the fp64 engine
 *        and rounded back (the arithmetic is f64); plan_create(PASTIX_AMD_COMPLEXSINGLE) returns PASTIX_AMD_ERR_UNSUPPORTED. */
int pastix_amd_s_po_sopalin(const pastix_amd_layout_t *layout, float *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_s_sy_sopalin(const pastix_amd_layout_t *layout, float *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_s_ge_sopalin(const pastix_amd_layout_t *layout, float *const *coeftab, float *const *ucoeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_c_sy_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_c_he_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, double critere,
                            const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);
int pastix_amd_c_ge_sopalin(const pastix_amd_layout_t *layout, void *const *coeftab, void *const *ucoeftab,
                            double critere, const pastix_amd_options_t *opts, pastix_amd_stats_t *stats);

/* The one-shot entry points keep the plan of their last call (host analysis, device tables AND the device arenas): a call
 * with the same layout, factorization, arithmetic and options -- what pastix() does when it re-factorizes on one analysis,
 * pastix.c:3439-3575 -- skips the analysis and the allocations.  This releases it (device memory included); it is also
 * released when a call with a different layout replaces it and at process exit. */
void pastix_amd_release_cached_plan(void);

/* ---- staged API (analysis once, many factorizations; panels may stay on the device) -------- */
int pastix_amd_plan_create(const pastix_amd_layout_t *layout, int factotype, int floattype,
                           const pastix_amd_options_t *opts, pastix_amd_plan_t **plan);
void pastix_amd_plan_destroy(pastix_amd_plan_t *plan);
int pastix_amd_plan_stats(const pastix_amd_plan_t *plan, pastix_amd_stats_t *stats);

/* panels <-> device arena.  "packed": all panels concatenated in cblk order (offset of cblk k =
 * sum_{c<k} stride(c)*width(c)); "tabs": the reference's one-buffer-per-cblk arrays. */
int pastix_amd_upload_packed(pastix_amd_plan_t *plan, const void *L, const void *U);
int pastix_amd_download_packed(pastix_amd_plan_t *plan, void *L, void *U);
int pastix_amd_upload_tabs(pastix_amd_plan_t *plan, void *const *coeftab, void *const *ucoeftab);
int pastix_amd_download_tabs(pastix_amd_plan_t *plan, void *const *coeftab, void *const *ucoeftab);

/* coefficient fill on the device: CoefMatrix_Init + Csc2solv_cblk (coefinit.c:283-296,
 * csc_intern_solve.c:65-132).  CSC 1-based; sym!=0: lower triangle only; perm 0-based old->new. */
int pastix_amd_fill_csc(pastix_amd_plan_t *plan, int sym, pastix_amd_int_t n, const pastix_amd_int_t *colptr,
                        const pastix_amd_int_t *rows, const void *vals, const pastix_amd_int_t *perm);

/* IPARM_FILL_MATRIX = API_YES, the reference's structure-only "fake factorisation" fill (CoefMatrix_Init,
 * coefinit.c:343-443): no CSC; coeftab all 1, ucoeftab all 2, diagonals gnodenbr^2, LU: strictly upper part of
 * coeftab's diagonal bloks 2.  The matching pivot threshold is (gnodenbr^2 + gnodenbr) sqrt(eps)
 * (sopalin3d.c:597-598).  Cached like pastix_amd_fill_csc.  One-GPU plans; any cblk width (cblks wider than 128 columns are
 * re-cut into column groups inside the engine like everywhere else). */
int pastix_amd_fill_fake(pastix_amd_plan_t *plan, pastix_amd_int_t gnodenbr);
/* re-apply the fill cached by the last pastix_amd_fill_csc (device only: memset + scatter). */
int pastix_amd_refill(pastix_amd_plan_t *plan);

/* numerical factorization of the device-resident panels (the hot path). */
int pastix_amd_factorize(pastix_amd_plan_t *plan, double critere, pastix_amd_stats_t *stats);

/* ---- multi-GPU: one plan per rank over the same layout ------------------------------
 * owner[k] = rank that factorizes cblk k.  The plan holds the owned panels plus zero-initialised
 * "shadow" panels for remote cblks that receive contributions from owned ones: the fan-in buffers of
 * add_contrib_target (sopalin_compute.c:600-733).  Contributions are SUBTRACTED into the shadow; the
 * caller ships each shadow to its owner (RCCL point-to-point) before the owner reaches that cblk's
 * level, and the owner ADDS it into its panel (recv_handle_fanin, sopalin_sendrecv.c:384-389).
 * The level-stepped calls let the caller interleave that exchange:
 *   begin; for l in levels: factorize_level(l,1); <exchange+add shadows of level l>; factorize_level(l,2); end. */
int pastix_amd_plan_create_dist(const pastix_amd_layout_t *layout, int factotype, int floattype,
                                const pastix_amd_options_t *opts, const int32_t *owner, int32_t myrank,
                                pastix_amd_plan_t **plan);
/* Fan-in regions (the reference's FanInTarget, ftgt.h:67-113, at blok granularity; host only): bit r of
 * mask[b] (b < bloknbr, r < 64) is set when rank r contributes into blok b of a cblk it does not own.  A rank's
 * shadow panel of a remote cblk holds exactly its marked bloks, in blok order, column-major with leading dimension
 * = the sum of their heights; sender and receiver both derive it from (layout, owner). */
int pastix_amd_fanin_touched(const pastix_amd_layout_t *layout, const int32_t *owner, uint64_t *mask);
/* owner side of the fan-in (recv_handle_fanin, sopalin_sendrecv.c:384-404): panel(cblk)[rows[r], c] += src[r + c*nrows]
 * for r < nrows, c < width(cblk); src and rows are device pointers, the add runs on the plan's stream. */
int pastix_amd_plan_fanin_add(pastix_amd_plan_t *plan, pastix_amd_int_t cblk, const void *src, const int32_t *rows,
                              pastix_amd_int_t nrows);
/* poff[cblknbr+1]: arena offset of every panel (absent cblks have size 0, shadows are compact); level[cblknbr];
 * role[cblknbr]: 1 owned, 2 shadow, 0 absent.  Any pointer may be NULL. */
int pastix_amd_plan_layout_info(const pastix_amd_plan_t *plan, pastix_amd_int_t *poff, int32_t *level,
                                int8_t *role);
/* with opts.external_arena: the caller owns the device memory of the panels (e.g. a torch tensor it also hands to
 * torch.distributed; real arithmetic only).  arena_info: *nelems = doubles each buffer must hold,
 * *first = element index at which the engine places the first panel (panel of cblk k starts at first + poff[k] of
 * pastix_amd_plan_layout_info; the elements before and after are slack for the update kernel's 16-byte DMA lanes, which
 * touch the neighbouring element when a contribution starts or ends on an odd row).  set_arena takes the allocations
 * themselves (dU NULL unless LU) and their size in elements; a buffer smaller than arena_info asks for is refused. */
int pastix_amd_plan_arena_info(const pastix_amd_plan_t *plan, pastix_amd_int_t *nelems, pastix_amd_int_t *first);
int pastix_amd_plan_set_arena(pastix_amd_plan_t *plan, void *dL, void *dU, pastix_amd_int_t nelems);
/* host-only schedule statistics of one rank (no device needed): per launch slot the update flops, the
 * largest task (multiply-adds), the task count, per level the panel (diag+trsm) flops, and the part of the slot's
 * flops whose targets are of the slot's own level (the urgent tasks the level's panel kernels wait for) */
int pastix_amd_plan_profile(const pastix_amd_layout_t *layout, int factotype, const pastix_amd_options_t *opts,
                            const int32_t *owner, int32_t myrank, pastix_amd_int_t maxlevels, double *slot_flops,
                            double *slot_maxwork, pastix_amd_int_t *slot_tasks, double *level_panel_flops,
                            pastix_amd_int_t *nlevels, double *slot_urgent_flops /* may be NULL */);
/* host-only: the run schedule of a layout (opts.run_schedule; the thin levels at the top of the tree in one dependency-driven
 * launch) and its replay check.  info[0..7] = first level of the run (-1: none), levels, update tasks in the run, source-tile
 * waits, resident workgroups for diagonal bloks, panel-solve tasks, update flops inside the run, result of the replay (0 =
 * every task can run when the tickets are served one at a time in order: the schedule cannot deadlock) */
int pastix_amd_plan_run_info(const pastix_amd_layout_t *layout, int factotype, int floattype, const pastix_amd_options_t *opts,
                             pastix_amd_int_t *info);
/* host-only (tests): the update pieces of the plan -- rectangles and gathered pieces (options.gather_min) -- against the
 * reference's definition: every product (rows of blok j) x (rows of blok i)^T, j >= i, of every source cblk subtracted once
 * where add_contrib_local puts it (sopalin_compute.c:427-429).  Real LLt / LDLt.  out[0..3] = products expected, products the
 * pieces make, mismatches, gathered pieces. */
int pastix_amd_plan_check_pieces(const pastix_amd_layout_t *layout, int factotype, const pastix_amd_options_t *opts,
                                 pastix_amd_int_t *out);
/* tests: a digest of the run schedule's dependency tables as they stand ON THE DEVICE (the counterpart of indtab /
 * TASK_CTRBCNT, solverMatrixGen.c:667-760, solver.h:71-75).  Round 6 builds the reader lists of the run on the GPU
 * (csrc/run_edges.hip); with PASTIX_AMD_DEV=run_host_edges the host builds them as before: both must give the same digest.
 * out[0] = (panel-solve ticket, reading update ticket) pairs, out[1] = order-independent hash of the pairs, out[2] = hash of
 * the tickets' initial counters, out[3] = tickets ready when the run starts; -1s when the plan has no run. */
int pastix_amd_plan_run_edges_digest(pastix_amd_plan_t *plan, pastix_amd_int_t *out);
/* host-only: the multi-GPU driver's partition -- owner[k] = the rank (GPU) that factorizes cblk k, for `world` <= 64 ranks.
 * Proportional mapping on the cblk elimination tree, blend's idea (splitpart.c:752-1012; GPU colouring
 * blend_distributeOnGPU.c:59-317): a subtree gets a set of candidate ranks; the cblks of the separator at its top are dealt
 * over the set, heaviest first onto the least loaded rank; at a branching the set is divided among the heavy children in
 * proportion to their flops; a subtree with one candidate goes to it whole; side subtrees lighter than `light` (<= 0: 0.05)
 * of their parent go whole to the least loaded candidate.  A rank then only contributes to separators on its own path to
 * the root.  Deterministic: every rank computes the same map from the layout alone; its result is what
 * pastix_amd_plan_create_dist / pastix_amd_fanin_touched / pastix_amd_dist_schedule take as `owner`. */
int pastix_amd_dist_partition(const pastix_amd_layout_t *layout, int world, double light, int32_t *owner);
/* ---- multi-GPU driver: asynchronous fan-in over RCCL point-to-point (csrc/dist.cpp) ------------------------------
 * One process per GPU.  Every rank: plan_create_dist (own arena), fill_csc, then ONCE pastix_amd_dist_attach_rccl,
 * then pastix_amd_factorize_dist as often as needed (pastix_amd_refill in between).  The reference's counterpart is the
 * fan-in protocol of sopalin_compute.c:600-733 (accumulate, send when the last local contribution has landed) and
 * sopalin_sendrecv.c:182-485,1219-1556,2393-2775 (receives posted ahead, the owner adds).  Here a rank's whole
 * factorization is enqueued on HIP streams -- one channel (2-rank RCCL communicator + stream) per peer -- and the host
 * never waits for a peer; no collective on the data path. */
#define PASTIX_AMD_DIST_ID_BYTES 128
typedef struct pastix_amd_dist_info_s {
  int32_t world, rank, npeers, nplanes;   /* nplanes: arenas per fan-in block (1 LLt/LDLt, 2 LU or complex, 4 complex LU) */
  pastix_amd_int_t nsend, nrecv;          /* fan-in blocks per factorization */
  double bytes_sent, bytes_recv;          /* per factorization */
  double staging_bytes;                   /* receive staging area */
  double fanin_buffer_bytes;              /* this rank's fan-in buffers (compact shadow panels), one plane */
  char transport[16];                     /* "rccl" | "loopback" */
} pastix_amd_dist_info_t;
/* an RCCL unique id (ncclGetUniqueId); the job needs one per communicating pair, made on any rank and shipped to both
 * members out of band (bench.py: torch.distributed) */
int pastix_amd_dist_unique_id(void *id128);
/* self-test of the RCCL binding on one device: a 1-rank communicator sends `count` doubles to itself in a group */
int pastix_amd_dist_selftest_rccl(int device, pastix_amd_int_t count);
/* ids: world*world entries of PASTIX_AMD_DIST_ID_BYTES, entry [a*world + b] for a < b (others unused).  Collective. */
int pastix_amd_dist_attach_rccl(pastix_amd_plan_t *plan, int32_t world, const void *ids);
/* the rank plans of ONE process wired to each other (device-to-device copies): single-GPU emulation of a job, used by
 * the tests to run the same driver where RCCL cannot (one GPU) */
int pastix_amd_dist_attach_local(pastix_amd_plan_t *const *plans, int32_t world);
int pastix_amd_dist_info(const pastix_amd_plan_t *plan, pastix_amd_dist_info_t *info);
/* host only: out[2q] = hash of the fan-in blocks this rank SENDS to rank q, out[2q+1] of
 * those it RECEIVES from q (level, cblk, rows, width, planes; channel order), q < world.  The two ends of a channel
 * agree iff a's out[2b] == b's out[2a+1] for every pair: the launcher compares them over its bootstrap BEFORE
 * pastix_amd_dist_attach_rccl (dist.py: check_schedule_hashes), attach_local compares them itself.  The reference has no
 * such check: a mismatched MPI fan-in hangs in recv_waitone_fob (sopalin_sendrecv.c:1219-1556). */
int pastix_amd_dist_schedule_hash(const pastix_amd_layout_t *layout, int factotype, int floattype, const int32_t *owner,
                                  int32_t myrank, int32_t world, uint64_t *out /* [2*world] */);
/* Both enqueue everything and wait at the end with a deadline (PASTIX_AMD_DIST_TIMEOUT seconds, default 300): on expiry,
 * or on any error with work in flight, the rank prints the first unmatched (level, peer, cblk, direction) of every
 * channel, aborts its communicators (ncclCommAbort), drains its streams and returns PASTIX_AMD_ERR_TIMEOUT / the error;
 * the plan's distributed state is dead afterwards (later calls return PASTIX_AMD_ERR_BADPARAMETER). */
int pastix_amd_factorize_dist(pastix_amd_plan_t *plan, double critere, pastix_amd_stats_t *stats);
/* drives the plans of pastix_amd_dist_attach_local with one host thread per rank; stats / rcs: [world] or NULL */
int pastix_amd_factorize_dist_local(pastix_amd_plan_t *const *plans, int32_t world, double critere,
                                    pastix_amd_stats_t *stats, int32_t *rcs);
/* Triangular solves on the distributed factors (one right-hand side; real plans: x is n doubles, complex plans: the
 * reference's interleaved n `double complex`; the data flow of updo.c with
 * several processes, updo_sendrecv.c): x (host, permuted numbering): in the right-hand side (full length on every
 * rank), out the solution on the columns of the cblks this rank owns and zeros elsewhere -- the sum over the ranks is
 * the solution.  Forward sweep: fan-in of the vector contributions at the target's level; backward sweep: the same
 * channels the other way (the owner sends the solved segment to the ranks whose panels have rows in it). */
int pastix_amd_solve_dist(pastix_amd_plan_t *plan, double *x);
int pastix_amd_solve_dist_local(pastix_amd_plan_t *const *plans, int32_t world, double *const *xs);
/* host only: the fan-in blocks of one rank in the order both ends of every channel issue them, 6 integers each:
 * {level, peer, cblk, dir (0 send, 1 receive), nrows, width}.  out may be NULL (count only). */
int pastix_amd_dist_schedule(const pastix_amd_layout_t *layout, int factotype, int floattype, const int32_t *owner,
                             int32_t myrank, int32_t world, pastix_amd_int_t cap, pastix_amd_int_t *out,
                             pastix_amd_int_t *nmsg, int32_t *nplanes);

int pastix_amd_plan_set_stream(pastix_amd_plan_t *plan, void *hip_stream);     /* run on the caller's stream */
int pastix_amd_factorize_begin(pastix_amd_plan_t *plan, double critere);
/* phase 0: contributions of slot `level` then the owned cblks of `level`; 1: contributions only;
 * 2: cblks only (the fan-in exchange for cblks of `level` goes between phase 1 and phase 2) */
int pastix_amd_factorize_level(pastix_amd_plan_t *plan, int level, int phase);
int pastix_amd_factorize_end(pastix_amd_plan_t *plan, pastix_amd_stats_t *stats);

/* one panel, device -> host (same layout as cblktab[cblk].coeftab / ucoeftab; U may be NULL): e.g. the Schur
 * complement of a Schur-mode plan, which is what pastix_getSchur copies (pastix.c:6434-6470) */
int pastix_amd_download_cblk(pastix_amd_plan_t *plan, pastix_amd_int_t cblk, void *L, void *U);
/* triangular solves on the device-resident factors (the data flow of up_down_smp, updo.c:114), x (permuted
 * numbering, n x nrhs, ld n; `double`, or interleaved `double complex` for complex plans) in place.  Not available
 * on distributed or Schur-mode plans (PASTIX_AMD_ERR_UNSUPPORTED). */
int pastix_amd_solve(pastix_amd_plan_t *plan, void *x, pastix_amd_int_t nrhs);
/* the same on a vector that already lives on the plan's device (device pointer, same layout): no host transfers */
int pastix_amd_solve_device(pastix_amd_plan_t *plan, void *dx, pastix_amd_int_t nrhs);

/* Iterative refinement on the device (csrc/refine.hip; pastix_task_raff, pastix.c:4300-4500): mode = IPARM_REFINEMENT
 * (api.h:353-365: 0 GMRES raff_gmres.c, 1 conjugate gradient raff_grad.c, 2 simple iterative refinement raff_pivot.c,
 * 3 BiCGStab raff_bicgstab.c), preconditioned by the device solve on the factors; Krylov vectors, the sparse
 * matrix-vector product and the dot products stay on the GPU.  CSC 1-based in the caller's numbering (sym: 0 full
 * pattern stored, 1 lower triangle of a symmetric matrix, 2 lower triangle of a Hermitian matrix), perm 0-based
 * old -> new (the factor's numbering), b: right-hand sides, x: in the solution to improve / out the refined one (host,
 * n x nrhs, `double` or interleaved `double complex` like the plan).  Stops at ||b - A x|| / ||b|| < eps or after
 * itermax iterations; *iters -> IPARM_NBITER, *relerr -> DPARM_RELATIVE_ERROR. */
int pastix_amd_refine(pastix_amd_plan_t *plan, int mode, int sym, pastix_amd_int_t n, const pastix_amd_int_t *colptr,
                      const pastix_amd_int_t *rows, const void *vals, const pastix_amd_int_t *perm, const void *b, void *x,
                      pastix_amd_int_t nrhs, double eps, pastix_amd_int_t itermax, int gmres_im, pastix_amd_int_t *iters,
                      double *relerr);

/* raw device pointers of the arenas (for callers that own device-side pipelines, e.g. RCCL fan-in) */
int pastix_amd_device_arenas(pastix_amd_plan_t *plan, void **dL, void **dU);

/* DPARM_FACT_FLOPS on a layout (host only) */
double pastix_amd_fact_flops(const pastix_amd_layout_t *layout, int factotype, int floattype);

const char *pastix_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PASTIX_AMD_H */
