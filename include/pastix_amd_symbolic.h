/*
 * pastix_amd_symbolic.h -- host-side layout producer (the callers' side of the drop-in boundary).
 *
 * In PaStiX the cblk/blok layout comes from order/ + kass/ + blend/ (kass(), src/kass/src/kass.c:93;
 * solverBlend(), src/blend/src/blend.c:115; splitting src/blend/src/splitpart.c; coefind/stride
 * src/blend/src/solverMatrixGen.c:1053-1069).  Those stay as-is for a real drop-in; this producer
 * exists so that benchmarks and the pastix()-style driver can run where PaStiX is not installed.
 * It emits the same data model (pastix_amd_layout_t) with GPU-friendly block sizes.
 */
#ifndef PASTIX_AMD_SYMBOLIC_H
#define PASTIX_AMD_SYMBOLIC_H
#include "pastix_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pastix_amd_symbolic_options_s {
  int max_blocksize;      /* IPARM_MAX_BLOCKSIZE role (pastix.c:372-373); default 128, capped at 256 */
  int amalgamation_pct;   /* IPARM_AMALGAMATION_LEVEL role: allowed extra fill in percent; default 5 */
  int max_merge_width;    /* do not create amalgamated nodes wider than this (0 = no limit) */
  int schur_n;            /* IPARM_SCHUR: the last schur_n unknowns of the ordering (which must be mutually coupled in
                             the pattern: a clique) stay ONE cblk, not split and not merged with anything else */
  int blend_split;        /* 0: cut wide supernodes into cblks of max_blocksize columns (last one narrower): full
                             128-column tiles for the device kernels.  1: blend's own rule (splitOnProcs,
                             src/blend/src/splitpart.c:387-516): equal pieces of width / (width / max) columns, and
                             supernodes that would give fewer than 4 pieces stay whole */
  int min_blocksize;      /* IPARM_MIN_BLOCKSIZE role (pastix.c:372): lower bound of the piece width with several
                             candidate processors; default max_blocksize / 2 */
  int candidate_procs;    /* candidate processors of a supernode in blend's rule (1 = sequential / one GPU) */
  int reserved[9];
} pastix_amd_symbolic_options_t;

typedef struct pastix_amd_symbol_s pastix_amd_symbol_t;

/* Geometric nested dissection of an nx*ny*nz 7-point grid (node id = x + nx*(y + ny*z)), split the
 * longest axis at its midpoint, separators last and numbered hierarchically.  perm: old->new,
 * invp: new->old, both 0-based, length nx*ny*nz. */
int pastix_amd_order_grid(pastix_amd_int_t nx, pastix_amd_int_t ny, pastix_amd_int_t nz, int leaf,
                          pastix_amd_int_t *perm, pastix_amd_int_t *invp);

/* Nested dissection of a general symmetric-pattern graph (the fallback where neither Scotch / METIS nor a grid hint is
 * available; the reference calls Scotch here, pastix.c:1540-1680): recursive bisection by level structures -- breadth
 * first search from a pseudo-peripheral vertex, the middle level is the separator (George's automatic nested
 * dissection) -- separators numbered last, parts of at most `leaf` vertices numbered in reverse Cuthill-McKee order.
 * CSC 1-based, lower triangle or full pattern.  perm: old->new, invp: new->old, 0-based. */
int pastix_amd_order_graph(pastix_amd_int_t n, const pastix_amd_int_t *colptr, const pastix_amd_int_t *rows, int leaf,
                           pastix_amd_int_t *perm, pastix_amd_int_t *invp);

/* Symbolic factorization of the pattern (CSC 1-based; lower triangle or full, symmetric pattern)
 * under the ordering perm (0-based old->new; NULL = natural).  The ordering is refined
 * (postorder + amalgamation moves); read the final one back with pastix_amd_symbol_perm. */
int pastix_amd_symbolic(pastix_amd_int_t n, const pastix_amd_int_t *colptr, const pastix_amd_int_t *rows,
                        const pastix_amd_int_t *perm, const pastix_amd_symbolic_options_t *opts,
                        pastix_amd_symbol_t **out);
int pastix_amd_symbol_layout(const pastix_amd_symbol_t *s, pastix_amd_layout_t *out);
int pastix_amd_symbol_perm(const pastix_amd_symbol_t *s, const pastix_amd_int_t **perm,
                           const pastix_amd_int_t **invp);
/* info[0..5] = n, cblknbr, bloknbr, nnz(L), fundamental supernodes, amalgamated supernodes */
int pastix_amd_symbol_info(const pastix_amd_symbol_t *s, pastix_amd_int_t *info);
void pastix_amd_symbol_destroy(pastix_amd_symbol_t *s);

#ifdef __cplusplus
}
#endif
#endif
