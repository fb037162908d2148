// driver.cpp -- pastix()-signature stand-in above the C ABI (host side).
//
// The reference's step sequencer is pastix() (src/sopalin/src/pastix.c:4734-5098) driven by
// iparm[IPARM_START_TASK..IPARM_END_TASK] (src/common/src/api.h:253-260) with defaults from
// pastix_initParam (pastix.c:334-456).  In a real drop-in that driver stays as-is and only
// {po,ge,sy}_sopalin_thread is rebound.  This stand-in keeps the same calling
// convention (1-based CSC, lower triangle for symmetric input, perm/invp in the CSC's base, iparm /
// dparm slots and enum values of api.h:124-234) so that the reference's example call sequence
// (src/example/src/simple.c:59-256) runs unchanged on a box without PaStiX: ordering and symbolic
// steps use this repo's producer (symbolic.cpp), the numerical factorization and the solves run on
// the device.  Refinement: GMRES(m), conjugate gradient, BiCGStab and plain iterative
// refinement, preconditioned by the device solve, vectors and SpMV on the device (refine.hip).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/pastix_amd.h"
#include "../../include/pastix_amd_symbolic.h"
#include "../../include/pastix_amd_driver.h"

struct pastix_amd_data_s {
  int64_t n = 0;
  pastix_amd_symbol_t* sym = nullptr;
  pastix_amd_plan_t* plan = nullptr;
  std::vector<int64_t> perm, invp;   // 0-based, final
  int64_t grid[3] = {0, 0, 0};
  double norm1 = 0;
  std::vector<double> rhs;           // right-hand side saved by the SOLVE step (sopar->b role)
  std::vector<int64_t> schur_list;   // 0-based unknowns isolated at the end (pastix_setSchurUnknownList)
  bool schur_on = false;             // the current analysis was made in Schur mode
  bool factorized = false;
};

static void init_param(pastix_amd_int_t* iparm, double* dparm) {   // pastix_initParam, pastix.c:334-456
  for (int i = 0; i < PASTIX_AMD_IPARM_SIZE; i++) iparm[i] = 0;
  for (int i = 0; i < PASTIX_AMD_DPARM_SIZE; i++) dparm[i] = 0;
  iparm[IPARM_MODIFY_PARAMETER] = API_YES;
  iparm[IPARM_START_TASK] = API_TASK_ORDERING;
  iparm[IPARM_END_TASK] = API_TASK_CLEAN;
  iparm[IPARM_VERBOSE] = 1;
  iparm[IPARM_DOF_NBR] = 1;
  iparm[IPARM_ITERMAX] = 250;
  iparm[IPARM_MATRIX_VERIFICATION] = API_YES;
  iparm[IPARM_AMALGAMATION_LEVEL] = 5;
  iparm[IPARM_ORDERING] = API_ORDER_SCOTCH;
  iparm[IPARM_BASEVAL] = 1;
  iparm[IPARM_MIN_BLOCKSIZE] = 60;
  iparm[IPARM_MAX_BLOCKSIZE] = 120;
  iparm[IPARM_SCHUR] = API_NO;
  iparm[IPARM_FACTORIZATION] = PASTIX_AMD_FACT_LDLT;
  iparm[IPARM_THREAD_NBR] = 1;
  iparm[IPARM_CUDA_NBR] = 0;
  iparm[IPARM_LEVEL_OF_FILL] = 1;
  iparm[IPARM_RHS_MAKING] = 0;
  iparm[IPARM_REFINEMENT] = API_RAF_GMRES;
  iparm[IPARM_GMRES_IM] = 25;
  iparm[IPARM_SYM] = API_SYM_YES;
  iparm[IPARM_INERTIA] = -1;
  iparm[IPARM_ESP_NBTASKS] = -1;
  iparm[IPARM_FLOAT] = PASTIX_AMD_REALDOUBLE;
  dparm[DPARM_EPSILON_REFINEMENT] = 1e-12;
  dparm[DPARM_RELATIVE_ERROR] = -1;
  dparm[DPARM_SCALED_RESIDUAL] = -1;
  dparm[DPARM_EPSILON_MAGN_CTRL] = 1e-31;
}

extern "C" {

int pastix_amd_set_grid(pastix_amd_data_t** pd, pastix_amd_int_t nx, pastix_amd_int_t ny, pastix_amd_int_t nz) {
  if (!pd) return PASTIX_AMD_ERR_BADPARAMETER;
  if (!*pd) { *pd = new (std::nothrow) pastix_amd_data_s(); if (!*pd) return PASTIX_AMD_ERR_ALLOC; }
  (*pd)->grid[0] = nx; (*pd)->grid[1] = ny; (*pd)->grid[2] = nz;
  return PASTIX_AMD_OK;
}

pastix_amd_plan_t* pastix_amd_data_plan(pastix_amd_data_t* pd) { return pd ? pd->plan : nullptr; }

int pastix_amd_set_schur_unknown_list(pastix_amd_data_t** pd, pastix_amd_int_t n, const pastix_amd_int_t* list) {
  if (!pd || n <= 0 || !list) return PASTIX_AMD_ERR_BADPARAMETER;
  if (!*pd) { *pd = new (std::nothrow) pastix_amd_data_s(); if (!*pd) return PASTIX_AMD_ERR_ALLOC; }
  (*pd)->schur_list.assign(list, list + n);          // converted to 0-based in the ordering task (CSC base)
  return PASTIX_AMD_OK;
}

int pastix_amd_get_schur(pastix_amd_data_t* pd, void* schur) {
  if (!pd || !schur || !pd->plan || !pd->sym || !pd->schur_on || !pd->factorized) return PASTIX_AMD_ERR_BADPARAMETER;
  pastix_amd_int_t info[8];
  pastix_amd_symbol_info(pd->sym, info);
  return pastix_amd_download_cblk(pd->plan, info[1] - 1, schur, nullptr);
}

}  // extern "C"

namespace {
inline double conj_(double x) { return x; }
inline std::complex<double> conj_(const std::complex<double>& x) { return std::conj(x); }
inline double real_(double x) { return x; }
inline double real_(const std::complex<double>& x) { return x.real(); }

// T = double (D_pastix) or std::complex<double> (Z_pastix): the reference compiles pastix.c once per precision
// (redefine_functions.h:73-98); here one template serves both.
// vals32 != NULL (S_pastix): the caller's FLOAT values of a real matrix -- the numerical factorization then runs on the
// native fp32 engine (float arenas, fp32 MFMA kernels) from them; `avals` / `b` are their double copies, which the norm,
// the solve (float factors under double vectors) and the refinement work on.
template <typename T>
void pastix_impl(pastix_amd_data_t** pastix_data, pastix_amd_int_t n, pastix_amd_int_t* colptr, pastix_amd_int_t* row,
                 T* avals, pastix_amd_int_t* perm, pastix_amd_int_t* invp, T* b, pastix_amd_int_t rhs,
                 pastix_amd_int_t* iparm, double* dparm, const float* vals32 = nullptr) {
  constexpr bool CPLX = !std::is_same<T, double>::value;
  const int floattype = CPLX ? PASTIX_AMD_COMPLEXDOUBLE : vals32 ? PASTIX_AMD_REALSINGLE : PASTIX_AMD_REALDOUBLE;
  constexpr size_t TW = sizeof(T) / sizeof(double);
  iparm[IPARM_ERROR_NUMBER] = PASTIX_AMD_OK;
#define FAIL(code) do { iparm[IPARM_ERROR_NUMBER] = (code); return; } while (0)
  if (!pastix_data || n <= 0) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
  if (!*pastix_data) { *pastix_data = new (std::nothrow) pastix_amd_data_s(); if (!*pastix_data) FAIL(PASTIX_AMD_ERR_ALLOC); }
  pastix_amd_data_s* D = *pastix_data;
  const int first = (int)iparm[IPARM_START_TASK], last = (int)iparm[IPARM_END_TASK];
  const int facto = (int)iparm[IPARM_FACTORIZATION];
  const int sym = iparm[IPARM_SYM] == API_SYM_YES || iparm[IPARM_SYM] == API_SYM_HER;
  const bool herm = CPLX && (iparm[IPARM_SYM] == API_SYM_HER || facto == PASTIX_AMD_FACT_LDLH);   // mirrored entries conjugated
  if (iparm[IPARM_DOF_NBR] != 1) FAIL(PASTIX_AMD_ERR_UNSUPPORTED);
  int rc;
  // IPARM_BASEVAL = colptr[0] (pastix.c:1671-1678): the steps below index a 1-based CSC; a 0-based one is rebased
  // into a private copy (perm / invp / the Schur list stay in the caller's base, kass.c:143-156)
  const int64_t base0 = colptr ? colptr[0] : 1;
  std::vector<pastix_amd_int_t> cp1, rw1;
  if (colptr && row && base0 != 1) {
    if (base0 != 0) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
    try {
      cp1.assign(colptr, colptr + n + 1);
      rw1.assign(row, row + (colptr[n] - base0));
    } catch (const std::bad_alloc&) { FAIL(PASTIX_AMD_ERR_ALLOC); }
    for (auto& v : cp1) v += 1 - base0;
    for (auto& v : rw1) v += 1 - base0;
    colptr = cp1.data();
    row = rw1.data();
  }
  if (colptr) iparm[IPARM_BASEVAL] = base0;

  for (int task = first; task <= last; task++) {
    switch (task) {
      case API_TASK_INIT:
        D->n = n;
        break;
      case API_TASK_ORDERING: {
        if (!colptr || !row) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        D->n = n;
        D->perm.assign((size_t)n, 0);
        D->invp.assign((size_t)n, 0);
        if (iparm[IPARM_ORDERING] == API_ORDER_PERSONAL) {
          if (!perm) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
          const int64_t base = base0;                // "same base as the CSC" (kass.c:143-156)
          for (int64_t i = 0; i < n; i++) D->perm[i] = perm[i] - base;
        } else if (D->grid[0] * D->grid[1] * D->grid[2] == n) {
          rc = pastix_amd_order_grid(D->grid[0], D->grid[1], D->grid[2], 8, D->perm.data(), D->invp.data());
          if (rc) FAIL(rc);
        } else {
          // no Scotch / METIS on this box: nested dissection of the graph by level structures (symbolic.cpp)
          rc = pastix_amd_order_graph(n, colptr, row, 64, D->perm.data(), D->invp.data());
          if (rc) FAIL(rc);
        }
        D->schur_on = iparm[IPARM_SCHUR] == API_YES && !D->schur_list.empty();
        if (D->schur_on) {
          // isolate the listed unknowns at the end, keeping the relative order of everything (pastix.c:1404-1540)
          const int64_t base = base0, ns = (int64_t)D->schur_list.size();
          std::vector<char> is_s((size_t)n, 0);
          for (int64_t u : D->schur_list) {
            if (u - base < 0 || u - base >= n || is_s[u - base]) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
            is_s[u - base] = 1;
          }
          std::vector<int64_t> inv((size_t)n);
          for (int64_t i = 0; i < n; i++) inv[D->perm[i]] = i;
          int64_t a = 0, bpos = n - ns;
          for (int64_t p2 = 0; p2 < n; p2++) { const int64_t i = inv[p2]; D->perm[i] = is_s[i] ? bpos++ : a++; }
        }
        break;
      }
      case API_TASK_SYMBFACT: {
        if (!colptr || !row || D->perm.empty()) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        pastix_amd_symbolic_options_t so{};
        so.max_blocksize = (int)iparm[IPARM_MAX_BLOCKSIZE];
        so.min_blocksize = (int)iparm[IPARM_MIN_BLOCKSIZE];
        so.blend_split = 1;                          // IPARM_MIN/MAX_BLOCKSIZE mean what they mean to blend (splitpart.c)
        so.amalgamation_pct = (int)iparm[IPARM_AMALGAMATION_LEVEL];
        if (D->sym) { pastix_amd_symbol_destroy(D->sym); D->sym = nullptr; }
        if (D->schur_on) {
          // the Schur complement is dense: couple the isolated unknowns pairwise in the pattern handed to the
          // symbolic step, which then keeps them as one unsplit cblk (schur_n)
          const int64_t ns = (int64_t)D->schur_list.size();
          std::vector<int64_t> sl(D->schur_list);
          for (auto& u : sl) u -= base0;
          const int64_t base = 1;                    // (colptr / row are 1-based here, see above)
          std::sort(sl.begin(), sl.end());
          std::vector<char> is_s((size_t)n, 0);
          for (int64_t u : sl) is_s[u] = 1;
          std::vector<int64_t> cp2((size_t)n + 1), rw2;
          rw2.reserve((size_t)(colptr[n] - base) + (size_t)ns * (size_t)(ns + 1) / 2);
          for (int64_t j = 0; j < n; j++) {
            cp2[j] = (int64_t)rw2.size() + 1;
            for (int64_t q = colptr[j] - base; q < colptr[j + 1] - base; q++) rw2.push_back(row[q] - base + 1);
            if (is_s[j]) for (int64_t u : sl) if (u > j) rw2.push_back(u + 1);
          }
          cp2[n] = (int64_t)rw2.size() + 1;
          so.schur_n = (int)ns;
          rc = pastix_amd_symbolic(n, cp2.data(), rw2.data(), D->perm.data(), &so, &D->sym);
        } else {
          rc = pastix_amd_symbolic(n, colptr, row, D->perm.data(), &so, &D->sym);
        }
        if (rc) FAIL(rc);
        const pastix_amd_int_t *p, *ip;
        pastix_amd_symbol_perm(D->sym, &p, &ip);
        D->perm.assign(p, p + n);
        D->invp.assign(ip, ip + n);
        const int64_t base = base0;
        if (perm) for (int64_t i = 0; i < n; i++) perm[i] = D->perm[i] + base;    // pastix.c:1734
        if (invp) for (int64_t i = 0; i < n; i++) invp[i] = D->invp[i] + base;
        pastix_amd_int_t info[8];
        pastix_amd_symbol_info(D->sym, info);
        iparm[IPARM_NNZEROS] = info[3];
        break;
      }
      case API_TASK_ANALYSE: {
        if (!D->sym) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        pastix_amd_layout_t lay;
        pastix_amd_symbol_layout(D->sym, &lay);
        pastix_amd_options_t o{};
        o.schur = D->schur_on ? 1 : 0;
        if (D->plan) { pastix_amd_plan_destroy(D->plan); D->plan = nullptr; }
        rc = pastix_amd_plan_create(&lay, facto, floattype, &o, &D->plan);
        if (rc) FAIL(rc);
        dparm[DPARM_FACT_FLOPS] = pastix_amd_fact_flops(&lay, facto, floattype);
        break;
      }
      case API_TASK_NUMFACT: {
        if (D->plan && iparm[IPARM_FILL_MATRIX] == API_YES) {
          // "fake factorisation" (pastix.c:3282, coefinit.c:343-443): no CSC values; critere from the node count
          // (sopalin3d.c:597-598)
          const double eps = dparm[DPARM_EPSILON_MAGN_CTRL];
          const double critere = eps < 0 ? -eps : ((double)n * (double)n + (double)n) * std::sqrt(eps);
          rc = pastix_amd_fill_fake(D->plan, n);
          if (rc) FAIL(rc);
          pastix_amd_stats_t st{};
          rc = pastix_amd_factorize(D->plan, critere, &st);
          dparm[DPARM_FACT_TIME] = st.fact_time;
          iparm[IPARM_STATIC_PIVOTING] = st.nbpivot;
          iparm[IPARM_INERTIA] = st.inertia;
          if (rc) FAIL(rc);
          D->factorized = true;
          break;
        }
        if (!D->plan || !avals) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        // critere = ||A||_1 * sqrt(eps)   (sopalin3d.c:586-606, CscNorm1 csc_intern_compute.c:120)
        std::vector<double> colsum((size_t)n, 0.0);
        for (int64_t j = 0; j < n; j++)
          for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
            const int64_t i = row[q] - 1;
            colsum[j] += std::abs(avals[q]);
            if (sym && i != j) colsum[i] += std::abs(avals[q]);
          }
        double nrm = 0;
        for (double v : colsum) nrm = std::max(nrm, v);
        D->norm1 = nrm;
        const double eps = dparm[DPARM_EPSILON_MAGN_CTRL];
        const double critere = eps < 0 ? -eps : nrm * std::sqrt(eps);
        rc = pastix_amd_fill_csc(D->plan, sym, n, colptr, row, vals32 ? (const void*)vals32 : (const void*)avals, D->perm.data());
        if (rc) FAIL(rc);
        pastix_amd_stats_t st{};
        rc = pastix_amd_factorize(D->plan, critere, &st);
        dparm[DPARM_FACT_TIME] = st.fact_time;           // sopalin3d.c:1125-1132
        iparm[IPARM_STATIC_PIVOTING] = st.nbpivot;        // pastix.c:3853
        iparm[IPARM_INERTIA] = st.inertia;
        if (rc) FAIL(rc);
        D->factorized = true;
        break;
      }
      case API_TASK_SOLVE: {
        if (!D->factorized || !b) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        if (D->schur_on) FAIL(PASTIX_AMD_ERR_UNSUPPORTED);
        std::vector<T> x((size_t)n);
        D->rhs.resize((size_t)(n * rhs) * TW);
        std::memcpy(D->rhs.data(), b, (size_t)(n * rhs) * sizeof(T));
        for (int64_t r = 0; r < rhs; r++) {
          T* br = b + r * n;
          for (int64_t i = 0; i < n; i++) x[D->perm[i]] = br[i];
          rc = pastix_amd_solve(D->plan, x.data(), 1);
          if (rc) FAIL(rc);
          for (int64_t i = 0; i < n; i++) br[i] = x[D->perm[i]];
        }
        break;
      }
      case API_TASK_REFINE: {
        // pastix_task_raff (pastix.c:4300-4500) picks the refiner from IPARM_REFINEMENT (api.h:353-365): GMRES
        // (raff_gmres.c), conjugate gradient for the symmetric factorizations (raff_grad.c), plain iterative
        // refinement (raff_pivot.c).  All three are preconditioned by the device solve with the factors and stop at
        // ||b - A x|| / ||b|| < DPARM_EPSILON_REFINEMENT or after IPARM_ITERMAX iterations; all four run on
        // the device (csrc/refine.hip): Krylov vectors, SpMV, dot products; the host sees scalars.
        if (D->schur_on) FAIL(PASTIX_AMD_ERR_UNSUPPORTED);
        if (!D->factorized || !b || !avals || D->rhs.size() != (size_t)(n * rhs) * TW) FAIL(PASTIX_AMD_ERR_BADPARAMETER);
        pastix_amd_int_t iters = 0;
        double relerr = 0;
        rc = pastix_amd_refine(D->plan, (int)iparm[IPARM_REFINEMENT], sym ? (herm ? 2 : 1) : 0, n, colptr, row, avals,
                               D->perm.data(), D->rhs.data(), b, rhs, dparm[DPARM_EPSILON_REFINEMENT],
                               iparm[IPARM_ITERMAX], (int)iparm[IPARM_GMRES_IM], &iters, &relerr);
        if (rc) FAIL(rc);
        iparm[IPARM_NBITER] = iters;
        dparm[DPARM_RELATIVE_ERROR] = relerr;
        break;
      }
      case API_TASK_CLEAN:
        if (D->plan) pastix_amd_plan_destroy(D->plan);
        if (D->sym) pastix_amd_symbol_destroy(D->sym);
        delete D;
        *pastix_data = nullptr;
        return;
      default:
        FAIL(PASTIX_AMD_ERR_BADPARAMETER);
    }
  }
#undef FAIL
}
}  // namespace

extern "C" {

void pastix_amd_pastix(pastix_amd_data_t** pastix_data, int pastix_comm, pastix_amd_int_t n,
                       pastix_amd_int_t* colptr, pastix_amd_int_t* row, void* avals, pastix_amd_int_t* perm,
                       pastix_amd_int_t* invp, void* b, pastix_amd_int_t rhs, pastix_amd_int_t* iparm,
                       double* dparm) {
  (void)pastix_comm;
  if (!iparm || !dparm) return;
  if (iparm[IPARM_MODIFY_PARAMETER] == API_NO) {      // pastix.c:4755-4761: fill defaults and return
    init_param(iparm, dparm);
    return;
  }
  if (iparm[IPARM_FLOAT] == PASTIX_AMD_REALDOUBLE)
    pastix_impl<double>(pastix_data, n, colptr, row, (double*)avals, perm, invp, (double*)b, rhs, iparm, dparm);
  else if (iparm[IPARM_FLOAT] == PASTIX_AMD_COMPLEXDOUBLE)
    pastix_impl<std::complex<double>>(pastix_data, n, colptr, row, (std::complex<double>*)avals, perm, invp,
                                      (std::complex<double>*)b, rhs, iparm, dparm);
  else if (iparm[IPARM_FLOAT] == PASTIX_AMD_REALSINGLE || iparm[IPARM_FLOAT] == PASTIX_AMD_COMPLEXSINGLE) {
    // S_pastix / C_pastix (the reference's -DPREC_SIMPLE builds, redefine_functions.h:73-98): avals / b are float /
    // float complex.  Real: the factorization runs on the native fp32 engine from the float values; complex: there is no
    // native complex-single engine, the factorization runs in fp64 on the widened values.  Either way the vectors are
    // double inside (norm, solve, refinement) and rounded to the caller's type at the end.
    static const float k_no_values = 0.0f;
    const bool cplx = iparm[IPARM_FLOAT] == PASTIX_AMD_COMPLEXSINGLE;
    const int64_t nnz = (colptr && n > 0) ? (int64_t)(colptr[n] - colptr[0]) : 0;
    const size_t w = cplx ? 2 : 1;
    try {
      std::vector<double> ad, bd;
      if (avals) { ad.resize((size_t)nnz * w); for (size_t i = 0; i < ad.size(); i++) ad[i] = (double)((const float*)avals)[i]; }
      if (b && rhs > 0) { bd.resize((size_t)n * (size_t)rhs * w); for (size_t i = 0; i < bd.size(); i++) bd[i] = (double)((const float*)b)[i]; }
      if (cplx)
        pastix_impl<std::complex<double>>(pastix_data, n, colptr, row, avals ? (std::complex<double>*)ad.data() : nullptr, perm, invp,
                                          b ? (std::complex<double>*)bd.data() : nullptr, rhs, iparm, dparm);
      else
        pastix_impl<double>(pastix_data, n, colptr, row, avals ? ad.data() : nullptr, perm, invp, b ? bd.data() : nullptr, rhs, iparm,
                            dparm, avals ? (const float*)avals : &k_no_values /* (steps without values: only the type counts) */);
      if (b) for (size_t i = 0; i < bd.size(); i++) ((float*)b)[i] = (float)bd[i];
    } catch (const std::bad_alloc&) {
      iparm[IPARM_ERROR_NUMBER] = PASTIX_AMD_ERR_ALLOC;
    }
  } else
    iparm[IPARM_ERROR_NUMBER] = PASTIX_AMD_ERR_UNSUPPORTED;
}

}  // extern "C"
