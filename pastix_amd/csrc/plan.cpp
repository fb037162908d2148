// plan.cpp -- host-side analysis of a SolverMatrix layout for the device engine.
//
// What the reference does at run time with counters, mutexes and linear searches
// (sopalin3d.c:790-1025, sopalin_compute.c:865-1032) is resolved here once:
//   * dependency levels: cblk t can be factorized once every source cblk k with a blok facing t
//     has been factorized and its contributions applied (TASK_CTRBCNT semantics, solver.h:71-75);
//   * for every (source cblk k, blok i, blok j >= i) the destination of the contribution inside
//     the facing cblk (add_contrib_local, sopalin_compute.c:427-429) is precomputed and cut into
//     TM x TN tiles of the target panel ("pieces");
//   * pieces are grouped by (launch slot, target tile): within one launch a target tile is
//     owned by exactly one workgroup, which replaces mutex_blok[b3] (sopalin_compute.c:563-580)
//     and makes the accumulation order deterministic.
#include "plan.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <sys/mman.h>
#include <memory>
#include <numeric>
#include <thread>

namespace pastix_amd {

// DPARM_FACT_FLOPS: symbCost (blend_symbol_cost.c:52-88), flops_dpotrf (:382-430),
// flops_dgetrf (:282-330), macros flops.h:74-75,91-100,103-108,116-117,211-214.
double fact_flops(const pastix_amd_layout_t* L, int factotype, int floattype) {
  auto fmuls_potrf = [](double n) { return n * (((1. / 6.) * n + 0.5) * n + (1. / 3.)); };
  auto fadds_potrf = [](double n) { return n * (((1. / 6.) * n) * n - (1. / 6.)); };
  auto fmuls_getrf = [](double n) { return 0.5 * n * (n * (n - (1. / 3.) * n - 1.) + n) + (2. / 3.) * n; };
  auto fadds_getrf = [](double n) { return 0.5 * n * (n * (n - (1. / 3.) * n) - n) + (1. / 6.) * n; };
  const bool lu = factotype == PASTIX_AMD_FACT_LU;
  const bool cplx = floattype == PASTIX_AMD_COMPLEXSINGLE || floattype == PASTIX_AMD_COMPLEXDOUBLE;
  double muls = 0, adds = 0;
  for (int64_t k = 0; k < L->cblknbr; k++) {
    const auto& c = L->cblktab[k];
    double N = double(c.lcolnum - c.fcolnum + 1), M = double(c.stride) - N, rem = M, fac = lu ? 2. : 1.;
    if (lu) { muls += fmuls_getrf(N); adds += fadds_getrf(N); }
    else    { muls += fmuls_potrf(N); adds += fadds_potrf(N); }
    muls += fac * 0.5 * M * N * (N + 1.);
    adds += fac * 0.5 * M * N * (N + 1.);   // FADDS_TRSM is defined as FMULS_TRMM (flops.h:99-100)
    for (int64_t b = c.bloknum + 1; b < L->cblktab[k + 1].bloknum; b++) {
      double h = double(L->bloktab[b].lrownum - L->bloktab[b].frownum + 1);
      muls += fac * rem * h * N;
      adds += fac * rem * h * N;
      rem -= h;
    }
  }
  return cplx ? 6. * muls + 2. * adds : muls + adds;
}

namespace {
constexpr uint16_t AB(int a, int b) { return (uint16_t)(a | (b << 2)); }   // piece flags: arenas of A and B
struct RawPiece {
  int64_t tile;   // target tile id (+ntile for the U arena)
  int32_t lvl;    // level of the source cblk
  uint8_t carena; // arena (plane) of the target: 0 L, 1 U, 2/3 their imaginary planes
  uint8_t shared; // target receives contributions from several ranks (windowed schedule)
  uint16_t gtid;  // gathered piece: the host thread whose list holds its row maps (p.dr | p.dc << 16 = index there)
  Piece p;
};

// Big, write-once host buffers of the plan (several GB at 200^3): 2-MB aligned and advised for transparent huge pages --
// a 4-KB first-touch fault per 85 pieces was a third of the planning time.
inline void advise_huge(void* p, size_t bytes) {
  const uintptr_t H = (uintptr_t)2 << 20;
  const uintptr_t a = ((uintptr_t)p + H - 1) & ~(H - 1), e = ((uintptr_t)p + bytes) & ~(H - 1);
  if (e > a) (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
}
template <class T>
T* huge_alloc(size_t n) {
  const size_t H = (size_t)2 << 20, raw = std::max<size_t>(n, 1) * sizeof(T);
  if (raw < H) {                                   // (small: an ordinary allocation)
    void* p = malloc(raw);
    if (!p) throw std::bad_alloc();
    return (T*)p;
  }
  const size_t bytes = (raw + H - 1) / H * H;
  void* p = aligned_alloc(H, bytes);
  if (!p) throw std::bad_alloc();
  advise_huge(p, bytes);
  return (T*)p;
}
struct FreeDeleter { void operator()(void* p) const { free(p); } };
// append-only list in blocks (4096 elements first, doubling up to 2^20): no reallocation copies, no estimate of the final
// size, and a plan of a few hundred pieces does not map megabytes per host thread
template <class T>
struct BlockList {
  static constexpr size_t BMAX = (size_t)1 << 20;
  std::vector<std::unique_ptr<T, FreeDeleter>> blk;
  std::vector<size_t> cap;
  size_t n = 0, used = 0;                          // elements in all / in the last block
  void push_back(const T& v) {
    if (blk.empty() || used == cap.back()) {
      const size_t c = blk.empty() ? 4096 : std::min(BMAX, cap.back() * 2);
      blk.emplace_back(huge_alloc<T>(c));
      cap.push_back(c);
      used = 0;
    }
    blk.back().get()[used++] = v;
    n++;
  }
  size_t size() const { return n; }
  template <class F> void for_each(F&& f) const {
    for (size_t b = 0; b < blk.size(); b++) {
      const T* q = blk[b].get();
      const size_t m = b + 1 < blk.size() ? cap[b] : used;
      for (size_t i = 0; i < m; i++) f(q[i]);
    }
  }
  void release() { blk.clear(); blk.shrink_to_fit(); cap.clear(); n = used = 0; }
};
}  // namespace

static int check_layout(const pastix_amd_layout_t* L) {
  if (!L || L->cblknbr <= 0 || !L->cblktab || !L->bloktab) return PASTIX_AMD_ERR_BADPARAMETER;
  if (L->cblktab[L->cblknbr].bloknum != L->bloknbr) return PASTIX_AMD_ERR_LAYOUT;
  int64_t prevl = -1;
  for (int64_t k = 0; k < L->cblknbr; k++) {
    const auto& c = L->cblktab[k];
    int64_t fb = c.bloknum, lb = L->cblktab[k + 1].bloknum;
    if (c.fcolnum != prevl + 1 || c.lcolnum < c.fcolnum || lb <= fb) return PASTIX_AMD_ERR_LAYOUT;
    prevl = c.lcolnum;
    // first blok is the diagonal blok (compute_diag.c:550)
    if (L->bloktab[fb].frownum != c.fcolnum || L->bloktab[fb].lrownum != c.lcolnum) return PASTIX_AMD_ERR_LAYOUT;
    int64_t off = 0, lastrow = -1;
    for (int64_t b = fb; b < lb; b++) {
      const auto& bl = L->bloktab[b];
      if (bl.coefind != off || bl.lrownum < bl.frownum || bl.frownum <= lastrow) return PASTIX_AMD_ERR_LAYOUT;
      if (bl.cblknum < k || bl.cblknum >= L->cblknbr) return PASTIX_AMD_ERR_LAYOUT;
      if (b > fb && bl.cblknum <= k) return PASTIX_AMD_ERR_LAYOUT;
      const auto& f = L->cblktab[bl.cblknum];
      if (bl.frownum < f.fcolnum || bl.lrownum > f.lcolnum) return PASTIX_AMD_ERR_LAYOUT;
      off += bl.lrownum - bl.frownum + 1;
      lastrow = bl.lrownum;
    }
    if (off != c.stride) return PASTIX_AMD_ERR_LAYOUT;
  }
  return PASTIX_AMD_OK;
}

// Which ranks contribute into which target blok (the reference's FanInTarget regions, ftgt.h:67-113, at blok
// granularity): bit r of mask[b] is set when a cblk owned by rank r != owner(cblk of b) has a blok pair (i, j >= i)
// whose rows land in blok b.  Deterministic from (layout, owner): sender and receiver derive the same compact
// fan-in buffers from it.
int fanin_touched(const pastix_amd_layout_t* L, const int32_t* owner, uint64_t* mask) {
  const int64_t nc = L->cblknbr;
  for (int64_t b = 0; b < L->bloknbr; b++) mask[b] = 0;
  for (int64_t k = 0; k < nc; k++) {
    const int32_t r = owner[k];
    if (r < 0 || r >= 64) return PASTIX_AMD_ERR_UNSUPPORTED;
    const int64_t fb = L->cblktab[k].bloknum, lb = L->cblktab[k + 1].bloknum;
    for (int64_t i = fb + 1; i < lb; i++) {
      const int64_t t = L->bloktab[i].cblknum;
      if (owner[t] == r) continue;
      const int64_t tlb = L->cblktab[t + 1].bloknum;
      int64_t b3 = L->cblktab[t].bloknum;
      for (int64_t j = i; j < lb; j++) {
        const int64_t fj = L->bloktab[j].frownum, lj = L->bloktab[j].lrownum;
        while (b3 < tlb && !(fj >= L->bloktab[b3].frownum && lj <= L->bloktab[b3].lrownum)) b3++;
        if (b3 >= tlb) return PASTIX_AMD_ERR_LAYOUT;
        mask[b3] |= 1ull << r;
      }
    }
  }
  return PASTIX_AMD_OK;
}

// Ownership view of a layout for one rank (multi-GPU fan-in): owned cblks are factorized here; a remote
// cblk that receives contributions from an owned source gets a zero-initialised shadow panel in which the (negated)
// local contributions are accumulated (add_contrib_target, sopalin_compute.c:600-733).  Shadow panels are COMPACT:
// only the bloks this rank contributes into (fanin_touched), packed in blok order with their own leading dimension.
// Fills role, tstride, tcoef, poff, fanin_mask, owner, myrank and the dependency levels from P.cblk / P.blok / P.opts.
int owner_view(const pastix_amd_layout_t* L, const int32_t* owner, int32_t myrank, Plan& P) {
  const int64_t nc = P.cblknbr;
  int rc;
  P.myrank = owner ? myrank : 0;
  P.owner.clear();
  if (owner) P.owner.assign(owner, owner + nc);
  P.role.assign(nc, owner ? 0 : 1);
  if (owner) {
    for (int64_t k = 0; k < nc; k++) if (owner[k] == myrank) P.role[k] = 1;
    for (int64_t k = 0; k < nc; k++) {
      if (P.role[k] != 1) continue;
      for (int64_t b = P.cblk[k].bloknum + 1; b < P.cblk[k + 1].bloknum; b++)
        if (P.role[P.blok[b].cblknum] == 0) P.role[P.blok[b].cblknum] = 2;
    }
  }
  P.tstride.resize(nc);
  P.tcoef.resize(P.bloknbr);
  for (int64_t k = 0; k < nc; k++) P.tstride[k] = P.cblk[k].stride;
  for (int64_t b = 0; b < P.bloknbr; b++) P.tcoef[b] = P.blok[b].coefind;
  P.fanin_mask.clear();
  if (owner) {
    P.fanin_mask.assign((size_t)P.bloknbr, 0);
    if ((rc = fanin_touched(L, owner, P.fanin_mask.data()))) return rc;
    const std::vector<uint64_t>& mask = P.fanin_mask;
    for (int64_t t = 0; t < nc; t++) {
      if (P.role[t] != 2) continue;
      int64_t off = 0;
      for (int64_t b = P.cblk[t].bloknum; b < P.cblk[t + 1].bloknum; b++) {
        if ((mask[b] >> myrank) & 1ull) { P.tcoef[b] = off; off += P.blok[b].lrownum - P.blok[b].frownum + 1; }
        else P.tcoef[b] = -1;
      }
      P.tstride[t] = off;
    }
  }
  P.poff.resize(nc + 1);
  P.poff[0] = 0;
  for (int64_t k = 0; k < nc; k++) {
    int64_t w = P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1;
    if (w > MAXW && !(P.opts.schur && k == nc - 1)) return PASTIX_AMD_ERR_UNSUPPORTED;
    if (P.cblk[k].stride > 0x7fffffffLL) return PASTIX_AMD_ERR_UNSUPPORTED;
    P.poff[k + 1] = P.poff[k] + (P.role[k] ? P.tstride[k] * w : 0);
  }
  // dependency levels: cblk t can be factorized once every source cblk with a blok facing it has been
  P.level.assign(nc, 0);
  for (int64_t k = 0; k < nc; k++)
    for (int64_t b = P.cblk[k].bloknum + 1; b < P.cblk[k + 1].bloknum; b++) {
      int64_t t = P.blok[b].cblknum;
      P.level[t] = std::max(P.level[t], P.level[k] + 1);
    }
  P.nlevels = 1 + *std::max_element(P.level.begin(), P.level.end());
  return PASTIX_AMD_OK;
}

int build_plan(const pastix_amd_layout_t* L, int factotype, int floattype,
               const pastix_amd_options_t* opts, const int32_t* owner, int32_t myrank, Plan& P) {
  int rc = check_layout(L);
  if (rc) return rc;
  const bool cplx = floattype == PASTIX_AMD_COMPLEXDOUBLE;
  // real single precision shares the plan with real double (the arenas hold floats, the kernels are kernels_f32.hip)
  if (floattype != PASTIX_AMD_REALDOUBLE && floattype != PASTIX_AMD_REALSINGLE && !cplx) return PASTIX_AMD_ERR_UNSUPPORTED;
  if (floattype == PASTIX_AMD_REALSINGLE && owner) return PASTIX_AMD_ERR_UNSUPPORTED;   // (one GPU)
  if (factotype == PASTIX_AMD_FACT_LDLH && !cplx) factotype = PASTIX_AMD_FACT_LDLT;   // real `he` is `sy`
  if (factotype != PASTIX_AMD_FACT_LLT && factotype != PASTIX_AMD_FACT_LDLT && factotype != PASTIX_AMD_FACT_LU &&
      factotype != PASTIX_AMD_FACT_LDLH)
    return PASTIX_AMD_ERR_UNSUPPORTED;
  // z: complex symmetric LDLt (`sy`), Hermitian LDLh (`he`) and LU (`ge`).  The reference's complex `po` mixes
  // symmetric (csqrt, geru, TRSM "T") and Hermitian (zherk, GEMM "N","C") kernels and is not reproduced.
  if (cplx && factotype == PASTIX_AMD_FACT_LLT) return PASTIX_AMD_ERR_UNSUPPORTED;
  P.factotype = factotype;
  P.floattype = floattype;
  if (opts) P.opts = *opts;
  // chunk size: contributions into one tile are applied in groups whose accumulated inner
  // dimension reaches `chunk_k` (<=0: default 512; 1: every source on its own = right-looking;
  // huge: one group per tile = left-looking)
  // Default: 512 (max 8 pieces per task) for small problems where parallelism is scarce, 1024 from 1e12 flop,
  // 2048 (16) for large ones where the tile read-modify-write and task prologue/epilogue matter more
  // (MI355X: 200^3 512 -> 7.81 s, 1024 -> 7.57 s, 2048 -> 7.53 s; 100^3 512 -> 0.1580 s, 1024 -> 0.1553 s;
  // 130^3 512 -> 0.6015 s, 1024 -> 0.5873 s; 60^3 512 -> 19.6 ms, 1024 -> 20.2 ms).
  const double fl_total = fact_flops(L, factotype, floattype);
  const bool big = fl_total > 5e13;
  // Where the run schedule is built (below) launches have no tails to balance, and what a task costs besides its chunks
  // weighs more than parallelism: longer tasks from 4e12 flop (MI355X, run schedule: 100^3 1024 -> 133.4 ms, 2048 -> 131.9;
  // 130^3 548.6 -> 533.9; 160^3 1815 -> 1774; 80^3 43.8 -> 45.4: stays 1024; 60^3 512 -> 14.56, 1024 -> 14.38).
  // Up to which size the run is built by default.  Round 5: 2e14 flop (200^3 has 4.1e14) -- its reader lists, 366 M pairs at
  // 200^3, cost 2.5 s of host analysis there for +1.3-1.6 % of the rate.  Round 6: where the lists are built on the device
  // (defer_run_edges, run_edges.hip) and the run is ONE kernel (real LLt / LDLt: no second persistent kernel)
  // there is no such price: the cap is the 32-bit range of the tables' indices.
  const double run_cap = (P.defer_run_edges && floattype == PASTIX_AMD_REALDOUBLE &&
                          (factotype == PASTIX_AMD_FACT_LLT || factotype == PASTIX_AMD_FACT_LDLT)) ? 1.5e15 : 2e14;
  const bool run_built = !owner && P.opts.run_schedule >= 0 &&
                         ((floattype == PASTIX_AMD_REALDOUBLE &&
                           (factotype == PASTIX_AMD_FACT_LLT || factotype == PASTIX_AMD_FACT_LDLT ||
                            factotype == PASTIX_AMD_FACT_LU)) ||
                          (cplx && (factotype == PASTIX_AMD_FACT_LDLT || factotype == PASTIX_AMD_FACT_LDLH)));
  if (P.opts.lookahead <= 0)
    P.opts.lookahead = (run_built && (P.opts.run_schedule == 1 || fl_total <= run_cap)) ? (fl_total > 4e12 ? 2048 : 1024)
                                                                                     : (big ? 2048 : fl_total > 1e12 ? 1024 : 512);
  const double chunk_work = double(TM) * TN * double(P.opts.lookahead);
  const int max_pieces = P.opts.lookahead >= 4096 ? 32 : P.opts.lookahead >= 2048 ? 16 : 8;
  const int64_t nc = L->cblknbr;
  P.cblknbr = nc;
  P.bloknbr = L->bloknbr;
  P.cblk.assign(L->cblktab, L->cblktab + nc + 1);
  P.blok.assign(L->bloktab, L->bloktab + L->bloknbr);
  // (the update kernels' LDS-DMA runs on 32-bit buffer descriptors whose out-of-range markers are 2^30 and 2^31,
  // kernels_update.hip: a source panel's k-lines -- width + one chunk of them -- must stay below 2^30 bytes; a panel of 10^6
  // rows by 128 columns: no layout that fits a device comes near)
  if (floattype != PASTIX_AMD_REALSINGLE)
    for (int64_t k = 0; k < nc; k++)
      if ((P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1 + 16) * (int64_t)P.cblk[k].stride * 8 >= ((int64_t)1 << 30)) return PASTIX_AMD_ERR_UNSUPPORTED;
  if ((rc = owner_view(L, owner, myrank, P))) return rc;
  // cblks that receive contributions from more than one rank ("shared"): their contributions are
  // scheduled left-looking with a window (see below): a source older than `window` levels before the target
  // contributes in the bulk launch `window` levels ahead of the target (second stream, off the critical
  // path), the last `window` sources contribute as usual, and only the level just before the target is
  // urgent -- so the fan-in exchange of a level never waits for a long left-looking update
  std::vector<uint8_t> shared(nc, 0);
  if (owner)
    for (int64_t k = 0; k < nc; k++)
      for (int64_t b = P.cblk[k].bloknum + 1; b < P.cblk[k + 1].bloknum; b++)
        if (owner[k] != owner[P.blok[b].cblknum]) shared[P.blok[b].cblknum] = 1;
  const int window = getenv("PASTIX_AMD_WINDOW") ? atoi(getenv("PASTIX_AMD_WINDOW")) : 4;
  P.coefnbr = P.poff[nc];
  P.ncol = P.cblk[nc - 1].lcolnum + 1;
  P.fact_flops = fact_flops(L, factotype, floattype);
  P.local_flops = 0;
  for (int64_t k = 0; k < nc; k++) {
    if (P.role[k] != 1) continue;
    pastix_amd_cblk_t two[2] = {P.cblk[k], P.cblk[k + 1]};
    two[0].fcolnum = 0; two[0].lcolnum = P.cblk[k].lcolnum - P.cblk[k].fcolnum;
    const int64_t b0 = two[0].bloknum;
    two[0].bloknum = 0; two[1].bloknum -= b0;
    pastix_amd_layout_t one{1, two[1].bloknum, two, P.blok.data() + b0};
    P.local_flops += fact_flops(&one, factotype, floattype);
  }

  const bool ptime = dev_opt("plan_timing") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tph = tnow();
  auto phase = [&](const char* name) { if (ptime) { double t = tnow(); fprintf(stderr, "[plan] %-28s %.2f s\n", name, t - tph); tph = t; } };
  const int NL = P.nlevels;             // (dependency levels: owner_view)

  // cblks by level
  P.lvl_cblk_ptr.assign(NL + 1, 0);
  // Schur mode: the last cblk receives its contributions but is not factorized (sopalin_compute.c:767-772)
  auto factored = [&](int64_t k) { return P.role[k] == 1 && !(P.opts.schur && k == nc - 1); };
  for (int64_t k = 0; k < nc; k++) if (factored(k)) P.lvl_cblk_ptr[P.level[k] + 1]++;
  for (int l = 0; l < NL; l++) P.lvl_cblk_ptr[l + 1] += P.lvl_cblk_ptr[l];
  const int64_t nowned = P.lvl_cblk_ptr[NL];
  P.lvl_cblk.resize(nowned);
  {
    std::vector<int64_t> pos(P.lvl_cblk_ptr.begin(), P.lvl_cblk_ptr.end() - 1);
    for (int64_t k = 0; k < nc; k++) if (factored(k)) P.lvl_cblk[pos[P.level[k]]++] = (int32_t)k;
  }

  // ---- the run: the longest suffix of levels with at most run_max_cblks cblks each (plan.h, RunInfo) -----------------
  // Built for one GPU, real double (LLt, LDLt, LU).  The plan carries BOTH schedules: the level-by-level launches work on
  // the same tables, which of the two runs is decided per factorization (api.cpp).
  P.run_L0 = -1;
  {
    const bool built = !owner && ((floattype == PASTIX_AMD_REALDOUBLE &&
                                   (factotype == PASTIX_AMD_FACT_LLT || factotype == PASTIX_AMD_FACT_LDLT ||
                                    factotype == PASTIX_AMD_FACT_LU)) ||
                                  (cplx && (factotype == PASTIX_AMD_FACT_LDLT || factotype == PASTIX_AMD_FACT_LDLH)));
    // Above 2e14 flop (200^3: 4.1e14) the run is on request only: launches of tens of rounds of workgroups have little to
    // gain (200^3: +0.6 %) and the run's tables cost there (366 M dependency edges: 1.8 s of analysis, 3 GB).
    if (built && P.opts.run_schedule >= 0 && (P.opts.run_schedule == 1 || fl_total <= run_cap)) {
      // (where the run begins: levels of at most 32 cblks; 128 from 1e14 flop -- 200^3, one box, run_max_cblks 32 / 64 / 128 /
      // 512: 6640 / 6617 / 6604 / 6616 ms; 130^3 32 / 128 / 1024: 527.4 / 527.0 / 527.4: flat below)
      const int64_t maxc = P.opts.run_max_cblks > 0 ? P.opts.run_max_cblks : (fl_total > 1e14 ? 128 : 32);
      int L0 = NL;
      auto narrow = [&](int l) {       // (the run's panel kernel takes cblks of at most 128 columns, like k_diag_llt_w)
        for (int64_t q = P.lvl_cblk_ptr[l]; q < P.lvl_cblk_ptr[l + 1]; q++) {
          const int32_t k = P.lvl_cblk[(size_t)q];
          if (P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1 > TN) return false;
        }
        return true;
      };
      while (L0 > 0 && P.lvl_cblk_ptr[L0] - P.lvl_cblk_ptr[L0 - 1] <= maxc && narrow(L0 - 1)) L0--;
      if (NL - L0 >= 2) P.run_L0 = L0;
    }
  }
  const int RL0 = P.run_L0 >= 0 ? P.run_L0 : NL + 1;
  const int near = dev_opt("near") ? atoi(dev_opt("near")) : 3;
  const int64_t nearc = dev_opt("nearc") ? atoi(dev_opt("nearc")) : 1;

  // ---- panel / trsm tasks per level ------------------------------------------------------------
  P.lvl_panel_ptr.assign(NL + 1, 0);
  P.lvl_trsm_ptr.assign(NL + 1, 0);
  P.panel_tasks.resize(nowned);
  P.dinv_ws = 0;
  int64_t ws_run = 0;                    // (levels of the run overlap in time: their tile inverses do not share space)
  for (int l = 0; l < NL; l++) {
    if (l == RL0) ws_run = P.dinv_ws;
    int64_t ws = l >= RL0 ? ws_run : 0;
    P.lvl_panel_ptr[l] = P.lvl_cblk_ptr[l];
    P.lvl_trsm_ptr[l] = (int64_t)P.trsm_tasks.size();
    // widest / tallest first: better tail behaviour inside a launch
    std::sort(P.lvl_cblk.begin() + P.lvl_cblk_ptr[l], P.lvl_cblk.begin() + P.lvl_cblk_ptr[l + 1],
              [&](int32_t a, int32_t b) {
                int64_t wa = P.cblk[a].lcolnum - P.cblk[a].fcolnum, wb = P.cblk[b].lcolnum - P.cblk[b].fcolnum;
                if (wa != wb) return wa > wb;
                return a < b;
              });
    for (int64_t q = P.lvl_cblk_ptr[l]; q < P.lvl_cblk_ptr[l + 1]; q++) {
      int32_t k = P.lvl_cblk[q];
      int32_t w = (int32_t)(P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1), s = (int32_t)P.cblk[k].stride;
      PanelTask pt{P.poff[k], s, w, ws};
      P.panel_tasks[q] = pt;
      for (int32_t r = w; r < s; r += 64) {
        TrsmTask tt{P.poff[k], s, w, r, std::min(64, s - r), ws};
        P.trsm_tasks.push_back(tt);
      }
      ws += (int64_t)((w + 15) / 16) * 256 * (factotype == PASTIX_AMD_FACT_LU ? 2 : 1) * (cplx ? 2 : 1);
    }
    if (l >= RL0) ws_run = ws;
    P.dinv_ws = std::max(P.dinv_ws, ws);
  }
  P.lvl_panel_ptr[NL] = nowned;
  P.lvl_trsm_ptr[NL] = (int64_t)P.trsm_tasks.size();

  phase("setup, levels, panel tasks");
  // ---- update pieces -----------------------------------------------------------------------------
  // tile numbering: tile_base[t] + rt * nct(t) + ct, times 2 arenas for LU
  std::vector<int64_t> tile_base(nc + 1, 0);
  for (int64_t t = 0; t < nc; t++) {
    int64_t w = P.cblk[t].lcolnum - P.cblk[t].fcolnum + 1;
    tile_base[t + 1] = tile_base[t] + ((P.tstride[t] + TM - 1) / TM) * ((w + TN - 1) / TN);
  }
  const int64_t ntile = tile_base[nc];
  const bool lu = factotype == PASTIX_AMD_FACT_LU;
  const bool ldlt = factotype == PASTIX_AMD_FACT_LDLT || factotype == PASTIX_AMD_FACT_LDLH;
  const bool herm = factotype == PASTIX_AMD_FACT_LDLH;
  // Source cblks are independent: piece generation runs on `nthr` host threads (PASTIX_AMD_PLAN_THREADS, default
  // min(32, cores)), every thread into its own list; the lists are then bucketed by target tile and sorted in
  // parallel.  The sort key is a total order, so the result does not depend on the number of threads.
  const int nthr = [] {
    const char* e = getenv("PASTIX_AMD_PLAN_THREADS");
    int n = e ? atoi(e) : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, std::min(n, 64));
  }();
  std::vector<BlockList<RawPiece>> traw((size_t)nthr);
  std::vector<std::vector<uint32_t>> tmaps((size_t)nthr);   // gathered pieces: their row maps, 64 words each
  // (rectangles per tile and source cblk from which they become one gathered piece; options.gather_min, default 2 for fragmented source cblks; the
  // fp32 kernel and the fan-in schedule of the multi-GPU driver take rectangles only)
  const int gmo = dev_opt("gather") ? atoi(dev_opt("gather")) : P.opts.gather_min;     // (developer override of the option)
  const int64_t gather_off = (int64_t)1 << 40;
  // By default a layout gathers when it IS fragmented: its off-diagonal bloks per (source cblk, facing cblk) pair average 1.5
  // or more (blend on lexicographically numbered separators: 3-4; blend on contiguously numbered ones and this repository's
  // own layouts: 1.0-1.2, and there the few tiles that two or three rectangles of one source reach are better off in the
  // rectangle loops -- a task with a gathered piece runs ALL its pieces through the 4-byte gathering loop: 60^3 on the own
  // layout -1.7 ... -2.6 % with gathering on).  An explicit options.gather_min gathers whatever the layout looks like.
  // (On such a layout every source cblk is looked at: skipping those whose off-diagonal bloks average 48 rows or more -- a few
  // tall bloks beside dozens of fragments -- cost 10 % at 100^3.)
  double frag = 0;
  {
    int64_t noff = 0, ngrp = 0;
    for (int64_t k = 0; k < nc; k++) {
      if (P.role[k] != 1) continue;
      for (int64_t b = P.cblk[k].bloknum + 1; b < P.cblk[k + 1].bloknum; b++) {
        noff++;
        ngrp += (b == P.cblk[k].bloknum + 1) || P.blok[b].cblknum != P.blok[b - 1].cblknum;
      }
    }
    frag = ngrp > 0 ? double(noff) / double(ngrp) : 0.0;
  }
  const int64_t gather_on = (owner || floattype == PASTIX_AMD_REALSINGLE || gmo < 0 || (gmo == 0 && frag < 1.5)) ? gather_off
                                                                                                          : (gmo > 0 ? gmo : 2);
  const double gather_tall = 1e30;
  std::vector<double> tuf((size_t)nthr, 0.0), tub((size_t)nthr, 0.0);
  std::vector<int> terr((size_t)nthr, 0);
  std::atomic<int64_t> gen_next{0};
  auto gen_body = [&](int tid) {
    BlockList<RawPiece>& raw = traw[(size_t)tid];
    double uflops = 0, ubytes = 0;

    // one piece (all its planes) into tile (rt, ct) of target panel t: target rows [r0, r1) x columns [c0, c1) (panel
    // coordinates) from the source rows a_src.. / b_src.. of panel k.  gmap >= 0: a GATHERED piece -- r0/r1/c0/c1 then only
    // count the source rows (m = r1 - r0, n = c1 - c0) and where they land is in this thread's map list at gmap.
    auto push_tile = [&](int64_t k, int64_t t, int64_t rt, int64_t ct, int64_t a_src, int64_t b_src, int64_t r0, int64_t r1,
                         int64_t c0, int64_t c1, uint16_t flags, uint8_t carena, int64_t gmap) {
      const int64_t w_t = P.cblk[t].lcolnum - P.cblk[t].fcolnum + 1;
      const int64_t nct = (w_t + TN - 1) / TN;
      const int64_t sk = P.cblk[k].stride;
      const int64_t wk = P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1;
      auto push = [&](uint16_t fl, uint8_t ca) {
        RawPiece rp;
        rp.tile = tile_base[t] + rt * nct + ct + (int64_t)ca * ntile;
        // "lvl" = launch slot - 1.  Local targets: as soon as the source is factorized.  Shared
        // targets: not before `window` levels ahead of the target's own level.
        rp.lvl = shared[t] ? std::max(P.level[k], P.level[t] - 1 - window) : P.level[k];
        rp.carena = ca;
        rp.shared = shared[t];
        rp.gtid = (uint16_t)tid;
        rp.p.a_off = P.poff[k] + a_src;
        rp.p.b_off = P.poff[k] + b_src;
        rp.p.lda = (int32_t)sk;
        rp.p.k = (uint16_t)wk;
        rp.p.m = (uint16_t)(r1 - r0);
        rp.p.n = (uint16_t)(c1 - c0);
        if (gmap >= 0) {
          rp.p.dr = (uint16_t)(gmap & 0xffff);
          rp.p.dc = (uint16_t)(gmap >> 16);
          rp.p.flags = (uint16_t)(fl | PIECE_GATHERED);
        } else {
          rp.p.dr = (uint16_t)(r0 - rt * TM);
          rp.p.dc = (uint16_t)(c0 - ct * TN);
          rp.p.flags = fl;
        }
        raw.push_back(rp);
        uflops += 2.0 * double(r1 - r0) * double(c1 - c0) * double(wk);
        ubytes += 8.0 * double(wk) * double((r1 - r0) + (c1 - c0));
      };
      if (!cplx) {
        push(flags, carena);
      } else if (herm) {
        // Hermitian product on split planes (SOPALIN_GEMM "N","C": the B operand, L D, is conjugated):
        //   C_re -= A_re B_re^T + A_im B_im^T ;  C_im -= A_im B_re^T - A_re B_im^T
        const int a = flags & 3, b = (flags >> 2) & 3;
        push(AB(a, b), carena);
        push(AB(a + 2, b + 2), carena);
        push(AB(a + 2, b), (uint8_t)(carena + 2));
        push((uint16_t)(AB(a, b + 2) | 16), (uint8_t)(carena + 2));
      } else {
        // complex symmetric product on split planes (no conjugation, SOPALIN_GEMM "N","T"):
        //   C_re -= A_re B_re^T - A_im B_im^T ;  C_im -= A_re B_im^T + A_im B_re^T
        const int a = flags & 3, b = (flags >> 2) & 3;
        push(AB(a, b), carena);
        push((uint16_t)(AB(a + 2, b + 2) | 16), carena);
        push(AB(a, b + 2), (uint8_t)(carena + 2));
        push(AB(a + 2, b), (uint8_t)(carena + 2));
      }
    };
    // the rectangle [trow,trow+nrows) x [tcol,tcol+ncols) of target panel t, split into tiles
    auto emit = [&](int64_t k, int64_t t, int64_t a_row, int64_t b_row, int64_t trow, int64_t nrows,
                    int64_t tcol, int64_t ncols, uint16_t flags, uint8_t carena) {
      for (int64_t rt = trow / TM; rt * TM < trow + nrows; rt++) {
        const int64_t r0 = std::max(trow, rt * TM), r1 = std::min(trow + nrows, (rt + 1) * TM);
        for (int64_t ct = tcol / TN; ct * TN < tcol + ncols; ct++) {
          const int64_t c0 = std::max(tcol, ct * TN), c1 = std::min(tcol + ncols, (ct + 1) * TN);
          push_tile(k, t, rt, ct, a_row + (r0 - trow), b_row + (c0 - tcol), r0, r1, c0, c1, flags, carena, -1);
        }
      }
    };
    // ---- gathered pieces -------------------------------------------------------------------------------------------------
    // What one source cblk k contributes to one tile of target t is a set of rectangles: (runs of its bloks j that land
    // contiguously in t's panel) x (its bloks i facing t).  On layouts whose bloks are fragments (blend on separators whose
    // nodes are numbered across their low-side neighbours: 2-4 rows every 50-60) that is dozens of rectangles of a few rows
    // and columns per tile, each a pass of K / 16 latency-bound chunks through the update loop.  Their SOURCE rows are
    // consecutive in k's panel (its bloks are stacked) -- only the landing is scattered --, so from `gather_min` rectangles
    // on the tile gets them as ONE piece: m consecutive source rows, n consecutive source rows for the columns, and two row
    // maps that say where each lands (Piece flag 32; the update kernel's MODE 3 loop gathers while it stages).  Which
    // products are formed and in which order they are accumulated per tile entry (source cblks in the order of the list)
    // does not change.  Real double and complex double on one GPU (the fp32 kernel and the fan-in schedule take rectangles).
    int64_t gather_min = gather_off;                          // (set per source cblk)
    struct Frag { int64_t src, dst, len; };                  // source row in k's panel, target row / column in t's panel
    auto clip_tiles = [](const std::vector<Frag>& in, int64_t T, std::vector<std::pair<int64_t, Frag>>& out) {
      out.clear();                                          // (tile, fragment with dst relative to the panel): dst ascending
      for (const Frag& f : in)
        for (int64_t tt = f.dst / T; tt * T < f.dst + f.len; tt++) {
          const int64_t d0 = std::max(f.dst, tt * T), d1 = std::min(f.dst + f.len, (tt + 1) * T);
          out.emplace_back(tt, Frag{f.src + (d0 - f.dst), d0, d1 - d0});
        }
    };
    std::vector<std::pair<int64_t, Frag>> ta, tb;
    auto make_map = [&](const std::pair<int64_t, Frag>* fa, size_t na, const std::pair<int64_t, Frag>* fb, size_t nb) -> int64_t {
      std::vector<uint32_t>& G = tmaps[(size_t)tid];
      const int64_t idx = (int64_t)(G.size() / 64);
      G.resize(G.size() + 64, 0xffffffffu);
      uint32_t* w = G.data() + idx * 64;
      auto put = [&](uint32_t* ww, const std::pair<int64_t, Frag>* f, size_t n, int64_t T) {
        const int64_t s0 = f[0].second.src;
        for (size_t x = 0; x < n; x++)
          for (int64_t r = 0; r < f[x].second.len; r++) {
            const int64_t slot = f[x].second.dst + r - f[x].first * T;         // row / column inside the tile
            const uint32_t v = (uint32_t)(f[x].second.src + r - s0);
            uint32_t& word = ww[slot & 31];
            const int sh = 8 * (int)(slot >> 5);
            word = (word & ~(0xffu << sh)) | (v << sh);
          }
      };
      put(w, fa, na, TM);
      put(w + 32, fb, nb, TN);
      return idx;
    };
    // rows `rows` x columns `cols` of target t from source k: per tile either the rectangles or one gathered piece
    auto emit_set = [&](int64_t k, int64_t t, const std::vector<Frag>& rows, const std::vector<Frag>& cols, uint16_t flags,
                        uint8_t carena) {
      if (gather_min == gather_off) {                       // (this source cblk does not gather: the rectangles as they are)
        for (const Frag& fr : rows)
          for (const Frag& fc : cols) emit(k, t, fr.src, fc.src, fr.dst, fr.len, fc.dst, fc.len, flags, carena);
        return;
      }
      clip_tiles(rows, TM, ta);
      clip_tiles(cols, TN, tb);
      for (size_t a0 = 0; a0 < ta.size();) {
        size_t a1 = a0;
        while (a1 < ta.size() && ta[a1].first == ta[a0].first) a1++;
        for (size_t b0 = 0; b0 < tb.size();) {
          size_t b1 = b0;
          while (b1 < tb.size() && tb[b1].first == tb[b0].first) b1++;
          const int64_t rt = ta[a0].first, ct = tb[b0].first;
          int64_t m = 0, n = 0;
          for (size_t x = a0; x < a1; x++) m += ta[x].second.len;
          for (size_t x = b0; x < b1; x++) n += tb[x].second.len;
          if ((int64_t)((a1 - a0) * (b1 - b0)) >= gather_min) {
            const int64_t g = make_map(&ta[a0], a1 - a0, &tb[b0], b1 - b0);
            push_tile(k, t, rt, ct, ta[a0].second.src, tb[b0].second.src, 0, m, 0, n, flags, carena, g);
          } else {
            for (size_t x = a0; x < a1; x++)
              for (size_t y = b0; y < b1; y++) {
                const Frag& fa = ta[x].second;
                const Frag& fb = tb[y].second;
                push_tile(k, t, rt, ct, fa.src, fb.src, fa.dst, fa.dst + fa.len, fb.dst, fb.dst + fb.len, flags, carena, -1);
              }
          }
          b0 = b1;
        }
        a0 = a1;
      }
    };
    std::vector<Frag> landD, runsB, rowsI, colsG, colI(1);

    // (source cblks are handed out from the top of the tree down, a few at a time: the cblks of the top separators have
    // hundreds of bloks -- O(bloks^2) pieces each -- and would otherwise decide which thread finishes last)
    for (;;) {
      const int64_t kc = gen_next.fetch_add(8);
      if (kc >= nc) break;
      for (int64_t k = nc - 1 - kc; k >= std::max<int64_t>(0, nc - 8 - kc); k--) {
      if (P.role[k] != 1) continue;               // contributions are computed by the source's owner
      const int64_t fb = P.cblk[k].bloknum, lb = P.cblk[k + 1].bloknum;
      {
        const int64_t wk0 = P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1;
        gather_min = (lb - fb > 1 && double(P.cblk[k].stride - wk0) < gather_tall * double(lb - fb - 1)) ? gather_on : gather_off;
        // (the gathering loop's buffer descriptors and k-line offsets are 32-bit byte counts, kernels_update.hip piece_loop_g:
        // a source panel whose (width + one chunk) k-lines do not fit 2^31 bytes keeps its rectangles)
        if ((wk0 + 16) * (int64_t)P.cblk[k].stride * 8 >= ((int64_t)1 << 31)) gather_min = gather_off;
      }
      // groups of consecutive bloks [g0, g1) facing the same cblk t (they land in t's diagonal blok; every later blok of
      // k lands in an off-diagonal blok of t: containment, sopalin_compute.c:558-559)
      for (int64_t g0 = fb + 1, g1; g0 < lb; g0 = g1) {
        const int64_t t = P.blok[g0].cblknum;
        g1 = g0;
        while (g1 < lb && P.blok[g1].cblknum == t) g1++;
        const int64_t tf = P.cblk[t].fcolnum;
        const int64_t tfb = P.cblk[t].bloknum, tlb = P.cblk[t + 1].bloknum;
        landD.clear();
        runsB.clear();
        colsG.clear();
        int64_t b3 = tfb;
        for (int64_t j = g0; j < lb; j++) {
          const int64_t fj = P.blok[j].frownum, lj = P.blok[j].lrownum, hj = lj - fj + 1;
          while (b3 < tlb && !(fj >= P.blok[b3].frownum && lj <= P.blok[b3].lrownum)) b3++;
          if (b3 >= tlb) { terr[tid] = PASTIX_AMD_ERR_LAYOUT; return; }   // containment (sopalin_compute.c:558-559)
          if (P.tcoef[b3] < 0) { terr[tid] = PASTIX_AMD_ERR_LAYOUT; return; }   // (cannot happen: fanin_touched marks exactly these)
          const int64_t dst = P.tcoef[b3] + (fj - P.blok[b3].frownum);
          if ((b3 == tfb) != (j < g1)) { terr[tid] = PASTIX_AMD_ERR_LAYOUT; return; }
          if (j < g1) {
            landD.push_back(Frag{P.blok[j].coefind, dst, hj});
            colsG.push_back(Frag{P.blok[j].coefind, fj - tf, hj});
          } else if (!runsB.empty() && dst == runsB.back().dst + runsB.back().len) {
            runsB.back().len += hj;                    // runs of source bloks that land contiguously in the target panel
          } else {
            runsB.push_back(Frag{P.blok[j].coefind, dst, hj});
          }
        }
        // the rows BELOW t's diagonal blok: every blok of the group against every run
        if (!runsB.empty()) {
          if (!lu) {
            // LLt: C_L -= L_j L_i^T ; LDLt: C_L -= L_j (L D)_i^T with L D kept in the U arena
            emit_set(k, t, runsB, colsG, ldlt ? AB(0, 1) : AB(0, 0), 0);
          } else {
            emit_set(k, t, runsB, colsG, AB(0, 1), 0);   // L U^T -> L arena
            emit_set(k, t, runsB, colsG, AB(1, 0), 1);   // U L^T -> U arena
          }
        }
        // t's diagonal blok: blok i against the bloks j >= i of the group
        for (int64_t i = g0; i < g1; i++) {
          const int64_t hi = P.blok[i].lrownum - P.blok[i].frownum + 1;
          const int64_t tcol = P.blok[i].frownum - tf;
          if (!lu) {
            rowsI.clear();
            for (int64_t j = i; j < g1; j++) {
              const Frag& f = landD[(size_t)(j - g0)];
              if (!rowsI.empty() && f.dst == rowsI.back().dst + rowsI.back().len) rowsI.back().len += f.len;
              else rowsI.push_back(f);
            }
            colI[0] = Frag{P.blok[i].coefind, tcol, hi};
            emit_set(k, t, rowsI, colI, ldlt ? AB(0, 1) : AB(0, 0), 0);
          } else {
            // LU: per blok pair -- the transposed U contribution needs the (i, j) roles individually
            // (sopalin_compute.c:430-435,567-579)
            for (int64_t j = i; j < g1; j++) {
              const Frag& f = landD[(size_t)(j - g0)];
              emit(k, t, f.src, P.blok[i].coefind, f.dst, f.len, tcol, hi, AB(0, 1), 0);   // lower/diag part
              // C_L[cols of i as rows, rows of j as cols] -= L_i ... transposed U result:
              // (U_j L_i^T)^T = L_i U_j^T  -> rows = rows of i (tcol..), cols = rows of j (dst..)
              if (j != i) emit(k, t, P.blok[i].coefind, f.src, tcol, hi, f.dst, f.len, AB(0, 1), 0);
            }
          }
        }
      }
      }
    }
    tuf[(size_t)tid] = uflops;
    tub[(size_t)tid] = ubytes;
  };
  // (an exception must not leave a worker thread: std::terminate would take the host process down)
  auto gen = [&](int tid) {
    try { gen_body(tid); } catch (const std::bad_alloc&) { terr[(size_t)tid] = PASTIX_AMD_ERR_ALLOC; }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; t++) th.emplace_back(gen, t);
    gen(0);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nthr; t++) if (terr[(size_t)t]) return terr[(size_t)t];
  double uflops = 0, ubytes = 0;                       // (integer-valued doubles below 2^53: the sums are exact)
  for (int t = 0; t < nthr; t++) { uflops += tuf[(size_t)t]; ubytes += tub[(size_t)t]; }
  P.update_flops = uflops;

  phase("piece generation");
  // ---- group into tasks --------------------------------------------------------------------------
  // Per target tile the contributions are ordered by source level and cut into chunks of about
  // chunk_work multiply-adds; a chunk is launched in the slot right after its last source level, so
  // the tile is read-modified-written once per chunk instead of once per source (the reference does
  // it once per source blok pair under mutex_blok, sopalin_compute.c:563-580).
  auto piece_less = [](const RawPiece& a, const RawPiece& b) {
    if (a.tile != b.tile) return a.tile < b.tile;
    if (a.lvl != b.lvl) return a.lvl < b.lvl;
    if (a.p.a_off != b.p.a_off) return a.p.a_off < b.p.a_off;   // deterministic accumulation order
    if (a.p.b_off != b.p.b_off) return a.p.b_off < b.p.b_off;
    if (a.p.flags != b.p.flags) return a.p.flags < b.p.flags;   // (complex: the planes of one product)
    if (a.p.dr != b.p.dr) return a.p.dr < b.p.dr;
    return a.p.dc < b.p.dc;
  };
  struct RawArr {                      // (uninitialised storage: a vector would zero ~10 GB on one thread first)
    std::unique_ptr<RawPiece, FreeDeleter> p;
    size_t n = 0;
    size_t size() const { return n; }
    RawPiece& operator[](size_t i) { return p.get()[i]; }
    const RawPiece& operator[](size_t i) const { return p.get()[i]; }
    RawPiece* begin() { return p.get(); }
  } raw;
  {
    // counting sort into tile bins (placement by thread order), then every bin sorted on its own.  The bins hold about
    // the same number of pieces each -- their boundaries are quantiles of a sample of the tile numbers; bins of equal
    // tile ranges left the tiles of the top separators, thousands of pieces each, to a few threads.  Which bins there are
    // does not matter for the result: the sort key is a total order with the tile first.
    constexpr int64_t NB = 8192;
    std::vector<int64_t> splitter;                 // first tile of bins 1 .. : ascending, distinct
    {
      std::vector<int64_t> sample;
      for (int t = 0; t < nthr; t++) {
        size_t i = 0;
        traw[(size_t)t].for_each([&](const RawPiece& r) { if (i++ % 61 == 0) sample.push_back(r.tile); });
      }
      std::sort(sample.begin(), sample.end());
      for (int64_t b2 = 1; b2 < NB && !sample.empty(); b2++) {
        const int64_t v = sample[(size_t)((__int128)sample.size() * b2 / NB)];
        if (splitter.empty() || splitter.back() < v) splitter.push_back(v);
      }
    }
    auto bin_of = [&](int64_t tile) {
      return (int64_t)(std::upper_bound(splitter.begin(), splitter.end(), tile) - splitter.begin());
    };
    std::vector<std::vector<int64_t>> cnt((size_t)nthr, std::vector<int64_t>((size_t)NB + 1, 0));
    auto par = [&](auto&& fn) {
      auto guarded = [&](int t) {
        try { fn(t); } catch (const std::bad_alloc&) { terr[(size_t)t] = PASTIX_AMD_ERR_ALLOC; }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < nthr; t++) th.emplace_back(guarded, t);
      guarded(0);
      for (auto& x : th) x.join();
    };
    par([&](int t) { traw[(size_t)t].for_each([&](const RawPiece& r) { cnt[(size_t)t][(size_t)bin_of(r.tile)]++; }); });
    std::vector<int64_t> binoff((size_t)NB + 1, 0);
    for (int64_t b2 = 0; b2 < NB; b2++) {
      int64_t c = 0;
      for (int t = 0; t < nthr; t++) { const int64_t x = cnt[(size_t)t][(size_t)b2]; cnt[(size_t)t][(size_t)b2] = binoff[(size_t)b2] + c; c += x; }
      binoff[(size_t)b2 + 1] = binoff[(size_t)b2] + c;
    }
    raw.n = (size_t)binoff[(size_t)NB];
    raw.p.reset(huge_alloc<RawPiece>(raw.n + 1));
    par([&](int t) {
      std::vector<int64_t>& pos = cnt[(size_t)t];
      traw[(size_t)t].for_each([&](const RawPiece& r) { raw[(size_t)pos[(size_t)bin_of(r.tile)]++] = r; });
      traw[(size_t)t].release();
    });
    std::atomic<int64_t> next{0};
    par([&](int) {
      for (;;) {
        const int64_t b2 = next.fetch_add(1);
        if (b2 >= NB) break;
        std::sort(raw.begin() + binoff[(size_t)b2], raw.begin() + binoff[(size_t)b2 + 1], piece_less);
      }
    });
  }
  for (int t = 0; t < nthr; t++) if (terr[(size_t)t]) return terr[(size_t)t];
  phase("piece sort");
  // (room for the clipped copies the quadrant tasks below append, without touching the memory now)
  P.pieces.reserve(raw.size() + raw.size() / 3 + 1024);
  advise_huge(P.pieces.data(), P.pieces.capacity() * sizeof(Piece));
  P.pieces.resize(raw.size());
  {
    // (the row maps of the gathered pieces are numbered in the order of the sorted list: the plan does not depend on which
    // host thread generated what)
    const size_t n = raw.size(), per = (n + (size_t)nthr - 1) / (size_t)nthr;
    std::vector<size_t> gcount((size_t)nthr + 1, 0);
    auto run_par = [&](auto&& fn) {
      std::vector<std::thread> th;
      for (int t = 1; t < nthr; t++) th.emplace_back(fn, t);
      fn(0);
      for (auto& x : th) x.join();
    };
    run_par([&](int t) {
      size_t c = 0;
      for (size_t i = (size_t)t * per; i < std::min(n, ((size_t)t + 1) * per); i++) c += (raw[i].p.flags & PIECE_GATHERED) != 0;
      gcount[(size_t)t + 1] = c;
    });
    for (int t = 0; t < nthr; t++) gcount[(size_t)t + 1] += gcount[(size_t)t];
    P.gmaps.assign(gcount[(size_t)nthr] * 64, 0xffffffffu);
    run_par([&](int t) {
      size_t g = gcount[(size_t)t];
      for (size_t i = (size_t)t * per; i < std::min(n, ((size_t)t + 1) * per); i++) {
        Piece pc = raw[i].p;
        if (pc.flags & PIECE_GATHERED) {
          const size_t loc = (size_t)pc.dr | ((size_t)pc.dc << 16);
          std::copy_n(tmaps[(size_t)raw[i].gtid].data() + loc * 64, 64, P.gmaps.data() + g * 64);
          pc.dr = (uint16_t)(g & 0xffff);
          pc.dc = (uint16_t)(g >> 16);
          g++;
        }
        P.pieces[i] = pc;
      }
    });
    for (auto& v : tmaps) std::vector<uint32_t>().swap(v);
  }
  P.tasks.clear();
  P.slot_task_ptr.assign(NL + 1, 0);
  std::vector<double> task_work;
  std::vector<int32_t> task_slot;
  std::vector<uint8_t> task_urgent;
  std::vector<int64_t> task_tile;
  P.slot_flops.assign(NL, 0.0);
  P.slot_urgent_flops.assign(NL, 0.0);
  P.slot_pieces.assign(NL, 0);
  P.slot_maxpn.assign(NL, 0);
  P.slot_maxwork.assign(NL, 0.0);
  // quadrant tasks (below): a task qualifies when its pieces fill less than quad_fill of the 128 x 128 x 16 chunks
  // k_update would run for them, and a slot gets them when it has at least quad_min candidates
  constexpr bool quad_on = true;
  const double quad_fill = P.opts.quadrant_fill_pct > 0 ? 0.01 * P.opts.quadrant_fill_pct : 0.25;
  // (single precision: no quadrant tasks, k_update_s takes every task)
  const int64_t quad_min = floattype == PASTIX_AMD_REALSINGLE ? (int64_t)1 << 60 : P.opts.quadrant_min > 0 ? P.opts.quadrant_min : 1024;
  const int quad_maxpiece = P.opts.quadrant_fill_pct > 100 ? 128 : 64;   // (tests force every partial task with fill > 100 %)
  // The tiles are independent: the sorted piece list is cut at tile boundaries into one range per host thread, every
  // thread groups its tiles into its own lists, which are concatenated in range order (= the serial result).
  struct GOut {
    std::vector<Task> tasks;
    std::vector<double> work;
    std::vector<int32_t> slot;
    std::vector<uint8_t> urgent;
    std::vector<int64_t> tile;
    std::vector<double> slot_flops, slot_urgent_flops, slot_maxwork;
    std::vector<int64_t> slot_pieces, slot_cnt;
    std::vector<int32_t> slot_maxpn;
    std::vector<Piece> part_tmp;
    double urgent_flops = 0, full_flops = 0, ubytes = 0;
  };
  const int gthr = nthr;
  std::vector<GOut> gout((size_t)gthr);
  std::vector<size_t> gcut((size_t)gthr + 1, raw.size());
  gcut[0] = 0;
  for (int t = 1; t < gthr; t++) {
    size_t c = raw.size() * (size_t)t / (size_t)gthr;
    while (c < raw.size() && c > 0 && raw[c].tile == raw[c - 1].tile) c++;
    gcut[(size_t)t] = std::max(c, gcut[(size_t)t - 1]);
  }
  auto group = [&](int gt) {
    GOut& O = gout[(size_t)gt];
    const size_t qb = gcut[(size_t)gt], qe = gcut[(size_t)gt + 1];
    O.slot_flops.assign(NL, 0.0);
    O.slot_urgent_flops.assign(NL, 0.0);
    O.slot_maxwork.assign(NL, 0.0);
    O.slot_pieces.assign(NL, 0);
    O.slot_cnt.assign(NL, 0);
    O.slot_maxpn.assign(NL, 0);
      for (size_t q = qb; q < qe;) {
      size_t e = q;
      double work = 0;
      int tlev;
      {
        const int64_t tile0 = raw[q].tile - (int64_t)raw[q].carena * ntile;
        const int64_t tt = std::upper_bound(tile_base.begin(), tile_base.end(), tile0) - tile_base.begin() - 1;
        tlev = P.level[tt];
      }
      while (e < qe && raw[e].tile == raw[q].tile) {
        work += double(raw[e].p.m) * raw[e].p.n * raw[e].p.k;
        e++;
        // close the chunk once enough work is gathered, but never split pieces of one source level
        // (many small pieces are flushed early by count: each costs a latency-bound pass, and keeping
        // them for the tile's last chunk would put them on the critical path of the dependency chain)
        if (raw[q].shared) {
          // shared tile: every slot gets its own tasks; a long list is cut into several tasks that run
          // concurrently and combine with f64 atomics (split-K), so that the few tiles of one target
          // cblk still fill the chip
          if (e == qe || raw[e].tile != raw[q].tile || raw[e].lvl != raw[e - 1].lvl ||
              work >= chunk_work || (int)(e - q) >= max_pieces) break;
          continue;
        }
        if ((work >= chunk_work || (int)(e - q) >= max_pieces) &&
            (e == qe || raw[e].tile != raw[q].tile || raw[e].lvl != raw[e - 1].lvl)) break;
        // contributions from the level right below the target's are the only ones that cannot be computed
        // before that level's panel kernels: keep them in tasks of their own (the urgent set of their slot) and
        // flush everything older one slot earlier, where it overlaps with the panel kernels (api.cpp, two streams)
        if (e < qe && raw[e].tile == raw[q].tile && raw[e].lvl == tlev - 1 &&
            raw[e - 1].lvl < tlev - 1) break;
        // run levels: a task computes all its pieces when its LAST source is solved, and the dependency chain reaches a
        // target k levels above that source k periods (~100 us) later -- sixteen accumulated pieces (~230 us) in front of
        // the urgent task of a tile were what the chain of the top separator waited for.  The sources of the last `near`
        // levels below the target stay in tasks of their own level.
        if (near > 0 && tlev >= RL0 && P.lvl_cblk_ptr[tlev + 1] - P.lvl_cblk_ptr[tlev] <= nearc && e < qe && raw[e].tile == raw[q].tile && raw[e].lvl != raw[e - 1].lvl &&
            raw[e].lvl >= tlev - near) break;
      }
      int slot = raw[e - 1].lvl + 1;
      int64_t tile = raw[q].tile;
      uint8_t carena = raw[q].carena;
      tile -= (int64_t)carena * ntile;
      int64_t t = std::upper_bound(tile_base.begin(), tile_base.end(), tile) - tile_base.begin() - 1;
      int64_t w_t = P.cblk[t].lcolnum - P.cblk[t].fcolnum + 1, nct = (w_t + TN - 1) / TN;
      int64_t rt = (tile - tile_base[t]) / nct, ct = (tile - tile_base[t]) % nct;
      Task tk{};
      tk.c_off = P.poff[t] + rt * TM + ct * TN * P.tstride[t];
      tk.ldc = (int32_t)P.tstride[t];
      tk.tm = (uint16_t)std::min<int64_t>(TM, P.tstride[t] - rt * TM);
      tk.tn = (uint16_t)std::min<int64_t>(TN, w_t - ct * TN);
      tk.p0 = (int32_t)q;
      tk.pn = (int32_t)(e - q);
      tk.flags = carena | (raw[q].shared ? 4u : 0u);
      {   // pieces that cover the whole valid tile (any K: the kernel pads the last chunk with zero lines) first: the kernel runs them
          // through its specialized loop; tk.nfull = how many
        auto isfull = [&](const Piece& pc) {   // covers the whole valid tile (tm x tn; 128 x 128 except at the edges)
          return !(pc.flags & PIECE_GATHERED) && pc.dr == 0 && pc.dc == 0 && pc.m == tk.tm && pc.n == tk.tn && pc.k > 0;
        };
        // (manual stable partition through a reused scratch vector: std::stable_partition allocates per call)
        O.part_tmp.clear();
        size_t wpos = q;
        for (size_t z = q; z < e; z++) {
          if (isfull(P.pieces[z])) P.pieces[wpos++] = P.pieces[z];
          else O.part_tmp.push_back(P.pieces[z]);
        }
        std::copy(O.part_tmp.begin(), O.part_tmp.end(), P.pieces.begin() + wpos);
        auto mid = P.pieces.begin() + wpos;
        tk.nfull = (uint32_t)(mid - (P.pieces.begin() + q));
        for (auto it = P.pieces.begin() + q; it != mid; ++it) O.full_flops += 2.0 * it->m * (double)it->n * it->k;
        for (auto it = P.pieces.begin() + q; it != P.pieces.begin() + e; ++it) {
          if (it->flags & 16) tk.flags |= 8u;            // the update kernel needs its sign-flipping variant
          if (it->flags & PIECE_GATHERED) tk.flags |= TASK_GATHERED;   // ... its gathering loop, for all pieces of the task
        }

      }
      const uint8_t urg = P.level[t] == slot ? 2 : P.level[t] == slot + 1 ? 1 : 0;
      if (P.level[t] == slot) { O.urgent_flops += 2.0 * work; O.slot_urgent_flops[slot] += 2.0 * work; }
      O.slot_flops[slot] += 2.0 * work;
      O.slot_maxwork[slot] = std::max(O.slot_maxwork[slot], work);
      // candidate for quadrant tasks (split after the grouping, once the number of candidates per slot is known)
      // -- tasks of SMALL pieces only: every piece at most 64 x 64 (the chains inside the leaf domains: 10-40 rows and
      // columns).  Larger pieces would be cut at the quadrant borders into up to four clipped copies for a kernel that
      // is built for latency, not for flops: on blend's layouts (cblks of 60-120 columns, bloks of at most 120 rows:
      // hardly any whole-tile piece, so that the fill rule alone diverted most of the work) the quadrant kernel took
      // 44 % of the time of an 80^3 factorization driven by the real PaStiX.
      if (quad_on && !raw[q].shared && tk.nfull == 0 && slot < RL0 && !(tk.flags & TASK_GATHERED)) {     // (the run takes whole tiles only)
        double iters = 0;
        int maxmn = 0;
        for (size_t z = q; z < e; z++) {
          iters += double((P.pieces[z].k + 15) / 16);
          maxmn = std::max<int>(maxmn, std::max<int>(P.pieces[z].m, P.pieces[z].n));
        }
        if (maxmn <= quad_maxpiece && work < quad_fill * iters * 16.0 * double(TM) * double(TN)) tk.flags |= 64u;
      }
      O.tasks.push_back(tk);
      O.work.push_back(work + 4096.0 * double(e - q));
      O.slot.push_back(slot);
      O.urgent.push_back(urg);
      O.tile.push_back(raw[q].tile);
      O.slot_cnt[slot]++;
      O.ubytes += 16.0 * double(tk.tm) * double(tk.tn);
      O.slot_pieces[slot] += (int64_t)(e - q);
      O.slot_maxpn[slot] = std::max<int32_t>(O.slot_maxpn[slot], (int32_t)(e - q));
      q = e;
    }
  };
  {
    std::vector<int> gerr((size_t)gthr, 0);
    auto guarded = [&](int t) { try { group(t); } catch (const std::bad_alloc&) { gerr[(size_t)t] = PASTIX_AMD_ERR_ALLOC; } };
    std::vector<std::thread> th;
    for (int t = 1; t < gthr; t++) th.emplace_back(guarded, t);
    guarded(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < gthr; t++) if (gerr[(size_t)t]) return gerr[(size_t)t];
  }
  {
    size_t nt = 0;
    for (const GOut& O : gout) nt += O.tasks.size();
    const size_t ntr = quad_on ? nt + nt / 2 : nt;   // (room for the quadrant tasks appended below)
    P.tasks.reserve(ntr);
    task_work.reserve(ntr);
    task_slot.reserve(ntr);
    task_urgent.reserve(ntr);
    task_tile.reserve(ntr);
    for (GOut& O : gout) {
      P.tasks.insert(P.tasks.end(), O.tasks.begin(), O.tasks.end());
      task_work.insert(task_work.end(), O.work.begin(), O.work.end());
      task_slot.insert(task_slot.end(), O.slot.begin(), O.slot.end());
      task_urgent.insert(task_urgent.end(), O.urgent.begin(), O.urgent.end());
      task_tile.insert(task_tile.end(), O.tile.begin(), O.tile.end());
      P.urgent_flops += O.urgent_flops;
      P.full_flops += O.full_flops;
      ubytes += O.ubytes;
      for (int sl = 0; sl < NL; sl++) {
        P.slot_flops[sl] += O.slot_flops[sl];
        P.slot_urgent_flops[sl] += O.slot_urgent_flops[sl];
        P.slot_pieces[sl] += O.slot_pieces[sl];
        P.slot_task_ptr[sl + 1] += O.slot_cnt[sl];
        P.slot_maxpn[sl] = std::max(P.slot_maxpn[sl], O.slot_maxpn[sl]);
        P.slot_maxwork[sl] = std::max(P.slot_maxwork[sl], O.slot_maxwork[sl]);
      }
      GOut().tasks.swap(O.tasks);
    }
  }
  if (raw.size() > 0x7fffffffULL) return PASTIX_AMD_ERR_UNSUPPORTED;
  // ---- quadrant tasks ----------------------------------------------------------------------------
  // A task of small pieces (the chains inside the leaf domains: a handful of pieces of 10-40 rows and columns, K of
  // 30-60) is latency-bound in k_update -- a chunk iteration costs the same whatever the piece covers, and two
  // workgroups fit a CU.  Where a launch has many of them (>= quad_min candidates among the urgent / among the bulk
  // tasks of a slot; a handful would only add a launch), each is cut into the four 64x64 quadrants of its tile, its
  // pieces clipped to them: ordinary tasks on a tile of valid extent <= 64 (Task flag 32), which k_update_small
  // (kernels_small.hip: four waves, eight workgroups per CU) runs right behind the slot's k_update launch.  Ownership
  // is unchanged: the quadrants are disjoint, every one sees its pieces in the order of the list.  The parent leaves
  // the schedule (slot -1); its pieces stay where they are, the clipped copies are appended.
  if (quad_on) {
    std::vector<int64_t> cand((size_t)NL * 2, 0);
    const size_t nt0 = P.tasks.size();
    bool any = false;
    for (size_t i = 0; i < nt0; i++)
      if (P.tasks[i].flags & 64u) cand[(size_t)task_slot[i] * 2 + (task_urgent[i] == 2 ? 1 : 0)]++;
    for (int64_t c : cand) any = any || c >= quad_min;
    // the candidates are cut on the host threads (ranges of the task list, every thread into lists of its own, which
    // are appended in range order: the result does not depend on the thread count)
    struct QOut {
      std::vector<Task> tasks;
      std::vector<double> work;
      std::vector<int32_t> slot;
      std::vector<uint8_t> urgent;
      std::vector<Piece> pieces;                   // Task::p0 is relative to this list until it is appended to P.pieces
      std::vector<std::pair<int32_t, int32_t>> dcnt;   // (slot, change of its task count)
      double dbytes = 0;
      int err = 0;
    };
    const int qthr = any ? nthr : 1;
    std::vector<QOut> qout((size_t)qthr);
    auto cut = [&](int qt) {
      QOut& O = qout[(size_t)qt];
      const size_t per = (nt0 + (size_t)qthr - 1) / (size_t)qthr;
      const size_t ib = (size_t)qt * per, ie = std::min(nt0, ib + per);
      for (size_t i = ib; i < ie; i++) {
        if (!(P.tasks[i].flags & 64u)) continue;
        P.tasks[i].flags &= ~64u;
        const int slot = task_slot[i];
        if (cand[(size_t)slot * 2 + (task_urgent[i] == 2 ? 1 : 0)] < quad_min) continue;
        const Task tk = P.tasks[i];
        int made = 0;
        for (int qd = 0; qd < 4; qd++) {
          const int qr = (qd & 1) * 64, qc = (qd >> 1) * 64;
          if (qr >= (int)tk.tm || qc >= (int)tk.tn) continue;
          const size_t sp0 = O.pieces.size();
          double wq = 0;
          bool neg = false;
          for (int z = 0; z < tk.pn; z++) {
            const Piece& pc = P.pieces[(size_t)tk.p0 + (size_t)z];
            const int r0 = std::max<int>(pc.dr, qr), r1 = std::min<int>(pc.dr + pc.m, qr + 64);
            const int c0 = std::max<int>(pc.dc, qc), c1 = std::min<int>(pc.dc + pc.n, qc + 64);
            if (r1 <= r0 || c1 <= c0) continue;
            Piece cp = pc;
            cp.a_off += r0 - pc.dr;
            cp.b_off += c0 - pc.dc;
            cp.dr = (uint16_t)(r0 - qr); cp.m = (uint16_t)(r1 - r0);
            cp.dc = (uint16_t)(c0 - qc); cp.n = (uint16_t)(c1 - c0);
            O.pieces.push_back(cp);
            wq += double(cp.m) * cp.n * cp.k;
            neg = neg || (cp.flags & 16);
          }
          const size_t np = O.pieces.size() - sp0;
          if (np == 0) continue;
          if (O.pieces.size() > 0x7fffffffULL) { O.err = PASTIX_AMD_ERR_UNSUPPORTED; return; }
          Task tq = tk;
          tq.c_off = tk.c_off + qr + (int64_t)qc * tk.ldc;
          tq.tm = (uint16_t)std::min<int>(64, (int)tk.tm - qr);
          tq.tn = (uint16_t)std::min<int>(64, (int)tk.tn - qc);
          tq.p0 = (int32_t)sp0;
          tq.pn = (int32_t)np;
          tq.nfull = 0;
          tq.flags = (tk.flags & ~(8u | 64u)) | (neg ? 8u : 0u) | 32u;
          O.tasks.push_back(tq);
          O.work.push_back(wq + 4096.0 * double(np));
          O.slot.push_back(slot);
          O.urgent.push_back(task_urgent[i]);
          O.dbytes += 16.0 * double(tq.tm) * double(tq.tn);
          made++;
        }
        O.dbytes -= 16.0 * double(tk.tm) * double(tk.tn);
        O.dcnt.emplace_back(slot, made - 1);
        task_slot[i] = -1;                          // the parent is not launched
      }
    };
    {
      auto guarded = [&](int t) { try { cut(t); } catch (const std::bad_alloc&) { qout[(size_t)t].err = PASTIX_AMD_ERR_ALLOC; } };
      std::vector<std::thread> th;
      for (int t = 1; t < qthr; t++) th.emplace_back(guarded, t);
      guarded(0);
      for (auto& x : th) x.join();
    }
    size_t nq = 0;
    for (const QOut& O : qout) { if (O.err) return O.err; nq += O.pieces.size(); }
    if (P.pieces.size() + nq > 0x7fffffffULL) return PASTIX_AMD_ERR_UNSUPPORTED;
    size_t base = P.pieces.size();
    P.pieces.resize(base + nq);
    for (QOut& O : qout) {
      std::copy(O.pieces.begin(), O.pieces.end(), P.pieces.begin() + base);
      for (Task& tq : O.tasks) tq.p0 += (int32_t)base;
      base += O.pieces.size();
      P.tasks.insert(P.tasks.end(), O.tasks.begin(), O.tasks.end());
      task_work.insert(task_work.end(), O.work.begin(), O.work.end());
      task_slot.insert(task_slot.end(), O.slot.begin(), O.slot.end());
      task_urgent.insert(task_urgent.end(), O.urgent.begin(), O.urgent.end());
      task_tile.insert(task_tile.end(), O.tasks.size(), (int64_t)-1);
      for (auto& d : O.dcnt) P.slot_task_ptr[(size_t)d.first + 1] += d.second;
      ubytes += O.dbytes;
      QOut().pieces.swap(O.pieces);
    }
  }
  P.update_bytes = ubytes;
  for (int s = 0; s < NL; s++) P.slot_task_ptr[s + 1] += P.slot_task_ptr[s];
  // Order inside a slot: the urgent tasks (targets of the slot's own level) first, the quadrant tasks at the end of
  // the urgent and of the bulk range, and inside a range the targets of the next level, then the heaviest task first
  // (the hardware dispatches workgroups in that order: shortest tail of the launch).  Measured alternatives (XCD-
  // locality orders that give every L2 8 x 8 blocks of tasks sharing operands) raise the L2 hit rate from 34 % to 62 %
  // and change nothing or lose: the kernel is not bound by operand traffic.
  {
    // bucket by slot (counting sort, keeps the creation order), then every slot's range sorted on its own, slots dealt
    // to the host threads: the comparison is a total order, the result does not depend on the thread count
    std::vector<int64_t> idx((size_t)P.slot_task_ptr[NL]);
    {
      std::vector<int64_t> pos(P.slot_task_ptr.begin(), P.slot_task_ptr.end() - 1);
      for (size_t q = 0; q < P.tasks.size(); q++)
        if (task_slot[q] >= 0) idx[(size_t)pos[(size_t)task_slot[q]]++] = (int64_t)q;
      std::atomic<int> nexts{0};
      auto sort_slots = [&](int) {
        for (;;) {
          const int sl = nexts.fetch_add(1);
          if (sl >= NL) break;
          std::sort(idx.begin() + P.slot_task_ptr[sl], idx.begin() + P.slot_task_ptr[sl + 1], [&](int64_t a, int64_t b) {
            if ((task_urgent[a] == 2) != (task_urgent[b] == 2)) return task_urgent[a] == 2;   // urgent tasks first
            if (((P.tasks[a].flags ^ P.tasks[b].flags) & 32u) != 0) return (P.tasks[a].flags & 32u) == 0;   // quadrant tasks last
            if (task_urgent[a] != task_urgent[b]) return task_urgent[a] > task_urgent[b];     // targets of the next level first
            return task_work[a] != task_work[b] ? task_work[a] > task_work[b] : a < b;
          });
        }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < nthr; t++) th.emplace_back(sort_slots, t);
      sort_slots(0);
      for (auto& x : th) x.join();
    }
    phase("task grouping");
    P.slot_urgent_end.assign(NL, 0);
    P.slot_next_end.assign(NL, 0);
    P.slot_small_begin.assign(NL, 0);
    P.slot_usmall_begin.assign(NL, 0);
    for (int sl = 0; sl < NL; sl++) {
      int64_t q = P.slot_task_ptr[sl];
      while (q < P.slot_task_ptr[sl + 1] && task_urgent[idx[q]] == 2) q++;
      P.slot_urgent_end[sl] = q;
      while (q < P.slot_task_ptr[sl + 1] && task_urgent[idx[q]] == 1) q++;
      P.slot_next_end[sl] = q;
      int64_t qs = P.slot_task_ptr[sl + 1];
      while (qs > P.slot_urgent_end[sl] && (P.tasks[(size_t)idx[qs - 1]].flags & 32u)) qs--;
      P.slot_small_begin[sl] = qs;
      qs = P.slot_urgent_end[sl];
      while (qs > P.slot_task_ptr[sl] && (P.tasks[(size_t)idx[qs - 1]].flags & 32u)) qs--;
      P.slot_usmall_begin[sl] = qs;
    }
    std::vector<Task> sorted(idx.size());
    {
      std::vector<std::thread> th;
      const size_t nq = idx.size(), per = (nq + (size_t)nthr - 1) / (size_t)nthr;
      auto cp = [&](int t) { for (size_t q = (size_t)t * per; q < std::min(nq, ((size_t)t + 1) * per); q++) sorted[q] = P.tasks[idx[q]]; };
      for (int t = 1; t < nthr; t++) th.emplace_back(cp, t);
      cp(0);
      for (auto& x : th) x.join();
    }
    P.tasks.swap(sorted);
    // ---- the run schedule (plan.h RunInfo) -------------------------------------------------------------------------
    // Tickets = the update tasks of slots >= L0 and the panel-solve tasks T(s) of levels >= L0 (one per 128-row tile with
    // off-diagonal rows); the diagonal tasks D(s) are popped by resident workgroups.  The ticket ARRAY is ordered as a merge
    // of the chain A(L0) T(L0) A(L0+1) T(L0+1) ... (A(s): urgent tasks of slot s -- targets of level s, sources of level
    // s-1) and the bulk B(L0).next B(L0).rest B(L0+1).next ... (B(s): sources <= s-1; .next = targets of level s+1) with
    // A(s) behind B(s-1).next (same tiles, older sources) and B(s) behind T(s-1) (it reads level s-1): a topological order,
    // in which the counters and consumer lists are built; at run time the order of execution is the order of readiness.
    P.ntile = ntile;
    P.nplanes = cplx ? 4 : (lu ? 2 : 1);          // target planes are indexed by arena number: L, U (LU), their imaginary parts
    if (P.run_L0 >= 0 && (int64_t)ntile * P.nplanes > 0x7fffffffLL) P.run_L0 = -1;
    if (P.run_L0 >= 0) {
      const int L0 = P.run_L0;
      struct RunT { TrsmTask tt; int32_t tile, dtask; };
      std::vector<int64_t> tptr((size_t)(NL - L0) + 1, 0);
      std::vector<RunT> rt;
      std::vector<int32_t> dtile0;
      P.run_d.clear();
      P.run_gd = 0;
      for (int l = L0; l < NL; l++) {
        tptr[(size_t)(l - L0)] = (int64_t)rt.size();
        P.run_gd = std::max<int32_t>(P.run_gd, (int32_t)(P.lvl_cblk_ptr[l + 1] - P.lvl_cblk_ptr[l]));
        for (int64_t q = P.lvl_cblk_ptr[l]; q < P.lvl_cblk_ptr[l + 1]; q++) {
          const int32_t k = P.lvl_cblk[(size_t)q];
          const PanelTask& pt = P.panel_tasks[(size_t)q];
          const int32_t w = pt.width, st = pt.stride;
          RunD d{};
          d.pt = pt;
          d.t0 = 0;
          d.tn = 0;
          const int32_t di = (int32_t)P.run_d.size();
          P.run_d.push_back(d);
          dtile0.push_back((int32_t)tile_base[(size_t)k]);
          // (complex: 64 rows per panel-solve ticket -- the parked solve keeps half of a wave's tiles in LDS and has
          // room for four waves' worth --, i.e. up to two tickets per 128-row tile, consecutive)
          const int32_t trows = cplx ? 64 : TM;
          for (int32_t r = w / TM; (int64_t)r * TM < st; r++) {
            const int32_t r0 = std::max<int32_t>(w, r * TM), r1 = std::min<int32_t>(st, (r + 1) * TM);
            for (int32_t q0 = r0; q0 < r1; q0 += trows) {
              RunT tt{};
              tt.tt = TrsmTask{pt.off, st, w, q0, std::min(trows, r1 - q0), pt.dinv_off};
              tt.tile = (int32_t)(tile_base[(size_t)k] + r);
              tt.dtask = di;
              rt.push_back(tt);
            }
          }
        }
      }
      tptr[(size_t)(NL - L0)] = (int64_t)rt.size();
      P.run_gd = std::min<int32_t>(P.run_gd, P.opts.run_d_workers > 0 ? P.opts.run_d_workers : 8);
      // the merge.  order[]: >= 0 an update task (index into the slot-ordered task list), < 0 panel-solve task -1 - i
      std::vector<int64_t> order;
      order.reserve((size_t)(P.slot_task_ptr[NL] - P.slot_task_ptr[L0]) + rt.size());
      {
        int cs = L0, ck = 0;                   // chain head: A(cs) (ck = 0) or T(cs) (ck = 1); cs == NL: exhausted
        int bs = L0;                           // bulk head: task bq of slot bs
        int64_t bq = P.slot_urgent_end[L0];
        auto bulk_skip = [&]() { while (bs < NL && bq >= P.slot_task_ptr[bs + 1]) { bs++; if (bs < NL) bq = P.slot_urgent_end[bs]; } };
        bulk_skip();
        for (;;) {
          const bool chain_left = cs < NL;
          // A(cs) may go once B(cs-1).next is out: the bulk head is past it
          const bool chain_ok = chain_left && (ck == 1 || cs == L0 || bs > cs - 1 || (bs == cs - 1 && bq >= P.slot_next_end[cs - 1]));
          // a bulk task of slot bs may go once T(bs-1) is out: the chain head is past it
          const bool bulk_ok = bs < NL && (bs == L0 || cs > bs - 1);
          if (!chain_left && bs >= NL) break;
          if (chain_ok) {
            if (ck == 0) {
              for (int64_t q = P.slot_task_ptr[cs]; q < P.slot_urgent_end[cs]; q++) order.push_back(q);
              ck = 1;
            } else {
              for (int64_t i2 = tptr[(size_t)(cs - L0)]; i2 < tptr[(size_t)(cs - L0) + 1]; i2++) order.push_back(-1 - i2);
              ck = 0;
              cs++;
            }
          } else if (bulk_ok) {
            order.push_back(bq++);
            bulk_skip();
          } else {
            return PASTIX_AMD_ERR_LAYOUT;       // (cannot happen: one of the two heads is always free to go)
          }
        }
      }
      phase("run: ticket order");
      const size_t nr = order.size();
      const size_t nd = P.run_d.size();
      if (nr + nd > 0x7ffffff0ULL) return PASTIX_AMD_ERR_UNSUPPORTED;
      auto slot_of = [&](int64_t q) { return (int)(std::upper_bound(P.slot_task_ptr.begin(), P.slot_task_ptr.end(), q) - P.slot_task_ptr.begin() - 1); };
      if (dev_opt("run_prof")) {
        // developer aid (tools/run_prof.py): category (0 A, 1 B.next, 2 B.rest, 3 T) and slot / level of every ticket
        P.run_cat.resize(nr);
        P.run_lvl.resize(nr);
        for (size_t i = 0; i < nr; i++) {
          const int64_t q = order[i];
          if (q < 0) {
            P.run_cat[i] = 3;
            P.run_lvl[i] = (int)(std::upper_bound(tptr.begin(), tptr.end(), -1 - q) - tptr.begin() - 1) + L0;
          } else {
            const int sl = slot_of(q);
            P.run_cat[i] = q < P.slot_urgent_end[sl] ? 0 : q < P.slot_next_end[sl] ? 1 : 2;
            P.run_lvl[i] = sl;
          }
        }
      }
      P.run_tasks.resize(nr);
      P.run_info.assign(nr, RunInfo{-1, 0, 0, 0});
      P.run_chk.assign(nr, RunCheck{0, 0, 0, 0});
      P.run_dep.assign(nr + nd, 0);
      std::vector<int32_t> tcount((size_t)ntile * (size_t)P.nplanes, 0);      // update tickets so far per tile counter
      std::vector<int32_t> last((size_t)ntile * (size_t)P.nplanes, -1);       // ... and the last of them
      std::vector<int32_t> tile_ticket((size_t)ntile, -1);                    // the first panel-solve ticket of a tile ...
      std::vector<uint8_t> tile_nt((size_t)ntile, 0);                         // ... and how many it has (consecutive)
      bool bad = false;
      for (size_t i = 0; i < nr && !bad; i++) {
        const int64_t q = order[i];
        RunInfo& ri = P.run_info[i];
        RunCheck& ck = P.run_chk[i];
        if (q >= 0) {
          P.run_tasks[i] = P.tasks[(size_t)q];
          const int64_t tl = task_tile[(size_t)idx[(size_t)q]];
          if (tl < 0 || (P.tasks[(size_t)q].flags & (4u | 32u))) { bad = true; break; }   // (quadrant / shared tasks: not in a run)
          const int sl = slot_of(q);
          ri.kind = q < P.slot_urgent_end[sl] ? 0 : q < P.slot_next_end[sl] ? 1 : 2;
          ck.tile = (int32_t)tl;
          ck.seq = tcount[(size_t)tl]++;
          if (last[(size_t)tl] >= 0) { P.run_info[(size_t)last[(size_t)tl]].succ = (int32_t)i; P.run_info[(size_t)last[(size_t)tl]].cn = 1; P.run_dep[i]++; }
          last[(size_t)tl] = (int32_t)i;
        } else {
          // the panel-solve ticket of a tile: every update ticket of the run on that tile precedes it (A(s) is the last
          // slot that targets level s).  It waits for the diagonal task and, unless the tile is the diagonal tile
          // (whose updates the diagonal task has waited for), for the tile's last update
          const RunT& tt = rt[(size_t)(-1 - q)];
          static_assert(sizeof(Task) == sizeof(TrsmTask), "a panel-solve ticket travels in a Task record");
          memcpy(&P.run_tasks[i], &tt.tt, sizeof(Task));
          ri.kind = 0 | 4;
          ck.tile = tt.tile;
          ck.seq = tcount[(size_t)tt.tile];
          ck.wptr = tt.dtask;
          ck.wn = -1;
          P.run_dep[i] = 1;
          RunD& d = P.run_d[(size_t)tt.dtask];
          if (d.tn == 0) d.t0 = (int32_t)i;
          if (d.t0 + d.tn != (int32_t)i) { bad = true; break; }       // (the tickets of a cblk are consecutive)
          d.tn++;
          if (tt.tile != dtile0[(size_t)tt.dtask])
            for (int pl = 0; pl < P.nplanes; pl++) {
              const int32_t lu2 = last[(size_t)tt.tile + (size_t)pl * (size_t)ntile];
              if (lu2 < 0) continue;
              RunInfo& pu = P.run_info[(size_t)lu2];             // the tile's last update on this plane: its successors are
              if (tile_nt[(size_t)tt.tile] == 0) { pu.succ = (int32_t)i; pu.cn = 1; }   // the tile's panel-solve tickets
              else pu.cn++;
              P.run_dep[i]++;
            }
          if (tile_nt[(size_t)tt.tile] == 0) tile_ticket[(size_t)tt.tile] = (int32_t)i;
          else if (tile_ticket[(size_t)tt.tile] + tile_nt[(size_t)tt.tile] != (int32_t)i) { bad = true; break; }
          tile_nt[(size_t)tt.tile]++;
        }
      }
      phase("run: tile chains");
      if (bad) {
        P.run_L0 = -1;
        P.run_tasks.clear();
        P.run_info.clear();
        P.run_chk.clear();
        P.run_dep.clear();
        P.run_d.clear();
      } else {
        // a diagonal task waits for the last update of the diagonal tile
        P.run_dchk.resize(nd);
        for (size_t d = 0; d < nd; d++) {
          for (int pl = 0; pl < P.nplanes; pl++) {       // (every plane of the diagonal tile that the run updates)
            const int32_t l0 = last[(size_t)dtile0[d] + (size_t)pl * (size_t)ntile];
            if (l0 >= 0) { P.run_info[(size_t)l0].succ = -2 - (int32_t)d; P.run_info[(size_t)l0].cn = 1; P.run_dep[nr + d]++; }
          }
          P.run_dchk[d] = {dtile0[d], 0};
        }
        P.run_edges_deferred = P.defer_run_edges;
        if (P.run_edges_deferred) {
          // the reader lists are built on the device from the uploaded tables (run_edges.hip, api.cpp): what it needs
          P.run_tile_ticket.assign(tile_ticket.begin(), tile_ticket.end());
          P.run_tile_base.resize((size_t)nc + 1);
          for (int64_t k = 0; k <= nc; k++) P.run_tile_base[(size_t)k] = (int32_t)tile_base[(size_t)k];
          P.run_flops = 0;
          for (int sl = L0; sl < NL; sl++) P.run_flops += P.slot_flops[(size_t)sl];
          P.run_waits.clear();
          P.run_cons.clear();
          phase("run: tables for the device");
        } else {
        // source tiles an update ticket reads: the 128-row tiles of the source panels of its pieces (A rows, B rows),
        // sources of the run's levels only -- older panels are final when the run starts
        std::vector<std::vector<int32_t>> tw((size_t)nthr);
        std::vector<std::vector<std::pair<int32_t, int32_t>>> tpw((size_t)nthr);   // per ticket of the range: (first, count) in tw
        std::vector<double> trf((size_t)nthr, 0.0);
        const size_t per = (nr + (size_t)nthr - 1) / (size_t)nthr;
        auto wbody = [&](int t) {
          std::vector<int32_t>& W = tw[(size_t)t];
          std::vector<int32_t> tmp;
          double fl = 0;
          for (size_t i = (size_t)t * per; i < std::min(nr, ((size_t)t + 1) * per); i++) {
            tmp.clear();
            if (!(P.run_info[i].kind & 4)) {
              const Task& tk = P.run_tasks[i];
              for (int z = 0; z < tk.pn; z++) {
                const Piece& pc = P.pieces[(size_t)tk.p0 + (size_t)z];
                fl += 2.0 * pc.m * (double)pc.n * pc.k;
                const int64_t k = std::upper_bound(P.poff.begin(), P.poff.end(), pc.a_off) - P.poff.begin() - 1;
                if (P.level[(size_t)k] < L0) continue;
                const int64_t a0 = pc.a_off - P.poff[(size_t)k], b0 = pc.b_off - P.poff[(size_t)k];   // rows (column 0 of the panel)
                for (int64_t r = a0 / TM; r <= (a0 + pc.m - 1) / TM; r++) tmp.push_back((int32_t)(tile_base[(size_t)k] + r));
                for (int64_t r = b0 / TM; r <= (b0 + pc.n - 1) / TM; r++) tmp.push_back((int32_t)(tile_base[(size_t)k] + r));
              }
              std::sort(tmp.begin(), tmp.end());
              tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            }
            tpw[(size_t)t].emplace_back((int32_t)W.size(), (int32_t)tmp.size());
            W.insert(W.end(), tmp.begin(), tmp.end());
          }
          trf[(size_t)t] = fl;
        };
        {
          std::vector<std::thread> th;
          for (int t = 1; t < nthr; t++) th.emplace_back(wbody, t);
          wbody(0);
          for (auto& x : th) x.join();
        }
        size_t tot = 0;
        for (int t = 0; t < nthr; t++) tot += tw[(size_t)t].size();
        if (tot > 0x7fffffffULL) return PASTIX_AMD_ERR_UNSUPPORTED;
        P.run_waits.resize(tot);
        size_t base = 0;
        P.run_flops = 0;
        for (int t = 0; t < nthr; t++) {
          std::copy(tw[(size_t)t].begin(), tw[(size_t)t].end(), P.run_waits.begin() + base);
          for (size_t j2 = 0; j2 < tpw[(size_t)t].size(); j2++) {
            RunCheck& ck = P.run_chk[(size_t)t * per + j2];
            if (ck.wn < 0) continue;
            ck.wptr = (int32_t)(base + (size_t)tpw[(size_t)t][j2].first);
            ck.wn = tpw[(size_t)t][j2].second;
          }
          base += tw[(size_t)t].size();
          P.run_flops += trf[(size_t)t];
        }
        phase("run: source tiles");
        // consumer lists of the panel-solve tickets (CSR by producer) and the counters of the readers.  On the host
        // threads: counts and cursors with atomic adds, every list sorted afterwards -- the tables do not depend on the
        // number of threads.
        {
          std::atomic<int> abad{0};
          auto par = [&](auto&& fn) {
            std::vector<std::thread> th;
            for (int t = 1; t < nthr; t++) th.emplace_back(fn, t);
            fn(0);
            for (auto& x : th) x.join();
          };
          par([&](int t) {
            for (size_t i = (size_t)t * per; i < std::min(nr, ((size_t)t + 1) * per); i++) {
              const RunCheck& ck = P.run_chk[i];
              if (ck.wn < 0) continue;
              for (int q = 0; q < ck.wn; q++) {
                const int32_t tl2 = P.run_waits[(size_t)ck.wptr + (size_t)q];
                const int32_t pt2 = tile_ticket[(size_t)tl2];
                if (pt2 < 0 || pt2 >= (int32_t)i) { abad = 1; break; }   // (a source tile of the run has its tickets, in front of its readers)
                for (int z = 0; z < tile_nt[(size_t)tl2]; z++) {
                  __atomic_fetch_add(&P.run_info[(size_t)pt2 + (size_t)z].cn, 1, __ATOMIC_RELAXED);
                  P.run_dep[i]++;
                }
              }
            }
          });
          if (abad) return PASTIX_AMD_ERR_LAYOUT;
          int64_t off = 0;
          for (size_t i = 0; i < nr; i++)
            if (P.run_info[i].kind & 4) { P.run_info[i].cptr = (int32_t)off; off += P.run_info[i].cn; P.run_info[i].cn = 0; }
          if (off > 0x7fffffffLL) return PASTIX_AMD_ERR_UNSUPPORTED;
          P.run_cons.resize((size_t)off);
          par([&](int t) {
            for (size_t i = (size_t)t * per; i < std::min(nr, ((size_t)t + 1) * per); i++) {
              const RunCheck& ck = P.run_chk[i];
              if (ck.wn < 0) continue;
              for (int q = 0; q < ck.wn; q++) {
                const int32_t tl2 = P.run_waits[(size_t)ck.wptr + (size_t)q];
                for (int z = 0; z < tile_nt[(size_t)tl2]; z++) {
                  RunInfo& pi = P.run_info[(size_t)tile_ticket[(size_t)tl2] + (size_t)z];
                  P.run_cons[(size_t)pi.cptr + (size_t)__atomic_fetch_add(&pi.cn, 1, __ATOMIC_RELAXED)] = (int32_t)i;
                }
              }
            }
          });
          par([&](int t) {
            for (size_t i = (size_t)t * per; i < std::min(nr, ((size_t)t + 1) * per); i++) {
              const RunInfo& ri = P.run_info[i];
              if ((ri.kind & 4) && ri.cn > 1) std::sort(P.run_cons.begin() + ri.cptr, P.run_cons.begin() + ri.cptr + ri.cn);
            }
          });
        }
        phase("run: consumer lists");
        }
        P.run_tile_nt.assign(tile_nt.begin(), tile_nt.end());
        // what is ready when the run starts, in ticket order (deferred reader lists: the caller does this once the device
        // has added the source inputs to the counters)
        P.run_ready.clear();
        P.run_dready.clear();
        if (!P.run_edges_deferred)
          for (size_t i = 0; i < nr; i++) if (P.run_dep[i] == 0) P.run_ready.push_back((int32_t)i);
        for (size_t d = 0; d < nd; d++) if (P.run_dep[nr + d] == 0) P.run_dready.push_back((int32_t)d);
      }
    }
  }
  phase("task ordering");
  if (P.opts.verbose >= 3) {
    // developer aid: the pieces of the tasks that run the masked loop (any partial piece), by extent in 16-row / 16-column
    // bands: share of those tasks' chunk iterations and of their flops
    double it[9][9] = {}, fl[9][9] = {}, tit = 0, tfl = 0, allfl = 0;
    for (const Task& t : P.tasks) {
      const bool masked = (int)t.nfull != t.pn && !(t.flags & 32u);
      for (int i = 0; i < t.pn; i++) {
        const Piece& pc = P.pieces[(size_t)t.p0 + i];
        const double f = 2.0 * pc.m * (double)pc.n * pc.k;
        allfl += f;
        if (!masked) continue;
        const int mb = (pc.m + 15) / 16, nb = (pc.n + 15) / 16;
        const double c = (pc.k + 15) / 16;
        it[mb][nb] += c; fl[mb][nb] += f; tit += c; tfl += f;
      }
    }
    fprintf(stderr, "[plan] masked-loop tasks: %.2f %% of the update flops; rows: m in 16-row bands, columns: n; %% of their chunk iterations / %% of their flops\n", 100.0 * tfl / std::max(allfl, 1.0));
    for (int a = 1; a <= 8; a++) {
      fprintf(stderr, "[plan]  m<=%3d:", 16 * a);
      for (int b = 1; b <= 8; b++) fprintf(stderr, " %5.1f/%-5.1f", 100.0 * it[a][b] / std::max(tit, 1.0), 100.0 * fl[a][b] / std::max(tfl, 1.0));
      fprintf(stderr, "\n");
    }
  }
  if (ptime) {
    // (a fingerprint of the schedule: equal for equal inputs whatever the number of host threads)
    auto mix = [](uint64_t h, const void* data, size_t bytes) {
      const uint64_t* w = (const uint64_t*)data;
      for (size_t i = 0; i < bytes / 8; i++) { h ^= w[i]; h *= 0x100000001b3ULL; h ^= h >> 29; }
      return h;
    };
    uint64_t h = mix(0xcbf29ce484222325ULL, P.tasks.data(), P.tasks.size() * sizeof(Task));
    h = mix(h, P.pieces.data(), P.pieces.size() * sizeof(Piece));
    fprintf(stderr, "[plan] tasks %zu pieces %zu fingerprint %016llx\n", P.tasks.size(), P.pieces.size(), (unsigned long long)h);
    phase("(fingerprint: only with plan_timing)");
  }
  // (the sorted raw piece list -- 5.6 GB at 200^3 -- and the other big temporaries are unmapped on a thread of their own:
  // returning them to the system is 0.2-0.4 s that nothing has to wait for)
  {
    struct Junk { decltype(raw) a; };
    Junk* j = new (std::nothrow) Junk{std::move(raw)};
    if (j) std::thread([j] { delete j; }).detach();
  }
  return PASTIX_AMD_OK;
}

// Host-only check of a run schedule (tests): replays the counter protocol with ONE worker -- pop the next ready ticket,
// diagonal tasks whenever one is ready, decrement the consumers, push what reaches zero -- and checks at
// every pop what the task must find (its tile written exactly `seq` times, its source tiles solved, its cblk's diagonal
// blok factorized).  Returns 0 when every ticket and diagonal task ran and found that; else the 1-based index of the first
// ticket that found something else, -1 for an inconsistent table, -2 when tasks were left (a cycle or a lost decrement).
int64_t run_verify(const Plan& P) {
  if (P.run_L0 < 0) return 0;
  const size_t nr = P.run_tasks.size(), nd = P.run_d.size();
  if (P.run_info.size() != nr || P.run_chk.size() != nr || P.run_dep.size() != nr + nd || P.run_tile_nt.size() != (size_t)P.ntile) return -1;
  std::vector<int32_t> cnt(P.run_dep);
  if (P.ntile <= 0) return -1;
  std::vector<int32_t> seq((size_t)P.ntile * (size_t)P.nplanes, 0), fin((size_t)P.ntile, 0), dfl(std::max<size_t>(nd, 1), 0);
  std::vector<int32_t> total(seq.size(), 0);           // update tickets of the run per tile counter
  for (size_t i = 0; i < nr; i++)
    if (!(P.run_info[i].kind & 4)) {
      if (P.run_chk[i].tile < 0 || (size_t)P.run_chk[i].tile >= total.size()) return -1;
      total[(size_t)P.run_chk[i].tile]++;
    }
  auto tile_complete = [&](int32_t tile) {
    for (int pl = 0; pl < P.nplanes; pl++)
      if (seq[(size_t)tile + (size_t)pl * (size_t)P.ntile] != total[(size_t)tile + (size_t)pl * (size_t)P.ntile]) return false;
    return true;
  };
  std::vector<int32_t> q(P.run_ready), qd(P.run_dready);
  size_t head = 0, hd = 0, done = 0, doned = 0;
  auto dec_ticket = [&](int32_t c) {
    if (c < 0 || (size_t)c >= nr || cnt[(size_t)c] <= 0) return false;
    if (--cnt[(size_t)c] == 0) q.push_back(c);
    return true;
  };
  for (;;) {
    while (hd < qd.size()) {                         // diagonal tasks first (resident workers)
      const int32_t d = qd[hd++];
      if (d < 0 || (size_t)d >= nd || dfl[(size_t)d] || P.run_dchk.size() != nd) return -1;
      if (!tile_complete(P.run_dchk[(size_t)d].first)) return -3 - (int64_t)d;
      dfl[(size_t)d] = 1;
      doned++;
      const RunD& rd = P.run_d[(size_t)d];
      for (int32_t t = rd.t0; t < rd.t0 + rd.tn; t++) if (!dec_ticket(t)) return -1;
    }
    if (head >= q.size()) break;
    const int32_t i = q[head++];
    const RunInfo& ri = P.run_info[(size_t)i];
    const RunCheck& ck = P.run_chk[(size_t)i];
    done++;
    if (ri.kind & 4) {
      if (ck.tile < 0 || (size_t)ck.tile >= fin.size() || ck.wptr < 0 || (size_t)ck.wptr >= nd) return -1;
      if (!dfl[(size_t)ck.wptr] || !tile_complete(ck.tile) || fin[(size_t)ck.tile] >= P.run_tile_nt[(size_t)ck.tile]) return (int64_t)i + 1;
      fin[(size_t)ck.tile]++;
      if (ri.cptr < 0 || (size_t)ri.cptr + (size_t)ri.cn > P.run_cons.size()) return -1;
      for (int z = 0; z < ri.cn; z++) if (!dec_ticket(P.run_cons[(size_t)ri.cptr + (size_t)z])) return -1;
    } else {
      if (ck.tile < 0 || (size_t)ck.tile >= seq.size() || ck.wptr < 0 || (size_t)ck.wptr + (size_t)ck.wn > P.run_waits.size()) return -1;
      if (seq[(size_t)ck.tile] != ck.seq) return (int64_t)i + 1;
      if (fin[(size_t)ck.tile % (size_t)P.ntile]) return (int64_t)i + 1;                   // (written after it was solved)
      for (int z = 0; z < ck.wn; z++) {
        const int32_t f = P.run_waits[(size_t)ck.wptr + (size_t)z];
        if (f < 0 || (size_t)f >= fin.size()) return -1;
        if (fin[(size_t)f] != P.run_tile_nt[(size_t)f] || !fin[(size_t)f]) return (int64_t)i + 1;
      }
      seq[(size_t)ck.tile] = ck.seq + 1;
      if (ri.succ >= 0) { for (int z = 0; z < std::max(ri.cn, 1); z++) if (!dec_ticket(ri.succ + z)) return -1; }
      else if (ri.succ <= -2) {
        const size_t d = (size_t)(-2 - ri.succ);
        if (d >= nd || cnt[nr + d] <= 0) return -1;
        if (--cnt[nr + d] == 0) qd.push_back((int32_t)d);
      }
    }
  }
  return (done == nr && doned == nd) ? 0 : -2;
}

// Host-only check of the update schedule against the reference's definition (compute_1dgemm, sopalin_compute.c:865-1032):
// for every source cblk k, every off-diagonal blok i and every blok j >= i the product (rows of j) x (rows of i)^T is
// subtracted ONCE from the entries of the facing panel where add_contrib_local puts it (:427-429).  Decodes every piece of
// the plan -- rectangles and gathered pieces -- into (target entry, source row of A, source row of B) triples and compares
// the sorted list with the one made from the layout.  Real LLt / LDLt (one plane, one product per pair).
// out[0] = products expected, out[1] = products the pieces make, out[2] = mismatching entries, out[3] = gathered pieces.
int verify_pieces(const Plan& P, int64_t out[4]) {
  struct T3 { int64_t c, a, b; bool operator<(const T3& o) const { return c != o.c ? c < o.c : a != o.a ? a < o.a : b < o.b; }
              bool operator==(const T3& o) const { return c == o.c && a == o.a && b == o.b; } };
  std::vector<T3> want, have;
  const int64_t nc = P.cblknbr;
  for (int64_t k = 0; k < nc; k++) {
    if (P.role[k] != 1) continue;
    const int64_t fb = P.cblk[k].bloknum, lb = P.cblk[k + 1].bloknum;
    for (int64_t i = fb + 1; i < lb; i++) {
      const int64_t t = P.blok[i].cblknum, tf = P.cblk[t].fcolnum;
      int64_t b3 = P.cblk[t].bloknum;
      for (int64_t j = i; j < lb; j++) {
        while (!(P.blok[j].frownum >= P.blok[b3].frownum && P.blok[j].lrownum <= P.blok[b3].lrownum)) b3++;
        const int64_t dst = P.tcoef[b3] + (P.blok[j].frownum - P.blok[b3].frownum);
        for (int64_t r = 0; r <= P.blok[j].lrownum - P.blok[j].frownum; r++)
          for (int64_t c = 0; c <= P.blok[i].lrownum - P.blok[i].frownum; c++)
            want.push_back(T3{P.poff[t] + dst + r + (P.blok[i].frownum - tf + c) * P.tstride[t],
                              P.poff[k] + P.blok[j].coefind + r, P.poff[k] + P.blok[i].coefind + c});
      }
    }
  }
  int64_t ngather = 0;
  for (size_t ti = 0; ti < P.tasks.size(); ti++) {
    const Task& tk = P.tasks[ti];
    for (int z = 0; z < tk.pn; z++) {
      const Piece& pc = P.pieces[(size_t)tk.p0 + (size_t)z];
      if (pc.flags & PIECE_GATHERED) {
        ngather++;
        const uint32_t* w = P.gmaps.data() + ((size_t)pc.dr | ((size_t)pc.dc << 16)) * 64;
        for (int r = 0; r < 128; r++) {
          const uint32_t sa = (w[r & 31] >> (8 * (r >> 5))) & 255u;
          if (sa == 255u) continue;
          for (int c = 0; c < 128; c++) {
            const uint32_t sb = (w[32 + (c & 31)] >> (8 * (c >> 5))) & 255u;
            if (sb == 255u) continue;
            have.push_back(T3{tk.c_off + r + (int64_t)c * tk.ldc, pc.a_off + sa, pc.b_off + sb});
          }
        }
      } else {
        for (int r = 0; r < pc.m; r++)
          for (int c = 0; c < pc.n; c++)
            have.push_back(T3{tk.c_off + pc.dr + r + (int64_t)(pc.dc + c) * tk.ldc, pc.a_off + r, pc.b_off + c});
      }
    }
  }
  std::sort(want.begin(), want.end());
  std::sort(have.begin(), have.end());
  int64_t bad = 0;
  size_t x = 0, y = 0;
  while (x < want.size() || y < have.size()) {
    if (x < want.size() && y < have.size() && want[x] == have[y]) { x++; y++; }
    else if (y >= have.size() || (x < want.size() && want[x] < have[y])) { bad++; x++; }
    else { bad++; y++; }
  }
  out[0] = (int64_t)want.size();
  out[1] = (int64_t)have.size();
  out[2] = bad;
  out[3] = ngather;
  return 0;
}

}  // namespace pastix_amd
