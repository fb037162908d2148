// engine.h -- internal: the plan object behind pastix_amd_plan_t and the kernel launchers (kernels*.hip) shared by the
// translation units of the engine (api.cpp: life cycle, transfers, single-GPU driver, solves; dist.cpp: the
// multi-GPU fan-in driver).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "plan.h"

namespace pastix_amd {
// run_edges.hip: the reader lists of the run schedule built on the device (plan.h Plan::run_edges_deferred)
size_t run_edges_device_bytes(const Plan& H, size_t nr, size_t npieces_run);
int run_edges_device(hipStream_t s, const Task* dRunTasks, RunInfo* dRunInfo, const Piece* dPieces, Plan& H, int32_t** cons_out,
                     size_t* ncons_out);
void launch_ring_scatter(hipStream_t s, int32_t* ring, const int32_t* vals, size_t n);   // ring[i * RUN_SLOT] = vals[i]
void launch_update(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks,
                   bool urgent);
// the quadrant tasks of a slot (Task flag 32, kernels_small.hip)
void launch_update_small(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks,
                         bool urgent);
// the run schedule (plan.h RunInfo): the update tasks of the thin levels in one launch, their panel tasks on resident
// workgroups of two kernels on streams of their own
void launch_run_update(hipStream_t s, int factotype, const Arenas& ar, const Task* tasks, const Piece* pieces, const RunInfo* info,
                       const int32_t* cons, const RunCtl& rc, double* dinv, int64_t ntasks, int nwg, long long limit, const RunD* rd,
                       double critere, long long* nbpivot, int* errflag);
void launch_run_panel(hipStream_t sd, int factotype, const Arenas& ar, const RunD* rd, const RunInfo* info, int gd, double* dinv,
                      double critere, long long* nbpivot, int* errflag, const RunCtl& rc, int* resident, long long limit);
void launch_diag_zsy(hipStream_t s, bool herm, const Arenas& ar, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                     long long* nbpivot, int maxw);
void launch_zsolve_level(hipStream_t s, bool fwd, int factotype, const Arenas& ar, const SolveTask* tasks, int64_t ntask,
                         const SolveChunk* chunks, int64_t nchunk, const DevBlok* bl, const int32_t* ridx, double* xr,
                         double* xi, int maxw);
void launch_zsolve_dscale(hipStream_t s, const Arenas& ar, const SolveTask* tasks, int64_t ntask, double* xr, double* xi);
void launch_fanin_add(hipStream_t s, double* dst, int64_t ldd, const double* src, const int32_t* rows, int64_t nrows,
                      int64_t ncols);
void launch_diag_zlu(hipStream_t s, const Arenas& ar, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                     long long* nbpivot, int maxw);
void launch_trsm_zlu(hipStream_t s, const Arenas& ar, const TrsmTask* tasks, int64_t n, const double* dinv, int maxw);
void launch_trsm_zsy(hipStream_t s, bool herm, const Arenas& ar, const TrsmTask* tasks, int64_t n, const double* dinv,
                     int maxw);
void launch_split(hipStream_t s, const double* z, double* re, double* im, int64_t n);
void launch_merge(hipStream_t s, double* z, const double* re, const double* im, int64_t n);
void launch_diag_llt(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                     long long* nbpivot, int* errflag, int maxw);
void launch_trsm_llt(hipStream_t s, double* L, const TrsmTask* tasks, int64_t n, const double* dinv, int maxw);
void launch_fill_const(hipStream_t s, double* dst, int64_t n, double v);
void launch_scatter(hipStream_t s, double* dst, const int64_t* idx, const double* val, int64_t n);
void launch_diag_ldlt(hipStream_t s, double* L, const PanelTask* tasks, int64_t n, double* dinv, double critere,
                      long long* nbpivot, int maxw);
void launch_diag_lu(hipStream_t s, double* L, double* U, const PanelTask* tasks, int64_t n, double* dinv,
                    double critere, long long* nbpivot);
void launch_trsm_ldlt(hipStream_t s, double* L, double* U, const TrsmTask* tasks, int64_t n, const double* dinv,
                      int maxw);
void launch_trsm_lu(hipStream_t s, double* L, double* U, const TrsmTask* tasks, int64_t n, const double* dinv,
                    int maxw);
void launch_solve_level(hipStream_t s, bool fwd, int factotype, const double* L, const double* U,
                        const SolveTask* tasks, int64_t ntask, int64_t nwide, const SolveChunk* chunks, int64_t nchunk,
                        const DevBlok* bl, const int32_t* ridx, double* x, int64_t ldx, int nr, int maxw, int lvlw);
// kernels_f32.hip: the single-precision engine (arenas of floats; Arenas::p reinterpreted)
void launch_update_s(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks, bool urgent);
void launch_diag_s(hipStream_t s, int factotype, float* L, float* U, const PanelTask* tasks, int64_t n, float* dinv,
                   double critere, long long* nbpivot, int* errflag, int maxw);
void launch_trsm_s(hipStream_t s, int factotype, float* L, float* U, const TrsmTask* tasks, int64_t n, const float* dinv,
                   int maxw);
void launch_fill_const_s(hipStream_t s, float* dst, int64_t n, float v);
void launch_scatter_s(hipStream_t s, float* dst, const int64_t* idx, const double* val, int64_t n);
void launch_solve_inv(hipStream_t s, const void* A, bool f32, const SolveTask* tasks, const int32_t* thin_tasks, int64_t n,
                      double* inv, int which, int unit);
void launch_solve_thin(hipStream_t s, bool fwd, const void* P, bool f32, const SolveChunk* chunks, int64_t nchunk,
                       const int32_t* ridx, const double* inv, int* ticket, const int32_t* tgt, const int32_t* expect, int* cnt,
                       int* flag, int* stuck, double* x);
void launch_solve_level_s(hipStream_t s, bool fwd, int factotype, const float* L, const float* U, const SolveTask* tasks,
                          int64_t ntask, int64_t nwide, const SolveChunk* chunks, int64_t nchunk, const int32_t* ridx, double* x,
                          int lvlw);
void launch_solve_dscale_s(hipStream_t s, const float* L, const SolveTask* tasks, int64_t ntask, double* x);
void launch_solve_rowidx(hipStream_t s, const SolveTask* tasks, int64_t ntask, const int64_t* roff,
                         const DevBlok* bl, int32_t* ridx);
void launch_solve_dscale(hipStream_t s, const double* L, const SolveTask* tasks, int64_t ntask, double* x);
}  // namespace pastix_amd

using namespace pastix_amd;

static constexpr size_t ARENA_PAD = 256;

// see build_split below
struct SplitMap {
  bool active = false;
  int64_t ocblknbr = 0;
  std::vector<int64_t> first;                    // [ocblknbr+1] index of the first sub-cblk of an original cblk
  std::vector<int64_t> owidth, ostride, ooff;    // original width, stride, offset in the packed original arena
  std::vector<pastix_amd_cblk_t> cblk;           // the split layout
  std::vector<pastix_amd_blok_t> blok;
  // LLt / LDLt: the blocks of a re-cut cblk's diagonal blok ABOVE its column groups are not part of the factor and not on
  // the device; the reference's buffers carry the input values there from fill to return (compute_diag.c:124-203 never
  // touches the strict upper triangle), so they are kept here: per re-cut cblk the diagonal blok as a dense ow x ow
  // array (ld ow; complex: interleaved), of which only those blocks are used -- recorded by the fills / uploads, written
  // back by the downloads.
  std::vector<std::vector<unsigned char>> upper; // [ocblknbr] (entries of the plan's type), empty for cblks not re-cut
};

struct pastix_amd_dist_s;            // dist.cpp: fan-in schedule, channels, transport

struct pastix_amd_plan_s {
  Plan host;
  pastix_amd_dist_s* dist = nullptr;
  void (*dist_free)(pastix_amd_dist_s*) = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // second stream: non-urgent contributions overlap the panel kernels
  double fillBaseL = 0.0, fillBaseU = 0.0;   // value every panel entry starts from in pastix_amd_refill (0; 1 / 2 after
                                             // pastix_amd_fill_fake)
  bool launch_events = true;          // per-launch timing events recorded (api.cpp, mode-1 driver)
  std::vector<hipEvent_t> evP, evB;   // per level: panels done (stream), bulk contributions done (stream2)
  std::vector<hipEvent_t> evT;        // timing pairs of the bulk launches
  int nupdB_run = 0;
  bool own_stream = true, own_arena = true, distributed = false, overlapped = false;
  int overlap_mode = 0;
  bool staged_overlap = false;        // two streams behind the level-stepped API (distributed plans)
  int staged_lastB = -1;
  int nupd_run = 0;
  double crit_run = 0;
  double* dL = nullptr;      // L  (real part)
  double* dU = nullptr;      // U / L*D (real part)
  double* dLi = nullptr;     // imaginary planes (complex double only)
  double* dUi = nullptr;
  bool cplx = false;
  bool f32 = false;          // single precision (PASTIX_AMD_REALSINGLE): dL / dU are arenas of FLOATS (kernels_f32.hip)
  size_t esz = sizeof(double);   // bytes per arena entry
  // arena + element offset, whatever the entry size (the pointers are typed double* for the fp64 kernels)
  double* at(double* arena, int64_t off) const { return (double*)((char*)arena + (size_t)off * esz); }
  Arenas arenas() const { return Arenas{{dL, dU, dLi, dUi}, dGmap}; }
  uint32_t* dGmap = nullptr; // row maps of the gathered pieces
  double* dDinv = nullptr;
  Task* dTasks = nullptr;
  Piece* dPieces = nullptr;
  PanelTask* dPanel = nullptr;
  TrsmTask* dTrsm = nullptr;
  long long* dNbpivot = nullptr;
  int* dErr = nullptr;
  int maxw = 0;
  double* dXws = nullptr;              // solve workspace (right-hand sides on the device), kept between calls
  size_t nXws = 0;
  SplitMap split;                      // cblks wider than MAXW are factorized in column groups (build_split)
  bool factored = false;               // panels hold factors (set by factorize, cleared by upload / fill)
  // cached coefficient fill (destinations + values) so that a re-fill is device-only
  int64_t* dFillIdxL = nullptr; double* dFillValL = nullptr; int64_t nFillL = 0;
  double* dFillValLi = nullptr;   // imaginary parts (complex)
  double* dFillValUi = nullptr;
  int64_t* dFillIdxU = nullptr; double* dFillValU = nullptr; int64_t nFillU = 0;
  SolveTask* dSolve = nullptr; DevBlok* dBlok = nullptr; SolveChunk *dChunk = nullptr, *dChunkB = nullptr; int32_t* dRidx = nullptr;
  std::vector<int> lvl_maxw;            // widest cblk of every level (LDS size of the solve's L^T diagonal kernel)
  // thin levels (kernels.hip, k_solve_inv): explicit inverses of their diagonal bloks, one launch per level and sweep
  std::vector<uint8_t> lvl_thin;        // [nlevels]
  std::vector<int64_t> lvl_nwide;       // [nlevels] cblks of the level wider than 64 columns (listed first)
  // a run = consecutive thin levels, one launch per sweep: [nlevels] workgroups of the run that STARTS at this level in
  // the sweep's direction (0: not a start), and where its list begins in dThinF / dThinB (dThinB: levels descending)
  std::vector<int64_t> runF_n, runF_at, runB_n, runB_at;
  SolveChunk *dThinF = nullptr, *dThinB = nullptr;      // the thin levels' workgroup lists (chunks + one per cblk)
  int32_t* dThinTasks = nullptr;        // SolveTask index of every thin cblk
  int32_t *dThinTgt = nullptr, *dThinExpect = nullptr;   // the chunks' target lists; forward contributors of every thin cblk
  int64_t nthin = 0;
  double *dInvF = nullptr, *dInvB = nullptr;            // nthin x 128 x 128: L^-1, and (L^-1)^T / U^-1 for the backward sweep
  int* dTicket = nullptr;               // 2 x nthin tickets, nthin counters, 1 "stuck" flag, the cblks' flags of both sweeps
  size_t nTicket = 0;                   // (ints; zeroed per solve)
  long long inv_gen = -1, fact_gen = 0; // the inverses belong to factorization number inv_gen
  std::vector<int64_t> lvl_chunk_ptr, lvl_chunkB_ptr;   // forward (64-row) and backward (256-row) chunk lists
  // the run schedule: device tables, the synchronisation words (zeroed per factorization), the panel kernels' streams
  bool run_ready = false, run_used = false;
  bool run_stuck = false;          // the last factorization's run launch gave up (bounded wait): ERR_DEVICE, see pastix_amd_factorize
  bool run_off_once = false;       // the next factorization takes the level-by-level schedule
  bool restorable = false;         // this factorization's input can be restored if the run stops (refillable, or the caller does it)
  bool caller_restores = false;    // the one-shot entry points: the caller's host buffers still hold the input
  bool refillable = false;         // the panels' input values are what pastix_amd_refill would write (fill_csc / refill were last)
  // PASTIX_AMD_RUN_DEBUG: host copies of the run's dependency tables, for the report of a stuck run (api.cpp)
  std::vector<RunInfo> dbg_info;
  std::vector<int32_t> dbg_cons, dbg_dep;
  std::vector<RunD> dbg_d;
  Task* dRunTasks = nullptr; RunInfo* dRunInfo = nullptr; int32_t* dRunCons = nullptr;
  size_t nRunCons = 0;                  // entries of dRunCons
  std::vector<int32_t> runDep0;         // the tickets' initial counters (what the state image was made of)
  RunD* dRunD = nullptr;
  int32_t *dRunState = nullptr, *dRunImage = nullptr;   // the counters / rings / control words and their initial image
  size_t nRunState = 0;
  int64_t run_nd = 0;
  int run_nwg = 512;                   // workgroups of the run launch: two per CU
  RunCtl runctl{};
  hipStream_t stream3 = nullptr;
  // one-shot entry points: the panels below the run are FINAL when the run's launch starts; they travel to the caller's
  // buffers on a stream of their own while it factorizes the rest (api.cpp staged_tabs_io, part 1 / part 2)
  hipStream_t stream_io = nullptr;
  void* const* early_tab = nullptr;
  void* const* early_utab = nullptr;
  bool early_done = false;
  bool ev1_recorded = false;            // the factorization's end event is already on the stream (pastix_amd_factorize_end)
  int64_t run_nticket = 0;
  int* hResident = nullptr;            // host memory the panel kernels' workgroups count themselves in
  std::vector<long long> runFeat;       // (run_prof) per ticket: what tools/run_fit.py fits the stamps against
  long long* dRunProf = nullptr;       // PASTIX_AMD_RUN_PROF: clock stamps of the run's tasks (developer aid)
  size_t nRunProf = 0;
  hipEvent_t evZ = nullptr, evS3 = nullptr;
  std::vector<hipEvent_t> ev;      // event pairs around update launches
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  pastix_amd_stats_t stats{};
};

void run_debug_report(pastix_amd_plan_t* p);   // run_debug.cpp: what a stopped run looked like (developer aid)

#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "pastix_amd: HIP error '%s' at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? PASTIX_AMD_ERR_ALLOC : PASTIX_AMD_ERR_DEVICE;   \
    }                                                                                    \
  } while (0)

// api.cpp: pieces of the device solve shared with the multi-GPU driver (dist.cpp)
extern "C" {
int pai_solve_tables(pastix_amd_plan_s* p);
void pai_solve_level(pastix_amd_plan_s* p, bool fwd, int l, double* dx, int nr);
void pai_solve_dscale(pastix_amd_plan_s* p, double* dx, int nr);
}

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T, class A>
static inline int to_device(T** d, const std::vector<T, A>& h) {
  size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc((void**)d, bytes));
  if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

