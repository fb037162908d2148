// plan.h -- internal data model of the MI355X sopalin engine (host plan + device tables).
//
// The plan is derived once from the SolverMatrix layout (cblk/blok tables, solver.h:94-168) and
// replaces the reference's mutable scheduling state (TASK_CTRBCNT counters, per-blok mutexes,
// sopalin3d.c:790-1025): dependencies become launch slots, the linear facing-blok search
// (sopalin_compute.c:938-945) becomes precomputed piece descriptors.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "../../include/pastix_amd.h"

namespace pastix_amd {



constexpr int TM = 128;   // update tile rows   (target panel row space)
constexpr int TN = 128;   // update tile cols   (target cblk column space)
constexpr int MAXW = 256; // widest cblk the panel kernels accept

// One GEMM contribution into one target tile:  C[dr:dr+m, dc:dc+n] -= A(m x k) * B(n x k)^T
// A = rows of the source panel from some blok j (or a run of bloks that land contiguously),
// B = rows of the source blok i facing the target cblk (compute_contrib_compact,
// sopalin_compute.c:270-374, fused with add_contrib_local :391-598).
struct Piece {
  int64_t a_off;      // arena offset of A(0,0)  (column-major, ld = lda)
  int64_t b_off;      // arena offset of B(0,0)  (ld = lda: same source panel)
  int32_t lda;
  uint16_t k;         // source cblk width
  uint16_t dr, m;     // destination rows inside the tile
  uint16_t dc, n;     // destination cols inside the tile
  uint16_t flags;     // bits 0-1: arena of A, bits 2-3: arena of B, bit 4: contribution is ADDED
                      // arenas: 0 = L (real part), 1 = U / L*D (real part), 2/3 = their imaginary planes
                      // bit 5: GATHERED piece (plan.cpp "gathered pieces"): A = the m CONSECUTIVE source rows from a_off,
                      // B = the n consecutive source rows from b_off, scattered over the tile's rows / columns by the row
                      // maps gmap[64 g .. 64 g + 63], g = dr | dc << 16: word i < 32 holds, one byte each, the source row of
                      // the tile rows i, i + 32, i + 64, i + 96 (255: none); words 32-63 the same for the tile's columns
};
constexpr uint16_t PIECE_GATHERED = 32;
constexpr uint32_t TASK_GATHERED = 128;   // Task flag: some piece of the task is gathered (the update kernel's MODE 3 loop)
static_assert(sizeof(Piece) == 32, "Piece must be 32 bytes");

struct Task {
  int64_t c_off;      // arena offset of the tile origin
  int32_t ldc;
  uint16_t tm, tn;    // valid extent of the tile (<= TM, TN)
  int32_t p0, pn;     // piece range
  uint32_t flags;     // bits 0-1: arena of C, bit 2: combine with atomics (tile shared by several tasks),
                      // bit 3: some pieces are "+=" pieces (Piece flag 16), bit 5: quadrant task (plan.cpp)
  uint32_t nfull;     // the first nfull pieces are full 128x128 tiles with K % 16 == 0 (specialized loop)
};
static_assert(sizeof(Task) == 32, "Task must be 32 bytes");

struct Arenas {
  double* p[4];              // device base pointers of the (up to) four planes
  const uint32_t* gmap;      // row maps of the gathered pieces (Piece flag 32): 64 words each, see Piece
};

struct PanelTask {    // one cblk for the diagonal-block kernel
  int64_t off;        // arena offset of the panel
  int32_t stride, width;
  int64_t dinv_off;   // offset (doubles) in the Tinv workspace: ceil(w/16) blocks of 16x16
};

struct TrsmTask {     // 64 panel rows of one cblk for the panel-solve kernel
  int64_t off;
  int32_t stride, width;
  int32_t row0, nrows;   // panel rows [row0, row0+nrows), row0 >= width
  int64_t dinv_off;
};

struct SolveTask {
  int64_t off;                 // panel offset in the arena
  int32_t stride, width;
  int32_t fcol;                // first column = first row of the diagonal blok
  int32_t fblok, lblok;        // blok range [fblok, lblok) in the device blok table (fblok = diagonal)
  int32_t thin;                // index among the cblks of the thin levels (explicit inverses, fused sweeps), else -1
};
struct DevBlok { int32_t frow, lrow, coefind; };
struct SolveChunk {            // 64 (forward) / 256 (backward) off-diagonal panel rows of one cblk
  int64_t off;
  int32_t stride, width, fcol, fblok, lblok;
  int32_t row0, nrows;
  int64_t roff;                // first entry of the cblk in the panel-row -> global-row table
  int32_t thin;                // thin levels: the cblk's index among the thin cblks (SolveTask::thin), else -1
  int32_t nwg;                 // thin levels: workgroups of the cblk in this list (its chunks + one with nrows = 0)
  // thin levels, runs in one launch: the thin cblks of the same run that face the chunk's rows (dThinTgt[tptr .. +tn));
  // forward: wait = the cblk receives contributions inside the run (its flag is raised by the last of them)
  int32_t tptr, tn, wait, pad_;
};


// ---- the run schedule (round 4): the thin levels at the top of the tree in ONE dependency-driven launch ---------------
// The reference's engine is counter-driven, not level-synchronous: a task runs when TASK_CTRBCNT reaches zero and the
// last contributor puts it on a ready queue (sopalin3d.c:790-1025, sopalin_compute.c:958-985, contrib.c:45-88).  The
// same here for the levels >= run_L0: every task carries a counter of the inputs that do not exist yet; whoever
// finishes an input decrements the counters of its consumers and pushes the ones that reach zero into a ready queue;
// the workgroups of ONE launch (k_run_update) each pop one ready ticket -- an update task (a tile of a target panel,
// its pieces in the plan's order), a panel-solve task (the off-diagonal rows of one 128-row tile) or, for real LLt / LDLt
// (round 5, RunCtl::onek), a diagonal-blok task; LU and complex diagonal tasks are popped by a few resident workgroups of
// a second kernel (k_run_diag_lu / k_run_diag_z).  Nobody waits for a particular
// task: a slot of the chip is idle only when nothing is ready.  Inputs: an update task waits for the previous update
// of its tile (tile ownership is a chain of tasks, not a mutex) and for the source tiles its pieces read to be solved;
// a panel-solve task for its cblk's diagonal blok and the last update of its tile; a diagonal task for the last update
// of the diagonal tile.
struct RunInfo {             // per ticket of the run
  int32_t succ;              // update ticket: who waits for this write of the tile: >= 0 the tickets [succ, succ + cn) (the
                             // tile's next update, or its panel solves), <= -2 the diagonal task -2 - succ, -1 nobody
  int32_t cptr, cn;          // panel-solve ticket: run_cons[cptr .. +cn) = the update tickets that read its tile
  int32_t kind;              // bits 0-1: 0 on the chain (urgent updates, panel solves), 1 updates of the next level's panels,
                             // 2 the rest (statistics); bit 2: panel-solve ticket (the Task record holds a TrsmTask)
};
struct RunD {                // a diagonal-blok task of the run
  PanelTask pt;
  int32_t t0, tn;            // the panel-solve tickets [t0, t0 + tn) of the cblk wait for it
};
struct RunCheck {            // host only (run_verify): what a ticket must find when it runs
  int32_t tile, seq;         // update: tile counter index and its value; panel solve: the L-arena tile it makes final
  int32_t wptr, wn;          // update: run_waits[wptr .. +wn) source tiles that must be final; panel solve: wn = -1, wptr = cblk's diagonal task
};

// device-side state of a run: one block of ints, reset from an image of the same layout before every factorization
struct RunCtl {
  int32_t* cnt;              // [ntickets + ndiag] inputs a task still waits for
  int32_t* q;                // ready ring of the tickets (every ticket is pushed once: no wrap); -1 = empty slot
  int32_t* qd;               // ready ring of the diagonal tasks
  int32_t* ctl;              // heads, tails, the "stuck" flag: RUN_* below, one 256-byte line each
  int32_t nd, nticket;       // ring sizes
  int32_t room;              // workgroups of the run launch that leave rather than wait when everybody else already waits (run_sync.h)
  int32_t onek;              // 1: the diagonal tasks are tickets of k_run_update too -- a diagonal task d that becomes ready is
                             // pushed into the TICKET ring as nticket + d (the ring has nticket + nd slots), no second kernel
  long long* prof;           // developer aid (PASTIX_AMD_RUN_PROF): 4 clock stamps per ticket, then per diagonal task; else null
};
constexpr int RUN_HEAD = 0, RUN_TAIL = 2 * 64, RUN_STUCK = 4 * 64, RUN_CTL_INTS = 6 * 64;   // (+ 64: the diagonal ring's)
constexpr int RUN_GO = RUN_STUCK + 32;   // set by the tickets' kernel when it starts: the resident diagonal workers (LU, complex)
                                         // time their waits from then on -- while the levels below the run are factorized, or
                                         // on a slower / shared device, they wait without a clock (60 s in all at most)
// a ring slot is one 128-byte line (slot i at ring[i * RUN_SLOT]): the workgroups that wait hold CONSECUTIVE slots, and five
// hundred of them polling sixteen lines starved the chip's memory system -- about one factorization in 200 had every running
// ticket's loads stand still until the pollers gave up
constexpr int RUN_SLOT = 32;

// Developer switches: ONE environment variable, PASTIX_AMD_DEV="key[=value],key[=value],..." (a key without a value reads
// "1").  Keys: plan_timing, near=<levels>, nearc=<cblks> (near-target task cutting, plan.cpp), dump_slot=<slot>[:file]
// (tools/replay_slot), run_prof=<file> (clock stamps of every task of the run, tools/run_prof.py), run_debug (keeps the run's
// dependency tables on the host: a stopped run is replayed and reported, run_debug.cpp), onek=0 (the run's diagonal tasks on
// a resident kernel of their own, the round-4 form: the tests' oracle for the one-kernel form), room=<workgroups> (run_sync.h),
// gather=<n> (overrides options.gather_min; -1: rectangles only).
// Returns the value (valid until the next call on this thread) or nullptr.
inline const char* dev_opt(const char* key) {
  static thread_local char buf[512];
  const char* e = getenv("PASTIX_AMD_DEV");
  if (!e) return nullptr;
  const size_t kl = strlen(key);
  for (const char* q = e; *q;) {
    const char* end = strchr(q, ',');
    const size_t len = end ? (size_t)(end - q) : strlen(q);
    if (len >= kl && !strncmp(q, key, kl) && (len == kl || q[kl] == '=')) {
      if (len == kl) { buf[0] = '1'; buf[1] = 0; return buf; }
      const size_t vl = std::min(len - kl - 1, sizeof(buf) - 1);
      memcpy(buf, q + kl + 1, vl);
      buf[vl] = 0;
      return buf;
    }
    if (!end) break;
    q = end + 1;
  }
  return nullptr;
}

// std::allocator whose value-less construct() default-initialises (leaves trivially constructible T untouched)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};

struct Plan {
  int factotype = 0, floattype = 1;
  pastix_amd_options_t opts{};
  int64_t cblknbr = 0, bloknbr = 0, coefnbr = 0, ncol = 0;
  std::vector<pastix_amd_cblk_t> cblk;   // [cblknbr+1]
  std::vector<pastix_amd_blok_t> blok;
  std::vector<int64_t> poff;             // panel offsets [cblknbr+1] (absent cblks have size 0)
  std::vector<int8_t> role;              // per cblk: 1 owned (factorized here), 2 shadow (fan-in
                                         // accumulator for a remote cblk), 0 absent
  std::vector<int32_t> owner;            // [cblknbr] rank that factorizes the cblk (empty on one GPU)
  int32_t myrank = 0;
  std::vector<uint64_t> fanin_mask;      // [bloknbr] fanin_touched (empty on one GPU)
  double local_flops = 0;                // fact_flops restricted to owned cblks
  std::vector<int32_t> level;            // dependency level of each cblk
  int32_t nlevels = 0;

  // per level
  std::vector<int64_t> lvl_panel_ptr;    // [nlevels+1] into panel_tasks
  std::vector<PanelTask> panel_tasks;
  std::vector<int64_t> lvl_trsm_ptr;     // [nlevels+1] into trsm_tasks
  std::vector<TrsmTask> trsm_tasks;
  int64_t dinv_ws = 0;                   // doubles needed for the Tinv workspace

  // per slot (slot s runs right before level s is factorized)
  std::vector<int64_t> tstride;          // [cblknbr] leading dimension of the panel as a contribution TARGET here
  std::vector<int64_t> tcoef;            // [bloknbr] row offset of the blok in that panel (-1: not present in a
                                         // compact shadow); equal to stride / coefind except for shadow cblks
  std::vector<int64_t> slot_task_ptr;    // [nlevels+1]
  std::vector<int64_t> slot_next_end;    // [nlevels] then, up to here, tasks whose targets are of level s+1
  std::vector<int64_t> slot_urgent_end;  // [nlevels] tasks [slot_task_ptr[s], slot_urgent_end[s]) target cblks of
                                         // level s itself (needed by this level's panel kernels); the rest of the
                                         // slot only feeds later levels and may overlap with the panel kernels
  std::vector<int64_t> slot_small_begin; // [nlevels] bulk tasks [slot_small_begin[s], slot_task_ptr[s+1]) are quadrant tasks (Task
                                         // flag 32: tile = a 64x64 quadrant, small pieces) for k_update_small
  std::vector<int64_t> slot_usmall_begin;// [nlevels] urgent tasks [slot_usmall_begin[s], slot_urgent_end[s]): the same
  std::vector<Task> tasks;
  std::vector<Piece, NoInitAlloc<Piece>> pieces;   // (filled by parallel copies: no serial zero fill of ~10 GB first)
  std::vector<uint32_t> gmaps;           // row maps of the gathered pieces, 64 words each (Piece)
  double update_flops = 0;
  double full_flops = 0;                 // part of update_flops in full 128x128 pieces (specialized loop)
  double urgent_flops = 0;               // part of update_flops in the urgent tasks of their slots
  double update_bytes = 0;               // algorithmic bytes of the update kernel: operands read once per
                                         // piece (8k(m+n)) + one read-modify-write of the tile per task (16 tm tn)
  std::vector<double> slot_flops;        // [nlevels] update flops per slot
  std::vector<double> slot_urgent_flops; // [nlevels] part of slot_flops in the urgent tasks (targets of level == slot)
  std::vector<double> slot_mode_flops;   // [nlevels][3] (verbose >= 2 only) bulk flops by update-loop instance
  std::vector<int64_t> slot_pieces;      // [nlevels]
  std::vector<int32_t> slot_maxpn;       // [nlevels] longest piece list of a task in the slot
  std::vector<double> slot_maxwork;      // [nlevels] largest task (multiply-adds)
  double fact_flops = 0;

  // run schedule (see RunInfo): levels [run_L0, nlevels); -1: none
  int32_t run_L0 = -1;
  int64_t ntile = 0;                     // target tiles per plane
  int32_t nplanes = 1;                   // planes that are update targets (1 LLt/LDLt, 2 LU, x2 complex)
  std::vector<Task> run_tasks;           // the tickets: update tasks of slots >= run_L0 and panel-solve tasks (a TrsmTask in
                                         // the record) of levels >= run_L0
  std::vector<RunInfo> run_info;         // [run_tasks.size()]
  std::vector<int32_t> run_cons;         // consumer lists of the panel-solve tickets
  std::vector<int32_t> run_dep;          // [tickets + diagonal tasks] initial counters
  std::vector<int32_t> run_ready, run_dready;      // tasks that are ready when the run starts, per ring
  std::vector<RunD> run_d;               // diagonal tasks, level-major
  int32_t run_gd = 0;                    // resident workgroups for the diagonal tasks
  double run_flops = 0;                  // update flops inside the run
  std::vector<RunCheck> run_chk;         // host only (run_verify)
  std::vector<uint8_t> run_tile_nt;      // ... panel-solve tickets per tile
  // Round 6: the reader lists of the run (run_cons, the panel-solve tickets' cptr / cn, the readers' share of run_dep) built
  // on the DEVICE from the uploaded tickets and pieces (run_edges.hip) instead of on host threads.  defer_run_edges is set by
  // the caller of build_plan (api.cpp, when it has a device); run_edges_deferred says that build_plan left them out: then
  // run_waits / run_cons / run_ready are empty, run_chk has no source tiles, and the three tables below are what the device
  // builder reads.
  bool defer_run_edges = false, run_edges_deferred = false;
  std::vector<int32_t> run_tile_ticket;  // [ntile] first panel-solve ticket of a tile (-1: none; run_tile_nt: how many, consecutive)
  std::vector<int32_t> run_tile_base;    // [cblknbr + 1] first tile of a cblk's panel
  std::vector<std::pair<int32_t, int32_t>> run_dchk;   // ... per diagonal task: its tile counter and the value it must find
  std::vector<int32_t> run_waits;
  std::vector<uint8_t> run_cat;          // (PASTIX_AMD_RUN_PROF) per ticket: 0 A, 1 B.next, 2 B.rest, 3 panel solve; its slot / level
  std::vector<int32_t> run_lvl;

  // solve schedule: cblks grouped by level (same levels as the factorization)
  std::vector<int64_t> lvl_cblk_ptr;     // [nlevels+1]
  std::vector<int32_t> lvl_cblk;
};

// Build the host plan. Returns PASTIX_AMD_OK or an error code.
int build_plan(const pastix_amd_layout_t* layout, int factotype, int floattype,
               const pastix_amd_options_t* opts, const int32_t* owner, int32_t myrank, Plan& plan);

int64_t run_verify(const Plan& plan);

int verify_pieces(const Plan& plan, int64_t out[4]);
int owner_view(const pastix_amd_layout_t* layout, const int32_t* owner, int32_t myrank, Plan& plan);
double fact_flops(const pastix_amd_layout_t* layout, int factotype, int floattype);
int fanin_touched(const pastix_amd_layout_t* layout, const int32_t* owner, uint64_t* mask);

}  // namespace pastix_amd
