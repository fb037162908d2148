// dist.cpp -- multi-GPU numerical factorization: one rank per GPU, elimination-tree subtrees mapped one per GPU,
// asynchronous fan-in of aggregated contributions over RCCL point-to-point.
//
// Reference model.  With several MPI processes PaStiX accumulates the contributions a process makes to a remote
// cblk in a FanInTarget buffer (add_contrib_target, src/sopalin/src/sopalin_compute.c:600-733: SUBTRACTED into a
// zero-initialised block), sends the block the moment its last local contribution has landed (:708-729, the
// communication thread of sopalin_sendrecv.c:2393-2775 with receives posted ahead :1219-1556) and the owner ADDS it
// into the panel before the cblk is factorized (recv_handle_fanin, sopalin_sendrecv.c:182-485, the add :384-404).
//
// Here the whole factorization of a rank is ENQUEUED without the host ever waiting for a peer:
//   * panel stream (high priority): per level l   A(l) urgent contributions -> adds of the received blocks of level l
//     -> P(l) diagonal factor + panel solve;   second stream: B(l), the bulk contributions (api.cpp, two-stream split);
//   * one channel per peer = one 2-rank RCCL communicator + one HIP stream.  The fan-in blocks a rank holds for the
//     cblks of level l are complete when A(l) is (every older contribution ran in an earlier bulk launch, which A(l)
//     waits for), so the channel waits for the event recorded behind A(l) and sends; the owner's channel receives into
//     a staging area and the panel stream waits for that channel's event before it adds.  A rank that only sends never
//     waits; a rank that receives waits on the device, for exactly the blocks its next panel needs.
//   * Both ends of a channel issue their operations in the same order -- by (level of the target cblk, peer, cblk) --
//     and all operations of one (level, peer) form one ncclGroup, so the rendezvous cannot deadlock: everything an
//     operation waits for was enqueued earlier, on every rank, by induction over the levels.
// No collective is involved on the data path.  The schedule (which blocks, in which order, which rows) is host logic,
// exported as pastix_amd_dist_schedule so that the CPU tests replay it over gloo.
//
// Transports: RCCL (librccl resolved at run time: dlopen, no link dependency, the copy PyTorch loaded is reused when
// there is one) and an in-process loopback (several rank plans sharing one GPU, host threads, device-to-device copies)
// used by the single-GPU tests to run the very same driver.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "engine.h"

namespace {

// arenas (planes) that receive contributions: L; + U for LU; + the imaginary planes for complex
int planes_of(int factotype, bool cplx, int out[4]) {
  int n = 0;
  out[n++] = 0;
  if (factotype == PASTIX_AMD_FACT_LU) out[n++] = 1;
  if (cplx) {
    out[n++] = 2;
    if (factotype == PASTIX_AMD_FACT_LU) out[n++] = 3;
  }
  return n;
}

struct DistMsg {
  int32_t level, peer, cblk, dir;   // dir 0: this rank sends its fan-in block of `cblk` to `peer` (= owner); 1: receives
  int64_t nrows, width;             // the block is nrows x width per plane, column-major, ld = nrows
  int64_t off;                      // send: arena offset of the compact fan-in panel; recv: offset in the staging area
                                    // of plane 0 of this message (plane q follows at + q * nrows * width)
  int64_t rows_off;                 // recv: first entry of the block's row map (panel row of every block row)
};

struct DistSchedule {
  int world = 1, myrank = 0, nplanes = 1;
  int planes[4] = {0, 0, 0, 0};
  std::vector<DistMsg> msgs;        // sorted by (level, peer, cblk)
  std::vector<int32_t> rows;
  int64_t stage_elems = 0;
  std::vector<int> peers;           // ascending
  std::vector<std::pair<int, int>> pairs;   // every communicating pair (a < b) of the whole job, lexicographic
};

// Deterministic from (layout, owner): every rank derives its own list and, implicitly, the matching one of its peers.
int build_schedule(const Plan& P, int world, DistSchedule& S) {
  if (P.owner.empty() || world < 1 || world > 64) return PASTIX_AMD_ERR_BADPARAMETER;
  const int64_t nc = P.cblknbr;
  const int me = P.myrank;
  S.world = world;
  S.myrank = me;
  S.nplanes = planes_of(P.factotype, P.floattype == PASTIX_AMD_COMPLEXDOUBLE, S.planes);
  S.msgs.clear();
  S.rows.clear();
  S.stage_elems = 0;
  std::vector<uint64_t> pairbits((size_t)world, 0);          // pairbits[a] bit b: a sends to b
  for (int64_t t = 0; t < nc; t++) {
    const int ot = P.owner[t];
    if (ot < 0 || ot >= world) return PASTIX_AMD_ERR_BADPARAMETER;
    uint64_t senders = 0;
    for (int64_t b = P.cblk[t].bloknum; b < P.cblk[t + 1].bloknum; b++) senders |= P.fanin_mask[b];
    senders &= ~(1ull << ot);
    for (int r = 0; r < world; r++)
      if ((senders >> r) & 1ull) pairbits[(size_t)r] |= 1ull << ot;
    const int64_t w = P.cblk[t].lcolnum - P.cblk[t].fcolnum + 1;
    if (ot == me) {
      for (int r = 0; r < world; r++) {
        if (!((senders >> r) & 1ull)) continue;
        DistMsg m{P.level[t], r, (int32_t)t, 1, 0, w, 0, (int64_t)S.rows.size()};
        for (int64_t b = P.cblk[t].bloknum; b < P.cblk[t + 1].bloknum; b++) {
          if (!((P.fanin_mask[b] >> r) & 1ull)) continue;
          const int64_t h = P.blok[b].lrownum - P.blok[b].frownum + 1;
          for (int64_t i = 0; i < h; i++) S.rows.push_back((int32_t)(P.blok[b].coefind + i));
          m.nrows += h;
        }
        S.msgs.push_back(m);
      }
    } else if ((senders >> me) & 1ull) {
      if (P.role[t] != 2) return PASTIX_AMD_ERR_LAYOUT;
      S.msgs.push_back(DistMsg{P.level[t], ot, (int32_t)t, 0, P.tstride[t], w, P.poff[t], 0});
    }
  }
  std::sort(S.msgs.begin(), S.msgs.end(), [](const DistMsg& a, const DistMsg& b) {
    if (a.level != b.level) return a.level < b.level;
    if (a.peer != b.peer) return a.peer < b.peer;
    return a.cblk < b.cblk;
  });
  for (DistMsg& m : S.msgs)
    if (m.dir == 1) { m.off = S.stage_elems; S.stage_elems += m.nrows * m.width * S.nplanes; }
  uint64_t mine = 0;
  S.pairs.clear();
  for (int a = 0; a < world; a++)
    for (int b = a + 1; b < world; b++)
      if (((pairbits[(size_t)a] >> b) & 1ull) || ((pairbits[(size_t)b] >> a) & 1ull)) {
        S.pairs.emplace_back(a, b);
        if (a == me) mine |= 1ull << b;
        if (b == me) mine |= 1ull << a;
      }
  S.peers.clear();
  for (int r = 0; r < world; r++)
    if ((mine >> r) & 1ull) S.peers.push_back(r);
  return PASTIX_AMD_OK;
}

// Deadline of one distributed factorization / solve on the host (seconds; PASTIX_AMD_DIST_TIMEOUT, default 300): a
// rank whose streams have not drained by then reports where they are stuck, aborts its channels and returns
// PASTIX_AMD_ERR_TIMEOUT instead of waiting for a peer for ever.
double dist_timeout_s() {
  static const double t = [] {
    const char* e = getenv("PASTIX_AMD_DIST_TIMEOUT");
    const double v = e ? atof(e) : 300.0;
    return v > 0 ? v : 300.0;
  }();
  return t;
}

// FNV-1a over 64-bit words
inline void fnv(uint64_t& h, uint64_t v) {
  for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xffu; h *= 1099511628211ull; }
}

// ------------------------------------------------------------------------------------------------
// transports
// ------------------------------------------------------------------------------------------------
struct Transport {
  virtual ~Transport() {}
  virtual int group_begin(int peer) = 0;
  virtual int send(int peer, const double* buf, int64_t count, hipStream_t s) = 0;
  virtual int recv(int peer, double* buf, int64_t count, hipStream_t s) = 0;
  virtual int group_end(int peer) = 0;
  // give up: make every operation already enqueued on the device return (peers see an error, not a hang)
  virtual void abort() = 0;
  virtual const char* name() const = 0;
};

// --- RCCL, resolved at run time ---
struct RcclApi {
  void* h = nullptr;
  char path[512] = {0};              // the object the entry points were resolved in (dladdr)
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// The library choice is deterministic: ONE copy of librccl per process.  If a librccl is already mapped (PyTorch's
// torch/lib/librccl.so once torch.distributed has been imported), exactly that object is used -- found by walking the
// loaded objects, reopened by its own path with RTLD_NOLOAD -- and a failure to do so is an error, not a reason to map
// a second copy next to it (two copies = two sets of global state = the exit-time aborts the first rounds worked
// around).  Only a process that has no librccl yet maps the ROCm installation's.
struct FoundLib { char path[512]; int n; };
int find_rccl_cb(struct dl_phdr_info* info, size_t, void* data) {
  FoundLib* f = (FoundLib*)data;
  if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) {
    if (f->n == 0) std::snprintf(f->path, sizeof(f->path), "%s", info->dlpi_name);
    f->n++;
  }
  return 0;
}

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    FoundLib f{{0}, 0};
    dl_iterate_phdr(find_rccl_cb, &f);
    if (f.n > 1) {
      fprintf(stderr, "pastix_amd: %d copies of librccl are mapped in this process (first: %s); refusing to pick one\n", f.n, f.path);
      return;
    }
    if (f.n == 1) {
      api.h = dlopen(f.path, RTLD_NOW | RTLD_NOLOAD);
      if (!api.h) {
        fprintf(stderr, "pastix_amd: librccl is mapped (%s) but cannot be reopened: %s\n", f.path, dlerror());
        return;
      }
    } else {
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names) {
        api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.h) break;
      }
      if (!api.h) return;
    }
#define SYM(f) api.f = (decltype(api.f))dlsym(api.h, "nccl" #f)
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(CommAbort); SYM(Send); SYM(Recv); SYM(GroupStart);
    SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.CommAbort || !api.Send || !api.Recv ||
        !api.GroupStart || !api.GroupEnd) {
      dlclose(api.h);
      api.h = nullptr;
      return;
    }
    Dl_info di;
    if (dladdr((void*)api.Send, &di) && di.dli_fname) std::snprintf(api.path, sizeof(api.path), "%s", di.dli_fname);
    if (getenv("PASTIX_AMD_VERBOSE")) fprintf(stderr, "pastix_amd: librccl resolved in %s\n", api.path);
  });
  return api.h ? &api : nullptr;
}

#define NCCLCHK(x)                                                                                     \
  do {                                                                                                 \
    ncclResult_t r_ = (x);                                                                             \
    if (r_ != ncclSuccess) {                                                                           \
      fprintf(stderr, "pastix_amd: RCCL error '%s' at %s:%d\n",                                        \
              rccl() && rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?", __FILE__, __LINE__); \
      return PASTIX_AMD_ERR_DEVICE;                                                                    \
    }                                                                                                  \
  } while (0)

struct RcclTransport : Transport {
  int me = 0;
  std::map<int, ncclComm_t> comm;     // per peer: a 2-rank communicator (rank 0 = the lower job rank)
  bool aborted = false;
  ~RcclTransport() override {
    if (aborted) return;
    if (RcclApi* a = rccl())
      for (auto& c : comm) if (c.second) (void)a->CommDestroy(c.second);
  }
  // ncclCommAbort: the kernels of pending sends / receives leave, the communicator is freed (no CommDestroy after it)
  void abort() override {
    if (aborted) return;
    aborted = true;
    if (RcclApi* a = rccl())
      for (auto& c : comm) if (c.second) { (void)a->CommAbort(c.second); c.second = nullptr; }
  }
  int group_begin(int) override { NCCLCHK(rccl()->GroupStart()); return 0; }
  int group_end(int) override { NCCLCHK(rccl()->GroupEnd()); return 0; }
  int send(int peer, const double* buf, int64_t count, hipStream_t s) override {
    NCCLCHK(rccl()->Send(buf, (size_t)count, ncclDouble, me < peer ? 1 : 0, comm[peer], s));
    return 0;
  }
  int recv(int peer, double* buf, int64_t count, hipStream_t s) override {
    NCCLCHK(rccl()->Recv(buf, (size_t)count, ncclDouble, me < peer ? 1 : 0, comm[peer], s));
    return 0;
  }
  const char* name() const override { return "rccl"; }
};

// --- loopback: rank plans of one process share a GPU; a send publishes (event, pointer), the matching receive -- the
// k-th receive from a peer matches that peer's k-th send, as on an in-order channel -- waits for the event on its own
// stream and copies device to device.  Test infrastructure for the driver on single-GPU boxes. ---
struct LocalHub {
  std::mutex mu;
  std::condition_variable cv;
  struct Posted { hipEvent_t ev; const double* ptr; int64_t count; };
  std::map<std::pair<int, int>, std::deque<Posted>> q;     // (src, dst) -> posted sends
  std::vector<hipEvent_t> pool;                            // all events ever made (destroyed with the hub)
  bool failed = false;
  ~LocalHub() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

struct LocalTransport : Transport {
  std::shared_ptr<LocalHub> hub;
  int me = 0;
  int group_begin(int) override { return 0; }
  int group_end(int) override { return 0; }
  int send(int peer, const double* buf, int64_t count, hipStream_t s) override {
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, s));
    {
      std::lock_guard<std::mutex> g(hub->mu);
      hub->pool.push_back(ev);
      hub->q[{me, peer}].push_back(LocalHub::Posted{ev, buf, count});
    }
    hub->cv.notify_all();
    return 0;
  }
  void abort() override {
    { std::lock_guard<std::mutex> g(hub->mu); hub->failed = true; }
    hub->cv.notify_all();
  }
  int recv(int peer, double* buf, int64_t count, hipStream_t s) override {
    LocalHub::Posted p;
    {
      std::unique_lock<std::mutex> g(hub->mu);
      auto& dq = hub->q[{peer, me}];
      // (the host-side rendezvous of the emulation has the same deadline as the device-side wait of the real driver)
      if (!hub->cv.wait_for(g, std::chrono::duration<double>(dist_timeout_s()), [&] { return !dq.empty() || hub->failed; })) {
        fprintf(stderr, "pastix_amd[rank %d]: loopback receive from rank %d (%lld doubles) not matched within %.0f s\n", me,
                peer, (long long)count, dist_timeout_s());
        return PASTIX_AMD_ERR_TIMEOUT;
      }
      if (hub->failed) return PASTIX_AMD_ERR_DEVICE;
      p = dq.front();
      dq.pop_front();
    }
    if (p.count != count) {                                    // the two ends disagree on the schedule
      fprintf(stderr, "pastix_amd[rank %d]: receive from rank %d expects %lld doubles, the matching send has %lld\n", me,
              peer, (long long)count, (long long)p.count);
      return PASTIX_AMD_ERR_LAYOUT;
    }
    HIPCHK(hipStreamWaitEvent(s, p.ev, 0));
    HIPCHK(hipMemcpyAsync(buf, p.ptr, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return 0;
  }
  const char* name() const override { return "loopback"; }
};

}  // namespace

// state a distributed plan carries once a transport is attached
struct pastix_amd_dist_s {
  DistSchedule S;
  std::unique_ptr<Transport> T;
  std::map<int, hipStream_t> chan;          // per peer
  double* dStage = nullptr;
  int32_t* dRows = nullptr;
  std::vector<hipEvent_t> events;           // pool reused by every factorization (grown on demand)
  size_t nev_used = 0;
  double bytes_sent = 0, bytes_recv = 0;
  int64_t nsend = 0, nrecv = 0;
  double* dXs = nullptr;                    // solve: the rank's copy of the vector
  double* dSolveStage = nullptr;            // solve: received segments (forward sweep)
  double* hXs = nullptr;                    // solve: pinned host staging of the vector (lives as long as the plan: an
  size_t nXs = 0;                           // asynchronous copy may still target it when a run is abandoned, see dist_finish)
  std::vector<int64_t> solve_stage_off;     // per message: offset in dSolveStage (receive messages)
  // progress marks of the run in flight: one event behind every (level, peer) group on its channel stream and one
  // behind every level on the panel stream -- what the deadline handler reads to say where a rank is stuck
  struct Mark { hipEvent_t ev; int level, peer; size_t m0, m1; };
  std::vector<Mark> marks;
  std::vector<std::pair<hipEvent_t, int>> level_marks;
  bool failed = false;                      // a run was aborted: the channels are gone, later calls refuse
};

static void dist_free(pastix_amd_dist_s* D) {
  if (!D) return;
  for (auto& c : D->chan) if (c.second) { (void)hipStreamSynchronize(c.second); (void)hipStreamDestroy(c.second); }
  D->T.reset();
  (void)hipFree(D->dStage);
  (void)hipFree(D->dRows);
  (void)hipFree(D->dXs);
  (void)hipFree(D->dSolveStage);
  (void)hipDeviceSynchronize();                      // (nothing enqueued may still write the host staging)
  if (D->hXs) (void)hipHostFree(D->hXs);
  for (hipEvent_t e : D->events) (void)hipEventDestroy(e);
  delete D;
}

static int dist_attach_common(pastix_amd_plan_t* p, int world, std::unique_ptr<Transport> T, DistSchedule&& S) {
  std::unique_ptr<pastix_amd_dist_s, void (*)(pastix_amd_dist_s*)> D(new (std::nothrow) pastix_amd_dist_s(), dist_free);
  if (!D) return PASTIX_AMD_ERR_ALLOC;
  (void)world;
  D->S = std::move(S);
  D->T = std::move(T);
  HIPCHK(hipSetDevice(p->device));
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  for (int peer : D->S.peers) {
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));   // like the panel stream: between the bulk workgroups
    D->chan[peer] = s;
  }
  HIPCHK(hipMalloc((void**)&D->dStage, (size_t)std::max<int64_t>(D->S.stage_elems, 1) * sizeof(double)));
  int r;
  if ((r = to_device(&D->dRows, D->S.rows))) return r;
  for (const DistMsg& m : D->S.msgs) {
    const double b = 8.0 * (double)m.nrows * (double)m.width * D->S.nplanes;
    if (m.dir == 0) { D->bytes_sent += b; D->nsend++; } else { D->bytes_recv += b; D->nrecv++; }
  }
  if (p->dist) p->dist_free(p->dist);
  p->dist = D.release();
  p->dist_free = dist_free;
  return PASTIX_AMD_OK;
}

extern "C" {

static int host_schedule(const pastix_amd_layout_t* layout, int factotype, int floattype, const int32_t* owner,
                         int32_t myrank, int32_t world, DistSchedule& S) {
  Plan P;
  P.factotype = factotype;
  P.floattype = floattype;
  P.cblknbr = layout->cblknbr;
  P.bloknbr = layout->bloknbr;
  P.cblk.assign(layout->cblktab, layout->cblktab + layout->cblknbr + 1);
  P.blok.assign(layout->bloktab, layout->bloktab + layout->bloknbr);
  int rc = owner_view(layout, owner, myrank, P);
  if (rc) return rc;
  return build_schedule(P, world, S);
}

// What a rank expects of every channel, as two hashes per peer: [2q] over the blocks it SENDS to rank q, [2q+1] over the
// blocks it RECEIVES from q, each over (level, cblk, nrows, width, planes) in channel order.  The two ends of a channel
// agree iff rank a's [2b] equals rank b's [2a+1] and vice versa; compared at attach time (loopback: here; RCCL: by the
// launcher over its bootstrap, dist.py) so that a disagreement is an error message, not a rendezvous that never ends.
static void schedule_hashes(const DistSchedule& S, uint64_t* out) {
  for (int q = 0; q < 2 * S.world; q++) out[q] = 14695981039346656037ull;
  for (const DistMsg& m : S.msgs) {
    uint64_t& h = out[2 * m.peer + (m.dir ? 1 : 0)];
    fnv(h, (uint64_t)m.level); fnv(h, (uint64_t)m.cblk); fnv(h, (uint64_t)m.nrows); fnv(h, (uint64_t)m.width);
    fnv(h, (uint64_t)S.nplanes);
  }
}

// Host only: the fan-in messages of one rank, in the order both ends of every channel issue them.
// out[i*6 .. i*6+5] = {level, peer, cblk, dir (0 send / 1 recv), nrows, width}; returns the count through *nmsg
// (out may be NULL to size it; at most `cap` messages are written).  *nplanes = arenas per message.
int pastix_amd_dist_schedule(const pastix_amd_layout_t* layout, int factotype, int floattype, const int32_t* owner,
                             int32_t myrank, int32_t world, pastix_amd_int_t cap, pastix_amd_int_t* out,
                             pastix_amd_int_t* nmsg, int32_t* nplanes) {
  if (!layout || !owner || !nmsg || !layout->cblktab || !layout->bloktab || layout->cblknbr < 1)
    return PASTIX_AMD_ERR_BADPARAMETER;
  try {
    DistSchedule S;
    int rc = host_schedule(layout, factotype, floattype, owner, myrank, world, S);
    if (rc) return rc;
    *nmsg = (pastix_amd_int_t)S.msgs.size();
    if (nplanes) *nplanes = S.nplanes;
    if (out)
      for (size_t i = 0; i < S.msgs.size() && (pastix_amd_int_t)i < cap; i++) {
        const DistMsg& m = S.msgs[i];
        pastix_amd_int_t* o = out + 6 * i;
        o[0] = m.level; o[1] = m.peer; o[2] = m.cblk; o[3] = m.dir; o[4] = m.nrows; o[5] = m.width;
      }
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_dist_schedule_hash(const pastix_amd_layout_t* layout, int factotype, int floattype, const int32_t* owner,
                                  int32_t myrank, int32_t world, uint64_t* out) {
  if (!layout || !owner || !out || !layout->cblktab || !layout->bloktab || layout->cblknbr < 1 || world < 1)
    return PASTIX_AMD_ERR_BADPARAMETER;
  try {
    DistSchedule S;
    const int rc = host_schedule(layout, factotype, floattype, owner, myrank, world, S);
    if (rc) return rc;
    schedule_hashes(S, out);
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_dist_unique_id(void* id128) {
  if (!id128) return PASTIX_AMD_ERR_BADPARAMETER;
  RcclApi* a = rccl();
  if (!a) { fprintf(stderr, "pastix_amd: librccl not found\n"); return PASTIX_AMD_ERR_DEVICE; }
  static_assert(sizeof(ncclUniqueId) == PASTIX_AMD_DIST_ID_BYTES, "unique id size");
  ncclUniqueId id;
  NCCLCHK(a->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return PASTIX_AMD_OK;
}

// One rank, one communicator, a grouped send-to-self through the very entry points (run-time resolved librccl, ncclDouble,
// group calls, a non-default stream) the fan-in channels use: the part of the RCCL path a single-GPU box can execute.
int pastix_amd_dist_selftest_rccl(int device, pastix_amd_int_t count) {
  if (count < 1) return PASTIX_AMD_ERR_BADPARAMETER;
  RcclApi* a = rccl();
  if (!a) { fprintf(stderr, "pastix_amd: librccl not found\n"); return PASTIX_AMD_ERR_DEVICE; }
  HIPCHK(hipSetDevice(device));
  ncclUniqueId id;
  NCCLCHK(a->GetUniqueId(&id));
  ncclComm_t c = nullptr;
  NCCLCHK(a->CommInitRank(&c, 1, id, 0));
  int rc = PASTIX_AMD_OK;
  double *src = nullptr, *dst = nullptr;
  hipStream_t s = nullptr;
  std::vector<double> h((size_t)count), back((size_t)count, 0.0);
  for (int64_t i = 0; i < count; i++) h[(size_t)i] = 0.5 * (double)i - 3.0;
  auto body = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void**)&src, (size_t)count * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dst, (size_t)count * sizeof(double)));
    HIPCHK(hipMemcpy(src, h.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dst, 0, (size_t)count * sizeof(double)));
    NCCLCHK(a->GroupStart());
    NCCLCHK(a->Send(src, (size_t)count, ncclDouble, 0, c, s));
    NCCLCHK(a->Recv(dst, (size_t)count, ncclDouble, 0, c, s));
    NCCLCHK(a->GroupEnd());
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(back.data(), dst, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return std::memcmp(back.data(), h.data(), (size_t)count * sizeof(double)) ? PASTIX_AMD_ERR_NUMERIC : PASTIX_AMD_OK;
  };
  rc = body();
  (void)hipFree(src);
  (void)hipFree(dst);
  if (s) (void)hipStreamDestroy(s);
  (void)a->CommDestroy(c);
  return rc;
}

// ids: world x world x 128 bytes; entry [a * world + b], a < b, made by pastix_amd_dist_unique_id on ANY one rank and
// distributed out of band (bench.py: all_gather over torch.distributed).  Collective over the job: every rank calls it,
// pairs are initialised in lexicographic order on both of their members.
int pastix_amd_dist_attach_rccl(pastix_amd_plan_t* p, int32_t world, const void* ids) {
  if (!p || !ids || !p->distributed || world < 2) return PASTIX_AMD_ERR_BADPARAMETER;
  RcclApi* a = rccl();
  if (!a) { fprintf(stderr, "pastix_amd: librccl not found\n"); return PASTIX_AMD_ERR_DEVICE; }
  DistSchedule S;
  int rc = build_schedule(p->host, world, S);
  if (rc) return rc;
  HIPCHK(hipSetDevice(p->device));
  std::unique_ptr<RcclTransport> T(new (std::nothrow) RcclTransport());
  if (!T) return PASTIX_AMD_ERR_ALLOC;
  T->me = S.myrank;
  for (auto& pr : S.pairs) {
    if (pr.first != S.myrank && pr.second != S.myrank) continue;
    const int peer = pr.first == S.myrank ? pr.second : pr.first;
    ncclUniqueId id;
    std::memcpy(&id, (const char*)ids + ((size_t)pr.first * world + pr.second) * PASTIX_AMD_DIST_ID_BYTES, sizeof(id));
    ncclComm_t c = nullptr;
    NCCLCHK(a->CommInitRank(&c, 2, id, S.myrank == pr.first ? 0 : 1));
    T->comm[peer] = c;
  }
  return dist_attach_common(p, world, std::move(T), std::move(S));
}

// Test / single-box emulation: `world` rank plans of ONE process (typically sharing one GPU) are wired to each other.
int pastix_amd_dist_attach_local(pastix_amd_plan_t* const* plans, int32_t world) {
  if (!plans || world < 2) return PASTIX_AMD_ERR_BADPARAMETER;
  std::shared_ptr<LocalHub> hub;
  try { hub = std::make_shared<LocalHub>(); } catch (const std::bad_alloc&) { return PASTIX_AMD_ERR_ALLOC; }
  {
    std::vector<uint64_t> hs((size_t)world * 2 * (size_t)world);
    for (int r = 0; r < world; r++) {
      if (!plans[r] || !plans[r]->distributed || plans[r]->host.myrank != r) return PASTIX_AMD_ERR_BADPARAMETER;
      DistSchedule S;
      const int rc = build_schedule(plans[r]->host, world, S);
      if (rc) return rc;
      schedule_hashes(S, hs.data() + (size_t)r * 2 * world);
    }
    for (int a = 0; a < world; a++)
      for (int b = 0; b < world; b++)
        if (a != b && hs[(size_t)a * 2 * world + 2 * b] != hs[(size_t)b * 2 * world + 2 * a + 1]) {
          fprintf(stderr, "pastix_amd: ranks %d and %d disagree on the fan-in blocks %d sends to %d\n", a, b, a, b);
          return PASTIX_AMD_ERR_LAYOUT;
        }
  }
  for (int r = 0; r < world; r++) {
    pastix_amd_plan_t* p = plans[r];
    if (!p || !p->distributed || p->host.myrank != r) return PASTIX_AMD_ERR_BADPARAMETER;
    DistSchedule S;
    int rc = build_schedule(p->host, world, S);
    if (rc) return rc;
    std::unique_ptr<LocalTransport> T(new (std::nothrow) LocalTransport());
    if (!T) return PASTIX_AMD_ERR_ALLOC;
    T->hub = hub;
    T->me = r;
    if ((rc = dist_attach_common(p, world, std::move(T), std::move(S)))) return rc;
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_dist_info(const pastix_amd_plan_t* p, pastix_amd_dist_info_t* info) {
  if (!p || !info || !p->dist) return PASTIX_AMD_ERR_BADPARAMETER;
  const pastix_amd_dist_s* D = p->dist;
  std::memset(info, 0, sizeof(*info));
  info->world = D->S.world;
  info->rank = D->S.myrank;
  info->npeers = (int32_t)D->S.peers.size();
  info->nplanes = D->S.nplanes;
  info->nsend = D->nsend;
  info->nrecv = D->nrecv;
  info->bytes_sent = D->bytes_sent;
  info->bytes_recv = D->bytes_recv;
  info->staging_bytes = 8.0 * (double)D->S.stage_elems;
  double fb = 0;
  const Plan& H = p->host;
  for (int64_t k = 0; k < H.cblknbr; k++) if (H.role[k] == 2) fb += 8.0 * (double)(H.poff[k + 1] - H.poff[k]);
  info->fanin_buffer_bytes = fb;
  std::strncpy(info->transport, D->T->name(), sizeof(info->transport) - 1);
  return PASTIX_AMD_OK;
}

// ---- the end of a distributed run: wait with a deadline, or give up cleanly ---------------------------------------
// `fin`: an event behind everything the run enqueued (recorded on the panel stream after every channel and the second
// stream joined it); enqueue_rc: what the enqueueing loop returned.  Success: fin reached within the deadline.  Otherwise
// -- an error while enqueueing, a HIP error, or the deadline -- the rank says where its streams stand (first unmatched
// group per channel, last level the panel stream finished), ABORTS its channels (ncclCommAbort: operations peers have
// already enqueued against this rank fail instead of spinning), drains its streams and marks the plan's distributed
// state failed.  A one-rank failure thus becomes an error code on every rank within the deadline, not a job-wide hang.
static int dist_finish(pastix_amd_plan_t* p, int enqueue_rc, hipEvent_t fin, const char* what) {
  pastix_amd_dist_s* D = p->dist;
  const DistSchedule& S = D->S;
  int rc = enqueue_rc;
  bool timed_out = false;
  if (!rc) {
    const double t0 = now_s(), limit = dist_timeout_s();
    for (;;) {
      const hipError_t e = hipEventQuery(fin);
      if (e == hipSuccess) return PASTIX_AMD_OK;
      if (e != hipErrorNotReady) {
        fprintf(stderr, "pastix_amd[rank %d]: %s: HIP error '%s' while waiting\n", S.myrank, what, hipGetErrorString(e));
        rc = PASTIX_AMD_ERR_DEVICE;
        break;
      }
      if (now_s() - t0 > limit) { timed_out = true; rc = PASTIX_AMD_ERR_TIMEOUT; break; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  if (timed_out) {
    int last_level = -1;
    for (auto& lm : D->level_marks) {
      if (hipEventQuery(lm.first) != hipSuccess) break;
      last_level = lm.second;
    }
    fprintf(stderr, "pastix_amd[rank %d]: %s did not finish within %.0f s (PASTIX_AMD_DIST_TIMEOUT); panel stream finished "
            "level %d of %d\n", S.myrank, what, dist_timeout_s(), last_level, p->host.nlevels);
    std::map<int, bool> seen;
    for (auto& mk : D->marks) {
      if (seen[mk.peer]) continue;
      if (hipEventQuery(mk.ev) == hipSuccess) continue;
      seen[mk.peer] = true;
      const DistMsg& m = S.msgs[mk.m0];
      fprintf(stderr, "pastix_amd[rank %d]:   channel to rank %d: first unmatched group at level %d (%zu blocks), first block: "
              "cblk %d, %s, %lld x %lld x %d planes\n", S.myrank, mk.peer, mk.level, mk.m1 - mk.m0, m.cblk,
              m.dir == 0 ? "send" : "receive", (long long)m.nrows, (long long)m.width, S.nplanes);
    }
    if (seen.empty()) fprintf(stderr, "pastix_amd[rank %d]:   every channel group completed: the compute streams are stuck\n", S.myrank);
  } else {
    fprintf(stderr, "pastix_amd[rank %d]: %s failed (code %d) with work in flight: aborting the channels\n", S.myrank, what, rc);
  }
  D->failed = true;
  D->T->abort();
  // drain what is left (bounded: a stream that survives the abort is left to process exit)
  std::vector<hipStream_t> ss{p->stream, p->stream2};
  for (auto& c : D->chan) ss.push_back(c.second);
  const double t1 = now_s();
  for (hipStream_t st : ss) {
    if (!st) continue;
    while (hipStreamQuery(st) == hipErrorNotReady && now_s() - t1 < 10.0)
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  (void)hipGetLastError();
  p->factored = false;
  return rc;
}

// The numerical factorization of this rank's share (the device replacement of sopalin_smp + the communication thread,
// sopalin3d.c:790-1025, sopalin_sendrecv.c:2393-2775).  Everything is enqueued; the host waits only at the end, with a
// deadline (dist_finish).
int pastix_amd_factorize_dist(pastix_amd_plan_t* p, double critere, pastix_amd_stats_t* stats) {
  if (!p || !p->distributed || !p->dist) return PASTIX_AMD_ERR_BADPARAMETER;
  pastix_amd_dist_s* D = p->dist;
  if (D->failed) {
    fprintf(stderr, "pastix_amd[rank %d]: the channels of this plan were aborted by an earlier failure\n", D->S.myrank);
    return PASTIX_AMD_ERR_BADPARAMETER;
  }
  const DistSchedule& S = D->S;
  const Plan& H = p->host;
  int rc = pastix_amd_factorize_begin(p, critere);
  if (rc) return rc;
  hipStream_t s1 = p->stream;
  D->nev_used = 0;
  D->marks.clear();
  D->level_marks.clear();
  auto new_event = [&](hipEvent_t* out) -> int {
    if (D->nev_used == D->events.size()) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      D->events.push_back(e);
    }
    *out = D->events[D->nev_used++];
    return 0;
  };
  double* const arena[4] = {p->dL, p->dU, p->dLi, p->dUi};
  hipEvent_t fin = nullptr;
  auto enqueue = [&]() -> int {
    int rc = 0;
    size_t mi = 0;
    const size_t nm = S.msgs.size();
    for (int l = 0; l < H.nlevels; l++) {
      if ((rc = pastix_amd_factorize_level(p, l, 1))) return rc;            // A(l) on the panel stream, B(l) beside it
      const size_t m0 = mi;
      while (mi < nm && S.msgs[mi].level == l) mi++;
      if (mi > m0) {
        bool any_send = false;
        for (size_t i = m0; i < mi; i++) any_send |= S.msgs[i].dir == 0;
        hipEvent_t evA = nullptr;
        if (any_send) {
          if ((rc = new_event(&evA))) return rc;
          HIPCHK(hipEventRecord(evA, s1));                                    // this rank's blocks for level l are complete
        }
        for (size_t g0 = m0; g0 < mi;) {                                      // one group per peer
          size_t g1 = g0;
          const int peer = S.msgs[g0].peer;
          bool gs = false, gr = false;
          while (g1 < mi && S.msgs[g1].peer == peer) { (S.msgs[g1].dir == 0 ? gs : gr) = true; g1++; }
          hipStream_t cs = D->chan[peer];
          if (gs) HIPCHK(hipStreamWaitEvent(cs, evA, 0));
          if ((rc = D->T->group_begin(peer))) return rc;
          for (size_t i = g0; i < g1; i++) {
            const DistMsg& m = S.msgs[i];
            const int64_t cnt = m.nrows * m.width;
            for (int q = 0; q < S.nplanes; q++) {
              if (m.dir == 0) rc = D->T->send(peer, arena[S.planes[q]] + m.off, cnt, cs);
              else rc = D->T->recv(peer, D->dStage + m.off + q * cnt, cnt, cs);
              if (rc) { (void)D->T->group_end(peer); return rc; }
            }
          }
          if ((rc = D->T->group_end(peer))) return rc;
          hipEvent_t evC;                                                     // progress mark of the group; receives:
          if ((rc = new_event(&evC))) return rc;                              // what the panel stream waits for
          HIPCHK(hipEventRecord(evC, cs));
          D->marks.push_back({evC, l, peer, g0, g1});
          if (gr) HIPCHK(hipStreamWaitEvent(s1, evC, 0));                     // the panel stream needs these blocks now
          g0 = g1;
        }
        // recv_handle_fanin (sopalin_sendrecv.c:384-404): the owner ADDS the aggregated blocks, in a fixed order
        for (size_t i = m0; i < mi; i++) {
          const DistMsg& m = S.msgs[i];
          if (m.dir != 1) continue;
          const int64_t cnt = m.nrows * m.width;
          for (int q = 0; q < S.nplanes; q++)
            launch_fanin_add(s1, arena[S.planes[q]] + H.poff[m.cblk], H.cblk[m.cblk].stride, D->dStage + m.off + q * cnt,
                             D->dRows + m.rows_off, m.nrows, m.width);
        }
      }
      if ((rc = pastix_amd_factorize_level(p, l, 2))) return rc;            // P(l)
      if (mi > m0 || l + 1 == H.nlevels) {                                  // (a mark behind the levels that exchange)
        hipEvent_t evL;
        if ((rc = new_event(&evL))) return rc;
        HIPCHK(hipEventRecord(evL, s1));
        D->level_marks.emplace_back(evL, l);
      }
    }
    // the channels join the panel stream: every send has left before the caller may zero the fan-in buffers again
    for (auto& c : D->chan) {
      hipEvent_t e;
      if ((rc = new_event(&e))) return rc;
      HIPCHK(hipEventRecord(e, c.second));
      HIPCHK(hipStreamWaitEvent(s1, e, 0));
    }
    if (p->staged_overlap && p->stream2) {                                  // ... and so does the second stream
      hipEvent_t e;
      if ((rc = new_event(&e))) return rc;
      HIPCHK(hipEventRecord(e, p->stream2));
      HIPCHK(hipStreamWaitEvent(s1, e, 0));
    }
    if ((rc = new_event(&fin))) return rc;
    HIPCHK(hipEventRecord(fin, s1));
    return 0;
  };
  rc = enqueue();
  if ((rc = dist_finish(p, rc, fin, "pastix_amd_factorize_dist"))) return rc;
  return pastix_amd_factorize_end(p, stats);
}

}  // extern "C"

namespace {
__global__ void k_vec_add(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
}  // namespace

extern "C" {

// Triangular solves on the distributed factors (the data flow of up_down_smp with several processes, updo.c:114 and
// updo_sendrecv.c): every rank keeps a full-length copy of the vector.  Forward sweep: a rank's copy holds the
// right-hand side on the columns of its own cblks and ZERO elsewhere, so what its panels subtract from the rows of a
// remote cblk t accumulates there; at t's level that segment travels to t's owner, which adds it before it solves t --
// the fan-in of the factorization, on vectors (same channels, same (level, peer, cblk) order).  Backward sweep: the
// same messages run the other way: the owner of t sends the solved segment x_t to every rank that holds panel rows in
// t, before they reach the levels below.  x (host, permuted numbering, one right-hand side): in b, out the solved
// values on the columns this rank owns and zeros elsewhere -- the sum over the ranks is the solution.
int pastix_amd_solve_dist(pastix_amd_plan_t* p, double* x) {
  if (!p || !x || !p->distributed || !p->dist) return PASTIX_AMD_ERR_BADPARAMETER;
  if (!p->factored || p->dist->failed) return PASTIX_AMD_ERR_BADPARAMETER;
  // complex plans: x is the reference's interleaved `double complex` vector; on the device (and in the staging buffer)
  // the real and imaginary parts are two planes of n doubles, and every message carries both
  const int np_ = p->cplx ? 2 : 1;
  pastix_amd_dist_s* D = p->dist;
  const DistSchedule& S = D->S;
  const Plan& H = p->host;
  HIPCHK(hipSetDevice(p->device));
  int rc = pai_solve_tables(p);
  if (rc) return rc;
  const int64_t n = H.ncol;
  if (!D->dXs) HIPCHK(hipMalloc((void**)&D->dXs, (size_t)n * np_ * sizeof(double)));
  if (D->solve_stage_off.empty() && !S.msgs.empty()) {
    D->solve_stage_off.assign(S.msgs.size(), 0);
    int64_t off = 0;
    for (size_t i = 0; i < S.msgs.size(); i++)
      if (S.msgs[i].dir == 1) { D->solve_stage_off[i] = off; off += S.msgs[i].width * np_; }
    HIPCHK(hipMalloc((void**)&D->dSolveStage, (size_t)std::max<int64_t>(off, 1) * sizeof(double)));
  }
  // the rank's view of b: own columns only.  The staging buffer belongs to the plan's distributed state, not to this
  // call: dist_finish's drain is bounded, so on the abandoned-run path an enqueued copy may still be in flight when this
  // function returns -- it must not target memory that dies with the call.
  if (D->nXs < (size_t)n * np_) {
    if (D->hXs) { HIPCHK(hipDeviceSynchronize()); (void)hipHostFree(D->hXs); D->hXs = nullptr; D->nXs = 0; }
    HIPCHK(hipHostMalloc((void**)&D->hXs, (size_t)n * np_ * sizeof(double), hipHostMallocDefault));
    D->nXs = (size_t)n * np_;
  }
  struct { double* p; double* data() const { return p; } } hx{D->hXs};
  std::memset(hx.data(), 0, (size_t)n * np_ * sizeof(double));
  for (int64_t k = 0; k < H.cblknbr; k++)
    if (H.role[k] == 1 && p->cplx) {
      for (int64_t j = H.cblk[k].fcolnum; j <= H.cblk[k].lcolnum; j++) { hx.data()[j] = x[2 * j]; hx.data()[n + j] = x[2 * j + 1]; }
    } else if (H.role[k] == 1)
      std::memcpy(hx.data() + H.cblk[k].fcolnum, x + H.cblk[k].fcolnum,
                  (size_t)(H.cblk[k].lcolnum - H.cblk[k].fcolnum + 1) * sizeof(double));
  hipStream_t s1 = p->stream;
  double* dx = D->dXs;
  D->nev_used = 0;
  D->marks.clear();
  D->level_marks.clear();
  auto new_event = [&](hipEvent_t* out) -> int {
    if (D->nev_used == D->events.size()) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      D->events.push_back(e);
    }
    *out = D->events[D->nev_used++];
    return 0;
  };
  // one sweep over the levels; fwd: contributors send, owners receive and add; bwd: owners send, contributors receive
  auto exchange = [&](int l, bool fwd, size_t m0, size_t m1) -> int {
    if (m1 <= m0) return 0;
    hipEvent_t evA;
    int r;
    if ((r = new_event(&evA))) return r;
    HIPCHK(hipEventRecord(evA, s1));                      // everything this rank computed so far
    for (size_t g0 = m0; g0 < m1;) {
      size_t g1 = g0;
      const int peer = S.msgs[g0].peer;
      while (g1 < m1 && S.msgs[g1].peer == peer) g1++;
      hipStream_t cs = D->chan[peer];
      HIPCHK(hipStreamWaitEvent(cs, evA, 0));
      if ((r = D->T->group_begin(peer))) return r;
      bool gr = false;
      for (size_t i = g0; i < g1; i++) {
        const DistMsg& m = S.msgs[i];
        const bool send = fwd ? m.dir == 0 : m.dir == 1;
        for (int pl = 0; pl < np_; pl++) {                 // (complex: the real plane, then the imaginary one)
          double* seg = dx + (int64_t)pl * n + H.cblk[m.cblk].fcolnum;
          if (send) r = D->T->send(peer, seg, m.width, cs);
          else { r = D->T->recv(peer, fwd ? D->dSolveStage + D->solve_stage_off[i] + (int64_t)pl * m.width : seg, m.width, cs); gr = true; }
          if (r) { (void)D->T->group_end(peer); return r; }
        }
      }
      if ((r = D->T->group_end(peer))) return r;
      hipEvent_t evC;
      if ((r = new_event(&evC))) return r;
      HIPCHK(hipEventRecord(evC, cs));
      D->marks.push_back({evC, l, peer, g0, g1});
      if (gr) HIPCHK(hipStreamWaitEvent(s1, evC, 0));
      g0 = g1;
    }
    if (fwd)
      for (size_t i = m0; i < m1; i++) {
        const DistMsg& m = S.msgs[i];
        if (m.dir != 1) continue;
        for (int pl = 0; pl < np_; pl++)
          hipLaunchKernelGGL(k_vec_add, dim3((unsigned)((m.width + 255) / 256)), dim3(256), 0, s1,
                             dx + (int64_t)pl * n + H.cblk[m.cblk].fcolnum, D->dSolveStage + D->solve_stage_off[i] + (int64_t)pl * m.width,
                             m.width);
      }
    return 0;
  };
  std::vector<size_t> lvl_m((size_t)H.nlevels + 1, S.msgs.size());
  {
    size_t mi = 0;
    for (int l = 0; l < H.nlevels; l++) {
      lvl_m[(size_t)l] = mi;
      while (mi < S.msgs.size() && S.msgs[mi].level == l) mi++;
    }
    lvl_m[(size_t)H.nlevels] = mi;
  }
  hipEvent_t fin = nullptr;
  auto enqueue = [&]() -> int {
    int rc = 0;
    HIPCHK(hipMemcpyAsync(dx, hx.data(), (size_t)n * np_ * sizeof(double), hipMemcpyHostToDevice, s1));
    auto level = [&](bool fwd, int l) {
      if (!p->cplx) { pai_solve_level(p, fwd, l, dx, 1); return; }
      launch_zsolve_level(s1, fwd, H.factotype, p->arenas(), p->dSolve + H.lvl_cblk_ptr[l], H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l],
                          fwd ? p->dChunk + p->lvl_chunk_ptr[l] : p->dChunkB + p->lvl_chunkB_ptr[l],
                          fwd ? p->lvl_chunk_ptr[l + 1] - p->lvl_chunk_ptr[l] : p->lvl_chunkB_ptr[l + 1] - p->lvl_chunkB_ptr[l], p->dBlok,
                          p->dRidx, dx, dx + n, p->maxw);
    };
    for (int l = 0; l < H.nlevels; l++) {
      if ((rc = exchange(l, true, lvl_m[(size_t)l], lvl_m[(size_t)l + 1]))) return rc;
      level(true, l);
    }
    if (p->cplx) {
      if (H.factotype == PASTIX_AMD_FACT_LDLT || H.factotype == PASTIX_AMD_FACT_LDLH)
        launch_zsolve_dscale(s1, p->arenas(), p->dSolve, H.lvl_cblk_ptr[H.nlevels], dx, dx + n);
    } else if (H.factotype == PASTIX_AMD_FACT_LDLT) pai_solve_dscale(p, dx, 1);
    for (int l = H.nlevels - 1; l >= 0; l--) {
      level(false, l);
      if ((rc = exchange(l, false, lvl_m[(size_t)l], lvl_m[(size_t)l + 1]))) return rc;
    }
    for (auto& c : D->chan) {                               // sends of the last levels have left
      hipEvent_t e;
      if ((rc = new_event(&e))) return rc;
      HIPCHK(hipEventRecord(e, c.second));
      HIPCHK(hipStreamWaitEvent(s1, e, 0));
    }
    HIPCHK(hipMemcpyAsync(hx.data(), dx, (size_t)n * np_ * sizeof(double), hipMemcpyDeviceToHost, s1));
    if ((rc = new_event(&fin))) return rc;
    HIPCHK(hipEventRecord(fin, s1));
    return 0;
  };
  rc = enqueue();
  if ((rc = dist_finish(p, rc, fin, "pastix_amd_solve_dist"))) { p->factored = true; return rc; }
  HIPCHK(hipStreamSynchronize(s1));
  HIPCHK(hipGetLastError());
  std::memset(x, 0, (size_t)n * np_ * sizeof(double));
  for (int64_t k = 0; k < H.cblknbr; k++)
    if (H.role[k] == 1 && p->cplx) {
      for (int64_t j = H.cblk[k].fcolnum; j <= H.cblk[k].lcolnum; j++) { x[2 * j] = hx.data()[j]; x[2 * j + 1] = hx.data()[n + j]; }
    } else if (H.role[k] == 1)
      std::memcpy(x + H.cblk[k].fcolnum, hx.data() + H.cblk[k].fcolnum,
                  (size_t)(H.cblk[k].lcolnum - H.cblk[k].fcolnum + 1) * sizeof(double));
  return PASTIX_AMD_OK;
}

// drives pastix_amd_solve_dist of the plans of pastix_amd_dist_attach_local with one host thread per rank;
// xs[r]: rank r's vector (each a full copy of b on entry)
int pastix_amd_solve_dist_local(pastix_amd_plan_t* const* plans, int32_t world, double* const* xs) {
  if (!plans || !xs || world < 2) return PASTIX_AMD_ERR_BADPARAMETER;
  std::vector<int> rc((size_t)world, 0);
  std::vector<std::thread> th;
  auto run = [&](int r) {
    rc[(size_t)r] = pastix_amd_solve_dist(plans[r], xs[r]);
    if (rc[(size_t)r])
      if (auto* lt = dynamic_cast<LocalTransport*>(plans[r]->dist ? plans[r]->dist->T.get() : nullptr)) {
        { std::lock_guard<std::mutex> g(lt->hub->mu); lt->hub->failed = true; }
        lt->hub->cv.notify_all();
      }
  };
  for (int r = 1; r < world; r++) th.emplace_back(run, r);
  run(0);
  for (auto& t : th) t.join();
  for (int r = 0; r < world; r++) if (rc[(size_t)r]) return rc[(size_t)r];
  return PASTIX_AMD_OK;
}

// Single-process emulation of a whole job: the rank plans (attached with pastix_amd_dist_attach_local) are driven by
// one host thread each, exactly as separate processes would drive them.  rcs[r] / stats[r] per rank (may be NULL).
int pastix_amd_factorize_dist_local(pastix_amd_plan_t* const* plans, int32_t world, double critere,
                                    pastix_amd_stats_t* stats, int32_t* rcs) {
  if (!plans || world < 2) return PASTIX_AMD_ERR_BADPARAMETER;
  for (int r = 0; r < world; r++)
    if (!plans[r] || !plans[r]->dist) return PASTIX_AMD_ERR_BADPARAMETER;
  std::vector<int> rc((size_t)world, 0);
  std::vector<std::thread> th;
  auto run = [&](int r) {
    try {
      rc[(size_t)r] = pastix_amd_factorize_dist(plans[r], critere, stats ? stats + r : nullptr);
    } catch (...) {
      rc[(size_t)r] = PASTIX_AMD_ERR_ALLOC;
    }
    if (rc[(size_t)r] && rc[(size_t)r] != PASTIX_AMD_ERR_NUMERIC) {
      // release peers blocked in a loopback receive
      if (auto* lt = dynamic_cast<LocalTransport*>(plans[r]->dist->T.get())) {
        { std::lock_guard<std::mutex> g(lt->hub->mu); lt->hub->failed = true; }
        lt->hub->cv.notify_all();
      }
    }
  };
  try {
    for (int r = 1; r < world; r++) th.emplace_back(run, r);
  } catch (...) {
    for (auto& t : th) t.join();
    return PASTIX_AMD_ERR_ALLOC;
  }
  run(0);
  for (auto& t : th) t.join();
  int first = 0;
  for (int r = 0; r < world; r++) {
    if (rcs) rcs[r] = rc[(size_t)r];
    if (!first && rc[(size_t)r]) first = rc[(size_t)r];
  }
  return first;
}

}  // extern "C"
