// api.cpp -- the C ABI declared in include/pastix_amd.h: plan life cycle, panel transfers, the device
// factorization driver (the GPU replacement of sopalin_thread / sopalin_smp, sopalin3d.c:666-1422).
#include <hip/hip_runtime.h>
#include <sched.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <new>
#include <system_error>
#include <thread>
#include <atomic>
#include <mutex>
#include <vector>

#include "engine.h"

extern "C" {

const char* pastix_amd_version(void) { return "pastix_amd 0.1 (gfx950)"; }

double pastix_amd_fact_flops(const pastix_amd_layout_t* layout, int factotype, int floattype) {
  if (!layout) return 0.0;
  return fact_flops(layout, factotype, floattype);
}

// ------------------------------------------------------------------------------------------------
// cblks wider than SPLITW = 128 columns: the engine factorizes an equivalent layout in which every such cblk is cut into
// column groups of 128 columns and a remainder.  blend's splitOnProcs (splitpart.c:431-475) cuts a supernode into pieces
// of width in [max, 2 max) -- 144 columns at its defaults, 152 with MAX_BLOCKSIZE 128, up to 239 -- and leaves some
// supernodes whole (334 columns on the reference's own orsirr.rua fixture): with the re-cut every layout the real PaStiX
// hands over runs on the kernels built for <= 128 columns (LDS-resident diagonal blok, whole 128-column update tiles),
// and there are no others.  Sub-cblk j keeps the columns [c_j, c_j+1) and the panel rows from its own
// diagonal blok down: the rest of the original diagonal blok becomes off-diagonal bloks facing the later
// sub-cblks, and bloks facing a split cblk are cut at its column groups.  Mathematically the blocked algorithm
// on the wide blok; only the host <-> device panel copies see the difference (split_io below).
// ------------------------------------------------------------------------------------------------

static constexpr int SPLITW = 128;

static int build_split(const pastix_amd_layout_t* L, bool schur, SplitMap& M) {
  const int64_t nc = L->cblknbr;
  M.ocblknbr = nc;
  M.first.assign((size_t)nc + 1, 0);
  M.owidth.resize((size_t)nc);
  M.ostride.resize((size_t)nc);
  M.ooff.assign((size_t)nc + 1, 0);
  bool any = false;
  for (int64_t k = 0; k < nc; k++) {
    const int64_t w = L->cblktab[k].lcolnum - L->cblktab[k].fcolnum + 1;
    if (w <= 0 || L->cblktab[k].stride < w) return PASTIX_AMD_ERR_LAYOUT;
    M.owidth[k] = w;
    M.ostride[k] = L->cblktab[k].stride;
    M.ooff[k + 1] = M.ooff[k] + w * M.ostride[k];
    int64_t ns = 1;
    if (w > SPLITW && !(schur && k == nc - 1)) { ns = (w + SPLITW - 1) / SPLITW; any = true; }
    M.first[k + 1] = M.first[k] + ns;
  }
  M.active = any;
  if (!any) return 0;
  // first column of sub j of cblk k: groups of SPLITW columns, the last one takes what is left (whole 128-column
  // update tiles for all but one group; even widths -- 76 + 76 for 152 -- would make every tile an edge tile)
  auto subcol = [&](int64_t k, int64_t j) {
    const int64_t ns = M.first[k + 1] - M.first[k];
    return L->cblktab[k].fcolnum + (j >= ns ? M.owidth[k] : j * (ns == 1 ? M.owidth[k] : (int64_t)SPLITW));
  };
  M.cblk.clear();
  M.blok.clear();
  for (int64_t k = 0; k < nc; k++) {
    const int64_t ns = M.first[k + 1] - M.first[k];
    const int64_t fb = L->cblktab[k].bloknum, lb = L->cblktab[k + 1].bloknum;
    if (lb <= fb) return PASTIX_AMD_ERR_LAYOUT;
    for (int64_t j = 0; j < ns; j++) {
      pastix_amd_cblk_t c;
      c.fcolnum = subcol(k, j);
      c.lcolnum = subcol(k, j + 1) - 1;
      c.bloknum = (pastix_amd_int_t)M.blok.size();
      int64_t coef = 0;
      for (int64_t j2 = j; j2 < ns; j2++) {                       // own diagonal blok, then the later column groups
        pastix_amd_blok_t b{subcol(k, j2), subcol(k, j2 + 1) - 1, M.first[k] + j2, coef};
        coef += b.lrownum - b.frownum + 1;
        M.blok.push_back(b);
      }
      for (int64_t q = fb + 1; q < lb; q++) {                     // the original off-diagonal bloks
        const pastix_amd_blok_t& ob = L->bloktab[q];
        const int64_t t = ob.cblknum;
        if (t <= k || t >= nc) return PASTIX_AMD_ERR_LAYOUT;
        const int64_t nt = M.first[t + 1] - M.first[t];
        for (int64_t s2 = 0; s2 < nt; s2++) {
          const int64_t lo = std::max<int64_t>(ob.frownum, subcol(t, s2));
          const int64_t hi = std::min<int64_t>(ob.lrownum, subcol(t, s2 + 1) - 1);
          if (lo > hi) continue;
          pastix_amd_blok_t b{lo, hi, M.first[t] + s2, coef};
          coef += hi - lo + 1;
          M.blok.push_back(b);
        }
      }
      c.stride = coef;
      M.cblk.push_back(c);
    }
  }
  pastix_amd_cblk_t last = L->cblktab[nc];
  last.bloknum = (pastix_amd_int_t)M.blok.size();
  M.cblk.push_back(last);
  return 0;
}

static int plan_create_common(const pastix_amd_layout_t* layout, int factotype, int floattype,
                              const pastix_amd_options_t* opts, const int32_t* owner, int32_t myrank,
                              pastix_amd_plan_t** out) {
  if (!out) return PASTIX_AMD_ERR_BADPARAMETER;
  *out = nullptr;
  pastix_amd_plan_s* p = new (std::nothrow) pastix_amd_plan_s();
  if (!p) return PASTIX_AMD_ERR_ALLOC;
  int rc;
  double oflops = 0;
  const bool ptime = dev_opt("plan_timing") != nullptr;
  double tph = now_s();
  auto phase = [&](const char* name) { if (ptime) { const double t = now_s(); fprintf(stderr, "[create] %-28s %.2f s\n", name, t - tph); tph = t; } };
  struct { std::thread th; int rc = 0, n = 0, device = 0; size_t bytes = 0; char* raw[4] = {nullptr, nullptr, nullptr, nullptr}; } pre;
  try {
    if (!layout || !layout->cblktab || !layout->bloktab || layout->cblknbr < 1) { delete p; return PASTIX_AMD_ERR_BADPARAMETER; }
    rc = build_split(layout, opts && opts->schur, p->split);
    pastix_amd_layout_t sl;
    std::vector<int32_t> sowner;
    if (!rc && p->split.active) {
      if (owner) {                                         // the column groups of a cblk stay with its owner
        sowner.resize(p->split.cblk.size() - 1);
        for (int64_t k = 0; k < p->split.ocblknbr; k++)
          for (int64_t q = p->split.first[k]; q < p->split.first[k + 1]; q++) sowner[(size_t)q] = owner[k];
        owner = sowner.data();
      }
      sl.cblknbr = (pastix_amd_int_t)p->split.cblk.size() - 1;
      sl.bloknbr = (pastix_amd_int_t)p->split.blok.size();
      sl.cblktab = p->split.cblk.data();
      sl.bloktab = p->split.blok.data();
      oflops = fact_flops(layout, factotype, floattype);
    }
    if (!rc && !owner && !(opts && opts->external_arena)) {
      // one GPU, library-owned panels: their size is known from the layout alone, so the device allocation (0.6 s for the
      // 150 GB of 200^3, more on a box that has just released memory) runs on a thread of its own beside the host plan
      const pastix_amd_layout_t* L = p->split.active ? &sl : layout;
      int64_t coef = 0;
      for (int64_t k = 0; k < L->cblknbr; k++)
        coef += L->cblktab[k].stride * (L->cblktab[k].lcolnum - L->cblktab[k].fcolnum + 1);
      const size_t esz = floattype == PASTIX_AMD_REALSINGLE ? sizeof(float) : sizeof(double);
      pre.bytes = (size_t)std::max<int64_t>(coef, 1) * esz;
      pre.n = (factotype != PASTIX_AMD_FACT_LLT ? 2 : 1) * (floattype == PASTIX_AMD_COMPLEXDOUBLE ? 2 : 1);
      pre.device = opts ? opts->device : 0;
      pre.th = std::thread([&pre] {
        if (hipSetDevice(pre.device) != hipSuccess) { pre.rc = PASTIX_AMD_ERR_DEVICE; return; }
        for (int i = 0; i < pre.n; i++) {
          if (hipMalloc((void**)&pre.raw[i], pre.bytes + 2 * ARENA_PAD) != hipSuccess) { pre.raw[i] = nullptr; pre.rc = PASTIX_AMD_ERR_ALLOC; return; }
          if (hipMemset(pre.raw[i], 0, ARENA_PAD) != hipSuccess || hipMemset(pre.raw[i] + ARENA_PAD + pre.bytes, 0, ARENA_PAD) != hipSuccess) {
            pre.rc = PASTIX_AMD_ERR_DEVICE;
            return;
          }
        }
      });
    }
    // (the reader lists of the run schedule are built on the device, run_edges.hip; the host builds them for the host-only
    // entry points, for the stopped-run replay and on request: PASTIX_AMD_DEV=run_host_edges)
    p->host.defer_run_edges = !owner && !dev_opt("run_debug") && !dev_opt("run_host_edges");
    if (!rc) rc = build_plan(p->split.active ? &sl : layout, factotype, floattype, opts, owner, myrank, p->host);
  } catch (const std::bad_alloc&) {
    rc = PASTIX_AMD_ERR_ALLOC;
  } catch (const std::system_error&) {
    rc = PASTIX_AMD_ERR_ALLOC;
  }
  if (pre.th.joinable()) pre.th.join();
  auto pre_free = [&] {
    for (int i = 0; i < 4; i++) { if (pre.raw[i]) (void)hipFree(pre.raw[i]); pre.raw[i] = nullptr; }
    (void)hipGetLastError();
  };
  if (rc) { pre_free(); delete p; return rc; }
  phase("re-cut + host plan");
  if (p->split.active) p->host.fact_flops = p->host.local_flops = oflops;   // DPARM_FACT_FLOPS is the caller's layout's
  if (p->split.active && p->host.factotype != PASTIX_AMD_FACT_LU) {
    SplitMap& M = p->split;
    try {
      M.upper.resize((size_t)M.ocblknbr);
      const size_t eb = floattype == PASTIX_AMD_COMPLEXDOUBLE ? 16 : floattype == PASTIX_AMD_REALSINGLE ? 4 : 8;
      for (int64_t k = 0; k < M.ocblknbr; k++)
        if (M.first[k + 1] - M.first[k] > 1 && p->host.role[(size_t)M.first[k]] == 1)
          M.upper[(size_t)k].assign((size_t)(M.owidth[k] * M.owidth[k]) * eb, 0);
    } catch (const std::bad_alloc&) { pre_free(); delete p; return PASTIX_AMD_ERR_ALLOC; }
  }
  Plan& H = p->host;
  p->device = H.opts.device;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= p->device) {
    fprintf(stderr, "pastix_amd: no HIP device %d (the product path has no CPU fallback)\n", p->device);
    pre_free();
    delete p;
    return PASTIX_AMD_ERR_DEVICE;
  }
#define CHK(x) do { int r_ = (x); if (r_) { pre_free(); pastix_amd_plan_destroy(p); return r_; } } while (0)
  auto body = [&]() -> int {
    HIPCHK(hipSetDevice(p->device));
    {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically smallest = highest priority
      HIPCHK(hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, hi));
      HIPCHK(hipStreamCreateWithPriority(&p->stream2, hipStreamNonBlocking, lo));
    }
    p->distributed = owner != nullptr;
    p->own_arena = !(opts && opts->external_arena);
    p->cplx = H.floattype == PASTIX_AMD_COMPLEXDOUBLE;
    p->f32 = H.floattype == PASTIX_AMD_REALSINGLE;
    p->esz = p->f32 ? sizeof(float) : sizeof(double);
    if (p->f32 && !p->own_arena) return PASTIX_AMD_ERR_UNSUPPORTED;
    if (p->cplx && !p->own_arena) return PASTIX_AMD_ERR_UNSUPPORTED;
    if (p->own_arena) {
      const size_t bytes = std::max<int64_t>(H.coefnbr, 1) * p->esz;
      // 256 B of slack on both sides: the update kernel's 16-byte DMA lanes may touch the element just
      // before / after a panel when a contribution starts or ends on an odd row (kernels.hip, k_update)
      auto alloc = [&](double** out) -> int {
        char* raw = nullptr;
        HIPCHK(hipMalloc((void**)&raw, bytes + 2 * ARENA_PAD));
        HIPCHK(hipMemset(raw, 0, ARENA_PAD));
        HIPCHK(hipMemset(raw + ARENA_PAD + bytes, 0, ARENA_PAD));
        *out = (double*)(raw + ARENA_PAD);
        return 0;
      };
      // (the arenas allocated beside the host plan, when they are what is needed; else they are released and made here)
      int npre = 0;
      const bool use_pre = pre.rc == 0 && pre.n > 0 && pre.bytes == bytes && p->device == pre.device;
      if (!use_pre) pre_free();
      auto take = [&](double** out) -> int {
        if (!use_pre) return alloc(out);
        *out = (double*)(pre.raw[npre] + ARENA_PAD);
        pre.raw[npre++] = nullptr;
        return 0;
      };
      int ra;
      if ((ra = take(&p->dL))) return ra;
      if (H.factotype != PASTIX_AMD_FACT_LLT && (ra = take(&p->dU))) return ra;
      if (p->cplx) {
        if ((ra = take(&p->dLi))) return ra;
        if (H.factotype != PASTIX_AMD_FACT_LLT && (ra = take(&p->dUi))) return ra;
      }
    }
    phase("streams + arenas");
    HIPCHK(hipMalloc((void**)&p->dDinv, std::max<int64_t>(H.dinv_ws, 256) * sizeof(double)));
    int r;
    if ((r = to_device(&p->dTasks, H.tasks))) return r;
    if ((r = to_device(&p->dPieces, H.pieces))) return r;
    if (!H.gmaps.empty() && (r = to_device(&p->dGmap, H.gmaps))) return r;
    phase("task + piece tables -> device");
    if ((r = to_device(&p->dPanel, H.panel_tasks))) return r;
    if ((r = to_device(&p->dTrsm, H.trsm_tasks))) return r;
    HIPCHK(hipMalloc((void**)&p->dNbpivot, 2 * sizeof(long long)));   // [static pivots, positive D entries]
    HIPCHK(hipMalloc((void**)&p->dErr, sizeof(int)));
    HIPCHK(hipEventCreate(&p->ev0));
    HIPCHK(hipEventCreate(&p->ev1));
    p->ev.resize(2 * (size_t)H.nlevels);
    for (auto& e : p->ev) HIPCHK(hipEventCreate(&e));
    p->evT.resize(2 * (size_t)H.nlevels);
    for (auto& e : p->evT) HIPCHK(hipEventCreate(&e));
    p->evP.resize((size_t)H.nlevels);
    p->evB.resize((size_t)H.nlevels);
    for (auto& e : p->evP) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : p->evB) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (H.run_L0 >= 0 && !H.run_tasks.empty()) {
      // the run schedule: its tables, its flags, two more streams for the resident panel kernels
      if ((r = to_device(&p->dRunTasks, H.run_tasks))) return r;
      if ((r = to_device(&p->dRunInfo, H.run_info))) return r;
      if (H.run_edges_deferred) {
        // the reader lists, the panel-solve tickets' offsets into them and the readers' counters: on the device, from the
        // tables just uploaded (run_edges.hip)
        size_t cn = 0;
        if ((r = run_edges_device(p->stream, p->dRunTasks, p->dRunInfo, p->dPieces, H, &p->dRunCons, &cn))) return r;
        p->nRunCons = cn;
        const size_t nr0 = H.run_tasks.size();
        H.run_ready.clear();
        for (size_t i = 0; i < nr0; i++) if (H.run_dep[i] == 0) H.run_ready.push_back((int32_t)i);
        phase("run: reader lists on the device");
      } else {
        if ((r = to_device(&p->dRunCons, H.run_cons))) return r;
        p->nRunCons = H.run_cons.size();
      }
      p->runDep0.assign(H.run_dep.begin(), H.run_dep.begin() + (ptrdiff_t)H.run_tasks.size());
      if ((r = to_device(&p->dRunD, H.run_d))) return r;
      // the run's state: [counters of the tickets and diagonal tasks | ticket ring | diagonal ring | control words] (a ring slot
      // is one 128-byte line, plan.h RUN_SLOT),
      // and its initial image (what is ready when the run starts sits in the rings, their tails behind it)
      const size_t nr = H.run_tasks.size(), nd = H.run_d.size();
      auto up64 = [](size_t x) { return (x + 63) / 64 * 64; };
      const size_t o_cnt = 0, o_q = up64(nr + nd), o_qd = o_q + (nr + nd) * RUN_SLOT, o_ctl = o_qd + nd * RUN_SLOT;
      p->nRunState = o_ctl + RUN_CTL_INTS;
      // one kernel (the diagonal tasks are tickets of k_run_update: real LLt / LDLt) unless PASTIX_AMD_RUN_ONEK=0
      {
        const char* er = dev_opt("room");                    // (72: see run_sync.h run_pop)
        p->runctl.room = er ? atoi(er) : 72;
        const char* e = dev_opt("onek");
        p->runctl.onek = (!p->cplx && (H.factotype == PASTIX_AMD_FACT_LLT || H.factotype == PASTIX_AMD_FACT_LDLT) && !(e && atoi(e) == 0)) ? 1 : 0;
      }
      {
        // (the image is a GB at 200^3 -- 8 M ring slots of one 128-byte line each --: it is made on the device: -1 everywhere,
        // the counters, the control words, and the few thousand tasks that are ready at the start scattered into their slots)
        HIPCHK(hipMalloc((void**)&p->dRunImage, p->nRunState * sizeof(int32_t)));
        HIPCHK(hipMalloc((void**)&p->dRunState, p->nRunState * sizeof(int32_t)));
        HIPCHK(hipMemsetAsync(p->dRunImage, 0xff, p->nRunState * sizeof(int32_t), p->stream));
        HIPCHK(hipMemcpyAsync(p->dRunImage + o_cnt, H.run_dep.data(), H.run_dep.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
        std::vector<int32_t> ctl((size_t)RUN_CTL_INTS, 0), rq, rqd;
        if (p->runctl.onek) {
          // (the diagonal tasks that are ready at the start come first: they head the chains)
          for (size_t i = 0; i < H.run_dready.size(); i++) rq.push_back((int32_t)nr + H.run_dready[i]);
          rq.insert(rq.end(), H.run_ready.begin(), H.run_ready.end());
        } else {
          rq = H.run_ready;
          rqd = H.run_dready;
          ctl[RUN_TAIL + 64] = (int32_t)rqd.size();
        }
        ctl[RUN_TAIL] = (int32_t)rq.size();
        HIPCHK(hipMemcpyAsync(p->dRunImage + o_ctl, ctl.data(), ctl.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
        for (int w = 0; w < 2; w++) {
          const std::vector<int32_t>& v = w ? rqd : rq;
          if (v.empty()) continue;
          int32_t* dv = nullptr;
          HIPCHK(hipMalloc((void**)&dv, v.size() * sizeof(int32_t)));
          HIPCHK(hipMemcpyAsync(dv, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
          launch_ring_scatter(p->stream, p->dRunImage + (w ? o_qd : o_q), dv, v.size());
          HIPCHK(hipStreamSynchronize(p->stream));
          HIPCHK(hipFree(dv));
        }
        HIPCHK(hipStreamSynchronize(p->stream));
      }
      p->runctl.cnt = p->dRunState + o_cnt;
      p->runctl.q = p->dRunState + o_q;
      p->runctl.qd = p->dRunState + o_qd;
      p->runctl.ctl = p->dRunState + o_ctl;
      p->runctl.nd = (int32_t)nd;
      p->runctl.nticket = (int32_t)nr;
      p->run_nd = (int64_t)nd;
      {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, p->device));
        p->run_nwg = 2 * std::max(prop.multiProcessorCount, 1);
      }
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      HIPCHK(hipStreamCreateWithPriority(&p->stream3, hipStreamNonBlocking, hi));
      HIPCHK(hipHostMalloc((void**)&p->hResident, 64, hipHostMallocCoherent | hipHostMallocMapped));
      *p->hResident = 0;
      HIPCHK(hipEventCreateWithFlags(&p->evZ, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&p->evS3, hipEventDisableTiming));
      p->run_nticket = (int64_t)H.run_tasks.size();
      if (dev_opt("run_prof") || dev_opt("run_debug")) {
        p->nRunProf = 4 * (H.run_tasks.size() + H.run_d.size());
        HIPCHK(hipMalloc((void**)&p->dRunProf, p->nRunProf * sizeof(long long)));
        HIPCHK(hipMemset(p->dRunProf, 0, p->nRunProf * sizeof(long long)));
        // per ticket, for tools/run_fit.py: flops, chunks on the branch-free loop, chunks of whole pieces of a smaller tile,
        // chunks of partial pieces, the same two weighted by the busiest wave's share of its eight sub-tiles, pieces, tm * tn
        std::vector<long long>& ft = p->runFeat;
        ft.assign(8 * H.run_tasks.size(), 0);
        for (size_t i = 0; i < H.run_tasks.size(); i++) {
          if (H.run_info[i].kind & 4) continue;
          const Task& tk = H.run_tasks[i];
          const bool fullt = tk.tm == TM && tk.tn == TN;
          long long* o = &ft[8 * i];
          for (int z = 0; z < tk.pn; z++) {
            const Piece& pc = H.pieces[(size_t)tk.p0 + (size_t)z];
            const long long c = (pc.k + 15) / 16;
            o[0] += 2LL * pc.m * pc.n * pc.k;
            int busiest = 0;                             // sub-tiles of the busiest wave (of 8)
            for (int w = 0; w < 8; w++) {
              int nr = 0, nc = 0;
              for (int q = 0; q < 2; q++) { const int r0 = (w >> 1) * 16 + 64 * q; nr += r0 < pc.dr + pc.m && r0 + 16 > pc.dr; }
              for (int q = 0; q < 4; q++) { const int c0 = (w & 1) * 16 + 32 * q; nc += c0 < pc.dc + pc.n && c0 + 16 > pc.dc; }
              busiest = std::max(busiest, nr * nc);
            }
            if (z < (int)tk.nfull && fullt) o[1] += c;
            else if (z < (int)tk.nfull && (int)tk.nfull == tk.pn) { o[2] += c; o[4] += c * busiest; }
            else { o[3] += c; o[5] += c * busiest; }
          }
          o[6] = tk.pn;
          o[7] = (long long)tk.tm * tk.tn;
        }
      }
      p->runctl.prof = p->dRunProf;
      p->run_ready = true;
    }
    return 0;
  };
  CHK(body());
#undef CHK
  phase("panel tables, events");
  for (int64_t k = 0; k < H.cblknbr - (H.opts.schur ? 1 : 0); k++)     // (the Schur cblk is never factorized)
    p->maxw = std::max<int>(p->maxw, (int)(H.cblk[k].lcolnum - H.cblk[k].fcolnum + 1));
  pastix_amd_stats_t& S = p->stats;
  S.fact_flops = H.fact_flops;
  S.local_flops = H.local_flops;
  S.coefnbr = H.coefnbr;
  S.nlevels = H.nlevels;
  S.ntasks = (int64_t)H.tasks.size();
  S.nquadrant_tasks = 0;
  for (const Task& t : H.tasks) if (t.flags & 32u) S.nquadrant_tasks += 1.0;
  S.npieces = (int64_t)H.pieces.size();
  S.update_flops = H.update_flops;
  S.update_bytes = H.update_bytes;
  S.full_flops = H.full_flops;
  S.urgent_flops = H.urgent_flops;
  // developer aid (tools/replay_slot.hip): PASTIX_AMD_DUMP_SLOT=<slot>[:file] writes the bulk tasks of one launch slot
  // with their pieces, to replay that launch alone under different task orders / kernel variants
  if (const char* ds = dev_opt("dump_slot")) {
    const int sl = atoi(ds);
    const char* fn = strchr(ds, ':') ? strchr(ds, ':') + 1 : "/tmp/pastix_amd_slot.bin";
    if (sl >= 0 && sl < H.nlevels) {
      const int64_t t0 = H.slot_urgent_end[sl], t1 = H.slot_task_ptr[sl + 1];
      std::vector<Task> tk(H.tasks.begin() + t0, H.tasks.begin() + t1);
      std::vector<Piece> pc;
      for (Task& t : tk) {
        const int32_t np0 = (int32_t)pc.size();
        pc.insert(pc.end(), H.pieces.begin() + t.p0, H.pieces.begin() + t.p0 + t.pn);
        t.p0 = np0;
      }
      if (FILE* f = fopen(fn, "wb")) {
        const int64_t hdr[4] = {H.coefnbr, (int64_t)tk.size(), (int64_t)pc.size(), sl};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(tk.data(), sizeof(Task), tk.size(), f);
        fwrite(pc.data(), sizeof(Piece), pc.size(), f);
        fclose(f);
        fprintf(stderr, "pastix_amd: slot %d dumped to %s (%zu tasks, %zu pieces, %.3e flops)\n", sl, fn, tk.size(), pc.size(),
                H.slot_flops[sl]);
      }
    }
  }
  if (H.opts.verbose >= 2) {
    // developer aid: per bulk launch, the update flops by instance of the update loop (whole 128x128 tiles / edge tiles /
    // tasks with partial pieces) -- printed beside the launch's duration by pastix_amd_factorize_end
    H.slot_mode_flops.assign((size_t)H.nlevels * 3, 0.0);
    for (int sl = 0; sl < H.nlevels; sl++)
      for (int64_t q = H.slot_urgent_end[sl]; q < H.slot_task_ptr[sl + 1]; q++) {
        const Task& t = H.tasks[(size_t)q];
        const int m = (t.flags & 32u) ? 2 : (int)t.nfull == t.pn ? ((t.tm == TM && t.tn == TN) ? 0 : 1) : 2;
        for (int i = 0; i < t.pn; i++) {
          const Piece& pc = H.pieces[(size_t)t.p0 + i];
          H.slot_mode_flops[(size_t)sl * 3 + m] += 2.0 * pc.m * (double)pc.n * pc.k;
        }
      }
  }
  // the piece/task tables now live on the device; keep only what the host driver reads.  (Returning several GB to the
  // system takes 0.4 s at 200^3: a thread of its own does it.)
  {
    struct Junk { decltype(H.pieces) pieces; std::vector<Task> tasks, rtasks; std::vector<RunInfo> rinfo; std::vector<int32_t> rwaits, rcons, rdep; std::vector<RunCheck> rchk; };
    if (dev_opt("run_debug") && H.run_L0 >= 0) { p->dbg_info = H.run_info; p->dbg_cons = H.run_cons; p->dbg_dep = H.run_dep; p->dbg_d = H.run_d; }
    Junk* junk = new (std::nothrow) Junk();
    if (junk) {
      junk->pieces.swap(H.pieces);
      junk->tasks.swap(H.tasks);
      junk->rtasks.swap(H.run_tasks);
      junk->rinfo.swap(H.run_info);
      junk->rwaits.swap(H.run_waits);
      junk->rcons.swap(H.run_cons);
      junk->rdep.swap(H.run_dep);
      junk->rchk.swap(H.run_chk);
      try { std::thread([junk] { delete junk; }).detach(); } catch (const std::system_error&) { delete junk; }
    } else {
      decltype(H.pieces)().swap(H.pieces);
      std::vector<Task>().swap(H.tasks);
    }
  }
  phase("statistics, host tables freed");
  *out = p;
  return PASTIX_AMD_OK;
}

int pastix_amd_plan_create(const pastix_amd_layout_t* layout, int factotype, int floattype,
                           const pastix_amd_options_t* opts, pastix_amd_plan_t** out) {
  return plan_create_common(layout, factotype, floattype, opts, nullptr, 0, out);
}

int pastix_amd_plan_create_dist(const pastix_amd_layout_t* layout, int factotype, int floattype,
                                const pastix_amd_options_t* opts, const int32_t* owner, int32_t myrank,
                                pastix_amd_plan_t** out) {
  if (!owner) return PASTIX_AMD_ERR_BADPARAMETER;
  return plan_create_common(layout, factotype, floattype, opts, owner, myrank, out);
}

// Host-only: build the schedule of one rank and report per-slot update flops / largest task / task
// count (capacity-planning aid for the multi-GPU partition; needs no device).
int pastix_amd_plan_profile(const pastix_amd_layout_t* layout, int factotype, const pastix_amd_options_t* opts,
                            const int32_t* owner, int32_t myrank, pastix_amd_int_t maxlevels, double* slot_flops,
                            double* slot_maxwork, pastix_amd_int_t* slot_tasks, double* level_panel_flops,
                            pastix_amd_int_t* nlevels, double* slot_urgent_flops) {
  if (!layout || !nlevels) return PASTIX_AMD_ERR_BADPARAMETER;
  Plan P;
  int rc;
  try {
    // (cblks wider than the panel kernels take are re-cut into column groups, as pastix_amd_plan_create does)
    SplitMap sm;
    pastix_amd_layout_t sl = *layout;
    std::vector<int32_t> sowner;
    rc = build_split(layout, opts && opts->schur, sm);
    if (!rc && sm.active) {
      if (owner) {
        sowner.resize(sm.cblk.size() - 1);
        for (int64_t k = 0; k < sm.ocblknbr; k++)
          for (int64_t q = sm.first[k]; q < sm.first[k + 1]; q++) sowner[(size_t)q] = owner[k];
        owner = sowner.data();
      }
      sl.cblknbr = (pastix_amd_int_t)sm.cblk.size() - 1;
      sl.bloknbr = (pastix_amd_int_t)sm.blok.size();
      sl.cblktab = sm.cblk.data();
      sl.bloktab = sm.blok.data();
    }
    if (!rc) rc = build_plan(&sl, factotype, PASTIX_AMD_REALDOUBLE, opts, owner, myrank, P);
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  if (rc) return rc;
  if (dev_opt("mode_stats")) {
    // developer aid: update flops and 16-deep chunks by the loop instance that runs them
    const char* nm[9] = {"mode0 full tile", "mode1 smaller tile", "mixed: full pieces", "mixed: partial pieces", "all partial", "quadrant",
                         "gathered: gathered pieces", "gathered: whole-tile pieces", "gathered: rectangles"};
    double fl[9] = {0}, ch[9] = {0}; int64_t nt[9] = {0};
    for (const Task& tk : P.tasks) {
      int cls;
      if (tk.flags & TASK_GATHERED) cls = 6;
      else if (tk.flags & 32u) cls = 5;
      else if ((int)tk.nfull == tk.pn) cls = (tk.tm == TM && tk.tn == TN) ? 0 : 1;
      else if (tk.nfull > 0) cls = 2;
      else cls = 4;
      nt[cls]++;
      for (int z = 0; z < tk.pn; z++) {
        const Piece& pc = P.pieces[(size_t)tk.p0 + (size_t)z];
        const int c2 = (cls == 2 && z >= (int)tk.nfull) ? 3 : cls == 6 ? ((pc.flags & PIECE_GATHERED) ? 6 : z < (int)tk.nfull ? 7 : 8) : cls;
        fl[c2] += 2.0 * pc.m * (double)pc.n * pc.k;
        ch[c2] += (pc.k + 15) / 16;
      }
    }
    double tf = 0, tc = 0;
    for (int i = 0; i < 9; i++) { tf += fl[i]; tc += ch[i]; }
    for (int i = 0; i < 9; i++)
      fprintf(stderr, "mode_stats %-28s tasks %9lld flops %6.2f %% chunks %6.2f %% flop/chunk %8.0f (full tile chunk = 524288)\n", nm[i],
              (long long)nt[i], 100 * fl[i] / tf, 100 * ch[i] / tc, ch[i] > 0 ? fl[i] / ch[i] : 0.0);
  }
  if (dev_opt("piece_hist")) {
    // developer aid: the chunks of the pieces that do not run on the branch-free loop, by the 16-row / 16-column bands of the
    // tile they touch (rows: bands, columns: bands; weight = 16-deep chunks), and how many of them touch <= 4 x <= 4 bands
    double h[9][9] = {{0}}, hk[5] = {0}, tot = 0, small = 0, allc = 0, hf[9][9] = {{0}};
    for (const Task& tk : P.tasks) {
      const bool whole = (int)tk.nfull == tk.pn && tk.tm == TM && tk.tn == TN;
      for (int z = 0; z < tk.pn; z++) {
        const Piece& pc = P.pieces[(size_t)tk.p0 + (size_t)z];
        const double c = (pc.k + 15) / 16;
        allc += c;
        if (whole || (z < (int)tk.nfull && tk.tm == TM && tk.tn == TN) || (pc.flags & PIECE_GATHERED)) continue;
        const int rb = (pc.dr + pc.m - 1) / 16 - pc.dr / 16 + 1, cb = (pc.dc + pc.n - 1) / 16 - pc.dc / 16 + 1;
        h[rb][cb] += c;
        hf[rb][cb] += 2.0 * pc.m * pc.n * pc.k;
        tot += c;
        if (rb <= 4 && cb <= 4) small += c;
        hk[pc.k <= 16 ? 0 : pc.k <= 32 ? 1 : pc.k <= 64 ? 2 : pc.k <= 96 ? 3 : 4] += c;
      }
    }
    fprintf(stderr, "piece_hist: %.1f %% of all chunks are not branch-free whole tiles; of those %.1f %% touch <= 4 x 4 bands\n",
            100 * tot / allc, 100 * small / tot);
    fprintf(stderr, "piece_hist: K <= 16 / 32 / 64 / 96 / 128: %.1f %.1f %.1f %.1f %.1f %%\n", 100 * hk[0] / tot, 100 * hk[1] / tot,
            100 * hk[2] / tot, 100 * hk[3] / tot, 100 * hk[4] / tot);
    fprintf(stderr, "piece_hist: %% of those chunks (fill of the touched bands in %%) by row bands (down) x column bands (across, 1..8)\n");
    for (int r = 1; r <= 8; r++) {
      fprintf(stderr, "piece_hist: %d |", r);
      for (int c = 1; c <= 8; c++) fprintf(stderr, " %5.1f(%3.0f)", 100 * h[r][c] / tot, h[r][c] > 0 ? 100 * hf[r][c] / (h[r][c] * 2.0 * 16 * 16 * 16 * r * c) : 0.0);
      fprintf(stderr, "\n");
    }
  }
  *nlevels = P.nlevels;
  for (int l = 0; l < P.nlevels && l < maxlevels; l++) {
    if (slot_flops) slot_flops[l] = P.slot_flops[l];
    if (slot_urgent_flops) slot_urgent_flops[l] = P.slot_urgent_flops[l];
    if (slot_maxwork) slot_maxwork[l] = P.slot_maxwork[l];
    if (slot_tasks) slot_tasks[l] = P.slot_task_ptr[l + 1] - P.slot_task_ptr[l];
    if (level_panel_flops) {
      double f = 0;
      for (int64_t q = P.lvl_cblk_ptr[l]; q < P.lvl_cblk_ptr[l + 1]; q++) {
        const int32_t k = P.lvl_cblk[q];
        const double N = double(P.cblk[k].lcolnum - P.cblk[k].fcolnum + 1), M = double(P.cblk[k].stride) - N;
        f += N * N * N / 3.0 + M * N * (N + 1.0);
      }
      level_panel_flops[l] = f;
    }
  }
  return PASTIX_AMD_OK;
}

// Host-only (tests, capacity planning): the run schedule of a layout (plan.h RunInfo) and its replay check.
// info[0..7] = first level of the run (-1: none), levels, update tasks in the run, source-tile waits, diagonal workers,
// panel-solve tasks, update flops inside the run (rounded), result of run_verify (0 = every ticket can run).
int pastix_amd_plan_run_info(const pastix_amd_layout_t* layout, int factotype, int floattype, const pastix_amd_options_t* opts,
                             pastix_amd_int_t* info) {
  if (!layout || !info) return PASTIX_AMD_ERR_BADPARAMETER;
  Plan P;
  int rc;
  try {
    rc = build_plan(layout, factotype, floattype, opts, nullptr, 0, P);
    if (rc) return rc;
    info[0] = P.run_L0;
    info[1] = P.nlevels;
    info[2] = (pastix_amd_int_t)P.run_tasks.size();
    info[3] = (pastix_amd_int_t)P.run_cons.size();
    info[4] = P.run_gd;
    info[5] = 0; for (const RunInfo& ri : P.run_info) info[5] += (ri.kind & 4) != 0;
    info[6] = (pastix_amd_int_t)P.run_flops;
    info[7] = run_verify(P);
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}

// Tests: a digest of the run's dependency tables as they stand on the device (include/pastix_amd.h).
int pastix_amd_plan_run_edges_digest(pastix_amd_plan_t* p, pastix_amd_int_t* out) {
  if (!p || !out) return PASTIX_AMD_ERR_BADPARAMETER;
  for (int i = 0; i < 4; i++) out[i] = -1;
  if (!p->run_ready || !p->dRunInfo) return PASTIX_AMD_OK;
  HIPCHK(hipSetDevice(p->device));
  const size_t nr = (size_t)p->run_nticket;
  std::vector<RunInfo> info(nr);
  std::vector<int32_t> cons(std::max<size_t>(p->nRunCons, 1));
  HIPCHK(hipMemcpy(info.data(), p->dRunInfo, nr * sizeof(RunInfo), hipMemcpyDeviceToHost));
  if (p->nRunCons) HIPCHK(hipMemcpy(cons.data(), p->dRunCons, p->nRunCons * sizeof(int32_t), hipMemcpyDeviceToHost));
  auto mix = [](uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; };
  uint64_t npair = 0, h = 0, hd = 0, nready = 0;
  for (size_t i = 0; i < nr; i++) {
    if (!(info[i].kind & 4)) continue;
    if (info[i].cn < 0 || (size_t)info[i].cptr + (size_t)info[i].cn > p->nRunCons) return PASTIX_AMD_ERR_LAYOUT;
    for (int32_t q = 0; q < info[i].cn; q++) {
      const int32_t c = cons[(size_t)info[i].cptr + (size_t)q];
      if (q > 0 && c <= cons[(size_t)info[i].cptr + (size_t)q - 1]) return PASTIX_AMD_ERR_LAYOUT;   // (lists are ordered by ticket)
      h += mix(((uint64_t)i << 32) | (uint32_t)c);
      npair++;
    }
  }
  for (size_t i = 0; i < p->runDep0.size(); i++) { hd += mix(((uint64_t)i << 32) | (uint32_t)p->runDep0[i]); nready += p->runDep0[i] == 0; }
  out[0] = (pastix_amd_int_t)npair;
  out[1] = (pastix_amd_int_t)(h >> 1);
  out[2] = (pastix_amd_int_t)(hd >> 1);
  out[3] = (pastix_amd_int_t)nready;
  return PASTIX_AMD_OK;
}

// Host-only (tests): the pieces of the plan against the reference's definition of the update -- every product (blok j x
// blok i of a source cblk, j >= i) subtracted once where add_contrib_local puts it (plan.cpp verify_pieces; real LLt / LDLt).
int pastix_amd_plan_check_pieces(const pastix_amd_layout_t* layout, int factotype, const pastix_amd_options_t* opts,
                                 pastix_amd_int_t* out) {
  if (!layout || !out || (factotype != PASTIX_AMD_FACT_LLT && factotype != PASTIX_AMD_FACT_LDLT)) return PASTIX_AMD_ERR_BADPARAMETER;
  Plan P;
  try {
    int rc = build_plan(layout, factotype, PASTIX_AMD_REALDOUBLE, opts, nullptr, 0, P);
    if (rc) return rc;
    int64_t o[4];
    verify_pieces(P, o);
    for (int i = 0; i < 4; i++) out[i] = o[i];
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_fanin_touched(const pastix_amd_layout_t* layout, const int32_t* owner, uint64_t* mask) {
  if (!layout || !owner || !mask) return PASTIX_AMD_ERR_BADPARAMETER;
  return fanin_touched(layout, owner, mask);
}

int pastix_amd_plan_fanin_add(pastix_amd_plan_t* p, pastix_amd_int_t cblk, const void* src, const int32_t* rows,
                              pastix_amd_int_t nrows) {
  if (p && p->split.active) return PASTIX_AMD_ERR_UNSUPPORTED;   // (addresses panels by original cblk)
  if (!p || !src || !rows || cblk < 0 || cblk >= p->host.cblknbr || nrows < 0) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  if (H.role[cblk] != 1 || p->cplx || p->f32 || !p->dL) return PASTIX_AMD_ERR_BADPARAMETER;
  p->refillable = false;                                     // (the panels are no longer what pastix_amd_refill would write)
  HIPCHK(hipSetDevice(p->device));
  const int64_t w = H.cblk[cblk].lcolnum - H.cblk[cblk].fcolnum + 1;
  launch_fanin_add(p->stream, p->dL + H.poff[cblk], H.cblk[cblk].stride, (const double*)src, rows, nrows, w);
  HIPCHK(hipGetLastError());
  return PASTIX_AMD_OK;
}

// The update kernel's 16-byte DMA lanes touch the element next to a panel when a contribution starts or ends on an
// odd row, so the panels sit EXT_ARENA_PAD elements inside the caller's buffer; the caller allocates what
// pastix_amd_plan_arena_info reports and passes the allocation itself (the size is checked here).
static const int64_t EXT_ARENA_PAD = 32;
int pastix_amd_plan_arena_info(const pastix_amd_plan_t* p, pastix_amd_int_t* nelems, pastix_amd_int_t* first) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  if (nelems) *nelems = std::max<int64_t>(p->host.poff[p->host.cblknbr], 1) + 2 * EXT_ARENA_PAD;   // (real plans only)
  if (first) *first = EXT_ARENA_PAD;
  return PASTIX_AMD_OK;
}
int pastix_amd_plan_set_arena(pastix_amd_plan_t* p, void* dL, void* dU, pastix_amd_int_t nelems) {
  if (p) p->refillable = false;
  if (p && p->split.active) return PASTIX_AMD_ERR_UNSUPPORTED;   // (addresses panels by original cblk)
  if (!p || !dL || p->own_arena) return PASTIX_AMD_ERR_BADPARAMETER;
  pastix_amd_int_t need = 0;
  (void)pastix_amd_plan_arena_info(p, &need, nullptr);
  if (nelems < need) return PASTIX_AMD_ERR_BADPARAMETER;
  p->dL = (double*)dL + EXT_ARENA_PAD;
  p->dU = dU ? (double*)dU + EXT_ARENA_PAD : nullptr;
  return PASTIX_AMD_OK;
}

int pastix_amd_plan_set_stream(pastix_amd_plan_t* p, void* stream) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  if (p->own_stream && p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
  p->stream = (hipStream_t)stream;
  p->own_stream = false;
  return PASTIX_AMD_OK;
}

int pastix_amd_plan_layout_info(const pastix_amd_plan_t* p, pastix_amd_int_t* poff, int32_t* level, int8_t* role) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  if (p->split.active) {
    // by ORIGINAL cblk: offsets in the caller's packed layout, level and role of the first column group (the groups of
    // a cblk share its owner; their levels follow each other)
    const SplitMap& M = p->split;
    if (poff) std::memcpy(poff, M.ooff.data(), (size_t)(M.ocblknbr + 1) * sizeof(int64_t));
    for (int64_t k = 0; k < M.ocblknbr; k++) {
      if (level) level[k] = H.level[(size_t)M.first[k]];
      if (role) role[k] = H.role[(size_t)M.first[k]];
    }
    return PASTIX_AMD_OK;
  }
  if (poff) std::memcpy(poff, H.poff.data(), (size_t)(H.cblknbr + 1) * sizeof(int64_t));
  if (level) std::memcpy(level, H.level.data(), (size_t)H.cblknbr * sizeof(int32_t));
  if (role) std::memcpy(role, H.role.data(), (size_t)H.cblknbr);
  return PASTIX_AMD_OK;
}

void pastix_amd_plan_destroy(pastix_amd_plan_t* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->dist && p->dist_free) { p->dist_free(p->dist); p->dist = nullptr; }
  if (p->own_arena)
    for (double* a : {p->dL, p->dU, p->dLi, p->dUi})
      if (a) (void)hipFree((char*)a - ARENA_PAD);
  (void)hipFree(p->dDinv); (void)hipFree(p->dTasks);
  (void)hipFree(p->dPieces); (void)hipFree(p->dPanel); (void)hipFree(p->dTrsm);
  (void)hipFree(p->dNbpivot); (void)hipFree(p->dErr);
  (void)hipFree(p->dFillIdxL); (void)hipFree(p->dFillValL); (void)hipFree(p->dFillValLi); (void)hipFree(p->dFillValUi); (void)hipFree(p->dFillIdxU); (void)hipFree(p->dFillValU);
  (void)hipFree(p->dThinF); (void)hipFree(p->dThinB); (void)hipFree(p->dThinTasks); (void)hipFree(p->dThinTgt); (void)hipFree(p->dThinExpect); (void)hipFree(p->dInvF); (void)hipFree(p->dInvB);
  (void)hipFree(p->dTicket);
  (void)hipFree(p->dSolve); (void)hipFree(p->dBlok); (void)hipFree(p->dChunk); (void)hipFree(p->dChunkB); (void)hipFree(p->dRidx); (void)hipFree(p->dXws);
  for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : p->evT) if (e) (void)hipEventDestroy(e);
  for (auto& e : p->evP) if (e) (void)hipEventDestroy(e);
  for (auto& e : p->evB) if (e) (void)hipEventDestroy(e);
  if (p->stream2) { (void)hipStreamSynchronize(p->stream2); (void)hipStreamDestroy(p->stream2); }
  if (p->stream3) { (void)hipStreamSynchronize(p->stream3); (void)hipStreamDestroy(p->stream3); }
  if (p->stream_io) { (void)hipStreamSynchronize(p->stream_io); (void)hipStreamDestroy(p->stream_io); }
  for (hipEvent_t e : {p->evZ, p->evS3}) if (e) (void)hipEventDestroy(e);
  (void)hipFree(p->dGmap);
  (void)hipFree(p->dRunTasks); (void)hipFree(p->dRunInfo); (void)hipFree(p->dRunCons); (void)hipFree(p->dRunD);
  (void)hipFree(p->dRunState); (void)hipFree(p->dRunImage); (void)hipFree(p->dRunProf);
  if (p->hResident) (void)hipHostFree(p->hResident);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream && p->own_stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

int pastix_amd_plan_stats(const pastix_amd_plan_t* p, pastix_amd_stats_t* stats) {
  if (!p || !stats) return PASTIX_AMD_ERR_BADPARAMETER;
  *stats = p->stats;
  return PASTIX_AMD_OK;
}

int pastix_amd_device_arenas(pastix_amd_plan_t* p, void** dL, void** dU) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  p->refillable = false;                                     // (the caller may write through these pointers)
  if (dL) *dL = p->dL;
  if (dU) *dU = p->dU;
  return PASTIX_AMD_OK;
}

// complex: the host side is interleaved (re,im) like the reference's `double complex` panels; the device
// keeps split planes.  Conversion goes through a bounded staging buffer.
static int z_transfer(pastix_amd_plan_t* p, bool to_device, void* host, double* re, double* im, int64_t off, int64_t n) {
  const int64_t CH = 1 << 24;   // elements per chunk (256 MiB of interleaved data)
  double* stage = nullptr;
  HIPCHK(hipMalloc((void**)&stage, std::min(CH, std::max<int64_t>(n, 1)) * 2 * sizeof(double)));
  for (int64_t c = 0; c < n; c += CH) {
    const int64_t m = std::min(CH, n - c);
    double* h = (double*)host + 2 * c;
    if (to_device) {
      HIPCHK(hipMemcpyAsync(stage, h, m * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
      launch_split(p->stream, stage, re + off + c, im + off + c, m);
    } else {
      launch_merge(p->stream, stage, re + off + c, im + off + c, m);
      HIPCHK(hipMemcpyAsync(h, stage, m * 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
  }
  HIPCHK(hipFree(stage));
  return 0;
}

// Host <-> device copies of ONE original cblk of a plan with split cblks (SplitMap).  hostL / hostU: the caller's
// panels of that cblk in the reference's layout (ostride x owidth, column-major; complex: interleaved).
// Sub-cblk j holds, for its columns, the panel rows from its own diagonal blok down: column c' of its panel is the
// tail [off_j, ostride) of original column off_j + c'.  The blocks of the original diagonal blok ABOVE a column
// group (rows of an earlier group) are not L entries: for LU they are U (kept transposed in the earlier group's U
// panel) and are written back into coeftab's diagonal blok after a factorization, as the reference has them
// (getrf works on coeftab's square blok, compute_diag.c:486-532); otherwise zeros.  ucoeftab's diagonal blok comes
// back with the factor's U^T blocks below the groups' diagonals and zeros above -- the reference leaves its
// initial fill there, nothing reads it.
static int split_cblk_io(pastix_amd_plan_t* p, int64_t k, bool up, void* hostL, void* hostU) {
  const Plan& H = p->host;
  SplitMap& M = p->split;
  const size_t eb = p->cplx ? 16 : p->esz;       // bytes per entry of the caller's panels (double complex: interleaved)
  const int64_t s0 = M.first[k], ns = M.first[k + 1] - s0, os = M.ostride[k], ow = M.owidth[k];
  auto xfer = [&](int arena, int64_t off, int64_t cnt, void* host) -> int {
    double* re = arena ? p->dU : p->dL;
    double* im = arena ? p->dUi : p->dLi;
    if (p->cplx) return z_transfer(p, up, host, re, im, off, cnt);
    if (up) HIPCHK(hipMemcpy(p->at(re, off), host, cnt * eb, hipMemcpyHostToDevice));
    else HIPCHK(hipMemcpy(host, p->at(re, off), cnt * eb, hipMemcpyDeviceToHost));
    return 0;
  };
  const bool haveU = p->dU && hostU;
  if (ns == 1) {
    int r = xfer(0, H.poff[s0], os * ow, hostL);
    if (!r && haveU) r = xfer(1, H.poff[s0], os * ow, hostU);
    return r;
  }
  const int64_t fcol = H.cblk[s0].fcolnum;
  std::vector<unsigned char>* keep = (!M.upper.empty() && !M.upper[(size_t)k].empty()) ? &M.upper[(size_t)k] : nullptr;
  if (up && keep) {                               // the diagonal blok's blocks above the column groups (see SplitMap::upper)
    const unsigned char* host = (const unsigned char*)hostL;
    for (int64_t j = 1; j < ns; j++) {
      const int64_t offj = H.cblk[s0 + j].fcolnum - fcol, wj = H.cblk[s0 + j].lcolnum - H.cblk[s0 + j].fcolnum + 1;
      for (int64_t c = 0; c < wj; c++)
        memcpy(keep->data() + (offj + c) * ow * eb, host + (offj + c) * os * eb, (size_t)offj * eb);
    }
  }
  // the column groups' panels pass through PINNED memory (a bump arena per host thread, grown on demand: the copies then
  // run at the link's rate), their columns are packed / unpacked by several host threads: blend's 144- / 152-column cblks
  // hold most of the bytes of a big layout
  struct Slice { unsigned char* d = nullptr; unsigned char* data() const { return d; } };
  static thread_local unsigned char* pin_base = nullptr;
  static thread_local size_t pin_cap = 0;
  {
    const size_t need = (size_t)2 * (size_t)os * (size_t)ow * eb + 4096;
    if (need > pin_cap) {
      if (pin_base) (void)hipHostFree(pin_base);
      pin_base = nullptr; pin_cap = 0;
      HIPCHK(hipHostMalloc((void**)&pin_base, need, hipHostMallocDefault));
      pin_cap = need;
    }
  }
  size_t pin_used = 0;
  auto par_cols = [&](int64_t ncol, size_t bytes_per_col, auto&& fn) {
    const int nt = (size_t)ncol * bytes_per_col < ((size_t)2 << 20) ? 1 : (int)std::min<int64_t>(8, std::max<int64_t>(1, ncol));
    if (nt == 1) { for (int64_t c = 0; c < ncol; c++) fn(c); return; }
    std::atomic<int64_t> next{0};
    auto work = [&] { for (;;) { const int64_t c = next.fetch_add(4); if (c >= ncol) break; for (int64_t q = c; q < std::min(ncol, c + 4); q++) fn(q); } };
    std::vector<std::thread> th;
    for (int t2 = 1; t2 < nt; t2++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  };
  std::vector<Slice> tmp[2];
  for (int a = 0; a < 2; a++) tmp[a].resize((size_t)ns);
  for (int a = 0; a < (haveU ? 2 : 1); a++) {
    unsigned char* host = (unsigned char*)(a ? hostU : hostL);
    for (int64_t j = 0; j < ns; j++) {
      const int64_t s = s0 + j, off = H.cblk[s].fcolnum - fcol, wj = H.cblk[s].lcolnum - H.cblk[s].fcolnum + 1;
      const int64_t nr = H.cblk[s].stride;                      // == os - off
      Slice& t = tmp[a][(size_t)j];
      t.d = pin_base + pin_used;
      pin_used += ((size_t)(nr * wj) * eb + 255) & ~(size_t)255;
      if (up) {
        par_cols(wj, (size_t)nr * eb, [&](int64_t c) {
          memcpy(t.data() + c * nr * eb, host + ((off + c) * os + off) * eb, (size_t)nr * eb); });
        if (a == 1 && H.factotype == PASTIX_AMD_FACT_LU) {
          // LU: the reference keeps the whole square A_kk in coeftab's diagonal blok and zeros in ucoeftab's
          // (csc_intern_solve.c:65-132); the U^T blocks facing the later column groups are the transposes of
          // coeftab's blocks right of this group's diagonal
          const unsigned char* hl = (const unsigned char*)hostL;
          for (int64_t c = 0; c < wj; c++)
            for (int64_t pr = off + wj; pr < ow; pr++)             // original diagonal-blok row/col index of the later groups
              memcpy(t.data() + ((pr - off) + c * nr) * eb, hl + ((off + c) + pr * os) * eb, eb);
        }
      }
      int r = xfer(a, H.poff[s], nr * wj, t.data());
      if (r) return r;
      if (!up) {
        par_cols(wj, (size_t)nr * eb, [&](int64_t c) {
          memcpy(host + ((off + c) * os + off) * eb, t.data() + c * nr * eb, (size_t)nr * eb); });
      }
    }
  }
  if (!up) {
    // the blocks of the original diagonal blok above a column group: zeros, except for LU, where coeftab's square blok
    // holds U there and ucoeftab's holds the transpose of L (both bloks are full squares in the reference, transposes
    // of each other at fill time, csc_intern_solve.c:65-132)
    const bool lu = H.factotype == PASTIX_AMD_FACT_LU && haveU;
    if (lu && !p->factored)                       // before a factorization ucoeftab's diagonal blok is all zeros
      for (int64_t c = 0; c < ow; c++) memset((unsigned char*)hostU + c * os * eb, 0, (size_t)ow * eb);
    for (int64_t j = 1; j < ns; j++) {
      const int64_t offj = H.cblk[s0 + j].fcolnum - fcol, wj = H.cblk[s0 + j].lcolnum - H.cblk[s0 + j].fcolnum + 1;
      for (int64_t c = 0; c < wj; c++) {
        unsigned char* col[2] = {(unsigned char*)hostL + (offj + c) * os * eb,
                                 haveU ? (unsigned char*)hostU + (offj + c) * os * eb : nullptr};
        for (int a = 0; a < (haveU ? 2 : 1); a++) {
          memset(col[a], 0, (size_t)offj * eb);
          if (a == 0 && keep) memcpy(col[0], keep->data() + (offj + c) * ow * eb, (size_t)offj * eb);
          if (!lu || (a == 1 && !p->factored)) continue;
          for (int64_t i = 0; i < j; i++) {                      // rows of the earlier group i, from the OTHER arena's panel of i
            const int64_t offi = H.cblk[s0 + i].fcolnum - fcol, wi = H.cblk[s0 + i].lcolnum - H.cblk[s0 + i].fcolnum + 1;
            const int64_t ldi = H.cblk[s0 + i].stride;
            const Slice& ot = tmp[1 - a][(size_t)i];
            for (int64_t q = 0; q < wi; q++)
              memcpy(col[a] + (offi + q) * eb, ot.data() + ((offj + c - offi) + q * ldi) * eb, eb);
          }
        }
      }
    }
  }
  return 0;
}

// all cblks of a plan with split cblks; tabs (per-cblk pointers) or packed arrays in the original layout
static int split_io(pastix_amd_plan_t* p, bool up, void* const* coeftab, void* const* ucoeftab, void* packedL,
                    void* packedU) {
  const SplitMap& M = p->split;
  const size_t eb = p->cplx ? 16 : p->esz;
  for (int64_t k = 0; k < M.ocblknbr; k++) {
    void* hl = coeftab ? coeftab[k] : (void*)((char*)packedL + M.ooff[k] * eb);
    void* hu = coeftab ? (ucoeftab ? ucoeftab[k] : nullptr) : (packedU ? (void*)((char*)packedU + M.ooff[k] * eb) : nullptr);
    if (p->host.role[(size_t)M.first[k]] != 1) continue;          // (distributed plans: only owned panels travel)
    if (!hl) return PASTIX_AMD_ERR_BADPARAMETER;
    int r = split_cblk_io(p, k, up, hl, hu);
    if (r) return r;
  }
  return 0;
}

// Host panels <-> device arena through pinned staging, for the one-buffer-per-cblk arrays of the reference (pageable
// memory, thousands of panels from a few KB to hundreds of MB): the panels of consecutive cblks are adjacent in the arena,
// so a CHUNK of them is packed by host threads into one pinned buffer and travels as one copy; four buffers, so that the
// packing of the next chunks runs beside the copies in flight (per-cblk copies from pageable memory: 100^3, 16 k panels of
// 9 GB in all, took 0.9 s each way).  Real arithmetic, cblks not re-cut; everything else keeps the per-cblk path.
// part: 0 = every panel; 1 = the panels of the levels below the run (not re-cut ones), on `strm`; 2 = the others.
static int staged_tabs_io(pastix_amd_plan_t* p, bool up, void* const* coeftab, void* const* ucoeftab, const int part = 0,
                          hipStream_t strm = nullptr) {
  const Plan& H = p->host;
  const SplitMap& M = p->split;
  if (!strm) strm = p->stream;
  // items = the caller's cblks: the plan's own, or -- where cblks wider than 128 columns were re-cut (SplitMap) -- the
  // original ones: an original cblk that was not re-cut is one panel of the arena as before, a re-cut one goes through
  // split_cblk_io (its column groups are separate panels)
  const bool sp = M.active;
  const int64_t nitem = sp ? M.ocblknbr : H.cblknbr;
  auto first = [&](int64_t k) { return sp ? M.first[(size_t)k] : k; };
  auto doff = [&](int64_t k) { return H.poff[(size_t)first(k)]; };            // (k == nitem: the end of the arena)
  auto recut = [&](int64_t k) { return sp && M.first[(size_t)k + 1] - M.first[(size_t)k] > 1; };
  // A re-cut cblk's column groups are consecutive panels of the arena like any run of cblks: they travel through the same
  // staging, the packing threads cutting the caller's one panel into the groups' panels (split_cblk_io's layout).
  // LU: the blocks of a re-cut cblk's diagonal blok above a column group live in the OTHER arena's panels (transposed): on
  // the way in the U panels gather them from the caller's coeftab, on the way out a host pass over the two returned
  // buffers puts them back (below).  Only a download BEFORE any factorization (ucoeftab's diagonal blok is zeros then)
  // keeps split_cblk_io.
  const bool lu = H.factotype == PASTIX_AMD_FACT_LU;
  // (part 1 runs inside a factorization, on panels that are final: the factors, whatever p->factored says yet)
  const bool recut_staged = sp && (!lu || (p->dU && ucoeftab && (up || p->factored || part == 1)));
  auto last_sub = [&](int64_t k) { return sp ? M.first[(size_t)k + 1] - 1 : k; };
  auto below_run = [&](int64_t k) {
    return H.run_L0 > 0 && (!recut(k) || recut_staged) && H.level[(size_t)last_sub(k)] < H.run_L0;
  };
  auto owned = [&](int64_t k) {
    return H.role[(size_t)first(k)] == 1 && (part == 0 || (part == 1) == below_run(k));
  };
  constexpr int NBUF = 4;
  const size_t CH = (size_t)96 << 20;                    // bytes per staging buffer
  struct Stage { char* buf = nullptr; hipEvent_t ev = nullptr; bool busy = false; int64_t k0 = 0, k1 = 0; int arena = 0; };
  // (kept for the thread's life: pinning 384 MB costs ~0.1 s; one set per DEVICE -- an event belongs to the device it was
  // created on, and a thread may drive plans on several)
  static thread_local std::map<int, std::array<Stage, NBUF>> st_by_dev;
  std::array<Stage, NBUF>& st = st_by_dev[p->device];
  for (auto& x : st) {
    if (!x.buf) { HIPCHK(hipHostMalloc((void**)&x.buf, CH, hipHostMallocPortable)); HIPCHK(hipEventCreateWithFlags(&x.ev, hipEventDisableTiming)); }
    x.busy = false;
  }
  // (host threads that pack / unpack a chunk: developer override PASTIX_AMD_DEV=io_threads=<n>)
  const int nthr = dev_opt("io_threads") ? std::max(1, atoi(dev_opt("io_threads")))
                                         : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  // pack / unpack the panels of the items [k0, k1) of one arena between the caller's buffers and a staging buffer
  auto move = [&](const Stage& x, bool to_stage) {
    void* const* tab = x.arena ? ucoeftab : coeftab;
    const int64_t base = doff(x.k0);
    // work items of about 1 MB: runs of small panels, slices of big ones (a chunk of the top separators is a handful of
    // 10 MB panels -- handed out panel-wise, one thread packed all of it: 25 GB/s for those chunks against 42 for the rest)
    struct Seg { int64_t q0, q1; size_t off, n; int64_t j, c0, c1; };   // panels [q0, q1) whole | bytes [off, off + n) of panel
    constexpr size_t SEG = (size_t)1 << 20;                               // q0 | columns [c0, c1) of group j of re-cut cblk q0
    std::vector<Seg> segs;
    {
      int64_t q0 = x.k0;
      size_t acc = 0;
      for (int64_t q = x.k0; q < x.k1; q++) {
        const size_t bytes = owned(q) ? (size_t)(doff(q + 1) - doff(q)) * p->esz : 0;
        if (bytes && recut(q)) {
          if (q > q0) segs.push_back(Seg{q0, q, 0, 0, -1, 0, 0});
          for (int64_t s2 = M.first[(size_t)q]; s2 < M.first[(size_t)q + 1]; s2++) {
            const int64_t wj = H.cblk[(size_t)s2].lcolnum - H.cblk[(size_t)s2].fcolnum + 1;
            const int64_t per = std::max<int64_t>(1, (int64_t)(SEG / std::max<size_t>(1, (size_t)H.cblk[(size_t)s2].stride * p->esz)));
            for (int64_t c = 0; c < wj; c += per) segs.push_back(Seg{q, q, 0, 0, s2 - M.first[(size_t)q], c, std::min(wj, c + per)});
          }
          q0 = q + 1;
          acc = 0;
        } else if (bytes > SEG) {
          if (q > q0) segs.push_back(Seg{q0, q, 0, 0, -1, 0, 0});
          for (size_t o = 0; o < bytes; o += SEG) segs.push_back(Seg{q, q, o, std::min(SEG, bytes - o), -1, 0, 0});
          q0 = q + 1;
          acc = 0;
        } else if ((acc += bytes) >= SEG) {
          segs.push_back(Seg{q0, q + 1, 0, 0, -1, 0, 0});
          q0 = q + 1;
          acc = 0;
        }
      }
      if (x.k1 > q0) segs.push_back(Seg{q0, x.k1, 0, 0, -1, 0, 0});
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= segs.size()) break;
        const Seg& g = segs[i];
        if (g.j >= 0) {
          // columns [c0, c1) of column group j of the re-cut cblk q0 (split_cblk_io: column c' of the group's panel is the
          // tail [off, ostride) of original column off + c'; the part of the diagonal blok above the group is not on the
          // device -- kept in SplitMap::upper on the way in, put back (or zeros) on the way out)
          const int64_t q = g.q0, s0 = M.first[(size_t)q], sj = s0 + g.j;
          const int64_t os = M.ostride[(size_t)q], ow = M.owidth[(size_t)q];
          const int64_t off = H.cblk[(size_t)sj].fcolnum - H.cblk[(size_t)s0].fcolnum, nr = H.cblk[(size_t)sj].stride;
          const size_t eb = p->esz;
          char* host = (char*)tab[q];
          char* stg = x.buf + (size_t)(H.poff[(size_t)sj] - base) * eb;
          std::vector<unsigned char>* keep = (x.arena == 0 && !p->split.upper.empty() && !p->split.upper[(size_t)q].empty())
                                                 ? &p->split.upper[(size_t)q] : nullptr;
          const int64_t wj = H.cblk[(size_t)sj].lcolnum - H.cblk[(size_t)sj].fcolnum + 1;
          for (int64_t c = g.c0; c < g.c1; c++) {
            char* hc = host + (size_t)((off + c) * os) * eb;                // original column off + c
            if (to_stage) {
              memcpy(stg + (size_t)(c * nr) * eb, hc + (size_t)off * eb, (size_t)nr * eb);
              if (keep && off) memcpy(keep->data() + (size_t)((off + c) * ow) * eb, hc, (size_t)off * eb);
              if (lu && x.arena == 1) {
                // (the reference keeps the whole square A_kk in coeftab's diagonal blok and zeros in ucoeftab's,
                // csc_intern_solve.c:65-132: the U^T blocks facing the later column groups are the transposes of coeftab's
                // blocks right of this group's diagonal)
                const char* hl = (const char*)coeftab[q];
                for (int64_t pr = off + wj; pr < ow; pr++)
                  memcpy(stg + (size_t)((pr - off) + c * nr) * eb, hl + (size_t)((off + c) + pr * os) * eb, eb);
              }
            } else {
              memcpy(hc + (size_t)off * eb, stg + (size_t)(c * nr) * eb, (size_t)nr * eb);
              if (off && !lu) {
                if (keep) memcpy(hc, keep->data() + (size_t)((off + c) * ow) * eb, (size_t)off * eb);
                else memset(hc, 0, (size_t)off * eb);
              }
            }
          }
          continue;
        }
        if (g.q1 == g.q0) {
          char* sp2 = x.buf + (size_t)(doff(g.q0) - base) * p->esz + g.off;
          char* h = (char*)tab[g.q0] + g.off;
          if (to_stage) memcpy(sp2, h, g.n); else memcpy(h, sp2, g.n);
          continue;
        }
        for (int64_t q = g.q0; q < g.q1; q++) {
          if (!owned(q)) continue;
          const size_t bytes = (size_t)(doff(q + 1) - doff(q)) * p->esz;
          char* sp2 = x.buf + (size_t)(doff(q) - base) * p->esz;
          if (to_stage) memcpy(sp2, tab[q], bytes); else memcpy(tab[q], sp2, bytes);
        }
      }
    };
    const size_t total = (size_t)(doff(x.k1) - base) * p->esz;
    const int nt = total < ((size_t)4 << 20) ? 1 : nthr;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  };
  auto drain = [&](Stage& x) -> int {
    if (!x.busy) return 0;
    HIPCHK(hipEventSynchronize(x.ev));
    if (!up) move(x, false);
    x.busy = false;
    return 0;
  };
  int cur = 0;
  double moved = 0;
  const double t_io = now_s();
  for (int arena = 0; arena < ((p->dU && ucoeftab) ? 2 : 1); arena++) {
    double* dev = arena ? p->dU : p->dL;
    for (int64_t k0 = 0; k0 < nitem;) {
      auto by_hand = [&](int64_t k) {                      // re-cut cblks that keep split_cblk_io
        return recut(k) && (!recut_staged || (size_t)(doff(k + 1) - doff(k)) * p->esz > CH);
      };
      if (by_hand(k0)) {                                   // (both arenas at once, on the first pass)
        if (arena == 0 && owned(k0)) {
          int r = split_cblk_io(p, k0, up, coeftab[k0], ucoeftab ? ucoeftab[k0] : nullptr);
          if (r) return r;
        }
        k0++;
        continue;
      }
      // Distributed plans: only OWNED panels travel.  A fan-in buffer (role 2) has a size in the arena and must keep what
      // the device holds (zeros before a factorization: dist.py upload_owned relies on it), so a chunk is a run of owned
      // panels; absent cblks (size 0) may lie inside one.
      if (!owned(k0)) { k0++; continue; }
      auto carried = [&](int64_t k) { return owned(k) || doff(k + 1) == doff(k); };
      int64_t k1 = k0 + 1;                                 // (a panel larger than the buffer travels alone, below)
      while (k1 < nitem && !by_hand(k1) && carried(k1) && (size_t)(doff(k1 + 1) - doff(k0)) * p->esz <= CH) k1++;
      const size_t bytes = (size_t)(doff(k1) - doff(k0)) * p->esz;
      moved += (double)bytes;
      if (bytes > CH) {                                    // one huge panel: straight from / to the caller's memory
        if (owned(k0)) {
          void* h = (arena ? ucoeftab : coeftab)[k0];
          if (up) HIPCHK(hipMemcpyAsync(p->at(dev, doff(k0)), h, bytes, hipMemcpyHostToDevice, strm));
          else HIPCHK(hipMemcpyAsync(h, p->at(dev, doff(k0)), bytes, hipMemcpyDeviceToHost, strm));
        }
        k0 = k1;
        continue;
      }
      Stage& x = st[cur];
      cur = (cur + 1) % NBUF;
      int r = drain(x);
      if (r) return r;
      x.k0 = k0; x.k1 = k1; x.arena = arena;
      if (up) {
        move(x, true);
        HIPCHK(hipMemcpyAsync(p->at(dev, doff(k0)), x.buf, bytes, hipMemcpyHostToDevice, strm));
      } else {
        HIPCHK(hipMemcpyAsync(x.buf, p->at(dev, doff(k0)), bytes, hipMemcpyDeviceToHost, strm));
      }
      HIPCHK(hipEventRecord(x.ev, strm));
      x.busy = true;
      k0 = k1;
    }
  }
  for (int i = 0; i < NBUF; i++) { int r = drain(st[(cur + i) % NBUF]); if (r) return r; }
  HIPCHK(hipStreamSynchronize(strm));
  if (!up && lu && recut_staged) {
    // LU, after both arenas are home: in a re-cut cblk's diagonal blok the rows of a column above its own group belong to
    // an earlier group's OTHER factor -- coeftab's square blok holds U there, ucoeftab's the transpose of L (both bloks are
    // full squares in the reference, transposes of each other, csc_intern_solve.c:65-132).  Both are in the other
    // buffer's lower part, transposed.
    std::atomic<int64_t> nextq{0};
    auto fix = [&] {
      for (;;) {
        const int64_t q = nextq.fetch_add(8);
        if (q >= nitem) break;
        for (int64_t k = q; k < std::min(nitem, q + 8); k++) {
          if (!recut(k) || !owned(k) || (size_t)(doff(k + 1) - doff(k)) * p->esz > CH) continue;
          const int64_t s0 = M.first[(size_t)k], ns = M.first[(size_t)k + 1] - s0, os = M.ostride[(size_t)k];
          char* hl = (char*)coeftab[k];
          char* hu = (char*)ucoeftab[k];
          const size_t eb = p->esz;
          for (int64_t j = 1; j < ns; j++) {
            const int64_t offj = H.cblk[(size_t)(s0 + j)].fcolnum - H.cblk[(size_t)s0].fcolnum;
            const int64_t wj = H.cblk[(size_t)(s0 + j)].lcolnum - H.cblk[(size_t)(s0 + j)].fcolnum + 1;
            for (int64_t c = offj; c < offj + wj; c++)
              for (int64_t r = 0; r < offj; r++) {
                memcpy(hl + (size_t)(c * os + r) * eb, hu + (size_t)(r * os + c) * eb, eb);
                memcpy(hu + (size_t)(c * os + r) * eb, hl + (size_t)(r * os + c) * eb, eb);
              }
          }
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; t++) th.emplace_back(fix);
    fix();
    for (auto& t : th) t.join();
  }
  static const bool verbose = getenv("PASTIX_AMD_VERBOSE") != nullptr;
  if (verbose) fprintf(stderr, "pastix_amd: staged panels %s, part %d: %.3f GB in %.3f s\n", up ? "in" : "out", part, moved * 1e-9, now_s() - t_io);
  return PASTIX_AMD_OK;
}

int pastix_amd_upload_packed(pastix_amd_plan_t* p, const void* L, const void* U) {
  if (p) p->refillable = false;
  if (!p || !L) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  double t0 = now_s();
  p->factored = false;
  if (p->split.active) {
    int r = split_io(p, true, nullptr, nullptr, (void*)L, (void*)U);
    p->stats.h2d_time = now_s() - t0;
    return r;
  }
  if (p->cplx) {
    int r = z_transfer(p, true, (void*)L, p->dL, p->dLi, 0, p->host.coefnbr);
    if (r) return r;
    if (p->dU) {   // second arena (L*D / U) starts from zeros unless provided
      if (U) { if ((r = z_transfer(p, true, (void*)U, p->dU, p->dUi, 0, p->host.coefnbr))) return r; }
    }
  } else {
    HIPCHK(hipMemcpy(p->dL, L, p->host.coefnbr * p->esz, hipMemcpyHostToDevice));
    if (p->dU && U) HIPCHK(hipMemcpy(p->dU, U, p->host.coefnbr * p->esz, hipMemcpyHostToDevice));
  }
  p->stats.h2d_time = now_s() - t0;
  return PASTIX_AMD_OK;
}

int pastix_amd_download_packed(pastix_amd_plan_t* p, void* L, void* U) {
  if (!p || !L) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  double t0 = now_s();
  HIPCHK(hipStreamSynchronize(p->stream));
  if (p->split.active) {
    int r = split_io(p, false, nullptr, nullptr, L, U);
    p->stats.d2h_time = now_s() - t0;
    return r;
  }
  if (p->cplx) {
    int r = z_transfer(p, false, L, p->dL, p->dLi, 0, p->host.coefnbr);
    if (r) return r;
    if (p->dU && U && (r = z_transfer(p, false, U, p->dU, p->dUi, 0, p->host.coefnbr))) return r;
  } else {
    HIPCHK(hipMemcpy(L, p->dL, p->host.coefnbr * p->esz, hipMemcpyDeviceToHost));
    if (p->dU && U) HIPCHK(hipMemcpy(U, p->dU, p->host.coefnbr * p->esz, hipMemcpyDeviceToHost));
  }
  p->stats.d2h_time = now_s() - t0;
  return PASTIX_AMD_OK;
}

// distributed plans: the fan-in buffers (role 2: accumulators for remote cblks, add_contrib_target's lazily zero-allocated
// ftgttab[].coeftab, sopalin_compute.c:622-638) start a factorization from zeros; an upload of the owned panels clears them
static int zero_fanin_buffers(pastix_amd_plan_t* p) {
  const Plan& H = p->host;
  double* arenas[4] = {p->dL, p->dU, p->dLi, p->dUi};
  for (int64_t k = 0; k < H.cblknbr;) {
    if (H.role[(size_t)k] != 2) { k++; continue; }
    int64_t k1 = k + 1;
    while (k1 < H.cblknbr && H.role[(size_t)k1] != 1) k1++;
    const size_t bytes = (size_t)(H.poff[(size_t)k1] - H.poff[(size_t)k]) * p->esz;
    for (double* a : arenas)
      if (a && bytes) HIPCHK(hipMemsetAsync(p->at(a, H.poff[(size_t)k]), 0, bytes, p->stream));
    k = k1;
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_upload_tabs(pastix_amd_plan_t* p, void* const* coeftab, void* const* ucoeftab) {
  if (p) p->refillable = false;
  if (!p || !coeftab) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  const Plan& H = p->host;
  double t0 = now_s();
  p->factored = false;
  if (p->distributed) { int r = zero_fanin_buffers(p); if (r) return r; }
  if (p->split.active && p->cplx) {
    int r = split_io(p, true, coeftab, ucoeftab, nullptr, nullptr);
    p->stats.h2d_time = now_s() - t0;
    return r;
  }
  if (!p->cplx) {
    const int64_t nitem = p->split.active ? p->split.ocblknbr : H.cblknbr;
    for (int64_t k = 0; k < nitem; k++)
      if (H.role[(size_t)(p->split.active ? p->split.first[(size_t)k] : k)] == 1 && (!coeftab[k] || (p->dU && ucoeftab && !ucoeftab[k])))
        return PASTIX_AMD_ERR_BADPARAMETER;
    const int r = staged_tabs_io(p, true, coeftab, ucoeftab);
    p->stats.h2d_time = now_s() - t0;
    return r;
  }
  for (int64_t k = 0; k < H.cblknbr; k++) {
    size_t bytes = (size_t)(H.poff[k + 1] - H.poff[k]) * p->esz;
    if (H.role[k] != 1) continue;
    if (!coeftab[k]) return PASTIX_AMD_ERR_BADPARAMETER;
    if (p->cplx) {
      int r = z_transfer(p, true, coeftab[k], p->dL, p->dLi, H.poff[k], H.poff[k + 1] - H.poff[k]);
      if (r) return r;
      if (p->dU && ucoeftab && ucoeftab[k] &&
          (r = z_transfer(p, true, ucoeftab[k], p->dU, p->dUi, H.poff[k], H.poff[k + 1] - H.poff[k])))
        return r;
      continue;
    }
    HIPCHK(hipMemcpyAsync(p->at(p->dL, H.poff[k]), coeftab[k], bytes, hipMemcpyHostToDevice, p->stream));
    if (p->dU && ucoeftab && ucoeftab[k])
      HIPCHK(hipMemcpyAsync(p->at(p->dU, H.poff[k]), ucoeftab[k], bytes, hipMemcpyHostToDevice, p->stream));
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  p->stats.h2d_time = now_s() - t0;
  return PASTIX_AMD_OK;
}

int pastix_amd_download_tabs(pastix_amd_plan_t* p, void* const* coeftab, void* const* ucoeftab) {
  if (!p || !coeftab) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  const Plan& H = p->host;
  double t0 = now_s();
  if (p->split.active && p->cplx) {
    HIPCHK(hipStreamSynchronize(p->stream));
    int r = split_io(p, false, coeftab, ucoeftab, nullptr, nullptr);
    p->stats.d2h_time = now_s() - t0;
    return r;
  }
  if (!p->cplx) {
    const int64_t nitem = p->split.active ? p->split.ocblknbr : H.cblknbr;
    for (int64_t k = 0; k < nitem; k++)
      if (H.role[(size_t)(p->split.active ? p->split.first[(size_t)k] : k)] == 1 && (!coeftab[k] || (p->dU && ucoeftab && !ucoeftab[k])))
        return PASTIX_AMD_ERR_BADPARAMETER;
    HIPCHK(hipStreamSynchronize(p->stream));
    // (early_done: this very caller's panels below the run went home during the factorization, pastix_amd_factorize)
    const bool rest = p->early_done && p->early_tab == coeftab && p->early_utab == ucoeftab;
    const double early_s = rest ? p->stats.d2h_time : 0.0;
    const int r = staged_tabs_io(p, false, coeftab, ucoeftab, rest ? 2 : 0);
    p->early_done = false;
    p->stats.d2h_time = now_s() - t0 + early_s;
    return r;
  }
  for (int64_t k = 0; k < H.cblknbr; k++) {
    size_t bytes = (size_t)(H.poff[k + 1] - H.poff[k]) * p->esz;
    if (H.role[k] != 1) continue;
    if (!coeftab[k]) return PASTIX_AMD_ERR_BADPARAMETER;
    if (p->cplx) {
      int r = z_transfer(p, false, coeftab[k], p->dL, p->dLi, H.poff[k], H.poff[k + 1] - H.poff[k]);
      if (r) return r;
      if (p->dU && ucoeftab && ucoeftab[k] &&
          (r = z_transfer(p, false, ucoeftab[k], p->dU, p->dUi, H.poff[k], H.poff[k + 1] - H.poff[k])))
        return r;
      continue;
    }
    HIPCHK(hipMemcpyAsync(coeftab[k], p->at(p->dL, H.poff[k]), bytes, hipMemcpyDeviceToHost, p->stream));
    if (p->dU && ucoeftab && ucoeftab[k])
      HIPCHK(hipMemcpyAsync(ucoeftab[k], p->at(p->dU, H.poff[k]), bytes, hipMemcpyDeviceToHost, p->stream));
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  p->stats.d2h_time = now_s() - t0;
  return PASTIX_AMD_OK;
}

int pastix_amd_download_cblk(pastix_amd_plan_t* p, pastix_amd_int_t k, void* L, void* U) {
  if (p && p->split.active) {
    if (!L || k < 0 || k >= p->split.ocblknbr || p->host.role[(size_t)p->split.first[k]] != 1) return PASTIX_AMD_ERR_BADPARAMETER;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    return split_cblk_io(p, k, false, L, U);
  }
  if (!p || !L || k < 0 || k >= p->host.cblknbr || p->host.role[k] != 1) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  const Plan& H = p->host;
  const int64_t off = H.poff[k], cnt = H.poff[k + 1] - H.poff[k];
  HIPCHK(hipStreamSynchronize(p->stream));
  if (p->cplx) {
    int r = z_transfer(p, false, L, p->dL, p->dLi, off, cnt);
    if (r) return r;
    if (U && p->dU && (r = z_transfer(p, false, U, p->dU, p->dUi, off, cnt))) return r;
    return PASTIX_AMD_OK;
  }
  HIPCHK(hipMemcpy(L, p->at(p->dL, off), cnt * p->esz, hipMemcpyDeviceToHost));
  if (U && p->dU) HIPCHK(hipMemcpy(U, p->at(p->dU, off), cnt * p->esz, hipMemcpyDeviceToHost));
  return PASTIX_AMD_OK;
}

// CoefMatrix_Init + Csc2solv_cblk (coefinit.c:283-296, csc_intern_solve.c:65-132): zero the panels,
// then place every entry of the permuted (and, for symmetric input, mirrored) matrix whose row is
// >= fcolnum of its column's cblk into its blok.  Destinations are computed on the host (binary
// search over the cblk's bloks instead of the reference's linear walk :94-99), the scatter runs on
// the device.
int pastix_amd_fill_csc(pastix_amd_plan_t* p, int sym, pastix_amd_int_t n, const pastix_amd_int_t* colptr,
                        const pastix_amd_int_t* rows, const void* vals_, const pastix_amd_int_t* perm) {
  if (!p || !colptr || !rows || !vals_ || !perm) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  if (n != H.ncol) return PASTIX_AMD_ERR_BADPARAMETER;
  const double* vals = (const double*)vals_;   // complex: interleaved (re,im); single-precision plans: float values
  const int vs = p->cplx ? 2 : 1;
  auto val_at = [&](int64_t q) -> double { return p->f32 ? (double)((const float*)vals_)[q] : vals[vs * q]; };
  std::vector<double> valLi, valUi;
  HIPCHK(hipSetDevice(p->device));
  std::vector<int32_t> col2cblk((size_t)n);
  for (int64_t k = 0; k < H.cblknbr; k++)
    for (int64_t j = H.cblk[k].fcolnum; j <= H.cblk[k].lcolnum; j++) col2cblk[j] = (int32_t)k;
  const int64_t nnz = colptr[n] - 1;
  std::vector<int64_t> idxL, idxU;
  std::vector<double> valL, valU;
  auto locate = [&](int64_t pr, int64_t pc, bool offdiag_only) -> int64_t {
    const int64_t kc = col2cblk[pc];
    if (H.role[kc] != 1 || pr < H.cblk[kc].fcolnum) return -1;   // only owned panels are filled
    int64_t lo = H.cblk[kc].bloknum, hi = H.cblk[kc + 1].bloknum - 1, fb = lo;
    while (lo < hi) {
      int64_t mid = (lo + hi + 1) >> 1;
      if (H.blok[mid].frownum <= pr) lo = mid; else hi = mid - 1;
    }
    if (H.blok[lo].frownum > pr || H.blok[lo].lrownum < pr) return -1;
    if (offdiag_only && lo == fb) return -1;
    return H.poff[kc] + H.blok[lo].coefind + (pr - H.blok[lo].frownum) + (pc - H.cblk[kc].fcolnum) * H.cblk[kc].stride;
  };
  const bool lu = H.factotype == PASTIX_AMD_FACT_LU;
  const bool keep_upper = p->split.active && !lu && !p->split.upper.empty();
  std::vector<int64_t> sub2orig;
  if (keep_upper) {
    sub2orig.resize((size_t)H.cblknbr);
    for (int64_t k = 0; k < p->split.ocblknbr; k++) {
      for (int64_t q = p->split.first[k]; q < p->split.first[k + 1]; q++) sub2orig[(size_t)q] = k;
      std::fill(p->split.upper[(size_t)k].begin(), p->split.upper[(size_t)k].end(), (unsigned char)0);
    }
  }
  // The columns are dealt to host threads in contiguous ranges of about equal entry counts; every thread fills lists of
  // its own, which are joined in range order: the same lists as one thread makes (3.2e7 entries at 200^3: 0.6 s on one).
  struct Lists { std::vector<int64_t> idxL, idxU; std::vector<double> valL, valU, valLi, valUi; int err = 0; };
  const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, nnz / 200000 + 1}));
  std::vector<Lists> part((size_t)nthr);
  std::vector<int64_t> jcut((size_t)nthr + 1, n);
  jcut[0] = 0;
  for (int t = 1; t < nthr; t++)
    jcut[(size_t)t] = std::upper_bound(colptr, colptr + n, 1 + nnz * t / nthr) - colptr - 1;
  auto fill_range = [&](int t) {
    Lists& O = part[(size_t)t];
    try {
      const int64_t j0 = std::max<int64_t>(jcut[(size_t)t], 0), j1 = std::max(j0, jcut[(size_t)t + 1]);
      O.idxL.reserve((size_t)(colptr[j1] - colptr[j0]) * (sym ? 2 : 1));
      O.valL.reserve((size_t)(colptr[j1] - colptr[j0]) * (sym ? 2 : 1));
      for (int64_t j = j0; j < j1; j++)
        for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
          const int64_t i = rows[q] - 1;
          if (i < 0 || i >= n) { O.err = PASTIX_AMD_ERR_BADPARAMETER; return; }
          const int npass = (sym && i != j) ? 2 : 1;
          for (int pass = 0; pass < npass; pass++) {
            const int64_t pr = perm[pass ? j : i], pc = perm[pass ? i : j];
            int64_t d = locate(pr, pc, false);
            if (d >= 0) {
              O.idxL.push_back(d);
              O.valL.push_back(val_at(q));
              // Hermitian input: the mirrored entry is the conjugate (CscOrdistrib type 'H', pastix.c:3309)
              if (p->cplx) O.valLi.push_back((pass && H.factotype == PASTIX_AMD_FACT_LDLH) ? -vals[2 * q + 1] : vals[2 * q + 1]);
            } else if (keep_upper) {
              // a re-cut cblk: entries of its diagonal blok above the column group of their column stay on the host
              // (SplitMap::upper) -- the reference's coeftab carries them, unread, from the fill to the caller.  (Distinct
              // matrix entries are distinct elements: no two threads write the same one.)
              const int64_t ko = sub2orig[(size_t)col2cblk[pc]];
              std::vector<unsigned char>& up = p->split.upper[(size_t)ko];
              const int64_t of = H.cblk[(size_t)p->split.first[ko]].fcolnum, ow = p->split.owidth[ko];
              if (!up.empty() && pr >= of && pr < H.cblk[col2cblk[pc]].fcolnum) {
                const size_t e = (size_t)((pc - of) * ow + (pr - of));
                if (p->f32) {
                  ((float*)up.data())[e] = (float)val_at(q);
                } else {
                  double* ud = (double*)up.data();
                  ud[e * vs] = vals[vs * q];
                  if (p->cplx) ud[e * vs + 1] = (pass && H.factotype == PASTIX_AMD_FACT_LDLH) ? -vals[2 * q + 1] : vals[2 * q + 1];
                }
              }
            }
            if (lu) {
              d = locate(pc, pr, true);
              if (d >= 0) { O.idxU.push_back(d); O.valU.push_back(val_at(q)); if (p->cplx) O.valUi.push_back(vals[2 * q + 1]); }
            }
          }
        }
    } catch (const std::bad_alloc&) { O.err = PASTIX_AMD_ERR_ALLOC; }
  };
  try {
    {
      std::vector<std::thread> th;
      std::vector<int> inline_ranges{0};
      for (int t = 1; t < nthr; t++) {
        try { th.emplace_back(fill_range, t); } catch (const std::system_error&) { inline_ranges.push_back(t); }
      }
      for (const int t : inline_ranges) fill_range(t);       // (a range whose thread could not be started runs here)
      for (auto& x : th) x.join();
    }
    for (const Lists& O : part) if (O.err) return O.err;
    auto join = [&](auto& dst, auto Lists::*m) {
      size_t tot = 0;
      for (const Lists& O : part) tot += (O.*m).size();
      dst.reserve(tot);
      for (const Lists& O : part) dst.insert(dst.end(), (O.*m).begin(), (O.*m).end());
    };
    join(idxL, &Lists::idxL); join(valL, &Lists::valL); join(valLi, &Lists::valLi);
    join(idxU, &Lists::idxU); join(valU, &Lists::valU); join(valUi, &Lists::valUi);
    std::vector<Lists>().swap(part);
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  } catch (const std::system_error&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  auto cache = [&](std::vector<int64_t>& idx, std::vector<double>& val, int64_t** di, double** dv, int64_t* cnt) -> int {
    (void)hipFree(*di); (void)hipFree(*dv);
    *di = nullptr; *dv = nullptr; *cnt = (int64_t)idx.size();
    if (idx.empty()) return 0;
    HIPCHK(hipMalloc((void**)di, idx.size() * sizeof(int64_t)));
    HIPCHK(hipMalloc((void**)dv, val.size() * sizeof(double)));
    HIPCHK(hipMemcpy(*di, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(*dv, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
    return 0;
  };
  int r;
  p->fillBaseL = p->fillBaseU = 0.0;
  if ((r = cache(idxL, valL, &p->dFillIdxL, &p->dFillValL, &p->nFillL))) return r;
  if ((r = cache(idxU, valU, &p->dFillIdxU, &p->dFillValU, &p->nFillU))) return r;
  if (p->cplx) {
    (void)hipFree(p->dFillValLi);
    p->dFillValLi = nullptr;
    if (!valLi.empty()) {
      HIPCHK(hipMalloc((void**)&p->dFillValLi, valLi.size() * sizeof(double)));
      HIPCHK(hipMemcpy(p->dFillValLi, valLi.data(), valLi.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    (void)hipFree(p->dFillValUi);
    p->dFillValUi = nullptr;
    if (!valUi.empty()) {
      HIPCHK(hipMalloc((void**)&p->dFillValUi, valUi.size() * sizeof(double)));
      HIPCHK(hipMemcpy(p->dFillValUi, valUi.data(), valUi.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  return pastix_amd_refill(p);
}

// The reference's "fake factorisation" fill (CoefMatrix_Init with IPARM_FILL_MATRIX = API_YES, coefinit.c:343-443): no
// CSC -- every entry of coeftab is 1, of ucoeftab 2, the diagonal of every diagonal blok gnodenbr^2, and for LU the
// strictly upper part of coeftab's diagonal blok is 2 (the copy of ucoeftab's lower part, :431-441).  Cached like
// pastix_amd_fill_csc (pastix_amd_refill re-applies it).  One GPU; cblks wider than 128 columns are re-cut like everywhere.
int pastix_amd_fill_fake(pastix_amd_plan_t* p, pastix_amd_int_t gnodenbr) {
  if (!p || gnodenbr < 1) return PASTIX_AMD_ERR_BADPARAMETER;
  if (p->distributed) return PASTIX_AMD_ERR_UNSUPPORTED;
  for (auto& up : p->split.upper) {                // (re-cut cblks, LLt / LDLt: every coeftab entry is 1, also above the groups)
    if (p->f32) { for (size_t e = 0; e < up.size() / 4; e++) ((float*)up.data())[e] = 1.0f; }
    else { for (size_t e = 0; e < up.size() / 8; e++) ((double*)up.data())[e] = (p->cplx && (e & 1)) ? 0.0 : 1.0; }
  }
  const Plan& H = p->host;
  HIPCHK(hipSetDevice(p->device));
  const bool lu = H.factotype == PASTIX_AMD_FACT_LU;
  std::vector<int64_t> idx;
  std::vector<double> val;
  for (int64_t k = 0; k < H.cblknbr; k++) {
    const int64_t w = H.cblk[k].lcolnum - H.cblk[k].fcolnum + 1, sd = H.cblk[k].stride;
    for (int64_t c = 0; c < w; c++) {
      idx.push_back(H.poff[k] + c + c * sd);
      val.push_back((double)gnodenbr * (double)gnodenbr);
      if (lu)
        for (int64_t r = c + 1; r < w; r++) { idx.push_back(H.poff[k] + c + r * sd); val.push_back(2.0); }
    }
  }
  (void)hipFree(p->dFillIdxL); (void)hipFree(p->dFillValL); (void)hipFree(p->dFillIdxU); (void)hipFree(p->dFillValU);
  (void)hipFree(p->dFillValLi); (void)hipFree(p->dFillValUi);
  p->dFillIdxL = p->dFillIdxU = nullptr;
  p->dFillValL = p->dFillValU = p->dFillValLi = p->dFillValUi = nullptr;
  p->nFillL = p->nFillU = 0;
  HIPCHK(hipMalloc((void**)&p->dFillIdxL, idx.size() * sizeof(int64_t)));
  HIPCHK(hipMalloc((void**)&p->dFillValL, val.size() * sizeof(double)));
  HIPCHK(hipMemcpy(p->dFillIdxL, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->dFillValL, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
  p->nFillL = (int64_t)idx.size();
  p->fillBaseL = 1.0;
  p->fillBaseU = lu ? 2.0 : 0.0;
  return pastix_amd_refill(p);
}

// Re-apply the cached coefficient fill (device only): zero the panels, scatter the values.
int pastix_amd_refill(pastix_amd_plan_t* p) {
  if (!p || (!p->dFillIdxL && p->nFillL != 0) || !p->dL) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  p->factored = false;
  HIPCHK(hipSetDevice(p->device));
  if (p->f32) {
    float *fL = (float*)p->dL, *fU = (float*)p->dU;
    if (p->fillBaseL != 0.0) launch_fill_const_s(p->stream, fL, H.coefnbr, (float)p->fillBaseL);
    else HIPCHK(hipMemsetAsync(fL, 0, H.coefnbr * sizeof(float), p->stream));
    if (fU && p->fillBaseU != 0.0) launch_fill_const_s(p->stream, fU, H.coefnbr, (float)p->fillBaseU);
    else if (fU) HIPCHK(hipMemsetAsync(fU, 0, H.coefnbr * sizeof(float), p->stream));
    launch_scatter_s(p->stream, fL, p->dFillIdxL, p->dFillValL, p->nFillL);
    if (fU && p->nFillU) launch_scatter_s(p->stream, fU, p->dFillIdxU, p->dFillValU, p->nFillU);
    HIPCHK(hipStreamSynchronize(p->stream));
    p->refillable = p->own_arena;                              // (an external arena is the caller's to write at any time)
    return PASTIX_AMD_OK;
  }
  if (p->fillBaseL != 0.0) launch_fill_const(p->stream, p->dL, H.coefnbr, p->fillBaseL);
  else HIPCHK(hipMemsetAsync(p->dL, 0, H.coefnbr * sizeof(double), p->stream));
  if (p->dU && p->fillBaseU != 0.0) launch_fill_const(p->stream, p->dU, H.coefnbr, p->fillBaseU);
  else if (p->dU) HIPCHK(hipMemsetAsync(p->dU, 0, H.coefnbr * sizeof(double), p->stream));
  if (p->dLi) HIPCHK(hipMemsetAsync(p->dLi, 0, H.coefnbr * sizeof(double), p->stream));
  if (p->dUi) HIPCHK(hipMemsetAsync(p->dUi, 0, H.coefnbr * sizeof(double), p->stream));
  launch_scatter(p->stream, p->dL, p->dFillIdxL, p->dFillValL, p->nFillL);
  if (p->cplx && p->dFillValLi) launch_scatter(p->stream, p->dLi, p->dFillIdxL, p->dFillValLi, p->nFillL);
  if (p->dU && p->nFillU) launch_scatter(p->stream, p->dU, p->dFillIdxU, p->dFillValU, p->nFillU);
  if (p->dUi && p->nFillU && p->dFillValUi) launch_scatter(p->stream, p->dUi, p->dFillIdxU, p->dFillValUi, p->nFillU);
  HIPCHK(hipStreamSynchronize(p->stream));
  p->refillable = p->own_arena;
  return PASTIX_AMD_OK;
}

// tasks [b, e) of launch slot `slot`: the quadrant tasks at the end of the slot's urgent range and at the end of its bulk
// range go to k_update_small, the rest to k_update
static void launch_update_range(pastix_amd_plan_t* p, hipStream_t s, int slot, int64_t b, int64_t e, bool urgent) {
  const Plan& H = p->host;
  if (p->f32) { launch_update_s(s, p->arenas(), p->dTasks + b, p->dPieces, e - b, urgent); return; }
  // the slot's ranges: [t0, us) urgent, [us, tu) urgent quadrants, [tu, sb) bulk, [sb, t1) bulk quadrants
  const int64_t cut[5] = {H.slot_task_ptr[(size_t)slot], H.slot_usmall_begin[(size_t)slot], H.slot_urgent_end[(size_t)slot],
                          H.slot_small_begin[(size_t)slot], H.slot_task_ptr[(size_t)slot + 1]};
  for (int k = 0; k < 4; k++) {
    const int64_t lo = std::max(b, cut[k]), hi = std::min(e, cut[k + 1]);
    if (hi <= lo) continue;
    if (k & 1) launch_update_small(s, p->arenas(), p->dTasks + lo, p->dPieces, hi - lo, urgent);
    else launch_update(s, p->arenas(), p->dTasks + lo, p->dPieces, hi - lo, urgent);
  }
}

int pastix_amd_factorize_begin(pastix_amd_plan_t* p, double critere) {
  if (!p || !p->dL || (p->host.factotype != PASTIX_AMD_FACT_LLT && !p->dU)) return PASTIX_AMD_ERR_BADPARAMETER;
  HIPCHK(hipSetDevice(p->device));
  hipStream_t s = p->stream;
  HIPCHK(hipMemsetAsync(p->dNbpivot, 0, 2 * sizeof(long long), s));
  HIPCHK(hipMemsetAsync(p->dErr, 0, sizeof(int), s));
  HIPCHK(hipEventRecord(p->ev0, s));
  p->refillable = false;                                     // (from here on the panels are being overwritten)
  p->ev1_recorded = false;
  p->nupd_run = 0;
  p->run_used = false;
  p->launch_events = true;
  p->nupdB_run = 0;
  p->crit_run = critere;
  // Level-stepped use (distributed plans): the same urgent / bulk split over two streams as pastix_amd_factorize's
  // two-stream driver, one level at a time.  (pastix_amd_factorize sets `overlapped` itself after this call.)
  p->staged_overlap = p->distributed && p->stream2;
  p->staged_lastB = -1;
  if (p->staged_overlap) { p->overlapped = true; p->overlap_mode = 1; }
  return PASTIX_AMD_OK;
}

// one dependency level: contributions scheduled into slot l, then the owned cblks of level l
static int launch_panels(pastix_amd_plan_t* p, int l);

int pastix_amd_factorize_level(pastix_amd_plan_t* p, int l, int phase) {
  if (!p || l < 0 || l >= p->host.nlevels) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  hipStream_t s = p->stream;
  const int64_t t0 = H.slot_task_ptr[l], t1 = H.slot_task_ptr[l + 1];
  if (p->staged_overlap) {
    hipStream_t s2 = p->stream2;
    const int64_t tu = H.slot_urgent_end[l];
    if (phase != 2) {
      // A(l): targets of level l (among them every fan-in buffer of level l), on the caller's stream, which also
      // carries the exchange and the panel kernels; B(l): the rest, beside them on the second stream
      if (p->staged_lastB >= 0) { HIPCHK(hipStreamWaitEvent(s, p->evB[p->staged_lastB], 0)); p->staged_lastB = -1; }
      if (tu > t0) {
        HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run], s));
        launch_update_range(p, s, l, t0, tu, true);
        HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run + 1], s));
        p->nupd_run++;
      }
      if (t1 > tu) {
        if (l > 0) HIPCHK(hipStreamWaitEvent(s2, p->evP[l - 1], 0));
        else HIPCHK(hipStreamWaitEvent(s2, p->ev0, 0));
        HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run], s2));
        launch_update_range(p, s2, l, tu, t1, false);
        HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run + 1], s2));
        HIPCHK(hipEventRecord(p->evB[l], s2));
        p->staged_lastB = l;
        p->nupdB_run++;
      }
    }
    if (phase == 1) return PASTIX_AMD_OK;
    const int rc = launch_panels(p, l);
    HIPCHK(hipEventRecord(p->evP[l], s));
    return rc;
  }
  if (t1 > t0 && phase != 2) {
    HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run], s));
    launch_update_range(p, s, l, t0, t1, false);
    HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run + 1], s));
    p->nupd_run++;
  }
  if (phase == 1) return PASTIX_AMD_OK;
  return launch_panels(p, l);
}

static int launch_panels(pastix_amd_plan_t* p, int l) {
  const Plan& H = p->host;
  hipStream_t s = p->stream;
  const PanelTask* pt = p->dPanel + H.lvl_panel_ptr[l];
  const int64_t npt = H.lvl_panel_ptr[l + 1] - H.lvl_panel_ptr[l];
  const TrsmTask* tt = p->dTrsm + H.lvl_trsm_ptr[l];
  const int64_t ntt = H.lvl_trsm_ptr[l + 1] - H.lvl_trsm_ptr[l];
  if (p->f32) {
    const int lw = npt > 0 ? (int)H.panel_tasks[(size_t)H.lvl_panel_ptr[l]].width : 1;    // (sorted widest first)
    launch_diag_s(s, H.factotype, (float*)p->dL, (float*)p->dU, pt, npt, (float*)p->dDinv, p->crit_run, p->dNbpivot, p->dErr, lw);
    launch_trsm_s(s, H.factotype, (float*)p->dL, (float*)p->dU, tt, ntt, (const float*)p->dDinv, lw);
    return PASTIX_AMD_OK;
  }
  if (p->cplx) {
    if (H.factotype == PASTIX_AMD_FACT_LU) {
      launch_diag_zlu(s, p->arenas(), pt, npt, p->dDinv, p->crit_run, p->dNbpivot, p->maxw);
      launch_trsm_zlu(s, p->arenas(), tt, ntt, p->dDinv, p->maxw);
      return PASTIX_AMD_OK;
    }
    const bool herm = H.factotype == PASTIX_AMD_FACT_LDLH;
    launch_diag_zsy(s, herm, p->arenas(), pt, npt, p->dDinv, p->crit_run, p->dNbpivot, p->maxw);
    launch_trsm_zsy(s, herm, p->arenas(), tt, ntt, p->dDinv, p->maxw);
  } else if (H.factotype == PASTIX_AMD_FACT_LLT) {
    launch_diag_llt(s, p->dL, pt, npt, p->dDinv, p->crit_run, p->dNbpivot, p->dErr, p->maxw);
    launch_trsm_llt(s, p->dL, tt, ntt, p->dDinv, p->maxw);
  } else if (H.factotype == PASTIX_AMD_FACT_LDLT) {
    launch_diag_ldlt(s, p->dL, pt, npt, p->dDinv, p->crit_run, p->dNbpivot, p->maxw);
    launch_trsm_ldlt(s, p->dL, p->dU, tt, ntt, p->dDinv, p->maxw);
  } else {
    launch_diag_lu(s, p->dL, p->dU, pt, npt, p->dDinv, p->crit_run, p->dNbpivot);
    launch_trsm_lu(s, p->dL, p->dU, tt, ntt, p->dDinv, p->maxw);
  }
  return PASTIX_AMD_OK;
}

int pastix_amd_factorize_end(pastix_amd_plan_t* p, pastix_amd_stats_t* stats) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  hipStream_t s = p->stream;
  if (p->staged_overlap && p->nupdB_run > 0) {          // join the second stream
    HIPCHK(hipEventRecord(p->evB[0], p->stream2));
    HIPCHK(hipStreamWaitEvent(s, p->evB[0], 0));
  }
  if (!p->ev1_recorded) HIPCHK(hipEventRecord(p->ev1, s));
  p->ev1_recorded = false;
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  p->factored = true;
  p->fact_gen++;
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
  p->stats.fact_time = ms * 1e-3;
  double upd = 0;
  if (!p->launch_events) { p->nupd_run = 0; p->nupdB_run = 0; }     // (no per-launch events were recorded)
  {
    int i = 0;
    for (int l = 0; l < H.nlevels && i < p->nupd_run; l++) {
      const int64_t t0 = H.slot_task_ptr[l], t1 = p->overlapped ? H.slot_urgent_end[l] : H.slot_task_ptr[l + 1];
      if (t1 <= t0) continue;
      float m2 = 0;
      HIPCHK(hipEventElapsedTime(&m2, p->ev[2 * i], p->ev[2 * i + 1]));
      upd += m2 * 1e-3;
      if (H.opts.verbose >= 2)
        fprintf(stderr, "slot %4d: cblks %6lld tasks %7lld pieces %8lld maxpn %5d maxwork %.2e flops %.3e  %9.1f us  %8.1f GF/s\n", l,
                (long long)(H.lvl_panel_ptr[l + 1] - H.lvl_panel_ptr[l]), (long long)(t1 - t0),
                (long long)H.slot_pieces[l], (int)H.slot_maxpn[l], H.slot_maxwork[l], H.slot_flops[l], m2 * 1e3,
                H.slot_flops[l] / (m2 * 1e-3) * 1e-9);
      i++;
    }
  }
  if (p->nupdB_run > 0 && H.opts.verbose >= 2 && p->overlap_mode == 1 && !H.slot_mode_flops.empty()) {
    int i = 0;
    for (int l = 0; l < (p->run_used ? H.run_L0 : H.nlevels) && i < p->nupdB_run; l++) {
      const int64_t tu = H.slot_urgent_end[l], t1 = H.slot_task_ptr[l + 1];
      if (t1 <= tu) continue;
      float m2 = 0;
      HIPCHK(hipEventElapsedTime(&m2, p->evT[2 * i], p->evT[2 * i + 1]));
      const double* mf = &H.slot_mode_flops[(size_t)l * 3];
      const double fl = mf[0] + mf[1] + mf[2];
      fprintf(stderr, "bulk %4d: tasks %7lld (quadrant %6lld) flops %.4e  full %.4f edge %.4f partial %.4f  %9.1f us  %8.1f GF/s\n", l,
              (long long)(t1 - tu), (long long)(t1 - H.slot_small_begin[l]), fl, mf[0] / std::max(fl, 1.0), mf[1] / std::max(fl, 1.0),
              mf[2] / std::max(fl, 1.0), m2 * 1e3, fl / (m2 * 1e-3) * 1e-9);
      i++;
    }
  }
  if (p->nupdB_run > 0) {
    // the launches of the two streams overlap: count the time at least one of them was in flight
    std::vector<std::pair<float, float>> iv;
    for (int i = 0; i < p->nupd_run; i++) {
      float a = 0, b = 0;
      HIPCHK(hipEventElapsedTime(&a, p->ev0, p->ev[2 * i]));
      HIPCHK(hipEventElapsedTime(&b, p->ev0, p->ev[2 * i + 1]));
      iv.emplace_back(a, b);
    }
    for (int i = 0; i < p->nupdB_run; i++) {
      float a = 0, b = 0;
      HIPCHK(hipEventElapsedTime(&a, p->ev0, p->evT[2 * i]));
      HIPCHK(hipEventElapsedTime(&b, p->ev0, p->evT[2 * i + 1]));
      iv.emplace_back(a, b);
    }
    double sumA = 0, sumB = 0;
    for (int i = 0; i < (int)iv.size(); i++) (i < p->nupd_run ? sumA : sumB) += iv[i].second - iv[i].first;
    if (p->overlap_mode == 1) {           // ev[] = urgent launches (k_update<.,1>), evT[] = bulk launches
      p->stats.update_time_sum = sumB * 1e-3;
      p->stats.urgent_time_sum = sumA * 1e-3;
    } else {
      p->stats.update_time_sum = (sumA + sumB) * 1e-3;
      p->stats.urgent_time_sum = 0;
    }
    std::sort(iv.begin(), iv.end());
    double tot = 0;
    float cs = 0, ce = -1;
    for (auto& q : iv) {
      if (ce < 0 || q.first > ce) { if (ce >= 0) tot += ce - cs; cs = q.first; ce = q.second; }
      else ce = std::max(ce, q.second);
    }
    if (ce >= 0) tot += ce - cs;
    upd = tot * 1e-3;
  }
  if (p->nupdB_run == 0) { p->stats.update_time_sum = upd; p->stats.urgent_time_sum = 0; }
  p->stats.update_time = upd;
  const bool two_kernels = p->nupdB_run > 0 && p->overlap_mode == 1;
  p->stats.nupdate_launches = two_kernels ? p->nupdB_run : p->nupd_run + p->nupdB_run;
  p->stats.nurgent_launches = two_kernels ? p->nupd_run : 0;
  p->stats.urgent_flops = two_kernels ? p->host.urgent_flops : 0.0;
  p->stats.run_time = 0;
  p->stats.run_flops = 0;
  p->stats.run_tickets = 0;
  p->stats.run_first_level = -1;
  if (p->run_used && p->nupdB_run > 0 && p->launch_events) {
    float m2 = 0;
    HIPCHK(hipEventElapsedTime(&m2, p->evT[2 * (p->nupdB_run - 1)], p->evT[2 * (p->nupdB_run - 1) + 1]));
    p->stats.run_time = m2 * 1e-3;
    p->stats.run_flops = H.run_flops;
    p->stats.run_tickets = p->run_nticket;
    p->stats.run_first_level = H.run_L0;
  }
  if (p->run_used) {
    // the run's update launch carries the urgent tasks of its levels too: update_time_sum = the bulk launches below the
    // run + the run launch, carrying update_flops - urgent_flops with urgent_flops = those of the levels below the run
    double uf = 0;
    for (int l = 0; l < H.run_L0; l++) uf += H.slot_urgent_flops[(size_t)l];
    p->stats.urgent_flops = uf;
    if (H.opts.verbose >= 2 && p->nupdB_run > 0) {
      float m2 = 0;
      HIPCHK(hipEventElapsedTime(&m2, p->evT[2 * (p->nupdB_run - 1)], p->evT[2 * (p->nupdB_run - 1) + 1]));
      fprintf(stderr, "run  levels %d..%d: tickets %lld flops %.4e  %9.1f us  %8.1f GF/s\n", H.run_L0, H.nlevels - 1,
              (long long)p->run_nticket, H.run_flops, m2 * 1e3, H.run_flops / (m2 * 1e-3) * 1e-9);
    }
  }
  long long nb[2] = {0, 0};
  int err = 0;
  if (p->run_used && p->dRunProf) {
    // developer aid: [n update tickets, n diagonal tasks, n panel-solve tasks] then 4 stamps each (drawn, ready, done, 0),
    // 100 MHz ticks; tools/run_prof.py reads it
    if (const char* fn = dev_opt("run_prof")) {
      std::vector<long long> h(p->nRunProf);
      HIPCHK(hipMemcpy(h.data(), p->dRunProf, p->nRunProf * sizeof(long long), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(fn, "wb")) {
        const long long hdr[4] = {(long long)p->run_nticket, (long long)H.run_d.size(), 0, H.run_L0};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(h.data(), sizeof(long long), h.size(), f);
        std::vector<long long> cl(H.run_cat.size());
        for (size_t i = 0; i < cl.size(); i++) cl[i] = (long long)H.run_cat[i] | ((long long)H.run_lvl[i] << 8);
        fwrite(cl.data(), sizeof(long long), cl.size(), f);
        const std::vector<long long>& ft = p->runFeat;
        fwrite(ft.data(), sizeof(long long), ft.size(), f);
        fclose(f);
      }
    }
  }
  if (p->run_used) {
    int stuck = 0;
    HIPCHK(hipMemcpy(&stuck, p->runctl.ctl + RUN_STUCK, sizeof(int), hipMemcpyDeviceToHost));
    if (stuck) {
      fprintf(stderr, "pastix_amd: a wait inside the run launch expired (PASTIX_AMD_RUN_TIMEOUT): the factorization failed\n");
      run_debug_report(p);                               // (run_debug.cpp: where it stopped; the full replay with PASTIX_AMD_DEV=run_debug)
      p->factored = false;
      p->run_stuck = true;
      return PASTIX_AMD_ERR_DEVICE;
    }
  }
  HIPCHK(hipMemcpy(nb, p->dNbpivot, sizeof(nb), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&err, p->dErr, sizeof(err), hipMemcpyDeviceToHost));
  p->stats.nbpivot = nb[0];
  // IPARM_INERTIA: number of positive D entries, real LDLt only, else -1 (sopalin3d.c:1144-1160)
  p->stats.inertia = (H.factotype == PASTIX_AMD_FACT_LDLT && !p->cplx) ? nb[1] : -1;
  if (stats) *stats = p->stats;
  return err ? PASTIX_AMD_ERR_NUMERIC : PASTIX_AMD_OK;
}

// The device replacement of sopalin_smp's task loop (sopalin3d.c:790-1025): for every dependency
// level s: apply the contributions scheduled into slot s (k_update), then factorize the cblks of
// level s (k_diag + k_trsm).  Time is measured like DPARM_FACT_TIME: panels resident, first launch
// to last completion (sopalin3d.c:775,1031,1125-1132).
static int factorize_once(pastix_amd_plan_t* p, double critere, pastix_amd_stats_t* stats);
// The run schedule has a bounded wait (PASTIX_AMD_RUN_TIMEOUT, default 1 s + 1 s per 1e14 flop) so that nothing can hang the device.  On MI355X
// about one factorization in 200 trips it -- every running ticket's loads stand still until the waiting workgroups leave
// (not understood) --, so a factorization that stopped this way is REDONE on the level-by-level schedule when
// the input can be restored: the plan's cached fill (pastix_amd_fill_csc / refill were the last writers of the panels) here,
// the caller's host buffers in the one-shot entry points.  Otherwise PASTIX_AMD_ERR_DEVICE is what the caller gets.
int pastix_amd_factorize(pastix_amd_plan_t* p, double critere, pastix_amd_stats_t* stats) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  p->run_stuck = false;
  const bool refillable = p->refillable;
  p->restorable = refillable || p->caller_restores;
  int rc = factorize_once(p, critere, stats);
  if (rc == PASTIX_AMD_ERR_DEVICE && p->run_stuck && refillable) {
    fprintf(stderr, "pastix_amd: restoring the input from the cached fill and factorizing on the level-by-level schedule\n");
    if ((rc = pastix_amd_refill(p))) return rc;
    p->run_off_once = true;
    rc = factorize_once(p, critere, stats);
    p->run_off_once = false;
    p->run_stuck = false;
  }
  p->refillable = false;                                    // (the panels hold factors now)
  return rc;
}
static int factorize_once(pastix_amd_plan_t* p, double critere, pastix_amd_stats_t* stats) {
  if (!p) return PASTIX_AMD_ERR_BADPARAMETER;
  if (p->distributed) return PASTIX_AMD_ERR_BADPARAMETER;   // needs the fan-in exchange between levels
  // Two streams unless PASTIX_AMD_OVERLAP=0 (one stream).  Measured gain over one stream: 60^3 +25 %, 100^3 +8 %,
  // 160^3 +3 %, 200^3 +1 %.
  static const char* ov_env = getenv("PASTIX_AMD_OVERLAP");
  const int want = ov_env ? (atoi(ov_env) != 0) : 1;
  p->overlap_mode = want;
  p->overlapped = p->own_stream && p->stream2 && want != 0;
  int rc = pastix_amd_factorize_begin(p, critere);
  if (rc) return rc;
  if (!p->overlapped) {
    for (int l = 0; l < p->host.nlevels; l++)
      if ((rc = pastix_amd_factorize_level(p, l, 0))) return rc;
    return pastix_amd_factorize_end(p, stats);
  }
  const Plan& H = p->host;
  hipStream_t s1 = p->stream, s2 = p->stream2;
  // stream (high priority): contributions of slot l to level l (A), then the panel kernels of level l (P).  stream2: the
  // rest of slot l, whose sources are of level <= l-1 and whose targets are of level l+1 or later (B); it runs beside
  // A(l) and P(l).  Orders kept: B(l) after P(l-1) and after B(l-1) (stream order); A(l+1) after B(l): all writers of a
  // tile stay ordered, results do not depend on timing.  (Letting the tasks of B(l) whose tile B(l-1) does not touch
  // start beside the tail of B(l-1) on a third stream was built and measured: +1.9 % at 100^3, +0.8 % at 160^3 -- the next
  // launch mostly waits for the panel chain, not for a free slot -- and removed again.)
  // Per-launch timing events (statistics, bench.py's roofline line) are markers the command processor handles one by
  // one: PASTIX_AMD_LAUNCH_EVENTS=0 leaves them out (update_time / update_time_sum then read 0; -1.3 % at 100^3).
  static const bool ev_env = !getenv("PASTIX_AMD_LAUNCH_EVENTS") || atoi(getenv("PASTIX_AMD_LAUNCH_EVENTS")) != 0;
  p->launch_events = ev_env;
  const bool tev = p->launch_events;
  int lastN = -1;
  bool s2_used = false;
  // The run schedule (plan.h RunInfo): levels [L0, nlevels) are not launched level by level -- their panel tasks go to
  // resident workgroups started NOW (streams 3 and 4: the chip is idle or about to be, they are placed at once and stay),
  // their update tasks to one launch behind the last level below L0.  PASTIX_AMD_RUN=0 keeps the level-by-level
  // schedule on the same plan (both give bitwise the same factors).
  // Which factorizations take it: the ONE-kernel form (runctl.onek: every task of the run is a ticket of one launch on one
  // queue; real LLt / LDLt) always -- its soak is in profiles/r05/soak_*.txt --; the two-kernel form (LU, complex: resident
  // diagonal workers beside the tickets' launch), whose rare stop is not understood, only when a stopped
  // factorization can be REDONE, i.e. when the input can be restored (`restorable`: the plan's cached fill or the one-shot
  // entry's host buffers), or on request (options.run_schedule = 1, PASTIX_AMD_RUN=1): no caller sees PASTIX_AMD_ERR_DEVICE
  // for a well-posed input because of the schedule.
  const char* run_env = getenv("PASTIX_AMD_RUN");          // (read per call: tests switch it between factorizations)
  const bool run_asked = H.opts.run_schedule == 1 || (run_env && atoi(run_env) == 1);
  const bool use_run = p->run_ready && H.run_L0 >= 0 && !(run_env && atoi(run_env) == 0) && !p->run_off_once &&
                       (p->runctl.onek || p->restorable || run_asked);
  const int L0 = use_run ? H.run_L0 : H.nlevels;
  // bound of a single wait inside the run, in ticks of the 100 MHz clock.  Default: 1 s + 1 s per 1e14 flop of the
  // factorization (100^3: 1.06 s, 160^3: 2.8 s) -- no ticket waits longer than the levels below the run take, and a
  // factorization that trips the wait is redone (pastix_amd_factorize), so the wait is what the rare stop costs
  const long long run_limit = [&] {
    const char* e = getenv("PASTIX_AMD_RUN_TIMEOUT");
    const double sec = e ? atof(e) : 1.0 + H.fact_flops * 1e-14;
    return (long long)(std::max(sec, 0.001) * 1e8);
  }();
  p->run_used = use_run;
  const int run_nwk = (use_run && !p->runctl.onek) ? H.run_gd : 0;
  if (use_run) {
    HIPCHK(hipMemcpyAsync(p->dRunState, p->dRunImage, p->nRunState * sizeof(int32_t), hipMemcpyDeviceToDevice, s1));
    if (p->dRunProf && dev_opt("run_debug")) HIPCHK(hipMemsetAsync(p->dRunProf, 0, p->nRunProf * sizeof(long long), s1));
    *(volatile int*)p->hResident = 0;
    HIPCHK(hipEventRecord(p->evZ, s1));
    if (!p->runctl.onek) HIPCHK(hipStreamWaitEvent(p->stream3, p->evZ, 0));
    HIPCHK(hipStreamWaitEvent(s2, p->evZ, 0));
    if (!p->runctl.onek) {
      launch_run_panel(p->stream3, H.factotype, p->arenas(), p->dRunD, p->dRunInfo, H.run_gd, p->dDinv, critere, p->dNbpivot,
                       p->dErr, p->runctl, p->hResident, run_limit);
      HIPCHK(hipEventRecord(p->evS3, p->stream3));
    }
  }
  for (int l = 0; l < L0; l++) {
    const int64_t t0 = H.slot_task_ptr[l], tu = H.slot_urgent_end[l], t1 = H.slot_task_ptr[l + 1];
    if (lastN >= 0) { HIPCHK(hipStreamWaitEvent(s1, p->evB[lastN], 0)); lastN = -1; }
    if (tu > t0) {
      if (tev) HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run], s1));
      launch_update_range(p, s1, l, t0, tu, true);
      if (tev) HIPCHK(hipEventRecord(p->ev[2 * p->nupd_run + 1], s1));
      p->nupd_run++;
    }
    if ((rc = launch_panels(p, l))) return rc;
    HIPCHK(hipEventRecord(p->evP[l], s1));
    if (t1 > tu) {
      if (l > 0) HIPCHK(hipStreamWaitEvent(s2, p->evP[l - 1], 0));
      if (tev) HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run], s2));
      // inside a launch the tasks for level l+1 come first.  (Launching them separately so that A(l+1) waits for
      // them only was measured slower: smaller launches, same chain.)
      launch_update_range(p, s2, l, tu, t1, false);
      if (tev) HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run + 1], s2));
      HIPCHK(hipEventRecord(p->evB[l], s2));
      lastN = l;
      p->nupdB_run++;
      s2_used = true;
    }
  }
  if (use_run) {
    // the update launch of the run: behind the panels of level L0 - 1 (stream 1) and the bulk launch of slot L0 - 1 (this
    // stream's order).  The panel kernels must be RESIDENT before it starts: its workgroups wait for them while they hold
    // their slots.  The workgroups counted themselves in host memory long ago (they were launched first); checked here.
    if (L0 > 0) HIPCHK(hipStreamWaitEvent(s2, p->evP[L0 - 1], 0));
    const double tw0 = now_s();
    while (*(volatile int*)p->hResident < run_nwk) {
      if (now_s() - tw0 > 10.0) {
        fprintf(stderr, "pastix_amd: the run's panel workgroups did not start (%d of %d)\n", *(volatile int*)p->hResident, run_nwk);
        int one = 1;                                     // (lets the ones that did start leave their loops)
        (void)hipMemcpyAsync(p->runctl.ctl + RUN_STUCK, &one, sizeof(int), hipMemcpyHostToDevice, s2);
        (void)hipStreamSynchronize(s2);
        (void)hipStreamSynchronize(p->stream3);
        (void)hipStreamSynchronize(s1);
        p->run_stuck = true;                             // (pastix_amd_factorize / the one-shot entries redo it level by level)
        p->factored = false;
        return PASTIX_AMD_ERR_DEVICE;
      }
      sched_yield();
    }
    if (tev) HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run], s2));
    launch_run_update(s2, H.factotype, p->arenas(), p->dRunTasks, p->dPieces, p->dRunInfo, p->dRunCons, p->runctl, p->dDinv,
                      p->run_nticket, p->run_nwg, run_limit, p->dRunD, critere, p->dNbpivot, p->dErr);
    if (tev) HIPCHK(hipEventRecord(p->evT[2 * p->nupdB_run + 1], s2));
    p->nupdB_run++;
    s2_used = true;
    if (!p->runctl.onek) HIPCHK(hipStreamWaitEvent(s1, p->evS3, 0));
  }
  if (s2_used) {
    HIPCHK(hipEventRecord(p->evB[0], s2));
    HIPCHK(hipStreamWaitEvent(s1, p->evB[0], 0));
  }
  p->early_done = false;
  if (use_run && L0 > 0 && p->early_tab && !p->cplx && !p->distributed) {
    // (the one-shot entry points: the caller's thread has nothing to do until the run ends -- it carries the panels of the
    // levels below the run home meanwhile; they are final behind the panel kernels of level L0 - 1)
    if (!p->stream_io) HIPCHK(hipStreamCreateWithFlags(&p->stream_io, hipStreamNonBlocking));
    HIPCHK(hipEventRecord(p->ev1, s1));                 // (fact_time ends with the last kernel, not with these copies)
    p->ev1_recorded = true;
    HIPCHK(hipStreamWaitEvent(p->stream_io, p->evP[L0 - 1], 0));
    const double te = now_s();
    const int r = staged_tabs_io(p, false, p->early_tab, p->early_utab, 1, p->stream_io);
    if (r) { (void)hipDeviceSynchronize(); p->ev1_recorded = false; return r; }
    p->stats.d2h_time = now_s() - te;
    p->early_done = true;
  }
  return pastix_amd_factorize_end(p, stats);
}

// Forward / (diagonal) / backward substitution on the device-resident factors (real LLt, LDLt, LU; the data
// flow of up_down_smp, updo.c:114), x in permuted numbering.
// device tables of the solve (built on first use): per factorized cblk a SolveTask, its off-diagonal rows in chunks
// of 64 (forward) / 256 (backward) rows, the blok table and the panel-row -> global-row map
int pai_solve_tables(pastix_amd_plan_t* p) {
  const Plan& H = p->host;
  const int64_t nown = H.lvl_cblk_ptr[H.nlevels];          // cblks factorized here (all of them on one GPU)
  if (!p->dSolve) {
    std::vector<SolveTask> st((size_t)nown);
    std::vector<SolveChunk> ch, chB;
    p->lvl_chunk_ptr.assign((size_t)H.nlevels + 1, 0);
    p->lvl_chunkB_ptr.assign((size_t)H.nlevels + 1, 0);
    p->lvl_maxw.assign((size_t)H.nlevels, 1);
    p->lvl_nwide.assign((size_t)H.nlevels, 0);
    std::vector<int64_t> roff((size_t)nown + 1, 0);          // in level order, like st
    for (int64_t q = 0; q < nown; q++) roff[q + 1] = roff[q] + H.cblk[H.lvl_cblk[q]].stride;
    // panel rows per chunk: 64 forward (many workgroups on the tall top panels), 256 backward (one butterfly and
    // one set of atomics per 256 rows)
    const int32_t CH = 64;
    const int32_t CHB = 256;
    // thin levels: at most THIN cblks (real arithmetic, one GPU): explicit inverses; their runs go in one launch per sweep
    constexpr int64_t THIN = 64;
    std::vector<SolveChunk> thF, thB;
    std::vector<int32_t> thin_tasks;
    p->lvl_thin.assign((size_t)H.nlevels, 0);
    std::vector<int64_t> thF_ptr((size_t)H.nlevels + 1, 0), thB_ptr((size_t)H.nlevels + 1, 0);
    for (int l = 0; l < H.nlevels; l++) {
      p->lvl_chunk_ptr[l] = (int64_t)ch.size();
      p->lvl_chunkB_ptr[l] = (int64_t)chB.size();
      thF_ptr[l] = (int64_t)thF.size();
      thB_ptr[l] = (int64_t)thB.size();
      const int64_t ncl = H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l];
      const bool thin = !p->cplx && !p->distributed && ncl > 0 && ncl <= THIN;
      p->lvl_thin[l] = thin ? 1 : 0;
      for (int64_t q = H.lvl_cblk_ptr[l]; q < H.lvl_cblk_ptr[l + 1]; q++) {
        const int32_t k = H.lvl_cblk[q];
        const int32_t w = (int32_t)(H.cblk[k].lcolnum - H.cblk[k].fcolnum + 1), sd = (int32_t)H.cblk[k].stride;
        st[q] = SolveTask{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum, (int32_t)H.cblk[k].bloknum,
                          (int32_t)H.cblk[k + 1].bloknum, thin ? (int32_t)thin_tasks.size() : -1};
        const int32_t tix = st[q].thin;
        if (thin) {
          // the level's workgroup lists for the fused sweeps: the cblk's chunks and one workgroup without rows (so that
          // a cblk without off-diagonal rows is solved too), each knowing how many there are (the ticket)
          thin_tasks.push_back((int32_t)q);
          // (256 rows per workgroup in both sweeps: every workgroup applies the inverse for itself)
          const int32_t nf = (sd - w + CHB - 1) / CHB + 1, nb = nf;
          for (int32_t r = w; r < sd; r += CHB)
            thF.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum, (int32_t)H.cblk[k].bloknum,
                                     (int32_t)H.cblk[k + 1].bloknum, r, std::min(CHB, sd - r), roff[q], tix, nf, 0, 0, 0, 0});
          thF.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum, (int32_t)H.cblk[k].bloknum,
                                   (int32_t)H.cblk[k + 1].bloknum, w, 0, roff[q], tix, nf, 0, 0, 0, 0});
          for (int32_t r = w; r < sd; r += CHB)
            thB.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum, (int32_t)H.cblk[k].bloknum,
                                     (int32_t)H.cblk[k + 1].bloknum, r, std::min(CHB, sd - r), roff[q], tix, nb, 0, 0, 0, 0});
          thB.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum, (int32_t)H.cblk[k].bloknum,
                                   (int32_t)H.cblk[k + 1].bloknum, w, 0, roff[q], tix, nb, 0, 0, 0, 0});
        }
        p->lvl_maxw[l] = std::max(p->lvl_maxw[l], (int)w);
        // (the level's cblks are listed widest first, plan.cpp: a prefix)
        if (w > 64 && q == H.lvl_cblk_ptr[l] + p->lvl_nwide[l]) p->lvl_nwide[l]++;
        for (int32_t r = w; r < sd; r += CH) {
          const int32_t n = std::min(CH, sd - r);
          ch.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum,
              (int32_t)H.cblk[k].bloknum, (int32_t)H.cblk[k + 1].bloknum, r, n, roff[q], -1, 0, 0, 0, 0, 0});
        }
        for (int32_t r = w; r < sd; r += CHB) {
          const int32_t n = std::min(CHB, sd - r);
          chB.push_back(SolveChunk{H.poff[k], sd, w, (int32_t)H.cblk[k].fcolnum,
              (int32_t)H.cblk[k].bloknum, (int32_t)H.cblk[k + 1].bloknum, r, n, roff[q], -1, 0, 0, 0, 0, 0});
        }
      }
    }
    p->lvl_chunk_ptr[H.nlevels] = (int64_t)ch.size();
    p->lvl_chunkB_ptr[H.nlevels] = (int64_t)chB.size();
    thF_ptr[H.nlevels] = (int64_t)thF.size();
    thB_ptr[H.nlevels] = (int64_t)thB.size();
    // runs of consecutive thin levels: one launch per run and sweep (k_solve_thin_*), its workgroups in sweep order (the
    // backward list is re-ordered by descending level).  Inside a run the cblks synchronise one by one: a chunk knows
    // the thin cblks of its run that face its rows (forward: it contributes to them; backward: it reads their solution)
    p->runF_n.assign((size_t)H.nlevels, 0); p->runF_at.assign((size_t)H.nlevels, 0);
    p->runB_n.assign((size_t)H.nlevels, 0); p->runB_at.assign((size_t)H.nlevels, 0);
    std::vector<int32_t> tixc((size_t)H.cblknbr, -1), runc((size_t)H.cblknbr, -1);   // thin index / run (first level) of a cblk
    for (int l = 0; l < H.nlevels;) {
      if (!p->lvl_thin[l]) { l++; continue; }
      int e = l;
      while (e < H.nlevels && p->lvl_thin[e]) e++;
      for (int m = l; m < e; m++)
        for (int64_t q = H.lvl_cblk_ptr[m]; q < H.lvl_cblk_ptr[m + 1]; q++) {
          tixc[(size_t)H.lvl_cblk[q]] = st[q].thin;
          runc[(size_t)H.lvl_cblk[q]] = l;
        }
      p->runF_n[l] = thF_ptr[e] - thF_ptr[l];
      p->runF_at[l] = thF_ptr[l];
      l = e;
    }
    {
      std::vector<SolveChunk> rev;
      rev.reserve(thB.size());
      for (int l = H.nlevels - 1; l >= 0;) {
        if (!p->lvl_thin[l]) { l--; continue; }
        int e = l;
        while (e >= 0 && p->lvl_thin[e]) e--;
        const int64_t at = (int64_t)rev.size();
        for (int m = l; m > e; m--)
          for (int64_t i = thB_ptr[m]; i < thB_ptr[m + 1]; i++) rev.push_back(thB[(size_t)i]);
        p->runB_n[l] = (int64_t)rev.size() - at;
        p->runB_at[l] = at;
        l = e;
      }
      thB.swap(rev);
    }
    std::vector<int32_t> thin_tgt, thin_expect(thin_tasks.size(), 0);
    auto targets = [&](SolveChunk& c, bool count) {
      c.tptr = (int32_t)thin_tgt.size();
      const int32_t k = (int32_t)H.blok[c.fblok].cblknum;      // (the diagonal blok faces its own cblk)
      if (c.nrows > 0)
        for (int32_t b = c.fblok + 1; b < c.lblok; b++) {
          const int64_t r0 = H.blok[b].coefind, r1 = r0 + (H.blok[b].lrownum - H.blok[b].frownum + 1);
          if (r1 <= c.row0 || r0 >= c.row0 + c.nrows) continue;
          const int32_t f = (int32_t)H.blok[b].cblknum, t = tixc[(size_t)f];
          if (t < 0 || runc[(size_t)f] != runc[(size_t)k]) continue;
          if ((int32_t)thin_tgt.size() > c.tptr && thin_tgt.back() == t) continue;
          thin_tgt.push_back(t);
          if (count) thin_expect[(size_t)t]++;
        }
      c.tn = (int32_t)thin_tgt.size() - c.tptr;
    };
    for (auto& c : thF) targets(c, true);
    for (auto& c : thF) c.wait = thin_expect[(size_t)c.thin] > 0;
    for (auto& c : thB) targets(c, false);
    if (thin_tgt.empty()) thin_tgt.push_back(0);
    std::vector<DevBlok> bl((size_t)H.bloknbr);
    for (int64_t b = 0; b < H.bloknbr; b++)
      bl[b] = DevBlok{(int32_t)H.blok[b].frownum, (int32_t)H.blok[b].lrownum, (int32_t)H.blok[b].coefind};
    // built into locals and published only when complete: a failed build leaves the plan without solve tables
    SolveTask* dS = nullptr; DevBlok* dB = nullptr; SolveChunk *dC = nullptr, *dCB = nullptr; int32_t* dR = nullptr;
    SolveChunk *dTF = nullptr, *dTB = nullptr; int32_t* dTT = nullptr; double *dIF = nullptr, *dIB = nullptr; int* dTk = nullptr;
    int32_t *dTg = nullptr, *dEx = nullptr;
    int64_t* droff = nullptr;
    size_t nTk = 0;
    const int64_t nthin = (int64_t)thin_tasks.size();
    bool thin_ok = true;
    auto build = [&]() -> int {
      int r;
      if ((r = to_device(&dTF, thF))) return r;
      if ((r = to_device(&dTB, thB))) return r;
      if ((r = to_device(&dTT, thin_tasks))) return r;
      if ((r = to_device(&dTg, thin_tgt))) return r;
      if ((r = to_device(&dEx, thin_expect))) return r;
      if (nthin > 0) {
        // The thin levels' storage is 2 x 128 x 128 doubles + ~4 KB of flags per thin cblk whatever its width: a layout
        // with long chains of narrow cblks (banded matrices, blend layouts cut fine) can ask for tens of GB.  It is an
        // accelerator, not a requirement: beyond a quarter of the free memory, or if the allocation fails, the solve keeps
        // the per-level kernels for every level (thin_ok = false below) instead of failing.
        const size_t ib = (size_t)nthin * 128 * 128 * sizeof(double);
        nTk = (size_t)nthin * 3 + 64 + 2 * (size_t)nthin * 64 * 8;
        size_t fr = 0, tot = 0;
        bool ok = hipMemGetInfo(&fr, &tot) == hipSuccess && 2 * ib + nTk * sizeof(int) <= fr / 4;
        ok = ok && hipMalloc((void**)&dIF, ib) == hipSuccess && hipMalloc((void**)&dIB, ib) == hipSuccess &&
             // 2 x nthin tickets, nthin forward counters, the "stuck" flag, then per sweep nthin x 8 padded flags (kernels.hip)
             hipMalloc((void**)&dTk, nTk * sizeof(int)) == hipSuccess;
        if (ok) {
          HIPCHK(hipMemset(dIF, 0, ib));
          HIPCHK(hipMemset(dIB, 0, ib));
        } else {
          (void)hipGetLastError();
          (void)hipFree(dIF); (void)hipFree(dIB); (void)hipFree(dTk);
          dIF = dIB = nullptr; dTk = nullptr; nTk = 0;
          thin_ok = false;
          if (H.opts.verbose >= 1) fprintf(stderr, "pastix_amd: %lld thin cblks: no room for their inverses, the solve keeps per-level kernels\n", (long long)nthin);
        }
      }
      if ((r = to_device(&dS, st))) return r;
      if ((r = to_device(&dB, bl))) return r;
      if ((r = to_device(&dC, ch))) return r;
      if ((r = to_device(&dCB, chB))) return r;
      if ((r = to_device(&droff, roff))) return r;
      HIPCHK(hipMalloc((void**)&dR, (size_t)std::max<int64_t>(roff[nown], 1) * sizeof(int32_t)));
      launch_solve_rowidx(p->stream, dS, nown, droff, dB, dR);
      HIPCHK(hipStreamSynchronize(p->stream));
      HIPCHK(hipGetLastError());
      return 0;
    };
    const int rb = build();
    (void)hipFree(droff);
    if (rb) {
      (void)hipFree(dS); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dCB); (void)hipFree(dR);
      (void)hipFree(dTF); (void)hipFree(dTB); (void)hipFree(dTT); (void)hipFree(dIF); (void)hipFree(dIB); (void)hipFree(dTk);
      (void)hipFree(dTg); (void)hipFree(dEx);
      return rb;
    }
    p->dSolve = dS; p->dBlok = dB; p->dChunk = dC; p->dChunkB = dCB; p->dRidx = dR;
    p->dThinF = dTF; p->dThinB = dTB; p->dThinTasks = dTT; p->dInvF = dIF; p->dInvB = dIB; p->dTicket = dTk;
    p->nthin = thin_ok ? nthin : 0;                 // (0: every level through the per-level kernels)
    p->nTicket = nTk;
    p->dThinTgt = dTg; p->dThinExpect = dEx;
    p->inv_gen = -1;
  }
  if (p->nthin > 0 && p->inv_gen != p->fact_gen) {
    // the inverses of the thin cblks' diagonal bloks, once per factorization (k_solve_inv)
    const int unit = H.factotype != PASTIX_AMD_FACT_LLT;
    launch_solve_inv(p->stream, p->dL, p->f32, p->dSolve, p->dThinTasks, p->nthin, p->dInvF, 0, unit);
    if (H.factotype == PASTIX_AMD_FACT_LU) launch_solve_inv(p->stream, p->dL, p->f32, p->dSolve, p->dThinTasks, p->nthin, p->dInvB, 2, 0);
    else launch_solve_inv(p->stream, p->dL, p->f32, p->dSolve, p->dThinTasks, p->nthin, p->dInvB, 1, unit);
    HIPCHK(hipGetLastError());
    p->inv_gen = p->fact_gen;
  }
  return PASTIX_AMD_OK;
}

// one level of the forward (fwd) or backward sweep on nr right-hand sides (real arithmetic), on the plan's stream
void pai_solve_level(pastix_amd_plan_t* p, bool fwd, int l, double* dx, int nr) {
  const Plan& H = p->host;
  if (nr == 1 && p->nthin > 0 && p->lvl_thin[(size_t)l]) {
    // the whole run of thin levels that starts here goes in one launch; its other levels have nothing left to do
    const int64_t n = (fwd ? p->runF_n : p->runB_n)[(size_t)l], at = (fwd ? p->runF_at : p->runB_at)[(size_t)l];
    if (n == 0) return;
    const double* P = fwd ? p->dL : (H.factotype == PASTIX_AMD_FACT_LU ? p->dU : p->dL);
    int* cnt = p->dTicket + 2 * p->nthin;
    int* stuck = cnt + p->nthin;
    int* flag = stuck + 64 + (fwd ? 0 : (size_t)p->nthin * 64 * 8);
    launch_solve_thin(p->stream, fwd, P, p->f32, (fwd ? p->dThinF : p->dThinB) + at, n, p->dRidx, fwd ? p->dInvF : p->dInvB,
                      p->dTicket + (fwd ? 0 : p->nthin), p->dThinTgt, p->dThinExpect, cnt, flag, stuck, dx);
    return;
  }
  if (p->f32) {
    // (single-precision factors: float panels under double vectors, one right-hand side per pass)
    for (int k = 0; k < nr; k++)
      launch_solve_level_s(p->stream, fwd, H.factotype, (const float*)p->dL, (const float*)p->dU, p->dSolve + H.lvl_cblk_ptr[l],
                           H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l], p->lvl_nwide[(size_t)l],
                           (fwd ? p->dChunk + p->lvl_chunk_ptr[l] : p->dChunkB + p->lvl_chunkB_ptr[l]),
                           fwd ? p->lvl_chunk_ptr[l + 1] - p->lvl_chunk_ptr[l] : p->lvl_chunkB_ptr[l + 1] - p->lvl_chunkB_ptr[l],
                           p->dRidx, dx + (int64_t)k * H.ncol, p->lvl_maxw[l]);
    return;
  }
  if (fwd)
    launch_solve_level(p->stream, true, H.factotype, p->dL, p->dU, p->dSolve + H.lvl_cblk_ptr[l],
                       H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l], p->lvl_nwide[(size_t)l], p->dChunk + p->lvl_chunk_ptr[l],
                       p->lvl_chunk_ptr[l + 1] - p->lvl_chunk_ptr[l], p->dBlok, p->dRidx, dx, H.ncol, nr, p->maxw, p->lvl_maxw[l]);
  else
    launch_solve_level(p->stream, false, H.factotype, p->dL, p->dU, p->dSolve + H.lvl_cblk_ptr[l],
                       H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l], p->lvl_nwide[(size_t)l], p->dChunkB + p->lvl_chunkB_ptr[l],
                       p->lvl_chunkB_ptr[l + 1] - p->lvl_chunkB_ptr[l], p->dBlok, p->dRidx, dx, H.ncol, nr, p->maxw, p->lvl_maxw[l]);
}
void pai_solve_dscale(pastix_amd_plan_t* p, double* dx, int nr) {     // LDLt: x <- D^-1 x on the cblks factorized here
  const Plan& H = p->host;
  if (p->f32) {
    for (int k = 0; k < nr; k++) launch_solve_dscale_s(p->stream, (const float*)p->dL, p->dSolve, H.lvl_cblk_ptr[H.nlevels], dx + k * H.ncol);
    return;
  }
  for (int k = 0; k < nr; k++) launch_solve_dscale(p->stream, p->dL, p->dSolve, H.lvl_cblk_ptr[H.nlevels], dx + k * H.ncol);
}

static int solve_impl(pastix_amd_plan_t* p, void* x_, pastix_amd_int_t nrhs, bool x_on_device) {
  if (!p || !x_ || nrhs < 1) return PASTIX_AMD_ERR_BADPARAMETER;
  const Plan& H = p->host;
  if (p->distributed || H.opts.schur) return PASTIX_AMD_ERR_UNSUPPORTED;
  if (!p->factored) return PASTIX_AMD_ERR_BADPARAMETER;       // panels hold no factors (refill / upload since)
  HIPCHK(hipSetDevice(p->device));
  { const int rt = pai_solve_tables(p); if (rt) return rt; }
  const int mode = x_on_device ? 2 : 1;
  if (p->cplx) {
    // x is interleaved `double complex` (n x nrhs, column-major) like the reference's; planes on the device
    const size_t needz = (size_t)H.ncol * 4;
    if (p->nXws < needz) {
      (void)hipFree(p->dXws);
      p->dXws = nullptr; p->nXws = 0;
      HIPCHK(hipMalloc((void**)&p->dXws, needz * sizeof(double)));
      p->nXws = needz;
    }
    double *dzs = p->dXws, *dxr = p->dXws + 2 * (size_t)H.ncol, *dxi = p->dXws + 3 * (size_t)H.ncol;
    double* xz = (double*)x_;
    const bool scale = H.factotype == PASTIX_AMD_FACT_LDLT || H.factotype == PASTIX_AMD_FACT_LDLH;
    p->stats.solve_time = 0.0;
    for (int64_t j = 0; j < nrhs; j++) {
      double* dz = mode == 2 ? xz + 2 * j * H.ncol : dzs;          // interleaved vector on the device
      if (mode == 1) HIPCHK(hipMemcpyAsync(dz, xz + 2 * j * H.ncol, H.ncol * 2 * sizeof(double), hipMemcpyHostToDevice, p->stream));
      HIPCHK(hipEventRecord(p->ev0, p->stream));
      launch_split(p->stream, dz, dxr, dxi, H.ncol);
      for (int l = 0; l < H.nlevels; l++)
        launch_zsolve_level(p->stream, true, H.factotype, p->arenas(), p->dSolve + H.lvl_cblk_ptr[l],
                            H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l], p->dChunk + p->lvl_chunk_ptr[l],
                            p->lvl_chunk_ptr[l + 1] - p->lvl_chunk_ptr[l], p->dBlok, p->dRidx, dxr, dxi, p->maxw);
      if (scale) launch_zsolve_dscale(p->stream, p->arenas(), p->dSolve, H.lvl_cblk_ptr[H.nlevels], dxr, dxi);
      for (int l = H.nlevels - 1; l >= 0; l--)
        launch_zsolve_level(p->stream, false, H.factotype, p->arenas(), p->dSolve + H.lvl_cblk_ptr[l],
                            H.lvl_cblk_ptr[l + 1] - H.lvl_cblk_ptr[l], p->dChunkB + p->lvl_chunkB_ptr[l],
                            p->lvl_chunkB_ptr[l + 1] - p->lvl_chunkB_ptr[l], p->dBlok, p->dRidx, dxr, dxi, p->maxw);
      launch_merge(p->stream, dz, dxr, dxi, H.ncol);
      HIPCHK(hipEventRecord(p->ev1, p->stream));
      if (mode == 1) HIPCHK(hipMemcpyAsync(xz + 2 * j * H.ncol, dz, H.ncol * 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
      HIPCHK(hipStreamSynchronize(p->stream));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
      p->stats.solve_time += 1e-3 * ms;
    }
    HIPCHK(hipGetLastError());
    return PASTIX_AMD_OK;
  }
  // up to four right-hand sides per pass over the panels (the sweeps are HBM-bound on the panel bytes).  One stream:
  // the far rows of a panel on a second stream beside the next level's chain were built and measured slower (200^3:
  // 154 ms against 111 ms -- a cross-stream event per level costs more than the overlap returns on 753 levels).
  const int64_t NRB = std::min<int64_t>(nrhs, 4);
  const size_t need = (size_t)H.ncol * (size_t)NRB;
  if (mode == 1 && p->nXws < need) {
    (void)hipFree(p->dXws);
    p->dXws = nullptr; p->nXws = 0;
    HIPCHK(hipMalloc((void**)&p->dXws, need * sizeof(double)));
    p->nXws = need;
  }
  double* x = (double*)x_;
  p->stats.solve_time = 0.0;
  for (int64_t j = 0; j < nrhs;) {
    const int nr = nrhs - j >= 4 ? 4 : nrhs - j >= 2 ? 2 : 1;
    double* dx = mode == 2 ? x + j * H.ncol : p->dXws;
    if (mode == 1) HIPCHK(hipMemcpyAsync(dx, x + j * H.ncol, H.ncol * nr * sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipEventRecord(p->ev0, p->stream));
    if (nr == 1 && p->nthin > 0) HIPCHK(hipMemsetAsync(p->dTicket, 0, p->nTicket * sizeof(int), p->stream));
    for (int l = 0; l < H.nlevels; l++) pai_solve_level(p, true, l, dx, nr);
    if (H.factotype == PASTIX_AMD_FACT_LDLT) pai_solve_dscale(p, dx, nr);
    for (int l = H.nlevels - 1; l >= 0; l--) pai_solve_level(p, false, l, dx, nr);
    HIPCHK(hipEventRecord(p->ev1, p->stream));
    if (mode == 1) HIPCHK(hipMemcpyAsync(x + j * H.ncol, dx, H.ncol * nr * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    int stuck = 0;
    if (nr == 1 && p->nthin > 0)
      HIPCHK(hipMemcpyAsync(&stuck, p->dTicket + 3 * p->nthin, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (stuck) {
      fprintf(stderr, "pastix_amd: solve: a workgroup of a thin-level run waited for its predecessors beyond the poll limit\n");
      return PASTIX_AMD_ERR_DEVICE;
    }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->stats.solve_time += 1e-3 * ms;
    j += nr;
  }
  HIPCHK(hipGetLastError());
  return PASTIX_AMD_OK;
}

int pastix_amd_solve(pastix_amd_plan_t* p, void* x, pastix_amd_int_t nrhs) { return solve_impl(p, x, nrhs, false); }
// the same on a vector that already lives on the plan's device (device pointer, same layout): no host transfers --
// what the device-resident refinement (refine.hip) preconditions with
int pastix_amd_solve_device(pastix_amd_plan_t* p, void* dx, pastix_amd_int_t nrhs) { return solve_impl(p, dx, nrhs, true); }

// The one-shot entry points keep the plan of their last call: pastix() re-factorizes on one analysis (pastix.c:3439-3575 --
// same SolverMatrix, new values), and the plan is a pure function of the layout, so the second and later calls of a
// time-stepping or Newton loop skip the host analysis (seconds at 100^3), the device tables and the allocation of the
// arenas.  One plan is kept PER DEVICE (round 6: a thread per GPU no longer destroys the other's plan, and callers on
// different devices do not wait for one another); a hit is a match of the fingerprint of everything the plan depends on AND
// of the layout itself, entry by entry.  What is kept is the plan with its device arenas -- the panels' size, twice for
// LDLt / LU -- until pastix_amd_release_cached_plan() or the end of the process.
namespace {
struct OneShotEntry {
  std::mutex mu;                                            // one one-shot call at a time per device
  pastix_amd_plan_t* plan = nullptr;
  uint64_t key = 0;
  std::vector<pastix_amd_cblk_t> cblk;                      // the caller's layout the plan was made for
  std::vector<pastix_amd_blok_t> blok;
};
struct OneShotCache {
  std::mutex mu;                                            // guards the map only
  std::map<int, std::unique_ptr<OneShotEntry>> by_dev;
  ~OneShotCache() { for (auto& e : by_dev) (void)e.second.release(); /* (the process is ending: the runtime may already be gone -- nothing is released here) */ }
} g_one_shot;
uint64_t fnv(uint64_t h, const void* d, size_t n) {
  const unsigned char* q = (const unsigned char*)d;
  for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; }
  return h;
}
uint64_t one_shot_key(int factotype, int floattype, const pastix_amd_layout_t* L, const pastix_amd_options_t* opts) {
  uint64_t h = 1469598103934665603ull;
  const int64_t head[4] = {factotype, floattype, L->cblknbr, L->bloknbr};
  h = fnv(h, head, sizeof(head));
  if (opts) h = fnv(h, opts, sizeof(*opts));
  if (const char* e = getenv("PASTIX_AMD_DEV")) h = fnv(h, e, strlen(e));
  return h;
}
OneShotEntry* one_shot_entry(int device) {
  std::lock_guard<std::mutex> g(g_one_shot.mu);
  std::unique_ptr<OneShotEntry>& e = g_one_shot.by_dev[device];
  if (!e) e.reset(new OneShotEntry());
  return e.get();
}
}  // namespace
void pastix_amd_release_cached_plan(void) {
  std::vector<OneShotEntry*> all;
  {
    std::lock_guard<std::mutex> g(g_one_shot.mu);
    for (auto& e : g_one_shot.by_dev) all.push_back(e.second.get());
  }
  for (OneShotEntry* e : all) {
    std::lock_guard<std::mutex> g(e->mu);
    if (e->plan) pastix_amd_plan_destroy(e->plan);
    e->plan = nullptr;
    std::vector<pastix_amd_cblk_t>().swap(e->cblk);
    std::vector<pastix_amd_blok_t>().swap(e->blok);
  }
}

static int one_shot(int factotype, const pastix_amd_layout_t* layout, double* const* coeftab,
                    double* const* ucoeftab, double critere, const pastix_amd_options_t* opts,
                    pastix_amd_stats_t* stats, int floattype = PASTIX_AMD_REALDOUBLE) {
  if (!layout || !layout->cblktab || (layout->bloknbr > 0 && !layout->bloktab) || layout->cblknbr < 0) return PASTIX_AMD_ERR_BADPARAMETER;
  OneShotEntry& E = *one_shot_entry(opts ? opts->device : 0);
  std::lock_guard<std::mutex> g(E.mu);                      // (one one-shot call at a time per device: they share its plan)
  const double t0 = now_s();
  const uint64_t key = one_shot_key(factotype, floattype, layout, opts);
  pastix_amd_plan_t* plan = nullptr;
  int rc = 0;
  double plan_time = 0;
  const size_t ncb = (size_t)layout->cblknbr + 1, nbl = (size_t)layout->bloknbr;
  const bool hit = E.plan && E.key == key && E.cblk.size() == ncb && E.blok.size() == nbl &&
                   !memcmp(E.cblk.data(), layout->cblktab, ncb * sizeof(pastix_amd_cblk_t)) &&
                   (nbl == 0 || !memcmp(E.blok.data(), layout->bloktab, nbl * sizeof(pastix_amd_blok_t)));
  if (hit) {
    plan = E.plan;
  } else {
    if (E.plan) { pastix_amd_plan_destroy(E.plan); E.plan = nullptr; }
    rc = pastix_amd_plan_create(layout, factotype, floattype, opts, &plan);
    if (rc) return rc;
    plan_time = now_s() - t0;
    try {
      E.cblk.assign(layout->cblktab, layout->cblktab + ncb);
      E.blok.assign(layout->bloktab, layout->bloktab + nbl);
    } catch (const std::bad_alloc&) { pastix_amd_plan_destroy(plan); return PASTIX_AMD_ERR_ALLOC; }
    E.plan = plan;
    E.key = key;
  }
  rc = pastix_amd_upload_tabs(plan, (void* const*)coeftab, (void* const*)ucoeftab);
  int rcf = 0;
  plan->caller_restores = true;                             // (a stopped run is redone below from the caller's buffers)
  // Finished panels go home while the run factorizes the rest (pastix_amd_factorize, staged_tabs_io part 1;
  // PASTIX_AMD_DEV=no_early_out: all of them afterwards).  They overwrite the caller's input, which is what a stopped run is
  // redone from: with early copies the input is kept on the DEVICE instead -- a copy of the arenas taken now (100^3: 8.8 GB,
  // 5 ms), freed when the call returns; no room for it, no early copies.
  void* snap[2] = {nullptr, nullptr};
  const size_t arena_bytes = (size_t)plan->host.coefnbr * plan->esz;
  plan->early_tab = plan->early_utab = nullptr;
  // (from 2e12 flop: every copy that ends beside a running kernel costs it a write-back of the L2s -- 60^3, a run of 10 ms:
  // 17 -> 20 ms with the early copies, 100^3: +1 ms of 121 for 80 ms less in the call)
  if (!rc && !dev_opt("no_early_out") && !plan->cplx && !plan->distributed && plan->run_ready && plan->host.run_L0 > 0 &&
      plan->host.fact_flops >= (dev_opt("early_out_min") ? atof(dev_opt("early_out_min")) : 2e12)) {
    size_t fr = 0, tot = 0;
    const int na = (plan->dU && ucoeftab) ? 2 : 1;
    bool ok = hipMemGetInfo(&fr, &tot) == hipSuccess && fr > (size_t)na * arena_bytes + ((size_t)2 << 30);
    for (int a = 0; ok && a < na; a++) ok = hipMalloc(&snap[a], arena_bytes) == hipSuccess;
    if (ok) {
      for (int a = 0; ok && a < na; a++)
        ok = hipMemcpyAsync(snap[a], plan->at(a ? plan->dU : plan->dL, 0), arena_bytes, hipMemcpyDeviceToDevice, plan->stream) == hipSuccess;
    }
    if (ok) {
      plan->early_tab = (void* const*)coeftab;
      plan->early_utab = (void* const*)ucoeftab;
    } else {
      (void)hipGetLastError();
      for (void*& x : snap) { if (x) (void)hipFree(x); x = nullptr; }
    }
  }
  if (!rc) rcf = pastix_amd_factorize(plan, critere, nullptr);
  const bool went_early = plan->early_done;
  if (rc || rcf) plan->early_done = false;
  if (!rc && rcf == PASTIX_AMD_ERR_DEVICE && plan->run_stuck) {      // (see pastix_amd_factorize)
    if (went_early) {
      fprintf(stderr, "pastix_amd: restoring the panels from the device copy and factorizing on the level-by-level schedule\n");
      for (int a = 0; a < 2 && !rc; a++)
        if (snap[a] && hipMemcpyAsync(plan->at(a ? plan->dU : plan->dL, 0), snap[a], arena_bytes, hipMemcpyDeviceToDevice, plan->stream) != hipSuccess)
          rc = PASTIX_AMD_ERR_DEVICE;
      plan->factored = false;
    } else {
      fprintf(stderr, "pastix_amd: uploading the panels again and factorizing on the level-by-level schedule\n");
      rc = pastix_amd_upload_tabs(plan, (void* const*)coeftab, (void* const*)ucoeftab);
    }
    plan->run_off_once = true;
    if (!rc) rcf = pastix_amd_factorize(plan, critere, nullptr);
    plan->run_off_once = false;
  }
  if (!rc && (rcf == 0 || rcf == PASTIX_AMD_ERR_NUMERIC))
    rc = pastix_amd_download_tabs(plan, (void* const*)coeftab, (void* const*)ucoeftab);
  plan->early_tab = plan->early_utab = nullptr;
  plan->early_done = false;
  for (void* x : snap) if (x) (void)hipFree(x);
  plan->stats.plan_time = plan_time;
  plan->stats.total_time = now_s() - t0;
  if (stats) pastix_amd_plan_stats(plan, stats);
  static const bool verbose = getenv("PASTIX_AMD_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "pastix_amd: one-shot call: plan %.3f s%s, host -> device %.3f s, factorization %.3f s, device -> host %.3f s, in all %.3f s\n",
            plan_time, plan_time == 0 ? " (cached)" : "", plan->stats.h2d_time, plan->stats.fact_time, plan->stats.d2h_time,
            plan->stats.total_time);
  if (rc || (rcf && rcf != PASTIX_AMD_ERR_NUMERIC)) {        // (a plan that failed is not kept)
    pastix_amd_plan_destroy(plan);
    E.plan = nullptr;
  }
  return rc ? rc : rcf;
}

int pastix_amd_d_po_sopalin(const pastix_amd_layout_t* layout, double* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LLT, layout, coeftab, nullptr, critere, opts, stats);
}
int pastix_amd_d_sy_sopalin(const pastix_amd_layout_t* layout, double* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LDLT, layout, coeftab, nullptr, critere, opts, stats);
}
int pastix_amd_z_sy_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LDLT, layout, (double* const*)coeftab, nullptr, critere, opts, stats,
                  PASTIX_AMD_COMPLEXDOUBLE);
}
int pastix_amd_z_he_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LDLH, layout, (double* const*)coeftab, nullptr, critere, opts, stats,
                  PASTIX_AMD_COMPLEXDOUBLE);
}
int pastix_amd_z_ge_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, void* const* ucoeftab,
                            double critere, const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LU, layout, (double* const*)coeftab, (double* const*)ucoeftab, critere, opts, stats,
                  PASTIX_AMD_COMPLEXDOUBLE);
}
int pastix_amd_d_ge_sopalin(const pastix_amd_layout_t* layout, double* const* coeftab, double* const* ucoeftab,
                            double critere, const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot(PASTIX_AMD_FACT_LU, layout, coeftab, ucoeftab, critere, opts, stats);
}


// Single-precision drop-ins (S_ / C_ {po,sy,he,ge}_sopalin_thread): the caller's float panels are widened on the
// host, factorized by the fp64 engine and rounded back -- results are at least as accurate as the reference's
// single-precision run (parity tolerance 1e-4).  An fp32-MFMA path is not built.
static int one_shot_single(int factotype, const pastix_amd_layout_t* layout, void* const* coeftab,
                           void* const* ucoeftab, double critere, const pastix_amd_options_t* opts,
                           pastix_amd_stats_t* stats, bool cplx) {
  if (!layout || !coeftab || !layout->cblktab) return PASTIX_AMD_ERR_BADPARAMETER;
  // real single precision: the fp32 engine on the caller's float panels, nothing is widened (kernels_f32.hip);
  // complex single precision is still widened on the host and factorized by the fp64 engine
  if (!cplx)
    return one_shot(factotype, layout, (double* const*)coeftab, (double* const*)ucoeftab, critere, opts, stats,
                    PASTIX_AMD_REALSINGLE);
  const int64_t nc = layout->cblknbr;
  const int es = cplx ? 2 : 1;
  std::vector<std::vector<double>> L((size_t)nc), U;
  std::vector<double*> lp((size_t)nc, nullptr), up;
  const bool lu = factotype == PASTIX_AMD_FACT_LU;
  if (lu) {
    if (!ucoeftab) return PASTIX_AMD_ERR_BADPARAMETER;
    U.resize((size_t)nc);
    up.assign((size_t)nc, nullptr);
  }
  auto cnt = [&](int64_t k) {
    return (int64_t)(layout->cblktab[k].lcolnum - layout->cblktab[k].fcolnum + 1) * layout->cblktab[k].stride * es;
  };
  try {
    for (int64_t k = 0; k < nc; k++) {
      if (!coeftab[k] || (lu && !ucoeftab[k])) return PASTIX_AMD_ERR_BADPARAMETER;
      const int64_t n = cnt(k);
      L[k].resize((size_t)n);
      const float* src = (const float*)coeftab[k];
      for (int64_t i = 0; i < n; i++) L[k][i] = (double)src[i];
      lp[k] = L[k].data();
      if (lu) {
        U[k].resize((size_t)n);
        const float* su = (const float*)ucoeftab[k];
        for (int64_t i = 0; i < n; i++) U[k][i] = (double)su[i];
        up[k] = U[k].data();
      }
    }
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  const int rc = one_shot(factotype, layout, lp.data(), lu ? up.data() : nullptr, critere, opts, stats,
                          cplx ? PASTIX_AMD_COMPLEXDOUBLE : PASTIX_AMD_REALDOUBLE);
  if (rc == 0 || rc == PASTIX_AMD_ERR_NUMERIC) {
    for (int64_t k = 0; k < nc; k++) {
      const int64_t n = cnt(k);
      float* dst = (float*)coeftab[k];
      for (int64_t i = 0; i < n; i++) dst[i] = (float)L[k][i];
      if (lu) {
        float* du = (float*)ucoeftab[k];
        for (int64_t i = 0; i < n; i++) du[i] = (float)U[k][i];
      }
    }
  }
  return rc;
}

int pastix_amd_s_po_sopalin(const pastix_amd_layout_t* layout, float* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LLT, layout, (void* const*)coeftab, nullptr, critere, opts, stats, false);
}
int pastix_amd_s_sy_sopalin(const pastix_amd_layout_t* layout, float* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LDLT, layout, (void* const*)coeftab, nullptr, critere, opts, stats, false);
}
int pastix_amd_s_ge_sopalin(const pastix_amd_layout_t* layout, float* const* coeftab, float* const* ucoeftab,
                            double critere, const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LU, layout, (void* const*)coeftab, (void* const*)ucoeftab, critere, opts, stats,
                         false);
}
int pastix_amd_c_sy_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LDLT, layout, coeftab, nullptr, critere, opts, stats, true);
}
int pastix_amd_c_he_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, double critere,
                            const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LDLH, layout, coeftab, nullptr, critere, opts, stats, true);
}
int pastix_amd_c_ge_sopalin(const pastix_amd_layout_t* layout, void* const* coeftab, void* const* ucoeftab,
                            double critere, const pastix_amd_options_t* opts, pastix_amd_stats_t* stats) {
  return one_shot_single(PASTIX_AMD_FACT_LU, layout, coeftab, ucoeftab, critere, opts, stats, true);
}

}  // extern "C"
