// run_debug.cpp -- developer aid: what a run launch that gave up (bounded wait, run_sync.h) looked like when it stopped.
// Always: the rings' heads and tails and the tasks whose inputs never all arrived.  With PASTIX_AMD_DEV=run_debug the plan
// keeps the run's dependency tables on the host and every task stamps its start and end (RunCtl::prof): the report then
// REPLAYS the counter protocol -- every task that finished must have counted down all its consumers, every task whose
// counter reached zero must be in a ring, nothing may be in a ring twice -- and lists the tasks that started and did not
// finish.  Not part of the factorization: pastix_amd_factorize_end calls it on the error path only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "engine.h"

using namespace pastix_amd;

void run_debug_report(pastix_amd_plan_t* p) {
  std::vector<int32_t> st(p->nRunState);
  if (hipMemcpy(st.data(), p->dRunState, p->nRunState * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return;
  const size_t nr = (size_t)p->runctl.nticket, nd = (size_t)p->runctl.nd;
  const bool onek = p->runctl.onek != 0;
  const int32_t* ctl = st.data() + (p->runctl.ctl - p->dRunState);
  size_t wt = 0, wd = 0;
  for (size_t i = 0; i < nr; i++) wt += st[i] > 0;
  for (size_t i = 0; i < nd; i++) wd += st[nr + i] > 0;
  fprintf(stderr, "pastix_amd:   the first to give up waited for slot %d of a ring of %d for %d ms (limit %d ms, %d polls)\n",
          ctl[RUN_STUCK + 2], ctl[RUN_STUCK + 3], ctl[RUN_STUCK + 4], ctl[RUN_STUCK + 5], ctl[RUN_STUCK + 6]);
  if (onek)
    fprintf(stderr, "pastix_amd:   one ring: %zu tickets + %zu diagonal tasks, popped %d, pushed %d; never ready: %zu tickets, %zu diagonal tasks\n",
            nr, nd, ctl[RUN_HEAD], ctl[RUN_TAIL], wt, wd);
  else
    fprintf(stderr, "pastix_amd:   tickets %zu: popped %d, pushed %d, %zu never ready; diagonal tasks %zu: popped %d, pushed %d, %zu never ready;"
            " resident workers %d of %d\n", nr, ctl[RUN_HEAD], ctl[RUN_TAIL], wt, nd, ctl[RUN_HEAD + 64], ctl[RUN_TAIL + 64], wd,
            p->hResident ? *(volatile int*)p->hResident : 0, (int)p->host.run_gd);
  int shown = 0;
  for (size_t i = 0; i < nd && shown < 4; i++) if (st[nr + i] > 0) { fprintf(stderr, "pastix_amd:   diagonal task %zu waits for %d input(s)\n", i, st[nr + i]); shown++; }
  shown = 0;
  for (size_t i = 0; i < nr && shown < 6; i++) if (st[i] > 0) { fprintf(stderr, "pastix_amd:   ticket %zu waits for %d input(s)\n", i, st[i]); shown++; }
  if (p->dbg_info.empty()) return;

  // replay: what the tasks that ran should have counted down (task ids: tickets [0, nr), diagonal tasks nr + d)
  const int32_t* q = st.data() + (p->runctl.q - p->dRunState);
  const int32_t* qd = st.data() + (p->runctl.qd - p->dRunState);
  std::vector<int32_t> exp(nr + nd, 0);
  std::vector<uint8_t> ran(nr + nd, 0);
  std::vector<long long> stamp;
  if (p->dRunProf) { stamp.resize(p->nRunProf); (void)hipMemcpy(stamp.data(), p->dRunProf, p->nRunProf * sizeof(long long), hipMemcpyDeviceToHost); }
  auto started = [&](size_t task) { return stamp.empty() || stamp[4 * task + 1] != 0; };
  auto done = [&](size_t task) { return stamp.empty() || stamp[4 * task + 2] != 0; };
  std::vector<size_t> inring;
  for (int i = 0; i < ctl[RUN_TAIL]; i++) {
    const int32_t t = q[(size_t)i * RUN_SLOT];
    if (t < 0 || (size_t)t >= (onek ? nr + nd : nr)) { fprintf(stderr, "pastix_amd:   ring slot %d holds %d\n", i, t); continue; }
    inring.push_back((size_t)t);
  }
  if (!onek)
    for (int i = 0; i < ctl[RUN_TAIL + 64]; i++) {
      const int32_t d = qd[(size_t)i * RUN_SLOT];
      if (d < 0 || (size_t)d >= nd) { fprintf(stderr, "pastix_amd:   diagonal ring slot %d holds %d\n", i, d); continue; }
      inring.push_back(nr + (size_t)d);
    }
  int dup = 0, pushed_not_started = 0, started_not_done = 0;
  for (size_t c : inring) {
    const char* what = c < nr ? "ticket" : "diagonal task";
    const size_t id = c < nr ? c : c - nr;
    if (!started(c)) { if (pushed_not_started++ < 6) fprintf(stderr, "pastix_amd:   %s %zu was pushed and never started\n", what, id); continue; }
    if (!done(c)) { if (started_not_done++ < 6) fprintf(stderr, "pastix_amd:   %s %zu started and did not finish\n", what, id); continue; }
    if (ran[c]++) dup++;
    if (c >= nr) { for (int z = 0; z < p->dbg_d[id].tn; z++) exp[(size_t)p->dbg_d[id].t0 + (size_t)z]++; continue; }
    const RunInfo& ri = p->dbg_info[c];
    if (ri.kind & 4) { for (int z = 0; z < ri.cn; z++) exp[(size_t)p->dbg_cons[(size_t)ri.cptr + (size_t)z]]++; }
    else if (ri.succ >= 0) { for (int z = 0; z < ri.cn; z++) exp[(size_t)ri.succ + (size_t)z]++; }
    else if (ri.succ <= -2) exp[nr + (size_t)(-2 - ri.succ)]++;
  }
  long long lost = 0, extra = 0;
  shown = 0;
  for (size_t c = 0; c < nr + nd; c++) {
    const int applied = p->dbg_dep[c] - st[c];
    if (applied == exp[c]) continue;
    (applied < exp[c] ? lost : extra) += std::abs(exp[c] - applied);
    if (shown++ < 12)
      fprintf(stderr, "pastix_amd:   %s %zu: %d inputs, %d counted down, %d of its producers ran (it %s)\n", c < nr ? "ticket" : "diagonal task",
              c < nr ? c : c - nr, p->dbg_dep[c], applied, exp[c], ran[c] ? "ran" : "did not run");
  }
  if (!stamp.empty()) {        // the longest tasks (100 MHz stamps: drawn / started / done / where)
    std::vector<std::pair<long long, size_t>> dur;
    long long tmax = 0;
    for (size_t c = 0; c < nr + nd; c++) if (stamp[4 * c + 2]) { dur.emplace_back(stamp[4 * c + 2] - stamp[4 * c + 1], c); tmax = std::max(tmax, stamp[4 * c + 2]); }
    std::sort(dur.begin(), dur.end());
    size_t nlong = 0;
    for (auto& d2 : dur) nlong += d2.first > 1000000;
    fprintf(stderr, "pastix_amd:   %zu tasks took longer than 10 ms\n", nlong);
    for (size_t i = dur.size() > 8 ? dur.size() - 8 : 0; i < dur.size(); i++) {
      const size_t c = dur[i].second;
      fprintf(stderr, "pastix_amd:   %s %zu (kind %d) ran %.3f ms, finished %.3f ms before the last one, on hw %llx\n", c < nr ? "ticket" : "diagonal task",
              c < nr ? c : c - nr, c < nr ? (int)p->dbg_info[c].kind : -1, dur[i].first * 1e-5, (tmax - stamp[4 * c + 2]) * 1e-5,
              (unsigned long long)stamp[4 * c + 3]);
    }
  }
  fprintf(stderr, "pastix_amd:   replay: %lld count-downs missing, %lld too many, %d tasks twice in a ring, %d pushed and never started, %d started and not finished\n",
          lost, extra, dup, pushed_not_started, started_not_done);
}
