// symbolic.cpp -- host-side producer of a SolverMatrix-compatible cblk/blok layout.
//
// In PaStiX this is the job of order/ + kass/ + blend/ (kass.c:93-176 supernodes + amalgamation,
// splitpart.c cblk splitting, solverMatrixGen.c:1053-1069 coefind/stride), which "stay as-is" for
// a real drop-in.  The GPU box has no PaStiX, Scotch or METIS, so the benchmarks need a producer of
// the same data model.  It is a fresh implementation of the standard pipeline, not a restatement:
//   ordering (geometric nested dissection for grids, separators numbered hierarchically so that the
//   rows a sub-box touches in an ancestor separator are contiguous)  ->  elimination tree (Liu)
//   -> postorder -> column counts (Gilbert-Ng-Peyton) -> supernodes -> fill-bounded amalgamation
//   (same criterion family as kass: cheapest merges first until a fill budget is spent)
//   -> supernodal symbolic factorization on interval lists -> cblk splitting at max blocksize
//   -> bloks cut at facing-cblk boundaries, coefind/stride.
// The layout it emits satisfies the invariants plan.cpp::check_layout enforces.
#include "plan.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <numeric>
#include <atomic>
#include <queue>
#include <thread>
#include <vector>

#include "../../include/pastix_amd.h"
#include "../../include/pastix_amd_symbolic.h"

namespace {

typedef int32_t idx;

// ---- geometric nested dissection ----------------------------------------------------------------
struct GridND {
  int NX, NY, NZ, leaf;
  std::vector<int64_t>& invp;   // new -> old
  std::vector<int64_t>& perm;   // old -> new (filled on the fly: needed for separator keys)
  int64_t next = 0;
  int64_t id(int x, int y, int z) const { return x + (int64_t)NX * (y + (int64_t)NY * z); }
  void put(int64_t old) { perm[old] = next; invp[next++] = old; }
  void rec(int x0, int x1, int y0, int y1, int z0, int z1) {
    const int dx = x1 - x0, dy = y1 - y0, dz = z1 - z0;
    const int64_t cnt = (int64_t)dx * dy * dz;
    if (cnt <= 0) return;
    if (cnt <= leaf || (dx <= 2 && dy <= 2 && dz <= 2)) {
      for (int z = z0; z < z1; z++) for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) put(id(x, y, z));
      return;
    }
    std::vector<std::pair<int64_t, int64_t>> sep;   // (key, old id)
    if (dx >= dy && dx >= dz) {
      const int m = x0 + dx / 2;
      rec(x0, m, y0, y1, z0, z1);
      rec(m + 1, x1, y0, y1, z0, z1);
      for (int z = z0; z < z1; z++) for (int y = y0; y < y1; y++)
        sep.emplace_back(m > x0 ? perm[id(m - 1, y, z)] : id(m, y, z), id(m, y, z));
    } else if (dy >= dz) {
      const int m = y0 + dy / 2;
      rec(x0, x1, y0, m, z0, z1);
      rec(x0, x1, m + 1, y1, z0, z1);
      for (int z = z0; z < z1; z++) for (int x = x0; x < x1; x++)
        sep.emplace_back(m > y0 ? perm[id(x, m - 1, z)] : id(x, m, z), id(x, m, z));
    } else {
      const int m = z0 + dz / 2;
      rec(x0, x1, y0, y1, z0, m);
      rec(x0, x1, y0, y1, m + 1, z1);
      for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++)
        sep.emplace_back(m > z0 ? perm[id(x, y, m - 1)] : id(x, y, m), id(x, y, m));
    }
    // number the separator in the order its neighbours on the low side were numbered: every
    // descendant box of the low side then touches a contiguous range of the separator
    std::sort(sep.begin(), sep.end());
    for (auto& s : sep) put(s.second);
  }
};

struct Interval { idx a, b; };   // rows [a, b] inclusive

}  // namespace

struct pastix_amd_symbol_s {
  int64_t n = 0;
  std::vector<int64_t> perm, invp;
  std::vector<pastix_amd_cblk_t> cblk;
  std::vector<pastix_amd_blok_t> blok;
  int64_t nnzl = 0;          // sum over cblks of stride*width - w(w-1)/2 (lower triangle + panel)
  int64_t nsuper_fund = 0, nsuper_amalg = 0;
};

extern "C" {

int pastix_amd_order_grid(pastix_amd_int_t nx, pastix_amd_int_t ny, pastix_amd_int_t nz, int leaf,
                          pastix_amd_int_t* perm, pastix_amd_int_t* invp) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || !perm || !invp) return PASTIX_AMD_ERR_BADPARAMETER;
  const int64_t n = nx * ny * nz;
  std::vector<int64_t> p((size_t)n, -1), ip((size_t)n, -1);
  GridND nd{(int)nx, (int)ny, (int)nz, leaf > 0 ? leaf : 8, ip, p};
  nd.rec(0, (int)nx, 0, (int)ny, 0, (int)nz);
  if (nd.next != n) return PASTIX_AMD_ERR_BADPARAMETER;
  std::memcpy(perm, p.data(), n * sizeof(int64_t));
  std::memcpy(invp, ip.data(), n * sizeof(int64_t));
  return PASTIX_AMD_OK;
}

// George's automatic nested dissection on a general graph (see the header): an explicit work list instead of recursion;
// `parts` holds vertex lists, numbering is assigned from the END of the order backwards so that every separator comes
// after both of its parts.
int pastix_amd_order_graph(pastix_amd_int_t n, const pastix_amd_int_t* colptr, const pastix_amd_int_t* rows, int leaf,
                           pastix_amd_int_t* perm, pastix_amd_int_t* invp) {
  if (n <= 0 || !colptr || !rows || !perm || !invp || n > 0x7ffffff0LL) return PASTIX_AMD_ERR_BADPARAMETER;
  if (leaf <= 0) leaf = 64;
  try {
    // symmetric adjacency, 0-based, no diagonal
    std::vector<int64_t> xadj((size_t)n + 1, 0);
    const int64_t nnz = colptr[n] - 1;
    for (int64_t j = 0; j < n; j++)
      for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
        const int64_t i = rows[q] - 1;
        if (i < 0 || i >= n) return PASTIX_AMD_ERR_BADPARAMETER;
        if (i != j) { xadj[i + 1]++; xadj[j + 1]++; }
      }
    (void)nnz;
    for (int64_t j = 0; j < n; j++) xadj[j + 1] += xadj[j];
    std::vector<idx> adj((size_t)xadj[n]);
    {
      std::vector<int64_t> pos(xadj.begin(), xadj.end() - 1);
      for (int64_t j = 0; j < n; j++)
        for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
          const int64_t i = rows[q] - 1;
          if (i != j) { adj[pos[i]++] = (idx)j; adj[pos[j]++] = (idx)i; }
        }
    }
    std::vector<int32_t> part((size_t)n, 0);       // id of the part a vertex currently belongs to (-1: numbered)
    std::vector<idx> lvl((size_t)n, -1), queue;
    std::vector<std::vector<idx>> parts;
    parts.emplace_back((size_t)n);
    std::iota(parts[0].begin(), parts[0].end(), 0);
    std::vector<size_t> work{0};
    int64_t next = n;                              // numbers are handed out downwards
    std::vector<int64_t> ip((size_t)n, -1);
    auto bfs = [&](idx root, int32_t pid) {        // level structure of root's component inside part pid -> queue, lvl
      queue.clear();
      queue.push_back(root);
      lvl[root] = 0;
      for (size_t h = 0; h < queue.size(); h++) {
        const idx v = queue[h];
        for (int64_t q = xadj[v]; q < xadj[v + 1]; q++) {
          const idx u = adj[q];
          if (part[u] == pid && lvl[u] < 0) { lvl[u] = lvl[v] + 1; queue.push_back(u); }
        }
      }
    };
    while (!work.empty()) {
      const size_t wi = work.back();
      work.pop_back();
      std::vector<idx> verts;
      verts.swap(parts[wi]);
      const int32_t pid = (int32_t)wi;
      if (verts.empty()) continue;
      // one connected component at a time
      idx root = verts[0];
      bfs(root, pid);
      if (queue.size() < verts.size()) {           // disconnected: split off this component, requeue both
        std::vector<idx> comp(queue), rest;
        for (idx v : comp) lvl[v] = -1;
        const int32_t cid = (int32_t)parts.size();
        for (idx v : comp) part[v] = cid;
        for (idx v : verts) if (part[v] == pid) rest.push_back(v);
        parts.push_back(std::move(comp));
        parts[wi] = std::move(rest);
        work.push_back(wi);
        work.push_back((size_t)cid);
        continue;
      }
      if ((int64_t)verts.size() <= leaf) {
        // leaf: reverse Cuthill-McKee = the BFS order from a pseudo-peripheral vertex, reversed; numbers go downwards,
        // so handing them out in BFS order gives exactly the reversed order
        for (idx v : queue) lvl[v] = -1;
        idx far = queue.back();
        bfs(far, pid);
        for (idx v : queue) { ip[--next] = v; part[v] = -1; lvl[v] = -1; }
        continue;
      }
      // pseudo-peripheral vertex: repeat BFS from the farthest vertex while the eccentricity grows
      idx ecc = lvl[queue.back()];
      for (int it = 0; it < 4; it++) {
        const idx far = queue.back();
        for (idx v : queue) lvl[v] = -1;
        bfs(far, pid);
        const idx e2 = lvl[queue.back()];
        if (e2 <= ecc) break;
        ecc = e2;
      }
      const idx depth = lvl[queue.back()];
      if (depth < 2) {                             // (nearly) a clique: no separator to find
        for (idx v : queue) { ip[--next] = v; part[v] = -1; lvl[v] = -1; }
        continue;
      }
      // separator = the level that halves the vertex count (by cumulative level sizes)
      std::vector<int64_t> cnt((size_t)depth + 1, 0);
      for (idx v : queue) cnt[(size_t)lvl[v]]++;
      int64_t acc = 0;
      idx mid = 1;
      for (idx l = 0; l <= depth; l++) {
        acc += cnt[(size_t)l];
        if (2 * acc >= (int64_t)queue.size()) { mid = l; break; }
      }
      mid = std::max<idx>(1, std::min<idx>(mid, depth - 1));
      const int32_t aid = (int32_t)parts.size(), bid = aid + 1;
      std::vector<idx> A, B;
      for (idx v : queue) {
        if (lvl[v] == mid) { ip[--next] = v; part[v] = -1; }      // separator: numbered now (after both halves)
        else if (lvl[v] < mid) { A.push_back(v); part[v] = aid; }
        else { B.push_back(v); part[v] = bid; }
        lvl[v] = -1;
      }
      parts.push_back(std::move(A));
      parts.push_back(std::move(B));
      work.push_back((size_t)aid);
      work.push_back((size_t)bid);
    }
    if (next != 0) return PASTIX_AMD_ERR_BADPARAMETER;
    for (int64_t k = 0; k < n; k++) { invp[k] = ip[(size_t)k]; perm[ip[(size_t)k]] = k; }
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}

void pastix_amd_symbol_destroy(pastix_amd_symbol_t* s) { delete s; }

int pastix_amd_symbol_layout(const pastix_amd_symbol_t* s, pastix_amd_layout_t* out) {
  if (!s || !out) return PASTIX_AMD_ERR_BADPARAMETER;
  out->cblknbr = (int64_t)s->cblk.size() - 1;
  out->bloknbr = (int64_t)s->blok.size();
  out->cblktab = s->cblk.data();
  out->bloktab = s->blok.data();
  return PASTIX_AMD_OK;
}

int pastix_amd_symbol_perm(const pastix_amd_symbol_t* s, const pastix_amd_int_t** perm,
                           const pastix_amd_int_t** invp) {
  if (!s) return PASTIX_AMD_ERR_BADPARAMETER;
  if (perm) *perm = s->perm.data();
  if (invp) *invp = s->invp.data();
  return PASTIX_AMD_OK;
}

int pastix_amd_symbol_info(const pastix_amd_symbol_t* s, pastix_amd_int_t* info) {
  if (!s || !info) return PASTIX_AMD_ERR_BADPARAMETER;
  info[0] = s->n; info[1] = (int64_t)s->cblk.size() - 1; info[2] = (int64_t)s->blok.size();
  info[3] = s->nnzl; info[4] = s->nsuper_fund; info[5] = s->nsuper_amalg;
  return PASTIX_AMD_OK;
}

int pastix_amd_symbolic(pastix_amd_int_t n, const pastix_amd_int_t* colptr, const pastix_amd_int_t* rows,
                        const pastix_amd_int_t* perm_in, const pastix_amd_symbolic_options_t* opts_in,
                        pastix_amd_symbol_t** out) {
  if (!out || n <= 0 || !colptr || !rows || n > 0x7ffffff0LL) return PASTIX_AMD_ERR_BADPARAMETER;
  *out = nullptr;
  pastix_amd_symbolic_options_t o{};
  if (opts_in) o = *opts_in;
  if (o.max_blocksize <= 0) o.max_blocksize = 128;
  if (o.max_blocksize > 256 && !o.blend_split) o.max_blocksize = 256;   // (blend's rule may leave wider cblks: the
                                                                        // engine factorizes those as column groups)
  if (o.min_blocksize <= 0) o.min_blocksize = std::max(1, o.max_blocksize / 2);
  if (o.candidate_procs <= 0) o.candidate_procs = 1;
  if (o.amalgamation_pct < 0) o.amalgamation_pct = 0;
  const double ratio = (o.amalgamation_pct == 0 && !opts_in) ? 0.05 : o.amalgamation_pct * 0.01;
  pastix_amd_symbol_t* S = new (std::nothrow) pastix_amd_symbol_t();
  if (!S) return PASTIX_AMD_ERR_ALLOC;
  try {
    S->n = n;
    std::vector<idx> perm((size_t)n);
    if (perm_in) { for (int64_t i = 0; i < n; i++) perm[i] = (idx)perm_in[i]; }
    else std::iota(perm.begin(), perm.end(), 0);
    {  // validate permutation
      std::vector<char> seen((size_t)n, 0);
      for (int64_t i = 0; i < n; i++) {
        if (perm[i] < 0 || perm[i] >= n || seen[perm[i]]) { delete S; return PASTIX_AMD_ERR_BADPARAMETER; }
        seen[perm[i]] = 1;
      }
    }
    const int64_t nnz = colptr[n] - 1;
    const bool ptime = pastix_amd::dev_opt("plan_timing") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tph = tnow();
    auto phase = [&](const char* name) { if (ptime) { const double t = tnow(); fprintf(stderr, "[symbolic] %-28s %.2f s\n", name, t - tph); tph = t; } };

    // symmetric adjacency (no diagonal) in the numbering given by `lab` (old -> label)
    std::vector<int64_t> xadj;
    std::vector<idx> adj;
    auto build_adj = [&](const std::vector<idx>& lab) {
      xadj.assign((size_t)n + 1, 0);
      for (int64_t j = 0; j < n; j++)
        for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
          int64_t i = rows[q] - 1;
          if (i == j) continue;
          xadj[lab[i] + 1]++; xadj[lab[j] + 1]++;
        }
      for (int64_t j = 0; j < n; j++) xadj[j + 1] += xadj[j];
      adj.resize((size_t)xadj[n]);
      std::vector<int64_t> pos(xadj.begin(), xadj.end() - 1);
      for (int64_t j = 0; j < n; j++)
        for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
          int64_t i = rows[q] - 1;
          if (i == j) continue;
          adj[pos[lab[i]]++] = lab[j];
          adj[pos[lab[j]]++] = lab[i];
        }
      // duplicates (full-pattern input lists both (i,j) and (j,i)) are harmless below
    };
    for (int64_t q = 0; q < nnz; q++)
      if (rows[q] < 1 || rows[q] > n) { delete S; return PASTIX_AMD_ERR_BADPARAMETER; }

    // ---- elimination tree (Liu, path compression) in the given order ---------------------------
    std::vector<idx> parent((size_t)n, -1);
    auto etree = [&]() {
      std::vector<idx> anc((size_t)n, -1);
      std::fill(parent.begin(), parent.end(), -1);
      for (idx j = 0; j < n; j++)
        for (int64_t q = xadj[j]; q < xadj[j + 1]; q++) {
          idx i = adj[q];
          while (i != -1 && i < j) {
            idx nx = anc[i];
            anc[i] = j;
            if (nx == -1) parent[i] = j;
            i = nx;
          }
        }
    };
    build_adj(perm);
    etree();
    phase("adjacency + etree");
    // ---- postorder; relabel ----------------------------------------------------------------------
    {
      std::vector<idx> head((size_t)n, -1), next((size_t)n, -1), post((size_t)n), stack;
      for (idx j = (idx)n - 1; j >= 0; j--)
        if (parent[j] != -1) { next[j] = head[parent[j]]; head[parent[j]] = j; }
      idx k = 0;
      for (idx r = 0; r < n; r++) {
        if (parent[r] != -1) continue;
        stack.push_back(r);
        while (!stack.empty()) {
          idx p = stack.back(), c = head[p];
          if (c == -1) { post[p] = k++; stack.pop_back(); }
          else { head[p] = next[c]; stack.push_back(c); }
        }
      }
      bool ident = true;
      for (idx j = 0; j < n; j++) if (post[j] != j) { ident = false; break; }
      if (!ident) {
        for (int64_t i = 0; i < n; i++) perm[i] = post[perm[i]];
        build_adj(perm);
        etree();
      }
    }
    phase("postorder + relabel");
    // ---- column counts (Gilbert, Ng, Peyton); labels are a postorder -----------------------------
    std::vector<int64_t> cc((size_t)n, 0);
    {
      std::vector<idx> first((size_t)n, -1), maxfirst((size_t)n, -1), prevleaf((size_t)n, -1), anc((size_t)n);
      for (idx k = 0; k < n; k++) {
        idx j = k;
        cc[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
      }
      std::iota(anc.begin(), anc.end(), 0);
      for (idx j = 0; j < n; j++) {
        if (parent[j] != -1) cc[parent[j]]--;
        for (int64_t q = xadj[j]; q < xadj[j + 1]; q++) {
          idx i = adj[q];
          if (i <= j || first[j] <= maxfirst[i]) continue;
          maxfirst[i] = first[j];
          idx jprev = prevleaf[i];
          prevleaf[i] = j;
          if (jprev == -1) { cc[j]++; continue; }
          idx qq = jprev;
          while (qq != anc[qq]) qq = anc[qq];
          for (idx s = jprev; s != qq;) { idx sp = anc[s]; anc[s] = qq; s = sp; }
          cc[j]++;
          cc[qq]--;
        }
        if (parent[j] != -1) anc[j] = parent[j];
      }
      for (idx j = 0; j < n; j++) if (parent[j] != -1) cc[parent[j]] += cc[j];
    }
    phase("column counts");
    // ---- supernodes ---------------------------------------------------------------------------------
    std::vector<idx> sfirst;   // first column of each supernode
    {
      std::vector<idx> nchild((size_t)n, 0);
      for (idx j = 0; j < n; j++) if (parent[j] != -1) nchild[parent[j]]++;
      for (idx j = 0; j < n; j++) {
        bool merge = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1 && nchild[j] == 1;
        if (o.schur_n > 0 && j == (idx)(n - o.schur_n)) merge = false;   // the Schur block starts its own supernode
        if (!merge) sfirst.push_back(j);
      }
    }
    const idx ns0 = (idx)sfirst.size();
    S->nsuper_fund = ns0;
    sfirst.push_back((idx)n);
    std::vector<idx> col2s((size_t)n);
    for (idx s = 0; s < ns0; s++) for (idx j = sfirst[s]; j < sfirst[s + 1]; j++) col2s[j] = s;
    std::vector<idx> sparent((size_t)ns0, -1);
    std::vector<int64_t> sw((size_t)ns0), sbelow((size_t)ns0);
    double nnz0 = 0;
    for (idx s = 0; s < ns0; s++) {
      idx l = sfirst[s + 1] - 1;
      sparent[s] = parent[l] == -1 ? -1 : col2s[parent[l]];
      sw[s] = sfirst[s + 1] - sfirst[s];
      sbelow[s] = cc[sfirst[s]] - sw[s];
      nnz0 += 0.5 * (double)sw[s] * (double)(sw[s] + 1) + (double)sw[s] * (double)sbelow[s];
    }
    phase("supernodes");
    // ---- amalgamation: cheapest merges first until the fill budget is spent ----------------------
    // merged node keeps the parent's id; `rep` = union-find to the surviving node
    std::vector<idx> rep((size_t)ns0);
    std::iota(rep.begin(), rep.end(), 0);
    auto find = [&](idx x) { while (rep[x] != x) { rep[x] = rep[rep[x]]; x = rep[x]; } return x; };
    {
      // Lazy heap: the cost of merging c into its (surviving) parent p, w_c (w_p + below_p - below_c) extra entries of L,
      // only GROWS while the algorithm runs (c and p widen as they absorb nodes; a parent merged into its own parent hands
      // c a wider one), so a popped entry whose stored cost is stale is a lower bound: re-evaluate, push back, go on.
      // No per-merge refresh of all the children of the merged node (that was quadratic in the fan-out).
      struct Ent { double cost; idx c; };
      auto cmp = [](const Ent& a, const Ent& b) { return a.cost > b.cost || (a.cost == b.cost && a.c < b.c); };   // (ties: the node nearer the root first)
      std::vector<Ent> heap0;                                  // (built in one make_heap: O(n) instead of n pushes)
      auto cost_of = [&](idx c, idx p) {
        return (double)sw[c] * ((double)sw[p] + (double)sbelow[p] - (double)sbelow[c]);
      };
      // first every merge that adds no fill (kass does the same before it starts its heap, amalgamate.c:318-372);
      // children before parents, so chains collapse in one pass
      for (idx s = 0; s < ns0; s++) {
        if (sparent[s] == -1) continue;
        const idx p = find(sparent[s]);
        if (cost_of(s, p) > 0) continue;
        if (o.schur_n > 0 && (sfirst[s] >= n - o.schur_n) != (sfirst[p] >= n - o.schur_n)) continue;
        rep[s] = p;
        sw[p] += sw[s];
      }
      const double budget = ratio * nnz0;
      double spent = 0;
      const int64_t maxw_merge = o.max_merge_width > 0 ? o.max_merge_width : (int64_t)1 << 40;
      auto schur_cross = [&](idx c, idx p) {
        return o.schur_n > 0 && (sfirst[c] >= n - o.schur_n) != (sfirst[p] >= n - o.schur_n);
      };
      // ---- rounds: the same greedy sequence, the cheap merges of a round found subtree by subtree on host threads ----
      // The costs the heap pops never decrease (above), so all merges of cost <= C happen before any dearer one; and an
      // edge (c, parent) whose cost exceeds C now exceeds it for good.  Cutting those edges leaves components that cannot
      // influence each other while only merges of cost <= C are made: a cost reads the widths of c and of its parent
      // only, and a component's root does not merge upward in this round.  Every component runs the heap loop on its own
      // -- the global sequence restricted to it -- and the union is what the single heap would have produced, provided the
      // budget cannot run out inside the round: C = (budget - spent) / (nodes left) guarantees that.  The rounds stop
      // when few nodes are left or a round achieves little; the single heap finishes (it must find the merge at which
      // the budget ends).
      std::vector<idx> alive;
      alive.reserve((size_t)ns0);
      for (idx s = 0; s < ns0; s++) if (rep[s] == s && sparent[s] != -1) alive.push_back(s);
      const int nthr = [] {
        const char* e = getenv("PASTIX_AMD_PLAN_THREADS");
        const int nn = e ? atoi(e) : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(nn, 64));
      }();
      {
        std::vector<idx> croot((size_t)ns0), cstart((size_t)ns0 + 1, 0), members, comps;
        std::iota(croot.begin(), croot.end(), 0);
        for (int round = 0; round < 16 && alive.size() > 50000; round++) {
          const double C = std::floor((budget - spent) / (double)alive.size());
          if (C < 1) break;
          // exact parents and component roots, parents before children (a parent has the larger index)
          size_t nmem = 0;
          for (size_t i = alive.size(); i-- > 0;) {
            const idx c = alive[i], p = find(sparent[c]);
            sparent[c] = p;
            const bool cut = cost_of(c, p) > C || schur_cross(c, p);
            croot[c] = cut ? c : croot[p];
            if (!cut) { cstart[(size_t)croot[c] + 1]++; nmem++; }
          }
          if (nmem == 0) break;
          // the components with members, in ascending root order (a root is alive or a tree root: scan the counters)
          comps.clear();
          for (idx r = 0; r < ns0; r++) if (cstart[(size_t)r + 1] > 0) comps.push_back(r);
          members.resize(nmem);
          std::vector<int64_t> cofs(comps.size() + 1, 0);
          for (size_t q = 0; q < comps.size(); q++) cofs[q + 1] = cofs[q] + cstart[(size_t)comps[q] + 1];
          for (size_t q = 0; q < comps.size(); q++) cstart[(size_t)comps[q] + 1] = (idx)q;      // root -> component number
          {
            std::vector<int64_t> pos(cofs.begin(), cofs.end() - 1);
            for (size_t i = 0; i < alive.size(); i++) {
              const idx c = alive[i], r = croot[c];
              if (r != c) members[(size_t)pos[(size_t)cstart[(size_t)r + 1]]++] = c;
            }
          }
          for (size_t q = 0; q < comps.size(); q++) cstart[(size_t)comps[q] + 1] = 0;            // (clean for the next round)
          std::vector<double> tspent((size_t)nthr, 0.0);
          std::vector<int> tbad((size_t)nthr, 0);
          std::atomic<size_t> next{0};
          auto work = [&](int t) {
            try {
              std::vector<Ent> heap;
              double ls = 0;
              for (;;) {
                const size_t q0 = next.fetch_add(64);
                if (q0 >= comps.size()) break;
                for (size_t q = q0; q < std::min(comps.size(), q0 + 64); q++) {
                  heap.clear();
                  for (int64_t i = cofs[q]; i < cofs[q + 1]; i++) {
                    const idx c = members[(size_t)i];
                    heap.push_back(Ent{cost_of(c, sparent[c]), c});
                  }
                  std::make_heap(heap.begin(), heap.end(), cmp);
                  while (!heap.empty()) {
                    std::pop_heap(heap.begin(), heap.end(), cmp);
                    const Ent e = heap.back();
                    heap.pop_back();
                    if (e.cost > C) break;
                    const idx c = e.c;
                    idx p = sparent[c];
                    while (rep[p] != p) { rep[p] = rep[rep[p]]; p = rep[p]; }
                    sparent[c] = p;
                    const double cost = cost_of(c, p);
                    if (cost > e.cost) { heap.push_back(Ent{cost, c}); std::push_heap(heap.begin(), heap.end(), cmp); continue; }
                    if (cost > 0 && sw[c] + sw[p] > maxw_merge) continue;
                    ls += std::max(0.0, cost);
                    rep[c] = p;
                    sw[p] += sw[c];
                  }
                }
              }
              tspent[(size_t)t] = ls;
            } catch (const std::bad_alloc&) { tbad[(size_t)t] = 1; }
          };
          {
            std::vector<std::thread> th;
            for (int t = 1; t < nthr; t++) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
          }
          for (int t = 0; t < nthr; t++) if (tbad[(size_t)t]) throw std::bad_alloc();
          for (int t = 0; t < nthr; t++) spent += tspent[(size_t)t];      // (integer-valued doubles: exact in any order)
          const size_t before = alive.size();
          size_t k2 = 0;
          for (size_t i = 0; i < alive.size(); i++) if (rep[alive[i]] == alive[i]) alive[k2++] = alive[i];
          alive.resize(k2);
          if (before - k2 < before / 50) break;
        }
      }
      heap0.reserve(alive.size());
      for (const idx s : alive) heap0.push_back(Ent{cost_of(s, find(sparent[s])), s});
      std::priority_queue<Ent, std::vector<Ent>, decltype(cmp)> pq(cmp, std::move(heap0));
      while (!pq.empty()) {
        const Ent e = pq.top();
        pq.pop();
        const idx c = e.c;
        if (rep[c] != c || sparent[c] == -1) continue;
        const idx p = find(sparent[c]);
        sparent[c] = p;
        const double cost = cost_of(c, p);
        if (cost > e.cost) { pq.push(Ent{cost, c}); continue; }                     // stale: a lower bound, see above
        if (schur_cross(c, p)) continue;                       // never across the Schur boundary
        if (cost > 0 && spent + cost > budget) break;          // cheapest remaining does not fit
        if (cost > 0 && sw[c] + sw[p] > maxw_merge) continue;
        spent += std::max(0.0, cost);
        rep[c] = p;                                            // merge c into p
        sw[p] += sw[c];
      }
      for (idx s = 0; s < ns0; s++)
        if (rep[s] == s && sparent[s] != -1) sparent[s] = find(sparent[s]);
    }
    phase("amalgamation");
    // ---- new ordering: postorder of the amalgamated tree, members in original order --------------
    std::vector<idx> aid((size_t)ns0, -1);      // fundamental supernode -> amalgamated node (dense ids)
    std::vector<idx> afirst;                    // first new column of each amalgamated node
    std::vector<idx> newlab((size_t)n);         // postorder label -> new label
    {
      std::vector<std::vector<idx>> members((size_t)ns0), akids((size_t)ns0);
      std::vector<idx> roots;
      for (idx s = 0; s < ns0; s++) members[find(s)].push_back(s);
      for (idx s = 0; s < ns0; s++) {
        if (rep[s] != s) continue;
        idx p = sparent[s] == -1 ? -1 : find(sparent[s]);
        if (p == -1) roots.push_back(s); else akids[p].push_back(s);
      }
      idx na = 0, col = 0;
      std::vector<std::pair<idx, size_t>> st;
      for (idx r : roots) {
        st.emplace_back(r, 0);
        while (!st.empty()) {
          auto& top = st.back();
          if (top.second < akids[top.first].size()) { idx c = akids[top.first][top.second++]; st.emplace_back(c, 0); }
          else {
            idx a = top.first;
            afirst.push_back(col);
            for (idx m : members[a]) { aid[m] = na; for (idx j = sfirst[m]; j < sfirst[m + 1]; j++) newlab[j] = col++; }
            na++;
            st.pop_back();
          }
        }
      }
      afirst.push_back(col);
      S->nsuper_amalg = na;
    }
    for (int64_t i = 0; i < n; i++) perm[i] = newlab[perm[i]];
    const idx na = (idx)afirst.size() - 1;
    build_adj(perm);
    std::vector<idx> col2a((size_t)n);
    for (idx a = 0; a < na; a++) for (idx j = afirst[a]; j < afirst[a + 1]; j++) col2a[j] = a;

    phase("reorder + adjacency");
    // ---- supernodal symbolic factorization on interval lists -------------------------------------
    // struct(a) = rows > last col of a reached from A's columns of a or from children's structs
    std::vector<std::vector<idx>> akids((size_t)na);
    std::vector<Interval> tmp;
    std::vector<std::vector<Interval>> keep((size_t)na);
    for (idx a = 0; a < na; a++) {
      const idx last = afirst[a + 1] - 1;
      tmp.clear();
      for (idx j = afirst[a]; j <= last; j++)
        for (int64_t q = xadj[j]; q < xadj[j + 1]; q++)
          if (adj[q] > last) tmp.push_back(Interval{adj[q], adj[q]});
      for (idx c : akids[a])
        for (const Interval& iv : keep[c]) {
          if (iv.b <= last) continue;
          tmp.push_back(Interval{std::max<idx>(iv.a, last + 1), iv.b});
        }
      std::sort(tmp.begin(), tmp.end(), [](const Interval& x, const Interval& y) { return x.a < y.a; });
      std::vector<Interval>& r = keep[a];
      for (const Interval& iv : tmp) {
        if (!r.empty() && iv.a <= r.back().b + 1) r.back().b = std::max(r.back().b, iv.b);
        else r.push_back(iv);
      }
      r.shrink_to_fit();
      if (!r.empty()) akids[col2a[r.front().a]].push_back(a);
    }
    std::vector<int64_t>().swap(xadj);
    std::vector<idx>().swap(adj);

    phase("interval symbolic");
    // ---- split wide nodes; final cblk boundaries -------------------------------------------------
    const idx maxbs = (idx)o.max_blocksize;
    std::vector<idx> cfirst;            // first column of each final cblk
    std::vector<idx> a_cblk0((size_t)na + 1);
    for (idx a = 0; a < na; a++) {
      a_cblk0[a] = (idx)cfirst.size();
      idx w = afirst[a + 1] - afirst[a];
      if (o.schur_n > 0 && afirst[a] >= (idx)(n - o.schur_n)) { cfirst.push_back(afirst[a]); continue; }   // the Schur cblk is not split
      if (!o.blend_split) {
        for (idx c = afirst[a]; c < afirst[a + 1]; c += maxbs) cfirst.push_back(c);
        continue;
      }
      // blend's splitOnProcs (src/blend/src/splitpart.c:387-516, DOF_CONSTANT build, dof 1).  One candidate processor:
      // a cblk no wider than IPARM_MAX_BLOCKSIZE stays; else nseq = width / max pieces of width / nseq columns, the
      // last one taking the remainder.  Several candidates (abs = 4, the reference's default): the piece width is
      // width / (abs * procs) clamped to [IPARM_MIN_BLOCKSIZE, IPARM_MAX_BLOCKSIZE].  In both cases "no parallelism
      // available above 4 splitted cblk": fewer than 4 pieces -> the cblk is left whole (:479-481).
      idx nseq;
      if (o.candidate_procs == 1) {
        if (w <= maxbs) { cfirst.push_back(afirst[a]); continue; }
        nseq = w / maxbs;
      } else {
        idx pas = w / (4 * (idx)o.candidate_procs);
        pas = std::max<idx>(pas, (idx)o.min_blocksize);
        pas = std::min<idx>(pas, maxbs);
        nseq = w / pas;
      }
      if (nseq < 4) { cfirst.push_back(afirst[a]); continue; }
      const idx pas = w / nseq;
      for (idx q = 0; q < nseq; q++) cfirst.push_back(afirst[a] + pas * q);
    }
    a_cblk0[na] = (idx)cfirst.size();
    if (o.schur_n > 0 && (o.schur_n > n || cfirst.back() != (idx)(n - o.schur_n))) {   // the last unknowns were not a clique
      delete S;
      return PASTIX_AMD_ERR_BADPARAMETER;
    }
    const idx ncb = (idx)cfirst.size();
    cfirst.push_back((idx)n);
    std::vector<idx> col2c((size_t)n);
    for (idx c = 0; c < ncb; c++) for (idx j = cfirst[c]; j < cfirst[c + 1]; j++) col2c[j] = c;

    phase("split");
    // ---- bloks -------------------------------------------------------------------------------------
    S->cblk.resize((size_t)ncb + 1);
    int64_t nnzl = 0;
    for (idx a = 0; a < na; a++) {
      for (idx c = a_cblk0[a]; c < a_cblk0[a + 1]; c++) {
        pastix_amd_cblk_t& cb = S->cblk[c];
        cb.fcolnum = cfirst[c];
        cb.lcolnum = cfirst[c + 1] - 1;
        cb.bloknum = (int64_t)S->blok.size();
        int64_t off = 0;
        auto push = [&](idx ra, idx rb) {   // rows [ra, rb], cut at facing cblk boundaries
          while (ra <= rb) {
            idx fc = col2c[ra];
            idx e = std::min<idx>(rb, cfirst[fc + 1] - 1);
            S->blok.push_back(pastix_amd_blok_t{ra, e, fc, off});
            off += e - ra + 1;
            ra = e + 1;
          }
        };
        push((idx)cb.fcolnum, (idx)cb.lcolnum);                       // diagonal blok
        if (cb.lcolnum + 1 <= afirst[a + 1] - 1) push((idx)cb.lcolnum + 1, afirst[a + 1] - 1);   // rest of the node
        for (const Interval& iv : keep[a]) push(iv.a, iv.b);
        cb.stride = off;
        int64_t w = cb.lcolnum - cb.fcolnum + 1;
        nnzl += off * w - w * (w - 1) / 2;
      }
      std::vector<Interval>().swap(keep[a]);
    }
    S->cblk[ncb].fcolnum = n; S->cblk[ncb].lcolnum = n; S->cblk[ncb].bloknum = (int64_t)S->blok.size(); S->cblk[ncb].stride = 0;
    phase("bloks");
    S->nnzl = nnzl;
    S->perm.resize((size_t)n);
    S->invp.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) { S->perm[i] = perm[i]; S->invp[perm[i]] = i; }
  } catch (const std::bad_alloc&) {
    delete S;
    return PASTIX_AMD_ERR_ALLOC;
  }
  *out = S;
  return PASTIX_AMD_OK;
}

}  // extern "C"
