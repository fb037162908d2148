// partition.cpp -- host only: which GPU factorizes which cblk (the multi-GPU driver's `owner` map, dist.cpp).
//
// Proportional mapping on the cblk elimination tree, the idea of PaStiX's blend (splitpart.c:752-1012 `propMappTree` /
// candidate sets, blend_distributeOnGPU.c:59-317 for the GPU colouring): a subtree is given a SET of ranks (its
// candidates); the chain of cblks at its top -- the column groups of one separator -- is dealt over that set, heaviest
// first onto the least loaded rank; where the tree branches the set is divided among the heavy children in proportion
// to their work; a subtree with one rank goes to it whole; light side subtrees (< `light` of their parent's work) go
// whole to the least loaded rank of the set.  Fan-in traffic therefore stays inside the rank set of the enclosing
// subtree: a rank only contributes to separators on its own path to the root.  Work = the cblk's share of
// DPARM_FACT_FLOPS (blend_symbol_cost.c:382-430, the LLt formula).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <new>
#include <utility>
#include <vector>

#include "../../include/pastix_amd.h"

namespace {

// divide the rank list among weights.size() <= ranks.size() children, proportionally, at least one each
std::vector<std::vector<int>> split_ranks(const std::vector<int>& ranks, const std::vector<double>& w) {
  const int m = (int)ranks.size(), c = (int)w.size();
  double tot = 0;
  for (double x : w) tot += x;
  std::vector<int> cnt((size_t)c);
  for (int j = 0; j < c; j++) cnt[(size_t)j] = std::max(1, (int)std::nearbyint((double)m * w[(size_t)j] / tot));   // (ties to even)
  auto sum = [&]() { int s = 0; for (int x : cnt) s += x; return s; };
  while (sum() > m) {
    int best = 0;
    for (int j = 1; j < c; j++) {                        // max of (cnt > 1, cnt - share), the first of equals
      const bool bj = cnt[(size_t)j] > 1, bb = cnt[(size_t)best] > 1;
      const double ej = cnt[(size_t)j] - (double)m * w[(size_t)j] / tot, eb = cnt[(size_t)best] - (double)m * w[(size_t)best] / tot;
      if ((bj && !bb) || (bj == bb && ej > eb)) best = j;
    }
    cnt[(size_t)best]--;
  }
  while (sum() < m) {
    int best = 0;
    for (int j = 1; j < c; j++)
      if ((double)m * w[(size_t)j] / tot - cnt[(size_t)j] > (double)m * w[(size_t)best] / tot - cnt[(size_t)best]) best = j;
    cnt[(size_t)best]++;
  }
  std::vector<std::vector<int>> out;
  int pos = 0;
  for (int j = 0; j < c; j++) {
    out.emplace_back(ranks.begin() + pos, ranks.begin() + pos + cnt[(size_t)j]);
    pos += cnt[(size_t)j];
  }
  return out;
}

}  // namespace

extern "C" int pastix_amd_dist_partition(const pastix_amd_layout_t* L, int world, double light, int32_t* owner) {
  if (!L || !owner || !L->cblktab || !L->bloktab || L->cblknbr < 1 || world < 1 || world > 64) return PASTIX_AMD_ERR_BADPARAMETER;
  if (!(light > 0)) light = 0.05;
  const int64_t nc = L->cblknbr;
  try {
    for (int64_t k = 0; k < nc; k++) owner[k] = world == 1 ? 0 : -1;
    if (world == 1) return PASTIX_AMD_OK;
    std::vector<double> fl((size_t)nc), sub;
    std::vector<int64_t> parent((size_t)nc, -1);
    for (int64_t k = 0; k < nc; k++) {
      const auto& c = L->cblktab[k];
      const double N = (double)(c.lcolnum - c.fcolnum + 1), S = (double)c.stride, M = S - N;
      double f = N * (((1. / 6.) * N + 0.5) * N + (1. / 3.)) + N * (((1. / 6.) * N) * N - (1. / 6.)) + M * N * (N + 1.);
      const int64_t fb = c.bloknum, lb = L->cblktab[k + 1].bloknum;
      double g = 0;
      for (int64_t b = fb; b < lb; b++) {
        const auto& bl = L->bloktab[b];
        const double h = (double)(bl.lrownum - bl.frownum + 1), rem = S - (double)bl.coefind;
        g += 2.0 * rem * h * N * (bl.coefind > 0 ? 1.0 : 0.0);
      }
      fl[(size_t)k] = f + g;
      if (lb - fb > 1) parent[(size_t)k] = L->bloktab[fb + 1].cblknum;     // facing cblk of the first off-diagonal blok
    }
    sub = fl;
    std::vector<std::vector<int64_t>> kids((size_t)nc);
    for (int64_t k = 0; k < nc; k++) {                     // children have smaller indices than their parents
      const int64_t q = parent[(size_t)k];
      if (q >= 0) {
        if (q <= k || q >= nc) return PASTIX_AMD_ERR_LAYOUT;
        sub[(size_t)q] += sub[(size_t)k];
        kids[(size_t)q].push_back(k);
      }
    }
    std::vector<double> load((size_t)world, 0.0);
    std::vector<std::pair<int64_t, int>> whole;            // (subtree root, rank): everything below goes to the rank
    auto least = [&](const std::vector<int>& ranks) {
      int q = ranks[0];
      for (int r : ranks) if (load[(size_t)r] < load[(size_t)q]) q = r;
      return q;
    };
    auto give_whole = [&](int64_t root, const std::vector<int>& ranks) {
      const int q = least(ranks);
      whole.emplace_back(root, q);
      load[(size_t)q] += sub[(size_t)root];
    };
    auto by_sub = [&](int64_t a, int64_t b) { return sub[(size_t)a] > sub[(size_t)b]; };
    std::vector<int> all((size_t)world);
    for (int r = 0; r < world; r++) all[(size_t)r] = r;
    std::vector<std::pair<int64_t, std::vector<int>>> stack;
    std::vector<int64_t> roots;
    for (int64_t k = 0; k < nc; k++) if (parent[(size_t)k] < 0) roots.push_back(k);
    if (roots.size() > 1) {                                // a forest: the roots are children of a virtual node
      std::stable_sort(roots.begin(), roots.end(), by_sub);
      const size_t nh = std::min(roots.size(), (size_t)world);
      std::vector<double> w;
      for (size_t i = 0; i < nh; i++) w.push_back(sub[(size_t)roots[i]]);
      auto parts = split_ranks(all, w);
      for (size_t i = 0; i < nh; i++) stack.emplace_back(roots[i], parts[i]);
      for (size_t i = nh; i < roots.size(); i++) give_whole(roots[i], all);
    } else {
      stack.emplace_back(roots[0], all);
    }
    while (!stack.empty()) {
      int64_t node = stack.back().first;
      const std::vector<int> ranks = std::move(stack.back().second);
      stack.pop_back();
      if (ranks.size() == 1) {
        whole.emplace_back(node, ranks[0]);
        load[(size_t)ranks[0]] += sub[(size_t)node];
        continue;
      }
      std::vector<int64_t> chain, heavy;
      for (;;) {                                           // walk down the separator chain to the branching point
        chain.push_back(node);
        std::vector<int64_t> ch(kids[(size_t)node]);
        std::stable_sort(ch.begin(), ch.end(), by_sub);
        heavy.clear();
        for (int64_t c : ch) if (sub[(size_t)c] >= light * sub[(size_t)node]) heavy.push_back(c);
        for (size_t i = heavy.size(); i < ch.size(); i++) give_whole(ch[i], ranks);
        if (heavy.size() != 1) break;
        node = heavy[0];
      }
      std::stable_sort(chain.begin(), chain.end(), [&](int64_t a, int64_t b) { return fl[(size_t)a] > fl[(size_t)b]; });
      for (int64_t k : chain) {
        const int q = least(ranks);
        owner[k] = q;
        load[(size_t)q] += fl[(size_t)k];
      }
      if (heavy.empty()) continue;
      if (heavy.size() > ranks.size()) {                   // more heavy children than ranks: the lightest go whole
        for (size_t i = ranks.size(); i < heavy.size(); i++) give_whole(heavy[i], ranks);
        heavy.resize(ranks.size());
      }
      std::vector<double> w;
      for (int64_t c : heavy) w.push_back(sub[(size_t)c]);
      auto parts = split_ranks(ranks, w);
      for (size_t i = 0; i < heavy.size(); i++) stack.emplace_back(heavy[i], parts[i]);
    }
    for (auto& wq : whole) owner[wq.first] = wq.second;
    for (int64_t k = nc - 1; k >= 0; k--)                  // parents have larger indices than their children
      if (owner[k] < 0) {
        if (parent[(size_t)k] < 0) return PASTIX_AMD_ERR_LAYOUT;
        owner[k] = owner[parent[(size_t)k]];
      }
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
  return PASTIX_AMD_OK;
}
