// Replays ONE bulk launch of a real factorization plan (dumped with PASTIX_AMD_DEV=dump_slot=<slot>:<file>, api.cpp) on an arena of the
// same size filled with noise, under different task orders:
//   0  as planned (heaviest first, ties in tile order = target cblk major, row tile minor)
//   1  2-D blocks over (A rows, B rows) of the tasks' first pieces: runs of 512 tasks = 16 x 32 blocks, dealt so that
//      workgroup g (-> XCD g % 8) belongs to an 8 x 8 sub-block
//   2  random
// Build: hipcc --offload-arch=gfx950 -O3 -I pastix_amd/csrc -o tools/replay_slot tools/replay_slot.hip
#include "../pastix_amd/csrc/kernels_update.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
using namespace pastix_amd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const char* fn = argc > 1 ? argv[1] : "/tmp/pastix_amd_slot.bin";
  FILE* f = fopen(fn, "rb");
  if (!f) { printf("cannot open %s\n", fn); return 1; }
  int64_t hdr[4];
  if (fread(hdr, sizeof(hdr), 1, f) != 1) return 1;
  std::vector<Task> tasks((size_t)hdr[1]);
  std::vector<Piece> pieces((size_t)hdr[2]);
  if (fread(tasks.data(), sizeof(Task), tasks.size(), f) != tasks.size()) return 1;
  if (fread(pieces.data(), sizeof(Piece), pieces.size(), f) != pieces.size()) return 1;
  fclose(f);
  const int64_t coefnbr = hdr[0];
  if (const char* fm = getenv("MODES")) {        // MODES=02: keep only the tasks run by these instances of the update loop
    std::vector<Task> keep;
    for (const Task& t : tasks) {
      const int m = (int)t.nfull == t.pn ? ((t.tm == 128 && t.tn == 128) ? 0 : 1) : 2;
      if (strchr(fm, '0' + m)) keep.push_back(t);
    }
    tasks.swap(keep);
  }
  double fl = 0, full = 0;
  for (const Task& t : tasks)
    for (int i = 0; i < t.pn; i++) {
      const Piece& p = pieces[(size_t)t.p0 + i];
      fl += 2.0 * p.m * (double)p.n * p.k;
      if (i < (int)t.nfull) full += 2.0 * p.m * (double)p.n * p.k;
    }
  printf("slot %lld: %zu tasks, %zu pieces, %.3e flops (%.1f%% in whole-tile pieces), arena %.1f GB\n", (long long)hdr[3],
         tasks.size(), pieces.size(), fl, 100 * full / fl, coefnbr * 8e-9);
  {  // task mix: which instance of the update loop runs (0 full tile, 1 edge tile, 2 has partial pieces), K per task
    double f[3] = {0, 0, 0}, kk[3] = {0, 0, 0}, exec2 = 0; long n[3] = {0, 0, 0};
    long hist[6] = {0, 0, 0, 0, 0, 0};
    for (const Task& t : tasks) {
      const int m = (int)t.nfull == t.pn ? ((t.tm == 128 && t.tn == 128) ? 0 : 1) : 2;
      double tf = 0, tk = 0;
      for (int i = 0; i < t.pn; i++) { const Piece& p = pieces[(size_t)t.p0 + i]; tf += 2.0 * p.m * (double)p.n * p.k; tk += ((p.k + 15) / 16) * 16;
        if (m == 2) { const int rs = (p.dr + p.m + 15) / 16 - p.dr / 16, cs = (p.dc + p.n + 15) / 16 - p.dc / 16; exec2 += 2.0 * 256.0 * rs * cs * ((p.k + 15) / 16 * 16); } }
      f[m] += tf; kk[m] += tk; n[m]++;
      hist[tk <= 128 ? 0 : tk <= 256 ? 1 : tk <= 512 ? 2 : tk <= 1024 ? 3 : tk <= 2048 ? 4 : 5]++;
    }
    for (int m = 0; m < 3; m++) printf("  mode %d: %ld tasks, %.1f%% of the flops, chunk-lines per task %.0f\n", m, n[m], 100 * f[m] / fl, n[m] ? kk[m] / n[m] : 0.0);
    printf("  mode 2: executed band flops / useful = %.2f; tasks by K (<=128,256,512,1024,2048,more): %ld %ld %ld %ld %ld %ld\n", f[2] > 0 ? exec2 / f[2] : 0.0, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
  }
  char* raw;
  CK(hipMalloc(&raw, (size_t)coefnbr * 8 + 512));
  double* d = (double*)(raw + 256);
  {  // noise without a host copy of the arena
    std::vector<double> h(1 << 22);
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> u(-1e-3, 1e-3);
    for (auto& v : h) v = u(rng);
    for (int64_t o = 0; o < coefnbr; o += (int64_t)h.size())
      CK(hipMemcpy(d + o, h.data(), (size_t)std::min<int64_t>((int64_t)h.size(), coefnbr - o) * 8, hipMemcpyHostToDevice));
  }
  Piece* dp; CK(hipMalloc(&dp, pieces.size() * sizeof(Piece)));
  CK(hipMemcpy(dp, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice));
  Task* dt; CK(hipMalloc(&dt, tasks.size() * sizeof(Task)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 4;
  for (int order = 0; order < 3; order++) {
    std::vector<Task> t2(tasks);
    const size_t n = t2.size();
    if (order == 1) {
      // ranks of the first piece's A / B rows
      std::vector<int64_t> ra(n), rb(n), idx(n);
      auto rank_by = [&](bool a, std::vector<int64_t>& out) {
        std::iota(idx.begin(), idx.end(), 0);
        auto key = [&](int64_t q) { const Piece& p = pieces[(size_t)tasks[(size_t)q].p0]; return a ? p.a_off : p.b_off; };
        std::sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return key(x) < key(y); });
        int64_t r = -1, last = -1;
        for (size_t i = 0; i < n; i++) { const int64_t k = key(idx[i]); if (i == 0 || k != last) { r++; last = k; } out[(size_t)idx[i]] = r; }
      };
      rank_by(true, ra); rank_by(false, rb);
      std::vector<int64_t> ord(n);
      std::iota(ord.begin(), ord.end(), 0);
      // superblock (ra/16, rb/32), inside it sub-block (ra/8 % 2, rb/8 % 4) = XCD, inside it (ra, rb)
      auto key = [&](int64_t q) {
        const int64_t a = ra[(size_t)q], b = rb[(size_t)q];
        return std::make_tuple(b / 32, a / 16, ((a / 8) % 2) * 4 + (b / 8) % 4, a % 8, b % 8);
      };
      std::sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return key(x) < key(y); });
      // deal every superblock's sub-blocks round-robin so that position % 8 = sub-block
      std::vector<Task> out; out.reserve(n);
      for (size_t i = 0; i < n;) {
        size_t j = i;
        std::vector<std::vector<int64_t>> L(8);
        const auto sb = std::make_pair(std::get<0>(key(ord[i])), std::get<1>(key(ord[i])));
        while (j < n && std::make_pair(std::get<0>(key(ord[j])), std::get<1>(key(ord[j]))) == sb) { L[(size_t)std::get<2>(key(ord[j]))].push_back(ord[j]); j++; }
        for (size_t v = 0;; v++) {
          bool any = false;
          for (int x = 0; x < 8; x++) if (v < L[(size_t)x].size()) { out.push_back(tasks[(size_t)L[(size_t)x][v]]); any = true; }
          if (!any) break;
        }
        i = j;
      }
      t2.swap(out);
    } else if (order == 2) {
      std::mt19937_64 rng(7);
      std::shuffle(t2.begin(), t2.end(), rng);
    }
    CK(hipMemcpy(dt, t2.data(), t2.size() * sizeof(Task), hipMemcpyHostToDevice));
    launch_update(0, Arenas{{d, d, d, d}}, dt, dp, (int64_t)n, false);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch_update(0, Arenas{{d, d, d, d}}, dt, dp, (int64_t)n, false);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("order %d: %.3f ms/launch, %.1f TFLOP/s\n", order, ms / reps, fl * reps / (ms * 1e-3) * 1e-12);
  }
  return 0;
}
