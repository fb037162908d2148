// Synthetic micro-benchmark of k_update: N tasks x P full pieces (128x128xK), operands drawn from a
// pool of `pool` source panels.  Build: hipcc --offload-arch=gfx950 -O3 -I pastix_amd/csrc -o tools/bench_update tools/bench_update.hip
// -DUSE_LIB: time the LIBRARY's own build of the kernel (csrc/Makefile: every accumulation register reserved) instead of a
// plain hipcc build of its source:  hipcc --offload-arch=gfx950 -O3 -DUSE_LIB -I pastix_amd/csrc -o tools/bench_update_lib
// tools/bench_update.hip -L pastix_amd/lib -lpastix_amd -Wl,-rpath,'$ORIGIN/../pastix_amd/lib'
#ifdef USE_LIB
#include <hip/hip_runtime.h>
#include "../pastix_amd/csrc/plan.h"
namespace pastix_amd {
void launch_update(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks, bool urgent);
}
#else
#include "../pastix_amd/csrc/kernels_update.hip"
#endif
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
using namespace pastix_amd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  int ntask = argc > 1 ? atoi(argv[1]) : 2048, P = argc > 2 ? atoi(argv[2]) : 8, K = argc > 3 ? atoi(argv[3]) : 128;
  int pool = argc > 4 ? atoi(argv[4]) : 64;        // number of distinct source panels
  const int pm = getenv("PM") ? atoi(getenv("PM")) : 128, pn = getenv("PN") ? atoi(getenv("PN")) : 128;   // piece extents
  const int pdr = getenv("PDR") ? atoi(getenv("PDR")) : 0, pdc = getenv("PDC") ? atoi(getenv("PDC")) : 0;
  int rows = getenv("ROWS") ? atoi(getenv("ROWS")) : 4096;   // rows per source panel (= lda of the pieces)
  int64_t src_elems = (int64_t)pool * rows * K;
  int64_t c_elems = (int64_t)ntask * 128 * 128;
  double *d; CK(hipMalloc(&d, (src_elems + c_elems) * 8));
  std::vector<double> h(src_elems + c_elems);
  std::mt19937_64 rng(1); std::uniform_real_distribution<double> u(-1, 1);
  const char* fillmode = getenv("FILL");
  for (auto& v : h) v = (fillmode && fillmode[0] == 'z') ? 0.0 : (fillmode && fillmode[0] == 'c') ? 1.25 : u(rng);
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  std::vector<Task> tasks(ntask); std::vector<Piece> pieces((size_t)ntask * P);
  for (int t = 0; t < ntask; t++) {
    tasks[t] = Task{src_elems + (int64_t)t * 128 * 128, 128, 128, 128, t * P, P, 0, (unsigned)((getenv("NOFAST") || pm != 128 || pn != 128) ? 0 : P)};
    for (int p = 0; p < P; p++) {
      int s = (t * 7 + p * 13) % pool; int ra = ((t * 31 + p) % (rows / 128)) * 128, rb = ((t * 17 + 3 * p) % (rows / 128)) * 128;
      // STRUCT: the access pattern of a top-of-tree launch: task (rt, ct) of the trailing matrix reads A(rt, source p) and
      // B(ct, source p) from P source panels of `rows` rows.  1: tasks in (ct, rt) order (heaviest-first order of the
      // plan = creation order); 2: 2-D blocks -- every run of 512 tasks is a 16 x 32 block of (rt, ct), and workgroup g
      // goes to XCD g % 8, which gets an 8 x 8 sub-block
      static const int st = getenv("STRUCT") ? atoi(getenv("STRUCT")) : 0;
      if (st) {
        const int R = rows / 128;
        int rt, ct;
        if (st == 1) { ct = t / R; rt = t % R; }
        else {
          const int blk = t / 512, u = t % 512;        // block of 512 tasks; u -> XCD u % 8, slot u / 8
          const int x = u % 8, v = u / 8;              // XCD x holds the 8 x 8 sub-block (x / 4, x % 4) of the 16 x 32 block
          const int br = blk % (R / 16), bc = blk / (R / 16);
          rt = br * 16 + (x / 4) * 8 + v / 8;
          ct = bc * 32 + (x % 4) * 8 + v % 8;
        }
        s = p % pool; ra = (rt % R) * 128; rb = (ct % R) * 128;
      }
      static const int offa = getenv("OFFA") ? atoi(getenv("OFFA")) : 0, offb = getenv("OFFB") ? atoi(getenv("OFFB")) : 0;
      if (ra + 128 + offa <= rows) ra += offa;          // operand rows not aligned to 16 B / to the 128-B cache line
      if (rb + 128 + offb <= rows) rb += offb;
      pieces[(size_t)t * P + p] = Piece{(int64_t)s * rows * K + ra, (int64_t)s * rows * K + rb, rows, (uint16_t)K, (uint16_t)pdr, (uint16_t)pm, (uint16_t)pdc, (uint16_t)pn, 0};
    }
  }
  Task* dt; Piece* dp; CK(hipMalloc(&dt, tasks.size() * sizeof(Task))); CK(hipMalloc(&dp, pieces.size() * sizeof(Piece)));
  CK(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch_update(0, Arenas{{d, d, d, d}}, dt, dp, ntask, false); CK(hipDeviceSynchronize());
  int reps = getenv("REPS") ? atoi(getenv("REPS")) : 5; CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) launch_update(0, Arenas{{d, d, d, d}}, dt, dp, ntask, false);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  if (getenv("CHECK")) {
    // the first and the last tile against the host (one launch on fresh C values)
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    launch_update(0, Arenas{{d, d, d, d}}, dt, dp, ntask, false); CK(hipDeviceSynchronize());
    std::vector<double> c(128 * 128);
    double worst = 0;
    for (int t : {0, ntask / 2, ntask - 1}) {
      CK(hipMemcpy(c.data(), d + src_elems + (int64_t)t * 128 * 128, 128 * 128 * 8, hipMemcpyDeviceToHost));
      for (int j = 0; j < 128; j++)
        for (int i = 0; i < 128; i++) {
          double ref = h[src_elems + (int64_t)t * 128 * 128 + i + 128 * j];
          if (i >= pdr && i < pdr + pm && j >= pdc && j < pdc + pn)
            for (int p = 0; p < P; p++) {
              const Piece& pc = pieces[(size_t)t * P + p];
              double acc = 0;
              for (int k = 0; k < K; k++) acc += h[pc.a_off + (i - pdr) + (int64_t)k * rows] * h[pc.b_off + (j - pdc) + (int64_t)k * rows];
              ref -= acc;
            }
          worst = std::max(worst, std::fabs(ref - c[i + 128 * j]));
        }
    }
    printf("CHECK: worst |device - host| over 3 tiles = %.3e %s\n", worst, worst < 1e-9 ? "ok" : "WRONG");
  }
  double fl = 2.0 * pm * pn * K * (double)P * ntask * reps;
  printf("tasks=%d pieces/task=%d K=%d pool=%d: %.3f ms/launch, %.1f TFLOP/s (%.1f%% of 78.6)\n", ntask, P, K, pool, ms / reps,
         fl / (ms * 1e-3) * 1e-12, fl / (ms * 1e-3) / 78.6e12 * 100);
  return 0;
}
