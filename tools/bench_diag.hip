// Phase timing of k_diag_llt_w on one 128x128 SPD blok.  Build: hipcc --offload-arch=gfx950 -O3 -DDIAG_PROFILE -I pastix_amd/csrc -I include -o tools/bench_diag tools/bench_diag.hip
#include "../pastix_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
using namespace pastix_amd;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  int w = argc > 1 ? atoi(argv[1]) : 128;
  std::vector<double> h((size_t)w * w);
  for (int c = 0; c < w; c++) for (int r = 0; r < w; r++) h[r + (size_t)c * w] = (r == c) ? 4.0 * w : 1.0 / (1 + abs(r - c));
  double *dA, *dws; long long* dnp; int* derr; PanelTask* dt;
  CK(hipMalloc(&dA, h.size() * 8)); CK(hipMalloc(&dws, 16384 * 8)); CK(hipMalloc(&dnp, 16)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&dt, sizeof(PanelTask)));
  PanelTask t{0, w, w, 0};
  CK(hipMemcpy(dt, &t, sizeof(t), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_diag_llt_w, dim3(1), dim3(256), 0, 0, dA, dt, dws, 1e-30, dnp, derr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double st[6]; CK(hipMemcpy(st, dws + 8192, sizeof(st), hipMemcpyDeviceToHost));
    printf("w=%d: %.1f us | cycles(100MHz ticks?): load %.0f  A(tile) %.0f  B(rows+inv) %.0f  C(syrk) %.0f  tail %.0f  store %.0f\n", w, ms * 1e3, st[0], st[1], st[2], st[3], st[4], st[5]);
  }
  return 0;
}
