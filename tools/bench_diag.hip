// Phase timing of k_diag_llt_w: N cblks of width w (panel leading dimension ld), one launch.
// Build: hipcc --offload-arch=gfx950 -O3 -DDIAG_PROFILE -I pastix_amd/csrc -I include -o tools/bench_diag tools/bench_diag.hip
// usage: bench_diag W [N [LD]]   (stamps are of the last workgroup that wrote them)
#include "../pastix_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
namespace pastix_amd {   // (defined in kernels_var.hip / kernels_z.hip; not linked into this tool)
void launch_run_diag_lu(hipStream_t, const Arenas&, const RunD*, const RunInfo*, int, double*, double, long long*, const RunCtl&, int*, long long) {}
void launch_run_diag_z(hipStream_t, bool, const Arenas&, const RunD*, const RunInfo*, int, double*, double, long long*, const RunCtl&, int*, long long) {}
}
using namespace pastix_amd;
// a few ms of arithmetic on every CU right before the timed launch: a 40 us kernel on an idle chip runs at an idle clock
__global__ void k_warm(double* out, int n) {
  double a = threadIdx.x, b = 1.000001;
  for (int i = 0; i < n; i++) a = __builtin_fma(a, b, 0.5);
  if (a == 123.456) out[0] = a;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int w = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 1, ld = argc > 3 ? atoi(argv[3]) : w;
  std::vector<double> h((size_t)ld * w * N, 0.0);
  for (int b = 0; b < N; b++)
    for (int c = 0; c < w; c++) for (int r = 0; r < w; r++) h[(size_t)b * ld * w + r + (size_t)c * ld] = (r == c) ? 4.0 * w : 1.0 / (1 + abs(r - c));
  double *dA, *dws; long long* dnp; int* derr; PanelTask* dt;
  const int64_t per = (int64_t)((w + 15) / 16) * 256;
  CK(hipMalloc(&dA, h.size() * 8)); CK(hipMalloc(&dws, (per * N + 16384) * 8)); CK(hipMalloc(&dnp, 16)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&dt, sizeof(PanelTask) * N));
  std::vector<PanelTask> t((size_t)N);
  for (int b = 0; b < N; b++) t[b] = PanelTask{(int64_t)b * ld * w, ld, w, per * b};
  CK(hipMemcpy(dt, t.data(), sizeof(PanelTask) * N, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_warm, dim3(1024), dim3(256), 0, 0, dws, 400000);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_diag_llt_w, dim3(N), dim3(512), 0, 0, dA, dt, dws, 1e-30, dnp, derr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double st[6]; CK(hipMemcpy(st, dws + 8192, sizeof(st), hipMemcpyDeviceToHost));
    printf("w=%d N=%d ld=%d: %.1f us = %.1f us per round of 512 | ticks: load %.0f  A(tile) %.0f  B(rows+inv) %.0f  C(syrk) %.0f  tail %.0f  store %.0f\n",
           w, N, ld, ms * 1e3, ms * 1e3 / ((N + 511) / 512), st[0], st[1], st[2], st[3], st[4], st[5]);
  }
  return 0;
}
