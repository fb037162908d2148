// How fast does v_mfma_f64_16x16x4_f64 run on non-trivial data?  Pure register MFMA loop, 4 waves/SIMD,
// operands = 8 rotating registers per lane filled with (a) constants (b) uniform random values.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_power tools/probe_mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(1024) void k(const double* __restrict__ in, double* out, int iters) {
  double a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = in[(threadIdx.x * 16 + i) % 65536]; b[i] = in[(threadIdx.x * 16 + 8 + i) % 65536]; }
  d4 c[8];
  for (int i = 0; i < 8; i++) c[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + it) & 7], c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount, iters = 40000;
  std::vector<double> h(65536);
  double *din, *dout; CK(hipMalloc(&din, 65536 * 8)); CK(hipMalloc(&dout, (size_t)cus * 1024 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; mode++) {
    std::mt19937_64 rng(7); std::uniform_real_distribution<double> u(-1, 1);
    for (auto& v : h) v = mode == 0 ? 0.0 : mode == 1 ? 1.5 : u(rng);
    CK(hipMemcpy(din, h.data(), 65536 * 8, hipMemcpyHostToDevice));
    k<<<cus, 1024>>>(din, dout, 1000); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0)); k<<<cus, 1024>>>(din, dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = (double)iters * 8 * 2048.0 * 16 * cus;
      printf("data=%s: %.1f TFLOP/s (%.2f ms)\n", mode == 0 ? "zeros" : mode == 1 ? "const 1.5" : "uniform(-1,1)", fl / (ms * 1e-3) * 1e-12, ms);
    }
  }
  return 0;
}
