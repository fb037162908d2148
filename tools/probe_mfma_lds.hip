// Does LDS read traffic next to v_mfma_f64_16x16x4_f64 cost MFMA throughput?  4 waves/SIMD, 8 MFMA per
// iteration with register operands + R ds_read_b64 per iteration into otherwise unused registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_lds tools/probe_mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifndef BLK
#define BLK 1024
#endif
template <int R, bool USE>
__global__ __launch_bounds__(BLK) void k(double* out, int iters, double seed) {
  __shared__ double sh[8192];
  for (int i = threadIdx.x; i < 8192; i += BLK) sh[i] = seed * (i % 97) * 0.013 - seed * 0.5;
  __syncthreads();
  double a = seed * threadIdx.x * 1e-3, b = seed * (1.0 + threadIdx.x * 1e-4);
  d4 c[8];
  for (int i = 0; i < 8; i++) c[i] = d4{0, 0, 0, 0};
  double r[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double* p = sh + (threadIdx.x & 63);
  for (int it = 0; it < iters; it++) {
    const double* q = p + ((it & 31) * 144);
#pragma unroll
    for (int j = 0; j < R; j++) r[j] = q[j * 16];
    if (USE) { a = r[0]; b = r[1 % (R > 0 ? R : 1)]; }
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(USE ? r[i % (R > 0 ? R : 1)] : a, b, c[i], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < R; j++) asm volatile("" ::"v"(r[j]));
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int R, bool USE>
void run(const char* name, double seed) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount, iters = 100000;
  double* dout; CK(hipMalloc(&dout, (size_t)cus * 1024 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<R, USE><<<cus * (1024 / BLK), BLK>>>(dout, 1000, seed); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<R, USE><<<cus * (1024 / BLK), BLK>>>(dout, iters, seed); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = (double)iters * 8 * 2048.0 * 16 * cus;
  printf("%-34s seed %.1f: %.1f TFLOP/s\n", name, seed, fl / (ms * 1e-3) * 1e-12);
  CK(hipFree(dout));
}
int main() {
  for (double seed : {0.0, 1.0}) {
    run<0, false>("8 MFMA, no LDS reads", seed);
    run<3, false>("8 MFMA + 3 ds_read (unused)", seed);
    run<6, false>("8 MFMA + 6 ds_read (unused)", seed);
    run<6, true>("8 MFMA + 6 ds_read (operands)", seed);
    run<2, false>("8 MFMA + 2 ds_read (unused)", seed);
    run<12, false>("8 MFMA + 12 ds_read (unused)", seed);
  }
  return 0;
}
