// Probe for gfx950: v_mfma_f64_16x16x4_f64 lane layout + issue rate, f64 global atomics rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_f64 tools/probe_mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// C(16x16) = A(16x4) * B(4x16); A,B,C row-major in global memory
__global__ void k_layout(const double* A, const double* B, double* C) {
  int lane = threadIdx.x;
  double a = A[(lane & 15) * 4 + (lane >> 4)];      // A[i=lane&15][k=lane>>4]
  double b = B[(lane >> 4) * 16 + (lane & 15)];     // B[k=lane>>4][j=lane&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = c[r];   // row=(lane>>4)+4r, col=lane&15
}

template <int NACC>
__global__ void k_rate(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  d4 c[NACC];
  for (int i = 0; i < NACC; i++) c[i] = d4{0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

__global__ void k_atomic(double* dst, long n, int reps) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (int r = 0; r < reps; r++)
    for (long j = i; j < n; j += stride) unsafeAtomicAdd(&dst[j], 1.0);
}
__global__ void k_rmw(double* dst, long n, int reps) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (int r = 0; r < reps; r++)
    for (long j = i; j < n; j += stride) dst[j] -= 1.0;
}

template <int NACC>
void run_rate(int waves_per_simd) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  int threads = 64 * 4 * waves_per_simd;
  int iters = 20000;
  double* out; CK(hipMalloc(&out, sizeof(double) * cus * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_rate<NACC><<<cus, threads>>>(out, 100); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<NACC><<<cus, threads>>>(out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double cyc; CK(hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost));
  double nm = (double)iters * NACC;            // MFMAs per wave
  double flops = nm * 2048.0 * (threads / 64) * cus;
  printf("NACC=%d waves/SIMD=%d: %.1f cycles(clock64)/MFMA/wave, %.2f TFLOP/s chip, %.3f ms\n", NACC,
         waves_per_simd, cyc / nm, flops / (ms * 1e-3) * 1e-12, ms);
  CK(hipFree(out));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz mem=%.1f GB\n", p.gcnArchName, p.multiProcessorCount, p.clockRate,
         p.totalGlobalMem / 1e9);
  // layout check with asymmetric data
  std::vector<double> A(64), B(64), C(256), R(256, 0.0);
  for (int i = 0; i < 64; i++) { A[i] = 1 + i * 0.5; B[i] = 3 - i * 0.25 + (i % 5); }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  k_layout<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
  double err = 0; for (int i = 0; i < 256; i++) err = fmax(err, fabs(C[i] - R[i]));
  printf("layout check max err = %g (%s)\n", err, err == 0 ? "OK" : "MISMATCH");
  run_rate<1>(1); run_rate<2>(1); run_rate<4>(1); run_rate<8>(1);
  run_rate<4>(2); run_rate<4>(4);
  // atomics vs plain RMW over a 1 GiB buffer
  long n = 1L << 27; double* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  k_atomic<<<2048, 256>>>(d, n, 1); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k_atomic<<<2048, 256>>>(d, n, 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("f64 atomic add: %.2f TB/s of added bytes\n", 4.0 * n * 8 / (ms * 1e-3) * 1e-12);
  CK(hipEventRecord(e0)); k_rmw<<<2048, 256>>>(d, n, 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("f64 plain rmw : %.2f TB/s of updated bytes (x2 traffic)\n", 4.0 * n * 8 / (ms * 1e-3) * 1e-12);
  return 0;
}
