// Which feature of k_update's inner structure costs MFMA issue slots?  Variants of an LDS-fed MFMA loop
// (2x4 accumulator tiles per wave, 6 ds_read_b64 per k-step as in k_update<8>), 512-thread blocks, 2 per CU.
//   V0: flat loop of k-steps
//   V1: k-steps grouped by 4 with a block of scalar bookkeeping between groups (like the chunk loop)
//   V2: V1 + __syncthreads() per group
//   V3: V2 + 8 global loads + 8 LDS stores per group (double-buffered staging)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_loop tools/probe_mfma_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int SLD = 144;
template <int V>
__global__ __launch_bounds__(512, 4) void k(const double* __restrict__ gin, double* out, int groups, const int* __restrict__ meta) {
  __shared__ double sh[2][2][16 * SLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 2 * 16 * SLD; i += 512) (&sh[0][0][0])[i] = gin[i % 4096];
  __syncthreads();
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, g = lane >> 4;
  d4 acc[2][4];
  for (int a = 0; a < 2; a++) for (int b = 0; b < 4; b++) acc[a][b] = d4{0, 0, 0, 0};
  int buf = 0;
  int book = meta[blockIdx.x & 63];
  double st[8];
  const double* gp = gin + (size_t)blockIdx.x * 4096 + tid;
  for (int gr = 0; gr < groups; gr++) {
    if (V >= 1) {   // scalar bookkeeping comparable to the piece/chunk state machine
      int m = meta[(gr + book) & 63];
#pragma unroll
      for (int i = 0; i < 24; i++) m = (m * 1103515245 + 12345) >> (i & 3);
      book = (book + (m & 1)) & 63;
    }
    if (V >= 3) {
#pragma unroll
      for (int q = 0; q < 8; q++) st[q] = gp[(size_t)((gr * 8 + q) & 255) * 512];
    }
    const double* sA = sh[buf][0] + wr * 32 + l15;
    const double* sB = sh[buf][1] + wc * 64 + l15;
    for (int ks = 0; ks < 4; ks++) {
      const int kk = (ks * 4 + g) * SLD;
      double bm[2], an[4];
      for (int s = 0; s < 2; s++) bm[s] = sA[kk + s * 16];
      for (int s = 0; s < 4; s++) an[s] = sB[kk + s * 16];
#pragma unroll
      for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[ni], bm[mi], acc[mi][ni], 0, 0, 0);
    }
    if (V >= 3) {
      double* dA = sh[buf ^ 1][0];
#pragma unroll
      for (int q = 0; q < 8; q++) dA[((tid >> 7) + 4 * (q & 3)) * SLD + (tid & 127) + (q >> 2) * 16 * SLD] = st[q];
    }
    if (V >= 2) __syncthreads();
    if (V >= 3) buf ^= 1;
  }
  double s = 0;
  for (int a = 0; a < 2; a++) for (int b = 0; b < 4; b++) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * 512 + tid] = s + book;
}
template <int V>
void run(const char* name) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount, groups = 4000, nblk = cus * 2;
  double *din, *dout; int* dmeta;
  CK(hipMalloc(&din, (size_t)(nblk + 300) * 4096 * 8 + 512 * 256 * 8)); CK(hipMemset(din, 0, (size_t)(nblk + 300) * 4096 * 8 + 512 * 256 * 8));
  CK(hipMalloc(&dout, (size_t)nblk * 512 * 8)); CK(hipMalloc(&dmeta, 64 * 4)); CK(hipMemset(dmeta, 0, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<V><<<nblk, 512>>>(din, dout, 100, dmeta); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<V><<<nblk, 512>>>(din, dout, groups, dmeta); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = (double)groups * 32 * 2048.0 * 8 * nblk;
  printf("%-52s %.1f TFLOP/s\n", name, fl / (ms * 1e-3) * 1e-12);
  CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dmeta));
}
int main() {
  run<0>("V0 flat k-steps");
  run<1>("V1 + scalar bookkeeping per 4 k-steps");
  run<2>("V2 + __syncthreads per 4 k-steps");
  run<3>("V3 + 8 global loads + 8 LDS stores per 4 k-steps");
  return 0;
}
