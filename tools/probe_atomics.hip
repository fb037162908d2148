#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// every workgroup: lanes of wave 0 count down pseudo-random counters (agent-scope relaxed atomics, as run_dec_ticket),
// other waves stream loads/stores through L2 (plain) to make traffic; some lanes poll neighbouring words with agent-scope loads
__global__ void k(int* cnt, int n, int iters, const double* src, double* dst, long long m, int* ring) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  if (tid < 64) {
    unsigned s = 1234567u * (wg + 1) + tid * 977u;
    for (int it = 0; it < iters; it++) {
      s = s * 1664525u + 1013904223u;
      const int c = (s >> 8) % n;
      const int old = __hip_atomic_fetch_add(cnt + c, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == 1) { const int pos = __hip_atomic_fetch_add(ring, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(ring + 64 + pos, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      if ((it & 15) == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
      (void)__hip_atomic_load(cnt + ((c + 1) % n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    for (long long i = (long long)wg * 448 + (tid - 64); i < m; i += (long long)gridDim.x * 448) dst[i] = src[i] * 1.0001;
  }
}
int main() {
  const int n = 1 << 16, nwg = 512, iters = 4000;
  std::vector<int> exp(n, 0);
  for (int wg = 0; wg < nwg; wg++) for (int t = 0; t < 64; t++) { unsigned s = 1234567u * (wg + 1) + t * 977u; for (int it = 0; it < iters; it++) { s = s * 1664525u + 1013904223u; exp[(s >> 8) % n]++; } }
  std::vector<int> img(n); for (int i = 0; i < n; i++) img[i] = exp[i];      // every counter reaches exactly 0
  int *dimg, *dcnt, *dring; double *a, *b; const long long m = 1ll << 26;
  CK(hipMalloc(&dimg, n * 4)); CK(hipMalloc(&dcnt, n * 4)); CK(hipMalloc(&dring, (n + 64) * 4)); CK(hipMalloc(&a, m * 8)); CK(hipMalloc(&b, m * 8));
  CK(hipMemcpy(dimg, img.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(a, 0, m * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  int bad_total = 0;
  for (int rep = 0; rep < 200; rep++) {
    CK(hipMemcpyAsync(dcnt, dimg, n * 4, hipMemcpyDeviceToDevice, s));
    CK(hipMemsetAsync(dring, 0xff, (n + 64) * 4, s)); CK(hipMemsetAsync(dring, 0, 4, s));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 0, s, dcnt, n, iters, a, b, m, dring);
    std::vector<int> out(n), ring(n + 64);
    CK(hipMemcpyAsync(out.data(), dcnt, n * 4, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(ring.data(), dring, (n + 64) * 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    int bad = 0, nz = 0; for (int i = 0; i < n; i++) { bad += out[i] != 0; nz += exp[i] > 0; }
    int miss = 0; for (int i = 0; i < ring[0]; i++) miss += ring[64 + i] < 0;
    if (bad || ring[0] != nz || miss) { printf("rep %d: %d counters wrong, pushed %d of %d, %d ring slots unfilled\n", rep, bad, ring[0], nz, miss); bad_total++; }
  }
  printf("done: %d bad repetitions of 200\n", bad_total);
  return 0;
}
