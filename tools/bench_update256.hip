// Experiment (timing + checksum only): the DMA loop of k_update on a 256 x 128 target tile -- 8 waves, each
// 4 row bands x 4 col bands (128 accumulator VGPRs, 2 waves per SIMD, 1 workgroup per CU), A image 256 rows,
// B image 128 rows: 25 % fewer operand bytes per flop, half the barriers per flop, 0.5 LDS reads per MFMA.
// Whole-tile pieces only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bench_update256 tools/bench_update256.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define GLDS(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
struct Piece { int64_t a_off, b_off; int32_t lda; int32_t k; };
struct Task { int64_t c_off; int32_t ldc, p0, pn; };
constexpr int KC = 16, SLA = 272, SLB = 144;   // LDS line lengths (doubles): 256 + 16, 128 + 16

template <int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void k_upd256(const double* __restrict__ src, double* __restrict__ dst,
                                                    const Task* __restrict__ tasks, const Piece* __restrict__ pieces) {
  __shared__ double sA[2][KC * SLA];
  __shared__ double sB[2][KC * SLB];
  const Task tk = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, g = lane >> 4;
  // NWV = 8: row bands wr + 4 s (16 bands, MI = 4); NWV = 16: wr + 8 s (MI = 2, 64 accumulator VGPRs, four waves per SIMD
  // at one workgroup per CU); col bands wc + 2 s (8 bands)
  constexpr int MI = 32 / NWV, NI = 4, RS = 8 * NWV, CS = 32;
  constexpr int LW = 16 / NWV;                         // k-lines per wave and operand
  const int row0 = wr * 16, col0 = wc * 16;
  d4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) acc[mi][ni] = d4{0, 0, 0, 0};
  int pi = tk.p0;
  const int pend = tk.p0 + tk.pn;
  Piece cur = pieces[pi];
  Piece nextp = pieces[min(pi + 1, pend - 1)];
  int64_t lda = cur.lda;
  // A: 16 k-lines x 2 KiB = 32 wave-instructions; B: 16 x 1 KiB = 16: wave w copies A lines 2w, 2w+1 (both halves)
  // and B lines 2w, 2w+1
  const double* pa = src + cur.a_off + (int64_t)(LW * wave) * lda + 2 * lane;
  const double* pb = src + cur.b_off + (int64_t)(LW * wave) * lda + 2 * lane;
  int left = cur.k / KC;
  auto dma = [&](int b) {
#pragma unroll
    for (int q = 0; q < LW; q++) {
      GLDS(pa + (int64_t)q * lda, sA[b] + (LW * wave + q) * SLA);
      GLDS(pa + (int64_t)q * lda + 128, sA[b] + (LW * wave + q) * SLA + 128);
      GLDS(pb + (int64_t)q * lda, sB[b] + (LW * wave + q) * SLB);
    }
  };
  dma(0);
  const double* sAw = sA[0] + row0 + l15 + g * SLA;
  const double* sBw = sB[0] + col0 + l15 + g * SLB;
  double bm0[MI], an0[NI], bm1[MI], an1[NI];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < MI; s++) bm0[s] = sAw[s * RS];
#pragma unroll
  for (int s = 0; s < NI; s++) an0[s] = sBw[s * CS];
  int buf = 0;
#define MFMA16(bm, an)                                                                            \
  _Pragma("unroll") for (int mi = 0; mi < MI; mi++)                                               \
  _Pragma("unroll") for (int ni = 0; ni < NI; ni++)                                               \
      acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[ni], bm[mi], acc[mi][ni], 0, 0, 0);
  while (true) {
    bool has_next = true;
    if (--left == 0) {
      if (++pi < pend) {
        cur = nextp;
        lda = cur.lda;
        pa = src + cur.a_off + (int64_t)(LW * wave) * lda + 2 * lane;
        pb = src + cur.b_off + (int64_t)(LW * wave) * lda + 2 * lane;
        left = cur.k / KC;
        nextp = pieces[min(pi + 1, pend - 1)];
      } else has_next = false;
    } else { pa += (int64_t)KC * lda; pb += (int64_t)KC * lda; }
    if (has_next) dma(buf ^ 1);
    const double* a_ = sAw + buf * (KC * SLA);
    const double* b_ = sBw + buf * (KC * SLB);
#pragma unroll
    for (int s = 0; s < MI; s++) bm1[s] = a_[4 * SLA + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an1[s] = b_[4 * SLB + s * CS];
    MFMA16(bm0, an0)
#pragma unroll
    for (int s = 0; s < MI; s++) bm0[s] = a_[8 * SLA + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an0[s] = b_[8 * SLB + s * CS];
    MFMA16(bm1, an1)
#pragma unroll
    for (int s = 0; s < MI; s++) bm1[s] = a_[12 * SLA + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an1[s] = b_[12 * SLB + s * CS];
    MFMA16(bm0, an0)
    __syncthreads();
    {
      const double* na = sAw + (buf ^ 1) * (KC * SLA);
      const double* nb = sBw + (buf ^ 1) * (KC * SLB);
#pragma unroll
      for (int s = 0; s < MI; s++) bm0[s] = na[s * RS];
#pragma unroll
      for (int s = 0; s < NI; s++) an0[s] = nb[s * CS];
    }
    __builtin_amdgcn_sched_barrier(0);
    MFMA16(bm1, an1)
    if (!has_next) break;
    buf ^= 1;
  }
  double* C = dst + tk.c_off;
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    const int r = row0 + mi * RS + l15;
    double cv[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int q = 0; q < 4; q++) cv[ni][q] = C[r + (int64_t)(col0 + ni * CS + g + 4 * q) * tk.ldc];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int q = 0; q < 4; q++) C[r + (int64_t)(col0 + ni * CS + g + 4 * q) * tk.ldc] = cv[ni][q] - acc[mi][ni][q];
  }
}

int main(int argc, char** argv) {
  int ntask = argc > 1 ? atoi(argv[1]) : 2048, P = argc > 2 ? atoi(argv[2]) : 16, K = argc > 3 ? atoi(argv[3]) : 128;
  int pool = argc > 4 ? atoi(argv[4]) : 64;
  int rows = 4096;
  int64_t src_elems = (int64_t)pool * rows * K, c_elems = (int64_t)ntask * 256 * 128;
  double *ds, *dc; CK(hipMalloc(&ds, src_elems * 8)); CK(hipMalloc(&dc, c_elems * 8));
  {
    std::vector<double> h((size_t)std::min<int64_t>(src_elems, (int64_t)1 << 26));
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> u(-1, 1);
    for (auto& v : h) v = u(rng);
    for (int64_t o = 0; o < src_elems; o += (int64_t)h.size())
      CK(hipMemcpy(ds + o, h.data(), std::min<int64_t>((int64_t)h.size(), src_elems - o) * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dc, 0, c_elems * 8));
  }
  std::vector<Task> tasks(ntask); std::vector<Piece> pieces((size_t)ntask * P);
  for (int t = 0; t < ntask; t++) {
    tasks[t] = Task{(int64_t)t * 256 * 128, 256, t * P, P};
    for (int p = 0; p < P; p++) {
      int s = (int)(((int64_t)t * 7 + p * 13) % pool); int ra = ((t * 31 + p) % (rows / 256)) * 256, rb = ((t * 17 + 3 * p) % (rows / 128)) * 128;
      pieces[(size_t)t * P + p] = Piece{(int64_t)s * rows * K + ra, (int64_t)s * rows * K + rb, rows, K};
    }
  }
  Task* dt; Piece* dp; CK(hipMalloc(&dt, tasks.size() * sizeof(Task))); CK(hipMalloc(&dp, pieces.size() * sizeof(Piece)));
  CK(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nwv = getenv("NWV") ? atoi(getenv("NWV")) : 8;
  auto launch = [&]() {
    if (nwv == 16) hipLaunchKernelGGL(k_upd256<16>, dim3(ntask), dim3(1024), 0, 0, ds, dc, dt, dp);
    else hipLaunchKernelGGL(k_upd256<8>, dim3(ntask), dim3(512), 0, 0, ds, dc, dt, dp);
  };
  launch(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  // spot check of one element against a host dot product (one launch: C = -sum_p A_p B_p^T)
  {
    std::vector<double> ha((size_t)P * K), hb((size_t)P * K); double c;
    const int t = 3 % ntask, r = 77, col = 41;
    for (int p = 0; p < P; p++)
      for (int k = 0; k < K; k++) {
        CK(hipMemcpy(&ha[(size_t)p * K + k], ds + pieces[(size_t)t * P + p].a_off + r + (int64_t)k * rows, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hb[(size_t)p * K + k], ds + pieces[(size_t)t * P + p].b_off + col + (int64_t)k * rows, 8, hipMemcpyDeviceToHost));
      }
    double ref = 0; for (size_t i = 0; i < ha.size(); i++) ref -= ha[i] * hb[i];
    CK(hipMemcpy(&c, dc + tasks[t].c_off + r + (int64_t)col * 256, 8, hipMemcpyDeviceToHost));
    printf("check C[%d,%d] of task %d: %.12e vs %.12e\n", r, col, t, c, ref);
  }
  int reps = getenv("REPS") ? atoi(getenv("REPS")) : 5; CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = 2.0 * 256 * 128 * K * (double)P * ntask * reps;
  printf("256x128 tiles, %d waves: tasks=%d pieces/task=%d K=%d pool=%d: %.3f ms/launch, %.1f TFLOP/s (%.1f%% of 78.6)\n", nwv, ntask, P, K, pool, ms / reps,
         fl / (ms * 1e-3) * 1e-12, fl / (ms * 1e-3) / 78.6e12 * 100);
  return 0;
}
