// probe_diag_chain.hip -- the per-column chain of the diagonal tile factorization (diag_body.h, wave 0 of diag_llt_body) on
// a lone wave, in variants: with / without the pivot test's branches, the inverse's MFMA, an f32 seed, one Newton step, the
// inverse one column behind, a hand-over of the column through LDS.  Cycles per column (s_memtime); DESIGN.md 9, round 6,
// negative results (5).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probe_diag_chain tools/probe_diag_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int F, int T> struct UF { template <class L> static __device__ __forceinline__ void run(L&& f) { f(std::integral_constant<int, F>{}); UF<F + 1, T>::run(f); } };
template <int T> struct UF<T, T> { template <class L> static __device__ __forceinline__ void run(L&&) {} };

template <int VAR>
__global__ __launch_bounds__(64) void k_chain(const double* __restrict__ A, double* __restrict__ out, long long* cyc, int reps, double critere, int nb) {
  const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4;
  const double cmin = fmax(critere, 2.2250738585072014e-308);
  __shared__ double Xs[400];
  d4 acc = {0, 0, 0, 0};
  int npiv = 0; bool bad = false;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    d4 S, V;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      S[q] = (g + 4 * q <= l15) ? -A[(g + 4 * q) * 16 + l15 + (r & 1) * 256] : 0.0;
      V[q] = (g + 4 * q == l15) ? -1.0 : 0.0;
    }
    double yprev = 0.0, xprev = 0.0;
    UF<0, 16>::run([&](auto J) {
      constexpr int j = decltype(J)::value, qj = j >> 2, gj = j & 3;
      if (VAR == 9 && j >= nb) return;
      const bool ing = (g == gj);
      double d = -readlane_f64(S[qj], j + 16 * gj);
      double y;
      if constexpr (VAR == 2 || VAR == 5) {
        // f32 seed; d within f32's range on the short path
        if (__builtin_expect(!(d >= fmax(cmin, 1e-30) && d <= 1e30), 0)) {
          if (fabs(d) < critere) { d = critere; npiv++; }
          if (!(d > 0.0)) bad = true;
          y = __builtin_amdgcn_rsq(d);
          S[qj] = (ing && l15 == j) ? -d : S[qj];
          y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
        } else {
          y = (double)__builtin_amdgcn_rsqf((float)d);
          y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
        }
        y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
      } else {
        y = __builtin_amdgcn_rsq(d);
        if constexpr (VAR != 1 && VAR < 6) {
          if (__builtin_expect(!(d >= cmin), 0)) {
            if (fabs(d) < critere) { d = critere; npiv++; }
            if (!(d > 0.0)) bad = true;
            y = __builtin_amdgcn_rsq(d);
            S[qj] = (ing && l15 == j) ? -d : S[qj];
          }
        }
        y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
        if constexpr (VAR != 3) y = __builtin_fma(0.5 * y, __builtin_fma(-d * y, y, 1.0), y);
      }
      const double sm = S[qj] * y;
      const double xm = (ing && l15 > j) ? sm : 0.0;
      S[qj] = (ing && l15 >= j) ? sm : S[qj];
      if (j < 15) S = __builtin_amdgcn_mfma_f64_16x16x4f64(xm, xm, S, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (VAR == 6) {
        if constexpr (j > 0) {
          constexpr int jp = j - 1, qp = jp >> 2, gp = jp & 3;
          const bool ingp = (g == gp);
          const double vm = ingp ? V[qp] * yprev : 0.0;
          V[qp] = ingp ? vm : V[qp];
          V = __builtin_amdgcn_mfma_f64_16x16x4f64(xprev, vm, V, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        yprev = y; xprev = xm;
        if constexpr (j == 15) { V[qj] = ing ? V[qj] * y : V[qj]; }
      } else if constexpr (VAR >= 8) {
        const double hand = (l15 == j) ? y : xm;
        const unsigned ha = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)(ing ? &Xs[j * 16 + l15] : &Xs[256 + lane]);
        const unsigned pa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&Xs[330];
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" :: "v"(ha), "v"(hand), "v"(pa), "v"(j + 1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (VAR != 4 && VAR != 5 && VAR != 7) {
        const double vm = ing ? V[qj] * y : 0.0;
        V[qj] = ing ? vm : V[qj];
        if (j < 15) V = __builtin_amdgcn_mfma_f64_16x16x4f64(xm, vm, V, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    acc += S + V;
  }
  long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int q = 0; q < 4; q++) out[lane * 4 + q] = acc[q];
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = npiv + bad; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int VAR> void run(const double* dA, double* dout, long long* dc, const char* name) {
  const int reps = 2000;
  for (int i = 0; i < 2; i++) {
    hipLaunchKernelGGL(k_chain<VAR>, dim3(1), dim3(64), 0, 0, dA, dout, dc, reps, 1e-30, 16);
    CK(hipDeviceSynchronize());
  }
  long long c[2]; CK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
  double o[256]; CK(hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < 256; i++) s += o[i];
  printf("%-44s %7.1f ticks per column  (sum %.12g flag %lld)\n", name, (double)c[0] / (reps * 16.0), s, c[1]);
}
int main() {
  double h[512];
  for (int b = 0; b < 2; b++) for (int c = 0; c < 16; c++) for (int r = 0; r < 16; r++) h[b * 256 + c * 16 + r] = (r == c) ? 40.0 + b : 1.0 / (1 + abs(r - c));
  double *dA, *dout; long long* dc;
  CK(hipMalloc(&dA, sizeof(h))); CK(hipMalloc(&dout, 2048)); CK(hipMalloc(&dc, 16));
  CK(hipMemcpy(dA, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>(dA, dout, dc, "0 current");
  run<1>(dA, dout, dc, "1 no pivot check");
  run<2>(dA, dout, dc, "2 f32 seed, 2 newton");
  run<3>(dA, dout, dc, "3 one newton step");
  run<4>(dA, dout, dc, "4 current without the inverse");
  run<5>(dA, dout, dc, "5 f32 seed without the inverse");
  run<6>(dA, dout, dc, "6 no check, inverse one column behind");
  run<7>(dA, dout, dc, "7 no check, no inverse");
  run<8>(dA, dout, dc, "8 = 7 + hand-over through LDS");
  run<9>(dA, dout, dc, "9 = 8 + nb branch");
  return 0;
}
