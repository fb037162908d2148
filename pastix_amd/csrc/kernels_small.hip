// kernels_small.hip -- gfx950: the update kernel for QUADRANT tasks.
//
// k_update (kernels.hip) gives a 128x128 target tile to eight waves and is built for whole-tile pieces: its chunk
// iteration costs ~1.7 us whatever the piece covers, and 128 VGPRs / 73.7 KB of LDS allow two workgroups per CU.  The
// chains inside the leaf domains of a nested dissection produce the opposite workload: tens of thousands of tasks per
// launch, each a handful of pieces of 10-40 rows and columns with K of 30-60 (100^3: the first 19 launches take 17 % of
// the time for 7 % of the flops).  The plan (plan.cpp, "quadrant tasks") cuts such a task into the four 64x64
// quadrants of its tile, clips the pieces to them, and marks the resulting tasks (Task flag 32); they are ordinary
// tasks on a tile of valid extent tm, tn <= 64 -- k_update runs them correctly too (PASTIX_AMD_SMALL_KERNEL=0) -- and
// this kernel runs them with four waves, 20 KB of LDS per workgroup (2 buffers x (A image + B image) x 16 k-lines x 80
// doubles), < 64 VGPRs, i.e. eight workgroups per CU: four times the tasks in flight, a chunk iteration of
// ~100 instructions.  Same arithmetic as k_update (compute_contrib_compact + add_contrib_local,
// sopalin_compute.c:270-374, :391-598): the pieces of a task are accumulated in MFMA registers in list order and
// subtracted from the tile once; one workgroup owns the quadrant, so the result does not depend on timing.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "plan.h"
#include "devmath.h"

namespace pastix_amd {

typedef double d4s __attribute__((ext_vector_type(4)));

namespace {
constexpr int QK = 16;        // k-lines per chunk
constexpr int QLD = 80;       // LDS line: 64 rows + 16 pad (lanes 16-31 of a ds_read_b64 on the other 32 banks)
}

// Wave (wr, wc) of the 2 x 2 wave grid owns the 16x16 sub-tiles (wr + 2 mi, wc + 2 ni), mi, ni = 0, 1 (cyclic, so that
// a small piece still spreads over the waves).  MFMA operand order as in k_update: the target column is the MFMA "i"
// index, the target row the "j" index, so accumulator register q of lane (l15, g) is C[row = .. + l15][col = .. + g + 4q].
template <int KIND>
__global__ __launch_bounds__(256, 8) void k_update_small(const Arenas ar, const Task* __restrict__ tasks,
                                                      const Piece* __restrict__ pieces) {
  __shared__ double sh[2][QK * QLD];             // [A | B] image of one chunk: 20 KB
  if (KIND == 1) PANEL_PRIO();
  const Task tk = tasks[blockIdx.x];
  if (tk.pn <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  d4s acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++) acc[mi][ni] = d4s{0, 0, 0, 0};
  unsigned touched = 0;                          // bit mi * 2 + ni: some piece reached the sub-tile

  // loader: thread -> (row = lane, k-line = wave + 4 j), j = 0..3, of the A image and of the B image
  const int pend = tk.p0 + tk.pn;
  int lp = tk.p0;                                // piece the loader is in
  Piece cur = pieces[lp];
  int kdone = 0;                                 // k-lines of `cur` already fetched
  double ra[4], rb[4];
  unsigned maskn = 0;                            // sub-tiles of this wave the chunk in flight touches
  auto band_mask = [&](const Piece& pc) {
    unsigned m = 0;
    const int re = (int)pc.dr + (int)pc.m, ce = (int)pc.dc + (int)pc.n;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
      for (int ni = 0; ni < 2; ni++) {
        const int r0 = (wr + 2 * mi) * 16, c0 = (wc + 2 * ni) * 16;
        if (r0 < re && r0 + 16 > (int)pc.dr && c0 < ce && c0 + 16 > (int)pc.dc) m |= 1u << (mi * 2 + ni);
      }
    return m;
  };
  auto fetch = [&]() {                           // next chunk of the piece list -> registers; false when the list is done
    if (lp >= pend) return false;
    const double* A = ar.p[cur.flags & 3] + cur.a_off;
    const double* B = ar.p[(cur.flags >> 2) & 3] + cur.b_off;
    const bool neg = (cur.flags & 16) != 0;
    const bool ina = lane >= (int)cur.dr && lane < (int)cur.dr + (int)cur.m;
    const bool inb = lane >= (int)cur.dc && lane < (int)cur.dc + (int)cur.n;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int kl = kdone + wave + 4 * j;
      const bool kv = kl < (int)cur.k;
      double a = (kv && ina) ? A[(lane - (int)cur.dr) + (int64_t)kl * cur.lda] : 0.0;
      ra[j] = neg ? -a : a;
      rb[j] = (kv && inb) ? B[(lane - (int)cur.dc) + (int64_t)kl * cur.lda] : 0.0;
    }
    maskn = band_mask(cur);
    kdone += QK;
    if (kdone >= (int)cur.k) {
      kdone = 0;
      if (++lp < pend) cur = pieces[lp];
    }
    return true;
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      sh[0][(wave + 4 * j) * QLD + lane] = ra[j];
      sh[1][(wave + 4 * j) * QLD + lane] = rb[j];
    }
  };

  // One LDS image, the next chunk waits in registers: its global loads are issued right after the barrier that
  // publishes the current image and fly under this chunk's MFMAs (and under the other seven workgroups of the CU).
  bool more = fetch();
  const double* sA = sh[0] + wr * 16 + l15 + g * QLD;
  const double* sB = sh[1] + wc * 16 + l15 + g * QLD;
  while (more) {
    stash();
    const unsigned maskc = maskn;
    __syncthreads();
    more = fetch();
    touched |= maskc;
    if (maskc) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        double bm[2], an[2];
#pragma unroll
        for (int s = 0; s < 2; s++) { bm[s] = sA[ks * 4 * QLD + s * 32]; an[s] = sB[ks * 4 * QLD + s * 32]; }
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
          for (int ni = 0; ni < 2; ni++)
            if (maskc & (1u << (mi * 2 + ni)))
              acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[ni], bm[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();                             // every wave has read the image before the next one is stored
  }

  // C -= acc on the sub-tiles some piece reached
  double* C = ar.p[tk.flags & 3] + tk.c_off;
#pragma unroll
  for (int mi = 0; mi < 2; mi++)
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      if (!(touched & (1u << (mi * 2 + ni)))) continue;
      const int r = (wr + 2 * mi) * 16 + l15;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = (wc + 2 * ni) * 16 + g + 4 * q;
        if (r < (int)tk.tm && c < (int)tk.tn) C[r + (int64_t)c * tk.ldc] -= acc[mi][ni][q];
      }
    }
}

void launch_update_small(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks,
                         bool urgent) {
  if (ntasks <= 0) return;
  const dim3 g((unsigned)ntasks);
  if (urgent) hipLaunchKernelGGL((k_update_small<1>), g, dim3(256), 0, s, ar, tasks, pieces);
  else hipLaunchKernelGGL((k_update_small<0>), g, dim3(256), 0, s, ar, tasks, pieces);
}

}  // namespace pastix_amd
