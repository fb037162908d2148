// kernels_f32.hip -- gfx950: the single-precision factorization kernels (the reference's S_ build: the whole sopalin
// path compiled with PASTIX_FLOAT = float, src/common/src/redefine_functions.h:42-98; its GPU precedent has dedicated
// single-precision tiles, src/sopalin/src/gemm_stencil.h:231-276).
//
//   k_update_s : compute_contrib_compact + add_contrib_local (sopalin_compute.c:270-374, :391-598) on
//                v_mfma_f32_32x32x2_f32 -- the fp32 matrix pipe runs at twice the fp64 rate (157 against 78.6 TFLOP/s) and
//                a panel entry is half the bytes.  Same plan, same tasks and pieces, same tile ownership (one workgroup
//                owns a 128 x 128 target tile and subtracts its accumulated pieces once: deterministic), so the plan,
//                the driver and the schedule are shared with the fp64 engine; only the arenas hold floats.
//   k_diag_s   : factor_diag (compute_diag.c:538-605) for LLt / LDLt / LU with the static-pivot clamp and count
//                (:133-137, :439-468), the diagonal blok resident in LDS;
//   k_trsm_s   : factor_trsm1d / kernel_trsm (compute_trsm.c:58-171), one panel row per thread against the diagonal
//                blok in LDS, 16-column blocks in registers.
// The panel kernels are plain (no MFMA): < 3 % of the flops; the update kernel stages its operands through registers
// (coalesced 4-byte loads, masks for partial pieces: any alignment, any rectangle) instead of the LDS-DMA of the fp64
// kernel -- a 16-byte DMA lane would carry four floats across a piece boundary.
#include <hip/hip_runtime.h>

#include "plan.h"
#include "devmath.h"

namespace pastix_amd {

typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // a 16-byte load from a 4-byte aligned address

namespace {
constexpr int SKC = 32;            // k-lines per chunk
constexpr int SLDF = 144;          // LDS line: 128 rows + 16 pad floats (two buffers of 2 x 32 lines = 73.7 KB: two
                                   // workgroups per CU; the two k-lines a 64-lane read touches overlap in 16 banks)
__device__ __forceinline__ float* arena_f(const Arenas& ar, int a) { return reinterpret_cast<float*>(ar.p[a]); }
}  // namespace

// MFMA f32 32x32x2 lane maps: A operand: lane l holds A[i = l & 31][k = l >> 5]; B operand: B[k = l >> 5][j = l & 31];
// D: lane l, register q holds D[i = 8 (q >> 2) + 4 (l >> 5) + (q & 3)][j = l & 31].  As in the fp64 kernel the target
// COLUMN is fed as "i" and the target ROW as "j": every accumulator register is 32 consecutive rows of one column.
// Workgroup: 512 threads = 8 waves, wave (wr, wc) owns rows [32 wr, +32) x columns [64 wc, +64) = two 32 x 32 tiles.
template <int KIND>
__global__ __launch_bounds__(512, 4) void k_update_s(const Arenas ar, const Task* __restrict__ tasks,
                                                    const Piece* __restrict__ pieces) {
  __shared__ float sh[2][2][SKC * SLDF];         // [buffer][A|B][k][row]   73,728 bytes
  if (KIND == 1) PANEL_PRIO();
  const Task tk = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // loader: thread -> operand (A | B), four consecutive tile rows / columns 4 lq .. 4 lq + 3, k-lines lk + 8 h, h < 4:
  // four 16-byte loads per thread and chunk (one wave-instruction moves 1 KiB)
  const int lo = tid >> 8, lq = tid & 31, lk = (tid >> 5) & 7;
  f16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
  const int pend = tk.p0 + tk.pn;                // (scalar: the Task came through a scalar load)
  int pi = tk.p0, kdone = 0;                     // piece / k-lines of it already fetched: wave-uniform
  // 32-deep chunks: a 16-deep fp32 chunk is ~1.7 us of matrix work per workgroup pair, about one memory round trip --
  // the fp64 kernel's one chunk of lead would leave it exposed, and a barrier per 16 k-lines costs twice what it costs
  // there.
  constexpr int NH = SKC / 8;
  f4 stA[NH];
  int actn = 0;                                  // tiles of this wave the chunk just fetched touches (bit t)
  // next chunk of the piece list -> registers.  Every load is unconditional, from an address clamped to the piece (a
  // quad that straddles a piece boundary brings up to three neighbouring panel entries along -- inside the arena or
  // its padding --; rows and k-lines outside the piece become zero by a select): no exec-masked branch per load.
  auto fetch = [&](f4 (&st)[NH]) -> bool {
    if (pi >= pend) return false;
    const Piece pc = pieces[__builtin_amdgcn_readfirstlane(pi)];
    const int K = (int)pc.k;
    const float* pa = arena_f(ar, pc.flags & 3) + pc.a_off;                  // (scalar address arithmetic)
    const float* pb = arena_f(ar, (pc.flags >> 2) & 3) + pc.b_off;
    const float* base = lo ? pb : pa;
    const int d0 = lo ? (int)pc.dc : (int)pc.dr, len = lo ? (int)pc.n : (int)pc.m;
    const int r0 = 4 * lq - d0;                                              // piece row of the quad's first element
    const float* src = base + min(max(r0, -3), len - 1);
    const int sh0 = r0 - min(max(r0, -3), len - 1);                          // (0 unless the quad lies wholly outside)
#pragma unroll
    for (int h = 0; h < NH; h++) {
      const int k = kdone + lk + 8 * h;
      const f4u v = *(const f4u*)(src + (int64_t)min(k, K - 1) * pc.lda);
      const bool kv = k < K && sh0 == 0;
      st[h][0] = (kv && r0 >= 0 && r0 < len) ? v.x : 0.f;
      st[h][1] = (kv && r0 + 1 >= 0 && r0 + 1 < len) ? v.y : 0.f;
      st[h][2] = (kv && r0 + 2 >= 0 && r0 + 2 < len) ? v.z : 0.f;
      st[h][3] = (kv && r0 + 3 >= 0 && r0 + 3 < len) ? v.w : 0.f;
    }
    const int re = (int)pc.dr + (int)pc.m, ce = (int)pc.dc + (int)pc.n;
    actn = 0;
    if (32 * wr < re && 32 * wr + 32 > (int)pc.dr) {
      if (64 * wc < ce && 64 * wc + 32 > (int)pc.dc) actn |= 1;
      if (64 * wc + 32 < ce && 64 * wc + 64 > (int)pc.dc) actn |= 2;
    }
    kdone += SKC;
    if (kdone >= K) { kdone = 0; pi++; }
    return true;
  };
  auto stash = [&](int buf, const f4 (&st)[NH]) {
#pragma unroll
    for (int h = 0; h < NH; h++) *(f4*)(sh[buf][lo] + (lk + 8 * h) * SLDF + 4 * lq) = st[h];
  };
  auto compute = [&](int buf, int act) {
    const float* sA = sh[buf][0] + lh * SLDF + 32 * wr + l31;
    const float* sB = sh[buf][1] + lh * SLDF + 64 * wc + l31;
    if (act == 3) {
#pragma unroll
      for (int s = 0; s < SKC / 2; s++) {
        const float bm = sA[2 * s * SLDF];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[2 * s * SLDF], bm, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[2 * s * SLDF + 32], bm, acc[1], 0, 0, 0);
      }
    } else if (act == 1) {
#pragma unroll
      for (int s = 0; s < SKC / 2; s++)
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[2 * s * SLDF], sA[2 * s * SLDF], acc[0], 0, 0, 0);
    } else if (act == 2) {
#pragma unroll
      for (int s = 0; s < SKC / 2; s++)
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[2 * s * SLDF + 32], sA[2 * s * SLDF], acc[1], 0, 0, 0);
    }
  };
  bool more = fetch(stA);
  int actc = actn;
  if (more) stash(0, stA);
  __syncthreads();
  int buf = 0;
  while (more) {
    const int act = __builtin_amdgcn_readfirstlane(actc);
    more = fetch(stA);                           // global loads of the next chunk fly under this chunk's MFMAs
    const int nact = actn;
    compute(buf, act);
    if (more) stash(buf ^ 1, stA);
    actc = nact;
    __syncthreads();
    buf ^= 1;
  }
  // C -= acc: the 32 loads of a lane first (clamped addresses), then the stores
  float* C = arena_f(ar, tk.flags & 3) + tk.c_off;
  const int r = 32 * wr + l31;
  const int rc = min(r, (int)tk.tm - 1);
  float cv[2][16];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int c = min(64 * wc + 32 * t + 8 * (q >> 2) + 4 * lh + (q & 3), (int)tk.tn - 1);
      cv[t][q] = C[rc + (int64_t)c * tk.ldc];
    }
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int c = 64 * wc + 32 * t + 8 * (q >> 2) + 4 * lh + (q & 3);
      if (r < (int)tk.tm && c < (int)tk.tn) C[r + (int64_t)c * tk.ldc] = cv[t][q] - acc[t][q];
    }
}

void launch_update_s(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks, bool urgent) {
  if (ntasks <= 0) return;
  const dim3 g((unsigned)ntasks);
  if (urgent) hipLaunchKernelGGL((k_update_s<1>), g, dim3(512), 0, s, ar, tasks, pieces);
  else hipLaunchKernelGGL((k_update_s<0>), g, dim3(512), 0, s, ar, tasks, pieces);
}

// ------------------------------------------------------------------------------------------------
// k_diag_s<FACTO>: 0 LLt (PASTIX_potrf_block, compute_diag.c:124-203), 1 LDLt (PASTIX_sytrf_block :213-307), 2 LU without
// row pivoting + DimTrans (PASTIX_getrf_block :432-532).  One workgroup per cblk (w <= 128), the blok in dynamic LDS,
// 16-column block steps with three barriers each:
//   (A) wave 0 factorizes the 16 x 16 tile in registers -- lane i holds row i, pivots and multipliers travel by
//       shuffles, the static-pivot clamp and count happen here (compute_diag.c:133-137, :439-443);
//   (B) one thread per row below the tile solves it against the tile (LU: and one thread per column right of it);
//   (C) all threads update the trailing square in 4 x 4 register tiles (LLt / LDLt: its lower part).
// ------------------------------------------------------------------------------------------------
template <int FACTO>
__global__ __launch_bounds__(256) void k_diag_s(float* __restrict__ L, float* __restrict__ U,
                                                const PanelTask* __restrict__ tasks, float critere,
                                                long long* __restrict__ nbpivot, int* __restrict__ errflag) {
  PANEL_PRIO();
  extern __shared__ float S[];                   // the blok [r + c * ldl]; LDLt: then Y[16][ldy] = (L D) of the block step
  const PanelTask tk = tasks[blockIdx.x];
  float* A = L + tk.off;
  const int64_t ld = tk.stride;
  const int w = tk.width, tid = threadIdx.x, ldl = w | 1;
  float* Y = S + (size_t)ldl * w;                // [p * ldy + r], LDLt only
  const int ldy = w | 1;
  for (int id = tid; id < w * w; id += 256) {
    const int r = id % w, c = id / w;
    S[r + c * ldl] = A[r + c * ld];
  }
  __syncthreads();
  int npiv = 0, npos = 0, bad = 0;
  for (int kb = 0; kb < w; kb += 16) {
    const int nb = min(16, w - kb), rem = w - kb - nb;
    // ---- (A) the tile, one wave, lane i = row i
    if (tid < 64) {
      const int i = tid & 15;
      float t[16];
#pragma unroll
      for (int c = 0; c < 16; c++) t[c] = (i < nb && c < nb) ? S[(kb + i) + (kb + c) * ldl] : (i == c ? 1.f : 0.f);
#pragma unroll
      for (int j = 0; j < 16; j++) {
        float d = __shfl(t[j], j, 16);
        if (j < nb) {
          if (fabsf(d) < critere) { d = critere; npiv += (tid == 0); }
          if (FACTO == 0) { d = sqrtf(d); if (!(d == d) || d == 0.f) bad = 1; }
          if (FACTO == 1 && d > 0.f) npos += (tid == 0);
        }
        const float inv = 1.0f / d;
        const float u = t[j];                                  // a_ij before scaling
        if (i == j) t[j] = d;
        else if (i > j) t[j] = u * inv;                        // l_ij
#pragma unroll
        for (int c = 0; c < 16; c++) {
          if (c > j) {
            // LLt: l_cj; LDLt: a_cj (unscaled); LU: u_jc (row j of the tile)
            const float o = FACTO == 2 ? __shfl(t[c], j, 16) : __shfl(FACTO == 1 ? u : t[j], c, 16);
            if (i > j) t[c] -= t[j] * o;
          }
        }
      }
      if (tid < 16 && i < nb) {
#pragma unroll
        for (int c = 0; c < 16; c++)
          if (c < nb && (FACTO == 2 || c <= i)) S[(kb + i) + (kb + c) * ldl] = t[c];
      }
    }
    __syncthreads();
    // ---- (B) rows below the tile (threads 0..rem-1); LU: columns right of it (threads 128..128+rem-1)
    if (tid < rem) {
      const int r = kb + nb + tid;
      float x[16];
#pragma unroll
      for (int c = 0; c < 16; c++) x[c] = c < nb ? S[r + (kb + c) * ldl] : 0.f;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nb) {
          float sum = x[c];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < c) sum -= x[p] * (FACTO == 2 ? S[(kb + p) + (kb + c) * ldl] : S[(kb + c) + (kb + p) * ldl]);
          x[c] = FACTO == 1 ? sum : sum / S[(kb + c) + (kb + c) * ldl];       // LDLt: y = L D, unit tile
        }
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < nb) {
          if (FACTO == 1) {
            Y[c * ldy + r] = x[c];
            S[r + (kb + c) * ldl] = x[c] / S[(kb + c) + (kb + c) * ldl];
          } else {
            S[r + (kb + c) * ldl] = x[c];
          }
        }
      }
    }
    if (FACTO == 2 && tid >= 128 && tid - 128 < rem) {
      const int cc = kb + nb + (tid - 128);
      float y[16];
#pragma unroll
      for (int r = 0; r < 16; r++) y[r] = r < nb ? S[(kb + r) + cc * ldl] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        if (r < nb) {
          float sum = y[r];
#pragma unroll
          for (int p = 0; p < 16; p++)
            if (p < r) sum -= S[(kb + r) + (kb + p) * ldl] * y[p];
          y[r] = sum;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (r < nb) S[(kb + r) + cc * ldl] = y[r];
    }
    __syncthreads();
    // ---- (C) trailing update in 4 x 4 tiles
    if (rem > 0) {
      const int nt = (rem + 3) >> 2, o = kb + nb;
      for (int id = tid; id < nt * nt; id += 256) {
        const int tr = id % nt, tc = id / nt;
        if (FACTO != 2 && tr < tc) continue;
        float c4[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) c4[a][b] = 0.f;
        for (int p = 0; p < nb; p++) {
          float xa[4], xb[4];
#pragma unroll
          for (int a = 0; a < 4; a++) {
            const int ra = min(o + 4 * tr + a, w - 1), rb = min(o + 4 * tc + a, w - 1);
            xa[a] = S[ra + (kb + p) * ldl];                                          // L[i][p]
            xb[a] = FACTO == 2 ? S[(kb + p) + rb * ldl] : FACTO == 1 ? Y[p * ldy + rb] : S[rb + (kb + p) * ldl];
          }
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) c4[a][b] += xa[a] * xb[b];
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int a = 0; a < 4; a++) {
            const int r = o + 4 * tr + a, c = o + 4 * tc + b;
            if (r < w && c < w && (FACTO == 2 || r >= c)) S[r + c * ldl] -= c4[a][b];
          }
      }
    }
    __syncthreads();
  }
  for (int id = tid; id < w * w; id += 256) {
    const int r = id % w, c = id / w;
    if (FACTO == 2 || r >= c) A[r + c * ld] = S[r + c * ldl];
    if (FACTO == 2) (U + tk.off)[c + r * ld] = S[r + c * ldl];      // DimTrans: ucoeftab's blok = the transpose
  }
  if (tid == 0) {
    if (npiv) atomicAdd((unsigned long long*)nbpivot, (unsigned long long)npiv);
    if (FACTO == 1 && npos) atomicAdd((unsigned long long*)nbpivot + 1, (unsigned long long)npos);
  }
  if (tid < 64 && bad) *errflag = 1;
}

// ------------------------------------------------------------------------------------------------
// k_trsm_s<MODE>: x_j = (a_j - sum_{p<j} x_p M[j,p]) s_j for the (at most 64) panel rows of a task, one WAVE per task
// (64 threads, 16 KB of LDS: fits any hole a bulk workgroup leaves).  32-column block steps: the lane keeps the 32
// columns of its row in registers; the strip M[jb .. jb+31][0 .. jb+31] of the diagonal blok is staged in LDS once per
// step (one memory round trip) and read back as broadcasts; the earlier blocks of the row are re-read from the panel.
//   0 LLt        M = L_d (lower), s = 1 / diag            panel in L                 (R,L,T,N compute_trsm.c:67-70)
//   1 LDLt       M = L_d unit;  L*D = x -> U arena, L = x / d_j -> L arena           (:92-113)
//   2 LU, L side M[j,p] = U_d[p,j], s = 1 / U_d[j,j]      panel in L                 (R,U,N,N :62-63)
//   3 LU, U side M = L_d unit                             panel in U                 (R,U,N,U on dU :64-66)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_trsm_s(float* __restrict__ L, float* __restrict__ U,
                                               const TrsmTask* __restrict__ tasks) {
  PANEL_PRIO();
  extern __shared__ float Ms[];                  // Ms[p * 32 + j] = M[jb + j][p], p < the widest cblk of the launch
  const TrsmTask tk = tasks[blockIdx.x];
  const int64_t ld = tk.stride;
  const int w = tk.width, tid = threadIdx.x;
  const float* Ad = L + tk.off;                  // the factored diagonal blok is in the L arena for every mode
  const int row = tk.row0 + min(tid, tk.nrows - 1);
  const bool rv = tid < tk.nrows;
  float* P = (MODE == 3 ? U : L) + tk.off + row;           // the row being solved (in place)
  float* Y = MODE == 1 ? U + tk.off + row : P;             // where the recurrence's x_p are re-read from
  for (int jb = 0; jb < w; jb += 32) {
    const int np = min(jb + 32, w);              // columns 0 .. np-1 of the strip are needed
    // (16 loads in flight per lane and batch: a plain loop would pay one memory round trip per element)
    for (int b0 = 0; b0 < 32 * np; b0 += 64 * 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int idx = min(b0 + 64 * i + tid, 32 * np - 1);
        int j, p;
        if (MODE == 2) { j = idx / np; p = idx - j * np; }                     // U_d[p][jb + j]: contiguous in p
        else { j = idx & 31; p = idx >> 5; }                                   // L_d[jb + j][p]: contiguous in j
        const int jj = min(jb + j, w - 1);
        v[i] = MODE == 2 ? Ad[p + (int64_t)jj * ld] : Ad[jj + (int64_t)p * ld];
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int idx = b0 + 64 * i + tid;
        if (idx < 32 * np) {
          int j, p;
          if (MODE == 2) { j = idx / np; p = idx - j * np; }
          else { j = idx & 31; p = idx >> 5; }
          Ms[p * 32 + j] = jb + j < w ? v[i] : 0.f;
        }
      }
    }
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; j++) x[j] = P[(int64_t)min(jb + j, w - 1) * ld];
    __builtin_amdgcn_wave_barrier();             // (one wave: its LDS writes are ordered in front of its reads)
    for (int pb = 0; pb < jb; pb += 32) {
      float xp[32];
#pragma unroll
      for (int p = 0; p < 32; p++) xp[p] = Y[(int64_t)(pb + p) * ld];
#pragma unroll
      for (int p = 0; p < 32; p++) {
        const f4* Mp = (const f4*)(Ms + (pb + p) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const f4 m = Mp[q];
          x[4 * q + 0] -= xp[p] * m[0];
          x[4 * q + 1] -= xp[p] * m[1];
          x[4 * q + 2] -= xp[p] * m[2];
          x[4 * q + 3] -= xp[p] * m[3];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 32; j++) {
#pragma unroll
      for (int p = 0; p < 32; p++)
        if (p < j) x[j] -= x[p] * Ms[(jb + p) * 32 + j];
      if (MODE == 0 || MODE == 2) x[j] /= Ms[min(jb + j, w - 1) * 32 + min(j, w - 1 - jb)];
    }
    if (rv) {
#pragma unroll
      for (int j = 0; j < 32; j++) {
        if (jb + j < w) {
          if (MODE == 1) {
            Y[(int64_t)(jb + j) * ld] = x[j];                                          // L*D  (compute_trsm.c:108-109)
            P[(int64_t)(jb + j) * ld] = x[j] / Ms[(jb + j) * 32 + j];                  // L    (:110)
          } else {
            P[(int64_t)(jb + j) * ld] = x[j];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();             // (the strip is rewritten by the next step)
  }
}

static bool lds_attr(const void* fn, int bytes) {
  // (the attribute is the MAXIMUM a launch may ask for: always the 128-column size; per-device, set on every call)
  (void)bytes;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (128 * 129 + 16 * 129) * (int)sizeof(float)) == hipSuccess;
}

// maxw: the widest cblk of the launch (LDS is sized for it: the narrow cblks of the leaf levels then fit many per CU)
void launch_diag_s(hipStream_t s, int factotype, float* L, float* U, const PanelTask* tasks, int64_t n, double critere,
                   long long* nbpivot, int* errflag, int maxw) {
  if (n <= 0) return;
  const int bytes = (maxw * (maxw | 1) + 16 * (maxw | 1)) * (int)sizeof(float);
  const dim3 g((unsigned)n), b(256);
  if (factotype == PASTIX_AMD_FACT_LLT) {
    if (!lds_attr((const void*)k_diag_s<0>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<0>, g, b, bytes, s, L, U, tasks, (float)critere, nbpivot, errflag);
  } else if (factotype == PASTIX_AMD_FACT_LU) {
    if (!lds_attr((const void*)k_diag_s<2>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<2>, g, b, bytes, s, L, U, tasks, (float)critere, nbpivot, errflag);
  } else {
    if (!lds_attr((const void*)k_diag_s<1>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<1>, g, b, bytes, s, L, U, tasks, (float)critere, nbpivot, errflag);
  }
}

void launch_trsm_s(hipStream_t s, int factotype, float* L, float* U, const TrsmTask* tasks, int64_t n, int maxw) {
  if (n <= 0) return;
  const dim3 g((unsigned)n), b(64);
  const int bytes = 32 * ((maxw + 31) & ~31) * (int)sizeof(float);      // <= 16 KB: the narrow cblks of the leaf levels fit many per CU
  if (factotype == PASTIX_AMD_FACT_LLT) {
    hipLaunchKernelGGL(k_trsm_s<0>, g, b, bytes, s, L, U, tasks);
  } else if (factotype == PASTIX_AMD_FACT_LU) {
    hipLaunchKernelGGL(k_trsm_s<2>, g, b, bytes, s, L, U, tasks);
    hipLaunchKernelGGL(k_trsm_s<3>, g, b, bytes, s, L, U, tasks);
  } else {
    hipLaunchKernelGGL(k_trsm_s<1>, g, b, bytes, s, L, U, tasks);
  }
}

// fill kernels on float arenas
__global__ void k_fill_const_s(float* __restrict__ dst, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = v;
}
__global__ void k_scatter_s(float* __restrict__ dst, const int64_t* __restrict__ idx, const double* __restrict__ val, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[idx[i]] = (float)val[i];
}
void launch_fill_const_s(hipStream_t s, float* dst, int64_t n, float v) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_const_s, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 16384)), dim3(256), 0, s, dst, n, v);
}
void launch_scatter_s(hipStream_t s, float* dst, const int64_t* idx, const double* val, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scatter_s, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, dst, idx, val, n);
}

}  // namespace pastix_amd
