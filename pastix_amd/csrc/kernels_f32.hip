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
//   k_trsm_s   : factor_trsm1d / kernel_trsm (compute_trsm.c:58-171): X^T in fp32 MFMA accumulators, 16 x 16 tile inverses
//                from k_diag_s (the fp64 kernels' organisation).
// k_diag_s is plain (no MFMA; < 1 % of the flops); the update kernel stages its operands through registers
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

// 512 B of zeros: the DMA source of k-lines beyond a piece's K and of tile rows / columns outside a partial piece
__device__ float g_zero_line_s[128];

// LDS-DMA, 4 bytes per lane: one wave-instruction copies 64 consecutive floats of a k-line straight into LDS (no staging
// registers, no ds_write).  Four-byte granularity: any alignment, any piece boundary -- the 16-byte form of the fp64 kernel
// would carry four floats across it.
#define PASTIX_AMD_GLDS4(gptr, lptr)                                                             \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),        \
                                   (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// MFMA f32 32x32x2 lane maps: A operand: lane l holds A[i = l & 31][k = l >> 5]; B operand: B[k = l >> 5][j = l & 31];
// D: lane l, register q holds D[i = 8 (q >> 2) + 4 (l >> 5) + (q & 3)][j = l & 31].  As in the fp64 kernel the target
// COLUMN is fed as "i" and the target ROW as "j": every accumulator register is 32 consecutive rows of one column.
// Workgroup: 512 threads = 8 waves, wave (wr, wc) owns rows [32 wr, +32) x columns [64 wc, +64) = two 32 x 32 tiles.
// Staging (round 3, second version): the fp64 kernel's loop -- the operands of chunk i+1 travel by LDS-DMA while chunk i
// is multiplied, one barrier per chunk -- with 32-deep chunks and 4-byte DMA lanes; wave w copies the k-lines w, w+8, w+16,
// w+24 of both operands, two instructions per line (rows 0-63, 64-127).  Per piece a lane keeps, for each of the four
// (operand, half) pairs, a source pointer and a line stride: inside the piece the panel entry and lda, outside it the
// zero line and 0 -- one loop for whole and partial pieces.  The first version staged through registers (16-byte loads,
// selects, ds_write) and kept the matrix pipe busy 65 % of the time on the big launches of 200^3 (PMC), the fp64 kernel 86 %.
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
  f16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
  const int pend = tk.p0 + tk.pn;                // (scalar: the Task came through a scalar load)
  int pi = tk.p0, kdone = 0;                     // piece / k-lines of it already issued: wave-uniform
  const float* zl = g_zero_line_s + lane;
  // the lane's sources for the current piece: [operand][half]
  const float* src[2][2];
  int str[2][2];
  int K = 0, actn = 0;
  auto setup = [&]() {
    const Piece pc = pieces[__builtin_amdgcn_readfirstlane(pi)];
    K = (int)pc.k;
    const float* pa = arena_f(ar, pc.flags & 3) + pc.a_off;                  // (scalar address arithmetic)
    const float* pb = arena_f(ar, (pc.flags >> 2) & 3) + pc.b_off;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int ra = 64 * h + lane - (int)pc.dr, rb = 64 * h + lane - (int)pc.dc;
      const bool va = ra >= 0 && ra < (int)pc.m, vb = rb >= 0 && rb < (int)pc.n;
      src[0][h] = va ? pa + ra : zl;
      str[0][h] = va ? pc.lda : 0;
      src[1][h] = vb ? pb + rb : zl;
      str[1][h] = vb ? pc.lda : 0;
    }
    const int re = (int)pc.dr + (int)pc.m, ce = (int)pc.dc + (int)pc.n;
    actn = 0;
    if (32 * wr < re && 32 * wr + 32 > (int)pc.dr) {
      if (64 * wc < ce && 64 * wc + 32 > (int)pc.dc) actn |= 1;
      if (64 * wc + 32 < ce && 64 * wc + 64 > (int)pc.dc) actn |= 2;
    }
  };
  // the next chunk of the piece list -> LDS buffer b (DMA); returns the tiles of this wave it touches
  auto issue = [&](int b) -> int {
    if (kdone == 0) setup();
#pragma unroll
    for (int q = 0; q < SKC / 8; q++) {
      const int line = wave + 8 * q, k = kdone + line;
      float* dA = sh[b][0] + line * SLDF;
      float* dB = sh[b][1] + line * SLDF;
      if (k < K) {                               // (wave-uniform: a scalar branch)
        PASTIX_AMD_GLDS4(src[0][0] + (int64_t)k * str[0][0], dA);
        PASTIX_AMD_GLDS4(src[0][1] + (int64_t)k * str[0][1], dA + 64);
        PASTIX_AMD_GLDS4(src[1][0] + (int64_t)k * str[1][0], dB);
        PASTIX_AMD_GLDS4(src[1][1] + (int64_t)k * str[1][1], dB + 64);
      } else {
        PASTIX_AMD_GLDS4(zl, dA);
        PASTIX_AMD_GLDS4(zl, dA + 64);
        PASTIX_AMD_GLDS4(zl, dB);
        PASTIX_AMD_GLDS4(zl, dB + 64);
      }
    }
    const int a = actn;
    kdone += SKC;
    if (kdone >= K) { kdone = 0; pi++; }
    return a;
  };
  auto compute = [&](int buf, int act) {
    const float* sA = sh[buf][0] + lh * SLDF + 32 * wr + l31;
    const float* sB = sh[buf][1] + lh * SLDF + 64 * wc + l31;
    if (act == 3) {
      // (the compiler's own schedule -- the next k-pair's reads issued behind this pair's MFMAs -- measured faster than
      // operands pinned a quarter of the chunk ahead: 97.8 against 93.7 TFLOP/s at 200^3)
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
  if (pi < pend) {
    int actc = issue(0);
    __syncthreads();                             // (vmcnt(0): chunk 0 has landed)
    int buf = 0;
    while (true) {
      const bool more = pi < pend;
      int nact = 0;
      if (more) nact = issue(buf ^ 1);           // the next chunk's DMA flies under this chunk's MFMAs
      compute(buf, __builtin_amdgcn_readfirstlane(actc));
      __syncthreads();                           // the next chunk has landed, this buffer is fully read
      if (!more) break;
      actc = nact;
      buf ^= 1;
    }
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
// column j of the inverse of the lower-triangular 16 x 16 tile at (kb, kb) of the blok S (TRANS: of the transposed upper
// tile; UNIT: unit diagonal; rows / columns beyond nb: identity), forward substitution in registers; dst[i + 16 j]
template <bool UNIT, bool TRANS>
__device__ __forceinline__ void tile_inverse_col(const float* S, const int ldl, const int kb, const int nb, const int j,
                                                 float* __restrict__ dst) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    float sum = (i == j) ? 1.f : 0.f;
#pragma unroll
    for (int p = 0; p < 16; p++) {
      if (p < i) {
        const int ii = min(kb + i, kb + nb - 1), pp = min(kb + p, kb + nb - 1);
        const float t = TRANS ? S[pp + ii * ldl] : S[ii + pp * ldl];
        sum -= ((i < nb && p < nb) ? t : 0.f) * x[p];
      }
    }
    const float d = (UNIT || i >= nb) ? 1.f : S[(kb + i) + (kb + i) * ldl];
    x[i] = sum / d;
  }
#pragma unroll
  for (int i = 0; i < 16; i++) dst[i + 16 * j] = x[i];
}

template <int FACTO>
__global__ __launch_bounds__(256) void k_diag_s(float* __restrict__ L, float* __restrict__ U,
                                                const PanelTask* __restrict__ tasks, float* __restrict__ dinv_ws,
                                                float critere, long long* __restrict__ nbpivot,
                                                int* __restrict__ errflag) {
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
    // ---- the tile's inverse for the panel solve (k_trsm_s), beside (B) on lanes (B) never uses: lane j = column j.
    // LLt: L_t^-1; LDLt: the unit L_t^-1; LU: (U_t^T)^-1 for the L side, then the unit L_t^-1 for the U side
    if (tid >= 240) {
      float* dst = dinv_ws + tk.dinv_off + (int64_t)(kb >> 4) * 256;
      if (FACTO == 0) tile_inverse_col<false, false>(S, ldl, kb, nb, tid - 240, dst);
      else if (FACTO == 1) tile_inverse_col<true, false>(S, ldl, kb, nb, tid - 240, dst);
      else tile_inverse_col<false, true>(S, ldl, kb, nb, tid - 240, dst);
    }
    if (FACTO == 2 && tid >= 112 && tid < 128)
      tile_inverse_col<true, false>(S, ldl, kb, nb, tid - 112, dinv_ws + tk.dinv_off + (int64_t)(((w + 15) >> 4) + (kb >> 4)) * 256);
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
// k_trsm_s<MODE>: the panel solve X = A T^-T for 64 panel rows per workgroup, one wave per 16 rows, X^T in MFMA
// accumulators -- the organisation of k_trsm_llt / k_trsm_var (kernels.hip, kernels_var.hip) on v_mfma_f32_16x16x4_f32:
//   X^T[ct] = Tinv[ct] (A^T[ct] - sum_{p<ct} T[ct,p] X^T[p]),   Tinv[ct] = the 16 x 16 diagonal-tile inverses of k_diag_s.
// An accumulator tile of the fp32 instruction has ROW 4 g + q in register q of lane group g (the fp64 one: g + 4 q), and a
// k-step takes k = g from lane group g.  For accumulator register q of tile p to BE the B operand of k-step q without any
// lane movement, the instruction's row index m stands for the logical column pi(m) = (m >> 2) + 4 (m & 3) of the tile:
// register q of lane group g is then logical column g + 4 q, exactly the fp64 kernel's picture, and the A operand -- which
// decides what row m means -- supplies the coefficients of logical row pi(l15).
//   0 LLt        T = L_d (lower)                          panel in L                 (R,L,T,N compute_trsm.c:67-70)
//   1 LDLt       T = L_d unit;  L*D = x -> U arena, L = x / d_j -> L arena           (:92-113)
//   2 LU, L side T[j,p] = U_d[p,j]                        panel in L                 (R,U,N,N :62-63)
//   3 LU, U side T = L_d unit                             panel in U                 (R,U,N,U on dU :64-66)
// The first version (a wave per 64 rows, the strip of the blok in LDS, scalar FMAs) took 267 us per level at 100^3 and
// was half of that factorization's time.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_trsm_s(float* __restrict__ L, float* __restrict__ U,
                                                   const TrsmTask* __restrict__ tasks,
                                                   const float* __restrict__ dinv_ws) {
  PANEL_PRIO();
  constexpr int NT = 8;
  const TrsmTask tk = tasks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int l15p = (l15 >> 2) + 4 * (l15 & 3);
  const int ld = tk.stride, w = tk.width;
  const int nbk = (w + 15) >> 4;
  const int rloc = wave * 16 + l15;
  if (wave * 16 >= tk.nrows) return;
  const bool rvalid = rloc < tk.nrows;
  float* X = (MODE == 3 ? U : L) + tk.off + tk.row0;
  float* Xp = X + rloc;
  const float* Xpc = X + min(rloc, tk.nrows - 1);
  const float* Td = L + tk.off;                        // the factored diagonal blok (always in the L arena)
  const float* Ti = dinv_ws + tk.dinv_off + (MODE == 3 ? (int64_t)nbk * 256 : 0);

  f4 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const float v = Xpc[(int64_t)min(col, w - 1) * ld];
      acc[ct][q] = (rvalid && col < w) ? v : 0.f;
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
    if (ct < nbk) {
      const int li = ct * 16 + l15p;
      const int lic = min(li, w - 1);
#pragma unroll
      for (int p = 0; p < ct; p++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int lc = p * 16 + g + 4 * q;
          const float tv = (MODE == 2) ? Td[lc + (int64_t)lic * ld] : Td[lic + (int64_t)lc * ld];
          const float a = (li < w) ? -tv : 0.f;
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[p][q], acc[ct], 0, 0, 0);
        }
      }
      f4 t = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float a = Ti[ct * 256 + l15p + 16 * (g + 4 * q)];
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[ct][q], t, 0, 0, 0);
      }
      acc[ct] = t;
    }
  }
#pragma unroll
  for (int ct = 0; ct < NT; ct++) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      if (MODE == 1) {
        const float dv = Td[min(col, w - 1) * (int64_t)(ld + 1)];
        if (rvalid && col < w) {
          (U + tk.off + tk.row0 + rloc)[(int64_t)col * ld] = acc[ct][q];           // L*D (compute_trsm.c:108-109)
          Xp[(int64_t)col * ld] = acc[ct][q] / dv;                                // L   (:110)
        }
      } else {
        if (rvalid && col < w) Xp[(int64_t)col * ld] = acc[ct][q];
      }
    }
  }
}

static bool lds_attr(const void* fn, int bytes) {
  // (the attribute is the MAXIMUM a launch may ask for: always the 128-column size; per-device, set on every call)
  (void)bytes;
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (128 * 129 + 16 * 129) * (int)sizeof(float)) == hipSuccess;
}

// maxw: the widest cblk of the launch (LDS is sized for it: the narrow cblks of the leaf levels then fit many per CU)
void launch_diag_s(hipStream_t s, int factotype, float* L, float* U, const PanelTask* tasks, int64_t n, float* dinv,
                   double critere, long long* nbpivot, int* errflag, int maxw) {
  if (n <= 0) return;
  const int bytes = (maxw * (maxw | 1) + 16 * (maxw | 1)) * (int)sizeof(float);
  const dim3 g((unsigned)n), b(256);
  if (factotype == PASTIX_AMD_FACT_LLT) {
    if (!lds_attr((const void*)k_diag_s<0>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<0>, g, b, bytes, s, L, U, tasks, dinv, (float)critere, nbpivot, errflag);
  } else if (factotype == PASTIX_AMD_FACT_LU) {
    if (!lds_attr((const void*)k_diag_s<2>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<2>, g, b, bytes, s, L, U, tasks, dinv, (float)critere, nbpivot, errflag);
  } else {
    if (!lds_attr((const void*)k_diag_s<1>, bytes)) return;
    hipLaunchKernelGGL(k_diag_s<1>, g, b, bytes, s, L, U, tasks, dinv, (float)critere, nbpivot, errflag);
  }
}

void launch_trsm_s(hipStream_t s, int factotype, float* L, float* U, const TrsmTask* tasks, int64_t n, const float* dinv,
                   int maxw) {
  (void)maxw;
  if (n <= 0) return;
  const dim3 g((unsigned)n), b(256);
  if (factotype == PASTIX_AMD_FACT_LLT) {
    hipLaunchKernelGGL(k_trsm_s<0>, g, b, 0, s, L, U, tasks, dinv);
  } else if (factotype == PASTIX_AMD_FACT_LU) {
    hipLaunchKernelGGL(k_trsm_s<2>, g, b, 0, s, L, U, tasks, dinv);
    hipLaunchKernelGGL(k_trsm_s<3>, g, b, 0, s, L, U, tasks, dinv);
  } else {
    hipLaunchKernelGGL(k_trsm_s<1>, g, b, 0, s, L, U, tasks, dinv);
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
