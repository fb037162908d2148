// kernels_update.hip -- gfx950 (MI355X / CDNA4): k_update, the trailing-update kernel of the sopalin factorization.
//
// k_update = compute_1dgemm = compute_contrib_compact + add_contrib_local (sopalin_compute.c:865-1032, :270-374,
// :391-598) fused: every workgroup owns one 128x128 tile of a target panel and accumulates all contributions
// ("pieces", plan.h) that the plan scheduled into this launch, then subtracts them from the tile once.  Tile ownership
// replaces mutex_blok[] and makes the result deterministic.  ~96 % of the factorization's time at 200^3.
//
// MFMA f64 16x16x4 lane maps (measured on gfx950, tools/probe_mfma_f64.hip):
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15];
//   C/D: lane l, register q holds D[i = (l>>4) + 4q][j = l&15].
// The target COLUMN index is fed as MFMA "i" and the target ROW index as "j", so that each accumulator register maps to
// 16 consecutive rows of the column-major panel (128-byte segments for the read-modify-write of the tile).
//
// Workgroup: 512 threads = 8 waves in a 4 x 2 grid; wave (wr, wc) owns the 16-row bands wr, wr + 4 and the 16-column
// bands wc, wc + 2, wc + 4, wc + 6 of the tile (cyclic ownership: a piece that covers part of the tile still spreads
// over all waves) = 2 x 4 MFMA sub-tiles.
//
// The ACCUMULATORS LIVE IN AGPRs a[0:63], outside the compiler's register allocation: sub-tile (mi, ni) is
// a[8 (4 mi + ni) : +7], every MFMA is an inline-asm statement naming them.  Reason: pieces that cover only part of
// the tile must skip the sub-tiles they do not touch.  With compiler-allocated accumulators that is an exec-mask guard
// per MFMA (s_and_saveexec / s_cbranch / s_or: >100 scalar instructions per 16-deep chunk, and a SIMD issues one
// scalar instruction per 4 cycles -- a 16x16 piece cost 0.42 of a whole-tile piece), and any attempt to branch ONCE per
// k-step into straight-line code for the wave's pattern of active sub-tiles (a switch: 30 patterns) made the compiler
// spill thousands of VGPRs on the phi copies of eight 256-bit accumulator tuples.  Accumulators the compiler does not
// see have no phis: the switch costs a handful of scalar compares, each arm is exactly the MFMAs the pattern needs.
// 64 AGPRs + < 64 VGPRs = the 128 registers of four waves per SIMD at two workgroups per CU, as before.
// Wait states the compiler cannot insert for code it does not see (cdna_hip_programming.md 5.7) are in the strings:
// accumulate chains need none; the zeroing writes and the epilogue's reads are padded.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "plan.h"
#include "devmath.h"
#include "run_sync.h"
#include "acc_regs.h"
#include "diag_body.h"

namespace pastix_amd {

constexpr int KC = 16;        // k-chunk staged per barrier
constexpr int SLD = 144;      // LDS line length in doubles: 128 rows + 16 pad -> lanes 16-31 of a
                              // ds_read_b64 land on banks 32-63 (conflict-free, MI355X_MICROARCH LDS)
constexpr int UW = 8;         // waves per workgroup
constexpr int MI = 2, NI = 4; // 16-row / 16-col sub-tiles per wave
constexpr int RS = 64, CS = 32;   // distance between a wave's consecutive row / col bands

// The thread's index WITHOUT a register that lives across the kernel: threadIdx.x arrives in v0, and every later use keeps a
// copy of it alive through all the loops -- one of the values the allocator ended up spilling inside the chunk arms.  The
// lane comes from v_mbcnt (asm volatile: recomputed where it is needed, never hoisted), the wave from the scalar side.
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ int tid_now(const int wave_s) { return (wave_s << 6) | lane_now(); }
// (a double zero made in place: as a constant the compiler keeps it in a register pair across the whole kernel -- and spills it)
__device__ __forceinline__ double zero_now() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return __hiloint2double(z, z);
}

// the MFMAs of one k-step for the sub-tiles RM (row bands, MI bits) x CM (col bands, NI bits) of this wave
template <unsigned RM, unsigned CM>
__device__ __forceinline__ void mfma_sel(const double (&an)[NI], const double (&bm)[MI]) {
  if constexpr ((RM & 1u) && (CM & 1u)) acc_mfma<0>(an[0], bm[0]);
  if constexpr ((RM & 1u) && (CM & 2u)) acc_mfma<1>(an[1], bm[0]);
  if constexpr ((RM & 1u) && (CM & 4u)) acc_mfma<2>(an[2], bm[0]);
  if constexpr ((RM & 1u) && (CM & 8u)) acc_mfma<3>(an[3], bm[0]);
  if constexpr ((RM & 2u) && (CM & 1u)) acc_mfma<4>(an[0], bm[1]);
  if constexpr ((RM & 2u) && (CM & 2u)) acc_mfma<5>(an[1], bm[1]);
  if constexpr ((RM & 2u) && (CM & 4u)) acc_mfma<6>(an[2], bm[1]);
  if constexpr ((RM & 2u) && (CM & 8u)) acc_mfma<7>(an[3], bm[1]);
}
// ---- epilogue ------------------------------------------------------------------------------------
// C -= acc for the row band MIX of the wave: loads of one 16-row band are issued together from clamped addresses (one
// latency per band, not per element); ATOMIC: tiles that several workgroups update in the same launch (split piece
// lists of the multi-GPU fan-in schedule) combine with f64 atomics instead of an exclusive read-modify-write.
// COH: the stores are write-through (the run launch: the tile is handed to another workgroup of the same launch).
template <int MIX, bool ATOMIC, bool COH, int NI0>
__device__ __forceinline__ void epilogue_half(double* __restrict__ C, const unsigned touched, const int row0, const int col0,
                                              const int l15, const int g, const int tm1, const int tn1, const int ldc) {
  // the column bands NI0, NI0 + 1 of the row band MIX: eight loads in flight, then the eight stores -- sixteen values of
  // the tile at a time (a whole band) are half of the 64 VGPRs the kernels have, and the run launch's instances spilled
  if (!((touched >> (MI + NI0)) & 3u)) return;
  const int r = row0 + MIX * RS + l15;
  const int rc = min(r, tm1);
  double cv[2][4];
  if (!ATOMIC) {
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = min(col0 + (NI0 + h) * CS + g + 4 * q, tn1);
        cv[h][q] = C[rc + (int64_t)c * ldc];
      }
  }
  // (the accumulator registers are read one at a time, next to their use: 64 VGPRs are all there is beside the AGPRs)
  auto put = [&](auto h_c, auto q_c) {
    constexpr int h = decltype(h_c)::value, ni = NI0 + h, q = decltype(q_c)::value;
    if (!((touched >> (MI + ni)) & 1u)) return;
    const int c = col0 + ni * CS + g + 4 * q;
    if (r <= tm1 && c <= tn1) {
      const double a = acc_read<4 * MIX + ni, q>();
      if (ATOMIC) unsafeAtomicAdd(&C[r + (int64_t)c * ldc], -a);
      else pst<COH>(&C[r + (int64_t)c * ldc], cv[h][q] - a);
    }
  };
#define PA_PUT4(h)                                                              \
  put(std::integral_constant<int, h>{}, std::integral_constant<int, 0>{});      \
  put(std::integral_constant<int, h>{}, std::integral_constant<int, 1>{});      \
  put(std::integral_constant<int, h>{}, std::integral_constant<int, 2>{});      \
  put(std::integral_constant<int, h>{}, std::integral_constant<int, 3>{});
  PA_PUT4(0) PA_PUT4(1)
#undef PA_PUT4
}
template <int MIX, bool ATOMIC, bool COH = false>
__device__ __forceinline__ void epilogue_band(double* __restrict__ C, const unsigned touched, const int row0, const int col0,
                                              const int l15, const int g, const int tm1, const int tn1, const int ldc) {
  if (!((touched >> MIX) & 1u)) return;
  epilogue_half<MIX, ATOMIC, COH, 0>(C, touched, row0, col0, l15, g, tm1, tn1, ldc);
  epilogue_half<MIX, ATOMIC, COH, 2>(C, touched, row0, col0, l15, g, tm1, tn1, ldc);
}

// ---- one 16-deep chunk of a piece that covers PART of the tile, for ONE pattern of active sub-tiles ------------------
// Round 6.  The masked loops used to pick the pattern's MFMAs with a scalar switch PER K-STEP (four per chunk): on the
// structurized control flow the compiler makes of it that is ~30 scalar instructions and ~10 branches per k-step and wave,
// and a sparse chunk cost 0.29 of a whole one whatever it multiplied (DESIGN.md 9 "where the chunks go": the floor was the
// instruction stream of a barrier-to-barrier iteration).  Now the switch is taken ONCE per chunk and every arm is the
// chunk's whole body as straight-line code for its pattern: the operand reads of the k-steps 1-3 fetch only the bands the
// pattern multiplies, the chunk barrier and the prefetch of the next chunk's first k-step (all six operands: the next
// chunk may belong to a piece with another pattern) sit in front of the last k-step's MFMAs as in the branch-free loop.
// Second step (same round): the operand reads and their waits are OURS, as in the whole-tile loop below (piece_loop_w: why):
// `ds_read_b64` with immediate offsets from the byte address of the lane's first operand in the chunk's buffer, and ONE
// counted wait per k-step -- "all but the reads just issued".
template <int OFF>
__device__ __forceinline__ double lds_rd(const uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b64 has a 16-bit offset");
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%c2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
constexpr int popc4(const unsigned x) { return (int)((x & 1u) + ((x >> 1) & 1u) + ((x >> 2) & 1u) + ((x >> 3) & 1u)); }
// the operands of the row bands RM / column bands CM of the k-step at byte offset OFF of the buffer at (aA, aB)
template <unsigned RM, unsigned CM, int OFF>
__device__ __forceinline__ void read_sel(double (&bm)[MI], double (&an)[NI], const uint32_t aA, const uint32_t aB) {
  if constexpr (RM & 1u) bm[0] = lds_rd<OFF>(aA);
  if constexpr (RM & 2u) bm[1] = lds_rd<OFF + RS * 8>(aA);
  if constexpr (CM & 1u) an[0] = lds_rd<OFF>(aB);
  if constexpr (CM & 2u) an[1] = lds_rd<OFF + CS * 8>(aB);
  if constexpr (CM & 4u) an[2] = lds_rd<OFF + 2 * CS * 8>(aB);
  if constexpr (CM & 8u) an[3] = lds_rd<OFF + 3 * CS * 8>(aB);
}
// wait until at most N LDS reads are outstanding; NEG: then flip the sign of the A operands (the wait takes them as
// operands, so that the flip cannot be scheduled in front of it; a v_xor result feeds the MFMA inside an asm statement: its
// wait states are ours)
template <int N, unsigned RM, bool NEG>
__device__ __forceinline__ void wait_sel(double (&bm)[MI], const bool neg) {
  if constexpr (NEG && RM == 3u) asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(bm[0]), "+v"(bm[1]) : "n"(N));
  else if constexpr (NEG && RM == 1u) asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(bm[0]) : "n"(N));
  else if constexpr (NEG && RM == 2u) asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(bm[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(N));
  if constexpr (NEG && RM != 0u) {
    if (neg) {
      if constexpr (RM & 1u) bm[0] = -bm[0];
      if constexpr (RM & 2u) bm[1] = -bm[1];
      if constexpr (RM == 3u) asm volatile("s_nop 1" : "+v"(bm[0]), "+v"(bm[1]));
      else if constexpr (RM == 1u) asm volatile("s_nop 1" : "+v"(bm[0]));
      else asm volatile("s_nop 1" : "+v"(bm[1]));
    }
  }
}
// (A k-step in two halves -- the wave's lower two column bands, then its upper two: the operands of the NEXT k-step are
// read half by half as well, so that at most 20 operand registers are live at a time instead of 24; with all eight
// sub-tiles active the arm otherwise spilled four registers around itself, behind a vmcnt(0) that waited for the DMA just
// issued.  Every sub-tile still receives its MFMAs in k order: the factors do not change.)
template <unsigned RM, unsigned CM, bool NEG, int OFFN>
__device__ __forceinline__ void kstep_halves(double (&an)[NI], double (&bm)[MI], double (&ann)[NI], double (&bmn)[MI],
                                             const uint32_t aA, const uint32_t aB, const bool neg) {
  constexpr unsigned CL = CM & 3u, CH = CM & 12u;
  read_sel<RM, CL, OFFN>(bmn, ann, aA, aB);
  wait_sel<popc4(RM) + popc4(CL), RM, NEG>(bm, neg);       // this k-step's operands are in (read one k-step ago)
  mfma_sel<RM, CL>(an, bm);
  if constexpr (CH != 0u) {
    read_sel<0u, CH, OFFN>(bmn, ann, aA, aB);
    mfma_sel<RM, CH>(an, bm);
  }
}
template <unsigned RM, unsigned CM, bool NEG>
__device__ __forceinline__ void chunk_arm(const uint32_t aA, const uint32_t aB, const uint32_t nA, const uint32_t nB,
                                          double (&bm0)[MI], double (&an0)[NI], double* fixb, const int fix, const bool neg) {
  double bm1[MI], an1[NI];
  constexpr int KB = 4 * SLD * 8;                          // bytes between k-steps of a buffer
  // (ks0's operands -- all six -- were waited for at the end of the previous chunk's arm / of the prologue)
  kstep_halves<RM, CM, NEG, 1 * KB>(an0, bm0, an1, bm1, aA, aB, neg);     // ks0, operands of ks1 read beside it
  kstep_halves<RM, CM, NEG, 2 * KB>(an1, bm1, an0, bm0, aA, aB, neg);     // ks1
  kstep_halves<RM, CM, NEG, 3 * KB>(an0, bm0, an1, bm1, aA, aB, neg);     // ks2
  __syncthreads();       // vmcnt(0) lgkmcnt(0) s_barrier: next chunk landed, this buffer fully read, ks3's operands in
  if (fix >= 0) {        // (the stray element of the next chunk's piece, before this wave's reads of that buffer)
    fixb[fix] = zero_now();
  }
  // ks3 from registers, the next chunk's first k-step read beside it (all six operands: the next chunk may belong to a
  // piece with another pattern), again in halves; nothing stays in flight across the pattern switch
  constexpr unsigned CL = CM & 3u, CH = CM & 12u;
  read_sel<3u, 3u, 0>(bm0, an0, nA, nB);
  wait_sel<4, RM, NEG>(bm1, neg);
  mfma_sel<RM, CL>(an1, bm1);
  read_sel<0u, 12u, 0>(bm0, an0, nA, nB);
  mfma_sel<RM, CH>(an1, bm1);
  asm volatile("s_waitcnt lgkmcnt(0)");
}
template <bool NEG>
__device__ __forceinline__ void chunk_pat(const int pat, const uint32_t aA, const uint32_t aB, const uint32_t nA, const uint32_t nB,
                                          double (&bm0)[MI], double (&an0)[NI], double* fixb, const int fix, const bool neg) {
#define PA_PAT(RM, CM) case (RM | (CM << 2)): chunk_arm<RM, CM, NEG>(aA, aB, nA, nB, bm0, an0, fixb, fix, neg); break;
#define PA_PAT_ROWS(CM) PA_PAT(3u, CM) PA_PAT(1u, CM) PA_PAT(2u, CM)
  switch (pat) {
    PA_PAT_ROWS(15u) PA_PAT_ROWS(3u) PA_PAT_ROWS(6u) PA_PAT_ROWS(12u) PA_PAT_ROWS(7u) PA_PAT_ROWS(14u)
    PA_PAT_ROWS(1u) PA_PAT_ROWS(2u) PA_PAT_ROWS(4u) PA_PAT_ROWS(8u)
    default: chunk_arm<0u, 0u, NEG>(aA, aB, nA, nB, bm0, an0, fixb, fix, neg); break;   // no sub-tile of this wave in the chunk's piece
  }
#undef PA_PAT_ROWS
#undef PA_PAT
}

// ---- the piece loop of whole tiles, without sign flips (round 6) ------------------------------------------------------
// Whole-tile pieces of a full 128 x 128 tile (97.6 % of the update flops at 200^3).  The pipeline of rounds 1-5 -- DMA(i+1) |
// ks0..ks2 | vmcnt(0)+lgkmcnt(0)+barrier | read (i+1, ks0) | MFMA ks3; RAW: own vmcnt(0), then the barrier, then the read; WAR:
// buffer i is re-filled by DMA(i+2), issued after this barrier, which every wave passes with its reads retired -- with two
// changes:
// (a) the DMA is a buffer load to LDS on one descriptor per operand and piece (as in the masked loops): the lane's offset
//     is ONE 32-bit register for the whole task, the k-line goes into the scalar offset, and a k-line beyond K gets a
//     scalar offset out of the descriptor's range (zeros land in LDS): no zero line, no 64-bit lane addresses;
// (b) the operand reads and their waits are OURS (inline assembly).  With plain C++ loads the compiler puts
//     `s_waitcnt lgkmcnt(0)` between the reads of k-step s + 1 and the MFMAs of k-step s in two of the four k-steps -- once a
//     scalar load (the next piece's record) may be in flight on some path into the loop it cannot count on the in-order
//     return of LDS reads --, so those reads were not under the MFMAs.  Here every read is a `ds_read_b64` with an
//     immediate offset from one base register per operand image (the loop is unrolled over the two buffers: every offset is
//     a constant) and the waits are counted: `lgkmcnt(6)` = "all but the six reads just issued" (LDS returns in order; an
//     outstanding scalar load only makes such a wait stricter, never wrong).  Nothing is in flight across the back edge.
// the six operands of k-step KS of buffer B: rows of the wave's two row bands (A image), of its four column bands (B image)
template <int B, int KS>
__device__ __forceinline__ void lds_operands(double (&bm)[MI], double (&an)[NI], const uint32_t aA, const uint32_t aB) {
  constexpr int O = (B * 2 * KC * SLD + 4 * KS * SLD) * 8;
  bm[0] = lds_rd<O>(aA);
  bm[1] = lds_rd<O + RS * 8>(aA);
  an[0] = lds_rd<O>(aB);
  an[1] = lds_rd<O + CS * 8>(aB);
  an[2] = lds_rd<O + 2 * CS * 8>(aB);
  an[3] = lds_rd<O + 3 * CS * 8>(aB);
}
// (NEG: the task has "+=" pieces, the cross terms of complex products.  The A operands of such a piece are negated on the
// vector unit between their arrival and the MFMAs: the counted wait takes them as operands, so that the sign flip cannot be
// scheduled in front of it.)
template <bool NEG, int N>
__device__ __forceinline__ void lds_wait(double (&bm)[MI], const bool neg) {
  if constexpr (NEG) {
    if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bm[0]), "+v"(bm[1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bm[0]), "+v"(bm[1]));
    if (neg) {
      bm[0] = -bm[0];
      bm[1] = -bm[1];
      asm volatile("s_nop 1" : "+v"(bm[0]), "+v"(bm[1]));   // (a v_xor result feeds the MFMA inside an asm statement: its wait states are ours)
    }
  } else {
    if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)");
    else asm volatile("s_waitcnt lgkmcnt(0)");
  }
}
template <bool NEG>
__device__ __forceinline__ unsigned piece_loop_w(double (&sh)[2][2][KC * SLD], const Arenas& ar, const Task& tk,
                                                 const Piece* __restrict__ pieces, const int row0, const int col0,
                                                 const int lane, const int l15, const int g, const int wave_s) {
  constexpr int NL = KC / UW;                    // k-lines per wave per operand per chunk
  const int wave = wave_s;
  const int pend = tk.p0 + tk.pn;
  int pi = tk.p0;
  Piece cur = pieces[pi];
  Piece nextp = pieces[min(pi + 1, pend - 1)];
  __amdgpu_buffer_rsrc_t ra, rb;
  int lda8 = 0, kk = 0, kb = 0;                  // bytes between k-lines, K, first k-line of the chunk being copied
  const uint32_t vo = 16u * (uint32_t)lane;      // this lane's 16 bytes of a k-line
  auto setup = [&](const Piece& pc) {
    lda8 = __builtin_amdgcn_readfirstlane(pc.lda * 8);
    kk = __builtin_amdgcn_readfirstlane((int)pc.k);
    const int ext = (kk - 1) * lda8 + 8 * TM;
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[pc.flags & 3] + pc.a_off), (short)0, ext, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[(pc.flags >> 2) & 3] + pc.b_off), (short)0, ext, 0x00020000);
    kb = 0;
  };
  auto dma = [&](double* dA, double* dB) {
#pragma unroll
    for (int q = 0; q < NL; q++) {
      const int kl = kb + wave + UW * q;                                   // wave-uniform
      const uint32_t so = kl < kk ? (uint32_t)(kl * lda8) : 0x40000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dA + UW * q * SLD), 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dB + UW * q * SLD), 16, vo, so, 0, 0);
    }
    kb += KC;
  };
  setup(cur);
  int left = ((int)cur.k + KC - 1) / KC;
  bool negn = NEG && (cur.flags & 16) != 0;       // sign of the piece whose chunk is being COPIED
  // the piece bookkeeping of one chunk iteration: which piece the NEXT chunk belongs to; false: there is none
  auto advance = [&]() -> bool {
    if (--left == 0) {
      if (++pi < pend) {
        cur = nextp;
        setup(cur);
        left = ((int)cur.k + KC - 1) / KC;
        negn = NEG && (cur.flags & 16) != 0;
        nextp = pieces[min(pi + 1, pend - 1)];
      } else {
        return false;
      }
    }
    return true;
  };
  dma(sh[0][0] + wave * SLD, sh[0][1] + wave * SLD);
  // LDS byte addresses of this lane's first A / B operand (buffer 0, k-step 0)
  const uint32_t aA = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)(sh[0][0] + row0 + l15 + g * SLD);
  const uint32_t aB = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)(sh[0][1] + col0 + l15 + g * SLD);
  double bm0[MI], an0[NI], bm1[MI], an1[NI];
  __syncthreads();                                // (emits vmcnt(0): the DMA of chunk 0 has landed)
  lds_operands<0, 0>(bm0, an0, aA, aB);
  lds_wait<NEG, 0>(bm0, negn);
  // one chunk iteration on buffer B (returns false after the last chunk)
  auto step = [&](auto BC) -> bool {
    constexpr int B = decltype(BC)::value;
    const bool negc = negn;                       // sign of the chunk being MULTIPLIED (its ks0 operands carry it already)
    const bool has_next = advance();
    if (has_next) dma(sh[B ^ 1][0] + wave * SLD, sh[B ^ 1][1] + wave * SLD);
    lds_operands<B, 1>(bm1, an1, aA, aB);
    mfma_sel<3u, 15u>(an0, bm0);                  // ks0 (its operands were waited for behind the previous chunk's ks3)
    lds_operands<B, 2>(bm0, an0, aA, aB);
    lds_wait<NEG, 6>(bm1, negc);                  // ks1's operands are in; ks2's six may be on their way
    mfma_sel<3u, 15u>(an1, bm1);                  // ks1
    lds_operands<B, 3>(bm1, an1, aA, aB);
    lds_wait<NEG, 6>(bm0, negc);
    mfma_sel<3u, 15u>(an0, bm0);                  // ks2
    __syncthreads();         // vmcnt(0) lgkmcnt(0) s_barrier: next chunk landed, this buffer fully read, ks3's operands in
    if constexpr (NEG) lds_wait<NEG, 0>(bm1, negc);
    // (unconditional: after the last chunk it re-reads a landed buffer; the values are not used)
    lds_operands<B ^ 1, 0>(bm0, an0, aA, aB);
    mfma_sel<3u, 15u>(an1, bm1);                  // ks3 from registers
    lds_wait<NEG, 0>(bm0, negn);                  // (nothing in flight across the back edge; hidden by the MFMAs above)
    return has_next;
  };
  while (step(std::integral_constant<int, 0>{}) && step(std::integral_constant<int, 1>{})) {}
  return 0x3Fu;
}

// ---- the piece loop of the MASKED instances (round 6) ---------------------------------------------------------------
// PART = false: whole-tile pieces of a smaller valid tile (MODE 1 above: lane masks and pattern fixed per task); PART = true:
// partial pieces (MODE 2: per piece).  Same pipeline as piece_loop_w -- DMA(i+1) | ks0..ks2 | barrier | read (i+1, ks0) | ks3 --,
// but (a) the pattern switch is taken once per CHUNK (chunk_pat) and (b) the DMA is a BUFFER load to LDS
// (buffer_load_dwordx4 ... offen lds): one descriptor per operand and piece, the lane's byte offset a 32-bit register
// computed once per piece -- 0x80000000, i.e. out of the descriptor's range, for lanes whose two rows lie outside the piece:
// the hardware writes zeros into LDS for those, and for every lane of a k-line beyond K (scalar offset 0x40000000) --: no
// zero line, no 64-bit address selects per DMA.  The descriptors cover whole 16-byte lanes: a lane that straddles an odd
// piece boundary brings the neighbouring source row along as before (zeroed in LDS, fixn).  The plan bounds a source
// panel's (width + 16) x stride x 8 below 2^30, so a valid offset never reaches the out-of-range markers and their sum does
// not wrap.
template <bool PART, bool NEG>
__device__ __forceinline__ unsigned piece_loop_m(double (&sh)[2][2][KC * SLD], const Arenas& ar, const Task& tk,
                                                 const Piece* __restrict__ pieces, const int row0, const int col0,
                                                 const int lane, const int l15, const int g, const int wave_s) {
  constexpr int NL = KC / UW;
  const int wave = wave_s;
  const int wrow0 = (wave >> 1) * 16, wcol0 = (wave & 1) * 16;
  const int pend = tk.p0 + tk.pn;
  int pi = tk.p0;
  Piece cur = pieces[pi];
  Piece nextp = pieces[min(pi + 1, pend - 1)];
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t va = 0, vb = 0;                         // this lane's byte offsets inside a k-line (0x80000000: none)
  int lda8 = 0, kk = 0, kb = 0;                    // bytes between k-lines, K, first k-line of the chunk being copied
  int patn = 0, fixn = -1;
  unsigned touched = 0;
  auto band_pattern = [&](const int r_lo, const int r_hi, const int c_lo, const int c_hi) {   // scalar arithmetic
    int p = 0;
#pragma unroll
    for (int s = 0; s < MI; s++) if (wrow0 + s * RS < r_hi && wrow0 + s * RS + 16 > r_lo) p |= 1 << s;
#pragma unroll
    for (int s = 0; s < NI; s++) if (wcol0 + s * CS < c_hi && wcol0 + s * CS + 16 > c_lo) p |= 1 << (MI + s);
    return ((p & 3) && (p >> MI)) ? p : 0;
  };
  auto setup = [&](const Piece& pc) {
    const int dr = PART ? (int)pc.dr : 0, re = PART ? (int)pc.dr + (int)pc.m : (int)tk.tm;
    const int dc = PART ? (int)pc.dc : 0, ce = PART ? (int)pc.dc + (int)pc.n : (int)tk.tn;
    const int dre = dr & ~1, dce = dc & ~1;        // the tile row / column of lane 0's first element of the operand image
    lda8 = __builtin_amdgcn_readfirstlane(pc.lda * 8);
    kk = __builtin_amdgcn_readfirstlane((int)pc.k);
    const int ext = (kk - 1) * lda8 + 8;
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[pc.flags & 3] + pc.a_off - (dr - dre)), (short)0, ext + 8 * (re - dre), 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[(pc.flags >> 2) & 3] + pc.b_off - (dc - dce)), (short)0, ext + 8 * (ce - dce), 0x00020000);
    // (what derives from the lane index is recomputed per piece from a laundered copy, not kept in registers across the
    // chunks: the arms of the chunk switch need the 64 VGPRs for operands)
    const int ln = lane_now();
    const int l2 = 2 * ln;
    const uint32_t oa = !(l2 + 1 >= dr && l2 < re), ob = !(l2 + 1 >= dc && l2 < ce);
    // (in range: < 1024; bit 31 = out of the descriptor's range, and nothing above bit 9 that could wrap the sum with the
    // scalar offset back into it)
    va = ((uint32_t)(8 * (l2 - dre)) & 0x3ffu) | (oa << 31);
    vb = ((uint32_t)(8 * (l2 - dce)) & 0x3ffu) | (ob << 31);
    kb = 0;
    if (PART) {
      patn = band_pattern(dr, re, dc, ce);
      // stray elements: a lane that straddles an odd boundary brings the source row next to the piece along (lanes 0-15 /
      // 16-31 / 32-47 / 48-63 look after the rows dr-1, dr+m of A and dc-1, dc+n of B, one k-line each)
      const int j = ln >> 4;
      const int e = j == 0 ? dr - 1 : j == 1 ? re : j == 2 ? dc - 1 : ce;      // the row next to the boundary
      const bool odd = (j == 0 || j == 2) ? (e & 1) == 0 && e >= 0 : (e & 1) != 0 && e < 128;   // shares a lane with a piece row
      fixn = odd ? (j >= 2 ? KC * SLD : 0) + (ln & 15) * SLD + e : -1;
      touched |= (unsigned)patn;
    }
  };
  // the DMA of one chunk: k-lines kb + wave, kb + wave + 8 of both operands
  auto dma = [&](double* dA, double* dB) {
#pragma unroll
    for (int q = 0; q < NL; q++) {
      const int kl = kb + wave + UW * q;                                   // wave-uniform
      const uint32_t so = kl < kk ? (uint32_t)(kl * lda8) : 0x40000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dA + UW * q * SLD), 16, va, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dB + UW * q * SLD), 16, vb, so, 0, 0);
    }
    kb += KC;
  };
  if (!PART) {
    patn = band_pattern(0, (int)tk.tm, 0, (int)tk.tn);
    touched = (unsigned)patn;
  }
  setup(cur);
  int left = ((int)cur.k + KC - 1) / KC;
  bool negn = (cur.flags & 16) != 0, negc = negn;
  int patc = patn;
  dma(sh[0][0] + wave * SLD, sh[0][1] + wave * SLD);
  // LDS byte addresses of this lane's first A / B operand (buffer 0, k-step 0)
  const uint32_t aA0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)(sh[0][0] + row0 + l15 + g * SLD);
  const uint32_t aB0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) double*)(sh[0][1] + col0 + l15 + g * SLD);
  double bm0[MI], an0[NI];
  __syncthreads();                                // (emits vmcnt(0): the DMA of chunk 0 has landed)
  if (PART && fixn >= 0) sh[0][0][fixn] = zero_now();    // (every wave, before its own reads: LDS is in order per wave)
  read_sel<3u, 15u, 0>(bm0, an0, aA0, aB0);
  asm volatile("s_waitcnt lgkmcnt(0)");
  int buf = 0;
  while (true) {
    bool has_next = true;
    negc = negn;
    patc = patn;
    if (--left == 0) {
      if (++pi < pend) {
        cur = nextp;
        setup(cur);
        left = ((int)cur.k + KC - 1) / KC;
        negn = (cur.flags & 16) != 0;
        nextp = pieces[min(pi + 1, pend - 1)];
      } else {
        has_next = false;
      }
    }
    if (has_next) dma(sh[buf ^ 1][0] + wave * SLD, sh[buf ^ 1][1] + wave * SLD);
    constexpr uint32_t BB = 2 * KC * SLD * 8;      // bytes between the two buffers
    chunk_pat<NEG>(patc, aA0 + buf * BB, aB0 + buf * BB, aA0 + (buf ^ 1) * BB, aB0 + (buf ^ 1) * BB, bm0, an0, &sh[buf ^ 1][0][0],
                   (PART && has_next) ? fixn : -1, negc);
    if (!has_next) break;
    buf ^= 1;
  }
  return touched;
}

// ---- the piece loop of tasks with GATHERED pieces (MODE 3) -----------------------------------------------------------
// On layouts whose bloks are fragments of a few rows (blend on separators numbered across their low-side neighbours: 2-4
// rows every 50-60) the contribution of one source cblk to one target tile is dozens of rectangles of 2-4 rows by 2-4
// columns, each a pass of K / 16 chunks through the loop above for a few percent of a tile's MFMA work.  The SOURCE rows of
// all of them are consecutive in the source panel (its bloks are stacked), only where they land is scattered.  A gathered
// piece is the whole set in one pass: the LDS images are laid out by TARGET row / column as always, and every 4-byte DMA
// lane fetches ITS tile row's half of a double from wherever that row's source is -- buffer_load_dword ... lds, 64 lanes =
// 32 tile rows, four instructions per k-line and operand instead of one; rows without a source read out of the
// descriptor's range, i.e. zeros.  (The
// reference's CUDA kernel walks blocktab inside one GEMM to the same end, sparse_gemm.cu:103-479.)  Ordinary pieces of the
// same task run through this loop too, their maps made up on the fly from (dr, m) / (dc, n).  Accumulation order = the order
// of the list, as in every other loop: the factors do not depend on which loop ran.
// the MFMAs of one k-step for ANY set of the wave's sub-tiles (a gathered piece's active column bands need not be a run)
__device__ __forceinline__ void mfma_any(const int pat, const double (&an)[NI], const double (&bm)[MI]) {
#define PA_PAT(RM, CM) case (RM | (CM << 2)): mfma_sel<RM, CM>(an, bm); break;
#define PA_PAT_ROWS(CM) PA_PAT(3u, CM) PA_PAT(1u, CM) PA_PAT(2u, CM)
  switch (pat) {
    PA_PAT_ROWS(1u) PA_PAT_ROWS(2u) PA_PAT_ROWS(3u) PA_PAT_ROWS(4u) PA_PAT_ROWS(5u) PA_PAT_ROWS(6u) PA_PAT_ROWS(7u) PA_PAT_ROWS(8u)
    PA_PAT_ROWS(9u) PA_PAT_ROWS(10u) PA_PAT_ROWS(11u) PA_PAT_ROWS(12u) PA_PAT_ROWS(13u) PA_PAT_ROWS(14u) PA_PAT_ROWS(15u)
    default: break;
  }
#undef PA_PAT_ROWS
#undef PA_PAT
}
template <bool NEG>
__device__ __forceinline__ unsigned piece_loop_g(double (&sh)[2][2][KC * SLD], const Arenas& ar, const Task& tk,
                                                 const Piece* __restrict__ pieces, const int row0, const int col0,
                                                 const int lane, const int l15, const int g, const int wave_s) {
  constexpr int NL = KC / UW;
  const int wave = wave_s;
  const int pend = tk.p0 + tk.pn;
  int pi = tk.p0;
  const int li = lane >> 1;                          // this lane's tile rows / columns: li + 32 q, q = 0..3
  const int half4 = (lane & 1) * 4;                  // which half of the double (bytes)
  // the packed maps of a piece: byte q = source row of tile row li + 32 q (255: none)
  auto maps_of = [&](const Piece& pc, uint32_t& ma, uint32_t& mb) {
    if (pc.flags & PIECE_GATHERED) {
      const uint32_t* mp = ar.gmap + (size_t)((uint32_t)pc.dr | ((uint32_t)pc.dc << 16)) * 64;
      ma = mp[li];
      mb = mp[32 + li];
    } else {
      ma = 0; mb = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = li + 32 * q;
        const uint32_t sa = (r >= (int)pc.dr && r < (int)pc.dr + (int)pc.m) ? (uint32_t)(r - (int)pc.dr) : 255u;
        const uint32_t sb = (r >= (int)pc.dc && r < (int)pc.dc + (int)pc.n) ? (uint32_t)(r - (int)pc.dc) : 255u;
        ma |= sa << (8 * q);
        mb |= sb << (8 * q);
      }
    }
  };
  // which of the wave's sub-tiles a piece touches: 16-row band b of the tile has a source iff a lane of its half of the
  // ballot of q = b / 2 has one
  auto pattern_of = [&](const uint32_t ma, const uint32_t mb) {
    int p = 0;
#pragma unroll
    for (int s = 0; s < MI; s++) {
      const int b = (wave >> 1) + 4 * s;             // row band
      const unsigned long long bal = __ballot(__builtin_amdgcn_ubfe(ma, 8 * (b >> 1), 8) != 255u);
      if ((uint32_t)(bal >> ((b & 1) * 32)) != 0u) p |= 1 << s;
    }
#pragma unroll
    for (int s = 0; s < NI; s++) {
      const int b = (wave & 1) + 2 * s;              // column band
      const unsigned long long bal = __ballot(__builtin_amdgcn_ubfe(mb, 8 * (b >> 1), 8) != 255u);
      if ((uint32_t)(bal >> ((b & 1) * 32)) != 0u) p |= 1 << (MI + s);
    }
    return ((p & 3) && (p >> MI)) ? p : 0;
  };
  Piece cur = pieces[pi];
  Piece nextp = pieces[min(pi + 1, pend - 1)];
  uint32_t ma, mb, nma, nmb;
  maps_of(cur, ma, mb);
  maps_of(nextp, nma, nmb);
  // The gathering DMA is a BUFFER load to LDS (buffer_load_dword ... offen lds): one descriptor per operand and piece whose
  // range is exactly the piece's source rows over its K columns, the lane's offset = 8 x (its source row) + its half, the
  // k-line in the scalar offset.  A lane without a source gets an offset beyond the range, and so does every lane of a
  // k-line beyond K: the hardware writes ZEROS into LDS for those (probed on gfx950) -- no zero line, no selects, no 64-bit
  // address arithmetic on the vector unit, no k-tail test.
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t va[4], vb[4];
  int lda8 = 0;                                      // bytes between k-lines (wave-uniform)
  auto setup = [&](const Piece& pc, const uint32_t pa_map, const uint32_t pb_map) {
    lda8 = __builtin_amdgcn_readfirstlane(pc.lda * 8);
    const int kk = __builtin_amdgcn_readfirstlane((int)pc.k), mm = __builtin_amdgcn_readfirstlane((int)pc.m),
              nn = __builtin_amdgcn_readfirstlane((int)pc.n);
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[pc.flags & 3] + pc.a_off), (short)0, (kk - 1) * lda8 + 8 * mm, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)(ar.p[(pc.flags >> 2) & 3] + pc.b_off), (short)0, (kk - 1) * lda8 + 8 * nn, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      // (v_bfe_u32 and a shift-or: no mask or marker constants for the compiler to keep in registers across the kernel)
      const uint32_t sa = __builtin_amdgcn_ubfe(pa_map, 8 * q, 8), sb = __builtin_amdgcn_ubfe(pb_map, 8 * q, 8);
      va[q] = (8u * sa + (uint32_t)half4) | ((uint32_t)(sa == 255u) << 31);
      vb[q] = (8u * sb + (uint32_t)half4) | ((uint32_t)(sb == 255u) << 31);
    }
  };
  setup(cur, ma, mb);
  int kb = 0;                                        // first k-line of the chunk being copied
  int left = ((int)cur.k + KC - 1) / KC;
  bool negn = (cur.flags & 16) != 0, negc = negn;
  int patn = pattern_of(ma, mb), patc = patn;
  unsigned touched = (unsigned)patn;
  // the DMA of one chunk: k-lines kb + wave, kb + wave + 8 of both operands, four 4-byte instructions each
  auto dma = [&](double* dA, double* dB) {
#pragma unroll
    for (int kq = 0; kq < NL; kq++) {
      const uint32_t so = (uint32_t)((kb + wave + UW * kq) * lda8);      // wave-uniform
#pragma unroll
      for (int q = 0; q < 4; q++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)((char*)(dA + UW * kq * SLD) + 256 * q), 4,
                                                 va[q], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)((char*)(dB + UW * kq * SLD) + 256 * q), 4,
                                                 vb[q], so, 0, 0);
      }
    }
  };
  dma(sh[0][0] + wave * SLD, sh[0][1] + wave * SLD);
  kb += KC;
  const double* sAw = sh[0][0] + row0 + l15 + g * SLD;
  const double* sBw = sh[0][1] + col0 + l15 + g * SLD;
  double bm0[MI], an0[NI], bm1[MI], an1[NI];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < MI; s++) bm0[s] = sAw[s * RS];
#pragma unroll
  for (int s = 0; s < NI; s++) an0[s] = sBw[s * CS];
  int buf = 0;
  while (true) {
    bool has_next = true;
    negc = negn;
    patc = patn;
    if (--left == 0) {
      if (++pi < pend) {
        cur = nextp;
        ma = nma; mb = nmb;
        setup(cur, ma, mb);
        kb = 0;
        left = ((int)cur.k + KC - 1) / KC;
        negn = (cur.flags & 16) != 0;
        nextp = pieces[min(pi + 1, pend - 1)];
        maps_of(nextp, nma, nmb);
        patn = pattern_of(ma, mb);
        touched |= (unsigned)patn;
      } else {
        has_next = false;
      }
    }
    if (has_next) {
      dma(sh[buf ^ 1][0] + wave * SLD, sh[buf ^ 1][1] + wave * SLD);
      kb += KC;
    }
    const double* sA = sAw + buf * (2 * KC * SLD);
    const double* sB = sBw + buf * (2 * KC * SLD);
#define PA_NEGATE(bm)                                        \
  if (NEG && negc) {                                         \
    _Pragma("unroll") for (int s = 0; s < MI; s++) bm[s] = -bm[s]; \
    asm volatile("s_nop 1");                                 \
  }
    PA_NEGATE(bm0)
#pragma unroll
    for (int s = 0; s < MI; s++) bm1[s] = sA[4 * SLD + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an1[s] = sB[4 * SLD + s * CS];
    mfma_any(patc, an0, bm0);
    PA_NEGATE(bm1)
#pragma unroll
    for (int s = 0; s < MI; s++) bm0[s] = sA[8 * SLD + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an0[s] = sB[8 * SLD + s * CS];
    mfma_any(patc, an1, bm1);
    PA_NEGATE(bm0)
#pragma unroll
    for (int s = 0; s < MI; s++) bm1[s] = sA[12 * SLD + s * RS];
#pragma unroll
    for (int s = 0; s < NI; s++) an1[s] = sB[12 * SLD + s * CS];
    mfma_any(patc, an0, bm0);
    PA_NEGATE(bm1)
    __syncthreads();       // vmcnt(0) lgkmcnt(0) s_barrier: next chunk landed, this buffer fully read
    {
      const double* nA = sAw + (buf ^ 1) * (2 * KC * SLD);
      const double* nB = sBw + (buf ^ 1) * (2 * KC * SLD);
#pragma unroll
      for (int s = 0; s < MI; s++) bm0[s] = nA[s * RS];
#pragma unroll
      for (int s = 0; s < NI; s++) an0[s] = nB[s * CS];
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_any(patc, an1, bm1);
    if (!has_next) break;
    buf ^= 1;
  }
#undef PA_NEGATE
  return touched;
}

// Which loop runs a task's pieces.  Tasks of whole-tile pieces take MODE 0 / 1, a gathered task the gathering loop; a task
// that MIXES whole-tile pieces (the plan lists them first: Task::nfull) with partial ones runs its head on the branch-free
// loop and its tail on the masked one (round 5; before, every piece of such a task went through the masked loop: 7-10 % of
// the chunks at 100^3 / 60^3) -- the accumulators stay where they are between the two, the order of the list is kept, so
// the factors do not change.  The same for a task with gathered pieces: its whole-tile head (15 % of the flops of a
// fragmented 60^3 layout) on the branch-free loop, the rest through the gathering loop.  One call site per instance (the
// loops are inlined).
__device__ __forceinline__ unsigned update_pieces(double (&sh)[2][2][KC * SLD], const Arenas& ar, const Task& tk,
                                                  const Piece* __restrict__ pieces, const int wave) {
  const bool neg = (tk.flags & 8u) != 0;
  const bool gath = (tk.flags & TASK_GATHERED) != 0;   // the tail of such a task (everything but whole-tile pieces) gathers
  const bool fullt = tk.tm == TM && tk.tn == TN;
  Task t = tk;
  int left = tk.pn, nfull = (int)tk.nfull;
  unsigned touched = 0;
  for (;;) {
    // (what derives from the thread index is recomputed per pass, not kept in registers across the loops: 64 VGPRs)
    const int lane = lane_now();
    const int row0 = (wave >> 1) * 16, col0 = (wave & 1) * 16;
    const int l15 = lane & 15, g = lane >> 4;
    int mode;
    if (nfull == left && !gath) { mode = fullt ? 0 : 1; t.pn = left; }
    else if (nfull > 0 && fullt) { mode = 0; t.pn = nfull; }
    else { mode = gath ? 3 : 2; t.pn = left; }
    if (mode == 0) {
      if (neg) touched |= piece_loop_w<true>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
      else touched |= piece_loop_w<false>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
    } else if (mode == 1) {
      if (neg) touched |= piece_loop_m<false, true>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
      else touched |= piece_loop_m<false, false>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
    } else if (mode == 2) {
      if (neg) touched |= piece_loop_m<true, true>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
      else touched |= piece_loop_m<true, false>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
    } else {
      if (neg) touched |= piece_loop_g<true>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
      else touched |= piece_loop_g<false>(sh, ar, t, pieces, row0, col0, lane, l15, g, wave);
    }
    left -= t.pn;
    if (left <= 0) break;
    t.p0 += t.pn;
    nfull = 0;
    __syncthreads();        // (the head's last chunk is read before the tail's first DMA overwrites its buffer)
  }
  return touched;
}

// KIND 0: the bulk launches.  KIND 1 (`k_update<1>` in profiles): the same code for the few latency-critical tasks of a
// level that the two-stream driver runs beside the bulk launch of the previous slot; a separate instantiation so that
// per-kernel profiles of the two do not mix.
template <int KIND>
__global__ __launch_bounds__(64 * UW, UW / 2) void k_update(const Arenas ar, const Task* __restrict__ tasks,
                                                           const Piece* __restrict__ pieces) {
  extern __shared__ double sh_dyn[];            // UPDATE_LDS_BYTES, dynamic: see the Makefile (register budget)
  double (&sh)[2][2][KC * SLD] = *reinterpret_cast<double (*)[2][2][KC * SLD]>(sh_dyn);   // [buffer][A|B]  73,728 bytes
  if (KIND == 1) PANEL_PRIO();
  const Task tk = tasks[blockIdx.x];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the only use of threadIdx in the kernel)
  acc_zero();
  // (update_pieces: which loop instance runs which pieces.  The plan puts the whole-tile pieces of a task first, Task::nfull,
  // and sets Task flag 8 for sign flips.)
  const unsigned touched = update_pieces(sh, ar, tk, pieces, wave);

  // ---- epilogue: C -= acc (each register = 16 consecutive rows of one column)
  acc_settle();
  // (the thread's coordinates are made here, behind the loops: lane_now)
  const int lane = lane_now();
  const int row0 = (wave >> 1) * 16, col0 = (wave & 1) * 16;      // this wave's first row / col band
  const int l15 = lane & 15, g = lane >> 4;
  double* C = ar.p[tk.flags & 3] + tk.c_off;
  const int tm1 = (int)tk.tm - 1, tn1 = (int)tk.tn - 1;
  if (tk.flags & 4) {
    epilogue_band<0, true>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
    epilogue_band<1, true>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
    return;
  }
  epilogue_band<0, false>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
  epilogue_band<1, false>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
}

// ---- the panel solve as a task of the run launch ---------------------------------------------------
// k_trsm_llt's algorithm (kernels.hip: X^T in 16 x 16 column tiles, X^T[ct] = Tinv[ct] (A^T[ct] - sum_p L[ct,p] X^T[p]), a
// wave per 16 panel rows) inside the register budget of the update kernel: the eight column tiles of a wave REST in the
// accumulation registers a[0:63] -- which the update tasks of the same kernel own anyway -- and only the tile being
// computed lives in VGPRs: finished tiles are read back one operand pair at a time where a later tile multiplies them.
// (With compiler-allocated tiles the routine needs 109 VGPRs beside the update path's 64 AGPRs: it spilled.)  The MFMAs
// here work on VGPRs only, so the AGPR traffic is plain VALU moves without MFMA hazards.
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
// MODE as in kernels_var.hip k_trsm_var (kernel_trsm, compute_trsm.c:58-114):
//   0 LLt   X = A L_d^-T                     (T = L_d read plain, Tinv = inverses of its 16 x 16 diagonal tiles)
//   1 LDLt  Y = A L_d^-T with unit L_d, then L = Y D^-1 into the L arena and Y = L D into the second arena (:92-113)
//   2 LU    L side: X = A U_d^-1              (T = U_d read transposed from the factored blok)
//   3 LU    U side: X = A' (L_d^T)^-1, unit   (in / out in the U arena, the second set of tile inverses)
template <int MODE, bool COH>
__device__ __forceinline__ void trsm_parked(double* __restrict__ L, double* __restrict__ U, const TrsmTask& tk,
                                            const double* __restrict__ dinv_ws, const int tid, double* __restrict__ ldsT = nullptr) {
  constexpr int NT = 8;
  // Round 6 (LLt / LDLt, MODE 0 / 1): the triangular factor in LDS.  Every one of the 36 steps below multiplies four operand
  // entries per lane of L_d (or of a tile inverse), and a step cannot start before they are there: read from memory that
  // was ONE L2 round trip per step on the chain of the top separator -- 31 us per ticket, a third of the period of a level at
  // 60^3.  The 28 off-diagonal 16 x 16 tiles of L_d and the 8 tile inverses are exactly the 73,728 bytes of the update
  // path's operand buffers, idle in such a ticket: all eight waves copy them in by LDS-DMA (72 wave-instructions of 1 KiB:
  // two per tile, 8 columns x 16 rows each), one barrier, and the steps read LDS (conflict-free: a wave reads 4 columns x
  // 16 rows = 512 contiguous bytes).
  // (MODE 3, the U side of LU, reads its factor plain like LLt and takes the same path; MODE 2 reads U_d transposed -- a
  // 16-byte DMA lane cannot transpose, and a transposed read of a plain tile is an 8-way bank conflict: it keeps its loads)
  constexpr bool LDST = MODE <= 1 || MODE == 3;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int ld = tk.stride, w = tk.width;
  const int rloc = wave * 16 + l15;
  const double* Ld = L + tk.off;                     // diagonal blok (factored; always in the L arena)
  const double* Ti = dinv_ws + tk.dinv_off + (MODE == 3 ? (int64_t)((w + 15) >> 4) * 256 : 0);
  if constexpr (LDST) {
    const int ws = __builtin_amdgcn_readfirstlane(wave);
    const int nb1 = ((w + 15) >> 4) - 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int j = ws + 8 * k;                      // wave-uniform: tile j >> 1, its columns 8 (j & 1) ...
      int t = j >> 1;
      const int h = j & 1;
      const double* src;
      if (t < 28) {
        int ct = 1;
        while (t >= ct) { t -= ct; ct++; }           // (ct, p = t): tile index ct (ct - 1) / 2 + p
        const int row = min(ct * 16 + (lane & 7) * 2, w - 1), col = min(t * 16 + 8 * h + (lane >> 3), w - 1);
        src = Ld + row + (int64_t)col * ld;
      } else {
        src = Ti + min(t - 28, nb1) * 256 + h * 128 + lane * 2;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ldsT + (j >> 1) * 256 + h * 128), 16, 0, 0);
    }
  }
  if (!LDST && wave * 16 >= tk.nrows) return;
  const bool rvalid = rloc < tk.nrows;
  double* X = (MODE == 3 ? U : L) + tk.off + tk.row0;
  double* Ap = X + rloc;                             // panel row of this lane
  const double* Apc = X + min(rloc, tk.nrows - 1);   // clamped: always readable
  static_for<NT>([&](auto CT) {
    constexpr int ct = decltype(CT)::value;
    double v[4];
    int lds = ld;
    asm volatile("" : "+s"(lds));
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = Apc[(int64_t)min(ct * 16 + g + 4 * q, w - 1) * lds];
    acc_write<ct, 0>((rvalid && ct * 16 + g + 0 < w) ? v[0] : 0.0);
    acc_write<ct, 1>((rvalid && ct * 16 + g + 4 < w) ? v[1] : 0.0);
    acc_write<ct, 2>((rvalid && ct * 16 + g + 8 < w) ? v[2] : 0.0);
    acc_write<ct, 3>((rvalid && ct * 16 + g + 12 < w) ? v[3] : 0.0);
    if (ct & 1) __builtin_amdgcn_sched_barrier(0);     // (two tiles' loads in flight at a time: the registers are few)
  });
  if constexpr (LDST) {
    __syncthreads();                                   // (vmcnt(0): the factor has landed in LDS, for every wave)
    if (wave * 16 >= tk.nrows) return;
  }
  // 36 steps in order: for ct = 0..7 the products with the finished tiles p < ct, then the tile's own step with Tinv[ct].
  // Every step multiplies four operand entries per lane read from memory (T[ct, p] or Tinv[ct]): the loads of step i + 1
  // are issued in front of the MFMAs of step i (two operand sets live; the scheduling barrier per step keeps the compiler
  // from hoisting more and spilling).
  constexpr int NSTEP = NT * (NT + 1) / 2;
  auto step_ct = [](int i) constexpr { int ct = 0; while (i > ct) { i -= ct + 1; ct++; } return ct; };
  auto step_p = [](int i) constexpr { int ct = 0; while (i > ct) { i -= ct + 1; ct++; } return i; };   // p == ct: the tile's own step
  auto load_step = [&](auto II, double (&o)[4]) {
    constexpr int i = decltype(II)::value;
    constexpr int ct = step_ct(i), p = step_p(i);
    // (the leading dimension is laundered per step: the address arithmetic of all 36 steps would otherwise be done up
    // front and kept in registers -- 67 spilled VGPRs)
    int lds = ld, ws = w;
    asm volatile("" : "+s"(lds), "+s"(ws));
    if constexpr (LDST) {
      constexpr int tile = p < ct ? ct * (ct - 1) / 2 + p : 28 + ct;
      const int li = ct * 16 + l15;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double lv = ldsT[tile * 256 + (g + 4 * q) * 16 + l15];
        o[q] = p < ct ? ((li < ws) ? -lv : 0.0) : lv;
      }
    } else if constexpr (p < ct) {
      const int li = ct * 16 + l15, lic = min(li, ws - 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int lc = min(p * 16 + g + 4 * q, ws - 1);
        const double lv = (MODE == 2) ? Ld[lc + (int64_t)lic * lds] : Ld[lic + (int64_t)lc * lds];
        o[q] = (li < ws) ? -lv : 0.0;
      }
    } else {
      const int nb1 = ((ws + 15) >> 4) - 1;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = Ti[min(ct, nb1) * 256 + l15 + 16 * (g + 4 * q)];
    }
  };
  double cu[4], nx[4];
  d4_t c;
  load_step(std::integral_constant<int, 0>{}, cu);
  static_for<NSTEP>([&](auto II) {
    constexpr int i = decltype(II)::value;
    constexpr int ct = step_ct(i), p = step_p(i);
    if constexpr (i + 1 < NSTEP) load_step(std::integral_constant<int, i + 1>{}, nx);
    // (tiles beyond the cblk's width hold zeros and stay zero: no branch on the width, which would cost registers)
    if constexpr (p == 0 || ct == 0) { c[0] = acc_read_m<ct, 0>(); c[1] = acc_read_m<ct, 1>(); c[2] = acc_read_m<ct, 2>(); c[3] = acc_read_m<ct, 3>(); }
    if constexpr (p < ct) {
      const double b0 = acc_read_m<p, 0>(), b1 = acc_read_m<p, 1>(), b2 = acc_read_m<p, 2>(), b3 = acc_read_m<p, 3>();
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[0], b0, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[1], b1, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[2], b2, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[3], b3, c, 0, 0, 0);
    } else {
      d4_t t = d4_t{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) t = __builtin_amdgcn_mfma_f64_16x16x4f64(cu[q], c[q], t, 0, 0, 0);
      // (the four writes read the D of the last MFMA: the first one waits for it, asm statements keep their order)
      acc_write<ct, 0, true>(t[0]); acc_write<ct, 1>(t[1]); acc_write<ct, 2>(t[2]); acc_write<ct, 3>(t[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) cu[q] = nx[q];
    __builtin_amdgcn_sched_barrier(0);
  });
  static_for<NT>([&](auto CT) {
    constexpr int ct = decltype(CT)::value;
    int lds = ld, gs = (tid & 63) >> 4;
    asm volatile("" : "+s"(lds), "+v"(gs));
    auto put = [&](auto QQ) {
      constexpr int q = decltype(QQ)::value;
      const int col = ct * 16 + gs + 4 * q;
      if constexpr (MODE == 1) {
        const double dv = Ld[(int64_t)min(col, w - 1) * (int64_t)(lds + 1)];
        if (rvalid && col < w) {
          const double y = acc_read<ct, q>();
          pst<COH>(&(U + tk.off + tk.row0 + rloc)[(int64_t)col * lds], y);           // L*D (compute_trsm.c:108-109)
          pst<COH>(&Ap[(int64_t)col * lds], y * fast_rcp(dv));                       // L   (:110)
        }
      } else {
        if (rvalid && col < w) pst<COH>(&Ap[(int64_t)col * lds], acc_read<ct, q>());
      }
    };
    put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{});
    put(std::integral_constant<int, 2>{}); put(std::integral_constant<int, 3>{});
  });
}

// ---- the complex panel solve as a ticket (z LDLt / LDLh; kernels_z.hip k_trsm_zsy's algorithm) ----------------------------------
// Y = A L_d^-T (unit lower; HERM: L_d^-H) on split planes, then L = Y D^-1 into the L planes and Y = L D into the second
// arena's planes (compute_trsm.c:92-113).  A wave's eight column tiles are 2 x 64 registers (re, im): tiles 0-3 rest in the
// accumulation registers a[0:63] (re of tile p in slot 2p, im in 2p + 1), tiles 4-7 in LDS -- the update path's operand
// buffers, idle in a panel-solve ticket: 16 KB per wave, room for FOUR waves, hence 64 rows per complex ticket (waves 4-7
// of the workgroup sit it out).  A complex product is four real MFMAs.
template <bool HERM, bool COH>
__device__ __forceinline__ void trsm_zsy_parked(const Arenas& ar, double* __restrict__ lds, const TrsmTask& tk,
                                                const double* __restrict__ dinv_ws, const int tid) {
  constexpr int NT = 8;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int ld = tk.stride, w = tk.width;
  const int rloc = wave * 16 + l15;
  if (wave >= 4 || wave * 16 >= tk.nrows) return;
  const bool rvalid = rloc < tk.nrows;
  const int64_t xo = tk.off + tk.row0 + min(rloc, tk.nrows - 1);
  const double* Tr = ar.p[0] + tk.off;
  const double* Tim = ar.p[2] + tk.off;
  const double* Ti = dinv_ws + tk.dinv_off;
  double* lw = lds + wave * 2048 + lane;             // this lane's column of the wave's 16 KB: [tile - 4][plane][q][lane]
  auto park = [&](auto CT, const d4_t& re, const d4_t& im, auto FRESH) {
    constexpr int ct = decltype(CT)::value;
    constexpr bool fresh = decltype(FRESH)::value;
    if constexpr (ct < 4) {
      acc_write<2 * ct, 0, fresh>(re[0]); acc_write<2 * ct, 1>(re[1]); acc_write<2 * ct, 2>(re[2]); acc_write<2 * ct, 3>(re[3]);
      acc_write<2 * ct + 1, 0>(im[0]); acc_write<2 * ct + 1, 1>(im[1]); acc_write<2 * ct + 1, 2>(im[2]); acc_write<2 * ct + 1, 3>(im[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) { lw[((ct - 4) * 2 + 0) * 256 + q * 64] = re[q]; lw[((ct - 4) * 2 + 1) * 256 + q * 64] = im[q]; }
    }
  };
  auto fetch = [&](auto PP, auto QQ, double& re, double& im) {
    constexpr int p = decltype(PP)::value, q = decltype(QQ)::value;
    if constexpr (p < 4) { re = acc_read_m<2 * p, q>(); im = acc_read_m<2 * p + 1, q>(); }
    else { re = lw[((p - 4) * 2 + 0) * 256 + q * 64]; im = lw[((p - 4) * 2 + 1) * 256 + q * 64]; }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  static_for<NT>([&](auto CT) {
    constexpr int ct = decltype(CT)::value;
    d4_t vr, vi;
    int lds2 = ld;
    asm volatile("" : "+s"(lds2));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int col = ct * 16 + g + 4 * q;
      const int64_t o = xo + (int64_t)min(col, w - 1) * lds2;
      const double a = ar.p[0][o], b = ar.p[2][o];
      vr[q] = (rvalid && col < w) ? a : 0.0;
      vi[q] = (rvalid && col < w) ? b : 0.0;
    }
    park(CT, vr, vi, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
  });
  static_for<NT>([&](auto CT) {
    constexpr int ct = decltype(CT)::value;
    d4_t yr, yi;
    {
      double r0, i0, r1, i1, r2, i2, r3, i3;
      fetch(CT, I0{}, r0, i0); fetch(CT, I1{}, r1, i1); fetch(CT, I2{}, r2, i2); fetch(CT, I3{}, r3, i3);
      yr = d4_t{r0, r1, r2, r3}; yi = d4_t{i0, i1, i2, i3};
    }
    static_for<ct>([&](auto PP) {
      constexpr int p = decltype(PP)::value;
      int lds2 = ld, ws = w;
      asm volatile("" : "+s"(lds2), "+s"(ws));
      const int li = ct * 16 + l15, lic = min(li, ws - 1);
      double tr[4], tm[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int64_t o = lic + (int64_t)min(p * 16 + g + 4 * q, ws - 1) * lds2;
        const double a = Tr[o], b = Tim[o];
        tr[q] = (li < ws) ? a : 0.0;
        tm[q] = (li < ws) ? (HERM ? -b : b) : 0.0;
      }
      auto prod = [&](auto QQ) {                     // y[ct] -= t * y[p]
        constexpr int q = decltype(QQ)::value;
        double pr, pi;
        fetch(PP, QQ, pr, pi);
        yr = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr[q], pr, yr, 0, 0, 0);
        yr = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[q], pi, yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f64_16x16x4f64(-tr[q], pi, yi, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f64_16x16x4f64(-tm[q], pr, yi, 0, 0, 0);
      };
      prod(I0{}); prod(I1{}); prod(I2{}); prod(I3{});
      __builtin_amdgcn_sched_barrier(0);
    });
    d4_t nr = d4_t{0, 0, 0, 0}, ni = d4_t{0, 0, 0, 0};
    {
      int ws = w;
      asm volatile("" : "+s"(ws));
      const int cb = min(ct, ((ws + 15) >> 4) - 1);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double ar_ = Ti[cb * 512 + l15 + 16 * (g + 4 * q)];
        const double ai0 = Ti[cb * 512 + 256 + l15 + 16 * (g + 4 * q)];
        const double ai_ = HERM ? -ai0 : ai0;                  // (conj L)^-1 = conj(L^-1)
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yr[q], nr, 0, 0, 0);
        nr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai_, yi[q], nr, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ar_, yi[q], ni, 0, 0, 0);
        ni = __builtin_amdgcn_mfma_f64_16x16x4f64(ai_, yr[q], ni, 0, 0, 0);
      }
    }
    park(CT, nr, ni, std::true_type{});
    __builtin_amdgcn_sched_barrier(0);
  });
  const int64_t so = tk.off + tk.row0 + rloc;
  static_for<NT>([&](auto CT) {
    constexpr int ct = decltype(CT)::value;
    int lds2 = ld, gs = (tid & 63) >> 4;
    asm volatile("" : "+s"(lds2), "+v"(gs));
    auto put = [&](auto QQ) {
      constexpr int q = decltype(QQ)::value;
      const int col = ct * 16 + gs + 4 * q;
      const int64_t dd = (int64_t)min(col, w - 1) * (lds2 + 1);
      const double dr = Tr[dd], di = Tim[dd];
      // 1 / d as kernels_z.hip cinv (Smith)
      double ir, ii;
      if (fabs(dr) >= fabs(di)) { const double r = di / dr, dn = dr + di * r; ir = 1.0 / dn; ii = -r / dn; }
      else { const double r = dr / di, dn = dr * r + di; ir = r / dn; ii = -1.0 / dn; }
      double yre, yim;
      fetch(CT, QQ, yre, yim);
      if (rvalid && col < w) {
        const int64_t o = so + (int64_t)col * lds2;
        pst<COH>(&ar.p[1][o], yre);                 // L*D (compute_trsm.c:108-109)
        pst<COH>(&ar.p[3][o], yim);
        pst<COH>(&ar.p[0][o], yre * ir - yim * ii); // L   (:110)
        pst<COH>(&ar.p[2][o], yre * ii + yim * ir);
      }
    };
    put(I0{}); put(I1{}); put(I2{}); put(I3{});
  });
}

// ---- the run launch ------------------------------------------------------------------------------
// The update and panel-solve tasks of the thin levels at the top of the tree, all in ONE launch (plan.h RunInfo; the
// reference's tasks wait for TASK_CTRBCNT == 0 and are queued by the last contributor the same way, sopalin3d.c:790-1025 /
// sopalin_compute.c:958-985).  As many workgroups as the chip holds (two per CU), each looping: pop the next ready ticket
// (run_sync.h), run it exactly as k_update / k_trsm_llt would -- same pieces, same order, same arithmetic: the factors
// are bitwise those of the level-by-level schedule --, store the result write-through, tell the consumers.  Only ready
// tickets are ever held, so the launch needs no assumption about the order in which the hardware starts workgroups and
// cannot deadlock: a workgroup without a ticket holds nothing anybody waits for.  (One workgroup per ticket instead of
// the loop measured 11 % of every slot's time empty between a workgroup's end and its successor's first instruction.)
// FT: 0 LLt, 1 LDLt, 2 LU, 3 complex LDLt, 4 complex LDLh (which panel solve a panel-solve ticket runs)
// ONEK (real LLt / LDLt): the diagonal-blok tasks are tickets of this launch too (ring entries >= rc.nticket; plan.h
// RunCtl::onek) -- one kernel, one queue, no second kernel that has to be resident beside this one.  The blok's packed lower
// triangle takes the place of the operand buffers in LDS (66 of the 73.7 KB); diag_llt_body / diag_ldlt_body keep nothing
// in registers across their barriers, so they fit this kernel's 64 VGPRs as they are.  (LU and complex bloks are resident
// in REGISTERS -- 80 VGPRs per wave --: they keep the kernel of their own, k_run_diag_lu / k_run_diag_z.)
template <int FT, bool ONEK>
__global__ __launch_bounds__(64 * UW, UW / 2) void k_run_update(const Arenas ar, const Task* __restrict__ tasks,
                                                               const Piece* __restrict__ pieces,
                                                               const RunInfo* __restrict__ info,
                                                               const int32_t* __restrict__ cons, const RunCtl rc,
                                                               double* __restrict__ dinv, const long long limit,
                                                               const RunD* __restrict__ rd, const double critere,
                                                               long long* __restrict__ nbpivot, int* __restrict__ errflag) {
  extern __shared__ double sh_dyn[];
  double (&sh)[2][2][KC * SLD] = *reinterpret_cast<double (*)[2][2][KC * SLD]>(sh_dyn);   // [buffer][A|B]  73,728 bytes
  static_assert(sizeof(double) * (DIAG_LDS_DOUBLES + 320) <= sizeof(sh), "the diagonal blok must fit the operand buffers");
  int* tick = (int*)&sh[0][0][0];
  const int nring = ONEK ? rc.nticket + rc.nd : rc.nticket;
  if (!ONEK && threadIdx.x == 0) run_st(rc.ctl + RUN_GO, 1);       // (the resident diagonal workers' clocks start now)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar, for the whole launch)
  for (;;) {
    // (the thread index is made anew per ticket -- lane_now --: what is derived from it is recomputed, not kept in registers
    // across the loop)
    int lane = lane_now();
    int tid = (wave << 6) | lane;
    if (tid == 0) {
      // (the stamps of the developer profile are written here, at once: nothing of them is live across the ticket)
      const long long tdraw = rc.prof ? wall_clock64() : 0;
      const int tk0 = run_pop(rc.q, rc.ctl + RUN_HEAD, nring, rc.ctl + RUN_STUCK, limit, (int)gridDim.x - rc.room);
      *tick = tk0;
      run_acquire();
      if (rc.prof && tk0 >= 0) {
        rc.prof[4 * (int64_t)tk0] = tdraw;
        rc.prof[4 * (int64_t)tk0 + 1] = wall_clock64();
        unsigned hw, xcc;                        // which CU ran the ticket (tools/run_prof.py: idle time per CU)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        rc.prof[4 * (int64_t)tk0 + 3] = ((long long)(xcc & 0xf) << 32) | hw;
      }
    }
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(*tick);
    if (t < 0) return;                           // (every ticket is taken, or the run is stuck)
    if constexpr (ONEK && FT <= 1) {
      if (t >= rc.nticket) {
        // a diagonal-blok ticket: k_diag_llt_w / k_diag_ldlt_w's body on this workgroup, then the cblk's panel solves count down
        __syncthreads();                         // (the ticket word in LDS is dead from here on)
        __builtin_amdgcn_s_setprio(3);
        const int di = t - rc.nticket;
        const RunD d = rd[di];
        asm volatile(";;#PASTIX_AMD_DIAG_TICKET_BEGIN");   // (tests/test_kernel_audit.py: what lies between the two markers)
        double* const Dl = &sh[0][0][0];
        if constexpr (FT == 0) diag_llt_body<true>(Dl, Dl + DIAG_LDS_DOUBLES, ar.p[0], d.pt, dinv, critere, nbpivot, errflag, tid);
        else diag_ldlt_body<true>(Dl, Dl + DIAG_LDS_DOUBLES, ar.p[0], d.pt, dinv, critere, nbpivot, tid);
        run_drain();
        __syncthreads();
        if (wave == 0) {
          for (int i = lane; i < d.tn; i += 64) run_dec_ticket(rc, info, d.t0 + i);
          if (rc.prof && lane == 0) rc.prof[4 * (int64_t)t + 2] = wall_clock64();
        }
        asm volatile(";;#PASTIX_AMD_DIAG_TICKET_END");
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        continue;
      }
    }
    const Task tk = tasks[t];
    const RunInfo ri = info[t];
    __syncthreads();                             // (the ticket word in LDS is dead from here on)
    if (ri.kind & 4) {
      // a panel-solve ticket (the Task record holds a TrsmTask): 128 panel rows, a wave per 16
      TrsmTask tt;
      __builtin_memcpy(&tt, &tk, sizeof(tt));
      if constexpr (FT == 0) trsm_parked<0, true>(ar.p[0], ar.p[1], tt, dinv, tid, &sh[0][0][0]);
      else if constexpr (FT == 1) trsm_parked<1, true>(ar.p[0], ar.p[1], tt, dinv, tid, &sh[0][0][0]);
      else if constexpr (FT == 2) {
        trsm_parked<2, true>(ar.p[0], ar.p[1], tt, dinv, tid);
        const int tid2 = tid_now(wave);            // (made again: the two solves must not share hoisted index arithmetic)
        trsm_parked<3, true>(ar.p[0], ar.p[1], tt, dinv, tid2, &sh[0][0][0]);
      } else {
        trsm_zsy_parked<FT == 4, true>(ar, &sh[0][0][0], tt, dinv, tid);
      }
      run_drain();
      __syncthreads();
      if (wave == 0) {
        for (int i = lane; i < ri.cn; i += 64) run_dec_ticket(rc, info, cons[ri.cptr + i]);
        if (rc.prof && lane == 0) rc.prof[4 * (int64_t)t + 2] = wall_clock64();
      }
      __syncthreads();
      continue;
    }
    acc_zero();
    const unsigned touched = update_pieces(sh, ar, tk, pieces, wave);
    acc_settle();
    lane = lane_now();                           // (again, behind the loops)
    tid = (wave << 6) | lane;
    const int row0 = (wave >> 1) * 16, col0 = (wave & 1) * 16;
    const int l15 = lane & 15, g = lane >> 4;
    double* C = ar.p[tk.flags & 3] + tk.c_off;
    const int tm1 = (int)tk.tm - 1, tn1 = (int)tk.tn - 1;
    epilogue_band<0, false, true>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
    epilogue_band<1, false, true>(C, touched, row0, col0, l15, g, tm1, tn1, tk.ldc);
    run_drain();
    __syncthreads();
    if (tid == 0) {
      if (ri.succ >= 0) { for (int z = 0; z < ri.cn; z++) run_dec_ticket(rc, info, ri.succ + z); }
      else if (ri.succ <= -2) run_dec_diag(rc, -2 - ri.succ);
      if (rc.prof) rc.prof[4 * (int64_t)t + 2] = wall_clock64();
    }
    __syncthreads();
  }
}

// The operand buffers are DYNAMIC LDS: the compiler then does not know the 73.7 KB that cap the kernels at four waves per
// SIMD, and the register budget of the kernels can be stated as what it is (Makefile: 64 VGPRs, no AGPR of the compiler's).
constexpr unsigned UPDATE_LDS_BYTES = 2 * 2 * KC * SLD * sizeof(double);
// (once per kernel and device, keyed on the kernel's ADDRESS: the instances of a template share one function-pointer type)
static void allow_lds_addr(const void* kernel) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  if (!done.insert({dev, kernel}).second) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)UPDATE_LDS_BYTES);
}
template <class K>
static void allow_lds(K kernel) { allow_lds_addr((const void*)kernel); }

void launch_run_update(hipStream_t s, int factotype, const Arenas& ar, const Task* tasks, const Piece* pieces, const RunInfo* info,
                       const int32_t* cons, const RunCtl& rc, double* dinv, int64_t ntasks, int nwg, long long limit, const RunD* rd,
                       double critere, long long* nbpivot, int* errflag) {
  if (ntasks <= 0) return;
  const dim3 g((unsigned)std::min<int64_t>(ntasks + (rc.onek ? rc.nd : 0), std::max(nwg, 1))), b(64 * UW);
#define PA_RUN(FT, ONEK)                                                                                                  \
  do {                                                                                                                    \
    allow_lds(k_run_update<FT, ONEK>);                                                                                    \
    hipLaunchKernelGGL((k_run_update<FT, ONEK>), g, b, UPDATE_LDS_BYTES, s, ar, tasks, pieces, info, cons, rc, dinv, limit, rd, \
                       critere, nbpivot, errflag);                                                                        \
  } while (0)
  if (ar.p[2]) {                                   // complex double (split planes)
    if (factotype == PASTIX_AMD_FACT_LDLH) PA_RUN(4, false);
    else PA_RUN(3, false);
  } else if (factotype == PASTIX_AMD_FACT_LLT) { if (rc.onek) PA_RUN(0, true); else PA_RUN(0, false); }
  else if (factotype == PASTIX_AMD_FACT_LDLT) { if (rc.onek) PA_RUN(1, true); else PA_RUN(1, false); }
  else PA_RUN(2, false);
#undef PA_RUN
}

void launch_update(hipStream_t s, const Arenas& ar, const Task* tasks, const Piece* pieces, int64_t ntasks,
                   bool urgent) {
  if (ntasks <= 0) return;
  const dim3 g((unsigned)ntasks);
  if (urgent) {
    allow_lds(k_update<1>);
    hipLaunchKernelGGL((k_update<1>), g, dim3(64 * UW), UPDATE_LDS_BYTES, s, ar, tasks, pieces);
  } else {
    allow_lds(k_update<0>);
    hipLaunchKernelGGL((k_update<0>), g, dim3(64 * UW), UPDATE_LDS_BYTES, s, ar, tasks, pieces);
  }
}

}  // namespace pastix_amd
