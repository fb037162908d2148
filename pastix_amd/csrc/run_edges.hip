// run_edges.hip -- gfx950: the reader lists of the run schedule, built ON THE DEVICE (round 6).
//
// The run launch (plan.h RunInfo) is counter-driven like the reference's engine (TASK_CTRBCNT, sopalin3d.c:790-1025;
// who contributes to whom comes from indtab there: solverMatrixGen.c:667-760).  What it needs beside the tile chains is,
// for every 128-row tile of a source panel of the run's levels, the list of the update tickets that READ it (they may
// start once its panel solve is done), and for every update ticket the number of such inputs.  At 200^3 that is 366 M
// (tile, ticket) pairs: built on host threads (plan.cpp "run: source tiles / consumer lists") it costs 2.5 s of analysis and
// was the reason the run was off by default at that size.  The tables the pairs come from -- the tickets and their
// pieces -- are on the device anyway, and "group pairs by tile" is a sort: one kernel lists every ticket's source tiles,
// rocPRIM sorts the 64-bit keys (tile << 32 | ticket) and drops the duplicates, two small kernels turn the sorted keys
// into the reader lists (already ordered by ticket: the tables do not depend on timing), their offsets in the panel-solve
// tickets' records and the readers' counters.  HBM-bound integer work: a few passes over ~4 GB.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "engine.h"

namespace pastix_amd {

namespace {

struct EdgeTabs {
  const int64_t* poff;        // [nc + 1] panel offsets
  const int32_t* level;       // [nc]
  const int32_t* tile_base;   // [nc + 1] first tile of a cblk's panel
  int32_t nc, L0;
};

// the source cblk of an arena offset: the last k with poff[k] <= off
__device__ __forceinline__ int32_t cblk_of(const int64_t* __restrict__ poff, const int32_t nc, const int64_t off) {
  int32_t lo = 0, hi = nc;            // poff[lo] <= off < poff[hi]
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (poff[mid] <= off) lo = mid; else hi = mid;
  }
  return lo;
}

// One wave per ticket, four lanes per piece (16 pieces per pass): the 128-row source tiles the piece reads -- first and last
// tile of its A rows, first and last of its B rows (a piece has at most 128 rows of either: two tiles) --, sources of the
// run's levels only (older panels are final when the run starts).  WRITE = false counts the candidates, WRITE = true
// stores the keys at the ticket's offset (exclusive scan of the counts).  Duplicates stay: the sort removes them.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_run_src_tiles(const Task* __restrict__ tasks, const RunInfo* __restrict__ info,
                                                       const Piece* __restrict__ pieces, const int64_t nr, const EdgeTabs T,
                                                       uint32_t* __restrict__ count, const uint64_t* __restrict__ offset,
                                                       uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= nr) return;
  if (info[i].kind & 4) {                       // a panel-solve ticket reads its own panel only
    if (!WRITE && lane == 0) count[i] = 0;
    return;
  }
  const Task tk = tasks[i];
  uint32_t total = 0;
  uint64_t base = WRITE ? offset[i] : 0;
  for (int z0 = 0; z0 < tk.pn; z0 += 16) {
    const int z = z0 + (lane >> 2), sub = lane & 3;
    int64_t tile = -1;
    if (z < tk.pn) {
      const Piece pc = pieces[(int64_t)tk.p0 + z];
      const int32_t k = cblk_of(T.poff, T.nc, pc.a_off);
      if (T.level[k] >= T.L0) {
        const int64_t o = ((sub & 2) ? pc.b_off : pc.a_off) - T.poff[k];     // row of the operand's first entry (column 0)
        const int ext = (sub & 2) ? (int)pc.n : (int)pc.m;
        const int64_t r0 = o / TM, r1 = (o + ext - 1) / TM;
        if (!(sub & 1)) tile = T.tile_base[k] + r0;
        else if (r1 != r0) tile = T.tile_base[k] + r1;
      }
    }
    const unsigned long long bal = __ballot(tile >= 0);
    if (WRITE && tile >= 0) {
      const unsigned rank = __popcll(bal & ((1ull << lane) - 1ull));
      keys[base + rank] = ((uint64_t)tile << 32) | (uint32_t)i;
    }
    const uint32_t c = (uint32_t)__popcll(bal);
    total += c;
    base += c;
  }
  if (!WRITE && lane == 0) count[i] = total;
}

// sorted unique keys -> reader lists: cons[j] = the ticket of key j; first / last key of every tile; the readers' counters
// (a tile has one panel-solve ticket, a complex one up to two: each is an input of its own)
__global__ void k_run_edges_fill(const uint64_t* __restrict__ keys, const int64_t m, const int32_t* __restrict__ tile_ticket,
                                 const uint8_t* __restrict__ tile_nt, int32_t* __restrict__ cons, int32_t* __restrict__ first,
                                 int32_t* __restrict__ last, int32_t* __restrict__ dep, int* __restrict__ bad) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const uint64_t key = keys[j];
  const int32_t tile = (int32_t)(key >> 32), ticket = (int32_t)(key & 0xffffffffu);
  cons[j] = ticket;
  const int32_t pt = tile_ticket[tile];
  // (a source tile of the run has its panel-solve ticket, in front of its readers in the ticket order)
  if (pt < 0 || pt >= ticket) { *bad = 1; return; }
  atomicAdd(&dep[ticket], (int32_t)tile_nt[tile]);
  if (j == 0 || (int32_t)(keys[j - 1] >> 32) != tile) first[tile] = (int32_t)j;
  if (j == m - 1 || (int32_t)(keys[j + 1] >> 32) != tile) last[tile] = (int32_t)j;
}

__global__ void k_run_edges_lists(const int64_t ntile, const int32_t* __restrict__ tile_ticket, const uint8_t* __restrict__ tile_nt,
                                  const int32_t* __restrict__ first, const int32_t* __restrict__ last, RunInfo* __restrict__ info) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntile) return;
  const int32_t pt = tile_ticket[t];
  if (pt < 0) return;
  const int32_t f = first[t];
  for (int z = 0; z < (int)tile_nt[t]; z++) {
    info[pt + z].cptr = f >= 0 ? f : 0;
    info[pt + z].cn = f >= 0 ? last[t] - f + 1 : 0;
  }
}

__global__ void k_ring_scatter(int32_t* __restrict__ ring, const int32_t* __restrict__ vals, const int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ring[i * RUN_SLOT] = vals[i];
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <class T> T* as() { return (T*)p; }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

}  // namespace

// the tasks that are ready when the run starts, into their ring slots (one 128-byte line each, plan.h RUN_SLOT)
void launch_ring_scatter(hipStream_t s, int32_t* ring, const int32_t* vals, size_t n) {
  if (n) hipLaunchKernelGGL(k_ring_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ring, vals, (int64_t)n);
}

#define RE_CHK(x)                                                                                              \
  do {                                                                                                         \
    hipError_t e_ = (x);                                                                                       \
    if (e_ != hipSuccess) {                                                                                    \
      fprintf(stderr, "pastix_amd: run_edges: %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
      (void)hipGetLastError();                                                                                 \
      return e_ == hipErrorOutOfMemory ? PASTIX_AMD_ERR_ALLOC : PASTIX_AMD_ERR_DEVICE;                          \
    }                                                                                                          \
  } while (0)

// Device bytes the builder needs beside its result (the caller checks them against what is free): two key buffers, the
// sort's own storage (about one more), the small tables.
size_t run_edges_device_bytes(const Plan& H, size_t nr, size_t npieces_run) {
  const size_t nkeys = 4 * npieces_run;
  return 3 * nkeys * sizeof(uint64_t) + nkeys * sizeof(int32_t) + nr * 16 + (size_t)H.ntile * 16 + (size_t)H.cblknbr * 24 + ((size_t)64 << 20);
}

// Builds the reader lists of the run on the device.  In: the tickets, their records and the pieces as uploaded; H.run_dep
// holds the chain inputs (plan.cpp "run: tile chains").  Out: *cons_out (device, the lists back to back, each ordered by
// ticket), the panel-solve tickets' cptr / cn in `info`, H.run_dep with the source inputs added.
int run_edges_device(hipStream_t s, const Task* dRunTasks, RunInfo* dRunInfo, const Piece* dPieces, Plan& H,
                     int32_t** cons_out, size_t* ncons_out) {
  const size_t nr = H.run_dep.size() - H.run_d.size();
  const int64_t nc = H.cblknbr, ntile = H.ntile;
  *cons_out = nullptr;
  *ncons_out = 0;
  if (nr == 0 || (size_t)ntile != H.run_tile_ticket.size() || (size_t)ntile != H.run_tile_nt.size() || (size_t)nc + 1 != H.run_tile_base.size())
    return PASTIX_AMD_ERR_BADPARAMETER;
  DevBuf poff, level, tbase, tticket, tnt, count, offs, keys, keys2, tmp, first, last, dep, flags;
  RE_CHK(poff.alloc((nc + 1) * sizeof(int64_t)));
  RE_CHK(level.alloc(nc * sizeof(int32_t)));
  RE_CHK(tbase.alloc((nc + 1) * sizeof(int32_t)));
  RE_CHK(tticket.alloc(ntile * sizeof(int32_t)));
  RE_CHK(tnt.alloc(ntile));
  RE_CHK(count.alloc(nr * sizeof(uint32_t)));
  RE_CHK(offs.alloc(nr * sizeof(uint64_t)));
  RE_CHK(dep.alloc(nr * sizeof(int32_t)));
  RE_CHK(flags.alloc(2 * sizeof(uint64_t)));
  RE_CHK(hipMemcpyAsync(poff.p, H.poff.data(), (nc + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
  RE_CHK(hipMemcpyAsync(level.p, H.level.data(), nc * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RE_CHK(hipMemcpyAsync(tbase.p, H.run_tile_base.data(), (nc + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RE_CHK(hipMemcpyAsync(tticket.p, H.run_tile_ticket.data(), ntile * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RE_CHK(hipMemcpyAsync(tnt.p, H.run_tile_nt.data(), ntile, hipMemcpyHostToDevice, s));
  RE_CHK(hipMemcpyAsync(dep.p, H.run_dep.data(), nr * sizeof(int32_t), hipMemcpyHostToDevice, s));
  const EdgeTabs T{poff.as<int64_t>(), level.as<int32_t>(), tbase.as<int32_t>(), (int32_t)nc, H.run_L0};
  const unsigned gw = (unsigned)((nr + 3) / 4);
  hipLaunchKernelGGL((k_run_src_tiles<false>), dim3(gw), dim3(256), 0, s, dRunTasks, dRunInfo, dPieces, (int64_t)nr, T,
                     count.as<uint32_t>(), (const uint64_t*)nullptr, (uint64_t*)nullptr);
  // offsets = exclusive scan of the counts (64-bit: 4 x pieces may pass 2^32), total = offset of the end
  size_t tb = 0;
  auto cin = rocprim::make_transform_iterator(count.as<uint32_t>(), [] __device__(uint32_t c) { return (uint64_t)c; });
  RE_CHK(rocprim::exclusive_scan(nullptr, tb, cin, offs.as<uint64_t>(), (uint64_t)0, nr, rocprim::plus<uint64_t>(), s));
  RE_CHK(tmp.alloc(tb));
  RE_CHK(rocprim::exclusive_scan(tmp.p, tb, cin, offs.as<uint64_t>(), (uint64_t)0, nr, rocprim::plus<uint64_t>(), s));
  uint64_t lastoff = 0;
  uint32_t lastcnt = 0;
  RE_CHK(hipMemcpyAsync(&lastoff, offs.as<uint64_t>() + (nr - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  RE_CHK(hipMemcpyAsync(&lastcnt, count.as<uint32_t>() + (nr - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RE_CHK(hipStreamSynchronize(s));
  const size_t nkeys = (size_t)(lastoff + lastcnt);
  RE_CHK(first.alloc(ntile * sizeof(int32_t)));
  RE_CHK(last.alloc(ntile * sizeof(int32_t)));
  RE_CHK(hipMemsetAsync(first.p, 0xff, ntile * sizeof(int32_t), s));
  RE_CHK(hipMemsetAsync(last.p, 0xff, ntile * sizeof(int32_t), s));
  size_t m = 0;
  int32_t* cons = nullptr;
  if (nkeys > 0) {
    RE_CHK(keys.alloc(nkeys * sizeof(uint64_t)));
    RE_CHK(keys2.alloc(nkeys * sizeof(uint64_t)));
    hipLaunchKernelGGL((k_run_src_tiles<true>), dim3(gw), dim3(256), 0, s, dRunTasks, dRunInfo, dPieces, (int64_t)nr, T,
                       (uint32_t*)nullptr, (const uint64_t*)offs.p, keys.as<uint64_t>());
    unsigned tbits = 1;
    while (((int64_t)1 << tbits) < ntile) tbits++;
    size_t sb = 0;
    RE_CHK(rocprim::radix_sort_keys(nullptr, sb, keys.as<uint64_t>(), keys2.as<uint64_t>(), nkeys, 0u, 32u + tbits, s));
    DevBuf stmp;
    RE_CHK(stmp.alloc(sb));
    RE_CHK(rocprim::radix_sort_keys(stmp.p, sb, keys.as<uint64_t>(), keys2.as<uint64_t>(), nkeys, 0u, 32u + tbits, s));
    // unique: keys2 -> keys, the count in flags[0]
    size_t ub = 0;
    RE_CHK(rocprim::unique(nullptr, ub, keys2.as<uint64_t>(), keys.as<uint64_t>(), flags.as<uint64_t>(), nkeys,
                           rocprim::equal_to<uint64_t>(), s));
    DevBuf utmp;
    RE_CHK(utmp.alloc(ub));
    RE_CHK(rocprim::unique(utmp.p, ub, keys2.as<uint64_t>(), keys.as<uint64_t>(), flags.as<uint64_t>(), nkeys,
                           rocprim::equal_to<uint64_t>(), s));
    uint64_t mu = 0;
    RE_CHK(hipMemcpyAsync(&mu, flags.p, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    RE_CHK(hipStreamSynchronize(s));
    m = (size_t)mu;
    if (m > 0x7fffffffULL) return PASTIX_AMD_ERR_UNSUPPORTED;
    RE_CHK(hipMalloc((void**)&cons, std::max<size_t>(m, 1) * sizeof(int32_t)));
    int* bad = (int*)(flags.as<uint64_t>() + 1);
    RE_CHK(hipMemsetAsync(bad, 0, sizeof(int), s));
    if (m > 0)
      hipLaunchKernelGGL(k_run_edges_fill, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const uint64_t*)keys.p, (int64_t)m,
                         (const int32_t*)tticket.p, (const uint8_t*)tnt.p, cons, first.as<int32_t>(), last.as<int32_t>(),
                         dep.as<int32_t>(), bad);
    int hbad = 0;
    RE_CHK(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
    RE_CHK(hipStreamSynchronize(s));
    if (hbad) { (void)hipFree(cons); return PASTIX_AMD_ERR_LAYOUT; }
  } else {
    RE_CHK(hipMalloc((void**)&cons, sizeof(int32_t)));
  }
  hipLaunchKernelGGL(k_run_edges_lists, dim3((unsigned)((ntile + 255) / 256)), dim3(256), 0, s, ntile, (const int32_t*)tticket.p,
                     (const uint8_t*)tnt.p, (const int32_t*)first.p, (const int32_t*)last.p, dRunInfo);
  RE_CHK(hipMemcpyAsync(H.run_dep.data(), dep.p, nr * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  RE_CHK(hipStreamSynchronize(s));
  RE_CHK(hipGetLastError());
  *cons_out = cons;
  *ncons_out = m;
  return PASTIX_AMD_OK;
}

}  // namespace pastix_amd
