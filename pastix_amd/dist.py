"""Multi-GPU factorization: elimination-tree subtrees mapped one per GPU, fan-in of aggregated
contributions over point-to-point messages.

Reference model: proportional mapping + fan-in of PaStiX (`FanInTarget`, src/blend/src/ftgt.h:67-113;
`add_contrib_target`, src/sopalin/src/sopalin_compute.c:600-733: local contributions are SUBTRACTED into a
zero-initialised buffer; `recv_handle_fanin`, src/sopalin/src/sopalin_sendrecv.c:384-389: the owner ADDS
the received block).  Here every rank holds one plan over the same layout
(`pastix_amd_plan_create_dist`): its owned panels plus "shadow" panels for the remote cblks it
contributes to.  All ranks walk the dependency levels in lockstep:

    for l in levels:
        update(l)                      # contributions scheduled into slot l (into owned or shadow panels)
        exchange shadows of the cblks of level l  (one message per (sender, cblk), RCCL send/recv)
        owner adds the received blocks
        panels(l)                      # diagonal factor + panel solve of the owned cblks of level l

There is no collective on the data path.  The orchestration is engine-agnostic (GPU engine below; the
CPU tests drive the same code with a numpy engine over gloo).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import LayoutArrays, Options, Stats, check


# ------------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------------
def cblk_flops(cblk4, blok4):
    """Per-cblk share of DPARM_FACT_FLOPS for LLt (blend_symbol_cost.c:382-430)."""
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    N = (c4[:-1, 1] - c4[:-1, 0] + 1).astype(np.float64)
    S = c4[:-1, 3].astype(np.float64)
    M = S - N
    fl = N * (((1. / 6.) * N + 0.5) * N + (1. / 3.)) + N * (((1. / 6.) * N) * N - (1. / 6.)) + M * N * (N + 1.)
    owner_of_blok = np.repeat(np.arange(nc), np.diff(c4[:, 2]))
    h = (b4[:, 1] - b4[:, 0] + 1).astype(np.float64)
    offd = b4[:, 3] > 0                         # off-diagonal bloks (coefind > 0)
    rem = S[owner_of_blok] - b4[:, 3]
    g = 2.0 * rem * h * N[owner_of_blok] * offd
    fl += np.bincount(owner_of_blok, weights=g, minlength=nc)
    return fl


def _etree(c4, b4):
    nc = len(c4) - 1
    nb = np.diff(c4[:, 2])
    parent = np.full(nc, -1, dtype=np.int64)
    has = nb > 1
    parent[has] = b4[c4[:-1, 2][has] + 1, 2]       # facing cblk of the first off-diagonal blok
    return parent


def partition(cblk4, blok4, world, split=6, light=0.05):
    """owner[k] for every cblk by proportional mapping (the idea of PaStiX's blend, splitpart.c:752-1012, on the
    cblk elimination tree): a subtree is given a SET of ranks; the chain of cblks at its top (the split cblks of
    one separator) is dealt over that set, longest first onto the least loaded rank; where the tree branches the
    set is divided among the heavy children in proportion to their work; a subtree with one rank goes to it
    whole.  Light side subtrees (< `light` of their parent's work) go whole to the least loaded rank of the set.
    Fan-in traffic therefore stays inside the rank set of the enclosing subtree: a rank only contributes to
    separators on its own path to the root.  `split` is kept for callers of the earlier interface (unused)."""
    del split
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    owner = np.full(nc, -1, dtype=np.int32)
    if world <= 1:
        owner[:] = 0
        return owner
    fl = cblk_flops(c4, b4)
    parent = _etree(c4, b4)
    sub = fl.copy()
    kids = [[] for _ in range(nc)]
    for k in range(nc):                           # children have smaller indices than their parents
        q = parent[k]
        if q >= 0:
            sub[q] += sub[k]
            kids[q].append(k)
    load = np.zeros(world)
    whole = []                                    # (subtree root, rank): everything below goes to the rank

    def give_whole(root, ranks):
        q = min(ranks, key=lambda r_: load[r_])
        whole.append((root, q))
        load[q] += sub[root]

    def split_ranks(ranks, weights):
        """Divide the rank list among len(weights) <= len(ranks) children, proportionally, at least one each."""
        m, tot = len(ranks), float(sum(weights))
        cnt = [max(1, int(round(m * w_ / tot))) for w_ in weights]
        while sum(cnt) > m:
            i = max(range(len(cnt)), key=lambda j: (cnt[j] > 1, cnt[j] - m * weights[j] / tot))
            cnt[i] -= 1
        while sum(cnt) < m:
            i = max(range(len(cnt)), key=lambda j: m * weights[j] / tot - cnt[j])
            cnt[i] += 1
        out, pos = [], 0
        for c_ in cnt:
            out.append(ranks[pos:pos + c_])
            pos += c_
        return out

    stack = [(r_, list(range(world))) for r_ in range(nc) if parent[r_] < 0]
    if len(stack) > 1:                            # a forest: treat the roots as children of a virtual node
        roots = sorted((r_ for r_, _ in stack), key=lambda r_: -sub[r_])
        stack = []
        heavy = roots[:world]
        for r_, rk in zip(heavy, split_ranks(list(range(world)), [sub[r_] for r_ in heavy])):
            stack.append((r_, rk))
        for r_ in roots[world:]:
            give_whole(r_, list(range(world)))
    while stack:
        node, ranks = stack.pop()
        if len(ranks) == 1:
            whole.append((node, ranks[0]))
            load[ranks[0]] += sub[node]
            continue
        chain = []
        while True:                               # walk down the separator chain to the branching point
            chain.append(node)
            ch = sorted(kids[node], key=lambda c_: -sub[c_])
            heavy = [c_ for c_ in ch if sub[c_] >= light * sub[node]]
            for c_ in ch[len(heavy):]:
                give_whole(c_, ranks)
            if len(heavy) != 1:
                break
            node = heavy[0]
        for k in sorted(chain, key=lambda c_: -fl[c_]):
            q = min(ranks, key=lambda r_: load[r_])
            owner[k] = q
            load[q] += fl[k]
        if not heavy:
            continue
        if len(heavy) > len(ranks):               # more heavy children than ranks: the lightest go whole
            for c_ in heavy[len(ranks):]:
                give_whole(c_, ranks)
            heavy = heavy[:len(ranks)]
        for c_, rk in zip(heavy, split_ranks(ranks, [sub[c_] for c_ in heavy])):
            stack.append((c_, rk))
    for root, q in whole:
        owner[root] = q
    for k in range(nc - 1, -1, -1):               # parents have larger indices than their children
        if owner[k] < 0:
            owner[k] = owner[parent[k]]
    assert (owner >= 0).all()
    return owner


def levels_of(cblk4, blok4):
    """Dependency level of every cblk (same rule as plan.cpp)."""
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    level = np.zeros(nc, dtype=np.int32)
    src = np.repeat(np.arange(nc), np.diff(c4[:, 2]))
    offd = b4[:, 3] > 0
    s, t = src[offd], b4[offd, 2]
    for k, f in zip(s.tolist(), t.tolist()):     # sources in increasing order => one pass suffices
        if level[f] < level[k] + 1:
            level[f] = level[k] + 1
    return level


def fanin_pairs(cblk4, blok4, owner):
    """All (sender rank, cblk) pairs: rank r owns a cblk with a blok facing cblk t owned by another rank."""
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    nc = len(c4) - 1
    src = np.repeat(np.arange(nc), np.diff(c4[:, 2]))
    offd = b4[:, 3] > 0
    r = owner[src[offd]].astype(np.int64)
    t = b4[offd, 2]
    m = r != owner[t]
    return np.unique(np.stack([r[m], t[m]], axis=1), axis=0)


def fanin_touched(cblk4, blok4, owner):
    """uint64 mask per blok: bit r set when rank r contributes into that blok of a cblk it does not own
    (pastix_amd_fanin_touched; the reference's FanInTarget regions at blok granularity)."""
    la = LayoutArrays(cblk4, blok4)
    own = np.ascontiguousarray(owner, dtype=np.int32)
    mask = np.zeros(len(np.asarray(blok4)), dtype=np.uint64)
    check(_lib.lib().pastix_amd_fanin_touched(ctypes.byref(la.c), _lib.ptr(own), _lib.ptr(mask)),
          "pastix_amd_fanin_touched")
    return mask


def fanin_rows(cblk4, blok4, mask, src, t):
    """Rows (0-based, inside the full panel of cblk t) of the compact block rank `src` sends for cblk t."""
    c4, b4 = np.asarray(cblk4, dtype=np.int64), np.asarray(blok4, dtype=np.int64)
    fb, lb = int(c4[t, 2]), int(c4[t + 1, 2])
    sel = np.nonzero((mask[fb:lb] >> np.uint64(src)) & np.uint64(1))[0] + fb
    if len(sel) == 0:
        return np.zeros(0, dtype=np.int32)
    h = (b4[sel, 1] - b4[sel, 0] + 1).astype(np.int64)
    start = np.repeat(b4[sel, 3], h)
    within = np.arange(int(h.sum()), dtype=np.int64) - np.repeat(np.cumsum(h) - h, h)
    return (start + within).astype(np.int32)


def plan_profile(cblk4, blok4, owner, rank, chunk=0, maxlevels=100000):
    """Host-only schedule statistics of one rank: (slot_flops, slot_maxwork, slot_tasks, panel_flops, slot_urgent_flops)."""
    la = LayoutArrays(cblk4, blok4)
    opts = Options()
    opts.lookahead = chunk
    sf = np.zeros(maxlevels)
    sm = np.zeros(maxlevels)
    stn = np.zeros(maxlevels, dtype=np.int64)
    pf = np.zeros(maxlevels)
    uf = np.zeros(maxlevels)
    nl = ctypes.c_int64(0)
    own = np.ascontiguousarray(owner, dtype=np.int32) if owner is not None else None
    check(_lib.lib().pastix_amd_plan_profile(ctypes.byref(la.c), 0, ctypes.byref(opts), _lib.ptr(own),
                                             ctypes.c_int32(rank), ctypes.c_int64(maxlevels), _lib.ptr(sf), _lib.ptr(sm),
                                             _lib.ptr(stn), _lib.ptr(pf), ctypes.byref(nl), _lib.ptr(uf)),
          "pastix_amd_plan_profile")
    n = nl.value
    return sf[:n], sm[:n], stn[:n], pf[:n], uf[:n]


class Exchange:
    """Per-level send / receive lists of one rank (deterministic order on every rank)."""

    def __init__(self, cblk4, blok4, owner, level, rank):
        pairs = fanin_pairs(cblk4, blok4, owner)
        nlev = int(level.max()) + 1
        self.sends = [[] for _ in range(nlev)]      # (cblk, destination rank)
        self.recvs = [[] for _ in range(nlev)]      # (cblk, source rank)
        for r, t in pairs.tolist():
            if r == rank:
                self.sends[level[t]].append((t, int(owner[t])))
            elif owner[t] == rank:
                self.recvs[level[t]].append((t, r))
        self.nlevels = nlev


def factorize_levels(engine, exch, transport):
    """The lockstep level loop.  engine: update(l), panels(l), panel(k) -> 1-D view of what this rank holds for
    cblk k (for a remote cblk: its fan-in buffer), recv_numel(k, src) -> elements rank src sends for cblk k,
    add(k, buf, src).  transport.exchange(sends=[(view, dst)], recvs=[(cblk, src, numel)]) -> list of received 1-D
    buffers (same order as recvs)."""
    for l in range(exch.nlevels):
        engine.update(l)
        if exch.sends[l] or exch.recvs[l]:
            sends = [(engine.panel(t), dst) for t, dst in exch.sends[l]]
            recvs = [(t, src, engine.recv_numel(t, src)) for t, src in exch.recvs[l]]
            bufs = transport.exchange(sends, recvs)
            for (t, src, _n), buf in zip(recvs, bufs):
                engine.add(t, buf, src)            # recv_handle_fanin: owner ADDS the aggregated block
        engine.panels(l)


class TorchTransport:
    """RCCL / gloo point-to-point through torch.distributed (one grouped batch per level).

    A level's batch is waited for only by a rank that RECEIVES in it: a pure sender's later work does not depend
    on the send (its fan-in buffer is not written again before the next refill), so it runs ahead instead of
    idling until the owner has posted its receives.  `drain()` retires the pending sends; call it before the
    buffers are reused (end of a factorization)."""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = device
        self._pending = []

    def exchange(self, sends, recvs):
        import torch.distributed as dist
        torch = self.torch
        # gloo moves host memory only: stage through the host (CPU tests, 1-GPU validation runs)
        staged = dist.get_backend() == "gloo" and self.device.type != "cpu"
        dev = "cpu" if staged else self.device
        ops, bufs, keep = [], [], []
        for t, src, n in recvs:
            b = torch.empty(n, dtype=torch.float64, device=dev)
            bufs.append(b)
            ops.append(dist.P2POp(dist.irecv, b, src))
        for view, dst in sends:
            v = view.cpu() if staged else view
            keep.append(v)
            ops.append(dist.P2POp(dist.isend, v, dst))
        if ops:
            works = dist.batch_isend_irecv(ops)
            if recvs:
                for w in works:
                    w.wait()
            else:
                self._pending.append((works, keep))
        return [b.to(self.device) for b in bufs] if staged else bufs

    def drain(self):
        for works, _keep in self._pending:
            for w in works:
                w.wait()
        self._pending = []


# ------------------------------------------------------------------------------------------------
# GPU engine (the C ABI) -- panels live in a torch tensor so that torch.distributed can ship them
# ------------------------------------------------------------------------------------------------
class GpuEngine:
    def __init__(self, cblk4, blok4, owner, rank, device_index, chunk=0):
        import torch
        self.torch = torch
        self.layout = LayoutArrays(cblk4, blok4)
        self.rank = rank
        opts = Options()
        opts.device = device_index
        opts.lookahead = chunk
        opts.external_arena = 1
        self._h = ctypes.c_void_p()
        own = np.ascontiguousarray(owner, dtype=np.int32)
        check(_lib.lib().pastix_amd_plan_create_dist(ctypes.byref(self.layout.c), 0, 1, ctypes.byref(opts),
                                                     _lib.ptr(own), ctypes.c_int32(rank), ctypes.byref(self._h)),
              "pastix_amd_plan_create_dist")
        nc = self.layout.cblknbr
        self.poff = np.zeros(nc + 1, dtype=np.int64)
        self.level = np.zeros(nc, dtype=np.int32)
        self.role = np.zeros(nc, dtype=np.int8)
        check(_lib.lib().pastix_amd_plan_layout_info(self._h, _lib.ptr(self.poff), _lib.ptr(self.level),
                                                     _lib.ptr(self.role)), "pastix_amd_plan_layout_info")
        self.device = torch.device("cuda", device_index)
        # 32 doubles of slack on both sides (pastix_amd_plan_set_arena: the update kernel's DMA lanes may touch
        # the element next to a panel); self.arena is the view the panels live in
        self._arena_store = torch.zeros(max(int(self.poff[-1]), 1) + 64, dtype=torch.float64, device=self.device)
        self.arena = self._arena_store[32:-32]
        check(_lib.lib().pastix_amd_plan_set_arena(self._h, ctypes.c_void_p(self.arena.data_ptr()), None),
              "pastix_amd_plan_set_arena")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().pastix_amd_plan_set_stream(self._h, ctypes.c_void_p(stream)), "pastix_amd_plan_set_stream")
        # receive side of the fan-in: row map of every (cblk owned here, sending rank) block, on the device
        self._c4 = np.asarray(cblk4, dtype=np.int64)
        self._width = (self._c4[:-1, 1] - self._c4[:-1, 0] + 1).astype(np.int64)
        mask = fanin_touched(cblk4, blok4, owner)
        self._rows = {}
        for src, t in fanin_pairs(cblk4, blok4, np.asarray(owner)).tolist():
            if owner[t] == rank:
                self._rows[(t, src)] = torch.from_numpy(fanin_rows(cblk4, blok4, mask, src, t)).to(self.device)

    def close(self):
        if self._h:
            _lib.lib().pastix_amd_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def stats(self):
        s = Stats()
        check(_lib.lib().pastix_amd_plan_stats(self._h, ctypes.byref(s)), "pastix_amd_plan_stats")
        return s.as_dict()

    def fill_csc(self, sym, n, colptr, rows, vals, perm):
        colptr, rows, perm = _lib.as_i64(colptr), _lib.as_i64(rows), _lib.as_i64(perm)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        check(_lib.lib().pastix_amd_fill_csc(self._h, int(sym), ctypes.c_int64(n), _lib.ptr(colptr), _lib.ptr(rows),
                                             _lib.ptr(vals), _lib.ptr(perm)), "pastix_amd_fill_csc")

    def refill(self):
        check(_lib.lib().pastix_amd_refill(self._h), "pastix_amd_refill")

    def begin(self, critere):
        check(_lib.lib().pastix_amd_factorize_begin(self._h, ctypes.c_double(critere)), "pastix_amd_factorize_begin")

    def update(self, l):
        check(_lib.lib().pastix_amd_factorize_level(self._h, int(l), 1), "pastix_amd_factorize_level")

    def panels(self, l):
        check(_lib.lib().pastix_amd_factorize_level(self._h, int(l), 2), "pastix_amd_factorize_level")

    def end(self):
        s = Stats()
        check(_lib.lib().pastix_amd_factorize_end(self._h, ctypes.byref(s)), "pastix_amd_factorize_end")
        return s.as_dict()

    def panel(self, k):
        return self.arena[int(self.poff[k]):int(self.poff[k + 1])]

    def recv_numel(self, k, src):
        return int(self._rows[(k, src)].numel() * self._width[k])

    def add(self, k, buf, src):
        rows = self._rows[(k, src)]
        if buf.device != self.device:
            buf = buf.to(self.device)
        check(_lib.lib().pastix_amd_plan_fanin_add(self._h, ctypes.c_int64(int(k)), ctypes.c_void_p(buf.data_ptr()),
                                                    ctypes.c_void_p(rows.data_ptr()), ctypes.c_int64(rows.numel())),
              "pastix_amd_plan_fanin_add")


def bench_distributed(a, rank, world, local):
    """bench.py's N>1 leg: every rank analyses the same matrix, owns a share of the elimination tree."""
    import time
    import torch
    import torch.distributed as dist
    from . import fact_flops
    from . import symbolic as sy
    N = a.grid
    t0 = time.time()
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=a.blocksize)
    c4, b4 = s["cblk4"], s["blok4"]
    flops = fact_flops(c4, b4, 0)
    owner = partition(c4, b4, world)
    level = levels_of(c4, b4)
    exch = Exchange(c4, b4, owner, level, rank)
    t_sym = time.time() - t0
    t0 = time.time()
    eng = GpuEngine(c4, b4, owner, rank, local, chunk=a.chunk)
    t_plan = time.time() - t0
    tr = TorchTransport(eng.device)
    crit = 6.0 * 2 * np.sqrt(1e-31)
    t0 = time.time()
    eng.fill_csc(1, n, cp, r, v, s["perm"])
    t_fill = time.time() - t0

    def step():
        eng.refill()
        eng.begin(crit)
        factorize_levels(eng, exch, tr)
        tr.drain()                                  # pending fan-in sends, before the buffers are zeroed again
        return eng.end()

    # open every point-to-point connection the factorization will use before anything is timed (RCCL sets a
    # pair up on its first message), whatever --warmup is
    peers = sorted({(int(q), int(owner[t])) for q, t in fanin_pairs(c4, b4, owner).tolist()})
    dev = "cpu" if dist.get_backend() == "gloo" else eng.device
    ops, keep = [], []
    for src, dst in peers:
        if src == rank:
            keep.append(torch.zeros(1, dtype=torch.float64, device=dev))
            ops.append(dist.P2POp(dist.isend, keep[-1], dst))
        elif dst == rank:
            keep.append(torch.zeros(1, dtype=torch.float64, device=dev))
            ops.append(dist.P2POp(dist.irecv, keep[-1], src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for _ in range(a.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    ft = ut = 0.0
    st = None
    for _ in range(a.steps):
        st = step()
        ft += st["fact_time"]
        ut += st["update_time"]
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.time() - t0
    ps = eng.stats()
    # size-independent check of the distributed factors (no solve across ranks here): log det A = 2 sum log L_kk
    # over the owned cblks of all ranks, against the analytic spectrum of the 7-point Dirichlet Laplacian
    # (eigenvalues 6 - 2cos(i pi/(N+1)) - 2cos(j pi/(N+1)) - 2cos(k pi/(N+1)))
    wid = (c4[:-1, 1] - c4[:-1, 0] + 1).astype(np.int64)
    own = np.nonzero(eng.role == 1)[0]
    rep = np.repeat(own, wid[own])
    col = np.arange(len(rep), dtype=np.int64) - np.repeat(np.cumsum(wid[own]) - wid[own], wid[own])
    didx = eng.poff[rep] + col * (c4[rep, 3] + 1)
    dvals = eng.arena[torch.from_numpy(didx).to(eng.device)]
    ld_local = float(2.0 * torch.log(dvals).sum().item()) if len(didx) else 0.0
    cs = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
    ld_exact = float(np.log(6.0 - cs[:, None, None] - cs[None, :, None] - cs[None, None, :]).sum())
    tw = torch.tensor([wall, ut, ps["update_flops"], ps["local_flops"], ld_local], dtype=torch.float64,
                      device="cpu" if dist.get_backend() == "gloo" else eng.device)
    mx = tw.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tw.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    nsend = sum(len(x) for x in exch.sends)
    ld_err = abs(float(sm[4]) - ld_exact) / abs(ld_exact)
    if not ld_err < 1e-9:
        raise RuntimeError("distributed factorization failed its log-det check: %.15g vs %.15g" % (float(sm[4]), ld_exact))
    res = dict(wall=float(mx[0]), flops=flops, logdet_rel_err=ld_err, fact_time=ft, update_time=float(sm[1]) / world,
               update_flops=float(sm[2]) / world, update_bytes=ps["update_bytes"], nlaunch=st["nupdate_launches"], resid=None, nbpivot=st["nbpivot"],
               n=n, cblk=len(c4) - 1, blok=len(b4), nnzl=s["nnzl"], coefnbr=ps["coefnbr"], t_sym=t_sym,
               t_plan=t_plan, t_fill=t_fill, ntasks=ps["ntasks"], npieces=ps["npieces"], nlevels=ps["nlevels"],
               parallelism="subtree-per-gpu fan-in x%d (rank0 owns %.1f%% of flops, %d fan-in sends/rank0, "
                           "rank0 arena %.1f GB of which fan-in buffers %.1f GB)"
                           % (world, 100.0 * ps["local_flops"] / flops, nsend, 8e-9 * float(eng.poff[-1]),
                              8e-9 * float(sum(int(eng.poff[k + 1] - eng.poff[k]) for k in np.nonzero(eng.role == 2)[0]))))
    eng.close()
    return res
