"""Multi-rank fan-in protocol on CPU: partition properties and a world_size-2 gloo run of
pastix_amd.dist.factorize_levels with the numpy engine, checked against the golden reference factors."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import HERE, ROOT
from pastix_amd import dist as pd
from pastix_amd import symbolic as sy


def test_partition_balanced_and_subtree_closed():
    n, cp, r, v = sy.laplacian_3d(20)
    perm, _ = sy.order_grid(20, 20, 20)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=32)
    c4, b4 = s["cblk4"], s["blok4"]
    fl = pd.cblk_flops(c4, b4)
    from pastix_amd import fact_flops
    assert abs(fl.sum() - fact_flops(c4, b4, 0)) <= 1e-9 * fl.sum()
    for P in (2, 4, 8):
        ow = pd.partition(c4, b4, P)
        loads = np.array([fl[ow == q].sum() for q in range(P)]) / fl.sum()
        assert loads.max() < 1.25 / P
        lv = pd.levels_of(c4, b4)
        pairs = pd.fanin_pairs(c4, b4, ow)
        # every fan-in message goes to a cblk of a strictly higher level than some source of the sender
        assert all(ow[t] != r_ for r_, t in pairs.tolist())
        ex = [pd.Exchange(c4, b4, ow, lv, q) for q in range(P)]
        nsend = sum(len(x) for e in ex for x in e.sends)
        nrecv = sum(len(x) for e in ex for x in e.recvs)
        assert nsend == nrecv == len(pairs)


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fixture_io
    from np_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = fixture_io.load_npz(os.path.join(HERE, "golden", name + ".npz"))
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world, split=2)
    level = pd.levels_of(c4, b4)
    eng = NumpyEngine(c4, b4, owner, level, rank, g["L0"])
    exch = pd.Exchange(c4, b4, owner, level, rank)
    tr = pd.TorchTransport(torch.device("cpu"))
    pd.factorize_levels(eng, exch, tr)
    tr.drain()
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    err = 0.0
    for k in np.nonzero(owner == rank)[0]:
        ref = g["L1"][off[k]:off[k + 1]].reshape(int(w[k]), -1).T
        got = eng.panel(k).numpy().reshape(int(w[k]), -1).T
        wk = int(w[k])
        # compare the lower triangle of the diagonal blok and the off-diagonal rows
        m = np.ones_like(ref, dtype=bool)
        m[:wk, :wk] = np.tril(np.ones((wk, wk), dtype=bool))
        err = max(err, float(np.abs(got - ref)[m].max()))
    nsend = sum(len(x) for x in exch.sends)
    q.put((rank, err, int((owner == rank).sum()), nsend))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("rlap3d_10_llt", 2), ("rlap3d_14_llt_bs24", 2), ("rlap3d_14_llt_bs24", 4)])
def test_fanin_over_gloo(name, world, golden):
    """world_size 2 and 4: per-rank plans, fan-in exchange in lockstep levels, pure senders running ahead."""
    g = golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scale = np.abs(g["L1"]).max()
    for rank, err, nown, nsend in res:
        assert err <= 1e-11 * scale
        assert nown > 0
    assert sum(r[3] for r in res) > 0          # the exchange path was exercised


def test_fanin_touched_regions(golden):
    """pastix_amd_fanin_touched (host only): a rank is marked on a blok only if it owns a source cblk facing the
    blok's cblk and does not own that cblk; the rows a sender packs are rows of exactly those bloks."""
    from pastix_amd import dist as pd
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    for world in (2, 3, 4):
        owner = pd.partition(c4, b4, world, split=2)
        mask = pd.fanin_touched(c4, b4, owner)
        pairs = {(int(r), int(t)) for r, t in pd.fanin_pairs(c4, b4, owner)}
        cb = np.repeat(np.arange(len(c4) - 1), np.diff(c4[:, 2]))        # cblk of every blok
        seen = set()
        for b in np.nonzero(mask)[0]:
            t = int(cb[b])
            for r in range(world):
                if (int(mask[b]) >> r) & 1:
                    assert r != owner[t] and (r, t) in pairs
                    seen.add((r, t))
        assert seen == pairs                      # every sender/cblk pair has at least one region
        for r, t in sorted(pairs)[:50]:
            rows = pd.fanin_rows(c4, b4, mask, r, t)
            assert len(rows) > 0 and len(np.unique(rows)) == len(rows) and rows.max() < c4[t, 3]


def _worker_sched(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fixture_io
    from np_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = fixture_io.load_npz(os.path.join(HERE, "golden", name + ".npz"))
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world)
    level = pd.levels_of(c4, b4)
    eng = NumpyEngine(c4, b4, owner, level, rank, g["L0"])
    msgs, npl = pd.schedule(c4, b4, owner, rank, world)
    mask = pd.fanin_touched(c4, b4, owner)
    pd.factorize_scheduled(eng, msgs, int(level.max()) + 1, lambda src, t: pd.fanin_rows(c4, b4, mask, src, t),
                           lambda buf, peer: dist.isend(buf, peer), lambda buf, peer: dist.irecv(buf, peer))
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    err = 0.0
    for k in np.nonzero(owner == rank)[0]:
        ref = g["L1"][off[k]:off[k + 1]].reshape(int(w[k]), -1).T
        got = eng.panel(k).numpy().reshape(int(w[k]), -1).T
        wk = int(w[k])
        m = np.ones_like(ref, dtype=bool)
        m[:wk, :wk] = np.tril(np.ones((wk, wk), dtype=bool))
        err = max(err, float(np.abs(got - ref)[m].max()))
    q.put((rank, err, int((msgs[:, 3] == 0).sum()), npl))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("rlap3d_10_llt", 2), ("rlap3d_14_llt_bs24", 2), ("rlap3d_14_llt_bs24", 4),
                                        ("rlap3d_14_llt_bs24", 3)])
def test_native_schedule_over_gloo(name, world, golden):
    """The message schedule of the native driver (pastix_amd_dist_schedule: which compact blocks, in which order per
    channel) replayed by world_size 2/3/4 gloo processes with the numpy engine: factors equal the reference's."""
    g = golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker_sched, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scale = np.abs(g["L1"]).max()
    for rank, err, nsend, npl in res:
        assert err <= 1e-11 * scale and npl == 1
    assert sum(r[2] for r in res) > 0


@pytest.mark.parametrize("facto,floattype,npl", [(0, 1, 1), (1, 1, 1), (2, 1, 2), (1, 3, 2), (3, 3, 2), (2, 3, 4)])
def test_native_schedule_is_consistent_between_ranks(facto, floattype, npl, golden):
    """Every send has exactly one matching receive (same level, cblk, extent) on the owner, both ends list the blocks
    of a channel in the same order, and the arenas per block follow the factorization (LU: L and U; complex: + the
    imaginary planes)."""
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    for world in (2, 3, 5, 8):
        owner = pd.partition(c4, b4, world)
        sch = []
        for r in range(world):
            m, n = pd.schedule(c4, b4, owner, r, world, facto, floattype)
            assert n == npl
            sch.append(m)
        level = pd.levels_of(c4, b4)
        nsend = 0
        for a in range(world):
            assert (np.diff(sch[a][:, 0]) >= 0).all()                      # by level
            for b in range(world):
                if a == b:
                    continue
                ab = sch[a][sch[a][:, 1] == b]                             # channel (a, b) seen from a
                ba = sch[b][sch[b][:, 1] == a]                             # ... and from b
                assert len(ab) == len(ba)
                assert np.array_equal(ab[:, [0, 2, 4, 5]], ba[:, [0, 2, 4, 5]])   # same blocks, same order
                assert np.array_equal(ab[:, 3], 1 - ba[:, 3])              # a send here is a receive there
                nsend += int((ab[:, 3] == 0).sum())
                for lvl, _p, t, d, nr, wd in ab.tolist():
                    assert lvl == level[t] and owner[t] == (b if d == 0 else a)
        assert nsend == len(pd.fanin_pairs(c4, b4, owner))


def test_schedule_hashes_detect_a_rank_that_planned_differently(golden):
    """pastix_amd_dist_schedule_hash (host only): with one owner map the two ends of every channel hash alike; a rank that
    planned from another owner map (a cblk given to someone else) is caught on exactly the channels it touches."""
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    for facto, ft in ((0, 1), (2, 1), (1, 3)):
        for world in (2, 4, 8):
            owner = pd.partition(c4, b4, world)
            table = [pd.schedule_hashes(c4, b4, owner, r, world, facto, ft) for r in range(world)]
            assert pd.mismatched_channels(table) == []
    world = 4
    owner = pd.partition(c4, b4, world)
    table = [pd.schedule_hashes(c4, b4, owner, r, world) for r in range(world)]
    pairs = pd.fanin_pairs(c4, b4, owner)
    t = int(pairs[len(pairs) // 2, 1])                  # a cblk that receives a fan-in block
    other = owner.copy()
    other[t] = (owner[t] + 1) % world
    bad_rank = int(owner[t])
    table[bad_rank] = pd.schedule_hashes(c4, b4, other, bad_rank, world)
    bad = pd.mismatched_channels(table)
    assert bad and all(bad_rank in ab for ab in bad)


def _worker_hash(rank, world, port, name, poison, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fixture_io
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = fixture_io.load_npz(os.path.join(HERE, "golden", name + ".npz"))
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world)
    if poison and rank == 1:                            # this rank analysed something else
        t = int(pd.fanin_pairs(c4, b4, owner)[0, 1])
        owner = owner.copy()
        owner[t] = (owner[t] + 1) % world
    try:
        pd.check_schedule_hashes(c4, b4, owner, rank, world)
        q.put((rank, True, ""))
    except RuntimeError as e:
        q.put((rank, False, str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("poison", [False, True])
def test_schedule_handshake_over_gloo(poison):
    """check_schedule_hashes, the collective bench.py --gpus N runs before any RCCL communicator exists: world 2 over
    gloo; a disagreement raises on EVERY rank (nobody is left waiting in a rendezvous)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    port = 33700 + (os.getpid() % 2000) + int(poison)
    procs = [ctx.Process(target=_worker_hash, args=(r, world, port, "rlap3d_14_llt_bs24", poison, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok != poison for _r, ok, _m in res), res
    if poison:
        assert all("disagree" in m for _r, _ok, m in res)


def test_c_abi_partition_equals_the_restatement_and_balances():
    """pastix_amd_dist_partition (csrc/partition.cpp, what a C caller of INTEGRATION.md section 3 uses) against the test-side
    numpy restatement of the same rule, rank for rank; every rank gets work; a subtree with one candidate rank is whole."""
    import np_partition as ref
    for N, bs in [(10, 16), (16, 32), (24, 64), (30, 128)]:
        n, cp, r, v = sy.laplacian_3d(N)
        perm, _ = sy.order_grid(N, N, N)
        s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
        c4, b4 = s["cblk4"], s["blok4"]
        fl = ref.cblk_flops(c4, b4)
        for W in (1, 2, 3, 4, 5, 8, 16):
            ow = pd.partition(c4, b4, W)
            assert np.array_equal(ow, ref.partition(c4, b4, W)), (N, bs, W)
            assert ow.min() >= 0 and ow.max() < W
            if W <= 8:
                share = np.bincount(ow, weights=fl, minlength=W) / fl.sum()
                assert share.min() > 0 and share.max() <= (0.75 if W > 1 else 1.0)

