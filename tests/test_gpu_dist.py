"""Multi-GPU driver (csrc/dist.cpp, pastix_amd_factorize_dist) on the GPU.

* single-GPU boxes: the rank plans of one process share the GPU, wired by the loopback transport and driven by one
  host thread per rank -- the same driver, schedule, channels, events and add kernels as over RCCL, against the
  reference's own factors for d LLt / LDLt / LU and z LDLt / LDLh / LU;
* boxes with >= 2 GPUs: world = 2 processes over the nccl backend (RCCL), one GPU each, same check.
"""
import os
import sys

import numpy as np
import pytest

from conftest import HERE, ROOT, recut_mask
from pastix_amd import dist as pd
from pastix_amd import symbolic as sy

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

TOL = 1e-12


def _lower_mask_cblk(w, s):
    m = np.ones((w, s), dtype=bool)           # [col][row]
    for c in range(w):
        m[c, :c] = False
    return m.ravel()


def _check_owned(g, plan, owner, rank):
    c4 = g["cblk4"]
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    scale = max(np.abs(g["L1"]).max(), np.abs(g["U1"]).max() if g["facto"] == 2 else 0.0)
    got = plan.download_owned()
    assert sorted(got) == sorted(np.nonzero(owner == rank)[0].tolist())
    for k, (L, U) in got.items():
        ref = g["L1"][off[k]:off[k + 1]]
        m = (_lower_mask_cblk(int(w[k]), int(c4[k, 3])) if g["facto"] in (1, 3)
             else recut_mask(c4[k:k + 2]) if g["facto"] == 0 else np.ones(ref.size, bool))
        assert np.abs(L - ref)[m].max() <= TOL * scale, k
        if g["facto"] == 2:
            assert np.abs(U - g["U1"][off[k]:off[k + 1]]).max() <= TOL * scale, k


CASES = [("rlap3d_14_llt_bs24", 2), ("rlap3d_14_llt_bs24", 4), ("rlap3d_14_llt_bs24", 3), ("rlap3d_20_llt_bs128", 4),
         ("rlap3d_12_ldlt", 2), ("rlap3d_12_lu", 2), ("rlap3d_12_lu", 4), ("rlap3d_20_lu_bs128", 2),
         ("zrlap3d_12_ldlt", 2), ("zrlap3d_12_ldlh", 3), ("zrlap3d_12_lu", 2), ("zrlap3d_8_lu", 4)]


@pytest.mark.parametrize("name,world", CASES)
def test_native_driver_loopback_matches_reference(name, world, golden):
    g = golden(name)
    c4, b4 = g["cblk4"], g["blok4"]
    cz = np.iscomplexobj(g["L0"])
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, r, 0, factotype=g["facto"], floattype=3 if cz else 1) for r in range(world)]
    try:
        pd.attach_local(plans)
        infos = [p.info() for p in plans]
        assert all(i["transport"] == "loopback" and i["world"] == world for i in infos)
        assert sum(i["nsend"] for i in infos) == sum(i["nrecv"] for i in infos) > 0
        assert abs(sum(i["bytes_sent"] for i in infos) - sum(i["bytes_recv"] for i in infos)) < 1
        for rep in range(2):                       # the second pass checks that nothing of the first one lingers
            for p in plans:
                if rep == 0:
                    p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
                else:
                    p.refill()
            sts = pd.factorize_local(plans, g["critere"])
            assert sum(s["nbpivot"] for s in sts) == g["nbpivot"]
            for r, p in enumerate(plans):
                _check_owned(g, p, owner, r)
    finally:
        for p in plans:
            p.close()


@pytest.mark.parametrize("name,world", [("rlap3d_14_llt_bs24", 3), ("rlap3d_12_ldlt", 2), ("rlap3d_20_lu_bs128", 4),
                                        ("zrlap3d_12_ldlt", 2)])
def test_upload_owned_then_factorize_dist(name, world, golden):
    """pastix_amd_upload_tabs on DISTRIBUTED plans (the caller's per-cblk buffers instead of the device fill): only owned
    panels travel, and the fan-in buffers that lie between them in the arena start from zeros -- twice, so that what the
    first factorization left in them, and what the staging buffers hold of another rank's panels, would show."""
    g = golden(name)
    c4, b4 = g["cblk4"], g["blok4"]
    cz = np.iscomplexobj(g["L0"])
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, r, 0, factotype=g["facto"], floattype=3 if cz else 1) for r in range(world)]
    try:
        pd.attach_local(plans)
        for rep in range(2):
            for p in plans:
                p.upload_owned(g["L0"], g["U0"] if g["facto"] == 2 else None)
            sts = pd.factorize_local(plans, g["critere"])
            assert sum(s["nbpivot"] for s in sts) == g["nbpivot"]
            for r, p in enumerate(plans):
                _check_owned(g, p, owner, r)
    finally:
        for p in plans:
            p.close()


@pytest.mark.parametrize("name,world", [("rlap3d_14_llt_bs24", 2), ("rlap3d_14_llt_bs24", 4), ("rlap3d_12_ldlt", 3),
                                        ("rlap3d_12_lu", 2), ("rlap3d_20_lu_bs128", 4), ("rlap3d_20_llt_bs128", 3)])
def test_distributed_solve_matches_reference(name, world, golden):
    """pastix_amd_solve_dist (fan-in of the vector contributions forward, the same channels backward) on the factors of
    the multi-GPU driver: the assembled solution equals the reference's own."""
    g = golden(name)
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, r, 0, factotype=g["facto"]) for r in range(world)]
    try:
        pd.attach_local(plans)
        for p in plans:
            p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        pd.factorize_local(plans, g["critere"])
        bp = np.empty(g["n"])
        bp[g["perm"]] = g["b"]
        for rep in range(2):
            x = pd.solve_local(plans, bp)[g["perm"]]
            assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()
    finally:
        for p in plans:
            p.close()


def test_native_driver_loopback_is_repeatable(golden):
    """The owner adds the received blocks in a fixed order; what remains timing dependent is the order of the fp64
    atomics with which the split-K tasks of a shared target tile combine (plan.cpp): repeated runs agree to rounding."""
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    world = 4
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, r, 0) for r in range(world)]
    try:
        pd.attach_local(plans)
        runs = []
        for rep in range(3):
            for p in plans:
                p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"]) if rep == 0 else p.refill()
            pd.factorize_local(plans, g["critere"])
            runs.append([p.download_owned() for p in plans])
        scale = np.abs(g["L1"]).max()
        for other in runs[1:]:
            for a, b in zip(runs[0], other):
                for k in a:
                    assert np.abs(a[k][0] - b[k][0]).max() <= 1e-14 * scale
    finally:
        for p in plans:
            p.close()


@pytest.mark.parametrize("N,world", [(36, 4), (40, 8)])
def test_native_driver_on_own_layout_at_scale(N, world):
    """36^3 / 40^3 on the repo's own layout (128-wide cblks), 4 / 8 emulated ranks (8 = the metric's largest job): log det A
    from the distributed factors against the analytic spectrum of the Dirichlet Laplacian (size-independent check, as
    bench.py --gpus N uses it)."""
    from pastix_amd import symbolic as sy
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    c4, b4 = s["cblk4"], s["blok4"]
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, q, 0) for q in range(world)]
    try:
        pd.attach_local(plans)
        for p in plans:
            p.fill_csc(1, n, cp, r, v, s["perm"])
        pd.factorize_local(plans, 1e-14)
        ld = 2.0 * sum(p.diag_logsum() for p in plans)
        cs = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
        exact = float(np.log(6.0 - cs[:, None, None] - cs[None, :, None] - cs[None, None, :]).sum())
        assert abs(ld - exact) <= 1e-11 * abs(exact)
    finally:
        for p in plans:
            p.close()


def test_rccl_binding_selftest():
    """librccl resolved at run time, ncclDouble, grouped send + receive on a non-default stream: a 1-rank communicator
    sending to itself -- the part of the RCCL path one GPU can execute.  In a child process that leaves with os._exit:
    RCCL's teardown at interpreter exit is not this repo's business (it has been seen to abort)."""
    import subprocess
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from pastix_amd import _lib; _lib.share_rccl_with_torch(); "
            "rc = _lib.lib().pastix_amd_dist_selftest_rccl(0, ctypes.c_int64(100003)); print('selftest rc', rc, flush=True); "
            "os._exit(0 if rc == 0 else 1)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "selftest rc 0" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


# ---- RCCL, world = 2 (needs two GPUs) -----------------------------------------------------------------
def _rccl_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import fixture_io
    from pastix_amd import dist as pdd
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    g = fixture_io.load_npz(os.path.join(HERE, "golden", name + ".npz"))
    c4, b4 = g["cblk4"], g["blok4"]
    cz = np.iscomplexobj(g["L0"])
    owner = pdd.partition(c4, b4, world)
    ok, msg = True, ""
    try:
        pdd.check_schedule_hashes(c4, b4, owner, rank, world, g["facto"], 3 if cz else 1)
        with pdd.DistPlan(c4, b4, owner, rank, rank, factotype=g["facto"], floattype=3 if cz else 1) as p:
            p.attach_rccl(world, pdd.exchange_unique_ids(c4, b4, owner, rank, world))
            assert p.info()["transport"] == "rccl"
            p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
            for rep in range(2):
                if rep:
                    p.refill()
                st = p.factorize(g["critere"])
                _check_owned(g, p, owner, rank)
            if not cz:
                bp = np.empty(g["n"])
                bp[g["perm"]] = g["b"]
                xpart = torch.from_numpy(p.solve(bp)).to("cuda")
                dist.all_reduce(xpart)
                x = xpart.cpu().numpy()[g["perm"]]
                assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()
            nb = torch.tensor([st["nbpivot"]], device="cuda")
            dist.all_reduce(nb)
            assert int(nb.item()) == g["nbpivot"]
    except Exception as e:  # noqa: BLE001
        ok, msg = False, repr(e)
    q.put((rank, ok, msg))
    q.close()
    q.join_thread()
    if ok:
        dist.barrier()
        dist.destroy_process_group()
    # (leave without interpreter teardown -- a failed rank may hold aborted channels -- with the code it earned)
    os._exit(0 if ok else 1)


@pytest.mark.parametrize("name", ["rlap3d_14_llt_bs24", "rlap3d_20_llt_bs128", "rlap3d_12_lu", "zrlap3d_12_ldlt"])
def test_native_driver_over_rccl_world2(name):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, ok, msg in res:
        assert ok, (rank, msg)
    for p in procs:
        assert p.exitcode == 0


def test_unmatched_rank_times_out_instead_of_hanging():
    """Hang protection of the multi-GPU driver (csrc/dist.cpp: dist_finish): rank 1 of a 2-rank loopback job runs ALONE --
    the peer it receives from never shows up.  Within PASTIX_AMD_DIST_TIMEOUT seconds it reports the unmatched receive on
    stderr, aborts its channels, returns PASTIX_AMD_ERR_TIMEOUT (-7); its distributed state is dead afterwards (the next
    call is refused, nothing hangs) while the plan can still be destroyed.  In a child process: the deadline is read
    once per process."""
    import subprocess
    code = r'''
import ctypes, os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import fixture_io
from pastix_amd import _lib
from pastix_amd import dist as pd
from pastix_amd import symbolic as sy
g = fixture_io.load_npz(os.path.join(%r, "golden", "rlap3d_14_llt_bs24.npz"))
c4, b4 = g["cblk4"], g["blok4"]
owner = pd.partition(c4, b4, 2)
plans = [pd.DistPlan(c4, b4, owner, r, 0) for r in range(2)]
pd.attach_local(plans)
for p in plans:
    p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
recv = [r for r in range(2) if plans[r].info()["nrecv"] > 0][0]
t0 = time.time()
st = _lib.Stats()
rc = _lib.lib().pastix_amd_factorize_dist(plans[recv]._h, ctypes.c_double(g["critere"]), ctypes.byref(st))
dt = time.time() - t0
rc2 = _lib.lib().pastix_amd_factorize_dist(plans[recv]._h, ctypes.c_double(g["critere"]), ctypes.byref(st))
for p in plans:
    p.close()
print("RESULT", rc, rc2, "%%.1f" %% dt, flush=True)
''' % (ROOT, HERE, os.path.join(HERE, "golden"), HERE)
    env = dict(os.environ, PASTIX_AMD_DIST_TIMEOUT="3")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, (out.stdout[-500:], out.stderr[-1500:])
    _tag, rc, rc2, dt = line[0].split()
    assert int(rc) == -7 and int(rc2) == -1, (line, out.stderr[-1500:])
    assert 2.5 <= float(dt) < 60.0
    assert "not matched within" in out.stderr and "aborting the channels" in out.stderr


def test_loopback_attach_rejects_mismatched_schedules(golden):
    """attach_local compares both ends of every channel (pastix_amd_dist_schedule_hash) before wiring them: rank plans
    built from different owner maps are refused with PASTIX_AMD_ERR_LAYOUT."""
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, 2)
    t = int(pd.fanin_pairs(c4, b4, owner)[0, 1])
    other = owner.copy()
    other[t] = 1 - owner[t]
    plans = [pd.DistPlan(c4, b4, owner, 0, 0), pd.DistPlan(c4, b4, other, 1, 0)]
    try:
        with pytest.raises(RuntimeError, match="-6"):
            pd.attach_local(plans)
    finally:
        for p in plans:
            p.close()


@pytest.mark.parametrize("world", [2, 4])
def test_config2_lu_across_ranks_at_a_size_one_gpu_holds(world):
    """BASELINE configs[2] -- 200^3 dLU, 2 x 150 GB of panels: a 2-GPU job (`bench.py --gpus 2 --facto lu --grid 200`) --
    through exactly that path (partition, per-rank LU plans with two planes of fan-in blocks, the native driver, the
    distributed solve) at 48^3 with the ranks sharing this GPU over the loopback transport: residual of the unsymmetric
    system, no static pivot, every rank's share of the flops positive."""
    import scipy.sparse as sp
    N = 48
    n, cp, r, v = sy.laplacian_3d(N, full=True)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    c4, b4 = s["cblk4"], s["blok4"]
    owner = pd.partition(c4, b4, world)
    table = [pd.schedule_hashes(c4, b4, owner, q, world, factotype=2) for q in range(world)]
    assert pd.mismatched_channels(table) == []
    plans = [pd.DistPlan(c4, b4, owner, q, 0, factotype=2) for q in range(world)]
    try:
        pd.attach_local(plans)
        for p in plans:
            p.fill_csc(0, n, cp, r, v, s["perm"])
        sts = pd.factorize_local(plans, 6.0 * 2 * np.sqrt(1e-31))
        assert sum(st["nbpivot"] for st in sts) == 0
        assert all(p.stats()["local_flops"] > 0 for p in plans)
        b = np.random.default_rng(2).random(n)
        bp = np.empty(n)
        bp[s["perm"]] = b
        x = pd.solve_local(plans, bp)[s["perm"]]
        A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-10
    finally:
        for p in plans:
            p.close()


@pytest.mark.parametrize("name,world", [("zrlap3d_12_ldlt", 2), ("zrlap3d_12_ldlh", 3), ("zrlap3d_12_lu", 2), ("zrlap3d_20_ldlt_bs128", 4)])
def test_distributed_solve_complex_matches_reference(name, world, golden):
    """pastix_amd_solve_dist on complex plans: every vector segment travels as two planes (re, im) over the channels of the
    factorization; the assembled solution equals the reference's own (z LDLt, LDLh, LU)."""
    g = golden(name)
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world)
    plans = [pd.DistPlan(c4, b4, owner, r, 0, factotype=g["facto"], floattype=3) for r in range(world)]
    try:
        pd.attach_local(plans)
        for p in plans:
            p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        pd.factorize_local(plans, g["critere"])
        bp = np.empty(g["n"], dtype=np.complex128)
        bp[g["perm"]] = g["b"]
        for rep in range(2):
            x = pd.solve_local(plans, bp)[g["perm"]]
            assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()
    finally:
        for p in plans:
            p.close()

