#!/usr/bin/env python3
"""bench.py -- factorization GFLOP/s of the MI355X sopalin path on BASELINE.json's headline config.

A step = one pass of the hot path over one synthetic matrix: device re-fill of the panels (zero +
scatter of the CSC values, inputs already resident in HBM) followed by the numerical factorization.
Workload at N=1: 3-D 7-point Laplacian 200^3, double LLt (the configuration the metric is quoted on;
~140 GB of panels, fits one 288 GB MI355X).  --grid overrides the size (e.g. 100 = configs[1]).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel k_update (GEMM + scatter), MFMA-bound; achieved = algorithmic update
                 flops / time inside k_update launches, both measured live with HIP events on the
                 engine's own stream.
  cpu_baseline : the REAL reference (oracle/_ref, PaStiX 5.2.2.16 CPU sopalin + OpenBLAS) timed on this
                 box's host cores on a bounded sample (smaller grid of the same stencil), rank 0, N=1.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F64_PEAK = 78.6e12   # dense fp64 matrix peak of MI355X (vendor); tools/probe_mfma_f64 measures 77.5e12
MFMA_F32_PEAK = 157.3e12  # dense fp32 (f32-input) matrix peak (MI355X_MICROARCH: 157.3 spec, 155 measured)


def cpu_baseline(sample_grid, threads):
    """Time the real reference on host cores; fall back to the oracle port when oracle/_ref is absent."""
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="1")
    for exe, blas in (("ref_harness_d_ob", "OpenBLAS"), ("ref_harness_d", "MKL")):
        path = os.path.join(ROOT, "oracle", "_ref", exe)
        if not os.path.exists(path):
            continue
        try:
            out = subprocess.run([path, "time", "lap3d", str(sample_grid), "llt", str(threads), "/dev/null"], env=env,
                                 capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            return {"value": round(d["gflops"], 2), "unit": "GFLOP/s", "cores": threads, "kind": "reference",
                    "sample": "PaStiX 5.2.2.16 CPU sopalin (%s, %d threads), 3-D Laplacian %d^3 dLLt, "
                              "%.3e flop in %.2f s" % (blas, threads, sample_grid, d["flops"], d["time"])}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("cpu_baseline: %s failed: %r\n" % (exe, e))
    # scalar port (the oracle), one core
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pastix_amd import symbolic as sy
    from pastix_amd import fact_flops
    N = 30
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm)
    L0, _ = oracle_lib.fill(0, 1, n, cp, r, v, s["perm"], s["cblk4"], s["blok4"])
    t = time.time()
    oracle_lib.sopalin(0, s["cblk4"], s["blok4"], L0, None, 1e-14)
    dt = time.time() - t
    fl = fact_flops(s["cblk4"], s["blok4"], 0)
    return {"value": round(fl / dt * 1e-9, 2), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": "oracle restatement (plain C, no BLAS), 3-D Laplacian %d^3 dLLt" % N}


class PowerWatch:
    """Socket power and shader clock while the timed steps run (rocm-smi in a child process every 0.5 s, text output; a box
    without rocm-smi, or one whose output has another shape, reports nulls).  VERDICT r4 item 8: the sustained clock beside
    the rate -- the fp64 MFMA peak is quoted at 2.4 GHz, the chip holds about 2.29 under this kernel at its power cap."""

    def __init__(self, enabled=True):
        self.pw, self.ck, self.stop, self.th = [], [], False, None
        import shutil
        self.exe = shutil.which("rocm-smi") if enabled else None

    def _loop(self):
        import re
        import subprocess
        while not self.stop:
            try:
                out = subprocess.run([self.exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                for line in out.splitlines():
                    if "sclk" in line:
                        m = re.search(r"(\d+)\s*Mhz", line, re.I)
                        if m:
                            self.ck.append(float(m.group(1)))
                            break
                for line in out.splitlines():
                    if "Power" in line and "(W)" in line:
                        m = re.search(r"([0-9]+(?:\.[0-9]+)?)\s*$", line.strip())
                        if m:
                            self.pw.append(float(m.group(1)))
                            break
            except Exception:
                pass
            time.sleep(0.5)

    def __enter__(self):
        if self.exe:
            import threading
            self.th = threading.Thread(target=self._loop, daemon=True)
            self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.th:
            self.th.join(timeout=6)

    def summary(self):
        f = lambda v: round(sum(v) / len(v), 1) if v else None
        return {"socket_power_w_avg": f(self.pw), "sclk_mhz_avg": f(self.ck), "samples": len(self.ck),
                "source": "rocm-smi --showclocks --showpower every 0.5 s during the timed steps" if self.exe else None}


def engine_source_sha():
    """Identity of the code the HBM-traffic counters were collected on: sha256 over EVERY engine source
    (pastix_amd/csrc/*.hip, *.cpp, *.h, the Makefile) and the values of the PASTIX_AMD_* environment knobs that shape the
    schedule.  profiles/*/traffic_k_update.json records it (tools/profile_round.sh); a traffic figure measured on other
    sources or under other knobs is stale and is reported as null."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pastix_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp")) + glob.glob(os.path.join(d, "*.h"))
                    + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    for k in sorted(os.environ):
        if k.startswith("PASTIX_AMD_") and k not in ("PASTIX_AMD_BENCH_GRID", "PASTIX_AMD_LIB", "PASTIX_AMD_DIST_TIMEOUT",
                                                    "PASTIX_AMD_BENCH_PREFLIGHT", "PASTIX_AMD_VERBOSE"):
            h.update(("%s=%s" % (k, os.environ[k])).encode())
    return h.hexdigest()[:16]


def measured_traffic(grid, facto, blocksize, dtype="f64", chunk=0, workload="laplacian"):
    """PMC-measured HBM bytes of the bulk update kernels of one factorization of this workload, from the newest
    profiles/rNN/traffic_k_update.json collected on exactly these engine sources; else None.  The file holds the fp64
    engine's default schedule on the Laplacian only: any other arithmetic, chunk size or workload has no measurement."""
    import glob
    if dtype != "f64" or chunk != 0 or workload != "laplacian":
        return None, None
    sha = engine_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_k_update.json")), reverse=True):
        try:
            tj = json.load(open(f))
            e = tj.get(str(grid))
            if e and e.get("source_sha") == sha and facto == "llt" and blocksize == 128:
                return e["bytes_per_factorization"], os.path.relpath(f, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def self_launch(a, argv):
    """`python bench.py --gpus N` (N>1) outside a launcher: start N ranks (one per GPU, RCCL) with torch.distributed.run
    as a CHILD process -- before this process has made any GPU call -- and leave with its exit code."""
    import socket
    import torch
    ndev = torch.cuda.device_count()          # (does not initialise the GPU)
    if ndev < a.gpus and os.environ.get("PASTIX_AMD_DIST_TEST") != "1":
        raise SystemExit("bench.py: --gpus %d but this box has %d GPU(s); refusing to report a %d-GPU number "
                         "(PASTIX_AMD_DIST_TEST=1 time-slices one GPU over gloo for validation only)"
                         % (a.gpus, ndev, a.gpus))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (a.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd))


def single_gpu_job(grid, workload, facto_name, steps, warmup, blocksize, chunk, local, f32=False, power=False, run_schedule=0):
    """One configuration on one GPU: analysis, plan, device fill, `warmup` untimed and `steps` timed steps (a step =
    device re-fill + factorization, inputs resident in HBM), then the end-to-end check ||Ax - b|| / ||b|| with the device
    solve on the last factors.  Returns the raw figures the JSON line is made of."""
    import numpy as np
    import torch
    from pastix_amd import Plan, fact_flops
    from pastix_amd import symbolic as sy
    N = grid
    t0 = time.time()
    zel = workload == "elasticity"
    if zel:
        from pastix_amd import COMPLEXDOUBLE
        if facto_name == "llt":
            facto_name = "ldlt"                    # (complex symmetric: LDLt, the `sy` variant)
        facto = {"ldlt": 1, "lu": 2}[facto_name]
        if facto == 2:
            raise SystemExit("bench.py: --workload elasticity is the complex-symmetric LDLt configuration")
        n, cp, r, v, _ = sy.elasticity_3d(N)
        perm, _ = sy.order_grid_dof(N, 3)
        ftype = COMPLEXDOUBLE
    else:
        facto = {"llt": 0, "ldlt": 1, "lu": 2}[facto_name]
        n, cp, r, v = sy.laplacian_3d(N, full=(facto == 2))
        perm, _ = sy.order_grid(N, N, N)
        ftype = 0 if f32 else 1                    # IPARM_FLOAT: API_REALSINGLE / API_REALDOUBLE
    t_matrix = time.time() - t0                    # (the synthetic matrix and its geometric ordering: the bench's input)
    t0 = time.time()
    s = sy.symbolic(n, cp, r, perm, max_blocksize=blocksize)
    c4, b4 = s["cblk4"], s["blok4"]
    flops = fact_flops(c4, b4, facto, ftype)
    t_sym = time.time() - t0
    t0 = time.time()
    plan = Plan(c4, b4, facto, floattype=ftype, device=local, lookahead=chunk, run_schedule=run_schedule)
    t_plan = time.time() - t0
    crit = (1e-12 if zel else 6.0 * 2 * np.sqrt(1e-31))
    t0 = time.time()
    plan.fill_csc(0 if facto == 2 else 1, n, cp, r, v, s["perm"])
    t_fill = time.time() - t0
    for _ in range(warmup):
        plan.refill()
        plan.factorize(crit)
    torch.cuda.synchronize()
    t0 = time.time()
    ft = ut = uts = urt = rnt = 0.0
    nrun = 0            # steps whose thin levels ran as the run launch (a step whose run stopped is redone level by level: run_time 0)
    run_flops = 0.0
    st = None
    watch = PowerWatch(enabled=power)
    with watch:
      for _ in range(steps):
        plan.refill()
        st = plan.factorize(crit)
        ft += st["fact_time"]
        ut += st["update_time"]
        uts += st["update_time_sum"]
        urt += st["urgent_time_sum"]
        rnt += st["run_time"]
        nrun += st["run_time"] > 0
        run_flops = max(run_flops, st["run_flops"])
      torch.cuda.synchronize()
      wall = time.time() - t0
    # end-to-end check on the last factorization: ||Ax-b||/||b|| with the device solve
    rng = np.random.default_rng(1)
    b = rng.random(n) + (1j * rng.random(n) if zel else 0)
    bp = np.empty(n, dtype=b.dtype)
    bp[s["perm"]] = b
    t1 = time.time()
    x = plan.solve(bp)[s["perm"]]
    solve_s = time.time() - t1          # host vector in -> host vector out (first call also builds the solve tables)
    t1 = time.time()
    plan.solve(bp.copy())
    solve_s = min(solve_s, time.time() - t1)
    import scipy.sparse as sp
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    Ax = A @ x if facto == 2 else A @ x + sp.tril(A, -1).T @ x
    resid = float(np.linalg.norm(Ax - b) / np.linalg.norm(b))
    ps = plan.stats()
    res = dict(wall=wall, flops=flops, fact_time=ft, update_time=ut, update_time_sum=uts, urgent_time_sum=urt, urgent_flops=st["urgent_flops"],
               nurgent=st["nurgent_launches"], update_flops=ps["update_flops"],
               update_bytes=ps["update_bytes"],
               run_time=rnt, run_steps=nrun, run_flops=run_flops, run_tickets=st["run_tickets"], run_first_level=st["run_first_level"],
               nlaunch=st["nupdate_launches"], solve_s=solve_s, solve_dev_s=ps["solve_time"], resid=resid, nbpivot=st["nbpivot"], n=n, cblk=len(c4) - 1,
               blok=len(b4), nnzl=s["nnzl"], coefnbr=ps["coefnbr"], t_matrix=t_matrix, t_sym=t_sym, t_plan=t_plan, t_fill=t_fill,
               ntasks=ps["ntasks"], npieces=ps["npieces"], nlevels=ps["nlevels"], parallelism="single-gpu", facto=facto_name,
               power=watch.summary())
    plan.close()
    del plan, A, Ax, x, b, bp, s, c4, b4
    import gc
    gc.collect()
    return res


# BASELINE.json configs[1], [2], [4] beside the headline (configs[3] at N = 1): run AFTER the headline's timed region and
# reported in "other_configs"; the headline's numbers are untouched.  configs[2] is 200^3 dLU, which needs 2 x 150 GB
# of panels: 192^3 (2 x 127 GB) is the largest grid that fits one 288 GB device, and is named as such.
OTHER_CONFIGS = [
    dict(config="configs[1]", workload="laplacian", grid=100, facto="llt", steps=3, warmup=1),
    dict(config="configs[2] at the largest grid one GPU holds (200^3 dLU: 2 x 150 GB of panels + tables > 288 GB)",
         workload="laplacian", grid=192, facto="lu", steps=2, warmup=1),
    dict(config="configs[4]", workload="elasticity", grid=48, facto="ldlt", steps=3, warmup=1),
]


def one_shot_cost(grid, blocksize, local):
    """What a drop-in caller pays (configs[1] through pastix_amd_d_po_sopalin = D_po_sopalin_thread: host panels in, host
    panels out): the first call analyses the layout and allocates, later calls on the same layout reuse the cached plan."""
    from pastix_amd.solver import sopalin_tabs
    import numpy as np
    from pastix_amd import _lib, Plan
    from pastix_amd import symbolic as sy
    n, cp, r, v = sy.laplacian_3d(grid)
    perm, _ = sy.order_grid(grid, grid, grid)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=blocksize)
    c4, b4 = s["cblk4"], s["blok4"]
    with Plan(c4, b4, 0, device=local) as p:         # the input panels as CoefMatrix_Init leaves them on the host
        p.fill_csc(1, n, cp, r, v, s["perm"])
        L0 = p.download()[0]
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    calls = []
    for _ in range(3):
        tabs = [L0[off[k]:off[k + 1]].copy() for k in range(len(w))]
        t0 = time.time()
        st = sopalin_tabs(0, c4, b4, tabs, critere=1e-14)
        calls.append({"wall_s": round(time.time() - t0, 4), "plan_s": round(st["plan_time"], 4), "h2d_s": round(st["h2d_time"], 4),
                      "fact_s": round(st["fact_time"], 4), "d2h_s": round(st["d2h_time"], 4)})
        del tabs
    _lib.lib().pastix_amd_release_cached_plan()
    return {"entry": "pastix_amd_d_po_sopalin (= D_po_sopalin_thread), 3-D Laplacian %d^3 dLLt, host panels in / out" % grid,
            "panel_bytes": int(8 * off[-1]), "first_call": calls[0], "later_calls": calls[1:],
            "one_shot_s": min(c["wall_s"] for c in calls[1:])}


def other_configs(blocksize, local):
    out = []
    for c in OTHER_CONFIGS:
        t0 = time.time()
        try:
            r = single_gpu_job(c["grid"], c["workload"], c["facto"], c["steps"], c["warmup"], blocksize, 0, local)
        except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            out.append({"config": c["config"], "error": repr(e)[:300]})
            continue
        K = c["steps"]
        zel = c["workload"] == "elasticity"
        bulk_flops = r["update_flops"] - r["urgent_flops"]
        out.append({
            "config": c["config"],
            "workload": ("3-dof elasticity pattern on %d^3 nodes (n=%d), complex double symmetric LDLt" if zel else
                         "3-D 7-point Laplacian %%d^3 (n=%%d), double %s" % {"llt": "LLt", "ldlt": "LDLt", "lu": "LU static pivoting"}[c["facto"]])
                        % (c["grid"], r["n"]),
            "dtype": "c128 (f64 MFMA on split re/im planes)" if zel else "f64",
            "value": round(r["flops"] * K / r["wall"] * 1e-9, 1), "unit": "GFLOP/s" + (" (complex flops)" if zel else ""),
            "steps": K, "warmup": c["warmup"], "ms_per_step": round(r["wall"] / K * 1e3, 2),
            "pct_of_mfma_f64_peak": round(r["flops"] * K / r["wall"] / MFMA_F64_PEAK * 100, 2),
            # the dominant kernel: the run launch where the run schedule is on, else the bulk launches of the levels
            "roofline_frac": round((r["run_flops"] * r["run_steps"] / r["run_time"] if r["run_time"] > 0 else
                                    bulk_flops * K / max(r["update_time_sum"], 1e-12)) / MFMA_F64_PEAK, 4),
            "roofline_kernel": "k_run_update" if r["run_time"] > 0 else "k_update<0>",
            "steps_redone_level_by_level": K - r["run_steps"] if r["run_time"] > 0 else 0,
            "residual": r["resid"], "static_pivots": r["nbpivot"], "fact_flops": r["flops"],
            "solve_device_s": round(r["solve_dev_s"], 4), "total_s_incl_analysis": round(time.time() - t0, 1)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=int(os.environ.get("PASTIX_AMD_BENCH_GRID", "200")))
    ap.add_argument("--blocksize", type=int, default=128)
    ap.add_argument("--chunk", type=int, default=0, help="update-schedule chunk (0 = engine default)")
    ap.add_argument("--run-schedule", type=int, default=0, choices=[-1, 0, 1],
                    help="options.run_schedule: 0 engine default (the dependency-driven launch up to 2e14 flop), 1 build it whatever "
                         "the size (200^3: +1.3 ... 1.6 %% for 2.5 s more analysis), -1 level by level only")
    ap.add_argument("--facto", choices=["llt", "ldlt", "lu"], default="llt")
    ap.add_argument("--workload", choices=["laplacian", "elasticity"], default="laplacian",
                    help="laplacian: 3-D 7-point Laplacian grid^3, double (the metric); elasticity: BASELINE configs[4], "
                         "complex double LDLt on the 3-dof elasticity pattern of a grid^3 node mesh (n = 3 grid^3)")
    ap.add_argument("--cpu-sample-grid", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample rocm-smi (socket power, shader clock) during the timed steps")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f32: the single-precision engine (the reference's S_ build; kernels_f32.hip), reported against the "
                         "fp32 matrix peak; one GPU, laplacian workload")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip BASELINE.json configs[1], [2], [4] after the headline (they run by default with the "
                         "default workload on one GPU)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # n_gpus in the JSON line is the number of ranks that ran; a mismatch would mislabel the measurement
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run "
                         "--nproc-per-node %d, or run `python bench.py --gpus %d` and let it launch the ranks)"
                         % (a.gpus, world, a.gpus, a.gpus))
    if world > 1:
        # every rank drives 2 compute streams + one channel stream per peer: give each its own hardware queue (the HIP
        # runtime multiplexes streams over 4 by default; a channel waiting for its peer must not stall a compute stream)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        # a rank whose streams have not drained by then reports the first unmatched fan-in block of every channel,
        # aborts its communicators and leaves with a non-zero code (csrc/dist.cpp: dist_finish)
        os.environ.setdefault("PASTIX_AMD_DIST_TIMEOUT", "120")
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # PASTIX_AMD_DIST_TEST=1 (validation on a 1-GPU box only): ranks share the visible GPUs and the fan-in
    # messages go through gloo (host staged) -- everything but the RCCL calls themselves is exercised.
    dist_test = os.environ.get("PASTIX_AMD_DIST_TEST") == "1"
    if dist_test:
        local = local % torch.cuda.device_count()
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) on this box: one rank per GPU" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # (the bootstrap collectives -- layout broadcast, schedule hashes, unique ids, barriers -- get a deadline too)
        tmo = datetime.timedelta(seconds=600)
        if dist_test:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            # RCCL's copy kernels go on a high-priority stream: they must get a CU slot between the bulk update
            # workgroups of the second stream, like the panel kernels do
            try:
                pgo = dist.ProcessGroupNCCL.Options()
                pgo.is_high_priority_stream = True
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=pgo, timeout=tmo)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)

    if world > 1:
        from pastix_amd import dist as pdist
        res = pdist.bench_distributed(a, rank, world, local)
    else:
        if a.dtype == "f32" and a.workload != "laplacian":
            raise SystemExit("bench.py: --dtype f32 is the real single-precision engine (laplacian workload)")
        res = single_gpu_job(a.grid, a.workload, a.facto, a.steps, a.warmup, a.blocksize, a.chunk, local, f32=a.dtype == "f32",
                             power=not a.no_power, run_schedule=a.run_schedule)
        a.facto = res["facto"]

    if rank == 0:
        PEAK = MFMA_F32_PEAK if (world == 1 and a.dtype == "f32") else MFMA_F64_PEAK
        K = a.steps
        value = res["flops"] * K / res["wall"] * 1e-9
        # HBM bytes of the bulk kernel from separate --pmc passes (tools/profile_round.sh); only a measurement taken
        # on exactly these engine sources counts, anything else is stale -> null
        traffic, traffic_src = (None, None)
        if world == 1:
            traffic, traffic_src = measured_traffic(a.grid, a.facto, a.blocksize, a.dtype, a.chunk, a.workload)
            if a.run_schedule != 0:
                traffic, traffic_src = None, None          # (the counters were collected on the default schedule)
        # Dominant kernel: k_update<0>, the bulk contribution launches.  achieved = its flops / the sum of its
        # launches' durations (HIP events around every launch, on the stream it is launched on) = what
        # rocprofv3 --kernel-trace --stats reports for that kernel.  The few urgent tasks of every level run as
        # k_update<1> on the other stream, beside the previous slot's bulk launch; they are reported apart.
        ut_sum = res.get("update_time_sum", res["update_time"])
        bulk_flops = res["update_flops"] - res.get("urgent_flops", 0.0)
        upd_rate = bulk_flops * K / max(ut_sum, 1e-12)
        # With the run schedule (engine default where it is built: real double LLt) the dominant kernel is ONE launch,
        # k_run_update: every update task of the thin levels -- all but the first few levels of the tree --, gated by
        # dependency counters.  Its duration includes whatever time its workgroups found nothing ready.  The levels below
        # it keep their per-level launches (k_update<0> bulk, k_update<1> urgent), reported apart.
        # (a step whose run stopped is redone level by level and has no run launch: the run's rate is taken over the steps
        # that had one, and only if all of them did is the launch the line's dominant kernel)
        run_on = res.get("run_time", 0.0) > 0 and res.get("run_steps", K) == K
        lvl_launches = None
        if run_on:
            nl_lv = max(res["nlaunch"] - 1, 0)
            lvl_launches = {"kernel": "k_update<0>", "levels": "0..%d" % (res["run_first_level"] - 1), "launches_per_step": nl_lv,
                            "share_of_update_flops": round((bulk_flops - res["run_flops"]) / max(res["update_flops"], 1.0), 4),
                            "avg_launch_ms": round((ut_sum - res["run_time"]) / K / max(nl_lv, 1) * 1e3, 4),
                            "achieved": round((bulk_flops - res["run_flops"]) * K / max(ut_sum - res["run_time"], 1e-12) * 1e-12, 3)}
            bulk_flops = res["run_flops"]
            ut_sum = res["run_time"]
            upd_rate = bulk_flops * K / max(ut_sum, 1e-12)
            res["nlaunch"] = 1
        busy_rate = res["update_flops"] * K / max(res["update_time"], 1e-12)
        esz = 16.0 if a.workload == "elasticity" else 4.0 if PEAK == MFMA_F32_PEAK else 8.0
        compulsory = 2.0 * esz * res["coefnbr"] * (2 if a.facto in ("ldlt", "lu") or a.workload == "elasticity" else 1)
        out = {
            "metric": ("factorization GFLOP/s (complex flops), 3-dof elasticity pattern %d^3 nodes zLDLt" % a.grid) if a.workload == "elasticity"
                      else "factorization GFLOP/s, 3D 7-point Laplacian %d^3 %s%s" % (a.grid, "s" if PEAK == MFMA_F32_PEAK else "d",
                                                                                      {"llt": "LLt", "ldlt": "LDLt", "lu": "LU"}[a.facto]),
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": round(res["wall"] / K * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "c128 (f64 MFMA on split re/im planes)" if a.workload == "elasticity" else ("f32" if PEAK == MFMA_F32_PEAK else "f64"), "data": "synthetic",
            "config": {"workload": ("3-dof elasticity pattern on %d^3 nodes (n=%d), complex double symmetric %s, geometric ND, max blocksize %d"
                                    if a.workload == "elasticity" else
                                    "3-D 7-point Laplacian %d^3 (n=%d), " + ("single" if PEAK == MFMA_F32_PEAK else "double")
                                    + " %s, geometric ND, max blocksize %d")
                                   % (a.grid, res["n"], a.facto, a.blocksize),
                       "cblknbr": res["cblk"], "bloknbr": res["blok"], "nnzL": res["nnzl"],
                       "fact_flops": res["flops"], "parallelism": res["parallelism"], "run_schedule_option": a.run_schedule,
                       ("pct_of_mfma_f32_peak" if PEAK == MFMA_F32_PEAK else "pct_of_mfma_f64_peak"): round(value * 1e9 / (PEAK * world) * 100, 2),
                       "fact_time_s_per_step": round(res["fact_time"] / K, 4),
                       "residual": res["resid"], "solve_s": round(res["solve_s"], 4) if "solve_s" in res else None, "logdet_rel_err": res.get("logdet_rel_err"),
                       "static_pivots": res["nbpivot"],
                       "steps_redone_level_by_level": (K - res.get("run_steps", K)) if res.get("run_time", 0.0) > 0 else 0,
                       "power": res.get("power"),
                       "analysis_s": {"symbolic": round(res["t_sym"], 2), "plan": round(res["t_plan"], 2),
                                      "fill_prepare": round(res["t_fill"], 2), "input_matrix": round(res.get("t_matrix", 0.0), 2)}},
            "roofline": {"bound": "mfma", "kernel": "k_update_s" if PEAK == MFMA_F32_PEAK else "k_run_update" if run_on else "k_update",
                         "achieved": round(upd_rate * 1e-12, 3),
                         "peak": PEAK * 1e-12, "unit": "TFLOP/s",
                         "frac": round(upd_rate / PEAK, 4),
                         "traffic": None if traffic is None else traffic / max(res["nlaunch"], 1),
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": res.get("update_bytes", 0.0) * (bulk_flops / max(res["update_flops"], 1.0)) / max(res["nlaunch"], 1),
                         # the other definition of the path's bytes: every panel entry read once and written once
                         # per factorization, 2 * sizeof(element) * coefnbr (the arenas of the variant counted)
                         "compulsory_bytes_per_step": compulsory,
                         "traffic_over_compulsory": None if traffic is None else round(traffic / max(compulsory, 1.0), 2),
                         "traffic_over_algorithmic": None if traffic is None else round(
                             traffic / max(res.get("update_bytes", 0.0) * (bulk_flops / max(res["update_flops"], 1.0)), 1.0), 3),
                         "launches_per_step": res["nlaunch"],
                         "avg_launch_ms": round(ut_sum / K / max(res["nlaunch"], 1) * 1e3, 4),
                         "achieved_while_in_flight": round(busy_rate * 1e-12, 3),
                         "flops_per_launch": bulk_flops / max(res["nlaunch"], 1),
                         "run_tickets": res.get("run_tickets", 0) if run_on else None,
                         "level_launches": lvl_launches,
                         "urgent_launches": {"kernel": "k_update<1>", "launches_per_step": res.get("nurgent", 0),
                                             "share_of_update_flops": round(res.get("urgent_flops", 0.0) / max(res["update_flops"], 1.0), 4),
                                             "avg_launch_ms": round(res.get("urgent_time_sum", 0.0) / K / max(res.get("nurgent", 0), 1) * 1e3, 4)}},
        }
        if res.get("solve_dev_s"):
            # the next row of the path: forward + backward sweep, HBM-bound -- every panel entry is read
            # once per sweep (LU: L forward, U backward)
            sb = 2.0 * (16.0 if a.workload == "elasticity" else 4.0 if PEAK == MFMA_F32_PEAK else 8.0) * res["nnzl"]
            out["solve"] = {"bound": "hbm", "achieved": round(sb / res["solve_dev_s"] * 1e-9, 1), "peak": 8000.0,
                            "unit": "GB/s", "frac": round(sb / res["solve_dev_s"] / 8e12, 4),
                            "device_s": round(res["solve_dev_s"], 4), "host_to_host_s": round(res["solve_s"], 4),
                            "panel_bytes_per_solve": sb, "nrhs": 1}
        if (world == 1 and not a.no_other_configs and a.grid == 200 and a.workload == "laplacian" and a.facto == "llt"
                and a.chunk == 0 and a.dtype == "f64"):
            out["other_configs"] = other_configs(a.blocksize, local)
            try:
                out["config"]["one_shot"] = one_shot_cost(100, a.blocksize, local)
                out["config"]["one_shot_s"] = out["config"]["one_shot"]["one_shot_s"]
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                out["config"]["one_shot"] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample_grid, min(os.cpu_count() or 1, 64))
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _main_guarded():
    """Ranks of a multi-GPU job leave through os._exit with the code they earned: 0 after a complete run, 1 after ANY
    exception (printed first).  A rank that failed may hold streams its aborted channels never drained, and a peer may
    be waiting in a collective: interpreter teardown (torch / RCCL destructors, atexit hooks) must not get the chance
    to hang the job or to turn a failure into exit code 0.  The engine itself keeps one librccl per process
    (csrc/dist.cpp: rccl()), which is what the exit-time aborts of the first rounds came from."""
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    try:
        main()
        code = 0
    except SystemExit as e:
        if not multi:
            raise
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code and not isinstance(e.code, int):
            sys.stderr.write(str(e.code) + "\n")
    except BaseException:  # noqa: BLE001
        if not multi:
            raise
        import traceback
        traceback.print_exc()
        code = 1
    if multi:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)


if __name__ == "__main__":
    _main_guarded()
