#!/usr/bin/env python3
"""The multi-GPU driver at scale on ONE GPU: WORLD emulated ranks (loopback transport: one host thread per rank,
device-to-device copies instead of xGMI; everything else -- partition, per-rank plans, fan-in buffers, channel order,
adds, deadline handling -- is the code the RCCL job runs) factorize GRID^3 dLLt, solve, and are checked with the
size-independent properties bench.py --gpus N uses: log det A against the analytic spectrum and ||Ax - b|| / ||b||.
Writes one JSON object (per-rank message counts, bytes, staging, fan-in buffers, device times) to stdout.
usage: loopback_scale.py GRID WORLD [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pastix_amd import fact_flops  # noqa: E402
from pastix_amd import dist as pd  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0 = time.time()
n, cp, r, v = sy.laplacian_3d(N)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
fl = fact_flops(c4, b4, 0)
t_sym = time.time() - t0
owner = pd.partition(c4, b4, W)
table = [pd.schedule_hashes(c4, b4, owner, q, W) for q in range(W)]
assert pd.mismatched_channels(table) == []
t0 = time.time()
plans = [pd.DistPlan(c4, b4, owner, q, 0) for q in range(W)]
t_plan = time.time() - t0
pd.attach_local(plans)
for q in plans:
    q.fill_csc(1, n, cp, r, v, s["perm"])
crit = 6.0 * 2 * np.sqrt(1e-31)
pd.factorize_local(plans, crit)
walls, sts = [], None
for _ in range(reps):
    for q in plans:
        q.refill()
    t0 = time.time()
    sts = pd.factorize_local(plans, crit)
    walls.append(time.time() - t0)
ld = 2.0 * sum(q.diag_logsum() for q in plans)
cs = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
exact = float(np.log(6.0 - cs[:, None, None] - cs[None, :, None] - cs[None, None, :]).sum())
rng = np.random.default_rng(1)
b = rng.random(n)
bp = np.empty(n)
bp[s["perm"]] = b
x = pd.solve_local(plans, bp)[s["perm"]]
import scipy.sparse as sp  # noqa: E402
A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
resid = float(np.linalg.norm(A @ x + sp.tril(A, -1).T @ x - b) / np.linalg.norm(b))
infos = [q.info() for q in plans]
stats = [q.stats() for q in plans]
out = {
    "what": "multi-GPU driver, loopback transport, %d emulated ranks on one MI355X" % W,
    "workload": "3-D 7-point Laplacian %d^3 (n=%d), double LLt, geometric ND, max blocksize 128" % (N, n),
    "fact_flops": fl, "world": W, "nlevels": stats[0]["nlevels"],
    "wall_s": [round(w, 4) for w in walls],
    "tflops_one_gpu_shared_by_all_ranks": round(fl / min(walls) * 1e-12, 2),
    "logdet_rel_err": abs(ld - exact) / abs(exact), "residual": resid,
    "static_pivots": int(sum(st["nbpivot"] for st in sts)),
    "analysis_s": {"symbolic": round(t_sym, 2), "plans_all_ranks": round(t_plan, 2)},
    "ranks": [{"rank": q, "share_of_flops": round(stats[q]["local_flops"] / fl, 4), "npeers": infos[q]["npeers"],
               "nsend": infos[q]["nsend"], "nrecv": infos[q]["nrecv"],
               "bytes_sent": infos[q]["bytes_sent"], "bytes_recv": infos[q]["bytes_recv"],
               "stage_elems": int(infos[q]["staging_bytes"] / 8), "fanin_buffer_bytes": infos[q]["fanin_buffer_bytes"],
               "owned_panel_bytes": 8.0 * stats[q]["coefnbr"] - infos[q]["fanin_buffer_bytes"],
               "device_fact_s": round(sts[q]["fact_time"], 4)} for q in range(W)],
}
assert out["logdet_rel_err"] < 1e-10 and resid < 1e-10 and out["static_pivots"] == 0, out
assert sum(i["nsend"] for i in infos) == sum(i["nrecv"] for i in infos)
print(json.dumps(out, indent=1))
for q in plans:
    q.close()
