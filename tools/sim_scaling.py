#!/usr/bin/env python3
"""Host-only estimate of multi-GPU scaling from the per-rank schedules (no GPU needed).

Two models of the same plans (pastix_amd_plan_profile per rank + the fan-in messages of dist.py):

  lockstep : round 1's schedule -- all ranks walk the levels together, a level with fan-in traffic adds
             t_msg + (largest per-rank bytes of the level) / BW_rank.
  async    : the dependency-driven schedule of the native engine (pastix_amd_factorize_dist): every rank runs its own
             two streams; per level l on rank q
                 A(l) = urgent contributions   starts after P(l-1) and B(l-1)              (panel stream)
                 sends of level-l fan-in blocks leave right after A(l) (one channel per peer: rendezvous with the
                 owner, which posts the receive after its own A(l); a message costs t_msg + bytes / BW_link)
                 P(l) = diag + panel solve      starts after A(l) and the arrival (+ add) of every block for level l
                 B(l) = bulk contributions      starts after P(l-1) and B(l-1)               (second stream)
             nothing else couples the ranks.
A launch of F flops whose largest task has W multiply-adds takes max(F / R(F), 2 W / R_wg) + t0, where R(F) is the MEASURED
rate of a bulk launch of that size on one MI355X (profiles/rNN/launch_rate_curve.json, tools/launch_curve.py: 4 TFLOP/s at 1e8
flop ... 63 at 5e11) when that file exists (argv[3] or the newest profiles/r*/launch_rate_curve.json), else a flat 60 TFLOP/s --
which is what round 3's 6.35x at 8 GPUs assumed although an eighth of a 200^3 launch runs at 45-56.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pastix_amd import dist as pd  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
RANKS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
R_chip, R_wg, t0 = 60e12, 60e12 / 400, 30e-6
R_panel = R_chip / 4                       # diag + panel-solve kernels
t_msg = 40e-6                              # per message: launch + rendezvous of a send/recv pair
BW_link = 120e9                            # one xGMI link, achieved (153 GB/s peak)
BW_rank = 150e9                            # lockstep model only: per-rank aggregate assumed in round 1
t_lock = 80e-6                             # lockstep model only: per-level exchange latency
HBM = 4e12                                 # owner-side add of a received block (read block + RMW panel)

n, cp, r, v = sy.laplacian_3d(N)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
level = pd.levels_of(c4, b4)
w = (c4[:-1, 1] - c4[:-1, 0] + 1)
cb = np.repeat(np.arange(len(c4) - 1), np.diff(c4[:, 2]))
h = b4[:, 1] - b4[:, 0] + 1


import glob  # noqa: E402
import json  # noqa: E402
curve_file = sys.argv[3] if len(sys.argv) > 3 else (sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                  "profiles", "r*", "launch_rate_curve.json"))) or [None])[-1]
CURVE = None
if curve_file and os.path.exists(curve_file):
    cj = json.load(open(curve_file))["curve"]
    CURVE = (np.array([c["log10_flops"] for c in cj]), np.array([c["median_TFLOPs"] * 1e12 for c in cj]))
    print("launch rate curve: %s (%d bins, %.1f ... %.1f TFLOP/s)" % (curve_file, len(cj), CURVE[1].min() * 1e-12, CURVE[1].max() * 1e-12))
else:
    print("launch rate curve: none found, flat %.0f TFLOP/s" % (R_chip * 1e-12))


def rate(F):
    if CURVE is None:
        return R_chip
    return float(np.interp(np.log10(max(F, 1.0)), CURVE[0], CURVE[1]))


def launch(F, W):
    return max(F / rate(F), 2 * W / R_wg) + t0 if F > 0 else 0.0


base = {}
for P in RANKS:
    owner = pd.partition(c4, b4, P)
    per = [pd.plan_profile(c4, b4, owner if P > 1 else None, q) for q in range(P)]
    nl = len(per[0][0])
    # messages: (level, sender, owner, bytes)
    msgs = [[] for _ in range(nl)]
    comm = np.zeros((nl, P, 2))
    if P > 1:
        mask = pd.fanin_touched(c4, b4, owner)
        for q in range(P):
            byt = 8.0 * h * w[cb] * ((mask >> np.uint64(q)) & np.uint64(1))
            per_t = np.bincount(cb, weights=byt, minlength=len(w))
            for t in np.nonzero(per_t)[0]:
                msgs[level[t]].append((q, int(owner[t]), per_t[t]))
            np.add.at(comm[:, q, 0], level[cb], byt)
            np.add.at(comm[:, :, 1], (level[cb], owner[cb]), byt)
    # ---- lockstep ----
    tot = tcomm = 0.0
    for l in range(nl):
        if comm[l].max() > 0:
            tc = t_lock + comm[l].max() / BW_rank
            tot += tc
            tcomm += tc
        tl = 0.0
        for sf, sm, stn, pf, uf in per:
            t = launch(sf[l], sm[l]) if stn[l] > 0 else 0.0
            if pf[l] > 0:
                t += pf[l] / R_panel + 2 * t0
            tl = max(tl, t)
        tot += tl
    # ---- async ----
    TP = np.zeros(P)      # panel stream: end of P(l-1)
    TB = np.zeros(P)      # bulk stream: end of B(l-1)
    wait_msg = np.zeros(P)
    for l in range(nl):
        TA = np.zeros(P)
        for q, (sf, sm, stn, pf, uf) in enumerate(per):
            a = launch(uf[l], min(sm[l], uf[l] / 2)) if uf[l] > 0 else 0.0
            TA[q] = max(TP[q], TB[q]) + a if uf[l] > 0 else TP[q]
        arrive = TA.copy()
        link_busy = {}
        for (src, dst, byt) in sorted(msgs[l], key=lambda m: TA[m[0]]):
            start = max(TA[src], TA[dst], link_busy.get((src, dst), 0.0))
            end = start + t_msg + byt / BW_link
            link_busy[(src, dst)] = end
            arrive[dst] = max(arrive[dst], end + byt * 3 / HBM)
        for q, (sf, sm, stn, pf, uf) in enumerate(per):
            b = launch(sf[l] - uf[l], sm[l]) if sf[l] - uf[l] > 0 else 0.0
            startB = max(TP[q], TB[q])
            wait_msg[q] += max(0.0, arrive[q] - TA[q]) if pf[l] > 0 else 0.0
            if pf[l] > 0:
                TP[q] = max(TA[q], arrive[q]) + pf[l] / R_panel + 2 * t0
            else:
                TP[q] = max(TA[q], arrive[q])
            TB[q] = startB + b if b > 0 else TB[q]
    tasync = float(max(TP.max(), TB.max()))
    if P == RANKS[0]:
        base = {"lock": tot, "async": tasync}
    print("P=%d levels=%d  lockstep %.4f s (fan-in %.4f s) speedup %.2f | async %.4f s speedup %.2f "
          "(panel stream waiting for fan-in blocks: max %.4f s) | busiest rank sends %.2f GB, flops share max %.3f"
          % (P, nl, tot, tcomm, base["lock"] / tot, tasync, base["async"] / tasync, wait_msg.max(),
             comm[:, :, 0].sum(axis=0).max() * 1e-9, max(p[0].sum() for p in per) / sum(p[0].sum() for p in per)),
          flush=True)
