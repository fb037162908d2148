#!/usr/bin/env python3
"""Fold the PMC sums of tools/profile_round.sh (gpurun_out/profile_<grid>/) into profiles/rNN/traffic_k_update.json.
usage: make_traffic_json.py rNN GRID [GRID ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
out = os.path.join(ROOT, "profiles", rnd, "traffic_k_update.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
d = json.load(open(out)) if os.path.exists(out) else {}
d["_doc"] = ("HBM-side traffic of k_update from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no other tracing, "
             "one factorization each: tools/profile_round.sh -> bench.py --steps 1 --warmup 0).  Counters are KiB summed over "
             "all launches of the bulk kernel k_update<8, 0>.  bytes_per_factorization = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
             "FETCH_SIZE is doubled (gfx950 reports half of coalesced reads; calibrated on an 8-B/lane streaming kernel, "
             "tools/probe_mfma_f64.hip k_rmw: 4 GiB read -> FETCH_SIZE 2 GiB; WRITE_SIZE exact).  source_sha = "
             "bench.engine_source_sha() of the engine sources the counters were collected on; bench.py reports the "
             "figure only while the sources still hash to it.")
for g in sys.argv[2:]:
    src = os.path.join(ROOT, "gpurun_out", "profile_%s" % g)
    f = json.load(open(os.path.join(src, "sum_FETCH_SIZE.json")))
    w = json.load(open(os.path.join(src, "sum_WRITE_SIZE.json")))
    sha = open(os.path.join(src, "source_sha.txt")).read().strip()
    d[str(g)] = {"kernel": f["kernel"], "launches": f["launches"], "fetch_kib_raw": f["FETCH_SIZE"],
                 "write_kib": w["WRITE_SIZE"], "bytes_per_factorization": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0,
                 "source_sha": sha}
json.dump(d, open(out, "w"), indent=1)
print(open(out).read())
