#!/usr/bin/env python3
"""Fold the counter sums of tools/profile_round.sh (gpurun_out/profile_<tag>/) into profiles/rNN/:
   traffic_k_update.json   HBM bytes of the bulk update kernels per factorization (what bench.py's roofline.traffic reads)
   pmc_<tag>.json          the same + MFMA-pipe busy %, CU busy %, L2 hit rate, HBM GB/s against the 8 TB/s peak
usage: make_profile_json.py rNN TAG [TAG ...]      (TAG = 200, 100, 48workloadelasticity, ...)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
pdir = os.path.join(ROOT, "profiles", rnd)
os.makedirs(pdir, exist_ok=True)
tpath = os.path.join(pdir, "traffic_k_update.json")
d = json.load(open(tpath)) if os.path.exists(tpath) else {}
d["_doc"] = ("HBM-side traffic of the bulk update kernels (k_run_update + k_update<0> + k_update_small<0>) from rocprofv3 --pmc FETCH_SIZE / "
             "WRITE_SIZE (separate passes, no other tracing, one factorization each: tools/profile_round.sh -> bench.py "
             "--steps 1 --warmup 0).  Counters are KiB summed over all launches.  bytes_per_factorization = (2*FETCH_SIZE + "
             "WRITE_SIZE)*1024: FETCH_SIZE is doubled (gfx950 reports half of coalesced reads; calibrated on an 8-B/lane "
             "streaming kernel, tools/probe_mfma_f64.hip k_rmw: 4 GiB read -> FETCH_SIZE 2 GiB; WRITE_SIZE exact).  source_sha "
             "= bench.engine_source_sha() of the engine sources the counters were collected on; bench.py reports the figure "
             "only while the sources still hash to it.")


def load(p):
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) > 2 else None


for tag in sys.argv[2:]:
    src = os.path.join(ROOT, "gpurun_out", "profile_%s" % tag)
    f, w = load(os.path.join(src, "sum_FETCH_SIZE.json")), load(os.path.join(src, "sum_WRITE_SIZE.json"))
    sha = open(os.path.join(src, "source_sha.txt")).read().strip()
    out = {"tag": tag, "source_sha": sha, "kernels": f["kernel"] if f else None}
    # duration of the same kernels in the kernel-trace pass
    kt = 0.0
    with open(os.path.join(src, "kernel_stats.csv"), newline="") as fh:
        for row in csv.DictReader(fh):
            if "k_update<0>" in row["Name"] or "k_update_small<0>" in row["Name"] or "k_run_update" in row["Name"]:
                kt += float(row["TotalDurationNs"]) * 1e-9
    out["kernel_time_s"] = kt
    if f and w:
        b = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0
        out["hbm"] = {"fetch_kib_raw": f["FETCH_SIZE"], "write_kib": w["WRITE_SIZE"], "bytes_per_factorization": b,
                      "GBps_while_kernels_run": b / kt * 1e-9 if kt else None, "frac_of_8TBps": b / kt / 8e12 if kt else None}
        if tag.isdigit():
            d[tag] = {"kernel": f["kernel"], "launches": f["launches"], "fetch_kib_raw": f["FETCH_SIZE"], "write_kib": w["WRITE_SIZE"],
                      "bytes_per_factorization": b, "source_sha": sha}
    bz = load(os.path.join(src, "sum_busy.json"))
    if bz:
        out["busy"] = {"SQ_VALU_MFMA_BUSY_CYCLES": bz.get("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CU_CYCLES": bz.get("SQ_BUSY_CU_CYCLES"),
                       "SQ_INSTS_VALU_MFMA_MOPS_F64": bz.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
                       "mfma_busy_over_cu_busy": (bz["SQ_VALU_MFMA_BUSY_CYCLES"] / bz["SQ_BUSY_CU_CYCLES"])
                       if bz.get("SQ_BUSY_CU_CYCLES") else None}
    wv = load(os.path.join(src, "sum_waves.json"))
    if wv:
        out["waves"] = {k: wv.get(k) for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}
    l2 = load(os.path.join(src, "sum_l2.json"))
    if l2 and l2.get("TCC_REQ_sum"):
        out["l2"] = {"TCC_HIT_sum": l2.get("TCC_HIT_sum"), "TCC_MISS_sum": l2.get("TCC_MISS_sum"), "TCC_REQ_sum": l2.get("TCC_REQ_sum"),
                     "hit_rate": l2.get("TCC_HIT_sum", 0.0) / max(l2.get("TCC_HIT_sum", 0.0) + l2.get("TCC_MISS_sum", 0.0), 1.0)}
    # derived: the counters are sums over the chip's 8 XCDs x 32 CUs x 4 SIMDs and over the launches; GRBM_GUI_ACTIVE counts
    # per XCD, so GRBM / 8 = the GPU cycles the kernels were running
    if bz and wv and wv.get("GRBM_GUI_ACTIVE"):
        cyc = wv["GRBM_GUI_ACTIVE"] / 8.0
        out["derived"] = {"gpu_cycles_while_kernels_run": cyc,
                          "clock_GHz": cyc / kt * 1e-9 if kt else None,
                          "mfma_pipe_busy_frac": bz["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                          "cu_busy_frac": bz["SQ_BUSY_CU_CYCLES"] / (cyc * 256.0),
                          "mfma_pipe_busy_frac_inside_busy_cus": bz["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * bz["SQ_BUSY_CU_CYCLES"]),
                          # (SQ_WAVE_CYCLES counts in units of 4 cycles)
                          "waves_per_simd_avg": 4.0 * wv["SQ_WAVE_CYCLES"] / (cyc * 1024.0) if wv.get("SQ_WAVE_CYCLES") else None}
    json.dump(out, open(os.path.join(pdir, "pmc_%s.json" % tag), "w"), indent=1)
    print(json.dumps(out.get("derived"), indent=1))
json.dump(d, open(tpath, "w"), indent=1)
