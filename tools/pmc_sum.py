#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters over the launches of some kernels.

usage: pmc_sum.py <dir with *_counter_collection.csv> [kernel-substring[,kernel-substring...]]
Prints {"kernel": ..., "launches": n, "<COUNTER>": sum, ...} as JSON.
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    subs = (sys.argv[2] if len(sys.argv) > 2 else "k_update").split(",")
    tot, disp = {}, set()
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if not any(s in row["Kernel_Name"] for s in subs):
                    continue
                disp.add((f, row["Dispatch_Id"]))
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {"kernel": ",".join(subs), "launches": len(disp)}
    out.update(tot)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
