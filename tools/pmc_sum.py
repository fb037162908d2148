#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.

usage: pmc_sum.py <dir with *_counter_collection.csv> [kernel-substring]
Prints {"kernel": ..., "launches": n, "<COUNTER>": sum, ...} as JSON.
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_update"
    tot, disp = {}, set()
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if sub not in row["Kernel_Name"]:
                    continue
                disp.add((f, row["Dispatch_Id"]))
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {"kernel": sub, "launches": len(disp)}
    out.update(tot)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
