#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: average counter value per kernel."""
import csv
import sys
from collections import defaultdict

def main(paths, only=None):
    agg = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            if only and only not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")

if __name__ == "__main__":
    only = None
    args = sys.argv[1:]
    if args and args[0].startswith("--only="):
        only = args.pop(0)[7:]
    main(args, only)
