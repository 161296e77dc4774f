#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, average and max of each counter per dispatch.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]"""
import csv, glob, os, sys
from collections import defaultdict

for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
    for cname, per in acc.items():
        print(f"## {cname}   ({d})")
        print(f"{'kernel':62s} {'calls':>6s} {'avg':>14s} {'max':>14s}")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k:62s} {len(v):6d} {sum(v) / len(v):14.1f} {max(v):14.1f}")
        print()
