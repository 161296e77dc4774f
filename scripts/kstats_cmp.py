#!/usr/bin/env python3
"""Side-by-side per-kernel average durations (us) of several rocprofv3 kernel_stats CSVs: kstats_cmp.py label=path ..."""
import csv, sys
cols = [a.split("=", 1) for a in sys.argv[1:]]
K = {l: {r['Name'][:56]: (float(r['AverageNs']) / 1e3, int(r['Calls'])) for r in csv.DictReader(open(p))} for l, p in cols}
first = cols[0][0]
names = sorted(K[first], key=lambda n: -(K[first][n][0] * K[first][n][1]))
print(f"{'kernel':56s} " + " ".join(f"{l:>7s}" for l, _ in cols))
for n in names[:int(40)]:
    print(f"{n:56s} " + " ".join(f"{K[l].get(n, (0, 0))[0]:7.1f}" for l, _ in cols))
