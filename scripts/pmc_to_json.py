#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs -> profiles/pmc_kernels.json, the per-kernel counter file bench.py reads for `roofline.traffic`, `hbm_frac`,
VALUBusy and VALUUtilization (rocprofv3 cannot run inside bench.py).

usage: pmc_to_json.py OUT.json --workload scene=city,tris=1000000,width=1920,height=1080 --round 2 DIR [DIR ...]
Each DIR holds one `rocprofv3 --pmc <counters> -d DIR -- python bench.py --no-overlap ...` pass (FETCH_SIZE and WRITE_SIZE need separate
passes: MI355X_MICROARCH.md, "rocprofv3 PMC slots"). Per kernel: the per-dispatch AVERAGE of every counter found.

Units / corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by this rocprofv3;
FETCH_SIZE counts 64 B per 128-B fabric read request for wide coalesced streams, i.e. HALF the bytes: `fetch_bytes` below is the raw
counter x 1024, and `fetch_bytes_corrected` = 2 x that (the guide's gfx950 correction); the calibration kernel (`--calibrate NAME:BYTES`,
a device-to-device copy of known size run in the same pass) is reported beside it so the factor can be checked on this box."""
import csv, glob, json, os, sys
from collections import defaultdict

args = sys.argv[1:]
out = args.pop(0)
workload, rnd, dirs, calib = {}, None, [], None
while args:
    a = args.pop(0)
    if a == "--workload":
        for kv in args.pop(0).split(","):
            k, v = kv.split("=")
            workload[k] = int(v) if v.isdigit() else v
    elif a == "--round":
        rnd = int(args.pop(0))
    elif a == "--calibrate":
        n, b = args.pop(0).split(":")
        calib = (n, int(b))
    else:
        dirs.append(a)
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
for name, ctrs in acc.items():
    e = {"calls": max(len(v) for v in ctrs.values())}
    for c, v in ctrs.items():
        avg = sum(v) / len(v)
        if c == "FETCH_SIZE":
            e["fetch_bytes"] = avg * 1024.0
            e["fetch_bytes_corrected"] = avg * 2048.0
        elif c == "WRITE_SIZE":
            e["write_bytes"] = avg * 1024.0
        else:
            e[c] = avg
    kernels[name] = e
doc = {"_comment": "per-dispatch averages of rocprofv3 --pmc passes; see scripts/pmc_to_json.py for units and the FETCH_SIZE correction",
       "workload": workload, "round": rnd, "kernels": kernels}
if calib and calib[0] in kernels:
    k = kernels[calib[0]]
    doc["calibration"] = {"kernel": calib[0], "true_bytes_read": calib[1], "true_bytes_written": calib[1],
                          "fetch_counter_x1024": k.get("fetch_bytes"), "write_counter_x1024": k.get("write_bytes"),
                          "fetch_true_over_counter": (calib[1] / k["fetch_bytes"]) if k.get("fetch_bytes") else None,
                          "write_true_over_counter": (calib[1] / k["write_bytes"]) if k.get("write_bytes") else None}
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(kernels), "kernels")
