#!/usr/bin/env python3
"""Static instruction mix per kernel from hipcc's -save-temps assembly (a proxy for the VALU-bound screen-space passes):
  python scripts/isa_stats.py <name>-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-substring]"""
import re, sys
path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cur, stats = None, {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = dict(valu=0, salu=0, vmem=0, lds=0, trans=0, branch=0, total=0); continue
    if cur is None: continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";", "//")): continue
    op = t[0]
    if op == "s_endpgm": cur = None; continue
    s = stats[cur]; s["total"] += 1
    if op.startswith("v_"):
        s["valu"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): s["trans"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): s["vmem"] += 1
    elif op.startswith("ds_"): s["lds"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): s["branch"] += 1
    elif op.startswith("s_"): s["salu"] += 1
for k, v in stats.items():
    if filt in k and v["total"] > 20:
        print(f"{k[:90]:<90s} " + " ".join(f"{a}={b}" for a, b in v.items()))
