#!/usr/bin/env python3
"""Bytes every rank RECEIVES per frame of the screen-tile split, per exchange point -- read off the orchestrator's own transfer plans (SplitRtdgi.exchange_log:
rows x row bytes of every block bound for the rank), not restated by hand. The plans depend on the extent, the rank count, the motion halo and the field of
view only, so the frames behind them may be rendered on anything: this runs the Python orchestrator with virtual ranks, two frames (the second has every
history exchange), on a GPU or -- `python tests/hip_emu/run_with_emu.py scripts/split_exchange_bytes.py ...` -- on the CPU stand-in. NOT a measurement of time.
usage: split_exchange_bytes.py [--res WxH] [--ranks N] [--motion-halo M] [--frame gi|config3]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame, multigpu

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1080"); ap.add_argument("--ranks", type=int, default=8); ap.add_argument("--motion-halo", type=int, default=16)
ap.add_argument("--frame", default="config3", choices=["gi", "config3"])
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
dev = lib.Device(0)
scene = lib.Scene(dev, scenes.glossy_test_scene())
pipes = {r: lib.GpuPipeline(dev, scene, W, H, use_ircache=False) for r in range(a.ranks)}      # (the cache's records are data-dependent and small: ~10 MB per 1080p frame in total)
split = multigpu.SplitRtdgi(multigpu.LocalComm(a.ranks), pipes, W, H, motion_halo=a.motion_halo)
if a.frame == "config3":
    split.enable_rtr()
fs = frame.FrameState((W, H), sun_size_multiplier=4.0)
log = None
for i in range(2):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.001)); fs.retire_frame()
    for r in pipes:
        pipes[r].render_inputs(fc); pipes[r].reprojection()
    split.exchange_log = log = []
    if a.frame == "config3":
        split.lighting_frame()
    else:
        split.ssgi_frame(); split.gi_frame(); split.taa_frame()
    torch.cuda.synchronize()
points = []
for key, into in log:
    names = sorted({(it[0].split(":")[0], "all rows" if it[1] is None else f"{it[1]} rows" + (" + row 0" if len(it) == 3 and it[2] else "")) for it in key})
    points.append({"surfaces": [f"{n} ({h})" for n, h in names], "max_bytes_into_a_rank": max(into.values()) if into else 0, "mean_bytes_into_a_rank": sum(into.values()) / a.ranks if into else 0})
total_max = max(sum(into.get(r, 0) for _, into in log) for r in range(a.ranks))
print(json.dumps({"what": f"bytes received per rank per frame, {a.frame} frame under the split (from the orchestrator's transfer plans; NOT a time measurement)", "extent": [W, H], "ranks": a.ranks,
                  "motion_halo": a.motion_halo, "vfov_deg": 52.0, "rtr_resolve_halo_half_rows": multigpu.rtr_resolve_halo(H, pipes[0].dev.clip_to_view_11) if a.frame == "config3" else None,
                  "total_MB_into_the_busiest_rank": round(total_max / 1e6, 2), "exchange_points": points}, indent=1))
