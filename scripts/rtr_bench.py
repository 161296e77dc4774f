#!/usr/bin/env python3
"""Times RtrRenderer::trace + filter_temporal (kj_rtr_*) inside the whole frame on the bench workload (1080p, ~1M-triangle procedural
city, orbiting camera): ms per frame of the six rtr passes (HIP events on the launch stream) and rays per frame.
usage: rtr_bench.py [--res WxH] [--tris N] [--frames K] [--warmup W]   (rocprofv3 --kernel-trace --stats around it gives the per-kernel split)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1080"); ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--frames", type=int, default=60); ap.add_argument("--warmup", type=int, default=30)
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
dev = lib.Device(0)
gp = lib.GpuPipeline(dev, lib.Scene(dev, scenes.procedural_city(target_tris=a.tris, seed=1234)), W, H, use_ircache=True)
fs = frame.FrameState((W, H)); fs.ircache_enabled = True
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rtr_ms = gi_ms = 0.0
rays = [0, 0]
for i in range(a.warmup + a.frames):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection(); gp.ssgi_frame()
    g0.record(); gp.gi_frame(); g1.record()
    e0.record(); gp.rtr_frame(); e1.record()
    torch.cuda.synchronize()
    if i >= a.warmup:
        rtr_ms += e0.elapsed_time(e1); gi_ms += g0.elapsed_time(g1)
        c, s = gp.rtr_ray_counts(); rays[0] += c; rays[1] += s
n = a.frames
print(json.dumps({"workload": f"procedural_city {a.tris} tris @ {W}x{H}", "rtr_ms_per_frame": rtr_ms / n, "gi_ms_per_frame_serial": gi_ms / n,
                  "rtr_closest_rays_per_frame": rays[0] / n, "rtr_any_rays_per_frame": rays[1] / n,
                  "rtr_mrays_per_s": (rays[0] + rays[1]) / n / (rtr_ms / n) / 1e3, "frames": n, "tables": "stand-in (kajiya_amd/rtr_tables.py)"}))
