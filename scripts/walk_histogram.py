#!/usr/bin/env python3
"""How full are the waves of the rtdgi ray passes while they walk? ("first record the histogram of live lanes per wave step")
The instrumented (STATS) instantiations of the fused ray kernels count, per wave step of the BVH walk, how many lanes still have a ray: closest-hit walks in four
bins (1-8, 9-16, 17-32, 33-64 lanes), occlusion walks in two (<= 16, > 16), next to the node / triangle counts bench.py already reports. One JSON line per workload.
usage: walk_histogram.py [--res WxH] [--scene city|ruins] [--tris N] [--frames K]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1080"); ap.add_argument("--scene", default="city"); ap.add_argument("--tris", type=int, default=1_000_000); ap.add_argument("--frames", type=int, default=6)
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
dev = lib.Device(0)
if a.scene == "ruins":
    desc, cam = scenes.procedural_ruins(target_tris=a.tris, seed=5678), dict(center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004)
else:
    desc, cam = scenes.procedural_city(target_tris=a.tris, seed=1234), dict(center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
fs = frame.FrameState((W, H)); fs.ircache_enabled = True
ptr, n = C.c_void_p(), C.c_uint64()
tot = torch.zeros(16, dtype=torch.int64, device="cuda")
for i in range(6 + a.frames):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), **cam)); fs.retire_frame()
    if i == 6:
        gp.set_profiling(False, True)      # the STATS instantiations from here on
    gp.frame(fc)
    if i >= 6:
        lib.check(gp.L.kj_rtdgi_surface(gp.rtdgi, b"ray_counters", C.byref(ptr), C.byref(n)))
        tot += lib.tensor_from_ptr(ptr.value, n.value, torch.int64, (64, 16)).sum(dim=0)
torch.cuda.synchronize()
t = [int(v) for v in tot.tolist()]
steps_c, steps_a = t[6] + t[7], t[8] + t[9]
hist = t[10:14]
out = {"workload": f"procedural_{a.scene} ~{a.tris} tris @ {W}x{H}, {a.frames} frames (validation + tracing), both fused ray kernels",
       "closest_rays": t[0], "any_rays": t[1],
       "closest": {"wave_steps": steps_c, "lane_steps": t[2] + t[3], "lane_utilisation": round((t[2] + t[3]) / max(1, 64 * steps_c), 4),
                   "wave_steps_by_live_lanes": {"1-8": hist[0], "9-16": hist[1], "17-32": hist[2], "33-64": hist[3]},
                   "fraction_of_wave_steps": {k: round(v / max(1, sum(hist)), 4) for k, v in zip(("1-8", "9-16", "17-32", "33-64"), hist)}},
       "any": {"wave_steps": steps_a, "lane_steps": t[4] + t[5], "lane_utilisation": round((t[4] + t[5]) / max(1, 64 * steps_a), 4),
               "wave_steps_by_live_lanes": {"<=16": t[14], ">16": t[15]}, "fraction_at_most_16": round(t[14] / max(1, t[14] + t[15]), 4)}}
print(json.dumps(out))
