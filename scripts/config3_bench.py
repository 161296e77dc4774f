#!/usr/bin/env python3
"""BASELINE.json configs[2]: "Ruins scene at 1440p, ReSTIR diffuse + ray-traced specular + sun soft shadows, 1x MI355X" — the whole lighting
frame in world_render_passes.rs order on the Ruins stand-in (procedural_ruins, ~4M triangles): SSAO guide, sun shadow mask + denoiser,
irradiance cache + rtdgi, rtr, deferred combine, TAA. Serial on one stream (no frame pipelining); HIP-event time per segment.
usage: config3_bench.py [--res WxH] [--tris N] [--frames K] [--warmup W]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="2560x1440"); ap.add_argument("--tris", type=int, default=4_000_000)
ap.add_argument("--frames", type=int, default=60); ap.add_argument("--warmup", type=int, default=24)
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
dev = lib.Device(0)
desc = scenes.procedural_ruins(target_tris=a.tris, seed=5678)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
fs = frame.FrameState((W, H), sun_size_multiplier=4.0); fs.ircache_enabled = True
SEG = ["ssgi", "sun shadows + denoise", "ircache + rtdgi", "rtr", "light_gbuffer", "taa"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(SEG) + 1)]
acc = [0.0] * len(SEG)
rays = {"rtdgi": [0, 0], "ircache": [0, 0], "rtr": [0, 0]}
for i in range(a.warmup + a.frames):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection()
    ev[0].record(); gp.ssgi_frame()
    ev[1].record(); shadow = gp.shadow_denoise(gp.sun_shadow_mask())
    ev[2].record(); gp.gi_frame()
    ev[3].record(); rtr = gp.rtr_frame()
    ev[4].record(); lit_t, lit = gp.light_gbuffer(shadow, rtr_ptr=rtr.data_ptr())
    ev[5].record(); gp.taa_frame(input_ptr=lit.data_ptr())
    ev[6].record(); torch.cuda.synchronize()
    if i >= a.warmup:
        for k in range(len(SEG)):
            acc[k] += ev[k].elapsed_time(ev[k + 1])
        for name, fn in (("rtdgi", gp.ray_counts), ("ircache", gp.ircache_ray_counts), ("rtr", gp.rtr_ray_counts)):
            c, s = fn(); rays[name][0] += c; rays[name][1] += s
n = a.frames
seg = {k: round(v / n, 4) for k, v in zip(SEG, acc)}
total = sum(seg.values())
shadow_rays = W * H
all_rays = sum(v[0] + v[1] for v in rays.values()) / n + shadow_rays
print(json.dumps({"config": "BASELINE configs[2]", "workload": f"procedural_ruins {a.tris} tris (Ruins stand-in) @ {W}x{H}", "frame_ms": round(total, 4), "fps": round(1000.0 / total, 1),
                  "segment_ms": seg, "rays_per_frame": {k: [v[0] / n, v[1] / n] for k, v in rays.items()} | {"sun shadow": [0, shadow_rays]},
                  "mrays_per_s": round(all_rays / total / 1e3, 1), "frames": n, "overlap": "none (one stream)", "rtr_tables": "stand-in"}))
