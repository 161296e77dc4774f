#!/usr/bin/env python3
"""Where does k_rtdgi_trace spend its time? Same 1080p city frames with (a) everything, (b) no irradiance cache bound,
(c) sun switched off (no shadow rays, no sun BRDF evaluation)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame

W, H = 1920, 1080
dev = lib.Device(0)
scene = lib.Scene(dev, scenes.procedural_city(target_tris=1_000_000, seed=1234))
for label, irc, kw in (("full", True, {}), ("no ircache", False, {}), ("no sun", True, dict(sun_color_multiplier=(0, 0, 0), sky_ambient=(1, 1, 1)))):
    gp = lib.GpuPipeline(dev, scene, W, H, use_ircache=irc)
    fs = frame.FrameState((W, H), **kw); fs.ircache_enabled = irc
    gp.set_profiling(True, False)
    acc = [0.0] * 11; n = 0
    for i in range(30):
        fc = fs.prepare_frame_constants(frame.orbit_camera(24 + i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)); fs.retire_frame()
        gp.frame(fc)
        torch.cuda.synchronize()
        if i >= 12:
            t = gp.pass_times_ms(); acc = [a + b for a, b in zip(acc, t)]; n += 1
    c, a = gp.ray_counts()
    print(f"{label:12s} trace {acc[3]/n:.4f} ms  validate {acc[2]/n:.4f} ms   last-frame rays closest {c} any {a}")
