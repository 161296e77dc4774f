#!/usr/bin/env python3
"""Time-averaged rtdgi irradiance vs the reference path tracer (first_bounce_mode 2) on the Cornell box (GPU).
Prints relative L2 / mean ratio for a few configurations; used to set the tolerance of tests/test_gpu_reference_pt.py."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from kajiya_amd import lib, scenes, frame

dev = lib.Device(0)
desc = scenes.cornell_box()
scene = lib.Scene(dev, desc)
CONFIGS = ((512, True, {}, "sun"), (512, False, {}, "sun"), (512, True, dict(sun_color_multiplier=(0, 0, 0), sky_ambient=(1, 1, 1)), "white sky"),
                          (256, True, {}, "sun"), (128, True, {}, "sun"))
for W, irc, kw, label in (CONFIGS[:1] if os.environ.get("KJ_PROBE_DUMP") else CONFIGS):
    H = W
    gp = lib.GpuPipeline(dev, scene, W, H, use_ircache=irc)
    fs = frame.FrameState((W, H), **kw)
    fs.ircache_enabled = irc
    cam = lambda: frame.orbit_camera(0, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01)
    n_warm, n_avg, n_pt = 64, 128, 512
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    gi = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for i in range(max(n_pt, n_warm + n_avg)):
        fc = fs.prepare_frame_constants(cam()); fs.retire_frame()
        if i < n_pt:
            dev.frame_begin(fc)
            gp.reference_path_trace(acc, first_bounce_mode=2)
        if i < n_warm + n_avg:
            gp.frame(fc)
            if i >= n_warm:
                gi += gp.surface("spatial_filtered_tex", torch.float16, (H, W, 4))[..., :3].float()
    torch.cuda.synchronize()
    g = (gi / n_avg).cpu().numpy(); p = acc[..., :3].cpu().numpy(); m = gp.depth.cpu().numpy() > 0
    rel = float(np.sqrt(((g - p)[m] ** 2).sum() / (p[m] ** 2).sum()))
    # blur both 8x8 to take the PT's per-pixel noise and the denoiser's footprint out of the comparison
    def blur(a):
        a = np.where(m[..., None], a, 0.0)
        return a.reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
    gb, pb = blur(g), blur(p)
    relb = float(np.sqrt(((gb - pb) ** 2).sum() / (pb ** 2).sum()))
    if os.environ.get("KJ_PROBE_DUMP") and W == 512 and irc and label == "sun":
        from PIL import Image
        os.makedirs("gpurun_out", exist_ok=True)
        tm = lambda a: (np.clip(a / (1 + a), 0, 1) ** (1 / 2.2) * 255).astype(np.uint8)
        ratio = np.where(m, g.sum(-1) / np.maximum(1e-4, p.sum(-1)), 1.0)
        rimg = np.stack([np.clip(ratio - 1, 0, 1), np.clip(1 - np.abs(ratio - 1), 0, 1) * 0.6, np.clip(1 - ratio, 0, 1)], -1)
        Image.fromarray(np.concatenate([tm(p * 3), tm(g * 3), (rimg * 255).astype(np.uint8)], axis=1)).save("gpurun_out/conv_pt_gi_ratio.png")
    print(f"{W}x{H} ircache={irc} {label}: rel L2 {rel:.4f}  (8x8 box-averaged {relb:.4f})  mean ratio {g[m].mean() / p[m].mean():.4f}  per-channel {g[m].mean(axis=0) / p[m].mean(axis=0)}")
