#!/usr/bin/env python3
"""Renders N frames of the whole path in world_render_passes.rs order — G-buffer stand-in, reprojection map, ssgi, sun shadow mask + denoise,
ircache, rtdgi, rtr, light_gbuffer, TAA on the lit image — and writes the last TAA output as a tone-mapped PNG (visual evidence).
usage: render_frame.py [city|cornell|glossy] [out.png]; also prints the per-frame time of the rtr passes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from PIL import Image
from kajiya_amd import lib, scenes, frame

scene_name = sys.argv[1] if len(sys.argv) > 1 else "city"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/frame.png"
W, H, N = 1280, 720, 48
dev = lib.Device(0)
if scene_name == "cornell":
    desc, cam = scenes.cornell_box(), dict(center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.002)
elif scene_name == "glossy":
    desc, cam = scenes.glossy_test_scene(), dict(center=(0.0, 1.0, 0.0), radius=9.0, height=3.0, rate=0.002)
else:
    desc, cam = scenes.procedural_city(target_tris=300_000, seed=1234), dict(center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.002)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
fs = frame.FrameState((W, H), sun_size_multiplier=4.0); fs.ircache_enabled = True
for i in range(N):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), **cam)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection()
    gp.ssgi_frame()
    shadow = gp.shadow_denoise(gp.sun_shadow_mask())      # world_render_passes.rs:123-136
    gp.gi_frame()
    if i == N - 8:
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); rtr_ms = 0.0
    if i >= N - 8:
        e0.record()
    rtr = gp.rtr_frame()                                  # world_render_passes.rs:172-210 (stand-in sampler tables, kajiya_amd/rtr_tables.py)
    if i >= N - 8:
        e1.record(); torch.cuda.synchronize(); rtr_ms += e0.elapsed_time(e1) / 8
    lit_t, lit = gp.light_gbuffer(shadow, rtr_ptr=rtr.data_ptr())
    gp.taa_frame(input_ptr=lit.data_ptr())
torch.cuda.synchronize()
img = gp.taa_surface("this_frame_output_img", torch.float16, (H, W, 4))[..., :3].float().cpu().numpy()
gi = gp.surface("spatial_filtered_tex", torch.float16, (H, W, 4))[..., :3].float().cpu().numpy()
ao = lib.tensor_from_ptr(gp.ssao_ptr.value, W * H, torch.uint8, (H, W)).cpu().numpy()
ru = rtr.cpu().numpy().view(np.uint32)
uf = lambda v, m: (v.astype(np.uint16) << (10 - m)).view(np.float16).astype(np.float32)
refl = np.stack([uf(ru & 0x7ff, 6), uf((ru >> 11) & 0x7ff, 6), uf(ru >> 22, 5)], -1)
tm = lambda a, e: (np.clip(1 - np.exp(-a * e), 0, 1) ** (1 / 2.2) * 255).astype(np.uint8)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
panel = np.concatenate([tm(img, 0.35), tm(gi, 1.2), tm(refl, 0.35), np.repeat(ao[..., None], 3, axis=2)], axis=1)
Image.fromarray(panel).resize((panel.shape[1] // 2, panel.shape[0] // 2), Image.BILINEAR).save(out)
print("wrote", out, "lit mean", img.mean(), "gi mean", gi.mean(), "rtr mean", refl.mean(), "rtr ms/frame (%dx%d) %.3f" % (W, H, rtr_ms), "rtr rays", gp.rtr_ray_counts())
