#!/usr/bin/env python3
"""The whole frame of world_render_passes.rs incl. its tail: ... light_gbuffer -> TAA -> motion_blur -> PostProcessRenderer, with the
exposure loop of world_renderer.rs:919-960 (dynamic exposure on). Writes the library's own display-referred output as an sRGB PNG and
prints the per-frame time of the tail. NOT YET RUN ON HARDWARE (written after round 1's GPU budget was spent): first thing to run next. It has run end to end against the product
source on the CPU stand-in for HIP (tests/hip_emu; profiles/r01_frame_cornell_post_480x270_cpu_stand_in.png).
usage: render_frame_post.py [city|cornell|glossy] [out.png] [W H]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from PIL import Image
from kajiya_amd import lib, scenes, frame, post_tables, exposure as E

scene_name = sys.argv[1] if len(sys.argv) > 1 else "city"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/frame_post.png"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
N = 64
dev = lib.Device(0)
if scene_name == "cornell":
    desc, cam = scenes.cornell_box(), dict(center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.004)
elif scene_name == "glossy":
    desc, cam = scenes.glossy_test_scene(), dict(center=(0.0, 1.0, 0.0), radius=9.0, height=3.0, rate=0.004)
else:
    desc, cam = scenes.procedural_city(target_tris=300_000, seed=1234), dict(center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
post = lib.GpuPost(dev, post_tables.zero_bezold_brucke_lut())       # a kajiya host passes its BezoldBruckeLutComputer image here
mblur = lib.GpuMotionBlur(dev)
ex = E.Exposure(dynamic_exposure=E.DynamicExposureState(enabled=True, speed_log2=2.5, histogram_clipping=E.HistogramClipping(0.1, 0.1)))
fs = frame.FrameState((W, H), sun_size_multiplier=4.0); fs.ircache_enabled = True
image_log2_lum, tail_ms = 0.0, 0.0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(N):
    ex.update_pre_exposure(image_log2_lum); ex.apply(fs)
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), **cam)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection()
    gp.ssgi_frame()
    shadow = gp.shadow_denoise(gp.sun_shadow_mask())
    gp.gi_frame()
    rtr = gp.rtr_frame()
    lit_t, lit = gp.light_gbuffer(shadow, rtr_ptr=rtr.data_ptr())
    gp.taa_frame(input_ptr=lit.data_ptr())
    taa_out = gp.taa_surface("this_frame_output_img", torch.float16, (H, W, 4))
    reproj = lib.tensor_from_ptr(gp.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4))
    e0.record()
    blurred = mblur.render(taa_out, gp.depth, reproj)
    ldr = post.render(blurred, float(ex.state.post_mult), ex.contrast)
    e1.record(); torch.cuda.synchronize()
    if i >= N - 16:
        tail_ms += e0.elapsed_time(e1) / 16
    image_log2_lum, _ = post.read_back_histogram(ex.dynamic_exposure.histogram_clipping.low, ex.dynamic_exposure.histogram_clipping.high)
u = ldr.cpu().numpy().view(np.uint32)
uf = lambda v, m: (v.astype(np.uint16) << (10 - m)).view(np.float16).astype(np.float32)
lin = np.clip(np.stack([uf(u & 0x7ff, 6), uf((u >> 11) & 0x7ff, 6), uf(u >> 22, 5)], -1), 0, 1)
srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)       # the swap chain's transfer function
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
Image.fromarray((srgb * 255 + 0.5).astype(np.uint8)).save(out)
print("wrote", out, f"motion blur + post {tail_ms:.3f} ms/frame at {W}x{H}; image_log2_lum {image_log2_lum:.2f}, pre_mult {float(ex.state.pre_mult):.4f}, post_mult {float(ex.state.post_mult):.4f}")
