#!/usr/bin/env python3
"""BASELINE.json configs[2]: "Ruins scene at 1440p, ReSTIR diffuse + ray-traced specular + sun soft shadows, 1x MI355X" — the whole lighting
frame in world_render_passes.rs order on the Ruins stand-in (procedural_ruins, ~4M triangles): SSAO guide, sun shadow mask + denoiser,
irradiance cache + rtdgi, rtr, deferred combine, TAA. Per-segment HIP-event times from a serial run on one stream; the frame time from the same frames with the
cache's work of the next frame on a second stream (as bench.py pipelines the GI frame).
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
import time as _t
_t0 = _t.time()
def _stage(what): print(f"[config3_bench] {what}: {_t.time() - _t0:.1f} s since start", file=sys.stderr, flush=True)
dev = lib.Device(0)
desc = scenes.procedural_ruins(target_tris=a.tris, seed=5678)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
_stage("scene + pipeline built")
fs = frame.FrameState((W, H), sun_size_multiplier=4.0); fs.ircache_enabled = True
SEG = ["ssgi", "sun shadows + denoise", "ircache + rtdgi", "rtr", "light_gbuffer", "taa"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(SEG) + 1)]
acc = [0.0] * len(SEG)
rays = {"rtdgi": [0, 0], "ircache": [0, 0], "rtr": [0, 0]}
SERIAL = min(12, a.frames)      # frames of the serial segment (per-segment HIP-event times, ray counts); the frame time comes from the overlapped run
cam = lambda i: frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004)
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
for i in range(a.warmup + SERIAL):
    fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection()
    ev[0].record(); gp.ssgi_frame()
    ev[1].record(); shadow = gp.shadow_denoise(gp.sun_shadow_mask(out=mask))
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
_stage("warm-up + serial frames done")
n = SERIAL
seg = {k: round(v / n, 4) for k, v in zip(SEG, acc)}
serial_total = sum(seg.values())
# ---- the same frames with the irradiance cache's work of frame N+1 (maintenance + its three ray passes: lone waves, latency-bound) on a second
# stream, started when frame N's reflections are done (their lookups are the last users of the cache's state) and running under frame N's deferred
# combine + TAA and frame N+1's SSAO guide, shadow rays + denoiser: same dependencies as the serial order of world_render_passes.rs, same results
# (as bench.py's pipelined GI frame). Inputs of all frames are generated beforehand and stay resident; time = wall clock over the frames.
import ctypes as C, time
from kajiya_amd.abi import KJ_RTDGI_PASS as P
K0, K = a.warmup + SERIAL, a.frames
fcs, inputs = [], []
for i in range(K0, K0 + K + 7):
    fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
    gp.render_inputs(fc); gp.reprojection()
    rp = lib.tensor_from_ptr(gp.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
    fcs.append(fc); inputs.append((gp.geometric_normal.clone(), gp.gbuffer.clone(), gp.depth.clone(), rp))
torch.cuda.synchronize()
_stage("inputs of the overlapped frames generated")
main, side, tail = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev_fc, ev_irc, ev_rtr, ev_lit, ev_taa = ([torch.cuda.Event(), torch.cuda.Event()] for _ in range(5))
L = gp.L


def enqueue_cache(j, wait_event):
    with torch.cuda.stream(side):
        side.wait_stream(main) if wait_event is None else side.wait_event(wait_event)
        gp.dev.frame_begin(fcs[j]); ev_fc[j & 1].record(side)
        s = lib._stream_ptr()
        lib.check(L.kj_ircache_prepare(gp.ircache, s))
        lib.check(L.kj_ircache_trace_irradiance(gp.ircache, gp.scene.h, gp.sky16.data_ptr(), 16, s))
        ev_irc[j & 1].record(side)


def overlapped_frame(j):
    gp.geometric_normal, gp.gbuffer, gp.depth, rp = inputs[j]
    gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
    main.wait_event(ev_fc[j & 1])
    gp.ssgi_frame()
    shadow = gp.shadow_denoise(gp.sun_shadow_mask(out=mask))
    s = lib._stream_ptr()
    lib.check(L.kj_rtdgi_reproject(gp.rtdgi, gp.reprojection_map_ptr, W, H, s))
    p = gp.params(P["EXTRACT_HALF"]); lib.check(L.kj_rtdgi_render(gp.rtdgi, C.byref(p), C.byref(gp.out), s))
    main.wait_event(ev_irc[j & 1])
    lib.check(L.kj_ircache_sum_up_irradiance_for_sampling(gp.ircache, s))
    p = gp.params((P["ALL"] & ~P["EXTRACT_HALF"]) | (1 << 31)); lib.check(L.kj_rtdgi_render(gp.rtdgi, C.byref(p), C.byref(gp.out), s))
    rtr = gp.rtr_frame()
    ev_rtr[j & 1].record(main)
    if j > 0:
        main.wait_event(ev_taa[(j - 1) & 1])      # last frame's TAA has read the lit image this deferred combine overwrites
    lit_t, lit = gp.light_gbuffer(shadow, rtr_ptr=rtr.data_ptr())
    ev_lit[j & 1].record(main)
    with torch.cuda.stream(tail):                # TAA (VALU-bound) on a third stream: under the next frame's SSAO guide and shadow rays
        tail.wait_event(ev_lit[j & 1])
        gp.taa_frame(input_ptr=lit.data_ptr())
        ev_taa[j & 1].record(tail)
    enqueue_cache(j + 1, ev_rtr[j & 1])


enqueue_cache(0, None)
for j in range(6):
    overlapped_frame(j)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(6, 6 + K):
    overlapped_frame(j)
torch.cuda.synchronize()
total = 1e3 * (time.perf_counter() - t0) / K
_stage("overlapped frames done")
shadow_rays = W * H
all_rays = sum(v[0] + v[1] for v in rays.values()) / n + shadow_rays
print(json.dumps({"config": "BASELINE configs[2]", "workload": f"procedural_ruins {a.tris} tris (Ruins stand-in) @ {W}x{H}", "frame_ms": round(total, 4), "fps": round(1000.0 / total, 1),
                  "serial_frame_ms": round(serial_total, 4), "segment_ms": seg, "rays_per_frame": {k: [v[0] / n, v[1] / n] for k, v in rays.items()} | {"sun shadow": [0, shadow_rays]},
                  "mrays_per_s": round(all_rays / total / 1e3, 1), "frames": K, "serial_frames": n,
                  "overlap": "ircache of frame N+1 on a second stream from the end of frame N's reflections; TAA of frame N on a third stream under frame N+1's SSAO guide + shadows; segment_ms from a serial run of the same frames",
                  "rtr_tables": "stand-in"}))
