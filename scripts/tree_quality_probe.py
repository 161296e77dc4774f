"""Nodes and triangles visited per ray by the rtdgi trace pass for the three BLAS builders (host SAH / device LBVH / device PLOC), counted
by the instrumented trace kernel (kj_rtdgi_set_profiling(count_traversal=1)). Hardware-independent: a tree-quality figure. Runs on the GPU,
or anywhere through `python tests/hip_emu/run_with_emu.py scripts/tree_quality_probe.py --tris 200000 --width 320 --height 192`."""
import argparse
import json
import time

import torch

from kajiya_amd import frame, lib, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="city")
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=3)
args = ap.parse_args()
desc = scenes.procedural_city(args.tris) if args.scene == "city" else scenes.procedural_ruins(args.tris)
dev = lib.Device(0)
out = {}
for name, fb in (("host_sah", False), ("device_lbvh", True), ("device_ploc", "ploc")):
    t0 = time.time()
    sc = lib.Scene(dev, desc, fast_build=fb)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    gp = lib.GpuPipeline(dev, sc, args.width, args.height)
    gp.set_profiling(True, True)
    fs = frame.FrameState((args.width, args.height))
    lo, hi = desc.bounds()
    for i in range(args.frames):
        fc = fs.prepare_frame_constants(frame.orbit_camera(i, (args.width, args.height), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004))
        gp.frame(fc)
        fs.retire_frame()
    torch.cuda.synchronize()
    c = gp.traversal_counts()
    st = sc.stats()
    out[name] = {"scene_commit_s": round(build_s, 3), "commit_ms": [round(x, 2) for x in sc.last_commit_ms()], "bvh_nodes": st.get("nodes"),
                 "nodes_per_closest_ray": round(c["closest_nodes"] / max(1, c["closest_rays"]), 2), "tris_per_closest_ray": round(c["closest_tris"] / max(1, c["closest_rays"]), 2),
                 "nodes_per_any_ray": round(c["any_nodes"] / max(1, c["any_rays"]), 2), "tris_per_any_ray": round(c["any_tris"] / max(1, c["any_rays"]), 2)}
    print(name, json.dumps(out[name]), flush=True)
