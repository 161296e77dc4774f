#!/usr/bin/env python3
"""The three BLAS builders side by side (include/kajiya_amd.h: KJ_BLAS_BUILD_*): host binned SAH, device LBVH, device PLOC.
Per builder: first-commit time (cold and with the allocator / code objects warm), traversal rates on the SAME incoherent hemisphere rays
(one ray per lane and the ray stream; closest / any hit), nodes and triangles visited per ray and the time of the rtdgi trace pass at
1080p. One JSON line per builder. usage: blas_builders_bench.py [--scene city|ruins] [--tris N] [--rays N] [--width W --height H]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from kajiya_amd import frame, lib, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="city")
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--rays", type=int, default=1 << 21)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=8)
args = ap.parse_args()
desc = scenes.procedural_city(target_tris=args.tris, seed=1234) if args.scene == "city" else scenes.procedural_ruins(target_tris=args.tris, seed=5678)
dev = lib.Device(0)
lo, hi = desc.bounds()
r1 = None
KNOBS = ("KJ_TRACE_QUAD_MAX_RAYS", "KJ_TRACE_PER_RAY")


def rate(scene, rays, n, reps=10):
    out = {}
    for name, fn in (("closest", lambda: scene.trace_closest(rays, n)), ("any", lambda: scene.trace_any(rays, n))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = max(e0.elapsed_time(e1) / reps, 1e-6)
        out[name] = round(n / ms / 1e3, 1)
    return out


for name, fb in (("host_sah", False), ("device_lbvh", True), ("device_ploc", "ploc")):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scene = lib.Scene(dev, desc, fast_build=fb)
    torch.cuda.synchronize(); cold = 1e3 * (time.perf_counter() - t0); cold_ph = scene.last_commit_ms()
    t0 = time.perf_counter()
    scene = lib.Scene(dev, desc, fast_build=fb)
    torch.cuda.synchronize(); warm = 1e3 * (time.perf_counter() - t0); warm_ph = scene.last_commit_ms()
    if r1 is None:      # rays leave scene surfaces in uniform upper-hemisphere directions (what the rtdgi trace pass issues); same set for all builders
        rng = np.random.RandomState(1)
        N = args.rays
        o = rng.uniform(lo, hi, size=(N, 3)); o[:, 1] = hi[1] + 5.0
        d = rng.normal(size=(N, 3)); d[:, 1] = -np.abs(d[:, 1]) - 0.5; d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.zeros((N, 8), np.float32); rays[:, :3] = o; rays[:, 4:7] = d; rays[:, 7] = 1e4
        r0 = torch.from_numpy(rays).cuda()
        hits = scene.trace_closest(r0, N)
        t = hits[:, 0]; ok = t < 1e30
        p = r0[:, :3] + r0[:, 4:7] * t[:, None]
        d2 = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32)).cuda(); d2 = d2 / d2.norm(dim=1, keepdim=True); d2[:, 1] = d2[:, 1].abs()
        r1 = torch.zeros((N, 8), device="cuda"); r1[:, :3] = p + 1e-3 * d2; r1[:, 4:7] = d2; r1[:, 7] = 1e4
        r1 = r1[ok].contiguous()
    M = r1.shape[0]
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(KJ_TRACE_PER_RAY="1", KJ_TRACE_QUAD_MAX_RAYS="0")
    per_lane = rate(scene, r1, M)
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(KJ_TRACE_QUAD_MAX_RAYS="0")
    stream = rate(scene, r1, M)
    for k in KNOBS: os.environ.pop(k, None)
    W, H = args.width, args.height
    gp = lib.GpuPipeline(dev, scene, W, H)
    gp.set_profiling(True, True)
    fs = frame.FrameState((W, H))
    for i in range(3):
        gp.frame(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004))); fs.retire_frame()
    torch.cuda.synchronize()
    c = gp.traversal_counts()
    gp.set_profiling(True, False)
    tr = []
    for i in range(3, 3 + args.frames):
        gp.frame(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004))); fs.retire_frame()
        torch.cuda.synchronize()
        tr.append(gp.pass_times_ms()[3])
    print(json.dumps({"builder": name, "scene": f"{args.scene} {scene.stats()['triangles']} tris", "first_commit_ms": {"cold": round(cold, 2), "warm": round(warm, 2), "warm_phases": [round(x, 2) for x in warm_ph]},
                      "rays": M, "mrays_per_s_one_ray_per_lane": per_lane, "mrays_per_s_stream": stream,
                      "nodes_per_closest_ray": round(c["closest_nodes"] / max(1, c["closest_rays"]), 2), "tris_per_closest_ray": round(c["closest_tris"] / max(1, c["closest_rays"]), 2),
                      "nodes_per_any_ray": round(c["any_nodes"] / max(1, c["any_rays"]), 2), "tris_per_any_ray": round(c["any_tris"] / max(1, c["any_rays"]), 2),
                      "rtdgi_trace_pass_ms": {"median": round(float(np.median(tr)), 4), "min": round(min(tr), 4), "max": round(max(tr), 4)}}), flush=True)
    del gp, scene
