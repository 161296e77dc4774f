#!/usr/bin/env python3
"""Closest-hit / occlusion ray rates under a host-built (binned SAH) and a device-built (linear BVH) TOP tree, by instance count: what the cheaper per-commit build
(scripts/top_tree_host_cost.py) costs the frame's rays. Instances of one 5 k-triangle mesh scattered over a slab; the same incoherent rays for both trees.
One JSON line per (instances, builder). usage: python scripts/top_tree_trace_rate.py [--rays N]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from kajiya_amd import lib, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=1 << 20)
args = ap.parse_args()
rng = np.random.default_rng(7)
# a bumpy blob: 50 x 50 grid of quads over a sphere-ish height field
g = 51
u, v = np.meshgrid(np.linspace(0, np.pi, g), np.linspace(0, 2 * np.pi, g), indexing="ij")
r = 1.0 + 0.15 * np.sin(5 * u) * np.cos(7 * v)
P = np.stack([r * np.sin(u) * np.cos(v), r * np.cos(u), r * np.sin(u) * np.sin(v)], -1).reshape(-1, 3).astype(np.float32)
idx = []
for i in range(g - 1):
    for j in range(g - 1):
        a = i * g + j
        idx += [a, a + 1, a + g, a + 1, a + g + 1, a + g]
mesh = scenes.TriangleMesh(P, P / np.linalg.norm(P, axis=1, keepdims=True), np.array(idx, np.uint32))
dev = lib.Device(0)


def rate(fn, n, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(n / max(e0.elapsed_time(e1) / reps, 1e-6) / 1e3, 1)


for n in (256, 1024, 4096, 16384):
    side = 3.0 * n ** 0.5
    xf = [scenes.affine(np.eye(3), float(rng.uniform(0.6, 1.4)), np.array([rng.uniform(-side, side), rng.uniform(-2, 2), rng.uniform(-side, side)])) for _ in range(n)]
    N = args.rays
    o = np.stack([rng.uniform(-side, side, N), rng.uniform(-3, 3, N), rng.uniform(-side, side, N)], -1)
    d = rng.normal(size=(N, 3)); d[:, 1] *= 0.3; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((N, 8), np.float32); rays[:, :3] = o; rays[:, 4:7] = d; rays[:, 7] = 1e4
    r0 = torch.from_numpy(rays).cuda()
    ref = None
    for mode in ("host", "device"):
        desc = scenes.SceneDesc()
        desc.add_mesh(mesh)
        for x in xf: desc.add_instance(0, x)
        sc = lib.Scene(dev, desc, top_build=mode)
        hits = sc.trace_closest(r0, N)
        if ref is None: ref = hits.clone()
        same = bool(torch.equal(torch.nan_to_num(ref), torch.nan_to_num(hits)))
        out = {"instances": n, "top_build": mode, "triangles": n * (len(idx) // 3), "closest_mrays_per_s": rate(lambda: sc.trace_closest(r0, N), N), "any_mrays_per_s": rate(lambda: sc.trace_any(r0, N), N),
               "hit_fraction": round(float((hits[:, 0] < 1e30).float().mean()), 3), "same_hits_as_host_tree": same, "top_tree": sc.top_tree_info(), "commit_ms": [round(v, 3) for v in sc.last_commit_ms()]}
        print(json.dumps(out), flush=True)
        del sc
