#!/usr/bin/env python3
"""Cost of scene edits (WorldRenderer::set_instance_transform + the per-frame TLAS build, world_renderer.rs:815,836-911) on the bench scene:
first commit (every mesh's BLAS + all instances), then commits after moving ONE instance, after moving ALL instances, and after
removing one. Prints one JSON line. usage: dynamic_scene_bench.py [tris]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from kajiya_amd import lib, scenes

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = lib.Device(0)
desc = scenes.procedural_city(target_tris=tris, seed=1234)
t0 = time.perf_counter()
scene = lib.Scene(dev, desc)
torch.cuda.synchronize()
first = dict(wall_ms=1e3 * (time.perf_counter() - t0), phases_ms=scene.last_commit_ms(), **scene.stats())


def timed_commit():
    torch.cuda.synchronize()
    t = time.perf_counter()
    scene.commit()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t), scene.last_commit_ms()


rng = np.random.RandomState(3)
one, allm = [], []
n_inst = len(desc.instances)
for it in range(20):
    i = int(rng.randint(n_inst))
    xf = np.array(desc.instances[i][1], np.float32).reshape(3, 4).copy()
    xf[:, 3] += rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    scene.set_instance_transform(i, xf)
    one.append(timed_commit())
for it in range(5):
    for i in range(n_inst):
        xf = np.array(desc.instances[i][1], np.float32).reshape(3, 4).copy()
        xf[:, 3] += rng.uniform(-0.5, 0.5, 3).astype(np.float32)
        scene.set_instance_transform(i, xf)
    allm.append(timed_commit())
lib.check(lib.load().kj_scene_remove_instance(scene.h, 5))
rem = timed_commit()
# the same scene with every BLAS built on the device (KJ_BLAS_BUILD_FAST_BUILD): first commit = full build of all meshes
torch.cuda.synchronize()
t0 = time.perf_counter()
scene_fast = lib.Scene(dev, desc, fast_build=True)
torch.cuda.synchronize()
fast_first = dict(wall_ms=1e3 * (time.perf_counter() - t0), phases_ms=scene_fast.last_commit_ms(), **scene_fast.stats())
t0 = time.perf_counter()
scene_fast2 = lib.Scene(dev, desc, fast_build=True)     # second time: allocator and code objects warm
torch.cuda.synchronize()
fast_second = dict(wall_ms=1e3 * (time.perf_counter() - t0), phases_ms=scene_fast2.last_commit_ms())
med = lambda xs: float(np.median(xs))
print(json.dumps({"scene": f"procedural_city {tris} tris, {n_inst} instances of {len(desc.meshes)} meshes", "first_commit": first,
                  "move_one_instance_commit_ms": {"wall_median": med([w for w, _ in one]), "wall_min": min(w for w, _ in one), "phases_median": [med([p[k] for _, p in one]) for k in range(4)]},
                  "move_all_instances_commit_ms": {"wall_median": med([w for w, _ in allm]), "phases_median": [med([p[k] for _, p in allm]) for k in range(4)]},
                  "remove_one_instance_commit_ms": {"wall": rem[0], "phases": rem[1]},
                  "device_lbvh_first_commit": fast_first, "device_lbvh_first_commit_warm": fast_second}))
