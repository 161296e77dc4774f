"""Host cost of the per-commit top tree by instance count: `python tests/hip_emu/run_with_emu.py scripts/top_tree_host_cost.py` (no GPU needed -- the numbers
that matter here are HOST milliseconds: stage [1] of kj_scene_last_commit_ms = instance records + top tree; kernels run on the CPU stand-in and their times
mean nothing) or, on a GPU box, `python scripts/top_tree_host_cost.py` for the device build's real cost in stage [2]."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kajiya_amd import lib, scenes

rng = np.random.default_rng(1)
mesh = scenes.TriangleMesh(rng.uniform(-1, 1, (12, 3)).astype(np.float32), np.tile(np.array([[0, 0, 1]], np.float32), (12, 1)), np.arange(12, dtype=np.uint32))
dev = lib.Device(0)
rows = []
for n in (64, 256, 1024, 4096, 8192, 32768):
    for mode in ("host", "device"):
        desc = scenes.SceneDesc()
        desc.add_mesh(mesh)
        xf = []
        for i in range(n):
            xf.append(scenes.affine(np.eye(3), 1.0, rng.uniform(-300, 300, 3)))
            desc.add_instance(0, xf[-1])
        sc = lib.Scene(dev, desc, top_build=mode)
        first = sc.last_commit_ms()
        moves = []
        for k in range(3):      # a per-frame commit: one instance moved
            sc.set_instance_transform(k, xf[k]); sc.commit()
            moves.append(sc.last_commit_ms())
        best = min(moves, key=lambda m: m[3])
        rows.append({"instances": n, "top_build": mode, "first_commit_ms": [round(v, 3) for v in first], "move_one_commit_ms": [round(v, 3) for v in best], "top_tree": sc.top_tree_info()})
        print(json.dumps(rows[-1]), flush=True)
