#!/usr/bin/env python3
"""Pure traversal throughput (kj_trace_closest / kj_trace_any) on incoherent rays in the bench scene: rays start on scene
surfaces (found by a first trace from random points) and leave in cosine-free uniform hemisphere directions -- the same
population the rtdgi trace kernel issues."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from kajiya_amd import lib, scenes

dev = lib.Device(0)
desc = scenes.procedural_city(target_tris=1_000_000, seed=1234)
TOP = ([a.split("=")[1] for a in sys.argv if a.startswith("--top-build=")] or [None])[0]      # --top-build=device|host: force the per-commit top tree's builder
INST = int(([a.split("=")[1] for a in sys.argv if a.startswith("--instances=")] or [0])[0])       # --instances=N: the city with N instances instead of 64
if INST:
    desc = scenes.procedural_city(target_tris=1_000_000, seed=1234, n_instances=INST)
scene = lib.Scene(dev, desc, fast_build="--fast-build" in sys.argv, **({"top_build": TOP} if TOP else {}))   # --fast-build: BLASes as device-built LBVHs
print("top tree:", scene.top_tree_info(), flush=True)
lo, hi = desc.bounds()
rng = np.random.RandomState(1)
N = 1 << 21
o = rng.uniform(lo, hi, size=(N, 3)); o[:, 1] = hi[1] + 5.0
d = rng.normal(size=(N, 3)); d[:, 1] = -np.abs(d[:, 1]) - 0.5; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((N, 8), np.float32); rays[:, :3] = o; rays[:, 4:7] = d; rays[:, 7] = 1e4
r0 = torch.from_numpy(rays).cuda()
hits = scene.trace_closest(r0, N)
t = hits[:, 0]
ok = t < 1e30
p = r0[:, :3] + r0[:, 4:7] * t[:, None]
d2 = torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32)).cuda(); d2 = d2 / d2.norm(dim=1, keepdim=True)
d2[:, 1] = d2[:, 1].abs()          # roofs / ground mostly face up: upper hemisphere
r1 = torch.zeros((N, 8), device="cuda"); r1[:, :3] = p + 1e-3 * d2; r1[:, 4:7] = d2; r1[:, 7] = 1e4
r1 = r1[ok].contiguous(); M = r1.shape[0]
def measure(label):
    out = []
    for name, fn in (("closest", lambda: scene.trace_closest(r1, M)), ("any", lambda: scene.trace_any(r1, M))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.append(f"{name}: {ms:.3f} ms = {M / ms / 1e3:.0f} Mrays/s")
    print(f"{label:<44s} {M} rays  " + "   ".join(out), flush=True)


# the C-ABI reads these switches at every call (device.hip: stream_waves / stream_tune / KJ_TRACE_PER_RAY)
KNOBS = ("KJ_TRACE_QUAD_MAX_RAYS", "KJ_TRACE_PER_RAY", "KJ_STREAM_WAVES_PER_CU", "KJ_STREAM_REFILL", "KJ_STREAM_NODE_WEIGHT", "KJ_STREAM_TRI_WEIGHT")
os.environ["KJ_DEBUG_ENV"] = "1"      # the library reads measurement switches only behind this gate (kj_host.hpp: kj_debug_getenv)
os.environ["KJ_TRACE_QUAD_MAX_RAYS"] = "0"
configs = [dict(KJ_TRACE_PER_RAY="1", KJ_TRACE_QUAD_MAX_RAYS="0"), dict(KJ_TRACE_QUAD_MAX_RAYS="0")]
if "--sweep" in sys.argv:
    configs += [dict(KJ_STREAM_WAVES_PER_CU=str(w)) for w in (8, 16, 32)]
    configs += [dict(KJ_STREAM_REFILL=str(t)) for t in (1, 8, 32, 48)]
    configs += [dict(KJ_STREAM_NODE_WEIGHT="2", KJ_STREAM_TRI_WEIGHT="1"), dict(KJ_STREAM_NODE_WEIGHT="1", KJ_STREAM_TRI_WEIGHT="2"),
                dict(KJ_STREAM_NODE_WEIGHT="1", KJ_STREAM_TRI_WEIGHT="0"), dict(KJ_STREAM_NODE_WEIGHT="1", KJ_STREAM_TRI_WEIGHT="3")]
for cfg in configs:
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(cfg)
    measure("one ray per lane (bvh_trace)" if cfg.get("KJ_TRACE_PER_RAY") else "stream " + (" ".join(f"{k[10:].lower()}={v}" for k, v in cfg.items()) or "(defaults)"))


# ---- small batches (the irradiance cache's ray passes issue ~28 k paths per launch): one ray per lane vs ray stream vs four lanes per ray
def measure_small(label, n):
    sub = r1[:n].contiguous()
    out = []
    for name, fn in (("closest", lambda: scene.trace_closest(sub, n)), ("any", lambda: scene.trace_any(sub, n))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out.append(f"{name}: {ms * 1e3:.1f} us = {n / ms / 1e3:.0f} Mrays/s")
    print(f"{label:<44s} {n} rays  " + "   ".join(out), flush=True)


for n in (8192, 28672, 65536):
    for label, cfg in (("one ray per lane", dict(KJ_TRACE_PER_RAY="1", KJ_TRACE_QUAD_MAX_RAYS="0")), ("stream", dict(KJ_TRACE_QUAD_MAX_RAYS="0")), ("four lanes per ray (quad)", dict(KJ_TRACE_QUAD_MAX_RAYS="1000000"))):
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        measure_small(label, n)
