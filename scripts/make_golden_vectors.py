#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_cornell_32.npz: outputs of the CPU oracle on fixed inputs (Cornell box, 32x32, seeded
rays). The reference itself cannot be built in this environment, so these vectors pin the ORACLE against silent
changes, not against kajiya; tests/test_oracle.py::test_oracle_matches_golden_vectors compares a fresh run to them."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from oracle import okj_py
from kajiya_amd import scenes, frame


def generate():
    desc = scenes.cornell_box()
    osc = okj_py.OracleScene(desc)
    W = H = 32
    lo, hi = desc.bounds()
    rng = np.random.RandomState(20260921)
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), size=(512, 3))
    d = rng.uniform(lo, hi, size=(512, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((512, 8), np.float32)
    rays[:, 0:3] = o; rays[:, 4:7] = d; rays[:, 7] = 1e4
    out = {"rays": rays, "hits": osc.trace_closest(rays), "any": osc.trace_any(rays)}
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = True
    op = okj_py.OraclePipeline(osc, W, H)
    opi = okj_py.OraclePipeline(osc, W, H, use_ircache=True)
    pt = np.zeros((H, W, 4), np.float32)
    for i in range(6):
        fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01))
        fs.retire_frame()
        op.frame(fc)
        opi.render_inputs(fc); opi.reprojection(fc); opi.gi_frame(fc)
        if i < 4:
            okj_py.reference_path_trace(osc, fc, pt)
    out["rtdgi_spatial_filtered"] = op.surface("spatial_filtered_tex", np.uint16, (H, W, 4)).copy()
    out["rtdgi_reservoir"] = op.surface("rtdgi.reservoir:1", np.uint32, (H // 2, W // 2, 2)).copy()
    out["gi_with_ircache_mean"] = opi.surface("spatial_filtered_tex", np.float16, (H, W, 4)).astype(np.float32)[..., :3].mean(axis=(0, 1))
    out["ircache_entry_count"] = np.array([int(opi.ircache_buffer("meta", np.uint32)[2])])
    out["depth"] = op.depth.copy()
    out["gbuffer"] = op.gbuffer.copy()
    out["reference_pt"] = pt
    out["brdf_lut"] = okj_py.brdf_lut().copy()
    return out


def generate_rtr():
    """Reflections (oracle/okj_rtr.hpp) on the glossy test scene, 48x32, 5 frames, stand-in sampler tables (kajiya_amd/rtr_tables.py)."""
    desc = scenes.glossy_test_scene()
    W, H = 48, 32
    op = okj_py.OraclePipeline(okj_py.OracleScene(desc), W, H)
    fs = frame.FrameState((W, H))
    for i in range(5):
        fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 1.0, 0.0), radius=9.0, height=3.0, rate=0.01))
        fs.retire_frame()
        op.frame(fc)
        res = op.rtr_frame(fc)
    return {"resolved": res.copy(), "temporal": op.rtr_surface("rtr.temporal:0", np.uint16, (H, W, 4)).copy(),
            "irradiance": op.rtr_surface("rtr.irradiance:0", np.uint16, (H // 2, W // 2, 4)).copy(),
            "reservoir": op.rtr_surface("rtr.reservoir:0", np.uint32, (H // 2, W // 2, 2)).copy(),
            "rng": op.rtr_surface("rtr.rng:0", np.uint32, (H // 2, W // 2)).copy(),
            "ray_counts": np.array(op.rtr_ray_counts(), np.uint64)}


if __name__ == "__main__":
    r = generate_rtr()
    path = os.path.join(ROOT, "tests", "golden", "oracle_rtr_glossy_48x32.npz")
    np.savez_compressed(path, **r)
    print("wrote", path, os.path.getsize(path), "bytes")
    okj_py.lib().okj_set_threads(1)   # the ircache passes are order-dependent: one thread = deterministic (the test does the same)
    g = generate()
    path = os.path.join(ROOT, "tests", "golden", "oracle_cornell_32.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes")
