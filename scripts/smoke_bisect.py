#!/usr/bin/env python3
"""why did smoke()'s second leg (cache + rtdgi + TAA, 6 free-running frames at 128^2, TAA output vs the oracle) go from
rel-L2 7.4e-2 (round 1's driver run) to 1.5e-1 (round 2's)? Runs the same six frames with each side's irradiance cache in its racy
(reference: atomics as they fall) or deterministic mode (deferred, canonically ordered updates -- kj_ircache_set_deferred_updates /
oracle okj_ircache.hpp `deferred`), twice each, and prints rel-L2 of the TAA output and of the GI image per combination.
On a GPU box: python scripts/smoke_bisect.py > gpurun_out/smoke_bisect.json   (KJ_HIP_EMU=fast: the CPU stand-in for HIP)"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
if os.environ.get("KJ_HIP_EMU"):
    sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
    import build_emu, cpu_as_cuda
    cpu_as_cuda.install(build_emu.build())
import numpy as np
import torch
from kajiya_amd import lib, scenes, frame
from oracle import okj_py

W = H = 128
desc = scenes.cornell_box()
dev = lib.Device(0)
gscene, oscene = lib.Scene(dev, desc), okj_py.OracleScene(desc)


def run(gpu_det, ora_det, frames=6):
    gp = lib.GpuPipeline(dev, gscene, W, H, use_ircache=True)
    op = okj_py.OraclePipeline(oscene, W, H, use_ircache=True)
    gp.ircache_set_deferred(gpu_det); op.ircache_set_deferred(ora_det)
    fs = frame.FrameState((W, H)); fs.ircache_enabled = True
    for i in range(frames):
        fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01))
        gp.render_inputs(fc); gp.reprojection(); gp.gi_frame(); gp.taa_frame()
        op.render_inputs(fc); op.reprojection(fc); op.gi_frame(fc); op.taa_frame(fc)
        fs.retire_frame()
    torch.cuda.synchronize()
    rel = lambda a, b: float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
    t = f"taa:{frames % 2}"
    taa = rel(gp.taa_surface(t, torch.float16, (H, W, 4)).float().cpu().numpy()[..., :3], op.taa_surface(t, np.float16, (H, W, 4)).astype(np.float32)[..., :3])
    gi = rel(gp.surface("spatial_filtered_tex", torch.float16, (H, W, 4)).float().cpu().numpy()[..., :3], op.surface("spatial_filtered_tex", np.float16, (H, W, 4)).astype(np.float32)[..., :3])
    return dict(taa_rel_l2=taa, gi_rel_l2=gi)


out = {"what": "cornell 128x128, 6 free-running frames, cache + rtdgi + TAA; product vs oracle", "device": torch.cuda.get_device_name(0), "runs": []}
for gpu_det, ora_det in ((False, False), (False, False), (False, False), (True, False), (False, True), (True, True), (True, True)):
    r = run(gpu_det, ora_det)
    r.update(product_cache="deterministic" if gpu_det else "racy", oracle_cache="deterministic" if ora_det else "racy")
    out["runs"].append(r)
    print(r, file=sys.stderr)
print(json.dumps(out, indent=1))
