#!/usr/bin/env python3
"""Where does the ReSTIR GI lose energy against the reference path tracer? (mean ratio 0.88 on Cornell.)
CPU only, oracle only (the GPU kernels are parity-tested against it): a static camera on the Cornell box, 128x128.

 1. path tracer (first_bounce_mode 2: indirect light through a white Lambert first bounce = what rtdgi estimates), truncated at
    k = 2, 3, 4, 5, 6, 8, 16 eye-path vertices (OKJ_PT_MAX_PATH_LENGTH): E_k = energy carried by the first k - 1 bounces;
 2. time-averaged rtdgi output (96 warm-up + 96 averaged frames) with the irradiance cache on / off and with the screen-space
    depth gate of diffuse_trace_common.inc.hlsl:85-107 at its 5e-3, wide open (any on-screen hit reuses last frame's output) and
    shut (every hit goes to the cache) -- OKJ_RTDGI_DEPTH_GATE.
Each configuration runs in a fresh process (the knobs are read once). Prints one JSON object."""
import json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from kajiya_amd import scenes, frame
from oracle import okj_py
W = H = 128
mode, n = sys.argv[1], int(sys.argv[2])
osc = okj_py.OracleScene(scenes.cornell_box())
cam = lambda: frame.orbit_camera(0, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01)
if mode == "pt":
    fs = frame.FrameState((W, H))
    acc = np.zeros((H, W, 4), np.float32)
    for i in range(n):
        fc = fs.prepare_frame_constants(cam()); fs.retire_frame()
        okj_py.reference_path_trace(osc, fc, acc, first_bounce_mode=2)
    img = acc[..., :3]          # the accumulator keeps the running mean in rgb and the sample count in w (reference.rs / okj_reference_pt.hpp)
else:
    irc = mode == "gi_irc"
    op = okj_py.OraclePipeline(osc, W, H, use_ircache=irc)
    fs = frame.FrameState((W, H)); fs.ircache_enabled = irc
    acc = np.zeros((H, W, 3), np.float64)
    for i in range(2 * n):
        fc = fs.prepare_frame_constants(cam()); fs.retire_frame()
        op.frame(fc)
        if i >= n:
            acc += op.surface("spatial_filtered_tex", np.float16, (H, W, 4))[..., :3].astype(np.float64)
    img = acc / n
    m = op.depth > 0
    np.save(sys.argv[3] + ".mask.npy", m)
np.save(sys.argv[3], img.astype(np.float32))
''' % os.path.abspath(ROOT)


def run(mode, n, out, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    subprocess.check_call([sys.executable, "-c", WORKER, mode, str(n), out], env=e, cwd=ROOT)


def main():
    import numpy as np
    tmp = "/tmp/pt_deficit"
    os.makedirs(tmp, exist_ok=True)
    spp, frames = int(os.environ.get("KJ_ATTR_SPP", 96)), int(os.environ.get("KJ_ATTR_FRAMES", 96))
    res = {"scene": "cornell_box 128x128, static camera", "pt_spp": spp, "gi_frames_averaged": frames}
    run("gi_irc", frames, f"{tmp}/gi_ref.npy")
    mask = np.load(f"{tmp}/gi_ref.npy.mask.npy")
    mean = lambda p: float(np.load(p)[mask].mean())
    pt = {}
    for k in (2, 3, 4, 5, 6, 8, 16):
        run("pt", spp, f"{tmp}/pt_{k}.npy", OKJ_PT_MAX_PATH_LENGTH=k)
        pt[k] = mean(f"{tmp}/pt_{k}.npy")
    full = pt[16]
    res["path_tracer_mean_by_max_path_length"] = pt
    res["path_tracer_fraction_of_full"] = {k: v / full for k, v in pt.items()}
    gi = {"ircache on, gate 5e-3 (the reference's)": mean(f"{tmp}/gi_ref.npy")}
    for label, mode, gate in (("ircache off, gate 5e-3", "gi_noirc", None), ("ircache on, gate wide open", "gi_irc", 1e9), ("ircache on, gate shut", "gi_irc", 0.0),
                              ("ircache off, gate wide open", "gi_noirc", 1e9)):
        env = {} if gate is None else {"OKJ_RTDGI_DEPTH_GATE": gate}
        run(mode, frames, f"{tmp}/gi_x.npy", **env)
        gi[label] = mean(f"{tmp}/gi_x.npy")
    res["rtdgi_mean"] = gi
    res["rtdgi_over_full_path_tracer"] = {k: v / full for k, v in gi.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
