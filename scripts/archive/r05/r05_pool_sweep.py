#!/usr/bin/env python3
"""Round 5: the pool form of the rtdgi ray passes (k_rtdgi_rays_pool) against the fused form, and a sweep of its scheduling knobs.
One process, one scene, the same pre-generated frames replayed for every configuration; serial frames with the per-pass HIP-event timers on.
  python scripts/r05_pool_sweep.py [--scene city --tris 1000000 --width 1920 --height 1080] [--quick]
Prints one JSON line per configuration: mean `rtdgi trace` ms (all frames) and `rtdgi validate` ms (validation frames only)."""
import argparse
import ctypes as C
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="city")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tile-waves", action="store_true", help="only the one-wave-per-tile launches of the pool form (waves_per_simd = 0): thresholds of the two shading blocks")
    args = ap.parse_args()
    import torch
    from kajiya_amd import lib
    W, H = args.width, args.height
    desc, cam_args, label = bench.make_scene(args.scene, args.tris)
    dev = lib.Device(0)
    scene = lib.Scene(dev, desc)
    gp = lib.GpuPipeline(dev, scene, W, H, use_ircache=True)
    n = args.frames + 6
    fcs = bench.frame_constants_list(W, H, n, cam_args)
    inputs = []
    for fc in fcs:
        gp.render_inputs(fc)
        gp.reprojection()
        rp = lib.tensor_from_ptr(gp.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((gp.geometric_normal.clone(), gp.gbuffer.clone(), gp.depth.clone(), rp))
    torch.cuda.synchronize()

    def step(i):
        gn, gb, d, rp = inputs[i]
        dev.frame_begin(fcs[i])
        gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
        gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
        gp.ssgi_frame()
        gp.gi_frame()
        gp.taa_frame()

    for i in range(6):      # warm the histories and the cache
        step(i)
    torch.cuda.synchronize()
    gp.set_profiling(True, False)

    def measure(name):
        tr, va = [], []
        for _ in range(args.rounds):
            for i in range(6, n):
                step(i)
                torch.cuda.synchronize()
                t = gp.pass_times_ms()
                tr.append(t[3])
                if fcs[i].frame_index % 3 == 0:
                    va.append(t[2])
        rec = {"config": name, "trace_ms": round(sum(tr) / len(tr), 4), "trace_min_ms": round(min(tr), 4), "validate_ms": round(sum(va) / max(1, len(va)), 4), "rays": gp.ray_counts()}
        print(json.dumps(rec), flush=True)
        return rec

    results = []
    gp.set_ray_pass_form("fused")
    results.append(measure("fused"))
    gp.set_ray_pass_form("pool")
    if args.tile_waves:
        grid = [(0, 64, a, b, 0) for a, b in ((64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (1, 1), (16, 64), (64, 16), (8, 32), (32, 8))]
    elif args.quick:
        grid = [(3, 16, 16, 16, 0), (2, 16, 16, 16, 0), (4, 16, 16, 16, 0), (3, 8, 16, 16, 0), (3, 24, 24, 24, 0), (3, 16, 16, 16, 1)]
    else:
        grid = []
        for w in (2, 3, 4):
            for r_, ab in itertools.product((8, 16, 24, 32), (8, 16, 24, 32)):
                grid.append((w, r_, ab, ab, 0))
        grid += [(3, 16, 8, 24, 0), (3, 16, 24, 8, 0), (3, 16, 16, 16, 1), (2, 16, 16, 16, 1), (4, 16, 16, 16, 1), (3, 1, 1, 1, 0), (3, 64, 64, 64, 0), (1, 16, 16, 16, 0)]
    for w, r_, a, b, d in grid:
        gp.set_pool_tune(w, r_, a, b, bool(d))
        results.append(measure(f"pool w{w} r{r_} a{a} b{b} dyn{d}"))
    gp.set_ray_pass_form("fused")
    results.append(measure("fused (again)"))
    best = min(results[1:-1], key=lambda r: r["trace_ms"])
    print(json.dumps({"workload": f"{label} {W}x{H}", "fused": results[0], "fused_again": results[-1], "best_pool": best}), flush=True)


if __name__ == "__main__":
    main()
