#!/usr/bin/env python3
"""Round 5: what the waves of the two forms of the rtdgi ray passes actually issue (instrumented STATS build of the kernels): node / triangle steps per wave,
lane utilisation of the walk, shading blocks and their fill. python scripts/r05_pool_probe.py [--scene ... --width ... --height ...]"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="city"); ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tunes", default="3,16,16,16,0;4,16,16,16,0;3,64,64,64,0;3,32,32,32,0")
    args = ap.parse_args()
    import torch
    from kajiya_amd import lib
    W, H = args.width, args.height
    desc, cam_args, label = bench.make_scene(args.scene, args.tris)
    dev = lib.Device(0); scene = lib.Scene(dev, desc)
    gp = lib.GpuPipeline(dev, scene, W, H, use_ircache=True)
    n = 10
    fcs = bench.frame_constants_list(W, H, n, cam_args)
    inputs = []
    for fc in fcs:
        gp.render_inputs(fc); gp.reprojection()
        rp = lib.tensor_from_ptr(gp.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((gp.geometric_normal.clone(), gp.gbuffer.clone(), gp.depth.clone(), rp))

    def step(i):
        gn, gb, d, rp = inputs[i]
        dev.frame_begin(fcs[i])
        gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
        gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
        gp.ssgi_frame(); gp.gi_frame(); gp.taa_frame()

    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    ptr, nb = bench._counter_ptr(gp, lib)
    ctr = lib.tensor_from_ptr(ptr, nb, torch.int64, (64, 16))
    gp.set_profiling(True, True)
    configs = [("fused", None)] + [("pool " + t, tuple(int(v) for v in t.split(","))) for t in args.tunes.split(";")]
    for name, tune in configs:
        gp.set_ray_pass_form("fused" if tune is None else "pool")
        if tune:
            gp.set_pool_tune(*tune[:4], bool(tune[4]))
        for i in (7, 8):      # frame 7 and 8: tracing frames (frame_index % 3 != 0): the counters are the trace pass' alone
            step(i); torch.cuda.synchronize()
            c = ctr.sum(dim=0).tolist(); t = gp.pass_times_ms()
            lane_steps = c[2] + c[3] + c[4] + c[5]
            wave_steps = c[6] + c[7] + c[8] + c[9]
            rec = {"config": name, "frame_index": int(fcs[i].frame_index), "trace_ms_instrumented": round(t[3], 4), "closest_rays": c[0], "any_rays": c[1], "lane_steps": lane_steps, "wave_steps": wave_steps,
                   "walk_lane_utilisation": round(lane_steps / (64.0 * max(1, wave_steps)), 4), "wave_node_steps": c[6] + c[8], "wave_tri_steps": c[7] + c[9],
                   "refill_blocks": c[10], "shade_a_blocks": c[11], "shade_b_blocks": c[12], "lanes_per_shade_a": round(c[13] / max(1, c[11]), 1), "lanes_per_shade_b": round(c[14] / max(1, c[12]), 1)}
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
