#!/usr/bin/env python3
"""Round 5: does replaying a captured frame help the launch-bound sizes? One period (6 frames: the 3-frame validation cadence x the 2-deep ping-pongs)
of the 512x512 Cornell frame -- SSAO guide, cache, rtdgi, TAA: ~35 dependent launches per frame -- captured with stream capture into ONE HIP graph and replayed, against
issuing the same calls. A TIMING experiment: the captured launches carry their frame constants by value (kj_frame_begin's kernarg), so every replay renders the same six
camera positions again on the histories the previous replay left. python scripts/r05_hip_graph_probe.py [--width 512 --height 512 --scene cornell]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell"); ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=512); ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--periods", type=int, default=40)
    args = ap.parse_args()
    import torch
    from kajiya_amd import lib
    W, H = args.width, args.height
    desc, cam_args, label = bench.make_scene(args.scene, args.tris)
    dev = lib.Device(0); scene = lib.Scene(dev, desc)
    gp = lib.GpuPipeline(dev, scene, W, H, use_ircache=True)
    n = 18
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

    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    out = {"workload": f"{label} {W}x{H}, serial frames (one stream), period of 6 frames = frames 12..17"}

    def timed(fn, reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (6 * reps) * 1e3, t_issue / (6 * reps) * 1e3

    def period():
        for i in range(12, 18):
            step(i)
    period(); period()
    out["issued_ms_per_frame"], out["issued_host_ms_per_frame"] = (round(v, 4) for v in timed(period, args.periods))
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            period()      # once more on the capture stream, uncaptured (handles that create per-stream state do it here)
            torch.cuda.synchronize()
            g.capture_begin()
            period()
            g.capture_end()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        out["graph_replay_ms_per_frame"], out["graph_replay_host_ms_per_frame"] = (round(v, 4) for v in timed(g.replay, args.periods))
    except Exception as e:
        out["graph_error"] = repr(e)[:300]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
