#!/usr/bin/env python3
"""When and where do the waves of the two fused rtdgi ray kernels run? (round 6: profiles/r06_ray_tile_order.md)
Needs a library built with -DKJ_WAVE_TIMELINE (scripts/build_variant.sh timeline "-DKJ_WAVE_TIMELINE=1"; KJ_AMD_LIB=kajiya_amd/libkajiya_amd_timeline.so): every wave
of k_rtdgi_validate_fused / k_rtdgi_trace_fused then stores {start, end (100 MHz wall clock), tile | sub << 24, XCC_ID | longest pixel's steps << 4} at its workgroup index. One JSON line per
(frame, kernel): the launch's span, the spread of wave durations, how many waves were resident over time, when the long waves started, and what an ideal
longest-first list schedule of the SAME durations on the same number of wave slots would take (durations depend on what runs beside them: indicative only).
usage: wave_timeline.py [--res WxH] [--scene city|ruins] [--tris N] [--order 0|1] [--split permille] [--dump file.npz]"""
import argparse, ctypes as C, heapq, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from kajiya_amd import lib, scenes, frame

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="1920x1080"); ap.add_argument("--scene", default="city"); ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--order", type=int, default=1); ap.add_argument("--split", type=int, default=80); ap.add_argument("--dump", default=""); ap.add_argument("--warm", type=int, default=9)
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
dev = lib.Device(0)
if a.scene == "ruins":
    desc, cam = scenes.procedural_ruins(target_tris=a.tris, seed=5678), dict(center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004)
else:
    desc, cam = scenes.procedural_city(target_tris=a.tris, seed=1234), dict(center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
gp = lib.GpuPipeline(dev, lib.Scene(dev, desc), W, H, use_ircache=True)
gp.set_cost_ordered_tiles(bool(a.order), a.split)
fs = frame.FrameState((W, H)); fs.ircache_enabled = True
tx, ty = ((W + 1) // 2 + 7) // 8, ((H + 1) // 2 + 7) // 8
n = tx * ty
cap = n + 3 * (n // 2)                                   # entries per kernel in the timeline surface
n_split = min(n // 2, (n * a.split + 999) // 1000) if a.order else 0
grid = n + 3 * n_split                                   # workgroups of a launch once the order is in use
SLOTS = 256 * 4 * 5      # wave slots of the chip at the kernels' five waves per SIMD


def list_schedule(durs, slots):
    """makespan of a list schedule: jobs in the given order, each to the slot that frees first"""
    h = [0.0] * min(slots, len(durs))
    heapq.heapify(h)
    for d in durs:
        heapq.heapreplace(h, h[0] + d)
    return max(h)


dump = {}
for i in range(a.warm + 3):
    fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), **cam)); fs.retire_frame()
    gp.frame(fc)
    if i < a.warm:
        continue
    torch.cuda.synchronize()
    tl = gp.surface("wave_timeline", torch.int32, (2, cap, 4)).cpu().numpy().astype(np.int64)[:, :grid] & 0xffffffff
    validation = int(fc.frame_index) % 3 == 0
    for k, name in enumerate(("validate", "trace")):
        t0, t1, ts_, xd = tl[k, :, 0], tl[k, :, 1], tl[k, :, 2], tl[k, :, 3]
        sub = ts_ >> 24
        dur = ((t1 - t0) & 0xffffffff) * 0.01      # us
        start = ((t0 - t0.min()) & 0xffffffff) * 0.01
        end = start + dur
        span = float(end.max())
        live = dur > 1.0                      # waves that did more than find their tile empty
        order_idx = np.arange(grid)
        xcc = xd & 15
        steps = xd >> 4
        # resident waves over time (sampled every span / 200)
        ts_ = np.linspace(0.0, span, 201)[:-1]
        resident = [(int(((start <= t) & (end > t)).sum())) for t in ts_]
        top = np.argsort(-dur)[: max(1, n // 100)]
        quad = sub != 0
        by_dispatch = dur[np.argsort(order_idx)]
        rec = {"frame_index": int(fc.frame_index), "validation_frame": validation, "kernel": name, "res": a.res, "scene": a.scene, "cost_ordered": bool(a.order),
               "tiles": n, "split_tiles": n_split, "waves": grid, "waves_with_work": int(live.sum()), "span_us": round(span, 1),
               "quad_wave_us": {q: round(float(np.percentile(dur[quad & live], p)), 1) for q, p in (("p50", 50), ("p90", 90), ("max", 100))} if (quad & live).any() else None,
               "whole_wave_us": {q: round(float(np.percentile(dur[~quad & live], p)), 1) for q, p in (("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))} if (~quad & live).any() else None,
               "steps_of_longest_pixel": {q: int(np.percentile(steps[live], p)) for q, p in (("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))} if live.any() else None,
               "corr_steps_vs_wave_us": round(float(np.corrcoef(steps[live], dur[live])[0, 1]), 3) if live.sum() > 2 else None,
               "wave_us": {q: round(float(np.percentile(dur[live], p)), 1) if live.any() else 0.0 for q, p in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
               "wave_us_mean": round(float(dur[live].mean()), 1) if live.any() else 0.0,
               "sum_wave_us_over_span_x_slots": round(float(dur.sum() / max(span, 1e-9) / SLOTS), 3),
               "resident_waves_at_10_25_50_75_90_pct_of_span": [resident[20], resident[50], resident[100], resident[150], resident[180]],
               "time_resident_below_half_peak_us": round(float(sum(1 for r_ in resident if r_ < max(resident) / 2) * span / 200), 1),
               "start_of_the_longest_1pct_waves_us": [round(float(start[top].min()), 1), round(float(np.median(start[top])), 1), round(float(start[top].max()), 1)],
               "last_wave_end_by_xcc_us": [round(float(end[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)],
               "sum_wave_us_by_xcc": [round(float(dur[xcc == x].sum())) for x in range(8)],
               "list_schedule_same_durations_us": {"this_dispatch_order": round(list_schedule(by_dispatch.tolist(), SLOTS), 1),
                                                   "longest_first": round(list_schedule(sorted(dur.tolist(), reverse=True), SLOTS), 1),
                                                   "lower_bound_sum_over_slots": round(float(dur.sum() / SLOTS), 1), "lower_bound_longest_wave": round(float(dur.max()), 1)}}
        print(json.dumps(rec), flush=True)
        dump[f"f{i}_{name}"] = tl[k]
if a.dump:
    np.savez_compressed(a.dump, **dump)
