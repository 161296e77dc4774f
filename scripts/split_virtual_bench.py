#!/usr/bin/env python3
"""GPU work per rank of the screen-tile split, measured on ONE GPU (bench.py's `also[]` lines `split_virtual_4` / `split_virtual_8`; VERDICT r5 next #1g).

The GI frame (SSAO guide + irradiance cache + rtdgi + TAA) of the 4K ruins workload is rendered
  (1) unsplit, serially, by one pipeline (the reference's racy cache, and once more with the deterministic cache every rank of a split runs), and
  (2) by N virtual ranks through the compiled orchestrator (csrc/split.cpp, every rank in this process, exchanges = device-to-device copies),
all on one stream, timed with HIP events around K frames. The ranks of (2) execute one after the other, so
    per-rank work = (frame time of all N ranks - time inside the exchanges) / N,
the exchanges' share coming from the orchestrator's own event pairs around every exchange (kj_split_set_profiling). That is what ONE rank of an N-GPU job
executes per frame without the wire; the wire itself (RCCL over xGMI) cannot be measured on one GPU -- the bytes arriving at the busiest rank are reported
instead. Host synchronisations per frame: the host issues all K frames before it waits once; `host_issue_ms_per_frame` < the GPU time per frame shows it
ran ahead (a blocking call inside the frame would pin the two together), and scripts/r06/ (rocprofv3 --hip-trace) counts hipStreamSynchronize calls.
usage: split_virtual_bench.py [--ranks 4,8] [--res 3840x2160] [--scene ruins] [--tris 4000000] [--frames 10] [--warmup 4]"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame, multigpu

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", default="4,8"); ap.add_argument("--res", default="3840x2160"); ap.add_argument("--scene", default="ruins"); ap.add_argument("--tris", type=int, default=4_000_000)
ap.add_argument("--frames", type=int, default=10); ap.add_argument("--warmup", type=int, default=4); ap.add_argument("--motion-halo", type=int, default=16)
ap.add_argument("--no-ssgi", action="store_true")
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
K, Wm = a.frames, a.warmup
torch.cuda.set_device(0)
dev = lib.Device(0)
if a.scene == "ruins":
    desc, cam = scenes.procedural_ruins(target_tris=a.tris, seed=5678), (lambda i: frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004))
else:
    desc, cam = scenes.procedural_city(target_tris=a.tris, seed=1234), (lambda i: frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004))
scene = lib.Scene(dev, desc)
mk = lambda: lib.GpuPipeline(dev, scene, W, H, device="cuda:0", use_ircache=True)
fs = frame.FrameState((W, H)); fs.ircache_enabled = True
gp0 = mk()
fcs, inputs = [], []
for i in range(Wm + K):
    fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
    gp0.render_inputs(fc); gp0.reprojection()
    rp = lib.tensor_from_ptr(gp0.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
    fcs.append(fc); inputs.append((gp0.geometric_normal.clone(), gp0.gbuffer.clone(), gp0.depth.clone(), rp))
torch.cuda.synchronize()


def bind(pipes, i):
    gn, gb, d, rp = inputs[i]
    for q in pipes:
        q.geometric_normal, q.gbuffer, q.depth = gn, gb, d
        q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())


def timed(step):
    """K frames after Wm warm-up frames, all issued before the one wait: (GPU ms per frame by HIP events, host issue ms per frame)."""
    for i in range(Wm):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(Wm, Wm + K):
        step(i)
    e1.record()
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K, 1e3 * issue / K


def one_gpu(deterministic):
    gp = mk()
    if deterministic:
        gp.ircache_set_deferred(True)

    def step(i):
        dev.frame_begin(fcs[i]); bind([gp], i)
        if not a.no_ssgi:
            gp.ssgi_frame()
        gp.gi_frame(); gp.taa_frame()
    return timed(step)


out = {"workload": f"procedural_{a.scene} ~{a.tris} tris @ {W}x{H}: SSAO guide + ircache + rtdgi + TAA, serial frames on one stream", "frames": K, "warmup": Wm}
ms_racy, issue_racy = one_gpu(False)
ms_det, issue_det = one_gpu(True)
out["one_gpu"] = {"serial_frame_ms_racy_cache": round(ms_racy, 4), "serial_frame_ms_deterministic_cache": round(ms_det, 4), "host_issue_ms_per_frame": round(issue_racy, 3)}
out["split"] = []
for n in [int(v) for v in a.ranks.split(",") if v]:
    pipes = {r: mk() for r in range(n)}
    split = multigpu.NativeSplit(n, pipes, W, H, motion_halo=a.motion_halo)
    assert split.self_test() is True

    def step(i):
        dev.frame_begin(fcs[i]); bind(pipes.values(), i)
        if not a.no_ssgi:
            split.ssgi_frame()
        split.gi_frame(); split.taa_frame()
    for i in range(Wm):      # warm up outside the profiled region
        step(i)
    torch.cuda.synchronize()
    lib.check(split.L.kj_split_set_profiling(split.h, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(Wm, Wm + K):
        step(i)
    e1.record()
    issue = 1e3 * (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    all_ms = e0.elapsed_time(e1) / K
    from kajiya_amd.abi import KjSplitProfile
    prof = KjSplitProfile()
    lib.check(split.L.kj_split_profile(split.h, C.byref(prof)))
    lib.check(split.L.kj_split_set_profiling(split.h, 0))
    ex_ms = prof.exchange_ms / K
    work = (all_ms - ex_ms) / n
    out["split"].append({
        "what": f"split_virtual_{n}", "ranks": n, "orchestrator": "compiled (csrc/split.cpp), virtual ranks, one stream",
        "all_ranks_frame_gpu_ms": round(all_ms, 4), "per_rank_gpu_ms": round(all_ms / n, 4),
        "exchange_ms_per_rank": round(ex_ms / n, 4), "per_rank_work_ms": round(work, 4),
        "speedup_on_work_alone_vs_racy_serial_frame": round(ms_racy / work, 3), "speedup_on_work_alone_vs_deterministic_serial_frame": round(ms_det / work, 3),
        "exchange_points_per_frame": round(prof.exchange_points / K, 2), "exchange_MB_arriving_at_busiest_rank_per_frame": round(prof.exchange_bytes_busiest_rank / K / 1e6, 2),
        "host_issue_ms_per_frame_all_ranks": round(issue, 3), "host_ran_ahead": bool(issue < 0.8 * all_ms), "host_syncs_per_frame": 0,
        "host_syncs_note": "no hipStreamSynchronize / device-to-host copy in kj_split_{ssgi,gi,taa}_frame since round 6 (fixed-size summaries); all frames were issued before the one wait",
    })
    split.close()
    del pipes, split
    torch.cuda.empty_cache()
print(json.dumps(out))
