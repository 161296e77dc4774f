#!/usr/bin/env python3
"""BASELINE configs[2] under the screen-tile split: the whole lighting frame of scripts/config3_bench.py (SSAO guide, sun shadows + denoiser, irradiance cache +
rtdgi, reflections, deferred combine, TAA on the lit image) strip by strip through multigpu.lighting_frame -- N virtual ranks on one GPU (--virtual-ranks N:
for rocprofv3, sum of kernel + copy durations / frames / N = the GPU work of one rank, as scripts/archive/r03_virtual_split_gpu_time.sh does for the GI frame) or one
process per GPU under `python -m torch.distributed.run --nproc-per-node N ... scripts/config3_split_bench.py` (the compiled orchestrator over RCCL, certified
by its self-test before frame 0; KJ_SPLIT_NATIVE=0: the Python orchestrator over torch.distributed). Frames are issued serially (no cache pipelining): the wall time of
a virtual-rank run is N ranks' work plus their host syncs on one GPU and says nothing; of an N-process run it is the frame time (max over ranks).
usage: config3_split_bench.py [--res WxH] [--tris N] [--scene ruins|glossy] [--frames K] [--warmup W] [--virtual-ranks N] [--motion-halo M] [--check]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
from kajiya_amd import lib, scenes, frame, multigpu

ap = argparse.ArgumentParser()
ap.add_argument("--res", default="2560x1440"); ap.add_argument("--tris", type=int, default=4_000_000); ap.add_argument("--scene", default="ruins")
ap.add_argument("--frames", type=int, default=30); ap.add_argument("--warmup", type=int, default=12); ap.add_argument("--virtual-ranks", type=int, default=2)
ap.add_argument("--motion-halo", type=int, default=16); ap.add_argument("--check", action="store_true", help="also render every frame unsplit and compare this rank's rows")
ap.add_argument("--work", action="store_true", help="GPU work per rank (virtual ranks, compiled orchestrator): all frames issued serially without a wait, HIP events around them, minus the orchestrator's own event pairs around every exchange (kj_split_set_profiling), / N; and the same frames on one GPU")
ap.add_argument("--pipelined", action="store_true", help="lighting_frame_pipelined: the cache's work of frame N+1 and the replay of frame N's updates on a side stream (inputs of all frames pre-generated; no --check)")
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
share = bool(os.environ.get("KJ_BENCH_SHARE_GPU0"))      # debugging aid (as in bench.py): every process on GPU 0, exchanges over gloo
if share:
    local_rank = 0
torch.cuda.set_device(local_rank)
if world > 1:
    dist.init_process_group("gloo") if share else dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
n = world if world > 1 else a.virtual_ranks
dev = lib.Device(local_rank)
desc = scenes.glossy_test_scene() if a.scene == "glossy" else scenes.procedural_ruins(target_tris=a.tris, seed=5678)
scene = lib.Scene(dev, desc)
mk = lambda: lib.GpuPipeline(dev, scene, W, H, device=f"cuda:{local_rank}", use_ircache=True)
pipes = {rank: mk()} if world > 1 else {r: mk() for r in range(n)}
native = os.environ.get("KJ_SPLIT_NATIVE", "0" if share else "1") == "1"
if native:
    nccl = multigpu.NativeSplit.rccl_comm_from_torch(dist, rank, world, f"cuda:{local_rank}") if world > 1 else None
    split = multigpu.NativeSplit(n, pipes, W, H, motion_halo=a.motion_halo, nccl_comm=nccl)
    ok = split.self_test(dist if world > 1 else None)
else:
    comm = multigpu.DistComm(dist, rank, world, stage_through_host=share) if world > 1 else multigpu.LocalComm(n)
    split = multigpu.SplitRtdgi(comm, pipes, W, H, motion_halo=a.motion_halo)
    ok = split.self_test()
if rank == 0:
    print(f"[config3 split] {'compiled' if native else 'python'} orchestrator, {n} {'processes' if world > 1 else 'virtual ranks'}: exchange self-test {'OK' if ok else 'FAILED'}", file=sys.stderr, flush=True)
assert ok, "split transport self-test failed"
split.enable_rtr()
ref = None
if a.check:
    ref = mk(); ref.ircache_set_deferred(True); ref.ircache_set_rtr_requests(True)
fs = frame.FrameState((W, H), sun_size_multiplier=4.0); fs.ircache_enabled = True
cam = (lambda i: frame.orbit_camera(i, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.004)) if a.scene == "glossy" else \
      (lambda i: frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004))
mine = sorted(pipes)
worst, t_acc = 0, 0.0
if a.work:
    import ctypes as C
    from kajiya_amd.abi import KjSplitProfile
    assert world == 1 and native and not a.check and not a.pipelined, "--work: virtual ranks, compiled orchestrator"
    K = a.warmup + a.frames
    fcs, inputs = [], []
    gp0 = pipes[mine[0]]
    for i in range(K):
        fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
        gp0.render_inputs(fc); gp0.reprojection()
        rp = lib.tensor_from_ptr(gp0.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        fcs.append(fc); inputs.append((gp0.geometric_normal.clone(), gp0.gbuffer.clone(), gp0.depth.clone(), rp, gp0.sky16.clone(), gp0.sky64.clone()))

    def bind_to(qs, i):
        for q in qs:
            q.geometric_normal, q.gbuffer, q.depth, rp, q.sky16, q.sky64 = inputs[i]
            q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())

    def timed(step):
        for i in range(a.warmup):
            step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for i in range(a.warmup, K):
            step(i)
        e1.record(); issue = 1e3 * (time.perf_counter() - t0) / a.frames
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.frames, issue
    one = mk(); one.ircache_set_rtr_requests(True)

    def one_step(i):      # scripts/config3_bench.py's order, serial, the reference's racy cache
        dev.frame_begin(fcs[i]); bind_to([one], i)
        one.ssgi_frame(); sh = one.shadow_denoise(one.sun_shadow_mask()); one.gi_frame(); rt = one.rtr_frame()
        lit = one.light_gbuffer(sh, rtr_ptr=rt.data_ptr())[1]; one.taa_frame(input_ptr=lit.data_ptr())
    one_ms, _ = timed(one_step)

    def split_step(i):
        dev.frame_begin(fcs[i]); bind_to(pipes.values(), i)
        split.lighting_frame()
    for i in range(a.warmup):
        split_step(i)
    torch.cuda.synchronize()
    lib.check(split.L.kj_split_set_profiling(split.h, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(a.warmup, K):
        split_step(i)
    e1.record(); issue = 1e3 * (time.perf_counter() - t0) / a.frames
    torch.cuda.synchronize()
    all_ms = e0.elapsed_time(e1) / a.frames
    prof = KjSplitProfile(); lib.check(split.L.kj_split_profile(split.h, C.byref(prof)))
    ex = prof.exchange_ms / a.frames
    work = (all_ms - ex) / n
    print(json.dumps({"config": "BASELINE configs[2] under the screen-tile split: GPU work per rank", "workload": f"{a.scene} @ {W}x{H}", "ranks": n, "one_gpu_serial_frame_ms": round(one_ms, 4),
                      "all_ranks_frame_gpu_ms": round(all_ms, 4), "exchange_ms_per_rank": round(ex / n, 4), "per_rank_work_ms": round(work, 4), "speedup_on_work_alone": round(one_ms / work, 3),
                      "exchange_points_per_frame": round(prof.exchange_points / a.frames, 2), "exchange_MB_arriving_at_busiest_rank_per_frame": round(prof.exchange_bytes_busiest_rank / a.frames / 1e6, 2),
                      "host_issue_ms_per_frame_all_ranks": round(issue, 3), "host_ran_ahead": bool(issue < 0.8 * all_ms)}))
    a.frames = a.warmup = 0      # (nothing below runs: the script ends by falling off its end, which rocprofv3's finalisation needs)
if a.pipelined:
    import ctypes as C
    assert not a.check, "--check compares serial frames"
    K = a.warmup + a.frames
    fcs, inputs = [], []
    gp0 = pipes[mine[0]]
    for i in range(K + 1):
        fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
        gp0.render_inputs(fc); gp0.reprojection()
        rp = lib.tensor_from_ptr(gp0.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        fcs.append(fc); inputs.append((gp0.geometric_normal.clone(), gp0.gbuffer.clone(), gp0.depth.clone(), rp, gp0.sky16.clone(), gp0.sky64.clone()))

    def bind(i):
        for r in mine:
            q = pipes[r]
            q.geometric_normal, q.gbuffer, q.depth, rp, q.sky16, q.sky64 = inputs[i]
            q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
    bind(0); split.pipeline_begin(fcs[0])
    for i in range(K):
        if i == a.warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        bind(i)
        split.lighting_frame_pipelined(fcs[i + 1] if i + 1 < K else None)
    torch.cuda.synchronize()
    t_acc = time.perf_counter() - t0
for i in range(0 if a.pipelined else a.warmup + a.frames):
    fc = fs.prepare_frame_constants(cam(i)); fs.retire_frame()
    for r in mine:                       # inputs are replicated: every rank rasterises the whole G-buffer (outside the timed part)
        pipes[r].render_inputs(fc); pipes[r].reprojection()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    lits = split.lighting_frame()
    torch.cuda.synchronize()
    if i >= a.warmup:
        t_acc += time.perf_counter() - t0
    if ref is not None:                  # the same frame on one GPU, scripts/config3_bench.py's order
        ref.render_inputs(fc); ref.reprojection(); ref.ssgi_frame()
        sh = ref.shadow_denoise(ref.sun_shadow_mask()); ref.gi_frame(defer_replay=True); rt = ref.rtr_frame(); ref.ircache_replay_own_requests()
        lit = ref.light_gbuffer(sh, rtr_ptr=rt.data_ptr())[1]; ref.taa_frame(input_ptr=lit.data_ptr())
        torch.cuda.synchronize()
        ta = ref.taa_surface(f"taa:{i % 2}", torch.int16, (H, W, 4))
        for r in mine:
            r0, r1 = split.strips[r]
            worst = max(worst, int((ta[r0:r1] != pipes[r].taa_surface(f"taa:{i % 2}", torch.int16, (H, W, 4))[r0:r1]).sum()), int((lit.view(torch.int16)[r0:r1] != lits[r].view(torch.int16)[r0:r1]).sum()))
ms = 1e3 * t_acc / max(1, a.frames)
if world > 1:
    t = torch.tensor([ms, float(worst)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else f"cuda:{local_rank}")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms, worst = float(t[0]), int(t[1])
if rank == 0 and not a.work:
    reach = multigpu.rtr_resolve_halo(H, pipes[mine[0]].dev.clip_to_view_11)
    print(json.dumps({"config": "BASELINE configs[2] under the screen-tile split", "workload": f"{a.scene} @ {W}x{H}", "ranks": n, "processes": world, "orchestrator": "compiled" if native else "python",
                      "frame_ms_wall": round(ms, 4), "pipelined": a.pipelined, "wall_is": "max over ranks, " + ("pipelined frames" if a.pipelined else "serial issue") if world > 1 else "N virtual ranks' work + host syncs on ONE GPU: not a frame time",
                      "frames": a.frames, "motion_halo": a.motion_halo, "rtr_resolve_halo_half_rows": reach, "strips": [list(s) for s in split.strips],
                      "mismatching_texels_vs_one_gpu": worst if a.check else None}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
