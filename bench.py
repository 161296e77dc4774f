#!/usr/bin/env python3
"""bench.py — GI-frame throughput of the MI355X-native ReSTIR-GI hot path.

A "step" is one GI frame (IrcacheRenderer::prepare + trace_irradiance + sum-up, RtdgiRenderer::reproject + render:
every pass from `scroll cascades` to `rtdgi spatial`, then TaaRenderer::render) over one frame's inputs, which are generated beforehand and stay
resident in HBM (G-buffer, depth, normals, velocity, reprojection map, sky cube, BVH).

Workload (BASELINE.json configs[1] stand-in): procedural "city" scene, ~1.0 M triangles,
64 instances of 8 meshes (seed 1234) — `battle.ron`'s mesh is missing from the reference checkout —
at 1920x1080, slow orbiting camera, frames exercise both tracing and validation cadence.

Prints ONE JSON line (rank 0). `value` = Mrays/s over the whole job (BVH ray queries: closest-hit +
any-hit, counted on device); `gi_frame_ms` = ms per GI frame. `roofline` describes the dominant kernel;
`cpu_baseline` times the oracle (the CPU restatement, used here only as the reported baseline).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--scene", default="city", choices=["city", "ruins", "cornell", "pica"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-frames", type=int, default=20, help="frames of the CPU oracle in the cpu_baseline leg (at this run's extent)")
    ap.add_argument("--no-scalar-baseline", action="store_true", help="skip the extra single-thread oracle frame of the cpu_baseline leg (minutes at 4K)")
    ap.add_argument("--no-also", action="store_true", help="skip the two headline-size measurements appended as `also` (4K ruins GI frame; 1440p config-3 lighting frame)")
    ap.add_argument("--profile-frames", type=int, default=12)
    ap.add_argument("--no-ssgi", action="store_true", help="drive rtdgi with the constant SSAO guide instead of running SsgiRenderer each frame")
    ap.add_argument("--no-overlap", action="store_true", help="serial frames: do not overlap the next frame's ircache work with this frame's screen-space tail")
    ap.add_argument("--deterministic-cache", action="store_true", help="the irradiance cache in its deterministic mode (deferred, canonically ordered updates: the mode the 1e-3 cache parity is "
                    "tested in and the mode every screen-tile split runs) for the timed region; serial frames (implies --no-overlap)")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="debug: run the N-way screen-tile split on ONE GPU (LocalComm)")
    ap.add_argument("--pmc-calibration-copy", action="store_true", help="after the timed region, copy 512 MiB with the library's `pmc_calibration_copy` kernel: a known byte count for scripts/pmc_collect.sh")
    ap.add_argument("--motion-halo", type=int, default=16, help="rows of history exchanged beyond the stencil (>= max |screen motion| per frame)")
    return ap.parse_args()


def make_scene(name, tris):
    from kajiya_amd import scenes
    if name == "city":
        return scenes.procedural_city(target_tris=tris, seed=1234), dict(center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004), \
            f"procedural_city seed 1234 (~{tris} tris, battle.ron stand-in)"
    if name == "ruins":
        return scenes.procedural_ruins(target_tris=tris, seed=5678), dict(center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004), \
            f"procedural_ruins seed 5678 (~{tris} tris, Ruins stand-in)"
    if name == "pica":     # the reference's own production asset (assets/scenes/pica.ron): 76 k triangles, one instance
        return scenes.pica_diorama(), dict(center=(-0.4, 0.5, -0.6), radius=5.0, height=1.6, rate=0.01), "pica_pica mini diorama (assets/scenes/pica.ron geometry, material factors; no image maps)"
    return scenes.cornell_box(), dict(center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01), "cornell_box"


def frame_constants_list(W, H, n, cam_args, phase=0.0):
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = True
    out = []
    for i in range(n):
        out.append(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), phase=phase, **cam_args)))
        fs.retire_frame()
    return out


def cpu_baseline(desc, cam_args, cores, W=1920, H=1080, frames=20, scalar=True):
    """Oracle (CPU restatement, kind='port') on a bounded sample of the same scene/camera: the bench workload itself, fewer frames
    (~10-30 s of CPU work on 16+ cores: 20 frames at 1080p, 1 frame at 4K)."""
    from oracle import okj_py
    t_build = time.time()
    osc = okj_py.OracleScene(desc)
    t_build = time.time() - t_build
    op = okj_py.OraclePipeline(osc, W, H, use_ircache=True)
    okj_py.lib().okj_set_threads(cores)
    fcs = frame_constants_list(W, H, frames, cam_args)
    rays, t_gi = 0, 0.0
    for fc in fcs:
        op.render_inputs(fc)
        op.reprojection(fc)
        t0 = time.time()
        op.gi_frame(fc)
        t_gi += time.time() - t0
        a, b = op.ray_counts()
        c, d = op.ircache_ray_counts()
        rays += a + b + c + d
    res = {"value": round(rays / t_gi / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"oracle ircache + rtdgi (all passes), same scene+camera, {frames} frame(s) at {W}x{H} ({rays} rays in {t_gi:.2f} s; "
                     f"oracle BVH build {t_build:.1f} s not counted)",
           "gi_frame_ms_at_sample_res": round(1e3 * t_gi / frames, 2)}
    if not scalar:
        return res
    # one more frame on a single thread: the "CPU scalar reference" of SURVEY 8d (the multi-core figure above is the headline)
    okj_py.lib().okj_set_threads(1)
    fc = frame_constants_list(W, H, frames + 1, cam_args)[-1]
    okj_py.lib().okj_set_threads(cores)
    op.render_inputs(fc)
    op.reprojection(fc)
    okj_py.lib().okj_set_threads(1)
    t0 = time.time()
    op.gi_frame(fc)
    t_1 = time.time() - t0
    a, b = op.ray_counts()
    c, d = op.ircache_ray_counts()
    okj_py.lib().okj_set_threads(cores)
    res["scalar_1core"] = {"value": round((a + b + c + d) / t_1 / 1e6, 4), "unit": "Mrays/s", "gi_frame_ms": round(1e3 * t_1, 1), "sample": "1 more frame, 1 thread"}
    return res


def also_measurements():
    """The headline configs of BASELINE.json next to the primary 1080p line, each measured by a child process after
    the primary timed region has finished (own scene, own inputs resident in HBM before its own timed region):
      * 3840x2160 GI frame on the ~4 M-triangle ruins (north-star target: >= 30 fps = 33.3 ms) -- this script with other arguments;
      * 2560x1440 full lighting frame of configs[2] (ssgi, sun shadows + denoise, ircache + rtdgi, rtr, light_gbuffer, TAA) -- scripts/config3_bench.py."""
    import subprocess
    out = []
    env = dict(os.environ)
    for label, cmd in (("4K GI frame, ruins ~4M tris (configs[3] on one GPU / north-star target)",
                        [sys.executable, os.path.join(ROOT, "bench.py"), "--no-also", "--cpu-baseline-frames", "1", "--no-scalar-baseline", "--scene", "ruins", "--tris", "4000000",
                         "--width", "3840", "--height", "2160", "--steps", "36", "--warmup", "12", "--profile-frames", "6"]),
                       ("1440p full lighting frame, ruins ~4M tris (configs[2])",
                        [sys.executable, os.path.join(ROOT, "scripts", "config3_bench.py"), "--frames", "36", "--warmup", "12"]),
                       ("reference path tracer, 4K, ruins ~4M tris, N = 1 of configs[4]'s 8-way interleave (ms per sample per pixel pass)",
                        [sys.executable, os.path.join(ROOT, "scripts", "pt_bench.py"), "8"]),
                       ("screen-tile split of the 4K GI frame on ONE GPU: GPU work per rank at 4 and 8 virtual ranks (configs[3]; SURVEY 8e)",
                        [sys.executable, os.path.join(ROOT, "scripts", "split_virtual_bench.py"), "--ranks", "4,8", "--frames", "10", "--warmup", "4"])):
        t0 = time.time()
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
        except Exception as e:      # a failed side measurement must not take the primary line with it
            out.append({"what": label, "error": repr(e)[:200]})
            continue
        e = {"what": label, "wall_s": round(time.time() - t0, 1)}
        if "split" in j:      # scripts/split_virtual_bench.py: one entry per rank count, named split_virtual_N
            for sp in j["split"]:
                out.append(dict(sp, workload=j["workload"], one_gpu=j["one_gpu"], wall_s=e["wall_s"], method="HIP events around K serial frames of all N ranks on one stream; "
                                "per_rank_work_ms = (that - the orchestrator's own event pairs around every exchange) / N"))
            continue
        if "gi_frame_ms" in j:
            rf = j.get("roofline") or {}
            e.update({"gi_frame_ms": j["gi_frame_ms"], "fps": round(1000.0 / j["gi_frame_ms"], 1), "mrays_per_s": j["value"], "workload": j["config"]["workload"][:120],
                      "rays_per_frame": j["config"]["rays_per_frame"], "segment_ms": j.get("segment_ms"), "pass_ms": j.get("pass_ms"),
                      "deterministic_cache": j.get("deterministic_cache"),
                      "roofline": {k: rf.get(k) for k in ("kernel", "bound", "priced_against", "bound_measured", "limited_by", "avg_launch_ms", "algorithmic_bytes_per_launch", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac")},
                      "cpu_baseline": j.get("cpu_baseline")})
        elif "ms_per_spp" in j:
            e.update({k: j[k] for k in ("workload", "seconds", "ms_per_spp", "Mrays_per_s", "rays_per_path")})
            e["spp_64_ms"] = round(64 * j["ms_per_spp"], 1)       # configs[4]: 64 spp at 4K on one GPU; the 8-way interleave deals tiles round-robin (exact: tests/test_gpu_headline_sizes.py)
        else:
            e.update({"frame_ms": j["frame_ms"], "fps": j["fps"], "mrays_per_s": j["mrays_per_s"], "workload": j["workload"], "segment_ms": j["segment_ms"], "rays_per_frame": j["rays_per_frame"],
                      "overlap": j.get("overlap")})
        out.append(e)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KJ_BENCH_SHARE_GPU0"):   # debugging aid: several ranks on one GPU (only works if RCCL accepts duplicate devices)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if os.environ.get("KJ_BENCH_SHARE_GPU0"):
            dist.init_process_group("gloo")      # RCCL rejects duplicate devices; exchanges are staged through host memory
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from kajiya_amd import lib
    if os.environ.get("KJ_PRIO_MAIN"):      # A/B knob: the GI chain on a stream of that priority (-1 = high), the side streams stay at KJ_PRIO_* (default 0)
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["KJ_PRIO_MAIN"])))
        print("[bench] stream priority range", torch.cuda.Stream.priority_range(), "main", os.environ["KJ_PRIO_MAIN"], file=sys.stderr)

    W, H = args.width, args.height
    K, Wm = args.steps, args.warmup
    desc, cam_args, scene_label = make_scene(args.scene, args.tris)
    dev = lib.Device(local_rank)
    scene = lib.Scene(dev, desc, fast_build=bool(os.environ.get("KJ_BENCH_FAST_BUILD")))     # KJ_BENCH_FAST_BUILD=1: BLASes as device-built LBVHs (measurements)
    stats = scene.stats()
    gp = lib.GpuPipeline(dev, scene, W, H, device=f"cuda:{local_rank}", use_ircache=True)
    # ---- N > 1: screen-tile split of the SAME frame (strong scaling): one strip per rank, halo exchange over RCCL
    split = None
    transport = "none"
    nsplit = world if world > 1 else args.virtual_ranks
    if nsplit > 1:
        from kajiya_amd import multigpu
        if world > 1:
            split_pipes = {rank: gp}
            comm = multigpu.DistComm(dist, rank, world, stage_through_host=bool(os.environ.get("KJ_BENCH_SHARE_GPU0")))
        else:
            split_pipes = {0: gp}
            for r in range(1, nsplit):
                split_pipes[r] = lib.GpuPipeline(dev, scene, W, H, device=f"cuda:{local_rank}", use_ircache=True)
            comm = multigpu.LocalComm(nsplit)
        # Default for a real N-process job: the compiled orchestrator (csrc/split.cpp) with its own RCCL communicator -- one C-ABI call per frame, one
        # packed ncclSend / ncclRecv per peer and exchange point. It certifies itself before frame 0 (kj_split_self_test, verdicts combined over the
        # ranks); if the communicator cannot be made or the self-test fails on ANY rank, every rank falls back to the Python orchestrator over
        # torch.distributed (KJ_SPLIT_NATIVE=0 selects that one outright; virtual ranks use it unless KJ_SPLIT_NATIVE=1).
        want_native = os.environ.get("KJ_SPLIT_NATIVE", "1" if world > 1 and not os.environ.get("KJ_BENCH_SHARE_GPU0") else "0") == "1"
        transport = "virtual"
        if want_native:
            try:
                nccl = multigpu.NativeSplit.rccl_comm_from_torch(dist, rank, world, f"cuda:{local_rank}") if world > 1 else None
                split = multigpu.NativeSplit(nsplit, split_pipes, W, H, motion_halo=args.motion_halo, nccl_comm=nccl, own_comm=True)
                created = True
            except Exception as e:
                print(f"[bench] rank {rank}: compiled split orchestrator unavailable ({e})", file=sys.stderr, flush=True)
                split, created = None, False
            if world > 1:      # creation is not collective-safe by itself: agree before anybody enters the self-test's exchanges
                flag = torch.tensor([1 if created else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else f"cuda:{local_rank}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if not int(flag.item()):
                    if split is not None:
                        split.close()
                    split = None
            if split is not None:
                passed = split.self_test(dist if world > 1 else None)
                transport = "RCCL (compiled orchestrator)" if world > 1 else "virtual (compiled orchestrator)"
                if rank == 0:
                    print(f"[bench] {transport} {nsplit} ranks {'OK' if passed else 'FAILED'}: exchange self-test ({'passed' if passed else 'wrong rows delivered'})", file=sys.stderr, flush=True)
                if not passed:
                    split.close()      # explicitly, before the Python orchestrator switches the caches' update mode (not left to __del__)
                    split = None
            if split is None and rank == 0:
                print("[bench] falling back to the Python orchestrator over torch.distributed", file=sys.stderr, flush=True)
        if split is None:
            split = multigpu.SplitRtdgi(comm, split_pipes, W, H, motion_halo=args.motion_halo)
            passed = split.self_test()      # before frame 0: every kind of exchange of the frame schedule once, on scratch images, checked on the device (multigpu.py)
            transport = ("RCCL" if not os.environ.get("KJ_BENCH_SHARE_GPU0") else "gloo") if world > 1 else "virtual"
            if rank == 0:
                print(f"[bench] {transport} {nsplit} ranks {'OK' if passed else 'FAILED'}: exchange self-test ({'passed' if passed else 'wrong rows delivered'})", file=sys.stderr, flush=True)
            assert passed, "split transport self-test failed"
    single = split is None

    # ---- pre-generate the inputs of every frame (resident in HBM before the timed region; replicated on every rank)
    n_frames = Wm + K + args.profile_frames + 3 + 1
    fcs = frame_constants_list(W, H, n_frames, cam_args)
    inputs = []
    for fc in fcs:
        gp.render_inputs(fc)
        gp.reprojection()
        rp = lib.tensor_from_ptr(gp.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((gp.geometric_normal.clone(), gp.gbuffer.clone(), gp.depth.clone(), rp))
    torch.cuda.synchronize()
    max_motion_rows = None
    if not single:      # the split reads histories through the motion vectors: a frame moving further than the halo would be rendered from rows a rank does not hold
        from kajiya_amd import multigpu as _mg
        max_motion_rows = max(_mg.max_vertical_motion_rows(t[3], H) for t in inputs)
        if max_motion_rows > args.motion_halo and rank == 0:      # (the default camera moves 2.6 rows per frame at 1080p, 4.8 at 4K: reported, not fatal)
            print(f"[bench] WARNING: the frames move {max_motion_rows:.1f} rows per frame, more than --motion-halo {args.motion_halo}: the split's frames are not the one-GPU frames", file=sys.stderr, flush=True)
    counters = gp_counters = None
    use_ssgi = not args.no_ssgi

    def step(i):
        gn, gb, d, rp = inputs[i]
        dev.frame_begin(fcs[i])
        if single:
            gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
            gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
            if use_ssgi:
                gp.ssgi_frame()   # SsgiRenderer::render -> the SSAO guide of this frame (world_render_passes.rs:90-96)
            gp.gi_frame()
            gp.taa_frame()   # TaaRenderer::render on the GI output (the reference feeds it the lit image; same kernels, same bytes)
        else:
            for q in split.pipes.values():
                q.geometric_normal, q.gbuffer, q.depth = gn, gb, d
                q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
            if use_ssgi:
                split.ssgi_frame()      # the SSAO guide strip by strip (+ its two halo exchanges)
            split.gi_frame()
            split.taa_frame()

    step(0)  # allocates surfaces
    all_pipes = [gp] if single else list(split.pipes.values())
    # device counters are striped over 64 cache lines of 16 u64 (kj_vec.hpp: KJ_COUNTER_SLOTS / KJ_COUNTER_STRIDE); sum the slots
    gp_counters = [lib.tensor_from_ptr(*_counter_ptr(q, lib), torch.int64, (64, 16)) for q in all_pipes]
    irc_counters = [q.ircache_buffer("ray_counters", torch.int64).view(64, 16) for q in all_pipes]
    ray_log = torch.zeros((n_frames + 1, 6), dtype=torch.int64, device=f"cuda:{local_rank}")
    irc_log = torch.zeros((n_frames + 1, 2), dtype=torch.int64, device=f"cuda:{local_rank}")
    overlap = not args.no_overlap and not args.deterministic_cache
    if args.deterministic_cache and single:
        gp.ircache_set_deferred(True)
    serial_step = step
    if overlap:
        # frame pipelining (GpuPipeline.frame_pipelined / SplitRtdgi.frame_pipelined): frame i+1's ircache maintenance + rays run
        # on a second stream under frame i's screen-space tail. Same work per step, same dependencies; only the schedule differs.
        irc_frame = [0]

        def log_irc():   # on the ircache stream, right after its rays
            irc_log[irc_frame[0]].copy_(irc_counters[0][:, :2].sum(dim=0))
            for ic_ in irc_counters[1:]:
                irc_log[irc_frame[0]] += ic_[:, :2].sum(dim=0)
        (gp if single else split).on_ircache_traced = log_irc

        def step(i):  # noqa: F811
            gn, gb, d, rp = inputs[i]
            for q in all_pipes:
                q.geometric_normal, q.gbuffer, q.depth = gn, gb, d
                q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
            irc_frame[0] = i + 1
            # the SSAO guide reads frame i's constants, which the SIDE stream wrote: frame_pipelined orders it behind that write
            (gp if single else split).frame_pipelined(fcs[i + 1], run_ssgi=use_ssgi)
        torch.cuda.synchronize()
        irc_frame[0] = 1
        (gp if single else split).pipeline_begin(fcs[1])
    def log_rays(i):      # tiny device-side reductions on the launch stream: this frame's ray counters into row i of the logs
        ray_log[i].copy_(gp_counters[0][:, :6].sum(dim=0))
        for c_ in gp_counters[1:]:   # virtual ranks only
            ray_log[i] += c_[:, :6].sum(dim=0)
        if not overlap:
            irc_log[i].copy_(irc_counters[0][:, :2].sum(dim=0))
            for ic_ in irc_counters[1:]:
                irc_log[i] += ic_[:, :2].sum(dim=0)

    for i in range(1, Wm):
        step(i)
        log_rays(i)       # warm-up runs EXACTLY what a timed step runs (torch loads a reduction kernel's code object at its first use: 40 ms)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    step_host_s = []      # KJ_BENCH_STEP_TIMES=1: host time at which each step had been ISSUED (no syncs added), printed to stderr afterwards
    for i in range(Wm, Wm + K):
        step(i)
        log_rays(i)
        step_host_s.append(time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("KJ_BENCH_STEP_TIMES"):
        print("[bench] issue times of the timed steps (ms since t0):", " ".join(f"{1e3 * v:.2f}" for v in step_host_s), f"| done {1e3 * elapsed:.2f}", file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rays_closest = int(ray_log[Wm:Wm + K, 0].sum().item())
    rays_any = int(ray_log[Wm:Wm + K, 1].sum().item())
    irc_rays = int(irc_log[Wm:Wm + K].sum().item())
    local_closest, local_any = rays_closest / max(1, nsplit if world == 1 else 1), rays_any / max(1, nsplit if world == 1 else 1)   # one rank's strip
    if world > 1:
        t = torch.tensor([rays_closest, rays_any, irc_rays], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rays_closest, rays_any, irc_rays = (int(v) for v in t.tolist())
    if nsplit > 1:
        # every rank keeps its own replica of the irradiance cache (fed by its strip's rays only); replicas overlap, so
        # their rays are counted as ONE cache's worth (mean over ranks) -- conservative for `value`.
        irc_rays //= nsplit
    total_rays = total_rays_all = rays_closest + rays_any + irc_rays

    seg, pass_ms, roofline, roofline_all = None, None, None, None
    if overlap:
        torch.cuda.synchronize()
        (gp if single else split).on_ircache_traced = None
        step = serial_step
    if True:
        # ---- per-pass GPU timestamps (HIP events on the launch stream), after the timed region. In the screen-tile split these
        # are the passes of THIS rank's strip (rank 0 reports).
        gp.set_profiling(True, False)
        pass_ms = [0.0] * 11
        trace_ms_list = []
        base = Wm + K
        for i in range(base, base + args.profile_frames):
            step(i)
            torch.cuda.synchronize()
            t = gp.pass_times_ms()
            pass_ms = [a + b for a, b in zip(pass_ms, t)]
            trace_ms_list.append(t[3])
        pass_ms = [p / max(1, args.profile_frames) for p in pass_ms]
        # `rtdgi trace` = k_rtdgi_trace_fused's launches: on a validation frame (one in three) the two ray passes are ONE launch since round 6
        # (k_rtdgi_validate_and_trace, timed under `rtdgi validate`; `rtdgi trace` reads 0 for such a frame): the mean is over the frames that have the launch
        trace_launches = [t_ for t_ in trace_ms_list if t_ > 0.0]
        if trace_launches and len(trace_launches) < len(trace_ms_list):
            pass_ms[3] = sum(trace_launches) / len(trace_launches)
        gp.set_profiling(False, False)
        # ---- segment timers (torch events on the launch stream): ircache / rtdgi / taa
        seg = {"ssgi": 0.0, "ircache": 0.0, "rtdgi": 0.0, "taa": 0.0}
        nseg = 6 if single else 0
        for i in range(base, base + nseg):   # replays inputs of already-used frames: timing only
            gn, gb, d, rp = inputs[i]
            gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
            gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
            dev.frame_begin(fcs[i])
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            s_ = lib._stream_ptr()
            evs = torch.cuda.Event(enable_timing=True); evs.record()
            if use_ssgi:
                gp.ssgi_frame()
            ev[0].record()
            lib.check(gp.L.kj_ircache_prepare(gp.ircache, s_))
            lib.check(gp.L.kj_ircache_trace_irradiance(gp.ircache, gp.scene.h, gp.sky16.data_ptr(), 16, s_))
            ev[1].record()
            lib.check(gp.L.kj_rtdgi_reproject(gp.rtdgi, gp.reprojection_map_ptr, W, H, s_))
            ev[2].record()
            lib.check(gp.L.kj_ircache_sum_up_irradiance_for_sampling(gp.ircache, s_))
            ev[3].record()
            p_ = gp.params()
            lib.check(gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(p_), C.byref(gp.out), s_))
            ev[4].record()
            gp.taa_frame()
            ev5 = torch.cuda.Event(enable_timing=True); ev5.record()
            torch.cuda.synchronize()
            seg["ssgi"] += evs.elapsed_time(ev[0])
            seg["ircache"] += ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])
            seg["rtdgi"] += ev[1].elapsed_time(ev[2]) + ev[3].elapsed_time(ev[4])
            seg["taa"] += ev[4].elapsed_time(ev5)
        seg = {k: round(v / nseg, 4) for k, v in seg.items()} if nseg else None
        # ---- instrumented traversal counters (3 frames: one validation + two tracing frames)
        # The frame is issued in two parts so that the TRACE kernel's own rays can be told from the validate kernel's (the per-frame ray
        # counters cover both): everything up to the validate pass, read the counters, the rest, read them again.
        gp.set_profiling(False, True)
        trav, trace_only, n_trace_only = None, None, 0
        Pm = lib.KJ_RTDGI_PASS
        head_mask = Pm["EXTRACT_HALF"] | Pm["VALIDATE"]
        for i in range(base + args.profile_frames, base + args.profile_frames + 3):
            if single:
                gn, gb, d, rp = inputs[i]
                dev.frame_begin(fcs[i])
                gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
                gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
                if use_ssgi:
                    gp.ssgi_frame()
                gp.gi_frame(head_mask)
                torch.cuda.synchronize()
                a = gp.traversal_counts()
                prm = gp.params((Pm["ALL"] & ~head_mask) | (1 << 31))
                lib.check(gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(prm), C.byref(gp.out), None))
                gp.taa_frame()
                torch.cuda.synchronize()
                c = gp.traversal_counts()
                t_only = {k: c[k] - a[k] for k in c}
                if int(fcs[i].frame_index) % 3 != 0 or not trace_launches or len(trace_launches) == len(trace_ms_list):      # (the trace kernel's launches the timer above averages over)
                    trace_only = t_only if trace_only is None else {k: trace_only[k] + t_only[k] for k in c}
                    n_trace_only += 1
            else:
                step(i)
                torch.cuda.synchronize()
                c = gp.traversal_counts()
            trav = c if trav is None else {k: trav[k] + c[k] for k in c}
        gp.set_profiling(False, False)

        hw, hh = (W + 1) // 2, (H + 1) // 2
        # dominant kernel: pick by measured time
        dom = max(range(11), key=lambda k: pass_ms[k])
        dom_name = lib.GpuPipeline.PASS_NAMES[dom]
        # algorithmic bytes of `rtdgi trace` per launch: 38 B/half-res px of surface I/O
        #   + per ray: nodes visited x 64 B + triangles tested x 48 B (instrumented) + 232 B hit shading per closest hit
        nodes_per_closest = trav["closest_nodes"] / max(1, trav["closest_rays"])
        tris_per_closest = trav["closest_tris"] / max(1, trav["closest_rays"])
        nodes_per_any = trav["any_nodes"] / max(1, trav["any_rays"])
        tris_per_any = trav["any_tris"] / max(1, trav["any_rays"])
        # rays issued by the trace kernel per frame ~ measured split: trace issues (hw*hh non-sky) closest + shadow rays;
        # use the per-frame average of the timed region minus the validate kernel's share (1/3 of frames run validate).
        closest_per_frame = local_closest / K
        any_per_frame = local_any / K
        strip_frac = 1.0 / max(1, nsplit)
        if trace_only is not None:      # the trace kernel's own rays and its own nodes / triangles, counted (3 instrumented frames)
            ray_bytes = (trace_only["closest_nodes"] + trace_only["any_nodes"]) * 64 + (trace_only["closest_tris"] + trace_only["any_tris"]) * 48 + trace_only["closest_rays"] * 232
            trace_bytes = hw * hh * 38 * strip_frac + ray_bytes / float(max(1, n_trace_only))
            trace_rays_note = {"trace_kernel_closest_rays_per_launch": round(trace_only["closest_rays"] / float(max(1, n_trace_only)), 1), "trace_kernel_any_rays_per_launch": round(trace_only["any_rays"] / float(max(1, n_trace_only)), 1),
                               "trace_kernel_launches_counted": n_trace_only}
        else:                           # split runs: the validate kernel traces about a third as many rays per frame as the trace kernel
            trace_share = 1.0 / (1.0 + 1.0 / 3.0)
            bytes_per_closest = nodes_per_closest * 64 + tris_per_closest * 48 + 232
            bytes_per_any = nodes_per_any * 64 + tris_per_any * 48
            trace_bytes = hw * hh * 38 * strip_frac + trace_share * (closest_per_frame * bytes_per_closest + any_per_frame * bytes_per_any)
            trace_rays_note = {"trace_kernel_ray_share": "estimated 0.75 of the frame's rays (split run)"}
        trace_ms = pass_ms[3]
        achieved = trace_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        # ---- per-kernel rooflines. `achieved` = ALGORITHMIC bytes (every input texel read once + every output written
        # once, x the units of one launch) / the kernel's average launch time measured above with HIP events on the launch stream.
        # `traffic` = HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_kernels.json, produced by
        # scripts/pmc_collect.sh on this workload; rocprofv3 cannot run inside this process), FETCH_SIZE doubled per the microarch
        # guide's gfx950 correction, and `hbm_frac` = traffic / launch time / peak: what the memory system actually moved. For the ray
        # kernel the two differ by design: its algorithmic bytes are BVH nodes and triangles that live in L2 / Infinity Cache.
        pmc, pmc_file = {}, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_kernels*.json"))):      # one file per profiled workload (scripts/pmc_collect.sh)
            try:
                pm = json.load(open(path))
                wl = pm["workload"]
                if (wl["scene"], wl["tris"], wl["width"], wl["height"]) == (args.scene, args.tris, W, H):
                    pmc, pmc_file = pm["kernels"], os.path.relpath(path, ROOT)
            except Exception:
                continue
        n_h2, n_f = hw * hh * strip_frac, W * H * strip_frac
        # pass -> (kernel name as rocprofv3 prints it, units per launch, algorithmic bytes per unit [SURVEY 8d table])
        # validity integrate + restir temporal are ONE launch by default since round 4 (k_validity_integrate_restir_temporal; KJ_RTDGI_FUSE_VT=0 splits them):
        # its time is reported under "restir temporal", "validity integrate" reads 0
        vi_fused = pass_ms[lib.GpuPipeline.PASS_NAMES.index("validity integrate")] == 0.0
        tile_order = 2 if (hw + 15) // 16 >= 96 else 3      # rtdgi_resample.hip: resample_tile_order (column bands from 96 tiles across, 4 x 4 super-tiles below)
        table = [("rtdgi reproject", "k_fullres_reproject", n_f, 24), ("extract half", "k_extract_half<0>", n_h2, 30)] + \
                ([("restir temporal", "k_validity_integrate_restir_temporal", n_h2, 25 + 168)] if vi_fused else
                 [("validity integrate", "k_validity_integrate", n_h2, 25), ("restir temporal", "k_restir_temporal", n_h2, 168)]) + \
                [("restir spatial 0", f"k_restir_spatial<32, 8, 16, 16, false, {tile_order}, false>", n_h2, 41),
                 ("restir spatial 1", f"k_restir_spatial<16, 5, 16, 16, false, {tile_order}, true>", n_h2, 41), ("restir resolve", f"k_restir_resolve<{2 if (W + 15) // 16 >= 96 else 3}>", n_f, 43),
                 ("rtdgi temporal", "k_temporal_filter", n_f, 49), ("rtdgi spatial", "k_spatial_filter", n_f, 25)]

        def entry(kernel, ms, algo_bytes, note=None):
            e = {"kernel": kernel, "bound": "hbm", "priced_against": "hbm", "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes),
                 "achieved": round(algo_bytes / (ms * 1e-3) / 1e9, 2) if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            e["frac"] = round(e["achieved"] / HBM_PEAK_GBS, 5)
            k = pmc.get(kernel) or pmc.get(kernel.split("<")[0])
            if k and "fetch_bytes_corrected" in k and "write_bytes" in k:
                e["traffic"] = int(k["fetch_bytes_corrected"] + k["write_bytes"])
                e["hbm_frac"] = round(e["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None
            else:
                e["traffic"], e["hbm_frac"] = None, None
            for src, dst in (("VALUBusy", "valu_busy_pct"), ("VALUUtilization", "valu_lane_utilization_pct"), ("MemUnitStalled", "mem_unit_stalled_pct")):
                if k and src in k:
                    e[dst] = round(k[src], 1)
            # `bound` is the bench contract's field: which of its two rooflines ("hbm" | "mfma") `achieved` / `peak` / `frac` are priced against -- HBM here, this path
            # has no dense contraction (same value in `priced_against`, which says so by name). What the counters of the same kernel show it limited BY is
            # `bound_measured` ("hbm" | "valu" | "memory unit" | "latency") and, spelled out, `limited_by`.
            vb, hf, ms_ = e.get("valu_busy_pct"), e.get("hbm_frac"), e.get("mem_unit_stalled_pct")
            # `bound_measured`: the same verdict as one word ("hbm" | "valu" | "memory unit" | "latency"), next to the contract's `bound`
            if vb is None or hf is None:
                e["limited_by"], e["bound_measured"] = None, None
            elif hf >= 0.6:
                e["limited_by"], e["bound_measured"] = f"hbm bandwidth (traffic at {hf:.2f} of peak)", "hbm"
            elif vb >= 75.0:
                e["limited_by"], e["bound_measured"] = f"valu issue (VALUBusy {vb:.0f} %)", "valu"
            elif ms_ is not None and ms_ >= 20.0:
                e["limited_by"], e["bound_measured"] = f"memory unit (MemUnitStalled {ms_:.0f} %)", "memory unit"
            else:
                e["limited_by"], e["bound_measured"] = f"latency: neither the VALUs ({vb:.0f} % busy) nor HBM ({hf:.2f} of peak) saturated -- dependent loads / occupancy", "latency"
            if note:
                e["note"] = note
            return e
        ray_kernel = "k_rtdgi_trace_fused<false, false>"
        roofline = entry(ray_kernel, trace_ms, trace_bytes)
        roofline.update({"traffic_source": f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)" if roofline["traffic"] else None,
                         "nodes_per_closest_ray": round(nodes_per_closest, 2), "tris_per_closest_ray": round(tris_per_closest, 2),
                         "nodes_per_any_ray": round(nodes_per_any, 2), "tris_per_any_ray": round(tris_per_any, 2), "dominant_by_time": dom_name})
        roofline.update(trace_rays_note)
        roofline_all = [roofline] + [entry(kern, pass_ms[lib.GpuPipeline.PASS_NAMES.index(pname)], units * bpu) for pname, kern, units, bpu in table]
        if seg:
            roofline_all.append(entry("taa (7 passes, 5 launches)", seg["taa"], W * H * 224, "segment: sum of the TAA launches (the two input filters and the two probability filters share a launch each)"))
        if args.pmc_calibration_copy and single:
            nbytes = 512 << 20
            a_ = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{local_rank}"); b_ = torch.empty_like(a_)
            lib.check(gp.L.kj_debug_calibration_copy(b_.data_ptr(), a_.data_ptr(), nbytes, lib._stream_ptr()))
            torch.cuda.synchronize()

    # ---- the cache's two modes side by side: `value` above times the reference's racy cache (pipelined frames); the 1e-3 cache parity, smoke's second leg
    # and every screen-tile split run the DETERMINISTIC mode (lookups record, one reduction + merge per frame). Same process, same inputs (replayed: timing only), serial frames.
    det_leg = None
    if single and world == 1 and not args.deterministic_cache:
        def serial_ms(nf):
            torch.cuda.synchronize(); t_ = time.perf_counter()
            for k_ in range(nf):
                serial_step(Wm + (k_ % max(1, K)))
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t_) / nf
        nf = min(24, K)
        serial_ms(6)
        racy_serial = serial_ms(nf)
        gp.ircache_set_deferred(True)
        serial_ms(6)
        det_serial = serial_ms(nf)
        gp.ircache_set_deferred(False)
        det_leg = {"frames": nf, "serial_racy_ms": round(racy_serial, 4), "serial_deterministic_ms": round(det_serial, 4),
                   "note": "serial frames on one stream, this run's own inputs replayed; deterministic = kj_ircache_set_deferred_updates(1): lookups record 32-byte requests, reduced into a fixed-size summary and "
                           "merged once per frame (no sort, no host read-back since round 6). An N > 1 line (screen-tile split) runs this mode on every rank: compare it with serial_deterministic_ms, not with `value`"}
    ms_per_step = 1e3 * elapsed / K
    out = {
        "metric": "gi_mrays_per_s", "value": round(total_rays_all / elapsed / 1e6, 3), "unit": "Mrays/s",
        "gi_frame_ms": round(ms_per_step, 4), "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene_label}, {W}x{H}, rtdgi: reproject+validate+trace+validity+temporal ReSTIR+2x spatial ReSTIR+"
                               "resolve+temporal+spatial denoise, irradiance cache (scroll/age/compact, accessibility+validate+trace rays, SH sum), TAA (7 passes) on the GI output"
                               + (", SSAO guide (ssgi, 4 passes)" if use_ssgi else ", constant SSAO guide"),
                   "triangles": stats["triangles"], "bvh_nodes": stats["nodes"], "bvh_bytes": stats["bvh_bytes"],
                   "rays_per_frame": round(total_rays / K, 1), "ircache_rays_per_frame": round(irc_rays / K, 1), "parallelism": ("single GPU, 3 HIP streams: next frame's ircache rays (side stream) and this frame's spatial filter + TAA (third stream) overlap the main stream's ray passes" if overlap else "single GPU, serial frames") if nsplit <= 1 else f"{nsplit}-way screen-tile split (16-row-aligned strips; 6 batched halo exchanges per frame incl. the temporal2 all-gather, over "
                                  + (("gloo, host-staged (debug)" if os.environ.get("KJ_BENCH_SHARE_GPU0") and "compiled" not in transport else "RCCL P2P") if world > 1 else "virtual ranks on one GPU") + f"; orchestrator: {'compiled (kj_split_*)' if 'compiled' in transport else 'python (multigpu.SplitRtdgi)'}" + f"; motion halo {args.motion_halo} rows; irradiance cache replicated per rank"
                                  + ("; next frame's ircache work overlapped on a second stream)" if overlap else ")")},
        "ircache_mode": "deterministic (deferred, canonically ordered updates)" if (args.deterministic_cache or nsplit > 1) else "racy (the reference's atomics)",
        "deterministic_cache": det_leg,
        "segment_ms": seg,
        "pass_ms": {n: round(v, 4) for n, v in zip(lib.GpuPipeline.PASS_NAMES, pass_ms)} if pass_ms else None,
        "roofline": roofline,
        "roofline_all": roofline_all,
    }
    if max_motion_rows is not None:
        out["config"]["max_vertical_motion_rows"] = round(max_motion_rows, 2)      # measured on the run's own reprojection maps before the timed region
        out["config"]["motion_within_halo"] = bool(max_motion_rows <= args.motion_halo)
    out["comm_ranks"] = (dist.get_world_size() if world > 1 else 1)     # what the communicator itself reports: an N > 1 run certifies its rank count
    if rank == 0 and world == 1 and not args.no_also and nsplit <= 1:
        inputs.clear()              # the primary workload's frames: free their HBM before the children build theirs
        torch.cuda.empty_cache()
        out["also"] = also_measurements()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:  # respect a cgroup CPU quota: oversubscribed OpenMP threads spin and distort the baseline
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cores = max(1, min(cores, int(int(q) / int(per))))
        except Exception:
            pass
        cores = min(cores, 64)
        out["cpu_baseline"] = cpu_baseline(desc, cam_args, cores, W, H, args.cpu_baseline_frames, not args.no_scalar_baseline)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _counter_ptr(gp, lib):
    ptr, n = C.c_void_p(), C.c_uint64()
    lib.check(gp.L.kj_rtdgi_surface(gp.rtdgi, b"ray_counters", C.byref(ptr), C.byref(n)))
    return ptr.value, n.value


if __name__ == "__main__":
    main()
