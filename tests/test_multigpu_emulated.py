"""The N>1 path end to end without a GPU: `world` PROCESSES, one rank each, torch.distributed over gloo on 127.0.0.1, every rank running the
product's kernels on the CPU stand-in for HIP (tests/hip_emu, fiber mode) for its strip of the screen and exchanging halos through
kajiya_amd.multigpu.DistComm — the same orchestrator code `bench.py --gpus N` runs over RCCL. After every frame each rank's gathered GI
and TAA images must equal the unsplit frame bit for bit (the irradiance cache unbound, as in tests/test_gpu_multigpu.py).

What the other multi-GPU tests leave open and this one closes: tests/test_multigpu_gloo.py moves synthetic bytes through DistComm,
tests/test_gpu_multigpu.py runs the real kernels but with all ranks in one process (LocalComm). No 8-GPU node is available to the
build, so this is the only place where separate processes, a real process group and the real kernels meet."""
import multiprocessing as mp
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_rccl_stub():
    """tests/rccl_stub/rccl_stub.cpp -> tests/_build/rccl_stub/librccl_stub.so: the socket-backed stand-in for RCCL the compiled transport is pointed at (KJ_RCCL_LIB)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "rccl_stub", "rccl_stub.cpp")
    out = os.path.join(ROOT, "tests", "_build", "rccl_stub", "librccl_stub.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    return out


def _worker(rank, world, port, W, H, frames, packed, q, with_ircache=False, with_ssgi=False, native=False):
    try:
        os.environ["KJ_HIP_EMU"] = "fast"
        os.environ.setdefault("HIP_EMU_WORKERS", "4")
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
        import build_emu, cpu_as_cuda
        cpu_as_cuda.install(build_emu.build())
        import torch
        import torch.distributed as dist
        from kajiya_amd import lib, multigpu
        import test_gpu_parity as T
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = lib.Device(0)
        scene = lib.Scene(dev, T._scenes()["city20k"])
        ref = lib.GpuPipeline(dev, scene, W, H, use_ircache=with_ircache)
        pipe = lib.GpuPipeline(dev, scene, W, H, use_ircache=with_ircache)
        if with_ircache:
            ref.ircache_set_deferred(True)      # the single-GPU frame in the same deterministic mode the split puts the replicas in
        if native:
            # the COMPILED orchestrator with its own communicator (csrc/split.cpp's RCCL branch), the communicator being the socket stand-in:
            # bootstrap exactly as bench.py does (rank 0 draws the id, torch.distributed broadcasts it), self-test, then frames
            os.environ["KJ_RCCL_LIB"] = build_rccl_stub()
            comm = multigpu.NativeSplit.rccl_comm_from_torch(dist, rank, world, "cuda:0")
            split = multigpu.NativeSplit(world, {rank: pipe}, W, H, motion_halo=8, nccl_comm=comm)
            assert split.self_test(dist) is True
            split.strips = [split.strip(r) for r in range(world)]
        else:
            split = multigpu.SplitRtdgi(multigpu.DistComm(dist, rank, world, packed=packed), {rank: pipe}, W, H, motion_halo=8)
        assert split.consistent_ircache == with_ircache
        worst = 0
        from kajiya_amd import frame as kframe
        fs = kframe.FrameState((W, H))
        fs.ircache_enabled = with_ircache
        fcs = []
        for i in range(frames):
            fcs.append(fs.prepare_frame_constants(kframe.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)))
            fs.retire_frame()
        for fi, fc in enumerate(fcs):
            if with_ssgi:      # the SSAO guide: whole-frame on the reference, strip by strip (+ its two halo exchanges over gloo) in the split
                ref.render_inputs(fc); ref.reprojection(); ref.ssgi_frame(); ref.gi_frame()
            else:
                ref.frame(fc)
            pipe.render_inputs(fc)
            pipe.reprojection()
            if with_ssgi:
                split.ssgi_frame()
            split.gi_frame()
            split.taa_frame()
            ref.taa_frame()
            if with_ssgi:      # (the cases with the guide also carry the sun shadows: mask + denoiser strip by strip, two more exchanges)
                dn = ref.shadow_denoise(ref.sun_shadow_mask()).view(torch.int16)
                a0, b0 = split.strips[rank]
                worst = max(worst, int((dn[a0:b0] != split.shadow_frame()[rank].view(torch.int16)[a0:b0]).sum()))
            split.gather_output("spatial_filtered_tex")
            split.gather_output(f"TAA/taa:{fi % 2}")
            a, b = ref.surface("spatial_filtered_tex", torch.int16, (H, W, 4)), pipe.surface("spatial_filtered_tex", torch.int16, (H, W, 4))
            ta, tb = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4)), pipe.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
            worst = max(worst, int((a != b).any(dim=-1).sum()), int((ta != tb).any(dim=-1).sum()))
            if with_ircache:      # every replica of the cache equals the single-GPU cache, buffer by buffer
                for name in ("meta", "grid_meta", "entry_cell", "irradiance", "life", "pool", "reposition_proposal", "reposition_proposal_count"):
                    worst = max(worst, int((ref.ircache_buffer(name, torch.uint8) != pipe.ircache_buffer(name, torch.uint8)).sum()))
        own = split.strips[rank]
        q.put((rank, worst, pipe.ray_counts(), ref.ray_counts(), own))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of a hang
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ as the host compiler")
@pytest.mark.parametrize("world,packed,with_ircache,with_ssgi,native", [(2, False, False, False, False), (3, True, False, True, False), (2, False, True, False, False),
                                                                        (2, False, True, True, True), (3, False, True, False, True)])
def test_strip_split_over_gloo_processes_with_the_real_kernels(world, packed, with_ircache, with_ssgi, native):
    """with_ircache: the irradiance cache bound on every rank; the strips' recorded cache updates travel through DistComm.all_gather_rows
    and every replica must stay bit-identical to the single-GPU cache.
    native: the compiled orchestrator (kj_split_*) with a communicator of its own instead of SplitRtdgi + DistComm -- the code path `bench.py --gpus N`
    takes by default -- its ncclSend / ncclRecv / ncclAllGather calls served by tests/rccl_stub (sockets between the processes). Packing per peer, the
    message order both ends assume, the all-gather of the record counts and lists, the self-test before frame 0: everything but RCCL itself."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
    env_before = os.environ.get("KJ_HIP_EMU")
    os.environ["KJ_HIP_EMU"] = "fast"
    try:
        import build_emu
        build_emu.build()                      # once, before the ranks race for it
        if native:
            build_rccl_stub()
    finally:
        if env_before is None:
            os.environ.pop("KJ_HIP_EMU", None)
        else:
            os.environ["KJ_HIP_EMU"] = env_before
    W, H, frames = 128, 96 if world == 2 else 144, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, frames, packed, q, with_ircache, with_ssgi, native)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert not [r for r in results if r[1] == "error"], "\n".join(str(r[2]) for r in results if r[1] == "error")
    results.sort()
    assert [r[1] for r in results] == [0] * world, results                     # every rank holds the unsplit frame, bit for bit
    total = results[0][3]
    assert (sum(r[2][0] for r in results), sum(r[2][1] for r in results)) == tuple(total), results    # the strips' rays add up to the unsplit frame's
    assert results[0][4][0] == 0 and results[-1][4][1] == H


def _self_test_worker(rank, world, port, W, H, corrupt, q):
    try:
        os.environ["KJ_HIP_EMU"] = "fast"
        os.environ.setdefault("HIP_EMU_WORKERS", "2")
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        if corrupt is not None:
            os.environ["KJ_RCCL_STUB_CORRUPT"] = str(corrupt)
            os.environ["KJ_RCCL_STUB_TIMEOUT_MS"] = "5000"      # a damaged length field leaves a rank waiting for bytes nobody sends: the stand-in gives up, RCCL would not
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
        import build_emu, cpu_as_cuda
        cpu_as_cuda.install(build_emu.build())
        import torch.distributed as dist
        from kajiya_amd import lib, multigpu
        import test_gpu_parity as T
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = lib.Device(0)
        pipe = lib.GpuPipeline(dev, lib.Scene(dev, T._scenes()["cornell"]), W, H, use_ircache=True)
        os.environ["KJ_RCCL_LIB"] = build_rccl_stub()
        split = multigpu.NativeSplit(world, {rank: pipe}, W, H, motion_halo=8, nccl_comm=multigpu.NativeSplit.rccl_comm_from_torch(dist, rank, world, "cuda:0"))
        q.put((rank, split.self_test(dist)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ as the host compiler")
@pytest.mark.parametrize("world,corrupt", [(3, None), (2, 1), (3, 0)])
def test_compiled_transport_self_test_notices_a_damaged_message(world, corrupt):
    """kj_split_self_test over the socket stand-in for RCCL: passes on a sound transport; when ONE rank's received messages arrive with a flipped
    byte, every rank reports failure (the verdicts are combined), which is what makes bench.py fall back to the Python orchestrator."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
    env_before = os.environ.get("KJ_HIP_EMU")
    os.environ["KJ_HIP_EMU"] = "fast"
    try:
        import build_emu
        build_emu.build()
        build_rccl_stub()
    finally:
        if env_before is None:
            os.environ.pop("KJ_HIP_EMU", None)
        else:
            os.environ["KJ_HIP_EMU"] = env_before
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_self_test_worker, args=(r, world, port, 128, 160, corrupt, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert sorted(got) == list(range(world))
    assert all(v is (corrupt is None) for v in got.values()), got


def _rtr_worker(rank, world, port, W, H, frames, q, native):
    try:
        os.environ["KJ_HIP_EMU"] = "fast"
        os.environ.setdefault("HIP_EMU_WORKERS", "4")
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
        import build_emu, cpu_as_cuda
        cpu_as_cuda.install(build_emu.build())
        import torch
        import torch.distributed as dist
        from kajiya_amd import lib, multigpu, scenes as S, frame as kframe
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = lib.Device(0)
        scene = lib.Scene(dev, S.glossy_test_scene())
        ref = lib.GpuPipeline(dev, scene, W, H, use_ircache=True)
        pipe = lib.GpuPipeline(dev, scene, W, H, use_ircache=True)
        ref.ircache_set_deferred(True)
        ref.ircache_set_rtr_requests(True)
        if native:
            os.environ["KJ_RCCL_LIB"] = build_rccl_stub()
            comm = multigpu.NativeSplit.rccl_comm_from_torch(dist, rank, world, "cuda:0")
            split = multigpu.NativeSplit(world, {rank: pipe}, W, H, motion_halo=8, nccl_comm=comm)
            assert split.self_test(dist) is True
            own = split.strip(rank)
        else:
            split = multigpu.SplitRtdgi(multigpu.DistComm(dist, rank, world), {rank: pipe}, W, H, motion_halo=8)
            own = split.strips[rank]
        split.enable_rtr()
        fs = kframe.FrameState((W, H))
        fs.ircache_enabled = True
        worst = 0
        a, b = own
        hh = (H + 1) // 2
        ha, hb = a // 2, (hh if b == H else b // 2)
        for fi in range(frames):
            fc = fs.prepare_frame_constants(kframe.orbit_camera(fi, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.008))
            fs.retire_frame()
            ref.render_inputs(fc); ref.reprojection()
            ref.gi_frame(defer_replay=True)
            ref.rtr_frame()
            ref.ircache_replay_own_requests()
            pipe.render_inputs(fc); pipe.reprojection()
            split.gi_frame()
            split.rtr_frame()
            x, y = ref.rtr_surface("resolved_tex", torch.int32, (H, W)), pipe.rtr_surface("resolved_tex", torch.int32, (H, W))
            worst = max(worst, int((x[a:b] != y[a:b]).sum()))
            for n, rows, lo, hi in ((f"rtr.temporal:{fi % 2}", H, a, b), (f"rtr.reservoir:{fi % 2}", hh, ha, hb), (f"rtr.irradiance:{fi % 2}", hh, ha, hb)):
                x, y = ref.rtr_surface(n, torch.uint8, (rows, -1)), pipe.rtr_surface(n, torch.uint8, (rows, -1))
                worst = max(worst, int((x[lo:hi] != y[lo:hi]).sum()))
            for name in ("meta", "grid_meta", "entry_cell", "irradiance", "life", "pool", "reposition_proposal", "reposition_proposal_count"):
                worst = max(worst, int((ref.ircache_buffer(name, torch.uint8) != pipe.ircache_buffer(name, torch.uint8)).sum()))
        q.put((rank, worst, pipe.rtr_ray_counts(), ref.rtr_ray_counts(), own))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ as the host compiler")
@pytest.mark.parametrize("world,native", [(2, False), (3, True)])
def test_reflections_split_over_processes_with_the_real_kernels(world, native):
    """SplitRtdgi.rtr_frame over DistComm / kj_split_rtr_frame over the socket stand-in for RCCL, one process per rank: the three all-gathers, the history
    halos with their pinned row 0, rtr's cache records in the merged replay. On its own rows every rank holds the unsplit frame's resolved image and
    reservoir state, every replica of the cache equals the single-GPU cache, and the strips' reflection rays add up to the unsplit frame's."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hip_emu"))
    env_before = os.environ.get("KJ_HIP_EMU")
    os.environ["KJ_HIP_EMU"] = "fast"
    try:
        import build_emu
        build_emu.build()
        if native:
            build_rccl_stub()
    finally:
        if env_before is None:
            os.environ.pop("KJ_HIP_EMU", None)
        else:
            os.environ["KJ_HIP_EMU"] = env_before
    W, H, frames = 128, 96 if world == 2 else 144, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rtr_worker, args=(r, world, port, W, H, frames, q, native)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert not [r for r in results if r[1] == "error"], "\n".join(str(r[2]) for r in results if r[1] == "error")
    results.sort()
    assert [r[1] for r in results] == [0] * world, results
    total = results[0][3]
    assert (sum(r[2][0] for r in results), sum(r[2][1] for r in results)) == tuple(total) and total[0] > 0, results      # (ray counters of the LAST frame)
