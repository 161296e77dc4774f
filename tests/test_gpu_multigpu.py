"""Screen-tile split on ONE GPU: N virtual ranks (LocalComm) must reproduce the unsplit frame
bit-for-bit when the irradiance cache is unbound (every exchanged halo is exactly what the consumer reaches)."""
import numpy as np
import pytest

import test_gpu_parity as T


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H,with_ssgi", [(2, 256, 160, False), (3, 320, 208, False), (2, 256, 160, True), (3, 320, 208, True), (2, 384, 800, True)])
def test_strip_split_is_bit_exact(gpu, device, n_ranks, W, H, with_ssgi):
    """`with_ssgi`: the SSAO guide computed strip by strip as well (SplitRtdgi.ssgi_frame: kj_ssgi_render_rows + the halo exchanges of its history
    and of the finished guide) instead of the constant guide; 384x800 on two ranks: strips of 400 rows, taller than every halo involved."""
    import torch
    from kajiya_amd import multigpu
    desc = T._scenes()["city20k"]
    scene = gpu.Scene(device, desc)
    ref = gpu.GpuPipeline(device, scene, W, H)
    pipes = {r: gpu.GpuPipeline(device, scene, W, H) for r in range(n_ranks)}
    split = multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=8)
    assert split.strips[0][0] == 0 and split.strips[-1][1] == H
    for fi, fc in enumerate(T._frame_constants(W, H, 5 if H > 400 else 7, "city")):
        if with_ssgi:
            ref.dev.frame_begin(fc); ref.render_inputs(fc); ref.reprojection(); ref.ssgi_frame(); ref.gi_frame()
        else:
            ref.frame(fc)
        for r in range(n_ranks):
            pipes[r].render_inputs(fc)
            pipes[r].reprojection()
        if with_ssgi:
            split.ssgi_frame()
        split.gi_frame()
        split.taa_frame()
        ref.taa_frame()
        split.gather_output("spatial_filtered_tex")
        split.gather_output(f"TAA/taa:{fi % 2}")
        torch.cuda.synchronize()
        if with_ssgi:      # the guide itself, on every rank's own rows
            ga = ref.ssgi_surface(f"filtered_output_tex:{fi % 2}", torch.uint8, (H, W))
            for r in range(n_ranks):
                r0, r1 = split.strips[r]
                gb = pipes[r].ssgi_surface(f"filtered_output_tex:{fi % 2}", torch.uint8, (H, W))
                assert torch.equal(ga[r0:r1], gb[r0:r1]), f"frame {fi} rank {r}: SSAO guide differs in {int((ga[r0:r1] != gb[r0:r1]).sum())} texels of its strip"
        a = ref.surface("spatial_filtered_tex", torch.int16, (H, W, 4))
        for r in range(n_ranks):
            b = pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4))
            neq = (a != b).any(dim=-1)
            assert not bool(neq.any()), f"frame {fi} rank {r}: {int(neq.sum())} texels differ (rows {torch.nonzero(neq.any(dim=1)).flatten()[:8].tolist()})"
        rc = [pipes[r].ray_counts() for r in range(n_ranks)]
        assert ref.ray_counts() == (sum(c[0] for c in rc), sum(c[1] for c in rc)), (fi, ref.ray_counts(), rc)
        ta = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
        for r in range(n_ranks):
            tb = pipes[r].taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
            neq = (ta != tb).any(dim=-1)
            assert not bool(neq.any()), f"TAA frame {fi} rank {r}: {int(neq.sum())} texels differ (rows {torch.nonzero(neq.any(dim=1)).flatten()[:8].tolist()})"


IRC_BUFS = ("meta", "grid_meta", "entry_cell", "spatial", "irradiance", "aux", "life", "pool", "reposition_proposal", "reposition_proposal_count")


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H", [(2, 256, 160), (3, 320, 208), (2, 2048, 1024)])
def test_strip_split_with_the_irradiance_cache_is_bit_exact(gpu, device, n_ranks, W, H):
    strip_split_with_the_irradiance_cache(gpu, device, n_ranks, W, H)


def strip_split_with_the_irradiance_cache(gpu, device, n_ranks, W, H, scene_name="city20k", frames=None):
    """with the cache bound every rank keeps a replica; the strips' lookups are recorded, all-gathered and replayed in
    one canonical order on every replica (kj_ircache_apply_requests). The replicas must stay bit-identical to each other AND to a
    single GPU running the same frames in the same (deferred, deterministic) mode: GI image, TAA image and every cache buffer.
    (2048x1024 on 2 ranks: a strip holds 262144 half-res pixels, so its request list has the capacity of the cache passes' list,
    2 x 65536 x 4 -- the two used to share one cached buffer)"""
    import torch
    from kajiya_amd import multigpu
    desc = T._scenes()[scene_name]
    scene = gpu.Scene(device, desc)
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=True)
    ref.ircache_set_deferred(True)
    pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
    split = multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=8)
    assert split.consistent_ircache
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = True
    for fi in range(frames or (8 if W < 1024 else 4)):
        cam = frame.orbit_camera(fi, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004) if scene_name.startswith("ruins") else \
            frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.02)
        fc = fs.prepare_frame_constants(cam)
        fs.retire_frame()
        ref.frame(fc)
        for r in range(n_ranks):
            pipes[r].render_inputs(fc)
            pipes[r].reprojection()
        split.gi_frame()
        split.taa_frame()
        ref.taa_frame()
        split.gather_output("spatial_filtered_tex")
        split.gather_output(f"TAA/taa:{fi % 2}")
        torch.cuda.synchronize()
        for name in IRC_BUFS:
            a = ref.ircache_buffer(name, torch.uint8)
            for r in range(n_ranks):
                b = pipes[r].ircache_buffer(name, torch.uint8)
                assert torch.equal(a, b), f"frame {fi} rank {r}: ircache buffer {name} differs in {int((a != b).sum())} bytes"
        a = ref.surface("spatial_filtered_tex", torch.int16, (H, W, 4))
        ta = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
        for r in range(n_ranks):
            neq = (a != pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4))).any(dim=-1)
            assert not bool(neq.any()), f"frame {fi} rank {r}: {int(neq.sum())} GI texels differ (rows {torch.nonzero(neq.any(dim=1)).flatten()[:8].tolist()})"
            neq = (ta != pipes[r].taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))).any(dim=-1)
            assert not bool(neq.any()), f"TAA frame {fi} rank {r}: {int(neq.sum())} texels differ"
    meta = ref.ircache_buffer("meta", torch.int32).cpu().numpy()
    assert meta[3] > 50, meta     # the cache did allocate entries


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H,with_cache,with_ssgi", [(2, 256, 160, False, False), (3, 320, 208, True, False), (8, 192, 256, True, False), (2, 256, 160, False, True), (2, 384, 800, True, True)])
def test_native_split_matches_the_python_orchestrator(gpu, device, n_ranks, W, H, with_cache, with_ssgi):
    """The compiled orchestrator (csrc/split.cpp: KjSplit, virtual ranks = device-to-device exchanges) against the reference
    implementation of the same schedule (multigpu.SplitRtdgi / LocalComm) and against ONE unsplit pipeline: GI image, TAA image and --
    with the cache bound -- every cache buffer bit for bit, over frames with a moving camera."""
    import torch
    from kajiya_amd import multigpu, frame
    desc = T._scenes()["city20k"]
    scene = gpu.Scene(device, desc)
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=with_cache)
    if with_cache:
        ref.ircache_set_deferred(True)
    py_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=with_cache) for r in range(n_ranks)}
    nat_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=with_cache) for r in range(n_ranks)}
    py = multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), py_pipes, W, H, motion_halo=8)
    nat = multigpu.NativeSplit(n_ranks, nat_pipes, W, H, motion_halo=8)
    assert nat.self_test() is True      # kj_split_self_test with every rank in this process: the packed exchanges and the record-list gather on scratch images
    assert [nat.strip(r) for r in range(n_ranks)] == py.strips
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = with_cache
    for fi in range(4 if H > 400 else 6):
        fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.02))
        fs.retire_frame()
        if with_ssgi:      # the SSAO guide: whole-frame on the reference, strip by strip in both orchestrators (kj_split_ssgi_frame / SplitRtdgi.ssgi_frame)
            ref.render_inputs(fc); ref.reprojection(); ref.ssgi_frame(); ref.gi_frame()
        else:
            ref.frame(fc)
        ref.taa_frame()
        for pipes in (py_pipes, nat_pipes):
            for r in range(n_ranks):
                pipes[r].render_inputs(fc)
                pipes[r].reprojection()
        if with_ssgi:
            py.ssgi_frame(); nat.ssgi_frame()
        py.gi_frame(); py.taa_frame()
        nat.gi_frame(); nat.taa_frame()
        for sp in (py, nat):
            sp.gather_output("spatial_filtered_tex")
            sp.gather_output(f"TAA/taa:{fi % 2}")
        torch.cuda.synchronize()
        a = ref.surface("spatial_filtered_tex", torch.int16, (H, W, 4))
        ta = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
        for r in range(n_ranks):
            for tag, pipes in (("python", py_pipes), ("native", nat_pipes)):
                assert torch.equal(a, pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4))), f"frame {fi} rank {r} ({tag}): GI image differs"
                assert torch.equal(ta, pipes[r].taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))), f"frame {fi} rank {r} ({tag}): TAA image differs"
            if with_cache:
                for name in IRC_BUFS:
                    assert torch.equal(ref.ircache_buffer(name, torch.uint8), nat_pipes[r].ircache_buffer(name, torch.uint8)), f"frame {fi} rank {r}: ircache buffer {name} differs"


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_pipelined_split_frames_match_serial_split_frames(gpu, device, native):
    """frame_pipelined of the split (the cache's work of frame N+1 on a side stream; the replay of frame N's recorded cache updates
    deferred to that stream, after the whole frame is enqueued) against the same frames issued serially: GI image, TAA image and every
    cache buffer of every rank bit for bit. Both orchestrators."""
    import ctypes as C
    import torch
    from kajiya_amd import multigpu, frame
    W, H, n_ranks, K = 256, 160, 2, 6
    desc = T._scenes()["city20k"]
    scene = gpu.Scene(device, desc)

    def make():
        pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
        sp = multigpu.NativeSplit(n_ranks, pipes, W, H, motion_halo=8) if native else multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=8)
        return pipes, sp
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = True
    fcs = []
    for fi in range(K + 1):
        fcs.append(fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.02)))
        fs.retire_frame()
    gen = gpu.GpuPipeline(device, scene, W, H)
    inputs = []
    for fc in fcs:
        gen.render_inputs(fc)
        gen.reprojection()
        rp = gpu.tensor_from_ptr(gen.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((gen.geometric_normal.clone(), gen.gbuffer.clone(), gen.depth.clone(), rp, gen.sky16.clone()))
    torch.cuda.synchronize()

    def bind(pipes, i):
        gn, gb, d, rp, sky = inputs[i]
        for q in pipes.values():
            q.geometric_normal, q.gbuffer, q.depth, q.sky16 = gn, gb, d, sky
            q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
    ser_pipes, ser = make()
    for i in range(K):
        bind(ser_pipes, i)
        device.frame_begin(fcs[i])
        ser.gi_frame()
        ser.taa_frame()
    torch.cuda.synchronize()
    pip_pipes, pip = make()
    bind(pip_pipes, 0)
    pip.pipeline_begin(fcs[0])
    for i in range(K):
        bind(pip_pipes, i)
        pip.frame_pipelined(fcs[i + 1] if i + 1 < K else None)
    torch.cuda.synchronize()
    for sp in (ser, pip):
        sp.gather_output("spatial_filtered_tex")
        sp.gather_output(f"TAA/taa:{(K - 1) % 2}")
    torch.cuda.synchronize()
    for r in range(n_ranks):
        assert torch.equal(ser_pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4)), pip_pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4))), f"rank {r}: GI image"
        assert torch.equal(ser_pipes[r].taa_surface(f"taa:{(K - 1) % 2}", torch.int16, (H, W, 4)), pip_pipes[r].taa_surface(f"taa:{(K - 1) % 2}", torch.int16, (H, W, 4))), f"rank {r}: TAA image"
        for name in IRC_BUFS:
            assert torch.equal(ser_pipes[r].ircache_buffer(name, torch.uint8), pip_pipes[r].ircache_buffer(name, torch.uint8)), f"rank {r}: ircache buffer {name}"


def test_strip_plan_and_transfers():
    from kajiya_amd import multigpu
    for H, n in ((1080, 8), (2160, 8), (1080, 3), (160, 2)):
        st = multigpu.plan_strips(H, n)
        assert st[0][0] == 0 and st[-1][1] == H
        assert all(a % 16 == 0 for a, _ in st) and all(st[i][1] == st[i + 1][0] for i in range(n - 1))
        x = multigpu.transfers(st, 51, "h", H)
        hh = (H + 1) // 2
        for dst in range(n):
            own = multigpu.half_rows(*st[dst], H)
            need = set(range(max(0, own[0] - 51), min(hh, own[1] + 51))) - set(range(*own))
            got = set()
            for (s, d, a, b) in x:
                if d == dst:
                    so = multigpu.half_rows(*st[s], H)
                    assert so[0] <= a < b <= so[1]
                    got |= set(range(a, b))
            assert got == need


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H", [(2, 256, 160), (3, 320, 208), (8, 192, 256)])
def test_strip_split_sun_shadows_are_bit_exact(gpu, device, n_ranks, W, H):
    """trace_sun_shadow_mask + ShadowDenoiseRenderer::render strip by strip (SURVEY 8f-2 under the screen-tile split): both orchestrators
    (SplitRtdgi.shadow_frame, kj_split_shadow_frame) against the unsplit passes over frames with a moving camera -- on every rank's own rows the
    mask, the denoised term and both histories of the denoiser bit for bit, and the rays traced add up to one per pixel; then light_gbuffer
    on each rank's rows against the whole-frame combine."""
    import torch
    from kajiya_amd import multigpu, frame
    desc = T._scenes()["city20k"]
    scene = gpu.Scene(device, desc)
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=False)
    py_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=False) for r in range(n_ranks)}
    nat_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=False) for r in range(n_ranks)}
    py = multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), py_pipes, W, H, motion_halo=8)
    nat = multigpu.NativeSplit(n_ranks, nat_pipes, W, H, motion_halo=8)
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = False
    counters = {tag: {r: torch.zeros(1, dtype=torch.int64, device=ref.depth.device) for r in range(n_ranks)} for tag in ("python", "native")}
    ref_counter = torch.zeros(1, dtype=torch.int64, device=ref.depth.device)
    for fi in range(5):
        fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.02))
        fs.retire_frame()
        ref.render_inputs(fc); ref.reprojection()
        mask = ref.sun_shadow_mask(ray_counter=ref_counter)
        dn = ref.shadow_denoise(mask).view(torch.int16)
        for pipes in (py_pipes, nat_pipes):
            for r in range(n_ranks):
                pipes[r].render_inputs(fc)
                pipes[r].reprojection()
        outs = {"python": py.shadow_frame(ray_counters=counters["python"]), "native": nat.shadow_frame(ray_counters=counters["native"])}
        torch.cuda.synchronize()
        for tag, pipes in (("python", py_pipes), ("native", nat_pipes)):
            for r in range(n_ranks):
                a, b = py.strips[r]
                assert torch.equal(mask[a:b], pipes[r].shadow_mask_img[a:b]), f"frame {fi} rank {r} ({tag}): mask differs"
                got = outs[tag][r].view(torch.int16)
                neq = (dn[a:b] != got[a:b]).any(dim=-1)
                assert not bool(neq.any()), f"frame {fi} rank {r} ({tag}): {int(neq.sum())} denoised texels differ (rows {(torch.nonzero(neq.any(dim=1)).flatten()[:8] + a).tolist()})"
                for name, words in (("shadow_denoise_moments", 4), ("shadow_denoise_accum", 2)):
                    x = ref.shadow_denoise_surface(f"{name}:{fi % 2}", torch.int16, (H, W, words))
                    y = pipes[r].shadow_denoise_surface(f"{name}:{fi % 2}", torch.int16, (H, W, words))
                    assert torch.equal(x[a:b], y[a:b]), f"frame {fi} rank {r} ({tag}): history {name} differs"
    for tag in ("python", "native"):
        assert sum(int(c.item()) for c in counters[tag].values()) == int(ref_counter.item()) > 0, tag       # every ray traced once across the ranks
    # the deferred combine on a rank's rows (every input read at the pixel itself) against the whole frame
    ref.gi_frame()
    dn2 = ref.shadow_denoise(mask)
    whole2 = [t.clone() for t in ref.light_gbuffer(dn2)]
    for t in ref._lit:
        t.zero_()
    for a, b in py.strips:
        ref.light_gbuffer(dn2, rows=(a, b))
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(whole2[k].view(torch.int16), ref._lit[k].view(torch.int16)), "light_gbuffer strip by strip differs from the whole-frame combine"
