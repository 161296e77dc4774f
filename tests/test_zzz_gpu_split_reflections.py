"""The reflections and the whole lighting frame of BASELINE configs[2] under the screen-tile split, bit for bit against the same frames
on one GPU. A file of its own that sorts behind every other GPU test file: these tests were written after round 3's GPU budget was spent and have run on the CPU
stand-in only (scripts/run_gpu_suite_on_cpu.sh); under `pytest -x` a hardware-only surprise here must not hide the suite that has run on hardware."""
import pytest

import test_gpu_parity as T  # noqa: F401  (fixtures' helpers)
from test_gpu_multigpu import IRC_BUFS


RTR_HALF = ("rtr.irradiance", "rtr.ray_orig", "rtr.ray", "rtr.reservoir", "rtr.rng", "rtr.hit_normal")
RTR_FULL = ("rtr.temporal", "rtr.ray_len")


def _rtr_own_rows_equal(ref, gp, strip, fi, H, torch, what):
    """Every surface of RtrRenderer on a rank's own rows against the unsplit frame: the eight ping-pong outputs of frame `fi`, the
    invalidity image, the resolved image light_gbuffer reads."""
    a, b = strip
    ha, hb = a // 2, ((H + 1) // 2 if b == H else b // 2)
    hh = (H + 1) // 2
    for name in RTR_HALF + ("refl_restir_invalidity_tex",):
        n = name if name.startswith("refl") else f"{name}:{fi % 2}"
        x, y = ref.rtr_surface(n, torch.uint8, (hh, -1)), gp.rtr_surface(n, torch.uint8, (hh, -1))
        neq = (x[ha:hb] != y[ha:hb]).any(dim=1)
        assert not bool(neq.any()), f"{what}: {n} differs on half-res rows {(torch.nonzero(neq).flatten()[:8] + ha).tolist()} of [{ha}, {hb})"
    for name in RTR_FULL + ("resolved_tex",):
        n = name if name == "resolved_tex" else f"{name}:{fi % 2}"
        x, y = ref.rtr_surface(n, torch.uint8, (H, -1)), gp.rtr_surface(n, torch.uint8, (H, -1))
        neq = (x[a:b] != y[a:b]).any(dim=1)
        assert not bool(neq.any()), f"{what}: {n} differs on rows {(torch.nonzero(neq).flatten()[:8] + a).tolist()} of [{a}, {b})"


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H,with_cache,lights,wide", [(2, 256, 160, False, False, False), (3, 320, 208, True, False, False), (2, 192, 416, True, True, False),
                                                                 (8, 192, 256, False, False, False), (2, 160, 416, False, False, True), (3, 64, 1248, False, False, True), (2, 171, 99, True, False, False), (3, 123, 77, False, True, False)])
def test_strip_split_reflections_are_bit_exact(gpu, device, n_ranks, W, H, with_cache, lights, wide):
    """RtrRenderer::trace + render_specular + filter_temporal strip by strip (SURVEY 8f-3 under the screen-tile split; VERDICT r2 missing #5): both
    orchestrators (SplitRtdgi.rtr_frame, kj_split_rtr_frame) against the unsplit passes over frames with a moving camera -- on every rank's own rows all
    eight ping-pong temporals, the invalidity image and the resolved image bit for bit, the rtdgi candidates the trace pass overwrites, and with the
    cache bound every cache buffer (rtr's rays record their lookups in slot ranges of their own; the replay follows them). 192x416 on two ranks: strips
    taller than every halo, with triangle lights (the lights' specular pass). `wide`: a 100-degree field of view from just above the mirror floor, strips of
    104 half-res rows -- the resolve's taps, whose halo is sized from the field of view (multigpu.rtr_resolve_halo: 31 / 43 rows here), at their longest:
    grazing surfaces under strong perspective."""
    import torch
    from kajiya_amd import multigpu, frame, scenes as S
    desc = S.glossy_test_scene()
    scene = gpu.Scene(device, desc, use_lights=lights)
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=with_cache)
    if with_cache:
        ref.ircache_set_deferred(True)
        ref.ircache_set_rtr_requests(True)
    splits = {}
    for tag in ("python", "native"):
        pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=with_cache) for r in range(n_ranks)}
        sp = multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=8) if tag == "python" else multigpu.NativeSplit(n_ranks, pipes, W, H, motion_halo=8)
        sp.enable_rtr()
        splits[tag] = (sp, pipes)
    strips = splits["python"][0].strips
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = with_cache
    fs.triangle_light_count = scene.triangle_light_count if lights else 0
    for fi in range(6):
        cam = frame.orbit_camera(fi, (W, H), center=(0.0, 0.6, 0.0), radius=4.0, height=0.25, rate=0.004, vfov=100.0) if wide else \
            frame.orbit_camera(fi, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.008)
        fc = fs.prepare_frame_constants(cam)
        fs.retire_frame()
        ref.render_inputs(fc); ref.reprojection()
        ref.gi_frame(defer_replay=True)
        ref.rtr_frame(specular_lights=lights)
        if with_cache:
            ref.ircache_replay_own_requests()
        for tag, (sp, pipes) in splits.items():
            for r in range(n_ranks):
                pipes[r].render_inputs(fc)
                pipes[r].reprojection()
            sp.gi_frame()
            sp.rtr_frame(specular_lights=lights)
        torch.cuda.synchronize()
        hh = (H + 1) // 2
        for tag, (sp, pipes) in splits.items():
            for r in range(n_ranks):
                # (the GI image first: this scene found the rtdgi split's own row-0 dependence -- a reservoir no history tap was selected into on a
                # validation frame keeps payload 0, and next frame's temporal pass follows it to pixel (0, 0) of the sample images, which here is geometry)
                x, y = ref.surface("spatial_filtered_tex", torch.uint8, (H, -1)), pipes[r].surface("spatial_filtered_tex", torch.uint8, (H, -1))
                assert torch.equal(x[strips[r][0]:strips[r][1]], y[strips[r][0]:strips[r][1]]), f"frame {fi} rank {r} ({tag}): the GI image differs"
                _rtr_own_rows_equal(ref, pipes[r], strips[r], fi, H, torch, f"frame {fi} rank {r} ({tag})")
                ha, hb = strips[r][0] // 2, (hh if strips[r][1] == H else strips[r][1] // 2)
                for n in ("candidate_radiance_tex", "candidate_hit_tex", "candidate_normal_tex"):
                    x, y = ref.surface(n, torch.uint8, (hh, -1)), pipes[r].surface(n, torch.uint8, (hh, -1))
                    assert torch.equal(x[ha:hb], y[ha:hb]), f"frame {fi} rank {r} ({tag}): {n} differs"
                if with_cache:
                    for name in IRC_BUFS:
                        x, y = ref.ircache_buffer(name, torch.uint8), pipes[r].ircache_buffer(name, torch.uint8)
                        assert torch.equal(x, y), f"frame {fi} rank {r} ({tag}): ircache buffer {name} differs in {int((x != y).sum())} bytes"
    closest, any_hit = ref.rtr_ray_counts()
    assert closest > 0
    assert (scene.triangle_light_count > 0) == lights
    if with_cache:
        assert ref.ircache_buffer("meta", torch.int32).cpu().numpy()[3] > 20     # the cache did allocate entries


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,W,H,native", [(2, 256, 160, False), (3, 192, 208, True), (8, 1920, 1080, True)])
def test_whole_lighting_frame_under_the_split_is_bit_exact(gpu, device, n_ranks, W, H, native):
    """BASELINE configs[2] under the split (lighting_frame: SSAO guide, sun shadows + denoiser, irradiance cache + rtdgi, reflections, the deferred combine,
    TAA on the lit image -- world_render_passes.rs:99-291) against the same frames on one GPU in scripts/config3_bench.py's order: on every rank's own rows the
    lit image and the TAA output bit for bit, and every replica of the cache. 1920x1080 on 8 ranks: the halos at the proportions of a real run (135-row strips, the
    resolve's 36 + 8 half-res rows, motion halo 16) -- too slow for the CPU stand-in's suite, where scripts/config3_split_bench.py --check ran it once (0 texels)."""
    import torch
    from kajiya_amd import multigpu, frame, scenes as S
    scene = gpu.Scene(device, S.glossy_test_scene())
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=True)
    ref.ircache_set_deferred(True)
    ref.ircache_set_rtr_requests(True)
    pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
    halo = 8 if H < 1000 else 16
    sp = multigpu.NativeSplit(n_ranks, pipes, W, H, motion_halo=halo) if native else multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=halo)
    sp.enable_rtr()
    fs = frame.FrameState((W, H), sun_size_multiplier=4.0)
    fs.ircache_enabled = True
    for fi in range(6):
        fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.008 if H < 1000 else 0.004))
        fs.retire_frame()
        ref.render_inputs(fc); ref.reprojection()
        ref.ssgi_frame()
        shadow = ref.shadow_denoise(ref.sun_shadow_mask())
        ref.gi_frame(defer_replay=True)
        rtr = ref.rtr_frame()
        ref.ircache_replay_own_requests()
        lit = ref.light_gbuffer(shadow, rtr_ptr=rtr.data_ptr())[1]
        ref.taa_frame(input_ptr=lit.data_ptr())
        for r in range(n_ranks):
            pipes[r].render_inputs(fc)
            pipes[r].reprojection()
        assert multigpu.max_vertical_motion_rows(gpu.tensor_from_ptr(ref.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)), H) <= halo      # the test's own precondition
        lits = sp.lighting_frame()
        torch.cuda.synchronize()
        ta = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
        for r in range(n_ranks):
            a, b = sp.strips[r]
            assert torch.equal(lit.view(torch.int16)[a:b], lits[r].view(torch.int16)[a:b]), f"frame {fi} rank {r}: the lit image differs"
            neq = (ta[a:b] != pipes[r].taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))[a:b]).any(dim=-1)
            assert not bool(neq.any()), f"frame {fi} rank {r}: {int(neq.sum())} TAA texels differ (rows {(torch.nonzero(neq.any(dim=1)).flatten()[:8] + a).tolist()})"
            for name in IRC_BUFS:
                x, y = ref.ircache_buffer(name, torch.uint8), pipes[r].ircache_buffer(name, torch.uint8)
                assert torch.equal(x, y), f"frame {fi} rank {r}: ircache buffer {name} differs"
    assert float(lit.float().abs().max()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_pipelined_lighting_frames_match_serial_lighting_frames(gpu, device, native):
    """lighting_frame_pipelined (BASELINE configs[2] under the split with the cache's work of frame N+1 and the replay of frame N's recorded updates on a side
    stream, started behind frame N's reflection rays) against the same frames issued serially with lighting_frame: every rank's rows of the lit image and of
    the TAA output, all of RtrRenderer's temporals on those rows, and every cache buffer, bit for bit. Both orchestrators."""
    import ctypes as C
    import torch
    from kajiya_amd import multigpu, frame, scenes as S
    W, H, n_ranks, K = 192, 160, 2, 5
    scene = gpu.Scene(device, S.glossy_test_scene())

    def make():
        pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
        sp = multigpu.NativeSplit(n_ranks, pipes, W, H, motion_halo=8) if native else multigpu.SplitRtdgi(multigpu.LocalComm(n_ranks), pipes, W, H, motion_halo=8)
        sp.enable_rtr()
        return pipes, sp
    fs = frame.FrameState((W, H), sun_size_multiplier=4.0)
    fs.ircache_enabled = True
    fcs = []
    for fi in range(K + 1):
        fcs.append(fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 1.5, 0.0), radius=9.0, height=3.5, rate=0.008)))
        fs.retire_frame()
    gen = gpu.GpuPipeline(device, scene, W, H)
    inputs = []
    for fc in fcs:
        gen.render_inputs(fc)
        gen.reprojection()
        rp = gpu.tensor_from_ptr(gen.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((gen.geometric_normal.clone(), gen.gbuffer.clone(), gen.depth.clone(), rp, gen.sky16.clone(), gen.sky64.clone()))
    torch.cuda.synchronize()

    def bind(pipes, i):
        gn, gb, d, rp, sky16, sky64 = inputs[i]
        for q in pipes.values():
            q.geometric_normal, q.gbuffer, q.depth, q.sky16, q.sky64 = gn, gb, d, sky16, sky64
            q.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
    ser_pipes, ser = make()
    for i in range(K):
        bind(ser_pipes, i)
        device.frame_begin(fcs[i])
        ser_lit = {r: t.clone() for r, t in ser.lighting_frame().items()}
    torch.cuda.synchronize()
    pip_pipes, pip = make()
    bind(pip_pipes, 0)
    pip.pipeline_begin(fcs[0])
    for i in range(K):
        bind(pip_pipes, i)
        pip_lit = pip.lighting_frame_pipelined(fcs[i + 1] if i + 1 < K else None)
    torch.cuda.synchronize()
    for r in range(n_ranks):
        a, b = ser.strips[r]
        assert torch.equal(ser_lit[r].view(torch.int16)[a:b], pip_lit[r].view(torch.int16)[a:b]), f"rank {r}: lit image"
        x, y = (p[r].taa_surface(f"taa:{(K - 1) % 2}", torch.int16, (H, W, 4)) for p in (ser_pipes, pip_pipes))
        assert torch.equal(x[a:b], y[a:b]), f"rank {r}: TAA image"
        _rtr_own_rows_equal(ser_pipes[r], pip_pipes[r], (a, b), K - 1, H, torch, f"rank {r}")
        for name in IRC_BUFS:
            assert torch.equal(ser_pipes[r].ircache_buffer(name, torch.uint8), pip_pipes[r].ircache_buffer(name, torch.uint8)), f"rank {r}: ircache buffer {name}"
