"""GPU-vs-oracle parity at the sizes of BASELINE.json's HEADLINE configs (until round 3 nothing above 1080p had a
parity test, and rtr / shadows / the deferred combine were only compared at <= 256^2):

  configs[3] / north-star target   3840x2160 on the ~4 M-triangle procedural ruins (the Ruins asset is not in the reference checkout):
                                   every rtdgi pass in isolation + TAA, on identical inputs
  configs[2]                       2560x1440 on the same scene, the full lighting frame of world_render_passes.rs:124-291: SSAO guide,
                                   sun shadow mask + denoiser, rtdgi, rtr (six passes), light_gbuffer, TAA on the lit image

Same bars as everywhere else (tests/parity.py): rel-L2 <= 1e-3 AND <= 0.2 % outlier texels AND no finite / non-finite disagreement,
the flips form for ray passes. Sized for the GPU box: the oracle needs 2-5 s per pass-frame at these extents, so fewer frames than
the small-extent tests (which cover the ragged / edge cases); each compared set still holds a validation and a tracing frame."""
import ctypes as C
import numpy as np
import pytest

import parity as P
import test_gpu_parity as T
import test_gpu_taa as TT
import test_gpu_rtr as TR
import test_gpu_shadow_denoise as TS

pytestmark = pytest.mark.gpu


def test_rtdgi_per_pass_and_taa_parity_at_4k_ruins(gpu, oracle, device):
    """Frames 0-2 warm up (whole frames, state forced to the oracle's); frame 3 is a validation frame (3 % 3 == 0), frame 4 a tracing
    frame: every rtdgi pass in isolation, then TAA (all 15 surfaces) on the frame the oracle has just finished -- a 4K oracle frame
    costs 10-30 s of host time, so the two share it."""
    taa = TT.TaaStep(gpu, 3840, 2160)
    T._per_pass_parity(gpu, oracle, device, "ruins4m", 3840, 2160, 2, False, n_frames=5, warmup=3, with_cache=True,    # cache BOUND
                       after_frame=lambda op, gp, fi, fc: taa(op, gp, fi, fc, compare=fi >= 3))      # the two pass-by-pass frames; TAA's history is dense by then
    print(f"TAA at 4K on identical inputs and history: worst per-surface rel-L2 {taa.worst:.2e}")


def test_config3_lighting_frame_parity_at_1440p_ruins(gpu, oracle, device):
    config3_lighting_frame_parity(gpu, oracle, device, "ruins4m", 2560, 1440)


def test_config3_lighting_frame_parity_small(gpu, oracle, device):
    """The same composite on a small extent (odd sizes: ragged half- / quarter-res images), cheap enough for the CPU stand-in."""
    config3_lighting_frame_parity(gpu, oracle, device, "city20k", 171, 99)


def config3_lighting_frame_parity(gpu, oracle, device, scene_name, W, H):
    """One frame = ssgi -> sun shadow mask -> shadow denoise -> rtdgi -> rtr -> light_gbuffer -> TAA (config3_bench.py's order). Every
    stage runs on both sides from identical inputs and identical temporal state (the oracle's, uploaded); 2 warm-up frames, then 2
    compared frames."""
    import torch
    from kajiya_amd import frame
    desc = T._scenes()[scene_name]
    op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
    fs = frame.FrameState((W, H), sun_size_multiplier=4.0)
    fcs = []
    for i in range(4):
        cam = frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004) if scene_name.startswith("ruins") else \
            frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
        fcs.append(fs.prepare_frame_constants(cam))
        fs.retire_frame()
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    ssao_dev = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    lit_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    report = {}

    def note(key, r):
        if key not in report or r["rel_l2"] > report[key]["rel_l2"]:
            report[key] = r

    for fi, fc in enumerate(fcs):
        compared = fi >= 2
        op.render_inputs(fc); op.reprojection(fc)
        gp.dev.frame_begin(fc)
        T._sync_inputs(op, gp, torch)
        gp.sky64.copy_(torch.from_numpy(op.sky64.view(np.int16)))
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        # ---- SSAO guide (SsgiRenderer::render), identical history
        if fi > 0:
            for n in ("ssgi:0", "ssgi:1"):
                gp.ssgi_surface(n, torch.uint8, (-1,)).copy_(torch.from_numpy(op.ssgi_surface(n, np.uint8, (-1,)).copy()))
        ref_ao = op.ssgi_frame(fc).copy()
        gp.ssgi_frame()
        torch.cuda.synchronize()
        got_ao = gpu.tensor_from_ptr(gp.ssao_ptr.value, W * H, torch.uint8, (H, W)).cpu().numpy()
        if compared:
            # R8 rounding flips (one level) anywhere; beyond that only where the horizon search took a different discrete step (measured at
            # 1440p on the ruins: 2.7e-4 of the pixels, by <= 4 levels)
            d = np.abs(got_ao.astype(np.int32) - ref_ao.astype(np.int32))
            rel = float(np.sqrt((d.astype(np.float64) ** 2).sum() / max(1.0, (ref_ao.astype(np.float64) ** 2).sum())))
            report[("ssao guide", "R8")] = dict(rel_l2=rel, mismatch_frac=float((d > 1).mean()))
            assert (d > 0).mean() < 5e-3 and (d > 1).mean() <= P.MISMATCH_TOL and rel <= P.REL_L2_TOL, f"ssao frame {fi}: max {d.max()}, differing {(d > 0).mean():.2e}, by more than one level {(d > 1).mean():.2e}, rel-L2 {rel:.2e}"
        ssao_dev.copy_(torch.from_numpy(ref_ao)); gp.ssao_ptr = C.c_void_p(ssao_dev.data_ptr())     # identical guide downstream
        # ---- sun shadow mask (one soft-shadow ray per pixel) + denoiser
        ref_mask = op.sun_shadow_mask(fc)
        got_mask = gp.sun_shadow_mask().cpu().numpy()
        if compared:
            flips = float((got_mask != ref_mask).mean())
            report[("shadow mask", "flipped pixels")] = dict(rel_l2=flips, mismatch_frac=flips)
            assert flips <= P.MISMATCH_TOL, ("sun shadow mask", fi, flips)
        if fi > 0:
            for n in ("shadow_denoise_accum:0", "shadow_denoise_accum:1", "shadow_denoise_moments:0", "shadow_denoise_moments:1"):
                gp.shadow_denoise_surface(n, torch.uint8, (-1,)).copy_(torch.from_numpy(op.shadow_denoise_surface(n, np.uint8, (-1,)).copy()))
        ref_shadow = op.shadow_denoise(fc, ref_mask)
        got_shadow = gp.shadow_denoise(torch.from_numpy(ref_mask).cuda())
        torch.cuda.synchronize()
        if compared:
            for name, fmt in TS.SURF.items():
                a = gp.shadow_denoise_surface(name, torch.uint8, (-1,)).cpu().numpy()
                b = op.shadow_denoise_surface(name, np.uint8, (-1,))
                if fmt == "u32":
                    assert np.array_equal(a, b), (fi, name)
                    continue
                r = P.compare(a, b, fmt)
                note(("shadow denoise", name.split(":")[0]), r)
                assert P.within_bars(r), (fi, name, r)
        # ---- rtdgi (per-pass parity at this scene: the 4K test above) + rtr, six passes in isolation on identical state
        if not compared:
            op.rtdgi_frame(fc); gp.rtdgi_frame()
            torch.cuda.synchronize()
            T._upload_state(gp, T._oracle_surfaces(op), torch)
            op.rtr_frame(fc); gp.rtr_frame()
            torch.cuda.synchronize()
            TR._upload_rtr_state(gp, TR._oracle_rtr_state(op), torch)
        else:
            pre = T._oracle_surfaces(op)
            T._upload_state(gp, pre, torch)
            op.rtdgi_frame(fc); gp.rtdgi_frame()
            torch.cuda.synchronize()
            r = P.compare(gp.surface("spatial_filtered_tex", torch.uint8, (-1,)).cpu().numpy(), op.surface("spatial_filtered_tex", np.uint8, (-1,)), "rgba16f")
            note(("rtdgi whole frame", "spatial_filtered_tex"), r)
            # the whole frame in one go (its passes in isolation: the 4K test above): a candidate whose shadow ray or depth gate flipped
            # (1e-4 of the half-res texels) reaches ~50 full-res neighbours through the resampling chain and the denoiser, each by a
            # little -- the image-level bar holds, the count of slightly-off texels is reported and bounded loosely
            # (measured on MI355X: round 4 4.5e-3 / 4.1e-4 of the texels at 1440p; round 6, with the ray passes and TAA compiled without FMA contraction since, 8.6e-4 at worst
            # (profiles/r06_gpu_tests_summary.txt): the bar is twice that, rounded up)
            P.measured("1440p whole rtdgi frame: mismatch fraction (bar 2e-3)", r["mismatch_frac"])
            assert r["rel_l2"] <= P.REL_L2_TOL and r["bad_class"] == 0 and r["mismatch_frac"] <= 2e-3, f"rtdgi whole frame {fi}: {r}"
            T._upload_state(gp, T._oracle_surfaces(op), torch)
            for k, pname in enumerate(TR.RTR_PASS_ORDER):
                mask = TR.KJ_RTR_PASS[pname] | (0 if k == 0 else TR.KJ_RTR_PASS["KEEP"])
                if k > 0:
                    TR._upload_rtr_state(gp, TR._oracle_rtr_state(op), torch)
                op.rtr_frame(fc, mask); gp.rtr_frame(mask)
                torch.cuda.synchronize()
                ref, got = TR._oracle_rtr_state(op), TR._download_rtr_state(gp, torch)
                for n in ref:
                    r, ok = TR.rtr_surface_within_bars(pname, n, got[n], ref[n])
                    note(("rtr " + pname, P.base_name(n)), r)
                    assert ok, f"frame {fi} rtr pass {pname} surface {n}: {r}"
            TR._upload_rtr_state(gp, TR._oracle_rtr_state(op), torch)
        # ---- deferred combine on the oracle's shadow mask, GI and reflections (the oracle's combine reads the R8 mask)
        rtr_ref = op.rtr_surface("resolved_tex", np.uint32, (H, W)).copy()
        gi_ref = op.surface("spatial_filtered_tex", np.float16, (H, W, 4)).copy()
        ref_t, ref_o = op.light_gbuffer(fc, ref_mask, gi_ref, rtr_ref, 0)
        d_mask, d_gi, d_rtr = torch.from_numpy(ref_mask).cuda(), torch.from_numpy(gi_ref.view(np.int16)).cuda(), torch.from_numpy(rtr_ref.view(np.int32)).cuda()
        got_t, got_o = gp.light_gbuffer(d_mask, rtdgi_ptr=d_gi.data_ptr(), rtr_ptr=d_rtr.data_ptr())
        torch.cuda.synchronize()
        if compared:
            for name, a, b in (("temporal_output", got_t, ref_t), ("output", got_o, ref_o)):
                r = P.compare(a.cpu().numpy().view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1), "rgba16f")
                note(("light_gbuffer", name), r)
                assert P.within_bars(r), (fi, name, r)
        # ---- TAA on the lit image, identical history
        if fi > 0:
            for n in ("taa:0", "taa:1", "taa.velocity:0", "taa.velocity:1", "taa.smooth_var:0", "taa.smooth_var:1"):
                gp.taa_surface(n, torch.uint8, (-1,)).copy_(torch.from_numpy(op.taa_surface(n, np.uint8, (-1,)).copy()))
        lit = np.ascontiguousarray(ref_o)
        op.taa_frame(fc, input_ptr=lit.ctypes.data)
        lit_dev.copy_(torch.from_numpy(lit.view(np.int16)))
        gp.taa_frame(input_ptr=lit_dev.data_ptr())
        torch.cuda.synchronize()
        if compared:
            r = P.compare(gp.taa_surface("this_frame_output_img", torch.uint8, (-1,)).cpu().numpy(), op.taa_surface("this_frame_output_img", np.uint8, (-1,)), "rgba16f")
            note(("taa", "this_frame_output_img"), r)
            assert P.within_bars(r), ("taa", fi, r)
    for k, v in sorted(report.items()):
        print(f"  {k[0]:>22s} {k[1]:<30s} rel_l2={v['rel_l2']:.2e} mismatch={v['mismatch_frac']:.2e}")


# ---------------------------------------------------------------------------------------------------------------- VERDICT r3 items 1b-1d
def test_rtdgi_per_pass_parity_with_the_cache_bound_small(gpu, oracle, device):
    """Every rtdgi pass in isolation with the irradiance cache bound (see _per_pass_parity: `with_cache`), small enough for the CPU stand-in;
    the 1080p and 4K per-pass tests run the same way. Frame 6 validates (6 % 3 == 0), 7 traces."""
    T._per_pass_parity(gpu, oracle, device, "cornell", 128, 128, 2, False, n_frames=8, warmup=6, with_cache=True)


def test_configs3_strip_split_4k_four_ranks_is_bit_exact(gpu, device):
    """BASELINE configs[3] as stated: the 4K frame of the ~4 M-triangle ruins under a 4-way screen-tile split (virtual ranks on one GPU:
    every rank's strip + halos, the exchanges as device copies), irradiance cache bound and kept consistent across the ranks: GI image,
    TAA image and every cache buffer bit-identical to one GPU running the same frames."""
    import test_gpu_multigpu as TM
    TM.strip_split_with_the_irradiance_cache(gpu, device, 4, 3840, 2160, scene_name="ruins4m", frames=3)


def test_configs4_reference_pt_at_4k(gpu, oracle, device):
    """BASELINE configs[4] as stated (minus the 64 spp, which only repeat the sample): the reference path tracer on the 4K frame of the
    ruins. (1) one sample per pixel against the oracle on a 256-row band through the middle of the frame (a 4K oracle sample is ~8 M
    paths: the band keeps the host time at seconds), per-pixel bars of test_reference_pt_matches_oracle; (2) the 8-way pixel-interleaved
    split: every pixel owned by exactly one rank, and the ranks' images sum to the unsplit one bit for bit, two samples deep."""
    import torch
    import test_gpu_reference_pt as TP
    W, H = 3840, 2160
    desc = T._scenes()["ruins4m"]
    osc = oracle.OracleScene(desc)
    gsc = gpu.Scene(device, desc)
    gp = gpu.GpuPipeline(device, gsc, W, H)
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    fcs = []
    for i in range(2):
        fcs.append(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.004)))
        fs.retire_frame()
    y0, y1 = 1000, 1256
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    parts = [torch.zeros_like(full) for _ in range(8)]
    for fi, fc in enumerate(fcs):
        device.frame_begin(fc)
        one = torch.zeros_like(full)
        gp.reference_path_trace(one)
        gp.reference_path_trace(full)
        for r in range(8):
            gp.reference_path_trace(parts[r], interleave=(8, r))
        if fi == 0:
            ref = np.zeros((H, W, 4), np.float32)
            oracle.reference_path_trace_rows(osc, fc, ref, y0, y1)
            g = one[y0:y1].cpu().numpy()
            o = ref[y0:y1]
            assert np.isfinite(g).all() and (g[..., 3] == 1.0).all() and (g[..., :3] >= 0).all()
            hit = o[..., 3] > 0
            assert hit.mean() > 0.5, hit.mean()
            err = np.abs(g[..., :3] - o[..., :3]).max(axis=-1) / (1e-3 + np.abs(o[..., :3]).max(axis=-1))
            frac = float((err > 1e-3).mean())
            mean_g, mean_o = g[..., :3].mean(axis=(0, 1)), o[..., :3].mean(axis=(0, 1))
            print(f"  reference PT at 4K, rows {y0}..{y1}: one-sample pixels off by more than 1e-3: {frac:.5f}; band mean gpu {mean_g} oracle {mean_o}")
            assert frac < 0.02, frac
            assert np.allclose(mean_g, mean_o, rtol=0.03), (mean_g, mean_o)
    torch.cuda.synchronize()
    owned = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    total = torch.zeros_like(full)
    for p in parts:
        owned += (p[..., 3] > 0).int()
        total += p
    assert bool((owned == 1).all())
    assert torch.equal(total, full)
