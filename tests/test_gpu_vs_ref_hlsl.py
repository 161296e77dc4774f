"""The DEVICE against the reference's own shader text, directly: the two halves of the parity chain -- kernels vs oracle at the sizes BASELINE
names, oracle vs the compiled HLSL at a few thousand texels -- met only through the oracle and at different sizes. Here one 1080p frame's state goes to BOTH the MI355X
kernels and oracle/_ref/libref_hlsl.so (kajiya's HLSL compiled for the CPU by oracle/ref_hlsl; the prebuilt library travels to the GPU box with the snapshot, the reference
checkout does not have to), pass by pass, and the kernels' surfaces are held to the text's under tests/parity.py's bars. The oracle only supplies the state the passes start from.

Passes: the ray-free ones of rtdgi (reproject, half-res extracts, validity integrate, restir temporal, restir spatial x2, resolve, temporal filter, spatial filter): 17 surfaces.
(The ray passes need the reference's hit shaders bound to a scene on the text's side: tests/test_ref_hlsl.py does that at small extents.) Round 6: TAA's seven passes
the same way (test_device_taa_passes_against_the_reference_text: 15 more surfaces) -- every device pass starts from the inputs the text's pass reads, uploaded, so that the
probability stage's amplification of one fp16 ulp to O(1) cannot compound from pass to pass. A minute or two of CPU for the text at 1080p per test; one
frame after five warm-up frames. KJ_TEST_VS_TEXT_EXTENT=WxH runs another extent (the CPU stand-in of the GPU suite uses a small one)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity as P  # noqa: E402
import ref_hlsl as R  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_name,W,H", [("city20k", 200, 120), ("city20k", 1920, 1080)])
def test_device_rtdgi_screen_passes_against_the_reference_text(gpu, oracle, device, scene_name, W, H):
    import torch
    import test_gpu_parity as T
    import test_ref_hlsl as RH
    from kajiya_amd.abi import KJ_RTDGI_PASS
    R.require_live("the device is compared with the compiled reference text itself")
    if os.environ.get("KJ_TEST_VS_TEXT_EXTENT"):
        W, H = (int(v) for v in os.environ["KJ_TEST_VS_TEXT_EXTENT"].split("x"))
    RH._bind_luts(oracle)
    desc = T._scenes()[scene_name]
    op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
    fcs = T._frame_constants(W, H, 6, T.camera_of(scene_name))
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    worst = {}
    for fi, fc in enumerate(fcs):
        op.render_inputs(fc); op.reprojection(fc)
        gp.dev.frame_begin(fc)
        T._sync_inputs(op, gp, torch)
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        if fi < 5:
            op.rtdgi_frame(fc); gp.rtdgi_frame()
            torch.cuda.synchronize()
            T._upload_state(gp, T._oracle_surfaces(op), torch)
            continue
        before = RH._surfaces(op)
        T._upload_state(gp, T._oracle_surfaces(op), torch)
        op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
        gpu.check(gp.L.kj_rtdgi_reproject(gp.rtdgi, gp.reprojection_map_ptr, W, H, None))
        first = True
        for pname in ["REPROJECT"] + RH.PASS_ORDER:
            if pname != "REPROJECT":
                before = RH._surfaces(op)
                T._upload_state(gp, T._oracle_surfaces(op), torch)       # every pass starts from the same state on the device, in the text's inputs and in the oracle
                mask = KJ_RTDGI_PASS[pname] | (0 if first else RH.KEEP)
                first = False
                p = op.params(mask); op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
                gpp = gp.params(mask); gpu.check(gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(gpp), C.byref(gp.out), None))
            if pname not in RH.RAY_FREE:
                continue
            torch.cuda.synchronize()
            written = RH._ref_rtdgi_pass(pname, RH._Frame(op, before, fi, W, H), fc)
            got = T._download_state(gp, list(written.keys()), torch)
            for n, t in written.items():
                r = P.compare(got[n], t.raw, P.fmt_of(n), vector=P.is_vector(n))
                worst[(pname, P.base_name(n))] = r
                assert P.pass_within_bars(pname, r), f"frame {fi} pass {pname} surface {n}: device vs the reference's HLSL text: {r}"
    print(f"device vs reference HLSL text, {scene_name} {W}x{H}, one frame, {len(worst)} surfaces:")
    for k, v in sorted(worst.items()):
        print(f"  {k[0]:>20s} {k[1]:<36s} rel_l2={v['rel_l2']:.2e} mismatch={v['mismatch_frac']:.2e} differ={v.get('differ_frac', float('nan')):.2e}")
    assert len(worst) >= 17, sorted(worst)


@pytest.mark.parametrize("scene_name,W,H", [("city20k", 200, 120), ("city20k", 1920, 1080)])
def test_device_taa_passes_against_the_reference_text(gpu, oracle, device, scene_name, W, H):
    """TaaRenderer::render (taa.rs:41-191), its seven passes one by one: kajiya's compiled HLSL and the MI355X kernel of the same pass read the same inputs -- the
    frame's GI image, depth, reprojection map, the three histories and the intermediates of the passes before it, all the oracle's, uploaded before every device
    pass -- and everything the pass writes is compared under the bars of tests/test_gpu_taa.py (the four images around the probability stage, ill-conditioned in the
    reference itself, under their inlier rule)."""
    import torch
    import test_gpu_parity as T
    import test_ref_hlsl as RH
    R.require_live("the device is compared with the compiled reference text itself")
    if os.environ.get("KJ_TEST_VS_TEXT_EXTENT"):
        W, H = (int(v) for v in os.environ["KJ_TEST_VS_TEXT_EXTENT"].split("x"))
    RH._bind_luts(oracle)
    desc = T._scenes()[scene_name]
    op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
    fcs = T._frame_constants(W, H, 6, T.camera_of(scene_name))
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    inp_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    g = R.extent_inv_extent(W, H)
    FMT = RH.TAA_FORMATS
    HISTORIES = ("taa:0", "taa:1", "taa.velocity:0", "taa.velocity:1", "taa.smooth_var:0", "taa.smooth_var:1")
    worst = {}

    def put(name, raw):
        gp.taa_surface(name, torch.uint8, (-1,)).copy_(torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1).copy()))

    for fi, fc in enumerate(fcs):
        op.frame(fc)                                   # oracle: inputs + reprojection + rtdgi
        before = RH._taa_surfaces(op) if fi else {}
        if fi > 0:
            for n in HISTORIES:
                put(n, op.taa_surface(n, np.uint8, (-1,)))
        op.taa_frame(fc)
        after = RH._taa_surfaces(op)
        gp.dev.frame_begin(fc)
        gp.depth.copy_(torch.from_numpy(op.depth))
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        inp_dev.copy_(torch.from_numpy(op.surface("spatial_filtered_tex", np.int16, (H, W, 4))))
        if fi < 5:
            gp.taa_frame(input_ptr=inp_dev.data_ptr())      # keeps the device's ping-pong parity in step with the oracle's
            torch.cuda.synchronize()
            continue
        out_sfx, hist_sfx = (":0", ":1") if fi % 2 == 0 else (":1", ":0")

        def tex(d, n):
            return R.Tex(d[n].copy(), W, H, FMT[n.split(":")[0]])
        inp = R.Tex(op.surface("spatial_filtered_tex", np.uint8, (-1,)).copy(), W, H, "rgba16f")
        depth, reproj = R.Tex(op.depth, W, H, "r32f"), R.Tex(op.reprojection_map, W, H, "rgba16s")
        written = {}

        def wr(n):
            written[n] = tex(after, n)
            written[n].raw[:] = 0xcd                     # every texel must be written by the pass
            return written[n]
        # (pass name, device pass bit, the text's call); `after` holds what the passes before this one wrote -- the oracle's, which tests/test_ref_hlsl.py holds to the text
        passes = [
            ("reproject_history", 1, lambda: R.run_pass("taa/reproject_history", [tex(before, "taa" + hist_sfx), reproj, depth, wr("reprojected_history_img"), wr("closest_velocity_img")], [g, g], fc, (W, H, 1))),
            ("filter_input", 2, lambda: R.run_pass("taa/filter_input", [inp, depth, wr("filtered_input_img"), wr("filtered_input_deviation_img")], None, fc, (W, H, 1))),
            ("filter_history", 4, lambda: R.run_pass("taa/filter_history", [tex(after, "reprojected_history_img"), wr("filtered_history_img")], [g, g], fc, (W, H, 1))),
            ("input_prob", 8, lambda: R.run_pass("taa/input_prob", [inp, tex(after, "filtered_input_img"), tex(after, "filtered_input_deviation_img"), tex(after, "reprojected_history_img"),
                                                                   tex(after, "filtered_history_img"), reproj, depth, tex(before, "taa.smooth_var" + hist_sfx), tex(before, "taa.velocity" + hist_sfx),
                                                                   wr("input_prob_img")], [g], fc, (W, H, 1))),
            ("filter_prob", 16, lambda: R.run_pass("taa/filter_prob", [tex(after, "input_prob_img"), wr("prob_filtered1_img")], None, fc, (W, H, 1))),
            ("filter_prob2", 32, lambda: R.run_pass("taa/filter_prob2", [tex(after, "prob_filtered1_img"), wr("prob_filtered2_img")], None, fc, (W, H, 1))),
            ("taa", 64, lambda: R.run_pass("taa/taa", [inp, tex(after, "reprojected_history_img"), reproj, tex(after, "closest_velocity_img"), tex(before, "taa.velocity" + hist_sfx), depth,
                                                       tex(before, "taa.smooth_var" + hist_sfx), tex(after, "prob_filtered2_img"),
                                                       wr("taa" + out_sfx), wr("this_frame_output_img"), wr("taa.smooth_var" + out_sfx), wr("taa.velocity" + out_sfx)], [g, g], fc, (W, H, 1))),
        ]
        transients = [n for n in FMT if not n.startswith("taa")]
        first = True
        for pname, bit, run_text in passes:
            for n in transients:                         # the device's pass reads what the text's pass reads
                if n in after:
                    put(n, after[n])
            written.clear()
            run_text()
            gpu.check(gp.L.kj_taa_render_rows(gp.taa, inp_dev.data_ptr(), W, H, gp.reprojection_map_ptr, gp.depth.data_ptr(), W, H, C.byref(gp.taa_out), None,
                                              bit | (0 if first else 0x80000000), 0, H))
            first = False
            torch.cuda.synchronize()
            for n, t in written.items():
                got = gp.taa_surface(n, torch.uint8, (-1,)).cpu().numpy()
                fmt = FMT[n.split(":")[0]]
                if fmt == "r16f":
                    import test_gpu_taa as TT
                    r = P.compare_decoded(TT._decode_r16f(got).reshape(-1, 1), TT._decode_r16f(t.raw.view(np.uint8).reshape(-1)).reshape(-1, 1), atol=1e-3)
                else:
                    r = P.compare(got, t.raw.view(np.uint8).reshape(-1), fmt)
                worst[(pname, n.split(":")[0])] = r
                if n.split(":")[0] in ("filtered_history_img", "input_prob_img", "prob_filtered1_img", "prob_filtered2_img"):      # tests/test_gpu_taa.py: TaaStep's rule for these four
                    ok = r["rel_l2_inliers"] <= P.REL_L2_TOL and r["mismatch_frac"] <= P.MISMATCH_TOL and r["rel_l2"] <= 2e-3 and \
                        r.get("bad_class", 0) <= (int(1e-5 * r.get("n", 0)) if n.startswith("filtered_history_img") else 0)
                else:
                    ok = P.within_bars(r)
                assert ok, f"frame {fi} TAA pass {pname} surface {n}: device vs the reference's HLSL text: {r}"
    print(f"device vs reference HLSL text, TAA, {scene_name} {W}x{H}, one frame, {len(worst)} surfaces:")
    for k, v in sorted(worst.items()):
        print(f"  {k[0]:>20s} {k[1]:<36s} rel_l2={v['rel_l2']:.2e} mismatch={v['mismatch_frac']:.2e} differ={v.get('differ_frac', float('nan')):.2e}")
    assert len(worst) >= 12, sorted(worst)      # everything TaaRenderer::render writes: 8 transients + 4 outputs


@pytest.mark.parametrize("scene_name,W,H", [("cornell", 200, 120), ("cornell", 960, 540)])
def test_device_rtdgi_ray_passes_against_the_reference_text(gpu, oracle, device, scene_name, W, H):
    """`rtdgi validate` and `rtdgi trace` (rtdgi.rs:283-349): diffuse_validate.rgen.hlsl / trace_diffuse.rgen.hlsl with diffuse_trace_common.inc.hlsl, inc/rt.hlsl and, on every
    hit, rt/gbuffer.rchit.hlsl reading the scene tables -- kajiya's text compiled for the CPU, its TraceRay answered by the oracle's scene (the one boundary no pin can
    reach: the query is the Vulkan driver's in the reference) -- against the MI355X's fused ray kernels on the same frame state, a validation frame and a tracing frame.
    The irradiance cache is unbound on both sides (BASELINE configs[0]'s form of the pass). Seven surfaces."""
    import torch
    import test_gpu_parity as T
    import test_ref_hlsl as RH
    from kajiya_amd.abi import KJ_RTDGI_PASS
    R.require_live("the device is compared with the compiled reference text itself")
    if os.environ.get("KJ_TEST_VS_TEXT_EXTENT"):
        W, H = (int(v) for v in os.environ["KJ_TEST_VS_TEXT_EXTENT"].split("x"))
    RH._bind_luts(oracle)
    desc = T._scenes()[scene_name]
    osc = oracle.OracleScene(desc)
    keep = RH._bind_scene(oracle, osc, desc)      # noqa: F841  (the bound tables must outlive the passes)
    op = oracle.OraclePipeline(osc, W, H)
    gp = gpu.GpuPipeline(device, gpu.Scene(device, desc), W, H)
    hw, hh = (W + 1) // 2, (H + 1) // 2
    g = R.extent_inv_extent(W, H)
    sky = R.Tex(op.sky16, 16, 16 * 6, "rgba16f")
    wrc = R.Tex.zeros(1, 1, "rgba16f")
    fcs = T._frame_constants(W, H, 8, T.camera_of(scene_name))
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    worst = {}
    with R.sincos("libm"):       # the oracle's (and the kernels') choice for sampled directions
        for fi, fc in enumerate(fcs):
            op.render_inputs(fc); op.reprojection(fc)
            gp.dev.frame_begin(fc)
            T._sync_inputs(op, gp, torch)
            repro_dev.copy_(torch.from_numpy(op.reprojection_map))
            gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
            if fi < 5:
                op.rtdgi_frame(fc); gp.rtdgi_frame()
                torch.cuda.synchronize()
                T._upload_state(gp, T._oracle_surfaces(op), torch)
                continue
            T._upload_state(gp, T._oracle_surfaces(op), torch)
            op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
            gpu.check(gp.L.kj_rtdgi_reproject(gp.rtdgi, gp.reprojection_map_ptr, W, H, None))
            first = True
            for pname in ("EXTRACT_HALF", "VALIDATE", "TRACE"):
                before = RH._surfaces(op)
                T._upload_state(gp, T._oracle_surfaces(op), torch)
                mask = KJ_RTDGI_PASS[pname] | (0 if first else RH.KEEP)
                first = False
                p = op.params(mask); op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
                gpp = gp.params(mask); gpu.check(gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(gpp), C.byref(gp.out), None))
                if pname == "EXTRACT_HALF":
                    continue
                torch.cuda.synchronize()
                f = RH._Frame(op, before, fi, W, H)
                irc = RH._ircache_bind_set(RH._empty_ircache(), 0)
                if pname == "VALIDATE":          # rtdgi.rs:283-311
                    R.run_pass("rtdgi/diffuse_validate.rgen",
                               [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.wr("rtdgi.reservoir" + f.hist_sfx), f.hist("rtdgi.ray"), f.reprojection_map()] + irc +
                               [wrc, sky, f.wr("rtdgi.radiance" + f.hist_sfx), f.hist("rtdgi.ray_orig"), f.wr("rt_history_validity_pre_input_tex")], [g], fc, (hw, hh, 1))
                else:                            # rtdgi.rs:316-349
                    R.run_pass("rtdgi/trace_diffuse.rgen",
                               [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.reprojection_map()] + irc +
                               [wrc, sky, f.hist("rtdgi.ray_orig"), f.wr("candidate_radiance_tex"), f.wr("candidate_normal_tex"), f.wr("candidate_hit_tex"),
                                f.rd("rt_history_validity_pre_input_tex"), f.wr("rt_history_validity_input_tex")], [g], fc, (hw, hh, 1))
                got = T._download_state(gp, list(f.written.keys()), torch)
                for n, t in f.written.items():
                    r = P.compare(got[n], t.raw, P.fmt_of(n), vector=P.is_vector(n))
                    key = (pname, P.base_name(n))
                    if key not in worst or r["rel_l2"] > worst[key]["rel_l2"]:
                        worst[key] = r
                    assert P.pass_within_bars(pname, r), f"frame {fi} pass {pname} surface {n}: device vs the reference's HLSL text: {r}"
    print(f"device vs reference HLSL text, rtdgi ray passes, {scene_name} {W}x{H}, frames 5-7, {len(worst)} surfaces:")
    for k, v in sorted(worst.items()):
        print(f"  {k[0]:>20s} {k[1]:<36s} rel_l2={v['rel_l2']:.2e} mismatch={v['mismatch_frac']:.2e} differ={v.get('differ_frac', float('nan')):.2e}")
    assert len(worst) >= 7, sorted(worst)
