"""The oracle pinned to the reference's own text. oracle/_ref/libref_hlsl.so is kajiya's HLSL -- read in place from
/root/reference/assets/shaders, rewritten token by token and compiled for the CPU (oracle/ref_hlsl/) -- and these tests run it and the
hand-written oracle (oracle/okj_*.hpp) on the same inputs:

  phase B  every ray-free compute pass of the GI path, dispatched the way the reference's Rust records it (renderers/rtdgi.rs,
           taa.rs, reprojection.rs, half_res.rs: binding order, constants tuple, dispatch extent), against the oracle's pass on the
           oracle's own frame state, under the same bars the GPU-vs-oracle tests use (tests/parity.py: 1e-3 relative L2, <= 0.2 % outliers).

Not GPU tests: this runs wherever the library is (built here, where the reference checkout is; it travels to the GPU box prebuilt)."""
import ctypes as C
import os
import numpy as np
import pytest

import parity as P
import ref_hlsl as R

pytestmark = pytest.mark.skipif(not R.available(), reason="no reference checkout, no prebuilt oracle/_ref/libref_hlsl.so and no recorded outputs under tests/golden/ref_hlsl")

KEEP = 1 << 31
# KJ_REF_HLSL_SCALE=k: every frame extent of the pass-level cases below times k (live runs only; the recorded cases are the unscaled ones). The stored formats are 16 bits
# or fewer for most surfaces, so a last-bit difference in the fp32 arithmetic shows in about one texel in ten thousand: the small extents of the default run find reading
# differences of association and evaluation order, the scaled run (scripts/ref_hlsl_at_scale.sh) is what has the power for rarer ones.
SCALE = int(os.environ.get("KJ_REF_HLSL_SCALE", "1"))


def _scaled(w, h):
    return w * SCALE, h * SCALE


def recorded_case(select):
    """One parametrisation of each test has the reference side's outputs COMMITTED (tests/golden/ref_hlsl/<name>.npz: every image / buffer each pass of the
    reference's text wrote, recorded by scripts/make_ref_hlsl_golden.sh from the live run): there the test -- the oracle against what the reference's text produced --
    also runs where neither the reference checkout nor the compiled library exists (ref_hlsl.golden / replaying). `select(kwargs)` -> file stem or None."""
    import contextlib
    import functools

    def deco(f):
        @functools.wraps(f)
        def w(*a, **k):
            name = select(k)
            with (R.golden(name) if name else contextlib.nullcontext()):
                return f(*a, **k)
        return w
    return deco


@pytest.fixture
def libm_sincos():
    """The passes of this test run from the build of the reference's text whose sin / cos are libm's -- the oracle's choice everywhere but
    the five spiral-tap sites of the rtdgi screen passes, which use the hardware's range reduction (ref_hlsl.sincos)."""
    with R.sincos("libm"):
        yield


def _frame_constants(W, H, n_frames, scene="cornell"):
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    out = []
    for i in range(n_frames):
        if scene == "cornell":
            cam = frame.orbit_camera(i, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01)
        else:
            cam = frame.orbit_camera(i, (W, H), center=(0.0, 6.0, 0.0), radius=60.0, height=14.0, rate=0.004)
        out.append(fs.prepare_frame_constants(cam))
        fs.retire_frame()
    return out


def _surfaces(op):
    out = {}
    for name in list(P.FORMATS.keys()):
        if name.startswith("rtr") or name in ("refl_restir_invalidity_tex", "resolved_tex"):
            continue
        for suffix in ("", ":0", ":1"):
            n = name + suffix
            try:
                out[n] = op.surface(n, np.uint8, (-1,)).copy()
            except KeyError:
                pass
    return out


class _Frame:
    """The oracle's state around one pass, as named textures for the reference pass: inputs from `before`, outputs into copies."""

    def __init__(self, op, before, frame_no, W, H):
        self.op, self.before, self.W, self.H = op, before, W, H
        self.hw, self.hh = (W + 1) // 2, (H + 1) // 2
        self.out_sfx, self.hist_sfx = (":0", ":1") if frame_no % 2 == 0 else (":1", ":0")      # PingPongTemporalResource (renderers/mod.rs:85-102)
        self.written = {}

    def _dims(self, name):
        return (self.W, self.H) if P.base_name(name) in P.FULL_RES else (self.hw, self.hh)

    def rd(self, name):
        w, h = self._dims(name)
        return R.Tex(self.before[name].copy(), w, h, P.fmt_of(name))

    def wr(self, name):
        w, h = self._dims(name)
        t = R.Tex(self.before[name].copy(), w, h, P.fmt_of(name))
        self.written[name] = t
        return t

    def hist(self, key):
        return self.rd(key + self.hist_sfx)

    def out(self, key):
        return self.wr(key + self.out_sfx)

    def out_as_input(self, key):
        return self.rd(key + self.out_sfx)

    # the frame's inputs (not rtdgi surfaces)
    def depth(self):
        return R.Tex(self.op.depth, self.W, self.H, "r32f")

    def gbuffer(self):
        return R.Tex(self.op.gbuffer, self.W, self.H, "rgba32f")

    def geometric_normal(self):
        return R.Tex(self.op.geometric_normal, self.W, self.H, "a2r10g10b10")

    def reprojection_map(self):
        return R.Tex(self.op.reprojection_map, self.W, self.H, "rgba16s")

    def ssao(self):
        return R.Tex(self.op.ssao, self.W, self.H, "r8")


def _ref_rtdgi_pass(pname, f, fc, spatial_passes=2, raytraced=False):
    """Records one rtdgi pass like renderers/rtdgi.rs does. Returns the surfaces it wrote (name -> Tex)."""
    W, H, hw, hh = f.W, f.H, f.hw, f.hh
    g, ho = R.extent_inv_extent(W, H), R.extent_inv_extent(hw, hh)
    if pname == "REPROJECT":                     # rtdgi.rs:143-170
        R.run_pass("rtdgi/fullres_reproject", [f.hist("rtdgi.temporal2"), f.reprojection_map(), f.wr("reprojected_history_tex")], [g], fc, (W, H, 1))
    elif pname == "EXTRACT_HALF":                # rtdgi.rs:189-202, half_res.rs:4-44 (no .constants(): the cbuffer is declared and unused)
        R.run_pass("extract_half_res_ssao", [f.ssao(), f.wr("half_ssao_tex")], None, fc, (hw, hh, 1))
        R.run_pass("extract_half_res_gbuffer_view_normal_rgba8", [f.gbuffer(), f.wr("half_view_normal_tex")], None, fc, (hw, hh, 1))
        R.run_pass("extract_half_res_depth", [f.depth(), f.wr("half_depth_tex")], None, fc, (hw, hh, 1))
    elif pname == "VALIDITY_INTEGRATE":          # rtdgi.rs:352-367
        R.run_pass("rtdgi/temporal_validity_integrate",
                   [f.rd("rt_history_validity_input_tex"), f.hist("rtdgi.invalidity"), f.reprojection_map(), f.rd("half_view_normal_tex"), f.rd("half_depth_tex"),
                    f.out("rtdgi.invalidity")], [g, ho], fc, (hw, hh, 1))
    elif pname == "RESTIR_TEMPORAL":             # rtdgi.rs:369-397
        R.run_pass("rtdgi/restir_temporal",
                   [f.rd("half_view_normal_tex"), f.depth(), f.rd("candidate_radiance_tex"), f.rd("candidate_normal_tex"), f.rd("candidate_hit_tex"),
                    f.hist("rtdgi.radiance"), f.hist("rtdgi.ray_orig"), f.hist("rtdgi.ray"), f.hist("rtdgi.reservoir"), f.reprojection_map(),
                    f.hist("rtdgi.hit_normal"), f.hist("rtdgi.candidate"), f.out_as_input("rtdgi.invalidity"),
                    f.out("rtdgi.radiance"), f.out("rtdgi.ray_orig"), f.out("rtdgi.ray"), f.out("rtdgi.hit_normal"), f.out("rtdgi.reservoir"), f.out("rtdgi.candidate"),
                    f.wr("temporal_reservoir_packed_tex")], [g], fc, (hw, hh, 1))
    elif pname == "RESTIR_SPATIAL":              # rtdgi.rs:402-474: two passes, ping-ponging reservoir_output_tex0 / 1
        names = ["reservoir_output_tex0", "reservoir_output_tex1"]
        reservoir_in = f.out_as_input("rtdgi.reservoir")
        bounced_in = R.Tex(f.before["rtdgi.radiance" + f.out_sfx].copy(), hw, hh, "rgba16f")           # `bounced_radiance_input_tex = &radiance_tex` for pass 0
        for idx in range(spatial_passes):
            out = f.wr(names[idx & 1])
            bounced_out = R.Tex.zeros(hw, hh, "r11g11b10f")
            R.run_pass("rtdgi/restir_spatial",
                       [reservoir_in, bounced_in, f.rd("half_view_normal_tex"), f.rd("half_depth_tex"), f.depth(), f.rd("half_ssao_tex"),
                        f.rd("temporal_reservoir_packed_tex"), f.rd("reprojected_history_tex"), out, bounced_out],
                       [g, ho, np.uint32(idx), np.uint32(1 if idx + 1 == spatial_passes else 0), np.uint32(1 if raytraced else 0)], fc, (hw, hh, 1))
            reservoir_in, bounced_in = R.Tex(out.raw.copy(), hw, hh, "rg32ui"), bounced_out
        if raytraced:                            # "restir check" (rtdgi.rs:478-494): restir_check.rgen.hlsl on the last reservoir image
            R.run_pass("rtdgi/restir_check.rgen", [f.rd("half_depth_tex"), f.rd("temporal_reservoir_packed_tex"), out], [g], fc, (hw, hh, 1))
    elif pname == "RESTIR_RESOLVE":              # rtdgi.rs:503-524
        last = "reservoir_output_tex%d" % ((spatial_passes - 1) & 1) if spatial_passes else "rtdgi.reservoir" + f.out_sfx
        R.run_pass("rtdgi/restir_resolve",
                   [f.out_as_input("rtdgi.radiance"), f.rd(last), f.gbuffer(), f.depth(), f.rd("half_view_normal_tex"), f.rd("half_depth_tex"), f.ssao(),
                    f.rd("candidate_radiance_tex"), f.rd("candidate_hit_tex"), f.rd("temporal_reservoir_packed_tex"), R.Tex.zeros(hw, hh, "r11g11b10f"),
                    f.wr("irradiance_output_tex")], [g, g], fc, (W, H, 1))
    elif pname == "TEMPORAL_FILTER":             # rtdgi.rs:71-115
        R.run_pass("rtdgi/temporal_filter",
                   [f.rd("irradiance_output_tex"), f.rd("reprojected_history_tex"), f.hist("rtdgi.temporal2_var"), f.reprojection_map(), f.out_as_input("rtdgi.invalidity"),
                    f.wr("temporal_filtered_tex"), f.out("rtdgi.temporal2"), f.out("rtdgi.temporal2_var")], [g, g], fc, (W, H, 1))
    elif pname == "SPATIAL_FILTER":              # rtdgi.rs:117-141
        R.run_pass("rtdgi/spatial_filter",
                   [f.rd("temporal_filtered_tex"), f.depth(), f.ssao(), f.geometric_normal(), f.wr("spatial_filtered_tex")], [g], fc, (W, H, 1))
    else:
        raise KeyError(pname)
    return f.written


RAY_FREE = ["REPROJECT", "EXTRACT_HALF", "VALIDITY_INTEGRATE", "RESTIR_TEMPORAL", "RESTIR_SPATIAL", "RESTIR_RESOLVE", "TEMPORAL_FILTER", "SPATIAL_FILTER"]
PASS_ORDER = ["EXTRACT_HALF", "VALIDATE", "TRACE", "VALIDITY_INTEGRATE", "RESTIR_TEMPORAL", "RESTIR_SPATIAL", "RESTIR_RESOLVE", "TEMPORAL_FILTER", "SPATIAL_FILTER"]


def _check(r, what, pname=""):
    """The bars of the GPU-vs-oracle tests -- and, on top of them, what this comparison actually delivers: with the oracle and the
    compiled reference text sharing the definitions of DESIGN.md §4 for everything HLSL leaves to the implementation, the two are
    the same arithmetic in the same order, and their outputs are identical number for number (the only bit patterns that may differ
    are the sign of a zero out of max(0.0, -0.0), which C leaves open, and NaN payloads). A texel that differs is a difference in reading."""
    assert P.pass_within_bars(pname, r), f"{what}: reference HLSL vs oracle {r}"
    assert r["differ_frac"] == 0.0, f"{what}: within the bars but not byte-identical: {r}"


def _bind_luts(oracle):
    """Descriptor set 1 (default_world_renderer.rs:20-40): the BRDF-FG LUT and the blue-noise image, the oracle's copies of both."""
    R.set_bindless(0, R.Tex(oracle.brdf_lut(), 64, 64, "rgba16f"))
    R.set_bindless(1, R.Tex(oracle.blue_noise(), 256, 256, "rgba8"))


def _rtdgi_chain(oracle, scene_name, W, H, n_frames, warmup, report=None, spatial_passes=2, raytraced=False):
    from kajiya_amd import scenes
    _bind_luts(oracle)
    from kajiya_amd.abi import KJ_RTDGI_PASS
    desc = scenes.cornell_box() if scene_name == "cornell" else scenes.procedural_city(seed=1234, target_tris=20000)
    osc = oracle.OracleScene(desc)
    op = oracle.OraclePipeline(osc, W, H)
    if raytraced:
        op._ref_keep = _bind_scene(oracle, osc, desc)      # restir_check.rgen.hlsl traces rays
    op.L.okj_rtdgi_set_options(op.rtdgi, spatial_passes)
    op.L.okj_rtdgi_set_raytraced_visibility(op.rtdgi, int(raytraced))
    fcs = _frame_constants(W, H, n_frames, scene_name)
    worst = {}
    for fi, fc in enumerate(fcs):
        op.render_inputs(fc); op.reprojection(fc)
        if fi < warmup:
            op.rtdgi_frame(fc)
            continue
        before = _surfaces(op)
        op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
        first = True
        for pname in ["REPROJECT"] + PASS_ORDER:
            if pname != "REPROJECT":
                before = _surfaces(op)
                mask = KJ_RTDGI_PASS[pname] | (0 if first else KEEP)
                first = False
                p = op.params(mask)
                op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
            if pname not in RAY_FREE:
                continue
            after = _surfaces(op)
            written = _ref_rtdgi_pass(pname, _Frame(op, before, fi, W, H), fc, spatial_passes=spatial_passes, raytraced=raytraced)
            for n, t in written.items():
                r = P.compare(t.raw, after[n], P.fmt_of(n), vector=P.is_vector(n))
                key = (pname, P.base_name(n))
                if key not in worst or r["rel_l2"] > worst[key]["rel_l2"]:
                    worst[key] = r
                if report is not None:
                    report.append((fi, pname, n, r))
                else:
                    _check(r, f"frame {fi} pass {pname} surface {n}", pname)
            # and nothing else changed on the oracle's side that the reference pass does not write
            for n in after:
                if n not in written and not np.array_equal(after[n], before[n]):
                    raise AssertionError(f"frame {fi} pass {pname}: the oracle wrote {n}, the reference pass does not")
    return worst


@pytest.mark.parametrize("scene_name,W,H,passes,raytraced", [("cornell", 64, 64, 2, False), ("cornell", 72, 40, 2, False), ("city", 96, 56, 2, False),
                                                             ("cornell", 40, 40, 1, False), ("cornell", 40, 40, 3, True)])
@recorded_case(lambda k: "rtdgi_screen_passes_cornell_64" if (k["scene_name"], k["W"], k["passes"], k["raytraced"]) == ("cornell", 64, 2, False) else None)
def test_rtdgi_ray_free_passes_reference_hlsl_vs_oracle(oracle, scene_name, W, H, passes, raytraced):
    """fullres_reproject, the half-res extracts, validity integrate, temporal + N x spatial ReSTIR, resolve, temporal and spatial filter:
    frames 5 (tracing), 6 (validation: frame_index % 3 == 0) and 7 after five warm-up frames. Extents that are not multiples of the
    8x8 group, a moving camera over a 20 k-triangle city, 1 / 2 / 3 spatial passes, occlusion_raymarch_importance_only on and off."""
    W, H = _scaled(W, H)
    worst = _rtdgi_chain(oracle, scene_name, W, H, n_frames=8, warmup=5, spatial_passes=passes, raytraced=raytraced)
    assert len(worst) >= (18 if passes >= 2 else 17), sorted(worst)


@pytest.mark.parametrize("scene_name,W,H", [("cornell", 64, 64), ("city", 104, 60)])
@recorded_case(lambda k: "reprojection_map_cornell_64" if k["scene_name"] == "cornell" else None)
def test_reprojection_map_reference_hlsl_vs_oracle(oracle, scene_name, W, H):
    """calculate_reprojection_map.hlsl as renderers/reprojection.rs:6-52 records it, on a moving camera, against the oracle's map."""
    W, H = _scaled(W, H)
    from kajiya_amd import scenes
    desc = scenes.cornell_box() if scene_name == "cornell" else scenes.procedural_city(seed=1234, target_tris=20000)
    op = oracle.OraclePipeline(oracle.OracleScene(desc), W, H)
    for fi, fc in enumerate(_frame_constants(W, H, 4, scene_name)):
        op.render_inputs(fc)
        prev_depth = op.prev_depth.copy()
        op.reprojection(fc)
        out = R.Tex.zeros(W, H, "rgba16s")
        out.raw[:] = 0xcd
        R.run_pass("calculate_reprojection_map", [R.Tex(op.depth, W, H, "r32f"), R.Tex(op.geometric_normal, W, H, "a2r10g10b10"), R.Tex(prev_depth, W, H, "r32f"),
                                                  R.Tex(op.velocity, W, H, "rgba16f"), out], [R.extent_inv_extent(W, H)], fc, (W, H, 1))
        a, b = out.raw.view(np.int16).reshape(-1, 4), op.reprojection_map.reshape(-1, 4)
        assert np.array_equal(a, b), (fi, int((a != b).any(axis=1).sum()), a[(a != b).any(axis=1)][:4], b[(a != b).any(axis=1)][:4])


# ---------------------------------------------------------------------------------------------------------------- TAA (renderers/taa.rs:41-191)
TAA_FORMATS = {"taa": "rgba16f", "taa.velocity": "rg16f", "taa.smooth_var": "rgba16f", "reprojected_history_img": "rgba16f", "closest_velocity_img": "rg16f",
               "filtered_input_img": "rgba16f", "filtered_input_deviation_img": "rgba16f", "filtered_history_img": "rgba16f", "input_prob_img": "r16f",
               "prob_filtered1_img": "r16f", "prob_filtered2_img": "r16f", "this_frame_output_img": "rgba16f"}


def _taa_surfaces(op):
    out = {}
    for name in TAA_FORMATS:
        for suffix in ("", ":0", ":1"):
            try:
                out[name + suffix] = op.taa_surface(name + suffix, np.uint8, (-1,)).copy()
            except (KeyError, AttributeError):
                pass
    return out


@pytest.mark.parametrize("W,H", [(64, 64), (72, 40)])
@recorded_case(lambda k: "taa_64" if k["W"] == 64 else None)
def test_taa_passes_reference_hlsl_vs_oracle(oracle, W, H):
    """The seven TAA passes on the GI output, frames 0..5 (frame 0: empty history), each reference pass fed the oracle's surfaces."""
    W, H = _scaled(W, H)
    from kajiya_amd import scenes
    _bind_luts(oracle)
    op = oracle.OraclePipeline(oracle.OracleScene(scenes.cornell_box()), W, H)
    g = R.extent_inv_extent(W, H)
    compared = set()
    for fi, fc in enumerate(_frame_constants(W, H, 6)):
        op.render_inputs(fc); op.reprojection(fc); op.rtdgi_frame(fc)
        before = _taa_surfaces(op) if fi else {}
        op.taa_frame(fc)
        after = _taa_surfaces(op)
        out_sfx, hist_sfx = (":0", ":1") if fi % 2 == 0 else (":1", ":0")

        def tex(d, n):
            raw = d.get(n)
            if raw is None:
                raw = np.zeros_like(after[n])          # frame 0: a temporal that does not exist yet is created cleared
            return R.Tex(raw.copy(), W, H, TAA_FORMATS[n.split(":")[0]])
        inp = R.Tex(op.surface("spatial_filtered_tex", np.uint8, (-1,)).copy(), W, H, "rgba16f")
        depth, reproj = R.Tex(op.depth, W, H, "r32f"), R.Tex(op.reprojection_map, W, H, "rgba16s")
        written = {}

        def wr(n):
            written[n] = tex(after, n)
            written[n].raw[:] = 0xcd                     # every texel must be written by the pass
            return written[n]
        R.run_pass("taa/reproject_history", [tex(before, "taa" + hist_sfx), reproj, depth, wr("reprojected_history_img"), wr("closest_velocity_img")], [g, g], fc, (W, H, 1))
        R.run_pass("taa/filter_input", [inp, depth, wr("filtered_input_img"), wr("filtered_input_deviation_img")], None, fc, (W, H, 1))
        R.run_pass("taa/filter_history", [tex(after, "reprojected_history_img"), wr("filtered_history_img")], [g, g], fc, (W, H, 1))
        R.run_pass("taa/input_prob", [inp, tex(after, "filtered_input_img"), tex(after, "filtered_input_deviation_img"), tex(after, "reprojected_history_img"),
                                      tex(after, "filtered_history_img"), reproj, depth, tex(before, "taa.smooth_var" + hist_sfx), tex(before, "taa.velocity" + hist_sfx),
                                      wr("input_prob_img")], [g], fc, (W, H, 1))
        R.run_pass("taa/filter_prob", [tex(after, "input_prob_img"), wr("prob_filtered1_img")], None, fc, (W, H, 1))
        R.run_pass("taa/filter_prob2", [tex(after, "prob_filtered1_img"), wr("prob_filtered2_img")], None, fc, (W, H, 1))
        R.run_pass("taa/taa", [inp, tex(after, "reprojected_history_img"), reproj, tex(after, "closest_velocity_img"), tex(before, "taa.velocity" + hist_sfx), depth,
                               tex(before, "taa.smooth_var" + hist_sfx), tex(after, "prob_filtered2_img"),
                               wr("taa" + out_sfx), wr("this_frame_output_img"), wr("taa.smooth_var" + out_sfx), wr("taa.velocity" + out_sfx)], [g, g], fc, (W, H, 1))
        for n, t in written.items():
            r = P.compare(t.raw, after[n], TAA_FORMATS[n.split(":")[0]])
            _check(r, f"frame {fi} TAA surface {n}")
            compared.add(n.split(":")[0])
    assert compared == set(TAA_FORMATS), compared


# ---------------------------------------------------------------------------------------------------------------- irradiance cache maintenance
IRC_BUFFERS = ["meta", "grid_meta0", "grid_meta1", "entry_cell", "spatial", "irradiance", "aux", "life", "pool", "entry_indirection", "reposition_proposal",
               "reposition_proposal_count", "entry_occupancy"]
MAX_ENTRIES = 65536


def _irc_snapshot(op):
    s = {n: op.ircache_buffer(n, np.uint8).copy() for n in IRC_BUFFERS}
    s["entry_occupancy"] = s["entry_occupancy"][:MAX_ENTRIES * 4]      # the oracle pads its copy by one group of 64 it never scans
    return s


def _irc_host_state(op):
    st = (C.c_int32 * 3)()
    op.L.okj_ircache_host_state(op.ircache, st)
    return dict(parity=st[0], initialized=bool(st[1]), cur=st[2])


def _irc_diff(got, ref, what):
    for n in ref:
        if n in got and not np.array_equal(got[n], ref[n]):
            a, b = got[n].view(np.uint32), ref[n].view(np.uint32)
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{what}: buffer {n}: {bad.size} of {a.size} dwords differ, first at {bad[:6]}: {a[bad[:6]]} vs {b[bad[:6]]}")


def _ref_ircache_prepare(s, host, fc):
    """IrcacheRenderer::prepare (renderers/ircache.rs:168-350) recorded on the buffers of snapshot `s` (modified in place)."""
    B = {n: R.Buf(s[n]) for n in s}
    a, b = ("grid_meta0", "grid_meta1") if host["parity"] == 0 else ("grid_meta1", "grid_meta0")       # :235-240
    if not host["initialized"]:
        R.run_pass("ircache/clear_ircache_pool", [B["pool"], B["life"]], None, fc, (MAX_ENTRIES, 1, 1))
    else:
        R.run_pass("ircache/scroll_cascades", [B[a], B[b], B["entry_cell"], B["irradiance"], B["life"], B["pool"], B["meta"]], None, fc, (32, 32, 32 * 12))
        a, b = b, a
    args = R.Buf(np.zeros(8, np.uint32))
    R.run_pass("ircache/prepare_age_dispatch_args", [B["meta"], args], None, fc, (1, 1, 1))
    groups = args.raw.view(np.uint32)[:3].copy()
    assert groups[1] == 1 and groups[2] == 1
    occupancy = R.Buf(np.zeros(MAX_ENTRIES, np.uint32))                      # a transient in the reference: size_of::<u32>() * MAX_ENTRIES (ircache.rs:307-310)
    R.run_pass("ircache/age_ircache_entries", [B["meta"], B[a], B["entry_cell"], B["life"], B["pool"], B["spatial"], B["reposition_proposal"],
                                               B["reposition_proposal_count"], B["irradiance"], occupancy], None, fc, (int(groups[0]) * 64, 1, 1))
    # inclusive_prefix_scan_u32_1m (renderers/prefix_scan.rs:10-42): three passes over 1 Mi elements of a 64 Ki-element buffer
    SEG = 1024
    R.run_pass("prefix_scan/inclusive_prefix_scan", [occupancy], None, None, (SEG * SEG // 2, 1, 1))
    segment_sum = R.Buf(np.zeros(SEG, np.uint32))
    R.run_pass("prefix_scan/inclusive_prefix_scan_segments", [occupancy, segment_sum], None, None, (SEG // 2, 1, 1))
    R.run_pass("prefix_scan/inclusive_prefix_scan_merge", [occupancy, segment_sum], None, None, (SEG * SEG // 2, 1, 1))
    R.run_pass("ircache/ircache_compact_entries", [B["meta"], B["life"], occupancy, B["entry_indirection"]], None, fc, (int(groups[0]) * 64, 1, 1))
    s["entry_occupancy"] = occupancy.raw
    return 0 if a == "grid_meta0" else 1


def test_ircache_maintenance_reference_hlsl_vs_oracle(oracle):
    """Every ray-free pass of the irradiance cache -- clear pool / scroll cascades, age, the three-pass prefix scan, compact
    (IrcacheRenderer::prepare), dispatch args + reset (head of trace_irradiance) and the SH sum-up -- from the reference's text on the
    oracle's live cache state, frame after frame under a moving camera: every buffer byte for byte. The passes that push onto the
    free list with atomics run in ascending thread order on both sides (any order is a legal schedule: ref_set_linear_order)."""
    R.require_live("the cache's buffers are 70 MB per call: not recorded")
    from kajiya_amd import scenes, frame
    _bind_luts(oracle)
    L = R.lib()
    L.ref_set_linear_order(1)
    try:
        W = H = 64
        op = oracle.OraclePipeline(oracle.OracleScene(scenes.cornell_box()), W, H, use_ircache=True)
        fs = frame.FrameState((W, H))
        fs.ircache_enabled = True
        seen_scroll = 0
        for fi in range(7):
            fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.03))
            seen_scroll += sum(abs(int(v)) for c in range(12) for v in fc.ircache_cascades[c].voxels_scrolled_this_frame[:3]) if fi else 0
            op.render_inputs(fc); op.reprojection(fc)
            # --- prepare
            s0, host = _irc_snapshot(op), _irc_host_state(op)
            op.L.okj_ircache_prepare(op.ircache, C.byref(fc))
            s1, host1 = _irc_snapshot(op), _irc_host_state(op)
            cur = _ref_ircache_prepare(s0, host, fc)
            assert cur == host1["cur"], (fi, cur, host1)
            live = "grid_meta%d" % cur
            # the buffer scrolled OUT of is dead until the next frame overwrites it; everything else must match
            _irc_diff({n: v for n, v in s0.items() if n != "grid_meta%d" % (1 - cur)}, s1, f"frame {fi} prepare")
            assert s1["meta"].view(np.uint32)[2] > 0 or fi == 0, "no live entries: the test would be vacuous"
            # --- head of trace_irradiance: dispatch args + reset (ircache.rs:369-394)
            s = {n: v.copy() for n, v in s1.items()}
            B = {n: R.Buf(s[n]) for n in s}
            args = R.Buf(np.zeros(16, np.uint32))
            R.run_pass("ircache/prepare_trace_dispatch_args", [B["meta"], args], None, fc, (1, 1, 1))
            g = args.raw.view(np.uint32)
            R.run_pass("ircache/reset_entry", [B["life"], B["meta"], B["irradiance"], B["aux"], B["entry_indirection"]], None, fc, (int(g[8]) * 64, 1, 1))
            op.L.okj_ircache_prepare_and_reset(op.ircache)
            _irc_diff(s, _irc_snapshot(op), f"frame {fi} dispatch args + reset")
            # --- the frame's ray work on the oracle, then the sum-up
            op.L.okj_ircache_trace_irradiance(op.ircache, C.byref(fc), op.scene.h, op.sky16.ctypes.data, 16)
            op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
            s = _irc_snapshot(op)
            B = {n: R.Buf(s[n]) for n in s}
            # dispatched with the `pending` args buffer written at the head of trace_irradiance (ircache.rs:487-506), not recomputed
            R.run_pass("ircache/sum_up_irradiance", [B["life"], B["meta"], B["irradiance"], B["aux"], B["entry_indirection"]], None, fc, (int(g[8]) * 64, 1, 1))
            op.ircache_sum_up(fc)
            _irc_diff(s, _irc_snapshot(op), f"frame {fi} sum-up")
            p = op.params()
            op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))      # the lookups that allocate next frame's entries
            fs.retire_frame()
        assert seen_scroll > 0, "the camera never crossed a cell: scroll_cascades was not exercised"
    finally:
        L.ref_set_linear_order(0)


# ---------------------------------------------------------------------------------------------------------------- the ray passes
def _bind_scene(oracle, osc, desc):
    """Descriptor sets 1-3 of the ray-tracing passes from the oracle's scene: `meshes` + `vertices` (inc/bindless.hlsl), the material maps
    behind the three LUTs of `bindless_textures[]` + `bindless_texture_sizes`, per-instance emissive multipliers, triangle lights; and
    TraceRay's intersection query."""
    L = oracle.lib()
    n_mesh, n_inst, n_map = L.okj_scene_mesh_count(osc.h), L.okj_scene_instance_count(osc.h), L.okj_scene_map_count(osc.h)
    meshes = np.zeros(n_mesh * 7, np.uint32)
    vertices = np.zeros(L.okj_scene_vertex_buffer_bytes(osc.h), np.uint8)
    emissive = np.zeros(max(1, n_inst), np.float32)
    counts = np.array([len(m.materials) for m in desc.meshes], np.uint32)
    L.okj_scene_export_tables(osc.h, meshes.ctypes.data, vertices.ctypes.data, emissive.ctypes.data, counts.ctypes.data, 3)
    R.set_named("meshes", R.Buf(meshes)); R.set_named("vertices", R.Buf(vertices)); R.set_named("instance_dynamic_parameters_dyn", R.Buf(emissive))
    R.set_named("triangle_lights_dyn", R.Buf(np.zeros(12, np.float32)))
    sizes = np.zeros((3 + n_map, 4), np.float32)
    sizes[0], sizes[1] = (64, 64, 1 / 64, 1 / 64), (256, 256, 1 / 256, 1 / 256)
    keep = []
    for i in range(n_map):
        rgba, w, h, mips, tex = (C.c_uint8 * 4)(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p()
        L.okj_scene_map_info(osc.h, i, rgba, C.byref(w), C.byref(h), C.byref(mips), C.byref(tex))
        assert mips.value == 0, "this harness binds placeholder (1x1) material maps only"
        t = R.Tex(np.array(list(rgba), np.uint8), 1, 1, "rgba8")
        keep.append(t)
        R.set_bindless(3 + i, t)
        sizes[3 + i] = (1, 1, 1, 1)
    R.set_named("bindless_texture_sizes", R.Buf(sizes))
    R.set_trace_hook(L.okj_ref_trace_hook_ptr(), osc.h)
    return keep


def _ircache_bind_set(bufs, cur):
    """IrcacheRenderState::bind_mut (renderers/ircache.rs:59-78): nine buffers in this order."""
    return [R.Buf(bufs[n]) for n in ("meta", "pool", "reposition_proposal", "reposition_proposal_count", "grid_meta%d" % cur, "entry_cell", "spatial", "irradiance", "life")]


def _empty_ircache():
    cells = 12 * 32 * 32 * 32
    z = lambda n: np.zeros(n, np.uint32)
    return {"meta": z(8), "pool": np.arange(MAX_ENTRIES, dtype=np.uint32), "reposition_proposal": z(MAX_ENTRIES * 4), "reposition_proposal_count": z(MAX_ENTRIES),
            "grid_meta0": z(cells * 2), "entry_cell": z(MAX_ENTRIES), "spatial": z(MAX_ENTRIES * 4), "irradiance": z(MAX_ENTRIES * 12), "life": np.full(MAX_ENTRIES, 0x8000001, np.uint32)}


@pytest.mark.parametrize("W,H", [(64, 64), (72, 40)])
@recorded_case(lambda k: "rtdgi_ray_passes_cornell_64" if k["W"] == 64 else None)
def test_rtdgi_ray_passes_reference_hlsl_vs_oracle(oracle, libm_sincos, W, H):
    """`rtdgi validate` and `rtdgi trace` from the reference's own text -- diffuse_validate.rgen.hlsl / trace_diffuse.rgen.hlsl with
    diffuse_trace_common.inc.hlsl, candidate_ray_dir.hlsl, inc/rt.hlsl, and on every hit rt/gbuffer.rchit.hlsl reading the scene tables --
    against the oracle's passes on the oracle's frame state. Only the intersection query itself (the driver's in the reference) is the
    oracle's. The irradiance cache is bound empty, as the oracle's null lookup hook models it (BASELINE configs[0])."""
    W, H = _scaled(W, H)
    from kajiya_amd import scenes
    from kajiya_amd.abi import KJ_RTDGI_PASS
    _bind_luts(oracle)
    desc = scenes.cornell_box()
    osc = oracle.OracleScene(desc)
    keep = _bind_scene(oracle, osc, desc)
    op = oracle.OraclePipeline(osc, W, H)
    hw, hh = (W + 1) // 2, (H + 1) // 2
    g = R.extent_inv_extent(W, H)
    sky = R.Tex(op.sky16, 16, 16 * 6, "rgba16f")
    wrc = R.Tex.zeros(1, 1, "rgba16f")
    compared = set()
    for fi, fc in enumerate(_frame_constants(W, H, 8)):
        op.render_inputs(fc); op.reprojection(fc)
        if fi < 4:
            op.rtdgi_frame(fc)
            continue
        op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
        first = True
        for pname in ("EXTRACT_HALF", "VALIDATE", "TRACE"):
            before = _surfaces(op)
            mask = KJ_RTDGI_PASS[pname] | (0 if first else KEEP)
            first = False
            p = op.params(mask)
            op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
            if pname == "EXTRACT_HALF":
                continue
            after = _surfaces(op)
            f = _Frame(op, before, fi, W, H)
            irc = _ircache_bind_set(_empty_ircache(), 0)
            if pname == "VALIDATE":          # rtdgi.rs:283-311
                R.run_pass("rtdgi/diffuse_validate.rgen",
                           [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.wr("rtdgi.reservoir" + f.hist_sfx), f.hist("rtdgi.ray"), f.reprojection_map()] + irc +
                           [wrc, sky, f.wr("rtdgi.radiance" + f.hist_sfx), f.hist("rtdgi.ray_orig"), f.wr("rt_history_validity_pre_input_tex")], [g], fc, (hw, hh, 1))
            else:                            # rtdgi.rs:316-349
                R.run_pass("rtdgi/trace_diffuse.rgen",
                           [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.reprojection_map()] + irc +
                           [wrc, sky, f.hist("rtdgi.ray_orig"), f.wr("candidate_radiance_tex"), f.wr("candidate_normal_tex"), f.wr("candidate_hit_tex"),
                            f.rd("rt_history_validity_pre_input_tex"), f.wr("rt_history_validity_input_tex")], [g], fc, (hw, hh, 1))
            for n, t in f.written.items():
                r = P.compare(t.raw, after[n], P.fmt_of(n), vector=P.is_vector(n))
                assert P.pass_within_bars(pname, r), f"frame {fi} pass {pname} surface {n}: reference HLSL vs oracle {r}"
                compared.add((pname, P.base_name(n), r["differ_frac"] == 0.0))
                if r["differ_frac"] != 0.0:
                    a, b = P.decode(t.raw, P.fmt_of(n)), P.decode(after[n], P.fmt_of(n))
                    bad = np.nonzero((a != b).any(axis=-1))[0]
                    print(f"frame {fi} {pname} {n}: {bad.size} texels differ, e.g. texel {bad[:3]}: ref {a[bad[:3]]} oracle {b[bad[:3]]}")
            for n in after:
                if n not in f.written and not np.array_equal(after[n], before[n]):
                    raise AssertionError(f"frame {fi} pass {pname}: the oracle wrote {n}, the reference pass does not")
    print(sorted(compared))
    assert len({c[:2] for c in compared}) >= 7, sorted(compared)


@recorded_case(lambda k: "sun_shadow_mask_and_reference_pt")
def test_sun_shadow_mask_and_reference_pt_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """rt/trace_sun_shadow_mask.rgen.hlsl (renderers/shadows.rs:10-40) and rt/reference_path_trace.rgen.hlsl (renderers/reference.rs:8-26; up to
    17 segments per path through rt/gbuffer.rchit.hlsl, the layered BRDF's sampling, Russian roulette, the sun and the sky) from the
    reference's text against the oracle: the mask bit for bit; the path tracer per pixel under the bars of the GPU-vs-oracle test
    (tests/test_gpu_reference_pt.py: <= 2 % of the one-sample pixels off by more than 1e-3 -- the two use different sin / cos range
    reductions for sampled directions, and a path that flips a lobe choice is unrelated afterwards), means within 1 %."""
    from kajiya_amd import scenes
    _bind_luts(oracle)
    desc = scenes.cornell_box()
    osc = oracle.OracleScene(desc)
    keep = _bind_scene(oracle, osc, desc)
    W, H = _scaled(64, 48)
    op = oracle.OraclePipeline(osc, W, H)
    acc_ref, acc_o = R.Tex.zeros(W, H, "rgba32f"), np.zeros((H, W, 4), np.float32)
    worst = 0.0
    for fi, fc in enumerate(_frame_constants(W, H, 6)):
        op.render_inputs(fc)
        mask_o = op.sun_shadow_mask(fc)
        mask = R.Tex.zeros(W, H, "r8")
        R.run_pass("rt/trace_sun_shadow_mask.rgen", [R.Tex(op.depth, W, H, "r32f"), R.Tex(op.geometric_normal, W, H, "a2r10g10b10"), mask], None, fc, (W, H, 1))
        assert np.array_equal(mask.raw.reshape(H, W), mask_o), (fi, int((mask.raw.reshape(H, W) != mask_o).sum()))
        assert 0 < (mask_o == 0).mean() < 1, "a mask without both lit and shadowed pixels tests little"
        one_ref, one_o = R.Tex.zeros(W, H, "rgba32f"), np.zeros((H, W, 4), np.float32)
        R.run_pass("rt/reference_path_trace.rgen", [one_ref], None, fc, (W, H, 1))
        R.run_pass("rt/reference_path_trace.rgen", [acc_ref], None, fc, (W, H, 1))
        oracle.reference_path_trace(osc, fc, one_o)
        oracle.reference_path_trace(osc, fc, acc_o)
        a = one_ref.raw.view(np.float32).reshape(H, W, 4)
        assert np.isfinite(a).all() and (a[..., 3] == one_o[..., 3]).all()
        err = np.abs(a[..., :3] - one_o[..., :3]).max(axis=-1) / (1e-3 + np.abs(one_o[..., :3]).max(axis=-1))
        frac = float((err > 1e-3).mean())
        worst = max(worst, frac)
        assert frac < 0.02, f"frame {fi}: {frac:.4f} of the one-sample pixels differ by more than 1e-3"
    a = acc_ref.raw.view(np.float32).reshape(H, W, 4)
    assert (a[..., 3] == 6).all() and (acc_o[..., 3] == 6).all()
    assert np.allclose(a[..., :3].mean(axis=(0, 1)), acc_o[..., :3].mean(axis=(0, 1)), rtol=0.01), (a[..., :3].mean(axis=(0, 1)), acc_o[..., :3].mean(axis=(0, 1)))
    print(f"reference PT from the reference's text vs oracle: worst one-sample mismatch fraction {worst:.4f}")


def _numbers_equal(a, b):
    """byte-identical, or equal as fp32 numbers (+0 == -0; NaN == NaN)"""
    if np.array_equal(a, b):
        return True
    fa, fb = a.view(np.float32), b.view(np.float32)
    same = (a.view(np.uint32) == b.view(np.uint32)) | ((fa == fb)) | (np.isnan(fa) & np.isnan(fb))
    return bool(same.all())


def test_ircache_ray_passes_and_live_lookups_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """The irradiance cache's three ray passes (trace_accessibility / ircache_validate / trace_irradiance .rgen.hlsl with
    ircache_trace_common.inc.hlsl and the PRECISE ircache/lookup.hlsl inside them) and the rtdgi validate + trace passes with the LIVE cache
    bound (ircache/lookup.hlsl allocating entries, refreshing lives, voting on positions) from the reference's own text, frame after
    frame, against the oracle on the oracle's cache state -- in the reference's own racy semantics, executed in ascending thread order on
    both sides (one oracle thread; ref_set_linear_order). Every cache buffer and every rtdgi surface must come out equal number for
    number, apart from texels / slots where the two sin / cos range reductions round a sampled direction differently (counted, <= 0.2 %)."""
    R.require_live("the cache's buffers are 70 MB per call: not recorded")
    from kajiya_amd import scenes, frame
    from kajiya_amd.abi import KJ_RTDGI_PASS
    _bind_luts(oracle)
    L = R.lib()
    L.ref_set_linear_order(1)
    n_threads = oracle.lib().okj_get_max_threads()
    oracle.lib().okj_set_threads(1)
    try:
        W, H = _scaled(64, 48)
        hw, hh = W // 2, H // 2
        desc = scenes.cornell_box()
        osc = oracle.OracleScene(desc)
        keep = _bind_scene(oracle, osc, desc)
        op = oracle.OraclePipeline(osc, W, H, use_ircache=True)
        fs = frame.FrameState((W, H))
        fs.ircache_enabled = True
        g = R.extent_inv_extent(W, H)
        sky, wrc = R.Tex(op.sky16, 16, 16 * 6, "rgba16f"), R.Tex.zeros(1, 1, "rgba16f")
        slots_off = 0
        for fi in range(6):
            fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.03))
            op.render_inputs(fc); op.reprojection(fc)
            op.L.okj_ircache_prepare(op.ircache, C.byref(fc))
            op.L.okj_ircache_prepare_and_reset(op.ircache)
            cur = _irc_host_state(op)["cur"]
            alloc = int(op.ircache_buffer("meta", np.uint32)[3])
            for which, pass_name in enumerate(("ircache/trace_accessibility.rgen", "ircache/ircache_validate.rgen", "ircache/trace_irradiance.rgen")):
                s = _irc_snapshot(op)
                B = {n: R.Buf(s[n]) for n in s}
                gm = B["grid_meta%d" % cur]
                if which == 0:       # ircache.rs:396-414, trace_rays_indirect(args + 16): alloc_count * max(16, 4, 4) rays
                    R.run_pass(pass_name, [B["spatial"], B["life"], B["reposition_proposal"], B["meta"], B["aux"], B["entry_indirection"]], None, fc, (alloc * 16, 1, 1))
                else:                # ircache.rs:416-481: trace_rays([MAX_ENTRIES * 4, 1, 1])
                    R.run_pass(pass_name, [B["spatial"], sky, gm, B["life"], B["reposition_proposal"], B["reposition_proposal_count"], wrc, B["meta"], B["aux"], B["pool"],
                                           B["entry_indirection"], B["entry_cell"]], None, fc, (MAX_ENTRIES * 4, 1, 1))
                op.L.okj_ircache_ray_pass(op.ircache, C.byref(fc), osc.h, op.sky16.ctypes.data, 16, which)
                ref = _irc_snapshot(op)
                for n in ref:
                    if n == "grid_meta%d" % (1 - cur):
                        continue
                    if not _numbers_equal(s[n], ref[n]):
                        a, b = s[n].view(np.uint32), ref[n].view(np.uint32)
                        bad = np.nonzero(a != b)[0]
                        frac = bad.size / max(1, a.size)
                        slots_off += bad.size
                        print(f"frame {fi} {pass_name} {n}: dwords {bad} : {a[bad]} vs {b[bad]}  (as float {a[bad].view(np.float32)} vs {b[bad].view(np.float32)})")
                        assert n in ("aux", "reposition_proposal") and frac <= 2e-3, f"frame {fi} {pass_name}: buffer {n}: {bad.size} of {a.size} dwords differ, first at {bad[:6]}: {a[bad[:6]]} vs {b[bad[:6]]}"
            op.L.okj_rtdgi_reproject(op.rtdgi, C.byref(fc), op.reprojection_map.ctypes.data, W, H)
            op.ircache_sum_up(fc)
            # ---- rtdgi with the live cache: the three passes that touch it, pass by pass; then the rest of the frame on the oracle
            first = True
            for pname in ("EXTRACT_HALF", "VALIDATE", "TRACE"):
                before, s = _surfaces(op), _irc_snapshot(op)
                mask = KJ_RTDGI_PASS[pname] | (0 if first else KEEP)
                first = False
                p = op.params(mask)
                op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
                if pname == "EXTRACT_HALF" or fi < 2:
                    continue
                after, ref = _surfaces(op), _irc_snapshot(op)
                f = _Frame(op, before, fi, W, H)
                irc = _ircache_bind_set(s, cur)
                if pname == "VALIDATE":
                    R.run_pass("rtdgi/diffuse_validate.rgen",
                               [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.wr("rtdgi.reservoir" + f.hist_sfx), f.hist("rtdgi.ray"), f.reprojection_map()] + irc +
                               [wrc, sky, f.wr("rtdgi.radiance" + f.hist_sfx), f.hist("rtdgi.ray_orig"), f.wr("rt_history_validity_pre_input_tex")], [g], fc, (hw, hh, 1))
                else:
                    R.run_pass("rtdgi/trace_diffuse.rgen",
                               [f.rd("half_view_normal_tex"), f.depth(), f.rd("reprojected_history_tex"), f.reprojection_map()] + irc +
                               [wrc, sky, f.hist("rtdgi.ray_orig"), f.wr("candidate_radiance_tex"), f.wr("candidate_normal_tex"), f.wr("candidate_hit_tex"),
                                f.rd("rt_history_validity_pre_input_tex"), f.wr("rt_history_validity_input_tex")], [g], fc, (hw, hh, 1))
                for n, t in f.written.items():
                    r = P.compare(t.raw, after[n], P.fmt_of(n), vector=P.is_vector(n))
                    assert P.pass_within_bars(pname, r), f"frame {fi} pass {pname} (live cache) surface {n}: reference HLSL vs oracle {r}"
                for n in ("meta", "pool", "reposition_proposal", "reposition_proposal_count", "grid_meta%d" % cur, "entry_cell", "life"):
                    if not _numbers_equal(s[n], ref[n]):
                        a, b = s[n].view(np.uint32), ref[n].view(np.uint32)
                        bad = np.nonzero(a != b)[0]
                        assert n == "reposition_proposal" and bad.size <= 2e-3 * a.size, f"frame {fi} pass {pname} (live cache): buffer {n}: {bad.size} dwords differ, first at {bad[:6]}: {a[bad[:6]]} vs {b[bad[:6]]}"
            rest = KJ_RTDGI_PASS["ALL"] & ~(KJ_RTDGI_PASS["EXTRACT_HALF"] | KJ_RTDGI_PASS["VALIDATE"] | KJ_RTDGI_PASS["TRACE"])
            p = op.params(rest | KEEP)
            op.L.okj_rtdgi_render(op.rtdgi, C.byref(fc), C.byref(p), C.byref(op.out))
            fs.retire_frame()
        assert int(op.ircache_buffer("meta", np.uint32)[3]) > 50, "the cache never filled: the test would be vacuous"
        print(f"cache slots differing over the run: {slots_off}")
    finally:
        L.ref_set_linear_order(0)
        oracle.lib().okj_set_threads(n_threads)


# ---------------------------------------------------------------------------------------------------------------- SURVEY 8f rows
@recorded_case(lambda k: "ssao_guide")
def test_ssao_guide_passes_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """SsgiRenderer::render + filter_ssgi (renderers/ssgi.rs:25-181): ssgi/{ssgi,spatial_filter,upsample,temporal_filter}.hlsl from the
    reference's text against the oracle's four stages, six frames of a moving camera over the 20 k-triangle city (USE_AO_ONLY: the
    previous-radiance input never reaches the output and is bound black)."""
    from kajiya_amd import scenes
    _bind_luts(oracle)
    W, H = _scaled(104, 60)
    hw, hh = W // 2, H // 2
    op = oracle.OraclePipeline(oracle.OracleScene(scenes.procedural_city(seed=1234, target_tris=20000)), W, H)
    g, ho = R.extent_inv_extent(W, H), R.extent_inv_extent(hw, hh)
    FM = {"ssgi_tex": "r16f", "spatially_filtered_tex": "r16f", "upsampled_tex": "r16f", "ssgi": "r16f", "filtered_output_tex": "r8", "half_view_normal_tex": "rgba8s", "half_depth_tex": "r32f"}
    FULL = {"upsampled_tex", "ssgi", "filtered_output_tex"}

    def snap():
        out = {}
        for n in FM:
            for sfx in ("", ":0", ":1"):
                try:
                    out[n + sfx] = op.ssgi_surface(n + sfx, np.uint8, (-1,)).copy()
                except (KeyError, AttributeError):
                    pass
        return out
    for fi, fc in enumerate(_frame_constants(W, H, 6, "city")):
        op.render_inputs(fc); op.reprojection(fc)
        before = snap() if fi else {}
        op.ssgi_frame(fc)
        after = snap()
        out_sfx, hist_sfx = (":0", ":1") if fi % 2 == 0 else (":1", ":0")

        def tex(d, n):
            raw = d.get(n)
            if raw is None:
                raw = np.zeros_like(after[n])
            w, h = (W, H) if n.split(":")[0] in FULL else (hw, hh)
            return R.Tex(raw.copy(), w, h, FM[n.split(":")[0]])
        written = {}

        def wr(n):
            written[n] = tex(after, n)
            written[n].raw[:] = 0xcd
            return written[n]
        gbuffer, depth, reproj = R.Tex(op.gbuffer, W, H, "rgba32f"), R.Tex(op.depth, W, H, "r32f"), R.Tex(op.reprojection_map, W, H, "rgba16s")
        R.run_pass("ssgi/ssgi", [gbuffer, tex(after, "half_depth_tex"), tex(after, "half_view_normal_tex"), R.Tex.zeros(W, H, "rgba16f"), reproj, wr("ssgi_tex")], [g, ho], fc, (hw, hh, 1))
        R.run_pass("ssgi/spatial_filter", [tex(after, "ssgi_tex"), tex(after, "half_depth_tex"), tex(after, "half_view_normal_tex"), wr("spatially_filtered_tex")], None, fc, (hw, hh, 1))
        R.run_pass("ssgi/upsample", [tex(after, "spatially_filtered_tex"), depth, gbuffer, wr("upsampled_tex")], None, fc, (W, H, 1))
        # the oracle double-buffers the R8 output like the product does (the reference's is a transient): the one written this frame
        fo = "filtered_output_tex" + (":0" if not np.array_equal(after["filtered_output_tex:0"], before.get("filtered_output_tex:0", None)) else ":1") if fi else \
            ("filtered_output_tex:0" if after["filtered_output_tex:0"].any() else "filtered_output_tex:1")
        R.run_pass("ssgi/temporal_filter", [tex(after, "upsampled_tex"), tex(before, "ssgi" + hist_sfx), reproj, wr(fo), wr("ssgi" + out_sfx)], [g], fc, (W, H, 1))
        for n, t in written.items():
            r = P.compare(t.raw, after[n], FM[n.split(":")[0]])
            _check(r, f"frame {fi} SSAO guide surface {n}")


@recorded_case(lambda k: "shadow_denoise")
def test_shadow_denoise_passes_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """ShadowDenoiseRenderer::render (renderers/shadow_denoise.rs:19-149): shadow_denoise/{bitpack_shadow_mask,megakernel,spatial_filter}.hlsl
    with AMD's ffx_denoiser_shadows_* code they include (wave ballots, quad reads, group-shared tiles: the lock-step scheduler), chained
    like the host records them -- bit-pack, temporal megakernel, three a-trous passes (step 1, 2, 4) -- against the oracle's frame."""
    from kajiya_amd import scenes
    _bind_luts(oracle)
    W, H = _scaled(104, 60)
    TW, TH = (W + 7) // 8, (H + 3) // 4
    op = oracle.OraclePipeline(oracle.OracleScene(scenes.procedural_city(seed=1234, target_tris=20000)), W, H)
    g = R.extent_inv_extent(W, H)
    ext = np.array([TW, TH], np.uint32)
    FM = {"bitpacked_shadows_image": ("r32ui", TW, TH), "metadata_image": ("r32ui", TW, TH), "spatial_input_image": ("rg16f", W, H), "temp": ("rg16f", W, H),
          "shadow_denoise_accum": ("rg16f", W, H), "shadow_denoise_moments": ("rgba16f", W, H)}

    def snap():
        out = {}
        for n in FM:
            for sfx in ("", ":0", ":1"):
                try:
                    out[n + sfx] = op.shadow_denoise_surface(n + sfx, np.uint8, (-1,)).copy()
                except (KeyError, AttributeError):
                    pass
        return out
    for fi, fc in enumerate(_frame_constants(W, H, 5, "city")):
        op.render_inputs(fc); op.reprojection(fc)
        mask = op.sun_shadow_mask(fc)
        before = snap() if fi else {}
        op.shadow_denoise(fc, mask)
        after = snap()
        out_sfx, hist_sfx = (":0", ":1") if fi % 2 == 0 else (":1", ":0")

        def tex(d, n):
            fmt, w, h = FM[n.split(":")[0]]
            raw = d.get(n)
            return R.Tex(raw.copy() if raw is not None else np.zeros_like(after[n]), w, h, fmt)
        mask_t, reproj = R.Tex(mask, W, H, "r8"), R.Tex(op.reprojection_map, W, H, "rgba16s")
        gn, depth = R.Tex(op.geometric_normal, W, H, "a2r10g10b10"), R.Tex(op.depth, W, H, "r32f")
        bitpacked, moments, spatial_input, metadata = tex({}, "bitpacked_shadows_image"), tex({}, "shadow_denoise_moments" + out_sfx), tex({}, "spatial_input_image"), tex({}, "metadata_image")
        accum, temp = tex({}, "shadow_denoise_accum" + out_sfx), tex({}, "temp")
        R.run_pass("shadow_denoise/bitpack_shadow_mask", [mask_t, bitpacked], [g, ext], fc, (TW * 2, TH, 1))
        R.run_pass("shadow_denoise/megakernel", [mask_t, bitpacked, tex(before, "shadow_denoise_moments" + hist_sfx), tex(before, "shadow_denoise_accum" + hist_sfx), reproj,
                                                 moments, spatial_input, metadata], [g, ext], fc, (W, H, 1))
        for step, (src, dst) in ((1, (spatial_input, accum)), (2, (accum, temp)), (4, (temp, spatial_input))):
            src_copy = R.Tex(src.raw.copy(), W, H, "rg16f")
            R.run_pass("shadow_denoise/spatial_filter", [src_copy, metadata, gn, depth, dst], [g, ext, np.uint32(step)], fc, (W, H, 1))
        for n, t in (("bitpacked_shadows_image", bitpacked), ("metadata_image", metadata), ("shadow_denoise_moments" + out_sfx, moments),
                     ("shadow_denoise_accum" + out_sfx, accum), ("temp", temp), ("spatial_input_image", spatial_input)):
            fmt = FM[n.split(":")[0]][0]
            if fmt == "r32ui":
                assert np.array_equal(t.raw, after[n]), (fi, n, int((t.raw.view(np.uint32) != after[n].view(np.uint32)).sum()))
            else:
                _check(P.compare(t.raw, after[n], fmt), f"frame {fi} shadow denoiser surface {n}")
        assert 0 < (mask == 0).mean() < 1


@recorded_case(lambda k: "light_gbuffer")
def test_light_gbuffer_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """light_gbuffer.hlsl as renderers/deferred.rs:8-46 records it (G-buffer, depth, shadow mask, reflections, GI, the nine cache buffers,
    the two sky cubes) against the oracle's combine: both outputs, sky pixels with the sun disc included."""
    from kajiya_amd import scenes
    _bind_luts(oracle)
    W, H = _scaled(96, 64)
    op = oracle.OraclePipeline(oracle.OracleScene(scenes.cornell_box()), W, H)
    g = R.extent_inv_extent(W, H)
    for fi, fc in enumerate(_frame_constants(W, H, 4)):
        op.render_inputs(fc); op.reprojection(fc); op.rtdgi_frame(fc)
        mask = op.sun_shadow_mask(fc)
        gi = op.surface("spatial_filtered_tex", np.float16, (H, W, 4)).copy()
        rtr = np.zeros((H, W), np.uint32)
        ref_t, ref_o = op.light_gbuffer(fc, mask, gi, rtr, 0)
        t_out, o_out = R.Tex.zeros(W, H, "rgba16f"), R.Tex.zeros(W, H, "rgba16f")
        R.run_pass("light_gbuffer", [R.Tex(op.gbuffer, W, H, "rgba32f"), R.Tex(op.depth, W, H, "r32f"), R.Tex(mask, W, H, "r8"), R.Tex(rtr, W, H, "r11g11b10f"), R.Tex(gi, W, H, "rgba16f")] +
                   _ircache_bind_set(_empty_ircache(), 0) + [R.Tex.zeros(1, 1, "rgba16f"), t_out, o_out, R.Tex(op.sky64, 64, 64 * 6, "rgba16f"), R.Tex(op.sky16, 16, 16 * 6, "rgba16f")],
                   [g, np.uint32(0), np.uint32(0)], fc, (W, H, 1))
        _check(P.compare(t_out.raw, ref_t.view(np.uint8).reshape(-1), "rgba16f"), f"frame {fi} light_gbuffer temporal_output")
        _check(P.compare(o_out.raw, ref_o.view(np.uint8).reshape(-1), "rgba16f"), f"frame {fi} light_gbuffer output")


# ---------------------------------------------------------------------------------------------------------------- reflections
RTR_PINGPONG = ["rtr.temporal", "rtr.ray_len", "rtr.irradiance", "rtr.ray_orig", "rtr.ray", "rtr.reservoir", "rtr.rng", "rtr.hit_normal"]
RTR_NAMES = [n + s for n in RTR_PINGPONG for s in (":0", ":1")] + ["refl_restir_invalidity_tex", "resolved_tex"]
RTR_CANDIDATES = ["candidate_radiance_tex", "candidate_hit_tex", "candidate_normal_tex"]
RTR_PASS_ORDER = ["TRACE", "VALIDATE", "RESTIR_TEMPORAL", "RESOLVE", "TEMPORAL_FILTER", "CLEANUP"]
_RTR_TEX_FMT = {"u32": "r32ui", "rtr_ray_orig": "rgba32f"}       # parity.py's names for the rng image and RtrRestirRayOrigin (RGBA32F, rtr.rs:160-169)


def _rtr_state(op):
    st = {n: op.rtr_surface(n, np.uint8, (-1,)).copy() for n in RTR_NAMES}
    st.update({n: op.surface(n, np.uint8, (-1,)).copy() for n in RTR_CANDIDATES})
    return st


class _RtrFrame(_Frame):
    """_Frame over the oracle's reflection state. All eight PingPongTemporalResources of RtrRenderer (rtr.rs:19-27) turn once per frame."""

    def _tex(self, name):
        w, h = (self.W, self.H) if P.base_name(name) in ("rtr.temporal", "rtr.ray_len", "resolved_tex") else (self.hw, self.hh)
        fmt = P.fmt_of(name)
        return R.Tex(self.before[name].copy(), w, h, _RTR_TEX_FMT.get(fmt, fmt))

    def rd(self, name):
        return self._tex(name)

    def wr(self, name):
        t = self._tex(name)
        self.written[name] = t
        return t


@pytest.mark.parametrize("reuse", [1, 0])
@recorded_case(lambda k: "rtr_reuse_rtdgi_rays" if k["reuse"] == 1 else None)
def test_rtr_passes_reference_hlsl_vs_oracle(oracle, libm_sincos, reuse):
    """RtrRenderer::trace + TracedRtr::filter_temporal (renderers/rtr.rs:97-400,440-480) from the reference's own text -- rtr/reflection.rgen.hlsl
    and reflection_validate.rgen.hlsl (with reflection_trace_common.inc.hlsl, inc/blue_noise.hlsl's sampler, rt/gbuffer.rchit.hlsl on every hit),
    rtr_restir_temporal.hlsl, resolve.hlsl, temporal_filter.hlsl, spatial_cleanup.hlsl -- pass by pass against oracle/okj_rtr.hpp on the
    oracle's frame state, recorded the way rtr.rs records them (binding order, constants, dispatch extents; `reuse_rtdgi_rays` on -- the
    rough pixels keep rtdgi's candidates, rtr.rs:32 -- and off). The irradiance cache is bound empty (the oracle's null lookup hook). Every surface every pass
    writes must come out number for number, with the two places where the oracle DEFINES what the text leaves to chance set
    aside and counted:
      (a) reflection_validate.rgen.hlsl:88 normalises `ray_hit_ws - ray_orig_ws`; where the history holds no ray yet that is normalize(0) = NaN
          handed to TraceRay (undefined in the API). The oracle and the kernels trace +Z. Quads whose validation pixel has a zero history ray
          are left out of the comparison;
      (b) resolve.hlsl:535-540 divides by |sample origin - pixel origin|, which for the pixel's own half-res sample is the rounding residue of
          `(origin - eye) + eye`: a direction made of rounding noise, different under any other contraction / association of the same
          arithmetic. The oracle and the kernels count a residue as zero. Here the oracle runs with that rule switched off (a test knob) --
          the text as written -- and the texels the rule changes are counted on one frame."""
    from kajiya_amd import scenes, rtr_tables
    from kajiya_amd.abi import KJ_RTR_PASS
    import test_gpu_parity as T
    _bind_luts(oracle)
    desc = scenes.glossy_test_scene()
    osc = oracle.OracleScene(desc)
    keep = _bind_scene(oracle, osc, desc)
    W, H = _scaled(72, 44)
    hw, hh = (W + 1) // 2, (H + 1) // 2
    qw, qh = (hw + 1) // 2, (hh + 1) // 2
    op = oracle.OraclePipeline(osc, W, H)
    g, ho = R.extent_inv_extent(W, H), R.extent_inv_extent(hw, hh)
    ranking, scrambling = rtr_tables.ranking_and_scrambling()
    sampler = [R.Buf(ranking), R.Buf(scrambling), R.Buf(rtr_tables.sobol_256x256())]
    offsets = rtr_tables.spatial_resolve_offsets()
    sky = R.Tex(op.sky64, 64, 64 * 6, "rgba16f")
    wrc = R.Tex.zeros(1, 1, "rgba16f")
    compared, undefined_quads, own_rule = {}, 0, {}
    for fi, fc in enumerate(T._frame_constants(W, H, 7, "textured")):
        op.render_inputs(fc); op.reprojection(fc)
        op.rtdgi_frame(fc)
        if fi < 4:
            op.rtr_frame(fc)
            op.L.okj_rtr_set_literal_own_sample_shadowing(op.rtr, 1)       # (b) of the docstring
            op.L.okj_rtr_set_options(op.rtr, reuse)
            continue
        rtdgi = R.Tex(np.frombuffer((C.c_uint8 * (W * H * 8)).from_address(op.out.screen_irradiance_tex), np.uint8).copy(), W, H, "rgba16f")
        for k, pname in enumerate(RTR_PASS_ORDER):
            before = _rtr_state(op)
            op.rtr_frame(fc, KJ_RTR_PASS[pname] | (0 if k == 0 else KJ_RTR_PASS["KEEP"]))
            after = _rtr_state(op)
            f = _RtrFrame(op, before, fi, W, H)
            irc = _ircache_bind_set(_empty_ircache(), 0)
            half_view_normal = R.Tex(op.rtr_surface("half_view_normal_tex", np.uint8, (-1,)).copy(), hw, hh, "rgba8s")
            half_depth = R.Tex(op.rtr_surface("half_depth_tex", np.uint8, (-1,)).copy(), hw, hh, "r32f")
            if pname == "TRACE":               # rtr.rs:118-151
                R.run_pass("rtr/reflection.rgen",
                           [f.gbuffer(), f.depth()] + sampler + [rtdgi, sky] + irc +
                           [wrc, f.wr("candidate_radiance_tex"), f.wr("candidate_hit_tex"), f.wr("candidate_normal_tex"), f.out("rtr.rng")],
                           [g, np.uint32(reuse)], fc, (hw, hh, 1))
            elif pname == "VALIDATE":          # rtr.rs:207-232: the invalidity image is a fresh transient; half of the half-res extent
                inval = f.wr("refl_restir_invalidity_tex")
                inval.raw[:] = 0
                R.run_pass("rtr/reflection_validate.rgen",
                           [f.gbuffer(), f.depth(), rtdgi, sky, inval] + irc +
                           [wrc, f.hist("rtr.ray_orig"), f.hist("rtr.ray"), f.hist("rtr.rng"), f.wr("rtr.irradiance" + f.hist_sfx), f.wr("rtr.reservoir" + f.hist_sfx)],
                           [g], fc, (qw, qh, 1))
            elif pname == "RESTIR_TEMPORAL":   # rtr.rs:234-262
                R.run_pass("rtr/rtr_restir_temporal",
                           [f.gbuffer(), half_view_normal, f.depth(), f.rd("candidate_radiance_tex"), f.rd("candidate_hit_tex"), f.rd("candidate_normal_tex"),
                            f.hist("rtr.irradiance"), f.hist("rtr.ray_orig"), f.hist("rtr.ray"), f.hist("rtr.rng"), f.hist("rtr.reservoir"), f.reprojection_map(),
                            f.hist("rtr.hit_normal"), f.out("rtr.irradiance"), f.out("rtr.ray_orig"), f.out("rtr.ray"), f.out("rtr.rng"), f.out("rtr.hit_normal"),
                            f.out("rtr.reservoir")], [g], fc, (hw, hh, 1))
            elif pname == "RESOLVE":           # rtr.rs:290-318
                R.run_pass("rtr/resolve",
                           [f.gbuffer(), f.depth(), f.rd("candidate_radiance_tex"), f.rd("candidate_hit_tex"), f.rd("candidate_normal_tex"), f.hist("rtr.temporal"),
                            f.reprojection_map(), half_view_normal, half_depth, f.hist("rtr.ray_len"), f.out_as_input("rtr.irradiance"), f.out_as_input("rtr.ray"),
                            f.out_as_input("rtr.reservoir"), f.out_as_input("rtr.ray_orig"), f.out_as_input("rtr.hit_normal"), f.wr("resolved_tex"), f.out("rtr.ray_len")],
                           [g, offsets], fc, (W, H, 1))
            elif pname == "TEMPORAL_FILTER":   # rtr.rs:449-465
                R.run_pass("rtr/temporal_filter",
                           [f.rd("resolved_tex"), f.hist("rtr.temporal"), f.depth(), f.out_as_input("rtr.ray_len"), f.reprojection_map(), f.rd("refl_restir_invalidity_tex"),
                            f.gbuffer(), f.out("rtr.temporal")], [g], fc, (W, H, 1))
            else:                              # rtr.rs:467-477
                R.run_pass("rtr/spatial_cleanup", [f.out_as_input("rtr.temporal"), f.depth(), f.geometric_normal(), f.wr("resolved_tex")], [offsets], fc, (W, H, 1))
            if pname == "VALIDATE":            # (a) of the docstring: quads whose validation ray has no direction
                off = np.array([(1, 1), (1, 0), (0, 0), (0, 1)])[fc.frame_index & 3]           # hi_px_subpixels (inc/frame_constants.hlsl:235-240)
                ray = P.decode(before["rtr.ray" + f.hist_sfx], "rgba16f").reshape(hh, hw, 4)[off[1]::2, off[0]::2, :3]
                # a zero-length ray where the quad's own pixel (hi_px = (2 q + off) * 2 + off, reflection_validate.rgen.hlsl:49-54) sees geometry
                dead = np.argwhere((ray == 0).all(axis=-1) & (op.depth.reshape(H, W)[3 * off[1]::4, 3 * off[0]::4][:ray.shape[0], :ray.shape[1]] != 0))
                undefined_quads += len(dead)
                for n, t in f.written.items():
                    bpt = t.raw.size // (hw * hh)
                    a, b = t.raw.reshape(hh, hw, bpt), after[n].reshape(hh, hw, bpt)
                    for qy, qx in dead:
                        a[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2] = b[2 * qy:2 * qy + 2, 2 * qx:2 * qx + 2]
            if pname == "RESOLVE" and fi == 6:     # (b): what the oracle's own rule changes, on the same inputs
                op.L.okj_rtr_set_literal_own_sample_shadowing(op.rtr, 0)
                op.rtr_frame(fc, KJ_RTR_PASS[pname] | KJ_RTR_PASS["KEEP"])
                ruled = _rtr_state(op)
                own_rule = {n: int((P.decode(ruled[n], P.fmt_of(n)) != P.decode(after[n], P.fmt_of(n))).any(axis=-1).sum()) for n in f.written}
                op.L.okj_rtr_set_literal_own_sample_shadowing(op.rtr, 1)
                op.rtr_frame(fc, KJ_RTR_PASS[pname] | KJ_RTR_PASS["KEEP"])
                assert all(np.array_equal(v, after[n]) for n, v in _rtr_state(op).items())
            for n, t in f.written.items():
                r = P.compare(t.raw, after[n], P.fmt_of(n), vector=P.is_vector(n))
                ident = _numbers_equal(t.raw, after[n]) if P.fmt_of(n) in ("u32", "reservoir") else r["differ_frac"] == 0.0
                compared[(pname, P.base_name(n))] = compared.get((pname, P.base_name(n)), True) and bool(ident)
                if not ident:
                    a, b = P.decode(t.raw, P.fmt_of(n)), P.decode(after[n], P.fmt_of(n))
                    bad = np.nonzero((a != b).any(axis=-1))[0]
                    print(f"frame {fi} {pname} {n}: {bad.size} of {a.shape[0]} texels differ, e.g. texel {bad[:3]}: ref {a[bad[:3]]} oracle {b[bad[:3]]}  {r}")
            for n in after:
                if n not in f.written and not np.array_equal(after[n], before[n]):
                    raise AssertionError(f"frame {fi} pass {pname}: the oracle wrote {n}, the reference pass does not")
    print(sorted(compared.items()))
    print(f"validate quads without a ray direction (left out): {undefined_quads} of {3 * qw * qh}; texels the oracle's own-sample rule changes at {W}x{H}: {own_rule}")
    assert len(compared) >= 17, sorted(compared)
    assert all(compared.values()), sorted(k for k, v in compared.items() if not v)
    assert undefined_quads <= 0.02 * 3 * qw * qh
    assert 0 < own_rule["resolved_tex"] <= 0.05 * W * H and own_rule["rtr.ray_len" + (":0" if 6 % 2 == 0 else ":1")] <= 0.08 * W * H, own_rule


@recorded_case(lambda k: "sky_cubes")
def test_sky_cubes_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """The two sky cubes every GI pass reads (renderers/sky.rs:4-35): sky/comp_cube.hlsl (64^2 x 6, the atmosphere integrated per texel) and
    convolve_cube.hlsl (16^2 x 6, 512 cone samples of the first through the cube sampler) from the reference's text against the oracle's
    images -- the ones all the other tests of this file bind."""
    from kajiya_amd import scenes
    osc = oracle.OracleScene(scenes.cornell_box())
    op = oracle.OraclePipeline(osc, 32, 32)
    for fi, fc in enumerate(_frame_constants(32, 32, 2)):
        op.render_inputs(fc)
        sky = R.Tex.zeros(64, 64 * 6, "rgba16f")
        sky.h = 64                                   # six slices of 64 x 64: the slice count follows from the bound size
        R.run_pass("sky/comp_cube", [sky], None, fc, (64, 64, 6))
        r = P.compare(sky.raw, op.sky64.reshape(-1).view(np.uint8), "rgba16f")
        _check(r, f"frame {fi} sky cube")
        conv = R.Tex.zeros(16, 16 * 6, "rgba16f")
        conv.h = 16
        R.run_pass("convolve_cube", [R.Tex(op.sky64, 64, 64 * 6, "rgba16f"), conv], [np.uint32(16)], fc, (16, 16, 6))
        r = P.compare(conv.raw, op.sky16.reshape(-1).view(np.uint8), "rgba16f")
        _check(r, f"frame {fi} convolved sky cube")


@recorded_case(lambda k: "light_specular")
def test_light_specular_reference_hlsl_vs_oracle(oracle, libm_sincos):
    """LightingRenderer::render_specular (renderers/lighting.rs:23-88): lighting/sample_lights.rgen.hlsl (one shadow ray per half-res pixel to a
    triangle light picked by the blue-noise image) and lighting/spatial_reuse_lights.hlsl (the rtr resolve kernel's footprint, added INTO
    rtr's resolved B10G11R11 image) from the reference's text, on the reference's half-res normal / depth extractions, against the oracle's
    restatement of the pair; the emissive box of the scene registered as 12 triangle lights."""
    from kajiya_amd import scenes, rtr_tables
    import test_gpu_parity as T
    _bind_luts(oracle)
    desc = scenes.glossy_test_scene()
    osc = oracle.OracleScene(desc, use_lights=True)
    keep = _bind_scene(oracle, osc, desc)
    n_lights = osc.triangle_light_count
    lights = np.zeros(n_lights * 12, np.float32)
    oracle.lib().okj_scene_triangle_lights(C.c_void_p(osc.h), C.c_void_p(lights.ctypes.data))
    R.set_named("triangle_lights_dyn", R.Buf(lights))
    W, H = _scaled(72, 44)
    hw, hh = (W + 1) // 2, (H + 1) // 2
    op = oracle.OraclePipeline(osc, W, H)
    g = R.extent_inv_extent(W, H)
    offsets = rtr_tables.spatial_resolve_offsets()
    rng = np.random.RandomState(9)
    for fi, fc in enumerate(T._frame_constants(W, H, 3, "textured")):
        fc.triangle_light_count = n_lights
        op.render_inputs(fc)
        base = (rng.randint(8 << 6, 15 << 6, size=(H, W)) | (rng.randint(8 << 6, 15 << 6, size=(H, W)) << 11) | (rng.randint(8 << 5, 15 << 5, size=(H, W)) << 22)).astype(np.uint32)
        ref = base.copy()
        rays = op.lighting_render_specular(fc, ref)
        gb, depth = R.Tex(op.gbuffer, W, H, "rgba32f"), R.Tex(op.depth, W, H, "r32f")
        refl0, refl1, refl2 = R.Tex.zeros(hw, hh, "rgba16f"), R.Tex.zeros(hw, hh, "rgba32f"), R.Tex.zeros(hw, hh, "rgba8s")
        R.run_pass("lighting/sample_lights.rgen", [depth, refl0, refl1, refl2], [g], fc, (hw, hh, 1))
        hvn, hd = R.Tex.zeros(hw, hh, "rgba8s"), R.Tex.zeros(hw, hh, "r32f")
        R.run_pass("extract_half_res_gbuffer_view_normal_rgba8", [gb, hvn], None, fc, (hw, hh, 1))
        R.run_pass("extract_half_res_depth", [depth, hd], None, fc, (hw, hh, 1))
        out = R.Tex(base.copy(), W, H, "r11g11b10f")
        R.run_pass("lighting/spatial_reuse_lights", [gb, depth, refl0, refl1, refl2, hvn, hd, out], [g, offsets], fc, (W, H, 1))
        r = P.compare(out.raw, ref.view(np.uint8).reshape(-1), "r11g11b10f")
        added = P.decode(ref.view(np.uint8).reshape(-1), "r11g11b10f") - P.decode(base.view(np.uint8).reshape(-1), "r11g11b10f")
        assert rays > 0.3 * hw * hh and (added.max(-1) > 0).mean() > 0.02, (rays, (added.max(-1) > 0).mean())
        _check(r, f"frame {fi} resolved image with the lights' specular")


@pytest.mark.parametrize("W,H,frame_index,mult,contrast,lut_seed", [(160, 90, 3, 1.3, 1.1, 0), (333, 187, 0, 1.0, 1.0, None)])
@recorded_case(lambda k: "post_160x90" if k["W"] == 160 else None)
def test_post_passes_reference_hlsl_vs_oracle(oracle, libm_sincos, W, H, frame_index, mult, contrast, lut_seed):
    """PostProcessRenderer::render (renderers/post.rs:10-272), the passes that are HLSL in the reference: blur.hlsl (mips 1.. of the blur pyramid),
    post/luminance_histogram_{clear,calculate,copy}.hlsl and post_combine.hlsl (glare, vignette, the display transform with its Bezold-Brucke
    LUT, contrast, dither), each on the oracle's own inputs to that pass, against the oracle. Mip 0 of the blur pyramid and the reverse pyramid
    are Rust kernels in the reference (rust-shaders/src/{blur,rev_blur}.rs; tests/test_post_oracle.py holds the oracle to their text)."""
    W, H = _scaled(W, H)
    from kajiya_amd import frame, post_tables
    _bind_luts(oracle)
    lut = post_tables.zero_bezold_brucke_lut() if lut_seed is None else post_tables.synthetic_bezold_brucke_lut(lut_seed)
    R.set_bindless(2, R.Tex(np.ascontiguousarray(lut, np.float16).reshape(64, 2), 64, 1, "rg16f"))
    rng = np.random.RandomState(W + H)
    ys, xs = np.mgrid[0:H, 0:W]
    img = np.stack([0.5 + 0.5 * np.sin(xs * 0.05), 0.5 + 0.5 * np.cos(ys * 0.07), 0.5 + 0.5 * np.sin((xs + ys) * 0.03)], -1) * rng.uniform(0.2, 3.0, (H, W, 1))
    for _ in range(W * H // 2000):
        img[rng.randint(H), rng.randint(W)] = rng.uniform(20, 900, 3)
    img[: H // 6, : W // 5] = 0.0                                    # black: the NaN-chromaticity path
    inp = np.concatenate([img, np.ones((H, W, 1))], -1).astype(np.float16)
    fs = frame.FrameState((W, H))
    fs.frame_idx, fs.pre_exposure = frame_index, 0.7
    fc = fs.prepare_frame_constants(frame.orbit_camera(0, (W, H)))
    op = oracle.OraclePost(lut)
    ref_out = op.render(fc, inp, mult, contrast).copy()
    levels = op.mip_levels()
    mip = lambda pyr, l: R.Tex(op.mip(pyr, l).copy(), *op.mip_extent(l), "r11g11b10f")
    for l in range(1, levels):                                     # post.rs:36-58
        w, h = op.mip_extent(l)
        out = R.Tex.zeros(w, h, "r11g11b10f")
        R.run_pass("blur", [mip("blur_pyramid", l - 1), out], None, fc, (w, h, 1))
        _check(P.compare(out.raw, op.mip("blur_pyramid", l).reshape(-1).view(np.uint8), "r11g11b10f"), f"blur pyramid mip {l}")
    hl = max(0, levels - 7)                                        # post.rs:144-183
    pw, ph = (W + 1) // 2, (H + 1) // 2
    ext = np.array([-(-pw // (1 << hl)), -(-ph // (1 << hl))], np.uint32)
    tmp, dst = R.Buf(np.full(256, 0xdeadbeef, np.uint32)), R.Buf(np.zeros(256, np.uint32))
    R.run_pass("post/luminance_histogram_clear", [tmp], None, fc, (256, 1, 1))
    R.run_pass("post/luminance_histogram_calculate", [mip("blur_pyramid", hl), tmp], [ext], fc, (int(ext[0]), int(ext[1]), 1))
    R.run_pass("post/luminance_histogram_copy", [tmp, dst], None, fc, (256, 1, 1))
    assert np.array_equal(dst.raw.view(np.uint32), op.histogram()), np.nonzero(dst.raw.view(np.uint32) != op.histogram())
    assert op.histogram().sum() > 0
    out = R.Tex.zeros(W, H, "r11g11b10f")                          # post.rs:252-269
    R.run_pass("post_combine", [R.Tex(inp, W, H, "rgba16f"), mip("blur_pyramid", 0), mip("rev_blur_pyramid", 0), tmp, out],
               [R.extent_inv_extent(W, H), np.float32(mult), np.float32(contrast)], fc, (W, H, 1))
    _check(P.compare(out.raw, ref_out.reshape(-1).view(np.uint8), "r11g11b10f"), "post combine")


# ---------------------------------------------------------------------------------------------------------------------------------------
# phase A: the leaf functions of the reference's headers, one by one

_PROBE_ROWS = [   # (what the row of oracle/ref_hlsl/probes/inc_functions.hlsl holds, which of its four words are floats)
    ("hash1, hash_combine2, hash2, hash3", ""),
    ("uint_to_u01_float, interleaved_gradient_noise", "xy"),
    ("unpack_unorm(8), pack_unorm(11), unpack_unorm(11), pack_unorm(10)", "xz"),
    ("pack_normal_11_10_11, unpack_normal_11_10_11", "yzw"),
    ("unpack_normal_11_10_11_no_normalize", "xyz"),
    ("unpack_normal_11_10_11_uint_no_normalize", "xyz"),
    ("pack_color_888, unpack_color_888", "yzw"),
    ("pack_2x16f_uint, unpack_2x16f_uint", "yz"),
    ("float3_to_rgb9e5, rgb9e5_to_float3", "yzw"),
    ("octa_decode", "xyz"),
    ("octa_wrap, max3", "xyz"),
    ("radical_inverse_vdc, hammersley", "xyz"),
    ("r2_sequence", "xy"),
    ("build_orthonormal_basis (column 0, column 1 .x)", "xyzw"),
    ("build_orthonormal_basis (column 1 .yz, column 2 .xy)", "xyzw"),
    ("uniform_sample_cone, build_orthonormal_basis (column 2 .z)", "xyzw"),
    ("uniform_sample_hemisphere, inverse_depth_relative_diff", "xyzw"),
    ("exponential_squish, exponential_unsquish", "xy"),
    ("sRGB_to_YCbCr, sRGB_to_luminance", "xyzw"),
    ("YCbCr_to_sRGB", "xyz"),
    ("Reservoir1spp::from_raw / update x 2 / as_raw", "z"),
    ("Reservoir1spp::init_with_stream / update_with_stream / finish_stream", "xyz"),
    ("SpecularBrdf::evaluate (value, pdf)", "xyzw"),
    ("SpecularBrdf::evaluate (value_over_pdf, transmission_fraction)", "xyzw"),
    ("SpecularBrdf::sample (wi, pdf)", "xyzw"),
    ("SpecularBrdf::sample (value_over_pdf, value)", "xyzw"),
    ("DiffuseBrdf::sample, DiffuseBrdf::evaluate", "xyzw"),
]


def _probe_inputs(n, seed):
    """uint4 per probe: three words that read as finite floats of moderate magnitude (2^-20 .. 2^20, either sign, random mantissa) and also serve as raw bits
    (hash inputs, packed words), one word of arbitrary bits; the first rows are the special values."""
    rng = np.random.default_rng(seed)
    u = np.zeros((n, 4), np.uint32)
    sign = rng.integers(0, 2, (n, 3), dtype=np.uint32) << 31
    expo = rng.integers(107, 148, (n, 3), dtype=np.uint32) << 23
    u[:, :3] = sign | expo | rng.integers(0, 1 << 23, (n, 3), dtype=np.uint32)
    u[:, 3] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    one, half = 0x3F800000, 0x3F000000
    special = [(one, 0, 0, 0), (0, one, 0, 0xFFFFFFFF), (0, 0, one, 0x80000000), (0, 0, one | 0x80000000, 1), (one, one, one, 0x7FFFFFFF),
               (half, half, half, 0x3FF), (one | 0x80000000, half, 0x3E800000, 0xFFE00000), (0x3F7FFFFF, 0x3F7FFFFF, 0x33800000, 0x001FFFFF)]
    u[:len(special)] = np.array(special, np.uint32)
    return u


@pytest.mark.parametrize("n", [4096, 1 << 18])
@recorded_case(lambda k: "inc_functions" if k["n"] == 4096 else None)
def test_inc_functions_reference_hlsl_vs_oracle(oracle, libm_sincos, n):
    """VERDICT r3 item 1(a), phase A: every leaf function of inc/*.hlsl the oracle restates -- hashes, the pack / unpack family, the quasi-random sequences, the
    sampling and basis helpers, the colour transforms, Reservoir1spp's methods, the specular and diffuse lobes -- evaluated by the reference's text (a probe pass of ours
    that includes the headers where they lie) and by the oracle (okj_probe_functions) on the same inputs (4096: recorded; 262144: live only), compared BIT FOR BIT, function by function."""
    if n != 4096:
        R.require_live()
    inp = _probe_inputs(n, 20260922 + n)
    rows = len(_PROBE_ROWS)
    out = np.zeros((rows, n, 4), np.uint32)
    R.run_pass("probes/inc_functions", [R.Buf(inp), R.Buf(out)], [np.uint32(n)], None, (n, 1, 1))
    ours = oracle.probe_functions(inp, rows)
    assert out.any(axis=(1, 2)).all(), "a row the probe never wrote"
    bad = []
    for r, (what, floats) in enumerate(_PROBE_ROWS):
        a, b = out[r], ours[r]
        same = a == b
        for c in floats:        # a NaN is a NaN whatever its payload
            ci = "xyzw".index(c)
            same[:, ci] |= np.isnan(a[:, ci].view(np.float32)) & np.isnan(b[:, ci].view(np.float32))
        if not same.all():
            i = int(np.argmin(same.all(axis=1)))
            bad.append((what, int((~same.all(axis=1)).sum()), i, [hex(v) for v in inp[i]], [hex(v) for v in a[i]], [hex(v) for v in b[i]]))
    assert not bad, bad


_COLOR_PROBE_ROWS = [
    ("sRGB_to_XYZ", "xyz"), ("XYZ_to_sRGB", "xyz"), ("CIE_XYZ_to_xyY", "xyz"), ("CIE_xyY_to_XYZ", "xyz"), ("XYZ_to_IPT", "xyz"), ("IPT_to_XYZ", "xyz"),
    ("CIE_xyY_xy_to_LUV_uv, CIE_XYZ_to_LUV_uv", "xyzw"), ("catmull_rom, compress_luminance", "xy"),
    ("XYZ_to_hk_luminance_multiplier_custom_g0, hk_from_sRGB, srgb_to_equivalent_luminance", "xyz"), ("XYZ_to_LAB, bb_xy_white_offset_to_lut_coord", "xyzw"),
    ("bezold_brucke_shift_XYZ_with_lut", "xyz"), ("display_transform_sRGB (a colour in [0, 1))", "xyz"), ("display_transform_sRGB (HDR)", "xyz"), ("display_transform_sRGB (any magnitude)", "xyz"),
    ("GbufferData::pack", ""), ("GbufferDataPacked::unpack (albedo, roughness)", "xyzw"), ("GbufferDataPacked::unpack (normal, metalness)", "xyzw"),
    ("GbufferDataPacked::unpack (emissive)", "xyz"), ("soft_color_clamp", "xyz"), ("get_uv (int2), get_uv (float2)", "xyzw"), ("cs_to_uv, uv_to_cs", "xyzw"),
    ("SphereIntersection, PhaseRayleigh, PhaseMie", "xyzw"), ("AtmosphereDensity, AtmosphereHeight", "xyzw"), ("IntegrateOpticalDepth", "xyz"), ("Absorb", "xyz"),
    ("IntegrateScattering", "xyz"),
]


@pytest.mark.parametrize("n", [4096, 1 << 17])
@recorded_case(lambda k: "inc_functions_color" if k["n"] == 4096 else None)
def test_inc_color_functions_reference_hlsl_vs_oracle(oracle, libm_sincos, n):
    """Phase A, second probe: the colour science of the display transform (inc/color/{srgb, xyz, ipt, luv, lab, math, helmholtz_kohlrausch, bezold_brucke,
    display_transform}.hlsl, the transform itself included, with a non-trivial Bezold-Brucke table), the G-buffer record (inc/gbuffer.hlsl), soft_color_clamp, inc/uv.hlsl
    and the sky model's functions (inc/atmosphere_felix.hlsl) -- the reference's text against the oracle's restatement on the same inputs, bit for bit, function by function."""
    from kajiya_amd import post_tables
    if n != 4096:
        R.require_live()
    lut = np.ascontiguousarray(post_tables.synthetic_bezold_brucke_lut(5), np.float16).reshape(64, 2)
    R.set_bindless(2, R.Tex(lut.copy(), 64, 1, "rg16f"))
    inp = _probe_inputs(n, 777 + n)
    rows = len(_COLOR_PROBE_ROWS)
    out = np.zeros((rows, n, 4), np.uint32)
    R.run_pass("probes/inc_functions_color", [R.Buf(inp), R.Buf(out)], [np.uint32(n)], None, (n, 1, 1))
    ours = oracle.probe_functions_color(inp, rows, lut)
    assert out.any(axis=(1, 2)).all(), "a row the probe never wrote"
    bad = []
    for r, (what, floats) in enumerate(_COLOR_PROBE_ROWS):
        a, b = out[r], ours[r]
        same = a == b
        for c in floats:        # a NaN is a NaN whatever its payload
            ci = "xyzw".index(c)
            same[:, ci] |= np.isnan(a[:, ci].view(np.float32)) & np.isnan(b[:, ci].view(np.float32))
        if not same.all():
            i = int(np.argmin(same.all(axis=1)))
            bad.append((what, int((~same.all(axis=1)).sum()), i, [hex(v) for v in inp[i]], a[i].view(np.float32).tolist(), b[i].view(np.float32).tolist()))
    assert not bad, bad


_SHADING_PROBE_ROWS = [
    ("ViewRayContext::from_uv: ray_dir_ws, ray_dir_vs", "xyzw"), ("ViewRayContext::ray_origin_ws", "xyz"), ("ViewRayContext::from_uv_and_depth: ray_hit_ws, ray_hit_vs", "xyzw"),
    ("biased_secondary_ray_origin_ws", "xyz"), ("biased_secondary_ray_origin_ws_with_normal", "xyz"), ("from_uv_and_biased_depth", "xyz"),
    ("get_eye_position, depth_to_view_z", "xyzw"), ("get_prev_eye_position, pixel_cone_spread_angle_from_image_height", "xyzw"), ("direction_view_to_world", "xyz"),
    ("direction_world_to_view", "xyz"), ("position_world_to_view", "xyz"), ("position_world_to_clip", "xyz"), ("position_world_to_sample", "xyz"),
    ("RayCone::propagate, width_at_t", "xyz"), ("metalness_albedo_boost", "xyz"), ("LayeredBrdf::from_gbuffer_ndotv: specular lobe", "xyzw"),
    ("from_gbuffer_ndotv: diffuse albedo, valid_sample_fraction", "xyzw"), ("SpecularBrdfEnergyPreservation: preintegrated_reflection", "xyz"),
    ("preintegrated_reflection_mult", "xyz"), ("preintegrated_transmission_fraction", "xyz"), ("LayeredBrdf::evaluate", "xyz"), ("LayeredBrdf::evaluate_directional_light", "xyz"),
    ("LayeredBrdf::sample (wi, pdf)", "xyzw"), ("LayeredBrdf::sample (value_over_pdf, value)", "xyzw"), ("sample_sun_direction", "xyz"), ("sun_color_in_direction", "xyz"),
    ("atmosphere_default", "xyz"), ("sample_triangle_light (pos, pdf)", "xyzw"), ("sample_triangle_light (normal), to_projected_solid_angle_measure", "xyzw"),
]


@pytest.mark.parametrize("n", [4096, 1 << 17])
@recorded_case(lambda k: "inc_functions_shading" if k["n"] == 4096 else None)
def test_inc_shading_functions_reference_hlsl_vs_oracle(oracle, libm_sincos, n):
    """Phase A, third probe: the view-ray helpers of inc/frame_constants.hlsl under a real camera (jittered, with a previous frame), inc/ray_cone.hlsl, the layered BRDF and
    its energy preservation off the BRDF table (inc/layered_brdf.hlsl, inc/brdf_lut.hlsl), inc/sun.hlsl, inc/atmosphere.hlsl and the triangle-light sampler
    (inc/lights/triangle.hlsl) -- the reference's text against the oracle's restatement, bit for bit, function by function."""
    if n != 4096:
        R.require_live()
    _bind_luts(oracle)
    fc = _frame_constants(320, 180, 3, "city")[2]
    inp = _probe_inputs(n, 4242 + n)
    rows = len(_SHADING_PROBE_ROWS)
    out = np.zeros((rows, n, 4), np.uint32)
    R.run_pass("probes/inc_functions_shading", [R.Buf(inp), R.Buf(out)], [np.uint32(n)], fc, (n, 1, 1))
    ours = oracle.probe_functions_shading(fc, inp, rows)
    assert out.any(axis=(1, 2)).all(), "a row the probe never wrote"
    bad = []
    for r, (what, floats) in enumerate(_SHADING_PROBE_ROWS):
        a, b = out[r], ours[r]
        same = a == b
        for c in floats:        # a NaN is a NaN whatever its payload
            ci = "xyzw".index(c)
            same[:, ci] |= np.isnan(a[:, ci].view(np.float32)) & np.isnan(b[:, ci].view(np.float32))
        if not same.all():
            i = int(np.argmin(same.all(axis=1)))
            bad.append((what, int((~same.all(axis=1)).sum()), i, [hex(v) for v in inp[i]], a[i].view(np.float32).tolist(), b[i].view(np.float32).tolist()))
    assert not bad, bad


_MISC_PROBE_ROWS = [
    ("taa decode_rgb", "xyz"), ("taa encode_rgb", "xyz"), ("get_bilinear_filter", "xyzw"), ("TemporalReservoirOutput::from_raw -> as_raw", ""),
    ("TemporalReservoirOutput: depth, ray_hit_offset_ws", "xyzw"), ("TemporalReservoirOutput: luminance, hit_normal_ws", "xyzw"),
    ("SampleParams: raw, rng, octa_uv", "zw"), ("SampleParams::direction, octa_idx", "xyz"), ("ws_pos_to_ircache_coord", ""),
    ("IrcacheCoord::cell_idx, ws_local_pos_to_cascade_idx, ircache_grid_cell_diameter_in_cascade", "z"),
]


@pytest.mark.parametrize("n", [4096, 1 << 17])
@recorded_case(lambda k: "inc_functions_misc" if k["n"] == 4096 else None)
def test_misc_functions_reference_hlsl_vs_oracle(oracle, libm_sincos, n):
    """Phase A, fourth probe: the helpers that live next to the passes -- taa/taa_common.hlsl, inc/bilinear.hlsl, rtdgi/rtdgi_common.hlsl, the cache's sample parameters and
    its grid addressing under a frame's cascades (positions from centimetres to kilometres around the grid centre) -- reference text against oracle, bit for bit."""
    if n != 4096:
        R.require_live()
    fc = _frame_constants(320, 180, 3, "city")[2]
    inp = _probe_inputs(n, 1234 + n)
    rows = len(_MISC_PROBE_ROWS)
    out = np.zeros((rows, n, 4), np.uint32)
    R.run_pass("probes/inc_functions_misc", [R.Buf(inp), R.Buf(out)], [np.uint32(n)], fc, (n, 1, 1))
    ours = oracle.probe_functions_misc(fc, inp, rows)
    assert out.any(axis=(1, 2)).all(), "a row the probe never wrote"
    bad = []
    for r, (what, floats) in enumerate(_MISC_PROBE_ROWS):
        a, b = out[r], ours[r]
        same = a == b
        if r == 3:              # words y, z are pairs of halves that went half -> float -> half: a NaN half stays a NaN, its payload is the conversion's business
            ha, hb = a[:, 1:3].copy().view(np.uint16), b[:, 1:3].copy().view(np.uint16)
            nan = lambda h: ((h & 0x7c00) == 0x7c00) & ((h & 0x3ff) != 0)
            ok = (ha == hb) | (nan(ha) & nan(hb))
            same[:, 1:3] = ok.reshape(-1, 2, 2).all(axis=2)
        for c in floats:        # a NaN is a NaN whatever its payload
            ci = "xyzw".index(c)
            same[:, ci] |= np.isnan(a[:, ci].view(np.float32)) & np.isnan(b[:, ci].view(np.float32))
        if not same.all():
            i = int(np.argmin(same.all(axis=1)))
            bad.append((what, int((~same.all(axis=1)).sum()), i, [hex(v) for v in inp[i]], [hex(v) for v in a[i]], [hex(v) for v in b[i]]))
    assert not bad, bad
    assert len(np.unique(out[8][:, 3])) >= 10, "the positions reach fewer than ten of the twelve cascades"
