"""CPU checks of the rtr restatement (oracle/okj_rtr.hpp). The reference holds no vectors for this path; the oracle's six
passes are held to the reference's shader text in tests/test_ref_hlsl.py, and here to properties: B10G11R11 packing known answers, the sampler-table arithmetic, and the
physical invariant that a smooth metal mirror under the sky resolves to the sky radiance along the mirrored view direction
(rtr output is "radiance not scaled by FG", rtr_settings.hlsl:7)."""
import ctypes as C

import numpy as np
import pytest

import parity as P
import test_gpu_parity as T
from kajiya_amd import rtr_tables, scenes as S


def test_standin_tables_have_reference_shapes_and_ranges():
    t, (ranking, scrambling, sobol, offsets) = rtr_tables.standin_tables()
    assert ranking.shape == (128 * 128 * 8,) and scrambling.shape == (128 * 128 * 8,) and sobol.shape == (256 * 256,)
    assert ranking.max() < 64 and scrambling.max() < 256 and sobol.max() < 256
    # Sobol: the first point is the origin; dimension 0 of the first 256 points is the van der Corput sequence (a permutation of 0..255)
    s = sobol.reshape(256, 256)
    assert (s[0] == 0).all() and sorted(s[:, 0]) == list(range(256)) and s[1, 0] == 128
    o = offsets.reshape(8, 4, 16, 4)
    assert (o[:, :, 0, :2] == 0).all() and (o[..., 2:] == 0).all()
    for f in range(8):
        pts = {tuple(p) for q in range(4) for p in o[f, q, 1:, :2]}
        assert len(pts) == 60 and (0, 0) not in pts                       # disjoint between the four quad variants
        d = (o[f, :, :, 0] ** 2 + o[f, :, :, 1] ** 2)
        assert (np.diff(d, axis=1) >= 0).all()                            # taps sorted by distance (spatial_cleanup takes the first n)


def test_r11g11b10_known_answers():
    """B10G11R11_UFLOAT: 5-bit exponent (bias 15), 6/6/5-bit mantissas, no sign. 1.0 = e15 m0; 0.5 = e14; 65024 = max (e30 m63)."""
    def dec(u):
        return P.decode(np.array([u], np.uint32).view(np.uint8), "r11g11b10f")[0]
    one11, one10 = 15 << 6, 15 << 5
    np.testing.assert_array_equal(dec(one11 | (one11 << 11) | (one10 << 22)), [1, 1, 1])
    np.testing.assert_array_equal(dec((14 << 6) | ((16 << 6 | 32) << 11) | ((15 << 5 | 16) << 22)), [0.5, 3.0, 1.5])
    np.testing.assert_array_equal(dec(0x7bf), [65024.0, 0, 0])


def _mirror_vs_sky(oracle, frames, W, H):
    sd = S.glossy_test_scene()
    osc = oracle.OracleScene(sd)
    op = oracle.OraclePipeline(osc, W, H)
    fcs = T._frame_constants(W, H, frames, "textured")
    for fc in fcs:
        op.frame(fc)
        res = op.rtr_frame(fc)
    img = P.decode(res.copy().view(np.uint8), "r11g11b10f").reshape(H, W, 3)
    return op, osc, fcs[-1], img


def mirror_vs_sky_ratio(osc, fc, depth_img, sky64, img):
    """rtr image / sky radiance along the mirrored view direction, for floor pixels (y = 0 plane of glossy_test_scene) whose mirror
    ray leaves the scene. Returns the per-pixel ratio array."""
    H, W = depth_img.shape
    vc = fc.view_constants

    def m44(a):
        return np.array(list(a), np.float64).reshape(4, 4).T
    s2v, v2w = m44(vc.sample_to_view), m44(vc.view_to_world)
    ys, xs = np.mgrid[0:H, 0:W]
    cs = np.stack([((xs + 0.5) / W - 0.5) * 2, ((ys + 0.5) / H - 0.5) * -2], -1)
    depth = depth_img.astype(np.float64)
    valid = depth > 0
    hit = np.concatenate([cs, depth[..., None], np.ones_like(depth)[..., None]], -1) @ s2v.T @ v2w.T
    hit_ws = hit[..., :3] / np.where(valid, hit[..., 3], 1.0)[..., None]
    eye = (np.array([0, 0, 0, 1.0]) @ v2w.T)[:3]
    vdir = hit_ws - eye
    vdir /= np.maximum(np.linalg.norm(vdir, axis=-1, keepdims=True), 1e-12)
    floor = valid & (np.abs(hit_ws[..., 1]) < 2e-3)
    assert floor.sum() > 2000
    n = np.array([0.0, 1.0, 0.0])
    rdir = vdir - 2 * (vdir @ n)[..., None] * n
    rays = np.zeros((int(floor.sum()), 8), np.float32)
    rays[:, :3] = hit_ws[floor] + n * 1e-3
    rays[:, 4:7] = rdir[floor]
    rays[:, 7] = 1e4
    miss = osc.trace_closest(rays)[:, 0] > 1e30
    assert miss.sum() > 500
    sky = sky64.view(np.float16).astype(np.float32).reshape(6, 64, 64, 4)
    d = rdir[floor][miss]
    ax, ay, az = np.abs(d[:, 0]), np.abs(d[:, 1]), np.abs(d[:, 2])
    face = np.where((az >= ax) & (az >= ay), np.where(d[:, 2] >= 0, 4, 5), np.where(ay >= ax, np.where(d[:, 1] >= 0, 2, 3), np.where(d[:, 0] >= 0, 0, 1)))
    sc = np.choose(face, [-d[:, 2], d[:, 2], d[:, 0], d[:, 0], d[:, 0], -d[:, 0]])
    tc = np.choose(face, [-d[:, 1], -d[:, 1], d[:, 2], -d[:, 2], -d[:, 1], -d[:, 1]])
    ma = np.choose(face, [ax, ax, ay, ay, az, az])
    px = np.clip((0.5 * (sc / ma + 1) * 64).astype(int), 0, 63)
    py = np.clip((0.5 * (tc / ma + 1) * 64).astype(int), 0, 63)
    expect = sky[face, py, px, :3]
    got = img[floor][miss]
    ratio = got.sum(-1) / np.maximum(expect.sum(-1), 1e-6)
    return ratio


def test_rtr_mirror_floor_reflects_the_sky(oracle):
    W, H = 256, 160
    op, osc, fc, img = _mirror_vs_sky(oracle, 10, W, H)
    assert np.isfinite(img).all()
    valid = op.depth > 0
    ratio = mirror_vs_sky_ratio(osc, fc, op.depth, op.sky64, img)
    print("rtr / sky on sky-reflecting mirror pixels: median %.3f p10 %.3f p90 %.3f" % (np.median(ratio), np.percentile(ratio, 10), np.percentile(ratio, 90)))
    assert 0.95 < np.median(ratio) < 1.06 and np.percentile(ratio, 10) > 0.85 and np.percentile(ratio, 90) < 1.3
    # temporal accumulation reached steady state and the rough wall (roughness 0.8) got its rays from rtdgi's candidates
    cnt = op.rtr_surface("rtr.temporal:0", np.float16, (H, W, 4))[..., 3].astype(np.float32)
    assert cnt[valid].mean() > 6
    closest, anyhit = op.rtr_ray_counts()
    assert 0 < closest < 10 * (W // 2) * (H // 2) * 1.3                      # <= one trace ray + 1/4 validation ray per half-res pixel per frame


def test_rtr_pass_by_pass_equals_whole_frame(oracle):
    """The pass-mask / KEEP machinery the GPU parity test relies on: running the six passes one at a time leaves exactly the
    state of one whole-frame call."""
    W, H = 96, 64
    sd = S.glossy_test_scene()
    a = oracle.OraclePipeline(oracle.OracleScene(sd), W, H)
    b = oracle.OraclePipeline(oracle.OracleScene(sd), W, H)
    names = [n + s for n in ("rtr.temporal", "rtr.ray_len", "rtr.irradiance", "rtr.ray_orig", "rtr.ray", "rtr.reservoir", "rtr.rng", "rtr.hit_normal") for s in (":0", ":1")]
    names += ["refl_restir_invalidity_tex", "resolved_tex"]
    for fc in T._frame_constants(W, H, 5, "textured"):
        a.frame(fc); b.frame(fc)
        a.rtr_frame(fc)
        for k, m in enumerate((1, 2, 4, 8, 16, 32)):
            b.rtr_frame(fc, m | (0 if k == 0 else 0x80000000))
        for n in names:
            np.testing.assert_array_equal(a.rtr_surface(n, np.uint8, (-1,)), b.rtr_surface(n, np.uint8, (-1,)), err_msg=n)
        for n in ("candidate_radiance_tex", "candidate_hit_tex", "candidate_normal_tex"):
            np.testing.assert_array_equal(a.surface(n, np.uint8, (-1,)), b.surface(n, np.uint8, (-1,)), err_msg=n)


def test_rtr_oracle_matches_golden_vectors(oracle):
    """tests/golden/oracle_rtr_glossy_48x32.npz (scripts/make_golden_vectors.py) pins the rtr restatement against silent changes
    (it is the oracle's own output: the reference holds no vectors for this path)."""
    import importlib.util, os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("make_golden_vectors", os.path.join(root, "scripts", "make_golden_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.generate_rtr()
    ref = np.load(os.path.join(root, "tests", "golden", "oracle_rtr_glossy_48x32.npz"))
    assert np.array_equal(got["ray_counts"], ref["ray_counts"])
    assert (got["rng"] != ref["rng"]).mean() < 0.01 and (got["reservoir"] != ref["reservoir"]).any(axis=-1).mean() < 0.01
    for k in ("temporal", "irradiance"):
        a, b = got[k].view(np.float16).astype(np.float32), ref[k].view(np.float16).astype(np.float32)
        assert np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()) < 1e-3, k
    r = P.compare(got["resolved"].view(np.uint8).reshape(-1), ref["resolved"].view(np.uint8).reshape(-1), "r11g11b10f")
    assert r["mismatch_frac"] < 0.01, r


@pytest.mark.parametrize("W,H", [(2, 2), (7, 5), (33, 17)])
def test_rtr_oracle_tiny_and_ragged_extents(oracle, W, H):
    """Extents that are odd, smaller than a tile or than the quarter-res validation grid: every pass must run, stay finite and touch only its
    own images (the reference's dispatches are rounded up to 8x8 groups with out-of-range stores dropped)."""
    sd = S.glossy_test_scene()
    op = oracle.OraclePipeline(oracle.OracleScene(sd), W, H)
    for i, fc in enumerate(T._frame_constants(W, H, 4, "textured")):
        op.frame(fc)
        if i == 2:
            op.L.okj_rtr_set_options(op.rtr, 0)        # every pixel traces its own reflection ray from here on
        res = op.rtr_frame(fc)
        img = P.decode(res.copy().view(np.uint8), "r11g11b10f")
        assert res.shape == (H, W) and np.isfinite(img).all()
        hw, hh = (W + 1) // 2, (H + 1) // 2
        for name, bpt in (("rtr.irradiance:0", 8), ("rtr.ray_orig:0", 16), ("rtr.reservoir:1", 8), ("rtr.rng:0", 4), ("refl_restir_invalidity_tex", 1)):
            assert op.rtr_surface(name, np.uint8, (-1,)).size == hw * hh * bpt, name
        t = op.rtr_surface("rtr.temporal:0", np.float16, (H, W, 4)).astype(np.float32)
        assert np.isfinite(t).all()
