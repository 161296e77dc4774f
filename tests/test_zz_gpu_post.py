"""PostProcessRenderer (SURVEY 8f-4 "minimal post"): the HIP kernels through the C-ABI vs the oracle, on identical inputs.

Verified on hardware since round 2 (the kernel source executed on the CPU also reproduces the oracle bit for bit,
tests/test_post_emulation.py, and the oracle is pinned against the Rust text of the kernels that run in the reference,
tests/test_post_oracle.py). The file sorts last for historical reasons.

Tolerances: every image here is B10G11R11_UFLOAT. GPU and CPU differ by rounding noise (FMA contraction, ocml's exp / pow / log2), which a
6- or 5-bit mantissa turns into whole-step flips at rounding boundaries; a later mip reads those flips, so "no texel further than TWO
storage steps, nearly all identical" is the statement of parity for a free-running pyramid; the final image gets the same bound."""
import numpy as np
import pytest

import parity as P
from kajiya_amd import frame, post_tables

pytestmark = pytest.mark.gpu

STEP = np.array([1 / 64, 1 / 64, 1 / 32])


def _fc(W, H, frame_index, pre_exposure):
    fs = frame.FrameState((W, H))
    fs.frame_idx = frame_index
    fs.pre_exposure = pre_exposure
    return fs.prepare_frame_constants(frame.orbit_camera(0, (W, H)))


def _compare(got_words, ref_words, what, max_differ, singular_texels=0):
    """`singular_texels`: how many texels may sit on the display transform's own singularity (the final image only). The second Helmholtz-
    Kohlrausch evaluation of display_transform.hlsl (inc/color/helmholtz_kohlrausch.hlsl:53-106) maps the hue angle to t = theta / pi * 0.5 + 0.5
    and indexes its 16 spline samples with uint(floor(t * 16)) % 16; a colour whose CIE LUV v' equals the white point's EXACTLY, on the cyan
    side, has theta = pi, t = 1.0, index 16 % 16 = 0 and a spline parameter of 16 instead of 0..1: the extrapolated multiplier is 27 instead of
    1.04 and the pixel comes out 24x darker. With fp32 chromaticities that is ~3e-7 of all colours -- about one pixel of a 1080p frame, in the
    reference too -- and whether a given texel is on it depends on the last bit of everything upstream (measured on MI355X: 2 of 2 M texels
    of the 1080p case, one on each side; oracle/okj_post.hpp and post.hip follow the text: DESIGN.md section 4)."""
    a = P.decode(np.ascontiguousarray(got_words).view(np.uint8), "r11g11b10f").astype(np.float64)
    b = P.decode(np.ascontiguousarray(ref_words).view(np.uint8), "r11g11b10f").astype(np.float64)
    assert np.isfinite(a).all() and np.isfinite(b).all(), what
    err = np.abs(a - b)
    tol = np.maximum(np.abs(b), 2.0 ** -14) * STEP
    beyond = (err > 2.05 * tol + 1e-9).any(-1)
    if beyond.sum() > singular_texels:
        bad = np.argwhere(beyond).reshape(-1)
        w = int(np.argmax((err / tol).max(-1)))
        raise AssertionError((what, float((err / tol).max()), f"{bad.size} texels beyond two steps, first {bad[:8]}, worst texel {w}: got {a[w]} ref {b[w]}"))
    if beyond.any():
        print(f"{what}: {int(beyond.sum())} texel(s) on the display transform's singularity: {np.argwhere(beyond).reshape(-1)[:4]}")
    err, a, b = err[~beyond], a[~beyond], b[~beyond]
    differ = float((err > 0).any(-1).mean())
    assert differ <= max_differ, (what, differ)
    return differ


@pytest.mark.parametrize("W,H,frame_index,mult,contrast,lut_seed", [(160, 90, 3, 1.3, 1.1, 0), (333, 187, 0, 1.0, 1.0, None), (1920, 1080, 5, 0.8, 1.0, 1)])
def test_post_matches_oracle(gpu, oracle, device, W, H, frame_index, mult, contrast, lut_seed):
    import torch
    lut = post_tables.zero_bezold_brucke_lut() if lut_seed is None else post_tables.synthetic_bezold_brucke_lut(lut_seed)
    rng = np.random.RandomState(W + H)
    ys, xs = np.mgrid[0:H, 0:W]
    img = np.stack([0.5 + 0.5 * np.sin(xs * 0.05), 0.5 + 0.5 * np.cos(ys * 0.07), 0.5 + 0.5 * np.sin((xs + ys) * 0.03)], -1) * rng.uniform(0.2, 3.0, (H, W, 1))
    for _ in range(W * H // 2000):
        img[rng.randint(H), rng.randint(W)] = rng.uniform(20, 900, 3)
    img[: H // 6, : W // 5] = 0.0                                    # black: the NaN-chromaticity path
    inp = np.concatenate([img, np.ones((H, W, 1))], -1).astype(np.float16)
    fc = _fc(W, H, frame_index, 0.7)
    op = oracle.OraclePost(lut)
    ref = op.render(fc, inp, mult, contrast).copy()

    gp = gpu.GpuPost(device, lut)
    device.frame_begin(fc)
    got = gp.render(torch.from_numpy(inp).cuda().contiguous(), mult, contrast)
    torch.cuda.synchronize()
    assert gp.mip_levels() == op.mip_levels()
    worst = {}
    for l in range(op.mip_levels()):
        w, h = op.mip_extent(l)
        assert gp.mip_extent(l) == (w, h)
        for pyr in ("blur_pyramid", "rev_blur_pyramid"):
            g = gp.surface(f"{pyr}:{l}", torch.int32, (h, w)).cpu().numpy().view(np.uint32)
            worst[pyr] = max(worst.get(pyr, 0.0), _compare(g, op.mip(pyr, l), f"{pyr}:{l}", 0.03 if w * h > 500 else 0.2))
    worst["output"] = _compare(got.cpu().numpy().view(np.uint32), ref, "output", 0.05, singular_texels=max(1, int(4e-6 * W * H)))
    hist_g = gp.surface("histogram", torch.int32, (256,)).cpu().numpy().view(np.uint32).astype(np.int64)
    hist_o = op.histogram().astype(np.int64)
    assert abs(int(hist_g.sum()) - int(hist_o.sum())) <= 0.002 * hist_o.sum()
    assert np.abs(np.cumsum(hist_g) - np.cumsum(hist_o)).max() <= 0.004 * hist_o.sum()
    lum, hist_rb = gp.read_back_histogram(0.1, 0.2)                  # the stream is synchronised: this frame's copy
    assert np.array_equal(hist_rb.astype(np.int64), hist_g)
    assert abs(lum - op.read_back_histogram(hist_o, 0.1, 0.2)) < 0.05
    print(f"{W}x{H}: texels differing (by <= 2 storage steps): " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))


def test_post_rgba32f_input_matches_oracle(gpu, oracle, device):
    """The reference mode's input: the path tracer's RGBA32F accumulation image (world_render_passes.rs:294-330)."""
    import torch
    W, H = 320, 200
    lut = post_tables.synthetic_bezold_brucke_lut(5)
    rng = np.random.RandomState(3)
    acc = (rng.uniform(0, 1, (H, W, 4)) ** 4 * 50).astype(np.float32)
    acc[30, 40, :3] = 3.0e5
    fc = _fc(W, H, 4, 1.0)
    op = oracle.OraclePost(lut)                                     # keep it alive: render() returns a view of its memory
    ref = op.render(fc, acc, 0.25, 1.0).copy()
    gp = gpu.GpuPost(device, lut)
    device.frame_begin(fc)
    got = gp.render(torch.from_numpy(acc).cuda().contiguous(), 0.25, 1.0)
    torch.cuda.synchronize()
    _compare(got.cpu().numpy().view(np.uint32), ref, "output (rgba32f input)", 0.05)


def test_post_is_deterministic_and_reusable_across_extents(gpu, device):
    """Same input twice -> identical words (the only atomics are integer adds); a different extent on the same handle reallocates."""
    import torch
    lut = post_tables.synthetic_bezold_brucke_lut(2)
    gp = gpu.GpuPost(device, lut)
    outs = []
    for (W, H) in ((256, 144), (256, 144), (100, 60), (256, 144)):
        inp = (np.random.RandomState(1).uniform(0, 2, (H, W, 4))).astype(np.float16)
        device.frame_begin(_fc(W, H, 1, 1.0))
        o = gp.render(torch.from_numpy(inp).cuda().contiguous())
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy().copy())
        h = gp.surface("histogram", torch.int32, (256,)).cpu().numpy().copy()
        assert h.sum() > 0
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[3]) and outs[2].shape == (60, 100)


@pytest.mark.parametrize("W,H,DW,DH", [(320, 180, 320, 180), (333, 187, 333, 187), (640, 360, 320, 180)])
def test_motion_blur_matches_oracle(gpu, oracle, device, W, H, DW, DH):
    """motion_blur (renderers/motion_blur.rs): the three velocity selections are exact (they only compare and copy fp16 values); the blurred
    image is RGBA16F: fp16 rounding noise, except where rounding moves a tap across a texel edge (`as_uvec2`) — a handful of pixels."""
    import torch
    from test_post_emulation import _motion_inputs
    inp, depth, rm = _motion_inputs(W, H, DW, DH, W + DH)
    fc = _fc(DW, DH, 2, 1.0)
    om = oracle.OracleMotionBlur()
    ref = om.render(fc, inp, depth, rm).astype(np.float32)
    gm = gpu.GpuMotionBlur(device)
    device.frame_begin(fc)
    got = gm.render(torch.from_numpy(inp).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(rm).cuda())
    torch.cuda.synchronize()
    tw, th = (DW + 15) // 16, (DH + 15) // 16
    for name, shape in (("velocity_reduced_x", (DH, tw, 2)), ("velocity_reduced_y", (th, tw, 2)), ("velocity_dilated", (th, tw, 2))):
        g = gm.surface(name, torch.int16, shape).cpu().numpy().view(np.uint16)
        assert np.array_equal(g, om.surface(name, np.uint16, shape)), name
    g = got.float().cpu().numpy()
    assert np.isfinite(g).all() and (g[..., 3] == 1.0).all()
    err = np.abs(g[..., :3] - ref[..., :3])
    close = (err <= np.abs(ref[..., :3]) * 2.0 ** -9 + 2e-4).all(-1)
    rel = float(np.sqrt((err ** 2).sum() / (ref[..., :3] ** 2).sum()))
    print(f"{W}x{H}: rel-L2 {rel:.2e}, pixels beyond fp16 rounding {1 - close.mean():.2e}")
    assert close.mean() > 0.998 and rel < 1e-3
