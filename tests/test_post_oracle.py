"""Oracle checks for the post-processing path (oracle/okj_post.hpp <-> renderers/post.rs, rust-shaders/src/{blur,rev_blur}.rs,
shaders/{blur,post_combine}.hlsl, shaders/post/luminance_histogram_*.hlsl, inc/color/*.hlsl). No GPU.

Pinning: for this path two of the kernels that RUN in the reference are plain Rust (`blur::blur_cs` for mip 0 of the blur pyramid,
`rev_blur::rev_blur_cs`) and a third exists in HLSL only but shares its text with the Rust one (blur.hlsl, one more vertical tap). They
are restated here in float64 numpy straight from the Rust text and the oracle's images must equal them up to the storage format's
rounding; the histogram read-back is host Rust restated in Python; the display transform has no second statement and is checked through
the properties its design notes promise (achromatic stays achromatic, monotonic tone curve, output inside the display gamut)."""
import ctypes as C
import math

import numpy as np
import pytest

import parity as P
from kajiya_amd import frame, post_tables


def _fc(W, H, frame_index=0, pre_exposure=1.0):
    fs = frame.FrameState((W, H))
    fs.frame_idx = frame_index
    fs.pre_exposure = pre_exposure
    return fs.prepare_frame_constants(frame.orbit_camera(0, (W, H)))


def _hdr_image(W, H, seed=0):
    """Smooth gradients + a few very bright texels + a black region: what a lit frame looks like to the pyramid."""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    img = np.stack([0.5 + 0.5 * np.sin(xs * 0.21 + 0.3), 0.5 + 0.5 * np.cos(ys * 0.17), 0.5 + 0.5 * np.sin((xs + ys) * 0.11)], -1) * 0.8
    img *= rng.uniform(0.6, 1.4, (H, W, 1))
    for _ in range(max(2, W * H // 400)):
        img[rng.randint(H), rng.randint(W)] = rng.uniform(20, 300, 3)
    img[: H // 5, : W // 4] = 0.0
    return np.concatenate([img, np.ones((H, W, 1))], -1).astype(np.float16)


def _unpack(words):
    return P.decode(np.ascontiguousarray(words).view(np.uint8), "r11g11b10f").astype(np.float64).reshape(words.shape + (3,))


def _assert_stored(got_words, ref, what):
    """`ref` (float64, unquantised) against a B10G11R11 image: every texel within one storage step, and nearly all within half a step."""
    got = _unpack(got_words)
    step = np.array([1 / 64, 1 / 64, 1 / 32])
    err = np.abs(got - ref)
    tol = np.maximum(np.abs(ref), 2.0 ** -14) * step
    assert (err <= tol * 1.001 + 1e-9).all(), (what, float((err / tol).max()))
    near = err <= tol * 0.5 * 1.02 + 1e-9            # float32 vs float64 may round a near-tie the other way
    assert near.mean() > 0.99 or (~near).sum() <= 2, (what, float(near.mean()))


def _gaussian_wt(dst_px, src_px):           # blur.rs:18-22
    px_off = (dst_px + 0.5) * 2.0 - (src_px + 0.5)
    sigma = 5 * 0.5
    return math.exp(-px_off * px_off / (sigma * sigma))


def _rust_blur(src, dw, dh, vtaps, uint_coords=False):
    """blur_cs (rust-shaders/src/blur.rs:41-91): vblur of 138 columns into shared memory, then 11 horizontal taps. `vtaps` = 10 for the
    Rust text (`while y < KERNEL_RADIUS * 2`), 11 for blur.hlsl (`y <= kernel_radius * 2`). Out-of-range fetches read 0.
    `uint_coords`: blur.hlsl's other difference -- the tap's source coordinate handed to gaussian_wt is a `uint` expression there (blur.hlsl:22,50;
    i32 in blur.rs:30,83), so a tap left of / above the image sits at 2^32 - k and weighs exactly 0 instead of counting with a zero texel."""
    coord = (lambda s: float(s % (1 << 32))) if uint_coords else float
    sh, sw = src.shape[:2]
    pad = np.zeros((sh + 2 * dh + 32, sw + 2 * dw + 32, 3))
    oy, ox = 8, 8
    pad[oy:oy + sh, ox:ox + sw] = src
    # vertical pass for every source column that any output column can touch
    cols = np.arange(-5, 2 * dw + 6)
    v = np.zeros((dh, len(cols), 3))
    for y in range(dh):
        acc, wsum = 0.0, 0.0
        for yi in range(vtaps):
            sy = y * 2 - 5 + yi
            wt = _gaussian_wt(y, coord(sy))
            acc = acc + pad[oy + sy, ox + cols[0]: ox + cols[-1] + 1] * wt
            wsum += wt
        v[y] = acc / wsum
    out = np.zeros((dh, dw, 3))
    for x in range(dw):
        acc, wsum = 0.0, 0.0
        for xi in range(11):
            sx = x * 2 + xi - 5
            wt = _gaussian_wt(x, coord(sx))
            acc = acc + v[:, sx + 5] * wt
            wsum += wt
        out[:, x] = acc / wsum
    return out


@pytest.mark.parametrize("W,H", [(70, 50), (129, 33)])
def test_blur_pyramid_matches_the_rust_and_hlsl_statements(oracle, W, H):
    op = oracle.OraclePost(post_tables.zero_bezold_brucke_lut())
    inp = _hdr_image(W, H)
    op.render(_fc(W, H), inp)
    levels = op.mip_levels()
    pw, ph = (W + 1) // 2, (H + 1) // 2
    assert levels == max(1, max(pw.bit_length(), ph.bit_length()) - 1)          # all_mip_levels() - 1 (post.rs:11-21, image.rs:35-38,115-119)
    assert [op.mip_extent(l) for l in range(levels)] == [(max(1, pw >> l), max(1, ph >> l)) for l in range(levels)]
    # mip 0: the Rust kernel on the RGBA16F input
    w0, h0 = op.mip_extent(0)
    _assert_stored(op.mip("blur_pyramid", 0), _rust_blur(inp[..., :3].astype(np.float64), w0, h0, 10), "blur mip 0")
    # the 10-tap / 11-tap difference is visible: an 11-tap restatement of mip 0 must NOT fit
    eleven = _rust_blur(inp[..., :3].astype(np.float64), w0, h0, 11)
    assert (np.abs(_unpack(op.mip("blur_pyramid", 0)) - eleven) > np.abs(eleven) / 32 + 1e-6).mean() > 0.05
    # mips 1..: blur.hlsl on the previous (stored) mip
    for l in range(1, levels):
        wl, hl = op.mip_extent(l)
        _assert_stored(op.mip("blur_pyramid", l), _rust_blur(_unpack(op.mip("blur_pyramid", l - 1)), wl, hl, 11, uint_coords=True), f"blur mip {l}")


def _bilinear_clamp(img, u, v):
    h, w = img.shape[:2]
    fx, fy = u * w - 0.5, v * h - 0.5
    x0, y0 = math.floor(fx), math.floor(fy)
    tx, ty = fx - x0, fy - y0
    cl = lambda a, n: min(max(a, 0), n - 1)
    s00, s10 = img[cl(y0, h), cl(x0, w)], img[cl(y0, h), cl(x0 + 1, w)]
    s01, s11 = img[cl(y0 + 1, h), cl(x0, w)], img[cl(y0 + 1, h), cl(x0 + 1, w)]
    return (s00 * (1 - tx) + s10 * tx) * (1 - ty) + (s01 * (1 - tx) + s11 * tx) * ty


def test_rev_blur_pyramid_matches_the_rust_statement(oracle):
    """rev_blur_cs (rust-shaders/src/rev_blur.rs:20-72) per level, fed with the oracle's own coarser level; post.rs:73-77 always passes
    self_weight = 0.5 (src_mip never equals mip_levels), so every level is 0.7 * box(coarser) + 0.3 * blur pyramid; the coarsest level is
    the never-written one (chosen: zeros)."""
    W, H = 96, 64
    op = oracle.OraclePost(post_tables.zero_bezold_brucke_lut())
    op.render(_fc(W, H), _hdr_image(W, H, 1))
    levels = op.mip_levels()
    assert not op.mip("rev_blur_pyramid", levels - 1).any()
    for target in range(levels - 2, -1, -1):
        tail, src = _unpack(op.mip("blur_pyramid", target)), _unpack(op.mip("rev_blur_pyramid", target + 1))
        w, h = op.mip_extent(target)
        ref = np.zeros((h, w, 3))
        for y in range(h):
            for x in range(w):
                acc = 0.0
                for yy in (-1, 0, 1):
                    for xx in (-1, 0, 1):
                        acc = acc + _bilinear_clamp(src, (x + 0.5 + xx) / w, (y + 0.5 + yy) / h)
                self_col = acc / 9.0
                t = 0.5 * 0.6
                ref[y, x] = self_col * (1.0 - t) + tail[y, x] * t
        _assert_stored(op.mip("rev_blur_pyramid", target), ref, f"rev blur mip {target}")


def _read_back_histogram(hist, lo, hi):
    """PostProcessRenderer::read_back_histogram (post.rs:188-235) in Python ints / floats (= Rust u32 / f64)."""
    lo = min(float(np.float32(lo)), 1.0)
    hi = min(float(np.float32(hi)), 1.0 - lo)
    total = int(np.sum(hist, dtype=np.uint64)) & 0xffffffff
    left_to_reject = int(total * lo)
    left_to_use = entry_count_to_use = int(total * (1.0 - lo - hi))
    s, used = 0.0, 0
    for i, count in enumerate(int(c) for c in hist):
        t = (i + 0.5) / 256.0
        count_to_use = min(max(count - left_to_reject, 0), left_to_use)
        left_to_reject = max(left_to_reject - count, 0)
        left_to_use = max(left_to_use - count_to_use, 0)
        s += t * count_to_use
        used += count_to_use
    assert used == entry_count_to_use         # the Rust asserts it too
    mean = s / max(used, 1)
    return np.float32(-16.0 + mean * 32.0)


def test_luminance_histogram_and_read_back(oracle):
    from kajiya_amd import lib
    W, H = 320, 200                         # 160 x 100 base, 7 levels: the histogram reads mip 0 (levels - 7)
    op = oracle.OraclePost(post_tables.zero_bezold_brucke_lut())
    pre_exposure = 0.5
    op.render(_fc(W, H, pre_exposure=pre_exposure), _hdr_image(W, H, 2))
    levels = op.mip_levels()
    assert levels == 7
    level = max(0, levels - 7)
    pw, ph = (W + 1) // 2, (H + 1) // 2
    ew, eh = max(1, -(-pw // (1 << level))), max(1, -(-ph // (1 << level)))
    src = _unpack(op.mip("blur_pyramid", level))
    lum = src @ np.array([0.2126, 0.7152, 0.0722])
    padded = np.zeros((eh, ew)); padded[:src.shape[0], :src.shape[1]] = lum[:eh, :ew]
    log_lum = np.log2(np.maximum(1e-20, padded / pre_exposure))
    t = np.clip((log_lum + 16.0) / 32.0, 0, 1)
    bins = np.minimum((t * 256).astype(np.int64), 255)
    ys, xs = np.mgrid[0:eh, 0:ew]
    infl = np.exp(-8.0 * (((xs + 0.5) / ew - 0.5) ** 2 + ((ys + 0.5) / eh - 0.5) ** 2))
    ref = np.bincount(bins.ravel(), weights=np.floor(infl * 256).ravel(), minlength=256)
    got = op.histogram().astype(np.int64)
    # float32 vs float64 can move a texel across a bin edge or a weight across an integer: a handful of counts, never the shape
    assert abs(int(got.sum()) - int(ref.sum())) <= 0.002 * ref.sum()
    assert np.abs(np.cumsum(got) - np.cumsum(ref)).max() <= 0.004 * ref.sum()
    assert got[0] > 0                         # the black corner lands in bin 0 (log2(1e-20) clamps)
    # read-back: oracle, product library (host-only entry point) and the Python restatement of the Rust agree exactly
    rng = np.random.RandomState(5)
    cases = [(got, 0.0, 0.0), (got, 0.1, 0.1), (got, 0.6, 0.9), (got, 1.5, 0.2), (np.zeros(256, np.int64), 0.1, 0.1),
             (rng.randint(0, 100000, 256), 0.25, 0.5), (rng.randint(0, 3, 256), 0.05, 0.9)]
    for hist, lo, hi in cases:
        want = _read_back_histogram(hist, lo, hi)
        assert np.float32(op.read_back_histogram(hist, lo, hi)) == want
        assert np.float32(lib.luminance_histogram_mean_log2(hist, lo, hi)) == want
    one = np.zeros(256, np.uint32); one[100] = 1234
    assert lib.luminance_histogram_mean_log2(one) == -16.0 + (100.5 / 256.0) * 32.0 == -3.4375
    # clipping the dark half away raises the mean
    assert op.read_back_histogram(got, 0.5, 0.0) > op.read_back_histogram(got, 0.0, 0.0) > op.read_back_histogram(got, 0.0, 0.5)


def test_display_transform_properties(oracle):
    zero = oracle.OraclePost(post_tables.zero_bezold_brucke_lut())
    # achromatic in -> achromatic out, on the Siragusano-Smith curve (display_transform.hlsl:67-83); the sRGB <-> XYZ matrices are not
    # exact inverses of each other, hence 1e-3
    greys = np.logspace(-4, 3, 60, dtype=np.float32)
    out = zero.display_transform(np.repeat(greys[:, None], 3, 1))
    assert np.abs(out - out.mean(1, keepdims=True)).max() < 1.5e-3
    d = np.diff(out[:, 1])
    assert (d > -1e-6).all() and (d[out[:-1, 1] < 0.99] > 0).all() and out.max() <= 1.0 + 1e-3
    # below the shoulder a grey follows sy * (v / (v + sx))^p (its Helmholtz-Kohlrausch multiplier is 1), times the final "reach 100 %
    # white" rescale 1 / lerp(0.5, 1, max_comp_dist = 0)^(1/12) = 2^(1/12) (display_transform.hlsl:196-209)
    curve = 1.0205 * (greys / (greys + 1.0)) ** 1.2 * 2.0 ** (1.0 / 12.0)
    lo = greys < 0.5
    assert np.abs(out[lo, 1] / curve[lo] - 1.0).max() < 2e-3
    # saturated primaries stay inside the gamut, keep their dominant channel, and desaturate towards white as they get brighter
    for prim in np.eye(3, dtype=np.float32):
        ramp = zero.display_transform(prim[None] * np.logspace(-2, 4, 40, dtype=np.float32)[:, None])
        assert np.isfinite(ramp).all() and ramp.min() >= 0.0 and ramp.max() <= 2.0 ** (1.0 / 12.0) + 1e-3     # the per-channel roll-off's ceiling
        k = int(prim.argmax())
        sat = 1.0 - ramp.min(1) / ramp.max(1)
        assert (ramp.argmax(1) == k)[sat > 0.01].all()
        assert sat[0] > 0.9 and sat[-1] < 0.15 and (np.diff(sat) < 1e-3).all()
    # black: the 0/0 chromaticity flows through as NaN and is stored as 0 (chosen, see okj_post.hpp)
    assert (zero.display_transform(np.zeros((1, 3), np.float32)) == 0).all()
    # Bezold-Brucke LUT: greys have no hue to shift; colours move by an amount that grows with luminance (shift_amount = t / (t + 1))
    syn = oracle.OraclePost(post_tables.synthetic_bezold_brucke_lut(magnitude=0.08))
    assert np.abs(syn.display_transform(np.repeat(greys[:, None], 3, 1)) - out).max() < 2e-3
    col = np.array([[0.9, 0.3, 0.1]], np.float32)
    d_dim = np.abs(syn.display_transform(col * 0.01) - zero.display_transform(col * 0.01)).max() / zero.display_transform(col * 0.01).max()
    d_bright = np.abs(syn.display_transform(col * 4.0) - zero.display_transform(col * 4.0)).max() / zero.display_transform(col * 4.0).max()
    assert d_bright > 10 * d_dim and d_bright > 1e-3
    # LUT coordinate mapping (bezold_brucke.hlsl:18-49): offset -> coord -> offset is the identity on directions
    ang = np.linspace(0, 2 * np.pi, 97)[:-1] + 0.013
    offs = np.stack([np.cos(ang), np.sin(ang)], -1)
    o = offs / np.abs(offs).max(1, keepdims=True)
    coord = np.where(o.sum(1) > 0, 1.0, -1.0) * (0.125 * (o[:, 0] - o[:, 1]) + 0.25)
    coord = coord - np.floor(coord)           # REPEAT addressing
    side = np.where(coord < 0.5, 1.0, -1.0)
    t = (coord * 2) % 1.0
    back = np.stack([-1 + 2 * t + (1 - np.abs(t - 0.5) * 2), 1 - 2 * t + (1 - np.abs(t - 0.5) * 2)], -1) * side[:, None]
    back /= np.linalg.norm(back, axis=1, keepdims=True)
    assert np.abs(back - offs).max() < 1e-6


def test_post_combine_wiring(oracle):
    """What post_combine.hlsl:112-191 adds around the display transform: exposure multiplier, vignette, contrast, the blue-noise dither
    (+- 1/256, changes with frame_index), glare = 5 % of the reverse blur pyramid."""
    W, H = 64, 48
    lut = post_tables.zero_bezold_brucke_lut()
    op = oracle.OraclePost(lut)
    flat = np.full((H, W, 4), 0.25, np.float16)
    out0 = _unpack(op.render(_fc(W, H, 0), flat).copy())
    out1 = _unpack(op.render(_fc(W, H, 1), flat).copy())
    assert np.abs(out0 - out1).max() > 0 and np.abs(out0 - out1).max() < 2.5 / 256 + out0.max() / 32      # only the dither pattern moved
    centre, corner = out0[H // 2, W // 2, 1], out0[0, 0, 1]
    assert corner < centre                                          # vignette: exp(-2 r^3)
    brighter = _unpack(op.render(_fc(W, H, 0), flat, post_exposure_mult=2.0).copy())
    assert brighter[H // 2, W // 2, 1] > centre * 1.5
    punchy = _unpack(op.render(_fc(W, H, 0), flat, contrast=1.5).copy())
    assert punchy[H // 2, W // 2, 1] < centre                       # pow(col < 1, 1.5)
    # glare: one very bright texel lifts its dark neighbourhood through the pyramid
    spot = np.zeros((H, W, 4), np.float16); spot[..., :3] = 0.01; spot[H // 2, W // 2, :3] = 2000.0
    with_spot = _unpack(op.render(_fc(W, H, 0), spot).copy())
    base = _unpack(op.render(_fc(W, H, 0), np.where(spot > 1, 0.01, spot).astype(np.float16)).copy())
    assert with_spot[H // 2 + 6, W // 2 + 6, 1] > 2 * base[H // 2 + 6, W // 2 + 6, 1]


# ---------------------------------------------------------------- motion_blur (renderers/motion_blur.rs, rust-shaders/src/motion_blur.rs)
def _snorm(rm):
    return np.maximum(rm.astype(np.float64) / 32767.0, -1.0)


def _rust_motion_blur(fc, inp, depth, rm):
    """`motion_blur` (rust-shaders/src/motion_blur.rs:47-187) and the three velocity kernels (:189-264), restated line by line in Python
    floats (float64; the fp16 stores of the intermediate images are applied). Casts `as_uvec2()` saturate as Rust's `as` does."""
    H, W = inp.shape[:2]
    DH, DW = depth.shape
    vel = _snorm(rm)[..., :2]
    tw, th = -(-DW // 16), -(-DH // 16)
    f16 = lambda a: np.asarray(a, np.float64).astype(np.float16).astype(np.float64)

    def largest(vs):
        best, m = np.zeros(2), 0.0
        for v in vs:
            m2 = float(v @ v)
            if m2 > m:
                best, m = v, m2
        return best
    fetch = lambda img, x, y: img[y, x] if 0 <= x < img.shape[1] and 0 <= y < img.shape[0] else np.zeros(img.shape[2:])
    rx = np.array([[f16(largest([fetch(vel, x * 16 + i, y) for i in range(16)])) for x in range(tw)] for y in range(DH)])
    ry = np.array([[f16(largest([fetch(rx, x, y * 16 + i) for i in range(16)])) for x in range(tw)] for y in range(th)])
    dil = np.array([[f16(largest([fetch(ry, x + xx, y + yy) for xx in range(-2, 3) for yy in range(-2, 3)])) for x in range(tw)] for y in range(th)])
    c2v = fc.view_constants.clip_to_view[11]
    view_z = lambda d: np.float64(1.0) / (np.float64(d) * -c2v) if d != 0 else -np.inf
    sat = lambda v: min(max(v, 0.0), 1.0) if v == v else 0.0
    as_u = lambda v: int(v) if v > 0 else 0
    col = inp.astype(np.float64)

    def bilinear(img, u, v):
        h, w = img.shape[:2]
        fx, fy = u * w - 0.5, v * h - 0.5
        x0, y0 = math.floor(fx), math.floor(fy)
        tx, ty = fx - x0, fy - y0
        cl = lambda a, n: min(max(a, 0), n - 1)
        return ((img[cl(y0, h), cl(x0, w)] * (1 - tx) + img[cl(y0, h), cl(x0 + 1, w)] * tx) * (1 - ty)
                + (img[cl(y0 + 1, h), cl(x0, w)] * (1 - tx) + img[cl(y0 + 1, h), cl(x0 + 1, w)] * tx) * ty)

    def weight(cd, sd, offset_len, cs, ss):
        with np.errstate(invalid="ignore"):
            d = sd - cd
        dc = (sat(0.5 + 16.0 * d), sat(0.5 - 16.0 * d))
        sc = (sat(cs - (offset_len + 1.0)), sat(ss - (offset_len + 1.0)))
        return dc[0] * sc[0] + dc[1] * sc[1]
    out = np.zeros((H, W, 3))
    M = 0xffffffff
    for y in range(H):
        for x in range(W):
            uv = np.array([(x + 0.5) / W, (y + 0.5) / H])
            tx_, ty_ = x, y
            tx_ = (tx_ + (tx_ << 4)) & M; tx_ ^= tx_ >> 6
            ty_ = (ty_ + (tx_ << 1)) & M; ty_ = (ty_ + (ty_ << 6)) & M; ty_ ^= ty_ >> 2
            tx_ ^= ty_
            noise1 = ((tx_ ^ (ty_ << 1)) & 31) - 15
            off = np.array([(tx_ & 31) - 15, (ty_ & 31) - 15], np.float64)
            tc = uv * [DW, DH] + off
            tile = 0.5 * dil[min(as_u(tc[1]), DH - 1) // 16, min(as_u(tc[0]), DW - 1) // 16]
            noise = 0.5 * noise1 / 15.0
            center_uv = uv + tile * (noise / 4 * 0.5)
            cp = center_uv * [W, H]
            center_color = col[min(as_u(cp[1]), H - 1), min(as_u(cp[0]), W - 1), :3]
            nx, ny = min(max(math.floor(center_uv[0] * DW), 0), DW - 1), min(max(math.floor(center_uv[1] * DH), 0), DH - 1)
            center_depth = -view_z(depth[ny, nx])
            cvel_px = 0.5 * bilinear(vel, center_uv[0], center_uv[1]) * [DW, DH]
            s, sw, count = np.zeros(3), 0.0, 1.0
            if np.hypot(*tile) > 0:
                for i in range(1, 4):
                    ol0, ol1 = (i + noise) / 4 * 0.5, (-i + noise) / 4 * 0.5
                    uv0, uv1 = uv + tile * ol0, uv + tile * ol1
                    p0, p1 = uv0 * [DW, DH], uv1 * [DW, DH]
                    q0, q1 = (as_u(p0[0]), as_u(p0[1])), (as_u(p1[0]), as_u(p1[1]))
                    dz = lambda q: -view_z(depth[q[1], q[0]] if q[0] < DW and q[1] < DH else 0.0)
                    vz = lambda q: np.hypot(*(0.5 * (vel[q[1], q[0]] if q[0] < DW and q[1] < DH else np.zeros(2)) * [DW, DH]))
                    d0, d1, v0, v1 = dz(q0), dz(q1), vz(q0), vz(q1)
                    w0 = weight(center_depth, d0, np.hypot(*((uv0 - uv) * [DW, DH])), np.hypot(*cvel_px), v0)
                    w1 = weight(center_depth, d1, np.hypot(*((uv1 - uv) * [DW, DH])), np.hypot(*cvel_px), v1)
                    m0, m1 = d0 > d1, v1 > v0
                    w0 = w1 if (m0 and m1) else w0
                    w1 = w1 if (m0 or m1) else w0
                    val0 = 1.0 if (0 <= uv0[0] <= 1 and 0 <= uv0[1] <= 1) else 0.0
                    val1 = 1.0 if (0 <= uv1[0] <= 1 and 0 <= uv1[1] <= 1) else 0.0
                    w0 *= val0; w1 *= val1
                    count += val0 + val1
                    s += bilinear(col, uv0[0], uv0[1])[:3] * w0 + bilinear(col, uv1[0], uv1[1])[:3] * w1
                    sw += w0 + w1
                s, sw = s / count, sw / count
            out[y, x] = s + (1.0 - sw) * center_color
    return out, rx, ry, dil


def test_motion_blur_matches_the_rust_statement(oracle):
    from test_post_emulation import _motion_inputs
    W, H = 72, 48
    inp, depth, rm = _motion_inputs(W, H, W, H, 9)
    fc = _fc(W, H, 1)
    om = oracle.OracleMotionBlur()
    got = om.render(fc, inp, depth, rm).astype(np.float64)
    ref, rx, ry, dil = _rust_motion_blur(fc, inp, depth, rm)
    tw, th = -(-W // 16), -(-H // 16)
    assert np.array_equal(om.surface("velocity_reduced_x", np.float16, (H, tw, 2)).astype(np.float64), rx)      # selections: exact
    assert np.array_equal(om.surface("velocity_reduced_y", np.float16, (th, tw, 2)).astype(np.float64), ry)
    assert np.array_equal(om.surface("velocity_dilated", np.float16, (th, tw, 2)).astype(np.float64), dil)
    err = np.abs(got[..., :3] - ref)
    # float32 vs float64 can move a tap across a texel edge (as_uvec2) on a handful of pixels; everything else is fp16 rounding
    close = err <= np.abs(ref) * 2.0 ** -10 + 1e-4
    assert close.all(-1).mean() > 0.995, close.all(-1).mean()
    assert (got[..., 3] == 1.0).all()
    assert (np.abs(got[..., :3] - inp[..., :3].astype(np.float64)) > 1e-3).any(-1).mean() > 0.15                 # and it is not the identity


def test_motion_blur_properties(oracle):
    W, H = 80, 48
    fc = _fc(W, H, 0)
    om = oracle.OracleMotionBlur()
    rng = np.random.RandomState(4)
    inp = rng.uniform(0, 2, (H, W, 4)).astype(np.float16)
    depth = np.full((H, W), 0.002, np.float32)
    still = np.zeros((H, W, 4), np.int16)
    out = om.render(fc, inp, depth, still)
    assert np.array_equal(out[..., :3], inp[..., :3])                      # no motion anywhere: the centre tap with weight 1
    # uniform horizontal motion over a flat depth: columns mix, rows do not; an image constant along x is a fixed point
    move = still.copy(); move[..., 0] = int(0.08 * 32767)
    stripes = np.repeat(rng.uniform(0, 2, (H, 1, 4)), W, 1).astype(np.float16)
    out = om.render(fc, stripes, depth, move).astype(np.float32)
    assert np.abs(out[..., :3] - stripes[..., :3].astype(np.float32)).max() < 2e-3
    out = om.render(fc, inp, depth, move).astype(np.float32)
    assert out[..., :3].var() < 0.6 * inp[..., :3].astype(np.float32).var()
    # sky (depth 0 -> infinite distance): inf - inf = NaN depth differences clamp to weight 0 (the Rust's comment), pixels stay as they are
    sky = np.zeros((H, W), np.float32)
    out = om.render(fc, inp, sky, move)
    assert np.array_equal(out[..., :3], inp[..., :3])
