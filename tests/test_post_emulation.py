"""The product's post-processing kernels executed on the CPU. No GPU.

tests/post_emu.cpp includes kajiya_amd/csrc/post.hip — the kernel source and the host sequencing of kj_post_* exactly as shipped — with
tests/hip_emu/ first on the include path, where a small stand-in for the HIP language/runtime runs every workgroup with blockDim host
threads and a real barrier (see tests/hip_emu/hip/hip_runtime.h for what is and is not emulated). Built with the oracle's flags
(-ffp-contract=off, same libm), so the kernel source must reproduce the oracle BIT FOR BIT: indexing, LDS staging, mip bookkeeping, the
colour-science twin in kj_color.hpp, packing. What this cannot show is what hipcc makes of the same source (FMA contraction, ocml's
pow / exp): that is the -m gpu parity test's job (tests/test_zz_gpu_post.py). TEST INFRASTRUCTURE: nothing here is linked into or
called by libkajiya_amd.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from kajiya_amd import frame, post_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
SO = os.path.join(BUILD, "libpost_emu.so")
SOURCES = [os.path.join(ROOT, "tests", "post_emu.cpp"), os.path.join(ROOT, "tests", "hip_emu", "hip", "hip_runtime.h"), os.path.join(ROOT, "tests", "hip_emu", "hip", "hip_fp16.h"),
           os.path.join(ROOT, "kajiya_amd", "csrc", "post.hip"), os.path.join(ROOT, "kajiya_amd", "csrc", "kj_color.hpp"), os.path.join(ROOT, "kajiya_amd", "csrc", "kj_vec.hpp"),
           os.path.join(ROOT, "kajiya_amd", "csrc", "kj_host.hpp"), os.path.join(ROOT, "include", "kajiya_amd.h")]


@pytest.fixture(scope="module")
def emu():
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in SOURCES):
        subprocess.check_call(["g++", "-O1", "-std=c++20", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-Wl,-Bsymbolic",
                               "-I", os.path.join(ROOT, "tests", "hip_emu"), "-x", "c++", SOURCES[0], "-o", SO])
    L = C.CDLL(SO)
    vp, u32, f = C.c_void_p, C.c_uint32, C.c_float
    L.emu_device_create.restype = vp; L.emu_device_create.argtypes = [vp]
    L.emu_device_destroy.argtypes = [vp]
    L.emu_frame_begin.argtypes = [vp, vp]
    L.emu_last_error.restype = C.c_char_p
    L.emu_f32_to_f16.restype = C.c_uint16; L.emu_f32_to_f16.argtypes = [f]
    L.emu_f16_to_f32.restype = f; L.emu_f16_to_f32.argtypes = [C.c_uint16]
    L.kj_post_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.kj_post_destroy.argtypes = [vp]; L.kj_post_destroy.restype = None
    L.kj_post_render.argtypes = [vp, vp, u32, u32, u32, f, f, C.POINTER(vp), vp]
    L.kj_post_surface.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.kj_post_mip_levels.argtypes = [vp, C.POINTER(u32)]
    L.kj_post_read_back_histogram.argtypes = [vp, f, f, C.POINTER(f), vp]
    L.kj_motion_blur_create.argtypes = [vp, C.POINTER(vp)]
    L.kj_motion_blur_destroy.argtypes = [vp]; L.kj_motion_blur_destroy.restype = None
    L.kj_motion_blur_render.argtypes = [vp, vp, u32, u32, vp, vp, u32, u32, C.POINTER(vp), vp]
    L.kj_motion_blur_surface.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)]
    return L


class EmuPost:
    def __init__(self, L, blue_noise, lut):
        self.L, self._bn, self._lut = L, np.ascontiguousarray(blue_noise), np.ascontiguousarray(lut, np.float16)
        self.dev = L.emu_device_create(self._bn.ctypes.data)
        self.h = C.c_void_p()
        assert L.kj_post_create(self.dev, self._lut.ctypes.data, C.byref(self.h)) == 0, L.emu_last_error()

    def render(self, fc, inp, mult=1.0, contrast=1.0):
        is32 = np.asarray(inp).dtype == np.float32                    # KJ_POST_INPUT_RGBA32F: the path tracer's accumulation image
        inp = np.ascontiguousarray(inp, np.float32 if is32 else np.float16)
        H, W = inp.shape[:2]
        self.L.emu_frame_begin(self.dev, C.byref(fc))
        out = C.c_void_p()
        assert self.L.kj_post_render(self.h, inp.ctypes.data, 1 if is32 else 0, W, H, mult, contrast, C.byref(out), None) == 0, self.L.emu_last_error()
        return np.frombuffer((C.c_uint8 * (W * H * 4)).from_address(out.value), np.uint32).reshape(H, W).copy()

    def surface(self, name):
        p, n = C.c_void_p(), C.c_uint64()
        assert self.L.kj_post_surface(self.h, name.encode(), C.byref(p), C.byref(n)) == 0, self.L.emu_last_error()
        return np.frombuffer((C.c_uint8 * n.value).from_address(p.value), np.uint32).copy()

    def close(self):
        self.L.kj_post_destroy(self.h)
        self.L.emu_device_destroy(self.dev)


def _fc(W, H, frame_index, pre_exposure=1.0):
    fs = frame.FrameState((W, H))
    fs.frame_idx = frame_index
    fs.pre_exposure = pre_exposure
    return fs.prepare_frame_constants(frame.orbit_camera(0, (W, H)))


def test_emulated_fp16_matches_ieee(emu):
    """The stand-in for __float2half_rn / __half2float (tests/hip_emu/hip/hip_fp16.h) against numpy's IEEE conversions."""
    rng = np.random.RandomState(0)
    xs = np.concatenate([rng.uniform(-70000, 70000, 3000), rng.uniform(-1e-4, 1e-4, 3000), rng.uniform(-1e-7, 1e-7, 1000),
                         [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25, 2.0 ** -14, 2.0 ** -14 - 2.0 ** -25, np.inf, -np.inf]]).astype(np.float32)
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    for x, r in zip(xs, ref):
        assert emu.emu_f32_to_f16(float(x)) == int(r), (x, hex(int(r)))
    for h in range(0, 0x10000, 3):
        if (h & 0x7c00) == 0x7c00 and (h & 0x3ff):
            continue
        want = np.array([h], np.uint16).view(np.float16).astype(np.float32)[0]
        assert np.float32(emu.emu_f16_to_f32(h)) == want or (want == 0 and emu.emu_f16_to_f32(h) == 0)


@pytest.mark.parametrize("W,H,frame_index,mult,contrast,lut_seed", [(70, 50, 3, 1.3, 1.1, 0), (129, 33, 0, 1.0, 1.0, 1), (3, 2, 7, 0.5, 1.0, 2), (1, 1, 1, 1.0, 1.0, 3), (200, 140, 2, 2.0, 0.9, None)])
def test_product_kernels_on_cpu_equal_the_oracle(emu, oracle, W, H, frame_index, mult, contrast, lut_seed):
    lut = post_tables.zero_bezold_brucke_lut() if lut_seed is None else post_tables.synthetic_bezold_brucke_lut(lut_seed)
    rng = np.random.RandomState(W * 7 + H)
    inp = (rng.uniform(0, 1, (H, W, 4)) ** 3 * 6).astype(np.float16)
    inp[rng.uniform(size=(H, W)) < 0.03] = 0.0                       # black texels: the NaN path
    if W > 8:
        inp[H // 2, W // 3, :3] = 900.0                              # a highlight for the pyramid
    fc = _fc(W, H, frame_index, pre_exposure=0.7)
    op = oracle.OraclePost(lut)
    ref = op.render(fc, inp, mult, contrast).copy()
    ep = EmuPost(emu, oracle.blue_noise(), lut)
    try:
        got = ep.render(fc, inp, mult, contrast)
        n = C.c_uint32()
        assert emu.kj_post_mip_levels(ep.h, C.byref(n)) == 0 and n.value == op.mip_levels()
        for l in range(op.mip_levels()):
            for pyr in ("blur_pyramid", "rev_blur_pyramid"):
                assert np.array_equal(ep.surface(f"{pyr}:{l}"), op.mip(pyr, l).ravel()), (pyr, l)
        assert np.array_equal(ep.surface("histogram"), op.histogram())
        assert np.array_equal(got, ref)
        lum, hist = C.c_float(), np.zeros(256, np.uint32)
        assert emu.kj_post_read_back_histogram(ep.h, 0.1, 0.2, C.byref(lum), hist.ctypes.data) == 0
        assert np.array_equal(hist, op.histogram()) and np.float32(lum.value) == np.float32(op.read_back_histogram(hist, 0.1, 0.2))
    finally:
        ep.close()


def test_rgba32f_input_of_the_reference_mode(emu, oracle):
    """prepare_render_graph_reference (world_render_passes.rs:294-330) hands the RGBA32F accumulation image to post: same passes, wider
    input. Values beyond fp16's range exercise the difference."""
    W, H = 90, 60
    lut = post_tables.synthetic_bezold_brucke_lut(5)
    rng = np.random.RandomState(3)
    acc = (rng.uniform(0, 1, (H, W, 4)) ** 4 * 50).astype(np.float32)
    acc[3, 4, :3] = 3.0e5                                             # > 65504
    acc[..., 3] = 17.0                                                # the accumulator's sample count: ignored
    fc = _fc(W, H, 4)
    op = oracle.OraclePost(lut)
    ref = op.render(fc, acc, 0.25, 1.0).copy()
    ep = EmuPost(emu, oracle.blue_noise(), lut)
    try:
        assert np.array_equal(ep.render(fc, acc, 0.25, 1.0), ref)
        assert not np.array_equal(ref, op.render(fc, acc.astype(np.float16), 0.25, 1.0))      # fp16 would have clipped the highlight to inf
    finally:
        ep.close()


def test_extent_change_and_error_paths(emu, oracle):
    """A second extent on the same handle reallocates every surface (post.hip: `surf.clear()`); NULL arguments and a missing
    kj_frame_begin are reported through the status code + last-error string, as everywhere in the C-ABI."""
    lut = post_tables.synthetic_bezold_brucke_lut(4)
    ep = EmuPost(emu, oracle.blue_noise(), lut)
    try:
        for (W, H) in ((40, 30), (64, 64), (40, 30)):
            inp = (np.random.RandomState(W).uniform(0, 2, (H, W, 4))).astype(np.float16)
            fc = _fc(W, H, 1)
            op = oracle.OraclePost(lut)
            assert np.array_equal(ep.render(fc, inp), op.render(fc, inp))
        out = C.c_void_p()
        assert emu.kj_post_render(ep.h, None, 0, 8, 8, 1.0, 1.0, C.byref(out), None) != 0 and b"null argument" in emu.emu_last_error()
        assert emu.kj_post_render(ep.h, C.c_void_p(1), 0, 0, 8, 1.0, 1.0, C.byref(out), None) != 0
        assert emu.kj_post_render(ep.h, C.c_void_p(1), 7, 8, 8, 1.0, 1.0, C.byref(out), None) != 0 and b"input_format" in emu.emu_last_error()
        p, n = C.c_void_p(), C.c_uint64()
        assert emu.kj_post_surface(ep.h, b"no_such_surface", C.byref(p), C.byref(n)) != 0 and b"no post surface" in emu.emu_last_error()
        h = C.c_void_p()
        assert emu.kj_post_create(ep.dev, None, C.byref(h)) != 0
    finally:
        ep.close()
    bn = oracle.blue_noise()
    dev = emu.emu_device_create(bn.ctypes.data)
    h = C.c_void_p()
    assert emu.kj_post_create(dev, np.ascontiguousarray(lut).ctypes.data, C.byref(h)) == 0
    out = C.c_void_p()
    buf = np.zeros((4, 4, 4), np.float16)
    assert emu.kj_post_render(h, buf.ctypes.data, 0, 4, 4, 1.0, 1.0, C.byref(out), None) != 0 and b"kj_frame_begin" in emu.emu_last_error()
    emu.kj_post_destroy(h)
    emu.emu_device_destroy(dev)


def test_dynamic_exposure_loop_converges(emu, oracle):
    """The frame loop of world_renderer.rs:953-960 + world_render_passes.rs:281-289 on the emulated device: update_pre_exposure from the
    previous read-back, pre-exposed lighting in, post_mult in post. A constant scene of luminance 8 must read back log2 = 3 (within a
    histogram bin: 32 / 256 EV) whatever the pre-exposure is, and the total exposure must settle at 2^(-3 + DYNAMIC_EXPOSURE_BIAS)."""
    from kajiya_amd import exposure as E
    W, H = 64, 40
    ex = E.Exposure(dynamic_exposure=E.DynamicExposureState(enabled=True, speed_log2=6.0, histogram_clipping=E.HistogramClipping(0.1, 0.1)))
    fs = frame.FrameState((W, H))
    ep = EmuPost(emu, oracle.blue_noise(), post_tables.zero_bezold_brucke_lut())
    try:
        image_log2_lum, outs = 0.0, []
        for i in range(90):
            ex.update_pre_exposure(image_log2_lum)
            ex.apply(fs)
            fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H)))
            fs.retire_frame()
            scene = np.full((H, W, 4), 8.0 * float(ex.state.pre_mult), np.float16)          # lighting is computed pre-exposed
            out = ep.render(fc, scene, float(ex.state.post_mult), ex.contrast)
            lum, hist = C.c_float(), np.zeros(256, np.uint32)
            c = ex.dynamic_exposure.histogram_clipping
            assert emu.kj_post_read_back_histogram(ep.h, c.low, c.high, C.byref(lum), hist.ctypes.data) == 0
            image_log2_lum = lum.value
            assert abs(image_log2_lum - 3.0) <= 0.13, (i, image_log2_lum)
            assert float(fc.pre_exposure) == float(ex.state.pre_mult) and abs(fc.pre_exposure_delta - ex.state.pre_mult / ex.state.pre_mult_prev) < 1e-6
            outs.append(out[H // 2, W // 2])
        total = float(ex.state.pre_mult) * float(ex.state.post_mult)
        assert abs(np.log2(total) - (-3.0 - 2.0)) < 0.15, np.log2(total)
        assert abs(np.log2(float(ex.state.pre_mult)) - np.log2(total)) < 0.05            # the 10 %-per-frame pre-exposure blend has caught up
        a, b = (float(np.array([int(o) & 0x7ff], np.uint16)[0] << 4) for o in outs[-2:])      # red channel's 5e6m bits as an integer
        assert abs(a - b) <= 2 * 16                                                      # the displayed value has settled (dither aside)
    finally:
        ep.close()


def _motion_inputs(W, H, DW, DH, seed):
    """A moving foreground band over a slower background, a sky region (depth 0) and a fast strip at the border: every branch of
    motion_blur.rs (mirrored weights, invalid taps outside [0, 1], NaN depth differences) gets exercised."""
    rng = np.random.RandomState(seed)
    inp = (rng.uniform(0, 1, (H, W, 4)) ** 2 * 3).astype(np.float16)
    depth = np.full((DH, DW), 0.0012, np.float32) * rng.uniform(0.9, 1.1, (DH, DW)).astype(np.float32)
    depth[DH // 3: DH // 2] = 0.004
    depth[: DH // 8] = 0.0
    vel = np.zeros((DH, DW, 2))
    vel[..., 0] = 0.01 + 0.004 * np.sin(np.arange(DW) * 0.1)[None, :]
    vel[DH // 3: DH // 2, :, 0] = -0.06
    vel[DH // 3: DH // 2, :, 1] = 0.03
    vel[:, : DW // 10, 0] = 0.2
    vel[-DH // 6:, :, :] = 0.0
    rm = np.zeros((DH, DW, 4), np.int16)
    rm[..., :2] = np.round(np.clip(vel, -1, 1) * 32767)
    rm[..., 2:] = rng.randint(-32767, 32767, (DH, DW, 2))
    return inp, depth, rm


@pytest.mark.parametrize("W,H,DW,DH", [(96, 64, 96, 64), (131, 77, 131, 77), (128, 96, 64, 48), (5, 3, 5, 3)])
def test_motion_blur_kernels_on_cpu_equal_the_oracle(emu, oracle, W, H, DW, DH):
    inp, depth, rm = _motion_inputs(W, H, DW, DH, W + DH)
    fc = _fc(DW, DH, 2)
    om = oracle.OracleMotionBlur()
    ref = om.render(fc, inp, depth, rm).copy()
    bn = oracle.blue_noise()
    dev = emu.emu_device_create(bn.ctypes.data)
    emu.emu_frame_begin(dev, C.byref(fc))
    h = C.c_void_p()
    assert emu.kj_motion_blur_create(dev, C.byref(h)) == 0
    try:
        out = C.c_void_p()
        assert emu.kj_motion_blur_render(h, inp.ctypes.data, W, H, depth.ctypes.data, rm.ctypes.data, DW, DH, C.byref(out), None) == 0, emu.emu_last_error()
        got = np.frombuffer((C.c_uint8 * (W * H * 8)).from_address(out.value), np.uint16).reshape(H, W, 4)
        assert np.array_equal(got, ref.view(np.uint16))
        tw, th = (DW + 15) // 16, (DH + 15) // 16
        for name, shape in (("velocity_reduced_x", (DH, tw, 2)), ("velocity_reduced_y", (th, tw, 2)), ("velocity_dilated", (th, tw, 2))):
            p, n = C.c_void_p(), C.c_uint64()
            assert emu.kj_motion_blur_surface(h, name.encode(), C.byref(p), C.byref(n)) == 0
            assert np.array_equal(np.frombuffer((C.c_uint8 * n.value).from_address(p.value), np.uint16).reshape(shape), om.surface(name, np.uint16, shape)), name
        if W > 16:
            assert (got[..., :3] != inp.view(np.uint16)[..., :3]).any(-1).mean() > 0.2            # it does blur
        assert emu.kj_motion_blur_render(h, None, W, H, depth.ctypes.data, rm.ctypes.data, DW, DH, C.byref(out), None) != 0
    finally:
        emu.kj_motion_blur_destroy(h)
        emu.emu_device_destroy(dev)


def test_kernels_are_clean_under_address_and_ub_sanitizers():
    """tests/post_emu_sanitize.cpp: every kernel of post.hip over exact-size heap buffers (1x1 ... 31x257, an upscaled motion-blur case,
    both input formats, black texels for the NaN path) with ASan + UBSan + float-cast-overflow. On the CPU stand-in device memory is heap
    memory, so an index a GPU would fault on (or quietly read garbage from) aborts here."""
    exe = os.path.join(BUILD, "post_emu_sanitize")
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "post_emu_sanitize.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in SOURCES + [src]):
        subprocess.check_call(["g++", "-g", "-O1", "-std=c++20", "-pthread", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                               "-I", os.path.join(ROOT, "tests", "hip_emu"), "-x", "c++", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok ") == 8
