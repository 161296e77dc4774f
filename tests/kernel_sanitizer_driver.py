"""TEST INFRASTRUCTURE, run by tests/test_kernel_sanitizers.py in a subprocess with libasan / libubsan preloaded:
    kernel_sanitizer_driver.py <ssgi|taa|shadow_denoise> <path to the instrumented lib<name>_emu_san.so>
Drives the product's real kj_* entry points — compiled from kajiya_amd/csrc/<name>.hip against the CPU stand-in for HIP — with the oracle
pipeline's inputs over a few frames at awkward extents and prints one line per frame: `<name> W H frame mismatching_bytes total_bytes`."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import okj_py as o
from kajiya_amd import frame, scenes, abi

which, so = sys.argv[1], sys.argv[2]
L = C.CDLL(so)
vp, u32 = C.c_void_p, C.c_uint32
L.emu_device_create.restype = vp; L.emu_device_create.argtypes = [vp]
L.emu_device_destroy.argtypes = [vp]
L.emu_frame_begin.argtypes = [vp, vp]
L.emu_last_error.restype = C.c_char_p


def gbuffer_depth(op):
    g = abi.KjGbufferDepth()
    g.geometric_normal, g.gbuffer, g.depth = op.geometric_normal.ctypes.data, op.gbuffer.ctypes.data, op.depth.ctypes.data
    g.width, g.height = op.W, op.H
    return g


def view(ptr, nbytes):
    return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), np.uint8)


for (W, H) in ((97, 61), (64, 48), (33, 17)):
    op = o.OraclePipeline(o.OracleScene(scenes.cornell_box()), W, H)
    bn = o.blue_noise()
    dev = L.emu_device_create(bn.ctypes.data)
    h = vp()
    if which == "ssgi":
        L.kj_ssgi_create.argtypes = [vp, C.POINTER(vp)]
        L.kj_ssgi_render.argtypes = [vp, C.POINTER(abi.KjGbufferDepth), vp, vp, C.POINTER(vp), vp]
        assert L.kj_ssgi_create(dev, C.byref(h)) == 0
    elif which == "taa":
        L.kj_taa_create.argtypes = [vp, C.POINTER(vp)]
        L.kj_taa_render.argtypes = [vp, vp, u32, u32, vp, vp, u32, u32, C.POINTER(abi.KjTaaOutput), vp]
        assert L.kj_taa_create(dev, C.byref(h)) == 0
    else:
        L.kj_shadow_denoise_create.argtypes = [vp, C.POINTER(vp)]
        L.kj_shadow_denoise_render.argtypes = [vp, C.POINTER(abi.KjGbufferDepth), vp, vp, C.POINTER(vp), vp]
        assert L.kj_shadow_denoise_create(dev, C.byref(h)) == 0
    fs = frame.FrameState((W, H))
    rng = np.random.RandomState(W)
    for i in range(4):
        fc = fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0, 1, 0), radius=6.5, height=0.0, rate=0.02)); fs.retire_frame()
        op.render_inputs(fc); op.reprojection(fc)
        L.emu_frame_begin(dev, C.byref(fc))
        g = gbuffer_depth(op)
        if which == "ssgi":
            ref = op.ssgi_frame(fc).copy().view(np.uint8).ravel()
            out = vp()
            assert L.kj_ssgi_render(h, C.byref(g), op.reprojection_map.ctypes.data, None, C.byref(out), None) == 0, L.emu_last_error()
            got = view(out.value, W * H)
        elif which == "taa":
            lit = (rng.uniform(0, 1, (H, W, 4)) ** 2 * 4).astype(np.float16)
            this_frame, temporal = op.taa_frame(fc, input_ptr=lit.ctypes.data)         # (this_frame_out, temporal_out) pointers
            ref = np.concatenate([view(this_frame, W * H * 8), view(temporal, W * H * 8)])
            out = abi.KjTaaOutput()
            assert L.kj_taa_render(h, lit.ctypes.data, W, H, op.reprojection_map.ctypes.data, op.depth.ctypes.data, W, H, C.byref(out), None) == 0, L.emu_last_error()
            got = np.concatenate([view(out.this_frame_out, W * H * 8), view(out.temporal_out, W * H * 8)])
        else:
            mask = np.ascontiguousarray(op.sun_shadow_mask(fc), np.uint8)
            ref = op.shadow_denoise(fc, mask).astype(np.float32).view(np.uint8).ravel()      # x of the RG16F image: the denoised shadow term
            out = vp()
            assert L.kj_shadow_denoise_render(h, C.byref(g), mask.ctypes.data, op.reprojection_map.ctypes.data, C.byref(out), None) == 0, L.emu_last_error()
            got = np.ascontiguousarray(view(out.value, W * H * 4).view(np.float16).reshape(H, W, 2)[..., 0].astype(np.float32)).view(np.uint8).ravel()
        print(which, W, H, i, int((got != ref).sum()), got.size, flush=True)
    L.emu_device_destroy(dev)
print("done")
