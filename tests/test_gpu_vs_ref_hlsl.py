"""The DEVICE against the reference's own shader text, directly: the two halves of the parity chain -- kernels vs oracle at the sizes BASELINE
names, oracle vs the compiled HLSL at a few thousand texels -- met only through the oracle and at different sizes. Here one 1080p frame's state goes to BOTH the MI355X
kernels and oracle/_ref/libref_hlsl.so (kajiya's HLSL compiled for the CPU by oracle/ref_hlsl; the prebuilt library travels to the GPU box with the snapshot, the reference
checkout does not have to), pass by pass, and the kernels' surfaces are held to the text's under tests/parity.py's bars. The oracle only supplies the state the passes start from.

Passes: the ray-free ones of rtdgi (reproject, half-res extracts, validity integrate, restir temporal, restir spatial x2, resolve, temporal filter, spatial filter): 17 surfaces.
(The ray passes need the reference's hit shaders bound to a scene on the text's side: tests/test_ref_hlsl.py does that at small extents. TAA's probability stage amplifies
one fp16 ulp to O(1), so its passes are compared as whole frames, not from shared intermediates: tests/test_gpu_taa.py.) A minute or two of CPU for the text at 1080p; one
tracing frame after five warm-up frames. KJ_TEST_VS_TEXT_EXTENT=WxH runs another extent (the CPU stand-in of the GPU suite uses a small one)."""
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
