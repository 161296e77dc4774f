"""SsgiRenderer: HIP kernels vs the oracle, frame by frame on identical inputs and history; and the rtdgi frame
driven by the real guide instead of the constant 1.0."""
import ctypes as C
import numpy as np
import pytest

import parity as P
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_name,W,H", [("cornell", 192, 160), ("city20k", 251, 141)])
def test_ssgi_per_frame_parity(gpu, oracle, device, scene_name, W, H):
    import torch
    desc = T._scenes()[scene_name]
    op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
    hw, hh = (W + 1) // 2, (H + 1) // 2
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    worst = {}
    for fi, fc in enumerate(T._frame_constants(W, H, 7, "cornell" if scene_name == "cornell" else "city")):
        op.render_inputs(fc); op.reprojection(fc)
        gp.dev.frame_begin(fc)
        T._sync_inputs(op, gp, torch)
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        if fi > 0:   # identical history on both sides
            for n in ("ssgi:0", "ssgi:1"):
                gp.ssgi_surface(n, torch.uint8, (-1,)).copy_(torch.from_numpy(op.ssgi_surface(n, np.uint8, (-1,)).copy()))
        ref_ao = op.ssgi_frame(fc).copy()
        gp.ssgi_frame()
        torch.cuda.synchronize()
        for name, fmt, shape in (("ssgi_tex", "r16f", (hh, hw)), ("spatially_filtered_tex", "r16f", (hh, hw)), ("upsampled_tex", "r16f", (H, W)),
                                 (f"ssgi:{fi % 2}", "r16f", (H, W))):
            a = gp.ssgi_surface(name, torch.float16, shape).float().cpu().numpy()
            b = op.ssgi_surface(name, np.float16, shape).astype(np.float32)
            rel = float(np.sqrt(((a - b) ** 2).sum() / max(1e-20, (b ** 2).sum())))
            mism = float((np.abs(a - b) > 1e-3 * (1 + np.abs(b))).mean())
            key = name.split(":")[0]
            worst[key] = max(worst.get(key, (0, 0)), (rel, mism))
            assert rel < 1e-3 or mism < 2e-3, (fi, name, rel, mism)
        got_ao = gpu.tensor_from_ptr(gp.ssao_ptr.value, W * H, torch.uint8, (H, W)).cpu().numpy()
        d = np.abs(got_ao.astype(np.int32) - ref_ao.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 5e-3, (fi, d.max(), (d > 0).mean())    # R8 rounding flips only
        m = op.depth > 0
        assert 0.3 < ref_ao[m].mean() / 255.0 < 1.0
    print({k: (f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in worst.items()})


def test_rtdgi_with_ssgi_guide_matches_oracle(gpu, oracle, device):
    """Free-running frames with the SSAO guide bound on both sides (world_render_passes.rs order: reprojection, ssgi, rtdgi)."""
    import torch
    W, H = 256, 256
    op, gp = T._make_pipelines(gpu, oracle, device, T._scenes()["cornell"], W, H)
    for fc in T._frame_constants(W, H, 10):
        op.render_inputs(fc); op.reprojection(fc); op.ssgi_frame(fc); op.rtdgi_frame(fc)
        gp.render_inputs(fc); gp.reprojection(); gp.ssgi_frame(); gp.rtdgi_frame()
    torch.cuda.synchronize()
    ref = op.surface("spatial_filtered_tex", np.uint8, (-1,))
    got = gp.surface("spatial_filtered_tex", torch.uint8, (-1,)).cpu().numpy()
    r = P.compare(got, ref, "rgba16f")
    print("free-running 10 frames with the SSGI guide:", r)
    assert r["rel_l2"] < 3e-2, r
    # and the guide does change the result
    op2, _ = T._make_pipelines(gpu, oracle, device, T._scenes()["cornell"], W, H)
    for fc in T._frame_constants(W, H, 10):
        op2.frame(fc)
    ref_const = op2.surface("spatial_filtered_tex", np.uint8, (-1,))
    assert P.compare(ref_const, ref, "rgba16f")["rel_l2"] > 5e-3
