"""ShadowDenoiseRenderer (shadow_denoise.rs + the FidelityFX shadow denoiser it wraps): HIP kernels against the oracle,
frame by frame on identical shadow masks, G-buffer inputs and temporal state."""
import ctypes as C
import numpy as np
import pytest

import test_gpu_parity as T

pytestmark = pytest.mark.gpu

SURF = {"bitpacked_shadows_image": "u32", "metadata_image": "u32", "spatial_input_image": "rg16f", "temp": "rg16f",
        "shadow_denoise_accum:0": "rg16f", "shadow_denoise_accum:1": "rg16f", "shadow_denoise_moments:0": "rgba16f", "shadow_denoise_moments:1": "rgba16f"}


@pytest.mark.parametrize("scene_name,W,H", [("city20k", 256, 160), ("cornell", 171, 99)])
def test_shadow_denoise_per_frame_parity(gpu, oracle, device, scene_name, W, H):
    import torch
    from kajiya_amd import frame
    desc = T._scenes()[scene_name]
    op, gp = T._make_pipelines(gpu, oracle, device, desc, W, H)
    fs = frame.FrameState((W, H), sun_size_multiplier=6.0)          # wide sun: real penumbrae
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    worst = {}
    for fi in range(7):
        cam = frame.orbit_camera(fi, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01) if scene_name == "cornell" else \
            frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
        fc = fs.prepare_frame_constants(cam); fs.retire_frame()
        op.render_inputs(fc); op.reprojection(fc)
        gp.dev.frame_begin(fc)
        T._sync_inputs(op, gp, torch)
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        mask = op.sun_shadow_mask(fc)
        d_mask = torch.from_numpy(mask).cuda()
        if fi > 0:   # identical temporal state
            for n in ("shadow_denoise_accum:0", "shadow_denoise_accum:1", "shadow_denoise_moments:0", "shadow_denoise_moments:1"):
                gp.shadow_denoise_surface(n, torch.uint8, (-1,)).copy_(torch.from_numpy(op.shadow_denoise_surface(n, np.uint8, (-1,)).copy()))
        ref = op.shadow_denoise(fc, mask)
        got = gp.shadow_denoise(d_mask)
        torch.cuda.synchronize()
        for name, fmt in SURF.items():
            a = gp.shadow_denoise_surface(name, torch.uint8, (-1,)).cpu().numpy()
            b = op.shadow_denoise_surface(name, np.uint8, (-1,))
            assert a.size == b.size, name
            if fmt == "u32":
                assert np.array_equal(a, b), (fi, name, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
                continue
            fa, fb = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
            assert np.isfinite(fa).all() and np.isfinite(fb).all(), (fi, name)
            rel = float(np.sqrt(((fa - fb) ** 2).sum() / max(1e-20, (fb ** 2).sum())))
            mism = float((np.abs(fa - fb) > 1e-3 * (1 + np.abs(fb))).mean())
            worst[name.split(":")[0]] = max(worst.get(name.split(":")[0], 0.0), rel)
            assert rel < 1e-3 or mism < 2e-3, (fi, name, rel, mism)
        g = got[..., 0].float().cpu().numpy()
        m = op.depth > 0
        assert np.abs(g - ref)[m].max() < 2e-2 and np.abs(g - ref)[m].mean() < 1e-4
    # the denoiser did something: smooth values strictly between 0 and 1 where the raw mask is binary
    assert ((ref > 0.05) & (ref < 0.95))[m].mean() > 0.01
    print({k: f"{v:.2e}" for k, v in worst.items()})
