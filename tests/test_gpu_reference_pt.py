"""Reference path tracer: HIP kernel vs the CPU restatement, the pixel-interleaved split, and the
convergence of the ReSTIR GI output towards it.

The path tracer is chaotic in the usual sense: a 1-ulp difference in sinf/logf/powf between the host libm and the device
can flip a lobe choice, a Russian-roulette decision or a hit near an edge, after which the two paths are unrelated. Parity is
therefore stated per pixel: all but a small fraction of the one-sample images agree to 1e-3, and the accumulated images
agree in the mean."""
import numpy as np
import pytest

import test_gpu_parity as T

pytestmark = pytest.mark.gpu


def _fcs(W, H, n, scene, static=False, lights=0):
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    fs.triangle_light_count = lights
    out = []
    for i in range(n):
        j = 0 if static else i
        if scene == "cornell":
            cam = frame.orbit_camera(j, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01)
        else:
            cam = frame.orbit_camera(j, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
        out.append(fs.prepare_frame_constants(cam))
        fs.retire_frame()
    return out


@pytest.mark.parametrize("name,use_lights", [("cornell", False), ("city20k", True)])
def test_reference_pt_matches_oracle(gpu, oracle, device, name, use_lights):
    import torch
    W, H, N = 96, 64, 24
    desc = T._scenes()[name]
    osc = oracle.OracleScene(desc, use_lights=use_lights)
    gsc = gpu.Scene(device, desc, use_lights=use_lights)
    nl = gsc.triangle_light_count
    assert nl == osc.triangle_light_count and (nl > 0) == (use_lights and name != "cornell")
    gp = gpu.GpuPipeline(device, gsc, W, H)
    fcs = _fcs(W, H, N, name, lights=nl)
    acc_g = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    acc_o = np.zeros((H, W, 4), np.float32)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    rays_o = 0
    worst_frac = 0.0
    for fi, fc in enumerate(fcs):
        one_g = torch.zeros_like(acc_g)
        one_o = np.zeros_like(acc_o)
        device.frame_begin(fc)
        gp.reference_path_trace(one_g)
        gp.reference_path_trace(acc_g, ray_counter=counter)
        oracle.reference_path_trace(osc, fc, one_o)
        rays_o += oracle.reference_path_trace(osc, fc, acc_o)
        g = one_g.cpu().numpy()
        assert np.isfinite(g).all() and (g[..., 3] == 1.0).all() and (g[..., :3] >= 0).all()
        err = np.abs(g[..., :3] - one_o[..., :3]).max(axis=-1) / (1e-3 + np.abs(one_o[..., :3]).max(axis=-1))
        frac = float((err > 1e-3).mean())
        worst_frac = max(worst_frac, frac)
        assert frac < 0.02, f"frame {fi}: {frac:.4f} of the one-sample pixels differ by more than 1e-3"
    g = acc_g.cpu().numpy()
    assert (g[..., 3] == N).all() and (acc_o[..., 3] == N).all()
    rays_g = int(counter.item())
    assert abs(rays_g - rays_o) / rays_o < 5e-3, (rays_g, rays_o)
    mean_g, mean_o = g[..., :3].mean(axis=(0, 1)), acc_o[..., :3].mean(axis=(0, 1))
    assert np.allclose(mean_g, mean_o, rtol=0.02), (mean_g, mean_o)
    med = np.median(np.abs(g[..., :3] - acc_o[..., :3]) / (1e-3 + acc_o[..., :3]))
    assert med < 1e-4, med
    print(f"[{name}] one-sample mismatch frac worst {worst_frac:.5f}; rays gpu {rays_g} oracle {rays_o}; mean gpu {mean_g} oracle {mean_o}")


def test_reference_pt_interleaved_split_is_exact(gpu, device):
    """BASELINE config 5: tiles dealt round-robin to N ranks; the sum of the per-rank images == the unsplit image."""
    import torch
    W, H = 104, 72            # not a multiple of 8 in y -> partial tiles
    desc = T._scenes()["city20k"]
    gsc = gpu.Scene(device, desc)
    gp = gpu.GpuPipeline(device, gsc, W, H)
    fcs = _fcs(W, H, 3, "city")
    full = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    parts = [torch.zeros_like(full) for _ in range(3)]
    for fc in fcs:
        device.frame_begin(fc)
        gp.reference_path_trace(full)
        for r in range(3):
            gp.reference_path_trace(parts[r], interleave=(3, r))
    torch.cuda.synchronize()
    owned = torch.stack([(p[..., 3] > 0) for p in parts]).sum(dim=0)
    assert bool((owned == 1).all())
    total = parts[0] + parts[1] + parts[2]
    assert torch.equal(total, full)


def test_restir_gi_converges_to_reference_pt(gpu, device):
    """time-averaged ReSTIR GI output (static camera, Cornell box 512x512 = BASELINE configs[0] extent,
    irradiance cache on) against the path tracer's indirect light through a white Lambert first bounce
    (first_bounce_mode 2 == what `gi_irradiance` multiplies in light_gbuffer.hlsl:158-170).

    ReSTIR GI is a biased, self-feeding estimator: bounce light at a hit comes from last frame's denoised output when the
    hit passes a 0.5 % screen-depth gate (diffuse_trace_common.inc.hlsl:85-107) and from the irradiance cache otherwise, so
    any per-bounce loss compounds. Measured on MI355X (scripts/convergence_probe.py, profiles/r01_convergence_*.png):
    mean ratio 0.88, relative L2 0.19-0.23 (mostly the path tracer's own 512-spp noise on sun-lit caustic paths plus
    darkening towards the open front of the box); the ReSTIR + denoiser chain itself preserves the candidates' mean
    within 2-3 %. With the oracle pinned to the reference's shader text (round 4) the 11-12 % deficit is the REFERENCE's own: its path
    tracer follows up to 17 segments where the GI estimator's self-feeding loop loses a few percent per bounce to the depth gate, the cache's
    3-4-vertex paths and the denoiser's clamps (scripts/pt_deficit_attribution.py) -- not an error of this implementation, so the window
    is held around the measured value instead of reaching up to 1. Stated tolerance: box-averaged relative L2 < 0.25, image mean within
    [0.84, 0.94] of the path tracer (measured 0.88 - 0.89 in rounds 1-5)."""
    import torch
    from kajiya_amd import frame
    W = H = 512
    desc = T._scenes()["cornell"]
    gsc = gpu.Scene(device, desc)
    gp = gpu.GpuPipeline(device, gsc, W, H, use_ircache=True)
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = True
    n_warm, n_avg, n_pt = 64, 128, 512
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    gi_sum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for i in range(n_pt):
        fc = fs.prepare_frame_constants(frame.orbit_camera(0, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.01))
        fs.retire_frame()
        device.frame_begin(fc)
        gp.reference_path_trace(acc, first_bounce_mode=2)
        if i < n_warm + n_avg:
            gp.frame(fc)
            if i >= n_warm:
                gi_sum += gp.surface("spatial_filtered_tex", torch.float16, (H, W, 4))[..., :3].float()
    torch.cuda.synchronize()
    gi = (gi_sum / n_avg).cpu().numpy()
    pt = acc[..., :3].cpu().numpy()
    m = gp.depth.cpu().numpy() > 0
    assert m.mean() > 0.5

    def box(a):
        return np.where(m[..., None], a, 0.0).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
    rel_l2 = float(np.sqrt(((gi - pt)[m] ** 2).sum() / (pt[m] ** 2).sum()))
    rel_l2_box = float(np.sqrt(((box(gi) - box(pt)) ** 2).sum() / (box(pt) ** 2).sum()))
    mean_ratio = float(gi[m].mean() / pt[m].mean())
    print(f"rtdgi vs reference PT (Cornell {W}x{H}, {n_avg} frames vs {n_pt} spp): rel L2 {rel_l2:.4f} (8x8 box-averaged {rel_l2_box:.4f}), mean ratio {mean_ratio:.4f}")
    assert rel_l2_box < 0.25 and 0.84 < mean_ratio < 0.94, (rel_l2, rel_l2_box, mean_ratio)
