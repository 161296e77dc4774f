"""Image material maps (albedo / spec / emissive maps sampled with ray-cone LOD, rt/gbuffer.rchit.hlsl:29-44,96-173):
HIP hit shading against the oracle on a scene with mip-mapped RGBA8 maps, uv transforms, sRGB and linear texels."""
import ctypes as C
import numpy as np
import pytest

import parity as P
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


def _unpack888(v):
    return np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], -1).astype(np.int32)


@pytest.mark.parametrize("W,H", [(320, 200), (80, 48)])
def test_textured_gbuffer_matches_oracle(gpu, oracle, device, W, H):
    """Primary-ray G-buffer (the raster stand-in shades hits with the same code as every GI ray). Two extents = two LOD ranges.
    Packed albedo is 8:8:8 sqrt-encoded: allow 1 LSB (bilinear weights go through different FMA contraction) on a few texels."""
    import torch
    op, gp = T._make_pipelines(gpu, oracle, device, T._scenes()["textured"], W, H)
    for fc in T._frame_constants(W, H, 3, "textured"):
        op.render_inputs(fc)
        gp.render_inputs(fc)
        torch.cuda.synchronize()
        d_ref, d = op.depth, gp.depth.cpu().numpy()
        both = (d_ref != 0) & (d != 0)
        assert ((d_ref == 0) == (d == 0)).mean() > 0.999
        ref, got = op.gbuffer[both], gp.gbuffer.cpu().numpy().view(np.uint32)[both]
        da = np.abs(_unpack888(ref[:, 0]) - _unpack888(got[:, 0])).max(axis=-1)
        assert (da > 1).mean() < 2e-3 and (da > 0).mean() < 0.05, ((da > 1).mean(), (da > 0).mean())
        assert (ref[:, 1] == got[:, 1]).mean() > 0.995                      # normal
        rm_ref, rm_got = ref[:, 2].copy().view(np.float16).reshape(-1, 2).astype(np.float32), got[:, 2].copy().view(np.float16).reshape(-1, 2).astype(np.float32)
        assert np.abs(rm_ref - rm_got).max(axis=-1).mean() < 1e-3 and (np.abs(rm_ref - rm_got).max(axis=-1) > 2e-2).mean() < 2e-3
        assert (ref[:, 3] == got[:, 3]).mean() > 0.98                       # rgb9e5 emissive
        # the maps really are sampled: many distinct albedo values, some emissive texels
        assert len(np.unique(ref[:, 0])) > 200 and (ref[:, 3] != 0).any()


def test_textured_rtdgi_per_pass_parity(gpu, oracle, device):
    """Every rtdgi pass on the textured scene: hit shading of the GI rays goes through sample_map with the reflected ray cone."""
    T._per_pass_parity(gpu, oracle, device, "textured", 256, 160, 2, False)


def test_textured_reference_pt(gpu, oracle, device):
    import torch
    from kajiya_amd import frame
    W, H = 96, 64
    desc = T._scenes()["textured"]
    osc, gsc = oracle.OracleScene(desc), gpu.Scene(device, desc)
    gp = gpu.GpuPipeline(device, gsc, W, H)
    for fc in T._frame_constants(W, H, 6, "textured"):
        one_g = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        one_o = np.zeros((H, W, 4), np.float32)
        device.frame_begin(fc)
        gp.reference_path_trace(one_g)
        oracle.reference_path_trace(osc, fc, one_o)
        g = one_g.cpu().numpy()
        err = np.abs(g[..., :3] - one_o[..., :3]).max(axis=-1) / (1e-3 + np.abs(one_o[..., :3]).max(axis=-1))
        assert float((err > 2e-3).mean()) < 0.03, float((err > 2e-3).mean())


def test_image_map_validation(gpu, device):
    from kajiya_amd import scenes
    import numpy as np
    img = np.zeros((8, 8, 4), np.uint8)
    m = scenes.TriangleMesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.tile(np.array([[0, 0, 1]], np.float32), (3, 1)), np.array([0, 1, 2], np.uint32),
                            materials=[dict(base_color=(1, 1, 1, 1), roughness=1.0, metalness=0.0, emissive=(0, 0, 0), albedo_image=img)],
                            uvs=np.zeros((3, 2), np.float32))
    d, keep = m.pack()
    maps = C.cast(d.maps, C.POINTER(gpu.KjMaterialMap if hasattr(gpu, "KjMaterialMap") else __import__("kajiya_amd.abi", fromlist=["KjMaterialMap"]).KjMaterialMap))
    maps[2].mip_count = 9          # an 8x8 image has 4 levels
    sc = gpu.Scene(device)
    out = C.c_uint32()
    assert gpu.load().kj_scene_add_mesh(sc.h, C.byref(d), C.byref(out)) != 0
    assert b"mip_count" in gpu.load().kj_last_error()


def test_baked_scene_matches_direct_scene_on_gpu(gpu, device, tmp_path):
    """Baked-asset reader (tests/test_baked_assets.py holds the format tests): the textured scene written as
    `.mesh` / `.image` files, loaded with load_baked_mesh and added through kj_scene_add_mesh, gives a bit-identical G-buffer."""
    import torch
    import baked_writer as BW
    from kajiya_amd import assets as A, scenes as S
    sd = S.textured_test_scene()
    baked = S.SceneDesc()
    for mi, m in enumerate(sd.meshes):
        mesh_bytes, images = BW.bake_triangle_mesh(m)
        (tmp_path / f"m{mi}.mesh").write_bytes(mesh_bytes)
        for ident, blob in images.items():
            (tmp_path / f"{ident:8x}.image").write_bytes(blob)
        baked.add_mesh(A.load_baked_mesh(str(tmp_path / f"m{mi}.mesh")))
    for mi, xf in sd.instances:
        baked.add_instance(mi, xf)
    W, H = 320, 200
    fc = T._frame_constants(W, H, 1, "textured")[0]
    outs = []
    for d in (sd, baked):
        gp = gpu.GpuPipeline(device, gpu.Scene(device, d), W, H)
        gp.render_inputs(fc)
        torch.cuda.synchronize()
        outs.append((gp.gbuffer.cpu().numpy().copy(), gp.depth.cpu().numpy().copy(), gp.geometric_normal.cpu().numpy().copy()))
    assert (outs[0][1] > 0).mean() > 0.5
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
