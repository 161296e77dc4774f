"""The BLAS builds of one kj_scene_commit as a batch (lbvh_build.hip: build_blas_lbvh_device_batch; scene.cpp): meshes built on the host (binned SAH), on the
device as an LBVH and on the device by PLOC arrive in the SAME commit, interleaved, from one triangle to a few thousand -- more meshes than one workgroup of the
batch's per-mesh kernels handles, single-triangle and two-triangle meshes among them (the builders' n == 1 paths) -- and a second commit adds meshes to pools that
already hold trees. The trees are written into a scratch area with mesh-relative child indices and moved to their dense places in the pool afterwards
(k_blas_place_nodes); whatever the trees, (t, u, v, triangle) of closest-hit, occlusion and back-face-culled queries must equal the oracle's bit for bit
(role of vulkan/ray_tracing.rs:96-169: one BLAS per mesh; world_renderer.rs:604-734: add_mesh)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _soup(rng, ntri, extent):
    """`ntri` small triangles scattered in a box of half-size `extent`, shared vertices among neighbours in the list (so that boxes overlap and Morton codes collide)."""
    centres = rng.uniform(-extent, extent, size=(ntri, 1, 3))
    v = (centres + rng.normal(scale=0.08 * extent + 0.02, size=(ntri, 3, 3))).astype(np.float32).reshape(-1, 3)
    idx = np.arange(3 * ntri, dtype=np.uint32)
    if ntri > 4:      # a few duplicated triangles: equal boxes, equal codes, equal t
        idx[-3:] = idx[:3]
    n = np.tile(np.array([[0, 1, 0]], np.float32), (len(v), 1))
    return v, n, idx


def _rays(rng, n, lo, hi):
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), size=(n, 3))
    d = rng.uniform(lo, hi, size=(n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o; r[:, 4:7] = d; r[:, 7] = np.where(rng.uniform(size=n) < 0.5, 1e4, rng.uniform(0.5, 8.0, size=n))
    return r


def test_one_commit_builds_host_lbvh_and_ploc_meshes_of_all_sizes_as_a_batch(gpu, oracle, device):
    import torch
    from kajiya_amd import scenes
    L = gpu.load()
    rng = np.random.RandomState(77)
    sizes = [1, 2, 3, 5, 9, 33, 130, 700, 2600] * 8 + [1, 2, 4000]      # 75 meshes: > 64 (two workgroups of the per-mesh kernels of a batch)
    desc = scenes.SceneDesc()
    gsc = gpu.Scene(device)
    modes = []

    def add(ntri, mode):
        m = scenes.TriangleMesh(*_soup(rng, ntri, rng.uniform(0.3, 1.5)))
        gpu.check(L.kj_scene_set_blas_build_mode(gsc.h, mode))
        mi = gsc.add_mesh(m)
        assert mi == desc.add_mesh(m)
        modes.append(mode)
        for _ in range(1 + (ntri < 100)):
            ang = rng.uniform(0, 2 * np.pi)
            rot = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            xf = scenes.affine(rot, float(rng.uniform(0.5, 2.0)), rng.uniform(-6, 6, 3))
            gsc.add_instance(mi, xf)
            desc.add_instance(mi, xf)

    for i, ntri in enumerate(sizes):
        add(ntri, (i * 7 + i // 9) % 3)      # host SAH / LBVH / PLOC interleaved, every size under every builder
    assert {(s, m) for s, m in zip(sizes, modes) if s <= 2} >= {(1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)}
    gsc.commit()

    def check(tag):
        osc = oracle.OracleScene(desc)
        assert gsc.stats()["triangles"] == osc.triangle_count, tag
        lo, hi = desc.bounds()
        rays = _rays(rng, 60_000, lo.astype(np.float32), hi.astype(np.float32))
        d_rays = torch.from_numpy(rays).cuda()
        ref, got = osc.trace_closest(rays), gsc.trace_closest(d_rays, len(rays)).cpu().numpy()
        bad = (ref.view(np.uint32) != got.view(np.uint32)).any(axis=1)
        assert not bad.any(), f"{tag}: {int(bad.sum())} of {len(rays)} rays differ"
        assert (ref[:, 0] < 3e38).mean() > 0.05, tag
        assert np.array_equal(osc.trace_any(rays), gsc.trace_any(d_rays, len(rays)).cpu().numpy()), tag
        ref_c, got_c = osc.trace_closest(rays[:20000], cull_back=True), gsc.trace_closest(d_rays[:20000].contiguous(), 20000, cull_back=True).cpu().numpy()
        assert np.array_equal(ref_c.view(np.uint32), got_c.view(np.uint32)), tag

    check("first commit: 75 meshes, three builders")
    for ntri, mode in ((1500, 2), (1, 1), (800, 0), (2200, 1)):      # a second commit: the pools already hold trees; a batch of two LBVH meshes, one PLOC mesh, one host build
        add(ntri, mode)
    gsc.commit()
    check("second commit: four more meshes")
