"""Scene inputs in the reference's packed mesh layout (kajiya-asset/src/mesh.rs:75-84,
447-459,796-809) plus seeded procedural stand-ins for the scenes whose assets are
missing from the reference checkout.

A `TriangleMesh` here is the numpy twin of kajiya-asset's `TriangleMesh`; `pack()`
produces the `PackedTriMesh` streams that `kj_scene_add_mesh` consumes.
"""
import ctypes as C
import os
import numpy as np
from .abi import KjMeshDesc, KjMeshMaterial, KjMaterialMap

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")   # reference-held DATA assets converted by scripts/make_golden_assets.py
GOLDEN_DIR = DATA_DIR


def pack_unit_direction_11_10_11(n):
    """kajiya-asset/src/mesh.rs:452-458 (truncating pack)."""
    n = np.clip(np.asarray(n, np.float32), -1.0, 1.0)
    x = ((n[:, 0] * np.float32(0.5) + np.float32(0.5)) * np.float32((1 << 11) - 1)).astype(np.uint32)
    y = ((n[:, 1] * np.float32(0.5) + np.float32(0.5)) * np.float32((1 << 10) - 1)).astype(np.uint32)
    z = ((n[:, 2] * np.float32(0.5) + np.float32(0.5)) * np.float32((1 << 11) - 1)).astype(np.uint32)
    return (z << np.uint32(21)) | (y << np.uint32(11)) | x


def build_mip_chain(img):
    """(h, w, 4) uint8 -> all mip levels back to back (2x2 box filter on the stored bytes, like the reference baker's
    non-gamma-correct resize; kajiya-asset/src/image.rs:252-275 uses Lanczos3 — the baker is out of scope, this is test data)."""
    img = np.ascontiguousarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[2] == 4
    levels = [img]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        a = levels[-1].astype(np.float32)
        h, w = a.shape[:2]
        nh, nw = max(1, h // 2), max(1, w // 2)
        a = a[: nh * 2 if h > 1 else 1, : nw * 2 if w > 1 else 1]
        if h > 1:
            a = 0.5 * (a[0::2] + a[1::2])
        if w > 1:
            a = 0.5 * (a[:, 0::2] + a[:, 1::2])
        levels.append(np.clip(np.rint(a), 0, 255).astype(np.uint8))
    return np.concatenate([l.reshape(-1) for l in levels]), len(levels)


class TriangleMesh:
    def __init__(self, positions, normals, indices, material_ids=None, materials=None, colors=None, uvs=None):
        self.positions = np.ascontiguousarray(positions, np.float32)
        self.normals = np.ascontiguousarray(normals, np.float32)
        self.indices = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        n = len(self.positions)
        self.material_ids = np.zeros(n, np.uint32) if material_ids is None else np.ascontiguousarray(material_ids, np.uint32)
        # materials: list of dict(base_color[4], roughness, metalness, emissive[3])
        self.materials = materials or [dict(base_color=(0.8, 0.8, 0.8, 1.0), roughness=0.9, metalness=0.0, emissive=(0, 0, 0))]
        self.colors = None if colors is None else np.ascontiguousarray(colors, np.float32)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, np.float32)

    @property
    def triangle_count(self):
        return len(self.indices) // 3

    def pack(self, use_lights=False):
        """Returns (KjMeshDesc, keepalive list). load_gltf_material (mesh.rs:108-258): 4 placeholder maps
        per material [normal, spec, albedo, emissive]."""
        n = len(self.positions)
        verts = np.zeros(n, dtype=[("pos", np.float32, 3), ("normal", np.uint32)])
        verts["pos"] = self.positions
        verts["normal"] = pack_unit_direction_11_10_11(self.normals)
        mats = (KjMeshMaterial * len(self.materials))()
        keep_images = []
        maps = (KjMaterialMap * (4 * len(self.materials)))()
        for i, m in enumerate(self.materials):
            mm = mats[i]
            for k in range(4):
                mm.base_color_mult[k] = float(m["base_color"][k])
                mm.maps[k] = 4 * i + k
            mm.roughness_mult = float(m["roughness"])
            mm.metalness_factor = float(m["metalness"])
            for k in range(3):
                mm.emissive[k] = float(m["emissive"][k])
            mm.flags = 0
            for k in range(4):
                for j, v in enumerate((1.0, 0.0, 0.0, 1.0, 0.0, 0.0)):
                    mm.map_transforms[k * 6 + j] = v
            for k, rgba in enumerate(((127, 127, 255, 255), (255, 255, 127, 255), (255, 255, 255, 255), (255, 255, 255, 255))):
                for j in range(4):
                    maps[4 * i + k].placeholder_rgba[j] = rgba[j]
                maps[4 * i + k].image_rgba8 = None
            # optional image maps (MeshMaterialMap::Image): "spec_image" / "albedo_image" / "emissive_image" = (h, w, 4) uint8 arrays;
            # "map_transforms" = {slot: (a, b, c, d, tx, ty)} with slots 0 albedo, 2 spec, 3 emissive (mesh.rs:125-230)
            for k, key, srgb in ((1, "spec_image", 0), (2, "albedo_image", 1), (3, "emissive_image", 1)):
                if m.get(key) is not None:
                    chain, nm = build_mip_chain(m[key])
                    keep_images.append(chain)
                    mp = maps[4 * i + k]
                    mp.image_rgba8 = chain.ctypes.data
                    mp.width, mp.height, mp.mip_count, mp.srgb = m[key].shape[1], m[key].shape[0], nm, srgb
            for slot, t in (m.get("map_transforms") or {}).items():
                for j, v in enumerate(t):
                    mm.map_transforms[slot * 6 + j] = float(v)
        d = KjMeshDesc()
        keep = [verts, self.indices, self.material_ids, mats, maps, keep_images]
        d.verts = verts.ctypes.data
        d.vertex_count = n
        d.uvs = self.uvs.ctypes.data if self.uvs is not None else None
        d.tangents = None
        d.colors = self.colors.ctypes.data if self.colors is not None else None
        d.material_ids = self.material_ids.ctypes.data
        d.indices = self.indices.ctypes.data
        d.index_count = len(self.indices)
        d.materials = C.cast(mats, C.c_void_p)
        d.material_count = len(self.materials)
        d.maps = C.cast(maps, C.c_void_p)
        d.map_count = 4 * len(self.materials)
        d.use_lights = 1 if use_lights else 0
        keep.append(d)
        return d, keep


def translation(t):
    m = np.zeros((3, 4), np.float32)
    m[:, :3] = np.eye(3)
    m[:, 3] = t
    return m


def affine(rot3=None, scale=1.0, t=(0, 0, 0)):
    m = np.zeros((3, 4), np.float32)
    m[:, :3] = (np.eye(3) if rot3 is None else np.asarray(rot3)) * scale
    m[:, 3] = t
    return m


class SceneDesc:
    """meshes + instances (mesh index, 3x4 transform); the numpy twin of a kajiya `.ron` scene."""

    def __init__(self):
        self.meshes = []
        self.instances = []

    def add_mesh(self, mesh):
        self.meshes.append(mesh)
        return len(self.meshes) - 1

    def add_instance(self, mesh_idx, xform3x4):
        self.instances.append((mesh_idx, np.ascontiguousarray(xform3x4, np.float32)))

    @property
    def triangle_count(self):
        return sum(self.meshes[m].triangle_count for m, _ in self.instances)

    def bounds(self):
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for m, x in self.instances:
            p = self.meshes[m].positions @ x[:, :3].T + x[:, 3]
            lo, hi = np.minimum(lo, p.min(0)), np.maximum(hi, p.max(0))
        return lo, hi


def cornell_box():
    """assets/scenes/cornell_box.ron: cornell_box/scene.gltf scaled x2 at (0,-1,0).
    Geometry from kajiya_amd/data/cornell_box.npz (scripts/make_golden_assets.py)."""
    s = SceneDesc()
    s.add_instance(s.add_mesh(_baked_gltf("cornell_box.npz")), translation((0.0, -1.0, 0.0)))
    return s


def _baked_gltf(name):
    z = np.load(os.path.join(DATA_DIR, name))
    mats = [dict(base_color=z["mat_base_color"][i], roughness=float(z["mat_roughness"][i]),
                 metalness=float(z["mat_metalness"][i]), emissive=z["mat_emissive"][i]) for i in range(len(z["mat_roughness"]))]
    return TriangleMesh(z["positions"], z["normals"], z["indices"], z["material_ids"], mats)


def pica_diorama():
    """assets/scenes/pica.ron: pica_pica_-_mini_diorama_01/scene.gltf scaled x0.1 at the origin -- the one real production
    asset in the reference checkout (76 k triangles, 170 primitives, node transforms baked by the importer). Geometry +
    material factors from kajiya_amd/data/pica_diorama.npz (scripts/make_golden_assets.py); image maps are not carried."""
    s = SceneDesc()
    s.add_instance(s.add_mesh(_baked_gltf("pica_diorama.npz")), affine())
    return s


# ---------------------------------------------------------------------------- procedural stand-ins
def _box(size=(1, 1, 1)):
    sx, sy, sz = (0.5 * s for s in size)
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    P, N, I = [], [], []
    for n, u, v in faces:
        n, u, v = (np.array(a, np.float32) for a in (n, u, v))
        base = len(P)
        for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            P.append((n + a * u + b * v) * (sx, sy, sz))
            N.append(n)
        I += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(P, np.float32), np.array(N, np.float32), np.array(I, np.uint32)


def _grid_patch(nu, nv, fn):
    """Tessellated parametric patch: fn(u,v)->(pos,normal) arrays; nu*nv*2 triangles."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32), indexing="ij")
    p, n = fn(u.reshape(-1), v.reshape(-1))
    idx = np.arange((nu + 1) * (nv + 1), dtype=np.uint32).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.stack([a, b, c, a, c, d], axis=-1).reshape(-1)
    return p.astype(np.float32), n.astype(np.float32), tris.astype(np.uint32)


def _sphere(nu, nv, r=1.0):
    def fn(u, v):
        th, ph = u * 2 * np.pi, v * np.pi
        n = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1)
        return n * r, n
    p, n, i = _grid_patch(nu, nv, fn)
    t = i.reshape(-1, 3)[:, ::-1].reshape(-1)  # outward-facing winding
    # drop degenerate pole triangles
    tri = t.reshape(-1, 3)
    e1 = p[tri[:, 1]] - p[tri[:, 0]]; e2 = p[tri[:, 2]] - p[tri[:, 0]]
    keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-12
    return p, n, tri[keep].reshape(-1).astype(np.uint32)


def _terrain(n, extent, amp, seed):
    rng = np.random.RandomState(seed)
    k = rng.uniform(0.3, 2.2, size=(6, 2)); ph = rng.uniform(0, 6.28, size=6); a = rng.uniform(0.2, 1.0, size=6)
    def h(x, z):
        y = np.zeros_like(x)
        for i in range(6):
            y += a[i] * np.sin(k[i, 0] * x + k[i, 1] * z + ph[i])
        return amp * y / a.sum()
    def fn(u, v):
        x = (u - 0.5) * extent; z = (v - 0.5) * extent
        e = extent / n
        y = h(x, z)
        dx = (h(x + e, z) - h(x - e, z)) / (2 * e); dz = (h(x, z + e) - h(x, z - e)) / (2 * e)
        nn = np.stack([-dx, np.ones_like(dx), -dz], -1)
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        return np.stack([x, y, z], -1), nn
    p, nn, i = _grid_patch(n, n, fn)
    return p, nn, i.reshape(-1, 3)[:, ::-1].reshape(-1).astype(np.uint32)


def _mat(rng, emissive=None):
    alb = rng.uniform(0.2, 0.8, size=3)
    return dict(base_color=(alb[0], alb[1], alb[2], 1.0), roughness=float(rng.uniform(0.3, 1.0)), metalness=0.0,
                emissive=(0, 0, 0) if emissive is None else emissive)


def textured_test_scene(seed=11):
    """A small scene whose materials use image maps with mip chains and uv transforms: a tiled floor, a wall, two boxes and a
    sphere with checker / noise albedo, a roughness-metalness map and an emissive map. Used by the texture parity tests."""
    rng = np.random.RandomState(seed)
    sd = SceneDesc()

    def checker(n, cells, c0, c1):
        yy, xx = np.mgrid[0:n, 0:n]
        m = (((xx * cells) // n + (yy * cells) // n) & 1).astype(bool)
        img = np.zeros((n, n, 4), np.uint8)
        img[...] = np.array(c0, np.uint8)
        img[m] = np.array(c1, np.uint8)
        return img
    noise = rng.randint(0, 256, size=(64, 32, 4)).astype(np.uint8); noise[..., 3] = 255
    spec = np.zeros((32, 32, 4), np.uint8); spec[..., 0] = rng.randint(60, 256, size=(32, 32)); spec[..., 1] = (rng.uniform(size=(32, 32)) < 0.3) * 255; spec[..., 2:] = 255
    emis = checker(16, 4, (0, 0, 0, 255), (255, 160, 40, 255))
    mats = [
        dict(base_color=(1, 1, 1, 1), roughness=1.0, metalness=1.0, emissive=(0, 0, 0), albedo_image=checker(128, 8, (200, 60, 40, 255), (235, 235, 220, 255)),
             spec_image=spec, map_transforms={0: (4.0, 0.0, 0.0, 4.0, 0.25, 0.0), 2: (1.0, 0.0, 0.0, 1.0, 0.0, 0.0)}),
        dict(base_color=(0.9, 0.9, 1.0, 1), roughness=0.8, metalness=0.0, emissive=(0, 0, 0), albedo_image=noise,
             map_transforms={0: (0.0, 1.0, -1.0, 0.0, 0.5, 0.5)}),
        dict(base_color=(0.6, 0.6, 0.6, 1), roughness=0.7, metalness=0.0, emissive=(2.0, 2.0, 2.0), emissive_image=emis,
             albedo_image=checker(64, 2, (90, 120, 200, 255), (30, 40, 60, 255)), map_transforms={3: (2.0, 0.0, 0.0, 2.0, 0.0, 0.0)}),
    ]

    def quad(p0, du, dv, n, mat, uvscale=1.0):
        P = np.array([p0, p0 + du, p0 + du + dv, p0 + dv], np.float32)
        N = np.tile(np.asarray(n, np.float32)[None], (4, 1))
        UV = np.array([[0, 0], [uvscale, 0], [uvscale, uvscale], [0, uvscale]], np.float32)
        return TriangleMesh(P, N, np.array([0, 1, 2, 0, 2, 3], np.uint32), materials=[mat], uvs=UV)
    sd.add_instance(sd.add_mesh(quad(np.array([-8.0, 0, -8.0]), np.array([16.0, 0, 0]), np.array([0, 0, 16.0]), (0, 1, 0), mats[0], 2.0)), affine())
    sd.add_instance(sd.add_mesh(quad(np.array([-8.0, 0, -6.0]), np.array([16.0, 0, 0]), np.array([0, 7.0, 0]), (0, 0, 1), mats[1], 3.0)), affine())
    bp, bn, bidx = _box((1.5, 1.5, 1.5))
    box_uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), (len(bp) // 4 + 1, 1))[: len(bp)]
    bm = TriangleMesh(bp, bn, bidx, materials=[mats[2]], uvs=box_uv)
    bi = sd.add_mesh(bm)
    sd.add_instance(bi, affine(t=(-2.5, 0.75, -1.0)))
    sd.add_instance(bi, affine(scale=0.6, t=(2.0, 0.45, 1.5)))
    sp, sn, sidx = _sphere(24, 16, 1.2)
    th = np.arctan2(sp[:, 2], sp[:, 0]) / (2 * np.pi) + 0.5
    ph = np.arccos(np.clip(sp[:, 1] / 1.2, -1, 1)) / np.pi
    sm = TriangleMesh(sp, sn, sidx, materials=[mats[0]], uvs=np.stack([th, ph], -1).astype(np.float32))
    sd.add_instance(sd.add_mesh(sm), affine(t=(0.5, 1.2, -2.5)))
    return sd


def glossy_test_scene():
    """Reflection test scene: a mirror-like metal floor (roughness 0.08), a glossy dielectric wall (0.3), a rough wall (0.8, its
    reflection rays come from rtdgi's candidates), an emissive box, two diffuse boxes and a smooth metal sphere."""
    sd = SceneDesc()

    def quad(p0, du, dv, n, mat):
        P = np.array([p0, p0 + du, p0 + du + dv, p0 + dv], np.float32)
        N = np.tile(np.asarray(n, np.float32)[None], (4, 1))
        return TriangleMesh(P, N, np.array([0, 1, 2, 0, 2, 3], np.uint32), materials=[mat])
    floor = dict(base_color=(0.95, 0.93, 0.88, 1), roughness=0.08, metalness=1.0, emissive=(0, 0, 0))
    gloss = dict(base_color=(0.2, 0.35, 0.7, 1), roughness=0.3, metalness=0.0, emissive=(0, 0, 0))
    rough = dict(base_color=(0.7, 0.7, 0.7, 1), roughness=0.8, metalness=0.0, emissive=(0, 0, 0))
    sd.add_instance(sd.add_mesh(quad(np.array([-8.0, 0, -8.0]), np.array([0, 0, 16.0]), np.array([16.0, 0, 0]), (0, 1, 0), floor)), affine())
    sd.add_instance(sd.add_mesh(quad(np.array([-8.0, 0, -6.0]), np.array([16.0, 0, 0]), np.array([0, 7.0, 0]), (0, 0, 1), gloss)), affine())
    sd.add_instance(sd.add_mesh(quad(np.array([-6.0, 0, 8.0]), np.array([0, 0, -16.0]), np.array([0, 7.0, 0]), (1, 0, 0), rough)), affine())
    bp, bn, bidx = _box((1.5, 1.5, 1.5))
    lamp = TriangleMesh(bp, bn, bidx, materials=[dict(base_color=(0.8, 0.8, 0.8, 1), roughness=0.6, metalness=0.0, emissive=(6.0, 3.0, 1.0))])
    red = TriangleMesh(bp, bn, bidx, materials=[dict(base_color=(0.8, 0.15, 0.1, 1), roughness=0.9, metalness=0.0, emissive=(0, 0, 0))])
    sd.add_instance(sd.add_mesh(lamp), affine(scale=0.6, t=(2.0, 0.45, 1.5)))
    ri = sd.add_mesh(red)
    sd.add_instance(ri, affine(t=(-2.5, 0.75, -1.0)))
    sd.add_instance(ri, affine(scale=0.5, t=(0.0, 0.375, 2.5)))
    sp, sn, sidx = _sphere(32, 20, 1.2)
    ball = TriangleMesh(sp, sn, sidx, materials=[dict(base_color=(1.0, 0.85, 0.55, 1), roughness=0.15, metalness=1.0, emissive=(0, 0, 0))])
    sd.add_instance(sd.add_mesh(ball), affine(t=(0.5, 1.2, -2.5)))
    return sd


def procedural_city(target_tris=1_000_000, seed=1234, n_instances=64):
    """Stand-in for `battle.ron` (asset missing): 8 meshes (boxes + tessellated
    spheres) x 64 instances over a rolling ground, ~target_tris triangles, 32 emissive triangles."""
    rng = np.random.RandomState(seed)
    s = SceneDesc()
    ground_n = max(8, int(np.sqrt(target_tris * 0.25 / 2)))
    gp, gn, gi = _terrain(ground_n, 60.0, 0.6, seed)
    s.add_instance(s.add_mesh(TriangleMesh(gp, gn, gi, materials=[_mat(rng)])), translation((0, 0, 0)))
    remaining = max(target_tris - len(gi) // 3, 1000)
    per_inst = remaining / n_instances
    mesh_ids = []
    for m in range(8):
        if m % 2 == 0:
            nu = max(4, int(np.sqrt(per_inst / 2)))
            p, n, i = _sphere(nu, nu, 1.0)
        else:
            # tessellated box: each face a grid
            nu = max(1, int(np.sqrt(per_inst / 12)))
            P, N, I = [], [], []
            bp, bn, bi = _box((2, 2, 2))
            for f in range(6):
                c = bp[f * 4:(f + 1) * 4]
                def fn(u, v, c=c, nrm=bn[f * 4]):
                    pos = (c[0][None] * ((1 - u) * (1 - v))[:, None] + c[1][None] * (u * (1 - v))[:, None] +
                           c[2][None] * (u * v)[:, None] + c[3][None] * ((1 - u) * v)[:, None])
                    return pos, np.repeat(nrm[None], len(u), 0)
                fp, fnn, fi = _grid_patch(nu, nu, fn)
                I.append(fi + sum(len(x) for x in P)); P.append(fp); N.append(fnn)
            p, n, i = np.concatenate(P), np.concatenate(N), np.concatenate(I).astype(np.uint32)
        mats = [_mat(rng)]
        mids = np.zeros(len(p), np.uint32)
        if m < 2:
            # 16 emissive triangles on each of the first two meshes: dedicated verts + material 1
            mats.append(dict(base_color=(1, 1, 1, 1), roughness=1.0, metalness=0.0, emissive=(8.0, 6.0, 4.0)))
            tri = i.reshape(-1, 3)
            sel = rng.choice(len(tri), 16, replace=False)
            newv = tri[sel].reshape(-1)
            base = len(p)
            p = np.concatenate([p, p[newv]]); n = np.concatenate([n, n[newv]])
            mids = np.concatenate([mids, np.ones(len(newv), np.uint32)])
            tri = tri.copy()
            tri[sel] = (base + np.arange(len(newv), dtype=np.uint32)).reshape(-1, 3)
            i = tri.reshape(-1)
        mesh_ids.append(s.add_mesh(TriangleMesh(p, n, i, mids, mats)))
    for k in range(n_instances):
        m = mesh_ids[k % 8]
        ang = rng.uniform(0, 2 * np.pi)
        c, sn = np.cos(ang), np.sin(ang)
        rot = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]])
        scale = rng.uniform(0.6, 2.2)
        pos = (rng.uniform(-22, 22), scale * rng.uniform(0.6, 1.4), rng.uniform(-22, 22))
        s.add_instance(m, affine(rot, scale, pos))
    return s


def procedural_ruins(target_tris=4_000_000, seed=5678):
    """Stand-in for the Ruins scene (never in the repo, README.md:44-45): terrain + rows of
    tessellated columns and arches, albedo U[0.2,0.8], roughness U[0.3,1], no textures."""
    rng = np.random.RandomState(seed)
    s = SceneDesc()
    ground_n = max(8, int(np.sqrt(target_tris * 0.3 / 2)))
    gp, gn, gi = _terrain(ground_n, 80.0, 1.2, seed)
    s.add_instance(s.add_mesh(TriangleMesh(gp, gn, gi, materials=[_mat(rng)])), translation((0, 0, 0)))
    n_cols, n_arch = 48, 16
    per = (target_tris - len(gi) // 3) / (n_cols + n_arch)
    def column(nu, nv):
        def fn(u, v):
            th = u * 2 * np.pi
            r = 0.5 * (1.0 + 0.06 * np.cos(12 * th)) * (1.0 - 0.15 * v)
            pos = np.stack([r * np.cos(th), v * 6.0, r * np.sin(th)], -1)
            nn = np.stack([np.cos(th), np.full_like(th, 0.08), np.sin(th)], -1)
            return pos, nn / np.linalg.norm(nn, axis=1, keepdims=True)
        p, n, i = _grid_patch(nu, nv, fn)
        return p, n, i.reshape(-1, 3)[:, ::-1].reshape(-1).astype(np.uint32)
    def arch(nu, nv):
        def fn(u, v):
            th = u * np.pi
            ph = v * 2 * np.pi
            R, r = 3.0, 0.45
            cx, cy = R * np.cos(th), R * np.sin(th)
            pos = np.stack([cx + r * np.cos(ph) * np.cos(th), cy + r * np.cos(ph) * np.sin(th) + 3.0, r * np.sin(ph)], -1)
            nn = np.stack([np.cos(ph) * np.cos(th), np.cos(ph) * np.sin(th), np.sin(ph)], -1)
            return pos, nn
        return _grid_patch(nu, nv, fn)
    nu = max(6, int(np.sqrt(per / 2)))
    cm = [s.add_mesh(TriangleMesh(*column(nu, nu), materials=[_mat(rng)])) for _ in range(4)]
    am = [s.add_mesh(TriangleMesh(*arch(nu, nu), materials=[_mat(rng)])) for _ in range(2)]
    for k in range(n_cols):
        x = (k % 12 - 5.5) * 5.0; z = (k // 12 - 1.5) * 9.0
        s.add_instance(cm[k % 4], affine(None, rng.uniform(0.8, 1.3), (x, -0.5, z)))
    for k in range(n_arch):
        x = (k % 8 - 3.5) * 7.5; z = (k // 8 - 0.5) * 18.0
        s.add_instance(am[k % 2], affine(None, 1.0, (x, 0.0, z)))
    return s
