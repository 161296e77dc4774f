"""Host mirror of the reference's baked-asset loading: `cache/<name>.mesh` + `cache/<identity:08x>.image`
files written by `bin/bake` (kajiya-asset-pipe/src/lib.rs:38-60) and consumed by WorldRenderer::add_mesh /
load_gpu_image_asset (world_renderer.rs:297-322,604-700).

The flat-format parsing is the C-ABI's (`kj_baked_mesh_view`, `kj_baked_image_view`, `kj_baked_image_mip`,
csrc/baked_asset.cpp); this module maps files, resolves the mesh's map identities to image files, decodes every mip level
to RGBA8 texels (natively: RGBA8, BC1, BC3, BC4, BC5, BC7 — the texture unit's job in the reference; Pillow is only consulted for
BC2 / BC6H, which kajiya's baker never emits) and hands `KjMeshDesc` to `kj_scene_add_mesh`, which is where the reference's own add_mesh starts.
"""
import ctypes as C
import mmap
import os

import numpy as np

from .abi import KjMaterialMap, KjMeshDesc
from . import lib as klib

VK_SRGB_FORMATS = {43, 132, 134, 136, 138, 146}   # R8G8B8A8_SRGB, BC1/BC2/BC3/BC7 *_SRGB_BLOCK
VK_BC7 = {145, 146}
VK_BC2 = {135, 136}
VK_BC6H = {143, 144}


class KjBakedMeshView(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("uvs", C.c_void_p), ("tangents", C.c_void_p), ("colors", C.c_void_p), ("indices", C.c_void_p),
                ("material_ids", C.c_void_p), ("materials", C.c_void_p), ("map_identities", C.POINTER(C.c_uint64)),
                ("vertex_count", C.c_uint32), ("index_count", C.c_uint32), ("material_count", C.c_uint32), ("map_count", C.c_uint32)]


class KjBakedImageView(C.Structure):
    _fields_ = [("vk_format", C.c_uint32), ("extent", C.c_uint32 * 3), ("mip_count", C.c_uint32)]


def _bind():
    L = klib.load()
    if not getattr(L, "_baked_bound", False):
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.kj_baked_mesh_view.argtypes = [vp, u64, C.POINTER(KjBakedMeshView)]
        L.kj_baked_image_view.argtypes = [vp, u64, C.POINTER(KjBakedImageView)]
        L.kj_baked_image_mip.argtypes = [vp, u64, u32, C.POINTER(vp), C.POINTER(u64)]
        L.kj_baked_image_decode_rgba8.argtypes = [u32, vp, u64, u32, u32, vp]
        for n in ("kj_baked_mesh_view", "kj_baked_image_view", "kj_baked_image_mip", "kj_baked_image_decode_rgba8"):
            getattr(L, n).restype = C.c_int32
        L._baked_bound = True
    return L


def _as_buffer(data):
    """bytes / bytearray / mmap / uint8 ndarray -> (uint8 ndarray over the same memory, address, size)."""
    arr = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    return arr, arr.ctypes.data, arr.size


def _map_file(path):
    with open(path, "rb") as f:
        if os.fstat(f.fileno()).st_size == 0:
            raise klib.KjError(f"{path}: empty file")
        return mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)


def decode_baked_image(data):
    """GpuImage::Flat bytes -> dict(format, extent, srgb, levels=[(h, w, 4) uint8 ...]) with level k = max(1, w >> k) x
    max(1, h >> k) as the sampler sees it (block-compressed tails are stored as one padded 4x4 block, image.rs:226-246)."""
    L = _bind()
    arr, addr, size = _as_buffer(data)
    view = KjBakedImageView()
    klib.check(L.kj_baked_image_view(addr, size, C.byref(view)))
    fmt, (w0, h0) = view.vk_format, (view.extent[0], view.extent[1])
    if w0 == 0 or h0 == 0:
        raise klib.KjError("baked image: zero extent")
    levels = []
    for k in range(view.mip_count):
        w, h = max(1, w0 >> k), max(1, h0 >> k)
        p, n = C.c_void_p(), C.c_uint64()
        klib.check(L.kj_baked_image_mip(addr, size, k, C.byref(p), C.byref(n)))
        out = np.empty((h, w, 4), np.uint8)
        st = L.kj_baked_image_decode_rgba8(fmt, p, n.value, w, h, out.ctypes.data)
        if st == 5 and (fmt in VK_BC2 or fmt in VK_BC6H):   # KJ_ERR_UNSUPPORTED: formats the baker never emits (DDS pass-through only)
            from PIL import Image
            pw, ph = (w + 3) // 4 * 4, (h + 3) // 4 * 4
            need = pw * ph   # 16 bytes per 4x4 block
            if n.value < need:
                raise klib.KjError("baked image: mip shorter than its block count")
            raw = C.string_at(p, need)
            bcn = 2 if fmt in VK_BC2 else 6
            out = np.asarray(Image.frombytes("RGBA", (pw, ph), raw, "bcn", bcn).convert("RGBA"))[:h, :w].copy()
        else:
            klib.check(st)
        levels.append(out)
    return dict(format=fmt, extent=(w0, h0, view.extent[2]), srgb=fmt in VK_SRGB_FORMATS, levels=levels)


class BakedMesh:
    """A mapped `.mesh` file plus its decoded images; `pack()` yields the KjMeshDesc for GpuScene.add_mesh (same protocol
    as scenes.TriangleMesh). The vertex streams stay zero-copy views of the file."""

    def __init__(self, mesh_bytes, image_loader):
        L = _bind()
        self._data = mesh_bytes
        self._arr, addr, size = _as_buffer(mesh_bytes)
        self.view = KjBakedMeshView()
        klib.check(L.kj_baked_mesh_view(addr, size, C.byref(self.view)))
        v = self.view
        self.map_identities = [int(v.map_identities[i]) for i in range(v.map_count)]
        # unique images are loaded once (world_renderer.rs:610-631)
        self.images = {ident: image_loader(ident) for ident in sorted(set(self.map_identities))}

    @property
    def triangle_count(self):
        return self.view.index_count // 3

    @property
    def vertex_count(self):
        return self.view.vertex_count

    def stream(self, name):
        """Numpy copy of one vertex stream (for inspection / tests)."""
        v = self.view
        spec = {"verts": (v.verts, v.vertex_count * 4, np.uint32), "uvs": (v.uvs, v.vertex_count * 2, np.float32),
                "tangents": (v.tangents, v.vertex_count * 4, np.float32), "colors": (v.colors, v.vertex_count * 4, np.float32),
                "indices": (v.indices, v.index_count, np.uint32), "material_ids": (v.material_ids, v.vertex_count, np.uint32),
                "materials": (v.materials, v.material_count * 38, np.uint32)}[name]
        if not spec[0]:
            return np.zeros(0, spec[2])
        return np.ctypeslib.as_array(C.cast(spec[0], C.POINTER(C.c_uint32)), (spec[1],)).view(spec[2]).copy()

    def pack(self, use_lights=False):
        v = self.view
        maps = (KjMaterialMap * max(1, v.map_count))()
        keep = [self._data, self._arr, maps]
        for k, ident in enumerate(self.map_identities):
            img = self.images[ident]
            mp = maps[k]
            if tuple(img["extent"][:2]) == (1, 1) and len(img["levels"]) == 1 and not img["srgb"]:
                # a baked MeshMaterialMap::Placeholder (mesh.rs:845-853: 1x1 linear RGBA8, no mips): constant for every uv / LOD
                for j in range(4):
                    mp.placeholder_rgba[j] = int(img["levels"][0][0, 0, j])
                mp.image_rgba8 = None
                continue
            chain = np.concatenate([lv.reshape(-1) for lv in img["levels"]])
            keep.append(chain)
            mp.image_rgba8 = chain.ctypes.data
            mp.width, mp.height = img["extent"][0], img["extent"][1]
            mp.mip_count = len(img["levels"])
            mp.srgb = 1 if img["srgb"] else 0
        d = KjMeshDesc()
        d.verts, d.vertex_count = v.verts, v.vertex_count
        d.uvs, d.tangents, d.colors, d.material_ids = v.uvs, v.tangents, v.colors, v.material_ids
        d.indices, d.index_count = v.indices, v.index_count
        d.materials, d.material_count = v.materials, v.material_count
        d.maps, d.map_count = C.cast(maps, C.c_void_p), v.map_count
        d.use_lights = 1 if use_lights else 0
        keep.append(d)
        return d, keep


def load_baked_mesh(mesh_path, cache_dir=None):
    """`cache/<name>.mesh` -> BakedMesh. Images come from `<cache_dir>/<identity:08x>.image` (world_renderer.rs:301-305);
    cache_dir defaults to the mesh file's directory."""
    cache_dir = cache_dir or os.path.dirname(os.path.abspath(mesh_path))

    def loader(ident):
        path = os.path.join(cache_dir, f"{ident:8x}.image")      # Rust `{:8.8x}`: width 8, space padded
        if not os.path.exists(path):
            path = os.path.join(cache_dir, f"{ident:08x}.image")
        if not os.path.exists(path):
            raise klib.KjError(f"baked mesh {mesh_path}: image {path} is missing")
        return decode_baked_image(_map_file(path))

    return BakedMesh(_map_file(mesh_path), loader)
