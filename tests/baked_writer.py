"""Test-side writer for the reference's baked asset files: restates FlattenCtx::finish (kajiya-asset/src/mesh.rs:551-632)
— every Vec becomes a {len, offset} header whose payload lives in a later section, sections are laid out level by level,
offsets are relative to the offset field — for PackedTriMesh (mesh.rs:796-807) and GpuImage (mesh.rs:787-793).
Test infrastructure: produces the fixtures the reader (kajiya_amd/assets.py, csrc/baked_asset.cpp) is checked against."""
import ctypes as C
import struct
import zlib

import numpy as np


class _Ctx:
    def __init__(self):
        self.bytes = bytearray()
        self.deferred = []   # (fixup_addr, nested _Ctx)

    def plain(self, b):
        self.bytes += bytes(b)

    def vec(self, count):
        self.bytes += struct.pack("<QQ", count, 0)
        nested = _Ctx()
        self.deferred.append((len(self.bytes) - 8, nested))
        return nested


def _finish(root):
    sections, level = [], [root]
    index = {}
    while level:
        nxt = []
        for ctx in level:
            index[id(ctx)] = len(sections)
            sections.append(ctx)
            nxt += [n for _, n in ctx.deferred]
        level = nxt
    base, total = [], 0
    for s in sections:
        base.append(total)
        total += len(s.bytes)
    for s, b in zip(sections, base):
        for fixup, nested in s.deferred:
            rel = base[index[id(nested)]] - (fixup + b)
            s.bytes[fixup:fixup + 8] = struct.pack("<Q", rel)
    return b"".join(bytes(s.bytes) for s in sections)


def write_gpu_image(vk_format, extent, mips):
    root = _Ctx()
    root.plain(struct.pack("<i3I", vk_format, *extent))
    mv = root.vec(len(mips))
    for m in mips:
        mv.vec(len(m)).plain(m)
    return _finish(root)


def write_packed_tri_mesh(verts, uvs, tangents, colors, indices, material_ids, materials, map_identities):
    """All arguments are bytes-like / numpy arrays in the Flat element layouts; counts are derived from the byte lengths."""
    root = _Ctx()
    for data, elem in ((verts, 16), (uvs, 8), (tangents, 16), (colors, 16), (indices, 4), (material_ids, 4), (materials, 152), (map_identities, 8)):
        b = np.ascontiguousarray(data).tobytes() if isinstance(data, np.ndarray) else bytes(data)
        assert len(b) % elem == 0
        root.vec(len(b) // elem).plain(b)
    return _finish(root)


def bake_triangle_mesh(mesh, use_lights=False):
    """scenes.TriangleMesh -> (mesh_bytes, {identity: image_bytes}) the way `bin/bake` would emit it with uncompressed RGBA8
    maps: placeholders become 1x1 single-mip images (mesh.rs:845-853), image maps keep their mip chain."""
    d, keep = mesh.pack(use_lights)
    from kajiya_amd.abi import KjMaterialMap, KjMeshMaterial
    n = d.vertex_count

    def grab(ptr, nbytes):
        return C.string_at(ptr, nbytes) if ptr and nbytes else b""
    maps = C.cast(d.maps, C.POINTER(KjMaterialMap))
    images, idents = {}, []
    for k in range(d.map_count):
        mp = maps[k]
        if mp.image_rgba8:
            w, h, levels = mp.width, mp.height, []
            off = 0
            for lv in range(mp.mip_count):
                lw, lh = max(1, w >> lv), max(1, h >> lv)
                levels.append(C.string_at(mp.image_rgba8 + off, lw * lh * 4))
                off += lw * lh * 4
            blob = write_gpu_image(43 if mp.srgb else 37, (w, h, 1), levels)
        else:
            blob = write_gpu_image(37, (1, 1, 1), [bytes(mp.placeholder_rgba)])   # TexGamma::Linear, no mips
        ident = (zlib.crc32(blob) << 32) | zlib.adler32(blob)
        images[ident] = blob
        idents.append(ident)
    mesh_bytes = write_packed_tri_mesh(
        grab(d.verts, n * 16), grab(d.uvs, n * 8), grab(d.tangents, n * 16), grab(d.colors, n * 16), grab(d.indices, d.index_count * 4),
        grab(d.material_ids, n * 4), grab(d.materials, d.material_count * C.sizeof(KjMeshMaterial)), np.array(idents, np.uint64))
    return mesh_bytes, images
