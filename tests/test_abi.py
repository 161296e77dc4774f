"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/kajiya_amd.h declares; struct layouts match the reference's; error paths report, not crash."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "kajiya_amd.h")).read()
    return sorted(set(re.findall(r"\b(kj_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from kajiya_amd import lib
    L = lib.load()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(lib.EXPORTS) <= set(names)
    assert L.kj_abi_version() == 1


def test_struct_layouts():
    from kajiya_amd import abi
    assert C.sizeof(abi.KjFrameConstants) == 1216      # frame_constants.rs:13-37
    assert C.sizeof(abi.KjViewConstants) == 11 * 64 + 16
    assert abi.KjFrameConstants.frame_index.offset == 736
    assert abi.KjFrameConstants.ircache_cascades.offset == 832
    assert C.sizeof(abi.KjMeshMaterial) == 152
    assert C.sizeof(abi.KjPackedVertex) == 16


ABI_IDS = ["KjFrameConstants", "KjViewConstants", "KjMeshMaterial", "KjPackedVertex", "KjMaterialMap", "KjMeshDesc", "KjTriangleLight", "KjGbufferDepth", "KjRtdgiRenderParams",
           "KjRtdgiOutput", "KjTaaOutput", "KjRtrTables", "KjRtrParams", "KjSplitRank", "KjSplitFrame", "KjBakedMeshView", "KjBakedImageView", "KjSplitProfile"]     # enum KjAbiStruct, in order


def _rust_repr_c_structs(text):
    """`#[repr(C)] pub struct Name { pub a: T, ... }` blocks of INTEGRATION.md -> {name: [(field, type)]} (opaque `_p: [u8; 0]` handles skipped)."""
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{([^{}]*)\}", text, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        fields = re.findall(r"pub\s+(\w+)\s*:\s*([^,}]+?)\s*(?:,|$)", body, re.S)
        if fields:
            out[m.group(1)] = [(n, t.strip()) for n, t in fields]
    return out


def _rust_size_align(ty, structs):
    ty = ty.strip()
    if ty.startswith("*"):
        return 8, 8
    prim = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8}
    if ty in prim:
        return prim[ty], prim[ty]
    m = re.match(r"\[(.+);\s*(\d+)\]$", ty)
    if m:
        s, a = _rust_size_align(m.group(1), structs)
        return s * int(m.group(2)), a
    off, align = 0, 1
    for _, ft in structs[ty]:
        s, a = _rust_size_align(ft, structs)
        off = (off + a - 1) // a * a + s
        align = max(align, a)
    return (off + align - 1) // align * align, align


def test_every_struct_of_the_boundary_has_the_library_s_size():
    """kj_abi_struct_size(id) is sizeof() inside the library. The ctypes binding (kajiya_amd/abi.py) must agree for every struct it declares,
    and so must every `#[repr(C)]` struct INTEGRATION.md hands a kajiya maintainer -- round 3's text had KjRtdgiRenderParams twelve bytes short."""
    from kajiya_amd import lib, abi
    L = lib.load()
    L.kj_abi_struct_size.restype = C.c_uint32
    L.kj_abi_struct_size.argtypes = [C.c_uint32]
    sizes = {name: L.kj_abi_struct_size(i) for i, name in enumerate(ABI_IDS)}
    assert all(v > 0 for v in sizes.values()) and L.kj_abi_struct_size(len(ABI_IDS)) == 0, sizes
    hdr = open(os.path.join(ROOT, "include", "kajiya_amd.h")).read()
    enum = re.search(r"enum KjAbiStruct \{(.*?)\}", hdr, re.S).group(1)
    assert len(re.findall(r"KJ_ABI_\w+", enum)) == len(ABI_IDS) + 1            # + KJ_ABI_STRUCT_COUNT
    checked = 0
    for name, size in sizes.items():
        if hasattr(abi, name):
            assert C.sizeof(getattr(abi, name)) == size, (name, C.sizeof(getattr(abi, name)), size)
            checked += 1
    assert checked >= 12, checked
    structs = _rust_repr_c_structs(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    in_doc = [n for n in sizes if n in structs]
    assert {"KjGbufferDepth", "KjRtdgiRenderParams", "KjRtdgiOutput"} <= set(in_doc), in_doc
    for name in in_doc:
        assert _rust_size_align(name, structs)[0] == sizes[name], (name, _rust_size_align(name, structs)[0], sizes[name], structs[name])


def test_error_reporting_without_gpu():
    """No compute calls: only argument validation paths (must not touch a device)."""
    from kajiya_amd import lib
    L = lib.load()
    out = C.c_void_p()
    st = L.kj_scene_create(None, C.byref(out))
    assert st != 0 and b"null" in L.kj_last_error()
    st = L.kj_rtdgi_render(None, None, None, None)
    assert st != 0
    # every handle-taking entry point rejects NULL with KJ_ERR_INVALID_ARGUMENT (1) and a message, without touching a device
    for name, args in (("kj_rtr_create", (None, None, None)), ("kj_rtr_trace", (None, None, None)), ("kj_rtr_filter_temporal", (None, None, None, None)),
                       ("kj_rtr_render_specular_lights", (None, None, None)), ("kj_rtr_surface", (None, None, None, None)), ("kj_rtr_ray_counts", (None, None, None)),
                       ("kj_ircache_prepare", (None, None)), ("kj_taa_render", (None, None, 0, 0, None, None, 0, 0, None, None)),
                       ("kj_ssgi_render", (None, None, None, None, None, None)), ("kj_shadow_denoise_render", (None, None, None, None, None, None)),
                       ("kj_baked_mesh_view", (None, 0, None)), ("kj_baked_image_view", (None, 0, None)), ("kj_baked_image_decode_rgba8", (37, None, 0, 4, 4, None)),
                       ("kj_post_create", (None, None, None)), ("kj_post_render", (None, None, 0, 0, 0, 1.0, 1.0, None, None)), ("kj_post_surface", (None, None, None, None)),
                       ("kj_post_read_back_histogram", (None, 0.0, 0.0, None, None)), ("kj_luminance_histogram_mean_log2", (None, 0.0, 0.0, None)), ("kj_post_mip_levels", (None, None)),
                       ("kj_motion_blur_create", (None, None)), ("kj_motion_blur_render", (None, None, 0, 0, None, None, 0, 0, None, None)), ("kj_motion_blur_surface", (None, None, None, None))):
        assert getattr(L, name)(*args) == 1, name
        assert L.kj_last_error(), name
    # destroy(NULL) is a no-op, like dropping a None
    for name in ("kj_rtr_destroy", "kj_rtdgi_destroy", "kj_scene_destroy", "kj_ircache_destroy", "kj_taa_destroy", "kj_ssgi_destroy", "kj_shadow_denoise_destroy", "kj_reprojection_destroy", "kj_post_destroy", "kj_motion_blur_destroy"):
        getattr(L, name)(None)


def test_frame_constants_builder():
    import numpy as np
    from kajiya_amd import frame
    fs = frame.FrameState((1920, 1080))
    cam = frame.CameraMatrices((1.0, 2.0, 3.0), np.eye(3), 52.0, 1920 / 1080)
    fc = fs.prepare_frame_constants(cam)
    v2c = np.array(fc.view_constants.view_to_clip[:]).reshape(4, 4).T
    c2v = np.array(fc.view_constants.clip_to_view[:]).reshape(4, 4).T
    assert np.allclose(v2c @ c2v, np.eye(4), atol=1e-5)
    assert abs(c2v[3, 2] - 1.0 / 0.01) < 1e-3                      # clip_to_view._43 = 1/znear (camera.rs:112-117)
    w2v = np.array(fc.view_constants.world_to_view[:]).reshape(4, 4).T
    v2w = np.array(fc.view_constants.view_to_world[:]).reshape(4, 4).T
    assert np.allclose(w2v @ v2w, np.eye(4), atol=1e-5)
    # jitter = halton(2,3)[0] - 0.5 = (0, -1/6)  (world_renderer.rs:425-428)
    assert abs(fc.view_constants.sample_offset_pixels[0] - 0.0) < 1e-7
    assert abs(fc.view_constants.sample_offset_pixels[1] - (1.0 / 3.0 - 0.5)) < 1e-6
    # first frame: prev == current => clip_to_prev_clip == identity
    c2p = np.array(fc.view_constants.clip_to_prev_clip[:]).reshape(4, 4).T
    assert np.allclose(c2p, np.eye(4), atol=1e-4)
