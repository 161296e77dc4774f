"""ctypes binding of libkajiya_amd.so (the C-ABI in include/kajiya_amd.h) and a thin
frame driver that mirrors the in-scope part of `prepare_render_graph_standard`
(crates/lib/kajiya/src/world_render_passes.rs:13-292).

The HIP library is the only compute path: if it cannot be loaded this module raises —
there is no CPU fallback. torch is used only for device memory and streams.
"""
import ctypes as C
import os
import numpy as np

from .abi import (KjFrameConstants, KjMeshDesc, KjGbufferDepth, KjRtdgiRenderParams, KjRtdgiOutput, KjTaaOutput, KJ_RTDGI_PASS, KjRtrTables, KjRtrParams, KjSplitRank, KjSplitFrame)
from . import scenes as kscenes

HERE = os.path.dirname(os.path.abspath(__file__))
# KJ_MEASURE_SKIP=taa,irc_rays (measurement only, behind KJ_DEBUG_ENV=1 like the library's switches: what a piece of the PIPELINED frame costs at the margin -- the frame's images are
# wrong without it): frame_pipelined leaves out TAA / the cache's ray launch
_MEASURE_SKIP = set(filter(None, os.environ.get("KJ_MEASURE_SKIP", "").split(","))) if os.environ.get("KJ_DEBUG_ENV") == "1" else set()
# where the pipelined frame runs the cache's SH sum-up (IrcacheRenderState::sum_up_irradiance_for_sampling): behind the cache's rays on the cache stream (1) or on the main stream behind
# its wait for them (0, rounds 1-5)
_SUMUP_ON_CACHE_STREAM = os.environ.get("KJ_SUMUP_ON_CACHE_STREAM", "1") != "0"
# the pipelined frame's half-res extract on the SSAO guide's stream beside `rtdgi reproject` (1: -1.3 % per 1080p frame, round 6) or behind it on the main stream (0)
_EXTRACT_BESIDE_REPROJECT = os.environ.get("KJ_EXTRACT_BESIDE_REPROJECT", "1") != "0"
LIB_PATH = os.environ.get("KJ_AMD_LIB") or os.path.join(HERE, "libkajiya_amd.so")   # KJ_AMD_LIB: A/B a differently built library

EXPORTS = [
    "kj_abi_struct_size", "kj_selftest_div_sqrt_nr", "kj_selftest_probe_functions", "kj_selftest_probe_functions_color", "kj_selftest_probe_functions_shading", "kj_selftest_probe_functions_misc",
    "kj_last_error", "kj_abi_version", "kj_device_create", "kj_device_destroy", "kj_device_brdf_lut",
    "kj_scene_create", "kj_scene_destroy", "kj_scene_add_mesh", "kj_scene_add_instance", "kj_scene_set_instance_transform",
    "kj_scene_set_instance_emissive_multiplier", "kj_scene_remove_instance", "kj_scene_commit", "kj_scene_triangle_light_count",
    "kj_scene_stats", "kj_scene_last_commit_ms", "kj_scene_set_blas_build_mode", "kj_scene_set_open_instances", "kj_scene_set_top_build_mode", "kj_scene_top_tree_info", "kj_frame_begin", "kj_trace_closest", "kj_trace_any", "kj_debug_calibration_copy", "kj_raster_gbuffer", "kj_sky_cube_render",
    "kj_sky_cube_convolve", "kj_reprojection_create", "kj_reprojection_destroy", "kj_calculate_reprojection_map",
    "kj_rtdgi_create", "kj_rtdgi_destroy", "kj_rtdgi_set_options", "kj_rtdgi_reproject", "kj_rtdgi_reproject_rows", "kj_rtdgi_render",
    "kj_rtdgi_surface", "kj_rtdgi_ray_counts", "kj_rtdgi_set_profiling", "kj_rtdgi_set_ray_pass_form", "kj_rtdgi_set_pool_tune", "kj_rtdgi_pass_times_ms", "kj_rtdgi_traversal_counts",
    "kj_ircache_create", "kj_ircache_destroy", "kj_ircache_update_eye_position", "kj_ircache_constants", "kj_ircache_set_enable_scroll",
    "kj_ircache_prepare", "kj_ircache_trace_irradiance", "kj_ircache_sum_up_irradiance_for_sampling", "kj_ircache_buffer", "kj_ircache_ray_counts", "kj_ircache_set_deferred_updates", "kj_ircache_set_ray_passes_side_by_side", "kj_ircache_set_ray_pass_schedule", "kj_ircache_begin_requests", "kj_ircache_begin_requests_rows", "kj_ircache_request_ranges", "kj_ircache_summary_bytes", "kj_ircache_summarize_requests", "kj_ircache_summary", "kj_ircache_apply_summaries", "kj_ircache_apply_requests", "kj_ircache_set_rtr_requests", "kj_ircache_rtr_request_ranges",
    "kj_taa_create", "kj_taa_destroy", "kj_taa_render", "kj_taa_render_rows", "kj_taa_surface", "kj_reference_path_trace",
    "kj_ssgi_create", "kj_ssgi_destroy", "kj_ssgi_render", "kj_ssgi_render_rows", "kj_ssgi_surface", "kj_trace_sun_shadow_mask", "kj_trace_sun_shadow_mask_rows", "kj_light_gbuffer", "kj_light_gbuffer_rows",
    "kj_shadow_denoise_create", "kj_shadow_denoise_destroy", "kj_shadow_denoise_render", "kj_shadow_denoise_render_rows", "kj_shadow_denoise_surface",
    "kj_baked_mesh_view", "kj_baked_image_view", "kj_baked_image_mip", "kj_baked_image_decode_rgba8",
    "kj_rtr_create", "kj_rtr_destroy", "kj_rtr_set_options", "kj_rtr_trace", "kj_rtr_render_specular_lights", "kj_rtr_filter_temporal", "kj_rtr_render_rows", "kj_rtr_surface", "kj_rtr_ray_counts",
    "kj_post_create", "kj_post_destroy", "kj_post_render", "kj_post_read_back_histogram", "kj_luminance_histogram_mean_log2", "kj_post_surface", "kj_post_mip_levels",
    "kj_motion_blur_create", "kj_motion_blur_destroy", "kj_motion_blur_render", "kj_motion_blur_surface",
    "kj_split_create", "kj_split_destroy", "kj_split_strip", "kj_split_gi_frame", "kj_split_merge_ircache", "kj_split_taa_frame", "kj_split_taa_frame_on", "kj_split_ssgi_frame", "kj_split_gather", "kj_split_self_test", "kj_split_shadow_frame", "kj_split_set_rtr", "kj_split_rtr_frame",
    "kj_split_set_profiling", "kj_split_profile", "kj_split_rccl_unique_id", "kj_split_rccl_comm_create", "kj_split_rccl_comm_info", "kj_split_rccl_comm_destroy",
]

_LIB = None


class KjError(RuntimeError):
    pass


def load():
    """Load the HIP library. Fails loudly when it is missing (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise KjError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(kajiya_amd has no CPU fallback)")
    try:      # torch brings its own HIP runtime: it has to be in the process first so that this library binds to the same one (loaded the
        import torch  # noqa: F401  # other way round, the second runtime finds no device: `hipGetDeviceCount -> no ROCm-capable device`)
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    L.kj_last_error.restype = C.c_char_p
    L.kj_abi_version.restype = u32
    L.kj_ircache_summary_bytes.argtypes = []
    L.kj_ircache_summary_bytes.restype = C.c_uint64
    sig = {
        "kj_device_create": [i32, vp, C.POINTER(vp)],
        "kj_device_brdf_lut": [vp, C.POINTER(vp)],
        "kj_scene_create": [vp, C.POINTER(vp)],
        "kj_scene_add_mesh": [vp, C.POINTER(KjMeshDesc), C.POINTER(u32)],
        "kj_scene_add_instance": [vp, u32, vp, C.POINTER(u32)],
        "kj_scene_set_instance_transform": [vp, u32, vp],
        "kj_scene_set_instance_emissive_multiplier": [vp, u32, C.c_float],
        "kj_scene_remove_instance": [vp, u32],
        "kj_scene_commit": [vp, vp],
        "kj_scene_triangle_light_count": [vp, C.POINTER(u32)],
        "kj_scene_stats": [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)],
        "kj_frame_begin": [vp, C.POINTER(KjFrameConstants), vp],
        "kj_trace_closest": [vp, vp, vp, u32, u32, vp],
        "kj_trace_any": [vp, vp, vp, u32, vp],
        "kj_debug_calibration_copy": [vp, vp, C.c_uint64, vp],
        "kj_scene_last_commit_ms": [vp, C.POINTER(C.c_double)],
        "kj_ircache_set_deferred_updates": [vp, u32], "kj_ircache_set_ray_passes_side_by_side": [vp, u32], "kj_ircache_set_ray_pass_schedule": [vp, u32],
        "kj_ircache_begin_requests": [vp, u32, u32, vp],
        "kj_ircache_begin_requests_rows": [vp, u32, u32, u32, u32, vp],
        "kj_ircache_request_ranges": [vp, C.POINTER(u32), C.POINTER(u32)],
        "kj_ircache_summarize_requests": [vp, C.POINTER(u32), C.POINTER(u32), u32, u32, vp],
        "kj_ircache_summary": [vp, u32, C.POINTER(vp)],
        "kj_ircache_apply_summaries": [vp, C.POINTER(vp), u32, vp],
        "kj_ircache_apply_requests": [vp, vp, u32, vp],
        "kj_ircache_set_rtr_requests": [vp, u32],
        "kj_ircache_rtr_request_ranges": [vp, C.POINTER(u32), C.POINTER(u32)],
        "kj_scene_set_blas_build_mode": [vp, u32],
        "kj_scene_set_open_instances": [vp, u32],
        "kj_scene_set_top_build_mode": [vp, u32],
        "kj_scene_top_tree_info": [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)],
        "kj_raster_gbuffer": [vp, vp, u32, u32, vp, vp, vp, vp, vp],
        "kj_sky_cube_render": [vp, vp, vp],
        "kj_sky_cube_convolve": [vp, vp, vp, vp],
        "kj_reprojection_create": [vp, C.POINTER(vp)],
        "kj_calculate_reprojection_map": [vp, C.POINTER(KjGbufferDepth), vp, C.POINTER(vp), vp],
        "kj_rtdgi_create": [vp, C.POINTER(vp)],
        "kj_rtdgi_set_options": [vp, u32, u32],
        "kj_rtdgi_reproject": [vp, vp, u32, u32, vp],
        "kj_rtdgi_reproject_rows": [vp, vp, u32, u32, u32, u32, vp],
        "kj_rtdgi_render": [vp, C.POINTER(KjRtdgiRenderParams), C.POINTER(KjRtdgiOutput), vp],
        "kj_rtdgi_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_rtdgi_ray_counts": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "kj_rtdgi_set_profiling": [vp, u32, u32],
        "kj_rtdgi_set_ray_pass_form": [vp, u32],
        "kj_rtdgi_set_pool_tune": [vp, u32, u32, u32, u32, u32],
        "kj_rtdgi_pass_times_ms": [vp, C.POINTER(C.c_float), u32],
        "kj_rtdgi_traversal_counts": [vp, C.POINTER(C.c_uint64)],
        "kj_ircache_create": [vp, C.POINTER(vp)],
        "kj_ircache_update_eye_position": [vp, vp],
        "kj_ircache_constants": [vp, C.POINTER(KjFrameConstants)],
        "kj_ircache_set_enable_scroll": [vp, u32],
        "kj_ircache_prepare": [vp, vp],
        "kj_ircache_trace_irradiance": [vp, vp, vp, u32, vp],
        "kj_ircache_sum_up_irradiance_for_sampling": [vp, vp],
        "kj_ircache_buffer": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_ircache_ray_counts": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "kj_taa_create": [vp, C.POINTER(vp)],
        "kj_taa_render": [vp, vp, u32, u32, vp, vp, u32, u32, C.POINTER(KjTaaOutput), vp],
        "kj_taa_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_light_gbuffer": [vp, C.POINTER(KjGbufferDepth), vp, u32, vp, vp, vp, u32, vp, vp, u32, vp],
        "kj_light_gbuffer_rows": [vp, C.POINTER(KjGbufferDepth), vp, u32, vp, vp, vp, u32, vp, vp, u32, u32, u32, vp],
        "kj_shadow_denoise_create": [vp, C.POINTER(vp)],
        "kj_shadow_denoise_render": [vp, C.POINTER(KjGbufferDepth), vp, vp, C.POINTER(vp), vp],
        "kj_shadow_denoise_render_rows": [vp, C.POINTER(KjGbufferDepth), vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp), vp],
        "kj_shadow_denoise_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_rtr_create": [vp, C.POINTER(KjRtrTables), C.POINTER(vp)],
        "kj_rtr_set_options": [vp, u32],
        "kj_rtr_trace": [vp, C.POINTER(KjRtrParams), vp],
        "kj_rtr_render_specular_lights": [vp, C.POINTER(KjRtrParams), vp],
        "kj_rtr_filter_temporal": [vp, C.POINTER(KjRtrParams), C.POINTER(vp), vp],
        "kj_rtr_render_rows": [vp, C.POINTER(KjRtrParams), u32, u32, C.POINTER(vp), vp],
        "kj_rtr_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_rtr_ray_counts": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "kj_trace_sun_shadow_mask": [vp, vp, C.POINTER(KjGbufferDepth), vp, vp, vp],
        "kj_trace_sun_shadow_mask_rows": [vp, vp, C.POINTER(KjGbufferDepth), vp, C.c_uint32, C.c_uint32, vp, vp],
        "kj_post_create": [vp, vp, C.POINTER(vp)],
        "kj_post_render": [vp, vp, u32, u32, u32, C.c_float, C.c_float, C.POINTER(vp), vp],
        "kj_post_read_back_histogram": [vp, C.c_float, C.c_float, C.POINTER(C.c_float), vp],
        "kj_luminance_histogram_mean_log2": [vp, C.c_float, C.c_float, C.POINTER(C.c_float)],
        "kj_post_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_post_mip_levels": [vp, C.POINTER(u32)],
        "kj_motion_blur_create": [vp, C.POINTER(vp)],
        "kj_motion_blur_render": [vp, vp, u32, u32, vp, vp, u32, u32, C.POINTER(vp), vp],
        "kj_motion_blur_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_ssgi_create": [vp, C.POINTER(vp)],
        "kj_ssgi_render": [vp, C.POINTER(KjGbufferDepth), vp, vp, C.POINTER(vp), vp],
        "kj_ssgi_render_rows": [vp, C.POINTER(KjGbufferDepth), vp, vp, u32, u32, C.POINTER(vp), vp],
        "kj_ssgi_surface": [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_uint64)],
        "kj_reference_path_trace": [vp, vp, vp, u32, u32, u32, u32, u32, vp, vp],
        "kj_taa_render_rows": [vp, vp, u32, u32, vp, vp, u32, u32, C.POINTER(KjTaaOutput), vp, u32, u32, u32],
        "kj_split_create": [u32, u32, u32, C.POINTER(KjSplitRank), u32, u32, u32, vp, C.POINTER(vp)],
        "kj_split_strip": [vp, u32, C.POINTER(u32), C.POINTER(u32)],
        "kj_split_gi_frame": [vp, C.POINTER(KjSplitFrame), u32, vp, vp],
        "kj_split_taa_frame": [vp, C.POINTER(KjSplitFrame), vp],
        "kj_split_taa_frame_on": [vp, C.POINTER(KjSplitFrame), C.POINTER(vp), vp],
        "kj_split_ssgi_frame": [vp, C.POINTER(vp), C.POINTER(KjSplitFrame), C.POINTER(vp), vp],
        "kj_split_merge_ircache": [vp, vp],
        "kj_split_gather": [vp, C.c_char_p, vp],
        "kj_split_self_test": [vp, C.POINTER(C.c_uint32), vp],
        "kj_split_shadow_frame": [vp, C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp],
        "kj_split_set_rtr": [vp, u32],
        "kj_split_rtr_frame": [vp, C.POINTER(vp), C.POINTER(KjRtrParams), u32, vp, C.POINTER(vp), vp],
        "kj_split_set_profiling": [vp, u32],
        "kj_split_profile": [vp, vp],
        "kj_split_rccl_unique_id": [vp],
        "kj_split_rccl_comm_create": [vp, u32, u32, C.POINTER(vp)],
        "kj_split_rccl_comm_info": [vp, C.POINTER(u32), C.POINTER(u32)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = i32
    for name in ("kj_device_destroy", "kj_scene_destroy", "kj_reprojection_destroy", "kj_rtdgi_destroy", "kj_ircache_destroy", "kj_taa_destroy", "kj_ssgi_destroy", "kj_shadow_denoise_destroy", "kj_rtr_destroy", "kj_post_destroy", "kj_motion_blur_destroy", "kj_split_destroy", "kj_split_rccl_comm_destroy"):
        f = getattr(L, name)
        f.argtypes = [vp]
        f.restype = None
    _LIB = L
    return L


def check(status):
    if status != 0:
        raise KjError(f"kajiya_amd status {status}: {load().kj_last_error().decode()}")


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Device:
    def __init__(self, ordinal=0, blue_noise=None):
        L = load()
        if blue_noise is None:
            blue_noise = np.fromfile(os.path.join(kscenes.GOLDEN_DIR, "bluenoise_256_rgba8.bin"), dtype=np.uint8)
        self._bn = np.ascontiguousarray(blue_noise, np.uint8)
        assert self._bn.size == 256 * 256 * 4
        self.h = C.c_void_p()
        check(L.kj_device_create(ordinal, self._bn.ctypes.data, C.byref(self.h)))

    def frame_begin(self, fc):
        check(load().kj_frame_begin(self.h, C.byref(fc), _stream_ptr()))
        self.clip_to_view_11 = float(fc.view_constants.clip_to_view[5])      # tan(vertical fov / 2): the split sizes rtr's resolve halo from it

    def brdf_lut_ptr(self):
        p = C.c_void_p()
        check(load().kj_device_brdf_lut(self.h, C.byref(p)))
        return p.value

    def __del__(self):
        try:
            load().kj_device_destroy(self.h)
        except Exception:
            pass


class Scene:
    """WorldRenderer scene state: add_mesh / add_instance / commit (builds the software LBVH)."""

    def __init__(self, dev: Device, desc: kscenes.SceneDesc = None, use_lights=False, fast_build=False, open_instances=False, top_build=None):
        L = load()
        self.dev = dev
        self.h = C.c_void_p()
        check(L.kj_scene_create(dev.h, C.byref(self.h)))
        if top_build is not None:   # who builds the per-commit top tree: "host" / 1, "device" / 2 (default: the host below 1024 leaves, the device from there on)
            check(L.kj_scene_set_top_build_mode(self.h, {"auto": 0, "host": 1, "device": 2}.get(top_build, top_build)))
        if open_instances:  # top-tree leaves = nodes of the instances' top levels instead of whole instances
            check(L.kj_scene_set_open_instances(self.h, 1))
        if fast_build:      # BLASes built on the device instead of SAH trees built on the host: True / 1 = LBVH, "ploc" / 2 = PLOC
            check(L.kj_scene_set_blas_build_mode(self.h, 2 if fast_build in ("ploc", 2) and fast_build is not True else 1))
        self._keep = []
        if desc is not None:
            for m in desc.meshes:
                self.add_mesh(m, use_lights)
            for mi, xf in desc.instances:
                self.add_instance(mi, xf)
            self.commit()

    def add_mesh(self, mesh: kscenes.TriangleMesh, use_lights=False):
        d, keep = mesh.pack(use_lights)
        self._keep.append(keep)
        out = C.c_uint32()
        check(load().kj_scene_add_mesh(self.h, C.byref(d), C.byref(out)))
        return out.value

    def add_instance(self, mesh_idx, xform3x4):
        xf = np.ascontiguousarray(xform3x4, np.float32)
        out = C.c_uint32()
        check(load().kj_scene_add_instance(self.h, mesh_idx, xf.ctypes.data, C.byref(out)))
        return out.value

    def commit(self):
        check(load().kj_scene_commit(self.h, _stream_ptr()))

    def set_instance_transform(self, instance, xform3x4):
        xf = np.ascontiguousarray(xform3x4, np.float32)
        check(load().kj_scene_set_instance_transform(self.h, instance, xf.ctypes.data))

    def top_tree_info(self):
        """{nodes, capacity, device}: the last commit's top tree and who built it."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(load().kj_scene_top_tree_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"nodes": a.value, "capacity": b.value, "device": bool(c.value)}

    def last_commit_ms(self):
        """[BLAS builds, instance records + TLAS, uploads + device transform, total] of the last commit, host ms."""
        out = (C.c_double * 4)()
        check(load().kj_scene_last_commit_ms(self.h, out))
        return list(out)

    def stats(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(load().kj_scene_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(triangles=a.value, nodes=b.value, bvh_bytes=c.value)

    @property
    def triangle_light_count(self):
        out = C.c_uint32()
        check(load().kj_scene_triangle_light_count(self.h, C.byref(out)))
        return out.value

    def trace_closest(self, rays_dev, count, cull_back=False):
        import torch
        hits = torch.empty((count, 4), dtype=torch.float32, device=rays_dev.device)
        check(load().kj_trace_closest(self.h, rays_dev.data_ptr(), hits.data_ptr(), count, int(cull_back), _stream_ptr()))
        return hits

    def trace_any(self, rays_dev, count):
        import torch
        out = torch.empty((count,), dtype=torch.uint8, device=rays_dev.device)
        check(load().kj_trace_any(self.h, rays_dev.data_ptr(), out.data_ptr(), count, _stream_ptr()))
        return out

    def __del__(self):
        try:
            load().kj_scene_destroy(self.h)
        except Exception:
            pass


class GpuPipeline:
    """One frame of the hot path on the GPU: sky cubes -> G-buffer stand-in -> reprojection map ->
    RtdgiRenderer::reproject -> RtdgiRenderer::render. All buffers stay resident in HBM."""

    def __init__(self, dev: Device, scene: Scene, width, height, device="cuda:0", use_ircache=False):
        import torch
        self.torch = torch
        L = load()
        self.L, self.dev, self.scene = L, dev, scene
        self.W, self.H = width, height
        W, H = width, height
        t = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.geometric_normal = t((H, W), torch.int32)
        self.gbuffer = t((H, W, 4), torch.int32)
        self.depth = t((H, W), torch.float32)
        self.velocity = t((H, W, 4), torch.int16)
        self.ssao = torch.full((H, W), 255, dtype=torch.uint8, device=device)
        self.sky64 = t((6, 64, 64, 4), torch.int16)
        self.sky16 = t((6, 16, 16, 4), torch.int16)
        self._sky_key = None
        self.reproj = C.c_void_p()
        check(L.kj_reprojection_create(dev.h, C.byref(self.reproj)))
        self.rtdgi = C.c_void_p()
        check(L.kj_rtdgi_create(dev.h, C.byref(self.rtdgi)))
        self.reprojection_map_ptr = C.c_void_p()
        self.out = KjRtdgiOutput()
        self.taa = C.c_void_p()
        check(L.kj_taa_create(dev.h, C.byref(self.taa)))
        self.taa_out = KjTaaOutput()
        self.on_ircache_traced = None
        self.ssgi = None
        self.ssao_ptr = C.c_void_p()
        self.ircache = None
        if use_ircache:
            self.ircache = C.c_void_p()
            check(L.kj_ircache_create(dev.h, C.byref(self.ircache)))

    def gbuffer_depth(self):
        g = KjGbufferDepth()
        g.geometric_normal = self.geometric_normal.data_ptr()
        g.gbuffer = self.gbuffer.data_ptr()
        g.depth = self.depth.data_ptr()
        g.width, g.height = self.W, self.H
        return g

    def render_inputs(self, fc):
        L, s = self.L, _stream_ptr()
        self.dev.frame_begin(fc)
        key = bytes(fc.sun_direction) + bytes(fc.sun_color_multiplier) + bytes(fc.sky_ambient) + bytes(C.c_float(fc.pre_exposure))
        if key != self._sky_key:
            check(L.kj_sky_cube_render(self.dev.h, self.sky64.data_ptr(), s))
            check(L.kj_sky_cube_convolve(self.dev.h, self.sky64.data_ptr(), self.sky16.data_ptr(), s))
            self._sky_key = key
        check(L.kj_raster_gbuffer(self.dev.h, self.scene.h, self.W, self.H, self.geometric_normal.data_ptr(), self.gbuffer.data_ptr(),
                                  self.depth.data_ptr(), self.velocity.data_ptr(), s))

    def reprojection(self):
        g = self.gbuffer_depth()
        check(self.L.kj_calculate_reprojection_map(self.reproj, C.byref(g), self.velocity.data_ptr(), C.byref(self.reprojection_map_ptr), _stream_ptr()))

    def params(self, pass_mask=KJ_RTDGI_PASS["ALL"]):
        p = KjRtdgiRenderParams()
        p.gbuffer_depth = self.gbuffer_depth()
        p.reprojection_map = self.reprojection_map_ptr
        p.sky_cube = self.sky16.data_ptr()
        p.sky_cube_width = 16
        p.scene = self.scene.h
        p.ircache = self.ircache
        p.ssao_tex = self.ssao_ptr.value if self.ssao_ptr.value else self.ssao.data_ptr()   # ssgi_frame() output, else the constant 1.0
        p.pass_mask = pass_mask
        return p

    def rtdgi_frame(self, pass_mask=KJ_RTDGI_PASS["ALL"]):
        """The GI frame proper (what `gi_frame_ms` times): rtdgi.reproject + rtdgi.render."""
        s = _stream_ptr()
        check(self.L.kj_rtdgi_reproject(self.rtdgi, self.reprojection_map_ptr, self.W, self.H, s))
        p = self.params(pass_mask)
        check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))

    def gi_frame(self, pass_mask=KJ_RTDGI_PASS["ALL"], defer_replay=False):
        """The GI frame in world_render_passes.rs order: ircache.prepare, trace_irradiance (:99,113-121),
        rtdgi.reproject (:129), ircache sum-up (:138-140), rtdgi.render (:145-163). `defer_replay` (deferred cache updates): the caller
        replays the frame's records itself (ircache_replay_own_requests) -- after rtr_frame, whose rays look the cache up as well."""
        s = _stream_ptr()
        deferred = self.ircache and getattr(self, "ircache_deferred", False)
        if self.ircache:
            if deferred:
                self.ircache_begin_requests()
            check(self.L.kj_ircache_prepare(self.ircache, s))
            check(self.L.kj_ircache_trace_irradiance(self.ircache, self.scene.h, self.sky16.data_ptr(), 16, s))
        check(self.L.kj_rtdgi_reproject(self.rtdgi, self.reprojection_map_ptr, self.W, self.H, s))
        if self.ircache:
            check(self.L.kj_ircache_sum_up_irradiance_for_sampling(self.ircache, s))
        p = self.params(pass_mask)
        check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
        if deferred and not defer_replay:
            self.ircache_replay_own_requests()

    # ---- frame pipelining (async compute). The irradiance cache's maintenance + ray kernels of frame N+1 only depend on
    # the cache state left by frame N's rtdgi validate/trace passes and on frame N+1's constants, and they are latency-bound
    # (a few hundred waves walking the BVH). They are therefore issued on a second HIP stream as soon as frame N's trace pass
    # is done and overlap frame N's screen-space tail (ReSTIR resampling, resolve, denoise, TAA). Same dependencies as the
    # serial order of world_render_passes.rs:99-163, hence the same results (up to the cache's own atomics races).
    def pipeline_begin(self, fc):
        """Prime the pipeline: upload `fc` and issue the ircache work of the first frame on the side stream."""
        torch = self.torch
        assert self.ircache, "pipelining overlaps the irradiance cache; nothing to do without it"
        self._s1 = torch.cuda.Stream(priority=int(os.environ.get("KJ_PRIO_IRC", "0")))      # stream priorities: A/B knobs (profiles/r04_stream_priorities.md)
        self._ev_irc = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_fc = [torch.cuda.Event(), torch.cuda.Event()]      # frame constants of frame i are in place (k_frame_begin ran on the side stream)
        self._ev_trace = [torch.cuda.Event(), torch.cuda.Event()]
        self._pipe_i = 0
        self._enqueue_ircache(fc, None)

    def _enqueue_ircache(self, fc, wait_event):
        torch = self.torch
        s0 = torch.cuda.current_stream()
        with torch.cuda.stream(self._s1):
            self._s1.wait_stream(s0) if wait_event is None else self._s1.wait_event(wait_event)
            self.dev.frame_begin(fc)
            self._ev_fc[self._pipe_i & 1].record(self._s1)
            s = _stream_ptr()
            check(self.L.kj_ircache_prepare(self.ircache, s))
            if "irc_rays" not in _MEASURE_SKIP:
                check(self.L.kj_ircache_trace_irradiance(self.ircache, self.scene.h, self.sky16.data_ptr(), 16, s))
            if self.on_ircache_traced is not None:
                self.on_ircache_traced()          # e.g. the bench logs the cache's ray counters, stream-ordered
            # the SH sum-up behind the rays on THIS stream (KJ_SUMUP_ON_CACHE_STREAM, see frame_pipelined): 0.03 ms off the main stream's chain between two trace passes
            if _SUMUP_ON_CACHE_STREAM and "irc_rays" not in _MEASURE_SKIP:
                check(self.L.kj_ircache_sum_up_irradiance_for_sampling(self.ircache, s))
            self._ev_irc[self._pipe_i & 1].record(self._s1)

    def frame_pipelined(self, next_fc, run_ssgi=False):
        """One GI + TAA frame whose ircache work was issued earlier; then issue the next frame's (if any). The caller binds this
        frame's G-buffer inputs before the call. `next_fc` = constants of the following frame or None for the last one.
        `run_ssgi`: compute the SSAO guide first (SsgiRenderer::render) -- it reads this frame's constants, which the SIDE stream
        wrote, so it is ordered behind that write here rather than issued by the caller.

        Three streams: the main one carries ssgi/rtdgi up to the temporal filter; the ircache stream runs the next frame's cache
        maintenance + rays from the moment this frame's trace pass is recorded; the third stream runs this frame's spatial
        filter + TAA (VALU-bound) under the next frame's ray passes (latency-bound: at 1080p they expose ~3 waves per SIMD).
        Hazards: the next frame's temporal filter may not overwrite `temporal_filtered_tex` before this spatial filter has read
        it (event below); the SSAO guide is double-buffered inside KjSsgi; everything else on the third stream is its own."""
        torch = self.torch
        s0 = torch.cuda.current_stream()
        i = self._pipe_i & 1
        if not hasattr(self, "_s2"):
            self._s2 = torch.cuda.Stream(priority=int(os.environ.get("KJ_PRIO_TAA", "0")))
            self._ev_gi = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev_taa = [torch.cuda.Event(), torch.cuda.Event()]
        # everything that does not read the cache goes first: the wait for the cache stream sits in the frame-to-frame critical
        # cycle (this frame's ray passes -> next frame's cache rays -> next frame's ray passes); measured -1.6 % per frame
        s0.wait_event(self._ev_fc[i])
        # The SSAO guide (VALU-bound, ~0.1 ms at 1080p) on a stream of its own UNDER the ray passes (latency-bound): nothing before
        # `restir spatial` reads it, so the half-res extract leaves the SSAO out here and adds it behind `restir temporal`
        # (KJ_RTDGI_PASS_EXTRACT_HALF_NO_SSAO / _SSAO_ONLY). Measured on MI355X (round 4, A/B/A/B in one lease): 1080p 1.002-1.009 ms against
        # 1.018-1.023 ms with the guide first on the main stream (-1.5 %); at 4K the ray passes are issue-bound (VALUBusy 82 %) and the guide
        # under them costs 2 % (3.55 vs 3.48 ms). Hence: on up to 1080p-sized frames, off above; KJ_SSGI_OVERLAP=0 / 1 forces it.
        overlap_ssgi = run_ssgi and os.environ.get("KJ_SSGI_OVERLAP", "1" if self.W * self.H <= 1920 * 1200 else "0") != "0"
        if run_ssgi and not overlap_ssgi:
            self.ssgi_frame()
        if overlap_ssgi:
            if not hasattr(self, "_s3"):
                self._s3 = torch.cuda.Stream(priority=int(os.environ.get("KJ_PRIO_SSGI", "0")))
                self._ev_ssgi = [torch.cuda.Event(), torch.cuda.Event()]
                self._ev_extract = [torch.cuda.Event(), torch.cuda.Event()]
            with torch.cuda.stream(self._s3):
                self._s3.wait_stream(s0)                         # the caller's work on the current stream up to here: this frame's G-buffer, depth and reprojection map
                self._s3.wait_event(self._ev_fc[i])
                if self._pipe_i > 0:
                    self._s3.wait_event(self._ev_gi[1 - i])      # last frame's resolve / filters have read the guide image this call overwrites... (double-buffered: two frames back)
                if _EXTRACT_BESIDE_REPROJECT:
                    # the half-res extract (G-buffer inputs only) at the head of THIS stream, beside the main stream's `rtdgi reproject` instead of behind it: everything of last frame
                    # that reads the half-res images is on the main stream ahead of the wait_stream above
                    P = KJ_RTDGI_PASS
                    check(self.L.kj_rtdgi_reproject(self.rtdgi, self.reprojection_map_ptr, self.W, self.H, C.c_void_p(s0.cuda_stream)))      # (host side first: the frame's first rtdgi call; its launch goes to the main stream)
                    p = self.params(P["EXTRACT_HALF"] | P["EXTRACT_HALF_NO_SSAO"])
                    check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), _stream_ptr()))
                    self._ev_extract[i].record(self._s3)
                self.ssgi_frame()
                if _EXTRACT_BESIDE_REPROJECT:      # ... and the guide's half-res copy right behind the guide, on this stream (its readers, `restir spatial` onwards, wait for _ev_ssgi)
                    p = self.params(KJ_RTDGI_PASS["EXTRACT_HALF_SSAO_ONLY"] | (1 << 31))
                    check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), _stream_ptr()))
                self._ev_ssgi[i].record(self._s3)
        s = _stream_ptr()
        P = KJ_RTDGI_PASS
        if overlap_ssgi and _EXTRACT_BESIDE_REPROJECT:
            s0.wait_event(self._ev_extract[i])
        else:
            check(self.L.kj_rtdgi_reproject(self.rtdgi, self.reprojection_map_ptr, self.W, self.H, s))
            p = self.params(P["EXTRACT_HALF"] | (P["EXTRACT_HALF_NO_SSAO"] if overlap_ssgi else 0))
            check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
        s0.wait_event(self._ev_irc[i])
        if "irc_rays" not in _MEASURE_SKIP and not _SUMUP_ON_CACHE_STREAM:
            check(self.L.kj_ircache_sum_up_irradiance_for_sampling(self.ircache, s))
        head = P["EXTRACT_HALF"] | P["VALIDATE"] | P["TRACE"]
        p = self.params(P["VALIDATE"] | P["TRACE"] | (1 << 31))
        check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
        self._ev_trace[i].record(s0)
        if self._pipe_i > 0:
            s0.wait_event(self._ev_taa[1 - i])           # last frame's spatial filter has consumed temporal_filtered_tex
        if overlap_ssgi:
            p = self.params(P["VALIDITY_INTEGRATE"] | P["RESTIR_TEMPORAL"] | (1 << 31))
            check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
            s0.wait_event(self._ev_ssgi[i])
            if not _EXTRACT_BESIDE_REPROJECT:
                p = self.params(P["EXTRACT_HALF_SSAO_ONLY"] | (1 << 31))       # params() picks up this frame's guide pointer (ssgi_frame set it)
                check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
            p = self.params((P["ALL"] & ~head & ~P["SPATIAL_FILTER"] & ~P["VALIDITY_INTEGRATE"] & ~P["RESTIR_TEMPORAL"]) | (1 << 31))
        else:
            p = self.params((P["ALL"] & ~head & ~P["SPATIAL_FILTER"]) | (1 << 31))
        check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), s))
        self._ev_gi[i].record(s0)
        p = self.params(P["SPATIAL_FILTER"] | (1 << 31))      # captures this frame's inputs (guide N, depth N) now
        with torch.cuda.stream(self._s2):
            self._s2.wait_event(self._ev_gi[i])
            check(self.L.kj_rtdgi_render(self.rtdgi, C.byref(p), C.byref(self.out), _stream_ptr()))
            if "taa" not in _MEASURE_SKIP:
                self.taa_frame()
            self._ev_taa[i].record(self._s2)
        self._pipe_i += 1
        if next_fc is not None:
            self._enqueue_ircache(next_fc, self._ev_trace[i])
        else:
            s0.wait_event(self._ev_taa[i])

    # ---- deferred irradiance-cache updates (include/kajiya_amd.h: kj_ircache_set_deferred_updates): lookups record, the host replays
    def ircache_set_deferred(self, enable=True):
        check(self.L.kj_ircache_set_deferred_updates(self.ircache, int(enable)))
        self.ircache_deferred = bool(enable)

    IRC_PASS_SCHEDULES = {"sequential": 0, "side_by_side": 1, "chain": 2}

    def ircache_set_ray_pass_schedule(self, schedule):
        """How the cache's three ray passes are scheduled (include/kajiya_amd.h: KJ_IRC_PASSES_*); "chain" is the default."""
        check(self.L.kj_ircache_set_ray_pass_schedule(self.ircache, self.IRC_PASS_SCHEDULES[schedule]))

    def ircache_begin_requests(self, half_rows=None):
        """`half_rows`: (begin, end) -- this pipeline's per-pixel passes only run on those half-res rows (a rank of the split): only their slots are cleared."""
        hw, hh = (self.W + 1) // 2, (self.H + 1) // 2
        a, b = half_rows if half_rows is not None else (0, hh)
        check(self.L.kj_ircache_begin_requests_rows(self.ircache, hw, hh, a, b, _stream_ptr()))

    def ircache_request_ranges(self):
        first, count = (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
        check(self.L.kj_ircache_request_ranges(self.ircache, first, count))
        return list(first), list(count)

    def ircache_set_rtr_requests(self, enable=True):
        """Reflections bound to a cache in deferred mode record their lookups in two slot ranges of their own (set before ircache_begin_requests)."""
        check(self.L.kj_ircache_set_rtr_requests(self.ircache, int(enable)))

    def ircache_rtr_request_ranges(self):
        first, count = (C.c_uint32 * 2)(), (C.c_uint32 * 2)()
        check(self.L.kj_ircache_rtr_request_ranges(self.ircache, first, count))
        return list(first), list(count)

    def ircache_summarize(self, ranges, which=0):
        """Reduces the recorded lookups of the slot ranges [(first, count), ...] into the cache's fixed-size summary `which` (0: the part a rank of the
        split sends to the others; 1: the part that stays local) and returns it as a uint8 device tensor (include/kajiya_amd.h: kj_ircache_summarize_requests)."""
        ranges = [(f, c) for f, c in ranges if c]
        first = (C.c_uint32 * max(1, len(ranges)))(*[f for f, _ in ranges])
        count = (C.c_uint32 * max(1, len(ranges)))(*[c for _, c in ranges])
        check(self.L.kj_ircache_summarize_requests(self.ircache, first, count, len(ranges), which, _stream_ptr()))
        return self.ircache_summary(which)

    def ircache_summary(self, which=0):
        p = C.c_void_p()
        check(self.L.kj_ircache_summary(self.ircache, which, C.byref(p)))
        return tensor_from_ptr(p.value, int(self.L.kj_ircache_summary_bytes()), self.torch.uint8, (-1,))

    def ircache_apply_summaries(self, summaries):
        """The replay: merges the summaries (uint8 device tensors; the same ones in the same order on every replica) and applies the result."""
        ptrs = (C.c_void_p * len(summaries))(*[t.data_ptr() for t in summaries])
        check(self.L.kj_ircache_apply_summaries(self.ircache, ptrs, len(summaries), _stream_ptr()))

    def ircache_apply(self, requests, count):
        """A plain device list of 32-byte records replayed at once (kj_ircache_apply_requests)."""
        if count:
            check(self.L.kj_ircache_apply_requests(self.ircache, requests.data_ptr(), int(count), _stream_ptr()))

    def ircache_replay_own_requests(self):
        """Single GPU in deferred mode: everything this frame's lookups recorded, reduced and replayed."""
        first, count = self.ircache_request_ranges()
        f2, c2 = self.ircache_rtr_request_ranges()       # (empty unless the frame has reflections: ircache_set_rtr_requests)
        self.ircache_apply_summaries([self.ircache_summarize(list(zip(first + f2, count + c2)))])

    def taa_frame(self, input_ptr=None, out_extent=None):
        """TaaRenderer::render on `input_ptr` (default: this frame's rtdgi screen_irradiance_tex)."""
        ow, oh = out_extent or (self.W, self.H)
        inp = input_ptr if input_ptr is not None else self.out.screen_irradiance_tex
        check(self.L.kj_taa_render(self.taa, inp, self.W, self.H, self.reprojection_map_ptr, self.depth.data_ptr(), ow, oh, C.byref(self.taa_out), _stream_ptr()))

    def sun_shadow_mask(self, out=None, ray_counter=None, rows=None):
        """trace_sun_shadow_mask (renderers/shadows.rs:10-40): R8 mask, one soft-shadow ray per pixel. `rows` = (row_begin, row_end): only those
        rows (the screen-tile split, kj_trace_sun_shadow_mask_rows)."""
        if out is None:
            out = self.torch.zeros((self.H, self.W), dtype=self.torch.uint8, device=self.depth.device)
        g = self.gbuffer_depth()
        r0, r1 = rows if rows is not None else (0, self.H)
        check(self.L.kj_trace_sun_shadow_mask_rows(self.dev.h, self.scene.h, C.byref(g), out.data_ptr(), r0, r1, ray_counter.data_ptr() if ray_counter is not None else None, _stream_ptr()))
        return out

    def shadow_denoiser(self):
        if getattr(self, "shadow_dn", None) is None:
            self.shadow_dn = C.c_void_p()
            check(self.L.kj_shadow_denoise_create(self.dev.h, C.byref(self.shadow_dn)))
        return self.shadow_dn

    def shadow_denoise(self, shadow_mask, rows=None):
        """ShadowDenoiseRenderer::render (world_render_passes.rs:131-136): returns the RG16F image (H, W, 2) float16, x = shadow term. `rows`: the
        strip form (kj_shadow_denoise_render_rows): the result is valid on those rows only."""
        g = self.gbuffer_depth()
        out = C.c_void_p()
        r0, r1 = rows if rows is not None else (0, self.H)
        check(self.L.kj_shadow_denoise_render_rows(self.shadow_denoiser(), C.byref(g), shadow_mask.data_ptr(), self.reprojection_map_ptr, r0, r1, C.byref(out), _stream_ptr()))
        return tensor_from_ptr(out.value, self.W * self.H * 4, self.torch.float16, (self.H, self.W, 2))

    def rtr_params(self, pass_mask=63):
        """KjRtrParams for RtrRenderer::trace / filter_temporal (world_render_passes.rs:172-210): the unconvolved sky cube, this
        frame's rtdgi output and candidates (which the trace pass overwrites where the surface is smooth)."""
        p = KjRtrParams()
        p.gbuffer_depth = self.gbuffer_depth()
        p.reprojection_map = self.reprojection_map_ptr
        p.sky_cube = self.sky64.data_ptr()
        p.sky_cube_width = 64
        p.scene = self.scene.h
        p.ircache = self.ircache
        p.rtdgi_irradiance = self.out.screen_irradiance_tex
        p.candidate_radiance_tex = self.out.candidate_radiance_tex
        p.candidate_hit_tex = self.out.candidate_hit_tex
        p.candidate_normal_tex = self.out.candidate_normal_tex
        p.pass_mask = pass_mask
        return p

    def rtr_handle(self, tables=None):
        """The RtrRenderer of this pipeline, made on first use. `tables`: KjRtrTables (default: rtr_tables.standin_tables())."""
        if getattr(self, "rtr", None) is None:
            if tables is None:
                from . import rtr_tables
                tables, self._rtr_keep = rtr_tables.standin_tables()
            self.rtr = C.c_void_p()
            check(self.L.kj_rtr_create(self.dev.h, C.byref(tables), C.byref(self.rtr)))
        return self.rtr

    def rtr_frame(self, pass_mask=63, tables=None, specular_lights=False):
        """RtrRenderer::trace + TracedRtr::filter_temporal after rtdgi.render (same stream). Returns the resolved
        B10G11R11_UFLOAT image as an int32 (H, W) tensor view. `tables`: KjRtrTables (default: rtr_tables.standin_tables())."""
        self.rtr_handle(tables)
        p = self.rtr_params(pass_mask)
        s = _stream_ptr()
        out = C.c_void_p()
        if pass_mask & 15:
            check(self.L.kj_rtr_trace(self.rtr, C.byref(p), s))
        if specular_lights:
            check(self.L.kj_rtr_render_specular_lights(self.rtr, C.byref(p), s))     # LightingRenderer::render_specular (no-op without triangle lights)
        if pass_mask & 48:
            check(self.L.kj_rtr_filter_temporal(self.rtr, C.byref(p), C.byref(out), s))
        return self.rtr_surface("resolved_tex", self.torch.int32, (self.H, self.W))

    def rtr_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_rtr_surface(self.rtr, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def rtr_ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.L.kj_rtr_ray_counts(self.rtr, C.byref(a), C.byref(b)))
        return a.value, b.value

    def shadow_denoise_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_shadow_denoise_surface(self.shadow_dn, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def light_gbuffer(self, shadow_mask, rtdgi_ptr=None, rtr_ptr=None, debug_shading_mode=0, rows=None):
        """light_gbuffer (renderers/deferred.rs:6-60): returns (temporal_output, output) RGBA16F images. `shadow_mask`: uint8 (H, W)
        raw mask or float16 (H, W, 2) denoised image. `rows`: only those rows (kj_light_gbuffer_rows)."""
        t = self.torch
        if not hasattr(self, "_lit"):
            self._lit = (t.zeros((self.H, self.W, 4), dtype=t.float16, device=self.depth.device), t.zeros((self.H, self.W, 4), dtype=t.float16, device=self.depth.device))
        g = self.gbuffer_depth()
        gi = rtdgi_ptr if rtdgi_ptr is not None else self.out.screen_irradiance_tex
        r0, r1 = rows if rows is not None else (0, self.H)
        check(self.L.kj_light_gbuffer_rows(self.dev.h, C.byref(g), shadow_mask.data_ptr(), 1 if shadow_mask.dtype == t.float16 else 0, rtr_ptr, gi, self.sky64.data_ptr(), 64,
                                           self._lit[0].data_ptr(), self._lit[1].data_ptr(), debug_shading_mode, r0, r1, _stream_ptr()))
        return self._lit

    def ssgi_frame(self, rows=None):
        """SsgiRenderer::render (world_render_passes.rs:90-96): computes the SSAO guide; rtdgi's `ssao_tex` then points at it.
        `rows` = (row_begin, row_end): only those full-res rows (the screen-tile split, kj_ssgi_render_rows)."""
        if self.ssgi is None:
            self.ssgi = C.c_void_p()
            check(self.L.kj_ssgi_create(self.dev.h, C.byref(self.ssgi)))
        g = self.gbuffer_depth()
        if rows is None:
            check(self.L.kj_ssgi_render(self.ssgi, C.byref(g), self.reprojection_map_ptr, None, C.byref(self.ssao_ptr), _stream_ptr()))
        else:
            check(self.L.kj_ssgi_render_rows(self.ssgi, C.byref(g), self.reprojection_map_ptr, None, rows[0], rows[1], C.byref(self.ssao_ptr), _stream_ptr()))

    def ssgi_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_ssgi_surface(self.ssgi, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def reference_path_trace(self, accum, first_bounce_mode=0, interleave=(1, 0), ray_counter=None):
        """reference_path_trace (reference.rs:8-26): one more sample per pixel into `accum` (H, W, 4) float32 cuda tensor."""
        assert accum.dtype == self.torch.float32 and accum.is_contiguous() and tuple(accum.shape) == (self.H, self.W, 4)
        check(self.L.kj_reference_path_trace(self.dev.h, self.scene.h, accum.data_ptr(), self.W, self.H, first_bounce_mode, interleave[0], interleave[1],
                                             ray_counter.data_ptr() if ray_counter is not None else None, _stream_ptr()))

    def taa_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_taa_surface(self.taa, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def frame(self, fc):
        self.render_inputs(fc)
        self.reprojection()
        self.gi_frame()

    def ircache_buffer(self, name, dtype):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_ircache_buffer(self.ircache, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, (-1,))

    def ircache_ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.L.kj_ircache_ray_counts(self.ircache, C.byref(a), C.byref(b)))
        return a.value, b.value

    def surface(self, name, dtype, shape):
        """Wrap a named renderer surface as a torch tensor (no copy)."""
        torch = self.torch
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_rtdgi_surface(self.rtdgi, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.L.kj_rtdgi_ray_counts(self.rtdgi, C.byref(a), C.byref(b)))
        return a.value, b.value

    PASS_NAMES = ["rtdgi reproject", "extract half", "rtdgi validate", "rtdgi trace", "validity integrate", "restir temporal",
                  "restir spatial 0", "restir spatial 1", "restir resolve", "rtdgi temporal", "rtdgi spatial"]

    RAY_PASS_FORMS = {"grouped": 0, "fused": 1, "staged": 2, "split": 3, "quad": 4, "pool": 5}

    def set_ray_pass_form(self, form):
        """How `rtdgi validate` / `rtdgi trace` are scheduled (include/kajiya_amd.h: KJ_RTDGI_RAYS_*); outputs are identical for all."""
        check(self.L.kj_rtdgi_set_ray_pass_form(self.rtdgi, self.RAY_PASS_FORMS[form]))

    def set_pool_tune(self, waves_per_simd=3, refill_min=16, shade_a_min=16, shade_b_min=16, dynamic_tiles=False):
        """Scheduling knobs of the pool form of the ray passes (kj_rtdgi_set_pool_tune); results do not depend on them."""
        check(self.L.kj_rtdgi_set_pool_tune(self.rtdgi, waves_per_simd, refill_min, shade_a_min, shade_b_min, int(dynamic_tiles)))

    def set_profiling(self, pass_timers=True, count_traversal=False):
        check(self.L.kj_rtdgi_set_profiling(self.rtdgi, int(pass_timers), int(count_traversal)))

    def pass_times_ms(self):
        buf = (C.c_float * 11)()
        check(self.L.kj_rtdgi_pass_times_ms(self.rtdgi, buf, 11))
        return list(buf)

    def traversal_counts(self):
        buf = (C.c_uint64 * 6)()
        check(self.L.kj_rtdgi_traversal_counts(self.rtdgi, buf))
        return dict(zip(["closest_rays", "any_rays", "closest_nodes", "closest_tris", "any_nodes", "any_tris"], list(buf)))

    def __del__(self):
        try:
            self.L.kj_rtdgi_destroy(self.rtdgi)
            self.L.kj_reprojection_destroy(self.reproj)
            if self.ircache:
                self.L.kj_ircache_destroy(self.ircache)
            self.L.kj_taa_destroy(self.taa)
            if getattr(self, "ssgi", None):
                self.L.kj_ssgi_destroy(self.ssgi)
            if getattr(self, "shadow_dn", None):
                self.L.kj_shadow_denoise_destroy(self.shadow_dn)
            if getattr(self, "rtr", None):
                self.L.kj_rtr_destroy(self.rtr)
        except Exception:
            pass


class GpuPost:
    """PostProcessRenderer (renderers/post.rs:112-272) through the C-ABI: blur pyramid, luminance histogram, reverse blur pyramid and the
    post combine with the display transform. `bezold_brucke_lut`: (64, 2) float16, caller data (kajiya_amd/post_tables.py)."""

    def __init__(self, dev: Device, bezold_brucke_lut):
        self.L = load()
        self.dev = dev
        self._lut = np.ascontiguousarray(bezold_brucke_lut, np.float16).reshape(64, 2)
        self.h = C.c_void_p()
        check(self.L.kj_post_create(dev.h, self._lut.ctypes.data, C.byref(self.h)))

    def render(self, input_rgba16f, post_exposure_mult=1.0, contrast=1.0):
        """input: (H, W, 4) float16 cuda tensor (TaaOutput.this_frame_out / the motion-blurred frame) or float32 (the path tracer's
        accumulation image) -> (H, W) B10G11R11_UFLOAT words (int32 tensor view, owned by the handle, valid until the next call).
        kj_frame_begin must have been called for this frame."""
        import torch
        assert input_rgba16f.dtype in (torch.float16, torch.float32) and input_rgba16f.is_contiguous() and input_rgba16f.shape[-1] == 4
        self.H, self.W = int(input_rgba16f.shape[0]), int(input_rgba16f.shape[1])
        out = C.c_void_p()
        fmt = 1 if input_rgba16f.dtype == torch.float32 else 0      # KJ_POST_INPUT_RGBA32F / KJ_POST_INPUT_RGBA16F
        check(self.L.kj_post_render(self.h, input_rgba16f.data_ptr(), fmt, self.W, self.H, post_exposure_mult, contrast, C.byref(out), _stream_ptr()))
        return tensor_from_ptr(out.value, self.W * self.H * 4, torch.int32, (self.H, self.W))

    def mip_levels(self):
        n = C.c_uint32()
        check(self.L.kj_post_mip_levels(self.h, C.byref(n)))
        return n.value

    def mip_extent(self, level):
        pw, ph = (self.W + 1) // 2, (self.H + 1) // 2
        return max(1, pw >> level), max(1, ph >> level)

    def surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_post_surface(self.h, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def read_back_histogram(self, clipping_low=0.0, clipping_high=0.0):
        """(image_log2_lum, histogram[256]) from the host-visible copy; synchronise the stream first for this frame's values."""
        lum = C.c_float()
        hist = np.zeros(256, np.uint32)
        check(self.L.kj_post_read_back_histogram(self.h, clipping_low, clipping_high, C.byref(lum), hist.ctypes.data))
        return lum.value, hist

    def __del__(self):
        try:
            self.L.kj_post_destroy(self.h)
        except Exception:
            pass


class GpuMotionBlur:
    """motion_blur (renderers/motion_blur.rs:5-72) through the C-ABI."""

    def __init__(self, dev: Device):
        self.L = load()
        self.dev = dev
        self.h = C.c_void_p()
        check(self.L.kj_motion_blur_create(dev.h, C.byref(self.h)))

    def render(self, input_rgba16f, depth, reprojection_map):
        """input (H, W, 4) float16, depth (DH, DW) float32, reprojection_map (DH, DW, 4) int16 cuda tensors -> (H, W, 4) float16 view owned by
        the handle. kj_frame_begin must have been called for this frame."""
        import torch
        assert input_rgba16f.dtype == torch.float16 and depth.dtype == torch.float32 and reprojection_map.dtype == torch.int16
        assert input_rgba16f.is_contiguous() and depth.is_contiguous() and reprojection_map.is_contiguous()
        H, W = int(input_rgba16f.shape[0]), int(input_rgba16f.shape[1])
        DH, DW = int(depth.shape[0]), int(depth.shape[1])
        out = C.c_void_p()
        check(self.L.kj_motion_blur_render(self.h, input_rgba16f.data_ptr(), W, H, depth.data_ptr(), reprojection_map.data_ptr(), DW, DH, C.byref(out), _stream_ptr()))
        return tensor_from_ptr(out.value, W * H * 8, torch.float16, (H, W, 4))

    def surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(self.L.kj_motion_blur_surface(self.h, name.encode(), C.byref(ptr), C.byref(n)))
        return tensor_from_ptr(ptr.value, n.value, dtype, shape)

    def __del__(self):
        try:
            self.L.kj_motion_blur_destroy(self.h)
        except Exception:
            pass


def luminance_histogram_mean_log2(histogram, clipping_low=0.0, clipping_high=0.0):
    """PostProcessRenderer::read_back_histogram's arithmetic (post.rs:188-235) on a caller's 256-bin histogram; needs no device."""
    h = np.ascontiguousarray(histogram, np.uint32)
    assert h.size == 256
    lum = C.c_float()
    check(load().kj_luminance_histogram_mean_log2(h.ctypes.data, clipping_low, clipping_high, C.byref(lum)))
    return lum.value


class _CudaArrayView:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


def tensor_from_ptr(ptr, nbytes, dtype, shape):
    """Device pointer -> torch tensor view (via __cuda_array_interface__)."""
    import torch
    t = torch.as_tensor(_CudaArrayView(ptr, nbytes), device="cuda")
    return t.view(dtype).reshape(shape)
