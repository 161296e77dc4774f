"""ORACLE (test infrastructure): ctypes loader + a frame driver for the CPU restatement.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from kajiya_amd.abi import (KjFrameConstants, KjMeshDesc, KjRtdgiRenderParams, KjRtdgiOutput, KJ_RTDGI_PASS, KjRtrTables, KjRtrParams)  # noqa: E402
from kajiya_amd import scenes as kscenes  # noqa: E402

_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle_kajiya.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.okj_hash1.restype = C.c_uint32; L.okj_hash1.argtypes = [C.c_uint32]
        L.okj_hash_combine2.restype = C.c_uint32; L.okj_hash_combine2.argtypes = [C.c_uint32, C.c_uint32]
        L.okj_hash3.restype = C.c_uint32; L.okj_hash3.argtypes = [C.c_uint32] * 3
        L.okj_uint_to_u01_float.restype = C.c_float; L.okj_uint_to_u01_float.argtypes = [C.c_uint32]
        L.okj_pack_normal_11_10_11.restype = C.c_uint32; L.okj_pack_normal_11_10_11.argtypes = [C.c_float] * 3
        L.okj_pack_color_888.restype = C.c_uint32; L.okj_pack_color_888.argtypes = [C.c_float] * 3
        L.okj_float3_to_rgb9e5.restype = C.c_uint32; L.okj_float3_to_rgb9e5.argtypes = [C.c_float] * 3
        L.okj_f32_to_f16.restype = C.c_uint16; L.okj_f32_to_f16.argtypes = [C.c_float]
        L.okj_f16_to_f32.restype = C.c_float; L.okj_f16_to_f32.argtypes = [C.c_uint16]
        L.okj_ris_estimate.restype = C.c_double; L.okj_ris_estimate.argtypes = [C.c_uint32] * 3
        L.okj_scene_create.restype = C.c_void_p
        L.okj_scene_destroy.argtypes = [C.c_void_p]
        L.okj_scene_add_mesh.restype = C.c_uint32; L.okj_scene_add_mesh.argtypes = [C.c_void_p, C.POINTER(KjMeshDesc)]
        L.okj_scene_add_instance.restype = C.c_uint32; L.okj_scene_add_instance.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.okj_scene_set_instance_transform.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.okj_scene_commit.argtypes = [C.c_void_p]
        L.okj_scene_use_bvh.argtypes = [C.c_void_p, C.c_int]
        L.okj_scene_triangle_count.restype = C.c_uint32; L.okj_scene_triangle_count.argtypes = [C.c_void_p]
        L.okj_scene_triangle_light_count.restype = C.c_uint32; L.okj_scene_triangle_light_count.argtypes = [C.c_void_p]
        L.okj_scene_tri_ids.argtypes = [C.c_void_p, C.c_void_p]
        L.okj_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.okj_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.okj_raster_gbuffer.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_uint32, C.c_uint32] + [C.c_void_p] * 4
        L.okj_calculate_reprojection_map.argtypes = [C.POINTER(KjFrameConstants), C.c_uint32, C.c_uint32] + [C.c_void_p] * 5
        L.okj_brdf_fg_lut.argtypes = [C.c_void_p]
        L.okj_sky_cube_render.argtypes = [C.POINTER(KjFrameConstants), C.c_void_p]
        L.okj_sky_cube_convolve.argtypes = [C.c_void_p, C.c_void_p]
        L.okj_sample_cube.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.okj_sun_color.argtypes = [C.POINTER(KjFrameConstants), C.c_void_p]
        L.okj_rtdgi_create.restype = C.c_void_p; L.okj_rtdgi_create.argtypes = [C.c_void_p, C.c_void_p]
        L.okj_rtdgi_destroy.argtypes = [C.c_void_p]
        L.okj_rtdgi_set_options.argtypes = [C.c_void_p, C.c_uint32]
        L.okj_rtdgi_set_raytraced_visibility.argtypes = [C.c_void_p, C.c_int]
        L.okj_rtdgi_reproject.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_uint32]
        L.okj_rtdgi_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.POINTER(KjRtdgiRenderParams), C.POINTER(KjRtdgiOutput)]
        L.okj_rtdgi_surface.restype = C.c_int
        L.okj_rtdgi_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_rtdgi_ray_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.okj_ircache_create.restype = C.c_void_p; L.okj_ircache_create.argtypes = [C.c_void_p]
        L.okj_ircache_destroy.argtypes = [C.c_void_p]
        L.okj_ircache_core.restype = C.c_void_p; L.okj_ircache_core.argtypes = [C.c_void_p]
        L.okj_ircache_update_eye_position.argtypes = [C.c_void_p, C.c_void_p]
        L.okj_ircache_constants.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants)]
        L.okj_ircache_prepare.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants)]
        L.okj_ircache_trace_irradiance.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_int]
        L.okj_ircache_sum_up.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants)]
        L.okj_ircache_buffer.restype = C.c_int
        L.okj_ircache_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_ircache_ray_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.okj_ircache_set_deferred_updates.argtypes = [C.c_void_p, C.c_int]
        L.okj_ircache_set_chain_schedule.argtypes = [C.c_void_p, C.c_int]
        L.okj_ref_trace_hook_ptr.restype = C.c_void_p
        L.okj_scene_mesh_count.argtypes = [C.c_void_p]; L.okj_scene_instance_count.argtypes = [C.c_void_p]; L.okj_scene_map_count.argtypes = [C.c_void_p]
        L.okj_scene_vertex_buffer_bytes.argtypes = [C.c_void_p]; L.okj_scene_vertex_buffer_bytes.restype = C.c_uint64
        L.okj_scene_export_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.okj_scene_map_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.okj_ircache_host_state.argtypes = [C.c_void_p, C.c_void_p]
        L.okj_ircache_prepare_and_reset.argtypes = [C.c_void_p]
        L.okj_ircache_ray_pass.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.okj_ircache_begin_requests.argtypes = [C.c_void_p]
        L.okj_ircache_apply_requests.argtypes = [C.c_void_p]
        L.okj_ircache_request_count.argtypes = [C.c_void_p]; L.okj_ircache_request_count.restype = C.c_uint64
        L.okj_taa_create.restype = C.c_void_p
        L.okj_taa_destroy.argtypes = [C.c_void_p]
        L.okj_taa_render.restype = C.c_void_p
        L.okj_taa_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.okj_taa_surface.restype = C.c_int
        L.okj_taa_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_trace_sun_shadow_mask.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.okj_rtr_create.restype = C.c_void_p; L.okj_rtr_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KjRtrTables)]
        L.okj_rtr_destroy.argtypes = [C.c_void_p]
        L.okj_rtr_set_options.argtypes = [C.c_void_p, C.c_uint32]
        L.okj_rtr_set_literal_own_sample_shadowing.argtypes = [C.c_void_p, C.c_uint32]
        L.okj_rtr_trace.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.POINTER(KjRtrParams)]
        L.okj_rtr_filter_temporal.restype = C.c_void_p; L.okj_rtr_filter_temporal.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.POINTER(KjRtrParams)]
        L.okj_rtr_surface.restype = C.c_int; L.okj_rtr_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_rtr_ray_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.okj_lighting_render_specular.restype = C.c_uint64
        L.okj_lighting_render_specular.argtypes = [C.POINTER(KjFrameConstants)] + [C.c_void_p] * 7 + [C.c_uint32, C.c_uint32]
        L.okj_light_gbuffer.argtypes = [C.POINTER(KjFrameConstants)] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.okj_shadow_denoise_create.restype = C.c_void_p
        L.okj_shadow_denoise_destroy.argtypes = [C.c_void_p]
        L.okj_shadow_denoise_render.restype = C.c_void_p
        L.okj_shadow_denoise_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.okj_shadow_denoise_surface.restype = C.c_int
        L.okj_shadow_denoise_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_ssgi_create.restype = C.c_void_p
        L.okj_ssgi_destroy.argtypes = [C.c_void_p]
        L.okj_ssgi_render.restype = C.c_void_p
        L.okj_ssgi_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.okj_ssgi_surface.restype = C.c_int
        L.okj_ssgi_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_post_create.restype = C.c_void_p
        L.okj_post_destroy.argtypes = [C.c_void_p]
        L.okj_post_render.restype = C.c_void_p
        L.okj_post_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.okj_post_surface.restype = C.c_int
        L.okj_post_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_post_mip_levels.restype = C.c_int
        L.okj_post_mip_levels.argtypes = [C.c_void_p]
        L.okj_post_read_back_histogram.restype = C.c_float
        L.okj_post_read_back_histogram.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.okj_display_transform_srgb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.okj_motion_blur_create.restype = C.c_void_p
        L.okj_motion_blur_destroy.argtypes = [C.c_void_p]
        L.okj_motion_blur_render.restype = C.c_void_p
        L.okj_motion_blur_render.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.okj_motion_blur_surface.restype = C.c_int
        L.okj_motion_blur_surface.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.okj_reference_path_trace.restype = C.c_uint64
        L.okj_reference_path_trace.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.okj_reference_path_trace_rows.restype = C.c_uint64
        L.okj_reference_path_trace_rows.argtypes = [C.c_void_p, C.POINTER(KjFrameConstants), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.okj_probe_functions.restype = C.c_uint32; L.okj_probe_functions.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.okj_probe_functions_color.restype = C.c_uint32; L.okj_probe_functions_color.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.okj_probe_functions_shading.restype = C.c_uint32; L.okj_probe_functions_shading.argtypes = [C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.okj_probe_functions_misc.restype = C.c_uint32; L.okj_probe_functions_misc.argtypes = [C.POINTER(KjFrameConstants), C.c_void_p, C.c_uint32, C.c_void_p]
        L.okj_set_threads.argtypes = [C.c_int]
        L.okj_get_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def blue_noise():
    return np.fromfile(os.path.join(kscenes.GOLDEN_DIR, "bluenoise_256_rgba8.bin"), dtype=np.uint8)


_BRDF_LUT = None


def brdf_lut():
    global _BRDF_LUT
    if _BRDF_LUT is None:
        out = np.zeros((64, 64, 4), np.uint16)
        lib().okj_brdf_fg_lut(out.ctypes.data)
        _BRDF_LUT = out
    return _BRDF_LUT


def probe_functions(inputs, rows):
    """The restated leaf functions on `inputs` (n, 4) uint32: (rows, n, 4) uint32, in the row order of oracle/ref_hlsl/probes/inc_functions.hlsl."""
    inputs = np.ascontiguousarray(inputs, np.uint32)
    out = np.zeros((rows, inputs.shape[0], 4), np.uint32)
    got = lib().okj_probe_functions(inputs.ctypes.data, inputs.shape[0], out.ctypes.data)
    assert got == rows, (got, rows)
    return out


def probe_functions_color(inputs, rows, bezold_brucke_lut):
    """The restated colour / G-buffer / sky functions on `inputs` (n, 4) uint32: (rows, n, 4) uint32, in the row order of oracle/ref_hlsl/probes/inc_functions_color.hlsl."""
    inputs = np.ascontiguousarray(inputs, np.uint32)
    lut = np.ascontiguousarray(bezold_brucke_lut, np.float16).reshape(64, 2)
    out = np.zeros((rows, inputs.shape[0], 4), np.uint32)
    got = lib().okj_probe_functions_color(inputs.ctypes.data, inputs.shape[0], lut.ctypes.data, out.ctypes.data)
    assert got == rows, (got, rows)
    return out


def probe_functions_shading(fc, inputs, rows):
    """The restated view-ray / layered-BRDF / sun / light-sampling functions on `inputs` (n, 4) uint32 under frame constants `fc`: (rows, n, 4) uint32, in the row order of
    oracle/ref_hlsl/probes/inc_functions_shading.hlsl."""
    inputs = np.ascontiguousarray(inputs, np.uint32)
    out = np.zeros((rows, inputs.shape[0], 4), np.uint32)
    got = lib().okj_probe_functions_shading(C.byref(fc), inputs.ctypes.data, inputs.shape[0], brdf_lut().ctypes.data, out.ctypes.data)
    assert got == rows, (got, rows)
    return out


def probe_functions_misc(fc, inputs, rows):
    """taa_common / bilinear / TemporalReservoirOutput / SampleParams / ws_pos_to_ircache_coord on `inputs` (n, 4) uint32 under `fc`: (rows, n, 4) uint32, in the row order
    of oracle/ref_hlsl/probes/inc_functions_misc.hlsl."""
    inputs = np.ascontiguousarray(inputs, np.uint32)
    out = np.zeros((rows, inputs.shape[0], 4), np.uint32)
    got = lib().okj_probe_functions_misc(C.byref(fc), inputs.ctypes.data, inputs.shape[0], out.ctypes.data)
    assert got == rows, (got, rows)
    return out


def reference_path_trace(scene, fc, output, first_bounce_mode=0):
    """reference.rs:8-26: accumulate one sample/pixel into `output` (H, W, 4) float32. Returns the number of rays traced."""
    assert output.dtype == np.float32 and output.flags["C_CONTIGUOUS"] and output.shape[2] == 4
    h, w = output.shape[:2]
    return lib().okj_reference_path_trace(scene.h, C.byref(fc), brdf_lut().ctypes.data, output.ctypes.data, w, h, first_bounce_mode)


def reference_path_trace_rows(scene, fc, output, row_begin, row_end):
    """One sample per pixel of rows [row_begin, row_end) of the frame `output` (H, W, 4 float32) belongs to."""
    h, w = output.shape[:2]
    return lib().okj_reference_path_trace_rows(scene.h, C.byref(fc), brdf_lut().ctypes.data, output.ctypes.data, w, h, row_begin, row_end)


class OracleScene:
    def __init__(self, desc: kscenes.SceneDesc, use_lights=False):
        L = lib()
        self.h = L.okj_scene_create()
        self._keep = []
        for m in desc.meshes:
            d, keep = m.pack(use_lights)
            self._keep.append(keep)
            L.okj_scene_add_mesh(self.h, C.byref(d))
        for mi, xf in desc.instances:
            L.okj_scene_add_instance(self.h, mi, xf.ctypes.data)
        L.okj_scene_commit(self.h)
        self.triangle_count = L.okj_scene_triangle_count(self.h)
        self.triangle_light_count = L.okj_scene_triangle_light_count(self.h)

    def trace_closest(self, rays, cull_back=False, brute=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), np.float32)
        lib().okj_trace_closest(self.h, rays.ctypes.data, hits.ctypes.data, len(rays), int(cull_back), int(brute))
        return hits

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), np.uint8)
        lib().okj_trace_any(self.h, rays.ctypes.data, out.ctypes.data, len(rays))
        return out

    def tri_ids(self):
        out = np.zeros((self.triangle_count, 2), np.uint32)
        lib().okj_scene_tri_ids(self.h, out.ctypes.data)
        return out

    def __del__(self):
        try:
            lib().okj_scene_destroy(self.h)
        except Exception:
            pass


class OraclePipeline:
    """CPU restatement of one frame of the hot path (world_render_passes.rs:13-292, the subset in
    scope): sky cubes -> G-buffer stand-in -> reprojection map -> rtdgi.reproject -> rtdgi.render."""

    def __init__(self, scene: OracleScene, width, height, use_ircache=False):
        L = lib()
        self.L = L
        self.scene = scene
        self.W, self.H = width, height
        self.ircache = L.okj_ircache_create(brdf_lut().ctypes.data) if use_ircache else None
        self.bn = blue_noise()
        self.rtdgi = L.okj_rtdgi_create(self.bn.ctypes.data, brdf_lut().ctypes.data)
        W, H = width, height
        self.geometric_normal = np.zeros((H, W), np.uint32)
        self.gbuffer = np.zeros((H, W, 4), np.uint32)
        self.depth = np.zeros((H, W), np.float32)
        self.velocity = np.zeros((H, W, 4), np.uint16)
        self.prev_depth = np.zeros((H, W), np.float32)
        self.reprojection_map = np.zeros((H, W, 4), np.int16)
        self.ssao = np.full((H, W), 255, np.uint8)
        self.sky64 = np.zeros((6, 64, 64, 4), np.uint16)
        self.sky16 = np.zeros((6, 16, 16, 4), np.uint16)
        self._sky_key = None
        self.out = KjRtdgiOutput()

    def render_inputs(self, fc):
        L = self.L
        key = bytes(fc.sun_direction) + bytes(fc.sun_color_multiplier) + bytes(fc.sky_ambient) + bytes(C.c_float(fc.pre_exposure))
        if key != self._sky_key:
            L.okj_sky_cube_render(C.byref(fc), self.sky64.ctypes.data)
            L.okj_sky_cube_convolve(self.sky64.ctypes.data, self.sky16.ctypes.data)
            self._sky_key = key
        L.okj_raster_gbuffer(self.scene.h, C.byref(fc), self.W, self.H, self.geometric_normal.ctypes.data,
                             self.gbuffer.ctypes.data, self.depth.ctypes.data, self.velocity.ctypes.data)

    def reprojection(self, fc):
        self.L.okj_calculate_reprojection_map(C.byref(fc), self.W, self.H, self.depth.ctypes.data, self.geometric_normal.ctypes.data,
                                              self.prev_depth.ctypes.data, self.velocity.ctypes.data, self.reprojection_map.ctypes.data)
        self.prev_depth[...] = self.depth  # "copy depth" pass (renderers/reprojection.rs:37-49)

    def params(self, pass_mask=KJ_RTDGI_PASS["ALL"]):
        p = KjRtdgiRenderParams()
        p.gbuffer_depth.geometric_normal = self.geometric_normal.ctypes.data
        p.gbuffer_depth.gbuffer = self.gbuffer.ctypes.data
        p.gbuffer_depth.depth = self.depth.ctypes.data
        p.gbuffer_depth.width, p.gbuffer_depth.height = self.W, self.H
        p.reprojection_map = self.reprojection_map.ctypes.data
        p.sky_cube = self.sky16.ctypes.data
        p.sky_cube_width = 16
        p.scene = self.scene.h
        p.ircache = self.L.okj_ircache_core(self.ircache) if self.ircache else None
        p.ssao_tex = self.ssao.ctypes.data
        p.pass_mask = pass_mask
        return p

    def rtdgi_frame(self, fc, pass_mask=KJ_RTDGI_PASS["ALL"]):
        self.L.okj_rtdgi_reproject(self.rtdgi, C.byref(fc), self.reprojection_map.ctypes.data, self.W, self.H)
        p = self.params(pass_mask)
        self.L.okj_rtdgi_render(self.rtdgi, C.byref(fc), C.byref(p), C.byref(self.out))

    def ircache_prepare_and_trace(self, fc):
        """ircache.prepare + trace_irradiance (world_render_passes.rs:99,113-121)."""
        self.L.okj_ircache_prepare(self.ircache, C.byref(fc))
        self.L.okj_ircache_trace_irradiance(self.ircache, C.byref(fc), self.scene.h, self.sky16.ctypes.data, 16)

    def ircache_sum_up(self, fc):
        self.L.okj_ircache_sum_up(self.ircache, C.byref(fc))

    def gi_frame(self, fc, pass_mask=KJ_RTDGI_PASS["ALL"]):
        """The GI frame in world_render_passes.rs order: ircache prepare/trace, rtdgi.reproject,
        ircache sum-up (deliberately delayed, :138-140), rtdgi.render."""
        deferred = self.ircache and getattr(self, "ircache_deferred", False)
        if self.ircache:
            if deferred:
                self.L.okj_ircache_begin_requests(self.ircache)
            self.ircache_prepare_and_trace(fc)
        self.L.okj_rtdgi_reproject(self.rtdgi, C.byref(fc), self.reprojection_map.ctypes.data, self.W, self.H)
        if self.ircache:
            self.ircache_sum_up(fc)
        p = self.params(pass_mask)
        self.L.okj_rtdgi_render(self.rtdgi, C.byref(fc), C.byref(p), C.byref(self.out))
        if deferred:
            self.L.okj_ircache_apply_requests(self.ircache)

    def ircache_set_deferred(self, enable=True):
        """The cache's deterministic mode (oracle/okj_ircache.hpp header): the same order-free semantics as the product's
        kj_ircache_set_deferred_updates, so that cache state can be compared under the deterministic passes' bars."""
        self.L.okj_ircache_set_deferred_updates(self.ircache, int(enable))
        self.ircache_deferred = bool(enable)

    def ircache_set_chain_schedule(self, enable=True):
        """Deterministic mode only: which schedule of the cache's three ray passes the oracle restates -- the product's default (KJ_IRC_PASSES_CHAIN: tracing's lookups read
        the state before the passes, on by default here too) or three launches with a snapshot between validation and tracing (kj_ircache_set_ray_pass_schedule(SEQUENTIAL))."""
        self.L.okj_ircache_set_chain_schedule(self.ircache, int(enable))

    def taa_frame(self, fc, input_ptr=None, out_extent=None):
        """TaaRenderer::render on `input_ptr` (default: this frame's rtdgi screen_irradiance_tex)."""
        if not hasattr(self, "taa"):
            self.taa = self.L.okj_taa_create()
        ow, oh = out_extent or (self.W, self.H)
        inp = input_ptr if input_ptr is not None else self.out.screen_irradiance_tex
        tout = C.c_void_p()
        r = self.L.okj_taa_render(self.taa, C.byref(fc), inp, self.W, self.H, self.reprojection_map.ctypes.data, self.depth.ctypes.data, ow, oh, C.byref(tout))
        return r, tout.value

    def sun_shadow_mask(self, fc):
        """trace_sun_shadow_mask (renderers/shadows.rs:10-40)."""
        out = np.zeros((self.H, self.W), np.uint8)
        self.L.okj_trace_sun_shadow_mask(self.scene.h, C.byref(fc), self.bn.ctypes.data, self.depth.ctypes.data, self.geometric_normal.ctypes.data, out.ctypes.data, self.W, self.H)
        return out

    def light_gbuffer(self, fc, shadow_mask, rtdgi=None, rtr=None, mode=0):
        """light_gbuffer (renderers/deferred.rs:6-60): returns (temporal_output, output) as (H, W, 4) float16 arrays."""
        t = np.zeros((self.H, self.W, 4), np.float16); o = np.zeros((self.H, self.W, 4), np.float16)
        gi = rtdgi if rtdgi is not None else self.surface("spatial_filtered_tex", np.float16, (self.H, self.W, 4))
        gi = np.ascontiguousarray(gi)
        self.L.okj_light_gbuffer(C.byref(fc), brdf_lut().ctypes.data, self.gbuffer.ctypes.data, self.depth.ctypes.data, np.ascontiguousarray(shadow_mask).ctypes.data,
                                 rtr.ctypes.data if rtr is not None else None, gi.ctypes.data, self.sky64.ctypes.data, 64, t.ctypes.data, o.ctypes.data, self.W, self.H, mode)
        return t, o

    def shadow_denoise(self, fc, shadow_mask):
        """ShadowDenoiseRenderer::render (world_render_passes.rs:131-136): returns the denoised shadow term as (H, W) float32."""
        if not hasattr(self, "shadow_dn"):
            self.shadow_dn = self.L.okj_shadow_denoise_create()
        m = np.ascontiguousarray(shadow_mask, np.uint8)
        ptr = self.L.okj_shadow_denoise_render(self.shadow_dn, C.byref(fc), m.ctypes.data, self.depth.ctypes.data, self.geometric_normal.ctypes.data, self.reprojection_map.ctypes.data, self.W, self.H)
        buf = (C.c_uint16 * (self.W * self.H * 2)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float16).reshape(self.H, self.W, 2)[..., 0].astype(np.float32)

    def shadow_denoise_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_shadow_denoise_surface(self.shadow_dn, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def ssgi_frame(self, fc):
        """SsgiRenderer::render (world_render_passes.rs:90-96): computes the SSAO guide and binds it as rtdgi's ssao_tex."""
        if not hasattr(self, "ssgi"):
            self.ssgi = self.L.okj_ssgi_create()
        ptr = self.L.okj_ssgi_render(self.ssgi, C.byref(fc), self.gbuffer.ctypes.data, self.depth.ctypes.data, self.reprojection_map.ctypes.data, self.W, self.H)
        buf = (C.c_uint8 * (self.W * self.H)).from_address(ptr)
        self.ssao = np.frombuffer(buf, dtype=np.uint8).reshape(self.H, self.W)
        return self.ssao

    def ssgi_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_ssgi_surface(self.ssgi, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def rtr_params(self, pass_mask=63):
        """KjRtrParams for RtrRenderer::trace / filter_temporal (world_render_passes.rs:172-210): the unconvolved sky cube, this
        frame's rtdgi output and candidates."""
        p = KjRtrParams()
        p.gbuffer_depth.geometric_normal = self.geometric_normal.ctypes.data
        p.gbuffer_depth.gbuffer = self.gbuffer.ctypes.data
        p.gbuffer_depth.depth = self.depth.ctypes.data
        p.gbuffer_depth.width, p.gbuffer_depth.height = self.W, self.H
        p.reprojection_map = self.reprojection_map.ctypes.data
        p.sky_cube = self.sky64.ctypes.data
        p.sky_cube_width = 64
        p.scene = self.scene.h
        p.ircache = self.L.okj_ircache_core(self.ircache) if self.ircache else None
        p.rtdgi_irradiance = self.out.screen_irradiance_tex
        p.candidate_radiance_tex = self.out.candidate_radiance_tex
        p.candidate_hit_tex = self.out.candidate_hit_tex
        p.candidate_normal_tex = self.out.candidate_normal_tex
        p.pass_mask = pass_mask
        return p

    def rtr_frame(self, fc, pass_mask=63):
        """RtrRenderer::trace + TracedRtr::filter_temporal after rtdgi.render; returns the resolved image as (H, W) uint32
        (B10G11R11_UFLOAT)."""
        if not hasattr(self, "rtr"):
            from kajiya_amd import rtr_tables
            t, self._rtr_keep = rtr_tables.standin_tables()
            self.rtr = self.L.okj_rtr_create(self.bn.ctypes.data, brdf_lut().ctypes.data, C.byref(t))
        p = self.rtr_params(pass_mask)
        if pass_mask & 15:
            self.L.okj_rtr_trace(self.rtr, C.byref(fc), C.byref(p))
        if pass_mask & 48:
            self.L.okj_rtr_filter_temporal(self.rtr, C.byref(fc), C.byref(p))
        return self.rtr_surface("resolved_tex", np.uint32, (self.H, self.W))

    def lighting_render_specular(self, fc, resolved_r11g11b10f):
        """LightingRenderer::render_specular (renderers/lighting.rs:23-88): adds the triangle lights' specular into `resolved_r11g11b10f`
        ((H, W) uint32, modified in place). Returns the number of shadow rays."""
        from kajiya_amd import rtr_tables
        t, keep = rtr_tables.standin_tables()
        assert resolved_r11g11b10f.dtype == np.uint32 and resolved_r11g11b10f.flags["C_CONTIGUOUS"]
        return self.L.okj_lighting_render_specular(C.byref(fc), self.scene.h, self.bn.ctypes.data, brdf_lut().ctypes.data, self.gbuffer.ctypes.data, self.depth.ctypes.data,
                                                   keep[3].ctypes.data, resolved_r11g11b10f.ctypes.data, self.W, self.H)

    def rtr_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_rtr_surface(self.rtr, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def rtr_ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.okj_rtr_ray_counts(self.rtr, C.byref(a), C.byref(b))
        return a.value, b.value

    def taa_surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_taa_surface(self.taa, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def frame(self, fc):
        self.render_inputs(fc)
        self.reprojection(fc)
        self.gi_frame(fc)

    def ircache_buffer(self, name, dtype):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_ircache_buffer(self.ircache, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype)

    def ircache_ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.okj_ircache_ray_counts(self.ircache, C.byref(a), C.byref(b))
        return a.value, b.value

    def surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_rtdgi_surface(self.rtdgi, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        buf = (C.c_uint8 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def ray_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.okj_rtdgi_ray_counts(self.rtdgi, C.byref(a), C.byref(b))
        return a.value, b.value


class OraclePost:
    """PostProcessRenderer (renderers/post.rs:112-272) on the CPU: blur pyramid, luminance histogram, reverse blur pyramid, post combine."""

    def __init__(self, bezold_brucke_lut):
        self.L = lib()
        self.h = self.L.okj_post_create()
        self.lut = np.ascontiguousarray(bezold_brucke_lut, np.float16).reshape(64, 2)
        self.bn = blue_noise()

    def render(self, fc, input_rgba16f, post_exposure_mult=1.0, contrast=1.0):
        """input (H, W, 4) float16 — or float32: the path tracer's accumulation image — -> (H, W) uint32 B10G11R11_UFLOAT"""
        is32 = np.asarray(input_rgba16f).dtype == np.float32
        inp = np.ascontiguousarray(input_rgba16f, np.float32 if is32 else np.float16)
        H, W = inp.shape[:2]
        self.W, self.H = W, H
        ptr = self.L.okj_post_render(self.h, C.byref(fc), inp.ctypes.data, 1 if is32 else 0, W, H, self.lut.ctypes.data, self.bn.ctypes.data, post_exposure_mult, contrast)
        return np.frombuffer((C.c_uint8 * (W * H * 4)).from_address(ptr), dtype=np.uint32).reshape(H, W)

    def mip_levels(self):
        return self.L.okj_post_mip_levels(self.h)

    def mip_extent(self, level):
        pw, ph = (self.W + 1) // 2, (self.H + 1) // 2
        return max(1, pw >> level), max(1, ph >> level)

    def surface(self, name, dtype=np.uint32, shape=None):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_post_surface(self.h, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        a = np.frombuffer((C.c_uint8 * n.value).from_address(ptr.value), dtype=dtype)
        return a.reshape(shape) if shape is not None else a

    def mip(self, pyramid, level):
        w, h = self.mip_extent(level)
        return self.surface(f"{pyramid}:{level}", np.uint32, (h, w))

    def histogram(self):
        return self.surface("histogram", np.uint32, (256,)).copy()

    def read_back_histogram(self, histogram, clipping_low=0.0, clipping_high=0.0):
        h = np.ascontiguousarray(histogram, np.uint32)
        return self.L.okj_post_read_back_histogram(h.ctypes.data, clipping_low, clipping_high)

    def display_transform(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        out = np.empty_like(rgb)
        self.L.okj_display_transform_srgb(self.lut.ctypes.data, rgb.ctypes.data, out.ctypes.data, len(rgb))
        return out

    def __del__(self):
        try:
            self.L.okj_post_destroy(self.h)
        except Exception:
            pass


class OracleMotionBlur:
    """motion_blur (renderers/motion_blur.rs:5-72) on the CPU."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.okj_motion_blur_create()

    def render(self, fc, input_rgba16f, depth, reprojection_map):
        """input (H, W, 4) float16, depth (DH, DW) float32, reprojection_map (DH, DW, 4) int16 -> (H, W, 4) float16"""
        inp = np.ascontiguousarray(input_rgba16f, np.float16)
        d = np.ascontiguousarray(depth, np.float32)
        r = np.ascontiguousarray(reprojection_map, np.int16)
        H, W = inp.shape[:2]
        DH, DW = d.shape
        assert r.shape == (DH, DW, 4)
        ptr = self.L.okj_motion_blur_render(self.h, C.byref(fc), inp.ctypes.data, W, H, d.ctypes.data, r.ctypes.data, DW, DH)
        return np.frombuffer((C.c_uint8 * (W * H * 8)).from_address(ptr), dtype=np.float16).reshape(H, W, 4)

    def surface(self, name, dtype, shape):
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.L.okj_motion_blur_surface(self.h, name.encode(), C.byref(ptr), C.byref(n)) != 0:
            raise KeyError(name)
        return np.frombuffer((C.c_uint8 * n.value).from_address(ptr.value), dtype=dtype).reshape(shape)

    def __del__(self):
        try:
            self.L.okj_motion_blur_destroy(self.h)
        except Exception:
            pass
