"""ctypes mirror of include/kajiya_amd.h (struct layouts only; no library loading here)."""
import ctypes as C

c_f = C.c_float
c_u32 = C.c_uint32
c_i32 = C.c_int32


class KjViewConstants(C.Structure):
    _fields_ = [(n, c_f * 16) for n in (
        "view_to_clip", "clip_to_view", "view_to_sample", "sample_to_view", "world_to_view", "view_to_world",
        "clip_to_prev_clip", "prev_view_to_prev_clip", "prev_clip_to_prev_view", "prev_world_to_prev_view",
        "prev_view_to_prev_world")] + [("sample_offset_pixels", c_f * 2), ("sample_offset_clip", c_f * 2)]


class KjIrcacheCascadeConstants(C.Structure):
    _fields_ = [("origin", c_i32 * 4), ("voxels_scrolled_this_frame", c_i32 * 4)]


class KjRenderOverrides(C.Structure):
    _fields_ = [("flags", c_u32), ("material_roughness_scale", c_f), ("pad0", c_u32), ("pad1", c_u32)]


class KjFrameConstants(C.Structure):
    _fields_ = [
        ("view_constants", KjViewConstants),
        ("sun_direction", c_f * 4),
        ("frame_index", c_u32), ("delta_time_seconds", c_f), ("sun_angular_radius_cos", c_f), ("triangle_light_count", c_u32),
        ("sun_color_multiplier", c_f * 4),
        ("sky_ambient", c_f * 4),
        ("pre_exposure", c_f), ("pre_exposure_prev", c_f), ("pre_exposure_delta", c_f), ("pad0", c_f),
        ("render_overrides", KjRenderOverrides),
        ("ircache_grid_center", c_f * 4),
        ("ircache_cascades", KjIrcacheCascadeConstants * 12),
    ]


assert C.sizeof(KjFrameConstants) == 1216, C.sizeof(KjFrameConstants)


class KjPackedVertex(C.Structure):
    _fields_ = [("pos", c_f * 3), ("normal", c_u32)]


class KjMeshMaterial(C.Structure):
    _fields_ = [("base_color_mult", c_f * 4), ("maps", c_u32 * 4), ("roughness_mult", c_f), ("metalness_factor", c_f),
                ("emissive", c_f * 3), ("flags", c_u32), ("map_transforms", c_f * 24)]


assert C.sizeof(KjMeshMaterial) == 152


class KjMaterialMap(C.Structure):
    _fields_ = [("placeholder_rgba", C.c_uint8 * 4), ("image_rgba8", C.c_void_p), ("width", c_u32), ("height", c_u32), ("mip_count", c_u32), ("srgb", c_u32)]


class KjMeshDesc(C.Structure):
    _fields_ = [
        ("verts", C.c_void_p), ("vertex_count", c_u32),
        ("uvs", C.c_void_p), ("tangents", C.c_void_p), ("colors", C.c_void_p), ("material_ids", C.c_void_p),
        ("indices", C.c_void_p), ("index_count", c_u32),
        ("materials", C.c_void_p), ("material_count", c_u32),
        ("maps", C.c_void_p), ("map_count", c_u32),
        ("use_lights", c_u32),
    ]


class KjGbufferDepth(C.Structure):
    _fields_ = [("geometric_normal", C.c_void_p), ("gbuffer", C.c_void_p), ("depth", C.c_void_p), ("width", c_u32), ("height", c_u32)]


class KjRtdgiRenderParams(C.Structure):
    _fields_ = [
        ("gbuffer_depth", KjGbufferDepth),
        ("reprojection_map", C.c_void_p),
        ("sky_cube", C.c_void_p), ("sky_cube_width", c_u32),
        ("scene", C.c_void_p),
        ("ircache", C.c_void_p),
        ("ssao_tex", C.c_void_p),
        ("pass_mask", c_u32),
        ("row_begin", c_u32), ("row_end", c_u32), ("spatial_pass_select", c_u32),
    ]


class KjRtdgiOutput(C.Structure):
    _fields_ = [("screen_irradiance_tex", C.c_void_p), ("candidate_radiance_tex", C.c_void_p),
                ("candidate_normal_tex", C.c_void_p), ("candidate_hit_tex", C.c_void_p)]


KJ_RTDGI_PASS = dict(
    EXTRACT_HALF_NO_SSAO=1 << 9, EXTRACT_HALF_SSAO_ONLY=1 << 10, TRACE_MAY_DEFER=1 << 11, TRACE_FINISH=1 << 12,
    EXTRACT_HALF=1 << 0, VALIDATE=1 << 1, TRACE=1 << 2, VALIDITY_INTEGRATE=1 << 3, RESTIR_TEMPORAL=1 << 4,
    RESTIR_SPATIAL=1 << 5, RESTIR_RESOLVE=1 << 6, TEMPORAL_FILTER=1 << 7, SPATIAL_FILTER=1 << 8, ALL=0x1ff)


class KjTaaOutput(C.Structure):
    _fields_ = [("temporal_out", C.c_void_p), ("this_frame_out", C.c_void_p)]


class KjSplitRank(C.Structure):
    _fields_ = [("rtdgi", C.c_void_p), ("taa", C.c_void_p), ("ircache", C.c_void_p), ("scene", C.c_void_p)]


class KjSplitFrame(C.Structure):
    _fields_ = [("rtdgi", KjRtdgiRenderParams), ("rtdgi_out", C.POINTER(KjRtdgiOutput)), ("taa_out", C.POINTER(KjTaaOutput)), ("sky_cube16", C.c_void_p)]


class KjSplitProfile(C.Structure):
    _fields_ = [("exchange_ms", C.c_double), ("exchange_bytes_busiest_rank", C.c_uint64), ("exchange_points", C.c_uint32), ("gi_frames", C.c_uint32)]


class KjRtrTables(C.Structure):
    _fields_ = [("ranking_tile", C.c_void_p), ("scrambling_tile", C.c_void_p), ("sobol", C.c_void_p), ("spatial_resolve_offsets", C.c_void_p)]


class KjRtrParams(C.Structure):
    _fields_ = [
        ("gbuffer_depth", KjGbufferDepth),
        ("reprojection_map", C.c_void_p),
        ("sky_cube", C.c_void_p), ("sky_cube_width", c_u32),
        ("scene", C.c_void_p),
        ("ircache", C.c_void_p),
        ("rtdgi_irradiance", C.c_void_p),
        ("candidate_radiance_tex", C.c_void_p), ("candidate_hit_tex", C.c_void_p), ("candidate_normal_tex", C.c_void_p),
        ("pass_mask", c_u32),
    ]


KJ_RTR_PASS = {"TRACE": 1, "VALIDATE": 2, "RESTIR_TEMPORAL": 4, "RESOLVE": 8, "TEMPORAL_FILTER": 16, "CLEANUP": 32, "ALL": 63, "KEEP": 0x80000000}

