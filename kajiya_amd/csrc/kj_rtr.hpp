// Shared by the two translation units of RtrRenderer (rtr.hip: the ray passes, the reservoir pass and the host side; rtr_screen.hip: the
// full-resolution screen-space passes): texel typedefs, the reference's rtr_settings.hlsl constants, small helpers, and the launchers of
// rtr_screen.hip's kernels.
#pragma once
#include "kj_host.hpp"
#include "kj_scene.hpp"
#include "kj_reservoir.hpp"

using namespace kj;

#define SKY_DIST 1e4f
#define RTR_ROUGHNESS_CLAMP 6e-4f
#define RTR_RESTIR_MAX_PDF_CLAMP 200.0f
#define RTR_RESTIR_TEMPORAL_M_CLAMP 8.0f
#define RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS 0.5f
#define RTR_SAMPLING_BIAS 0.15f

typedef Img<uint2> ImgH4;     // RGBA16F
typedef Img<uint32_t> ImgU32; // RGBA8_SNORM / RG16F / A2R10G10B10 / B10G11R11_UFLOAT / R32_UINT
typedef Img<float> ImgF32;
typedef Img<uint4> ImgU4;
typedef Img<uint2> ImgU2;     // RG32UI (reservoirs) and RGBA16_SNORM share the 8-byte texel
typedef Img<uint8_t> ImgR8;
typedef Img<float4> ImgF4;

// workgroup -> tile order (kj_vec.hpp: tile_order; profiles/r03_xcd_tile_order.md): the passes whose every tile costs the same take whole tile rows per
// XCD (TILE_XY_ROWS: temporal filter 180 -> 172 us, cleanup 51 -> 44, extract_half 14.6 -> 13.0 at 1440p); the ray passes, the reservoir pass and
// the resolve keep the plain order (measured equal or 1-2 % slower with rows)
// `tile_row0` (a kernel argument in scope): the first 8-row tile row of the launch -- 0 for the whole image; the screen-tile split launches a strip's tile rows only
#define TILE_XY_ORDER(W_, H_, MODE_)                                      \
    const int lane = threadIdx.x;                                         \
    const uint2 kj_tb = kj::tile_order<MODE_>();                          \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = (int(kj_tb.y) + tile_row0) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < (H_);
#define TILE_XY(W_, H_) TILE_XY_ORDER(W_, H_, KJ_TILES_PLAIN)
#define TILE_XY_ROWS(W_, H_) TILE_XY_ORDER(W_, H_, KJ_TILES_ROWS)

// ------------------------------------------------------------------ small device helpers
KJ_D V3 get_prev_eye_position(const FrameConstants& fc) { const V4 e = mul44(fc.view_constants.prev_view_to_prev_world, V4{0, 0, 0, 1}); return xyz(e) / e.w; }
KJ_D V3 position_world_to_view(const FrameConstants& fc, V3 v) { return xyz(mul44(fc.view_constants.world_to_view, v4(v, 1))); }
KJ_D float depth_to_view_z(const FrameConstants& fc, float depth) { return 1.0f / (depth * -fc.view_constants.clip_to_view[11]); }
KJ_D I2 hi_px_subpixel(uint32_t k) { return halfres_subsample_offset(k); }   // hi_px_subpixels[k & 3]
KJ_D float ggx_ndf_0_1(float a2, float cos_theta) { const float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f; return a2 * a2 / (d * d); }
KJ_D float exponential_squish(float len, float s) { return exp2f(-clampf(s * len, 0.0f, 100.0f)); }
KJ_D float exponential_unsquish(float len, float s) { return fmaxf(0.0f, -1.0f / s * log2f(1e-30f + len)); }
KJ_D V3 soft_color_clamp(V3 center, V3 history, V3 ex, V3 dev) {
    const V3 history_dist = vabs(history - ex) / vmax(vabs(history * 0.1f), dev);
    const V3 closest_pt = vclamp(history, center - dev, center + dev);
    return V3{lerp(history.x, closest_pt.x, smoothstep(1.0f, 3.0f, history_dist.x)), lerp(history.y, closest_pt.y, smoothstep(1.0f, 3.0f, history_dist.y)),
              lerp(history.z, closest_pt.z, smoothstep(1.0f, 3.0f, history_dist.z))};
}
// rtr_restir_pack_unpack.inc.hlsl
struct RtrRestirRayOrigin { V3 ray_origin_eye_offset_ws; float roughness; uint32_t frame_index_mod4; };
KJ_D RtrRestirRayOrigin ray_origin_from_raw(float4 raw) {
    const V2 misc = unpack_2x16f_uint(asuint(raw.w));
    return RtrRestirRayOrigin{V3{raw.x, raw.y, raw.z}, misc.x, uint32_t(misc.y) & 3u};
}
KJ_D float4 ray_origin_to_raw(V3 o, float roughness, uint32_t frame_index_mod4) { return make_float4(o.x, o.y, o.z, asfloat(pack_2x16f_uint(roughness, float(frame_index_mod4)))); }

// ---- rtr_screen.hip
struct RtrResolveArgs {
    const FrameConstants* fc;
    ImgU4 gbuffer_tex; ImgF32 depth_tex; ImgH4 hit1_tex; ImgU2 reprojection_tex; ImgU32 half_view_normal_tex; ImgU32 ray_len_history_tex;
    ImgH4 restir_irradiance_tex, restir_ray_tex; ImgU2 restir_reservoir_tex; ImgF4 restir_ray_orig_tex;
    ImgU32 output_tex; ImgU32 ray_len_output_tex;
    const uint32_t* blue_noise; const uint2* brdf_fg_lut;
    int tile_row0, tile_rows;       // full-res 8-row tile rows [tile_row0, tile_row0 + tile_rows) (kj_rtr_render_rows); the whole image: 0, (h + 7) / 8
};
struct RtrTemporalFilterArgs {
    const FrameConstants* fc;
    ImgU32 input_tex; ImgH4 history_tex; ImgF32 depth_tex; ImgU32 ray_len_tex; ImgU2 reprojection_tex; ImgR8 refl_restir_invalidity_tex; ImgU4 gbuffer_tex; ImgH4 output_tex;
    int tile_row0, tile_rows;
};
struct RtrCleanupArgs {
    const FrameConstants* fc;
    ImgH4 input_tex; ImgF32 depth_tex; ImgU32 geometric_normal_tex; ImgU32 output_tex; const int4* spatial_resolve_offsets;
    int tile_row0, tile_rows;
};
hipError_t launch_rtr_resolve(const RtrResolveArgs& a, hipStream_t s);
hipError_t launch_rtr_temporal_filter(const RtrTemporalFilterArgs& a, hipStream_t s);
hipError_t launch_rtr_cleanup(const RtrCleanupArgs& a, hipStream_t s);
