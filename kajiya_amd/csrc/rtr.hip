// RtrRenderer for gfx950: ray-traced specular reflections. One HIP kernel per reference pass
// (renderers/rtr.rs:97-480; shaders/rtr/{reflection.rgen, reflection_trace_common.inc, reflection_validate.rgen,
// rtr_restir_temporal, resolve, temporal_filter, spatial_cleanup}.hlsl with rtr_settings.hlsl as checked in), 8x8 pixel tile =
// one wave64, the software BVH of kj_bvh.hpp instead of TraceRay. Host orchestration mirrors RtrRenderer::trace and
// TracedRtr::filter_temporal including the eight ping-pong temporal resources.
//
// Where the reference leaves a result undefined this file picks the same value as the oracle (oracle/okj_rtr.hpp header):
// the validate pass' partially written invalidity image is cleared first, a zero-length history ray is traced along +Z,
// B10G11R11_UFLOAT stores round to nearest through fp16, resolve's sample-shadowing test ignores rounding-residue offsets.
#include "kj_rtr.hpp"
#include "kj_ircache.hpp"
#include "kj_ircache_host.hpp"

namespace kj { SceneView scene_view(const KjScene& s); }
struct FgLut { V3 preintegrated_reflection, preintegrated_reflection_mult; float valid_sample_fraction; };
KJ_D FgLut specular_energy_preservation(const uint2* __restrict__ fg_lut, float roughness, V3 specular_albedo, float ndotv) {   // brdf_lut.hlsl:15-93
    const float s = 63.0f / 64.0f, b = 0.5f / 64.0f;
    const V4 fg = sample_bilinear_clamp_rgba16f(fg_lut, 64, 64, V2{ndotv * s + b, roughness * s + b});
    const V3 single_scatter = specular_albedo * fg.x + fg.y;
    const float e_ss = fg.x + fg.y;
    const V3 f_ss = single_scatter / e_ss;
    const V3 f_ss_tail = lerp(f_ss, v3(1.0f), 0.4f);
    const V3 bounce_radiance = (1.0f - e_ss) * f_ss_tail;
    const V3 mult = 1.0f + bounce_radiance / (1.0f - bounce_radiance);
    return FgLut{single_scatter * mult, mult, fg.z};
}

struct RtrCtx {
    const FrameConstants* __restrict__ fc;
    SceneView sc;
    ImgF32 depth; ImgU4 gbuffer;   // full res
    ImgH4 rtdgi_tex;               // full res
    const uint2* __restrict__ sky_cube; int sky_cube_width;
    const uint2* __restrict__ brdf_fg_lut;
    const float4* __restrict__ sun_color;
    IrcacheView irc; bool has_ircache;
    unsigned long long* __restrict__ ray_counters;
    const uint32_t* __restrict__ ranking_tile; const uint32_t* __restrict__ scrambling_tile; const uint32_t* __restrict__ sobol;
    uint32_t reuse_rtdgi_rays;
    uint32_t request_slot_base, request_key_base, request_stride;   // deferred ircache updates: slot / key of half-res pixel (x, y) = base + y * stride + x
    int tile_row0;                  // first 8-row tile row of the launch (half-res tiles; the validate pass: its own row bounds below)
    int quad_row0, quad_row1;       // the validate pass' rows of 2x2 half-res quads [quad_row0, quad_row1)
};

KJ_D void count_rays(unsigned long long* counters, int which, bool active) {
    const unsigned long long m = __ballot(active);
    if (m != 0ull && (__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(&counter_slot(counters)[which], (unsigned long long)__popcll(m));
}
KJ_D float blue_noise_sampler(const RtrCtx& c, int pixel_i, int pixel_j, int sample_index, int sample_dimension) {   // inc/blue_noise.hlsl:31-60
    pixel_i &= 127; pixel_j &= 127; sample_index &= 255; sample_dimension &= 255;
    const int ranked = sample_index ^ int(c.ranking_tile[sample_dimension + (pixel_i + pixel_j * 128) * 8]);
    int value = int(c.sobol[sample_dimension + ranked * 256]);
    value ^= int(c.scrambling_tile[(sample_dimension % 8) + (pixel_i + pixel_j * 128) * 8]);
    return (0.5f + float(value)) / 256.0f;
}

// ------------------------------------------------------------------ GbufferDepth::half_view_normal / half_depth (renderers/mod.rs:44-71)
__global__ void __launch_bounds__(64) k_rtr_extract_half(const FrameConstants* __restrict__ fcp, ImgU4 gbuffer, ImgF32 depth, ImgU32 half_view_normal, ImgF32 half_depth, int tile_row0) {
    TILE_XY_ROWS(half_depth.w, half_depth.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int sx = x * 2 + off.x, sy = y * 2 + off.y;
    const V3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(sx, sy).y);
    const V3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, v4(normal_ws, 0))));
    half_view_normal.st(x, y, pack_rgba8_snorm(v4(normal_vs, 1.0f)));
    half_depth.st(x, y, depth.ld(sx, sy));
}

// ------------------------------------------------------------------ reflection_trace_common.inc.hlsl:56-257
struct RtrTraceResult { V3 total_radiance; float hit_t; V3 hit_normal_vs; };
KJ_D RtrTraceResult rtr_trace_ray(const RtrCtx& c, float roughness, uint32_t& rng, V3 ray_o, V3 ray_d, uint32_t* stack, uint32_t request_pixel) {
    const FrameConstants& fc = *c.fc;
    const float roughness_bias = roughness;
    const float reflected_cone_spread_angle = sqrtf(roughness) * 0.05f;
    const RayCone ray_cone = pixel_ray_cone_from_image_height(fc, float(c.depth.h)).propagate(reflected_cone_spread_angle, length(ray_o - get_eye_position(fc)));
    count_rays(c.ray_counters, 0, true);
    const GbufferPathVertex primary_hit = gbuffer_raytrace<false>(c.sc, fc, ray_o, ray_d, 0.0f, SKY_DIST, 1, false, stack, 64, nullptr, ray_cone);
    if (primary_hit.is_hit) {
        GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
        gbuffer.roughness = lerp(gbuffer.roughness, 1.0f, roughness_bias);
        const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
        const V3 wo = to_local(tangent_to_world, -ray_d);
        const LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(c.brdf_fg_lut, gbuffer, wo.z);
        const V4 gts = tex_size4(c.depth.w, c.depth.h);
        const V3 hit_cs = position_world_to_sample(fc, primary_hit.position);
        const V2 hit_uv = cs_to_uv(V2{hit_cs.x, hit_cs.y});
        const float screen_depth = sample_nearest_clamp(c.depth, hit_uv);
        const uint4 screen_gbuffer = c.gbuffer.ld(int(hit_uv.x * gts.x), int(hit_uv.y * gts.y));
        const V3 screen_normal_ws = unpack_normal_11_10_11(screen_gbuffer.y);
        const bool is_on_screen = fabsf(hit_cs.x) < 1.0f && fabsf(hit_cs.y) < 1.0f && inverse_depth_relative_diff(hit_cs.z, screen_depth) < 5e-3f &&
                                  dot(screen_normal_ws, -ray_d) > 0.0f && dot(screen_normal_ws, gbuffer.normal) > 0.7f;
        V3 total_radiance = v3(0.0f);
        {
            V2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const float4 sc4 = *c.sun_color;
            const V3 sun_radiance{sc4.x, sc4.y, sc4.z};
            if (sun_radiance.x != 0 || sun_radiance.y != 0 || sun_radiance.z != 0) {
                const V3 to_light_norm = sample_sun_direction(fc, urand, true);
                count_rays(c.ray_counters, 1, true);
                const bool is_shadowed = rt_is_shadowed<false>(c.sc, primary_hit.position, to_light_norm, 1e-4f, SKY_DIST, stack, 64, nullptr);
                const V3 wi = to_local(tangent_to_world, to_light_norm);
                const V3 brdf_value = layered_brdf_evaluate(brdf, wo, wi) * fmaxf(0.0f, wi.z);
                total_radiance += brdf_value * (is_shadowed ? v3(0.0f) : sun_radiance);
            }
        }
        const V3 reflected_normal_vs = direction_world_to_view(fc, gbuffer.normal);
        total_radiance += gbuffer.emissive;
        if (is_on_screen) {
            const V3 reprojected_radiance = xyz(unpack_rgba16f(sample_nearest_clamp(c.rtdgi_tex, hit_uv))) * fc.pre_exposure_delta;
            total_radiance += reprojected_radiance * gbuffer.albedo;
        } else {
            V2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const uint32_t nl = min(fc.triangle_light_count, c.sc.light_count);
            for (uint32_t li = 0; li < nl; ++li) {
                const KjTriangleLight tl = c.sc.lights[li];
                const V3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
                const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
                const V3 to_light_ws = ls.pos - primary_hit.position;
                const float dist2 = dot(to_light_ws, to_light_ws);
                const V3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist2));
                const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist2;
                if (to_psa_metric > 0.0f) {
                    count_rays(c.ray_counters, 1, true);
                    const bool is_shadowed = rt_is_shadowed<false>(c.sc, primary_hit.position, to_light_norm_ws, 1e-4f, sqrtf(dist2) - 2e-4f, stack, 64, nullptr);
                    const V3 bounce_albedo = lerp(gbuffer.albedo, v3(1.0f), 0.04f);
                    const V3 brdf_value = bounce_albedo * to_psa_metric / KJ_PI;
                    if (!is_shadowed) total_radiance += V3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf_value / ls.pdf;
                }
            }
            if (c.has_ircache) {
                const float cone_width = ray_cone.propagate(0.0f, primary_hit.ray_t).width;
                const V3 gi = ircache_lookup<false, true>(c.irc, fc, ray_o, primary_hit.position, gbuffer.normal, 1u, rng, cone_width < 0.1f, c.request_slot_base + request_pixel, c.request_key_base | request_pixel);
                total_radiance += gi * gbuffer.albedo;
            }
        }
        return RtrTraceResult{total_radiance, primary_hit.ray_t, reflected_normal_vs};
    }
    const V3 far_gi = xyz(sample_cube_rgba16f(c.sky_cube, c.sky_cube_width, ray_d));
    return RtrTraceResult{far_gi, SKY_DIST, -direction_world_to_view(fc, ray_d)};
}

// ------------------------------------------------------------------ reflection.rgen.hlsl:45-169
// waves per SIMD the two ray kernels are compiled for (rtdgi's fused ray kernels: 5, rtdgi.hip)
#ifndef KJ_RTR_WAVES
#define KJ_RTR_WAVES 4
#endif
__global__ void __launch_bounds__(64, KJ_RTR_WAVES) k_rtr_trace(RtrCtx c, ImgH4 out0_tex, ImgH4 out1_tex, ImgU32 out2_tex, ImgU32 rng_out_tex) {
    extern __shared__ uint32_t lds_stack[];
    const int tile_row0 = c.tile_row0;
    TILE_XY(out0_tex.w, out0_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *c.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = c.depth.ld(hx, hy);
    if (0.0f == depth) { st4(out0_tex, x, y, V4{0.0f, 0.0f, 0.0f, -SKY_DIST}); return; }
    const V4 gts = tex_size4(c.depth.w, c.depth.h);
    const V2 uv = get_uv(float(hx), float(hy), gts);
    GbufferData gbuffer = gbuffer_unpack(c.gbuffer.ld(hx, hy));
    gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
    if (c.reuse_rtdgi_rays && gbuffer.roughness > 0.6f) return;
    const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
    const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
    const V3 refl_ray_origin_ws = vr.biased_secondary_ray_origin_ws_with_normal(gbuffer.normal);
    V3 wo = to_local(tangent_to_world, -vr.dir_ws);
    if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
    const V3 spec_albedo = lerp(v3(0.04f), gbuffer.albedo, gbuffer.metalness);
    const uint32_t noise_offset = fc.frame_index;
    uint32_t rng = hash3(uint32_t(x), uint32_t(y), noise_offset);
    V2 urand{blue_noise_sampler(c, x, y, int(noise_offset), 0), blue_noise_sampler(c, x, y, int(noise_offset), 1)};
    urand.x = lerp(urand.x, 0.0f, RTR_SAMPLING_BIAS);
    BrdfSample brdf_sample = specular_sample(gbuffer.roughness, spec_albedo, wo, urand);
    for (uint32_t retry_i = 0; retry_i < 4u && !(brdf_sample.wi.z > 1e-6f); ++retry_i) {
        urand.x = uint_to_u01_float(hash1_mut(rng));
        urand.y = uint_to_u01_float(hash1_mut(rng));
        urand.x = lerp(urand.x, 0.0f, RTR_SAMPLING_BIAS);
        brdf_sample = specular_sample(gbuffer.roughness, spec_albedo, wo, urand);
    }
    if (brdf_sample.wi.z > 1e-6f) {
        const float cos_theta = normalize(wo + brdf_sample.wi).z;
        const V3 ray_d = to_world(tangent_to_world, brdf_sample.wi);
        rng_out_tex.st(x, y, rng);
        const RtrTraceResult result = rtr_trace_ray(c, gbuffer.roughness, rng, refl_ray_origin_ws, ray_d, lds_stack + lane, uint32_t(y) * c.request_stride + uint32_t(x));
        const V3 hit_offset_ws = ray_d * result.hit_t;
        const FgLut brdf_lut = specular_energy_preservation(c.brdf_fg_lut, gbuffer.roughness, spec_albedo, wo.z);
        const float pdf = brdf_sample.pdf / brdf_lut.valid_sample_fraction;
        st4(out0_tex, x, y, v4(result.total_radiance, 1.0f - cos_theta));
        st4(out1_tex, x, y, v4(hit_offset_ws, pdf));
        out2_tex.st(x, y, pack_rgba8_snorm(v4(result.hit_normal_vs, 0.0f)));
    } else {
        st4(out0_tex, x, y, V4{1.0f, 0.0f, 1.0f, 0.0f});
        st4(out1_tex, x, y, v4(0.0f));
    }
}

// ------------------------------------------------------------------ reflection_validate.rgen.hlsl:42-146 (one thread per 2x2 half-res quad)
__global__ void __launch_bounds__(64, KJ_RTR_WAVES) k_rtr_validate(RtrCtx c, ImgF4 ray_orig_history_tex, ImgH4 ray_history_tex, ImgU32 rng_history_tex, ImgH4 irradiance_history_tex,
                                                      ImgU2 reservoir_history_tex, ImgR8 refl_restir_invalidity_tex, int qw, int qh) {
    extern __shared__ uint32_t lds_stack[];
    const int lane = threadIdx.x;
    const uint2 tb = tile_order<KJ_TILES_PLAIN>();
    const int qx = int(tb.x) * 8 + (lane & 7), qy = (int(tb.y) + c.tile_row0) * 8 + (lane >> 3);
    if (qx >= qw || qy >= qh || qy < c.quad_row0 || qy >= c.quad_row1) return;
    const FrameConstants& fc = *c.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int x = qx * 2 + off.x, y = qy * 2 + off.y;
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = c.depth.ld(hx, hy);
    if (0.0f == depth) { refl_restir_invalidity_tex.st(x, y, to_unorm8(1.0f)); return; }
    GbufferData gbuffer = gbuffer_unpack(c.gbuffer.ld(hx, hy));
    gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
    const float4 ro = ray_orig_history_tex.ld(x, y);
    const V3 ray_orig_ws = V3{ro.x, ro.y, ro.z} + get_prev_eye_position(fc);
    const V3 ray_hit_ws = xyz(ld4(ray_history_tex, x, y)) + ray_orig_ws;
    const V3 d = ray_hit_ws - ray_orig_ws;
    const float dl = length(d);
    const V3 ray_d = dl > 0.0f ? d / dl : V3{0, 0, 1};
    uint32_t rng = rng_history_tex.ld(x, y);
    const RtrTraceResult result = rtr_trace_ray(c, gbuffer.roughness, rng, ray_orig_ws, ray_d, lds_stack + lane, uint32_t(y) * c.request_stride + uint32_t(x));
    Reservoir1spp r = Reservoir1spp::from_raw(reservoir_history_tex.ld(x, y));
    const V4 prev_irradiance_packed = ld4(irradiance_history_tex, x, y);
    const V3 prev_irradiance = vmax(v3(0.0f), xyz(prev_irradiance_packed) * fc.pre_exposure_delta);
    const V3 check_radiance = vmax(v3(0.0f), result.total_radiance);
    const float rad_diff = length(vabs(prev_irradiance - check_radiance) / vmax(v3(1e-3f), prev_irradiance + check_radiance));
    const float invalidity = smoothstep(0.1f, 0.5f, rad_diff / length(v3(1.0f)));
    r.M *= 1.0f - invalidity;
    st4(irradiance_history_tex, x, y, v4(check_radiance, prev_irradiance_packed.w));
    refl_restir_invalidity_tex.st(x, y, to_unorm8(invalidity));
    reservoir_history_tex.st(x, y, r.as_raw());
    for (uint32_t i = 1; i <= 3u; ++i) {
        const I2 o = hi_px_subpixel(fc.frame_index + i);
        const int nx = qx * 2 + o.x, ny = qy * 2 + o.y;
        const V4 neighbor_prev_irradiance_packed = ld4(irradiance_history_tex, nx, ny);
        const V3 a = vmax(v3(0.0f), xyz(neighbor_prev_irradiance_packed) * fc.pre_exposure_delta);
        const V3 b = prev_irradiance;
        const float neigh_rad_diff = length(vabs(a - b) / vmax(v3(1e-8f), a + b));
        if (neigh_rad_diff < 0.2f) st4(irradiance_history_tex, nx, ny, v4(check_radiance, neighbor_prev_irradiance_packed.w));
        refl_restir_invalidity_tex.st(nx, ny, to_unorm8(invalidity));
        if (invalidity > 0.0f) {
            Reservoir1spp rn = Reservoir1spp::from_raw(reservoir_history_tex.ld(nx, ny));
            rn.M *= 1.0f - invalidity;
            reservoir_history_tex.st(nx, ny, rn.as_raw());
        }
    }
}

// ------------------------------------------------------------------ rtr_restir_temporal.hlsl:105-153
KJ_D void find_best_reprojection_in_neighborhood(const FrameConstants& fc, const ImgF4& ray_orig_history_tex, V4 gts, V2 base_px, I2& best_px, V3 refl_ray_origin_ws, bool wide) {
    float best_dist = 1e10f;
    const V2 clip_scale{fc.view_constants.clip_to_view[0], fc.view_constants.clip_to_view[5]};
    const V2 offset_scale{1.0f * -2.0f * clip_scale.x * gts.z, -1.0f * -2.0f * clip_scale.y * gts.w};
    const V3 look_direction = direction_view_to_world(fc, V3{0, 0, -1});
    const I2 off = halfres_subsample_offset(fc.frame_index);
    {
        const float z_offset = dot(look_direction, refl_ray_origin_ws - get_eye_position(fc));
        refl_ray_origin_ws += direction_view_to_world(fc, V3{float(off.x) * offset_scale.x * z_offset, float(off.y) * offset_scale.y * z_offset, 0.0f});
    }
    const int start_coord = wide ? -1 : 0;
    for (int yy = start_coord; yy <= 1; ++yy)
        for (int xx = start_coord; xx <= 1; ++xx) {
            const I2 spx{int(floorf(base_px.x + float(xx))), int(floorf(base_px.y + float(yy)))};
            const RtrRestirRayOrigin ray_orig = ray_origin_from_raw(ray_orig_history_tex.ld(spx.x, spx.y));
            V3 orig = ray_orig.ray_origin_eye_offset_ws + get_prev_eye_position(fc);
            const I2 orig_jitter = hi_px_subpixel(ray_orig.frame_index_mod4);
            {
                const float z_offset = dot(look_direction, orig);
                orig += direction_view_to_world(fc, V3{float(orig_jitter.x) * offset_scale.x * z_offset, float(orig_jitter.y) * offset_scale.y * z_offset, 0.0f});
            }
            const float d = length(orig - refl_ray_origin_ws);
            if (d < best_dist) { best_dist = d; best_px = spx; }
        }
}

struct RtrTemporalArgs {
    const FrameConstants* fc;
    ImgU4 gbuffer_tex; ImgU32 half_view_normal_tex; ImgF32 depth_tex;
    ImgH4 candidate0_tex, candidate1_tex; ImgU32 candidate2_tex;
    ImgH4 irradiance_history_tex; ImgF4 ray_orig_history_tex; ImgH4 ray_history_tex; ImgU32 rng_history_tex; ImgU2 reservoir_history_tex;
    ImgU2 reprojection_tex; ImgH4 hit_normal_history_tex;
    ImgH4 irradiance_out_tex; ImgF4 ray_orig_output_tex; ImgH4 ray_output_tex; ImgU32 rng_output_tex; ImgH4 hit_normal_output_tex; ImgU2 reservoir_out_tex;
    int tile_row0;
};
// ------------------------------------------------------------------ rtr_restir_temporal.hlsl:155-533
__global__ void __launch_bounds__(64) k_rtr_restir_temporal(RtrTemporalArgs a) {
    const int tile_row0 = a.tile_row0;
    TILE_XY(a.irradiance_out_tex.w, a.irradiance_out_tex.h)
    const FrameConstants& fc = *a.fc;
    // the five taps' spiral directions depend on the tap and the frame only: evaluated once per workgroup with the reference's cosf / sinf
    __shared__ float tap_cos[5], tap_sin[5];
    if (lane < 5) {
        const float ang_offset_t = float(((fc.frame_index + 7u) * 11u) % 32u) * KJ_TAU;
        const float ang_t = (float(lane) + ang_offset_t) * KJ_GOLDEN_ANGLE;
        tap_cos[lane] = cosf(ang_t); tap_sin[lane] = sinf(ang_t);
    }
    __syncthreads();
    if (!in_image) return;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = a.depth_tex.ld(hx, hy);
    if (0.0f == depth) {
        st4(a.irradiance_out_tex, x, y, V4{0.0f, 0.0f, 0.0f, -SKY_DIST});
        st4(a.hit_normal_output_tex, x, y, v4(0.0f));
        a.reservoir_out_tex.st(x, y, make_uint2(0, 0));
        return;
    }
    const V4 gts = tex_size4(a.depth_tex.w, a.depth_tex.h);
    const V2 uv = get_uv(float(hx), float(hy), gts);
    const V3 normal_vs = ld_nrm_snorm8(a.half_view_normal_tex, x, y);
    const V3 normal_ws = direction_view_to_world(fc, normal_vs);
    float local_normal_flatness = 1.0f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) local_normal_flatness *= saturate(dot(normal_vs, ld_nrm_snorm8(a.half_view_normal_tex, x + dx, y + dy)));
    float reprojection_neighborhood_stability = 1.0f;
    for (int dy = 0; dy <= 1; ++dy)
        for (int dx = 0; dx <= 1; ++dx) reprojection_neighborhood_stability *= ld_reproj(a.reprojection_tex, x * 2 + dx, y * 2 + dy).z;
    const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
    const V3 refl_ray_origin_ws = vr.biased_secondary_ray_origin_ws_with_normal(normal_ws);
    const V3 refl_ray_origin_vs = position_world_to_view(fc, refl_ray_origin_ws);
    V3 outgoing_dir{0, 0, 1};
    uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index);
    const GbufferData gbuffer = gbuffer_unpack(a.gbuffer_tex.ld(hx, hy));
    const float a2 = fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness) * fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness);
    float pdf_sel = 0.0f, cos_theta = 0.0f;
    V3 irradiance_sel = v3(0.0f);
    float4 ray_orig_sel = make_float4(0, 0, 0, 0);
    V3 ray_hit_sel_ws = v3(1.0f), hit_normal_sel = v3(1.0f);
    uint32_t rng_sel = a.rng_output_tex.ld(x, y);
    StreamState stream_state{0.0f, 0.0f};
    Reservoir1spp reservoir = Reservoir1spp::create();
    const uint32_t reservoir_payload = uint32_t(x) | (uint32_t(y) << 16);
    reservoir.payload = reservoir_payload;
    {
        const V4 hit0 = ld4(a.candidate0_tex, x, y), hit1 = ld4(a.candidate1_tex, x, y);
        const V3 hit2 = xyz(unpack_rgba8_snorm(a.candidate2_tex.ld(x, y)));
        const V3 out_value = xyz(hit0);
        const float pdf = fminf(hit1.w, RTR_RESTIR_MAX_PDF_CLAMP);
        const V3 hit_vs = xyz(hit1);
        if (pdf > 0.0f) {
            outgoing_dir = normalize(hit_vs);
            const float p_q = fmaxf(1e-3f, sRGB_to_luminance(out_value)) * pdf;
            const float inv_pdf_q = 1.0f / pdf;
            pdf_sel = pdf;
            cos_theta = 1.0f - hit0.w;
            irradiance_sel = out_value;
            ray_orig_sel = ray_origin_to_raw(refl_ray_origin_ws, gbuffer.roughness, fc.frame_index & 3u);
            ray_hit_sel_ws = hit_vs + refl_ray_origin_ws;
            hit_normal_sel = direction_view_to_world(fc, hit2);
            if (p_q * inv_pdf_q > 0.0f) reservoir.init_with_stream(p_q, inv_pdf_q, stream_state, reservoir_payload);
        }
    }
    const V4 center_reproj = ld_reproj(a.reprojection_tex, hx, hy);
    {
        const uint32_t sample_count = center_reproj.z < 1.0f ? 5u : 1u;
        const V3 prev_eye = get_prev_eye_position(fc);
        for (uint32_t sample_i = 0; sample_i < sample_count && stream_state.M_sum < RTR_RESTIR_TEMPORAL_M_CLAMP; ++sample_i) {
            const float rpx_offset_radius = sqrtf(float(((sample_i - 1u) + fc.frame_index) & 3u) + 1.0f) * clampf(8.0f - stream_state.M_sum, 1.0f, 7.0f);
            const V2 reservoir_px_offset_base{tap_cos[sample_i] * rpx_offset_radius, tap_sin[sample_i] * rpx_offset_radius};
            const I2 rpx_offset = sample_i == 0 ? I2{0, 0} : I2{int(reservoir_px_offset_base.x), int(reservoir_px_offset_base.y)};
            const V4 reproj = ld_reproj(a.reprojection_tex, hx + rpx_offset.x * 2, hy + rpx_offset.y * 2);
            const V2 base_px{float(x) + gts.x * reproj.x / 2.0f, float(y) + gts.y * reproj.y / 2.0f};
            I2 best_px{int(floorf(base_px.x + 0.5f)), int(floorf(base_px.y + 0.5f))};
            if (reprojection_neighborhood_stability >= 1.0f) {
                if (fabsf(gts.x * reproj.x) > 0.1f || fabsf(gts.y * reproj.y) > 0.1f)
                    find_best_reprojection_in_neighborhood(fc, a.ray_orig_history_tex, gts, base_px, best_px, refl_ray_origin_ws, false);
            } else {
                find_best_reprojection_in_neighborhood(fc, a.ray_orig_history_tex, gts, base_px, best_px, refl_ray_origin_ws, true);
            }
            const I2 rpx{best_px.x + rpx_offset.x, best_px.y + rpx_offset.y};
            Reservoir1spp r = Reservoir1spp::from_raw(a.reservoir_history_tex.ld(rpx.x, rpx.y));
            const int spx = int(r.payload & 0xffffu), spy = int(r.payload >> 16);
            float4 prev_ray_orig_and_roughness = a.ray_orig_history_tex.ld(spx, spy);
            prev_ray_orig_and_roughness.x += prev_eye.x; prev_ray_orig_and_roughness.y += prev_eye.y; prev_ray_orig_and_roughness.z += prev_eye.z;   // .w: packed bits, untouched
            const V3 prev_orig{prev_ray_orig_and_roughness.x, prev_ray_orig_and_roughness.y, prev_ray_orig_and_roughness.z};
            const V3 od = refl_ray_origin_ws - prev_orig;
            if (dot(od, od) > 0.05f * refl_ray_origin_vs.z * refl_ray_origin_vs.z) continue;
            const V4 prev_irrad_raw = ld4(a.irradiance_history_tex, spx, spy);
            const V3 prev_irrad = xyz(prev_irrad_raw) * fc.pre_exposure_delta;
            const float prev_cos_theta = 1.0f - prev_irrad_raw.w;
            const V4 sample_hit_ws_and_pdf_packed = ld4(a.ray_history_tex, spx, spy);
            const float prev_pdf = sample_hit_ws_and_pdf_packed.w;
            const V3 sample_hit_ws = xyz(sample_hit_ws_and_pdf_packed) + prev_orig;
            const float prev_dist = length(xyz(sample_hit_ws_and_pdf_packed));
            const V4 hn_raw = ld4(a.hit_normal_history_tex, spx, spy);
            const V4 sample_hit_normal_ws_dot{hn_raw.x * 2.0f - 1.0f, hn_raw.y * 2.0f - 1.0f, hn_raw.z * 2.0f - 1.0f, hn_raw.w};
            const V3 dir_to_sample_hit_unnorm = sample_hit_ws - refl_ray_origin_ws;
            const float dist_to_sample_hit = length(dir_to_sample_hit_unnorm);
            const V3 dir_to_sample_hit = normalize(dir_to_sample_hit_unnorm);
            r.M = fminf(r.M, RTR_RESTIR_TEMPORAL_M_CLAMP);
            {
                const V3 current_wo = normalize(vr.hit_ws - get_eye_position(fc));
                const V3 prev_wo = normalize(vr.hit_ws - prev_eye);
                const float wo_dot = saturate(dot(current_wo, prev_wo));
                float wo_similarity = saturate(ggx_ndf_0_1(fmaxf(3e-5f, a2), wo_dot));      // ^64 by squaring (libm powf: 163 VALU instructions)
                wo_similarity *= wo_similarity; wo_similarity *= wo_similarity; wo_similarity *= wo_similarity;
                wo_similarity *= wo_similarity; wo_similarity *= wo_similarity; wo_similarity *= wo_similarity;
                float mult = lerp(wo_similarity, 1.0f, smoothstep(0.05f, 0.5f, sqrtf(gbuffer.roughness)));
                mult = lerp(1.0f, mult, local_normal_flatness);
                r.M *= mult;
            }
            float p_q = 1.0f;
            p_q *= fmaxf(1e-3f, sRGB_to_luminance(prev_irrad));
            p_q *= stepf(0.0f, dot(dir_to_sample_hit, normal_ws));
            p_q *= prev_pdf;
            float jacobian = 1.0f;
            jacobian *= clampf(prev_dist / dist_to_sample_hit, 1e-4f, 1e4f);
            jacobian *= jacobian;
            jacobian *= fmaxf(0.0f, -dot(xyz(sample_hit_normal_ws_dot), dir_to_sample_hit)) / fmaxf(1e-5f, sample_hit_normal_ws_dot.w);
            {
                const float threshold = lerp(1.1f, 4.0f, gbuffer.roughness * gbuffer.roughness);
                if (!(jacobian < threshold && jacobian > 1.0f / threshold)) continue;
            }
            p_q *= jacobian;
            if (reservoir.update_with_stream(r, p_q, 1.0f, stream_state, reservoir_payload, rng)) {
                outgoing_dir = dir_to_sample_hit;
                pdf_sel = prev_pdf;
                cos_theta = prev_cos_theta;
                irradiance_sel = prev_irrad;
                ray_orig_sel = prev_ray_orig_and_roughness;
                ray_hit_sel_ws = sample_hit_ws;
                hit_normal_sel = xyz(sample_hit_normal_ws_dot);
                rng_sel = a.rng_history_tex.ld(spx, spy);
            }
        }
        reservoir.finish_stream(stream_state);
        reservoir.W = fminf(reservoir.W, 1e20f);
    }
    const V4 hit_normal_ws_dot = v4(hit_normal_sel, -dot(hit_normal_sel, outgoing_dir));
    st4(a.irradiance_out_tex, x, y, v4(irradiance_sel, 1.0f - cos_theta));
    const V3 eye = get_eye_position(fc);
    a.ray_orig_output_tex.st(x, y, make_float4(ray_orig_sel.x - eye.x, ray_orig_sel.y - eye.y, ray_orig_sel.z - eye.z, ray_orig_sel.w));
    st4(a.hit_normal_output_tex, x, y, V4{hit_normal_ws_dot.x * 0.5f + 0.5f, hit_normal_ws_dot.y * 0.5f + 0.5f, hit_normal_ws_dot.z * 0.5f + 0.5f, hit_normal_ws_dot.w});
    st4(a.ray_output_tex, x, y, v4(ray_hit_sel_ws - V3{ray_orig_sel.x, ray_orig_sel.y, ray_orig_sel.z}, pdf_sel));
    a.rng_output_tex.st(x, y, rng_sel);
    a.reservoir_out_tex.st(x, y, reservoir.as_raw());
}

// ------------------------------------------------------------------ LightingRenderer::render_specular (renderers/lighting.rs:23-88)
// sample_lights.rgen.hlsl:18-63: one triangle-light sample + shadow ray per half-res pixel
__global__ void __launch_bounds__(64) k_lighting_sample_lights(const FrameConstants* __restrict__ fcp, SceneView sc, ImgF32 depth_tex, const uint32_t* __restrict__ blue_noise,
                                                                ImgH4 out0_tex, ImgF4 out1_tex, ImgU32 out2_tex, unsigned long long* ray_counters, int tile_row0) {
    extern __shared__ uint32_t lds_stack[];
    TILE_XY(out0_tex.w, out0_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = depth_tex.ld(hx, hy);
    if (0.0f == depth) { st4(out0_tex, x, y, v4(0.0f)); return; }
    const V2 uv = get_uv(float(hx), float(hy), tex_size4(depth_tex.w, depth_tex.h));
    const ViewRay vr = view_ray_from_uv_and_depth(fc, uv, depth);
    const V3 shadow_ray_origin = vr.biased_secondary_ray_origin_ws();
    const V4 urand3 = blue_noise_for_pixel(blue_noise, uint32_t(x), uint32_t(y), fc.frame_index);
    const uint32_t light_count = min(fc.triangle_light_count, sc.light_count);
    const uint32_t light_idx = uint32_t(urand3.z * float(light_count)) % light_count;
    const float light_choice_pmf = 1.0f / float(light_count);
    const KjTriangleLight tl = sc.lights[light_idx];
    const V3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
    const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, V2{urand3.x, urand3.y});
    const V3 to_light_ws = ls.pos - shadow_ray_origin;
    const float dist_to_light = length(to_light_ws);
    count_rays(ray_counters, 1, true);
    const bool is_shadowed = rt_is_shadowed<false>(sc, shadow_ray_origin, to_light_ws / fmaxf(1e-8f, dist_to_light), 0.0f, dist_to_light - 1e-4f, lds_stack + lane, 64, nullptr);
    st4(out0_tex, x, y, is_shadowed ? V4{0.0f, 0.0f, 0.0f, 1.0f} : V4{tl.radiance[0], tl.radiance[1], tl.radiance[2], 1.0f});
    const V3 hit_vs = vr.hit_vs + direction_world_to_view(fc, to_light_ws);
    out1_tex.st(x, y, make_float4(hit_vs.x, hit_vs.y, hit_vs.z, ls.pdf * light_choice_pmf));
    out2_tex.st(x, y, pack_rgba8_snorm(v4(direction_world_to_view(fc, ls.normal), 0.0f)));
}
// spatial_reuse_lights.hlsl:33-168 (RENDER_INTO_RTR: the result is added to rtr's resolved image)
__global__ void __launch_bounds__(64) k_lighting_spatial_reuse(const FrameConstants* __restrict__ fcp, ImgU4 gbuffer_tex, ImgF32 depth_tex, ImgH4 hit0_tex, ImgF4 hit1_tex, ImgU32 hit2_tex,
                                                                ImgU32 half_view_normal_tex, ImgF32 half_depth_tex, ImgU32 output_tex, const int4* __restrict__ spatial_resolve_offsets,
                                                                const uint2* __restrict__ brdf_fg_lut, int tile_row0) {
    TILE_XY(output_tex.w, output_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const V4 ts = tex_size4(output_tex.w, output_tex.h);
    const float depth = depth_tex.ld(x, y);
    if (0.0f == depth) return;
    const ViewRay vr = view_ray_from_uv_and_depth(fc, get_uv(float(x), float(y), ts), depth);
    GbufferData g = gbuffer_unpack(gbuffer_tex.ld(x, y));
    g.roughness = fmaxf(g.roughness, 3e-4f);
    const Basis tangent_to_world = build_orthonormal_basis(g.normal);
    V3 wo = to_local(tangent_to_world, -vr.dir_ws);
    if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
    const LayeredBrdf lb = layered_brdf_from_gbuffer_ndotv(brdf_fg_lut, g, wo.z);
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
    V4 contrib_accum = v4(0.0f);
    const V3 normal_vs = direction_world_to_view(fc, g.normal);
    for (uint32_t sample_i = 0; sample_i < 8u; ++sample_i) {
        const int4 o = spatial_resolve_offsets[(px_idx_in_quad * 16u + sample_i) + 64u * 3u];
        const int sx = x / 2 + o.x, sy = y / 2 + o.y;
        const float sample_depth = half_depth_tex.ld(sx, sy);
        const V4 packed0 = ld4(hit0_tex, sx, sy);
        if (packed0.w != 0.0f && sample_depth != 0.0f) {
            const ViewRay sr = view_ray_from_uv_and_depth(fc, get_uv(float(sx * 2 + off.x), float(sy * 2 + off.y), ts), sample_depth);
            const V3 sample_origin_vs = sr.hit_vs;
            const float4 packed1 = hit1_tex.ld(sx, sy);
            float neighbor_sampling_pdf = packed1.w;
            const V3 sample_hit_normal_vs = xyz(unpack_rgba8_snorm(hit2_tex.ld(sx, sy)));
            const V3 center_to_hit_vs = V3{packed1.x, packed1.y, packed1.z} - lerp(vr.hit_vs, sample_origin_vs, 0.5f);
            const V3 wi = normalize(to_local(tangent_to_world, direction_view_to_world(fc, center_to_hit_vs)));
            const V3 sample_normal_vs = ld_nrm_snorm8(half_view_normal_tex, sx, sy);
            float rejection_bias = 1.0f;
            rejection_bias *= saturate((dot(normal_vs, sample_normal_vs) - 0.9f) / (0.999f - 0.9f));
            rejection_bias *= exp2f(-10.0f * fabsf(depth / sample_depth - 1.0f));
            {
                const V3 surface_offset = sample_origin_vs - vr.hit_vs;
                const float fraction_of_normal_direction_as_offset = dot(surface_offset, normal_vs) / length(surface_offset);   // 0/0 = NaN for the pixel's own sample: no rejection
                if (wi.z > 0.0f && wi.z * 0.2f < fraction_of_normal_direction_as_offset) rejection_bias *= sample_i == 0u ? 1.0f : 0.0f;
            }
            const BrdfValue spec = specular_evaluate(lb.roughness, lb.spec_albedo, wo, wi);
            const float center_to_hit_dist2 = dot(center_to_hit_vs, center_to_hit_vs);
            const float to_psa_metric = fmaxf(0.0f, wi.z) * fmaxf(0.0f, dot(sample_hit_normal_vs, -normalize(center_to_hit_vs))) / center_to_hit_dist2;
            neighbor_sampling_pdf /= to_psa_metric;
            const V3 contrib_rgb = xyz(packed0) * spec.value * lb.preintegrated_reflection_mult * stepf(0.0f, wi.z) * (neighbor_sampling_pdf > 0.0f ? 1.0f / neighbor_sampling_pdf : 0.0f);
            contrib_accum = contrib_accum + v4(contrib_rgb, 1.0f) * rejection_bias;
        }
    }
    const V3 out_color = xyz(contrib_accum) / fmaxf(1e-8f, contrib_accum.w);
    output_tex.st(x, y, pack_r11g11b10f(unpack_r11g11b10f(output_tex.ld(x, y)) + out_color));
}

// ------------------------------------------------------------------ host
struct KjRtr {
    KjDevice* dev = nullptr;
    bool reuse_rtdgi_rays = true;                       // rtr.rs:32,70
    int W = 0, H = 0, hw = 0, hh = 0;
    std::map<std::string, kj::DevBuf> surf;
    bool flip[8] = {false, false, false, false, false, false, false, false};
    kj::DevBuf ranking, scrambling, sobol, offsets, ray_counters;
    // TracedRtr (rtr.rs:74-80)
    void *resolved_tex = nullptr, *temporal_output_tex = nullptr, *history_tex = nullptr, *ray_len_tex = nullptr, *refl_restir_invalidity_tex = nullptr;
    hipError_t err = hipSuccess;
    // reflection trace and reflection validate are independent (new candidates vs. last frame's reservoirs) and both latency-bound:
    // validate runs on a side stream, forked after the frame's inputs are ready and joined before the temporal pass.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ~KjRtr() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }

    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
    void pingpong(const char* key, int idx, size_t bytes, hipStream_t s, void*& output, void*& history) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx]) std::swap(a, b);
        output = get(a, bytes, s);
        history = get(b, bytes, s);
        flip[idx] = !flip[idx];
    }
    void resize(int W_, int H_) {
        if (W == W_ && H == H_) return;
        W = W_; H = H_; hw = (W + 1) / 2; hh = (H + 1) / 2;
        surf.clear();
    }
};

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_rtr_create(KjDevice* dev, const KjRtrTables* t, KjRtr** out) {
    KJ_REQUIRE(dev && t && out, "null argument");
    KJ_REQUIRE(t->ranking_tile && t->scrambling_tile && t->sobol && t->spatial_resolve_offsets, "all four tables are required");
    KjRtr* r = new KjRtr();
    r->dev = dev;
    hipError_t e = r->ranking.upload(t->ranking_tile, 128 * 128 * 8 * 4);
    if (e == hipSuccess) e = r->scrambling.upload(t->scrambling_tile, 128 * 128 * 8 * 4);
    if (e == hipSuccess) e = r->sobol.upload(t->sobol, 256 * 256 * 4);
    if (e == hipSuccess) e = r->offsets.upload(t->spatial_resolve_offsets, 16 * 4 * 8 * 16);
    if (e == hipSuccess) e = r->ray_counters.alloc(KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // the tables are host memory owned by the caller
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) { delete r; set_last_error("kj_rtr_create: %s", hipGetErrorString(e)); return KJ_ERR_HIP; }
    *out = r;
    return KJ_OK;
}
void kj_rtr_destroy(KjRtr* r) { delete r; }
KjStatus kj_rtr_set_options(KjRtr* r, uint32_t reuse_rtdgi_rays) {
    KJ_REQUIRE(r, "null argument");
    r->reuse_rtdgi_rays = reuse_rtdgi_rays != 0;
    return KJ_OK;
}

static KjStatus rtr_check_params(KjRtr* r, const KjRtrParams* p) {
    KJ_REQUIRE(r && p, "null argument");
    KJ_REQUIRE(p->scene && p->reprojection_map && p->sky_cube && p->rtdgi_irradiance && p->candidate_radiance_tex && p->candidate_hit_tex && p->candidate_normal_tex &&
               p->gbuffer_depth.depth && p->gbuffer_depth.gbuffer && p->gbuffer_depth.geometric_normal, "missing input");
    KJ_REQUIRE(p->gbuffer_depth.width > 0 && p->gbuffer_depth.height > 0 && p->gbuffer_depth.width < 65536 && p->gbuffer_depth.height < 65536, "bad extent");
    KJ_REQUIRE(r->dev->fc_dev, "kj_frame_begin not called");
    if (!p->scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    return KJ_OK;
}

} // extern "C"

// Full-res rows [row_begin, row_end) -> tile rows of the full-res and half-res launches and the validate pass' quad rows. row_begin is a multiple of 16
// (or the whole image): half-res 8x8 tiles and the validate pass' 2x2 quads never straddle a strip's edge.
struct RtrRows { int f0, fn, h0, hn, hrow0, hrow1, q0, q1; bool whole; };
static RtrRows rtr_rows(const KjRtr* r, uint32_t row_begin, uint32_t row_end) {
    RtrRows o;
    o.whole = row_begin == 0u && int(row_end) == r->H;
    o.f0 = int(row_begin) / 8; o.fn = (int(row_end) + 7) / 8 - o.f0;
    o.hrow0 = int(row_begin) / 2; o.hrow1 = int(row_end) == r->H ? r->hh : int(row_end) / 2;
    o.h0 = o.hrow0 / 8; o.hn = (o.hrow1 + 7) / 8 - o.h0;
    o.q0 = o.hrow0 / 2; o.q1 = (o.hrow1 + 1) / 2;
    return o;
}

static KjStatus rtr_trace_rows(KjRtr* r, const KjRtrParams* p, uint32_t row_begin, uint32_t row_end, bool extract_half, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    const int W = r->W, H = r->H, hw = r->hw, hh = r->hh;
    const FrameConstants* fc = r->dev->fc_dev;
    const uint32_t mask = p->pass_mask;
    if (mask & KJ_RTR_PASS_KEEP) for (bool& f : r->flip) f = !f;
    const RtrRows rows = rtr_rows(r, row_begin, row_end);
    const dim3 gh((hw + 7) / 8, rows.hn), gf((W + 7) / 8, rows.fn), blk(64);
    const size_t HB = size_t(hw) * hh, FB = size_t(W) * H;
    const ImgU4 gbuffer = img<uint4>(p->gbuffer_depth.gbuffer, W, H);
    const ImgF32 depth = img<float>(p->gbuffer_depth.depth, W, H);
    const ImgU2 reprojection = img<uint2>(p->reprojection_map, W, H);
    const ImgH4 refl0 = img<uint2>(p->candidate_radiance_tex, hw, hh), refl1 = img<uint2>(p->candidate_hit_tex, hw, hh);
    const ImgU32 refl2 = img<uint32_t>(p->candidate_normal_tex, hw, hh);

    void *rng_out, *rng_hist;               r->pingpong("rtr.rng", 6, HB * 4, s, rng_out, rng_hist);
    void *ray_orig_out, *ray_orig_hist;     r->pingpong("rtr.ray_orig", 3, HB * 16, s, ray_orig_out, ray_orig_hist);
    void* invalidity = r->get("refl_restir_invalidity_tex", HB, s);
    void *hit_normal_out, *hit_normal_hist; r->pingpong("rtr.hit_normal", 7, HB * 8, s, hit_normal_out, hit_normal_hist);
    void *irradiance_out, *irradiance_hist; r->pingpong("rtr.irradiance", 2, HB * 8, s, irradiance_out, irradiance_hist);
    void *reservoir_out, *reservoir_hist;   r->pingpong("rtr.reservoir", 5, HB * 8, s, reservoir_out, reservoir_hist);
    void *ray_out, *ray_hist;               r->pingpong("rtr.ray", 4, HB * 8, s, ray_out, ray_hist);
    void* resolved = r->get("resolved_tex", FB * 4, s);
    void *temporal_out, *temporal_hist;     r->pingpong("rtr.temporal", 0, FB * 8, s, temporal_out, temporal_hist);
    void *ray_len_out, *ray_len_hist;       r->pingpong("rtr.ray_len", 1, FB * 4, s, ray_len_out, ray_len_hist);
    void* half_view_normal = r->get("half_view_normal_tex", HB * 4, s);
    void* half_depth = r->get("half_depth_tex", HB * 4, s);
    KJ_TRY_HIP(r->err);
    if (!(mask & KJ_RTR_PASS_KEEP)) KJ_TRY_HIP(hipMemsetAsync(r->ray_counters.p, 0, KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8, s));

    RtrCtx c;
    c.fc = fc;
    c.sc = scene_view(*p->scene);
    c.depth = depth; c.gbuffer = gbuffer;
    c.rtdgi_tex = img<uint2>(p->rtdgi_irradiance, W, H);
    c.sky_cube = (const uint2*)p->sky_cube; c.sky_cube_width = int(p->sky_cube_width);
    c.brdf_fg_lut = (const uint2*)r->dev->brdf_fg_lut.p;
    c.sun_color = (const float4*)r->dev->sun_color.p + r->dev->fc_slot;
    c.has_ircache = p->ircache != nullptr;
    if (p->ircache) { KJ_REQUIRE(!p->ircache->pending_irradiance_sum, "ircache sum-up pending (ircache.rs:67 assert)"); c.irc = p->ircache->view(); } else { memset(&c.irc, 0, sizeof(c.irc)); }
    c.ray_counters = (unsigned long long*)r->ray_counters.p;
    c.ranking_tile = (const uint32_t*)r->ranking.p; c.scrambling_tile = (const uint32_t*)r->scrambling.p; c.sobol = (const uint32_t*)r->sobol.p;
    c.reuse_rtdgi_rays = r->reuse_rtdgi_rays ? 1u : 0u;
    c.request_stride = uint32_t(hw); c.request_slot_base = 0; c.request_key_base = 0;
    c.tile_row0 = rows.h0; c.quad_row0 = rows.q0; c.quad_row1 = rows.q1;
    if (p->ircache && p->ircache->deferred) {      // the lookups of the two ray passes record into slot ranges of their own (kj_ircache_set_rtr_requests)
        KJ_REQUIRE(p->ircache->rtr_requests && p->ircache->req_half_pixels == uint32_t(hw) * uint32_t(hh),
                   "a cache in deferred-update mode needs kj_ircache_set_rtr_requests(cache, 1) before kj_ircache_begin_requests (this frame's half-res extent)");
        c.request_slot_base = p->ircache->rtr_request_base() + uint32_t(hw) * uint32_t(hh); c.request_key_base = 6u << 28;      // the trace pass; validate re-bases below
    }
    const size_t trace_lds = size_t(c.sc.bvh.stack_entries) * 64 * 4;
    KJ_REQUIRE(trace_lds <= 64 * 1024, "BVH too deep for the LDS traversal stack");

    const bool fork = (mask & KJ_RTR_PASS_TRACE) && (mask & KJ_RTR_PASS_VALIDATE);
    hipStream_t sv = fork ? r->side : s;
    if (fork) { KJ_TRY_HIP(hipEventRecord(r->ev_fork, s)); KJ_TRY_HIP(hipStreamWaitEvent(r->side, r->ev_fork, 0)); }
    if (mask & KJ_RTR_PASS_VALIDATE) {
        KJ_TRY_HIP(hipMemsetAsync((uint8_t*)invalidity + size_t(rows.hrow0) * hw, 0, size_t(rows.hrow1 - rows.hrow0) * hw, sv));
        const int qw = (hw + 1) / 2, qh = (hh + 1) / 2;
        RtrCtx vc = c;
        vc.tile_row0 = rows.q0 / 8;
        if (p->ircache && p->ircache->deferred) { vc.request_slot_base = p->ircache->rtr_request_base(); vc.request_key_base = 5u << 28; }
        hipLaunchKernelGGL(k_rtr_validate, dim3((qw + 7) / 8, (rows.q1 + 7) / 8 - rows.q0 / 8), blk, trace_lds, sv, vc, img<float4>(ray_orig_hist, hw, hh), img<uint2>(ray_hist, hw, hh), img<uint32_t>(rng_hist, hw, hh),
                           img<uint2>(irradiance_hist, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint8_t>(invalidity, hw, hh), qw, qh);
        KJ_CHECK_LAUNCH();
    }
    if (fork) KJ_TRY_HIP(hipEventRecord(r->ev_join, r->side));
    if (extract_half) {      // the whole frame whatever the rows: the resolve reads the view normal wherever its taps land
        hipLaunchKernelGGL(k_rtr_extract_half, dim3((hw + 7) / 8, (hh + 7) / 8), blk, 0, s, fc, gbuffer, depth, img<uint32_t>(half_view_normal, hw, hh), img<float>(half_depth, hw, hh), 0);
        KJ_CHECK_LAUNCH();
    }
    if (mask & KJ_RTR_PASS_TRACE) {
        hipLaunchKernelGGL(k_rtr_trace, gh, blk, trace_lds, s, c, refl0, refl1, refl2, img<uint32_t>(rng_out, hw, hh));
        KJ_CHECK_LAUNCH();
    }
    if (fork) KJ_TRY_HIP(hipStreamWaitEvent(s, r->ev_join, 0));
    if (mask & KJ_RTR_PASS_RESTIR_TEMPORAL) {
        RtrTemporalArgs a;
        a.fc = fc; a.gbuffer_tex = gbuffer; a.half_view_normal_tex = img<uint32_t>(half_view_normal, hw, hh); a.depth_tex = depth;
        a.candidate0_tex = refl0; a.candidate1_tex = refl1; a.candidate2_tex = refl2;
        a.irradiance_history_tex = img<uint2>(irradiance_hist, hw, hh); a.ray_orig_history_tex = img<float4>(ray_orig_hist, hw, hh); a.ray_history_tex = img<uint2>(ray_hist, hw, hh);
        a.rng_history_tex = img<uint32_t>(rng_hist, hw, hh); a.reservoir_history_tex = img<uint2>(reservoir_hist, hw, hh);
        a.reprojection_tex = reprojection; a.hit_normal_history_tex = img<uint2>(hit_normal_hist, hw, hh);
        a.irradiance_out_tex = img<uint2>(irradiance_out, hw, hh); a.ray_orig_output_tex = img<float4>(ray_orig_out, hw, hh); a.ray_output_tex = img<uint2>(ray_out, hw, hh);
        a.rng_output_tex = img<uint32_t>(rng_out, hw, hh); a.hit_normal_output_tex = img<uint2>(hit_normal_out, hw, hh); a.reservoir_out_tex = img<uint2>(reservoir_out, hw, hh);
        a.tile_row0 = rows.h0;
        hipLaunchKernelGGL(k_rtr_restir_temporal, gh, blk, 0, s, a);
        KJ_CHECK_LAUNCH();
    }
    if (mask & KJ_RTR_PASS_RESOLVE) {
        RtrResolveArgs a;
        a.fc = fc; a.gbuffer_tex = gbuffer; a.depth_tex = depth; a.hit1_tex = refl1; a.reprojection_tex = reprojection;
        a.half_view_normal_tex = img<uint32_t>(half_view_normal, hw, hh); a.ray_len_history_tex = img<uint32_t>(ray_len_hist, W, H);
        a.restir_irradiance_tex = img<uint2>(irradiance_out, hw, hh); a.restir_ray_tex = img<uint2>(ray_out, hw, hh); a.restir_reservoir_tex = img<uint2>(reservoir_out, hw, hh);
        a.restir_ray_orig_tex = img<float4>(ray_orig_out, hw, hh);
        a.output_tex = img<uint32_t>(resolved, W, H); a.ray_len_output_tex = img<uint32_t>(ray_len_out, W, H);
        a.blue_noise = (const uint32_t*)r->dev->blue_noise.p; a.brdf_fg_lut = (const uint2*)r->dev->brdf_fg_lut.p;
        a.tile_row0 = rows.f0; a.tile_rows = rows.fn;
        KJ_TRY_HIP(launch_rtr_resolve(a, s));
    }
    r->resolved_tex = resolved; r->temporal_output_tex = temporal_out; r->history_tex = temporal_hist; r->ray_len_tex = ray_len_out; r->refl_restir_invalidity_tex = invalidity;
    return KJ_OK;
}

// sample_lights on half-res rows grown by 8 either side of the strip (spatial_reuse_lights taps reach <= 6 half-res rows with rtr.rs's offset table), reuse on the rows
static KjStatus rtr_specular_lights_rows(KjRtr* r, const KjRtrParams* p, uint32_t row_begin, uint32_t row_end, void* stream_) {
    if (p->scene->light_count == 0 || r->dev->fc_host.triangle_light_count == 0) return KJ_OK;   // world_render_passes.rs:166-170,190: only with triangle lights
    hipStream_t s = (hipStream_t)stream_;
    const int W = r->W, H = r->H, hw = r->hw, hh = r->hh;
    const FrameConstants* fc = r->dev->fc_dev;
    const size_t HB = size_t(hw) * hh;
    void* l0 = r->get("lighting.refl0_tex", HB * 8, s);
    void* l1 = r->get("lighting.refl1_tex", HB * 16, s);
    void* l2 = r->get("lighting.refl2_tex", HB * 4, s);
    void* half_view_normal = r->get("half_view_normal_tex", HB * 4, s);
    void* half_depth = r->get("half_depth_tex", HB * 4, s);
    KJ_TRY_HIP(r->err);
    const SceneView sc = scene_view(*p->scene);
    const size_t trace_lds = size_t(sc.bvh.stack_entries) * 64 * 4;
    const ImgF32 depth = img<float>(p->gbuffer_depth.depth, W, H);
    const RtrRows rows = rtr_rows(r, row_begin, row_end);
    const int sh0 = rows.whole ? 0 : std::max(0, rows.h0 - 1), sh1 = rows.whole ? (hh + 7) / 8 : std::min((hh + 7) / 8, rows.h0 + rows.hn + 1);
    const dim3 blk(64);
    hipLaunchKernelGGL(k_lighting_sample_lights, dim3((hw + 7) / 8, sh1 - sh0), blk, trace_lds, s, fc, sc, depth, (const uint32_t*)r->dev->blue_noise.p, img<uint2>(l0, hw, hh), img<float4>(l1, hw, hh),
                       img<uint32_t>(l2, hw, hh), (unsigned long long*)r->ray_counters.p, sh0);
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_lighting_spatial_reuse, dim3((W + 7) / 8, rows.fn), blk, 0, s, fc, img<uint4>(p->gbuffer_depth.gbuffer, W, H), depth, img<uint2>(l0, hw, hh), img<float4>(l1, hw, hh), img<uint32_t>(l2, hw, hh),
                       img<uint32_t>(half_view_normal, hw, hh), img<float>(half_depth, hw, hh), img<uint32_t>(r->resolved_tex, W, H), (const int4*)r->offsets.p,
                       (const uint2*)r->dev->brdf_fg_lut.p, rows.f0);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

static KjStatus rtr_filter_rows(KjRtr* r, const KjRtrParams* p, uint32_t row_begin, uint32_t row_end, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    const int W = r->W, H = r->H, hw = r->hw, hh = r->hh;
    const FrameConstants* fc = r->dev->fc_dev;
    const RtrRows rows = rtr_rows(r, row_begin, row_end);
    const ImgF32 depth = img<float>(p->gbuffer_depth.depth, W, H);
    if (p->pass_mask & KJ_RTR_PASS_TEMPORAL_FILTER) {
        const RtrTemporalFilterArgs a{fc, img<uint32_t>(r->resolved_tex, W, H), img<uint2>(r->history_tex, W, H), depth, img<uint32_t>(r->ray_len_tex, W, H), img<uint2>(p->reprojection_map, W, H),
                                      img<uint8_t>(r->refl_restir_invalidity_tex, hw, hh), img<uint4>(p->gbuffer_depth.gbuffer, W, H), img<uint2>(r->temporal_output_tex, W, H), rows.f0, rows.fn};
        KJ_TRY_HIP(launch_rtr_temporal_filter(a, s));
    }
    if (p->pass_mask & KJ_RTR_PASS_CLEANUP) {
        const RtrCleanupArgs a{fc, img<uint2>(r->temporal_output_tex, W, H), depth, img<uint32_t>(p->gbuffer_depth.geometric_normal, W, H), img<uint32_t>(r->resolved_tex, W, H), (const int4*)r->offsets.p,
                               rows.f0, rows.fn};
        KJ_TRY_HIP(launch_rtr_cleanup(a, s));
    }
    return KJ_OK;
}

extern "C" {

KjStatus kj_rtr_trace(KjRtr* r, const KjRtrParams* p, void* stream_) {
    if (KjStatus st = rtr_check_params(r, p)) return st;
    r->resize(int(p->gbuffer_depth.width), int(p->gbuffer_depth.height));
    return rtr_trace_rows(r, p, 0u, uint32_t(r->H), true, stream_);
}

KjStatus kj_rtr_render_specular_lights(KjRtr* r, const KjRtrParams* p, void* stream_) {
    if (KjStatus st = rtr_check_params(r, p)) return st;
    KJ_REQUIRE(r->resolved_tex && int(p->gbuffer_depth.width) == r->W && int(p->gbuffer_depth.height) == r->H, "kj_rtr_trace must run first with the same extent");
    return rtr_specular_lights_rows(r, p, 0u, uint32_t(r->H), stream_);
}

KjStatus kj_rtr_filter_temporal(KjRtr* r, const KjRtrParams* p, const void** out_resolved, void* stream_) {
    if (KjStatus st = rtr_check_params(r, p)) return st;
    KJ_REQUIRE(r->resolved_tex && int(p->gbuffer_depth.width) == r->W && int(p->gbuffer_depth.height) == r->H, "kj_rtr_trace must run first with the same extent");
    if (KjStatus st = rtr_filter_rows(r, p, 0u, uint32_t(r->H), stream_)) return st;
    if (out_resolved) *out_resolved = r->resolved_tex;
    return KJ_OK;
}

// The screen-tile split's entry point (kj_split_rtr_frame, multigpu.py: SplitRtdgi.rtr_frame): the passes of params->pass_mask on full-res rows
// [row_begin, row_end) -- the ray passes, the reservoir pass and the specular lights on the half-res rows underneath. What a pass reads beyond the rows is the
// caller's to provide (DESIGN 7 lists every reach). KJ_RTR_PASS_EXTRACT_HALF: the half-res view normal / depth of the WHOLE frame (inputs are replicated;
// the resolve reads the normal wherever its taps land). Without KJ_RTR_PASS_KEEP the call opens the frame (ping-pong flip, ray counters cleared).
KjStatus kj_rtr_render_rows(KjRtr* r, const KjRtrParams* p, uint32_t row_begin, uint32_t row_end, const void** out_resolved, void* stream_) {
    if (KjStatus st = rtr_check_params(r, p)) return st;
    KJ_REQUIRE(row_begin < row_end && row_end <= p->gbuffer_depth.height && (row_begin % 16u) == 0u && (row_end % 16u == 0u || row_end == p->gbuffer_depth.height),
               "rows must be a non-empty range cut on 16-row boundaries");
    r->resize(int(p->gbuffer_depth.width), int(p->gbuffer_depth.height));
    const uint32_t mask = p->pass_mask;
    if (mask & (KJ_RTR_PASS_VALIDATE | KJ_RTR_PASS_TRACE | KJ_RTR_PASS_RESTIR_TEMPORAL | KJ_RTR_PASS_RESOLVE | KJ_RTR_PASS_EXTRACT_HALF) || !(mask & KJ_RTR_PASS_KEEP))
        if (KjStatus st = rtr_trace_rows(r, p, row_begin, row_end, (mask & KJ_RTR_PASS_EXTRACT_HALF) != 0, stream_)) return st;
    KJ_REQUIRE(r->resolved_tex, "the frame has not been opened (a call without KJ_RTR_PASS_KEEP comes first)");
    if (mask & KJ_RTR_PASS_SPECULAR_LIGHTS) if (KjStatus st = rtr_specular_lights_rows(r, p, row_begin, row_end, stream_)) return st;
    if (mask & (KJ_RTR_PASS_TEMPORAL_FILTER | KJ_RTR_PASS_CLEANUP)) if (KjStatus st = rtr_filter_rows(r, p, row_begin, row_end, stream_)) return st;
    if (out_resolved) *out_resolved = r->resolved_tex;
    return KJ_OK;
}

KjStatus kj_rtr_surface(KjRtr* r, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(r && name && out_dev_ptr && out_bytes, "null argument");
    auto it = r->surf.find(name);
    KJ_REQUIRE(it != r->surf.end(), "unknown surface");
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}

KjStatus kj_rtr_ray_counts(KjRtr* r, uint64_t* out_closest, uint64_t* out_any) {
    KJ_REQUIRE(r && out_closest && out_any, "null argument");
    std::vector<uint64_t> h(KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE);
    KJ_TRY_HIP(hipMemcpy(h.data(), r->ray_counters.p, h.size() * 8, hipMemcpyDeviceToHost));
    uint64_t a = 0, b = 0;
    for (uint32_t i = 0; i < KJ_COUNTER_SLOTS; ++i) { a += h[i * KJ_COUNTER_STRIDE]; b += h[i * KJ_COUNTER_STRIDE + 1]; }
    *out_closest = a; *out_any = b;
    return KJ_OK;
}

} // extern "C"
