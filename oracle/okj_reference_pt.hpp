// TEST INFRASTRUCTURE (CPU oracle). Restatement of the reference path tracer, the convergence oracle of the GI path:
//   renderers/reference.rs:8-26 (one RGBA32F accumulation image, one ray-gen dispatch per frame)
//   rt/reference_path_trace.rgen.hlsl:75-377
// Compile-time switches of the shader as shipped: FIREFLY_SUPPRESSION, USE_PIXEL_FILTER, USE_SOFT_SHADOWS, USE_LIGHTS,
// USE_EMISSIVE on; FURNACE_TEST, INDIRECT_ONLY, *_FIRST_BOUNCE, RESET/ROLLING_ACCUMULATION off (:29-43).
// `first_bounce_mode` is ours: 0 = as shipped; 1 = the shader's INDIRECT_ONLY (:33,160,196-199); 2 = indirect only through a
// white Lambert first bounce without the specular layer (what rtdgi's irradiance output estimates; used by the convergence test).
#pragma once
#include "okj_scene.hpp"

namespace okj {

static const uint32_t PT_MAX_EYE_PATH_LENGTH = 16;
static const uint32_t PT_RUSSIAN_ROULETTE_START_PATH_LENGTH = 3;

// reference_path_trace.rgen.hlsl:61-73
static inline float pt_inv_error_function(float x, float truncation) {
    const float ALPHA = 0.14f;
    const float INV_ALPHA = 1.0f / ALPHA;
    const float K = 2.0f / (M_PI_F * ALPHA);
    const float y = logf(fmaxf(truncation, 1.0f - x * x));
    const float z = K + 0.5f * y;
    const float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    return sqrtf(fmaxf(0.0f, sqrtf(z * z - y * INV_ALPHA) - z)) * s;
}
static inline float pt_remap_unorm_to_gaussian(float x, float truncation) { return pt_inv_error_function(x * 2.0f - 1.0f, truncation); }

struct ReferencePtInputs {
    const Scene* scene = nullptr;
    const h4* brdf_fg_lut = nullptr;
    int first_bounce_mode = 0;
    uint32_t max_path_length = PT_MAX_EYE_PATH_LENGTH;   // test knob (truncated transport for stage-by-stage checks)
};

// One pixel, one sample: returns false when the sample is rejected (:364, any negative channel).
static inline bool reference_pt_sample(const FrameConstants& fc, const ReferencePtInputs& in, uint32_t pxx, uint32_t pxy, uint32_t W, uint32_t H,
                                       f3& total_radiance, uint64_t* ray_count) {
    uint32_t rng = hash_combine2(hash_combine2(pxx, hash1(pxy)), fc.frame_index);
    float px_off0 = 0.5f, px_off1 = 0.5f;
    const float psf_scale = 0.4f;
    px_off0 += psf_scale * pt_remap_unorm_to_gaussian(uint_to_u01_float(hash1_mut(rng)), 1e-8f);
    px_off1 += psf_scale * pt_remap_unorm_to_gaussian(uint_to_u01_float(hash1_mut(rng)), 1e-8f);
    const f2 uv{(float(pxx) + px_off0) / float(W), (float(pxy) + px_off1) / float(H)};
    const ViewRayContext vrc = ViewRayContext::from_uv(fc, uv);
    Ray outgoing_ray{vrc.ray_origin_ws(), 0.0f, normalize(vrc.ray_dir_ws()), FLT_MAX};
    f3 throughput = mk3(1.0f);
    total_radiance = mk3(0.0f);
    float roughness_bias = 0.0f;
    const f3 sun_color = sun_color_in_direction(fc, sun_direction(fc));
    const bool indirect_only = in.first_bounce_mode != 0;
    RayCone ray_cone = pixel_ray_cone_from_image_height(fc, float(H));   // :123-128
    ray_cone.spread_angle *= 0.3f;

    for (uint32_t path_length = 0; path_length < in.max_path_length; ++path_length) {
        if (path_length == 1) outgoing_ray.tmax = FLT_MAX;
        ++*ray_count;
        const GbufferPathVertex primary_hit = gbuffer_raytrace(*in.scene, fc, outgoing_ray, path_length, false, ray_cone);
        if (!primary_hit.is_hit) {
            total_radiance += throughput * atmosphere_default(fc, outgoing_ray.d, sun_direction(fc));
            break;
        }
        ray_cone = ray_cone.propagate(0.0f, primary_hit.ray_t);   // :151-152
        f2 su;
        su.x = uint_to_u01_float(hash1_mut(rng));
        su.y = uint_to_u01_float(hash1_mut(rng));
        const f3 to_light_norm = sample_sun_direction(fc, su, true);
        bool is_shadowed = true;
        if (!(indirect_only && path_length == 0)) {
            ++*ray_count;
            is_shadowed = in.scene->trace_any(Ray{primary_hit.position, 1e-4f, to_light_norm, FLT_MAX});
        }
        GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
        if (dot(gbuffer.normal, outgoing_ray.d) >= 0.0f) {
            if (path_length == 0) gbuffer.normal = -gbuffer.normal;
            else break;
        }
        if (indirect_only && path_length == 0) { gbuffer.albedo = mk3(1.0f); gbuffer.metalness = 0.0f; }
        const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
        const f3 wi = mul(to_light_norm, tangent_to_world);
        f3 wo = mul(-outgoing_ray.d, tangent_to_world);
        if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
        LayeredBrdf brdf = LayeredBrdf::from_gbuffer_ndotv(in.brdf_fg_lut, gbuffer, wo.z);
        brdf.specular_brdf.roughness = lerp(brdf.specular_brdf.roughness, 1.0f, roughness_bias);  // FIREFLY_SUPPRESSION
        const bool lambert_first = in.first_bounce_mode == 2 && path_length == 0;
        if (!lambert_first) {
            const f3 brdf_value = brdf.evaluate_directional_light(wo, wi);
            const f3 light_radiance = is_shadowed ? mk3(0.0f) : sun_color;
            total_radiance += throughput * brdf_value * light_radiance * fmaxf(0.0f, wi.z);
            total_radiance += gbuffer.emissive * throughput;
        }
        const auto& lights = in.scene->triangle_lights;
        if (!lambert_first && fc.triangle_light_count > 0 && !lights.empty()) {
            const float light_selection_pmf = 1.0f / float(fc.triangle_light_count);
            const uint32_t light_idx = hash1_mut(rng) % fc.triangle_light_count;
            f2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const KjTriangleLight& tl = lights[std::min<size_t>(light_idx, lights.size() - 1)];
            const f3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
            const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
            const f3 to_light_ws = ls.pos - primary_hit.position;
            const float dist_to_light2 = dot(to_light_ws, to_light_ws);
            const f3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist_to_light2));
            const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist_to_light2;
            if (to_psa_metric > 0.0f) {
                const f3 wi2 = mul(to_light_norm_ws, tangent_to_world);
                ++*ray_count;
                const bool sh = in.scene->trace_any(Ray{primary_hit.position, 1e-3f, to_light_norm_ws, sqrtf(dist_to_light2) - 2e-3f});
                if (!sh) total_radiance += throughput * f3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf.evaluate(wo, wi2) / ls.pdf * to_psa_metric / light_selection_pmf;
            }
        }
        f3 urand3;
        urand3.x = uint_to_u01_float(hash1_mut(rng));
        urand3.y = uint_to_u01_float(hash1_mut(rng));
        urand3.z = uint_to_u01_float(hash1_mut(rng));
        BrdfSample brdf_sample;
        if (lambert_first) {
            DiffuseBrdf white{mk3(1.0f)};
            brdf_sample = white.sample(wo, f2{urand3.x, urand3.y});
        } else {
            brdf_sample = brdf.sample(wo, urand3);
        }
        if (!brdf_sample.is_valid()) break;
        roughness_bias = lerp(roughness_bias, 1.0f, 0.5f * brdf_sample.approx_roughness);
        outgoing_ray.o = primary_hit.position;
        outgoing_ray.d = mul(tangent_to_world, brdf_sample.wi);
        outgoing_ray.tmin = 1e-4f;
        throughput = throughput * brdf_sample.value_over_pdf;
        if (path_length >= PT_RUSSIAN_ROULETTE_START_PATH_LENGTH) {
            const float rr_coin = uint_to_u01_float(hash1_mut(rng));
            const float continue_p = fmaxf(gbuffer.albedo.x, fmaxf(gbuffer.albedo.y, gbuffer.albedo.z));
            if (rr_coin > continue_p) break;
            throughput = throughput / continue_p;
        }
    }
    return total_radiance.x >= 0.0f && total_radiance.y >= 0.0f && total_radiance.z >= 0.0f;
}

// One dispatch: accumulate one sample per pixel into `output` (RGBA32F, a = sample count), :79-87,369-375.
static inline uint64_t reference_path_trace(const FrameConstants& fc, const ReferencePtInputs& in, f4* output, uint32_t W, uint32_t H, uint32_t row_begin = 0, uint32_t row_end = ~0u) {
    uint64_t rays_total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays_total)
    for (int y = int(row_begin); y < int(row_end < H ? row_end : H); ++y)      // a band of rows of the W x H frame (tests at 4K): same pixels, same rng
        for (uint32_t x = 0; x < W; ++x) {
            const f4 prev = output[size_t(y) * W + x];
            if (!(prev.w < 1000.0f)) continue;
            f4 cur = mk4(0.0f);
            f3 rad;
            uint64_t rays = 0;
            if (reference_pt_sample(fc, in, x, uint32_t(y), W, H, rad, &rays)) cur = mk4(rad, 1.0f);
            rays_total += rays;
            const float tsc = cur.w + prev.w;
            const float lrp = cur.w / fmaxf(1.0f, tsc);
            const f3 c = xyz(cur) / fmaxf(1.0f, cur.w);
            const f3 o = vmax(mk3(0.0f), lerp(xyz(prev), c, lrp));
            output[size_t(y) * W + x] = mk4(o, fmaxf(1.0f, tsc));
        }
    return rays_total;
}

}  // namespace okj
