// ORACLE (test infrastructure). Irradiance-cache ray passes restated from
// ircache/{ircache_trace_common.inc,trace_irradiance.rgen,ircache_validate.rgen,trace_accessibility.rgen,
// reset_entry,sum_up_irradiance,prepare_trace_dispatch_args}.hlsl and IrcacheRenderState::{trace_irradiance,
// sum_up_irradiance_for_sampling} (renderers/ircache.rs:360-506).
#pragma once
#include "okj_ircache.hpp"

namespace okj {

struct IrcacheTraceInputs {
    const Scene* scene = nullptr;
    const h4* sky_cube = nullptr; int sky_cube_width = 16;   // convolved cube (world_render_passes.rs:113-121)
    const h4* brdf_fg_lut = nullptr;
};

struct IrcacheTraceResult { f3 incident_radiance, direction, hit_pos; };

// ircache_trace_common.inc.hlsl:37-227 with MAX_PATH_LENGTH = 1
static inline IrcacheTraceResult ircache_trace(Ircache& ic, const FrameConstants& fc, const IrcacheTraceInputs& in, f3 sun_color,
                                               const IrcacheVertex& entry, SampleParams sample_params, uint32_t life) {
    uint32_t rng = sample_params.rng();
    Ray outgoing_ray{entry.position, 0.0f, sample_params.direction(), FLT_MAX};
    IrcacheTraceResult result;
    result.direction = outgoing_ray.d;
    result.hit_pos = mk3(0.0f);
    f3 throughput = mk3(1.0f);
    const float roughness_bias = 0.5f;
    f3 irradiance_sum = mk3(0.0f);
    ic.rays_closest.fetch_add(1, std::memory_order_relaxed);
    const GbufferPathVertex primary_hit = gbuffer_raytrace(*in.scene, fc, outgoing_ray, 1, false, RayCone{0.0f, 0.1f});   // ircache_trace_common.inc.hlsl:83
    if (primary_hit.is_hit) {
        result.hit_pos = primary_hit.position;
        const f3 to_light_norm = sun_direction(fc);
        ic.rays_any.fetch_add(1, std::memory_order_relaxed);
        const bool is_shadowed = in.scene->trace_any(Ray{primary_hit.position, 1e-4f, to_light_norm, FLT_MAX});
        GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
        const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
        const f3 wi = mul(to_light_norm, tangent_to_world);
        f3 wo = mul(-outgoing_ray.d, tangent_to_world);
        if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
        LayeredBrdf brdf = LayeredBrdf::from_gbuffer_ndotv(in.brdf_fg_lut, gbuffer, wo.z);
        brdf.specular_brdf.roughness = lerp(brdf.specular_brdf.roughness, 1.0f, roughness_bias);  // FIREFLY_SUPPRESSION
        const f3 brdf_value = brdf.evaluate_directional_light(wo, wi);
        const f3 light_radiance = is_shadowed ? mk3(0.0f) : sun_color;
        irradiance_sum += throughput * brdf_value * light_radiance * fmaxf(0.0f, wi.z);
        irradiance_sum += gbuffer.emissive * throughput;
        const auto& lights = in.scene->triangle_lights;
        if (fc.triangle_light_count > 0 && !lights.empty()) {
            const float light_selection_pmf = 1.0f / float(fc.triangle_light_count);
            const uint32_t light_idx = hash1_mut(rng) % fc.triangle_light_count;
            f2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const KjTriangleLight& tl = lights[std::min<size_t>(light_idx, lights.size() - 1)];
            f3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
            LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
            const f3 to_light_ws = ls.pos - primary_hit.position;
            const float dist_to_light2 = dot(to_light_ws, to_light_ws);
            const f3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist_to_light2));
            const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist_to_light2;
            if (to_psa_metric > 0.0f) {
                const f3 wi2 = mul(to_light_norm_ws, tangent_to_world);
                ic.rays_any.fetch_add(1, std::memory_order_relaxed);
                const bool sh = in.scene->trace_any(Ray{primary_hit.position, 1e-3f, to_light_norm_ws, sqrtf(dist_to_light2) - 2e-3f});
                if (!sh) irradiance_sum += throughput * f3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf.evaluate(wo, wi2) / ls.pdf * to_psa_metric / light_selection_pmf;
            }
        }
        // SAMPLE_IRCACHE_AT_LAST_VERTEX (IRCACHE_LOOKUP_PRECISE is defined in both ray-gen shaders)
        irradiance_sum += ic.lookup(fc, entry.position, primary_hit.position, gbuffer.normal, 1 + ircache_entry_life_to_rank(life), rng, true) * throughput * gbuffer.albedo;
        // the BRDF sample that would continue the path only affects state that is discarded when MAX_PATH_LENGTH == 1
    } else {
        result.hit_pos = outgoing_ray.o + outgoing_ray.d * 1000.0f;
        irradiance_sum += throughput * xyz(sample_cube_rgba16f(in.sky_cube, in.sky_cube_width, outgoing_ray.d));
    }
    result.incident_radiance = irradiance_sum;
    return result;
}

struct IrcacheTracer {
    // prepare_trace_dispatch_args.hlsl + reset_entry.hlsl
    static void prepare_and_reset(Ircache& ic) {
        const uint32_t alloc_count = ic.meta[META_ALLOC_COUNT];
        ic.meta[META_TRACING_ALLOC_COUNT] = alloc_count;
        for (uint32_t d = 0; d < alloc_count; ++d) {
            const uint32_t entry_idx = ic.entry_indirection[d];
            const f4 i0 = ic.irradiance[entry_idx * 3];
            if (i0.x == 0.0f && i0.y == 0.0f && i0.z == 0.0f && i0.w == 0.0f)
                for (uint32_t i = 0; i < IRCACHE_AUX_STRIDE; ++i) ic.aux[size_t(entry_idx) * IRCACHE_AUX_STRIDE + i] = f4{0, 0, 0, 0};
        }
    }
    // trace_accessibility.rgen.hlsl:21-66
    static void trace_accessibility(Ircache& ic, const IrcacheTraceInputs& in) {
        const uint32_t alloc_count = ic.meta[META_TRACING_ALLOC_COUNT];
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t d = 0; d < int64_t(alloc_count) * IRCACHE_OCTA_DIMS2; ++d) {
            const uint32_t entry_idx = ic.entry_indirection[d / IRCACHE_OCTA_DIMS2];
            const uint32_t octa_idx = uint32_t(d % IRCACHE_OCTA_DIMS2);
            if (!is_ircache_entry_life_valid(ic.life[entry_idx])) continue;
            const IrcacheVertex entry = unpack_vertex(ic.spatial[entry_idx]);
            const size_t output_idx = size_t(entry_idx) * IRCACHE_AUX_STRIDE + octa_idx;
            Reservoir1spp r = Reservoir1spp::from_raw(u2{asuint(ic.aux[output_idx].x), asuint(ic.aux[output_idx].y)});
            const IrcacheVertex prev_entry = unpack_vertex(ic.aux[output_idx + IRCACHE_OCTA_DIMS2 * 2]);
            ic.rays_any.fetch_add(1, std::memory_order_relaxed);
            if (in.scene->trace_any(Ray{entry.position, 0.001f, prev_entry.position - entry.position, 0.999f})) {
                r.M *= 0.8f;
                const u2 raw = r.as_raw();
                ic.aux[output_idx].x = asfloat(raw.x); ic.aux[output_idx].y = asfloat(raw.y);
            }
        }
    }
    // ircache_validate.rgen.hlsl:44-131 (IRCACHE_VALIDATION_SAMPLES_PER_FRAME == IRCACHE_SAMPLES_PER_FRAME => no neighbour pass)
    static void validate(Ircache& ic, const FrameConstants& fc, const IrcacheTraceInputs& in, f3 sun_color) {
        const uint32_t alloc_count = ic.meta[META_TRACING_ALLOC_COUNT];
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t d = 0; d < int64_t(alloc_count) * IRCACHE_VALIDATION_SAMPLES_PER_FRAME; ++d) {
            const uint32_t entry_idx = ic.entry_indirection[d / IRCACHE_VALIDATION_SAMPLES_PER_FRAME];
            const uint32_t sample_idx = uint32_t(d % IRCACHE_VALIDATION_SAMPLES_PER_FRAME);
            const uint32_t life = ic.life[entry_idx];
            const SampleParams sp = SampleParams::from_spf_entry_sample_frame(IRCACHE_VALIDATION_SAMPLES_PER_FRAME, entry_idx, sample_idx, fc.frame_index);
            const size_t output_idx = size_t(entry_idx) * IRCACHE_AUX_STRIDE + sp.octa_idx();
            Reservoir1spp r = Reservoir1spp::from_raw(u2{asuint(ic.aux[output_idx].x), asuint(ic.aux[output_idx].y)});
            if (r.M > 0) {
                f4 prev_value_and_count = ic.aux[output_idx + IRCACHE_OCTA_DIMS2] * f4{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
                const IrcacheVertex prev_entry = unpack_vertex(ic.aux[output_idx + IRCACHE_OCTA_DIMS2 * 2]);
                Ircache::request_key() = (3u << 28) | uint32_t(d);      // deterministic mode: this lookup's place in the frame
                const IrcacheTraceResult prev_traced = ircache_trace(ic, fc, in, sun_color, prev_entry, SampleParams{r.payload}, life);
                const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(prev_traced.direction, prev_entry.normal)));
                const f3 a = prev_traced.incident_radiance * limiter;
                const f3 b = xyz(prev_value_and_count);
                const f3 dist3 = vabs(a - b) / (a + b);
                const float dist = fmaxf(dist3.x, fmaxf(dist3.y, dist3.z));
                const float invalidity = smoothstep(0.1f, 0.5f, dist);
                r.M = fmaxf(0.0f, fminf(r.M, exp2f(log2f(float(IRCACHE_RESTIR_M_CLAMP)) * (1.0f - invalidity))));
                prev_value_and_count.x = a.x; prev_value_and_count.y = a.y; prev_value_and_count.z = a.z;
                const u2 raw = r.as_raw();
                ic.aux[output_idx].x = asfloat(raw.x); ic.aux[output_idx].y = asfloat(raw.y);
                ic.aux[output_idx + IRCACHE_OCTA_DIMS2] = prev_value_and_count;
            }
        }
    }
    // trace_irradiance.rgen.hlsl:44-145
    static void trace_irradiance(Ircache& ic, const FrameConstants& fc, const IrcacheTraceInputs& in, f3 sun_color) {
        const uint32_t alloc_count = ic.meta[META_TRACING_ALLOC_COUNT];
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t d = 0; d < int64_t(alloc_count) * IRCACHE_SAMPLES_PER_FRAME; ++d) {
            const uint32_t entry_idx = ic.entry_indirection[d / IRCACHE_SAMPLES_PER_FRAME];
            const uint32_t sample_idx = uint32_t(d % IRCACHE_SAMPLES_PER_FRAME);
            const uint32_t life = ic.life[entry_idx];
            const f4 packed_entry = ic.spatial[entry_idx];
            const IrcacheVertex entry = unpack_vertex(packed_entry);
            uint32_t rng = hash1(hash1(entry_idx) + fc.frame_index);
            const SampleParams sp = SampleParams::from_spf_entry_sample_frame(IRCACHE_SAMPLES_PER_FRAME, entry_idx, sample_idx, fc.frame_index);
            Ircache::request_key() = (4u << 28) | uint32_t(d);
            const IrcacheTraceResult traced = ircache_trace(ic, fc, in, sun_color, entry, sp, life);
            const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(traced.direction, entry.normal)));
            const f3 new_value = traced.incident_radiance * limiter;
            const float new_lum = sRGB_to_luminance(new_value);
            StreamState stream_state;
            Reservoir1spp reservoir;
            reservoir.init_with_stream(new_lum, 1.0f, stream_state, sp.value);
            const size_t output_idx = size_t(entry_idx) * IRCACHE_AUX_STRIDE + sp.octa_idx();
            const f4 prev_value_and_count = ic.aux[output_idx + IRCACHE_OCTA_DIMS2] * f4{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
            f3 val_sel = new_value;
            bool selected_new = true;
            {
                Reservoir1spp r = Reservoir1spp::from_raw(u2{asuint(ic.aux[output_idx].x), asuint(ic.aux[output_idx].y)});
                if (r.M > 0) {
                    r.M = fminf(r.M, 30.0f);
                    if (reservoir.update_with_stream(r, sRGB_to_luminance(xyz(prev_value_and_count)), 1.0f, stream_state, r.payload, rng)) {
                        val_sel = xyz(prev_value_and_count);
                        selected_new = false;
                    }
                }
            }
            reservoir.finish_stream(stream_state);
            const u2 raw = reservoir.as_raw();
            ic.aux[output_idx].x = asfloat(raw.x); ic.aux[output_idx].y = asfloat(raw.y);
            ic.aux[output_idx + IRCACHE_OCTA_DIMS2] = mk4(val_sel, reservoir.W);
            if (selected_new) ic.aux[output_idx + IRCACHE_OCTA_DIMS2 * 2] = packed_entry;
        }
    }
    // sum_up_irradiance.hlsl:34-89
    static void sum_up(Ircache& ic, const FrameConstants& fc) {
        const uint32_t alloc_count = ic.meta[META_TRACING_ALLOC_COUNT];
        for (uint32_t d = 0; d < alloc_count; ++d) {
            const uint32_t entry_idx = ic.entry_indirection[d];
            f4 sh_rgb[3] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
            float valid_samples = 0;
            for (uint32_t octa_idx = 0; octa_idx < IRCACHE_OCTA_DIMS2; ++octa_idx) {
                const f4 r0 = ic.aux[size_t(entry_idx) * IRCACHE_AUX_STRIDE + octa_idx];
                const f3 dir = SampleParams{asuint(r0.x)}.direction();
                const f4 contrib = ic.aux[size_t(entry_idx) * IRCACHE_AUX_STRIDE + IRCACHE_OCTA_DIMS2 + octa_idx];
                const f3 radiance = xyz(contrib) * contrib.w;
                const f4 sh = mk4(0.282095f, dir.x * 0.488603f, dir.y * 0.488603f, dir.z * 0.488603f) * 4.0f;
                sh_rgb[0] += sh * radiance.x; sh_rgb[1] += sh * radiance.y; sh_rgb[2] += sh * radiance.z;
                valid_samples += contrib.w > 0 ? 1.0f : 0.0f;
            }
            const float scale = 1.0f / fmaxf(1.0f, valid_samples);
            for (uint32_t basis_i = 0; basis_i < 3; ++basis_i) {
                const f4 new_value = sh_rgb[basis_i] * scale;
                f4 prev_value = ic.irradiance[entry_idx * 3 + basis_i] * fc.pre_exposure_delta;
                const bool should_reset = !(prev_value.x != 0.0f || prev_value.y != 0.0f || prev_value.z != 0.0f || prev_value.w != 0.0f);
                if (should_reset) prev_value = new_value;
                ic.irradiance[entry_idx * 3 + basis_i] = lerp(prev_value, new_value, 0.25f);
            }
        }
    }
};

} // namespace okj
