// ORACLE (test infrastructure). LightingRenderer::render_specular restated (renderers/lighting.rs:23-88; shaders/lighting/
// sample_lights.rgen.hlsl, spatial_reuse_lights.hlsl): specular lighting from the triangle lights — one light sample + shadow ray per
// half-res pixel, eight-tap ratio-estimator reuse at full res — ADDED into rtr's resolved image (RENDER_INTO_RTR) so both are filtered
// together (world_render_passes.rs:190-203). Runs only when the scene has triangle lights. Pinned to the two shaders' own text: tests/test_ref_hlsl.py::test_light_specular_reference_hlsl_vs_oracle.
#pragma once
#include "okj_rtr.hpp"

namespace okj {

struct Lighting {
    std::vector<h4> refl0; std::vector<f4> refl1; std::vector<uint32_t> refl2, half_view_normal; std::vector<float> half_depth;
    std::atomic<uint64_t> rays_any{0};

    void render_specular(const FrameConstants& fc, const Scene& scene, ImgU4 gbuffer, ImgR32F depth, const uint8_t* blue_noise, const h4* brdf_fg_lut,
                         const int32_t* spatial_resolve_offsets, Img<uint32_t> output_tex) {
        const int W = depth.w, H = depth.h, hw = (W + 1) / 2, hh = (H + 1) / 2;
        if (fc.triangle_light_count == 0 || scene.triangle_lights.empty()) return;   // world_render_passes.rs:166-170,190
        refl0.assign(size_t(hw) * hh, h4{0, 0, 0, 0}); refl1.assign(size_t(hw) * hh, f4{0, 0, 0, 0}); refl2.assign(size_t(hw) * hh, 0);
        half_view_normal.assign(size_t(hw) * hh, 0); half_depth.assign(size_t(hw) * hh, 0.0f);
        ImgRGBA16F out0(refl0.data(), hw, hh); Img<f4> out1(refl1.data(), hw, hh); ImgU32 out2(refl2.data(), hw, hh);
        ImgU32 hvn(half_view_normal.data(), hw, hh); ImgR32F hd(half_depth.data(), hw, hh);
        const i2 off = halfres_subsample_offset(fc);
        const f4 ts = tex_size4(W, H);
        const uint32_t light_count = std::min<uint32_t>(fc.triangle_light_count, uint32_t(scene.triangle_lights.size()));
        // ---- sample_lights.rgen.hlsl:18-63 (+ the half-res view normal / depth of GbufferDepth)
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                const float d = depth.ld(hx, hy);
                {
                    const f3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(hx, hy).y);
                    hvn.st(x, y, pack_rgba8_snorm(mk4(normalize(xyz(mul44(fc.view_constants.world_to_view, mk4(normal_ws, 0)))), 1.0f)));
                    hd.st(x, y, d);
                }
                if (0.0f == d) { st4(out0, x, y, mk4(0.0f)); continue; }
                const f2 uv = get_uv(float(hx), float(hy), ts);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, d);
                const f3 shadow_ray_origin = vrc.biased_secondary_ray_origin_ws();
                const f4 urand3 = blue_noise_for_pixel(blue_noise, uint32_t(x), uint32_t(y), fc.frame_index);
                const uint32_t light_idx = uint32_t(urand3.z * float(light_count)) % light_count;
                const float light_choice_pmf = 1.0f / float(light_count);
                const KjTriangleLight& tl = scene.triangle_lights[light_idx];
                const f3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
                const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, f2{urand3.x, urand3.y});
                const f3 to_light_ws = ls.pos - shadow_ray_origin;
                const float dist_to_light = length(to_light_ws);
                rays_any.fetch_add(1, std::memory_order_relaxed);
                const bool is_shadowed = scene.trace_any(Ray{shadow_ray_origin, 0.0f, to_light_ws / fmaxf(1e-8f, dist_to_light), dist_to_light - 1e-4f});
                st4(out0, x, y, is_shadowed ? mk4(0.0f, 0.0f, 0.0f, 1.0f) : mk4(tl.radiance[0], tl.radiance[1], tl.radiance[2], 1.0f));
                out1.st(x, y, mk4(vrc.ray_hit_vs() + direction_world_to_view(fc, to_light_ws), ls.pdf * light_choice_pmf));
                out2.st(x, y, pack_rgba8_snorm(mk4(direction_world_to_view(fc, ls.normal), 0.0f)));
            }
        // ---- spatial_reuse_lights.hlsl:33-168
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f2 uv = get_uv(float(x), float(y), ts);
                const float d = depth.ld(x, y);
                if (0.0f == d) continue;
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, d);
                GbufferData g = gbuffer_unpack(gbuffer.ld(x, y));
                g.roughness = fmaxf(g.roughness, 3e-4f);
                const m33 tangent_to_world = build_orthonormal_basis(g.normal);
                f3 wo = mul(-normalize(vrc.ray_dir_ws()), tangent_to_world);
                if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
                const LayeredBrdf lb = LayeredBrdf::from_gbuffer_ndotv(brdf_fg_lut, g, wo.z);
                const f3 energy_preservation_mult = lb.energy_preservation.preintegrated_reflection_mult;
                const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
                const uint32_t filter_idx = 3;
                f4 contrib_accum = mk4(0.0f);
                const f3 normal_vs = direction_world_to_view(fc, g.normal);
                const f3 center_hit_vs = vrc.ray_hit_vs();
                for (uint32_t sample_i = 0; sample_i < 8; ++sample_i) {
                    const int32_t* o = spatial_resolve_offsets + 4 * ((px_idx_in_quad * 16 + sample_i) + 64 * filter_idx);
                    const int sx = x / 2 + o[0], sy = y / 2 + o[1];
                    const float sample_depth = hd.ld(sx, sy);
                    const f4 packed0 = ld4(out0, sx, sy);
                    if (packed0.w != 0.0f && sample_depth != 0.0f) {
                        const f2 sample_uv = get_uv(float(sx * 2 + off.x), float(sy * 2 + off.y), ts);
                        const ViewRayContext src = ViewRayContext::from_uv_and_depth(fc, sample_uv, sample_depth);
                        const f3 sample_origin_vs = src.ray_hit_vs();
                        const f4 packed1 = out1.ld(sx, sy);
                        float neighbor_sampling_pdf = packed1.w;
                        const f3 sample_hit_normal_vs = xyz(unpack_rgba8_snorm(out2.ld(sx, sy)));
                        const f3 center_to_hit_vs = xyz(packed1) - lerp(center_hit_vs, sample_origin_vs, 0.5f);
                        const f3 wi = normalize(mul(direction_view_to_world(fc, center_to_hit_vs), tangent_to_world));
                        const f3 sample_normal_vs = ld_nrm_snorm8(hvn, sx, sy);
                        float rejection_bias = 1.0f;
                        rejection_bias *= saturate((dot(normal_vs, sample_normal_vs) - 0.9f) / (0.999f - 0.9f));
                        rejection_bias *= exp2f(-10.0f * fabsf(d / sample_depth - 1.0f));
                        {
                            const f3 surface_offset = sample_origin_vs - center_hit_vs;
                            const float fraction_of_normal_direction_as_offset = dot(surface_offset, normal_vs) / length(surface_offset);   // 0/0 = NaN for the pixel's own sample: no rejection
                            if (wi.z > 0.0f && wi.z * 0.2f < fraction_of_normal_direction_as_offset) rejection_bias *= sample_i == 0 ? 1.0f : 0.0f;
                        }
                        const BrdfValue spec = lb.specular_brdf.evaluate(wo, wi);
                        const float center_to_hit_dist2 = dot(center_to_hit_vs, center_to_hit_vs);
                        const float to_psa_metric = fmaxf(0.0f, wi.z) * fmaxf(0.0f, dot(sample_hit_normal_vs, -normalize(center_to_hit_vs))) / center_to_hit_dist2;
                        neighbor_sampling_pdf /= to_psa_metric;
                        const f3 contrib_rgb = xyz(packed0) * spec.value * energy_preservation_mult * step(0.0f, wi.z) * (neighbor_sampling_pdf > 0.0f ? 1.0f / neighbor_sampling_pdf : 0.0f);
                        contrib_accum = contrib_accum + mk4(contrib_rgb, 1.0f) * rejection_bias;
                    }
                }
                const f3 out_color = xyz(contrib_accum) / fmaxf(1e-8f, contrib_accum.w);
                output_tex.st(x, y, pack_r11g11b10f(unpack_r11g11b10f(output_tex.ld(x, y)) + out_color));
            }
    }
};

} // namespace okj
