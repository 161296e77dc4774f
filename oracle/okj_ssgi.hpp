// ORACLE (test infrastructure). SsgiRenderer restated from crates/lib/kajiya/src/renderers/ssgi.rs:25-181 and
// assets/shaders/ssgi/{ssgi,spatial_filter,upsample,temporal_filter}.hlsl with the shipped switches
// (USE_AO_ONLY 1, USE_SSGI_FACING_CORRECTION 1, SSGI_HALF_SAMPLE_COUNT 6, kernel radius 60 px, no random jitter).
// With USE_AO_ONLY the colour accumulation (fetch_lighting / prev_radiance_tex / facing correction) never reaches the
// output, so only the horizon search is restated. Output = the R8_UNORM "ssao" guide every rtdgi filter consumes.
#pragma once
#include "okj_passes.hpp"

namespace okj {

typedef Img<uint16_t> ImgR16Fs;

// [Drobot2014a] / [Eberly2014] approximations used by ssgi.hlsl:51-62
static inline float ssgi_fast_sqrt(float x) { return asfloat(0x1fbd1df5u + (asuint(x) >> 1u)); }
static inline float ssgi_fast_acos(float in_x) {
    const float x = fabsf(in_x);
    float res = -0.156583f * x + 1.57079632679f;
    res *= ssgi_fast_sqrt(1.0f - x);
    return in_x >= 0 ? res : 3.14159265359f - res;
}
static inline float ssgi_integrate_arc(float h1, float h2, float n) {
    const float a = -cosf(2.0f * h1 - n) + cosf(n) + 2.0f * h1 * sinf(n);
    const float b = -cosf(2.0f * h2 - n) + cosf(n) + 2.0f * h2 * sinf(n);
    return 0.25f * (a + b);
}
static inline float ssgi_update_horizon(float prev, float cur, float blend) { return cur > prev ? lerp(prev, cur, blend) : prev; }

struct Ssgi {
    std::map<std::string, std::vector<uint8_t>> surf;
    bool flip = false;
    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }

    // ssgi.hlsl:174-222 with the colour path removed
    static float process_sample(const FrameConstants& fc, f4 sample_cs, f3 center_vs, f3 v_vs, float kernel_radius_ws, float theta_cos_max) {
        if (sample_cs.z > 0) {
            const f4 sample_vs4 = mul44(fc.view_constants.sample_to_view, sample_cs);
            const f3 sample_vs = xyz(sample_vs4) / sample_vs4.w;
            const f3 off = sample_vs - center_vs;
            const float len = length(off);
            const float sample_theta_cos = dot(off, v_vs) / len;
            const float dn = len / kernel_radius_ws;
            if (dn < 1.0f) theta_cos_max = ssgi_update_horizon(theta_cos_max, sample_theta_cos, smoothstep(1.0f, 0.0f, dn));
        } else {
            theta_cos_max = ssgi_update_horizon(theta_cos_max, -1.0f, 1.0f);
        }
        return theta_cos_max;
    }

    // SsgiRenderer::render (ssgi.rs:25-81) + filter_ssgi (:83-149). Returns the R8 full-res guide.
    ImgR8 render(const FrameConstants& fc, ImgU4 gbuffer, ImgR32F depth, ImgRGBA16S reprojection_tex) {
        const int W = depth.w, H = depth.h, hw = (W + 1) / 2, hh = (H + 1) / 2;
        // GbufferDepth::half_view_normal / half_depth (renderers/mod.rs:31-71)
        ImgU32 half_view_normal = get<uint32_t>("half_view_normal_tex", hw, hh);
        ImgR32F half_depth = get<float>("half_depth_tex", hw, hh);
        {
            const i2 off = halfres_subsample_offset(fc);
            for (int y = 0; y < hh; ++y)
                for (int x = 0; x < hw; ++x) {
                    const int sx = x * 2 + off.x, sy = y * 2 + off.y;
                    const f3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(sx, sy).y);
                    const f3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, mk4(normal_ws, 0))));
                    half_view_normal.st(x, y, pack_rgba8_snorm(mk4(normal_vs, 1.0f)));
                    half_depth.st(x, y, depth.ld(sx, sy));
                }
        }
        // ---- "ssao" (ssgi.hlsl:230-341), half res, R16F
        ImgR16Fs ssgi_tex = get<uint16_t>("ssgi_tex", hw, hh);
        const f4 input_tex_size = tex_size4(W, H), output_tex_size = tex_size4(hw, hh);
        static const float temporal_rotations[6] = {60.0f, 300.0f, 180.0f, 240.0f, 120.0f, 0.0f};
        static const float temporal_offsets[4] = {0.0f, 0.5f, 0.25f, 0.75f};
        const float* s2v = fc.view_constants.sample_to_view;   // column-major: M[r][c] = s2v[c * 4 + r]
#pragma omp parallel for schedule(static)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const f2 uv = get_uv(float(x), float(y), output_tex_size);
                const float d = half_depth.ld(x, y);
                if (d == 0.0f) { ssgi_tex.st(x, y, f32_to_f16(0.0f)); continue; }
                const GbufferData g = gbuffer_unpack(gbuffer.ld(x * 2, y * 2));
                const f3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, mk4(g.normal, 0))));
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, d);
                const f3 v_vs = -normalize(vrc.ray_dir_vs());
                const f4 ray_hit_cs = vrc.ray_hit_cs;
                const f3 ray_hit_vs = vrc.ray_hit_vs();
                const uint32_t ux = uint32_t(x), uy = uint32_t(y);
                const float spatial_direction_noise = 1.0f / 16.0f * float((((ux + uy) & 3u) << 2) + (ux & 3u));
                const float temporal_direction_noise = temporal_rotations[fc.frame_index % 6] / 360.0f;
                const float spatial_offset_noise = (1.0f / 4.0f) * float((uy - ux) & 3u);
                const float temporal_offset_noise = temporal_offsets[fc.frame_index / 6 % 4];
                const float ss_angle = frac(spatial_direction_noise + temporal_direction_noise) * 3.14159265359f;
                const float rand_offset = frac(spatial_offset_noise + temporal_offset_noise);
                f2 cs_slice_dir{cosf(ss_angle) * input_tex_size.y / input_tex_size.x, sinf(ss_angle)};
                float kernel_radius_ws, kernel_radius_shrinkage;
                {
                    const float ws_to_cs = 0.5f / -ray_hit_vs.z * fc.view_constants.view_to_clip[5];
                    const float cs_kernel_radius_scaled = 60.0f * output_tex_size.w;
                    kernel_radius_ws = cs_kernel_radius_scaled / ws_to_cs;
                    cs_slice_dir = cs_slice_dir * cs_kernel_radius_scaled;
                    kernel_radius_shrinkage = fminf(1.0f, 0.4f / cs_kernel_radius_scaled);
                }
                cs_slice_dir = cs_slice_dir * kernel_radius_shrinkage;
                kernel_radius_ws *= kernel_radius_shrinkage;
                const f3 center_vs = ray_hit_vs;
                cs_slice_dir = cs_slice_dir * (1.0f / 6.0f);
                // mul(float4(cs_slice_dir, 0, 0), sample_to_view).xy: row vector times matrix
                const f2 vs_slice_dir{cs_slice_dir.x * s2v[0] + cs_slice_dir.y * s2v[1], cs_slice_dir.x * s2v[4] + cs_slice_dir.y * s2v[5]};
                const f3 slice_normal_vs = normalize(cross(v_vs, f3{vs_slice_dir.x, vs_slice_dir.y, 0}));
                f3 proj_normal_vs = normal_vs - slice_normal_vs * dot(slice_normal_vs, normal_vs);
                const float slice_contrib_weight = length(proj_normal_vs);
                proj_normal_vs = proj_normal_vs / slice_contrib_weight;
                const float sd = dot(vs_slice_dir, f2{proj_normal_vs.x - v_vs.x, proj_normal_vs.y - v_vs.y});
                const float sgn = sd > 0 ? 1.0f : (sd < 0 ? -1.0f : 0.0f);
                const float n_angle = ssgi_fast_acos(clampf(dot(proj_normal_vs, v_vs), -1.0f, 1.0f)) * sgn;
                float theta_cos_max1 = cosf(n_angle - 1.57079632679f);
                float theta_cos_max2 = cosf(n_angle + 1.57079632679f);
                int pc0x = x, pc0y = y, pc1x = x, pc1y = y;
                for (uint32_t i = 0; i < 6; ++i) {
                    {
                        const float t = float(i) + rand_offset;
                        f4 sample_cs{ray_hit_cs.x - cs_slice_dir.x * t, ray_hit_cs.y - cs_slice_dir.y * t, 0, 1};
                        const f2 suv = cs_to_uv(f2{sample_cs.x, sample_cs.y});
                        const int spx = int(output_tex_size.x * suv.x), spy = int(output_tex_size.y * suv.y);
                        if (spx != pc0x || spy != pc0y) {
                            pc0x = spx; pc0y = spy;
                            sample_cs.z = half_depth.ld(spx, spy);
                            theta_cos_max1 = process_sample(fc, sample_cs, center_vs, v_vs, kernel_radius_ws, theta_cos_max1);
                        }
                    }
                    {
                        const float t = float(i) + (1.0f - rand_offset);
                        f4 sample_cs{ray_hit_cs.x + cs_slice_dir.x * t, ray_hit_cs.y + cs_slice_dir.y * t, 0, 1};
                        const f2 suv = cs_to_uv(f2{sample_cs.x, sample_cs.y});
                        const int spx = int(output_tex_size.x * suv.x), spy = int(output_tex_size.y * suv.y);
                        if (spx != pc1x || spy != pc1y) {
                            pc1x = spx; pc1y = spy;
                            sample_cs.z = half_depth.ld(spx, spy);
                            theta_cos_max2 = process_sample(fc, sample_cs, center_vs, v_vs, kernel_radius_ws, theta_cos_max2);
                        }
                    }
                }
                const float h1 = -ssgi_fast_acos(theta_cos_max1);
                const float h2 = +ssgi_fast_acos(theta_cos_max2);
                const float h1p = n_angle + fmaxf(h1 - n_angle, -1.57079632679f);
                const float h2p = n_angle + fminf(h2 - n_angle, 1.57079632679f);
                const float inv_ao = ssgi_integrate_arc(h1p, h2p, n_angle);
                const float col = fmaxf(0.0f, inv_ao) * slice_contrib_weight;
                ssgi_tex.st(x, y, f32_to_f16(fmaxf(0.0f, col)));
            }
        // ---- "ssao spatial" (spatial_filter.hlsl), half res
        ImgR16Fs spatial = get<uint16_t>("spatially_filtered_tex", hw, hh);
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                float result = 0, w_sum = 0;
                const float center_depth = half_depth.ld(x, y);
                if (center_depth != 0.0f) {
                    const f3 center_normal = xyz(unpack_rgba8_snorm(half_view_normal.ld(x, y)));
                    w_sum = 1.0f;
                    result = f16_to_f32(ssgi_tex.ld(x, y));
                    for (int yy = -1; yy <= 1; ++yy)
                        for (int xx = -1; xx <= 1; ++xx) {
                            if (xx == 0 && yy == 0) continue;
                            const float sd = half_depth.ld(x + xx, y + yy);
                            if (sd == 0.0f) continue;
                            const float s = f16_to_f32(ssgi_tex.ld(x + xx, y + yy));
                            const f3 n = xyz(unpack_rgba8_snorm(half_view_normal.ld(x + xx, y + yy)));
                            const float depth_diff = 1.0f - (center_depth / sd);
                            const float depth_factor = exp2f(-200.0f * fabsf(depth_diff));
                            float nf = fmaxf(0.0f, dot(n, center_normal));
                            nf *= nf; nf *= nf;
                            float w = 1;
                            w *= depth_factor;
                            w *= nf;
                            w_sum += w;
                            result += s * w;
                        }
                }
                spatial.st(x, y, f32_to_f16(result / fmaxf(w_sum, 1e-5f)));
            }
        // ---- "ssao upsample" (upsample.hlsl), full res, R16F
        ImgR16Fs upsampled = get<uint16_t>("upsampled_tex", W, H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float result = 0, w_sum = 0;
                const float center_depth = depth.ld(x, y);
                if (center_depth != 0.0f) {
                    for (int yy = -1; yy <= 1; ++yy)
                        for (int xx = -1; xx <= 1; ++xx) {
                            const int sx = x / 2 + xx, sy = y / 2 + yy;
                            const float sd = depth.ld(sx * 2, sy * 2);
                            if (sd == 0.0f) continue;
                            const float s = f16_to_f32(spatial.ld(sx, sy));
                            const float depth_diff = 1.0f - (center_depth / sd);
                            float w = 1;
                            w *= exp2f(-200.0f * fabsf(depth_diff));
                            w *= expf(-float(xx * xx + yy * yy));
                            w_sum += w;
                            result += s * w;
                        }
                }
                if (w_sum > 1e-6f) upsampled.st(x, y, f32_to_f16(result / w_sum));
                else upsampled.st(x, y, spatial.ld(x / 2, y / 2));
            }
        // ---- "ssao temporal" (temporal_filter.hlsl), full res; history R16F ping-pong "ssgi", output R8_UNORM
        ImgR16Fs hist_out = get<uint16_t>(flip ? "ssgi:1" : "ssgi:0", W, H), hist = get<uint16_t>(flip ? "ssgi:0" : "ssgi:1", W, H);
        flip = !flip;
        ImgR8 final_out = get<uint8_t>(flip ? "filtered_output_tex:0" : "filtered_output_tex:1", W, H);   // double-buffered like the HIP side
        const f4 ots = tex_size4(W, H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f2 uv = get_uv(float(x), float(y), ots);
                const float center = f16_to_f32(upsampled.ld(x, y));
                const f4 reproj = ld_reproj(reprojection_tex, x, y);
                // bilinear clamp sample of an R16F image
                float history;
                {
                    const f2 huv = uv + f2{reproj.x, reproj.y};
                    const float fx = huv.x * float(W) - 0.5f, fy = huv.y * float(H) - 0.5f;
                    const float x0f = floorf(fx), y0f = floorf(fy);
                    const float tx = fx - x0f, ty = fy - y0f;
                    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
                    const int xa = cl(int(x0f), W), xb = cl(int(x0f) + 1, W), ya = cl(int(y0f), H), yb = cl(int(y0f) + 1, H);
                    const float s00 = f16_to_f32(hist.p[ya * W + xa]), s10 = f16_to_f32(hist.p[ya * W + xb]);
                    const float s01 = f16_to_f32(hist.p[yb * W + xa]), s11 = f16_to_f32(hist.p[yb * W + xb]);
                    const float a = s00 * (1.0f - tx) + s10 * tx, b = s01 * (1.0f - tx) + s11 * tx;
                    history = a * (1.0f - ty) + b * ty;
                }
                float vsum = 0, vsum2 = 0, wsum = 0;
                for (int yy = -2; yy <= 2; ++yy)
                    for (int xx = -2; xx <= 2; ++xx) {
                        const float neigh = f16_to_f32(upsampled.ld(x + xx * 2, y + yy * 2));
                        const float w = expf(-3.0f * float(xx * xx + yy * yy) / float((2 + 1.) * (2 + 1.)));
                        vsum += neigh * w;
                        vsum2 += neigh * neigh * w;
                        wsum += w;
                    }
                const float ex = vsum / wsum, ex2 = vsum2 / wsum;
                const float dev = sqrtf(fmaxf(0.0f, ex2 - ex * ex));
                const float box_size = 0.5f, n_deviations = 5.0f;
                const float nmin = lerp(center, ex, box_size * box_size) - dev * box_size * n_deviations;
                const float nmax = lerp(center, ex, box_size * box_size) + dev * box_size * n_deviations;
                const float clamped_history = clampf(history, nmin, nmax);
                const float res = lerp(clamped_history, center, 1.0f / 8.0f);
                hist_out.st(x, y, f32_to_f16(res));
                final_out.st(x, y, to_unorm8(res));
            }
        return final_out;
    }
};

}  // namespace okj
