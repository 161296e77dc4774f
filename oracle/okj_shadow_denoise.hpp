// ORACLE (test infrastructure). ShadowDenoiseRenderer restated from crates/lib/kajiya/src/renderers/shadow_denoise.rs:19-148,
// assets/shaders/shadow_denoise/{bitpack_shadow_mask,megakernel,spatial_filter}.hlsl and the FidelityFX shadow denoiser headers
// they include (shadow_denoise/ffx/ffx_denoiser_shadows_{prepare,tileclassification,filter,util}.hlsl, AMD, MIT licence) with
// kajiya's callbacks: every pixel is a shadow receiver, reprojection comes from the reprojection map (history read with the
// 16-tap Catmull-Rom fetch of inc/image.hlsl:41-82), disocclusion from the map's 2x2 validity bits, soft_color_clamp
// (inc/soft_color_clamp.hlsl) instead of the hard clamp, linear temporal blend, moments' sample count capped at 32.
#pragma once
#include "okj_passes.hpp"

namespace okj {

struct ShadowDenoise {
    std::map<std::string, std::vector<uint8_t>> surf;
    bool flip_accum = false, flip_moments = false;
    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    static f4 cubic_hermite(f4 A, f4 B, f4 C, f4 D, float t) {   // inc/curve.hlsl:4-13
        const float t2 = t * t, t3 = t * t * t;
        const f4 a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
        const f4 b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
        const f4 c = -A / 2.0f + C / 2.0f;
        return a * t3 + b * t2 + c * t + B;
    }
    // image_sample_catmull_rom (inc/image.hlsl:41-82) on a texel-fetch functor
    template <typename Fetch> static f4 sample_catmull_rom(Fetch fetch, int W, int H, f2 P) {
        const f2 pixel{P.x * float(W) + 0.5f, P.y * float(H) + 0.5f};
        const f2 frc{frac(pixel.x), frac(pixel.y)};
        const int ix = int(pixel.x) - 1, iy = int(pixel.y) - 1;
        f4 rows[4];
        for (int j = 0; j < 4; ++j)
            rows[j] = cubic_hermite(fetch(ix - 1, iy - 1 + j), fetch(ix, iy - 1 + j), fetch(ix + 1, iy - 1 + j), fetch(ix + 2, iy - 1 + j), frc.x);
        return cubic_hermite(rows[0], rows[1], rows[2], rows[3], frc.y);
    }
    static float kernel_weight(float fi) {   // FFX_DNSR_Shadows_KernelWeight, KERNEL_RADIUS 8 (integer arguments only)
        static float table[9];
        static bool init = false;
        if (!init) {
            auto kw = [](float v) { return expf(-3.0f * (v * v) / ((8 + 1.0f) * (8 + 1.0f))); };
            float sum = kw(0);
            for (int c = 1; c <= 8; ++c) sum += 2 * kw(float(c));
            for (int c = 0; c <= 8; ++c) table[c] = kw(float(c)) * (1.0f / sum);
            init = true;
        }
        return table[int(fi)];
    }
    static float soft_color_clamp1(float center, float history, float ex, float dev) {   // inc/soft_color_clamp.hlsl, scalar
        const float history_dist = fabsf(history - ex) / fmaxf(fabsf(history * 0.1f), dev);
        const float closest_pt = clampf(history, center - dev, center + dev);
        return lerp(history, closest_pt, smoothstep(1.0f, 3.0f, history_dist));
    }

    // returns the RG16F image whose .x is the denoised shadow term (shadow_denoise.rs:112-113)
    ImgRG16F render(const FrameConstants& fc, ImgR8 shadow_mask, ImgR32F depth_tex, ImgU32 geometric_normal_tex, ImgRGBA16S reprojection_tex) {
        const int W = depth_tex.w, H = depth_tex.h;
        const int TW = (W + 7) / 8, TH = (H + 3) / 4;                      // bitpacked_shadow_mask_extent
        // ---- "shadow bitpack" (ffx prepare): one bit per pixel of an 8x4 tile, set when the ray reached the light
        Img<uint32_t> bitpacked = get<uint32_t>("bitpacked_shadows_image", TW, TH);
        for (int ty = 0; ty < TH; ++ty)
            for (int tx = 0; tx < TW; ++tx) {
                uint32_t m = 0;
                for (int j = 0; j < 4; ++j)
                    for (int i = 0; i < 8; ++i)
                        if (from_unorm8(shadow_mask.ld(tx * 8 + i, ty * 4 + j)) > 0.5f) m |= 1u << (j * 8 + i);
                bitpacked.st(tx, ty, m);
            }
        auto read_mask = [&](int linear) { return bitpacked.ld(linear % TW, linear / TW); };
        // ---- "shadow temporal" (megakernel.hlsl + ffx tileclassification)
        ImgRGBA16F moments_out = get<h4>(flip_moments ? "shadow_denoise_moments:1" : "shadow_denoise_moments:0", W, H);
        ImgRGBA16F moments_prev = get<h4>(flip_moments ? "shadow_denoise_moments:0" : "shadow_denoise_moments:1", W, H);
        flip_moments = !flip_moments;
        ImgRG16F accum_out = get<h2>(flip_accum ? "shadow_denoise_accum:1" : "shadow_denoise_accum:0", W, H);
        ImgRG16F accum_prev = get<h2>(flip_accum ? "shadow_denoise_accum:0" : "shadow_denoise_accum:1", W, H);
        flip_accum = !flip_accum;
        ImgRG16F spatial_input = get<h2>("spatial_input_image", W, H);
        Img<uint32_t> metadata = get<uint32_t>("metadata_image", TW, TH);
        const int GW = (W + 7) / 8, GH = (H + 7) / 8;
        auto horizontal_neighborhood = [&](int dx, int dy) -> float {
            if (dy < 0 || dy >= H) return 0.0f;
            const int tix = dx / 8, tiy = dy / 4;
            const int linear = tiy * TW + tix;
            const uint32_t left_tile = tix == 0 ? 0u : read_mask(linear - 1);
            const uint32_t center_tile = read_mask(linear);
            const uint32_t right_tile = tix == TW - 1 ? 0u : read_mask(linear + 1);
            const uint32_t row = uint32_t(dy % 4) * 8;
            uint32_t nb = ((left_tile >> row) & 0xFFu) | (((center_tile >> row) & 0xFFu) << 8) | (((right_tile >> row) & 0xFFu) << 16);
            nb >>= uint32_t(dx % 8);
            float moment = 0;
            for (int i = 0; i < 8; ++i) if (nb & (1u << i)) moment += kernel_weight(float(8 - i));
            if (nb & (1u << 8)) moment += kernel_weight(0);
            for (int i = 1; i <= 8; ++i) if (nb & (1u << (8 + i))) moment += kernel_weight(float(i));
            return moment;
        };
        auto write_moments = [&](int x, int y, f4 m) { m.z = fminf(m.z, 32.0f); st4(moments_out, x, y, m); };
        auto st2 = [&](ImgRG16F& img, int x, int y, f2 v) { img.st(x, y, h2{f32_to_f16(v.x), f32_to_f16(v.y)}); };
#pragma omp parallel for schedule(dynamic, 2)
        for (int gy = 0; gy < GH; ++gy)
            for (int gx = 0; gx < GW; ++gx) {
                // FFX_DNSR_Shadows_SearchSpatialRegion
                uint32_t or_mask = 0, and_mask = 0xFFFFFFFFu;
                {
                    const int btx = gx * 8 / 8, bty = gy * 8 / 4;
                    for (int j = -2; j <= 3; ++j)
                        for (int i = -1; i <= 1; ++i) {
                            const int tix = std::min(std::max(btx + i, 0), TW - 1), tiy = std::min(std::max(bty + j, 0), TH - 1);
                            const uint32_t m = read_mask(tiy * TW + tix);
                            or_mask |= m; and_mask &= m;
                        }
                }
                const bool all_in_light = and_mask == 0xFFFFFFFFu, all_in_shadow = or_mask == 0u;
                const float shadow_value = all_in_light ? 1.0f : 0.0f;
                if (all_in_light || all_in_shadow) {   // FFX_DNSR_Shadows_ClearTargets (every pixel is a receiver in kajiya)
                    metadata.st(gx, gy, (all_in_light ? 2u : 0u) | 1u);
                    for (int ly = 0; ly < 8; ++ly)
                        for (int lx = 0; lx < 8; ++lx) {
                            const int x = gx * 8 + lx, y = gy * 8 + ly;
                            st2(spatial_input, x, y, f2{shadow_value, 0});
                            write_moments(x, y, f4{shadow_value, 0, 8, shadow_value});
                        }
                    continue;
                }
                metadata.st(gx, gy, 0);
                float hn[8][24];   // g_FFX_DNSR_Shadows_neighborhood[gtid.x][...]
                for (int ly = 0; ly < 8; ++ly)
                    for (int lx = 0; lx < 8; ++lx) {
                        const int x = gx * 8 + lx, y = gy * 8 + ly;
                        hn[lx][ly] = horizontal_neighborhood(x, y - 8);
                        hn[lx][ly + 8] = horizontal_neighborhood(x, y);
                        hn[lx][ly + 16] = horizontal_neighborhood(x, y + 8);
                    }
                for (int ly = 0; ly < 8; ++ly)
                    for (int lx = 0; lx < 8; ++lx) {
                        const int x = gx * 8 + lx, y = gy * 8 + ly;
                        float local_neighborhood = 0;
                        local_neighborhood += hn[lx][ly + 8] * kernel_weight(0);
                        local_neighborhood += hn[lx][ly] * kernel_weight(8);
                        local_neighborhood += hn[lx][ly + 16] * kernel_weight(8);
                        for (int i = 1; i < 8; ++i) {
                            const float w = kernel_weight(float(i));
                            local_neighborhood += hn[lx][8 + ly - i] * w;
                            local_neighborhood += hn[lx][8 + ly + i] * w;
                        }
                        const f4 reproj = ld_reproj(reprojection_tex, x, y);
                        // ffx_denoiser_shadows_tileclassification.hlsl:351-352: (did + 0.5) * texel_size, texel_size = the host's f32 reciprocals (input_tex_size.zw) -- a
                        // MULTIPLICATION, not a division by the extent (one ulp apart in uv; found by running the reference's own text)
                        const f2 uv{(float(x) + 0.5f) * (1.0f / float(W)), (float(y) + 0.5f) * (1.0f / float(H))};
                        const f2 history_uv = uv + f2{reproj.x, reproj.y};
                        const float shadow_current = from_unorm8(shadow_mask.ld(x, y));
                        const uint32_t qv = uint32_t(reproj.z * 15.0f + 0.5f);
                        const bool is_disoccluded = ((qv & 1u) + ((qv >> 1) & 1u) + ((qv >> 2) & 1u) + ((qv >> 3) & 1u)) < 4u;
                        f4 previous_moments = mk4(0.0f);
                        if (!is_disoccluded) {
                            previous_moments = sample_catmull_rom([&](int a, int b) { return ld4(moments_prev, a, b); }, W, H, history_uv);
                            previous_moments.y = fmaxf(0.0f, previous_moments.y);
                            previous_moments.z = fmaxf(0.0f, previous_moments.z);
                        }
                        const float old_m = previous_moments.x, old_s = previous_moments.y;
                        const float sample_count = previous_moments.z + 1.0f;
                        const float new_m = lerp(old_m, shadow_current, 1.0f / sample_count);
                        const float new_s = lerp(old_s, (shadow_current - old_m) * (shadow_current - new_m), 1.0f / sample_count);
                        float variance = new_s;
                        f4 moments_current{new_m, new_s, sample_count, local_neighborhood};
                        const float mean = local_neighborhood;
                        float spatial_variance = fmaxf(local_neighborhood - mean * mean, 0.0f);
                        const float std_deviation = sqrtf(spatial_variance);
                        float shadow_previous = shadow_current;
                        if (fc.frame_index != 0) {
                            auto fetch = [&](int a, int b) { const h2 v = accum_prev.ld(a, b); return f4{f16_to_f32(v.x), f16_to_f32(v.y), 0, 0}; };
                            shadow_previous = sample_catmull_rom(fetch, W, H, history_uv).x;
                        }
                        const float sigma = 2.0f;
                        const float temporal_discontinuity = (previous_moments.w - moments_current.w) / fmaxf(0.5f * std_deviation, 0.001f);
                        const float sample_counter_damper = expf(-temporal_discontinuity * temporal_discontinuity / sigma);
                        moments_current.z *= fmaxf(0.5f, sample_counter_damper);
                        float shadow_clamped = soft_color_clamp1(shadow_current, shadow_previous, mean, std_deviation * 0.5f);
                        if (moments_current.z < 16.0f) {
                            const float variance_boost = fmaxf(16.0f - moments_current.z, 1.0f);
                            variance = fmaxf(variance, spatial_variance);
                            variance *= variance_boost;
                        }
                        shadow_clamped = lerp(shadow_clamped, shadow_current, 1.0f / fmaxf(1.0f, moments_current.z));
                        st2(spatial_input, x, y, f2{shadow_clamped, variance});
                        write_moments(x, y, moments_current);
                    }
            }
        // ---- "shadow spatial" x3 (spatial_filter.hlsl + ffx filter), step sizes 1, 2, 4: spatial_input -> accum -> temp -> spatial_input
        ImgRG16F temp = get<h2>("temp", W, H);
        filter_spatial(fc, 1, spatial_input, accum_out, metadata, geometric_normal_tex, depth_tex, TW);
        filter_spatial(fc, 2, accum_out, temp, metadata, geometric_normal_tex, depth_tex, TW);
        filter_spatial(fc, 4, temp, spatial_input, metadata, geometric_normal_tex, depth_tex, TW);
        return spatial_input;
    }

    static float unpack_lo(uint32_t p) { return f16_to_f32(uint16_t(p & 0xffff)); }
    static float unpack_hi(uint32_t p) { return f16_to_f32(uint16_t(p >> 16)); }
    static uint32_t pack2(float a, float b) { return uint32_t(f32_to_f16(a)) | (uint32_t(f32_to_f16(b)) << 16); }

    void filter_spatial(const FrameConstants&, int stepsize, ImgRG16F input, ImgRG16F output, Img<uint32_t> metadata, ImgU32 geometric_normal_tex, ImgR32F depth_tex, int TW) {
        const int W = depth_tex.w, H = depth_tex.h, GW = (W + 7) / 8, GH = (H + 7) / 8;
#pragma omp parallel for schedule(dynamic, 2)
        for (int gy = 0; gy < GH; ++gy)
            for (int gx = 0; gx < GW; ++gx) {
                const int linear = gy * ((W + 7) / 8) + gx;
                const uint32_t meta = metadata.ld(linear % TW, linear / TW);
                const bool is_cleared = (meta & 1u) != 0, all_in_light = (meta & 2u) != 0;
                if (is_cleared) {   // pass index is 0 in kajiya's dispatch: write the constant
                    for (int ly = 0; ly < 8; ++ly)
                        for (int lx = 0; lx < 8; ++lx) output.st(gx * 8 + lx, gy * 8 + ly, h2{f32_to_f16(all_in_light ? 1.0f : 0.0f), f32_to_f16(0.0f)});
                    continue;
                }
                // 16x16 group-shared tile, values stored as packed halves exactly like the shader
                uint32_t s_in[16][16], s_nxy[16][16], s_nzw[16][16];
                float s_depth[16][16];
                for (int ty = 0; ty < 16; ++ty)
                    for (int tx = 0; tx < 16; ++tx) {
                        // ffx_denoiser_shadows_filter.hlsl:76: clamp(did, int2(0, 0), FFX_DNSR_Shadows_GetBufferDimensions() - 1) with the dimensions a
                        // uint2 (shadow_denoise/spatial_filter.hlsl:19-21): int and uint unify to uint, so a NEGATIVE coordinate becomes a huge one
                        // and clamps to the FAR edge, not to 0 (found by running the reference's own text: tests/test_ref_hlsl.py)
                        const int px = int(std::min(uint32_t(gx * 8 - 4 + tx), uint32_t(W - 1))), py = int(std::min(uint32_t(gy * 8 - 4 + ty), uint32_t(H - 1)));
                        const f3 n = unpack_a2r10g10b10(geometric_normal_tex.ld(px, py)) * 2.0f - 1.0f;
                        const h2 v = input.ld(px, py);
                        s_in[ty][tx] = pack2(f16_to_f32(v.x), f16_to_f32(v.y));
                        s_depth[ty][tx] = depth_tex.ld(px, py);
                        s_nxy[ty][tx] = pack2(n.x, n.y);
                        s_nzw[ty][tx] = pack2(n.z, 0.0f);
                    }
                for (int ly = 0; ly < 8; ++ly)
                    for (int lx = 0; lx < 8; ++lx) {
                        const int x = gx * 8 + lx, y = gy * 8 + ly;
                        float weight_sum = 1.0f;
                        f2 shadow_sum{0, 0};
                        if (depth_tex.ld(x, y) != 0.0f) {
                            const float depth = depth_tex.ld(x, y);
                            const int cx = lx + 4, cy = ly + 4;
                            const f2 shadow_center{unpack_lo(s_in[cy][cx]), unpack_hi(s_in[cy][cx])};
                            const f3 normal_center{unpack_lo(s_nxy[cy][cx]), unpack_hi(s_nxy[cy][cx]), unpack_lo(s_nzw[cy][cx])};
                            weight_sum = 1.0f;
                            shadow_sum = shadow_center;
                            const float variance = shadow_center.y;
                            const float std_deviation = sqrtf(fmaxf(variance + 1e-9f, 0.0f));
                            const float t_ = fmaxf(0.0f, 1.0f - 2.0f * std_deviation);
                            const float kernel_sharpening = fmaxf(1e-10f, 1.0f - t_ * t_);
                            const float kernel[3] = {1.0f, exp2f(-0.5849625007211563f / kernel_sharpening), exp2f(-2.584962500721156f / kernel_sharpening)};
                            for (int yy = -1; yy <= 1; ++yy)
                                for (int xx = -1; xx <= 1; ++xx) {
                                    const int tx = cx + xx * stepsize, ty = cy + yy * stepsize;
                                    const float depth_neigh = s_depth[ty][tx];
                                    const f3 normal_neigh{unpack_lo(s_nxy[ty][tx]), unpack_hi(s_nxy[ty][tx]), unpack_lo(s_nzw[ty][tx])};
                                    const f2 shadow_neigh{unpack_lo(s_in[ty][tx]), unpack_hi(s_in[ty][tx])};
                                    const float sky_mul = ((xx == 0 && yy == 0) || depth_neigh >= 1.0f || depth_neigh <= 0.0f) ? 0.0f : 1.0f;
                                    float w = kernel[abs(xx)] * kernel[abs(yy)];
                                    w *= expf(-fabsf(shadow_center.x - shadow_neigh.x) / std_deviation);
                                    w *= exp2f(-fabsf(1.0f - (depth / depth_neigh)) / 0.01f);
                                    w *= powf(saturate(dot(normal_center, normal_neigh)), 32.0f);
                                    w *= sky_mul;
                                    shadow_sum += f2{w, w * w} * shadow_neigh;
                                    weight_sum += w;
                                }
                        }
                        const float mean = shadow_sum.x / weight_sum, variance = shadow_sum.y / (weight_sum * weight_sum);
                        output.st(x, y, h2{f32_to_f16(fmaxf(0.0f, mean)), f32_to_f16(fmaxf(0.0f, variance))});
                    }
            }
    }
};

}  // namespace okj
