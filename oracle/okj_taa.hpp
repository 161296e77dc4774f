// ORACLE (test infrastructure). TaaRenderer restated from crates/lib/kajiya/src/renderers/taa.rs:41-191 and
// assets/shaders/taa/{reproject_history,filter_input,filter_history,input_prob,filter_prob,filter_prob2,taa,
// taa_common}.hlsl, inc/unjitter_taa.hlsl, inc/image.hlsl (5-tap Catmull-Rom).
#pragma once
#include "okj_passes.hpp"

namespace okj {

// taa_common.hlsl (TAA_NONLINEARITY_TYPE 1, TAA_COLOR_MAPPING_MODE 1)
static inline f3 taa_decode_rgb(f3 v) {
    const float m = max3(v.x, v.y, v.z);
    return v * sqrtf(fmaxf(0.0f, m)) / fmaxf(1e-20f, m);
}
static inline f3 taa_encode_rgb(f3 v) {
    const float m = max3(v.x, v.y, v.z);
    return v * (m * m) / fmaxf(1e-20f, m);
}
static inline f3 vsquare(f3 v) { return v * v; }
typedef Img<uint16_t> ImgR16F;
static inline float ld1h(const ImgR16F& i, int x, int y) { return f16_to_f32(i.ld(x, y)); }

struct Taa {
    std::map<std::string, std::vector<uint8_t>> surf;
    int IW = 0, IH = 0, OW = 0, OH = 0;
    bool flip[3] = {false, false, false};

    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    template <typename T> void pingpong(const char* key, int idx, int w, int h, Img<T>& output, Img<T>& history) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx]) std::swap(a, b);
        output = get<T>(a, w, h);
        history = get<T>(b, w, h);
        flip[idx] = !flip[idx];
    }

    // image_sample_catmull_rom_5tap (inc/image.hlsl:88-172, useCornerTaps=false); remap applied to each bilinear tap
    template <typename Remap> static f4 catmull_rom_5tap(const ImgRGBA16F& tex, f2 uv, f2 tex_size, Remap remap) {
        const f2 sample_pos = uv * tex_size;
        const f2 tex_pos1{floorf(sample_pos.x - 0.5f) + 0.5f, floorf(sample_pos.y - 0.5f) + 0.5f};
        const f2 f = sample_pos - tex_pos1;
        const f2 w0 = f * (-0.5f + f * (1.0f - 0.5f * f));
        const f2 w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
        const f2 w2 = f * (0.5f + f * (2.0f - 1.5f * f));
        const f2 w3 = f * f * (-0.5f + 0.5f * f);
        const f2 w12 = w1 + w2;
        const f2 offset12 = w2 / (w1 + w2);
        const f2 p0 = (tex_pos1 - 1.0f) / tex_size, p3 = (tex_pos1 + 2.0f) / tex_size, p12 = (tex_pos1 + offset12) / tex_size;
        f4 result = mk4(0.0f);
        result += remap(sample_bilinear_clamp(tex, f2{p12.x, p0.y})) * w12.x * w0.y;
        result += remap(sample_bilinear_clamp(tex, f2{p0.x, p12.y})) * w0.x * w12.y;
        result += remap(sample_bilinear_clamp(tex, f2{p12.x, p12.y})) * w12.x * w12.y;
        result += remap(sample_bilinear_clamp(tex, f2{p3.x, p12.y})) * w3.x * w12.y;
        result += remap(sample_bilinear_clamp(tex, f2{p12.x, p3.y})) * w12.x * w3.y;
        return result / (w12.x * w0.y + w0.x * w12.y + w12.x * w12.y + w3.x * w12.y + w12.x * w3.y);
    }

    // reproject_history.hlsl:42-129 (wave64 = 8x8 tile; lane^2 = x^2, lane^16 = y^2)
    void pass_reproject(const FrameConstants& fc, ImgRGBA16F history_tex, ImgRGBA16S reprojection_tex, ImgR32F depth_tex, ImgRGBA16F output_tex, ImgRG16F closest_velocity_output) {
        const f4 input_tex_size = tex_size4(IW, IH), output_tex_size = tex_size4(OW, OH);
        const int tiles_x = (OW + 7) / 8, tiles_y = (OH + 7) / 8;
#pragma omp parallel for schedule(static)
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                bool dil[64];
                for (int l = 0; l < 64; ++l) {
                    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                    const f2 scale{input_tex_size.x / output_tex_size.x, input_tex_size.y / output_tex_size.y};
                    const int rx = int(uint32_t((float(x) + 0.5f) * scale.x)), ry = int(uint32_t((float(y) + 0.5f) * scale.y));
                    f2 vmin_, vmax_;
                    const int offs[4][2] = {{-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
                    for (int k = 0; k < 4; ++k) {
                        f4 r = ld_reproj(reprojection_tex, rx + offs[k][0], ry + offs[k][1]);
                        f2 v{r.x, r.y};
                        if (k == 0) { vmin_ = v; vmax_ = v; } else { vmin_ = vmin(vmin_, v); vmax_ = f2{fmaxf(vmax_.x, v.x), fmaxf(vmax_.y, v.y)}; }
                    }
                    const f2 d = vmax_ - vmin_, s = vmax_ + vmin_;
                    dil[l] = d.x > 0.1f * fmaxf(input_tex_size.z, fabsf(s.x)) || d.y > 0.1f * fmaxf(input_tex_size.w, fabsf(s.y));
                }
                bool d1[64];
                for (int l = 0; l < 64; ++l) d1[l] = dil[l] | dil[l ^ 2];
                for (int l = 0; l < 64; ++l) dil[l] = d1[l] | d1[l ^ 16];
                for (int l = 0; l < 64; ++l) {
                    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                    const f2 scale{input_tex_size.x / output_tex_size.x, input_tex_size.y / output_tex_size.y};
                    const int rx = int(uint32_t((float(x) + 0.5f) * scale.x)), ry = int(uint32_t((float(y) + 0.5f) * scale.y));
                    int cx = rx, cy = ry;
                    if (dil[l]) {
                        float reproj_depth = depth_tex.ld(rx, ry);
                        for (int oy = -1; oy <= 1; ++oy)
                            for (int ox = -1; ox <= 1; ++ox) {
                                const float dd = depth_tex.ld(rx + ox, ry + oy);
                                if (dd > reproj_depth) { reproj_depth = dd; cx = rx + ox; cy = ry + oy; }
                            }
                    }
                    const f4 rr = ld_reproj(reprojection_tex, cx, cy);
                    const f2 reproj_xy{rr.x, rr.y};
                    st2(closest_velocity_output, x, y, reproj_xy);
                    const f2 uv = get_uv(float(x), float(y), output_tex_size);
                    const f2 history_uv = uv + reproj_xy;
                    const float ped = fc.pre_exposure_delta;
                    const f4 hp = catmull_rom_5tap(history_tex, history_uv, f2{output_tex_size.x, output_tex_size.y},
                                                   [ped](f4 v) { return mk4(taa_decode_rgb(xyz(v) * ped), v.w); });
                    st4(output_tex, x, y, mk4(xyz(hp), fmaxf(0.0f, hp.w)));
                }
            }
    }

    // filter_input.hlsl:33-88
    struct FilteredInput { f3 clamped_ex, var; };
    static FilteredInput filter_input_inner(const ImgRGBA16F& input_tex, const ImgR32F& depth_tex, int px, int py, float center_depth, float luma_cutoff, float depth_scale) {
        f3 iex = mk3(0.0f), iex2 = mk3(0.0f), clamped_iex = mk3(0.0f);
        float iwsum = 0, clamped_iwsum = 0;
        for (int y = -1; y <= 1; ++y)
            for (int x = -1; x <= 1; ++x) {
                const float distance_w = expf(-(0.8f / float(1 * 1)) * float(x * x + y * y));
                const f3 s = sRGB_to_YCbCr(taa_decode_rgb(xyz(ld4(input_tex, px + x, py + y))));
                const float depth = depth_tex.ld(px + x, py + y);
                float w = 1;
                w *= exp2f(-fminf(16.0f, depth_scale * inverse_depth_relative_diff(center_depth, depth)));
                w *= distance_w;
                w *= powf(saturate(luma_cutoff / s.x), 8.0f);
                clamped_iwsum += w;
                clamped_iex += s * w;
                iwsum += 1;
                iex += s;
                iex2 += s * s;
            }
        clamped_iex = clamped_iex / clamped_iwsum;
        iex = iex / iwsum;
        iex2 = iex2 / iwsum;
        return FilteredInput{clamped_iex, vmax(mk3(0.0f), iex2 - iex * iex)};
    }
    void pass_filter_input(ImgRGBA16F input_tex, ImgR32F depth_tex, ImgRGBA16F output_tex, ImgRGBA16F dev_output_tex) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < IH; ++y)
            for (int x = 0; x < IW; ++x) {
                const float center_depth = depth_tex.ld(x, y);
                const FilteredInput a = filter_input_inner(input_tex, depth_tex, x, y, center_depth, 1e10f, 200.0f);
                const FilteredInput b = filter_input_inner(input_tex, depth_tex, x, y, center_depth, a.clamped_ex.x * 1.001f, 200.0f);
                st4(output_tex, x, y, mk4(b.clamped_ex, 0.0f));
                st4(dev_output_tex, x, y, mk4(vsqrt(a.var), 0.0f));
            }
    }

    // filter_history.hlsl:15-61. in_size = reprojected history extent (output res), out extent = input res
    static f3 fh_filter_input(const ImgRGBA16F& input_tex, f2 uv, float luma_cutoff, int k) {
        f3 iex = mk3(0.0f);
        float iwsum = 0;
        const int sx = int(floorf(uv.x * float(input_tex.w) + 1e-3f)), sy = int(floorf(uv.y * float(input_tex.h) + 1e-3f));
        for (int y = -k; y <= k; ++y)
            for (int x = -k; x <= k; ++x) {
                const float distance_w = expf(-(0.8f / float(k * k)) * float(x * x + y * y));
                const f3 s = sRGB_to_YCbCr(xyz(ld4(input_tex, sx + x, sy + y)));
                float w = 1;
                w *= distance_w;
                w *= powf(saturate(luma_cutoff / s.x), 8.0f);
                iwsum += w;
                iex += s * w;
            }
        return iex / iwsum;
    }
    void pass_filter_history(ImgRGBA16F reprojected_history, ImgRGBA16F output_tex) {
        const int k = (float(reprojected_history.w) / float(output_tex.w) > 1.75f) ? 2 : 1;
        const f4 ots = tex_size4(output_tex.w, output_tex.h);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < output_tex.h; ++y)
            for (int x = 0; x < output_tex.w; ++x) {
                const f2 uv = get_uv(float(x), float(y), ots);
                const float filtered_luma = fh_filter_input(reprojected_history, uv, 1e10f, k).x;
                st4(output_tex, x, y, mk4(fh_filter_input(reprojected_history, uv, filtered_luma * 1.001f, k), 0.0f));
            }
    }

    // input_prob.hlsl:50-108
    void pass_input_prob(const FrameConstants& fc, ImgRGBA16F filtered_input_tex, ImgRGBA16F filtered_input_dev_tex, ImgRGBA16F filtered_history_tex,
                         ImgRGBA16S reprojection_tex, ImgRGBA16F smooth_var_history_tex, ImgRG16F velocity_history_tex, ImgR16F output_tex) {
        const f4 its = tex_size4(IW, IH);
        const f2 sop{fc.view_constants.sample_offset_pixels[0], fc.view_constants.sample_offset_pixels[1]};
#pragma omp parallel for schedule(static)
        for (int y = 0; y < IH; ++y)
            for (int x = 0; x < IW; ++x) {
                float input_prob = 0;
                f3 ivar = mk3(0.0f);
                for (int oy = -1; oy <= 1; ++oy)
                    for (int ox = -1; ox <= 1; ++ox) ivar = vmax(ivar, xyz(ld4(filtered_input_dev_tex, x + ox * 2, y + oy * 2)));
                ivar = vsquare(ivar);
                const f2 input_uv{(float(x) + sop.x) * its.z, (float(y) + sop.y) * its.w};
                const f4 closest_history = unpack_rgba16f(sample_nearest_clamp(filtered_history_tex, input_uv));
                const f4 rp = ld_reproj(reprojection_tex, x, y);
                const f2 huv = input_uv + f2{rp.x, rp.y};
                const f3 closest_smooth_var = xyz(sample_bilinear_clamp(smooth_var_history_tex, huv));
                const f2 closest_vel = sample_bilinear_clamp(velocity_history_tex, huv) * fc.delta_time_seconds;
                const f3 combined_var = vmin(closest_smooth_var, ivar * 10.0f);
                for (int oy = -1; oy <= 1; ++oy)
                    for (int ox = -1; ox <= 1; ++ox) {
                        const f3 s = xyz(ld4(filtered_input_tex, x + ox, y + oy));
                        const f3 idiff = s - xyz(closest_history);
                        const f4 rv = ld_reproj(reprojection_tex, x + ox, y + oy);
                        const f2 vel{rv.x, rv.y};
                        const f2 q{(vel.x - closest_vel.x) / fmaxf(1.0f, fabsf(vel.x + closest_vel.x)), (vel.y - closest_vel.y) / fmaxf(1.0f, fabsf(vel.y + closest_vel.y))};
                        const float vdiff = length(q);
                        const float prob = exp2f(-1.0f * length(idiff * idiff / vmax(mk3(1e-6f), combined_var)) - 1000.0f * vdiff);
                        input_prob = fmaxf(input_prob, prob);
                    }
                output_tex.st(x, y, f32_to_f16(input_prob));
            }
    }
    // filter_prob.hlsl / filter_prob2.hlsl
    void pass_filter_prob(ImgR16F input_tex, ImgR16F output_tex) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < IH; ++y)
            for (int x = 0; x < IW; ++x) {
                float prob = ld1h(input_tex, x, y);
                for (int oy = -1; oy <= 1; ++oy)
                    for (int ox = -1; ox <= 1; ++ox) prob = fmaxf(prob, ld1h(input_tex, x + ox, y + oy));
                output_tex.st(x, y, f32_to_f16(prob));
            }
    }
    void pass_filter_prob2(ImgR16F input_tex, ImgR16F output_tex) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < IH; ++y)
            for (int x = 0; x < IW; ++x) {
                f2 weighted{0, 0};
                for (int oy = -2; oy <= 2; ++oy)
                    for (int ox = -2; ox <= 2; ++ox) {
                        const float np = ld1h(input_tex, x + ox * 2, y + oy * 2);
                        weighted += f2{exp2f(-clampf(10.0f * np, 0.0f, 100.0f)), 1.0f};  // exponential_squish (inc/math.hlsl:63-65)
                    }
                const float prob = fmaxf(0.0f, -1.0f / 10.0f * log2f(1e-30f + weighted.x / weighted.y));  // exponential_unsquish
                output_tex.st(x, y, f32_to_f16(prob));
            }
    }

    // inc/unjitter_taa.hlsl:58-125
    struct Unjittered { f4 color; float coverage; f3 ex, ex2; };
    Unjittered sample_image_unjitter_taa(const ImgRGBA16F& img, int opx, int opy, f2 sample_offset_pixels, float kernel_scale, int k) const {
        const f2 input_tex_size{float(IW), float(IH)}, output_tex_size{float(OW), float(OH)};
        const f2 scale = input_tex_size / output_tex_size;
        const int bx = int((float(opx) + 0.5f) * scale.x), by = int((float(opy) + 0.5f) * scale.y);
        const f2 dst_sample_loc{float(opx) + 0.5f, float(opy) + 0.5f};
        const f2 base_src_sample_loc = f2{float(bx) + 0.5f + sample_offset_pixels.x, float(by) + 0.5f - sample_offset_pixels.y} / scale;
        f4 res = mk4(0.0f);
        f3 ex = mk3(0.0f), ex2 = mk3(0.0f);
        float dev_wt_sum = 0, wt_sum = 0;
        for (int y = -k; y <= k; ++y)
            for (int x = -k; x <= k; ++x) {
                const f2 src_sample_loc = base_src_sample_loc + f2{float(x), float(y)} / scale;
                const f4 c = ld4(img, bx + x, by + y);
                const f4 col = mk4(sRGB_to_YCbCr(taa_decode_rgb(xyz(c))), 1.0f);
                const f2 o = (src_sample_loc - dst_sample_loc) * kernel_scale;
                const float dist2 = dot(o, o);
                const float dev_wt = exp2f(-dist2 * scale.x);
                const float wt = exp2f(-10.0f * dist2 * scale.x);
                res += col * wt;
                wt_sum += wt;
                ex += xyz(col) * dev_wt;
                ex2 += xyz(col) * xyz(col) * dev_wt;
                dev_wt_sum += dev_wt;
            }
        return Unjittered{res, wt_sum, ex / dev_wt_sum, ex2 / dev_wt_sum};
    }

    // taa.hlsl:94-338
    void pass_taa(const FrameConstants& fc, ImgRGBA16F input_tex, ImgRGBA16F history_tex, ImgRGBA16S reprojection_tex, ImgRG16F closest_velocity_tex,
                  ImgRG16F velocity_history_tex, ImgRGBA16F smooth_var_history_tex, ImgR16F input_prob_tex, ImgRGBA16F temporal_output_tex,
                  ImgRGBA16F output_tex, ImgRGBA16F smooth_var_output_tex, ImgRG16F velocity_output_tex) {
        const f4 ots = tex_size4(OW, OH);
        const f2 frac_{float(IW) / float(OW), float(IH) / float(OH)};
        const f2 sop{fc.view_constants.sample_offset_pixels[0], fc.view_constants.sample_offset_pixels[1]};
#pragma omp parallel for schedule(static)
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x) {
                const int rx = int(uint32_t((float(x) + 0.5f) * frac_.x)), ry = int(uint32_t((float(y) + 0.5f) * frac_.y));
                const f2 uv = get_uv(float(x), float(y), ots);
                const f4 history_packed = ld4(history_tex, x, y);
                f3 history = xyz(history_packed);
                float history_coverage = fmaxf(0.0f, history_packed.w);
                f4 csum = mk4(0.0f); float wsum = 0;                         // fetch_blurred_history(px, 2, 1)
                for (int oy = -2; oy <= 2; ++oy)
                    for (int ox = -2; ox <= 2; ++ox) {
                        const float w = expf(-float(ox * ox + oy * oy));
                        csum += ld4(history_tex, x + ox, y + oy) * w;
                        wsum += w;
                    }
                const f4 bhistory_packed = csum / wsum;
                f3 bhistory = xyz(bhistory_packed);
                const float bhistory_coverage = bhistory_packed.w;
                history = sRGB_to_YCbCr(history);
                bhistory = sRGB_to_YCbCr(bhistory);
                const f4 reproj = ld_reproj(reprojection_tex, rx, ry);
                const f2 reproj_xy = ld2(closest_velocity_tex, x, y);
                const Unjittered center_sample = sample_image_unjitter_taa(input_tex, x, y, sop, 1.0f, 1);
                const Unjittered bcenter_sample = sample_image_unjitter_taa(input_tex, x, y, sop, 0.333f, 1);
                float coverage = center_sample.coverage;
                f3 center = xyz(center_sample.color);
                const f3 bcenter = xyz(bcenter_sample.color) / bcenter_sample.coverage;
                history = lerp(history, bcenter, saturate(1.0f - history_coverage));
                bhistory = lerp(bhistory, bcenter, saturate(1.0f - bhistory_coverage));
                const float input_prob = ld1h(input_prob_tex, rx, ry);
                const f3 ex = center_sample.ex, ex2 = center_sample.ex2;
                const f3 var = vmax(mk3(0.0f), ex2 - ex * ex);
                const f3 prev_var = mk3(sample_bilinear_clamp(smooth_var_history_tex, uv + reproj_xy).x);
                const f2 vel_now = reproj_xy / fc.delta_time_seconds;
                const f2 vel_prev = sample_bilinear_clamp(velocity_history_tex, uv + reproj_xy);
                const f2 vq{(vel_now.x - vel_prev.x) / fmaxf(1.0f, fabsf(vel_now.x + vel_prev.x)), (vel_now.y - vel_prev.y) / fmaxf(1.0f, fabsf(vel_now.y + vel_prev.y))};
                const float vel_diff = length(vq);
                const float var_blend = saturate(0.3f + 0.7f * (1 - reproj.z) + vel_diff);
                f3 smooth_var = vmax(var, lerp(prev_var, var, var_blend));
                smooth_var = lerp(var, smooth_var, saturate(input_prob));
                const f3 input_dev = vsqrt(var);
                f3 clamped_history;
                {
                    float box_n_deviations = lerp(0.8f, 3.0f, input_prob);
                    const f3 nmin = ex - input_dev * box_n_deviations, nmax = ex + input_dev * box_n_deviations;
                    const f3 clamped_bhistory = vclamp(bhistory, nmin, nmax);
                    const float clamping_event = length(vmax(mk3(0.0f), vmax(bhistory - nmax, nmin - bhistory)) / vmax(mk3(0.01f), ex));
                    const f3 outlier3 = vmax(mk3(0.0f), vmax(nmin - history, history - nmax) / (0.1f + vmax(vmax(vabs(history), vabs(ex)), mk3(1e-5f))));
                    const f3 boutlier3 = vmax(mk3(0.0f), vmax(nmin - bhistory, bhistory - nmax) / (0.1f + vmax(vmax(vabs(bhistory), vabs(ex)), mk3(1e-5f))));
                    const float outlier = fmaxf(outlier3.x, fmaxf(outlier3.y, outlier3.z));
                    const float boutlier = fmaxf(boutlier3.x, fmaxf(boutlier3.y, boutlier3.z));
                    const f2 huv = uv + reproj_xy;
                    const bool history_valid = huv.x == saturate(huv.x) && huv.y == saturate(huv.y);
                    if (history_valid) {
                        const float non_disoccluding_outliers = fmaxf(0.0f, outlier - boutlier) * 10;
                        const f3 unclamped_history_detail = history - clamped_bhistory;
                        const float temporal_clamping_detail = fabsf(unclamped_history_detail.x / fmaxf(1e-3f, input_dev.x)) * 0.05f;
                        const float temporal_stability = saturate(1 - temporal_clamping_detail);
                        const float allow_unclamped_detail = saturate(non_disoccluding_outliers) * temporal_stability;
                        f3 history_detail = history - bhistory;
                        history_detail = lerp(history_detail, unclamped_history_detail, allow_unclamped_detail);
                        const float initial_bclamp_amount = saturate(dot(clamped_bhistory - bhistory, bcenter - bhistory) /
                                                                     fmaxf(1e-5f, length(clamped_bhistory - bhistory) * length(bcenter - bhistory)));
                        const float effective_clamp_amount = saturate(initial_bclamp_amount) * (1 - allow_unclamped_detail);
                        const float keep_detail = 1 - effective_clamp_amount;
                        history_detail *= keep_detail;
                        clamped_history = clamped_bhistory + history_detail;
                        if (frac_.x < 1.0f) history_coverage *= lerp(lerp(0.0f, 0.9f, keep_detail), 1.0f, saturate(10 * clamping_event));
                    } else {
                        clamped_history = clamped_bhistory;
                        coverage = 1;
                        center = bcenter;
                        history_coverage = 0;
                    }
                    clamped_history = lerp(clamped_history, history, smoothstep(0.5f, 1.0f, input_prob));
                }
                float total_coverage = fmaxf(1e-5f, history_coverage + coverage);
                f3 temporal_result = (clamped_history * history_coverage + center) / total_coverage;
                const float max_coverage = fmaxf(2.0f, 8.0f / (frac_.x * frac_.y));
                total_coverage = fminf(max_coverage, total_coverage);
                coverage = total_coverage;
                st4(smooth_var_output_tex, x, y, mk4(smooth_var, 0.0f));
                temporal_result = YCbCr_to_sRGB(temporal_result);
                temporal_result = taa_encode_rgb(temporal_result);
                temporal_result = vmax(mk3(0.0f), temporal_result);
                st4(temporal_output_tex, x, y, mk4(temporal_result, coverage));
                st4(output_tex, x, y, mk4(temporal_result, 0.0f));   // this_frame_result = lerp(temporal_result, 0, a=0), alpha 0
                st2(velocity_output_tex, x, y, reproj_xy / fc.delta_time_seconds);
            }
    }

    // TaaRenderer::render (taa.rs:41-191)
    ImgRGBA16F render(const FrameConstants& fc, ImgRGBA16F input_tex, ImgRGBA16S reprojection_map, ImgR32F depth_tex, int out_w, int out_h, ImgRGBA16F* temporal_out = nullptr) {
        if (IW != input_tex.w || IH != input_tex.h || OW != out_w || OH != out_h) { surf.clear(); IW = input_tex.w; IH = input_tex.h; OW = out_w; OH = out_h; }
        ImgRGBA16F temporal_output_tex, history_tex; pingpong("taa", 0, OW, OH, temporal_output_tex, history_tex);
        ImgRG16F temporal_velocity_output_tex, velocity_history_tex; pingpong("taa.velocity", 1, OW, OH, temporal_velocity_output_tex, velocity_history_tex);
        ImgRGBA16F reprojected_history_img = get<h4>("reprojected_history_img", OW, OH);
        ImgRG16F closest_velocity_img = get<h2>("closest_velocity_img", OW, OH);
        pass_reproject(fc, history_tex, reprojection_map, depth_tex, reprojected_history_img, closest_velocity_img);
        ImgRGBA16F smooth_var_output_tex, smooth_var_history_tex; pingpong("taa.smooth_var", 2, OW, OH, smooth_var_output_tex, smooth_var_history_tex);
        ImgRGBA16F filtered_input_img = get<h4>("filtered_input_img", IW, IH);
        ImgRGBA16F filtered_input_deviation_img = get<h4>("filtered_input_deviation_img", IW, IH);
        pass_filter_input(input_tex, depth_tex, filtered_input_img, filtered_input_deviation_img);
        ImgRGBA16F filtered_history_img = get<h4>("filtered_history_img", IW, IH);
        pass_filter_history(reprojected_history_img, filtered_history_img);
        ImgR16F input_prob_img = get<uint16_t>("input_prob_img", IW, IH);
        pass_input_prob(fc, filtered_input_img, filtered_input_deviation_img, filtered_history_img, reprojection_map, smooth_var_history_tex, velocity_history_tex, input_prob_img);
        ImgR16F prob_filtered1_img = get<uint16_t>("prob_filtered1_img", IW, IH);
        pass_filter_prob(input_prob_img, prob_filtered1_img);
        ImgR16F prob_filtered2_img = get<uint16_t>("prob_filtered2_img", IW, IH);
        pass_filter_prob2(prob_filtered1_img, prob_filtered2_img);
        ImgRGBA16F this_frame_output_img = get<h4>("this_frame_output_img", OW, OH);
        pass_taa(fc, input_tex, reprojected_history_img, reprojection_map, closest_velocity_img, velocity_history_tex, smooth_var_history_tex, prob_filtered2_img,
                 temporal_output_tex, this_frame_output_img, smooth_var_output_tex, temporal_velocity_output_tex);
        if (temporal_out) *temporal_out = temporal_output_tex;
        return this_frame_output_img;
    }
};

} // namespace okj
