// ORACLE (test infrastructure). PostProcessRenderer::render restated from crates/lib/kajiya/src/renderers/post.rs:10-272:
//   blur_pyramid      (post.rs:10-61):  mip 0 by the RUST kernel `blur::blur_cs` (rust-shaders/src/blur.rs: 10 vertical taps),
//                                       mips 1.. by assets/shaders/blur.hlsl (11 vertical taps); B10G11R11_UFLOAT, half-res base,
//                                       all mip levels minus one
//   luminance histogram (post.rs:138-186; shaders/post/luminance_histogram_{clear,calculate,copy}.hlsl), read_back_histogram (:188-235)
//   rev_blur_pyramid  (post.rs:63-110; rust-shaders/src/rev_blur.rs)
//   post combine      (shaders/post_combine.hlsl with the shipped switches: glare 0.05, vignette, display transform, dither; no grade,
//                      no sharpen) + inc/color/{display_transform,bezold_brucke,helmholtz_kohlrausch,ipt,lab,luv,xyz,srgb,math}.hlsl
// and the exposure state of world_renderer.rs:218-285,919-948 (host arithmetic, restated in okj_api.cpp callers' language).
// Undefined in the reference -> chosen here (and in csrc/post.hip):
//  * the coarsest mip of the rev-blur pyramid is read (post.rs:73 never takes the `self_weight = 0` branch) but never written:
//    transient image memory. Chosen: zeros.
//  * out-of-bounds image fetches in the blur kernels return 0 and still count in the weight sum (robust buffer access).
//  * a NaN texture coordinate / NaN -> uint conversion (a black pixel makes xyY = 0/0): coordinate 0 / index 0. The NaN colour
//    itself flows through and is stored as 0 by the B10G11R11 pack.
//  * pow(x, 2) / pow(x, 3) of the histogram weight and the vignette are written as products.
// The Bezold-Brucke LUT (BINDLESS_LUT_BEZOLD_BRUCKE, 64x1 RG16F, lut_renderers.rs:45-76) and the blue-noise image are inputs.
#pragma once
#include "okj_rtr.hpp"

namespace okj {

// ---------------------------------------------------------------- inc/color/*.hlsl
static const f2 white_D65_xy = {0.31271f, 0.32902f};
static inline f3 mul33r(const float* m, f3 v) {   // HLSL float3x3(row, row, row) * v
    return f3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
static inline f3 post_sRGB_to_XYZ(f3 c) {          // srgb.hlsl:11-17
    static const float m[9] = {0.4124564f, 0.3575761f, 0.1804375f, 0.2126729f, 0.7151522f, 0.0721750f, 0.0193339f, 0.1191920f, 0.9503041f};
    return mul33r(m, c);
}
static inline f3 post_XYZ_to_sRGB(f3 c) {          // srgb.hlsl:22-28
    static const float m[9] = {3.2404542f, -1.5371385f, -0.4985314f, -0.9692660f, 1.8760108f, 0.0415560f, 0.0556434f, -0.2040259f, 1.0572252f};
    return mul33r(m, c);
}
static inline f3 CIE_xyY_to_XYZ(f3 xyY) {          // xyz.hlsl:8-19
    const float X = (xyY.z / xyY.y) * xyY.x;
    const float Z = (xyY.z / xyY.y) * (1.0f - xyY.x - xyY.y);
    return f3{X, xyY.z, Z};
}
static inline f3 CIE_XYZ_to_xyY(f3 XYZ) {          // xyz.hlsl:21-34
    const float N = XYZ.x + XYZ.y + XYZ.z;
    return f3{XYZ.x / N, XYZ.y / N, XYZ.y};
}
static inline float spow(float x, float p) { return x >= 0.0f ? powf(x, p) : -powf(-x, p); }
static inline f3 XYZ_to_IPT(f3 xyz_) {             // ipt.hlsl:4-25
    static const float a[9] = {0.4002f, 0.7075f, -0.0807f, -0.2280f, 1.1500f, 0.0612f, 0.0f, 0.0f, 0.9184f};
    static const float b[9] = {0.4000f, 0.4000f, 0.2000f, 4.4550f, -4.8510f, 0.3960f, 0.8056f, 0.3572f, -1.1628f};
    f3 lms = mul33r(a, xyz_);
    lms = f3{spow(lms.x, 0.43f), spow(lms.y, 0.43f), spow(lms.z, 0.43f)};
    return mul33r(b, lms);
}
static inline f3 IPT_to_XYZ(f3 ipt) {              // ipt.hlsl:28-47
    static const float a[9] = {1.0f, 0.0976f, 0.2052f, 1.0f, -0.1139f, 0.1332f, 1.0f, 0.0326f, -0.6769f};
    static const float b[9] = {1.8501f, -1.1383f, 0.2385f, 0.3668f, 0.6439f, -0.0107f, 0.0f, 0.0f, 1.0889f};
    f3 lms = mul33r(a, ipt);
    const float e = 1.0f / 0.43f;
    lms = f3{spow(lms.x, e), spow(lms.y, e), spow(lms.z, e)};
    return mul33r(b, lms);
}
static inline f2 CIE_xyY_xy_to_LUV_uv(f2 xy) { return xy * f2{4.0f, 9.0f} / (-2.0f * xy.x + 12.0f * xy.y + 3.0f); }                 // luv.hlsl:12-14
static inline f2 CIE_XYZ_to_LUV_uv(f3 v) { return f2{v.x, v.y} * f2{4.0f, 9.0f} / dot(v, f3{1.0f, 15.0f, 3.0f}); }                   // luv.hlsl:16-18
static inline float post_catmull_rom(float x, float v0, float v1, float v2, float v3) {                                            // math.hlsl:14-19
    const float c2 = -.5f * v0 + 0.5f * v2;
    const float c3 = v0 + -2.5f * v1 + 2.0f * v2 + -.5f * v3;
    const float c4 = -.5f * v0 + 1.5f * v1 + -1.5f * v2 + 0.5f * v3;
    return ((c4 * x + c3) * x + c2) * x + v1;
}
// float -> uint conversion as the GPU does it: NaN and negatives -> 0
static inline uint32_t f2u_sat(float f) { return f > 0.0f ? (f >= 4294967296.0f ? 0xffffffffu : uint32_t(f)) : 0u; }
// helmholtz_kohlrausch.hlsl:53-106 (HK_ADJUSTMENT_METHOD_CUSTOM_G0)
static inline float XYZ_to_hk_luminance_multiplier_custom_g0(f3 XYZ) {
    f2 uv = CIE_XYZ_to_LUV_uv(XYZ);
    const f2 d65_uv = CIE_xyY_xy_to_LUV_uv(white_D65_xy);
    uv = uv - d65_uv;
    const float theta = atan2f(uv.y, uv.x);
    static const float samples[16] = {-0.006f, -0.021f, -0.033f, -0.009f, 0.14f, 0.114f, 0.111f, 0.1005f,
                                      0.069f, 0.0135f, -0.045f, -0.075f, -0.075f, -0.03f, 0.006f, 0.006f};
    const float t = (theta / M_PI_F) * 0.5f + 0.5f;
    const uint32_t i0 = f2u_sat(floorf(t * 16.0f)) % 16u;
    const uint32_t i1 = (i0 + 1u) % 16u;
    const float q0 = samples[(i0 + 15u) % 16u], q1 = samples[i0], q2 = samples[i1], q3 = samples[(i1 + 1u) % 16u];
    const float interp = (t - float(i0) / 16.0f) * 16.0f;
    const float q = post_catmull_rom(interp, q0, q1, q2, q3);
    const float adapt_lum = 20.0f;
    const float kbr = 0.2717f * (6.469f + 6.362f * powf(adapt_lum, 0.4495f)) / (6.469f + powf(adapt_lum, 0.4495f));
    const float suv = 13.0f * length(uv);
    const float mult_cbrt = 1.0f + (q + 0.0872f * kbr) * suv;
    return mult_cbrt * mult_cbrt * mult_cbrt;
}
static inline float hk_from_sRGB(f3 stimulus) { return XYZ_to_hk_luminance_multiplier_custom_g0(post_sRGB_to_XYZ(stimulus)); }
static inline float srgb_to_equivalent_luminance(float hk_mult, f3 stimulus) { return hk_mult * post_sRGB_to_XYZ(stimulus).y; }
static inline float lab_f(float v) { return v > 0.008856f ? powf(fabsf(v), 1.0f / 3.0f) : v * 7.787f + 16.0f / 116.0f; }
static inline f3 XYZ_to_LAB(f3 v) {                // lab.hlsl:21-38
    v = v / f3{0.9504f, 1.0000f, 1.0888f};
    v = f3{lab_f(v.x), lab_f(v.y), lab_f(v.z)};
    return f3{116.0f * v.y - 16.0f, 500.0f * (v.x - v.y), 200.0f * (v.y - v.z)};
}
// bezold_brucke.hlsl:18-35 (BB_LUT_LUT_MAPPING_QUAD)
static inline float bb_xy_white_offset_to_lut_coord(f2 offset) {
    offset = offset / fmaxf(fabsf(offset.x), fabsf(offset.y));
    const float sgn = (offset.x + offset.y) > 0.0f ? 1.0f : -1.0f;
    return sgn * (0.125f * (offset.x - offset.y) + 0.25f);
}
// SAMPLE_BEZOLD_BRUCKE_LUT (post_combine.hlsl:7-10): sampler_llr = bilinear, REPEAT; 64x1 RG16F
static inline f2 sample_bezold_brucke_lut(const h2* lut, float coord) {
    if (!(coord == coord)) coord = 0.0f;
    const float fx = coord * 64.0f - 0.5f;
    const float x0f = floorf(fx), tx = fx - x0f;
    const int x0 = f2i_sat(x0f);
    const int xa = ((x0 % 64) + 64) % 64, xb = (xa + 1) % 64;
    const f2 a = f2{f16_to_f32(lut[xa].x), f16_to_f32(lut[xa].y)}, b = f2{f16_to_f32(lut[xb].x), f16_to_f32(lut[xb].y)};
    return a * (1.0f - tx) + b * tx;
}
// bezold_brucke.hlsl:138-149
static inline f3 bezold_brucke_shift_XYZ_with_lut(const h2* lut, f3 XYZ, float amount) {
    const f3 xyY = CIE_XYZ_to_xyY(XYZ);
    const f2 offset = f2{xyY.x, xyY.y} - white_D65_xy;
    const float lut_coord = bb_xy_white_offset_to_lut_coord(offset);
    const f2 shifted_xy = f2{xyY.x, xyY.y} + sample_bezold_brucke_lut(lut, lut_coord) * length(offset) * amount;
    return CIE_xyY_to_XYZ(f3{shifted_xy.x, shifted_xy.y, xyY.z});
}
// display_transform.hlsl:67-83 (BRIGHTNESS_COMPRESSION_CURVE_SIRAGUSANO_SMITH)
static inline float compress_luminance(float v) { return saturate(1.0205f * powf(v / (v + 1.0f), 1.2f)); }
static inline f3 vpow(f3 v, float p) { return f3{powf(v.x, p), powf(v.y, p), powf(v.z, p)}; }
// display_transform.hlsl:85-216 with PERCEPTUAL_SPACE_IPT, USE_BEZOLD_BRUCKE_SHIFT (LUT), USE_LONG_TAILED_CHROMA_ATTENUATION,
// USE_BRIGHTNESS_LINEAR_CHROMA_ATTENUATION
static inline f3 display_transform_sRGB(const h2* bb_lut, f3 input_stimulus) {
    {
        const float t = sRGB_to_luminance(input_stimulus) / 5.0f;
        const float shift_amount = t / (t + 1.0f);
        input_stimulus = post_XYZ_to_sRGB(bezold_brucke_shift_XYZ_with_lut(bb_lut, post_sRGB_to_XYZ(input_stimulus), shift_amount));
    }
    const float hk = hk_from_sRGB(input_stimulus);
    const float input_equiv_lum = srgb_to_equivalent_luminance(hk, input_stimulus);
    const f3 max_intensity_rgb = input_stimulus / max3(input_stimulus.x, input_stimulus.y, input_stimulus.z);
    const float max_intensity_equiv_lum = srgb_to_equivalent_luminance(hk, max_intensity_rgb);
    const float max_output_scale = 1.0f;
    const float compressed_achromatic_luminance = compress_luminance(input_equiv_lum / max_output_scale) * max_output_scale;
    f3 compressed_rgb = (max_intensity_rgb / max_intensity_equiv_lum) * compressed_achromatic_luminance;
    const float clamped_compressed_achromatic_luminance = fminf(1.0f, compressed_achromatic_luminance);
    const f3 perceptual = XYZ_to_IPT(post_sRGB_to_XYZ(compressed_rgb));
    const f3 perceptual_white = XYZ_to_IPT(post_sRGB_to_XYZ(mk3(clamped_compressed_achromatic_luminance)));
    // chroma_strength only feeds chroma_attenuation_exponent, unused under USE_LONG_TAILED_CHROMA_ATTENUATION (:149-154)
    const float chroma_attenuation_start = 0.0f;
    const float chroma_attenuation_t = saturate(
        (compressed_achromatic_luminance - fminf(1.0f, max_intensity_equiv_lum) * chroma_attenuation_start) /
        (1.03f * max_output_scale - fminf(1.0f, max_intensity_equiv_lum) * chroma_attenuation_start));
    float chroma_attenuation = asinf(powf(chroma_attenuation_t, 3.0f)) / M_PI_F * 2.0f;      // pow(), as the text has it (:161): t * t * t rounds twice and differs in the last bit for a third of the arguments
    {
        const float compressed_achromatic_luminance2 = compress_luminance(0.125f * input_equiv_lum / max_output_scale) * max_output_scale;
        const float chroma_attenuation_t2 = saturate((compressed_achromatic_luminance2 - fminf(1.0f, max_intensity_equiv_lum) * 0.5f) /
                                                     (max_output_scale - fminf(1.0f, max_intensity_equiv_lum) * 0.5f));
        chroma_attenuation = lerp(chroma_attenuation, 1.0f, 1.0f - saturate(1.0f - powf(chroma_attenuation_t2, 4.0f)));
    }
    {
        const f3 perceptual_mid = lerp(perceptual, perceptual_white, chroma_attenuation);
        compressed_rgb = post_XYZ_to_sRGB(IPT_to_XYZ(perceptual_mid));
        const float hk2 = hk_from_sRGB(compressed_rgb);
        for (int i = 0; i < 2; ++i) {
            const float current_brightness = srgb_to_equivalent_luminance(hk2, compressed_rgb);
            compressed_rgb = compressed_rgb * (compressed_achromatic_luminance / fmaxf(1e-10f, current_brightness));
        }
    }
    compressed_rgb = vmax(compressed_rgb, mk3(0.0f));
    const float p = 12.0f;
    compressed_rgb = compressed_rgb * vpow(vpow(compressed_rgb, p) + 1.0f, -1.0f / p);
    const float max_comp = max3(compressed_rgb.x, compressed_rgb.y, compressed_rgb.z);
    const float max_comp_dist = max3(max_comp - compressed_rgb.x, max_comp - compressed_rgb.y, max_comp - compressed_rgb.z);
    compressed_rgb = compressed_rgb / powf(lerp(0.5f, 1.0f, max_comp_dist), 1.0f / p);
    return compressed_rgb;
}
// post_combine.hlsl:44-50
static inline float triangle_remap(float n) {
    const float origin = n * 2.0f - 1.0f;
    float v = origin * (1.0f / sqrtf(fabsf(origin)));
    v = fmaxf(-1.0f, v);
    v -= origin > 0.0f ? 1.0f : (origin < 0.0f ? -1.0f : 0.0f);
    return v;
}

// ---------------------------------------------------------------- PostProcessRenderer
struct Post {
    std::map<std::string, std::vector<uint8_t>> surf;
    uint32_t histogram_readback[256] = {};     // post.rs:121-129 "luminance histogram" (gpu-to-cpu)
    float image_log2_lum = 0.0f;
    int mip_levels = 0;
    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    static int mip_count_1d(uint32_t e) { int n = 0; while (e) { ++n; e >>= 1; } return n; }   // floor(log2) + 1 (image.rs:35-38)
    static int pyramid_mip_levels(int pw, int ph) { return std::max(1, std::max(mip_count_1d(pw), mip_count_1d(ph)) - 1); }   // post.rs:11-21

    static float gaussian_wt(float dst_px, float src_px) {       // blur.rs:18-22 == blur.hlsl:11-15
        const float px_off = (dst_px + 0.5f) * 2.0f - (src_px + 0.5f);
        const float sigma = 5.0f * 0.5f;
        return expf(-px_off * px_off / (sigma * sigma));
    }
    // one blur pass: dst (w x h) from a source fetched by `fetch(x, y)` (0 out of bounds); `vtaps` = 10 (Rust mip 0) or 11 (HLSL).
    // The two texts differ in one more way: blur.rs computes the tap's source coordinate in i32, blur.hlsl in `uint` (`src_px.y + y` with
    // `uint y`, :22; `px.x * 2 + x - kernel_radius` with all three uint, :50) -- a tap left of / above the image has coordinate 2^32 - k there,
    // its Gaussian weight underflows to exactly 0, and it does NOT count in the weight sum (in mip 0 it does, with its zero texel).
    template <typename Fetch> static void blur_pass(ImgU32 dst, int vtaps, Fetch fetch) {
        const bool hlsl = vtaps == 11;
        auto src_coord = [hlsl](int s) { return hlsl ? float(uint32_t(s)) : float(s); };
        for (int y = 0; y < dst.h; ++y)
            for (int x = 0; x < dst.w; ++x) {
                f3 res = mk3(0.0f);
                float wt_sum = 0.0f;
                for (int xi = 0; xi <= 10; ++xi) {
                    const int sx = x * 2 + xi - 5;
                    f3 v = mk3(0.0f);                              // vblur (blur.rs:24-39 / blur.hlsl:17-28)
                    float vw = 0.0f;
                    for (int yi = 0; yi < vtaps; ++yi) {
                        const int sy = y * 2 - 5 + yi;
                        const float wt = gaussian_wt(float(y), src_coord(sy));
                        v += fetch(sx, sy) * wt;
                        vw += wt;
                    }
                    v = v / vw;
                    const float wt = gaussian_wt(float(x), src_coord(sx));
                    res += v * wt;
                    wt_sum += wt;
                }
                dst.st(x, y, pack_r11g11b10f(res / wt_sum));
            }
    }
    ImgU32 mip(const char* pyramid, int level, int pw, int ph) {
        return get<uint32_t>(std::string(pyramid) + ":" + std::to_string(level), std::max(1, pw >> level), std::max(1, ph >> level));
    }

    // PostProcessRenderer::render (post.rs:237-271) without its leading read_back_histogram (a separate call here).
    // input RGBA16F W x H; returns the B10G11R11_UFLOAT W x H output.
    // `input`: RGBA16F (standard frame) or, with input32 != nullptr, RGBA32F (the path tracer's accumulation image, world_render_passes.rs:294-330)
    ImgU32 render(const FrameConstants& fc, ImgRGBA16F input, const h2* bb_lut, const uint32_t* blue_noise_rgba8, float post_exposure_mult, float contrast, const f4* input32 = nullptr) {
        const int W = input.w, H = input.h, pw = (W + 1) / 2, ph = (H + 1) / 2;
        auto in_rgb = [&](int x, int y) -> f3 {
            if (!input32) return xyz(ld4(input, x, y));
            return (x >= 0 && y >= 0 && x < W && y < H) ? xyz(input32[size_t(y) * W + x]) : mk3(0.0f);
        };
        mip_levels = pyramid_mip_levels(pw, ph);
        // ---- blur_pyramid
        blur_pass(mip("blur_pyramid", 0, pw, ph), 10, in_rgb);
        for (int m = 1; m < mip_levels; ++m) {
            const ImgU32 src = mip("blur_pyramid", m - 1, pw, ph);
            blur_pass(mip("blur_pyramid", m, pw, ph), 11, [&](int x, int y) { return unpack_r11g11b10f(src.ld(x, y)); });
        }
        // ---- luminance histogram (post.rs:138-186)
        {
            uint32_t hist[256] = {};
            const int level = std::max(0, mip_levels - 7);
            const int ew = std::max(1, (pw + (1 << level) - 1) >> level), eh = std::max(1, (ph + (1 << level) - 1) >> level);
            const ImgU32 src = mip("blur_pyramid", level, pw, ph);
            for (int y = 0; y < eh; ++y)
                for (int x = 0; x < ew; ++x) {
                    const float log_lum = log2f(fmaxf(1e-20f, sRGB_to_luminance(unpack_r11g11b10f(src.ld(x, y))) / fc.pre_exposure));
                    const float t = saturate((log_lum - -16.0f) / (16.0f - -16.0f));
                    const uint32_t bin = std::min(f2u_sat(t * 256.0f), 255u);
                    const f2 uv = f2{float(x) + 0.5f, float(y) + 0.5f} / f2{float(ew), float(eh)};
                    const float l = length(uv - 0.5f);
                    const float infl = expf(-8.0f * powf(l, 2.0f));
                    hist[bin] += f2u_sat(infl * 256.0f);
                }
            auto& hb = surf["histogram"];
            hb.assign(sizeof hist, 0);
            memcpy(hb.data(), hist, sizeof hist);
            memcpy(histogram_readback, hist, sizeof hist);        // "_copy histogram"
        }
        // ---- rev_blur_pyramid (post.rs:63-110, rev_blur.rs)
        {
            ImgU32 top = mip("rev_blur_pyramid", mip_levels - 1, pw, ph);
            for (size_t i = 0; i < size_t(top.w) * top.h; ++i) top.p[i] = 0;
            for (int target = mip_levels - 2; target >= 0; --target) {
                const ImgU32 tail = mip("blur_pyramid", target, pw, ph), src = mip("rev_blur_pyramid", target + 1, pw, ph), dst = mip("rev_blur_pyramid", target, pw, ph);
                const float self_weight = 0.5f;                    // post.rs:73-77: src_mip never equals mip_levels
                const f2 inv_size = f2{1.0f, 1.0f} / f2{float(dst.w), float(dst.h)};
                for (int y = 0; y < dst.h; ++y)
                    for (int x = 0; x < dst.w; ++x) {
                        const f3 pyramid_col = unpack_r11g11b10f(tail.ld(x, y));
                        f3 self_col = mk3(0.0f);
                        for (int yy = -1; yy <= 1; ++yy)
                            for (int xx = -1; xx <= 1; ++xx) {
                                const f2 uv = (f2{float(x), float(y)} + f2{0.5f, 0.5f} + f2{float(xx), float(yy)}) * inv_size;
                                self_col += sample_r11g11b10f_bilinear_clamp(src, uv);
                            }
                        self_col = self_col / 9.0f;
                        dst.st(x, y, pack_r11g11b10f(lerp(self_col, pyramid_col, self_weight * 0.6f)));
                    }
            }
        }
        // ---- post combine (post_combine.hlsl:112-191)
        ImgU32 out = get<uint32_t>("output", W, H);
        const ImgU32 glare_tex = mip("rev_blur_pyramid", 0, pw, ph);
        const f2 inv_extent = f2{1.0f / float(W), 1.0f / float(H)};
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f2 uv = f2{float(x) + 0.5f, float(y) + 0.5f} * inv_extent;
                const f3 glare = sample_r11g11b10f_bilinear_clamp(glare_tex, uv);
                f3 col = in_rgb(x, y);
                col = lerp(col, glare, 0.05f);
                col = vmax(mk3(0.0f), col);
                col = col * post_exposure_mult;
                const float l = length(uv - 0.5f);
                col = col * expf(-2.0f * powf(l, 3.0f));      // pow(), as the text has it (post_combine.hlsl:156)
                col = display_transform_sRGB(bb_lut, col);
                col = vpow(col, contrast);
                const uint32_t idx = fc.frame_index;
                const uint32_t bx = (uint32_t(x) + idx * 59u) & 255u, by = (uint32_t(y) + idx * 37u) & 255u;
                const float dither = triangle_remap(float(blue_noise_rgba8[by * 256u + bx] & 0xffu) / 255.0f);
                col = col + dither / 256.0f;
                out.st(x, y, pack_r11g11b10f(col));
            }
        return out;
    }
    static f3 sample_r11g11b10f_bilinear_clamp(const ImgU32& i, f2 uv) {   // sampler_lnc
        const float fx = uv.x * float(i.w) - 0.5f, fy = uv.y * float(i.h) - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float tx = fx - x0f, ty = fy - y0f;
        const int x0 = f2i_sat(x0f), y0 = f2i_sat(y0f);
        auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
        const int xa = cl(x0, i.w), xb = cl(x0 + 1, i.w), ya = cl(y0, i.h), yb = cl(y0 + 1, i.h);
        const f3 s00 = unpack_r11g11b10f(i.p[size_t(ya) * i.w + xa]), s10 = unpack_r11g11b10f(i.p[size_t(ya) * i.w + xb]);
        const f3 s01 = unpack_r11g11b10f(i.p[size_t(yb) * i.w + xa]), s11 = unpack_r11g11b10f(i.p[size_t(yb) * i.w + xb]);
        const f3 a = s00 * (1.0f - tx) + s10 * tx;
        const f3 b = s01 * (1.0f - tx) + s11 * tx;
        return a * (1.0f - ty) + b * ty;
    }

    // PostProcessRenderer::read_back_histogram (post.rs:188-235), f64 as there
    static float read_back_histogram(const uint32_t* histogram, float clipping_low, float clipping_high) {
        const double outlier_frac_lo = std::min(double(clipping_low), 1.0) ;
        const double outlier_frac_hi = std::min(double(clipping_high), 1.0 - outlier_frac_lo);
        uint32_t total = 0;
        for (int i = 0; i < 256; ++i) total += histogram[i];
        const uint32_t reject_lo = uint32_t(double(total) * outlier_frac_lo);
        const uint32_t to_use = uint32_t(double(total) * (1.0 - outlier_frac_lo - outlier_frac_hi));
        double sum = 0.0;
        uint32_t used = 0, left_to_reject = reject_lo, left_to_use = to_use;
        for (int i = 0; i < 256; ++i) {
            const double t = (double(i) + 0.5) / 256.0;
            const uint32_t count = histogram[i];
            const uint32_t count_to_use = std::min(count > left_to_reject ? count - left_to_reject : 0u, left_to_use);
            left_to_reject = left_to_reject > count ? left_to_reject - count : 0u;
            left_to_use = left_to_use > count_to_use ? left_to_use - count_to_use : 0u;
            sum += t * double(count_to_use);
            used += count_to_use;
        }
        const double mean = sum / double(std::max(used, 1u));
        return float(-16.0 + mean * (16.0 - -16.0));
    }
};


// ---------------------------------------------------------------- motion_blur (renderers/motion_blur.rs:5-72; rust-shaders/src/motion_blur.rs —
// all four kernels are the Rust that runs). reprojection_map RGBA16_SNORM (xy = uv-space motion to the previous frame), depth R32F at
// (dw, dh); input / output RGBA16F at (w, h). Chosen where the Rust leaves it to the GPU: float -> uint casts saturate (negative / NaN ->
// 0), out-of-range fetches read 0, `sample_by_lod(.., 1.0)` of the single-mip input reads mip 0, output alpha = 1.
struct MotionBlur {
    std::map<std::string, std::vector<uint8_t>> surf;
    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    static float depth_to_view_z(float depth, const FrameConstants& fc) { return 1.0f / (depth * -fc.view_constants.clip_to_view[11]); }   // util.rs:69-76
    static f2 depth_cmp(float center_depth, float sample_depth, float depth_scale) {                                                       // motion_blur.rs:18-22
        const float d = sample_depth - center_depth;
        return f2{saturate(0.5f + depth_scale * d), saturate(0.5f + -depth_scale * d)};
    }
    static f2 spread_cmp(float offset_len, f2 spread_len) { return f2{saturate(spread_len.x - (offset_len + 1.0f)), saturate(spread_len.y - (offset_len + 1.0f))}; }
    static float sample_weight(float center_depth, float sample_depth, float offset_len, float center_spread_len, float sample_spread_len, float depth_scale) {
        return dot(depth_cmp(center_depth, sample_depth, depth_scale), spread_cmp(offset_len, f2{center_spread_len, sample_spread_len}));
    }
    // one step of the three "largest velocity" reductions: keep v when its squared length beats the running maximum
    static void keep_largest(f3& largest, f2 v) { const float m2 = dot(v, v); if (m2 > largest.z) largest = f3{v.x, v.y, m2}; }
    static f4 sample_bilinear_clamp_snorm16(const ImgRGBA16S& i, f2 uv) {
        const float fx = uv.x * float(i.w) - 0.5f, fy = uv.y * float(i.h) - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float tx = fx - x0f, ty = fy - y0f;
        const int x0 = f2i_sat(x0f), y0 = f2i_sat(y0f);
        auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
        const int xa = cl(x0, i.w), xb = cl(x0 + 1, i.w), ya = cl(y0, i.h), yb = cl(y0 + 1, i.h);
        const f4 a = ld_reproj(i, xa, ya) * (1.0f - tx) + ld_reproj(i, xb, ya) * tx;
        const f4 b = ld_reproj(i, xa, yb) * (1.0f - tx) + ld_reproj(i, xb, yb) * tx;
        return a * (1.0f - ty) + b * ty;
    }

    ImgRGBA16F render(const FrameConstants& fc, ImgRGBA16F input, ImgR32F depth, ImgRGBA16S reprojection_map) {
        const int W = input.w, H = input.h, DW = depth.w, DH = depth.h;
        const int tw = (DW + 15) / 16, th = (DH + 15) / 16;
        ImgRG16F reduced_x = get<h2>("velocity_reduced_x", tw, DH), reduced_y = get<h2>("velocity_reduced_y", tw, th), dilated = get<h2>("velocity_dilated", tw, th);
        for (int y = 0; y < DH; ++y)                                  // velocity_reduce_x (:189-211)
            for (int x = 0; x < tw; ++x) {
                f3 largest = mk3(0.0f);
                for (int i = 0; i < 16; ++i) { const f4 v = ld_reproj(reprojection_map, x * 16 + i, y); keep_largest(largest, f2{v.x, v.y}); }
                st2(reduced_x, x, y, f2{largest.x, largest.y});
            }
        for (int y = 0; y < th; ++y)                                  // velocity_reduce_y (:213-235)
            for (int x = 0; x < tw; ++x) {
                f3 largest = mk3(0.0f);
                for (int i = 0; i < 16; ++i) keep_largest(largest, ld2(reduced_x, x, y * 16 + i));
                st2(reduced_y, x, y, f2{largest.x, largest.y});
            }
        for (int y = 0; y < th; ++y)                                  // velocity_dilate (:237-264): x outer, y inner
            for (int x = 0; x < tw; ++x) {
                f3 largest = mk3(0.0f);
                for (int xx = -2; xx <= 2; ++xx)
                    for (int yy = -2; yy <= 2; ++yy) keep_largest(largest, ld2(reduced_y, x + xx, y + yy));
                st2(dilated, x, y, f2{largest.x, largest.y});
            }
        ImgRGBA16F out = get<h4>("output", W, H);
        const f2 depth_tex_size = f2{float(DW), float(DH)}, output_tex_size = f2{float(W), float(H)};
        const float blur_scale = 0.5f * 1.0f;                         // motion_blur_scale = 1.0 (motion_blur.rs:53)
        auto clampu = [](uint32_t v, uint32_t hi) { return v > hi ? hi : v; };
        for (int y = 0; y < H; ++y)                                   // motion_blur (:47-187)
            for (int x = 0; x < W; ++x) {
                const f2 uv = f2{float(x) + 0.5f, float(y) + 0.5f} * f2{1.0f / float(W), 1.0f / float(H)};
                int32_t tox = x, toy = y, noise1;
                {   // wrapping i32 arithmetic, arithmetic right shifts (the operands stay non-negative for any real extent)
                    uint32_t ux = uint32_t(tox), uy = uint32_t(toy);
                    ux += ux << 4; ux ^= uint32_t(int32_t(ux) >> 6);
                    uy += ux << 1; uy += uy << 6; uy ^= uint32_t(int32_t(uy) >> 2);
                    ux ^= uy;
                    noise1 = int32_t(ux ^ (uy << 1));
                    tox = int32_t(ux & 31u) - 15; toy = int32_t(uy & 31u) - 15;
                    noise1 = (noise1 & 31) - 15;
                }
                const f2 tile_coord_f = uv * depth_tex_size + f2{float(tox), float(toy)};
                const uint32_t tcx = clampu(f2u_sat(tile_coord_f.x), uint32_t(DW - 1)) / 16u, tcy = clampu(f2u_sat(tile_coord_f.y), uint32_t(DH - 1)) / 16u;
                const f2 tile_velocity = blur_scale * ld2(dilated, int(tcx), int(tcy));
                const int kernel_width = 4;
                const float noise = 0.5f * float(noise1) / 15.0f;
                const float center_offset_len = noise / float(kernel_width) * 0.5f;
                const f2 center_uv = uv + tile_velocity * center_offset_len;
                const f2 cpx = center_uv * output_tex_size;
                const f3 center_color = xyz(ld4(input, int(clampu(f2u_sat(cpx.x), uint32_t(W - 1))), int(clampu(f2u_sat(cpx.y), uint32_t(H - 1)))));
                const float center_depth = -depth_to_view_z(sample_nearest_clamp(depth, center_uv), fc);
                const f4 cv = sample_bilinear_clamp_snorm16(reprojection_map, center_uv);
                const f2 center_velocity_px = (blur_scale * f2{cv.x, cv.y}) * depth_tex_size;
                const float soft_z = 16.0f;
                f4 sum = mk4(0.0f);
                float sample_count = 1.0f;
                if (length(tile_velocity) > 0.0f) {
                    for (int i = 1; i < kernel_width; ++i) {
                        const float offset_len0 = (float(i) + noise) / float(kernel_width) * 0.5f;
                        const float offset_len1 = (float(-i) + noise) / float(kernel_width) * 0.5f;
                        const f2 uv0 = uv + tile_velocity * offset_len0, uv1 = uv + tile_velocity * offset_len1;
                        const f2 p0 = uv0 * depth_tex_size, p1 = uv1 * depth_tex_size;
                        const int px0 = int(std::min(f2u_sat(p0.x), 0x7fffffffu)), py0 = int(std::min(f2u_sat(p0.y), 0x7fffffffu));
                        const int px1 = int(std::min(f2u_sat(p1.x), 0x7fffffffu)), py1 = int(std::min(f2u_sat(p1.y), 0x7fffffffu));
                        const float d0 = -depth_to_view_z(depth.ld(px0, py0), fc), d1 = -depth_to_view_z(depth.ld(px1, py1), fc);
                        const f4 r0 = ld_reproj(reprojection_map, px0, py0), r1 = ld_reproj(reprojection_map, px1, py1);
                        const float v0 = length(blur_scale * f2{r0.x, r0.y} * depth_tex_size), v1 = length(blur_scale * f2{r1.x, r1.y} * depth_tex_size);
                        float weight0 = sample_weight(center_depth, d0, length((uv0 - uv) * depth_tex_size), length(center_velocity_px), v0, soft_z);
                        float weight1 = sample_weight(center_depth, d1, length((uv1 - uv) * depth_tex_size), length(center_velocity_px), v1, soft_z);
                        const bool m0 = d0 > d1, m1 = v1 > v0;
                        weight0 = (m0 && m1) ? weight1 : weight0;
                        weight1 = (m0 || m1) ? weight1 : weight0;
                        const float valid0 = (uv0.x == saturate(uv0.x) && uv0.y == saturate(uv0.y)) ? 1.0f : 0.0f;
                        const float valid1 = (uv1.x == saturate(uv1.x) && uv1.y == saturate(uv1.y)) ? 1.0f : 0.0f;
                        weight0 *= valid0; weight1 *= valid1;
                        sample_count += valid0 + valid1;
                        f4 c0 = sample_bilinear_clamp(input, uv0); c0.w = 1.0f;
                        sum += c0 * weight0;
                        f4 c1 = sample_bilinear_clamp(input, uv1); c1.w = 1.0f;
                        sum += c1 * weight1;
                    }
                    sum = sum * (1.0f / sample_count);
                }
                const f3 result = xyz(sum) + (1.0f - sum.w) * center_color;
                st4(out, x, y, mk4(result, 1.0f));
            }
        return out;
    }
};

}  // namespace okj
