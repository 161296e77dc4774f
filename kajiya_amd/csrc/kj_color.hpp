// Colour science of the display transform (assets/shaders/inc/color/{display_transform,bezold_brucke,helmholtz_kohlrausch,ipt,lab,luv,
// xyz,srgb,math}.hlsl with the switches as checked in: IPT perceptual space, Siragusano-Smith tone curve, Bezold-Brucke shift through
// the LUT, custom-G0 Helmholtz-Kohlrausch, long-tailed chroma attenuation, brightness-linear chroma attenuation). Device code only.
#pragma once
#include "kj_vec.hpp"

namespace kj {

#define KJ_WHITE_D65_X 0.31271f
#define KJ_WHITE_D65_Y 0.32902f

KJ_D V3 mul33r(float m0, float m1, float m2, float m3, float m4, float m5, float m6, float m7, float m8, V3 v) {   // float3x3(rows) * v
    return V3{m0 * v.x + m1 * v.y + m2 * v.z, m3 * v.x + m4 * v.y + m5 * v.z, m6 * v.x + m7 * v.y + m8 * v.z};
}
KJ_D V3 col_sRGB_to_XYZ(V3 c) { return mul33r(0.4124564f, 0.3575761f, 0.1804375f, 0.2126729f, 0.7151522f, 0.0721750f, 0.0193339f, 0.1191920f, 0.9503041f, c); }      // srgb.hlsl:11-17
KJ_D V3 col_XYZ_to_sRGB(V3 c) { return mul33r(3.2404542f, -1.5371385f, -0.4985314f, -0.9692660f, 1.8760108f, 0.0415560f, 0.0556434f, -0.2040259f, 1.0572252f, c); }  // srgb.hlsl:22-28
KJ_D V3 CIE_xyY_to_XYZ(V3 xyY) {        // xyz.hlsl:8-19
    const float X = (xyY.z / xyY.y) * xyY.x;
    const float Z = (xyY.z / xyY.y) * (1.0f - xyY.x - xyY.y);
    return V3{X, xyY.z, Z};
}
KJ_D V3 CIE_XYZ_to_xyY(V3 XYZ) {        // xyz.hlsl:21-34
    const float N = XYZ.x + XYZ.y + XYZ.z;
    return V3{XYZ.x / N, XYZ.y / N, XYZ.y};
}
KJ_D float spow(float x, float p) { return x >= 0.0f ? powf(x, p) : -powf(-x, p); }
KJ_D V3 XYZ_to_IPT(V3 c) {              // ipt.hlsl:4-25
    V3 lms = mul33r(0.4002f, 0.7075f, -0.0807f, -0.2280f, 1.1500f, 0.0612f, 0.0f, 0.0f, 0.9184f, c);
    lms = V3{spow(lms.x, 0.43f), spow(lms.y, 0.43f), spow(lms.z, 0.43f)};
    return mul33r(0.4000f, 0.4000f, 0.2000f, 4.4550f, -4.8510f, 0.3960f, 0.8056f, 0.3572f, -1.1628f, lms);
}
KJ_D V3 IPT_to_XYZ(V3 ipt) {            // ipt.hlsl:28-47
    V3 lms = mul33r(1.0f, 0.0976f, 0.2052f, 1.0f, -0.1139f, 0.1332f, 1.0f, 0.0326f, -0.6769f, ipt);
    const float e = 1.0f / 0.43f;
    lms = V3{spow(lms.x, e), spow(lms.y, e), spow(lms.z, e)};
    return mul33r(1.8501f, -1.1383f, 0.2385f, 0.3668f, 0.6439f, -0.0107f, 0.0f, 0.0f, 1.0889f, lms);
}
KJ_D V2 CIE_xyY_xy_to_LUV_uv(V2 xy) { return xy * V2{4.0f, 9.0f} / (-2.0f * xy.x + 12.0f * xy.y + 3.0f); }     // luv.hlsl:12-14
KJ_D V2 CIE_XYZ_to_LUV_uv(V3 v) { return V2{v.x, v.y} * V2{4.0f, 9.0f} / dot(v, V3{1.0f, 15.0f, 3.0f}); }        // luv.hlsl:16-18
KJ_D float col_catmull_rom(float x, float v0, float v1, float v2, float v3) {                                  // math.hlsl:14-19
    const float c2 = -.5f * v0 + 0.5f * v2;
    const float c3 = v0 + -2.5f * v1 + 2.0f * v2 + -.5f * v3;
    const float c4 = -.5f * v0 + 1.5f * v1 + -1.5f * v2 + 0.5f * v3;
    return ((c4 * x + c3) * x + c2) * x + v1;
}
// float -> uint as v_cvt_u32_f32 does it: NaN and negatives -> 0, saturating
KJ_D uint32_t f2u_sat(float f) { return f > 0.0f ? (f >= 4294967296.0f ? 0xffffffffu : uint32_t(f)) : 0u; }
// helmholtz_kohlrausch.hlsl:53-106 (HK_ADJUSTMENT_METHOD_CUSTOM_G0)
KJ_D float hk_q_sample(uint32_t i) {
    switch (i & 15u) {
        case 0: return -0.006f; case 1: return -0.021f; case 2: return -0.033f; case 3: return -0.009f;
        case 4: return 0.14f; case 5: return 0.114f; case 6: return 0.111f; case 7: return 0.1005f;
        case 8: return 0.069f; case 9: return 0.0135f; case 10: return -0.045f; case 11: return -0.075f;
        case 12: return -0.075f; case 13: return -0.03f; case 14: return 0.006f; default: return 0.006f;
    }
}
KJ_D float XYZ_to_hk_luminance_multiplier_custom_g0(V3 XYZ) {
    V2 uv = CIE_XYZ_to_LUV_uv(XYZ);
    const V2 d65_uv = CIE_xyY_xy_to_LUV_uv(V2{KJ_WHITE_D65_X, KJ_WHITE_D65_Y});
    uv = uv - d65_uv;
    const float theta = atan2f(uv.y, uv.x);
    const float t = (theta / 3.14159265358979323846f) * 0.5f + 0.5f;
    const uint32_t i0 = f2u_sat(floorf(t * 16.0f)) % 16u;
    const uint32_t i1 = (i0 + 1u) % 16u;
    const float q0 = hk_q_sample(i0 + 15u), q1 = hk_q_sample(i0), q2 = hk_q_sample(i1), q3 = hk_q_sample(i1 + 1u);
    const float interp = (t - float(i0) / 16.0f) * 16.0f;
    const float q = col_catmull_rom(interp, q0, q1, q2, q3);
    const float adapt_lum = 20.0f;
    const float kbr = 0.2717f * (6.469f + 6.362f * powf(adapt_lum, 0.4495f)) / (6.469f + powf(adapt_lum, 0.4495f));
    const float suv = 13.0f * length(uv);
    const float mult_cbrt = 1.0f + (q + 0.0872f * kbr) * suv;
    return mult_cbrt * mult_cbrt * mult_cbrt;
}
KJ_D float hk_from_sRGB(V3 stimulus) { return XYZ_to_hk_luminance_multiplier_custom_g0(col_sRGB_to_XYZ(stimulus)); }
KJ_D float srgb_to_equivalent_luminance(float hk_mult, V3 stimulus) { return hk_mult * col_sRGB_to_XYZ(stimulus).y; }
// bezold_brucke.hlsl:18-35 (BB_LUT_LUT_MAPPING_QUAD)
KJ_D float bb_xy_white_offset_to_lut_coord(V2 offset) {
    offset = offset / fmaxf(fabsf(offset.x), fabsf(offset.y));
    const float sgn = (offset.x + offset.y) > 0.0f ? 1.0f : -1.0f;
    return sgn * (0.125f * (offset.x - offset.y) + 0.25f);
}
// SAMPLE_BEZOLD_BRUCKE_LUT (post_combine.hlsl:7-10): bilinear + REPEAT over the 64x1 RG16F LUT; a NaN coordinate samples at 0
KJ_D V2 sample_bezold_brucke_lut(const uint32_t* __restrict__ lut, float coord) {
    if (!(coord == coord)) coord = 0.0f;
    const float fx = coord * 64.0f - 0.5f;
    const float x0f = floorf(fx), tx = fx - x0f;
    const int x0 = f2i_sat(x0f);
    const int xa = ((x0 % 64) + 64) % 64, xb = (xa + 1) % 64;
    const V2 a = unpack_2x16f_uint(lut[xa]), b = unpack_2x16f_uint(lut[xb]);
    return a * (1.0f - tx) + b * tx;
}
// bezold_brucke.hlsl:138-149
KJ_D V3 bezold_brucke_shift_XYZ_with_lut(const uint32_t* __restrict__ lut, V3 XYZ, float amount) {
    const V3 xyY = CIE_XYZ_to_xyY(XYZ);
    const V2 offset = V2{xyY.x, xyY.y} - V2{KJ_WHITE_D65_X, KJ_WHITE_D65_Y};
    const float lut_coord = bb_xy_white_offset_to_lut_coord(offset);
    const V2 shifted_xy = V2{xyY.x, xyY.y} + sample_bezold_brucke_lut(lut, lut_coord) * length(offset) * amount;
    return CIE_xyY_to_XYZ(V3{shifted_xy.x, shifted_xy.y, xyY.z});
}
// display_transform.hlsl:67-83 (BRIGHTNESS_COMPRESSION_CURVE_SIRAGUSANO_SMITH)
KJ_D float compress_luminance(float v) { return saturate(1.0205f * powf(v / (v + 1.0f), 1.2f)); }
KJ_D V3 vpow(V3 v, float p) { return V3{powf(v.x, p), powf(v.y, p), powf(v.z, p)}; }
// display_transform.hlsl:85-216
KJ_D V3 display_transform_sRGB(const uint32_t* __restrict__ bb_lut, V3 input_stimulus) {
    {
        const float t = sRGB_to_luminance(input_stimulus) / 5.0f;
        const float shift_amount = t / (t + 1.0f);
        input_stimulus = col_XYZ_to_sRGB(bezold_brucke_shift_XYZ_with_lut(bb_lut, col_sRGB_to_XYZ(input_stimulus), shift_amount));
    }
    const float hk = hk_from_sRGB(input_stimulus);
    const float input_equiv_lum = srgb_to_equivalent_luminance(hk, input_stimulus);
    const V3 max_intensity_rgb = input_stimulus / max3(input_stimulus.x, input_stimulus.y, input_stimulus.z);
    const float max_intensity_equiv_lum = srgb_to_equivalent_luminance(hk, max_intensity_rgb);
    const float max_output_scale = 1.0f;
    const float compressed_achromatic_luminance = compress_luminance(input_equiv_lum / max_output_scale) * max_output_scale;
    V3 compressed_rgb = (max_intensity_rgb / max_intensity_equiv_lum) * compressed_achromatic_luminance;
    const float clamped_compressed_achromatic_luminance = fminf(1.0f, compressed_achromatic_luminance);
    const V3 perceptual = XYZ_to_IPT(col_sRGB_to_XYZ(compressed_rgb));
    const V3 perceptual_white = XYZ_to_IPT(col_sRGB_to_XYZ(v3(clamped_compressed_achromatic_luminance)));
    // chroma_strength (:149) only feeds chroma_attenuation_exponent, which the long-tailed branch does not use
    const float chroma_attenuation_start = 0.0f;
    const float chroma_attenuation_t = saturate(
        (compressed_achromatic_luminance - fminf(1.0f, max_intensity_equiv_lum) * chroma_attenuation_start) /
        (1.03f * max_output_scale - fminf(1.0f, max_intensity_equiv_lum) * chroma_attenuation_start));
    float chroma_attenuation = asinf(powf(chroma_attenuation_t, 3.0f)) / 3.14159265358979323846f * 2.0f;      // pow(), as the text has it (:161)
    {
        const float compressed_achromatic_luminance2 = compress_luminance(0.125f * input_equiv_lum / max_output_scale) * max_output_scale;
        const float chroma_attenuation_t2 = saturate((compressed_achromatic_luminance2 - fminf(1.0f, max_intensity_equiv_lum) * 0.5f) /
                                                     (max_output_scale - fminf(1.0f, max_intensity_equiv_lum) * 0.5f));
        chroma_attenuation = lerp(chroma_attenuation, 1.0f, 1.0f - saturate(1.0f - powf(chroma_attenuation_t2, 4.0f)));
    }
    {
        const V3 perceptual_mid = lerp(perceptual, perceptual_white, chroma_attenuation);
        compressed_rgb = col_XYZ_to_sRGB(IPT_to_XYZ(perceptual_mid));
        const float hk2 = hk_from_sRGB(compressed_rgb);
        for (int i = 0; i < 2; ++i) {
            const float current_brightness = srgb_to_equivalent_luminance(hk2, compressed_rgb);
            compressed_rgb = compressed_rgb * (compressed_achromatic_luminance / fmaxf(1e-10f, current_brightness));
        }
    }
    compressed_rgb = vmax(compressed_rgb, v3(0.0f));
    const float p = 12.0f;
    compressed_rgb = compressed_rgb * vpow(vpow(compressed_rgb, p) + v3(1.0f), -1.0f / p);
    const float max_comp = max3(compressed_rgb.x, compressed_rgb.y, compressed_rgb.z);
    const float max_comp_dist = max3(max_comp - compressed_rgb.x, max_comp - compressed_rgb.y, max_comp - compressed_rgb.z);
    compressed_rgb = compressed_rgb / powf(lerp(0.5f, 1.0f, max_comp_dist), 1.0f / p);
    return compressed_rgb;
}
// post_combine.hlsl:44-50
KJ_D float triangle_remap(float n) {
    const float origin = n * 2.0f - 1.0f;
    float v = origin * (1.0f / sqrtf(fabsf(origin)));
    v = fmaxf(-1.0f, v);
    v -= origin > 0.0f ? 1.0f : (origin < 0.0f ? -1.0f : 0.0f);
    return v;
}

}  // namespace kj
