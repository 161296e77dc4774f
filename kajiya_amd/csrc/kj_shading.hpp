// Device shading library: view-ray reconstruction, BRDFs + FG LUT, sun/atmosphere,
// cube-map sampling. Behaviour per inc/frame_constants.hlsl, inc/brdf.hlsl,
// inc/brdf_lut.hlsl, inc/layered_brdf.hlsl, lut/brdf_fg.hlsl, inc/sun.hlsl,
// inc/atmosphere.hlsl + atmosphere_felix.hlsl, inc/cube_map.hlsl.
#pragma once
#include "kj_vec.hpp"
#include "../../include/kajiya_amd.h"

namespace kj {

typedef KjFrameConstants FrameConstants;

// ---- inc/uv.hlsl
KJ_HD V2 get_uv(float px, float py, V4 ts) { return V2{(px + 0.5f) * ts.z, (py + 0.5f) * ts.w}; }
KJ_HD V2 cs_to_uv(V2 cs) { return V2{cs.x * 0.5f + 0.5f, cs.y * -0.5f + 0.5f}; }
KJ_HD V2 uv_to_cs(V2 uv) { return V2{(uv.x - 0.5f) * 2.0f, (uv.y - 0.5f) * -2.0f}; }
KJ_HD V4 tex_size4(int w, int h) { return V4{float(w), float(h), 1.0f / float(w), 1.0f / float(h)}; }

// ---- inc/frame_constants.hlsl:84-250. Only the members the kernels use are materialised.
struct ViewRay {
    V3 dir_ws;       // ray_dir_ws()
    V3 origin_ws;    // ray_origin_ws()
    V3 hit_vs;       // ray_hit_vs()
    V3 hit_ws;       // ray_hit_ws()
    V3 hit_cs;       // ray_hit_cs.xyz
    KJ_HD V3 biased_secondary_ray_origin_ws() const {   // frame_constants.hlsl:133-135
        return hit_ws - dir_ws * (length(hit_vs) + length(hit_ws)) * 1e-4f;
    }
    KJ_HD V3 biased_secondary_ray_origin_ws_with_normal(V3 normal) const {
        V3 ws_abs = vabs(hit_ws);
        float max_comp = fmaxf(fmaxf(ws_abs.x, ws_abs.y), fmaxf(ws_abs.z, -hit_vs.z));
        return hit_ws + (normal - dir_ws) * fmaxf(1e-4f, max_comp * 1e-6f);
    }
};
KJ_HD ViewRay view_ray_from_uv(const FrameConstants& fc, V2 uv) {
    const KjViewConstants& vc = fc.view_constants;
    ViewRay r;
    V2 cs = uv_to_cs(uv);
    V4 dir_vs_h = mul44(vc.sample_to_view, V4{cs.x, cs.y, 0.0f, 1.0f});
    V4 dir_ws_h = mul44(vc.view_to_world, dir_vs_h);
    r.dir_ws = normalize(xyz(dir_ws_h));
    V4 o_vs_h = mul44(vc.sample_to_view, V4{cs.x, cs.y, 1.0f, 1.0f});
    V4 o_ws_h = mul44(vc.view_to_world, o_vs_h);
    r.origin_ws = xyz(o_ws_h) / o_ws_h.w;
    r.hit_vs = r.hit_ws = r.hit_cs = v3(0.0f);
    return r;
}
KJ_HD ViewRay view_ray_from_uv_and_depth(const FrameConstants& fc, V2 uv, float depth) {
    const KjViewConstants& vc = fc.view_constants;
    ViewRay r = view_ray_from_uv(fc, uv);
    V2 cs = uv_to_cs(uv);
    r.hit_cs = V3{cs.x, cs.y, depth};
    V4 h_vs_h = mul44(vc.sample_to_view, V4{cs.x, cs.y, depth, 1.0f});
    V4 h_ws_h = mul44(vc.view_to_world, h_vs_h);
    r.hit_vs = xyz(h_vs_h) / h_vs_h.w;
    r.hit_ws = xyz(h_ws_h) / h_ws_h.w;
    return r;
}
KJ_HD ViewRay view_ray_from_uv_and_biased_depth(const FrameConstants& fc, V2 uv, float depth) {
    return view_ray_from_uv_and_depth(fc, uv, fminf(1.0f, depth * asfloat(0x3f800040u)));
}
KJ_HD V3 get_eye_position(const FrameConstants& fc) { V4 e = mul44(fc.view_constants.view_to_world, V4{0, 0, 0, 1}); return xyz(e) / e.w; }
KJ_HD V3 direction_view_to_world(const FrameConstants& fc, V3 v) { return xyz(mul44(fc.view_constants.view_to_world, v4(v, 0))); }
KJ_HD V3 direction_world_to_view(const FrameConstants& fc, V3 v) { return xyz(mul44(fc.view_constants.world_to_view, v4(v, 0))); }
KJ_HD V3 position_world_to_clip(const FrameConstants& fc, V3 v) {
    V4 p = mul44(fc.view_constants.view_to_clip, mul44(fc.view_constants.world_to_view, v4(v, 1)));
    return xyz(p) / p.w;
}
KJ_HD V3 position_world_to_sample(const FrameConstants& fc, V3 v) {
    V4 p = mul44(fc.view_constants.view_to_sample, mul44(fc.view_constants.world_to_view, v4(v, 1)));
    return xyz(p) / p.w;
}
KJ_HD I2 halfres_subsample_offset(uint32_t frame_index) {
    // hi_px_subpixels = {(1,1),(1,0),(0,0),(0,1)}[frame_index & 3] (frame_constants.hlsl:235-250)
    uint32_t i = frame_index & 3u;
    return I2{int(i < 2u), int(i == 0u || i == 3u)};
}
KJ_HD bool is_rtdgi_validation_frame(uint32_t frame_index) { return frame_index % 3u == 0u; }  // rtdgi_restir_settings.hlsl:41-46

// ---- inc/brdf.hlsl
struct BrdfValue { V3 value_over_pdf, value; float pdf; V3 transmission_fraction; };
struct BrdfSample { V3 value_over_pdf, value; float pdf; V3 transmission_fraction; V3 wi; float approx_roughness; };
KJ_HD BrdfValue brdf_value_invalid() { return BrdfValue{v3(0.0f), v3(0.0f), 0.0f, v3(0.0f)}; }
KJ_HD BrdfSample brdf_sample_invalid() { return BrdfSample{v3(0.0f), v3(0.0f), 0.0f, v3(0.0f), V3{0, 0, -1}, 0.0f}; }
KJ_HD V3 eval_fresnel_schlick(V3 f0, V3 f90, float cos_theta) { return lerp(f0, f90, powf(fmaxf(0.0f, 1.0f - cos_theta), 5.0f)); }
KJ_HD float g_smith_ggx_correlated(float ndotv, float ndotl, float a2) {
    float lambda_v = ndotl * sqrtf((-ndotv * a2 + ndotv) * ndotv + a2);
    float lambda_l = ndotv * sqrtf((-ndotl * a2 + ndotl) * ndotl + a2);
    return 2.0f * ndotl * ndotv / (lambda_v + lambda_l);
}
KJ_HD float g_smith_ggx1(float ndotv, float a2) {
    float tan2_v = (1.0f - ndotv * ndotv) / (ndotv * ndotv);
    return 2.0f / (1.0f + sqrtf(1.0f + a2 * tan2_v));
}
KJ_HD float ggx_ndf(float a2, float cos_theta) { float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f; return a2 / (KJ_PI * d * d); }
KJ_HD float pdf_ggx_vn(float a2, V3 wo, V3 h) { return g_smith_ggx1(wo.z, a2) * ggx_ndf(a2, h.z) * fmaxf(0.0f, dot(wo, h)) / wo.z; }
KJ_HD V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }

KJ_HD BrdfValue diffuse_evaluate(V3 albedo, V3 wi) {
    BrdfValue r;
    r.pdf = wi.z > 0.0f ? KJ_FRAC_1_PI : 0.0f;
    r.value_over_pdf = wi.z > 0.0f ? albedo : v3(0.0f);
    r.value = r.value_over_pdf * r.pdf;
    r.transmission_fraction = v3(0.0f);
    return r;
}
KJ_HD BrdfSample diffuse_sample(V3 albedo, V2 urand) {
    float phi = urand.x * KJ_TAU;
    float cos_theta = sqrtf(fmaxf(0.0f, 1.0f - urand.y));
    float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    BrdfSample r;
    r.wi = V3{cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
    r.pdf = KJ_FRAC_1_PI;
    r.value_over_pdf = albedo;
    r.value = albedo * r.pdf;
    r.transmission_fraction = v3(0.0f);
    r.approx_roughness = 1.0f;
    return r;
}
KJ_HD BrdfValue specular_evaluate(float roughness, V3 albedo, V3 wo, V3 wi) {
    if (wi.z <= 0.0f || wo.z <= 0.0f) return brdf_value_invalid();
    const float a2 = roughness * roughness;
    const V3 m = normalize(wo + wi);
    const float pdf_h = pdf_ggx_vn(a2, wo, m);
    const float jacobian = 1.0f / (4.0f * dot(wi, m));
    const V3 fresnel = eval_fresnel_schlick(albedo, v3(1.0f), dot(m, wi));
    const float g = g_smith_ggx_correlated(wo.z, wi.z, a2);
    BrdfValue r;
    r.pdf = pdf_h * jacobian / wi.z;
    r.transmission_fraction = v3(1.0f) - fresnel;
    r.value_over_pdf = fresnel * (g / g_smith_ggx1(wo.z, a2));
    r.value = fresnel * g * ggx_ndf(a2, m.z) / (4.0f * wo.z * wi.z);
    return r;
}
// brdf.hlsl:171-254 (VNDF sampling)
KJ_HD BrdfSample specular_sample(float roughness, V3 albedo, V3 wo, V2 urand) {
    const float alpha = roughness, a2 = alpha * alpha;
    V3 Vh = normalize(V3{alpha * wo.x, alpha * wo.y, wo.z});
    V3 T1 = (Vh.z < 0.9999f) ? normalize(cross(V3{0, 0, 1}, Vh)) : V3{1, 0, 0};
    V3 T2 = cross(Vh, T1);
    float r = sqrtf(urand.x);
    float phi = (2.0f * KJ_PI) * urand.y;
    float t1 = r * cosf(phi), t2 = r * sinf(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    const V3 m = normalize(V3{alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)});
    const float ndf_pdf = pdf_ggx_vn(a2, wo, m);
    const V3 wi = reflect(-wo, m);
    if (m.z <= 1e-5f || wi.z <= 1e-5f || wo.z <= 1e-5f) return brdf_sample_invalid();
    const float jacobian = 1.0f / (4.0f * dot(wi, m));
    const V3 fresnel = eval_fresnel_schlick(albedo, v3(1.0f), dot(m, wi));
    const float g = g_smith_ggx_correlated(wo.z, wi.z, a2);
    BrdfSample o;
    o.pdf = ndf_pdf * jacobian / wi.z;
    o.wi = wi;
    o.transmission_fraction = v3(1.0f) - fresnel;
    o.approx_roughness = roughness;
    o.value_over_pdf = fresnel * (g / g_smith_ggx1(wo.z, a2));
    o.value = fresnel * g * ggx_ndf(a2, m.z) / (4.0f * wo.z * wi.z);
    return o;
}
// lut/brdf_fg.hlsl:6-46
KJ_HD V3 integrate_brdf_fg(float roughness, float ndotv) {
    V3 wo{sqrtf(1.0f - ndotv * ndotv), 0, ndotv};
    float a = 0, b = 0, valid = 0;
    for (uint32_t i = 0; i < 1024u; ++i) {
        V2 urand = hammersley(i, 1024u);
        BrdfSample v_a = specular_sample(roughness, v3(1.0f), wo, urand);
        if (v_a.wi.z > 1e-6f) {
            BrdfValue v_b = specular_evaluate(roughness, v3(0.0f), wo, v_a.wi);
            a += (v_a.value_over_pdf.x - v_b.value_over_pdf.x);
            b += v_b.value_over_pdf.x;
            valid += 1;
        }
    }
    return V3{a, b, valid} / 1024.0f;
}

// inc/brdf_lut.hlsl:4-93 (active branch) + inc/layered_brdf.hlsl:11-100
struct LayeredBrdf {
    float roughness;
    V3 spec_albedo, diff_albedo;
    V3 preintegrated_reflection, preintegrated_reflection_mult, preintegrated_transmission_fraction;
};
KJ_D LayeredBrdf layered_brdf_from_gbuffer_ndotv(const uint2* __restrict__ fg_lut, const GbufferData& g, float ndotv) {
    LayeredBrdf r;
    r.roughness = g.roughness;
    V3 spec = v3(0.04f);
    const V3 albedo = g.albedo;
    spec = lerp(spec, albedo, g.metalness);
    V3 diff = fmaxf(0.0f, 1.0f - g.metalness) * albedo;
    const float x = g.metalness;
    const V3 y3 = albedo * albedo * albedo;
    const V3 boost = 1.0f + (0.25f - (x - 0.5f) * (x - 0.5f)) * (1.749f + -1.61f * fabsf(x - 0.5f)) * (0.5555f * albedo + 0.8244f * y3);
    r.spec_albedo = vmin(v3(1.0f), spec * boost);
    r.diff_albedo = vmin(v3(1.0f), diff * boost);
    const float s = 63.0f / 64.0f, b = 0.5f / 64.0f;
    V4 fg = sample_bilinear_clamp_rgba16f(fg_lut, 64, 64, V2{ndotv * s + b, r.roughness * s + b});
    V3 single_scatter = r.spec_albedo * fg.x + fg.y;
    float e_ss = fg.x + fg.y;
    V3 f_ss = single_scatter / e_ss;
    V3 f_ss_tail = lerp(f_ss, v3(1.0f), 0.4f);
    V3 bounce_radiance = (1.0f - e_ss) * f_ss_tail;
    V3 mult = 1.0f + bounce_radiance / (1.0f - bounce_radiance);
    r.preintegrated_reflection = single_scatter * mult;
    r.preintegrated_reflection_mult = mult;
    r.preintegrated_transmission_fraction = 1.0f - r.preintegrated_reflection;
    return r;
}
KJ_D V3 layered_brdf_evaluate(const LayeredBrdf& b, V3 wo, V3 wi) {
    if (wo.z <= 0 || wi.z <= 0) return v3(0.0f);
    const BrdfValue diff = diffuse_evaluate(b.diff_albedo, wi);
    const BrdfValue spec = specular_evaluate(b.roughness, b.spec_albedo, wo, wi);
    return spec.value * b.preintegrated_reflection_mult + diff.value * spec.transmission_fraction;
}
KJ_D V3 layered_brdf_evaluate_directional_light(const LayeredBrdf& b, V3 wo, V3 wi) {
    if (wo.z <= 0 || wi.z <= 0) return v3(0.0f);
    const BrdfValue diff = diffuse_evaluate(b.diff_albedo, wi);
    const BrdfValue spec = specular_evaluate(b.roughness, b.spec_albedo, wo, wi);
    const V3 m = lerp(v3(1.0f), b.preintegrated_reflection_mult, sqrtf(fabsf(wi.z)));
    return spec.value * m + diff.value * spec.transmission_fraction;
}
KJ_D BrdfSample layered_brdf_sample(const LayeredBrdf& b, V3 wo, V3 urand) {
    const float spec_wt = sRGB_to_luminance(b.preintegrated_reflection);
    const float diffuse_wt = sRGB_to_luminance(b.preintegrated_transmission_fraction * b.diff_albedo);
    const float transmission_p = diffuse_wt / (spec_wt + diffuse_wt);
    BrdfSample s;
    if (urand.z < transmission_p) {
        s = diffuse_sample(b.diff_albedo, V2{urand.x, urand.y});
        s.value_over_pdf = s.value_over_pdf / transmission_p;
        s.pdf *= transmission_p;
        s.value_over_pdf = s.value_over_pdf * b.preintegrated_transmission_fraction;
        s.value = s.value * b.preintegrated_transmission_fraction;
    } else {
        s = specular_sample(b.roughness, b.spec_albedo, wo, V2{urand.x, urand.y});
        const float lobe_pdf = 1.0f - transmission_p;
        s.value_over_pdf = s.value_over_pdf / lobe_pdf;
        s.pdf *= lobe_pdf;
        s.value_over_pdf = s.value_over_pdf * b.preintegrated_reflection_mult;
        s.value = s.value * b.preintegrated_reflection_mult;
    }
    return s;
}

// ---- atmosphere (inc/atmosphere_felix.hlsl:33-243, inc/atmosphere.hlsl:7-24, inc/sun.hlsl:21-41)
#define KJ_PLANET_RADIUS 6371000.0f
#define KJ_ATMOSPHERE_HEIGHT 100000.0f
KJ_HD V2 atmosphere_intersection(V3 ray_start, V3 ray_dir) {
    const float radius = KJ_PLANET_RADIUS + KJ_ATMOSPHERE_HEIGHT;
    ray_start = ray_start - V3{0, -KJ_PLANET_RADIUS, 0};
    float a = dot(ray_dir, ray_dir);
    float b = 2.0f * dot(ray_start, ray_dir);
    float c = dot(ray_start, ray_start) - (radius * radius);
    float d = b * b - 4 * a * c;
    if (d < 0) return V2{-1, -1};
    d = sqrtf(d);
    return V2{-b - d, -b + d} / (2 * a);
}
KJ_HD V3 atmosphere_density_at(V3 p) {
    float h = length(p - V3{0, -KJ_PLANET_RADIUS, 0}) - KJ_PLANET_RADIUS;
    return V3{expf(-fmaxf(0.0f, h / (KJ_ATMOSPHERE_HEIGHT * 0.08f))), expf(-fmaxf(0.0f, h / (KJ_ATMOSPHERE_HEIGHT * 0.012f))),
              fmaxf(0.0f, 1 - fabsf(h - 25000.0f) / 15000.0f)};
}
KJ_HD V3 integrate_optical_depth(V3 ray_start, V3 ray_dir) {
    float ray_length = atmosphere_intersection(ray_start, ray_dir).y;
    float step_size = ray_length / 8;
    V3 od = v3(0.0f);
    for (int i = 0; i < 8; i++) od += atmosphere_density_at(ray_start + ray_dir * (i + 0.5f) * step_size) * step_size;   // left to right, as the shader text associates
    return od;
}
KJ_HD V3 atmosphere_absorb(V3 od) {
    const V3 C_R = V3{5.802f, 13.558f, 33.100f} * 1e-6f, C_M = V3{3.996f, 3.996f, 3.996f} * 1e-6f, C_O = V3{0.650f, 1.881f, 0.085f} * 1e-6f;
    V3 e = -(od.x * C_R + od.y * C_M * 1.1f + od.z * C_O) * 1.0f;
    return V3{expf(e.x), expf(e.y), expf(e.z)};
}
KJ_HD V3 integrate_scattering(V3 ray_start, V3 ray_dir, float ray_length, V3 light_dir, V3 light_color) {
    const V3 C_R = V3{5.802f, 13.558f, 33.100f} * 1e-6f, C_M = V3{3.996f, 3.996f, 3.996f} * 1e-6f;
    V2 isect = atmosphere_intersection(ray_start, ray_dir);
    ray_length = fminf(ray_length, isect.y);
    if (isect.x > 0) { ray_start = ray_start + ray_dir * isect.x; ray_length -= isect.x; }
    float costh = dot(ray_dir, light_dir);
    float phase_r = 3 * (1 + costh * costh) / (16 * 3.14159265359f);
    float g = fminf(0.85f, 0.9381f);
    float k = 1.55f * g - 0.55f * g * g * g;
    float kcosth = k * costh;
    float phase_m = (1 - k * k) / ((4 * 3.14159265359f) * (1 - kcosth) * (1 - kcosth));
    V3 od = v3(0.0f), rayleigh = v3(0.0f), mie = v3(0.0f);
    float prev_t = 0;
    for (int i = 1; i <= 16; i++) {
        float t = powf(float(i) / 16, 5.0f) * ray_length;
        float step_size = (t - prev_t);
        V3 p = ray_start + ray_dir * lerp(prev_t, t, 0.5f);
        V3 dens = atmosphere_density_at(p);
        od += dens * step_size;
        V3 view_t = atmosphere_absorb(od);
        V3 light_t = atmosphere_absorb(integrate_optical_depth(p, light_dir));
        rayleigh += view_t * light_t * phase_r * dens.x * step_size;    // left to right, as the shader text associates
        mie += view_t * light_t * phase_m * dens.y * step_size;
        prev_t = t;
    }
    return (rayleigh * C_R + mie * C_M) * light_color * 20.0f;
}
KJ_HD V3 sun_direction(const FrameConstants& fc) { return V3{fc.sun_direction[0], fc.sun_direction[1], fc.sun_direction[2]}; }
KJ_HD V3 atmosphere_default(const FrameConstants& fc, V3 wi, V3 light_dir) {
    V3 sky_ambient{fc.sky_ambient[0], fc.sky_ambient[1], fc.sky_ambient[2]};
    V3 sun_mult{fc.sun_color_multiplier[0], fc.sun_color_multiplier[1], fc.sun_color_multiplier[2]};
    return (sky_ambient + sun_mult * integrate_scattering(v3(0.0f), wi, INFINITY, light_dir, v3(1.0f))) * fc.pre_exposure;
}
KJ_HD V3 sun_color_in_direction(const FrameConstants& fc, V3 dir) {
    V3 sun_mult{fc.sun_color_multiplier[0], fc.sun_color_multiplier[1], fc.sun_color_multiplier[2]};
    return 20.0f * sun_mult * fc.pre_exposure * atmosphere_absorb(integrate_optical_depth(v3(0.0f), dir));
}
KJ_HD V3 sample_sun_direction(const FrameConstants& fc, V2 urand, bool soft) {
    if (soft && fc.sun_angular_radius_cos < 1.0f) {
        return to_world(build_orthonormal_basis(normalize(sun_direction(fc))), uniform_sample_cone(urand, fc.sun_angular_radius_cos));
    }
    return sun_direction(fc);
}

// ---- cube maps (inc/cube_map.hlsl; Vulkan face selection; bilinear inside the face, clamp at edges)
KJ_HD V3 cube_face_dir(int face, V2 uv) {
    const float x = uv.x * 2 - 1, y = uv.y * 2 - 1;
    V3 d;
    switch (face) {
        case 0: d = V3{1.0f, -y, -x}; break;
        case 1: d = V3{-1.0f, -y, x}; break;
        case 2: d = V3{x, 1.0f, y}; break;
        case 3: d = V3{x, -1.0f, -y}; break;
        case 4: d = V3{x, -y, 1.0f}; break;
        default: d = V3{-x, -y, -1.0f}; break;
    }
    return normalize(d);
}
KJ_D V4 sample_cube_rgba16f(const uint2* __restrict__ cube, int width, V3 d) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) {
        if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; }
        ma = az;
    } else if (ay >= ax) {
        if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; }
        ma = ay;
    } else {
        if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; }
        ma = ax;
    }
    V2 uv{0.5f * (sc / ma + 1.0f), 0.5f * (tc / ma + 1.0f)};
    return sample_bilinear_clamp_rgba16f(cube + size_t(face) * width * width, width, width, uv);
}

// lights/triangle.hlsl:52-88
struct LightSampleArea { V3 pos, normal; float pdf; };
KJ_HD LightSampleArea sample_triangle_light(V3 v, V3 e0, V3 e1, V2 urand) {
    V3 perp = cross(e0, e1);
    float perp_inv_len = 1.0f / sqrtf(dot(perp, perp));
    float su0 = sqrtf(urand.x);
    LightSampleArea r;
    r.pos = v + (1.0f - su0) * e0 + (urand.y * su0) * e1;
    r.normal = perp * perp_inv_len;
    r.pdf = 2.0f * perp_inv_len;
    return r;
}

} // namespace kj
