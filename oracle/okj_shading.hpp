// ORACLE (test infrastructure). View math, BRDFs, sun/atmosphere, sky cube.
// Restates inc/frame_constants.hlsl, inc/brdf.hlsl, inc/brdf_lut.hlsl,
// inc/layered_brdf.hlsl, lut/brdf_fg.hlsl, inc/sun.hlsl, inc/atmosphere*.hlsl,
// inc/cube_map.hlsl, sky/comp_cube.hlsl, convolve_cube.hlsl.
#pragma once
#include "okj_math.hpp"
#include "../include/kajiya_amd.h"
#include <vector>

namespace okj {

typedef KjFrameConstants FrameConstants;

// ------------------------------------------------------------------ uv.hlsl
static inline f2 get_uv(float px, float py, f4 tex_size) { return f2{(px + 0.5f) * tex_size.z, (py + 0.5f) * tex_size.w}; }
static inline f2 cs_to_uv(f2 cs) { return f2{cs.x * 0.5f + 0.5f, cs.y * -0.5f + 0.5f}; }
static inline f2 uv_to_cs(f2 uv) { return f2{(uv.x - 0.5f) * 2.0f, (uv.y - 0.5f) * -2.0f}; }
static inline f4 tex_size4(uint32_t w, uint32_t h) { return f4{float(w), float(h), 1.0f / float(w), 1.0f / float(h)}; }

// ------------------------------------------------------------------ frame_constants.hlsl:84-250
struct ViewRayContext {
    f4 ray_dir_cs, ray_dir_vs_h, ray_dir_ws_h;
    f4 ray_origin_cs, ray_origin_vs_h, ray_origin_ws_h;
    f4 ray_hit_cs, ray_hit_vs_h, ray_hit_ws_h;

    f3 ray_dir_vs() const { return normalize(xyz(ray_dir_vs_h)); }
    f3 ray_dir_ws() const { return normalize(xyz(ray_dir_ws_h)); }
    f3 ray_origin_ws() const { return xyz(ray_origin_ws_h) / ray_origin_ws_h.w; }
    f3 ray_hit_vs() const { return xyz(ray_hit_vs_h) / ray_hit_vs_h.w; }
    f3 ray_hit_ws() const { return xyz(ray_hit_ws_h) / ray_hit_ws_h.w; }

    f3 biased_secondary_ray_origin_ws() const {   // frame_constants.hlsl:133-135
        return ray_hit_ws() - ray_dir_ws() * (length(ray_hit_vs()) + length(ray_hit_ws())) * 1e-4f;
    }
    f3 biased_secondary_ray_origin_ws_with_normal(f3 normal) const {
        f3 ws_abs = vabs(ray_hit_ws());
        float max_comp = fmaxf(fmaxf(ws_abs.x, ws_abs.y), fmaxf(ws_abs.z, -ray_hit_vs().z));
        return ray_hit_ws() + (normal - ray_dir_ws()) * fmaxf(1e-4f, max_comp * 1e-6f);
    }

    static ViewRayContext from_uv(const FrameConstants& fc, f2 uv) {
        const KjViewConstants& vc = fc.view_constants;
        ViewRayContext r{};
        f2 cs = uv_to_cs(uv);
        r.ray_dir_cs = f4{cs.x, cs.y, 0.0f, 1.0f};
        r.ray_dir_vs_h = mul44(vc.sample_to_view, r.ray_dir_cs);
        r.ray_dir_ws_h = mul44(vc.view_to_world, r.ray_dir_vs_h);
        r.ray_origin_cs = f4{cs.x, cs.y, 1.0f, 1.0f};
        r.ray_origin_vs_h = mul44(vc.sample_to_view, r.ray_origin_cs);
        r.ray_origin_ws_h = mul44(vc.view_to_world, r.ray_origin_vs_h);
        return r;
    }
    static ViewRayContext from_uv_and_depth(const FrameConstants& fc, f2 uv, float depth) {
        const KjViewConstants& vc = fc.view_constants;
        ViewRayContext r = from_uv(fc, uv);
        f2 cs = uv_to_cs(uv);
        r.ray_hit_cs = f4{cs.x, cs.y, depth, 1.0f};
        r.ray_hit_vs_h = mul44(vc.sample_to_view, r.ray_hit_cs);
        r.ray_hit_ws_h = mul44(vc.view_to_world, r.ray_hit_vs_h);
        return r;
    }
    static ViewRayContext from_uv_and_biased_depth(const FrameConstants& fc, f2 uv, float depth) {
        return from_uv_and_depth(fc, uv, fminf(1.0f, depth * asfloat(0x3f800040u)));
    }
};

static inline f3 get_eye_position(const FrameConstants& fc) {
    f4 e = mul44(fc.view_constants.view_to_world, f4{0, 0, 0, 1});
    return xyz(e) / e.w;
}
static inline f3 get_prev_eye_position(const FrameConstants& fc) {
    f4 e = mul44(fc.view_constants.prev_view_to_prev_world, f4{0, 0, 0, 1});
    return xyz(e) / e.w;
}
// clip_to_view._43 = row 4 (1-based) col 3 => column-major index [2*4 + 3]
static inline float depth_to_view_z(const FrameConstants& fc, float depth) {
    return 1.0f / (depth * -fc.view_constants.clip_to_view[11]);
}
static inline f3 direction_view_to_world(const FrameConstants& fc, f3 v) { return xyz(mul44(fc.view_constants.view_to_world, mk4(v, 0))); }
static inline f3 direction_world_to_view(const FrameConstants& fc, f3 v) { return xyz(mul44(fc.view_constants.world_to_view, mk4(v, 0))); }
static inline f3 position_world_to_view(const FrameConstants& fc, f3 v) { return xyz(mul44(fc.view_constants.world_to_view, mk4(v, 1))); }
static inline f3 position_world_to_clip(const FrameConstants& fc, f3 v) {
    f4 p = mul44(fc.view_constants.world_to_view, mk4(v, 1));
    p = mul44(fc.view_constants.view_to_clip, p);
    return xyz(p) / p.w;
}
static inline f3 position_world_to_sample(const FrameConstants& fc, f3 v) {
    f4 p = mul44(fc.view_constants.world_to_view, mk4(v, 1));
    p = mul44(fc.view_constants.view_to_sample, p);
    return xyz(p) / p.w;
}
static inline float pixel_cone_spread_angle_from_image_height(const FrameConstants& fc, float image_height) {
    return atanf(2.0f * fc.view_constants.clip_to_view[0] / image_height);
}
static const int HI_PX_SUBPIXELS[4][2] = {{1, 1}, {1, 0}, {0, 0}, {0, 1}};
static inline i2 halfres_subsample_offset(const FrameConstants& fc) {
    const int* o = HI_PX_SUBPIXELS[fc.frame_index & 3];
    return i2{o[0], o[1]};
}

// ray_cone.hlsl
struct RayCone {
    float width, spread_angle;
    RayCone propagate(float surface_spread_angle, float hit_t) const {
        return RayCone{spread_angle * hit_t + width, spread_angle + surface_spread_angle};
    }
    float width_at_t(float t) const { return width + spread_angle * t; }
};

static inline RayCone pixel_ray_cone_from_image_height(const FrameConstants& fc, float image_height) {   // frame_constants.hlsl:227-233
    return RayCone{0.0f, pixel_cone_spread_angle_from_image_height(fc, image_height)};
}

// ------------------------------------------------------------------ brdf.hlsl
static const float BRDF_SAMPLING_MIN_COS = 1e-5f;
struct BrdfValue { f3 value_over_pdf{0, 0, 0}; f3 value{0, 0, 0}; float pdf = 0; f3 transmission_fraction{0, 0, 0}; };
struct BrdfSample : BrdfValue {
    f3 wi{0, 0, -1};
    float approx_roughness = 0;
    bool is_valid() const { return wi.z > 1e-6f; }
};

static inline f3 eval_fresnel_schlick(f3 f0, f3 f90, float cos_theta) {
    return lerp(f0, f90, powf(fmaxf(0.0f, 1.0f - cos_theta), 5.0f));
}
static inline float g_smith_ggx_correlated(float ndotv, float ndotl, float a2) {
    float lambda_v = ndotl * sqrtf((-ndotv * a2 + ndotv) * ndotv + a2);
    float lambda_l = ndotv * sqrtf((-ndotl * a2 + ndotl) * ndotl + a2);
    return 2.0f * ndotl * ndotv / (lambda_v + lambda_l);
}
static inline float g_smith_ggx1(float ndotv, float a2) {
    float tan2_v = (1.0f - ndotv * ndotv) / (ndotv * ndotv);
    return 2.0f / (1.0f + sqrtf(1.0f + a2 * tan2_v));
}
static inline float ggx_ndf(float a2, float cos_theta) {
    float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
    return a2 / (M_PI_F * d * d);
}
static inline float pdf_ggx_vn(float a2, f3 wo, f3 h) {
    float g1 = g_smith_ggx1(wo.z, a2);
    float d = ggx_ndf(a2, h.z);
    return g1 * d * fmaxf(0.0f, dot(wo, h)) / wo.z;
}
static inline f3 reflect(f3 i, f3 n) { return i - 2.0f * dot(n, i) * n; }

struct DiffuseBrdf {
    f3 albedo;
    BrdfValue evaluate(f3, f3 wi) const {
        BrdfValue r;
        r.pdf = wi.z > 0.0f ? M_FRAC_1_PI_F : 0.0f;
        r.value_over_pdf = wi.z > 0.0f ? albedo : mk3(0.0f);
        r.value = r.value_over_pdf * r.pdf;
        r.transmission_fraction = mk3(0.0f);
        return r;
    }
    BrdfSample sample(f3, f2 urand) const {
        float phi = urand.x * M_TAU_F;
        float cos_theta = sqrtf(fmaxf(0.0f, 1.0f - urand.y));
        float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        BrdfSample r;
        r.wi = f3{cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
        r.pdf = M_FRAC_1_PI_F;
        r.value_over_pdf = albedo;
        r.value = r.value_over_pdf * r.pdf;
        r.transmission_fraction = mk3(0.0f);
        r.approx_roughness = 1.0f;
        return r;
    }
};

struct SpecularBrdf {
    float roughness;
    f3 albedo;

    // brdf.hlsl:171-205 (VNDF sampling)
    void sample_vndf(float alpha, f3 wo, f2 urand, f3& m, float& pdf) const {
        float a2 = alpha * alpha;
        f3 Vh = normalize(f3{alpha * wo.x, alpha * wo.y, wo.z});
        f3 T1 = (Vh.z < 0.9999f) ? normalize(cross(f3{0, 0, 1}, Vh)) : f3{1, 0, 0};
        f3 T2 = cross(Vh, T1);
        float r = sqrtf(urand.x);
        float phi = (2.0f * M_PI_F) * urand.y;
        float t1 = r * cosf(phi);
        float t2 = r * sinf(phi);
        float s = 0.5f * (1.0f + Vh.z);
        t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
        f3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
        m = normalize(f3{alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)});
        pdf = pdf_ggx_vn(a2, wo, m);
    }

    BrdfSample sample(f3 wo, f2 urand) const {
        f3 m; float ndf_pdf;
        sample_vndf(roughness, wo, urand, m, ndf_pdf);
        const f3 wi = reflect(-wo, m);
        if (m.z <= BRDF_SAMPLING_MIN_COS || wi.z <= BRDF_SAMPLING_MIN_COS || wo.z <= BRDF_SAMPLING_MIN_COS) {
            return BrdfSample();
        }
        const float jacobian = 1.0f / (4.0f * dot(wi, m));
        const f3 fresnel = eval_fresnel_schlick(albedo, mk3(1.0f), dot(m, wi));
        const float a2 = roughness * roughness;
        const float cos_theta = m.z;
        float g = g_smith_ggx_correlated(wo.z, wi.z, a2);
        float g_over_g1_wo = g / g_smith_ggx1(wo.z, a2);
        BrdfSample r;
        r.pdf = ndf_pdf * jacobian / wi.z;
        r.wi = wi;
        r.transmission_fraction = mk3(1.0f) - fresnel;
        r.approx_roughness = roughness;
        r.value_over_pdf = fresnel * g_over_g1_wo;
        r.value = fresnel * g * ggx_ndf(a2, cos_theta) / (4.0f * wo.z * wi.z);
        return r;
    }

    BrdfValue evaluate(f3 wo, f3 wi) const {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return BrdfValue();
        const float a2 = roughness * roughness;
        const f3 m = normalize(wo + wi);
        const float cos_theta = m.z;
        const float pdf_h = pdf_ggx_vn(a2, wo, m);
        const float jacobian = 1.0f / (4.0f * dot(wi, m));
        const f3 fresnel = eval_fresnel_schlick(albedo, mk3(1.0f), dot(m, wi));
        float g = g_smith_ggx_correlated(wo.z, wi.z, a2);
        float g_over_g1_wo = g / g_smith_ggx1(wo.z, a2);
        BrdfValue r;
        r.pdf = pdf_h * jacobian / wi.z;
        r.transmission_fraction = mk3(1.0f) - fresnel;
        r.value_over_pdf = fresnel * g_over_g1_wo;
        r.value = fresnel * g * ggx_ndf(a2, cos_theta) / (4.0f * wo.z * wi.z);
        return r;
    }
};

// lut/brdf_fg.hlsl — 64x64 LUT, stored RGBA16F (image_lut.rs), sampled with sampler_lnc
static inline f3 integrate_brdf_fg(float roughness, float ndotv) {
    f3 wo{sqrtf(1.0f - ndotv * ndotv), 0, ndotv};
    float a = 0, b = 0, valid = 0;
    SpecularBrdf brdf_a{roughness, mk3(1.0f)};
    SpecularBrdf brdf_b{roughness, mk3(0.0f)};
    const uint32_t num_samples = 1024;
    for (uint32_t i = 0; i < num_samples; ++i) {
        f2 urand = hammersley(i, num_samples);
        BrdfSample v_a = brdf_a.sample(wo, urand);
        if (v_a.is_valid()) {
            BrdfValue v_b = brdf_b.evaluate(wo, v_a.wi);
            a += (v_a.value_over_pdf.x - v_b.value_over_pdf.x);
            b += v_b.value_over_pdf.x;
            valid += 1;
        }
    }
    return f3{a, b, valid} / float(num_samples);
}
static inline void build_brdf_fg_lut(h4* out /*64*64*/) {
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x) {
            float ndotv = (float(x) / (64.0f - 1.0f)) * (1.0f - 1e-3f) + 1e-3f;
            float roughness = fmaxf(1e-5f, float(y) / (64.0f - 1.0f));
            f3 v = integrate_brdf_fg(roughness, ndotv);
            out[y * 64 + x] = pack_rgba16f(mk4(v, 0.0f));
        }
}

// Bilinear, clamp-to-edge sample of an RGBA16F image at normalised uv (sampler_lnc).
static inline f4 sample_bilinear_clamp_rgba16f(const h4* img, int w, int h, f2 uv) {
    float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    int xa = cl(x0, w), xb = cl(x0 + 1, w), ya = cl(y0, h), yb = cl(y0 + 1, h);
    f4 s00 = unpack_rgba16f(img[ya * w + xa]), s10 = unpack_rgba16f(img[ya * w + xb]);
    f4 s01 = unpack_rgba16f(img[yb * w + xa]), s11 = unpack_rgba16f(img[yb * w + xb]);
    f4 a = s00 * (1.0f - tx) + s10 * tx;
    f4 b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}

// brdf_lut.hlsl:4-93 (the active `#elif 1` branch)
struct SpecularBrdfEnergyPreservation {
    f3 preintegrated_reflection, preintegrated_reflection_mult, preintegrated_transmission_fraction;
    float valid_sample_fraction;
    static SpecularBrdfEnergyPreservation from_brdf_ndotv(const h4* fg_lut, const SpecularBrdf& brdf, float ndotv) {
        const float s = 63.0f / 64.0f, b = 0.5f / 64.0f;
        f2 uv{ndotv * s + b, brdf.roughness * s + b};
        f4 fg = sample_bilinear_clamp_rgba16f(fg_lut, 64, 64, uv);
        f3 single_scatter = brdf.albedo * fg.x + fg.y;
        SpecularBrdfEnergyPreservation r;
        r.valid_sample_fraction = fg.z;
        float e_ss = fg.x + fg.y;
        f3 f_ss = single_scatter / e_ss;
        f3 f_ss_tail = lerp(f_ss, mk3(1.0f), 0.4f);
        f3 bounce_radiance = (1.0f - e_ss) * f_ss_tail;
        f3 mult = 1.0f + bounce_radiance / (1.0f - bounce_radiance);
        r.preintegrated_reflection = single_scatter * mult;
        r.preintegrated_reflection_mult = mult;
        r.preintegrated_transmission_fraction = 1.0f - r.preintegrated_reflection;
        return r;
    }
};

// layered_brdf.hlsl
static inline f3 metalness_albedo_boost(float metalness, f3 diffuse_albedo) {
    const float a0 = 1.749f, a1 = -1.61f, e1 = 0.5555f, e3 = 0.8244f;
    const float x = metalness;
    const f3 y = diffuse_albedo;
    const f3 y3 = y * y * y;
    return 1.0f + (0.25f - (x - 0.5f) * (x - 0.5f)) * (a0 + a1 * fabsf(x - 0.5f)) * (e1 * y + e3 * y3);
}
struct LayeredBrdf {
    SpecularBrdf specular_brdf;
    DiffuseBrdf diffuse_brdf;
    SpecularBrdfEnergyPreservation energy_preservation;

    static LayeredBrdf from_gbuffer_ndotv(const h4* fg_lut, const GbufferData& g, float ndotv) {
        SpecularBrdf spec{g.roughness, mk3(0.04f)};
        DiffuseBrdf diff{g.albedo};
        const f3 albedo = diff.albedo;
        spec.albedo = lerp(spec.albedo, albedo, g.metalness);
        diff.albedo = fmaxf(0.0f, 1.0f - g.metalness) * albedo;
        const f3 boost = metalness_albedo_boost(g.metalness, albedo);
        spec.albedo = vmin(mk3(1.0f), spec.albedo * boost);
        diff.albedo = vmin(mk3(1.0f), diff.albedo * boost);
        LayeredBrdf r;
        r.energy_preservation = SpecularBrdfEnergyPreservation::from_brdf_ndotv(fg_lut, spec, ndotv);
        r.specular_brdf = spec;
        r.diffuse_brdf = diff;
        return r;
    }
    f3 evaluate(f3 wo, f3 wi) const {
        if (wo.z <= 0 || wi.z <= 0) return mk3(0.0f);
        const BrdfValue diff = diffuse_brdf.evaluate(wo, wi);
        const BrdfValue spec = specular_brdf.evaluate(wo, wi);
        return spec.value * energy_preservation.preintegrated_reflection_mult + diff.value * spec.transmission_fraction;
    }
    f3 evaluate_directional_light(f3 wo, f3 wi) const {
        if (wo.z <= 0 || wi.z <= 0) return mk3(0.0f);
        const BrdfValue diff = diffuse_brdf.evaluate(wo, wi);
        const BrdfValue spec = specular_brdf.evaluate(wo, wi);
        const f3 m = lerp(mk3(1.0f), energy_preservation.preintegrated_reflection_mult, sqrtf(fabsf(wi.z)));
        return spec.value * m + diff.value * spec.transmission_fraction;
    }
    BrdfSample sample(f3 wo, f3 urand) const {
        BrdfSample s;
        const float spec_wt = sRGB_to_luminance(energy_preservation.preintegrated_reflection);
        const float diffuse_wt = sRGB_to_luminance(energy_preservation.preintegrated_transmission_fraction * diffuse_brdf.albedo);
        const float transmission_p = diffuse_wt / (spec_wt + diffuse_wt);
        const float lobe_xi = urand.z;
        if (lobe_xi < transmission_p) {
            s = diffuse_brdf.sample(wo, f2{urand.x, urand.y});
            const float lobe_pdf = transmission_p;
            s.value_over_pdf = s.value_over_pdf / lobe_pdf;
            s.pdf *= lobe_pdf;
            s.value_over_pdf = s.value_over_pdf * energy_preservation.preintegrated_transmission_fraction;
            s.value = s.value * energy_preservation.preintegrated_transmission_fraction;
        } else {
            s = specular_brdf.sample(wo, f2{urand.x, urand.y});
            const float lobe_pdf = (1.0f - transmission_p);
            s.value_over_pdf = s.value_over_pdf / lobe_pdf;
            s.pdf *= lobe_pdf;
            s.value_over_pdf = s.value_over_pdf * energy_preservation.preintegrated_reflection_mult;
            s.value = s.value * energy_preservation.preintegrated_reflection_mult;
        }
        return s;
    }
};

// ------------------------------------------------------------------ atmosphere_felix.hlsl / atmosphere.hlsl / sun.hlsl
namespace atm {
static const float PLANET_RADIUS = 6371000.0f;
static const float ATMOSPHERE_HEIGHT = 100000.0f;
static const float RAYLEIGH_HEIGHT = ATMOSPHERE_HEIGHT * 0.08f;
static const float MIE_HEIGHT = ATMOSPHERE_HEIGHT * 0.012f;
static inline f3 planet_center() { return f3{0, -PLANET_RADIUS, 0}; }
static inline f3 C_RAYLEIGH() { return f3{5.802f, 13.558f, 33.100f} * 1e-6f; }
static inline f3 C_MIE() { return f3{3.996f, 3.996f, 3.996f} * 1e-6f; }
static inline f3 C_OZONE() { return f3{0.650f, 1.881f, 0.085f} * 1e-6f; }

static inline f2 sphere_intersection(f3 ray_start, f3 ray_dir, f3 center, float radius) {
    ray_start = ray_start - center;
    float a = dot(ray_dir, ray_dir);
    float b = 2.0f * dot(ray_start, ray_dir);
    float c = dot(ray_start, ray_start) - (radius * radius);
    float d = b * b - 4 * a * c;
    if (d < 0) return f2{-1, -1};
    d = sqrtf(d);
    return f2{-b - d, -b + d} / (2 * a);
}
static inline f2 atmosphere_intersection(f3 s, f3 d) { return sphere_intersection(s, d, planet_center(), PLANET_RADIUS + ATMOSPHERE_HEIGHT); }
static inline float phase_rayleigh(float costh) { return 3 * (1 + costh * costh) / (16 * 3.14159265359f); }
static inline float phase_mie(float costh, float g = 0.85f) {
    g = fminf(g, 0.9381f);
    float k = 1.55f * g - 0.55f * g * g * g;
    float kcosth = k * costh;
    return (1 - k * k) / ((4 * 3.14159265359f) * (1 - kcosth) * (1 - kcosth));
}
static inline float atmosphere_height(f3 p) { return length(p - planet_center()) - PLANET_RADIUS; }
static inline f3 atmosphere_density(float h) {
    return f3{expf(-fmaxf(0.0f, h / RAYLEIGH_HEIGHT)), expf(-fmaxf(0.0f, h / MIE_HEIGHT)),
              fmaxf(0.0f, 1 - fabsf(h - 25000.0f) / 15000.0f)};
}
static inline f3 integrate_optical_depth(f3 ray_start, f3 ray_dir) {
    f2 isect = atmosphere_intersection(ray_start, ray_dir);
    float ray_length = isect.y;
    int sample_count = 8;
    float step_size = ray_length / sample_count;
    f3 od = mk3(0.0f);
    for (int i = 0; i < sample_count; i++) {
        f3 p = ray_start + ray_dir * (i + 0.5f) * step_size;           // atmosphere_felix.hlsl:133, left to right: (rayDir * (i + 0.5)) * stepSize
        od += atmosphere_density(atmosphere_height(p)) * step_size;
    }
    return od;
}
static inline f3 absorb(f3 od) {
    f3 e = -(od.x * C_RAYLEIGH() + od.y * C_MIE() * 1.1f + od.z * C_OZONE()) * 1.0f;
    return f3{expf(e.x), expf(e.y), expf(e.z)};
}
static inline f3 integrate_scattering(f3 ray_start, f3 ray_dir, float ray_length, f3 light_dir, f3 light_color) {
    const float exponent = 5;
    f2 isect = atmosphere_intersection(ray_start, ray_dir);
    ray_length = fminf(ray_length, isect.y);
    if (isect.x > 0) {
        ray_start = ray_start + ray_dir * isect.x;
        ray_length -= isect.x;
    }
    float costh = dot(ray_dir, light_dir);
    float phase_r = phase_rayleigh(costh);
    float phase_m = phase_mie(costh);
    const int sample_count = 16;
    f3 od = mk3(0.0f), rayleigh = mk3(0.0f), mie = mk3(0.0f);
    float prev_t = 0;
    for (int i = 1; i <= sample_count; i++) {
        float t = powf(float(i) / sample_count, exponent) * ray_length;
        float step_size = (t - prev_t);
        f3 p = ray_start + ray_dir * lerp(prev_t, t, 0.5f);
        f3 dens = atmosphere_density(atmosphere_height(p));
        od += dens * step_size;
        f3 view_t = absorb(od);
        f3 light_t = absorb(integrate_optical_depth(p, light_dir));
        rayleigh += view_t * light_t * phase_r * dens.x * step_size;    // :228-229, left to right: the vector product first, then the three scalars one by one
        mie += view_t * light_t * phase_m * dens.y * step_size;
        prev_t = t;
    }
    return (rayleigh * C_RAYLEIGH() + mie * C_MIE()) * light_color * 20.0f;
}
} // namespace atm

// atmosphere.hlsl:7-24
static inline f3 atmosphere_default(const FrameConstants& fc, f3 wi, f3 light_dir) {
    f3 sky_ambient{fc.sky_ambient[0], fc.sky_ambient[1], fc.sky_ambient[2]};
    f3 sun_mult{fc.sun_color_multiplier[0], fc.sun_color_multiplier[1], fc.sun_color_multiplier[2]};
    return (sky_ambient + sun_mult * atm::integrate_scattering(mk3(0.0f), wi, INFINITY, light_dir, mk3(1.0f))) * fc.pre_exposure;
}
// sun.hlsl:21-41
static inline f3 sun_direction(const FrameConstants& fc) { return f3{fc.sun_direction[0], fc.sun_direction[1], fc.sun_direction[2]}; }
static inline f3 sun_color_in_direction(const FrameConstants& fc, f3 dir) {
    f3 sun_mult{fc.sun_color_multiplier[0], fc.sun_color_multiplier[1], fc.sun_color_multiplier[2]};
    return 20.0f * sun_mult * fc.pre_exposure * atm::absorb(atm::integrate_optical_depth(mk3(0.0f), dir));
}
static inline f3 sample_sun_direction(const FrameConstants& fc, f2 urand, bool soft) {
    if (soft) {
        if (fc.sun_angular_radius_cos < 1.0f) {
            const m33 basis = build_orthonormal_basis(normalize(sun_direction(fc)));
            return mul(basis, uniform_sample_cone(urand, fc.sun_angular_radius_cos));
        }
    }
    return sun_direction(fc);
}

// ------------------------------------------------------------------ cube maps
// inc/cube_map.hlsl:1-9: row-major float3x3 constructors; mul(M, v) = rows . v
static const float CUBE_MAP_FACE_ROTATIONS[6][9] = {
    {0, 0, -1, 0, -1, 0, -1, 0, 0}, {0, 0, 1, 0, -1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, -1, 0, 1, 0},
    {1, 0, 0, 0, 0, 1, 0, -1, 0},   {1, 0, 0, 0, -1, 0, 0, 0, -1}, {-1, 0, 0, 0, -1, 0, 0, 0, 1}};
static inline f3 cube_face_dir(int face, f2 uv) {
    const float* m = CUBE_MAP_FACE_ROTATIONS[face];
    f3 v{uv.x * 2 - 1, uv.y * 2 - 1, -1.0f};
    return normalize(f3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z});
}
// Cube sampling (Vulkan spec 16.5.4 "Cube Map Face Selection" + bilinear within the
// face, clamp at face edges). Fixed-function in the reference; restated here and
// in the HIP kernels identically (DESIGN.md "sampler semantics").
static inline f4 sample_cube_rgba16f(const h4* cube, int width, f3 d) {
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
    f2 uv{0.5f * (sc / ma + 1.0f), 0.5f * (tc / ma + 1.0f)};
    return sample_bilinear_clamp_rgba16f(cube + size_t(face) * width * width, width, width, uv);
}
// sky/comp_cube.hlsl
static inline void render_sky_cube(const FrameConstants& fc, h4* out, int width = 64) {
    for (int face = 0; face < 6; ++face)
        for (int y = 0; y < width; ++y)
            for (int x = 0; x < width; ++x) {
                f2 uv{(x + 0.5f) / float(width), (y + 0.5f) / float(width)};
                f3 dir = cube_face_dir(face, uv);
                f3 c = atmosphere_default(fc, dir, sun_direction(fc));
                out[(size_t(face) * width + y) * width + x] = pack_rgba16f(mk4(c, 1.0f));
            }
}
// convolve_cube.hlsl
static inline void convolve_sky_cube(const h4* in, int in_width, h4* out, int width = 16) {
    for (int face = 0; face < 6; ++face)
        for (int y = 0; y < width; ++y)
            for (int x = 0; x < width; ++x) {
                f2 uv{(x + 0.5f) / float(width), (y + 0.5f) / float(width)};
                f3 output_dir = cube_face_dir(face, uv);
                const m33 basis = build_orthonormal_basis(output_dir);
                const uint32_t sample_count = 512;
                f4 result = mk4(0.0f);
                for (uint32_t i = 0; i < sample_count; ++i) {
                    f2 urand = hammersley(i, sample_count);
                    f3 input_dir = mul(basis, uniform_sample_cone(urand, 0.99f));
                    result += sample_cube_rgba16f(in, in_width, input_dir);
                }
                out[(size_t(face) * width + y) * width + x] = pack_rgba16f(result / float(sample_count));
            }
}

// lights/triangle.hlsl:52-88
struct LightSampleArea { f3 pos, normal; float pdf; };
static inline LightSampleArea sample_triangle_light(f3 v, f3 e0, f3 e1, f2 urand) {
    f3 perp = cross(e0, e1);
    float perp_inv_len = 1.0f / sqrtf(dot(perp, perp));
    float su0 = sqrtf(urand.x);
    float b0 = 1.0f - su0;
    float b1 = urand.y * su0;
    LightSampleArea r;
    r.pos = v + b0 * e0 + b1 * e1;
    r.normal = perp * perp_inv_len;
    r.pdf = 2.0f * perp_inv_len;
    return r;
}

} // namespace okj
