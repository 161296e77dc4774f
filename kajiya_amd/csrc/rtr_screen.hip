// RtrRenderer's full-resolution screen-space passes for gfx950: resolve (resolve.hlsl), temporal filter (temporal_filter.hlsl) and
// spatial cleanup (spatial_cleanup.hlsl). A translation unit of its own so that it can be compiled like the other screen-space units
// (csrc/Makefile: SCREEN_TUS) -- nothing in here has to round like the oracle's ray/triangle test.
#include "kj_rtr.hpp"
#include "kj_screen.hpp"

// What changed against the shader text, and what did not (the same rules as rtdgi_resample.hip / taa.hip):
//  * quantities that end in a DISCRETE decision keep the reference's operations -- the tap's screen position (which half-res pixel it
//    lands in: world -> sample space with its IEEE divisions), the rejection tests;
//  * weights take single-instruction reciprocals / square roots / exp2 / log2 (<= 1 ulp each) and integer powers by multiplication:
//    libm's powf is 163 VALU instructions on gfx950, sinf + cosf 235, an IEEE division 12;
//  * the resolve's spiral: the angle of tap i depends on (i, the pixel's place in its 2x2 quad, the frame) only -- 32 (cos, sin) pairs per
//    frame, evaluated ONCE per workgroup with the reference's own cosf / sinf (bit-identical taps) instead of 8 times per pixel.
namespace {
KJ_D float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
KJ_D float ggx_ndf_fast(float a2, float cos_theta) { const float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f; return a2 * rcp_fast(KJ_PI * d * d); }
KJ_D float g_smith_ggx1_fast(float ndotv, float a2) {
    const float n2 = ndotv * ndotv;
    const float tan2_v = (1.0f - n2) * rcp_fast(n2);
    return 2.0f * rcp_fast(1.0f + sqrt_fast(1.0f + a2 * tan2_v));
}
KJ_D float g_smith_ggx_correlated_fast(float ndotv, float ndotl, float a2) {
    const float lambda_v = ndotl * sqrt_fast((-ndotv * a2 + ndotv) * ndotv + a2);
    const float lambda_l = ndotv * sqrt_fast((-ndotl * a2 + ndotl) * ndotl + a2);
    return 2.0f * ndotl * ndotv * rcp_fast(lambda_v + lambda_l);
}
// specular_evaluate (inc/brdf.hlsl:140-169) for a resampling weight: `g1_wo` = g_smith_ggx1(wo.z, a2), constant per pixel
struct SpecWeight { V3 value_over_pdf; float pdf; };
KJ_D SpecWeight specular_evaluate_weight(float a2, V3 albedo, V3 wo, V3 wi, float g1_wo) {
    if (wi.z <= 0.0f || wo.z <= 0.0f) return SpecWeight{v3(0.0f), 0.0f};
    const V3 m = normalize_fast(wo + wi);
    const float pdf_h = g1_wo * ggx_ndf_fast(a2, m.z) * fmaxf(0.0f, dot(wo, m)) * rcp_fast(wo.z);
    const float wi_dot_m = dot(wi, m);
    const float jacobian = rcp_fast(4.0f * wi_dot_m);
    const V3 fresnel = lerp(albedo, v3(1.0f), pow5(fmaxf(0.0f, 1.0f - wi_dot_m)));
    const float g = g_smith_ggx_correlated_fast(wo.z, wi.z, a2);
    return SpecWeight{fresnel * (g * rcp_fast(g1_wo)), pdf_h * jacobian * rcp_fast(wi.z)};
}
}  // namespace

// ------------------------------------------------------------------ resolve.hlsl:66-663 (USE_RESTIR, CUT_CORNERS_IN_MATH, BORROW_SAMPLES)
__global__ void __launch_bounds__(64) k_rtr_resolve(RtrResolveArgs a) {
    const int tile_row0 = a.tile_row0;
    TILE_XY(a.output_tex.w, a.output_tex.h)
    const FrameConstants& fc = *a.fc;
    // (cos, sin) of the eight taps' angles for the four quad positions: lane l < 32 evaluates tap (l >> 2) + 1 at quad position l & 3
    __shared__ float spiral_cos[32], spiral_sin[32];
    if (lane < 32) {
        // ang reaches several hundred radians (ulp 3e-5): a fused multiply-add here moves the tap by ~1e-4 px and flips which
        // half-res pixel it lands in for ~0.5 % of the pixels, so the two roundings of the shader's expression are kept
        const float ang_offset_t = float(fc.frame_index * 59u % 128u) * KJ_PLASTIC;
        const float ang = __fadd_rn(__fmul_rn(float((lane >> 2) + 1) + ang_offset_t, KJ_GOLDEN_ANGLE), (float(lane & 3) / 4.0f) * KJ_TAU);
        spiral_cos[lane] = cosf(ang); spiral_sin[lane] = sinf(ang);
    }
    __syncthreads();
    if (!in_image) return;
    const int hpx = x / 2, hpy = y / 2;
    const V4 ots = tex_size4(a.output_tex.w, a.output_tex.h);
    const V2 uv = get_uv(float(x), float(y), ots);
    const float depth = a.depth_tex.ld(x, y);
    if (0.0f == depth) { a.output_tex.st(x, y, 0u); return; }
    GbufferData gbuffer = gbuffer_unpack(a.gbuffer_tex.ld(x, y));
    const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
    const V3 refl_ray_origin_ws = vr.biased_secondary_ray_origin_ws_with_normal(gbuffer.normal);
    const V3 refl_ray_origin_vs = position_world_to_view(fc, refl_ray_origin_ws);
    gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
    const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
    V3 wo = to_local(tangent_to_world, -vr.dir_ws);
    if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
    const LayeredBrdf lb = layered_brdf_from_gbuffer_ndotv(a.brdf_fg_lut, gbuffer, wo.z);
    const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
    const float a2 = fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness) * fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness);
    const float surf_to_hit_dist = length(xyz(ld4(a.hit1_tex, hpx, hpy)));
    const float eye_to_surf_dist = length(refl_ray_origin_vs);
    V3 ray_dir_vs;
    {
        const V2 cs = uv_to_cs(uv);
        ray_dir_vs = normalize(xyz(mul44(fc.view_constants.sample_to_view, V4{cs.x, cs.y, 0.0f, 1.0f})));
    }
    const float eye_ray_z_scale = -ray_dir_vs.z;
    const V4 reprojection_params = ld_reproj(a.reprojection_tex, x, y);
    const float ray_squish_scale = 16.0f / fmaxf(1e-5f, eye_to_surf_dist);
    const float ray_len_avg = exponential_unsquish(lerp(
        exponential_squish(sample_bilinear_clamp_rg16f(a.ray_len_history_tex.p, a.ray_len_history_tex.w, a.ray_len_history_tex.h, V2{uv.x + reprojection_params.x, uv.y + reprojection_params.y}).y, ray_squish_scale),
        exponential_squish(surf_to_hit_dist, ray_squish_scale), 0.1f), ray_squish_scale);
    V4 contrib_accum = v4(0.0f);
    float ray_len_accum = 0.0f;
    const V3 normal_vs = direction_world_to_view(fc, gbuffer.normal);
    const float tan_theta = sqrtf(gbuffer.roughness) * 0.25f;
    const float clip_to_view_11 = fc.view_constants.clip_to_view[5];
    float kernel_size_ws;
    {
        const float clamped_ray_len_avg = fmaxf(ray_len_avg, eye_to_surf_dist / eye_ray_z_scale * clip_to_view_11 * 0.2f * smoothstep(0.0f, 0.05f * eye_to_surf_dist, ray_len_avg));
        const float kernel_size_vs = clamped_ray_len_avg / (clamped_ray_len_avg + eye_to_surf_dist);
        kernel_size_ws = kernel_size_vs * eye_to_surf_dist * eye_ray_z_scale;
        kernel_size_ws *= tan_theta;
    }
    {
        const float scale_factor = eye_to_surf_dist * eye_ray_z_scale * clip_to_view_11;
        kernel_size_ws = fminf(kernel_size_ws, 0.1f * scale_factor);
        kernel_size_ws = fmaxf(kernel_size_ws, ots.w * 4.0f * scale_factor);
    }
    V3 kernel_t1, kernel_t2;
    {   // get_specular_filter_kernel_basis (resolve.hlsl:72-79) with specular_dominant_direction (brdf.hlsl:313-317)
        const V3 v = -vr.dir_ws, n = gbuffer.normal;
        const V3 r = reflect(-v, n);
        const float f = (1.0f - gbuffer.roughness) * (sqrtf(1.0f - gbuffer.roughness) + gbuffer.roughness);
        const V3 dominant = normalize(lerp(n, r, f));
        const V3 reflected = reflect(-dominant, n);
        kernel_t1 = normalize(cross(n, reflected)) * kernel_size_ws;
        kernel_t2 = cross(reflected, kernel_t1);
    }
    const V4 blue = blue_noise_for_pixel(a.blue_noise, uint32_t(hpx + 16), uint32_t(hpy + 16), fc.frame_index);
    const float KERNEL_SHARPNESS = 0.666f;
    const float RADIUS_SAMPLE_MULT = 1.0f / powf(8.0f, KERNEL_SHARPNESS);
    const float RADIUS_INC_ON_FAIL = 0.25f;
    const float g1_wo = g_smith_ggx1_fast(wo.z, lb.roughness * lb.roughness);
    const float surf_squished = exponential_squish(surf_to_hit_dist, ray_squish_scale);
    const float sqrt_roughness = sqrtf(gbuffer.roughness);
    const float pdf_lerp_t = smoothstep(0.4f, 0.7f, sqrt_roughness) * smoothstep(0.0f, 0.1f, ray_len_avg / eye_to_surf_dist);
    const float origin_bias = lerp(1.0f, RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS, 0.4f * fminf(1.0f, 3.0f * sqrt_roughness));
    const float inv_kernel_size = rcp_fast(fmaxf(1e-10f, kernel_size_ws));
    const V3 eye = get_eye_position(fc);
    float sample_radius_accum = 1.0f;
    for (uint32_t sample_i = 1; sample_i <= 8u; ++sample_i, sample_radius_accum += RADIUS_INC_ON_FAIL) {
        const bool is_center_sample = sample_i == 8u;
        int sample_px_x, sample_px_y;
        {
            const uint32_t spiral_idx = (sample_i - 1u) * 4u + px_idx_in_quad;
            float sample_i_with_jitter = sample_radius_accum;
            if (is_center_sample) sample_i_with_jitter = contrib_accum.w > 1e-8f ? blue.y : 0.0f;
            else sample_i_with_jitter += blue.y;
            const float radius = exp2_fast(KERNEL_SHARPNESS * log2_fast(sample_i_with_jitter)) * RADIUS_SAMPLE_MULT;      // pow(0, .) = exp2(-inf) = 0
            const V3 offset_ws = (spiral_cos[spiral_idx] * kernel_t1 + spiral_sin[spiral_idx] * kernel_t2) * radius;
            const V3 sample_ws = refl_ray_origin_ws + offset_ws;
            const V3 sample_cs = position_world_to_sample(fc, sample_ws);
            const V2 sample_uv = cs_to_uv(V2{sample_cs.x, sample_cs.y});
            sample_px_x = int(floorf(sample_uv.x * ots.x / 2.0f));
            sample_px_y = int(floorf(sample_uv.y * ots.y / 2.0f));
        }
        float rejection_bias = 1.0f;
        // (round 6: a tap's gathers leave in two groups -- [normal, reservoir] at the tap's pixel, then [ray origin, ray, irradiance] at the pixel the reservoir points
        // at -- instead of one by one with the rejection test between them; the tap's position depends on the taps before it, so taps cannot share a round trip)
        bool in_t, in_s;      // (the half-res images share one extent)
        const uint32_t normal_raw = a.half_view_normal_tex.ld_raw(sample_px_x, sample_px_y, in_t);
        const uint2 reservoir_raw_ = a.restir_reservoir_tex.ld_raw(sample_px_x, sample_px_y, in_t);
        const V3 sample_normal_vs = xyz(unpack_rgba8_snorm(in_t ? normal_raw : 0u));
        float pdf0_mult = 1.0f, pdf1_mult = 1.0f;
        const uint2 reservoir_raw = in_t ? reservoir_raw_ : make_uint2(0u, 0u);
        const Reservoir1spp r = Reservoir1spp::from_raw(reservoir_raw);
        const int spx = int(r.payload & 0xffffu), spy = int(r.payload >> 16);
        const auto origin_raw = a.restir_ray_orig_tex.ld_raw(spx, spy, in_s);
        const uint2 ray_raw = a.restir_ray_tex.ld_raw(spx, spy, in_s);
        const uint2 irr_raw = a.restir_irradiance_tex.ld_raw(spx, spy, in_s);
        const RtrRestirRayOrigin sample_origin = ray_origin_from_raw(in_s ? origin_raw : decltype(origin_raw)());
        const V3 sample_origin_ws = sample_origin.ray_origin_eye_offset_ws + eye;
        if (reservoir_raw.x == 0u || sample_origin.roughness > gbuffer.roughness * 2.0f) continue;
        const V4 restir_ray = unpack_rgba16f(in_s ? ray_raw : make_uint2(0u, 0u));
        const V3 sample_hit_ws = xyz(restir_ray) + sample_origin_ws;
        const V3 sample_origin_vs = position_world_to_view(fc, sample_origin_ws);
        const V4 restir_irr = unpack_rgba16f(in_s ? irr_raw : make_uint2(0u, 0u));
        const V3 sample_radiance = xyz(restir_irr);
        const float sample_ray_pdf = restir_ray.w;
        const float inv_neighbor_sampling_pdf = r.W;      // 1 / neighbor_sampling_pdf
        const V3 sample_hit_vs_abs = position_world_to_view(fc, sample_hit_ws);
        const V3 center_to_hit_vs = sample_hit_vs_abs - lerp(refl_ray_origin_vs, sample_origin_vs, RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS);
        const float sample_cos_theta = 1.0f - restir_irr.w;
        const float center_to_hit_dist2 = dot(center_to_hit_vs, center_to_hit_vs);
        const V3 s2h = sample_hit_ws - sample_origin_ws;
        const float inv_sample_to_hit_dist2 = rcp_fast(dot(s2h, s2h));
        {
            const V3 dv = sample_hit_vs_abs - lerp(refl_ray_origin_vs, sample_origin_vs, origin_bias);
            pdf0_mult *= fmaxf(1e-5f, dot(dv, dv) * inv_sample_to_hit_dist2);                     // (d / sample_to_hit_dist)^2
            pdf1_mult *= fmaxf(1.0f, center_to_hit_dist2 * inv_sample_to_hit_dist2);
        }
        const V3 wi = normalize_fast(to_local(tangent_to_world, direction_view_to_world(fc, center_to_hit_vs)));
        if (wi.z < 1e-5f) continue;
        rejection_bias *= dot(normal_vs, sample_normal_vs) > 0.7f ? 1.0f : 0.0f;
        {
            const float depth_diff = fabsf(refl_ray_origin_vs.z - sample_origin_vs.z) * inv_kernel_size;
            rejection_bias *= exp2_fast(-fmaxf(0.3f, normal_vs.z) * depth_diff * depth_diff);
        }
        const V3 surface_offset = sample_origin_vs - refl_ray_origin_vs;
        // own half-res sample: the offset is a rounding residue (0/0 or a random direction in the shader) -> no rejection, as in the oracle
        const float surface_offset_len = length(surface_offset);
        if (surface_offset_len > 1e-5f * eye_to_surf_dist &&
            dot(center_to_hit_vs, normal_vs) * 0.2f / sqrtf(center_to_hit_dist2) < dot(surface_offset, normal_vs) / surface_offset_len) rejection_bias *= is_center_sample ? 1.0f : 0.0f;
        const SpecWeight spec = specular_evaluate_weight(lb.roughness * lb.roughness, lb.spec_albedo, wo, wi, g1_wo);
        const float spec_weight = spec.pdf * stepf(0.0f, wi.z);
        float contrib_wt = 0.0f;
        {
            const float cos_theta = normalize_fast(wo + wi).z;
            const float bent_cos_theta = fminf(sample_cos_theta, cos_theta * 1.25f);
            // ggx_ndf(a2, bent) / ggx_ndf(a2, cos) = (d_cos / d_bent)^2 with d = c^2 (a2 - 1) + 1: the a2 / pi factors cancel
            const float d_bent = bent_cos_theta * bent_cos_theta * (a2 - 1.0f) + 1.0f, d_cos = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
            const float ndf_ratio = d_cos * rcp_fast(d_bent);
            const float bent_sample_pdf0 = spec.pdf * (ndf_ratio * ndf_ratio);
            const float pdf_x[2] = {fminf(bent_sample_pdf0, RTR_RESTIR_MAX_PDF_CLAMP), fminf(spec.pdf, RTR_RESTIR_MAX_PDF_CLAMP)};
            const float inv_pdf_y[2] = {inv_neighbor_sampling_pdf * rcp_fast(pdf0_mult), inv_neighbor_sampling_pdf * rcp_fast(pdf1_mult)};
            const float pdf_z[2] = {1.0f - pdf_lerp_t, pdf_lerp_t};
            const float mis_weight = fmaxf(1e-4f, spec.pdf * rcp_fast(sample_ray_pdf + spec.pdf));
#pragma unroll
            for (int pdf_i = 0; pdf_i < 2; ++pdf_i) {
                const float bent_sample_pdf = pdf_x[pdf_i], pdf_influence = pdf_z[pdf_i];
                contrib_wt = rejection_bias * mis_weight * fmaxf(1e-10f, spec_weight * rcp_fast(bent_sample_pdf));
                contrib_accum = contrib_accum + v4(sample_radiance * (bent_sample_pdf * inv_pdf_y[pdf_i]) * spec.value_over_pdf, 1.0f) * contrib_wt * pdf_influence;
            }
        }
        ray_len_accum += surf_squished * contrib_wt;
        sample_radius_accum += 1.0f - RADIUS_INC_ON_FAIL;
    }
    const float contrib_norm_factor = fmaxf(1e-14f, contrib_accum.w);
    V3 rgb = xyz(contrib_accum) / contrib_norm_factor;
    ray_len_accum /= contrib_norm_factor;
    rgb = rgb / lb.preintegrated_reflection;
    rgb = rgb * lb.preintegrated_reflection_mult;
    ray_len_accum = exponential_unsquish(ray_len_accum, ray_squish_scale);
    a.output_tex.st(x, y, pack_r11g11b10f(rgb));
    st2h(a.ray_len_output_tex, x, y, V2{ray_len_accum, ray_len_avg});
}

// image_sample_catmull_rom_5tap (inc/image.hlsl:88-172) with the identity remap. The reference issues five BILINEAR fetches; their
// positions are texel centres in one axis (the outer taps) or between the two middle texels, so the five footprints are 12 texels, not
// 20, and the bilinear weights are known without the round trip through normalised coordinates: (1, 0) in the centred axis, the
// Catmull-Rom offset w2 / (w1 + w2) in the other. Same filter; the weights differ from the fetch unit's by the round trip's rounding
// (~1e-4 of a texel at 1440p), far inside the image's fp16 storage.
KJ_D V4 catmull_rom_5tap(const ImgH4& tex, V2 uv, V2 tex_size) {
    const V2 sample_pos = uv * tex_size;
    const V2 tex_pos1{floorf(sample_pos.x - 0.5f) + 0.5f, floorf(sample_pos.y - 0.5f) + 0.5f};
    const V2 f = sample_pos - tex_pos1;
    const V2 w0 = f * (-0.5f + f * (1.0f - 0.5f * f));
    const V2 w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
    const V2 w2 = f * (0.5f + f * (2.0f - 1.5f * f));
    const V2 w3 = f * f * (-0.5f + 0.5f * f);
    const V2 w12 = w1 + w2;
    const V2 o{w2.x * rcp_fast(w12.x), w2.y * rcp_fast(w12.y)};
    const int ix = int(floorf(sample_pos.x - 0.5f)), iy = int(floorf(sample_pos.y - 0.5f));
    auto T = [&](int dx, int dy) { return unpack_rgba16f(tex.ldc(ix + dx, iy + dy)); };      // clamp to edge, as the sampler
    auto mix = [](V4 a, V4 b, float t) { return a * (1.0f - t) + b * t; };
    V4 result = mix(T(0, -1), T(1, -1), o.x) * (w12.x * w0.y);
    result += mix(T(-1, 0), T(-1, 1), o.y) * (w0.x * w12.y);
    result += mix(mix(T(0, 0), T(1, 0), o.x), mix(T(0, 1), T(1, 1), o.x), o.y) * (w12.x * w12.y);
    result += mix(T(2, 0), T(2, 1), o.y) * (w3.x * w12.y);
    result += mix(T(0, 2), T(1, 2), o.x) * (w12.x * w3.y);
    return result * rcp_fast(w12.x * w0.y + w0.x * w12.y + w12.x * w12.y + w3.x * w12.y + w12.x * w3.y);
}
KJ_D V4 crunch_fast(V4 v) {      // linear_rgb_to_crunched_luma_chroma with single-instruction sqrt / rcp (not for the 3x3 moments below)
    const V3 y = sRGB_to_YCbCr(xyz(v));
    return v4(y * (sqrt_fast(y.x) * rcp_fast(fmaxf(1e-8f, y.x))), v.w);
}
KJ_D V3 soft_color_clamp_fast(V3 center, V3 history, V3 ex, V3 dev) {
    const V3 m = vmax(vabs(history * 0.1f), dev);
    const V3 history_dist = vabs(history - ex) * V3{rcp_fast(m.x), rcp_fast(m.y), rcp_fast(m.z)};
    const V3 closest_pt = vclamp(history, center - dev, center + dev);
    return V3{lerp(history.x, closest_pt.x, smoothstep(1.0f, 3.0f, history_dist.x)), lerp(history.y, closest_pt.y, smoothstep(1.0f, 3.0f, history_dist.y)),
              lerp(history.z, closest_pt.z, smoothstep(1.0f, 3.0f, history_dist.z))};
}

// ------------------------------------------------------------------ temporal_filter.hlsl:37-259
__global__ void __launch_bounds__(64) k_rtr_temporal_filter(const FrameConstants* __restrict__ fcp, ImgU32 input_tex, ImgH4 history_tex, ImgF32 depth_tex, ImgU32 ray_len_tex,
                                                             ImgU2 reprojection_tex, ImgR8 refl_restir_invalidity_tex, ImgU4 gbuffer_tex, ImgH4 output_tex, int tile_row0) {
    TILE_XY_ROWS(output_tex.w, output_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const V4 ots = tex_size4(output_tex.w, output_tex.h);
    // (round 6: everything whose address is known up front -- the pixel's own six texels and the 3x3 neighbourhood's {input, depth} -- is requested before the first
    // value is used; the moments' loop as the text has it is 18 round trips to memory one after the other)
    bool nb_in[9]; uint32_t nb_c[9]; float nb_d[9];      // (input and depth share the extent)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        nb_c[i] = input_tex.ld_raw(x + i % 3 - 1, y + i / 3 - 1, nb_in[i]);
        nb_d[i] = depth_tex.ld_raw(x + i % 3 - 1, y + i / 3 - 1, nb_in[i]);
    }
    bool c_in, c_in_h;
    const uint32_t ray_len_raw = ray_len_tex.ld_raw(x, y, c_in);
    const uint2 reproj_raw = reprojection_tex.ld_raw(x, y, c_in);
    const uint4 gbuffer_raw = gbuffer_tex.ld_raw(x, y, c_in);
    const uint8_t restir_inv_raw = refl_restir_invalidity_tex.ld_raw(x / 2, y / 2, c_in_h);
    auto ld_in = [&](int i) { return nb_in[i] ? v4(unpack_r11g11b10f(nb_c[i]), 1.0f) : v4(0.0f); };
    const V4 center = crunch_fast(ld_in(4));
    const float refl_ray_length = clampf(unpack_2x16f_uint(c_in ? ray_len_raw : 0u).x, 0.0f, 1e3f);
    const V2 uv = get_uv(float(x), float(y), ots);
    const float center_depth = nb_in[4] ? nb_d[4] : 0.0f;
    const ViewRay vr = view_ray_from_uv_and_depth(fc, uv, center_depth);
    V3 ray_dir_vs;
    {
        const V2 cs = uv_to_cs(uv);
        ray_dir_vs = normalize(xyz(mul44(fc.view_constants.sample_to_view, V4{cs.x, cs.y, 0.0f, 1.0f})));
    }
    const V3 reflection_hit_vs = vr.hit_vs + ray_dir_vs * refl_ray_length;
    const V4 reflection_hit_cs = mul44(fc.view_constants.view_to_sample, v4(reflection_hit_vs, 1.0f));
    const V4 prev_hit_cs = mul44(fc.view_constants.clip_to_prev_clip, reflection_hit_cs);
    const float inv_phw = rcp_fast(prev_hit_cs.w);
    V2 hit_prev_uv = cs_to_uv(V2{prev_hit_cs.x * inv_phw, prev_hit_cs.y * inv_phw});
    const V4 prev_reflector_cs = mul44(fc.view_constants.clip_to_prev_clip, v4(vr.hit_cs, 1.0f));
    const float inv_prw = rcp_fast(prev_reflector_cs.w);
    const V2 reflector_prev_uv = cs_to_uv(V2{prev_reflector_cs.x * inv_prw, prev_reflector_cs.y * inv_prw});
    const uint2 rpr = c_in ? reproj_raw : make_uint2(0u, 0u);
    const V4 reproj{from_snorm16(int16_t(rpr.x & 0xffff)), from_snorm16(int16_t(rpr.x >> 16)), from_snorm16(int16_t(rpr.y & 0xffff)), from_snorm16(int16_t(rpr.y >> 16))};
    const V2 rmv = reflector_prev_uv - uv;
    const float reflector_move_rate = fminf(1.0f, sqrt_fast(reproj.x * reproj.x + reproj.y * reproj.y) * rsq_fast(dot(rmv, rmv)));
    hit_prev_uv = lerp(uv, hit_prev_uv, reflector_move_rate);
    const uint32_t quad_reproj_valid_packed = uint32_t(reproj.z * 15.0f + 0.5f);
    const V4 history_mult{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1.0f};
    V4 history0 = v4(0.0f);
    float history0_valid = 1.0f;
    const V2 reproj_uv{uv.x + reproj.x, uv.y + reproj.y};
    if (0u == quad_reproj_valid_packed) {
        history0_valid = 0.0f;
    } else if (15u == quad_reproj_valid_packed) {
        history0 = vmax(v4(0.0f), catmull_rom_5tap(history_tex, reproj_uv, V2{ots.x, ots.y})) * history_mult;
    } else {
        const V4 qv{(quad_reproj_valid_packed & 1u) ? 1.0f : 0.0f, (quad_reproj_valid_packed & 2u) ? 1.0f : 0.0f, (quad_reproj_valid_packed & 4u) ? 1.0f : 0.0f,
                    (quad_reproj_valid_packed & 8u) ? 1.0f : 0.0f};
        // get_bilinear_filter (inc/bilinear.hlsl)
        const V2 pxf{reproj_uv.x * ots.x - 0.5f, reproj_uv.y * ots.y - 0.5f};
        const V2 origin{floorf(pxf.x), floorf(pxf.y)};
        const V2 wts{pxf.x - origin.x, pxf.y - origin.y};
        const int ox = int(origin.x), oy = int(origin.y);
        const V4 s00 = ld4(history_tex, ox, oy) * history_mult, s10 = ld4(history_tex, ox + 1, oy) * history_mult;
        const V4 s01 = ld4(history_tex, ox, oy + 1) * history_mult, s11 = ld4(history_tex, ox + 1, oy + 1) * history_mult;
        V4 w{(1.0f - wts.x) * (1.0f - wts.y), wts.x * (1.0f - wts.y), (1.0f - wts.x) * wts.y, wts.x * wts.y};
        w = w * qv;
        const float wsum = dot(w, v4(1.0f));
        if (wsum > 1e-5f) history0 = (s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w) * (1.0f / wsum);
        else history0 = (s00 + s10 + s01 + s11) / 4.0f;
    }
    history0 = crunch_fast(history0);
    const V4 history1 = crunch_fast(sample_bilinear_clamp_rgba16f(history_tex.p, history_tex.w, history_tex.h, hit_prev_uv) * history_mult);
    const float history1_valid = quad_reproj_valid_packed == 15u ? 1.0f : 0.0f;
    V4 vsum = v4(0.0f), vsum2 = v4(0.0f);
    float wsum = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int i = (dy + 1) * 3 + (dx + 1);
            const float sample_depth = nb_in[i] ? nb_d[i] : 0.0f;
            const V4 neigh = linear_rgb_to_crunched_luma_chroma(ld_in(i));
            const float w = exp2_fast(-200.0f * fabsf(center_depth * rcp_fast(sample_depth) - 1.0f));
            vsum = vsum + neigh * w;
            vsum2 = vsum2 + neigh * neigh * w;
            wsum += w;
        }
    const V4 ex = vsum / wsum, ex2 = vsum2 / wsum;      // the moments stay IEEE: their difference below is a variance (cancellation amplifies an ulp)
    const V4 dvar = vmax(v4(0.0f), ex2 - ex * ex);
    const V4 dev{sqrt_fast(dvar.x), sqrt_fast(dvar.y), sqrt_fast(dvar.z), sqrt_fast(dvar.w)};
    const GbufferData gbuffer = gbuffer_unpack(c_in ? gbuffer_raw : make_uint4(0u, 0u, 0u, 0u));
    const float restir_invalidity = from_unorm8(c_in_h ? restir_inv_raw : uint8_t(0));
    const float n_deviations = lerp(reproj.z > 0.0f ? 2.0f : 1.25f, 0.625f, restir_invalidity);
    float wo_similarity;
    {
        const V3 current_wo = normalize_fast(vr.hit_ws - get_eye_position(fc));
        const V3 prev_wo = normalize_fast(vr.hit_ws - get_prev_eye_position(fc));
        const float clamped_roughness = fmaxf(0.1f, gbuffer.roughness);
        const float a2s = clamped_roughness * clamped_roughness, cwp = dot(current_wo, prev_wo), dd = cwp * cwp * (a2s - 1.0f) + 1.0f;
        float t = saturate(a2s * a2s * rcp_fast(dd * dd));      // ggx_ndf_0_1
        t *= t; t *= t; t *= t; t *= t; t *= t;                 // ^32
        wo_similarity = t;
    }
    const V3 inv_dev{rcp_fast(dev.x), rcp_fast(dev.y), rcp_fast(dev.z)};
    const float h0diff = length_fast((xyz(history0) - xyz(ex)) * inv_dev);
    const float h1diff = length_fast((xyz(history1) - xyz(ex)) * inv_dev);
    const float sqrt_roughness = sqrt_fast(gbuffer.roughness);
    float h0_score = 1.0f * smoothstep_fast(0.0f, 0.5f, sqrt_roughness) * lerp(wo_similarity, 1.0f, sqrt_roughness);
    float h1_score = (1.0f - h0_score) * lerp(1.0f, smoothstep_fast(0.0f, 1.0f, h0diff - h1diff), smoothstep_fast(0.0f, 0.15f, sqrt_roughness));
    h0_score *= history0_valid;
    h1_score *= history1_valid;
    const float score_sum = h0_score + h1_score;
    h0_score *= rcp_fast(score_sum);
    h1_score = 1.0f - h0_score;
    if (!(h0_score < 1.001f)) { h0_score = 1.0f; h1_score = 0.0f; }
    const V4 clamped_history0 = v4(soft_color_clamp_fast(xyz(center), xyz(history0), xyz(ex), xyz(dev) * n_deviations), history0.w);
    const V4 clamped_history1 = v4(soft_color_clamp_fast(xyz(center), xyz(history1), xyz(ex), xyz(dev) * n_deviations), history1.w);
    const V4 clamped_history = clamped_history0 * h0_score + clamped_history1 * h1_score;
    const float max_sample_count = 16.0f;
    const float current_sample_count = clamped_history.w * saturate(h0_score * history0_valid + h1_score * history1_valid);
    V4 res = lerp(clamped_history, center, rcp_fast(1.0f + fminf(max_sample_count, current_sample_count * lerp(wo_similarity, 1.0f, 0.5f))));
    res.w = fminf(current_sample_count, max_sample_count) + 1.0f;
    res = crunched_luma_chroma_to_linear_rgb(res);
    st4(output_tex, x, y, vmax(v4(0.0f), res));
}

// ------------------------------------------------------------------ spatial_cleanup.hlsl:20-65
__global__ void __launch_bounds__(64) k_rtr_cleanup(const FrameConstants* __restrict__ fcp, ImgH4 input_tex, ImgF32 depth_tex, ImgU32 geometric_normal_tex, ImgU32 output_tex,
                                                     const int4* __restrict__ spatial_resolve_offsets, int tile_row0) {
    TILE_XY_ROWS(output_tex.w, output_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const V4 center = ld4(input_tex, x, y);
    const float center_depth = depth_tex.ld(x, y);
    const float center_sample_count = center.w;
    if (center_sample_count >= 8.0f || center_depth == 0.0f) { output_tex.st(x, y, pack_r11g11b10f(xyz(center))); return; }
    const V3 center_normal_vs = unpack_a2r10g10b10(geometric_normal_tex.ld(x, y)) * 2.0f - 1.0f;
    const float filter_radius_ss = 0.5f * fc.view_constants.view_to_clip[5] / -depth_to_view_z(fc, center_depth);
    const uint32_t filter_idx = uint32_t(clampf(filter_radius_ss * 7.0f, 0.0f, 7.0f));
    V3 vsum = v3(0.0f);
    float wsum = 0.0f;
    const int sc = int(8.0f - center_sample_count / 2.0f);
    const uint32_t sample_count = uint32_t(min(max(sc, 2), 8));
    const int kernel_scale = center_sample_count < 4.0f ? 2 : 1;
    const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
    for (uint32_t sample_i = 0; sample_i < sample_count; ++sample_i) {
        const int4 o = spatial_resolve_offsets[(px_idx_in_quad * 16u + sample_i) + 64u * filter_idx];
        const int sx = x + kernel_scale * o.x, sy = y + kernel_scale * o.y;
        const V3 neigh = vsqrt(xyz(ld4(input_tex, sx, sy)));
        const float sample_depth = depth_tex.ld(sx, sy);
        const V3 sample_normal_vs = geometric_normal_tex.inb(sx, sy) ? unpack_a2r10g10b10(geometric_normal_tex.ld(sx, sy)) * 2.0f - 1.0f : v3(-1.0f);
        float w = 1.0f;
        w *= exp2f(-50.0f * fabsf(center_normal_vs.z * (center_depth / sample_depth - 1.0f)));
        const float dp = saturate(dot(center_normal_vs, sample_normal_vs));
        w *= dp * dp * dp;
        vsum += neigh * w;
        wsum += w;
    }
    const V3 v = vsum / wsum;
    output_tex.st(x, y, pack_r11g11b10f(v * v));
}

hipError_t launch_rtr_resolve(const RtrResolveArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_rtr_resolve, dim3((a.output_tex.w + 7) / 8, a.tile_rows), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_rtr_temporal_filter(const RtrTemporalFilterArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_rtr_temporal_filter, dim3((a.output_tex.w + 7) / 8, a.tile_rows), dim3(64), 0, s, a.fc, a.input_tex, a.history_tex, a.depth_tex, a.ray_len_tex, a.reprojection_tex,
                       a.refl_restir_invalidity_tex, a.gbuffer_tex, a.output_tex, a.tile_row0);
    return hipGetLastError();
}
hipError_t launch_rtr_cleanup(const RtrCleanupArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_rtr_cleanup, dim3((a.output_tex.w + 7) / 8, a.tile_rows), dim3(64), 0, s, a.fc, a.input_tex, a.depth_tex, a.geometric_normal_tex, a.output_tex, a.spatial_resolve_offsets, a.tile_row0);
    return hipGetLastError();
}
