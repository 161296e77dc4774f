// ORACLE (test infrastructure). RtrRenderer restated: host orchestration from
// crates/lib/kajiya/src/renderers/rtr.rs:97-400 (`trace`) and :440-480 (`filter_temporal`), and every shader they record:
// rtr/reflection.rgen.hlsl + reflection_trace_common.inc.hlsl, reflection_validate.rgen.hlsl, rtr_restir_temporal.hlsl,
// resolve.hlsl, temporal_filter.hlsl, spatial_cleanup.hlsl, with the compile-time switches of rtr_settings.hlsl as checked
// in (ReSTIR on, path validation on, approximate measure conversion, world radiance cache off).
//
// Third-party data: the sampler tables of the `blue-noise-sampler 0.1.0` crate (rtr.rs:16,66-68: spp64 RANKING_TILE,
// SCRAMBLING_TILE, SOBOL) and rtr.rs's own SPATIAL_RESOLVE_OFFSETS are inputs handed over by the caller; the arithmetic that
// consumes them (inc/blue_noise.hlsl:31-60) is restated here. PARITY UNPINNED: the reference holds no vectors for this path.
//
// Places where the reference leaves the result undefined and this restatement (and the HIP path) picks a value:
//  * refl_restir_invalidity_tex is a transient the validate pass only partly writes (quads whose jittered pixel is sky keep
//    whatever the pooled image held): cleared to 0 at the start of each frame here.
//  * reflection_validate normalises a zero vector where no history exists yet (first frames): the ray is traced along +Z.
//  * B10G11R11_UFLOAT stores: round to nearest (ties up) through fp16, negative -> 0 (okj::pack_r11g11b10f).
//  * resolve's approximate sample shadowing divides by the length of (sample origin - pixel origin), which for the pixel's own
//    half-res sample is 0 or a 1-ulp rounding residue (NaN / random direction in the shader): residues count as zero = no rejection.
#pragma once
#include "okj_rtdgi.hpp"
#include "okj_taa.hpp"

namespace okj {

static const float RTR_ROUGHNESS_CLAMP = 6e-4f;            // rtr_settings.hlsl:47
static const float RTR_RESTIR_MAX_PDF_CLAMP = 200.0f;      // :51
static const float RTR_RESTIR_TEMPORAL_M_CLAMP = 8.0f;     // :11
static const float RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS = 0.5f;   // :21
static const float RTR_SAMPLING_BIAS = 0.15f;              // reflection_trace_common.inc.hlsl:41-47 (USE_HEAVY_BIAS)

// B10G11R11_UFLOAT_PACK32: r in bits 0..10 (5e6m), g 11..21 (5e6m), b 22..31 (5e5m)
static inline uint32_t f32_to_ufloat(float v, int mant_bits) {
    if (!(v > 0.0f)) return 0;                                   // negative, zero, NaN -> 0
    const uint32_t h = f32_to_f16(v) & 0x7fffu;                  // 5e10m
    const int drop = 10 - mant_bits;
    uint32_t r = (h + (1u << (drop - 1))) >> drop;
    const uint32_t max_finite = (30u << mant_bits) | ((1u << mant_bits) - 1u);
    if (h >= 0x7c00u) return 31u << mant_bits;                   // inf stays inf
    return r > max_finite ? max_finite : r;
}
static inline float ufloat_to_f32(uint32_t v, int mant_bits) { return f16_to_f32(uint16_t(v << (10 - mant_bits))); }
static inline uint32_t pack_r11g11b10f(f3 c) { return f32_to_ufloat(c.x, 6) | (f32_to_ufloat(c.y, 6) << 11) | (f32_to_ufloat(c.z, 5) << 22); }
static inline f3 unpack_r11g11b10f(uint32_t p) { return f3{ufloat_to_f32(p & 0x7ffu, 6), ufloat_to_f32((p >> 11) & 0x7ffu, 6), ufloat_to_f32(p >> 22, 5)}; }

// rtr_restir_pack_unpack.inc.hlsl:1-22
struct RtrRestirRayOrigin {
    f3 ray_origin_eye_offset_ws; float roughness; uint32_t frame_index_mod4;
    static RtrRestirRayOrigin from_raw(f4 raw) {
        RtrRestirRayOrigin r;
        r.ray_origin_eye_offset_ws = xyz(raw);
        const f2 misc = unpack_2x16f_uint(asuint(raw.w));
        r.roughness = misc.x;
        r.frame_index_mod4 = uint32_t(misc.y) & 3u;
        return r;
    }
    f4 to_raw() const { return mk4(ray_origin_eye_offset_ws, asfloat(pack_2x16f_uint(roughness, float(frame_index_mod4)))); }
};

typedef std::function<f3(f3 query_from_ws, f3 pt_ws, f3 normal_ws, uint32_t rank, bool stochastic_interpolation, uint32_t& rng)> IrcacheLookupStochasticFn;

struct RtrInputs {
    int W = 0, H = 0;
    ImgU32 geometric_normal; ImgU4 gbuffer; ImgR32F depth;
    ImgRGBA16S reprojection_map;
    const h4* sky_cube = nullptr; int sky_cube_width = 64;       // the UNconvolved cube (world_render_passes.rs:178)
    const Scene* scene = nullptr;
    const uint8_t* blue_noise = nullptr;
    const h4* brdf_fg_lut = nullptr;
    ImgRGBA16F rtdgi_irradiance;                                // this frame's RtdgiOutput::screen_irradiance_tex
    ImgRGBA16F refl0_tex, refl1_tex; ImgU32 refl2_tex;          // RtdgiCandidates (radiance, hit, normal RGBA8_SNORM); rtr overwrites smooth pixels
    ImgU32 half_view_normal; ImgR32F half_depth;
    const uint32_t* ranking_tile = nullptr; const uint32_t* scrambling_tile = nullptr; const uint32_t* sobol = nullptr;
    const int32_t* spatial_resolve_offsets = nullptr;          // 16 * 4 * 8 x int4
    IrcacheLookupStochasticFn ircache_lookup;
};

enum : uint32_t { RTR_PASS_TRACE = 1, RTR_PASS_VALIDATE = 2, RTR_PASS_RESTIR_TEMPORAL = 4, RTR_PASS_RESOLVE = 8, RTR_PASS_TEMPORAL_FILTER = 16,
                  RTR_PASS_CLEANUP = 32, RTR_PASS_ALL = 63, RTR_PASS_KEEP = 0x80000000u };

struct Rtr {
    std::map<std::string, std::vector<uint8_t>> surf;
    int W = 0, H = 0, hw = 0, hh = 0;
    bool flip[8] = {false, false, false, false, false, false, false, false};
    bool reuse_rtdgi_rays = true;                               // rtr.rs:32,70
    bool literal_own_sample_shadowing = false;                  // test knob (tests/test_ref_hlsl.py): resolve.hlsl:535-540 as written, residues and all (see the header)
    std::atomic<uint64_t> rays_closest{0}, rays_any{0};
    f3 sun_color;

    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    void resize(int W_, int H_) {
        if (W == W_ && H == H_) return;
        W = W_; H = H_; hw = (W + 1) / 2; hh = (H + 1) / 2;
        surf.clear();
    }
    template <typename T> void pingpong(const char* key, int idx, int w, int h, Img<T>& output, Img<T>& history, bool advance) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx] != !advance) std::swap(a, b);             // advance: use the current flip, then toggle; keep: the previous call's
        output = get<T>(a, w, h);
        history = get<T>(b, w, h);
        if (advance) flip[idx] = !flip[idx];
    }

    // inc/blue_noise.hlsl:31-60
    static float blue_noise_sampler(const RtrInputs& in, int pixel_i, int pixel_j, int sample_index, int sample_dimension) {
        pixel_i &= 127; pixel_j &= 127; sample_index &= 255; sample_dimension &= 255;
        const int ranked = sample_index ^ int(in.ranking_tile[sample_dimension + (pixel_i + pixel_j * 128) * 8]);
        int value = int(in.sobol[sample_dimension + ranked * 256]);
        value ^= int(in.scrambling_tile[(sample_dimension % 8) + (pixel_i + pixel_j * 128) * 8]);
        return (0.5f + float(value)) / 256.0f;
    }
    static float ggx_ndf_0_1(float a2, float cos_theta) {       // brdf.hlsl:151-154
        const float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
        return a2 * a2 / (d * d);
    }
    static float exponential_squish(float len, float s) { return exp2f(-clampf(s * len, 0.0f, 100.0f)); }   // inc/math.hlsl:69-76
    static float exponential_unsquish(float len, float s) { return fmaxf(0.0f, -1.0f / s * log2f(1e-30f + len)); }
    static f3 soft_color_clamp(f3 center, f3 history, f3 ex, f3 dev) {   // inc/soft_color_clamp.hlsl
        const f3 history_dist = vabs(history - ex) / vmax(vabs(history * 0.1f), dev);
        const f3 closest_pt = vclamp(history, center - dev, center + dev);
        return f3{lerp(history.x, closest_pt.x, smoothstep(1.0f, 3.0f, history_dist.x)), lerp(history.y, closest_pt.y, smoothstep(1.0f, 3.0f, history_dist.y)),
                  lerp(history.z, closest_pt.z, smoothstep(1.0f, 3.0f, history_dist.z))};
    }
    static i2 hi_px_subpixel(uint32_t k) { return i2{HI_PX_SUBPIXELS[k & 3][0], HI_PX_SUBPIXELS[k & 3][1]}; }

    // ------------------------------------------------------------------ reflection_trace_common.inc.hlsl:56-257
    struct RtrTraceResult { f3 total_radiance; float hit_t; f3 hit_normal_vs; };
    RtrTraceResult do_the_thing(const FrameConstants& fc, const RtrInputs& in, f3 normal_ws, float roughness, uint32_t& rng, Ray outgoing_ray) {
        (void)normal_ws;
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const float roughness_bias = roughness;                  // USE_AGGRESSIVE_SECONDARY_ROUGHNESS_BIAS
        const float reflected_cone_spread_angle = sqrtf(roughness) * 0.05f;
        const RayCone ray_cone = pixel_ray_cone_from_image_height(fc, gbuffer_tex_size.y).propagate(reflected_cone_spread_angle, length(outgoing_ray.o - get_eye_position(fc)));
        rays_closest.fetch_add(1, std::memory_order_relaxed);
        const GbufferPathVertex primary_hit = gbuffer_raytrace(*in.scene, fc, outgoing_ray, 1, false, ray_cone);
        if (primary_hit.is_hit) {
            GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
            gbuffer.roughness = lerp(gbuffer.roughness, 1.0f, roughness_bias);
            const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
            const f3 wo = mul(-outgoing_ray.d, tangent_to_world);
            const LayeredBrdf brdf = LayeredBrdf::from_gbuffer_ndotv(in.brdf_fg_lut, gbuffer, wo.z);
            const f3 primary_hit_cs = position_world_to_sample(fc, primary_hit.position);
            const f2 primary_hit_uv = cs_to_uv(f2{primary_hit_cs.x, primary_hit_cs.y});
            const float primary_hit_screen_depth = sample_nearest_clamp(in.depth, primary_hit_uv);
            const u4 screen_gbuffer = in.gbuffer.ld(int(primary_hit_uv.x * gbuffer_tex_size.x), int(primary_hit_uv.y * gbuffer_tex_size.y));
            const f3 screen_normal_ws = unpack_normal_11_10_11(screen_gbuffer.y);
            const bool is_on_screen = fabsf(primary_hit_cs.x) < 1.0f && fabsf(primary_hit_cs.y) < 1.0f &&
                                      inverse_depth_relative_diff(primary_hit_cs.z, primary_hit_screen_depth) < 5e-3f &&
                                      dot(screen_normal_ws, -outgoing_ray.d) > 0.0f && dot(screen_normal_ws, gbuffer.normal) > 0.7f;
            f3 total_radiance = mk3(0.0f);
            {   // Sun (soft shadows, rng-driven)
                f2 urand;
                urand.x = uint_to_u01_float(hash1_mut(rng));
                urand.y = uint_to_u01_float(hash1_mut(rng));
                if (sun_color.x != 0 || sun_color.y != 0 || sun_color.z != 0) {
                    const f3 to_light_norm = sample_sun_direction(fc, urand, true);
                    rays_any.fetch_add(1, std::memory_order_relaxed);
                    const bool is_shadowed = in.scene->trace_any(Ray{primary_hit.position, 1e-4f, to_light_norm, SKY_DIST});
                    const f3 wi = mul(to_light_norm, tangent_to_world);
                    const f3 brdf_value = brdf.evaluate(wo, wi) * fmaxf(0.0f, wi.z);
                    total_radiance += brdf_value * (is_shadowed ? mk3(0.0f) : sun_color);
                }
            }
            const f3 reflected_normal_vs = direction_world_to_view(fc, gbuffer.normal);
            total_radiance += gbuffer.emissive;
            if (is_on_screen) {
                const f3 reprojected_radiance = xyz(unpack_rgba16f(sample_nearest_clamp(in.rtdgi_irradiance, primary_hit_uv))) * fc.pre_exposure_delta;
                total_radiance += reprojected_radiance * gbuffer.albedo;
            } else {
                f2 urand;
                urand.x = uint_to_u01_float(hash1_mut(rng));
                urand.y = uint_to_u01_float(hash1_mut(rng));
                const auto& lights = in.scene->triangle_lights;
                for (uint32_t li = 0; li < fc.triangle_light_count && li < lights.size(); ++li) {
                    const KjTriangleLight& tl = lights[li];
                    f3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
                    const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
                    const f3 to_light_ws = ls.pos - primary_hit.position;
                    const float dist_to_light2 = dot(to_light_ws, to_light_ws);
                    const f3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist_to_light2));
                    const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist_to_light2;
                    if (to_psa_metric > 0.0f) {
                        rays_any.fetch_add(1, std::memory_order_relaxed);
                        const bool is_shadowed = in.scene->trace_any(Ray{primary_hit.position, 1e-4f, to_light_norm_ws, sqrtf(dist_to_light2) - 2e-4f});
                        const f3 bounce_albedo = lerp(gbuffer.albedo, mk3(1.0f), 0.04f);
                        const f3 brdf_value = bounce_albedo * to_psa_metric / M_PI_F;
                        if (!is_shadowed) total_radiance += f3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf_value / ls.pdf;
                    }
                }
                if (in.ircache_lookup) {
                    const float cone_width = ray_cone.propagate(0.0f, primary_hit.ray_t).width;
                    const f3 gi = in.ircache_lookup(outgoing_ray.o, primary_hit.position, gbuffer.normal, 1, cone_width < 0.1f, rng);
                    total_radiance += gi * gbuffer.albedo;
                }
            }
            return RtrTraceResult{total_radiance, primary_hit.ray_t, reflected_normal_vs};
        }
        const f3 far_gi = xyz(sample_cube_rgba16f(in.sky_cube, in.sky_cube_width, outgoing_ray.d));
        return RtrTraceResult{far_gi, SKY_DIST, -direction_world_to_view(fc, outgoing_ray.d)};
    }

    // ------------------------------------------------------------------ reflection.rgen.hlsl:45-169
    void pass_trace(const FrameConstants& fc, const RtrInputs& in, Img<uint32_t> rng_out_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                const float depth = in.depth.ld(hx, hy);
                if (0.0f == depth) { st4(in.refl0_tex, x, y, mk4(0.0f, 0.0f, 0.0f, -SKY_DIST)); continue; }
                const f2 uv = get_uv(float(hx), float(hy), gbuffer_tex_size);
                GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(hx, hy));
                gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
                if (reuse_rtdgi_rays && gbuffer.roughness > 0.6f) continue;
                const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                const f3 refl_ray_origin_ws = vrc.biased_secondary_ray_origin_ws_with_normal(gbuffer.normal);
                f3 wo = mul(-vrc.ray_dir_ws(), tangent_to_world);
                if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
                SpecularBrdf specular_brdf;
                specular_brdf.albedo = lerp(mk3(0.04f), gbuffer.albedo, gbuffer.metalness);
                specular_brdf.roughness = gbuffer.roughness;
                const uint32_t noise_offset = fc.frame_index;
                uint32_t rng = hash3(uint32_t(x), uint32_t(y), noise_offset);
                f2 urand{blue_noise_sampler(in, x, y, int(noise_offset), 0), blue_noise_sampler(in, x, y, int(noise_offset), 1)};
                urand.x = lerp(urand.x, 0.0f, RTR_SAMPLING_BIAS);
                BrdfSample brdf_sample = specular_brdf.sample(wo, urand);
                for (uint32_t retry_i = 0; retry_i < 4 && !brdf_sample.is_valid(); ++retry_i) {
                    urand.x = uint_to_u01_float(hash1_mut(rng));
                    urand.y = uint_to_u01_float(hash1_mut(rng));
                    urand.x = lerp(urand.x, 0.0f, RTR_SAMPLING_BIAS);
                    brdf_sample = specular_brdf.sample(wo, urand);
                }
                if (brdf_sample.is_valid()) {
                    const float cos_theta = normalize(wo + brdf_sample.wi).z;
                    Ray outgoing_ray{refl_ray_origin_ws, 0.0f, mul(tangent_to_world, brdf_sample.wi), SKY_DIST};
                    rng_out_tex.st(x, y, rng);
                    const RtrTraceResult result = do_the_thing(fc, in, gbuffer.normal, gbuffer.roughness, rng, outgoing_ray);
                    const f3 hit_offset_ws = outgoing_ray.d * result.hit_t;
                    const SpecularBrdfEnergyPreservation brdf_lut = SpecularBrdfEnergyPreservation::from_brdf_ndotv(in.brdf_fg_lut, specular_brdf, wo.z);
                    const float pdf = brdf_sample.pdf / brdf_lut.valid_sample_fraction;
                    st4(in.refl0_tex, x, y, mk4(result.total_radiance, 1.0f - cos_theta));
                    st4(in.refl1_tex, x, y, mk4(hit_offset_ws, pdf));
                    in.refl2_tex.st(x, y, pack_rgba8_snorm(mk4(result.hit_normal_vs, 0.0f)));
                } else {
                    st4(in.refl0_tex, x, y, mk4(1.0f, 0.0f, 1.0f, 0.0f));
                    st4(in.refl1_tex, x, y, mk4(0.0f));
                }
            }
    }

    // ------------------------------------------------------------------ reflection_validate.rgen.hlsl:42-146 (one thread per 2x2 half-res quad)
    void pass_validate(const FrameConstants& fc, const RtrInputs& in, Img<f4> ray_orig_history_tex, ImgRGBA16F ray_history_tex, Img<uint32_t> rng_history_tex,
                       ImgRGBA16F irradiance_history_tex, ImgU2 reservoir_history_tex, ImgR8 refl_restir_invalidity_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const int qw = (hw + 1) / 2, qh = (hh + 1) / 2;
#pragma omp parallel for schedule(dynamic, 2)
        for (int qy = 0; qy < qh; ++qy)
            for (int qx = 0; qx < qw; ++qx) {
                const int x = qx * 2 + off.x, y = qy * 2 + off.y;
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                const float depth = in.depth.ld(hx, hy);
                if (0.0f == depth) { refl_restir_invalidity_tex.st(x, y, to_unorm8(1.0f)); continue; }
                GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(hx, hy));
                gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
                const f3 ray_orig_ws = xyz(ray_orig_history_tex.ld(x, y)) + get_prev_eye_position(fc);
                const f3 ray_offset = xyz(ld4(ray_history_tex, x, y));
                const f3 ray_hit_ws = ray_offset + ray_orig_ws;
                const f3 d = ray_hit_ws - ray_orig_ws;
                const float dl = length(d);
                Ray outgoing_ray{ray_orig_ws, 0.0f, dl > 0.0f ? d / dl : f3{0, 0, 1}, SKY_DIST};
                uint32_t rng = rng_history_tex.ld(x, y);
                const RtrTraceResult result = do_the_thing(fc, in, gbuffer.normal, gbuffer.roughness, rng, outgoing_ray);
                Reservoir1spp r = Reservoir1spp::from_raw(reservoir_history_tex.ld(x, y));
                const f4 prev_irradiance_packed = ld4(irradiance_history_tex, x, y);
                const f3 prev_irradiance = vmax(mk3(0.0f), xyz(prev_irradiance_packed) * fc.pre_exposure_delta);
                const f3 check_radiance = vmax(mk3(0.0f), result.total_radiance);
                const float rad_diff = length(vabs(prev_irradiance - check_radiance) / vmax(mk3(1e-3f), prev_irradiance + check_radiance));
                const float invalidity = smoothstep(0.1f, 0.5f, rad_diff / length(mk3(1.0f)));
                r.M *= 1.0f - invalidity;
                st4(irradiance_history_tex, x, y, mk4(check_radiance, prev_irradiance_packed.w));
                refl_restir_invalidity_tex.st(x, y, to_unorm8(invalidity));
                reservoir_history_tex.st(x, y, r.as_raw());
                for (uint32_t i = 1; i <= 3; ++i) {
                    const i2 o = hi_px_subpixel(fc.frame_index + i);
                    const int nx = qx * 2 + o.x, ny = qy * 2 + o.y;
                    const f4 neighbor_prev_irradiance_packed = ld4(irradiance_history_tex, nx, ny);
                    const f3 a = vmax(mk3(0.0f), xyz(neighbor_prev_irradiance_packed) * fc.pre_exposure_delta);
                    const f3 b = prev_irradiance;
                    const float neigh_rad_diff = length(vabs(a - b) / vmax(mk3(1e-8f), a + b));
                    if (neigh_rad_diff < 0.2f) st4(irradiance_history_tex, nx, ny, mk4(check_radiance, neighbor_prev_irradiance_packed.w));
                    refl_restir_invalidity_tex.st(nx, ny, to_unorm8(invalidity));
                    if (invalidity > 0.0f) {
                        Reservoir1spp rn = Reservoir1spp::from_raw(reservoir_history_tex.ld(nx, ny));
                        rn.M *= 1.0f - invalidity;
                        reservoir_history_tex.st(nx, ny, rn.as_raw());
                    }
                }
            }
    }

    // ------------------------------------------------------------------ rtr_restir_temporal.hlsl:105-153
    void find_best_reprojection_in_neighborhood(const FrameConstants& fc, const Img<f4>& ray_orig_history_tex, f2 base_px, i2& best_px, f3 refl_ray_origin_ws, bool wide) const {
        float best_dist = 1e10f;
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const f2 clip_scale{fc.view_constants.clip_to_view[0], fc.view_constants.clip_to_view[5]};
        const f2 offset_scale{1.0f * -2.0f * clip_scale.x * gbuffer_tex_size.z, -1.0f * -2.0f * clip_scale.y * gbuffer_tex_size.w};
        const f3 look_direction = direction_view_to_world(fc, f3{0, 0, -1});
        const i2 off = halfres_subsample_offset(fc);
        {
            const float z_offset = dot(look_direction, refl_ray_origin_ws - get_eye_position(fc));
            refl_ray_origin_ws += direction_view_to_world(fc, f3{float(off.x) * offset_scale.x * z_offset, float(off.y) * offset_scale.y * z_offset, 0.0f});
        }
        const int start_coord = wide ? -1 : 0;
        for (int y = start_coord; y <= 1; ++y)
            for (int x = start_coord; x <= 1; ++x) {
                const i2 spx{int(floorf(base_px.x + float(x))), int(floorf(base_px.y + float(y)))};
                const RtrRestirRayOrigin ray_orig = RtrRestirRayOrigin::from_raw(ray_orig_history_tex.ld(spx.x, spx.y));
                f3 orig = ray_orig.ray_origin_eye_offset_ws + get_prev_eye_position(fc);
                const i2 orig_jitter = hi_px_subpixel(ray_orig.frame_index_mod4);
                {
                    const float z_offset = dot(look_direction, orig);
                    orig += direction_view_to_world(fc, f3{float(orig_jitter.x) * offset_scale.x * z_offset, float(orig_jitter.y) * offset_scale.y * z_offset, 0.0f});
                }
                const float d = length(orig - refl_ray_origin_ws);
                if (d < best_dist) { best_dist = d; best_px = spx; }
            }
    }

    // ------------------------------------------------------------------ rtr_restir_temporal.hlsl:155-533
    void pass_restir_temporal(const FrameConstants& fc, const RtrInputs& in, ImgRGBA16F irradiance_history_tex, Img<f4> ray_orig_history_tex, ImgRGBA16F ray_history_tex,
                              Img<uint32_t> rng_history_tex, ImgU2 reservoir_history_tex, ImgRGBA16F hit_normal_history_tex, ImgRGBA16F irradiance_out_tex,
                              Img<f4> ray_orig_output_tex, ImgRGBA16F ray_output_tex, Img<uint32_t> rng_output_tex, ImgRGBA16F hit_normal_output_tex, ImgU2 reservoir_out_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                const float depth = in.depth.ld(hx, hy);
                if (0.0f == depth) {
                    st4(irradiance_out_tex, x, y, mk4(0.0f, 0.0f, 0.0f, -SKY_DIST));
                    st4(hit_normal_output_tex, x, y, mk4(0.0f));
                    reservoir_out_tex.st(x, y, u2{0, 0});
                    continue;
                }
                const f2 uv = get_uv(float(hx), float(hy), gbuffer_tex_size);
                const f3 normal_vs = ld_nrm_snorm8(in.half_view_normal, x, y);
                const f3 normal_ws = direction_view_to_world(fc, normal_vs);
                float local_normal_flatness = 1.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) local_normal_flatness *= saturate(dot(normal_vs, ld_nrm_snorm8(in.half_view_normal, x + dx, y + dy)));
                float reprojection_neighborhood_stability = 1.0f;
                for (int dy = 0; dy <= 1; ++dy)
                    for (int dx = 0; dx <= 1; ++dx) reprojection_neighborhood_stability *= ld_reproj(in.reprojection_map, x * 2 + dx, y * 2 + dy).z;
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                const f3 refl_ray_origin_ws = vrc.biased_secondary_ray_origin_ws_with_normal(normal_ws);
                const f3 refl_ray_origin_vs = position_world_to_view(fc, refl_ray_origin_ws);
                f3 outgoing_dir{0, 0, 1};
                uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index);
                const GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(hx, hy));
                const float a2 = fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness) * fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness);
                float pdf_sel = 0.0f, cos_theta = 0.0f;
                f3 irradiance_sel = mk3(0.0f);
                f4 ray_orig_sel = mk4(0.0f);
                f3 ray_hit_sel_ws = mk3(1.0f), hit_normal_sel = mk3(1.0f);
                uint32_t rng_sel = rng_output_tex.ld(x, y);
                StreamState stream_state;
                Reservoir1spp reservoir;
                const uint32_t reservoir_payload = uint32_t(x) | (uint32_t(y) << 16);
                reservoir.payload = reservoir_payload;
                {
                    const f4 hit0 = ld4(in.refl0_tex, x, y), hit1 = ld4(in.refl1_tex, x, y);
                    const f3 hit2 = xyz(unpack_rgba8_snorm(in.refl2_tex.ld(x, y)));
                    const f3 out_value = xyz(hit0);
                    const float pdf = fminf(hit1.w, RTR_RESTIR_MAX_PDF_CLAMP);
                    const f3 hit_vs = xyz(hit1);
                    if (pdf > 0.0f) {
                        outgoing_dir = normalize(hit_vs);
                        const float p_q = fmaxf(1e-3f, sRGB_to_luminance(out_value)) * pdf;
                        const float inv_pdf_q = 1.0f / pdf;
                        pdf_sel = pdf;
                        cos_theta = 1.0f - hit0.w;
                        irradiance_sel = out_value;
                        RtrRestirRayOrigin ray_orig;
                        ray_orig.ray_origin_eye_offset_ws = refl_ray_origin_ws;
                        ray_orig.roughness = gbuffer.roughness;
                        ray_orig.frame_index_mod4 = fc.frame_index & 3u;
                        ray_orig_sel = ray_orig.to_raw();
                        ray_hit_sel_ws = hit_vs + refl_ray_origin_ws;
                        hit_normal_sel = direction_view_to_world(fc, hit2);
                        if (p_q * inv_pdf_q > 0.0f) reservoir.init_with_stream(p_q, inv_pdf_q, stream_state, reservoir_payload);
                    }
                }
                const f4 center_reproj = ld_reproj(in.reprojection_map, hx, hy);
                {
                    const float ang_offset = float(((fc.frame_index + 7u) * 11u) % 32u) * M_TAU_F;
                    const uint32_t sample_count = center_reproj.z < 1.0f ? 5u : 1u;
                    for (uint32_t sample_i = 0; sample_i < sample_count && stream_state.M_sum < RTR_RESTIR_TEMPORAL_M_CLAMP; ++sample_i) {
                        const float ang = (float(sample_i) + ang_offset) * GOLDEN_ANGLE;
                        const float rpx_offset_radius = sqrtf(float(((sample_i - 1u) + fc.frame_index) & 3u) + 1.0f) * clampf(8.0f - stream_state.M_sum, 1.0f, 7.0f);
                        const f2 reservoir_px_offset_base{cosf(ang) * rpx_offset_radius, sinf(ang) * rpx_offset_radius};
                        const i2 rpx_offset = sample_i == 0 ? i2{0, 0} : i2{int(reservoir_px_offset_base.x), int(reservoir_px_offset_base.y)};
                        const f4 reproj = ld_reproj(in.reprojection_map, hx + rpx_offset.x * 2, hy + rpx_offset.y * 2);
                        const f2 base_px{float(x) + gbuffer_tex_size.x * reproj.x / 2.0f, float(y) + gbuffer_tex_size.y * reproj.y / 2.0f};
                        i2 best_px{int(floorf(base_px.x + 0.5f)), int(floorf(base_px.y + 0.5f))};
                        if (reprojection_neighborhood_stability >= 1.0f) {
                            if (fabsf(gbuffer_tex_size.x * reproj.x) > 0.1f || fabsf(gbuffer_tex_size.y * reproj.y) > 0.1f)
                                find_best_reprojection_in_neighborhood(fc, ray_orig_history_tex, base_px, best_px, refl_ray_origin_ws, false);
                        } else {
                            find_best_reprojection_in_neighborhood(fc, ray_orig_history_tex, base_px, best_px, refl_ray_origin_ws, true);
                        }
                        const i2 rpx{best_px.x + rpx_offset.x, best_px.y + rpx_offset.y};
                        Reservoir1spp r = Reservoir1spp::from_raw(reservoir_history_tex.ld(rpx.x, rpx.y));
                        const int spx = int(r.payload & 0xffffu), spy = int(r.payload >> 16);
                        f4 prev_ray_orig_and_roughness = ray_orig_history_tex.ld(spx, spy);
                        const f3 pe = get_prev_eye_position(fc);
                        prev_ray_orig_and_roughness.x += pe.x; prev_ray_orig_and_roughness.y += pe.y; prev_ray_orig_and_roughness.z += pe.z;   // .w: packed bits, untouched
                        const f3 prev_orig = xyz(prev_ray_orig_and_roughness);
                        const f3 od = refl_ray_origin_ws - prev_orig;
                        if (dot(od, od) > 0.05f * refl_ray_origin_vs.z * refl_ray_origin_vs.z) continue;
                        const f4 prev_irrad_raw = ld4(irradiance_history_tex, spx, spy);
                        const f3 prev_irrad = xyz(prev_irrad_raw) * fc.pre_exposure_delta;
                        const float prev_cos_theta = 1.0f - prev_irrad_raw.w;
                        const f4 sample_hit_ws_and_pdf_packed = ld4(ray_history_tex, spx, spy);
                        const float prev_pdf = sample_hit_ws_and_pdf_packed.w;
                        const f3 sample_hit_ws = xyz(sample_hit_ws_and_pdf_packed) + prev_orig;
                        const float prev_dist = length(xyz(sample_hit_ws_and_pdf_packed));
                        const f4 hn_raw = ld4(hit_normal_history_tex, spx, spy);
                        const f4 sample_hit_normal_ws_dot{hn_raw.x * 2.0f - 1.0f, hn_raw.y * 2.0f - 1.0f, hn_raw.z * 2.0f - 1.0f, hn_raw.w};
                        const f3 dir_to_sample_hit_unnorm = sample_hit_ws - refl_ray_origin_ws;
                        const float dist_to_sample_hit = length(dir_to_sample_hit_unnorm);
                        const f3 dir_to_sample_hit = normalize(dir_to_sample_hit_unnorm);
                        r.M = fminf(r.M, RTR_RESTIR_TEMPORAL_M_CLAMP);
                        {   // USE_TRANSLATIONAL_CLAMP
                            const f3 current_wo = normalize(vrc.ray_hit_ws() - get_eye_position(fc));
                            const f3 prev_wo = normalize(vrc.ray_hit_ws() - get_prev_eye_position(fc));
                            const float wo_dot = saturate(dot(current_wo, prev_wo));
                            const float wo_similarity = powf(saturate(ggx_ndf_0_1(fmaxf(3e-5f, a2), wo_dot)), 64.0f);
                            float mult = lerp(wo_similarity, 1.0f, smoothstep(0.05f, 0.5f, sqrtf(gbuffer.roughness)));
                            mult = lerp(1.0f, mult, local_normal_flatness);
                            r.M *= mult;
                        }
                        float p_q = 1.0f;
                        p_q *= fmaxf(1e-3f, sRGB_to_luminance(prev_irrad));
                        p_q *= step(0.0f, dot(dir_to_sample_hit, normal_ws));
                        p_q *= prev_pdf;
                        float jacobian = 1.0f;
                        jacobian *= clampf(prev_dist / dist_to_sample_hit, 1e-4f, 1e4f);
                        jacobian *= jacobian;
                        jacobian *= fmaxf(0.0f, -dot(xyz(sample_hit_normal_ws_dot), dir_to_sample_hit)) / fmaxf(1e-5f, sample_hit_normal_ws_dot.w);
                        {   // USE_JACOBIAN_BASED_REJECTION
                            const float threshold = lerp(1.1f, 4.0f, gbuffer.roughness * gbuffer.roughness);
                            if (!(jacobian < threshold && jacobian > 1.0f / threshold)) continue;
                        }
                        p_q *= jacobian;
                        if (reservoir.update_with_stream(r, p_q, 1.0f, stream_state, reservoir_payload, rng)) {
                            outgoing_dir = dir_to_sample_hit;
                            pdf_sel = prev_pdf;
                            cos_theta = prev_cos_theta;
                            irradiance_sel = prev_irrad;
                            ray_orig_sel = prev_ray_orig_and_roughness;
                            ray_hit_sel_ws = sample_hit_ws;
                            hit_normal_sel = xyz(sample_hit_normal_ws_dot);
                            rng_sel = rng_history_tex.ld(spx, spy);
                        }
                    }
                    reservoir.finish_stream(stream_state);
                    reservoir.W = fminf(reservoir.W, 1e20f);
                }
                const f4 hit_normal_ws_dot = mk4(hit_normal_sel, -dot(hit_normal_sel, outgoing_dir));
                st4(irradiance_out_tex, x, y, mk4(irradiance_sel, 1.0f - cos_theta));
                const f3 eye = get_eye_position(fc);
                ray_orig_output_tex.st(x, y, f4{ray_orig_sel.x - eye.x, ray_orig_sel.y - eye.y, ray_orig_sel.z - eye.z, ray_orig_sel.w});
                st4(hit_normal_output_tex, x, y, f4{hit_normal_ws_dot.x * 0.5f + 0.5f, hit_normal_ws_dot.y * 0.5f + 0.5f, hit_normal_ws_dot.z * 0.5f + 0.5f, hit_normal_ws_dot.w});
                st4(ray_output_tex, x, y, mk4(ray_hit_sel_ws - xyz(ray_orig_sel), pdf_sel));
                rng_output_tex.st(x, y, rng_sel);
                reservoir_out_tex.st(x, y, reservoir.as_raw());
            }
    }

    // ------------------------------------------------------------------ resolve.hlsl:66-663 (USE_RESTIR, CUT_CORNERS_IN_MATH, BORROW_SAMPLES)
    static void get_specular_filter_kernel_basis(f3 v, f3 n, float roughness, float scale, f3& t1, f3& t2) {
        const f3 r = reflect(-v, n);                             // specular_dominant_direction (brdf.hlsl:313-317)
        const float f = (1.0f - roughness) * (sqrtf(1.0f - roughness) + roughness);
        const f3 dominant = normalize(lerp(n, r, f));
        const f3 reflected = reflect(-dominant, n);
        t1 = normalize(cross(n, reflected)) * scale;
        t2 = cross(reflected, t1);
    }
    void pass_resolve(const FrameConstants& fc, const RtrInputs& in, ImgRG16F ray_len_history_tex, ImgRGBA16F restir_irradiance_tex, ImgRGBA16F restir_ray_tex,
                      ImgU2 restir_reservoir_tex, Img<f4> restir_ray_orig_tex, ImgRGBA16F restir_hit_normal_tex, Img<uint32_t> output_tex, ImgRG16F ray_len_output_tex) {
        const f4 output_tex_size = tex_size4(W, H);
        const uint32_t MAX_SAMPLE_COUNT = 8;
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int hpx = x / 2, hpy = y / 2;
                const f2 uv = get_uv(float(x), float(y), output_tex_size);
                const float depth = in.depth.ld(x, y);
                if (0.0f == depth) { output_tex.st(x, y, pack_r11g11b10f(mk3(0.0f))); continue; }
                GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(x, y));
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                const f3 refl_ray_origin_ws = vrc.biased_secondary_ray_origin_ws_with_normal(gbuffer.normal);
                const f3 refl_ray_origin_vs = position_world_to_view(fc, refl_ray_origin_ws);
                gbuffer.roughness = fmaxf(gbuffer.roughness, RTR_ROUGHNESS_CLAMP);
                const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
                f3 wo = mul(-normalize(vrc.ray_dir_ws()), tangent_to_world);
                if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
                const SpecularBrdf specular_brdf = LayeredBrdf::from_gbuffer_ndotv(in.brdf_fg_lut, gbuffer, wo.z).specular_brdf;
                const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
                const float a2 = fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness) * fmaxf(RTR_ROUGHNESS_CLAMP, gbuffer.roughness);
                const float surf_to_hit_dist = length(xyz(ld4(in.refl1_tex, hpx, hpy)));
                const float eye_to_surf_dist = length(refl_ray_origin_vs);
                const float eye_ray_z_scale = -vrc.ray_dir_vs().z;
                const f4 reprojection_params = ld_reproj(in.reprojection_map, x, y);
                const float ray_squish_scale = 16.0f / fmaxf(1e-5f, eye_to_surf_dist);
                const float ray_len_avg = exponential_unsquish(lerp(
                    exponential_squish(sample_bilinear_clamp(ray_len_history_tex, f2{uv.x + reprojection_params.x, uv.y + reprojection_params.y}).y, ray_squish_scale),
                    exponential_squish(surf_to_hit_dist, ray_squish_scale), 0.1f), ray_squish_scale);
                f4 contrib_accum = mk4(0.0f);
                float ray_len_accum = 0.0f;
                const f3 normal_vs = direction_world_to_view(fc, gbuffer.normal);
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
                    kernel_size_ws = fmaxf(kernel_size_ws, output_tex_size.w * 4.0f * scale_factor);
                }
                f3 kernel_t1, kernel_t2;
                get_specular_filter_kernel_basis(-normalize(vrc.ray_dir_ws()), gbuffer.normal, gbuffer.roughness, kernel_size_ws, kernel_t1, kernel_t2);
                const f4 blue = blue_noise_for_pixel(in.blue_noise, uint32_t(hpx + 16), uint32_t(hpy + 16), fc.frame_index);
                const float KERNEL_SHARPNESS = 0.666f;
                const float RADIUS_SAMPLE_MULT = 1.0f / powf(float(MAX_SAMPLE_COUNT), KERNEL_SHARPNESS);
                const float ang_offset = float(fc.frame_index * 59u % 128u) * M_PLASTIC_F;
                const float RADIUS_INC_ON_FAIL = 0.25f;
                float sample_radius_accum = 1.0f;
                for (uint32_t sample_i = 1; sample_i <= MAX_SAMPLE_COUNT; ++sample_i, sample_radius_accum += RADIUS_INC_ON_FAIL) {
                    const bool is_center_sample = sample_i == MAX_SAMPLE_COUNT;
                    int sample_px_x, sample_px_y;
                    {
                        const float ang = (float(sample_i) + ang_offset) * GOLDEN_ANGLE + (float(px_idx_in_quad) / 4.0f) * M_TAU_F;
                        float sample_i_with_jitter = sample_radius_accum;
                        if (is_center_sample) sample_i_with_jitter = contrib_accum.w > 1e-8f ? blue.y : 0.0f;
                        else sample_i_with_jitter += blue.y;
                        const float radius = powf(sample_i_with_jitter, KERNEL_SHARPNESS) * RADIUS_SAMPLE_MULT;
                        const f3 offset_ws = (cosf(ang) * kernel_t1 + sinf(ang) * kernel_t2) * radius;
                        const f3 sample_ws = refl_ray_origin_ws + offset_ws;
                        const f3 sample_cs = position_world_to_sample(fc, sample_ws);
                        const f2 sample_uv = cs_to_uv(f2{sample_cs.x, sample_cs.y});
                        sample_px_x = int(floorf(sample_uv.x * output_tex_size.x / 2.0f));
                        sample_px_y = int(floorf(sample_uv.y * output_tex_size.y / 2.0f));
                    }
                    float rejection_bias = 1.0f;
                    const f3 sample_normal_vs = ld_nrm_snorm8(in.half_view_normal, sample_px_x, sample_px_y);
                    float pdf0_mult = 1.0f, pdf1_mult = 1.0f;
                    const float bent_pdf_ndotl_fix = 1.0f;
                    const u2 reservoir_raw = restir_reservoir_tex.ld(sample_px_x, sample_px_y);
                    const Reservoir1spp r = Reservoir1spp::from_raw(reservoir_raw);
                    const int spx = int(r.payload & 0xffffu), spy = int(r.payload >> 16);
                    const RtrRestirRayOrigin sample_origin = RtrRestirRayOrigin::from_raw(restir_ray_orig_tex.ld(spx, spy));
                    const f3 sample_origin_ws = sample_origin.ray_origin_eye_offset_ws + get_eye_position(fc);
                    const float sample_roughness = sample_origin.roughness;
                    if (reservoir_raw.x == 0 || sample_roughness > gbuffer.roughness * 2.0f) continue;
                    const f4 restir_ray = ld4(restir_ray_tex, spx, spy);
                    const f3 sample_hit_ws = xyz(restir_ray) + sample_origin_ws;
                    const f3 sample_origin_vs = position_world_to_view(fc, sample_origin_ws);
                    const f4 restir_irr = ld4(restir_irradiance_tex, spx, spy);
                    const f3 sample_radiance = xyz(restir_irr);
                    const float sample_ray_pdf = restir_ray.w;
                    float neighbor_sampling_pdf = 1.0f / r.W;
                    const f3 sample_hit_vs_abs = position_world_to_view(fc, sample_hit_ws);
                    const f3 center_to_hit_vs = sample_hit_vs_abs - lerp(refl_ray_origin_vs, sample_origin_vs, RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS);
                    const float sample_cos_theta = 1.0f - restir_irr.w;
                    const float center_to_hit_dist = length(center_to_hit_vs);
                    const float sample_to_hit_dist = length(sample_hit_ws - sample_origin_ws);
                    {   // RTR_USE_BULLSHIT_TO_FIX_EDGE_HALOS
                        const float d = length(sample_hit_vs_abs - lerp(refl_ray_origin_vs, sample_origin_vs,
                                                                       lerp(1.0f, RTR_NEIGHBOR_RAY_ORIGIN_CENTER_BIAS, 0.4f * fminf(1.0f, 3.0f * sqrtf(gbuffer.roughness)))));
                        pdf0_mult *= fmaxf(1e-5f, powf(d / sample_to_hit_dist, 2.0f));
                        pdf1_mult *= fmaxf(1.0f, powf(center_to_hit_dist / sample_to_hit_dist, 2.0f));
                    }
                    const f3 wi = normalize(mul(direction_view_to_world(fc, center_to_hit_vs), tangent_to_world));
                    if (wi.z < 1e-5f) continue;
                    rejection_bias *= dot(normal_vs, sample_normal_vs) > 0.7f ? 1.0f : 0.0f;
                    {
                        const float depth_diff = fabsf(refl_ray_origin_vs.z - sample_origin_vs.z) / fmaxf(1e-10f, kernel_size_ws);
                        rejection_bias *= exp2f(-fmaxf(0.3f, normal_vs.z) * depth_diff * depth_diff);
                    }
                    const f3 surface_offset = sample_origin_vs - refl_ray_origin_vs;
                    // For the pixel that IS the half-res sample, surface_offset is the rounding residue of (origin - eye) + eye: exactly zero
                    // gives the shader 0/0 = NaN (comparison false, no rejection), a 1-ulp residue gives it a random direction. Residues are
                    // treated as zero (see the header): real neighbours are >= a pixel footprint (~4e-3 x distance) apart.
                    const float surface_offset_len = length(surface_offset);
                    if ((literal_own_sample_shadowing || surface_offset_len > 1e-5f * eye_to_surf_dist) &&
                        dot(center_to_hit_vs, normal_vs) * 0.2f / length(center_to_hit_vs) < dot(surface_offset, normal_vs) / surface_offset_len)
                        rejection_bias *= is_center_sample ? 1.0f : 0.0f;
                    const BrdfValue spec = specular_brdf.evaluate(wo, wi);
                    const float spec_weight = spec.pdf * step(0.0f, wi.z);
                    float contrib_wt = 0.0f;
                    {
                        const float cos_theta = normalize(wo + wi).z;
                        const float bent_cos_theta = fminf(sample_cos_theta, cos_theta * 1.25f);
                        const float sample_ray_ndf = ggx_ndf(a2, bent_cos_theta);
                        const float center_ndf = ggx_ndf(a2, cos_theta);
                        const float bent_sample_pdf0 = spec.pdf * sample_ray_ndf / center_ndf;
                        const float pdf_lerp_t = smoothstep(0.4f, 0.7f, sqrtf(gbuffer.roughness)) * smoothstep(0.0f, 0.1f, ray_len_avg / eye_to_surf_dist);
                        const f3 pdfs[2] = {
                            f3{fminf(bent_sample_pdf0, RTR_RESTIR_MAX_PDF_CLAMP) * bent_pdf_ndotl_fix, neighbor_sampling_pdf * pdf0_mult, 1.0f - pdf_lerp_t},
                            f3{fminf(spec.pdf, RTR_RESTIR_MAX_PDF_CLAMP), neighbor_sampling_pdf * pdf1_mult, pdf_lerp_t}};
                        for (int pdf_i = 0; pdf_i < 2; ++pdf_i) {
                            const float bent_sample_pdf = pdfs[pdf_i].x, nsp = pdfs[pdf_i].y, pdf_influence = pdfs[pdf_i].z;
                            const float mis_weight = fmaxf(1e-4f, spec.pdf / (sample_ray_pdf + spec.pdf));
                            contrib_wt = rejection_bias * mis_weight * fmaxf(1e-10f, spec_weight / bent_sample_pdf);
                            contrib_accum = contrib_accum + mk4(sample_radiance * bent_sample_pdf / nsp * spec.value_over_pdf, 1.0f) * contrib_wt * pdf_influence;
                        }
                    }
                    ray_len_accum += exponential_squish(surf_to_hit_dist, ray_squish_scale) * contrib_wt;
                    sample_radius_accum += 1.0f - RADIUS_INC_ON_FAIL;
                }
                const float contrib_norm_factor = fmaxf(1e-14f, contrib_accum.w);
                f3 rgb = xyz(contrib_accum) / contrib_norm_factor;
                ray_len_accum /= contrib_norm_factor;
                const SpecularBrdfEnergyPreservation brdf_lut = SpecularBrdfEnergyPreservation::from_brdf_ndotv(in.brdf_fg_lut, specular_brdf, wo.z);
                rgb = rgb / brdf_lut.preintegrated_reflection;            // !RTR_RENDER_SCALED_BY_FG
                rgb = rgb * brdf_lut.preintegrated_reflection_mult;
                ray_len_accum = exponential_unsquish(ray_len_accum, ray_squish_scale);
                output_tex.st(x, y, pack_r11g11b10f(rgb));
                st2(ray_len_output_tex, x, y, f2{ray_len_accum, ray_len_avg});
            }
    }

    // ------------------------------------------------------------------ temporal_filter.hlsl:37-259
    void pass_temporal_filter(const FrameConstants& fc, const RtrInputs& in, Img<uint32_t> input_tex, ImgRGBA16F history_tex, ImgRG16F ray_len_tex,
                              ImgR8 refl_restir_invalidity_tex, ImgRGBA16F output_tex) {
        const f4 output_tex_size = tex_size4(W, H);
        auto ld_in = [&](int x, int y) { return input_tex.inb(x, y) ? mk4(unpack_r11g11b10f(input_tex.ld(x, y)), 1.0f) : mk4(0.0f); };
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f4 center = linear_rgb_to_crunched_luma_chroma(ld_in(x, y));
                const float refl_ray_length = clampf(ld2(ray_len_tex, x, y).x, 0.0f, 1e3f);
                const f2 uv = get_uv(float(x), float(y), output_tex_size);
                const float center_depth = in.depth.ld(x, y);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, center_depth);
                const f3 reflector_vs = vrc.ray_hit_vs();
                const f3 reflection_hit_vs = reflector_vs + vrc.ray_dir_vs() * refl_ray_length;
                const f4 reflection_hit_cs = mul44(fc.view_constants.view_to_sample, mk4(reflection_hit_vs, 1.0f));
                const f4 prev_hit_cs = mul44(fc.view_constants.clip_to_prev_clip, reflection_hit_cs);
                f2 hit_prev_uv = cs_to_uv(f2{prev_hit_cs.x / prev_hit_cs.w, prev_hit_cs.y / prev_hit_cs.w});
                const f4 prev_reflector_cs = mul44(fc.view_constants.clip_to_prev_clip, vrc.ray_hit_cs);
                const f2 reflector_prev_uv = cs_to_uv(f2{prev_reflector_cs.x / prev_reflector_cs.w, prev_reflector_cs.y / prev_reflector_cs.w});
                const f4 reproj = ld_reproj(in.reprojection_map, x, y);
                const float reflector_move_rate = fminf(1.0f, length(f2{reproj.x, reproj.y}) / length(reflector_prev_uv - uv));
                hit_prev_uv = lerp(uv, hit_prev_uv, reflector_move_rate);
                const uint32_t quad_reproj_valid_packed = uint32_t(reproj.z * 15.0f + 0.5f);
                const f4 history_mult{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1.0f};
                f4 history0 = mk4(0.0f);
                float history0_valid = 1.0f;
                const f2 reproj_uv{uv.x + reproj.x, uv.y + reproj.y};
                if (0 == quad_reproj_valid_packed) {
                    history0_valid = 0.0f;
                } else if (15 == quad_reproj_valid_packed) {
                    history0 = vmax(mk4(0.0f), Taa::catmull_rom_5tap(history_tex, reproj_uv, f2{output_tex_size.x, output_tex_size.y}, [](f4 v) { return v; })) * history_mult;
                } else {
                    const f4 qv{(quad_reproj_valid_packed & 1) ? 1.0f : 0.0f, (quad_reproj_valid_packed & 2) ? 1.0f : 0.0f, (quad_reproj_valid_packed & 4) ? 1.0f : 0.0f,
                                (quad_reproj_valid_packed & 8) ? 1.0f : 0.0f};
                    const Bilinear bl = get_bilinear_filter(reproj_uv, f2{output_tex_size.x, output_tex_size.y});
                    const int ox = int(bl.origin.x), oy = int(bl.origin.y);
                    const f4 s00 = ld4(history_tex, ox, oy) * history_mult, s10 = ld4(history_tex, ox + 1, oy) * history_mult;
                    const f4 s01 = ld4(history_tex, ox, oy + 1) * history_mult, s11 = ld4(history_tex, ox + 1, oy + 1) * history_mult;
                    f4 w{(1.0f - bl.weights.x) * (1.0f - bl.weights.y), bl.weights.x * (1.0f - bl.weights.y), (1.0f - bl.weights.x) * bl.weights.y, bl.weights.x * bl.weights.y};
                    w = w * qv;
                    const float wsum = dot(w, mk4(1.0f));
                    if (wsum > 1e-5f) history0 = (s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w) * (1.0f / wsum);   // apply_bilinear_custom_weights (inc/bilinear.hlsl)
                    else history0 = (s00 + s10 + s01 + s11) / 4.0f;
                }
                history0 = linear_rgb_to_crunched_luma_chroma(history0);
                const f4 history1 = linear_rgb_to_crunched_luma_chroma(sample_bilinear_clamp(history_tex, hit_prev_uv) * history_mult);
                const float history1_valid = quad_reproj_valid_packed == 15 ? 1.0f : 0.0f;
                f4 vsum = mk4(0.0f), vsum2 = mk4(0.0f);
                float wsum = 0.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float sample_depth = in.depth.ld(x + dx, y + dy);
                        const f4 neigh = linear_rgb_to_crunched_luma_chroma(ld_in(x + dx, y + dy));
                        const float w = exp2f(-200.0f * fabsf(center_depth / sample_depth - 1.0f));
                        vsum = vsum + neigh * w;
                        vsum2 = vsum2 + neigh * neigh * w;
                        wsum += w;
                    }
                const f4 ex = vsum / wsum, ex2 = vsum2 / wsum;
                const f4 dev = vsqrt(vmax(mk4(0.0f), ex2 - ex * ex));
                const GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(x, y));
                const float restir_invalidity = from_unorm8(refl_restir_invalidity_tex.ld(x / 2, y / 2));
                const float n_deviations = lerp(reproj.z > 0.0f ? 2.0f : 1.25f, 0.625f, restir_invalidity);
                float wo_similarity;
                {
                    const f3 current_wo = normalize(vrc.ray_hit_ws() - get_eye_position(fc));
                    const f3 prev_wo = normalize(vrc.ray_hit_ws() - get_prev_eye_position(fc));
                    const float clamped_roughness = fmaxf(0.1f, gbuffer.roughness);
                    wo_similarity = powf(saturate(ggx_ndf_0_1(clamped_roughness * clamped_roughness, dot(current_wo, prev_wo))), 32.0f);
                }
                const float h0diff = length((xyz(history0) - xyz(ex)) / xyz(dev));
                const float h1diff = length((xyz(history1) - xyz(ex)) / xyz(dev));
                float h0_score = 1.0f * smoothstep(0.0f, 0.5f, sqrtf(gbuffer.roughness)) * lerp(wo_similarity, 1.0f, sqrtf(gbuffer.roughness));
                float h1_score = (1.0f - h0_score) * lerp(1.0f, smoothstep(0.0f, 1.0f, h0diff - h1diff), smoothstep(0.0f, 0.15f, sqrtf(gbuffer.roughness)));
                h0_score *= history0_valid;
                h1_score *= history1_valid;
                const float score_sum = h0_score + h1_score;
                h0_score /= score_sum;
                h1_score = 1.0f - h0_score;
                if (!(h0_score < 1.001f)) { h0_score = 1.0f; h1_score = 0.0f; }
                f4 clamped_history0 = history0, clamped_history1 = history1;
                {
                    const f3 c0 = soft_color_clamp(xyz(center), xyz(history0), xyz(ex), xyz(dev) * n_deviations);
                    const f3 c1 = soft_color_clamp(xyz(center), xyz(history1), xyz(ex), xyz(dev) * n_deviations);
                    clamped_history0 = mk4(c0, history0.w);
                    clamped_history1 = mk4(c1, history1.w);
                }
                const f4 clamped_history = clamped_history0 * h0_score + clamped_history1 * h1_score;
                const float max_sample_count = 16.0f;
                const float current_sample_count = clamped_history.w * saturate(h0_score * history0_valid + h1_score * history1_valid);
                f4 res = lerp(clamped_history, center, 1.0f / (1.0f + fminf(max_sample_count, current_sample_count * lerp(wo_similarity, 1.0f, 0.5f))));
                res.w = fminf(current_sample_count, max_sample_count) + 1.0f;
                res = crunched_luma_chroma_to_linear_rgb(res);
                st4(output_tex, x, y, vmax(mk4(0.0f), res));
            }
    }

    // ------------------------------------------------------------------ spatial_cleanup.hlsl:20-65
    void pass_cleanup(const FrameConstants& fc, const RtrInputs& in, ImgRGBA16F input_tex, Img<uint32_t> output_tex) {
        const float min_sample_count = 8.0f;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f4 center = ld4(input_tex, x, y);
                const float center_depth = in.depth.ld(x, y);
                const float center_sample_count = center.w;
                if (center_sample_count >= min_sample_count || center_depth == 0.0f) { output_tex.st(x, y, pack_r11g11b10f(xyz(center))); continue; }
                const f3 center_normal_vs = unpack_a2r10g10b10(in.geometric_normal.ld(x, y)) * 2.0f - 1.0f;
                const float filter_radius_ss = 0.5f * fc.view_constants.view_to_clip[5] / -depth_to_view_z(fc, center_depth);
                const uint32_t filter_idx = uint32_t(clampf(filter_radius_ss * 7.0f, 0.0f, 7.0f));
                f3 vsum = mk3(0.0f);
                float wsum = 0.0f;
                const int sc = int(8.0f - center_sample_count / 2.0f);
                const uint32_t sample_count = uint32_t(std::min(std::max(sc, 2), 8));
                const int kernel_scale = center_sample_count < 4.0f ? 2 : 1;
                const uint32_t px_idx_in_quad = ((uint32_t(x & 1) | uint32_t(y & 1) * 2u) + fc.frame_index) & 3u;
                for (uint32_t sample_i = 0; sample_i < sample_count; ++sample_i) {
                    const int32_t* o = in.spatial_resolve_offsets + 4 * ((px_idx_in_quad * 16 + sample_i) + 64 * filter_idx);
                    const int sx = x + kernel_scale * o[0], sy = y + kernel_scale * o[1];
                    const f4 nraw = ld4(input_tex, sx, sy);
                    const f3 neigh = vsqrt(xyz(nraw));                 // linear_rgb_to_crunched_rgb
                    const float sample_depth = in.depth.ld(sx, sy);
                    const f3 sample_normal_vs = in.geometric_normal.inb(sx, sy) ? unpack_a2r10g10b10(in.geometric_normal.ld(sx, sy)) * 2.0f - 1.0f : mk3(-1.0f);
                    float w = 1.0f;
                    w *= exp2f(-50.0f * fabsf(center_normal_vs.z * (center_depth / sample_depth - 1.0f)));
                    const float dp = saturate(dot(center_normal_vs, sample_normal_vs));
                    w *= dp * dp * dp;
                    vsum += neigh * w;
                    wsum += w;
                }
                const f3 v = vsum / wsum;
                output_tex.st(x, y, pack_r11g11b10f(v * v));           // crunched_rgb_to_linear_rgb
            }
    }

    // ------------------------------------------------------------------ rtr.rs:97-400 (`trace`)
    struct Traced { Img<uint32_t> resolved_tex; ImgRGBA16F temporal_output_tex, history_tex; ImgRG16F ray_len_tex; ImgR8 refl_restir_invalidity_tex; };
    Traced traced;
    Traced trace(const FrameConstants& fc, RtrInputs in, uint32_t pass_mask = RTR_PASS_ALL) {
        resize(in.W, in.H);
        const bool advance = !(pass_mask & RTR_PASS_KEEP);
        sun_color = sun_color_in_direction(fc, sun_direction(fc));
        Img<uint32_t> rng_output_tex, rng_history_tex;
        pingpong<uint32_t>("rtr.rng", 6, hw, hh, rng_output_tex, rng_history_tex, advance);
        if (pass_mask & RTR_PASS_TRACE) pass_trace(fc, in, rng_output_tex);
        Img<f4> ray_orig_output_tex, ray_orig_history_tex;
        pingpong<f4>("rtr.ray_orig", 3, hw, hh, ray_orig_output_tex, ray_orig_history_tex, advance);
        in.half_view_normal = get<uint32_t>("half_view_normal_tex", hw, hh);   // GbufferDepth::half_view_normal / half_depth (renderers/mod.rs:44-71)
        in.half_depth = get<float>("half_depth_tex", hw, hh);
        {
            const i2 off = halfres_subsample_offset(fc);
            for (int y = 0; y < hh; ++y)
                for (int x = 0; x < hw; ++x) {
                    const int sx = x * 2 + off.x, sy = y * 2 + off.y;
                    const f3 normal_ws = unpack_normal_11_10_11_no_normalize(in.gbuffer.ld(sx, sy).y);
                    const f3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, mk4(normal_ws, 0))));
                    in.half_view_normal.st(x, y, pack_rgba8_snorm(mk4(normal_vs, 1.0f)));
                    in.half_depth.st(x, y, in.depth.ld(sx, sy));
                }
        }
        ImgR8 refl_restir_invalidity_tex = get<uint8_t>("refl_restir_invalidity_tex", hw, hh);
        ImgRGBA16F hit_normal_output_tex, hit_normal_history_tex, irradiance_output_tex, irradiance_history_tex, ray_output_tex, ray_history_tex;
        ImgU2 reservoir_output_tex, reservoir_history_tex;
        pingpong<h4>("rtr.hit_normal", 7, hw, hh, hit_normal_output_tex, hit_normal_history_tex, advance);
        pingpong<h4>("rtr.irradiance", 2, hw, hh, irradiance_output_tex, irradiance_history_tex, advance);
        pingpong<u2>("rtr.reservoir", 5, hw, hh, reservoir_output_tex, reservoir_history_tex, advance);
        pingpong<h4>("rtr.ray", 4, hw, hh, ray_output_tex, ray_history_tex, advance);
        if (pass_mask & RTR_PASS_VALIDATE) {
            memset(refl_restir_invalidity_tex.p, 0, size_t(hw) * hh);
            pass_validate(fc, in, ray_orig_history_tex, ray_history_tex, rng_history_tex, irradiance_history_tex, reservoir_history_tex, refl_restir_invalidity_tex);
        }
        if (pass_mask & RTR_PASS_RESTIR_TEMPORAL)
            pass_restir_temporal(fc, in, irradiance_history_tex, ray_orig_history_tex, ray_history_tex, rng_history_tex, reservoir_history_tex, hit_normal_history_tex,
                                 irradiance_output_tex, ray_orig_output_tex, ray_output_tex, rng_output_tex, hit_normal_output_tex, reservoir_output_tex);
        Img<uint32_t> resolved_tex = get<uint32_t>("resolved_tex", W, H);
        ImgRGBA16F temporal_output_tex, history_tex;
        pingpong<h4>("rtr.temporal", 0, W, H, temporal_output_tex, history_tex, advance);
        ImgRG16F ray_len_output_tex, ray_len_history_tex;
        pingpong<h2>("rtr.ray_len", 1, W, H, ray_len_output_tex, ray_len_history_tex, advance);
        if (pass_mask & RTR_PASS_RESOLVE)
            pass_resolve(fc, in, ray_len_history_tex, irradiance_output_tex, ray_output_tex, reservoir_output_tex, ray_orig_output_tex, hit_normal_output_tex, resolved_tex, ray_len_output_tex);
        traced = Traced{resolved_tex, temporal_output_tex, history_tex, ray_len_output_tex, refl_restir_invalidity_tex};
        return traced;
    }
    // rtr.rs:440-480 (`TracedRtr::filter_temporal`): returns resolved_tex (reused as the cleanup output)
    Img<uint32_t> filter_temporal(const FrameConstants& fc, RtrInputs in, uint32_t pass_mask = RTR_PASS_ALL) {
        in.half_view_normal = get<uint32_t>("half_view_normal_tex", hw, hh);
        in.half_depth = get<float>("half_depth_tex", hw, hh);
        if (pass_mask & RTR_PASS_TEMPORAL_FILTER)
            pass_temporal_filter(fc, in, traced.resolved_tex, traced.history_tex, traced.ray_len_tex, traced.refl_restir_invalidity_tex, traced.temporal_output_tex);
        if (pass_mask & RTR_PASS_CLEANUP) pass_cleanup(fc, in, traced.temporal_output_tex, traced.resolved_tex);
        return traced.resolved_tex;
    }
};

} // namespace okj
