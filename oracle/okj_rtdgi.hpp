// ORACLE (test infrastructure). RtdgiRenderer restated: host orchestration from
// crates/lib/kajiya/src/renderers/rtdgi.rs:143-554 and every shader it records
// (assets/shaders/rtdgi/*.hlsl). One function per reference pass.
#pragma once
#include <cstdlib>
#include "okj_passes.hpp"
#include "okj_reservoir.hpp"
#include "okj_ircache.hpp"
#include <functional>
#include <atomic>

namespace okj {

static const float SKY_DIST = 1e4f;                       // diffuse_trace_common.inc.hlsl:16
static const float RESTIR_TEMPORAL_M_CLAMP = 20.0f;       // rtdgi_restir_settings.hlsl:2
static const float RESTIR_RESERVOIR_W_CLAMP = 10.0f;      // :5
static const float SSGI_NEAR_FIELD_RADIUS = 80.0f;        // near_field_settings.hlsl:2
static const float ROUGHNESS_BIAS = 0.5f;                 // diffuse_trace_common.inc.hlsl:7

static inline bool is_rtdgi_validation_frame(const FrameConstants& fc) { return fc.frame_index % 3 == 0; } // settings:41-46
static inline bool is_rtdgi_tracing_frame(const FrameConstants& fc) { return !is_rtdgi_validation_frame(fc); }

// rtdgi_common.hlsl:12-39
struct TemporalReservoirOutput {
    float depth; f3 ray_hit_offset_ws; float luminance; f3 hit_normal_ws;
    static TemporalReservoirOutput from_raw(u4 raw) {
        f2 a = unpack_2x16f_uint(raw.y), b = unpack_2x16f_uint(raw.z);
        TemporalReservoirOutput r;
        r.depth = asfloat(raw.x);
        r.ray_hit_offset_ws = f3{a.x, a.y, b.x};
        r.luminance = b.y;
        r.hit_normal_ws = unpack_normal_11_10_11(raw.w);
        return r;
    }
    u4 as_raw() const {
        return u4{asuint(depth), pack_2x16f_uint(ray_hit_offset_ws.x, ray_hit_offset_ws.y),
                  pack_2x16f_uint(ray_hit_offset_ws.z, luminance), pack_normal_11_10_11(hit_normal_ws)};
    }
};

typedef Img<f4> ImgRGBA32F;

// IrcacheLookupParams::lookup hook; null => 0 (no ircache bound: BASELINE config 1)
// `key` = the lookup's canonical position in the frame (pass << 28 | half-res pixel index), used by the cache's deterministic mode
typedef std::function<f3(f3 query_from_ws, f3 pt_ws, f3 normal_ws, uint32_t rank, uint32_t& rng, uint32_t key)> IrcacheLookupFn;

struct RtdgiInputs {
    int W = 0, H = 0;
    ImgU32 geometric_normal; ImgU4 gbuffer; ImgR32F depth;
    ImgRGBA16S reprojection_map;
    const h4* sky_cube = nullptr; int sky_cube_width = 16;
    const Scene* scene = nullptr;
    ImgR8 ssao;
    const uint8_t* blue_noise = nullptr;
    const h4* brdf_fg_lut = nullptr;
    IrcacheLookupFn ircache_lookup;
};

struct Rtdgi {
    // --- state (names = reference temporal keys / variable names)
    std::map<std::string, std::vector<uint8_t>> surf;
    int W = 0, H = 0, hw = 0, hh = 0;
    uint32_t spatial_reuse_pass_count = 2;
    bool use_raytraced_reservoir_visibility = false;   // rtdgi.rs:25,43
    bool flip[9] = {false, false, false, false, false, false, false, false, false};
    bool temporal2_flip = false;
    std::atomic<uint64_t> rays_closest{0}, rays_any{0};

    template <typename T> Img<T> get(const std::string& name, int w, int h) {
        auto& v = surf[name];
        if (v.size() != size_t(w) * h * sizeof(T)) v.assign(size_t(w) * h * sizeof(T), 0);
        return Img<T>(v.data(), w, h);
    }
    void resize(int W_, int H_) {
        if (W == W_ && H == H_) return;
        W = W_; H = H_; hw = (W + 1) / 2; hh = (H + 1) / 2; // ImageDesc::half_res (image.rs:140-142)
        surf.clear();
    }
    // PingPongTemporalResource::get_output_and_history (renderers/mod.rs:85-102)
    template <typename T> void pingpong(const char* key, int idx, int w, int h, Img<T>& output, Img<T>& history) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx]) std::swap(a, b);
        output = get<T>(a, w, h);
        history = get<T>(b, w, h);
        flip[idx] = !flip[idx];
    }

    // ------------------------------------------------------------------ rtdgi.rs:143-170, fullres_reproject.hlsl:29-76
    ImgRGBA16F temporal_output_tex, reprojected_history_tex;
    static f4 cubic_hermite(f4 A, f4 B, f4 C, f4 D, float t) { // inc/curve.hlsl
        float t2 = t * t, t3 = t * t * t;
        f4 a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
        f4 b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
        f4 c = -A / 2.0f + C / 2.0f;
        f4 d = B;
        return a * t3 + b * t2 + c * t + d;
    }
    void reproject(const FrameConstants& fc, ImgRGBA16S reprojection_map, int W_, int H_) {
        (void)fc;
        resize(W_, H_);
        ImgRGBA16F history_tex;
        std::string a = "rtdgi.temporal2:0", b = "rtdgi.temporal2:1";
        if (temporal2_flip) std::swap(a, b);
        temporal_output_tex = get<h4>(a, W, H);
        history_tex = get<h4>(b, W, H);
        temporal2_flip = !temporal2_flip;
        reprojected_history_tex = get<h4>("reprojected_history_tex", W, H);
        const f4 ts = tex_size4(W, H);
        const ImgRGBA16F input_tex = history_tex;
        const ImgRGBA16F output_tex = reprojected_history_tex;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                f2 uv = get_uv(float(x), float(y), ts);
                f4 reproj = ld_reproj(reprojection_map, x, y);
                f2 prev_uv = uv + f2{reproj.x, reproj.y};
                uint32_t quad_valid = uint32_t(reproj.z * 15.0f + 0.5f);
                // GatherBlue(sampler_nnc, uv + 0.5*sign(prev_uv)*texel): validity of the 2x2 quad towards +x,+y
                f2 guv = uv + 0.5f * f2{float((prev_uv.x > 0) - (prev_uv.x < 0)), float((prev_uv.y > 0) - (prev_uv.y < 0))} * f2{ts.z, ts.w};
                bool all_neigh_valid = true;
                {
                    float gx = guv.x * float(W) - 0.5f, gy = guv.y * float(H) - 0.5f;
                    int ox = int(floorf(gx)), oy = int(floorf(gy));
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            int sx = std::min(std::max(ox + dx, 0), W - 1), sy = std::min(std::max(oy + dy, 0), H - 1);
                            uint32_t v = uint32_t(ld_reproj(reprojection_map, sx, sy).z * 15.0f + 0.5f);
                            all_neigh_valid = all_neigh_valid && (v == 15);
                        }
                }
                f4 history = mk4(0.0f);
                if (quad_valid == 0) {
                } else if (quad_valid == 15) {
                    if (all_neigh_valid) {
                        // image_sample_catmull_rom (inc/image.hlsl:40-80)
                        f2 pixel = prev_uv * f2{float(W), float(H)} + 0.5f;
                        f2 frc{frac(pixel.x), frac(pixel.y)};
                        int ipx = int(pixel.x) - 1, ipy = int(pixel.y) - 1;
                        f4 rows[4];
                        for (int j = 0; j < 4; ++j) {
                            f4 c0 = ld4(input_tex, ipx - 1, ipy - 1 + j), c1 = ld4(input_tex, ipx, ipy - 1 + j);
                            f4 c2 = ld4(input_tex, ipx + 1, ipy - 1 + j), c3 = ld4(input_tex, ipx + 2, ipy - 1 + j);
                            rows[j] = cubic_hermite(c0, c1, c2, c3, frc.x);
                        }
                        history = vmax(mk4(0.0f), cubic_hermite(rows[0], rows[1], rows[2], rows[3], frc.y));
                    } else {
                        history = sample_bilinear_clamp(input_tex, prev_uv);
                    }
                } else {
                    f4 qv{(quad_valid & 1) ? 1.0f : 0.0f, (quad_valid & 2) ? 1.0f : 0.0f, (quad_valid & 4) ? 1.0f : 0.0f, (quad_valid & 8) ? 1.0f : 0.0f};
                    const Bilinear bl = get_bilinear_filter(prev_uv, f2{float(W), float(H)});
                    int ox = int(bl.origin.x), oy = int(bl.origin.y);
                    f4 s00 = ld4(input_tex, ox, oy), s10 = ld4(input_tex, ox + 1, oy), s01 = ld4(input_tex, ox, oy + 1), s11 = ld4(input_tex, ox + 1, oy + 1);
                    f4 w{(1.0f - bl.weights.x) * (1.0f - bl.weights.y), bl.weights.x * (1.0f - bl.weights.y),
                         (1.0f - bl.weights.x) * bl.weights.y, bl.weights.x * bl.weights.y};
                    w = w * qv;
                    float wsum = dot(w, mk4(1.0f));
                    if (wsum > 1e-5f) {
                        f4 r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
                        history = r * (1.0f / wsum);
                    }
                }
                st4(output_tex, x, y, history);
            }
    }

    // ------------------------------------------------------------------ diffuse_trace_common.inc.hlsl:38-221
    struct TraceResult { f3 out_value; f3 hit_normal_ws; float hit_t; float pdf; bool is_hit; };
    f3 sun_color; // SUN_COLOR hoisted: depends only on frame constants (inc/sun.hlsl:21-33)

    bool dbg_enabled = false;
    uint32_t ircache_key_base = 0;     // 1 << 28 in the validate pass, 2 << 28 in the trace pass (kajiya_amd: kj_ircache.hpp IrcRequest::key)
    std::atomic<uint64_t> dbg[8] = {};
    TraceResult do_the_thing(const FrameConstants& fc, const RtdgiInputs& in, uint32_t px, uint32_t py, f3 normal_ws, uint32_t& rng, Ray outgoing_ray) {
        const f4 gbuffer_tex_size = tex_size4(W, H);
        f3 total_radiance = mk3(0.0f);
        f3 hit_normal_ws = -outgoing_ray.d;
        float hit_t = outgoing_ray.tmax;
        float pdf = fmaxf(0.0f, 1.0f / (dot(normal_ws, outgoing_ray.d) * 2 * M_PI_F));
        // ray cone only affects texture LOD (see okj_scene.hpp note)
        rays_closest.fetch_add(1, std::memory_order_relaxed);
        const RayCone ray_cone = pixel_ray_cone_from_image_height(fc, gbuffer_tex_size.y * 0.5f).propagate(0.03f, length(outgoing_ray.o - get_eye_position(fc)));   // :68-71
        const GbufferPathVertex primary_hit = gbuffer_raytrace(*in.scene, fc, outgoing_ray, 1, false, ray_cone);
        if (primary_hit.is_hit) {
            hit_t = primary_hit.ray_t;
            GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
            hit_normal_ws = gbuffer.normal;
            const f3 primary_hit_cs = position_world_to_sample(fc, primary_hit.position);
            const f2 primary_hit_uv = cs_to_uv(f2{primary_hit_cs.x, primary_hit_cs.y});
            const float primary_hit_screen_depth = sample_nearest_clamp(in.depth, primary_hit_uv);
            // OKJ_RTDGI_DEPTH_GATE: experiment knob of scripts/pt_deficit_attribution.py (default = the shader's 5e-3)
            static const float depth_gate = getenv("OKJ_RTDGI_DEPTH_GATE") ? float(atof(getenv("OKJ_RTDGI_DEPTH_GATE"))) : 5e-3f;
            bool is_on_screen = fabsf(primary_hit_cs.x) < 1.0f && fabsf(primary_hit_cs.y) < 1.0f &&
                                inverse_depth_relative_diff(primary_hit_cs.z, primary_hit_screen_depth) < depth_gate;
            f4 reprojected_radiance = mk4(0.0f);
            if (dbg_enabled) {
                dbg[0].fetch_add(1);
                const bool in_cs = fabsf(primary_hit_cs.x) < 1.0f && fabsf(primary_hit_cs.y) < 1.0f;
                if (in_cs) dbg[1].fetch_add(1);
                if (is_on_screen) dbg[2].fetch_add(1);
                if (in_cs && !is_on_screen && primary_hit_cs.z > primary_hit_screen_depth) dbg[4].fetch_add(1);  // hit is in front of the visible surface
            }
            if (is_on_screen) {
                reprojected_radiance = unpack_rgba16f(sample_nearest_clamp(reprojected_history_tex, primary_hit_uv)) * fc.pre_exposure_delta;
                is_on_screen = reprojected_radiance.w > 0;
                if (dbg_enabled && is_on_screen) {
                    dbg[3].fetch_add(1);
                    if (in.ircache_lookup) {   // experiment only: compare the two sources of bounce light at the same hit
                        uint32_t rng2 = rng;
                        const f3 gi = in.ircache_lookup(outgoing_ray.o, primary_hit.position, gbuffer.normal, 1, rng2, 0);
                        dbg[5].fetch_add(uint64_t(1e4f * sRGB_to_luminance(xyz(reprojected_radiance))));
                        dbg[6].fetch_add(uint64_t(1e4f * sRGB_to_luminance(gi)));
                    }
                }
            }
            gbuffer.roughness = lerp(gbuffer.roughness, 1.0f, ROUGHNESS_BIAS);
            const m33 tangent_to_world = build_orthonormal_basis(gbuffer.normal);
            const f3 wo = mul(-outgoing_ray.d, tangent_to_world);
            const LayeredBrdf brdf = LayeredBrdf::from_gbuffer_ndotv(in.brdf_fg_lut, gbuffer, wo.z);
            // Sun
            if (sun_color.x != 0 || sun_color.y != 0 || sun_color.z != 0) {
                f4 bn = blue_noise_for_pixel(in.blue_noise, px, py, rng);
                const f3 to_light_norm = sample_sun_direction(fc, f2{bn.x, bn.y}, false);
                rays_any.fetch_add(1, std::memory_order_relaxed);
                const bool is_shadowed = in.scene->trace_any(Ray{primary_hit.position, 1e-4f, to_light_norm, SKY_DIST});
                const f3 wi = mul(to_light_norm, tangent_to_world);
                const f3 brdf_value = brdf.evaluate(wo, wi) * fmaxf(0.0f, wi.z);
                const f3 light_radiance = is_shadowed ? mk3(0.0f) : sun_color;
                total_radiance += brdf_value * light_radiance;
            }
            total_radiance += gbuffer.emissive;
            if (is_on_screen) {
                total_radiance += xyz(reprojected_radiance) * gbuffer.albedo;
            } else {
                f2 urand;
                urand.x = uint_to_u01_float(hash1_mut(rng));
                urand.y = uint_to_u01_float(hash1_mut(rng));
                const auto& lights = in.scene->triangle_lights;
                for (uint32_t li = 0; li < fc.triangle_light_count && li < lights.size(); ++li) {
                    const KjTriangleLight& tl = lights[li];
                    f3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
                    LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
                    const f3 shadow_ray_origin = primary_hit.position;
                    const f3 to_light_ws = ls.pos - shadow_ray_origin;
                    const float dist_to_light2 = dot(to_light_ws, to_light_ws);
                    const f3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist_to_light2));
                    const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist_to_light2;
                    if (to_psa_metric > 0.0f) {
                        rays_any.fetch_add(1, std::memory_order_relaxed);
                        const bool is_shadowed = in.scene->trace_any(Ray{shadow_ray_origin, 1e-3f, to_light_norm_ws, sqrtf(dist_to_light2) - 2e-3f});
                        const f3 bounce_albedo = lerp(gbuffer.albedo, mk3(1.0f), 0.04f);
                        const f3 brdf_value = bounce_albedo * to_psa_metric / M_PI_F;
                        if (!is_shadowed) total_radiance += f3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf_value / ls.pdf;
                    }
                }
                if (in.ircache_lookup) {
                    const f3 gi = in.ircache_lookup(outgoing_ray.o, primary_hit.position, gbuffer.normal, 1, rng, ircache_key_base | (py * uint32_t(hw) + px));
                    total_radiance += gi * gbuffer.albedo;
                }
            }
        } else {
            total_radiance += xyz(sample_cube_rgba16f(in.sky_cube, in.sky_cube_width, outgoing_ray.d));
        }
        (void)gbuffer_tex_size;
        return TraceResult{total_radiance, hit_normal_ws, hit_t, pdf, primary_hit.is_hit};
    }

    // ------------------------------------------------------------------ diffuse_validate.rgen.hlsl:46-111
    void pass_validate(const FrameConstants& fc, const RtdgiInputs& in, ImgU32 half_view_normal_tex, ImgU2 reservoir_tex,
                       ImgRGBA16F reservoir_ray_history_tex, ImgRGBA16F irradiance_history_tex, ImgRGBA32F ray_orig_history_tex,
                       ImgR8 rt_history_invalidity_out_tex) {
        const i2 off = halfres_subsample_offset(fc);
        ircache_key_base = 1u << 28;
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                if (0.0f == in.depth.ld(x * 2 + off.x, y * 2 + off.y)) {
                    rt_history_invalidity_out_tex.st(x, y, to_unorm8(1.0f));
                    continue;
                }
                float invalidity = 0.0f;
                if (is_rtdgi_validation_frame(fc)) {
                    const f3 normal_vs = ld_nrm_snorm8(half_view_normal_tex, x, y);
                    const f3 normal_ws = direction_view_to_world(fc, normal_vs);
                    const f3 prev_ray_orig = xyz(ray_orig_history_tex.ld(x, y));
                    const f3 prev_hit_pos = xyz(ld4(reservoir_ray_history_tex, x, y)) + prev_ray_orig;
                    const f4 prev_radiance_packed = ld4(irradiance_history_tex, x, y);
                    const f3 prev_radiance = vmax(mk3(0.0f), xyz(prev_radiance_packed));
                    Ray prev_ray{prev_ray_orig, 0.0f, normalize(prev_hit_pos - prev_ray_orig), SKY_DIST};
                    uint32_t rng = hash3(uint32_t(x), uint32_t(y), 0);
                    TraceResult result = do_the_thing(fc, in, x, y, normal_ws, rng, prev_ray);
                    const f3 new_radiance = vmax(mk3(0.0f), result.out_value);
                    const f3 d = vabs(prev_radiance - new_radiance) / vmax(mk3(1e-3f), prev_radiance + new_radiance);
                    const float rad_diff = length(d);
                    invalidity = smoothstep(0.1f, 0.5f, rad_diff / length(mk3(1.0f)));
                    const float prev_hit_dist = length(prev_hit_pos - prev_ray_orig);
                    if (fabsf(result.hit_t - prev_hit_dist) / (prev_hit_dist + prev_hit_dist) < 0.2f) {
                        st4(irradiance_history_tex, x, y, mk4(new_radiance, prev_radiance_packed.w));
                        Reservoir1spp r = Reservoir1spp::from_raw(reservoir_tex.ld(x, y));
                        const float lum_old = sRGB_to_luminance(prev_radiance);
                        const float lum_new = sRGB_to_luminance(new_radiance);
                        r.M *= clampf(lum_old / fmaxf(1e-8f, lum_new), 0.03f, 1.0f);
                        const float allowed_luminance_increment = 10.0f;
                        r.W *= clampf(lum_old / fmaxf(1e-8f, lum_new) * allowed_luminance_increment, 0.01f, 1.0f);
                        reservoir_tex.st(x, y, r.as_raw());
                    }
                }
                rt_history_invalidity_out_tex.st(x, y, to_unorm8(invalidity));
            }
    }

    // ------------------------------------------------------------------ trace_diffuse.rgen.hlsl:49-120, candidate_ray_dir.hlsl:1-24
    void pass_trace(const FrameConstants& fc, const RtdgiInputs& in, ImgU32 half_view_normal_tex,
                    ImgRGBA16F candidate_irradiance_out_tex, ImgU32 candidate_normal_out_tex, ImgRGBA16F candidate_hit_out_tex,
                    ImgR8 rt_history_invalidity_in_tex, ImgR8 rt_history_invalidity_out_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
        ircache_key_base = 2u << 28;
#pragma omp parallel for schedule(dynamic, 2)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                float depth = in.depth.ld(hx, hy);
                if (0.0f == depth) {
                    st4(candidate_irradiance_out_tex, x, y, mk4(0.0f));
                    candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(f4{0, 0, 1, 0}));
                    rt_history_invalidity_out_tex.st(x, y, 0);
                    continue;
                }
                const f2 uv = get_uv(float(hx), float(hy), gbuffer_tex_size);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                const float NEAR_FIELD_FADE_OUT_END = -vrc.ray_hit_vs().z * (SSGI_NEAR_FIELD_RADIUS * gbuffer_tex_size.w * 0.5f);
                {
                    const f3 normal_vs = ld_nrm_snorm8(half_view_normal_tex, x, y);
                    const f3 normal_ws = direction_view_to_world(fc, normal_vs);
                    const m33 tangent_to_world = build_orthonormal_basis(normal_ws);
                    f4 bn = blue_noise_for_pixel(in.blue_noise, x, y, fc.frame_index);
                    const f3 outgoing_dir = mul(tangent_to_world, uniform_sample_hemisphere(f2{bn.x, bn.y}));
                    Ray outgoing_ray;
                    outgoing_ray.d = outgoing_dir;
                    outgoing_ray.o = vrc.biased_secondary_ray_origin_ws_with_normal(normal_ws);
                    outgoing_ray.tmin = 0;
                    outgoing_ray.tmax = is_rtdgi_tracing_frame(fc) ? SKY_DIST : NEAR_FIELD_FADE_OUT_END;
                    uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index & 31);
                    TraceResult result = do_the_thing(fc, in, x, y, normal_ws, rng, outgoing_ray);
                    if (!is_rtdgi_tracing_frame(fc) && !result.is_hit) {
                        result.out_value = mk3(0.0f);
                        result.hit_t = SKY_DIST;
                    }
                    const f3 hit_offset_ws = outgoing_ray.d * result.hit_t;
                    const float cos_theta = dot(normalize(outgoing_dir - vrc.ray_dir_ws()), normal_ws);
                    st4(candidate_irradiance_out_tex, x, y, mk4(result.out_value, 1.0f - cos_theta)); // rtr_encode_cos_theta_for_fp16
                    st4(candidate_hit_out_tex, x, y, mk4(hit_offset_ws, result.pdf * (is_rtdgi_tracing_frame(fc) ? 1.0f : -1.0f)));
                    candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(mk4(direction_world_to_view(fc, result.hit_normal_ws), 0)));
                }
                const f4 reproj = ld_reproj(in.reprojection_map, hx, hy);
                const int rx = int(floorf(float(x) + gbuffer_tex_size.x * reproj.x / 2 + 0.5f));
                const int ry = int(floorf(float(y) + gbuffer_tex_size.y * reproj.y / 2 + 0.5f));
                rt_history_invalidity_out_tex.st(x, y, rt_history_invalidity_in_tex.ld(rx, ry));
            }
    }

    // ------------------------------------------------------------------ temporal_validity_integrate.hlsl:21-119
    // Wave intrinsics: [numthreads(8,8,1)] => lane = x%8 + 8*(y%8); lane^2 is x^2, lane^16 is y^2,
    // lane^1 is x^1, lane^8 is y^1 inside the 8x8 group. Lanes outside the image
    // still execute (loads return 0), so partial tiles at the edge are evaluated too.
    void pass_validity_integrate(const FrameConstants& fc, const RtdgiInputs& in, ImgR8 input_tex, ImgRG16F history_tex,
                                 ImgR32F half_depth_tex, ImgRG16F output_tex) {
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const int tiles_x = (hw + 7) / 8, tiles_y = (hh + 7) / 8;
#pragma omp parallel for schedule(static)
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                float blurred[64], edge_v[64];
                for (int l = 0; l < 64; ++l) {
                    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                    f2 invalid_blurred{0, 0};
                    for (int dy = -2; dy <= 2; ++dy)
                        for (int dx = -2; dx <= 2; ++dx) {
                            float w = exp2f(-0.1f * float(dx * dx + dy * dy));
                            invalid_blurred += f2{from_unorm8(input_tex.ld(x + dx, y + dy)), 1.0f} * w;
                        }
                    invalid_blurred = invalid_blurred / invalid_blurred.y;
                    blurred[l] = invalid_blurred.x;
                    const float center_depth = half_depth_tex.ld(x, y);
                    float edge = 1;
                    bool brk = false;
                    for (int oy = 0; oy <= 2 && !brk; ++oy) {
                        for (int ox = 1; ox <= 2; ++ox) {
                            const f4 reproj = ld_reproj(in.reprojection_map, x * 2 + ox, y * 2 + oy);
                            const float sample_depth = half_depth_tex.ld(x + ox / 2, y + oy / 2);
                            if (reproj.w < 0 || inverse_depth_relative_diff(center_depth, sample_depth) > 0.1f) {
                                edge = 0;
                                break; // HLSL `break` leaves the inner loop only
                            }
                            edge *= (reproj.z == 0 && sample_depth != 0) ? 1.0f : 0.0f;
                        }
                    }
                    edge_v[l] = edge;
                }
                float b1[64];
                for (int l = 0; l < 64; ++l) b1[l] = lerp(blurred[l], blurred[l ^ 2], 0.5f);
                for (int l = 0; l < 64; ++l) blurred[l] = lerp(b1[l], b1[l ^ 16], 0.5f);
                float e1[64];
                for (int l = 0; l < 64; ++l) e1[l] = fmaxf(edge_v[l], edge_v[l ^ 1]);
                for (int l = 0; l < 64; ++l) edge_v[l] = fmaxf(e1[l], e1[l ^ 8]);
                for (int l = 0; l < 64; ++l) {
                    const int x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                    float ib = smoothstep(0.0f, 1.0f, blurred[l]);
                    ib += edge_v[l];
                    ib = saturate(ib);
                    const f4 reproj = ld_reproj(in.reprojection_map, x * 2, y * 2);
                    const f2 reproj_px{float(x) + gbuffer_tex_size.x * reproj.x / 2 + 0.5f, float(y) + gbuffer_tex_size.y * reproj.y / 2 + 0.5f};
                    float history = 0;
                    const int sample_count = 8;
                    float ang_off = uint_to_u01_float(hash3(uint32_t(x), uint32_t(y), fc.frame_index)) * M_PI_F * 2;
                    for (uint32_t si = 0; si < uint32_t(sample_count); ++si) {
                        float ang = (float(si) + ang_off) * GOLDEN_ANGLE;
                        float radius = float(si) * 1.0f;
                        f2 so = cos_sin_turns(ang) * radius;
                        const int sx = int(reproj_px.x + so.x), sy = int(reproj_px.y + so.y);
                        history += ld2(history_tex, sx, sy).x;
                    }
                    history /= sample_count;
                    st2(output_tex, x, y, f2{fmaxf(history * 0.75f, ib), from_unorm8(input_tex.ld(x, y))});
                }
            }
    }

    // ------------------------------------------------------------------ restir_temporal.hlsl:83-422
    static i2 get_rpx_offset(uint32_t sample_i, uint32_t frame_index) {
        const i2 offsets[4] = {i2{-1, -1}, i2{1, 1}, i2{-1, 1}, i2{1, -1}};
        const i2 base = offsets[frame_index & 3] + offsets[(sample_i + (frame_index ^ 1)) & 3];
        return sample_i == 0 ? i2{0, 0} : base;
    }
    void pass_restir_temporal(const FrameConstants& fc, const RtdgiInputs& in, ImgU32 half_view_normal_tex,
                              ImgRGBA16F candidate_radiance_tex, ImgU32 candidate_normal_tex, ImgRGBA16F candidate_hit_tex,
                              ImgRGBA16F radiance_history_tex, ImgRGBA32F ray_orig_history_tex, ImgRGBA16F ray_history_tex,
                              ImgU2 reservoir_history_tex, ImgRGBA16F hit_normal_history_tex, ImgRGBA16F candidate_history_tex,
                              ImgRG16F rt_invalidity_tex, ImgRGBA16F radiance_out_tex, ImgRGBA32F ray_orig_output_tex,
                              ImgRGBA16F ray_output_tex, ImgRGBA16F hit_normal_output_tex, ImgU2 reservoir_out_tex,
                              ImgRGBA16F candidate_out_tex, ImgU4 temporal_reservoir_packed_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const ImgR32F depth_tex = in.depth;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                float depth = depth_tex.ld(hx, hy);
                if (0.0f == depth) {
                    st4(radiance_out_tex, x, y, f4{0, 0, 0, -SKY_DIST});
                    st4(hit_normal_output_tex, x, y, mk4(0.0f));
                    reservoir_out_tex.st(x, y, u2{0, 0});
                    continue;
                }
                const f2 uv = get_uv(float(hx), float(hy), gbuffer_tex_size);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                const f3 normal_vs = ld_nrm_snorm8(half_view_normal_tex, x, y);
                const f3 normal_ws = direction_view_to_world(fc, normal_vs);
                const f3 refl_ray_origin_ws = vrc.biased_secondary_ray_origin_ws_with_normal(normal_ws);
                const f3 hit_offset_ws = xyz(ld4(candidate_hit_tex, x, y));
                f3 outgoing_dir = normalize(hit_offset_ws);
                uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index);
                f3 radiance_sel = mk3(0.0f), ray_orig_sel_ws = mk3(0.0f), ray_hit_sel_ws = mk3(1.0f), hit_normal_sel = mk3(1.0f);
                StreamState stream_state;
                Reservoir1spp reservoir;
                const uint32_t reservoir_payload = uint32_t(x) | (uint32_t(y) << 16);
                if (is_rtdgi_tracing_frame(fc)) {
                    const float hit_t = length(hit_offset_ws);
                    const f3 out_value = xyz(ld4(candidate_radiance_tex, x, y));
                    const f3 result_hit_normal_ws = direction_view_to_world(fc, ld_nrm_snorm8(candidate_normal_tex, x, y));
                    const float p_q = 1.0f * fmaxf(0.0f, sRGB_to_luminance(out_value)) * step(0.0f, dot(outgoing_dir, normal_ws));
                    const float inv_pdf_q = 1.0f;
                    radiance_sel = out_value;
                    ray_orig_sel_ws = refl_ray_origin_ws;
                    ray_hit_sel_ws = refl_ray_origin_ws + outgoing_dir * hit_t;
                    hit_normal_sel = result_hit_normal_ws;
                    reservoir.init_with_stream(p_q, inv_pdf_q, stream_state, reservoir_payload);
                    float rl = lerp(ld4(candidate_history_tex, x, y).y, sqrtf(hit_t), 0.05f);
                    st4(candidate_out_tex, x, y, f4{sqrtf(hit_t), rl, 0, 0});
                }
                const float rt_invalidity = sqrtf(saturate(ld2(rt_invalidity_tex, x, y).y));
                float center_M = 0;
                for (uint32_t sample_i = 0; sample_i < 5 && stream_state.M_sum < 1.25f * RESTIR_TEMPORAL_M_CLAMP; ++sample_i) {
                    const i2 rpx_offset = get_rpx_offset(sample_i, fc.frame_index);
                    if (sample_i > 0 && rpx_offset.x == 0 && rpx_offset.y == 0) continue;
                    const f4 reproj = ld_reproj(in.reprojection_map, hx + rpx_offset.x * 2, hy + rpx_offset.y * 2);
                    const uint32_t xor_seq[4][2] = {{3, 3}, {2, 1}, {1, 2}, {3, 3}};
                    const uint32_t* pxv = xor_seq[fc.frame_index & 3];
                    // ((px + rpx_offset) ^ xor): int2 + int2 -> int, ^ uint -> uint, then to float
                    f2 base = sample_i == 0 ? f2{float(x), float(y)}
                                            : f2{float(uint32_t(x + rpx_offset.x) ^ pxv[0]), float(uint32_t(y + rpx_offset.y) ^ pxv[1])};
                    const int prx = f2i_sat(floorf(base.x + gbuffer_tex_size.x * reproj.x * 0.5f + 0.0f + 0.5f));
                    const int pry = f2i_sat(floorf(base.y + gbuffer_tex_size.y * reproj.y * 0.5f + 0.0f + 0.5f));
                    const i2 rpx{wrap_add(prx, rpx_offset.x), wrap_add(pry, rpx_offset.y)};
                    const int pnx = f2i_sat(floorf(base.x + 0.5f)), pny = f2i_sat(floorf(base.y + 0.5f));
                    const i2 neighbor_px{wrap_add(pnx, rpx_offset.x), wrap_add(pny, rpx_offset.y)};
                    const int nhx = wrap_mul2_add(neighbor_px.x, off.x), nhy = wrap_mul2_add(neighbor_px.y, off.y);
                    Reservoir1spp r = Reservoir1spp::from_raw(reservoir_history_tex.ld(rpx.x, rpx.y));
                    const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
                    float visibility = 1, relevance = 1;
                    const float sample_depth = depth_tex.ld(nhx, nhy);
                    const f3 prev_ray_orig = xyz(ray_orig_history_tex.ld(spx_x, spx_y));
                    if (length(prev_ray_orig - refl_ray_origin_ws) > 0.1f * -vrc.ray_hit_vs().z) continue;
                    if (0 == sample_depth) continue;
                    if (reproj.z == 0) continue;
                    relevance *= 1 - smoothstep(0.0f, 0.1f, inverse_depth_relative_diff(depth, sample_depth));
                    const f3 sample_normal_vs = ld_nrm_snorm8(half_view_normal_tex, neighbor_px.x, neighbor_px.y);
                    const float normal_similarity_dot = fmaxf(0.0f, dot(sample_normal_vs, normal_vs));
                    const float normal_cutoff = 0.2f;
                    if (sample_i != 0 && normal_similarity_dot < normal_cutoff) continue;
                    relevance *= powf(normal_similarity_dot, 4.0f);
                    const f4 rh = ld4(ray_history_tex, spx_x, spx_y);
                    const f3 sample_hit_ws = xyz(rh) + prev_ray_orig;
                    const float prev_dist = rh.w;
                    const f4 hn = ld4(hit_normal_history_tex, spx_x, spx_y);
                    const f4 sample_hit_normal_ws_dot{hn.x * 2 - 1, hn.y * 2 - 1, hn.z * 2 - 1, hn.w};
                    const f3 dir_to_sample_hit_unnorm = sample_hit_ws - refl_ray_origin_ws;
                    const float dist_to_sample_hit = length(dir_to_sample_hit_unnorm);
                    const f3 dir_to_sample_hit = normalize(dir_to_sample_hit_unnorm);
                    const float center_to_hit_vis = -dot(xyz(sample_hit_normal_ws_dot), dir_to_sample_hit);
                    const f4 prev_rad = ld4(radiance_history_tex, spx_x, spx_y) * f4{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
                    r.M = fmaxf(0.0f, fminf(r.M, exp2f(log2f(RESTIR_TEMPORAL_M_CLAMP) * (1.0f - rt_invalidity))));
                    const float p_q = 1 * fmaxf(0.0f, sRGB_to_luminance(xyz(prev_rad))) * step(0.0f, dot(dir_to_sample_hit, normal_ws));
                    float jacobian = 1;
                    jacobian *= clampf(prev_dist / dist_to_sample_hit, 1e-4f, 1e4f);
                    jacobian *= jacobian;
                    jacobian *= clampf(center_to_hit_vis / sample_hit_normal_ws_dot.w, 0.0f, 1e4f);
                    r.M *= relevance;
                    if (0 == sample_i) center_M = r.M;
                    if (reservoir.update_with_stream(r, p_q, jacobian * visibility, stream_state, reservoir_payload, rng)) {
                        outgoing_dir = dir_to_sample_hit;
                        radiance_sel = xyz(prev_rad);
                        ray_orig_sel_ws = prev_ray_orig;
                        ray_hit_sel_ws = sample_hit_ws;
                        hit_normal_sel = xyz(sample_hit_normal_ws_dot);
                    }
                }
                reservoir.finish_stream(stream_state);
                reservoir.W = fminf(reservoir.W, RESTIR_RESERVOIR_W_CLAMP);
                reservoir.M = center_M + 0.5f;
                const f4 hit_normal_ws_dot = mk4(hit_normal_sel, -dot(hit_normal_sel, outgoing_dir));
                st4(radiance_out_tex, x, y, mk4(radiance_sel, dot(normal_ws, outgoing_dir)));
                ray_orig_output_tex.st(x, y, mk4(ray_orig_sel_ws, 0.0f));
                st4(hit_normal_output_tex, x, y, f4{hit_normal_ws_dot.x * 0.5f + 0.5f, hit_normal_ws_dot.y * 0.5f + 0.5f, hit_normal_ws_dot.z * 0.5f + 0.5f, hit_normal_ws_dot.w});
                st4(ray_output_tex, x, y, mk4(ray_hit_sel_ws - ray_orig_sel_ws, length(ray_hit_sel_ws - refl_ray_origin_ws)));
                reservoir_out_tex.st(x, y, reservoir.as_raw());
                TemporalReservoirOutput rp;
                rp.depth = depth;
                rp.ray_hit_offset_ws = ray_hit_sel_ws - vrc.ray_hit_ws();
                rp.luminance = fmaxf(0.0f, sRGB_to_luminance(radiance_sel));
                rp.hit_normal_ws = xyz(hit_normal_ws_dot);
                temporal_reservoir_packed_tex.st(x, y, rp.as_raw());
            }
    }

    // ------------------------------------------------------------------ occlusion_raymarch.hlsl:75-146 (half-res depth, no colour bounce)
    void occlusion_raymarch(const FrameConstants& fc, f2 raymarch_start_uv, f3 raymarch_start_cs, f3 raymarch_end_ws, uint32_t max_sample_count,
                            ImgR32F halfres_depth_tex, float& visibility) const {
        const f2 fullres{float(W), float(H)}, halfres{float(hw), float(hh)};
        const i2 off = halfres_subsample_offset(fc);
        const f3 raymarch_end_cs = position_world_to_clip(fc, raymarch_end_ws);
        const f2 raymarch_end_uv = cs_to_uv(f2{raymarch_end_cs.x, raymarch_end_cs.y});
        const f2 raymarch_uv_delta = raymarch_end_uv - raymarch_start_uv;
        const f2 raymarch_len_px = raymarch_uv_delta * halfres;
        const uint32_t MIN_PX_PER_STEP = 2;
        const int k_count = std::min(int(max_sample_count), int(floorf(length(raymarch_len_px) / float(MIN_PX_PER_STEP))));
        const float Z_LAYER_THICKNESS = 0.05f;
        const float depth_step_per_z = (raymarch_end_cs.z - raymarch_start_cs.z) / length(f2{raymarch_end_cs.x, raymarch_end_cs.y} - f2{raymarch_start_cs.x, raymarch_start_cs.y});
        float t_step = 1.0f / float(k_count);
        float t = 0.5f * t_step;
        for (int k = 0; k < k_count; ++k) {
            const f3 interp_pos_cs = lerp(raymarch_start_cs, raymarch_end_cs, t);
            const f2 uv_at_interp = cs_to_uv(f2{interp_pos_cs.x, interp_pos_cs.y});
            // uint2(floor(uv*size - offset)) & ~1u) + offset ; float->uint of negatives saturates to 0 on AMD
            f2 fp{floorf(uv_at_interp.x * fullres.x - float(off.x)), floorf(uv_at_interp.y * fullres.y - float(off.y))};
            uint32_t ux = fp.x > 0 ? uint32_t(fp.x) : 0u, uy = fp.y > 0 ? uint32_t(fp.y) : 0u;
            uint32_t pxi = (ux & ~1u) + uint32_t(off.x), pyi = (uy & ~1u) + uint32_t(off.y);
            const float depth_at_interp = halfres_depth_tex.ld(int(pxi >> 1u), int(pyi >> 1u));
            const f2 quantized_cs_at_interp = uv_to_cs(f2{(float(pxi) + 0.5f) / fullres.x, (float(pyi) + 0.5f) / fullres.y});
            const float biased_interp_z = raymarch_start_cs.z + depth_step_per_z * length(quantized_cs_at_interp - f2{raymarch_start_cs.x, raymarch_start_cs.y});
            if (depth_at_interp > biased_interp_z) {
                const float depth_diff = inverse_depth_relative_diff(interp_pos_cs.z, depth_at_interp);
                float hit = smoothstep(Z_LAYER_THICKNESS, Z_LAYER_THICKNESS * 0.5f, depth_diff);
                visibility *= 1 - hit;
            }
            t += t_step;
        }
    }

    // ------------------------------------------------------------------ restir_spatial.hlsl:48-372
    static float normal_inluence_nonlinearity(float x, float b) { return x < -b ? 0.0f : (x + b) * (x + b) / (4 * b); }
    void pass_restir_spatial(const FrameConstants& fc, const RtdgiInputs& in, ImgU2 reservoir_input_tex, ImgU32 half_view_normal_tex,
                             ImgR32F half_depth_tex, ImgR8S half_ssao_tex, ImgU4 temporal_reservoir_packed_tex,
                             ImgU2 reservoir_output_tex, uint32_t spatial_reuse_pass_idx, uint32_t perform_occlusion_raymarch,
                             uint32_t occlusion_raymarch_importance_only) {
        (void)in;
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const f4 output_tex_size = tex_size4(hw, hh);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                float depth = half_depth_tex.ld(x, y);
                const uint32_t seed = fc.frame_index + spatial_reuse_pass_idx * 123;
                uint32_t rng = hash3(uint32_t(x), uint32_t(y), seed);
                const f2 uv = get_uv(float(hx), float(hy), gbuffer_tex_size);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, depth);
                const f3 center_normal_vs = ld_nrm_snorm8(half_view_normal_tex, x, y);
                const f3 center_normal_ws = direction_view_to_world(fc, center_normal_vs);
                const float center_depth = half_depth_tex.ld(x, y);
                const float center_ssao = from_snorm8(half_ssao_tex.ld(x, y));
                StreamState stream_state;
                Reservoir1spp reservoir;
                float sample_radius_offset = uint_to_u01_float(hash1_mut(rng));
                Reservoir1spp center_r = Reservoir1spp::from_raw(reservoir_input_tex.ld(x, y));
                float kernel_tightness = 1.0f - center_ssao;
                const uint32_t SAMPLE_COUNT_PASS0 = 8, SAMPLE_COUNT_PASS1 = 5;
                const float MAX_INPUT_M_IN_PASS0 = RESTIR_TEMPORAL_M_CLAMP;
                const float MAX_INPUT_M_IN_PASS1 = MAX_INPUT_M_IN_PASS0 * SAMPLE_COUNT_PASS0;
                const float MAX_INPUT_M_IN_PASS = spatial_reuse_pass_idx == 0 ? MAX_INPUT_M_IN_PASS0 : MAX_INPUT_M_IN_PASS1;
                kernel_tightness = lerp(kernel_tightness, 1.0f, 0.5f * smoothstep(MAX_INPUT_M_IN_PASS * 0.5f, MAX_INPUT_M_IN_PASS, center_r.M));
                float max_kernel_radius = spatial_reuse_pass_idx == 0 ? lerp(32.0f, 12.0f, kernel_tightness) : lerp(16.0f, 6.0f, kernel_tightness);
                if (spatial_reuse_pass_idx >= 2) max_kernel_radius = 8;
                const f2 dist_to_edge_xy = vmin(f2{float(x), float(y)}, f2{output_tex_size.x - float(x), output_tex_size.y - float(y)});
                const float allow_edge_overstep = center_r.M < 10 ? 100.0f : 1.25f;
                const f2 kernel_radius = vmin(f2{max_kernel_radius, max_kernel_radius}, dist_to_edge_xy * allow_edge_overstep);
                uint32_t sample_count = spatial_reuse_pass_idx == 0 ? SAMPLE_COUNT_PASS0 : SAMPLE_COUNT_PASS1;
                const uint32_t sx_seed = spatial_reuse_pass_idx == 0 ? (uint32_t(x) >> 3) : (uint32_t(x) >> 2);
                const uint32_t sy_seed = spatial_reuse_pass_idx == 0 ? (uint32_t(y) >> 3) : (uint32_t(y) >> 2);
                float ang_offset = uint_to_u01_float(hash3(sx_seed, sy_seed, fc.frame_index * 2 + spatial_reuse_pass_idx)) * M_PI_F * 2;
                for (uint32_t sample_i = 0; sample_i < sample_count; ++sample_i) {
                    float ang = (float(sample_i) + ang_offset) * GOLDEN_ANGLE;
                    f2 radius = 0 == sample_i ? f2{0, 0} : (powf((float(sample_i) + sample_radius_offset) / float(sample_count), 0.5f) * kernel_radius);
                    const f2 cs_ang = cos_sin_turns(ang);
                    const i2 rpx_offset{int(cs_ang.x * radius.x), int(cs_ang.y * radius.y)};
                    const bool is_center_sample = sample_i == 0;
                    const i2 rpx{x + rpx_offset.x, y + rpx_offset.y};
                    const u2 reservoir_raw = reservoir_input_tex.ld(rpx.x, rpx.y);
                    if (0 == reservoir_raw.x) continue;
                    Reservoir1spp r = Reservoir1spp::from_raw(reservoir_raw);
                    r.M = fminf(r.M, 500.0f);
                    const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
                    const TemporalReservoirOutput spx_packed = TemporalReservoirOutput::from_raw(temporal_reservoir_packed_tex.ld(spx_x, spx_y));
                    const float reused_luminance = spx_packed.luminance;
                    float visibility = 1, relevance = 1;
                    const f3 sample_normal_vs = ld_nrm_snorm8(half_view_normal_tex, rpx.x, rpx.y);
                    const float normal_similarity_dot = dot(sample_normal_vs, center_normal_vs);
                    relevance *= normal_inluence_nonlinearity(normal_similarity_dot, 0.5f) / normal_inluence_nonlinearity(1.0f, 0.5f);
                    const float sample_ssao = from_snorm8(half_ssao_tex.ld(rpx.x, rpx.y));
                    relevance *= 1 - fabsf(sample_ssao - center_ssao);
                    const f2 rpx_uv = get_uv(float(rpx.x * 2 + off.x), float(rpx.y * 2 + off.y), gbuffer_tex_size);
                    const float rpx_depth = half_depth_tex.ld(rpx.x, rpx.y);
                    if (rpx_depth == 0.0f) continue;
                    const ViewRayContext rpx_ray_ctx = ViewRayContext::from_uv_and_depth(fc, rpx_uv, rpx_depth);
                    const f2 spx_uv = get_uv(float(spx_x * 2 + off.x), float(spx_y * 2 + off.y), gbuffer_tex_size);
                    const ViewRayContext spx_ray_ctx = ViewRayContext::from_uv_and_depth(fc, spx_uv, spx_packed.depth);
                    const f3 sample_hit_ws = spx_packed.ray_hit_offset_ws + spx_ray_ctx.ray_hit_ws();
                    const f3 reused_dir_to_sample_hit_unnorm_ws = sample_hit_ws - rpx_ray_ctx.ray_hit_ws();
                    const float reused_dist = length(reused_dir_to_sample_hit_unnorm_ws);
                    const f3 reused_dir_to_sample_hit_ws = reused_dir_to_sample_hit_unnorm_ws / reused_dist;
                    const f3 dir_to_sample_hit_unnorm = sample_hit_ws - vrc.ray_hit_ws();
                    const float dist_to_sample_hit = length(dir_to_sample_hit_unnorm);
                    const f3 dir_to_sample_hit = normalize(dir_to_sample_hit_unnorm);
                    if (!is_center_sample) {
                        const float depth_diff = fabsf(fmaxf(0.3f, center_normal_vs.z) * (center_depth / rpx_depth - 1.0f));
                        const float depth_threshold = spatial_reuse_pass_idx == 0 ? 0.15f : 0.1f;
                        relevance *= 1 - smoothstep(0.0f, depth_threshold, depth_diff);
                    }
                    if (perform_occlusion_raymarch) {
                        const f2 ray_orig_uv = spx_uv;
                        const float surface_offset_len = length(ViewRayContext::from_uv_and_depth(fc, ray_orig_uv, depth).ray_hit_vs() - vrc.ray_hit_vs());
                        const float MAX_RAYMARCH_DIST_MULT = 3.0f;
                        const f3 raymarch_dir_unnorm_ws = sample_hit_ws - vrc.ray_hit_ws();
                        const f3 raymarch_end_ws = vrc.ray_hit_ws() + raymarch_dir_unnorm_ws * fminf(1.0f, MAX_RAYMARCH_DIST_MULT * surface_offset_len / length(raymarch_dir_unnorm_ws));
                        occlusion_raymarch(fc, uv, xyz(vrc.ray_hit_cs), raymarch_end_ws, 6, half_depth_tex, visibility);
                    }
                    const f3 sample_hit_normal_ws = spx_packed.hit_normal_ws;
                    const float center_to_hit_vis = -dot(sample_hit_normal_ws, dir_to_sample_hit);
                    const float reused_to_hit_vis = -dot(sample_hit_normal_ws, reused_dir_to_sample_hit_ws);
                    float p_q = 1;
                    p_q *= reused_luminance;
                    p_q *= fmaxf(0.0f, dot(dir_to_sample_hit, center_normal_ws));
                    float jacobian = 1;
                    jacobian *= reused_dist / dist_to_sample_hit;
                    jacobian *= jacobian;
                    jacobian *= clampf(center_to_hit_vis / reused_to_hit_vis, 0.0f, 1e4f);
                    jacobian = sqrtf(jacobian);
                    if (is_center_sample) jacobian = 1;
                    if (!(p_q >= 0)) continue;
                    r.M *= relevance;
                    if (occlusion_raymarch_importance_only) {
                        p_q *= lerp(0.25f, 1.0f, visibility);
                        visibility = 1;
                    }
                    reservoir.update_with_stream(r, p_q, visibility * jacobian, stream_state, r.payload, rng);
                }
                reservoir.finish_stream(stream_state);
                reservoir.W = fminf(reservoir.W, RESTIR_RESERVOIR_W_CLAMP);
                reservoir_output_tex.st(x, y, reservoir.as_raw());
            }
    }

    // ------------------------------------------------------------------ restir_check.rgen.hlsl:21-70 (use_raytraced_reservoir_visibility)
    void pass_restir_check(const FrameConstants& fc, const RtdgiInputs& in, ImgR32F half_depth_tex, ImgU4 temporal_reservoir_packed_tex, ImgU2 reservoir_input_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < hw; ++x) {
                const float depth = half_depth_tex.ld(x, y);
                const f2 uv = get_uv(float(x * 2 + off.x), float(y * 2 + off.y), gbuffer_tex_size);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_biased_depth(fc, uv, depth);
                Reservoir1spp r = Reservoir1spp::from_raw(reservoir_input_tex.ld(x, y));
                const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
                const TemporalReservoirOutput spx_packed = TemporalReservoirOutput::from_raw(temporal_reservoir_packed_tex.ld(spx_x, spx_y));
                const f2 spx_uv = get_uv(float(spx_x * 2 + off.x), float(spx_y * 2 + off.y), gbuffer_tex_size);
                const ViewRayContext spx_ctx = ViewRayContext::from_uv_and_depth(fc, spx_uv, spx_packed.depth);
                const f3 spx_pos_ws = spx_ctx.ray_hit_ws();
                const f3 hit_ws = spx_packed.ray_hit_offset_ws + spx_pos_ws;
                const f3 trace_origin_ws = vrc.biased_secondary_ray_origin_ws();
                const f3 trace_vec = hit_ws - trace_origin_ws;
                rays_any.fetch_add(1, std::memory_order_relaxed);
                if (in.scene->trace_any(Ray{trace_origin_ws, 0.0f, normalize(trace_vec), fminf(5.0f * length(spx_pos_ws - trace_origin_ws), length(trace_vec) * 0.999f)})) {
                    r.W = 0;
                    reservoir_input_tex.st(x, y, r.as_raw());
                }
            }
    }

    // ------------------------------------------------------------------ restir_resolve.hlsl:42-205
    static float ggx_ndf_unnorm(float a2, float cos_theta) {
        float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
        return a2 / (d * d);
    }
    void pass_restir_resolve(const FrameConstants& fc, const RtdgiInputs& in, ImgRGBA16F radiance_tex, ImgU2 reservoir_input_tex,
                             ImgU32 half_view_normal_tex, ImgR32F half_depth_tex, ImgRGBA16F candidate_radiance_tex,
                             ImgRGBA16F candidate_hit_tex, ImgU4 temporal_reservoir_packed_tex, ImgRGBA16F irradiance_output_tex) {
        const i2 off = halfres_subsample_offset(fc);
        const f4 gbuffer_tex_size = tex_size4(W, H);
        const f4 output_tex_size = tex_size4(W, H);
        const uint32_t frame_hash = hash1(fc.frame_index);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float depth = in.depth.ld(x, y);
                if (0 == depth) { st4(irradiance_output_tex, x, y, mk4(0.0f)); continue; }
                const f2 uv = get_uv(float(x), float(y), gbuffer_tex_size);
                const ViewRayContext vrc = ViewRayContext::from_uv_and_depth(fc, uv, depth);
                GbufferData gbuffer = gbuffer_unpack(in.gbuffer.ld(x, y));
                const f3 center_normal_ws = gbuffer.normal;
                const f3 center_normal_vs = direction_world_to_view(fc, center_normal_ws);
                const float center_depth = depth;
                const float center_ssao = from_unorm8(in.ssao.ld(x, y));
                const uint32_t px_idx_in_quad = (((uint32_t(x) & 1) | (uint32_t(y) & 1) * 2) + frame_hash) & 3;
                const f4 blue = blue_noise_for_pixel(in.blue_noise, x, y, fc.frame_index) * M_TAU_F;
                const float NEAR_FIELD_FADE_OUT_END = -vrc.ray_hit_vs().z * (SSGI_NEAR_FIELD_RADIUS * output_tex_size.w * 0.5f);
                const float NEAR_FIELD_FADE_OUT_START = NEAR_FIELD_FADE_OUT_END * 0.5f;
                const float near_field_influence = center_ssao;
                f3 total_irradiance = mk3(0.0f);
                bool sharpen_gi_kernel = false;
                {
                    float w_sum = 0;
                    f3 weighted_irradiance = mk3(0.0f);
                    for (uint32_t sample_i = 0; sample_i < 4; ++sample_i) {
                        const float ang = (float(sample_i) + blue.x) * GOLDEN_ANGLE + (float(px_idx_in_quad) / 4.0f) * M_TAU_F;
                        const float radius = powf(float(sample_i), 0.666f) * 1.0f + 0.4f;
                        const f2 rpo = cos_sin_turns(ang) * radius;
                        const int rx = int(floorf(float(x) * 0.5f + rpo.x)), ry = int(floorf(float(y) * 0.5f + rpo.y));
                        const f2 rpx_uv = get_uv(float(rx * 2 + off.x), float(ry * 2 + off.y), gbuffer_tex_size);
                        const float rpx_depth = half_depth_tex.ld(rx, ry);
                        const ViewRayContext rpx_ray_ctx = ViewRayContext::from_uv_and_depth(fc, rpx_uv, rpx_depth);
                        const f3 hit_ws = xyz(ld4(candidate_hit_tex, rx, ry)) + rpx_ray_ctx.ray_hit_ws();
                        const f3 sample_offset = hit_ws - vrc.ray_hit_ws();
                        const float sample_dist = length(sample_offset);
                        const f3 sample_dir = sample_offset / sample_dist;
                        const float geometric_term = 2 * fmaxf(0.0f, dot(center_normal_ws, sample_dir));
                        const float atten = smoothstep(NEAR_FIELD_FADE_OUT_END, NEAR_FIELD_FADE_OUT_START, sample_dist);
                        sharpen_gi_kernel |= atten > 0.9f;
                        f3 contribution = xyz(ld4(candidate_radiance_tex, rx, ry)) * geometric_term;
                        contribution *= lerp(0.0f, atten, near_field_influence);
                        f3 sample_normal_vs = ld_nrm_snorm8(half_view_normal_tex, rx, ry);
                        float w = 1;
                        w *= ggx_ndf_unnorm(0.01f, saturate(dot(center_normal_vs, sample_normal_vs)));
                        w *= exp2f(-200.0f * fabsf(center_normal_vs.z * (center_depth / rpx_depth - 1.0f)));
                        weighted_irradiance += contribution * w;
                        w_sum += w;
                    }
                    total_irradiance += weighted_irradiance / fmaxf(1e-20f, w_sum);
                }
                {
                    float w_sum = 0;
                    f3 weighted_irradiance = mk3(0.0f);
                    const float kernel_scale = sharpen_gi_kernel ? 0.5f : 1.0f;
                    for (uint32_t sample_i = 0; sample_i < 4; ++sample_i) {
                        const float ang = (float(sample_i) + blue.x) * GOLDEN_ANGLE + (float(px_idx_in_quad) / 4.0f) * M_TAU_F;
                        const float radius = powf(float(sample_i), 0.666f) * 1.0f * kernel_scale + 0.4f * kernel_scale;
                        const f2 rpo = cos_sin_turns(ang) * radius;
                        const int rx = int(floorf(float(x) * 0.5f + rpo.x)), ry = int(floorf(float(y) * 0.5f + rpo.y));
                        Reservoir1spp r = Reservoir1spp::from_raw(reservoir_input_tex.ld(rx, ry));
                        const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
                        const TemporalReservoirOutput spx_packed = TemporalReservoirOutput::from_raw(temporal_reservoir_packed_tex.ld(spx_x, spx_y));
                        const f2 spx_uv = get_uv(float(spx_x * 2 + off.x), float(spx_y * 2 + off.y), gbuffer_tex_size);
                        const ViewRayContext spx_ray_ctx = ViewRayContext::from_uv_and_depth(fc, spx_uv, spx_packed.depth);
                        const float rpx_depth = half_depth_tex.ld(rx, ry);
                        const f3 hit_ws = spx_packed.ray_hit_offset_ws + spx_ray_ctx.ray_hit_ws();
                        const f3 sample_offset = hit_ws - vrc.ray_hit_ws();
                        const float sample_dist = length(sample_offset);
                        const f3 sample_dir = sample_offset / sample_dist;
                        const float geometric_term = 2 * fmaxf(0.0f, dot(center_normal_ws, sample_dir));
                        f3 radiance = xyz(ld4(radiance_tex, spx_x, spx_y));
                        {
                            const float atten = smoothstep(NEAR_FIELD_FADE_OUT_START, NEAR_FIELD_FADE_OUT_END, sample_dist);
                            radiance *= lerp(1.0f, atten, near_field_influence);
                        }
                        const f3 contribution = radiance * geometric_term * r.W;
                        f3 sample_normal_vs = ld_nrm_snorm8(half_view_normal_tex, spx_x, spx_y);
                        const float sample_ssao = from_unorm8(in.ssao.ld(rx * 2 + off.x, ry * 2 + off.y));
                        float w = 1;
                        w *= ggx_ndf_unnorm(0.01f, saturate(dot(center_normal_vs, sample_normal_vs)));
                        w *= exp2f(-200.0f * fabsf(center_normal_vs.z * (center_depth / rpx_depth - 1.0f)));
                        w *= exp2f(-20.0f * fabsf(center_ssao - sample_ssao));
                        weighted_irradiance += contribution * w;
                        w_sum += w;
                    }
                    total_irradiance += weighted_irradiance / fmaxf(1e-20f, w_sum);
                }
                st4(irradiance_output_tex, x, y, mk4(total_irradiance, 1));
            }
    }

    // ------------------------------------------------------------------ temporal_filter.hlsl:39-252
    void pass_temporal_filter(const FrameConstants& fc, const RtdgiInputs& in, ImgRGBA16F input_tex, ImgRGBA16F history_tex,
                              ImgRG16F variance_history_tex, ImgRG16F rt_history_invalidity_tex, ImgRGBA16F output_tex,
                              ImgRGBA16F history_output_tex, ImgRG16F variance_history_output_tex) {
        const f4 output_tex_size = tex_size4(W, H);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                f2 uv = get_uv(float(x), float(y), output_tex_size);
                f4 center = linear_rgb_to_crunched_luma_chroma(ld4(input_tex, x, y));
                f4 reproj = ld_reproj(in.reprojection_map, x, y);
                const f4 history_mult{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
                f4 history = linear_rgb_to_crunched_luma_chroma(ld4(history_tex, x, y) * history_mult);
                f4 vsum = mk4(0.0f), vsum2 = mk4(0.0f);
                float wsum = 0, hist_diff = 0, hist_vsum = 0, hist_vsum2 = 0;
                const int k = 2;
                for (int dy = -k; dy <= k; ++dy)
                    for (int dx = -k; dx <= k; ++dx) {
                        f4 neigh = linear_rgb_to_crunched_luma_chroma(ld4(input_tex, x + dx, y + dy));
                        f4 hist_neigh = linear_rgb_to_crunched_luma_chroma(ld4(history_tex, x + dx, y + dy) * history_mult);
                        float neigh_luma = neigh.x, hist_luma = hist_neigh.x;
                        float w = expf(-3.0f * float(dx * dx + dy * dy) / float((k + 1.) * (k + 1.)));
                        vsum += neigh * w;
                        vsum2 += neigh * neigh * w;
                        wsum += w;
                        hist_diff += fabsf(neigh_luma - hist_luma) / fmaxf(1e-5f, neigh_luma + hist_luma) * w;
                        hist_vsum += hist_luma * w;
                        hist_vsum2 += hist_luma * hist_luma * w;
                    }
                f4 ex = vsum / wsum, ex2 = vsum2 / wsum;
                f4 dev = vsqrt(vmax(mk4(0.0f), ex2 - ex * ex));
                hist_diff /= wsum; hist_vsum /= wsum; hist_vsum2 /= wsum;
                const f2 moments_history = sample_bilinear_clamp(variance_history_tex, uv + f2{reproj.x, reproj.y}) *
                                           f2{fc.pre_exposure_delta, fc.pre_exposure_delta * fc.pre_exposure_delta};
                const float center_luma = center.x + (hist_vsum - ex.x);
                const f2 current_moments{center_luma, center_luma * center_luma};
                f2 mo = lerp(moments_history, current_moments, 0.25f);
                st2(variance_history_output_tex, x, y, f2{fmaxf(0.0f, mo.x), fmaxf(0.0f, mo.y)});
                const float center_temporal_dev = sqrtf(fmaxf(0.0f, moments_history.y - moments_history.x * moments_history.x));
                float temporal_change = fabsf(hist_vsum - ex.x) / fmaxf(1e-8f, hist_vsum + ex.x);
                const float rt_invalid = saturate(sqrtf(ld2(rt_history_invalidity_tex, x / 2, y / 2).x) * 4);
                const float current_sample_count = history.w;
                float clamp_box_size = 1 * lerp(0.25f, 2.0f, 1.0f - rt_invalid) * lerp(0.333f, 1.0f, saturate(reproj.w)) * 2;
                clamp_box_size = fmaxf(clamp_box_size, 0.5f);
                f4 nmin = center - dev * clamp_box_size, nmax = center + dev * clamp_box_size;
                f4 clamped_history = mk4(vclamp(xyz(history), xyz(nmin), xyz(nmax)), history.w);
                const float variance_adjusted_temporal_change = smoothstep(0.1f, 1.0f, 0.05f * temporal_change / center_temporal_dev);
                float max_sample_count = 32;
                max_sample_count = lerp(max_sample_count, 4.0f, variance_adjusted_temporal_change);
                max_sample_count *= lerp(1.0f, 0.5f, rt_invalid);
                f3 res = lerp(xyz(clamped_history), xyz(center), 1.0f / (1.0f + fminf(max_sample_count, current_sample_count)));
                const float output_sample_count = fminf(current_sample_count, max_sample_count) + 1;
                f4 output = crunched_luma_chroma_to_linear_rgb(mk4(res, output_sample_count));
                st4(history_output_tex, x, y, output);
                st4(output_tex, x, y, mk4(xyz(output), saturate(output_sample_count * lerp(1.0f, 0.5f, rt_invalid) * smoothstep(0.3f, 0.0f, temporal_change) / 32.0f)));
            }
    }

    // ------------------------------------------------------------------ spatial_filter.hlsl:34-101
    static f3 crunch(f3 v) { return v * (1.0f / (max3(v.x, v.y, v.z) + 1.0f)); }
    static f3 uncrunch(f3 v) { return v * (1.0f / (1.0f - max3(v.x, v.y, v.z))); }
    void pass_spatial_filter(const FrameConstants& fc, const RtdgiInputs& in, ImgRGBA16F input_tex, ImgRGBA16F output_tex) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const f4 c = ld4(input_tex, x, y);
                const float center_validity = c.w;
                const float center_depth = in.depth.ld(x, y);
                const float center_ssao = from_unorm8(in.ssao.ld(x, y));
                const f3 center_value = xyz(c);
                const f3 center_normal_vs = unpack_a2r10g10b10(in.geometric_normal.ld(x, y)) * 2.0f - 1.0f;
                if (center_validity == 1) { st4(output_tex, x, y, mk4(center_value, 1.0f)); continue; }
                const float ang_off = float((fc.frame_index * 23u) % 32u) * M_TAU_F + interleaved_gradient_noise(x, y) * M_PI_F;
                const uint32_t MAX_SAMPLE_COUNT = 8;
                const float MAX_RADIUS_PX = sqrtf(lerp(16.0f * 16.0f, 2.0f * 2.0f, center_validity));
                const float KERNEL_SHARPNESS = 0.666f;
                // clamp(uint(exp2(...)), 2, 8)
                const uint32_t sample_count = std::min(std::max(uint32_t(exp2f(4.0f * square(1.0f - center_validity))), 2u), MAX_SAMPLE_COUNT);
                f4 sum = mk4(crunch(center_value), 1);
                const float RADIUS_SAMPLE_MULT = MAX_RADIUS_PX / powf(float(MAX_SAMPLE_COUNT - 1), KERNEL_SHARPNESS);
                for (uint32_t sample_i = 1; sample_i < MAX_SAMPLE_COUNT; ++sample_i) {
                    const float ang = (float(sample_i) + ang_off) * GOLDEN_ANGLE;
                    float radius = powf(float(sample_i), KERNEL_SHARPNESS) * RADIUS_SAMPLE_MULT;
                    f2 so = cos_sin_turns(ang) * radius;
                    // int2 sample_px = px + sample_offset : uint2 + float2 -> float2 -> int2 (truncation)
                    const int sx = int(float(x) + so.x), sy = int(float(y) + so.y);
                    const float sample_depth = in.depth.ld(sx, sy);
                    const f3 sample_val = xyz(ld4(input_tex, sx, sy));
                    const float sample_ssao = from_unorm8(in.ssao.ld(sx, sy));
                    if (sample_depth != 0 && sample_i < sample_count) {
                        float wt = 1;
                        wt *= exp2f(-100.0f * fabsf(center_normal_vs.z * (center_depth / sample_depth - 1.0f)));
                        wt *= exp2f(-20.0f * fabsf(sample_ssao - center_ssao));
                        sum += mk4(crunch(sample_val), 1.0f) * wt;
                    }
                }
                float norm_factor = 1.0f / fmaxf(1e-5f, sum.w);
                f3 filtered = uncrunch(xyz(sum) * norm_factor);
                st4(output_tex, x, y, mk4(filtered, 1.0f));
            }
    }

    // ------------------------------------------------------------------ RtdgiRenderer::render (rtdgi.rs:173-554)
    struct Output { ImgRGBA16F screen_irradiance_tex, candidate_radiance_tex, candidate_hit_tex; ImgU32 candidate_normal_tex; };
    Output render(const FrameConstants& fc, const RtdgiInputs& in, uint32_t pass_mask = KJ_RTDGI_PASS_ALL) {
        sun_color = sun_color_in_direction(fc, sun_direction(fc));
        rays_closest = 0; rays_any = 0;
        if (pass_mask & KJ_RTDGI_PASS_KEEP_TEMPORALS) for (bool& f : flip) f = !f;
        ImgR8S half_ssao_tex = get<int8_t>("half_ssao_tex", hw, hh);
        ImgU32 half_view_normal_tex = get<uint32_t>("half_view_normal_tex", hw, hh);
        ImgR32F half_depth_tex = get<float>("half_depth_tex", hw, hh);
        if (pass_mask & KJ_RTDGI_PASS_EXTRACT_HALF)
            extract_half_res(fc, W, H, in.gbuffer, in.depth, in.ssao, half_view_normal_tex, half_depth_tex, half_ssao_tex);

        ImgRGBA16F hit_normal_output_tex, hit_normal_history_tex; pingpong("rtdgi.hit_normal", 0, hw, hh, hit_normal_output_tex, hit_normal_history_tex);
        ImgRGBA16F candidate_output_tex, candidate_history_tex;   pingpong("rtdgi.candidate", 1, hw, hh, candidate_output_tex, candidate_history_tex);
        ImgRGBA16F candidate_radiance_tex = get<h4>("candidate_radiance_tex", hw, hh);
        ImgU32 candidate_normal_tex = get<uint32_t>("candidate_normal_tex", hw, hh);
        ImgRGBA16F candidate_hit_tex = get<h4>("candidate_hit_tex", hw, hh);
        ImgU4 temporal_reservoir_packed_tex = get<u4>("temporal_reservoir_packed_tex", hw, hh);
        ImgRG16F invalidity_output_tex, invalidity_history_tex;   pingpong("rtdgi.invalidity", 2, hw, hh, invalidity_output_tex, invalidity_history_tex);
        ImgRGBA16F radiance_output_tex, radiance_history_tex;     pingpong("rtdgi.radiance", 3, hw, hh, radiance_output_tex, radiance_history_tex);
        ImgRGBA32F ray_orig_output_tex, ray_orig_history_tex;     pingpong("rtdgi.ray_orig", 4, hw, hh, ray_orig_output_tex, ray_orig_history_tex);
        ImgRGBA16F ray_output_tex, ray_history_tex;               pingpong("rtdgi.ray", 5, hw, hh, ray_output_tex, ray_history_tex);
        ImgR8 rt_history_validity_pre_input_tex = get<uint8_t>("rt_history_validity_pre_input_tex", hw, hh);
        ImgU2 reservoir_output_tex, reservoir_history_tex;        pingpong("rtdgi.reservoir", 6, hw, hh, reservoir_output_tex, reservoir_history_tex);
        ImgR8 rt_history_validity_input_tex = get<uint8_t>("rt_history_validity_input_tex", hw, hh);

        if (pass_mask & KJ_RTDGI_PASS_VALIDATE)
            pass_validate(fc, in, half_view_normal_tex, reservoir_history_tex, ray_history_tex, radiance_history_tex, ray_orig_history_tex, rt_history_validity_pre_input_tex);
        if (pass_mask & KJ_RTDGI_PASS_TRACE)
            pass_trace(fc, in, half_view_normal_tex, candidate_radiance_tex, candidate_normal_tex, candidate_hit_tex, rt_history_validity_pre_input_tex, rt_history_validity_input_tex);
        if (pass_mask & KJ_RTDGI_PASS_VALIDITY_INTEGRATE)
            pass_validity_integrate(fc, in, rt_history_validity_input_tex, invalidity_history_tex, half_depth_tex, invalidity_output_tex);
        if (pass_mask & KJ_RTDGI_PASS_RESTIR_TEMPORAL)
            pass_restir_temporal(fc, in, half_view_normal_tex, candidate_radiance_tex, candidate_normal_tex, candidate_hit_tex,
                                 radiance_history_tex, ray_orig_history_tex, ray_history_tex, reservoir_history_tex, hit_normal_history_tex,
                                 candidate_history_tex, invalidity_output_tex, radiance_output_tex, ray_orig_output_tex, ray_output_tex,
                                 hit_normal_output_tex, reservoir_output_tex, candidate_output_tex, temporal_reservoir_packed_tex);

        ImgU2 reservoir_output_tex0 = get<u2>("reservoir_output_tex0", hw, hh);
        ImgU2 reservoir_output_tex1 = get<u2>("reservoir_output_tex1", hw, hh);
        ImgU2 reservoir_input_tex = reservoir_output_tex;
        for (uint32_t i = 0; i < spatial_reuse_pass_count; ++i) {
            const uint32_t perform_occlusion_raymarch = (i + 1 == spatial_reuse_pass_count) ? 1 : 0;
            if (pass_mask & KJ_RTDGI_PASS_RESTIR_SPATIAL)
                pass_restir_spatial(fc, in, reservoir_input_tex, half_view_normal_tex, half_depth_tex, half_ssao_tex,
                                    temporal_reservoir_packed_tex, reservoir_output_tex0, i, perform_occlusion_raymarch, use_raytraced_reservoir_visibility ? 1 : 0);
            std::swap(reservoir_output_tex0, reservoir_output_tex1);
            reservoir_input_tex = reservoir_output_tex1;
        }
        if (use_raytraced_reservoir_visibility && (pass_mask & KJ_RTDGI_PASS_RESTIR_SPATIAL))   // "restir check" (rtdgi.rs:478-494)
            pass_restir_check(fc, in, half_depth_tex, temporal_reservoir_packed_tex, reservoir_input_tex);
        ImgRGBA16F irradiance_output_tex = get<h4>("irradiance_output_tex", W, H);
        if (pass_mask & KJ_RTDGI_PASS_RESTIR_RESOLVE)
            pass_restir_resolve(fc, in, radiance_output_tex, reservoir_input_tex, half_view_normal_tex, half_depth_tex,
                                candidate_radiance_tex, candidate_hit_tex, temporal_reservoir_packed_tex, irradiance_output_tex);

        ImgRG16F temporal_variance_output_tex, variance_history_tex; pingpong("rtdgi.temporal2_var", 7, W, H, temporal_variance_output_tex, variance_history_tex);
        ImgRGBA16F temporal_filtered_tex = get<h4>("temporal_filtered_tex", W, H);
        if (pass_mask & KJ_RTDGI_PASS_TEMPORAL_FILTER)
            pass_temporal_filter(fc, in, irradiance_output_tex, reprojected_history_tex, variance_history_tex, invalidity_output_tex,
                                 temporal_filtered_tex, temporal_output_tex, temporal_variance_output_tex);
        ImgRGBA16F spatial_filtered_tex = get<h4>("spatial_filtered_tex", W, H);
        if (pass_mask & KJ_RTDGI_PASS_SPATIAL_FILTER)
            pass_spatial_filter(fc, in, temporal_filtered_tex, spatial_filtered_tex);
        return Output{spatial_filtered_tex, candidate_radiance_tex, candidate_hit_tex, candidate_normal_tex};
    }
};

} // namespace okj
