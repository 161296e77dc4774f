// Device self test (no reference counterpart): the leaf functions of the device headers -- kj_vec.hpp (inc/hash.hlsl, pack_unpack.hlsl, quasi_random.hlsl, math.hlsl,
// color.hlsl), kj_reservoir.hpp (inc/reservoir.hlsl), kj_shading.hpp (inc/brdf.hlsl), kj_ircache.hpp (octa_decode), kj_rtr.hpp (exponential_(un)squish) -- evaluated on a
// buffer of inputs, one output row per function, in the row order of oracle/ref_hlsl/probes/inc_functions.hlsl. That probe runs the reference's own text and
// tests/test_ref_hlsl.py holds the oracle's restatement to it bit for bit; tests/test_gpu_parity.py holds THESE rows to the oracle's: the chain reference text ->
// oracle -> device code, function by function instead of pass by pass. Compiled without FMA contraction like the ray passes' units (csrc/Makefile).
#include "kj_rtr.hpp"
#include "kj_ircache.hpp"
#include "kj_color.hpp"

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())
#define KJ_PROBE_ROWS 27u
#define KJ_PROBE_COLOR_ROWS 26u
#define KJ_PROBE_SHADING_ROWS 29u
#define KJ_PROBE_MISC_ROWS 10u

__global__ void __launch_bounds__(256) k_probe_functions(const uint4* __restrict__ in4, uint32_t n, uint4* __restrict__ out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 u = in4[i];
    const V3 f{asfloat(u.x), asfloat(u.y), asfloat(u.z)};
    const V3 unit = normalize(f);
    const V2 urand{uint_to_u01_float(u.x), uint_to_u01_float(u.y)};
    const V3 col = vabs(f);
    const V3 scol{saturate(col.x), saturate(col.y), saturate(col.z)};
    uint32_t k = 0;
    auto OUT = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { out4[size_t(k++) * n + i] = make_uint4(a, b, c, d); };
    auto U = [](float v) { return asuint(v); };
    OUT(hash1(u.x), hash_combine2(u.x, u.y), hash2(u.x, u.y), hash3(u.x, u.y, u.z));
    OUT(U(uint_to_u01_float(u.x)), U(interleaved_gradient_noise(u.x & 4095u, u.y & 4095u)), 0, 0);
    OUT(U(unpack_unorm(u.x, 8)), pack_unorm(urand.x, 11), U(unpack_unorm(u.y, 11)), pack_unorm(urand.y, 10));
    const uint32_t packed_n = pack_normal_11_10_11(unit);
    { const V3 v = unpack_normal_11_10_11(packed_n); OUT(packed_n, U(v.x), U(v.y), U(v.z)); }
    { const V3 v = unpack_normal_11_10_11_no_normalize(u.w); OUT(U(v.x), U(v.y), U(v.z), 0); }
    { const V3 v = unpack_normal_11_10_11_no_normalize(u.w); OUT(U(v.x), U(v.y), U(v.z), 0); }     // (the float- and the uint-argument forms are one function here)
    { const V3 v = unpack_color_888(u.x); OUT(pack_color_888(scol), U(v.x), U(v.y), U(v.z)); }
    { const V2 v = unpack_2x16f_uint(u.z); OUT(pack_2x16f_uint(f.x, f.y), U(v.x), U(v.y), 0); }
    { const V3 v = rgb9e5_to_float3(u.y); OUT(float3_to_rgb9e5(col), U(v.x), U(v.y), U(v.z)); }
    { const V3 v = octa_decode(urand); OUT(U(v.x), U(v.y), U(v.z), 0); }
    OUT(0, 0, U(max3(f.x, f.y, f.z)), 0);                                                            // (octa_wrap has no device form: octa_decode inlines it)
    { const V2 v = hammersley(u.y & 1023u, 1024u); OUT(U(radical_inverse_vdc(u.x)), U(v.x), U(v.y), 0); }
    { const V2 v = r2_sequence(u.z & 0xffffu); OUT(U(v.x), U(v.y), 0, 0); }
    const Basis basis = build_orthonormal_basis(unit);
    const V3 b0 = to_world(basis, V3{1, 0, 0}), b1 = to_world(basis, V3{0, 1, 0}), b2 = to_world(basis, V3{0, 0, 1});
    OUT(U(b0.x), U(b0.y), U(b0.z), U(b1.x));
    OUT(U(b1.y), U(b1.z), U(b2.x), U(b2.y));
    { const V3 v = uniform_sample_cone(urand, 0.5f + 0.5f * urand.x); OUT(U(v.x), U(v.y), U(v.z), U(b2.z)); }
    { const V3 v = uniform_sample_hemisphere(urand); OUT(U(v.x), U(v.y), U(v.z), U(inverse_depth_relative_diff(fabsf(f.x), fabsf(f.y)))); }
    OUT(U(exponential_squish(fabsf(f.x), urand.y * 8.0f)), U(exponential_unsquish(urand.x, 0.25f + urand.y)), 0, 0);
    { const V3 v = sRGB_to_YCbCr(col); OUT(U(v.x), U(v.y), U(v.z), U(sRGB_to_luminance(col))); }
    { const V3 v = YCbCr_to_sRGB(f); OUT(U(v.x), U(v.y), U(v.z), 0); }
    {
        Reservoir1spp r = Reservoir1spp::from_raw(make_uint2(u.x, u.y));
        uint32_t rng = u.z;
        const bool a = r.update(urand.x * 3.0f, u.w, rng);
        const bool b = r.update(urand.y, u.w ^ 0x5555u, rng);
        r.M = fminf(r.M, 500.0f);
        r.W = fminf(r.W, 1000.0f);
        const uint2 raw = r.as_raw();
        OUT(raw.x, raw.y, U(r.w_sum), (a ? 1u : 0u) | (b ? 2u : 0u) | (rng << 2));
        Reservoir1spp s = Reservoir1spp::create();
        StreamState st{0, 0};
        s.init_with_stream(urand.x, urand.y * 4.0f, st, 17);
        const bool c = s.update_with_stream(r, urand.y + 0.125f, 0.75f, st, u.w, rng);
        s.finish_stream(st);
        OUT(U(s.M), U(s.W), U(s.w_sum), s.payload ^ (c ? 0x80000000u : 0u));
    }
    {
        const float roughness = 0.02f + 0.96f * urand.x;
        const V3 wo = uniform_sample_hemisphere(V2{urand.y, urand.x});
        const V3 wi = uniform_sample_hemisphere(V2{uint_to_u01_float(u.z), uint_to_u01_float(u.w)});
        const BrdfValue v = specular_evaluate(roughness, scol, wo, wi);
        OUT(U(v.value.x), U(v.value.y), U(v.value.z), U(v.pdf));
        OUT(U(v.value_over_pdf.x), U(v.value_over_pdf.y), U(v.value_over_pdf.z), U(v.transmission_fraction.x));
        const BrdfSample s = specular_sample(roughness, scol, wo, V2{uint_to_u01_float(u.w), uint_to_u01_float(u.z)});
        OUT(U(s.wi.x), U(s.wi.y), U(s.wi.z), U(s.pdf));
        OUT(U(s.value_over_pdf.x), U(s.value_over_pdf.y), U(s.value_over_pdf.z), U(s.value.y));
        const BrdfSample d = diffuse_sample(scol, urand);
        OUT(U(d.wi.x), U(d.wi.y), U(d.wi.z), U(diffuse_evaluate(scol, wi).value.z));
    }
}

extern "C" KjStatus kj_selftest_probe_functions(const void* in4_device, uint32_t n, void* out4_device, uint32_t rows_capacity, uint32_t* out_rows, void* stream) {
    KJ_REQUIRE(in4_device && out4_device && out_rows, "null argument");
    KJ_REQUIRE(rows_capacity >= KJ_PROBE_ROWS, "the output buffer holds fewer rows than the probe writes");
    *out_rows = KJ_PROBE_ROWS;
    if (n == 0) return KJ_OK;
    hipLaunchKernelGGL(k_probe_functions, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, (const uint4*)in4_device, n, (uint4*)out4_device);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// The second probe (oracle/ref_hlsl/probes/inc_functions_color.hlsl): the display transform's colour science (kj_color.hpp), the G-buffer record, soft_color_clamp, the
// uv helpers and the sky model (kj_shading.hpp). Words of functions that have no device form of their own (XYZ_to_LAB: unused by the active branch of the transform; the
// phase functions and the density-by-height, which the device folds into integrate_scattering / atmosphere_density_at) stay 0 and the test leaves them out.
__global__ void __launch_bounds__(256) k_probe_functions_color(const uint4* __restrict__ in4, uint32_t n, const uint32_t* __restrict__ bb_lut, uint4* __restrict__ out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 u = in4[i];
    const V3 f{asfloat(u.x), asfloat(u.y), asfloat(u.z)};
    const V3 unit = normalize(f);
    const V3 col = vabs(f);
    const V3 ucol{uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z)};
    const V2 urand{uint_to_u01_float(u.w), uint_to_u01_float(hash1(u.w))};
    uint32_t k = 0;
    auto OUT = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { out4[size_t(k++) * n + i] = make_uint4(a, b, c, d); };
    auto U = [](float v) { return asuint(v); };
    auto OUT3 = [&](V3 v, float w = 0.0f) { OUT(U(v.x), U(v.y), U(v.z), U(w)); };
    OUT3(col_sRGB_to_XYZ(col));
    OUT3(col_XYZ_to_sRGB(f));
    OUT3(CIE_XYZ_to_xyY(col));
    OUT3(CIE_xyY_to_XYZ(V3{ucol.x * 0.8f + 0.1f, ucol.y * 0.8f + 0.1f, col.z}));
    OUT3(XYZ_to_IPT(f));
    OUT3(IPT_to_XYZ(V3{ucol.x, ucol.y - 0.5f, ucol.z - 0.5f}));
    { const V2 a = CIE_xyY_xy_to_LUV_uv(V2{ucol.x, ucol.y}), b = CIE_XYZ_to_LUV_uv(col); OUT(U(a.x), U(a.y), U(b.x), U(b.y)); }
    OUT(U(col_catmull_rom(ucol.x, f.x, f.y, f.z, urand.x)), U(compress_luminance(col.x)), 0, 0);
    {
        const float hk = hk_from_sRGB(ucol);
        OUT(U(XYZ_to_hk_luminance_multiplier_custom_g0(col)), U(hk), U(srgb_to_equivalent_luminance(hk, V3{ucol.z, ucol.x, ucol.y})), 0);
    }
    OUT(0, 0, 0, U(bb_xy_white_offset_to_lut_coord(V2{ucol.x - 0.5f, ucol.y - 0.5f})));
    OUT3(bezold_brucke_shift_XYZ_with_lut(bb_lut, col_sRGB_to_XYZ(ucol), urand.x));
    OUT3(display_transform_sRGB(bb_lut, ucol));
    OUT3(display_transform_sRGB(bb_lut, ucol * fminf(col.x, 4096.0f)));
    OUT3(display_transform_sRGB(bb_lut, col));
    {
        GbufferData g;
        g.albedo = ucol; g.normal = unit; g.roughness = urand.x; g.metalness = urand.y; g.emissive = col;
        const uint4 p = gbuffer_pack(g);
        OUT(p.x, p.y, p.z, p.w);
        const GbufferData d = gbuffer_unpack(u);
        OUT3(d.albedo, d.roughness);
        OUT3(d.normal, d.metalness);
        OUT3(d.emissive);
    }
    OUT3(soft_color_clamp(ucol, col, V3{ucol.z, ucol.x, ucol.y}, V3{ucol.y, ucol.z, ucol.x} * 0.3f));
    {
        const V4 tex_size{1920.0f, 1080.0f, 1.0f / 1920.0f, 1.0f / 1080.0f};
        const V2 a = get_uv(float(int(u.x & 4095u)), float(int(u.y & 4095u)), tex_size), b = get_uv(col.x, col.y, tex_size);
        OUT(U(a.x), U(a.y), U(b.x), U(b.y));
        const V2 c = cs_to_uv(V2{f.x, f.y}), d = uv_to_cs(V2{ucol.x, ucol.y});
        OUT(U(c.x), U(c.y), U(d.x), U(d.y));
    }
    {
        const V3 start{f.x, col.y * 0.05f, f.z};
        const V2 s = atmosphere_intersection(start, unit);
        OUT(U(s.x), U(s.y), 0, 0);
        OUT(0, 0, 0, 0);
        OUT3(integrate_optical_depth(start, unit));
        OUT3(atmosphere_absorb(col));
        const V3 light_dir = normalize(ucol * 2.0f - 1.0f);
        OUT3(integrate_scattering(start, unit, INFINITY, light_dir, v3(1.0f)));
    }
}

extern "C" KjStatus kj_selftest_probe_functions_color(const void* in4_device, uint32_t n, const void* bezold_brucke_lut_rg16f_device, void* out4_device, uint32_t rows_capacity,
                                                      uint32_t* out_rows, void* stream) {
    KJ_REQUIRE(in4_device && bezold_brucke_lut_rg16f_device && out4_device && out_rows, "null argument");
    KJ_REQUIRE(rows_capacity >= KJ_PROBE_COLOR_ROWS, "the output buffer holds fewer rows than the probe writes");
    *out_rows = KJ_PROBE_COLOR_ROWS;
    if (n == 0) return KJ_OK;
    hipLaunchKernelGGL(k_probe_functions_color, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, (const uint4*)in4_device, n,
                       (const uint32_t*)bezold_brucke_lut_rg16f_device, (uint4*)out4_device);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// The third probe (oracle/ref_hlsl/probes/inc_functions_shading.hlsl): the view-ray helpers, the ray cone, the layered BRDF with its energy preservation, the sun,
// atmosphere_default, the triangle-light sampler (kj_shading.hpp, kj_scene.hpp, kj_rtr.hpp). Words with no device form of their own (ray_dir_vs, the boost factor -- folded
// into layered_brdf_from_gbuffer_ndotv --, valid_sample_fraction, the measure conversion) stay 0.
__global__ void __launch_bounds__(256) k_probe_functions_shading(FrameConstants fc, const uint4* __restrict__ in4, uint32_t n, const uint2* __restrict__ fg_lut, uint4* __restrict__ out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 u = in4[i];
    const V3 f{asfloat(u.x), asfloat(u.y), asfloat(u.z)};
    const V3 unit = normalize(f);
    const V3 ucol{uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z)};
    const V3 urand{uint_to_u01_float(u.w), uint_to_u01_float(hash1(u.w)), uint_to_u01_float(hash1(u.w + 1u))};
    const float depth = ucol.z * 0.25f + 1e-5f;
    const V2 uv{ucol.x, ucol.y};
    uint32_t k = 0;
    auto OUT = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { out4[size_t(k++) * n + i] = make_uint4(a, b, c, d); };
    auto U = [](float v) { return asuint(v); };
    auto OUT3 = [&](V3 v, float w = 0.0f) { OUT(U(v.x), U(v.y), U(v.z), U(w)); };
    {
        const ViewRay v = view_ray_from_uv(fc, uv);
        OUT3(v.dir_ws);
        OUT3(v.origin_ws);
        const ViewRay h = view_ray_from_uv_and_depth(fc, uv, depth);
        OUT3(h.hit_ws, h.hit_vs.z);
        OUT3(h.biased_secondary_ray_origin_ws());
        OUT3(h.biased_secondary_ray_origin_ws_with_normal(unit));
        OUT3(view_ray_from_uv_and_biased_depth(fc, uv, depth).hit_ws);
    }
    OUT3(get_eye_position(fc), depth_to_view_z(fc, depth));
    OUT3(get_prev_eye_position(fc), pixel_ray_cone_from_image_height(fc, 1080.0f).spread_angle);
    OUT3(direction_view_to_world(fc, f));
    OUT3(direction_world_to_view(fc, f));
    OUT3(position_world_to_view(fc, f));
    OUT3(position_world_to_clip(fc, f));
    OUT3(position_world_to_sample(fc, f));
    {
        const RayCone c = pixel_ray_cone_from_image_height(fc, 720.0f).propagate(urand.x * 0.1f, fabsf(f.x));
        OUT(U(c.width), U(c.spread_angle), U(c.width_at_t(fabsf(f.y))), 0);
    }
    {
        GbufferData g;
        g.albedo = ucol; g.normal = unit; g.roughness = 0.02f + 0.96f * urand.x; g.emissive = v3(0.0f);
        g.metalness = (u.w & 1u) ? urand.y : float((u.w >> 1) & 1u);
        const V3 wo = uniform_sample_hemisphere(V2{urand.y, urand.z});
        const V3 wi = uniform_sample_hemisphere(V2{urand.z, urand.x});
        OUT(0, 0, 0, 0);
        const LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(fg_lut, g, wo.z);
        OUT3(brdf.spec_albedo, brdf.roughness);
        OUT3(brdf.diff_albedo);
        OUT3(brdf.preintegrated_reflection);
        OUT3(brdf.preintegrated_reflection_mult);
        OUT3(brdf.preintegrated_transmission_fraction);
        OUT3(layered_brdf_evaluate(brdf, wo, wi));
        OUT3(layered_brdf_evaluate_directional_light(brdf, wo, wi));
        const BrdfSample s = layered_brdf_sample(brdf, wo, urand);
        OUT3(s.wi, s.pdf);
        OUT3(s.value_over_pdf, s.value.x);
    }
    OUT3(sample_sun_direction(fc, V2{urand.x, urand.y}, true));
    OUT3(sun_color_in_direction(fc, V3{unit.x, fabsf(unit.y), unit.z}));
    OUT3(atmosphere_default(fc, unit, normalize(sun_direction(fc))));
    {
        const LightSampleArea l = sample_triangle_light(f, ucol * 4.0f - 2.0f, urand * 4.0f - 2.0f, V2{ucol.y, ucol.x});
        OUT3(l.pos, l.pdf);
        OUT3(l.normal);
    }
}

extern "C" KjStatus kj_selftest_probe_functions_shading(const KjFrameConstants* frame_constants, const void* in4_device, uint32_t n, const void* brdf_fg_lut_rgba16f_device,
                                                        void* out4_device, uint32_t rows_capacity, uint32_t* out_rows, void* stream) {
    KJ_REQUIRE(frame_constants && in4_device && brdf_fg_lut_rgba16f_device && out4_device && out_rows, "null argument");
    KJ_REQUIRE(rows_capacity >= KJ_PROBE_SHADING_ROWS, "the output buffer holds fewer rows than the probe writes");
    *out_rows = KJ_PROBE_SHADING_ROWS;
    if (n == 0) return KJ_OK;
    hipLaunchKernelGGL(k_probe_functions_shading, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *frame_constants, (const uint4*)in4_device, n,
                       (const uint2*)brdf_fg_lut_rgba16f_device, (uint4*)out4_device);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// The fourth probe (oracle/ref_hlsl/probes/inc_functions_misc.hlsl): TemporalReservoirOutput (kj_reservoir.hpp), the cache's sample parameters and grid addressing
// (kj_ircache.hpp). Rows of functions that are local to a kernel's translation unit on the device (taa's colour mapping, the bilinear helper) stay 0.
__global__ void __launch_bounds__(256) k_probe_functions_misc(FrameConstants fc, const uint4* __restrict__ in4, uint32_t n, uint4* __restrict__ out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 u = in4[i];
    const V3 f{asfloat(u.x), asfloat(u.y), asfloat(u.z)};
    const V3 unit = normalize(f);
    const V3 ucol{uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z)};
    uint32_t k = 0;
    auto OUT = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { out4[size_t(k++) * n + i] = make_uint4(a, b, c, d); };
    auto U = [](float v) { return asuint(v); };
    OUT(0, 0, 0, 0);
    OUT(0, 0, 0, 0);
    OUT(0, 0, 0, 0);
    {
        const TemporalReservoirOutput t = TemporalReservoirOutput::from_raw(u);
        const uint4 r = t.as_raw();
        OUT(r.x, r.y, r.z, r.w);
        OUT(U(t.depth), U(t.ray_hit_offset_ws.x), U(t.ray_hit_offset_ws.y), U(t.ray_hit_offset_ws.z));
        OUT(U(t.luminance), U(t.hit_normal_ws.x), U(t.hit_normal_ws.y), U(t.hit_normal_ws.z));
    }
    {
        const uint32_t s = irc_sample_params(4, u.x & 0xffffu, u.y & 3u, u.z & 0xffffu);
        OUT(s, hash1(s >> 4u), 0, 0);
        const V3 d = irc_sample_direction(s);
        OUT(U(d.x), U(d.y), U(d.z), s % IRC_OCTA_DIMS2);
    }
    {
        const V3 center{fc.ircache_grid_center[0], fc.ircache_grid_center[1], fc.ircache_grid_center[2]};
        const IrcCoord c = irc_ws_pos_to_coord<true>(fc, center + f * 0.01f, unit, ucol - 0.5f);
        OUT(c.x, c.y, c.z, c.cascade);
        OUT(irc_cell_idx(c.x, c.y, c.z, c.cascade), irc_cascade_idx(f * 0.01f, 1u), 0, 0);
    }
}

extern "C" KjStatus kj_selftest_probe_functions_misc(const KjFrameConstants* frame_constants, const void* in4_device, uint32_t n, void* out4_device, uint32_t rows_capacity,
                                                     uint32_t* out_rows, void* stream) {
    KJ_REQUIRE(frame_constants && in4_device && out4_device && out_rows, "null argument");
    KJ_REQUIRE(rows_capacity >= KJ_PROBE_MISC_ROWS, "the output buffer holds fewer rows than the probe writes");
    *out_rows = KJ_PROBE_MISC_ROWS;
    if (n == 0) return KJ_OK;
    hipLaunchKernelGGL(k_probe_functions_misc, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *frame_constants, (const uint4*)in4_device, n, (uint4*)out4_device);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
