// TEST INFRASTRUCTURE (ours, not the reference's): the third probe pass -- the view-ray helpers of inc/frame_constants.hlsl, inc/ray_cone.hlsl, the layered BRDF with its
// energy preservation (inc/layered_brdf.hlsl, inc/brdf_lut.hlsl: the BRDF table comes from bindless slot 0), inc/sun.hlsl, inc/atmosphere.hlsl and the triangle-light sampler
// (inc/lights/triangle.hlsl), included from /root/reference/assets/shaders/inc where they lie. tests/test_ref_hlsl.py compares every row with the oracle's restatement
// (oracle/okj_api.cpp: okj_probe_functions_shading), bit for bit.
#include "../inc/frame_constants.hlsl"
#include "../inc/hash.hlsl"
#include "../inc/math.hlsl"
#include "../inc/brdf.hlsl"
#include "../inc/brdf_lut.hlsl"
#include "../inc/layered_brdf.hlsl"
#include "../inc/atmosphere.hlsl"
#include "../inc/sun.hlsl"
#include "../inc/lights/triangle.hlsl"

[[vk::binding(0)]] StructuredBuffer<uint4> probe_in;
[[vk::binding(1)]] RWStructuredBuffer<uint4> probe_out;
[[vk::binding(2)]] cbuffer _ {
    uint probe_count;
};

[numthreads(64, 1, 1)]
void main(uint i : SV_DispatchThreadID) {
    if (i >= probe_count) {
        return;
    }
    const uint4 u = probe_in[i];
    const float4 f = asfloat(u);                       // finite floats of moderate magnitude, either sign (the test makes them so)
    const float3 unit = normalize(f.xyz);
    const float3 ucol = float3(uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z));
    const float3 urand = float3(uint_to_u01_float(u.w), uint_to_u01_float(hash1(u.w)), uint_to_u01_float(hash1(u.w + 1)));
    const float depth = ucol.z * 0.25 + 1e-5;          // reversed, infinite far plane: near = large
    uint k = 0;
    #define OUT(v) probe_out[(k++) * probe_count + i] = (v)
    {
        const ViewRayContext v = ViewRayContext::from_uv(ucol.xy);
        OUT(uint4(asuint(v.ray_dir_ws()), asuint(v.ray_dir_vs().z)));
        OUT(uint4(asuint(v.ray_origin_ws()), 0));
        const ViewRayContext h = ViewRayContext::from_uv_and_depth(ucol.xy, depth);
        OUT(uint4(asuint(h.ray_hit_ws()), asuint(h.ray_hit_vs().z)));
        OUT(uint4(asuint(h.biased_secondary_ray_origin_ws()), 0));
        OUT(uint4(asuint(h.biased_secondary_ray_origin_ws_with_normal(unit)), 0));
        OUT(uint4(asuint(ViewRayContext::from_uv_and_biased_depth(ucol.xy, depth).ray_hit_ws()), 0));
    }
    OUT(uint4(asuint(get_eye_position()), asuint(depth_to_view_z(depth))));
    OUT(uint4(asuint(get_prev_eye_position()), asuint(pixel_cone_spread_angle_from_image_height(1080.0))));
    OUT(uint4(asuint(direction_view_to_world(f.xyz)), 0));
    OUT(uint4(asuint(direction_world_to_view(f.xyz)), 0));
    OUT(uint4(asuint(position_world_to_view(f.xyz)), 0));
    OUT(uint4(asuint(position_world_to_clip(f.xyz)), 0));
    OUT(uint4(asuint(position_world_to_sample(f.xyz)), 0));
    {
        const RayCone c = pixel_ray_cone_from_image_height(720.0).propagate(urand.x * 0.1, abs(f.x));
        OUT(uint4(asuint(c.width), asuint(c.spread_angle), asuint(c.width_at_t(abs(f.y))), 0));
    }
    {
        GbufferData g = GbufferData::create_zero();
        g.albedo = ucol;
        g.normal = unit;
        g.roughness = 0.02 + 0.96 * urand.x;
        g.metalness = (u.w & 1) ? urand.y : float((u.w >> 1) & 1);
        const float3 wo = uniform_sample_hemisphere(urand.yz);
        const float3 wi = uniform_sample_hemisphere(urand.zx);
        OUT(uint4(asuint(metalness_albedo_boost(g.metalness, g.albedo)), 0));
        const LayeredBrdf brdf = LayeredBrdf::from_gbuffer_ndotv(g, wo.z);
        OUT(uint4(asuint(brdf.specular_brdf.albedo), asuint(brdf.specular_brdf.roughness)));
        OUT(uint4(asuint(brdf.diffuse_brdf.albedo), asuint(brdf.energy_preservation.valid_sample_fraction)));
        OUT(uint4(asuint(brdf.energy_preservation.preintegrated_reflection), 0));
        OUT(uint4(asuint(brdf.energy_preservation.preintegrated_reflection_mult), 0));
        OUT(uint4(asuint(brdf.energy_preservation.preintegrated_transmission_fraction), 0));
        OUT(uint4(asuint(brdf.evaluate(wo, wi)), 0));
        OUT(uint4(asuint(brdf.evaluate_directional_light(wo, wi)), 0));
        const BrdfSample s = brdf.sample(wo, urand);
        OUT(uint4(asuint(s.wi), asuint(s.pdf)));
        OUT(uint4(asuint(s.value_over_pdf), asuint(s.value.x)));
    }
    OUT(uint4(asuint(sample_sun_direction(urand.xy, true)), 0));
    OUT(uint4(asuint(sun_color_in_direction(float3(unit.x, abs(unit.y), unit.z))), 0));
    OUT(uint4(asuint(atmosphere_default(unit, normalize(SUN_DIRECTION))), 0));
    {
        Triangle tri;
        tri.v = f.xyz;
        tri.e0 = ucol * 4 - 2;
        tri.e1 = urand * 4 - 2;
        const LightSampleResultArea l = sample_triangle_light(tri, ucol.yx);
        OUT(uint4(asuint(l.pos), asuint(l.pdf.value)));
        PdfArea pdf;
        pdf.value = l.pdf.value;
        OUT(uint4(asuint(l.normal), asuint(to_projected_solid_angle_measure(pdf, urand.x + 1e-3, urand.y + 1e-3, abs(f.y)))));
    }
}
