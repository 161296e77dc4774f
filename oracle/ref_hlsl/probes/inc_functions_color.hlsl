// TEST INFRASTRUCTURE (ours, not the reference's): the second probe pass -- the colour science of the display transform (inc/color/*.hlsl), the G-buffer record
// (inc/gbuffer.hlsl), inc/soft_color_clamp.hlsl, inc/uv.hlsl and the sky model (inc/atmosphere_felix.hlsl), included from /root/reference/assets/shaders/inc where they
// lie, evaluated on a buffer of inputs, one output row per (function group, input). tests/test_ref_hlsl.py compares every row with the oracle's restatement of the same
// function (oracle/okj_api.cpp: okj_probe_functions_color), bit for bit.
#include "../inc/samplers.hlsl"
#include "../inc/bindless_textures.hlsl"
#include "../inc/hash.hlsl"

// (the Bezold-Brucke table reaches the display transform the way post_combine.hlsl hands it over: bindless slot 2 through sampler_llr)
#define DECLARE_BEZOLD_BRUCKE_LUT
static float2 SAMPLE_BEZOLD_BRUCKE_LUT(float coord) {
    return bindless_textures[BINDLESS_LUT_BEZOLD_BRUCKE].SampleLevel(sampler_llr, float2(coord, 0.5), 0).xy;
}
#include "../inc/color/display_transform.hlsl"
#include "../inc/gbuffer.hlsl"
#include "../inc/soft_color_clamp.hlsl"
#include "../inc/uv.hlsl"
#include "../inc/atmosphere_felix.hlsl"

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
    const float3 col = abs(f.xyz);                     // a non-negative colour of any magnitude
    const float3 ucol = float3(uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z));       // a colour in [0, 1)
    const float2 urand = float2(uint_to_u01_float(u.w), uint_to_u01_float(hash1(u.w)));
    uint k = 0;
    #define OUT(v) probe_out[(k++) * probe_count + i] = (v)
    OUT(uint4(asuint(sRGB_to_XYZ(col)), 0));
    OUT(uint4(asuint(XYZ_to_sRGB(f.xyz)), 0));
    OUT(uint4(asuint(CIE_XYZ_to_xyY(col)), 0));
    OUT(uint4(asuint(CIE_xyY_to_XYZ(float3(ucol.xy * 0.8 + 0.1, col.z))), 0));
    OUT(uint4(asuint(XYZ_to_IPT(f.xyz)), 0));
    OUT(uint4(asuint(IPT_to_XYZ(float3(ucol.x, ucol.y - 0.5, ucol.z - 0.5))), 0));
    OUT(uint4(asuint(CIE_xyY_xy_to_LUV_uv(ucol.xy)), asuint(CIE_XYZ_to_LUV_uv(col))));
    OUT(uint4(asuint(catmull_rom(ucol.x, f.x, f.y, f.z, urand.x)), asuint(compress_luminance(col.x)), 0, 0));
    {
        const HelmholtzKohlrauschEffect hk = hk_from_sRGB(ucol);
        OUT(uint4(asuint(XYZ_to_hk_luminance_multiplier_custom_g0(col)), asuint(hk.mult), asuint(srgb_to_equivalent_luminance(hk, ucol.zxy)), 0));
    }
    OUT(uint4(asuint(XYZ_to_LAB(col)), asuint(bb_xy_white_offset_to_lut_coord(ucol.xy - 0.5))));
    OUT(uint4(asuint(bezold_brucke_shift_XYZ_with_lut(sRGB_to_XYZ(ucol), urand.x)), 0));
    OUT(uint4(asuint(display_transform_sRGB(ucol)), 0));
    OUT(uint4(asuint(display_transform_sRGB(ucol * min(col.x, 4096.0))), 0));
    OUT(uint4(asuint(display_transform_sRGB(col)), 0));                             // (any magnitude: the test also feeds it the colours of a rendered frame)
    {
        GbufferData g = GbufferData::create_zero();
        g.albedo = ucol;
        g.normal = unit;
        g.roughness = urand.x;
        g.metalness = urand.y;
        g.emissive = col;
        OUT(g.pack().data0);
        const GbufferData d = GbufferDataPacked::from_uint4(u).unpack();
        OUT(uint4(asuint(d.albedo), asuint(d.roughness)));
        OUT(uint4(asuint(d.normal), asuint(d.metalness)));
        OUT(uint4(asuint(d.emissive), 0));
    }
    OUT(uint4(asuint(soft_color_clamp(ucol, col, ucol.zxy, ucol.yzx * 0.3)), 0));
    {
        const float4 tex_size = float4(1920, 1080, 1.0 / 1920, 1.0 / 1080);
        OUT(uint4(asuint(get_uv(int2(u.xy & 4095), tex_size)), asuint(get_uv(float2(col.x, col.y), tex_size))));
        OUT(uint4(asuint(cs_to_uv(f.xy)), asuint(uv_to_cs(ucol.xy))));
    }
    {   // the sky: a start point up to 50 km above the ground, any direction
        const float3 start = float3(f.x, col.y * 0.05, f.z);
        const float costh = ucol.x * 2 - 1;
        OUT(uint4(asuint(SphereIntersection(start, unit, PLANET_CENTER, PLANET_RADIUS + ATMOSPHERE_HEIGHT)), asuint(PhaseRayleigh(costh)), asuint(PhaseMie(costh))));
        OUT(uint4(asuint(AtmosphereDensity(col.x)), asuint(AtmosphereHeight(start))));
        OUT(uint4(asuint(IntegrateOpticalDepth(start, unit)), 0));
        OUT(uint4(asuint(Absorb(col)), 0));
        float3 transmittance;
        const float3 light_dir = normalize(ucol * 2 - 1);
        OUT(uint4(asuint(IntegrateScattering(start, unit, INFINITY, light_dir, 1.0.xxx, transmittance)), 0));
    }
}
