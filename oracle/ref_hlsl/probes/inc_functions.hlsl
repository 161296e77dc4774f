// TEST INFRASTRUCTURE (ours, not the reference's): a compute pass that evaluates the leaf functions of the reference's headers -- included from
// /root/reference/assets/shaders/inc where they lie -- on a buffer of inputs, one output row per (function, input). tests/test_ref_hlsl.py compares every row with
// the oracle's restatement of the same function (oracle/okj_api.cpp: okj_probe_functions), bit for bit.
#include "../inc/hash.hlsl"
#include "../inc/pack_unpack.hlsl"
#include "../inc/quasi_random.hlsl"
#include "../inc/math.hlsl"
#include "../inc/color.hlsl"
#include "../inc/working_color_space.hlsl"
#include "../inc/reservoir.hlsl"
#include "../inc/brdf.hlsl"

[[vk::binding(0)]] StructuredBuffer<uint4> probe_in;
[[vk::binding(1)]] RWStructuredBuffer<uint4> probe_out;
[[vk::binding(2)]] cbuffer _ {
    uint probe_count;
};

#define PROBE_ROWS 27

[numthreads(64, 1, 1)]
void main(uint i : SV_DispatchThreadID) {
    if (i >= probe_count) {
        return;
    }
    const uint4 u = probe_in[i];
    const float4 f = asfloat(u);                       // arbitrary bit patterns of finite floats (the test filters NaN / inf out)
    const float3 unit = normalize(f.xyz);              // a direction
    const float2 urand = float2(uint_to_u01_float(u.x), uint_to_u01_float(u.y));
    const float3 col = abs(f.xyz);                     // a non-negative colour
    uint k = 0;
    #define OUT(v) probe_out[(k++) * probe_count + i] = (v)
    OUT(uint4(hash1(u.x), hash_combine2(u.x, u.y), hash2(u.xy), hash3(u.xyz)));
    OUT(uint4(asuint(uint_to_u01_float(u.x)), asuint(interleaved_gradient_noise(u.xy & 4095)), 0, 0));
    OUT(uint4(asuint(unpack_unorm(u.x, 8)), pack_unorm(urand.x, 11), asuint(unpack_unorm(u.y, 11)), pack_unorm(urand.y, 10)));
    const float packed_n = pack_normal_11_10_11(unit);
    OUT(uint4(asuint(packed_n), asuint(unpack_normal_11_10_11(packed_n)).xyz));
    OUT(uint4(asuint(unpack_normal_11_10_11_no_normalize(asfloat(u.w))), 0));
    OUT(uint4(asuint(unpack_normal_11_10_11_uint_no_normalize(u.w)), 0));
    OUT(uint4(pack_color_888(saturate(col)), asuint(unpack_color_888(u.x))));
    OUT(uint4(pack_2x16f_uint(f.xy), asuint(unpack_2x16f_uint(u.z)), 0));
    OUT(uint4(float3_to_rgb9e5(col), asuint(rgb9e5_to_float3(u.y))));
    OUT(uint4(asuint(octa_decode(urand)), 0));
    OUT(uint4(asuint(octa_wrap(urand * 2 - 1)), asuint(max3(f.x, f.y, f.z)), 0));
    OUT(uint4(asuint(radical_inverse_vdc(u.x)), asuint(hammersley(u.y & 1023, 1024)), 0));
    OUT(uint4(asuint(r2_sequence(u.z & 0xffff)), 0, 0));
    const float3x3 basis = build_orthonormal_basis(unit);
    const float3 b0 = mul(basis, float3(1, 0, 0)), b1 = mul(basis, float3(0, 1, 0)), b2 = mul(basis, float3(0, 0, 1));     // its columns
    OUT(uint4(asuint(b0), asuint(b1.x)));
    OUT(uint4(asuint(b1.yz), asuint(b2.xy)));
    OUT(uint4(asuint(uniform_sample_cone(urand, 0.5 + 0.5 * urand.x)), asuint(b2.z)));
    OUT(uint4(asuint(uniform_sample_hemisphere(urand)), asuint(inverse_depth_relative_diff(abs(f.x), abs(f.y)))));
    OUT(uint4(asuint(exponential_squish(abs(f.x), urand.y * 8)), asuint(exponential_unsquish(urand.x, 0.25 + urand.y)), 0, 0));
    OUT(uint4(asuint(sRGB_to_YCbCr(col)), asuint(sRGB_to_luminance(col))));
    OUT(uint4(asuint(YCbCr_to_sRGB(f.xyz)), 0));
    {   // Reservoir1spp: unpack, two updates, pack; then a stream of two reservoirs as the resampling passes run it
        Reservoir1spp r = Reservoir1spp::from_raw(u.xy);
        uint rng = u.z;
        const bool a = r.update(urand.x * 3, u.w, rng);
        const bool b = r.update(urand.y, u.w ^ 0x5555, rng);
        r.M = min(r.M, 500.0);
        r.W = min(r.W, 1000.0);     // (as_raw packs M and W as halves)
        OUT(uint4(r.as_raw(), asuint(r.w_sum), (a ? 1 : 0) | (b ? 2 : 0) | (rng << 2)));
        Reservoir1spp s = Reservoir1spp::create();
        Reservoir1sppStreamState st = Reservoir1sppStreamState::create();
        s.init_with_stream(urand.x, urand.y * 4, st, 17);
        const bool c = s.update_with_stream(r, urand.y + 0.125, 0.75, st, u.w, rng);
        s.finish_stream(st);
        OUT(uint4(asuint(s.M), asuint(s.W), asuint(s.w_sum), s.payload ^ (c ? 0x80000000 : 0)));
    }
    {   // the two lobes: evaluate and sample
        SpecularBrdf brdf;
        brdf.albedo = saturate(col);
        brdf.roughness = 0.02 + 0.96 * urand.x;
        float3 wo = uniform_sample_hemisphere(urand.yx);
        float3 wi = uniform_sample_hemisphere(float2(uint_to_u01_float(u.z), uint_to_u01_float(u.w)));
        BrdfValue v = brdf.evaluate(wo, wi);
        OUT(uint4(asuint(v.value), asuint(v.pdf)));
        OUT(uint4(asuint(v.value_over_pdf), asuint(v.transmission_fraction.x)));
        BrdfSample s = brdf.sample(wo, float2(uint_to_u01_float(u.w), uint_to_u01_float(u.z)));
        OUT(uint4(asuint(s.wi), asuint(s.pdf)));
        OUT(uint4(asuint(s.value_over_pdf), asuint(s.value.y)));
        DiffuseBrdf diffuse;
        diffuse.albedo = saturate(col);
        BrdfSample d = diffuse.sample(wo, urand);
        OUT(uint4(asuint(d.wi), asuint(diffuse.evaluate(wo, wi).value.z)));
    }
}
