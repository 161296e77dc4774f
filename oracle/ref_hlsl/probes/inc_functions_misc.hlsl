// TEST INFRASTRUCTURE (ours, not the reference's): the fourth probe pass -- small helpers that live next to the passes rather than under inc/: taa/taa_common.hlsl
// (decode_rgb / encode_rgb), inc/bilinear.hlsl (get_bilinear_filter), rtdgi/rtdgi_common.hlsl (TemporalReservoirOutput), ircache/ircache_sampler_common.inc.hlsl (SampleParams)
// and ircache/ircache_grid.hlsl (ws_pos_to_ircache_coord under the frame's cascades), included where they lie. tests/test_ref_hlsl.py compares every row with the oracle's
// restatement (oracle/okj_api.cpp: okj_probe_functions_misc), bit for bit.
#include "../inc/frame_constants.hlsl"
#include "../inc/hash.hlsl"
#include "../inc/math.hlsl"
#include "../inc/pack_unpack.hlsl"
#include "../inc/quasi_random.hlsl"
#include "../inc/bilinear.hlsl"
#include "../taa/taa_common.hlsl"
#include "../rtdgi/rtdgi_common.hlsl"
#include "../ircache/ircache_grid.hlsl"
#include "../ircache/ircache_sampler_common.inc.hlsl"

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
    const float4 f = asfloat(u);
    const float3 unit = normalize(f.xyz);
    const float3 col = abs(f.xyz);
    const float3 ucol = float3(uint_to_u01_float(u.x), uint_to_u01_float(u.y), uint_to_u01_float(u.z));
    uint k = 0;
    #define OUT(v) probe_out[(k++) * probe_count + i] = (v)
    OUT(uint4(asuint(decode_rgb(col)), 0));
    OUT(uint4(asuint(encode_rgb(col)), 0));
    {
        const Bilinear b = get_bilinear_filter(ucol.xy * 1.25 - 0.125, float2(1920, 1080));
        OUT(uint4(asuint(b.origin), asuint(b.weights)));
    }
    {
        const TemporalReservoirOutput t = TemporalReservoirOutput::from_raw(u);
        OUT(t.as_raw());
        OUT(uint4(asuint(t.depth), asuint(t.ray_hit_offset_ws)));
        OUT(uint4(asuint(t.luminance), asuint(t.hit_normal_ws)));
    }
    {
        const SampleParams s = SampleParams::from_spf_entry_sample_frame(4, u.x & 0xffff, u.y & 3, u.z & 0xffff);
        OUT(uint4(s.raw(), s.rng(), asuint(s.octa_uv())));
        OUT(uint4(asuint(s.direction()), s.octa_idx()));
    }
    {
        // positions from centimetres to kilometres around the grid centre
        const float3 pos = frame_constants.ircache_grid_center.xyz + f.xyz * 0.01;
        const IrcacheCoord c = ws_pos_to_ircache_coord(pos, unit, ucol - 0.5);
        OUT(uint4(c.coord, c.cascade));
        OUT(uint4(c.cell_idx(), ws_local_pos_to_cascade_idx(f.xyz * 0.01, 1), asuint(ircache_grid_cell_diameter_in_cascade(u.w % 12)), 0));
    }
}
