// Reservoir1spp (inc/reservoir.hlsl:6-98) and TemporalReservoirOutput (rtdgi/rtdgi_common.hlsl:12-39), device side.
#pragma once
#include "kj_vec.hpp"

namespace kj {

struct StreamState { float p_q_sel, M_sum; };
struct Reservoir1spp {
    float w_sum; uint32_t payload; float M, W;
    KJ_D static Reservoir1spp create() { return Reservoir1spp{0, 0, 0, 0}; }
    KJ_D static Reservoir1spp from_raw(uint2 raw) { V2 mw = unpack_2x16f_uint(raw.y); return Reservoir1spp{0, raw.x, mw.x, mw.y}; }
    KJ_D uint2 as_raw() const { return make_uint2(payload, pack_2x16f_uint(M, fmaxf(0.0f, W))); }
    KJ_D bool update(float w, uint32_t sample_payload, uint32_t& rng) {
        w_sum += w;
        M += 1;
        const float dart = uint_to_u01_float(hash1_mut(rng));
        const float prob = w / w_sum;
        if (prob >= dart) { payload = sample_payload; return true; }
        return false;
    }
    KJ_D bool update_with_stream(const Reservoir1spp& r, float p_q, float weight, StreamState& ss, uint32_t sample_payload, uint32_t& rng) {
        ss.M_sum += r.M;
        if (update(p_q * weight * r.W * r.M, sample_payload, rng)) { ss.p_q_sel = p_q; return true; }
        return false;
    }
    KJ_D void init_with_stream(float p_q, float weight, StreamState& ss, uint32_t sample_payload) {
        payload = sample_payload;
        w_sum = p_q * weight;
        M = weight != 0 ? 1.0f : 0.0f;
        W = weight;
        ss.p_q_sel = p_q;
        ss.M_sum = M;
    }
    KJ_D void finish_stream(const StreamState& ss) {
        M = ss.M_sum;
        W = w_sum / (fmaxf(1e-8f, M * ss.p_q_sel));
    }
};
// rtdgi_common.hlsl:12-39
struct TemporalReservoirOutput {
    float depth; V3 ray_hit_offset_ws; float luminance; V3 hit_normal_ws;
    KJ_D static TemporalReservoirOutput from_raw(uint4 raw) {
        V2 a = unpack_2x16f_uint(raw.y), b = unpack_2x16f_uint(raw.z);
        return TemporalReservoirOutput{asfloat(raw.x), V3{a.x, a.y, b.x}, b.y, unpack_normal_11_10_11(raw.w)};
    }
    KJ_D uint4 as_raw() const {
        return make_uint4(asuint(depth), pack_2x16f_uint(ray_hit_offset_ws.x, ray_hit_offset_ws.y), pack_2x16f_uint(ray_hit_offset_ws.z, luminance),
                          pack_normal_11_10_11(hit_normal_ws));
    }
};


} // namespace kj
