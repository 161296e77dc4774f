// ORACLE (test infrastructure). Reservoir1spp restated from inc/reservoir.hlsl:6-98.
#pragma once
#include "okj_math.hpp"

namespace okj {

// inc/reservoir.hlsl:6-98
struct StreamState { float p_q_sel = 0, M_sum = 0; };
struct Reservoir1spp {
    float w_sum = 0; uint32_t payload = 0; float M = 0, W = 0;
    static Reservoir1spp from_raw(u2 raw) {
        Reservoir1spp r;
        r.payload = raw.x;
        f2 mw = unpack_2x16f_uint(raw.y);
        r.M = mw.x; r.W = mw.y;
        return r;
    }
    u2 as_raw() const { return u2{payload, pack_2x16f_uint(M, fmaxf(0.0f, W))}; }
    bool update(float w, uint32_t sample_payload, uint32_t& rng) {
        w_sum += w;
        M += 1;
        const float dart = uint_to_u01_float(hash1_mut(rng));
        const float prob = w / w_sum;
        if (prob >= dart) { payload = sample_payload; return true; }
        return false;
    }
    bool update_with_stream(const Reservoir1spp& r, float p_q, float weight, StreamState& ss, uint32_t sample_payload, uint32_t& rng) {
        ss.M_sum += r.M;
        if (update(p_q * weight * r.W * r.M, sample_payload, rng)) { ss.p_q_sel = p_q; return true; }
        return false;
    }
    void init_with_stream(float p_q, float weight, StreamState& ss, uint32_t sample_payload) {
        payload = sample_payload;
        w_sum = p_q * weight;
        M = weight != 0 ? 1.0f : 0.0f;
        W = weight;
        ss.p_q_sel = p_q;
        ss.M_sum = M;
    }
    void finish_stream(const StreamState& ss) {
        M = ss.M_sum;
        W = w_sum / (fmaxf(1e-8f, M * ss.p_q_sel));
    }
};


} // namespace okj
