// The gfx950 design of rtdgi's screen-space resampling chain: ReSTIR spatial reuse (restir_spatial.hlsl:48-372 +
// occlusion_raymarch.hlsl), the half->full resolve (restir_resolve.hlsl:42-205) and the GI spatial filter
// (spatial_filter.hlsl:34-101). Same results as the reference's shaders (tests/test_gpu_parity.py compares every surface
// with the CPU oracle), different program:
//
//  * G-buffer tiles in LDS. Every tap of these passes needs {depth, view normal, ssao} of a neighbour. k_extract_half also
//    writes them as ONE 8-byte record per half-res pixel; a workgroup stages the records of its pixels + the pass's reach
//    (32 / 16 px for spatial reuse, 3 px for the resolve) into LDS with coalesced row loads and every tap is one ds_read_b64.
//    What stays a global gather is what is genuinely data dependent: the neighbour's reservoir and the sample it points at.
//  * One wave = one 8x8 pixel tile. The spiral's angle offset is constant per 8x8 (first pass) or 4x4 (later passes) block
//    (restir_spatial.hlsl:113-123), so the cos/sin of all taps of all blocks of a wave are evaluated ONCE, one (tap, block) per
//    lane, and handed to the lanes through v_readlane / ds_bpermute instead of being recomputed by every lane for every tap.
//  * Wave votes skip the occlusion ray-march when no lane's sample can contribute (zero target function or weight) and keep
//    its loop wave-uniform.
//  * Positions come from FrameDerived (one affine 4x4 + one v_rcp per position instead of ViewRayContext's six mat4 products),
//    weights use v_rcp / v_rsq / v_sqrt / v_exp directly (kj_screen.hpp).
#include "kj_host.hpp"
#include "kj_screen.hpp"
#include "kj_reservoir.hpp"
#include "rtdgi_resample.hpp"

using namespace kj;

#define RESTIR_TEMPORAL_M_CLAMP 20.0f
#define RESTIR_RESERVOIR_W_CLAMP 10.0f
#define SSGI_NEAR_FIELD_RADIUS 80.0f

// workgroup -> tile order of the resampling passes (kj_vec.hpp: tile_order), chosen per launch by its width in tiles (round 6, profiles/r06_screen_passes.md): with
// the plain order every XCD's L2 pulled the whole working set (restir spatial: 148 / 96 MB per 1080p launch for 21 MB of algorithmic bytes). Column bands cut that
// 4-7x and are faster where there are at least two bands per XCD (4K: -10 % / -4 % for the second spatial pass / the resolve) but slower at 1080p (one band per
// XCD: the busiest band sets the pace); 4 x 4 super-tiles dealt round-robin cut it 3-4x at equal time there. -DKJ_SPATIAL_TILES=n / -DKJ_RESOLVE_TILES=n force one order.
KJ_HD int resample_tile_order(int tiles_x) { return tiles_x >= 96 ? KJ_TILES_BANDS : KJ_TILES_SUPER; }

namespace {

KJ_D V3 unpack_view_normal(uint32_t p) {   // RGBA8_SNORM xyz
    return V3{from_snorm8(int8_t(p & 0xff)), from_snorm8(int8_t((p >> 8) & 0xff)), from_snorm8(int8_t((p >> 16) & 0xff))};
}
KJ_D V3 rotate_view_to_world(const FrameConstants& fc, V3 v) {   // direction_view_to_world: upper 3x3
    const float* m = fc.view_constants.view_to_world;
    return V3{m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z};
}
KJ_D V3 rotate_world_to_view(const FrameConstants& fc, V3 v) {
    const float* m = fc.view_constants.world_to_view;
    return V3{m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z};
}
KJ_D float normal_influence(float x, float b) { return x < -b ? 0.0f : (x + b) * (x + b) / (4 * b); }

struct PackedSample { float depth; V3 hit_offset_ws; float luminance; V3 hit_normal_ws; };
KJ_D PackedSample unpack_sample(uint4 raw) {   // TemporalReservoirOutput::from_raw
    const V2 a = unpack_2x16f_uint(raw.y), b = unpack_2x16f_uint(raw.z);
    return PackedSample{asfloat(raw.x), V3{a.x, a.y, b.x}, b.y, normalize_fast(unpack_normal_11_10_11_no_normalize(raw.w))};
}

}  // namespace

// ------------------------------------------------------------------ restir_spatial.hlsl:48-372
struct SpatialArgs {
    const FrameConstants* __restrict__ fc;
    Img<uint2> reservoir_input_tex;       // RG32UI
    Img<uint2> half_gbuf;                 // {depth bits, view normal snorm8 x3 | ssao snorm8 << 24} (k_extract_half)
    Img<float> half_depth_tex;            // the ray-march's taps outside the staged tile
    Img<uint4> temporal_reservoir_packed_tex;
    Img<uint2> reservoir_output_tex;
    int W, H;                             // full-res extent
    uint32_t pass_idx, perform_occlusion_raymarch, occlusion_raymarch_importance_only;
    int row0, row1;
};

// R = reach of the pass in half-res pixels (32: first pass, 16: later passes), SAMPLES = 8 / 5, BW x BH = workgroup in pixels
// (multiples of 8: one wave per 8x8 block), TILE = stage the G-buffer records in LDS (else gather them from HBM/L2).
#ifndef KJ_MARCH_BATCH
#define KJ_MARCH_BATCH 6      // the second pass' march: its six depth taps requested together (0: the rolled loop)
#endif
#ifndef KJ_SPATIAL_BATCH
#define KJ_SPATIAL_BATCH 0      // taps of restir spatial whose gathers are in flight together: 0 = the measured choice per pass (below), 1 = the shader text's order everywhere
#endif
// MARCH: the launch performs the occlusion march (the last spatial pass). A pass that does not is built without the march's code and registers (KJ_SPATIAL_MARCH_SPLIT).
template <int R, int SAMPLES, int BW, int BH, bool TILE, int ORDER, bool MARCH>
__global__ void __launch_bounds__(BW * BH) k_restir_spatial(SpatialArgs a) {
    constexpr int TW = BW + 2 * R, TH = BH + 2 * R;
    __shared__ uint2 tile[TILE ? TW * TH : 1];
    const FrameConstants& fc = *a.fc;
    const FrameDerived& fd = frame_derived(a.fc);
    const int hw = a.reservoir_output_tex.w, hh = a.reservoir_output_tex.h;
    const int tid = int(threadIdx.x), wave = tid >> 6, lane = tid & 63;
    const uint2 tb = tile_order<ORDER, 4>();
    const int bx0 = int(tb.x) * BW, by0 = a.row0 + int(tb.y) * BH;
    const int tx0 = bx0 - R, ty0 = by0 - R;
    if (TILE) {
        for (int i = tid; i < TW * TH; i += BW * BH) {
            const int ty = i / TW, tx = i - ty * TW;
            tile[i] = a.half_gbuf.ld(tx0 + tx, ty0 + ty);
        }
    }
    const int ox = bx0 + (wave % (BW / 8)) * 8, oy = by0 + (wave / (BW / 8)) * 8;   // this wave's 8x8 block
    const int x = ox + (lane & 7), y = oy + (lane >> 3);
    const uint32_t pass_idx = a.pass_idx;
    const uint32_t ang_seed = fc.frame_index * 2u + pass_idx;
    // ---- spiral directions, one (tap, angle block) per lane
#if KJ_WAVE_SHARED
    float tap_cos[SAMPLES], tap_sin[SAMPLES];
    {
        const int tap = lane & 7, blk = (lane >> 3) & 3;
        const uint32_t hx = R == 32 ? uint32_t(ox) >> 3 : (uint32_t(ox) >> 2) + uint32_t(blk & 1);
        const uint32_t hy = R == 32 ? uint32_t(oy) >> 3 : (uint32_t(oy) >> 2) + uint32_t(blk >> 1);
        const float ang_offset = uint_to_u01_float(hash3(hx, hy, ang_seed)) * KJ_PI * 2;
        const V2 cs = cos_sin_turns_fast((float(tap) + ang_offset) * KJ_GOLDEN_ANGLE);
        const int my_blk = R == 32 ? 0 : ((lane >> 2) & 1) + 2 * ((lane >> 5) & 1);
#pragma unroll
        for (int i = 0; i < SAMPLES; ++i) { tap_cos[i] = wave_read(cs.x, i + 8 * my_blk); tap_sin[i] = wave_read(cs.y, i + 8 * my_blk); }
    }
#endif
    if (TILE) __syncthreads();
    const bool in_image = x < hw && y < (hh < a.row1 ? hh : a.row1);
    if (!in_image) return;

    const I2 off = halfres_subsample_offset(fc.frame_index);
    const HalfPxToCs px_to_cs = HalfPxToCs::make(a.W, a.H, off);
    auto gbuf_at = [&](int px, int py) -> uint2 {
        if (TILE) return tile[(py - ty0) * TW + (px - tx0)];      // |tap offset| <= R by construction
        return a.half_gbuf.ld(px, py);
    };
    const uint2 cg = gbuf_at(x, y);
    const float depth = asfloat(cg.x);
    uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index + pass_idx * 123u);
    const V4 gts = tex_size4(a.W, a.H);
    const V2 uv = get_uv(float(x * 2 + off.x), float(y * 2 + off.y), gts);
    const V2 c_cs = uv_to_cs(uv);
    const V3 c_ws = hit_ws_from_cs(fd, c_cs, depth);
    const V3 c_vs = hit_vs_from_cs(fc, c_cs, depth);
    const V3 center_normal_vs = unpack_view_normal(cg.y);
    const V3 center_normal_ws = rotate_view_to_world(fc, center_normal_vs);
    const float center_ssao = from_snorm8(int8_t(cg.y >> 24));
    StreamState stream_state{0, 0};
    Reservoir1spp reservoir = Reservoir1spp::create();
    const float sample_radius_offset = uint_to_u01_float(hash1_mut(rng));
    const Reservoir1spp center_r = Reservoir1spp::from_raw(a.reservoir_input_tex.ld(x, y));
    float kernel_tightness = 1.0f - center_ssao;
    const float MAX_INPUT_M_IN_PASS = pass_idx == 0 ? RESTIR_TEMPORAL_M_CLAMP : RESTIR_TEMPORAL_M_CLAMP * 8.0f;
    kernel_tightness = lerp(kernel_tightness, 1.0f, 0.5f * smoothstep(MAX_INPUT_M_IN_PASS * 0.5f, MAX_INPUT_M_IN_PASS, center_r.M));
    float max_kernel_radius = pass_idx == 0 ? lerp(32.0f, 12.0f, kernel_tightness) : lerp(16.0f, 6.0f, kernel_tightness);
    if (pass_idx >= 2) max_kernel_radius = 8;
    const V2 dist_to_edge_xy = vmin(V2{float(x), float(y)}, V2{float(hw) - float(x), float(hh) - float(y)});
    const float allow_edge_overstep = center_r.M < 10 ? 100.0f : 1.25f;
    const V2 kernel_radius = vmin(V2{max_kernel_radius, max_kernel_radius}, dist_to_edge_xy * allow_edge_overstep);
    const float depth_gate = pass_idx == 0 ? 0.15f : 0.1f;
    const float nz = fmaxf(0.3f, center_normal_vs.z);
#if !KJ_WAVE_SHARED
    const uint32_t shift = pass_idx == 0 ? 3u : 2u;
    const float ang_offset = uint_to_u01_float(hash3(uint32_t(x) >> shift, uint32_t(y) >> shift, ang_seed)) * KJ_PI * 2;
#endif

    // One tap of the text's loop once its three gathers are known: the neighbour's reservoir (non-empty), its G-buffer record (not sky) and the packed sample the
    // reservoir points at. Everything the shader does with them, in its order; a `return` is the shader's `continue`.
    auto tap = [&](const int sample_i, const int rx, const int ry, const uint2 reservoir_raw, const uint2 rg, const uint4 sp_raw) {
        const bool is_center_sample = sample_i == 0;
        const float rpx_depth = asfloat(rg.x);
        Reservoir1spp r = Reservoir1spp::from_raw(reservoir_raw);
        r.M = fminf(r.M, 500.0f);
        const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
        const PackedSample sp = unpack_sample(sp_raw);
        float relevance = 1;
        relevance *= normal_influence(dot(unpack_view_normal(rg.y), center_normal_vs), 0.5f) * (1.0f / 1.125f);
        relevance *= 1 - fabsf(from_snorm8(int8_t(rg.y >> 24)) - center_ssao);
        const V3 rpx_ws = hit_ws_from_cs(fd, px_to_cs(rx, ry), rpx_depth);
        const V2 spx_cs = px_to_cs(spx_x, spx_y);
        const V3 sample_hit_ws = sp.hit_offset_ws + hit_ws_from_cs(fd, spx_cs, sp.depth);
        const V3 reused_unnorm = sample_hit_ws - rpx_ws, dir_unnorm = sample_hit_ws - c_ws;
        const float reused_d2 = dot(reused_unnorm, reused_unnorm), d2 = dot(dir_unnorm, dir_unnorm);
        const float inv_reused = rsq_fast(reused_d2), inv_dist = rsq_fast(d2);
        const V3 dir_to_sample_hit = dir_unnorm * inv_dist;
        if (!is_center_sample) {
            const float depth_diff = fabsf(nz * (depth * rcp_fast(rpx_depth) - 1.0f));
            relevance *= 1 - smoothstep_fast(0.0f, depth_gate, depth_diff);
        }
        float p_q = sp.luminance * fmaxf(0.0f, dot(dir_to_sample_hit, center_normal_ws));
        if (!(p_q >= 0)) return;
        r.M *= relevance;
        float jacobian = 1;
        if (!is_center_sample) {
            const float center_to_hit_vis = -dot(sp.hit_normal_ws, dir_to_sample_hit);
            const float reused_to_hit_vis = -dot(sp.hit_normal_ws, reused_unnorm) * inv_reused;
            const float dist_ratio = (reused_d2 * inv_reused) * inv_dist;     // reused_dist / dist_to_sample_hit
            jacobian = sqrt_fast(dist_ratio * dist_ratio * clampf(center_to_hit_vis * rcp_fast(reused_to_hit_vis), 0.0f, 1e4f));
        }
        float visibility = 1;
        if (MARCH && a.perform_occlusion_raymarch) {
            // the march only scales this sample's weight: skipped (wave-wide when possible) for samples whose weight is zero anyway
            const bool contributes = p_q > 0 && r.M != 0 && r.W != 0;
            if (wave_any(contributes)) {
                // occlusion_raymarch.hlsl:75-146 on the half-res depth, no colour bounce
                const float surface_offset_len = length_fast(hit_vs_from_cs(fc, spx_cs, depth) - c_vs);
                const V3 end_ws = c_ws + dir_unnorm * fminf(1.0f, 3.0f * surface_offset_len * inv_dist);
                const V3 end_cs = world_to_clip_fast(fd, end_ws);
                const V2 fullres{float(a.W), float(a.H)}, halfres{float(hw), float(hh)};
                const V2 len_px = (cs_to_uv(V2{end_cs.x, end_cs.y}) - uv) * halfres;
                const int k_count = contributes ? min(6, int(floorf(length_fast(len_px) * 0.5f))) : 0;
                const float depth_step_per_z = (end_cs.z - depth) * rcp_fast(length_fast(V2{end_cs.x - c_cs.x, end_cs.y - c_cs.y}));
                const float t_step = rcp_fast(float(k_count));
                float t = 0.5f * t_step;
#if KJ_MARCH_BATCH
                // The march's depth taps -- their addresses depend on nothing loaded -- requested KJ_MARCH_BATCH at a time, the text's arithmetic behind them in the text's
                // order (a lane past its k_count reads texel 0 and ignores it). In the 5-tap kernel without LDS tile only (94 -> 117 VGPRs, four waves per SIMD): pass 1
                // 41.1 -> 39.9 us at 1080p, 129 -> 123.6 us at 4K. While every build carried the march this cost the FIRST pass a wave and was not adopted; with the march
                // out of the first pass' build (MARCH, above: 121 -> 86 VGPRs, 26.2 -> 25.6 us / 85.0 -> 79.0 us) it pays (profiles/r06_screen_passes.md).
                if (!TILE && SAMPLES == 5) {
#pragma unroll
                    for (int k0 = 0; k0 < 6; k0 += KJ_MARCH_BATCH) {
                        if (!wave_any(k0 < k_count)) break;
                        float m_depth[KJ_MARCH_BATCH], m_iz[KJ_MARCH_BATCH], m_bz[KJ_MARCH_BATCH]; bool m_in[KJ_MARCH_BATCH];
#pragma unroll
                        for (int j = 0; j < KJ_MARCH_BATCH; ++j) if (k0 + j < 6) {
                            const V3 interp_cs = lerp(V3{c_cs.x, c_cs.y, depth}, end_cs, t);
                            const V2 uv_at = cs_to_uv(V2{interp_cs.x, interp_cs.y});
                            const float fpx = floorf(uv_at.x * fullres.x - float(off.x)), fpy = floorf(uv_at.y * fullres.y - float(off.y));
                            const uint32_t ux = fpx > 0 ? uint32_t(fpx) : 0u, uy = fpy > 0 ? uint32_t(fpy) : 0u;
                            const uint32_t pxi = (ux & ~1u) + uint32_t(off.x), pyi = (uy & ~1u) + uint32_t(off.y);
                            const int hx = int(pxi >> 1u), hy = int(pyi >> 1u);
                            m_in[j] = k0 + j < k_count && a.half_depth_tex.inb(hx, hy);
                            m_depth[j] = a.half_depth_tex.p[m_in[j] ? size_t(hy) * a.half_depth_tex.w + hx : size_t(0)];
                            const V2 qcs = uv_to_cs(V2{(float(pxi) + 0.5f) * gts.z, (float(pyi) + 0.5f) * gts.w});
                            m_bz[j] = depth + depth_step_per_z * length_fast(qcs - c_cs);
                            m_iz[j] = interp_cs.z;
                            t += t_step;
                        }
#pragma unroll
                        for (int j = 0; j < KJ_MARCH_BATCH; ++j) if (k0 + j < 6) {
                            const float depth_at = m_in[j] ? m_depth[j] : 0.0f;
                            if (k0 + j < k_count && depth_at > m_bz[j]) {
                                const float depth_diff = fabsf(fmaxf(1e-20f, m_iz[j]) * rcp_fast(fmaxf(1e-20f, depth_at)) - 1.0f);
                                visibility *= 1 - smoothstep_fast(0.05f, 0.025f, depth_diff);
                            }
                        }
                    }
                } else
#endif
#pragma unroll 1
                for (int k = 0; k < 6; ++k) {
                    const bool active = k < k_count;
                    if (!wave_any(active)) break;
                    if (active) {
                        const V3 interp_cs = lerp(V3{c_cs.x, c_cs.y, depth}, end_cs, t);
                        const V2 uv_at = cs_to_uv(V2{interp_cs.x, interp_cs.y});
                        const float fpx = floorf(uv_at.x * fullres.x - float(off.x)), fpy = floorf(uv_at.y * fullres.y - float(off.y));
                        const uint32_t ux = fpx > 0 ? uint32_t(fpx) : 0u, uy = fpy > 0 ? uint32_t(fpy) : 0u;
                        const uint32_t pxi = (ux & ~1u) + uint32_t(off.x), pyi = (uy & ~1u) + uint32_t(off.y);
                        const int hx = int(pxi >> 1u), hy = int(pyi >> 1u);
                        float depth_at;
                        const bool staged = TILE && uint32_t(hx - tx0) < uint32_t(TW) && uint32_t(hy - ty0) < uint32_t(TH);
                        if (staged) depth_at = asfloat(tile[(hy - ty0) * TW + (hx - tx0)].x);
                        else depth_at = a.half_depth_tex.ld(hx, hy);
                        const V2 qcs = uv_to_cs(V2{(float(pxi) + 0.5f) * gts.z, (float(pyi) + 0.5f) * gts.w});
                        const float biased_z = depth + depth_step_per_z * length_fast(qcs - c_cs);
                        if (depth_at > biased_z) {
                            const float depth_diff = fabsf(fmaxf(1e-20f, interp_cs.z) * rcp_fast(fmaxf(1e-20f, depth_at)) - 1.0f);
                            visibility *= 1 - smoothstep_fast(0.05f, 0.025f, depth_diff);
                        }
                        t += t_step;
                    }
                }
            }
        }
        if (a.occlusion_raymarch_importance_only) { p_q *= lerp(0.25f, 1.0f, visibility); visibility = 1; }
        // Reservoir1spp::update_with_stream
        stream_state.M_sum += r.M;
        const float w = p_q * (visibility * jacobian) * r.W * r.M;
        reservoir.w_sum += w;
        reservoir.M += 1;
        const float dart = uint_to_u01_float(hash1_mut(rng));
        if (w * rcp_fast(reservoir.w_sum) >= dart) { reservoir.payload = r.payload; stream_state.p_q_sel = p_q; }
    };
    auto tap_pixel = [&](const int sample_i, int& rx, int& ry) {
#if KJ_WAVE_SHARED
        const V2 cs_ang{tap_cos[sample_i], tap_sin[sample_i]};
#else
        const V2 cs_ang = cos_sin_turns_fast((float(sample_i) + ang_offset) * KJ_GOLDEN_ANGLE);
#endif
        const V2 radius = sample_i == 0 ? V2{0, 0} : sqrt_fast((float(sample_i) + sample_radius_offset) * (1.0f / float(SAMPLES))) * kernel_radius;
        rx = x + int(cs_ang.x * radius.x); ry = y + int(cs_ang.y * radius.y);
    };
    // Memory-level parallelism (round 6, profiles/r06_screen_passes.md): a tap is a chain of three dependent gathers, and the loop as the shader text has it issues
    // them one at a time with a `continue` between them: 24 round trips to memory for the 8 taps of the first pass. With BATCH > 1 the taps of a batch put their loads
    // in flight TOGETHER -- every tap's reservoir + record (their addresses depend on nothing loaded), then every packed sample -- and the arithmetic follows tap by tap
    // in the text's order: same operations on the same values. Measured (MI355X, profiles/r06/gathers_in_flight_call11_summary.txt): the first pass (8 taps, no march) in
    // batches of 4 -- 121 VGPRs, four waves per SIMD instead of five -- 32.9 -> 26.7 us at 1080p, 97.3 -> 86.5 us at 4K; batches of 8 (155 VGPRs) 39.8 us; the second
    // pass, whose taps spend their time in the occlusion march, loses 7-9 % with any batching and keeps the text's order (BATCH = 1).
    constexpr int BATCH = KJ_SPATIAL_BATCH == 0 ? (SAMPLES == 8 ? 4 : 1) : (SAMPLES < KJ_SPATIAL_BATCH ? SAMPLES : KJ_SPATIAL_BATCH);
    if (BATCH == 1) {
#pragma unroll
        for (int sample_i = 0; sample_i < SAMPLES; ++sample_i) {
            int rx, ry;
            tap_pixel(sample_i, rx, ry);
            const uint2 reservoir_raw = a.reservoir_input_tex.ld(rx, ry);
            if (0 == reservoir_raw.x) continue;
            const uint2 rg = gbuf_at(rx, ry);
            if (asfloat(rg.x) == 0.0f) continue;
            tap(sample_i, rx, ry, reservoir_raw, rg, a.temporal_reservoir_packed_tex.ld(int(reservoir_raw.x & 0xffff), int(reservoir_raw.x >> 16)));
        }
    } else {
#pragma unroll
        for (int base = 0; base < SAMPLES; base += BATCH) {
            uint32_t tap_xy[BATCH];      // rx + 0x8000 | (ry + 0x8000) << 16
            uint2 tap_res[BATCH], tap_g[BATCH];
            uint4 tap_sp[BATCH];
            bool tap_in[BATCH], tap_sp_in[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (base + j >= SAMPLES) break;
                int rx, ry;
                tap_pixel(base + j, rx, ry);
                tap_xy[j] = uint32_t(rx + 0x8000) | (uint32_t(ry + 0x8000) << 16);
                tap_res[j] = a.reservoir_input_tex.ld_raw(rx, ry, tap_in[j]);
                if (TILE) tap_g[j] = gbuf_at(rx, ry);
                else { bool in_; tap_g[j] = a.half_gbuf.ld_raw(rx, ry, in_); }      // (same extent as the reservoir image: one in-bounds flag serves both)
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (base + j >= SAMPLES) break;
                if (!tap_in[j]) { tap_res[j] = make_uint2(0u, 0u); if (!TILE) tap_g[j] = make_uint2(0u, 0u); }
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (base + j >= SAMPLES) break;
                const uint32_t payload = tap_res[j].x;      // an empty reservoir's payload is pixel (0, 0): a valid address; the value is not used
                tap_sp[j] = a.temporal_reservoir_packed_tex.ld_raw(int(payload & 0xffff), int(payload >> 16), tap_sp_in[j]);
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (base + j >= SAMPLES) break;
                if (0 == tap_res[j].x || asfloat(tap_g[j].x) == 0.0f) continue;
                tap(base + j, int(tap_xy[j] & 0xffffu) - 0x8000, int(tap_xy[j] >> 16) - 0x8000, tap_res[j], tap_g[j], tap_sp_in[j] ? tap_sp[j] : make_uint4(0u, 0u, 0u, 0u));
            }
        }
    }
    reservoir.finish_stream(stream_state);
    reservoir.W = fminf(reservoir.W, RESTIR_RESERVOIR_W_CLAMP);
    a.reservoir_output_tex.st(x, y, reservoir.as_raw());
}

// ------------------------------------------------------------------ restir_resolve.hlsl:42-205
// A 16x16 full-res workgroup reconstructs from an 8x8 block of half-res pixels + 3 px of reach. Everything a tap reads AT the
// tap position is staged per half-res texel once: its candidate's hit point in world space (position reconstruction + hit
// offset: 8 taps x 256 pixels -> 196 texels), depth, view normal, the full-res ssao of its representative pixel, the candidate
// radiance and the reservoir. The spiral's four directions are evaluated once per pixel and serve both loops (the second loop's
// radii are the first's times kernel_scale, exactly). What the reservoir points at (sample position, radiance, normal) is
// fetched from memory.
struct ResolveArgs2 {
    const FrameConstants* __restrict__ fc;
    Img<uint2> radiance_tex, reservoir_input_tex; Img<uint4> gbuffer_tex; Img<float> depth_tex; Img<uint2> half_gbuf; Img<uint8_t> ssao_tex;
    Img<uint2> candidate_radiance_tex, candidate_hit_tex; Img<uint4> temporal_reservoir_packed_tex; Img<uint2> irradiance_output_tex;
    const uint32_t* __restrict__ blue_noise;
    int row0, row1;
};
KJ_D float ggx_ndf_unnorm_fast(float a2, float cos_theta) { const float d = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f; return a2 * rcp_fast(d * d); }

template <int ORDER>
__global__ void __launch_bounds__(256) k_restir_resolve(ResolveArgs2 a) {
    constexpr int TW = 14, HALO = 3;
    __shared__ float4 t_hit[TW * TW];    // candidate hit point (world) , half-res depth
    __shared__ uint2 t_rad[TW * TW];     // candidate radiance RGBA16F
    __shared__ uint2 t_res[TW * TW];     // reservoir
    __shared__ uint32_t t_nrm[TW * TW];  // view normal snorm8 x3 | full-res ssao unorm8 << 24
    const FrameConstants& fc = *a.fc;
    const FrameDerived& fd = frame_derived(a.fc);
    const int W = a.irradiance_output_tex.w, H = a.irradiance_output_tex.h;
    const int tid = int(threadIdx.x), wave = tid >> 6, lane = tid & 63;
    const uint2 tb = tile_order<ORDER, 4>();
    const int bx0 = int(tb.x) * 16, by0 = a.row0 + int(tb.y) * 16;
    const int hx0 = (bx0 >> 1) - HALO, hy0 = (by0 >> 1) - HALO;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const HalfPxToCs px_to_cs = HalfPxToCs::make(W, H, off);
    // (round 6: every load of the prologue -- this pixel's own depth / G-buffer / ssao and the staged texel's five images -- is issued before the first one is used:
    // one round trip to memory instead of eight; the out-of-bounds selects of Img::ld follow the loads. Same values, same arithmetic.)
    const int x = bx0 + (wave & 1) * 8 + (lane & 7), y = by0 + (wave >> 1) * 8 + (lane >> 3);
    bool c_in_d, c_in_g, c_in_s;
    const float depth_raw = a.depth_tex.ld_raw(x, y, c_in_d);
    const uint4 gbuffer_raw = a.gbuffer_tex.ld_raw(x, y, c_in_g);
    const uint8_t ssao_raw = a.ssao_tex.ld_raw(x, y, c_in_s);
    if (tid < TW * TW) {
        const int ty = tid / TW, tx = tid - ty * TW;
        const int px = hx0 + tx, py = hy0 + ty;
        bool in_h, in_f;      // the half-res images share one extent, the full-res ssao has its own
        uint2 g = a.half_gbuf.ld_raw(px, py, in_h);
        uint2 ch = a.candidate_hit_tex.ld_raw(px, py, in_h);
        uint2 cr = a.candidate_radiance_tex.ld_raw(px, py, in_h);
        uint2 rs = a.reservoir_input_tex.ld_raw(px, py, in_h);
        uint8_t ao = a.ssao_tex.ld_raw(px * 2 + off.x, py * 2 + off.y, in_f);
        if (!in_h) { g = make_uint2(0u, 0u); ch = make_uint2(0u, 0u); cr = make_uint2(0u, 0u); rs = make_uint2(0u, 0u); }
        if (!in_f) ao = 0;
        const float d = asfloat(g.x);
        const V3 pos = hit_ws_from_cs(fd, px_to_cs(px, py), d);
        const V3 hit = xyz(unpack_rgba16f(ch)) + pos;
        t_hit[tid] = make_float4(hit.x, hit.y, hit.z, d);
        t_rad[tid] = cr;
        t_res[tid] = rs;
        t_nrm[tid] = (g.y & 0x00ffffffu) | (uint32_t(ao) << 24);
    }
    __syncthreads();
    if (!(x < W && y < (H < a.row1 ? H : a.row1))) return;
    const float depth = c_in_d ? depth_raw : 0.0f;
    if (0 == depth) { st4(a.irradiance_output_tex, x, y, v4(0.0f)); return; }
    const V4 gts = tex_size4(W, H);
    const V2 c_cs = uv_to_cs(get_uv(float(x), float(y), gts));
    const V3 c_ws = hit_ws_from_cs(fd, c_cs, depth);
    const float c_vs_z = hit_vs_from_cs(fc, c_cs, depth).z;
    const V3 center_normal_ws = normalize_fast(unpack_normal_11_10_11_no_normalize(c_in_g ? gbuffer_raw.y : 0u));
    const V3 center_normal_vs = rotate_world_to_view(fc, center_normal_ws);
    const float center_ssao = from_unorm8(c_in_s ? ssao_raw : uint8_t(0));
    const uint32_t px_idx_in_quad = (((uint32_t(x) & 1u) | (uint32_t(y) & 1u) * 2u) + hash1(fc.frame_index)) & 3u;
    const float blue_x = blue_noise_for_pixel(a.blue_noise, x, y, fc.frame_index).x * KJ_TAU;
    const float near_end = -c_vs_z * (SSGI_NEAR_FIELD_RADIUS * gts.w * 0.5f);
    const float near_start = near_end * 0.5f;
    const float inv_depth_scale = -200.0f * center_normal_vs.z;      // exp2(-200 |nz (d/ds - 1)|) = exp2(-|inv_depth_scale (d/ds - 1)|)
    // spiral taps: pow(float(si), 0.666) * 1.0 + 0.4 for si = 0..3
    const float tap_radius[4] = {powf(0.0f, 0.666f) + 0.4f, powf(1.0f, 0.666f) + 0.4f, powf(2.0f, 0.666f) + 0.4f, powf(3.0f, 0.666f) + 0.4f};
    V2 tap_dir[4];
#pragma unroll
    for (int si = 0; si < 4; ++si) tap_dir[si] = cos_sin_turns_fast((float(si) + blue_x) * KJ_GOLDEN_ANGLE + (float(px_idx_in_quad) / 4.0f) * KJ_TAU);
    const float hxf = float(x) * 0.5f, hyf = float(y) * 0.5f;
    V3 total_irradiance = v3(0.0f);
    bool sharpen_gi_kernel = false;
    {   // near field: this frame's candidates
        float w_sum = 0;
        V3 weighted = v3(0.0f);
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const V2 rpo = tap_dir[si] * tap_radius[si];
            const int rx = int(floorf(hxf + rpo.x)), ry = int(floorf(hyf + rpo.y));
            const int ti = (ry - hy0) * TW + (rx - hx0);
            const float4 hd = t_hit[ti];
            const V3 sample_offset = V3{hd.x, hd.y, hd.z} - c_ws;
            const float d2 = dot(sample_offset, sample_offset);
            const float inv_d = rsq_fast(d2), sample_dist = d2 * inv_d;
            const float geometric_term = 2 * fmaxf(0.0f, dot(center_normal_ws, sample_offset) * inv_d);
            const float atten = smoothstep_fast(near_end, near_start, sample_dist);
            sharpen_gi_kernel |= atten > 0.9f;
            V3 contribution = xyz(unpack_rgba16f(t_rad[ti])) * geometric_term;
            contribution *= lerp(0.0f, atten, center_ssao);
            float w = ggx_ndf_unnorm_fast(0.01f, saturate(dot(center_normal_vs, unpack_view_normal(t_nrm[ti]))));
            w *= exp2_fast(-fabsf(inv_depth_scale * (depth * rcp_fast(hd.w) - 1.0f)));
            weighted += contribution * w;
            w_sum += w;
        }
        total_irradiance += weighted * rcp_fast(fmaxf(1e-20f, w_sum));
    }
    {   // far field: the resampled reservoirs
        float w_sum = 0;
        V3 weighted = v3(0.0f);
        const float kernel_scale = sharpen_gi_kernel ? 0.5f : 1.0f;
        // the three gathers at each tap's sample pixel: all twelve in flight before the first is used (round 6)
        int tap_ti[4];
        uint4 tap_sp[4]; uint2 tap_rad[4], tap_g[4];
        bool tap_in[4];      // (the three images share the half-res extent)
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const V2 rpo = tap_dir[si] * (tap_radius[si] * kernel_scale);
            const int rx = int(floorf(hxf + rpo.x)), ry = int(floorf(hyf + rpo.y));
            tap_ti[si] = (ry - hy0) * TW + (rx - hx0);
            const uint32_t payload = t_res[tap_ti[si]].x;
            const int spx_x = int(payload & 0xffff), spx_y = int(payload >> 16);
            tap_sp[si] = a.temporal_reservoir_packed_tex.ld_raw(spx_x, spx_y, tap_in[si]);
            tap_rad[si] = a.radiance_tex.ld_raw(spx_x, spx_y, tap_in[si]);
            tap_g[si] = a.half_gbuf.ld_raw(spx_x, spx_y, tap_in[si]);
        }
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const int ti = tap_ti[si];
            const Reservoir1spp r = Reservoir1spp::from_raw(t_res[ti]);
            const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
            const PackedSample sp = unpack_sample(tap_in[si] ? tap_sp[si] : make_uint4(0u, 0u, 0u, 0u));
            const V3 hit_ws = sp.hit_offset_ws + hit_ws_from_cs(fd, px_to_cs(spx_x, spx_y), sp.depth);
            const V3 sample_offset = hit_ws - c_ws;
            const float d2 = dot(sample_offset, sample_offset);
            const float inv_d = rsq_fast(d2), sample_dist = d2 * inv_d;
            const float geometric_term = 2 * fmaxf(0.0f, dot(center_normal_ws, sample_offset) * inv_d);
            V3 radiance = xyz(unpack_rgba16f(tap_in[si] ? tap_rad[si] : make_uint2(0u, 0u)));
            radiance *= lerp(1.0f, smoothstep_fast(near_start, near_end, sample_dist), center_ssao);
            const V3 contribution = radiance * geometric_term * r.W;
            const V3 sample_normal_vs = unpack_view_normal(tap_in[si] ? tap_g[si].y : 0u);
            const uint32_t tn = t_nrm[ti];
            float w = ggx_ndf_unnorm_fast(0.01f, saturate(dot(center_normal_vs, sample_normal_vs)));
            w *= exp2_fast(-fabsf(inv_depth_scale * (depth * rcp_fast(t_hit[ti].w) - 1.0f)));
            w *= exp2_fast(-20.0f * fabsf(center_ssao - float(tn >> 24) * (1.0f / 255.0f)));
            weighted += contribution * w;
            w_sum += w;
        }
        total_irradiance += weighted * rcp_fast(fmaxf(1e-20f, w_sum));
    }
    st4(a.irradiance_output_tex, x, y, v4(total_irradiance, 1));
}

// ------------------------------------------------------------------ spatial_filter.hlsl:34-101
// Seven spiral taps of at most 16 px around every pixel whose temporal history is short. Per tap: one angle evaluation on the
// reduced polynomial, v_exp_f32 weights, and the tap radii are constants (pow(si, 0.666) for si = 1..7).
KJ_D V3 crunch(V3 v) { return v * rcp_fast(max3(v.x, v.y, v.z) + 1.0f); }
KJ_D V3 uncrunch(V3 v) { return v * rcp_fast(1.0f - max3(v.x, v.y, v.z)); }
// -DKJ_SPATIAL_FILTER_TILED=1 (round 6, measured and REJECTED: 47 -> 73 us at 1080p, 195 -> 307 us at 4K, profiles/r06_screen_passes.md): a 16x16-pixel workgroup
// stages its pixels + the taps' 16-px reach (48 x 48 texels: crunched value, depth, ssao) in LDS and every tap is two LDS reads. Bit-identical, and slower: most
// pixels of a converged image take ONE tap (sample_count = 2 for validity near 1), so nine staged texels per pixel replace one or two gathers.
#ifndef KJ_SPATIAL_FILTER_TILED
#define KJ_SPATIAL_FILTER_TILED 0
#endif
#ifndef KJ_SF_BATCH
#define KJ_SF_BATCH 1      // 0: one tap at a time, as the shader text has the loop
#endif
#ifndef KJ_SF_SIZES
#define KJ_SF_SIZES 1, 6
#endif
#define KJ_SF_B 16
#define KJ_SF_R 16
#define KJ_SF_T (KJ_SF_B + 2 * KJ_SF_R)
__global__ void __launch_bounds__(KJ_SPATIAL_FILTER_TILED ? 256 : 64) k_spatial_filter(const FrameConstants* __restrict__ fcp, Img<uint2> input_tex, Img<float> depth_tex, Img<uint8_t> ssao_tex,
                                                        Img<uint32_t> geometric_normal_tex, Img<uint2> output_tex, int row0, int row1) {
#if KJ_SPATIAL_FILTER_TILED
    __shared__ float4 t_val[KJ_SF_T * KJ_SF_T];      // crunched rgb, depth
    __shared__ uint8_t t_ssao[KJ_SF_T * KJ_SF_T];
    const int tid = int(threadIdx.x), wave = tid >> 6, lane = tid & 63;
    const uint2 tb = tile_order<KJ_TILES_BANDS>();
    const int bx0 = int(tb.x) * KJ_SF_B, by0 = row0 + int(tb.y) * KJ_SF_B;
    const int tx0 = bx0 - KJ_SF_R, ty0 = by0 - KJ_SF_R;
    const int x = bx0 + (wave & 1) * 8 + (lane & 7), y = by0 + (wave >> 1) * 8 + (lane >> 3);
    const bool in_image = x < output_tex.w && y < (output_tex.h < row1 ? output_tex.h : row1);
    // a workgroup whose every pixel is fully converged (or outside the image) copies its centres and leaves: no tile
    const V4 c = in_image ? ld4(input_tex, x, y) : v4(0.0f, 0.0f, 0.0f, 1.0f);
    const float center_validity = c.w;
    __shared__ int s_any;
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (in_image && center_validity != 1) s_any = 1;
    __syncthreads();
    if (s_any) {
        for (int i = tid; i < KJ_SF_T * KJ_SF_T; i += 256) {
            const int ty = i / KJ_SF_T, tx = i - ty * KJ_SF_T;
            const float d = depth_tex.ld(tx0 + tx, ty0 + ty);
            const V3 cv = crunch(xyz(ld4(input_tex, tx0 + tx, ty0 + ty)));
            t_val[i] = make_float4(cv.x, cv.y, cv.z, d);
            t_ssao[i] = ssao_tex.ld(tx0 + tx, ty0 + ty);
        }
        __syncthreads();
    }
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const V3 center_value = xyz(c);
    if (center_validity == 1) { st4(output_tex, x, y, v4(center_value, 1.0f)); return; }
    const int lt = (y - ty0) * KJ_SF_T + (x - tx0);
    const float center_depth = t_val[lt].w;
    const float center_ssao = from_unorm8(t_ssao[lt]);
#else
    const int lane = threadIdx.x;
    const uint2 tb = tile_order<KJ_TILES_BANDS>();
    const int x = int(tb.x) * 8 + (lane & 7), y = row0 + int(tb.y) * 8 + (lane >> 3);
    if (!(x < output_tex.w && y < (output_tex.h < row1 ? output_tex.h : row1))) return;
    const FrameConstants& fc = *fcp;
    const V4 c = ld4(input_tex, x, y);
    const float center_validity = c.w;
    const V3 center_value = xyz(c);
    if (center_validity == 1) { st4(output_tex, x, y, v4(center_value, 1.0f)); return; }
    const float center_depth = depth_tex.ld(x, y);
    const float center_ssao = from_unorm8(ssao_tex.ld(x, y));
#endif
    const float center_nz = unpack_a2r10g10b10(geometric_normal_tex.ld(x, y)).z * 2.0f - 1.0f;
    const float ang_off = float((fc.frame_index * 23u) % 32u) * KJ_TAU + interleaved_gradient_noise(x, y) * KJ_PI;
    const float MAX_RADIUS_PX = sqrtf(lerp(16.0f * 16.0f, 2.0f * 2.0f, center_validity));
    const float KERNEL_SHARPNESS = 0.666f;
    const uint32_t sample_count = min(max(uint32_t(exp2f(4.0f * square(1.0f - center_validity))), 2u), 8u);
    V4 sum = v4(crunch(center_value), 1);
    const float RADIUS_SAMPLE_MULT = MAX_RADIUS_PX / powf(7.0f, KERNEL_SHARPNESS);
    const float depth_scale = -100.0f * center_nz;
    const float tap_pow[8] = {0.0f, powf(1.0f, KERNEL_SHARPNESS), powf(2.0f, KERNEL_SHARPNESS), powf(3.0f, KERNEL_SHARPNESS), powf(4.0f, KERNEL_SHARPNESS),
                              powf(5.0f, KERNEL_SHARPNESS), powf(6.0f, KERNEL_SHARPNESS), powf(7.0f, KERNEL_SHARPNESS)};
#if KJ_SF_BATCH && !KJ_SPATIAL_FILTER_TILED
    // The seven taps in batches (round 6): a batch's depth gathers are requested together, then the value + ssao gathers of the taps that pass the text's test (the
    // others read texel 0: no branch between the requests), then the text's arithmetic in the text's order -- same operations on the same values; a batch no lane of the
    // wave reaches is skipped. sample_count is 2 (tap 1 only) wherever validity > 0.5, hence a first batch of one, then the other six: 14 round trips to memory become 4.
    // MI355X (profiles/r06_screen_passes.md): 44.0 -> 39.8 us at 1080p, 178 -> 161 us at 4K; {1,3,3} 40.2 / 165, {3,4} 40.7 / 164, {7} 40.9 / 165.
    constexpr int sf_sizes[] = {KJ_SF_SIZES};
    constexpr int sf_batches = int(sizeof(sf_sizes) / sizeof(int));
    uint32_t b = 1;
#pragma unroll
    for (int bi = 0; bi < sf_batches; ++bi) {
        if (!wave_any(b < sample_count)) break;
        constexpr int NB = 7;      // largest batch
        int t_idx[NB]; float t_depth[NB]; bool t_use[NB]; uint2 t_v[NB]; uint32_t t_s[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) if (k < sf_sizes[bi]) {
            const uint32_t si = b + k;
            const V2 so = cos_sin_turns_fast((float(si) + ang_off) * KJ_GOLDEN_ANGLE) * (tap_pow[si & 7u] * RADIUS_SAMPLE_MULT);
            const int sx = int(float(x) + so.x), sy = int(float(y) + so.y);
            t_idx[k] = (si < sample_count && depth_tex.inb(sx, sy)) ? sy * depth_tex.w + sx : -1;
            t_depth[k] = depth_tex.p[t_idx[k] < 0 ? 0 : t_idx[k]];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) if (k < sf_sizes[bi]) {
            t_use[k] = t_idx[k] >= 0 && t_depth[k] != 0;
            t_v[k] = input_tex.p[t_use[k] ? t_idx[k] : 0];
            t_s[k] = ssao_tex.p[t_use[k] ? t_idx[k] : 0];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) if (k < sf_sizes[bi]) {
            if (t_use[k]) {
                const V3 sample_val = xyz(unpack_rgba16f(t_v[k]));
                const float sample_ssao = from_unorm8(uint8_t(t_s[k]));
                float wt = exp2_fast(-fabsf(depth_scale * (center_depth * rcp_fast(t_depth[k]) - 1.0f)));
                wt *= exp2_fast(-20.0f * fabsf(sample_ssao - center_ssao));
                sum += v4(crunch(sample_val), 1.0f) * wt;
            }
        }
        b += sf_sizes[bi];
    }
#else
#pragma unroll
    for (uint32_t si = 1; si < 8u; ++si) {
        if (!wave_any(si < sample_count)) break;       // sample_count is per pixel; the wave stops at its largest
        const V2 so = cos_sin_turns_fast((float(si) + ang_off) * KJ_GOLDEN_ANGLE) * (tap_pow[si] * RADIUS_SAMPLE_MULT);
        const int sx = int(float(x) + so.x), sy = int(float(y) + so.y);
#if KJ_SPATIAL_FILTER_TILED
        // |so| <= 16 px: inside the staged tile (clamped for safety: a tap cannot leave it by more than rounding)
        const int ti = min(max(sy - ty0, 0), KJ_SF_T - 1) * KJ_SF_T + min(max(sx - tx0, 0), KJ_SF_T - 1);
        const float4 tv = t_val[ti];
        const float sample_depth = tv.w;
        if (sample_depth != 0 && si < sample_count) {
            const float sample_ssao = from_unorm8(t_ssao[ti]);
            float wt = exp2_fast(-fabsf(depth_scale * (center_depth * rcp_fast(sample_depth) - 1.0f)));
            wt *= exp2_fast(-20.0f * fabsf(sample_ssao - center_ssao));
            sum += v4(V3{tv.x, tv.y, tv.z}, 1.0f) * wt;
        }
#else
        const float sample_depth = depth_tex.ld(sx, sy);
        if (sample_depth != 0 && si < sample_count) {
            const V3 sample_val = xyz(ld4(input_tex, sx, sy));
            const float sample_ssao = from_unorm8(ssao_tex.ld(sx, sy));
            float wt = exp2_fast(-fabsf(depth_scale * (center_depth * rcp_fast(sample_depth) - 1.0f)));
            wt *= exp2_fast(-20.0f * fabsf(sample_ssao - center_ssao));
            sum += v4(crunch(sample_val), 1.0f) * wt;
        }
#endif
    }
#endif
    const float norm_factor = rcp_fast(fmaxf(1e-5f, sum.w));
    st4(output_tex, x, y, v4(uncrunch(xyz(sum) * norm_factor), 1.0f));
}

// ------------------------------------------------------------------ temporal_filter.hlsl:39-252
// temporal_filter.hlsl is instruction-bound here (VALUBusy 85 %): its quotients and square roots feed blends and clamp boxes, so they
// take the single-instruction forms (v_rsq_f32 / v_rcp_f32 / v_sqrt_f32, 1 ulp); the 5x5 weights sum to a compile-time constant.
typedef Img<uint2> ImgH4;     // RGBA16F
typedef Img<uint32_t> ImgU32; // RG16F
typedef Img<uint2> ImgU2;     // RGBA16_SNORM
// (Lives here since round 4: rtdgi.hip is compiled without FMA contraction for the passes that take discrete decisions; this filter, approximate by design, is not one.)
#define TF_TILE_XY(W_, H_)                                                          \
    const int lane = threadIdx.x;                                                   \
    const uint2 kj_tb = kj::tile_order<KJ_TILES_ROWS>();                            \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = row0 + int(kj_tb.y) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < ((H_) < row1 ? (H_) : row1);
KJ_D V4 crunch_fast(V4 v) {      // linear_rgb_to_crunched_luma_chroma: y * sqrt(y.x) / max(1e-8, y.x)
    const V3 y = sRGB_to_YCbCr(xyz(v));
    return v4(y * (sqrt_fast(y.x) * rcp_fast(fmaxf(1e-8f, y.x))), v.w);
}
__global__ void __launch_bounds__(64) k_temporal_filter(const FrameConstants* __restrict__ fcp, ImgH4 input_tex, ImgH4 history_tex, ImgU32 variance_history_tex /*RG16F*/,
                                                         ImgU2 reprojection_tex, ImgU32 rt_history_invalidity_tex /*RG16F half*/, ImgH4 output_tex, ImgH4 history_output_tex,
                                                         ImgU32 variance_history_output_tex, int row0, int row1) {
    const int W = output_tex.w, H = output_tex.h;
    TF_TILE_XY(W, H)
    const FrameConstants& fc = *fcp;
    const V4 history_mult{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
    // LDS-staged 12x12 tile (8x8 outputs + the 5x5 stencil's halo): each texel's colour-space conversion is done once per
    // tile instead of once per tap (25x per texture): .xyz = crunched luma-chroma of the input, .w = crunched history luma.
    __shared__ float4 tile[12 * 12];
    // (round 6: the pixel's own four texels and the up to three staged texels' two images are requested together, before the first is used, and the moments' bilinear
    // footprint as soon as the reprojection is known -- under the tile's staging instead of behind the 5x5 loop: three round trips to memory instead of seven)
    bool c_in, c_in_inv;      // (x, y) lies in all three full-res images
    const uint2 center_raw = input_tex.ld_raw(x, y, c_in);
    const uint2 reproj_raw = reprojection_tex.ld_raw(x, y, c_in);
    const uint2 history_raw = history_tex.ld_raw(x, y, c_in);
    const uint32_t rt_inv_raw = rt_history_invalidity_tex.ld_raw(x / 2, y / 2, c_in_inv);
    {
        const int tx0 = int(kj_tb.x) * 8 - 2, ty0 = row0 + int(kj_tb.y) * 8 - 2;
        uint2 sn[3], sh[3]; bool s_in[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + 64 * k;
            const int tx = tx0 + i % 12, ty = ty0 + i / 12;
            sn[k] = input_tex.ld_raw(tx, ty, s_in[k]);
            sh[k] = history_tex.ld_raw(tx, ty, s_in[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane + 64 * k;
            if (i < 144) {
                const V4 n = crunch_fast(unpack_rgba16f(s_in[k] ? sn[k] : make_uint2(0u, 0u)));
                const V4 hn = crunch_fast(unpack_rgba16f(s_in[k] ? sh[k] : make_uint2(0u, 0u)) * history_mult);
                tile[i] = make_float4(n.x, n.y, n.z, hn.x);
            }
        }
    }
    const uint2 rpr = c_in ? reproj_raw : make_uint2(0u, 0u);
    const V4 reproj{from_snorm16(int16_t(rpr.x & 0xffff)), from_snorm16(int16_t(rpr.x >> 16)), from_snorm16(int16_t(rpr.y & 0xffff)), from_snorm16(int16_t(rpr.y >> 16))};
    const V2 uv = get_uv(float(x), float(y), tex_size4(W, H));
    const V2 moments_history_raw = sample_bilinear_clamp_rg16f(variance_history_tex.p, W, H, uv + V2{reproj.x, reproj.y});      // (clamped addresses: safe for the lanes outside the image too)
    __syncthreads();
    if (!in_image) return;
    const V4 center = crunch_fast(unpack_rgba16f(c_in ? center_raw : make_uint2(0u, 0u)));
    const V4 history = crunch_fast(unpack_rgba16f(c_in ? history_raw : make_uint2(0u, 0u)) * history_mult);
    V3 vsum = v3(0.0f), vsum2 = v3(0.0f);
    float wsum = 0, hist_vsum = 0;
    const int lt = ((lane >> 3) + 2) * 12 + (lane & 7) + 2;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const float4 t = tile[lt + dy * 12 + dx];
            const V3 neigh{t.x, t.y, t.z};
            const float hist_luma = t.w;
            const float w = expf(-3.0f * float(dx * dx + dy * dy) / float((2 + 1.) * (2 + 1.)));
            vsum += neigh * w;
            vsum2 += neigh * neigh * w;      // (the squares from a second LDS tile, once per texel: measured 51 -> 75 us at 1080p, round 5 -- the kernel is LDS-bound on its 25 tile reads; reverted)
            wsum += w;
            hist_vsum += hist_luma * w;
        }
    const float inv_wsum = 1.0f / wsum;      // (compile-time constant)
    const V3 ex = vsum * inv_wsum, ex2 = vsum2 * inv_wsum;
    const V3 var = vmax(v3(0.0f), ex2 - ex * ex);
    const V3 dev{sqrt_fast(var.x), sqrt_fast(var.y), sqrt_fast(var.z)};
    hist_vsum *= inv_wsum;
    const V2 moments_history = moments_history_raw * V2{fc.pre_exposure_delta, fc.pre_exposure_delta * fc.pre_exposure_delta};
    const float center_luma = center.x + (hist_vsum - ex.x);
    const V2 mo = lerp(moments_history, V2{center_luma, center_luma * center_luma}, 0.25f);
    st2h(variance_history_output_tex, x, y, V2{fmaxf(0.0f, mo.x), fmaxf(0.0f, mo.y)});
    const float center_temporal_dev = sqrt_fast(fmaxf(0.0f, moments_history.y - moments_history.x * moments_history.x));
    const float temporal_change = fabsf(hist_vsum - ex.x) * rcp_fast(fmaxf(1e-8f, hist_vsum + ex.x));
    const float rt_invalid = saturate(sqrt_fast(unpack_2x16f_uint(c_in_inv ? rt_inv_raw : 0u).x) * 4);
    const float current_sample_count = history.w;
    float clamp_box_size = 1 * lerp(0.25f, 2.0f, 1.0f - rt_invalid) * lerp(0.333f, 1.0f, saturate(reproj.w)) * 2;
    clamp_box_size = fmaxf(clamp_box_size, 0.5f);
    const V3 nmin = xyz(center) - dev * clamp_box_size, nmax = xyz(center) + dev * clamp_box_size;
    const V3 clamped_history = vclamp(xyz(history), nmin, nmax);
    const float variance_adjusted_temporal_change = smoothstep(0.1f, 1.0f, 0.05f * temporal_change * rcp_fast(center_temporal_dev));
    float max_sample_count = 32;
    max_sample_count = lerp(max_sample_count, 4.0f, variance_adjusted_temporal_change);
    max_sample_count *= lerp(1.0f, 0.5f, rt_invalid);
    const V3 res = lerp(clamped_history, xyz(center), rcp_fast(1.0f + fminf(max_sample_count, current_sample_count)));
    const float output_sample_count = fminf(current_sample_count, max_sample_count) + 1;
    const V4 output = crunched_luma_chroma_to_linear_rgb(v4(res, output_sample_count));
    st4(history_output_tex, x, y, output);
    st4(output_tex, x, y, v4(xyz(output), saturate(output_sample_count * lerp(1.0f, 0.5f, rt_invalid) * smoothstep(0.3f, 0.0f, temporal_change) / 32.0f)));
}

namespace kj {
hipError_t launch_temporal_filter(const KjFrameConstants* fc, const void* input, const void* history, const void* variance_history, const void* reprojection, const void* rt_history_invalidity,
                                  void* output, void* history_output, void* variance_history_output, int W, int H, int row0, int row1, hipStream_t s) {
    const int hw = (W + 1) / 2, hh = (H + 1) / 2;
    hipLaunchKernelGGL(k_temporal_filter, dim3((W + 7) / 8, (row1 - row0 + 7) / 8), dim3(64), 0, s, fc, img<uint2>(input, W, H), img<uint2>(history, W, H), img<uint32_t>(variance_history, W, H),
                       img<uint2>(reprojection, W, H), img<uint32_t>(rt_history_invalidity, hw, hh), img<uint2>(output, W, H), img<uint2>(history_output, W, H),
                       img<uint32_t>(variance_history_output, W, H), row0, row1);
    return hipGetLastError();
}
}  // namespace kj

namespace kj {

hipError_t launch_restir_spatial(const SpatialLaunch& L, hipStream_t s) {
    SpatialArgs a;
    a.fc = L.fc;
    a.reservoir_input_tex = img<uint2>(L.reservoir_input, L.hw, L.hh);
    a.half_gbuf = img<uint2>(L.half_gbuf, L.hw, L.hh);
    a.half_depth_tex = img<float>(L.half_depth, L.hw, L.hh);
    a.temporal_reservoir_packed_tex = img<uint4>(L.temporal_reservoir_packed, L.hw, L.hh);
    a.reservoir_output_tex = img<uint2>(L.reservoir_output, L.hw, L.hh);
    a.W = L.W; a.H = L.H;
    a.pass_idx = L.pass_idx; a.perform_occlusion_raymarch = L.perform_occlusion_raymarch; a.occlusion_raymarch_importance_only = L.occlusion_raymarch_importance_only;
    a.row0 = L.row0; a.row1 = L.row1;
    const int rows = L.row1 - L.row0;
#ifdef KJ_SPATIAL_TILES
    const int order = KJ_SPATIAL_TILES;
#else
    const int order = resample_tile_order((L.hw + 15) / 16);
#endif
#ifndef KJ_SPATIAL_MARCH_SPLIT
#define KJ_SPATIAL_MARCH_SPLIT 1      // 0: every kernel carries the march (one build per shape)
#endif
#define KJ_SPATIAL_O(R_, S_, BW_, BH_, T_, O_, M_) hipLaunchKernelGGL((k_restir_spatial<R_, S_, BW_, BH_, T_, O_, M_>), dim3((L.hw + BW_ - 1) / BW_, (rows + BH_ - 1) / BH_), dim3(BW_ * BH_), 0, s, a)
#define KJ_SPATIAL_M(R_, S_, BW_, BH_, T_, M_) do { if (order == KJ_TILES_BANDS) KJ_SPATIAL_O(R_, S_, BW_, BH_, T_, KJ_TILES_BANDS, M_); else if (order == KJ_TILES_SUPER) KJ_SPATIAL_O(R_, S_, BW_, BH_, T_, KJ_TILES_SUPER, M_); \
                                                   else KJ_SPATIAL_O(R_, S_, BW_, BH_, T_, KJ_TILES_PLAIN, M_); } while (0)
    // the first pass of the default two does not march: its kernel is built without the march; any later pass uses the build that has it (the run-time flag decides)
#define KJ_SPATIAL(R_, S_, BW_, BH_, T_) do { if (KJ_SPATIAL_MARCH_SPLIT && L.pass_idx == 0 && !L.perform_occlusion_raymarch) KJ_SPATIAL_M(R_, S_, BW_, BH_, T_, false); else KJ_SPATIAL_M(R_, S_, BW_, BH_, T_, true); } while (0)
    if (L.pass_idx == 0) {
        if (L.variant == 0) KJ_SPATIAL(32, 8, 32, 32, true);
        else if (L.variant == 1) KJ_SPATIAL(32, 8, 16, 16, true);
        else KJ_SPATIAL(32, 8, 16, 16, false);
    } else {
        if (L.variant == 0 || L.variant == 1) KJ_SPATIAL_M(16, 5, 16, 16, true, true);
        else KJ_SPATIAL_M(16, 5, 16, 16, false, true);
    }
#undef KJ_SPATIAL_O
#undef KJ_SPATIAL_M
#undef KJ_SPATIAL
    return hipGetLastError();
}

hipError_t launch_restir_resolve(const ResolveLaunch& L, hipStream_t s) {
    ResolveArgs2 a;
    a.fc = L.fc;
    a.radiance_tex = img<uint2>(L.radiance, L.hw, L.hh);
    a.reservoir_input_tex = img<uint2>(L.reservoir_input, L.hw, L.hh);
    a.gbuffer_tex = img<uint4>(L.gbuffer, L.W, L.H);
    a.depth_tex = img<float>(L.depth, L.W, L.H);
    a.half_gbuf = img<uint2>(L.half_gbuf, L.hw, L.hh);
    a.ssao_tex = img<uint8_t>(L.ssao, L.W, L.H);
    a.candidate_radiance_tex = img<uint2>(L.candidate_radiance, L.hw, L.hh);
    a.candidate_hit_tex = img<uint2>(L.candidate_hit, L.hw, L.hh);
    a.temporal_reservoir_packed_tex = img<uint4>(L.temporal_reservoir_packed, L.hw, L.hh);
    a.irradiance_output_tex = img<uint2>(L.irradiance_output, L.W, L.H);
    a.blue_noise = (const uint32_t*)L.blue_noise;
    a.row0 = L.row0; a.row1 = L.row1;
#ifdef KJ_RESOLVE_TILES
    const int order = KJ_RESOLVE_TILES;
#else
    const int order = resample_tile_order((L.W + 15) / 16);
#endif
    const dim3 grid((L.W + 15) / 16, (L.row1 - L.row0 + 15) / 16);
    if (order == KJ_TILES_BANDS) hipLaunchKernelGGL(k_restir_resolve<KJ_TILES_BANDS>, grid, dim3(256), 0, s, a);
    else if (order == KJ_TILES_SUPER) hipLaunchKernelGGL(k_restir_resolve<KJ_TILES_SUPER>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_restir_resolve<KJ_TILES_PLAIN>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_spatial_filter(const KjFrameConstants* fc, const void* input, const void* depth, const void* ssao, const void* geometric_normal, void* output,
                                 int W, int H, int row0, int row1, hipStream_t s) {
#if KJ_SPATIAL_FILTER_TILED
    hipLaunchKernelGGL(k_spatial_filter, dim3((W + KJ_SF_B - 1) / KJ_SF_B, (row1 - row0 + KJ_SF_B - 1) / KJ_SF_B), dim3(256), 0, s, fc, img<uint2>(input, W, H), img<float>(depth, W, H), img<uint8_t>(ssao, W, H),
                       img<uint32_t>(geometric_normal, W, H), img<uint2>(output, W, H), row0, row1);
#else
    hipLaunchKernelGGL(k_spatial_filter, dim3((W + 7) / 8, (row1 - row0 + 7) / 8), dim3(64), 0, s, fc, img<uint2>(input, W, H), img<float>(depth, W, H), img<uint8_t>(ssao, W, H),
                       img<uint32_t>(geometric_normal, W, H), img<uint2>(output, W, H), row0, row1);
#endif
    return hipGetLastError();
}

}  // namespace kj
