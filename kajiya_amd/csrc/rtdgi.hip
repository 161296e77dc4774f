// RtdgiRenderer for gfx950: one HIP kernel per reference pass (renderers/rtdgi.rs:143-554,
// assets/shaders/rtdgi/*.hlsl), 8x8 pixel tile = one wave64, neighbour exchange by
// __shfl_xor, software BVH traversal instead of TraceRay. Host orchestration mirrors
// RtdgiRenderer::{reproject,render} including the ping-pong temporal resources.
#include "kj_host.hpp"
#include "kj_scene.hpp"
#include "kj_reservoir.hpp"
#include "kj_ircache.hpp"
#include "kj_ircache_host.hpp"
#include "kj_screen.hpp"
#include "rtdgi_resample.hpp"
#include <cstdlib>
#include <algorithm>

using namespace kj;
namespace kj { SceneView scene_view(const KjScene& s); }

#define SKY_DIST 1e4f
#define RESTIR_TEMPORAL_M_CLAMP 20.0f
#define RESTIR_RESERVOIR_W_CLAMP 10.0f
#define SSGI_NEAR_FIELD_RADIUS 80.0f
#define ROUGHNESS_BIAS 0.5f

typedef Img<uint2> ImgH4;     // RGBA16F
typedef Img<uint32_t> ImgU32; // RGBA8_SNORM / RG16F / A2R10G10B10
typedef Img<float> ImgF32;
typedef Img<uint4> ImgU4;
typedef Img<uint2> ImgU2;     // RG32UI (reservoirs) and RGBA16_SNORM share the 8-byte texel
typedef Img<uint8_t> ImgR8;
typedef Img<int8_t> ImgR8S;
typedef Img<float4> ImgF4;

// Row-range aware tile mapping: a launch covers image rows [row0, row1) of the kernel's own resolution
// (row0 a multiple of 8) so the same kernels serve the screen-tile split across GPUs.
#define TILE_XY(W_, H_) TILE_XY_M(W_, H_, KJ_TILES_PLAIN)
#define TILE_XY_M(W_, H_, MODE_)                                          \
    const int lane = threadIdx.x;                                         \
    const uint2 kj_tb = kj::tile_order<MODE_>();                          \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = row0 + int(kj_tb.y) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < ((H_) < row1 ? (H_) : row1);

// (the QUAD form of the ray passes -- experiments builds: four lanes per pixel, a wave covers 8 x 2 pixels, lane = 4 * pixel slot + k; `lead` = the lane that stores)
// the two fused ray kernels' pixel of a lane, given the tile (the one-launch form of a validation frame's two passes hands each workgroup its tile)
#define RAY_TILE_XY(W_, H_, QUAD_, TB_)                                                                                    \
    const int lane = (QUAD_) ? int(threadIdx.x >> 2) : int(threadIdx.x);                                                   \
    const bool lead = !(QUAD_) || (threadIdx.x & 3u) == 0u;                                                                \
    const int x = int((TB_).x) * 8 + (lane & 7), y = row0 + int((TB_).y) * ((QUAD_) ? 2 : 8) + (lane >> 3);                \
    const bool in_image = x < (W_) && y < ((H_) < row1 ? (H_) : row1);

// ------------------------------------------------------------------ extract_half_res_{gbuffer_view_normal_rgba8,depth,ssao}.hlsl (fused)
// Besides the reference's three half-res images this writes them once more as ONE 8-byte record per pixel
// {depth bits, view normal snorm8 x3 | ssao snorm8 << 24}: what the resampling passes stage in LDS / fetch per tap (rtdgi_resample.hip).
// SSAO_MODE 0: all three (the reference's pass); 1: everything but the SSAO (the record keeps a zero in its SSAO byte); 2: only the SSAO, into
// the image and the record's byte -- the two halves of a frame whose SSAO guide is still being computed on another stream while the ray
// passes run (KJ_RTDGI_PASS_EXTRACT_HALF_NO_SSAO / _SSAO_ONLY: nothing before `restir spatial` reads the half-res SSAO)
template <int SSAO_MODE>
__global__ void __launch_bounds__(64) k_extract_half(const FrameConstants* __restrict__ fcp, ImgU4 gbuffer, ImgF32 depth, ImgR8 ssao, ImgU32 half_view_normal,
                                                      ImgF32 half_depth, ImgR8S half_ssao, ImgU2 half_gbuf, int row0, int row1) {
    TILE_XY_M(half_depth.w, half_depth.h, KJ_TILES_ROWS)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int sx = x * 2 + off.x, sy = y * 2 + off.y;
    if (SSAO_MODE == 2) {
        const int8_t ao = to_snorm8(from_unorm8(ssao.ld(sx, sy)));
        half_ssao.st(x, y, ao);
        ((uint8_t*)half_gbuf.p)[(size_t(y) * half_gbuf.w + x) * 8 + 7] = uint8_t(ao);
        return;
    }
    const V3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(sx, sy).y);
    const V3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, v4(normal_ws, 0))));
    const uint32_t packed_normal = pack_rgba8_snorm(v4(normal_vs, 1.0f));
    const float d = depth.ld(sx, sy);
    const int8_t ao = SSAO_MODE == 1 ? int8_t(0) : to_snorm8(from_unorm8(ssao.ld(sx, sy)));
    half_view_normal.st(x, y, packed_normal);
    half_depth.st(x, y, d);
    if (SSAO_MODE == 0) half_ssao.st(x, y, ao);
    half_gbuf.st(x, y, make_uint2(asuint(d), (packed_normal & 0x00ffffffu) | (uint32_t(uint8_t(ao)) << 24)));
}

// ------------------------------------------------------------------ fullres_reproject.hlsl:29-76
KJ_D V4 cubic_hermite(V4 A, V4 B, V4 C, V4 D, float t) {  // inc/curve.hlsl:4-13
    const float t2 = t * t, t3 = t * t * t;
    const V4 a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
    const V4 b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
    const V4 c = -A / 2.0f + C / 2.0f;
    return a * t3 + b * t2 + c * t + B;
}
__global__ void __launch_bounds__(64) k_fullres_reproject(ImgH4 input_tex, ImgU2 reprojection_tex, ImgH4 output_tex, int row0, int row1) {
    const int W = output_tex.w, H = output_tex.h;
    TILE_XY(W, H)
    if (!in_image) return;
    const V4 ts = tex_size4(W, H);
    const V2 uv = get_uv(float(x), float(y), ts);
    const V4 reproj = ld_reproj(reprojection_tex, x, y);
    const V2 prev_uv = uv + V2{reproj.x, reproj.y};
    const uint32_t quad_valid = uint32_t(reproj.z * 15.0f + 0.5f);
    V4 history = v4(0.0f);
    if (quad_valid == 15) {
        // GatherBlue(sampler_nnc, uv + 0.5*sign(prev_uv)*texel): validity of the 2x2 footprint
        const V2 guv = uv + 0.5f * V2{float((prev_uv.x > 0) - (prev_uv.x < 0)), float((prev_uv.y > 0) - (prev_uv.y < 0))} * V2{ts.z, ts.w};
        const int ox = int(floorf(guv.x * float(W) - 0.5f)), oy = int(floorf(guv.y * float(H) - 0.5f));
        // (round 6: the four texels of the footprint are requested together -- `&&` between them made every load wait for the previous one's answer)
        uint2 fp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sx = min(max(ox + (i & 1), 0), W - 1), sy = min(max(oy + (i >> 1), 0), H - 1);
            fp[i] = reprojection_tex.p[size_t(sy) * W + sx];      // clamped: in bounds
        }
        bool all_valid = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) all_valid = all_valid & (uint32_t(from_snorm16(int16_t(fp[i].y & 0xffff)) * 15.0f + 0.5f) == 15u);
        if (all_valid) {
            const V2 pixel = prev_uv * V2{float(W), float(H)} + 0.5f;
            const V2 frc{frac(pixel.x), frac(pixel.y)};
            const int ipx = int(pixel.x) - 1, ipy = int(pixel.y) - 1;
            V4 rows[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rows[j] = cubic_hermite(ld4(input_tex, ipx - 1, ipy - 1 + j), ld4(input_tex, ipx, ipy - 1 + j), ld4(input_tex, ipx + 1, ipy - 1 + j),
                                        ld4(input_tex, ipx + 2, ipy - 1 + j), frc.x);
            history = vmax(v4(0.0f), cubic_hermite(rows[0], rows[1], rows[2], rows[3], frc.y));
        } else {
            history = sample_bilinear_clamp_rgba16f(input_tex.p, W, H, prev_uv);
        }
    } else if (quad_valid != 0) {
        const V4 qv{(quad_valid & 1) ? 1.0f : 0.0f, (quad_valid & 2) ? 1.0f : 0.0f, (quad_valid & 4) ? 1.0f : 0.0f, (quad_valid & 8) ? 1.0f : 0.0f};
        const V2 bp = prev_uv * V2{float(W), float(H)} - 0.5f;
        const int ox = int(truncf(bp.x)), oy = int(truncf(bp.y));
        const V2 bw{frac(bp.x), frac(bp.y)};
        V4 w{(1.0f - bw.x) * (1.0f - bw.y), bw.x * (1.0f - bw.y), (1.0f - bw.x) * bw.y, bw.x * bw.y};
        w = w * qv;
        const float wsum = dot(w, v4(1.0f));
        if (wsum > 1e-5f) {
            const V4 r = ld4(input_tex, ox, oy) * w.x + ld4(input_tex, ox + 1, oy) * w.y + ld4(input_tex, ox, oy + 1) * w.z + ld4(input_tex, ox + 1, oy + 1) * w.w;
            history = r * (1.0f / wsum);
        }
    }
    st4(output_tex, x, y, history);
}

// ------------------------------------------------------------------ diffuse_trace_common.inc.hlsl:38-221
struct TraceCtx {
    const FrameConstants* __restrict__ fc;
    SceneView sc;
    ImgF32 depth;                 // full res
    ImgH4 reprojected_gi;         // full res
    const uint2* __restrict__ sky_cube; int sky_cube_width;
    const uint32_t* __restrict__ blue_noise;
    const uint2* __restrict__ brdf_fg_lut;
    const float4* __restrict__ sun_color;
    IrcacheView irc; bool has_ircache;              // IrcacheRenderState bound via bind_mut (rtdgi.rs:321,350)
    uint32_t request_slot_base, request_key_base, request_stride;   // deferred ircache updates: slot / key of pixel (x, y) = base + y * stride + x
    unsigned long long* __restrict__ ray_counters;  // [0]=closest rays, [1]=any-hit rays, [2..5]=nodes/tris visited (closest, any) in STATS builds
};

KJ_D void count_rays(unsigned long long* counters, int which, bool active) {
#if !defined(__HIP_DEVICE_COMPILE__)      // the tests' CPU stand-in for HIP has no wave votes in multi-wave workgroups: one atomic per lane
    if (active) atomicAdd(&counter_slot(counters)[which], 1ull);
    return;
#endif
    const unsigned long long m = __ballot(active);
    if (m != 0ull && (__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(&counter_slot(counters)[which], (unsigned long long)__popcll(m));
}

// ---- fused form of the two ray passes (the default): one invocation per pixel does ray generation, both traversals, hit shading
// and the bookkeeping, as the reference's ray-generation shaders do; the traversal inside votes per wave on node vs triangle steps.
// Measured against the staged form below (profiles/r02_ray_pipeline_variants.md): trace pass 0.259 ms fused / 0.437 staged at
// 1080p (292 k closest + 91 k occlusion rays per launch: 47 rays per persistent wave, nothing to refill from, and five dependent
// launches instead of one) and 0.777 / 0.918 ms at 4K. The kernel is bound by the wave-instructions it issues at 34 % lane
// utilisation (two thirds of the rays leave the scene, and their lanes idle through hit shading and the shadow ray): its rays alone
// would take 0.13 of its 0.28 ms at the microbench's rates.
struct TraceResult { V3 out_value; V3 hit_normal_ws; float hit_t; float pdf; bool is_hit; };
// QUAD form: four lanes per pixel run the same code on the same inputs (kj_bvh.hpp: bvh_trace_quad); the first lane of each quad counts
KJ_D void count_rays_quad(unsigned long long* counters, int which) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if ((threadIdx.x & 3u) == 0u) atomicAdd(&counter_slot(counters)[which], 1ull);
    return;
#endif
    const unsigned long long m = __ballot(true) & 0x1111111111111111ull;
    if (m != 0ull && (__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(&counter_slot(counters)[which], (unsigned long long)__popcll(m));
}
// Everything diffuse_trace_common.inc.hlsl:80-200 does for a ray that HIT (the divergent part: G-buffer of the hit, the sun's shadow ray,
// the triangle lights, last frame's GI or the irradiance cache). Shared by the fused and the grouped form of the ray passes so that both
// sum the radiance terms with the same arithmetic in the same order. Returns the radiance; `hit_normal_ws` = the hit's shading normal.
// `sun_shadowed(position, to_light_norm)` answers the sun's shadow ray: the fused form walks it on the spot; the pool form (k_rtdgi_rays_pool) has walked it already, among
// the other rays of its wave, and hands the answer in.
template <bool STATS, bool QUAD = false, typename SunShadowed>
KJ_D V3 shade_candidate_hit_with(const TraceCtx& c, uint32_t px, uint32_t py, uint32_t& rng, V3 ray_o, V3 ray_d, const GbufferPathVertex& primary_hit, uint32_t* stack, uint32_t stride,
                                 TraverseStats* st_any, V3& hit_normal_ws, SunShadowed sun_shadowed) {
    const FrameConstants& fc = *c.fc;
    V3 total_radiance = v3(0.0f);
    GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
    hit_normal_ws = gbuffer.normal;
    const V3 hit_cs = position_world_to_sample(fc, primary_hit.position);
    const V2 hit_uv = cs_to_uv(V2{hit_cs.x, hit_cs.y});
    const float screen_depth = sample_nearest_clamp(c.depth, hit_uv);
    bool is_on_screen = fabsf(hit_cs.x) < 1.0f && fabsf(hit_cs.y) < 1.0f && inverse_depth_relative_diff(hit_cs.z, screen_depth) < 5e-3f;
    V4 reprojected_radiance = v4(0.0f);
    if (is_on_screen) {
        reprojected_radiance = unpack_rgba16f(sample_nearest_clamp(c.reprojected_gi, hit_uv)) * fc.pre_exposure_delta;
        is_on_screen = reprojected_radiance.w > 0;
    }
    gbuffer.roughness = lerp(gbuffer.roughness, 1.0f, ROUGHNESS_BIAS);
    const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
    const V3 wo = to_local(tangent_to_world, -ray_d);
    const LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(c.brdf_fg_lut, gbuffer, wo.z);
    const float4 sc4 = *c.sun_color;
    const V3 sun_radiance{sc4.x, sc4.y, sc4.z};
    if (sun_radiance.x != 0 || sun_radiance.y != 0 || sun_radiance.z != 0) {
        const V4 bn = blue_noise_for_pixel(c.blue_noise, px, py, rng);
        const V3 to_light_norm = sample_sun_direction(fc, V2{bn.x, bn.y}, false);
        const bool is_shadowed = sun_shadowed(primary_hit.position, to_light_norm);
        const V3 wi = to_local(tangent_to_world, to_light_norm);
        const V3 brdf_value = layered_brdf_evaluate(brdf, wo, wi) * fmaxf(0.0f, wi.z);
        total_radiance += brdf_value * (is_shadowed ? v3(0.0f) : sun_radiance);
    }
    total_radiance += gbuffer.emissive;
    if (is_on_screen) {
        total_radiance += xyz(reprojected_radiance) * gbuffer.albedo;
    } else {
        V2 urand;
        urand.x = uint_to_u01_float(hash1_mut(rng));
        urand.y = uint_to_u01_float(hash1_mut(rng));
        const uint32_t nl = min(fc.triangle_light_count, c.sc.light_count);
        for (uint32_t li = 0; li < nl; ++li) {
            const KjTriangleLight tl = c.sc.lights[li];
            const V3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
            const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
            const V3 to_light_ws = ls.pos - primary_hit.position;
            const float dist2 = dot(to_light_ws, to_light_ws);
            const V3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist2));
            const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist2;
            if (to_psa_metric > 0.0f) {
                if (QUAD) count_rays_quad(c.ray_counters, 1); else count_rays(c.ray_counters, 1, true);
                const bool is_shadowed = QUAD ? rt_is_shadowed_quad(c.sc, true, primary_hit.position, to_light_norm_ws, 1e-3f, sqrtf(dist2) - 2e-3f, stack, stride)
                                              : rt_is_shadowed<STATS>(c.sc, primary_hit.position, to_light_norm_ws, 1e-3f, sqrtf(dist2) - 2e-3f, stack, stride, st_any);
                const V3 bounce_albedo = lerp(gbuffer.albedo, v3(1.0f), 0.04f);
                const V3 brdf_value = bounce_albedo * to_psa_metric / KJ_PI;
                if (!is_shadowed) total_radiance += V3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * brdf_value / ls.pdf;
            }
        }
        if (c.has_ircache) {  // USE_IRCACHE (diffuse_trace_common.inc.hlsl:189-198); unbound => contributes 0 (BASELINE config 1)
            const uint32_t rq = py * c.request_stride + px;
            V3 gi = v3(0.0f);
            // the lookup's side effects (atomics, or the recorded request) are the quad's first lane's; the rng is not used after it
            if (!QUAD || (threadIdx.x & 3u) == 0u) gi = ircache_lookup<false>(c.irc, fc, ray_o, primary_hit.position, gbuffer.normal, 1u, rng, false, c.request_slot_base + rq, c.request_key_base | rq);
            if (QUAD) gi = quad_broadcast0(gi);
            total_radiance += gi * gbuffer.albedo;
        }
    }
    return total_radiance;
}
template <bool STATS, bool QUAD = false>
KJ_D V3 shade_candidate_hit(const TraceCtx& c, uint32_t px, uint32_t py, uint32_t& rng, V3 ray_o, V3 ray_d, const GbufferPathVertex& primary_hit, uint32_t* stack, uint32_t stride,
                            TraverseStats* st_any, V3& hit_normal_ws) {
    return shade_candidate_hit_with<STATS, QUAD>(c, px, py, rng, ray_o, ray_d, primary_hit, stack, stride, st_any, hit_normal_ws, [&](V3 position, V3 to_light_norm) -> bool {
        if (QUAD) count_rays_quad(c.ray_counters, 1); else count_rays(c.ray_counters, 1, true);
        return QUAD ? rt_is_shadowed_quad(c.sc, true, position, to_light_norm, 1e-4f, SKY_DIST, stack, stride)
                    : rt_is_shadowed<STATS>(c.sc, position, to_light_norm, 1e-4f, SKY_DIST, stack, stride, st_any);
    });
}
// diffuse_trace_common.inc.hlsl:68-71: reflected cone = the half-res pixel cone propagated from the eye to the ray origin
KJ_D RayCone candidate_ray_cone(const TraceCtx& c, V3 ray_o) {
    return pixel_ray_cone_from_image_height(*c.fc, float(c.depth.h) * 0.5f).propagate(0.03f, length(ray_o - get_eye_position(*c.fc)));
}
template <bool STATS>
KJ_D void add_traversal_stats(const TraceCtx& c, const TraverseStats& st_closest, const TraverseStats& st_any) {
    if (STATS) {  // instrumentation build only: traversal work per ray type (SURVEY 8d "measured by the instrumented kernel")
        atomicAdd(&counter_slot(c.ray_counters)[2], (unsigned long long)st_closest.nodes);
        atomicAdd(&counter_slot(c.ray_counters)[3], (unsigned long long)st_closest.tris);
        atomicAdd(&counter_slot(c.ray_counters)[4], (unsigned long long)st_any.nodes);
        atomicAdd(&counter_slot(c.ray_counters)[5], (unsigned long long)st_any.tris);
        // steps the waves issued (one lane of a wave holds its count): [6], [7] node / triangle steps of closest-hit walks, [8], [9] of occlusion walks
        if (st_closest.wave_node_steps | st_closest.wave_tri_steps) { atomicAdd(&counter_slot(c.ray_counters)[6], (unsigned long long)st_closest.wave_node_steps); atomicAdd(&counter_slot(c.ray_counters)[7], (unsigned long long)st_closest.wave_tri_steps); }
        if (st_any.wave_node_steps | st_any.wave_tri_steps) { atomicAdd(&counter_slot(c.ray_counters)[8], (unsigned long long)st_any.wave_node_steps); atomicAdd(&counter_slot(c.ray_counters)[9], (unsigned long long)st_any.wave_tri_steps); }
        // [10..13]: wave steps of the closest-hit walks by lanes still walking (1-8, 9-16, 17-32, 33-64); [14], [15]: of the occlusion walks (<= 16, > 16)
        for (int b = 0; b < 4; ++b) if (st_closest.live_hist[b]) atomicAdd(&counter_slot(c.ray_counters)[10 + b], (unsigned long long)st_closest.live_hist[b]);
        if (st_any.live_hist[0] | st_any.live_hist[1]) atomicAdd(&counter_slot(c.ray_counters)[14], (unsigned long long)(st_any.live_hist[0] + st_any.live_hist[1]));
        if (st_any.live_hist[2] | st_any.live_hist[3]) atomicAdd(&counter_slot(c.ray_counters)[15], (unsigned long long)(st_any.live_hist[2] + st_any.live_hist[3]));
    }
}
template <bool STATS, bool QUAD = false>
KJ_D TraceResult trace_candidate(const TraceCtx& c, uint32_t px, uint32_t py, V3 normal_ws, uint32_t& rng, V3 ray_o, V3 ray_d, float ray_tmax, uint32_t* stack) {
    const FrameConstants& fc = *c.fc;
    V3 total_radiance = v3(0.0f);
    V3 hit_normal_ws = -ray_d;
    float hit_t = ray_tmax;
    const float pdf = fmaxf(0.0f, 1.0f / (dot(normal_ws, ray_d) * 2 * KJ_PI));
    if (QUAD) count_rays_quad(c.ray_counters, 0); else count_rays(c.ray_counters, 0, true);
    TraverseStats st_closest{0, 0}, st_any{0, 0};
    constexpr uint32_t STRIDE = QUAD ? 16u : 64u;     // LDS stack layout [level][lane] / [level][quad]
    const GbufferPathVertex primary_hit = QUAD ? gbuffer_raytrace_quad(c.sc, fc, true, ray_o, ray_d, 0.0f, ray_tmax, 1, false, stack, STRIDE, candidate_ray_cone(c, ray_o))
                                               : gbuffer_raytrace<STATS>(c.sc, fc, ray_o, ray_d, 0.0f, ray_tmax, 1, false, stack, STRIDE, &st_closest, candidate_ray_cone(c, ray_o));
    if (primary_hit.is_hit) {
        hit_t = primary_hit.ray_t;
        total_radiance = shade_candidate_hit<STATS, QUAD>(c, px, py, rng, ray_o, ray_d, primary_hit, stack, STRIDE, &st_any, hit_normal_ws);
    } else {
        total_radiance += xyz(sample_cube_rgba16f(c.sky_cube, c.sky_cube_width, ray_d));
    }
    add_traversal_stats<STATS>(c, st_closest, st_any);
    return TraceResult{total_radiance, hit_normal_ws, hit_t, pdf, primary_hit.is_hit};
}

// ------------------------------------------------------------------ diffuse_validate.rgen.hlsl:46-111
template <bool STATS, bool QUAD = false>
// waves per SIMD the fused ray kernels are compiled for: 4 = 123 VGPRs, no spills; 5 = 96 VGPRs + 20 spilled dwords outside the traversal
// loop: trace pass -2 %; 6 = 80 VGPRs + 48 dwords: +30 % (measured, same box)
#ifndef KJ_FUSED_WAVES
#define KJ_FUSED_WAVES 5
#endif
KJ_D void rtdgi_validate_tile(const TraceCtx& c, ImgU32 half_view_normal_tex, ImgU2 reservoir_tex, ImgH4 reservoir_ray_history_tex,
                              ImgH4 irradiance_history_tex, ImgF4 ray_orig_history_tex, ImgR8 invalidity_out_tex, int row0, int row1, uint2 kj_tb, uint32_t* lds_stack) {
    RAY_TILE_XY(invalidity_out_tex.w, invalidity_out_tex.h, QUAD, kj_tb)
    if (!in_image) return;
    const FrameConstants& fc = *c.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    if (0.0f == c.depth.ld(x * 2 + off.x, y * 2 + off.y)) { if (lead) invalidity_out_tex.st(x, y, to_unorm8(1.0f)); return; }
    float invalidity = 0.0f;
    if (is_rtdgi_validation_frame(fc.frame_index)) {
        const V3 normal_ws = direction_view_to_world(fc, ld_nrm_snorm8(half_view_normal_tex, x, y));
        const float4 ro = ray_orig_history_tex.ld(x, y);
        const V3 prev_ray_orig{ro.x, ro.y, ro.z};
        const V3 prev_hit_pos = xyz(ld4(reservoir_ray_history_tex, x, y)) + prev_ray_orig;
        const V4 prev_radiance_packed = ld4(irradiance_history_tex, x, y);
        const V3 prev_radiance = vmax(v3(0.0f), xyz(prev_radiance_packed));
        uint32_t rng = hash3(uint32_t(x), uint32_t(y), 0);
        const TraceResult result = trace_candidate<STATS, QUAD>(c, x, y, normal_ws, rng, prev_ray_orig, normalize(prev_hit_pos - prev_ray_orig), SKY_DIST, lds_stack + lane);
        const V3 new_radiance = vmax(v3(0.0f), result.out_value);
        const float rad_diff = length(vabs(prev_radiance - new_radiance) / vmax(v3(1e-3f), prev_radiance + new_radiance));
        invalidity = smoothstep(0.1f, 0.5f, rad_diff / length(v3(1.0f)));
        const float prev_hit_dist = length(prev_hit_pos - prev_ray_orig);
        if (lead && fabsf(result.hit_t - prev_hit_dist) / (prev_hit_dist + prev_hit_dist) < 0.2f) {
            st4(irradiance_history_tex, x, y, v4(new_radiance, prev_radiance_packed.w));
            Reservoir1spp r = Reservoir1spp::from_raw(reservoir_tex.ld(x, y));
            const float lum_old = sRGB_to_luminance(prev_radiance), lum_new = sRGB_to_luminance(new_radiance);
            r.M *= clampf(lum_old / fmaxf(1e-8f, lum_new), 0.03f, 1.0f);
            r.W *= clampf(lum_old / fmaxf(1e-8f, lum_new) * 10.0f, 0.01f, 1.0f);
            reservoir_tex.st(x, y, r.as_raw());
        }
    }
    if (lead) invalidity_out_tex.st(x, y, to_unorm8(invalidity));
}
template <bool STATS, bool QUAD = false>
__global__ void __launch_bounds__(64, KJ_FUSED_WAVES) k_rtdgi_validate_fused(TraceCtx c, ImgU32 half_view_normal_tex, ImgU2 reservoir_tex, ImgH4 reservoir_ray_history_tex,
                                                        ImgH4 irradiance_history_tex, ImgF4 ray_orig_history_tex, ImgR8 invalidity_out_tex, int row0, int row1) {
    extern __shared__ uint32_t lds_stack[];
    rtdgi_validate_tile<STATS, QUAD>(c, half_view_normal_tex, reservoir_tex, reservoir_ray_history_tex, irradiance_history_tex, ray_orig_history_tex, invalidity_out_tex, row0, row1,
                                     kj::tile_order<KJ_TILES_PLAIN>(), lds_stack);
}

// ------------------------------------------------------------------ trace_diffuse.rgen.hlsl:49-120 + candidate_ray_dir.hlsl:1-24
// COPY_VALIDITY false: the pass' last statement -- rt_history_validity_input_tex[px] = the validate pass' output at the reprojected pixel -- is left to
// k_rtdgi_validity_reproject (the one-launch form of the two ray passes on a validation frame, below: the validate pass' tiles are still in flight)
template <bool STATS, bool QUAD = false, bool COPY_VALIDITY = true>
KJ_D void rtdgi_trace_tile(const TraceCtx& c, ImgU32 half_view_normal_tex, ImgU2 reprojection_tex, ImgH4 candidate_irradiance_out_tex,
                           ImgU32 candidate_normal_out_tex, ImgH4 candidate_hit_out_tex, ImgR8 invalidity_in_tex, ImgR8 invalidity_out_tex, int row0, int row1, uint2 kj_tb, uint32_t* lds_stack) {
    RAY_TILE_XY(invalidity_out_tex.w, invalidity_out_tex.h, QUAD, kj_tb)
    if (!in_image) return;
    const FrameConstants& fc = *c.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = c.depth.ld(hx, hy);
    if (0.0f == depth) {
        if (lead) {
            st4(candidate_irradiance_out_tex, x, y, v4(0.0f));
            candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(V4{0, 0, 1, 0}));
            invalidity_out_tex.st(x, y, 0);
        }
        return;
    }
    const V4 gts = tex_size4(c.depth.w, c.depth.h);
    const V2 uv = get_uv(float(hx), float(hy), gts);
    const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
    const float near_field_fade_out_end = -vr.hit_vs.z * (SSGI_NEAR_FIELD_RADIUS * gts.w * 0.5f);
    const bool tracing_frame = !is_rtdgi_validation_frame(fc.frame_index);
    {
        const V3 normal_ws = direction_view_to_world(fc, ld_nrm_snorm8(half_view_normal_tex, x, y));
        const Basis tangent_to_world = build_orthonormal_basis(normal_ws);
        const V4 bn = blue_noise_for_pixel(c.blue_noise, x, y, fc.frame_index);
        const V3 outgoing_dir = to_world(tangent_to_world, uniform_sample_hemisphere(V2{bn.x, bn.y}));
        const V3 origin = vr.biased_secondary_ray_origin_ws_with_normal(normal_ws);
        uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index & 31u);
        TraceResult result = trace_candidate<STATS, QUAD>(c, x, y, normal_ws, rng, origin, outgoing_dir, tracing_frame ? SKY_DIST : near_field_fade_out_end, lds_stack + lane);
        if (!tracing_frame && !result.is_hit) { result.out_value = v3(0.0f); result.hit_t = SKY_DIST; }
        const V3 hit_offset_ws = outgoing_dir * result.hit_t;
        const float cos_theta = dot(normalize(outgoing_dir - vr.dir_ws), normal_ws);
        if (lead) {
            st4(candidate_irradiance_out_tex, x, y, v4(result.out_value, 1.0f - cos_theta));
            st4(candidate_hit_out_tex, x, y, v4(hit_offset_ws, result.pdf * (tracing_frame ? 1.0f : -1.0f)));
            candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(v4(direction_world_to_view(fc, result.hit_normal_ws), 0)));
        }
    }
    if (COPY_VALIDITY) {
        const V4 reproj = ld_reproj(reprojection_tex, hx, hy);
        const int rx = int(floorf(float(x) + gts.x * reproj.x / 2 + 0.5f)), ry = int(floorf(float(y) + gts.y * reproj.y / 2 + 0.5f));
        if (lead) invalidity_out_tex.st(x, y, invalidity_in_tex.ld(rx, ry));
    }
}
template <bool STATS, bool QUAD = false>
__global__ void __launch_bounds__(64, KJ_FUSED_WAVES) k_rtdgi_trace_fused(TraceCtx c, ImgU32 half_view_normal_tex, ImgU2 reprojection_tex, ImgH4 candidate_irradiance_out_tex,
                                                     ImgU32 candidate_normal_out_tex, ImgH4 candidate_hit_out_tex, ImgR8 invalidity_in_tex, ImgR8 invalidity_out_tex, int row0, int row1) {
    extern __shared__ uint32_t lds_stack[];
    rtdgi_trace_tile<STATS, QUAD, true>(c, half_view_normal_tex, reprojection_tex, candidate_irradiance_out_tex, candidate_normal_out_tex, candidate_hit_out_tex, invalidity_in_tex, invalidity_out_tex,
                                        row0, row1, kj::tile_order<KJ_TILES_PLAIN>(), lds_stack);
}

// The two ray passes of a VALIDATION frame as ONE launch (round 6). On such a frame (one in three) `rtdgi validate` re-traces every reservoir's path with full-length rays and
// `rtdgi trace` walks short near-field rays; at 1080p each launch lasts exactly as long as its slowest wave (289 us and 181 us against 140 / 74 us of wave time per slot:
// profiles/r06_ray_tile_order.md) and runs its second half on a nearly empty chip. The passes do not depend on one another except for the trace pass' last statement, the copy of the
// validate pass' output at the reprojected pixel, which moves into a small launch behind both (k_rtdgi_validity_reproject). Here the validate tiles are dispatched first (the long
// chains), the trace tiles fill the slots they leave: rows [0, tiles_y) of the grid are validate tiles, [tiles_y, 2 tiles_y) trace tiles. Same functions on the same inputs, same
// outputs; with the racy cache the two passes' lookups race with one another as each pass' lookups already do among themselves.
struct ValidateTraceArgs {
    ImgU32 half_view_normal_tex;
    ImgU2 reservoir_tex; ImgH4 reservoir_ray_history_tex, irradiance_history_tex; ImgF4 ray_orig_history_tex; ImgR8 validity_pre_tex;                       // validate
    ImgU2 reprojection_tex; ImgH4 candidate_irradiance_out_tex; ImgU32 candidate_normal_out_tex; ImgH4 candidate_hit_out_tex; ImgR8 validity_in_tex;    // trace
    int row0, row1;
    uint32_t tiles_y, trace_request_slot_base, trace_request_key_base, order;
};
#ifndef KJ_FUSED_BOTH_WAVES
#define KJ_FUSED_BOTH_WAVES KJ_FUSED_WAVES
#endif
template <bool STATS>
__global__ void __launch_bounds__(64, KJ_FUSED_BOTH_WAVES) k_rtdgi_validate_and_trace(TraceCtx c, ValidateTraceArgs a) {
    extern __shared__ uint32_t lds_stack[];
    // dispatch order (workgroups are handed out x-fastest, then y): 0 = every validate tile, then every trace tile (the default); 1 = row by row, validate row then trace row;
    // 2 = the trace tiles first (KJ_RTDGI_FUSE_ORDER, measured: profiles/r06_ray_tile_order.md)
    const bool validate = a.order == 1u ? (blockIdx.y & 1u) == 0u : a.order == 2u ? blockIdx.y >= a.tiles_y : blockIdx.y < a.tiles_y;
    const uint32_t tile_row = a.order == 1u ? blockIdx.y >> 1 : blockIdx.y >= a.tiles_y ? blockIdx.y - a.tiles_y : blockIdx.y;
    if (validate) {
        rtdgi_validate_tile<STATS, false>(c, a.half_view_normal_tex, a.reservoir_tex, a.reservoir_ray_history_tex, a.irradiance_history_tex, a.ray_orig_history_tex, a.validity_pre_tex, a.row0, a.row1,
                                          make_uint2(blockIdx.x, tile_row), lds_stack);
    } else {
        c.request_slot_base = a.trace_request_slot_base; c.request_key_base = a.trace_request_key_base;
        rtdgi_trace_tile<STATS, false, false>(c, a.half_view_normal_tex, a.reprojection_tex, a.candidate_irradiance_out_tex, a.candidate_normal_out_tex, a.candidate_hit_out_tex, a.validity_pre_tex,
                                              a.validity_in_tex, a.row0, a.row1, make_uint2(blockIdx.x, tile_row), lds_stack);
    }
}
// trace_diffuse.rgen.hlsl:119: rt_history_validity_input_tex[px] = rt_history_validity_pre_input_tex[reprojected px], for the pixels whose trace pass left it out (sky pixels hold the 0 the pass stored)
__global__ void __launch_bounds__(64) k_rtdgi_validity_reproject(const FrameConstants* __restrict__ fcp, ImgF32 depth_tex, ImgU2 reprojection_tex, ImgR8 invalidity_in_tex, ImgR8 invalidity_out_tex, int row0, int row1) {
    TILE_XY_M(invalidity_out_tex.w, invalidity_out_tex.h, KJ_TILES_ROWS)
    if (!in_image) return;
    const I2 off = halfres_subsample_offset(fcp->frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    if (0.0f == depth_tex.ld(hx, hy)) return;
    const V4 gts = tex_size4(depth_tex.w, depth_tex.h);
    const V4 reproj = ld_reproj(reprojection_tex, hx, hy);
    const int rx = int(floorf(float(x) + gts.x * reproj.x / 2 + 0.5f)), ry = int(floorf(float(y) + gts.y * reproj.y / 2 + 0.5f));
    invalidity_out_tex.st(x, y, invalidity_in_tex.ld(rx, ry));
}

#ifdef KJ_RAY_PASS_EXPERIMENTS      // measured-and-rejected forms of the ray passes (make EXPERIMENTS=1): the pool form below (1.4-1.9x slower than the fused form: profiles/r05_ray_pass_experiments.md) and rtdgi_ray_experiments.inc
// ------------------------------------------------------------------ the POOL form of the two ray passes (round 5)
// The fused kernels above give every 8x8 tile a wave that lives as long as its slowest pixel: sky pixels never start, two rays of three leave the scene after
// a short walk, a sixth of the lanes shade a hit and walk a shadow ray -- 34 % of the lanes of an issued instruction do work (PMC, rounds 2-4). Here a launch is a few
// PERSISTENT waves per SIMD, each working through a strided list of tiles, and a lane is a slot that holds one pixel's job in one of four states:
//     idle -> [raygen] -> closest-hit walk -> miss: [retire: sky, outputs] -> idle
//                                         -> hit : [shade A: G-buffer of the hit, the sun's shadow ray] -> occlusion walk -> [shade B: the rest of
//                                                  diffuse_trace_common.inc.hlsl:80-200, outputs] -> idle
// Every iteration the wave votes (ballots) on ONE block to issue: refill (retire misses + ray generation for the free lanes, from the tile list), shade A, shade B, or
// a traversal step (node or triangle block, by majority, as in bvh_trace) in which closest-hit and occlusion rays of different pixels walk side by side. A shading
// block waits until enough lanes want it (tunable thresholds), so hit shading runs on fuller waves; free lanes are refilled before the wave's other rays have finished.
// Each pixel's arithmetic is that of the fused kernels, expression for expression (same device functions, same rng streams, FP contraction off in this unit), so every
// output is bit-identical to theirs -- tests/test_gpu_parity.py::test_ray_pass_forms_agree -- and only the interleaving differs; with the racy cache the order of the
// lookups' atomics differs, as it does from run to run of the fused form.
// LDS per wave: the traversal stacks ([level][lane]) + KJ_POOL_REC dwords per lane ([field][lane]) for what a job carries across its shadow walk.
#define KJ_POOL_REC 13u           // ray origin 3, direction 3, hit t, packed G-buffer 4 | written at ray generation: pdf, 1 - cos(theta)
#ifndef KJ_POOL_WAVES
#define KJ_POOL_WAVES 4           // waves per SIMD the kernel is compiled for (its VGPR budget); the launch picks how many it actually starts
#endif
struct PoolArgs {
    ImgU32 half_view_normal_tex;
    ImgU2 reprojection_tex; ImgH4 candidate_irradiance_out_tex; ImgU32 candidate_normal_out_tex; ImgH4 candidate_hit_out_tex; ImgR8 invalidity_in_tex;     // trace pass
    ImgU2 reservoir_tex; ImgH4 reservoir_ray_history_tex; ImgH4 irradiance_history_tex; ImgF4 ray_orig_history_tex;                                      // validate pass
    ImgR8 invalidity_out_tex;
    int row0, row1;
    uint32_t tiles_x, tiles_y;
    uint32_t refill_min, shade_a_min, shade_b_min;     // lanes that must be waiting before the block is issued (while other lanes still walk)
    uint32_t* tile_counter;                            // non-null: tiles beyond a wave's first are handed out by this counter instead of by stride
};
enum : uint32_t { KJ_PH_IDLE = 0u, KJ_PH_CLOSEST = 1u, KJ_PH_SHADOW = 2u, KJ_PH_NOSUN = 3u };
KJ_D uint32_t pool_first_lane(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
#else
    return __shfl(v, 0);
#endif
}
KJ_HD size_t pool_lds_bytes(uint32_t stack_entries) { return (size_t(stack_entries) + KJ_POOL_REC) * 64u * 4u; }

// what trace_diffuse.rgen.hlsl:103-118 stores for a pixel whose candidate is known
KJ_D void pool_finish_trace(const FrameConstants& fc, const PoolArgs& a, int x, int y, bool tracing_frame, V3 outgoing_dir, V3 out_value, V3 hit_normal_ws, float hit_t, float pdf, float one_minus_cos) {
    const V3 hit_offset_ws = outgoing_dir * hit_t;
    st4(a.candidate_irradiance_out_tex, x, y, v4(out_value, one_minus_cos));
    st4(a.candidate_hit_out_tex, x, y, v4(hit_offset_ws, pdf * (tracing_frame ? 1.0f : -1.0f)));
    a.candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(v4(direction_world_to_view(fc, hit_normal_ws), 0)));
}
// diffuse_validate.rgen.hlsl:84-110 for a pixel whose re-traced radiance is known
KJ_D void pool_finish_validate(const PoolArgs& a, int x, int y, V3 out_value, float hit_t) {
    const float4 ro = a.ray_orig_history_tex.ld(x, y);
    const V3 prev_ray_orig{ro.x, ro.y, ro.z};
    const V3 prev_hit_pos = xyz(ld4(a.reservoir_ray_history_tex, x, y)) + prev_ray_orig;
    const V4 prev_radiance_packed = ld4(a.irradiance_history_tex, x, y);
    const V3 prev_radiance = vmax(v3(0.0f), xyz(prev_radiance_packed));
    const V3 new_radiance = vmax(v3(0.0f), out_value);
    const float rad_diff = length(vabs(prev_radiance - new_radiance) / vmax(v3(1e-3f), prev_radiance + new_radiance));
    const float invalidity = smoothstep(0.1f, 0.5f, rad_diff / length(v3(1.0f)));
    const float prev_hit_dist = length(prev_hit_pos - prev_ray_orig);
    if (fabsf(hit_t - prev_hit_dist) / (prev_hit_dist + prev_hit_dist) < 0.2f) {
        st4(a.irradiance_history_tex, x, y, v4(new_radiance, prev_radiance_packed.w));
        Reservoir1spp r = Reservoir1spp::from_raw(a.reservoir_tex.ld(x, y));
        const float lum_old = sRGB_to_luminance(prev_radiance), lum_new = sRGB_to_luminance(new_radiance);
        r.M *= clampf(lum_old / fmaxf(1e-8f, lum_new), 0.03f, 1.0f);
        r.W *= clampf(lum_old / fmaxf(1e-8f, lum_new) * 10.0f, 0.01f, 1.0f);
        a.reservoir_tex.st(x, y, r.as_raw());
    }
    a.invalidity_out_tex.st(x, y, to_unorm8(invalidity));
}

template <bool VALIDATE, bool STATS>
__global__ void __launch_bounds__(64, KJ_POOL_WAVES) k_rtdgi_rays_pool(TraceCtx c, PoolArgs a) {
    extern __shared__ uint32_t lds_pool[];
    const uint32_t NONE = KJ_BVH_NONE;
    const uint32_t lane = threadIdx.x;
    const unsigned long long lane_bit = 1ull << lane;
    uint32_t* stack = lds_pool + lane;
    uint32_t* rec = lds_pool + c.sc.bvh.stack_entries * 64u + lane;
    const FrameConstants& fc = *c.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const bool tracing_frame = !is_rtdgi_validation_frame(fc.frame_index);
    const int W = a.invalidity_out_tex.w, H = a.invalidity_out_tex.h;
    const int y_end = H < a.row1 ? H : a.row1;
    const uint32_t n_tiles = a.tiles_x * a.tiles_y;
    uint32_t tile = blockIdx.x, cursor = 0;
    bool exhausted = tile >= n_tiles;
    RayState S;
    S.cur = NONE; S.sp = 0; S.cull_back = false;
    S.h.t = FLT_MAX; S.h.u = S.h.v = 0; S.h.slot = S.h.world_id = 0xffffffffu;
    S.wo = S.wd = S.binv = v3(0.0f); S.tmin = S.tmax = 0.0f;
    uint32_t spill[KJ_BVH_SPILL_STACK];
    uint32_t phase = KJ_PH_IDLE, pix = 0, rng = 0;
    uint32_t n_closest = 0, n_any = 0;      // wave-uniform ray counts, added to the device counters once
    TraverseStats st_closest{0, 0}, st_any{0, 0};
    uint32_t it_node = 0, it_tri = 0, it_refill = 0, it_a = 0, it_b = 0, lanes_a = 0, lanes_b = 0;      // STATS: blocks this wave issued, lanes that worked in the shading blocks
    for (;;) {
        const bool walking = S.cur != NONE;
        const bool want_node = walking && !(S.cur & KJ_BVH_LEAF), want_tri = walking && (S.cur & KJ_BVH_LEAF) != 0u;
        const bool done_closest = phase == KJ_PH_CLOSEST && !walking;
        const bool is_miss = done_closest && S.h.slot == 0xffffffffu, wait_a = done_closest && S.h.slot != 0xffffffffu;
        const bool wait_b = (phase == KJ_PH_SHADOW || phase == KJ_PH_NOSUN) && !walking;
        const uint32_t nn = uint32_t(__popcll(__ballot(want_node))), nt = uint32_t(__popcll(__ballot(want_tri)));
        const uint32_t n_miss = uint32_t(__popcll(__ballot(is_miss))), n_free = n_miss + uint32_t(__popcll(__ballot(phase == KJ_PH_IDLE)));
        const uint32_t n_a = uint32_t(__popcll(__ballot(wait_a))), n_b = uint32_t(__popcll(__ballot(wait_b)));
        const uint32_t n_walk = nn + nt;
        // in the tail (no tiles left) a shading block goes as soon as its lanes are as many as the walking ones
        const uint32_t floor_ = n_walk > 1u ? n_walk : 1u;
        const uint32_t thr_a = exhausted ? (a.shade_a_min < floor_ ? a.shade_a_min : floor_) : a.shade_a_min;
        const uint32_t thr_b = exhausted ? (a.shade_b_min < floor_ ? a.shade_b_min : floor_) : a.shade_b_min;
        int block;      // 0 refill, 1 shade A, 2 shade B, 3 walk
        if (!exhausted && n_free >= a.refill_min) block = 0;
        else if (n_b != 0u && n_b >= thr_b) block = 2;
        else if (n_a != 0u && n_a >= thr_a) block = 1;
        else if (n_walk != 0u) block = 3;
        else if (n_b != 0u) block = 2;
        else if (n_a != 0u) block = 1;
        else if (n_miss != 0u || (!exhausted && n_free != 0u)) block = 0;
        else break;

        if (block == 3) {
            TraverseStats* st = phase == KJ_PH_CLOSEST ? &st_closest : &st_any;
            if (nt == 0u || (nn != 0u && nn >= nt * 2u)) { if (STATS) it_node++; if (want_node) node_step<false, STATS>(c.sc.bvh, S, stack, 64, spill, st); }
            else { if (STATS) it_tri++; if (want_tri) tri_step_mixed<STATS>(c.sc.bvh, S, phase != KJ_PH_CLOSEST, stack, 64, spill, st); }
        } else if (block == 0) {
            if (STATS) it_refill++;
            if (is_miss) {      // diffuse_trace_common.inc.hlsl:201-207: the sky
                const int x = int(pix & 0xffffu), y = int(pix >> 16);
                const V3 ray_d = S.wd;
                V3 total_radiance = v3(0.0f);
                total_radiance += xyz(sample_cube_rgba16f(c.sky_cube, c.sky_cube_width, ray_d));
                if (VALIDATE) pool_finish_validate(a, x, y, total_radiance, SKY_DIST);
                else {
                    if (!tracing_frame) total_radiance = v3(0.0f);
                    pool_finish_trace(fc, a, x, y, tracing_frame, ray_d, total_radiance, -ray_d, SKY_DIST, __uint_as_float(rec[11u * 64u]), __uint_as_float(rec[12u * 64u]));
                }
                phase = KJ_PH_IDLE;
            }
            if (!exhausted) {
                const unsigned long long im = __ballot(phase == KJ_PH_IDLE);
                const uint32_t n_idle = uint32_t(__popcll(im)), left = 64u - cursor;
                const uint32_t take = n_idle < left ? n_idle : left;
                const uint32_t rank = uint32_t(__popcll(im & (lane_bit - 1ull)));
                bool started = false;
                if (phase == KJ_PH_IDLE && rank < take) {
                    const uint32_t p = cursor + rank, ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
                    const int x = int(tx * 8u + (p & 7u)), y = a.row0 + int(ty * 8u + (p >> 3));
                    if (x < W && y < y_end) {
                        const int hx = x * 2 + off.x, hy = y * 2 + off.y;
                        const float depth = c.depth.ld(hx, hy);
                        if (VALIDATE) {
                            if (0.0f == depth) a.invalidity_out_tex.st(x, y, to_unorm8(1.0f));
                            else {
                                const float4 ro = a.ray_orig_history_tex.ld(x, y);
                                const V3 prev_ray_orig{ro.x, ro.y, ro.z};
                                const V3 prev_hit_pos = xyz(ld4(a.reservoir_ray_history_tex, x, y)) + prev_ray_orig;
                                rng = hash3(uint32_t(x), uint32_t(y), 0);
                                ray_begin<false>(S, prev_ray_orig, normalize(prev_hit_pos - prev_ray_orig), 0.0f, SKY_DIST, false);
                                started = true;
                            }
                        } else if (0.0f == depth) {
                            st4(a.candidate_irradiance_out_tex, x, y, v4(0.0f));
                            a.candidate_normal_out_tex.st(x, y, pack_rgba8_snorm(V4{0, 0, 1, 0}));
                            a.invalidity_out_tex.st(x, y, 0);
                        } else {
                            const V4 gts = tex_size4(c.depth.w, c.depth.h);
                            const V2 uv = get_uv(float(hx), float(hy), gts);
                            const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
                            const float near_field_fade_out_end = -vr.hit_vs.z * (SSGI_NEAR_FIELD_RADIUS * gts.w * 0.5f);
                            const V3 normal_ws = direction_view_to_world(fc, ld_nrm_snorm8(a.half_view_normal_tex, x, y));
                            const Basis tangent_to_world = build_orthonormal_basis(normal_ws);
                            const V4 bn = blue_noise_for_pixel(c.blue_noise, x, y, fc.frame_index);
                            const V3 outgoing_dir = to_world(tangent_to_world, uniform_sample_hemisphere(V2{bn.x, bn.y}));
                            const V3 origin = vr.biased_secondary_ray_origin_ws_with_normal(normal_ws);
                            rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index & 31u);
                            const float pdf = fmaxf(0.0f, 1.0f / (dot(normal_ws, outgoing_dir) * 2 * KJ_PI));
                            const float cos_theta = dot(normalize(outgoing_dir - vr.dir_ws), normal_ws);
                            rec[11u * 64u] = __float_as_uint(pdf);
                            rec[12u * 64u] = __float_as_uint(1.0f - cos_theta);
                            ray_begin<false>(S, origin, outgoing_dir, 0.0f, tracing_frame ? SKY_DIST : near_field_fade_out_end, false);
                            started = true;
                            const V4 reproj = ld_reproj(a.reprojection_tex, hx, hy);
                            const int rx = int(floorf(float(x) + gts.x * reproj.x / 2 + 0.5f)), ry = int(floorf(float(y) + gts.y * reproj.y / 2 + 0.5f));
                            a.invalidity_out_tex.st(x, y, a.invalidity_in_tex.ld(rx, ry));
                        }
                        if (started) { pix = uint32_t(x) | (uint32_t(y) << 16); phase = KJ_PH_CLOSEST; }
                    }
                }
                n_closest += uint32_t(__popcll(__ballot(started)));
                cursor += take;
                if (cursor == 64u) {
                    cursor = 0;
                    if (a.tile_counter) {
                        uint32_t t = 0;
                        if (lane == 0u) t = atomicAdd(a.tile_counter, 1u) + gridDim.x;
                        tile = pool_first_lane(t);
                    } else tile += gridDim.x;
                    exhausted = tile >= n_tiles;
                }
            }
        } else if (block == 1) {
            if (STATS) { it_a++; lanes_a += n_a; }
            bool sun_ray = false;
            if (wait_a) {       // rt/gbuffer.rchit.hlsl + diffuse_trace_common.inc.hlsl:80-130 up to the sun's shadow ray
                const int x = int(pix & 0xffffu), y = int(pix >> 16);
                const V3 ray_o = S.wo, ray_d = S.wd;
                const RayHit h = S.h;
                const uint4 gp = shade_gbuffer_hit(c.sc, fc, ray_d, h, 1, candidate_ray_cone(c, ray_o).width_at_t(h.t * length(ray_d)));
                const V3 position = mad_nc(ray_o, ray_d, h.t);
                rec[0u * 64u] = __float_as_uint(ray_o.x); rec[1u * 64u] = __float_as_uint(ray_o.y); rec[2u * 64u] = __float_as_uint(ray_o.z);
                rec[3u * 64u] = __float_as_uint(ray_d.x); rec[4u * 64u] = __float_as_uint(ray_d.y); rec[5u * 64u] = __float_as_uint(ray_d.z);
                rec[6u * 64u] = __float_as_uint(h.t);
                rec[7u * 64u] = gp.x; rec[8u * 64u] = gp.y; rec[9u * 64u] = gp.z; rec[10u * 64u] = gp.w;
                const float4 sc4 = *c.sun_color;
                if (sc4.x != 0 || sc4.y != 0 || sc4.z != 0) {
                    const V4 bn = blue_noise_for_pixel(c.blue_noise, uint32_t(x), uint32_t(y), rng);
                    const V3 to_light_norm = sample_sun_direction(fc, V2{bn.x, bn.y}, false);
                    ray_begin<true>(S, position, to_light_norm, 1e-4f, SKY_DIST, false);
                    phase = KJ_PH_SHADOW;
                    sun_ray = true;
                } else { S.cur = NONE; phase = KJ_PH_NOSUN; }
            }
            n_any += uint32_t(__popcll(__ballot(sun_ray)));
        } else {
            if (STATS) { it_b++; lanes_b += n_b; }
            if (wait_b) {       // the rest of diffuse_trace_common.inc.hlsl:80-200 with the shadow ray's answer, then the pass' stores
                const int x = int(pix & 0xffffu), y = int(pix >> 16);
                const bool sun_is_shadowed = S.h.slot != 0xffffffffu;
                const V3 ray_o{__uint_as_float(rec[0u * 64u]), __uint_as_float(rec[1u * 64u]), __uint_as_float(rec[2u * 64u])};
                const V3 ray_d{__uint_as_float(rec[3u * 64u]), __uint_as_float(rec[4u * 64u]), __uint_as_float(rec[5u * 64u])};
                GbufferPathVertex pv;
                pv.is_hit = true;
                pv.ray_t = __uint_as_float(rec[6u * 64u]);
                pv.gbuffer_packed = make_uint4(rec[7u * 64u], rec[8u * 64u], rec[9u * 64u], rec[10u * 64u]);
                pv.position = mad_nc(ray_o, ray_d, pv.ray_t);
                V3 hit_normal_ws;
                const V3 radiance = shade_candidate_hit_with<STATS>(c, uint32_t(x), uint32_t(y), rng, ray_o, ray_d, pv, stack, 64, &st_any, hit_normal_ws,
                                                                    [&](V3, V3) -> bool { return sun_is_shadowed; });
                if (VALIDATE) pool_finish_validate(a, x, y, radiance, pv.ray_t);
                else pool_finish_trace(fc, a, x, y, tracing_frame, ray_d, radiance, hit_normal_ws, pv.ray_t, __uint_as_float(rec[11u * 64u]), __uint_as_float(rec[12u * 64u]));
                phase = KJ_PH_IDLE;
            }
        }
    }
    if (lane == 0u) {
        if (n_closest) atomicAdd(&counter_slot(c.ray_counters)[0], (unsigned long long)n_closest);
        if (n_any) atomicAdd(&counter_slot(c.ray_counters)[1], (unsigned long long)n_any);
        if (STATS) {    // [6], [7]: node / triangle steps this wave issued (closest-hit and occlusion rays walk together here); [10..14]: refill / shade A / shade B blocks, lanes in A, lanes in B
            atomicAdd(&counter_slot(c.ray_counters)[6], (unsigned long long)it_node); atomicAdd(&counter_slot(c.ray_counters)[7], (unsigned long long)it_tri);
            atomicAdd(&counter_slot(c.ray_counters)[10], (unsigned long long)it_refill); atomicAdd(&counter_slot(c.ray_counters)[11], (unsigned long long)it_a);
            atomicAdd(&counter_slot(c.ray_counters)[12], (unsigned long long)it_b); atomicAdd(&counter_slot(c.ray_counters)[13], (unsigned long long)lanes_a);
            atomicAdd(&counter_slot(c.ray_counters)[14], (unsigned long long)lanes_b);
        }
    }
    add_traversal_stats<STATS>(c, st_closest, st_any);
}

#include "rtdgi_ray_experiments.inc"
#endif

// ------------------------------------------------------------------ temporal_validity_integrate.hlsl:21-119
// WaveReadLaneAt(v, lane^k) inside the 8x8 group == __shfl_xor(v, k) on wave64.
struct ValidityIntegrateArgs { const FrameConstants* __restrict__ fc; ImgR8 input_tex; ImgU32 history_tex /*RG16F*/; ImgU2 reprojection_tex; ImgF32 half_depth_tex; ImgU32 output_tex /*RG16F*/; int W, H, row0, row1; };
KJ_D void validity_integrate_body(const ValidityIntegrateArgs& v) {
    const FrameConstants* fcp = v.fc;
    const ImgR8& input_tex = v.input_tex; const ImgU32& history_tex = v.history_tex; const ImgU2& reprojection_tex = v.reprojection_tex;
    const ImgF32& half_depth_tex = v.half_depth_tex; const ImgU32& output_tex = v.output_tex;
    const int W = v.W, H = v.H, row0 = v.row0, row1 = v.row1;
    TILE_XY_M(output_tex.w, output_tex.h, KJ_TILES_ROWS)
    const FrameConstants& fc = *fcp;
    V2 invalid_blurred{0, 0};
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const float w = exp2f(-0.1f * float(dx * dx + dy * dy));
            invalid_blurred += V2{from_unorm8(input_tex.ld(x + dx, y + dy)), 1.0f} * w;
        }
    invalid_blurred = invalid_blurred / invalid_blurred.y;
    float ib = invalid_blurred.x;
    ib = lerp(ib, __shfl_xor(ib, 2), 0.5f);
    ib = lerp(ib, __shfl_xor(ib, 16), 0.5f);
    ib = smoothstep(0.0f, 1.0f, ib);
    const float center_depth = half_depth_tex.ld(x, y);
    // the shader's loop leaves its inner loop at the first failed texel with edge = 0, and edge only ever takes the values 0 and 1:
    // the result is 1 exactly when all six texels pass both tests, so all twelve loads are issued together and nothing branches
    bool edge_ok = true;
#pragma unroll
    for (int oy = 0; oy <= 2; ++oy)
#pragma unroll
        for (int ox = 1; ox <= 2; ++ox) {
            const V4 reproj = ld_reproj(reprojection_tex, x * 2 + ox, y * 2 + oy);
            const float sample_depth = half_depth_tex.ld(x + ox / 2, y + oy / 2);
            // (`&`, not `&&`: a short-circuit makes every texel's loads wait for the previous texel's verdict)
            edge_ok = edge_ok & !((reproj.w < 0) | (inverse_depth_relative_diff(center_depth, sample_depth) > 0.1f)) & ((reproj.z == 0) & (sample_depth != 0));
        }
    float edge = edge_ok ? 1.0f : 0.0f;
    edge = fmaxf(edge, __shfl_xor(edge, 1));
    edge = fmaxf(edge, __shfl_xor(edge, 8));
    ib = saturate(ib + edge);
    const V4 reproj = ld_reproj(reprojection_tex, x * 2, y * 2);
    const V2 reproj_px{float(x) + float(W) * reproj.x / 2 + 0.5f, float(y) + float(H) * reproj.y / 2 + 0.5f};
    float history = 0;
    const float ang_off = uint_to_u01_float(hash3(uint32_t(x), uint32_t(y), fc.frame_index)) * KJ_PI * 2;
    uint32_t hist_raw[8]; bool hist_in[8];
#pragma unroll
    for (uint32_t si = 0; si < 8u; ++si) {
        const float ang = (float(si) + ang_off) * KJ_GOLDEN_ANGLE;
        const float radius = float(si) * 1.0f;
        const V2 so = cos_sin_turns_fast(ang) * radius;      // same reduction in revolutions, < 1 ulp (kj_screen.hpp): libm's sinf + cosf are 235 instructions per tap
        hist_raw[si] = history_tex.ld_raw(int(reproj_px.x + so.x), int(reproj_px.y + so.y), hist_in[si]);      // the eight taps in flight together
    }
#pragma unroll
    for (uint32_t si = 0; si < 8u; ++si) history += unpack_2x16f_uint(hist_in[si] ? hist_raw[si] : 0u).x;
    history /= 8;
    if (in_image) st2h(output_tex, x, y, V2{fmaxf(history * 0.75f, ib), from_unorm8(input_tex.ld(x, y))});
}
__global__ void __launch_bounds__(64) k_validity_integrate(ValidityIntegrateArgs v) { validity_integrate_body(v); }

// ------------------------------------------------------------------ restir_temporal.hlsl:83-422
struct RestirTemporalArgs {
    const FrameConstants* __restrict__ fc;
    ImgF32 depth_tex; ImgU32 half_view_normal_tex; ImgH4 candidate_radiance_tex; ImgU32 candidate_normal_tex; ImgH4 candidate_hit_tex;
    ImgH4 radiance_history_tex; ImgF4 ray_orig_history_tex; ImgH4 ray_history_tex; ImgU2 reservoir_history_tex; ImgU2 reprojection_tex;
    ImgH4 hit_normal_history_tex; ImgH4 candidate_history_tex; ImgU32 rt_invalidity_tex;
    ImgH4 radiance_out_tex; ImgF4 ray_orig_output_tex; ImgH4 ray_output_tex; ImgH4 hit_normal_output_tex; ImgU2 reservoir_out_tex;
    ImgH4 candidate_out_tex; ImgU4 temporal_reservoir_packed_tex;
    int row0, row1;
};
KJ_D void restir_temporal_body(const RestirTemporalArgs& a) {
    const int row0 = a.row0, row1 = a.row1;
    TILE_XY_M(a.reservoir_out_tex.w, a.reservoir_out_tex.h, KJ_TILES_ROWS)
    if (!in_image) return;
    const FrameConstants& fc = *a.fc;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int hx = x * 2 + off.x, hy = y * 2 + off.y;
    const float depth = a.depth_tex.ld(hx, hy);
    if (0.0f == depth) {
        st4(a.radiance_out_tex, x, y, V4{0, 0, 0, -SKY_DIST});
        st4(a.hit_normal_output_tex, x, y, v4(0.0f));
        a.reservoir_out_tex.st(x, y, make_uint2(0, 0));
        return;
    }
    const V4 gts = tex_size4(a.depth_tex.w, a.depth_tex.h);
    const V2 uv = get_uv(float(hx), float(hy), gts);
    const ViewRay vr = view_ray_from_uv_and_biased_depth(fc, uv, depth);
    const V3 normal_vs = ld_nrm_snorm8(a.half_view_normal_tex, x, y);
    const V3 normal_ws = direction_view_to_world(fc, normal_vs);
    const V3 refl_ray_origin_ws = vr.biased_secondary_ray_origin_ws_with_normal(normal_ws);
    const V3 hit_offset_ws = xyz(ld4(a.candidate_hit_tex, x, y));
    V3 outgoing_dir = normalize(hit_offset_ws);
    uint32_t rng = hash3(uint32_t(x), uint32_t(y), fc.frame_index);
    V3 radiance_sel = v3(0.0f), ray_orig_sel_ws = v3(0.0f), ray_hit_sel_ws = v3(1.0f), hit_normal_sel = v3(1.0f);
    StreamState stream_state{0, 0};
    Reservoir1spp reservoir = Reservoir1spp::create();
    const uint32_t reservoir_payload = uint32_t(x) | (uint32_t(y) << 16);
    const bool tracing_frame = !is_rtdgi_validation_frame(fc.frame_index);
    if (tracing_frame) {
        const float hit_t = length(hit_offset_ws);
        const V3 out_value = xyz(ld4(a.candidate_radiance_tex, x, y));
        const float p_q = 1.0f * fmaxf(0.0f, sRGB_to_luminance(out_value)) * stepf(0.0f, dot(outgoing_dir, normal_ws));
        radiance_sel = out_value;
        ray_orig_sel_ws = refl_ray_origin_ws;
        ray_hit_sel_ws = refl_ray_origin_ws + outgoing_dir * hit_t;
        hit_normal_sel = direction_view_to_world(fc, ld_nrm_snorm8(a.candidate_normal_tex, x, y));
        reservoir.init_with_stream(p_q, 1.0f, stream_state, reservoir_payload);
        const float rl = lerp(ld4(a.candidate_history_tex, x, y).y, sqrtf(hit_t), 0.05f);
        st4(a.candidate_out_tex, x, y, V4{sqrtf(hit_t), rl, 0, 0});
    }
    const float rt_invalidity = sqrtf(saturate(ld2h(a.rt_invalidity_tex, x, y).y));
    float center_M = 0;
    const uint32_t fi = fc.frame_index;
    // xor_seq[frame&3] = {(3,3),(2,1),(1,2),(3,3)} ; offsets[4] = {(-1,-1),(1,1),(-1,1),(1,-1)}
    const uint32_t pxv_x = (fi & 3u) == 1u ? 2u : ((fi & 3u) == 2u ? 1u : 3u);
    const uint32_t pxv_y = (fi & 3u) == 1u ? 1u : ((fi & 3u) == 2u ? 2u : 3u);
    // The history taps, walked one after the other as in the shader: the loop stops once the accumulated M passes 1.25 x the clamp, with
    // converged history after the second tap. Measured alternatives (profiles/r03_screen_chain.md): fetching all five taps level by level
    // up front (52 us against the loop's 37: three taps' worth of scattered loads nobody uses, 120 registers) and prefetching the first
    // two (48 us). What did pay: pow(x, 4) as two squarings (libm's powf is 163 instructions per tap) and the single-instruction
    // reciprocal / rsqrt in the weights; the rejection tests, which are comparisons, keep IEEE arithmetic.
    const float M_clamp_now = exp2f(log2f(RESTIR_TEMPORAL_M_CLAMP) * (1.0f - rt_invalidity));
    for (uint32_t sample_i = 0; sample_i < 5u && stream_state.M_sum < 1.25f * RESTIR_TEMPORAL_M_CLAMP; ++sample_i) {
        I2 rpx_offset{0, 0};
        if (sample_i != 0) {
            const uint32_t ia = fi & 3u, ib = (sample_i + (fi ^ 1u)) & 3u;
            auto ofs = [](uint32_t i) { return I2{(i == 1u || i == 3u) ? 1 : -1, (i == 1u || i == 2u) ? 1 : -1}; };
            const I2 oa = ofs(ia), ob = ofs(ib);
            rpx_offset = I2{oa.x + ob.x, oa.y + ob.y};
            if (rpx_offset.x == 0 && rpx_offset.y == 0) continue;
        }
        // (round 6: a tap's gathers leave in three groups instead of one by one: [reprojection, neighbour depth, neighbour normal] -- the last two depend on the tap's
        // position only --, [the reprojected reservoir], [the four histories at the pixel the reservoir points at]; the rejection tests follow in the text's order.)
        const V2 base = sample_i == 0 ? V2{float(x), float(y)} : V2{float(uint32_t(x + rpx_offset.x) ^ pxv_x), float(uint32_t(y + rpx_offset.y) ^ pxv_y)};
        const int pnx = f2i_sat(floorf(base.x + 0.5f)), pny = f2i_sat(floorf(base.y + 0.5f));
        const I2 neighbor_px{wrap_add(pnx, rpx_offset.x), wrap_add(pny, rpx_offset.y)};
        const int nhx = wrap_mul2_add(neighbor_px.x, off.x), nhy = wrap_mul2_add(neighbor_px.y, off.y);
        bool in_rp, in_d, in_n, in_h;
        const uint2 reproj_raw = a.reprojection_tex.ld_raw(hx + rpx_offset.x * 2, hy + rpx_offset.y * 2, in_rp);
        const float sample_depth_raw = a.depth_tex.ld_raw(nhx, nhy, in_d);
        const uint32_t sample_normal_raw = a.half_view_normal_tex.ld_raw(neighbor_px.x, neighbor_px.y, in_n);
        const uint2 rpr = in_rp ? reproj_raw : make_uint2(0u, 0u);
        const V4 reproj{from_snorm16(int16_t(rpr.x & 0xffff)), from_snorm16(int16_t(rpr.x >> 16)), from_snorm16(int16_t(rpr.y & 0xffff)), from_snorm16(int16_t(rpr.y >> 16))};
        const int prx = f2i_sat(floorf(base.x + gts.x * reproj.x * 0.5f + 0.0f + 0.5f)), pry = f2i_sat(floorf(base.y + gts.y * reproj.y * 0.5f + 0.0f + 0.5f));
        const I2 rpx{wrap_add(prx, rpx_offset.x), wrap_add(pry, rpx_offset.y)};
        Reservoir1spp r = Reservoir1spp::from_raw(a.reservoir_history_tex.ld(rpx.x, rpx.y));
        const int spx_x = int(r.payload & 0xffff), spx_y = int(r.payload >> 16);
        float relevance = 1;
        const float sample_depth = in_d ? sample_depth_raw : 0.0f;
        const float4 pro_raw = a.ray_orig_history_tex.ld_raw(spx_x, spx_y, in_h);      // (the four histories share the half-res extent)
        const uint2 rh_raw = a.ray_history_tex.ld_raw(spx_x, spx_y, in_h);
        const uint2 hn_raw = a.hit_normal_history_tex.ld_raw(spx_x, spx_y, in_h);
        const uint2 prad_raw = a.radiance_history_tex.ld_raw(spx_x, spx_y, in_h);
        const float4 pro = in_h ? pro_raw : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const V3 prev_ray_orig{pro.x, pro.y, pro.z};
        if (length(prev_ray_orig - refl_ray_origin_ws) > 0.1f * -vr.hit_vs.z) continue;
        if (0 == sample_depth) continue;
        if (reproj.z == 0) continue;
        relevance *= 1 - smoothstep_fast(0.0f, 0.1f, fabsf(fmaxf(1e-20f, depth) * rcp_fast(fmaxf(1e-20f, sample_depth)) - 1.0f));
        const V3 sample_normal_vs = xyz(unpack_rgba8_snorm(in_n ? sample_normal_raw : 0u));
        const float normal_similarity_dot = fmaxf(0.0f, dot(sample_normal_vs, normal_vs));
        if (sample_i != 0 && normal_similarity_dot < 0.2f) continue;
        relevance *= square(square(normal_similarity_dot));
        const V4 rh = unpack_rgba16f(in_h ? rh_raw : make_uint2(0u, 0u));
        const V3 sample_hit_ws = xyz(rh) + prev_ray_orig;
        const float prev_dist = rh.w;
        const V4 hn = unpack_rgba16f(in_h ? hn_raw : make_uint2(0u, 0u));
        const V4 sample_hit_normal_ws_dot{hn.x * 2 - 1, hn.y * 2 - 1, hn.z * 2 - 1, hn.w};
        const V3 dir_to_sample_hit_unnorm = sample_hit_ws - refl_ray_origin_ws;
        const float inv_dist_to_sample_hit = rsq_fast(dot(dir_to_sample_hit_unnorm, dir_to_sample_hit_unnorm));
        const V3 dir_to_sample_hit = dir_to_sample_hit_unnorm * inv_dist_to_sample_hit;
        const float center_to_hit_vis = -dot(xyz(sample_hit_normal_ws_dot), dir_to_sample_hit);
        const V4 prev_rad = unpack_rgba16f(in_h ? prad_raw : make_uint2(0u, 0u)) * V4{fc.pre_exposure_delta, fc.pre_exposure_delta, fc.pre_exposure_delta, 1};
        r.M = fmaxf(0.0f, fminf(r.M, M_clamp_now));
        const float p_q = 1 * fmaxf(0.0f, sRGB_to_luminance(xyz(prev_rad))) * stepf(0.0f, dot(dir_to_sample_hit, normal_ws));
        float jacobian = 1;
        jacobian *= clampf(prev_dist * inv_dist_to_sample_hit, 1e-4f, 1e4f);
        jacobian *= jacobian;
        jacobian *= clampf(center_to_hit_vis * rcp_fast(sample_hit_normal_ws_dot.w), 0.0f, 1e4f);
        r.M *= relevance;
        if (0 == sample_i) center_M = r.M;
        if (reservoir.update_with_stream(r, p_q, jacobian * 1.0f, stream_state, reservoir_payload, rng)) {
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
    const V4 hit_normal_ws_dot = v4(hit_normal_sel, -dot(hit_normal_sel, outgoing_dir));
    st4(a.radiance_out_tex, x, y, v4(radiance_sel, dot(normal_ws, outgoing_dir)));
    a.ray_orig_output_tex.st(x, y, make_float4(ray_orig_sel_ws.x, ray_orig_sel_ws.y, ray_orig_sel_ws.z, 0.0f));
    st4(a.hit_normal_output_tex, x, y, V4{hit_normal_ws_dot.x * 0.5f + 0.5f, hit_normal_ws_dot.y * 0.5f + 0.5f, hit_normal_ws_dot.z * 0.5f + 0.5f, hit_normal_ws_dot.w});
    st4(a.ray_output_tex, x, y, v4(ray_hit_sel_ws - ray_orig_sel_ws, length(ray_hit_sel_ws - refl_ray_origin_ws)));
    a.reservoir_out_tex.st(x, y, reservoir.as_raw());
    TemporalReservoirOutput rp;
    rp.depth = depth;
    rp.ray_hit_offset_ws = ray_hit_sel_ws - vr.hit_ws;
    rp.luminance = fmaxf(0.0f, sRGB_to_luminance(radiance_sel));
    rp.hit_normal_ws = xyz(hit_normal_ws_dot);
    a.temporal_reservoir_packed_tex.st(x, y, rp.as_raw());
}
__global__ void __launch_bounds__(64) k_restir_temporal(RestirTemporalArgs a) { restir_temporal_body(a); }
// `validity integrate` + `restir temporal` in ONE launch (the default when a frame runs both): same tiles, same code, one dependent launch
// less on the frame's critical chain. The temporal pass reads `rt_invalidity_tex[px].y` -- of its own pixel only -- which the same lane
// has just stored (a thread observes its own earlier stores), so nothing needs to be exchanged between the two halves.
__global__ void __launch_bounds__(64) k_validity_integrate_restir_temporal(ValidityIntegrateArgs v, RestirTemporalArgs a) {
    validity_integrate_body(v);
    restir_temporal_body(a);
}

// restir_spatial.hlsl + occlusion_raymarch.hlsl: rtdgi_resample.hip (k_restir_spatial<>)

// ------------------------------------------------------------------ restir_check.rgen.hlsl:21-70 (use_raytraced_reservoir_visibility, rtdgi.rs:478-494)
__global__ void __launch_bounds__(64) k_restir_check(const FrameConstants* __restrict__ fcp, SceneView sc, ImgF32 half_depth_tex, ImgU4 temporal_reservoir_packed_tex,
                                                      ImgU2 reservoir_input_tex, unsigned long long* __restrict__ ray_counters, int W, int H, int row0, int row1) {
    extern __shared__ uint32_t lds_stack[];
    TILE_XY(reservoir_input_tex.w, reservoir_input_tex.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const V4 gts = tex_size4(W, H);
    const float depth = half_depth_tex.ld(x, y);
    const ViewRay vrc = view_ray_from_uv_and_biased_depth(fc, get_uv(float(x * 2 + off.x), float(y * 2 + off.y), gts), depth);
    Reservoir1spp r = Reservoir1spp::from_raw(reservoir_input_tex.ld(x, y));
    const int spx_x = int(r.payload & 0xffffu), spx_y = int(r.payload >> 16);
    const TemporalReservoirOutput spx_packed = TemporalReservoirOutput::from_raw(temporal_reservoir_packed_tex.ld(spx_x, spx_y));
    const ViewRay spx_ctx = view_ray_from_uv_and_depth(fc, get_uv(float(spx_x * 2 + off.x), float(spx_y * 2 + off.y), gts), spx_packed.depth);
    const V3 spx_pos_ws = spx_ctx.hit_ws;
    const V3 hit_ws = spx_packed.ray_hit_offset_ws + spx_pos_ws;
    const V3 trace_origin_ws = vrc.biased_secondary_ray_origin_ws();
    const V3 trace_vec = hit_ws - trace_origin_ws;
    count_rays(ray_counters, 1, true);
    if (rt_is_shadowed<false>(sc, trace_origin_ws, normalize(trace_vec), 0.0f, fminf(5.0f * length(spx_pos_ws - trace_origin_ws), length(trace_vec) * 0.999f), lds_stack + lane, 64)) {
        r.W = 0;
        reservoir_input_tex.st(x, y, r.as_raw());
    }
}

// restir_resolve.hlsl: rtdgi_resample.hip (k_restir_resolve)

// temporal_filter.hlsl: rtdgi_resample.hip (k_temporal_filter; approximate by design, it lives with the other instruction-bound screen passes)

// spatial_filter.hlsl: rtdgi_resample.hip (k_spatial_filter)

// ================================================================== host side
#ifndef KJ_POOL_WAVES
#define KJ_POOL_WAVES 4
#endif
struct KjRtdgi {
    KjDevice* dev = nullptr;
    uint32_t spatial_reuse_pass_count = 2;      // rtdgi.rs:43-44
    bool use_raytraced_reservoir_visibility = false;
    int W = 0, H = 0, hw = 0, hh = 0;
    std::map<std::string, kj::DevBuf> surf;
    bool flip[8] = {false, false, false, false, false, false, false, false};
    bool temporal2_flip = false;
    void* temporal_output_tex = nullptr;        // ReprojectedRtdgi (rtdgi.rs:48-51)
    void* reprojected_history_tex = nullptr;
    kj::DevBuf ray_counters;                    // KJ_COUNTER_SLOTS x (6 used of KJ_COUNTER_STRIDE) u64, see kj_vec.hpp
    bool profiling = false;                     // per-pass GPU timestamps (gpu-profiler scopes, kajiya-rg/src/graph.rs:941-944)
    bool count_traversal = false;               // instrumented trace kernels
    uint32_t staged_min_rays = 0xffffffffu;     // ray passes run staged (ray streams) from this many ray slots per launch (KJ_RTDGI_STAGED_MIN_RAYS); default: never, see below
    uint32_t stream_waves_per_cu = 24;          // persistent waves per CU of a ray-stream launch (measured best of 8 / 16 / 24 / 32: scripts/traversal_microbench.py)
    bool fuse_validity_temporal = true;         // `validity integrate` + `restir temporal` as one launch (KJ_RTDGI_FUSE_VT=0: two)
    bool trace_deferred = false, validity_copy_pending = false;      // an open KJ_RTDGI_PASS_TRACE_MAY_DEFER call: the whole trace pass / its last statement is left to KJ_RTDGI_PASS_TRACE_FINISH
    uint32_t fuse_order = 0;                    // k_rtdgi_validate_and_trace's dispatch order (KJ_RTDGI_FUSE_ORDER)
    bool fuse_validate_trace = true;            // a validation frame's `rtdgi validate` + `rtdgi trace` as one launch (k_rtdgi_validate_and_trace; KJ_RTDGI_FUSE_RAYS=0: two)
    bool quad_rays = false;                     // the fused ray kernels with four lanes per pixel (kj_rtdgi_set_ray_pass_form KJ_RTDGI_RAYS_QUAD)
    bool split_rays = false;                    // the ray passes as two launches each: closest-hit + misses | hit shading on compacted records (kj_rtdgi_set_ray_pass_form)
    bool grouped_rays = false;                  // the ray passes' form when not staged: grouped (hit shading regrouped inside a 256-thread workgroup) or fused (KJ_RTDGI_GROUPED=0)
    uint32_t ray_waves_per_simd = 0;            // 0 = whatever fits
    bool pool_rays = false;                     // the ray passes as persistent waves over a pool of pixel jobs (k_rtdgi_rays_pool; kj_rtdgi_set_ray_pass_form KJ_RTDGI_RAYS_POOL)
    uint32_t pool_waves_per_simd = 3, pool_refill_min = 16, pool_shade_a_min = 16, pool_shade_b_min = 16, pool_dynamic_tiles = 0;   // kj_rtdgi_set_pool_tune
    kj::DevBuf pool_tile_counters;              // two u32 (validate, trace), zeroed with the ray counters
    int resample_variant = 2;                   // spatial reuse: 2 = per-tap gathers (fastest measured), 0 / 1 = LDS-staged tiles (KJ_RTDGI_RESAMPLE_VARIANT; rtdgi_resample.hpp)
    static const int NUM_SCOPES = 11;
    hipEvent_t ev[NUM_SCOPES][2] = {};
    bool ev_valid[NUM_SCOPES] = {};
    bool ev_created = false;
    hipError_t err = hipSuccess;

    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
    // PingPongTemporalResource::get_output_and_history (renderers/mod.rs:85-102)
    void pingpong(const char* key, int idx, size_t bytes, hipStream_t s, void*& output, void*& history) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx]) std::swap(a, b);
        output = get(a, bytes, s);
        history = get(b, bytes, s);
        flip[idx] = !flip[idx];
    }
    void resize(int W_, int H_) {
        if (W == W_ && H == H_) return;
        W = W_; H = H_; hw = (W + 1) / 2; hh = (H + 1) / 2;  // ImageDesc::half_res
        surf.clear();
    }
};

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())
#define SCOPE_BEGIN(i) do { if (r->profiling) { KJ_TRY_HIP(hipEventRecord(r->ev[i][0], s)); } } while (0)
#define SCOPE_END(i) do { if (r->profiling) { KJ_TRY_HIP(hipEventRecord(r->ev[i][1], s)); r->ev_valid[i] = true; } } while (0)

extern "C" {

KjStatus kj_rtdgi_create(KjDevice* dev, KjRtdgi** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjRtdgi* r = new KjRtdgi();
    r->dev = dev;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_RESAMPLE_VARIANT")) r->resample_variant = atoi(v);
    if (const char* v = kj_debug_getenv("KJ_RTDGI_STAGED_MIN_RAYS")) r->staged_min_rays = uint32_t(atoll(v));
    if (const char* v = kj_debug_getenv("KJ_RTDGI_GROUPED")) r->grouped_rays = atoi(v) != 0;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_SPLIT")) r->split_rays = atoi(v) != 0;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_QUAD")) r->quad_rays = atoi(v) != 0;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_FUSE_VT")) r->fuse_validity_temporal = atoi(v) != 0;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_FUSE_RAYS")) r->fuse_validate_trace = atoi(v) != 0;
    if (const char* v = kj_debug_getenv("KJ_RTDGI_FUSE_ORDER")) r->fuse_order = uint32_t(std::max(0, atoi(v)));
    if (const char* v = kj_debug_getenv("KJ_RTDGI_WAVES_PER_SIMD")) r->ray_waves_per_simd = uint32_t(std::max(0, atoi(v)));
    if (const char* v = kj_debug_getenv("KJ_RTDGI_POOL")) r->pool_rays = atoi(v) != 0;      // A/B runs of bench.py: the pool form of the ray passes on / off
    if (const char* v = kj_debug_getenv("KJ_RTDGI_POOL_TUNE")) {                             // "waves,refill,shade_a,shade_b,dynamic"
        unsigned w = 0, f = 0, a = 0, b = 0, d = 0;
        if (sscanf(v, "%u,%u,%u,%u,%u", &w, &f, &a, &b, &d) == 5 && w <= KJ_POOL_WAVES && f >= 1 && f <= 64 && a >= 1 && a <= 64 && b >= 1 && b <= 64) {
            r->pool_waves_per_simd = w; r->pool_refill_min = f; r->pool_shade_a_min = a; r->pool_shade_b_min = b; r->pool_dynamic_tiles = d ? 1u : 0u;
        } else fprintf(stderr, "kajiya_amd: KJ_RTDGI_POOL_TUNE=%s ignored (want \"waves,refill,shade_a,shade_b,dynamic\")\n", v);
    }
#ifndef KJ_RAY_PASS_EXPERIMENTS
    if (r->grouped_rays || r->split_rays || r->quad_rays || r->pool_rays || r->staged_min_rays != 0xffffffffu) {      // this build carries the fused form only: say so instead of measuring the same kernel twice
        fprintf(stderr, "kajiya_amd: KJ_RTDGI_GROUPED / _SPLIT / _QUAD / _POOL / _STAGED_MIN_RAYS ask for a form of the ray passes this build does not carry (make EXPERIMENTS=1): the fused form runs\n");
        r->grouped_rays = r->split_rays = r->quad_rays = r->pool_rays = false; r->staged_min_rays = 0xffffffffu;
    }
#endif
    if (r->ray_counters.alloc(KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8) != hipSuccess) { delete r; set_last_error("out of device memory"); return KJ_ERR_OUT_OF_MEMORY; }
    *out = r;
    return KJ_OK;
}
void kj_rtdgi_destroy(KjRtdgi* r) { delete r; }
KjStatus kj_rtdgi_set_options(KjRtdgi* r, uint32_t spatial_reuse_pass_count, uint32_t use_raytraced_reservoir_visibility) {
    KJ_REQUIRE(r, "null argument");
    KJ_REQUIRE(spatial_reuse_pass_count >= 1 && spatial_reuse_pass_count <= 8, "spatial_reuse_pass_count out of range");
    r->use_raytraced_reservoir_visibility = use_raytraced_reservoir_visibility != 0;
    r->spatial_reuse_pass_count = spatial_reuse_pass_count;
    return KJ_OK;
}

KjStatus kj_rtdgi_reproject(KjRtdgi* r, const void* reprojection_map, uint32_t width, uint32_t height, void* stream_) {
    return kj_rtdgi_reproject_rows(r, reprojection_map, width, height, 0u, height, stream_);
}
// rows [row_begin, row_end) only (the screen-tile split: a rank reprojects its own strip; the trace pass reads the reprojected image anywhere on
// screen, so the orchestrator all-gathers the strips afterwards)
KjStatus kj_rtdgi_reproject_rows(KjRtdgi* r, const void* reprojection_map, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, void* stream_) {
    KJ_REQUIRE(r && reprojection_map && width && height, "null argument");
    KJ_REQUIRE(row_begin < row_end && row_end <= height && (row_begin % 8u) == 0u, "rows must be a non-empty range starting on a tile row");
    hipStream_t s = (hipStream_t)stream_;
    r->resize(int(width), int(height));
    const int W = r->W, H = r->H;
    std::string a = "rtdgi.temporal2:0", b = "rtdgi.temporal2:1";
    if (r->temporal2_flip) std::swap(a, b);
    r->temporal_output_tex = r->get(a, size_t(W) * H * 8, s);
    void* history = r->get(b, size_t(W) * H * 8, s);
    r->temporal2_flip = !r->temporal2_flip;
    r->reprojected_history_tex = r->get("reprojected_history_tex", size_t(W) * H * 8, s);
    KJ_TRY_HIP(r->err);
    SCOPE_BEGIN(0);
    hipLaunchKernelGGL(k_fullres_reproject, dim3((W + 7) / 8, (int(row_end - row_begin) + 7) / 8), dim3(64), 0, s, img<uint2>(history, W, H), img<uint2>(reprojection_map, W, H),
                       img<uint2>(r->reprojected_history_tex, W, H), int(row_begin), int(row_end));
    KJ_CHECK_LAUNCH();
    SCOPE_END(0);
    return KJ_OK;
}

KjStatus kj_rtdgi_render(KjRtdgi* r, const KjRtdgiRenderParams* p, KjRtdgiOutput* out, void* stream_) {
    KJ_REQUIRE(r && p, "null argument");
    KJ_REQUIRE(p->scene && p->reprojection_map && p->sky_cube && p->ssao_tex && p->gbuffer_depth.depth && p->gbuffer_depth.gbuffer && p->gbuffer_depth.geometric_normal, "missing input");
    KJ_REQUIRE(int(p->gbuffer_depth.width) == r->W && int(p->gbuffer_depth.height) == r->H && r->reprojected_history_tex, "kj_rtdgi_reproject must run first with the same extent");
    KJ_REQUIRE(r->dev->fc_dev, "kj_frame_begin not called");
    if (!p->scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    hipStream_t s = (hipStream_t)stream_;
    const int W = r->W, H = r->H, hw = r->hw, hh = r->hh;
    const FrameConstants* fc = r->dev->fc_dev;
    uint32_t mask = p->pass_mask;
    // KJ_RTDGI_PASS_TRACE_MAY_DEFER / KJ_RTDGI_PASS_TRACE_FINISH (include/kajiya_amd.h; the screen-tile split: `rtdgi validate`'s output needs a halo exchange before the trace pass'
    // LAST statement reads it at the reprojected pixel, and nothing else of the trace pass reads what `rtdgi validate` writes). A call with VALIDATE | TRACE | TRACE_MAY_DEFER either runs
    // both passes as one launch without that statement (a validation frame in the product's form of the ray passes) or leaves the trace pass out altogether; the TRACE_FINISH call
    // behind the exchange runs what is left: the statement alone (k_rtdgi_validity_reproject) or the whole pass.
    bool trace_without_copy = false, validity_copy_only = false;
    if (mask & KJ_RTDGI_PASS_TRACE_FINISH) {
        KJ_REQUIRE(!(mask & KJ_RTDGI_PASS_TRACE) && (r->trace_deferred || r->validity_copy_pending), "KJ_RTDGI_PASS_TRACE_FINISH completes a call with KJ_RTDGI_PASS_TRACE | KJ_RTDGI_PASS_TRACE_MAY_DEFER (none is open)");
        if (r->trace_deferred) mask |= KJ_RTDGI_PASS_TRACE; else validity_copy_only = true;
    }
    if ((mask & KJ_RTDGI_PASS_TRACE_MAY_DEFER) && (mask & KJ_RTDGI_PASS_TRACE) && !(mask & KJ_RTDGI_PASS_TRACE_FINISH)) {
        KJ_REQUIRE(!r->trace_deferred && !r->validity_copy_pending, "the previous KJ_RTDGI_PASS_TRACE_MAY_DEFER call was never completed (KJ_RTDGI_PASS_TRACE_FINISH)");
        bool product_form = true;
#ifdef KJ_RAY_PASS_EXPERIMENTS
        product_form = !(r->quad_rays || r->split_rays || r->grouped_rays || r->pool_rays || r->staged_min_rays != 0xffffffffu);
#endif
        if (product_form && r->fuse_validate_trace && is_rtdgi_validation_frame(r->dev->fc_host.frame_index) && (mask & KJ_RTDGI_PASS_VALIDATE)) trace_without_copy = true;
        else mask &= ~uint32_t(KJ_RTDGI_PASS_TRACE);
    }
    // rows of this call (screen-tile split): full-res [fr0, fr1), half-res [hr0, hr1); 0,0 = whole image
    int fr0 = 0, fr1 = H;
    if (p->row_end > p->row_begin) {
        KJ_REQUIRE(p->row_begin % 16 == 0 && (p->row_end % 16 == 0 || int(p->row_end) == H) && int(p->row_end) <= H, "row range must be 16-aligned (8x8 half-res tiles)");
        fr0 = int(p->row_begin); fr1 = int(p->row_end);
    }
    if (p->ircache && (mask & (KJ_RTDGI_PASS_VALIDATE | KJ_RTDGI_PASS_TRACE))) KJ_REQUIRE(!p->ircache->pending_irradiance_sum, "ircache sum-up pending (ircache.rs:67 assert)");   // the passes that look the cache up
    KJ_REQUIRE(size_t(scene_view(*p->scene).bvh.stack_entries) * 64 * 4 <= 64 * 1024, "BVH too deep for the LDS traversal stack");
    // every argument check is above: from here on the ping-pong state may change (an early return after this point would leave
    // output and history swapped for the next call)
    if (mask & KJ_RTDGI_PASS_KEEP_TEMPORALS) for (bool& f : r->flip) f = !f;
    if (p->pass_mask & KJ_RTDGI_PASS_TRACE_FINISH) r->trace_deferred = r->validity_copy_pending = false;
    else if ((p->pass_mask & KJ_RTDGI_PASS_TRACE_MAY_DEFER) && (p->pass_mask & KJ_RTDGI_PASS_TRACE)) { r->trace_deferred = !trace_without_copy; r->validity_copy_pending = trace_without_copy; }
    const int hr0 = fr0 / 2, hr1 = fr1 == H ? hh : fr1 / 2;
    const dim3 gh((hw + 7) / 8, (hr1 - hr0 + 7) / 8), gf((W + 7) / 8, (fr1 - fr0 + 7) / 8), blk(64);
    const size_t HB = size_t(hw) * hh, FB = size_t(W) * H;

    const ImgU4 gbuffer = img<uint4>(p->gbuffer_depth.gbuffer, W, H);
    const ImgF32 depth = img<float>(p->gbuffer_depth.depth, W, H);
    const ImgU32 geometric_normal = img<uint32_t>(p->gbuffer_depth.geometric_normal, W, H);
    const ImgR8 ssao = img<uint8_t>(p->ssao_tex, W, H);
    const ImgU2 reprojection = img<uint2>(p->reprojection_map, W, H);

    void* half_ssao = r->get("half_ssao_tex", HB, s);
    void* half_view_normal = r->get("half_view_normal_tex", HB * 4, s);
    void* half_depth = r->get("half_depth_tex", HB * 4, s);
    void* half_gbuf = r->get("half_gbuf", HB * 8, s);
    void *hit_normal_out, *hit_normal_hist; r->pingpong("rtdgi.hit_normal", 0, HB * 8, s, hit_normal_out, hit_normal_hist);
    void *candidate_out, *candidate_hist;   r->pingpong("rtdgi.candidate", 1, HB * 8, s, candidate_out, candidate_hist);
    void* candidate_radiance = r->get("candidate_radiance_tex", HB * 8, s);
    void* candidate_normal = r->get("candidate_normal_tex", HB * 4, s);
    void* candidate_hit = r->get("candidate_hit_tex", HB * 8, s);
    void* temporal_reservoir_packed = r->get("temporal_reservoir_packed_tex", HB * 16, s);
    void *invalidity_out, *invalidity_hist; r->pingpong("rtdgi.invalidity", 2, HB * 4, s, invalidity_out, invalidity_hist);
    void *radiance_out, *radiance_hist;     r->pingpong("rtdgi.radiance", 3, HB * 8, s, radiance_out, radiance_hist);
    void *ray_orig_out, *ray_orig_hist;     r->pingpong("rtdgi.ray_orig", 4, HB * 16, s, ray_orig_out, ray_orig_hist);
    void *ray_out, *ray_hist;               r->pingpong("rtdgi.ray", 5, HB * 8, s, ray_out, ray_hist);
    void* validity_pre = r->get("rt_history_validity_pre_input_tex", HB, s);
    void *reservoir_out, *reservoir_hist;   r->pingpong("rtdgi.reservoir", 6, HB * 8, s, reservoir_out, reservoir_hist);
    void* validity_in = r->get("rt_history_validity_input_tex", HB, s);
    void* reservoir_tex0 = r->get("reservoir_output_tex0", HB * 8, s);
    void* reservoir_tex1 = r->get("reservoir_output_tex1", HB * 8, s);
    void* irradiance = r->get("irradiance_output_tex", FB * 8, s);
    void *variance_out, *variance_hist;     r->pingpong("rtdgi.temporal2_var", 7, FB * 4, s, variance_out, variance_hist);
    void* temporal_filtered = r->get("temporal_filtered_tex", FB * 8, s);
    void* spatial_filtered = r->get("spatial_filtered_tex", FB * 8, s);
    KJ_TRY_HIP(r->err);
    if (!(p->pass_mask & KJ_RTDGI_PASS_KEEP_TEMPORALS)) KJ_TRY_HIP(hipMemsetAsync(r->ray_counters.p, 0, KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8, s));  // per frame; pass-by-pass calls accumulate

    TraceCtx tc;
    tc.fc = fc;
    tc.sc = scene_view(*p->scene);
    tc.depth = depth;
    tc.reprojected_gi = img<uint2>(r->reprojected_history_tex, W, H);
    tc.sky_cube = (const uint2*)p->sky_cube;
    tc.sky_cube_width = int(p->sky_cube_width);
    tc.blue_noise = (const uint32_t*)r->dev->blue_noise.p;
    tc.brdf_fg_lut = (const uint2*)r->dev->brdf_fg_lut.p;
    tc.sun_color = (const float4*)r->dev->sun_color.p + r->dev->fc_slot;
    tc.has_ircache = p->ircache != nullptr;
    if (p->ircache) tc.irc = p->ircache->view(); else memset(&tc.irc, 0, sizeof(tc.irc));
    tc.ray_counters = (unsigned long long*)r->ray_counters.p;
    tc.request_stride = uint32_t(hw); tc.request_slot_base = 0; tc.request_key_base = 1u << 28;     // validate pass; the trace pass re-bases below
    if (p->ircache && p->ircache->deferred) KJ_REQUIRE(p->ircache->req_half_pixels == uint32_t(hw) * uint32_t(hh), "kj_ircache_begin_requests must be called with this frame's half-res extent");
    size_t trace_lds = size_t(tc.sc.bvh.stack_entries) * 64 * 4;
    // experiment knob (KJ_RTDGI_WAVES_PER_SIMD=n): cap the fused ray kernels' occupancy through their LDS request, so that tiles are handed
    // out as waves retire instead of all at once (160 KB of LDS per CU, four SIMDs)
    if (r->ray_waves_per_simd) trace_lds = std::max(trace_lds, (size_t(160 * 1024) / (4u * r->ray_waves_per_simd)) & ~size_t(255));

    if (mask & (KJ_RTDGI_PASS_EXTRACT_HALF | KJ_RTDGI_PASS_EXTRACT_HALF_SSAO_ONLY)) {
        // (a frame that splits the extract in two -- everything but the SSAO early, the SSAO byte behind `restir temporal` -- times the first, the bulk, as `extract half`;
        // the SSAO-only launch, ~2 us, is not timed: both in one scope would report the last one only)
        const bool timed_extract = !(mask & KJ_RTDGI_PASS_EXTRACT_HALF_SSAO_ONLY);
        if (timed_extract) SCOPE_BEGIN(1);
        hipLaunchKernelGGL((mask & KJ_RTDGI_PASS_EXTRACT_HALF_SSAO_ONLY) ? k_extract_half<2> : (mask & KJ_RTDGI_PASS_EXTRACT_HALF_NO_SSAO) ? k_extract_half<1> : k_extract_half<0>, gh, blk, 0, s, fc, gbuffer, depth, ssao, img<uint32_t>(half_view_normal, hw, hh), img<float>(half_depth, hw, hh), img<int8_t>(half_ssao, hw, hh), img<uint2>(half_gbuf, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        if (timed_extract) SCOPE_END(1);
    }
    // A validation frame's two ray passes as one launch (k_rtdgi_validate_and_trace) when the call runs both: the pass timers then report the launch (and the small
    // launch that completes rt_history_validity_input_tex behind it) as `rtdgi validate`; `rtdgi trace` reads 0 for that frame.
    const bool rays_in_one_launch = r->fuse_validate_trace && is_rtdgi_validation_frame(r->dev->fc_host.frame_index) && (mask & KJ_RTDGI_PASS_VALIDATE) && (mask & KJ_RTDGI_PASS_TRACE);
    auto launch_validate_and_trace = [&]() -> KjStatus {
        ValidateTraceArgs a;
        a.half_view_normal_tex = img<uint32_t>(half_view_normal, hw, hh);
        a.reservoir_tex = img<uint2>(reservoir_hist, hw, hh); a.reservoir_ray_history_tex = img<uint2>(ray_hist, hw, hh); a.irradiance_history_tex = img<uint2>(radiance_hist, hw, hh);
        a.ray_orig_history_tex = img<float4>(ray_orig_hist, hw, hh); a.validity_pre_tex = img<uint8_t>(validity_pre, hw, hh);
        a.reprojection_tex = reprojection; a.candidate_irradiance_out_tex = img<uint2>(candidate_radiance, hw, hh); a.candidate_normal_out_tex = img<uint32_t>(candidate_normal, hw, hh);
        a.candidate_hit_out_tex = img<uint2>(candidate_hit, hw, hh); a.validity_in_tex = img<uint8_t>(validity_in, hw, hh);
        a.row0 = hr0; a.row1 = hr1; a.tiles_y = gh.y; a.order = r->fuse_order;
        a.trace_request_slot_base = uint32_t(hw) * uint32_t(hh); a.trace_request_key_base = 2u << 28;      // (tc holds the validate pass' bases)
        SCOPE_BEGIN(2);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_and_trace<true> : k_rtdgi_validate_and_trace<false>, dim3(gh.x, gh.y * 2u), blk, trace_lds, s, tc, a);
        KJ_CHECK_LAUNCH();
        if (!trace_without_copy) {
            hipLaunchKernelGGL(k_rtdgi_validity_reproject, gh, blk, 0, s, fc, depth, reprojection, img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
            KJ_CHECK_LAUNCH();
        }
        SCOPE_END(2);
        r->ev_valid[3] = false;
        return KJ_OK;
    };
    if (validity_copy_only) {      // KJ_RTDGI_PASS_TRACE_FINISH behind a one-launch call: the trace pass' last statement
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL(k_rtdgi_validity_reproject, gh, blk, 0, s, fc, depth, reprojection, img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
#ifdef KJ_RAY_PASS_EXPERIMENTS
    // the pool form of the ray passes (k_rtdgi_rays_pool): persistent waves over the launch's tiles
    PoolArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.half_view_normal_tex = img<uint32_t>(half_view_normal, hw, hh);
    pa.reprojection_tex = reprojection; pa.candidate_irradiance_out_tex = img<uint2>(candidate_radiance, hw, hh); pa.candidate_normal_out_tex = img<uint32_t>(candidate_normal, hw, hh);
    pa.candidate_hit_out_tex = img<uint2>(candidate_hit, hw, hh); pa.invalidity_in_tex = img<uint8_t>(validity_pre, hw, hh);
    pa.reservoir_tex = img<uint2>(reservoir_hist, hw, hh); pa.reservoir_ray_history_tex = img<uint2>(ray_hist, hw, hh); pa.irradiance_history_tex = img<uint2>(radiance_hist, hw, hh);
    pa.ray_orig_history_tex = img<float4>(ray_orig_hist, hw, hh);
    pa.row0 = hr0; pa.row1 = hr1; pa.tiles_x = gh.x; pa.tiles_y = gh.y;
    pa.refill_min = std::max(1u, std::min(64u, r->pool_refill_min)); pa.shade_a_min = std::max(1u, std::min(64u, r->pool_shade_a_min)); pa.shade_b_min = std::max(1u, std::min(64u, r->pool_shade_b_min));
    const bool pool = r->pool_rays;
    const size_t pool_lds = pool_lds_bytes(tc.sc.bvh.stack_entries);
    // waves_per_simd = 0: one wave per tile (no persistence: the pool only interleaves a tile's own closest-hit and shadow walks)
    const uint32_t pool_grid = r->pool_waves_per_simd == 0u ? uint32_t(gh.x * gh.y) : std::max(1u, std::min(uint32_t(gh.x * gh.y), uint32_t(r->dev->num_cus) * 4u * std::min(uint32_t(KJ_POOL_WAVES), r->pool_waves_per_simd)));
    if (pool && r->pool_dynamic_tiles) {
        if (!r->pool_tile_counters.p) KJ_TRY_HIP(r->pool_tile_counters.alloc(8));
        KJ_TRY_HIP(hipMemsetAsync(r->pool_tile_counters.p, 0, 8, s));
    }
    auto launch_pool = [&](bool validate) {
        PoolArgs q = pa;
        if (validate) q.invalidity_out_tex = img<uint8_t>(validity_pre, hw, hh); else q.invalidity_out_tex = img<uint8_t>(validity_in, hw, hh);
        q.tile_counter = r->pool_dynamic_tiles ? (uint32_t*)r->pool_tile_counters.p + (validate ? 0 : 1) : nullptr;
        if (validate) hipLaunchKernelGGL((r->count_traversal ? k_rtdgi_rays_pool<true, true> : k_rtdgi_rays_pool<true, false>), dim3(pool_grid), blk, pool_lds, s, tc, q);
        else hipLaunchKernelGGL((r->count_traversal ? k_rtdgi_rays_pool<false, true> : k_rtdgi_rays_pool<false, false>), dim3(pool_grid), blk, pool_lds, s, tc, q);
    };
    // the ray passes' stage buffers (dense, one slot per lane of every 8x8 tile of the launch)
    const uint32_t stage_rays = gh.x * gh.y * 64u;
    const size_t stage_full = size_t((hw + 7) / 8) * ((hh + 7) / 8) * 64;
    RayStage st{};
    if (stage_rays >= r->staged_min_rays) {
    st.rays_a = (float4*)r->get("stage.rays_a", stage_full * 32, s);
    st.hits_a = (float4*)r->get("stage.hits_a", stage_full * 16, s);
    st.rays_b = (float4*)r->get("stage.rays_b", stage_full * 32, s);
    st.occl_b = (uint32_t*)r->get("stage.occl_b", stage_full * 4, s);
    st.state = (float4*)r->get("stage.state", stage_full * 32, s);
    }
    KJ_TRY_HIP(r->err);
    uint32_t stream_grid;
    const StreamTune stream_tune = stream_tune_for(stage_rays, r->dev->num_cus * r->stream_waves_per_cu, &stream_grid);
    auto trace_streams = [&](bool closest) {
        if (closest) hipLaunchKernelGGL((r->count_traversal ? k_rtdgi_ray_stream<false, true> : k_rtdgi_ray_stream<false, false>), dim3(stream_grid), blk, trace_lds, s, tc.sc.bvh,
                                        (const float4*)st.rays_a, st.hits_a, (uint32_t*)nullptr, stage_rays, tc.ray_counters, stream_tune);
        else hipLaunchKernelGGL((r->count_traversal ? k_rtdgi_ray_stream<true, true> : k_rtdgi_ray_stream<true, false>), dim3(stream_grid), blk, trace_lds, s, tc.sc.bvh,
                                (const float4*)st.rays_b, (float4*)nullptr, st.occl_b, stage_rays, tc.ray_counters, stream_tune);
    };
    const bool staged = stage_rays >= r->staged_min_rays;
    // the fused form (one wave per tile does everything) or the grouped one (hit shading regrouped inside a 256-thread workgroup); same
    // outputs bit for bit (kj_rtdgi_set_ray_pass_form)
    const bool split = r->split_rays && !staged;
    const bool grouped = r->grouped_rays && !split;
    SplitStage sst{};
    if (split) {
        sst.tiles = gh.x * gh.y;
        sst.records = (SplitRecord*)r->get("split.records", size_t(((hw + 7) / 8) * ((hh + 7) / 8)) * 64 * sizeof(SplitRecord), s);
        sst.counts = (uint32_t*)r->get("split.counts", size_t(((hw + 7) / 8) * ((hh + 7) / 8)) * 4, s);
        KJ_TRY_HIP(r->err);
    }
    const dim3 gshade((gh.x * gh.y + KJ_SPLIT_TILES - 1) / KJ_SPLIT_TILES);
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && split) {
        SCOPE_BEGIN(2);
        if (is_rtdgi_validation_frame(r->dev->fc_host.frame_index)) {
            hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_closest<true> : k_rtdgi_validate_closest<false>, gh, blk, trace_lds, s, tc, sst, img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                               img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
            hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_shade<true> : k_rtdgi_validate_shade<false>, gshade, blk, trace_lds, s, tc, sst, img<uint2>(reservoir_hist, hw, hh), img<uint2>(radiance_hist, hw, hh),
                               img<uint8_t>(validity_pre, hw, hh));
        } else   // two frames of three the pass only writes the invalidity image: the fused kernel does just that
            hipLaunchKernelGGL(k_rtdgi_validate_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                               img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(2);
    }
    const dim3 gg((hw + 15) / 16, (uint32_t(hr1 - hr0) + 15) / 16), gblk(KJ_GROUP_THREADS);
    const size_t grouped_lds = grouped_lds_bytes(tc.sc.bvh.stack_entries);
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && !staged && grouped) {
        SCOPE_BEGIN(2);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_grouped<true> : k_rtdgi_validate_grouped<false>, gg, gblk, grouped_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                           img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(2);
    }
    // QUAD form of the fused kernels: four lanes per pixel (bvh_trace_quad), a wave covers 8 x 2 pixels, four times the waves
    const bool quad = r->quad_rays && !staged && !grouped && !split && !r->count_traversal;
    const dim3 ghq(gh.x, (uint32_t(hr1 - hr0) + 1) / 2);
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && quad) {
        SCOPE_BEGIN(2);
        if (is_rtdgi_validation_frame(r->dev->fc_host.frame_index))
            hipLaunchKernelGGL((k_rtdgi_validate_fused<false, true>), ghq, blk, quad_stack_bytes(), s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                               img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        else   // two frames of three the pass only writes the invalidity image
            hipLaunchKernelGGL(k_rtdgi_validate_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                               img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(2);
    }
    const bool fused_default = !staged && !grouped && !split && !quad && !pool;      // the product's form of the ray passes
    if (fused_default && rays_in_one_launch) { const KjStatus st_ = launch_validate_and_trace(); if (st_ != KJ_OK) return st_; }
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && !staged && !grouped && !split && !quad && !(fused_default && rays_in_one_launch)) {
        SCOPE_BEGIN(2);
        if (pool && is_rtdgi_validation_frame(r->dev->fc_host.frame_index)) launch_pool(true);
        else hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_fused<true> : k_rtdgi_validate_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                           img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(2);
    }
    tc.request_slot_base = uint32_t(hw) * uint32_t(hh); tc.request_key_base = 2u << 28;
    if ((mask & KJ_RTDGI_PASS_TRACE) && !staged && grouped) {
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_trace_grouped<true> : k_rtdgi_trace_grouped<false>, gg, gblk, grouped_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
    if ((mask & KJ_RTDGI_PASS_TRACE) && split) {
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_trace_closest<true> : k_rtdgi_trace_closest<false>, gh, blk, trace_lds, s, tc, sst, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_trace_shade<true> : k_rtdgi_trace_shade<false>, gshade, blk, trace_lds, s, tc, sst, img<uint2>(candidate_radiance, hw, hh), img<uint32_t>(candidate_normal, hw, hh),
                           img<uint2>(candidate_hit, hw, hh));
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
    if ((mask & KJ_RTDGI_PASS_TRACE) && quad) {
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL((k_rtdgi_trace_fused<false, true>), ghq, blk, quad_stack_bytes(), s, tc, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
    if ((mask & KJ_RTDGI_PASS_TRACE) && !staged && !grouped && !split && !quad && !(fused_default && rays_in_one_launch)) {
        SCOPE_BEGIN(3);
        if (pool) launch_pool(false);
        else hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_trace_fused<true> : k_rtdgi_trace_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && staged) {
        SCOPE_BEGIN(2);
        hipLaunchKernelGGL(k_rtdgi_validate_raygen, gh, blk, 0, s, tc, st, img<uint2>(ray_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        if (is_rtdgi_validation_frame(r->dev->fc_host.frame_index)) {   // on the two tracing frames of three the pass has no rays: only the invalidity image is written
            trace_streams(true);
            TraceCtx vc = tc;        // the validate pass' own request slots / keys (tc was re-based for the trace pass above)
            vc.request_slot_base = 0; vc.request_key_base = 1u << 28;
            hipLaunchKernelGGL(k_rtdgi_shade<true>, gh, blk, trace_lds, s, vc, st, img<uint32_t>(half_view_normal, hw, hh), img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), hw, hh, hr0, hr1);
            trace_streams(false);
            hipLaunchKernelGGL(k_rtdgi_validate_finish, gh, blk, 0, s, st, img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh), img<uint2>(radiance_hist, hw, hh),
                               img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
            KJ_CHECK_LAUNCH();
        }
        SCOPE_END(2);
    }
    if ((mask & KJ_RTDGI_PASS_TRACE) && staged) {
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL(k_rtdgi_trace_raygen, gh, blk, 0, s, tc, st, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        trace_streams(true);
        hipLaunchKernelGGL(k_rtdgi_shade<false>, gh, blk, trace_lds, s, tc, st, img<uint32_t>(half_view_normal, hw, hh), img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), hw, hh, hr0, hr1);
        trace_streams(false);
        hipLaunchKernelGGL(k_rtdgi_trace_finish, gh, blk, 0, s, fc, st, img<uint2>(candidate_radiance, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
#else
    // The product build carries the fused form of the two ray passes only (one wave per 8x8 tile: ray generation, both traversals, hit shading);
    // the other forms live behind KJ_RAY_PASS_EXPERIMENTS (rtdgi_ray_experiments.inc)
    if (rays_in_one_launch) { const KjStatus st_ = launch_validate_and_trace(); if (st_ != KJ_OK) return st_; }
    if ((mask & KJ_RTDGI_PASS_VALIDATE) && !rays_in_one_launch) {
        SCOPE_BEGIN(2);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_validate_fused<true> : k_rtdgi_validate_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), img<uint2>(reservoir_hist, hw, hh), img<uint2>(ray_hist, hw, hh),
                           img<uint2>(radiance_hist, hw, hh), img<float4>(ray_orig_hist, hw, hh), img<uint8_t>(validity_pre, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(2);
    }
    tc.request_slot_base = uint32_t(hw) * uint32_t(hh); tc.request_key_base = 2u << 28;
    if ((mask & KJ_RTDGI_PASS_TRACE) && !rays_in_one_launch) {
        SCOPE_BEGIN(3);
        hipLaunchKernelGGL(r->count_traversal ? k_rtdgi_trace_fused<true> : k_rtdgi_trace_fused<false>, gh, blk, trace_lds, s, tc, img<uint32_t>(half_view_normal, hw, hh), reprojection, img<uint2>(candidate_radiance, hw, hh),
                           img<uint32_t>(candidate_normal, hw, hh), img<uint2>(candidate_hit, hw, hh), img<uint8_t>(validity_pre, hw, hh), img<uint8_t>(validity_in, hw, hh), hr0, hr1);
        KJ_CHECK_LAUNCH();
        SCOPE_END(3);
    }
#endif
    const ValidityIntegrateArgs via{fc, img<uint8_t>(validity_in, hw, hh), img<uint32_t>(invalidity_hist, hw, hh), reprojection, img<float>(half_depth, hw, hh),
                                    img<uint32_t>(invalidity_out, hw, hh), W, H, hr0, hr1};
    const bool fuse_validity_temporal = r->fuse_validity_temporal && (mask & KJ_RTDGI_PASS_VALIDITY_INTEGRATE) && (mask & KJ_RTDGI_PASS_RESTIR_TEMPORAL);
    if (fuse_validity_temporal) r->ev_valid[4] = false;
    if ((mask & KJ_RTDGI_PASS_VALIDITY_INTEGRATE) && !fuse_validity_temporal) {
        SCOPE_BEGIN(4);
        hipLaunchKernelGGL(k_validity_integrate, gh, blk, 0, s, via);
        KJ_CHECK_LAUNCH();
        SCOPE_END(4);
    }
    if (mask & KJ_RTDGI_PASS_RESTIR_TEMPORAL) {
        RestirTemporalArgs a;
        a.fc = fc; a.depth_tex = depth;
        a.half_view_normal_tex = img<uint32_t>(half_view_normal, hw, hh);
        a.candidate_radiance_tex = img<uint2>(candidate_radiance, hw, hh);
        a.candidate_normal_tex = img<uint32_t>(candidate_normal, hw, hh);
        a.candidate_hit_tex = img<uint2>(candidate_hit, hw, hh);
        a.radiance_history_tex = img<uint2>(radiance_hist, hw, hh);
        a.ray_orig_history_tex = img<float4>(ray_orig_hist, hw, hh);
        a.ray_history_tex = img<uint2>(ray_hist, hw, hh);
        a.reservoir_history_tex = img<uint2>(reservoir_hist, hw, hh);
        a.reprojection_tex = reprojection;
        a.hit_normal_history_tex = img<uint2>(hit_normal_hist, hw, hh);
        a.candidate_history_tex = img<uint2>(candidate_hist, hw, hh);
        a.rt_invalidity_tex = img<uint32_t>(invalidity_out, hw, hh);
        a.radiance_out_tex = img<uint2>(radiance_out, hw, hh);
        a.ray_orig_output_tex = img<float4>(ray_orig_out, hw, hh);
        a.ray_output_tex = img<uint2>(ray_out, hw, hh);
        a.hit_normal_output_tex = img<uint2>(hit_normal_out, hw, hh);
        a.reservoir_out_tex = img<uint2>(reservoir_out, hw, hh);
        a.candidate_out_tex = img<uint2>(candidate_out, hw, hh);
        a.temporal_reservoir_packed_tex = img<uint4>(temporal_reservoir_packed, hw, hh);
        SCOPE_BEGIN(5);
        a.row0 = hr0; a.row1 = hr1;
        if (fuse_validity_temporal) hipLaunchKernelGGL(k_validity_integrate_restir_temporal, gh, blk, 0, s, via, a);      // timed as scope 5; scope 4 reads 0
        else hipLaunchKernelGGL(k_restir_temporal, gh, blk, 0, s, a);
        KJ_CHECK_LAUNCH();
        SCOPE_END(5);
    }
    void* reservoir_input = reservoir_out;
    for (uint32_t i = 0; i < r->spatial_reuse_pass_count; ++i) {
        const uint32_t perform_occlusion_raymarch = (i + 1 == r->spatial_reuse_pass_count) ? 1u : 0u;
        if ((mask & KJ_RTDGI_PASS_RESTIR_SPATIAL) && (p->spatial_pass_select == 0 || p->spatial_pass_select == i + 1)) {
            SCOPE_BEGIN((6 + (i ? 1 : 0)));
            SpatialLaunch L;
            L.fc = fc; L.reservoir_input = reservoir_input; L.half_gbuf = half_gbuf; L.half_depth = half_depth; L.temporal_reservoir_packed = temporal_reservoir_packed;
            L.reservoir_output = reservoir_tex0; L.W = W; L.H = H; L.hw = hw; L.hh = hh;
            L.pass_idx = i; L.perform_occlusion_raymarch = perform_occlusion_raymarch; L.occlusion_raymarch_importance_only = r->use_raytraced_reservoir_visibility ? 1u : 0u;
            L.row0 = hr0; L.row1 = hr1; L.variant = r->resample_variant;
            KJ_TRY_HIP(launch_restir_spatial(L, s));
            SCOPE_END((6 + (i ? 1 : 0)));
        }
        std::swap(reservoir_tex0, reservoir_tex1);
        reservoir_input = reservoir_tex1;
    }
    if (r->use_raytraced_reservoir_visibility && (mask & KJ_RTDGI_PASS_RESTIR_SPATIAL) && (p->spatial_pass_select == 0 || p->spatial_pass_select == r->spatial_reuse_pass_count)) {
        // "restir check": one visibility ray per reservoir towards its selected sample; occluded reservoirs get W = 0
        hipLaunchKernelGGL(k_restir_check, gh, blk, trace_lds, s, fc, tc.sc, img<float>(half_depth, hw, hh), img<uint4>(temporal_reservoir_packed, hw, hh),
                           img<uint2>(reservoir_input, hw, hh), tc.ray_counters, W, H, hr0, hr1);
        KJ_CHECK_LAUNCH();
    }
    if (mask & KJ_RTDGI_PASS_RESTIR_RESOLVE) {
        ResolveLaunch L;
        L.fc = fc; L.radiance = radiance_out; L.reservoir_input = reservoir_input; L.gbuffer = p->gbuffer_depth.gbuffer; L.depth = p->gbuffer_depth.depth;
        L.half_gbuf = half_gbuf; L.ssao = p->ssao_tex; L.candidate_radiance = candidate_radiance; L.candidate_hit = candidate_hit;
        L.temporal_reservoir_packed = temporal_reservoir_packed; L.blue_noise = r->dev->blue_noise.p; L.irradiance_output = irradiance;
        L.W = W; L.H = H; L.hw = hw; L.hh = hh; L.row0 = fr0; L.row1 = fr1;
        SCOPE_BEGIN(8);
        KJ_TRY_HIP(launch_restir_resolve(L, s));
        SCOPE_END(8);
    }
    if (mask & KJ_RTDGI_PASS_TEMPORAL_FILTER) {
        SCOPE_BEGIN(9);
        KJ_TRY_HIP(launch_temporal_filter(fc, irradiance, r->reprojected_history_tex, variance_hist, reprojection.p, invalidity_out, temporal_filtered, r->temporal_output_tex, variance_out,
                                          W, H, fr0, fr1, s));
        SCOPE_END(9);
    }
    if (mask & KJ_RTDGI_PASS_SPATIAL_FILTER) {
        SCOPE_BEGIN(10);
        KJ_TRY_HIP(launch_spatial_filter(fc, temporal_filtered, p->gbuffer_depth.depth, p->ssao_tex, p->gbuffer_depth.geometric_normal, spatial_filtered, W, H, fr0, fr1, s));
        SCOPE_END(10);
    }
    if (out) {
        out->screen_irradiance_tex = spatial_filtered;
        out->candidate_radiance_tex = candidate_radiance;
        out->candidate_normal_tex = candidate_normal;
        out->candidate_hit_tex = candidate_hit;
    }
    return KJ_OK;
}

KjStatus kj_rtdgi_surface(KjRtdgi* r, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(r && name && out_dev_ptr && out_bytes, "null argument");
    if (strcmp(name, "ray_counters") == 0) { *out_dev_ptr = r->ray_counters.p; *out_bytes = r->ray_counters.bytes; return KJ_OK; }
    auto it = r->surf.find(name);
    if (it == r->surf.end()) { set_last_error("no rtdgi surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}
KjStatus kj_rtdgi_ray_counts(KjRtdgi* r, uint64_t* out_closest, uint64_t* out_any) {
    KJ_REQUIRE(r && out_closest && out_any, "null argument");
    uint64_t v[2];
    unsigned long long all_[KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE];
    KJ_TRY_HIP(hipMemcpy(all_, r->ray_counters.p, sizeof(all_), hipMemcpyDeviceToHost));
    v[0] = v[1] = 0;
    for (uint32_t sl = 0; sl < KJ_COUNTER_SLOTS; ++sl) { v[0] += all_[sl * KJ_COUNTER_STRIDE]; v[1] += all_[sl * KJ_COUNTER_STRIDE + 1]; }
    *out_closest = v[0]; *out_any = v[1];
    return KJ_OK;
}
KjStatus kj_rtdgi_set_profiling(KjRtdgi* r, uint32_t enable_pass_timers, uint32_t count_traversal) {
    KJ_REQUIRE(r, "null argument");
    if (enable_pass_timers && !r->ev_created) {
        for (int i = 0; i < KjRtdgi::NUM_SCOPES; ++i)
            for (int k = 0; k < 2; ++k) KJ_TRY_HIP(hipEventCreate(&r->ev[i][k]));
        r->ev_created = true;
    }
    r->profiling = enable_pass_timers != 0;
    r->count_traversal = count_traversal != 0;
    return KJ_OK;
}
KjStatus kj_rtdgi_set_ray_pass_form(KjRtdgi* r, uint32_t form) {
    KJ_REQUIRE(r && form <= KJ_RTDGI_RAYS_POOL, "null argument / unknown form");
#ifndef KJ_RAY_PASS_EXPERIMENTS
    KJ_REQUIRE(form == KJ_RTDGI_RAYS_FUSED, "this build carries the fused form only (the measured-and-rejected others, the pool form included: make EXPERIMENTS=1, -DKJ_RAY_PASS_EXPERIMENTS)");
#endif
    r->pool_rays = form == KJ_RTDGI_RAYS_POOL;
    r->grouped_rays = form == KJ_RTDGI_RAYS_GROUPED;
    r->split_rays = form == KJ_RTDGI_RAYS_SPLIT;
    r->quad_rays = form == KJ_RTDGI_RAYS_QUAD;
    r->staged_min_rays = form == KJ_RTDGI_RAYS_STAGED ? 0u : 0xffffffffu;
    return KJ_OK;
}
KjStatus kj_rtdgi_set_pool_tune(KjRtdgi* r, uint32_t waves_per_simd, uint32_t refill_min, uint32_t shade_a_min, uint32_t shade_b_min, uint32_t dynamic_tiles) {
    KJ_REQUIRE(r, "null argument");
    KJ_REQUIRE(waves_per_simd <= KJ_POOL_WAVES, "waves_per_simd out of range (0 = one wave per tile, else 1 .. the kernel's compiled occupancy)");
    KJ_REQUIRE(refill_min >= 1 && refill_min <= 64 && shade_a_min >= 1 && shade_a_min <= 64 && shade_b_min >= 1 && shade_b_min <= 64, "thresholds are lane counts (1 .. 64)");
    r->pool_waves_per_simd = waves_per_simd; r->pool_refill_min = refill_min; r->pool_shade_a_min = shade_a_min; r->pool_shade_b_min = shade_b_min; r->pool_dynamic_tiles = dynamic_tiles ? 1u : 0u;
    return KJ_OK;
}
KjStatus kj_rtdgi_pass_times_ms(KjRtdgi* r, float* out_ms, uint32_t count) {
    KJ_REQUIRE(r && out_ms && count >= (uint32_t)KjRtdgi::NUM_SCOPES, "need room for KJ_RTDGI_NUM_SCOPES floats");
    for (int i = 0; i < KjRtdgi::NUM_SCOPES; ++i) {
        out_ms[i] = 0.0f;
        if (r->ev_valid[i]) KJ_TRY_HIP(hipEventElapsedTime(&out_ms[i], r->ev[i][0], r->ev[i][1]));
    }
    return KJ_OK;
}
KjStatus kj_rtdgi_traversal_counts(KjRtdgi* r, uint64_t out[6]) {
    KJ_REQUIRE(r && out, "null argument");
    unsigned long long all_[KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE];
    KJ_TRY_HIP(hipMemcpy(all_, r->ray_counters.p, sizeof(all_), hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k) { out[k] = 0; for (uint32_t sl = 0; sl < KJ_COUNTER_SLOTS; ++sl) out[k] += all_[sl * KJ_COUNTER_STRIDE + k]; }
    return KJ_OK;
}

}  // extern "C"
