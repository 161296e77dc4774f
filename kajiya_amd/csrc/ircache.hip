// IrcacheRenderer / IrcacheRenderState for gfx950 (renderers/ircache.rs:92-506, assets/shaders/ircache/*.hlsl,
// prefix_scan/*). Maintenance kernels are grid-stride over entries/cells; the 64 Ki-element prefix scan is
// one 1024-thread workgroup in LDS (the reference launches a 1 Mi-element three-pass scan, prefix_scan.rs:10-38);
// ray kernels use the same software BVH traversal as rtdgi with a grid sized to the chip instead of the
// reference's MAX_ENTRIES*4 launch with early-out (ircache.rs:416-476).
#include "kj_host.hpp"
#include "kj_scene.hpp"
#include "kj_ircache.hpp"
#include "kj_reservoir.hpp"
#include "kj_ircache_host.hpp"
#include <algorithm>

using namespace kj;
namespace kj { SceneView scene_view(const KjScene& s); }

// ------------------------------------------------------------------ maintenance
__global__ void k_irc_clear_pool(uint32_t* __restrict__ pool, uint32_t* __restrict__ life) {  // clear_ircache_pool.hlsl
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < IRC_MAX_ENTRIES) { pool[i] = i; life[i] = IRC_LIFE_RECYCLED; }
}
// scroll_cascades.hlsl:13-69 — one thread per destination cell
__global__ void __launch_bounds__(256) k_irc_scroll_cascades(const FrameConstants* __restrict__ fc, const uint2* __restrict__ src, uint2* __restrict__ dst,
                                                              uint32_t* __restrict__ entry_cell, float4* __restrict__ irradiance, uint32_t* __restrict__ life,
                                                              uint32_t* __restrict__ pool, uint32_t* __restrict__ meta, uint32_t* __restrict__ freed) {
    const uint32_t dst_cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (dst_cell >= IRC_MAX_GRID_CELLS) return;
    const uint32_t x = dst_cell & 31u, y = (dst_cell >> 5) & 31u, z = (dst_cell >> 10) & 31u, cascade = dst_cell >> 15;
    const int32_t* sb = fc->ircache_cascades[cascade].voxels_scrolled_this_frame;
    const uint32_t bx = uint32_t(int(x) - sb[0]), by = uint32_t(int(y) - sb[1]), bz = uint32_t(int(z) - sb[2]);
    if (!(bx < 32u && by < 32u && bz < 32u)) {
        const uint2 m = src[dst_cell];  // deallocate_cell
        if (m.y & IRC_META_OCCUPIED) {
            const uint32_t entry_idx = m.x;
            life[entry_idx] = IRC_LIFE_RECYCLED;
            for (int i = 0; i < 3; ++i) irradiance[entry_idx * 3 + i] = make_float4(0, 0, 0, 0);
            if (freed) freed[entry_idx] = 1u;      // deterministic mode: returned to the pool in entry order by k_irc_push_freed
            else { const uint32_t c = atomicAdd(&meta[IRC_META_ALLOC_COUNT], 0xffffffffu); pool[c - 1u] = entry_idx; }
        }
    }
    const uint32_t sx = uint32_t(int(x) + sb[0]), sy = uint32_t(int(y) + sb[1]), sz = uint32_t(int(z) + sb[2]);
    if (sx < 32u && sy < 32u && sz < 32u) {
        const uint2 cm = src[irc_cell_idx(sx, sy, sz, cascade)];
        dst[dst_cell] = cm;
        if (cm.y & IRC_META_OCCUPIED) entry_cell[cm.x] = dst_cell;
    } else {
        dst[dst_cell] = make_uint2(0, 0);
    }
}
// age_ircache_entries.hlsl:22-94 (+ prepare_age_dispatch_args.hlsl: the reference dispatches ceil(entry_count/64) groups)
__global__ void __launch_bounds__(256) k_irc_age(IrcacheView ic, uint32_t* __restrict__ occupancy, uint32_t* __restrict__ freed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= IRC_MAX_ENTRIES) return;
    const uint32_t total_entry_count = ic.meta[IRC_META_ENTRY_COUNT];
    const uint32_t dispatched = ((total_entry_count + 63u) / 64u) * 64u;
    if (e >= dispatched) { occupancy[e] = 0; return; }
    if (e < total_entry_count) {
        const uint32_t l = ic.life[e];
        if (l != IRC_LIFE_RECYCLED) {
            const uint32_t new_age = l + 1u;
            if (irc_life_valid(new_age)) {
                ic.life[e] = new_age;
                atomicAnd(&ic.grid_meta[ic.entry_cell[e]].y, ~IRC_META_JUST_ALLOCATED);
            } else {
                ic.life[e] = IRC_LIFE_RECYCLED;
                for (int i = 0; i < 3; ++i) ic.irradiance[e * 3 + i] = make_float4(0, 0, 0, 0);
                if (freed) freed[e] = 1u;
                else { const uint32_t c = atomicAdd(&ic.meta[IRC_META_ALLOC_COUNT], 0xffffffffu); ic.pool[c - 1u] = e; }
                atomicAnd(&ic.grid_meta[ic.entry_cell[e]].y, ~(IRC_META_OCCUPIED | IRC_META_JUST_ALLOCATED));
            }
        }
        ic.spatial[e] = ic.reposition_proposal[e];
        ic.reposition_proposal_count[e] = 0;
    } else {
        ic.spatial[e] = make_float4(0, 0, 0, 0);
    }
    occupancy[e] = (e < total_entry_count && irc_life_valid(ic.life[e])) ? 1u : 0u;
}
// inclusive prefix scan over 64 Ki u32 in one workgroup (prefix_scan/*.hlsl semantics)
__global__ void __launch_bounds__(1024) k_irc_scan(uint32_t* __restrict__ data, uint32_t* __restrict__ data2) {
    __shared__ uint32_t partial[1024];
    if (blockIdx.x == 1u) data = data2;      // two independent scans in one launch (the freed-entry flags and the occupancy flags of the deterministic mode's prepare)
    const uint32_t t = threadIdx.x;
    uint4 v[16];
    uint32_t run = 0;
    uint4* p = (uint4*)data + t * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint4 a = p[i];
        a.x += run; a.y += a.x; a.z += a.y; a.w += a.z; run = a.w;
        v[i] = a;
    }
    partial[t] = run;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t add = t >= off ? partial[t - off] : 0u;
        __syncthreads();
        partial[t] += add;
        __syncthreads();
    }
    const uint32_t base = t ? partial[t - 1] : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) { uint4 a = v[i]; a.x += base; a.y += base; a.z += base; a.w += base; p[i] = a; }
}
// deterministic mode (deferred updates): entries freed by the scroll and age passes go back to the pool in ascending entry order
// (`freed_scan` = inclusive scan of the flags) instead of in the order their threads happened to run
__global__ void __launch_bounds__(256) k_irc_push_freed(const uint32_t* __restrict__ freed_scan, uint32_t* __restrict__ pool, uint32_t* __restrict__ meta_alloc_count_after) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= IRC_MAX_ENTRIES) return;
    const uint32_t total = freed_scan[IRC_MAX_ENTRIES - 1u], incl = freed_scan[e], before = e ? freed_scan[e - 1u] : 0u;
    const uint32_t new_count = meta_alloc_count_after[IRC_META_ALLOC_COUNT] - total;
    if (incl != before) pool[new_count + before] = e;
}
__global__ void k_irc_pop_freed_count(const uint32_t* __restrict__ freed_scan, uint32_t* __restrict__ meta) { meta[IRC_META_ALLOC_COUNT] -= freed_scan[IRC_MAX_ENTRIES - 1u]; }
// ircache_compact_entries.hlsl
__global__ void __launch_bounds__(256) k_irc_compact(const uint32_t* __restrict__ meta, const uint32_t* __restrict__ life, const uint32_t* __restrict__ occupancy,
                                                      uint32_t* __restrict__ indirection) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= IRC_MAX_ENTRIES) return;
    if (e < meta[IRC_META_ENTRY_COUNT] && irc_life_valid(life[e])) indirection[occupancy[e]] = e;
}
// prepare_trace_dispatch_args.hlsl
__global__ void k_irc_prepare_trace(uint32_t* __restrict__ meta) { meta[IRC_META_TRACING_ALLOC_COUNT] = meta[IRC_META_ALLOC_COUNT]; }
// reset_entry.hlsl: one wave per entry (64 float4 of aux = one wave-wide store)
__global__ void __launch_bounds__(64) k_irc_reset(IrcacheView ic) {
    const uint32_t alloc_count = ic.meta[IRC_META_TRACING_ALLOC_COUNT];
    for (uint32_t d = blockIdx.x; d < alloc_count; d += gridDim.x) {
        const uint32_t entry_idx = ic.entry_indirection[d];
        const float4 i0 = ic.irradiance[entry_idx * 3];
        if (i0.x == 0.0f && i0.y == 0.0f && i0.z == 0.0f && i0.w == 0.0f) ic.aux[size_t(entry_idx) * IRC_AUX_STRIDE + threadIdx.x] = make_float4(0, 0, 0, 0);
    }
}

// deterministic mode: the half of every live entry's aux block that lookups read (reservoirs + contributions), as it is before a pass
__global__ void __launch_bounds__(256) k_irc_snapshot_aux(const uint32_t* __restrict__ meta, const float4* __restrict__ aux, float4* __restrict__ snap) {
    const uint32_t n = meta[IRC_META_ENTRY_COUNT] * 32u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const size_t o = size_t(i >> 5) * IRC_AUX_STRIDE + (i & 31u);
        snap[o] = aux[o];
    }
}

// ------------------------------------------------------------------ ray passes
struct IrcTraceCtx {
    const FrameConstants* __restrict__ fc;
    SceneView sc;
    IrcacheView ic;
    const uint2* __restrict__ sky_cube; int sky_cube_width;
    const uint2* __restrict__ brdf_fg_lut;
    const float4* __restrict__ sun_color;
    unsigned long long* __restrict__ ray_counters;
    uint32_t request_slot_base;      // first slot of the cache's own passes in IrcacheView::requests (deferred updates)
    uint32_t lanes;                  // work items per wave (<= 64): see kj_ircache_trace_irradiance
    uint32_t part_index, part_count; // this launch takes the entries at positions part_index, part_index + part_count, ... of the tracing list
};
// one count per path: in the QUAD form the four lanes of a path run the same code, lane 0 counts
template <bool QUAD> KJ_D void irc_count_path_rays(unsigned long long* counters, int which) {
#if !defined(__HIP_DEVICE_COMPILE__)      // the tests' CPU stand-in for HIP has no votes in divergent code: one atomic per path
    if (!QUAD || (threadIdx.x & 3u) == 0u) atomicAdd(&counter_slot(counters)[which], 1ull);
    return;
#endif
    const unsigned long long m = __ballot(true) & (QUAD ? 0x1111111111111111ull : ~0ull);
    if (m != 0ull && (__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(&counter_slot(counters)[which], (unsigned long long)__popcll(m));
}
// Work distribution of the three ray kernels. QUAD (the default): 16 paths per wave, four lanes each; else `lanes` paths per wave, one lane each.
#define IRC_PATH_LOOP(entries_, per_entry_)                                                                                      \
    const uint32_t per_wave_ = QUAD ? 16u : c.lanes;                                                                                \
    const uint32_t slot_ = QUAD ? (threadIdx.x >> 2) : threadIdx.x;                                                                 \
    const bool lead = !QUAD || (threadIdx.x & 3u) == 0u;                                                                            \
    uint32_t* const stack = lds_stack + slot_;                                                                                      \
    const uint32_t stride = QUAD ? 16u : 64u;                                                                                       \
    if (slot_ >= per_wave_) return;                                                                                                 \
    const uint32_t own_entries_ = ((entries_) + c.part_count - 1u - c.part_index) / c.part_count;                                   \
    for (uint32_t local_ = block_ * per_wave_ + slot_, d = 0; local_ < own_entries_ * (per_entry_) &&                               \
         ((d = ((local_ / (per_entry_)) * c.part_count + c.part_index) * (per_entry_) + local_ % (per_entry_)), true); local_ += nblocks_ * per_wave_)
// trace_accessibility.rgen.hlsl:21-66
// (the three passes are device functions over a block range so that one launch can carry all of them: k_irc_ray_passes below)
template <bool QUAD>
KJ_D void irc_trace_accessibility_blocks(const IrcTraceCtx& c, uint32_t* lds_stack, uint32_t block_, uint32_t nblocks_) {
    const IrcacheView& ic = c.ic;
    IRC_PATH_LOOP(ic.meta[IRC_META_TRACING_ALLOC_COUNT], IRC_OCTA_DIMS2) {
        const uint32_t entry_idx = ic.entry_indirection[d / IRC_OCTA_DIMS2];
        const uint32_t octa_idx = d % IRC_OCTA_DIMS2;
        if (!irc_life_valid(ic.life[entry_idx])) continue;
        const IrcVertex entry = irc_unpack_vertex(ic.spatial[entry_idx]);
        const size_t output_idx = size_t(entry_idx) * IRC_AUX_STRIDE + octa_idx;
        const float4 r0 = ic.aux[output_idx];
        Reservoir1spp r = Reservoir1spp::from_raw(make_uint2(asuint(r0.x), asuint(r0.y)));
        const IrcVertex prev_entry = irc_unpack_vertex(ic.aux[output_idx + IRC_OCTA_DIMS2 * 2]);
        irc_count_path_rays<QUAD>(c.ray_counters, 1);
        const bool blocked = QUAD ? rt_is_shadowed_quad(c.sc, true, entry.position, prev_entry.position - entry.position, 0.001f, 0.999f, stack, stride)
                                  : rt_is_shadowed(c.sc, entry.position, prev_entry.position - entry.position, 0.001f, 0.999f, stack, stride);
        if (blocked && lead) {
            r.M *= 0.8f;
            const uint2 raw = r.as_raw();
            float2* dst = (float2*)&ic.aux[output_idx];
            *dst = make_float2(asfloat(raw.x), asfloat(raw.y));
        }
    }
}
template <bool QUAD>
__global__ void __launch_bounds__(64) k_irc_trace_accessibility(IrcTraceCtx c) {
    extern __shared__ uint32_t lds_stack[];
    irc_trace_accessibility_blocks<QUAD>(c, lds_stack, blockIdx.x, gridDim.x);
}

struct IrcTraceResult { V3 incident_radiance, direction, hit_pos; };
// QUAD: four lanes per path (kj_bvh.hpp: bvh_trace_quad). All four run the shading code below on the same inputs; the ray counters and
// the cache lookup (atomics, or the recorded request) are lane 0's, whose result is handed to the others.
// ircache_trace_common.inc.hlsl:37-227 (MAX_PATH_LENGTH = 1)
template <bool QUAD>
KJ_D IrcTraceResult ircache_trace(const IrcTraceCtx& c, const IrcVertex& entry, uint32_t sample_params, uint32_t life, uint32_t* stack, uint32_t stride, uint32_t request_slot, uint32_t request_key) {
    const FrameConstants& fc = *c.fc;
    const bool lead = !QUAD || (threadIdx.x & 3u) == 0u;
    uint32_t rng = hash1(sample_params >> 4u);
    const V3 ray_o = entry.position, ray_d = irc_sample_direction(sample_params);
    IrcTraceResult result;
    result.direction = ray_d;
    result.hit_pos = v3(0.0f);
    V3 irradiance_sum = v3(0.0f);
    irc_count_path_rays<QUAD>(c.ray_counters, 0);
    const GbufferPathVertex primary_hit = QUAD ? gbuffer_raytrace_quad(c.sc, fc, true, ray_o, ray_d, 0.0f, FLT_MAX, 1, false, stack, stride, RayCone::from_spread_angle(0.1f))
                                               : gbuffer_raytrace(c.sc, fc, ray_o, ray_d, 0.0f, FLT_MAX, 1, false, stack, stride, nullptr, RayCone::from_spread_angle(0.1f));   // ircache_trace_common.inc.hlsl:83
    if (primary_hit.is_hit) {
        result.hit_pos = primary_hit.position;
        const V3 to_light_norm = sun_direction(fc);
        irc_count_path_rays<QUAD>(c.ray_counters, 1);
        const bool is_shadowed = QUAD ? rt_is_shadowed_quad(c.sc, true, primary_hit.position, to_light_norm, 1e-4f, FLT_MAX, stack, stride)
                                      : rt_is_shadowed(c.sc, primary_hit.position, to_light_norm, 1e-4f, FLT_MAX, stack, stride);
        const GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
        const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
        const V3 wi = to_local(tangent_to_world, to_light_norm);
        V3 wo = to_local(tangent_to_world, -ray_d);
        if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
        LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(c.brdf_fg_lut, gbuffer, wo.z);
        brdf.roughness = lerp(brdf.roughness, 1.0f, 0.5f);  // FIREFLY_SUPPRESSION with roughness_bias = 0.5
        const V3 brdf_value = layered_brdf_evaluate_directional_light(brdf, wo, wi);
        const float4 sc4 = *c.sun_color;
        const V3 light_radiance = is_shadowed ? v3(0.0f) : V3{sc4.x, sc4.y, sc4.z};
        irradiance_sum += brdf_value * light_radiance * fmaxf(0.0f, wi.z);
        irradiance_sum += gbuffer.emissive;
        if (fc.triangle_light_count > 0 && c.sc.light_count > 0) {
            const float light_selection_pmf = 1.0f / float(fc.triangle_light_count);
            const uint32_t light_idx = hash1_mut(rng) % fc.triangle_light_count;
            V2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const KjTriangleLight tl = c.sc.lights[min(light_idx, c.sc.light_count - 1u)];
            const V3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
            const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
            const V3 to_light_ws = ls.pos - primary_hit.position;
            const float dist2 = dot(to_light_ws, to_light_ws);
            const V3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist2));
            const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist2;
            if (to_psa_metric > 0.0f) {
                const V3 wi2 = to_local(tangent_to_world, to_light_norm_ws);
                irc_count_path_rays<QUAD>(c.ray_counters, 1);
                const bool sh = QUAD ? rt_is_shadowed_quad(c.sc, true, primary_hit.position, to_light_norm_ws, 1e-3f, sqrtf(dist2) - 2e-3f, stack, stride)
                                     : rt_is_shadowed(c.sc, primary_hit.position, to_light_norm_ws, 1e-3f, sqrtf(dist2) - 2e-3f, stack, stride);
                if (!sh) irradiance_sum += V3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * layered_brdf_evaluate(brdf, wo, wi2) / ls.pdf * to_psa_metric / light_selection_pmf;
            }
        }
        V3 cached = v3(0.0f);
        if (lead) cached = ircache_lookup<true>(c.ic, fc, entry.position, primary_hit.position, gbuffer.normal, 1u + life / IRC_LIFE_PER_RANK, rng, false, request_slot, request_key);
        if (QUAD) cached = quad_broadcast0(cached);
        irradiance_sum += cached * gbuffer.albedo;
    } else {
        result.hit_pos = ray_o + ray_d * 1000.0f;
        irradiance_sum += xyz(sample_cube_rgba16f(c.sky_cube, c.sky_cube_width, ray_d));
    }
    result.incident_radiance = irradiance_sum;
    return result;
}
// ircache_validate.rgen.hlsl:44-131
template <bool QUAD>
KJ_D void irc_validate_blocks(const IrcTraceCtx& c, uint32_t* lds_stack, uint32_t block_, uint32_t nblocks_) {
    const IrcacheView& ic = c.ic;
    const FrameConstants& fc = *c.fc;
    IRC_PATH_LOOP(ic.meta[IRC_META_TRACING_ALLOC_COUNT], IRC_VALIDATION_SAMPLES_PER_FRAME) {
        const uint32_t entry_idx = ic.entry_indirection[d / IRC_VALIDATION_SAMPLES_PER_FRAME];
        const uint32_t sample_idx = d % IRC_VALIDATION_SAMPLES_PER_FRAME;
        const uint32_t life = ic.life[entry_idx];
        const uint32_t sp = irc_sample_params(IRC_VALIDATION_SAMPLES_PER_FRAME, entry_idx, sample_idx, fc.frame_index);
        const size_t output_idx = size_t(entry_idx) * IRC_AUX_STRIDE + (sp % IRC_OCTA_DIMS2);
        const float4 r0 = ic.aux[output_idx];
        Reservoir1spp r = Reservoir1spp::from_raw(make_uint2(asuint(r0.x), asuint(r0.y)));
        if (r.M > 0) {
            float4 pv = ic.aux[output_idx + IRC_OCTA_DIMS2];
            pv.x *= fc.pre_exposure_delta; pv.y *= fc.pre_exposure_delta; pv.z *= fc.pre_exposure_delta;
            const IrcVertex prev_entry = irc_unpack_vertex(ic.aux[output_idx + IRC_OCTA_DIMS2 * 2]);
            const IrcTraceResult prev_traced = ircache_trace<QUAD>(c, prev_entry, r.payload, life, stack, stride, c.request_slot_base + d, (3u << 28) | d);
            const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(prev_traced.direction, prev_entry.normal)));
            const V3 a = prev_traced.incident_radiance * limiter;
            const V3 b{pv.x, pv.y, pv.z};
            const V3 dist3 = vabs(a - b) / (a + b);
            const float dist = fmaxf(dist3.x, fmaxf(dist3.y, dist3.z));
            const float invalidity = smoothstep(0.1f, 0.5f, dist);
            r.M = fmaxf(0.0f, fminf(r.M, exp2f(log2f(float(IRC_RESTIR_M_CLAMP)) * (1.0f - invalidity))));
            if (lead) {
                const uint2 raw = r.as_raw();
                float2* dst = (float2*)&ic.aux[output_idx];
                *dst = make_float2(asfloat(raw.x), asfloat(raw.y));
                ic.aux[output_idx + IRC_OCTA_DIMS2] = make_float4(a.x, a.y, a.z, pv.w);
            }
        }
    }
}
template <bool QUAD>
__global__ void __launch_bounds__(64) k_irc_validate(IrcTraceCtx c) {
    extern __shared__ uint32_t lds_stack[];
    irc_validate_blocks<QUAD>(c, lds_stack, blockIdx.x, gridDim.x);
}
// trace_irradiance.rgen.hlsl:44-145
template <bool QUAD>
KJ_D void irc_trace_irradiance_blocks(const IrcTraceCtx& c, uint32_t* lds_stack, uint32_t block_, uint32_t nblocks_) {
    const IrcacheView& ic = c.ic;
    const FrameConstants& fc = *c.fc;
    IRC_PATH_LOOP(ic.meta[IRC_META_TRACING_ALLOC_COUNT], IRC_SAMPLES_PER_FRAME) {
        const uint32_t entry_idx = ic.entry_indirection[d / IRC_SAMPLES_PER_FRAME];
        const uint32_t sample_idx = d % IRC_SAMPLES_PER_FRAME;
        const uint32_t life = ic.life[entry_idx];
        const float4 packed_entry = ic.spatial[entry_idx];
        const IrcVertex entry = irc_unpack_vertex(packed_entry);
        uint32_t rng = hash1(hash1(entry_idx) + fc.frame_index);
        const uint32_t sp = irc_sample_params(IRC_SAMPLES_PER_FRAME, entry_idx, sample_idx, fc.frame_index);
        const IrcTraceResult traced = ircache_trace<QUAD>(c, entry, sp, life, stack, stride, c.request_slot_base + KjIrcache::REQ_E + d, (4u << 28) | d);
        const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(traced.direction, entry.normal)));
        const V3 new_value = traced.incident_radiance * limiter;
        StreamState stream_state{0, 0};
        Reservoir1spp reservoir = Reservoir1spp::create();
        reservoir.init_with_stream(sRGB_to_luminance(new_value), 1.0f, stream_state, sp);
        const size_t output_idx = size_t(entry_idx) * IRC_AUX_STRIDE + (sp % IRC_OCTA_DIMS2);
        float4 pv = ic.aux[output_idx + IRC_OCTA_DIMS2];
        const V3 prev_value{pv.x * fc.pre_exposure_delta, pv.y * fc.pre_exposure_delta, pv.z * fc.pre_exposure_delta};
        V3 val_sel = new_value;
        bool selected_new = true;
        {
            const float4 r0 = ic.aux[output_idx];
            Reservoir1spp r = Reservoir1spp::from_raw(make_uint2(asuint(r0.x), asuint(r0.y)));
            if (r.M > 0) {
                r.M = fminf(r.M, 30.0f);
                if (reservoir.update_with_stream(r, sRGB_to_luminance(prev_value), 1.0f, stream_state, r.payload, rng)) {
                    val_sel = prev_value;
                    selected_new = false;
                }
            }
        }
        reservoir.finish_stream(stream_state);
        if (lead) {
            const uint2 raw = reservoir.as_raw();
            float2* dst = (float2*)&ic.aux[output_idx];
            *dst = make_float2(asfloat(raw.x), asfloat(raw.y));
            ic.aux[output_idx + IRC_OCTA_DIMS2] = make_float4(val_sel.x, val_sel.y, val_sel.z, reservoir.W);
            if (selected_new) ic.aux[output_idx + IRC_OCTA_DIMS2 * 2] = packed_entry;
        }
    }
}
template <bool QUAD>
__global__ void __launch_bounds__(64) k_irc_trace_irradiance(IrcTraceCtx c) {
    extern __shared__ uint32_t lds_stack[];
    irc_trace_irradiance_blocks<QUAD>(c, lds_stack, blockIdx.x, gridDim.x);
}
// The three ray passes in ONE launch, side by side: blocks [0, n) validate, [n, 2n) trace, [2n, 3n) accessibility (the two long chains are
// dispatched first). This is the schedule the reference asks for: ircache.rs:396-481 records the three passes with `write_no_sync` on every
// buffer they share ("if we use `write_no_sync`, we can overlap with the next pass", :411-412), i.e. WITHOUT barriers between them, and each is
// a few hundred waves -- on any GPU they run concurrently, racing on the aux slots exactly as they race inside one pass. Launched one after
// the other (each a chain of ~100 dependent traversal steps, ~0.1 ms regardless of its size) they are the longest link of the frame-to-frame
// cycle: this frame's ray passes -> next frame's cache rays -> next frame's ray passes. Not in the cache's deterministic mode, whose snapshots
// define an order between the passes (kj_ircache_trace_irradiance).
template <bool QUAD>
__global__ void __launch_bounds__(64) k_irc_ray_passes(IrcTraceCtx c, uint32_t blocks_per_pass) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t pass = blockIdx.x / blocks_per_pass, block = blockIdx.x % blocks_per_pass;
    if (pass == 0u) irc_validate_blocks<QUAD>(c, lds_stack, block, blocks_per_pass);
    else if (pass == 1u) irc_trace_irradiance_blocks<QUAD>(c, lds_stack, block, blocks_per_pass);
    else irc_trace_accessibility_blocks<QUAD>(c, lds_stack, block, blocks_per_pass);
}
// ---- the CHAIN schedule of the three ray passes (round 5; the default with four lanes per path): ONE launch in which every aux slot still sees its own passes in the
// reference's recording order -- accessibility, then validation, then the new sample (ircache.rs:396-481) -- while the RAYS of the three run side by side.
// The passes of one frame meet on the same slots: validation and tracing both work on the four octahedral cells irc_sample_params() picks for the frame (same formula,
// same count), accessibility on all sixteen. Per slot the dependency is only in the arithmetic at the END of each pass (the reservoir's M after accessibility feeds
// validation's clamp; validation's reservoir and radiance feed the new sample's merge); the three rays themselves depend on nothing the others write. So:
//   * a CHAIN item = one (entry, sample) = one slot, on EIGHT lanes: quad A walks the accessibility ray, then validation's path (closest hit + shadow rays + lookup);
//     quad B walks the new sample's path at the same time (same instruction stream, its own arguments); A's results go to B through lane shuffles -- as the packed
//     values the sequential passes would have stored and re-read -- and B merges and stores. A's own stores would be overwritten by B's: they are not made.
//   * the twelve slots of an entry that neither validation nor tracing touch this frame get their accessibility ray from a plain quad (second block range).
// Own-slot results are those of the three launches. What differs is what a pass sees of OTHER entries while it runs (the lookups' reads of contributions and, in the
// racy mode, their atomics): with the three launches tracing's lookups see every validation update of the frame; here they see whichever have happened. The racy
// mode races on these inside a pass anyway; the deterministic mode defines them: lookups of validation AND tracing read the snapshot taken before the launch (one
// snapshot instead of two; oracle: okj_ircache_set_chain_schedule). The launch is as long as its longest path chain instead of the sum of three.
KJ_D uint32_t irc_octa_swizzle(uint32_t xy) { return xy ^ ((xy & 4u) >> 2u); }       // irc_sample_params()'s cell order
KJ_D void irc_count_quads(unsigned long long* counters, int which, bool active) {     // one count per active quad; called in wave-uniform control flow
#if !defined(__HIP_DEVICE_COMPILE__)      // the tests' CPU stand-in for HIP: its lanes are not in lockstep after a divergent section, a vote here would mix call sites
    if (active && (threadIdx.x & 3u) == 0u) atomicAdd(&counter_slot(counters)[which], 1ull);
    return;
#endif
    const unsigned long long m = __ballot(active) & 0x1111111111111111ull;
    if (m != 0ull && (__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(&counter_slot(counters)[which], (unsigned long long)__popcll(m));
}
KJ_D void irc_chain_blocks(const IrcTraceCtx& c, uint32_t* lds_stack, uint32_t block, uint32_t nblocks) {
    const IrcacheView& ic = c.ic;
    const FrameConstants& fc = *c.fc;
    const uint32_t octet = threadIdx.x >> 3, half = (threadIdx.x >> 2) & 1u;
    const bool lead = (threadIdx.x & 3u) == 0u;
    const int a_lead_lane = int(threadIdx.x & ~7u);
    uint32_t* const stack = lds_stack + (threadIdx.x >> 2);
    const uint32_t stride = 16u;
    const uint32_t n_items = ic.meta[IRC_META_TRACING_ALLOC_COUNT] * IRC_SAMPLES_PER_FRAME;
    static_assert(IRC_SAMPLES_PER_FRAME == IRC_VALIDATION_SAMPLES_PER_FRAME, "validation and tracing work on the same slots of a frame");
#if !defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t emu_mail[8][8];
    __shared__ uint32_t emu_read[8];
    if (int(threadIdx.x) == a_lead_lane) { emu_mail[octet][6] = 0u; emu_read[octet] = 0u; }
    __syncthreads();
#endif
    for (uint32_t base = block * 8u; base < n_items; base += nblocks * 8u) {       // wave-uniform trip count
        const uint32_t d = base + octet;
        const bool item = d < n_items;
        const uint32_t entry_idx = item ? ic.entry_indirection[d / IRC_SAMPLES_PER_FRAME] : 0u;
        const uint32_t sample_idx = d % IRC_SAMPLES_PER_FRAME;
        const uint32_t life = ic.life[entry_idx];
        const float4 packed_entry = ic.spatial[entry_idx];
        const IrcVertex entry = irc_unpack_vertex(packed_entry);
        const uint32_t sp = irc_sample_params(IRC_SAMPLES_PER_FRAME, entry_idx, sample_idx, fc.frame_index);
        const size_t output_idx = size_t(entry_idx) * IRC_AUX_STRIDE + (sp % IRC_OCTA_DIMS2);
        const float4 r0 = ic.aux[output_idx];
        uint2 raw = make_uint2(asuint(r0.x), asuint(r0.y));
        float4 value = ic.aux[output_idx + IRC_OCTA_DIMS2];
        const IrcVertex prev_entry = irc_unpack_vertex(ic.aux[output_idx + IRC_OCTA_DIMS2 * 2]);
        // ---- quad A: trace_accessibility.rgen.hlsl:21-66 on this slot
        const bool acc = item && half == 0u && irc_life_valid(life);
        irc_count_quads(c.ray_counters, 1, acc);
        const bool blocked = rt_is_shadowed_quad(c.sc, acc, entry.position, prev_entry.position - entry.position, 0.001f, 0.999f, stack, stride);
        if (acc && blocked) { Reservoir1spp r = Reservoir1spp::from_raw(raw); r.M *= 0.8f; raw = r.as_raw(); }
        // ---- both quads: one path each (ircache_trace_common.inc.hlsl) -- A re-traces the slot's stored sample from where it was taken, B the frame's new sample
        Reservoir1spp rv = Reservoir1spp::from_raw(raw);
        const bool validate = item && half == 0u && rv.M > 0;
        const bool active = item && (half == 1u || validate);
        IrcTraceResult traced;
        traced.incident_radiance = traced.direction = traced.hit_pos = v3(0.0f);
        if (active) {
            // ONE call site with per-lane arguments: a ternary of two calls is two divergent call sites, which the wave executes one after the other with half its
            // lanes masked -- here quad A's and quad B's paths share every traversal step
            const bool b_side = half == 1u;
            const IrcVertex from = b_side ? entry : prev_entry;
            traced = ircache_trace<true>(c, from, b_side ? sp : rv.payload, life, stack, stride, c.request_slot_base + (b_side ? KjIrcache::REQ_E : 0u) + d, ((b_side ? 4u : 3u) << 28) | d);
        }
        // ---- quad A: ircache_validate.rgen.hlsl:96-128
        if (validate) {
            const V3 b{value.x * fc.pre_exposure_delta, value.y * fc.pre_exposure_delta, value.z * fc.pre_exposure_delta};
            const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(traced.direction, prev_entry.normal)));
            const V3 a = traced.incident_radiance * limiter;
            const V3 dist3 = vabs(a - b) / (a + b);
            const float dist = fmaxf(dist3.x, fmaxf(dist3.y, dist3.z));
            const float invalidity = smoothstep(0.1f, 0.5f, dist);
            rv.M = fmaxf(0.0f, fminf(rv.M, exp2f(log2f(float(IRC_RESTIR_M_CLAMP)) * (1.0f - invalidity))));
            raw = rv.as_raw();
            value = make_float4(a.x, a.y, a.z, value.w);
        }
        // ---- A -> B: the slot as the sequential passes would have left it in memory (packed reservoir, radiance)
#if defined(__HIP_DEVICE_COMPILE__)
        raw.x = __shfl(raw.x, a_lead_lane); raw.y = __shfl(raw.y, a_lead_lane);
        value.x = __shfl(value.x, a_lead_lane); value.y = __shfl(value.y, a_lead_lane); value.z = __shfl(value.z, a_lead_lane); value.w = __shfl(value.w, a_lead_lane);
#else
        {   // the tests' CPU stand-in for HIP runs lanes as fibers that are NOT in lockstep after a divergent section: hand the values over through a mailbox with a
            // sequence number; the poster waits until the previous item's eight readers are through
            const uint32_t seq = base / (nblocks * 8u) + 1u;
            if (int(threadIdx.x) == a_lead_lane) {
                while (emu_read[octet] != 8u * (seq - 1u)) __syncthreads();
                emu_mail[octet][0] = raw.x; emu_mail[octet][1] = raw.y; emu_mail[octet][2] = asuint(value.x); emu_mail[octet][3] = asuint(value.y); emu_mail[octet][4] = asuint(value.z); emu_mail[octet][5] = asuint(value.w);
                emu_mail[octet][6] = seq;
            }
            while (emu_mail[octet][6] != seq) __syncthreads();
            raw.x = emu_mail[octet][0]; raw.y = emu_mail[octet][1]; value = make_float4(asfloat(emu_mail[octet][2]), asfloat(emu_mail[octet][3]), asfloat(emu_mail[octet][4]), asfloat(emu_mail[octet][5]));
            emu_read[octet] += 1u;
        }
#endif
        // ---- quad B: trace_irradiance.rgen.hlsl:75-143
        if (item && half == 1u) {
            uint32_t rng = hash1(hash1(entry_idx) + fc.frame_index);
            const float limiter = lerp(0.5f, 1.0f, smoothstep(-0.1f, 0.0f, dot(traced.direction, entry.normal)));
            const V3 new_value = traced.incident_radiance * limiter;
            StreamState stream_state{0, 0};
            Reservoir1spp reservoir = Reservoir1spp::create();
            reservoir.init_with_stream(sRGB_to_luminance(new_value), 1.0f, stream_state, sp);
            const V3 prev_value{value.x * fc.pre_exposure_delta, value.y * fc.pre_exposure_delta, value.z * fc.pre_exposure_delta};
            V3 val_sel = new_value;
            bool selected_new = true;
            {
                Reservoir1spp r = Reservoir1spp::from_raw(raw);
                if (r.M > 0) {
                    r.M = fminf(r.M, 30.0f);
                    if (reservoir.update_with_stream(r, sRGB_to_luminance(prev_value), 1.0f, stream_state, r.payload, rng)) {
                        val_sel = prev_value;
                        selected_new = false;
                    }
                }
            }
            reservoir.finish_stream(stream_state);
            if (lead) {
                const uint2 out_raw = reservoir.as_raw();
                float2* dst = (float2*)&ic.aux[output_idx];
                *dst = make_float2(asfloat(out_raw.x), asfloat(out_raw.y));
                ic.aux[output_idx + IRC_OCTA_DIMS2] = make_float4(val_sel.x, val_sel.y, val_sel.z, reservoir.W);
                if (selected_new) ic.aux[output_idx + IRC_OCTA_DIMS2 * 2] = packed_entry;
            }
        }
    }
}
// accessibility rays of the slots the frame's validation / tracing do not touch: twelve per entry, one quad each
KJ_D void irc_accessibility_rest_blocks(const IrcTraceCtx& c, uint32_t* lds_stack, uint32_t block, uint32_t nblocks) {
    const IrcacheView& ic = c.ic;
    const uint32_t quad = threadIdx.x >> 2;
    const bool lead = (threadIdx.x & 3u) == 0u;
    uint32_t* const stack = lds_stack + quad;
    const uint32_t period = IRC_OCTA_DIMS2 / IRC_SAMPLES_PER_FRAME, rest = period - 1u;       // 4 cells per sample group, 3 of them untouched
    const uint32_t per_entry = IRC_SAMPLES_PER_FRAME * rest;
    const uint32_t n_items = ic.meta[IRC_META_TRACING_ALLOC_COUNT] * per_entry;
    const uint32_t phase = c.fc->frame_index % period;
    for (uint32_t base = block * 16u; base < n_items; base += nblocks * 16u) {
        const uint32_t d = base + quad;
        const bool item = d < n_items;
        const uint32_t entry_idx = item ? ic.entry_indirection[d / per_entry] : 0u;
        const uint32_t j = d % per_entry;
        const uint32_t octa_idx = irc_octa_swizzle((j / rest) * period + (phase + 1u + j % rest) % period);
        const bool acc = item && irc_life_valid(ic.life[entry_idx]);
        const IrcVertex entry = irc_unpack_vertex(ic.spatial[entry_idx]);
        const size_t output_idx = size_t(entry_idx) * IRC_AUX_STRIDE + octa_idx;
        const float4 r0 = ic.aux[output_idx];
        Reservoir1spp r = Reservoir1spp::from_raw(make_uint2(asuint(r0.x), asuint(r0.y)));
        const IrcVertex prev_entry = irc_unpack_vertex(ic.aux[output_idx + IRC_OCTA_DIMS2 * 2]);
        irc_count_quads(c.ray_counters, 1, acc);
        const bool blocked = rt_is_shadowed_quad(c.sc, acc, entry.position, prev_entry.position - entry.position, 0.001f, 0.999f, stack, 16u);
        if (acc && blocked && lead) {
            r.M *= 0.8f;
            const uint2 raw = r.as_raw();
            float2* dst = (float2*)&ic.aux[output_idx];
            *dst = make_float2(asfloat(raw.x), asfloat(raw.y));
        }
    }
}
__global__ void __launch_bounds__(64) k_irc_ray_chain(IrcTraceCtx c, uint32_t chain_blocks) {
    extern __shared__ uint32_t lds_stack[];
    if (blockIdx.x < chain_blocks) irc_chain_blocks(c, lds_stack, blockIdx.x, chain_blocks);
    else irc_accessibility_rest_blocks(c, lds_stack, blockIdx.x - chain_blocks, gridDim.x - chain_blocks);
}
// sum_up_irradiance.hlsl:34-89 — 16 lanes per entry (one per octahedral cell), shuffle-reduced
__global__ void __launch_bounds__(64) k_irc_sum_up(const FrameConstants* __restrict__ fcp, IrcacheView ic) {
    const uint32_t alloc_count = ic.meta[IRC_META_TRACING_ALLOC_COUNT];
    const uint32_t sub = threadIdx.x >> 4, octa_idx = threadIdx.x & 15u;
    for (uint32_t d0 = blockIdx.x * 4u; d0 < alloc_count; d0 += gridDim.x * 4u) {
        const uint32_t d = d0 + sub;
        const bool active = d < alloc_count;
        const uint32_t entry_idx = active ? ic.entry_indirection[d] : 0u;
        float4 sh[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
        float valid = 0;
        if (active) {
            const uint32_t payload = asuint(ic.aux[size_t(entry_idx) * IRC_AUX_STRIDE + octa_idx].x);
            const V3 dir = irc_sample_direction(payload);
            const float4 contrib = ic.aux[size_t(entry_idx) * IRC_AUX_STRIDE + IRC_OCTA_DIMS2 + octa_idx];
            const V3 radiance = V3{contrib.x, contrib.y, contrib.z} * contrib.w;
            const float4 basis = make_float4(0.282095f * 4.0f, dir.x * 0.488603f * 4.0f, dir.y * 0.488603f * 4.0f, dir.z * 0.488603f * 4.0f);
            const float rad[3] = {radiance.x, radiance.y, radiance.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) sh[k] = make_float4(basis.x * rad[k], basis.y * rad[k], basis.z * rad[k], basis.w * rad[k]);
            valid = contrib.w > 0 ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                sh[k].x += __shfl_xor(sh[k].x, off); sh[k].y += __shfl_xor(sh[k].y, off);
                sh[k].z += __shfl_xor(sh[k].z, off); sh[k].w += __shfl_xor(sh[k].w, off);
            }
            valid += __shfl_xor(valid, off);
        }
        if (active && octa_idx < 3u) {
            const float scale = 1.0f / fmaxf(1.0f, valid);
            const float4 s = sh[octa_idx];
            const V4 new_value = V4{s.x, s.y, s.z, s.w} * scale;
            const float4 pvf = ic.irradiance[entry_idx * 3u + octa_idx];
            V4 prev_value = V4{pvf.x, pvf.y, pvf.z, pvf.w} * fcp->pre_exposure_delta;
            if (!(prev_value.x != 0.0f || prev_value.y != 0.0f || prev_value.z != 0.0f || prev_value.w != 0.0f)) prev_value = new_value;
            const V4 b = lerp(prev_value, new_value, 0.25f);
            ic.irradiance[entry_idx * 3u + octa_idx] = make_float4(b.x, b.y, b.z, b.w);
        }
    }
}


// ------------------------------------------------------------------ deferred updates: replay of recorded lookups
// What a frame's recorded lookups do to the cache is stated so that it is (i) one of the racy reference program's legal outcomes, (ii) a REDUCTION -- min, sum,
// arg-min -- over the records of a cell, in any order and any grouping, so that a rank of the screen-tile split reduces its own strip's records first and only a
// fixed-size SUMMARY travels (no sort, no record lists, no list lengths on the host), and (iii) the same whatever the number of ranks:
//   * an unoccupied cell is allocated if any of its lookups may allocate; the one with the lowest position in the frame (`key`) is "the thread whose atomicOr
//     came first" (lookup.hlsl:118-150): its rank sets the entry's life, its proposal the entry's position. New cells take pool entries in cell order;
//   * an occupied cell's lookups (entry alive, not allocated this frame) all read the entry's life BEFORE any of them lowers it -- the interleaving in which
//     every load of lookup.hlsl:287 precedes every InterlockedMin of :291 --, so lookup i votes iff rank_i <= life0 / IRC_LIFE_PER_RANK, and the life ends as
//     min(life0, min_i rank_i * L);
//   * the position vote (lookup.hlsl:293-301): every voter's InterlockedAdd returns a different count, in ANY order, and the proposal that stays is the last
//     plain store among the accepted voters', again in any order. The outcome chosen: the voter with the smallest dart (ties: lowest key) is the one that drew
//     count v0 -- accepted iff dart <= 1 / (v0 + 1) -- and its store lands last. If even that dart fails nobody can be accepted (later counts only lower the
//     bar). Like the sequential reservoir it picks uniformly among the voters (the darts are i.i.d.), and v0 = 0 whenever a frame is replayed once: the age pass
//     zeroes the counts.
// Round 6: this replaces the (cell, key)-ordered replay of rounds 2-5 (a 51-bit radix sort of ~1.2 M records at 4K + three segmented scans: 0.15-0.20 ms of every
// rank's frame whatever the rank count, and two host syncs for the list lengths).
//
// A summary (IRC_SUMMARY_BYTES, fixed): 64-byte header {alloc_count}, then per ENTRY winner (dart << 32 | key) u64, rank_min u32, votes u32, the winner's
// proposal float4; then up to IRC_MAX_ENTRIES allocation requests (cell order; more cannot be served by the pool anyway) as IrcRequest records.
#define IRC_SUMMARY_HEADER_BYTES 64u
#define IRC_SUMMARY_BYTES (size_t(IRC_SUMMARY_HEADER_BYTES) + size_t(IRC_MAX_ENTRIES) * (8u + 4u + 4u + 16u + 32u))
#define IRC_MAX_SUMMARIES 32u
struct IrcSummaryView { uint32_t* header; unsigned long long* winner; uint32_t* rank_min; uint32_t* votes; float4* proposal; IrcRequest* allocs; };
KJ_HD IrcSummaryView irc_summary_view(void* p) {
    uint8_t* b = (uint8_t*)p;
    IrcSummaryView v;
    v.header = (uint32_t*)b; b += IRC_SUMMARY_HEADER_BYTES;
    v.winner = (unsigned long long*)b; b += size_t(IRC_MAX_ENTRIES) * 8;
    v.rank_min = (uint32_t*)b; b += size_t(IRC_MAX_ENTRIES) * 4;
    v.votes = (uint32_t*)b; b += size_t(IRC_MAX_ENTRIES) * 4;
    v.proposal = (float4*)b; b += size_t(IRC_MAX_ENTRIES) * 16;
    v.allocs = (IrcRequest*)b;
    return v;
}
// one launch clears everything a frame's records need cleared (round 5: ~30 fills per rank and frame, 0.07-0.13 ms): word ranges with a pattern; a range flagged
// `own` is one of the cache's own two slot ranges and is cleared up to last frame's path count only
#define IRC_CLEAR_MAX 12
struct IrcClearSegs { uint32_t* p[IRC_CLEAR_MAX]; uint32_t words[IRC_CLEAR_MAX]; uint32_t pattern[IRC_CLEAR_MAX]; uint32_t own[IRC_CLEAR_MAX]; uint32_t n; };
__global__ void __launch_bounds__(256) k_irc_clear_segments(IrcClearSegs sg, const uint32_t* __restrict__ meta) {
    for (uint32_t k = 0; k < sg.n; ++k) {
        const uint32_t words = sg.own[k] ? min(meta[IRC_META_TRACING_ALLOC_COUNT] * IRC_SAMPLES_PER_FRAME, sg.words[k]) : sg.words[k];
        uint32_t* const p = sg.p[k];
        const uint32_t pat = sg.pattern[k];
        // 16-byte stores where the range allows (every range here starts on a 16-byte boundary or is short)
        if ((uintptr_t(p) & 15u) == 0u) {
            const uint32_t quads = words / 4u;
            for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < quads; i += gridDim.x * 256u) ((uint4*)p)[i] = make_uint4(pat, pat, pat, pat);
            for (uint32_t i = quads * 4u + blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) p[i] = pat;
        } else
            for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) p[i] = pat;
    }
}
// slot ranges of one reduce launch; `own`: scanned up to this frame's path count only (slot = path index)
#define IRC_REDUCE_MAX_RANGES 6
struct IrcReduceRanges { uint32_t first[IRC_REDUCE_MAX_RANGES], count[IRC_REDUCE_MAX_RANGES], own[IRC_REDUCE_MAX_RANGES]; uint32_t n; };
KJ_D bool irc_replay_entry(const IrcacheView& ic, uint2 gm, uint32_t* entry, uint32_t* life0) {
    if ((gm.y & IRC_META_OCCUPIED) == 0 || (gm.y & IRC_META_JUST_ALLOCATED) != 0) return false;
    *entry = gm.x; *life0 = ic.life[gm.x];
    return *life0 < IRC_LIFE_RECYCLE;          // the guard of lookup.hlsl:287
}
// Records -> per-entry (rank_min, votes, winner) and per-cell allocation winner (key << 32 | record index). A workgroup first reduces into an LDS hash table
// keyed by cell (a cell near the camera receives tens of thousands of lookups per frame: ~6 ns per same-address device atomic would serialise to 0.3 ms), then
// flushes one device atomic per touched cell and field. `cells`: the 4-byte-per-slot copy of the records' cell field (0xffffffff = no record), or null for a
// plain list (then rq[i].cell itself).
#define IRC_RED_SLOTS 1024u
#ifndef IRC_RED_PER_THREAD
#define IRC_RED_PER_THREAD 4u      // slots per thread and chunk: 1024-slot chunks (8: 22 us per reduce of a 4K strip, 4: 18.7, 2: 18.0; round 6)
#endif
__global__ void __launch_bounds__(256) k_irc_reduce_requests(IrcacheView ic, const IrcRequest* __restrict__ rq, const uint32_t* __restrict__ cells, IrcReduceRanges rr, IrcSummaryView sum,
                                                              unsigned long long* __restrict__ alloc_min) {
    __shared__ uint32_t h_cell[IRC_RED_SLOTS], h_rank[IRC_RED_SLOTS], h_votes[IRC_RED_SLOTS];
    __shared__ unsigned long long h_win[IRC_RED_SLOTS], h_alloc[IRC_RED_SLOTS];
    for (uint32_t t = threadIdx.x; t < IRC_RED_SLOTS; t += 256u) { h_cell[t] = 0xffffffffu; h_rank[t] = 0xffffffffu; h_votes[t] = 0u; h_win[t] = ~0ull; h_alloc[t] = ~0ull; }
    __syncthreads();
    const uint32_t per_block = 256u * IRC_RED_PER_THREAD;
    const uint32_t used_paths = ic.meta[IRC_META_TRACING_ALLOC_COUNT] * IRC_SAMPLES_PER_FRAME;
    uint32_t chunk0 = 0;      // chunks (of per_block slots) of the ranges before range k
    for (uint32_t k = 0; k < rr.n; ++k) {
        const uint32_t n = rr.own[k] ? min(rr.count[k], used_paths) : rr.count[k];
        const uint32_t chunks = (n + per_block - 1u) / per_block;
        // chunk c of this range belongs to block (chunk0 + c) % gridDim.x
        for (uint32_t c = (blockIdx.x + gridDim.x - chunk0 % gridDim.x) % gridDim.x; c < chunks; c += gridDim.x) {
            const uint32_t i0 = c * per_block + threadIdx.x;
            for (uint32_t j = 0; j < IRC_RED_PER_THREAD; ++j) {
                const uint32_t li = i0 + j * 256u;      // coalesced over the workgroup
                if (li >= n) break;
                const uint32_t i = rr.first[k] + li;
                const uint32_t cell = cells ? cells[i] : rq[i].cell;
                if (cell == 0xffffffffu) continue;
                const uint2 gm = ic.grid_meta[cell];
                const uint32_t key = rq[i].key, bits = rq[i].bits;
                uint32_t entry = 0, life0 = 0;
                const bool unoccupied = (gm.y & IRC_META_OCCUPIED) == 0;
                const bool may_alloc = unoccupied && (bits & 0x100u) == 0u;
                const bool replay = !unoccupied && irc_replay_entry(ic, gm, &entry, &life0);
                if (!may_alloc && !replay) continue;
                const uint32_t rank = bits & 0xffu;
                const bool voter = replay && rank <= life0 / IRC_LIFE_PER_RANK;
                const unsigned long long win = ((unsigned long long)asuint(rq[i].dart) << 32) | key, alc = ((unsigned long long)key << 32) | i;
                // find or claim the cell's slot in the workgroup's table
                uint32_t h = (cell * 2654435761u) >> 22, slot = 0xffffffffu;
                for (uint32_t probe = 0; probe < 16u; ++probe) {
                    const uint32_t old = atomicCAS(&h_cell[h], 0xffffffffu, cell);
                    if (old == 0xffffffffu || old == cell) { slot = h; break; }
                    h = (h + 1u) & (IRC_RED_SLOTS - 1u);
                }
                if (slot != 0xffffffffu) {
                    if (may_alloc) atomicMin(&h_alloc[slot], alc);
                    else { atomicMin(&h_rank[slot], rank); if (voter) { atomicAdd(&h_votes[slot], 1u); atomicMin(&h_win[slot], win); } }
                } else if (may_alloc) atomicMin(&alloc_min[cell], alc);      // table full around this hash: straight to memory
                else { atomicMin(&sum.rank_min[entry], rank); if (voter) { atomicAdd(&sum.votes[entry], 1u); atomicMin(&sum.winner[entry], win); } }
            }
        }
        chunk0 += chunks;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < IRC_RED_SLOTS; t += 256u) {
        const uint32_t cell = h_cell[t];
        if (cell == 0xffffffffu) continue;
        if (h_alloc[t] != ~0ull) atomicMin(&alloc_min[cell], h_alloc[t]);
        if (h_rank[t] != 0xffffffffu) {
            const uint32_t entry = ic.grid_meta[cell].x;
            atomicMin(&sum.rank_min[entry], h_rank[t]);
            if (h_votes[t]) { atomicAdd(&sum.votes[entry], h_votes[t]); atomicMin(&sum.winner[entry], h_win[t]); }
        }
    }
}
// second pass over the records: the one whose (dart, key) won its entry's vote leaves its proposal in the summary
__global__ void __launch_bounds__(256) k_irc_reduce_winners(IrcacheView ic, const IrcRequest* __restrict__ rq, const uint32_t* __restrict__ cells, IrcReduceRanges rr, IrcSummaryView sum) {
    const uint32_t used_paths = ic.meta[IRC_META_TRACING_ALLOC_COUNT] * IRC_SAMPLES_PER_FRAME;
    for (uint32_t k = 0; k < rr.n; ++k) {
        const uint32_t n = rr.own[k] ? min(rr.count[k], used_paths) : rr.count[k];
        for (uint32_t li = blockIdx.x * 256u + threadIdx.x; li < n; li += gridDim.x * 256u) {
            const uint32_t i = rr.first[k] + li;
            const uint32_t cell = cells ? cells[i] : rq[i].cell;
            if (cell == 0xffffffffu) continue;
            const uint2 gm = ic.grid_meta[cell];
            if ((gm.y & IRC_META_OCCUPIED) == 0 || (gm.y & IRC_META_JUST_ALLOCATED) != 0) continue;
            const unsigned long long w = sum.winner[gm.x];
            if (uint32_t(w) != rq[i].key || uint32_t(w >> 32) != asuint(rq[i].dart)) continue;      // keys are unique in a frame: at most one record matches
            sum.proposal[gm.x] = rq[i].proposal;
        }
    }
}
// ---- cells with an allocation winner, in cell order: count per 1024-cell block, then every block sums the counts in front of it and emits
#define IRC_CELL_BLOCKS (IRC_MAX_GRID_CELLS / 1024u)
static_assert(IRC_MAX_GRID_CELLS % 1024u == 0, "cell blocks");
__global__ void __launch_bounds__(256) k_irc_alloc_count(const unsigned long long* __restrict__ alloc_min, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0u) s_cnt = 0u;
    __syncthreads();
    const uint32_t c0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) cnt += alloc_min[c0 + k] != ~0ull ? 1u : 0u;
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0u) block_counts[blockIdx.x] = s_cnt;
}
static_assert((IRC_CELL_BLOCKS + 4u) % 2u == 0u, "the source table behind the block counts must be 8-byte aligned");
struct IrcSummarySources { const void* p[IRC_MAX_SUMMARIES]; uint32_t n; };
// the merge kernels index the sources by a run-time number: from a table in memory (indexing a by-value kernel argument makes every thread copy the whole struct
// to scratch first: k_irc_alloc_emit<true> took 33 us against 5 us for its list-writing twin, round 6). One 32-thread launch stores the table.
__global__ void k_irc_store_sources(IrcSummarySources src, const void** __restrict__ table) { if (threadIdx.x < IRC_MAX_SUMMARIES) table[threadIdx.x] = threadIdx.x < src.n ? src.p[threadIdx.x] : nullptr; }
// APPLY = false: this rank's winners -> the summary's allocation list (cell order, truncated to IRC_MAX_ENTRIES: the pool cannot serve more, and the lowest
// cells are the ones that get entries), records read from `rq`. APPLY = true: the merged winners take pool entries (locator = source << 16 | index in that
// source's list). Both leave alloc_min all-ones behind.
template <bool APPLY>
__global__ void __launch_bounds__(256) k_irc_alloc_emit(IrcacheView ic, unsigned long long* __restrict__ alloc_min, const uint32_t* __restrict__ block_counts, const IrcRequest* __restrict__ rq,
                                                         IrcSummaryView sum, const void* const* __restrict__ src, uint32_t* __restrict__ out_total) {
    __shared__ uint32_t s_scan[256], s_base, s_total;
    // the counts in front of this block (and all of them): IRC_CELL_BLOCKS = 384 words, two per thread
    uint32_t before = 0, all = 0;
    for (uint32_t b = threadIdx.x; b < IRC_CELL_BLOCKS; b += 256u) { const uint32_t v = block_counts[b]; all += v; if (b < blockIdx.x) before += v; }
    if (threadIdx.x == 0u) { s_base = 0u; s_total = 0u; }
    __syncthreads();
    if (before) atomicAdd(&s_base, before);
    if (all) atomicAdd(&s_total, all);
    __syncthreads();
    // nothing to hand out anywhere (most frames of a settled cache, every summary of the cache's own passes): leave before touching the 3 MB of winners
    if (s_total == 0u) { if (blockIdx.x == 0u && threadIdx.x == 0u) { if (APPLY) *out_total = 0u; else sum.header[0] = 0u; } return; }
    const uint32_t c0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    unsigned long long v[4];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) { v[k] = alloc_min[c0 + k]; cnt += v[k] != ~0ull ? 1u : 0u; }
    s_scan[threadIdx.x] = cnt;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t add = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t r = s_base + s_scan[threadIdx.x] - cnt;
    if (blockIdx.x == 0u && threadIdx.x == 0u) { if (APPLY) *out_total = s_total; else sum.header[0] = min(s_total, uint32_t(IRC_MAX_ENTRIES)); }
    __shared__ uint32_t s_entry_max;      // APPLY: the highest entry index this block hands out + 1 -- ONE device atomic per block (an atomicMax per new cell on the one
    if (APPLY) { if (threadIdx.x == 0u) s_entry_max = 0u; __syncthreads(); }      // meta word serialised: 33 us for a few thousand new cells per 4K frame, round 6)
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        if (v[k] == ~0ull) continue;
        const uint32_t cell = c0 + k, loc = uint32_t(v[k]);
        alloc_min[cell] = ~0ull;
        if (!APPLY) {
            if (r < IRC_MAX_ENTRIES) { IrcRequest a = rq[loc]; a.cell = cell; a.dart = 0.0f; sum.allocs[r] = a; }
        } else {
            const uint32_t alloc_idx = ic.meta[IRC_META_ALLOC_COUNT] + r;
            if (alloc_idx < IRC_MAX_ENTRIES) {                                           // else: pool exhausted, the cell stays empty
                const IrcRequest a = irc_summary_view((void*)src[loc >> 16]).allocs[loc & 0xffffu];
                const uint32_t entry_idx = ic.pool[alloc_idx];
                atomicMax(&s_entry_max, entry_idx + 1u);
                ic.life[entry_idx] = (a.bits & 0xffu) * IRC_LIFE_PER_RANK;
                ic.entry_cell[entry_idx] = cell;
                ic.grid_meta[cell] = make_uint2(entry_idx, ic.grid_meta[cell].y | IRC_META_OCCUPIED | IRC_META_JUST_ALLOCATED);
                ic.reposition_proposal[entry_idx] = a.proposal;
            }
        }
        ++r;
    }
    if (APPLY) {
        __syncthreads();
        if (threadIdx.x == 0u && s_entry_max != 0u) atomicMax(&ic.meta[IRC_META_ENTRY_COUNT], s_entry_max);
    }
}
// ---- apply: the merge of every source's summary
__global__ void __launch_bounds__(256) k_irc_merge_entries(IrcacheView ic, const void* const* __restrict__ src, uint32_t n_src) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= IRC_MAX_ENTRIES) return;
    uint32_t rank_min = 0xffffffffu, votes = 0u, win_src = 0u;
    unsigned long long win = ~0ull;
    for (uint32_t s = 0; s < n_src; ++s) {
        const IrcSummaryView v = irc_summary_view((void*)src[s]);
        rank_min = min(rank_min, v.rank_min[e]);
        votes += v.votes[e];
        const unsigned long long w = v.winner[e];
        if (w < win) { win = w; win_src = s; }
    }
    if (rank_min == 0xffffffffu) return;      // no replayable lookup reached this entry
    ic.life[e] = min(ic.life[e], rank_min * IRC_LIFE_PER_RANK);
    if (votes) {
        const uint32_t v0 = ic.reposition_proposal_count[e];
        ic.reposition_proposal_count[e] = v0 + votes;
        if (asfloat(uint32_t(win >> 32)) <= 1.0f / (float(v0) + 1.0f)) ic.reposition_proposal[e] = irc_summary_view((void*)src[win_src]).proposal[e];
    }
}
__global__ void __launch_bounds__(256) k_irc_merge_allocs(const void* const* __restrict__ src, unsigned long long* __restrict__ alloc_min) {
    const uint32_t s = blockIdx.y, i = blockIdx.x * 256u + threadIdx.x;
    const IrcSummaryView v = irc_summary_view((void*)src[s]);
    if (i >= min(v.header[0], uint32_t(IRC_MAX_ENTRIES))) return;
    atomicMin(&alloc_min[v.allocs[i].cell], ((unsigned long long)v.allocs[i].key << 32) | (s << 16) | i);
}
__global__ void k_irc_request_finish(IrcacheView ic, const uint32_t* __restrict__ total) {
    ic.meta[IRC_META_ALLOC_COUNT] = min(ic.meta[IRC_META_ALLOC_COUNT] + *total, uint32_t(IRC_MAX_ENTRIES));
}

// ================================================================== host
#include "kj_ircache_host.hpp"

IrcacheView KjIrcache::view() const {
    IrcacheView v;
    v.meta = (uint32_t*)meta.p;
    v.grid_meta = (uint2*)grid_meta[cur].p;
    v.entry_cell = (uint32_t*)entry_cell.p;
    v.spatial = (float4*)spatial.p;
    v.irradiance = (float4*)irradiance.p;
    v.aux = (float4*)aux.p;
    v.aux_read = (const float4*)aux.p;
    v.life = (uint32_t*)life.p;
    v.pool = (uint32_t*)pool.p;
    v.reposition_proposal = (float4*)reposition_proposal.p;
    v.reposition_proposal_count = (uint32_t*)reposition_proposal_count.p;
    v.entry_indirection = (const uint32_t*)entry_indirection.p;
    v.requests = deferred ? (IrcRequest*)requests.p : nullptr;
    v.request_cells = deferred ? (uint32_t*)request_cells.p : nullptr;
    return v;
}

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_ircache_create(KjDevice* dev, KjIrcache** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjIrcache* c = new KjIrcache();
    c->dev = dev;
    if (kj_debug_getenv("KJ_IRC_SIDE_BY_SIDE") && atoi(kj_debug_getenv("KJ_IRC_SIDE_BY_SIDE")) != 0) c->ray_pass_schedule = KJ_IRC_PASSES_SIDE_BY_SIDE;
    if (const char* v = kj_debug_getenv("KJ_IRC_SCHEDULE")) { const int k = atoi(v); if (k >= 0 && k <= 2) c->ray_pass_schedule = uint32_t(k); }      // A/B runs
    hipError_t e = hipSuccess;
    auto A = [&](kj::DevBuf& b, size_t n) { if (e == hipSuccess) e = b.alloc(n); };
    A(c->meta, 32); A(c->grid_meta[0], size_t(IRC_MAX_GRID_CELLS) * 8); A(c->grid_meta[1], size_t(IRC_MAX_GRID_CELLS) * 8);
    A(c->entry_cell, IRC_MAX_ENTRIES * 4); A(c->spatial, IRC_MAX_ENTRIES * 16); A(c->irradiance, size_t(IRC_MAX_ENTRIES) * 48);
    A(c->aux, size_t(IRC_MAX_ENTRIES) * 64 * 16); A(c->life, IRC_MAX_ENTRIES * 4); A(c->pool, IRC_MAX_ENTRIES * 4);
    A(c->entry_indirection, (IRC_MAX_ENTRIES + 64) * 4); A(c->reposition_proposal, IRC_MAX_ENTRIES * 16);
    A(c->reposition_proposal_count, IRC_MAX_ENTRIES * 4); A(c->occupancy, IRC_MAX_ENTRIES * 4); A(c->ray_counters, KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8);
    if (e != hipSuccess) { delete c; set_last_error("ircache allocation failed: %s", hipGetErrorString(e)); return KJ_ERR_OUT_OF_MEMORY; }
    *out = c;
    return KJ_OK;
}
void kj_ircache_destroy(KjIrcache* c) { delete c; }

// IrcacheRenderer::update_eye_position (ircache.rs:126-141)
KjStatus kj_ircache_update_eye_position(KjIrcache* c, const float eye[3]) {
    KJ_REQUIRE(c && eye, "null argument");
    if (!c->enable_scroll) return KJ_OK;
    for (int k = 0; k < 3; ++k) c->grid_center[k] = eye[k];
    for (int cs = 0; cs < 12; ++cs) {
        const float cell_diameter = IRC_GRID_CELL_DIAMETER * float(1 << cs);
        for (int k = 0; k < 3; ++k) {
            c->prev_scroll[cs][k] = c->cur_scroll[cs][k];
            c->cur_scroll[cs][k] = int(floorf(eye[k] / cell_diameter)) - int(IRC_CASCADE_SIZE) / 2;
        }
    }
    return KJ_OK;
}
// IrcacheRenderer::constants + grid_center (ircache.rs:143-162): fills the ircache members of the frame constants
KjStatus kj_ircache_constants(KjIrcache* c, KjFrameConstants* fc) {
    KJ_REQUIRE(c && fc, "null argument");
    for (int k = 0; k < 3; ++k) fc->ircache_grid_center[k] = c->grid_center[k];
    fc->ircache_grid_center[3] = 1.0f;
    for (int cs = 0; cs < 12; ++cs)
        for (int k = 0; k < 4; ++k) {
            fc->ircache_cascades[cs].origin[k] = k < 3 ? c->cur_scroll[cs][k] : 0;
            fc->ircache_cascades[cs].voxels_scrolled_this_frame[k] = k < 3 ? c->cur_scroll[cs][k] - c->prev_scroll[cs][k] : 0;
        }
    return KJ_OK;
}
KjStatus kj_ircache_set_enable_scroll(KjIrcache* c, uint32_t enable) { KJ_REQUIRE(c, "null argument"); c->enable_scroll = enable != 0; return KJ_OK; }

// IrcacheRenderer::prepare (ircache.rs:168-350)
KjStatus kj_ircache_prepare(KjIrcache* c, void* stream_) {
    KJ_REQUIRE(c && c->dev->fc_dev, "null argument / kj_frame_begin not called");
    // deferred mode: this frame's lookups only RECORD; without a cleared slot array and a later replay the cache would silently stop
    // allocating and refreshing entries
    KJ_REQUIRE(!c->deferred || c->requests_begun, "deferred updates are on: call kj_ircache_begin_requests before kj_ircache_prepare every frame (or kj_ircache_set_deferred_updates(c, 0))");
    c->requests_begun = false;
    hipStream_t s = (hipStream_t)stream_;
    int a = 0, b = 1;
    if (c->parity == 1) std::swap(a, b);
    uint32_t* freed = nullptr;
    if (c->deferred) {
        if (c->freed.bytes != IRC_MAX_ENTRIES * 4) { KJ_TRY_HIP(c->freed.alloc(IRC_MAX_ENTRIES * 4, s)); c->begin_cleared_frame_state = false; }
        if (!c->begin_cleared_frame_state) KJ_TRY_HIP(hipMemsetAsync(c->freed.p, 0, IRC_MAX_ENTRIES * 4, s));      // (normally part of kj_ircache_begin_requests' one clear launch)
        freed = (uint32_t*)c->freed.p;
    }
    if (!c->initialized) {
        hipLaunchKernelGGL(k_irc_clear_pool, dim3(IRC_MAX_ENTRIES / 256), dim3(256), 0, s, (uint32_t*)c->pool.p, (uint32_t*)c->life.p);
        KJ_CHECK_LAUNCH();
        c->initialized = true;
    } else {
        hipLaunchKernelGGL(k_irc_scroll_cascades, dim3(IRC_MAX_GRID_CELLS / 256), dim3(256), 0, s, c->dev->fc_dev, (const uint2*)c->grid_meta[a].p, (uint2*)c->grid_meta[b].p,
                           (uint32_t*)c->entry_cell.p, (float4*)c->irradiance.p, (uint32_t*)c->life.p, (uint32_t*)c->pool.p, (uint32_t*)c->meta.p, freed);
        KJ_CHECK_LAUNCH();
        std::swap(a, b);
        c->parity = (c->parity + 1) % 2;
    }
    c->cur = a;
    const IrcacheView v = c->view();
    hipLaunchKernelGGL(k_irc_age, dim3(IRC_MAX_ENTRIES / 256), dim3(256), 0, s, v, (uint32_t*)c->occupancy.p, freed);
    KJ_CHECK_LAUNCH();
    if (freed) {      // both inclusive scans -- the freed-entry flags and the occupancy flags -- in one launch of two workgroups
        hipLaunchKernelGGL(k_irc_scan, dim3(2), dim3(1024), 0, s, freed, (uint32_t*)c->occupancy.p);
        hipLaunchKernelGGL(k_irc_push_freed, dim3(IRC_MAX_ENTRIES / 256), dim3(256), 0, s, (const uint32_t*)freed, (uint32_t*)c->pool.p, (uint32_t*)c->meta.p);
        hipLaunchKernelGGL(k_irc_pop_freed_count, dim3(1), dim3(1), 0, s, (const uint32_t*)freed, (uint32_t*)c->meta.p);
        KJ_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(k_irc_scan, dim3(1), dim3(1024), 0, s, (uint32_t*)c->occupancy.p, (uint32_t*)nullptr);
        KJ_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_irc_compact, dim3(IRC_MAX_ENTRIES / 256), dim3(256), 0, s, (const uint32_t*)c->meta.p, (const uint32_t*)c->life.p, (const uint32_t*)c->occupancy.p,
                       (uint32_t*)c->entry_indirection.p);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// IrcacheRenderState::trace_irradiance (ircache.rs:360-481): args, reset, accessibility, validate, trace
KjStatus kj_ircache_trace_irradiance(KjIrcache* c, KjScene* scene, const void* sky_cube, uint32_t sky_cube_width, void* stream_) {
    KJ_REQUIRE(c && scene && sky_cube && c->dev->fc_dev, "null argument / kj_frame_begin not called");
    if (!scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    hipStream_t s = (hipStream_t)stream_;
    IrcTraceCtx tc;
    tc.fc = c->dev->fc_dev;
    tc.sc = scene_view(*scene);
    tc.ic = c->view();
    tc.sky_cube = (const uint2*)sky_cube; tc.sky_cube_width = int(sky_cube_width);
    tc.brdf_fg_lut = (const uint2*)c->dev->brdf_fg_lut.p;
    tc.sun_color = (const float4*)c->dev->sun_color.p + c->dev->fc_slot;
    tc.ray_counters = (unsigned long long*)c->ray_counters.p;
    tc.request_slot_base = 2u * c->req_half_pixels;
    const size_t lds = size_t(tc.sc.bvh.stack_entries) * 64 * 4;
    KJ_REQUIRE(lds <= 64 * 1024, "BVH too deep for the LDS traversal stack");
    // The cache's ray passes are small (a few thousand entries x 4 paths: ~107 full waves on 1024 SIMDs) and each path is a chain of
    // ~150 dependent traversal steps, so a pass takes as long as its slowest WAVE. KJ_IRC_LANES=n spreads them thin -- n paths per
    // wave, the other lanes idle -- which measured -2.5 % per frame at n = 8 (each wave's step count is the maximum over fewer
    // divergent paths). Not the default: with 8x as many waves in flight a lookup sees fewer of the same pass' updates, which moves the
    // racy passes further from the sequential oracle (SH rel-L2 on identical state, 1080p city: 1.1e-2 at 64, 2.3e-2 at 8; bar 2e-2).
    static const uint32_t lanes_env = kj_debug_getenv("KJ_IRC_LANES") ? uint32_t(atoi(kj_debug_getenv("KJ_IRC_LANES"))) : 0u;
    tc.lanes = lanes_env >= 1u && lanes_env <= 64u ? lanes_env : 64u;
    // KJ_IRC_PART="i/n" (a measurement switch, scripts/ircache_partition_probe.sh): the three ray passes take every n-th entry only -- what
    // one rank of n would trace if the cache's own rays were dealt out across ranks. Results are then incomplete by construction.
    tc.part_index = 0u; tc.part_count = 1u;
    if (const char* pe = kj_debug_getenv("KJ_IRC_PART")) {
        unsigned pi = 0, pn = 1;
        if (sscanf(pe, "%u/%u", &pi, &pn) == 2 && pn >= 1u && pi < pn) { tc.part_index = pi; tc.part_count = pn; }
        static bool warned = false;      // a variable left over from a profiling script degrades GI: never silently
        if (!warned && tc.part_count > 1u) { fprintf(stderr, "kajiya_amd: KJ_IRC_PART=%s is set: the irradiance cache traces only every %u-th entry (a MEASUREMENT switch; results are incomplete)\n", pe, tc.part_count); warned = true; }
    }
    // Default: four lanes per path (kj_bvh.hpp: bvh_trace_quad) -- 16 paths per wave, a ~75-instruction step instead of ~200, four
    // times the waves. KJ_IRC_QUAD=0: one lane per path.
    static const bool quad = !(kj_debug_getenv("KJ_IRC_QUAD") && atoi(kj_debug_getenv("KJ_IRC_QUAD")) == 0);
    const uint32_t grid = c->dev->num_cus * (quad ? 32u : (tc.lanes < 64u ? 32u : 8u));
    const size_t lds_rays = quad ? quad_stack_bytes() : lds;
    if (!c->begin_cleared_frame_state) KJ_TRY_HIP(hipMemsetAsync(c->ray_counters.p, 0, KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 8, s));
    c->begin_cleared_frame_state = false;
    hipLaunchKernelGGL(k_irc_prepare_trace, dim3(1), dim3(1), 0, s, (uint32_t*)c->meta.p);
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_irc_reset, dim3(grid), dim3(64), 0, s, tc.ic);
    KJ_CHECK_LAUNCH();
    // racy (the reference's) mode, opt-in (kj_ircache_set_ray_passes_side_by_side, or KJ_IRC_SIDE_BY_SIDE=1 at creation): the three passes side by side in one launch, as the reference's barrier-free
    // recording lets them run. Measured on MI355X (round 4, profiles/r04_ab_runs.md): the cache's segment 0.40 -> 0.21 ms at 1080p, the PIPELINED
    // frame unchanged (it is VALU-bound, not waiting for this chain), and the SH sums on identical state move from 1.3e-2 to 5e-2 .. 1.3e-1 of the
    // sequential oracle's (tests/test_gpu_ircache.py; bar 5e-2) -- so it is not the default.
    // the chain (default): one launch, own-slot order kept (k_irc_ray_chain). Needs the four-lanes-per-path form; a partitioned measurement run takes the three launches
    if (quad && tc.part_count == 1u && (c->ray_pass_schedule == KJ_IRC_PASSES_CHAIN || (c->deferred && c->ray_pass_schedule == KJ_IRC_PASSES_SIDE_BY_SIDE))) {
        if (c->deferred) {   // what validation's and tracing's lookups read of other entries: the state before the launch
            if (c->aux_snapshot.bytes != c->aux.bytes) KJ_TRY_HIP(c->aux_snapshot.alloc(c->aux.bytes, s));
            tc.ic.aux_read = (const float4*)c->aux_snapshot.p;
            hipLaunchKernelGGL(k_irc_snapshot_aux, dim3(c->dev->num_cus * 4), dim3(256), 0, s, (const uint32_t*)c->meta.p, (const float4*)c->aux.p, (float4*)c->aux_snapshot.p);
        }
        hipLaunchKernelGGL(k_irc_ray_chain, dim3(grid * 2u), dim3(64), lds_rays, s, tc, grid);
        KJ_CHECK_LAUNCH();
        c->pending_irradiance_sum = true;
        return KJ_OK;
    }
    if (!c->deferred && c->ray_pass_schedule == KJ_IRC_PASSES_SIDE_BY_SIDE) {
        hipLaunchKernelGGL(quad ? k_irc_ray_passes<true> : k_irc_ray_passes<false>, dim3(grid * 3u), dim3(64), lds_rays, s, tc, grid);
        KJ_CHECK_LAUNCH();
        c->pending_irradiance_sum = true;
        return KJ_OK;
    }
    hipLaunchKernelGGL(quad ? k_irc_trace_accessibility<true> : k_irc_trace_accessibility<false>, dim3(grid), dim3(64), lds_rays, s, tc);
    KJ_CHECK_LAUNCH();
    if (c->deferred) {   // lookups inside the next two passes read other entries' aux while those are rewritten: give them a snapshot
        if (c->aux_snapshot.bytes != c->aux.bytes) KJ_TRY_HIP(c->aux_snapshot.alloc(c->aux.bytes, s));
        tc.ic.aux_read = (const float4*)c->aux_snapshot.p;
        hipLaunchKernelGGL(k_irc_snapshot_aux, dim3(c->dev->num_cus * 4), dim3(256), 0, s, (const uint32_t*)c->meta.p, (const float4*)c->aux.p, (float4*)c->aux_snapshot.p);
    }
    hipLaunchKernelGGL(quad ? k_irc_validate<true> : k_irc_validate<false>, dim3(grid), dim3(64), lds_rays, s, tc);
    KJ_CHECK_LAUNCH();
    if (c->deferred) hipLaunchKernelGGL(k_irc_snapshot_aux, dim3(c->dev->num_cus * 4), dim3(256), 0, s, (const uint32_t*)c->meta.p, (const float4*)c->aux.p, (float4*)c->aux_snapshot.p);
    hipLaunchKernelGGL(quad ? k_irc_trace_irradiance<true> : k_irc_trace_irradiance<false>, dim3(grid), dim3(64), lds_rays, s, tc);
    KJ_CHECK_LAUNCH();
    c->pending_irradiance_sum = true;
    return KJ_OK;
}
// IrcacheRenderState::sum_up_irradiance_for_sampling (ircache.rs:487-506)
KjStatus kj_ircache_sum_up_irradiance_for_sampling(KjIrcache* c, void* stream_) {
    KJ_REQUIRE(c && c->dev->fc_dev, "null argument");
    KJ_REQUIRE(c->pending_irradiance_sum, "trace_irradiance must run first (ircache.rs:492 assert)");
#ifndef KJ_IRC_SUMUP_GRID
#define KJ_IRC_SUMUP_GRID 4
#endif
    hipLaunchKernelGGL(k_irc_sum_up, dim3(c->dev->num_cus * KJ_IRC_SUMUP_GRID), dim3(64), 0, (hipStream_t)stream_, c->dev->fc_dev, c->view());
    KJ_CHECK_LAUNCH();
    c->pending_irradiance_sum = false;
    return KJ_OK;
}
// ---- deferred updates: begin (clear the frame's slots and summaries), summarize (reduce slot ranges into a fixed-size summary), apply (merge summaries)
KjStatus kj_ircache_set_deferred_updates(KjIrcache* c, uint32_t enable) {
    KJ_REQUIRE(c, "null argument");
    // racy frames in between overwrite the path count the own-range clears are bounded by: the next begin clears every slot
    if (enable && !c->deferred) c->req_clear_all = true;
    c->deferred = enable != 0;
    return KJ_OK;
}
KjStatus kj_ircache_set_ray_passes_side_by_side(KjIrcache* c, uint32_t enable) { KJ_REQUIRE(c, "null argument"); c->ray_pass_schedule = enable ? KJ_IRC_PASSES_SIDE_BY_SIDE : KJ_IRC_PASSES_CHAIN; return KJ_OK; }
KjStatus kj_ircache_set_ray_pass_schedule(KjIrcache* c, uint32_t schedule) { KJ_REQUIRE(c && schedule <= KJ_IRC_PASSES_CHAIN, "null argument / unknown schedule"); c->ray_pass_schedule = schedule; return KJ_OK; }
KjStatus kj_ircache_set_rtr_requests(KjIrcache* c, uint32_t enable) { KJ_REQUIRE(c, "null argument"); c->rtr_requests = enable != 0; return KJ_OK; }
KjStatus kj_ircache_begin_requests(KjIrcache* c, uint32_t rtdgi_half_width, uint32_t rtdgi_half_height, void* stream_) {
    return kj_ircache_begin_requests_rows(c, rtdgi_half_width, rtdgi_half_height, 0u, rtdgi_half_height, stream_);
}
static hipError_t irc_summary_buffers(KjIrcache* c, hipStream_t s) {
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k)
        if (c->summary[k].bytes != IRC_SUMMARY_BYTES) { e = c->summary[k].alloc(IRC_SUMMARY_BYTES, s); c->summary_fresh = true; }
    if (e == hipSuccess && c->alloc_min.bytes != size_t(IRC_MAX_GRID_CELLS) * 8) {
        e = c->alloc_min.alloc(size_t(IRC_MAX_GRID_CELLS) * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(c->alloc_min.p, 0xff, size_t(IRC_MAX_GRID_CELLS) * 8, s);      // every emit leaves it all-ones again
    }
    if (e == hipSuccess && c->req_scratch.bytes != (IRC_CELL_BLOCKS + 4u) * 4u + IRC_MAX_SUMMARIES * 8u) e = c->req_scratch.alloc((IRC_CELL_BLOCKS + 4u) * 4u + IRC_MAX_SUMMARIES * 8u, s);
    return e;
}
// what a summary needs cleared before records are reduced into it: header + winner + rank_min -> all-ones, votes -> 0 (the proposals are only read where a winner is)
static void irc_summary_clear_segments(void* summary, IrcClearSegs& sg) {
    const IrcSummaryView v = irc_summary_view(summary);
    sg.p[sg.n] = v.header; sg.words[sg.n] = (IRC_SUMMARY_HEADER_BYTES + IRC_MAX_ENTRIES * 12u) / 4u; sg.pattern[sg.n] = 0xffffffffu; sg.own[sg.n] = 0u; ++sg.n;
    sg.p[sg.n] = v.votes; sg.words[sg.n] = IRC_MAX_ENTRIES; sg.pattern[sg.n] = 0u; sg.own[sg.n] = 0u; ++sg.n;
}
// The same for a caller whose per-pixel passes (rtdgi's and rtr's validate / trace) run on half-res rows [half_row_begin, half_row_end) only -- a rank of the
// screen-tile split: only those rows of the per-pixel slot ranges are cleared (and the cache's own two ranges up to last frame's path count): at 4K the slot array
// is 150 MB, 280 MB with reflections. Slots of other rows stay as the (re)allocation left them: unused. ONE launch clears the cell words of every range and both
// summaries (round 5: 6-8 fills here, ~30 per rank and frame overall).
KjStatus kj_ircache_begin_requests_rows(KjIrcache* c, uint32_t rtdgi_half_width, uint32_t rtdgi_half_height, uint32_t half_row_begin, uint32_t half_row_end, void* stream_) {
    KJ_REQUIRE(c && c->deferred, "deferred updates are off (kj_ircache_set_deferred_updates)");
    KJ_REQUIRE(half_row_begin <= half_row_end && half_row_end <= rtdgi_half_height, "bad row range");
    hipStream_t s = (hipStream_t)stream_;
    c->req_half_pixels = rtdgi_half_width * rtdgi_half_height;
    const size_t RQ = sizeof(IrcRequest), bytes = size_t(c->request_slots()) * RQ, CB = 4, cell_bytes = size_t(c->request_slots()) * CB;
    const bool fresh = c->requests.bytes != bytes || c->req_clear_all;
    if (c->requests.bytes != bytes) { KJ_TRY_HIP(c->requests.alloc(bytes, s)); KJ_TRY_HIP(c->request_cells.alloc(cell_bytes, s)); }
    KJ_TRY_HIP(irc_summary_buffers(c, s));
    // what is cleared (and what a reduce scans) is the 4-byte cell copy of every slot, 0xffffffff = no record; the 32-byte records themselves are only ever read where a cell says so
    uint32_t* const base = (uint32_t*)c->request_cells.p;
    const size_t own_first = 2 * size_t(c->req_half_pixels);
    IrcClearSegs sg{};
    auto seg = [&](size_t first, size_t n, uint32_t own) { if (n) { sg.p[sg.n] = base + first; sg.words[sg.n] = uint32_t(n); sg.pattern[sg.n] = 0xffffffffu; sg.own[sg.n] = own; ++sg.n; } };
    if (fresh) seg(0, c->request_slots(), 0u);
    else {
        const size_t hb = c->req_half_pixels, row0 = size_t(half_row_begin) * rtdgi_half_width, n = size_t(half_row_end - half_row_begin) * rtdgi_half_width;
        seg(row0, n, 0u); seg(hb + row0, n, 0u);
        if (c->rtr_requests) { seg(c->rtr_request_base() + row0, n, 0u); seg(c->rtr_request_base() + hb + row0, n, 0u); }
        seg(own_first, KjIrcache::REQ_E, 1u); seg(own_first + KjIrcache::REQ_E, KjIrcache::REQ_E, 1u);      // the cache's own validate and trace rays: what last frame can have written
    }
    irc_summary_clear_segments(c->summary[0].p, sg); irc_summary_clear_segments(c->summary[1].p, sg);
    // the frame's other two clears ride along: the freed-entry flags of kj_ircache_prepare and the ray counters of kj_ircache_trace_irradiance
    c->begin_cleared_frame_state = c->freed.bytes == IRC_MAX_ENTRIES * 4 && sg.n + 2u <= IRC_CLEAR_MAX;
    if (c->begin_cleared_frame_state) {
        sg.p[sg.n] = (uint32_t*)c->freed.p; sg.words[sg.n] = IRC_MAX_ENTRIES; sg.pattern[sg.n] = 0u; sg.own[sg.n] = 0u; ++sg.n;
        sg.p[sg.n] = (uint32_t*)c->ray_counters.p; sg.words[sg.n] = KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE * 2u; sg.pattern[sg.n] = 0u; sg.own[sg.n] = 0u; ++sg.n;
    }
    hipLaunchKernelGGL(k_irc_clear_segments, dim3(c->dev->num_cus * 4), dim3(256), 0, s, sg, (const uint32_t*)c->meta.p);
    KJ_CHECK_LAUNCH();
    c->req_clear_all = false;
    c->requests_begun = true;
    return KJ_OK;
}
KjStatus kj_ircache_request_ranges(KjIrcache* c, uint32_t out_first_slot[4], uint32_t out_slot_count[4]) {
    KJ_REQUIRE(c && out_first_slot && out_slot_count, "null argument");
    const uint32_t hb = c->req_half_pixels, e = KjIrcache::REQ_E;
    const uint32_t first[4] = {0u, hb, 2u * hb, 2u * hb + e}, count[4] = {hb, hb, e, e};
    for (int k = 0; k < 4; ++k) { out_first_slot[k] = first[k]; out_slot_count[k] = count[k]; }
    return KJ_OK;
}
KjStatus kj_ircache_rtr_request_ranges(KjIrcache* c, uint32_t out_first_slot[2], uint32_t out_slot_count[2]) {
    KJ_REQUIRE(c && out_first_slot && out_slot_count, "null argument");
    const uint32_t hb = c->rtr_requests ? c->req_half_pixels : 0u;
    out_first_slot[0] = c->rtr_request_base(); out_first_slot[1] = c->rtr_request_base() + hb;
    out_slot_count[0] = out_slot_count[1] = hb;
    return KJ_OK;
}
uint64_t kj_ircache_summary_bytes(void) { return IRC_SUMMARY_BYTES; }
// records (rq / cells, ranges rr) -> summary: reduce, winners' proposals, the allocation list in cell order
static KjStatus irc_summarize(KjIrcache* c, const IrcRequest* rq, const uint32_t* cells, const IrcReduceRanges& rr, void* summary, hipStream_t s) {
    IrcacheView v = c->view();
    v.requests = nullptr;
    const IrcSummaryView sum = irc_summary_view(summary);
    uint64_t slots = 0;
    for (uint32_t k = 0; k < rr.n; ++k) slots += rr.count[k];
    const uint32_t per_block = 256u * IRC_RED_PER_THREAD;
    const uint32_t grid = uint32_t(std::min<uint64_t>(std::max<uint64_t>(1, (slots + per_block - 1) / per_block), uint64_t(c->dev->num_cus) * 8));
    unsigned long long* const alloc_min = (unsigned long long*)c->alloc_min.p;
    uint32_t* const block_counts = (uint32_t*)c->req_scratch.p;
    hipLaunchKernelGGL(k_irc_reduce_requests, dim3(grid), dim3(256), 0, s, v, rq, cells, rr, sum, alloc_min);
    hipLaunchKernelGGL(k_irc_reduce_winners, dim3(grid), dim3(256), 0, s, v, rq, cells, rr, sum);
    hipLaunchKernelGGL(k_irc_alloc_count, dim3(IRC_CELL_BLOCKS), dim3(256), 0, s, (const unsigned long long*)alloc_min, block_counts);
    hipLaunchKernelGGL(k_irc_alloc_emit<false>, dim3(IRC_CELL_BLOCKS), dim3(256), 0, s, v, alloc_min, (const uint32_t*)block_counts, rq, sum, (const void* const*)nullptr, (uint32_t*)nullptr);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
// Reduces the records of up to 6 slot ranges into the cache's summary `which` (0: what a rank of the split sends to the others -- its strip's per-pixel lookups;
// 1: what stays local -- the cache's own ray passes, identical on every replica). Once per summary and frame (kj_ircache_begin_requests clears both).
KjStatus kj_ircache_summarize_requests(KjIrcache* c, const uint32_t* first_slots, const uint32_t* slot_counts, uint32_t n_ranges, uint32_t which, void* stream_) {
    KJ_REQUIRE(c && c->deferred && c->requests.p && c->summary[0].p, "null argument / no requests recorded (kj_ircache_begin_requests)");
    KJ_REQUIRE(which < 2u && n_ranges <= IRC_REDUCE_MAX_RANGES && (n_ranges == 0 || (first_slots && slot_counts)), "bad range list / summary index");
    IrcReduceRanges rr{};
    const uint32_t own0 = 2u * c->req_half_pixels, own1 = own0 + KjIrcache::REQ_E;
    for (uint32_t k = 0; k < n_ranges; ++k) {
        KJ_REQUIRE(uint64_t(first_slots[k]) + slot_counts[k] <= c->request_slots(), "slot range out of bounds");
        // the cache's own two ranges (validation's and tracing's lookups, slot = path index) are scanned up to this frame's path count only
        rr.first[k] = first_slots[k]; rr.count[k] = slot_counts[k]; rr.own[k] = slot_counts[k] <= KjIrcache::REQ_E && (first_slots[k] == own0 || first_slots[k] == own1) ? 1u : 0u;
    }
    rr.n = n_ranges;
    return irc_summarize(c, (const IrcRequest*)c->requests.p, (const uint32_t*)c->request_cells.p, rr, c->summary[which].p, (hipStream_t)stream_);
}
KjStatus kj_ircache_summary(KjIrcache* c, uint32_t which, void** out_dev_ptr) {
    KJ_REQUIRE(c && which < 2u && out_dev_ptr && c->summary[which].p, "null argument / no summary yet (kj_ircache_begin_requests)");
    *out_dev_ptr = c->summary[which].p;
    return KJ_OK;
}
// The replay: merges `n` summaries (device pointers; every replica passes the same ones in the same order -- the ranks' strip summaries in rank order, then
// its own local one) and applies the result. No host synchronisation, nothing sized by a count the host would have to read.
KjStatus kj_ircache_apply_summaries(KjIrcache* c, const void* const* summaries, uint32_t n, void* stream_) {
    KJ_REQUIRE(c && summaries && n >= 1u && n <= IRC_MAX_SUMMARIES, "null argument / too many summaries (32)");
    hipStream_t s = (hipStream_t)stream_;
    KJ_TRY_HIP(irc_summary_buffers(c, s));
    IrcacheView v = c->view();
    v.requests = nullptr;
    IrcSummarySources src{};
    for (uint32_t k = 0; k < n; ++k) { KJ_REQUIRE(summaries[k], "null summary"); src.p[k] = summaries[k]; }
    src.n = n;
    unsigned long long* const alloc_min = (unsigned long long*)c->alloc_min.p;
    uint32_t* const block_counts = (uint32_t*)c->req_scratch.p; uint32_t* const total = block_counts + IRC_CELL_BLOCKS;
    const void** const table = (const void**)(total + 4);      // 8-byte aligned: req_scratch = 384 counts + 4 words + the table
    hipLaunchKernelGGL(k_irc_store_sources, dim3(1), dim3(32), 0, s, src, table);
    hipLaunchKernelGGL(k_irc_merge_entries, dim3(IRC_MAX_ENTRIES / 256), dim3(256), 0, s, v, (const void* const*)table, n);
    hipLaunchKernelGGL(k_irc_merge_allocs, dim3(IRC_MAX_ENTRIES / 256, n), dim3(256), 0, s, (const void* const*)table, alloc_min);
    hipLaunchKernelGGL(k_irc_alloc_count, dim3(IRC_CELL_BLOCKS), dim3(256), 0, s, (const unsigned long long*)alloc_min, block_counts);
    hipLaunchKernelGGL(k_irc_alloc_emit<true>, dim3(IRC_CELL_BLOCKS), dim3(256), 0, s, v, alloc_min, (const uint32_t*)block_counts, (const IrcRequest*)nullptr, IrcSummaryView{}, (const void* const*)table, total);
    hipLaunchKernelGGL(k_irc_request_finish, dim3(1), dim3(1), 0, s, v, (const uint32_t*)total);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
// A plain list of records (32 bytes each; cell 0xffffffff = unused) replayed at once: reduce into summary 0, apply it. For callers that assemble lists themselves
// and for tests; the frame path is begin_requests -> summarize_requests -> (exchange) -> apply_summaries.
KjStatus kj_ircache_apply_requests(KjIrcache* c, const void* list, uint32_t count, void* stream_) {
    KJ_REQUIRE(c && (list || count == 0), "null argument");
    if (count == 0) return KJ_OK;
    hipStream_t s = (hipStream_t)stream_;
    KJ_TRY_HIP(irc_summary_buffers(c, s));
    IrcClearSegs sg{};
    irc_summary_clear_segments(c->summary[0].p, sg);
    hipLaunchKernelGGL(k_irc_clear_segments, dim3(c->dev->num_cus), dim3(256), 0, s, sg, (const uint32_t*)c->meta.p);
    IrcReduceRanges rr{};
    rr.first[0] = 0u; rr.count[0] = count; rr.own[0] = 0u; rr.n = 1u;
    const KjStatus e = irc_summarize(c, (const IrcRequest*)list, nullptr, rr, c->summary[0].p, s);
    if (e != KJ_OK) return e;
    const void* one[1] = {c->summary[0].p};
    return kj_ircache_apply_summaries(c, one, 1u, s);
}
KjStatus kj_ircache_buffer(KjIrcache* c, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(c && name && out_dev_ptr && out_bytes, "null argument");
    struct { const char* n; kj::DevBuf* b; } tbl[] = {
        {"meta", &c->meta}, {"grid_meta", &c->grid_meta[c->cur]}, {"entry_cell", &c->entry_cell}, {"spatial", &c->spatial}, {"irradiance", &c->irradiance},
        {"aux", &c->aux}, {"life", &c->life}, {"pool", &c->pool}, {"entry_indirection", &c->entry_indirection},
        {"reposition_proposal", &c->reposition_proposal}, {"reposition_proposal_count", &c->reposition_proposal_count}, {"ray_counters", &c->ray_counters}};
    for (auto& t : tbl)
        if (strcmp(t.n, name) == 0) { *out_dev_ptr = t.b->p; *out_bytes = t.b->bytes; return KJ_OK; }
    set_last_error("no ircache buffer named '%s'", name);
    return KJ_ERR_INVALID_ARGUMENT;
}
KjStatus kj_ircache_ray_counts(KjIrcache* c, uint64_t* out_closest, uint64_t* out_any) {
    KJ_REQUIRE(c && out_closest && out_any, "null argument");
    uint64_t v[2];
    unsigned long long all_[KJ_COUNTER_SLOTS * KJ_COUNTER_STRIDE];
    KJ_TRY_HIP(hipMemcpy(all_, c->ray_counters.p, sizeof(all_), hipMemcpyDeviceToHost));
    v[0] = v[1] = 0;
    for (uint32_t sl = 0; sl < KJ_COUNTER_SLOTS; ++sl) { v[0] += all_[sl * KJ_COUNTER_STRIDE]; v[1] += all_[sl * KJ_COUNTER_STRIDE + 1]; }
    *out_closest = v[0]; *out_any = v[1];
    return KJ_OK;
}

}  // extern "C"
