// Irradiance cache — device-side lookup / lock-free allocation shared by the rtdgi and ircache ray
// kernels. Behaviour per assets/shaders/ircache/{lookup,ircache_grid,ircache_constants,
// ircache_sampler_common.inc}.hlsl. The cache is intentionally racy in the reference
// (buffers bound write_no_sync, docs/gi-overview.md:296); the same atomics are used here
// (device-scope atomicOr/Add/Max/Min on u32 in HBM).
#pragma once
#include "kj_shading.hpp"

namespace kj {

#define IRC_MAX_ENTRIES (1024u * 64u)
#define IRC_GRID_CELL_DIAMETER (0.16f * 0.125f)
#define IRC_CASCADE_SIZE 32u
#define IRC_CASCADE_COUNT 12u
#define IRC_MAX_GRID_CELLS (32u * 32u * 32u * 12u)
#define IRC_META_OCCUPIED 1u
#define IRC_META_JUST_ALLOCATED 2u
#define IRC_LIFE_RECYCLE 0x8000000u
#define IRC_LIFE_RECYCLED 0x8000001u
#define IRC_LIFE_PER_RANK 4u
#define IRC_RANK_COUNT 3u
#define IRC_OCTA_DIMS 4u
#define IRC_OCTA_DIMS2 16u
#define IRC_AUX_STRIDE 64u
#define IRC_SAMPLES_PER_FRAME 4u
#define IRC_VALIDATION_SAMPLES_PER_FRAME 4u
#define IRC_RESTIR_M_CLAMP 30u
#define IRC_META_TRACING_ALLOC_COUNT 0
#define IRC_META_ENTRY_COUNT 2
#define IRC_META_ALLOC_COUNT 3

struct IrcacheView {
    uint32_t* meta;                     // 8 x u32
    uint2* grid_meta;                   // current (post-scroll) cell table: (entry idx, flags)
    uint32_t* entry_cell;
    float4* spatial;                    // pos + 11:10:11 normal
    float4* irradiance;                 // 3 per entry (L1 SH per colour channel)
    float4* aux;                        // 64 per entry: 16 reservoirs, 16 (radiance, W), 16 origin vertices, 16 unused
    const float4* aux_read;             // what PRECISE lookups read: `aux` itself (the reference's racy read-while-written), or a snapshot taken
                                        // before the pass (deterministic mode: one of the racy program's legal outcomes, the same on every replica)
    uint32_t* life;
    uint32_t* pool;
    float4* reposition_proposal;
    uint32_t* reposition_proposal_count;
    const uint32_t* entry_indirection;
    // Deferred updates (screen-tile split across GPUs): when `requests` is set, a lookup changes nothing in the cache;
    // it records what it WOULD have done (allocate the cell, refresh the entry's life, vote for its position) in its own slot
    // requests[request_slot], and kj_ircache_apply_requests replays the merged records of all ranks in one canonical order.
    struct IrcRequest* requests;
    uint32_t* request_cells;      // requests[i].cell once more, 4 bytes per slot: what a frame's begin clears and its collects scan (0xffffffff = no record) instead of the 32-byte records
};
// One lookup's side effects. `cell` = 0xffffffff marks an unused slot. 32 bytes.
struct IrcRequest {
    uint32_t cell;          // grid cell the lookup resolved to
    uint32_t key;           // canonical position of the lookup in the frame: pass << 28 | pixel (or sample) index
    uint32_t bits;          // query_rank | skip_allocation << 8
    float dart;             // the lookup's random number for the position vote (lookup.hlsl:299-301)
    float4 proposal;        // packed IrcVertex it proposes for the entry
};

KJ_HD bool irc_life_valid(uint32_t life) { return life < IRC_LIFE_PER_RANK * IRC_RANK_COUNT; }

// pack_unpack.hlsl:84-96
KJ_HD V3 octa_decode(V2 f) {
    f = f * 2.0f - 1.0f;
    V3 n{f.x, f.y, 1.0f - fabsf(f.x) - fabsf(f.y)};
    const float t = clampf(-n.z, 0.0f, 1.0f);
    n.x -= (stepf(0.0f, n.x) * 2 - 1) * t;
    n.y -= (stepf(0.0f, n.y) * 2 - 1) * t;
    return normalize(n);
}
// ircache_sampler_common.inc.hlsl:6-57
KJ_HD uint32_t irc_sample_params(uint32_t spf, uint32_t entry_idx, uint32_t sample_idx, uint32_t frame_idx) {
    const uint32_t period = IRC_OCTA_DIMS2 / spf;
    uint32_t xy = sample_idx * period + (frame_idx % period);
    xy ^= (xy & 4u) >> 2u;
    return xy + ((frame_idx << 16u) ^ entry_idx) * IRC_OCTA_DIMS2;
}
KJ_HD V3 irc_sample_direction(uint32_t value) {
    const uint32_t oi = value % IRC_OCTA_DIMS2;
    const V2 urand = r2_sequence(hash1(value >> 4u) % 1024u);
    return octa_decode(V2{(float(oi % IRC_OCTA_DIMS) + urand.x) / 4.0f, (float(oi / IRC_OCTA_DIMS) + urand.y) / 4.0f});
}
struct IrcVertex { V3 position, normal; };
KJ_HD IrcVertex irc_unpack_vertex(float4 d) { return IrcVertex{V3{d.x, d.y, d.z}, unpack_unit_direction_11_10_11(asuint(d.w))}; }
KJ_HD float4 irc_pack_vertex(const IrcVertex& v) {
    float4 r; r.x = v.position.x; r.y = v.position.y; r.z = v.position.z; r.w = asfloat(pack_normal_11_10_11(v.normal));
    return r;
}

// ircache_grid.hlsl:34-80
KJ_HD uint32_t irc_cell_idx(uint32_t x, uint32_t y, uint32_t z, uint32_t cascade) {
    x = x < 31u ? x : 31u; y = y < 31u ? y : 31u; z = z < 31u ? z : 31u; cascade = cascade < 11u ? cascade : 11u;
    return x + y * 32u + z * 1024u + cascade * 32768u;
}
KJ_HD uint32_t irc_cascade_idx(V3 local_pos, uint32_t reserved_cells) {
    const V3 fcoord = local_pos / IRC_GRID_CELL_DIAMETER;
    const float max_coord = fmaxf(fabsf(fcoord.x), fmaxf(fabsf(fcoord.y), fabsf(fcoord.z)));
    const float cascade_float = log2f(max_coord / float(IRC_CASCADE_SIZE / 2u - reserved_cells));
    return uint32_t(clampf(ceilf(fmaxf(0.0f, cascade_float)), 0.0f, float(IRC_CASCADE_COUNT - 1u)));
}
struct IrcCoord { uint32_t x, y, z, cascade; };
template <bool JITTER = false>
KJ_HD IrcCoord irc_ws_pos_to_coord(const FrameConstants& fc, V3 pos, V3 normal, V3 jitter = V3{0.0f, 0.0f, 0.0f}) {
    const V3 center{fc.ircache_grid_center[0], fc.ircache_grid_center[1], fc.ircache_grid_center[2]};
    if (JITTER) {   // stochastic interpolation (ircache_grid.hlsl:51-56); compiled out for every caller but rtr's lookups
        const uint32_t c0 = irc_cascade_idx(pos - center, 1u);
        pos = pos + (IRC_GRID_CELL_DIAMETER * float(1u << c0)) * jitter;
    }
    const uint32_t cascade = irc_cascade_idx(pos - center, 1u);
    const float cell_diameter = IRC_GRID_CELL_DIAMETER * float(1u << cascade);
    const int32_t* org = fc.ircache_cascades[cascade].origin;
    const V3 q = (pos + normal * cell_diameter * 0.5f) / cell_diameter;
    const int cx = int(floorf(q.x)) - org[0], cy = int(floorf(q.y)) - org[1], cz = int(floorf(q.z)) - org[2];
    // clamp(coord, (0).xxx, (IRCACHE_CASCADE_SIZE - 1).xxx) with a uint constant (ircache_grid.hlsl:7,73): int and uint unify to uint -- a coordinate below the cascade's
    // first cell wraps and lands on its LAST cell
    const uint32_t ux = uint32_t(cx), uy = uint32_t(cy), uz = uint32_t(cz);
    return IrcCoord{ux < 31u ? ux : 31u, uy < 31u ? uy : 31u, uz < 31u ? uz : 31u, cascade};
}

#ifdef __HIPCC__
// The position a lookup proposes for its entry (lookup.hlsl:262-272): the hit point pulled towards the query by at most one cell.
// Every product and sum rounds separately: the same lookup is inlined into several kernels (the ray passes' fused / grouped / staged
// forms, rtr, the cache's own passes) and an fma contraction that differs between them would make the proposals -- and through them next
// frame's entry positions -- depend on which kernel ran.
KJ_D IrcVertex irc_reposition_proposal(V3 query_from_ws, V3 pt_ws, V3 normal_ws, float cell_diameter) {
#pragma clang fp contract(off)
    const V3 otq{query_from_ws.x - pt_ws.x, query_from_ws.y - pt_ws.y, query_from_ws.z - pt_ws.z};
    const float len = sqrtf(otq.x * otq.x + otq.y * otq.y + otq.z * otq.z);
    const float scale = cell_diameter / fmaxf(cell_diameter / 0.5f, len);
    return IrcVertex{V3{pt_ws.x + otq.x * scale, pt_ws.y + otq.y * scale, pt_ws.z + otq.z * scale}, normal_ws};
}
// lookup.hlsl:197-212
KJ_D float irc_eval_sh_geometrics(float4 sh, V3 normal) {
    const float R0 = sh.x;
    const V3 R1 = 0.5f * V3{sh.y, sh.z, sh.w};
    const float lenR1 = length(R1);
    const float q = 0.5f * (1.0f + dot(R1 / lenR1, normal));
    const float p = 1.0f + 2.0f * lenR1 / R0;
    const float a = (1.0f - lenR1 / R0) / (1.0f + lenR1 / R0);
    return R0 * (a + (1.0f - a) * (p + 1.0f) * powf(q, p));
}

// IrcacheLookupParams::lookup (lookup.hlsl:76-311). PRECISE = IRCACHE_LOOKUP_PRECISE.
// STOCHASTIC: the caller may ask for stochastic interpolation (rtr only); otherwise the jitter code is compiled out.
template <bool PRECISE, bool STOCHASTIC = false>
KJ_D V3 ircache_lookup(const IrcacheView& ic, const FrameConstants& fc, V3 query_from_ws, V3 pt_ws, V3 normal_ws, uint32_t query_rank, uint32_t& rng,
                       bool stochastic_interpolation = false, uint32_t request_slot = 0, uint32_t request_key = 0) {
    bool allocated_by_us = false, just_allocated = false;
    // select(stochastic_interpolation, float3(hash1_mut x3) - 0.5, 0): both arms are evaluated => rng advances 3x
    V3 jitter = v3(0.0f);
    if (STOCHASTIC) {
        jitter.x = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.y = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.z = uint_to_u01_float(hash1_mut(rng)) - 0.5f;
        if (!stochastic_interpolation) jitter = v3(0.0f);
    } else {
        hash1_mut(rng); hash1_mut(rng); hash1_mut(rng);
    }
    const IrcCoord rc = irc_ws_pos_to_coord<STOCHASTIC>(fc, pt_ws, normal_ws, jitter);
    const uint32_t cell = irc_cell_idx(rc.x, rc.y, rc.z, rc.cascade);
    {
        const int32_t* so = fc.ircache_cascades[rc.cascade].voxels_scrolled_this_frame;
        const int c[3] = {int(rc.x), int(rc.y), int(rc.z)};
        bool was_just_scrolled_in = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) was_just_scrolled_in |= so[k] > 0 ? (c[k] + so[k] >= int(IRC_CASCADE_SIZE)) : (c[k] < -so[k]);
        const bool skip_allocation = query_rank >= IRC_RANK_COUNT || (was_just_scrolled_in && query_rank > 0);
        const uint32_t entry_flags = ic.grid_meta[cell].y;
        just_allocated = (entry_flags & IRC_META_JUST_ALLOCATED) != 0;
        if (ic.requests) {
            // deferred: the value returned below does not depend on this frame's updates (an unoccupied cell yields 0 whether or not
            // someone allocates it now; an occupied one reads irradiance no lookup writes), so the updates can be replayed later
            IrcRequest rq;
            rq.cell = cell; rq.key = request_key; rq.bits = query_rank | (skip_allocation ? 0x100u : 0u);
            rq.dart = uint_to_u01_float(hash1_mut(rng));
            rq.proposal = irc_pack_vertex(irc_reposition_proposal(query_from_ws, pt_ws, normal_ws, IRC_GRID_CELL_DIAMETER * float(1u << rc.cascade)));
            ic.requests[request_slot] = rq;
            ic.request_cells[request_slot] = cell;
            if ((entry_flags & IRC_META_OCCUPIED) == 0 || just_allocated) return v3(0.0f);
        } else
        if (!skip_allocation && (entry_flags & IRC_META_OCCUPIED) == 0) {
            const uint32_t prev = atomicOr(&ic.grid_meta[cell].y, IRC_META_OCCUPIED | IRC_META_JUST_ALLOCATED);
            if ((prev & IRC_META_OCCUPIED) == 0) {
                just_allocated = true;
                allocated_by_us = true;
                const uint32_t alloc_idx = atomicAdd(&ic.meta[IRC_META_ALLOC_COUNT], 1u);
                if (alloc_idx >= IRC_MAX_ENTRIES) {
                    atomicAdd(&ic.meta[IRC_META_ALLOC_COUNT], 0xffffffffu);
                    atomicAnd(&ic.grid_meta[cell].y, ~(IRC_META_OCCUPIED | IRC_META_JUST_ALLOCATED));
                } else {
                    const uint32_t entry_idx = ic.pool[alloc_idx];
                    atomicMax(&ic.meta[IRC_META_ENTRY_COUNT], entry_idx + 1u);
                    ic.life[entry_idx] = query_rank * IRC_LIFE_PER_RANK;
                    ic.entry_cell[entry_idx] = cell;
                    ic.grid_meta[cell].x = entry_idx;
                }
            }
        }
    }
    const uint2 cell_meta = ic.grid_meta[cell];
    const bool found = (cell_meta.y & IRC_META_OCCUPIED) != 0;
    const uint32_t entry_idx = cell_meta.x;
    const float cell_diameter = IRC_GRID_CELL_DIAMETER * float(1u << rc.cascade);
    const IrcVertex proposal = irc_reposition_proposal(query_from_ws, pt_ws, normal_ws, cell_diameter);
    if (allocated_by_us && found) ic.reposition_proposal[entry_idx] = irc_pack_vertex(proposal);
    if (just_allocated) return v3(0.0f);
    V3 irradiance_sum = v3(0.0f);
    if (found) {
        V3 irr = v3(0.0f);
        if (PRECISE) {
            float weight_sum = 0;
            for (uint32_t octa_idx = 0; octa_idx < IRC_OCTA_DIMS2; ++octa_idx) {
                const uint32_t payload = asuint(ic.aux_read[size_t(entry_idx) * IRC_AUX_STRIDE + octa_idx].x);
                const float wt = dot(irc_sample_direction(payload), normal_ws);
                if (wt > 0.0f) {
                    const float4 contrib = ic.aux_read[size_t(entry_idx) * IRC_AUX_STRIDE + IRC_OCTA_DIMS2 + octa_idx];
                    irr += V3{contrib.x, contrib.y, contrib.z} * (wt * contrib.w);
                    weight_sum += wt;
                }
            }
            irr = irr / fmaxf(1.0f, weight_sum);
        } else {
            irr.x = irc_eval_sh_geometrics(ic.irradiance[entry_idx * 3u + 0u], normal_ws);
            irr.y = irc_eval_sh_geometrics(ic.irradiance[entry_idx * 3u + 1u], normal_ws);
            irr.z = irc_eval_sh_geometrics(ic.irradiance[entry_idx * 3u + 2u], normal_ws);
        }
        irradiance_sum = vmax(v3(0.0f), irr);
        const uint32_t prev_life = ic.life[entry_idx];
        if (!ic.requests && prev_life < IRC_LIFE_RECYCLE) {
            const uint32_t new_life = query_rank * IRC_LIFE_PER_RANK;
            if (new_life < prev_life) atomicMin(&ic.life[entry_idx], new_life);
            if (query_rank <= prev_life / IRC_LIFE_PER_RANK) {
                const uint32_t prev_vote_count = atomicAdd(&ic.reposition_proposal_count[entry_idx], 1u);
                const float dart = uint_to_u01_float(hash1_mut(rng));
                if (dart <= 1.0f / (float(prev_vote_count) + 1.0f)) ic.reposition_proposal[entry_idx] = irc_pack_vertex(proposal);
            }
        }
    }
    return irradiance_sum;
}
#endif

} // namespace kj
