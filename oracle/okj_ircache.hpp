// ORACLE (test infrastructure). Irradiance cache restated from
// crates/lib/kajiya/src/renderers/ircache.rs:26-506 (host) and assets/shaders/ircache/*.hlsl,
// prefix_scan/* (semantics: inclusive scan). Atomics use __atomic builtins so the passes can
// run under OpenMP like the GPU (order-dependent, as in the reference: docs/gi-overview.md:296);
// run single-threaded for a deterministic order.
// `deferred` = the deterministic mode: the statement of the SAME order-free semantics the product defines for its multi-GPU split
// (include/kajiya_amd.h: kj_ircache_set_deferred_updates) -- lookups return their value and only RECORD their side effects
// (lookup.hlsl:118-150,287-301), which are replayed after the frame's ray passes in (cell, position of the lookup in the frame)
// order; entries freed by scroll / age return to the pool in entry order; precise lookups inside the cache's own ray passes read a
// snapshot of `aux` taken before the pass. Every outcome is one the racy reference program can produce; none depends on thread
// interleaving, so GPU-vs-oracle comparisons of the cache can be held to the same bars as the deterministic passes.
#pragma once
#include <map>
#include <cstring>
#include "okj_scene.hpp"
#include "okj_reservoir.hpp"
#include <atomic>
#include <functional>
#include <algorithm>
#include <mutex>

namespace okj {

static const uint32_t IRCACHE_MAX_ENTRIES = 1024 * 64;             // ircache.rs:30
static const float IRCACHE_GRID_CELL_DIAMETER = 0.16f * 0.125f;    // ircache_grid.hlsl:5
static const uint32_t IRCACHE_CASCADE_SIZE = 32, IRCACHE_CASCADE_COUNT = 12;
static const uint32_t IRCACHE_MAX_GRID_CELLS = 32 * 32 * 32 * 12;
static const uint32_t IRCACHE_ENTRY_META_OCCUPIED = 1u, IRCACHE_ENTRY_META_JUST_ALLOCATED = 2u;
static const uint32_t IRCACHE_ENTRY_LIFE_RECYCLE = 0x8000000u, IRCACHE_ENTRY_LIFE_RECYCLED = 0x8000001u;
static const uint32_t IRCACHE_ENTRY_LIFE_PER_RANK = 4, IRCACHE_ENTRY_RANK_COUNT = 3;
static const uint32_t IRCACHE_OCTA_DIMS = 4, IRCACHE_OCTA_DIMS2 = 16, IRCACHE_IRRADIANCE_STRIDE = 3, IRCACHE_AUX_STRIDE = 64;
static const uint32_t IRCACHE_SAMPLES_PER_FRAME = 4, IRCACHE_VALIDATION_SAMPLES_PER_FRAME = 4, IRCACHE_RESTIR_M_CLAMP = 30;
enum { META_TRACING_ALLOC_COUNT = 0, META_ENTRY_COUNT = 2, META_ALLOC_COUNT = 3 };

static inline bool is_ircache_entry_life_valid(uint32_t life) { return life < IRCACHE_ENTRY_LIFE_PER_RANK * IRCACHE_ENTRY_RANK_COUNT; }
static inline uint32_t ircache_entry_life_to_rank(uint32_t life) { return life / IRCACHE_ENTRY_LIFE_PER_RANK; }
static inline uint32_t ircache_entry_life_for_rank(uint32_t rank) { return rank * IRCACHE_ENTRY_LIFE_PER_RANK; }

static inline uint32_t atomic_or(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomic_and(uint32_t* p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomic_add(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline void atomic_max(uint32_t* p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
static inline void atomic_min(uint32_t* p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }

// pack_unpack.hlsl:69-96
static inline f2 octa_wrap(f2 v) { return f2{(1.0f - fabsf(v.y)) * (step(0.0f, v.x) * 2.0f - 1.0f), (1.0f - fabsf(v.x)) * (step(0.0f, v.y) * 2.0f - 1.0f)}; }
static inline f3 octa_decode(f2 f) {
    f = f * 2.0f - 1.0f;
    f3 n{f.x, f.y, 1.0f - fabsf(f.x) - fabsf(f.y)};
    float t = clampf(-n.z, 0.0f, 1.0f);
    n.x -= (step(0.0f, n.x) * 2 - 1) * t;
    n.y -= (step(0.0f, n.y) * 2 - 1) * t;
    return normalize(n);
}

// ircache_sampler_common.inc.hlsl:6-57
struct SampleParams {
    uint32_t value;
    static SampleParams from_spf_entry_sample_frame(uint32_t spf, uint32_t entry_idx, uint32_t sample_idx, uint32_t frame_idx) {
        const uint32_t PERIOD = IRCACHE_OCTA_DIMS2 / spf;
        uint32_t xy = sample_idx * PERIOD + (frame_idx % PERIOD);
        xy ^= (xy & 4u) >> 2u;
        return SampleParams{xy + ((frame_idx << 16u) ^ (entry_idx)) * IRCACHE_OCTA_DIMS2};
    }
    uint32_t octa_idx() const { return value % IRCACHE_OCTA_DIMS2; }
    uint32_t rng() const { return hash1(value >> 4u); }
    f2 octa_uv() const {
        const uint32_t oi = octa_idx();
        const f2 urand = r2_sequence(rng() % 1024u);
        return f2{(float(oi % IRCACHE_OCTA_DIMS) + urand.x) / 4.0f, (float(oi / IRCACHE_OCTA_DIMS) + urand.y) / 4.0f};
    }
    f3 direction() const { return octa_decode(octa_uv()); }
};

struct IrcacheVertex { f3 position, normal; };
static inline IrcacheVertex unpack_vertex(f4 d) { return IrcacheVertex{xyz(d), unpack_unit_direction_11_10_11(asuint(d.w))}; }
static inline f4 pack_vertex(const IrcacheVertex& v) { return mk4(v.position, asfloat(pack_normal_11_10_11(v.normal))); }

struct Ircache {
    // ---- persistent buffers (ircache.rs:172-232)
    std::vector<uint32_t> meta, entry_cell, life, pool, entry_indirection, reposition_proposal_count, entry_occupancy;
    std::vector<u2> grid_meta[2];
    std::vector<f4> spatial, irradiance, aux, reposition_proposal;
    // ---- IrcacheRenderer host state (ircache.rs:92-100)
    bool initialized = false;
    f3 grid_center{0, 0, 0};
    int cur_scroll[12][3] = {}, prev_scroll[12][3] = {};
    int parity = 0;
    bool enable_scroll = true;
    int cur = 0;  // index of the live grid_meta buffer after prepare()
    std::atomic<uint64_t> rays_closest{0}, rays_any{0};
    // ---- deterministic mode (see the header comment)
    struct Request { uint32_t cell, key, bits; float dart; f4 proposal; };   // bits = query_rank | skip_allocation << 8
    bool deferred = false;
    std::vector<Request> requests;
    std::mutex requests_mutex;
    std::vector<f4> aux_snapshot;
    std::vector<uint32_t> freed;
    bool read_aux_snapshot = false;                                            // precise lookups read aux_snapshot instead of aux
    static uint32_t& request_key() { static thread_local uint32_t k = 0; return k; }   // set by the caller right before lookup()

    Ircache() {
        meta.assign(8, 0);
        grid_meta[0].assign(IRCACHE_MAX_GRID_CELLS, u2{0, 0});
        grid_meta[1].assign(IRCACHE_MAX_GRID_CELLS, u2{0, 0});
        entry_cell.assign(IRCACHE_MAX_ENTRIES, 0);
        spatial.assign(IRCACHE_MAX_ENTRIES, f4{0, 0, 0, 0});
        irradiance.assign(3 * IRCACHE_MAX_ENTRIES, f4{0, 0, 0, 0});
        aux.assign(size_t(64) * IRCACHE_MAX_ENTRIES, f4{0, 0, 0, 0});
        life.assign(IRCACHE_MAX_ENTRIES, 0);
        pool.assign(IRCACHE_MAX_ENTRIES, 0);
        entry_indirection.assign(IRCACHE_MAX_ENTRIES + 64, 0);
        reposition_proposal.assign(IRCACHE_MAX_ENTRIES, f4{0, 0, 0, 0});
        reposition_proposal_count.assign(IRCACHE_MAX_ENTRIES, 0);
        entry_occupancy.assign(IRCACHE_MAX_ENTRIES + 64, 0);
    }

    // ircache.rs:126-141
    void update_eye_position(f3 eye) {
        if (!enable_scroll) return;
        grid_center = eye;
        for (int c = 0; c < 12; ++c) {
            const float cell_diameter = IRCACHE_GRID_CELL_DIAMETER * float(1 << c);
            const float e[3] = {eye.x, eye.y, eye.z};
            for (int k = 0; k < 3; ++k) {
                prev_scroll[c][k] = cur_scroll[c][k];
                cur_scroll[c][k] = int(floorf(e[k] / cell_diameter)) - int(IRCACHE_CASCADE_SIZE) / 2;
            }
        }
    }
    // ircache.rs:143-158
    void constants(KjFrameConstants& fc) const {
        fc.ircache_grid_center[0] = grid_center.x; fc.ircache_grid_center[1] = grid_center.y; fc.ircache_grid_center[2] = grid_center.z; fc.ircache_grid_center[3] = 1.0f;
        for (int c = 0; c < 12; ++c)
            for (int k = 0; k < 4; ++k) {
                fc.ircache_cascades[c].origin[k] = k < 3 ? cur_scroll[c][k] : 0;
                fc.ircache_cascades[c].voxels_scrolled_this_frame[k] = k < 3 ? cur_scroll[c][k] - prev_scroll[c][k] : 0;
            }
    }

    u2* gm() { return grid_meta[cur].data(); }

    // ---- ircache_grid.hlsl:14-80
    static uint32_t cell_idx(uint32_t x, uint32_t y, uint32_t z, uint32_t cascade) {
        x = std::min(x, 31u); y = std::min(y, 31u); z = std::min(z, 31u); cascade = std::min(cascade, 11u);
        return x + y * 32 + z * 32 * 32 + cascade * 32 * 32 * 32;
    }
    static uint32_t ws_local_pos_to_cascade_idx(f3 local_pos, uint32_t reserved_cells) {
        const f3 fcoord = local_pos / IRCACHE_GRID_CELL_DIAMETER;
        const float max_coord = fmaxf(fabsf(fcoord.x), fmaxf(fabsf(fcoord.y), fabsf(fcoord.z)));
        const float cascade_float = log2f(max_coord / float(IRCACHE_CASCADE_SIZE / 2 - reserved_cells));
        return uint32_t(clampf(ceilf(fmaxf(0.0f, cascade_float)), 0.0f, float(IRCACHE_CASCADE_COUNT - 1)));
    }
    struct Coord { uint32_t x, y, z, cascade; uint32_t cell() const { return cell_idx(x, y, z, cascade); } };
    static Coord ws_pos_to_ircache_coord(const FrameConstants& fc, f3 pos, f3 normal, f3 jitter) {
        const f3 center{fc.ircache_grid_center[0], fc.ircache_grid_center[1], fc.ircache_grid_center[2]};
        const uint32_t reserved_cells = 1;
        {
            const uint32_t cascade = ws_local_pos_to_cascade_idx(pos - center, reserved_cells);
            const float cell_diameter = IRCACHE_GRID_CELL_DIAMETER * float(1u << cascade);
            pos = pos + cell_diameter * jitter;
        }
        const uint32_t cascade = ws_local_pos_to_cascade_idx(pos - center, reserved_cells);
        const float cell_diameter = IRCACHE_GRID_CELL_DIAMETER * float(1u << cascade);
        const int32_t* org = fc.ircache_cascades[cascade].origin;
        const f3 cell_offset = normal * cell_diameter * 0.5f;
        const f3 q = (pos + cell_offset) / cell_diameter;
        const int cx = int(floorf(q.x)) - org[0], cy = int(floorf(q.y)) - org[1], cz = int(floorf(q.z)) - org[2];
        // clamp(coord, (0).xxx, (IRCACHE_CASCADE_SIZE - 1).xxx) with IRCACHE_CASCADE_SIZE a uint (ircache_grid.hlsl:7,73): int and uint unify to uint, so a coordinate
        // below the cascade's first cell wraps and lands on its LAST cell, not on cell 0
        auto cl = [](int v) { return std::min(uint32_t(v), 31u); };
        return Coord{cl(cx), cl(cy), cl(cz), cascade};
    }

    // ---- lookup.hlsl:18-313
    static float eval_sh_geometrics(f4 sh, f3 normal) {
        const float R0 = sh.x;
        const f3 R1 = 0.5f * f3{sh.y, sh.z, sh.w};
        const float lenR1 = length(R1);
        const float q = 0.5f * (1.0f + dot(R1 / lenR1, normal));
        const float p = 1.0f + 2.0f * lenR1 / R0;
        const float a = (1.0f - lenR1 / R0) / (1.0f + lenR1 / R0);
        return R0 * (a + (1.0f - a) * (p + 1.0f) * powf(q, p));
    }
    struct LookupMaybeAllocate { bool found; uint32_t entry_idx; IrcacheVertex proposal; bool just_allocated; };
    LookupMaybeAllocate lookup_maybe_allocate(const FrameConstants& fc, f3 query_from_ws, f3 pt_ws, f3 normal_ws, uint32_t query_rank, uint32_t& rng, bool stochastic_interpolation = false) {
        bool allocated_by_us = false, just_allocated = false;
        // `select(stochastic_interpolation, float3(hash1_mut(rng)...) - 0.5, 0)` (lookup.hlsl:87-93): select() is a
        // function, so its arguments are evaluated and the rng advances three times even though
        // stochastic interpolation is never enabled on this path; the jitter itself is zero.
        // stochastic interpolation is never enabled on the rtdgi path; rtr enables it for wide ray cones (reflection_trace_common.inc.hlsl:212-220).
        f3 jitter;
        jitter.x = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.y = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.z = uint_to_u01_float(hash1_mut(rng)) - 0.5f;
        if (!stochastic_interpolation) jitter = mk3(0.0f);
        {
            const Coord rc = ws_pos_to_ircache_coord(fc, pt_ws, normal_ws, jitter);
            const int32_t* so = fc.ircache_cascades[rc.cascade].voxels_scrolled_this_frame;
            const int c[3] = {int(rc.x), int(rc.y), int(rc.z)};
            bool was_just_scrolled_in = false;
            for (int k = 0; k < 3; ++k) was_just_scrolled_in |= so[k] > 0 ? (c[k] + so[k] >= int(IRCACHE_CASCADE_SIZE)) : (c[k] < -so[k]);
            const bool skip_allocation = query_rank >= IRCACHE_ENTRY_RANK_COUNT || (was_just_scrolled_in && query_rank > 0);
            const uint32_t cell = rc.cell();
            const uint32_t entry_flags = __atomic_load_n(&gm()[cell].y, __ATOMIC_RELAXED);
            just_allocated = (entry_flags & IRCACHE_ENTRY_META_JUST_ALLOCATED) != 0;
            if (!skip_allocation && (entry_flags & IRCACHE_ENTRY_META_OCCUPIED) == 0) {
                const uint32_t prev = atomic_or(&gm()[cell].y, IRCACHE_ENTRY_META_OCCUPIED | IRCACHE_ENTRY_META_JUST_ALLOCATED);
                if ((prev & IRCACHE_ENTRY_META_OCCUPIED) == 0) {
                    just_allocated = true;
                    allocated_by_us = true;
                    const uint32_t alloc_idx = atomic_add(&meta[META_ALLOC_COUNT], 1);
                    if (alloc_idx >= 1024 * 64) {
                        atomic_add(&meta[META_ALLOC_COUNT], uint32_t(-1));
                        atomic_and(&gm()[cell].y, ~(IRCACHE_ENTRY_META_OCCUPIED | IRCACHE_ENTRY_META_JUST_ALLOCATED));
                    } else {
                        const uint32_t entry_idx = pool[alloc_idx];
                        atomic_max(&meta[META_ENTRY_COUNT], entry_idx + 1);
                        life[entry_idx] = ircache_entry_life_for_rank(query_rank);
                        entry_cell[entry_idx] = cell;
                        __atomic_store_n(&gm()[cell].x, entry_idx, __ATOMIC_RELAXED);
                    }
                }
            }
        }
        LookupMaybeAllocate res;
        const Coord rc = ws_pos_to_ircache_coord(fc, pt_ws, normal_ws, jitter);
        const u2 cell_meta{__atomic_load_n(&gm()[rc.cell()].x, __ATOMIC_RELAXED), __atomic_load_n(&gm()[rc.cell()].y, __ATOMIC_RELAXED)};
        res.found = (cell_meta.y & IRCACHE_ENTRY_META_OCCUPIED) != 0;
        res.entry_idx = cell_meta.x;
        const float cell_diameter = IRCACHE_GRID_CELL_DIAMETER * float(1u << rc.cascade);
        f3 offset_towards_query = query_from_ws - pt_ws;
        const float MAX_OFFSET = cell_diameter, MAX_OFFSET_AS_FRAC = 0.5f;
        offset_towards_query = offset_towards_query * (MAX_OFFSET / fmaxf(MAX_OFFSET / MAX_OFFSET_AS_FRAC, length(offset_towards_query)));
        res.proposal = IrcacheVertex{pt_ws + offset_towards_query, normal_ws};
        if (allocated_by_us && res.found) reposition_proposal[res.entry_idx] = pack_vertex(res.proposal);
        res.just_allocated = just_allocated;
        return res;
    }
    // `precise` = IRCACHE_LOOKUP_PRECISE (defined by the ircache's own trace/validate shaders)
    f3 lookup(const FrameConstants& fc, f3 query_from_ws, f3 pt_ws, f3 normal_ws, uint32_t query_rank, uint32_t& rng, bool precise, bool stochastic_interpolation = false) {
        if (deferred) return lookup_deferred(fc, query_from_ws, pt_ws, normal_ws, query_rank, rng, precise, stochastic_interpolation);
        const LookupMaybeAllocate lk = lookup_maybe_allocate(fc, query_from_ws, pt_ws, normal_ws, query_rank, rng, stochastic_interpolation);
        if (lk.just_allocated) return mk3(0.0f);
        f3 irradiance_sum = mk3(0.0f);
        if (lk.found) {
            const uint32_t entry_idx = lk.entry_idx;
            const f3 irr = entry_irradiance(entry_idx, normal_ws, precise);
            irradiance_sum += irr;
            const uint32_t prev_life = __atomic_load_n(&life[entry_idx], __ATOMIC_RELAXED);
            if (prev_life < IRCACHE_ENTRY_LIFE_RECYCLE) {
                const uint32_t new_life = ircache_entry_life_for_rank(query_rank);
                if (new_life < prev_life) atomic_min(&life[entry_idx], new_life);
                const uint32_t prev_rank = ircache_entry_life_to_rank(prev_life);
                if (query_rank <= prev_rank) {
                    const uint32_t prev_vote_count = atomic_add(&reposition_proposal_count[entry_idx], 1);
                    const float dart = uint_to_u01_float(hash1_mut(rng));
                    const float prob = 1.0f / (float(prev_vote_count) + 1.0f);
                    if (dart <= prob) reposition_proposal[entry_idx] = pack_vertex(lk.proposal);
                }
            }
        }
        return irradiance_sum;
    }

    // the value half of lookup() (lookup.hlsl:239-285), no side effects
    f3 entry_irradiance(uint32_t entry_idx, f3 normal_ws, bool precise) const {
        f3 irr = mk3(0.0f);
        if (precise) {
            const f4* a = read_aux_snapshot ? aux_snapshot.data() : aux.data();
            float weight_sum = 0;
            for (uint32_t octa_idx = 0; octa_idx < IRCACHE_OCTA_DIMS2; ++octa_idx) {
                const f4 r0 = a[size_t(entry_idx) * IRCACHE_AUX_STRIDE + octa_idx];
                const f3 dir = SampleParams{asuint(r0.x)}.direction();
                const float wt = dot(dir, normal_ws);
                if (wt > 0.0f) {
                    const f4 contrib = a[size_t(entry_idx) * IRCACHE_AUX_STRIDE + IRCACHE_OCTA_DIMS2 + octa_idx];
                    irr += xyz(contrib) * (wt * contrib.w);
                    weight_sum += wt;
                }
            }
            irr = irr / fmaxf(1.0f, weight_sum);
        } else {
            irr.x = eval_sh_geometrics(irradiance[entry_idx * 3 + 0], normal_ws);
            irr.y = eval_sh_geometrics(irradiance[entry_idx * 3 + 1], normal_ws);
            irr.z = eval_sh_geometrics(irradiance[entry_idx * 3 + 2], normal_ws);
        }
        return vmax(mk3(0.0f), irr);
    }
    // Deterministic mode: the lookup's value does not depend on this frame's updates (an unoccupied cell yields 0 whether or not someone
    // allocates it now; an occupied one reads irradiance no lookup writes), so the updates are recorded and replayed by apply_requests().
    f3 lookup_deferred(const FrameConstants& fc, f3 query_from_ws, f3 pt_ws, f3 normal_ws, uint32_t query_rank, uint32_t& rng, bool precise, bool stochastic_interpolation) {
        f3 jitter;
        jitter.x = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.y = uint_to_u01_float(hash1_mut(rng)) - 0.5f; jitter.z = uint_to_u01_float(hash1_mut(rng)) - 0.5f;
        if (!stochastic_interpolation) jitter = mk3(0.0f);
        const Coord rc = ws_pos_to_ircache_coord(fc, pt_ws, normal_ws, jitter);
        const int32_t* so = fc.ircache_cascades[rc.cascade].voxels_scrolled_this_frame;
        const int c[3] = {int(rc.x), int(rc.y), int(rc.z)};
        bool was_just_scrolled_in = false;
        for (int k = 0; k < 3; ++k) was_just_scrolled_in |= so[k] > 0 ? (c[k] + so[k] >= int(IRCACHE_CASCADE_SIZE)) : (c[k] < -so[k]);
        const bool skip_allocation = query_rank >= IRCACHE_ENTRY_RANK_COUNT || (was_just_scrolled_in && query_rank > 0);
        const uint32_t cell = rc.cell();
        const u2 cell_meta = gm()[cell];
        Request rq;
        rq.cell = cell; rq.key = request_key(); rq.bits = query_rank | (skip_allocation ? 0x100u : 0u);
        rq.dart = uint_to_u01_float(hash1_mut(rng));
        const float cell_diameter = IRCACHE_GRID_CELL_DIAMETER * float(1u << rc.cascade);
        f3 offset_towards_query = query_from_ws - pt_ws;
        offset_towards_query = offset_towards_query * (cell_diameter / fmaxf(cell_diameter / 0.5f, length(offset_towards_query)));
        rq.proposal = pack_vertex(IrcacheVertex{pt_ws + offset_towards_query, normal_ws});
        { std::lock_guard<std::mutex> g(requests_mutex); requests.push_back(rq); }
        if ((cell_meta.y & IRCACHE_ENTRY_META_OCCUPIED) == 0 || (cell_meta.y & IRCACHE_ENTRY_META_JUST_ALLOCATED) != 0) return mk3(0.0f);
        return entry_irradiance(cell_meta.x, normal_ws, precise);
    }
    void begin_requests() { requests.clear(); }
    void snapshot_aux() {   // the half of every entry's block that lookups read, as it is before a pass
        if (aux_snapshot.size() != aux.size()) aux_snapshot.assign(aux.size(), f4{0, 0, 0, 0});
        const size_t n = size_t(meta[META_ENTRY_COUNT]) * 32;
        for (size_t i = 0; i < n; ++i) { const size_t o = (i >> 5) * IRCACHE_AUX_STRIDE + (i & 31u); aux_snapshot[o] = aux[o]; }
    }
    // Replay of the frame's recorded lookups: a reduction over the records of a cell, so that a rank of the screen-tile split can reduce its own strip first and
    // the result does not depend on the number of ranks. One of the racy program's legal outcomes (lookup.hlsl:118-150, 287-301):
    //   * a cell nobody occupies is allocated by the lookup with the lowest position in the frame among those that may allocate ("the thread whose
    //     InterlockedOr came first"); new cells take pool entries in cell order;
    //   * an occupied cell's lookups all read the entry's life before any of them lowers it (every load of :287 before every InterlockedMin of :291):
    //     lookup i votes iff rank_i <= life0 / LIFE_PER_RANK; the life ends as min(life0, min_i rank_i * LIFE_PER_RANK);
    //   * the position vote: the voter with the smallest dart (ties: lowest key) is the one whose InterlockedAdd returned the count before the frame's
    //     votes -- accepted iff dart <= 1 / (v0 + 1) -- and its store lands last; if that dart fails no voter can be accepted.
    void apply_requests() {
        std::vector<Request>& rq = requests;
        const size_t n = rq.size();
        // per cell: the allocation winner; per entry: rank minimum, voters, vote winner
        std::map<uint32_t, size_t> alloc_winner;
        struct Acc { uint32_t rank_min = 0xffffffffu, votes = 0; uint64_t win = ~0ull; size_t win_idx = 0; };
        std::map<uint32_t, Acc> acc;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t cell = rq[i].cell;
            if (cell == 0xffffffffu) continue;
            const u2 m = gm()[cell];
            if ((m.y & IRCACHE_ENTRY_META_OCCUPIED) == 0) {
                if (rq[i].bits & 0x100u) continue;
                auto it = alloc_winner.find(cell);
                if (it == alloc_winner.end() || rq[i].key < rq[it->second].key) alloc_winner[cell] = i;
            } else if ((m.y & IRCACHE_ENTRY_META_JUST_ALLOCATED) == 0) {
                const uint32_t entry_idx = m.x, life0 = life[entry_idx];
                if (!(life0 < IRCACHE_ENTRY_LIFE_RECYCLE)) continue;
                const uint32_t query_rank = rq[i].bits & 0xffu;
                Acc& a = acc[entry_idx];
                a.rank_min = std::min(a.rank_min, query_rank);
                if (query_rank <= ircache_entry_life_to_rank(life0)) {
                    uint32_t dart_bits; memcpy(&dart_bits, &rq[i].dart, 4);
                    const uint64_t w = (uint64_t(dart_bits) << 32) | rq[i].key;
                    ++a.votes;
                    if (w < a.win) { a.win = w; a.win_idx = i; }
                }
            }
        }
        for (const auto& kv : acc) {
            const uint32_t entry_idx = kv.first;
            const Acc& a = kv.second;
            life[entry_idx] = std::min(life[entry_idx], ircache_entry_life_for_rank(a.rank_min));
            if (a.votes) {
                const uint32_t v0 = reposition_proposal_count[entry_idx];
                reposition_proposal_count[entry_idx] = v0 + a.votes;
                if (rq[a.win_idx].dart <= 1.0f / (float(v0) + 1.0f)) reposition_proposal[entry_idx] = rq[a.win_idx].proposal;
            }
        }
        const uint32_t alloc_before = meta[META_ALLOC_COUNT];
        uint32_t allocated = 0;
        for (const auto& kv : alloc_winner) {          // std::map: ascending cell order
            const uint32_t cell = kv.first;
            const Request& w = rq[kv.second];
            const uint32_t alloc_idx = alloc_before + allocated++;
            if (alloc_idx >= IRCACHE_MAX_ENTRIES) continue;                    // pool exhausted, the cell stays empty
            const uint32_t entry_idx = pool[alloc_idx];
            meta[META_ENTRY_COUNT] = std::max(meta[META_ENTRY_COUNT], entry_idx + 1);
            life[entry_idx] = ircache_entry_life_for_rank(w.bits & 0xffu);
            entry_cell[entry_idx] = cell;
            gm()[cell] = u2{entry_idx, gm()[cell].y | IRCACHE_ENTRY_META_OCCUPIED | IRCACHE_ENTRY_META_JUST_ALLOCATED};
            reposition_proposal[entry_idx] = w.proposal;
        }
        meta[META_ALLOC_COUNT] = std::min(alloc_before + allocated, IRCACHE_MAX_ENTRIES);
        requests.clear();
    }

    // ---- prepare (ircache.rs:168-350)
    void prepare(const FrameConstants& fc) {
        int a = 0, b = 1;                      // grid_meta_buf, grid_meta_buf2
        if (parity == 1) std::swap(a, b);
        if (deferred) freed.assign(IRCACHE_MAX_ENTRIES, 0);
        if (!initialized) {
            for (uint32_t i = 0; i < IRCACHE_MAX_ENTRIES; ++i) { pool[i] = i; life[i] = IRCACHE_ENTRY_LIFE_RECYCLED; }  // clear_ircache_pool.hlsl
            initialized = true;
        } else {
            scroll_cascades(fc, grid_meta[a].data(), grid_meta[b].data());
            std::swap(a, b);
            parity = (parity + 1) % 2;
        }
        cur = a;
        const uint32_t entry_count = meta[META_ENTRY_COUNT];
        const uint32_t groups = (entry_count + 63) / 64;                     // prepare_age_dispatch_args.hlsl
        age_entries(groups * 64);
        if (deferred) {   // entries freed by the two passes above go back to the pool in ascending entry order
            uint32_t total = 0;
            for (uint32_t e = 0; e < IRCACHE_MAX_ENTRIES; ++e) total += freed[e];
            const uint32_t new_count = meta[META_ALLOC_COUNT] - total;
            uint32_t before = 0;
            for (uint32_t e = 0; e < IRCACHE_MAX_ENTRIES; ++e) if (freed[e]) pool[new_count + before++] = e;
            meta[META_ALLOC_COUNT] = new_count;
        }
        // inclusive prefix scan of the occupancy flags (prefix_scan/*.hlsl)
        uint32_t run = 0;
        for (uint32_t i = 0; i < IRCACHE_MAX_ENTRIES; ++i) { run += entry_occupancy[i]; entry_occupancy[i] = run; }
        // ircache_compact_entries.hlsl
        for (uint32_t e = 0; e < groups * 64 && e < IRCACHE_MAX_ENTRIES; ++e) {
            if (e < entry_count && is_ircache_entry_life_valid(life[e])) entry_indirection[entry_occupancy[e]] = e;
        }
    }
    // scroll_cascades.hlsl:13-69
    void scroll_cascades(const FrameConstants& fc, const u2* src, u2* dst) {
        for (uint32_t cascade = 0; cascade < 12; ++cascade) {
            const int32_t* sb = fc.ircache_cascades[cascade].voxels_scrolled_this_frame;
            for (uint32_t z = 0; z < 32; ++z)
                for (uint32_t y = 0; y < 32; ++y)
                    for (uint32_t x = 0; x < 32; ++x) {
                        const uint32_t dst_cell = cell_idx(x, y, z, cascade);
                        const uint32_t bx = uint32_t(int(x) - sb[0]), by = uint32_t(int(y) - sb[1]), bz = uint32_t(int(z) - sb[2]);
                        if (!(bx < 32 && by < 32 && bz < 32)) {
                            const u2 m = src[dst_cell];                       // deallocate_cell
                            if (m.y & IRCACHE_ENTRY_META_OCCUPIED) {
                                const uint32_t entry_idx = m.x;
                                life[entry_idx] = IRCACHE_ENTRY_LIFE_RECYCLED;
                                for (int i = 0; i < 3; ++i) irradiance[entry_idx * 3 + i] = f4{0, 0, 0, 0};
                                if (deferred) freed[entry_idx] = 1;
                                else { const uint32_t c = atomic_add(&meta[META_ALLOC_COUNT], uint32_t(-1)); pool[c - 1] = entry_idx; }
                            }
                        }
                        const uint32_t sx = uint32_t(int(x) + sb[0]), sy = uint32_t(int(y) + sb[1]), sz = uint32_t(int(z) + sb[2]);
                        if (sx < 32 && sy < 32 && sz < 32) {
                            const u2 cm = src[cell_idx(sx, sy, sz, cascade)];
                            dst[dst_cell] = cm;
                            if (cm.y & IRCACHE_ENTRY_META_OCCUPIED) entry_cell[cm.x] = dst_cell;
                        } else {
                            dst[dst_cell] = u2{0, 0};
                        }
                    }
        }
    }
    // age_ircache_entries.hlsl:22-94
    void age_entries(uint32_t thread_count) {
        const uint32_t total_entry_count = meta[META_ENTRY_COUNT];
        for (uint32_t e = 0; e < thread_count && e < IRCACHE_MAX_ENTRIES; ++e) {
            if (e < total_entry_count) {
                const uint32_t l = life[e];
                if (l != IRCACHE_ENTRY_LIFE_RECYCLED) {
                    const uint32_t new_age = l + 1;
                    if (is_ircache_entry_life_valid(new_age)) {
                        life[e] = new_age;
                        atomic_and(&gm()[entry_cell[e]].y, ~IRCACHE_ENTRY_META_JUST_ALLOCATED);
                    } else {
                        life[e] = IRCACHE_ENTRY_LIFE_RECYCLED;
                        for (int i = 0; i < 3; ++i) irradiance[e * 3 + i] = f4{0, 0, 0, 0};
                        if (deferred) freed[e] = 1;
                        else { const uint32_t c = atomic_add(&meta[META_ALLOC_COUNT], uint32_t(-1)); pool[c - 1] = e; }
                        atomic_and(&gm()[entry_cell[e]].y, ~(IRCACHE_ENTRY_META_OCCUPIED | IRCACHE_ENTRY_META_JUST_ALLOCATED));
                    }
                }
                spatial[e] = reposition_proposal[e];   // flush the reposition proposal
                reposition_proposal_count[e] = 0;
            } else {
                spatial[e] = f4{0, 0, 0, 0};
            }
            entry_occupancy[e] = (e < total_entry_count && is_ircache_entry_life_valid(life[e])) ? 1u : 0u;
        }
        // the occupancy buffer is a transient in the reference: entries beyond the dispatched range are
        // undefined there and never consumed (the scan is a prefix); define them as 0.
        for (uint32_t e = thread_count; e < IRCACHE_MAX_ENTRIES; ++e) entry_occupancy[e] = 0;
    }
};

} // namespace okj
