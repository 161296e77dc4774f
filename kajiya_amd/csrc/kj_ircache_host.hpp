// Host-side state of the irradiance cache handle (IrcacheRenderer + IrcacheRenderState, renderers/ircache.rs:38-100).
#pragma once
#include "kj_host.hpp"
#include "kj_ircache.hpp"

struct KjIrcache {
    KjDevice* dev = nullptr;
    // persistent buffers (ircache.rs:172-232)
    kj::DevBuf meta, grid_meta[2], entry_cell, spatial, irradiance, aux, life, pool, entry_indirection, reposition_proposal,
        reposition_proposal_count, occupancy, ray_counters;
    // IrcacheRenderer host state
    bool initialized = false, enable_scroll = true, pending_irradiance_sum = false;
    float grid_center[3] = {0, 0, 0};
    int cur_scroll[12][3] = {}, prev_scroll[12][3] = {};
    int parity = 0;
    int cur = 0;  // which grid_meta buffer is live after prepare()
    // deferred updates (kj_ircache.hpp: IrcRequest): one slot per possible lookup of the frame, in four ranges
    //   [0, HB) rtdgi validate | [HB, 2 HB) rtdgi trace | [2 HB, 2 HB + E) the cache's validate rays | [2 HB + E, 2 HB + 2 E) its trace rays
    // and, when the frame has reflections (kj_ircache_set_rtr_requests), two more behind them: [.., + HB) rtr validate | [.., + HB) rtr trace
    bool deferred = false;
    uint32_t ray_pass_schedule = 2;           // KJ_IRC_PASSES_*: kj_ircache_set_ray_pass_schedule (default: the chain; env KJ_IRC_SCHEDULE / KJ_IRC_SIDE_BY_SIDE at creation)
    bool rtr_requests = false;
    bool requests_begun = false;        // kj_ircache_begin_requests ran for the frame kj_ircache_prepare is about to open (deferred mode)
    uint32_t req_half_pixels = 0;       // HB of the current frame
    kj::DevBuf freed, aux_snapshot, requests, request_cells;
    // the reduction of a frame's records (ircache.hip: IRC_SUMMARY_BYTES each): [0] what a rank of the split sends -- its strip's per-pixel lookups --, [1] what stays local (the cache's own
    // ray passes); alloc_min: per cell, the allocation winner (key << 32 | locator), all-ones between uses; req_scratch: 384 block counts + the total
    kj::DevBuf summary[2], alloc_min, req_scratch;
    bool summary_fresh = false, req_clear_all = false;
    bool begin_cleared_frame_state = false;      // this frame's kj_ircache_begin_requests also cleared `freed` and the ray counters (one launch instead of three)
    hipError_t err = hipSuccess;
    static constexpr uint32_t REQ_E = IRC_MAX_ENTRIES * IRC_SAMPLES_PER_FRAME;
    uint32_t rtr_request_base() const { return 2u * req_half_pixels + 2u * REQ_E; }
    uint32_t request_slots() const { return (rtr_requests ? 4u : 2u) * req_half_pixels + 2u * REQ_E; }
    kj::IrcacheView view() const;
};
