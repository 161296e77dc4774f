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
    kj::IrcacheView view() const;
};
