// Native orchestrator of the screen-tile split (the compiled counterpart of kajiya_amd/multigpu.py, which stays the
// reference implementation it is tested against: tests/test_gpu_multigpu.py::test_native_split_*).
//
// One KjSplit per process. It owns no renderer state: it is handed the renderers of the ranks that live in this process
// (one for a real run, all of them for the virtual-rank tests) and drives RtdgiRenderer::render / TaaRenderer::render strip by strip
// through the same C-ABI a single-GPU caller uses, exchanging halos between passes:
//   * every rank keeps full-size surfaces; a pass runs on the rank's own full-res rows (16-aligned cuts, so 8x8 half-res tiles never
//     straddle one), reads reach into the neighbours' rows, and the rows a consumer can reach are fetched from their owners first;
//   * ONE message per peer and exchange point: the row blocks bound for a peer are packed into a staging buffer (device-to-device
//     copies on the frame's stream), sent with ncclSend inside one ncclGroup with the matching ncclRecv, and scattered on arrival.
//     Both ends enumerate (item, destination, source) in the same order, so the packed layouts agree without a header;
//   * RCCL is resolved with dlopen at the first use (the library has no link-time dependency on it; a single-GPU process never loads
//     it). With every rank in the process the same packed buffers travel by a device-to-device copy instead.
// The schedule (which surfaces, how many halo rows, which passes over-compute instead of exchanging) is the one documented in
// kajiya_amd/multigpu.py and DESIGN 7; the reach of every pass is cited there.
#include "kj_host.hpp"
#include "kj_ircache_host.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <vector>

using namespace kj;

namespace {

struct SurfInfo { uint32_t bytes_per_texel; bool half; };
// surface name (without the ping-pong suffix) -> texel size and resolution; the split's working set (multigpu.py: SURF / TAA_SURF)
const std::map<std::string, SurfInfo>& surf_table() {
    static const std::map<std::string, SurfInfo> t = {
        {"rtdgi.reservoir", {8, true}}, {"rtdgi.ray_orig", {16, true}}, {"rtdgi.ray", {8, true}}, {"rtdgi.radiance", {8, true}}, {"rtdgi.hit_normal", {8, true}},
        {"rtdgi.invalidity", {4, true}}, {"rtdgi.candidate", {8, true}}, {"rtdgi.temporal2", {8, false}}, {"rtdgi.temporal2_var", {4, false}},
        {"rt_history_validity_pre_input_tex", {1, true}}, {"rt_history_validity_input_tex", {1, true}}, {"candidate_radiance_tex", {8, true}},
        {"candidate_hit_tex", {8, true}}, {"temporal_reservoir_packed_tex", {16, true}}, {"reservoir_output_tex0", {8, true}}, {"reservoir_output_tex1", {8, true}},
        {"irradiance_output_tex", {8, false}}, {"temporal_filtered_tex", {8, false}}, {"spatial_filtered_tex", {8, false}}, {"reprojected_history_tex", {8, false}},
        {"SSGI/ssgi", {2, false}}, {"SSGI/filtered_output_tex", {1, false}},
        {"SHADOW/shadow_denoise_moments", {8, false}}, {"SHADOW/shadow_denoise_accum", {4, false}}, {"SHADOW/mask", {1, false}},      // "SHADOW/mask": the caller's image (kj_split_shadow_frame)
        {"LIT/input", {8, false}},        // the caller's RGBA16F image TAA resolves instead of the GI image (kj_split_taa_frame_on)
        {"RTR/rtr.temporal", {8, false}}, {"RTR/rtr.ray_len", {4, false}}, {"RTR/rtr.irradiance", {8, true}}, {"RTR/rtr.ray_orig", {16, true}}, {"RTR/rtr.ray", {8, true}},
        {"RTR/rtr.reservoir", {8, true}}, {"RTR/rtr.rng", {4, true}}, {"RTR/rtr.hit_normal", {8, true}},      // RtrRenderer's eight ping-pong temporals (kj_split_rtr_frame)
        {"TAA/taa", {8, false}}, {"TAA/taa.velocity", {4, false}}, {"TAA/taa.smooth_var", {8, false}}, {"TAA/this_frame_output_img", {8, false}},
        {"selftest.h16", {16, true}}, {"selftest.h1", {1, true}}, {"selftest.f8", {8, false}}, {"selftest.f4", {4, false}},      // kj_split_self_test's scratch images
    };
    return t;
}

// the handful of RCCL entry points the exchange needs, resolved at run time
struct Rccl {
    struct Id128 { char b[128]; };      // ncclUniqueId, passed by value
    void* lib = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        // KJ_RCCL_LIB: a specific RCCL build (or, in tests/test_multigpu_emulated.py, a socket-backed stand-in that lets this transport run between CPU processes)
        if (const char* over = getenv("KJ_RCCL_LIB")) lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
        if (!lib) return false;
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv"); AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        CommCount = (decltype(CommCount))sym("ncclCommCount"); CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
        return GroupStart && GroupEnd && Send && Recv && AllGather && GetUniqueId && CommInitRank && CommDestroy;
    }
};
Rccl g_rccl;
constexpr int NCCL_UINT8 = 1;     // ncclDataType_t (nccl.h): ncclInt8 0, ncclUint8 1

struct Item { std::string name; int halo; uint32_t pin = 0; };        // halo < 0: every row (all-gather); pin: the image's first rows, which every rank holds as well
struct Block { uint32_t src, dst; std::string name; uint32_t row0, row1; };

}  // namespace

struct KjSplit {
    uint32_t world = 0, first = 0, local = 0, W = 0, H = 0, hw = 0, hh = 0, motion_halo = 8;
    std::vector<KjSplitRank> ranks;                        // the local ones, rank = first + index
    std::vector<std::pair<uint32_t, uint32_t>> strips;     // full-res rows [r0, r1) of every rank of the job
    uint32_t frame = 0, taa_frames = 0, ssgi_frames = 0;
    std::vector<KjSsgi*> ssgi;                             // the local ranks' SsgiRenderers (kj_split_ssgi_frame)
    std::vector<KjShadowDenoise*> shadow;                  // the local ranks' ShadowDenoiseRenderers (kj_split_shadow_frame)
    uint32_t shadow_frames = 0;
    std::vector<KjRtr*> rtr;                               // the local ranks' RtrRenderers (kj_split_rtr_frame)
    uint32_t rtr_frames = 0;
    bool with_rtr = false;                                 // kj_split_set_rtr: reflections follow the GI frame -- the cache's replay moves behind their ray passes
    bool consistent_ircache = false;
    std::vector<uint8_t> ircache_was_deferred;             // each local cache's mode before kj_split_create changed it: restored by kj_split_destroy
    bool rtr_requests_set = false;                         // kj_split_set_rtr switched the caches to the four-range slot layout: undone by kj_split_destroy
    bool merge_pending = false;                            // a frame recorded cache lookups that nobody has replayed yet: the next frame's head requires it clear
    void* nccl = nullptr;                                  // ncclComm_t; null: every rank is local
    bool loopback = false;                                 // every rank is local AND a one-rank communicator is the wire: each message is an ncclSend to self matched by an ncclRecv from self
                                                           // (the transport code and RCCL itself on the one GPU a build box has; tests/test_gpu_rccl_one_rank.py)
    std::map<std::pair<uint32_t, std::string>, std::pair<uint8_t*, uint64_t>> surfaces;     // (local index, name) -> base pointer, bytes
    std::vector<DevBuf> send_stage, recv_stage;            // [local rank * world + peer]: the packed rows of one exchange
    // the cache's recorded updates of a frame: every rank's strip summary (kj_ircache_summarize_requests), all-gathered -- RCCL only; virtual ranks read each other's
    DevBuf gathered;
    DevBuf selftest_gather[2];                             // kj_split_self_test's fixed-size all-gather
    // kj_split_set_profiling: HIP events around every exchange (pack + transport + scatter), bytes arriving at the busiest local rank
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::vector<hipEvent_t> ev_free;
    double exchange_ms = 0.0;
    uint64_t exchange_bytes = 0;
    uint32_t exchange_points = 0, frames_profiled = 0;
    hipEvent_t take_event() { hipEvent_t e = nullptr; if (!ev_free.empty()) { e = ev_free.back(); ev_free.pop_back(); } else if (hipEventCreate(&e) != hipSuccess) e = nullptr; return e; }
    ~KjSplit() { for (auto& p : ev_pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); } for (hipEvent_t e : ev_free) (void)hipEventDestroy(e); }
};

namespace {

std::pair<uint32_t, uint32_t> half_rows(const KjSplit& s, std::pair<uint32_t, uint32_t> f) { return {f.first / 2, f.second == s.H ? s.hh : f.second / 2}; }

bool is_local(const KjSplit& s, uint32_t rank) { return rank >= s.first && rank < s.first + s.local; }

const SurfInfo* info_of(const std::string& name) {
    const std::string base = name.substr(0, name.find(':'));
    auto it = surf_table().find(base);
    return it == surf_table().end() ? nullptr : &it->second;
}

// the cached base pointers of one renderer kind ("SSGI/", "SHADOW/", "RTR/"): dropped when the caller binds other handles than last frame
void forget_surfaces(KjSplit& s, const char* prefix) {
    for (auto it = s.surfaces.begin(); it != s.surfaces.end();) it = it->first.second.rfind(prefix, 0) == 0 ? s.surfaces.erase(it) : std::next(it);
}

KjStatus surface_of(KjSplit& s, uint32_t rank, const std::string& name, uint8_t** out, uint32_t* row_bytes) {
    const SurfInfo* si = info_of(name);
    KJ_REQUIRE(si && is_local(s, rank), "unknown surface / rank not in this process");
    const uint32_t li = rank - s.first;
    auto key = std::make_pair(li, name);
    auto it = s.surfaces.find(key);
    if (it == s.surfaces.end()) {     // renderer surfaces are allocated once per extent: the pointer is stable
        void* p = nullptr; uint64_t bytes = 0;
        KJ_REQUIRE(name.rfind("SSGI/", 0) != 0 || (li < s.ssgi.size() && s.ssgi[li]), "no SsgiRenderer bound to this rank");
        KJ_REQUIRE(name.rfind("SHADOW/", 0) != 0 || (li < s.shadow.size() && s.shadow[li]), "no ShadowDenoiseRenderer bound to this rank");
        KJ_REQUIRE(name.rfind("RTR/", 0) != 0 || (li < s.rtr.size() && s.rtr[li]), "no RtrRenderer bound to this rank");
        const KjStatus st = name.rfind("TAA/", 0) == 0    ? kj_taa_surface(s.ranks[li].taa, name.c_str() + 4, &p, &bytes)
                            : name.rfind("SSGI/", 0) == 0 ? kj_ssgi_surface(s.ssgi[li], name.c_str() + 5, &p, &bytes)
                            : name.rfind("SHADOW/", 0) == 0 ? kj_shadow_denoise_surface(s.shadow[li], name.c_str() + 7, &p, &bytes)
                            : name.rfind("RTR/", 0) == 0 ? kj_rtr_surface(s.rtr[li], name.c_str() + 4, &p, &bytes)
                                                          : kj_rtdgi_surface(s.ranks[li].rtdgi, name.c_str(), &p, &bytes);
        if (st != KJ_OK) return st;
        it = s.surfaces.emplace(key, std::make_pair((uint8_t*)p, bytes)).first;
    }
    *row_bytes = (si->half ? s.hw : s.W) * si->bytes_per_texel;
    KJ_REQUIRE(it->second.second == uint64_t(*row_bytes) * (si->half ? s.hh : s.H), "surface extent does not match the split's");
    *out = it->second.first;
    return KJ_OK;
}

// every (source rank, destination rank, rows) so that each rank holds [own0 - halo, own1 + halo) of the surface afterwards
void plan(const KjSplit& s, const std::vector<Item>& items, std::vector<Block>& out) {
    for (const Item& it : items) {
        const SurfInfo* si = info_of(it.name);
        const uint32_t total = si->half ? s.hh : s.H;
        for (uint32_t dst = 0; dst < s.world; ++dst) {
            const auto od = si->half ? half_rows(s, s.strips[dst]) : s.strips[dst];
            const uint32_t lo = it.halo < 0 ? 0u : uint32_t(std::max<int64_t>(0, int64_t(od.first) - it.halo));
            const uint32_t hi = it.halo < 0 ? total : std::min<uint32_t>(total, od.second + uint32_t(it.halo));
            // the rows around the strip, and the pinned top rows where they are not among them (multigpu.py: transfers)
            const uint32_t pin = std::min(it.pin, total);
            std::pair<uint32_t, uint32_t> spans[2] = {{lo, hi}, {0, 0}};
            if (pin > 0) { if (lo <= pin) spans[0] = {0u, std::max(hi, pin)}; else { spans[0] = {0u, pin}; spans[1] = {lo, hi}; } }
            for (const auto& sp : spans)
                for (uint32_t src = 0; src < s.world; ++src) {
                    if (src == dst) continue;
                    const auto os = si->half ? half_rows(s, s.strips[src]) : s.strips[src];
                    const uint32_t a = std::max(sp.first, os.first), b = std::min(sp.second, os.second);
                    if (b > a) out.push_back(Block{src, dst, it.name, a, b});
                }
        }
    }
}

// ONE batched exchange for all items: pack per (local rank, peer), transport, scatter. Virtual ranks take the same path with a
// device-to-device copy as the transport, so the packing order and offsets RCCL relies on are what the virtual-rank tests exercise.
KjStatus exchange(KjSplit& s, const std::vector<Item>& items, hipStream_t st) {
    if (items.empty() || s.world == 1) return KJ_OK;
    for (const Item& it : items) KJ_REQUIRE(info_of(it.name), "unknown surface name in an exchange");
    std::vector<Block> blocks;
    plan(s, items, blocks);
    auto row_bytes = [&](const Block& b) { return size_t((info_of(b.name)->half ? s.hw : s.W)) * info_of(b.name)->bytes_per_texel; };
    // bytes per (local rank, peer), both directions
    std::vector<std::vector<size_t>> send_bytes(s.local, std::vector<size_t>(s.world, 0)), recv_bytes(s.local, std::vector<size_t>(s.world, 0));
    for (const Block& b : blocks) {
        if (is_local(s, b.src)) send_bytes[b.src - s.first][b.dst] += size_t(b.row1 - b.row0) * row_bytes(b);
        if (is_local(s, b.dst)) recv_bytes[b.dst - s.first][b.src] += size_t(b.row1 - b.row0) * row_bytes(b);
    }
    for (uint32_t li = 0; li < s.local; ++li)
        for (uint32_t p = 0; p < s.world; ++p) {
            DevBuf &sb = s.send_stage[li * s.world + p], &rbuf = s.recv_stage[li * s.world + p];
            if (sb.bytes < send_bytes[li][p]) KJ_TRY_HIP(sb.alloc(send_bytes[li][p] + send_bytes[li][p] / 4, st));
            if (rbuf.bytes < recv_bytes[li][p]) KJ_TRY_HIP(rbuf.alloc(recv_bytes[li][p] + recv_bytes[li][p] / 4, st));
        }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (s.profiling) {
        size_t busiest = 0;
        for (uint32_t li = 0; li < s.local; ++li) { size_t n = 0; for (uint32_t p = 0; p < s.world; ++p) n += recv_bytes[li][p]; busiest = std::max(busiest, n); }
        s.exchange_bytes += busiest; ++s.exchange_points;
        ev0 = s.take_event(); ev1 = s.take_event();
        if (ev0) KJ_TRY_HIP(hipEventRecord(ev0, st));
    }
    std::vector<std::vector<size_t>> off(s.local, std::vector<size_t>(s.world, 0));
    std::vector<CopyBlock> copies;       // the row blocks of a step, moved by ONE launch (kj_host.hpp: launch_copy_blocks) instead of one hipMemcpyAsync each
    for (const Block& b : blocks) {      // pack: the blocks bound for a peer, in plan order
        if (!is_local(s, b.src)) continue;
        const uint32_t li = b.src - s.first;
        uint8_t* ps; uint32_t rb;
        const KjStatus e = surface_of(s, b.src, b.name, &ps, &rb); if (e != KJ_OK) return e;
        const size_t n = size_t(b.row1 - b.row0) * rb;
        copies.push_back(CopyBlock{ps + size_t(b.row0) * rb, (uint8_t*)s.send_stage[li * s.world + b.dst].p + off[li][b.dst], n});
        off[li][b.dst] += n;
    }
    KJ_TRY_HIP(launch_copy_blocks(copies.data(), copies.size(), st));
    if (s.nccl && s.loopback) {       // every rank lives here, the wire is RCCL all the same: one send to self + the matching receive per (source, destination), in one group
        KJ_REQUIRE(g_rccl.GroupStart() == 0, "ncclGroupStart failed");
        bool ok = true;      // an error inside the group must not return before ncclGroupEnd
        for (uint32_t src = 0; src < s.world && ok; ++src)
            for (uint32_t dst = 0; dst < s.world && ok; ++dst)
                if (send_bytes[src][dst]) {
                    ok = ok && send_bytes[src][dst] == recv_bytes[dst][src];
                    ok = ok && g_rccl.Send(s.send_stage[src * s.world + dst].p, send_bytes[src][dst], NCCL_UINT8, 0, s.nccl, st) == 0;
                    ok = ok && g_rccl.Recv(s.recv_stage[dst * s.world + src].p, send_bytes[src][dst], NCCL_UINT8, 0, s.nccl, st) == 0;
                }
        const bool ended = g_rccl.GroupEnd() == 0;
        KJ_REQUIRE(ok && ended, "ncclSend / ncclRecv to self / ncclGroupEnd failed");
    } else if (s.nccl) {       // one message per peer, all of them in one group
        KJ_REQUIRE(g_rccl.GroupStart() == 0, "ncclGroupStart failed");
        bool ok = true;
        for (uint32_t p = 0; p < s.world && ok; ++p) {
            if (send_bytes[0][p]) ok = ok && g_rccl.Send(s.send_stage[p].p, send_bytes[0][p], NCCL_UINT8, int(p), s.nccl, st) == 0;
            if (recv_bytes[0][p]) ok = ok && g_rccl.Recv(s.recv_stage[p].p, recv_bytes[0][p], NCCL_UINT8, int(p), s.nccl, st) == 0;
        }
        const bool ended = g_rccl.GroupEnd() == 0;
        KJ_REQUIRE(ok && ended, "ncclSend / ncclRecv / ncclGroupEnd failed");
    } else {            // every rank lives here: the "wire" is a copy from the sender's staging buffer to the receiver's
        copies.clear();
        for (uint32_t src = 0; src < s.world; ++src)
            for (uint32_t dst = 0; dst < s.world; ++dst)
                if (send_bytes[src][dst]) {
                    KJ_REQUIRE(send_bytes[src][dst] == recv_bytes[dst][src], "packed sizes disagree between the two ends of an exchange");
                    copies.push_back(CopyBlock{s.send_stage[src * s.world + dst].p, s.recv_stage[dst * s.world + src].p, send_bytes[src][dst]});
                }
        KJ_TRY_HIP(launch_copy_blocks(copies.data(), copies.size(), st));
    }
    for (auto& o : off) std::fill(o.begin(), o.end(), 0);
    copies.clear();
    for (const Block& b : blocks) {      // scatter, in the same order
        if (!is_local(s, b.dst)) continue;
        const uint32_t li = b.dst - s.first;
        uint8_t* pd; uint32_t rb;
        const KjStatus e = surface_of(s, b.dst, b.name, &pd, &rb); if (e != KJ_OK) return e;
        const size_t n = size_t(b.row1 - b.row0) * rb;
        copies.push_back(CopyBlock{(const uint8_t*)s.recv_stage[li * s.world + b.src].p + off[li][b.src], pd + size_t(b.row0) * rb, n});
        off[li][b.src] += n;
    }
    KJ_TRY_HIP(launch_copy_blocks(copies.data(), copies.size(), st));
    if (ev0 && ev1) { KJ_TRY_HIP(hipEventRecord(ev1, st)); s.ev_pending.push_back({ev0, ev1}); }
    return KJ_OK;
}

std::pair<uint32_t, uint32_t> grow(const KjSplit& s, uint32_t rank, uint32_t rows) {
    const auto f = s.strips[rank];
    return {f.first > rows ? f.first - rows : 0u, std::min(s.H, f.second + rows)};
}

KjStatus render(KjSplit& s, uint32_t li, const KjSplitFrame& fr, uint32_t mask, std::pair<uint32_t, uint32_t> rows, uint32_t spatial_select, hipStream_t st) {
    KjRtdgiRenderParams p = fr.rtdgi;
    p.pass_mask = mask;
    p.row_begin = rows.first; p.row_end = rows.second;
    p.spatial_pass_select = spatial_select;
    return kj_rtdgi_render(s.ranks[li].rtdgi, &p, fr.rtdgi_out, st);
}

KjStatus ircache_head(KjSplit& s, uint32_t li, const KjSplitFrame& fr, hipStream_t st) {
    KjIrcache* c = s.ranks[li].ircache;
    KjStatus e;
    const auto h = half_rows(s, s.strips[s.first + li]);      // this rank's per-pixel passes run on its strip only: only those rows' record slots are cleared
    if (s.consistent_ircache && (e = kj_ircache_begin_requests_rows(c, s.hw, s.hh, h.first, h.second, st)) != KJ_OK) return e;
    if ((e = kj_ircache_prepare(c, st)) != KJ_OK) return e;
    return kj_ircache_trace_irradiance(c, s.ranks[li].scene, fr.sky_cube16, 16, st);
}

// This frame's recorded cache updates: every local rank reduces its strip's per-pixel lookups (rtdgi's validate and trace pass -- rows of the half-res image are
// contiguous slots --, and rtr's two with reflections in the frame) into its cache's summary 0 and the cache's own ray passes' (replicated: identical on every
// rank) into summary 1; the strip summaries are all-gathered -- ONE fixed-size ncclAllGather; virtual ranks read each other's buffers -- and every rank merges
// the same summaries in rank order plus its local one (ircache.hip: a reduction whose result does not depend on the number of ranks). Nothing is read back:
// no list lengths, no host synchronisation (rounds 2-5: two hipStreamSynchronize per split frame for the record counts, a sort of the merged records).
KjStatus merge_ircache_requests(KjSplit& s, hipStream_t st) {
    s.merge_pending = false;
    const size_t SB = size_t(kj_ircache_summary_bytes());
    std::vector<void*> strip_sum(s.local, nullptr), own_sum(s.local, nullptr);
    for (uint32_t li = 0; li < s.local; ++li) {
        KjIrcache* c = s.ranks[li].ircache;
        uint32_t first[4], count[4];
        KjStatus e = kj_ircache_request_ranges(c, first, count); if (e != KJ_OK) return e;
        const auto h = half_rows(s, s.strips[s.first + li]);
        const uint32_t px = (h.second - h.first) * s.hw;
        uint32_t rtr_first[2] = {0, 0}, rtr_count[2] = {0, 0};
        if (s.with_rtr && (e = kj_ircache_rtr_request_ranges(c, rtr_first, rtr_count)) != KJ_OK) return e;
        const uint32_t sf[4] = {first[0] + h.first * s.hw, first[1] + h.first * s.hw, rtr_first[0] + h.first * s.hw, rtr_first[1] + h.first * s.hw}, sc[4] = {px, px, px, px};
        if ((e = kj_ircache_summarize_requests(c, sf, sc, s.with_rtr ? 4u : 2u, 0u, st)) != KJ_OK) return e;
        if ((e = kj_ircache_summarize_requests(c, first + 2, count + 2, 2u, 1u, st)) != KJ_OK) return e;
        if ((e = kj_ircache_summary(c, 0u, &strip_sum[li])) != KJ_OK || (e = kj_ircache_summary(c, 1u, &own_sum[li])) != KJ_OK) return e;
    }
    std::vector<const void*> src(size_t(s.world) + 1);
    if (s.profiling) { s.exchange_bytes += SB * (s.world - 1u); ++s.exchange_points; }      // what the all-gather delivers to every rank (virtual ranks read in place)
    if (s.nccl) {
        if (s.gathered.bytes < SB * s.world) KJ_TRY_HIP(s.gathered.alloc(SB * s.world, st));
        if (s.loopback)      // a one-rank communicator: every virtual rank's summary goes through an all-gather of one
            for (uint32_t p = 0; p < s.world; ++p) KJ_REQUIRE(g_rccl.AllGather(strip_sum[p], (uint8_t*)s.gathered.p + size_t(p) * SB, SB, NCCL_UINT8, s.nccl, st) == 0, "ncclAllGather failed");
        else KJ_REQUIRE(g_rccl.AllGather(strip_sum[0], s.gathered.p, SB, NCCL_UINT8, s.nccl, st) == 0, "ncclAllGather failed");
        for (uint32_t p = 0; p < s.world; ++p) src[p] = (const uint8_t*)s.gathered.p + size_t(p) * SB;
    } else
        for (uint32_t p = 0; p < s.world; ++p) src[p] = strip_sum[p];
    for (uint32_t li = 0; li < s.local; ++li) {
        src[s.world] = own_sum[li];
        const KjStatus e = kj_ircache_apply_summaries(s.ranks[li].ircache, src.data(), s.world + 1u, st);
        if (e != KJ_OK) return e;
    }
    return KJ_OK;
}

std::string sfx(const char* name, uint32_t k) { return std::string(name) + ":" + std::to_string(k); }

// Half-res rows the taps of rtr's resolve can land beyond the rows it runs on (multigpu.py: rtr_resolve_halo has the derivation), or -1: no useful bound.
// The same expression in double on both sides: the two ends of an exchange must agree on it.
int rtr_resolve_halo(uint32_t height, float clip_to_view_11) {
    const double t = std::fabs(double(clip_to_view_11)), k = std::max(0.1, 4.0 / double(height));
    if (!(k * t < 0.5)) return -1;
    return int(std::ceil(k * std::sqrt(1.0 + t * t) / (1.0 - k * t) * double(height) / 4.0)) + 4;
}

}  // namespace

extern "C" {

KjStatus kj_split_create(uint32_t world, uint32_t first_rank, uint32_t local_ranks, const KjSplitRank* ranks, uint32_t width, uint32_t height, uint32_t motion_halo,
                         void* nccl_comm, KjSplit** out) {
    KJ_REQUIRE(out && ranks && world >= 1 && local_ranks >= 1 && first_rank + local_ranks <= world, "bad rank layout");
    KJ_REQUIRE(nccl_comm ? (local_ranks == 1 || local_ranks == world) : local_ranks == world,
               "without a communicator every rank must live in this process; with one, exactly one does -- or all of them over a ONE-rank communicator (loopback)");
    KJ_REQUIRE(width >= 16 && height >= 16 * world, "image too small for this many strips");
    if (nccl_comm) KJ_REQUIRE(g_rccl.load(), "librccl.so could not be loaded");
    const bool loopback = nccl_comm && local_ranks == world && world > 1;
    if (loopback) {
        int n = 0;
        KJ_REQUIRE(g_rccl.CommCount && g_rccl.CommCount(nccl_comm, &n) == 0 && n == 1, "loopback (every rank local, RCCL as the wire) needs a communicator of exactly one rank");
    }
    KjSplit* s = new KjSplit();
    s->world = world; s->first = first_rank; s->local = local_ranks; s->W = width; s->H = height; s->hw = (width + 1) / 2; s->hh = (height + 1) / 2;
    s->motion_halo = motion_halo; s->nccl = nccl_comm; s->loopback = loopback;
    s->ranks.assign(ranks, ranks + local_ranks);
    // 16-aligned cuts, as even as possible (multigpu.py: plan_strips)
    const uint32_t units = (height + 15) / 16, base = units / world, extra = units % world;
    uint32_t u = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const uint32_t cnt = base + (r < extra ? 1u : 0u);
        s->strips.push_back({std::min(u * 16, height), std::min((u + cnt) * 16, height)});
        u += cnt;
    }
    bool all_cached = true;
    for (const KjSplitRank& r : s->ranks) { if (!r.rtdgi || !r.taa || !r.scene) { delete s; KJ_REQUIRE(false, "a rank needs rtdgi, taa and scene handles"); } all_cached &= r.ircache != nullptr; }
    s->consistent_ircache = all_cached;
    for (const KjSplitRank& r : s->ranks) {
        s->ircache_was_deferred.push_back(r.ircache && r.ircache->deferred ? 1 : 0);
        if (r.ircache) { const KjStatus e = kj_ircache_set_deferred_updates(r.ircache, all_cached ? 1u : 0u); if (e != KJ_OK) { delete s; return e; } }
    }
    KJ_REQUIRE(world + 1u <= 32u || !all_cached, "the cache's replay merges at most 32 summaries (31 ranks)");
    s->send_stage = std::vector<DevBuf>(size_t(local_ranks) * world); s->recv_stage = std::vector<DevBuf>(size_t(local_ranks) * world);
    *out = s;
    return KJ_OK;
}
void kj_split_destroy(KjSplit* s) {
    if (!s) return;
    // hand the caches back in the mode they came in: a plain kj_rtdgi_render caller would otherwise keep RECORDING lookups that nobody replays
    for (size_t i = 0; i < s->ranks.size() && i < s->ircache_was_deferred.size(); ++i)
        if (s->ranks[i].ircache) {
            kj_ircache_set_deferred_updates(s->ranks[i].ircache, s->ircache_was_deferred[i]);
            if (s->rtr_requests_set) kj_ircache_set_rtr_requests(s->ranks[i].ircache, 0);      // back to the two-range slot layout a plain kj_rtdgi_render caller expects
        }
    delete s;
}

KjStatus kj_split_strip(KjSplit* s, uint32_t rank, uint32_t* out_row_begin, uint32_t* out_row_end) {
    KJ_REQUIRE(s && rank < s->world && out_row_begin && out_row_end, "bad argument");
    *out_row_begin = s->strips[rank].first; *out_row_end = s->strips[rank].second;
    return KJ_OK;
}

#define KJ_SPLIT_TRY(expr) do { const KjStatus e_ = (expr); if (e_ != KJ_OK) return e_; } while (0)

// One rtdgi frame (multigpu.py: SplitRtdgi.gi_frame). `frames`: one entry per local rank.
KjStatus kj_split_gi_frame(KjSplit* s, const KjSplitFrame* frames, uint32_t flags, void* trace_done_event, void* stream) {
    KJ_REQUIRE(s && frames, "null argument");
    const bool ircache_done = (flags & KJ_SPLIT_IRCACHE_DONE) != 0, defer_merge = (flags & KJ_SPLIT_DEFER_IRCACHE_MERGE) != 0;
    hipStream_t st = (hipStream_t)stream;
    // a frame whose recorded lookups were never replayed (kj_split_rtr_frame / kj_split_merge_ircache left out) would silently stop the caches
    // from allocating and refreshing entries
    KJ_REQUIRE(!s->merge_pending, "the previous frame's recorded cache updates were never replayed: call kj_split_rtr_frame or kj_split_merge_ircache before the next kj_split_gi_frame");
    const uint32_t KEEP = KJ_RTDGI_PASS_KEEP_TEMPORALS, M = s->motion_halo;
    const uint32_t out_i = s->frame % 2, hist_i = 1 - s->frame % 2;
    std::vector<Item> items;
    // ---- A (gone since round 6): last frame's denoised GI and its variance are read through the motion vectors only (fullres_reproject: a 4x4 footprint around
    // the reprojected pixel; temporal_filter: a bilinear tap) -- halos, like TAA's histories. They are final when their frame's temporal filter / taa pass has run, so
    // they travel THEN, with exchange H of that frame and with kj_split_taa_frame's closing exchange, instead of in an exchange point of their own at the start of this
    // one. What the trace pass reads ANYWHERE on screen is the reprojected image: every rank reprojects its own strip and the strips are all-gathered (A').
    for (uint32_t li = 0; li < s->local; ++li) {
        const KjSplitRank& r = s->ranks[li];
        if (r.ircache && !ircache_done) KJ_SPLIT_TRY(ircache_head(*s, li, frames[li], st));
        const auto own = s->strips[s->first + li];
        KJ_SPLIT_TRY(kj_rtdgi_reproject_rows(r.rtdgi, frames[li].rtdgi.reprojection_map, s->W, s->H, own.first, own.second, st));
    }
    items.clear();
    items.push_back({"reprojected_history_tex", -1});
    KJ_SPLIT_TRY(exchange(*s, items, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        const KjSplitRank& r = s->ranks[li];
        if (r.ircache) KJ_SPLIT_TRY(kj_ircache_sum_up_irradiance_for_sampling(r.ircache, st));
        // the half-res images and G-buffer records for the WHOLE frame, from replicated inputs (27 us per rank at 4K). Not only on own +- 64 half-res rows (tried in round 6:
        // the wide-angle reflections case of the bit-exactness tests fails): the second spatial pass' occlusion march walks from the pixel towards the sample's HIT point for up to
        // three times the screen distance of the sample's pixel (occlusion_raymarch.hlsl via restir_spatial.hlsl:235-262) and reads the half-res depth along the way -- well beyond
        // the halo of everything else under a grazing, wide field of view.
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_EXTRACT_HALF, {0, 0}, 0, st));
        // `rtdgi validate`, and with it -- where the library has the two ray passes of a validation frame as ONE launch (k_rtdgi_validate_and_trace: a strip's launch of either pass
        // lasts as long as its slowest waves) -- `rtdgi trace` without its last statement, the one read of the validate pass' output (at the reprojected pixel: exchange B's halo)
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_VALIDATE | KJ_RTDGI_PASS_TRACE | KJ_RTDGI_PASS_TRACE_MAY_DEFER | KEEP, s->strips[s->first + li], 0, st));
    }
    // ---- B: validate rewrites the reservoir histories in place
    items.clear();
    items.push_back({"rt_history_validity_pre_input_tex", int(M + 1)});
    if (s->frame > 0) {
        // (row 0 of the four sample images for everybody: a reservoir nothing was ever selected into keeps payload 0 = pixel (0, 0), and the temporal pass
        // follows the payload whatever the reservoir's weight)
        items.push_back({sfx("rtdgi.reservoir", hist_i), int(M + 4)});
        for (const char* n : {"rtdgi.ray_orig", "rtdgi.ray", "rtdgi.radiance", "rtdgi.hit_normal"}) items.push_back({sfx(n, hist_i), int(M + 4), 1u});
        items.push_back({sfx("rtdgi.invalidity", hist_i), int(M + 8)});
    }
    KJ_SPLIT_TRY(exchange(*s, items, st));
    for (uint32_t li = 0; li < s->local; ++li) KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_TRACE_FINISH | KEEP, s->strips[s->first + li], 0, st));      // the rest of `rtdgi trace`: that statement, or the whole pass
    if (s->consistent_ircache) s->merge_pending = true;
    if (s->consistent_ircache && !defer_merge && !s->with_rtr) KJ_SPLIT_TRY(merge_ircache_requests(*s, st));      // (with reflections in the frame: after THEIR ray passes)
    if (trace_done_event) KJ_TRY_HIP(hipEventRecord((hipEvent_t)trace_done_event, st));
    // ---- C
    KJ_SPLIT_TRY(exchange(*s, {{"rt_history_validity_input_tex", 2}, {"candidate_radiance_tex", 8 + 3}, {"candidate_hit_tex", 8 + 3}}, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        // (one call: the library has the two passes as one launch -- a pixel's integrated validity is read by that pixel alone)
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_VALIDITY_INTEGRATE | KJ_RTDGI_PASS_RESTIR_TEMPORAL | KEEP, s->strips[s->first + li], 0, st));
    }
    // ---- D: the one-deep halo; the spatial passes and the resolve over-compute inside it instead of exchanging again
    KJ_SPLIT_TRY(exchange(*s, {{sfx("rtdgi.reservoir", out_i), 64, 1u}, {"temporal_reservoir_packed_tex", 64, 1u}, {sfx("rtdgi.radiance", out_i), 64, 1u}}, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        const uint32_t rank = s->first + li;
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_RESTIR_SPATIAL | KEEP, grow(*s, rank, 64), 1, st));
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_RESTIR_SPATIAL | KEEP, grow(*s, rank, 32), 2, st));
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_RESTIR_RESOLVE | KEEP, grow(*s, rank, 16), 0, st));
        KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_TEMPORAL_FILTER | KEEP, s->strips[rank], 0, st));
    }
    // ---- H: the spatial filter's 16-row reach, 32 rows further because it over-computes what TAA's first passes read around the strip (TAA's input halo is
    // 25 rows: no exchange point of its own), and next frame's history halos of the temporal filter's two outputs
    KJ_SPLIT_TRY(exchange(*s, {{"temporal_filtered_tex", 16 + 32}, {sfx("rtdgi.temporal2", out_i), int(M + 3)}, {sfx("rtdgi.temporal2_var", out_i), int(M + 2)}}, st));
    for (uint32_t li = 0; li < s->local; ++li) KJ_SPLIT_TRY(render(*s, li, frames[li], KJ_RTDGI_PASS_SPATIAL_FILTER | KEEP, grow(*s, s->first + li, 32), 0, st));
    s->frame++;
    if (s->profiling) ++s->frames_profiled;
    return KJ_OK;
}

// The replay a frame issued with KJ_SPLIT_DEFER_IRCACHE_MERGE left out: all-gather of the ranks' summaries of the frame's recorded cache updates + the same
// merge on every rank (no host synchronisation since round 6). A pipelining caller runs it on its cache stream -- before the next frame's kj_ircache_prepare --
// so that the all-gather overlaps the main stream's resampling chain.
KjStatus kj_split_merge_ircache(KjSplit* s, void* stream) {
    KJ_REQUIRE(s, "null argument");
    return s->consistent_ircache ? merge_ircache_requests(*s, (hipStream_t)stream) : KJ_OK;
}

// SsgiRenderer::render strip by strip, before kj_split_gi_frame (multigpu.py: SplitRtdgi.ssgi_frame): the halo of the temporal pass' history, every local
// rank's own rows (kj_ssgi_render_rows over-computes what its passes reach into), then the halo of the finished guide that the rtdgi passes read beyond
// the strip (144 rows: the first spatial pass runs on own +- 64 rows and its taps reach 32 half-res rows further). out_ssao_r8[li] -> rtdgi.ssao_tex.
KjStatus kj_split_ssgi_frame(KjSplit* s, KjSsgi* const* ssgi, const KjSplitFrame* frames, const void** out_ssao_r8, void* stream) {
    KJ_REQUIRE(s && ssgi && frames && out_ssao_r8, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (s->ssgi.size() != s->local || !std::equal(s->ssgi.begin(), s->ssgi.end(), ssgi)) forget_surfaces(*s, "SSGI/");     // another renderer handle: its surfaces live elsewhere
    s->ssgi.assign(ssgi, ssgi + s->local);
    for (KjSsgi* g : s->ssgi) KJ_REQUIRE(g, "null SsgiRenderer");
    for (uint32_t li = 0; li < s->local; ++li) {
        const auto own = s->strips[s->first + li];
        KJ_SPLIT_TRY(kj_ssgi_render_rows(ssgi[li], &frames[li].rtdgi.gbuffer_depth, frames[li].rtdgi.reprojection_map, nullptr, own.first, own.second, &out_ssao_r8[li], st));
    }
    // ONE exchange: the finished guide's halo and, for next frame's temporal pass, the halo of the history this frame has just written (round 5: a second exchange
    // point at the start of the next frame)
    KJ_SPLIT_TRY(exchange(*s, {{sfx("SSGI/filtered_output_tex", s->ssgi_frames % 2), 144 + 2}, {sfx("SSGI/ssgi", s->ssgi_frames % 2), int(s->motion_halo + 2)}}, st));
    ++s->ssgi_frames;
    return KJ_OK;
}

// trace_sun_shadow_mask + ShadowDenoiseRenderer::render strip by strip (multigpu.py: SplitRtdgi.shadow_frame; world_render_passes.rs:124-136): the halo of the
// denoiser's two histories (read through the motion vectors by a temporal pass that over-computes 24 rows either side), every local rank's rays for its own
// rows into the caller's mask image, the mask's 32-row halo (one byte per pixel: cheaper than tracing the neighbours' rays again), then the denoiser's
// passes, each over-computing what the next one reaches into (kj_shadow_denoise_render_rows). out_rg16f[li] is valid on the rank's own rows, which is all
// kj_light_gbuffer_rows reads. `ray_counters_dev`: NULL or one optional device u64 per local rank.
KjStatus kj_split_shadow_frame(KjSplit* s, KjShadowDenoise* const* denoisers, const KjSplitFrame* frames, void* const* mask_r8, uint64_t* const* ray_counters_dev,
                               const void** out_rg16f, void* stream) {
    KJ_REQUIRE(s && denoisers && frames && mask_r8 && out_rg16f, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (s->shadow.size() != s->local || !std::equal(s->shadow.begin(), s->shadow.end(), denoisers)) forget_surfaces(*s, "SHADOW/");
    s->shadow.assign(denoisers, denoisers + s->local);
    for (uint32_t li = 0; li < s->local; ++li) {
        KJ_REQUIRE(s->shadow[li] && mask_r8[li], "null ShadowDenoiseRenderer / mask image");
        s->surfaces[{li, std::string("SHADOW/mask")}] = {(uint8_t*)mask_r8[li], uint64_t(s->W) * s->H};      // the caller's image: bound anew every frame
    }
    if (s->shadow_frames > 0) {
        const uint32_t h = 1 - s->shadow_frames % 2, halo = s->motion_halo + 3 + 24;
        KJ_SPLIT_TRY(exchange(*s, {{sfx("SHADOW/shadow_denoise_moments", h), int(halo)}, {sfx("SHADOW/shadow_denoise_accum", h), int(halo)}}, st));
    }
    for (uint32_t li = 0; li < s->local; ++li) {
        const auto own = s->strips[s->first + li];
        KJ_SPLIT_TRY(kj_trace_sun_shadow_mask_rows(s->ranks[li].scene->dev, s->ranks[li].scene, &frames[li].rtdgi.gbuffer_depth, mask_r8[li], own.first, own.second,
                                                   ray_counters_dev ? ray_counters_dev[li] : nullptr, st));
    }
    KJ_SPLIT_TRY(exchange(*s, {{"SHADOW/mask", 32}}, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        const auto own = s->strips[s->first + li];
        KJ_SPLIT_TRY(kj_shadow_denoise_render_rows(s->shadow[li], &frames[li].rtdgi.gbuffer_depth, mask_r8[li], frames[li].rtdgi.reprojection_map, own.first, own.second, &out_rg16f[li], st));
    }
    ++s->shadow_frames;
    return KJ_OK;
}

// Reflections join the frame (kj_split_rtr_frame after kj_split_gi_frame): the caches reserve slot ranges for the lookups of rtr's rays, and the replay of a
// frame's recorded cache updates moves behind rtr's ray passes (multigpu.py: SplitRtdgi.enable_rtr).
KjStatus kj_split_set_rtr(KjSplit* s, uint32_t enable) {
    KJ_REQUIRE(s, "null argument");
    s->with_rtr = enable != 0;
    if (s->consistent_ircache) for (const KjSplitRank& r : s->ranks) KJ_SPLIT_TRY(kj_ircache_set_rtr_requests(r.ircache, enable));
    s->rtr_requests_set = s->consistent_ircache && enable != 0;
    return KJ_OK;
}

// RtrRenderer::trace + LightingRenderer::render_specular + TracedRtr::filter_temporal strip by strip (kj_rtr_render_rows), after kj_split_gi_frame
// (multigpu.py: SplitRtdgi.rtr_frame documents the reach of every pass; world_render_passes.rs:172-210). Four exchange points: the all-gather of this frame's GI
// image (a reflection's hit reads it anywhere on screen); after the ray passes the six reservoir histories (motion + 16 half-res rows, and row 0 for everybody:
// an empty reservoir's payload is pixel (0, 0)); after the reservoir pass the all-gather of its four outputs (the resolve's taps land where a world-space kernel
// projects to) and the ray-length history's halo; after the temporal filter the all-gather of its output (the cleanup's taps; next frame's filter reads it at the
// reflection's reprojected virtual position). The resolve and the lights' specular over-compute 16 rows either side instead of a fifth exchange.
// `rtr_params`: one per local rank, as for kj_rtr_trace (pass_mask ignored). out_resolved[li] (may be NULL) is valid on the rank's own rows: all kj_light_gbuffer_rows reads.
KjStatus kj_split_rtr_frame(KjSplit* s, KjRtr* const* rtr, const KjRtrParams* rtr_params, uint32_t flags, void* trace_done_event, const void** out_resolved, void* stream) {
    KJ_REQUIRE(s && rtr && rtr_params, "null argument");
    KJ_REQUIRE(s->with_rtr, "kj_split_set_rtr(split, 1) comes first (before the frame's kj_ircache_begin_requests)");
    hipStream_t st = (hipStream_t)stream;
    if (s->rtr.size() != s->local || !std::equal(s->rtr.begin(), s->rtr.end(), rtr)) forget_surfaces(*s, "RTR/");
    s->rtr.assign(rtr, rtr + s->local);
    for (KjRtr* r : s->rtr) KJ_REQUIRE(r, "null RtrRenderer");
    const bool lights = (flags & KJ_SPLIT_RTR_SPECULAR_LIGHTS) != 0, defer_merge = (flags & KJ_SPLIT_DEFER_IRCACHE_MERGE) != 0;
    const uint32_t KEEP = KJ_RTR_PASS_KEEP, M = s->motion_halo, o = s->rtr_frames % 2, h = 1 - s->rtr_frames % 2;
    auto run = [&](uint32_t li, uint32_t mask, std::pair<uint32_t, uint32_t> rows) {
        KjRtrParams p = rtr_params[li];
        p.pass_mask = mask;
        return kj_rtr_render_rows(s->rtr[li], &p, rows.first, rows.second, out_resolved ? &out_resolved[li] : nullptr, st);
    };
    KJ_SPLIT_TRY(exchange(*s, {{"spatial_filtered_tex", -1}}, st));
    for (uint32_t li = 0; li < s->local; ++li) KJ_SPLIT_TRY(run(li, KJ_RTR_PASS_EXTRACT_HALF | KJ_RTR_PASS_VALIDATE | KJ_RTR_PASS_TRACE, s->strips[s->first + li]));
    if (s->consistent_ircache && !defer_merge) KJ_SPLIT_TRY(merge_ircache_requests(*s, st));
    if (trace_done_event) KJ_TRY_HIP(hipEventRecord((hipEvent_t)trace_done_event, st));
    std::vector<Item> items;
    if (s->rtr_frames > 0) {
        for (const char* n : {"RTR/rtr.irradiance", "RTR/rtr.ray_orig", "RTR/rtr.ray", "RTR/rtr.rng", "RTR/rtr.reservoir", "RTR/rtr.hit_normal"}) items.push_back({sfx(n, h), int(M + 16), 1u});
        KJ_SPLIT_TRY(exchange(*s, items, st));
    }
    for (uint32_t li = 0; li < s->local; ++li) KJ_SPLIT_TRY(run(li, KJ_RTR_PASS_RESTIR_TEMPORAL | KEEP, s->strips[s->first + li]));
    items.clear();
    const int reach = rtr_resolve_halo(s->H, s->ranks[0].scene->dev->fc_host.view_constants.clip_to_view[5]);
    for (const char* n : {"RTR/rtr.irradiance", "RTR/rtr.ray", "RTR/rtr.reservoir", "RTR/rtr.ray_orig"}) items.push_back({sfx(n, o), reach < 0 ? -1 : 8 + reach, 1u});      // (the resolve runs on own +- 16 rows)
    items.push_back({"candidate_hit_tex", 8});      // the pixel's own hit distance on the rows the resolve over-computes
    if (s->rtr_frames > 0) items.push_back({sfx("RTR/rtr.ray_len", h), int(M + 2 + 16)});
    KJ_SPLIT_TRY(exchange(*s, items, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        KJ_SPLIT_TRY(run(li, KJ_RTR_PASS_RESOLVE | KEEP | (lights ? KJ_RTR_PASS_SPECULAR_LIGHTS : 0u), grow(*s, s->first + li, 16)));
        KJ_SPLIT_TRY(run(li, KJ_RTR_PASS_TEMPORAL_FILTER | KEEP, s->strips[s->first + li]));
    }
    KJ_SPLIT_TRY(exchange(*s, {{sfx("RTR/rtr.temporal", o), -1}}, st));
    for (uint32_t li = 0; li < s->local; ++li) KJ_SPLIT_TRY(run(li, KJ_RTR_PASS_CLEANUP | KEEP, s->strips[s->first + li]));
    ++s->rtr_frames;
    return KJ_OK;
}

// TaaRenderer::render on this frame's GI image, strip by strip (multigpu.py: SplitRtdgi.taa_frame): the GI image arrives with its halo (kj_split_gi_frame's spatial
// filter over-computes it), the intermediates are over-computed on up to 32 extra rows per side, and ONE exchange closes the frame: the three histories' halos for the
// next frame, final now.
KjStatus kj_split_taa_frame(KjSplit* s, const KjSplitFrame* frames, void* stream) { return kj_split_taa_frame_on(s, frames, nullptr, stream); }

// The same on images of the caller's: input_rgba16f[li] (full res, valid on the rank's own rows -- e.g. what kj_light_gbuffer_rows wrote: the lighting frame of
// world_render_passes.rs:212-291 resolves the LIT image) instead of the GI image; NULL: the GI image. The halo rows are written into the caller's images.
KjStatus kj_split_taa_frame_on(KjSplit* s, const KjSplitFrame* frames, void* const* input_rgba16f, void* stream) {
    KJ_REQUIRE(s && frames, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const char* name = input_rgba16f ? "LIT/input" : "spatial_filtered_tex";
    if (input_rgba16f)
        for (uint32_t li = 0; li < s->local; ++li) {
            KJ_REQUIRE(input_rgba16f[li], "null input image");
            s->surfaces[{li, std::string("LIT/input")}] = {(uint8_t*)input_rgba16f[li], uint64_t(s->W) * s->H * 8};      // the caller's image: bound anew every frame
        }
    // the GI image arrives with its halo already (kj_split_gi_frame's spatial filter over-computes 32 rows either side); a caller's image needs the exchange
    if (input_rgba16f) KJ_SPLIT_TRY(exchange(*s, {{name, 1 + 24}}, st));
    for (uint32_t li = 0; li < s->local; ++li) {
        const uint32_t rank = s->first + li;
        uint8_t* inp; uint32_t rb;
        KJ_SPLIT_TRY(surface_of(*s, rank, name, &inp, &rb));
        const struct { uint32_t mask, grow; bool keep; } steps[5] = {{1, 32, false}, {2 | 4, 24, true}, {8, 16, true}, {16, 8, true}, {32 | 64, 0, true}};
        for (const auto& stp : steps) {
            const auto rows = grow(*s, rank, stp.grow);
            KJ_SPLIT_TRY(kj_taa_render_rows(s->ranks[li].taa, inp, s->W, s->H, frames[li].rtdgi.reprojection_map, frames[li].rtdgi.gbuffer_depth.depth, s->W, s->H, frames[li].taa_out, st,
                                            stp.mask | (stp.keep ? KJ_RTDGI_PASS_KEEP_TEMPORALS : 0u), rows.first, rows.second));
        }
    }
    {   // next frame's history halos (read through the motion vectors by reproject / input_prob / taa), sent now that they are final
        const uint32_t M = s->motion_halo, to = s->taa_frames % 2;
        KJ_SPLIT_TRY(exchange(*s, {{sfx("TAA/taa", to), int(M + 4 + 32)}, {sfx("TAA/taa.velocity", to), int(M + 2 + 16)}, {sfx("TAA/taa.smooth_var", to), int(M + 2 + 16)}}, st));
    }
    s->taa_frames++;
    return KJ_OK;
}

// Assemble a full image on every rank from the owners' rows (result collection; not part of a timed frame).
KjStatus kj_split_gather(KjSplit* s, const char* surface_name, void* stream) {
    KJ_REQUIRE(s && surface_name, "null argument");
    return exchange(*s, {{surface_name, -1}}, (hipStream_t)stream);
}

// Start-up check of the transport, before frame 0 (multigpu.py: SplitRtdgi.self_test is the same check of the Python orchestrator's): every kind of exchange the
// frame schedule uses -- the all-gather of a full-res image, several surfaces of different texel sizes and halos packed into one message per peer, the 64-row
// one-deep halo, stencil halos, and the fixed-size all-gather of the cache's summaries -- runs once on scratch images whose rows carry their OWNER's
// rank, through exchange() (and the fixed-size ncclAllGather the cache's summaries travel by) itself; each local rank's images are then read back and every row it is entitled to must hold the
// owner's pattern, every other row must be untouched. *out_passed = 1 when all local ranks passed; the caller combines the ranks' verdicts (bench.py:
// all-reduce MIN) -- a transport error is returned as an error. Synchronises `stream`; not for use inside a frame.
KjStatus kj_split_self_test(KjSplit* s, uint32_t* out_passed, void* stream) {
    KJ_REQUIRE(s && out_passed, "null argument");
    hipStream_t st = (hipStream_t)stream;
    *out_passed = 1;
    if (s->world == 1) return KJ_OK;
    const int M = int(s->motion_halo);
    const std::vector<std::vector<Item>> rounds = {
        {{"selftest.f8", -1}},
        {{"selftest.h16", M + 4, 1u}, {"selftest.h1", M + 1}, {"selftest.f8", M + 3}, {"selftest.f4", M + 2 + 16}},      // (h16: with row 0 pinned for everybody)
        {{"selftest.h16", 64}, {"selftest.f4", 16}},
        {{"selftest.f8", 1 + 24}, {"selftest.h1", 2}},
    };
    const char* names[4] = {"selftest.h16", "selftest.h1", "selftest.f8", "selftest.f4"};
    std::vector<DevBuf> scratch(size_t(s->local) * 4);
    struct Unregister {          // the scratch images leave the surface cache whatever happens
        KjSplit* s; const char* const* names;
        ~Unregister() { for (uint32_t li = 0; li < s->local; ++li) for (int k = 0; k < 4; ++k) s->surfaces.erase({li, names[k]}); }
    } unregister{s, names};
    for (uint32_t li = 0; li < s->local; ++li)
        for (int k = 0; k < 4; ++k) {
            const SurfInfo* si = info_of(names[k]);
            const size_t bytes = size_t(si->half ? s->hw : s->W) * si->bytes_per_texel * (si->half ? s->hh : s->H);
            KJ_TRY_HIP(scratch[li * 4 + k].alloc(bytes, st));
            s->surfaces[{li, names[k]}] = {(uint8_t*)scratch[li * 4 + k].p, bytes};
        }
    auto rows_of = [&](uint32_t rank, bool half) { return half ? half_rows(*s, s->strips[rank]) : s->strips[rank]; };
    std::vector<uint8_t> host;
    bool ok = true;
    for (size_t ri = 0; ri < rounds.size(); ++ri) {
        for (uint32_t li = 0; li < s->local; ++li)
            for (size_t ii = 0; ii < rounds[ri].size(); ++ii) {
                uint8_t* p; uint32_t rb;
                KJ_SPLIT_TRY(surface_of(*s, s->first + li, rounds[ri][ii].name, &p, &rb));
                const SurfInfo* si = info_of(rounds[ri][ii].name);
                const auto own = rows_of(s->first + li, si->half);
                KJ_TRY_HIP(hipMemsetAsync(p, 0, size_t(rb) * (si->half ? s->hh : s->H), st));
                KJ_TRY_HIP(hipMemsetAsync(p + size_t(own.first) * rb, int((s->first + li + 1) * 8 + ii) & 0xff, size_t(own.second - own.first) * rb, st));
            }
        KJ_SPLIT_TRY(exchange(*s, rounds[ri], st));
        for (uint32_t li = 0; li < s->local; ++li)
            for (size_t ii = 0; ii < rounds[ri].size(); ++ii) {
                uint8_t* p; uint32_t rb;
                KJ_SPLIT_TRY(surface_of(*s, s->first + li, rounds[ri][ii].name, &p, &rb));
                const SurfInfo* si = info_of(rounds[ri][ii].name);
                const uint32_t total = si->half ? s->hh : s->H;
                host.resize(size_t(rb) * total);
                KJ_TRY_HIP(hipMemcpyAsync(host.data(), p, host.size(), hipMemcpyDeviceToHost, st));
                KJ_TRY_HIP(hipStreamSynchronize(st));
                const auto own = rows_of(s->first + li, si->half);
                const int halo = rounds[ri][ii].halo;
                const uint32_t lo = halo < 0 ? 0u : uint32_t(std::max<int64_t>(0, int64_t(own.first) - halo)), hi = halo < 0 ? total : std::min<uint32_t>(total, own.second + uint32_t(halo));
                const uint32_t pin = std::min(rounds[ri][ii].pin, total);
                for (uint32_t row = 0; row < total && ok; ++row) {
                    uint32_t owner = 0;
                    while (owner + 1 < s->world && row >= rows_of(owner, si->half).second) ++owner;
                    const uint8_t want = (row >= lo && row < hi) || row < pin ? uint8_t(((owner + 1) * 8 + ii) & 0xff) : uint8_t(0);
                    const uint8_t* r = &host[size_t(row) * rb];
                    for (uint32_t x = 0; x < rb; ++x) if (r[x] != want) { ok = false; break; }
                }
            }
    }
    // the fixed-size all-gather the cache's summaries travel by: rank r contributes 4 KB of bytes r + 1 (plain virtual ranks read each other's buffers: nothing to test)
    if (s->nccl && ok) {
        const size_t N = 4096;
        KJ_TRY_HIP(s->selftest_gather[0].alloc(N * s->local, st)); KJ_TRY_HIP(s->selftest_gather[1].alloc(N * s->world, st));
        for (uint32_t li = 0; li < s->local; ++li) KJ_TRY_HIP(hipMemsetAsync((uint8_t*)s->selftest_gather[0].p + N * li, int(s->first + li + 1), N, st));
        KJ_TRY_HIP(hipMemsetAsync(s->selftest_gather[1].p, 0, N * s->world, st));
        if (s->loopback)
            for (uint32_t p = 0; p < s->world; ++p)
                KJ_REQUIRE(g_rccl.AllGather((uint8_t*)s->selftest_gather[0].p + N * p, (uint8_t*)s->selftest_gather[1].p + N * p, N, NCCL_UINT8, s->nccl, st) == 0, "ncclAllGather failed");
        else KJ_REQUIRE(g_rccl.AllGather(s->selftest_gather[0].p, s->selftest_gather[1].p, N, NCCL_UINT8, s->nccl, st) == 0, "ncclAllGather failed");
        host.resize(N * s->world);
        KJ_TRY_HIP(hipMemcpyAsync(host.data(), s->selftest_gather[1].p, host.size(), hipMemcpyDeviceToHost, st));
        KJ_TRY_HIP(hipStreamSynchronize(st));
        for (uint32_t r = 0; r < s->world; ++r)
            for (size_t b = 0; b < N; ++b) ok = ok && host[size_t(r) * N + b] == uint8_t(r + 1);
    }
    *out_passed = ok ? 1u : 0u;
    return KJ_OK;
}

// Measurement aid (bench.py's split_virtual_* lines, scripts/split_virtual_bench.py): with profiling on, every exchange is bracketed by two HIP events on the frame's
// stream (pack + transport + scatter: with virtual ranks the device copies that stand in for the wire) and the bytes arriving at the busiest local rank are summed.
// kj_split_profile waits for the recorded events and reports the totals since profiling was switched on.
KjStatus kj_split_set_profiling(KjSplit* s, uint32_t enable) {
    KJ_REQUIRE(s, "null argument");
    for (auto& p : s->ev_pending) { s->ev_free.push_back(p.first); s->ev_free.push_back(p.second); }
    s->ev_pending.clear();
    s->profiling = enable != 0; s->exchange_ms = 0.0; s->exchange_bytes = 0; s->exchange_points = 0; s->frames_profiled = 0;
    return KJ_OK;
}
KjStatus kj_split_profile(KjSplit* s, KjSplitProfile* out) {
    KJ_REQUIRE(s && out, "null argument");
    for (auto& p : s->ev_pending) {
        KJ_TRY_HIP(hipEventSynchronize(p.second));
        float ms = 0.0f;
        KJ_TRY_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        s->exchange_ms += double(ms);
        s->ev_free.push_back(p.first); s->ev_free.push_back(p.second);
    }
    s->ev_pending.clear();
    out->exchange_ms = s->exchange_ms; out->exchange_bytes_busiest_rank = s->exchange_bytes; out->exchange_points = s->exchange_points; out->gi_frames = s->frames_profiled;
    return KJ_OK;
}

// RCCL bootstrap without a link-time dependency: rank 0 makes the id, the caller broadcasts its 128 bytes by whatever means it has
// (torch.distributed, MPI, a file), every rank turns it into a communicator.
KjStatus kj_split_rccl_unique_id(uint8_t out_id[128]) {
    KJ_REQUIRE(out_id, "null argument");
    KJ_REQUIRE(g_rccl.load(), "librccl.so could not be loaded");
    KJ_REQUIRE(g_rccl.GetUniqueId(out_id) == 0, "ncclGetUniqueId failed");
    return KJ_OK;
}
KjStatus kj_split_rccl_comm_create(const uint8_t id[128], uint32_t world, uint32_t rank, void** out_comm) {
    KJ_REQUIRE(id && out_comm && rank < world, "bad argument");
    KJ_REQUIRE(g_rccl.load(), "librccl.so could not be loaded");
    Rccl::Id128 v; memcpy(v.b, id, 128);
    KJ_REQUIRE(g_rccl.CommInitRank(out_comm, int(world), v, int(rank)) == 0, "ncclCommInitRank failed");
    return KJ_OK;
}
// What the communicator itself says about its size and this process' place in it (ncclCommCount / ncclCommUserRank).
KjStatus kj_split_rccl_comm_info(void* comm, uint32_t* out_ranks, uint32_t* out_rank) {
    KJ_REQUIRE(comm && out_ranks && out_rank, "null argument");
    KJ_REQUIRE(g_rccl.load() && g_rccl.CommCount && g_rccl.CommUserRank, "librccl.so could not be loaded");
    int n = 0, r = 0;
    KJ_REQUIRE(g_rccl.CommCount(comm, &n) == 0 && g_rccl.CommUserRank(comm, &r) == 0, "ncclCommCount / ncclCommUserRank failed");
    *out_ranks = uint32_t(n); *out_rank = uint32_t(r);
    return KJ_OK;
}
void kj_split_rccl_comm_destroy(void* comm) { if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }

}  // extern "C"
