// TEST INFRASTRUCTURE (oracle/_ref): run time behind hlsl_compat.hpp -- the pass registry, the dispatch loop, a cooperative lane
// scheduler (ucontext) that gives wave intrinsics and group barriers lock-step meaning, and the C entry points tests/ref_hlsl.py binds.
// Nothing of the product links this.
#include "hlsl_compat.hpp"
#include <ucontext.h>
#include <vector>
#include <string>
#include <map>

namespace hlsl {

struct ResourceSlot { std::string name, type; ResourceBase* res; int binding, set; };
struct ConstantSlot { std::string name; void* ptr; size_t bytes; int binding, set; };
struct PassInfo { std::string name_s; const char* name = ""; uint nt[3] = {1, 1, 1}; bool needs_lockstep = false; void (*invoke)(const LaneInfo&) = nullptr;
    std::vector<ResourceSlot> resources; std::vector<ConstantSlot> constants; };
static std::vector<PassInfo*>& registry() { static std::vector<PassInfo*> r; return r; }
static PassInfo*& open_pass() { static PassInfo* p = nullptr; return p; }
void hlsl_pass_begin(const char* name) { PassInfo* p = new PassInfo; p->name_s = name; p->name = p->name_s.c_str(); open_pass() = p; }
void hlsl_pass_end(const uint nt[3], bool lockstep, void (*invoke)(const LaneInfo&)) {
    PassInfo* p = open_pass(); for (int i = 0; i < 3; ++i) p->nt[i] = nt[i]; p->needs_lockstep = lockstep; p->invoke = invoke; registry().push_back(p); open_pass() = nullptr; }
void hlsl_register_resource(const ResName& n, ResourceBase* r) { if (open_pass()) open_pass()->resources.push_back(ResourceSlot{n.name, n.type, r, n.binding, n.set}); }
void hlsl_register_constant(const char* name, void* ptr, size_t bytes, int binding, int set) { if (open_pass()) open_pass()->constants.push_back(ConstantSlot{name, ptr, bytes, binding, set}); }

// ------------------------------------------------------------------------------------------------ ray tracing hooks
static TraceHook g_trace_hook = nullptr; static void* g_trace_user = nullptr;
const RtPipeline*& hlsl_rt_pipeline() { static thread_local const RtPipeline* p = nullptr; return p; }
RtHitContext*& hlsl_rt_hit() { static thread_local RtHitContext* h = nullptr; return h; }
void hlsl_trace(const float* ray8, uint flags, RayHitInfo* out) {
    if (!g_trace_hook) { fprintf(stderr, "hlsl_compat: TraceRay without a trace hook (ref_set_trace_hook)\n"); abort(); }
    g_trace_hook(g_trace_user, ray8, flags, out);
}

// ------------------------------------------------------------------------------------------------ lanes
enum LaneState { LANE_READY = 0, LANE_WAIT_WAVE = 1, LANE_WAIT_GROUP = 2, LANE_DONE = 3 };
struct Lane { ucontext_t ctx; LaneInfo info; int state; };
struct GroupRun {
    std::vector<Lane> lanes; std::vector<std::vector<uint8_t>> stacks;
    ucontext_t sched; int current = -1; const PassInfo* pass = nullptr;
    // per wave: two slot buffers (alternating per publish), which lanes took part in the round a buffer holds
    std::vector<uint8_t> slots[2]; std::vector<uint8_t> active[2]; std::vector<uint8_t> wave_round;   // per wave: parity of the NEXT publish
};
static thread_local GroupRun* g_run = nullptr;
static thread_local LaneInfo g_plain_lane;          // the lane of a pass that runs without the scheduler

const LaneInfo& hlsl_lane() { return (g_run && g_run->current >= 0) ? g_run->lanes[size_t(g_run->current)].info : g_plain_lane; }

static void lane_yield(int state) {
    GroupRun* r = g_run; Lane& l = r->lanes[size_t(r->current)];
    l.state = state; swapcontext(&l.ctx, &r->sched);
}
WaveView hlsl_wave_publish(const void* value, size_t bytes) {
    GroupRun* r = g_run;
    if (!r || r->current < 0) { fprintf(stderr, "hlsl_compat: wave intrinsic in a pass dispatched without lock-step (%s)\n", r && r->pass ? r->pass->name : "?"); abort(); }
    const uint gi = uint(r->current), wave = gi / HLSL_WAVE, lane = gi % HLSL_WAVE;
    const int par = r->wave_round[wave] & 1;      // every running lane of the wave sees the same parity: the scheduler flips it on release
    memcpy(r->slots[par].data() + (size_t(wave) * HLSL_WAVE + lane) * HLSL_WAVE_SLOT, value, bytes);
    r->active[par][size_t(wave) * HLSL_WAVE + lane] = 1;
    lane_yield(LANE_WAIT_WAVE);
    return WaveView{r->slots[par].data() + size_t(wave) * HLSL_WAVE * HLSL_WAVE_SLOT, r->active[par].data() + size_t(wave) * HLSL_WAVE, lane};
}
void hlsl_group_barrier() {
    if (!g_run || g_run->current < 0) { fprintf(stderr, "hlsl_compat: group barrier in a pass dispatched without lock-step\n"); abort(); }
    lane_yield(LANE_WAIT_GROUP);
}
static void lane_entry() {
    GroupRun* r = g_run; Lane& l = r->lanes[size_t(r->current)];
    r->pass->invoke(l.info);
    l.state = LANE_DONE;
    swapcontext(&l.ctx, &r->sched);
}

static void run_group_lockstep(const PassInfo* p, GroupRun& r, uint3 group_id) {
    const uint nx = p->nt[0], ny = p->nt[1], nz = p->nt[2], n = nx * ny * nz, waves = (n + HLSL_WAVE - 1) / HLSL_WAVE;
    const size_t STACK = 512 * 1024;
    if (r.lanes.size() != n) {
        r.lanes.resize(n); r.stacks.resize(n);
        for (auto& s : r.stacks) s.resize(STACK);
        for (int b = 0; b < 2; ++b) { r.slots[b].assign(size_t(waves) * HLSL_WAVE * HLSL_WAVE_SLOT, 0); r.active[b].assign(size_t(waves) * HLSL_WAVE, 0); }
        r.wave_round.assign(waves, 0);
    }
    for (int b = 0; b < 2; ++b) std::fill(r.active[b].begin(), r.active[b].end(), 0);
    std::fill(r.wave_round.begin(), r.wave_round.end(), 0);
    r.pass = p; g_run = &r;
    for (uint i = 0; i < n; ++i) {
        Lane& l = r.lanes[i];
        const uint tx = i % nx, ty = (i / nx) % ny, tz = i / (nx * ny);
        l.info.group_thread_id = uint3(tx, ty, tz); l.info.group_id = group_id; l.info.group_index = i;
        l.info.dispatch_thread_id = uint3(group_id.x * nx + tx, group_id.y * ny + ty, group_id.z * nz + tz);
        l.state = LANE_READY;
        getcontext(&l.ctx); l.ctx.uc_stack.ss_sp = r.stacks[i].data(); l.ctx.uc_stack.ss_size = STACK; l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, lane_entry, 0);
    }
    for (;;) {
        bool ran = false;
        for (uint i = 0; i < n; ++i) if (r.lanes[i].state == LANE_READY) { r.current = int(i); swapcontext(&r.sched, &r.lanes[i].ctx); ran = true; }
        r.current = -1;
        // release waves whose running lanes all wait at a wave meeting point
        bool released = false;
        for (uint w = 0; w < waves; ++w) {
            bool any_wait = false, all = true;
            for (uint l = w * HLSL_WAVE; l < std::min(n, (w + 1) * HLSL_WAVE); ++l) { const int s = r.lanes[l].state; if (s == LANE_WAIT_WAVE) any_wait = true; else if (s != LANE_DONE) all = false; }
            if (any_wait && all) {
                const int next = (r.wave_round[w] + 1) & 1;
                std::fill(r.active[next].begin() + size_t(w) * HLSL_WAVE, r.active[next].begin() + size_t(w + 1) * HLSL_WAVE, 0);   // the buffer the NEXT publish fills
                r.wave_round[w] = uint8_t(next);
                for (uint l = w * HLSL_WAVE; l < std::min(n, (w + 1) * HLSL_WAVE); ++l) if (r.lanes[l].state == LANE_WAIT_WAVE) r.lanes[l].state = LANE_READY;
                released = true;
            }
        }
        if (released) continue;
        bool any_group = false, all_group = true, all_done = true;
        for (uint i = 0; i < n; ++i) { const int s = r.lanes[i].state; if (s != LANE_DONE) all_done = false; if (s == LANE_WAIT_GROUP) any_group = true; else if (s != LANE_DONE) all_group = false; }
        if (all_done) break;
        if (any_group && all_group) { for (uint i = 0; i < n; ++i) if (r.lanes[i].state == LANE_WAIT_GROUP) r.lanes[i].state = LANE_READY; continue; }
        if (!ran) { fprintf(stderr, "hlsl_compat: lanes of %s diverged at a wave / group meeting point\n", p->name); abort(); }
    }
    g_run = nullptr;
}

// Scheduling choice for passes without lock-step needs: group by group (default), or every thread of the dispatch in ascending
// (z, y, x) order. Any order is a legal schedule; passes that push onto shared lists with atomics (the irradiance cache's free list)
// produce those lists in schedule order, and the oracle's loops are the ascending one.
static bool g_linear_order = false;
static uint3 g_dispatch_dims;
uint3 hlsl_dispatch_dims() { return g_dispatch_dims; }
static void dispatch(const PassInfo* p, uint tx, uint ty, uint tz) {
    g_dispatch_dims = uint3(tx, ty, tz);
    const uint gx = (tx + p->nt[0] - 1) / p->nt[0], gy = (ty + p->nt[1] - 1) / p->nt[1], gz = (tz + p->nt[2] - 1) / p->nt[2];
    if (!p->needs_lockstep && g_linear_order) {
        for (uint z = 0; z < gz * p->nt[2]; ++z) for (uint y = 0; y < gy * p->nt[1]; ++y) for (uint x = 0; x < gx * p->nt[0]; ++x) {
            LaneInfo& l = g_plain_lane;
            l.group_id = uint3(x / p->nt[0], y / p->nt[1], z / p->nt[2]); l.group_thread_id = uint3(x % p->nt[0], y % p->nt[1], z % p->nt[2]);
            l.group_index = l.group_thread_id.x + p->nt[0] * (l.group_thread_id.y + p->nt[1] * l.group_thread_id.z);
            l.dispatch_thread_id = uint3(x, y, z);
            p->invoke(l);
        }
        return;
    }
    if (p->needs_lockstep) {
        GroupRun run;
        for (uint z = 0; z < gz; ++z) for (uint y = 0; y < gy; ++y) for (uint x = 0; x < gx; ++x) run_group_lockstep(p, run, uint3(x, y, z));
        return;
    }
    for (uint z = 0; z < gz; ++z) for (uint y = 0; y < gy; ++y) for (uint x = 0; x < gx; ++x)
        for (uint lz = 0; lz < p->nt[2]; ++lz) for (uint ly = 0; ly < p->nt[1]; ++ly) for (uint lx = 0; lx < p->nt[0]; ++lx) {
            LaneInfo& l = g_plain_lane;
            l.group_thread_id = uint3(lx, ly, lz); l.group_id = uint3(x, y, z); l.group_index = lx + p->nt[0] * (ly + p->nt[1] * lz);
            l.dispatch_thread_id = uint3(x * p->nt[0] + lx, y * p->nt[1] + ly, z * p->nt[2] + lz);
            p->invoke(l);
        }
}

}  // namespace hlsl

using namespace hlsl;
static PassInfo* find(const char* name) { for (PassInfo* p : registry()) if (!strcmp(p->name, name)) return p; return nullptr; }

extern "C" {
int ref_pass_count() { return int(registry().size()); }
const char* ref_pass_name(int i) { return registry()[size_t(i)]->name; }
int ref_pass_exists(const char* pass) { return find(pass) != nullptr; }
int ref_pass_resource_count(const char* pass) { const PassInfo* p = find(pass); return p ? int(p->resources.size()) : -1; }
const char* ref_pass_resource_name(const char* pass, int i) { return find(pass)->resources[size_t(i)].name.c_str(); }
const char* ref_pass_resource_type(const char* pass, int i) { return find(pass)->resources[size_t(i)].type.c_str(); }
int ref_pass_resource_binding(const char* pass, int i) { return find(pass)->resources[size_t(i)].binding; }
int ref_pass_resource_set(const char* pass, int i) { return find(pass)->resources[size_t(i)].set; }
int ref_pass_constant_binding(const char* pass, int i) { return find(pass)->constants[size_t(i)].binding; }
int ref_pass_constant_set(const char* pass, int i) { return find(pass)->constants[size_t(i)].set; }
int ref_pass_constant_count(const char* pass) { const PassInfo* p = find(pass); return p ? int(p->constants.size()) : -1; }
const char* ref_pass_constant_name(const char* pass, int i) { return find(pass)->constants[size_t(i)].name.c_str(); }
int ref_pass_constant_bytes(const char* pass, int i) { return int(find(pass)->constants[size_t(i)].bytes); }
void ref_pass_numthreads(const char* pass, unsigned* out3) { const PassInfo* p = find(pass); for (int i = 0; i < 3; ++i) out3[i] = p->nt[i]; }

// bind memory to a resource of the pass: textures give (w, h, format), buffers give bytes (w = h = 0)
int ref_bind(const char* pass, const char* name, void* data, int w, int h, int fmt, unsigned long long bytes) {
    const PassInfo* p = find(pass); if (!p) return -1;
    int found = 0;      // every resource of that name: a hit shader compiled into the same wrapper declares its own copy of some set-1 / set-2 resources
    for (size_t i = 0; i < p->resources.size(); ++i) if (p->resources[i].name == name) {
        ResourceBase* r = p->resources[i].res; r->data = data; r->w = w; r->h = h; r->fmt = fmt;
        r->bytes = bytes ? size_t(bytes) : size_t(w) * size_t(h) * size_t(format_bytes(fmt)); ++found; }
    return found ? 0 : -2;
}
// one slot of a resource array (`Texture2D bindless_textures[]`)
int ref_bind_slot(const char* pass, const char* name, unsigned index, void* data, int w, int h, int fmt) {
    PassInfo* p = find(pass); if (!p) return -1;
    int found = 0;
    for (size_t i = 0; i < p->resources.size(); ++i) if (p->resources[i].name == name) {
        ResourceArrayBase* a = dynamic_cast<ResourceArrayBase*>(p->resources[i].res); if (!a) return -4;
        ResourceBase* r = a->slot(index); if (!r) return -5;
        r->data = data; r->w = w; r->h = h; r->fmt = fmt; r->bytes = size_t(w) * size_t(h) * size_t(format_bytes(fmt)); ++found; }
    return found ? 0 : -2;
}
int ref_set_constant(const char* pass, const char* name, const void* src, unsigned long long bytes) {
    const PassInfo* p = find(pass); if (!p) return -1;
    int found = 0;
    for (size_t i = 0; i < p->constants.size(); ++i) if (p->constants[i].name == name) {
        if (bytes != p->constants[i].bytes) return -3;
        memcpy(p->constants[i].ptr, src, size_t(bytes)); ++found; }
    return found ? 0 : -2;
}
// `threads`: the extent kajiya's .dispatch([x, y, z]) is given -- threads, rounded up to whole groups like the backend does
void ref_set_trace_hook(void* fn, void* user) { g_trace_hook = (TraceHook)fn; g_trace_user = user; }
void ref_set_linear_order(int on) { g_linear_order = on != 0; }
int ref_dispatch(const char* pass, unsigned tx, unsigned ty, unsigned tz) { const PassInfo* p = find(pass); if (!p) return -1; dispatch(p, tx, ty, tz); return 0; }
}
