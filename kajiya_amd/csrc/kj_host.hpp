// Host-side objects behind the C-ABI handles (include/kajiya_amd.h).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../../include/kajiya_amd.h"
#include "kj_scene_types.hpp"
#include <cstdlib>

// MEASUREMENT SWITCHES (A/B runs of bench.py and the scripts under scripts/): environment variables that change what the library runs are honoured only when
// KJ_DEBUG_ENV=1 is set as well; a set switch without the gate is ignored with one line on stderr. What a host may configure goes through the C-ABI
// (kj_*_set_*); the ungated variables left are diagnostics (KJ_BVH_TIMING, KJ_SCENE_DEBUG), the host build's thread count (KJ_BVH_THREADS) and the RCCL library path (KJ_RCCL_LIB).
inline const char* kj_debug_getenv(const char* name) {
    static const bool gate = [] { const char* g = getenv("KJ_DEBUG_ENV"); return g && atoi(g) != 0; }();
    const char* v = getenv(name);
    if (v && !gate) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "kajiya_amd: %s is set but ignored: measurement switches need KJ_DEBUG_ENV=1 (this line is printed once)\n", name); warned = true; }
        return nullptr;
    }
    return v;
}

namespace kj {

void set_last_error(const char* fmt, ...);

#define KJ_TRY_HIP(expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            kj::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return KJ_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)
#define KJ_REQUIRE(cond, msg)                                            \
    do {                                                                 \
        if (!(cond)) {                                                   \
            kj::set_last_error("%s:%d: %s", __FILE__, __LINE__, msg);    \
            return KJ_ERR_INVALID_ARGUMENT;                              \
        }                                                                \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    // (re)allocate, zero-filled
    hipError_t alloc(size_t n, hipStream_t s = nullptr) {
        if (n == bytes && p) return hipSuccess;
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return e; }
        bytes = n;
        return hipMemsetAsync(p, 0, n, s);
    }
    hipError_t upload(const void* src, size_t n, hipStream_t s = nullptr) {
        hipError_t e = alloc(n, s);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, s);
    }
};

struct LbvhScratch;     // kj_scene_device.hpp

// Many contiguous device-to-device copies in ONE launch (device.hip: k_copy_blocks): the pack and the scatter of a halo exchange (split.cpp) are dozens of row
// blocks of a few hundred KB each -- as hipMemcpyAsync calls each is a launch of its own (~10 us apiece on the device, 128 per rank and frame: round 6).
struct CopyBlock { const void* src; void* dst; uint64_t bytes; };
hipError_t launch_copy_blocks(const CopyBlock* blocks, size_t n, hipStream_t s);

} // namespace kj

struct KjDevice {
    int ordinal = 0;
    kj::DevBuf blue_noise;      // 256x256 RGBA8
    kj::DevBuf brdf_fg_lut;     // 64x64 RGBA16F
    kj::DevBuf frame_constants; // ring of kj::FrameBlock (the caller's KjFrameConstants + the per-frame products derived from it, kj_screen.hpp)
    uint32_t fc_slot = 0;
    static const uint32_t FC_RING = 16;
    KjFrameConstants fc_host{};           // last uploaded
    const KjFrameConstants* fc_dev = nullptr;
    kj::DevBuf sun_color;                 // float4: SUN_COLOR hoisted per frame (inc/sun.hlsl:21-33)
    uint32_t num_cus = 256;
};

struct KjScene {
    KjDevice* dev = nullptr;
    // host-side tables (the reference keeps these in one 1 GiB vertex buffer + mesh buffer)
    std::vector<uint8_t> vertex_buffer;
    std::vector<kj::GpuMesh> meshes;
    std::vector<std::vector<KjTriangleLight>> mesh_lights;
    struct Inst { uint32_t mesh; float xform[12]; float emissive_multiplier; bool alive; };
    std::vector<Inst> instances;
    std::vector<kj::MapDesc> maps;      // one per material map
    std::vector<uint8_t> tex_data;      // RGBA8 mip chains of the image maps
    // per-mesh acceleration structure (BLAS), built once: node topology (+ parent links) and object-space triangles in leaf order, appended to the shared arrays
    struct Blas { uint32_t node_base = 0, node_count = 0, tri_base = 0, tri_count = 0, max_stack = 1; float bounds[6] = {0, 0, 0, 0, 0, 0}; bool built = false;
                  uint32_t root = 0, heights_base = 0, height_count = 0, wide_heights = 0; };   // root node (relative); the refit's bottom-up steps (kj_scene_device.hpp: InstanceRefitJob)
    std::vector<Blas> blas;                       // one per mesh
    // The top levels of every BLAS as the host sees them: node (relative to the mesh's first node), its object-space box (decoded from its parent's
    // quantised child slot: conservative) and its inner children. What kj_scene_commit opens into the top tree instead of whole instances.
    struct BlasTopNode { uint32_t node; float box[6]; uint32_t first_child, child_count; };      // child_count 0: has a leaf child (or lies below the kept levels): not opened
    std::vector<std::vector<BlasTopNode>> blas_top;   // one list per mesh, [0] = the root
    bool open_instances = false;                  // kj_scene_set_open_instances / KJ_SCENE_OPEN_INSTANCES=1: top-tree leaves are nodes of the instances' top levels instead of whole instances
    uint32_t blas_nodes_used = 0, obj_tris_used = 0;   // fill of the two device pools every BLAS lives in (d_blas_nodes, d_obj_tris)
    uint32_t top_build_mode = 0;                  // kj_scene_set_top_build_mode: 0 = host SAH below KJ_TOP_DEVICE_MIN_LEAVES leaves and a device LBVH from there on, 1 = host, 2 = device
    bool top_built_on_device = false;             // how the last commit built it (kj_scene_stats)
    kj::DevBuf d_top_boxes, d_top_refs;           // the device build's inputs
    kj::LbvhScratch* top_scratch = nullptr;       // ... and its working set, kept across commits
    uint32_t blas_build_mode = 0;                 // for meshes added from now on: 0 = SAH on the host (fast trace), 1 = LBVH on the device (fast build), 2 = PLOC on the device
    std::vector<uint8_t> mesh_build_mode;         // per mesh
    // what changed since the last commit
    bool meshes_dirty = true, instance_set_dirty = true;
    bool instances_added = true;                  // since the last commit: the world arrays must be laid out anew (a removal alone leaves a hole instead)
    std::vector<uint8_t> xform_dirty;             // per instance slot
    // committed device state
    bool committed = false, committed_once = false;
    kj::DevBuf d_vertex_buffer, d_meshes, d_instances, d_maps, d_tex_data, d_lights, d_blas_nodes, d_obj_tris, d_tris, d_jobs, d_refit_jobs;
    kj::DevBuf d_nodes, d_node_boxes;               // the world-space tree (top tree + one region per instance), the refit's per-node scratch box
    kj::DevBuf d_blas_steps;                        // per mesh: {first, end} node of every step of its instances' bottom-up refit
    std::vector<uint32_t> blas_steps;               // host copy of d_blas_steps
    std::vector<uint32_t> inst_node_base;         // per instance slot: first node of its region in d_nodes
    uint32_t tlas_capacity = 0, world_nodes = 0;   // nodes reserved for the top tree at the front of d_nodes; nodes in use overall
    std::vector<uint32_t> inst_id_base;           // per instance slot: first world triangle ID (dense over the live instances)
    kj::DevBuf d_renumber_jobs;
    std::vector<uint32_t> inst_tri_base;          // per instance slot: first world triangle (valid for live instances after a commit)
    uint32_t live_tri_count = 0;                  // triangles of the live instances (tri_count = size of the world array: holes of removed instances included)
    uint32_t light_count = 0, tri_count = 0, node_count = 0, tlas_node_count = 0, bvh_root = 0, bvh_max_depth = 0;
    double last_commit_ms[4] = {0, 0, 0, 0};      // host time of the last commit: BLAS builds, instance tables + top tree, uploads + device transform / refit, total
    kj::SceneView view() const;
};
