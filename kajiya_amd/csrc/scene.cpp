// KjScene: WorldRenderer's scene state for the GI path (world_renderer.rs:604-911): mesh upload into one byte-addressed
// vertex buffer + GpuMesh table, instances, triangle lights, and -- in the role of the driver's acceleration structures
// (add_mesh -> BLAS, build_ray_tracing_top_level_acceleration / prepare_top_level_acceleration -> TLAS:
// world_renderer.rs:694-724,836-911) -- a two-level BVH:
//   * one BLAS per mesh in object space, built once (4-wide SAH tree with quantised child boxes, bvh_build.cpp; or, for meshes
//     added with KJ_MESH_BUILD_PREFER_FAST_BUILD, an LBVH built on the device, scene_device.hip), shared by the mesh's instances;
//   * per instance, its triangles in world space (device kernel, same fp32 arithmetic as the oracle's flattening);
//   * a TLAS over the live instances, rebuilt at every commit.
// A commit after set_instance_transform re-derives that instance's world triangles and the TLAS: no mesh is rebuilt.
// Plain C++ (no device code); compiled with -ffp-contract=off.
#include "kj_host.hpp"
#include "kj_bvh_build.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <numeric>
#include <future>
#include <thread>
#include <map>
#include <memory>
#include <chrono>
#include <cstdlib>
#include "kj_scene_device.hpp"

namespace kj {

static thread_local char g_last_error[1024] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

SceneView scene_view(const KjScene& s) {
    SceneView v{};
    v.vertex_buffer = (const uint8_t*)s.d_vertex_buffer.p;
    v.meshes = (const GpuMesh*)s.d_meshes.p;
    v.instances = (const GpuInstance*)s.d_instances.p;
    v.maps = (const MapDesc*)s.d_maps.p;
    v.tex_data = (const uint8_t*)s.d_tex_data.p;
    v.lights = (const KjTriangleLight*)s.d_lights.p;
    v.light_count = s.light_count;
    v.bvh.nodes = (const F4*)s.d_nodes.p;
    v.bvh.tris = (const F4*)s.d_tris.p;
    v.bvh.root = s.bvh_root;
    v.bvh.stack_entries = KJ_BVH_LDS_STACK;   // LDS part of the traversal stack (deeper entries spill, kj_bvh.hpp)
    return v;
}

} // namespace kj

kj::SceneView KjScene::view() const { return kj::scene_view(*this); }

using namespace kj;

template <typename T> static uint32_t vb_append(std::vector<uint8_t>& vb, const T* data, size_t count) {
    size_t off = (vb.size() + 63) & ~size_t(63);
    vb.resize(off + sizeof(T) * count);
    if (count) memcpy(vb.data() + off, data, sizeof(T) * count);
    return uint32_t(off);
}

extern "C" {

const char* kj_last_error(void) { return kj::g_last_error; }
uint32_t kj_abi_version(void) { return 1; }
uint32_t kj_abi_struct_size(uint32_t id) {
    static const uint32_t sizes[KJ_ABI_STRUCT_COUNT] = {
        sizeof(KjFrameConstants), sizeof(KjViewConstants), sizeof(KjMeshMaterial), sizeof(KjPackedVertex), sizeof(KjMaterialMap), sizeof(KjMeshDesc), sizeof(KjTriangleLight),
        sizeof(KjGbufferDepth), sizeof(KjRtdgiRenderParams), sizeof(KjRtdgiOutput), sizeof(KjTaaOutput), sizeof(KjRtrTables), sizeof(KjRtrParams), sizeof(KjSplitRank),
        sizeof(KjSplitFrame), sizeof(KjBakedMeshView), sizeof(KjBakedImageView), sizeof(KjSplitProfile)};
    return id < KJ_ABI_STRUCT_COUNT ? sizes[id] : 0u;
}

KjStatus kj_scene_create(KjDevice* dev, KjScene** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjScene* s = new KjScene();
    s->dev = dev;
    s->vertex_buffer.resize(64, 0);  // offset 0 stays unused: `vertex_aux_offset != 0` means "has colours"
    *out = s;
    return KJ_OK;
}
void kj_scene_destroy(KjScene* scene) {
    if (scene) delete scene->top_scratch;
    delete scene;
}

KjStatus kj_scene_add_mesh(KjScene* s, const KjMeshDesc* d, uint32_t* out_mesh) {
    KJ_REQUIRE(s && d && out_mesh, "null argument");
    KJ_REQUIRE(d->verts && d->indices && d->materials && d->maps, "mesh streams missing");
    KJ_REQUIRE(d->index_count >= 3 && d->index_count % 3 == 0, "mesh must not be empty (world_renderer.rs:706-711)");
    for (uint32_t i = 0; i < d->map_count; ++i)
        if (d->maps[i].image_rgba8) {
            const KjMaterialMap& mm = d->maps[i];
            KJ_REQUIRE(mm.width >= 1 && mm.height >= 1 && mm.width <= 16384 && mm.height <= 16384, "image map extent out of range");
            uint32_t full = 1;
            while ((mm.width >> full) || (mm.height >> full)) ++full;
            KJ_REQUIRE(mm.mip_count >= 1 && mm.mip_count <= full, "image map mip_count out of range");
        }
    for (uint32_t i = 0; i < d->index_count; ++i) KJ_REQUIRE(d->indices[i] < d->vertex_count, "index out of range");
    // every check before the scene is touched: a rejected mesh must not leave maps / texels / vertex data behind
    for (uint32_t i = 0; i < d->material_count; ++i)
        for (int k = 0; k < 4; ++k) KJ_REQUIRE(d->materials[i].maps[k] < d->map_count, "material map index out of range");
    if (d->material_ids)
        for (uint32_t i = 0; i < d->vertex_count; ++i) KJ_REQUIRE(d->material_ids[i] < d->material_count, "material id out of range");
    else KJ_REQUIRE(d->material_count >= 1, "mesh has no material");
    {
        size_t tex_bytes = s->tex_data.size();
        for (uint32_t i = 0; i < d->map_count; ++i)
            if (d->maps[i].image_rgba8) {
                tex_bytes = (tex_bytes + 15) & ~size_t(15);
                for (uint32_t k = 0; k < d->maps[i].mip_count; ++k) tex_bytes += size_t(std::max(1u, d->maps[i].width >> k)) * std::max(1u, d->maps[i].height >> k) * 4;
            }
        KJ_REQUIRE(tex_bytes < (size_t(1) << 32), "more than 4 GiB of material maps");
        const size_t vb_bytes = s->vertex_buffer.size() + size_t(d->index_count) * 4 + size_t(d->vertex_count) * (16 + 8 + 4 + 16 + 16) + size_t(d->material_count) * sizeof(KjMeshMaterial) + 8 * 64;
        KJ_REQUIRE(vb_bytes < (size_t(1) << 32), "vertex buffer would exceed 4 GiB (32-bit offsets)");
    }
    std::vector<KjMeshMaterial> mats(d->materials, d->materials + d->material_count);
    const uint32_t map_base = uint32_t(s->maps.size());
    for (uint32_t i = 0; i < d->map_count; ++i) {
        const KjMaterialMap& mm = d->maps[i];
        MapDesc md{};
        md.color = F4{float(mm.placeholder_rgba[0]) / 255.0f, float(mm.placeholder_rgba[1]) / 255.0f, float(mm.placeholder_rgba[2]) / 255.0f, float(mm.placeholder_rgba[3]) / 255.0f};
        if (mm.image_rgba8) {
            size_t bytes = 0;
            for (uint32_t k = 0; k < mm.mip_count; ++k) bytes += size_t(std::max(1u, mm.width >> k)) * std::max(1u, mm.height >> k) * 4;
            while (s->tex_data.size() % 16) s->tex_data.push_back(0);
            md.offset = uint32_t(s->tex_data.size());
            md.width = mm.width; md.height = mm.height;
            md.flags = mm.mip_count | (mm.srgb ? 0x100u : 0u);
            s->tex_data.insert(s->tex_data.end(), mm.image_rgba8, mm.image_rgba8 + bytes);
        }
        s->maps.push_back(md);
    }
    for (auto& m : mats) {
        for (int k = 0; k < 4; ++k) m.maps[k] += map_base;
        if (d->use_lights) m.flags |= KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT;
    }
    std::vector<float> uvs(size_t(d->vertex_count) * 2, 0.0f);
    if (d->uvs) memcpy(uvs.data(), d->uvs, uvs.size() * 4);
    std::vector<uint32_t> mids(d->vertex_count, 0);
    if (d->material_ids) memcpy(mids.data(), d->material_ids, size_t(d->vertex_count) * 4);
    GpuMesh m{};
    std::vector<uint8_t>& vb = s->vertex_buffer;
    m.index_offset = vb_append(vb, d->indices, d->index_count);
    m.vertex_core_offset = vb_append(vb, d->verts, d->vertex_count);
    m.vertex_uv_offset = vb_append(vb, uvs.data(), uvs.size());
    m.vertex_mat_offset = vb_append(vb, mids.data(), mids.size());
    m.vertex_aux_offset = d->colors ? vb_append(vb, d->colors, size_t(d->vertex_count) * 4) : 0;
    m.vertex_tangent_offset = d->tangents ? vb_append(vb, d->tangents, size_t(d->vertex_count) * 4) : 0;
    m.mat_data_offset = vb_append(vb, mats.data(), mats.size());
    m.index_count = d->index_count;
    s->meshes.push_back(m);
    // emissive triangles -> TriangleLight list (world_renderer.rs:741-773)
    std::vector<KjTriangleLight> lights;
    if (d->use_lights) {
        for (uint32_t i = 0; i + 2 < d->index_count; i += 3) {
            const KjMeshMaterial& mat = d->materials[mids[d->indices[i]]];
            if (!(mat.emissive[0] > 0 || mat.emissive[1] > 0 || mat.emissive[2] > 0)) continue;
            KjTriangleLight l;
            for (int k = 0; k < 3; ++k) memcpy(&l.verts[k * 3], d->verts[d->indices[i + k]].pos, 12);
            memcpy(l.radiance, mat.emissive, 12);
            lights.push_back(l);
        }
    }
    s->mesh_lights.push_back(std::move(lights));
    s->blas.emplace_back();
    s->blas_top.emplace_back();
    s->mesh_build_mode.push_back(uint8_t(s->blas_build_mode));
    s->meshes_dirty = true;
    s->committed = false;
    *out_mesh = uint32_t(s->meshes.size() - 1);
    return KJ_OK;
}

KjStatus kj_scene_add_instance(KjScene* s, uint32_t mesh, const float* xf, uint32_t* out_instance) {
    KJ_REQUIRE(s && xf && out_instance, "null argument");
    KJ_REQUIRE(mesh < s->meshes.size(), "bad mesh handle");
    KjScene::Inst i{};
    i.mesh = mesh;
    memcpy(i.xform, xf, 48);
    i.emissive_multiplier = 1.0f;
    i.alive = true;
    s->instances.push_back(i);
    s->xform_dirty.push_back(1);
    s->instance_set_dirty = true;
    s->instances_added = true;
    s->committed = false;
    *out_instance = uint32_t(s->instances.size() - 1);
    return KJ_OK;
}
KjStatus kj_scene_set_instance_transform(KjScene* s, uint32_t instance, const float* xf) {
    KJ_REQUIRE(s && xf && instance < s->instances.size() && s->instances[instance].alive, "bad instance handle");
    memcpy(s->instances[instance].xform, xf, 48);
    s->xform_dirty[instance] = 1;
    s->committed = false;
    return KJ_OK;
}
KjStatus kj_scene_set_instance_emissive_multiplier(KjScene* s, uint32_t instance, float v) {
    KJ_REQUIRE(s && instance < s->instances.size() && s->instances[instance].alive, "bad instance handle");
    s->instances[instance].emissive_multiplier = v;
    s->committed = false;
    return KJ_OK;
}
KjStatus kj_scene_remove_instance(KjScene* s, uint32_t instance) {
    KJ_REQUIRE(s && instance < s->instances.size() && s->instances[instance].alive, "bad instance handle");
    s->instances[instance].alive = false;
    s->instance_set_dirty = true;
    s->committed = false;
    return KJ_OK;
}

// nodes of a BLAS' top levels kept on the host for the top-tree build (root + up to four levels below it)
#define KJ_BLAS_TOP_NODES 341u
// from this many top-tree leaves on, the commit builds the top tree on the device (kj_scene_set_top_build_mode overrides). Measured on MI355X (round 6,
// profiles/r06_blas_builds.md): a commit with one moved instance costs 0.14 ms (host) / 0.22 ms (device) at 256 instances, 0.63 / 0.32 ms at 1024, 2.6 / 0.47 ms at 4096,
// 23.6 / 1.7 ms at 32 k; closest-hit rays under the device's linear tree run within 2 % of the host's SAH tree at every count, occlusion rays 0-11 % slower.
#define KJ_TOP_DEVICE_MIN_LEAVES 1024u

// world box of an object-space box under a 3x4 transform (all eight corners), padded for fp32 rounding
static void world_box(const float* x, const float* ob, float* wb) {
    for (int k = 0; k < 3; ++k) { wb[k] = FLT_MAX; wb[3 + k] = -FLT_MAX; }
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {ob[(c & 1) ? 3 : 0], ob[(c & 2) ? 4 : 1], ob[(c & 4) ? 5 : 2]};
        for (int r = 0; r < 3; ++r) {
            const float v = x[r * 4] * p[0] + x[r * 4 + 1] * p[1] + x[r * 4 + 2] * p[2] + x[r * 4 + 3];
            wb[r] = std::min(wb[r], v); wb[3 + r] = std::max(wb[3 + r], v);
        }
    }
}

// The top levels of a BLAS for the commit's top-tree build (KjScene::BlasTopNode): breadth first from the root, `max_nodes` at most. `nodes` holds
// the first `available` nodes of the mesh (child indices absolute: + node_base); a node is "openable" when all its children are inner nodes we hold.
static void extract_blas_top(const BvhNode* nodes, uint32_t available, uint32_t node_base, uint32_t root, const float root_box[6], uint32_t max_nodes,
                             std::vector<KjScene::BlasTopNode>& out) {
    out.clear();
    KjScene::BlasTopNode r{};
    r.node = root; memcpy(r.box, root_box, 24);
    out.push_back(r);
    for (size_t q = 0; q < out.size(); ++q) {
        const uint32_t rel = out[q].node;
        if (rel >= available) continue;
        const BvhNode& n = nodes[rel];
        uint32_t inner = 0; bool openable = true;
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = n.child[i];
            if (c == 0xffffffffu) continue;
            if ((c & KJ_BVH_LEAF) || c - node_base >= available) { openable = false; break; }
            ++inner;
        }
        if (!openable || inner < 2 || out.size() + inner > max_nodes) continue;
        out[q].first_child = uint32_t(out.size()); out[q].child_count = inner;
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = n.child[i];
            if (c == 0xffffffffu) continue;
            KjScene::BlasTopNode t{};
            t.node = c - node_base;
            for (int k = 0; k < 3; ++k) {       // the traversal's own decode: origin + q * 2^(e - 127) (q * step is exact)
                const float step = std::ldexp(1.0f, int(n.exp8[k]) - 127);
                t.box[k] = n.origin[k] + float(n.qlo[k][i]) * step;
                t.box[3 + k] = n.origin[k] + float(n.qhi[k][i]) * step;
            }
            out.push_back(t);
        }
    }
}

// Reorders a BLAS (child indices absolute, root = nodes[0]) by node HEIGHT: 0 = all children are leaves, else 1 + the tallest inner
// child -- the order the per-instance refit works in (scene_device.hip): a height only reads boxes of lower heights, i.e. of lower
// indices. The root ends up LAST. starts[h] = first node of height h (starts.back() = count).
static void blas_sort_by_height(std::vector<BvhNode>& nodes, uint32_t node_base, std::vector<uint32_t>& starts) {
    const uint32_t count = uint32_t(nodes.size());
    std::vector<uint32_t> bfs;
    bfs.reserve(count);
    bfs.push_back(0);
    for (size_t q = 0; q < bfs.size(); ++q)
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = nodes[bfs[q]].child[i];
            if (c != 0xffffffffu && !(c & KJ_BVH_LEAF)) bfs.push_back(c - node_base);
        }
    std::vector<uint32_t> height(count, 0);
    uint32_t tallest = 0;
    for (size_t q = bfs.size(); q-- > 0;) {     // children come after their parent in breadth-first order
        uint32_t h = 0;
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = nodes[bfs[q]].child[i];
            if (c != 0xffffffffu && !(c & KJ_BVH_LEAF)) h = std::max(h, height[c - node_base] + 1u);
        }
        height[bfs[q]] = h;
        tallest = std::max(tallest, h);
    }
    starts.assign(tallest + 2, 0);
    for (uint32_t n : bfs) starts[height[n] + 1]++;
    for (uint32_t h = 0; h <= tallest; ++h) starts[h + 1] += starts[h];
    std::vector<uint32_t> fill(starts.begin(), starts.end() - 1), new_index(count, 0);
    for (uint32_t n : bfs) new_index[n] = fill[height[n]]++;
    std::vector<BvhNode> sorted(bfs.size());     // (nodes unreachable from the root, if a builder left any, are dropped)
    for (uint32_t n : bfs) {
        BvhNode v = nodes[n];
        for (int i = 0; i < 4; ++i)
            if (v.child[i] != 0xffffffffu && !(v.child[i] & KJ_BVH_LEAF)) v.child[i] = node_base + new_index[v.child[i] - node_base];
        sorted[new_index[n]] = v;
    }
    nodes.swap(sorted);
}

KjStatus kj_scene_commit(KjScene* s, void* stream_) {
    KJ_REQUIRE(s, "null scene");
    hipStream_t stream = (hipStream_t)stream_;
    KJ_TRY_HIP(hipSetDevice(s->dev->ordinal));
    typedef std::chrono::steady_clock Clock;
    const auto t0 = Clock::now();
    auto ms_since = [](Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); };
    // 1. static mesh data, then the BLAS of every mesh that does not have one yet (object space; the role of add_mesh's
    //    acceleration-structure build). Both BLAS pools only ever grow: existing meshes keep their place.
    if (s->meshes_dirty) {
        KJ_TRY_HIP(s->d_vertex_buffer.upload(s->vertex_buffer.data(), s->vertex_buffer.size(), stream));
        KJ_TRY_HIP(s->d_meshes.upload(s->meshes.data(), s->meshes.size() * sizeof(GpuMesh), stream));
        KJ_TRY_HIP(s->d_maps.upload(s->maps.data(), s->maps.size() * sizeof(MapDesc), stream));
        if (s->tex_data.empty()) s->tex_data.resize(16, 0);
        KJ_TRY_HIP(s->d_tex_data.upload(s->tex_data.data(), s->tex_data.size(), stream));
    }
    auto grow_pool = [&](kj::DevBuf& pool, size_t used_bytes, size_t need_bytes) -> hipError_t {
        if (need_bytes <= pool.bytes) return hipSuccess;
        kj::DevBuf bigger;
        hipError_t e = bigger.alloc(std::max(need_bytes, pool.bytes + pool.bytes / 2), stream);
        if (e != hipSuccess) return e;
        if (used_bytes) { e = hipMemcpyAsync(bigger.p, pool.p, used_bytes, hipMemcpyDeviceToDevice, stream); if (e != hipSuccess) return e; }
        e = hipStreamSynchronize(stream);
        std::swap(pool.p, bigger.p); std::swap(pool.bytes, bigger.bytes);
        return e;
    };
    LbvhScratch lbvh_scratch;     // device-side builds of this commit share their working buffers
    // Device-side builds (LBVH / PLOC) of ALL new meshes of the commit as one batch per builder (round 6: per-mesh stages one after the other, the 4-wide collapse level
    // by level for all meshes together, one read-back -- lbvh_build.hip). A tree's node count is known only afterwards, so the batch writes every tree with
    // mesh-relative child indices into a scratch area (one node per triangle + 1 reserved) and the loop below moves each into the pool at its final, dense place.
    std::map<uint32_t, LbvhResult> device_built;
    std::map<uint32_t, size_t> device_nodes_at;      // mesh -> first node of its tree in blas_tmp_nodes
    kj::DevBuf blas_tmp_nodes;
    {
        size_t tmp_nodes = 0, tris_after = s->obj_tris_used, nodes_after = s->blas_nodes_used;
        std::vector<uint32_t> tri_base_of(s->meshes.size(), 0u);
        for (uint32_t mi = 0; mi < s->meshes.size(); ++mi) {
            if (s->blas[mi].built) continue;
            const size_t ntri = s->meshes[mi].index_count / 3;
            tri_base_of[mi] = uint32_t(tris_after); tris_after += ntri;
            if (s->mesh_build_mode[mi] != 0) { device_nodes_at[mi] = tmp_nodes; tmp_nodes += ntri + 1; }
        }
        if (tmp_nodes) {
            KJ_TRY_HIP(grow_pool(s->d_obj_tris, size_t(s->obj_tris_used) * sizeof(BvhTri), tris_after * sizeof(BvhTri)));      // the pools grow ONCE per commit, not per mesh
            KJ_TRY_HIP(blas_tmp_nodes.alloc(tmp_nodes * sizeof(BvhNode), stream));
            for (int mode = 1; mode <= 2; ++mode) {
                std::vector<LbvhBatchMesh> batch;
                for (auto& kv : device_nodes_at)
                    if (s->mesh_build_mode[kv.first] == mode)
                        batch.push_back(LbvhBatchMesh{s->meshes[kv.first], 0u, (Bvh4Node*)blas_tmp_nodes.p + kv.second, (BvhTri*)s->d_obj_tris.p + tri_base_of[kv.first], &device_built[kv.first]});
                if (!batch.empty()) KJ_TRY_HIP(build_blas_lbvh_device_batch((const uint8_t*)s->d_vertex_buffer.p, batch.data(), uint32_t(batch.size()), &lbvh_scratch, stream, mode == 2));
            }
            for (auto& kv : device_built) nodes_after += kv.second.node_count;
            KJ_TRY_HIP(grow_pool(s->d_blas_nodes, size_t(s->blas_nodes_used) * sizeof(BvhNode), nodes_after * sizeof(BvhNode)));
        }
    }
    // host SAH builds of all new meshes run concurrently (a small mesh builds on one thread: nine of them one after the other were
    // half of a first commit); each result is the same tree whatever runs beside it
    auto host_build = [s](uint32_t mi) {
        const GpuMesh& m = s->meshes[mi];
        const uint32_t ntri = m.index_count / 3;
        std::vector<BvhTri> ot(ntri);
        for (uint32_t p = 0; p < ntri; ++p) {
            BvhTri& t = ot[p];
            float* dst[3] = {t.v0, t.v1, t.v2};
            for (int k = 0; k < 3; ++k) {
                uint32_t idx;
                memcpy(&idx, s->vertex_buffer.data() + m.index_offset + (p * 3 + k) * 4, 4);
                memcpy(dst[k], s->vertex_buffer.data() + m.vertex_core_offset + size_t(idx) * 16, 12);
            }
            t.world_id = 0; t.inst = 0; t.prim = p;
        }
        std::unique_ptr<BuiltBvh> b(new BuiltBvh());
        build_bvh4(ot, *b);
        return b;
    };
    // at most `hardware_concurrency` builds in flight (each may spawn workers of its own for a large mesh, and holds a copy of its
    // triangles): a first commit with thousands of small meshes must not start thousands of threads. A build that
    // cannot get its thread or its memory is reported as a status, not as an exception through the C boundary.
    std::map<uint32_t, std::future<std::unique_ptr<BuiltBvh>>> host_builds;
    std::vector<uint32_t> to_build;
    for (uint32_t mi = 0; mi < s->meshes.size(); ++mi)
        if (!s->blas[mi].built && s->mesh_build_mode[mi] == 0) to_build.push_back(mi);
    const size_t max_in_flight = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    size_t next_build = 0;
    auto start_builds = [&]() -> bool {
        try {
            while (next_build < to_build.size() && host_builds.size() < max_in_flight) {
                const uint32_t mi = to_build[next_build];
                host_builds[mi] = std::async(std::launch::async, host_build, mi);
                ++next_build;
            }
        } catch (const std::exception&) { return false; }
        return true;
    };
    if (!start_builds() && host_builds.empty()) { set_last_error("could not start a BLAS build thread"); return KJ_ERR_OUT_OF_MEMORY; }
    for (uint32_t mi = 0; mi < s->meshes.size(); ++mi) {
        KjScene::Blas& bl = s->blas[mi];
        if (bl.built) continue;
        const GpuMesh& m = s->meshes[mi];
        const uint32_t ntri = m.index_count / 3;
        bl.node_base = s->blas_nodes_used; bl.tri_base = s->obj_tris_used; bl.tri_count = ntri;
        std::vector<uint32_t> steps;          // {first, end} node of every step of the refit's bottom-up order
        if (s->mesh_build_mode[mi] != 0) {   // built on the device above (LBVH or PLOC): the tree moves from the batch's scratch area to its place in the pool
            const LbvhResult& lr = device_built[mi];
            bl.node_count = lr.node_count; bl.max_stack = lr.max_stack;
            KJ_TRY_HIP(grow_pool(s->d_blas_nodes, size_t(s->blas_nodes_used) * sizeof(BvhNode), size_t(s->blas_nodes_used + bl.node_count) * sizeof(BvhNode)));      // (a no-op unless host builds came in between)
            KJ_TRY_HIP(launch_blas_place_nodes((const Bvh4Node*)blas_tmp_nodes.p + device_nodes_at[mi], (Bvh4Node*)s->d_blas_nodes.p + bl.node_base, bl.node_count, bl.node_base, stream));
            memcpy(bl.bounds, lr.bounds, 24);
            // laid out by depth on the device: the refit walks the levels deepest first, the root is node 0
            for (size_t d = lr.level_starts.size() - 1; d-- > 0;) { steps.push_back(lr.level_starts[d]); steps.push_back(lr.level_starts[d + 1]); }
            bl.root = 0;
            extract_blas_top(lr.head.data(), uint32_t(lr.head.size()), 0u, 0u, bl.bounds, KJ_BLAS_TOP_NODES, s->blas_top[mi]);      // its top levels (mesh-relative child indices), read back with the build's own results
        } else {                              // binned SAH on the host
            if (!host_builds.count(mi)) {      // not started yet (the in-flight limit): start it now, alone if need be
                try { host_builds[mi] = std::async(std::launch::async, host_build, mi); }
                catch (const std::exception&) { host_builds[mi] = std::async(std::launch::deferred, host_build, mi); }
                next_build = std::max(next_build, size_t(std::find(to_build.begin(), to_build.end(), mi) - to_build.begin()) + 1);
            }
            std::unique_ptr<BuiltBvh> built;
            try { built = host_builds[mi].get(); }
            catch (const std::bad_alloc&) { set_last_error("out of host memory building the BLAS of mesh %u", mi); return KJ_ERR_OUT_OF_MEMORY; }
            catch (const std::exception& e) { set_last_error("BLAS build of mesh %u failed: %s", mi, e.what()); return KJ_ERR_OUT_OF_MEMORY; }
            host_builds.erase(mi);
            start_builds();                    // a slot is free: keep the builders busy
            BuiltBvh& b = *built;
            bl.node_count = uint32_t(b.nodes.size());
            bl.max_stack = b.max_stack;
            for (int k = 0; k < 3; ++k) { bl.bounds[k] = FLT_MAX; bl.bounds[3 + k] = -FLT_MAX; }
            for (const BvhTri& t : b.tris)
                for (const float* v : {t.v0, t.v1, t.v2})
                    for (int k = 0; k < 3; ++k) { bl.bounds[k] = std::min(bl.bounds[k], v[k]); bl.bounds[3 + k] = std::max(bl.bounds[3 + k], v[k]); }
            for (BvhNode& n : b.nodes)      // child node indices become absolute in the pool; leaf references stay relative to the mesh
                for (int i = 0; i < 4; ++i)
                    if (n.child[i] != 0xffffffffu && !(n.child[i] & KJ_BVH_LEAF)) n.child[i] += bl.node_base;
            KJ_TRY_HIP(grow_pool(s->d_blas_nodes, size_t(s->blas_nodes_used) * sizeof(BvhNode), size_t(s->blas_nodes_used + bl.node_count) * sizeof(BvhNode)));
            KJ_TRY_HIP(grow_pool(s->d_obj_tris, size_t(s->obj_tris_used) * sizeof(BvhTri), size_t(s->obj_tris_used + ntri) * sizeof(BvhTri)));
            KJ_TRY_HIP(hipMemcpyAsync((BvhTri*)s->d_obj_tris.p + bl.tri_base, b.tris.data(), b.tris.size() * sizeof(BvhTri), hipMemcpyHostToDevice, stream));
            // nodes sorted by height (what the refit of this mesh's instances walks), then into the pool
            std::vector<uint32_t> starts;
            blas_sort_by_height(b.nodes, bl.node_base, starts);
            bl.node_count = uint32_t(b.nodes.size());
            bl.root = bl.node_count - 1u;
            extract_blas_top(b.nodes.data(), bl.node_count, bl.node_base, bl.root, bl.bounds, KJ_BLAS_TOP_NODES, s->blas_top[mi]);
            for (size_t h = 0; h + 1 < starts.size(); ++h) { steps.push_back(starts[h]); steps.push_back(starts[h + 1]); }
            KJ_TRY_HIP(hipMemcpyAsync((BvhNode*)s->d_blas_nodes.p + bl.node_base, b.nodes.data(), b.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, stream));
            KJ_TRY_HIP(hipStreamSynchronize(stream));    // b goes out of scope
        }
        bl.heights_base = uint32_t(s->blas_steps.size() / 2);
        bl.height_count = uint32_t(steps.size() / 2);
        bl.wide_heights = bl.height_count;     // steps [0, wide) get a launch of their own: up to the last one too populous for one workgroup
        while (bl.wide_heights > 0 && steps[2 * bl.wide_heights - 1] - steps[2 * bl.wide_heights - 2] <= KJ_REFIT_TOP_NODES) --bl.wide_heights;
        s->blas_steps.insert(s->blas_steps.end(), steps.begin(), steps.end());
        if (getenv("KJ_SCENE_DEBUG")) {
            fprintf(stderr, "[kj] mesh %u: %u tris, %u nodes, %u refit steps (%u wide):", mi, ntri, bl.node_count, bl.height_count, bl.wide_heights);
            for (size_t h = 0; h < steps.size(); h += 2) fprintf(stderr, " %u", steps[h + 1] - steps[h]);
            fprintf(stderr, "\n");
        }
        s->blas_nodes_used += bl.node_count;
        s->obj_tris_used += ntri;
        bl.built = true;
    }
    if (s->meshes_dirty) KJ_TRY_HIP(s->d_blas_steps.upload(s->blas_steps.data(), s->blas_steps.size() * 4, stream));
    s->last_commit_ms[0] = ms_since(t0);
    const auto t1 = Clock::now();
    // 2. instances: world-triangle and world-node ranges (numbered over the live instances in slot order, like a flattened scene), world boxes
    const uint32_t ni = uint32_t(s->instances.size());
    std::vector<GpuInstance> ginst(ni);
    std::vector<BvhTri> tlas_prims;
    struct TopLeaf { uint32_t inst, top; float area, wb[6]; };     // a node of an instance's top levels (KjScene::blas_top[mesh][top]) and its padded world box
    std::vector<TopLeaf> top_open, top_leaves;
    std::vector<KjTriangleLight> lights;
    std::vector<uint32_t> tri_base(ni, 0), node_base(ni, 0), id_base(ni, 0);
    // Top-tree leaves: not whole instances but the largest nodes of their top levels, opened greedily by world-space surface area until the budget
    // is used (an instance's copy of its BLAS lives in the world arrays, so ANY of its nodes can hang off the top tree: no transform, no
    // change to the walk). A terrain under 64 objects is then a few dozen blocks among them instead of one box around everything. The budget
    // depends on the number of instance slots only, so the reservation (and with it the layout of the world arrays) is stable across commits.
    // Built, measured (200 k-triangle city, instrumented trace pass: 16.3 -> 16.0 node visits per closest-hit ray, 15.0 -> 15.2 per shadow ray --
    // the terrain's top levels move into a top tree that is one level deeper for it) and therefore NOT the default: kj_scene_set_open_instances.
    // Who builds the top tree: the host (binned SAH, the better tree) while that is cheap, the device (a linear BVH over the same boxes) once the host's
    // build would be what a per-frame commit costs -- 1.1 ms at 1 k leaves, 10 ms at 8 k, 46 ms at 32 k against 0.1 ms for the refit of a moved
    // instance (profiles/r03_top_tree_build.md). kj_scene_set_top_build_mode / KJ_SCENE_TOP_BUILD: 0 = by leaf count, 1 = host, 2 = device.
    static const int top_env = kj_debug_getenv("KJ_SCENE_TOP_BUILD") ? atoi(kj_debug_getenv("KJ_SCENE_TOP_BUILD")) : -1;
    const uint32_t top_mode = top_env >= 0 && top_env <= 2 ? uint32_t(top_env) : s->top_build_mode;
    auto device_top_wanted = [&](uint32_t leaves) { return top_mode == 2u || (top_mode == 0u && leaves >= KJ_TOP_DEVICE_MIN_LEAVES); };
    static const bool open_env = kj_debug_getenv("KJ_SCENE_OPEN_INSTANCES") && atoi(kj_debug_getenv("KJ_SCENE_OPEN_INSTANCES")) != 0;
    const bool open_instances = s->open_instances || open_env;
    static const uint32_t budget_env = kj_debug_getenv("KJ_SCENE_OPEN_BUDGET") ? uint32_t(std::max(0, atoi(kj_debug_getenv("KJ_SCENE_OPEN_BUDGET")))) : 0u;      // measurement switch: leaves of the top tree when instances are opened
    const uint32_t top_budget = open_instances ? (budget_env ? std::max(ni, std::min(32768u, budget_env)) : std::min(4096u, 4u * ni + 16u)) : ni;
    const uint32_t tlas_capacity = std::max(1u, top_budget);     // a 4-wide tree over n single-node leaves has fewer than n nodes
    uint32_t total_tris = 0, total_nodes = tlas_capacity, max_blas_stack = 1;
    // A commit that only REMOVED instances (or moved some) keeps the layout of the world arrays: the removed instance's triangles and
    // nodes stay where they are, unreferenced by the new top tree -- a hole -- until the next commit that adds something lays everything
    // out anew. World triangle ids are positions in these arrays, so the ids of the surviving instances keep their order and equal-t
    // ties resolve as in a freshly flattened scene. (Round 2 re-derived every instance on a removal: 0.84 ms at 65 instances.)
    const bool keep_layout = s->committed_once && !s->instances_added && !s->meshes_dirty && s->inst_tri_base.size() == ni && s->tlas_capacity == tlas_capacity;
    for (uint32_t ii = 0; ii < ni; ++ii) {
        const KjScene::Inst& inst = s->instances[ii];
        GpuInstance& g = ginst[ii];
        memcpy(g.xform, inst.xform, 48);
        g.mesh = inst.mesh; g.emissive_multiplier = inst.emissive_multiplier; g.pad0 = g.pad1 = 0;
        if (!inst.alive) continue;
        const KjScene::Blas& bl = s->blas[inst.mesh];
        const float* x = inst.xform;
        id_base[ii] = total_tris;              // dense over the live instances, whatever the array layout
        if (keep_layout) { tri_base[ii] = s->inst_tri_base[ii]; node_base[ii] = s->inst_node_base[ii]; }
        else { tri_base[ii] = total_tris; node_base[ii] = total_nodes; }
        total_tris += bl.tri_count; total_nodes += bl.node_count;
        max_blas_stack = std::max(max_blas_stack, bl.max_stack);
        top_open.push_back(TopLeaf{ii, 0u, 0.0f, {0, 0, 0, 0, 0, 0}});
        auto xf_point = [&](const float* p, float* o) {
            o[0] = x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[3];
            o[1] = x[4] * p[0] + x[5] * p[1] + x[6] * p[2] + x[7];
            o[2] = x[8] * p[0] + x[9] * p[1] + x[10] * p[2] + x[11];
        };
        for (const KjTriangleLight& l : s->mesh_lights[inst.mesh]) {  // TriangleLight::transform, scale_radiance
            KjTriangleLight w = l;
            for (int k = 0; k < 3; ++k) xf_point(&l.verts[k * 3], &w.verts[k * 3]);
            for (int k = 0; k < 3; ++k) w.radiance[k] = l.radiance[k] * inst.emissive_multiplier;
            lights.push_back(w);
        }
    }
    {   // the (padded) world box of a top-level node encloses everything below it in the instance's refit copy: its object-space box holds the
        // node's triangles, the transformed corners' box holds the transformed triangles up to rounding, which the pad covers. Any transform
        // will do, singular ones included (a mesh flattened into a plane still has triangles to hit): nothing here needs its inverse.
        auto place = [&](TopLeaf& t) {
            const KjScene::Inst& inst = s->instances[t.inst];
            const KjScene::BlasTopNode& tn = s->blas_top[inst.mesh][t.top];
            float wb[6];
            world_box(inst.xform, tn.box, wb);
            float max_abs = 0.0f;
            for (int k = 0; k < 6; ++k) max_abs = std::max(max_abs, std::fabs(wb[k]));
            const float pad_world = 16.0f * FLT_EPSILON * std::max(max_abs, 1e-3f);
            for (int k = 0; k < 3; ++k) { t.wb[k] = wb[k] - pad_world; t.wb[3 + k] = wb[3 + k] + pad_world; }
            const float dx = t.wb[3] - t.wb[0], dy = t.wb[4] - t.wb[1], dz = t.wb[5] - t.wb[2];
            t.area = dx * dy + dy * dz + dz * dx;
        };
        auto smaller = [](const TopLeaf& a, const TopLeaf& b) { return a.area != b.area ? a.area < b.area : (a.inst != b.inst ? a.inst > b.inst : a.top > b.top); };
        for (TopLeaf& t : top_open) place(t);
        uint32_t leaves = uint32_t(top_open.size());
        // nothing to open and a device build ahead (it orders the leaves itself, by Morton code): the leaves are the instances as they come -- the
        // largest-first order below only matters to the host's builder, and popping a heap of n entries is a third of what is left of this stage
        if (!open_instances && device_top_wanted(leaves)) { top_leaves.swap(top_open); }
        std::make_heap(top_open.begin(), top_open.end(), smaller);
        while (!top_open.empty()) {
            std::pop_heap(top_open.begin(), top_open.end(), smaller);
            const TopLeaf t = top_open.back();
            top_open.pop_back();
            const KjScene::BlasTopNode& tn = s->blas_top[s->instances[t.inst].mesh][t.top];
            if (!open_instances || tn.child_count == 0 || leaves + tn.child_count - 1 > top_budget) { top_leaves.push_back(t); continue; }
            for (uint32_t c = 0; c < tn.child_count; ++c) {
                TopLeaf ch{t.inst, tn.first_child + c, 0.0f, {0, 0, 0, 0, 0, 0}};
                place(ch);
                top_open.push_back(ch);
                std::push_heap(top_open.begin(), top_open.end(), smaller);
            }
            leaves += tn.child_count - 1;
        }
        for (size_t j = 0; j < top_leaves.size(); ++j) {
            BvhTri t{};
            for (int k = 0; k < 3; ++k) { t.v0[k] = t.v2[k] = top_leaves[j].wb[k]; t.v1[k] = top_leaves[j].wb[3 + k]; }
            t.prim = uint32_t(j);
            tlas_prims.push_back(t);
        }
    }
    const bool device_top = !tlas_prims.empty() && device_top_wanted(uint32_t(tlas_prims.size()));
    s->live_tri_count = total_tris;                                                      // what kj_scene_stats reports: triangles a ray can hit
    if (keep_layout) { total_tris = s->tri_count; total_nodes = s->world_nodes; }      // array sizes as laid out, holes included
    KJ_REQUIRE(total_tris > 0, "scene has no triangles");
    KJ_REQUIRE(total_tris < (1u << 28), "too many triangles for 28-bit leaf references");
    // 3. top tree over the opened nodes of the live instances (one node per leaf): a leaf child becomes a reference to that NODE of the instance's copy
    BuiltBvh tl;
    if (tlas_prims.empty()) {     // nothing to hit: a root with four empty children
        BvhNode n;
        memset(&n, 0, sizeof(n));
        for (int i = 0; i < 4; ++i) { n.child[i] = 0xffffffffu; for (int k = 0; k < 3; ++k) { n.qlo[k][i] = 255; n.qhi[k][i] = 0; } }
        for (int k = 0; k < 3; ++k) n.exp8[k] = 127;
        tl.nodes.push_back(n);
        tl.max_stack = 1;
    } else if (!device_top) {
        build_bvh4(tlas_prims, tl, 1);
        for (BvhNode& n : tl.nodes)
            for (int i = 0; i < 4; ++i)
                if (n.child[i] != 0xffffffffu && (n.child[i] & KJ_BVH_LEAF)) {
                    const TopLeaf& t = top_leaves[tl.tris[n.child[i] & 0x0fffffffu].prim];
                    n.child[i] = node_base[t.inst] + s->blas_top[s->instances[t.inst].mesh][t.top].node;
                }
    }
    uint32_t top_node_count = uint32_t(tl.nodes.size()), top_max_stack = tl.max_stack;      // (the device build reports its own below)
    if (!device_top) {
        KJ_REQUIRE(top_node_count <= tlas_capacity, "top tree larger than its reservation");
        KJ_REQUIRE(top_max_stack + 1 + max_blas_stack <= KJ_BVH_LDS_STACK + KJ_BVH_SPILL_STACK, "BVH too deep for the traversal stack");
    }
    s->last_commit_ms[1] = ms_since(t1);
    const auto t2 = Clock::now();
    // 4. per-commit tables
    KJ_TRY_HIP(s->d_instances.upload(ginst.data(), ginst.size() * sizeof(GpuInstance), stream));
    const uint32_t light_count = uint32_t(lights.size());
    if (lights.empty()) lights.push_back(KjTriangleLight{});     // keep the buffer non-empty
    KJ_TRY_HIP(s->d_lights.upload(lights.data(), lights.size() * sizeof(KjTriangleLight), stream));
    // 5. world-space triangles and nodes: all instances when the set (hence the numbering) changed, else the moved ones -- on the device
    const bool relayout = s->d_tris.bytes != size_t(total_tris) * sizeof(BvhTri) || s->world_nodes != total_nodes || s->tlas_capacity != tlas_capacity;
    const bool all = !keep_layout || relayout;
    if (s->d_tris.bytes != size_t(total_tris) * sizeof(BvhTri)) KJ_TRY_HIP(s->d_tris.alloc(size_t(total_tris) * sizeof(BvhTri), stream));
    if (s->d_nodes.bytes != size_t(total_nodes) * sizeof(BvhNode)) {
        KJ_TRY_HIP(s->d_nodes.alloc(size_t(total_nodes) * sizeof(BvhNode), stream));
        KJ_TRY_HIP(s->d_node_boxes.alloc(size_t(total_nodes) * 24, stream));
    }
    if (device_top) {
        // the same leaves the host build would have been given -- box and the world node each stands for -- as two small uploads; nodes straight into the
        // world array's reservation. The builder's working set stays with the scene (a per-frame commit must not allocate).
        std::vector<float> boxes(tlas_prims.size() * 6);
        std::vector<uint32_t> refs(tlas_prims.size());
        for (size_t j = 0; j < tlas_prims.size(); ++j) {
            for (int k = 0; k < 3; ++k) { boxes[j * 6 + k] = tlas_prims[j].v0[k]; boxes[j * 6 + 3 + k] = tlas_prims[j].v1[k]; }
            refs[j] = node_base[top_leaves[j].inst] + s->blas_top[s->instances[top_leaves[j].inst].mesh][top_leaves[j].top].node;
        }
        if (s->d_top_boxes.bytes < boxes.size() * 4) { KJ_TRY_HIP(s->d_top_boxes.alloc(boxes.size() * 4 + boxes.size(), stream)); KJ_TRY_HIP(s->d_top_refs.alloc(refs.size() * 4 + refs.size(), stream)); }
        KJ_TRY_HIP(hipMemcpyAsync(s->d_top_boxes.p, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice, stream));
        KJ_TRY_HIP(hipMemcpyAsync(s->d_top_refs.p, refs.data(), refs.size() * 4, hipMemcpyHostToDevice, stream));
        if (!s->top_scratch) s->top_scratch = new LbvhScratch();
        LbvhResult tr;
        KJ_TRY_HIP(build_top_lbvh_device((const float*)s->d_top_boxes.p, (const uint32_t*)s->d_top_refs.p, uint32_t(refs.size()), (Bvh4Node*)s->d_nodes.p, &tr, s->top_scratch, stream));
        top_node_count = tr.node_count; top_max_stack = tr.max_stack;
        KJ_REQUIRE(top_node_count <= tlas_capacity, "top tree larger than its reservation");
        KJ_REQUIRE(top_max_stack + 1 + max_blas_stack <= KJ_BVH_LDS_STACK + KJ_BVH_SPILL_STACK, "BVH too deep for the traversal stack");
    } else
        KJ_TRY_HIP(hipMemcpyAsync(s->d_nodes.p, tl.nodes.data(), tl.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, stream));
    std::vector<InstanceTriJob> jobs, renumber;
    std::vector<InstanceRefitJob> refits;
    uint32_t max_wide = 0;
    for (uint32_t ii = 0; ii < ni; ++ii) {
        const KjScene::Inst& inst = s->instances[ii];
        if (!inst.alive) continue;
        const KjScene::Blas& bl = s->blas[inst.mesh];
        InstanceTriJob j;
        memset(&j, 0, sizeof(j));
        memcpy(j.xform, inst.xform, 48);
        j.src = bl.tri_base; j.dst = tri_base[ii]; j.count = bl.tri_count; j.instance = ii; j.id_base = id_base[ii];
        if (!(all || s->xform_dirty[ii])) {     // stays where it is; its ids move down when an instance in front of it was removed
            if (ii < s->inst_id_base.size() && s->inst_id_base[ii] != id_base[ii]) renumber.push_back(j);
            continue;
        }
        jobs.push_back(j);
        refits.push_back(InstanceRefitJob{bl.node_base, node_base[ii], bl.node_count, tri_base[ii], bl.heights_base, bl.height_count, bl.wide_heights, 0});
        max_wide = std::max(max_wide, bl.wide_heights);
    }
    if (!jobs.empty()) {
        KJ_TRY_HIP(s->d_jobs.upload(jobs.data(), jobs.size() * sizeof(InstanceTriJob), stream));
        KJ_TRY_HIP(s->d_refit_jobs.upload(refits.data(), refits.size() * sizeof(InstanceRefitJob), stream));
        KJ_TRY_HIP(launch_instance_triangles((const BvhTri*)s->d_obj_tris.p, (BvhTri*)s->d_tris.p, (const InstanceTriJob*)s->d_jobs.p, uint32_t(jobs.size()), stream));
        KJ_TRY_HIP(launch_instance_refit((const Bvh4Node*)s->d_blas_nodes.p, (const uint2*)s->d_blas_steps.p, (const BvhTri*)s->d_tris.p,
                                         (const InstanceRefitJob*)s->d_refit_jobs.p, uint32_t(refits.size()), max_wide, (Bvh4Node*)s->d_nodes.p, s->d_node_boxes.p, stream));
    }
    if (!renumber.empty()) {
        KJ_TRY_HIP(s->d_renumber_jobs.upload(renumber.data(), renumber.size() * sizeof(InstanceTriJob), stream));
        KJ_TRY_HIP(launch_instance_renumber((BvhTri*)s->d_tris.p, (const InstanceTriJob*)s->d_renumber_jobs.p, uint32_t(renumber.size()), stream));
    }
    KJ_TRY_HIP(hipStreamSynchronize(stream));  // host vectors go out of scope
    s->inst_id_base = id_base;
    s->inst_tri_base = tri_base;
    s->inst_node_base = node_base;
    s->tri_count = total_tris;
    s->node_count = total_nodes;
    s->world_nodes = total_nodes; s->tlas_capacity = tlas_capacity;
    s->tlas_node_count = top_node_count;
    s->top_built_on_device = device_top;
    s->bvh_root = 0;
    s->bvh_max_depth = top_max_stack + 1 + max_blas_stack;
    s->light_count = light_count;
    s->meshes_dirty = false; s->instance_set_dirty = false; s->instances_added = false; s->committed_once = true;
    std::fill(s->xform_dirty.begin(), s->xform_dirty.end(), uint8_t(0));
    s->committed = true;
    s->last_commit_ms[2] = ms_since(t2);
    s->last_commit_ms[3] = ms_since(t0);
    return KJ_OK;
}
KjStatus kj_scene_set_open_instances(KjScene* s, uint32_t enable) {
    KJ_REQUIRE(s, "null scene");
    if (s->open_instances != (enable != 0)) { s->open_instances = enable != 0; s->instance_set_dirty = true; s->instances_added = true; s->committed = false; }   // the reservation changes: lay out anew
    return KJ_OK;
}
KjStatus kj_scene_set_top_build_mode(KjScene* s, uint32_t mode) {
    KJ_REQUIRE(s && mode <= 2, "mode must be KJ_TOP_BUILD_AUTO (0), KJ_TOP_BUILD_HOST (1) or KJ_TOP_BUILD_DEVICE (2)");
    if (s->top_build_mode != mode) { s->top_build_mode = mode; s->instance_set_dirty = true; s->committed = false; }     // the next commit builds the top tree the new way
    return KJ_OK;
}
KjStatus kj_scene_top_tree_info(KjScene* s, uint32_t* out_nodes, uint32_t* out_capacity, uint32_t* out_built_on_device) {
    KJ_REQUIRE(s && out_nodes && out_capacity && out_built_on_device, "null argument");
    if (!s->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    *out_nodes = s->tlas_node_count; *out_capacity = s->tlas_capacity; *out_built_on_device = s->top_built_on_device ? 1u : 0u;
    return KJ_OK;
}
KjStatus kj_scene_set_blas_build_mode(KjScene* s, uint32_t mode) {
    KJ_REQUIRE(s && mode <= 2, "mode must be KJ_BLAS_BUILD_FAST_TRACE (0), KJ_BLAS_BUILD_FAST_BUILD (1) or KJ_BLAS_BUILD_DEVICE_PLOC (2)");
    s->blas_build_mode = mode;
    return KJ_OK;
}
KjStatus kj_scene_last_commit_ms(KjScene* s, double out_ms[4]) {
    KJ_REQUIRE(s && out_ms, "null argument");
    for (int k = 0; k < 4; ++k) out_ms[k] = s->last_commit_ms[k];
    return KJ_OK;
}

KjStatus kj_scene_triangle_light_count(KjScene* s, uint32_t* out) {
    KJ_REQUIRE(s && out, "null argument");
    if (!s->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    *out = s->light_count;
    return KJ_OK;
}
KjStatus kj_scene_stats(KjScene* s, uint32_t* out_tri_count, uint32_t* out_node_count, uint64_t* out_bvh_bytes) {
    KJ_REQUIRE(s, "null scene");
    if (!s->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    if (out_tri_count) *out_tri_count = s->live_tri_count;
    if (out_node_count) *out_node_count = s->node_count;
    // everything the acceleration structure keeps on the device: per-INSTANCE world nodes (64 B) and world triangles (48 B) -- unlike
    // the reference's TLAS / BLAS, instancing saves no memory here: an instance costs its mesh's whole tree again -- plus the per-mesh
    // BLAS pool (object-space nodes + triangles), the refit's box scratch (24 B per world node) and the refit step table
    if (out_bvh_bytes) *out_bvh_bytes = uint64_t(s->node_count) * sizeof(BvhNode) + uint64_t(s->tri_count) * sizeof(BvhTri) + uint64_t(s->obj_tris_used) * sizeof(BvhTri) +
                                        uint64_t(s->blas_nodes_used) * sizeof(BvhNode) + uint64_t(s->node_count) * 24u + uint64_t(s->blas_steps.size()) * 4u;
    return KJ_OK;
}

} // extern "C"
