// KjScene: WorldRenderer's scene state for the GI path (world_renderer.rs:604-911):
// mesh upload into one byte-addressed vertex buffer + GpuMesh table, instances,
// triangle lights, and — replacing the driver's BLAS/TLAS — a host-built 4-wide
// SAH BVH with quantised child boxes (bvh_build.cpp) uploaded as 64-byte nodes
// and 48-byte leaf-ordered world-space triangles.
// Plain C++ (no device code); compiled with -ffp-contract=off so the instance
// transform of vertices rounds exactly like the oracle's.
#include "kj_host.hpp"
#include "kj_bvh_build.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <numeric>

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
    v.bvh.stack_entries = KJ_BVH_LDS_STACK;      // LDS part of the traversal stack; deeper entries spill (kj_bvh.hpp)
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

KjStatus kj_scene_create(KjDevice* dev, KjScene** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjScene* s = new KjScene();
    s->dev = dev;
    s->vertex_buffer.resize(64, 0);  // offset 0 stays unused: `vertex_aux_offset != 0` means "has colours"
    *out = s;
    return KJ_OK;
}
void kj_scene_destroy(KjScene* scene) { delete scene; }

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
            KJ_REQUIRE(s->tex_data.size() + bytes < (size_t(1) << 32), "more than 4 GiB of material maps");
            md.offset = uint32_t(s->tex_data.size());
            md.width = mm.width; md.height = mm.height;
            md.flags = mm.mip_count | (mm.srgb ? 0x100u : 0u);
            s->tex_data.insert(s->tex_data.end(), mm.image_rgba8, mm.image_rgba8 + bytes);
        }
        s->maps.push_back(md);
    }
    for (auto& m : mats) {
        for (int k = 0; k < 4; ++k) { KJ_REQUIRE(m.maps[k] < d->map_count, "material map index out of range"); m.maps[k] += map_base; }
        if (d->use_lights) m.flags |= KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT;
    }
    std::vector<float> uvs(size_t(d->vertex_count) * 2, 0.0f);
    if (d->uvs) memcpy(uvs.data(), d->uvs, uvs.size() * 4);
    std::vector<uint32_t> mids(d->vertex_count, 0);
    if (d->material_ids) memcpy(mids.data(), d->material_ids, size_t(d->vertex_count) * 4);
    for (uint32_t v : mids) KJ_REQUIRE(v < d->material_count, "material id out of range");
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
    s->committed = false;
    *out_instance = uint32_t(s->instances.size() - 1);
    return KJ_OK;
}
KjStatus kj_scene_set_instance_transform(KjScene* s, uint32_t instance, const float* xf) {
    KJ_REQUIRE(s && xf && instance < s->instances.size() && s->instances[instance].alive, "bad instance handle");
    memcpy(s->instances[instance].xform, xf, 48);
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
    s->committed = false;
    return KJ_OK;
}

KjStatus kj_scene_commit(KjScene* s, void* stream_) {
    KJ_REQUIRE(s, "null scene");
    hipStream_t stream = (hipStream_t)stream_;
    KJ_TRY_HIP(hipSetDevice(s->dev->ordinal));
    // 1. flatten instances into world space (same arithmetic as the oracle: row-major 3x4 times point)
    std::vector<BvhTri> wt;
    std::vector<KjTriangleLight> lights;
    std::vector<GpuInstance> ginst(s->instances.size());
    for (uint32_t ii = 0; ii < s->instances.size(); ++ii) {
        const KjScene::Inst& inst = s->instances[ii];
        GpuInstance& g = ginst[ii];
        memcpy(g.xform, inst.xform, 48);
        g.mesh = inst.mesh; g.emissive_multiplier = inst.emissive_multiplier; g.pad0 = g.pad1 = 0;
        if (!inst.alive) continue;
        const GpuMesh& m = s->meshes[inst.mesh];
        const float* x = inst.xform;
        auto xf_point = [&](const float* p, float* o) {
            o[0] = x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[3];
            o[1] = x[4] * p[0] + x[5] * p[1] + x[6] * p[2] + x[7];
            o[2] = x[8] * p[0] + x[9] * p[1] + x[10] * p[2] + x[11];
        };
        for (uint32_t p = 0; p < m.index_count / 3; ++p) {
            BvhTri t{};
            float* dst[3] = {t.v0, t.v1, t.v2};
            for (int k = 0; k < 3; ++k) {
                uint32_t idx;
                memcpy(&idx, s->vertex_buffer.data() + m.index_offset + (p * 3 + k) * 4, 4);
                xf_point((const float*)(s->vertex_buffer.data() + m.vertex_core_offset + size_t(idx) * 16), dst[k]);
            }
            t.world_id = uint32_t(wt.size());
            t.inst = ii; t.prim = p;
            wt.push_back(t);
        }
        for (const KjTriangleLight& l : s->mesh_lights[inst.mesh]) {  // TriangleLight::transform, scale_radiance
            KjTriangleLight w = l;
            for (int k = 0; k < 3; ++k) xf_point(&l.verts[k * 3], &w.verts[k * 3]);
            for (int k = 0; k < 3; ++k) w.radiance[k] = l.radiance[k] * inst.emissive_multiplier;
            lights.push_back(w);
        }
    }
    KJ_REQUIRE(!wt.empty(), "scene has no triangles");
    KJ_REQUIRE(wt.size() < (1u << 28), "too many triangles for 28-bit leaf references");
    // 2. hierarchy (bvh_build.cpp)
    BuiltBvh b;
    build_bvh4(wt, b);
    KJ_REQUIRE(b.max_stack + 1 <= KJ_BVH_LDS_STACK + KJ_BVH_SPILL_STACK, "BVH too deep for the traversal stack");
    s->tri_count = uint32_t(b.tris.size());
    s->node_count = uint32_t(b.nodes.size());
    s->bvh_root = 0;
    s->bvh_max_depth = b.max_stack;
    s->light_count = uint32_t(lights.size());
    // 4. upload
    KJ_TRY_HIP(s->d_vertex_buffer.upload(s->vertex_buffer.data(), s->vertex_buffer.size(), stream));
    KJ_TRY_HIP(s->d_meshes.upload(s->meshes.data(), s->meshes.size() * sizeof(GpuMesh), stream));
    KJ_TRY_HIP(s->d_instances.upload(ginst.data(), ginst.size() * sizeof(GpuInstance), stream));
    KJ_TRY_HIP(s->d_maps.upload(s->maps.data(), s->maps.size() * sizeof(MapDesc), stream));
    if (s->tex_data.empty()) s->tex_data.resize(16, 0);
    KJ_TRY_HIP(s->d_tex_data.upload(s->tex_data.data(), s->tex_data.size(), stream));
    if (lights.empty()) lights.push_back(KjTriangleLight{});
    KJ_TRY_HIP(s->d_lights.upload(lights.data(), lights.size() * sizeof(KjTriangleLight), stream));
    KJ_TRY_HIP(s->d_nodes.upload(b.nodes.data(), b.nodes.size() * sizeof(BvhNode), stream));
    KJ_TRY_HIP(s->d_tris.upload(b.tris.data(), b.tris.size() * sizeof(BvhTri), stream));
    KJ_TRY_HIP(hipStreamSynchronize(stream));  // host vectors go out of scope
    s->committed = true;
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
    if (out_tri_count) *out_tri_count = s->tri_count;
    if (out_node_count) *out_node_count = s->node_count;
    if (out_bvh_bytes) *out_bvh_bytes = uint64_t(s->node_count) * sizeof(BvhNode) + uint64_t(s->tri_count) * sizeof(BvhTri);
    return KJ_OK;
}

} // extern "C"
