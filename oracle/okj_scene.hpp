// ORACLE (test infrastructure). Scene tables in the reference's packed layout,
// ray casting (brute force and a simple median-split BVH validated against it),
// and the closest-hit "gbuffer ray" shading of rt/gbuffer.rchit.hlsl:46-202.
//
// The reference delegates BVH build/traversal/intersection to the Vulkan driver
// (inc/rt.hlsl:63-67,122 -> TraceRay). The semantics restated here:
//   - closest hit over the OPEN interval (TMin, TMax) (Vulkan ray-triangle rule),
//   - all geometry opaque, instance custom index = mesh index,
//   - gbuffer rays on this path never cull (diffuse_trace_common.inc.hlsl:75),
//   - shadow rays accept any hit.
// Ray/triangle test: Moller-Trumbore in the operation order below; ties in t are
// broken towards the lowest world-triangle index so the result does not depend
// on traversal order. The HIP kernel uses the same operation order with FP
// contraction disabled, so (t,u,v,prim) are bit-identical.  PARITY UNPINNED at
// this boundary: the reference has no test vectors for the driver's traversal.
#pragma once
#include "okj_shading.hpp"
#include <vector>
#include <numeric>

namespace okj {

struct GpuMesh { // inc/mesh.hlsl:10-18; world_renderer.rs:43-54
    uint32_t vertex_core_offset, vertex_uv_offset, vertex_mat_offset, vertex_aux_offset,
        vertex_tangent_offset, mat_data_offset, index_offset;
    uint32_t index_count;
};
struct Instance {
    uint32_t mesh;
    float xform[12]; // row-major 3x4
    float emissive_multiplier;
    bool alive;
};
struct WorldTri {
    f3 v0, v1, v2;
    uint32_t inst, prim;
};
struct Hit {
    float t = FLT_MAX, u = 0, v = 0;
    uint32_t tri = 0xffffffffu;
    bool is_hit() const { return t != FLT_MAX; }
};
struct Ray { f3 o; float tmin; f3 d; float tmax; };

static inline f3 xform_point(const float* m, f3 p) {
    return f3{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
static inline f3 xform_dir(const float* m, f3 p) {
    return f3{m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
              m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// Returns true and updates `h` if the triangle is a closer hit.
static inline bool intersect_tri(const Ray& r, const WorldTri& tr, uint32_t tri_idx, bool cull_back, Hit& h) {
    const f3 e1 = tr.v1 - tr.v0;
    const f3 e2 = tr.v2 - tr.v0;
    const f3 pvec = cross(r.d, e2);
    const float det = dot(e1, pvec);
    if (cull_back ? (det <= 0.0f) : (det == 0.0f)) return false;
    const float inv_det = 1.0f / det;
    const f3 tvec = r.o - tr.v0;
    const float u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(r.d, qvec) * inv_det;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    const float t = dot(e2, qvec) * inv_det;
    if (!(t > r.tmin && t < r.tmax)) return false;
    if (t < h.t || (t == h.t && tri_idx < h.tri)) {
        h.t = t; h.u = u; h.v = v; h.tri = tri_idx;
        return true;
    }
    return false;
}

struct BvhNode {
    f3 bmin, bmax;
    uint32_t left, right; // children (internal) ; leaf: left = first tri, right = count | 0x80000000
};

struct Scene {
    std::vector<uint8_t> vertex_buffer; // byte-addressed, like the reference's 1 GiB buffer
    std::vector<GpuMesh> meshes;
    std::vector<Instance> instances;
    // bindless "textures" (inc/bindless_textures.hlsl): placeholder colour, or an RGBA8 mip chain (all levels back to back)
    struct Map { f4 color; std::vector<uint8_t> texels; uint32_t width = 0, height = 0, mips = 0; bool srgb = false; };
    std::vector<Map> maps;
    std::vector<std::vector<KjTriangleLight>> mesh_lights; // object-space, per mesh
    // committed
    std::vector<WorldTri> tris;
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> tri_order;
    std::vector<KjTriangleLight> triangle_lights; // world space (world_renderer.rs:1037-1058)
    bool use_bvh = true;

    template <typename T> uint32_t append(const T* data, size_t count) {
        // BufferBuilder::append (buffer_builder.rs): align to 64 bytes? The reference aligns
        // every sub-buffer; offsets only need to be consistent within this oracle.
        size_t off = (vertex_buffer.size() + 63) & ~size_t(63);
        vertex_buffer.resize(off + sizeof(T) * count);
        if (count) memcpy(vertex_buffer.data() + off, data, sizeof(T) * count);
        return uint32_t(off);
    }
    uint32_t load_u32(uint32_t off) const { uint32_t v; memcpy(&v, vertex_buffer.data() + off, 4); return v; }
    f4 load_f4(uint32_t off) const { f4 v; memcpy(&v, vertex_buffer.data() + off, 16); return v; }
    f2 load_f2(uint32_t off) const { f2 v; memcpy(&v, vertex_buffer.data() + off, 8); return v; }

    uint32_t add_mesh(const KjMeshDesc& d) {
        // keep offset 0 unused so that `vertex_aux_offset != 0` can mean "has colours"
        if (vertex_buffer.empty()) vertex_buffer.resize(64, 0);
        GpuMesh m{};
        std::vector<KjMeshMaterial> mats(d.materials, d.materials + d.material_count);
        uint32_t map_base = uint32_t(maps.size());
        for (uint32_t i = 0; i < d.map_count; ++i) {
            const KjMaterialMap& mm = d.maps[i];
            const uint8_t* c = mm.placeholder_rgba;
            Map m;
            m.color = f4{c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f, c[3] / 255.0f};
            if (mm.image_rgba8) {
                size_t bytes = 0;
                for (uint32_t k = 0; k < mm.mip_count; ++k) bytes += size_t(std::max(1u, mm.width >> k)) * std::max(1u, mm.height >> k) * 4;
                m.texels.assign(mm.image_rgba8, mm.image_rgba8 + bytes);
                m.width = mm.width; m.height = mm.height; m.mips = mm.mip_count; m.srgb = mm.srgb != 0;
            }
            maps.push_back(std::move(m));
        }
        for (auto& mat : mats) {
            for (int k = 0; k < 4; ++k) mat.maps[k] += map_base;
            if (d.use_lights) mat.flags |= KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT;
        }
        std::vector<f2> uvs(d.vertex_count, f2{0, 0});
        if (d.uvs) memcpy(uvs.data(), d.uvs, sizeof(f2) * d.vertex_count);
        std::vector<uint32_t> mids(d.vertex_count, 0);
        if (d.material_ids) memcpy(mids.data(), d.material_ids, 4 * d.vertex_count);
        m.index_offset = append(d.indices, d.index_count);
        m.vertex_core_offset = append(d.verts, d.vertex_count);
        m.vertex_uv_offset = append(uvs.data(), uvs.size());
        m.vertex_mat_offset = append(mids.data(), mids.size());
        m.vertex_aux_offset = d.colors ? append((const f4*)d.colors, d.vertex_count) : 0;
        m.vertex_tangent_offset = d.tangents ? append((const f4*)d.tangents, d.vertex_count) : 0;
        m.mat_data_offset = append(mats.data(), mats.size());
        m.index_count = d.index_count;
        meshes.push_back(m);
        // world_renderer.rs:741-773
        std::vector<KjTriangleLight> lights;
        if (d.use_lights) {
            for (uint32_t i = 0; i + 2 < d.index_count; i += 3) {
                uint32_t mat_idx = mids[d.indices[i]];
                const KjMeshMaterial& mat = d.materials[mat_idx];
                if (!(mat.emissive[0] > 0 || mat.emissive[1] > 0 || mat.emissive[2] > 0)) continue;
                KjTriangleLight l;
                for (int k = 0; k < 3; ++k) memcpy(&l.verts[k * 3], d.verts[d.indices[i + k]].pos, 12);
                memcpy(l.radiance, mat.emissive, 12);
                lights.push_back(l);
            }
        }
        mesh_lights.push_back(lights);
        return uint32_t(meshes.size() - 1);
    }
    uint32_t add_instance(uint32_t mesh, const float* xf) {
        Instance i{};
        i.mesh = mesh;
        memcpy(i.xform, xf, 48);
        i.emissive_multiplier = 1.0f;
        i.alive = true;
        instances.push_back(i);
        return uint32_t(instances.size() - 1);
    }

    void commit() {
        tris.clear();
        triangle_lights.clear();
        for (uint32_t ii = 0; ii < instances.size(); ++ii) {
            const Instance& inst = instances[ii];
            if (!inst.alive) continue;
            const GpuMesh& m = meshes[inst.mesh];
            for (uint32_t p = 0; p < m.index_count / 3; ++p) {
                WorldTri t;
                f3 v[3];
                for (int k = 0; k < 3; ++k) {
                    uint32_t idx = load_u32(m.index_offset + (p * 3 + k) * 4);
                    f4 vp = load_f4(m.vertex_core_offset + idx * 16);
                    v[k] = xform_point(inst.xform, xyz(vp));
                }
                t.v0 = v[0]; t.v1 = v[1]; t.v2 = v[2];
                t.inst = ii; t.prim = p;
                tris.push_back(t);
            }
            // TriangleLight::transform (world_renderer.rs:114-133): rotation + translation only
            for (const KjTriangleLight& l : mesh_lights[inst.mesh]) {
                KjTriangleLight w = l;
                for (int k = 0; k < 3; ++k) {
                    f3 p = xform_point(inst.xform, f3{l.verts[k * 3], l.verts[k * 3 + 1], l.verts[k * 3 + 2]});
                    w.verts[k * 3] = p.x; w.verts[k * 3 + 1] = p.y; w.verts[k * 3 + 2] = p.z;
                }
                for (int k = 0; k < 3; ++k) w.radiance[k] = l.radiance[k] * inst.emissive_multiplier;
                triangle_lights.push_back(w);
            }
        }
        build_bvh();
    }

    // ---- median-split BVH (oracle-only; validated against brute force in tests)
    void build_bvh() {
        nodes.clear();
        tri_order.resize(tris.size());
        std::iota(tri_order.begin(), tri_order.end(), 0u);
        if (tris.empty()) return;
        nodes.reserve(tris.size() * 2);
        nodes.push_back(BvhNode{});
        build_rec(0, 0, uint32_t(tris.size()));
    }
    void bounds(uint32_t first, uint32_t count, f3& bmin, f3& bmax) const {
        bmin = mk3(FLT_MAX); bmax = mk3(-FLT_MAX);
        for (uint32_t i = first; i < first + count; ++i) {
            const WorldTri& t = tris[tri_order[i]];
            bmin = vmin(bmin, vmin(t.v0, vmin(t.v1, t.v2)));
            bmax = vmax(bmax, vmax(t.v0, vmax(t.v1, t.v2)));
        }
    }
    void build_rec(uint32_t node, uint32_t first, uint32_t count) {
        f3 bmin, bmax;
        bounds(first, count, bmin, bmax);
        nodes[node].bmin = bmin; nodes[node].bmax = bmax;
        if (count <= 4) {
            nodes[node].left = first; nodes[node].right = count | 0x80000000u;
            return;
        }
        f3 cmin = mk3(FLT_MAX), cmax = mk3(-FLT_MAX);
        for (uint32_t i = first; i < first + count; ++i) {
            const WorldTri& t = tris[tri_order[i]];
            f3 c = (t.v0 + t.v1 + t.v2);
            cmin = vmin(cmin, c); cmax = vmax(cmax, c);
        }
        f3 e = cmax - cmin;
        int axis = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
        auto key = [&](uint32_t ti) {
            const WorldTri& t = tris[ti];
            f3 c = (t.v0 + t.v1 + t.v2);
            return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
        };
        uint32_t mid = first + count / 2;
        std::nth_element(tri_order.begin() + first, tri_order.begin() + mid, tri_order.begin() + first + count,
                         [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        uint32_t l = uint32_t(nodes.size()); nodes.push_back(BvhNode{});
        uint32_t r = uint32_t(nodes.size()); nodes.push_back(BvhNode{});
        nodes[node].left = l; nodes[node].right = r;
        build_rec(l, first, mid - first);
        build_rec(r, mid, first + count - mid);
    }
    static bool hit_box(const Ray& r, f3 inv_d, f3 bmin, f3 bmax, float tbest) {
        // conservative slab test (inflated by a few ulps through the >=/<= comparisons on padded t)
        f3 t0 = (bmin - r.o) * inv_d, t1 = (bmax - r.o) * inv_d;
        f3 tn = vmin(t0, t1), tf = vmax(t0, t1);
        float tnear = fmaxf(fmaxf(tn.x, tn.y), fmaxf(tn.z, r.tmin));
        float tfar = fminf(fminf(tf.x, tf.y), fminf(tf.z, tbest));
        return tnear <= tfar * 1.0000004f + 1e-30f || !(tnear == tnear) || !(tfar == tfar);
    }

    // Non-finite rays are misses (see header note; the reference issues them from zero history).
    static bool ray_finite(const Ray& r) {
        return fabsf(r.o.x) <= FLT_MAX && fabsf(r.o.y) <= FLT_MAX && fabsf(r.o.z) <= FLT_MAX && fabsf(r.d.x) <= FLT_MAX && fabsf(r.d.y) <= FLT_MAX && fabsf(r.d.z) <= FLT_MAX;
    }
    Hit trace_closest_brute(const Ray& r, bool cull_back = false) const {
        Hit h;
        if (!ray_finite(r)) return h;
        for (uint32_t i = 0; i < tris.size(); ++i) intersect_tri(r, tris[i], i, cull_back, h);
        return h;
    }
    Hit trace_closest(const Ray& r, bool cull_back = false) const {
        if (!use_bvh || nodes.empty()) return trace_closest_brute(r, cull_back);
        Hit h;
        if (!ray_finite(r)) return h;
        f3 inv_d{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        uint32_t stack[128]; int sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const BvhNode& n = nodes[stack[--sp]];
            if (!hit_box(r, inv_d, n.bmin, n.bmax, fminf(h.t, r.tmax))) continue;
            if (n.right & 0x80000000u) {
                uint32_t cnt = n.right & 0x7fffffffu;
                for (uint32_t i = n.left; i < n.left + cnt; ++i) intersect_tri(r, tris[tri_order[i]], tri_order[i], cull_back, h);
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
        return h;
    }
    bool trace_any(const Ray& r) const {
        if (!ray_finite(r)) return false;
        if (!use_bvh || nodes.empty()) {
            Hit h;
            for (uint32_t i = 0; i < tris.size(); ++i) if (intersect_tri(r, tris[i], i, false, h)) return true;
            return false;
        }
        Hit h;
        f3 inv_d{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        uint32_t stack[128]; int sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const BvhNode& n = nodes[stack[--sp]];
            if (!hit_box(r, inv_d, n.bmin, n.bmax, r.tmax)) continue;
            if (n.right & 0x80000000u) {
                uint32_t cnt = n.right & 0x7fffffffu;
                for (uint32_t i = n.left; i < n.left + cnt; ++i)
                    if (intersect_tri(r, tris[tri_order[i]], tri_order[i], false, h)) return true;
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
        return false;
    }
};

// ---- material-map sampling: bindless_textures[idx].SampleLevel(sampler_llr, uv, lod) (rchit:96-99,106,172) — fixed-function in the
// reference, defined as: RGBA8 texels -> float (sRGB maps decode rgb per texel first), bilinear with repeat addressing inside a
// level, linear between floor(lod) and floor(lod)+1, lod clamped to [0, mips-1] (NaN -> 0).
static inline float srgb8_to_linear(float c) { return c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f); }
static inline f4 map_texel(const Scene::Map& m, size_t level_offset, int lw, int lh, int x, int y) {
    x %= lw; if (x < 0) x += lw;
    y %= lh; if (y < 0) y += lh;
    const uint8_t* t = m.texels.data() + level_offset + (size_t(y) * lw + x) * 4;
    f4 v{float(t[0]) * (1.0f / 255.0f), float(t[1]) * (1.0f / 255.0f), float(t[2]) * (1.0f / 255.0f), float(t[3]) * (1.0f / 255.0f)};
    if (m.srgb) { v.x = srgb8_to_linear(v.x); v.y = srgb8_to_linear(v.y); v.z = srgb8_to_linear(v.z); }
    return v;
}
static inline f4 map_bilinear(const Scene::Map& m, uint32_t level, f2 uv) {
    size_t off = 0;
    for (uint32_t k = 0; k < level; ++k) off += size_t(std::max(1u, m.width >> k)) * std::max(1u, m.height >> k) * 4;
    const int lw = int(std::max(1u, m.width >> level)), lh = int(std::max(1u, m.height >> level));
    const float fx = uv.x * float(lw) - 0.5f, fy = uv.y * float(lh) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = int(x0f), y0 = int(y0f);
    const f4 s00 = map_texel(m, off, lw, lh, x0, y0), s10 = map_texel(m, off, lw, lh, x0 + 1, y0);
    const f4 s01 = map_texel(m, off, lw, lh, x0, y0 + 1), s11 = map_texel(m, off, lw, lh, x0 + 1, y0 + 1);
    const f4 a = s00 * (1.0f - tx) + s10 * tx, b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}
static inline f4 sample_map(const Scene& sc, uint32_t idx, f2 uv, float lod) {
    const Scene::Map& m = sc.maps[idx];
    if (m.mips == 0) return m.color;
    if (!(fabsf(uv.x) < 1e6f && fabsf(uv.y) < 1e6f)) uv = f2{0, 0};
    lod = fminf(fmaxf(lod, 0.0f), float(m.mips - 1));
    const float l0f = floorf(lod);
    const uint32_t l0 = uint32_t(l0f);
    const float f = lod - l0f;
    const f4 c0 = map_bilinear(m, l0, uv);
    if (f == 0.0f || l0 + 1 >= m.mips) return c0;
    const f4 c1 = map_bilinear(m, l0 + 1, uv);
    return c0 * (1.0f - f) + c1 * f;
}
// compute_texture_lod (rchit:29-44)
static inline float texture_lod(const Scene& sc, uint32_t idx, float triangle_constant, f3 ray_direction, f3 surf_normal, float cone_width) {
    const Scene::Map& m = sc.maps[idx];
    const float w = m.mips ? float(m.width) : 1.0f, h = m.mips ? float(m.height) : 1.0f;
    float lambda = triangle_constant;
    lambda += log2f(fabsf(cone_width));
    lambda += 0.5f * log2f(w * h);
    lambda -= log2f(fabsf(dot(normalize(ray_direction), surf_normal)));
    return lambda;
}
static inline f2 transform_material_uv(const KjMeshMaterial& mat, f2 uv, uint32_t map_idx) {   // inc/mesh.hlsl:63-68
    const float* t = mat.map_transforms + map_idx * 6;
    return f2{t[0] * uv.x + t[1] * uv.y + t[4], t[2] * uv.x + t[3] * uv.y + t[5]};
}

// inc/rt.hlsl:81-137 + rt/gbuffer.rchit.hlsl:46-202. Texture sampling of 1x1
// placeholder maps returns their constant colour for every uv/LOD, so the
// ray-cone LOD (rchit:29-44) does not influence the result for such maps.
struct GbufferPathVertex {
    bool is_hit = false;
    u4 gbuffer_packed{0, 0, 0, 0};
    f3 position{0, 0, 0};
    float ray_t = FLT_MAX;
};

static inline GbufferPathVertex gbuffer_raytrace(const Scene& sc, const FrameConstants& fc, const Ray& ray,
                                                 uint32_t path_length, bool cull_back_faces, RayCone ray_cone = RayCone{0.0f, 1.0f}) {
    GbufferPathVertex res;
    Hit h = sc.trace_closest(ray, cull_back_faces);
    if (!h.is_hit()) return res;
    const WorldTri& wt = sc.tris[h.tri];
    const Instance& inst = sc.instances[wt.inst];
    const GpuMesh& mesh = sc.meshes[inst.mesh];
    const f3 bary{1.0f - h.u - h.v, h.u, h.v};
    uint32_t ind[3];
    for (int k = 0; k < 3; ++k) ind[k] = sc.load_u32(mesh.index_offset + (wt.prim * 3 + k) * 4);
    f3 vpos[3], vnrm[3];
    for (int k = 0; k < 3; ++k) {
        f4 d = sc.load_f4(mesh.vertex_core_offset + ind[k] * 16);
        vpos[k] = xyz(d);
        vnrm[k] = unpack_unit_direction_11_10_11(asuint(d.w));
    }
    f3 normal = vnrm[0] * bary.x + vnrm[1] * bary.y + vnrm[2] * bary.z;
    const f3 surf_normal_os = normalize(cross(vpos[1] - vpos[0], vpos[2] - vpos[0]));
    if (fc.render_overrides.flags & KJ_OVERRIDE_FORCE_FACE_NORMALS) normal = surf_normal_os;

    f4 v_color = mk4(1.0f);
    if (mesh.vertex_aux_offset != 0) {
        f4 c0 = sc.load_f4(mesh.vertex_aux_offset + ind[0] * 16);
        f4 c1 = sc.load_f4(mesh.vertex_aux_offset + ind[1] * 16);
        f4 c2 = sc.load_f4(mesh.vertex_aux_offset + ind[2] * 16);
        v_color = c0 * bary.x + c1 * bary.y + c2 * bary.z;
    }
    uint32_t material_id = sc.load_u32(mesh.vertex_mat_offset + ind[0] * 4);
    KjMeshMaterial material;
    memcpy(&material, sc.vertex_buffer.data() + mesh.mat_data_offset + material_id * sizeof(KjMeshMaterial), sizeof(KjMeshMaterial));

    // texture coordinates + ray-cone LOD (only evaluated when one of the three maps is an image: placeholders ignore both)
    const bool any_image = (sc.maps[material.maps[1]].mips | sc.maps[material.maps[2]].mips | sc.maps[material.maps[3]].mips) != 0;
    f2 uv{0, 0};
    float lod_triangle_constant = 0;
    f3 surf_normal_ws = mk3(0.0f);
    const float cone_width = ray_cone.width_at_t(h.t * length(ray.d));
    if (any_image) {
        const f2 t0 = sc.load_f2(mesh.vertex_uv_offset + ind[0] * 8), t1 = sc.load_f2(mesh.vertex_uv_offset + ind[1] * 8), t2 = sc.load_f2(mesh.vertex_uv_offset + ind[2] * 8);
        uv = t0 * bary.x + t1 * bary.y + t2 * bary.z;
        const float twice_uv_area = fabsf((t1.x - t0.x) * (t2.y - t0.y) - (t2.x - t0.x) * (t1.y - t0.y));
        const float twice_tri_area = length(cross(wt.v1 - wt.v0, wt.v2 - wt.v0));
        lod_triangle_constant = 0.5f * log2f(twice_uv_area / twice_tri_area);
        surf_normal_ws = normalize(xform_dir(inst.xform, surf_normal_os));
    }
    const f4 albedo_texel = sample_map(sc, material.maps[2], transform_material_uv(material, uv, 0), texture_lod(sc, material.maps[2], lod_triangle_constant, ray.d, surf_normal_ws, cone_width));
    f3 albedo = xyz(albedo_texel) * f3{material.base_color_mult[0], material.base_color_mult[1], material.base_color_mult[2]} * xyz(v_color);
    const f4 metalness_roughness = sample_map(sc, material.maps[1], transform_material_uv(material, uv, 2), texture_lod(sc, material.maps[1], lod_triangle_constant, ray.d, surf_normal_ws, cone_width));
    float perceptual_roughness = material.roughness_mult * metalness_roughness.x;
    float roughness = clampf(perceptual_roughness * perceptual_roughness, 1e-4f, 1.0f);
    float metalness = metalness_roughness.y * material.metalness_factor;
    if (fc.render_overrides.flags & KJ_OVERRIDE_NO_METAL) metalness = 0;
    const float rs = fc.render_overrides.material_roughness_scale;
    if (rs <= 1) roughness *= rs;
    else roughness = square(lerp(sqrtf(roughness), 1.0f, 1.0f - 1.0f / rs));

    f3 emissive = mk3(0.0f);
    if (0 == path_length || 0 == (material.flags & KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT)) {
        const f4 e = sample_map(sc, material.maps[3], transform_material_uv(material, uv, 3), texture_lod(sc, material.maps[3], lod_triangle_constant, ray.d, surf_normal_ws, cone_width));
        emissive = mk3(1.0f) * xyz(e) * f3{material.emissive[0], material.emissive[1], material.emissive[2]}
            * inst.emissive_multiplier * fc.pre_exposure;
    }
    GbufferData g;
    g.albedo = albedo;
    g.normal = normalize(xform_dir(inst.xform, normal));
    g.roughness = roughness;
    g.metalness = metalness;
    g.emissive = emissive;
    if (dot(ray.d, g.normal) > 0) g.normal = -g.normal;
    res.is_hit = true;
    res.gbuffer_packed = gbuffer_pack(g);
    res.ray_t = h.t;
    res.position = ray.o + ray.d * h.t;
    return res;
}

} // namespace okj
