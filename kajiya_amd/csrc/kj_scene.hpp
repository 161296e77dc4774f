// Scene tables shared between host (scene.cpp) and device, and the closest-hit
// "gbuffer ray" shading that replaces rt/gbuffer.rchit.hlsl:46-202 + inc/rt.hlsl:81-137.
#pragma once
#include "kj_bvh.hpp"
#include "kj_shading.hpp"

namespace kj {

struct GbufferPathVertex {
    bool is_hit;
    uint4 gbuffer_packed;
    V3 position;
    float ray_t;
};

#ifdef __HIPCC__
KJ_D uint32_t vb_u32(const SceneView& s, uint32_t off) { return *(const uint32_t*)(s.vertex_buffer + off); }
KJ_D float4 vb_f4(const SceneView& s, uint32_t off) { return *(const float4*)(s.vertex_buffer + off); }

// Shades a closest hit. 1x1 placeholder maps return a constant for every uv/LOD, so the
// ray-cone LOD term (rchit:29-44) cannot change the result and is not evaluated.
KJ_D uint4 shade_gbuffer_hit(const SceneView& sc, const FrameConstants& fc, V3 ray_d, const RayHit& h, uint32_t path_length) {
    const float4* __restrict__ tp = (const float4*)sc.bvh.tris + size_t(h.slot) * 3;
    const uint32_t inst_idx = __float_as_uint(tp[1].w), prim = __float_as_uint(tp[2].w);
    const GpuInstance inst = sc.instances[inst_idx];
    const GpuMesh mesh = sc.meshes[inst.mesh];
    const V3 bary{1.0f - h.u - h.v, h.u, h.v};
    uint32_t ind[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ind[k] = vb_u32(sc, mesh.index_offset + (prim * 3 + k) * 4);
    V3 vpos[3], vnrm[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float4 d = vb_f4(sc, mesh.vertex_core_offset + ind[k] * 16);
        vpos[k] = V3{d.x, d.y, d.z};
        vnrm[k] = unpack_unit_direction_11_10_11(__float_as_uint(d.w));
    }
    V3 normal = vnrm[0] * bary.x + vnrm[1] * bary.y + vnrm[2] * bary.z;
    if (fc.render_overrides.flags & KJ_OVERRIDE_FORCE_FACE_NORMALS) normal = normalize(cross(vpos[1] - vpos[0], vpos[2] - vpos[0]));
    V4 v_color = v4(1.0f);
    if (mesh.vertex_aux_offset != 0) {
        float4 c0 = vb_f4(sc, mesh.vertex_aux_offset + ind[0] * 16), c1 = vb_f4(sc, mesh.vertex_aux_offset + ind[1] * 16), c2 = vb_f4(sc, mesh.vertex_aux_offset + ind[2] * 16);
        v_color = V4{c0.x, c0.y, c0.z, c0.w} * bary.x + V4{c1.x, c1.y, c1.z, c1.w} * bary.y + V4{c2.x, c2.y, c2.z, c2.w} * bary.z;
    }
    const uint32_t material_id = vb_u32(sc, mesh.vertex_mat_offset + ind[0] * 4);
    const KjMeshMaterial* mat = (const KjMeshMaterial*)(sc.vertex_buffer + mesh.mat_data_offset + material_id * sizeof(KjMeshMaterial));
    const float4* __restrict__ map_colors = (const float4*)sc.map_colors;
    const float4 albedo_texel = map_colors[mat->maps[2]];
    const V3 albedo = V3{albedo_texel.x, albedo_texel.y, albedo_texel.z} * V3{mat->base_color_mult[0], mat->base_color_mult[1], mat->base_color_mult[2]} * xyz(v_color);
    const float4 mr = map_colors[mat->maps[1]];
    const float perceptual_roughness = mat->roughness_mult * mr.x;
    float roughness = clampf(perceptual_roughness * perceptual_roughness, 1e-4f, 1.0f);
    float metalness = mr.y * mat->metalness_factor;
    if (fc.render_overrides.flags & KJ_OVERRIDE_NO_METAL) metalness = 0;
    const float rs = fc.render_overrides.material_roughness_scale;
    if (rs <= 1) roughness *= rs;
    else roughness = square(lerp(sqrtf(roughness), 1.0f, 1.0f - 1.0f / rs));
    V3 emissive = v3(0.0f);
    if (0 == path_length || 0 == (mat->flags & KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT)) {
        const float4 e = map_colors[mat->maps[3]];
        emissive = v3(1.0f) * V3{e.x, e.y, e.z} * V3{mat->emissive[0], mat->emissive[1], mat->emissive[2]} * inst.emissive_multiplier * fc.pre_exposure;
    }
    GbufferData g;
    g.albedo = albedo;
    const float* m = inst.xform;
    g.normal = normalize(V3{m[0] * normal.x + m[1] * normal.y + m[2] * normal.z, m[4] * normal.x + m[5] * normal.y + m[6] * normal.z,
                            m[8] * normal.x + m[9] * normal.y + m[10] * normal.z});
    g.roughness = roughness;
    g.metalness = metalness;
    g.emissive = emissive;
    if (dot(ray_d, g.normal) > 0) g.normal = -g.normal;
    return gbuffer_pack(g);
}

// GbufferRaytrace::trace (inc/rt.hlsl:112-137)
template <bool STATS = false>
KJ_D GbufferPathVertex gbuffer_raytrace(const SceneView& sc, const FrameConstants& fc, V3 o, V3 d, float tmin, float tmax, uint32_t path_length,
                                        bool cull_back_faces, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
    GbufferPathVertex res;
    const RayHit h = bvh_trace<false, STATS>(sc.bvh, o, d, tmin, tmax, cull_back_faces, stack, stride, stats);
    res.is_hit = h.slot != 0xffffffffu;
    res.ray_t = h.t;
    if (res.is_hit) {
        res.gbuffer_packed = shade_gbuffer_hit(sc, fc, d, h, path_length);
        res.position = mad_nc(o, d, h.t);
    } else {
        res.gbuffer_packed = make_uint4(0, 0, 0, 0);
        res.position = v3(0.0f);
    }
    return res;
}
// rt_is_shadowed (inc/rt.hlsl:58-70)
template <bool STATS = false>
KJ_D bool rt_is_shadowed(const SceneView& sc, V3 o, V3 d, float tmin, float tmax, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
    return bvh_trace<true, STATS>(sc.bvh, o, d, tmin, tmax, false, stack, stride, stats).slot != 0xffffffffu;
}
#endif

} // namespace kj
