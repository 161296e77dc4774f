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

// inc/ray_cone.hlsl
struct RayCone {
    float width, spread_angle;
    KJ_HD static RayCone from_spread_angle(float a) { return RayCone{0.0f, a}; }
    KJ_HD RayCone propagate(float surface_spread_angle, float hit_t) const { return RayCone{spread_angle * hit_t + width, spread_angle + surface_spread_angle}; }
    KJ_HD float width_at_t(float t) const { return width + spread_angle * t; }
};
// pixel_ray_cone_from_image_height (inc/frame_constants.hlsl:227-233)
KJ_HD RayCone pixel_ray_cone_from_image_height(const FrameConstants& fc, float image_height) {
    return RayCone{0.0f, atanf(2.0f * fc.view_constants.clip_to_view[0] / image_height)};
}

// ---- material-map sampling: bindless_textures[idx].SampleLevel(sampler_llr, uv, lod) (rchit:96-99,106,172) — fixed-function in
// the reference, defined here as: RGBA8 texels -> float (sRGB maps decode rgb per texel first), bilinear with repeat addressing
// inside a level, linear between floor(lod) and floor(lod)+1, lod clamped to [0, mips-1] (NaN -> 0).
KJ_D float srgb8_to_linear(float c) { return c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f); }
KJ_D V4 map_texel(const SceneView& sc, const MapDesc& m, uint32_t level_offset, int lw, int lh, int x, int y) {
    x %= lw; if (x < 0) x += lw;
    y %= lh; if (y < 0) y += lh;
    const uint32_t t = *(const uint32_t*)(sc.tex_data + m.offset + level_offset + (size_t(y) * lw + x) * 4);
    V4 v{float(t & 0xffu) * (1.0f / 255.0f), float((t >> 8) & 0xffu) * (1.0f / 255.0f), float((t >> 16) & 0xffu) * (1.0f / 255.0f), float(t >> 24) * (1.0f / 255.0f)};
    if (m.flags & 0x100u) { v.x = srgb8_to_linear(v.x); v.y = srgb8_to_linear(v.y); v.z = srgb8_to_linear(v.z); }
    return v;
}
KJ_D V4 map_bilinear(const SceneView& sc, const MapDesc& m, uint32_t level, V2 uv) {
    uint32_t off = 0;
    for (uint32_t k = 0; k < level; ++k) off += max(1u, m.width >> k) * max(1u, m.height >> k) * 4u;
    const int lw = int(max(1u, m.width >> level)), lh = int(max(1u, m.height >> level));
    const float fx = uv.x * float(lw) - 0.5f, fy = uv.y * float(lh) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = int(x0f), y0 = int(y0f);
    const V4 s00 = map_texel(sc, m, off, lw, lh, x0, y0), s10 = map_texel(sc, m, off, lw, lh, x0 + 1, y0);
    const V4 s01 = map_texel(sc, m, off, lw, lh, x0, y0 + 1), s11 = map_texel(sc, m, off, lw, lh, x0 + 1, y0 + 1);
    const V4 a = s00 * (1.0f - tx) + s10 * tx, b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}
KJ_D V4 sample_map(const SceneView& sc, uint32_t idx, V2 uv, float lod) {
    const MapDesc m = sc.maps[idx];
    const uint32_t mips = m.flags & 0xffu;
    if (mips == 0) return V4{m.color.x, m.color.y, m.color.z, m.color.w};
    if (!(fabsf(uv.x) < 1e6f && fabsf(uv.y) < 1e6f)) uv = V2{0, 0};    // non-finite / absurd uv: texel 0
    lod = fminf(fmaxf(lod, 0.0f), float(mips - 1));                    // fmaxf(NaN, 0) = 0
    const float l0f = floorf(lod);
    const uint32_t l0 = uint32_t(l0f);
    const float f = lod - l0f;
    const V4 c0 = map_bilinear(sc, m, l0, uv);
    if (f == 0.0f || l0 + 1 >= mips) return c0;
    const V4 c1 = map_bilinear(sc, m, l0 + 1, uv);
    return c0 * (1.0f - f) + c1 * f;
}
// compute_texture_lod (rchit:29-44)
KJ_D float texture_lod(const SceneView& sc, uint32_t idx, float triangle_constant, V3 ray_direction, V3 surf_normal, float cone_width) {
    const MapDesc m = sc.maps[idx];
    const float w = (m.flags & 0xffu) ? float(m.width) : 1.0f, h = (m.flags & 0xffu) ? float(m.height) : 1.0f;
    float lambda = triangle_constant;
    lambda += log2f(fabsf(cone_width));
    lambda += 0.5f * log2f(w * h);
    lambda -= log2f(fabsf(dot(normalize(ray_direction), surf_normal)));
    return lambda;
}
KJ_D V2 transform_material_uv(const KjMeshMaterial* mat, V2 uv, uint32_t map_idx) {   // inc/mesh.hlsl:63-68
    const float* t = mat->map_transforms + map_idx * 6;
    return V2{t[0] * uv.x + t[1] * uv.y + t[4], t[2] * uv.x + t[3] * uv.y + t[5]};
}

// Shades a closest hit (rt/gbuffer.rchit.hlsl:46-202). `cone_width` = payload.ray_cone.width_at_t(hit distance).
KJ_D uint4 shade_gbuffer_hit(const SceneView& sc, const FrameConstants& fc, V3 ray_d, const RayHit& h, uint32_t path_length, float cone_width) {
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
    const V3 surf_normal_os = normalize(cross(vpos[1] - vpos[0], vpos[2] - vpos[0]));
    if (fc.render_overrides.flags & KJ_OVERRIDE_FORCE_FACE_NORMALS) normal = surf_normal_os;
    V4 v_color = v4(1.0f);
    if (mesh.vertex_aux_offset != 0) {
        float4 c0 = vb_f4(sc, mesh.vertex_aux_offset + ind[0] * 16), c1 = vb_f4(sc, mesh.vertex_aux_offset + ind[1] * 16), c2 = vb_f4(sc, mesh.vertex_aux_offset + ind[2] * 16);
        v_color = V4{c0.x, c0.y, c0.z, c0.w} * bary.x + V4{c1.x, c1.y, c1.z, c1.w} * bary.y + V4{c2.x, c2.y, c2.z, c2.w} * bary.z;
    }
    const uint32_t material_id = vb_u32(sc, mesh.vertex_mat_offset + ind[0] * 4);
    const KjMeshMaterial* mat = (const KjMeshMaterial*)(sc.vertex_buffer + mesh.mat_data_offset + material_id * sizeof(KjMeshMaterial));
    // texture coordinates + ray-cone LOD (only evaluated when one of the three maps is an image: placeholders ignore both)
    const bool any_image = ((sc.maps[mat->maps[1]].flags | sc.maps[mat->maps[2]].flags | sc.maps[mat->maps[3]].flags) & 0xffu) != 0;
    V2 uv{0, 0};
    float lod_triangle_constant = 0;
    V3 surf_normal_ws = v3(0.0f);
    if (any_image) {
        const float2 t0 = *(const float2*)(sc.vertex_buffer + mesh.vertex_uv_offset + ind[0] * 8), t1 = *(const float2*)(sc.vertex_buffer + mesh.vertex_uv_offset + ind[1] * 8),
                     t2 = *(const float2*)(sc.vertex_buffer + mesh.vertex_uv_offset + ind[2] * 8);
        uv = V2{t0.x, t0.y} * bary.x + V2{t1.x, t1.y} * bary.y + V2{t2.x, t2.y} * bary.z;
        const float twice_uv_area = fabsf((t1.x - t0.x) * (t2.y - t0.y) - (t2.x - t0.x) * (t1.y - t0.y));
        const V3 w0{tp[0].x, tp[0].y, tp[0].z}, w1{tp[1].x, tp[1].y, tp[1].z}, w2{tp[2].x, tp[2].y, tp[2].z};   // == mul(ObjectToWorld3x4, v.position)
        const float twice_tri_area = length(cross(w1 - w0, w2 - w0));
        lod_triangle_constant = 0.5f * log2f(twice_uv_area / twice_tri_area);
        const float* m = inst.xform;
        surf_normal_ws = normalize(V3{m[0] * surf_normal_os.x + m[1] * surf_normal_os.y + m[2] * surf_normal_os.z, m[4] * surf_normal_os.x + m[5] * surf_normal_os.y + m[6] * surf_normal_os.z,
                                      m[8] * surf_normal_os.x + m[9] * surf_normal_os.y + m[10] * surf_normal_os.z});
    }
    const V4 albedo_texel = sample_map(sc, mat->maps[2], transform_material_uv(mat, uv, 0), texture_lod(sc, mat->maps[2], lod_triangle_constant, ray_d, surf_normal_ws, cone_width));
    const V3 albedo = xyz(albedo_texel) * V3{mat->base_color_mult[0], mat->base_color_mult[1], mat->base_color_mult[2]} * xyz(v_color);
    const V4 mr = sample_map(sc, mat->maps[1], transform_material_uv(mat, uv, 2), texture_lod(sc, mat->maps[1], lod_triangle_constant, ray_d, surf_normal_ws, cone_width));
    const float perceptual_roughness = mat->roughness_mult * mr.x;
    float roughness = clampf(perceptual_roughness * perceptual_roughness, 1e-4f, 1.0f);
    float metalness = mr.y * mat->metalness_factor;
    if (fc.render_overrides.flags & KJ_OVERRIDE_NO_METAL) metalness = 0;
    const float rs = fc.render_overrides.material_roughness_scale;
    if (rs <= 1) roughness *= rs;
    else roughness = square(lerp(sqrtf(roughness), 1.0f, 1.0f - 1.0f / rs));
    V3 emissive = v3(0.0f);
    if (0 == path_length || 0 == (mat->flags & KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT)) {
        const V4 e = sample_map(sc, mat->maps[3], transform_material_uv(mat, uv, 3), texture_lod(sc, mat->maps[3], lod_triangle_constant, ray_d, surf_normal_ws, cone_width));
        emissive = v3(1.0f) * xyz(e) * V3{mat->emissive[0], mat->emissive[1], mat->emissive[2]} * inst.emissive_multiplier * fc.pre_exposure;
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
                                        bool cull_back_faces, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr,
                                        RayCone ray_cone = RayCone::from_spread_angle(1.0f)) {
    GbufferPathVertex res;
    const RayHit h = bvh_trace<false, STATS>(sc.bvh, o, d, tmin, tmax, cull_back_faces, stack, stride, stats);
    res.is_hit = h.slot != 0xffffffffu;
    res.ray_t = h.t;
    if (res.is_hit) {
        res.gbuffer_packed = shade_gbuffer_hit(sc, fc, d, h, path_length, ray_cone.width_at_t(h.t * length(d)));
        res.position = mad_nc(o, d, h.t);
    } else {
        res.gbuffer_packed = make_uint4(0, 0, 0, 0);
        res.position = v3(0.0f);
    }
    return res;
}
// the same two queries with four lanes per ray (kj_bvh.hpp: bvh_trace_quad); every lane of the quad gets the result
KJ_D GbufferPathVertex gbuffer_raytrace_quad(const SceneView& sc, const FrameConstants& fc, bool active, V3 o, V3 d, float tmin, float tmax, uint32_t path_length,
                                             bool cull_back_faces, uint32_t* stack, uint32_t stride, RayCone ray_cone) {
    GbufferPathVertex res;
    const RayHit h = bvh_trace_quad<false>(sc.bvh, active, o, d, tmin, tmax, cull_back_faces, stack, stride);
    res.is_hit = active && h.slot != 0xffffffffu;
    res.ray_t = h.t;
    if (res.is_hit) {
        res.gbuffer_packed = shade_gbuffer_hit(sc, fc, d, h, path_length, ray_cone.width_at_t(h.t * length(d)));
        res.position = mad_nc(o, d, h.t);
    } else {
        res.gbuffer_packed = make_uint4(0, 0, 0, 0);
        res.position = v3(0.0f);
    }
    return res;
}
KJ_D bool rt_is_shadowed_quad(const SceneView& sc, bool active, V3 o, V3 d, float tmin, float tmax, uint32_t* stack, uint32_t stride) {
    return bvh_trace_quad<true>(sc.bvh, active, o, d, tmin, tmax, false, stack, stride).slot != 0xffffffffu;
}
// rt_is_shadowed (inc/rt.hlsl:58-70)
template <bool STATS = false>
KJ_D bool rt_is_shadowed(const SceneView& sc, V3 o, V3 d, float tmin, float tmax, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
    return bvh_trace<true, STATS>(sc.bvh, o, d, tmin, tmax, false, stack, stride, stats).slot != 0xffffffffu;
}
#endif

} // namespace kj
