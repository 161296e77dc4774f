// Software ray traversal for gfx950 — replaces VK_KHR_ray_tracing's TraceRay
// (inc/rt.hlsl:58-70,112-137; BLAS/TLAS in kajiya-backend/src/vulkan/ray_tracing.rs).
//
// Layout (built by bvh_build.cpp, resident in HBM / Infinity Cache):
//   BvhNode  64 B : both children's AABBs + child references. One node = 4 x 16-B loads
//                   issued by each lane; a visit tests two boxes.
//   BvhTri   48 B : world-space vertices + ids, stored in leaf order (3 x 16-B loads).
// Child reference: bit31 = leaf; leaf => bits[30:28] = count-1, bits[27:0] = first tri slot.
// Traversal: while-while, near child first, per-lane stack in LDS laid out
// [level][lane] (bank-conflict free: consecutive lanes hit consecutive banks).
// Ray/triangle: Moller-Trumbore, FP contraction OFF so (t,u,v) are bit-identical to
// the oracle; equal-t ties go to the lowest world triangle id.
#pragma once
#include "kj_vec.hpp"
#include "kj_scene_types.hpp"

namespace kj {

struct RayHit {
    float t, u, v;
    uint32_t slot;      // index into the leaf-ordered triangle array (0xffffffff = miss)
    uint32_t world_id;
};

#ifdef __HIPCC__
// No-contraction helpers: every mul/add rounds separately (matches the oracle's -ffp-contract=off).
KJ_D V3 sub_nc(V3 a, V3 b) {
#pragma clang fp contract(off)
    return V3{a.x - b.x, a.y - b.y, a.z - b.z};
}
KJ_D float dot_nc(V3 a, V3 b) {
#pragma clang fp contract(off)
    float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z;
    float s = x + y;
    return s + z;
}
KJ_D V3 cross_nc(V3 a, V3 b) {
#pragma clang fp contract(off)
    float x0 = a.y * b.z, x1 = a.z * b.y, y0 = a.z * b.x, y1 = a.x * b.z, z0 = a.x * b.y, z1 = a.y * b.x;
    return V3{x0 - x1, y0 - y1, z0 - z1};
}
KJ_D V3 mad_nc(V3 o, V3 d, float t) {
#pragma clang fp contract(off)
    float x = d.x * t, y = d.y * t, z = d.z * t;
    return V3{o.x + x, o.y + y, o.z + z};
}
// Returns true when the triangle is a closer hit (and updates h).
KJ_D bool intersect_tri(V3 o, V3 d, float tmin, float tmax, const float4 a, const float4 b, const float4 c, uint32_t slot, bool cull_back, RayHit& h) {
#pragma clang fp contract(off)
    const V3 v0{a.x, a.y, a.z}, v1{b.x, b.y, b.z}, v2{c.x, c.y, c.z};
    const V3 e1 = sub_nc(v1, v0);
    const V3 e2 = sub_nc(v2, v0);
    const V3 pvec = cross_nc(d, e2);
    const float det = dot_nc(e1, pvec);
    if (cull_back ? (det <= 0.0f) : (det == 0.0f)) return false;
    const float inv_det = 1.0f / det;
    const V3 tvec = sub_nc(o, v0);
    const float u = dot_nc(tvec, pvec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const V3 qvec = cross_nc(tvec, e1);
    const float v = dot_nc(d, qvec) * inv_det;
    const float upv = u + v;
    if (!(v >= 0.0f && upv <= 1.0f)) return false;
    const float t = dot_nc(e2, qvec) * inv_det;
    if (!(t > tmin && t < tmax)) return false;
    const uint32_t wid = __float_as_uint(a.w);
    if (t < h.t || (t == h.t && wid < h.world_id)) {
        h.t = t; h.u = u; h.v = v; h.slot = slot; h.world_id = wid;
        return true;
    }
    return false;
}

// stack: LDS base for this lane; entries at stack[level * stride]
struct TraverseStats { uint32_t nodes, tris; };

template <bool ANY_HIT, bool STATS = false>
KJ_D RayHit bvh_trace(const BvhView& bvh, V3 o, V3 d, float tmin, float tmax, bool cull_back, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
    RayHit h;
    h.t = FLT_MAX; h.u = 0; h.v = 0; h.slot = 0xffffffffu; h.world_id = 0xffffffffu;
    // Rays with a non-finite origin or direction are misses (the reference's validation pass issues such
    // rays for pixels without history; a hardware traversal unit rejects every box for them). Without
    // this, NaN slabs pass the fmin/fmax test and the whole tree is walked.
    if (!(fabsf(o.x) <= FLT_MAX && fabsf(o.y) <= FLT_MAX && fabsf(o.z) <= FLT_MAX && fabsf(d.x) <= FLT_MAX && fabsf(d.y) <= FLT_MAX && fabsf(d.z) <= FLT_MAX)) return h;
    const float eps = 1e-20f;
    const V3 inv_d{1.0f / (fabsf(d.x) < eps ? copysignf(eps, d.x) : d.x), 1.0f / (fabsf(d.y) < eps ? copysignf(eps, d.y) : d.y),
                   1.0f / (fabsf(d.z) < eps ? copysignf(eps, d.z) : d.z)};
    uint32_t sp = 0;
    uint32_t cur = bvh.root;
    const uint32_t NONE = 0xffffffffu;
    while (cur != NONE) {
        if (!(cur & KJ_BVH_LEAF)) {
            const float4* __restrict__ n = (const float4*)bvh.nodes + size_t(cur) * 4;
            const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            if (STATS) stats->nodes++;
            const float tlimit = ANY_HIT ? tmax : fminf(h.t, tmax);
            // left box
            float t0x = (n0.x - o.x) * inv_d.x, t1x = (n1.x - o.x) * inv_d.x;
            float t0y = (n0.y - o.y) * inv_d.y, t1y = (n1.y - o.y) * inv_d.y;
            float t0z = (n0.z - o.z) * inv_d.z, t1z = (n1.z - o.z) * inv_d.z;
            float ln = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
            float lf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tlimit));
            t0x = (n2.x - o.x) * inv_d.x; t1x = (n3.x - o.x) * inv_d.x;
            t0y = (n2.y - o.y) * inv_d.y; t1y = (n3.y - o.y) * inv_d.y;
            t0z = (n2.z - o.z) * inv_d.z; t1z = (n3.z - o.z) * inv_d.z;
            float rn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
            float rf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tlimit));
            // conservative acceptance (a few ulps of slack on the far side)
            const bool hl = ln <= lf * 1.0000004f + 1e-30f;
            const bool hr = rn <= rf * 1.0000004f + 1e-30f;
            const uint32_t lc = __float_as_uint(n0.w), rc = __float_as_uint(n1.w);
            if (hl && hr) {
                const bool left_first = ln <= rn;
                stack[sp * stride] = left_first ? rc : lc;
                sp++;
                cur = left_first ? lc : rc;
            } else if (hl) {
                cur = lc;
            } else if (hr) {
                cur = rc;
            } else {
                cur = sp ? stack[(--sp) * stride] : NONE;
            }
        } else {
            const uint32_t first = cur & 0x0fffffffu;
            const uint32_t count = ((cur >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const float4* __restrict__ tp = (const float4*)bvh.tris + size_t(first + i) * 3;
                const float4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) stats->tris++;
                if (intersect_tri(o, d, tmin, tmax, a, b, c, first + i, cull_back, h)) {
                    if (ANY_HIT) return h;
                }
            }
            cur = sp ? stack[(--sp) * stride] : NONE;
        }
    }
    return h;
}
#endif // __HIPCC__

} // namespace kj
