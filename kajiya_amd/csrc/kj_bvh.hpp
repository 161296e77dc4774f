// Software ray traversal for gfx950 — replaces VK_KHR_ray_tracing's TraceRay
// (inc/rt.hlsl:58-70,112-137; BLAS/TLAS in kajiya-backend/src/vulkan/ray_tracing.rs).
//
// One tree in world space (kj_scene_types.hpp: BvhView): a top tree over the instances whose leaf children are the root nodes of the
// instances' own trees -- each a copy of its mesh's BLAS refit around the instance's world-space triangles when the instance moves
// (scene_device.hip). The walk therefore never transforms a ray or keeps per-instance state; (t, u, v) come from world-space
// triangles and do not depend on how the scene is partitioned into instances. (A walk that descended into object-space BLASes
// through per-instance ray transforms was built first and measured 22-28 % fewer rays/s: DESIGN 3.1.)
// Node / triangle layout (bvh_build.cpp, resident in HBM / Infinity Cache):
//   Bvh4Node 64 B : up to four children; each child's AABB is 6 bytes (8 bits per plane) inside the node's own
//                   frame (origin + power-of-two step per axis). One visit = 3.5 x 16-B loads per lane and tests
//                   four boxes, so a ray takes about half the dependent steps and a quarter of the node bytes
//                   of a two-box fp32 node. Decoded planes are fma(q, step, origin): never inside the true box.
//   BvhTri   48 B : world-space fp32 vertices + ids, stored in leaf order (3 x 16-B loads).
// Child reference: bit31 = leaf; leaf => bits[30:28] = count-1, bits[27:0] = first tri slot; 0xffffffff = empty.
// Traversal: nearest child first (4-key sorting network on the entry distances), per-lane stack in LDS laid out
// [level][lane] (bank-conflict free: consecutive lanes hit consecutive banks). Two drivers share the step functions:
// bvh_trace() (one ray per lane, inside a caller's kernel) and bvh_trace_stream() (a persistent wave over a ray array:
// lanes refill as their rays finish, and each wave step issues the block -- node or triangle -- most lanes wait for).
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
#ifndef KJ_TRI_BRANCHFREE
#define KJ_TRI_BRANCHFREE 0      // 1 (round 6 experiment): every lane evaluates the whole test and the hit record is updated by selects -- no nested exec masks, no PHI copies
#endif
KJ_D bool intersect_tri(V3 o, V3 d, float tmin, float tmax, const float4 a, const float4 b, const float4 c, uint32_t slot, bool cull_back, RayHit& h) {
#pragma clang fp contract(off)
    const V3 v0{a.x, a.y, a.z}, v1{b.x, b.y, b.z}, v2{c.x, c.y, c.z};
#if KJ_TRI_BRANCHFREE && defined(__HIP_DEVICE_COMPILE__)
    {
        const V3 e1 = sub_nc(v1, v0);
        const V3 e2 = sub_nc(v2, v0);
        const V3 pvec = cross_nc(d, e2);
        const float det = dot_nc(e1, pvec);
        const bool ok_det = cull_back ? (det > 0.0f) : (det != 0.0f);
        const float inv_det = 1.0f / det;
        const V3 tvec = sub_nc(o, v0);
        const float u = dot_nc(tvec, pvec) * inv_det;
        const V3 qvec = cross_nc(tvec, e1);
        const float v = dot_nc(d, qvec) * inv_det;
        const float upv = u + v;
        const float t = dot_nc(e2, qvec) * inv_det;
        const uint32_t wid = __float_as_uint(a.w);
        const bool ok = ok_det & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (upv <= 1.0f) & (t > tmin) & (t < tmax);
        const bool closer = ok & ((t < h.t) | ((t == h.t) & (wid < h.world_id)));
        h.t = closer ? t : h.t; h.u = closer ? u : h.u; h.v = closer ? v : h.v; h.slot = closer ? slot : h.slot; h.world_id = closer ? wid : h.world_id;
        return closer;
    }
#endif
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
// KJ_WALK_PREFETCH (round 6 experiment): in a wave's TAIL -- at most KJ_WALK_PREFETCH_LANES live rays, where every step is one dependent round trip to L2 / the Infinity Cache
// on a nearly idle SIMD -- a node step requests the first bytes of the SECOND nearest child it pushes (the next thing popped when the nearest child's subtree is done), so that
// its line is in the L2 by then. The value is never used.
#ifndef KJ_WALK_PREFETCH
#define KJ_WALK_PREFETCH 0
#endif
#ifndef KJ_WALK_PREFETCH_LANES
#define KJ_WALK_PREFETCH_LANES 16u
#endif
struct TraverseStats { uint32_t nodes, tris; uint32_t wave_node_steps = 0, wave_tri_steps = 0; uint32_t live_hist[4] = {0, 0, 0, 0}; };   // live_hist: wave steps by the number of lanes still walking (1-8, 9-16, 17-32, 33-64), counted by one lane (round 6: profiles/r06_walk.md)   // wave_*: steps the WAVE issued (counted by one lane of it): lane utilisation of the walk = (nodes + tris) / (64 * steps)

typedef float f32x2 __attribute__((ext_vector_type(2)));
KJ_D float q8(uint32_t packed, int i) { return float((packed >> (8 * i)) & 0xffu); }   // v_cvt_f32_ubyte<i>

#define KJ_BVH_NONE 0xffffffffu
#if KJ_BVH_WIDTH != 4
#error "the traversal is written for the 4-wide node (an 8-wide variant measured 24 % slower in round 1 and was dropped)"
#endif

// One ray in flight. `cur` is the reference about to be visited (a node, a leaf's next triangle, or KJ_BVH_NONE = finished).
struct RayState {
    V3 wo, wd;          // the ray (world space, like every box and triangle of the tree)
    V3 binv;            // reciprocal direction for the slab tests
    float tmin, tmax;
    RayHit h;
    uint32_t sp, cur;
    bool cull_back;
#if KJ_WALK_PREFETCH
    uint32_t pf_prev = 0u, pf_acc = 0u;      // KJ_WALK_PREFETCH: the last requested word (never used), and where it ends up
    uint32_t pf_ref = 0xffffffffu;           // what the node step just taken wants requested
    bool pf_on = false;                      // wave-uniform: the wave is in its tail (few live rays: the step's round trip is what it waits for)
#endif
};
// reciprocal direction for the slab tests (v_rcp_f32: the boxes are conservative by several ulps, the triangles never see this)
KJ_D float rcp_box(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
KJ_D V3 safe_rcp3(V3 d) {
    const float eps = 1e-20f;
    return V3{rcp_box(fabsf(d.x) < eps ? copysignf(eps, d.x) : d.x), rcp_box(fabsf(d.y) < eps ? copysignf(eps, d.y) : d.y),
              rcp_box(fabsf(d.z) < eps ? copysignf(eps, d.z) : d.z)};
}

template <bool ANY_HIT>
KJ_D void ray_begin(RayState& S, V3 o, V3 d, float tmin, float tmax, bool cull_back) {
    S.wo = o; S.wd = d; S.tmin = tmin; S.tmax = tmax; S.cull_back = cull_back;
    S.h.t = FLT_MAX; S.h.u = 0; S.h.v = 0; S.h.slot = 0xffffffffu; S.h.world_id = 0xffffffffu;
    S.sp = 0;
    // Rays with a non-finite origin or direction are misses (the reference's validation pass issues such
    // rays for pixels without history; a hardware traversal unit rejects every box for them). Without
    // this, NaN slabs pass the fmin/fmax test and the whole tree is walked.
    const bool finite = fabsf(o.x) <= FLT_MAX && fabsf(o.y) <= FLT_MAX && fabsf(o.z) <= FLT_MAX && fabsf(d.x) <= FLT_MAX && fabsf(d.y) <= FLT_MAX && fabsf(d.z) <= FLT_MAX;
    S.cur = finite ? 0u : KJ_BVH_NONE;   // node 0 is the root
    S.binv = safe_rcp3(d);
}

// Traversal stack: the first KJ_BVH_LDS_STACK entries live in LDS ([level][lane]); the rare deeper ones spill to a
// private array (scratch). Near-first ordering keeps a typical ray's stack far below the builder's worst-case bound,
// so the LDS footprint (4 KB / wave) does not cap occupancy the way a bound-sized LDS stack did (11 KB / wave).
#define KJ_PUSH(v_) { const uint32_t pv_ = (v_); if (S.sp < KJ_BVH_LDS_STACK) stack[S.sp * stride] = pv_; else spill[S.sp - KJ_BVH_LDS_STACK] = pv_; S.sp++; }
// next reference off the stack
// (KJ_POP_DS, round 5: the LDS part is read unconditionally and the rare deep entries from the spill copy afterwards. Selecting between an LDS and a private POINTER
// made the load a flat_load -- generic address, counted on both memory counters -- in the middle of every step's dependent chain; this way it is a ds_read.)
#ifndef KJ_POP_DS
#define KJ_POP_DS 0      // measured on MI355X: the ds_read form is 4-5 % SLOWER on the rtdgi trace pass (0.290-0.295 against 0.275-0.281 ms at 1080p, 0.819 against 0.785 at 4K; profiles/r05_ray_pass_experiments.md)
#endif
#ifndef KJ_WALK_UNIFIED
#define KJ_WALK_UNIFIED 0    // measured: 24 % fewer iterations per wave and the trace pass 18 % SLOWER (0.317 against 0.268 ms at 1080p, 0.938 against 0.790 at 4K): what these kernels pay for is issued instructions
#endif
#ifndef KJ_WALK_UNIFIED_TAIL
#define KJ_WALK_UNIFIED_TAIL 0
#endif
#ifndef KJ_BVH_FOLD_INVD
#define KJ_BVH_FOLD_INVD 0
#endif
KJ_D void pop_next(RayState& S, uint32_t* stack, uint32_t stride, uint32_t* spill) {
    if (S.sp == 0) { S.cur = KJ_BVH_NONE; return; }
    --S.sp;
#if KJ_POP_DS && defined(__HIP_DEVICE_COMPILE__)
    // (an explicit LDS pointer and a compiler barrier on the value: left to itself the optimiser merges the two loads back into one flat_load of a selected address)
    uint32_t v = *((const __attribute__((address_space(3))) uint32_t*)stack + (S.sp < KJ_BVH_LDS_STACK ? S.sp : KJ_BVH_LDS_STACK - 1u) * stride);
    asm volatile("" : "+v"(v));
    if (S.sp >= KJ_BVH_LDS_STACK) v = spill[S.sp - KJ_BVH_LDS_STACK];
    S.cur = v;
#else
    S.cur = S.sp < KJ_BVH_LDS_STACK ? stack[S.sp * stride] : spill[S.sp - KJ_BVH_LDS_STACK];
#endif
}
#define KJ_POP(dst_) pop_next(S, stack, stride, spill);
// Visit the 4-wide node S.cur: test its four quantised child boxes, continue with the nearest hit child, push the others.
KJ_D bool wants_node_step(const RayState& S) { return S.cur != KJ_BVH_NONE && !(S.cur & KJ_BVH_LEAF); }
KJ_D bool wants_tri_step(const RayState& S) { return S.cur != KJ_BVH_NONE && (S.cur & KJ_BVH_LEAF); }
// (node_step_data / tri_step_data: the step on a node / triangle whose 64 / 48 bytes are in registers already; node_step / tri_step fetch them first)
template <bool ANY_HIT, bool STATS>
KJ_D void node_step_data(RayState& S, const float4 n0, const uint4 ch, const uint4 qa, const uint2 qb, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t NONE = KJ_BVH_NONE;
    if (STATS) stats->nodes++;
    const V3 o = S.wo, inv_d = S.binv;
    const float tmin = S.tmin;
    const bool neg_x = inv_d.x < 0.0f, neg_y = inv_d.y < 0.0f, neg_z = inv_d.z < 0.0f;
    const float tlimit = ANY_HIT ? S.tmax : fminf(S.h.t, S.tmax);
    const uint32_t e = __float_as_uint(n0.w);
    const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
    // The kernels that call this are VALU-bound, so the four slab tests are written for few instructions:
    //  * the ray's direction signs pick the near / far plane bytes once per node (6 selects) instead of a min/max per plane;
    //  * `origin - o` is folded into the decode: plane - o = fma(q, step, origin - o) (one more rounding than the builder's
    //    check, i.e. <= 1 ulp of the plane distance -- covered many times over by the slack on the far side below);
    //  * children are processed in pairs so the compiler can use packed-fp32 fma / mul (v_pk_fma_f32, v_pk_mul_f32).
    const uint32_t nqx = neg_x ? qa.w : qa.x, fqx = neg_x ? qa.x : qa.w;
    const uint32_t nqy = neg_y ? qb.x : qa.y, fqy = neg_y ? qa.y : qb.x;
    const uint32_t nqz = neg_z ? qb.y : qa.z, fqz = neg_z ? qa.z : qb.y;
#if KJ_BVH_FOLD_INVD
    // the reciprocal direction folded into the decode: t = fma(q, step / d, (origin - o) / d) -- six multiplies per node instead of one per plane (experiment, round 5)
    const float sxi = sx * inv_d.x, syi = sy * inv_d.y, szi = sz * inv_d.z;
    const float bx = (n0.x - o.x) * inv_d.x, by = (n0.y - o.y) * inv_d.y, bz = (n0.z - o.z) * inv_d.z;
    uint32_t key[4];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int i0 = pr * 2, i1 = pr * 2 + 1;
        const f32x2 tnx = __builtin_elementwise_fma(f32x2{q8(nqx, i0), q8(nqx, i1)}, f32x2{sxi, sxi}, f32x2{bx, bx});
        const f32x2 tny = __builtin_elementwise_fma(f32x2{q8(nqy, i0), q8(nqy, i1)}, f32x2{syi, syi}, f32x2{by, by});
        const f32x2 tnz = __builtin_elementwise_fma(f32x2{q8(nqz, i0), q8(nqz, i1)}, f32x2{szi, szi}, f32x2{bz, bz});
        const f32x2 tfx = __builtin_elementwise_fma(f32x2{q8(fqx, i0), q8(fqx, i1)}, f32x2{sxi, sxi}, f32x2{bx, bx});
        const f32x2 tfy = __builtin_elementwise_fma(f32x2{q8(fqy, i0), q8(fqy, i1)}, f32x2{syi, syi}, f32x2{by, by});
        const f32x2 tfz = __builtin_elementwise_fma(f32x2{q8(fqz, i0), q8(fqz, i1)}, f32x2{szi, szi}, f32x2{bz, bz});
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = pr * 2 + k;
            const float tn = fmaxf(fmaxf(fmaxf(tnx[k], tny[k]), tnz[k]), tmin);
            const float tf = fminf(fminf(fminf(tfx[k], tfy[k]), tfz[k]), tlimit);
            const uint32_t c = i == 0 ? ch.x : (i == 1 ? ch.y : (i == 2 ? ch.z : ch.w));
            const bool hit = (tn <= tf * 1.000001f + 1e-30f) && c != NONE;
            key[i] = hit ? __float_as_uint(tn) : NONE;
        }
    }
#else
    const float bx = n0.x - o.x, by = n0.y - o.y, bz = n0.z - o.z;
    const float fx = bx, fy = by, fz = bz;
    uint32_t key[4];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int i0 = pr * 2, i1 = pr * 2 + 1;
        const f32x2 tnx = __builtin_elementwise_fma(f32x2{q8(nqx, i0), q8(nqx, i1)}, f32x2{sx, sx}, f32x2{bx, bx}) * f32x2{inv_d.x, inv_d.x};
        const f32x2 tny = __builtin_elementwise_fma(f32x2{q8(nqy, i0), q8(nqy, i1)}, f32x2{sy, sy}, f32x2{by, by}) * f32x2{inv_d.y, inv_d.y};
        const f32x2 tnz = __builtin_elementwise_fma(f32x2{q8(nqz, i0), q8(nqz, i1)}, f32x2{sz, sz}, f32x2{bz, bz}) * f32x2{inv_d.z, inv_d.z};
        const f32x2 tfx = __builtin_elementwise_fma(f32x2{q8(fqx, i0), q8(fqx, i1)}, f32x2{sx, sx}, f32x2{fx, fx}) * f32x2{inv_d.x, inv_d.x};
        const f32x2 tfy = __builtin_elementwise_fma(f32x2{q8(fqy, i0), q8(fqy, i1)}, f32x2{sy, sy}, f32x2{fy, fy}) * f32x2{inv_d.y, inv_d.y};
        const f32x2 tfz = __builtin_elementwise_fma(f32x2{q8(fqz, i0), q8(fqz, i1)}, f32x2{sz, sz}, f32x2{fz, fz}) * f32x2{inv_d.z, inv_d.z};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = pr * 2 + k;
            const float tn = fmaxf(fmaxf(fmaxf(tnx[k], tny[k]), tnz[k]), tmin);
            const float tf = fminf(fminf(fminf(tfx[k], tfy[k]), tfz[k]), tlimit);
            // conservative acceptance (a few ulps of slack on the far side); empty slots hold an inverted box and a NONE reference
            const uint32_t c = i == 0 ? ch.x : (i == 1 ? ch.y : (i == 2 ? ch.z : ch.w));
            const bool hit = (tn <= tf * 1.000001f + 1e-30f) && c != NONE;
            key[i] = hit ? __float_as_uint(tn) : NONE;   // tn >= tmin >= 0: float order == integer order
        }
    }
#endif
    // Branch-free tail: (key, reference) pairs go through a 5-comparator network as selects and the three farther children are
    // stored to the LDS stack unconditionally, the stack pointer advancing only for hits (sorted order puts the hits first; a
    // non-hit's store is overwritten by the next push or ignored).
    uint32_t ref[4] = {ch.x, ch.y, ch.z, ch.w};
#define KJ_CSWAP(a, b) { const bool sw_ = key[b] < key[a]; const uint32_t ka_ = key[a], kb_ = key[b], ra_ = ref[a], rb_ = ref[b]; \
                         key[a] = sw_ ? kb_ : ka_; key[b] = sw_ ? ka_ : kb_; ref[a] = sw_ ? rb_ : ra_; ref[b] = sw_ ? ra_ : rb_; }
    // closest-hit rays visit children nearest first; occlusion rays take them in slot order: any hit ends the ray, the push logic below
    // is order-agnostic, and skipping the network measured +8 % any-hit rays/s for a few more node visits
    if (!ANY_HIT) { KJ_CSWAP(0, 1) KJ_CSWAP(2, 3) KJ_CSWAP(0, 2) KJ_CSWAP(1, 3) KJ_CSWAP(1, 2) }
#undef KJ_CSWAP
    if (S.sp + 3u <= KJ_BVH_LDS_STACK) {
        stack[S.sp * stride] = ref[3]; S.sp += key[3] != NONE ? 1u : 0u;
        stack[S.sp * stride] = ref[2]; S.sp += key[2] != NONE ? 1u : 0u;
        stack[S.sp * stride] = ref[1]; S.sp += key[1] != NONE ? 1u : 0u;
    } else {   // deep lanes: entries beyond the LDS part spill to private memory
        if (key[3] != NONE) KJ_PUSH(ref[3])
        if (key[2] != NONE) KJ_PUSH(ref[2])
        if (key[1] != NONE) KJ_PUSH(ref[1])
    }
    if (key[0] != NONE) S.cur = ref[0];
    else KJ_POP(S.cur)
#if KJ_WALK_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
    S.pf_ref = (S.pf_on && key[1] != NONE) ? ref[1] : NONE;
#endif
}
template <bool ANY_HIT, bool STATS>
KJ_D void node_step(const BvhView& bvh, RayState& S, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const float4* __restrict__ n = (const float4*)bvh.nodes + size_t(S.cur) * 4;
    float4 n0 = n[0]; const uint4 ch = *(const uint4*)(n + 1), qa = *(const uint4*)(n + 2); const uint2 qb = *(const uint2*)(n + 3);
#if KJ_WALK_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(S.pf_prev), "+v"(n0.x));      // the previous request is consumed behind this node's own (younger) loads: loads return in order
    S.pf_acc ^= S.pf_prev;
#endif
    node_step_data<ANY_HIT, STATS>(S, n0, ch, qa, qb, stack, stride, spill, stats);
#if KJ_WALK_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
    if (S.pf_ref != KJ_BVH_NONE) {
        const uint32_t c = S.pf_ref;
        const uintptr_t pfp = (c & KJ_BVH_LEAF) ? uintptr_t(bvh.tris) + size_t(c & 0x0fffffffu) * 48u : uintptr_t(bvh.nodes) + size_t(c) * 64u;
        S.pf_prev = *(const __attribute__((address_space(1))) uint32_t*)pfp;
    }
#endif
}
template <bool ANY_HIT, bool STATS>
KJ_D void tri_step_data(RayState& S, const float4 a, const float4 b, const float4 c, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t first = S.cur & 0x0fffffffu, rest = (S.cur >> 28) & 7u;
    if (STATS) stats->tris += 1u;
    const bool hit = intersect_tri(S.wo, S.wd, S.tmin, S.tmax, a, b, c, first, S.cull_back, S.h);
    if (ANY_HIT && hit) { S.cur = KJ_BVH_NONE; return; }
    if (rest != 0u) S.cur = KJ_BVH_LEAF | ((rest - 1u) << 28) | (first + 1u);
    else KJ_POP(S.cur)
}

// Test the next triangle(s) of the leaf S.cur; an occlusion ray that hits is finished. KJ_TRI_PAIR (experiment, round 4): a leaf with two or
// more triangles left tests TWO per step -- both fetches issued before the first test -- so a 4-triangle leaf is two dependent round
// trips instead of four. The closest hit (ties to the lowest world id) does not depend on how many triangles a step takes.
#ifndef KJ_TRI_PAIR
#define KJ_TRI_PAIR 0
#endif
template <bool ANY_HIT, bool STATS>
KJ_D void tri_step(const BvhView& bvh, RayState& S, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t first = S.cur & 0x0fffffffu;
    uint32_t rest = (S.cur >> 28) & 7u;      // triangles left after this one
    const uint32_t slot = first;
    const float4* __restrict__ tp = (const float4*)bvh.tris + size_t(slot) * 3;
    const float4 a = tp[0], b = tp[1], c = tp[2];
#if KJ_TRI_PAIR
    const uint32_t two = rest != 0u ? 1u : 0u;
    const float4* __restrict__ tq = tp + two * 3u;      // the same triangle again when there is no second one: the load is always in range
    const float4 a2 = tq[0], b2 = tq[1], c2 = tq[2];
#else
    const uint32_t two = 0u;
#endif
    if (STATS) stats->tris += 1u + two;
    bool hit = intersect_tri(S.wo, S.wd, S.tmin, S.tmax, a, b, c, slot, S.cull_back, S.h);
#if KJ_TRI_PAIR
    if (two && !(ANY_HIT && hit)) hit |= intersect_tri(S.wo, S.wd, S.tmin, S.tmax, a2, b2, c2, slot + 1u, S.cull_back, S.h);
#endif
    if (ANY_HIT && hit) { S.cur = KJ_BVH_NONE; return; }
    if (rest > two) S.cur = KJ_BVH_LEAF | ((rest - 1u - two) << 28) | (first + 1u + two);
    else KJ_POP(S.cur)
}

// The same step for a wave that carries closest-hit and occlusion rays side by side (the pool form of the rtdgi ray passes): `any_hit` is the lane's.
// (Such a wave visits every node's children nearest first, node_step<false>: an occlusion ray's answer does not depend on the order.)
template <bool STATS>
KJ_D void tri_step_mixed(const BvhView& bvh, RayState& S, bool any_hit, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t first = S.cur & 0x0fffffffu;
    const uint32_t rest = (S.cur >> 28) & 7u;
    const float4* __restrict__ tp = (const float4*)bvh.tris + size_t(first) * 3;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    if (STATS) stats->tris += 1u;
    const bool hit = intersect_tri(S.wo, S.wd, S.tmin, S.tmax, a, b, c, first, S.cull_back, S.h);
    if (any_hit && hit) { S.cur = KJ_BVH_NONE; return; }
    if (rest != 0u) S.cur = KJ_BVH_LEAF | ((rest - 1u) << 28) | (first + 1u);
    else KJ_POP(S.cur)
}

#if defined(__HIP_DEVICE_COMPILE__)
template <bool ANY_HIT, bool STATS, uint32_t CAP>
KJ_D void quad_walk(const BvhView& bvh, RayState& S, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats);      // below: four lanes per ray
#endif
// KJ_WALK_TAIL_QUAD (round 6 experiment, profiles/r06_walk.md): 54 % of the wave steps of the closest-hit walks and 68 % of the occlusion walks' run with at most 16 of
// the wave's 64 lanes still holding a ray. With the switch on, a wave whose live rays have dropped to KJ_WALK_TAIL_LANES or fewer (and whose stacks are all within
// their LDS columns) hands each of them to a QUAD of lanes -- state broadcast from the ray's lane, the quad working on that lane's LDS stack column -- finishes them
// with the four-lanes-per-ray step (one child box / one triangle per lane: ~75 instructions instead of ~200, a leaf per step instead of a triangle), and returns
// the hits to the rays' lanes. Same steps per ray as far as results go: the closest hit with ties to the lowest world triangle id does not depend on visiting order.
#ifndef KJ_WALK_TAIL_QUAD
#define KJ_WALK_TAIL_QUAD 0
#endif
#ifndef KJ_WALK_TAIL_LANES
#define KJ_WALK_TAIL_LANES 16u
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// bit 4q of the result is set when all four lanes of quad q are in the call (a caller's sky pixels have left the kernel, a shadow ray is traced under `if (hit)`:
// only whole quads can carry a ray)
KJ_D unsigned long long walk_full_quads() {
    const unsigned long long act = __ballot(true);
    return act & (act >> 1) & (act >> 2) & (act >> 3) & 0x1111111111111111ull;
}
KJ_D int walk_nth_set_bit(unsigned long long m, uint32_t n) {      // position of the n-th (0-based) set bit of m, or -1
    for (uint32_t i = 0; i < n; ++i) m &= m - 1ull;
    return m ? __ffsll((long long)m) - 1 : -1;
}
template <bool ANY_HIT, bool STATS>
KJ_D void walk_tail_on_quads(const BvhView& bvh, RayState& S, bool alive, unsigned long long fq, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t lane = __lane_id() & 63u, q = lane >> 2;
    const unsigned long long m = __ballot(alive);
    // this lane's quad, if whole, is the qr-th whole quad and serves the qr-th live ray
    const bool whole = ((fq >> (4u * q)) & 1ull) != 0ull;
    const uint32_t qr = uint32_t(__popcll(fq & ((1ull << (4u * q)) - 1ull)));
    const int src_ = whole ? walk_nth_set_bit(m, qr) : -1;
    const bool has = src_ >= 0;
    const int src = has ? src_ : int(lane);
    RayState T;
    T.wo = V3{__shfl(S.wo.x, src), __shfl(S.wo.y, src), __shfl(S.wo.z, src)};
    T.wd = V3{__shfl(S.wd.x, src), __shfl(S.wd.y, src), __shfl(S.wd.z, src)};
    T.binv = V3{__shfl(S.binv.x, src), __shfl(S.binv.y, src), __shfl(S.binv.z, src)};
    T.tmin = __shfl(S.tmin, src); T.tmax = __shfl(S.tmax, src);
    T.h.t = __shfl(S.h.t, src); T.h.u = __shfl(S.h.u, src); T.h.v = __shfl(S.h.v, src); T.h.slot = __shfl(S.h.slot, src); T.h.world_id = __shfl(S.h.world_id, src);
    T.sp = __shfl(S.sp, src); T.cur = __shfl(S.cur, src);
    T.cull_back = __shfl(uint32_t(S.cull_back ? 1u : 0u), src) != 0u;
    const uint32_t lo = uint32_t(uintptr_t(stack)), hi = uint32_t(uintptr_t(stack) >> 32);
    uint32_t* const col = (uint32_t*)(uintptr_t(__shfl(lo, src)) | (uintptr_t(__shfl(hi, src)) << 32));      // the ray's own stack column
    if (!has) T.cur = KJ_BVH_NONE;
    // (`spill`, the caller's: only entries pushed from here on can land in it -- every live stack was within its LDS column at the hand-over)
    quad_walk<ANY_HIT, STATS, KJ_BVH_LDS_STACK>(bvh, T, col, stride, spill, stats);
    // the hits go back to the rays' lanes: live lane L, the r-th live one, was served by the r-th whole quad
    const uint32_t r = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
    const int from_ = alive ? walk_nth_set_bit(fq, r) : -1;
    const int from = from_ >= 0 ? from_ : int(lane);
    const float ht = __shfl(T.h.t, from), hu = __shfl(T.h.u, from), hv = __shfl(T.h.v, from);
    const uint32_t hs = __shfl(T.h.slot, from), hw = __shfl(T.h.world_id, from);
    if (alive) { S.h.t = ht; S.h.u = hu; S.h.v = hv; S.h.slot = hs; S.h.world_id = hw; S.cur = KJ_BVH_NONE; S.sp = 0; }
}
#endif

// One ray per lane, start to finish, inside a caller's kernel. The lanes of the wave that are in the call step together: each
// wave step issues EITHER the node block or the triangle block, whichever more lanes are waiting for (triangle lanes count
// double: their block is the cheaper one), instead of a mixed wave paying for both blocks in every iteration. Per-ray results do
// not depend on the interleaving. (On the tests' CPU stand-in for HIP lanes run one at a time and each simply loops by itself.)
template <bool ANY_HIT, bool STATS = false>
KJ_D RayHit bvh_trace(const BvhView& bvh, V3 o, V3 d, float tmin, float tmax, bool cull_back, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
    RayState S;
    ray_begin<ANY_HIT>(S, o, d, tmin, tmax, cull_back);
    uint32_t spill[KJ_BVH_SPILL_STACK];
#if defined(__HIP_DEVICE_COMPILE__) && KJ_WALK_UNIFIED
    // KJ_WALK_UNIFIED (round 5): every lane with a ray advances in EVERY iteration -- ONE fetch serves both kinds of lanes (a node's 64 bytes or a triangle's 48 from
    // the lane's own address), then the node block runs for the lanes at a node and the triangle block for the lanes at a leaf. An iteration issues both blocks (when
    // both kinds of lanes exist) but a wave needs max over its lanes of (nodes + triangles) iterations instead of the sum of the two votes' rounds: for launches that
    // are a single round of waves bound by their dependent chain, not by issue. Per-ray results are unchanged (same steps in the same order for each ray).
    uint32_t it_node = 0;
    for (;;) {
        const bool want_node = wants_node_step(S), want_tri = wants_tri_step(S);
#if KJ_WALK_UNIFIED_TAIL
        // KJ_WALK_UNIFIED_TAIL = n (round 6 experiment): the per-wave vote (one kind of step per iteration) while more than n rays are alive, every live lane in every iteration below
        const uint32_t nn = uint32_t(__popcll(__ballot(want_node))), nt = uint32_t(__popcll(__ballot(want_tri)));
        if (nn + nt == 0u) break;
        const bool vote_node = nt == 0u || (nn != 0u && nn >= nt * 2u);
        const bool go = nn + nt <= uint32_t(KJ_WALK_UNIFIED_TAIL) ? (want_node | want_tri) : (vote_node ? want_node : want_tri);
#else
        if (__ballot(want_node | want_tri) == 0ull) break;
        const bool go = want_node | want_tri;
#endif
        if (STATS) it_node++;
        if (go) {
            const float4* __restrict__ p = want_node ? (const float4*)bvh.nodes + size_t(S.cur) * 4 : (const float4*)bvh.tris + size_t(S.cur & 0x0fffffffu) * 3;
            const float4 d0 = p[0], d1 = p[1], d2 = p[2];
            if (want_node) {
                const uint2 qb = *(const uint2*)(p + 3);
                node_step_data<ANY_HIT, STATS>(S, d0, make_uint4(__float_as_uint(d1.x), __float_as_uint(d1.y), __float_as_uint(d1.z), __float_as_uint(d1.w)),
                                               make_uint4(__float_as_uint(d2.x), __float_as_uint(d2.y), __float_as_uint(d2.z), __float_as_uint(d2.w)), qb, stack, stride, spill, stats);
            } else tri_step_data<ANY_HIT, STATS>(S, d0, d1, d2, stack, stride, spill, stats);
        }
    }
    if (STATS && (__ffsll((long long)__ballot(true)) - 1) == int(__lane_id())) stats->wave_node_steps += it_node;
#elif defined(__HIP_DEVICE_COMPILE__)
    uint32_t it_node = 0, it_tri = 0;
    uint32_t live_hist[4] = {0, 0, 0, 0};
    for (;;) {
        const bool want_node = wants_node_step(S), want_tri = wants_tri_step(S);
        const uint32_t nn = uint32_t(__popcll(__ballot(want_node))), nt = uint32_t(__popcll(__ballot(want_tri)));
        if (nn + nt == 0u) break;
#if KJ_WALK_TAIL_QUAD
        if (nn + nt <= KJ_WALK_TAIL_LANES) {
            const bool alive = want_node | want_tri;
            const unsigned long long fq = walk_full_quads();
            // every live ray needs a whole quad of lanes that are in this call; a live ray whose stack has spilled to private memory cannot move (try again once it has popped back)
            if (nn + nt <= uint32_t(__popcll(fq)) && __ballot(alive && S.sp > KJ_BVH_LDS_STACK) == 0ull) {
                walk_tail_on_quads<ANY_HIT, STATS>(bvh, S, alive, fq, stack, stride, spill, stats);
                break;
            }
        }
#endif
        if (STATS) { const uint32_t live = nn + nt; live_hist[live <= 8u ? 0 : (live <= 16u ? 1 : (live <= 32u ? 2 : 3))]++; }
#if KJ_WALK_PREFETCH
        S.pf_on = nn + nt <= KJ_WALK_PREFETCH_LANES;
#endif
        if (nt == 0u || (nn != 0u && nn >= nt * 2u)) { if (STATS) it_node++; if (want_node) node_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats); }
        else { if (STATS) it_tri++; if (want_tri) tri_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats); }
    }
    if (STATS && (__ffsll((long long)__ballot(true)) - 1) == int(__lane_id())) {
        stats->wave_node_steps += it_node; stats->wave_tri_steps += it_tri;
        for (int b = 0; b < 4; ++b) stats->live_hist[b] += live_hist[b];
    }
#if KJ_WALK_PREFETCH
    S.pf_acc ^= S.pf_prev;
    if (S.pf_acc == 0x9e3779b9u && __float_as_uint(S.tmax) == 0xffffffffu) S.h.u = 2.0f;      // never true: keeps the requests from being optimised away
#endif
#else
    while (S.cur != KJ_BVH_NONE) {
        if (wants_node_step(S)) node_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats);
        else tri_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats);
    }
#endif
    return S.h;
}

// ---- four lanes per ray ("quad"): for launches too small to fill the chip with one ray per lane (the irradiance cache's ray passes:
// ~30 k rays = 470 waves on 1024 SIMDs, each a chain of ~60 dependent ~200-instruction steps). The four lanes 4q .. 4q+3 of a wave carry
// the SAME ray; at a node lane k tests child k's box, at a leaf it tests triangle k (and k + 4); the four results meet through DPP
// quad permutes (one VALU instruction each, no LDS). A step is ~75 instructions instead of ~200, a wave waits for the slowest of 16
// rays instead of 64, and the launch has four times the waves. The caller runs its shading code on all four lanes (same inputs, same
// results) and lets lane 0 of each quad perform the side effects. Per-ray results are those of bvh_trace(): the closest hit with
// equal-t ties going to the lowest world triangle id does not depend on the order boxes and triangles are visited in.
#ifndef KJ_QUAD_PREFETCH
#define KJ_QUAD_PREFETCH 1      // measured on MI355X (round 6, profiles/r06_walk.md): k_irc_ray_chain 152.9 -> 145.3 us on the 4 M-triangle scene (-5 %), the frames unchanged
#endif
#define KJ_QUAD_LDS_STACK 32u     // stack entries per quad kept in LDS ([level][quad]: 16 quads x 32 levels x 4 B = 2 KB per wave); deeper ones spill
KJ_HD size_t quad_stack_bytes() { return size_t(KJ_QUAD_LDS_STACK) * 16u * 4u; }
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL> KJ_D uint32_t quad_perm(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xf, 0xf, false)); }
template <int CTRL> KJ_D float quad_perm(float v) { return __uint_as_float(quad_perm<CTRL>(__float_as_uint(v))); }
#define KJ_QP_ROT1 0x39   // quad_perm:[1,2,3,0]: lane k reads lane (k + 1) & 3
#define KJ_QP_ROT2 0x4E   // [2,3,0,1]
#define KJ_QP_ROT3 0x93   // [3,0,1,2]
#define KJ_QP_XOR1 0xB1   // [1,0,3,2]
#define KJ_QP_BCAST0 0x00 // [0,0,0,0]
KJ_D float quad_broadcast0(float v) { return quad_perm<KJ_QP_BCAST0>(v); }
KJ_D uint32_t quad_broadcast0(uint32_t v) { return quad_perm<KJ_QP_BCAST0>(v); }
#else
KJ_D float quad_broadcast0(float v) { return v; }       // the tests' CPU stand-in runs one lane at a time: every lane computes everything itself
KJ_D uint32_t quad_broadcast0(uint32_t v) { return v; }
#endif
KJ_D V3 quad_broadcast0(V3 v) { return V3{quad_broadcast0(v.x), quad_broadcast0(v.y), quad_broadcast0(v.z)}; }

// next reference off a quad's stack: the LDS part is read unconditionally (one ds_read, no generic pointer), the rare deep entries from the spill copy
template <uint32_t CAP = KJ_QUAD_LDS_STACK>
KJ_D void quad_pop(RayState& S, const uint32_t* stack, uint32_t stride, const uint32_t* spill) {
    if (S.sp == 0) { S.cur = KJ_BVH_NONE; return; }
    --S.sp;
    uint32_t v = stack[(S.sp < CAP ? S.sp : CAP - 1u) * stride];
    if (S.sp >= CAP) v = spill[S.sp - CAP];
    S.cur = v;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The walk of a ray carried by the four lanes 4q .. 4q+3 (identical S in all four; `stack`: the LDS column of its stack, CAP levels deep; deeper entries in `spill`).
template <bool ANY_HIT, bool STATS, uint32_t CAP>
KJ_D void quad_walk(const BvhView& bvh, RayState& S, uint32_t* stack, uint32_t stride, uint32_t* spill, TraverseStats* stats) {
    const uint32_t NONE = KJ_BVH_NONE;
    const uint32_t k = __lane_id() & 3u;
    const V3 inv_d = S.binv;
    const bool neg_x = inv_d.x < 0.0f, neg_y = inv_d.y < 0.0f, neg_z = inv_d.z < 0.0f;
#if KJ_QUAD_PREFETCH
    uint32_t pf_prev = 0u, pf_acc = 0u;
#endif
    for (;;) {
        const bool want_node = wants_node_step(S), want_tri = wants_tri_step(S);
        const uint32_t nn = uint32_t(__popcll(__ballot(want_node))), nt = uint32_t(__popcll(__ballot(want_tri)));
        if (nn + nt == 0u) break;
        if (nt == 0u || (nn != 0u && nn >= nt * 2u)) {
            if (want_node) {
                const float4* __restrict__ n = (const float4*)bvh.nodes + size_t(S.cur) * 4;
                float4 n0 = n[0]; const uint4 ch = *(const uint4*)(n + 1), qa = *(const uint4*)(n + 2); const uint2 qb = *(const uint2*)(n + 3);
#if KJ_QUAD_PREFETCH
                // the previous step's request is consumed HERE, behind this node's own (younger) loads: loads return in order, so the wait for n0 has covered it
                asm volatile("" : "+v"(pf_prev), "+v"(n0.x));
                pf_acc ^= pf_prev;
#endif
                if (STATS && k == 0u) stats->nodes++;
                const float tlimit = ANY_HIT ? S.tmax : fminf(S.h.t, S.tmax);
                const uint32_t e = __float_as_uint(n0.w);
                const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
                const uint32_t sh = k * 8u;
                const float nqx = float(((neg_x ? qa.w : qa.x) >> sh) & 0xffu), fqx = float(((neg_x ? qa.x : qa.w) >> sh) & 0xffu);
                const float nqy = float(((neg_y ? qb.x : qa.y) >> sh) & 0xffu), fqy = float(((neg_y ? qa.y : qb.x) >> sh) & 0xffu);
                const float nqz = float(((neg_z ? qb.y : qa.z) >> sh) & 0xffu), fqz = float(((neg_z ? qa.z : qb.y) >> sh) & 0xffu);
                const float bx = n0.x - S.wo.x, by = n0.y - S.wo.y, bz = n0.z - S.wo.z;
                const float tnx = fmaf(nqx, sx, bx) * inv_d.x, tny = fmaf(nqy, sy, by) * inv_d.y, tnz = fmaf(nqz, sz, bz) * inv_d.z;
                const float tfx = fmaf(fqx, sx, bx) * inv_d.x, tfy = fmaf(fqy, sy, by) * inv_d.y, tfz = fmaf(fqz, sz, bz) * inv_d.z;
                const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), S.tmin);
                const float tf = fminf(fminf(fminf(tfx, tfy), tfz), tlimit);
                uint32_t c = ch.x;
                c = k == 1u ? ch.y : c; c = k == 2u ? ch.z : c; c = k == 3u ? ch.w : c;
                const bool hit = (tn <= tf * 1.000001f + 1e-30f) & (c != NONE);
#if KJ_QUAD_PREFETCH
                // KJ_QUAD_PREFETCH (round 6): a lane whose child the ray enters requests the child's first bytes NOW -- the node, or a leaf's first triangle --, before the
                // sort and the pushes: these launches are lone waves bound by the round trip of every step's dependent load; the nearest child's line is then on its way (or there)
                // when the next step asks for it, the pushed ones' when they are popped. The value is never used (xor-ed into a word nothing reads out).
                if (hit) {
                    const uintptr_t pfp = (c & KJ_BVH_LEAF) ? uintptr_t(bvh.tris) + size_t(c & 0x0fffffffu) * 48u : uintptr_t(bvh.nodes) + size_t(c) * 64u;
                    pf_prev = *(const __attribute__((address_space(1))) uint32_t*)pfp;      // (a global_load: a generic pointer would make it a flat_load, counted on the LDS counter too)
                }
#endif
                // key = entry distance with the child index in its two low bits (all four keys distinct; tn >= tmin >= 0: float order ==
                // integer order); occlusion rays take the children in slot order. (key, reference) pairs of the four lanes travel
                // through quad rotations and a 5-comparator network of selects: every lane ends up with the same sorted list.
                const uint32_t key = hit ? ((ANY_HIT ? 0u : (__float_as_uint(tn) & ~3u)) | k) : NONE;
                uint32_t s0 = key, s1 = quad_perm<KJ_QP_ROT1>(key), s2 = quad_perm<KJ_QP_ROT2>(key), s3 = quad_perm<KJ_QP_ROT3>(key);
                uint32_t r0 = c, r1 = quad_perm<KJ_QP_ROT1>(c), r2 = quad_perm<KJ_QP_ROT2>(c), r3 = quad_perm<KJ_QP_ROT3>(c);
#define KJ_QSWAP(a, b, ra, rb) { const bool sw_ = b < a; const uint32_t ka_ = a, kb_ = b, pa_ = ra, pb_ = rb; a = sw_ ? kb_ : ka_; b = sw_ ? ka_ : kb_; ra = sw_ ? pb_ : pa_; rb = sw_ ? pa_ : pb_; }
                KJ_QSWAP(s0, s1, r0, r1) KJ_QSWAP(s2, s3, r2, r3) KJ_QSWAP(s0, s2, r0, r2) KJ_QSWAP(s1, s3, r1, r3) KJ_QSWAP(s1, s2, r1, r2)
#undef KJ_QSWAP
                const uint32_t pushes = (s1 != NONE ? 1u : 0u) + (s2 != NONE ? 1u : 0u) + (s3 != NONE ? 1u : 0u);      // hits sort first: s0 is a hit when any is
                if (S.sp + 3u <= CAP) {
                    // lane j (1..3) stores the j-th nearest child, the farthest deepest; a non-hit's slot lies beyond the new stack pointer
                    const uint32_t mine = k == 1u ? r1 : (k == 2u ? r2 : r3);
                    if ((k != 0u) & (k <= pushes)) stack[(S.sp + pushes - k) * stride] = mine;
                    __builtin_amdgcn_wave_barrier();      // the pops read what OTHER lanes of the quad stored (LDS is in order per wave; this pins the compiler)
                    S.sp += pushes;
                } else {
                    const uint32_t okey[3] = {s3, s2, s1}, oref[3] = {r3, r2, r1};
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (okey[j] != NONE) {
                            if (S.sp < CAP) { if (k == 0u) stack[S.sp * stride] = oref[j]; } else spill[S.sp - CAP] = oref[j];
                            S.sp++;
                        }
                }
                if (s0 != NONE) S.cur = r0;
                else quad_pop<CAP>(S, stack, stride, spill);
            }
        } else if (want_tri) {
            const uint32_t first = S.cur & 0x0fffffffu;
            const uint32_t count = ((S.cur >> 28) & 7u) + 1u;
            bool any = false;
            for (uint32_t base = 0; base < count; base += 4u) {
                const uint32_t i = base + k;
                if (i < count) {
                    const uint32_t slot = first + i;
                    const float4* __restrict__ tp = (const float4*)bvh.tris + size_t(slot) * 3;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    if (STATS) stats->tris++;
                    any |= intersect_tri(S.wo, S.wd, S.tmin, S.tmax, a, b, c, slot, S.cull_back, S.h);
                }
            }
            // the quad's best candidate: (t, world id) lexicographic minimum, two exchange rounds
#define KJ_QMIN(CTRL) { const float t_ = quad_perm<CTRL>(S.h.t), u_ = quad_perm<CTRL>(S.h.u), v_ = quad_perm<CTRL>(S.h.v); \
                        const uint32_t sl_ = quad_perm<CTRL>(S.h.slot), w_ = quad_perm<CTRL>(S.h.world_id);                    \
                        const bool take_ = (t_ < S.h.t) | ((t_ == S.h.t) & (w_ < S.h.world_id));                                      \
                        S.h.t = take_ ? t_ : S.h.t; S.h.u = take_ ? u_ : S.h.u; S.h.v = take_ ? v_ : S.h.v; S.h.slot = take_ ? sl_ : S.h.slot; S.h.world_id = take_ ? w_ : S.h.world_id; }
            KJ_QMIN(KJ_QP_XOR1) KJ_QMIN(KJ_QP_ROT2)
#undef KJ_QMIN
            if (ANY_HIT && S.h.slot != 0xffffffffu) S.cur = NONE;
            else quad_pop<CAP>(S, stack, stride, spill);
            (void)any;
        }
    }
#if KJ_QUAD_PREFETCH
    pf_acc ^= pf_prev;
    if (pf_acc == 0x9e3779b9u && __float_as_uint(S.tmax) == 0xffffffffu) S.h.u = 2.0f;      // never true (tmax is a finite distance): keeps the requests from being optimised away
#endif
}
#endif
// `stack`: LDS base of this QUAD's stack (entries at stack[level * stride]); all four lanes pass the same ray and the same `active`.
template <bool ANY_HIT, bool STATS = false>
KJ_D RayHit bvh_trace_quad(const BvhView& bvh, bool active, V3 o, V3 d, float tmin, float tmax, bool cull_back, uint32_t* stack, uint32_t stride, TraverseStats* stats = nullptr) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (!active) { RayHit h; h.t = FLT_MAX; h.u = h.v = 0; h.slot = h.world_id = 0xffffffffu; return h; }
    uint32_t own[KJ_BVH_LDS_STACK];      // the stand-in's lanes run one after the other: each walks the ray alone, on a private stack
    return bvh_trace<ANY_HIT, STATS>(bvh, o, d, tmin, tmax, cull_back, own, 1, stats);
#else
    RayState S;
    ray_begin<ANY_HIT>(S, o, d, tmin, tmax, cull_back);
    if (!active) S.cur = KJ_BVH_NONE;
    uint32_t spill[KJ_BVH_SPILL_STACK];   // identical in the four lanes (every lane pushes every entry)
    quad_walk<ANY_HIT, STATS, KJ_QUAD_LDS_STACK>(bvh, S, stack, stride, spill, stats);
    return S.h;
#endif
}

// ---- ray streams: a persistent wave works through a dense array of rays, keeping all 64 lanes busy.
//  * rays[i] = {origin.xyz, tmin}, {direction.xyz, tmax}; tmax < 0 marks "no ray here" (a sky pixel's slot): result = miss.
//  * A wave owns chunks of KJ_STREAM_CHUNK consecutive rays, interleaved with the other waves of the launch (no atomics). A lane
//    whose ray has finished parks; once a quarter of the wave is parked, all parked lanes store their results and take the next
//    rays of the chunk in one go (wave vote + prefix count) -- so one long ray no longer holds 63 finished lanes hostage.
//  * A wave step runs EITHER the node block or the triangle block, whichever the larger share of its lanes is waiting for; lanes
//    waiting for the other block sit the step out. In a single-ray-per-lane loop a mixed wave pays for both blocks every
//    iteration; here every issued block works for the majority of the lanes.
//  Per-ray results are identical to bvh_trace() (same steps in the same order for each ray; only the interleaving differs).
#define KJ_STREAM_CHUNK 256u     // rays a wave takes at a time; a launch shrinks it (down to 64) when that is needed to give every wave slot a chunk
// scheduling knobs of a stream launch (uniform): lanes parked before a refill; a step runs the node block when nodes * node_weight >= tris * tri_weight
struct StreamTune { uint32_t refill_threshold, node_weight, tri_weight, chunk; };
// waves and chunk size of a stream launch over `count` rays on a chip with `wave_slots` persistent waves to fill
KJ_HD StreamTune stream_tune_for(uint32_t count, uint32_t wave_slots, uint32_t* out_waves) {
    uint32_t chunk = ((count / (wave_slots ? wave_slots : 1u) + 63u) / 64u) * 64u;
    chunk = chunk < 64u ? 64u : (chunk > KJ_STREAM_CHUNK ? KJ_STREAM_CHUNK : chunk);
    const uint32_t chunks = (count + chunk - 1u) / chunk;
    *out_waves = chunks < 1u ? 1u : (chunks < wave_slots ? chunks : wave_slots);
    return StreamTune{16u, 1u, 2u, chunk};
}
struct StreamHit { float t, u, v; uint32_t slot; };   // closest hit: slot = leaf-order triangle index, 0xffffffff = miss; occlusion: slot != 0xffffffff = blocked

// `emit(ray_index, hit)` stores one finished ray's result.
template <bool ANY_HIT, bool STATS = false, typename Emit>
KJ_D void bvh_trace_stream(const BvhView& bvh, const float4* __restrict__ rays, uint32_t count, bool cull_back,
                           uint32_t wave_index, uint32_t wave_count, uint32_t* stack, uint32_t stride, Emit emit, StreamTune tune = StreamTune{16u, 1u, 2u, KJ_STREAM_CHUNK},
                           TraverseStats* stats = nullptr) {
    const uint32_t lane = __lane_id() & 63u;
    const unsigned long long lane_bit = 1ull << lane;
    RayState S;
    S.cur = KJ_BVH_NONE; S.sp = 0;
    S.h.t = FLT_MAX; S.h.u = S.h.v = 0; S.h.slot = S.h.world_id = 0xffffffffu;
    uint32_t spill[KJ_BVH_SPILL_STACK];
    bool live = false;
    uint32_t ray_index = 0;
    uint32_t chunk = wave_index, cursor = wave_index * tune.chunk;
    uint32_t chunk_end = min(count, cursor + tune.chunk);
    bool exhausted = cursor >= count;
    for (;;) {
        const bool parked = S.cur == KJ_BVH_NONE;
        const unsigned long long pm = __ballot(parked);
        const uint32_t n_parked = uint32_t(__popcll(pm));
        if (!exhausted && (n_parked >= tune.refill_threshold || pm == ~0ull)) {
            if (parked && live) { emit(ray_index, S.h); live = false; }
            if (cursor == chunk_end) {
                chunk += wave_count; cursor = chunk * tune.chunk; chunk_end = min(count, cursor + tune.chunk);
                if (cursor >= count) { exhausted = true; chunk_end = cursor; }
            }
            const uint32_t take = min(n_parked, chunk_end - cursor);
            const uint32_t rank = uint32_t(__popcll(pm & (lane_bit - 1ull)));
            if (parked && rank < take) {
                ray_index = cursor + rank;
                const float4 a = rays[size_t(ray_index) * 2], b = rays[size_t(ray_index) * 2 + 1];
                ray_begin<ANY_HIT>(S, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, a.w, b.w, cull_back);
                if (!(b.w >= 0.0f)) S.cur = KJ_BVH_NONE;     // "no ray here"
                live = true;
            }
            cursor += take;
        }
        const bool want_node = wants_node_step(S), want_tri = wants_tri_step(S);
        const uint32_t nn = uint32_t(__popcll(__ballot(want_node))), nt = uint32_t(__popcll(__ballot(want_tri)));
        if (nn + nt == 0u) { if (exhausted) break; else continue; }
        if (nt == 0u || (nn != 0u && nn * tune.node_weight >= nt * tune.tri_weight)) { if (want_node) node_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats); }
        else { if (want_tri) tri_step<ANY_HIT, STATS>(bvh, S, stack, stride, spill, stats); }
    }
    if (live) emit(ray_index, S.h);
}
#undef KJ_PUSH
#undef KJ_POP
#endif // __HIPCC__

} // namespace kj
