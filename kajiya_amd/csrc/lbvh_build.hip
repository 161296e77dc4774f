// Acceleration-structure build ON THE DEVICE: a mesh's BLAS as a linear BVH -- the "prefer fast build" counterpart of the
// host SAH builder (bvh_build.cpp), same node format, same traversal.
//   1. per triangle: object-space box + centroid, mesh bounds by atomic min / max                        (k_lbvh_prims)
//   2. 63-bit Morton code (21 bits per axis) of the centroid inside the mesh bounds                                       (k_lbvh_morton)
//   3. radix sort of (code, triangle) pairs                                                             (rocPRIM via hipCUB)
//   4. binary radix tree over the sorted codes, one thread per internal node (Karras 2012; equal codes are told
//      apart by their position), then boxes bottom-up: the second thread to reach a node carries on         (k_lbvh_hierarchy, k_lbvh_refit)
//   5. collapse to 4-wide nodes level by level -- a subtree of <= 4 triangles becomes a leaf, otherwise the inner child with
//      the largest surface area is opened until there are four -- and quantise every child box to 8 bits per plane in its
//      parent's frame, rounding outwards against the kernel's own decode fma(q, step, origin)              (k_lbvh_collapse)
//   6. triangles in sorted (= leaf) order                                                                 (k_lbvh_emit_tris)
// Trees are shallower in quality than SAH ones (more node visits per ray); hits are identical: the triangles decide, and equal-t
// ties go to the lowest world triangle id whatever the tree.
//
// KJ_BLAS_BUILD_DEVICE_PLOC replaces step 4 by bottom-up agglomerative clustering over the Morton order (PLOC, Meister & Bittner 2018):
//   4'. every cluster (at first: every triangle) looks PLOC_RADIUS neighbours to either side of its place in the Morton order for the one
//       whose union with it has the smallest surface area; mutual nearest neighbours merge into a new node; the cluster list is
//       compacted in order (block scan + scan of the block totals); repeated until one cluster is left (~log n rounds, four launches
//       each, no read-back except every fourth round for the launch size). Each node carries its subtree's triangle count and SAH
//       cost; a subtree of <= KJ_BVH_MAX_LEAF_TRIS triangles becomes a leaf where that is the cheaper of the two (the host builder's rule). (k_ploc_*)
//   5'. the same 4-wide collapse; a subtree's triangles are no longer contiguous in the Morton order, so the collapse hands out
//       triangle slots top-down (first slot of a child = first slot of its parent + the counts of the children before it) and a leaf
//       writes its triangles' ids there.                                                                  (k_ploc_collapse)
// Trees built this way trace within a few percent of the host SAH trees (scripts/traversal_microbench.py) at a fraction of the build time.
#include "kj_host.hpp"
#include "kj_scene_device.hpp"
#include "kj_vec.hpp"
#include <cfloat>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

#include <hipcub/hipcub.hpp>

using namespace kj;

namespace {

struct Box6 { float mn[3], mx[3]; };
KJ_D uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }   // order-preserving
KJ_D float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
KJ_D float half_area(const Box6& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(64) k_lbvh_init(uint32_t* __restrict__ ob, uint32_t meshes) {   // ordered-uint bounds of every mesh of the batch (8 dwords each): min = +inf, max = -inf
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < meshes * 8u) ob[i] = (i & 7u) < 3u ? 0xffffffffu : 0u;
}
// (Round 6: the mesh bounds are reduced in LDS first and leave the workgroup as six atomics -- they were six global atomics per TRIANGLE on the same six words:
// 45 us per 110 k triangles, profiles/r06_blas_builds.md)
KJ_D void lbvh_bounds_reduce(bool valid, const Box6& b, uint32_t* __restrict__ ob) {
    __shared__ uint32_t ob_l[6];
    if (threadIdx.x < 6) ob_l[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    if (valid) for (int k = 0; k < 3; ++k) { atomicMin(&ob_l[k], f2o(b.mn[k])); atomicMax(&ob_l[3 + k], f2o(b.mx[k])); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&ob[threadIdx.x], ob_l[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&ob[threadIdx.x], ob_l[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_lbvh_prims(const uint8_t* __restrict__ vb, GpuMesh m, uint32_t n, Box6* __restrict__ pbox, uint32_t* __restrict__ ob, uint32_t* __restrict__ visits) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) visits[i] = 0u;      // the refit's arrival counters (was a fill of its own)
    Box6 b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = FLT_MAX; b.mx[k] = -FLT_MAX; }
    if (i < n) {
        for (int v = 0; v < 3; ++v) {
            const uint32_t idx = *(const uint32_t*)(vb + m.index_offset + (size_t(i) * 3 + v) * 4);
            const float* p = (const float*)(vb + m.vertex_core_offset + size_t(idx) * 16);
            for (int k = 0; k < 3; ++k) { b.mn[k] = fminf(b.mn[k], p[k]); b.mx[k] = fmaxf(b.mx[k], p[k]); }
        }
        pbox[i] = b;
    }
    lbvh_bounds_reduce(i < n, b, ob);
}
// the top tree's primitives: boxes given as they are (the padded world boxes of the instances' root or opened nodes)
__global__ void __launch_bounds__(256) k_lbvh_prims_boxes(const Box6* __restrict__ boxes, uint32_t n, Box6* __restrict__ pbox, uint32_t* __restrict__ ob, uint32_t* __restrict__ visits) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) visits[i] = 0u;
    Box6 b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = FLT_MAX; b.mx[k] = -FLT_MAX; }
    if (i < n) { b = boxes[i]; pbox[i] = b; }
    lbvh_bounds_reduce(i < n, b, ob);
}
// 21 bits per axis -> every third bit of a 63-bit code. (Round 2 used 10 bits per axis: in a 250 k-triangle mesh whole neighbourhoods
// share one 30-bit code, and triangles with equal codes are split by their POSITION in the sorted array, i.e. arbitrarily in space.)
typedef unsigned long long MortonCode;
#ifdef KJ_LBVH_MORTON30
#define KJ_MORTON_BITS 10
#else
#define KJ_MORTON_BITS 21
#endif
KJ_D MortonCode spread21(uint32_t v) {
    MortonCode x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull; x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void __launch_bounds__(256) k_lbvh_morton(const Box6* __restrict__ pbox, const uint32_t* __restrict__ ob, uint32_t n, MortonCode* __restrict__ codes, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // one scale for the three axes (the cube around the mesh): with a scale per axis a flat mesh -- a terrain -- would spend a third of
    // its code bits on height noise and neighbours in the code order would not be neighbours in space
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, o2f(ob[3 + k]) - o2f(ob[k]));
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = o2f(ob[k]);
        const float c = 0.5f * (pbox[i].mn[k] + pbox[i].mx[k]);
        const float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
        const float cells = float(1u << KJ_MORTON_BITS);
        q[k] = uint32_t(fminf(fmaxf(t * cells, 0.0f), cells - 1.0f));
    }
    codes[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    ids[i] = i;
}
// number of leading bits codes i and j share; equal codes are ordered by position
KJ_D int lcp(const MortonCode* __restrict__ codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const MortonCode a = codes[i], b = codes[j];
    return a != b ? __clzll((long long)(a ^ b)) : 64 + __clz(int(uint32_t(i) ^ uint32_t(j)));
}
// Binary radix tree. Node ids: internal i in [0, n-1), leaf k as (n - 1 + k). parent[] over all 2n-1 ids.
__global__ void __launch_bounds__(256) k_lbvh_hierarchy(const MortonCode* __restrict__ codes, int n, uint2* __restrict__ children, uint2* __restrict__ range, uint32_t* __restrict__ parent) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = lcp(codes, n, i, i + 1) - lcp(codes, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lcp(codes, n, i, i - d);
    int lmax = 2;
    while (lcp(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lcp(codes, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lcp(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lcp(codes, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t left = lo == gamma ? uint32_t(n - 1 + gamma) : uint32_t(gamma);
    const uint32_t right = hi == gamma + 1 ? uint32_t(n - 1 + gamma + 1) : uint32_t(gamma + 1);
    children[i] = make_uint2(left, right);
    range[i] = make_uint2(uint32_t(lo), uint32_t(hi));
    parent[left] = uint32_t(i);
    parent[right] = uint32_t(i);
    if (i == 0) parent[0] = 0xffffffffu;
}
// Node boxes, bottom-up: a thread per leaf climbs; at every node the first arrival stops and the second, which finds both children's boxes, goes on.
// Round 6: the climb is local to the workgroup as far as it can be. Internal node i of the radix tree has i as one end of its range of leaves, so the nodes whose
// range lies inside the workgroup's KJ_REFIT_BLOCK consecutive leaves are the workgroup's own: their links are staged in LDS, their arrival counters and boxes
// live there, and the climb through them is LDS traffic under workgroup-scope fences. Only the nodes that span workgroups -- ~log2(n / KJ_REFIT_BLOCK) levels, a
// few arrivals per workgroup -- go through the global counters with device-scope fences, with the links fetched alongside the arrival atomic and the climber's own
// box kept in registers (only the sibling's is read). Every node of a tree was a chain of global atomics and three device fences per step before:
// 232 us per 110 k triangles, a quarter of a device build (profiles/r06_blas_builds.md). Same boxes: min / max in the same (left, right) order.
#define KJ_REFIT_BLOCK 512u
__global__ void __launch_bounds__(KJ_REFIT_BLOCK) k_lbvh_refit(const Box6* __restrict__ pbox, const uint32_t* __restrict__ ids, int n_, const uint2* __restrict__ children, const uint2* __restrict__ range,
                                                                const uint32_t* __restrict__ parent, uint32_t* __restrict__ visits, Box6* __restrict__ nbox) {
    constexpr uint32_t B = KJ_REFIT_BLOCK, NONE = 0xffffffffu;
    __shared__ Box6 leaf_l[B], node_l[B];      // boxes of leaf base + t / of internal node base + t
    __shared__ uint2 child_l[B];
    __shared__ uint32_t parent_l[B], visits_l[B];
    __shared__ uint8_t own_l[B];               // internal node base + t has its whole range inside this workgroup
    const uint32_t n = uint32_t(n_), nint = n - 1u, t = threadIdx.x, base = blockIdx.x * B, k = base + t;
    Box6 mine;
    uint32_t me = nint + k, cur = NONE;
    if (k < n) { mine = pbox[ids[k]]; leaf_l[t] = mine; nbox[me] = mine; cur = parent[me]; }
    visits_l[t] = 0u;
    own_l[t] = 0;
    if (k < nint) {
        const uint2 r = range[k];
        child_l[t] = children[k]; parent_l[t] = parent[k];
        own_l[t] = uint8_t(r.x / B == blockIdx.x && r.y / B == blockIdx.x);
    }
    __syncthreads();
    if (k >= n) return;
    auto unite = [](const Box6& a, const Box6& b) { Box6 u; for (int q = 0; q < 3; ++q) { u.mn[q] = fminf(a.mn[q], b.mn[q]); u.mx[q] = fmaxf(a.mx[q], b.mx[q]); } return u; };
    // the workgroup's own nodes (cur - base wraps for a node below the block: not ours either)
    while (cur != NONE && cur - base < B && own_l[cur - base]) {
        const uint32_t l = cur - base;
        __threadfence_block();                                   // my box is in LDS before my arrival is
        if (atomicAdd(&visits_l[l], 1u) == 0u) return;
        __threadfence_block();
        const uint2 c = child_l[l];                              // both children lie inside the node's range: leaves / nodes of this workgroup
        const Box6 a = c.x >= nint ? leaf_l[c.x - nint - base] : node_l[c.x - base];
        const Box6 b = c.y >= nint ? leaf_l[c.y - nint - base] : node_l[c.y - base];
        mine = unite(a, b);
        node_l[l] = mine; nbox[cur] = mine;
        me = cur; cur = parent_l[l];
    }
    // nodes that span workgroups
    while (cur != NONE) {
        const uint2 c = children[cur];
        const uint32_t up = parent[cur];
        __threadfence();                                         // release: nbox[me]
        if (atomicAdd(&visits[cur], 1u) == 0u) return;
        __threadfence();                                         // acquire: the sibling's box was written before its climber's arrival
        const bool left = c.x == me;
        const Box6 sib = nbox[left ? c.y : c.x];
        mine = left ? unite(mine, sib) : unite(sib, mine);
        nbox[cur] = mine;
        me = cur; cur = up;
    }
}
// The frame of a 4-wide node (origin + power-of-two step per axis around its children's boxes) ...
KJ_D void node_frame(const Box6* cb, int nch, Bvh4Node& node, float scale[3]) {
    Box6 frame;
    for (int k = 0; k < 3; ++k) { frame.mn[k] = FLT_MAX; frame.mx[k] = -FLT_MAX; }
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < nch) for (int k = 0; k < 3; ++k) { frame.mn[k] = fminf(frame.mn[k], cb[i].mn[k]); frame.mx[k] = fmaxf(frame.mx[k], cb[i].mx[k]); }
    memset(&node, 0, sizeof(node));
    for (int k = 0; k < 3; ++k) {
        node.origin[k] = frame.mn[k];
        const float ext = frame.mx[k] - frame.mn[k];
        int e = -120;
        if (ext > 0.0f) { (void)frexpf(ext / 254.0f, &e); }     // smallest power of two with ext / 2^e <= 254
        e = min(127, max(-120, e));
        scale[k] = ldexpf(1.0f, e);
        node.exp8[k] = uint8_t(e + 127);
    }
}
// ... and child i's box in it: 8 bits per plane, rounded outwards against the traversal's own decode fma(q, step, origin)
KJ_D void quantise_child(Bvh4Node& node, const float scale[3], int i, const Box6& b) {
    for (int k = 0; k < 3; ++k) {
#pragma clang fp contract(off)
        const float inv = 1.0f / scale[k];
        int lo = int(floorf((b.mn[k] - node.origin[k]) * inv)), hi = int(ceilf((b.mx[k] - node.origin[k]) * inv));
        lo = min(255, max(0, lo)); hi = min(255, max(0, hi));
        while (lo > 0 && node.origin[k] + float(lo) * scale[k] > b.mn[k]) --lo;
        while (hi < 255 && node.origin[k] + float(hi) * scale[k] < b.mx[k]) ++hi;
        node.qlo[k][i] = uint8_t(lo); node.qhi[k][i] = uint8_t(hi);
    }
}
#ifndef KJ_LBVH_BATCH
#define KJ_LBVH_BATCH 14u      // collapse levels issued between two read-backs (tests build a variant with 3 to walk the continuation)
#endif
#define KJ_LBVH_RB_LEVELS (KJ_LBVH_BATCH + 2u)
#define KJ_LBVH_RB_MESH_HEAD (KJ_LBVH_RB_LEVELS + 4u + 8u)
#define KJ_LBVH_RB_MESH_DWORDS (KJ_LBVH_RB_MESH_HEAD + uint32_t(kj::LbvhResult::HEAD_NODES * sizeof(kj::Bvh4Node) / 4))
// One level of the 4-wide trees of ALL meshes of a batch. item = (binary internal node, output node index, depth, mesh).
// Round 6: the collapse is batched over the meshes of a commit -- a level is one launch for all of them. A level of a small mesh is a chain of ~8 dependent round
// trips whatever its item count (14-25 us), and nine meshes one after the other paid it nine times per level: half of a device build (profiles/r06_blas_builds.md).
struct CollapseMesh {
    uint32_t n, node_base, max_leaf, head_dwords;
    const uint2* children; const uint2* range; const uint32_t* cnt; const Box6* nbox; const uint32_t* sorted_ids; const uint32_t* leaf_refs;
    const uint32_t* root_cluster; const uint32_t* ob;
    Bvh4Node* nodes; uint32_t* tri_order; uint32_t* counters /*[1] = nodes, [2] = max depth*/;
    uint32_t* level_nodes;      // [1] = nodes before the batch's first level; [KJ_LBVH_RB_LEVELS + l] = nodes level l of the batch allocated (the levels of a tree are contiguous runs,
                                // which the per-instance refit walks deepest first; where each starts is summed up from these in k_collapse_pack_results -- it was a launch per level)
};
struct CollapseItem { uint32_t bin, out, depth, mesh; };
// The level's queue length lives on the device (queue_len[level]; the kernel appends to queue_len[level + 1]): the host launches every
// level with a grid that is large enough by construction (<= 4^level items per mesh, <= one per triangle) and reads nothing back in between.
__global__ void __launch_bounds__(64) k_lbvh_collapse(const CollapseMesh* __restrict__ meshes, const CollapseItem* __restrict__ in, uint32_t* __restrict__ queue_len, uint32_t level,
                                                       CollapseItem* __restrict__ out) {
    // leaf_refs (the top tree): every leaf holds ONE primitive and becomes the child reference leaf_refs[primitive] -- a node of an instance's tree -- as it is
    // Round 6: everything a binary node contributes -- links, range, box -- is requested in ONE round per opened node (`fetch`), and the choice of the node to
    // open works on registers. (The text of the selection is unchanged; as written before, is_leaf -> box -> links were three dependent round trips per candidate
    // and a level cost ~11 of them, ~21 us however few items it had: 14 levels x 9 meshes = a third of a device build, profiles/r06_blas_builds.md.)
    const uint32_t in_count = queue_len[level];
    struct Bin { uint2 c, r; Box6 b; bool leaf; };
    for (uint32_t w = blockIdx.x * 64 + threadIdx.x; w < in_count; w += gridDim.x * 64) {
    const CollapseItem it = in[w];
    const CollapseMesh& M = meshes[it.mesh];
    const int n = int(M.n);
    const uint2* __restrict__ children = M.children; const uint2* __restrict__ range = M.range; const Box6* __restrict__ nbox = M.nbox;
    const uint32_t* __restrict__ leaf_refs = M.leaf_refs; const uint32_t* __restrict__ sorted_ids = M.sorted_ids;
    uint32_t* const counters = M.counters; Bvh4Node* const nodes = M.nodes;
    const uint32_t max_leaf = M.max_leaf, node_base = M.node_base;
    auto fetch = [&](uint32_t id) {
        Bin f;
        const bool internal = id < uint32_t(n - 1);
        f.b = nbox[id];
        f.c = internal ? children[id] : make_uint2(0u, 0u);
        f.r = internal ? range[id] : make_uint2(0u, 0u);
        f.leaf = !internal || f.r.y - f.r.x + 1u <= max_leaf;
        return f;
    };
    uint32_t ch[4] = {0u, 0u, 0u, 0u}; Bin fi[4]; int nch = 0;
    if (n == 1) { ch[0] = 0u; fi[0].b = nbox[0]; fi[0].leaf = true; fi[0].c = fi[0].r = make_uint2(0u, 0u); nch = 1; }    // whole mesh fits one leaf
    else {
        const Bin root = fetch(it.bin);
        if (root.leaf) { ch[0] = it.bin; fi[0] = root; nch = 1; }
        else { ch[0] = root.c.x; ch[1] = root.c.y; fi[0] = fetch(ch[0]); fi[1] = fetch(ch[1]); nch = 2; }
        while (nch < 4) {
            int best = -1; float ba = -1.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nch && !fi[i].leaf) { const float a = half_area(fi[i].b); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            uint2 c = make_uint2(0u, 0u);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i == best) c = fi[i].c;
            const Bin fx = fetch(c.x), fy = fetch(c.y);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == best) { ch[i] = c.x; fi[i] = fx; }
                if (i == nch) { ch[i] = c.y; fi[i] = fy; }
            }
            ++nch;
        }
    }
    Box6 cb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cb[i] = fi[i < nch ? i : 0].b;
    Bvh4Node node;
    float scale[3];
    node_frame(cb, nch, node, scale);
    node.exp8[3] = uint8_t(nch);
    atomicMax(&counters[2], it.depth + uint32_t(nch));     // stack entries below this node + what its visit can push, + 1
    // node indices and queue places of the children that go on: ONE request each per item (they were a returning atomic per child, one after the other: up to eight
    // dependent round trips, half of what a level of few items took). Which index a node gets inside its level was never defined; now a node's children are neighbours.
    uint32_t inner = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) inner += (n != 1 && i < nch && !fi[i].leaf) ? 1u : 0u;
    uint32_t o = 0u, place = 0u;
    if (inner) { o = atomicAdd(&counters[1], inner); place = atomicAdd(&queue_len[level + 1], inner); atomicAdd(&M.level_nodes[KJ_LBVH_RB_LEVELS + level], inner); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= nch) {
            node.child[i] = 0xffffffffu;
            for (int k = 0; k < 3; ++k) { node.qlo[k][i] = 255; node.qhi[k][i] = 0; }
            continue;
        }
        quantise_child(node, scale, i, cb[i]);
        if (n == 1) node.child[i] = leaf_refs ? leaf_refs[0] : KJ_BVH_LEAF;
        else if (fi[i].leaf) {
            const uint32_t first = ch[i] >= uint32_t(n - 1) ? ch[i] - uint32_t(n - 1) : fi[i].r.x;
            const uint32_t cnt = ch[i] >= uint32_t(n - 1) ? 1u : fi[i].r.y - fi[i].r.x + 1u;
            node.child[i] = leaf_refs ? leaf_refs[sorted_ids[first]] : (KJ_BVH_LEAF | ((cnt - 1u) << 28) | first);
        } else {
            node.child[i] = node_base + o;
            out[place++] = CollapseItem{ch[i], o++, it.depth + uint32_t(nch - 1), it.mesh};
        }
    }
    nodes[it.out] = node;
    }
}
__global__ void __launch_bounds__(256) k_lbvh_emit_tris(const uint8_t* __restrict__ vb, GpuMesh m, const uint32_t* __restrict__ ids, uint32_t n, BvhTri* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = ids[i];
    BvhTri t;
    float* dst[3] = {t.v0, t.v1, t.v2};
    for (int v = 0; v < 3; ++v) {
        const uint32_t idx = *(const uint32_t*)(vb + m.index_offset + (size_t(p) * 3 + v) * 4);
        const float* s = (const float*)(vb + m.vertex_core_offset + size_t(idx) * 16);
        dst[v][0] = s[0]; dst[v][1] = s[1]; dst[v][2] = s[2];
    }
    t.world_id = 0; t.inst = 0; t.prim = p;
    out[i] = t;
}

// ------------------------------------------------------------------ PLOC: bottom-up clustering over the Morton order (header, step 4')
// Binary node ids here: leaf k (the k-th triangle in Morton order) = k, inner node j = n + j. Per node: box, `cnt` = triangles below it with
// bit 31 set when the subtree is emitted as ONE leaf, `cost` = SAH cost of the subtree (traversal step = 1, triangle test = 1: bvh_build.cpp).
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 16
#endif
#define PLOC_BLOCK 256
#define PLOC_LEAF_FLAG 0x80000000u
#define PLOC_NONE 0xffffffffu
KJ_D Box6 box_union(const Box6& a, const Box6& b) {
    Box6 u;
    for (int q = 0; q < 3; ++q) { u.mn[q] = fminf(a.mn[q], b.mn[q]); u.mx[q] = fmaxf(a.mx[q], b.mx[q]); }
    return u;
}
// exclusive scan of one value per thread of a PLOC_BLOCK-thread workgroup through LDS; returns the thread's prefix, *total = the block's sum
KJ_D uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /*[PLOC_BLOCK]*/, uint32_t* total) {
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < PLOC_BLOCK; off <<= 1) {
        const uint32_t add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const uint32_t incl = lds[t];
    *total = lds[PLOC_BLOCK - 1];
    __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(256) k_ploc_init(const Box6* __restrict__ pbox, const uint32_t* __restrict__ ids, uint32_t n, Box6* __restrict__ nbox, uint32_t* __restrict__ cnt, float* __restrict__ cost,
                                                    uint32_t* __restrict__ clusters, uint32_t* __restrict__ mcount) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) { mcount[0] = n; mcount[1] = 0u; /* inner nodes made so far */ }
    if (k >= n) return;
    nbox[k] = pbox[ids[k]];
    cnt[k] = 1u | PLOC_LEAF_FLAG;
    cost[k] = 1.0f;
    clusters[k] = k;
}
// nearest neighbour of every cluster inside the window. Pairs are compared by (area of the union, index distance, parity of the lower
// index, lower index): a strict order on unordered pairs, the same from both ends, so the smallest pair of a neighbourhood is always mutual
// -- and runs of identical boxes (coincident triangles) pair up (0,1)(2,3).. in one round instead of one merge per round.
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_nearest(const uint32_t* __restrict__ clusters, const Box6* __restrict__ nbox, const uint32_t* __restrict__ mcount, uint32_t* __restrict__ nearest) {
    __shared__ Box6 lb[PLOC_BLOCK + 2 * PLOC_RADIUS];
    const uint32_t m = mcount[0];
    const uint32_t base = blockIdx.x * PLOC_BLOCK;
    if (base >= m || m < 2u) return;
    for (uint32_t t = threadIdx.x; t < PLOC_BLOCK + 2 * PLOC_RADIUS; t += PLOC_BLOCK) {
        const long long g = (long long)base + t - PLOC_RADIUS;
        if (g >= 0 && g < (long long)m) lb[t] = nbox[clusters[g]];
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    if (i >= m) return;
    const Box6 mine = lb[threadIdx.x + PLOC_RADIUS];
    float best_d = FLT_MAX; uint32_t best_j = PLOC_NONE, best_dist = 0, best_par = 0, best_lo = 0;
    for (int o = -PLOC_RADIUS; o <= PLOC_RADIUS; ++o) {
        const long long g = (long long)i + o;
        if (o == 0 || g < 0 || g >= (long long)m) continue;
        const float d = half_area(box_union(mine, lb[int(threadIdx.x) + PLOC_RADIUS + o]));
        const uint32_t j = uint32_t(g), dist = uint32_t(o < 0 ? -o : o), lo = min(i, j), par = lo & 1u;
        const bool better = best_j == PLOC_NONE || d < best_d || (d == best_d && (dist < best_dist || (dist == best_dist && (par < best_par || (par == best_par && lo < best_lo)))));
        if (better) { best_d = d; best_j = j; best_dist = dist; best_par = par; best_lo = lo; }
    }
    nearest[i] = best_j;
}
// mutual nearest neighbours merge: the lower one makes the node and keeps the place, the upper one leaves the list
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_merge(const uint32_t* __restrict__ clusters, const uint32_t* __restrict__ nearest, uint32_t n, uint32_t* __restrict__ mcount, Box6* __restrict__ nbox,
                                                            uint2* __restrict__ children, uint32_t* __restrict__ cnt, float* __restrict__ cost, uint32_t* __restrict__ merged, uint32_t* __restrict__ block_valid) {
    __shared__ uint32_t lds[PLOC_BLOCK];
    __shared__ uint32_t node_base;
    const uint32_t m = mcount[0];
    const uint32_t base = blockIdx.x * PLOC_BLOCK;
    if (base >= m || m < 2u) return;
    const uint32_t i = base + threadIdx.x;
    uint32_t j = PLOC_NONE;
    bool makes = false, leaves = false;
    if (i < m) {
        j = nearest[i];
        const bool mutual = j != PLOC_NONE && nearest[j] == i;
        makes = mutual && i < j;
        leaves = mutual && i > j;
    }
    uint32_t total;
    const uint32_t my = block_exclusive_scan(makes ? 1u : 0u, lds, &total);
    if (threadIdx.x == 0) node_base = total ? atomicAdd(&mcount[1], total) : 0u;
    __syncthreads();
    uint32_t valid_total;
    (void)block_exclusive_scan((i < m && !leaves) ? 1u : 0u, lds, &valid_total);
    if (threadIdx.x == 0) block_valid[blockIdx.x] = valid_total;
    if (i >= m) return;
    uint32_t keep = clusters[i];
    if (makes) {
        const uint32_t a = keep, b = clusters[j], inner = node_base + my, id = n + inner;
        const Box6 ba = nbox[a], bb = nbox[b], u = box_union(ba, bb);
        nbox[id] = u;
        children[inner] = make_uint2(a, b);
        const uint32_t count = (cnt[a] & ~PLOC_LEAF_FLAG) + (cnt[b] & ~PLOC_LEAF_FLAG);
        const float as_inner = 1.0f + (half_area(ba) * cost[a] + half_area(bb) * cost[b]) / fmaxf(half_area(u), 1e-30f);
        const bool leaf = count <= KJ_BVH_MAX_LEAF_TRIS && float(count) <= as_inner;
        cnt[id] = count | (leaf ? PLOC_LEAF_FLAG : 0u);
        cost[id] = leaf ? float(count) : as_inner;
        keep = id;
    }
    merged[i] = leaves ? PLOC_NONE : keep;
}
// offsets of the blocks' survivors (one workgroup; `nb` block totals) and the new cluster count
__global__ void __launch_bounds__(1024) k_ploc_block_offsets(uint32_t* __restrict__ block_valid, uint32_t* __restrict__ mcount) {
    __shared__ uint32_t part[1024];
    const uint32_t m = mcount[0];
    if (m < 2u) return;
    const uint32_t nb = (m + PLOC_BLOCK - 1) / PLOC_BLOCK, per = (nb + 1023u) / 1024u;
    const uint32_t t = threadIdx.x, lo = min(nb, t * per), hi = min(nb, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += block_valid[b];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t add = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t b = lo; b < hi; ++b) { const uint32_t v = block_valid[b]; block_valid[b] = run; run += v; }
    if (t == 1023u) mcount[2] = part[1023];      // next round's count; k_ploc_compact moves it to mcount[0] when it is done with the old one
}
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_compact(const uint32_t* __restrict__ merged, const uint32_t* __restrict__ block_valid, uint32_t* __restrict__ mcount, uint32_t* __restrict__ clusters_out,
                                                              uint32_t* __restrict__ done_blocks) {
    __shared__ uint32_t lds[PLOC_BLOCK];
    const uint32_t m = mcount[0];
    const uint32_t base = blockIdx.x * PLOC_BLOCK;
    if (base >= m || m < 2u) return;
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < m ? merged[i] : PLOC_NONE;
    uint32_t total;
    const uint32_t my = block_exclusive_scan(v != PLOC_NONE ? 1u : 0u, lds, &total);
    if (v != PLOC_NONE) clusters_out[block_valid[blockIdx.x] + my] = v;
    // the last block to finish publishes the new count (every block of this launch has read the old one by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t nb = (m + PLOC_BLOCK - 1) / PLOC_BLOCK;
        if (atomicAdd(done_blocks, 1u) + 1u == nb) { *done_blocks = 0u; __threadfence(); mcount[0] = mcount[2]; }
    }
}

// The last rounds in ONE launch: once PLOC_TAIL clusters or fewer are left a single workgroup carries on by itself -- cluster list,
// boxes and nearest neighbours in LDS, a barrier where the multi-launch rounds have a kernel boundary. (The tail is where rounds are
// many and small: a 2 k-triangle mesh takes 36 rounds, a 250 k-triangle one ~50, the last ~25 of them on fewer than a thousand clusters.)
#define PLOC_TAIL 1024
KJ_D uint32_t tail_exclusive_scan(uint32_t v, uint32_t* lds /*[PLOC_TAIL]*/, uint32_t* total) {
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < PLOC_TAIL; off <<= 1) {
        const uint32_t add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const uint32_t incl = lds[t];
    *total = lds[PLOC_TAIL - 1];
    __syncthreads();
    return incl - v;
}
// (Round 6: ONE launch for the tails of all meshes of a batch, a workgroup each -- nine tails one after the other, each a single workgroup on an otherwise idle chip, were 2.2 ms of a city commit.)
struct PlocTailJob { uint32_t* clusters; uint32_t* mcount; Box6* nbox; uint2* children; uint32_t* cnt; float* cost; uint32_t n, pad; };
KJ_D void ploc_tail_body(uint32_t* __restrict__ clusters, uint32_t n, uint32_t* __restrict__ mcount, Box6* __restrict__ nbox, uint2* __restrict__ children,
                         uint32_t* __restrict__ cnt, float* __restrict__ cost) {
    __shared__ Box6 lb[PLOC_TAIL];
    __shared__ uint32_t lc[PLOC_TAIL], ln[PLOC_TAIL], scan[PLOC_TAIL];
    const uint32_t i = threadIdx.x;
    uint32_t m = mcount[0], made = mcount[1];
    if (m > PLOC_TAIL) return;
    if (i < m) { lc[i] = clusters[i]; lb[i] = nbox[lc[i]]; }
    __syncthreads();
    while (m > 1u) {
        uint32_t best_j = PLOC_NONE;
        if (i < m) {
            const Box6 mine = lb[i];
            float best_d = FLT_MAX; uint32_t best_dist = 0, best_par = 0, best_lo = 0;
            for (int o = -PLOC_RADIUS; o <= PLOC_RADIUS; ++o) {
                const int g = int(i) + o;
                if (o == 0 || g < 0 || g >= int(m)) continue;
                const float d = half_area(box_union(mine, lb[g]));
                const uint32_t j = uint32_t(g), dist = uint32_t(o < 0 ? -o : o), lo = min(i, j), par = lo & 1u;
                const bool better = best_j == PLOC_NONE || d < best_d || (d == best_d && (dist < best_dist || (dist == best_dist && (par < best_par || (par == best_par && lo < best_lo)))));
                if (better) { best_d = d; best_j = j; best_dist = dist; best_par = par; best_lo = lo; }
            }
            ln[i] = best_j;
        }
        __syncthreads();
        const bool mutual = i < m && best_j != PLOC_NONE && ln[best_j] == i;
        const bool makes = mutual && i < best_j, leaves = mutual && i > best_j;
        uint32_t total_made, total_valid;
#if defined(__HIP_DEVICE_COMPILE__)
        // both counts of a round from wave votes + sixteen wave totals: two barriers (round 6; the two LDS scans below were forty, half of k_ploc_tail's 314 us per mesh)
        uint32_t my_node, my_place;
        {
            const uint32_t wave = i >> 6, lane = i & 63u;
            const unsigned long long below = (1ull << lane) - 1ull, bm = __ballot(makes), bv = __ballot(i < m && !leaves);
            if (lane == 0u) scan[wave] = uint32_t(__popcll(bm)) | (uint32_t(__popcll(bv)) << 16);
            __syncthreads();
            uint32_t before = 0u, total = 0u;
            for (uint32_t w = 0; w < PLOC_TAIL / 64u; ++w) { const uint32_t v = scan[w]; total += v; before += w < wave ? v : 0u; }
            __syncthreads();
            my_node = uint32_t(__popcll(bm & below)) + (before & 0xffffu); my_place = uint32_t(__popcll(bv & below)) + (before >> 16);
            total_made = total & 0xffffu; total_valid = total >> 16;
        }
#else
        const uint32_t my_node = tail_exclusive_scan(makes ? 1u : 0u, scan, &total_made);
        const uint32_t my_place = tail_exclusive_scan((i < m && !leaves) ? 1u : 0u, scan, &total_valid);
#endif
        uint32_t keep = i < m ? lc[i] : PLOC_NONE;
        Box6 kb = i < m ? lb[i] : Box6{};
        if (makes) {
            const uint32_t a = keep, b = lc[best_j], inner = made + my_node, id = n + inner;
            const Box6 ba = kb, bb = lb[best_j], u = box_union(ba, bb);
            nbox[id] = u;
            children[inner] = make_uint2(a, b);
            const uint32_t count = (cnt[a] & ~PLOC_LEAF_FLAG) + (cnt[b] & ~PLOC_LEAF_FLAG);
            const float as_inner = 1.0f + (half_area(ba) * cost[a] + half_area(bb) * cost[b]) / fmaxf(half_area(u), 1e-30f);
            const bool leaf = count <= KJ_BVH_MAX_LEAF_TRIS && float(count) <= as_inner;
            cnt[id] = count | (leaf ? PLOC_LEAF_FLAG : 0u);
            cost[id] = leaf ? float(count) : as_inner;
            keep = id; kb = u;
        }
        __syncthreads();      // everyone has read its neighbours' lc / lb / cnt / cost of this round
        if (i < m && !leaves) { lc[my_place] = keep; lb[my_place] = kb; }
        made += total_made;
        m = total_valid;
        __threadfence();      // this workgroup's own cnt / cost writes are read back next round by other lanes
        __syncthreads();
    }
    if (i == 0) { clusters[0] = lc[0]; mcount[0] = 1u; mcount[1] = made; }
}
__global__ void __launch_bounds__(PLOC_TAIL) k_ploc_tail(const PlocTailJob* __restrict__ jobs) {
    const PlocTailJob J = jobs[blockIdx.x];
    ploc_tail_body(J.clusters, J.n, J.mcount, J.nbox, J.children, J.cnt, J.cost);
}
struct PlocItem { uint32_t bin, out, depth, first, mesh; };
// One level of the 4-wide trees over the PLOC hierarchies of a batch: as k_lbvh_collapse, plus the top-down hand-out of triangle slots.
__global__ void __launch_bounds__(64) k_ploc_collapse(const CollapseMesh* __restrict__ meshes, const PlocItem* __restrict__ in, uint32_t* __restrict__ queue_len, uint32_t level, PlocItem* __restrict__ out) {
    const uint32_t in_count = queue_len[level];
    // (round 6: as in k_lbvh_collapse, one round of requests per opened node)
    struct Bin { uint2 c; uint32_t cnt; Box6 b; bool leaf; };
    for (uint32_t w = blockIdx.x * 64 + threadIdx.x; w < in_count; w += gridDim.x * 64) {
        const PlocItem it = in[w];
        const CollapseMesh& M = meshes[it.mesh];
        const uint32_t n = M.n, node_base = M.node_base;
        const uint2* __restrict__ children = M.children; const uint32_t* __restrict__ cnt = M.cnt; const Box6* __restrict__ nbox = M.nbox; const uint32_t* __restrict__ sorted_ids = M.sorted_ids;
        uint32_t* const counters = M.counters; uint32_t* const tri_order = M.tri_order; Bvh4Node* const nodes = M.nodes;
        auto fetch = [&](uint32_t id) {
            Bin f;
            f.b = nbox[id];
            f.cnt = cnt[id];
            f.c = id >= n ? children[id - n] : make_uint2(0u, 0u);
            f.leaf = (f.cnt & PLOC_LEAF_FLAG) != 0u;
            return f;
        };
        uint32_t ch[4] = {0u, 0u, 0u, 0u}; Bin fi[4]; int nch = 0;
        const Bin root = fetch(it.bin);
        if (root.leaf) { ch[0] = it.bin; fi[0] = root; nch = 1; }          // the whole mesh fits one leaf
        else { ch[0] = root.c.x; ch[1] = root.c.y; fi[0] = fetch(ch[0]); fi[1] = fetch(ch[1]); nch = 2; }
        while (nch < 4) {
            int best = -1; float ba = -1.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < nch && !fi[i].leaf) { const float a = half_area(fi[i].b); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            uint2 c = make_uint2(0u, 0u);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i == best) c = fi[i].c;
            const Bin fx = fetch(c.x), fy = fetch(c.y);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == best) { ch[i] = c.x; fi[i] = fx; }
                if (i == nch) { ch[i] = c.y; fi[i] = fy; }
            }
            ++nch;
        }
        Box6 cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cb[i] = fi[i < nch ? i : 0].b;
        Bvh4Node node;
        float scale[3];
        node_frame(cb, nch, node, scale);
        node.exp8[3] = uint8_t(nch);
        atomicMax(&counters[2], it.depth + uint32_t(nch));
        uint32_t first = it.first;
        uint32_t inner = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) inner += (i < nch && !fi[i].leaf) ? 1u : 0u;
        uint32_t o = 0u, place = 0u;
        if (inner) { o = atomicAdd(&counters[1], inner); place = atomicAdd(&queue_len[level + 1], inner); atomicAdd(&M.level_nodes[KJ_LBVH_RB_LEVELS + level], inner); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nch) {
                node.child[i] = 0xffffffffu;
                for (int k = 0; k < 3; ++k) { node.qlo[k][i] = 255; node.qhi[k][i] = 0; }
                continue;
            }
            quantise_child(node, scale, i, cb[i]);
            const uint32_t count = fi[i].cnt & ~PLOC_LEAF_FLAG;
            if (fi[i].leaf) {
                node.child[i] = KJ_BVH_LEAF | ((count - 1u) << 28) | first;
                uint32_t stack[KJ_BVH_MAX_LEAF_TRIS + 1]; int sp = 0; uint32_t o = first;      // the leaf's triangles, left to right
                stack[sp++] = ch[i];
                while (sp) {
                    const uint32_t id = stack[--sp];
                    if (id < n) tri_order[o++] = sorted_ids[id];
                    else { const uint2 c = children[id - n]; stack[sp++] = c.y; stack[sp++] = c.x; }
                }
            } else {
                node.child[i] = node_base + o;
                out[place++] = PlocItem{ch[i], o++, it.depth + uint32_t(nch - 1), first, it.mesh};
            }
            first += count;
        }
        nodes[it.out] = node;
    }
}

#define KJ_NODE_CHILD_DWORD 4u
static_assert(offsetof(kj::Bvh4Node, child) == KJ_NODE_CHILD_DWORD * 4 && sizeof(kj::Bvh4Node) == 64, "k_blas_place_nodes walks a node as sixteen dwords");
// A tree moves to its place in the pool: dword by dword; the four child references of a node (dwords 4..7 of its 16: Bvh4Node::child) that name inner nodes get the base
__global__ void __launch_bounds__(256) k_blas_place_nodes(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t dwords, uint32_t node_base) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dwords) return;
    uint32_t v = src[i];
    const uint32_t w = i & 15u;
    if (w >= KJ_NODE_CHILD_DWORD && w < KJ_NODE_CHILD_DWORD + 4u && v != 0xffffffffu && !(v & KJ_BVH_LEAF)) v += node_base;
    dst[i] = v;
}
// What the host reads after a batch of levels, gathered into one block on the device and fetched with ONE copy (five copies into pageable memory per mesh were
// ~100 us of idle GPU each time): [queue lengths] then per mesh [levels | counters[4] | bounds[8] | the tree's first nodes].
__global__ void __launch_bounds__(256) k_collapse_pack_results(const CollapseMesh* __restrict__ meshes, const uint32_t* __restrict__ queue_len, uint32_t* __restrict__ out) {
    const uint32_t t = threadIdx.x, L = KJ_LBVH_RB_LEVELS;
    const CollapseMesh& M = meshes[blockIdx.x];
    if (blockIdx.x == 0 && t < L) out[t] = queue_len[t];
    uint32_t* const o = out + L + size_t(blockIdx.x) * KJ_LBVH_RB_MESH_DWORDS;
    if (t == 0) {      // nodes before every level of the batch
        uint32_t acc = M.level_nodes[1];
        o[0] = 0u; o[1] = acc;
        for (uint32_t l = 0; l < KJ_LBVH_BATCH; ++l) { acc += M.level_nodes[L + l]; o[l + 2] = acc; }
    }
    if (t < 4u) o[L + t] = M.counters[t];
    if (t < 8u) o[L + 4u + t] = t < 6u ? M.ob[t] : 0u;
    const uint32_t* __restrict__ head = (const uint32_t*)M.nodes;
    for (uint32_t i = t; i < M.head_dwords; i += 256u) o[KJ_LBVH_RB_MESH_HEAD + i] = head[i];
}
// Start of a batch of levels. first: every mesh's root item, counters = {-, 1 node (the root), 0}; otherwise the items left by the previous batch go on.
template <typename Item>
__global__ void __launch_bounds__(64) k_collapse_begin(const CollapseMesh* __restrict__ meshes, uint32_t nmesh, uint32_t* __restrict__ queue_len, Item* __restrict__ q0, int first, int ploc) {
    const uint32_t m = blockIdx.x * 64 + threadIdx.x;
    if (m < nmesh) {
        const CollapseMesh& M = meshes[m];
        if (first) { M.counters[0] = 0u; M.counters[1] = 1u; M.counters[2] = 0u; M.counters[3] = 0u; }
        for (uint32_t l = 0; l < 2u * KJ_LBVH_RB_LEVELS; ++l) M.level_nodes[l] = 0u;
        M.level_nodes[1] = first ? 1u : M.counters[1];
        if (first) { Item it{}; it.bin = ploc ? M.root_cluster[0] : 0u; it.mesh = m; q0[m] = it; }
    }
    if (m == 0) {
        const uint32_t carry = first ? nmesh : queue_len[KJ_LBVH_BATCH];
        for (uint32_t l = 0; l < KJ_LBVH_RB_LEVELS; ++l) queue_len[l] = 0u;
        queue_len[0] = carry;
    }
}

}  // namespace

namespace kj {

#define KJ_LB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
// One builder, two kinds of primitives: a mesh's triangles (boxes == nullptr) or given boxes with a child reference each (the top tree: leaves of one).
// A call builds a BATCH of meshes: the per-mesh stages (boxes, codes, sort, hierarchy / clustering, node boxes) one mesh after the other, the collapse into 4-wide
// nodes level by level for all of them together, one read-back at the end.
// (The per-mesh stages of a batch on up to four side streams between two events on the caller's stream: measured, city 2.6 -> 2.2 ms, ruins 8.7 -> 8.1 ms per commit -- the
// stages are bound by the host's launch rate, ~17 launches per mesh -- against 16 ms once for creating the streams, and a commit with several new meshes is a scene load:
// not kept. profiles/r06_blas_builds.md)
struct LbvhJob { const uint8_t* vb; GpuMesh mesh; const Box6* boxes; const uint32_t* leaf_refs; uint32_t n, node_base; Bvh4Node* nodes_out; BvhTri* tris_out; LbvhResult* result; };
static hipError_t build_lbvh_batch(const LbvhJob* jobs, uint32_t njobs, LbvhScratch* scratch, hipStream_t s, bool ploc) {
    if (!njobs || !scratch) return hipErrorInvalidValue;
    for (uint32_t j = 0; j < njobs; ++j) if (jobs[j].n == 0) return hipErrorInvalidValue;
    // working set: ONE allocation (grown when a commit needs more), carved into the shared pieces and every job's own buffers
    enum { PBOX, CODES, IDS, CODES2, IDS2, CHILDREN, RANGE, PARENT, VISITS, NBOX, COUNTERS, LEVELS, JOB_SLOTS };
    enum { Q0, Q1, QLEN, TABLE, READBACK, SORT_TMP, BOUNDS, TAILS, SHARED_SLOTS };
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t total_n = 0, sort_bytes = 16;
    for (uint32_t j = 0; j < njobs; ++j) {
        total_n += jobs[j].n;
        size_t b = 0;
        KJ_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const MortonCode*)nullptr, (MortonCode*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, int(jobs[j].n), 0, 3 * KJ_MORTON_BITS, s));
        sort_bytes = std::max(sort_bytes, b);
    }
    const size_t rb_dwords = KJ_LBVH_RB_LEVELS + size_t(njobs) * KJ_LBVH_RB_MESH_DWORDS;
    const size_t shared_bytes[SHARED_SLOTS] = {(total_n + njobs) * sizeof(PlocItem), (total_n + njobs) * sizeof(PlocItem), KJ_LBVH_RB_LEVELS * 4, njobs * sizeof(CollapseMesh), rb_dwords * 4, sort_bytes, size_t(njobs) * 32, njobs * sizeof(PlocTailJob)};
    auto job_bytes = [&](uint32_t n, int k) -> size_t {
        const size_t c = n;
        switch (k) {
            case PBOX: return c * sizeof(Box6); case CODES: return c * 8; case IDS: return c * 4; case CODES2: return c * 8; case IDS2: return c * 4;
            case CHILDREN: return c * 8; case RANGE: return c * 8; case PARENT: return 2 * c * 4; case VISITS: return c * 4; case NBOX: return 2 * c * sizeof(Box6);
            case COUNTERS: return 64; default: return 2 * KJ_LBVH_RB_LEVELS * 4;
        }
    };
    size_t total = 0;
    for (int k = 0; k < SHARED_SLOTS; ++k) total += up(shared_bytes[k]);
    for (uint32_t j = 0; j < njobs; ++j) for (int k = 0; k < JOB_SLOTS; ++k) total += up(job_bytes(jobs[j].n, k));
    if (scratch->arena.bytes < total) KJ_LB(scratch->arena.alloc(total + total / 4, s));
    uint8_t* at = (uint8_t*)scratch->arena.p;
    void* shared[SHARED_SLOTS];
    for (int k = 0; k < SHARED_SLOTS; ++k) { shared[k] = at; at += up(shared_bytes[k]); }
    std::vector<void*> slots(size_t(njobs) * JOB_SLOTS);
    for (uint32_t j = 0; j < njobs; ++j) for (int k = 0; k < JOB_SLOTS; ++k) { slots[size_t(j) * JOB_SLOTS + k] = at; at += up(job_bytes(jobs[j].n, k)); }
    uint32_t* const queue_len = (uint32_t*)shared[QLEN];
    CollapseMesh* const d_table = (CollapseMesh*)shared[TABLE];
    uint32_t* const readback = (uint32_t*)shared[READBACK];
    std::vector<CollapseMesh> table(njobs);
    std::vector<PlocTailJob> tails(ploc ? njobs : 0);

    // ---- per mesh: primitive boxes, Morton codes, sort, the binary hierarchy with its node boxes
    for (uint32_t j = 0; j < njobs; ++j) {
        const LbvhJob& job = jobs[j];
        const uint32_t n = job.n;
        void* const sort_tmp = shared[SORT_TMP];
        const bool top = job.boxes != nullptr;
        void* const* const slot = &slots[size_t(j) * JOB_SLOTS];
        // typed views of the working set (a buffer serves several stages: what PLOC clusters in is what the sort is done with)
        Box6* const pbox = (Box6*)slot[PBOX];
        uint32_t* const ob = (uint32_t*)shared[BOUNDS] + size_t(j) * 8;
        MortonCode* const codes = (MortonCode*)slot[CODES];
        uint32_t* const ids = (uint32_t*)slot[IDS];
        MortonCode* const codes2 = (MortonCode*)slot[CODES2];
        uint32_t* const ids2 = (uint32_t*)slot[IDS2];            // triangle ids in Morton order
        uint2* const children = (uint2*)slot[CHILDREN];
        uint2* const range = (uint2*)slot[RANGE];
        uint32_t* const parent = (uint32_t*)slot[PARENT];
        uint32_t* const visits = (uint32_t*)slot[VISITS];
        Box6* const nbox = (Box6*)slot[NBOX];
        uint32_t* const counters = (uint32_t*)slot[COUNTERS];
        uint32_t* const clusters = (uint32_t*)codes;              // PLOC: the cluster list, ...
        uint32_t* const merged = (uint32_t*)codes + n;            // ... its next state before compaction,
        uint32_t* const nearest = ids;                            // every cluster's nearest neighbour,
        uint32_t* const cnt = parent;                             // per node: triangles below (bit 31: emitted as one leaf)
        float* const cost = (float*)range;                        // per node: SAH cost of the subtree
        uint32_t* const block_valid = visits;                     // survivors per block, then their offsets
        uint32_t* const mcount = counters + 4;                    // {clusters, inner nodes made, next round's clusters, blocks done}
        uint32_t* const tri_order = (uint32_t*)codes2;            // triangle ids in leaf order (the collapse writes it)
        const dim3 g((n + 255) / 256), b(256);
        if (j == 0) hipLaunchKernelGGL(k_lbvh_init, dim3((njobs * 8 + 63) / 64), dim3(64), 0, s, (uint32_t*)shared[BOUNDS], njobs);
        if (top) hipLaunchKernelGGL(k_lbvh_prims_boxes, g, b, 0, s, job.boxes, n, pbox, ob, visits);
        else hipLaunchKernelGGL(k_lbvh_prims, g, b, 0, s, job.vb, job.mesh, n, pbox, ob, visits);
        hipLaunchKernelGGL(k_lbvh_morton, g, b, 0, s, (const Box6*)pbox, (const uint32_t*)ob, n, codes, ids);
        size_t tmp_bytes = 0;
        KJ_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const MortonCode*)codes, codes2, (const uint32_t*)ids, ids2, int(n), 0, 3 * KJ_MORTON_BITS, s));
        if (tmp_bytes > sort_bytes) return hipErrorUnknown;
        KJ_LB(hipcub::DeviceRadixSort::SortPairs(sort_tmp, tmp_bytes, (const MortonCode*)codes, codes2, (const uint32_t*)ids, ids2, int(n), 0, 3 * KJ_MORTON_BITS, s));
        if (ploc) {
            KJ_LB(hipMemsetAsync(counters, 0, 64, s));
            hipLaunchKernelGGL(k_ploc_init, g, b, 0, s, (const Box6*)pbox, (const uint32_t*)ids2, n, nbox, cnt, cost, clusters, mcount);
            uint32_t m = n, rounds = 0;
            while (m > PLOC_TAIL) {
                const dim3 gr((m + PLOC_BLOCK - 1) / PLOC_BLOCK);
                for (int k = 0; k < 4; ++k) {      // four rounds on the last known count (it only shrinks; blocks past the end leave at once)
                    hipLaunchKernelGGL(k_ploc_nearest, gr, dim3(PLOC_BLOCK), 0, s, (const uint32_t*)clusters, (const Box6*)nbox, (const uint32_t*)mcount, nearest);
                    hipLaunchKernelGGL(k_ploc_merge, gr, dim3(PLOC_BLOCK), 0, s, (const uint32_t*)clusters, (const uint32_t*)nearest, n, mcount, nbox, children, cnt, cost, merged, block_valid);
                    hipLaunchKernelGGL(k_ploc_block_offsets, dim3(1), dim3(1024), 0, s, block_valid, mcount);
                    hipLaunchKernelGGL(k_ploc_compact, gr, dim3(PLOC_BLOCK), 0, s, (const uint32_t*)merged, (const uint32_t*)block_valid, mcount, clusters, mcount + 3);
                }
                KJ_LB(hipMemcpyAsync(&m, mcount, 4, hipMemcpyDeviceToHost, s));
                KJ_LB(hipStreamSynchronize(s));
                if (++rounds > n) return hipErrorUnknown;      // every round merges at least one pair
                if (getenv("KJ_BVH_TIMING")) fprintf(stderr, "[ploc] after %u rounds: %u clusters\n", rounds * 4u, m);
            }
            tails[j] = PlocTailJob{clusters, mcount, nbox, children, cnt, cost, n, 0u};      // the last <= PLOC_TAIL clusters: all meshes' tails in one launch below
            KJ_LB(hipMemsetAsync(tri_order, 0, size_t(n) * 4, s));      // slots a deep tree has not reached after the first batch must still name a triangle (emit below)
        } else {
            if (n > 1) hipLaunchKernelGGL(k_lbvh_hierarchy, g, b, 0, s, (const MortonCode*)codes2, int(n), children, range, parent);
            else KJ_LB(hipMemsetAsync(parent, 0xff, 8, s));
            hipLaunchKernelGGL(k_lbvh_refit, dim3((n + KJ_REFIT_BLOCK - 1) / KJ_REFIT_BLOCK), dim3(KJ_REFIT_BLOCK), 0, s, (const Box6*)pbox, (const uint32_t*)ids2, int(n), (const uint2*)children, (const uint2*)range,
                               (const uint32_t*)parent, visits, nbox);
            if (!top) hipLaunchKernelGGL(k_lbvh_emit_tris, g, b, 0, s, job.vb, job.mesh, (const uint32_t*)ids2, n, job.tris_out);      // the triangles in leaf order = in Morton order
        }
        job.result->level_starts.assign({0u});
        job.result->head.resize(top ? 0 : std::min<size_t>(size_t(n) + 1, LbvhResult::HEAD_NODES));
        CollapseMesh& M = table[j];
        M.n = n; M.node_base = job.node_base; M.max_leaf = top ? 1u : uint32_t(KJ_BVH_MAX_LEAF_TRIS); M.head_dwords = uint32_t(job.result->head.size() * (sizeof(Bvh4Node) / 4));
        M.children = children; M.range = range; M.cnt = cnt; M.nbox = nbox; M.sorted_ids = ids2; M.leaf_refs = job.leaf_refs; M.root_cluster = clusters; M.ob = ob;
        M.nodes = job.nodes_out; M.tri_order = tri_order; M.counters = counters; M.level_nodes = (uint32_t*)slot[LEVELS];
    }
    // ---- the collapse, level by level, all meshes together: counters = {-, nodes allocated, max stack} per mesh; queue_len[l] = items of level l (the kernel of level l
    // appends to queue_len[l + 1]); level_nodes[l] = a mesh's nodes allocated before level l's children (a level's nodes are one contiguous run). Levels are issued
    // KJ_LBVH_BATCH at a time without looking at the queues -- every launch is sized by the bound 4^level per mesh, <= one item per triangle --, then ONE read-back
    // says whether a tree goes deeper (a 250 k-triangle mesh has ~13 levels; many coincident centroids make deep ones).
    if (ploc) {
        KJ_LB(hipMemcpyAsync(shared[TAILS], tails.data(), njobs * sizeof(PlocTailJob), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_ploc_tail, dim3(njobs), dim3(PLOC_TAIL), 0, s, (const PlocTailJob*)shared[TAILS]);
    }
    KJ_LB(hipMemcpyAsync(d_table, table.data(), njobs * sizeof(CollapseMesh), hipMemcpyHostToDevice, s));
    void* qin = shared[Q0]; void* qout = shared[Q1];
    const dim3 mg((njobs + 63) / 64);
    if (ploc) hipLaunchKernelGGL(k_collapse_begin<PlocItem>, mg, dim3(64), 0, s, (const CollapseMesh*)d_table, njobs, queue_len, (PlocItem*)qin, 1, 1);
    else hipLaunchKernelGGL(k_collapse_begin<CollapseItem>, mg, dim3(64), 0, s, (const CollapseMesh*)d_table, njobs, queue_len, (CollapseItem*)qin, 1, 0);
    scratch->readback.resize(rb_dwords);
    uint64_t per_mesh_bound = 1, carried = 0;      // items of the next level: <= per_mesh_bound per mesh (first batch), <= carried (later batches), <= a mesh's triangles
    uint32_t in_count = njobs;
    size_t batches = 0;
    while (in_count) {
        for (uint32_t level = 0; level < KJ_LBVH_BATCH; ++level) {
            uint64_t items = 0;
            if (carried) items = std::min<uint64_t>(carried, total_n);
            else for (uint32_t j = 0; j < njobs; ++j) items += std::min<uint64_t>(per_mesh_bound, jobs[j].n);
            const dim3 cg(uint32_t(std::min<uint64_t>(4096u, (items + 63) / 64)));
            if (ploc) hipLaunchKernelGGL(k_ploc_collapse, cg, dim3(64), 0, s, (const CollapseMesh*)d_table, (const PlocItem*)qin, queue_len, level, (PlocItem*)qout);
            else hipLaunchKernelGGL(k_lbvh_collapse, cg, dim3(64), 0, s, (const CollapseMesh*)d_table, (const CollapseItem*)qin, queue_len, level, (CollapseItem*)qout);
            if (carried) carried = std::min<uint64_t>(carried * 4, total_n); else per_mesh_bound = std::min<uint64_t>(per_mesh_bound * 4, uint64_t(1) << 40);
            std::swap(qin, qout);
        }
        // PLOC's triangle order is written by the collapse; if a tree turns out deeper, the next batch emits its triangles again
        if (ploc) for (uint32_t j = 0; j < njobs; ++j)
            if (!jobs[j].boxes) hipLaunchKernelGGL(k_lbvh_emit_tris, dim3((jobs[j].n + 255) / 256), dim3(256), 0, s, jobs[j].vb, jobs[j].mesh, (const uint32_t*)table[j].tri_order, jobs[j].n, jobs[j].tris_out);
        hipLaunchKernelGGL(k_collapse_pack_results, dim3(njobs), dim3(256), 0, s, (const CollapseMesh*)d_table, (const uint32_t*)queue_len, readback);
        KJ_LB(hipMemcpyAsync(scratch->readback.data(), readback, rb_dwords * 4, hipMemcpyDeviceToHost, s));
        KJ_LB(hipStreamSynchronize(s));
        const uint32_t* rb = scratch->readback.data();
        in_count = rb[KJ_LBVH_BATCH];
        for (uint32_t j = 0; j < njobs; ++j) {
            LbvhResult* r = jobs[j].result;
            const uint32_t* o = rb + KJ_LBVH_RB_LEVELS + size_t(j) * KJ_LBVH_RB_MESH_DWORDS;
            for (uint32_t l = 1; l <= KJ_LBVH_BATCH + 1; ++l)
                if (o[l] > r->level_starts.back()) r->level_starts.push_back(o[l]);
            if (r->level_starts.size() > 4096) return hipErrorUnknown;
            if (in_count) continue;
            const uint32_t* host_counters = o + KJ_LBVH_RB_LEVELS; const uint32_t* hob = host_counters + 4;
            for (int k = 0; k < 6; ++k) {
                const uint32_t ov = hob[k];
                const uint32_t u = (ov & 0x80000000u) ? (ov & 0x7fffffffu) : ~ov;
                memcpy(&r->bounds[k], &u, 4);
            }
            r->head.resize(std::min<size_t>(r->head.size(), host_counters[1]));
            if (!r->head.empty()) memcpy(r->head.data(), o + KJ_LBVH_RB_MESH_HEAD, r->head.size() * sizeof(Bvh4Node));
            r->node_count = host_counters[1];
            r->max_stack = host_counters[2] > 0 ? host_counters[2] : 1;
        }
        if (in_count) {
            carried = in_count;
            if (ploc) hipLaunchKernelGGL(k_collapse_begin<PlocItem>, mg, dim3(64), 0, s, (const CollapseMesh*)d_table, njobs, queue_len, (PlocItem*)qin, 0, 1);
            else hipLaunchKernelGGL(k_collapse_begin<CollapseItem>, mg, dim3(64), 0, s, (const CollapseMesh*)d_table, njobs, queue_len, (CollapseItem*)qin, 0, 0);
        }
        if (++batches > 4096) return hipErrorUnknown;
    }
    KJ_LB(hipGetLastError());
    return hipSuccess;
}
hipError_t build_blas_lbvh_device(const uint8_t* d_vertex_buffer, const GpuMesh& mesh, uint32_t node_base, Bvh4Node* d_nodes_out, BvhTri* d_tris_out, LbvhResult* result, LbvhScratch* scratch, hipStream_t s, bool ploc) {
    const LbvhJob job{d_vertex_buffer, mesh, nullptr, nullptr, mesh.index_count / 3, node_base, d_nodes_out, d_tris_out, result};
    return build_lbvh_batch(&job, 1, scratch, s, ploc);
}
// The meshes of a commit in one go. Nodes of mesh j into batch[j].nodes_out[0 .. node_count) -- at most one per triangle + 1 --, child node indices offset by batch[j].node_base.
hipError_t build_blas_lbvh_device_batch(const uint8_t* d_vertex_buffer, const LbvhBatchMesh* batch, uint32_t count, LbvhScratch* scratch, hipStream_t s, bool ploc) {
    std::vector<LbvhJob> jobs(count);
    for (uint32_t j = 0; j < count; ++j) jobs[j] = LbvhJob{d_vertex_buffer, batch[j].mesh, nullptr, nullptr, batch[j].mesh.index_count / 3, batch[j].node_base, batch[j].nodes_out, batch[j].tris_out, batch[j].result};
    return build_lbvh_batch(jobs.data(), count, scratch, s, ploc);
}
hipError_t launch_blas_place_nodes(const Bvh4Node* src, Bvh4Node* dst, uint32_t count, uint32_t node_base, hipStream_t s) {
    if (count) hipLaunchKernelGGL(k_blas_place_nodes, dim3((count * 16u + 255u) / 256u), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, count * 16u, node_base);
    return hipGetLastError();
}
// The per-commit top tree as a linear BVH over the leaves' world boxes (kj_scene_device.hpp): d_leaf_boxes[i] = {min xyz, max xyz}, d_leaf_refs[i] = the world
// node that leaf stands for. Nodes into d_nodes_out[0 .. node_count), node_count < max(leaf_count, 2). One synchronisation.
hipError_t build_top_lbvh_device(const float* d_leaf_boxes, const uint32_t* d_leaf_refs, uint32_t leaf_count, Bvh4Node* d_nodes_out, LbvhResult* result, LbvhScratch* scratch, hipStream_t s) {
    static_assert(sizeof(Box6) == 24, "a box is six floats");
    if (!d_leaf_boxes || !d_leaf_refs) return hipErrorInvalidValue;
    const LbvhJob job{nullptr, GpuMesh{}, (const Box6*)d_leaf_boxes, d_leaf_refs, leaf_count, 0u, d_nodes_out, nullptr, result};
    return build_lbvh_batch(&job, 1, scratch, s, false);
}

}  // namespace kj
