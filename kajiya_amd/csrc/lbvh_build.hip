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
#include "kj_host.hpp"
#include "kj_scene_device.hpp"
#include "kj_vec.hpp"
#include <cfloat>

#ifndef KJ_HIP_EMU_HOST
#include <hipcub/hipcub.hpp>
#endif

using namespace kj;

#ifndef KJ_HIP_EMU_HOST
namespace {

struct Box6 { float mn[3], mx[3]; };
KJ_D uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }   // order-preserving
KJ_D float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
KJ_D float half_area(const Box6& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_lbvh_init(uint32_t* __restrict__ ob) {   // ordered-uint bounds: min = +inf, max = -inf
    if (threadIdx.x < 3) ob[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) ob[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(256) k_lbvh_prims(const uint8_t* __restrict__ vb, GpuMesh m, uint32_t n, Box6* __restrict__ pbox, uint32_t* __restrict__ ob) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Box6 b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = FLT_MAX; b.mx[k] = -FLT_MAX; }
    for (int v = 0; v < 3; ++v) {
        const uint32_t idx = *(const uint32_t*)(vb + m.index_offset + (size_t(i) * 3 + v) * 4);
        const float* p = (const float*)(vb + m.vertex_core_offset + size_t(idx) * 16);
        for (int k = 0; k < 3; ++k) { b.mn[k] = fminf(b.mn[k], p[k]); b.mx[k] = fmaxf(b.mx[k], p[k]); }
    }
    pbox[i] = b;
    for (int k = 0; k < 3; ++k) { atomicMin(&ob[k], f2o(b.mn[k])); atomicMax(&ob[3 + k], f2o(b.mx[k])); }
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
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = o2f(ob[k]), hi = o2f(ob[3 + k]);
        const float c = 0.5f * (pbox[i].mn[k] + pbox[i].mx[k]);
        const float t = hi > lo ? (c - lo) / (hi - lo) : 0.0f;
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
__global__ void __launch_bounds__(256) k_lbvh_refit(const Box6* __restrict__ pbox, const uint32_t* __restrict__ ids, int n, const uint2* __restrict__ children,
                                                     const uint32_t* __restrict__ parent, uint32_t* __restrict__ visits, Box6* __restrict__ nbox) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    nbox[n - 1 + k] = pbox[ids[k]];
    __threadfence();
    uint32_t cur = parent[n - 1 + k];
    while (cur != 0xffffffffu) {
        if (atomicAdd(&visits[cur], 1u) == 0u) return;     // the first arrival stops; the second has both children's boxes
        __threadfence();
        const uint2 c = children[cur];
        const Box6 a = nbox[c.x], b = nbox[c.y];
        Box6 u;
        for (int q = 0; q < 3; ++q) { u.mn[q] = fminf(a.mn[q], b.mn[q]); u.mx[q] = fmaxf(a.mx[q], b.mx[q]); }
        nbox[cur] = u;
        __threadfence();
        cur = parent[cur];
    }
}
// One level of the 4-wide tree. item = (binary internal node, output node index, depth).
struct CollapseItem { uint32_t bin, out, depth; };
// The level's queue length lives on the device (queue_len[level]; the kernel appends to queue_len[level + 1]): the host launches every
// level with a grid that is large enough by construction (<= 4^level items, <= one per triangle) and reads nothing back in between.
__global__ void __launch_bounds__(64) k_lbvh_collapse(int n, const uint2* __restrict__ children, const uint2* __restrict__ range, const Box6* __restrict__ nbox, const CollapseItem* __restrict__ in,
                                                       uint32_t* __restrict__ queue_len, uint32_t level, CollapseItem* __restrict__ out, uint32_t* __restrict__ counters /*[1]=nodes, [2]=max depth*/,
                                                       Bvh4Node* __restrict__ nodes, uint32_t node_base) {
    const uint32_t in_count = queue_len[level];
    for (uint32_t w = blockIdx.x * 64 + threadIdx.x; w < in_count; w += gridDim.x * 64) {
    const CollapseItem it = in[w];
    auto is_leaf = [&](uint32_t id) { return id >= uint32_t(n - 1) || range[id].y - range[id].x + 1u <= KJ_BVH_MAX_LEAF_TRIS; };
    uint32_t ch[4]; int nch = 0;
    if (n == 1 || is_leaf(it.bin)) ch[nch++] = n == 1 ? 0u : it.bin;    // whole mesh fits one leaf
    else { const uint2 c = children[it.bin]; ch[nch++] = c.x; ch[nch++] = c.y; }
    while (nch < 4) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < nch; ++i)
            if (!is_leaf(ch[i])) { const float a = half_area(nbox[ch[i]]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        const uint2 c = children[ch[best]];
        ch[best] = c.x; ch[nch++] = c.y;
    }
    Box6 frame;
    for (int k = 0; k < 3; ++k) { frame.mn[k] = FLT_MAX; frame.mx[k] = -FLT_MAX; }
    for (int i = 0; i < nch; ++i) { const Box6 b = nbox[n == 1 ? 0 : ch[i]]; for (int k = 0; k < 3; ++k) { frame.mn[k] = fminf(frame.mn[k], b.mn[k]); frame.mx[k] = fmaxf(frame.mx[k], b.mx[k]); } }
    Bvh4Node node;
    memset(&node, 0, sizeof(node));
    float scale[3];
    for (int k = 0; k < 3; ++k) {
        node.origin[k] = frame.mn[k];
        const float ext = frame.mx[k] - frame.mn[k];
        int e = -120;
        if (ext > 0.0f) { (void)frexpf(ext / 254.0f, &e); }     // smallest power of two with ext / 2^e <= 254
        e = min(127, max(-120, e));
        scale[k] = ldexpf(1.0f, e);
        node.exp8[k] = uint8_t(e + 127);
    }
    node.exp8[3] = uint8_t(nch);
    atomicMax(&counters[2], it.depth + uint32_t(nch));     // stack entries below this node + what its visit can push, + 1
    for (int i = 0; i < 4; ++i) {
        if (i >= nch) {
            node.child[i] = 0xffffffffu;
            for (int k = 0; k < 3; ++k) { node.qlo[k][i] = 255; node.qhi[k][i] = 0; }
            continue;
        }
        const Box6 b = nbox[n == 1 ? 0 : ch[i]];
        for (int k = 0; k < 3; ++k) {
#pragma clang fp contract(off)
            const float inv = 1.0f / scale[k];
            int lo = int(floorf((b.mn[k] - node.origin[k]) * inv)), hi = int(ceilf((b.mx[k] - node.origin[k]) * inv));
            lo = min(255, max(0, lo)); hi = min(255, max(0, hi));
            while (lo > 0 && node.origin[k] + float(lo) * scale[k] > b.mn[k]) --lo;
            while (hi < 255 && node.origin[k] + float(hi) * scale[k] < b.mx[k]) ++hi;
            node.qlo[k][i] = uint8_t(lo); node.qhi[k][i] = uint8_t(hi);
        }
        if (n == 1) node.child[i] = KJ_BVH_LEAF;
        else if (is_leaf(ch[i])) {
            const uint32_t first = ch[i] >= uint32_t(n - 1) ? ch[i] - uint32_t(n - 1) : range[ch[i]].x;
            const uint32_t cnt = ch[i] >= uint32_t(n - 1) ? 1u : range[ch[i]].y - range[ch[i]].x + 1u;
            node.child[i] = KJ_BVH_LEAF | ((cnt - 1u) << 28) | first;
        } else {
            const uint32_t o = atomicAdd(&counters[1], 1u);
            node.child[i] = node_base + o;
            out[atomicAdd(&queue_len[level + 1], 1u)] = CollapseItem{ch[i], o, it.depth + uint32_t(nch - 1)};
        }
    }
    nodes[it.out] = node;
    }
}
// after a level: how many nodes exist now (= where the next level's nodes start)
__global__ void k_lbvh_level_end(const uint32_t* __restrict__ counters, uint32_t* __restrict__ level_nodes, uint32_t level) { level_nodes[level + 1] = counters[1]; }
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

}  // namespace
#endif

namespace kj {

#ifdef KJ_HIP_EMU_HOST
hipError_t build_blas_lbvh_device(const uint8_t*, const GpuMesh&, uint32_t, Bvh4Node*, BvhTri*, LbvhResult*, LbvhScratch*, hipStream_t) { return hipErrorInvalidValue; }   // the CPU stand-in has no device sort: fast-build meshes need the real device
#else
#define KJ_LB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)
hipError_t build_blas_lbvh_device(const uint8_t* d_vertex_buffer, const GpuMesh& mesh, uint32_t node_base, Bvh4Node* d_nodes_out, BvhTri* d_tris_out, LbvhResult* result, LbvhScratch* scratch, hipStream_t s) {
    const uint32_t n = mesh.index_count / 3;
    if (n == 0 || !scratch) return hipErrorInvalidValue;
    // working set: 15 device buffers, kept by the caller across the meshes of a commit (allocating and freeing them per mesh cost a
    // third of a nine-mesh build: hipFree synchronises the device) and grown when a larger mesh comes along
    if (scratch->capacity < n) {
        const size_t c = size_t(n) + n / 4;
        const size_t sizes[LbvhScratch::BUFFERS] = {c * sizeof(Box6), 32, c * 8, c * 4, c * 8, c * 4, c * 8, c * 8, 2 * c * 4, c * 4, 2 * c * sizeof(Box6),
                                                    (c + 1) * sizeof(CollapseItem), (c + 1) * sizeof(CollapseItem), 16};
        for (int k = 0; k < LbvhScratch::BUFFERS; ++k) KJ_LB(scratch->buf[k].alloc(sizes[k], s));
        scratch->capacity = uint32_t(c);
    }
    DevBuf &pbox = scratch->buf[0], &ob = scratch->buf[1], &codes = scratch->buf[2], &ids = scratch->buf[3], &codes2 = scratch->buf[4], &ids2 = scratch->buf[5],
           &children = scratch->buf[6], &range = scratch->buf[7], &parent = scratch->buf[8], &visits = scratch->buf[9], &nbox = scratch->buf[10], &q0 = scratch->buf[11],
           &q1 = scratch->buf[12], &counters = scratch->buf[13], &tmp = scratch->tmp;
    KJ_LB(hipMemsetAsync(visits.p, 0, size_t(n) * 4, s));      // the refit's arrival counters
    const dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(64), 0, s, (uint32_t*)ob.p);
    hipLaunchKernelGGL(k_lbvh_prims, g, b, 0, s, d_vertex_buffer, mesh, n, (Box6*)pbox.p, (uint32_t*)ob.p);
    hipLaunchKernelGGL(k_lbvh_morton, g, b, 0, s, (const Box6*)pbox.p, (const uint32_t*)ob.p, n, (MortonCode*)codes.p, (uint32_t*)ids.p);
    size_t tmp_bytes = 0;
    KJ_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const MortonCode*)codes.p, (MortonCode*)codes2.p, (const uint32_t*)ids.p, (uint32_t*)ids2.p, int(n), 0, 3 * KJ_MORTON_BITS, s));
    if (tmp.bytes < (tmp_bytes ? tmp_bytes : 16)) KJ_LB(tmp.alloc(tmp_bytes ? tmp_bytes : 16, s));
    KJ_LB(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, (const MortonCode*)codes.p, (MortonCode*)codes2.p, (const uint32_t*)ids.p, (uint32_t*)ids2.p, int(n), 0, 3 * KJ_MORTON_BITS, s));
    if (n > 1) hipLaunchKernelGGL(k_lbvh_hierarchy, g, b, 0, s, (const MortonCode*)codes2.p, int(n), (uint2*)children.p, (uint2*)range.p, (uint32_t*)parent.p);
    else KJ_LB(hipMemsetAsync(parent.p, 0xff, 8, s));
    hipLaunchKernelGGL(k_lbvh_refit, g, b, 0, s, (const Box6*)pbox.p, (const uint32_t*)ids2.p, int(n), (const uint2*)children.p, (const uint32_t*)parent.p, (uint32_t*)visits.p, (Box6*)nbox.p);
    // collapse, level by level, without a read-back per level: counters = {-, nodes allocated, max stack}; queue_len[l] = items of level l;
    // level_nodes[l] = nodes allocated before level l's children (a level's nodes are one contiguous run). KJ_LBVH_LEVELS levels are
    // issued blind -- far more than a tree over distinct Morton codes needs (13 for 250 k triangles) --, then ONE read-back; a deeper
    // tree (many coincident centroids) continues level by level with a read-back each.
    constexpr uint32_t KJ_LBVH_LEVELS = 40;
    DevBuf &queue_len = scratch->queue_len, &level_nodes = scratch->level_nodes;
    KJ_LB(queue_len.alloc((KJ_LBVH_LEVELS + 2) * 4, s)); KJ_LB(level_nodes.alloc((KJ_LBVH_LEVELS + 2) * 4, s));      // (no-ops after the first mesh)
    KJ_LB(hipMemsetAsync(queue_len.p, 0, (KJ_LBVH_LEVELS + 2) * 4, s)); KJ_LB(hipMemsetAsync(level_nodes.p, 0, (KJ_LBVH_LEVELS + 2) * 4, s));
    const uint32_t init_counters[4] = {0u, 1u, 0u, 0u}, one = 1u;
    const CollapseItem root{0u, 0u, 0u};
    KJ_LB(hipMemcpyAsync(counters.p, init_counters, 16, hipMemcpyHostToDevice, s));
    KJ_LB(hipMemcpyAsync(queue_len.p, &one, 4, hipMemcpyHostToDevice, s));
    KJ_LB(hipMemcpyAsync((uint32_t*)level_nodes.p + 1, &one, 4, hipMemcpyHostToDevice, s));      // level 0 = the root = node 0
    KJ_LB(hipMemcpyAsync(q0.p, &root, sizeof(root), hipMemcpyHostToDevice, s));
    DevBuf* qin = &q0; DevBuf* qout = &q1;
    uint64_t bound = 1;
    for (uint32_t level = 0; level < KJ_LBVH_LEVELS; ++level) {
        const uint32_t items = uint32_t(std::min<uint64_t>(bound, n));
        hipLaunchKernelGGL(k_lbvh_collapse, dim3(std::min(4096u, (items + 63) / 64)), dim3(64), 0, s, int(n), (const uint2*)children.p, (const uint2*)range.p, (const Box6*)nbox.p,
                           (const CollapseItem*)qin->p, (uint32_t*)queue_len.p, level, (CollapseItem*)qout->p, (uint32_t*)counters.p, d_nodes_out, node_base);
        hipLaunchKernelGGL(k_lbvh_level_end, dim3(1), dim3(1), 0, s, (const uint32_t*)counters.p, (uint32_t*)level_nodes.p, level + 1);
        bound = std::min<uint64_t>(bound * 4, uint64_t(n));
        std::swap(qin, qout);
    }
    uint32_t host_counters[4] = {0, 1, 0, 0}, host_levels[KJ_LBVH_LEVELS + 2], host_queue[KJ_LBVH_LEVELS + 2];
    KJ_LB(hipMemcpyAsync(host_levels, level_nodes.p, sizeof(host_levels), hipMemcpyDeviceToHost, s));
    KJ_LB(hipMemcpyAsync(host_queue, queue_len.p, sizeof(host_queue), hipMemcpyDeviceToHost, s));
    KJ_LB(hipMemcpyAsync(host_counters, counters.p, 16, hipMemcpyDeviceToHost, s));
    KJ_LB(hipStreamSynchronize(s));
    result->level_starts.assign({0u});
    for (uint32_t l = 1; l <= KJ_LBVH_LEVELS + 1; ++l)
        if (host_levels[l] > result->level_starts.back()) result->level_starts.push_back(host_levels[l]);
    uint32_t in_count = host_queue[KJ_LBVH_LEVELS];
    while (in_count) {      // deeper than the blind part: one level at a time
        const uint32_t lens[2] = {in_count, 0u};
        KJ_LB(hipMemcpyAsync(queue_len.p, lens, 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_lbvh_collapse, dim3(std::min(4096u, (in_count + 63) / 64)), dim3(64), 0, s, int(n), (const uint2*)children.p, (const uint2*)range.p, (const Box6*)nbox.p,
                           (const CollapseItem*)qin->p, (uint32_t*)queue_len.p, 0u, (CollapseItem*)qout->p, (uint32_t*)counters.p, d_nodes_out, node_base);
        KJ_LB(hipMemcpyAsync(host_queue, queue_len.p, 8, hipMemcpyDeviceToHost, s));
        KJ_LB(hipMemcpyAsync(host_counters, counters.p, 16, hipMemcpyDeviceToHost, s));
        KJ_LB(hipStreamSynchronize(s));
        in_count = host_queue[1];
        if (host_counters[1] > result->level_starts.back()) result->level_starts.push_back(host_counters[1]);
        std::swap(qin, qout);
    }
    hipLaunchKernelGGL(k_lbvh_emit_tris, g, b, 0, s, d_vertex_buffer, mesh, (const uint32_t*)ids2.p, n, d_tris_out);
    uint32_t hob[8];
    KJ_LB(hipMemcpyAsync(hob, ob.p, 24, hipMemcpyDeviceToHost, s));
    KJ_LB(hipStreamSynchronize(s));
    KJ_LB(hipGetLastError());
    for (int k = 0; k < 6; ++k) {
        const uint32_t o = hob[k];
        const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
        memcpy(&result->bounds[k], &u, 4);
    }
    result->node_count = host_counters[1];
    result->max_stack = host_counters[2] > 0 ? host_counters[2] : 1;
    return hipSuccess;
}
#endif

}  // namespace kj
