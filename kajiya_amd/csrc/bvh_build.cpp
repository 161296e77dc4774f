// Host-side acceleration-structure build — the role of vkCmdBuildAccelerationStructuresKHR in the reference
// (kajiya-backend/src/vulkan/ray_tracing.rs:96-275,408-570): world-space triangles in, a 4-wide BVH with
// quantised child boxes out (layout in kj_scene_types.hpp, traversal in kj_bvh.hpp).
//
//   1. binary BVH by binned SAH (16 bins x 3 axes, leaves of <= 4 triangles, median fallback);
//   2. collapse to KJ_BVH_WIDTH-wide (4 or 8): repeatedly open the child with the largest surface area;
//   3. quantise each child box to 8 bits per plane inside its parent's frame, rounding outwards and
//      verifying with the exact decode the kernel uses (fma(q, scale, origin)), so a decoded box always
//      contains the true one: traversal never loses a hit, and the triangles themselves stay fp32.
// Plain C++ (no device code), compiled with -ffp-contract=off.
#include "kj_bvh_build.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace kj {

namespace {

struct Box {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* p) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
    void grow(const Box& o) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], o.mn[k]); mx[k] = std::max(mx[k], o.mx[k]); } }
    float half_area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

struct BinNode {          // binary tree
    Box box;
    uint32_t left = 0, right = 0;   // children (inner) ...
    uint32_t first = 0, count = 0;  // ... or primitive range (leaf when count > 0)
};

constexpr int NBINS = 16;
constexpr uint32_t MAX_BINARY_DEPTH = 56;

struct SahBuilder {
    const std::vector<BvhTri>& tris;
    std::vector<Box> pbox;
    std::vector<float> cen;        // 3 per prim
    std::vector<uint32_t> idx;     // permutation
    std::vector<BinNode> nodes;
    uint32_t max_leaf = KJ_BVH_MAX_LEAF_TRIS;   // 1 for a TLAS (one instance per leaf)

    explicit SahBuilder(const std::vector<BvhTri>& t) : tris(t) {
        const size_t n = t.size();
        pbox.resize(n); cen.resize(n * 3); idx.resize(n);
        std::iota(idx.begin(), idx.end(), 0u);
        const uint32_t nt = n >= 65536 ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        auto fill = [&](size_t a, size_t e) {
            for (size_t i = a; i < e; ++i) {
                Box b; b.grow(t[i].v0); b.grow(t[i].v1); b.grow(t[i].v2);
                pbox[i] = b;
                for (int k = 0; k < 3; ++k) cen[i * 3 + k] = 0.5f * (b.mn[k] + b.mx[k]);
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t k = 1; k < nt; ++k) pool.emplace_back(fill, n * k / nt, n * (k + 1) / nt);
        fill(0, n / nt);
        for (auto& th : pool) th.join();
        nodes.reserve(n);
    }

    // Subtrees of at most `grain` primitives are not built here when `deferred` is given: a placeholder node is emitted and the range is
    // recorded, to be built by a worker thread into its own vector and spliced in afterwards (build_parallel). A subtree only reads and
    // permutes its own slice of `idx`, so the result is the same tree whichever thread builds it, and whatever the node numbering.
    struct Deferred { uint32_t placeholder, first, count, depth; };
    uint32_t build(std::vector<BinNode>& nodes, uint32_t first, uint32_t count, uint32_t depth, uint32_t grain = 0, std::vector<Deferred>* deferred = nullptr) {
        const uint32_t me = uint32_t(nodes.size());
        nodes.emplace_back();
        if (deferred && count <= grain && depth > 0) { deferred->push_back({me, first, count, depth}); return me; }
        Box box, cbox;
        for (uint32_t i = first; i < first + count; ++i) { box.grow(pbox[idx[i]]); cbox.grow(&cen[size_t(idx[i]) * 3]); }
        nodes[me].box = box;
        if (count <= 1) { nodes[me].first = first; nodes[me].count = count; return me; }
        // --- binned SAH over the three axes
        float best_cost = FLT_MAX; int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            const float lo = cbox.mn[ax], ext = cbox.mx[ax] - cbox.mn[ax];
            if (!(ext > 0.0f)) continue;
            Box bb[NBINS]; uint32_t bc[NBINS] = {};
            const float scale = float(NBINS) / ext;
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t p = idx[i];
                const int b = std::min(NBINS - 1, std::max(0, int((cen[size_t(p) * 3 + ax] - lo) * scale)));
                bb[b].grow(pbox[p]); bc[b]++;
            }
            float ra[NBINS]; uint32_t rc[NBINS];
            Box acc; uint32_t cnt = 0;
            for (int b = NBINS - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; ra[b] = acc.half_area(); rc[b] = cnt; }
            acc = Box(); cnt = 0;
            for (int b = 0; b < NBINS - 1; ++b) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || rc[b + 1] == 0) continue;
                const float cost = acc.half_area() * float(cnt) + ra[b + 1] * float(rc[b + 1]);
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }
        const float parent_area = std::max(box.half_area(), 1e-30f);
        const bool can_be_leaf = count <= max_leaf;
        if (can_be_leaf && (best_axis < 0 || 1.0f + best_cost / parent_area >= float(count))) {
            nodes[me].first = first; nodes[me].count = count; return me;
        }
        uint32_t mid;
        if (best_axis >= 0 && depth < MAX_BINARY_DEPTH) {
            const float lo = cbox.mn[best_axis], scale = float(NBINS) / (cbox.mx[best_axis] - cbox.mn[best_axis]);
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t p) {
                return std::min(NBINS - 1, std::max(0, int((cen[size_t(p) * 3 + best_axis] - lo) * scale))) <= best_bin;
            });
            mid = uint32_t(it - idx.begin());
        } else {  // coincident centroids or runaway depth: split the range in half (along the widest axis if there is one)
            int ax = 0;
            for (int k = 1; k < 3; ++k) if (cbox.mx[k] - cbox.mn[k] > cbox.mx[ax] - cbox.mn[ax]) ax = k;
            mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return cen[size_t(a) * 3 + ax] < cen[size_t(b) * 3 + ax]; });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        const uint32_t l = build(nodes, first, mid - first, depth + 1, grain, deferred);
        const uint32_t r = build(nodes, mid, first + count - mid, depth + 1, grain, deferred);
        nodes[me].left = l; nodes[me].right = r;
        return me;
    }

    // The whole tree into `nodes`, root at the returned index: the top levels on this thread, the deferred subtrees on `n_threads` workers.
    uint32_t build_parallel(uint32_t n_threads) {
        const uint32_t n = uint32_t(tris.size());
        if (n_threads <= 1 || n < 65536) return build(nodes, 0, n, 0);
        std::vector<Deferred> deferred;
        const uint32_t root = build(nodes, 0, n, 0, std::max(4096u, n / (n_threads * 8u)), &deferred);
        if (getenv("KJ_BVH_TIMING")) fprintf(stderr, "[bvh build] top levels done: %zu nodes, %zu deferred subtrees, %u threads\n", nodes.size(), deferred.size(), n_threads);
        std::vector<std::vector<BinNode>> local(deferred.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < deferred.size();) {
                local[i].reserve(size_t(deferred[i].count) * 2);
                build(local[i], deferred[i].first, deferred[i].count, deferred[i].depth);
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        for (size_t i = 0; i < deferred.size(); ++i) {       // splice: local root replaces the placeholder, the rest is appended
            const uint32_t base = uint32_t(nodes.size());
            auto remap = [&](BinNode b) { if (b.count == 0) { b.left += base - 1; b.right += base - 1; } return b; };
            nodes[deferred[i].placeholder] = remap(local[i][0]);
            for (size_t j = 1; j < local[i].size(); ++j) nodes.push_back(remap(local[i][j]));
        }
        return root;
    }
};

inline float dec(uint32_t q, float scale, float origin) { return origin + float(q) * scale; }   // == fmaf(q, scale, origin): q*scale is exact

}  // namespace

void build_bvh4(const std::vector<BvhTri>& world_tris, BuiltBvh& out, uint32_t max_leaf) {
    const bool timing = getenv("KJ_BVH_TIMING") != nullptr;     // prints the phase times of the host build to stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) fprintf(stderr, "[bvh build] %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    SahBuilder sb(world_tris);
    sb.max_leaf = std::min(std::max(max_leaf, 1u), KJ_BVH_MAX_LEAF_TRIS);
    lap("primitive boxes");
    uint32_t n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("KJ_BVH_THREADS")) n_threads = uint32_t(std::max(1, atoi(e)));   // 1 = the sequential build (tests compare the two)
    const uint32_t root = sb.build_parallel(n_threads);
    lap("binary SAH tree");
    const std::vector<BinNode>& bn = sb.nodes;
    out.nodes.clear(); out.tris.clear();
    out.nodes.reserve(bn.size() / 2 + 1);
    out.tris.reserve(world_tris.size());
    out.max_stack = 1;

    // Wide nodes are emitted depth-first (a node's subtree is contiguous); leaves append their triangles in visit order.
    struct Item { uint32_t bin; uint32_t wide; uint32_t stack_before; };
    std::vector<Item> work;
    auto emit_leaf = [&](const BinNode& n) -> uint32_t {
        const uint32_t first = uint32_t(out.tris.size());
        for (uint32_t i = 0; i < n.count; ++i) out.tris.push_back(world_tris[sb.idx[n.first + i]]);
        return KJ_BVH_LEAF | ((n.count - 1) << 28) | first;
    };
    out.nodes.emplace_back();
    work.push_back({root, 0, 0});
    constexpr int W = KJ_BVH_WIDTH;
    while (!work.empty()) {
        const Item it = work.back(); work.pop_back();
        // gather up to W children by opening the inner child with the largest surface area
        uint32_t ch[W]; int nch = 0;
        if (bn[it.bin].count > 0) ch[nch++] = it.bin;   // root is a leaf (tiny scene)
        else { ch[nch++] = bn[it.bin].left; ch[nch++] = bn[it.bin].right; }
        while (nch < W) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nch; ++i)
                if (bn[ch[i]].count == 0) { const float a = bn[ch[i]].box.half_area(); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const uint32_t open = ch[best];
            ch[best] = bn[open].left; ch[nch++] = bn[open].right;
        }
        Box frame;
        for (int i = 0; i < nch; ++i) frame.grow(bn[ch[i]].box);
        BvhNode node;
        memset(&node, 0, sizeof(node));
        float scale[3];
        for (int k = 0; k < 3; ++k) {
            node.origin[k] = frame.mn[k];
            const float ext = frame.mx[k] - frame.mn[k];
            int e = 0;
            // smallest power of two with ext / 2^e <= 254 (one code of headroom for the outward rounding)
            if (ext > 0.0f) { (void)frexpf(ext / 254.0f, &e); } else e = -120;
            e = std::min(127, std::max(-120, e));
            scale[k] = ldexpf(1.0f, e);
            node.exp8[k] = uint8_t(e + 127);
        }
        node.exp8[3] = uint8_t(nch);
        uint32_t child_ref[W];
#if KJ_BVH_WIDTH == 8
        node.child_base = uint32_t(out.nodes.size());
        node.tri_base = uint32_t(out.tris.size());
        uint32_t inner_rank = 0;
#endif
        for (int i = 0; i < W; ++i) {
            if (i >= nch) {  // empty slot: inverted box, never hit
#if KJ_BVH_WIDTH == 8
                node.meta[i] = 0xffu;
#else
                node.child[i] = 0xffffffffu;
#endif
                child_ref[i] = 0xffffffffu;
                for (int k = 0; k < 3; ++k) { node.qlo[k][i] = 255; node.qhi[k][i] = 0; }
                continue;
            }
            const BinNode& c = bn[ch[i]];
            for (int k = 0; k < 3; ++k) {
                const float inv = 1.0f / scale[k];
                int lo = int(std::floor((c.box.mn[k] - node.origin[k]) * inv));
                int hi = int(std::ceil((c.box.mx[k] - node.origin[k]) * inv));
                lo = std::min(255, std::max(0, lo)); hi = std::min(255, std::max(0, hi));
                while (lo > 0 && dec(uint32_t(lo), scale[k], node.origin[k]) > c.box.mn[k]) --lo;
                while (hi < 255 && dec(uint32_t(hi), scale[k], node.origin[k]) < c.box.mx[k]) ++hi;
                node.qlo[k][i] = uint8_t(lo); node.qhi[k][i] = uint8_t(hi);
            }
            if (c.count > 0) {
                child_ref[i] = emit_leaf(c);
#if KJ_BVH_WIDTH == 8
                node.meta[i] = uint8_t(0x80u | ((c.count - 1) << 5) | ((child_ref[i] & 0x0fffffffu) - node.tri_base));
#endif
            } else {
                child_ref[i] = uint32_t(out.nodes.size());
                out.nodes.emplace_back();
#if KJ_BVH_WIDTH == 8
                node.meta[i] = uint8_t(inner_rank++);
#endif
            }
#if KJ_BVH_WIDTH != 8
            node.child[i] = child_ref[i];
#endif
        }
        // traversal pushes at most (children hit - 1) entries per visited node
        const uint32_t stack_here = it.stack_before + uint32_t(nch > 0 ? nch - 1 : 0);
        out.max_stack = std::max(out.max_stack, stack_here + 1);
        for (int i = 0; i < nch; ++i)
            if (bn[ch[i]].count == 0) work.push_back({ch[i], child_ref[i], stack_here});
        out.nodes[it.wide] = node;
    }
    lap("collapse + quantise + emit");
}

}  // namespace kj
