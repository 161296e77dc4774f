// Host-side BVH build interface (bvh_build.cpp).
#pragma once
#include <vector>
#include "kj_scene_types.hpp"

namespace kj {

struct BuiltBvh {
    std::vector<BvhNode> nodes;    // node 0 is the root (Bvh4Node or Bvh8Node, KJ_BVH_WIDTH)
    std::vector<BvhTri> tris;      // leaf order
    uint32_t max_stack = 1;        // upper bound of the traversal stack depth
};
// `tris`: any space (a mesh's object space for a BLAS). For a TLAS pass one degenerate "triangle" per instance whose vertices span the
// instance's world box (v0 = min, v1 = max, v2 = min; prim = instance index) and max_leaf = 1.
void build_bvh4(const std::vector<BvhTri>& tris, BuiltBvh& out, uint32_t max_leaf = KJ_BVH_MAX_LEAF_TRIS);

}  // namespace kj
