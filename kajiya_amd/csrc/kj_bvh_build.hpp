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
void build_bvh4(const std::vector<BvhTri>& world_tris, BuiltBvh& out);

}  // namespace kj
