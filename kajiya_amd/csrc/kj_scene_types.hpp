// POD types shared by host code (scene.cpp, plain C++) and device code (*.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/kajiya_amd.h"

namespace kj {

struct alignas(16) F4 { float x, y, z, w; };

// 64 B: both children's AABBs + child references (see kj_bvh.hpp)
struct BvhNode {
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
// 48 B: world-space triangle in leaf order
struct BvhTri {
    float v0[3]; uint32_t world_id;
    float v1[3]; uint32_t inst;
    float v2[3]; uint32_t prim;
};
static_assert(sizeof(BvhNode) == 64, "node size");
static_assert(sizeof(BvhTri) == 48, "tri size");

#define KJ_BVH_LEAF 0x80000000u
#define KJ_BVH_MAX_LEAF_TRIS 4u

struct BvhView {
    const F4* nodes;         // 4 x 16 B per node
    const F4* tris;          // 3 x 16 B per tri
    uint32_t root;           // child reference of the root
    uint32_t stack_entries;  // per-lane LDS stack depth a tracing kernel must provide
};

struct GpuMesh {  // inc/mesh.hlsl:10-18 (+ index_count)
    uint32_t vertex_core_offset, vertex_uv_offset, vertex_mat_offset, vertex_aux_offset, vertex_tangent_offset, mat_data_offset, index_offset;
    uint32_t index_count;
};
struct GpuInstance {
    float xform[12];  // row-major 3x4 object->world
    uint32_t mesh;
    float emissive_multiplier;
    uint32_t pad0, pad1;
};
struct SceneView {
    const uint8_t* vertex_buffer;
    const GpuMesh* meshes;
    const GpuInstance* instances;
    const F4* map_colors;   // bindless "textures": 1x1 placeholders => constant colour
    const KjTriangleLight* lights;
    uint32_t light_count;
    BvhView bvh;
};

} // namespace kj
