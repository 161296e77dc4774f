// POD types shared by host code (scene.cpp, plain C++) and device code (*.hip).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/kajiya_amd.h"

namespace kj {

struct alignas(16) F4 { float x, y, z, w; };

// 64 B: 4-wide node, child boxes quantised to 8 bits per plane in the node's own frame (see kj_bvh.hpp).
//   decoded plane = fma(q, 2^(exp8 - 127), origin); the builder rounds outwards and verifies with this exact expression.
struct Bvh4Node {
    float origin[3];
    uint8_t exp8[4];        // per-axis IEEE exponent byte of the quantisation step; [3] = number of children
    uint32_t child[4];      // inner: node index; leaf: KJ_BVH_LEAF | (count-1) << 28 | first triangle; 0xffffffff = empty
    uint8_t qlo[3][4];      // [axis][child]
    uint8_t qhi[3][4];
    uint32_t pad[2];
};
// 80 B: 8-wide node (KJ_BVH_WIDTH 8). Same quantised child boxes; the children are stored contiguously so that one base index per kind
// replaces the eight references: inner child i lives at node `child_base + rank`, leaf child i owns triangles from `tri_base + offset`.
//   meta[i]: 0xff = empty; inner: rank among the inner children (0..7); leaf: 0x80 | (count - 1) << 5 | offset (<= 28: 8 leaves x 4 triangles)
struct Bvh8Node {
    float origin[3];
    uint8_t exp8[4];        // [3] = number of children (they occupy slots 0..n-1)
    uint32_t child_base, tri_base;
    uint8_t meta[8];
    uint8_t qlo[3][8];      // [axis][child]
    uint8_t qhi[3][8];
};
static_assert(sizeof(Bvh8Node) == 80, "node size");
#ifndef KJ_BVH_WIDTH
#define KJ_BVH_WIDTH 4
#endif
#if KJ_BVH_WIDTH == 8
typedef Bvh8Node BvhNode;
#define KJ_BVH_NODE_F4 5
#else
typedef Bvh4Node BvhNode;
#define KJ_BVH_NODE_F4 4
#endif
// 48 B: world-space triangle in leaf order
struct BvhTri {
    float v0[3]; uint32_t world_id;
    float v1[3]; uint32_t inst;
    float v2[3]; uint32_t prim;
};
static_assert(sizeof(Bvh4Node) == 64, "node size");
static_assert(sizeof(BvhTri) == 48, "tri size");

#define KJ_BVH_LEAF 0x80000000u
#define KJ_BVH_MAX_LEAF_TRIS 4u
#define KJ_BVH_LDS_STACK 16u      // traversal stack entries kept in LDS per lane ...
#define KJ_BVH_SPILL_STACK 112u   // ... deeper entries spill to private (scratch) memory; builds needing more are rejected

// The reference's TLAS over per-mesh BLASes (kajiya-backend/src/vulkan/ray_tracing.rs:96-275), laid out so that a ray walks ONE tree in
// world space -- no ray transform, no per-instance state in the traversal loop:
//   * per MESH, built once: the BLAS topology (Bvh4Node tree over the mesh's object-space triangles, host SAH or device LBVH) and the
//     triangles in leaf order, plus the nodes listed by height for the refit below.
//   * per INSTANCE, re-derived on the device when the instance moves (scene_device.hip): its triangles in WORLD space (the same fp32
//     arithmetic as flattening the scene, so hits are those of a world-space scene) and a copy of the mesh's tree whose boxes are refit
//     bottom-up around those world-space triangles and re-quantised -- tight under rotation, unlike transformed object-space boxes.
//   * per COMMIT: a small tree over the instances' world boxes (host), whose leaf children point straight at the instances' root nodes.
// All of it lives in one node array: [0, tlas_capacity) the top tree (node 0 = root), then one region per instance.
struct BvhView {
    const F4* nodes;         // 4 x 16 B per node; node 0 is the root
    const F4* tris;          // 3 x 16 B per tri
    uint32_t root;           // always 0 (kept for the C-ABI debug query)
    uint32_t stack_entries;  // dynamic LDS a tracing kernel must provide, in units of 64 dwords: the KJ_BVH_LDS_STACK levels of the per-lane stacks
};

// 32 B per material map. flags: bits 0-7 = mip count (0 => 1x1 placeholder, `color`), bit 8 = sRGB texels.
struct MapDesc { F4 color; uint32_t offset, width, height, flags; };

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
    const MapDesc* maps;    // bindless "textures" (inc/bindless_textures.hlsl): placeholder colour or an RGBA8 mip chain in tex_data
    const uint8_t* tex_data;
    const KjTriangleLight* lights;
    uint32_t light_count;
    BvhView bvh;
};

} // namespace kj
