// Device-side parts of the scene's acceleration structures (scene_device.hip), called from scene.cpp's commit.
#pragma once
#include <hip/hip_runtime_api.h>
#include <vector>
#include "kj_scene_types.hpp"

namespace kj {

// One instance's world-space triangles: dst[i] = xform * src[i] for `count` triangles of the mesh's leaf-ordered list.
// `id_base`: the instance's first WORLD TRIANGLE ID -- the dense numbering over the live instances in slot order that ray queries report
// and equal-t ties are broken by; `dst`: where its triangles live in the world array (the same number until an instance is removed:
// a removal leaves a hole in the array, the ids stay dense).
struct InstanceTriJob { float xform[12]; uint32_t src, dst, count, instance, id_base, pad[3]; };
static_assert(sizeof(InstanceTriJob) == 80, "job size");
hipError_t launch_instance_triangles(const BvhTri* obj_tris, BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s);
// only the ids again (after a removal shifted the numbering of the instances behind it): 4 bytes per triangle instead of the transform + refit
hipError_t launch_instance_renumber(BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s);

// One instance's world-space tree: nodes[dst + i] = BLAS node blas_nodes[src + i] refit around world_tris[tri_base ..] (which must have
// been derived already on the same stream), child references rebased to `dst` / `tri_base`.
//   steps[heights + h]    : {first, end} node (relative) of the h-th step of the bottom-up order, h in [0, height_count): a step only reads
//                           boxes that earlier steps wrote. Host-built BLASes are sorted by node height (0 = all children are leaves, the
//                           root last); device-built ones are laid out by depth and walked deepest level first.
//   wide_heights          : steps [0, wide_heights) get a launch of their own, the rest (each <= KJ_REFIT_TOP_NODES nodes) one workgroup
//   boxes                 : 24 B per world node (scratch)
#define KJ_REFIT_TOP_NODES 1024
struct InstanceRefitJob { uint32_t src, dst, node_count, tri_base, heights, height_count, wide_heights, pad; };
hipError_t launch_instance_refit(const Bvh4Node* blas_nodes, const uint2* steps, const BvhTri* world_tris, const InstanceRefitJob* jobs, uint32_t job_count,
                                 uint32_t max_wide_heights, Bvh4Node* nodes, void* boxes, hipStream_t s);

// A mesh's BLAS built on the device as a linear BVH (lbvh_build.hip): nodes into d_nodes_out[0 .. node_count) with child node
// indices offset by node_base, object-space triangles in leaf order into d_tris_out[0 .. index_count / 3). Synchronises the stream.
struct LbvhResult { static constexpr size_t HEAD_NODES = 341; float bounds[6]; uint32_t node_count, max_stack; std::vector<uint32_t> level_starts; std::vector<Bvh4Node> head; /* the first nodes (the tree is laid out level by level) */ };   // level d (0 = the root) = nodes [level_starts[d], level_starts[d + 1])
// The builder's working set: ONE allocation, grown when a call needs more and carved into the call's buffers (round 6: it was 17 allocations per commit, and as many
// hipFree calls -- each a device synchronisation -- when the commit's scratch went out of scope).
struct LbvhScratch { DevBuf arena; std::vector<uint32_t> readback; /* host copy of the builder's read-back block */ };
// The meshes of a commit as ONE batch: the per-mesh stages one after the other, the 4-wide collapse level by level for all meshes together, one read-back.
struct LbvhBatchMesh { GpuMesh mesh; uint32_t node_base; Bvh4Node* nodes_out; BvhTri* tris_out; LbvhResult* result; };
// nodes[0 .. count) of a tree with mesh-relative child indices -> dst, inner child indices + node_base
hipError_t launch_blas_place_nodes(const Bvh4Node* src, Bvh4Node* dst, uint32_t count, uint32_t node_base, hipStream_t s);
hipError_t build_blas_lbvh_device_batch(const uint8_t* d_vertex_buffer, const LbvhBatchMesh* batch, uint32_t count, LbvhScratch* scratch, hipStream_t s, bool ploc);
hipError_t build_blas_lbvh_device(const uint8_t* d_vertex_buffer, const GpuMesh& mesh, uint32_t node_base, Bvh4Node* d_nodes_out, BvhTri* d_tris_out, LbvhResult* result, LbvhScratch* scratch, hipStream_t s, bool ploc);   // ploc: hierarchy by agglomerative clustering instead of Morton-code splits

// The per-commit top tree built on the device: a linear BVH over the leaves' padded world boxes (six floats each: min xyz, max xyz), every leaf holding one
// box and becoming the child reference d_leaf_refs[i] (the world node the leaf stands for). For commits with thousands of instances, where the host's SAH
// build of the top tree is what a per-frame commit costs (profiles/r03_top_tree_build.md). Nodes into d_nodes_out[0 .. node_count); one synchronisation.
hipError_t build_top_lbvh_device(const float* d_leaf_boxes, const uint32_t* d_leaf_refs, uint32_t leaf_count, Bvh4Node* d_nodes_out, LbvhResult* result, LbvhScratch* scratch, hipStream_t s);

}  // namespace kj
