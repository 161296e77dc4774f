// Device-side parts of the scene's acceleration structures (scene_device.hip), called from scene.cpp's commit.
#pragma once
#include <hip/hip_runtime_api.h>
#include "kj_scene_types.hpp"

namespace kj {

// One instance's world-space triangles: dst[i] = xform * src[i] for `count` triangles of the mesh's leaf-ordered list.
struct InstanceTriJob { float xform[12]; uint32_t src, dst, count, instance; };
static_assert(sizeof(InstanceTriJob) == 64, "job size");
hipError_t launch_instance_triangles(const BvhTri* obj_tris, BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s);

// recs[i].root = blas_nodes[recs[i].node_root] for every instance record.
hipError_t launch_instance_roots(InstanceRecord* recs, const Bvh4Node* blas_nodes, uint32_t count, hipStream_t s);

// A mesh's BLAS built on the device as a linear BVH (lbvh_build.hip): nodes into d_nodes_out[0 .. node_count) with child node
// indices offset by node_base, object-space triangles in leaf order into d_tris_out[0 .. index_count / 3). Synchronises the stream.
struct LbvhResult { float bounds[6]; uint32_t node_count, max_stack; };
hipError_t build_blas_lbvh_device(const uint8_t* d_vertex_buffer, const GpuMesh& mesh, uint32_t node_base, Bvh4Node* d_nodes_out, BvhTri* d_tris_out, LbvhResult* result, hipStream_t s);

}  // namespace kj
