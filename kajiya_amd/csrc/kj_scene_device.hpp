// Device-side parts of the scene's acceleration structures (scene_device.hip), called from scene.cpp's commit.
#pragma once
#include <hip/hip_runtime_api.h>
#include "kj_scene_types.hpp"

namespace kj {

// One instance's world-space triangles: dst[i] = xform * src[i] for `count` triangles of the mesh's leaf-ordered list.
struct InstanceTriJob { float xform[12]; uint32_t src, dst, count, instance; };
static_assert(sizeof(InstanceTriJob) == 64, "job size");
hipError_t launch_instance_triangles(const BvhTri* obj_tris, BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s);

}  // namespace kj
