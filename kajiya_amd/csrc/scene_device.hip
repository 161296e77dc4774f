// Device-side parts of the scene's acceleration structures.
#include "kj_host.hpp"
#include "kj_scene_device.hpp"
#include "kj_vec.hpp"

using namespace kj;

// World-space copy of an instance's triangles: the row-major 3x4 transform applied to every vertex with each product and sum
// rounded separately -- bit for bit what flattening the scene on the host gives (scene.cpp round 1, the oracle's OracleScene).
// world_id = position in the scene-wide numbering (first triangle of the instance + the triangle's index in the mesh).
__global__ void __launch_bounds__(256) k_instance_triangles(const BvhTri* __restrict__ obj_tris, BvhTri* __restrict__ world_tris, const InstanceTriJob* __restrict__ jobs) {
#pragma clang fp contract(off)
    const InstanceTriJob j = jobs[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < j.count; i += gridDim.x * 256) {
        const BvhTri t = obj_tris[j.src + i];
        BvhTri o;
        const float* x = j.xform;
        const float* src[3] = {t.v0, t.v1, t.v2};
        float* dst[3] = {o.v0, o.v1, o.v2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* p = src[k];
            dst[k][0] = x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[3];
            dst[k][1] = x[4] * p[0] + x[5] * p[1] + x[6] * p[2] + x[7];
            dst[k][2] = x[8] * p[0] + x[9] * p[1] + x[10] * p[2] + x[11];
        }
        o.world_id = j.dst + t.prim;
        o.inst = j.instance;
        o.prim = t.prim;
        world_tris[j.dst + i] = o;
    }
}

// Each instance record carries a copy of its mesh's BLAS root node (kj_scene_types.hpp): 4 lanes x 16 B per record.
__global__ void __launch_bounds__(256) k_instance_roots(InstanceRecord* __restrict__ recs, const Bvh4Node* __restrict__ blas_nodes, uint32_t count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, r = i >> 2, q = i & 3u;
    if (r >= count) return;
    const uint4* src = (const uint4*)(blas_nodes + recs[r].node_root);
    ((uint4*)&recs[r].root)[q] = src[q];
}

namespace kj {

hipError_t launch_instance_roots(InstanceRecord* recs, const Bvh4Node* blas_nodes, uint32_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_instance_roots, dim3((count * 4 + 255) / 256), dim3(256), 0, s, recs, blas_nodes, count);
    return hipGetLastError();
}

hipError_t launch_instance_triangles(const BvhTri* obj_tris, BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s) {
    if (job_count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_instance_triangles, dim3(64, job_count), dim3(256), 0, s, obj_tris, world_tris, jobs);
    return hipGetLastError();
}

}  // namespace kj
