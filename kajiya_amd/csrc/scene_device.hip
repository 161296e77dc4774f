// Device-side parts of the scene's acceleration structures.
#include "kj_host.hpp"
#include "kj_scene_device.hpp"
#include "kj_vec.hpp"
#include <cfloat>
#include <algorithm>

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
        o.world_id = j.id_base + t.prim;
        o.inst = j.instance;
        o.prim = t.prim;
        world_tris[j.dst + i] = o;
    }
}

__global__ void __launch_bounds__(256) k_instance_renumber(BvhTri* __restrict__ world_tris, const InstanceTriJob* __restrict__ jobs) {
    const InstanceTriJob j = jobs[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < j.count; i += gridDim.x * 256) world_tris[j.dst + i].world_id = j.id_base + world_tris[j.dst + i].prim;
}

// One instance's world-space tree: the mesh's BLAS topology with every child box refit around the instance's WORLD-space triangles
// (k_instance_triangles ran before) and quantised in the node's own frame, child references rebased into the scene-wide arrays.
// Bottom-up by HEIGHT (0 = all children are leaves; else 1 + the tallest inner child), the mesh's BLAS having been sorted that way
// when it was built (the root is its last node): a height's nodes only read boxes of lower heights. The populous heights take one
// launch each over all jobs; the remaining small ones (<= KJ_REFIT_TOP_NODES nodes per height) are finished by one workgroup per job
// with a barrier between heights -- a device-wide bottom-up climb with arrival counters needs agent-scope fences, i.e. L2 write-backs
// and invalidations across the 8 XCDs at every step, and measured 0.17 ms for a single moved instance.
// The work is a chain of dependent round trips, so it is spread thin: FOUR lanes per node, one per child. A lane fetches its child's
// box (a leaf's <= 4 triangles with clamped indices: all loads in flight together; an inner child's box from the height below),
// the four meet in LDS for the node's frame, each quantises its own child, and the first lane stores the assembled 64 bytes.
struct Box6 { float mn[3], mx[3]; };
template <int THREADS> struct RefitShared { float box[THREADS / 4][4][6]; uint32_t q[THREADS / 4][4][3]; };
template <int THREADS>
KJ_D void refit_pass(const InstanceRefitJob& j, uint32_t first_node, uint32_t end, const Bvh4Node* __restrict__ blas, const BvhTri* __restrict__ world_tris,
                     Bvh4Node* __restrict__ nodes, Box6* __restrict__ boxes, RefitShared<THREADS>& sh) {
#pragma clang fp contract(off)
    const uint32_t g = threadIdx.x >> 2, ci = threadIdx.x & 3u, rel = first_node + g;
    const bool active = rel < end;
    const uint32_t c = active ? blas[j.src + rel].child[ci] : 0xffffffffu;
    const uint32_t children = active ? blas[j.src + rel].exp8[3] : 0u;
    const bool empty = c == 0xffffffffu, leaf = !empty && (c & KJ_BVH_LEAF);
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (leaf) {
        const uint32_t first = j.tri_base + (c & 0x0fffffffu), last = (c >> 28) & 7u;
        const float4* t0 = (const float4*)(world_tris + first);
        const float4* t1 = (const float4*)(world_tris + first + min(1u, last));
        const float4* t2 = (const float4*)(world_tris + first + min(2u, last));
        const float4* t3 = (const float4*)(world_tris + first + min(3u, last));
        const float4 v[12] = {t0[0], t0[1], t0[2], t1[0], t1[1], t1[2], t2[0], t2[1], t2[2], t3[0], t3[1], t3[2]};
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            mn[0] = fminf(mn[0], v[q].x); mn[1] = fminf(mn[1], v[q].y); mn[2] = fminf(mn[2], v[q].z);
            mx[0] = fmaxf(mx[0], v[q].x); mx[1] = fmaxf(mx[1], v[q].y); mx[2] = fmaxf(mx[2], v[q].z);
        }
    } else if (!empty) {
        const Box6 b = boxes[j.dst + (c - j.src)];
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = b.mn[k]; mx[k] = b.mx[k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { sh.box[g][ci][k] = mn[k]; sh.box[g][ci][3 + k] = mx[k]; }
    __syncthreads();
    float fmn[3], fmx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        fmn[k] = fminf(fminf(sh.box[g][0][k], sh.box[g][1][k]), fminf(sh.box[g][2][k], sh.box[g][3][k]));
        fmx[k] = fmaxf(fmaxf(sh.box[g][0][3 + k], sh.box[g][1][3 + k]), fmaxf(sh.box[g][2][3 + k], sh.box[g][3][3 + k]));
    }
    // step per axis: the power of two just above extent / 254 (so every plane lands in 0 .. 254), exponent byte clamped like the builders'
    uint32_t be[3], qlo = 0, qhi = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ext = fmx[k] - fmn[k];
        be[k] = ext > 0.0f ? min(250u, max(7u, (__float_as_uint(ext * (1.0f / 254.0f)) >> 23) + 1u)) : 7u;
        const float scale = __uint_as_float(be[k] << 23), inv = __uint_as_float((254u - be[k]) << 23);
        int lo = 255, hi = 0;
        if (!empty) {    // round outwards, then check against the decode fma(q, step, origin)
            lo = min(255, max(0, int(floorf((mn[k] - fmn[k]) * inv)))); hi = min(255, max(0, int(ceilf((mx[k] - fmn[k]) * inv))));
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (lo > 0 && fmn[k] + float(lo) * scale > mn[k]) --lo;
                if (hi < 255 && fmn[k] + float(hi) * scale < mx[k]) ++hi;
            }
        }
        qlo |= uint32_t(lo) << (8 * k); qhi |= uint32_t(hi) << (8 * k);
    }
    sh.q[g][ci][0] = qlo; sh.q[g][ci][1] = qhi;
    sh.q[g][ci][2] = empty ? 0xffffffffu : (leaf ? ((c & 0xf0000000u) | (j.tri_base + (c & 0x0fffffffu))) : (j.dst + (c - j.src)));
    __syncthreads();
    if (active && ci == 0u) {
        uint32_t lo4[4], hi4[4], ch[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo4[i] = sh.q[g][i][0]; hi4[i] = sh.q[g][i][1]; ch[i] = sh.q[g][i][2]; }
        auto col = [](const uint32_t* w, int k) { return ((w[0] >> (8 * k)) & 0xffu) | (((w[1] >> (8 * k)) & 0xffu) << 8) | (((w[2] >> (8 * k)) & 0xffu) << 16) | (((w[3] >> (8 * k)) & 0xffu) << 24); };
        uint4* dst = (uint4*)(nodes + j.dst + rel);
        dst[0] = make_uint4(__float_as_uint(fmn[0]), __float_as_uint(fmn[1]), __float_as_uint(fmn[2]), be[0] | (be[1] << 8) | (be[2] << 16) | (children << 24));
        dst[1] = make_uint4(ch[0], ch[1], ch[2], ch[3]);
        dst[2] = make_uint4(col(lo4, 0), col(lo4, 1), col(lo4, 2), col(hi4, 0));
        dst[3] = make_uint4(col(hi4, 1), col(hi4, 2), 0u, 0u);
        Box6 f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { f.mn[k] = fmn[k]; f.mx[k] = fmx[k]; }
        boxes[j.dst + rel] = f;
    }
    __syncthreads();     // the next pass reuses the LDS slots
}
// one height of every job that has that many populous heights
__global__ void __launch_bounds__(256) k_instance_refit_height(const Bvh4Node* __restrict__ blas, const uint2* __restrict__ steps, const BvhTri* __restrict__ world_tris,
                                                                const InstanceRefitJob* __restrict__ jobs, Bvh4Node* __restrict__ nodes, Box6* __restrict__ boxes, uint32_t h) {
    __shared__ RefitShared<256> sh;
    const InstanceRefitJob j = jobs[blockIdx.y];
    if (h >= j.wide_heights) return;
    const uint32_t begin = steps[j.heights + h].x, end = steps[j.heights + h].y;
    for (uint32_t first = begin + blockIdx.x * 64u; first < end; first += gridDim.x * 64u) refit_pass<256>(j, first, end, blas, world_tris, nodes, boxes, sh);
}
// the remaining heights of a job, one workgroup
__global__ void __launch_bounds__(KJ_REFIT_TOP_NODES) k_instance_refit_top(const Bvh4Node* __restrict__ blas, const uint2* __restrict__ steps, const BvhTri* __restrict__ world_tris,
                                                                            const InstanceRefitJob* __restrict__ jobs, Bvh4Node* __restrict__ nodes, Box6* __restrict__ boxes) {
    __shared__ RefitShared<KJ_REFIT_TOP_NODES> sh;
    __shared__ uint2 staged_steps[128];     // keep the per-step lookups out of the chain of dependent round trips
    const InstanceRefitJob j = jobs[blockIdx.x];
    const bool staged = j.height_count <= 128u;
    if (staged) {
        for (uint32_t t = threadIdx.x; t < j.height_count; t += KJ_REFIT_TOP_NODES) staged_steps[t] = steps[j.heights + t];
        __syncthreads();
    }
    for (uint32_t h = j.wide_heights; h < j.height_count; ++h) {
        const uint2 range = staged ? staged_steps[h] : steps[j.heights + h];
        const uint32_t begin = range.x, end = range.y;
        // (refit_pass ends on a barrier: this height's boxes are visible to the workgroup -- one CU, one L1 -- before the next height reads them)
        for (uint32_t first = begin; first < end; first += KJ_REFIT_TOP_NODES / 4) refit_pass<KJ_REFIT_TOP_NODES>(j, first, end, blas, world_tris, nodes, boxes, sh);
    }
}

namespace kj {

hipError_t launch_instance_refit(const Bvh4Node* blas_nodes, const uint2* steps, const BvhTri* world_tris, const InstanceRefitJob* jobs, uint32_t job_count,
                                 uint32_t max_wide_heights, Bvh4Node* nodes, void* boxes, hipStream_t s) {
    if (job_count == 0) return hipSuccess;      // (and leave a sticky error of some earlier, unrelated call where it is)
    // blocks along x per job: enough for the populous heights of a large mesh, fewer when there are thousands of jobs (a grid of
    // 256 x 32768 mostly idle workgroups per height is seconds of launch overhead)
    const uint32_t gx = job_count <= 64u ? 256u : (job_count <= 1024u ? 64u : 16u);
    for (uint32_t j0 = 0; j0 < job_count; j0 += 32768u) {      // (grid.y is limited to 65535)
        const uint32_t n = std::min(32768u, job_count - j0);
        for (uint32_t h = 0; h < max_wide_heights; ++h)
            hipLaunchKernelGGL(k_instance_refit_height, dim3(gx, n), dim3(256), 0, s, blas_nodes, steps, world_tris, jobs + j0, nodes, (Box6*)boxes, h);
        hipLaunchKernelGGL(k_instance_refit_top, dim3(n), dim3(KJ_REFIT_TOP_NODES), 0, s, blas_nodes, steps, world_tris, jobs + j0, nodes, (Box6*)boxes);
        const hipError_t e = hipGetLastError();      // per chunk: a failed launch is reported where it happened
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_instance_triangles(const BvhTri* obj_tris, BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s) {
    if (job_count == 0) return hipSuccess;
    const uint32_t gx = job_count <= 64u ? 64u : (job_count <= 1024u ? 16u : 4u);
    for (uint32_t j0 = 0; j0 < job_count; j0 += 32768u) {      // (grid.y is limited to 65535)
        hipLaunchKernelGGL(k_instance_triangles, dim3(gx, std::min(32768u, job_count - j0)), dim3(256), 0, s, obj_tris, world_tris, jobs + j0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_instance_renumber(BvhTri* world_tris, const InstanceTriJob* jobs, uint32_t job_count, hipStream_t s) {
    if (job_count == 0) return hipSuccess;
    const uint32_t gx = job_count <= 64u ? 64u : (job_count <= 1024u ? 16u : 4u);
    for (uint32_t j0 = 0; j0 < job_count; j0 += 32768u) {
        hipLaunchKernelGGL(k_instance_renumber, dim3(gx, std::min(32768u, job_count - j0)), dim3(256), 0, s, world_tris, jobs + j0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace kj
