// KjDevice + frame prologue kernels for gfx950:
//   BRDF FG LUT (lut/brdf_fg.hlsl), sky cube + convolution (sky/comp_cube.hlsl,
//   convolve_cube.hlsl), primary-ray G-buffer stand-in (raster_simple_ps.hlsl:126-137
//   packing), reprojection map (calculate_reprojection_map.hlsl:17-142) and the
//   raw ray-query entry points (inc/rt.hlsl:58-137).
#include "kj_host.hpp"
#include "kj_scene.hpp"
#include "kj_screen.hpp"
#include <cstdlib>
#include <algorithm>

using namespace kj;

namespace kj { SceneView scene_view(const KjScene& s); }

// ------------------------------------------------------------------ LUT / sun / sky
__global__ void k_brdf_fg_lut(uint2* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= 64 || y >= 64) return;
    const float ndotv = (float(x) / 63.0f) * (1.0f - 1e-3f) + 1e-3f;
    const float roughness = fmaxf(1e-5f, float(y) / 63.0f);
    out[y * 64 + x] = pack_rgba16f(v4(integrate_brdf_fg(roughness, ndotv), 0.0f));
}
// kj_frame_begin: store the frame constants (passed by value in the kernarg segment) into their ring slot, hoist SUN_COLOR and
// derive the matrix products the screen-space passes use per tap (kj_screen.hpp: FrameDerived), in double, rounded once.
__global__ void __launch_bounds__(64) k_frame_begin(const KjFrameConstants fc, FrameBlock* __restrict__ dst, float4* __restrict__ sun_out) {
    const uint32_t* src = (const uint32_t*)&fc;
    uint32_t* d = (uint32_t*)&dst->fc;
    for (uint32_t i = threadIdx.x; i < sizeof(KjFrameConstants) / 4; i += 64) d[i] = src[i];
    const KjViewConstants& vc = fc.view_constants;
    if (threadIdx.x < 48) {
        const uint32_t which = threadIdx.x >> 4, e = threadIdx.x & 15u, c = e >> 2, r = e & 3u;
        const float* A = which == 0 ? vc.view_to_world : (which == 1 ? vc.view_to_clip : vc.view_to_sample);
        const float* B = which == 0 ? vc.sample_to_view : vc.world_to_view;
        double acc = 0.0;
        for (uint32_t k = 0; k < 4; ++k) acc += double(A[k * 4 + r]) * double(B[c * 4 + k]);
        (which == 0 ? dst->fd.sample_to_world : (which == 1 ? dst->fd.world_to_clip : dst->fd.world_to_sample))[e] = float(acc);
    }
    if (threadIdx.x == 0) {
        const V3 c = sun_color_in_direction(fc, sun_direction(fc));
        *sun_out = make_float4(c.x, c.y, c.z, 0.0f);
        const V3 eye = get_eye_position(fc);
        dst->fd.eye_ws[0] = eye.x; dst->fd.eye_ws[1] = eye.y; dst->fd.eye_ws[2] = eye.z; dst->fd.eye_ws[3] = 1.0f;
    }
}
__global__ void k_sky_cube(const FrameConstants* __restrict__ fc, uint2* __restrict__ out, int width) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, face = blockIdx.z;
    if (x >= width || y >= width) return;
    const V3 dir = cube_face_dir(face, V2{(x + 0.5f) / float(width), (y + 0.5f) / float(width)});
    const V3 c = atmosphere_default(*fc, dir, sun_direction(*fc));
    out[(size_t(face) * width + y) * width + x] = pack_rgba16f(v4(c, 1.0f));
}
// one wave per output texel: 512 cone samples, 8 per lane, wave-reduced
__global__ void __launch_bounds__(64) k_convolve_cube(const uint2* __restrict__ in, int in_width, uint2* __restrict__ out, int width) {
    const int texel = blockIdx.x;
    const int face = texel / (width * width), y = (texel / width) % width, x = texel % width;
    const V3 output_dir = cube_face_dir(face, V2{(x + 0.5f) / float(width), (y + 0.5f) / float(width)});
    const Basis basis = build_orthonormal_basis(output_dir);
    V4 acc = v4(0.0f);
    for (uint32_t i = threadIdx.x; i < 512u; i += 64u) {
        const V3 input_dir = to_world(basis, uniform_sample_cone(hammersley(i, 512u), 0.99f));
        acc += sample_cube_rgba16f(in, in_width, input_dir);
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    }
    if (threadIdx.x == 0) out[texel] = pack_rgba16f(acc / 512.0f);
}

// ------------------------------------------------------------------ G-buffer stand-in
__global__ void __launch_bounds__(64) k_raster_gbuffer(const FrameConstants* __restrict__ fcp, SceneView sc, int W, int H, uint32_t* __restrict__ geometric_normal,
                                                        uint4* __restrict__ gbuffer, float* __restrict__ depth, uint2* __restrict__ velocity) {
    extern __shared__ uint32_t lds_stack[];
    const uint2 tb = tile_order<KJ_TILES_PLAIN>();
    const int x = int(tb.x) * 8 + int(threadIdx.x & 7), y = int(tb.y) * 8 + int(threadIdx.x >> 3);
    if (x >= W || y >= H) return;
    const FrameConstants& fc = *fcp;
    const V2 uv = get_uv(float(x), float(y), tex_size4(W, H));
    const ViewRay vr = view_ray_from_uv(fc, uv);
    const RayHit h = bvh_trace<false>(sc.bvh, vr.origin_ws, vr.dir_ws, 0.0f, FLT_MAX, false, lds_stack + threadIdx.x, 64);
    const size_t idx = size_t(y) * W + x;
    if (h.slot == 0xffffffffu) {
        geometric_normal[idx] = 0; gbuffer[idx] = make_uint4(0, 0, 0, 0); depth[idx] = 0.0f; velocity[idx] = make_uint2(0, 0);
        return;
    }
    const float4* __restrict__ tp = (const float4*)sc.bvh.tris + size_t(h.slot) * 3;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    V3 gn_ws = normalize(cross(V3{b.x - a.x, b.y - a.y, b.z - a.z}, V3{c.x - a.x, c.y - a.y, c.z - a.z}));
    if (dot(gn_ws, vr.dir_ws) > 0) gn_ws = -gn_ws;
    const V3 gn_vs = normalize(direction_world_to_view(fc, gn_ws));
    const V3 pos = mad_nc(vr.origin_ws, vr.dir_ws, h.t);
    const V3 cs = position_world_to_sample(fc, pos);
    geometric_normal[idx] = pack_a2r10g10b10(gn_vs * 0.5f + 0.5f);
    // the raster pass samples with implicit derivatives; this stand-in uses the pixel's ray cone (width 0 at the eye)
    gbuffer[idx] = shade_gbuffer_hit(sc, fc, vr.dir_ws, h, 0, pixel_ray_cone_from_image_height(fc, float(H)).width_at_t(h.t));
    depth[idx] = cs.z;
    velocity[idx] = make_uint2(0, 0);
}

// ------------------------------------------------------------------ reprojection map
__global__ void __launch_bounds__(64) k_reprojection_map(const FrameConstants* __restrict__ fcp, int W, int H, const float* __restrict__ depth_p,
                                                          const uint32_t* __restrict__ gn_p, const float* __restrict__ prev_depth_p,
                                                          const uint2* __restrict__ velocity_p, uint2* __restrict__ out_p) {
    const uint2 tb = tile_order<KJ_TILES_PLAIN>();
    const int x = int(tb.x) * 8 + int(threadIdx.x & 7), y = int(tb.y) * 8 + int(threadIdx.x >> 3);
    if (x >= W || y >= H) return;
    const FrameConstants& fc = *fcp;
    const KjViewConstants& vc = fc.view_constants;
    const size_t idx = size_t(y) * W + x;
    const V2 uv = get_uv(float(x), float(y), tex_size4(W, H));
    auto store = [&](V4 v) {
        out_p[idx] = make_uint2(uint32_t(uint16_t(to_snorm16(v.x))) | (uint32_t(uint16_t(to_snorm16(v.y))) << 16),
                                uint32_t(uint16_t(to_snorm16(v.z))) | (uint32_t(uint16_t(to_snorm16(v.w))) << 16));
    };
    const float depth = depth_p[idx];
    const V2 cs = uv_to_cs(uv);
    if (depth == 0.0f) {
        const V4 pos_vs = mul44(vc.clip_to_view, V4{cs.x, cs.y, 0.0f, 1.0f});
        const V4 prev_pcs = mul44(vc.clip_to_prev_clip, mul44(vc.view_to_clip, pos_vs));
        const V2 uv_diff = cs_to_uv(V2{prev_pcs.x, prev_pcs.y}) - uv;
        store(V4{uv_diff.x, uv_diff.y, 0, 0});
        return;
    }
    const V3 normal_vs = unpack_a2r10g10b10(gn_p[idx]) * 2.0f - 1.0f;
    const V3 normal_pvs = xyz(mul44(vc.prev_clip_to_prev_view, mul44(vc.clip_to_prev_clip, mul44(vc.view_to_clip, v4(normal_vs, 0)))));
    const V4 pos_vs = mul44(vc.clip_to_view, V4{cs.x, cs.y, depth, 1.0f});
    const float dist_to_point = -(pos_vs.z / pos_vs.w);
    V4 prev_vs = pos_vs / pos_vs.w;
    const V4 vel = unpack_rgba16f(velocity_p[idx]);
    prev_vs.x += vel.x; prev_vs.y += vel.y; prev_vs.z += vel.z;
    const V4 prev_pcs = mul44(vc.clip_to_prev_clip, mul44(vc.view_to_clip, prev_vs));
    V2 prev_uv = cs_to_uv(V2{prev_pcs.x / prev_pcs.w, prev_pcs.y / prev_pcs.w});
    V2 uv_diff = prev_uv - uv;
    uv_diff = V2{floorf(uv_diff.x * 32767.0f + 0.5f) / 32767.0f, floorf(uv_diff.y * 32767.0f + 0.5f) / 32767.0f};
    prev_uv = uv + uv_diff;
    V4 prev_pvs = mul44(vc.prev_clip_to_prev_view, prev_pcs);
    prev_pvs = prev_pvs / prev_pvs.w;
    const float plane_dist_prev_dz = fminf(-0.2f, normal_vs.z);
    const V2 bp = prev_uv * V2{float(W), float(H)} - 0.5f;
    const int ox = int(truncf(bp.x)), oy = int(truncf(bp.y));
    const Img<float> pdimg = img<float>(prev_depth_p, W, H);
    const V4 prev_depth{pdimg.ldc(ox, oy), pdimg.ldc(ox + 1, oy), pdimg.ldc(ox, oy + 1), pdimg.ldc(ox + 1, oy + 1)};
    const float m43 = -vc.prev_clip_to_prev_view[11];
    const V4 pvz{1.0f / (prev_depth.x * m43), 1.0f / (prev_depth.y * m43), 1.0f / (prev_depth.z * m43), 1.0f / (prev_depth.w * m43)};
    const V4 qd{fabsf(plane_dist_prev_dz * (pvz.x - prev_pvs.z)), fabsf(plane_dist_prev_dz * (pvz.y - prev_pvs.z)),
                fabsf(plane_dist_prev_dz * (pvz.z - prev_pvs.z)), fabsf(plane_dist_prev_dz * (pvz.w - prev_pvs.z))};
    const float acceptance_threshold = 0.001f * (1080.0f / float(H));
    const V3 pos_vs_norm = normalize(xyz(pos_vs) / pos_vs.w);
    const float ndotv = dot(normal_vs, pos_vs_norm);
    const float prev_ndotv = dot(normal_pvs, normalize(xyz(prev_pvs)));
    const float thr = acceptance_threshold * dist_to_point / -ndotv;
    V4 qv{stepf(qd.x, thr), stepf(qd.y, thr), stepf(qd.z, thr), stepf(qd.w, thr)};
    qv.x *= pdimg.inb(ox, oy) ? 1.0f : 0.0f;
    qv.y *= pdimg.inb(ox + 1, oy) ? 1.0f : 0.0f;
    qv.z *= pdimg.inb(ox, oy + 1) ? 1.0f : 0.0f;
    qv.w *= pdimg.inb(ox + 1, oy + 1) ? 1.0f : 0.0f;
    const float validity = dot(qv, V4{1, 2, 4, 8}) / 15.0f;
    float accuracy = smoothstep(0.8f, 0.95f, prev_ndotv / ndotv);
    if (saturate(prev_uv.x) != prev_uv.x || saturate(prev_uv.y) != prev_uv.y) accuracy = -1;
    store(V4{uv_diff.x, uv_diff.y, validity, accuracy});
}

// ------------------------------------------------------------------ raw ray queries
// Ray streams (kj_bvh.hpp: bvh_trace_stream): persistent waves, lanes refill as rays finish. KJ_TRACE_PER_RAY=1 selects the
// one-ray-per-lane kernels (A/B measurements, scripts/traversal_microbench.py).
__global__ void __launch_bounds__(64) k_trace_closest(SceneView sc, const float4* __restrict__ rays, float4* __restrict__ hits, uint32_t count, int cull_back) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const float4 a = rays[i * 2], b = rays[i * 2 + 1];
    const RayHit h = bvh_trace<false>(sc.bvh, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, a.w, b.w, cull_back != 0, lds_stack + threadIdx.x, 64);
    hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.world_id));
}
// four lanes per ray (kj_bvh.hpp: bvh_trace_quad): what small batches use -- 16 rays per wave, four times the waves, shorter steps
__global__ void __launch_bounds__(64) k_trace_closest_quad(SceneView sc, const float4* __restrict__ rays, float4* __restrict__ hits, uint32_t count, int cull_back) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * 16 + (threadIdx.x >> 2);
    const bool active = i < count;
    const float4 a = active ? rays[i * 2] : make_float4(0, 0, 0, 0), b = active ? rays[i * 2 + 1] : make_float4(0, 0, 1, 0);
    const RayHit h = bvh_trace_quad<false>(sc.bvh, active, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, a.w, b.w, cull_back != 0, lds_stack + (threadIdx.x >> 2), 16);
    if (active && (threadIdx.x & 3u) == 0u) hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.world_id));
}
__global__ void __launch_bounds__(64) k_trace_any_quad(SceneView sc, const float4* __restrict__ rays, uint8_t* __restrict__ out, uint32_t count) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * 16 + (threadIdx.x >> 2);
    const bool active = i < count;
    const float4 a = active ? rays[i * 2] : make_float4(0, 0, 0, 0), b = active ? rays[i * 2 + 1] : make_float4(0, 0, 1, 0);
    const RayHit h = bvh_trace_quad<true>(sc.bvh, active, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, a.w, b.w, false, lds_stack + (threadIdx.x >> 2), 16);
    if (active && (threadIdx.x & 3u) == 0u) out[i] = h.slot != 0xffffffffu ? 1 : 0;
}
__global__ void __launch_bounds__(64) k_trace_any(SceneView sc, const float4* __restrict__ rays, uint8_t* __restrict__ out, uint32_t count) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const float4 a = rays[i * 2], b = rays[i * 2 + 1];
    out[i] = bvh_trace<true>(sc.bvh, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, a.w, b.w, false, lds_stack + threadIdx.x, 64).slot != 0xffffffffu ? 1 : 0;
}
__global__ void __launch_bounds__(64) k_trace_closest_stream(SceneView sc, const float4* __restrict__ rays, float4* __restrict__ hits, uint32_t count, int cull_back, StreamTune tune) {
    extern __shared__ uint32_t lds_stack[];
    bvh_trace_stream<false>(sc.bvh, rays, count, cull_back != 0, blockIdx.x, gridDim.x, lds_stack + threadIdx.x, 64,
                            [&](uint32_t i, const RayHit& h) { hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.world_id)); }, tune);
}
__global__ void __launch_bounds__(64) k_trace_any_stream(SceneView sc, const float4* __restrict__ rays, uint8_t* __restrict__ out, uint32_t count, StreamTune tune) {
    extern __shared__ uint32_t lds_stack[];
    bvh_trace_stream<true>(sc.bvh, rays, count, false, blockIdx.x, gridDim.x, lds_stack + threadIdx.x, 64,
                           [&](uint32_t i, const RayHit& h) { out[i] = h.slot != 0xffffffffu ? 1 : 0; }, tune);
}

// ---- many contiguous copies in one launch (kj_host.hpp: launch_copy_blocks). A workgroup takes one 32 KB chunk of one block; the block is found by a scan of the
// (wave-uniform) chunk prefix. 16-byte moves where source, destination and length allow, bytes otherwise (odd test extents).
#define KJ_COPY_MAX_BLOCKS 48
#define KJ_COPY_CHUNK (32u * 1024u)
struct CopyBatch { kj::CopyBlock blk[KJ_COPY_MAX_BLOCKS]; uint32_t first_chunk[KJ_COPY_MAX_BLOCKS + 1]; uint32_t n; };
__global__ void __launch_bounds__(256) k_copy_blocks(CopyBatch b) {
    uint32_t k = 0;
    while (k + 1u < b.n && blockIdx.x >= b.first_chunk[k + 1u]) ++k;
    const kj::CopyBlock blk = b.blk[k];
    const uint64_t off = uint64_t(blockIdx.x - b.first_chunk[k]) * KJ_COPY_CHUNK;
    const uint32_t len = uint32_t(blk.bytes - off < KJ_COPY_CHUNK ? blk.bytes - off : KJ_COPY_CHUNK);
    const uint8_t* src = (const uint8_t*)blk.src + off;
    uint8_t* dst = (uint8_t*)blk.dst + off;
    if (((uintptr_t(src) | uintptr_t(dst)) & 15u) == 0u) {
        const uint32_t quads = len / 16u;
        for (uint32_t i = threadIdx.x; i < quads; i += 256u) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        for (uint32_t i = quads * 16u + threadIdx.x; i < len; i += 256u) dst[i] = src[i];
    } else
        for (uint32_t i = threadIdx.x; i < len; i += 256u) dst[i] = src[i];
}
namespace kj {
hipError_t launch_copy_blocks(const CopyBlock* blocks, size_t n, hipStream_t s) {
    size_t i = 0;
    while (i < n) {
        CopyBatch b;
        b.n = 0;
        uint32_t chunks = 0;
        for (; i < n && b.n < KJ_COPY_MAX_BLOCKS; ++i) {
            if (blocks[i].bytes == 0) continue;
            b.blk[b.n] = blocks[i]; b.first_chunk[b.n] = chunks;
            chunks += uint32_t((blocks[i].bytes + KJ_COPY_CHUNK - 1) / KJ_COPY_CHUNK);
            ++b.n;
        }
        if (b.n == 0) break;
        b.first_chunk[b.n] = chunks;
        hipLaunchKernelGGL(k_copy_blocks, dim3(chunks), dim3(256), 0, s, b);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace kj
__global__ void __launch_bounds__(256) pmc_calibration_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}

// light_gbuffer.hlsl:60-260 — the deferred combine: sun direct light through the shadow mask, emissive, diffuse
// GI (rtdgi) times albedo and the specular layer's transmission, optional specular (rtr) term, sky + sun disc for depth == 0.
// Debug shading modes 0-4 as in the shader; mode 5 (ircache view) and the wrc overlay are not built.
struct LightGbufferArgs {
    const FrameConstants* __restrict__ fc;
    Img<uint4> gbuffer_tex; Img<float> depth_tex; Img<uint8_t> shadow_mask_tex; Img<uint32_t> shadow_mask_rg16f; Img<uint32_t> rtr_tex; Img<uint2> rtdgi_tex;
    Img<uint2> temporal_output_tex, output_tex;
    const uint2* __restrict__ unconvolved_sky_cube; int sky_width;
    const uint2* __restrict__ brdf_fg_lut;
    const float4* __restrict__ sun_color;
    uint32_t debug_shading_mode;
    int row0, row1;      // rows [row0, row1) of the image
};
__global__ void __launch_bounds__(64) k_light_gbuffer(LightGbufferArgs a) {
    const int lane = threadIdx.x;
    const uint2 tb = tile_order<KJ_TILES_PLAIN>();
    const int x = int(tb.x) * 8 + (lane & 7), y = a.row0 + int(tb.y) * 8 + (lane >> 3);
    const int W = a.output_tex.w, H = a.output_tex.h;
    if (x >= W || y >= a.row1) return;
    const FrameConstants& fc = *a.fc;
    const V4 ots = tex_size4(W, H);
    const V2 uv = get_uv(float(x), float(y), ots);
    const ViewRay vrc = view_ray_from_uv(fc, uv);
    const V3 ray_d = vrc.dir_ws;
    const float depth = a.depth_tex.ld(x, y);
    const V3 sun_dir = sun_direction(fc);
    if (depth == 0.0f) {
        const float real_sun_angular_radius = 0.53f * 0.5f * KJ_PI / 180.0f;
        const float sun_angular_radius_cos = fminf(cosf(real_sun_angular_radius), fc.sun_angular_radius_cos);
        const float current_sun_angular_radius = acosf(sun_angular_radius_cos);
        const float sun_radius_ratio = real_sun_angular_radius / current_sun_angular_radius;
        V3 output = xyz(sample_cube_rgba16f(a.unconvolved_sky_cube, a.sky_width, ray_d));
        if (dot(ray_d, sun_dir) > sun_angular_radius_cos) output += 800.0f * sun_color_in_direction(fc, ray_d) * sun_radius_ratio * sun_radius_ratio;
        st4(a.temporal_output_tex, x, y, v4(output, 1.0f));
        st4(a.output_tex, x, y, v4(output, 1.0f));
        return;
    }
    float shadow_mask = a.shadow_mask_rg16f.p ? ld2h(a.shadow_mask_rg16f, x, y).x : from_unorm8(a.shadow_mask_tex.ld(x, y));
    if (a.debug_shading_mode == 4u) shadow_mask = 1;
    const GbufferData true_gbuffer = gbuffer_unpack(a.gbuffer_tex.ld(x, y));
    GbufferData gbuffer = true_gbuffer;
    if (a.debug_shading_mode == 1u) gbuffer.albedo = v3(0.5f);
    const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
    const V3 wi = to_local(tangent_to_world, sun_dir);
    V3 wo = to_local(tangent_to_world, -ray_d);
    if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
    const LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(a.brdf_fg_lut, gbuffer, wo.z);
    const V3 brdf_value = layered_brdf_evaluate_directional_light(brdf, wo, wi) * fmaxf(0.0f, wi.z);
    const float4 sc4 = *a.sun_color;
    const V3 light_radiance = shadow_mask * V3{sc4.x, sc4.y, sc4.z};
    V3 total_radiance = brdf_value * light_radiance;
    total_radiance += gbuffer.emissive;
    V3 gi_irradiance = v3(0.0f);
    if (a.debug_shading_mode != 4u) gi_irradiance = xyz(ld4(a.rtdgi_tex, x, y));
    total_radiance += gi_irradiance * brdf.diff_albedo * brdf.preintegrated_transmission_fraction;
    const V3 rtr = a.rtr_tex.p ? unpack_r11g11b10f(a.rtr_tex.ld(x, y)) : v3(0.0f);
    if (a.debug_shading_mode != 4u) {
        V3 rtr_radiance = rtr * brdf.preintegrated_reflection;
        if (a.debug_shading_mode == 1u) {
            const LayeredBrdf true_brdf = layered_brdf_from_gbuffer_ndotv(a.brdf_fg_lut, true_gbuffer, wo.z);
            rtr_radiance = rtr_radiance / true_brdf.preintegrated_reflection;
        }
        total_radiance += rtr_radiance;
    }
    st4(a.temporal_output_tex, x, y, v4(total_radiance, 1.0f));
    V3 output = total_radiance;
    if (a.debug_shading_mode == 3u) {
        const LayeredBrdf true_brdf = layered_brdf_from_gbuffer_ndotv(a.brdf_fg_lut, true_gbuffer, wo.z);
        output = rtr * brdf.preintegrated_reflection / true_brdf.preintegrated_reflection;
    }
    if (a.debug_shading_mode == 2u) output = gi_irradiance;
    st4(a.output_tex, x, y, v4(output, 1.0f));
}

// rt/trace_sun_shadow_mask.rgen.hlsl:19-60 (USE_SOFT_SHADOWS 1): one shadow ray per full-res pixel, R8_UNORM mask
__global__ void __launch_bounds__(64) k_sun_shadow_mask(const FrameConstants* __restrict__ fcp, SceneView sc, const uint32_t* __restrict__ blue_noise, Img<float> depth_tex,
                                                         Img<uint32_t> geometric_normal_tex, Img<uint8_t> output_tex, unsigned long long* __restrict__ ray_counter, int row0, int row1) {
    extern __shared__ uint32_t lds_stack[];
    const int lane = threadIdx.x;
    const uint2 tb = tile_order<KJ_TILES_PLAIN>();
    const int x = int(tb.x) * 8 + (lane & 7), y = row0 + int(tb.y) * 8 + (lane >> 3);      // rows [row0, row1): the launch covers just those tiles
    if (x >= output_tex.w || y >= row1) return;
    const FrameConstants& fc = *fcp;
    const V2 uv{(float(x) + 0.5f) / float(output_tex.w), (float(y) + 0.5f) / float(output_tex.h)};
    const float z_over_w = depth_tex.ld(x, y);
    if (0.0f == z_over_w) { output_tex.st(x, y, 255); return; }
    const V2 cs = uv_to_cs(uv);
    V4 pt_vs = mul44(fc.view_constants.sample_to_view, V4{cs.x, cs.y, z_over_w, 1.0f});
    V4 pt_ws = mul44(fc.view_constants.view_to_world, pt_vs);
    pt_ws = pt_ws / pt_ws.w;
    pt_vs = pt_vs / pt_vs.w;
    const V3 normal_vs = unpack_a2r10g10b10(geometric_normal_tex.ld(x, y)) * 2.0f - 1.0f;
    const V3 normal_ws = xyz(mul44(fc.view_constants.view_to_world, v4(normal_vs, 0.0f)));
    const float bias_amount = (-pt_vs.z + length(xyz(pt_ws))) * 1e-5f;
    const V3 ray_origin = xyz(pt_ws) + normal_ws * bias_amount;
    const V4 bn = blue_noise_for_pixel(blue_noise, uint32_t(x), uint32_t(y), fc.frame_index);
    const V3 dir = sample_sun_direction(fc, V2{bn.x, bn.y}, true);
    const bool is_shadowed = rt_is_shadowed<false>(sc, ray_origin, dir, 0.0f, FLT_MAX, lds_stack + lane, 64);
    output_tex.st(x, y, is_shadowed ? 0 : 255);
    if (ray_counter) {
        const unsigned long long m = __ballot(true);
        if ((__ffsll((long long)m) - 1) == int(__lane_id())) atomicAdd(ray_counter, (unsigned long long)__popcll(m));
    }
}

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

// light_gbuffer (renderers/deferred.rs:6-60; shaders/light_gbuffer.hlsl): shadow_mask R8_UNORM (raw) or RG16F (.x, denoised), rtr_tex B10G11R11_UFLOAT (kj_rtr_filter_temporal's output) or NULL (= black),
// rtdgi_tex RGBA16F, unconvolved_sky_cube 6 x w x w RGBA16F; outputs RGBA16F.
KjStatus kj_light_gbuffer(KjDevice* dev, const KjGbufferDepth* gd, const void* shadow_mask, uint32_t shadow_mask_is_rg16f, const void* rtr_tex, const void* rtdgi_tex,
                          const void* unconvolved_sky_cube, uint32_t sky_cube_width, void* out_temporal, void* out, uint32_t debug_shading_mode, void* stream) {
    KJ_REQUIRE(gd, "null argument");
    return kj_light_gbuffer_rows(dev, gd, shadow_mask, shadow_mask_is_rg16f, rtr_tex, rtdgi_tex, unconvolved_sky_cube, sky_cube_width, out_temporal, out, debug_shading_mode, 0u, gd->height, stream);
}
// rows [row_begin, row_end) of the combine (every input is read at the pixel itself: a strip needs nothing from outside it)
KjStatus kj_light_gbuffer_rows(KjDevice* dev, const KjGbufferDepth* gd, const void* shadow_mask, uint32_t shadow_mask_is_rg16f, const void* rtr_tex, const void* rtdgi_tex,
                               const void* unconvolved_sky_cube, uint32_t sky_cube_width, void* out_temporal, void* out, uint32_t debug_shading_mode,
                               uint32_t row_begin, uint32_t row_end, void* stream) {
    KJ_REQUIRE(dev && gd && gd->gbuffer && gd->depth && shadow_mask && rtdgi_tex && unconvolved_sky_cube && out_temporal && out && gd->width && gd->height, "null argument");
    KJ_REQUIRE(row_begin < row_end && row_end <= gd->height && (row_begin % 8u) == 0u, "rows must be a non-empty range starting on an 8-row boundary");
    KJ_REQUIRE(dev->fc_dev, "kj_frame_begin not called");
    if (debug_shading_mode > 4) { set_last_error("debug_shading_mode %u (ircache view) is not built", debug_shading_mode); return KJ_ERR_UNSUPPORTED; }
    const int W = int(gd->width), H = int(gd->height);
    LightGbufferArgs a;
    a.fc = dev->fc_dev;
    a.gbuffer_tex = img<uint4>(gd->gbuffer, W, H); a.depth_tex = img<float>(gd->depth, W, H); a.shadow_mask_tex = img<uint8_t>(shadow_mask_is_rg16f ? nullptr : shadow_mask, W, H);
    a.shadow_mask_rg16f = img<uint32_t>(shadow_mask_is_rg16f ? shadow_mask : nullptr, W, H);
    a.rtr_tex = img<uint32_t>(rtr_tex, W, H); a.rtdgi_tex = img<uint2>(rtdgi_tex, W, H);
    a.temporal_output_tex = img<uint2>(out_temporal, W, H); a.output_tex = img<uint2>(out, W, H);
    a.unconvolved_sky_cube = (const uint2*)unconvolved_sky_cube; a.sky_width = int(sky_cube_width);
    a.brdf_fg_lut = (const uint2*)dev->brdf_fg_lut.p;
    a.sun_color = (const float4*)dev->sun_color.p + dev->fc_slot;
    a.debug_shading_mode = debug_shading_mode;
    a.row0 = int(row_begin); a.row1 = int(row_end);
    hipLaunchKernelGGL(k_light_gbuffer, dim3((W + 7) / 8, (row_end - row_begin + 7) / 8), dim3(64), 0, (hipStream_t)stream, a);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// trace_sun_shadow_mask(rg, &GbufferDepth, tlas, bindless_set) -> Handle<Image> (renderers/shadows.rs:10-40)
KjStatus kj_trace_sun_shadow_mask(KjDevice* dev, KjScene* scene, const KjGbufferDepth* gd, void* out_mask_r8, uint64_t* ray_counter_dev, void* stream) {
    KJ_REQUIRE(gd, "null argument");
    return kj_trace_sun_shadow_mask_rows(dev, scene, gd, out_mask_r8, 0u, gd->height, ray_counter_dev, stream);
}
// rows [row_begin, row_end) of the mask: one ray per pixel of the strip, nothing read from outside it
KjStatus kj_trace_sun_shadow_mask_rows(KjDevice* dev, KjScene* scene, const KjGbufferDepth* gd, void* out_mask_r8, uint32_t row_begin, uint32_t row_end, uint64_t* ray_counter_dev, void* stream) {
    KJ_REQUIRE(dev && scene && gd && gd->depth && gd->geometric_normal && out_mask_r8 && gd->width && gd->height, "null argument");
    KJ_REQUIRE(row_begin < row_end && row_end <= gd->height && (row_begin % 8u) == 0u, "rows must be a non-empty range starting on an 8-row boundary");
    KJ_REQUIRE(dev->fc_dev, "kj_frame_begin not called");
    if (!scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    const SceneView sv = scene_view(*scene);
    const int W = int(gd->width), H = int(gd->height);
    hipLaunchKernelGGL(k_sun_shadow_mask, dim3((W + 7) / 8, (row_end - row_begin + 7) / 8), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, dev->fc_dev, sv,
                       (const uint32_t*)dev->blue_noise.p, img<float>(gd->depth, W, H), img<uint32_t>(gd->geometric_normal, W, H), img<uint8_t>(out_mask_r8, W, H),
                       (unsigned long long*)ray_counter_dev, int(row_begin), int(row_end));
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

KjStatus kj_device_create(int32_t ordinal, const uint8_t* blue_noise_rgba8_256, KjDevice** out) {
    KJ_REQUIRE(out && blue_noise_rgba8_256, "null argument");
    int n = 0;
    KJ_TRY_HIP(hipGetDeviceCount(&n));
    if (ordinal < 0 || ordinal >= n) { set_last_error("HIP device %d not available (%d devices)", ordinal, n); return KJ_ERR_HIP; }
    KJ_TRY_HIP(hipSetDevice(ordinal));
    KjDevice* d = new KjDevice();
    d->ordinal = ordinal;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) == hipSuccess) d->num_cus = uint32_t(prop.multiProcessorCount);
    KjStatus st = KJ_OK;
    do {
        if (d->blue_noise.upload(blue_noise_rgba8_256, 256 * 256 * 4) != hipSuccess) { st = KJ_ERR_OUT_OF_MEMORY; break; }
        if (d->brdf_fg_lut.alloc(64 * 64 * 8) != hipSuccess) { st = KJ_ERR_OUT_OF_MEMORY; break; }
        if (d->frame_constants.alloc(sizeof(FrameBlock) * KjDevice::FC_RING) != hipSuccess) { st = KJ_ERR_OUT_OF_MEMORY; break; }
        if (d->sun_color.alloc(16 * KjDevice::FC_RING) != hipSuccess) { st = KJ_ERR_OUT_OF_MEMORY; break; }
        hipLaunchKernelGGL(k_brdf_fg_lut, dim3(8, 8), dim3(8, 8), 0, 0, (uint2*)d->brdf_fg_lut.p);
        if (hipDeviceSynchronize() != hipSuccess) { st = KJ_ERR_HIP; break; }
    } while (0);
    if (st != KJ_OK) { set_last_error("kj_device_create failed: %s", hipGetErrorString(hipGetLastError())); delete d; return st; }
    *out = d;
    return KJ_OK;
}
void kj_device_destroy(KjDevice* dev) { delete dev; }
KjStatus kj_device_brdf_lut(KjDevice* dev, const void** out) {
    KJ_REQUIRE(dev && out, "null argument");
    *out = dev->brdf_fg_lut.p;
    return KJ_OK;
}

KjStatus kj_frame_begin(KjDevice* dev, const KjFrameConstants* fc, void* stream_) {
    KJ_REQUIRE(dev && fc, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    dev->fc_slot = (dev->fc_slot + 1) % KjDevice::FC_RING;
    dev->fc_host = *fc;
    FrameBlock* dst = (FrameBlock*)dev->frame_constants.p + dev->fc_slot;
    // The 1216-byte block travels as a kernel argument: no pageable-memory staging copy, fully asynchronous.
    dev->fc_dev = &dst->fc;
    hipLaunchKernelGGL(k_frame_begin, dim3(1), dim3(64), 0, stream, *fc, dst, (float4*)dev->sun_color.p + dev->fc_slot);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

KjStatus kj_sky_cube_render(KjDevice* dev, void* out_cube64, void* stream) {
    KJ_REQUIRE(dev && out_cube64 && dev->fc_dev, "null argument / kj_frame_begin not called");
    hipLaunchKernelGGL(k_sky_cube, dim3(8, 8, 6), dim3(8, 8), 0, (hipStream_t)stream, dev->fc_dev, (uint2*)out_cube64, 64);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
KjStatus kj_sky_cube_convolve(KjDevice* dev, const void* cube64, void* out_cube16, void* stream) {
    KJ_REQUIRE(dev && cube64 && out_cube16, "null argument");
    hipLaunchKernelGGL(k_convolve_cube, dim3(6 * 16 * 16), dim3(64), 0, (hipStream_t)stream, (const uint2*)cube64, 64, (uint2*)out_cube16, 16);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

KjStatus kj_raster_gbuffer(KjDevice* dev, KjScene* scene, uint32_t W, uint32_t H, void* geometric_normal, void* gbuffer, void* depth, void* velocity, void* stream) {
    KJ_REQUIRE(dev && scene && geometric_normal && gbuffer && depth && velocity && dev->fc_dev, "null argument / kj_frame_begin not called");
    if (!scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    const SceneView sv = scene_view(*scene);
    hipLaunchKernelGGL(k_raster_gbuffer, dim3((W + 7) / 8, (H + 7) / 8), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, dev->fc_dev, sv, int(W), int(H),
                       (uint32_t*)geometric_normal, (uint4*)gbuffer, (float*)depth, (uint2*)velocity);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// waves + scheduling knobs of a ray-stream launch: enough waves to fill the chip (KJ_STREAM_WAVES_PER_CU per CU, default 24); knobs overridable for measurements
static StreamTune stream_launch(const KjDevice* dev, uint32_t count, uint32_t* waves) {
    const uint32_t per_cu = kj_debug_getenv("KJ_STREAM_WAVES_PER_CU") ? uint32_t(atoi(kj_debug_getenv("KJ_STREAM_WAVES_PER_CU"))) : 24u;
    StreamTune t = stream_tune_for(count, dev->num_cus * per_cu, waves);
    if (const char* v = kj_debug_getenv("KJ_STREAM_REFILL")) t.refill_threshold = uint32_t(atoi(v));
    if (const char* v = kj_debug_getenv("KJ_STREAM_NODE_WEIGHT")) t.node_weight = uint32_t(atoi(v));
    if (const char* v = kj_debug_getenv("KJ_STREAM_TRI_WEIGHT")) t.tri_weight = uint32_t(atoi(v));
    return t;
}
// Batches too small to fill the chip with one ray per lane (fewer rays than ~4 waves per SIMD would hold) walk with four lanes per ray
// (kj_bvh.hpp: bvh_trace_quad). KJ_TRACE_QUAD_MAX_RAYS overrides the threshold (0 = never).
static uint32_t quad_max_rays(const KjDevice* dev) {
    const long env = kj_debug_getenv("KJ_TRACE_QUAD_MAX_RAYS") ? atol(kj_debug_getenv("KJ_TRACE_QUAD_MAX_RAYS")) : -1;
    return env >= 0 ? uint32_t(env) : dev->num_cus * 4u * 4u * 16u;      // 4 SIMDs x 4 waves x 16 rays per CU: 65536 on MI355X
}
KjStatus kj_trace_closest(KjScene* scene, const void* rays, void* hits, uint32_t count, uint32_t cull_back_faces, void* stream) {
    KJ_REQUIRE(scene && rays && hits, "null argument");
    if (!scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    if (count == 0) return KJ_OK;
    const SceneView sv = scene_view(*scene);
    if (count <= quad_max_rays(scene->dev))
        hipLaunchKernelGGL(k_trace_closest_quad, dim3((count + 15) / 16), dim3(64), quad_stack_bytes(), (hipStream_t)stream, sv, (const float4*)rays, (float4*)hits, count, int(cull_back_faces));
    else if (kj_debug_getenv("KJ_TRACE_PER_RAY"))
        hipLaunchKernelGGL(k_trace_closest, dim3((count + 63) / 64), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, sv, (const float4*)rays, (float4*)hits, count, int(cull_back_faces));
    else {
        uint32_t waves; const StreamTune tune = stream_launch(scene->dev, count, &waves);
        hipLaunchKernelGGL(k_trace_closest_stream, dim3(waves), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, sv, (const float4*)rays, (float4*)hits, count, int(cull_back_faces), tune);
    }
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
KjStatus kj_trace_any(KjScene* scene, const void* rays, void* out_u8, uint32_t count, void* stream) {
    KJ_REQUIRE(scene && rays && out_u8, "null argument");
    if (!scene->committed) { set_last_error("scene not committed"); return KJ_ERR_NOT_COMMITTED; }
    if (count == 0) return KJ_OK;
    const SceneView sv = scene_view(*scene);
    if (count <= quad_max_rays(scene->dev))
        hipLaunchKernelGGL(k_trace_any_quad, dim3((count + 15) / 16), dim3(64), quad_stack_bytes(), (hipStream_t)stream, sv, (const float4*)rays, (uint8_t*)out_u8, count);
    else if (kj_debug_getenv("KJ_TRACE_PER_RAY"))
        hipLaunchKernelGGL(k_trace_any, dim3((count + 63) / 64), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, sv, (const float4*)rays, (uint8_t*)out_u8, count);
    else {
        uint32_t waves; const StreamTune tune = stream_launch(scene->dev, count, &waves);
        hipLaunchKernelGGL(k_trace_any_stream, dim3(waves), dim3(64), sv.bvh.stack_entries * 64 * 4, (hipStream_t)stream, sv, (const float4*)rays, (uint8_t*)out_u8, count, tune);
    }
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

KjStatus kj_debug_calibration_copy(void* dst, const void* src, uint64_t bytes, void* stream) {
    KJ_REQUIRE(dst && src && bytes % 16 == 0, "null argument / size not a multiple of 16");
    hipLaunchKernelGGL(pmc_calibration_copy, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, size_t(bytes / 16));
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}

// ---- reprojection
}  // extern "C"

struct KjReprojection {
    KjDevice* dev = nullptr;
    kj::DevBuf prev_depth, output;
    uint32_t W = 0, H = 0;
};

extern "C" {

KjStatus kj_reprojection_create(KjDevice* dev, KjReprojection** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjReprojection* r = new KjReprojection();
    r->dev = dev;
    *out = r;
    return KJ_OK;
}
void kj_reprojection_destroy(KjReprojection* r) { delete r; }

KjStatus kj_calculate_reprojection_map(KjReprojection* r, const KjGbufferDepth* gd, const void* velocity, const void** out_map, void* stream_) {
    KJ_REQUIRE(r && gd && velocity && out_map && r->dev->fc_dev, "null argument / kj_frame_begin not called");
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t W = gd->width, H = gd->height;
    if (W != r->W || H != r->H) {
        KJ_TRY_HIP(r->prev_depth.alloc(size_t(W) * H * 4, stream));  // temporal "reprojection.prev_depth", zero-initialised
        KJ_TRY_HIP(r->output.alloc(size_t(W) * H * 8, stream));
        r->W = W; r->H = H;
    }
    hipLaunchKernelGGL(k_reprojection_map, dim3((W + 7) / 8, (H + 7) / 8), dim3(64), 0, stream, r->dev->fc_dev, int(W), int(H), (const float*)gd->depth,
                       (const uint32_t*)gd->geometric_normal, (const float*)r->prev_depth.p, (const uint2*)velocity, (uint2*)r->output.p);
    KJ_CHECK_LAUNCH();
    // "copy depth" pass (renderers/reprojection.rs:37-49)
    KJ_TRY_HIP(hipMemcpyAsync(r->prev_depth.p, gd->depth, size_t(W) * H * 4, hipMemcpyDeviceToDevice, stream));
    *out_map = r->output.p;
    return KJ_OK;
}

}  // extern "C"
