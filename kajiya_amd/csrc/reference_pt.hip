// reference_path_trace for gfx950 (renderers/reference.rs:8-26; rt/reference_path_trace.rgen.hlsl:75-377): the
// convergence oracle of the GI path. One lane per pixel, 8x8 tile per wave64, BVH stack in LDS ([level][lane]).
// Shader switches as shipped: FIREFLY_SUPPRESSION, USE_PIXEL_FILTER, USE_SOFT_SHADOWS, USE_LIGHTS, USE_EMISSIVE.
// Multi-GPU (BASELINE config 5): tiles are dealt round-robin to `interleave_count` ranks; every rank accumulates
// its own tiles and the images are summed once at the end (non-owned texels stay 0).
#include "kj_host.hpp"
#include "kj_shading.hpp"
#include "kj_scene.hpp"

using namespace kj;
namespace kj { SceneView scene_view(const KjScene& s); }

static constexpr uint32_t PT_MAX_EYE_PATH_LENGTH = 16;
static constexpr uint32_t PT_RUSSIAN_ROULETTE_START_PATH_LENGTH = 3;

// reference_path_trace.rgen.hlsl:61-73
KJ_D float pt_inv_error_function(float x, float truncation) {
    const float ALPHA = 0.14f;
    const float INV_ALPHA = 1.0f / ALPHA;
    const float K = 2.0f / (KJ_PI * ALPHA);
    const float y = logf(fmaxf(truncation, 1.0f - x * x));
    const float z = K + 0.5f * y;
    const float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    return sqrtf(fmaxf(0.0f, sqrtf(z * z - y * INV_ALPHA) - z)) * s;
}
KJ_D float pt_remap_unorm_to_gaussian(float x, float truncation) { return pt_inv_error_function(x * 2.0f - 1.0f, truncation); }

struct PtArgs {
    const FrameConstants* __restrict__ fc;
    SceneView sc;
    const uint2* __restrict__ brdf_fg_lut;
    const float4* __restrict__ sun_color;
    float4* __restrict__ output;
    int W, H;
    int first_bounce_mode;   // 0 as shipped; 1 = the shader's INDIRECT_ONLY; 2 = indirect through a white Lambert first bounce
    uint32_t interleave_count, interleave_index;
    unsigned long long* __restrict__ ray_counter;
};

__global__ void __launch_bounds__(64) k_reference_path_trace(PtArgs a) {
    extern __shared__ uint32_t lds_stack[];
    const int tiles_x = (a.W + 7) / 8;
    const uint32_t tile = blockIdx.x * a.interleave_count + a.interleave_index;
    const int lane = threadIdx.x;
    const int x = int(tile % uint32_t(tiles_x)) * 8 + (lane & 7), y = int(tile / uint32_t(tiles_x)) * 8 + (lane >> 3);
    if (x >= a.W || y >= a.H) return;
    uint32_t* stack = lds_stack + lane;
    const FrameConstants& fc = *a.fc;
    const float4 prev = a.output[size_t(y) * a.W + x];
    if (!(prev.w < 1000.0f)) return;

    uint32_t rng = hash_combine2(hash_combine2(uint32_t(x), hash1(uint32_t(y))), fc.frame_index);
    float px_off0 = 0.5f, px_off1 = 0.5f;
    const float psf_scale = 0.4f;
    px_off0 += psf_scale * pt_remap_unorm_to_gaussian(uint_to_u01_float(hash1_mut(rng)), 1e-8f);
    px_off1 += psf_scale * pt_remap_unorm_to_gaussian(uint_to_u01_float(hash1_mut(rng)), 1e-8f);
    const V2 uv{(float(x) + px_off0) / float(a.W), (float(y) + px_off1) / float(a.H)};
    const ViewRay vrc = view_ray_from_uv(fc, uv);
    V3 ray_o = vrc.origin_ws, ray_d = normalize(vrc.dir_ws);
    float ray_tmin = 0.0f;
    V3 throughput = v3(1.0f), total_radiance = v3(0.0f);
    float roughness_bias = 0.0f;
    const float4 sc4 = *a.sun_color;
    const V3 sun_color{sc4.x, sc4.y, sc4.z};
    const bool indirect_only = a.first_bounce_mode != 0;
    uint32_t rays = 0;
    RayCone ray_cone = pixel_ray_cone_from_image_height(fc, float(a.H));   // :123-128
    ray_cone.spread_angle *= 0.3f;                                       // "bias for texture sharpness"

    for (uint32_t path_length = 0; path_length < PT_MAX_EYE_PATH_LENGTH; ++path_length) {
        ++rays;
        const GbufferPathVertex primary_hit = gbuffer_raytrace<false>(a.sc, fc, ray_o, ray_d, ray_tmin, FLT_MAX, path_length, false, stack, 64, nullptr, ray_cone);
        if (!primary_hit.is_hit) {
            total_radiance += throughput * atmosphere_default(fc, ray_d, sun_direction(fc));
            break;
        }
        ray_cone = ray_cone.propagate(0.0f, primary_hit.ray_t);          // :151-152
        V2 su;
        su.x = uint_to_u01_float(hash1_mut(rng));
        su.y = uint_to_u01_float(hash1_mut(rng));
        const V3 to_light_norm = sample_sun_direction(fc, su, true);
        bool is_shadowed = true;
        if (!(indirect_only && path_length == 0)) {
            ++rays;
            is_shadowed = rt_is_shadowed<false>(a.sc, primary_hit.position, to_light_norm, 1e-4f, FLT_MAX, stack, 64);
        }
        GbufferData gbuffer = gbuffer_unpack(primary_hit.gbuffer_packed);
        if (dot(gbuffer.normal, ray_d) >= 0.0f) {
            if (path_length == 0) gbuffer.normal = -gbuffer.normal;
            else break;
        }
        if (indirect_only && path_length == 0) { gbuffer.albedo = v3(1.0f); gbuffer.metalness = 0.0f; }
        const Basis tangent_to_world = build_orthonormal_basis(gbuffer.normal);
        const V3 wi = to_local(tangent_to_world, to_light_norm);
        V3 wo = to_local(tangent_to_world, -ray_d);
        if (wo.z < 0.0f) { wo.z *= -0.25f; wo = normalize(wo); }
        LayeredBrdf brdf = layered_brdf_from_gbuffer_ndotv(a.brdf_fg_lut, gbuffer, wo.z);
        brdf.roughness = lerp(brdf.roughness, 1.0f, roughness_bias);  // FIREFLY_SUPPRESSION
        const bool lambert_first = a.first_bounce_mode == 2 && path_length == 0;
        if (!lambert_first) {
            const V3 brdf_value = layered_brdf_evaluate_directional_light(brdf, wo, wi);
            const V3 light_radiance = is_shadowed ? v3(0.0f) : sun_color;
            total_radiance += throughput * brdf_value * light_radiance * fmaxf(0.0f, wi.z);
            total_radiance += gbuffer.emissive * throughput;
        }
        if (!lambert_first && fc.triangle_light_count > 0 && a.sc.light_count > 0) {
            const float light_selection_pmf = 1.0f / float(fc.triangle_light_count);
            const uint32_t light_idx = hash1_mut(rng) % fc.triangle_light_count;
            V2 urand;
            urand.x = uint_to_u01_float(hash1_mut(rng));
            urand.y = uint_to_u01_float(hash1_mut(rng));
            const KjTriangleLight tl = a.sc.lights[min(light_idx, a.sc.light_count - 1)];
            const V3 v0{tl.verts[0], tl.verts[1], tl.verts[2]}, v1{tl.verts[3], tl.verts[4], tl.verts[5]}, v2{tl.verts[6], tl.verts[7], tl.verts[8]};
            const LightSampleArea ls = sample_triangle_light(v0, v1 - v0, v2 - v0, urand);
            const V3 to_light_ws = ls.pos - primary_hit.position;
            const float dist_to_light2 = dot(to_light_ws, to_light_ws);
            const V3 to_light_norm_ws = to_light_ws * (1.0f / sqrtf(dist_to_light2));
            const float to_psa_metric = fmaxf(0.0f, dot(to_light_norm_ws, gbuffer.normal)) * fmaxf(0.0f, dot(to_light_norm_ws, -ls.normal)) / dist_to_light2;
            if (to_psa_metric > 0.0f) {
                const V3 wi2 = to_local(tangent_to_world, to_light_norm_ws);
                ++rays;
                const bool sh = rt_is_shadowed<false>(a.sc, primary_hit.position, to_light_norm_ws, 1e-3f, sqrtf(dist_to_light2) - 2e-3f, stack, 64);
                if (!sh) total_radiance += throughput * V3{tl.radiance[0], tl.radiance[1], tl.radiance[2]} * layered_brdf_evaluate(brdf, wo, wi2) / ls.pdf * to_psa_metric / light_selection_pmf;
            }
        }
        V3 urand3;
        urand3.x = uint_to_u01_float(hash1_mut(rng));
        urand3.y = uint_to_u01_float(hash1_mut(rng));
        urand3.z = uint_to_u01_float(hash1_mut(rng));
        BrdfSample brdf_sample;
        if (lambert_first) {
            brdf_sample = diffuse_sample(v3(1.0f), V2{urand3.x, urand3.y});
        } else {
            brdf_sample = layered_brdf_sample(brdf, wo, urand3);
        }
        if (!(brdf_sample.wi.z > 1e-6f)) break;   // BrdfSample::is_valid
        roughness_bias = lerp(roughness_bias, 1.0f, 0.5f * brdf_sample.approx_roughness);
        ray_o = primary_hit.position;
        ray_d = to_world(tangent_to_world, brdf_sample.wi);
        ray_tmin = 1e-4f;
        throughput = throughput * brdf_sample.value_over_pdf;
        if (path_length >= PT_RUSSIAN_ROULETTE_START_PATH_LENGTH) {
            const float rr_coin = uint_to_u01_float(hash1_mut(rng));
            const float continue_p = fmaxf(gbuffer.albedo.x, fmaxf(gbuffer.albedo.y, gbuffer.albedo.z));
            if (rr_coin > continue_p) break;
            throughput = throughput / continue_p;
        }
    }
    float4 cur = make_float4(0, 0, 0, 0);
    if (total_radiance.x >= 0.0f && total_radiance.y >= 0.0f && total_radiance.z >= 0.0f) cur = make_float4(total_radiance.x, total_radiance.y, total_radiance.z, 1.0f);
    const float tsc = cur.w + prev.w;
    const float lrp = cur.w / fmaxf(1.0f, tsc);
    const V3 c = V3{cur.x, cur.y, cur.z} / fmaxf(1.0f, cur.w);
    const V3 o = vmax(v3(0.0f), lerp(V3{prev.x, prev.y, prev.z}, c, lrp));
    a.output[size_t(y) * a.W + x] = make_float4(o.x, o.y, o.z, fmaxf(1.0f, tsc));
    if (a.ray_counter) {   // optional; one atomic per lane would serialise on a single address
        atomicAdd(a.ray_counter, (unsigned long long)rays);
    }
}

extern "C" {

// reference_path_trace(rg, &mut output_img, bindless_set, tlas) (reference.rs:8-13). `output` = RGBA32F accumulation image
// (rgb = running mean, a = sample count), persistent across calls like the reference's `refpt.accum` image.
KjStatus kj_reference_path_trace(KjDevice* dev, const KjScene* scene, void* output, uint32_t width, uint32_t height, uint32_t first_bounce_mode,
                                 uint32_t interleave_count, uint32_t interleave_index, uint64_t* ray_counter_dev, void* stream) {
    KJ_REQUIRE(dev && scene && output && width && height, "null argument");
    KJ_REQUIRE(dev->fc_dev, "kj_frame_begin not called");
    KJ_REQUIRE(scene->committed, "scene not committed");
    KJ_REQUIRE(first_bounce_mode <= 2, "first_bounce_mode must be 0, 1 or 2");
    if (interleave_count == 0) interleave_count = 1;
    KJ_REQUIRE(interleave_index < interleave_count, "interleave_index out of range");
    hipStream_t s = (hipStream_t)stream;
    PtArgs a;
    a.fc = dev->fc_dev;
    a.sc = scene_view(*scene);
    a.brdf_fg_lut = (const uint2*)dev->brdf_fg_lut.p;
    a.sun_color = (const float4*)dev->sun_color.p + dev->fc_slot;
    a.output = (float4*)output;
    a.W = int(width); a.H = int(height);
    a.first_bounce_mode = int(first_bounce_mode);
    a.interleave_count = interleave_count; a.interleave_index = interleave_index;
    a.ray_counter = (unsigned long long*)ray_counter_dev;
    const uint32_t tiles = ((width + 7) / 8) * ((height + 7) / 8);
    const uint32_t my_tiles = (tiles - interleave_index + interleave_count - 1) / interleave_count;
    const size_t lds = size_t(a.sc.bvh.stack_entries) * 64 * 4;
    KJ_REQUIRE(lds <= 64 * 1024, "BVH too deep for the LDS traversal stack");
    if (my_tiles) hipLaunchKernelGGL(k_reference_path_trace, dim3(my_tiles), dim3(64), lds, s, a);
    KJ_TRY_HIP(hipGetLastError());
    return KJ_OK;
}

}  // extern "C"
