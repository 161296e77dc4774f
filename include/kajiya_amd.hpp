// kajiya_amd.hpp — C++17 host mirror of the reference's renderer interface for the hot path, header-only, on top of the C-ABI
// (kajiya_amd.h). The reference's host language is Rust (absent from this build image); this is the compiled-code host side a
// kajiya-style application links against: the same type and method names, argument meaning and call order as
//   crates/lib/kajiya/src/world_render_passes.rs:13-292   (WorldRenderer::prepare_render_graph_standard)
//   crates/lib/kajiya/src/world_renderer.rs:604-911,1001-1129 (scene edits, prepare_frame_constants, supersample offsets)
//   crates/lib/kajiya/src/camera.rs:47-125, rust-shaders-shared/src/view_constants.rs:25-121
//   crates/lib/kajiya/src/renderers/{rtdgi,ircache,rtr,taa,ssgi,shadows,shadow_denoise,deferred,reprojection,sky,reference}.rs
// Error behaviour: the reference panics / returns anyhow errors; every failed C-ABI call throws kajiya_amd::Error carrying
// kj_last_error(). All GPU work is enqueued on the caller's stream; nothing here synchronises.
// Device images are plain linear allocations (DeviceImage); a Vulkan host imports its images as external memory instead.
#pragma once
#include <hip/hip_runtime_api.h>
#include <array>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "kajiya_amd.h"

namespace kajiya_amd {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(KjStatus st, const char* what) {
    if (st != KJ_OK) throw Error(std::string(what) + ": " + kj_last_error());
}
inline void check_hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------- small linear algebra (glam-compatible memory order: column-major)
struct Mat4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float& at(int r, int c) { return m[c * 4 + r]; }
    float at(int r, int c) const { return m[c * 4 + r]; }
};
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 o;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.at(r, k) * b.at(k, c);
            o.at(r, c) = s;
        }
    return o;
}

// CameraLens::calc_matrices + CameraBodyMatrices (camera.rs:66-125): infinite reverse-Z projection, camera looks down -Z.
struct CameraMatrices {
    Mat4 view_to_clip, clip_to_view, view_to_world, world_to_view;
    static CameraMatrices look_at(const double eye[3], const double target[3], double vfov_deg, double aspect, double znear = 0.01) {
        auto norm = [](double* v) { const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
        double f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
        norm(f);
        const double up[3] = {0, 1, 0};
        double r[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
        norm(r);
        const double u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
        const double rot[3][3] = {{r[0], u[0], -f[0]}, {r[1], u[1], -f[1]}, {r[2], u[2], -f[2]}};   // columns: right, up, -forward
        CameraMatrices c;
        const double fov = vfov_deg * 3.14159265358979323846 / 180.0;
        const double h = std::cos(0.5 * fov) / std::sin(0.5 * fov), w = h / aspect;
        c.view_to_clip = Mat4{}; std::memset(c.view_to_clip.m, 0, sizeof(c.view_to_clip.m));
        c.view_to_clip.at(0, 0) = float(w); c.view_to_clip.at(1, 1) = float(h); c.view_to_clip.at(3, 2) = -1.0f; c.view_to_clip.at(2, 3) = float(znear);
        std::memset(c.clip_to_view.m, 0, sizeof(c.clip_to_view.m));
        c.clip_to_view.at(0, 0) = float(1.0 / w); c.clip_to_view.at(1, 1) = float(1.0 / h); c.clip_to_view.at(3, 2) = float(1.0 / znear); c.clip_to_view.at(2, 3) = -1.0f;
        for (int i = 0; i < 3; ++i) {
            double t = 0.0;
            for (int j = 0; j < 3; ++j) {
                c.view_to_world.at(i, j) = float(rot[i][j]);
                c.world_to_view.at(i, j) = float(rot[j][i]);
                t += rot[j][i] * eye[j];
            }
            c.view_to_world.at(i, 3) = float(eye[i]);
            c.world_to_view.at(i, 3) = float(-t);
        }
        return c;
    }
};

// Halton(2,3) supersample offsets (world_renderer.rs:425-428,1116-1129)
inline float radical_inverse(uint32_t n, uint32_t base) {
    float val = 0.0f;
    const float inv_base = 1.0f / float(base);
    float inv_bi = inv_base;
    while (n > 0) {
        const uint32_t d_i = n % base;
        val += float(d_i) * inv_bi;
        n = uint32_t(float(n) * inv_base);
        inv_bi *= inv_base;
    }
    return val;
}

// ---------------------------------------------------------------- exposure (world_renderer.rs:217-285,919-948), f32 as there
struct HistogramClipping { float low = 0.0f, high = 0.0f; };
struct DynamicExposureState {
    bool enabled = false;
    float speed_log2 = 0.0f;
    HistogramClipping histogram_clipping;
    float ev_fast = 0.0f, ev_slow = 0.0f;
    float ev_smoothed() const { return enabled ? (ev_slow + ev_fast) * 0.5f + -2.0f /* DYNAMIC_EXPOSURE_BIAS */ : 0.0f; }
    void update(float ev, float dt) {
        if (!enabled) return;
        ev = std::min(std::max(ev, -16.0f), 16.0f);
        dt = dt * std::exp2(speed_log2);
        const float t_fast = 1.0f - std::exp(-1.0f * dt);
        ev_fast = (ev - ev_fast) * t_fast + ev_fast;
        const float t_slow = 1.0f - std::exp(-0.25f * dt);
        ev_slow = (ev - ev_slow) * t_slow + ev_slow;
    }
};
struct ExposureState { float pre_mult = 1.0f, post_mult = 1.0f, pre_mult_prev = 1.0f, pre_mult_delta = 1.0f; };
enum class RenderMode { Standard = 0, Reference = 1 };
// WorldRenderer::update_pre_exposure (world_renderer.rs:919-948); image_log2_lum = PostProcessRenderer::image_log2_lum
inline void update_pre_exposure(ExposureState& st, DynamicExposureState& dynamic_exposure, float ev_shift, float image_log2_lum, RenderMode mode) {
    const float dt = 1.0f / 60.0f;
    dynamic_exposure.update(-image_log2_lum, dt);
    const float ev_mult = std::exp2(ev_shift + dynamic_exposure.ev_smoothed());
    st.pre_mult_prev = st.pre_mult;
    if (mode == RenderMode::Standard) {
        st.pre_mult = st.pre_mult * 0.9f + ev_mult * 0.1f;
        st.post_mult = ev_mult / st.pre_mult;
    } else {
        st.pre_mult = 1.0f;
        st.post_mult = ev_mult;
    }
    st.pre_mult_delta = st.pre_mult / st.pre_mult_prev;
}

// The part of WorldRenderer that produces FrameConstants each frame (prepare_frame_constants, world_renderer.rs:1001-1108).
struct FrameState {
    uint32_t render_extent[2];
    float sun_direction[3];
    float sun_size_multiplier = 1.0f;
    float sun_color_multiplier[3] = {1, 1, 1}, sky_ambient[3] = {0, 0, 0};
    bool use_taa_jitter = true;
    uint32_t frame_idx = 0, triangle_light_count = 0;
    float pre_exposure = 1.0f, pre_exposure_prev = 1.0f, pre_exposure_delta = 1.0f;   // ExposureState (world_renderer.rs:1084-1086)
    bool have_prev = false;
    CameraMatrices prev_camera;

    FrameState(uint32_t w, uint32_t h, double sx = 4.0, double sy = 1.0, double sz = 1.0) {
        render_extent[0] = w; render_extent[1] = h;
        const double l = std::sqrt(sx * sx + sy * sy + sz * sz);
        sun_direction[0] = float(sx / l); sun_direction[1] = float(sy / l); sun_direction[2] = float(sz / l);
    }
    // `ircache` (may be null): IrcacheRenderer::update_eye_position + constants run here, as in world_renderer.rs:1061-1069.
    KjFrameConstants prepare_frame_constants(const CameraMatrices& cam, KjIrcache* ircache = nullptr, float delta_time_seconds = 1.0f / 60.0f) {
        const CameraMatrices& prev = have_prev ? prev_camera : cam;
        KjFrameConstants fc;
        std::memset(&fc, 0, sizeof(fc));
        KjViewConstants& vc = fc.view_constants;
        const Mat4 clip_to_prev_clip = (prev.view_to_clip * prev.world_to_view) * (cam.view_to_world * cam.clip_to_view);
        auto set = [](float* dst, const Mat4& m) { std::memcpy(dst, m.m, sizeof(m.m)); };
        set(vc.view_to_clip, cam.view_to_clip); set(vc.clip_to_view, cam.clip_to_view);
        set(vc.world_to_view, cam.world_to_view); set(vc.view_to_world, cam.view_to_world);
        set(vc.clip_to_prev_clip, clip_to_prev_clip);
        set(vc.prev_view_to_prev_clip, prev.view_to_clip); set(vc.prev_clip_to_prev_view, prev.clip_to_view);
        set(vc.prev_world_to_prev_view, prev.world_to_view); set(vc.prev_view_to_prev_world, prev.view_to_world);
        float off[2] = {0.0f, 0.0f};
        if (use_taa_jitter) {
            const uint32_t i = frame_idx % 128u + 1u;
            off[0] = radical_inverse(i, 2) - 0.5f; off[1] = radical_inverse(i, 3) - 0.5f;
        }
        const float soc[2] = {float(2.0 * double(off[0])) / float(render_extent[0]), float(2.0 * double(off[1])) / float(render_extent[1])};
        Mat4 jitter, jitter_inv;
        jitter.at(0, 3) = -soc[0]; jitter.at(1, 3) = -soc[1];
        jitter_inv.at(0, 3) = soc[0]; jitter_inv.at(1, 3) = soc[1];
        set(vc.view_to_sample, jitter * cam.view_to_clip);
        set(vc.sample_to_view, cam.clip_to_view * jitter_inv);
        vc.sample_offset_pixels[0] = off[0]; vc.sample_offset_pixels[1] = off[1];
        vc.sample_offset_clip[0] = soc[0]; vc.sample_offset_clip[1] = soc[1];
        for (int i = 0; i < 3; ++i) { fc.sun_direction[i] = sun_direction[i]; fc.sun_color_multiplier[i] = sun_color_multiplier[i]; fc.sky_ambient[i] = sky_ambient[i]; }
        fc.frame_index = frame_idx;
        fc.delta_time_seconds = delta_time_seconds;
        const double real_sun_angular_radius = (0.53 * 3.14159265358979323846 / 180.0) * 0.5;
        fc.sun_angular_radius_cos = float(std::cos(double(sun_size_multiplier) * real_sun_angular_radius));
        fc.triangle_light_count = triangle_light_count;
        fc.pre_exposure = pre_exposure; fc.pre_exposure_prev = pre_exposure_prev; fc.pre_exposure_delta = pre_exposure_delta;
        fc.render_overrides.flags = 0; fc.render_overrides.material_roughness_scale = 1.0f;
        fc.ircache_grid_center[3] = 1.0f;
        if (ircache) {
            const float eye[3] = {cam.view_to_world.at(0, 3), cam.view_to_world.at(1, 3), cam.view_to_world.at(2, 3)};
            check(kj_ircache_update_eye_position(ircache, eye), "kj_ircache_update_eye_position");
            check(kj_ircache_constants(ircache, &fc), "kj_ircache_constants");
        }
        prev_camera = cam; have_prev = true;
        return fc;
    }
    void retire_frame() { ++frame_idx; }
};

// ---------------------------------------------------------------- device memory
struct DeviceImage {
    void* p = nullptr; size_t bytes = 0;
    DeviceImage() {}
    explicit DeviceImage(size_t n) { alloc(n); }
    DeviceImage(const DeviceImage&) = delete; DeviceImage& operator=(const DeviceImage&) = delete;
    ~DeviceImage() { if (p) (void)hipFree(p); }
    void alloc(size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        check_hip(hipMalloc(&p, n), "hipMalloc"); bytes = n;
        check_hip(hipMemset(p, 0, n), "hipMemset");
    }
};

// ---------------------------------------------------------------- RenderBackend / WorldRenderer::new
struct Device {
    KjDevice* h = nullptr;
    Device(int ordinal, const uint8_t* blue_noise_rgba8_256) { check(kj_device_create(ordinal, blue_noise_rgba8_256, &h), "kj_device_create"); }
    Device(const Device&) = delete; Device& operator=(const Device&) = delete;
    ~Device() { kj_device_destroy(h); }
};

// WorldRenderer scene state (world_renderer.rs:604-911)
struct MeshHandle { uint32_t idx; };
struct InstanceHandle { uint32_t idx; };
struct Scene {
    KjScene* h = nullptr;
    explicit Scene(Device& dev) { check(kj_scene_create(dev.h, &h), "kj_scene_create"); }
    Scene(const Scene&) = delete; Scene& operator=(const Scene&) = delete;
    ~Scene() { kj_scene_destroy(h); }
    MeshHandle add_mesh(const KjMeshDesc& d) { uint32_t i = 0; check(kj_scene_add_mesh(h, &d, &i), "kj_scene_add_mesh"); return MeshHandle{i}; }
    // `cache/<name>.mesh` bytes (PackedTriMesh::Flat) + a resolver from image identity to `cache/<id>.image` bytes; images are decoded to RGBA8 here.
    template <typename ImageBytes> MeshHandle add_baked_mesh(const void* mesh_bytes, uint64_t mesh_size, ImageBytes&& image_bytes, bool use_lights = false) {
        KjBakedMeshView v;
        check(kj_baked_mesh_view(mesh_bytes, mesh_size, &v), "kj_baked_mesh_view");
        std::vector<KjMaterialMap> maps(v.map_count);
        std::vector<std::vector<uint8_t>> texels(v.map_count);
        for (uint32_t k = 0; k < v.map_count; ++k) {
            const std::vector<uint8_t>& img = image_bytes(v.map_identities[k]);
            KjBakedImageView iv;
            check(kj_baked_image_view(img.data(), img.size(), &iv), "kj_baked_image_view");
            KjMaterialMap& m = maps[k];
            std::memset(&m, 0, sizeof(m));
            size_t total = 0;
            for (uint32_t l = 0; l < iv.mip_count; ++l) total += size_t(std::max(1u, iv.extent[0] >> l)) * std::max(1u, iv.extent[1] >> l) * 4;
            texels[k].resize(total);
            size_t off = 0;
            for (uint32_t l = 0; l < iv.mip_count; ++l) {
                const uint32_t w = std::max(1u, iv.extent[0] >> l), hgt = std::max(1u, iv.extent[1] >> l);
                const uint8_t* data; uint64_t len;
                check(kj_baked_image_mip(img.data(), img.size(), l, &data, &len), "kj_baked_image_mip");
                check(kj_baked_image_decode_rgba8(iv.vk_format, data, len, w, hgt, texels[k].data() + off), "kj_baked_image_decode_rgba8");
                off += size_t(w) * hgt * 4;
            }
            const bool srgb = iv.vk_format == 43 || iv.vk_format == 132 || iv.vk_format == 134 || iv.vk_format == 138 || iv.vk_format == 146;
            if (iv.extent[0] == 1 && iv.extent[1] == 1 && iv.mip_count == 1 && !srgb) {   // a baked MeshMaterialMap::Placeholder (mesh.rs:845-853)
                std::memcpy(m.placeholder_rgba, texels[k].data(), 4);
            } else {
                m.image_rgba8 = texels[k].data(); m.width = iv.extent[0]; m.height = iv.extent[1]; m.mip_count = iv.mip_count; m.srgb = srgb ? 1u : 0u;
            }
        }
        KjMeshDesc d;
        std::memset(&d, 0, sizeof(d));
        d.verts = v.verts; d.vertex_count = v.vertex_count; d.uvs = v.uvs; d.tangents = v.tangents; d.colors = v.colors; d.material_ids = v.material_ids;
        d.indices = v.indices; d.index_count = v.index_count; d.materials = v.materials; d.material_count = v.material_count;
        d.maps = maps.data(); d.map_count = v.map_count; d.use_lights = use_lights ? 1u : 0u;
        return add_mesh(d);
    }
    InstanceHandle add_instance(MeshHandle mesh, const float transform3x4[12]) { uint32_t i = 0; check(kj_scene_add_instance(h, mesh.idx, transform3x4, &i), "kj_scene_add_instance"); return InstanceHandle{i}; }
    void set_instance_transform(InstanceHandle inst, const float transform3x4[12]) { check(kj_scene_set_instance_transform(h, inst.idx, transform3x4), "kj_scene_set_instance_transform"); }
    void remove_instance(InstanceHandle inst) { check(kj_scene_remove_instance(h, inst.idx), "kj_scene_remove_instance"); }
    // how the BLAS of meshes added from now on is built (ray_tracing.rs:438 build flags): PREFER_FAST_TRACE = binned SAH on the host,
    // PREFER_FAST_BUILD = linear BVH on the device
    void set_blas_build_mode(uint32_t kj_blas_build_mode) { check(kj_scene_set_blas_build_mode(h, kj_blas_build_mode), "kj_scene_set_blas_build_mode"); }     // KJ_BLAS_BUILD_*
    void set_top_build_mode(uint32_t kj_top_build_mode) { check(kj_scene_set_top_build_mode(h, kj_top_build_mode), "kj_scene_set_top_build_mode"); }           // KJ_TOP_BUILD_*: who builds the per-frame TLAS
    void set_blas_build_mode(bool prefer_fast_build) { check(kj_scene_set_blas_build_mode(h, prefer_fast_build ? KJ_BLAS_BUILD_FAST_BUILD : KJ_BLAS_BUILD_FAST_TRACE), "kj_scene_set_blas_build_mode"); }
    // host ms of the last build_ray_tracing_top_level_acceleration: {BLAS builds, instance tables + top tree, uploads + device refit, total}
    std::array<double, 4> last_commit_ms() const { std::array<double, 4> t{}; check(kj_scene_last_commit_ms(h, t.data()), "kj_scene_last_commit_ms"); return t; }
    // build_ray_tracing_top_level_acceleration + prepare_top_level_acceleration
    void build_ray_tracing_top_level_acceleration(hipStream_t s) { check(kj_scene_commit(h, s), "kj_scene_commit"); }
    uint32_t triangle_light_count() const { uint32_t n = 0; check(kj_scene_triangle_light_count(h, &n), "kj_scene_triangle_light_count"); return n; }
};

// ---------------------------------------------------------------- renderers (names and methods of crates/lib/kajiya/src/renderers/*.rs)
struct GbufferDepth {   // renderers/mod.rs:31-71
    DeviceImage geometric_normal, gbuffer, depth, velocity;
    uint32_t width = 0, height = 0;
    GbufferDepth(uint32_t w, uint32_t h) : geometric_normal(size_t(w) * h * 4), gbuffer(size_t(w) * h * 16), depth(size_t(w) * h * 4), velocity(size_t(w) * h * 8), width(w), height(h) {}
    KjGbufferDepth view() const { KjGbufferDepth g; g.geometric_normal = geometric_normal.p; g.gbuffer = gbuffer.p; g.depth = depth.p; g.width = width; g.height = height; return g; }
};

struct ReprojectionRenderer {
    KjReprojection* h = nullptr;
    explicit ReprojectionRenderer(Device& d) { check(kj_reprojection_create(d.h, &h), "kj_reprojection_create"); }
    ~ReprojectionRenderer() { kj_reprojection_destroy(h); }
    const void* calculate_reprojection_map(const GbufferDepth& g, hipStream_t s) {
        const KjGbufferDepth v = g.view(); const void* out = nullptr;
        check(kj_calculate_reprojection_map(h, &v, g.velocity.p, &out, s), "kj_calculate_reprojection_map"); return out;
    }
};

struct SsgiRenderer {
    KjSsgi* h = nullptr;
    explicit SsgiRenderer(Device& d) { check(kj_ssgi_create(d.h, &h), "kj_ssgi_create"); }
    ~SsgiRenderer() { kj_ssgi_destroy(h); }
    const void* render(const GbufferDepth& g, const void* reprojection_map, const void* prev_radiance, hipStream_t s) {
        const KjGbufferDepth v = g.view(); const void* out = nullptr;
        check(kj_ssgi_render(h, &v, reprojection_map, prev_radiance, &out, s), "kj_ssgi_render"); return out;
    }
};

struct ShadowDenoiseRenderer {
    KjShadowDenoise* h = nullptr;
    explicit ShadowDenoiseRenderer(Device& d) { check(kj_shadow_denoise_create(d.h, &h), "kj_shadow_denoise_create"); }
    ~ShadowDenoiseRenderer() { kj_shadow_denoise_destroy(h); }
    const void* render(const GbufferDepth& g, const void* shadow_mask, const void* reprojection_map, hipStream_t s) {
        const KjGbufferDepth v = g.view(); const void* out = nullptr;
        check(kj_shadow_denoise_render(h, &v, shadow_mask, reprojection_map, &out, s), "kj_shadow_denoise_render"); return out;
    }
};

struct IrcacheRenderState;
struct IrcacheRenderer {
    KjIrcache* h = nullptr;
    explicit IrcacheRenderer(Device& d) { check(kj_ircache_create(d.h, &h), "kj_ircache_create"); }
    ~IrcacheRenderer() { kj_ircache_destroy(h); }
    inline IrcacheRenderState prepare(hipStream_t s);
};
struct IrcacheRenderState {   // ircache.rs:59-78,360-506
    KjIrcache* h;
    void trace_irradiance(Scene& scene, const void* convolved_sky_cube, uint32_t width, hipStream_t s) { check(kj_ircache_trace_irradiance(h, scene.h, convolved_sky_cube, width, s), "kj_ircache_trace_irradiance"); }
    void sum_up_irradiance_for_sampling(hipStream_t s) { check(kj_ircache_sum_up_irradiance_for_sampling(h, s), "kj_ircache_sum_up_irradiance_for_sampling"); }
};
inline IrcacheRenderState IrcacheRenderer::prepare(hipStream_t s) { check(kj_ircache_prepare(h, s), "kj_ircache_prepare"); return IrcacheRenderState{h}; }

struct RtdgiRenderer {
    KjRtdgi* h = nullptr;
    explicit RtdgiRenderer(Device& d) { check(kj_rtdgi_create(d.h, &h), "kj_rtdgi_create"); }
    ~RtdgiRenderer() { kj_rtdgi_destroy(h); }
    void reproject(const void* reprojection_map, uint32_t w, uint32_t hgt, hipStream_t s) { check(kj_rtdgi_reproject(h, reprojection_map, w, hgt, s), "kj_rtdgi_reproject"); }
    KjRtdgiOutput render(const GbufferDepth& g, const void* reprojection_map, const void* convolved_sky_cube, uint32_t sky_width, Scene& scene,
                         IrcacheRenderState* ircache, const void* ssao_tex, hipStream_t s) {
        KjRtdgiRenderParams p;
        std::memset(&p, 0, sizeof(p));
        p.gbuffer_depth = g.view(); p.reprojection_map = reprojection_map; p.sky_cube = convolved_sky_cube; p.sky_cube_width = sky_width;
        p.scene = scene.h; p.ircache = ircache ? ircache->h : nullptr; p.ssao_tex = ssao_tex; p.pass_mask = KJ_RTDGI_PASS_ALL;
        KjRtdgiOutput out;
        check(kj_rtdgi_render(h, &p, &out, s), "kj_rtdgi_render"); return out;
    }
};

struct TracedRtr;
struct RtrRenderer {
    KjRtr* h = nullptr;
    RtrRenderer(Device& d, const KjRtrTables& tables) { check(kj_rtr_create(d.h, &tables, &h), "kj_rtr_create"); }
    ~RtrRenderer() { kj_rtr_destroy(h); }
    inline TracedRtr trace(const GbufferDepth& g, const void* reprojection_map, const void* sky_cube, uint32_t sky_width, Scene& scene,
                           const KjRtdgiOutput& rtdgi, IrcacheRenderState* ircache, hipStream_t s);
};
struct TracedRtr {   // rtr.rs:74-80,440-480
    KjRtr* h; KjRtrParams params;
    const void* filter_temporal(hipStream_t s) { const void* out = nullptr; check(kj_rtr_filter_temporal(h, &params, &out, s), "kj_rtr_filter_temporal"); return out; }
};
inline TracedRtr RtrRenderer::trace(const GbufferDepth& g, const void* reprojection_map, const void* sky_cube, uint32_t sky_width, Scene& scene,
                                    const KjRtdgiOutput& rtdgi, IrcacheRenderState* ircache, hipStream_t s) {
    KjRtrParams p;
    std::memset(&p, 0, sizeof(p));
    p.gbuffer_depth = g.view(); p.reprojection_map = reprojection_map; p.sky_cube = sky_cube; p.sky_cube_width = sky_width; p.scene = scene.h;
    p.ircache = ircache ? ircache->h : nullptr; p.rtdgi_irradiance = rtdgi.screen_irradiance_tex;
    p.candidate_radiance_tex = const_cast<void*>(rtdgi.candidate_radiance_tex); p.candidate_hit_tex = const_cast<void*>(rtdgi.candidate_hit_tex);
    p.candidate_normal_tex = const_cast<void*>(rtdgi.candidate_normal_tex); p.pass_mask = KJ_RTR_PASS_ALL;
    check(kj_rtr_trace(h, &p, s), "kj_rtr_trace");
    return TracedRtr{h, p};
}

// LightingRenderer (renderers/lighting.rs): specular from the triangle lights, rendered into rtr's resolved image so both are filtered jointly
struct LightingRenderer {
    void render_specular(TracedRtr& rtr, hipStream_t s) { check(kj_rtr_render_specular_lights(rtr.h, &rtr.params, s), "kj_rtr_render_specular_lights"); }
};

struct TaaRenderer {
    KjTaa* h = nullptr;
    explicit TaaRenderer(Device& d) { check(kj_taa_create(d.h, &h), "kj_taa_create"); }
    ~TaaRenderer() { kj_taa_destroy(h); }
    KjTaaOutput render(const void* input_tex, uint32_t w, uint32_t hgt, const void* reprojection_map, const void* depth_tex, const uint32_t output_extent[2], hipStream_t s) {
        KjTaaOutput out;
        check(kj_taa_render(h, input_tex, w, hgt, reprojection_map, depth_tex, output_extent[0], output_extent[1], &out, s), "kj_taa_render"); return out;
    }
};

// motion_blur(rg, input, depth, reprojection_map) -> Handle<Image>   (renderers/motion_blur.rs:5-72)
struct MotionBlurRenderer {
    KjMotionBlur* h = nullptr;
    explicit MotionBlurRenderer(Device& d) { check(kj_motion_blur_create(d.h, &h), "kj_motion_blur_create"); }
    ~MotionBlurRenderer() { kj_motion_blur_destroy(h); }
    MotionBlurRenderer(const MotionBlurRenderer&) = delete;
    MotionBlurRenderer& operator=(const MotionBlurRenderer&) = delete;
    const void* render(const void* input_rgba16f, const uint32_t extent[2], const void* depth, const void* reprojection_map, const uint32_t depth_extent[2], hipStream_t s) {
        const void* out = nullptr;
        check(kj_motion_blur_render(h, input_rgba16f, extent[0], extent[1], depth, reprojection_map, depth_extent[0], depth_extent[1], &out, s), "kj_motion_blur_render");
        return out;
    }
};

// PostProcessRenderer (renderers/post.rs:112-272). `image_log2_lum` is refreshed by read_back_histogram at the top of render(), from
// whatever histogram copy has completed (the reference reads its mapped buffer the same way, a frame or more behind).
struct PostProcessRenderer {
    KjPost* h = nullptr;
    float image_log2_lum = 0.0f;
    PostProcessRenderer(Device& d, const uint16_t* bezold_brucke_lut_rg16f_64) { check(kj_post_create(d.h, bezold_brucke_lut_rg16f_64, &h), "kj_post_create"); }
    ~PostProcessRenderer() { kj_post_destroy(h); }
    PostProcessRenderer(const PostProcessRenderer&) = delete;
    PostProcessRenderer& operator=(const PostProcessRenderer&) = delete;
    // -> B10G11R11_UFLOAT image, linear display-referred
    const void* render(const void* input, uint32_t w, uint32_t hgt, float post_exposure_mult, float contrast, HistogramClipping exposure_histogram_clipping, hipStream_t s,
                       uint32_t input_format = KJ_POST_INPUT_RGBA16F) {
        check(kj_post_read_back_histogram(h, exposure_histogram_clipping.low, exposure_histogram_clipping.high, &image_log2_lum, nullptr), "kj_post_read_back_histogram");
        const void* out = nullptr;
        check(kj_post_render(h, input, input_format, w, hgt, post_exposure_mult, contrast, &out, s), "kj_post_render");
        return out;
    }
};

// ---------------------------------------------------------------- WorldRenderer::prepare_render_graph_standard (world_render_passes.rs:13-292)
struct FrameOutput {
    const void* reprojection_map; const void* ssgi_tex; const void* denoised_shadow_mask; KjRtdgiOutput rtdgi; const void* rtr;
    const void* lit /* RGBA16F "debug_out_tex" */; KjTaaOutput anti_aliased;
    const void* final_post_input = nullptr; // RGBA16F after motion blur, when WorldRenderer::post is set
    const void* post_processed = nullptr;   // B10G11R11_UFLOAT, when WorldRenderer::post is set
};
struct WorldRenderer {
    Device& device; Scene& scene;
    uint32_t render_extent[2], temporal_upscale_extent[2];
    GbufferDepth gbuffer_depth;
    DeviceImage sky_cube, convolved_sky_cube, sun_shadow_mask, accum_img, debug_out_tex;
    ReprojectionRenderer reprojection; SsgiRenderer ssgi; ShadowDenoiseRenderer shadow_denoise; IrcacheRenderer ircache; RtdgiRenderer rtdgi;
    RtrRenderer rtr; LightingRenderer lighting; TaaRenderer taa;
    FrameState frame_state;
    uint32_t debug_shading_mode = 0;
    bool reset_reference_accumulation = false;      // world_renderer.rs:183
    DeviceImage refpt_accum;                         // "refpt.accum" temporal (RGBA32F: running mean + sample count)
    float sky_key[8] = {-1, 0, 0, 0, 0, 0, 0, 0};
    // the tail of the frame (world_render_passes.rs:265-289): motion blur, then post
    std::unique_ptr<PostProcessRenderer> post;      // set by enable_post(); without it the frame ends at TAA, pre-exposure stays 1
    std::unique_ptr<MotionBlurRenderer> motion_blur;
    float ev_shift = 0.0f, contrast = 1.0f;
    DynamicExposureState dynamic_exposure;
    ExposureState exposure_state[2];                 // one per render mode
    void enable_post(const uint16_t* bezold_brucke_lut_rg16f_64) {
        post.reset(new PostProcessRenderer(device, bezold_brucke_lut_rg16f_64));
        motion_blur.reset(new MotionBlurRenderer(device));
    }
    void update_pre_exposure(RenderMode mode) {      // world_renderer.rs:919-948
        ExposureState& st = exposure_state[int(mode)];
        kajiya_amd::update_pre_exposure(st, dynamic_exposure, ev_shift, post ? post->image_log2_lum : 0.0f, mode);
        frame_state.pre_exposure = st.pre_mult; frame_state.pre_exposure_prev = st.pre_mult_prev; frame_state.pre_exposure_delta = st.pre_mult_delta;
    }

    WorldRenderer(Device& dev, Scene& sc, uint32_t w, uint32_t h, const KjRtrTables& rtr_tables)
        : device(dev), scene(sc), render_extent{w, h}, temporal_upscale_extent{w, h}, gbuffer_depth(w, h), sky_cube(6 * 64 * 64 * 8), convolved_sky_cube(6 * 16 * 16 * 8),
          sun_shadow_mask(size_t(w) * h), accum_img(size_t(w) * h * 8), debug_out_tex(size_t(w) * h * 8), reprojection(dev), ssgi(dev), shadow_denoise(dev), ircache(dev),
          rtdgi(dev), rtr(dev, rtr_tables), taa(dev), frame_state(w, h) {}

    // One frame of the lighting path for `camera`. The G-buffer comes from kj_raster_gbuffer here (the reference rasterises it; a host with
    // its own raster fills gbuffer_depth and calls the rest).
    FrameOutput prepare_render_graph_standard(const CameraMatrices& camera, hipStream_t s) {
        if (post) update_pre_exposure(RenderMode::Standard);                                                     // world_renderer.rs:959
        frame_state.triangle_light_count = scene.triangle_light_count();
        const KjFrameConstants fc = frame_state.prepare_frame_constants(camera, ircache.h);
        check(kj_frame_begin(device.h, &fc, s), "kj_frame_begin");
        const float key[8] = {fc.sun_direction[0], fc.sun_direction[1], fc.sun_direction[2], fc.sun_color_multiplier[0], fc.sun_color_multiplier[1], fc.sun_color_multiplier[2], fc.sky_ambient[0], fc.pre_exposure};
        if (std::memcmp(key, sky_key, sizeof(key)) != 0) {          // sky cube + convolution (world_render_passes.rs:30-38)
            check(kj_sky_cube_render(device.h, sky_cube.p, s), "kj_sky_cube_render");
            check(kj_sky_cube_convolve(device.h, sky_cube.p, convolved_sky_cube.p, s), "kj_sky_cube_convolve");
            std::memcpy(sky_key, key, sizeof(key));
        }
        const uint32_t W = render_extent[0], H = render_extent[1];
        check(kj_raster_gbuffer(device.h, scene.h, W, H, gbuffer_depth.geometric_normal.p, gbuffer_depth.gbuffer.p, gbuffer_depth.depth.p, gbuffer_depth.velocity.p, s), "kj_raster_gbuffer");
        FrameOutput o;
        o.reprojection_map = reprojection.calculate_reprojection_map(gbuffer_depth, s);                         // :79-80
        o.ssgi_tex = ssgi.render(gbuffer_depth, o.reprojection_map, accum_img.p, s);                           // :90-96
        IrcacheRenderState ircache_state = ircache.prepare(s);                                                  // :99
        ircache_state.trace_irradiance(scene, convolved_sky_cube.p, 16, s);                                     // :113-121
        const KjGbufferDepth gd = gbuffer_depth.view();
        check(kj_trace_sun_shadow_mask(device.h, scene.h, &gd, sun_shadow_mask.p, nullptr, s), "kj_trace_sun_shadow_mask");   // :123-125
        o.denoised_shadow_mask = shadow_denoise.render(gbuffer_depth, sun_shadow_mask.p, o.reprojection_map, s);               // :131-136
        rtdgi.reproject(o.reprojection_map, W, H, s);                                                           // :129
        ircache_state.sum_up_irradiance_for_sampling(s);                                                        // :138-140
        o.rtdgi = rtdgi.render(gbuffer_depth, o.reprojection_map, convolved_sky_cube.p, 16, scene, &ircache_state, o.ssgi_tex, s);   // :145-163
        TracedRtr traced = rtr.trace(gbuffer_depth, o.reprojection_map, sky_cube.p, 64, scene, o.rtdgi, &ircache_state, s);        // :172-188
        if (frame_state.triangle_light_count > 0) lighting.render_specular(traced, s);                          // :164-203 (any_triangle_lights)
        o.rtr = traced.filter_temporal(s);                                                                      // :205
        check(kj_light_gbuffer(device.h, &gd, o.denoised_shadow_mask, 1, o.rtr, o.rtdgi.screen_irradiance_tex, sky_cube.p, 64, accum_img.p, debug_out_tex.p,
                               debug_shading_mode, s), "kj_light_gbuffer");                                      // :219-234
        o.lit = debug_out_tex.p;
        o.anti_aliased = taa.render(debug_out_tex.p, W, H, o.reprojection_map, gbuffer_depth.depth.p, temporal_upscale_extent, s);   // :254-263
        if (post) {
            o.final_post_input = motion_blur->render(o.anti_aliased.this_frame_out, temporal_upscale_extent, gbuffer_depth.depth.p, o.reprojection_map, render_extent, s);   // :265-266
            o.post_processed = post->render(o.final_post_input, temporal_upscale_extent[0], temporal_upscale_extent[1],
                                            exposure_state[0].post_mult, contrast, dynamic_exposure.histogram_clipping, s);                                          // :281-289
        }
        frame_state.retire_frame();
        return o;
    }
    // WorldRenderer::prepare_render_graph_reference (world_render_passes.rs:294-330): one more path-traced sample per pixel into the persistent
    // "refpt.accum" image (cleared when reset_reference_accumulation is set, e.g. after the camera moved). Returns the RGBA32F accumulator,
    // or — once enable_post() has been called — what the reference returns: post.render on the accumulator (:320-329), B10G11R11_UFLOAT.
    const void* prepare_render_graph_reference(const CameraMatrices& camera, hipStream_t s) {
        if (post) update_pre_exposure(RenderMode::Reference);
        const size_t bytes = size_t(render_extent[0]) * render_extent[1] * 16;
        if (refpt_accum.bytes != bytes) { refpt_accum.alloc(bytes); reset_reference_accumulation = false; }
        frame_state.triangle_light_count = scene.triangle_light_count();
        const KjFrameConstants fc = frame_state.prepare_frame_constants(camera, nullptr);
        check(kj_frame_begin(device.h, &fc, s), "kj_frame_begin");
        if (reset_reference_accumulation) {
            reset_reference_accumulation = false;
            check_hip(hipMemsetAsync(refpt_accum.p, 0, bytes, s), "hipMemsetAsync");
        }
        check(kj_reference_path_trace(device.h, scene.h, refpt_accum.p, render_extent[0], render_extent[1], 0, 1, 0, nullptr, s), "kj_reference_path_trace");
        const void* out = refpt_accum.p;
        if (post)
            out = post->render(refpt_accum.p, render_extent[0], render_extent[1], exposure_state[1].post_mult, contrast, dynamic_exposure.histogram_clipping, s, KJ_POST_INPUT_RGBA32F);
        frame_state.retire_frame();
        return out;
    }
};

// ---------------------------------------------------------------- the screen-tile split across the GPUs of a node (no counterpart in the reference: kajiya renders on one
// GPU). One process per GPU; this process's rank with its renderers and an ncclComm_t made from kj_split_rccl_unique_id / kj_split_rccl_comm_create.
// Per frame, in world_render_passes.rs order: ssgi_frame, shadow_frame, gi_frame, rtr_frame (after set_rtr once), kj_light_gbuffer_rows on strip(), taa_frame.
struct ScreenTileSplit {
    KjSplit* h = nullptr;
    uint32_t rank;
    ScreenTileSplit(uint32_t world, uint32_t rank_, RtdgiRenderer& rtdgi, TaaRenderer& taa, IrcacheRenderer* ircache, Scene& scene, uint32_t width, uint32_t height, uint32_t motion_halo,
                    void* nccl_comm) : rank(rank_) {
        KjSplitRank me;
        std::memset(&me, 0, sizeof(me));
        me.rtdgi = rtdgi.h; me.taa = taa.h; me.ircache = ircache ? ircache->h : nullptr; me.scene = scene.h;
        check(kj_split_create(world, rank_, 1, &me, width, height, motion_halo, nccl_comm, &h), "kj_split_create");
    }
    ~ScreenTileSplit() { kj_split_destroy(h); }
    ScreenTileSplit(const ScreenTileSplit&) = delete;
    ScreenTileSplit& operator=(const ScreenTileSplit&) = delete;
    std::array<uint32_t, 2> strip() const { std::array<uint32_t, 2> r{}; check(kj_split_strip(h, rank, &r[0], &r[1]), "kj_split_strip"); return r; }
    bool self_test(hipStream_t s) { uint32_t ok = 0; check(kj_split_self_test(h, &ok, s), "kj_split_self_test"); return ok != 0; }      // combine over the ranks by the host's own means
    const void* ssgi_frame(SsgiRenderer& ssgi, const KjSplitFrame& f, hipStream_t s) { KjSsgi* g = ssgi.h; const void* out = nullptr; check(kj_split_ssgi_frame(h, &g, &f, &out, s), "kj_split_ssgi_frame"); return out; }
    const void* shadow_frame(ShadowDenoiseRenderer& dn, const KjSplitFrame& f, void* mask_r8, hipStream_t s) {
        KjShadowDenoise* d = dn.h; const void* out = nullptr;
        check(kj_split_shadow_frame(h, &d, &f, &mask_r8, nullptr, &out, s), "kj_split_shadow_frame"); return out;
    }
    void gi_frame(const KjSplitFrame& f, uint32_t flags, hipEvent_t trace_done, hipStream_t s) { check(kj_split_gi_frame(h, &f, flags, trace_done, s), "kj_split_gi_frame"); }
    void set_rtr(bool enable) { check(kj_split_set_rtr(h, enable ? 1u : 0u), "kj_split_set_rtr"); }
    const void* rtr_frame(RtrRenderer& rtr, const KjRtrParams& p, uint32_t flags, hipEvent_t trace_done, hipStream_t s) {
        KjRtr* r = rtr.h; const void* out = nullptr;
        check(kj_split_rtr_frame(h, &r, &p, flags, trace_done, &out, s), "kj_split_rtr_frame"); return out;
    }
    void merge_ircache(hipStream_t s) { check(kj_split_merge_ircache(h, s), "kj_split_merge_ircache"); }
    void taa_frame(const KjSplitFrame& f, void* lit_rgba16f, hipStream_t s) {      // lit_rgba16f: the image to resolve (valid on strip()), or nullptr: the GI image
        if (lit_rgba16f) check(kj_split_taa_frame_on(h, &f, &lit_rgba16f, s), "kj_split_taa_frame_on"); else check(kj_split_taa_frame(h, &f, s), "kj_split_taa_frame");
    }
};

}  // namespace kajiya_amd
