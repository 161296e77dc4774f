/*
 * kajiya_amd.h — C-ABI drop-in boundary for the MI355X-native ReSTIR-GI hot path.
 *
 * Every entry point replaces one pass-level method of the reference's Rust
 * renderer (`/root/reference/crates/lib/kajiya/src/...`, cited per function).
 * Plain pointers and sizes only: no C++ / torch types cross this boundary.
 *
 * Conventions
 *   - Every function returns KjStatus (0 = OK). `kj_last_error()` returns a
 *     thread-local message for the last failure. Nothing unwinds.
 *   - All image arguments are DEVICE pointers to linear, row-major, tightly
 *     packed surfaces in the texel format named in the comment (the formats are
 *     the reference's Vulkan formats; they are part of the algorithm).
 *   - All launches are asynchronous on the caller's `stream` (a hipStream_t
 *     passed as void*); no host synchronisation happens inside render calls.
 *   - Temporal (ping-pong) state lives inside the Kj* handles, mirroring
 *     `PingPongTemporalResource` (renderers/mod.rs:73-103).
 */
#ifndef KAJIYA_AMD_H
#define KAJIYA_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t KjStatus;
enum {
    KJ_OK = 0,
    KJ_ERR_INVALID_ARGUMENT = 1,
    KJ_ERR_HIP = 2,
    KJ_ERR_OUT_OF_MEMORY = 3,
    KJ_ERR_NOT_COMMITTED = 4,
    KJ_ERR_UNSUPPORTED = 5
};

/* ------------------------------------------------------------------------ */
/* Frame constants: bit-compatible with the reference's 1216-byte UBO        */
/* (rust-shaders-shared/src/frame_constants.rs:13-37, view_constants.rs:4-23,*/
/*  assets/shaders/inc/frame_constants.hlsl:8-82). Matrices are column-major */
/* (glam::Mat4 memory order): m[c*4 + r].                                    */
/* ------------------------------------------------------------------------ */
typedef struct KjViewConstants {
    float view_to_clip[16];
    float clip_to_view[16];
    float view_to_sample[16];
    float sample_to_view[16];
    float world_to_view[16];
    float view_to_world[16];
    float clip_to_prev_clip[16];
    float prev_view_to_prev_clip[16];
    float prev_clip_to_prev_view[16];
    float prev_world_to_prev_view[16];
    float prev_view_to_prev_world[16];
    float sample_offset_pixels[2];
    float sample_offset_clip[2];
} KjViewConstants;

typedef struct KjIrcacheCascadeConstants {
    int32_t origin[4];
    int32_t voxels_scrolled_this_frame[4];
} KjIrcacheCascadeConstants;

typedef struct KjRenderOverrides {
    uint32_t flags;
    float material_roughness_scale;
    uint32_t pad0, pad1;
} KjRenderOverrides;

enum {
    KJ_OVERRIDE_FORCE_FACE_NORMALS = 1u << 0,
    KJ_OVERRIDE_NO_NORMAL_MAPS = 1u << 1,
    KJ_OVERRIDE_FLIP_NORMAL_MAP_YZ = 1u << 2,
    KJ_OVERRIDE_NO_METAL = 1u << 3
};

#define KJ_IRCACHE_CASCADE_COUNT 12

typedef struct KjFrameConstants {
    KjViewConstants view_constants;
    float sun_direction[4];
    uint32_t frame_index;
    float delta_time_seconds;
    float sun_angular_radius_cos;
    uint32_t triangle_light_count;
    float sun_color_multiplier[4];
    float sky_ambient[4];
    float pre_exposure;
    float pre_exposure_prev;
    float pre_exposure_delta;
    float pad0;
    KjRenderOverrides render_overrides;
    float ircache_grid_center[4];
    KjIrcacheCascadeConstants ircache_cascades[KJ_IRCACHE_CASCADE_COUNT];
} KjFrameConstants; /* sizeof == 1216 */

/* ------------------------------------------------------------------------ */
/* Scene data: the reference's packed mesh layout                            */
/* (kajiya-asset/src/mesh.rs:75-84,447-459; world_renderer.rs:43-54,604-776) */
/* ------------------------------------------------------------------------ */
typedef struct KjPackedVertex {
    float pos[3];
    uint32_t normal; /* 11:10:11 unorm of n*0.5+0.5, x in the low bits */
} KjPackedVertex;

typedef struct KjMeshMaterial { /* 152 bytes, inc/mesh.hlsl:49-59; kajiya-asset mesh.rs:75-84 */
    float base_color_mult[4];
    uint32_t maps[4]; /* normal, spec, albedo, emissive — indices into KjMeshDesc.maps */
    float roughness_mult;
    float metalness_factor;
    float emissive[3];
    uint32_t flags;
    float map_transforms[24];
} KjMeshMaterial;

#define KJ_MESH_MATERIAL_FLAG_EMISSIVE_USED_AS_LIGHT 1u

/* A material map: the reference's `MeshMaterialMap::Placeholder` (a 1x1 RGBA8 image, mesh.rs:60-66) or an RGBA8 image with
 * its baked mip chain (`MeshMaterialMap::Image`, TexParams at mesh.rs:161-230: albedo / emissive are sRGB, spec is linear with
 * x = perceptual roughness and y = metalness after the baker's channel swizzle). The asset baker (Lanczos mips, BC5/BC7
 * compression, kajiya-asset/src/image.rs) is out of scope: the caller hands over decoded RGBA8 texels for every level. */
typedef struct KjMaterialMap {
    uint8_t placeholder_rgba[4];
    const uint8_t* image_rgba8; /* NULL => placeholder; else all mip levels back to back, level k = max(1,w>>k) x max(1,h>>k) x 4 bytes (host) */
    uint32_t width, height;
    uint32_t mip_count;         /* >= 1 for images */
    uint32_t srgb;              /* 1: R8G8B8A8_SRGB (rgb decoded per texel before filtering), 0: R8G8B8A8_UNORM */
} KjMaterialMap;

typedef struct KjMeshDesc {
    const KjPackedVertex* verts;
    uint32_t vertex_count;
    const float* uvs;            /* 2 per vertex, or NULL (zeros) */
    const float* tangents;       /* 4 per vertex, or NULL */
    const float* colors;         /* 4 per vertex, or NULL (mesh.vertex_aux_offset == 0) */
    const uint32_t* material_ids;/* 1 per vertex, or NULL (zeros) */
    const uint32_t* indices;
    uint32_t index_count;
    const KjMeshMaterial* materials;
    uint32_t material_count;
    const KjMaterialMap* maps;
    uint32_t map_count;
    uint32_t use_lights;         /* AddMeshOptions::use_lights, world_renderer.rs:325-339 */
} KjMeshDesc;

typedef struct KjTriangleLight { /* 48 bytes: 3 verts + radiance (world_renderer.rs:107-112) */
    float verts[9];
    float radiance[3];
} KjTriangleLight;

typedef struct KjDevice KjDevice;
typedef struct KjScene KjScene;
typedef struct KjRtdgi KjRtdgi;
typedef struct KjIrcache KjIrcache;
typedef struct KjTaa KjTaa;
typedef struct KjReprojection KjReprojection;

const char* kj_last_error(void);
uint32_t kj_abi_version(void);
/* sizeof() of the library's own build of every struct that crosses this boundary by value or by pointer, so that a binding in another
 * language can check its layout at start-up (tests/test_abi.py does, for the ctypes binding and for INTEGRATION.md's `-sys` text).
 * Returns 0 for an unknown id. */
enum KjAbiStruct { KJ_ABI_FRAME_CONSTANTS = 0, KJ_ABI_VIEW_CONSTANTS, KJ_ABI_MESH_MATERIAL, KJ_ABI_PACKED_VERTEX, KJ_ABI_MATERIAL_MAP, KJ_ABI_MESH_DESC, KJ_ABI_TRIANGLE_LIGHT,
                   KJ_ABI_GBUFFER_DEPTH, KJ_ABI_RTDGI_RENDER_PARAMS, KJ_ABI_RTDGI_OUTPUT, KJ_ABI_TAA_OUTPUT, KJ_ABI_RTR_TABLES, KJ_ABI_RTR_PARAMS, KJ_ABI_SPLIT_RANK,
                   KJ_ABI_SPLIT_FRAME, KJ_ABI_BAKED_MESH_VIEW, KJ_ABI_BAKED_IMAGE_VIEW, KJ_ABI_SPLIT_PROFILE, KJ_ABI_STRUCT_COUNT };
uint32_t kj_abi_struct_size(uint32_t id);
/* Device self test (no reference counterpart): the cheap exact division / square root the screen passes use (csrc/kj_screen.hpp: div_nr, sqrt_nr) against the
 * IEEE operations over `n` pseudo-random operand pairs (odd `seed`s draw TAA's own operand classes). `counts4_u64_device`: four uint64 on the device, ADDED to:
 * quotients whose bits differ, roots whose bits differ, quotients off by more than an ulp, roots off by more than an ulp. scripts/selftest_div_sqrt_nr.py. */
KjStatus kj_selftest_div_sqrt_nr(uint32_t n, uint32_t seed, void* counts4_u64_device, void* stream);
/* Device self test (no reference counterpart): the leaf functions of the device headers (hashes, pack / unpack family, quasi-random sequences, basis and samplers,
 * colour transforms, Reservoir1spp's methods, the specular and diffuse lobes) on `n` inputs (uint4 each, device), one output ROW of n uint4 per function group, in the
 * row order of oracle/ref_hlsl/probes/inc_functions.hlsl -- the probe that runs the reference's own inc/ headers on the same inputs (csrc/probe.hip;
 * tests/test_gpu_parity.py compares the rows). `out4_device` holds rows_capacity x n uint4; *out_rows = rows written. */
KjStatus kj_selftest_probe_functions(const void* in4_device, uint32_t n, void* out4_device, uint32_t rows_capacity, uint32_t* out_rows, void* stream);
/* The same for the second probe (oracle/ref_hlsl/probes/inc_functions_color.hlsl): the display transform's colour science incl. the transform itself, the G-buffer record,
 * soft_color_clamp, the uv helpers, the sky model. `bezold_brucke_lut_rg16f_device`: the 64-texel RG16F table (256 bytes) on the device. */
KjStatus kj_selftest_probe_functions_color(const void* in4_device, uint32_t n, const void* bezold_brucke_lut_rg16f_device, void* out4_device, uint32_t rows_capacity,
                                           uint32_t* out_rows, void* stream);
/* The same for the third probe (oracle/ref_hlsl/probes/inc_functions_shading.hlsl): the view-ray helpers under `frame_constants` (host pointer), the ray cone, the layered
 * BRDF and its energy preservation off the device's BRDF table (`brdf_fg_lut_rgba16f_device`: 64 x 64 RGBA16F, e.g. kj_device_brdf_lut), the sun, atmosphere_default, the
 * triangle-light sampler. */
KjStatus kj_selftest_probe_functions_shading(const KjFrameConstants* frame_constants, const void* in4_device, uint32_t n, const void* brdf_fg_lut_rgba16f_device,
                                             void* out4_device, uint32_t rows_capacity, uint32_t* out_rows, void* stream);
/* The same for the fourth probe (oracle/ref_hlsl/probes/inc_functions_misc.hlsl): TemporalReservoirOutput, the irradiance cache's sample parameters and its grid
 * addressing (ws_pos_to_ircache_coord under `frame_constants`' cascades). */
KjStatus kj_selftest_probe_functions_misc(const KjFrameConstants* frame_constants, const void* in4_device, uint32_t n, void* out4_device, uint32_t rows_capacity,
                                          uint32_t* out_rows, void* stream);

/* RenderBackend / WorldRenderer::new analogue (default_world_renderer.rs:14-58):
 * picks the HIP device, builds the BRDF-FG LUT (bindless #0, lut/brdf_fg.hlsl),
 * and takes the 256x256 RGBA8 blue-noise table (bindless #1, host pointer,
 * 262144 bytes). */
KjStatus kj_device_create(int32_t hip_device_ordinal, const uint8_t* blue_noise_rgba8_256, KjDevice** out);
void kj_device_destroy(KjDevice* dev);
/* Device pointer to the 64x64 RGBA16F BRDF FG LUT (for tests). */
KjStatus kj_device_brdf_lut(KjDevice* dev, const void** out_dev_ptr);

/* WorldRenderer::{add_mesh, add_instance, set_instance_transform, remove_instance}
 * (world_renderer.rs:604,778,800,815). Transforms are row-major 3x4 affine. */
KjStatus kj_scene_create(KjDevice* dev, KjScene** out);
void kj_scene_destroy(KjScene* scene);
KjStatus kj_scene_add_mesh(KjScene* scene, const KjMeshDesc* desc, uint32_t* out_mesh);
KjStatus kj_scene_add_instance(KjScene* scene, uint32_t mesh, const float transform3x4[12], uint32_t* out_instance);
KjStatus kj_scene_set_instance_transform(KjScene* scene, uint32_t instance, const float transform3x4[12]);
KjStatus kj_scene_set_instance_emissive_multiplier(KjScene* scene, uint32_t instance, float v);
KjStatus kj_scene_remove_instance(KjScene* scene, uint32_t instance);
/* build_ray_tracing_top_level_acceleration + prepare_top_level_acceleration (world_renderer.rs:836,865) and the BLAS builds of
 * add_mesh (:694-724): builds the BLAS of every mesh added since the last commit, re-derives the world-space triangles and nodes of
 * the instances that moved (on the device), rebuilds the tree over the instances and uploads the scene tables. Stream-ordered, but
 * returns after the stream has drained. */
KjStatus kj_scene_commit(KjScene* scene, void* stream);
/* Number of world-space triangle lights after commit (frame_constants.triangle_light_count). */
KjStatus kj_scene_triangle_light_count(KjScene* scene, uint32_t* out);
KjStatus kj_scene_stats(KjScene* scene, uint32_t* out_tri_count, uint32_t* out_node_count, uint64_t* out_bvh_bytes);
/* Host time of the last kj_scene_commit in ms: [0] BLAS builds of newly added meshes, [1] instance tables + top-tree build,
 * [2] uploads + the device kernels that re-derive moved instances' world-space triangles and nodes (incl. the stream sync), [3] total.
 * The reference's counterpart is the GPU time of build_ray_tracing_top_level_acceleration (world_renderer.rs:836-911). */
KjStatus kj_scene_last_commit_ms(KjScene* scene, double out_ms[4]);
/* How the BLAS of meshes added FROM NOW ON is built at the next commit (vk::BuildAccelerationStructureFlagsKHR, ray_tracing.rs:438):
 * KJ_BLAS_BUILD_FAST_TRACE = binned-SAH tree built on the host (kajiya's PREFER_FAST_TRACE; the default),
 * KJ_BLAS_BUILD_FAST_BUILD = linear BVH built on the device (Morton sort + Karras hierarchy + bottom-up boxes + 4-wide collapse),
 * KJ_BLAS_BUILD_DEVICE_PLOC = built on the device as well, the hierarchy by bottom-up agglomerative clustering over the Morton order
 *   (PLOC) with SAH leaf selection: trace rates close to the host SAH trees at a device build's cost (no Vulkan counterpart: the
 *   driver's PREFER_FAST_TRACE build also runs on the GPU). */
#define KJ_BLAS_BUILD_FAST_TRACE 0u
#define KJ_BLAS_BUILD_FAST_BUILD 1u
#define KJ_BLAS_BUILD_DEVICE_PLOC 2u
KjStatus kj_scene_set_blas_build_mode(KjScene* scene, uint32_t mode);
/* Top-tree granularity from the next commit on. 0 (default): one top-tree leaf per instance, as a driver's TLAS (ray_tracing.rs:277-407). 1: the
 * largest nodes of the instances' top levels become the leaves (opened greedily by world-space surface area, 4 per instance slot on average):
 * large instances that overlap many others -- a terrain -- stop being one box around everything. Same hits either way. */
KjStatus kj_scene_set_open_instances(KjScene* scene, uint32_t enable);
/* Who builds the per-commit top tree (the TLAS build of WorldRenderer::build_ray_tracing_top_level_acceleration, world_renderer.rs:836-911): the host
 * (binned SAH: the better tree, 0.06 ms per commit at 64 instances but 2.6 ms at 4 k and 24 ms at 32 k) or the device (a linear BVH over the instances' world boxes:
 * 0.2-0.5 ms up to 4 k, 1.7 ms at 32 k; closest-hit rays within 2 % of the host tree's rate, measured). KJ_TOP_BUILD_AUTO: the host below 1024 top-tree leaves, the device from there on. */
enum { KJ_TOP_BUILD_AUTO = 0u, KJ_TOP_BUILD_HOST = 1u, KJ_TOP_BUILD_DEVICE = 2u };
KjStatus kj_scene_set_top_build_mode(KjScene* scene, uint32_t mode);
/* The last commit's top tree: its nodes, the reservation at the head of the world node array, and who built it (1 = the device). */
KjStatus kj_scene_top_tree_info(KjScene* scene, uint32_t* out_nodes, uint32_t* out_capacity, uint32_t* out_built_on_device);

/* Baked assets (`bin/bake` output, kajiya-asset-pipe/src/lib.rs:38-60): zero-copy, bounds-checked views of
 * `cache/<name>.mesh` (PackedTriMesh::Flat, kajiya-asset/src/mesh.rs:796-807) and `cache/<identity:08x>.image`
 * (GpuImage::Flat, mesh.rs:787-793) — the FlatVec layout of mesh.rs:460-632 — as WorldRenderer::add_mesh and
 * load_gpu_image_asset read them (world_renderer.rs:297-322,604-700). Pointers alias the caller's bytes (host). The stream
 * pointers drop straight into KjMeshDesc; `map_identities[k]` names the image file of the mesh's k-th material map. */
typedef struct KjBakedMeshView {
    const KjPackedVertex* verts;
    const float* uvs;            /* NULL when the vector is empty */
    const float* tangents;
    const float* colors;
    const uint32_t* indices;
    const uint32_t* material_ids;
    const KjMeshMaterial* materials;
    const uint64_t* map_identities; /* ONLY 4-BYTE ALIGNED inside the packed file (FlatVec has no padding): read entries with memcpy, never dereference */
    uint32_t vertex_count, index_count, material_count, map_count;
} KjBakedMeshView;
typedef struct KjBakedImageView {
    uint32_t vk_format;          /* ash::vk::Format: 37/43 RGBA8 unorm/srgb, 131-134 BC1, 137/138 BC3, 139 BC4, 141 BC5, 145/146 BC7 */
    uint32_t extent[3];
    uint32_t mip_count;
} KjBakedImageView;
KjStatus kj_baked_mesh_view(const void* bytes, uint64_t size, KjBakedMeshView* out);
KjStatus kj_baked_image_view(const void* bytes, uint64_t size, KjBakedImageView* out);
KjStatus kj_baked_image_mip(const void* bytes, uint64_t size, uint32_t level, const uint8_t** out_data, uint64_t* out_len);
/* Texel decode of one mip level to RGBA8 for KjMaterialMap (the reference leaves this to the texture unit): RGBA8, BC1, BC3,
 * BC4 (r,0,0,1), BC5 (r,g,0,1), BC7 — every format kajiya's baker emits (kajiya-asset/src/image.rs:131-336). Other formats return
 * KJ_ERR_UNSUPPORTED. */
KjStatus kj_baked_image_decode_rgba8(uint32_t vk_format, const uint8_t* mip_data, uint64_t mip_len, uint32_t width, uint32_t height, uint8_t* out_rgba8);

/* prepare_frame_constants (world_renderer.rs:1001-1108): upload this frame's UBO. */
KjStatus kj_frame_begin(KjDevice* dev, const KjFrameConstants* fc, void* stream);

/* Ray "ISA" exposed for tests and for other subsystems (inc/rt.hlsl:58-137):
 * rays = float4 origin+tmin, float4 dir+tmax (32 B/ray); hits = {t, u, v, prim}
 * with t = FLT_MAX on miss; any-hit variant writes 1 byte per ray. */
KjStatus kj_trace_closest(KjScene* scene, const void* rays, void* hits, uint32_t count, uint32_t cull_back_faces, void* stream);
KjStatus kj_trace_any(KjScene* scene, const void* rays, void* out_u8, uint32_t count, void* stream);
/* Measurement aid, no reference counterpart: copies `bytes` (multiple of 16) device-to-device with a plain 16-B-per-lane kernel named
 * `pmc_calibration_copy`, a known byte count against which rocprofv3's FETCH_SIZE / WRITE_SIZE are calibrated (scripts/pmc_collect.sh). */
KjStatus kj_debug_calibration_copy(void* dst, const void* src, uint64_t bytes, void* stream);

/* G-buffer stand-in for raster_meshes (renderers/raster_meshes.rs; packing as in
 * raster_simple_ps.hlsl:126-137): primary rays through the jittered camera.
 * Outputs: geometric_normal A2R10G10B10_UNORM (u32), gbuffer RGBA32F-as-uint4,
 * depth R32F (reverse-Z, 0 = sky), velocity RGBA16F. Test/bench input generator. */
KjStatus kj_raster_gbuffer(KjDevice* dev, KjScene* scene, uint32_t width, uint32_t height,
                           void* geometric_normal, void* gbuffer, void* depth, void* velocity, void* stream);

/* sky::render_sky_cube / sky::convolve_cube (renderers/sky.rs:4-36): RGBA16F cubes,
 * 6 faces x width x width. */
KjStatus kj_sky_cube_render(KjDevice* dev, void* out_cube64, void* stream);
KjStatus kj_sky_cube_convolve(KjDevice* dev, const void* cube64, void* out_cube16, void* stream);

typedef struct KjGbufferDepth { /* renderers/mod.rs:31-71 */
    const void* geometric_normal; /* A2R10G10B10_UNORM_PACK32 */
    const void* gbuffer;          /* R32G32B32A32 (4 packed dwords, inc/gbuffer.hlsl:51-64) */
    const void* depth;            /* R32F */
    uint32_t width, height;
} KjGbufferDepth;

/* calculate_reprojection_map (renderers/reprojection.rs:6-52). The handle owns the
 * `reprojection.prev_depth` temporal. Output RGBA16_SNORM, full res. */
KjStatus kj_reprojection_create(KjDevice* dev, KjReprojection** out);
void kj_reprojection_destroy(KjReprojection* r);
KjStatus kj_calculate_reprojection_map(KjReprojection* r, const KjGbufferDepth* gbuffer_depth,
                                       const void* velocity, const void** out_reprojection_map, void* stream);

/* RtdgiRenderer (renderers/rtdgi.rs:11-46). */
KjStatus kj_rtdgi_create(KjDevice* dev, KjRtdgi** out);
void kj_rtdgi_destroy(KjRtdgi* r);
KjStatus kj_rtdgi_set_options(KjRtdgi* r, uint32_t spatial_reuse_pass_count, uint32_t use_raytraced_reservoir_visibility);

/* RtdgiRenderer::reproject (rtdgi.rs:143-170). */
KjStatus kj_rtdgi_reproject(KjRtdgi* r, const void* reprojection_map, uint32_t width, uint32_t height, void* stream);
/* The same for full-res rows [row_begin, row_end) only (row_begin a multiple of 8): the screen-tile split reprojects strip by strip and all-gathers
 * `reprojected_history_tex` (kj_rtdgi_surface), which the trace pass reads anywhere on screen. Reads the history within motion + 2 rows of the range. */
KjStatus kj_rtdgi_reproject_rows(KjRtdgi* r, const void* reprojection_map, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, void* stream);

/* Pass bits for kj_rtdgi_render's pass_mask (data-flow order, rtdgi.rs:189-553). */
enum {
    KJ_RTDGI_PASS_EXTRACT_HALF = 1u << 0,   /* extract ssao/2, view normal/2, depth/2 */
    KJ_RTDGI_PASS_VALIDATE = 1u << 1,
    KJ_RTDGI_PASS_TRACE = 1u << 2,
    KJ_RTDGI_PASS_VALIDITY_INTEGRATE = 1u << 3,
    KJ_RTDGI_PASS_RESTIR_TEMPORAL = 1u << 4,
    KJ_RTDGI_PASS_RESTIR_SPATIAL = 1u << 5,
    KJ_RTDGI_PASS_RESTIR_RESOLVE = 1u << 6,
    KJ_RTDGI_PASS_TEMPORAL_FILTER = 1u << 7,
    KJ_RTDGI_PASS_SPATIAL_FILTER = 1u << 8,
    KJ_RTDGI_PASS_ALL = 0x1ffu,
    /* Scheduling aids for a host that computes the SSAO guide on another stream while the ray passes run (nothing before `restir spatial`
     * reads the half-res SSAO): with _NO_SSAO the extract pass leaves the SSAO out; _SSAO_ONLY (its own call, after the guide is done and
     * before `restir spatial`) adds it. Together they write exactly what KJ_RTDGI_PASS_EXTRACT_HALF alone writes. */
    KJ_RTDGI_PASS_EXTRACT_HALF_NO_SSAO = 1u << 9,
    KJ_RTDGI_PASS_EXTRACT_HALF_SSAO_ONLY = 1u << 10,
    /* For a host that has to do something between `rtdgi validate` and the LAST statement of `rtdgi trace` (trace_diffuse.rgen.hlsl:119 reads the validate pass' output at the
     * reprojected pixel: in the screen-tile split that is a halo exchange; nothing else of the trace pass reads what validate writes). A call with VALIDATE | TRACE | _TRACE_MAY_DEFER
     * runs the trace pass without that statement where the library has the two passes as one launch (validation frames), and leaves the trace pass out altogether otherwise;
     * a later call with _TRACE_FINISH alone runs what is left (the statement, or the whole pass). Same images as VALIDATE then TRACE. */
    KJ_RTDGI_PASS_TRACE_MAY_DEFER = 1u << 11,
    KJ_RTDGI_PASS_TRACE_FINISH = 1u << 12,
    /* Debug: re-use the previous call's ping-pong assignment instead of advancing it, so a frame can be
     * executed pass by pass (render-graph debug hook analogue). Never set on the product path. */
    KJ_RTDGI_PASS_KEEP_TEMPORALS = 1u << 31
};

typedef struct KjRtdgiRenderParams {
    KjGbufferDepth gbuffer_depth;
    const void* reprojection_map;  /* RGBA16_SNORM full res */
    const void* sky_cube;          /* convolved 6x16x16 RGBA16F (world_render_passes.rs:150) */
    uint32_t sky_cube_width;
    KjScene* scene;                /* tlas + bindless set */
    KjIrcache* ircache;            /* &mut IrcacheRenderState; NULL => lookups return 0 (BASELINE config 1) */
    const void* ssao_tex;          /* R8_UNORM full res */
    uint32_t pass_mask;            /* KJ_RTDGI_PASS_ALL for the product path */
    /* Screen-tile split (multi-GPU): process only full-res rows [row_begin, row_end) (16-aligned; half-res
     * passes use [row_begin/2, row_end/2)). 0,0 = whole image. Reads reach outside the range; the caller
     * exchanges halos between passes (kajiya_amd/multigpu.py). */
    uint32_t row_begin, row_end;
    uint32_t spatial_pass_select;  /* 0 = run every spatial reuse pass; k>0 = only pass k-1 (halo exchange between passes) */
} KjRtdgiRenderParams;

typedef struct KjRtdgiOutput { /* RtdgiOutput / RtdgiCandidates, rtdgi.rs:53-62 */
    const void* screen_irradiance_tex;  /* RGBA16F full res */
    const void* candidate_radiance_tex; /* RGBA16F half res */
    const void* candidate_normal_tex;   /* RGBA8_SNORM half res */
    const void* candidate_hit_tex;      /* RGBA16F half res */
} KjRtdgiOutput;

/* RtdgiRenderer::render (rtdgi.rs:173-554). */
KjStatus kj_rtdgi_render(KjRtdgi* r, const KjRtdgiRenderParams* params, KjRtdgiOutput* out, void* stream);

/* Render-graph debug hook analogue (kajiya-rg/src/graph.rs:124-145): access any
 * named surface of the renderer (names are the reference's temporal keys /
 * variable names, e.g. "rtdgi.radiance:history", "candidate_radiance_tex").
 * `kj_rtdgi_surface` returns the device pointer and byte size. */
KjStatus kj_rtdgi_surface(KjRtdgi* r, const char* name, void** out_dev_ptr, uint64_t* out_bytes);
/* Per-pass GPU timestamps (the reference scopes every render-graph pass with a timestamp query,
 * kajiya-rg/src/graph.rs:941-944) and instrumented traversal counters. Scope order:
 * reproject, extract, validate, trace, validity integrate, restir temporal, spatial 0, spatial 1,
 * resolve, temporal filter, spatial filter. `count_traversal` switches the trace kernels to the
 * instrumented build (nodes visited / triangles tested per ray type) — never timed. */
#define KJ_RTDGI_NUM_SCOPES 11
KjStatus kj_rtdgi_set_profiling(KjRtdgi* r, uint32_t enable_pass_timers, uint32_t count_traversal);
/* (`validity integrate` reads 0 while it shares `restir temporal`'s launch, the default; a frame that issues the half-res extract in two parts reports the first, the bulk, as `extract half`.) */
KjStatus kj_rtdgi_pass_times_ms(KjRtdgi* r, float* out_ms, uint32_t count);
/* out[6] = closest rays, any-hit rays, nodes (closest), tris (closest), nodes (any), tris (any) */
KjStatus kj_rtdgi_traversal_counts(KjRtdgi* r, uint64_t out[6]);
/* How the two ray-tracing passes (`rtdgi validate`, `rtdgi trace`: rtdgi.rs:283-365) are scheduled on the device. The outputs are
 * bit-identical for every form; the knob exists for A/B measurements in one process (kj_rtdgi_create takes its default from
 * KJ_RTDGI_GROUPED / KJ_RTDGI_STAGED_MIN_RAYS).
 *   KJ_RTDGI_RAYS_FUSED (default, the fastest measured: profiles/r03_ray_pass_forms.md): one wave per 8x8 tile runs ray generation, both
 *     traversals and hit shading (the ray-generation shader's shape)
 *   KJ_RTDGI_RAYS_GROUPED: 256-thread workgroups, hit shading regrouped onto full waves through LDS
 *   KJ_RTDGI_RAYS_STAGED: five launches over dense ray arrays (ray streams)
 *   KJ_RTDGI_RAYS_SPLIT: two launches: closest-hit traversal + everything a miss needs | hit shading on records compacted across tiles
 *   KJ_RTDGI_RAYS_QUAD: the fused kernels with four lanes per pixel (a wave covers 8 x 2 pixels; the traversal's four child tests of a node
 *     run on the four lanes, the shading code on all of them, the first lane stores)
 *   KJ_RTDGI_RAYS_POOL (round 5): a few persistent waves per SIMD; a lane holds one pixel's job (ray generation -> closest-hit walk -> hit shading ->
 *     shadow walk -> the rest of the shading), free lanes are refilled from the wave's tile list while its other rays still walk, closest-hit and
 *     occlusion rays share traversal steps, and the shading blocks wait for enough lanes (kj_rtdgi_set_pool_tune) */
enum { KJ_RTDGI_RAYS_GROUPED = 0, KJ_RTDGI_RAYS_FUSED = 1, KJ_RTDGI_RAYS_STAGED = 2, KJ_RTDGI_RAYS_SPLIT = 3, KJ_RTDGI_RAYS_QUAD = 4, KJ_RTDGI_RAYS_POOL = 5 };
KjStatus kj_rtdgi_set_ray_pass_form(KjRtdgi* r, uint32_t form);
/* Scheduling of the pool form: persistent waves per SIMD (1 .. 4), the lane counts that must be waiting before the refill / first / second shading
 * block is issued while other lanes still walk (1 .. 64), and whether tiles are handed out by a device counter (1) or by stride (0). Results do not depend on any of them. */
KjStatus kj_rtdgi_set_pool_tune(KjRtdgi* r, uint32_t waves_per_simd, uint32_t refill_min, uint32_t shade_a_min, uint32_t shade_b_min, uint32_t dynamic_tiles);
/* Ray counters of the last kj_rtdgi_render (closest-hit rays, any-hit rays). */
KjStatus kj_rtdgi_ray_counts(KjRtdgi* r, uint64_t* out_closest, uint64_t* out_any);

/* ------------------------------------------------------------------------ */
/* Irradiance cache: IrcacheRenderer / IrcacheRenderState (renderers/ircache.rs) */
/* ------------------------------------------------------------------------ */
KjStatus kj_ircache_create(KjDevice* dev, KjIrcache** out);
void kj_ircache_destroy(KjIrcache* c);
/* IrcacheRenderer::update_eye_position (ircache.rs:126-141) and ::constants/::grid_center (:143-162):
 * the host calls these while building the frame constants (world_renderer.rs:1061-1069). */
KjStatus kj_ircache_update_eye_position(KjIrcache* c, const float eye[3]);
KjStatus kj_ircache_constants(KjIrcache* c, KjFrameConstants* fc_inout);
KjStatus kj_ircache_set_enable_scroll(KjIrcache* c, uint32_t enable);
/* IrcacheRenderer::prepare (ircache.rs:168-350): [clear pool | scroll cascades], age, prefix scan, compact. */
KjStatus kj_ircache_prepare(KjIrcache* c, void* stream);
/* IrcacheRenderState::trace_irradiance (ircache.rs:360-481): dispatch args, reset, accessibility rays,
 * validation rays, irradiance rays. `sky_cube` is the convolved cube (world_render_passes.rs:113-121). */
KjStatus kj_ircache_trace_irradiance(KjIrcache* c, KjScene* scene, const void* sky_cube, uint32_t sky_cube_width, void* stream);
/* IrcacheRenderState::sum_up_irradiance_for_sampling (ircache.rs:487-506). */
KjStatus kj_ircache_sum_up_irradiance_for_sampling(KjIrcache* c, void* stream);
/* Debug access to the persistent buffers by the reference's names without the "ircache." prefix
 * ("meta", "grid_meta", "entry_cell", "spatial", "irradiance", "aux", "life", "pool",
 *  "entry_indirection", "reposition_proposal", "reposition_proposal_count"). */
/* Deferred updates -- the irradiance cache under a screen-tile split. The reference's lookups allocate cells, refresh
 * entry lives and vote for entry positions with atomics as they go (lookup.hlsl:118-150,287-301): replicas of the cache fed by
 * different strips drift apart. With deferred updates on, a lookup returns the same value but only RECORDS those effects (one 32-byte
 * KjIrcacheRequest per lookup, in a slot of its own). What the records of a frame do to the cache is defined as a REDUCTION over the
 * records of a cell (ircache.hip: one of the racy program's legal outcomes -- every lookup reads the entry's life before any lowers it;
 * the vote's winner is the voter with the smallest dart, accepted against the count before the frame's votes; an empty cell is allocated
 * by its lowest-positioned lookup that may allocate; new cells take pool entries in cell order), so a rank reduces its own strip's
 * records into a fixed-size SUMMARY (kj_ircache_summary_bytes: 4 MB), the summaries are all-gathered and every rank merges the same
 * ones: all replicas stay bit-identical -- and identical to one GPU running the same frame in this mode. No list lengths, no sort, no
 * host synchronisation. Per frame: begin_requests (one launch clears the frame's slots and both summaries), the frame's passes,
 * summarize_requests(strip ranges -> summary 0; the cache's own two ranges -> summary 1: kj_ircache_request_ranges gives rtdgi validate,
 * rtdgi trace -- rows of a strip are contiguous slots --, the cache's validate and trace rays), all-gather of summary 0, apply_summaries
 * (every rank's summary 0 in rank order, then the local summary 1). */
KjStatus kj_ircache_set_deferred_updates(KjIrcache* ircache, uint32_t enable);
/* Schedule of the three ray passes of kj_ircache_trace_irradiance in the racy (default, the reference's) mode. The reference records "ircache trace access",
 * "ircache validate" and "ircache trace" with `write_no_sync` on every buffer they share (ircache.rs:396-481; :411-412 "if we use `write_no_sync`, we can overlap
 * with the next pass"), i.e. without barriers: on a GPU they overlap as far as the hardware lets them. enable = 0: back to the library default (the chain, below;
 * rounds 1-4: three launches). enable = 1: one launch carrying the three passes side by side -- the cache's segment of the frame 0.40 -> 0.21 ms on
 * MI355X at 1080p, and the racy cache's distance from the sequential oracle on identical state 1.3e-2 -> 5e-2 .. 1.3e-1 of its SH sums. Ignored in the deterministic mode. */
KjStatus kj_ircache_set_ray_passes_side_by_side(KjIrcache* ircache, uint32_t enable);      /* deprecated: = kj_ircache_set_ray_pass_schedule(enable ? KJ_IRC_PASSES_SIDE_BY_SIDE : KJ_IRC_PASSES_CHAIN (the default)) */
/* The schedule of the three ray passes, in full (round 5):
 *   KJ_IRC_PASSES_CHAIN (default): ONE launch in which every aux slot still sees its own passes in recording order -- accessibility, validation, the new sample --
 *     while the three rays of a slot walk side by side (validation's and the new sample's path on two quads of the same wave; the slot's values go from one to the
 *     other through lane shuffles, as the packed values the sequential passes would have stored). Own-slot results are those of three launches; what a pass sees of
 *     OTHER entries (the lookups' reads) is unordered in the racy mode, as it is inside a pass, and in the deterministic mode it is the snapshot taken before the
 *     launch for validation and tracing alike (the oracle's rule too: okj_ircache_set_chain_schedule). Measured on MI355X at 1080p (profiles/r05/ircache_schedules_bench_lines.jsonl):
 *     the cache's segment 0.40 -> 0.30 ms (side by side: 0.21), the pipelined frame 1.01 -> 0.975 ms (side by side: 0.971), the racy cache's distance from the sequential oracle on
 *     identical state 1.3e-2 as with three launches (side by side: 1.3e-1).
 *   KJ_IRC_PASSES_SEQUENTIAL: three launches one after the other (rounds 1-4), a snapshot between validation and tracing in the deterministic mode.
 *   KJ_IRC_PASSES_SIDE_BY_SIDE: one launch, the passes racing on their shared slots (racy mode only; the deterministic mode runs the chain instead). */
enum { KJ_IRC_PASSES_SEQUENTIAL = 0, KJ_IRC_PASSES_SIDE_BY_SIDE = 1, KJ_IRC_PASSES_CHAIN = 2 };
KjStatus kj_ircache_set_ray_pass_schedule(KjIrcache* ircache, uint32_t schedule);
KjStatus kj_ircache_begin_requests(KjIrcache* ircache, uint32_t rtdgi_half_width, uint32_t rtdgi_half_height, void* stream);
/* For a caller whose per-pixel passes run on half-res rows [half_row_begin, half_row_end) only (a rank of the screen-tile split): clears those rows' slots and the
 * cache's own two ranges instead of the whole slot array (150 MB at 4K, 280 MB with reflections). Lookups recorded outside the rows would survive into next frame. */
KjStatus kj_ircache_begin_requests_rows(KjIrcache* ircache, uint32_t rtdgi_half_width, uint32_t rtdgi_half_height, uint32_t half_row_begin, uint32_t half_row_end, void* stream);
KjStatus kj_ircache_request_ranges(KjIrcache* ircache, uint32_t out_first_slot[4], uint32_t out_slot_count[4]);
/* A frame with reflections (kj_rtr_trace bound to this cache) records the lookups of rtr's validate and trace rays too: two more slot ranges of one slot
 * per half-res pixel behind the four above. Sticky; set before kj_ircache_begin_requests. kj_rtr_trace refuses a deferred cache without them. */
KjStatus kj_ircache_set_rtr_requests(KjIrcache* ircache, uint32_t enable);
KjStatus kj_ircache_rtr_request_ranges(KjIrcache* ircache, uint32_t out_first_slot[2], uint32_t out_slot_count[2]);
uint64_t kj_ircache_summary_bytes(void);
/* Reduces the records of up to 6 slot ranges into the cache's summary `which` (0: the part a rank of the split sends to the others; 1: the part that stays
 * local -- the cache's own ray passes, identical on every replica). Once per summary and frame. */
KjStatus kj_ircache_summarize_requests(KjIrcache* ircache, const uint32_t* first_slots, const uint32_t* slot_counts, uint32_t n_ranges, uint32_t which, void* stream);
KjStatus kj_ircache_summary(KjIrcache* ircache, uint32_t which, void** out_dev_ptr /* kj_ircache_summary_bytes() bytes */);
/* Merges `n` (<= 32) summaries -- device pointers, the same ones in the same order on every replica -- and applies the result to the cache. */
KjStatus kj_ircache_apply_summaries(KjIrcache* ircache, const void* const* summaries /* host array of device pointers */, uint32_t n, void* stream);
/* A plain device list of records replayed at once (reduce into summary 0 + apply): for callers that assemble lists themselves, and for tests. */
KjStatus kj_ircache_apply_requests(KjIrcache* ircache, const void* list /* device, 32 B each; cell 0xffffffff = unused */, uint32_t count, void* stream);
KjStatus kj_ircache_buffer(KjIrcache* c, const char* name, void** out_dev_ptr, uint64_t* out_bytes);
KjStatus kj_ircache_ray_counts(KjIrcache* c, uint64_t* out_closest, uint64_t* out_any);

/* ------------------------------------------------------------------------ */
/* TAA / temporal super-resolution: TaaRenderer (renderers/taa.rs:41-191)     */
/* ------------------------------------------------------------------------ */
typedef struct KjTaaOutput { /* TaaOutput, taa.rs:30-33 */
    const void* temporal_out;    /* RGBA16F output res, a = accumulated coverage ("taa" temporal) */
    const void* this_frame_out;  /* RGBA16F output res */
} KjTaaOutput;
KjStatus kj_taa_create(KjDevice* dev, KjTaa** out);
void kj_taa_destroy(KjTaa* t);
/* TaaRenderer::render(rg, input_tex, reprojection_map, depth_tex, output_extent). input_tex RGBA16F,
 * reprojection_map RGBA16_SNORM and depth R32F are at the input (render) extent. */
KjStatus kj_taa_render(KjTaa* t, const void* input_tex, uint32_t input_width, uint32_t input_height, const void* reprojection_map,
                       const void* depth_tex, uint32_t output_width, uint32_t output_height, KjTaaOutput* out, void* stream);
/* Strip / pass-by-pass variant used by the screen-tile split (kajiya_amd/multigpu.py). pass_mask bits 0..6 =
 * reproject, filter input, filter history, input prob, prob filter, prob filter2, taa; bit 31 = keep the
 * previous call's ping-pong assignment. Rows are 8-aligned and need input extent == output extent. */
KjStatus kj_taa_render_rows(KjTaa* t, const void* input_tex, uint32_t input_width, uint32_t input_height, const void* reprojection_map,
                            const void* depth_tex, uint32_t output_width, uint32_t output_height, KjTaaOutput* out, void* stream,
                            uint32_t pass_mask, uint32_t row_begin, uint32_t row_end);
KjStatus kj_taa_surface(KjTaa* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes);

/* trace_sun_shadow_mask(rg, &GbufferDepth, tlas, bindless_set) -> Handle<Image>   renderers/shadows.rs:10-40,
 * rt/trace_sun_shadow_mask.rgen.hlsl:19-60: one soft-shadow ray per full-res pixel towards a blue-noise sample of the sun
 * disc; out_mask_r8 = R8_UNORM (255 lit / 0 shadowed, 255 for sky). The shadow denoiser is kj_shadow_denoise_* below.
 * ray_counter_dev: optional device u64 that receives += rays traced. */
KjStatus kj_trace_sun_shadow_mask(KjDevice* dev, KjScene* scene, const KjGbufferDepth* gbuffer_depth, void* out_mask_r8,
                                  uint64_t* ray_counter_dev, void* stream);
/* Rows [row_begin, row_end) of the mask (row_begin a multiple of 8): one strip of the screen-tile split. Nothing outside the strip is read. */
KjStatus kj_trace_sun_shadow_mask_rows(KjDevice* dev, KjScene* scene, const KjGbufferDepth* gbuffer_depth, void* out_mask_r8, uint32_t row_begin, uint32_t row_end,
                                       uint64_t* ray_counter_dev, void* stream);

/* ShadowDenoiseRenderer::render(rg, &GbufferDepth, shadow_mask, reprojection_map) -> ReadOnlyHandle<Image>
 *   renderers/shadow_denoise.rs:19-148; shaders/shadow_denoise/{bitpack_shadow_mask,megakernel,spatial_filter}.hlsl over the
 *   FidelityFX shadow denoiser (the shadow_denoise/ffx shaders): bit-packed masks, temporal accumulation with moments and a
 *   17x17 local neighbourhood clamp, three edge-stopping a-trous passes (steps 1, 2, 4).
 * shadow_mask R8_UNORM (kj_trace_sun_shadow_mask); *out_rg16f = RG16F image owned by the handle (x = denoised shadow term,
 * y = variance), valid until the next call. kajiya runs it only when sun_size_multiplier > 0 (world_render_passes.rs:131-136). */
typedef struct KjShadowDenoise KjShadowDenoise;
KjStatus kj_shadow_denoise_create(KjDevice* dev, KjShadowDenoise** out);
void kj_shadow_denoise_destroy(KjShadowDenoise* s);
KjStatus kj_shadow_denoise_render(KjShadowDenoise* s, const KjGbufferDepth* gbuffer_depth, const void* shadow_mask_r8, const void* reprojection_map,
                                  const void** out_rg16f, void* stream);
/* The denoised term for rows [row_begin, row_end) only (row_begin a multiple of 16): the screen-tile split's form. The passes over-compute what the next one
 * reaches into (up to 24 rows either side), so the caller provides the mask on [row_begin - 32, row_end + 32) and the two histories
 * ("shadow_denoise_moments:<k>", "shadow_denoise_accum:<k>" through kj_shadow_denoise_surface) on the strip +- (motion reach + 26) rows; *out_rg16f is valid on
 * the strip's rows. kj_split_shadow_frame does exactly that. */
KjStatus kj_shadow_denoise_render_rows(KjShadowDenoise* s, const KjGbufferDepth* gbuffer_depth, const void* shadow_mask_r8, const void* reprojection_map,
                                       uint32_t row_begin, uint32_t row_end, const void** out_rg16f, void* stream);
KjStatus kj_shadow_denoise_surface(KjShadowDenoise* s, const char* name, void** out_dev_ptr, uint64_t* out_bytes);

/* light_gbuffer(rg, gbuffer_depth, shadow_mask, rtr, rtdgi, ircache, wrc, temporal_output, output, sky_cube, convolved_sky_cube,
 * bindless_set, debug_shading_mode, debug_show_wrc)   renderers/deferred.rs:6-60, shaders/light_gbuffer.hlsl:60-260 — the deferred
 * combine: sun light through the shadow mask + emissive + diffuse GI * albedo * transmission (+ specular when rtr_tex is given) and
 * the sky / sun disc where depth == 0. shadow_mask = the raw R8_UNORM mask of kj_trace_sun_shadow_mask or (shadow_mask_is_rg16f) the RG16F
 * image of kj_shadow_denoise_render; rtr_tex B10G11R11_UFLOAT (the image kj_rtr_filter_temporal returns) or NULL (black); rtdgi_tex RGBA16F; outputs RGBA16F
 * (`out_temporal` is the image kajiya keeps as next frame's prev_radiance). debug_shading_mode 0-4 as in the shader (5 = ircache
 * view and the wrc overlay: KJ_ERR_UNSUPPORTED). */
KjStatus kj_light_gbuffer(KjDevice* dev, const KjGbufferDepth* gbuffer_depth, const void* shadow_mask, uint32_t shadow_mask_is_rg16f, const void* rtr_tex,
                          const void* rtdgi_tex, const void* unconvolved_sky_cube, uint32_t sky_cube_width, void* out_temporal, void* out,
                          uint32_t debug_shading_mode, void* stream);
/* Rows [row_begin, row_end) of the combine (row_begin a multiple of 8): every input is read at the pixel itself, a strip needs nothing from outside it. */
KjStatus kj_light_gbuffer_rows(KjDevice* dev, const KjGbufferDepth* gbuffer_depth, const void* shadow_mask, uint32_t shadow_mask_is_rg16f, const void* rtr_tex,
                               const void* rtdgi_tex, const void* unconvolved_sky_cube, uint32_t sky_cube_width, void* out_temporal, void* out,
                               uint32_t debug_shading_mode, uint32_t row_begin, uint32_t row_end, void* stream);

/* ---------------------------------------------------------------------------
 * SSAO / SSGI guide — feeds kernel radii and edge-stopping weights of the rtdgi spatial passes, resolve
 * and spatial filter (KjRtdgiRenderParams.ssao_tex)
 *   SsgiRenderer::render(rg, &GbufferDepth, reprojection_map, prev_radiance, bindless_set) -> ReadOnlyHandle<Image>
 *   renderers/ssgi.rs:25-181; shaders/ssgi/{ssgi,spatial_filter,upsample,temporal_filter}.hlsl (USE_AO_ONLY 1)
 * prev_radiance (RGBA16F lit image of the previous frame) is accepted for signature parity and ignored: with
 * USE_AO_ONLY the colour accumulation never reaches the output. *out_ssao_r8 = R8_UNORM full-res image owned by the
 * handle, valid until the next kj_ssgi_render.
 * --------------------------------------------------------------------------- */
typedef struct KjSsgi KjSsgi;
KjStatus kj_ssgi_create(KjDevice* dev, KjSsgi** out);
void kj_ssgi_destroy(KjSsgi* s);
KjStatus kj_ssgi_render(KjSsgi* s, const KjGbufferDepth* gbuffer_depth, const void* reprojection_map, const void* prev_radiance,
                        const void** out_ssao_r8, void* stream);
/* The same for full-res rows [row_begin, row_end) (row_begin a multiple of 16): the screen-tile split computes the guide strip by strip. The
 * intermediate passes over-compute the rows the next pass reaches into; only the temporal pass' history (`ssgi:0|1`, kj_ssgi_surface) is read
 * outside the strip, within motion + 1 rows: the orchestrator exchanges that halo, and the finished guide's halo its consumers reach into. */
KjStatus kj_ssgi_render_rows(KjSsgi* ssgi, const KjGbufferDepth* gbuffer_depth, const void* reprojection_map, const void* prev_radiance, uint32_t row_begin, uint32_t row_end,
                             const void** out_ssao_r8, void* stream);
KjStatus kj_ssgi_surface(KjSsgi* s, const char* name, void** out_dev_ptr, uint64_t* out_bytes);

/* ---------------------------------------------------------------------------
 * Reference path tracer — the convergence oracle of the GI path
 *   reference_path_trace(rg, &mut output_img, bindless_set, tlas)   renderers/reference.rs:8-26
 *   rt/reference_path_trace.rgen.hlsl:75-377 (16-segment eye paths, sun NEE with soft shadows, one triangle
 *   light per vertex, Russian roulette from the 4th vertex, Gaussian pixel filter, firefly suppression)
 * `output` is the persistent RGBA32F accumulation image (rgb = running mean, a = sample count; the reference's
 * "refpt.accum" temporal); each call adds one sample per pixel with the rng seeded from frame_index.
 * first_bounce_mode: 0 = as shipped; 1 = the shader's INDIRECT_ONLY switch; 2 = indirect light through a white
 * Lambert first bounce (the quantity rtdgi's irradiance output estimates; used by the convergence test).
 * interleave_count/index: 8x8 tiles are dealt round-robin to `count` ranks and this call renders tiles with
 * tile % count == index (BASELINE config 5: pixel-interleaved multi-GPU split; sum the images at the end).
 * ray_counter_dev: optional device u64 that receives += rays traced.
 * --------------------------------------------------------------------------- */
KjStatus kj_reference_path_trace(KjDevice* dev, const KjScene* scene, void* output, uint32_t width, uint32_t height,
                                 uint32_t first_bounce_mode, uint32_t interleave_count, uint32_t interleave_index,
                                 uint64_t* ray_counter_dev, void* stream);

/* ---------------------------------------------------------------------------
 * Ray-traced specular reflections
 *   RtrRenderer::trace(rg, &GbufferDepth, reprojection_map, sky_cube, bindless_set, tlas, rtdgi_irradiance, RtdgiCandidates,
 *                      &mut IrcacheRenderState, &WrcRenderState) -> TracedRtr                 renderers/rtr.rs:97-400
 *   TracedRtr::filter_temporal(rg, &GbufferDepth, reprojection_map) -> Handle<Image>          renderers/rtr.rs:440-480
 *   shaders/rtr/{reflection.rgen, reflection_trace_common.inc, reflection_validate.rgen, rtr_restir_temporal, resolve,
 *   temporal_filter, spatial_cleanup}.hlsl with rtr_settings.hlsl as checked in.
 * The caller owns the data tables the reference uploads in RtrRenderer::new: the three `blue-noise-sampler` (crate 0.1.0,
 * spp64) tables RANKING_TILE / SCRAMBLING_TILE (128*128*8 u32 each) and SOBOL (256*256 u32), and rtr.rs's
 * SPATIAL_RESOLVE_OFFSETS (16*4*8 int4). Host pointers, copied at create.
 * kj_rtr_trace OVERWRITES the rtdgi candidate images where the surface is smooth (roughness <= 0.6, `reuse_rtdgi_rays`),
 * exactly as the reference aliases them (rtr.rs:112-116). Temporal state ("rtr.temporal", "rtr.ray_len", "rtr.irradiance",
 * "rtr.ray_orig", "rtr.ray", "rtr.reservoir", "rtr.rng", "rtr.hit_normal", each ":0"/":1") lives in the handle.
 * The world radiance cache (USE_WORLD_RADIANCE_CACHE 0 in the shader) is not included; LightingRenderer::render_specular is kj_rtr_render_specular_lights.
 * Output of kj_rtr_filter_temporal: the B10G11R11_UFLOAT full-res image light_gbuffer consumes.
 * --------------------------------------------------------------------------- */
typedef struct KjRtr KjRtr;
typedef struct KjRtrTables {
    const uint32_t* ranking_tile;
    const uint32_t* scrambling_tile;
    const uint32_t* sobol;
    const int32_t* spatial_resolve_offsets;
} KjRtrTables;
typedef struct KjRtrParams {
    KjGbufferDepth gbuffer_depth;
    const void* reprojection_map;     /* RGBA16_SNORM full res */
    const void* sky_cube;             /* UNconvolved 6 x w x w RGBA16F (world_render_passes.rs:178) */
    uint32_t sky_cube_width;
    KjScene* scene;
    KjIrcache* ircache;               /* NULL => lookups return 0 */
    const void* rtdgi_irradiance;     /* RtdgiOutput::screen_irradiance_tex, RGBA16F full res */
    void* candidate_radiance_tex;     /* RtdgiCandidates (KjRtdgiOutput): RGBA16F, RGBA16F, RGBA8_SNORM half res; read + written */
    void* candidate_hit_tex;
    void* candidate_normal_tex;
    uint32_t pass_mask;               /* KJ_RTR_PASS_ALL for the product path */
} KjRtrParams;
#define KJ_RTR_PASS_TRACE 1u
#define KJ_RTR_PASS_VALIDATE 2u
#define KJ_RTR_PASS_RESTIR_TEMPORAL 4u
#define KJ_RTR_PASS_RESOLVE 8u
#define KJ_RTR_PASS_TEMPORAL_FILTER 16u
#define KJ_RTR_PASS_CLEANUP 32u
#define KJ_RTR_PASS_ALL 63u
#define KJ_RTR_PASS_EXTRACT_HALF 64u      /* kj_rtr_render_rows only (kj_rtr_trace always runs it): half-res view normal / depth of the whole frame */
#define KJ_RTR_PASS_SPECULAR_LIGHTS 128u  /* kj_rtr_render_rows only: kj_rtr_render_specular_lights on the rows */
#define KJ_RTR_PASS_KEEP 0x80000000u  /* tests: do not advance the ping-pong state (run one pass of an already started frame) */
KjStatus kj_rtr_create(KjDevice* dev, const KjRtrTables* tables, KjRtr** out);
void kj_rtr_destroy(KjRtr* r);
KjStatus kj_rtr_set_options(KjRtr* r, uint32_t reuse_rtdgi_rays);
KjStatus kj_rtr_trace(KjRtr* r, const KjRtrParams* params, void* stream);
/* LightingRenderer::render_specular(&mut rtr.resolved_tex, rg, &GbufferDepth, bindless_set, tlas)   renderers/lighting.rs:23-88,
 * shaders/lighting/{sample_lights.rgen, spatial_reuse_lights}.hlsl — specular from the triangle lights (one light sample + shadow ray per
 * half-res pixel, 8-tap reuse), ADDED into the resolved image between kj_rtr_trace and kj_rtr_filter_temporal so both are filtered together
 * (world_render_passes.rs:190-203). A no-op when the scene has no triangle lights, as in the reference. */
KjStatus kj_rtr_render_specular_lights(KjRtr* r, const KjRtrParams* params, void* stream);
KjStatus kj_rtr_filter_temporal(KjRtr* r, const KjRtrParams* params, const void** out_resolved_r11g11b10f, void* stream);
/* The passes of params->pass_mask on full-res rows [row_begin, row_end) (cuts on 16-row boundaries; the ray passes, the reservoir pass and the specular
 * lights run on the half-res rows underneath): the screen-tile split renders reflections strip by strip with it (kj_split_rtr_frame). A call without
 * KJ_RTR_PASS_KEEP opens the frame (ping-pong flip, ray counters cleared); every later call of the frame carries it. What a pass reads beyond the rows
 * (RtrRenderer has no notion of rows: rtr.rs:97-480) must be in place: DESIGN 7 lists the reach of every pass. */
KjStatus kj_rtr_render_rows(KjRtr* r, const KjRtrParams* params, uint32_t row_begin, uint32_t row_end, const void** out_resolved_r11g11b10f, void* stream);
KjStatus kj_rtr_surface(KjRtr* r, const char* name, void** out_dev_ptr, uint64_t* out_bytes);
KjStatus kj_rtr_ray_counts(KjRtr* r, uint64_t* out_closest, uint64_t* out_any);

/* ---------------------------------------------------------------------------
 * Post-processing (SURVEY 8f-4 "minimal post": the tail of the frame, from the anti-aliased image to the display-referred one)
 *   PostProcessRenderer::render(rg, input, bindless_set, post_exposure_mult, contrast, exposure_histogram_clipping) -> Handle<Image>
 *                                                                                               renderers/post.rs:237-271
 *   blur_pyramid (post.rs:10-61; rust-shaders/src/blur.rs for mip 0, shaders/blur.hlsl for the rest), luminance histogram
 *   (post.rs:138-186, shaders/post/luminance_histogram_*.hlsl), rev_blur_pyramid (post.rs:63-110, rust-shaders/src/rev_blur.rs),
 *   "post combine" (shaders/post_combine.hlsl + inc/color/display_transform.hlsl and the colour headers it pulls in).
 * input: full-res image, RGBA16F (TaaOutput.this_frame_out / kj_motion_blur_render's output: the standard frame) or RGBA32F (the path
 * tracer's accumulation image: prepare_render_graph_reference hands it to post as is, world_render_passes.rs:294-330). Output:
 * B10G11R11_UFLOAT full-res, LINEAR display-referred values in [0, ~1] — kajiya's swap chain applies the sRGB transfer function.
 * The Bezold-Brucke LUT (bindless texture 2: 64 x 1 RG16F, lut_renderers.rs:45-76) is caller data like the blue-noise image: host
 * pointer, copied at create. frame_index (dither offset) and pre_exposure (histogram) come from kj_frame_begin's constants.
 * Surfaces: "blur_pyramid:<mip>", "rev_blur_pyramid:<mip>" (B10G11R11, mip k = max(1, ceil(W/2) >> k) x max(1, ceil(H/2) >> k)),
 * "histogram" (256 x u32), "output".
 * kj_post_read_back_histogram = PostProcessRenderer::read_back_histogram (post.rs:188-235) on the host-visible copy of the
 * histogram: like the reference's mapped buffer it holds whatever copy has completed (synchronise the stream for a deterministic
 * answer); out_histogram256 optional. kj_luminance_histogram_mean_log2 is the same arithmetic on a caller's histogram (no device).
 * --------------------------------------------------------------------------- */
typedef struct KjPost KjPost;
KjStatus kj_post_create(KjDevice* dev, const uint16_t* bezold_brucke_lut_rg16f_64, KjPost** out);
void kj_post_destroy(KjPost* p);
enum { KJ_POST_INPUT_RGBA16F = 0, KJ_POST_INPUT_RGBA32F = 1 };
KjStatus kj_post_render(KjPost* p, const void* input, uint32_t input_format, uint32_t width, uint32_t height, float post_exposure_mult, float contrast,
                        const void** out_b10g11r11, void* stream);
KjStatus kj_post_read_back_histogram(KjPost* p, float clipping_low, float clipping_high, float* out_image_log2_lum, uint32_t* out_histogram256);
KjStatus kj_luminance_histogram_mean_log2(const uint32_t* histogram256, float clipping_low, float clipping_high, float* out_image_log2_lum);
KjStatus kj_post_surface(KjPost* p, const char* name, void** out_dev_ptr, uint64_t* out_bytes);
KjStatus kj_post_mip_levels(KjPost* p, uint32_t* out_levels);

/* motion_blur(rg, input, depth, reprojection_map) -> Handle<Image>     renderers/motion_blur.rs:5-72; rust-shaders/src/motion_blur.rs
 * (velocity_reduce_x, velocity_reduce_y, velocity_dilate, motion_blur — all four are the Rust kernels the reference runs). Sits between
 * TaaRenderer::render and PostProcessRenderer::render (world_render_passes.rs:265-266). input / output RGBA16F at (width, height) — the
 * TAA output extent; depth R32F and reprojection_map RGBA16_SNORM at (depth_width, depth_height) — the render extent. *out_rgba16f is
 * owned by the handle, valid until the next call. Surfaces: "velocity_reduced_x", "velocity_reduced_y", "velocity_dilated" (RG16F),
 * "output". motion_blur_scale is the reference's constant 1.0. */
typedef struct KjMotionBlur KjMotionBlur;
KjStatus kj_motion_blur_create(KjDevice* dev, KjMotionBlur** out);
void kj_motion_blur_destroy(KjMotionBlur* m);
KjStatus kj_motion_blur_render(KjMotionBlur* m, const void* input_rgba16f, uint32_t width, uint32_t height, const void* depth_r32f, const void* reprojection_map,
                               uint32_t depth_width, uint32_t depth_height, const void** out_rgba16f, void* stream);
KjStatus kj_motion_blur_surface(KjMotionBlur* m, const char* name, void** out_dev_ptr, uint64_t* out_bytes);

/* ---------------------------------------------------------------------------
 * Screen-tile split across the GPUs of a node (north star: "partition across the GPUs of one node by screen-space tile with
 * a halo exchange of reservoirs / history over RCCL / xGMI"). The reference has no counterpart: kajiya renders on one GPU; what is split
 * is RtdgiRenderer::render (rtdgi.rs:173-554) + TaaRenderer::render (taa.rs:41-191) of ONE frame, pass by pass, with the irradiance
 * cache replicated and kept bit-identical across ranks (kj_ircache_set_deferred_updates). KjSplit is the compiled orchestrator:
 * one per process, created over the renderers of the ranks living in that process -- exactly one with an RCCL communicator
 * (`nccl_comm`, an ncclComm_t), or all `world` of them without (virtual ranks on one device: the exchange is device-to-device copies),
 * or all of them WITH a communicator of exactly one rank (loopback: every message is an ncclSend to self matched by an ncclRecv from
 * self, the summaries' all-gather an all-gather of one -- the transport code and RCCL itself on the single GPU of a build box).
 * It calls the entry points above with row ranges and exchanges halos in between: one packed message per peer and exchange point,
 * ncclSend / ncclRecv inside one group. kajiya_amd/multigpu.py is the reference implementation of the same schedule.
 * Strips are 16-row aligned; surfaces are looked up by name through kj_rtdgi_surface / kj_taa_surface. RCCL is loaded with dlopen on
 * first use: kj_split_rccl_* bootstrap a communicator from a 128-byte id the caller broadcasts by its own means.
 * --------------------------------------------------------------------------- */
typedef struct KjSplit KjSplit;
typedef struct KjSplitRank { KjRtdgi* rtdgi; KjTaa* taa; KjIrcache* ircache /* NULL: cache unbound */; KjScene* scene; } KjSplitRank;
typedef struct KjSplitFrame {            /* one rank's inputs of a frame */
    KjRtdgiRenderParams rtdgi;           /* as for kj_rtdgi_render; pass_mask, row_begin / row_end and spatial_pass_select are overwritten */
    KjRtdgiOutput* rtdgi_out;
    KjTaaOutput* taa_out;
    const void* sky_cube16;              /* convolved cube for kj_ircache_trace_irradiance (unused when the cache's passes ran already) */
} KjSplitFrame;
KjStatus kj_split_create(uint32_t world, uint32_t first_rank, uint32_t local_ranks, const KjSplitRank* ranks, uint32_t width, uint32_t height,
                         uint32_t motion_halo /* rows of history exchanged beyond the stencils: >= max |screen motion| per frame */, void* nccl_comm, KjSplit** out);
void kj_split_destroy(KjSplit* split);
KjStatus kj_split_strip(KjSplit* split, uint32_t rank, uint32_t* out_row_begin, uint32_t* out_row_end);
/* One GI frame: [cache prepare + rays unless KJ_SPLIT_IRCACHE_DONE], reproject, the rtdgi passes with exchanges A-D and H, and -- unless
 * KJ_SPLIT_DEFER_IRCACHE_MERGE -- the replay of the cache's recorded updates (each rank's summary, one fixed-size all-gather, the same merge
 * everywhere: no host synchronisation; a pipelining caller still defers it and calls kj_split_merge_ircache on its cache stream once the frame
 * is enqueued, before the next frame's cache work, so that the all-gather overlaps the resampling chain). `frames`: one entry per LOCAL rank. `trace_done_event`: optional hipEvent_t recorded once the ray passes are enqueued. */
enum { KJ_SPLIT_IRCACHE_DONE = 1u, KJ_SPLIT_DEFER_IRCACHE_MERGE = 2u };
KjStatus kj_split_gi_frame(KjSplit* split, const KjSplitFrame* frames, uint32_t flags, void* trace_done_event, void* stream);
KjStatus kj_split_merge_ircache(KjSplit* split, void* stream);
/* The SSAO guide of the frame, before kj_split_gi_frame: SsgiRenderer::render strip by strip (kj_ssgi_render_rows) with the halo exchanges of its temporal
 * history and of the finished guide. `ssgi`, `out_ssao_r8`: one entry per LOCAL rank; out_ssao_r8[i] is what frames[i].rtdgi.ssao_tex must point at. */
KjStatus kj_split_ssgi_frame(KjSplit* split, KjSsgi* const* ssgi, const KjSplitFrame* frames, const void** out_ssao_r8, void* stream);
/* Sun shadows of the frame strip by strip: trace_sun_shadow_mask (each rank's rays for its own rows, into the caller's R8 image mask_r8[i]) +
 * ShadowDenoiseRenderer::render (kj_shadow_denoise_render_rows) with the halo exchanges of the denoiser's two histories and of the mask. One entry per LOCAL
 * rank in every array; out_rg16f[i] is valid on rank i's own rows (what kj_light_gbuffer_rows reads); ray_counters_dev: NULL or optional device u64s. */
KjStatus kj_split_shadow_frame(KjSplit* split, KjShadowDenoise* const* denoisers, const KjSplitFrame* frames, void* const* mask_r8, uint64_t* const* ray_counters_dev,
                               const void** out_rg16f, void* stream);
/* Reflections of the frame strip by strip, after kj_split_gi_frame: RtrRenderer::trace + LightingRenderer::render_specular + TracedRtr::filter_temporal
 * (renderers/rtr.rs:97-480, world_render_passes.rs:172-210) through kj_rtr_render_rows, with four exchange points: the all-gather of this frame's GI image (a
 * reflection's hit reads it anywhere on screen), the reservoir histories' halos after the ray passes, the all-gather of the reservoir pass' outputs (the resolve's
 * taps land where a world-space kernel projects to), the all-gather of the temporal filter's output (its own history next frame, read at the reflection's
 * reprojected virtual position). kj_split_set_rtr(split, 1) comes first, once: the caches reserve slot ranges for the lookups of rtr's rays
 * (kj_ircache_set_rtr_requests) and the replay of the frame's recorded cache updates moves from kj_split_gi_frame behind rtr's ray passes (here, or
 * kj_split_merge_ircache with KJ_SPLIT_DEFER_IRCACHE_MERGE). `rtr`, `rtr_params`, `out_resolved_r11g11b10f`: one entry per LOCAL rank; rtr_params[i] as for
 * kj_rtr_trace (pass_mask ignored); out_resolved_r11g11b10f (may be NULL) [i] is valid on rank i's own rows, which is all kj_light_gbuffer_rows reads. */
enum { KJ_SPLIT_RTR_SPECULAR_LIGHTS = 1u };      /* with KJ_SPLIT_DEFER_IRCACHE_MERGE: the flags of kj_split_rtr_frame */
KjStatus kj_split_set_rtr(KjSplit* split, uint32_t enable);
KjStatus kj_split_rtr_frame(KjSplit* split, KjRtr* const* rtr, const KjRtrParams* rtr_params, uint32_t flags, void* trace_done_event, const void** out_resolved_r11g11b10f,
                            void* stream);
/* TAA on the GI output of the frame just rendered (exchange I + strip-wise TaaRenderer::render). */
KjStatus kj_split_taa_frame(KjSplit* split, const KjSplitFrame* frames, void* stream);
/* The same on images of the caller's -- input_rgba16f[i] (one per LOCAL rank, full res, valid on the rank's own rows; its halo rows are written into it): the
 * lighting frame resolves the LIT image (world_render_passes.rs:212-291: light_gbuffer's output is what TaaRenderer::render takes), i.e. what
 * kj_light_gbuffer_rows wrote on each rank's strip. NULL: the GI image (kj_split_taa_frame). */
KjStatus kj_split_taa_frame_on(KjSplit* split, const KjSplitFrame* frames, void* const* input_rgba16f, void* stream);
/* Every rank receives the owners' rows of a surface ("spatial_filtered_tex", "TAA/taa:0", ...): result collection. */
KjStatus kj_split_gather(KjSplit* split, const char* surface_name, void* stream);
/* Start-up check of the transport, before frame 0: every kind of exchange of the frame schedule (all-gather of an image, mixed surfaces packed into one message
 * per peer, the 64-row one-deep halo, stencil halos, the fixed-size all-gather of the cache's summaries) once on scratch images whose rows carry their
 * owner's rank, read back and checked row by row. *out_passed: 1 when every rank of this process holds exactly the rows it is entitled to (combine the
 * processes' verdicts with the caller's own collective). Synchronises `stream`. No counterpart in the reference (single-GPU). */
KjStatus kj_split_self_test(KjSplit* split, uint32_t* out_passed, void* stream);
/* Measurement aid: with profiling on every exchange is bracketed by two HIP events on the frame's stream (pack + transport + scatter; with virtual ranks: the
 * device copies standing in for the wire) and the bytes arriving at the busiest local rank are summed. kj_split_profile waits for the recorded events and
 * reports the totals since kj_split_set_profiling(split, 1). */
typedef struct KjSplitProfile { double exchange_ms; uint64_t exchange_bytes_busiest_rank; uint32_t exchange_points; uint32_t gi_frames; } KjSplitProfile;
KjStatus kj_split_set_profiling(KjSplit* split, uint32_t enable);
KjStatus kj_split_profile(KjSplit* split, KjSplitProfile* out);
KjStatus kj_split_rccl_unique_id(uint8_t out_id[128]);
KjStatus kj_split_rccl_comm_create(const uint8_t id[128], uint32_t world, uint32_t rank, void** out_comm);
/* ncclCommCount / ncclCommUserRank of a communicator. */
KjStatus kj_split_rccl_comm_info(void* comm, uint32_t* out_ranks, uint32_t* out_rank);
void kj_split_rccl_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* KAJIYA_AMD_H */
