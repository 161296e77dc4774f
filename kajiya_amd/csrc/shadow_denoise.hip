// ShadowDenoiseRenderer for gfx950 (renderers/shadow_denoise.rs:19-148; shaders/shadow_denoise/{bitpack_shadow_mask,megakernel,
// spatial_filter}.hlsl + the FidelityFX shadow-denoiser headers they include, with kajiya's callbacks — see oracle/okj_shadow_denoise.hpp
// for the list). 8x8 tile = one wave64: the 8x4 bit masks of the "prepare" pass are the two halves of one 64-bit ballot.
#include "kj_host.hpp"
#include "kj_shading.hpp"
#include "kj_screen.hpp"

using namespace kj;

typedef Img<uint2> ImgH4;
typedef Img<uint32_t> ImgU32;
typedef Img<float> ImgF32;
typedef Img<uint8_t> ImgR8;

// whole tile rows per XCD (kj_vec.hpp: tile_order; every tile of these passes costs the same): spatial 37.5 -> 30.6 us per pass at 1440p
// tile_row0: the first 8-row tile row of the launch (strips of the screen-tile split start there; 0 for the whole image)
#define TILE_XY()                                                                          \
    const int lane = threadIdx.x;                                                          \
    uint2 kj_tb = kj::tile_order<KJ_TILES_ROWS>();                                        \
    kj_tb.y += uint32_t(tile_row0);                                                        \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = int(kj_tb.y) * 8 + (lane >> 3);

// "shadow bitpack" (bitpack_shadow_mask.hlsl + ffx prepare): bit (y%4)*8 + x%8 of tile (x/8, y/4) = ray reached the light
__global__ void __launch_bounds__(64) k_shadow_bitpack(ImgR8 input_tex, ImgU32 output_tex, int tile_row0) {
    TILE_XY()
    const bool hit = from_unorm8(input_tex.ld(x, y)) > 0.5f;     // OOB = 0 = shadowed, like the shader's OOB load
    const unsigned long long m = __ballot(hit);
    if (lane == 0) {
        output_tex.st(int(kj_tb.x), int(kj_tb.y) * 2, uint32_t(m & 0xffffffffull));
        output_tex.st(int(kj_tb.x), int(kj_tb.y) * 2 + 1, uint32_t(m >> 32));
    }
}

struct KernelWeights { float w[9]; };   // FFX_DNSR_Shadows_KernelWeight(0..8), KERNEL_RADIUS 8, normalised (host-computed constants)

KJ_D V4 cubic_hermite4(V4 A, V4 B, V4 C, V4 D, float t) {   // inc/curve.hlsl:4-13
    const float t2 = t * t, t3 = t * t * t;
    const V4 a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
    const V4 b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
    const V4 c = -A / 2.0f + C / 2.0f;
    return a * t3 + b * t2 + c * t + B;
}
// image_sample_catmull_rom (inc/image.hlsl:41-82)
KJ_D V4 catmull_rom_rgba16f(const ImgH4& img, V2 P) {
    const V2 pixel{P.x * float(img.w) + 0.5f, P.y * float(img.h) + 0.5f};
    const V2 frc{frac(pixel.x), frac(pixel.y)};
    const int ix = int(pixel.x) - 1, iy = int(pixel.y) - 1;
    V4 rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[j] = cubic_hermite4(ld4(img, ix - 1, iy - 1 + j), ld4(img, ix, iy - 1 + j), ld4(img, ix + 1, iy - 1 + j), ld4(img, ix + 2, iy - 1 + j), frc.x);
    return cubic_hermite4(rows[0], rows[1], rows[2], rows[3], frc.y);
}
KJ_D V4 ld2h4(const ImgU32& img, int x, int y) { const V2 v = ld2h(img, x, y); return V4{v.x, v.y, 0, 0}; }
KJ_D float catmull_rom_rg16f_x(const ImgU32& img, V2 P) {
    const V2 pixel{P.x * float(img.w) + 0.5f, P.y * float(img.h) + 0.5f};
    const V2 frc{frac(pixel.x), frac(pixel.y)};
    const int ix = int(pixel.x) - 1, iy = int(pixel.y) - 1;
    V4 rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[j] = cubic_hermite4(ld2h4(img, ix - 1, iy - 1 + j), ld2h4(img, ix, iy - 1 + j), ld2h4(img, ix + 1, iy - 1 + j), ld2h4(img, ix + 2, iy - 1 + j), frc.x);
    return cubic_hermite4(rows[0], rows[1], rows[2], rows[3], frc.y).x;
}
KJ_D float soft_color_clamp1(float center, float history, float ex, float dev) {   // inc/soft_color_clamp.hlsl
    const float history_dist = fabsf(history - ex) / fmaxf(fabsf(history * 0.1f), dev);
    const float closest_pt = clampf(history, center - dev, center + dev);
    return lerp(history, closest_pt, smoothstep(1.0f, 3.0f, history_dist));
}

struct ShadowTemporalArgs {
    const FrameConstants* __restrict__ fc;
    ImgR8 shadow_mask_tex; ImgU32 bitpacked_tex; ImgH4 prev_moments_tex; ImgU32 prev_accum_tex /*RG16F*/; ImgH4 reprojection_tex;
    ImgH4 output_moments_tex; ImgU32 temporal_output_tex /*RG16F*/; ImgU32 meta_output_tex;
    KernelWeights kw;
    int W, H, TW, TH;
    int tile_row0;
};
// (round 6: the three mask words of a row are fetched by the caller -- nine loads in flight for the three rows of a block instead of nine round trips; `lcr` holds what
// bitpacked_tex.ld returns at (tix - 1, tiy), (tix, tiy), (tix + 1, tiy): zero outside the image)
KJ_D void shadow_neighborhood_masks(const ShadowTemporalArgs& a, int dx, int dy, uint32_t lcr[3], bool in_[3]) {
    const int tix = dx / 8, tiy = (dy < 0 ? 0 : dy) / 4;      // (rows above the image are rejected by the consumer)
#pragma unroll
    for (int k = 0; k < 3; ++k) lcr[k] = a.bitpacked_tex.ld_raw(tix - 1 + k, tiy, in_[k]);
}
KJ_D float shadow_horizontal_neighborhood(const ShadowTemporalArgs& a, int dx, int dy, const uint32_t lcr[3], const bool in_[3]) {
    if (dy < 0 || dy >= a.H) return 0.0f;
    const int tix = dx / 8;
    const uint32_t left_tile = (tix == 0 || !in_[0]) ? 0u : lcr[0];
    const uint32_t center_tile = in_[1] ? lcr[1] : 0u;
    const uint32_t right_tile = (tix == a.TW - 1 || !in_[2]) ? 0u : lcr[2];
    const uint32_t row = uint32_t(dy % 4) * 8u;
    uint32_t nb = ((left_tile >> row) & 0xFFu) | (((center_tile >> row) & 0xFFu) << 8) | (((right_tile >> row) & 0xFFu) << 16);
    nb >>= uint32_t(dx % 8);
    float moment = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (nb & (1u << i)) moment += a.kw.w[8 - i];
    if (nb & (1u << 8)) moment += a.kw.w[0];
#pragma unroll
    for (int i = 1; i <= 8; ++i) if (nb & (1u << (8 + i))) moment += a.kw.w[i];
    return moment;
}
// "shadow temporal" (megakernel.hlsl + ffx tile classification)
__global__ void __launch_bounds__(64) k_shadow_temporal(ShadowTemporalArgs a) {
    const int tile_row0 = a.tile_row0;
    TILE_XY()
    const int gx = int(kj_tb.x), gy = int(kj_tb.y);
    // FFX_DNSR_Shadows_SearchSpatialRegion: 3 x 6 masks around the two 8x4 tiles of this block (wave-uniform)
    uint32_t or_mask = 0, and_mask = 0xFFFFFFFFu;
    for (int j = -2; j <= 3; ++j)
        for (int i = -1; i <= 1; ++i) {
            const uint32_t m = a.bitpacked_tex.ld(min(max(gx + i, 0), a.TW - 1), min(max(gy * 2 + j, 0), a.TH - 1));
            or_mask |= m; and_mask &= m;
        }
    const bool all_in_light = and_mask == 0xFFFFFFFFu, all_in_shadow = or_mask == 0u;
    if (all_in_light || all_in_shadow) {   // FFX_DNSR_Shadows_ClearTargets (every pixel is a shadow receiver in kajiya's callbacks)
        const float shadow_value = all_in_light ? 1.0f : 0.0f;
        if (lane == 0) a.meta_output_tex.st(gx, gy, (all_in_light ? 2u : 0u) | 1u);
        st2h(a.temporal_output_tex, x, y, V2{shadow_value, 0});
        st4(a.output_moments_tex, x, y, V4{shadow_value, 0, 8, shadow_value});
        return;
    }
    if (lane == 0) a.meta_output_tex.st(gx, gy, 0u);
    __shared__ float hn[8][24];
    const int lx = lane & 7, ly = lane >> 3;
    uint32_t m0[3], m1[3], m2[3]; bool i0[3], i1[3], i2[3];
    shadow_neighborhood_masks(a, x, y - 8, m0, i0); shadow_neighborhood_masks(a, x, y, m1, i1); shadow_neighborhood_masks(a, x, y + 8, m2, i2);
    bool c_in;      // the pixel's own reprojection and mask texels: requested with the masks, used behind the barrier
    const uint2 reproj_raw = a.reprojection_tex.ld_raw(x, y, c_in);
    const uint8_t mask_raw = a.shadow_mask_tex.ld_raw(x, y, c_in);
    hn[lx][ly] = shadow_horizontal_neighborhood(a, x, y - 8, m0, i0);
    hn[lx][ly + 8] = shadow_horizontal_neighborhood(a, x, y, m1, i1);
    hn[lx][ly + 16] = shadow_horizontal_neighborhood(a, x, y + 8, m2, i2);
    __syncthreads();
    float local_neighborhood = 0;
    local_neighborhood += hn[lx][ly + 8] * a.kw.w[0];
    local_neighborhood += hn[lx][ly] * a.kw.w[8];
    local_neighborhood += hn[lx][ly + 16] * a.kw.w[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        local_neighborhood += hn[lx][8 + ly - i] * a.kw.w[i];
        local_neighborhood += hn[lx][8 + ly + i] * a.kw.w[i];
    }
    const uint2 rpr = c_in ? reproj_raw : make_uint2(0u, 0u);
    const V4 reproj{from_snorm16(int16_t(rpr.x & 0xffff)), from_snorm16(int16_t(rpr.x >> 16)), from_snorm16(int16_t(rpr.y & 0xffff)), from_snorm16(int16_t(rpr.y >> 16))};
    const V2 uv{(float(x) + 0.5f) * (1.0f / float(a.W)), (float(y) + 0.5f) * (1.0f / float(a.H))};     // ffx ...tileclassification.hlsl:351-352: times texel_size (the host's f32 reciprocals)
    const V2 history_uv = uv + V2{reproj.x, reproj.y};
    const float shadow_current = from_unorm8(c_in ? mask_raw : uint8_t(0));
    const uint32_t qv = uint32_t(reproj.z * 15.0f + 0.5f);
    const bool is_disoccluded = ((qv & 1u) + ((qv >> 1) & 1u) + ((qv >> 2) & 1u) + ((qv >> 3) & 1u)) < 4u;
    V4 previous_moments = v4(0.0f);
    if (!is_disoccluded) {
        previous_moments = catmull_rom_rgba16f(a.prev_moments_tex, history_uv);
        previous_moments.y = fmaxf(0.0f, previous_moments.y);
        previous_moments.z = fmaxf(0.0f, previous_moments.z);
    }
    const float old_m = previous_moments.x, old_s = previous_moments.y;
    const float sample_count = previous_moments.z + 1.0f;
    const float new_m = lerp(old_m, shadow_current, 1.0f / sample_count);
    const float new_s = lerp(old_s, (shadow_current - old_m) * (shadow_current - new_m), 1.0f / sample_count);
    float variance = new_s;
    V4 moments_current{new_m, new_s, sample_count, local_neighborhood};
    const float mean = local_neighborhood;
    const float spatial_variance = fmaxf(local_neighborhood - mean * mean, 0.0f);
    const float std_deviation = sqrtf(spatial_variance);
    float shadow_previous = shadow_current;
    if (a.fc->frame_index != 0) shadow_previous = catmull_rom_rg16f_x(a.prev_accum_tex, history_uv);
    const float sigma = 2.0f;
    const float temporal_discontinuity = (previous_moments.w - moments_current.w) / fmaxf(0.5f * std_deviation, 0.001f);
    const float sample_counter_damper = expf(-temporal_discontinuity * temporal_discontinuity / sigma);
    moments_current.z *= fmaxf(0.5f, sample_counter_damper);
    float shadow_clamped = soft_color_clamp1(shadow_current, shadow_previous, mean, std_deviation * 0.5f);
    if (moments_current.z < 16.0f) {
        const float variance_boost = fmaxf(16.0f - moments_current.z, 1.0f);
        variance = fmaxf(variance, spatial_variance);
        variance *= variance_boost;
    }
    shadow_clamped = lerp(shadow_clamped, shadow_current, 1.0f / fmaxf(1.0f, moments_current.z));
    st2h(a.temporal_output_tex, x, y, V2{shadow_clamped, variance});
    moments_current.z = fminf(moments_current.z, 32.0f);
    st4(a.output_moments_tex, x, y, moments_current);
}

// "shadow spatial" (spatial_filter.hlsl + ffx filter): 16x16 LDS tile holding packed halves exactly like the shader
__global__ void __launch_bounds__(64) k_shadow_spatial(ImgU32 input_tex /*RG16F*/, ImgU32 meta_tex, ImgU32 geometric_normal_tex, ImgF32 depth_tex, ImgU32 output_tex, int stepsize, int tile_row0) {
    TILE_XY()
    const int W = depth_tex.w, H = depth_tex.h;
    const uint32_t meta = meta_tex.ld(int(kj_tb.x), int(kj_tb.y));
    if (meta & 1u) {   // cleared tile, pass index 0: write the constant
        st2h(output_tex, x, y, V2{(meta & 2u) ? 1.0f : 0.0f, 0.0f});
        return;
    }
    __shared__ uint32_t s_in[16][16], s_nxy[16][16], s_nzw[16][16];
    __shared__ float s_depth[16][16];
    // (round 6: a lane's four staged texels x three images and its own depth are requested together: the rolled loop was four round trips of three loads each)
    bool d_in;
    const float depth_raw = depth_tex.ld_raw(x, y, d_in);
    uint32_t st_n[4], st_i[4]; float st_d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, tx = i & 15, ty = i >> 4;
        // ffx_denoiser_shadows_filter.hlsl:76 clamps an int2 against uint2 dimensions: the comparison is unsigned, a negative coordinate lands on the FAR edge
        const int px = int(min(uint32_t(int(kj_tb.x) * 8 - 4 + tx), uint32_t(W - 1))), py = int(min(uint32_t(int(kj_tb.y) * 8 - 4 + ty), uint32_t(H - 1)));
        const size_t at = size_t(py) * W + px;      // clamped: in bounds (the three images share the extent)
        st_n[k] = geometric_normal_tex.p[at]; st_i[k] = input_tex.p[at]; st_d[k] = depth_tex.p[at];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k, tx = i & 15, ty = i >> 4;
        const V3 n = unpack_a2r10g10b10(st_n[k]) * 2.0f - 1.0f;
        s_in[ty][tx] = st_i[k];                       // already two packed halves
        s_depth[ty][tx] = st_d[k];
        s_nxy[ty][tx] = pack_2x16f_uint(n.x, n.y);
        s_nzw[ty][tx] = pack_2x16f_uint(n.z, 0.0f);
    }
    __syncthreads();
    float weight_sum = 1.0f;
    V2 shadow_sum{0, 0};
    const float depth = d_in ? depth_raw : 0.0f;
    if (depth != 0.0f) {
        const int cx = (lane & 7) + 4, cy = (lane >> 3) + 4;
        const V2 shadow_center = unpack_2x16f_uint(s_in[cy][cx]);
        const V2 nxy = unpack_2x16f_uint(s_nxy[cy][cx]);
        const V3 normal_center{nxy.x, nxy.y, unpack_2x16f_uint(s_nzw[cy][cx]).x};
        shadow_sum = shadow_center;
        const float std_deviation = sqrtf(fmaxf(shadow_center.y + 1e-9f, 0.0f));
        const float t_ = fmaxf(0.0f, 1.0f - 2.0f * std_deviation);
        const float kernel_sharpening = fmaxf(1e-10f, 1.0f - t_ * t_);
        const float kernel[3] = {1.0f, exp2f(-0.5849625007211563f / kernel_sharpening), exp2f(-2.584962500721156f / kernel_sharpening)};
        const float inv_std_log2e = 1.4426950408889634f / std_deviation;
#pragma unroll
        for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
            for (int xx = -1; xx <= 1; ++xx) {
                const int tx = cx + xx * stepsize, ty = cy + yy * stepsize;
                const float depth_neigh = s_depth[ty][tx];
                const V2 mxy = unpack_2x16f_uint(s_nxy[ty][tx]);
                const V3 normal_neigh{mxy.x, mxy.y, unpack_2x16f_uint(s_nzw[ty][tx]).x};
                const V2 shadow_neigh = unpack_2x16f_uint(s_in[ty][tx]);
                const float sky_mul = ((xx == 0 && yy == 0) || depth_neigh >= 1.0f || depth_neigh <= 0.0f) ? 0.0f : 1.0f;
                float w = kernel[xx < 0 ? -xx : xx] * kernel[yy < 0 ? -yy : yy];
                // weights: exp() as v_exp_f32 of the argument in base 2, pow(x, 32) as five squarings (libm's powf is 163 VALU instructions on
                // gfx950 and ran nine times per pixel in each of the three passes)
                w *= exp2_fast(-fabsf(shadow_center.x - shadow_neigh.x) * inv_std_log2e);
                w *= exp2_fast(-fabsf(1.0f - depth * rcp_fast(depth_neigh)) * 100.0f);
                float nd = saturate(dot(normal_center, normal_neigh));
                nd *= nd; nd *= nd; nd *= nd; nd *= nd; nd *= nd;
                w *= nd;
                w *= sky_mul;
                shadow_sum += V2{w, w * w} * shadow_neigh;
                weight_sum += w;
            }
    }
    const float mean = shadow_sum.x / weight_sum, variance = shadow_sum.y / (weight_sum * weight_sum);
    st2h(output_tex, x, y, V2{fmaxf(0.0f, mean), fmaxf(0.0f, variance)});
}

// ================================================================== host
struct KjShadowDenoise {
    KjDevice* dev = nullptr;
    int W = 0, H = 0;
    std::map<std::string, kj::DevBuf> surf;
    bool flip_accum = false, flip_moments = false;
    hipError_t err = hipSuccess;
    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
};

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_shadow_denoise_create(KjDevice* dev, KjShadowDenoise** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjShadowDenoise* t = new KjShadowDenoise();
    t->dev = dev;
    *out = t;
    return KJ_OK;
}
void kj_shadow_denoise_destroy(KjShadowDenoise* t) { delete t; }

// ShadowDenoiseRenderer::render(rg, &GbufferDepth, shadow_mask, reprojection_map) -> ReadOnlyHandle<Image> (shadow_denoise.rs:19-25)
KjStatus kj_shadow_denoise_render(KjShadowDenoise* t, const KjGbufferDepth* gd, const void* shadow_mask_r8, const void* reprojection_map, const void** out_rg16f, void* stream_) {
    KJ_REQUIRE(gd, "null argument");
    return kj_shadow_denoise_render_rows(t, gd, shadow_mask_r8, reprojection_map, 0u, gd->height, out_rg16f, stream_);
}
// The denoised term for full-res rows [row_begin, row_end) (row_begin a multiple of 16): the screen-tile split computes it strip by strip. Every pass runs on
// the rows the next one reaches into, so only the mask (valid on [r0 - 32, r1 + 32)) and the two histories the temporal pass reads through the motion
// vectors come from outside the strip:
//   spatial step 4 on [r0, r1) reads step 2's output +-4 rows -> step 2 on [r0 - 8, r1 + 8) reads +-2 -> step 1 on [r0 - 16, r1 + 16) reads the temporal
//   pass' output +-1 -> temporal (and the tile metadata) on [r0 - 24, r1 + 24) reads the bit-packed mask +-8 rows -> bitpack on [r0 - 32, r1 + 32).
KjStatus kj_shadow_denoise_render_rows(KjShadowDenoise* t, const KjGbufferDepth* gd, const void* shadow_mask_r8, const void* reprojection_map, uint32_t row_begin, uint32_t row_end,
                                       const void** out_rg16f, void* stream_) {
    KJ_REQUIRE(t && gd && gd->depth && gd->geometric_normal && shadow_mask_r8 && reprojection_map && out_rg16f && gd->width && gd->height, "null argument");
    KJ_REQUIRE(t->dev->fc_dev, "kj_frame_begin not called");
    KJ_REQUIRE(row_begin < row_end && row_end <= gd->height && (row_begin % 16u) == 0u, "rows must be a non-empty range starting on a 16-row boundary");
    hipStream_t s = (hipStream_t)stream_;
    const int W = int(gd->width), H = int(gd->height), TW = (W + 7) / 8, TH = (H + 3) / 4, GW = (W + 7) / 8, GH = (H + 7) / 8;
    if (W != t->W || H != t->H) { t->surf.clear(); t->W = W; t->H = H; t->flip_accum = t->flip_moments = false; }
    const size_t FB = size_t(W) * H, TB = size_t(TW) * TH;
    void* bitpacked = t->get("bitpacked_shadows_image", TB * 4, s);
    void* moments_out = t->get(t->flip_moments ? "shadow_denoise_moments:1" : "shadow_denoise_moments:0", FB * 8, s);
    void* moments_prev = t->get(t->flip_moments ? "shadow_denoise_moments:0" : "shadow_denoise_moments:1", FB * 8, s);
    t->flip_moments = !t->flip_moments;
    void* accum_out = t->get(t->flip_accum ? "shadow_denoise_accum:1" : "shadow_denoise_accum:0", FB * 4, s);
    void* accum_prev = t->get(t->flip_accum ? "shadow_denoise_accum:0" : "shadow_denoise_accum:1", FB * 4, s);
    t->flip_accum = !t->flip_accum;
    void* spatial_input = t->get("spatial_input_image", FB * 4, s);
    void* metadata = t->get("metadata_image", TB * 4, s);
    void* temp = t->get("temp", FB * 4, s);
    KJ_TRY_HIP(t->err);
    const dim3 blk(64);
    const bool whole = row_begin == 0u && int(row_end) == H;
    // 8-row tile rows [t0, t1) of a pass that over-computes `grow` pixel rows on either side of the strip
    const int tr0 = int(row_begin) / 8, tr1 = (int(row_end) + 7) / 8;
    auto tiles = [&](int grow, int& t0, int& t1) { t0 = whole ? 0 : std::max(0, tr0 - grow / 8); t1 = whole ? GH : std::min(GH, tr1 + grow / 8); };
    int t0, t1;
    const ImgR8 mask = img<uint8_t>(shadow_mask_r8, W, H);
    const ImgF32 depth = img<float>(gd->depth, W, H);
    const ImgU32 gnormal = img<uint32_t>(gd->geometric_normal, W, H);
    tiles(32, t0, t1);
    hipLaunchKernelGGL(k_shadow_bitpack, dim3(GW, t1 - t0), blk, 0, s, mask, img<uint32_t>(bitpacked, TW, TH), t0);
    KJ_CHECK_LAUNCH();
    ShadowTemporalArgs a;
    a.fc = t->dev->fc_dev;
    a.shadow_mask_tex = mask; a.bitpacked_tex = img<uint32_t>(bitpacked, TW, TH); a.prev_moments_tex = img<uint2>(moments_prev, W, H);
    a.prev_accum_tex = img<uint32_t>(accum_prev, W, H); a.reprojection_tex = img<uint2>(reprojection_map, W, H);
    a.output_moments_tex = img<uint2>(moments_out, W, H); a.temporal_output_tex = img<uint32_t>(spatial_input, W, H); a.meta_output_tex = img<uint32_t>(metadata, TW, TH);
    {   // FFX_DNSR_Shadows_KernelWeight: compile-time constants in the shader; same float expressions here
        auto kw = [](float v) { return expf(-3.0f * (v * v) / ((8 + 1.0f) * (8 + 1.0f))); };
        float sum = kw(0);
        for (int c = 1; c <= 8; ++c) sum += 2 * kw(float(c));
        for (int c = 0; c <= 8; ++c) a.kw.w[c] = kw(float(c)) * (1.0f / sum);
    }
    a.W = W; a.H = H; a.TW = TW; a.TH = TH;
    tiles(24, t0, t1);
    a.tile_row0 = t0;
    hipLaunchKernelGGL(k_shadow_temporal, dim3(GW, t1 - t0), blk, 0, s, a);
    KJ_CHECK_LAUNCH();
    const ImgU32 meta = img<uint32_t>(metadata, TW, TH);
    tiles(16, t0, t1);
    hipLaunchKernelGGL(k_shadow_spatial, dim3(GW, t1 - t0), blk, 0, s, img<uint32_t>(spatial_input, W, H), meta, gnormal, depth, img<uint32_t>(accum_out, W, H), 1, t0);
    KJ_CHECK_LAUNCH();
    tiles(8, t0, t1);
    hipLaunchKernelGGL(k_shadow_spatial, dim3(GW, t1 - t0), blk, 0, s, img<uint32_t>(accum_out, W, H), meta, gnormal, depth, img<uint32_t>(temp, W, H), 2, t0);
    KJ_CHECK_LAUNCH();
    tiles(0, t0, t1);
    hipLaunchKernelGGL(k_shadow_spatial, dim3(GW, t1 - t0), blk, 0, s, img<uint32_t>(temp, W, H), meta, gnormal, depth, img<uint32_t>(spatial_input, W, H), 4, t0);
    KJ_CHECK_LAUNCH();
    *out_rg16f = spatial_input;
    return KJ_OK;
}
KjStatus kj_shadow_denoise_surface(KjShadowDenoise* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(t && name && out_dev_ptr && out_bytes, "null argument");
    auto it = t->surf.find(name);
    if (it == t->surf.end()) { set_last_error("no shadow-denoise surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}

}  // extern "C"
