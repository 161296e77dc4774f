// SsgiRenderer for gfx950 (renderers/ssgi.rs:25-181; assets/shaders/ssgi/{ssgi,spatial_filter,upsample,temporal_filter}.hlsl)
// with the shipped switches: USE_AO_ONLY 1 (the colour accumulation never reaches the output, so prev_radiance is not
// read), 6 half-samples per direction, 60 px kernel, no random jitter. Output: the R8_UNORM full-res "ssao" guide that
// rtdgi's resolve / filters take as `ssao_tex` (world_render_passes.rs:90-96,156). 8x8 tile = one wave64.
#include "kj_host.hpp"
#include "kj_shading.hpp"
#include "kj_screen.hpp"

using namespace kj;

typedef Img<uint4> ImgU4;
typedef Img<uint2> ImgU2;
typedef Img<uint32_t> ImgU32;
typedef Img<uint16_t> ImgH1;
typedef Img<float> ImgF32;
typedef Img<uint8_t> ImgR8;

#define TILE_XY(W_, H_) TILE_XY_M(W_, H_, KJ_TILES_PLAIN)
#define TILE_XY_M(W_, H_, MODE_)                                                           \
    const int lane = threadIdx.x;                                                          \
    const uint2 kj_tb = kj::tile_order<MODE_>();                                           \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = row0 + int(kj_tb.y) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < ((H_) < row1 ? (H_) : row1);      /* rows [row0, row1) of the image: the launch covers just those tiles */

// GbufferDepth::half_view_normal / half_depth (renderers/mod.rs:31-71; extract_half_res_{gbuffer_view_normal_rgba8,depth}.hlsl)
__global__ void __launch_bounds__(64) k_ssgi_extract_half(const FrameConstants* __restrict__ fcp, ImgU4 gbuffer, ImgF32 depth, ImgU32 half_view_normal, ImgF32 half_depth, int row0, int row1) {
    TILE_XY_M(half_depth.w, half_depth.h, KJ_TILES_ROWS)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const I2 off = halfres_subsample_offset(fc.frame_index);
    const int sx = x * 2 + off.x, sy = y * 2 + off.y;
    const V3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(sx, sy).y);
    const V3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, v4(normal_ws, 0))));
    half_view_normal.st(x, y, pack_rgba8_snorm(v4(normal_vs, 1.0f)));
    half_depth.st(x, y, depth.ld(sx, sy));
}

// ssgi.hlsl:51-62
KJ_D float ssgi_fast_sqrt(float v) { return __uint_as_float(0x1fbd1df5u + (__float_as_uint(v) >> 1u)); }
KJ_D float ssgi_fast_acos(float in_x) {
    const float ax = fabsf(in_x);
    float res = -0.156583f * ax + 1.57079632679f;
    res *= ssgi_fast_sqrt(1.0f - ax);
    return in_x >= 0 ? res : 3.14159265359f - res;
}
// cos / sin through cos_sin_turns_fast (kj_screen.hpp: quadrant reduction + two short polynomials, < 1 ulp): libm's cosf / sinf are ~115
// VALU instructions each on gfx950 and this kernel evaluated ten of them per pixel
KJ_D float ssgi_integrate_arc(float h1, float h2, float n) {
    const V2 csn = cos_sin_turns_fast(n);
    const float a = -cos_sin_turns_fast(2.0f * h1 - n).x + csn.x + 2.0f * h1 * csn.y;
    const float b = -cos_sin_turns_fast(2.0f * h2 - n).x + csn.x + 2.0f * h2 * csn.y;
    return 0.25f * (a + b);
}
KJ_D float ssgi_update_horizon(float prev, float cur, float blend) { return cur > prev ? lerp(prev, cur, blend) : prev; }
// ssgi.hlsl:174-222 without the colour path
KJ_D float ssgi_process_sample(const FrameConstants& fc, V4 sample_cs, V3 center_vs, V3 v_vs, float kernel_radius_ws, float theta_cos_max) {
    if (sample_cs.z > 0) {
        const V4 sample_vs4 = mul44(fc.view_constants.sample_to_view, sample_cs);
        const V3 sample_vs = xyz(sample_vs4) / sample_vs4.w;
        const V3 off = sample_vs - center_vs;
        const float len = length(off);
        const float sample_theta_cos = dot(off, v_vs) / len;
        const float dn = len / kernel_radius_ws;
        if (dn < 1.0f) theta_cos_max = ssgi_update_horizon(theta_cos_max, sample_theta_cos, smoothstep(1.0f, 0.0f, dn));
    } else {
        theta_cos_max = ssgi_update_horizon(theta_cos_max, -1.0f, 1.0f);
    }
    return theta_cos_max;
}

// "ssao" (ssgi.hlsl:230-341), half res, R16F
__global__ void __launch_bounds__(64) k_ssgi(const FrameConstants* __restrict__ fcp, ImgU4 gbuffer, ImgF32 half_depth, ImgH1 output_tex, int W, int H, int row0, int row1) {
    const int hw = output_tex.w, hh = output_tex.h;
    TILE_XY(hw, hh)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const V4 input_tex_size = tex_size4(W, H), output_tex_size = tex_size4(hw, hh);
    const V2 uv = get_uv(float(x), float(y), output_tex_size);
    const float d = half_depth.ld(x, y);
    if (d == 0.0f) { output_tex.st(x, y, f32_to_f16(0.0f)); return; }
    const GbufferData g = gbuffer_unpack(gbuffer.ld(x * 2, y * 2));
    const V3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, v4(g.normal, 0))));
    const ViewRay vrc = view_ray_from_uv_and_depth(fc, uv, d);
    // ray_dir_vs: normalize(sample_to_view * (cs, 0, 1)).xyz
    const V2 cs = uv_to_cs(uv);
    const V3 ray_dir_vs = normalize(xyz(mul44(fc.view_constants.sample_to_view, V4{cs.x, cs.y, 0.0f, 1.0f})));
    const V3 v_vs = -normalize(ray_dir_vs);
    const V3 ray_hit_vs = vrc.hit_vs;
    const uint32_t ux = uint32_t(x), uy = uint32_t(y);
    const float temporal_rotations[6] = {60.0f, 300.0f, 180.0f, 240.0f, 120.0f, 0.0f};
    const float temporal_offsets[4] = {0.0f, 0.5f, 0.25f, 0.75f};
    const float spatial_direction_noise = 1.0f / 16.0f * float((((ux + uy) & 3u) << 2) + (ux & 3u));
    const float temporal_direction_noise = temporal_rotations[fc.frame_index % 6] / 360.0f;
    const float spatial_offset_noise = (1.0f / 4.0f) * float((uy - ux) & 3u);
    const float temporal_offset_noise = temporal_offsets[fc.frame_index / 6 % 4];
    const float ss_angle = frac(spatial_direction_noise + temporal_direction_noise) * 3.14159265359f;
    const float rand_offset = frac(spatial_offset_noise + temporal_offset_noise);
    const V2 cs_ss = cos_sin_turns_fast(ss_angle);
    V2 cs_slice_dir{cs_ss.x * input_tex_size.y / input_tex_size.x, cs_ss.y};
    float kernel_radius_ws, kernel_radius_shrinkage;
    {
        const float ws_to_cs = 0.5f / -ray_hit_vs.z * fc.view_constants.view_to_clip[5];
        const float cs_kernel_radius_scaled = 60.0f * output_tex_size.w;
        kernel_radius_ws = cs_kernel_radius_scaled / ws_to_cs;
        cs_slice_dir = cs_slice_dir * cs_kernel_radius_scaled;
        kernel_radius_shrinkage = fminf(1.0f, 0.4f / cs_kernel_radius_scaled);
    }
    cs_slice_dir = cs_slice_dir * kernel_radius_shrinkage;
    kernel_radius_ws *= kernel_radius_shrinkage;
    const V3 center_vs = ray_hit_vs;
    cs_slice_dir = cs_slice_dir * (1.0f / 6.0f);
    const float* s2v = fc.view_constants.sample_to_view;   // column-major: M[r][c] = s2v[c * 4 + r]; row vector times matrix below
    const V2 vs_slice_dir{cs_slice_dir.x * s2v[0] + cs_slice_dir.y * s2v[1], cs_slice_dir.x * s2v[4] + cs_slice_dir.y * s2v[5]};
    const V3 slice_normal_vs = normalize(cross(v_vs, V3{vs_slice_dir.x, vs_slice_dir.y, 0}));
    V3 proj_normal_vs = normal_vs - slice_normal_vs * dot(slice_normal_vs, normal_vs);
    const float slice_contrib_weight = length(proj_normal_vs);
    proj_normal_vs = proj_normal_vs / slice_contrib_weight;
    const float sd = dot(vs_slice_dir, V2{proj_normal_vs.x - v_vs.x, proj_normal_vs.y - v_vs.y});
    const float sgn = sd > 0 ? 1.0f : (sd < 0 ? -1.0f : 0.0f);
    const float n_angle = ssgi_fast_acos(clampf(dot(proj_normal_vs, v_vs), -1.0f, 1.0f)) * sgn;
    float theta_cos_max1 = cos_sin_turns_fast(n_angle - 1.57079632679f).x;
    float theta_cos_max2 = cos_sin_turns_fast(n_angle + 1.57079632679f).x;
    int pc0x = x, pc0y = y, pc1x = x, pc1y = y;
    const V2 hit_cs{vrc.hit_cs.x, vrc.hit_cs.y};
    // (round 6: the twelve depths along the slice depend on the geometry only: requested together, then the two marches run on registers in the text's order --
    // the loop as the text has it is twelve round trips to memory one after the other)
    float md[12]; bool md_in[12];
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) {
        const float t0 = float(i) + rand_offset, t1 = float(i) + (1.0f - rand_offset);
        const V2 suv0 = cs_to_uv(V2{hit_cs.x - cs_slice_dir.x * t0, hit_cs.y - cs_slice_dir.y * t0}), suv1 = cs_to_uv(V2{hit_cs.x + cs_slice_dir.x * t1, hit_cs.y + cs_slice_dir.y * t1});
        md[2 * i] = half_depth.ld_raw(int(output_tex_size.x * suv0.x), int(output_tex_size.y * suv0.y), md_in[2 * i]);
        md[2 * i + 1] = half_depth.ld_raw(int(output_tex_size.x * suv1.x), int(output_tex_size.y * suv1.y), md_in[2 * i + 1]);
    }
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) {
        {
            const float t = float(i) + rand_offset;
            V4 sample_cs{hit_cs.x - cs_slice_dir.x * t, hit_cs.y - cs_slice_dir.y * t, 0, 1};
            const V2 suv = cs_to_uv(V2{sample_cs.x, sample_cs.y});
            const int spx = int(output_tex_size.x * suv.x), spy = int(output_tex_size.y * suv.y);
            if (spx != pc0x || spy != pc0y) {
                pc0x = spx; pc0y = spy;
                sample_cs.z = md_in[2 * i] ? md[2 * i] : 0.0f;
                theta_cos_max1 = ssgi_process_sample(fc, sample_cs, center_vs, v_vs, kernel_radius_ws, theta_cos_max1);
            }
        }
        {
            const float t = float(i) + (1.0f - rand_offset);
            V4 sample_cs{hit_cs.x + cs_slice_dir.x * t, hit_cs.y + cs_slice_dir.y * t, 0, 1};
            const V2 suv = cs_to_uv(V2{sample_cs.x, sample_cs.y});
            const int spx = int(output_tex_size.x * suv.x), spy = int(output_tex_size.y * suv.y);
            if (spx != pc1x || spy != pc1y) {
                pc1x = spx; pc1y = spy;
                sample_cs.z = md_in[2 * i + 1] ? md[2 * i + 1] : 0.0f;
                theta_cos_max2 = ssgi_process_sample(fc, sample_cs, center_vs, v_vs, kernel_radius_ws, theta_cos_max2);
            }
        }
    }
    const float h1 = -ssgi_fast_acos(theta_cos_max1);
    const float h2 = +ssgi_fast_acos(theta_cos_max2);
    const float h1p = n_angle + fmaxf(h1 - n_angle, -1.57079632679f);
    const float h2p = n_angle + fminf(h2 - n_angle, 1.57079632679f);
    const float inv_ao = ssgi_integrate_arc(h1p, h2p, n_angle);
    const float col = fmaxf(0.0f, inv_ao) * slice_contrib_weight;
    output_tex.st(x, y, f32_to_f16(fmaxf(0.0f, col)));
}

// "ssao spatial" (spatial_filter.hlsl), half res
__global__ void __launch_bounds__(64) k_ssgi_spatial(ImgH1 ssgi_tex, ImgF32 half_depth, ImgU32 half_view_normal, ImgH1 output_tex, int row0, int row1) {
    TILE_XY_M(output_tex.w, output_tex.h, KJ_TILES_ROWS)
    if (!in_image) return;
    float result = 0, w_sum = 0;
    // (round 6: the 3 x 3 neighbourhood's {depth, value, normal} -- one in-bounds flag per texel, the three images share the half-res extent -- requested together:
    // 27 loads in flight instead of up to 19 round trips one after the other)
    bool t_in[9];
    float d_raw[9]; uint16_t s_raw[9]; uint32_t n_raw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int sx = x + (i % 3 - 1), sy = y + (i / 3 - 1);
        d_raw[i] = half_depth.ld_raw(sx, sy, t_in[i]);
        s_raw[i] = ssgi_tex.ld_raw(sx, sy, t_in[i]);
        n_raw[i] = half_view_normal.ld_raw(sx, sy, t_in[i]);
    }
    const float center_depth = t_in[4] ? d_raw[4] : 0.0f;
    if (center_depth != 0.0f) {
        const V3 center_normal = xyz(unpack_rgba8_snorm(t_in[4] ? n_raw[4] : 0u));
        w_sum = 1.0f;
        result = f16_to_f32(t_in[4] ? s_raw[4] : uint16_t(0));
#pragma unroll
        for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
            for (int xx = -1; xx <= 1; ++xx) {
                if (xx == 0 && yy == 0) continue;
                const int i = (yy + 1) * 3 + (xx + 1);
                const float sdp = t_in[i] ? d_raw[i] : 0.0f;
                if (sdp == 0.0f) continue;
                const float s = f16_to_f32(t_in[i] ? s_raw[i] : uint16_t(0));
                const V3 n = xyz(unpack_rgba8_snorm(t_in[i] ? n_raw[i] : 0u));
                const float depth_diff = 1.0f - (center_depth / sdp);
                const float depth_factor = exp2f(-200.0f * fabsf(depth_diff));
                float nf = fmaxf(0.0f, dot(n, center_normal));
                nf *= nf; nf *= nf;
                float w = 1;
                w *= depth_factor;
                w *= nf;
                w_sum += w;
                result += s * w;
            }
    }
    output_tex.st(x, y, f32_to_f16(result / fmaxf(w_sum, 1e-5f)));
}

// "ssao upsample" (upsample.hlsl), full res, R16F
__global__ void __launch_bounds__(64) k_ssgi_upsample(ImgH1 ssgi_tex, ImgF32 depth, ImgH1 output_tex, int row0, int row1) {
    TILE_XY_M(output_tex.w, output_tex.h, KJ_TILES_ROWS)
    if (!in_image) return;
    float result = 0, w_sum = 0;
    // (round 6: the pixel's depth and its nine taps' {depth, value} are requested together -- the loop as the text has it waits for a tap's depth before it asks for its
    // value, 19 round trips to memory one after the other; same values, same arithmetic)
    bool c_in, d_in[9], s_in[9];
    const float center_raw = depth.ld_raw(x, y, c_in);
    float d_raw[9]; uint16_t s_raw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int sx = x / 2 + (i % 3 - 1), sy = y / 2 + (i / 3 - 1);
        d_raw[i] = depth.ld_raw(sx * 2, sy * 2, d_in[i]);
        s_raw[i] = ssgi_tex.ld_raw(sx, sy, s_in[i]);
    }
    const float center_depth = c_in ? center_raw : 0.0f;
    if (center_depth != 0.0f) {
#pragma unroll
        for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
            for (int xx = -1; xx <= 1; ++xx) {
                const int i = (yy + 1) * 3 + (xx + 1);
                const float sdp = d_in[i] ? d_raw[i] : 0.0f;
                if (sdp == 0.0f) continue;
                const float s = f16_to_f32(s_in[i] ? s_raw[i] : uint16_t(0));
                const float depth_diff = 1.0f - (center_depth / sdp);
                float w = 1;
                w *= exp2f(-200.0f * fabsf(depth_diff));
                w *= expf(-float(xx * xx + yy * yy));
                w_sum += w;
                result += s * w;
            }
    }
    if (w_sum > 1e-6f) output_tex.st(x, y, f32_to_f16(result / w_sum));
    else output_tex.st(x, y, s_in[4] ? s_raw[4] : uint16_t(0));      // ssgi_tex at (x / 2, y / 2): the centre tap
}

// "ssao temporal" (temporal_filter.hlsl), full res; history R16F, final R8_UNORM
KJ_D float sample_bilinear_clamp_r16f(const uint16_t* __restrict__ p, int w, int h, V2 uv) {
    const float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = int(x0f), y0 = int(y0f);
    const int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1), ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
    const float s00 = f16_to_f32(p[size_t(ya) * w + xa]), s10 = f16_to_f32(p[size_t(ya) * w + xb]);
    const float s01 = f16_to_f32(p[size_t(yb) * w + xa]), s11 = f16_to_f32(p[size_t(yb) * w + xb]);
    const float a = s00 * (1.0f - tx) + s10 * tx, b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}
__global__ void __launch_bounds__(64) k_ssgi_temporal(ImgH1 input_tex, ImgH1 history_tex, ImgU2 reprojection_tex, ImgR8 final_output_tex, ImgH1 history_output_tex, int row0, int row1) {
    const int W = final_output_tex.w, H = final_output_tex.h;
    TILE_XY_M(W, H, KJ_TILES_ROWS)
    if (!in_image) return;
    const V2 uv = get_uv(float(x), float(y), tex_size4(W, H));
    const float center = f16_to_f32(input_tex.ld(x, y));
    const V4 reproj = ld_reproj(reprojection_tex, x, y);
    const float history = sample_bilinear_clamp_r16f(history_tex.p, W, H, uv + V2{reproj.x, reproj.y});
    float vsum = 0, vsum2 = 0, wsum = 0;
#pragma unroll
    for (int yy = -2; yy <= 2; ++yy)
#pragma unroll
        for (int xx = -2; xx <= 2; ++xx) {
            const float neigh = f16_to_f32(input_tex.ld(x + xx * 2, y + yy * 2));
            const float w = expf(-3.0f * float(xx * xx + yy * yy) / float((2 + 1.) * (2 + 1.)));
            vsum += neigh * w;
            vsum2 += neigh * neigh * w;
            wsum += w;
        }
    const float ex = vsum / wsum, ex2 = vsum2 / wsum;
    const float dev = sqrtf(fmaxf(0.0f, ex2 - ex * ex));
    const float box_size = 0.5f, n_deviations = 5.0f;
    const float nmin = lerp(center, ex, box_size * box_size) - dev * box_size * n_deviations;
    const float nmax = lerp(center, ex, box_size * box_size) + dev * box_size * n_deviations;
    const float clamped_history = clampf(history, nmin, nmax);
    const float res = lerp(clamped_history, center, 1.0f / 8.0f);
    history_output_tex.st(x, y, f32_to_f16(res));
    final_output_tex.st(x, y, to_unorm8(res));
}

// ================================================================== host
struct KjSsgi {
    KjDevice* dev = nullptr;
    int W = 0, H = 0;
    std::map<std::string, kj::DevBuf> surf;
    bool flip = false;
    hipError_t err = hipSuccess;
    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
};

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_ssgi_create(KjDevice* dev, KjSsgi** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjSsgi* t = new KjSsgi();
    t->dev = dev;
    *out = t;
    return KJ_OK;
}
void kj_ssgi_destroy(KjSsgi* t) { delete t; }

// SsgiRenderer::render(rg, gbuffer_depth, reprojection_map, prev_radiance, bindless_descriptor_set) -> ssgi_tex (ssgi.rs:25-81)
KjStatus kj_ssgi_render(KjSsgi* t, const KjGbufferDepth* gd, const void* reprojection_map, const void* prev_radiance, const void** out_ssao_r8, void* stream_) {
    KJ_REQUIRE(gd, "null argument");
    return kj_ssgi_render_rows(t, gd, reprojection_map, prev_radiance, 0u, gd->height, out_ssao_r8, stream_);
}
// The guide for full-res rows [row_begin, row_end) (row_begin a multiple of 16): the screen-tile split computes it strip by strip. Every pass runs on
// the rows the next one reaches into, so nothing but the temporal pass' history (read through the motion vectors) comes from outside the strip:
//   temporal [r0, r1) reads the upsampled image +-4 rows -> upsample [r0 - 8, r1 + 8) reads the filtered half-res image +-1 -> spatial
//   [r0/2 - 8, r1/2 + 8) reads ssgi_tex +-1 -> k_ssgi [r0/2 - 16, r1/2 + 16) reads half_depth along its slices, <= 70 rows away -> extract +-88.
KjStatus kj_ssgi_render_rows(KjSsgi* t, const KjGbufferDepth* gd, const void* reprojection_map, const void* prev_radiance, uint32_t row_begin, uint32_t row_end,
                             const void** out_ssao_r8, void* stream_) {
    KJ_REQUIRE(t && gd && gd->gbuffer && gd->depth && reprojection_map && out_ssao_r8 && gd->width && gd->height, "null argument");
    KJ_REQUIRE(t->dev->fc_dev, "kj_frame_begin not called");
    KJ_REQUIRE(row_begin < row_end && row_end <= gd->height && (row_begin % 16u) == 0u, "rows must be a non-empty range starting on a 16-row boundary");
    (void)prev_radiance;   // only feeds the colour accumulation, which USE_AO_ONLY discards (ssgi.hlsl:318-324)
    hipStream_t s = (hipStream_t)stream_;
    const int W = int(gd->width), H = int(gd->height), hw = (W + 1) / 2, hh = (H + 1) / 2;
    if (W != t->W || H != t->H) { t->surf.clear(); t->W = W; t->H = H; t->flip = false; }
    const FrameConstants* fc = t->dev->fc_dev;
    const size_t FB = size_t(W) * H, HB = size_t(hw) * hh;
    void* half_view_normal = t->get("half_view_normal_tex", HB * 4, s);
    void* half_depth = t->get("half_depth_tex", HB * 4, s);
    void* ssgi_tex = t->get("ssgi_tex", HB * 2, s);
    void* spatial = t->get("spatially_filtered_tex", HB * 2, s);
    void* upsampled = t->get("upsampled_tex", FB * 2, s);
    void* hist_out = t->get(t->flip ? "ssgi:1" : "ssgi:0", FB * 2, s);
    void* hist = t->get(t->flip ? "ssgi:0" : "ssgi:1", FB * 2, s);
    t->flip = !t->flip;
    // the guide is double-buffered: a host that overlaps frames (GpuPipeline.frame_pipelined) lets frame N's spatial filter read
    // guide N while frame N+1's ssgi pass is already writing guide N+1
    void* final_out = t->get(t->flip ? "filtered_output_tex:0" : "filtered_output_tex:1", FB, s);
    KJ_TRY_HIP(t->err);
    const dim3 blk(64);
    const ImgU4 gbuffer = img<uint4>(gd->gbuffer, W, H);
    const ImgF32 depth = img<float>(gd->depth, W, H);
    const bool whole = row_begin == 0u && int(row_end) == H;
    const int r0 = int(row_begin), r1 = int(row_end), h0 = r0 / 2, h1 = r1 == H ? hh : r1 / 2;
    auto rows = [&](int a, int b, int limit, int& o0, int& o1) { o0 = whole ? 0 : std::max(0, a); o1 = whole ? limit : std::min(limit, b); };
    auto grid = [&](int width, int a, int b) { return dim3((width + 7) / 8, (b - a + 7) / 8); };
    int a, b;
    rows(h0 - 88, h1 + 88, hh, a, b);
    hipLaunchKernelGGL(k_ssgi_extract_half, grid(hw, a, b), blk, 0, s, fc, gbuffer, depth, img<uint32_t>(half_view_normal, hw, hh), img<float>(half_depth, hw, hh), a, b);
    KJ_CHECK_LAUNCH();
    rows(h0 - 16, h1 + 16, hh, a, b);
    hipLaunchKernelGGL(k_ssgi, grid(hw, a, b), blk, 0, s, fc, gbuffer, img<float>(half_depth, hw, hh), img<uint16_t>(ssgi_tex, hw, hh), W, H, a, b);
    KJ_CHECK_LAUNCH();
    rows(h0 - 8, h1 + 8, hh, a, b);
    hipLaunchKernelGGL(k_ssgi_spatial, grid(hw, a, b), blk, 0, s, img<uint16_t>(ssgi_tex, hw, hh), img<float>(half_depth, hw, hh), img<uint32_t>(half_view_normal, hw, hh), img<uint16_t>(spatial, hw, hh), a, b);
    KJ_CHECK_LAUNCH();
    rows(r0 - 8, r1 + 8, H, a, b);
    hipLaunchKernelGGL(k_ssgi_upsample, grid(W, a, b), blk, 0, s, img<uint16_t>(spatial, hw, hh), depth, img<uint16_t>(upsampled, W, H), a, b);
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ssgi_temporal, grid(W, r0, r1), blk, 0, s, img<uint16_t>(upsampled, W, H), img<uint16_t>(hist, W, H), img<uint2>(reprojection_map, W, H), img<uint8_t>(final_out, W, H),
                       img<uint16_t>(hist_out, W, H), r0, r1);
    KJ_CHECK_LAUNCH();
    *out_ssao_r8 = final_out;
    return KJ_OK;
}
KjStatus kj_ssgi_surface(KjSsgi* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(t && name && out_dev_ptr && out_bytes, "null argument");
    auto it = t->surf.find(name);
    if (it == t->surf.end()) { set_last_error("no ssgi surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}

}  // extern "C"
