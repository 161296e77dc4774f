// PostProcessRenderer for gfx950 (crates/lib/kajiya/src/renderers/post.rs:10-272):
//   blur_pyramid       post.rs:10-61   mip 0 = rust-shaders/src/blur.rs `blur_cs` (10 vertical taps), mips 1.. = shaders/blur.hlsl (11)
//   luminance histogram post.rs:138-186, shaders/post/luminance_histogram_{clear,calculate,copy}.hlsl; read_back_histogram :188-235
//   rev_blur_pyramid   post.rs:63-110, rust-shaders/src/rev_blur.rs
//   post combine       shaders/post_combine.hlsl (glare 0.05, vignette, display transform, contrast, blue-noise dither)
// and, ahead of it in the frame, motion_blur (renderers/motion_blur.rs + rust-shaders/src/motion_blur.rs), further down in this file.
// Pyramids are B10G11R11_UFLOAT, half-res base, all mip levels minus one; each mip is its own flat surface ("blur_pyramid:<k>").
// Choices where the reference leaves the result undefined are listed in oracle/okj_post.hpp's header and made identically here:
// coarsest rev-blur mip = zeros, out-of-bounds blur fetches = 0 (and counted in the weight), NaN coordinate / NaN->uint = 0.
// The blur keeps the reference's shape — one 64x1 row segment per wave64, the vertical pass staged through LDS (138 columns), the
// horizontal pass out of LDS — because that shape is already the coalesced one: a wave reads 138 consecutive texels per tap row.
#include "kj_host.hpp"
#include "kj_color.hpp"

using namespace kj;

typedef KjFrameConstants FrameConstants;

typedef Img<uint2> ImgU2;
typedef Img<uint32_t> ImgU32;

#define TILE_XY(W_, H_)                                                                    \
    const int lane = threadIdx.x;                                                          \
    const int x = int(blockIdx.x) * 8 + (lane & 7), y = int(blockIdx.y) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < (H_);

KJ_D float gaussian_wt(float dst_px, float src_px) {     // blur.rs:18-22 == blur.hlsl:11-15
    const float px_off = (dst_px + 0.5f) * 2.0f - (src_px + 0.5f);
    const float sigma = 5.0f * 0.5f;
    return expf(-px_off * px_off / (sigma * sigma));
}
KJ_D V3 blur_fetch(const ImgU2& i, int x, int y) { return xyz(unpack_rgba16f(i.ld(x, y))); }
KJ_D V3 blur_fetch(const ImgU32& i, int x, int y) { return unpack_r11g11b10f(i.ld(x, y)); }
typedef Img<float4> ImgF4;     // RGBA32F: the reference path tracer's accumulation image (world_render_passes.rs:294-330 feeds it to post as is)
KJ_D V3 blur_fetch(const ImgF4& i, int x, int y) { const float4 v = i.ld(x, y); return V3{v.x, v.y, v.z}; }

// one blur + 2x downsample pass; VTAPS = 10 for the Rust kernel of mip 0 (`while y < KERNEL_RADIUS * 2`), 11 for blur.hlsl. blur.hlsl computes
// the taps' source coordinates in `uint` (:22,50): a tap left of / above the image sits at 2^32 - k, its weight underflows to 0 (blur.rs: i32).
template <int VTAPS> KJ_D float blur_src_coord(int s) { return VTAPS == 11 ? float(uint32_t(s)) : float(s); }
template <typename SRC, int VTAPS>
__global__ void __launch_bounds__(64) k_post_blur(SRC src, ImgU32 dst) {
    __shared__ float vblur_out[3][138];                       // (group_width + kernel_radius) * 2 columns, SoA: conflict-free
    const int lx = threadIdx.x, gx = blockIdx.x, y = blockIdx.y;
    const int x = gx * 64 + lx;
    for (int xfetch = lx; xfetch < 138; xfetch += 64) {
        const int sx = gx * 128 + xfetch - 5;
        V3 v = v3(0.0f);
        float vw = 0.0f;
#pragma unroll
        for (int yi = 0; yi < VTAPS; ++yi) {
            const int sy = y * 2 - 5 + yi;
            const float wt = gaussian_wt(float(y), blur_src_coord<VTAPS>(sy));
            v += blur_fetch(src, sx, sy) * wt;
            vw += wt;
        }
        v = v / vw;
        vblur_out[0][xfetch] = v.x; vblur_out[1][xfetch] = v.y; vblur_out[2][xfetch] = v.z;
    }
    __syncthreads();
    V3 res = v3(0.0f);
    float wt_sum = 0.0f;
#pragma unroll
    for (int xi = 0; xi <= 10; ++xi) {
        const float wt = gaussian_wt(float(x), blur_src_coord<VTAPS>(x * 2 + xi - 5));
        const int c = lx * 2 + xi;
        res += V3{vblur_out[0][c], vblur_out[1][c], vblur_out[2][c]} * wt;
        wt_sum += wt;
    }
    dst.st(x, y, pack_r11g11b10f(res / wt_sum));
}

// luminance_histogram_calculate.hlsl:15-31; (ew, eh) = the div_up extent post.rs:151-154 dispatches, which can exceed the mip
__global__ void __launch_bounds__(64) k_post_histogram(const FrameConstants* __restrict__ fcp, ImgU32 src, int ew, int eh, uint32_t* __restrict__ histogram) {
    TILE_XY(ew, eh)
    if (!in_image) return;
    const float log_lum = log2f(fmaxf(1e-20f, sRGB_to_luminance(unpack_r11g11b10f(src.ld(x, y))) / fcp->pre_exposure));
    const float t = saturate((log_lum - -16.0f) / (16.0f - -16.0f));
    const uint32_t bin = min(f2u_sat(t * 256.0f), 255u);
    const V2 uv = V2{float(x) + 0.5f, float(y) + 0.5f} / V2{float(ew), float(eh)};
    const float l = length(uv - V2{0.5f, 0.5f});
    const float infl = expf(-8.0f * powf(l, 2.0f));
    atomicAdd(&histogram[bin], f2u_sat(infl * 256.0f));
}

KJ_D V3 sample_r11g11b10f_bilinear_clamp(const ImgU32& i, V2 uv) {    // sampler_lnc
    const float fx = uv.x * float(i.w) - 0.5f, fy = uv.y * float(i.h) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = f2i_sat(x0f), y0 = f2i_sat(y0f);
    const int xa = min(max(x0, 0), i.w - 1), xb = min(max(x0 + 1, 0), i.w - 1), ya = min(max(y0, 0), i.h - 1), yb = min(max(y0 + 1, 0), i.h - 1);
    const V3 s00 = unpack_r11g11b10f(i.p[size_t(ya) * i.w + xa]), s10 = unpack_r11g11b10f(i.p[size_t(ya) * i.w + xb]);
    const V3 s01 = unpack_r11g11b10f(i.p[size_t(yb) * i.w + xa]), s11 = unpack_r11g11b10f(i.p[size_t(yb) * i.w + xb]);
    const V3 a = s00 * (1.0f - tx) + s10 * tx;
    const V3 b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}

// rev_blur.rs:20-72: dst = lerp(3x3 box of bilinear taps of the coarser rev-blur mip, blur-pyramid mip, self_weight * 0.6)
__global__ void __launch_bounds__(64) k_post_rev_blur(ImgU32 tail, ImgU32 src, ImgU32 dst, float self_weight) {
    TILE_XY(dst.w, dst.h)
    if (!in_image) return;
    const V3 pyramid_col = unpack_r11g11b10f(tail.ld(x, y));
    const V2 inv_size = V2{1.0f, 1.0f} / V2{float(dst.w), float(dst.h)};
    V3 self_col = v3(0.0f);
#pragma unroll
    for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
        for (int xx = -1; xx <= 1; ++xx) {
            const V2 uv = (V2{float(x), float(y)} + V2{0.5f, 0.5f} + V2{float(xx), float(yy)}) * inv_size;
            self_col += sample_r11g11b10f_bilinear_clamp(src, uv);
        }
    self_col = self_col / 9.0f;
    dst.st(x, y, pack_r11g11b10f(lerp(self_col, pyramid_col, self_weight * 0.6f)));
}

// post_combine.hlsl:112-191
template <typename SRC>
__global__ void __launch_bounds__(64) k_post_combine(const FrameConstants* __restrict__ fcp, SRC input, ImgU32 glare_tex, const uint32_t* __restrict__ bb_lut,
                                                     const uint32_t* __restrict__ blue_noise, float input_multiplier, float contrast, ImgU32 output) {
    TILE_XY(output.w, output.h)
    if (!in_image) return;
    const V2 uv = V2{float(x) + 0.5f, float(y) + 0.5f} * V2{1.0f / float(output.w), 1.0f / float(output.h)};
    const V3 glare = sample_r11g11b10f_bilinear_clamp(glare_tex, uv);
    V3 col = blur_fetch(input, x, y);
    col = lerp(col, glare, 0.05f);
    col = vmax(v3(0.0f), col);
    col = col * input_multiplier;
    const float l = length(uv - V2{0.5f, 0.5f});
    col = col * expf(-2.0f * powf(l, 3.0f));      // pow(), as the text has it (post_combine.hlsl:156)
    col = display_transform_sRGB(bb_lut, col);
    col = vpow(col, contrast);
    const uint32_t idx = fcp->frame_index;
    const uint32_t bx = (uint32_t(x) + idx * 59u) & 255u, by = (uint32_t(y) + idx * 37u) & 255u;
    const float dither = triangle_remap(float(blue_noise[by * 256u + bx] & 0xffu) / 255.0f);
    col = col + v3(dither / 256.0f);
    output.st(x, y, pack_r11g11b10f(col));
}

// ================================================================== motion_blur (renderers/motion_blur.rs:5-72; rust-shaders/src/motion_blur.rs)
// Four kernels, all Rust in the reference. reprojection_map RGBA16_SNORM + depth R32F at (DW, DH), input / output RGBA16F at (W, H).
// Float -> uint casts saturate (negative / NaN -> 0), out-of-range fetches read 0, the lod-1 taps of the single-mip input read mip 0,
// output alpha = 1 (oracle/okj_post.hpp lists the same choices).
KJ_D void keep_largest(V3& largest, V2 v) { const float m2 = dot(v, v); if (m2 > largest.z) largest = V3{v.x, v.y, m2}; }
__global__ void __launch_bounds__(64) k_velocity_reduce_x(ImgU2 reprojection_map, ImgU32 out) {           // :189-211
    TILE_XY(out.w, out.h)
    if (!in_image) return;
    V3 largest = v3(0.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const V4 v = ld_reproj(reprojection_map, x * 16 + i, y); keep_largest(largest, V2{v.x, v.y}); }
    st2h(out, x, y, V2{largest.x, largest.y});
}
__global__ void __launch_bounds__(64) k_velocity_reduce_y(ImgU32 in, ImgU32 out) {                        // :213-235
    TILE_XY(out.w, out.h)
    if (!in_image) return;
    V3 largest = v3(0.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) keep_largest(largest, ld2h(in, x, y * 16 + i));
    st2h(out, x, y, V2{largest.x, largest.y});
}
__global__ void __launch_bounds__(64) k_velocity_dilate(ImgU32 in, ImgU32 out) {                          // :237-264 (x outer, y inner)
    TILE_XY(out.w, out.h)
    if (!in_image) return;
    V3 largest = v3(0.0f);
    for (int xx = -2; xx <= 2; ++xx)
        for (int yy = -2; yy <= 2; ++yy) keep_largest(largest, ld2h(in, x + xx, y + yy));
    st2h(out, x, y, V2{largest.x, largest.y});
}
KJ_D float mb_depth_to_view_z(float depth, const FrameConstants& fc) { return 1.0f / (depth * -fc.view_constants.clip_to_view[11]); }     // util.rs:69-76
KJ_D float mb_sample_weight(float center_depth, float sample_depth, float offset_len, float center_spread_len, float sample_spread_len, float depth_scale) {   // :18-42
    const float d = sample_depth - center_depth;
    const V2 dc = V2{saturate(0.5f + depth_scale * d), saturate(0.5f + -depth_scale * d)};
    const V2 sc = V2{saturate(center_spread_len - (offset_len + 1.0f)), saturate(sample_spread_len - (offset_len + 1.0f))};
    return dot(dc, sc);
}
KJ_D V4 sample_bilinear_clamp_snorm16(const ImgU2& i, V2 uv) {
    const float fx = uv.x * float(i.w) - 0.5f, fy = uv.y * float(i.h) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = f2i_sat(x0f), y0 = f2i_sat(y0f);
    const int xa = min(max(x0, 0), i.w - 1), xb = min(max(x0 + 1, 0), i.w - 1), ya = min(max(y0, 0), i.h - 1), yb = min(max(y0 + 1, 0), i.h - 1);
    const V4 a = ld_reproj(i, xa, ya) * (1.0f - tx) + ld_reproj(i, xb, ya) * tx;
    const V4 b = ld_reproj(i, xa, yb) * (1.0f - tx) + ld_reproj(i, xb, yb) * tx;
    return a * (1.0f - ty) + b * ty;
}
__global__ void __launch_bounds__(64) k_motion_blur(const FrameConstants* __restrict__ fcp, ImgU2 input, ImgU2 reprojection_map, ImgU32 tile_velocity_tex, Img<float> depth,
                                                    ImgU2 output) {                                      // :47-187
    TILE_XY(output.w, output.h)
    if (!in_image) return;
    const FrameConstants& fc = *fcp;
    const int W = output.w, H = output.h, DW = depth.w, DH = depth.h;
    const V2 depth_tex_size = V2{float(DW), float(DH)}, output_tex_size = V2{float(W), float(H)};
    const float blur_scale = 0.5f * 1.0f;                  // motion_blur_scale = 1.0 (motion_blur.rs:53)
    const V2 uv = V2{float(x) + 0.5f, float(y) + 0.5f} * V2{1.0f / float(W), 1.0f / float(H)};
    int tox, toy, noise1;
    {   // scrambled tile coordinates (:68-84), wrapping i32 arithmetic
        uint32_t ux = uint32_t(x), uy = uint32_t(y);
        ux += ux << 4; ux ^= uint32_t(int32_t(ux) >> 6);
        uy += ux << 1; uy += uy << 6; uy ^= uint32_t(int32_t(uy) >> 2);
        ux ^= uy;
        noise1 = int32_t(ux ^ (uy << 1));
        tox = int32_t(ux & 31u) - 15; toy = int32_t(uy & 31u) - 15;
        noise1 = (noise1 & 31) - 15;
    }
    const V2 tile_coord_f = uv * depth_tex_size + V2{float(tox), float(toy)};
    const uint32_t tcx = min(f2u_sat(tile_coord_f.x), uint32_t(DW - 1)) / 16u, tcy = min(f2u_sat(tile_coord_f.y), uint32_t(DH - 1)) / 16u;
    const V2 tile_velocity = ld2h(tile_velocity_tex, int(tcx), int(tcy)) * blur_scale;
    const int kernel_width = 4;
    const float noise = 0.5f * float(noise1) / 15.0f;
    const float center_offset_len = noise / float(kernel_width) * 0.5f;
    const V2 center_uv = uv + tile_velocity * center_offset_len;
    const V2 cpx = center_uv * output_tex_size;
    const V3 center_color = xyz(ld4(input, int(min(f2u_sat(cpx.x), uint32_t(W - 1))), int(min(f2u_sat(cpx.y), uint32_t(H - 1)))));
    const float center_depth = -mb_depth_to_view_z(sample_nearest_clamp(depth, center_uv), fc);
    const V4 cv = sample_bilinear_clamp_snorm16(reprojection_map, center_uv);
    const V2 center_velocity_px = (V2{cv.x, cv.y} * blur_scale) * depth_tex_size;
    const float soft_z = 16.0f;
    V4 sum = v4(0.0f);
    float sample_count = 1.0f;
    if (length(tile_velocity) > 0.0f) {
        for (int i = 1; i < kernel_width; ++i) {
            const float offset_len0 = (float(i) + noise) / float(kernel_width) * 0.5f;
            const float offset_len1 = (float(-i) + noise) / float(kernel_width) * 0.5f;
            const V2 uv0 = uv + tile_velocity * offset_len0, uv1 = uv + tile_velocity * offset_len1;
            const V2 p0 = uv0 * depth_tex_size, p1 = uv1 * depth_tex_size;
            const int px0 = int(min(f2u_sat(p0.x), 0x7fffffffu)), py0 = int(min(f2u_sat(p0.y), 0x7fffffffu));
            const int px1 = int(min(f2u_sat(p1.x), 0x7fffffffu)), py1 = int(min(f2u_sat(p1.y), 0x7fffffffu));
            const float d0 = -mb_depth_to_view_z(depth.ld(px0, py0), fc), d1 = -mb_depth_to_view_z(depth.ld(px1, py1), fc);
            const V4 r0 = ld_reproj(reprojection_map, px0, py0), r1 = ld_reproj(reprojection_map, px1, py1);
            const float v0 = length((V2{r0.x, r0.y} * blur_scale) * depth_tex_size), v1 = length((V2{r1.x, r1.y} * blur_scale) * depth_tex_size);
            float weight0 = mb_sample_weight(center_depth, d0, length((uv0 - uv) * depth_tex_size), length(center_velocity_px), v0, soft_z);
            float weight1 = mb_sample_weight(center_depth, d1, length((uv1 - uv) * depth_tex_size), length(center_velocity_px), v1, soft_z);
            const bool m0 = d0 > d1, m1 = v1 > v0;
            weight0 = (m0 && m1) ? weight1 : weight0;
            weight1 = (m0 || m1) ? weight1 : weight0;
            const float valid0 = (uv0.x == saturate(uv0.x) && uv0.y == saturate(uv0.y)) ? 1.0f : 0.0f;
            const float valid1 = (uv1.x == saturate(uv1.x) && uv1.y == saturate(uv1.y)) ? 1.0f : 0.0f;
            weight0 *= valid0; weight1 *= valid1;
            sample_count += valid0 + valid1;
            V4 c0 = sample_bilinear_clamp_rgba16f(input.p, W, H, uv0); c0.w = 1.0f;
            sum += c0 * weight0;
            V4 c1 = sample_bilinear_clamp_rgba16f(input.p, W, H, uv1); c1.w = 1.0f;
            sum += c1 * weight1;
        }
        sum = sum * (1.0f / sample_count);
    }
    const V3 result = xyz(sum) + center_color * (1.0f - sum.w);
    st4(output, x, y, v4(result, 1.0f));
}

// ================================================================== host
struct KjMotionBlur {
    KjDevice* dev = nullptr;
    int W = 0, H = 0, DW = 0, DH = 0;
    std::map<std::string, kj::DevBuf> surf;
    hipError_t err = hipSuccess;
    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
};

struct KjPost {
    KjDevice* dev = nullptr;
    int W = 0, H = 0, mip_levels = 0;
    std::map<std::string, kj::DevBuf> surf;
    kj::DevBuf bb_lut;                     // 64 x RG16F
    uint32_t* histogram_host = nullptr;    // pinned; the reference's gpu-to-cpu "luminance histogram" buffer (post.rs:121-129)
    hipError_t err = hipSuccess;
    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
    ~KjPost() { if (histogram_host) (void)hipHostFree(histogram_host); }
};

static int mip_count_1d(uint32_t e) { int n = 0; while (e) { ++n; e >>= 1; } return n; }   // image.rs:35-38

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_post_create(KjDevice* dev, const uint16_t* bezold_brucke_lut_rg16f_64, KjPost** out) {
    KJ_REQUIRE(dev && bezold_brucke_lut_rg16f_64 && out, "null argument");
    KjPost* t = new KjPost();
    t->dev = dev;
    hipError_t e = t->bb_lut.upload(bezold_brucke_lut_rg16f_64, 64 * 4);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->histogram_host, 256 * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipDeviceSynchronize();     // the LUT's host buffer may go away as soon as we return
    if (e != hipSuccess) { delete t; KJ_TRY_HIP(e); }
    memset(t->histogram_host, 0, 256 * sizeof(uint32_t));
    *out = t;
    return KJ_OK;
}
void kj_post_destroy(KjPost* t) { delete t; }

KjStatus kj_post_render(KjPost* t, const void* input, uint32_t input_format, uint32_t width, uint32_t height, float post_exposure_mult, float contrast,
                        const void** out_b10g11r11, void* stream_) {
    KJ_REQUIRE(t && input && out_b10g11r11 && width && height, "null argument");
    KJ_REQUIRE(input_format == KJ_POST_INPUT_RGBA16F || input_format == KJ_POST_INPUT_RGBA32F, "input_format must be KJ_POST_INPUT_RGBA16F or KJ_POST_INPUT_RGBA32F");
    KJ_REQUIRE(t->dev->fc_dev, "kj_frame_begin not called");
    hipStream_t s = (hipStream_t)stream_;
    const int W = int(width), H = int(height), pw = (W + 1) / 2, ph = (H + 1) / 2;
    if (W != t->W || H != t->H) { t->surf.clear(); t->W = W; t->H = H; }
    const int levels = std::max(1, std::max(mip_count_1d(pw), mip_count_1d(ph)) - 1);   // post.rs:11-21
    t->mip_levels = levels;
    const FrameConstants* fc = t->dev->fc_dev;
    auto mw = [&](int l) { return std::max(1, pw >> l); };
    auto mh = [&](int l) { return std::max(1, ph >> l); };
    std::vector<void*> blur(levels), rev(levels);
    for (int l = 0; l < levels; ++l) {
        blur[l] = t->get("blur_pyramid:" + std::to_string(l), size_t(mw(l)) * mh(l) * 4, s);
        rev[l] = t->get("rev_blur_pyramid:" + std::to_string(l), size_t(mw(l)) * mh(l) * 4, s);
    }
    uint32_t* histogram = (uint32_t*)t->get("histogram", 256 * 4, s);
    void* output = t->get("output", size_t(W) * H * 4, s);
    KJ_TRY_HIP(t->err);
    const dim3 blk(64);
    // ---- blur_pyramid
    if (input_format == KJ_POST_INPUT_RGBA32F)
        hipLaunchKernelGGL((k_post_blur<ImgF4, 10>), dim3((mw(0) + 63) / 64, mh(0)), blk, 0, s, img<float4>(input, W, H), img<uint32_t>(blur[0], mw(0), mh(0)));
    else
        hipLaunchKernelGGL((k_post_blur<ImgU2, 10>), dim3((mw(0) + 63) / 64, mh(0)), blk, 0, s, img<uint2>(input, W, H), img<uint32_t>(blur[0], mw(0), mh(0)));
    KJ_CHECK_LAUNCH();
    for (int l = 1; l < levels; ++l) {
        hipLaunchKernelGGL((k_post_blur<ImgU32, 11>), dim3((mw(l) + 63) / 64, mh(l)), blk, 0, s, img<uint32_t>(blur[l - 1], mw(l - 1), mh(l - 1)),
                           img<uint32_t>(blur[l], mw(l), mh(l)));
        KJ_CHECK_LAUNCH();
    }
    // ---- luminance histogram: clear, calculate, copy to the host-visible buffer
    {
        const int l = std::max(0, levels - 7);
        const int ew = std::max(1, (pw + (1 << l) - 1) >> l), eh = std::max(1, (ph + (1 << l) - 1) >> l);
        KJ_TRY_HIP(hipMemsetAsync(histogram, 0, 256 * 4, s));
        hipLaunchKernelGGL(k_post_histogram, dim3((ew + 7) / 8, (eh + 7) / 8), blk, 0, s, fc, img<uint32_t>(blur[l], mw(l), mh(l)), ew, eh, histogram);
        KJ_CHECK_LAUNCH();
        KJ_TRY_HIP(hipMemcpyAsync(t->histogram_host, histogram, 256 * 4, hipMemcpyDeviceToHost, s));
    }
    // ---- rev_blur_pyramid
    KJ_TRY_HIP(hipMemsetAsync(rev[levels - 1], 0, size_t(mw(levels - 1)) * mh(levels - 1) * 4, s));
    for (int target = levels - 2; target >= 0; --target) {
        hipLaunchKernelGGL(k_post_rev_blur, dim3((mw(target) + 7) / 8, (mh(target) + 7) / 8), blk, 0, s, img<uint32_t>(blur[target], mw(target), mh(target)),
                           img<uint32_t>(rev[target + 1], mw(target + 1), mh(target + 1)), img<uint32_t>(rev[target], mw(target), mh(target)), 0.5f);
        KJ_CHECK_LAUNCH();
    }
    // ---- post combine
    if (input_format == KJ_POST_INPUT_RGBA32F)
        hipLaunchKernelGGL((k_post_combine<ImgF4>), dim3((W + 7) / 8, (H + 7) / 8), blk, 0, s, fc, img<float4>(input, W, H), img<uint32_t>(rev[0], mw(0), mh(0)),
                           (const uint32_t*)t->bb_lut.p, (const uint32_t*)t->dev->blue_noise.p, post_exposure_mult, contrast, img<uint32_t>(output, W, H));
    else
        hipLaunchKernelGGL((k_post_combine<ImgU2>), dim3((W + 7) / 8, (H + 7) / 8), blk, 0, s, fc, img<uint2>(input, W, H), img<uint32_t>(rev[0], mw(0), mh(0)),
                           (const uint32_t*)t->bb_lut.p, (const uint32_t*)t->dev->blue_noise.p, post_exposure_mult, contrast, img<uint32_t>(output, W, H));
    KJ_CHECK_LAUNCH();
    *out_b10g11r11 = output;
    return KJ_OK;
}

// PostProcessRenderer::read_back_histogram (post.rs:188-235), f64 as there
KjStatus kj_luminance_histogram_mean_log2(const uint32_t* histogram, float clipping_low, float clipping_high, float* out_image_log2_lum) {
    KJ_REQUIRE(histogram && out_image_log2_lum, "null argument");
    // Rust's `as u32` saturates and maps NaN to 0; a C++ cast of a negative or NaN double is undefined: clamp first
    auto frac01 = [](double v) { return v >= 0.0 ? std::min(v, 1.0) : 0.0; };   // NaN -> 0
    const double outlier_frac_lo = frac01(double(clipping_low));
    const double outlier_frac_hi = std::min(frac01(double(clipping_high)), 1.0 - outlier_frac_lo);
    uint32_t total = 0;
    for (int i = 0; i < 256; ++i) total += histogram[i];
    const uint32_t reject_lo = uint32_t(double(total) * outlier_frac_lo);
    const uint32_t to_use = uint32_t(double(total) * (1.0 - outlier_frac_lo - outlier_frac_hi));
    double sum = 0.0;
    uint32_t used = 0, left_to_reject = reject_lo, left_to_use = to_use;
    for (int i = 0; i < 256; ++i) {
        const double tt = (double(i) + 0.5) / 256.0;
        const uint32_t count = histogram[i];
        const uint32_t count_to_use = std::min(count > left_to_reject ? count - left_to_reject : 0u, left_to_use);
        left_to_reject = left_to_reject > count ? left_to_reject - count : 0u;
        left_to_use = left_to_use > count_to_use ? left_to_use - count_to_use : 0u;
        sum += tt * double(count_to_use);
        used += count_to_use;
    }
    const double mean = sum / double(std::max(used, 1u));
    *out_image_log2_lum = float(-16.0 + mean * (16.0 - -16.0));
    return KJ_OK;
}
// Reads the host-visible histogram as it is -- like the reference's mapped buffer it holds the last copy that has completed.
KjStatus kj_post_read_back_histogram(KjPost* t, float clipping_low, float clipping_high, float* out_image_log2_lum, uint32_t* out_histogram256) {
    KJ_REQUIRE(t && out_image_log2_lum, "null argument");
    uint32_t histogram[256];
    memcpy(histogram, t->histogram_host, sizeof histogram);
    if (out_histogram256) memcpy(out_histogram256, histogram, sizeof histogram);
    return kj_luminance_histogram_mean_log2(histogram, clipping_low, clipping_high, out_image_log2_lum);
}

KjStatus kj_post_surface(KjPost* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(t && name && out_dev_ptr && out_bytes, "null argument");
    auto it = t->surf.find(name);
    if (it == t->surf.end()) { set_last_error("no post surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}
KjStatus kj_post_mip_levels(KjPost* t, uint32_t* out_levels) {
    KJ_REQUIRE(t && out_levels, "null argument");
    *out_levels = uint32_t(t->mip_levels);
    return KJ_OK;
}

KjStatus kj_motion_blur_create(KjDevice* dev, KjMotionBlur** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjMotionBlur* t = new KjMotionBlur();
    t->dev = dev;
    *out = t;
    return KJ_OK;
}
void kj_motion_blur_destroy(KjMotionBlur* t) { delete t; }

// motion_blur(rg, input, depth, reprojection_map) -> Handle<Image>   (renderers/motion_blur.rs:5-72)
KjStatus kj_motion_blur_render(KjMotionBlur* t, const void* input_rgba16f, uint32_t width, uint32_t height, const void* depth_r32f, const void* reprojection_map,
                               uint32_t depth_width, uint32_t depth_height, const void** out_rgba16f, void* stream_) {
    KJ_REQUIRE(t && input_rgba16f && depth_r32f && reprojection_map && out_rgba16f && width && height && depth_width && depth_height, "null argument");
    KJ_REQUIRE(t->dev->fc_dev, "kj_frame_begin not called");
    hipStream_t s = (hipStream_t)stream_;
    const int W = int(width), H = int(height), DW = int(depth_width), DH = int(depth_height);
    if (W != t->W || H != t->H || DW != t->DW || DH != t->DH) { t->surf.clear(); t->W = W; t->H = H; t->DW = DW; t->DH = DH; }
    const int tw = (DW + 15) / 16, th = (DH + 15) / 16;                   // VELOCITY_TILE_SIZE = 16
    void* reduced_x = t->get("velocity_reduced_x", size_t(tw) * DH * 4, s);
    void* reduced_y = t->get("velocity_reduced_y", size_t(tw) * th * 4, s);
    void* dilated = t->get("velocity_dilated", size_t(tw) * th * 4, s);
    void* output = t->get("output", size_t(W) * H * 8, s);
    KJ_TRY_HIP(t->err);
    const dim3 blk(64);
    const ImgU2 reproj = img<uint2>(reprojection_map, DW, DH);
    hipLaunchKernelGGL(k_velocity_reduce_x, dim3((tw + 7) / 8, (DH + 7) / 8), blk, 0, s, reproj, img<uint32_t>(reduced_x, tw, DH));
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_velocity_reduce_y, dim3((tw + 7) / 8, (th + 7) / 8), blk, 0, s, img<uint32_t>(reduced_x, tw, DH), img<uint32_t>(reduced_y, tw, th));
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_velocity_dilate, dim3((tw + 7) / 8, (th + 7) / 8), blk, 0, s, img<uint32_t>(reduced_y, tw, th), img<uint32_t>(dilated, tw, th));
    KJ_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_motion_blur, dim3((W + 7) / 8, (H + 7) / 8), blk, 0, s, t->dev->fc_dev, img<uint2>(input_rgba16f, W, H), reproj, img<uint32_t>(dilated, tw, th),
                       img<float>(depth_r32f, DW, DH), img<uint2>(output, W, H));
    KJ_CHECK_LAUNCH();
    *out_rgba16f = output;
    return KJ_OK;
}
KjStatus kj_motion_blur_surface(KjMotionBlur* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(t && name && out_dev_ptr && out_bytes, "null argument");
    auto it = t->surf.find(name);
    if (it == t->surf.end()) { set_last_error("no motion-blur surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}

}  // extern "C"
