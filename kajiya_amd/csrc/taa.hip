// TaaRenderer for gfx950 (renderers/taa.rs:41-191; assets/shaders/taa/*.hlsl, inc/unjitter_taa.hlsl,
// inc/image.hlsl). Seven kernels as in the reference; 8x8 tile = one wave64 (the velocity-dilation vote
// uses __shfl_xor like WaveReadLaneAt in reproject_history.hlsl:80-82).
#include "kj_host.hpp"
#include "kj_shading.hpp"
#include "kj_screen.hpp"
#include <cstdlib>

using namespace kj;

typedef Img<uint2> ImgH4;      // RGBA16F (and RGBA16_SNORM)
typedef Img<uint32_t> ImgH2;   // RG16F
typedef Img<uint16_t> ImgH1;   // R16F
typedef Img<float> ImgF32;

// These kernels are VALU-bound (PMC: VALUBusy 82-97 % at full lane utilisation), so they are written for few instructions: an IEEE
// division is ~12 VALU instructions on gfx950, sqrtf 18, libm exp2f / log2f ~10 (v_exp_f32 / v_log_f32 wrapped for denormals).
// One stage is hypersensitive, and that decides what may change where: input_prob divides the squared difference of the filtered
// input and the filtered history by a variance floored at 1e-6, so ONE fp16 ulp in either image moves a texel's probability by O(1)
// in flat regions. Everything upstream of it (history reprojection, the input / history filters, their colour decode) therefore
// keeps the reference's operations and (nearly always) their bits, and only takes transformations that cannot change a result, or change one in ~1e-7 of cases by one ulp:
//  * quotients and square roots go through div_nr / sqrt_nr (kj_screen.hpp: the hardware estimate + one Newton step on the result through an exact fma
//    residual: correctly rounded except ~1e-7 of operand pairs) wherever the operands are provably in their domain, with the IEEE sequence behind a
//    branch for black / non-finite texels; a third of this file's instructions were IEEE division sequences (round 4: TAA 0.29 -> 0.2x ms at 1080p);
//  * exp2 / log2 of arguments whose results stay in the normal range go straight to v_exp_f32 / v_log_f32 (same bits as libm there);
//  * `pow8(saturate(1e10 / luma))` -- the first of the two passes of the input / history filters -- is 1 for every luma in
//    [+0, 1e10], decided by one integer compare on the bits; the division runs only for the others.
// Inside input_prob and in the final pass (k_taa: blends, clamps, the fp16 stores of the frame's outputs), quotients use v_rcp_f32 +
// multiply and square roots v_sqrt_f32 (1 ulp each); what stays IEEE there: the moments E[x], E[x^2] a variance is formed from.
// Row-range aware tile mapping (see rtdgi.hip): rows [row0, row1) of the kernel's own resolution.
#define TILE_XY(W_, H_) TILE_XY_M(W_, H_, KJ_TILES_PLAIN)
#define TILE_XY_M(W_, H_, MODE_)                                                                 \
    const int lane = threadIdx.x;                                                                \
    const uint2 kj_tb = kj::tile_order<MODE_>();                                                 \
    const int x = int(kj_tb.x) * 8 + (lane & 7), y = row0 + int(kj_tb.y) * 8 + (lane >> 3); \
    const bool in_image = x < (W_) && y < ((H_) < row1 ? (H_) : row1);

// taa_common.hlsl (TAA_NONLINEARITY_TYPE 1, TAA_COLOR_MAPPING_MODE 1)
#ifndef KJ_TAA_NR_MASK
#define KJ_TAA_NR_MASK (31 | 512)
#endif
// KJ_TAA_NR_MASK (default: all groups on) switches groups of quotients / roots back to the IEEE sequences: what the round-4 A/B runs and the bisection that found the one
// site that has to stay IEEE (catmull_rom_5tap_history's last line) were built with (scripts/archive/r04_taa_bisect.sh, profiles/r04_ab_runs.md)
#define NRDIV(bit_, a_, b_) (((KJ_TAA_NR_MASK) & (bit_)) ? div_nr(a_, b_) : ((a_) / (b_)))
KJ_D V3 taa_decode_rgb(V3 v) {       // v * sqrt(max(0, m)) / max(1e-20, m), m = the largest component (upstream of input_prob: the reference's operations,
    const float m = max3(v.x, v.y, v.z);     // nearly always its bits -- kj_screen.hpp: div_nr / sqrt_nr), without a branch:
    if (!((KJ_TAA_NR_MASK) & 1)) return v * sqrtf(fmaxf(0.0f, m)) / fmaxf(1e-20f, m);
    const V3 q = div_nr(v * sqrt_nr_pos(fmaxf(m, FLT_MIN)), fmaxf(1e-20f, m));
    // m <= 0 (a black texel): sqrt(max(0, m)) = 0 and the quotient is v * 0 / 1e-20 = a signed zero (NaN for a NaN / infinite component, as v * 0 is);
    // m = inf or NaN: NaN from both forms. (0 < m < 2^-126 cannot come out of an fp16 image times an exposure ratio; it gives 0 here.)
    return V3{m >= FLT_MIN ? q.x : v.x * 0.0f, m >= FLT_MIN ? q.y : v.y * 0.0f, m >= FLT_MIN ? q.z : v.z * 0.0f};
}
KJ_D V3 taa_encode_rgb(V3 v) { const float m = max3(v.x, v.y, v.z); return v * ((m * m) * rcp_fast(fmaxf(1e-20f, m))); }
// pow8(saturate(cutoff / luma)) of the input / history filters. cutoff = 1e10 ("no cutoff"): 1 for every luma in [+0, 1e10]
KJ_D float pow8_(float x) { const float x2 = x * x, x4 = x2 * x2; return x4 * x4; }
KJ_D float luma_weight_uncut(float luma) { return __float_as_uint(luma) <= __float_as_uint(1e10f) ? 1.0f : pow8_(saturate(1e10f / luma)); }
// saturate(cutoff / luma) is 1 wherever 0 < luma <= cutoff (a correctly rounded quotient >= 1); above the cutoff luma > 0 and the quotient is in div_nr's domain
#ifndef KJ_TAA_LUMA_SELECT
#define KJ_TAA_LUMA_SELECT 0      // measured on MI355X (round 5): TAA 0.270-0.271 ms against 0.267-0.269 at 1080p, 1.072 against 1.059 at 4K with the select form: the branch form skips what no lane needs; off
#endif
KJ_D float luma_weight(float cutoff, float luma) {
#if KJ_TAA_LUMA_SELECT
    // Round 5: black texels without the IEEE sequence. The quotient is computed in div_nr's domain (both operands positive and finite); for a black texel under such a
    // cutoff `saturate(cutoff / +-0)` is 1 or 0 by the zero's sign, and under a zero cutoff (a black neighbourhood: a sky region is whole waves of them) every quotient
    // is a zero or NaN: weight 0 -- both read off the operands. What is left for the IEEE sequence (negative, infinite or NaN operands) no image of radiance produces;
    // it sits behind one branch per tap that is never taken. (Until round 4 every tap of a black texel or neighbourhood took the 12-instruction division.)
    const bool cdom = cutoff > 0.0f && cutoff < INFINITY;                         // the same for the nine taps of a pixel
    const bool dom = cdom && luma > 0.0f && luma < INFINITY, black = cdom && luma == 0.0f;
    const float q = div_nr(dom ? cutoff : 1.0f, dom ? luma : 1.0f);
    float w = luma <= cutoff ? 1.0f : pow8_(fminf(q, 1.0f));                       // (+0 <= cutoff: 1)
    w = (black && (__float_as_uint(luma) >> 31) != 0u) ? 0.0f : w;                  // cutoff / -0 = -inf
    w = (dom || black) ? w : 0.0f;                                                 // cutoff = +-0: 0
    if (!(dom || black || cutoff == 0.0f)) w = pow8_(saturate(cutoff / luma));
    return w;
#else
    if (((KJ_TAA_NR_MASK) & 2) && luma > 0.0f && luma < INFINITY && cutoff > 0.0f && cutoff < INFINITY) return luma <= cutoff ? 1.0f : pow8_(fminf(div_nr(cutoff, luma), 1.0f));     // finite operands: div_nr's domain
    return pow8_(saturate(cutoff / luma));       // black texels / a black neighbourhood (0 / 0 = NaN -> 0), negative lumas: as written
#endif
}
KJ_D float ld1h(const ImgH1& i, int x, int y) { return f16_to_f32(i.ld(x, y)); }

// image_sample_catmull_rom_5tap (inc/image.hlsl:88-172); the history remap (decode_rgb * pre_exposure_delta) is applied per tap
KJ_D V4 taa_history_tap(const ImgH4& tex, V2 uv, float ped) {
    const V4 v = sample_bilinear_clamp_rgba16f(tex.p, tex.w, tex.h, uv);
    return v4(taa_decode_rgb(xyz(v) * ped), v.w);
}
KJ_D V4 catmull_rom_5tap_history(const ImgH4& tex, V2 uv, V2 tex_size, float ped) {
    const V2 sample_pos = uv * tex_size;
    const V2 tex_pos1{floorf(sample_pos.x - 0.5f) + 0.5f, floorf(sample_pos.y - 0.5f) + 0.5f};
    const V2 f = sample_pos - tex_pos1;
    const V2 w0 = f * (-0.5f + f * (1.0f - 0.5f * f));
    const V2 w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
    const V2 w2 = f * (0.5f + f * (2.0f - 1.5f * f));
    const V2 w3 = f * f * (-0.5f + 0.5f * f);
    const V2 w12 = w1 + w2;
    const V2 offset12 = NRDIV(4 | 32, w2, w1 + w2);                                      // w1 + w2 in [0.5, 1.125]
    const V2 p0 = NRDIV(4 | 64, tex_pos1 - 1.0f, tex_size), p3 = NRDIV(4 | 64, tex_pos1 + 2.0f, tex_size), p12 = NRDIV(4 | 64, tex_pos1 + offset12, tex_size);
    V4 result = v4(0.0f);
    result += taa_history_tap(tex, V2{p12.x, p0.y}, ped) * w12.x * w0.y;
    result += taa_history_tap(tex, V2{p0.x, p12.y}, ped) * w0.x * w12.y;
    result += taa_history_tap(tex, V2{p12.x, p12.y}, ped) * w12.x * w12.y;
    result += taa_history_tap(tex, V2{p3.x, p12.y}, ped) * w3.x * w12.y;
    result += taa_history_tap(tex, V2{p12.x, p3.y}, ped) * w12.x * w3.y;
    // (this quotient stays on the IEEE sequence: with div_nr here -- bit-exact by itself, scripts/selftest_div_sqrt_nr.py -- the compiler schedules the five
    // accumulations above differently and channels that are exactly zero come out as -0 instead of +0, which TAA's luma weights turn into 0 against 1: measured)
    return NRDIV(128, result, w12.x * w0.y + w0.x * w12.y + w12.x * w12.y + w3.x * w12.y + w12.x * w3.y);      // the five weights sum to 0.9 .. 1
}

// reproject_history.hlsl:42-129
__global__ void __launch_bounds__(64) k_taa_reproject(const FrameConstants* __restrict__ fc, ImgH4 history_tex, ImgH4 reprojection_tex, ImgF32 depth_tex, ImgH4 output_tex,
                                                       ImgH2 closest_velocity_output, int IW, int IH, int row0, int row1) {
    const int OW = output_tex.w, OH = output_tex.h;
    TILE_XY_M(OW, OH, KJ_TILES_ROWS)
    const V4 its = tex_size4(IW, IH), ots = tex_size4(OW, OH);
    const V2 scale{NRDIV(4 | 256, its.x, ots.x), NRDIV(4 | 256, its.y, ots.y)};
    const int rx = int(uint32_t((float(x) + 0.5f) * scale.x)), ry = int(uint32_t((float(y) + 0.5f) * scale.y));
    // (round 6: the four corner texels and the pixel's own reprojection -- the one the history is fetched with unless the velocity gets dilated -- are requested
    // together; the text's order is two round trips for the corners, the dilation's, then one more for the velocity before the history's twelve texels can be asked for)
    V2 vmn, vmx;
    uint2 rc_raw;
    {
        bool in_[5]; uint2 q[5];
        q[0] = reprojection_tex.ld_raw(rx - 1, ry - 1, in_[0]); q[1] = reprojection_tex.ld_raw(rx + 1, ry - 1, in_[1]);
        q[2] = reprojection_tex.ld_raw(rx - 1, ry + 1, in_[2]); q[3] = reprojection_tex.ld_raw(rx + 1, ry + 1, in_[3]);
        q[4] = reprojection_tex.ld_raw(rx, ry, in_[4]);
#pragma unroll
        for (int i = 0; i < 5; ++i) if (!in_[i]) q[i] = make_uint2(0u, 0u);
        auto xy = [](uint2 p_) { return V2{from_snorm16(int16_t(p_.x & 0xffff)), from_snorm16(int16_t(p_.x >> 16))}; };
        V2 r = xy(q[0]); vmn = vmx = r;
        r = xy(q[1]); vmn = vmin(vmn, r); vmx = V2{fmaxf(vmx.x, r.x), fmaxf(vmx.y, r.y)};
        r = xy(q[2]); vmn = vmin(vmn, r); vmx = V2{fmaxf(vmx.x, r.x), fmaxf(vmx.y, r.y)};
        r = xy(q[3]); vmn = vmin(vmn, r); vmx = V2{fmaxf(vmx.x, r.x), fmaxf(vmx.y, r.y)};
        rc_raw = q[4];
    }
    const V2 d = vmx - vmn, s = vmx + vmn;
    int should_dilate = (d.x > 0.1f * fmaxf(its.z, fabsf(s.x)) || d.y > 0.1f * fmaxf(its.w, fabsf(s.y))) ? 1 : 0;
    should_dilate |= __shfl_xor(should_dilate, 2);
    should_dilate |= __shfl_xor(should_dilate, 16);
    int cx = rx, cy = ry;
    if (should_dilate) {
        float reproj_depth = depth_tex.ld(rx, ry);
        for (int oy = -1; oy <= 1; ++oy)
            for (int ox = -1; ox <= 1; ++ox) {
                const float dd = depth_tex.ld(rx + ox, ry + oy);
                if (dd > reproj_depth) { reproj_depth = dd; cx = rx + ox; cy = ry + oy; }
            }
    }
    if (cx != rx || cy != ry) rc_raw = reprojection_tex.ld(cx, cy);
    const V2 reproj_xy{from_snorm16(int16_t(rc_raw.x & 0xffff)), from_snorm16(int16_t(rc_raw.x >> 16))};
    if (!in_image) return;  // after the wave vote
    st2h(closest_velocity_output, x, y, reproj_xy);
    const V2 uv = get_uv(float(x), float(y), ots);
    const V4 hp = catmull_rom_5tap_history(history_tex, uv + reproj_xy, V2{ots.x, ots.y}, fc->pre_exposure_delta);
    st4(output_tex, x, y, v4(xyz(hp), fmaxf(0.0f, hp.w)));
}

// filter_input.hlsl:33-88. The shader calls filter_input_inner twice over the same 3x3 taps (first with an infinite
// luma cutoff, then with 1.001x the first pass' luma); here the taps are decoded once and kept in registers, and
// pow(x, 8) is three squarings.
KJ_D void taa_filter_input_body(const ImgH4& input_tex, const ImgF32& depth_tex, const ImgH4& output_tex, const ImgH4& dev_output_tex, int row0, int row1) {
    TILE_XY_M(output_tex.w, output_tex.h, KJ_TILES_ROWS)
    // LDS-staged 10x10 tile: .xyz = decoded YCbCr of the input texel, .w = depth (one decode per texel instead of nine)
    __shared__ float4 tile[10 * 10];
    {
        const int tx0 = int(kj_tb.x) * 8 - 1, ty0 = row0 + int(kj_tb.y) * 8 - 1;
        for (int i = lane; i < 100; i += 64) {
            const int tx = tx0 + i % 10, ty = ty0 + i / 10;
            const V3 c = sRGB_to_YCbCr(taa_decode_rgb(xyz(ld4(input_tex, tx, ty))));
            tile[i] = make_float4(c.x, c.y, c.z, depth_tex.ld(tx, ty));
        }
    }
    __syncthreads();
    if (!in_image) return;
    const int lt = ((lane >> 3) + 1) * 10 + (lane & 7) + 1;
    const float center_depth = tile[lt].w;
    const float depth_scale = 200.0f;
    V3 s[9]; float wd[9];
#pragma unroll
    for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
        for (int xx = -1; xx <= 1; ++xx) {
            const int i = (yy + 1) * 3 + (xx + 1);
            const float distance_w = expf(-0.8f * float(xx * xx + yy * yy));
            const float4 t = tile[lt + yy * 10 + xx];
            s[i] = V3{t.x, t.y, t.z};
            float w = 1;
            // inverse_depth_relative_diff(center, tap): both operands are >= 1e-20 and finite, the quotient is a normal number: div_nr's domain (round 5; 9 IEEE sequences per pixel before)
            const float depth_diff = ((KJ_TAA_NR_MASK) & 512) ? fabsf(div_nr(fmaxf(1e-20f, center_depth), fmaxf(1e-20f, t.w)) - 1.0f) : inverse_depth_relative_diff(center_depth, t.w);
            w *= exp2_fast(-fminf(16.0f, depth_scale * depth_diff));     // >= 2^-16: same bits as exp2f
            w *= distance_w;
            wd[i] = w;
        }
    // pass 1: luma_cutoff = 1e10
    V3 iex = v3(0.0f), iex2 = v3(0.0f), clamped_iex = v3(0.0f);
    float clamped_iwsum = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float w = wd[i] * luma_weight_uncut(s[i].x);
        clamped_iwsum += w;
        clamped_iex += s[i] * w;
        iex += s[i];
        iex2 += s[i] * s[i];
    }
    clamped_iex = NRDIV(8, clamped_iex, clamped_iwsum);      // (a zero weight sum -- a neighbourhood of NaNs -- is NaN either way)
    iex = NRDIV(8, iex, 9.0f);
    iex2 = NRDIV(8, iex2, 9.0f);
    // E[x^2] - E[x]^2 cancels: its products and sums round one by one, as in the text (this file is compiled without FMA contraction: a fused multiply-add here moves
    // the deviation image by an fp16 step in 5 % of the texels)
    const V3 var_a = vmax(v3(0.0f), iex2 - iex * iex);
    // pass 2: luma_cutoff = first pass' luma * 1.001
    const float cutoff = clamped_iex.x * 1.001f;
    V3 cex = v3(0.0f); float cws = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float w = wd[i] * luma_weight(cutoff, s[i].x);
        cws += w;
        cex += s[i] * w;
    }
    cex = NRDIV(8, cex, cws);          // (cws = 0 in a black neighbourhood: 0 / 0 = NaN either way)
    st4(output_tex, x, y, v4(cex, 0.0f));
    st4(dev_output_tex, x, y, v4(((KJ_TAA_NR_MASK) & 8) ? V3{sqrt_nr(var_a.x), sqrt_nr(var_a.y), sqrt_nr(var_a.z)} : vsqrt(var_a), 0.0f));
}
__global__ void __launch_bounds__(64) k_taa_filter_input(ImgH4 input_tex, ImgF32 depth_tex, ImgH4 output_tex, ImgH4 dev_output_tex, int row0, int row1) {
    taa_filter_input_body(input_tex, depth_tex, output_tex, dev_output_tex, row0, row1);
}

// filter_history.hlsl:15-61. Same two-pass structure as filter_input; K = 1 unless the history is > 1.75x the input extent.
template <int K, bool UNCUT>
KJ_D V3 fh_filter_input(const V3* taps, float luma_cutoff) {
    V3 iex = v3(0.0f);
    float iwsum = 0;
#pragma unroll
    for (int yy = -K; yy <= K; ++yy)
#pragma unroll
        for (int xx = -K; xx <= K; ++xx) {
            const float distance_w = expf(-(0.8f / float(K * K)) * float(xx * xx + yy * yy));
            const V3 s = taps[(yy + K) * (2 * K + 1) + (xx + K)];
            const float w = distance_w * (UNCUT ? luma_weight_uncut(s.x) : luma_weight(luma_cutoff, s.x));
            iwsum += w;
            iex += s * w;
        }
    return NRDIV(8, iex, iwsum);
}
template <int K, bool TILED>
KJ_D void taa_filter_history_body(const ImgH4& reprojected_history, const ImgH4& output_tex, int row0, int row1) {
    TILE_XY_M(output_tex.w, output_tex.h, KJ_TILES_ROWS)
    constexpr int TW = 8 + 2 * K;
    __shared__ float4 tile[TILED ? TW * TW : 1];
    V3 taps[(2 * K + 1) * (2 * K + 1)];
    if (TILED) {   // same extent: the stencil of pixel (x, y) is centred on texel (x, y); stage the converted tile once
        const int tx0 = int(kj_tb.x) * 8 - K, ty0 = row0 + int(kj_tb.y) * 8 - K;
        for (int i = lane; i < TW * TW; i += 64) {
            const V3 c = sRGB_to_YCbCr(xyz(ld4(reprojected_history, tx0 + i % TW, ty0 + i / TW)));
            tile[i] = make_float4(c.x, c.y, c.z, 0.0f);
        }
        __syncthreads();
        if (!in_image) return;
        const int lt = ((lane >> 3) + K) * TW + (lane & 7) + K;
#pragma unroll
        for (int yy = -K; yy <= K; ++yy)
#pragma unroll
            for (int xx = -K; xx <= K; ++xx) { const float4 t = tile[lt + yy * TW + xx]; taps[(yy + K) * (2 * K + 1) + (xx + K)] = V3{t.x, t.y, t.z}; }
    } else {
        if (!in_image) return;
        const V2 uv = get_uv(float(x), float(y), tex_size4(output_tex.w, output_tex.h));
        const int sx = int(floorf(uv.x * float(reprojected_history.w) + 1e-3f)), sy = int(floorf(uv.y * float(reprojected_history.h) + 1e-3f));
#pragma unroll
        for (int yy = -K; yy <= K; ++yy)
#pragma unroll
            for (int xx = -K; xx <= K; ++xx) taps[(yy + K) * (2 * K + 1) + (xx + K)] = sRGB_to_YCbCr(xyz(ld4(reprojected_history, sx + xx, sy + yy)));
    }
    const float filtered_luma = fh_filter_input<K, true>(taps, 1e10f).x;
    st4(output_tex, x, y, v4(fh_filter_input<K, false>(taps, filtered_luma * 1.001f), 0.0f));
}
template <int K, bool TILED>
__global__ void __launch_bounds__(64) k_taa_filter_history(ImgH4 reprojected_history, ImgH4 output_tex, int row0, int row1) {
    taa_filter_history_body<K, TILED>(reprojected_history, output_tex, row0, row1);
}
// "taa filter input" + "taa filter history" in ONE launch (same extent: both stencils are centred on the wave's tile): two independent
// passes of the reference's seven on the same tiles -- five launches per TAA frame instead of six; each half is its own function, so
// the early exit of a lane outside the image leaves only that half.
__global__ void __launch_bounds__(64) k_taa_filter_input_and_history(ImgH4 input_tex, ImgF32 depth_tex, ImgH4 filtered_input, ImgH4 filtered_input_dev, ImgH4 reprojected_history,
                                                                      ImgH4 filtered_history, int row0, int row1) {
    taa_filter_input_body(input_tex, depth_tex, filtered_input, filtered_input_dev, row0, row1);
    taa_filter_history_body<1, true>(reprojected_history, filtered_history, row0, row1);
}

#ifndef KJ_TAA_PROB_TILE
#define KJ_TAA_PROB_TILE 1
#endif
// input_prob.hlsl:50-108
__global__ void __launch_bounds__(64) k_taa_input_prob(const FrameConstants* __restrict__ fc, ImgH4 filtered_input_tex, ImgH4 filtered_input_dev_tex, ImgH4 filtered_history_tex,
                                                        ImgH4 reprojection_tex, ImgH4 smooth_var_history_tex, ImgH2 velocity_history_tex, ImgH1 output_tex, int row0, int row1) {
    const int IW = output_tex.w, IH = output_tex.h;
    TILE_XY_M(IW, IH, KJ_TILES_ROWS)
#if KJ_TAA_PROB_TILE
    // Round 5: the 3x3 taps of the filtered input and of the reprojection map come from a 10x10 LDS tile decoded once per texel (same decode, same values)
    // instead of nine times per pixel
    __shared__ float4 in_tile[10 * 10];
    __shared__ float2 rv_tile[10 * 10];
    __shared__ float4 dev_tile[12 * 12];      // the deviation image's stride-2 3x3 taps reach two texels either side
    {
        const int dx0 = int(kj_tb.x) * 8 - 2, dy0 = row0 + int(kj_tb.y) * 8 - 2;
        for (int i = lane; i < 144; i += 64) {
            const V3 d = xyz(ld4(filtered_input_dev_tex, dx0 + i % 12, dy0 + i / 12));
            dev_tile[i] = make_float4(d.x, d.y, d.z, 0.0f);
        }
        const int tx0 = int(kj_tb.x) * 8 - 1, ty0 = row0 + int(kj_tb.y) * 8 - 1;
        for (int i = lane; i < 100; i += 64) {
            const int tx = tx0 + i % 10, ty = ty0 + i / 10;
            const V3 c = xyz(ld4(filtered_input_tex, tx, ty));
            const V4 r = ld_reproj(reprojection_tex, tx, ty);
            in_tile[i] = make_float4(c.x, c.y, c.z, 0.0f);
            rv_tile[i] = make_float2(r.x, r.y);
        }
    }
    __syncthreads();
    const int lt = ((lane >> 3) + 1) * 10 + (lane & 7) + 1;
#endif
    if (!in_image) return;
    const V4 its = tex_size4(IW, IH);
    V3 ivar = v3(0.0f);
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
#if KJ_TAA_PROB_TILE
            const float4 td = dev_tile[((lane >> 3) + 2 + oy * 2) * 12 + (lane & 7) + 2 + ox * 2];
            ivar = vmax(ivar, V3{td.x, td.y, td.z});
#else
            ivar = vmax(ivar, xyz(ld4(filtered_input_dev_tex, x + ox * 2, y + oy * 2)));
#endif
        }
    ivar = ivar * ivar;
    const V2 input_uv{(float(x) + fc->view_constants.sample_offset_pixels[0]) * its.z, (float(y) + fc->view_constants.sample_offset_pixels[1]) * its.w};
    const V4 closest_history = unpack_rgba16f(sample_nearest_clamp(filtered_history_tex, input_uv));
    const V4 rp = ld_reproj(reprojection_tex, x, y);
    const V2 huv = input_uv + V2{rp.x, rp.y};
    const V3 closest_smooth_var = xyz(sample_bilinear_clamp_rgba16f(smooth_var_history_tex.p, smooth_var_history_tex.w, smooth_var_history_tex.h, huv));
    const V2 closest_vel = sample_bilinear_clamp_rg16f(velocity_history_tex.p, velocity_history_tex.w, velocity_history_tex.h, huv) * fc->delta_time_seconds;
    const V3 combined_var = vmin(closest_smooth_var, ivar * 10.0f);
    const V3 inv_var{rcp_fast(fmaxf(1e-6f, combined_var.x)), rcp_fast(fmaxf(1e-6f, combined_var.y)), rcp_fast(fmaxf(1e-6f, combined_var.z))};
    float input_prob = 0;
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
#if KJ_TAA_PROB_TILE
            const float4 ti = in_tile[lt + oy * 10 + ox]; const float2 tr = rv_tile[lt + oy * 10 + ox];
            const V3 idiff = V3{ti.x, ti.y, ti.z} - xyz(closest_history);
            const V2 rv{tr.x, tr.y};
#else
            const V3 idiff = xyz(ld4(filtered_input_tex, x + ox, y + oy)) - xyz(closest_history);
            const V4 rv = ld_reproj(reprojection_tex, x + ox, y + oy);
#endif
            const V2 q{(rv.x - closest_vel.x) * rcp_fast(fmaxf(1.0f, fabsf(rv.x + closest_vel.x))), (rv.y - closest_vel.y) * rcp_fast(fmaxf(1.0f, fabsf(rv.y + closest_vel.y)))};
            // (results below 2^-126 come out as 0: the image is fp16)
            const float prob = exp2_fast(-1.0f * length_fast(idiff * idiff * inv_var) - 1000.0f * length_fast(q));
            input_prob = fmaxf(input_prob, prob);
        }
    output_tex.st(x, y, f32_to_f16(input_prob));
}
// ---- from here on (the probability dilations, the final pass): downstream of the hypersensitive stage, tolerant of a last bit -- FMA contraction back on
// (this file is compiled with -ffp-contract=off: csrc/Makefile)
#pragma clang fp contract(fast)
// filter_prob.hlsl, filter_prob2.hlsl
__global__ void __launch_bounds__(64) k_taa_filter_prob(ImgH1 input_tex, ImgH1 output_tex, int row0, int row1) {
    TILE_XY(output_tex.w, output_tex.h)
    if (!in_image) return;
    float prob = ld1h(input_tex, x, y);
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) prob = fmaxf(prob, ld1h(input_tex, x + ox, y + oy));
    output_tex.st(x, y, f32_to_f16(prob));
}
__global__ void __launch_bounds__(64) k_taa_filter_prob2(ImgH1 input_tex, ImgH1 output_tex, int row0, int row1) {
    TILE_XY(output_tex.w, output_tex.h)
    if (!in_image) return;
    V2 weighted{0, 0};
#pragma unroll
    for (int oy = -2; oy <= 2; ++oy)
#pragma unroll
        for (int ox = -2; ox <= 2; ++ox) weighted += V2{exp2_fast(-clampf(10.0f * ld1h(input_tex, x + ox * 2, y + oy * 2), 0.0f, 100.0f)), 1.0f};     // >= 2^-100: normal range
    output_tex.st(x, y, f32_to_f16(fmaxf(0.0f, -1.0f / 10.0f * log2_fast(1e-30f + weighted.x / weighted.y))));
}

// filter_prob.hlsl + filter_prob2.hlsl in one launch (same extent, whole-image calls): a 16x16 pixel workgroup stages the 26x26
// probabilities its outputs can reach (4 px of filter_prob2's stride-2 5x5 + 1 px of filter_prob's 3x3), takes the 3x3 maxima of the
// inner 24x24 in LDS and runs the second filter from there. Both images are still written (prob_filtered1_img is a named surface);
// same values bit for bit: the maxima of fp16 values are exact and the second filter sums in the same order.
__global__ void __launch_bounds__(256) k_taa_filter_prob_both(ImgH1 input_tex, ImgH1 prob1_tex, ImgH1 output_tex) {
    __shared__ float s_in[26 * 26];
    __shared__ float s_p1[24 * 24];
    __shared__ float s_e1[24 * 24];      // exp2(-clamp(10 p, 0, 100)) of every staged probability, once per texel instead of once per tap (25 taps read each texel; round 5)
    const int W = output_tex.w, H = output_tex.h;
    const int tid = int(threadIdx.x);
    const uint2 tb = tile_order<KJ_TILES_ROWS>();
    const int bx0 = int(tb.x) * 16, by0 = int(tb.y) * 16;
    for (int i = tid; i < 26 * 26; i += 256) { const int ty = i / 26, tx = i - ty * 26; s_in[i] = ld1h(input_tex, bx0 - 5 + tx, by0 - 5 + ty); }
    __syncthreads();
    for (int i = tid; i < 24 * 24; i += 256) {
        const int ty = i / 24, tx = i - ty * 24;
        const int px = bx0 - 4 + tx, py = by0 - 4 + ty;
        float prob = 0.0f;     // outside the image the second filter reads 0 (an out-of-bounds load of prob_filtered1_img)
        if (uint32_t(px) < uint32_t(W) && uint32_t(py) < uint32_t(H)) {
            prob = s_in[(ty + 1) * 26 + tx + 1];
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                for (int ox = -1; ox <= 1; ++ox) prob = fmaxf(prob, s_in[(ty + 1 + oy) * 26 + tx + 1 + ox]);
        }
        s_p1[i] = prob;
        s_e1[i] = exp2_fast(-clampf(10.0f * prob, 0.0f, 100.0f));
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int x = bx0 + lx, y = by0 + ly;
    if (!(x < W && y < H)) return;
    prob1_tex.st(x, y, f32_to_f16(s_p1[(ly + 4) * 24 + lx + 4]));
    V2 weighted{0, 0};
#pragma unroll
    for (int oy = -2; oy <= 2; ++oy)
#pragma unroll
        for (int ox = -2; ox <= 2; ++ox) weighted += V2{s_e1[(ly + 4 + oy * 2) * 24 + lx + 4 + ox * 2], 1.0f};
    output_tex.st(x, y, f32_to_f16(fmaxf(0.0f, -1.0f / 10.0f * log2_fast(1e-30f + weighted.x / weighted.y))));
}

// inc/unjitter_taa.hlsl:58-125 (kernel half width 1). taa.hlsl calls it twice on the same taps (kernel_scale 1 and 0.333);
// the taps are decoded once.
struct Unjittered { V4 color; float coverage; V3 ex, ex2; };
template <bool TILED>
KJ_D void sample_image_unjitter_taa2(const ImgH4& img, const float4* tile_cols /* 10x10, centred on this lane */, int opx, int opy, V2 out_size, V2 sample_offset_pixels, float ks_a, float ks_b, Unjittered& ra, Unjittered& rb) {
    const V2 scale = NRDIV(16, (V2{float(img.w), float(img.h)}), out_size);
    const int bx = int((float(opx) + 0.5f) * scale.x), by = int((float(opy) + 0.5f) * scale.y);
    const V2 dst_sample_loc{float(opx) + 0.5f, float(opy) + 0.5f};
    const V2 base_src_sample_loc = NRDIV(16, (V2{float(bx) + 0.5f + sample_offset_pixels.x, float(by) + 0.5f - sample_offset_pixels.y}), scale);
    V4 res_a = v4(0.0f), res_b = v4(0.0f);
    V3 ex_a = v3(0.0f), ex2_a = v3(0.0f), ex_b = v3(0.0f), ex2_b = v3(0.0f);
    float dev_wt_sum_a = 0, wt_sum_a = 0, dev_wt_sum_b = 0, wt_sum_b = 0;
#pragma unroll
    for (int yy = -1; yy <= 1; ++yy)
#pragma unroll
        for (int xx = -1; xx <= 1; ++xx) {
            const V2 src_sample_loc = base_src_sample_loc + NRDIV(16, (V2{float(xx), float(yy)}), scale);
            V3 col;
            if (TILED) { const float4 t = tile_cols[yy * 10 + xx]; col = V3{t.x, t.y, t.z}; }
            else col = sRGB_to_YCbCr(taa_decode_rgb(xyz(ld4(img, bx + xx, by + yy))));
            const V2 off = src_sample_loc - dst_sample_loc;
            {
                const V2 o = off * ks_a;
                const float dist2 = dot(o, o);
                const float dev_wt = exp2_fast(-dist2 * scale.x), wt = exp2_fast(-10.0f * dist2 * scale.x);     // |arg| < 100: normal range
                res_a += v4(col, 1.0f) * wt; wt_sum_a += wt;
                ex_a += col * dev_wt; ex2_a += col * col * dev_wt; dev_wt_sum_a += dev_wt;
            }
            {
                const V2 o = off * ks_b;
                const float dist2 = dot(o, o);
                const float dev_wt = exp2_fast(-dist2 * scale.x), wt = exp2_fast(-10.0f * dist2 * scale.x);
                res_b += v4(col, 1.0f) * wt; wt_sum_b += wt;
                ex_b += col * dev_wt; ex2_b += col * col * dev_wt; dev_wt_sum_b += dev_wt;
            }
        }
    ra = Unjittered{res_a, wt_sum_a, NRDIV(16, ex_a, dev_wt_sum_a), NRDIV(16, ex2_a, dev_wt_sum_a)};     // a variance is formed from these moments: the (nearly always) correctly rounded quotients
    rb = Unjittered{res_b, wt_sum_b, NRDIV(16, ex_b, dev_wt_sum_b), NRDIV(16, ex2_b, dev_wt_sum_b)};
}

// taa.hlsl:94-338
struct TaaArgs {
    const FrameConstants* __restrict__ fc;
    ImgH4 input_tex, history_tex, reprojection_tex; ImgH2 closest_velocity_tex, velocity_history_tex; ImgH4 smooth_var_history_tex; ImgH1 input_prob_tex;
    ImgH4 temporal_output_tex, output_tex, smooth_var_output_tex; ImgH2 velocity_output_tex;
    int row0, row1;
};
template <bool TILED>   // TILED: input extent == output extent, so both stencils are centred on texel (x, y) and can be staged through LDS
__global__ void __launch_bounds__(64) k_taa(TaaArgs a) {
    const int row0 = a.row0, row1 = a.row1;
    const int OW = a.temporal_output_tex.w, OH = a.temporal_output_tex.h;
    TILE_XY_M(OW, OH, KJ_TILES_ROWS)
    __shared__ float4 hist_tile[TILED ? 12 * 12 : 1];   // raw reprojected history (5x5 blur)
    __shared__ float4 col_tile[TILED ? 10 * 10 : 1];    // decoded YCbCr of the input (3x3 unjitter taps)
    if (TILED) {
        const int tx0 = int(kj_tb.x) * 8, ty0 = row0 + int(kj_tb.y) * 8;
        for (int i = lane; i < 144; i += 64) {
            const V4 h = ld4(a.history_tex, tx0 - 2 + i % 12, ty0 - 2 + i / 12);
            hist_tile[i] = make_float4(h.x, h.y, h.z, h.w);
        }
        for (int i = lane; i < 100; i += 64) {
            const V3 c = sRGB_to_YCbCr(taa_decode_rgb(xyz(ld4(a.input_tex, tx0 - 1 + i % 10, ty0 - 1 + i / 10))));
            col_tile[i] = make_float4(c.x, c.y, c.z, 0.0f);
        }
        __syncthreads();
    }
    if (!in_image) return;
    const FrameConstants& fc = *a.fc;
    const V4 ots = tex_size4(OW, OH);
    const V2 frac_{NRDIV(16, float(a.input_tex.w), float(OW)), NRDIV(16, float(a.input_tex.h), float(OH))};
    const V2 sop{fc.view_constants.sample_offset_pixels[0], fc.view_constants.sample_offset_pixels[1]};
    const int rx = int(uint32_t((float(x) + 0.5f) * frac_.x)), ry = int(uint32_t((float(y) + 0.5f) * frac_.y));
    const V2 uv = get_uv(float(x), float(y), ots);
    const int lth = ((lane >> 3) + 2) * 12 + (lane & 7) + 2, ltc = ((lane >> 3) + 1) * 10 + (lane & 7) + 1;
    V4 history_packed;
    if (TILED) { const float4 t = hist_tile[lth]; history_packed = V4{t.x, t.y, t.z, t.w}; }
    else history_packed = ld4(a.history_tex, x, y);
    V3 history = xyz(history_packed);
    float history_coverage = fmaxf(0.0f, history_packed.w);
    V4 csum = v4(0.0f); float wsum = 0;
#pragma unroll
    for (int oy = -2; oy <= 2; ++oy)
#pragma unroll
        for (int ox = -2; ox <= 2; ++ox) {
            const float w = expf(-float(ox * ox + oy * oy));
            V4 hv;
            if (TILED) { const float4 t = hist_tile[lth + oy * 12 + ox]; hv = V4{t.x, t.y, t.z, t.w}; }
            else hv = ld4(a.history_tex, x + ox, y + oy);
            csum += hv * w;
            wsum += w;
        }
    const V4 bhistory_packed = csum * (1.0f / wsum);     // (wsum is a compile-time constant)
    V3 bhistory = xyz(bhistory_packed);
    const float bhistory_coverage = bhistory_packed.w;
    history = sRGB_to_YCbCr(history);
    bhistory = sRGB_to_YCbCr(bhistory);
    const V4 reproj = ld_reproj(a.reprojection_tex, rx, ry);
    const V2 reproj_xy = ld2h(a.closest_velocity_tex, x, y);
    Unjittered center_sample, bcenter_sample;
    sample_image_unjitter_taa2<TILED>(a.input_tex, col_tile + (TILED ? ltc : 0), x, y, V2{ots.x, ots.y}, sop, 1.0f, 0.333f, center_sample, bcenter_sample);
    float coverage = center_sample.coverage;
    V3 center = xyz(center_sample.color);
    const V3 bcenter = xyz(bcenter_sample.color) * rcp_fast(bcenter_sample.coverage);
    history = lerp(history, bcenter, saturate(1.0f - history_coverage));
    bhistory = lerp(bhistory, bcenter, saturate(1.0f - bhistory_coverage));
    const float input_prob = ld1h(a.input_prob_tex, rx, ry);
    const V3 ex = center_sample.ex, ex2 = center_sample.ex2;
    const V3 var = vmax(v3(0.0f), ex2 - ex * ex);
    const V3 prev_var = v3(sample_bilinear_clamp_rgba16f(a.smooth_var_history_tex.p, OW, OH, uv + reproj_xy).x);
    const V2 vel_now = reproj_xy / fc.delta_time_seconds;
    const V2 vel_prev = sample_bilinear_clamp_rg16f(a.velocity_history_tex.p, OW, OH, uv + reproj_xy);
    const V2 vq{(vel_now.x - vel_prev.x) * rcp_fast(fmaxf(1.0f, fabsf(vel_now.x + vel_prev.x))), (vel_now.y - vel_prev.y) * rcp_fast(fmaxf(1.0f, fabsf(vel_now.y + vel_prev.y)))};
    const float var_blend = saturate(0.3f + 0.7f * (1 - reproj.z) + length_fast(vq));
    V3 smooth_var = vmax(var, lerp(prev_var, var, var_blend));
    smooth_var = lerp(var, smooth_var, saturate(input_prob));
    const V3 input_dev{sqrt_fast(var.x), sqrt_fast(var.y), sqrt_fast(var.z)};
    V3 clamped_history;
    {
        const float box_n_deviations = lerp(0.8f, 3.0f, input_prob);
        const V3 nmin = ex - input_dev * box_n_deviations, nmax = ex + input_dev * box_n_deviations;
        const V3 clamped_bhistory = vclamp(bhistory, nmin, nmax);
        auto rcp3 = [](V3 a) { return V3{rcp_fast(a.x), rcp_fast(a.y), rcp_fast(a.z)}; };
        const float clamping_event = length_fast(vmax(v3(0.0f), vmax(bhistory - nmax, nmin - bhistory)) * rcp3(vmax(v3(0.01f), ex)));
        const V3 outlier3 = vmax(v3(0.0f), vmax(nmin - history, history - nmax) * rcp3(0.1f + vmax(vmax(vabs(history), vabs(ex)), v3(1e-5f))));
        const V3 boutlier3 = vmax(v3(0.0f), vmax(nmin - bhistory, bhistory - nmax) * rcp3(0.1f + vmax(vmax(vabs(bhistory), vabs(ex)), v3(1e-5f))));
        const float outlier = fmaxf(outlier3.x, fmaxf(outlier3.y, outlier3.z));
        const float boutlier = fmaxf(boutlier3.x, fmaxf(boutlier3.y, boutlier3.z));
        const V2 huv = uv + reproj_xy;
        const bool history_valid = huv.x == saturate(huv.x) && huv.y == saturate(huv.y);
        if (history_valid) {
            const float non_disoccluding_outliers = fmaxf(0.0f, outlier - boutlier) * 10;
            const V3 unclamped_history_detail = history - clamped_bhistory;
            const float temporal_clamping_detail = fabsf(unclamped_history_detail.x * rcp_fast(fmaxf(1e-3f, input_dev.x))) * 0.05f;
            const float temporal_stability = saturate(1 - temporal_clamping_detail);
            const float allow_unclamped_detail = saturate(non_disoccluding_outliers) * temporal_stability;
            V3 history_detail = history - bhistory;
            history_detail = lerp(history_detail, unclamped_history_detail, allow_unclamped_detail);
            const float initial_bclamp_amount = saturate(dot(clamped_bhistory - bhistory, bcenter - bhistory) *
                                                         rcp_fast(fmaxf(1e-5f, length_fast(clamped_bhistory - bhistory) * length_fast(bcenter - bhistory))));
            const float keep_detail = 1 - saturate(initial_bclamp_amount) * (1 - allow_unclamped_detail);
            history_detail *= keep_detail;
            clamped_history = clamped_bhistory + history_detail;
            if (frac_.x < 1.0f) history_coverage *= lerp(lerp(0.0f, 0.9f, keep_detail), 1.0f, saturate(10 * clamping_event));
        } else {
            clamped_history = clamped_bhistory;
            coverage = 1;
            center = bcenter;
            history_coverage = 0;
        }
        clamped_history = lerp(clamped_history, history, smoothstep(0.5f, 1.0f, input_prob));
    }
    float total_coverage = fmaxf(1e-5f, history_coverage + coverage);
    V3 temporal_result = (clamped_history * history_coverage + center) * rcp_fast(total_coverage);
    total_coverage = fminf(fmaxf(2.0f, NRDIV(16, 8.0f, frac_.x * frac_.y)), total_coverage);
    st4(a.smooth_var_output_tex, x, y, v4(smooth_var, 0.0f));
    temporal_result = vmax(v3(0.0f), taa_encode_rgb(YCbCr_to_sRGB(temporal_result)));
    st4(a.temporal_output_tex, x, y, v4(temporal_result, total_coverage));
    st4(a.output_tex, x, y, v4(temporal_result, 0.0f));
    st2h(a.velocity_output_tex, x, y, vel_now);
}

// ================================================================== host
struct KjTaa {
    KjDevice* dev = nullptr;
    int IW = 0, IH = 0, OW = 0, OH = 0;
    std::map<std::string, kj::DevBuf> surf;
    bool flip[3] = {false, false, false};
    bool merge_prob_filters = true;             // filter_prob + filter_prob2 as one launch when both run over the whole image (KJ_TAA_MERGE_PROB=0: two launches)
    hipError_t err = hipSuccess;
    void* get(const std::string& name, size_t bytes, hipStream_t s) {
        kj::DevBuf& b = surf[name];
        if (b.bytes != bytes) { hipError_t e = b.alloc(bytes, s); if (e != hipSuccess) err = e; }
        return b.p;
    }
    void pingpong(const char* key, int idx, size_t bytes, hipStream_t s, void*& output, void*& history) {
        std::string a = std::string(key) + ":0", b = std::string(key) + ":1";
        if (flip[idx]) std::swap(a, b);
        output = get(a, bytes, s);
        history = get(b, bytes, s);
        flip[idx] = !flip[idx];
    }
};

#define KJ_CHECK_LAUNCH() KJ_TRY_HIP(hipGetLastError())

extern "C" {

KjStatus kj_taa_create(KjDevice* dev, KjTaa** out) {
    KJ_REQUIRE(dev && out, "null argument");
    KjTaa* t = new KjTaa();
    t->dev = dev;
    if (const char* v = kj_debug_getenv("KJ_TAA_MERGE_PROB")) t->merge_prob_filters = atoi(v) != 0;
    *out = t;
    return KJ_OK;
}
void kj_taa_destroy(KjTaa* t) { delete t; }

// TaaRenderer::render (taa.rs:41-191)
static KjStatus taa_render_impl(KjTaa* t, const void* input_tex, uint32_t input_width, uint32_t input_height, const void* reprojection_map, const void* depth_tex,
                                uint32_t output_width, uint32_t output_height, KjTaaOutput* out, void* stream_, uint32_t mask, uint32_t row_begin, uint32_t row_end) {
    KJ_REQUIRE(t && input_tex && reprojection_map && depth_tex && out && input_width && input_height && output_width && output_height, "null argument");
    KJ_REQUIRE(t->dev->fc_dev, "kj_frame_begin not called");
    hipStream_t s = (hipStream_t)stream_;
    const int IW = int(input_width), IH = int(input_height), OW = int(output_width), OH = int(output_height);
    if (IW != t->IW || IH != t->IH || OW != t->OW || OH != t->OH) { t->surf.clear(); t->IW = IW; t->IH = IH; t->OW = OW; t->OH = OH; }
    const FrameConstants* fc = t->dev->fc_dev;
    const size_t OB = size_t(OW) * OH, IB = size_t(IW) * IH;
    int or0 = 0, or1 = OH;
    if (row_end > row_begin) {
        KJ_REQUIRE(IW == OW && IH == OH, "row ranges need input extent == output extent");
        KJ_REQUIRE(row_begin % 8 == 0 && (row_end % 8 == 0 || int(row_end) == OH) && int(row_end) <= OH, "row range must be 8-aligned");
        or0 = int(row_begin); or1 = int(row_end);
    }
    if (mask & 0x80000000u) for (bool& f : t->flip) f = !f;  // KEEP_TEMPORALS: same ping-pong assignment as the previous call (after every argument check)
    const int ir0 = or0, ir1 = or1;
    void *temporal_out, *history; t->pingpong("taa", 0, OB * 8, s, temporal_out, history);
    void *vel_out, *vel_hist;     t->pingpong("taa.velocity", 1, OB * 4, s, vel_out, vel_hist);
    void* reprojected_history = t->get("reprojected_history_img", OB * 8, s);
    void* closest_velocity = t->get("closest_velocity_img", OB * 4, s);
    void *sv_out, *sv_hist;       t->pingpong("taa.smooth_var", 2, OB * 8, s, sv_out, sv_hist);
    void* filtered_input = t->get("filtered_input_img", IB * 8, s);
    void* filtered_input_dev = t->get("filtered_input_deviation_img", IB * 8, s);
    void* filtered_history = t->get("filtered_history_img", IB * 8, s);
    void* input_prob = t->get("input_prob_img", IB * 2, s);
    void* prob1 = t->get("prob_filtered1_img", IB * 2, s);
    void* prob2 = t->get("prob_filtered2_img", IB * 2, s);
    void* this_frame = t->get("this_frame_output_img", OB * 8, s);
    KJ_TRY_HIP(t->err);
    const dim3 go((OW + 7) / 8, (or1 - or0 + 7) / 8), gi((IW + 7) / 8, (ir1 - ir0 + 7) / 8), blk(64);
    const ImgH4 input = img<uint2>(input_tex, IW, IH), reproj = img<uint2>(reprojection_map, IW, IH);
    const ImgF32 depth = img<float>(depth_tex, IW, IH);
    if (mask & 1u) {
        hipLaunchKernelGGL(k_taa_reproject, go, blk, 0, s, fc, img<uint2>(history, OW, OH), reproj, depth, img<uint2>(reprojected_history, OW, OH), img<uint32_t>(closest_velocity, OW, OH), IW, IH, or0, or1);
        KJ_CHECK_LAUNCH();
    }
    const bool both_input_filters = t->merge_prob_filters && (mask & 6u) == 6u && OW == IW && OH == IH;     // same extent, both passes: one launch
    if (both_input_filters) {
        hipLaunchKernelGGL(k_taa_filter_input_and_history, gi, blk, 0, s, input, depth, img<uint2>(filtered_input, IW, IH), img<uint2>(filtered_input_dev, IW, IH),
                           img<uint2>(reprojected_history, OW, OH), img<uint2>(filtered_history, IW, IH), ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    if ((mask & 2u) && !both_input_filters) {
        hipLaunchKernelGGL(k_taa_filter_input, gi, blk, 0, s, input, depth, img<uint2>(filtered_input, IW, IH), img<uint2>(filtered_input_dev, IW, IH), ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    if ((mask & 4u) && !both_input_filters) {
        const ImgH4 rh = img<uint2>(reprojected_history, OW, OH), fh = img<uint2>(filtered_history, IW, IH);
        if (float(OW) / float(IW) > 1.75f) hipLaunchKernelGGL((k_taa_filter_history<2, false>), gi, blk, 0, s, rh, fh, ir0, ir1);
        else if (OW == IW && OH == IH) hipLaunchKernelGGL((k_taa_filter_history<1, true>), gi, blk, 0, s, rh, fh, ir0, ir1);
        else hipLaunchKernelGGL((k_taa_filter_history<1, false>), gi, blk, 0, s, rh, fh, ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    if (mask & 8u) {
        hipLaunchKernelGGL(k_taa_input_prob, gi, blk, 0, s, fc, img<uint2>(filtered_input, IW, IH), img<uint2>(filtered_input_dev, IW, IH), img<uint2>(filtered_history, IW, IH), reproj,
                       img<uint2>(sv_hist, OW, OH), img<uint32_t>(vel_hist, OW, OH), img<uint16_t>(input_prob, IW, IH), ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    const bool both_prob_filters = t->merge_prob_filters && (mask & 48u) == 48u && ir0 == 0 && ir1 == IH;     // whole image, both passes: one launch through LDS
    if (both_prob_filters) {
        hipLaunchKernelGGL(k_taa_filter_prob_both, dim3((IW + 15) / 16, (IH + 15) / 16), dim3(256), 0, s, img<uint16_t>(input_prob, IW, IH), img<uint16_t>(prob1, IW, IH), img<uint16_t>(prob2, IW, IH));
        KJ_CHECK_LAUNCH();
    }
    if ((mask & 16u) && !both_prob_filters) {
        hipLaunchKernelGGL(k_taa_filter_prob, gi, blk, 0, s, img<uint16_t>(input_prob, IW, IH), img<uint16_t>(prob1, IW, IH), ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    if ((mask & 32u) && !both_prob_filters) {
        hipLaunchKernelGGL(k_taa_filter_prob2, gi, blk, 0, s, img<uint16_t>(prob1, IW, IH), img<uint16_t>(prob2, IW, IH), ir0, ir1);
        KJ_CHECK_LAUNCH();
    }
    TaaArgs a;
    a.fc = fc; a.input_tex = input; a.history_tex = img<uint2>(reprojected_history, OW, OH); a.reprojection_tex = reproj;
    a.closest_velocity_tex = img<uint32_t>(closest_velocity, OW, OH); a.velocity_history_tex = img<uint32_t>(vel_hist, OW, OH);
    a.smooth_var_history_tex = img<uint2>(sv_hist, OW, OH); a.input_prob_tex = img<uint16_t>(prob2, IW, IH);
    a.temporal_output_tex = img<uint2>(temporal_out, OW, OH); a.output_tex = img<uint2>(this_frame, OW, OH);
    a.smooth_var_output_tex = img<uint2>(sv_out, OW, OH); a.velocity_output_tex = img<uint32_t>(vel_out, OW, OH);
    a.row0 = or0; a.row1 = or1;
    if (mask & 64u) {
        if (IW == OW && IH == OH) hipLaunchKernelGGL(k_taa<true>, go, blk, 0, s, a);
        else hipLaunchKernelGGL(k_taa<false>, go, blk, 0, s, a);
        KJ_CHECK_LAUNCH();
    }
    out->temporal_out = temporal_out;
    out->this_frame_out = this_frame;
    return KJ_OK;
}
KjStatus kj_taa_render(KjTaa* t, const void* input_tex, uint32_t input_width, uint32_t input_height, const void* reprojection_map, const void* depth_tex,
                       uint32_t output_width, uint32_t output_height, KjTaaOutput* out, void* stream) {
    return taa_render_impl(t, input_tex, input_width, input_height, reprojection_map, depth_tex, output_width, output_height, out, stream, 127u, 0, 0);
}
// Pass-by-pass / strip variant for the screen-tile split: pass_mask bits 0..6 = reproject, filter input, filter history,
// input prob, prob filter, prob filter2, taa; bit 31 = keep the previous call's ping-pong assignment.
KjStatus kj_taa_render_rows(KjTaa* t, const void* input_tex, uint32_t input_width, uint32_t input_height, const void* reprojection_map, const void* depth_tex,
                            uint32_t output_width, uint32_t output_height, KjTaaOutput* out, void* stream, uint32_t pass_mask, uint32_t row_begin, uint32_t row_end) {
    return taa_render_impl(t, input_tex, input_width, input_height, reprojection_map, depth_tex, output_width, output_height, out, stream, pass_mask, row_begin, row_end);
}
KjStatus kj_taa_surface(KjTaa* t, const char* name, void** out_dev_ptr, uint64_t* out_bytes) {
    KJ_REQUIRE(t && name && out_dev_ptr && out_bytes, "null argument");
    auto it = t->surf.find(name);
    if (it == t->surf.end()) { set_last_error("no taa surface named '%s'", name); return KJ_ERR_INVALID_ARGUMENT; }
    *out_dev_ptr = it->second.p;
    *out_bytes = it->second.bytes;
    return KJ_OK;
}

}  // extern "C"

// ---- self test of kj_screen.hpp's div_nr / sqrt_nr against the IEEE operations, on the device: counts[0] = quotients that differ, counts[1] = roots that
// differ, counts[2] = quotients off by more than one ulp, counts[3] = roots off by more than one ulp, over `n` pseudo-random operand pairs
__global__ void __launch_bounds__(256) k_selftest_div_sqrt_nr(uint32_t n, uint32_t seed, unsigned long long* __restrict__ counts) {
    uint32_t bad_q = 0, bad_s = 0, far_q = 0, far_s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t h0 = hash1(i ^ seed), h1 = hash1(h0 + 0x9e3779b9u), h2 = hash1(h1 + i);
        // numerator and denominator: random mantissas, exponents in [-20, 20]; every 8th numerator negative
        const float num = __uint_as_float(((107u + h2 % 41u) << 23) | (h0 & 0x7fffffu)) * ((h2 >> 8) % 8u == 0u ? -1.0f : 1.0f);
        const float den = __uint_as_float(((107u + (h2 >> 16) % 41u) << 23) | (h1 & 0x7fffffu));
        float num_ = num, den_ = den;
        if (seed & 1u) {         // odd seeds: TAA's own operand classes -- half-integer texel positions over image extents, Catmull-Rom weight ratios
            if (i & 1u) { num_ = float(int(h0 % 4200u) - 3) + 0.5f + (h2 & 2u ? 0.0f : __uint_as_float(0x3f000000u | (h1 & 0x7fffffu)) - 0.5f); den_ = float(64u + h1 % 4033u); }
            else { const float f = __uint_as_float(0x3f800000u | (h0 & 0x7fffffu)) - 1.0f; const float w1 = 1.0f + f * f * (-2.5f + 1.5f * f), w2 = f * (0.5f + f * (2.0f - 1.5f * f)); num_ = w2; den_ = w1 + w2; }
        }
        if ((h1 >> 23) % 64u == 0u) num_ = (h2 & 4u) ? -0.0f : 0.0f;       // zero numerators of either sign
        const float q_ieee = num_ / den_, q_nr = div_nr(num_, den_);
        const float s_ieee = sqrtf(den), s_nr = sqrt_nr(den);
        if (__float_as_uint(q_ieee) != __float_as_uint(q_nr)) { ++bad_q; if (fabsf(q_ieee - q_nr) > fabsf(q_ieee) * 1.3e-7f) ++far_q; }
        if (__float_as_uint(s_ieee) != __float_as_uint(s_nr)) { ++bad_s; if (fabsf(s_ieee - s_nr) > fabsf(s_ieee) * 1.3e-7f) ++far_s; }
    }
    if (bad_q) atomicAdd(&counts[0], (unsigned long long)bad_q);
    if (bad_s) atomicAdd(&counts[1], (unsigned long long)bad_s);
    if (far_q) atomicAdd(&counts[2], (unsigned long long)far_q);
    if (far_s) atomicAdd(&counts[3], (unsigned long long)far_s);
}
extern "C" KjStatus kj_selftest_div_sqrt_nr(uint32_t n, uint32_t seed, void* counts4_u64_device, void* stream) {
    KJ_REQUIRE(counts4_u64_device, "null argument");
    hipLaunchKernelGGL(k_selftest_div_sqrt_nr, dim3(1024), dim3(256), 0, (hipStream_t)stream, n, seed, (unsigned long long*)counts4_u64_device);
    KJ_CHECK_LAUNCH();
    return KJ_OK;
}
