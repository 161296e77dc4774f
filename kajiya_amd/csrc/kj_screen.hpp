// Building blocks of the gfx950 screen-space passes (ReSTIR spatial resampling, resolve, GI spatial filter):
//  * FrameDerived: per-frame matrix products computed once (k_frame_begin) so that a view-ray reconstruction -- which the
//    reference's ViewRayContext (inc/frame_constants.hlsl:84-215) spells as two 4x4 products per position, three positions per
//    context -- is one 4x3 affine evaluation + one reciprocal per position here;
//  * single-instruction reciprocal / rsqrt / sqrt (v_rcp_f32, v_rsq_f32, v_sqrt_f32: 1 ulp) where the reference's result feeds a
//    weight, not a discrete decision;
//  * cos/sin of an angle given in revolutions by quadrant reduction + two short polynomials (what `sin`/`cos` of a golden-angle
//    spiral lower to on the reference's GPUs is v_sin/v_cos of the angle in revolutions; libm's sinf/cosf on radians is 235
//    VALU instructions on gfx950);
//  * wave votes. The CPU stand-in for HIP used by the tests (tests/hip_emu) runs lanes one at a time; there a vote degenerates to
//    the lane's own predicate, which gives the same results because votes are only used to skip work no lane needs.
#pragma once
#include "kj_shading.hpp"

namespace kj {

struct FrameDerived {
    float sample_to_world[16];   // view_to_world * sample_to_view   (clip-space sample position -> world, homogeneous)
    float world_to_clip[16];     // view_to_clip * world_to_view
    float world_to_sample[16];   // view_to_sample * world_to_view
    float eye_ws[4];
};
// one ring slot of KjDevice::frame_constants: the caller's block followed by what k_frame_begin derives from it
struct FrameBlock { KjFrameConstants fc; FrameDerived fd; };
KJ_HD const FrameDerived& frame_derived(const FrameConstants* fc) { return ((const FrameBlock*)fc)->fd; }

#if defined(__HIP_DEVICE_COMPILE__)
KJ_D float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
KJ_D float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
KJ_D float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
KJ_D float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
KJ_D float log2_fast(float x) { return __builtin_amdgcn_logf(x); }
#else
KJ_HD float rcp_fast(float x) { return 1.0f / x; }
KJ_HD float rsq_fast(float x) { return 1.0f / sqrtf(x); }
KJ_HD float sqrt_fast(float x) { return sqrtf(x); }
KJ_HD float exp2_fast(float x) { return exp2f(x); }
KJ_HD float log2_fast(float x) { return log2f(x); }
#endif
// Division and square root for code that has to keep the reference's results (nearly always) bit for bit and cannot afford the ~11 / ~18
// instruction IEEE sequences (TAA upstream of its hypersensitive probability stage): the hardware estimate (v_rcp_f32 / v_rsq_f32, 1 ulp) followed
// by ONE Newton step on the RESULT through an exact fma residual -- e = fma(-d, q, n), q' = fma(e, r, q) -- whose error before the single final rounding
// is second order (~1e-14 relative): the correctly rounded result except where the exact one lies within ~1e-14 of a rounding boundary (about
// 1e-7 of operand pairs, then one ulp off). 4 instructions per quotient (3 when the reciprocal is shared by the components of a vector), 5 per root of a positive normal number.
// The quotient of a zero numerator keeps IEEE's sign. Domain: d finite, normal, non-zero; n finite; results in the normal range (the IEEE sequence's operand scaling and its fix-up of infinities are
// what is left out). The CPU stand-in (tests/hip_emu) divides and takes the root exactly.
#if defined(__HIP_DEVICE_COMPILE__)
KJ_D float div_nr_with(float n, float d, float r) {
#pragma clang fp contract(off)      // n * r must round by itself: with contraction allowed, a numerator `a - 1.0f` is folded INTO the multiply ((a - 1) r -> fma(a, r, -r): measured)
    const float q = n * r;
    // the sign is q's: for n = -0 the refinement ends in (+0) + (-0) = +0 where the quotient is -0, and TAA's luma weights turn on the sign of a zero
    // (cutoff / -0 = -inf -> weight 0, cutoff / +0 -> 1: the reference's own ill-conditioning; measured: 0.3 % of the history's texels without this)
    return __builtin_copysignf(__builtin_fmaf(__builtin_fmaf(-d, q, n), r, q), q);
}
KJ_D float div_nr(float n, float d) { return div_nr_with(n, d, __builtin_amdgcn_rcpf(d)); }
KJ_D float sqrt_nr_pos(float x) {        // x a positive, finite, normal number
#pragma clang fp contract(off)
    const float y = __builtin_amdgcn_rsqf(x), s = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
KJ_D float sqrt_nr(float x) {
    if (x >= FLT_MIN && x < INFINITY) {
        return sqrt_nr_pos(x);
    }
    return sqrtf(x);                     // 0, denormals, negative, NaN, inf: the IEEE sequence
}
#else
KJ_HD float div_nr_with(float n, float d, float) { return n / d; }
KJ_HD float div_nr(float n, float d) { return n / d; }
KJ_HD float sqrt_nr(float x) { return sqrtf(x); }
KJ_HD float sqrt_nr_pos(float x) { return sqrtf(x); }
#endif
KJ_HD V2 div_nr(V2 n, float d) { const float r = rcp_fast(d); return V2{div_nr_with(n.x, d, r), div_nr_with(n.y, d, r)}; }
KJ_HD V3 div_nr(V3 n, float d) { const float r = rcp_fast(d); return V3{div_nr_with(n.x, d, r), div_nr_with(n.y, d, r), div_nr_with(n.z, d, r)}; }
KJ_HD V4 div_nr(V4 n, float d) { const float r = rcp_fast(d); return V4{div_nr_with(n.x, d, r), div_nr_with(n.y, d, r), div_nr_with(n.z, d, r), div_nr_with(n.w, d, r)}; }
KJ_HD V2 div_nr(V2 n, V2 d) { return V2{div_nr(n.x, d.x), div_nr(n.y, d.y)}; }
KJ_HD float length_fast(V3 a) { return sqrt_fast(dot(a, a)); }
KJ_HD float length_fast(V2 a) { return sqrt_fast(dot(a, a)); }
KJ_HD V3 normalize_fast(V3 a) { return a * rsq_fast(dot(a, a)); }
KJ_HD float smoothstep_fast(float a, float b, float x) {
    const float t = saturate((x - a) * rcp_fast(b - a));
    return t * t * (3.0f - 2.0f * t);
}

// m * (x, y, z, 1), column-major
KJ_HD V4 affine44(const float* m, float x, float y, float z) {
    return V4{m[0] * x + m[4] * y + m[8] * z + m[12], m[1] * x + m[5] * y + m[9] * z + m[13],
              m[2] * x + m[6] * y + m[10] * z + m[14], m[3] * x + m[7] * y + m[11] * z + m[15]};
}
// ViewRayContext::from_uv_and_depth(...).ray_hit_ws() / .ray_hit_vs() for a clip-space sample position
KJ_HD V3 hit_ws_from_cs(const FrameDerived& fd, V2 cs, float depth) {
    const V4 h = affine44(fd.sample_to_world, cs.x, cs.y, depth);
    return xyz(h) * rcp_fast(h.w);
}
KJ_HD V3 hit_vs_from_cs(const FrameConstants& fc, V2 cs, float depth) {
    const V4 h = affine44(fc.view_constants.sample_to_view, cs.x, cs.y, depth);
    return xyz(h) * rcp_fast(h.w);
}
KJ_HD V3 world_to_clip_fast(const FrameDerived& fd, V3 p) {
    const V4 h = affine44(fd.world_to_clip, p.x, p.y, p.z);
    return xyz(h) * rcp_fast(h.w);
}
// clip-space xy of the centre of half-res pixel (x, y) whose full-res representative is (2x + off.x, 2y + off.y):
// uv_to_cs(get_uv(2x + off, full_size)) = x * a + b per axis -- one FMA per axis per tap.
struct HalfPxToCs {
    float ax, bx, ay, by;
    KJ_HD static HalfPxToCs make(int full_w, int full_h, I2 off) {
        const float iw = 1.0f / float(full_w), ih = 1.0f / float(full_h);
        return HalfPxToCs{4.0f * iw, (2.0f * float(off.x) + 1.0f) * iw - 1.0f, -4.0f * ih, 1.0f - (2.0f * float(off.y) + 1.0f) * ih};
    }
    KJ_HD V2 operator()(int x, int y) const { return V2{float(x) * ax + bx, float(y) * ay + by}; }
};

// cos, sin of 2*pi*frac(ang / 2*pi): `ang` in radians, reduced in revolutions exactly as cos_sin_turns() does, then evaluated on
// [-pi/4, pi/4] (minimax coefficients of the classic single-precision kernels; error < 1 ulp of the result's quadrant).
KJ_HD V2 cos_sin_turns_fast(float ang) {
#pragma clang fp contract(off)
    float t = ang * 0.15915494309189535f;
    t = t - floorf(t);
    const float q = rintf(t * 4.0f);                  // nearest quarter turn, 0..4
    const float r = t - q * 0.25f;                    // exact: |r| <= 1/8 turn
    const float x = r * KJ_TAU;
    const float x2 = x * x;
    const float s = x + x * x2 * (-1.6666654611e-1f + x2 * (8.3321608736e-3f + x2 * -1.9515295891e-4f));
    const float c = 1.0f - 0.5f * x2 + x2 * x2 * (4.166664568298827e-2f + x2 * (-1.388731625493765e-3f + x2 * 2.443315711809948e-5f));
    const int qi = int(q) & 3;
    const float cc = (qi & 1) ? s : c, ss = (qi & 1) ? c : s;
    return V2{(qi == 1 || qi == 2) ? -cc : cc, (qi >= 2) ? -ss : ss};
}

// ---- wave votes / broadcasts (wave64)
#if defined(__HIP_DEVICE_COMPILE__)
KJ_D bool wave_any(bool p) { return __ballot(p) != 0ull; }
KJ_D float wave_read(float v, int src_lane) { return __shfl(v, src_lane); }
#define KJ_WAVE_SHARED 1
#else
KJ_HD bool wave_any(bool p) { return p; }
#define KJ_WAVE_SHARED 0
#endif

} // namespace kj
