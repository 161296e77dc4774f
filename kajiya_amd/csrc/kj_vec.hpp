// Device math library for the gfx950 ReSTIR-GI kernels: vectors, HLSL-style
// intrinsics, RNG, packing and typed-format conversions.
// Behaviour follows the reference shader library (assets/shaders/inc/{math,hash,
// quasi_random,blue_noise,pack_unpack,gbuffer,color/*,working_color_space}.hlsl);
// file:line citations are on each block.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <float.h>

#define KJ_HD __host__ __device__ __forceinline__
#define KJ_D __device__ __forceinline__

namespace kj {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct I2 { int x, y; };

KJ_HD V2 v2(float x, float y) { return V2{x, y}; }
KJ_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
KJ_HD V3 v3(float s) { return V3{s, s, s}; }
KJ_HD V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
KJ_HD V4 v4(V3 a, float w) { return V4{a.x, a.y, a.z, w}; }
KJ_HD V4 v4(float s) { return V4{s, s, s, s}; }
KJ_HD V3 xyz(V4 a) { return V3{a.x, a.y, a.z}; }

#define KJ_VOP(op) \
    KJ_HD V2 operator op(V2 a, V2 b) { return V2{a.x op b.x, a.y op b.y}; } \
    KJ_HD V2 operator op(V2 a, float b) { return V2{a.x op b, a.y op b}; } \
    KJ_HD V2 operator op(float a, V2 b) { return V2{a op b.x, a op b.y}; } \
    KJ_HD V3 operator op(V3 a, V3 b) { return V3{a.x op b.x, a.y op b.y, a.z op b.z}; } \
    KJ_HD V3 operator op(V3 a, float b) { return V3{a.x op b, a.y op b, a.z op b}; } \
    KJ_HD V3 operator op(float a, V3 b) { return V3{a op b.x, a op b.y, a op b.z}; } \
    KJ_HD V4 operator op(V4 a, V4 b) { return V4{a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; } \
    KJ_HD V4 operator op(V4 a, float b) { return V4{a.x op b, a.y op b, a.z op b, a.w op b}; } \
    KJ_HD V4 operator op(float a, V4 b) { return V4{a op b.x, a op b.y, a op b.z, a op b.w}; }
KJ_VOP(+) KJ_VOP(-) KJ_VOP(*) KJ_VOP(/)
KJ_HD V2 operator-(V2 a) { return V2{-a.x, -a.y}; }
KJ_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
KJ_HD V4 operator-(V4 a) { return V4{-a.x, -a.y, -a.z, -a.w}; }
KJ_HD V2& operator+=(V2& a, V2 b) { a = a + b; return a; }
KJ_HD V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
KJ_HD V4& operator+=(V4& a, V4 b) { a = a + b; return a; }
KJ_HD V3& operator*=(V3& a, float b) { a = a * b; return a; }
KJ_HD V3& operator*=(V3& a, V3 b) { a = a * b; return a; }

KJ_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
KJ_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KJ_HD float dot(V4 a, V4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
KJ_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
KJ_HD float length(V2 a) { return sqrtf(dot(a, a)); }
KJ_HD float length(V3 a) { return sqrtf(dot(a, a)); }
KJ_HD V3 normalize(V3 a) { return a / sqrtf(dot(a, a)); }
KJ_HD float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
KJ_HD float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
// HLSL lerp -> SPIR-V FMix: x*(1-a) + y*a
KJ_HD float lerp(float a, float b, float t) { return a * (1.0f - t) + b * t; }
KJ_HD V2 lerp(V2 a, V2 b, float t) { return a * (1.0f - t) + b * t; }
KJ_HD V3 lerp(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
KJ_HD V4 lerp(V4 a, V4 b, float t) { return a * (1.0f - t) + b * t; }
KJ_HD float frac(float x) { return x - floorf(x); }
KJ_HD float stepf(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
KJ_HD float smoothstep(float a, float b, float x) {
    float t = saturate((x - a) / (b - a));
    return t * t * (3.0f - 2.0f * t);
}
KJ_HD float square(float x) { return x * x; }
KJ_HD float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
KJ_HD V2 vmin(V2 a, V2 b) { return V2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
KJ_HD V3 vmin(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
KJ_HD V3 vmax(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
KJ_HD V4 vmin(V4 a, V4 b) { return V4{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)}; }
KJ_HD V4 vmax(V4 a, V4 b) { return V4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
KJ_HD V3 vabs(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
KJ_HD V3 vsqrt(V3 a) { return V3{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
KJ_HD V4 vsqrt(V4 a) { return V4{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z), sqrtf(a.w)}; }
KJ_HD V3 vclamp(V3 v, V3 a, V3 b) { return vmin(vmax(v, a), b); }

KJ_HD uint32_t asuint(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
KJ_HD float asfloat(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
// saturating float->int (matches v_cvt_i32_f32; explicit so host/device agree), NaN -> 0
KJ_HD int f2i_sat(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return int(f);
}
KJ_HD int wrap_add(int a, int b) { return int(uint32_t(a) + uint32_t(b)); }
KJ_HD int wrap_mul2_add(int a, int b) { return int(uint32_t(a) * 2u + uint32_t(b)); }

#define KJ_PI 3.14159265358979323846f
#define KJ_TAU 6.28318530717958647692f
#define KJ_FRAC_1_PI 0.318309886183790671537767526745028724f
#define KJ_PLASTIC 1.32471795724474602596f
#define KJ_GOLDEN_ANGLE 2.39996323f

// column-major 4x4 times column vector (glam::Mat4 memory order)
KJ_HD V4 mul44(const float* m, V4 v) {
    return V4{m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
              m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w};
}
// Basis with columns (c0,c1,c2): to_world(v) = c0*v.x+c1*v.y+c2*v.z ; to_local(v) = dots
struct Basis { V3 c0, c1, c2; };
KJ_HD V3 to_world(const Basis& b, V3 v) { return b.c0 * v.x + b.c1 * v.y + b.c2 * v.z; }
KJ_HD V3 to_local(const Basis& b, V3 v) { return V3{dot(v, b.c0), dot(v, b.c1), dot(v, b.c2)}; }
// inc/math.hlsl:21-42 (Duff et al. orthonormal basis)
KJ_HD Basis build_orthonormal_basis(V3 n) {
    V3 b1, b2;
    if (n.z < 0.0f) {
        const float a = 1.0f / (1.0f - n.z);
        const float b = n.x * n.y * a;
        b1 = V3{1.0f - n.x * n.x * a, -b, n.x};
        b2 = V3{b, n.y * n.y * a - 1.0f, -n.y};
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        b1 = V3{1.0f - n.x * n.x * a, b, -n.x};
        b2 = V3{b, 1.0f - n.y * n.y * a, -n.y};
    }
    return Basis{b1, b2, n};
}
// inc/math.hlsl:44-49,72-77
KJ_HD V3 uniform_sample_cone(V2 urand, float cos_theta_max) {
    float cos_theta = (1.0f - urand.x) + urand.x * cos_theta_max;
    float sin_theta = sqrtf(saturate(1.0f - cos_theta * cos_theta));
    float phi = urand.y * KJ_TAU;
    return V3{sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta};
}
KJ_HD V3 uniform_sample_hemisphere(V2 urand) {
    float phi = urand.y * KJ_TAU;
    float cos_theta = 1.0f - urand.x;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return V3{cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
}
KJ_HD float inverse_depth_relative_diff(float primary_depth, float secondary_depth) {
    return fabsf(fmaxf(1e-20f, primary_depth) / fmaxf(1e-20f, secondary_depth) - 1.0f);
}

// sin/cos of a spiral-tap angle with the range reduction of v_sin_f32/v_cos_f32 (angle in revolutions,
// fract) — what HLSL sin/cos lower to on the reference's GPUs; keeps ~500 rad arguments well-defined.
KJ_HD V2 cos_sin_turns(float ang) {
#pragma clang fp contract(off)
    float t = ang * 0.15915494309189535f;
    t = t - floorf(t);
    const float a = t * KJ_TAU;
    return V2{cosf(a), sinf(a)};
}

// ---- RNG (inc/hash.hlsl:7-55, inc/quasi_random.hlsl:6-24)
KJ_HD uint32_t hash1(uint32_t x) {
    x += (x << 10u); x ^= (x >> 6u); x += (x << 3u); x ^= (x >> 11u); x += (x << 15u);
    return x;
}
KJ_HD uint32_t hash1_mut(uint32_t& h) { uint32_t r = h; h = hash1(h); return r; }
KJ_HD uint32_t hash_combine2(uint32_t x, uint32_t y) {
    uint32_t seed = (x * 1664525u + y + 1013904223u) * 1664525u;
    seed ^= (seed >> 11u);
    seed ^= (seed << 7u) & 0x9d2c5680u;
    seed ^= (seed << 15u) & 0xefc60000u;
    seed ^= (seed >> 18u);
    return seed;
}
KJ_HD uint32_t hash2(uint32_t x, uint32_t y) { return hash_combine2(x, hash1(y)); }
KJ_HD uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) { return hash_combine2(x, hash2(y, z)); }
KJ_HD float uint_to_u01_float(uint32_t h) { return asfloat((h & 0x007FFFFFu) | 0x3F800000u) - 1.0f; }
// frac() amplifies last-bit differences here, so this is evaluated without FMA contraction
// (every product rounds, as in the literal HLSL expression).
KJ_HD float interleaved_gradient_noise(uint32_t px, uint32_t py) {
#pragma clang fp contract(off)
    const float a = 0.06711056f * float(px), b = 0.00583715f * float(py);
    const float s = a + b;
    const float f = s - floorf(s);
    const float m = 52.9829189f * f;
    return m - floorf(m);
}
KJ_HD float radical_inverse_vdc(uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    bits = __brev(bits);
#else
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
#endif
    return float(bits) * 2.3283064365386963e-10f;
}
KJ_HD V2 hammersley(uint32_t i, uint32_t n) { return V2{float(i + 1) / float(n), radical_inverse_vdc(i + 1)}; }
KJ_HD V2 r2_sequence(uint32_t i) {
#pragma clang fp contract(off)
    const float a1 = 1.0f / KJ_PLASTIC;
    const float a2 = 1.0f / (KJ_PLASTIC * KJ_PLASTIC);
    const float x = a1 * float(i), y = a2 * float(i);
    const float xs = x + 0.5f, ys = y + 0.5f;
    return V2{xs - floorf(xs), ys - floorf(ys)};
}
// A texel code divided by its format's maximum (n / 255, / 127, / 1023, / 32767). On the device the IEEE division (11 instructions per channel, and every screen
// pass decodes the reprojection map, view normals, blue noise) is the constant reciprocal and one Newton step on the quotient through an exact fma residual:
// the correctly rounded quotient for EVERY code of each format (tests/test_oracle.py::test_texel_code_division_through_the_reciprocal_is_exact walks them all).
template <int MAXV> KJ_HD float texel_code_div(float n) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
    const float d = float(MAXV), r = 1.0f / float(MAXV), q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
#else
    return n / float(MAXV);
#endif
}
// inc/blue_noise.hlsl:8-15; tex = 256x256 RGBA8 packed as one u32 per texel
KJ_D V4 blue_noise_for_pixel(const uint32_t* __restrict__ tex, uint32_t px, uint32_t py, uint32_t n) {
    V2 r = r2_sequence(n);
    uint32_t ox = uint32_t(r.x * 256.0f), oy = uint32_t(r.y * 256.0f);
    uint32_t t = tex[((py + oy) & 255u) * 256u + ((px + ox) & 255u)];
    const float s = 255.0f / 256.0f, b = 0.5f / 256.0f;
    return V4{texel_code_div<255>(float(t & 255u)) * s + b, texel_code_div<255>(float((t >> 8) & 255u)) * s + b,
              texel_code_div<255>(float((t >> 16) & 255u)) * s + b, texel_code_div<255>(float(t >> 24)) * s + b};
}

// ---- fp16 storage (round-to-nearest-even, both directions exact)
KJ_D uint16_t f32_to_f16(float f) { return __half_as_ushort(__float2half_rn(f)); }
KJ_D float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
struct H4 { uint16_t x, y, z, w; };  // RGBA16F texel
struct H2 { uint16_t x, y; };        // RG16F texel
struct S4 { int16_t x, y, z, w; };   // RGBA16_SNORM texel
KJ_D uint2 pack_rgba16f(V4 v) {
    return make_uint2(uint32_t(f32_to_f16(v.x)) | (uint32_t(f32_to_f16(v.y)) << 16), uint32_t(f32_to_f16(v.z)) | (uint32_t(f32_to_f16(v.w)) << 16));
}
KJ_D V4 unpack_rgba16f(uint2 p) {
    return V4{f16_to_f32(uint16_t(p.x & 0xffff)), f16_to_f32(uint16_t(p.x >> 16)), f16_to_f32(uint16_t(p.y & 0xffff)), f16_to_f32(uint16_t(p.y >> 16))};
}
KJ_D uint32_t pack_2x16f_uint(float a, float b) { return uint32_t(f32_to_f16(a)) | (uint32_t(f32_to_f16(b)) << 16u); }
KJ_D V2 unpack_2x16f_uint(uint32_t u) { return V2{f16_to_f32(uint16_t(u & 0xffff)), f16_to_f32(uint16_t(u >> 16))}; }

// ---- packing (inc/pack_unpack.hlsl:4-162)
KJ_HD float unpack_unorm(uint32_t pckd, uint32_t bits) { uint32_t maxv = (1u << bits) - 1u; return float(pckd & maxv) / float(maxv); }
KJ_HD uint32_t pack_unorm(float v, uint32_t bits) { uint32_t maxv = (1u << bits) - 1u; return uint32_t(clampf(v, 0.0f, 1.0f) * float(maxv) + 0.5f); }
KJ_HD uint32_t pack_normal_11_10_11(V3 n) {
    return pack_unorm(n.x * 0.5f + 0.5f, 11) + (pack_unorm(n.y * 0.5f + 0.5f, 10) << 11) + (pack_unorm(n.z * 0.5f + 0.5f, 11) << 21);
}
KJ_HD V3 unpack_normal_11_10_11_no_normalize(uint32_t p) {
    return V3{unpack_unorm(p, 11), unpack_unorm(p >> 11, 10), unpack_unorm(p >> 21, 11)} * 2.0f - 1.0f;
}
KJ_HD V3 unpack_normal_11_10_11(uint32_t p) { return normalize(unpack_normal_11_10_11_no_normalize(p)); }
// inc/mesh.hlsl:27-33
KJ_HD V3 unpack_unit_direction_11_10_11(uint32_t pck) {
    return V3{float(pck & 2047u) * (2.0f / 2047.0f) - 1.0f, float((pck >> 11u) & 1023u) * (2.0f / 1023.0f) - 1.0f,
              float(pck >> 21u) * (2.0f / 2047.0f) - 1.0f};
}
KJ_HD uint32_t pack_color_888(V3 c) {
    c = vsqrt(c);
    return pack_unorm(c.x, 8) + (pack_unorm(c.y, 8) << 8) + (pack_unorm(c.z, 8) << 16);
}
KJ_HD V3 unpack_color_888(uint32_t p) {
    V3 c{unpack_unorm(p, 8), unpack_unorm(p >> 8, 8), unpack_unorm(p >> 16, 8)};
    return c * c;
}
KJ_HD uint32_t float3_to_rgb9e5(V3 rgb) {
    const float MAX_RGB9E5 = (511.0f / 512.0f) * 65536.0f;
    float rc = clampf(rgb.x, 0.0f, MAX_RGB9E5), gc = clampf(rgb.y, 0.0f, MAX_RGB9E5), bc = clampf(rgb.z, 0.0f, MAX_RGB9E5);
    float maxrgb = fmaxf(rc, fmaxf(gc, bc));
    int fl2 = int((asuint(maxrgb) & 0x7F800000u) >> 23) - 127;
    int exp_shared = (fl2 > -16 ? fl2 : -16) + 1 + 15;
    float denom = exp2f(float(exp_shared - 15 - 9));
    int maxm = int(floorf(maxrgb / denom + 0.5f));
    if (maxm == 512) { denom *= 2.0f; exp_shared += 1; }
    int rm = int(floorf(rc / denom + 0.5f)), gm = int(floorf(gc / denom + 0.5f)), bm = int(floorf(bc / denom + 0.5f));
    return (uint32_t(rm) << 23) | (uint32_t(gm) << 14) | (uint32_t(bm) << 5) | uint32_t(exp_shared);
}
KJ_HD V3 rgb9e5_to_float3(uint32_t v) {
    float scale = exp2f(float(int(v & 31u) - 24));
    return V3{float((v >> 23) & 511u) * scale, float((v >> 14) & 511u) * scale, float((v >> 5) & 511u) * scale};
}

// ---- typed-format conversions (fixed-function image load/store in the reference; RNE)
KJ_HD int8_t to_snorm8(float v) { return int8_t(rintf(clampf(v, -1.0f, 1.0f) * 127.0f)); }
KJ_HD float from_snorm8(int8_t v) { return fmaxf(texel_code_div<127>(float(v)), -1.0f); }
KJ_HD uint8_t to_unorm8(float v) { return uint8_t(rintf(clampf(v, 0.0f, 1.0f) * 255.0f)); }
KJ_HD float from_unorm8(uint8_t v) { return texel_code_div<255>(float(v)); }
KJ_HD int16_t to_snorm16(float v) { return int16_t(rintf(clampf(v, -1.0f, 1.0f) * 32767.0f)); }
KJ_HD float from_snorm16(int16_t v) { return fmaxf(texel_code_div<32767>(float(v)), -1.0f); }
KJ_HD uint32_t pack_a2r10g10b10(V3 rgb) {
    uint32_t r = uint32_t(rintf(clampf(rgb.x, 0.0f, 1.0f) * 1023.0f)), g = uint32_t(rintf(clampf(rgb.y, 0.0f, 1.0f) * 1023.0f)),
             b = uint32_t(rintf(clampf(rgb.z, 0.0f, 1.0f) * 1023.0f));
    return (r << 20) | (g << 10) | b;
}
KJ_HD V3 unpack_a2r10g10b10(uint32_t p) { return V3{texel_code_div<1023>(float((p >> 20) & 1023u)), texel_code_div<1023>(float((p >> 10) & 1023u)), texel_code_div<1023>(float(p & 1023u))}; }
KJ_HD uint32_t pack_rgba8_snorm(V4 v) {
    return uint32_t(uint8_t(to_snorm8(v.x))) | (uint32_t(uint8_t(to_snorm8(v.y))) << 8) | (uint32_t(uint8_t(to_snorm8(v.z))) << 16) | (uint32_t(uint8_t(to_snorm8(v.w))) << 24);
}
KJ_HD V4 unpack_rgba8_snorm(uint32_t p) {
    return V4{from_snorm8(int8_t(p & 0xff)), from_snorm8(int8_t((p >> 8) & 0xff)), from_snorm8(int8_t((p >> 16) & 0xff)), from_snorm8(int8_t(p >> 24))};
}

// ---- G-buffer (inc/gbuffer.hlsl:26-87)
struct GbufferData { V3 albedo, emissive, normal; float roughness, metalness; };
KJ_D uint4 gbuffer_pack(const GbufferData& g) {
    return make_uint4(pack_color_888(g.albedo), pack_normal_11_10_11(g.normal), pack_2x16f_uint(sqrtf(g.roughness), g.metalness), float3_to_rgb9e5(g.emissive));
}
KJ_D GbufferData gbuffer_unpack(uint4 d) {
    GbufferData g;
    g.albedo = unpack_color_888(d.x);
    g.normal = unpack_normal_11_10_11(d.y);
    V2 rm = unpack_2x16f_uint(d.z);
    g.roughness = rm.x * rm.x;
    g.metalness = rm.y;
    g.emissive = rgb9e5_to_float3(d.w);
    return g;
}

// ---- colour (inc/color/srgb.hlsl:4-6, ycbcr.hlsl:4-10, working_color_space.hlsl:9-19)
KJ_HD float sRGB_to_luminance(V3 c) { return dot(c, V3{0.2126f, 0.7152f, 0.0722f}); }
KJ_HD V3 sRGB_to_YCbCr(V3 c) { return V3{dot(V3{0.2126f, 0.7152f, 0.0722f}, c), dot(V3{-0.1146f, -0.3854f, 0.5f}, c), dot(V3{0.5f, -0.4542f, -0.0458f}, c)}; }
KJ_HD V3 YCbCr_to_sRGB(V3 c) { return vmax(v3(0.0f), V3{dot(V3{1.0f, 0.0f, 1.5748f}, c), dot(V3{1.0f, -0.1873f, -0.4681f}, c), dot(V3{1.0f, 1.8556f, 0.0f}, c)}); }
KJ_HD V4 linear_rgb_to_crunched_luma_chroma(V4 v) {
    V3 y = sRGB_to_YCbCr(xyz(v));
    float k = sqrtf(y.x) / fmaxf(1e-8f, y.x);
    return v4(y * k, v.w);
}
KJ_HD V4 crunched_luma_chroma_to_linear_rgb(V4 v) { return v4(YCbCr_to_sRGB(xyz(v) * v.x), v.w); }

// B10G11R11_UFLOAT_PACK32 (rtr's resolved image): r bits 0..10 (5e6m), g 11..21, b 22..31 (5e5m); stores round to nearest through fp16
#ifdef __HIPCC__
KJ_D uint32_t f32_to_ufloat(float v, int mant_bits) {              // see okj::f32_to_ufloat
    if (!(v > 0.0f)) return 0u;
    const uint32_t h = uint32_t(f32_to_f16(v)) & 0x7fffu;
    const int drop = 10 - mant_bits;
    const uint32_t r = (h + (1u << (drop - 1))) >> drop;
    const uint32_t max_finite = (30u << mant_bits) | ((1u << mant_bits) - 1u);
    if (h >= 0x7c00u) return 31u << mant_bits;
    return r > max_finite ? max_finite : r;
}
KJ_D float ufloat_to_f32(uint32_t v, int mant_bits) { return f16_to_f32(uint16_t(v << (10 - mant_bits))); }
KJ_D uint32_t pack_r11g11b10f(V3 c) { return f32_to_ufloat(c.x, 6) | (f32_to_ufloat(c.y, 6) << 11) | (f32_to_ufloat(c.z, 5) << 22); }
KJ_D V3 unpack_r11g11b10f(uint32_t p) { return V3{ufloat_to_f32(p & 0x7ffu, 6), ufloat_to_f32((p >> 11) & 0x7ffu, 6), ufloat_to_f32(p >> 22, 5)}; }
#endif

// ---- device ray counters: one atomic per wave per ray query. All waves hammering ONE address serialise in L2 (measured: 10 %
// of the trace kernel), so the counters are striped over KJ_COUNTER_SLOTS cache lines picked by workgroup id; readers sum them.
#define KJ_COUNTER_SLOTS 64u
#define KJ_COUNTER_STRIDE 16u   // u64 per slot = 128 B
#ifdef __HIPCC__
KJ_D unsigned long long* counter_slot(unsigned long long* base) {
    return base + ((blockIdx.x + blockIdx.y * gridDim.x) & (KJ_COUNTER_SLOTS - 1u)) * KJ_COUNTER_STRIDE;
}
#endif

// ---- XCD-aware tile order. MI355X dispatches workgroup `b` (x fastest) to XCD `b % 8`, each XCD with its own 4 MiB L2. With the
// plain blockIdx -> tile mapping horizontally adjacent tiles of a screen pass sit on eight different L2s (and, image widths being
// multiples of 64, an XCD owns 8-pixel COLUMNS spaced 64 pixels apart): every halo texel and every gather of a neighbour's
// reservoir is fetched into up to eight L2s. tile_order<MODE>() remaps the workgroup id to a tile so that neighbours share an XCD:
//   KJ_TILES_ROWS   whole tile rows, dealt round-robin (XCD k takes rows k, k + 8, ...): horizontal neighbours share an L2 and the
//                   eight rows in flight at any time are adjacent;
//   KJ_TILES_BANDS  column bands ~8 tiles wide and as tall as the launch, dealt round-robin: neighbours in both directions share an
//                   L2 except across band edges, and every XCD sees the image top to bottom (sky and ground alike);
//   KJ_TILES_SUPER  (round 6) SUPER-TILES of S x S tiles dealt round-robin in row-major order: XCD k works through super-tiles k, k + 8, ... -- a tile's
//                   neighbours inside its super-tile share its L2 (a 16-px reach around 64 x 64 px costs 2.25x the tile's own texels instead of every XCD
//                   pulling the whole image), and each XCD's super-tiles are scattered over the whole frame, so sky and geometry reach every XCD alike;
//   KJ_TILES_PLAIN  blockIdx as is.
// Which one a kernel uses is a measured choice per kernel (profiles/r03_xcd_tile_order.md): passes whose work per tile is uniform
// gain 3-17 % from sharing an L2 with their neighbours; passes that skip sky tiles (the ray passes, the resampling passes, the
// resolve) need the load balance of the plain order more than the locality -- contiguous eighths of the image, the textbook remap
// for GEMM tiles, made the trace pass 49 % slower (a launch lasts as long as its busiest XCD). Every mode is a bijection for any
// grid size (what does not fill a whole group of eight keeps the plain order). The dispatch order is not a contract
// (MI355X_MICROARCH.md, "Workgroup dispatch"): a different placement costs speed, never correctness.
#define KJ_TILES_PLAIN 0
#define KJ_TILES_ROWS 1
#define KJ_TILES_BANDS 2
#define KJ_TILES_SUPER 3
#ifdef __HIPCC__
template <int MODE_, int S_ = 4>
KJ_D uint2 tile_order() {
#ifdef KJ_TILES_ALL
    constexpr int MODE = KJ_TILES_ALL;          // experiment builds: one order for every kernel
#else
    constexpr int MODE = MODE_;
#endif
    const uint32_t gx = gridDim.x, gy = gridDim.y, id = blockIdx.y * gx + blockIdx.x;
    if (MODE == KJ_TILES_SUPER) {
        constexpr uint32_t S = uint32_t(S_), SS = S * S;
        const uint32_t sgx = gx / S, sgy = gy / S;                 // whole super-tiles
        const uint32_t n_super = sgx * sgy, n_super8 = n_super & ~7u;
        const uint32_t n_a = n_super8 * SS;                         // tiles of the super-tiles dealt eight at a time
        if (id < n_a) {
            const uint32_t xcd = id & 7u, k = id >> 3;              // the k-th workgroup this XCD receives
            const uint32_t j = k / SS, w = k - j * SS;              // its j-th super-tile = super-tile j * 8 + xcd, tile w inside it
            const uint32_t q = j * 8u + xcd, sy = q / sgx, sx = q - sy * sgx, wy = w / S;
            return make_uint2(sx * S + (w - wy * S), sy * S + wy);
        }
        uint32_t r = id - n_a;
        const uint32_t n_left = (n_super - n_super8) * SS;          // the < 8 super-tiles left over, one after the other
        if (r < n_left) {
            const uint32_t q = n_super8 + r / SS, w = r % SS, sy = q / sgx, sx = q - sy * sgx, wy = w / S;
            return make_uint2(sx * S + (w - wy * S), sy * S + wy);
        }
        r -= n_left;
        const uint32_t rw = gx - sgx * S;                           // the columns right of the super-tiles, every row
        if (r < rw * gy) { const uint32_t ry = r / rw; return make_uint2(sgx * S + (r - ry * rw), ry); }
        r -= rw * gy;
        const uint32_t bw = sgx * S, by = r / (bw ? bw : 1u);       // the rows below them
        return make_uint2(r - by * bw, sgy * S + by);
    }
    if (MODE == KJ_TILES_ROWS) {
        const uint32_t n_full = (gy & ~7u) * gx;                    // tiles in whole groups of eight rows
        if (id >= n_full) return make_uint2(blockIdx.x, blockIdx.y);
        const uint32_t xcd = id & 7u, k = id >> 3;                  // the k-th workgroup this XCD receives
        const uint32_t c = k / gx;                                  // its c-th row = row c * 8 + xcd of the image
        return make_uint2(k - c * gx, c * 8u + xcd);
    }
    if (MODE == KJ_TILES_BANDS) {
        const uint32_t g = gx >= 96u ? (gx + 32u) / 64u : 1u;       // bands per XCD (about 8 tiles wide each)
        const uint32_t bw = gx / (8u * g);                          // band width in tiles
        const uint32_t n_full = 8u * g * bw * gy;                   // tiles inside the bands; the columns right of them keep the plain order
        if (bw == 0u || id >= n_full) {
            const uint32_t rest = id - (bw ? n_full : 0u), rw = gx - 8u * g * bw, ry = rest / rw;
            return make_uint2(8u * g * bw + (rest - ry * rw), ry);
        }
        const uint32_t xcd = id & 7u, k = id >> 3;
        const uint32_t per_band = bw * gy;
        const uint32_t b = k / per_band, w = k - b * per_band;      // this XCD's b-th band = band b * 8 + xcd of the image
        const uint32_t wy = w / bw;
        return make_uint2((b * 8u + xcd) * bw + (w - wy * bw), wy);
    }
    return make_uint2(blockIdx.x, blockIdx.y);
}
#endif

// ---- flat-buffer "textures": OOB load = 0, OOB store dropped
template <typename T> struct Img {
    T* p; int w, h;
    KJ_HD bool inb(int x, int y) const { return uint32_t(x) < uint32_t(w) && uint32_t(y) < uint32_t(h); }
    // OOB load = 0. Branch-free: always load (texel 0 when out of bounds) and select the VALUE -- selecting between
    // &p[i] and the address of a zero temporary makes the compiler spill the temporary to scratch and use flat loads.
    KJ_D T ld(int x, int y) const {
        const bool in = inb(x, y);
        const T v = p[in ? size_t(y) * w + x : size_t(0)];
        return in ? v : T();
    }
    // the same load with the select left to the caller (`in` false: the value is texel 0's, to be replaced by T()): kernels that put many gathers in flight before
    // using any of them keep the selects -- the first USE of each loaded value -- out of the issue sequence
    KJ_D T ld_raw(int x, int y, bool& in) const {
        in = inb(x, y);
        return p[in ? size_t(y) * w + x : size_t(0)];
    }
    KJ_D T ldc(int x, int y) const {  // clamp-to-edge
        x = x < 0 ? 0 : (x >= w ? w - 1 : x); y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        return p[size_t(y) * w + x];
    }
    KJ_D void st(int x, int y, T v) const { if (inb(x, y)) p[size_t(y) * w + x] = v; }
};
template <typename T> KJ_HD Img<T> img(const void* p, int w, int h) { return Img<T>{(T*)p, w, h}; }
KJ_D V4 ld4(const Img<uint2>& i, int x, int y) { return unpack_rgba16f(i.ld(x, y)); }
KJ_D void st4(const Img<uint2>& i, int x, int y, V4 v) { i.st(x, y, pack_rgba16f(v)); }
KJ_D V2 ld2h(const Img<uint32_t>& i, int x, int y) { return unpack_2x16f_uint(i.ld(x, y)); }
KJ_D void st2h(const Img<uint32_t>& i, int x, int y, V2 v) { i.st(x, y, pack_2x16f_uint(v.x, v.y)); }
KJ_D V4 ld_reproj(const Img<uint2>& i, int x, int y) {  // RGBA16_SNORM
    uint2 p = i.ld(x, y);
    return V4{from_snorm16(int16_t(p.x & 0xffff)), from_snorm16(int16_t(p.x >> 16)), from_snorm16(int16_t(p.y & 0xffff)), from_snorm16(int16_t(p.y >> 16))};
}
KJ_D V3 ld_nrm_snorm8(const Img<uint32_t>& i, int x, int y) { return xyz(unpack_rgba8_snorm(i.ld(x, y))); }
// bilinear, clamp-to-edge, normalised uv (sampler_lnc / sampler_llc)
KJ_D V4 sample_bilinear_clamp_rgba16f(const uint2* __restrict__ p, int w, int h, V2 uv) {
    float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1), ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
    V4 s00 = unpack_rgba16f(p[ya * w + xa]), s10 = unpack_rgba16f(p[ya * w + xb]);
    V4 s01 = unpack_rgba16f(p[yb * w + xa]), s11 = unpack_rgba16f(p[yb * w + xb]);
    V4 a = s00 * (1.0f - tx) + s10 * tx;
    V4 b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}
KJ_D V2 sample_bilinear_clamp_rg16f(const uint32_t* __restrict__ p, int w, int h, V2 uv) {
    float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1), ya = min(max(y0, 0), h - 1), yb = min(max(y0 + 1, 0), h - 1);
    V2 s00 = unpack_2x16f_uint(p[ya * w + xa]), s10 = unpack_2x16f_uint(p[ya * w + xb]);
    V2 s01 = unpack_2x16f_uint(p[yb * w + xa]), s11 = unpack_2x16f_uint(p[yb * w + xb]);
    V2 a = s00 * (1.0f - tx) + s10 * tx;
    V2 b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}
template <typename T> KJ_D T sample_nearest_clamp(const Img<T>& i, V2 uv) {
    return i.ldc(int(floorf(uv.x * float(i.w))), int(floorf(uv.y * float(i.h))));
}

} // namespace kj
