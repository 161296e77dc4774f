// ORACLE (test infrastructure, never shipped, never called by the product path).
// CPU restatement of the reference's shader math library. Each block cites the
// reference file it restates (paths relative to /root/reference/assets/shaders).
// Scalar C++, no intrinsics; compiled with -ffp-contract=off so every operation
// rounds exactly once, as written.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <algorithm>

namespace okj {

// ------------------------------------------------------------------ vectors
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct i2 { int x, y; };
struct u2 { uint32_t x, y; };
struct u4 { uint32_t x, y, z, w; };

static inline f2 mk2(float x, float y) { return f2{x, y}; }
static inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
static inline f3 mk3(float s) { return f3{s, s, s}; }
static inline f4 mk4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
static inline f4 mk4(f3 v, float w) { return f4{v.x, v.y, v.z, w}; }
static inline f4 mk4(float s) { return f4{s, s, s, s}; }
static inline f3 xyz(f4 v) { return f3{v.x, v.y, v.z}; }

#define OKJ_OP2(T, op) \
    static inline T operator op(T a, T b) { return T{a.x op b.x, a.y op b.y}; } \
    static inline T operator op(T a, float b) { return T{a.x op b, a.y op b}; } \
    static inline T operator op(float a, T b) { return T{a op b.x, a op b.y}; }
#define OKJ_OP3(T, op) \
    static inline T operator op(T a, T b) { return T{a.x op b.x, a.y op b.y, a.z op b.z}; } \
    static inline T operator op(T a, float b) { return T{a.x op b, a.y op b, a.z op b}; } \
    static inline T operator op(float a, T b) { return T{a op b.x, a op b.y, a op b.z}; }
#define OKJ_OP4(T, op) \
    static inline T operator op(T a, T b) { return T{a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; } \
    static inline T operator op(T a, float b) { return T{a.x op b, a.y op b, a.z op b, a.w op b}; } \
    static inline T operator op(float a, T b) { return T{a op b.x, a op b.y, a op b.z, a op b.w}; }
OKJ_OP2(f2, +) OKJ_OP2(f2, -) OKJ_OP2(f2, *) OKJ_OP2(f2, /)
OKJ_OP3(f3, +) OKJ_OP3(f3, -) OKJ_OP3(f3, *) OKJ_OP3(f3, /)
OKJ_OP4(f4, +) OKJ_OP4(f4, -) OKJ_OP4(f4, *) OKJ_OP4(f4, /)
static inline f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
static inline f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
static inline f4 operator-(f4 a) { return f4{-a.x, -a.y, -a.z, -a.w}; }
static inline f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
static inline f4& operator+=(f4& a, f4 b) { a = a + b; return a; }
static inline f2& operator+=(f2& a, f2 b) { a = a + b; return a; }
static inline f3& operator*=(f3& a, float b) { a = a * b; return a; }
static inline f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
static inline i2 operator+(i2 a, i2 b) { return i2{a.x + b.x, a.y + b.y}; }
static inline i2 operator-(i2 a, i2 b) { return i2{a.x - b.x, a.y - b.y}; }
static inline i2 operator*(i2 a, int b) { return i2{a.x * b, a.y * b}; }

static inline float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline f3 cross(f3 a, f3 b) {
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(f2 a) { return sqrtf(dot(a, a)); }
static inline float length(f3 a) { return sqrtf(dot(a, a)); }
static inline f3 normalize(f3 a) { return a / sqrtf(dot(a, a)); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
// GLSL.std.450 FMix: x*(1-a) + y*a  (DXC lowers HLSL `lerp` to it)
static inline float lerp(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline f3 lerp(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
static inline f4 lerp(f4 a, f4 b, float t) { return a * (1.0f - t) + b * t; }
static inline f2 lerp(f2 a, f2 b, float t) { return a * (1.0f - t) + b * t; }
static inline float frac(float x) { return x - floorf(x); }
static inline float step(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
static inline float smoothstep(float a, float b, float x) {
    float t = saturate((x - a) / (b - a));
    return t * t * (3.0f - 2.0f * t);
}
static inline float square(float x) { return x * x; }
static inline float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
static inline f3 vmax(f3 a, f3 b) { return f3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
static inline f3 vmin(f3 a, f3 b) { return f3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
static inline f4 vmax(f4 a, f4 b) { return f4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
static inline f4 vmin(f4 a, f4 b) { return f4{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)}; }
static inline f2 vmin(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
static inline f3 vabs(f3 a) { return f3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
static inline f3 vsqrt(f3 a) { return f3{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
static inline f4 vsqrt(f4 a) { return f4{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z), sqrtf(a.w)}; }
static inline f3 vclamp(f3 v, f3 a, f3 b) { return vmin(vmax(v, a), b); }

static const float M_PI_F = 3.14159265358979323846f;
static const float M_TAU_F = 6.28318530717958647692f;
// float -> int32 conversion with the GPU's saturating behaviour (v_cvt_i32_f32); NaN -> 0
static inline int f2i_sat(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return int(f);
}
static inline int wrap_add(int a, int b) { return int(uint32_t(a) + uint32_t(b)); }
static inline int wrap_mul2_add(int a, int b) { return int(uint32_t(a) * 2u + uint32_t(b)); }
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static const float M_FRAC_1_PI_F = 0.318309886183790671537767526745028724f;
static const float M_PLASTIC_F = 1.32471795724474602596f;
static const float GOLDEN_ANGLE = 2.39996323f;

// Column-major 4x4 (glam::Mat4 memory order); mul(M, v) with column vectors
// (inc/frame_constants.hlsl:148-153).
static inline f4 mul44(const float* m, f4 v) {
    return f4{
        m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w,
        m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
        m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w,
        m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w};
}

// 3x3 whose COLUMNS are c0,c1,c2. `mul(M, v)` = c0*v.x + c1*v.y + c2*v.z;
// `mul(v, M)` = (dot(v,c0), dot(v,c1), dot(v,c2)).
struct m33 { f3 c0, c1, c2; };
static inline f3 mul(const m33& m, f3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }
static inline f3 mul(f3 v, const m33& m) { return f3{dot(v, m.c0), dot(v, m.c1), dot(v, m.c2)}; }

// inc/math.hlsl:21-42 — returns matrix with columns (b1, b2, n)
static inline m33 build_orthonormal_basis(f3 n) {
    f3 b1, b2;
    if (n.z < 0.0f) {
        const float a = 1.0f / (1.0f - n.z);
        const float b = n.x * n.y * a;
        b1 = f3{1.0f - n.x * n.x * a, -b, n.x};
        b2 = f3{b, n.y * n.y * a - 1.0f, -n.y};
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        b1 = f3{1.0f - n.x * n.x * a, b, -n.x};
        b2 = f3{b, 1.0f - n.y * n.y * a, -n.y};
    }
    return m33{b1, b2, n};
}

// inc/math.hlsl:44-49, 72-77
static inline f3 uniform_sample_cone(f2 urand, float cos_theta_max) {
    float cos_theta = (1.0f - urand.x) + urand.x * cos_theta_max;
    float sin_theta = sqrtf(saturate(1.0f - cos_theta * cos_theta));
    float phi = urand.y * M_TAU_F;
    return f3{sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta};
}
static inline f3 uniform_sample_hemisphere(f2 urand) {
    float phi = urand.y * M_TAU_F;
    float cos_theta = 1.0f - urand.x;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return f3{cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
}
static inline float inverse_depth_relative_diff(float primary_depth, float secondary_depth) {
    return fabsf(fmaxf(1e-20f, primary_depth) / fmaxf(1e-20f, secondary_depth) - 1.0f);
}

// sin/cos of a spiral-tap angle. The reference ran on GCN/RDNA hardware where HLSL sin/cos lower to
// v_sin_f32/v_cos_f32, which take the angle in revolutions (x * 1/2pi, fract). We restate that range
// reduction explicitly so the large angles of the denoiser kernels (up to ~500 rad in
// rtdgi/spatial_filter.hlsl:58) do not depend on a libm's argument reduction.
static inline f2 cos_sin_turns(float ang) {
    float t = ang * 0.15915494309189535f;
    t = t - floorf(t);
    const float a = t * M_TAU_F;
    return f2{cosf(a), sinf(a)};
}

// ------------------------------------------------------------------ hash (inc/hash.hlsl:7-55)
static inline uint32_t hash1(uint32_t x) {
    x += (x << 10u);
    x ^= (x >> 6u);
    x += (x << 3u);
    x ^= (x >> 11u);
    x += (x << 15u);
    return x;
}
static inline uint32_t hash1_mut(uint32_t& h) { uint32_t r = h; h = hash1(h); return r; }
static inline uint32_t hash_combine2(uint32_t x, uint32_t y) {
    const uint32_t M = 1664525u, C = 1013904223u;
    uint32_t seed = (x * M + y + C) * M;
    seed ^= (seed >> 11u);
    seed ^= (seed << 7u) & 0x9d2c5680u;
    seed ^= (seed << 15u) & 0xefc60000u;
    seed ^= (seed >> 18u);
    return seed;
}
static inline uint32_t hash2(uint32_t x, uint32_t y) { return hash_combine2(x, hash1(y)); }
static inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) { return hash_combine2(x, hash2(y, z)); }
static inline float uint_to_u01_float(uint32_t h) {
    h &= 0x007FFFFFu;
    h |= 0x3F800000u;
    return asfloat(h) - 1.0f;
}
static inline float interleaved_gradient_noise(uint32_t px, uint32_t py) {
    return frac(52.9829189f * frac(0.06711056f * float(px) + 0.00583715f * float(py)));
}
// inc/quasi_random.hlsl:6-24
static inline float radical_inverse_vdc(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return float(bits) * 2.3283064365386963e-10f;
}
static inline f2 hammersley(uint32_t i, uint32_t n) {
    return f2{float(i + 1) / float(n), radical_inverse_vdc(i + 1)};
}
static inline f2 r2_sequence(uint32_t i) {
    const float a1 = 1.0f / M_PLASTIC_F;
    const float a2 = 1.0f / (M_PLASTIC_F * M_PLASTIC_F);
    return f2{frac(a1 * float(i) + 0.5f), frac(a2 * float(i) + 0.5f)};
}
// inc/blue_noise.hlsl:8-15; `tex` = 256x256 RGBA8
static inline f4 blue_noise_for_pixel(const uint8_t* tex, uint32_t px, uint32_t py, uint32_t n) {
    f2 r = r2_sequence(n);
    uint32_t ox = uint32_t(r.x * 256.0f), oy = uint32_t(r.y * 256.0f);
    uint32_t x = (px + ox) % 256u, y = (py + oy) % 256u;
    const uint8_t* t = tex + (size_t(y) * 256 + x) * 4;
    const float s = 255.0f / 256.0f, b = 0.5f / 256.0f;
    // texel fetch of RGBA8_UNORM returns v/255
    return f4{(t[0] / 255.0f) * s + b, (t[1] / 255.0f) * s + b, (t[2] / 255.0f) * s + b, (t[3] / 255.0f) * s + b};
}

// ------------------------------------------------------------------ fp16 (RTE both ways)
static inline uint16_t f32_to_f16(float f) {
    uint32_t x = asuint(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t exp = int32_t((x >> 23) & 0xff);
    if (exp == 255) return uint16_t(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
    int32_t e = exp - 127 + 15;
    if (e >= 31) return uint16_t(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return uint16_t(sign);
        mant |= 0x00800000u;
        uint32_t shift = uint32_t(14 - e);
        uint32_t hm = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return uint16_t(sign | hm);
    }
    uint32_t hm = mant >> 13;
    uint32_t rem = mant & 0x1fffu;
    uint32_t h = (uint32_t(e) << 10) | hm;
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) h++;
    return uint16_t(sign | h);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t mant = h & 0x3ffu;
    if (exp == 0) {
        if (mant == 0) return asfloat(sign);
        float v = float(mant) * (1.0f / 16777216.0f); // 2^-24
        return (sign ? -v : v);
    }
    if (exp == 31) return asfloat(sign | 0x7f800000u | (mant << 13));
    return asfloat(sign | ((exp + 112u) << 23) | (mant << 13));
}

// ------------------------------------------------------------------ packing (inc/pack_unpack.hlsl)
static inline float unpack_unorm(uint32_t pckd, uint32_t bits) {
    uint32_t maxv = (1u << bits) - 1u;
    return float(pckd & maxv) / float(maxv);
}
static inline uint32_t pack_unorm(float v, uint32_t bits) {
    uint32_t maxv = (1u << bits) - 1u;
    return uint32_t(clampf(v, 0.0f, 1.0f) * float(maxv) + 0.5f);
}
static inline uint32_t pack_normal_11_10_11(f3 n) {
    uint32_t p = 0;
    p += pack_unorm(n.x * 0.5f + 0.5f, 11);
    p += pack_unorm(n.y * 0.5f + 0.5f, 10) << 11;
    p += pack_unorm(n.z * 0.5f + 0.5f, 11) << 21;
    return p;
}
static inline f3 unpack_normal_11_10_11_no_normalize(uint32_t p) {
    return f3{unpack_unorm(p, 11), unpack_unorm(p >> 11, 10), unpack_unorm(p >> 21, 11)} * 2.0f - 1.0f;
}
static inline f3 unpack_normal_11_10_11(uint32_t p) { return normalize(unpack_normal_11_10_11_no_normalize(p)); }
// inc/mesh.hlsl:27-33 (vertex normal decode, different rounding constants)
static inline f3 unpack_unit_direction_11_10_11(uint32_t pck) {
    return f3{
        float(pck & ((1u << 11u) - 1u)) * (2.0f / float((1u << 11u) - 1u)) - 1.0f,
        float((pck >> 11u) & ((1u << 10u) - 1u)) * (2.0f / float((1u << 10u) - 1u)) - 1.0f,
        float((pck >> 21u)) * (2.0f / float((1u << 11u) - 1u)) - 1.0f};
}
static inline uint32_t pack_color_888(f3 c) {
    c = vsqrt(c);
    uint32_t p = 0;
    p += pack_unorm(c.x, 8);
    p += pack_unorm(c.y, 8) << 8;
    p += pack_unorm(c.z, 8) << 16;
    return p;
}
static inline f3 unpack_color_888(uint32_t p) {
    f3 c{unpack_unorm(p, 8), unpack_unorm(p >> 8, 8), unpack_unorm(p >> 16, 8)};
    return c * c;
}
static inline uint32_t pack_2x16f_uint(float a, float b) {
    return uint32_t(f32_to_f16(a)) | (uint32_t(f32_to_f16(b)) << 16u);
}
static inline f2 unpack_2x16f_uint(uint32_t u) {
    return f2{f16_to_f32(uint16_t(u & 0xffff)), f16_to_f32(uint16_t((u >> 16) & 0xffff))};
}
// rgb9e5 (inc/pack_unpack.hlsl:99-162)
static inline int floor_log2(float x) { return int((asuint(x) & 0x7F800000u) >> 23) - 127; }
static inline uint32_t float3_to_rgb9e5(f3 rgb) {
    const float MAX_RGB9E5 = (511.0f / 512.0f) * 65536.0f;
    float rc = clampf(rgb.x, 0.0f, MAX_RGB9E5);
    float gc = clampf(rgb.y, 0.0f, MAX_RGB9E5);
    float bc = clampf(rgb.z, 0.0f, MAX_RGB9E5);
    float maxrgb = fmaxf(rc, fmaxf(gc, bc));
    int exp_shared = std::max(-15 - 1, floor_log2(maxrgb)) + 1 + 15;
    float denom = exp2f(float(exp_shared - 15 - 9));
    int maxm = int(floorf(maxrgb / denom + 0.5f));
    if (maxm == 511 + 1) { denom *= 2.0f; exp_shared += 1; }
    int rm = int(floorf(rc / denom + 0.5f));
    int gm = int(floorf(gc / denom + 0.5f));
    int bm = int(floorf(bc / denom + 0.5f));
    return (uint32_t(rm) << 23) | (uint32_t(gm) << 14) | (uint32_t(bm) << 5) | uint32_t(exp_shared);
}
static inline f3 rgb9e5_to_float3(uint32_t v) {
    int exponent = int(v & 31u) - 15 - 9;
    float scale = exp2f(float(exponent));
    return f3{float((v >> 23) & 511u) * scale, float((v >> 14) & 511u) * scale, float((v >> 5) & 511u) * scale};
}

// ------------------------------------------------------------------ typed-format stores/loads
// (Vulkan fixed-function conversions: round-to-nearest-even)
static inline int8_t to_snorm8(float v) { return int8_t(rintf(clampf(v, -1.0f, 1.0f) * 127.0f)); }
static inline float from_snorm8(int8_t v) { return fmaxf(float(v) / 127.0f, -1.0f); }
static inline uint8_t to_unorm8(float v) { return uint8_t(rintf(clampf(v, 0.0f, 1.0f) * 255.0f)); }
static inline float from_unorm8(uint8_t v) { return float(v) / 255.0f; }
static inline int16_t to_snorm16(float v) { return int16_t(rintf(clampf(v, -1.0f, 1.0f) * 32767.0f)); }
static inline float from_snorm16(int16_t v) { return fmaxf(float(v) / 32767.0f, -1.0f); }
static inline uint32_t pack_a2r10g10b10(f3 rgb) {
    uint32_t r = uint32_t(rintf(clampf(rgb.x, 0.0f, 1.0f) * 1023.0f));
    uint32_t g = uint32_t(rintf(clampf(rgb.y, 0.0f, 1.0f) * 1023.0f));
    uint32_t b = uint32_t(rintf(clampf(rgb.z, 0.0f, 1.0f) * 1023.0f));
    return (r << 20) | (g << 10) | b;
}
static inline f3 unpack_a2r10g10b10(uint32_t p) {
    return f3{float((p >> 20) & 1023u) / 1023.0f, float((p >> 10) & 1023u) / 1023.0f, float(p & 1023u) / 1023.0f};
}
static inline uint32_t pack_rgba8_snorm(f4 v) {
    return uint32_t(uint8_t(to_snorm8(v.x))) | (uint32_t(uint8_t(to_snorm8(v.y))) << 8) |
           (uint32_t(uint8_t(to_snorm8(v.z))) << 16) | (uint32_t(uint8_t(to_snorm8(v.w))) << 24);
}
static inline f4 unpack_rgba8_snorm(uint32_t p) {
    return f4{from_snorm8(int8_t(p & 0xff)), from_snorm8(int8_t((p >> 8) & 0xff)),
              from_snorm8(int8_t((p >> 16) & 0xff)), from_snorm8(int8_t((p >> 24) & 0xff))};
}
struct h4 { uint16_t x, y, z, w; };
static inline h4 pack_rgba16f(f4 v) { return h4{f32_to_f16(v.x), f32_to_f16(v.y), f32_to_f16(v.z), f32_to_f16(v.w)}; }
static inline f4 unpack_rgba16f(h4 v) { return f4{f16_to_f32(v.x), f16_to_f32(v.y), f16_to_f32(v.z), f16_to_f32(v.w)}; }

// ------------------------------------------------------------------ gbuffer (inc/gbuffer.hlsl:26-87)
struct GbufferData {
    f3 albedo{0, 0, 0};
    f3 emissive{0, 0, 0};
    f3 normal{0, 0, 0};
    float roughness = 0;
    float metalness = 0;
};
static inline u4 gbuffer_pack(const GbufferData& g) {
    u4 r;
    r.x = pack_color_888(g.albedo);
    r.y = pack_normal_11_10_11(g.normal);
    r.z = pack_2x16f_uint(sqrtf(g.roughness), g.metalness);
    r.w = float3_to_rgb9e5(g.emissive);
    return r;
}
static inline GbufferData gbuffer_unpack(u4 d) {
    GbufferData g;
    g.albedo = unpack_color_888(d.x);
    g.normal = unpack_normal_11_10_11(d.y);
    f2 rm = unpack_2x16f_uint(d.z);
    g.roughness = rm.x * rm.x;
    g.metalness = rm.y;
    g.emissive = rgb9e5_to_float3(d.w);
    return g;
}

// ------------------------------------------------------------------ colour (inc/color/*.hlsl, working_color_space.hlsl)
static inline float sRGB_to_luminance(f3 c) { return dot(c, f3{0.2126f, 0.7152f, 0.0722f}); }
static inline f3 sRGB_to_YCbCr(f3 c) {
    return f3{dot(f3{0.2126f, 0.7152f, 0.0722f}, c), dot(f3{-0.1146f, -0.3854f, 0.5f}, c), dot(f3{0.5f, -0.4542f, -0.0458f}, c)};
}
static inline f3 YCbCr_to_sRGB(f3 c) {
    return vmax(mk3(0.0f), f3{dot(f3{1.0f, 0.0f, 1.5748f}, c), dot(f3{1.0f, -0.1873f, -0.4681f}, c), dot(f3{1.0f, 1.8556f, 0.0f}, c)});
}
static inline f4 linear_rgb_to_crunched_luma_chroma(f4 v) {
    f3 y = sRGB_to_YCbCr(xyz(v));
    float k = sqrtf(y.x) / fmaxf(1e-8f, y.x);
    return mk4(y * k, v.w);
}
static inline f4 crunched_luma_chroma_to_linear_rgb(f4 v) {
    f3 c = xyz(v) * v.x;
    return mk4(YCbCr_to_sRGB(c), v.w);
}

} // namespace okj
