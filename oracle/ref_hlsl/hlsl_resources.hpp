// TEST INFRASTRUCTURE (oracle/_ref), second half of hlsl_compat.hpp: typed resource views over flat, row-major, tightly packed memory
// in the reference's texel formats (SURVEY.md App. A / DESIGN.md §2), samplers, atomics, and the lane-scheduler hooks behind the wave
// intrinsics and group barriers. Fixed-function behaviour (format conversion, filtering, out-of-bounds) follows DESIGN.md §4.
#pragma once
#include <vector>

namespace hlsl {

// texel formats (same numbering in tests/ref_hlsl.py)
enum Format {
    FMT_NONE = 0, FMT_R32F = 1, FMT_RG32F = 2, FMT_RGBA32F = 3, FMT_R16F = 4, FMT_RG16F = 5, FMT_RGBA16F = 6, FMT_R8_UNORM = 7, FMT_RGBA8_UNORM = 8,
    FMT_R8_SNORM = 9, FMT_RGBA8_SNORM = 10, FMT_RGBA16_SNORM = 11, FMT_R11G11B10F = 12, FMT_A2R10G10B10_UNORM = 13, FMT_R32UI = 14, FMT_RG32UI = 15,
    FMT_RGBA32UI = 16, FMT_RG16_SNORM = 17, FMT_R16_UNORM = 18, FMT_RG8_UNORM = 19
};
static inline int format_bytes(int f) {
    switch (f) {
        case FMT_R32F: case FMT_RG16F: case FMT_RGBA8_UNORM: case FMT_RGBA8_SNORM: case FMT_R11G11B10F: case FMT_A2R10G10B10_UNORM: case FMT_R32UI: case FMT_RG16_SNORM: return 4;
        case FMT_RG32F: case FMT_RGBA16F: case FMT_RGBA16_SNORM: case FMT_RG32UI: return 8;
        case FMT_RGBA32F: case FMT_RGBA32UI: return 16;
        case FMT_R16F: case FMT_R16_UNORM: case FMT_RG8_UNORM: return 2;
        case FMT_R8_UNORM: case FMT_R8_SNORM: return 1;
    }
    return 0;
}
static inline bool format_is_uint(int f) { return f == FMT_R32UI || f == FMT_RG32UI || f == FMT_RGBA32UI; }

static inline float unorm_to_f(uint v, uint maxv) { return float(v) / float(maxv); }
static inline uint f_to_unorm(float v, uint maxv) { if (!(v == v)) v = 0.0f; v = std::fmin(std::fmax(v, 0.0f), 1.0f); return uint(std::nearbyint(v * float(maxv))); }
static inline float snorm_to_f(int v, int maxv) { return std::fmax(float(v) / float(maxv), -1.0f); }
static inline int f_to_snorm(float v, int maxv) { if (!(v == v)) v = 0.0f; v = std::fmin(std::fmax(v, -1.0f), 1.0f); return int(std::nearbyint(v * float(maxv))); }
// unsigned small floats (B10G11R11_UFLOAT): the format conversion is the API's, not the shaders' text; this is the definition the oracle and the
// kernels share (okj::f32_to_ufloat): through fp16 (RTE), then the dropped mantissa bits round half up; finite values saturate at
// the largest finite one, +inf stays inf, negative / zero / NaN -> 0
static inline uint f_to_uf(float v, int mbits) { if (!(v > 0.0f)) return 0u; const uint h = f32tof16_(v) & 0x7fffu; const uint drop = 10 - mbits;
    const uint r = (h + (1u << (drop - 1))) >> drop; const uint max_finite = (30u << mbits) | ((1u << mbits) - 1u);
    if (h >= 0x7c00u) return 31u << mbits; return r > max_finite ? max_finite : r; }
static inline float uf_to_f(uint v, int mbits) { return f16tof32_(v << (10 - mbits)); }

// Texel <-> four 32-bit lanes. Float formats go through float4, integer formats through uint4; a float view of a 32-bit float
// format keeps the bit pattern (the packed G-buffer is an RGBA32F image read with asuint()).
struct Texel { uint u[4]; };
static inline Texel load_texel(const void* base, size_t index, int fmt) {
    Texel t{{0, 0, 0, 0}};
    const uint8_t* p = (const uint8_t*)base + index * size_t(format_bytes(fmt));
    auto F = [](float f) { return asuint_(f); };
    switch (fmt) {
        case FMT_R32F: case FMT_R32UI: memcpy(t.u, p, 4); if (fmt == FMT_R32F) t.u[3] = F(1.0f); break;
        case FMT_RG32F: case FMT_RG32UI: memcpy(t.u, p, 8); if (fmt == FMT_RG32F) t.u[3] = F(1.0f); break;
        case FMT_RGBA32F: case FMT_RGBA32UI: memcpy(t.u, p, 16); break;
        case FMT_R16F: { uint16_t h; memcpy(&h, p, 2); t.u[0] = F(f16tof32_(h)); t.u[3] = F(1.0f); break; }
        case FMT_RG16F: { uint16_t h[2]; memcpy(h, p, 4); t.u[0] = F(f16tof32_(h[0])); t.u[1] = F(f16tof32_(h[1])); t.u[3] = F(1.0f); break; }
        case FMT_RGBA16F: { uint16_t h[4]; memcpy(h, p, 8); for (int i = 0; i < 4; ++i) t.u[i] = F(f16tof32_(h[i])); break; }
        case FMT_R8_UNORM: t.u[0] = F(unorm_to_f(p[0], 255)); t.u[3] = F(1.0f); break;
        case FMT_RG8_UNORM: t.u[0] = F(unorm_to_f(p[0], 255)); t.u[1] = F(unorm_to_f(p[1], 255)); t.u[3] = F(1.0f); break;
        case FMT_RGBA8_UNORM: for (int i = 0; i < 4; ++i) t.u[i] = F(unorm_to_f(p[i], 255)); break;
        case FMT_R8_SNORM: t.u[0] = F(snorm_to_f((int8_t)p[0], 127)); t.u[3] = F(1.0f); break;
        case FMT_RGBA8_SNORM: for (int i = 0; i < 4; ++i) t.u[i] = F(snorm_to_f((int8_t)p[i], 127)); break;
        case FMT_RGBA16_SNORM: { int16_t s[4]; memcpy(s, p, 8); for (int i = 0; i < 4; ++i) t.u[i] = F(snorm_to_f(s[i], 32767)); break; }
        case FMT_RG16_SNORM: { int16_t s[2]; memcpy(s, p, 4); for (int i = 0; i < 2; ++i) t.u[i] = F(snorm_to_f(s[i], 32767)); t.u[3] = F(1.0f); break; }
        case FMT_R16_UNORM: { uint16_t s; memcpy(&s, p, 2); t.u[0] = F(unorm_to_f(s, 65535)); t.u[3] = F(1.0f); break; }
        case FMT_R11G11B10F: { uint v; memcpy(&v, p, 4); t.u[0] = F(uf_to_f(v & 0x7ffu, 6)); t.u[1] = F(uf_to_f((v >> 11) & 0x7ffu, 6)); t.u[2] = F(uf_to_f(v >> 22, 5)); t.u[3] = F(1.0f); break; }
        case FMT_A2R10G10B10_UNORM: { uint v; memcpy(&v, p, 4);   // VK_FORMAT_A2R10G10B10_UNORM_PACK32: B in bits 0-9, G 10-19, R 20-29, A 30-31
            t.u[0] = F(unorm_to_f((v >> 20) & 1023u, 1023)); t.u[1] = F(unorm_to_f((v >> 10) & 1023u, 1023)); t.u[2] = F(unorm_to_f(v & 1023u, 1023)); t.u[3] = F(unorm_to_f(v >> 30, 3)); break; }
        default: fprintf(stderr, "hlsl_compat: load from unbound / unknown format %d\n", fmt); abort();
    }
    return t;
}
static inline void store_texel(void* base, size_t index, int fmt, const Texel& t) {
    uint8_t* p = (uint8_t*)base + index * size_t(format_bytes(fmt));
    auto f = [&](int i) { return asfloat_(t.u[i]); };
    switch (fmt) {
        case FMT_R32F: case FMT_R32UI: memcpy(p, t.u, 4); break;
        case FMT_RG32F: case FMT_RG32UI: memcpy(p, t.u, 8); break;
        case FMT_RGBA32F: case FMT_RGBA32UI: memcpy(p, t.u, 16); break;
        case FMT_R16F: { uint16_t h = uint16_t(f32tof16_(f(0))); memcpy(p, &h, 2); break; }
        case FMT_RG16F: { uint16_t h[2] = {uint16_t(f32tof16_(f(0))), uint16_t(f32tof16_(f(1)))}; memcpy(p, h, 4); break; }
        case FMT_RGBA16F: { uint16_t h[4]; for (int i = 0; i < 4; ++i) h[i] = uint16_t(f32tof16_(f(i))); memcpy(p, h, 8); break; }
        case FMT_R8_UNORM: p[0] = uint8_t(f_to_unorm(f(0), 255)); break;
        case FMT_RG8_UNORM: p[0] = uint8_t(f_to_unorm(f(0), 255)); p[1] = uint8_t(f_to_unorm(f(1), 255)); break;
        case FMT_RGBA8_UNORM: for (int i = 0; i < 4; ++i) p[i] = uint8_t(f_to_unorm(f(i), 255)); break;
        case FMT_R8_SNORM: p[0] = uint8_t(int8_t(f_to_snorm(f(0), 127))); break;
        case FMT_RGBA8_SNORM: for (int i = 0; i < 4; ++i) p[i] = uint8_t(int8_t(f_to_snorm(f(i), 127))); break;
        case FMT_RGBA16_SNORM: { int16_t s[4]; for (int i = 0; i < 4; ++i) s[i] = int16_t(f_to_snorm(f(i), 32767)); memcpy(p, s, 8); break; }
        case FMT_RG16_SNORM: { int16_t s[2]; for (int i = 0; i < 2; ++i) s[i] = int16_t(f_to_snorm(f(i), 32767)); memcpy(p, s, 4); break; }
        case FMT_R16_UNORM: { uint16_t s = uint16_t(f_to_unorm(f(0), 65535)); memcpy(p, &s, 2); break; }
        case FMT_R11G11B10F: { const uint v = f_to_uf(f(0), 6) | (f_to_uf(f(1), 6) << 11) | (f_to_uf(f(2), 5) << 22); memcpy(p, &v, 4); break; }
        case FMT_A2R10G10B10_UNORM: { const uint v = (f_to_unorm(f(0), 1023) << 20) | (f_to_unorm(f(1), 1023) << 10) | f_to_unorm(f(2), 1023) | (f_to_unorm(f(3), 3) << 30); memcpy(p, &v, 4); break; }
        default: fprintf(stderr, "hlsl_compat: store to unbound / unknown format %d\n", fmt); abort();
    }
}

struct ResName { const char* name; const char* type; int binding; int set; };       // [[vk::binding(binding, set)]]; -1 when unknown
struct ResourceBase;
void hlsl_register_resource(const ResName& n, ResourceBase* r);       // ref_runtime.cpp: into the pass whose wrapper is being initialised
struct ResourceBase { void* data = nullptr; int w = 0, h = 0, depth = 1; int fmt = 0; size_t bytes = 0;
    ResourceBase() {}
    explicit ResourceBase(const ResName& n) { hlsl_register_resource(n, this); }
    virtual ~ResourceBase() {} };
#define HLSL_RES_CTORS(Type, Base) Type() {} explicit Type(const ResName& n) : Base(n) {}

// texel <-> shader type T (scalar or vec of float / uint / int)
template <class T> static inline T texel_to(const Texel& t) {
    typedef typename VT<T>::elem E; T r; for (int i = 0; i < VT<T>::n; ++i) relem(r, i) = std::is_same<E, float>::value ? E(asfloat_(t.u[i])) : E(t.u[i]); return r; }
template <class T, class V> static inline Texel texel_from(const V& v_) {
    typedef typename VT<T>::elem E; const T v(v_); Texel t{{0, 0, 0, 0}};
    for (int i = 0; i < VT<T>::n; ++i) { const E e = VT<T>::get(v, i); t.u[i] = std::is_same<E, float>::value ? asuint_(float(e)) : uint(e); } return t; }

struct SamplerState { bool linear = false, mip_linear = false; int address = 0; /* 0 clamp, 1 repeat, 2 mirror, 3 border */
    SamplerState() {}
    explicit SamplerState(const char* name) {          // kajiya names its samplers sampler_{l|n}{l|n}{c|r|mr|cb} (vulkan/shader.rs:208-230)
        const char* s = strrchr(name, '_'); s = s ? s + 1 : name;
        linear = s[0] == 'l'; mip_linear = s[0] && s[1] == 'l';
        const char* a = (s[0] && s[1]) ? s + 2 : "c";
        address = !strcmp(a, "r") ? 1 : !strcmp(a, "mr") ? 2 : !strcmp(a, "cb") ? 3 : 0; } };
typedef SamplerState SamplerComparisonState;

static inline int wrap_coord(int v, int n, int mode, bool& border) {
    border = false;
    if (mode == 1) { v %= n; return v < 0 ? v + n : v; }
    if (mode == 2) { const int p = 2 * n; v %= p; if (v < 0) v += p; return v < n ? v : p - 1 - v; }
    if (mode == 3) { if (v < 0 || v >= n) { border = true; return 0; } return v; }
    return v < 0 ? 0 : (v >= n ? n - 1 : v);
}

// what a Texture2D<float>[px] yields: a scalar that still answers to `.x` / `.r` (HLSL allows one-component swizzles of scalars)
template <class T> struct sc1 { union { T x; T r; }; sc1() : x() {} sc1(T v) : x(v) {} operator T() const { return x; } };
template <class T> struct VT<sc1<T>, void> { static constexpr bool ok = true; static constexpr bool isvec = false; static constexpr int n = 1; typedef T elem;
    static inline T get(const sc1<T>& a, int) { return a.x; } };
template <class T, bool IsVec = VT<T>::isvec> struct texel_value { typedef T type; };
template <class T> struct texel_value<T, false> { typedef sc1<T> type; };

template <class T> struct Texture2D : ResourceBase {
    typedef typename texel_value<T>::type TV;
    HLSL_RES_CTORS(Texture2D, ResourceBase)
    typedef vec<float, 4> F4;
    Texel fetch(int x, int y) const { if (x < 0 || y < 0 || x >= w || y >= h || !data) return Texel{{0, 0, 0, 0}}; return load_texel(data, size_t(y) * size_t(w) + size_t(x), fmt); }
    template <class P, HLSL_REQ(VT<P>::isvec && VT<P>::n == 2)> TV operator[](const P& p) const { return texel_to<T>(fetch(cv<int>(VT<P>::get(p, 0)), cv<int>(VT<P>::get(p, 1)))); }
    template <class P, HLSL_REQ(VT<P>::isvec && VT<P>::n == 3)> T Load(const P& p) const { return texel_to<T>(fetch(cv<int>(VT<P>::get(p, 0)), cv<int>(VT<P>::get(p, 1)))); }
    template <class P, class O, HLSL_REQ(VT<P>::isvec && VT<P>::n == 3 && VT<O>::isvec)> T Load(const P& p, const O& o) const { return texel_to<T>(fetch(cv<int>(VT<P>::get(p, 0)) + cv<int>(VT<O>::get(o, 0)), cv<int>(VT<P>::get(p, 1)) + cv<int>(VT<O>::get(o, 1)))); }
    F4 fetch_f4(int x, int y, const SamplerState& s) const {
        bool bx, by; x = wrap_coord(x, w, s.address, bx); y = wrap_coord(y, h, s.address, by);
        if (bx || by) return F4(0.0f);
        return texel_to<F4>(load_texel(data, size_t(y) * size_t(w) + size_t(x), fmt)); }
    // lod ignored: every texture bound through this view has one level
    template <class UV> T SampleLevel(const SamplerState& s, const UV& uv_, float, int2 offset = int2(0, 0)) const {
        const float2 uv(uv_);
        if (!s.linear) { const int x = cv<int>(std::floor(uv.x * float(w))) + offset.x, y = cv<int>(std::floor(uv.y * float(h))) + offset.y; return T(narrow(fetch_f4(x, y, s))); }
        const float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
        const float x0f = std::floor(fx), y0f = std::floor(fy); const float tx = fx - x0f, ty = fy - y0f;
        const int x0 = cv<int>(x0f) + offset.x, y0 = cv<int>(y0f) + offset.y;
        const F4 s00 = fetch_f4(x0, y0, s), s10 = fetch_f4(x0 + 1, y0, s), s01 = fetch_f4(x0, y0 + 1, s), s11 = fetch_f4(x0 + 1, y0 + 1, s);
        const F4 a = s00 * (1.0f - tx) + s10 * tx, b = s01 * (1.0f - tx) + s11 * tx;
        return T(narrow(F4(a * (1.0f - ty) + b * ty))); }
    template <class UV> T Sample(const SamplerState& s, const UV& uv) const { return SampleLevel(s, uv, 0.0f); }
    // Gather: (-,+), (+,+), (+,-), (-,-) = w z / x y of the 2x2 footprint around uv, channel c
    template <class UV> F4 gather_(const SamplerState& s, const UV& uv_, int c) const {
        const float2 uv(uv_); const float fx = uv.x * float(w) - 0.5f, fy = uv.y * float(h) - 0.5f;
        const int x0 = cv<int>(std::floor(fx)), y0 = cv<int>(std::floor(fy));
        return F4(fetch_f4(x0, y0 + 1, s).d[c], fetch_f4(x0 + 1, y0 + 1, s).d[c], fetch_f4(x0 + 1, y0, s).d[c], fetch_f4(x0, y0, s).d[c]); }
    template <class UV> F4 GatherRed(const SamplerState& s, const UV& uv) const { return gather_(s, uv, 0); }
    template <class UV> F4 GatherGreen(const SamplerState& s, const UV& uv) const { return gather_(s, uv, 1); }
    template <class UV> F4 GatherBlue(const SamplerState& s, const UV& uv) const { return gather_(s, uv, 2); }
    template <class UV> F4 GatherAlpha(const SamplerState& s, const UV& uv) const { return gather_(s, uv, 3); }
    template <class A, class B> void GetDimensions(A& ow, B& oh) const { ow = A(w); oh = B(h); }
    static vres<float, VT<T>::n> narrow(const F4& v) { vres<float, VT<T>::n> r; for (int i = 0; i < VT<T>::n; ++i) relem(r, i) = v.d[i]; return r; }
};

// what RWTexture2D<T>::operator[] returns: reads as T (loaded when formed), assignment stores through the format
template <class T, bool IsVec = VT<T>::isvec> struct RWTexel;
template <class T> struct RWTexel<T, true> : T {
    ResourceBase* res; int x, y;
    RWTexel(ResourceBase* r, int x_, int y_, const T& v) : T(v), res(r), x(x_), y(y_) {}
    void put(const T& v) { static_cast<T&>(*this) = v; if (x >= 0 && y >= 0 && x < res->w && y < res->h && res->data) store_texel(res->data, size_t(y) * size_t(res->w) + size_t(x), res->fmt, texel_from<T>(v)); }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator=(const B& b) { put(T(b)); return *this; }
    RWTexel& operator=(const RWTexel& b) { put(static_cast<const T&>(b)); return *this; }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator+=(const B& b) { put(T(static_cast<const T&>(*this) + b)); return *this; }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator*=(const B& b) { put(T(static_cast<const T&>(*this) * b)); return *this; }
};
template <class T> struct RWTexel<T, false> {
    ResourceBase* res; int x, y; T val;
    RWTexel(ResourceBase* r, int x_, int y_, const T& v) : res(r), x(x_), y(y_), val(v) {}
    operator T() const { return val; }
    void put(const T& v) { val = v; if (x >= 0 && y >= 0 && x < res->w && y < res->h && res->data) store_texel(res->data, size_t(y) * size_t(res->w) + size_t(x), res->fmt, texel_from<T>(v)); }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator=(const B& b) { put(cv<T>(VT<B>::get(b, 0))); return *this; }
    RWTexel& operator=(const RWTexel& b) { put(b.val); return *this; }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator+=(const B& b) { put(T(val + b)); return *this; }
    template <class B, HLSL_REQ(VT<B>::ok)> RWTexel& operator*=(const B& b) { put(T(val * b)); return *this; }
};
template <class T> struct VT<RWTexel<T, false>, void> {
    static constexpr bool ok = true; static constexpr bool isvec = false; static constexpr int n = 1; typedef typename VT<T>::elem elem;
    static inline elem get(const RWTexel<T, false>& a, int) { return a.val; } };
template <class T> struct VT<RWTexel<T, true>, void> : VT<T> {
    static inline typename VT<T>::elem get(const RWTexel<T, true>& a, int i) { return VT<T>::get(static_cast<const T&>(a), i); } };

template <class T> struct RWTexture2D : ResourceBase {
    HLSL_RES_CTORS(RWTexture2D, ResourceBase)
    Texel fetch(int x, int y) const { if (x < 0 || y < 0 || x >= w || y >= h || !data) return Texel{{0, 0, 0, 0}}; return load_texel(data, size_t(y) * size_t(w) + size_t(x), fmt); }
    template <class P, HLSL_REQ(VT<P>::isvec && VT<P>::n == 2)> RWTexel<T> operator[](const P& p) {
        const int x = cv<int>(VT<P>::get(p, 0)), y = cv<int>(VT<P>::get(p, 1)); return RWTexel<T>(this, x, y, texel_to<T>(fetch(x, y))); }
    template <class A, class B> void GetDimensions(A& ow, B& oh) const { ow = A(w); oh = B(h); }
};

// image arrays (the sky cubes as storage images): slices of w x h texels one after another; the slice count follows from the bound size
template <class T> struct RWTexture2DArray : ResourceBase {
    HLSL_RES_CTORS(RWTexture2DArray, ResourceBase)
    std::vector<RWTexture2D<T>> slices;
    template <class P, HLSL_REQ(VT<P>::isvec && VT<P>::n == 3)> RWTexel<T> operator[](const P& p) {
        const size_t slice_bytes = size_t(w) * size_t(h) * size_t(format_bytes(fmt)); const size_t n = slice_bytes ? bytes / slice_bytes : 0;
        if (slices.size() != n + 1) slices.resize(n + 1);                   // [n] stays unbound: out-of-range slices read 0 and drop stores
        const uint z = cv<uint>(VT<P>::get(p, 2)); RWTexture2D<T>& sl = slices[z < n ? z : n];
        if (z < n) { sl.data = (uint8_t*)data + size_t(z) * slice_bytes; sl.w = w; sl.h = h; sl.fmt = fmt; sl.bytes = slice_bytes; }
        return sl[int2(cv<int>(VT<P>::get(p, 0)), cv<int>(VT<P>::get(p, 1)))]; }
};

// cube maps: six w x w faces, face-major (+X -X +Y -Y +Z -Z), Vulkan face selection, bilinear inside the face with clamped texel coordinates
template <class T> struct TextureCube : ResourceBase {
    HLSL_RES_CTORS(TextureCube, ResourceBase)
    template <class D> T SampleLevel(const SamplerState& s, const D& dir_, float) const {
        const float3 d(dir_); const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
        int face; float sc, tc, ma;
        // Vulkan "Cube Map Face Selection": on ties rz wins over ry and rx, ry over rx
        if (az >= ax && az >= ay) { face = d.z >= 0 ? 4 : 5; sc = d.z >= 0 ? d.x : -d.x; tc = -d.y; ma = az; }
        else if (ay >= ax) { face = d.y >= 0 ? 2 : 3; sc = d.x; tc = d.y >= 0 ? d.z : -d.z; ma = ay; }
        else { face = d.x >= 0 ? 0 : 1; sc = d.x >= 0 ? -d.z : d.z; tc = -d.y; ma = ax; }
        const float u = 0.5f * (sc / ma + 1.0f), v = 0.5f * (tc / ma + 1.0f);
        Texture2D<T> f; f.data = (uint8_t*)data + size_t(face) * size_t(w) * size_t(w) * size_t(format_bytes(fmt)); f.w = w; f.h = w; f.fmt = fmt;
        SamplerState cl = s; cl.address = 0;
        return f.SampleLevel(cl, float2(u, v), 0.0f); }
};

// structured buffers: element references straight into memory; out-of-bounds reads give a zero element, writes go to a scratch one
template <class T> struct StructuredBuffer : ResourceBase {
    HLSL_RES_CTORS(StructuredBuffer, ResourceBase)
    const T& operator[](uint i) const { static thread_local T z; if (!data || size_t(i) >= bytes / sizeof(T)) { memset((void*)&z, 0, sizeof(T)); return z; } return ((const T*)data)[i]; }
    const T& Load(uint i) const { return (*this)[i]; }
};
template <class T> struct RWStructuredBuffer : ResourceBase {
    HLSL_RES_CTORS(RWStructuredBuffer, ResourceBase)
    T& operator[](uint i) { static thread_local T z; if (!data || size_t(i) >= bytes / sizeof(T)) { memset((void*)&z, 0, sizeof(T)); return z; } return ((T*)data)[i]; }
};
template <class T> using Buffer = StructuredBuffer<T>;
template <class T> using RWBuffer = RWStructuredBuffer<T>;
template <class T> using Texture3D = Texture2D<T>;          // declared by headers on the path, never sampled by an in-scope pass
template <class T> using RWTexture3D = RWTexture2D<T>;
struct RaytracingAccelerationStructure : ResourceBase { HLSL_RES_CTORS(RaytracingAccelerationStructure, ResourceBase) };

struct ByteAddressBuffer : ResourceBase {
    HLSL_RES_CTORS(ByteAddressBuffer, ResourceBase)
    uint ld(uint a) const { if (!data || size_t(a) + 4 > bytes) return 0u; uint v; memcpy(&v, (const uint8_t*)data + a, 4); return v; }
    uint Load(uint a) const { return ld(a); }
    template <class T> T Load(uint a) const { T t; memset((void*)&t, 0, sizeof(T)); if (data && size_t(a) + sizeof(T) <= bytes) memcpy((void*)&t, (const uint8_t*)data + a, sizeof(T)); return t; }
    uint2 Load2(uint a) const { return uint2(ld(a), ld(a + 4)); }
    uint3 Load3(uint a) const { return uint3(ld(a), ld(a + 4), ld(a + 8)); }
    uint4 Load4(uint a) const { return uint4(ld(a), ld(a + 4), ld(a + 8), ld(a + 12)); }
};
struct RWByteAddressBuffer : ByteAddressBuffer {
    HLSL_RES_CTORS(RWByteAddressBuffer, ByteAddressBuffer)
    void st(uint a, uint v) { if (!data || size_t(a) + 4 > bytes) return; memcpy((uint8_t*)data + a, &v, 4); }
    void Store(uint a, uint v) { st(a, v); }
    template <class V> void Store2(uint a, const V& v_) { const uint2 v(v_); st(a, v.x); st(a + 4, v.y); }
    template <class V> void Store3(uint a, const V& v_) { const uint3 v(v_); st(a, v.x); st(a + 4, v.y); st(a + 8, v.z); }
    template <class V> void Store4(uint a, const V& v_) { const uint4 v(v_); st(a, v.x); st(a + 4, v.y); st(a + 8, v.z); st(a + 12, v.w); }
    void InterlockedAdd(uint a, uint v, uint& orig) { orig = ld(a); st(a, orig + v); }
    void InterlockedAdd(uint a, uint v) { st(a, ld(a) + v); }
    void InterlockedMax(uint a, uint v, uint& orig) { orig = ld(a); st(a, orig > v ? orig : v); }
    void InterlockedMax(uint a, uint v) { const uint o = ld(a); st(a, o > v ? o : v); }
    void InterlockedOr(uint a, uint v, uint& orig) { orig = ld(a); st(a, orig | v); }
    void InterlockedOr(uint a, uint v) { st(a, ld(a) | v); }
    void InterlockedAnd(uint a, uint v, uint& orig) { orig = ld(a); st(a, orig & v); }
    void InterlockedAnd(uint a, uint v) { st(a, ld(a) & v); }
    void InterlockedMin(uint a, uint v, uint& orig) { orig = ld(a); st(a, orig < v ? orig : v); }
    void InterlockedMin(uint a, uint v) { const uint o = ld(a); st(a, o < v ? o : v); }
    void InterlockedExchange(uint a, uint v, uint& orig) { orig = ld(a); st(a, v); }
};
// lanes run one at a time (cooperative scheduler), so "atomics" are plain read-modify-writes
template <class D, class V, class O> static inline void InterlockedAdd(D& d, V v, O& orig) { orig = O(d); d = D(d + D(v)); }
template <class D, class V> static inline void InterlockedAdd(D& d, V v) { d = D(d + D(v)); }
template <class D, class V, class O> static inline void InterlockedMax(D& d, V v, O& orig) { orig = O(d); if (D(v) > d) d = D(v); }
template <class D, class V> static inline void InterlockedMax(D& d, V v) { if (D(v) > d) d = D(v); }
template <class D, class V, class O> static inline void InterlockedMin(D& d, V v, O& orig) { orig = O(d); if (D(v) < d) d = D(v); }
template <class D, class V> static inline void InterlockedMin(D& d, V v) { if (D(v) < d) d = D(v); }
template <class D, class V, class O> static inline void InterlockedOr(D& d, V v, O& orig) { orig = O(d); d = D(d | D(v)); }
template <class D, class V> static inline void InterlockedOr(D& d, V v) { d = D(d | D(v)); }
template <class D, class V, class O> static inline void InterlockedAnd(D& d, V v, O& orig) { orig = O(d); d = D(d & D(v)); }
template <class D, class V> static inline void InterlockedAnd(D& d, V v) { d = D(d & D(v)); }
template <class D, class V, class O> static inline void InterlockedExchange(D& d, V v, O& orig) { orig = O(d); d = D(v); }
template <class D, class C, class V, class O> static inline void InterlockedCompareExchange(D& d, C c, V v, O& orig) { orig = O(d); if (d == D(c)) d = D(v); }

// `Texture2D bindless_textures[];`: a table of views, slots bound by index (ref_bind_slot)
struct ResourceArrayBase : ResourceBase { ResourceArrayBase() {} explicit ResourceArrayBase(const ResName& n) : ResourceBase(n) {} virtual ResourceBase* slot(uint i) = 0; };
template <class T> struct ResourceArray : ResourceArrayBase {
    enum { SLOTS = 1024 }; T slots[SLOTS];
    HLSL_RES_CTORS(ResourceArray, ResourceArrayBase)
    T& operator[](uint i) { return slots[i < SLOTS ? i : 0]; }
    ResourceBase* slot(uint i) override { return i < SLOTS ? &slots[i] : nullptr; } };
static inline uint NonUniformResourceIndex(uint i) { return i; }
template <class T> struct ConstantBuffer : T {};
struct RayDesc { float3 Origin; float TMin; float3 Direction; float TMax; };

// ------------------------------------------------------------------------------------------------ lanes, waves, groups (runtime: ref_runtime.cpp)
struct LaneInfo { uint3 dispatch_thread_id, group_thread_id, group_id; uint group_index; };
const LaneInfo& hlsl_lane();
// All lanes of the calling lane's wave (64 consecutive group indices) that are still running publish `bytes` of `value` and meet; returns
// the wave's slots (stride HLSL_WAVE_SLOT bytes) and which lanes took part
enum { HLSL_WAVE_SLOT = 64, HLSL_WAVE = 64 };
struct WaveView { const uint8_t* slots; const uint8_t* active; uint lane; };
WaveView hlsl_wave_publish(const void* value, size_t bytes);
void hlsl_group_barrier();

static inline uint WaveGetLaneIndex() { return hlsl_lane().group_index % HLSL_WAVE; }
static inline uint WaveGetLaneCount() { return HLSL_WAVE; }
template <class T> static inline T WaveReadLaneAt(const T& v, uint lane) {
    static_assert(sizeof(T) <= HLSL_WAVE_SLOT, "wave slot too small"); const WaveView w = hlsl_wave_publish(&v, sizeof(T));
    T r; if (lane < HLSL_WAVE && w.active[lane]) memcpy((void*)&r, w.slots + size_t(lane) * HLSL_WAVE_SLOT, sizeof(T)); else memset((void*)&r, 0, sizeof(T)); return r; }
template <class T> static inline T WaveReadLaneFirst(const T& v) {
    const WaveView w = hlsl_wave_publish(&v, sizeof(T)); T r = v; for (uint l = 0; l < HLSL_WAVE; ++l) if (w.active[l]) { memcpy((void*)&r, w.slots + size_t(l) * HLSL_WAVE_SLOT, sizeof(T)); break; } return r; }
template <class T, class F> static inline T wave_reduce_(const T& v, F f) {
    const WaveView w = hlsl_wave_publish(&v, sizeof(T)); bool first = true; T acc = v;
    for (uint l = 0; l < HLSL_WAVE; ++l) if (w.active[l]) { T x; memcpy((void*)&x, w.slots + size_t(l) * HLSL_WAVE_SLOT, sizeof(T)); acc = first ? x : f(acc, x); first = false; } return acc; }
template <class T> static inline T WaveActiveSum(const T& v) { return wave_reduce_(v, [](const T& a, const T& b) { return T(a + b); }); }
template <class T> static inline T WaveActiveMin(const T& v) { return wave_reduce_(v, [](const T& a, const T& b) { return T(min(a, b)); }); }
template <class T> static inline T WaveActiveMax(const T& v) { return wave_reduce_(v, [](const T& a, const T& b) { return T(max(a, b)); }); }
template <class T> static inline T WaveActiveBitOr(const T& v) { return wave_reduce_(v, [](const T& a, const T& b) { return T(a | b); }); }
template <class T> static inline T WavePrefixSum(const T& v) {
    const WaveView w = hlsl_wave_publish(&v, sizeof(T)); T acc = T(0);
    for (uint l = 0; l < w.lane; ++l) if (w.active[l]) { T x; memcpy((void*)&x, w.slots + size_t(l) * HLSL_WAVE_SLOT, sizeof(T)); acc = T(acc + x); } return acc; }
static inline bool WaveActiveAnyTrue(bool b) { return wave_reduce_(uint(b), [](uint a, uint c) { return a | c; }) != 0; }
static inline bool WaveActiveAllTrue(bool b) { return wave_reduce_(uint(b), [](uint a, uint c) { return a & c; }) != 0; }
static inline bool WaveIsFirstLane() { const uint one = 1; const WaveView w = hlsl_wave_publish(&one, 4); for (uint l = 0; l < HLSL_WAVE; ++l) if (w.active[l]) return l == w.lane; return true; }
// quad intrinsics in a compute shader: a quad = four consecutive lanes; X flips bit 0, Y bit 1, diagonal both
template <class T> static inline T QuadReadAcrossX(const T& v) { return WaveReadLaneAt(v, WaveGetLaneIndex() ^ 1u); }
template <class T> static inline T QuadReadAcrossY(const T& v) { return WaveReadLaneAt(v, WaveGetLaneIndex() ^ 2u); }
template <class T> static inline T QuadReadAcrossDiagonal(const T& v) { return WaveReadLaneAt(v, WaveGetLaneIndex() ^ 3u); }
static inline void GroupMemoryBarrierWithGroupSync() { hlsl_group_barrier(); }
static inline void GroupMemoryBarrier() {}
static inline void AllMemoryBarrierWithGroupSync() { hlsl_group_barrier(); }
static inline void DeviceMemoryBarrierWithGroupSync() { hlsl_group_barrier(); }
static inline void DeviceMemoryBarrier() {}
static inline void AllMemoryBarrier() {}

// ------------------------------------------------------------------------------------------------ ray tracing
// TraceRay stands in for VK_KHR_ray_tracing_pipeline: the intersection query goes to a hook the test installs (the oracle's scene: the
// driver's traversal is a black box in the reference too); the closest-hit and miss shaders that then run are the
// reference's own text (rt/gbuffer.rchit.hlsl, rt/*.rmiss.hlsl), compiled into the same translation unit by the wrapper.
enum { RAY_FLAG_NONE = 0, RAY_FLAG_FORCE_OPAQUE = 1, RAY_FLAG_FORCE_NON_OPAQUE = 2, RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH = 4, RAY_FLAG_SKIP_CLOSEST_HIT_SHADER = 8,
       RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 0x10, RAY_FLAG_CULL_FRONT_FACING_TRIANGLES = 0x20, RAY_FLAG_CULL_OPAQUE = 0x40, RAY_FLAG_CULL_NON_OPAQUE = 0x80 };
struct RayHitInfo { int hit; float t, bary_u, bary_v; uint instance_index, instance_id, primitive_index; float object_to_world[12]; /* row-major 3x4 */ };
typedef void (*TraceHook)(void* user, const float* ray8 /* origin, tmin, direction, tmax */, uint flags, RayHitInfo* out);
struct RtPipeline { void (*closest_hit)(void* payload, float bary_u, float bary_v); void (*miss[2])(void* payload); };
const RtPipeline*& hlsl_rt_pipeline();
void hlsl_trace(const float* ray8, uint flags, RayHitInfo* out);
struct RtHitContext { RayDesc ray; RayHitInfo hit; };
RtHitContext*& hlsl_rt_hit();
static inline float3 WorldRayOrigin() { return hlsl_rt_hit()->ray.Origin; }
static inline float3 WorldRayDirection() { return hlsl_rt_hit()->ray.Direction; }
static inline float RayTCurrent() { return hlsl_rt_hit()->hit.t; }
static inline float RayTMin() { return hlsl_rt_hit()->ray.TMin; }
static inline uint InstanceID() { return hlsl_rt_hit()->hit.instance_id; }
static inline uint InstanceIndex() { return hlsl_rt_hit()->hit.instance_index; }
static inline uint PrimitiveIndex() { return hlsl_rt_hit()->hit.primitive_index; }
static inline float3x4 ObjectToWorld3x4() { const float* m = hlsl_rt_hit()->hit.object_to_world; return float3x4(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11]); }
static inline uint3 DispatchRaysIndex() { return hlsl_lane().dispatch_thread_id; }
uint3 hlsl_dispatch_dims();
static inline uint3 DispatchRaysDimensions() { return hlsl_dispatch_dims(); }
template <class P> static inline void TraceRay(const RaytracingAccelerationStructure&, uint flags, uint /*mask*/, uint /*sbt offset*/, uint /*sbt stride*/, uint miss_index, const RayDesc& ray, P& payload) {
    RtHitContext ctx; ctx.ray = ray;
    const float r8[8] = {ray.Origin.x, ray.Origin.y, ray.Origin.z, ray.TMin, ray.Direction.x, ray.Direction.y, ray.Direction.z, ray.TMax};
    hlsl_trace(r8, flags, &ctx.hit);
    const RtPipeline* pl = hlsl_rt_pipeline();
    if (ctx.hit.hit) {
        if (!(flags & RAY_FLAG_SKIP_CLOSEST_HIT_SHADER)) { RtHitContext* saved = hlsl_rt_hit(); hlsl_rt_hit() = &ctx; pl->closest_hit((void*)&payload, ctx.hit.bary_u, ctx.hit.bary_v); hlsl_rt_hit() = saved; }
    } else {
        pl->miss[miss_index < 2 ? miss_index : 1]((void*)&payload);
    }
}

// ------------------------------------------------------------------------------------------------ pass registry (filled by the generated wrappers)
// A wrapper is one translation unit: PassBegin, then the rewritten shader text (whose resource / constant declarations register
// themselves as they are constructed), then PassEnd. Static initialisation inside a translation unit runs in declaration order.
void hlsl_pass_begin(const char* name);
void hlsl_pass_end(const uint nt[3], bool lockstep, void (*invoke)(const LaneInfo&));
void hlsl_register_constant(const char* name, void* ptr, size_t bytes, int binding, int set);
struct PassBegin { explicit PassBegin(const char* name) { hlsl_pass_begin(name); } };
struct PassEnd { PassEnd(const uint nt[3], bool lockstep, void (*invoke)(const LaneInfo&)) { hlsl_pass_end(nt, lockstep, invoke); } };
struct ConstReg { ConstReg(const char* name, void* ptr, size_t bytes, int binding = -1, int set = -1) { hlsl_register_constant(name, ptr, bytes, binding, set); } };
template <class T> static inline T lane_arg(const uint3& v) { return T(v); }          // uint3 -> uint3 / (explicitly truncated) uint2 / int2 ...
template <> inline uint lane_arg<uint>(const uint3& v) { return v.x; }
template <> inline int lane_arg<int>(const uint3& v) { return int(v.x); }
template <class T> static inline T lane_arg(uint v) { return T(v); }

}  // namespace hlsl
