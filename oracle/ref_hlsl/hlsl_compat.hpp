// TEST INFRASTRUCTURE (oracle/_ref). Never shipped, never linked or called by the product path (kajiya_amd/).
//
// A C++17 stand-in for the HLSL *language and intrinsics* -- nothing of kajiya's is restated here. With it, and the token-level
// rewriter next to it (hlsl2cpp.py), the reference's own shader text under /root/reference/assets/shaders compiles on the CPU in
// place: the build reads those files where they lie, writes rewritten copies only into oracle/_ref/gen/ (git-ignored) and links
// them into oracle/_ref/libref_hlsl.so. That library is what pins the hand-written oracle (oracle/okj_*.hpp) to the reference's
// text: tests/test_ref_hlsl.py runs both on the same inputs.
//
// What is here: vector types with swizzles (float2/3/4, int*, uint*, bool*), HLSL's implicit scalar<->vector and element-type
// conversions, column-major matrices with mul(), the intrinsics the in-scope shaders use, typed resource views over flat memory in
// the reference's texel formats (the same flat layouts the oracle and the product use), samplers, and a cooperative lane scheduler
// (ucontext) so that wave intrinsics and group barriers have lock-step meaning.
//
// Intrinsics whose result the HLSL / SPIR-V specs leave to the implementation are pinned to the definitions DESIGN.md §4 lists
// (and the oracle + product use): f32->f16 rounds to nearest even, lerp is x*(1-a)+y*a (GLSL.std.450 FMix), float->int conversion
// saturates, out-of-bounds loads return 0 and out-of-bounds stores are dropped, transcendental functions are libm's.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <algorithm>

namespace hlsl {

typedef uint32_t uint;
typedef float half;          // kajiya compiles without -enable-16bit-types: `half` is float
typedef float min16float;

// ------------------------------------------------------------------------------------------------ element types and promotion
template <class T> struct is_elem : std::false_type {};
template <> struct is_elem<bool> : std::true_type {};
template <> struct is_elem<int> : std::true_type {};
template <> struct is_elem<uint> : std::true_type {};
template <> struct is_elem<float> : std::true_type {};
template <> struct is_elem<double> : std::true_type {};          // an un-suffixed literal the rewriter missed; treated as float
template <> struct is_elem<long> : std::true_type {};
template <> struct is_elem<unsigned long> : std::true_type {};
template <> struct is_elem<short> : std::true_type {};
template <> struct is_elem<unsigned short> : std::true_type {};
template <> struct is_elem<char> : std::true_type {};
template <> struct is_elem<unsigned char> : std::true_type {};

template <class T> struct canon { typedef T type; };
template <> struct canon<double> { typedef float type; };
template <> struct canon<long> { typedef int type; };
template <> struct canon<unsigned long> { typedef uint type; };
template <> struct canon<short> { typedef int type; };
template <> struct canon<unsigned short> { typedef uint type; };
template <> struct canon<char> { typedef int type; };
template <> struct canon<unsigned char> { typedef uint type; };

// HLSL's usual arithmetic conversions: anything with float -> float; int with uint -> uint; bool -> int
template <class A, class B> struct promote {
    typedef typename canon<A>::type a; typedef typename canon<B>::type b;
    typedef typename std::conditional<std::is_same<a, float>::value || std::is_same<b, float>::value, float,
            typename std::conditional<std::is_same<a, uint>::value || std::is_same<b, uint>::value, uint, int>::type>::type type;
};

// element conversion. float -> int / uint saturates (v_cvt_i32_f32 / v_cvt_u32_f32; NaN -> 0): DESIGN.md §4
template <class To, class From> struct conv { static inline To f(From v) { return To(v); } };
template <> struct conv<int, float> { static inline int f(float v) {
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return int(v); } };
template <> struct conv<uint, float> { static inline uint f(float v) {
    if (!(v == v)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    if (v <= 0.0f) return 0u;
    return uint(v); } };
template <> struct conv<int, double> { static inline int f(double v) { return conv<int, float>::f(float(v)); } };
template <> struct conv<uint, double> { static inline uint f(double v) { return conv<uint, float>::f(float(v)); } };
template <> struct conv<bool, float> { static inline bool f(float v) { return v != 0.0f; } };
template <class To, class From> static inline To cv(From v) { return conv<To, From>::f(v); }

template <class T, int N> struct vec;
template <class T, int N, int... I> struct swz;

// ------------------------------------------------------------------------------------------------ "vector-like" trait
template <class A, class = void> struct VT { static constexpr bool ok = false; static constexpr bool isvec = false; static constexpr int n = 0; };
template <class A> struct VT<A, typename std::enable_if<is_elem<A>::value>::type> {
    static constexpr bool ok = true; static constexpr bool isvec = false; static constexpr int n = 1;
    typedef typename canon<A>::type elem;
    static inline elem get(const A& a, int) { return elem(a); }
};
template <class T, int N> struct VT<vec<T, N>, void> {
    static constexpr bool ok = true; static constexpr bool isvec = true; static constexpr int n = N;
    typedef T elem;
    static inline T get(const vec<T, N>& a, int i) { return a.d[i]; }
};
template <class T, int N, int... I> struct VT<swz<T, N, I...>, void> {
    static constexpr bool ok = true; static constexpr bool isvec = true; static constexpr int n = sizeof...(I);
    typedef T elem;
    static inline T get(const swz<T, N, I...>& a, int i) { return a.d[swz<T, N, I...>::idx(i)]; }
};
template <class E, int N> struct vres_ { typedef vec<E, N> type; };
template <class E> struct vres_<E, 1> { typedef E type; };
template <class E, int N> using vres = typename vres_<E, N>::type;
template <class E, int N> static inline E& relem(vec<E, N>& v, int i) { return v.d[i]; }
template <class E> static inline E& relem(E& v, int) { return v; }

#define HLSL_REQ(cond) class = typename std::enable_if<(cond)>::type
template <class A, class B> struct dim2 { static constexpr int n = VT<A>::n > VT<B>::n ? VT<A>::n : VT<B>::n;
    static_assert((VT<A>::n == n || VT<A>::n == 1) && (VT<B>::n == n || VT<B>::n == 1), "vector dimension mismatch"); };

// ------------------------------------------------------------------------------------------------ swizzle proxy
template <class T, int N, int... I> struct swz {
    T d[N];
    static constexpr int n = sizeof...(I);
    static inline int idx(int i) { constexpr int t[] = {I...}; return t[i]; }
    template <class B, HLSL_REQ(VT<B>::ok && (VT<B>::n == n || VT<B>::n == 1))> swz& operator=(const B& b) {
        T tmp[n]; for (int i = 0; i < n; ++i) tmp[i] = cv<T>(VT<B>::get(b, i));
        for (int i = 0; i < n; ++i) d[idx(i)] = tmp[i];
        return *this;
    }
    swz& operator=(const swz& o) { T tmp[n]; for (int i = 0; i < n; ++i) tmp[i] = o.d[idx(i)]; for (int i = 0; i < n; ++i) d[idx(i)] = tmp[i]; return *this; }
    T& operator[](int i) { return d[idx(i)]; }
    T operator[](int i) const { return d[idx(i)]; }
};

// component count of a constructor argument list, and flattening
template <class... A> struct ncomp;
template <> struct ncomp<> { static constexpr int n = 0; };
template <class A, class... R> struct ncomp<A, R...> { static constexpr int n = VT<A>::n + ncomp<R...>::n; };
template <class... A> struct all_ok;
template <> struct all_ok<> { static constexpr bool v = true; };
template <class A, class... R> struct all_ok<A, R...> { static constexpr bool v = VT<A>::ok && all_ok<R...>::v; };
template <class T> static inline void flatten_(T*, int) {}
template <class T, class A, class... R> static inline void flatten_(T* out, int at, const A& a, const R&... r) {
    for (int i = 0; i < VT<A>::n; ++i) out[at + i] = cv<T>(VT<A>::get(a, i));
    flatten_<T>(out, at + VT<A>::n, r...);
}

#define HLSL_VEC_COMMON(N) \
    vec() { for (int i = 0; i < N; ++i) d[i] = T(); } \
    vec(const vec& o) { for (int i = 0; i < N; ++i) d[i] = o.d[i]; } \
    vec& operator=(const vec& o) { for (int i = 0; i < N; ++i) d[i] = o.d[i]; return *this; } \
    template <class B, HLSL_REQ(VT<B>::ok && !VT<B>::isvec)> vec(const B& s) { for (int i = 0; i < N; ++i) d[i] = cv<T>(VT<B>::get(s, 0)); } \
    template <class B, HLSL_REQ(VT<B>::isvec && VT<B>::n == N), class = void> vec(const B& b) { T t[N]; for (int i = 0; i < N; ++i) t[i] = cv<T>(VT<B>::get(b, i)); for (int i = 0; i < N; ++i) d[i] = t[i]; } \
    template <class B, HLSL_REQ(VT<B>::isvec && (VT<B>::n > N)), class = void, class = void> explicit vec(const B& b) { for (int i = 0; i < N; ++i) d[i] = cv<T>(VT<B>::get(b, i)); } \
    template <class A0, class A1, class... R, HLSL_REQ((all_ok<A0, A1, R...>::v && ncomp<A0, A1, R...>::n == N))> vec(const A0& a0, const A1& a1, const R&... r) { flatten_<T>(d, 0, a0, a1, r...); } \
    template <class U, HLSL_REQ(is_elem<U>::value)> explicit vec(const U (&a)[N]) { for (int i = 0; i < N; ++i) d[i] = cv<T>(a[i]); }   /* float4(float[4]) */ \
    T& operator[](int i) { return d[i]; } \
    T operator[](int i) const { return d[i]; } \
    T& operator[](uint i) { return d[i]; } \
    T operator[](uint i) const { return d[i]; }

#include "hlsl_swizzles.inc"   // generated by hlsl2cpp.py --swizzles: HLSL_SWZ2 / HLSL_SWZ3 / HLSL_SWZ4 member lists

template <class T> struct vec<T, 2> {
    union { T d[2]; struct { T x, y; }; struct { T r, g; }; HLSL_SWZ2 };
    HLSL_VEC_COMMON(2)
};
template <class T> struct vec<T, 3> {
    union { T d[3]; struct { T x, y, z; }; struct { T r, g, b; }; HLSL_SWZ3 };
    HLSL_VEC_COMMON(3)
};
template <class T> struct vec<T, 4> {
    union { T d[4]; struct { T x, y, z, w; }; struct { T r, g, b, a; }; HLSL_SWZ4 };
    HLSL_VEC_COMMON(4)
};

typedef vec<float, 2> float2; typedef vec<float, 3> float3; typedef vec<float, 4> float4;
typedef vec<int, 2> int2; typedef vec<int, 3> int3; typedef vec<int, 4> int4;
typedef vec<uint, 2> uint2; typedef vec<uint, 3> uint3; typedef vec<uint, 4> uint4;
typedef vec<bool, 2> bool2; typedef vec<bool, 3> bool3; typedef vec<bool, 4> bool4;
typedef float2 half2; typedef float3 half3; typedef float4 half4;

// `x.xxx` / `1.0.xxx` / `(expr).xx`: the rewriter turns an all-x (all-r) swizzle into `->*_sw<N>()`, which has the same meaning for
// scalars and vectors and binds tighter than `*` (only unary operators bind tighter, and they commute with a broadcast)
template <int N> struct sw_tag {};
template <class A, int N, HLSL_REQ(VT<A>::ok)> static inline vec<typename VT<A>::elem, N> operator->*(const A& a, sw_tag<N>) {
    vec<typename VT<A>::elem, N> r; for (int i = 0; i < N; ++i) r.d[i] = VT<A>::get(a, 0); return r;
}
static const sw_tag<2> _sw2; static const sw_tag<3> _sw3; static const sw_tag<4> _sw4;
// `(T)0` on a struct type
template <class T> static inline T hlsl_zero() { T t; memset((void*)&t, 0, sizeof(T)); return t; }

// ------------------------------------------------------------------------------------------------ operators
static inline int idiv(int a, int b) { return b == 0 ? 0 : (b == -1 ? int(0u - uint(a)) : a / b); }
static inline uint idiv(uint a, uint b) { return b == 0 ? 0xffffffffu : a / b; }
static inline float idiv(float a, float b) { return a / b; }
static inline int imod(int a, int b) { return (b == 0 || b == -1) ? 0 : a % b; }
static inline uint imod(uint a, uint b) { return b == 0 ? 0u : a % b; }
static inline float imod(float a, float b) { return std::fmod(a, b); }
static inline int ishl(int a, int b) { return int(uint(a) << (uint(b) & 31u)); }
static inline uint ishl(uint a, uint b) { return a << (b & 31u); }
static inline int ishr(int a, int b) { return a >> (uint(b) & 31u); }
static inline uint ishr(uint a, uint b) { return a >> (b & 31u); }

#define HLSL_BINOP(op, expr) \
    template <class A, class B, HLSL_REQ((VT<A>::isvec || VT<B>::isvec) && VT<A>::ok && VT<B>::ok)> \
    static inline auto operator op(const A& a_, const B& b_) { \
        typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; constexpr int N = dim2<A, B>::n; \
        vec<E, N> r_; for (int i = 0; i < N; ++i) { const E a = cv<E>(VT<A>::get(a_, i)), b = cv<E>(VT<B>::get(b_, i)); r_.d[i] = (expr); } return r_; }
HLSL_BINOP(+, a + b) HLSL_BINOP(-, a - b) HLSL_BINOP(*, a * b) HLSL_BINOP(/, idiv(a, b)) HLSL_BINOP(%, imod(a, b))
#define HLSL_INTOP(op, expr) \
    template <class A, class B, HLSL_REQ((VT<A>::isvec || VT<B>::isvec) && VT<A>::ok && VT<B>::ok)> \
    static inline auto operator op(const A& a_, const B& b_) { \
        typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; constexpr int N = dim2<A, B>::n; \
        static_assert(!std::is_same<E, float>::value, "bitwise operator on float"); \
        vec<E, N> r_; for (int i = 0; i < N; ++i) { const E a = cv<E>(VT<A>::get(a_, i)), b = cv<E>(VT<B>::get(b_, i)); r_.d[i] = (expr); } return r_; }
HLSL_INTOP(&, a & b) HLSL_INTOP(|, a | b) HLSL_INTOP(^, a ^ b)
// shifts keep the LEFT operand's type
#define HLSL_SHIFT(op, fn) \
    template <class A, class B, HLSL_REQ((VT<A>::isvec || VT<B>::isvec) && VT<A>::ok && VT<B>::ok)> \
    static inline auto operator op(const A& a_, const B& b_) { \
        typedef typename promote<typename VT<A>::elem, int>::type E; constexpr int N = dim2<A, B>::n; \
        vec<E, N> r_; for (int i = 0; i < N; ++i) r_.d[i] = fn(cv<E>(VT<A>::get(a_, i)), cv<E>(VT<B>::get(b_, i))); return r_; }
HLSL_SHIFT(<<, ishl) HLSL_SHIFT(>>, ishr)
#define HLSL_CMPOP(op) \
    template <class A, class B, HLSL_REQ((VT<A>::isvec || VT<B>::isvec) && VT<A>::ok && VT<B>::ok)> \
    static inline auto operator op(const A& a_, const B& b_) { \
        typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; constexpr int N = dim2<A, B>::n; \
        vec<bool, N> r_; for (int i = 0; i < N; ++i) r_.d[i] = cv<E>(VT<A>::get(a_, i)) op cv<E>(VT<B>::get(b_, i)); return r_; }
HLSL_CMPOP(==) HLSL_CMPOP(!=) HLSL_CMPOP(<) HLSL_CMPOP(<=) HLSL_CMPOP(>) HLSL_CMPOP(>=)
template <class A, HLSL_REQ(VT<A>::isvec)> static inline auto operator-(const A& a) {
    typedef typename promote<typename VT<A>::elem, int>::type E; vec<E, VT<A>::n> r; for (int i = 0; i < VT<A>::n; ++i) r.d[i] = E(0) - cv<E>(VT<A>::get(a, i)); return r; }
template <class A, HLSL_REQ(VT<A>::isvec)> static inline auto operator+(const A& a) { return vec<typename VT<A>::elem, VT<A>::n>(a); }
template <class A, HLSL_REQ(VT<A>::isvec)> static inline auto operator~(const A& a) {
    typedef typename promote<typename VT<A>::elem, int>::type E; vec<E, VT<A>::n> r; for (int i = 0; i < VT<A>::n; ++i) r.d[i] = ~cv<E>(VT<A>::get(a, i)); return r; }
template <class A, HLSL_REQ(VT<A>::isvec)> static inline auto operator!(const A& a) {
    vec<bool, VT<A>::n> r; for (int i = 0; i < VT<A>::n; ++i) r.d[i] = !cv<bool>(VT<A>::get(a, i)); return r; }
#define HLSL_ASSIGNOP(op, bin) \
    template <class A, class B, HLSL_REQ(VT<A>::isvec && VT<B>::ok)> static inline A& operator op(A& a, const B& b) { a = (a bin b); return a; }
HLSL_ASSIGNOP(+=, +) HLSL_ASSIGNOP(-=, -) HLSL_ASSIGNOP(*=, *) HLSL_ASSIGNOP(/=, /) HLSL_ASSIGNOP(%=, %)
HLSL_ASSIGNOP(&=, &) HLSL_ASSIGNOP(|=, |) HLSL_ASSIGNOP(^=, ^) HLSL_ASSIGNOP(<<=, <<) HLSL_ASSIGNOP(>>=, >>)

// ------------------------------------------------------------------------------------------------ intrinsics
template <class R, class A, class F> static inline vres<R, VT<A>::n> map1(const A& a, F f) {
    vres<R, VT<A>::n> r; for (int i = 0; i < VT<A>::n; ++i) relem(r, i) = f(VT<A>::get(a, i)); return r; }
#define HLSL_FLOAT1(name, expr) \
    template <class A, HLSL_REQ(VT<A>::ok)> static inline vres<float, VT<A>::n> name(const A& a_) { return map1<float>(a_, [](typename VT<A>::elem v) { const float x = cv<float>(v); (void)x; return float(expr); }); }
HLSL_FLOAT1(sqrt, std::sqrt(x)) HLSL_FLOAT1(rsqrt, 1.0f / std::sqrt(x)) HLSL_FLOAT1(rcp, 1.0f / x)
HLSL_FLOAT1(exp, std::exp(x)) HLSL_FLOAT1(exp2, std::exp2(x)) HLSL_FLOAT1(log, std::log(x)) HLSL_FLOAT1(log2, std::log2(x)) HLSL_FLOAT1(log10, std::log10(x))
// sin / cos: range reduction of the hardware the reference ran on (v_sin_f32 / v_cos_f32 take revolutions: x * 1/2pi, fract) -- DESIGN.md §4;
// it matters for the ~500 rad spiral-tap angles of the denoiser kernels, whose int() taps a last-bit difference moves
static inline float hw_turns_(float x) { float t = x * 0.15915494309189535f; t = t - std::floor(t); return t * 6.28318530717958647692f; }
#ifdef HLSL_LIBM_SINCOS
HLSL_FLOAT1(sin, std::sin(x)) HLSL_FLOAT1(cos, std::cos(x))
#else
HLSL_FLOAT1(sin, std::sin(hw_turns_(x))) HLSL_FLOAT1(cos, std::cos(hw_turns_(x)))
#endif
HLSL_FLOAT1(tan, std::tan(x))
HLSL_FLOAT1(asin, std::asin(x)) HLSL_FLOAT1(acos, std::acos(x)) HLSL_FLOAT1(atan, std::atan(x))
HLSL_FLOAT1(sinh, std::sinh(x)) HLSL_FLOAT1(cosh, std::cosh(x)) HLSL_FLOAT1(tanh, std::tanh(x))
HLSL_FLOAT1(floor, std::floor(x)) HLSL_FLOAT1(ceil, std::ceil(x)) HLSL_FLOAT1(trunc, std::trunc(x)) HLSL_FLOAT1(round, std::nearbyint(x))
HLSL_FLOAT1(frac, x - std::floor(x)) HLSL_FLOAT1(saturate, std::fmin(std::fmax(x, 0.0f), 1.0f))
HLSL_FLOAT1(degrees, x * 57.295779513082320876f) HLSL_FLOAT1(radians, x * 0.017453292519943295769f)
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto abs(const A& a_) {
    typedef typename promote<typename VT<A>::elem, int>::type E;
    return map1<E>(a_, [](typename VT<A>::elem v) { const E x = cv<E>(v); return std::is_same<E, float>::value ? E(std::fabs(float(x))) : (x < E(0) ? E(E(0) - x) : x); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto sign(const A& a_) {
    return map1<int>(a_, [](typename VT<A>::elem v) { return (v > 0) - (v < 0); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto isnan(const A& a_) { return map1<bool>(a_, [](typename VT<A>::elem v) { const float x = cv<float>(v); return x != x; }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto isinf(const A& a_) { return map1<bool>(a_, [](typename VT<A>::elem v) { return bool(std::isinf(cv<float>(v))); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto isfinite(const A& a_) { return map1<bool>(a_, [](typename VT<A>::elem v) { return bool(std::isfinite(cv<float>(v))); }); }

static inline uint asuint_(float f) { uint u; memcpy(&u, &f, 4); return u; }
static inline uint asuint_(uint u) { return u; }
static inline uint asuint_(int u) { return uint(u); }
static inline float asfloat_(uint u) { float f; memcpy(&f, &u, 4); return f; }
static inline float asfloat_(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float asfloat_(float u) { return u; }
static inline int asint_(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline int asint_(uint u) { return int(u); }
static inline int asint_(int u) { return u; }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto asuint(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return asuint_(v); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto asint(const A& a_) { return map1<int>(a_, [](typename VT<A>::elem v) { return asint_(v); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto asfloat(const A& a_) { return map1<float>(a_, [](typename VT<A>::elem v) { return asfloat_(v); }); }

// f32 <-> f16, round to nearest even
static inline uint f32tof16_(float f) {
    const uint x = asuint_(f); const uint sgn = (x >> 16) & 0x8000u; const uint ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return sgn | 0x7e00u;                       // NaN
    if (ax >= 0x477ff000u) return sgn | 0x7c00u;                      // rounds to >= 65520: inf
    if (ax < 0x33000001u) return sgn;                                 // <= 2^-25: rounds to zero
    if (ax < 0x38800000u) {                                           // subnormal half
        const uint e = ax >> 23; const uint man = (ax & 0x7fffffu) | 0x800000u; const uint shift = 126u - e;   // 14..24
        uint h = man >> shift; const uint rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
        return sgn | h;
    }
    uint h = (ax - 0x38000000u) >> 13; const uint rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return sgn | h;
}
static inline float f16tof32_(uint h) {
    const uint sgn = (h & 0x8000u) << 16; const uint e = (h >> 10) & 31u; const uint m = h & 0x3ffu;
    if (e == 0) { if (m == 0) return asfloat_(sgn); const float v = float(m) * 5.9604644775390625e-8f; return (sgn ? -v : v); }
    if (e == 31) return asfloat_(sgn | 0x7f800000u | (m << 13));
    return asfloat_(sgn | ((e + 112u) << 23) | (m << 13));
}
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto f32tof16(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return f32tof16_(cv<float>(v)); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto f16tof32(const A& a_) { return map1<float>(a_, [](typename VT<A>::elem v) { return f16tof32_(cv<uint>(v)); }); }

static inline uint countbits_(uint v) { return uint(__builtin_popcount(v)); }
static inline uint firstbithigh_(uint v) { return v ? 31u - uint(__builtin_clz(v)) : 0xffffffffu; }
static inline uint firstbitlow_(uint v) { return v ? uint(__builtin_ctz(v)) : 0xffffffffu; }
static inline uint reversebits_(uint v) { uint r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto countbits(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return countbits_(cv<uint>(v)); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto firstbithigh(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return firstbithigh_(cv<uint>(v)); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto firstbitlow(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return firstbitlow_(cv<uint>(v)); }); }
template <class A, HLSL_REQ(VT<A>::ok)> static inline auto reversebits(const A& a_) { return map1<uint>(a_, [](typename VT<A>::elem v) { return reversebits_(cv<uint>(v)); }); }

template <class A, class B, class F> static inline auto map2(const A& a_, const B& b_, F f) {
    typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; constexpr int N = dim2<A, B>::n;
    vres<decltype(f(E(), E())), N> r; for (int i = 0; i < N; ++i) relem(r, i) = f(cv<E>(VT<A>::get(a_, i)), cv<E>(VT<B>::get(b_, i))); return r; }
template <class E> static inline E min_(E a, E b) { return b < a ? b : a; }
template <class E> static inline E max_(E a, E b) { return a < b ? b : a; }
template <> inline float min_<float>(float a, float b) { return std::fmin(a, b); }      // NaN-dropping, like v_min_f32 / the oracle's fminf
template <> inline float max_<float>(float a, float b) { return std::fmax(a, b); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto min(const A& a, const B& b) {
    typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; return map2(a, b, [](E x, E y) { return min_<E>(x, y); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto max(const A& a, const B& b) {
    typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; return map2(a, b, [](E x, E y) { return max_<E>(x, y); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto pow(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return float(std::pow(float(x), float(y))); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto fmod(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return float(std::fmod(float(x), float(y))); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto atan2(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return float(std::atan2(float(x), float(y))); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto step(const A& edge, const B& x) { return map2(edge, x, [](auto e, auto v) { return float(v) >= float(e) ? 1.0f : 0.0f; }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto ldexp(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return float(x) * std::exp2(float(y)); }); }

template <class A, class B, class C, class F> static inline auto map3(const A& a_, const B& b_, const C& c_, F f) {
    typedef typename promote<typename promote<typename VT<A>::elem, typename VT<B>::elem>::type, typename VT<C>::elem>::type E;
    constexpr int N0 = dim2<A, B>::n; constexpr int N = N0 > VT<C>::n ? N0 : VT<C>::n;
    vres<decltype(f(E(), E(), E())), N> r;
    for (int i = 0; i < N; ++i) relem(r, i) = f(cv<E>(VT<A>::get(a_, i)), cv<E>(VT<B>::get(b_, i)), cv<E>(VT<C>::get(c_, i))); return r; }
template <class A, class B, class C, HLSL_REQ(VT<A>::ok && VT<B>::ok && VT<C>::ok)> static inline auto clamp(const A& x, const B& lo, const C& hi) {
    typedef typename promote<typename promote<typename VT<A>::elem, typename VT<B>::elem>::type, typename VT<C>::elem>::type E;
    return map3(x, lo, hi, [](E v, E a, E b) { return min_<E>(max_<E>(v, a), b); }); }
template <class A, class B, class C, HLSL_REQ(VT<A>::ok && VT<B>::ok && VT<C>::ok)> static inline auto lerp(const A& x, const B& y, const C& s) {
    return map3(x, y, s, [](auto a, auto b, auto t) { return float(float(a) * (1.0f - float(t)) + float(b) * float(t)); }); }
template <class A, class B, class C, HLSL_REQ(VT<A>::ok && VT<B>::ok && VT<C>::ok)> static inline auto mad(const A& x, const B& y, const C& z) {
    return map3(x, y, z, [](auto a, auto b, auto c) { return a * b + c; }); }
template <class A, class B, class C, HLSL_REQ(VT<A>::ok && VT<B>::ok && VT<C>::ok)> static inline auto smoothstep(const A& e0, const B& e1, const C& x) {
    return map3(e0, e1, x, [](auto a_, auto b_, auto v_) { const float a = float(a_), b = float(b_), v = float(v_);
        const float t = std::fmin(std::fmax((v - a) / (b - a), 0.0f), 1.0f); return float(t * t * (3.0f - 2.0f * t)); }); }
// select(cond, a, b): HLSL 2021's component-wise ?:
template <class Cn, class A, class B, HLSL_REQ(VT<Cn>::ok && VT<A>::ok && VT<B>::ok)> static inline auto select(const Cn& c_, const A& a_, const B& b_) {
    typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E0;
    typedef typename std::conditional<std::is_same<typename VT<A>::elem, bool>::value && std::is_same<typename VT<B>::elem, bool>::value, bool, E0>::type E;
    constexpr int N0 = dim2<A, B>::n; constexpr int N = N0 > VT<Cn>::n ? N0 : VT<Cn>::n;
    vres<E, N> r; for (int i = 0; i < N; ++i) relem(r, i) = cv<bool>(VT<Cn>::get(c_, i)) ? cv<E>(VT<A>::get(a_, i)) : cv<E>(VT<B>::get(b_, i)); return r; }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto and_(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return bool(x) && bool(y); }); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto or_(const A& a, const B& b) { return map2(a, b, [](auto x, auto y) { return bool(x) || bool(y); }); }

template <class A, HLSL_REQ(VT<A>::ok)> static inline bool any(const A& a) { for (int i = 0; i < VT<A>::n; ++i) if (cv<bool>(VT<A>::get(a, i))) return true; return false; }
template <class A, HLSL_REQ(VT<A>::ok)> static inline bool all(const A& a) { for (int i = 0; i < VT<A>::n; ++i) if (!cv<bool>(VT<A>::get(a, i))) return false; return true; }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline auto dot(const A& a, const B& b) {
    typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; constexpr int N = dim2<A, B>::n;
    E s = cv<E>(VT<A>::get(a, 0)) * cv<E>(VT<B>::get(b, 0)); for (int i = 1; i < N; ++i) s = s + cv<E>(VT<A>::get(a, i)) * cv<E>(VT<B>::get(b, i)); return s; }
template <class A, HLSL_REQ(VT<A>::ok)> static inline float length(const A& a) { return std::sqrt(float(dot(a, a))); }
template <class A, class B, HLSL_REQ(VT<A>::ok && VT<B>::ok)> static inline float distance(const A& a, const B& b) { return length(a - b); }
template <class A, HLSL_REQ(VT<A>::isvec)> static inline auto normalize(const A& a) { return vec<float, VT<A>::n>(a) / std::sqrt(float(dot(a, a))); }
static inline float normalize(float a) { return a / std::sqrt(a * a); }
template <class A, class B, HLSL_REQ(VT<A>::isvec && VT<B>::isvec)> static inline float3 cross(const A& a_, const B& b_) {
    const float3 a(a_), b(b_); return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class A, class B, HLSL_REQ(VT<A>::isvec && VT<B>::isvec)> static inline auto reflect(const A& i, const B& n) { return i - 2.0f * dot(n, i) * n; }
template <class A, class O1, class O2> static inline void sincos(const A& a, O1& s, O2& c) { s = sin(a); c = cos(a); }
template <class A, class O> static inline auto modf(const A& a, O& ip) { auto t = trunc(a); ip = t; return a - t; }

// ------------------------------------------------------------------------------------------------ matrices
// Column-major storage (m[c][r]), the layout DXC gives cbuffer matrices by default and glam::Mat4 has on the CPU side, so the
// reference's FrameConstants bytes can be copied in as they are. M[i] is ROW i; constructors take rows; mul(M, v) treats v as a column.
template <int R, int C> struct matrix;
template <int R, int C> struct mrow { matrix<R, C>* mm; int r;      // M[i] as an lvalue
    template <class B, HLSL_REQ(VT<B>::ok)> mrow& operator=(const B& b) { vec<float, C> v(b); for (int c = 0; c < C; ++c) mm->m[c][r] = v.d[c]; return *this; }
    mrow& operator=(const mrow& o) { for (int c = 0; c < C; ++c) mm->m[c][r] = o.mm->m[c][o.r]; return *this; }
    float& operator[](int c) { return mm->m[c][r]; } };
template <int R, int C> struct VT<mrow<R, C>, void> {
    static constexpr bool ok = true; static constexpr bool isvec = true; static constexpr int n = C; typedef float elem;
    static inline float get(const mrow<R, C>& a, int i) { return a.mm->m[i][a.r]; } };
template <int R, int C> struct matrix {
    float m[C][R];
    matrix() { for (int c = 0; c < C; ++c) for (int r = 0; r < R; ++r) m[c][r] = 0.0f; }
    template <class... A, HLSL_REQ((sizeof...(A) >= 2) && all_ok<A...>::v && ncomp<A...>::n == R * C)> matrix(const A&... a) {
        float t[R * C]; flatten_<float>(t, 0, a...); for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m[c][r] = t[r * C + c]; }
    template <int R2, int C2, HLSL_REQ(R2 >= R && C2 >= C && (R2 != R || C2 != C))> explicit matrix(const matrix<R2, C2>& o) { for (int c = 0; c < C; ++c) for (int r = 0; r < R; ++r) m[c][r] = o.m[c][r]; }
    mrow<R, C> operator[](int r) { return mrow<R, C>{this, r}; }
    vec<float, C> operator[](int r) const { vec<float, C> v; for (int c = 0; c < C; ++c) v.d[c] = m[c][r]; return v; }
    float& e(int r, int c) { return m[c][r]; }
    float e(int r, int c) const { return m[c][r]; }
    vec<float, 2> msw2(int r0, int c0, int r1, int c1) const { return vec<float, 2>(m[c0][r0], m[c1][r1]); }          // M._m00_m11
    vec<float, 3> msw3(int r0, int c0, int r1, int c1, int r2, int c2) const { return vec<float, 3>(m[c0][r0], m[c1][r1], m[c2][r2]); }
    vec<float, 4> msw4(int r0, int c0, int r1, int c1, int r2, int c2, int r3, int c3) const { return vec<float, 4>(m[c0][r0], m[c1][r1], m[c2][r2], m[c3][r3]); }
};
typedef matrix<2, 2> float2x2; typedef matrix<3, 3> float3x3; typedef matrix<4, 4> float4x4; typedef matrix<3, 4> float3x4; typedef matrix<4, 3> float4x3;
template <int R, int C, class V, HLSL_REQ(VT<V>::isvec && VT<V>::n == C)> static inline vec<float, R> mul(const matrix<R, C>& M, const V& v_) {
    const vec<float, C> v(v_); vec<float, R> r;
    for (int i = 0; i < R; ++i) { float s = M.m[0][i] * v.d[0]; for (int c = 1; c < C; ++c) s = s + M.m[c][i] * v.d[c]; r.d[i] = s; } return r; }
template <int R, int C, class V, HLSL_REQ(VT<V>::isvec && VT<V>::n == R), class = void> static inline vec<float, C> mul(const V& v_, const matrix<R, C>& M) {
    const vec<float, R> v(v_); vec<float, C> r;
    for (int c = 0; c < C; ++c) { float s = v.d[0] * M.m[c][0]; for (int i = 1; i < R; ++i) s = s + v.d[i] * M.m[c][i]; r.d[c] = s; } return r; }
template <int R, int K, int C> static inline matrix<R, C> mul(const matrix<R, K>& A, const matrix<K, C>& B) {
    matrix<R, C> r; for (int i = 0; i < R; ++i) for (int c = 0; c < C; ++c) { float s = A.m[0][i] * B.m[c][0]; for (int k = 1; k < K; ++k) s = s + A.m[k][i] * B.m[c][k]; r.m[c][i] = s; } return r; }
template <int R, int C> static inline matrix<R, C> mul(const matrix<R, C>& A, float s) { matrix<R, C> r; for (int c = 0; c < C; ++c) for (int i = 0; i < R; ++i) r.m[c][i] = A.m[c][i] * s; return r; }
template <int R, int C> static inline matrix<C, R> transpose(const matrix<R, C>& A) { matrix<C, R> r; for (int c = 0; c < C; ++c) for (int i = 0; i < R; ++i) r.m[i][c] = A.m[c][i]; return r; }
template <int R, int C> static inline matrix<R, C> operator*(const matrix<R, C>& A, float s) { return mul(A, s); }
template <int R, int C> static inline matrix<R, C> operator+(const matrix<R, C>& A, const matrix<R, C>& B) { matrix<R, C> r; for (int c = 0; c < C; ++c) for (int i = 0; i < R; ++i) r.m[c][i] = A.m[c][i] + B.m[c][i]; return r; }
static inline float determinant(const float3x3& M) {
    return M.e(0, 0) * (M.e(1, 1) * M.e(2, 2) - M.e(1, 2) * M.e(2, 1)) - M.e(0, 1) * (M.e(1, 0) * M.e(2, 2) - M.e(1, 2) * M.e(2, 0)) + M.e(0, 2) * (M.e(1, 0) * M.e(2, 1) - M.e(1, 1) * M.e(2, 0)); }

}  // namespace hlsl

#include "hlsl_resources.hpp"

// <math.h>'s double constants: the shaders define their own (float) ones
#undef M_PI
#undef M_E
#undef M_LOG2E
#undef M_LOG10E
#undef M_LN2
#undef M_LN10
#undef M_PI_2
#undef M_PI_4
#undef M_1_PI
#undef M_2_PI
#undef M_2_SQRTPI
#undef M_SQRT2
#undef M_SQRT1_2
