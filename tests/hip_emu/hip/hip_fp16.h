// TEST INFRASTRUCTURE: see hip_runtime.h in this directory. IEEE binary16 <-> binary32, round to nearest even.
#pragma once
#include "hip_runtime.h"
struct __half { uint16_t bits; };
inline __half __ushort_as_half(uint16_t u) { return __half{u}; }
inline uint16_t __half_as_ushort(__half h) { return h.bits; }
inline __half __float2half_rn(float f) {
    const uint32_t x = __float_as_uint(f), sign = (x >> 16) & 0x8000u, absx = x & 0x7fffffffu;
    if (absx > 0x7f800000u) return __half{uint16_t(sign | 0x7e00u)};                 // NaN
    if (absx >= 0x477ff000u) return __half{uint16_t(sign | 0x7c00u)};                // rounds to inf (>= 65520)
    if (absx < 0x33000001u) return __half{uint16_t(sign)};                           // < 2^-25 (or exactly 2^-25: ties to even = 0)
    const int e = int(absx >> 23) - 127;
    uint32_t mant = (absx & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                                       // subnormal halves lose more bits
    const uint32_t half_ulp = 1u << (shift - 1), rest = mant & ((1u << shift) - 1u);
    uint32_t m = mant >> shift;
    if (rest > half_ulp || (rest == half_ulp && (m & 1u))) ++m;
    uint32_t h = e >= -14 ? (uint32_t(e + 15) << 10) + (m - 0x400u) : m;              // a mantissa carry bumps the exponent by itself
    return __half{uint16_t(sign | h)};
}
inline float __half2float(__half hh) {
    const uint32_t h = hh.bits, sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 31u) return __uint_as_float(sign | 0x7f800000u | (m << 13));
    if (e == 0u) {
        const float v = float(m) * 5.9604644775390625e-8f;                            // m * 2^-24
        return sign ? -v : v;
    }
    return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
