#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Lists every min / max / clamp call of the compiled reference shaders (oracle/_ref/gen, written by `make`) in which a signed and an unsigned integer
meet: HLSL unifies the two to uint, so a negative value on the signed side wraps -- the rule behind two of the reading differences DESIGN section 5 lists (the FFX shadow
filter's tap clamp, the irradiance cache's cell-coordinate clamp). Works on a scratch copy of hlsl_compat.hpp in which those three intrinsics instantiate a [[deprecated]]
marker when the signedness is mixed, and collects the compiler's notes. Needs the reference checkout (the generated sources include the shaders in place)."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "..", "_ref", "gen")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def instrumented_header(dst):
    s = open(os.path.join(HERE, "hlsl_compat.hpp")).read()
    helper = '''template <bool MIXED> struct mixed_sign { static inline void note() {} };
template <> struct mixed_sign<true> { [[deprecated("MIXED_SIGN")]] static inline void note() {} };
template <class X, class Y> struct is_mixed { static constexpr bool value = (std::is_same<X, int>::value && std::is_same<Y, unsigned>::value) || (std::is_same<X, unsigned>::value && std::is_same<Y, int>::value); };
'''
    s = s.replace("template <class A, class B> struct promote {", helper + "template <class A, class B> struct promote {", 1)
    two = "mixed_sign<is_mixed<typename VT<A>::elem, typename VT<B>::elem>::value>::note(); "
    n = 0
    for fn in ("min_", "max_"):
        old = "typedef typename promote<typename VT<A>::elem, typename VT<B>::elem>::type E; return map2(a, b, [](E x, E y) { return %s<E>(x, y); }); }" % fn
        n += s.count(old)
        s = s.replace(old, old.replace("return map2", two + "return map2"))
    old = "    return map3(x, lo, hi, [](E v, E a, E b) { return min_<E>(max_<E>(v, a), b); }); }"
    n += s.count(old)
    s = s.replace(old, "    mixed_sign<is_mixed<typename VT<A>::elem, typename VT<B>::elem>::value || is_mixed<typename VT<A>::elem, typename VT<C>::elem>::value || "
                       "is_mixed<typename VT<B>::elem, typename VT<C>::elem>::value>::note();\n" + old)
    assert n == 3, "hlsl_compat.hpp's min / max / clamp no longer look the way this script expects"
    open(os.path.join(dst, "hlsl_compat.hpp"), "w").write(s)


def sites(tmp, src):
    r = subprocess.run([CXX, "-O0", "-std=c++20", "-fsyntax-only", "-Wno-attributes", "-Wno-narrowing", "-Wdeprecated-declarations", "-I", tmp, "-I", HERE, "-I", GEN, src],
                       capture_output=True, text=True)
    out = set()
    blocks = r.stderr.split("warning: ")
    for b in blocks:
        if "MIXED_SIGN" not in b:
            continue
        for m in re.finditer(r"gen/(?:\w+/\.\./)*([\w/.]+\.hlsl):(\d+):\d+: note: in instantiation of function template specialization 'hlsl::(\w+)<([^']*)>' requested here", b):
            if m.group(3) in ("min", "max", "clamp"):
                out.add((os.path.normpath(m.group(1)), int(m.group(2)), m.group(3), m.group(4).replace("hlsl::", "").replace(", void", "")))
    return out


if __name__ == "__main__":
    srcs = [s for s in sorted(glob.glob(os.path.join(GEN, "pass_*.cpp"))) if "probes" not in s]
    assert srcs, "run `make -C oracle/ref_hlsl` first"
    with tempfile.TemporaryDirectory() as tmp:
        instrumented_header(tmp)
        with ThreadPoolExecutor(8) as ex:
            found = set().union(*ex.map(lambda s: sites(tmp, s), srcs))
    for f, line, fn, types in sorted(found):
        print(f"{f}:{line}: {fn}<{types}>")
    print(f"{len(found)} site(s) in {len(srcs)} passes", file=sys.stderr)
