#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/_ref): token-level rewriter that makes the reference's HLSL text acceptable to a C++17 compiler
together with hlsl_compat.hpp. It reads /root/reference/assets/shaders/**.hlsl where they lie and writes rewritten copies ONLY
under the output directory (oracle/_ref/gen/, git-ignored): no reference source enters the repository.

The rewrite is lexical and meaning-preserving -- it does not know kajiya, only HLSL:
  * float literals get an `f` suffix (HLSL literals are float, C++ ones double); `1.0.xxx`, `x.xxx`, `(expr).xx` (broadcast swizzles,
    legal on scalars in HLSL) become `->*_swN`, an operator hlsl_compat.hpp defines for scalars and vectors alike
  * `[[vk::...]]`, `[numthreads]`, `[unroll]`, `[loop]` ... attributes are dropped (numthreads is recorded), `: SV_*` semantics too
  * `in` / `out` / `inout` parameters become values / references; `this.` becomes `this->`; `groupshared` becomes `static`
  * `cbuffer _ { ... }` wrappers are dropped: the members become globals, each followed by a registration object, as is every resource
    declaration (`Texture2D<float4> t;` -> `Texture2D<float4> t{ResName{"t", ...}};`) so that tests can bind memory by name
  * `const` is dropped except in `static const` (HLSL lets a const object call its non-const-qualified methods; C++ does not)
  * `M._43` -> `M.e(3, 2)`; `(Struct)0` -> `hlsl_zero<Struct>()`; `main` -> `cs_main`, its `SV_*` parameters recorded for the wrapper
  * a short, explicit list of per-file patches (PATCHES below) where C++ overload resolution needs a cast HLSL applies implicitly
"""
import argparse
import os
import re
import sys

TOKEN_RE = re.compile(r"""
    (?P<ws>[ \t\r\n]+|\\\n)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<num>0[xX][0-9a-fA-F]+[uUlL]*|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFhHuUlL]*)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>::|<<=|>>=|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|\+\+|--|->|\#\#|.)
""", re.S | re.X)

ATTRS = {"unroll", "loop", "branch", "flatten", "numthreads", "shader", "allow_uav_condition", "fastopt", "forcecase", "call", "earlydepthstencil", "noinline"}
RESOURCE_TYPES = {"Texture2D", "RWTexture2D", "Texture3D", "RWTexture3D", "TextureCube", "StructuredBuffer", "RWStructuredBuffer", "ByteAddressBuffer",
                  "RWByteAddressBuffer", "Buffer", "RWBuffer", "SamplerState", "SamplerComparisonState", "RaytracingAccelerationStructure", "Texture2DArray", "RWTexture2DArray"}
BUILTIN_TYPES = set("float int uint bool half double min16float".split()) | {t + str(n) for t in ("float", "int", "uint", "bool", "half") for n in (1, 2, 3, 4)} | \
    {"float%dx%d" % (r, c) for r in (2, 3, 4) for c in (2, 3, 4)}
BRACE_CTOR_TYPES = {t + str(n) for t in ("float", "int", "uint", "bool", "half") for n in (2, 3, 4)} | {"float%dx%d" % (r, c) for r in (2, 3, 4) for c in (2, 3, 4)}
SEMANTIC_RE = re.compile(r"^(SV_\w+|TEXCOORD\d*|POSITION\d*|COLOR\d*|NORMAL\d*|TANGENT\d*)$")
CPP_KEYWORD_IDS = {"and": "and_", "or": "or_", "not": "not_", "xor": "xor_", "new": "new_", "delete": "delete_", "register": "register_", "auto": "auto_", "union": "union_",
                   "export": "export_", "friend": "friend_", "mutable": "mutable_", "virtual": "virtual_", "explicit": "explicit_", "near": "near_", "far": "far_", "typeid": "typeid_"}
LANE_VALUE = {"SV_DispatchThreadID": "dispatch_thread_id", "SV_GroupThreadID": "group_thread_id", "SV_GroupID": "group_id", "SV_GroupIndex": "group_index"}
LOCKSTEP_IDS = re.compile(r"^(Wave[A-Z]\w*|Quad[A-Z]\w*|GroupMemoryBarrierWithGroupSync|AllMemoryBarrierWithGroupSync|DeviceMemoryBarrierWithGroupSync)$")

# (file relative to assets/shaders) -> [(regex on the REWRITTEN text, replacement, why)]. Each one only names a conversion HLSL performs implicitly.
PATCHES = {
    "lighting/spatial_reuse_lights.hlsl": [(r"output_tex\[orig_px\]\.rgb \+= out_color;", "{ float4 v_ = output_tex[orig_px]; v_.rgb += out_color; output_tex[orig_px] = v_; }",
                                            "a swizzled compound assignment to an image texel: load, modify, store")],
    "lighting/sample_lights.rgen.hlsl": [(r"float4\{select\(is_shadowed, 0, triangle_light\.radiance\(\), 1\)\}", "float4{select(is_shadowed, 0, triangle_light.radiance()), 1}",
                                          "as shipped the call reads select(c, 0, radiance, 1) inside float4(): one parenthesis late, not a valid select(); what it means is unambiguous")],
    "inc/lights/triangle.hlsl": [(r"res\.packed = p\.packed;", "for (int i_ = 0; i_ < 12; ++i_) res.packed[i_] = p.packed[i_];", "HLSL arrays are values and assign element-wise")],
}
# (file) -> text inserted before the file's closing include guard: forwarding overloads that spell out which conversion HLSL's overload
# resolution picks where C++ finds two user-defined conversions equally good
APPEND = {
    "inc/uv.hlsl": "float2 get_uv(uint2 pix, float4 texSize) { return get_uv(int2(pix), texSize); }   // HLSL: uint2 -> int2 (integral) beats uint2 -> float2\n",
}


class Tok:
    __slots__ = ("kind", "text", "bind")

    def __init__(self, kind, text):
        self.kind, self.text, self.bind = kind, text, None

    def sig(self):
        return self.kind not in ("ws", "lc", "bc")


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOKEN_RE.match(src, pos)
        kind = m.lastgroup
        text = m.group(kind)
        if kind == "num" and text.endswith(".") and pos + len(text) < len(src) and src[pos + len(text)] in "xyzwrgba":
            text = text[:-1]          # `1.xxx`: the dot belongs to the swizzle
        out.append(Tok(kind, text))
        pos += len(text)
    return out


def fix_number(t):
    s = t.text
    if s[:2] in ("0x", "0X"):
        return s
    is_float = "." in s or re.search(r"\d[eE][+-]?\d", s) is not None or s[-1] in "fFhH"
    if not is_float:
        return s.rstrip("lL")
    body = s.rstrip("fFhHlL")
    return body + "f"


class Rewriter:
    def __init__(self, src, relpath):
        self.relpath = relpath
        self.toks = tokenize(src)
        self.numthreads = None
        self.main_args = None
        self.lockstep = False

    # -- helpers over the token list (indices into self.toks)
    def nsig(self, i, step=1):
        i += step
        while 0 <= i < len(self.toks) and not self.toks[i].sig():
            i += step
        return i if 0 <= i < len(self.toks) else None

    def is_(self, i, text):
        return i is not None and self.toks[i].text == text

    def match_close(self, i, open_, close):
        depth = 0
        while i < len(self.toks):
            t = self.toks[i].text
            if self.toks[i].kind == "op":
                if t == open_:
                    depth += 1
                elif t == close:
                    depth -= 1
                    if depth == 0:
                        return i
            i += 1
        raise ValueError("unbalanced %s in %s" % (open_, self.relpath))

    def binding_before(self, i):
        """(binding, set) of the [[vk::binding]] attribute directly in front of token i, or ("-1", "-1")."""
        k = i - 1
        while k >= 0 and not self.toks[k].sig():
            if self.toks[k].bind:
                return self.toks[k].bind
            k -= 1
        return ("-1", "-1")

    def drop(self, a, b):
        for k in range(a, b + 1):
            if self.toks[k].kind != "ws" or "\n" not in self.toks[k].text:
                self.toks[k] = Tok("ws", "")
            # keep newlines so that compiler diagnostics still point at the reference's line numbers

    def run(self):
        T = self.toks
        # directive lines: leave `#include ...` untouched
        protected = set()
        i = 0
        line_start = True
        while i < len(T):
            t = T[i]
            if t.kind == "op" and t.text == "#" and line_start:
                j = self.nsig(i)
                if j is not None and T[j].text in ("include", "pragma"):
                    k = j
                    while k < len(T) and not (T[k].kind == "ws" and "\n" in T[k].text and not T[k].text.startswith("\\")):
                        protected.add(k)
                        k += 1
                    if T[j].text == "pragma":
                        self.drop(i, k - 1)
            if t.kind == "ws" and "\n" in t.text and not t.text.startswith("\\"):
                line_start = True
            elif t.sig():
                line_start = False
            i += 1

        # pass 1: literals, keywords-as-identifiers, lock-step detection
        for i, t in enumerate(T):
            if i in protected:
                continue
            if t.kind == "num":
                t.text = fix_number(t)
            elif t.kind == "id":
                if LOCKSTEP_IDS.match(t.text):
                    self.lockstep = True
                if t.text in CPP_KEYWORD_IDS:
                    t.text = CPP_KEYWORD_IDS[t.text]
                elif t.text == "groupshared":
                    t.text = "static"
                elif t.text in ("precise", "nointerpolation", "row_major", "column_major", "globallycoherent", "uniform", "unorm", "snorm"):
                    t.text = ""

        # pass 2: attributes
        i = 0
        while i < len(T):
            t = T[i]
            if i not in protected and t.kind == "op" and t.text == "[":
                j = self.nsig(i)
                if self.is_(j, "["):                       # [[vk::...]]
                    e = self.match_close(i, "[", "]")
                    m = re.match(r"\[\[vk::binding\((\w+)(?:,(\w+))?\)\]\]$", re.sub(r"\s+", "", "".join(x.text for x in T[i:e + 1])))
                    self.drop(i, e)
                    if m:
                        T[i].bind = (m.group(1), m.group(2) or "0")     # (binding, set); names when the declaration sits in a macro
                    i = e + 1
                    continue
                jn = self.nsig(j) if j is not None else None
                if j is not None and T[j].kind == "id" and T[j].text in ATTRS and jn is not None and T[jn].text in ("]", "("):
                    e = self.match_close(i, "[", "]")
                    if T[j].text == "numthreads":
                        self.numthreads = "".join(x.text for x in T[j + 1:e]).strip().strip("()")
                    self.drop(i, e)
                    i = e + 1
                    continue
            i += 1

        # pass 3: cbuffer wrappers, resource / constant registration (global scope only)
        depth = 0
        i = 0
        while i < len(T):
            t = T[i]
            if i in protected or not t.sig():
                i += 1
                continue
            if t.kind == "op" and t.text == "{":
                depth += 1
            elif t.kind == "op" and t.text == "}":
                depth -= 1
            elif t.kind == "id" and t.text == "cbuffer":
                cb_bind = self.binding_before(i)
                n = self.nsig(i)
                b = self.nsig(n) if T[n].kind == "id" else n
                if self.is_(b, ":"):                          # `cbuffer X : register(b0)`
                    while not self.is_(b, "{"):
                        b = self.nsig(b)
                e = self.match_close(b, "{", "}")
                # members: TYPE NAME [N]? ;
                k = self.nsig(b)
                while k is not None and k < e:
                    semi = k
                    while not self.is_(semi, ";"):
                        semi += 1
                    names = [x for x in range(k, semi) if T[x].kind == "id"]
                    name = None
                    for x in names:
                        nx = self.nsig(x)
                        if self.is_(nx, ";") or self.is_(nx, "["):
                            name = T[x].text
                    if name:
                        T[semi].text = '; static hlsl::ConstReg _creg_%s("%s", &%s, sizeof(%s), %s, %s);' % (name, name, name, name, cb_bind[0], cb_bind[1])
                    k = self.nsig(semi)
                self.drop(i, b)
                s = self.nsig(e)
                self.drop(e, s if self.is_(s, ";") else e)
                i = b + 1
                continue
            elif t.kind == "id" and (t.text in RESOURCE_TYPES or t.text == "ConstantBuffer"):
                # TYPE [<...>] NAME [ [..] ] ;   -- only declarations (a NAME then `;`), at any depth inside a macro body or depth 0
                res_bind = self.binding_before(i)
                j = self.nsig(i)
                type_end = i
                if self.is_(j, "<"):
                    type_end = self.match_close(j, "<", ">")
                    j = self.nsig(type_end)
                elif t.text in ("Texture2D", "RWTexture2D", "Texture3D", "TextureCube", "Texture2DArray"):
                    t.text = t.text + "<float4>"
                if j is not None and T[j].kind == "id" and depth == 0:
                    name = T[j].text
                    k = self.nsig(j)
                    if self.is_(k, "["):                     # arrays of resources (bindless tables): hlsl::ResourceArray, slots bound by index
                        ke = self.match_close(k, "[", "]")
                        type_text = re.sub(r"\s+", "", "".join(x.text for x in T[i:type_end + 1]))
                        self.drop(k, ke)
                        T[i].text = "hlsl::ResourceArray<" + T[i].text
                        T[type_end].text = T[type_end].text + ">"
                        T[j].text = '%s{hlsl::ResName{"%s", "%s[]", %s, %s}}' % (name, name, type_text, res_bind[0], res_bind[1])
                        i = self.nsig(ke)
                        continue
                    if self.is_(k, ";"):
                        type_text = "".join(x.text for x in T[i:type_end + 1]).replace('"', "").strip()
                        type_text = re.sub(r"\s+", "", type_text)
                        if t.text == "ConstantBuffer":
                            T[k].text = '; static hlsl::ConstReg _creg_%s("%s", &%s, sizeof(%s), %s, %s);' % (name, name, name, name, res_bind[0], res_bind[1])
                        elif t.text in ("SamplerState", "SamplerComparisonState"):
                            T[j].text = '%s{"%s"}' % (name, name)
                        else:
                            T[j].text = '%s{hlsl::ResName{"%s", "%s", %s, %s}}' % (name, name, type_text, res_bind[0], res_bind[1])
                        i = k
                        continue
            i += 1

        # pass 4: everything else, token by token
        paren_stack = []          # for each open paren: True when it opens a parameter list of a function DECLARATION
        i = 0
        while i < len(T):
            t = T[i]
            if i in protected or not t.sig():
                i += 1
                continue
            p = self.nsig(i, -1)
            n = self.nsig(i)
            if t.kind == "id" and t.text in BRACE_CTOR_TYPES and self.is_(n, "(") and not (p is not None and (T[p].text in (".", "->", "::") or T[p].kind == "id")):
                # `float2(a(rng), b(rng))`: DXC evaluates constructor arguments left to right and the shaders rely on it (two hash1_mut(rng)
                # draws in one constructor); a C++ braced-init-list guarantees that order, a parenthesised argument list does not
                e = self.match_close(n, "(", ")")
                t.text = "(" + t.text          # parenthesised as a whole: inside a macro argument the commas between braces would split it
                T[n].text = "{"
                T[e].text = "})"
            if t.kind == "id":
                if t.text == "const":
                    if not (p is not None and T[p].text == "static"):
                        t.text = ""
                elif t.text == "this":
                    if self.is_(n, "."):
                        T[n].text = "->"
                    elif not self.is_(n, "->"):
                        t.text = "(*this)"
                elif t.text in ("in", "out", "inout") and p is not None and T[p].text in ("(", ","):
                    # qualifier of a parameter: find the parameter's name = the identifier followed by , ) [ : =
                    k = n
                    name_at = None
                    angle = 0
                    while k is not None:
                        tx = T[k].text
                        if tx == "<":
                            angle += 1
                        elif tx == ">":
                            angle -= 1
                        elif angle == 0 and T[k].kind == "id":
                            nk = self.nsig(k)
                            if nk is not None and T[nk].text in (",", ")", "[", ":", "="):
                                name_at = k
                                break
                        k = self.nsig(k)
                    if name_at is not None and T[n].kind == "id":
                        if t.text != "in":
                            nk = self.nsig(name_at)
                            if self.is_(nk, "["):
                                T[name_at].text = "(&" + T[name_at].text + ")"
                            else:
                                T[name_at].text = "&" + T[name_at].text
                        t.text = ""
                elif t.text == "main" and self.is_(n, "(") and p is not None and T[p].text == "void":
                    t.text = "cs_main"
                    e = self.match_close(n, "(", ")")
                    self.main_args = self.parse_main(n, e)
                elif SEMANTIC_RE.match(t.text) and self.is_(p, ":"):
                    self.drop(p, i)
            elif t.kind == "op":
                if t.text == "." and n is not None and T[n].kind == "id":
                    name = T[n].text
                    m = re.match(r"^_([1-4])([1-4])$", name)
                    mm = re.match(r"^(?:_m[0-3][0-3]){1,4}$", name) or re.match(r"^(?:_[1-4][1-4]){2,4}$", name)
                    if m:
                        T[n].text = "e(%d, %d)" % (int(m.group(1)) - 1, int(m.group(2)) - 1)
                    elif mm:          # matrix swizzles: `_m00_m11` (zero-based) / `_11_22` (one-based) -> one element, or a vector of them
                        zero = name.startswith("_m")
                        rc = [(int(a) - (0 if zero else 1), int(b) - (0 if zero else 1)) for a, b in re.findall(r"_m?([0-4])([0-4])", name)]
                        T[n].text = ("e(%d, %d)" % rc[0]) if len(rc) == 1 else "msw%d(%s)" % (len(rc), ", ".join("%d, %d" % x for x in rc))
                    elif re.match(r"^(x{2,4}|r{2,4})$", name) and not (p is not None and T[p].text == "this"):
                        t.text = "->*"
                        T[n].text = "hlsl::_sw%d" % len(name)
                elif t.text == "(" and n is not None and T[n].kind == "id" and T[n].text not in BUILTIN_TYPES:
                    # `(Struct)0`
                    c = self.nsig(n)
                    z = self.nsig(c) if self.is_(c, ")") else None
                    if z is not None and T[z].kind == "num" and T[z].text in ("0", "0f", "0.0f", "0.f") and p is not None and T[p].text in ("=", "return", ",", "("):
                        after = self.nsig(z)
                        if after is not None and T[after].text in (";", ",", ")"):
                            ty = T[n].text
                            self.drop(i, z)
                            T[i].text = "hlsl::hlsl_zero<%s>()" % ty
            i += 1

        text = "".join(t.text for t in T)
        for rx, rep, _why in PATCHES.get(self.relpath, []):
            text, cnt = re.subn(rx, rep, text)
            if cnt == 0:
                sys.stderr.write("hlsl2cpp: patch did not apply in %s: %s\n" % (self.relpath, rx))
        if self.relpath in APPEND:
            k = text.rstrip().rfind("#endif")
            text = text[:k] + APPEND[self.relpath] + text[k:] if k >= 0 and text.rstrip().endswith(text[k:].rstrip()) else text + "\n" + APPEND[self.relpath]
        head = ""
        if self.lockstep:
            head += "#undef HLSL_LOCKSTEP\n#define HLSL_LOCKSTEP 1\n"
        tail = ""
        if self.numthreads:
            tail += "\n#define HLSL_NUMTHREADS %s\n" % self.numthreads
        if self.main_args is not None:
            tail += "#define HLSL_MAIN_ARGS(L) %s\n" % ", ".join(self.main_args)
        # `#line`-free: newlines are preserved, so diagnostics carry the reference's own line numbers (+ len(head) lines)
        return head + text + tail

    def parse_main(self, lp, rp):
        """Parameters of the entry point -> expressions over a LaneInfo L for the generated wrapper."""
        T = self.toks
        args, cur, depth = [], [], 0
        for k in range(lp + 1, rp):
            tx = T[k]
            if not tx.sig():
                continue
            if tx.text in ("(", "<", "["):
                depth += 1
            elif tx.text in (")", ">", "]"):
                depth -= 1
            if tx.text == "," and depth == 0:
                args.append(cur)
                cur = []
            else:
                cur.append(tx.text)
        if cur:
            args.append(cur)
        out = []
        for a in args:
            a = [x for x in a if x not in ("in", "const", "")]
            if ":" not in a:
                raise ValueError("entry point parameter without semantic in %s: %s" % (self.relpath, a))
            c = a.index(":")
            ty, sem = a[0], a[c + 1]
            if sem in ("SV_RayPayload", "SV_IntersectionAttributes"):
                return None          # a hit / miss shader: called by TraceRay (hlsl_resources.hpp), not dispatched
            if sem not in LANE_VALUE:
                raise ValueError("unsupported semantic %s in %s" % (sem, self.relpath))
            out.append("hlsl::lane_arg<%s>(L.%s)" % (ty, LANE_VALUE[sem]))
        return out


def swizzle_members():
    """HLSL_SWZ2/3/4: union members for every 2-, 3- and 4-component swizzle of a 2-, 3-, 4-vector, xyzw and rgba spellings."""
    out = []
    for n in (2, 3, 4):
        members = []
        for letters in ("xyzw", "rgba"):
            for ln in (2, 3, 4):
                def rec(prefix):
                    if len(prefix) == ln:
                        name = "".join(letters[k] for k in prefix)
                        members.append("swz<T, %d, %s> %s;" % (n, ", ".join(str(k) for k in prefix), name))
                        return
                    for k in range(n):
                        rec(prefix + [k])
                rec([])
        out.append("#define HLSL_SWZ%d %s" % (n, " ".join(members)))
    return "\n".join(out) + "\n"


RT_WRAPPER = """// generated by oracle/ref_hlsl/hlsl2cpp.py: ray-generation shader {rel} with its hit / miss shaders (text not copied into the repository)
#include "hlsl_compat.hpp"
namespace hlsl {{ namespace {{      // internal linkage
static hlsl::PassBegin _pass_begin("{name}");
#include "{rel}"
namespace chit_ns {{
#include "{chit}"
}}
namespace miss0_ns {{
#include "{miss0}"
}}
namespace miss1_ns {{
#include "{miss1}"
}}
static void _chit(void* payload, float bu, float bv) {{ chit_ns::RayHitAttrib a; a.bary = float2(bu, bv); chit_ns::cs_main(*(GbufferRayPayload*)payload, a); }}
static void _miss0(void* payload) {{ miss0_ns::cs_main(*(GbufferRayPayload*)payload); }}
static void _miss1(void* payload) {{ miss1_ns::cs_main(*(miss1_ns::ShadowPayload*)payload); }}
static const hlsl::RtPipeline _rt = {{_chit, {{_miss0, _miss1}}}};
static void _invoke(const hlsl::LaneInfo& L) {{ hlsl::hlsl_rt_pipeline() = &_rt; cs_main(); }}
static const hlsl::uint _nt[3] = {{1, 1, 1}};
static hlsl::PassEnd _pass_end(_nt, false, _invoke);
}} }}
"""

WRAPPER = """// generated by oracle/ref_hlsl/hlsl2cpp.py from the reference's {rel} (text not copied into the repository)
#include "hlsl_compat.hpp"
namespace hlsl {{ namespace {{      // internal linkage: every pass declares its own `input_tex`, `cs_main`, ...
static hlsl::PassBegin _pass_begin("{name}");
#include "{rel}"
#ifndef HLSL_LOCKSTEP
#define HLSL_LOCKSTEP 0
#endif
static void _invoke(const hlsl::LaneInfo& L) {{ cs_main(HLSL_MAIN_ARGS(L)); }}
static const hlsl::uint _nt[3] = {{HLSL_NUMTHREADS}};
static hlsl::PassEnd _pass_end(_nt, HLSL_LOCKSTEP != 0, _invoke);
}} }}
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shaders", default="/root/reference/assets/shaders")
    ap.add_argument("--out", required=True)
    ap.add_argument("--passes", nargs="*", default=[], help="entry files (relative to --shaders) to emit wrappers for")
    ap.add_argument("--rt-passes", nargs="*", default=[], help="rgen[:chit[:miss0[:miss1]]] (relative to --shaders)")
    ap.add_argument("--probes", default=None, help="a directory of OUR OWN test shaders (oracle/ref_hlsl/probes): rewritten into <out>/probes/, where `../inc/...` is the reference's header")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "hlsl_swizzles.inc"), "w") as f:
        f.write(swizzle_members())
    n = 0
    for root, _dirs, files in os.walk(args.shaders):
        for fn in files:
            if not fn.endswith(".hlsl"):
                continue
            src_path = os.path.join(root, fn)
            rel = os.path.relpath(src_path, args.shaders)
            try:
                text = Rewriter(open(src_path, encoding="utf-8", errors="replace").read(), rel).run()
            except Exception as e:   # a file outside the path that this lexer cannot handle is only a problem if something includes it
                text = '#error "hlsl2cpp could not rewrite %s: %s"\n' % (rel, str(e).replace('"', "'"))
            dst = os.path.join(args.out, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, "w") as f:
                f.write(text)
            n += 1
    if args.probes:
        for fn in sorted(os.listdir(args.probes)):
            if fn.endswith(".hlsl"):
                rel = os.path.join("probes", fn)
                os.makedirs(os.path.join(args.out, "probes"), exist_ok=True)
                with open(os.path.join(args.out, rel), "w") as f:
                    f.write(Rewriter(open(os.path.join(args.probes, fn), encoding="utf-8").read(), rel).run())
                n += 1
    for rel in args.passes:
        name = rel[:-5] if rel.endswith(".hlsl") else rel
        with open(os.path.join(args.out, "pass_" + name.replace("/", "_").replace(".", "_") + ".cpp"), "w") as f:
            f.write(WRAPPER.format(rel=rel, name=name))
    for spec in args.rt_passes:
        parts = spec.split(":")
        rel = parts[0]
        chit = parts[1] if len(parts) > 1 and parts[1] else "rt/gbuffer.rchit.hlsl"
        miss0 = parts[2] if len(parts) > 2 and parts[2] else "rt/gbuffer.rmiss.hlsl"
        miss1 = parts[3] if len(parts) > 3 and parts[3] else "rt/shadow.rmiss.hlsl"
        name = rel[:-5] if rel.endswith(".hlsl") else rel
        with open(os.path.join(args.out, "pass_" + name.replace("/", "_").replace(".", "_") + ".cpp"), "w") as f:
            f.write(RT_WRAPPER.format(rel=rel, name=name, chit=chit, miss0=miss0, miss1=miss1))
    print("hlsl2cpp: rewrote %d files into %s, %d pass wrappers" % (n, args.out, len(args.passes) + len(args.rt_passes)))


if __name__ == "__main__":
    main()
