"""TEST INFRASTRUCTURE: loader for oracle/_ref/libref_hlsl.so -- the reference's own HLSL text compiled for the CPU (oracle/ref_hlsl/) --
and a host side for it that binds memory the way kajiya's SimpleRenderPass does: positionally, in `.read()` / `.write()` order, to the
`[[vk::binding(n)]]` numbers of descriptor set 0, the `.constants((...))` tuple into the cbuffer that follows them."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
LIB_PATH = os.path.join(REF_DIR, "libref_hlsl.so")
REFERENCE = "/root/reference"

# hlsl_resources.hpp: enum Format
FMT = dict(r32f=1, rg32f=2, rgba32f=3, r16f=4, rg16f=5, rgba16f=6, r8=7, rgba8=8, r8s=9, rgba8s=10, rgba16s=11, r11g11b10f=12, a2r10g10b10=13,
           r32ui=14, rg32ui=15, rgba32ui=16, rg16s=17, r16=18, rg8=19,
           reservoir=15, trp=16)   # parity.py's names for RG32UI reservoirs / the RGBA32UI packed temporal reservoir
FMT_BYTES = {1: 4, 2: 8, 3: 16, 4: 2, 5: 4, 6: 8, 7: 1, 8: 4, 9: 1, 10: 4, 11: 8, 12: 4, 13: 4, 14: 4, 15: 8, 16: 16, 17: 4, 18: 2, 19: 2}

_LIB = {}
_VARIANT = "hw"      # which build run_pass() uses: "hw" = sin / cos reduced in revolutions like v_sin_f32, "libm" = libm's
_HOOK = None


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden", "ref_hlsl")
_SCOPE = None       # the recording / replay scope of the running test (golden below)


def lib_available():
    return os.path.exists(LIB_PATH) or os.path.isdir(os.path.join(REFERENCE, "assets", "shaders"))


def replaying():
    """No compiled reference text in reach (no checkout, no prebuilt oracle/_ref/libref_hlsl.so) -- or KJ_REF_HLSL_REPLAY=1: run_pass() hands back what the
    reference's text wrote when scripts/make_ref_hlsl_golden.sh recorded it (tests/golden/ref_hlsl/*.npz), for the test cases that were recorded."""
    return bool(os.environ.get("KJ_REF_HLSL_REPLAY")) or not lib_available()


def available():
    return lib_available() or os.path.isdir(GOLDEN_DIR)


def require_live(why="this case drives the compiled reference text directly"):
    import pytest
    if replaying():
        pytest.skip(f"needs oracle/_ref/libref_hlsl.so ({why}); only recorded cases run without it")


class golden:
    """with golden("name"): ... -- a test case whose reference-side outputs are committed. Live (the library is there): runs the reference's text; with
    KJ_REF_GOLDEN_RECORD=1 also writes every image / buffer each pass WROTE to tests/golden/ref_hlsl/name.npz. Replaying: run_pass() fills the outputs from that file
    instead, so the same test -- the oracle against what the reference's text produced -- runs where neither the checkout nor the library exists."""

    def __init__(self, name):
        self.name = name
        self.path = os.path.join(GOLDEN_DIR, name + ".npz")

    def __enter__(self):
        global _SCOPE
        _SCOPE = {"name": self.name, "seq": 0, "rec": {}, "data": None}
        if replaying():
            if not os.path.exists(self.path):
                import pytest
                pytest.skip(f"no recorded outputs for {self.name}")
            _SCOPE["data"] = np.load(self.path)
        return self

    def __exit__(self, et, ev, tb):
        global _SCOPE
        sc, _SCOPE = _SCOPE, None
        if et is None and not replaying() and os.environ.get("KJ_REF_GOLDEN_RECORD") and sc["rec"]:
            os.makedirs(GOLDEN_DIR, exist_ok=True)
            np.savez_compressed(self.path, **sc["rec"])
        return False


def build():
    """Rebuild from the reference checkout when there is one (this container); elsewhere the prebuilt library is what there is."""
    if os.path.isdir(os.path.join(REFERENCE, "assets", "shaders")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle", "ref_hlsl")])


class sincos:
    """with sincos("libm"): ... -- the passes inside run from the build whose sin / cos are libm's (what the oracle uses outside the five
    spiral-tap sites of the rtdgi screen passes); the default build reduces the angle in revolutions like the hardware the reference ran on."""

    def __init__(self, variant):
        self.variant = variant

    def __enter__(self):
        global _VARIANT
        self.prev, _VARIANT = _VARIANT, self.variant

    def __exit__(self, *a):
        global _VARIANT
        _VARIANT = self.prev


def lib(variant=None):
    assert not replaying() or lib_available(), "replay mode has no library: call require_live() first"
    variant = variant or _VARIANT
    if variant not in _LIB:
        if not _LIB:
            build()
        L = C.CDLL(LIB_PATH if variant == "hw" else LIB_PATH.replace(".so", "_libm.so"))
        for f in ("ref_pass_name", "ref_pass_resource_name", "ref_pass_resource_type", "ref_pass_constant_name"):
            getattr(L, f).restype = C.c_char_p
        L.ref_bind.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ulonglong]
        L.ref_set_constant.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_ulonglong]
        L.ref_dispatch.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint]
        L.ref_bind_slot.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_set_trace_hook.argtypes = [C.c_void_p, C.c_void_p]
        _LIB[variant] = L
    return _LIB[variant]


def passes():
    L = lib()
    return [L.ref_pass_name(i).decode() for i in range(L.ref_pass_count())]


class Tex:
    """A flat row-major image in one of the reference's texel formats, over a numpy byte buffer."""

    def __init__(self, raw, w, h, fmt):
        self.fmt = FMT[fmt] if isinstance(fmt, str) else fmt
        self.raw = np.ascontiguousarray(raw).reshape(-1).view(np.uint8)
        self.w, self.h = int(w), int(h)
        assert self.raw.size == self.w * self.h * FMT_BYTES[self.fmt], (self.raw.size, w, h, fmt)

    @staticmethod
    def zeros(w, h, fmt):
        f = FMT[fmt] if isinstance(fmt, str) else fmt
        return Tex(np.zeros(int(w) * int(h) * FMT_BYTES[f], np.uint8), w, h, f)


class Buf:
    def __init__(self, raw):
        self.raw = np.ascontiguousarray(raw).reshape(-1).view(np.uint8)


def extent_inv_extent(w, h):
    """ImageDesc::extent_inv_extent_2d (kajiya-backend image.rs): [w, h, 1/w, 1/h] as f32"""
    return np.array([w, h, np.float32(1.0) / np.float32(w), np.float32(1.0) / np.float32(h)], np.float32)


BINDLESS = {}       # slot -> Tex: descriptor set 1's `bindless_textures[]` (inc/bindless_textures.hlsl: 0 BRDF-FG LUT, 1 blue noise, 2 Bezold-Brucke LUT)


def set_bindless(slot, tex):
    BINDLESS[slot] = tex


NAMED = {}          # name -> Tex / Buf: descriptor sets 1 and 2 (`meshes`, `vertices`, `bindless_texture_sizes`, `instance_dynamic_parameters_dyn`, `triangle_lights_dyn`)


def set_named(name, res):
    NAMED[name] = res


def set_trace_hook(fn_ptr, user):
    """TraceRay's intersection query (the driver's black box in the reference): a C function (user, ray8, flags, RayHitInfo*)."""
    global _HOOK
    _HOOK = (fn_ptr, user)


def run_pass(name, resources, constants, frame_constants, dispatch):
    """resources: Tex / Buf objects in .read()/.write() order; constants: numpy scalars / arrays in .constants((...)) order;
    dispatch: the thread extent given to .dispatch([x, y, z]) / .trace_rays(tlas, [x, y, z])."""
    if replaying():
        if _SCOPE is None or _SCOPE["data"] is None:
            import pytest
            pytest.skip("no reference checkout and no prebuilt oracle/_ref/libref_hlsl.so; this case has no recorded outputs")
        seq, data = _SCOPE["seq"], _SCOPE["data"]
        _SCOPE["seq"] += 1
        prefix = f"{seq:04d}|{name}|"
        keys = [k for k in data.files if k.startswith(prefix)]
        assert keys, f"recorded outputs of {_SCOPE['name']} have no call {prefix}: the test changed since scripts/make_ref_hlsl_golden.sh ran"
        for k in keys:
            r = resources[int(k.split("|")[2])]
            assert r.raw.size == data[k].size, (k, r.raw.size, data[k].size)
            r.raw[:] = data[k]
        return list(resources)
    L = lib()
    if _HOOK:
        L.ref_set_trace_hook(_HOOK[0], _HOOK[1])
    pn = name.encode()
    assert L.ref_pass_exists(pn), (name, passes())
    slots, types = [], {}
    for i in range(L.ref_pass_resource_count(pn)):
        if L.ref_pass_resource_set(pn, i) == 0:
            slots.append((L.ref_pass_resource_binding(pn, i), L.ref_pass_resource_name(pn, i)))
        elif L.ref_pass_resource_set(pn, i) == -1:       # no [[vk::binding]] (blur.hlsl): the compiler numbers them in declaration order
            slots.append((len(slots), L.ref_pass_resource_name(pn, i)))
        else:
            continue
        types[slots[-1][1]] = L.ref_pass_resource_type(pn, i)
    slots = sorted(set(slots))
    assert [b for b, _ in slots] == list(range(len(slots))), (name, slots)          # set 0 is dense from binding 0, like SimpleRenderPass binds it
    assert len(slots) == len(resources), (name, [n for _, n in slots], len(resources))
    keep = []
    for (_, rn), r in zip(slots, resources):
        if isinstance(r, Tex):
            rc = L.ref_bind(pn, rn, r.raw.ctypes.data, r.w, r.h, r.fmt, r.raw.size)      # size: an image array's slice count follows from it
        else:
            rc = L.ref_bind(pn, rn, r.raw.ctypes.data, 0, 0, 0, r.raw.size)
        assert rc == 0, (name, rn, rc)
        keep.append(r)
    # the cbuffer of set 0 sits right after the resources; members in declaration order = the order of the .constants() tuple
    cmembers = [i for i in range(L.ref_pass_constant_count(pn)) if L.ref_pass_constant_set(pn, i) == 0]
    if constants is None:       # a pass recorded without .constants(): whatever cbuffer the shader declares is unbound (and must be unused)
        constants, cmembers = [], []
    assert len(cmembers) == len(constants), (name, [L.ref_pass_constant_name(pn, i) for i in cmembers], len(constants))
    for i, v in zip(cmembers, constants):
        assert L.ref_pass_constant_binding(pn, i) == len(slots), (name, L.ref_pass_constant_name(pn, i))
        b = np.ascontiguousarray(v).reshape(-1).view(np.uint8)
        assert L.ref_set_constant(pn, L.ref_pass_constant_name(pn, i), b.ctypes.data, b.size) == 0, (name, L.ref_pass_constant_name(pn, i), b.size, L.ref_pass_constant_bytes(pn, i))
    if frame_constants is not None:
        names = [L.ref_pass_constant_name(pn, i) for i in range(L.ref_pass_constant_count(pn))]
        if b"frame_constants" in names:
            assert C.sizeof(frame_constants) == 1216
            assert L.ref_set_constant(pn, b"frame_constants", C.byref(frame_constants), 1216) == 0
    for i in range(L.ref_pass_resource_count(pn)):
        rn = L.ref_pass_resource_name(pn, i)
        if L.ref_pass_resource_set(pn, i) != 0 and rn.decode() in NAMED:
            r = NAMED[rn.decode()]
            if isinstance(r, Tex):
                assert L.ref_bind(pn, rn, r.raw.ctypes.data, r.w, r.h, r.fmt, 0) == 0
            else:
                assert L.ref_bind(pn, rn, r.raw.ctypes.data, 0, 0, 0, r.raw.size) == 0
        if L.ref_pass_resource_name(pn, i) == b"bindless_textures":
            for slot, t in BINDLESS.items():
                assert L.ref_bind_slot(pn, b"bindless_textures", slot, t.raw.ctypes.data, t.w, t.h, t.fmt) == 0
    d = list(dispatch) + [1] * (3 - len(dispatch))
    assert L.ref_dispatch(pn, int(d[0]), int(d[1]), int(d[2])) == 0
    if _SCOPE is not None:
        seq = _SCOPE["seq"]
        _SCOPE["seq"] += 1
        if os.environ.get("KJ_REF_GOLDEN_RECORD"):
            for idx, ((_, rn), r) in enumerate(zip(slots, resources)):
                if types[rn].startswith(b"RW"):
                    _SCOPE["rec"][f"{seq:04d}|{name}|{idx}"] = r.raw.copy()
    return keep
