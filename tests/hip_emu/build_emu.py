"""TEST INFRASTRUCTURE: builds the WHOLE product (every .hip and .cpp under kajiya_amd/csrc) against the CPU stand-in for HIP in this
directory -> tests/_build/emu_all/libkajiya_amd_emu.so, with AddressSanitizer + UBSan. The only edit made to the product source is the
rewrite of the dynamic-LDS declaration form (`extern __shared__ T name[];`, which has no host-C++ spelling) in a scratch copy; everything
else is compiled as it lies. Host compiler: ROCm's clang++ (the traversal code uses clang vector extensions). The result is never shipped
and never loaded by the product: tests load it explicitly (KJ_AMD_LIB / tests/hip_emu/cpu_as_cuda.py)."""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "kajiya_amd", "csrc")
# KJ_EMU_DEFINES="-DKJ_BVH_FOLD_INVD ...": build an experiment variant of the product (same switches as scripts/build_variant.sh) into its own directory
EXTRA = os.environ.get("KJ_EMU_DEFINES", "").split()
# the CPU stand-in always carries the measured-and-rejected forms of the ray passes too (rtdgi_ray_experiments.inc), so that
# test_ray_pass_forms_agree keeps holding them to the fused form although the product library no longer compiles them
ALWAYS = ["-DKJ_RAY_PASS_EXPERIMENTS"]
# KJ_HIP_EMU=fast: lanes as fibers, workgroups spread over the host cores, -O2, NO sanitizers (tests/hip_emu/hip/hip_runtime.h, fiber mode)
FAST = os.environ.get("KJ_HIP_EMU") == "fast"
OUT = os.path.join(ROOT, "tests", "_build", ("emu_fast" if FAST else "emu_all") + ("_" + re.sub(r"[^A-Za-z0-9]+", "_", "".join(EXTRA)) if EXTRA else ""))
SO = os.path.join(OUT, "libkajiya_amd_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-g", "-O1", "-std=c++20", "-fPIC", "-pthread", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
         # float -> int casts of NaN / out-of-range values are DEFINED on the GPU (v_cvt_i32_f32 saturates, NaN -> 0) and the kernels rely on that
         # where the reference's shaders do (e.g. a NaN direction reaching a cube lookup from an empty reservoir, clamped right after): not an error
         "-fno-sanitize=float-cast-overflow",
         "-I", os.path.join(ROOT, "tests", "hip_emu"), "-I", CSRC, "-D__HIP_PLATFORM_AMD__"] + ALWAYS + EXTRA
if FAST:
    FLAGS = ["-g", "-O2", "-std=c++20", "-fPIC", "-pthread", "-ffp-contract=off", "-DHIP_EMU_FIBERS", "-I", os.path.join(ROOT, "tests", "hip_emu"), "-I", CSRC, "-D__HIP_PLATFORM_AMD__"] + ALWAYS + EXTRA
DYNAMIC_LDS = re.compile(r"extern __shared__ ([A-Za-z0-9_]+) ([A-Za-z0-9_]+)\[\];")


def sanitizer_preload():
    """LD_PRELOAD value for a python process that loads the instrumented library."""
    rt = subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return rt


def build():
    """One build at a time: the ranks of a torch.distributed.run launch all come here, and a stale library must not be rebuilt by two of them into the same files."""
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked()


def _build_locked():
    deps = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(ROOT, "tests", "hip_emu", "hip", "*.h")) + glob.glob(os.path.join(ROOT, "tests", "hip_emu", "hipcub", "*.hpp")) + [os.path.join(ROOT, "include", "kajiya_amd.h"), os.path.abspath(__file__)]
    if os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(d) for d in deps):
        return SO

    def compile_one(src):
        name = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OUT, name + ".o")
        if src.endswith(".hip"):
            text = open(src).read()
            # kernel text kept in a .inc next to the .hip (rtdgi_ray_experiments.inc) is spliced in, so that its dynamic-LDS declarations get rewritten too
            text = re.sub(r'#include "(\w+\.inc)"', lambda m: open(os.path.join(CSRC, m.group(1))).read() if "experiments" in m.group(1) else m.group(0), text)
            text = DYNAMIC_LDS.sub(r"\1* \2 = (\1*)hip_emu::dynamic_lds();", text)
            src = os.path.join(OUT, name + ".emu.cpp")
            open(src, "w").write(text)
        subprocess.check_call([CLANG] + FLAGS + ["-x", "c++", "-c", src, "-o", obj])
        return obj
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    srcs = [s for s in srcs if not os.path.basename(s).startswith("_")]
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(compile_one, srcs))
    link = ["-shared"] if FAST else ["-shared", "-shared-libsan", "-fsanitize=address,undefined"]
    subprocess.check_call([CLANG] + link + ["-Wl,-Bsymbolic", "-pthread", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build())
