"""The `-m gpu` parity tests, unchanged, against the product source compiled for the CPU (tests/hip_emu) under AddressSanitizer + UBSan.
No GPU.

`KJ_HIP_EMU=1` makes tests/conftest.py's `gpu` fixture build tests/_build/emu_all/libkajiya_amd_emu.so (every .hip / .cpp of the product
against the stand-in headers) and map torch's CUDA tensors to host memory; the tests then call the very same kj_* entry points. What a
pass means: the kernel SOURCE and the host sequencing reproduce the oracle within the GPU tests' own tolerances, and no kernel reads or
writes a byte outside its buffers, overflows a signed integer or shifts out of range — on the inputs of those tests. What it does not
mean: anything about hipcc's code generation or the hardware (that is what the same tests do on an MI355X).

Two modes (tests/hip_emu/hip/hip_runtime.h): one host thread per lane, which the sanitizers understand — a SMALL slice runs that way here
(about a minute) — and lanes as fibers with workgroups spread over the cores, 30x faster, which runs nearly the whole GPU suite here.
`scripts/run_gpu_suite_on_cpu.sh [--sanitize]` runs either mode by hand.
Reading an exited lane in a wave exchange, divergent wave operations and float->int casts of NaN (defined on the GPU, relied upon where
the reference's shaders do) are outside what the stand-in checks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def run_emulated(pytest_args, timeout, fast=False):
    if fast:      # lanes as fibers, workgroups over the host cores, no sanitizers
        env = dict(os.environ, KJ_HIP_EMU="fast")
    else:         # one host thread per lane, ASan + UBSan
        rt = subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
        env = dict(os.environ, KJ_HIP_EMU="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("KJ_AMD_LIB", None)
    # -s: a sanitizer report goes to stderr and the process dies; with pytest's capture on it would vanish
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider"] + pytest_args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ as the host compiler")
def test_a_slice_of_the_gpu_suite_passes_on_the_cpu_stand_in_under_sanitizers():
    r = run_emulated(["tests/test_gpu_parity.py", "tests/test_gpu_ssgi.py", "tests/test_gpu_shadow_denoise.py", "tests/test_zz_gpu_post.py",
                      "-k", "(test_rtdgi_per_pass_parity and city20k-123-77) or light_gbuffer[0] or 160-90 or 320-180-320-180"], timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ as the host compiler")
def test_the_gpu_suite_passes_on_the_cpu_stand_in():
    """A broad slice of the `-m gpu` parity tests in the stand-in's fiber mode (about four minutes on 8 cores; the tests' own tolerances apply unchanged): every renderer's
    per-pass parity on small extents, the cache, TAA, the path tracer, instance edits and the device-built trees on Cornell, the strip split and the compiled orchestrator on
    two ranks. Left to `scripts/run_gpu_suite_on_cpu.sh` (which runs all of it, ~9 minutes) and to the hardware: the cases sized for hardware (full-size frames, 1080p, 4K), the
    long free-running / pipelining / convergence runs, the larger scenes of the ray-query, tree-build and ray-pass-form tests, the three-rank and whole-lighting-frame split
    cases (tests/test_multigpu_emulated.py and tests/test_bench_emulated.py run those over real processes), and the compiled C++ host, which links the real library."""
    r = run_emulated(["tests", "--deselect", "tests/test_gpu_fullsize.py", "--deselect", "tests/test_gpu_baseline_sizes.py", "-k", "not ruins and not 4k and not 2-2048-1024 and not 1920 and not cpp_world_render_passes and not converges_to_reference_pt and not pipelined_frames and not free_running_structure and not with_ssgi_guide and not cornell-256-256 and not city20k-320-192 and not (test_strip_split_with_the_irradiance_cache_is_bit_exact and 3-320-208) and not (test_ray_queries_bit_exact and pica) and not 8-192-256 and not 384-800 and not (reflections and (3-320-208 or 192-416 or 64-1248 or 160-416)) and not small_batches_walk and not (native_split and 3-320-208) and not (ray_pass_forms and city20k) and not (device_built_lbvh and (pica or city20k)) and not (reflections and 2-171-99-True) and not whole_lighting_frame and not strip_split_with_the_irradiance_cache and not pipelined_split_frames and not pipelined_lighting_frames and not (test_ray_queries_bit_exact and city20k)"],
                     timeout=2400, fast=True)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout and " error" not in r.stdout, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 41, tail
