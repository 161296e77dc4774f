"""Host BVH build (kajiya_amd/csrc/bvh_build.cpp — the role of vkCmdBuildAccelerationStructuresKHR): the multi-threaded build must emit
byte-identical nodes and leaf-ordered triangles to the sequential one (subtrees are built by worker threads and spliced in), on regular and
degenerate inputs, and reference every triangle exactly once."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh") / "bvh_build_check")
    csrc = os.path.join(ROOT, "kajiya_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
                           os.path.join(ROOT, "tests", "bvh_build_check.cpp"), os.path.join(csrc, "bvh_build.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("n,kind", [(1000, 0), (70000, 0), (300000, 0), (120000, 1), (150000, 2), (200000, 3)])
def test_parallel_build_equals_sequential(harness, n, kind):
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ, KJ_BVH_THREADS=threads)
        outs.append(subprocess.check_output([harness, str(n), str(kind)], env=env).decode().split())
    assert outs[0] == outs[1] == outs[2], outs
    nodes, tris, max_stack, dup = (int(v) for v in outs[0][:4])
    assert tris == n and dup == 0 and nodes >= 1 and max_stack >= 1
