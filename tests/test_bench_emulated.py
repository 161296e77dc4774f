"""bench.py's contract, exercised without a GPU: the driver's exact launch line for N > 1 (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`) with bench.py running against the
product source on the CPU stand-in for HIP (tests/hip_emu/run_with_emu.py) at a toy size, and the N = 1 form. Checks the LOGIC — rank /
world handling, the strip split over a process group (gloo, host-staged: bench.py's debugging transport), barrier + max-over-ranks
timing, exactly one JSON line from rank 0 with the contract's keys. Every number in that line is meaningless here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tests", "hip_emu", "run_with_emu.py")
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]
TOY = ["--steps", "3", "--warmup", "2", "--width", "128", "--height", "96", "--tris", "20000", "--profile-frames", "2", "--no-also"]   # --no-also: the 4K / 1440p side measurements are sized for the GPU


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++ as the host compiler")
@pytest.mark.parametrize("orchestrator", ["python", "compiled", "compiled, damaged transport"])
def test_bench_two_ranks_under_torch_distributed_run(orchestrator):
    """orchestrator "compiled": what a real `bench.py --gpus N` does by default -- the compiled orchestrator with a communicator of its own -- with the
    socket stand-in for RCCL (tests/rccl_stub) as that communicator; "damaged transport": one rank's messages arrive with a flipped byte, the
    self-test before frame 0 fails on every rank and the job falls back to the Python orchestrator instead of rendering garbage."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, KJ_BENCH_SHARE_GPU0="1", HIP_EMU_WORKERS="4")
    if orchestrator != "python":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_multigpu_emulated as TM
        env.update(KJ_SPLIT_NATIVE="1", KJ_RCCL_LIB=TM.build_rccl_stub())
        if "damaged" in orchestrator:
            env["KJ_RCCL_STUB_CORRUPT"] = "1"
            env["KJ_RCCL_STUB_TIMEOUT_MS"] = "5000"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        RUN, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + TOY, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = _json_line(r.stdout)
    assert all(k in j for k in CONTRACT), sorted(j)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 2 and j["scaling"] == "strong" and j["higher_is_better"] is True
    assert j["metric"] == "gi_mrays_per_s" and j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "2-way screen-tile split" in j["config"]["parallelism"]
    assert j["config"]["rays_per_frame"] > 1000                                   # both strips' rays were counted
    assert j["comm_ranks"] == 2                                                    # what the communicator reports, not what --gpus asked for
    if orchestrator == "compiled":
        assert "orchestrator: compiled" in j["config"]["parallelism"] and "2 ranks OK" in r.stderr
    else:
        assert "orchestrator: python" in j["config"]["parallelism"]
    if "damaged" in orchestrator:
        assert "2 ranks FAILED" in r.stderr and "falling back to the Python orchestrator" in r.stderr


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++ as the host compiler")
def test_bench_single_rank_line_has_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, RUN, os.path.join(ROOT, "bench.py")] + TOY, cwd=ROOT, env=dict(os.environ, HIP_EMU_WORKERS="8"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = _json_line(r.stdout)
    assert all(k in j for k in CONTRACT + ["cpu_baseline"]), sorted(j)
    assert j["n_gpus"] == 1 and j["scaling"] in ("weak", "strong") and j["comm_ranks"] == 1
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic"]) <= set(j["roofline"])
    cb = j["cpu_baseline"]
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(cb) and cb["kind"] in ("port", "reference") and cb["value"] > 0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++ as the host compiler")
def test_config3_lighting_frame_two_processes_over_the_compiled_transport():
    """scripts/config3_split_bench.py --check under the driver's kind of launch line: BASELINE configs[2]'s whole lighting frame (SSAO guide, sun shadows,
    cache + rtdgi, reflections, deferred combine, TAA on the lit image) split over two PROCESSES, the compiled orchestrator's ncclSend / ncclRecv served by the
    socket stand-in for RCCL -- every rank's rows of the lit image and of the TAA output equal the one-GPU frame's, texel for texel."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_multigpu_emulated as TM
    env = dict(os.environ, KJ_BENCH_SHARE_GPU0="1", HIP_EMU_WORKERS="4", KJ_SPLIT_NATIVE="1", KJ_RCCL_LIB=TM.build_rccl_stub())
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        RUN, os.path.join(ROOT, "scripts", "config3_split_bench.py"), "--scene", "glossy", "--res", "128x96", "--frames", "2", "--warmup", "2", "--motion-halo", "8", "--check"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"config"' in l]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["ranks"] == 2 and j["processes"] == 2 and j["orchestrator"] == "compiled" and j["mismatching_texels_vs_one_gpu"] == 0 and j["frame_ms_wall"] > 0
    assert "exchange self-test OK" in r.stderr
