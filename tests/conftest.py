import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import okj_py
    okj_py.lib()
    return okj_py


@pytest.fixture(scope="session")
def gpu():
    """The product library on cuda:0. Fails (does not skip) if the HIP extension is missing."""
    import torch
    if os.environ.get("KJ_HIP_EMU") in ("1", "fast"):      # "1": threads + sanitizers; "fast": fibers, no sanitizers
        # explicit opt-in (tests/test_emulated_gpu_suite.py): the same tests drive the product source compiled against the CPU stand-in
        # for HIP, under sanitizers — see tests/hip_emu/. Never the default; a GPU box runs the branch below.
        sys.path.insert(0, os.path.join(TESTS, "hip_emu"))
        import build_emu, cpu_as_cuda
        cpu_as_cuda.install(build_emu.build())
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from kajiya_amd import lib
    lib.load()
    return lib


@pytest.fixture(scope="session")
def device(gpu):
    return gpu.Device(0)


def pytest_sessionfinish(session, exitstatus):
    """What the loosest bars were set against, as this session measured it (tests/parity.py: MEASURED)."""
    try:
        import parity
    except Exception:
        return
    if parity.MEASURED:
        print("\n[parity] worst values this session saw of the quantities the loose bars bound:")
        for k in sorted(parity.MEASURED):
            print(f"[parity]   {k}: {parity.MEASURED[k]:.3e}")
