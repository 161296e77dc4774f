"""TEST INFRASTRUCTURE: lets the `-m gpu` parity tests drive the CPU stand-in build of the product (tests/hip_emu/build_emu.py) unchanged.
The tests are written against torch CUDA tensors; on the stand-in "device memory" is host memory, so:
  * a torch function mode maps device="cuda..." to "cpu" and makes Tensor.cuda() the identity,
  * torch.cuda.{synchronize, is_available, current_stream, Stream, Event, stream} become no-ops / trivial objects,
  * kajiya_amd.lib.tensor_from_ptr builds a CPU tensor over the raw pointer, lib._stream_ptr returns the null stream,
  * kajiya_amd.lib.LIB_PATH points at the instrumented library.
Installed by tests/conftest.py only when KJ_HIP_EMU=1 is set explicitly; the product package is not modified on disk and a normal run
(no such variable) never sees any of this."""
import contextlib
import ctypes as C

import numpy as np
import torch
from torch.overrides import TorchFunctionMode


class _CpuAsCuda(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and str(dev).startswith("cuda"):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to and len(args) > 1 and isinstance(args[1], (str, torch.device)) and str(args[1]).startswith("cuda"):
            return args[0]
        return func(*args, **kwargs)


class _Stream:
    cuda_stream = 0
    def wait_event(self, e): pass
    def wait_stream(self, s): pass
    def synchronize(self): pass


class _Event:
    def __init__(self, enable_timing=False): pass
    def record(self, stream=None): pass
    def synchronize(self): pass
    def elapsed_time(self, other): return 0.0
    def wait(self, stream=None): pass


_mode = None


def install(lib_path):
    global _mode
    from kajiya_amd import lib
    lib.LIB_PATH = lib_path
    lib._LIB = None
    lib._stream_ptr = lambda: C.c_void_p(0)

    def tensor_from_ptr(ptr, nbytes, dtype, shape):
        a = np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), np.uint8)
        return torch.from_numpy(a).view(dtype).reshape(shape)
    lib.tensor_from_ptr = tensor_from_ptr
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.Event = _Event
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.device_count = lambda: 8
    torch.cuda.current_device = lambda: 0
    torch.cuda.get_device_name = lambda *a, **k: "hip_emu (CPU stand-in)"
    torch.cuda.empty_cache = lambda: None
    _mode = _CpuAsCuda()
    _mode.__enter__()
