"""TEST INFRASTRUCTURE: `python tests/hip_emu/run_with_emu.py <script.py> [args...]` runs a script of this repository (bench.py,
scripts/*.py) against the product source compiled for the CPU stand-in for HIP, fiber mode — torch's CUDA tensors mapped to host memory
(cpu_as_cuda.py). For checking a script's LOGIC where no GPU is at hand (rank handling, JSON output, exchange plans); every time it prints is
meaningless."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ["KJ_HIP_EMU"] = "fast"
import build_emu, cpu_as_cuda

cpu_as_cuda.install(build_emu.build())
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
