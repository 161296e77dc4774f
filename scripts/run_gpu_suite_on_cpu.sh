#!/bin/bash
# Runs the `-m gpu` parity tests against the product source compiled for the CPU (tests/hip_emu) under ASan + UBSan: no GPU needed.
# Slow (every workgroup = 64 host threads): about an hour on 8 cores; left out: the full-size tests, the 1080p cases, the 512^2 convergence run, the
# 48-frame pipelining test and the compiled C++ host (it links the real library). Usage: scripts/run_gpu_suite_on_cpu.sh [pytest args]
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
KJ_HIP_EMU=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests -q -s -m gpu -p no:cacheprovider --durations=15 --deselect tests/test_gpu_fullsize.py -k "not 1920 and not fullsize and not converges_to_reference_pt and not pipelined_frames and not cpp_world_render_passes" "$@"
