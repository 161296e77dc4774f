#!/bin/bash
# Runs the `-m gpu` parity tests against the product source compiled for the CPU (tests/hip_emu): no GPU needed.
#   scripts/run_gpu_suite_on_cpu.sh [pytest args]              fiber mode: ~9 min on 8 cores for everything but the full-size / 1080p cases
#   scripts/run_gpu_suite_on_cpu.sh --sanitize [pytest args]   one host thread per lane under ASan + UBSan: ~1.5 h; leaves out the slowest tests
# Never a statement about the hardware or about hipcc's code generation: the same tests on an MI355X are.
cd "$(dirname "$0")/.."
if [ "$1" = "--sanitize" ]; then
  shift
  RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
  KJ_HIP_EMU=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests -q -s -m gpu -p no:cacheprovider --durations=15 --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_baseline_sizes.py \
    -k "not ruins and not 4k and not 2-2048-1024 and not 1920 and not converges_to_reference_pt and not pipelined_frames and not cpp_world_render_passes" "$@"
else
  KJ_HIP_EMU=fast python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 -rxX --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_baseline_sizes.py -k "not ruins and not 4k and not 2-2048-1024 and not 1920 and not cpp_world_render_passes" "$@"
fi
