#!/bin/bash
# Records tests/golden/ref_hlsl/*.npz: for one parametrisation of each test of tests/test_ref_hlsl.py, every image / buffer the reference's own shader text
# (compiled for the CPU by oracle/ref_hlsl, read in place from /root/reference) wrote, pass by pass. With them the same tests -- the oracle against what the
# reference's text produced -- run where neither the checkout nor oracle/_ref/libref_hlsl.so exists (tests/ref_hlsl.py: golden / replaying).
# Needs the reference checkout (this container). Re-run after changing a recorded test's sequence of passes.
cd "$(dirname "$0")/.." && rm -rf tests/golden/ref_hlsl && KJ_REF_GOLDEN_RECORD=1 python -m pytest tests/test_ref_hlsl.py -q -x "$@" && du -sh tests/golden/ref_hlsl && ls -la tests/golden/ref_hlsl
