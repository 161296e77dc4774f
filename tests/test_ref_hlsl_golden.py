"""The committed outputs of the reference's shader text (tests/golden/ref_hlsl/*.npz, recorded by scripts/make_ref_hlsl_golden.sh from the live run of
oracle/_ref/libref_hlsl.so) replayed against the oracle: tests/test_ref_hlsl.py with KJ_REF_HLSL_REPLAY=1, i.e. WITHOUT the compiled reference text -- what a
checkout without /root/reference and without the prebuilt library can still verify. Run here too so that the fixtures cannot go stale unnoticed."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_reference_outputs_replay_against_the_oracle():
    env = dict(os.environ, KJ_REF_HLSL_REPLAY="1")
    env.pop("KJ_REF_GOLDEN_RECORD", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "tests/test_ref_hlsl.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-2500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 16 and " failed" not in r.stdout, tail
    assert len([f for f in os.listdir(os.path.join(ROOT, "tests", "golden", "ref_hlsl")) if f.endswith(".npz")]) >= 16
