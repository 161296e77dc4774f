#!/bin/bash
# The pass-level pin (tests/test_ref_hlsl.py: the reference's own shader text against the oracle, number for number) at k times the default frame extents, live only.
# Most surfaces are stored in 16 bits or fewer, so a last-bit difference in the fp32 arithmetic moves about one texel in ten thousand: the default extents (a few thousand
# texels per surface) see differences of association and evaluation order, this run has the power for rarer ones. k = 8: 512x512 .. 832x480 frames, 2 minutes; k = 16: 8 minutes.
cd "$(dirname "$0")/.." && KJ_REF_HLSL_SCALE=${1:-8} python -m pytest tests/test_ref_hlsl.py -q -p no:cacheprovider --durations=25 -k "not inc_ and not ircache" "${@:2}"
