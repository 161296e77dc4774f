#!/usr/bin/env python3
"""Reads the constant table SPATIAL_RESOLVE_OFFSETS out of the reference's renderers/rtr.rs (:402-915, `[(i32,i32,i32,i32); 16*4*8]`)
and writes it as caller data for kj_rtr_create: kajiya_amd/data/spatial_resolve_offsets_i32x4.bin (512 x 4 little-endian int32).
Run in the build container only (/root/reference does not exist on the GPU box); the output is committed."""
import os, re, sys
import numpy as np

REF = "/root/reference/crates/lib/kajiya/src/renderers/rtr.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kajiya_amd", "data", "spatial_resolve_offsets_i32x4.bin")

src = open(REF).read()
body = src[src.index("pub const SPATIAL_RESOLVE_OFFSETS"):]
body = body[body.index("= [") + 3:body.index("];")]
rows = re.findall(r"\(\s*(-?\d+)(?:i32)?\s*,\s*(-?\d+)(?:i32)?\s*,\s*(-?\d+)(?:i32)?\s*,\s*(-?\d+)(?:i32)?\s*\)", body)
t = np.array(rows, dtype=np.int32)
assert t.shape == (16 * 4 * 8, 4), t.shape
assert (t[::16, :2] == 0).all(), "tap 0 of every 16-tap group is the centre"
t.tofile(OUT)
print("wrote", OUT, t.shape, "xy range", t[:, :2].min(), t[:, :2].max())
