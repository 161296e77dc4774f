#!/usr/bin/env python3
"""Bakes a scene to kajiya's .mesh/.image format and runs the compiled C++ host (examples/world_render_passes) on it: whole lighting frame
(ssgi, sun shadows + denoise, ircache, rtdgi, rtr, light_gbuffer, TAA) issued from C++ on one stream. usage: cpp_host_bench.py [city|glossy] [WxH] [frames] [tris]"""
import os, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import baked_writer as BW
from kajiya_amd import rtr_tables, scenes as S

name = sys.argv[1] if len(sys.argv) > 1 else "glossy"
W, H = map(int, (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tris = int(sys.argv[4]) if len(sys.argv) > 4 else 200_000
sd, cam = (S.glossy_test_scene(), "0 1 0 9 3 0.004") if name == "glossy" else (S.procedural_city(target_tris=tris, seed=1234), "0 2 0 30 6 0.004")
d = tempfile.mkdtemp(prefix="kj_baked_")
lines = []
for mi, m in enumerate(sd.meshes):
    mesh_bytes, images = BW.bake_triangle_mesh(m)
    open(os.path.join(d, f"m{mi}.mesh"), "wb").write(mesh_bytes)
    for ident, blob in images.items():
        open(os.path.join(d, f"{ident:8x}.image"), "wb").write(blob)
    lines.append(f"mesh m{mi}.mesh")
for mi, xf in sd.instances:
    lines.append("instance %d %s" % (mi, " ".join(repr(float(v)) for v in np.asarray(xf, np.float32).reshape(-1))))
lines.append("camera " + cam)
open(os.path.join(d, "scene.txt"), "w").write("\n".join(lines) + "\n")
t, (ranking, scrambling, sobol, offsets) = rtr_tables.standin_tables()
open(os.path.join(d, "rtr_tables.bin"), "wb").write(ranking.tobytes() + scrambling.tobytes() + sobol.tobytes() + offsets.tobytes())
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
out = subprocess.check_output([os.path.join(ROOT, "examples", "world_render_passes"), os.path.join(ROOT, "kajiya_amd", "data", "bluenoise_256_rgba8.bin"), d, str(W), str(H), str(frames),
                               os.path.join(d, "out")])
print(out.decode().strip()[:-1] + f', "scene": "{name}", "triangles": {sum(m.triangle_count for m in sd.meshes)}}}')
