#!/usr/bin/env python3
"""Host-side (Python + ctypes) cost per frame of the screen-tile split orchestrator, measured with N virtual ranks on one GPU:
enqueue time without synchronisation vs GPU time. Per-rank cost in a real run ~= enqueue time / N.
usage: split_host_overhead.py N [WxH] [--native]  (--native: the compiled orchestrator, csrc/split.cpp, instead of multigpu.SplitRtdgi; at 1080p with many virtual ranks the one GPU is the bottleneck and back-pressures the enqueue;
pass a small extent, e.g. 512x288, to read the host cost alone)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from kajiya_amd import lib, scenes, frame, multigpu

native = "--native" in sys.argv
sys.argv = [a for a in sys.argv if a != "--native"]
W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2:   # a tiny extent makes the GPU side negligible, so the enqueue time is pure host (Python + ctypes + launch) cost
    W, H = map(int, sys.argv[2].split("x"))
dev = lib.Device(0)
scene = lib.Scene(dev, scenes.procedural_city(target_tris=200_000, seed=1234))
pipes = {r: lib.GpuPipeline(dev, scene, W, H, use_ircache=True) for r in range(N)}
split = multigpu.NativeSplit(N, pipes, W, H, motion_halo=16) if native else multigpu.SplitRtdgi(multigpu.LocalComm(N), pipes, W, H, motion_halo=16)
fs = frame.FrameState((W, H)); fs.ircache_enabled = True
fcs = []
for i in range(40):
    fcs.append(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004))); fs.retire_frame()
pipes[0].render_inputs(fcs[0]); pipes[0].reprojection()
for r in range(1, N):
    q = pipes[r]; p0 = pipes[0]
    q.geometric_normal, q.gbuffer, q.depth, q.velocity, q.sky16, q.sky64 = p0.geometric_normal, p0.gbuffer, p0.depth, p0.velocity, p0.sky16, p0.sky64
    q.reprojection_map_ptr = p0.reprojection_map_ptr
def frame_(i):
    dev.frame_begin(fcs[i]); split.gi_frame(); split.taa_frame()
for i in range(8): frame_(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8, 40): frame_(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{'native' if native else 'python'} orchestrator, N={N} virtual ranks: host enqueue {1e3*(t1-t0)/32:.3f} ms/frame ({1e3*(t1-t0)/32/N:.3f} per rank), total incl. GPU drain {1e3*(t2-t0)/32:.3f} ms/frame")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(8, 24): frame_(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
