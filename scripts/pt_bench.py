#!/usr/bin/env python3
"""Reference path tracer throughput (BASELINE configs[4] shape: 4K, N spp). Single process: one GPU renders its tile-interleaved
share (interleave = WORLD_SIZE / RANK when launched under torch.distributed.run; images are summed at the end)."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from kajiya_amd import lib, scenes, frame

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
W, H, SPP = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = lib.Device(local)
scene = lib.Scene(dev, scenes.procedural_ruins(target_tris=4_000_000, seed=5678))
gp = lib.GpuPipeline(dev, scene, W, H, device=f"cuda:{local}")
fs = frame.FrameState((W, H))
acc = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local}")
counter = torch.zeros(1, dtype=torch.int64, device=f"cuda:{local}")
fcs = [fs.prepare_frame_constants(frame.orbit_camera(0, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.0)) or fs.retire_frame() for _ in range(SPP + 2)]
fcs = []
for i in range(SPP + 2):
    fcs.append(fs.prepare_frame_constants(frame.orbit_camera(0, (W, H), center=(0.0, 3.0, 0.0), radius=34.0, height=5.0, rate=0.0))); fs.retire_frame()
for fc in fcs[:2]:
    dev.frame_begin(fc); gp.reference_path_trace(acc, interleave=(world, rank))
acc.zero_(); counter.zero_(); torch.cuda.synchronize()
if world > 1: dist.barrier()
t0 = time.perf_counter()
for fc in fcs[2:]:
    dev.frame_begin(fc); gp.reference_path_trace(acc, interleave=(world, rank), ray_counter=counter)
if world > 1:
    dist.all_reduce(acc); dist.all_reduce(counter)     # non-owned tiles are zero: the sum assembles the image
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if rank == 0:
    print(json.dumps({"workload": f"reference path tracer, procedural_ruins ~4M tris, {W}x{H}, {SPP} spp, {world} GPU(s) tile-interleaved",
                      "seconds": round(dt, 4), "ms_per_spp": round(1e3 * dt / SPP, 3), "Mrays_per_s": round(counter.item() / dt / 1e6, 1),
                      "rays_per_path": round(counter.item() / (W * H * SPP), 2), "mean_radiance": [round(float(v), 4) for v in acc[..., :3].mean(dim=(0, 1))]}))
