"""RCCL first contact: the compiled orchestrator's transport -- the hand-declared prototypes split.cpp resolves with dlopen -- against
the REAL librccl on the one GPU a build box has. A communicator of ONE rank (ncclGetUniqueId -> ncclCommInitRank(nranks = 1)) is the wire of a split whose
ranks all live in this process (kj_split_create's loopback mode): every packed message of the schedule is an ncclSend to self matched by an ncclRecv from self
inside one group, the irradiance cache's summaries travel through ncclAllGather. Compared bit for bit with the same frames over the device-to-device
"virtual" transport and with the unsplit frame."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_parity as T  # noqa: E402

IRC_BUFS = ("meta", "grid_meta", "entry_cell", "spatial", "irradiance", "aux", "life", "pool", "reposition_proposal", "reposition_proposal_count")


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 3])
def test_native_split_over_a_one_rank_rccl_communicator_matches_the_virtual_transport(gpu, device, n_ranks):
    import torch
    from kajiya_amd import multigpu, frame
    if os.environ.get("KJ_HIP_EMU"):
        pytest.skip("needs the real RCCL and a GPU (the socket stand-in covers the transport between CPU processes: tests/test_multigpu_emulated.py)")
    W, H = 320, 208
    comm, comm_ranks, comm_rank = multigpu.NativeSplit.rccl_one_rank_comm()
    print(f"RCCL communicator: comm_ranks = {comm_ranks}, rank = {comm_rank} (ncclCommCount / ncclCommUserRank)")
    assert (comm_ranks, comm_rank) == (1, 0)
    scene = gpu.Scene(device, T._scenes()["city20k"])
    ref = gpu.GpuPipeline(device, scene, W, H, use_ircache=True)
    ref.ircache_set_deferred(True)
    virt_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
    rccl_pipes = {r: gpu.GpuPipeline(device, scene, W, H, use_ircache=True) for r in range(n_ranks)}
    virt = multigpu.NativeSplit(n_ranks, virt_pipes, W, H, motion_halo=8)
    over_rccl = multigpu.NativeSplit(n_ranks, rccl_pipes, W, H, motion_halo=8, nccl_comm=comm, own_comm=True)
    try:
        assert over_rccl.self_test() is True      # every kind of exchange of the schedule + the fixed-size all-gather, through RCCL, checked row by row
        fs = frame.FrameState((W, H))
        fs.ircache_enabled = True
        for fi in range(5):
            fc = fs.prepare_frame_constants(frame.orbit_camera(fi, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.02))
            fs.retire_frame()
            ref.frame(fc); ref.taa_frame()
            for pipes in (virt_pipes, rccl_pipes):
                for r in range(n_ranks):
                    pipes[r].render_inputs(fc)
                    pipes[r].reprojection()
            for sp in (virt, over_rccl):
                sp.gi_frame(); sp.taa_frame()
                sp.gather_output("spatial_filtered_tex")
                sp.gather_output(f"TAA/taa:{fi % 2}")
            torch.cuda.synchronize()
            a = ref.surface("spatial_filtered_tex", torch.int16, (H, W, 4))
            ta = ref.taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))
            for r in range(n_ranks):
                for tag, pipes in (("virtual", virt_pipes), ("rccl", rccl_pipes)):
                    assert torch.equal(a, pipes[r].surface("spatial_filtered_tex", torch.int16, (H, W, 4))), f"frame {fi} rank {r} ({tag}): GI image differs"
                    assert torch.equal(ta, pipes[r].taa_surface(f"taa:{fi % 2}", torch.int16, (H, W, 4))), f"frame {fi} rank {r} ({tag}): TAA image differs"
                for name in IRC_BUFS:
                    assert torch.equal(ref.ircache_buffer(name, torch.uint8), rccl_pipes[r].ircache_buffer(name, torch.uint8)), f"frame {fi} rank {r}: ircache buffer {name} differs over RCCL"
        meta = ref.ircache_buffer("meta", torch.int32).cpu().numpy()
        assert meta[3] > 50, meta      # the cache did allocate entries
    finally:
        over_rccl.close()
        virt.close()
