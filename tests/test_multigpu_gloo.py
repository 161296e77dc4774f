"""N>1 path on CPU: two processes over `gloo` run DistComm's halo exchange / all-gather on strip-owned images
(the same call SplitRtdgi makes between passes, with host tensors standing in for the renderer surfaces)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q, prepared=False, packed=False):
    import torch
    import torch.distributed as dist
    from kajiya_amd import multigpu
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = multigpu.DistComm(dist, rank, world, packed=packed)
        strips = multigpu.plan_strips(H, world)
        g = torch.Generator().manual_seed(1234)
        results = {}
        for res, bpt, halo in (("h", 8, 51), ("f", 8, 2), ("f", 4, None), ("h", 1, 9)):
            hh = (H + 1) // 2 if res == "h" else H
            ww = ((W + 1) // 2 if res == "h" else W) * bpt
            truth = torch.randint(0, 256, (hh, ww), dtype=torch.uint8, generator=g)   # same on both ranks (same seed)
            own = multigpu.half_rows(*strips[rank], H) if res == "h" else strips[rank]
            mine = torch.full_like(truth, 0xEE if rank == 0 else 0x77)
            mine[own[0]:own[1]] = truth[own[0]:own[1]]
            xf = multigpu.transfers(strips, halo, res, H)
            if prepared:      # the path SplitRtdgi._exchange takes: resolve the plan once, replay it (twice here: the views must stay valid)
                spec = comm.prepare(xf, lambda r, a, b: mine[a:b])
                comm.run_prepared(spec)
                comm.run_prepared(spec)
            else:
                comm.run(xf, lambda r, a, b: mine[a:b])
            lo = 0 if halo is None else max(0, own[0] - halo)
            hi = hh if halo is None else min(hh, own[1] + halo)
            ok = bool((mine[lo:hi] == truth[lo:hi]).all())
            # rows outside the halo must be untouched
            outside = torch.cat([mine[:lo], mine[hi:]])
            ok_out = bool((outside == (0xEE if rank == 0 else 0x77)).all()) if outside.numel() else True
            results[(res, bpt, halo)] = (ok, ok_out)
        if prepared:
            # several surfaces in ONE batched group, as one exchange point of the frame does (names tag the rows)
            imgs = {}
            truth = {}
            xfers = []
            for name, (res, bpt, halo) in {"a": ("h", 8, 12), "b": ("f", 4, None), "c": ("f", 8, 3)}.items():
                hh = (H + 1) // 2 if res == "h" else H
                ww = ((W + 1) // 2 if res == "h" else W) * bpt
                truth[name] = torch.randint(0, 256, (hh, ww), dtype=torch.uint8, generator=g)
                own = multigpu.half_rows(*strips[rank], H) if res == "h" else strips[rank]
                imgs[name] = torch.full_like(truth[name], 0x10 + rank)
                imgs[name][own[0]:own[1]] = truth[name][own[0]:own[1]]
                xfers += [(src, dst, (name, a), b) for (src, dst, a, b) in multigpu.transfers(strips, halo, res, H)]
            comm.run_prepared(comm.prepare(xfers, lambda r, na, b: imgs[na[0]][na[1]:b]))
            for name, (res, bpt, halo) in {"a": ("h", 8, 12), "b": ("f", 4, None), "c": ("f", 8, 3)}.items():
                hh = truth[name].shape[0]
                own = multigpu.half_rows(*strips[rank], H) if res == "h" else strips[rank]
                lo = 0 if halo is None else max(0, own[0] - halo)
                hi = hh if halo is None else min(hh, own[1] + halo)
                results[("batched", name)] = (bool((imgs[name][lo:hi] == truth[name][lo:hi]).all()), True)
        dist.barrier()
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_distcomm_halo_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    H, W = 208, 96
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(r for r, _ in got) == [0, 1]
    for rank, results in got:
        for key, (ok, ok_out) in results.items():
            assert ok, f"rank {rank}: halo rows wrong for {key}"
            assert ok_out, f"rank {rank}: rows outside the halo were written for {key}"


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,packed", [(2, False), (4, False), (3, True), (4, True)])
def test_distcomm_prepared_plans_gloo(world, packed):
    """The cached-plan path of the frame loop (DistComm.prepare + run_prepared, several surfaces per batched group, replayed) with 2 to 4
    processes: halo exchanges reach beyond the direct neighbour when strips are thin, all-gathers talk to every peer. `packed`: the opt-in
    one-message-per-peer variant (KJ_SPLIT_PACKED=1)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    H, W = 208, 64
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q, True, packed)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(r for r, _ in got) == list(range(world))
    for rank, results in got:
        assert any(k[0] == "batched" for k in results)
        for key, (ok, ok_out) in results.items():
            assert ok and ok_out, f"rank {rank}: {key}"


def test_transfer_plan_is_symmetric_and_minimal():
    from kajiya_amd import multigpu
    st = multigpu.plan_strips(1080, 8)
    x = multigpu.transfers(st, 16, "f", 1080)
    # with a 16-row halo and >=128-row strips every rank only talks to its neighbours
    assert all(abs(s - d) == 1 for s, d, _, _ in x)
    assert sum(b - a for _, _, a, b in x) == 16 * 2 * 7
    xa = multigpu.transfers(st, None, "f", 1080)
    assert sum(b - a for _, _, a, b in xa) == 1080 * 7
    # pinned top rows: everybody but their owner receives them once, whether or not the halo already reaches them
    xp = multigpu.transfers(st, 16, "f", 1080, pin=1)
    extra = sorted(set(xp) - set(x))
    assert extra == [(0, d, 0, 1) for d in range(1, 8)] and set(x) <= set(xp)
    for d in range(8):
        rows = sorted(r for s_, d_, a, b in xp if d_ == d for r in range(a, b))
        assert len(rows) == len(set(rows))                                            # no row travels twice to the same rank
    hp = multigpu.transfers(multigpu.plan_strips(208, 3), 12, "h", 208, pin=1)
    assert (0, 2, 0, 1) in hp and (0, 1, 0, 1) in hp and not any(d == 0 and a == 0 for s_, d, a, b in hp)
    hq = multigpu.transfers(multigpu.plan_strips(64, 2), 15, "h", 64, pin=1)          # strips of 16 half-res rows: the halo reaches row 0 -- one merged span
    assert [t for t in hq if t[1] == 1] == [(0, 1, 0, 16)]


class _StubPipe:
    """what SplitRtdgi needs from a GpuPipeline before frame 0"""
    ircache = None


def _self_test_worker(rank, world, port, H, W, q, sabotage):
    import torch
    import torch.distributed as dist
    from kajiya_amd import multigpu
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = multigpu.DistComm(dist, rank, world, packed=(world == 3))
        split = multigpu.SplitRtdgi(comm, {rank: _StubPipe()}, W, H, motion_halo=8)
        if sabotage and rank == 1:      # a transport that drops what rank 1 receives: the self-test must notice (on every rank: collective verdict)
            real = comm.run_prepared

            def lossy(spec):
                if isinstance(spec, list):
                    spec = [(s, (t.clone() if not s else t), p) for (s, t, p) in spec]
                real(spec)
            comm.run_prepared = lossy
        q.put((rank, split.self_test(device="cpu")))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,sabotage", [(2, False), (3, False), (2, True)])
def test_split_transport_self_test_gloo(world, sabotage):
    """SplitRtdgi.self_test (what bench.py runs before frame 0 of an N > 1 job and reports as "RCCL <n> ranks OK"): passes over a working
    transport with 2 and 3 processes (3: the packed one-message-per-peer mode), and fails -- on EVERY rank -- when one rank's receives
    never reach its images."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_self_test_worker, args=(r, world, port, 208, 64, q, sabotage)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(got) == list(range(world))
    assert all(v is (not sabotage) for v in got.values()), got
