"""GPU parity for the irradiance cache. The cache is order-dependent in the reference (lock-free allocation,
last-writer-wins votes: docs/gi-overview.md:296), so entry indices differ between runs. Deterministic pieces are
compared exactly on identical state; the full cache is compared structurally (per grid cell) and statistically."""
import ctypes as C
import numpy as np
import pytest

import parity as P
import test_gpu_parity as T

pytestmark = pytest.mark.gpu

BUFS = {"meta": np.uint32, "grid_meta": np.uint32, "entry_cell": np.uint32, "spatial": np.float32, "irradiance": np.float32, "aux": np.float32,
        "life": np.uint32, "pool": np.uint32, "entry_indirection": np.uint32, "reposition_proposal": np.float32, "reposition_proposal_count": np.uint32}


def _frames(W, H, n, use_ircache=True, scene="cornell"):
    from kajiya_amd import frame
    fs = frame.FrameState((W, H))
    fs.ircache_enabled = use_ircache
    out = []
    for i in range(n):
        if scene == "cornell":
            cam = frame.orbit_camera(i, (W, H), center=(0.0, 1.0, 0.0), radius=6.5, height=0.0, rate=0.02)
        else:
            cam = frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)
        out.append(fs.prepare_frame_constants(cam))
        fs.retire_frame()
    return out


def _upload_ircache(op, gp, torch):
    for name, dt in BUFS.items():
        src = op.ircache_buffer(name, np.uint8)
        dst = gp.ircache_buffer(name, torch.uint8)
        n = min(src.size, dst.numel())
        dst[:n].copy_(torch.from_numpy(src[:n].copy()))


def test_ircache_maintenance_and_sum_are_exact_on_identical_state(gpu, oracle, device):
    """scroll/age/scan/compact and the SH sum-up are deterministic given the same state (single-threaded oracle
    order == any GPU order up to the free-list permutation): run the oracle for several frames, upload its
    state, run one prepare() on both, compare."""
    one_frame_on_identical_state(gpu, oracle, device, "cornell", 128, 128)


def test_ircache_ray_passes_three_launches_on_identical_state(gpu, oracle, device):
    """kj_ircache_set_ray_pass_schedule(KJ_IRC_PASSES_SEQUENTIAL): the three ray passes as three launches (the default of rounds 1-4), against the same sequential oracle
    and the same statistical bar as the default chain schedule above."""
    one_frame_on_identical_state(gpu, oracle, device, "cornell", 128, 128, schedule="sequential")


def test_ircache_ray_passes_side_by_side_on_identical_state(gpu, oracle, device):
    """kj_ircache_set_ray_passes_side_by_side(1): the three ray passes in one launch, racing as the reference's barrier-free recording lets them
    (ircache.rs:396-481). Same maintenance results bit for bit, same ray counts; the SH sums sit much further from the SEQUENTIAL oracle than with
    three launches (measured 5.3e-2 and 1.25e-1 in two runs on MI355X against 1.3e-2 on this case: which pass's update of a slot a lookup or the next
    pass sees is decided by the race) -- a different valid schedule of a racy algorithm, not a parity claim: this test exercises the path (maintenance
    exact, ray counts equal) and holds the sums to a 0.3 sanity bar."""
    one_frame_on_identical_state(gpu, oracle, device, "cornell", 128, 128, schedule="side_by_side")


def one_frame_on_identical_state(gpu, oracle, device, scene_name, W, H, schedule="chain"):
    side_by_side = schedule == "side_by_side"
    import torch
    desc = T._scenes()[scene_name]
    oracle.lib().okj_set_threads(1)
    try:
        op = oracle.OraclePipeline(oracle.OracleScene(desc), W, H, use_ircache=True)
        gp = gpu.GpuPipeline(device, gpu.Scene(device, desc), W, H, use_ircache=True)
        gp.ircache_set_ray_pass_schedule(schedule)      # "chain" = the library's default (one launch, own-slot order kept); the oracle is the sequential one for all three
        fcs = _frames(W, H, 8, scene="cornell" if scene_name == "cornell" else "city")
        for fc in fcs[:6]:
            op.frame(fc)
        # GPU: run one frame to initialise, then overwrite its state with the oracle's
        gp.frame(fcs[0])
        torch.cuda.synchronize()
        gp_parity_fix = op  # noqa
        _upload_ircache(op, gp, torch)
        # both: next frame's prepare (scroll with the camera move, age, scan, compact)
        fc = fcs[6]
        op.render_inputs(fc)
        gp.dev.frame_begin(fc)
        op.L.okj_ircache_prepare(op.ircache, C.byref(fc))
        # the GPU handle's ping-pong parity must match the oracle's: both ran the same number of prepare() calls? no -> align by
        # comparing against whichever grid buffer is live: kj_ircache_buffer("grid_meta") returns the live one on both sides.
        gpu.check(gp.L.kj_ircache_prepare(gp.ircache, None))
        torch.cuda.synchronize()
        for name in ("meta", "entry_cell", "life", "spatial", "irradiance", "reposition_proposal_count"):
            a = op.ircache_buffer(name, np.uint8)
            b = gp.ircache_buffer(name, torch.uint8).cpu().numpy()
            n = min(a.size, b.size)
            assert np.array_equal(a[:n], b[:n]), name
        gm_a = op.ircache_buffer("grid_meta", np.uint32).reshape(-1, 2)
        gm_b = gp.ircache_buffer("grid_meta", torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 2)
        assert np.array_equal(gm_a, gm_b)
        meta = op.ircache_buffer("meta", np.uint32)
        alloc = int(meta[3])
        assert alloc > 50
        # free list: same set of free entries (order may differ)
        pa = op.ircache_buffer("pool", np.uint32)[alloc:]
        pb = gp.ircache_buffer("pool", torch.int32).cpu().numpy().view(np.uint32)[alloc:]
        assert np.array_equal(np.sort(pa), np.sort(pb))
        # compaction: same set of live entries in [1..alloc] (inclusive-scan off-by-one of the reference is preserved)
        ia = op.ircache_buffer("entry_indirection", np.uint32)[1:alloc + 1]
        ib = gp.ircache_buffer("entry_indirection", torch.int32).cpu().numpy().view(np.uint32)[1:alloc + 1]
        assert np.array_equal(ia, ib)
        # trace + sum-up on identical state: rays are deterministic per entry; lookups may allocate in different order,
        # so compare the SH of entries that existed before this frame
        op.L.okj_ircache_trace_irradiance(op.ircache, C.byref(fc), op.scene.h, op.sky16.ctypes.data, 16)
        gp.sky16.copy_(torch.from_numpy(op.sky16.view(np.int16)))
        gpu.check(gp.L.kj_ircache_trace_irradiance(gp.ircache, gp.scene.h, gp.sky16.data_ptr(), 16, None))
        op.L.okj_ircache_sum_up(op.ircache, C.byref(fc))
        gpu.check(gp.L.kj_ircache_sum_up_irradiance_for_sampling(gp.ircache, None))
        torch.cuda.synchronize()
        oc, oa = op.ircache_ray_counts(); gc, ga = gp.ircache_ray_counts()
        assert oc == gc and abs(oa - ga) <= 0.01 * oa + 4, (oc, oa, gc, ga)
        irr_a = op.ircache_buffer("irradiance", np.float32).reshape(-1, 12)
        irr_b = gp.ircache_buffer("irradiance", torch.float32).cpu().numpy().reshape(-1, 12)
        live = ia
        num = np.sqrt(((irr_a[live] - irr_b[live]) ** 2).sum()); den = np.sqrt((irr_a[live] ** 2).sum())
        print("ircache SH rel-L2 on identical state:", num / den, "entries", len(live))
        # A statistical bar, not the parity bar: in this (the reference's racy) mode a lookup sees however many of the same pass' updates
        # happen to have landed -- the sequential oracle sees all earlier ones, the GPU with four lanes per path and four times the waves in
        # flight sees fewer (measured 1.1e-2 with one lane per path, 2.3e-2 with four, 1080p city). Parity of the cache is held at 1e-3 by
        # the deterministic mode on both sides (deterministic_frames_on_identical_state below).
        assert num / den < (0.3 if side_by_side else 5e-2)
    finally:
        oracle.lib().okj_set_threads(oracle.lib().okj_get_max_threads())


def test_ircache_free_running_structure_and_statistics(gpu, oracle, device):
    """Independent 20-frame runs: the set of occupied grid cells and the mean cached irradiance agree."""
    import torch
    from kajiya_amd import scenes
    W = H = 160
    desc = scenes.cornell_box()
    op = oracle.OraclePipeline(oracle.OracleScene(desc), W, H, use_ircache=True)
    gp = gpu.GpuPipeline(device, gpu.Scene(device, desc), W, H, use_ircache=True)
    for fc in _frames(W, H, 20):
        op.frame(fc)
        gp.frame(fc)
    torch.cuda.synchronize()
    gm_a = op.ircache_buffer("grid_meta", np.uint32).reshape(-1, 2)
    gm_b = gp.ircache_buffer("grid_meta", torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 2)
    occ_a, occ_b = (gm_a[:, 1] & 1) != 0, (gm_b[:, 1] & 1) != 0
    inter, union = (occ_a & occ_b).sum(), (occ_a | occ_b).sum()
    print("ircache occupied cells: oracle", occ_a.sum(), "gpu", occ_b.sum(), "IoU", inter / union)
    assert occ_a.sum() > 100 and inter / union > 0.9
    irr_a = op.ircache_buffer("irradiance", np.float32).reshape(-1, 3, 4)
    irr_b = gp.ircache_buffer("irradiance", torch.float32).cpu().numpy().reshape(-1, 3, 4)
    both = np.nonzero(occ_a & occ_b)[0]
    l0_a = irr_a[gm_a[both, 0]][:, :, 0]; l0_b = irr_b[gm_b[both, 0]][:, :, 0]
    ma, mb = l0_a.mean(axis=0), l0_b.mean(axis=0)
    print("mean SH L0 per channel: oracle", ma, "gpu", mb)
    assert np.allclose(ma, mb, rtol=0.12, atol=2e-3)  # statistical: both runs are racy (OpenMP / GPU atomics)
    # per-cell values are independent noisy ReSTIR estimates (4 rays/entry/frame, 0.25 blend): require strong
    # correlation and a bounded mean absolute deviation rather than equality
    per_cell = np.abs(l0_a - l0_b).mean() / (np.abs(l0_a).mean() + 1e-6)
    corr = np.corrcoef(l0_a.sum(axis=1), l0_b.sum(axis=1))[0, 1]
    print("per-cell mean abs diff / mean:", per_cell, "correlation:", corr)
    assert per_cell < 0.45 and corr > 0.85
    # and the GI image with the cache bound agrees within the free-running bar
    ref = op.surface("spatial_filtered_tex", np.uint8, (-1,))
    got = gp.surface("spatial_filtered_tex", torch.uint8, (-1,)).cpu().numpy()
    r = P.compare(got, ref, "rgba16f")
    print("free-running GI with ircache:", r)
    assert r["rel_l2"] < 5e-2


@pytest.mark.parametrize("with_ssgi", [False, True])
def test_pipelined_frames_match_serial_frames(gpu, device, with_ssgi):
    """(with_ssgi: the SSAO guide is computed every frame as the bench does; it reads the frame constants the SIDE stream writes.)
    GpuPipeline.frame_pipelined issues frame N+1's ircache work on a second stream under frame N's screen-space tail.
    Dependencies are those of the serial order, so the result may differ only through the cache's own atomics races
    (which also make two serial runs differ): compare the time-averaged GI and TAA outputs."""
    import torch
    from kajiya_amd import frame
    W, H, N = 256, 160, 24
    desc = T._scenes()["city20k"]
    scene = gpu.Scene(device, desc)

    def fcs():
        fs = frame.FrameState((W, H))
        fs.ircache_enabled = True
        out = []
        for i in range(N + 1):
            out.append(fs.prepare_frame_constants(frame.orbit_camera(i, (W, H), center=(0.0, 2.0, 0.0), radius=30.0, height=6.0, rate=0.004)))
            fs.retire_frame()
        return out
    F = fcs()
    # inputs rendered once (they do not depend on the GI state)
    src = gpu.GpuPipeline(device, scene, W, H)
    inputs = []
    for fc in F:
        src.render_inputs(fc)
        src.reprojection()
        rp = gpu.tensor_from_ptr(src.reprojection_map_ptr.value, W * H * 8, torch.int16, (H, W, 4)).clone()
        inputs.append((src.geometric_normal.clone(), src.gbuffer.clone(), src.depth.clone(), rp))
    torch.cuda.synchronize()

    def run(pipelined):
        gp = gpu.GpuPipeline(device, scene, W, H, use_ircache=True)
        gp.sky64, gp.sky16 = src.sky64, src.sky16
        acc_gi = torch.zeros((H, W, 3), device="cuda")
        acc_taa = torch.zeros((H, W, 3), device="cuda")
        if pipelined:
            gp.pipeline_begin(F[0])
        for i in range(N):
            gn, gb, d, rp = inputs[i]
            gp.geometric_normal, gp.gbuffer, gp.depth = gn, gb, d
            gp.reprojection_map_ptr = C.c_void_p(rp.data_ptr())
            if pipelined:
                gp.frame_pipelined(F[i + 1], run_ssgi=with_ssgi)
                torch.cuda.current_stream().wait_event(gp._ev_taa[i & 1])    # spatial filter + TAA of this frame run on the third stream
            else:
                device.frame_begin(F[i])
                if with_ssgi:
                    gp.ssgi_frame()
                gp.gi_frame()
                gp.taa_frame()
            if i >= 8:
                acc_gi += gp.surface("spatial_filtered_tex", torch.float16, (H, W, 4))[..., :3].float()
                acc_taa += gp.taa_surface(f"taa:{i % 2}", torch.float16, (H, W, 4))[..., :3].float()
        torch.cuda.synchronize()
        return acc_gi.cpu().numpy(), acc_taa.cpu().numpy(), gp.ircache_ray_counts()
    a_gi, a_taa, a_rays = run(False)
    b_gi, b_taa, b_rays = run(True)
    c_gi, c_taa, _ = run(False)          # serial vs serial: the noise floor of the comparison
    rel = lambda x, y: float(np.sqrt(((x - y) ** 2).sum() / (y ** 2).sum()))
    floor_gi, floor_taa = rel(c_gi, a_gi), rel(c_taa, a_taa)
    print(f"pipelined vs serial: GI rel-L2 {rel(b_gi, a_gi):.5f} (serial-vs-serial floor {floor_gi:.5f}), TAA {rel(b_taa, a_taa):.5f} (floor {floor_taa:.5f}); ircache rays {a_rays} vs {b_rays}")
    assert np.isfinite(b_gi).all() and np.isfinite(b_taa).all()
    assert rel(b_gi, a_gi) < max(2e-2, 3 * floor_gi) and rel(b_taa, a_taa) < max(2e-2, 3 * floor_taa)
    assert abs(sum(a_rays) - sum(b_rays)) / max(1, sum(a_rays)) < 0.05


# ---------------------------------------------------------------------------------------------------------------------------------
# The cache's DETERMINISTIC mode on both sides (product: kj_ircache_set_deferred_updates; oracle: okj_ircache.hpp `deferred`): no
# outcome depends on thread interleaving, so the cache is held to the deterministic passes' bars instead of statistical ones.
def cache_state_per_cell(get):
    """`get(name, dtype)` -> flat numpy array of a cache buffer. Returns the cache as PER-CELL records (entry indices are an
    allocation-order artefact: one lookup that resolves to a neighbouring cell on one side shifts every later allocation)."""
    gm = get("grid_meta", np.uint32).reshape(-1, 2)
    occ = (gm[:, 1] & 1) != 0
    meta = get("meta", np.uint32)
    return dict(gm=gm, occ=occ, meta=meta, life=get("life", np.uint32), irradiance=get("irradiance", np.float32).reshape(-1, 12),
                aux=get("aux", np.float32).reshape(-1, 64, 4), spatial=get("spatial", np.float32).reshape(-1, 4),
                proposal=get("reposition_proposal", np.float32).reshape(-1, 4), votes=get("reposition_proposal_count", np.uint32), pool=get("pool", np.uint32))


def _vertex(v):
    """packed IrcVertex records (xyz f32, 11:10:11 normal in w) -> 6 floats"""
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 4)
    return np.concatenate([v[:, :3], P.unpack_11_10_11(v[:, 3].copy().view(np.uint32))], -1)


def assert_cache_parity(a, b, what, flip_cells=8, flip_tol=P.MISMATCH_TOL, verbose=True):
    """a = product, b = oracle (cache_state_per_cell). Integer state must agree except for a handful of cells (`flip_cells`, or 0.2 %)
    whose lookup resolved differently by a last-bit difference in a hit position; float state of the cells both sides occupy meets
    parity.within_bars_with_flips (an aux slot whose reservoir kept the other sample, or whose shadow ray grazed an edge the other way,
    is replaced as a whole: `flip_tol` of the records may, measured 1e-4 .. 7e-4 per frame on hardware, 0 on the CPU stand-in)."""
    occ_a, occ_b = a["occ"], b["occ"]
    n_occ = int(occ_b.sum())
    differ = int((occ_a != occ_b).sum())
    cap = max(flip_cells, int(2e-3 * n_occ))
    exact_layout = np.array_equal(a["gm"], b["gm"]) and np.array_equal(a["pool"], b["pool"]) and np.array_equal(a["meta"][:4], b["meta"][:4])
    assert differ <= cap, (what, "occupied-cell sets differ", differ, n_occ)
    assert abs(int(a["meta"][3]) - int(b["meta"][3])) <= cap and abs(int(a["meta"][2]) - int(b["meta"][2])) <= 4 * cap + 64, (what, a["meta"][:4], b["meta"][:4])
    both = np.nonzero(occ_a & occ_b)[0]
    ea, eb = a["gm"][both, 0], b["gm"][both, 0]
    flags = int(((a["gm"][both, 1] ^ b["gm"][both, 1]) != 0).sum())
    life = int((a["life"][ea] != b["life"][eb]).sum())
    votes = int((a["votes"][ea] != b["votes"][eb]).sum())
    assert flags <= cap and life <= cap and votes <= cap, (what, flags, life, votes)
    res = {}
    res["irradiance"] = P.compare_decoded(a["irradiance"][ea].reshape(-1, 4), b["irradiance"][eb].reshape(-1, 4))
    ra, rb = a["aux"][ea, 0:16].reshape(-1, 4), b["aux"][eb, 0:16].reshape(-1, 4)
    res["aux.reservoir"] = P.compare(np.ascontiguousarray(ra[:, :2]).view(np.uint8), np.ascontiguousarray(rb[:, :2]).view(np.uint8), "reservoir")
    res["aux.radiance"] = P.compare_decoded(a["aux"][ea, 16:32].reshape(-1, 4), b["aux"][eb, 16:32].reshape(-1, 4))
    res["aux.origin"] = P.compare_decoded(_vertex(a["aux"][ea, 32:48]), _vertex(b["aux"][eb, 32:48]), vector=True)
    res["spatial"] = P.compare_decoded(_vertex(a["spatial"][ea]), _vertex(b["spatial"][eb]), vector=True)
    res["reposition_proposal"] = P.compare_decoded(_vertex(a["proposal"][ea]), _vertex(b["proposal"][eb]), vector=True)
    if verbose:
        print(f"{what}: {n_occ} cells, layout {'bit-identical' if exact_layout else 'differs in %d cells' % differ}; life/flags/votes differ in {life}/{flags}/{votes}; " +
              ", ".join(f"{k} {v['rel_l2']:.1e} ({v['mismatch_frac']:.1e})" for k, v in res.items()))
    for k, v in res.items():
        assert P.within_bars_with_flips(v, flip_tol=flip_tol), f"{what}: {k}: {v}"
    return exact_layout


def deterministic_frames_on_identical_state(gpu, oracle, device, scene_name, W, H, warmup=5, frames=3, schedule="chain"):
    """Whole GI frames (cache maintenance, its three ray passes, rtdgi with its cache lookups, the replay of the recorded updates) in the
    cache's deterministic mode, each from IDENTICAL state: before every compared frame the oracle's cache buffers, rtdgi surfaces and
    G-buffer are uploaded to the product. Compared after each frame: the whole cache per cell, and the GI output."""
    import torch
    desc = T._scenes()[scene_name]
    op = oracle.OraclePipeline(oracle.OracleScene(desc), W, H, use_ircache=True)
    gp = gpu.GpuPipeline(device, gpu.Scene(device, desc), W, H, use_ircache=True)
    op.ircache_set_deferred(True)
    gp.ircache_set_deferred(True)
    # the schedule of the cache's three ray passes: the library's default chain (one launch; lookups of validation and tracing read the state before it) or three
    # launches (a snapshot between validation and tracing); the oracle restates whichever the product runs
    gp.ircache_set_ray_pass_schedule(schedule)
    op.ircache_set_chain_schedule(schedule == "chain")
    fcs = _frames(W, H, warmup + frames, scene="cornell" if scene_name == "cornell" else "city")
    repro_dev = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    get_g = lambda name, dt: gp.ircache_buffer(name, torch.uint8).cpu().numpy().view(dt)
    get_o = lambda name, dt: op.ircache_buffer(name, np.uint8).view(dt)
    exact = []
    for fi, fc in enumerate(fcs):
        op.render_inputs(fc); op.reprojection(fc)
        gp.dev.frame_begin(fc)
        if fi < warmup:               # the product runs the warm-up frames too: its handles' ping-pong parities then match the oracle's
            gp.render_inputs(fc); gp.reprojection(); gp.gi_frame()
            op.gi_frame(fc)
            continue
        T._sync_inputs(op, gp, torch)
        repro_dev.copy_(torch.from_numpy(op.reprojection_map))
        gp.reprojection_map_ptr = C.c_void_p(repro_dev.data_ptr())
        T._upload_state(gp, T._oracle_surfaces(op), torch)
        _upload_ircache(op, gp, torch)
        op.gi_frame(fc)
        gp.gi_frame()
        torch.cuda.synchronize()
        exact.append(assert_cache_parity(cache_state_per_cell(get_g), cache_state_per_cell(get_o), f"{scene_name} {W}x{H} frame {fi}"))
        r = P.compare(gp.surface("spatial_filtered_tex", torch.uint8, (-1,)).cpu().numpy(), op.surface("spatial_filtered_tex", np.uint8, (-1,)), "rgba16f")
        print(f"  GI output: rel-L2 {r['rel_l2']:.2e}, outliers {r['mismatch_frac']:.2e}")
        # one flipped slot of THIS frame's cache passes moves its entry's SH by a percent, and every pixel whose ray ends in that cell
        # reads it: the image-level bar holds; the count of slightly-off texels is reported and bounded loosely
        # measured on MI355X: round 4 3.5e-3 / 2.0e-3 of the texels at 1080p, 133 / 0 / 0 texels at 128^2; round 6 1.4e-3 at 1080p, 114 texels at 128^2
        # (profiles/r06_gpu_tests_summary.txt): the bar is twice the round-6 values, rounded up
        P.measured("GI output behind the deterministic cache, n = %d: mismatch fraction (bar max(3e-3, 270 / n))" % r["n"], r["mismatch_frac"])
        assert r["rel_l2"] <= P.REL_L2_TOL and r["bad_class"] == 0 and r["mismatch_frac"] <= max(3e-3, 270.0 / r["n"]), f"GI output: {r}"
        oc, oa = op.ircache_ray_counts(); gc, ga = gp.ircache_ray_counts()
        assert oc == gc and abs(oa - ga) <= 0.002 * oa + 4, (oc, oa, gc, ga)
    return exact


def test_ircache_deterministic_mode_parity(gpu, oracle, device):
    """with deferred, canonically ordered updates on BOTH sides the cache meets the 1e-3 bar (a12 / a13 were
    'statistical')."""
    deterministic_frames_on_identical_state(gpu, oracle, device, "cornell", 128, 128)


def test_ircache_deterministic_mode_parity_three_launches(gpu, oracle, device):
    """The same with the three ray passes as three launches on both sides (kj_ircache_set_ray_pass_schedule(SEQUENTIAL) / okj_ircache_set_chain_schedule(0))."""
    deterministic_frames_on_identical_state(gpu, oracle, device, "cornell", 128, 128, schedule="sequential")


def test_ircache_deterministic_free_running(gpu, oracle, device):
    """Six free-running frames (each side consumes its own G-buffer, history and cache) in the deterministic mode: nothing is
    re-synchronised, so last-bit differences may move single lookups to a neighbouring cell; the cache must still agree per cell."""
    import torch
    from kajiya_amd import scenes
    W = H = 128
    desc = scenes.cornell_box()
    op = oracle.OraclePipeline(oracle.OracleScene(desc), W, H, use_ircache=True)
    gp = gpu.GpuPipeline(device, gpu.Scene(device, desc), W, H, use_ircache=True)
    op.ircache_set_deferred(True); gp.ircache_set_deferred(True)
    for fc in _frames(W, H, 6):
        op.frame(fc)
        gp.frame(fc)
    torch.cuda.synchronize()
    get_g = lambda name, dt: gp.ircache_buffer(name, torch.uint8).cpu().numpy().view(dt)
    get_o = lambda name, dt: op.ircache_buffer(name, np.uint8).view(dt)
    # free-running: a flipped slot stays flipped (and feeds the next frames), so the per-frame rate accumulates: 1 % of the records
    assert_cache_parity(cache_state_per_cell(get_g), cache_state_per_cell(get_o), "cornell 128x128, 6 free-running frames", flip_cells=32, flip_tol=1e-2)
    r = P.compare(gp.surface("spatial_filtered_tex", torch.uint8, (-1,)).cpu().numpy(), op.surface("spatial_filtered_tex", np.uint8, (-1,)), "rgba16f")
    print("free-running GI with the deterministic cache:", r)
    assert r["rel_l2"] < 5e-3, r


def _sequential_replay(state, req):
    """What kj_ircache_apply_requests computes by reduction (ircache.hip), stated the slow way: one of the legal outcomes of lookup.hlsl:118-150, 287-301 that
    does not depend on how the records are grouped -- an empty cell is allocated by its lowest-positioned lookup that may allocate (pool entries in cell order);
    an occupied cell's lookups all see the life from before the frame's refreshes (vote iff rank <= life0 / 4; the life ends as the minimum); the vote's winner is
    the voter with the smallest (dart, key), accepted against the count before the frame's votes. `state`: dict of numpy arrays (modified in place); `req`:
    [n, 8] uint32 records {cell, key, bits, dart, proposal x4}."""
    LIFE_PER_RANK, LIFE_RECYCLE, MAX_ENTRIES, OCC, JUST = 4, 0x8000000, 65536, 1, 2
    gm, life, votes, prop, pool, meta, ecell = state["grid_meta"], state["life"], state["reposition_proposal_count"], state["reposition_proposal"], state["pool"], state["meta"], state["entry_cell"]
    alloc_winner, acc = {}, {}
    for i in range(len(req)):
        cell = int(req[i, 0])
        if cell == 0xffffffff:
            continue
        flags = int(gm[cell, 1])
        if not (flags & OCC):
            if int(req[i, 2]) & 0x100:
                continue
            if cell not in alloc_winner or int(req[i, 1]) < int(req[alloc_winner[cell], 1]):
                alloc_winner[cell] = i
        elif not (flags & JUST):
            e = int(gm[cell, 0])
            life0 = int(life[e])
            if life0 >= LIFE_RECYCLE:
                continue
            rank = int(req[i, 2]) & 0xff
            a = acc.setdefault(e, {"rank_min": 0xffffffff, "votes": 0, "win": None})
            a["rank_min"] = min(a["rank_min"], rank)
            if rank <= life0 // LIFE_PER_RANK:
                a["votes"] += 1
                w = (int(req[i, 3]), int(req[i, 1]), i)      # darts are in [0, 1): their bit patterns order like the floats
                if a["win"] is None or w < a["win"]:
                    a["win"] = w
    for e, a in acc.items():
        life[e] = min(int(life[e]), a["rank_min"] * LIFE_PER_RANK)
        if a["votes"]:
            v0 = int(votes[e])
            votes[e] = v0 + a["votes"]
            k = a["win"][2]
            if req[k, 3:4].view(np.float32)[0] <= np.float32(1.0) / (np.float32(v0) + np.float32(1.0)):
                prop[e] = req[k, 4:8]
    alloc0 = int(meta[3])
    for n_new, cell in enumerate(sorted(alloc_winner)):
        alloc_idx = alloc0 + n_new
        if alloc_idx >= MAX_ENTRIES:
            continue
        first = alloc_winner[cell]
        e = int(pool[alloc_idx])
        meta[2] = max(int(meta[2]), e + 1)
        life[e] = (int(req[first, 2]) & 0xff) * LIFE_PER_RANK
        ecell[e] = cell
        gm[cell] = (e, int(gm[cell, 1]) | OCC | JUST)
        prop[e] = req[first, 4:8]
    meta[3] = min(alloc0 + len(alloc_winner), MAX_ENTRIES)


@pytest.mark.parametrize("case", ["random", "one hot cell", "nobody may allocate", "recycled and fresh"])
def test_replay_of_recorded_lookups_equals_the_sequential_recurrence(gpu, device, case):
    """kj_ircache_apply_requests (reduce into a summary + merge) on synthetic record lists against `_sequential_replay`, every touched buffer bit for bit: 60 k records
    over occupied and empty cells; 40 k records on ONE cell (every workgroup's LDS table holds one hot slot); lists whose lookups may not allocate;
    entries past IRC_LIFE_RECYCLE and cells allocated this frame (both left alone)."""
    import torch
    W = H = 96
    gp = gpu.GpuPipeline(device, gpu.Scene(device, T._scenes()["cornell"]), W, H, use_ircache=True)
    gp.ircache_set_deferred(True)
    for fc in _frames(W, H, 4, scene="cornell"):
        gp.dev.frame_begin(fc); gp.render_inputs(fc); gp.reprojection(); gp.gi_frame()
    torch.cuda.synchronize()
    names = {"grid_meta": np.uint32, "life": np.uint32, "reposition_proposal_count": np.uint32, "reposition_proposal": np.uint32, "pool": np.uint32, "meta": np.uint32, "entry_cell": np.uint32}
    state = {n: gp.ircache_buffer(n, torch.uint8).cpu().numpy().view(dt).copy() for n, dt in names.items()}
    state["grid_meta"] = state["grid_meta"].reshape(-1, 2); state["reposition_proposal"] = state["reposition_proposal"].reshape(-1, 4)
    gm = state["grid_meta"]
    occupied = np.nonzero(gm[:, 1] & 1)[0]
    empty = np.nonzero((gm[:, 1] & 1) == 0)[0]
    assert len(occupied) > 50
    rng = np.random.RandomState(11)
    if case == "recycled and fresh":      # a third of the occupied cells' entries recycled, a third of the cells marked just allocated
        sel = rng.permutation(len(occupied))
        for c in occupied[sel[: len(sel) // 3]]:
            state["life"][gm[c, 0]] = 0x8000000 + rng.randint(0, 2)
        for c in occupied[sel[len(sel) // 3: 2 * len(sel) // 3]]:
            gm[c, 1] |= 2
        for n in ("life", "grid_meta"):
            gp.ircache_buffer(n, torch.uint8).copy_(torch.from_numpy(state[n].reshape(-1).view(np.uint8)))
    n = 40_000 if case == "one hot cell" else 60_000
    req = np.zeros((n, 8), np.uint32)
    if case == "one hot cell":
        req[:, 0] = occupied[len(occupied) // 2]
    else:
        pick_empty = rng.uniform(size=n) < 0.15
        req[:, 0] = np.where(pick_empty, empty[rng.randint(0, min(len(empty), 400), size=n)], occupied[rng.randint(0, len(occupied), size=n)])
        req[rng.uniform(size=n) < 0.02, 0] = 0xffffffff      # unused slots inside the list
    req[:, 1] = rng.permutation(n).astype(np.uint32) | (np.uint32(3) << 28)      # distinct positions in the frame
    rank = rng.randint(0, 5, size=n).astype(np.uint32)
    skip = (rank >= 3) | (rng.uniform(size=n) < (1.0 if case == "nobody may allocate" else 0.2))
    req[:, 2] = rank | (skip.astype(np.uint32) << 8)
    req[:, 3] = rng.uniform(size=n).astype(np.float32).view(np.uint32)
    req[:, 4:8] = rng.randint(0, 2**31, size=(n, 4)).astype(np.uint32)
    d_req = torch.from_numpy(req.view(np.int32)).cuda()
    gp.ircache_apply(d_req, n)
    torch.cuda.synchronize()
    _sequential_replay(state, req)
    for name, dt in names.items():
        got = gp.ircache_buffer(name, torch.uint8).cpu().numpy().view(dt)
        want = state[name].reshape(-1)
        if name == "pool":      # new cells take entries pool[alloc0 ..]: the pool itself is not written
            assert np.array_equal(got, want)
            continue
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (case, name, len(bad), bad[:8].tolist(), got[bad[:4]].tolist(), want[bad[:4]].tolist())
