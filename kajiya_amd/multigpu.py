"""Screen-tile split of the rtdgi frame across GPUs (BASELINE north_star: "partition across the
GPUs of one node by screen-space tile with a halo exchange of reservoirs/history over RCCL/xGMI").

One process per GPU. The full-res image is cut into horizontal strips aligned to 16 rows (so 8x8 half-res
tiles never straddle a cut). Every rank keeps full-size surfaces but runs each pass only on its own rows
(`KjRtdgiRenderParams.row_begin/row_end`); between passes the rows a consumer pass can reach are fetched
from their owners:

  * the REPROJECTED GI history is ALL-GATHERED after every rank has reprojected its own strip: the trace pass reads it at the
    hit point's screen position, anywhere on screen (diffuse_trace_common.inc.hlsl:85-107); last frame's denoised GI
    (`rtdgi.temporal2`) and its variance themselves are only read through the motion vectors: halos;
  * everything else is a bounded HALO: motion+4 rows of the five reservoir histories before the temporal
    pass, 51 half-res rows of {reservoir, packed reservoir, radiance} after it (spatial 32 + 16, resolve 3;
    restir_spatial.hlsl:89-97, restir_resolve.hlsl:89-96), 16 / 3 rows between the spatial passes, 2 / 16
    full-res rows around the denoiser stencils.

The other renderers of the lighting frame split the same way (SplitRtdgi.ssgi_frame / shadow_frame / rtr_frame, whose docstrings list the reach of every pass;
lighting_frame strings the whole of BASELINE configs[2] together). One dependence is easy to miss: an EMPTY reservoir's payload is pixel (0, 0), and the passes
that follow payloads do so before they look at the weight -- so row 0 of the payload-followed images travels to every rank (`transfers(..., pin=1)`).

Inputs (G-buffer, depth, reprojection map, sky, BVH) are replicated, as is the world-space irradiance cache: the
replicas record their lookups' side effects, the records of all strips are all-gathered and every rank replays the
same merged list (kj_ircache_set_deferred_updates), so that -- cache bound or not -- every rank's rows are
bit-identical to the single-GPU frame, which tests/test_gpu_multigpu.py checks.

Communication backends: `DistComm` = torch.distributed point-to-point (backend "nccl" = RCCL over xGMI, or
"gloo" on CPU for the tests); `LocalComm` = N virtual ranks inside one process (single-GPU emulation used by
the exactness test).
"""
import ctypes as C
import os

from .abi import KJ_RTDGI_PASS
from . import lib as klib

KEEP = 1 << 31

# surface name -> (bytes per texel, resolution: "h" half / "f" full)
SURF = {
    "rtdgi.reservoir": (8, "h"), "rtdgi.ray_orig": (16, "h"), "rtdgi.ray": (8, "h"), "rtdgi.radiance": (8, "h"), "rtdgi.hit_normal": (8, "h"),
    "rtdgi.invalidity": (4, "h"), "rtdgi.candidate": (8, "h"), "rtdgi.temporal2": (8, "f"), "rtdgi.temporal2_var": (4, "f"),
    "rt_history_validity_pre_input_tex": (1, "h"), "rt_history_validity_input_tex": (1, "h"), "candidate_radiance_tex": (8, "h"),
    "candidate_hit_tex": (8, "h"), "temporal_reservoir_packed_tex": (16, "h"), "reservoir_output_tex0": (8, "h"), "reservoir_output_tex1": (8, "h"),
    "irradiance_output_tex": (8, "f"), "temporal_filtered_tex": (8, "f"), "spatial_filtered_tex": (8, "f"), "reprojected_history_tex": (8, "f"),
}
# TaaRenderer surfaces (all full-res; input extent == output extent in the split path)
TAA_SURF = {"taa": 8, "taa.velocity": 4, "taa.smooth_var": 8, "this_frame_output_img": 8}
# SsgiRenderer surfaces the split exchanges (full-res): the temporal pass' history (R16F) and the finished guide (R8)
SSGI_SURF = {"ssgi": 2, "filtered_output_tex": 1}
# rows of the SSAO guide a rank's rtdgi passes reach beyond its strip: the first spatial pass runs on own +- 64 full-res rows and its taps reach another
# 32 half-res rows (the guide travels inside the half-res G-buffer record extract_half writes), + the half-res subsample offset
GUIDE_HALO = 144
# ShadowDenoiseRenderer surfaces the split exchanges (full-res): the temporal pass' two histories (RGBA16F moments, RG16F accumulated term)
SHADOW_SURF = {"shadow_denoise_moments": 8, "shadow_denoise_accum": 4}
# RtrRenderer's eight ping-pong temporals (rtr.rs:36-52): name -> (bytes per texel, resolution)
RTR_SURF = {"rtr.temporal": (8, "f"), "rtr.ray_len": (4, "f"), "rtr.irradiance": (8, "h"), "rtr.ray_orig": (16, "h"), "rtr.ray": (8, "h"), "rtr.reservoir": (8, "h"),
            "rtr.rng": (4, "h"), "rtr.hit_normal": (8, "h")}
RTR_PASS = {"TRACE": 1, "VALIDATE": 2, "RESTIR_TEMPORAL": 4, "RESOLVE": 8, "TEMPORAL_FILTER": 16, "CLEANUP": 32, "EXTRACT_HALF": 64, "SPECULAR_LIGHTS": 128}


def plan_strips(height, n):
    """Full-res row ranges [(r0, r1)] per rank: 16-aligned cuts, as even as possible."""
    units = (height + 15) // 16
    base, extra = divmod(units, n)
    out, u = [], 0
    for r in range(n):
        cnt = base + (1 if r < extra else 0)
        r0, r1 = min(u * 16, height), min((u + cnt) * 16, height)
        out.append((r0, r1))
        u += cnt
    assert out[-1][1] == height and all(b > a for a, b in out), out
    return out


def rtr_resolve_halo(height, clip_to_view_11):
    """Half-res rows the taps of rtr's resolve can land beyond the rows it runs on, or None (no useful bound: all-gather). resolve.hlsl:296-330: a tap is the
    pixel's reflection-ray origin plus a WORLD-space offset o, |o| <= kernel_size_ws * radius, radius <= 1 (the accumulated radius stays below 8, the
    multiplier is 8^-0.666), kernel_size_ws <= k * z * t after its two clamps (:268-271), with k = max(0.1, 4 / height), z the origin's view depth and
    t = clip_to_view[1][1] = tan(vertical fov / 2). A point at view height Y and depth z sits at NDC y = Y / (z t); moving it by (oy, dz) moves y by
    (oy z - Y dz) / (z (z + dz) t), and with |Y| <= z t (the pixel is on screen) and |(oy, dz)| <= k z t that is at most k sqrt(1 + t^2) / (1 - k t). NDC spans 2
    over height / 2 half-res rows. + 4 rows for what the derivation leaves out (the biased ray origin, the sub-pixel jitter of the projection, rounding).
    Same expression, in double, in csrc/split.cpp (rtr_resolve_halo): both ends of an exchange must agree on it."""
    import math
    t = abs(float(clip_to_view_11))
    k = max(0.1, 4.0 / height)
    if not k * t < 0.5:
        return None
    return int(math.ceil(k * math.sqrt(1.0 + t * t) / (1.0 - k * t) * height / 4.0)) + 4


def max_vertical_motion_rows(reprojection_map, height):
    """Largest |vertical screen motion| of a reprojection map (RGBA16_SNORM as an int16 (H, W, 4) tensor; .y = motion in screen heights), in full-res rows:
    what `motion_halo` has to cover. The split reads histories through these vectors; a frame that moves further than the halo is rendered from rows the
    rank does not hold -- silently. bench.py checks its frames against it before the timed region."""
    return float(reprojection_map[..., 1].abs().max().item()) / 32767.0 * height


def half_rows(r0, r1, height):
    hh = (height + 1) // 2
    return r0 // 2, (hh if r1 == height else r1 // 2)


def transfers(strips, halo, res, height, pin=0):
    """[(src_rank, dst_rank, row0, row1)] in the surface's own resolution so that every rank holds rows
    [own0 - halo, own1 + halo) after the exchange (halo=None: all rows = all-gather). `pin`: every rank also holds the image's first `pin`
    rows (an empty reservoir's payload is pixel (0, 0): rtr's reservoir pass follows it into the histories, rtr_restir_temporal.hlsl:383-392)."""
    n = len(strips)
    own = [half_rows(a, b, height) if res == "h" else (a, b) for a, b in strips]
    total = (height + 1) // 2 if res == "h" else height
    out = []
    for dst in range(n):
        lo = 0 if halo is None else max(0, own[dst][0] - halo)
        hi = total if halo is None else min(total, own[dst][1] + halo)
        spans = [(lo, hi)]
        if pin > 0:
            spans = [(0, max(hi, min(pin, total)))] if lo <= pin else [(0, min(pin, total)), (lo, hi)]
        for lo2, hi2 in spans:
            for src in range(n):
                if src == dst:
                    continue
                a, b = max(lo2, own[src][0]), min(hi2, own[src][1])
                if b > a:
                    out.append((src, dst, a, b))
    return out


class LocalComm:
    """N virtual ranks in one process: transfers are device-to-device copies. For the single-GPU exactness test."""

    def __init__(self, n):
        self.n = n
        self.ranks = list(range(n))

    def run(self, xfers, get_rows):
        self.run_prepared(self.prepare(xfers, get_rows))

    def prepare(self, xfers, get_rows):
        """Resolve a transfer plan to (source rows, destination rows) tensor views once; the views stay valid as long as the surfaces do."""
        return [(get_rows(src, a, b), get_rows(dst, a, b)) for (src, dst, a, b) in xfers]

    def run_prepared(self, pairs):
        # snapshot sources first so that in-place updates of one rank cannot leak into another's copy
        staged = [src.clone() for src, _ in pairs]
        for (_, dst), data in zip(pairs, staged):
            dst.copy_(data)

    def all_gather_fixed(self, parts):
        """parts: {rank: tensor} (the same shape on every rank) -> {rank: [every rank's tensor, in rank order]}. Virtual ranks: nothing moves."""
        every = [parts[r] for r in sorted(parts)]
        return {r: every for r in self.ranks}


class DistComm:
    """torch.distributed point-to-point exchange (NCCL = RCCL on ROCm; gloo for CPU tests).
    `stage_through_host`: debugging aid for the gloo backend with device tensors (several ranks sharing one GPU, where RCCL
    refuses to start): sends are copied to host memory first, receives land in host memory and are copied back."""

    def __init__(self, dist, rank, world, stage_through_host=False, packed=None):
        self.dist, self.rank, self.n = dist, rank, world
        self.ranks = [rank]
        self.stage = stage_through_host
        # packed (opt-in, KJ_SPLIT_PACKED=1): one message per peer and exchange point instead of one per surface — the rows bound for a peer
        # are concatenated into a staging buffer (one torch.cat), sent as one P2P op and scattered back on arrival. Fewer, larger RCCL
        # operations; not measurable in the build environment (no multi-GPU node), hence off by default.
        self.packed = bool(int(os.environ.get("KJ_SPLIT_PACKED", "0"))) if packed is None else packed

    def prepare(self, xfers, get_rows):
        """This rank's share of a transfer plan as [(is_send, rows view, peer)], in plan order (both ends of a pair enumerate the plan
        identically, so sends and receives match up). Cached by the caller: per frame only the P2POp objects are rebuilt."""
        if self.stage:
            return ("staged", xfers, get_rows)
        spec = []
        for (src, dst, a, b) in xfers:
            if src == self.rank:
                spec.append((True, get_rows(src, a, b), dst))
            elif dst == self.rank:
                spec.append((False, get_rows(dst, a, b), src))
        if self.packed and spec:
            import torch
            groups = {}          # (is_send, peer) -> [views] in plan order: the sender's order to a peer is the receiver's order from it
            for is_send, t, peer in spec:
                groups.setdefault((is_send, peer), []).append(t)
            packed = []
            for (is_send, peer), views in sorted(groups.items(), key=lambda kv: (kv[0][1], not kv[0][0])):
                flat = [v.reshape(-1) for v in views]
                buf = torch.empty(sum(f.numel() for f in flat), dtype=views[0].dtype, device=views[0].device)
                packed.append((is_send, peer, buf, views, flat))
            return ("packed", packed)
        return spec

    def run_prepared(self, spec):
        if isinstance(spec, tuple) and spec and spec[0] == "staged":
            return self.run(spec[1], spec[2])
        if isinstance(spec, tuple) and spec and spec[0] == "packed":
            import torch
            d = self.dist
            ops = []
            for is_send, peer, buf, views, flat in spec[1]:
                if is_send:
                    torch.cat(flat, out=buf)
                ops.append(d.P2POp(d.isend if is_send else d.irecv, buf, peer))
            for w in d.batch_isend_irecv(ops):
                w.wait()
            for is_send, peer, buf, views, flat in spec[1]:
                if not is_send:
                    off = 0
                    for v in views:
                        n = v.numel()
                        v.copy_(buf[off:off + n].view(v.shape))
                        off += n
            return
        if not spec:
            return
        d = self.dist
        ops = [d.P2POp(d.isend if is_send else d.irecv, t, peer) for (is_send, t, peer) in spec]
        for w in d.batch_isend_irecv(ops):
            w.wait()

    def all_gather_fixed(self, parts):
        """All-gather of one fixed-size tensor per rank (the irradiance cache's summaries: no counts, nothing the host has to read)."""
        import torch
        d = self.dist
        mine = parts[self.rank]
        dev = mine.device
        stage = self.stage or d.get_backend() == "gloo" and dev.type != "cpu"
        send = (mine.cpu() if stage else mine).contiguous().reshape(-1)
        recv = torch.empty(self.n * send.numel(), dtype=send.dtype, device=send.device)      # (flat: gloo's all-gather takes no leading rank dimension)
        d.all_gather_into_tensor(recv, send)
        if stage:
            recv = recv.to(dev)
        self._keep_gathered = recv
        return {self.rank: [recv[i * send.numel():(i + 1) * send.numel()].reshape(mine.shape) for i in range(self.n)]}

    def run(self, xfers, get_rows):
        ops, landing = [], []
        for (src, dst, a, b) in xfers:
            if src == self.rank:
                t = get_rows(src, a, b)
                ops.append(self.dist.P2POp(self.dist.isend, t.cpu() if self.stage else t, dst))
            elif dst == self.rank:
                t = get_rows(dst, a, b)
                if self.stage:
                    import torch
                    h = torch.empty(t.shape, dtype=t.dtype)
                    landing.append((t, h))
                    t = h
                ops.append(self.dist.P2POp(self.dist.irecv, t, src))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for t, h in landing:
            t.copy_(h)


def _lighting_frame(split, ranks, with_ssgi, specular_lights):
    """BASELINE configs[2] under the split -- the whole lighting frame of world_render_passes.rs:99-291 strip by strip, for either orchestrator: SSAO guide, sun
    shadow mask + denoiser, irradiance cache + rtdgi, reflections (enable_rtr() comes first), the deferred combine on each rank's rows (it reads every input
    at the pixel itself) and TAA on the lit image. Returns {rank: (lit RGBA16F image, valid on the rank's rows)}; the TAA output is TaaRenderer's as usual."""
    if with_ssgi:
        split.ssgi_frame()
    shadows = split.shadow_frame()
    split.gi_frame()
    rtr = split.rtr_frame(specular_lights=specular_lights)
    lit = {}
    for r in ranks:
        a, b = split.strips[r]
        lit[r] = split.pipes[r].light_gbuffer(shadows[r], rtr_ptr=rtr[r].data_ptr(), rows=(a, b))[1]
    split.taa_frame(inputs=lit)
    return lit


def _lighting_frame_pipelined(split, ranks, next_fc, with_ssgi, specular_lights):
    """lighting_frame with every rank's cache replica updated on a side stream, as frame_pipelined does for the GI frame (pipeline_begin(fc) primes it): the
    replay of this frame's recorded cache updates and the NEXT frame's cache maintenance + its three ray passes start when this frame's reflection rays --
    the cache's last readers -- are done, and run under the reservoir pass, resolve, filters, deferred combine and TAA. Every launch of this frame is issued
    before the next frame's constants are uploaded (the host-side order the constants ring needs). Same results as lighting_frame, frame by frame."""
    import torch
    sd = split._side
    i = split.frame & 1
    main = torch.cuda.current_stream()
    main.wait_event(sd["fc"][i])
    if with_ssgi:
        split.ssgi_frame()
    shadows = split.shadow_frame()
    main.wait_event(sd["irc"][i])
    split.gi_frame(ircache_done=True, defer_merge=True)            # (advances split.frame)
    rtr = split.rtr_frame(specular_lights=specular_lights, trace_event=sd["trace"][i], defer_merge=True)
    lit = {}
    for r in ranks:
        a, b = split.strips[r]
        lit[r] = split.pipes[r].light_gbuffer(shadows[r], rtr_ptr=rtr[r].data_ptr(), rows=(a, b))[1]
    split.taa_frame(inputs=lit)
    if next_fc is not None:
        split._enqueue_ircache(next_fc, sd["trace"][i])
    elif split.consistent_ircache:                                 # last frame: nothing follows on the side stream, replay there all the same
        with torch.cuda.stream(sd["stream"]):
            sd["stream"].wait_event(sd["trace"][i])
            split._replay_recorded_cache_updates()
        main.wait_stream(sd["stream"])
    return lit


class SplitRtdgi:
    """Drives RtdgiRenderer::{reproject,render} + TaaRenderer::render strip by strip with halo exchanges.
    `pipes`: {rank: GpuPipeline} for the ranks living in this process (one for DistComm, N for LocalComm).

    Exchange points per GI + TAA frame (each ONE batched send/recv group; round 6 took three away: a history's halo for the NEXT frame travels with the
    exchange that follows the pass which makes it final, and TAA's input halo is over-computed by the spatial filter instead of exchanged):
      A' after reproject all-gather of the reprojected GI history (the trace pass reads it at the hit's screen position, anywhere)
      B  after validate  the five reservoir histories (validate rewrites them in place), invalidity, validity_pre
      C  after trace     validity_in (2), candidate radiance / hit (11)
      D  after temporal  reservoir, packed reservoir, radiance: 64 half-res rows (the "one-deep" exchange of SURVEY 8e-2)
      H  after the temporal filter   16 + 32 full-res rows for the spatial filter's taps (it over-computes +-32 rows: what TAA's first passes read
                         around the strip) and the motion halos of rtdgi.temporal2 (+variance) for next frame's reproject / temporal filter
      T  after TAA       TAA's three histories: motion halos for next frame
      M  the cache's summaries: one fixed-size all-gather (_merge_ircache_requests)
    Between D and H nothing is exchanged: spatial pass 0 is over-computed on +-32 half-res rows, pass 1 on +-16, the resolve on
    +-16 full-res rows, which covers every tap of the next pass (restir_spatial.hlsl:89-97,155-157; restir_resolve.hlsl:89-96;
    temporal_filter.hlsl:69-89). TAA over-computes its intermediates the same way (taa_frame). Outside gi_frame: the SSAO guide (ssgi_frame: its
    history's motion halo, the finished guide's halo) and the sun shadows (shadow_frame: the denoiser's histories' motion halos, the mask's halo)."""

    def __init__(self, comm, pipes, width, height, motion_halo=8):
        self.comm, self.pipes = comm, pipes
        self.W, self.H = width, height
        self.strips = plan_strips(height, comm.n)
        self.motion_halo = motion_halo
        self.frame = 0
        self.taa_frames = 0
        self.ssgi_frames = 0
        self.shadow_frames = 0
        self.rtr_frames = 0
        self.with_rtr = False
        self._views = {}
        self._plans = {}
        self._plan_bytes = {}
        self.exchange_log = None       # set to [] to record (items, {rank: bytes arriving}) per exchange (scripts/split_exchange_bytes.py)
        self._params = {}
        self._s = None
        self._side = None          # side stream state for pipelined ircache work
        self.on_ircache_traced = None
        # every rank keeps a replica of the irradiance cache. With `consistent_ircache` the replicas record their lookups'
        # side effects instead of applying them (kj_ircache_set_deferred_updates), the records of all strips are all-gathered after the
        # trace pass and every rank replays the same merged list: the replicas stay bit-identical (no seams between strips).
        self.consistent_ircache = all(gp.ircache for gp in pipes.values())
        for gp in pipes.values():
            if gp.ircache:
                gp.ircache_set_deferred(self.consistent_ircache)

    # -- helpers
    def _surface(self, rank, name):
        key = (rank, name)
        t = self._views.get(key)
        if t is None:
            import torch
            gp = self.pipes[rank]
            if name.startswith("TAA/"):
                t = gp.taa_surface(name[4:], torch.uint8, (self.H, self.W * TAA_SURF[name[4:].split(":")[0]]))
            elif name.startswith("SSGI/"):
                t = gp.ssgi_surface(name[5:], torch.uint8, (self.H, self.W * SSGI_SURF[name[5:].split(":")[0]]))
            elif name == "SHADOW/mask":
                return gp.shadow_mask_img            # the caller's image (shadow_frame): not cached, it may change between frames
            elif name == "LIT/input":
                return gp.taa_input_img.view(torch.uint8).view(self.H, self.W * 8)      # the caller's image (taa_frame(inputs=...))
            elif name.startswith("SHADOW/"):
                t = gp.shadow_denoise_surface(name[7:], torch.uint8, (self.H, self.W * SHADOW_SURF[name[7:].split(":")[0]]))
            elif name.startswith("RTR/"):
                bpt, res = RTR_SURF[name[4:].split(":")[0]]
                t = gp.rtr_surface(name[4:], torch.uint8, ((self.H + 1) // 2, (self.W + 1) // 2 * bpt) if res == "h" else (self.H, self.W * bpt))
            else:
                bpt, res = SURF[name.split(":")[0]]
                w = (self.W + 1) // 2 if res == "h" else self.W
                h = (self.H + 1) // 2 if res == "h" else self.H
                t = gp.surface(name, torch.uint8, (h, w * bpt))
            self._views[key] = t      # renderer surfaces are allocated once per extent: the pointer is stable
        return t

    def _rows_view(self, rank, name, a, b):
        return self._surface(rank, name)[a:b]

    def _exchange(self, items):
        """items: [(surface name, halo rows or None[, pinned top rows])] -- ONE batched exchange for all of them (a single RCCL group:
        both ends enumerate (item, dst, src) in the same order, so per-pair send/recv order matches)."""
        key = tuple(items)
        prepared = self._plans.get(key)
        if prepared is None:
            xfers = []
            for item in items:
                name, halo, pin = item if len(item) == 3 else (item[0], item[1], 0)
                res = RTR_SURF[name[4:].split(":")[0]][1] if name.startswith("RTR/") else "f" if name.startswith(("TAA/", "SSGI/", "SHADOW/", "LIT/")) else SURF[name.split(":")[0]][1]
                xfers += [(src, dst, (name, a), b) for (src, dst, a, b) in transfers(self.strips, halo, res, self.H, pin)]
            # renderer surfaces keep their address for a given extent, so the row views can be resolved once per distinct item list
            # (two per exchange point: the ping-pong suffixes alternate)
            prepared = self.comm.prepare(xfers, lambda r, na, b: self._rows_view(r, na[0], na[1], b))
            self._plans[key] = prepared
            into = {}                 # bytes arriving at each rank in this exchange: rows x the surface's row bytes (any rank can tell for all of them)
            for (src, dst, (name, a), b) in xfers:
                into[dst] = into.get(dst, 0) + (b - a) * self._row_bytes(name)
            self._plan_bytes[key] = into
        self.comm.run_prepared(prepared)
        if self.exchange_log is not None:
            self.exchange_log.append((key, self._plan_bytes.get(key, {})))

    def _row_bytes(self, name):
        base = name.split(":")[0]
        hw = (self.W + 1) // 2
        if base.startswith("RTR/"):
            bpt, res = RTR_SURF[base[4:]]
            return (hw if res == "h" else self.W) * bpt
        if base.startswith("TAA/"):
            return self.W * TAA_SURF[base[4:]]
        if base.startswith("SSGI/"):
            return self.W * SSGI_SURF[base[5:]]
        if base == "SHADOW/mask":
            return self.W
        if base.startswith("SHADOW/"):
            return self.W * SHADOW_SURF[base[7:]]
        if base == "LIT/input":
            return self.W * 8
        bpt, res = SURF[base]
        return (hw if res == "h" else self.W) * bpt

    def _grow(self, rank, rows):
        r0, r1 = self.strips[rank]
        return max(0, r0 - rows), min(self.H, r1 + rows)

    def self_test(self, device=None):
        """Start-up check of the transport, before frame 0: every kind of exchange the frame schedule uses -- the all-gather of a
        full-res image, motion halos, the 64-row one-deep halo of half-res records, full-res stencil halos, the fixed-size
        all-gather of the cache's summaries -- runs once on scratch images whose rows carry their OWNER's rank, through the same
        `comm.prepare` / `run_prepared` / `all_gather_fixed` code (incl. the packed mode), and each rank then checks on the device that
        every row it is entitled to holds the owner's pattern. Returns True when every rank passed (collective); the first N > 1 run on
        real hardware certifies its own communicator this way (bench.py prints "RCCL <n> ranks OK")."""
        import torch
        n = self.comm.n
        dev = device or next(iter(self.pipes.values())).depth.device
        M = self.motion_halo
        plans = [("all-gather, full res, 8 B", "f", 8, None, 0), ("motion halo + row 0 for everybody, half res, 16 B", "h", 16, M + 4, 1), ("one-deep halo, half res, 16 B", "h", 16, 64, 0),
                 ("stencil halo, full res, 8 B", "f", 8, 16, 0), ("TAA input halo, full res, 8 B", "f", 8, 25, 0), ("validity halo, half res, 1 B", "h", 1, M + 1, 0)]
        ok = True
        owner_of = {}
        for res in ("h", "f"):
            total = (self.H + 1) // 2 if res == "h" else self.H
            o = torch.zeros(total, dtype=torch.uint8)
            for r, (a, b) in enumerate(self.strips):
                a2, b2 = half_rows(a, b, self.H) if res == "h" else (a, b)
                o[a2:b2] = r + 1
            owner_of[res] = o.to(dev)
        for k, (what, res, bpt, halo, pin) in enumerate(plans):
            h = (self.H + 1) // 2 if res == "h" else self.H
            w = ((self.W + 1) // 2 if res == "h" else self.W) * bpt
            scratch = {}
            for r in self.comm.ranks:
                t = torch.zeros((h, w), dtype=torch.uint8, device=dev)
                a, b = self.strips[r]
                a2, b2 = half_rows(a, b, self.H) if res == "h" else (a, b)
                t[a2:b2] = (r + 1) * 8 + k            # the owner's pattern; every other row stays 0 until the exchange fills it
                scratch[r] = t
            xfers = [(src, dst, (k, a), b) for (src, dst, a, b) in transfers(self.strips, halo, res, self.H, pin)]
            self.comm.run_prepared(self.comm.prepare(xfers, lambda r, na, b: scratch[r][na[1]:b]))
            for r in self.comm.ranks:
                a, b = self.strips[r]
                a2, b2 = half_rows(a, b, self.H) if res == "h" else (a, b)
                lo, hi = (0, h) if halo is None else (max(0, a2 - halo), min(h, b2 + halo))
                expect = (owner_of[res][lo:hi].to(torch.int32) * 8 + k).to(torch.uint8)
                good = bool((scratch[r][lo:hi] == expect[:, None]).all().item())
                p0 = min(pin, lo)                      # the pinned top rows, where they are not inside [lo, hi) anyway
                good = good and bool((scratch[r][:p0] == (owner_of[res][:p0].to(torch.int32) * 8 + k).to(torch.uint8)[:, None]).all().item())
                outside = bool((scratch[r][p0:lo] == 0).all().item()) and bool((scratch[r][hi:] == 0).all().item())
                if not (good and outside):
                    ok = False
                    import sys
                    print(f"[kajiya_amd split self-test] rank {r}: exchange '{what}' delivered wrong rows", file=sys.stderr, flush=True)
        # the fixed-size all-gather of the cache's summaries: rank r contributes 4 KB of value r + 1
        parts = {r: torch.full((4096,), r + 1, dtype=torch.uint8, device=dev) for r in self.comm.ranks}
        got = self.comm.all_gather_fixed(parts)
        for r in self.comm.ranks:
            if len(got[r]) != n or not all(bool((got[r][q] == q + 1).all().item()) for q in range(n)):
                ok = False
        if isinstance(self.comm, DistComm):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if (self.comm.stage or self.comm.dist.get_backend() == "gloo") else dev)
            self.comm.dist.all_reduce(flag, op=self.comm.dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        return ok

    def _render(self, rank, mask, rows=None, spatial_select=0):
        """One kj_rtdgi_render call. The parameter struct is built once per rank per frame (gi_frame) and only the pass mask, row range
        and spatial-pass selector change between calls; the stream handle is looked up once per frame as well (host time per rank per
        frame matters once the strips are small: scripts/split_host_overhead.py)."""
        gp = self.pipes[rank]
        p = self._params[rank]
        p.pass_mask = mask
        p.row_begin, p.row_end = rows if rows is not None else (0, 0)
        p.spatial_pass_select = spatial_select
        st = gp.L.kj_rtdgi_render(gp.rtdgi, C.byref(p), C.byref(gp.out), self._s)
        if st != 0:
            klib.check(st)

    def _ircache_head(self, gp, s):
        if self.consistent_ircache:
            gp.ircache_begin_requests(half_rows=half_rows(*self.strips[next(r for r in self.comm.ranks if self.pipes[r] is gp)], self.H))
        klib.check(gp.L.kj_ircache_prepare(gp.ircache, s))
        klib.check(gp.L.kj_ircache_trace_irradiance(gp.ircache, gp.scene.h, gp.sky16.data_ptr(), 16, s))

    # -- frame pipelining (see GpuPipeline.frame_pipelined): each rank's replica of the ircache is updated on a side stream
    def pipeline_begin(self, fc):
        import torch
        self._side = {"stream": torch.cuda.Stream(), "irc": [torch.cuda.Event(), torch.cuda.Event()], "trace": [torch.cuda.Event(), torch.cuda.Event()],
                      "fc": [torch.cuda.Event(), torch.cuda.Event()]}
        self._enqueue_ircache(fc, None)

    def _enqueue_ircache(self, fc, wait_event):
        import torch
        sd = self._side
        s0 = torch.cuda.current_stream()
        with torch.cuda.stream(sd["stream"]):
            sd["stream"].wait_stream(s0) if wait_event is None else sd["stream"].wait_event(wait_event)
            if self.consistent_ircache and wait_event is not None:
                self._merge_ircache_requests()   # last frame's recorded cache updates (its ray passes are behind wait_event)
            first = True
            for r in self.comm.ranks:
                gp = self.pipes[r]
                if first:
                    gp.dev.frame_begin(fc)     # one device (and one constants ring) per process
                    sd["fc"][self.frame & 1].record(sd["stream"])
                    first = False
                if gp.ircache:
                    self._ircache_head(gp, klib._stream_ptr())
            if self.on_ircache_traced is not None:
                self.on_ircache_traced()
            sd["irc"][self.frame & 1].record(sd["stream"])

    def frame_pipelined(self, next_fc, run_ssgi=False):
        """gi_frame + taa_frame with the ircache work issued ahead on the side stream; then issue the next frame's.
        `run_ssgi`: every rank's SSAO guide first, ordered behind the side stream's write of this frame's constants."""
        import torch
        i = self.frame & 1
        if run_ssgi:
            torch.cuda.current_stream().wait_event(self._side["fc"][i])
            self.ssgi_frame()
        torch.cuda.current_stream().wait_event(self._side["irc"][i])
        self.gi_frame(ircache_done=True, trace_event=self._side["trace"][i], defer_merge=True)
        self.taa_frame()
        if next_fc is not None:
            self._enqueue_ircache(next_fc, self._side["trace"][i])
        elif self.consistent_ircache:             # last frame: nothing follows on the side stream, replay there all the same
            sd = self._side
            with torch.cuda.stream(sd["stream"]):
                sd["stream"].wait_event(sd["trace"][i])
                self._merge_ircache_requests()
            torch.cuda.current_stream().wait_stream(sd["stream"])

    def ssgi_frame(self):
        """SsgiRenderer::render strip by strip (kj_ssgi_render_rows), before gi_frame: every rank computes the SSAO guide for its own rows -- the
        intermediate passes over-compute what the next pass reaches into -- after the halo of the temporal pass' history has arrived, then the
        finished guide's halo is exchanged: rtdgi's passes read it up to GUIDE_HALO rows beyond the strip (gi_frame runs extract_half on those
        rows). Round 2 computed the whole frame's guide on every rank: 0.25 ms of replicated work per rank at 4K."""
        M = self.motion_halo
        for r in self.comm.ranks:
            self.pipes[r].ssgi_frame(rows=self.strips[r])
        # ONE exchange (round 6): the finished guide's halo and, for next frame's temporal pass, the halo of the history this frame has just written
        self._exchange([(f"SSGI/filtered_output_tex:{self.ssgi_frames % 2}", GUIDE_HALO + 2), (f"SSGI/ssgi:{self.ssgi_frames % 2}", M + 2)])
        self.ssgi_frames += 1

    def shadow_frame(self, masks=None, ray_counters=None):
        """trace_sun_shadow_mask + ShadowDenoiseRenderer::render strip by strip (world_render_passes.rs:124-136): the halo of the denoiser's two
        histories (its temporal pass reads them through the motion vectors and over-computes 24 rows either side), every rank's rays for its OWN
        rows, the mask's 32-row halo (one byte per pixel -- cheaper than tracing the neighbours' rays again), then the denoiser's passes, each
        over-computing what the next one reaches into (kj_shadow_denoise_render_rows). Returns {rank: RG16F image valid on the rank's own rows} --
        all light_gbuffer(rows=strip) reads. `masks`: {rank: uint8 (H, W) image}; made on first use otherwise."""
        import torch
        M = self.motion_halo
        R = self.comm.ranks
        for r in R:
            gp = self.pipes[r]
            gp.shadow_denoiser()
            if masks is not None:
                gp.shadow_mask_img = masks[r]
            elif getattr(gp, "shadow_mask_img", None) is None:
                gp.shadow_mask_img = torch.zeros((self.H, self.W), dtype=torch.uint8, device=gp.depth.device)
        if self.shadow_frames > 0:
            h = f":{1 - self.shadow_frames % 2}"
            self._exchange([("SHADOW/shadow_denoise_moments" + h, M + 3 + 24), ("SHADOW/shadow_denoise_accum" + h, M + 3 + 24)])
        for r in R:
            self.pipes[r].sun_shadow_mask(out=self.pipes[r].shadow_mask_img, ray_counter=ray_counters[r] if ray_counters else None, rows=self.strips[r])
        if masks is not None:
            self._plans.pop((("SHADOW/mask", 32),), None)       # the caller's image may be another one than last frame's: resolve its rows anew
        self._exchange([("SHADOW/mask", 32)])
        out = {r: self.pipes[r].shadow_denoise(self.pipes[r].shadow_mask_img, rows=self.strips[r]) for r in R}
        self.shadow_frames += 1
        return out

    def enable_rtr(self, tables=None):
        """Reflections join the frame (rtr_frame after gi_frame): every rank gets its RtrRenderer, the caches reserve slot ranges for the lookups of rtr's
        rays (kj_ircache_set_rtr_requests), and the replay of a frame's recorded cache updates moves behind rtr's ray passes."""
        self.with_rtr = True
        for gp in self.pipes.values():
            gp.rtr_handle(tables)
            if gp.ircache and self.consistent_ircache:
                gp.ircache_set_rtr_requests(True)

    def rtr_frame(self, specular_lights=False, trace_event=None, defer_merge=False):
        """RtrRenderer::trace + LightingRenderer::render_specular + TracedRtr::filter_temporal strip by strip (kj_rtr_render_rows), after gi_frame
        (world_render_passes.rs:172-210). Reach of the passes and what travels for it:
          X1  before the ray passes   ALL-GATHER of this frame's GI image: a reflection ray's hit reads it at the hit's screen position, anywhere
                                      (reflection_trace_common.inc.hlsl:159-166). Validate rewrites the reservoir histories of its own quads in place.
          X2  after the ray passes    the six reservoir histories: motion + the search's 1 + the taps' 14 half-res rows (rtr_restir_temporal.hlsl:
                                      232-262,383-390), and row 0 for everybody -- an empty reservoir's payload is pixel (0, 0), and the pass follows it
          X3  after the reservoir pass  {irradiance, ray, reservoir, ray origin}: the resolve's taps land where a WORLD-space kernel projects to
                                      (resolve.hlsl:296-330: up to a tenth of the frustum's height at the surface's depth, times perspective) -- a halo
                                      sized from the field of view (rtr_resolve_halo: 8 + 0.12 x height / 4 + 4 half-res rows at 52 degrees), and row 0
                                      (an empty reservoir's payload; a tap whose kernel basis degenerates at normal incidence);
                                      the ray-length history's motion halo; 8 half-res rows of the hit image the trace pass just wrote (the resolve reads
                                      the pixel's own hit distance from it, :112). The resolve (and the lights' specular) over-computes 16 rows either side: the
                                      temporal filter's 3x3 moments read them.
          X4  after the temporal filter  ALL-GATHER of its output: the cleanup's taps reach 24 rows, and next frame's filter reads this image at the
                                      reflection's reprojected virtual position -- anywhere (temporal_filter.hlsl:60-84).
        Each rank's resolved image (what light_gbuffer reads, at the pixel itself) is valid on its own rows. Returns {rank: int32 (H, W) view}."""
        assert self.with_rtr, "enable_rtr() first"
        M, R = self.motion_halo, self.comm.ranks
        k = self.rtr_frames
        o, h = f":{k % 2}", f":{1 - k % 2}"
        s = klib._stream_ptr()
        P = RTR_PASS
        params = {r: self.pipes[r].rtr_params(0) for r in R}

        def run(r, mask, rows):
            p = params[r]
            p.pass_mask = mask
            klib.check(self.pipes[r].L.kj_rtr_render_rows(self.pipes[r].rtr, C.byref(p), rows[0], rows[1], None, s))
        self._exchange([("spatial_filtered_tex", None)])
        for r in R:
            run(r, P["EXTRACT_HALF"] | P["VALIDATE"] | P["TRACE"], self.strips[r])
        if self.consistent_ircache and not defer_merge:
            self._merge_ircache_requests()
        if trace_event is not None:
            import torch
            trace_event.record(torch.cuda.current_stream())
        if k > 0:
            self._exchange([(f"RTR/{n}{h}", M + 16, 1) for n in ("rtr.irradiance", "rtr.ray_orig", "rtr.ray", "rtr.rng", "rtr.reservoir", "rtr.hit_normal")])
        for r in R:
            run(r, P["RESTIR_TEMPORAL"] | KEEP, self.strips[r])
        reach = rtr_resolve_halo(self.H, self.pipes[R[0]].dev.clip_to_view_11)
        x3 = None if reach is None else 8 + reach         # (the resolve runs on own +- 16 full-res rows)
        self._exchange([(f"RTR/{n}{o}", x3, 1) for n in ("rtr.irradiance", "rtr.ray", "rtr.reservoir", "rtr.ray_orig")] + [("candidate_hit_tex", 8)] + ([(f"RTR/rtr.ray_len{h}", M + 2 + 16)] if k > 0 else []))
        for r in R:
            run(r, P["RESOLVE"] | KEEP | (P["SPECULAR_LIGHTS"] if specular_lights else 0), self._grow(r, 16))
            run(r, P["TEMPORAL_FILTER"] | KEEP, self.strips[r])
        self._exchange([(f"RTR/rtr.temporal{o}", None)])
        for r in R:
            run(r, P["CLEANUP"] | KEEP, self.strips[r])
        self.rtr_frames += 1
        import torch
        return {r: self.pipes[r].rtr_surface("resolved_tex", torch.int32, (self.H, self.W)) for r in R}

    def gi_frame(self, ircache_done=False, trace_event=None, defer_merge=False):
        """One rtdgi frame. Unless `ircache_done`, each rank's ircache.prepare + trace_irradiance run here first (serial order).
        `defer_merge`: leave the replay of the cache's recorded updates to the caller (frame_pipelined runs it on the side stream
        once the whole frame is enqueued: its host syncs then wait for the ray passes while the main stream still has work queued)."""
        P = KJ_RTDGI_PASS
        out_sfx, hist_sfx = f":{self.frame % 2}", f":{1 - self.frame % 2}"
        M = self.motion_halo
        R = self.comm.ranks
        self._s = klib._stream_ptr()
        self._params = {r: self.pipes[r].params(0) for r in R}
        # ---- A (gone since round 6): last frame's denoised GI and its variance are read through the motion vectors only (fullres_reproject: a 4x4 footprint
        # around the reprojected pixel; temporal_filter: a bilinear tap): halos, like TAA's histories. They are final once their frame's temporal filter / taa pass has
        # run, so they travel THEN -- with exchange H of that frame, with taa_frame's closing exchange -- instead of in an exchange point of their own here. What the
        # trace pass reads ANYWHERE on screen is the REPROJECTED image, all-gathered below after every rank has reprojected its own strip.
        for r in R:
            gp = self.pipes[r]
            s = self._s
            if gp.ircache and not ircache_done:
                self._ircache_head(gp, s)
            klib.check(gp.L.kj_rtdgi_reproject_rows(gp.rtdgi, gp.reprojection_map_ptr, self.W, self.H, self.strips[r][0], self.strips[r][1], s))
        self._exchange([("reprojected_history_tex", None)])
        for r in R:
            gp = self.pipes[r]
            s = self._s
            if gp.ircache:
                klib.check(gp.L.kj_ircache_sum_up_irradiance_for_sampling(gp.ircache, s))
            # the half-res images and G-buffer records for the WHOLE frame (replicated inputs, cheap): the resolve reads the view normal at every
            # reservoir's sample pixel, an empty reservoir's payload is pixel (0, 0), and the second spatial pass' occlusion march reads the half-res depth
            # along a screen-space ray up to three times as long as the sample pixel is far (round 6 tried own +- 64 half-res rows + row 0: the wide-angle
            # reflections case of the bit-exactness tests fails). With the strip-wise guide (self.ssgi_frame) the records' ssao byte is only meaningful
            # on own +- GUIDE_HALO rows, which is where it is read.
            self._render(r, P["EXTRACT_HALF"])
            self._render(r, P["VALIDATE"] | KEEP, self.strips[r])
        # ---- B
        items = [("rt_history_validity_pre_input_tex", M + 1)]
        if self.frame > 0:
            # (row 0 for everybody: a reservoir nothing was ever selected into -- a validation frame without an accepted history tap -- keeps payload 0,
            # pixel (0, 0), and the temporal pass follows the payload into the four sample images whatever the reservoir's weight: restir_temporal.hlsl:262-275)
            items += [("rtdgi.reservoir" + hist_sfx, M + 4)] + [(n + hist_sfx, M + 4, 1) for n in ("rtdgi.ray_orig", "rtdgi.ray", "rtdgi.radiance", "rtdgi.hit_normal")]
            items += [("rtdgi.invalidity" + hist_sfx, M + 8)]
        self._exchange(items)
        for r in R:
            self._render(r, P["TRACE"] | KEEP, self.strips[r])
        if self.consistent_ircache and not defer_merge and not self.with_rtr:      # (with reflections in the frame the replay follows THEIR ray passes: rtr_frame)
            self._merge_ircache_requests()
        if trace_event is not None:
            import torch
            trace_event.record(torch.cuda.current_stream())
        # ---- C
        self._exchange([("rt_history_validity_input_tex", 2), ("candidate_radiance_tex", 8 + 3), ("candidate_hit_tex", 8 + 3)])
        for r in R:
            self._render(r, P["VALIDITY_INTEGRATE"] | KEEP, self.strips[r])
            self._render(r, P["RESTIR_TEMPORAL"] | KEEP, self.strips[r])
        # ---- D: the one-deep halo. Reach (half-res rows): pass 0 runs on own+-32 and taps +-32; pass 1 runs on own+-16, taps +-16
        # and follows payloads another +-32; the resolve runs on own+-8, taps +-3 and follows payloads +-48.
        # (and row 0: the payload of an empty reservoir is pixel (0, 0) here too)
        self._exchange([("rtdgi.reservoir" + out_sfx, 64, 1), ("temporal_reservoir_packed_tex", 64, 1), ("rtdgi.radiance" + out_sfx, 64, 1)])
        for r in R:
            self._render(r, P["RESTIR_SPATIAL"] | KEEP, self._grow(r, 64), spatial_select=1)
            self._render(r, P["RESTIR_SPATIAL"] | KEEP, self._grow(r, 32), spatial_select=2)
            self._render(r, P["RESTIR_RESOLVE"] | KEEP, self._grow(r, 16))
            self._render(r, P["TEMPORAL_FILTER"] | KEEP, self.strips[r])
        # ---- H: the spatial filter's 16-row reach + the 32 rows it over-computes either side for TAA's first passes (their input halo is 25 rows: no exchange
        # point of its own), and next frame's history halos of the temporal filter's two outputs
        self._exchange([("temporal_filtered_tex", 16 + 32), ("rtdgi.temporal2" + out_sfx, M + 3), ("rtdgi.temporal2_var" + out_sfx, M + 2)])
        for r in R:
            self._render(r, P["SPATIAL_FILTER"] | KEEP, self._grow(r, 32))
        self.frame += 1

    def _merge_ircache_requests(self):
        """This frame's recorded cache updates: every rank reduces its strip's rtdgi lookups (validate and trace pass) -- and, with reflections in the
        frame, rtr's (validate and trace rays); contiguous slots: rows of the half-res image -- into a fixed-size summary, the summaries are all-gathered
        and every rank merges the same ones plus the summary of the cache's own ray passes (replicated: identical on every rank, it stays local).
        No list lengths, nothing read back (include/kajiya_amd.h: kj_ircache_summarize_requests / kj_ircache_apply_summaries)."""
        hw = (self.W + 1) // 2
        strip_sums, own_sums = {}, {}
        for r in self.comm.ranks:
            gp = self.pipes[r]
            first, count = gp.ircache_request_ranges()
            h0, h1 = half_rows(*self.strips[r], self.H)
            bases = [first[0], first[1]] + (gp.ircache_rtr_request_ranges()[0] if self.with_rtr else [])
            strip_sums[r] = gp.ircache_summarize([(b + h0 * hw, (h1 - h0) * hw) for b in bases], which=0)
            own_sums[r] = gp.ircache_summarize([(first[2], count[2]), (first[3], count[3])], which=1)
        gathered = self.comm.all_gather_fixed(strip_sums)
        for r in self.comm.ranks:
            self.pipes[r].ircache_apply_summaries(list(gathered[r]) + [own_sums[r]])

    def taa_frame(self, inputs=None):
        """TaaRenderer::render on this frame's GI image -- or on `inputs`: {rank: RGBA16F (H, W, 4) image valid on the rank's own rows}, e.g. the lit image of
        light_gbuffer(rows=strip) (lighting_frame) -- strip by strip. Exchanges: a caller's image's halo first (the GI image arrives with its halo), the three
        histories' halos for the NEXT frame at the end; the intermediate images are over-computed on up to 32 extra rows per side (8-row tile
        granularity) instead of being exchanged: prob_filter2 reaches +-4 rows of prob_filter, that +-1 of input_prob,
        that +-1 of the filtered history / input and +-2 of the input deviation, those +-1 of the reprojected history /
        input (taa/*.hlsl)."""
        import torch
        gi_out = "spatial_filtered_tex" if inputs is None else "LIT/input"
        stream = klib._stream_ptr()
        if inputs is not None:
            for r in self.comm.ranks:
                self.pipes[r].taa_input_img = inputs[r]
            self._plans.pop(((gi_out, 1 + 24),), None)       # the caller's images may be others than last frame's: resolve their rows anew
        # ---- I: a caller's image needs its halo (filter_input runs on +-24 rows); the GI image has it already -- gi_frame's spatial filter over-computes 32 rows either side
        if inputs is not None:
            self._exchange([(gi_out, 1 + 24)])
        for r in self.comm.ranks:
            gp = self.pipes[r]
            r0, r1 = self.strips[r]
            inp = self._surface(r, gi_out).data_ptr()
            depth_ptr = gp.depth.data_ptr()

            def run(mask, grow, keep=True):
                a, b = max(0, r0 - grow), min(self.H, r1 + grow)
                klib.check(gp.L.kj_taa_render_rows(gp.taa, inp, self.W, self.H, gp.reprojection_map_ptr, depth_ptr, self.W, self.H,
                                                   C.byref(gp.taa_out), stream, mask | (KEEP if keep else 0), a, b))
            run(1, 32, keep=False)         # reproject history (5-tap Catmull-Rom around uv + motion)
            run(2 | 4, 24)                 # filter input (+-1 input), filter history (+-1 reprojected history)
            run(8, 16)                     # input prob (+-2 deviation, +-1 filtered input / history)
            run(16, 8)                     # prob filter (+-1)
            run(32 | 64, 0)                # prob filter 2 (+-4), taa (+-2 reprojected history, +-1 input)
        # next frame's history halos (read through the motion vectors by reproject / input_prob / taa), sent now that they are final
        M, to = self.motion_halo, f":{self.taa_frames % 2}"
        self._exchange([("TAA/taa" + to, M + 4 + 32), ("TAA/taa.velocity" + to, M + 2 + 16), ("TAA/taa.smooth_var" + to, M + 2 + 16)])
        self.taa_frames += 1

    def lighting_frame(self, with_ssgi=True, specular_lights=False):
        return _lighting_frame(self, self.comm.ranks, with_ssgi, specular_lights)

    def lighting_frame_pipelined(self, next_fc, with_ssgi=True, specular_lights=False):
        return _lighting_frame_pipelined(self, self.comm.ranks, next_fc, with_ssgi, specular_lights)

    def _replay_recorded_cache_updates(self):
        self._merge_ircache_requests()

    def gather_output(self, name="spatial_filtered_tex"):
        """Assemble the full image from every rank's own rows (result collection; not part of the timed frame)."""
        self._exchange([(name, None)])


class NativeSplit:
    """The same frame schedule driven by the compiled orchestrator (csrc/split.cpp: KjSplit) instead of SplitRtdgi: one C-ABI call per
    GI frame and one per TAA frame; the strip planning, the pass-by-pass kj_rtdgi_render / kj_taa_render_rows calls, the exchanges
    (one packed message per peer and exchange point over RCCL, or device-to-device copies between virtual ranks) and the merged replay
    of the irradiance cache's recorded updates all happen on the other side of the boundary. SplitRtdgi stays the reference it is
    tested against (tests/test_gpu_multigpu.py::test_native_split_matches_the_python_orchestrator).

    `pipes`: {rank: GpuPipeline} for the ranks living in this process -- all `world` of them (virtual ranks on one device), or one
    together with `nccl_comm` (an ncclComm_t as an integer: NativeSplit.rccl_comm_from_torch makes one)."""

    def __init__(self, world, pipes, width, height, motion_halo=8, nccl_comm=None, own_comm=False):
        """`own_comm`: close() -- an explicit call, never the garbage collector -- also destroys `nccl_comm`. Off by default: a communicator the caller made and may keep
        using is the caller's; bench.py, which makes one with rccl_comm_from_torch for this object alone, passes True."""
        from .abi import KjSplitRank, KjSplitFrame
        self.L = klib.load()
        self._own_comm = nccl_comm if (own_comm and nccl_comm) else None
        self.pipes = pipes
        self.ranks = sorted(pipes)
        self.W, self.H = width, height
        arr = (KjSplitRank * len(self.ranks))()
        for i, r in enumerate(self.ranks):
            gp = pipes[r]
            arr[i].rtdgi, arr[i].taa, arr[i].ircache, arr[i].scene = gp.rtdgi, gp.taa, gp.ircache, gp.scene.h
        self.h = C.c_void_p()
        klib.check(self.L.kj_split_create(world, self.ranks[0], len(self.ranks), arr, width, height, motion_halo, nccl_comm, C.byref(self.h)))
        self._frames = (KjSplitFrame * len(self.ranks))()
        self.frame = 0
        self.on_ircache_traced = None
        self.consistent_ircache = all(gp.ircache for gp in pipes.values())
        for gp in pipes.values():
            gp.ircache_deferred = bool(gp.ircache) and self.consistent_ircache
        self.with_rtr = False
        self.strips = [self.strip(r) for r in range(world)]

    def close(self):
        """Destroys the orchestrator NOW (kj_split_destroy hands the caches back in the mode they came in) and its RCCL communicator, if this
        object made one. A caller that falls back to another orchestrator must not leave this to the garbage collector: a late destroy would
        flip the caches' update mode under the other one's feet."""
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.kj_split_destroy(h)
        comm, self._own_comm = getattr(self, "_own_comm", None), None
        if comm:
            try:
                self.L.kj_split_rccl_comm_destroy(comm)
            except Exception:
                pass

    def __del__(self):
        # the orchestrator handle only: ncclCommDestroy from a finaliser can block on peers at interpreter shutdown, and the communicator may not be ours
        self._own_comm = None
        try:
            self.close()
        except Exception:
            pass

    def strip(self, rank):
        a, b = C.c_uint32(), C.c_uint32()
        klib.check(self.L.kj_split_strip(self.h, rank, C.byref(a), C.byref(b)))
        return a.value, b.value

    def _fill(self):
        for i, r in enumerate(self.ranks):
            gp = self.pipes[r]
            f = self._frames[i]
            f.rtdgi = gp.params(0)
            f.rtdgi_out = C.pointer(gp.out)
            f.taa_out = C.pointer(gp.taa_out)
            f.sky_cube16 = gp.sky16.data_ptr()

    def gi_frame(self, ircache_done=False, trace_event=None, defer_merge=False):
        self._fill()
        handle = None
        if trace_event is not None:
            try:                           # a torch.cuda.Event's hipEvent_t exists once it has been recorded
                trace_event.record()
                handle = int(trace_event.cuda_event)
            except Exception:
                handle = None
        klib.check(self.L.kj_split_gi_frame(self.h, self._frames, int(ircache_done) | (2 if defer_merge else 0), handle, klib._stream_ptr()))
        if trace_event is not None and handle is None:
            trace_event.record()           # no native handle (the tests' CPU stand-in): recorded after the frame instead of mid-frame
        self.frame += 1

    # -- frame pipelining, as SplitRtdgi: every rank's cache replica is updated on a side stream under the previous frame's tail
    def pipeline_begin(self, fc):
        import torch
        self._side = {"stream": torch.cuda.Stream(), "irc": [torch.cuda.Event(), torch.cuda.Event()], "trace": [torch.cuda.Event(), torch.cuda.Event()],
                      "fc": [torch.cuda.Event(), torch.cuda.Event()]}
        self._enqueue_ircache(fc, None)

    def _enqueue_ircache(self, fc, wait_event):
        import torch
        sd = self._side
        s0 = torch.cuda.current_stream()
        with torch.cuda.stream(sd["stream"]):
            sd["stream"].wait_stream(s0) if wait_event is None else sd["stream"].wait_event(wait_event)
            if self.consistent_ircache and wait_event is not None:
                klib.check(self.L.kj_split_merge_ircache(self.h, klib._stream_ptr()))
            first = True
            for r in self.ranks:
                gp = self.pipes[r]
                if first:
                    gp.dev.frame_begin(fc)
                    sd["fc"][self.frame & 1].record(sd["stream"])
                    first = False
                if gp.ircache:
                    s = klib._stream_ptr()
                    if self.consistent_ircache:
                        gp.ircache_begin_requests(half_rows=half_rows(*self.strips[r], self.H))
                    klib.check(gp.L.kj_ircache_prepare(gp.ircache, s))
                    klib.check(gp.L.kj_ircache_trace_irradiance(gp.ircache, gp.scene.h, gp.sky16.data_ptr(), 16, s))
            if self.on_ircache_traced is not None:
                self.on_ircache_traced()
            sd["irc"][self.frame & 1].record(sd["stream"])

    def ssgi_frame(self):
        """The SSAO guide before gi_frame, strip by strip with its two halo exchanges (kj_split_ssgi_frame; SplitRtdgi.ssgi_frame is the reference)."""
        n = len(self.ranks)
        handles, outs = (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, r in enumerate(self.ranks):
            gp = self.pipes[r]
            if gp.ssgi is None:
                gp.ssgi = C.c_void_p()
                klib.check(self.L.kj_ssgi_create(gp.dev.h, C.byref(gp.ssgi)))
            handles[i] = gp.ssgi.value
        self._fill()
        klib.check(self.L.kj_split_ssgi_frame(self.h, handles, self._frames, outs, klib._stream_ptr()))
        for i, r in enumerate(self.ranks):
            self.pipes[r].ssao_ptr = C.c_void_p(outs[i])

    def enable_rtr(self, tables=None):
        """Reflections join the frame (kj_split_set_rtr; SplitRtdgi.enable_rtr is the reference)."""
        self.with_rtr = True
        for gp in self.pipes.values():
            gp.rtr_handle(tables)
        klib.check(self.L.kj_split_set_rtr(self.h, 1))

    def rtr_frame(self, specular_lights=False, trace_event=None, defer_merge=False):
        """kj_split_rtr_frame: reflections strip by strip after gi_frame (SplitRtdgi.rtr_frame is the reference). Returns {rank: int32 (H, W) view of the
        resolved image, valid on the rank's own rows}."""
        import torch
        from .abi import KjRtrParams
        n = len(self.ranks)
        handles, outs, params = (C.c_void_p * n)(), (C.c_void_p * n)(), (KjRtrParams * n)()
        for i, r in enumerate(self.ranks):
            handles[i] = self.pipes[r].rtr_handle().value
            params[i] = self.pipes[r].rtr_params(0)
        handle = None
        if trace_event is not None:
            try:
                trace_event.record()
                handle = int(trace_event.cuda_event)
            except Exception:
                handle = None
        klib.check(self.L.kj_split_rtr_frame(self.h, handles, params, (1 if specular_lights else 0) | (2 if defer_merge else 0), handle, outs, klib._stream_ptr()))
        if trace_event is not None and handle is None:
            trace_event.record()
        return {r: self.pipes[r].rtr_surface("resolved_tex", torch.int32, (self.H, self.W)) for r in self.ranks}

    def shadow_frame(self, masks=None, ray_counters=None):
        """kj_split_shadow_frame: the sun shadow mask + its denoiser strip by strip (SplitRtdgi.shadow_frame is the reference). Returns {rank: RG16F
        image valid on the rank's own rows}."""
        import torch
        n = len(self.ranks)
        handles, mk, outs, ctr = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, r in enumerate(self.ranks):
            gp = self.pipes[r]
            handles[i] = gp.shadow_denoiser().value
            if masks is not None:
                gp.shadow_mask_img = masks[r]
            elif getattr(gp, "shadow_mask_img", None) is None:
                gp.shadow_mask_img = torch.zeros((self.H, self.W), dtype=torch.uint8, device=gp.depth.device)
            mk[i] = gp.shadow_mask_img.data_ptr()
            ctr[i] = ray_counters[r].data_ptr() if ray_counters else None
        self._fill()
        klib.check(self.L.kj_split_shadow_frame(self.h, handles, self._frames, mk, ctr if ray_counters else None, outs, klib._stream_ptr()))
        return {r: klib.tensor_from_ptr(outs[i], self.W * self.H * 4, torch.float16, (self.H, self.W, 2)) for i, r in enumerate(self.ranks)}

    def frame_pipelined(self, next_fc, run_ssgi=False):
        import torch
        i = self.frame & 1
        if run_ssgi:
            torch.cuda.current_stream().wait_event(self._side["fc"][i])
            self.ssgi_frame()
        torch.cuda.current_stream().wait_event(self._side["irc"][i])
        self.gi_frame(ircache_done=True, trace_event=self._side["trace"][i], defer_merge=True)     # (advances self.frame)
        self.taa_frame()
        if next_fc is not None:
            self._enqueue_ircache(next_fc, self._side["trace"][i])
        elif self.consistent_ircache:
            sd = self._side
            with torch.cuda.stream(sd["stream"]):
                sd["stream"].wait_event(sd["trace"][i])
                klib.check(self.L.kj_split_merge_ircache(self.h, klib._stream_ptr()))
            torch.cuda.current_stream().wait_stream(sd["stream"])

    def taa_frame(self, inputs=None):
        self._fill()
        if inputs is None:
            klib.check(self.L.kj_split_taa_frame(self.h, self._frames, klib._stream_ptr()))
        else:
            ptrs = (C.c_void_p * len(self.ranks))(*[inputs[r].data_ptr() for r in self.ranks])
            klib.check(self.L.kj_split_taa_frame_on(self.h, self._frames, ptrs, klib._stream_ptr()))

    def lighting_frame(self, with_ssgi=True, specular_lights=False):
        return _lighting_frame(self, self.ranks, with_ssgi, specular_lights)

    def lighting_frame_pipelined(self, next_fc, with_ssgi=True, specular_lights=False):
        return _lighting_frame_pipelined(self, self.ranks, next_fc, with_ssgi, specular_lights)

    def _replay_recorded_cache_updates(self):
        klib.check(self.L.kj_split_merge_ircache(self.h, klib._stream_ptr()))

    def gather_output(self, name="spatial_filtered_tex"):
        klib.check(self.L.kj_split_gather(self.h, name.encode(), klib._stream_ptr()))

    def self_test(self, dist=None):
        """Start-up check of the compiled transport before frame 0 (kj_split_self_test: every kind of exchange of the frame schedule on scratch
        images, checked row by row). `dist`: the torch.distributed module of an N-process job -- the ranks' verdicts are combined (MIN), so
        every rank returns the same answer; a transport ERROR on any rank counts as a failure on all of them instead of raising on one."""
        passed = C.c_uint32(0)
        try:
            klib.check(self.L.kj_split_self_test(self.h, C.byref(passed), klib._stream_ptr()))
            ok = bool(passed.value)
        except Exception as e:
            import sys
            print(f"[kajiya_amd split self-test] transport error: {e}", file=sys.stderr, flush=True)
            ok = False
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            import torch
            on_host = dist.get_backend() == "gloo"
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_host else next(iter(self.pipes.values())).depth.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        return ok

    @staticmethod
    def rccl_comm_from_torch(dist, rank, world, device):
        """An RCCL communicator of the library's own (torch does not hand out its ncclComm_t): rank 0 draws the id, torch.distributed
        broadcasts its 128 bytes, every rank initialises. Returns the ncclComm_t as an integer."""
        import torch
        L = klib.load()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            klib.check(L.kj_split_rccl_unique_id(ident))
        t = torch.tensor(list(ident), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        comm = C.c_void_p()
        klib.check(L.kj_split_rccl_comm_create(ident, world, rank, C.byref(comm)))
        return comm.value

    @staticmethod
    def rccl_one_rank_comm():
        """A communicator of ONE rank (ncclGetUniqueId -> ncclCommInitRank(nranks = 1)) for the loopback mode: NativeSplit(world, all pipes, ..., nccl_comm=this)
        sends every message of the schedule to self through RCCL. Returns (ncclComm_t as an integer, ranks and rank as the communicator reports them)."""
        L = klib.load()
        ident = (C.c_uint8 * 128)()
        klib.check(L.kj_split_rccl_unique_id(ident))
        comm = C.c_void_p()
        klib.check(L.kj_split_rccl_comm_create(ident, 1, 0, C.byref(comm)))
        n, r = C.c_uint32(), C.c_uint32()
        klib.check(L.kj_split_rccl_comm_info(comm, C.byref(n), C.byref(r)))
        return comm.value, n.value, r.value
