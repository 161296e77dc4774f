"""Data tables RtrRenderer::new uploads (renderers/rtr.rs:16,66-68,482-915) and `kj_rtr_create` takes from the caller.

The reference gets RANKING_TILE / SCRAMBLING_TILE / SOBOL from the third-party crate `blue-noise-sampler 0.1.0` (spp64; the
tables of Heitz et al., "A Low-Discrepancy Sampler that Distributes Monte Carlo Errors as a Blue Noise in Screen Space") and
SPATIAL_RESOLVE_OFFSETS from its own source (rtr.rs:402-915). A kajiya integration passes its own statics. Here:

* SPATIAL_RESOLVE_OFFSETS is the reference's table, read out of rtr.rs by scripts/extract_rtr_offsets.py into
  kajiya_amd/data/spatial_resolve_offsets_i32x4.bin and handed to kj_rtr_create as caller data (`spatial_resolve_offsets()`);
* the sampler crate is absent from the checkout (a Cargo.lock dependency), so RANKING_TILE / SCRAMBLING_TILE / SOBOL are
  STAND-INS with the same shapes and value ranges — a real (unscrambled) Sobol sequence and seeded white-noise ranking /
  scrambling tiles (no blue-noise optimisation): parity against the oracle is exact on identical tables, image noise is only
  representative."""
import os

import numpy as np

from .abi import KjRtrTables


def sobol_256x256():
    """sobol[dim + index * 256] = floor(256 * x_index[dim]) for the first 256 points of a 256-dimensional Sobol sequence."""
    from scipy.stats import qmc
    pts = qmc.Sobol(d=256, scramble=False).random_base2(8)      # (256 points, 256 dims)
    return np.ascontiguousarray(np.minimum((pts * 256.0).astype(np.uint32), 255).reshape(-1))


def ranking_and_scrambling(seed=2024):
    rng = np.random.RandomState(seed)
    ranking = rng.randint(0, 64, size=128 * 128 * 8).astype(np.uint32)       # xor-ed into the 64-spp sample index
    scrambling = rng.randint(0, 256, size=128 * 128 * 8).astype(np.uint32)   # xor-ed into the 8-bit sample value
    return ranking, scrambling


def spatial_resolve_offsets():
    """(16 * 4 * 8, 4) int32, the reference's SPATIAL_RESOLVE_OFFSETS: for each of 8 filter sizes, 4 quad-pixel variants x 16 taps."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "spatial_resolve_offsets_i32x4.bin")
    t = np.fromfile(path, dtype=np.int32).reshape(-1, 4)
    assert t.shape == (16 * 4 * 8, 4)
    return np.ascontiguousarray(t)


def synthetic_spatial_resolve_offsets(seed=7):
    """A table of the same shape made up from scratch (distance-sorted, quad-disjoint rings): only used by tests that want a
    second, different table to show the kernels take it as data."""
    rng = np.random.RandomState(seed)
    out = np.zeros((8, 4, 16, 4), np.int32)
    for f in range(8):
        radius = 4.6 + 0.45 * f
        r = int(np.ceil(radius))
        cand = [(x, y) for y in range(-r, r + 1) for x in range(-r, r + 1) if (x or y) and x * x + y * y <= radius * radius]
        order = rng.permutation(len(cand))
        cand = [cand[i] for i in order]
        assert len(cand) >= 60, (f, len(cand))
        for q in range(4):
            pts = sorted(cand[q * 15:(q + 1) * 15], key=lambda p: (p[0] * p[0] + p[1] * p[1], p))
            for k, (x, y) in enumerate(pts):
                out[f, q, k + 1, 0], out[f, q, k + 1, 1] = x, y
    return np.ascontiguousarray(out.reshape(-1, 4))


_CACHE = {}


def standin_tables():
    """(KjRtrTables, keepalive) with host pointers to the stand-in tables."""
    if "t" not in _CACHE:
        ranking, scrambling = ranking_and_scrambling()
        _CACHE["t"] = (ranking, scrambling, sobol_256x256(), spatial_resolve_offsets())
    ranking, scrambling, sobol, offsets = _CACHE["t"]
    t = KjRtrTables()
    t.ranking_tile = ranking.ctypes.data
    t.scrambling_tile = scrambling.ctypes.data
    t.sobol = sobol.ctypes.data
    t.spatial_resolve_offsets = offsets.ctypes.data
    return t, _CACHE["t"]
