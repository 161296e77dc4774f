"""Caller data of the post-processing path.

post_combine.hlsl reads bindless texture 2 (`BINDLESS_LUT_BEZOLD_BRUCKE`, default_world_renderer.rs:49-51): a 64x1 RG16F image
kajiya computes once from the CIE 1931 standard observer (lut_renderers.rs:45-76, shaders/lut/bezold_brucke.hlsl). Like the blue-noise
image and the rtr tables it is an INPUT of this library (`kj_post_create`); a kajiya host hands over its own LUT. For tests and
viewable frames without one:

  zero_bezold_brucke_lut()       no hue shift (the display transform without USE_BEZOLD_BRUCKE_SHIFT)
  synthetic_bezold_brucke_lut()  a smooth, small, seeded xy offset per LUT cell: exercises the sampling path in the parity tests; it is
                                 NOT the perceptual data
"""
import numpy as np


def zero_bezold_brucke_lut():
    return np.zeros((64, 2), np.float16)


def synthetic_bezold_brucke_lut(seed=0, magnitude=0.05):
    rng = np.random.RandomState(seed)
    t = (np.arange(64) + 0.5) / 64.0 * 2.0 * np.pi
    ph = rng.uniform(0, 2 * np.pi, 4)
    lut = np.stack([np.sin(t + ph[0]) + 0.5 * np.sin(3 * t + ph[1]), np.cos(2 * t + ph[2]) + 0.5 * np.sin(5 * t + ph[3])], -1)
    return (lut * magnitude / 1.5).astype(np.float16)
