"""Host-side exposure state, mirroring the reference's CPU code (f32 arithmetic as there):

  HistogramClipping, DynamicExposureState::{ev_smoothed, update}    crates/lib/kajiya/src/world_renderer.rs:217-259
  ExposureState                                                     world_renderer.rs:261-285
  WorldRenderer::update_pre_exposure                                world_renderer.rs:919-948

One frame of the loop (world_renderer.rs:953-960, world_render_passes.rs:281-289, post.rs:246):
  exposure.update_pre_exposure(image_log2_lum)       # image_log2_lum = PostProcessRenderer::read_back_histogram of an earlier frame
  fc.pre_exposure / _prev / _delta = exposure.state.pre_mult / pre_mult_prev / pre_mult_delta
  ... render ...; post.render(taa_out, post_exposure_mult=exposure.state.post_mult, contrast)
"""
import math

import numpy as np

f32 = np.float32
DYNAMIC_EXPOSURE_BIAS = f32(-2.0)


def _exp(x):      # f32::exp: the double result rounded once (numpy's own float32 exp is a few ulp off, which 1 - exp(-dt/4) magnifies)
    return f32(math.exp(float(x)))


def _exp2(x):
    return f32(2.0 ** float(x))


class HistogramClipping:
    def __init__(self, low=0.0, high=0.0):
        self.low, self.high = low, high


class DynamicExposureState:
    def __init__(self, enabled=False, speed_log2=0.0, histogram_clipping=None):
        self.enabled = enabled
        self.speed_log2 = f32(speed_log2)
        self.histogram_clipping = histogram_clipping or HistogramClipping()
        self.ev_fast = f32(0.0)
        self.ev_slow = f32(0.0)

    def ev_smoothed(self):
        if not self.enabled:
            return f32(0.0)
        return f32(f32(f32(self.ev_slow + self.ev_fast) * f32(0.5)) + DYNAMIC_EXPOSURE_BIAS)

    def update(self, ev, dt):
        if not self.enabled:
            return
        ev = f32(min(max(f32(ev), f32(-16.0)), f32(16.0)))
        dt = f32(f32(dt) * _exp2(self.speed_log2))
        t_fast = f32(f32(1.0) - _exp(f32(f32(-1.0) * dt)))
        self.ev_fast = f32(f32(f32(ev - self.ev_fast) * t_fast) + self.ev_fast)
        t_slow = f32(f32(1.0) - _exp(f32(f32(-0.25) * dt)))
        self.ev_slow = f32(f32(f32(ev - self.ev_slow) * t_slow) + self.ev_slow)


class ExposureState:
    def __init__(self):
        self.pre_mult = f32(1.0)
        self.post_mult = f32(1.0)
        self.pre_mult_prev = f32(1.0)
        self.pre_mult_delta = f32(1.0)


class Exposure:
    """The exposure-related fields of WorldRenderer (ev_shift, dynamic_exposure, contrast, exposure_state[render_mode])."""

    def __init__(self, ev_shift=0.0, dynamic_exposure=None, contrast=1.0):
        self.ev_shift = f32(ev_shift)
        self.dynamic_exposure = dynamic_exposure or DynamicExposureState()
        self.contrast = contrast
        self.exposure_state = [ExposureState(), ExposureState()]    # one per render mode: standard, reference
        self.render_mode = 0

    @property
    def state(self):
        return self.exposure_state[self.render_mode]

    def update_pre_exposure(self, image_log2_lum):
        dt = f32(1.0 / 60.0)
        self.dynamic_exposure.update(f32(-f32(image_log2_lum)), dt)
        ev_mult = _exp2(f32(self.ev_shift + self.dynamic_exposure.ev_smoothed()))
        st = self.state
        st.pre_mult_prev = st.pre_mult
        if self.render_mode == 0:
            st.pre_mult = f32(f32(st.pre_mult * f32(0.9)) + f32(ev_mult * f32(0.1)))
            st.post_mult = f32(ev_mult / st.pre_mult)
        else:
            st.pre_mult = f32(1.0)
            st.post_mult = ev_mult
        st.pre_mult_delta = f32(st.pre_mult / st.pre_mult_prev)

    def apply(self, frame_state):
        """Copy pre_mult / prev / delta into a kajiya_amd.frame.FrameState (world_renderer.rs:1084-1086)."""
        st = self.state
        frame_state.pre_exposure = float(st.pre_mult)
        frame_state.pre_exposure_prev = float(st.pre_mult_prev)
        frame_state.pre_exposure_delta = float(st.pre_mult_delta)
