"""Host-side frame constants, mirroring the reference's CPU code:

  CameraLens::calc_matrices            crates/lib/kajiya/src/camera.rs:88-125
  CameraBodyMatrices                   crates/lib/kajiya/src/camera.rs:66-85
  ViewConstants::builder/set_pixel_offset   rust-shaders-shared/src/view_constants.rs:25-121
  WorldRenderer::prepare_frame_constants    crates/lib/kajiya/src/world_renderer.rs:1001-1108
  supersample offsets (Halton 2,3)          world_renderer.rs:425-428,1116-1129
"""
import math
import numpy as np
from .abi import KjFrameConstants


def radical_inverse(n, base):
    val = np.float32(0.0)
    inv_base = np.float32(1.0) / np.float32(base)
    inv_bi = inv_base
    while n > 0:
        d_i = n % base
        val = np.float32(val + np.float32(d_i) * inv_bi)
        n = int(np.float32(n) * inv_base)
        inv_bi = np.float32(inv_bi * inv_base)
    return float(val)


SUPERSAMPLE_OFFSETS = [(radical_inverse(i, 2) - 0.5, radical_inverse(i, 3) - 0.5) for i in range(1, 129)]


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle * 0.5)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle * 0.5)])


def quat_to_mat3(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def look_at_rotation(eye, target, up=(0, 1, 0)):
    """Rotation matrix (view->world, camera looks down -Z)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    return np.stack([r, u, -f], axis=1)


class CameraMatrices:
    def __init__(self, position, rot3, vfov_deg, aspect, znear=0.01):
        fov = math.radians(vfov_deg)
        h = math.cos(0.5 * fov) / math.sin(0.5 * fov)
        w = h / aspect
        v2c = np.zeros((4, 4))
        v2c[0, 0] = w; v2c[1, 1] = h; v2c[3, 2] = -1.0; v2c[2, 3] = znear
        c2v = np.zeros((4, 4))
        c2v[0, 0] = 1.0 / w; c2v[1, 1] = 1.0 / h; c2v[3, 2] = 1.0 / znear; c2v[2, 3] = -1.0
        v2w = np.eye(4); v2w[:3, :3] = rot3; v2w[:3, 3] = position
        w2v = np.eye(4); w2v[:3, :3] = rot3.T; w2v[:3, 3] = -rot3.T @ np.asarray(position, np.float64)
        self.view_to_clip = v2c.astype(np.float32)
        self.clip_to_view = c2v.astype(np.float32)
        self.view_to_world = v2w.astype(np.float32)
        self.world_to_view = w2v.astype(np.float32)


def _set_mat(dst, m):
    flat = np.asarray(m, np.float32).flatten(order="F")  # column-major memory (glam::Mat4)
    for i in range(16):
        dst[i] = float(flat[i])


class FrameState:
    """The part of WorldRenderer that produces FrameConstants each frame."""

    def __init__(self, render_extent, sun_direction=(4.0, 1.0, 1.0), sun_size_multiplier=1.0,
                 sun_color_multiplier=(1.0, 1.0, 1.0), sky_ambient=(0.0, 0.0, 0.0), use_taa_jitter=True):
        self.render_extent = tuple(render_extent)
        d = np.asarray(sun_direction, np.float64)
        self.sun_direction = (d / np.linalg.norm(d)).astype(np.float32)
        self.sun_size_multiplier = sun_size_multiplier
        self.sun_color_multiplier = sun_color_multiplier
        self.sky_ambient = sky_ambient
        self.use_taa_jitter = use_taa_jitter
        self.frame_idx = 0
        self.prev_camera = None
        self.triangle_light_count = 0
        self.pre_exposure = 1.0        # ExposureState (kajiya_amd/exposure.py: Exposure.apply), world_renderer.rs:1084-1086
        self.pre_exposure_prev = 1.0
        self.pre_exposure_delta = 1.0
        self.ircache_grid_center = (0.0, 0.0, 0.0, 1.0)
        self.ircache_cascades = None  # list of 12 (origin[4], scrolled[4]) once the ircache is enabled
        # IrcacheRenderer host state (renderers/ircache.rs:92-158)
        self.ircache_enabled = False
        self._irc_cur = np.zeros((12, 3), np.int32)
        self._irc_prev = np.zeros((12, 3), np.int32)

    def ircache_update_eye_position(self, eye):
        """IrcacheRenderer::update_eye_position + constants (ircache.rs:126-158), f32 arithmetic."""
        eye = np.asarray(eye, np.float32)
        self.ircache_grid_center = (float(eye[0]), float(eye[1]), float(eye[2]), 1.0)
        casc = []
        for c in range(12):
            cell_diameter = np.float32(0.16 * 0.125) * np.float32(1 << c)
            center = np.floor(eye / cell_diameter).astype(np.int32)
            origin = center - np.int32(16)
            self._irc_prev[c] = self._irc_cur[c]
            self._irc_cur[c] = origin
            casc.append((list(origin) + [0], list(self._irc_cur[c] - self._irc_prev[c]) + [0]))
        self.ircache_cascades = casc

    def prepare_frame_constants(self, cam: CameraMatrices, delta_time_seconds=1.0 / 60.0) -> KjFrameConstants:
        prev = self.prev_camera or cam
        if self.ircache_enabled:
            # world_renderer.rs:1061-1069: update_eye_position(view_constants.eye_position()) before the constants are pushed
            self.ircache_update_eye_position(cam.view_to_world[:3, 3])
        fc = KjFrameConstants()
        vc = fc.view_constants
        f32 = np.float32
        clip_to_prev_clip = (prev.view_to_clip.astype(f32) @ prev.world_to_view.astype(f32)) @ (cam.view_to_world.astype(f32) @ cam.clip_to_view.astype(f32))
        _set_mat(vc.view_to_clip, cam.view_to_clip)
        _set_mat(vc.clip_to_view, cam.clip_to_view)
        _set_mat(vc.world_to_view, cam.world_to_view)
        _set_mat(vc.view_to_world, cam.view_to_world)
        _set_mat(vc.clip_to_prev_clip, clip_to_prev_clip)
        _set_mat(vc.prev_view_to_prev_clip, prev.view_to_clip)
        _set_mat(vc.prev_clip_to_prev_view, prev.clip_to_view)
        _set_mat(vc.prev_world_to_prev_view, prev.world_to_view)
        _set_mat(vc.prev_view_to_prev_world, prev.view_to_world)
        off = SUPERSAMPLE_OFFSETS[self.frame_idx % len(SUPERSAMPLE_OFFSETS)] if self.use_taa_jitter else (0.0, 0.0)
        w, h = self.render_extent
        soc = (f32(2.0 * off[0]) / f32(w), f32(2.0 * off[1]) / f32(h))
        jitter = np.eye(4, dtype=f32); jitter[0, 3] = -soc[0]; jitter[1, 3] = -soc[1]
        jitter_inv = np.eye(4, dtype=f32); jitter_inv[0, 3] = soc[0]; jitter_inv[1, 3] = soc[1]
        _set_mat(vc.view_to_sample, jitter @ cam.view_to_clip)
        _set_mat(vc.sample_to_view, cam.clip_to_view @ jitter_inv)
        vc.sample_offset_pixels[0], vc.sample_offset_pixels[1] = off
        vc.sample_offset_clip[0], vc.sample_offset_clip[1] = float(soc[0]), float(soc[1])
        for i in range(3):
            fc.sun_direction[i] = float(self.sun_direction[i])
            fc.sun_color_multiplier[i] = self.sun_color_multiplier[i]
            fc.sky_ambient[i] = self.sky_ambient[i]
        fc.frame_index = self.frame_idx
        fc.delta_time_seconds = delta_time_seconds
        real_sun_angular_radius = math.radians(0.53) * 0.5
        fc.sun_angular_radius_cos = math.cos(self.sun_size_multiplier * real_sun_angular_radius)
        fc.triangle_light_count = self.triangle_light_count
        fc.pre_exposure = self.pre_exposure
        fc.pre_exposure_prev = self.pre_exposure_prev
        fc.pre_exposure_delta = self.pre_exposure_delta
        fc.render_overrides.flags = 0
        fc.render_overrides.material_roughness_scale = 1.0
        for i in range(4):
            fc.ircache_grid_center[i] = self.ircache_grid_center[i]
        if self.ircache_cascades is not None:
            for i, (origin, scrolled) in enumerate(self.ircache_cascades):
                for k in range(4):
                    fc.ircache_cascades[i].origin[k] = int(origin[k])
                    fc.ircache_cascades[i].voxels_scrolled_this_frame[k] = int(scrolled[k])
        self.prev_camera = cam
        return fc

    def retire_frame(self):
        self.frame_idx += 1


def orbit_camera(frame, extent, center=(0.0, 1.0, 0.0), radius=6.0, height=1.0, vfov=52.0, rate=0.004, phase=0.0):
    """Slow orbit used by bench/tests so reprojection and both frame types are exercised."""
    ang = phase + rate * frame
    eye = (center[0] + radius * math.sin(ang), center[1] + height, center[2] + radius * math.cos(ang))
    rot = look_at_rotation(eye, center)
    return CameraMatrices(eye, rot, vfov, extent[0] / extent[1])
