"""Camera pose smoothing of the interactive viewer, as a pure host function.

reference: src/camera.py:82-112 ``SmoothCamera._update`` — every frame the rendered pose moves a fraction
``clamp(velocity * dt, 0, 1)`` of the way to the pose the user steers, and ``moving`` says whether any component
still differs by more than 1e-3; src/renderer.py:26-27 — ``render(refreshing)`` clears the accumulation when
``refreshing or smooth.moving``.  The GGUI window, key bindings and the Euler steering (camera.py:66-80) are UI and
are not rebuilt; this is the part of the contract the sample path depends on: WHEN the buffers are refreshed.
"""
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

Vec = Tuple[float, float, float]


@dataclass
class SmoothCamera:
    position: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    lookat: np.ndarray = field(default_factory=lambda: np.array([0, 0, 1], np.float32))
    up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], np.float32))
    position_velocity: float = 10.0          # src/camera.py:51-53
    lookat_velocity: float = 10.0
    up_velocity: float = 10.0
    moving: bool = False
    frame: int = 0                           # u_frame, incremented per update (camera.py:112)

    def init(self, position: Vec, lookat: Vec = (0, 0, 1), up: Vec = (0, 1, 0)):
        """camera.py:56-59"""
        self.position = np.asarray(position, np.float32)
        self.lookat = np.asarray(lookat, np.float32)
        self.up = np.asarray(up, np.float32)
        return self

    def update(self, dt: float, curr_position: Vec, curr_lookat: Vec, curr_up: Vec) -> bool:
        """camera.py:82-112 in f32; returns ``moving``"""
        f = np.float32
        tgt = [np.asarray(v, np.float32) for v in (curr_position, curr_lookat, curr_up)]
        cur = [self.position, self.lookat, self.up]
        vel = [self.position_velocity, self.lookat_velocity, self.up_velocity]
        diffs = [t - c for t, c in zip(tgt, cur)]
        for i in range(3):
            w = f(min(max(f(vel[i]) * f(dt), f(0)), f(1)))
            cur[i] = (cur[i] + diffs[i] * w).astype(np.float32)
        self.position, self.lookat, self.up = cur
        self.moving = bool(max(float(np.abs(d).max()) for d in diffs) > 1e-3)
        self.frame += 1
        return self.moving


def render_interactive_frame(renderer, smooth: SmoothCamera, refreshing: bool = False):
    """src/renderer.py:25-32 with the smoothed pose: upload the pose, refresh if asked or moving, sample, tone map"""
    from .dataclass import Camera
    c = renderer.camera
    renderer.set_camera(Camera(tuple(smooth.position), tuple(smooth.lookat), tuple(smooth.up), c.vfov, c.aspect, c.aperture, c.focus))
    renderer.render(refreshing=bool(refreshing or smooth.moving))
