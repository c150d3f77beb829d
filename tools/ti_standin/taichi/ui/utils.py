"""taichi.ui.utils of the stand-in: only imported by src/camera.py's interactive code, never called."""


def euler_to_vec(yaw, pitch):
    raise NotImplementedError("interactive camera code is outside the cross-check")


def vec_to_euler(v):
    raise NotImplementedError("interactive camera code is outside the cross-check")
