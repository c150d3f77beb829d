"""GUI stubs of the stand-in runtime: a window that is never running, a camera that only stores its pose
(defaults as documented for ti.ui.Camera: position (0,0,0), lookat (0,0,1), up (0,1,0))."""
from ..math import vec3

LMB, RMB, MMB = "LMB", "RMB", "MMB"
LEFT, RIGHT, UP, DOWN, RELEASE, PRESS, SPACE, SHIFT, ESCAPE = "Left", "Right", "Up", "Down", 0, 1, " ", "Shift", "Escape"


class _Canvas:
    def set_image(self, img):
        pass


class Window:
    running = False

    def __init__(self, *a, **kw):
        pass

    def get_canvas(self): return _Canvas()
    def get_events(self, *a): return []
    def is_pressed(self, *a): return False
    def show(self): pass
    def save_image(self, path): pass
    def destroy(self): pass


class Camera:
    def __init__(self):
        self.curr_position = vec3(0.0, 0.0, 0.0)
        self.curr_lookat = vec3(0.0, 0.0, 1.0)
        self.curr_up = vec3(0.0, 1.0, 0.0)

    def position(self, x, y, z): self.curr_position = vec3(x, y, z)
    def lookat(self, x, y, z): self.curr_lookat = vec3(x, y, z)
    def up(self, x, y, z): self.curr_up = vec3(x, y, z)
    def track_user_inputs(self, *a, **kw): pass
