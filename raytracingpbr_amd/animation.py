"""Offline animation driver.

reference: examples/bunny/bunny_sdf_glass.py:434-451 — for each of 241 frames: set the
``u_frame`` uniform, ``refresh()``, SAMPLE_PER_PIXEL x ``sample()``, ``render()`` (tone map),
``ti.tools.imwrite(image_pixels, 'out/frame_%d.png')``.  Same loop here on the Renderer.
"""
import os

from .imageio import imwrite


def render_animation(renderer, frames, spp, out_dir=None, pattern="frame_{:04d}.png", on_frame=None):
    """Render ``frames`` (an iterable of frame numbers).  Returns the list of written paths (or
    of image_pixels arrays when out_dir is None)."""
    out = []
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
    for f in frames:
        renderer.set_config(renderer.config.copy(frame=int(f)))      # u_frame[None] = frame
        renderer.refresh()
        renderer.sample(spp)
        renderer.post_process()
        px = renderer.image_pixels
        if on_frame is not None:
            on_frame(f, px)
        if out_dir is not None:
            path = os.path.join(out_dir, pattern.format(int(f)))
            imwrite(px, path)
            out.append(path)
        else:
            out.append(px)
    return out
