"""Tile partition of the frame across GPUs.

The reference is single-device; its only parallelism is Taichi's data-parallel
``for i, j in image_pixels`` (src/pathtracer.py:96).  Pixels are independent, so the frame
is cut into small tiles dealt round-robin (tile t -> rank t % world): cost per pixel is very
uneven (black margins at 16:9, the light, glass), and interleaving balances it.  Results do
not depend on the partition because the RNG is keyed by absolute pixel coordinates.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class TileLayout:
    width: int
    height: int
    tile_w: int
    tile_h: int
    world: int

    @property
    def ntx(self):
        return (self.width + self.tile_w - 1) // self.tile_w

    @property
    def nty(self):
        return (self.height + self.tile_h - 1) // self.tile_h

    @property
    def n_tiles(self):
        return self.ntx * self.nty

    @property
    def n_local_tiles(self):
        """tiles per rank, padded so every rank packs the same number of bytes"""
        return (self.n_tiles + self.world - 1) // self.world

    @property
    def packed_pixels(self):
        return self.n_local_tiles * self.tile_w * self.tile_h

    def tiles_of(self, rank):
        return [t for t in range(rank, self.n_tiles, self.world)]

    def owner(self, x, y):
        return ((y // self.tile_h) * self.ntx + (x // self.tile_w)) % self.world

    def owner_map(self):
        x = np.arange(self.width)[:, None]
        y = np.arange(self.height)[None, :]
        return ((y // self.tile_h) * self.ntx + (x // self.tile_w)) % self.world

    def pixel_index(self, rank):
        """(x, y, valid) arrays of length packed_pixels: the packed order of rank's buffer
        (tile-major, x-major inside a tile, y fastest) — the order rtpbr_pack_tiles uses."""
        tpix = self.tile_w * self.tile_h
        q = np.arange(self.packed_pixels)
        tl, r = q // tpix, q % tpix
        lx, ly = r // self.tile_h, r % self.tile_h
        tid = rank + tl * self.world
        ty, tx = tid // self.ntx, tid % self.ntx
        x, y = tx * self.tile_w + lx, ty * self.tile_h + ly
        valid = (x < self.width) & (y < self.height) & (ty < self.nty)
        return x, y, valid

    def pack(self, image_buffer, rank):
        """numpy model of rtpbr_pack_tiles: (W,H,4) -> (packed_pixels,4), padding = 0"""
        x, y, valid = self.pixel_index(rank)
        out = np.zeros((self.packed_pixels, image_buffer.shape[2]), image_buffer.dtype)
        out[valid] = image_buffer[x[valid], y[valid]]
        return out

    def unpack_into(self, image_buffer, packed, rank):
        x, y, valid = self.pixel_index(rank)
        image_buffer[x[valid], y[valid]] = packed[valid]
        return image_buffer


def default_tile(width, height, world):
    """Small square-ish tiles; at least ~8 tiles per rank so the round-robin deal balances."""
    if world <= 1:
        return width, height
    t = 32
    while t > 8 and ((width + t - 1) // t) * ((height + t - 1) // t) < world * 16:
        t //= 2
    return t, t
