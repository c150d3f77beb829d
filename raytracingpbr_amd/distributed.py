"""Multi-GPU rendering: tile-partitioned frame + ONE gather of per-tile radiance.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  Each rank renders its own tiles into its local image_buffer,
packs them tile-major into one contiguous buffer, and a single ``dist.gather`` moves the
packed buffers to rank 0, which scatters them into the full (W,H,4) image_buffer
(SURVEY.md §8(e)).  There is no communication while rendering.  The reference has no
multi-device path; this is new.
"""
import numpy as np

from .tiles import TileLayout, default_tile


class TileGather:
    """Owns the layout and the packed send/receive tensors of one rank."""

    def __init__(self, renderer, rank, world, tile=None, device=None):
        import torch
        self.torch = torch
        self.r, self.rank, self.world = renderer, rank, world
        W, H = renderer.config.width, renderer.config.height
        tw, th = tile if tile is not None else default_tile(W, H, world)
        self.layout = TileLayout(W, H, tw, th, world)
        renderer.set_tiles(tw, th, rank, world)
        self.device = device
        n = self.layout.packed_pixels * 4
        self.send = torch.empty(n, dtype=torch.float32, device=device)
        self.recv = [torch.empty(n, dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None

    def _pack(self):
        t = self.send
        if t.is_cuda:
            self.r.pack_tiles(t.data_ptr())
            self.r.sync()                       # our stream -> visible to the collective's stream
        else:                                   # host tensors (gloo tests): pack on the host
            t.copy_(self.torch.from_numpy(self.layout.pack(self.r.image_buffer, self.rank).reshape(-1)))

    def gather(self):
        """Single gather of the packed tiles to rank 0; rank 0 ends with the full frame."""
        import torch.distributed as dist
        self._pack()
        if self.world > 1:
            dist.gather(self.send, self.recv, dst=0)
        else:
            self.recv[0].copy_(self.send)
        if self.rank != 0:
            return
        if self.send.is_cuda:
            self.torch.cuda.current_stream().synchronize()
            for src in range(1, self.world):
                self.r.unpack_tiles(self.recv[src].data_ptr(), src)
            self.r.sync()
        else:
            full = self.r.image_buffer
            for src in range(1, self.world):
                self.layout.unpack_into(full, self.recv[src].numpy().reshape(-1, 4), src)
            self.r.image_buffer = full


def rccl_group(renderers):
    """One process driving G devices: ncclCommInitAll over the renderers' devices (rank i = renderers[i]); afterwards
    gather_group(renderers) runs the single gather.  Tiles must have been set with rank i of len(renderers)."""
    import ctypes as C
    arr = (C.c_void_p * len(renderers))(*[r._ctx for r in renderers])
    renderers[0].api.call("rccl_init_all", arr, len(renderers))


def gather_group(renderers):
    import ctypes as C
    arr = (C.c_void_p * len(renderers))(*[r._ctx for r in renderers])
    renderers[0].api.call("gather_tiles_all", arr, len(renderers))


def render_distributed(renderer, spp, rank, world, tile=None, device=None, refresh=True):
    """refresh -> spp samples on this rank's tiles -> one gather -> (rank 0) post_process."""
    tg = TileGather(renderer, rank, world, tile, device)
    if refresh:
        renderer.refresh()
    renderer.sample(spp)
    tg.gather()
    if rank == 0:
        renderer.post_process()
    return tg
