"""Renderer: the host-side mirror of the reference's render loop.

reference: src/renderer.py:12-32 — ``refresh()``, ``render(refreshing)`` =
``[refresh()] ; SAMPLES_PER_FRAME x pathtrace() ; post_process()`` — and the examples'
kernels ``render(camera_position, camera_lookat, camera_up, moving)``
(cornell_box_v3/renderer.py:11-42) / ``sample() refresh() render()``
(bunny_sdf_glass.py:393-432).  Buffers keep the reference's field names and layout:
``image_buffer`` (W,H,4) f32 = (sum r,g,b, count), ``image_pixels`` (W,H,3) f32,
``ray_buffer`` (W,H) of Ray; index [i, j] = column from the left, row from the bottom.
"""
import ctypes as C

import numpy as np

from . import _capi
from .config import FORM, Config
from .dataclass import Camera, Counters, Ray, SDFObject
from .scene import Scene

BUF_IMAGE_BUFFER, BUF_IMAGE_PIXELS, BUF_RAY_BUFFER, BUF_DIFF_BUFFER, BUF_DIFF_PIXELS = 0, 1, 2, 3, 4
ENV_RGB8, ENV_RGB32F = 0, 1


class Renderer:
    def __init__(self, scene: Scene, config: Config, camera: Camera = None, device: int = 0, api=None):
        self.api = api if api is not None else _capi.hip_api()
        self._ctx = C.c_void_p()
        self.api.call("create", int(device), C.byref(self._ctx))
        self.device = device
        self.samples_per_frame = 1           # SAMPLES_PER_FRAME, src/config.py:9
        self._host_arrays = {}               # address -> bytes of the page-locked blocks handed out by host_array()
        self.set_config(config)
        self.set_scene(scene)
        self.set_camera(camera if camera is not None else scene.camera)

    # ------------------------------------------------------------ setup
    def set_config(self, config: Config):
        self.config = config.copy()
        self.api.call("set_config", self._ctx, C.byref(self.config))

    def set_scene(self, scene: Scene):
        n = len(scene.objects)
        arr = (SDFObject * n)(*scene.objects)
        self.api.call("set_scene", self._ctx, arr, n, 1 if scene.scale10 else 0)
        self.scene = scene

    def get_scene(self):
        n = len(self.scene.objects)
        arr = (SDFObject * n)()
        self.api.call("get_scene", self._ctx, arr, n)
        return list(arr)

    def set_camera(self, camera: Camera):
        self.camera = camera
        self.api.call("set_camera", self._ctx, C.byref(camera))

    def set_env(self, texels: np.ndarray, exposure: float = 1.0, gamma: float = 1.0):
        """texels: (W_e,H_e,3) uint8 (as ti.tools.imread returns, src/ibl.py:15) or float32."""
        t = np.ascontiguousarray(texels)
        if t.ndim != 3 or t.shape[2] != 3:
            raise ValueError("env texels must have shape (W, H, 3)")
        if t.dtype == np.uint8:
            fmt = ENV_RGB8
        elif t.dtype == np.float32:
            fmt = ENV_RGB32F
        else:
            raise TypeError("env texels must be uint8 or float32")
        self.api.call("set_env", self._ctx, t.ctypes.data_as(C.c_void_p), t.shape[0], t.shape[1], fmt,
                      float(exposure), float(gamma))

    def set_shape_data(self, shape: int, data: np.ndarray):
        """Per-shape data blob; the bunny MLP's 625 weights (bunny_sdf_glass.py:157-201)."""
        d = np.ascontiguousarray(data, dtype=np.float32)
        self.api.call("set_shape_data", self._ctx, int(shape), d.ctypes.data_as(C.c_void_p), d.size)

    def set_tiles(self, tile_w, tile_h, rank, world):
        self.api.call("set_tiles", self._ctx, tile_w, tile_h, rank, world)
        self.tiles = (tile_w, tile_h, rank, world)

    def set_option(self, key: str, value: int):
        self.api.call("set_option", self._ctx, key.encode(), int(value))

    # ------------------------------------------------------------ the reference's calls
    def refresh(self):
        """src/renderer.py:12-22."""
        self.api.call("refresh", self._ctx)

    def sample(self, n: int = 1):
        """complete-path form: n samples per pixel; persistent-ray form: n pathtrace() launches."""
        self.api.call("sample", self._ctx, int(n))

    pathtrace = sample

    def post_process(self):
        """src/postprocessor.py:24-43."""
        self.api.call("post_process", self._ctx)

    def render(self, refreshing: bool = False, spp: int = None):
        """src/renderer.py:25-32.  ``spp`` overrides SAMPLES_PER_FRAME for this call."""
        if refreshing:
            self.refresh()
        self.sample(self.samples_per_frame if spp is None else spp)
        self.post_process()

    def sync(self):
        self.api.call("sync", self._ctx)

    # ------------------------------------------------------------ buffers (field.to_numpy())
    def _shape(self, which):
        W, H = self.config.width, self.config.height
        return {BUF_IMAGE_BUFFER: ((W, H, 4), np.float32), BUF_IMAGE_PIXELS: ((W, H, 3), np.float32),
                BUF_RAY_BUFFER: ((W, H, 10), np.float32), BUF_DIFF_BUFFER: ((W, H, 2), np.float32),
                BUF_DIFF_PIXELS: ((W, H), np.float32)}[which]

    def _read(self, which):
        shape, dt = self._shape(which)
        out = np.empty(shape, dt)
        self.api.call("read_buffer", self._ctx, which, out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def host_array(self, which):
        """A page-locked numpy array of a buffer's shape (rtpbr_host_alloc): the destination a host that shows every frame reads
        into again and again — ``r.read_into(BUF_IMAGE_PIXELS, a)`` or ``r.read_async(BUF_IMAGE_PIXELS, a)``.  The memory
        belongs to the renderer (numpy cannot own page-locked memory): it is released by ``host_release(a)`` or, with every
        other block, by ``close()`` — the array must not be touched after either.  Copy (``a.copy()``) what has to outlive them."""
        shape, dt = self._shape(which)
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        ptr = C.c_void_p()
        self.api.call("host_alloc", self._ctx, n, C.byref(ptr))
        buf = (C.c_char * n).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._host_arrays[ptr.value] = n
        return a

    def host_release(self, arr):
        """Give a host_array() block back (rtpbr_host_free); waits for copies into it.  The array is dead afterwards."""
        addr = arr.ctypes.data
        if addr not in self._host_arrays:
            raise ValueError("not an array of this renderer's host_array()")
        self.api.call("host_free", self._ctx, C.c_void_p(addr))
        del self._host_arrays[addr]

    # ------------------------------------------------------------ the frame without a stall (round 6)
    def device_ptr(self, which):
        """(device address, bytes) of a buffer: zero copy for a consumer on the same GPU — what ``canvas.set_image(image_pixels)``
        does in the reference (src/main.py:64).  Order the consumer behind ``stream()`` (or call ``sync()``)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        self.api.call("buffer_device_ptr", self._ctx, which, C.byref(ptr), C.byref(n))
        return ptr.value, n.value

    def device_array(self, which):
        """The buffer as an object with ``__cuda_array_interface__`` (no copy): ``torch.as_tensor(r.device_array(BUF_IMAGE_PIXELS),
        device="cuda")`` is a tensor ON the renderer's memory."""
        addr, _ = self.device_ptr(which)
        shape, dt = self._shape(which)

        class _DeviceView:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dt).str, "data": (addr, False), "version": 2, "strides": None}
        return _DeviceView()

    def read_async(self, which, out) -> int:
        """Enqueue the copy of a buffer into a ``host_array()`` behind everything enqueued so far and return a ticket at once; the
        renderer's next kernels run while the copy is in flight (only a call that overwrites the buffer waits for it, on the
        device).  ``read_wait(ticket)`` blocks until ``out`` holds the frame."""
        shape, dt = self._shape(which)
        if out.shape != tuple(shape) or out.dtype != np.dtype(dt) or not out.flags.c_contiguous:
            raise ValueError(f"expected a C-contiguous {dt} array of shape {shape}")
        t = C.c_int()
        self.api.call("read_buffer_async", self._ctx, which, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(t))
        return t.value

    def read_wait(self, ticket: int):
        self.api.call("read_wait", self._ctx, int(ticket))

    def read_into(self, which, out):
        """field.to_numpy() into an array the caller keeps (any C-contiguous array of the buffer's shape and dtype)."""
        shape, dt = self._shape(which)
        if out.shape != tuple(shape) or out.dtype != np.dtype(dt) or not out.flags.c_contiguous:
            raise ValueError(f"expected a C-contiguous {dt} array of shape {shape}")
        self.api.call("read_buffer", self._ctx, which, out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def _write(self, which, arr):
        shape, dt = self._shape(which)
        a = np.ascontiguousarray(arr, dtype=dt)
        if a.shape != shape:
            raise ValueError(f"expected shape {shape}, got {a.shape}")
        self.api.call("write_buffer", self._ctx, which, a.ctypes.data_as(C.c_void_p), a.nbytes)

    @property
    def image_buffer(self):
        return self._read(BUF_IMAGE_BUFFER)

    @image_buffer.setter
    def image_buffer(self, arr):
        self._write(BUF_IMAGE_BUFFER, arr)

    @property
    def image_pixels(self):
        return self._read(BUF_IMAGE_PIXELS)

    @property
    def ray_buffer(self):
        """(W,H,10) float32 view of Ray; the last column holds the int32 depth bit pattern."""
        return self._read(BUF_RAY_BUFFER)

    @ray_buffer.setter
    def ray_buffer(self, arr):
        self._write(BUF_RAY_BUFFER, arr)

    @property
    def diff_buffer(self):
        """adaptive-sampling statistics (src/fileds.py:21): (sum of display change, count)"""
        return self._read(BUF_DIFF_BUFFER)

    @property
    def diff_pixels(self):
        return self._read(BUF_DIFF_PIXELS)

    def ray_depth(self):
        return self.ray_buffer[..., 9].view(np.int32)

    # ------------------------------------------------------------ measurement
    def counters(self) -> Counters:
        c = Counters()
        self.api.call("get_counters", self._ctx, C.byref(c))
        return c

    def counter(self, name: str) -> int:
        """one named work counter of the last sample() call; beyond Counters: "mlp_wave_evals", "mlp_lane_evals"."""
        v = C.c_uint64()
        self.api.call("get_counter", self._ctx, name.encode(), C.byref(v))
        return int(v.value)

    def last_sample_ms(self):
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        self.api.call("last_sample_ms", self._ctx, C.byref(a), C.byref(b), C.byref(n))
        return a.value, b.value, n.value

    def last_primary_ms(self):
        """(device ms of the primary_rays launches of the last sample(), number of launches)."""
        a, n = C.c_float(), C.c_int()
        self.api.call("last_primary_ms", self._ctx, C.byref(a), C.byref(n))
        return a.value, n.value

    # ------------------------------------------------------------ multi-GPU helpers
    def packed_bytes(self) -> int:
        n = C.c_size_t()
        self.api.call("packed_bytes", self._ctx, C.byref(n))
        return n.value

    def pack_tiles(self, device_ptr: int):
        self.api.call("pack_tiles", self._ctx, C.c_void_p(device_ptr))

    def unpack_tiles(self, device_ptr: int, src_rank: int):
        self.api.call("unpack_tiles", self._ctx, C.c_void_p(device_ptr), int(src_rank))

    # RCCL directly, without torch.distributed (include/rtpbr.h "the ONE collective")
    def rccl_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.api.call("rccl_unique_id", buf, 128)
        return buf.raw

    def rccl_init(self, unique_id: bytes, rank: int, world: int):
        """collective over all ranks (ncclCommInitRank); rank/world as given to set_tiles"""
        self.api.call("rccl_init", self._ctx, C.c_char_p(unique_id), len(unique_id), int(rank), int(world))

    def gather_tiles(self):
        """collective: pack -> one ncclGather to rank 0 -> rank 0 unpacks, on this context's stream"""
        self.api.call("gather_tiles", self._ctx)

    def rccl_info(self):
        """(ranks in this context's communicator, its rank, RCCL version) as RCCL reports them"""
        n, r, v = C.c_int(), C.c_int(), C.c_int()
        self.api.call("rccl_info", self._ctx, C.byref(n), C.byref(r), C.byref(v))
        return n.value, r.value, v.value

    def stream(self) -> int:
        s = C.c_void_p()
        self.api.call("get_stream", self._ctx, C.byref(s))
        return s.value or 0

    def close(self):
        """rtpbr_destroy: frees the device buffers AND every host_array() block — arrays obtained from host_array() must not be
        used afterwards (their memory is gone)."""
        if self._ctx:
            self.api.call("destroy", self._ctx)
            self._ctx = C.c_void_p()
            self._host_arrays.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def display_image(image_pixels: np.ndarray) -> np.ndarray:
    """(W,H,3) field layout -> (H,W,3) top-down image, the transform ti.tools.imwrite /
    canvas.set_image apply (src/main.py:55)."""
    return np.ascontiguousarray(np.swapaxes(image_pixels, 0, 1)[::-1])
