"""Fields of the stand-in runtime: dense scalar / vector / struct fields, zero-initialised."""
import copy
import itertools

import numpy as np

from . import _rt
from .math import Vector, Matrix, _VecType


class I32(int):
    """Taichi i32 as seen by arithmetic: int (op) int -> i32, int (op) real -> f32 (default_fp)."""
    __slots__ = ()

    @staticmethod
    def _f(o):
        return isinstance(o, (float, np.floating))

    def _r(self, o, iop, fop, rev=False):
        if I32._f(o):
            a, b = np.float32(int(self)), np.float32(o)
            return fop(b, a) if rev else fop(a, b)
        if isinstance(o, (int, np.integer)) and not isinstance(o, bool):
            r = iop(int(o), int(self)) if rev else iop(int(self), int(o))
            return I32(r)
        return NotImplemented

    def __add__(self, o): return self._r(o, int.__add__, np.add)
    def __radd__(self, o): return self._r(o, int.__add__, np.add, True)
    def __sub__(self, o): return self._r(o, int.__sub__, np.subtract)
    def __rsub__(self, o): return self._r(o, int.__sub__, np.subtract, True)
    def __mul__(self, o): return self._r(o, int.__mul__, np.multiply)
    def __rmul__(self, o): return self._r(o, int.__mul__, np.multiply, True)
    def __neg__(self): return I32(-int(self))

    def __truediv__(self, o):
        a, b = np.float32(int(self)), np.float32(o)
        return a / b

    def __rtruediv__(self, o):
        return np.float32(o) / np.float32(int(self))


def _idx(key):
    if key is None:
        return ()
    if isinstance(key, tuple):
        return tuple(int(k) for k in key)
    return (int(key),)


class _FieldBase:
    shape = None

    def _place(self, shape):
        self.shape = tuple(int(s) for s in shape)
        self._alloc()

    def _indices(self):
        it = None
        if _rt.pixels is not None:
            it = _rt.pixels(self)
        if it is None:
            it = itertools.product(*[range(s) for s in self.shape])
        for ix in it:
            if _rt.on_index is not None:
                _rt.on_index(ix)
            yield I32(ix[0]) if len(ix) == 1 else tuple(I32(v) for v in ix)

    def __iter__(self):
        return self._indices()


class ScalarField(_FieldBase):
    def __init__(self, dtype, shape=None):
        self.dtype = dtype
        if shape is not None:
            self._place(shape if isinstance(shape, (tuple, list)) else (shape,))

    def _alloc(self):
        self.data = np.zeros(self.shape, dtype=np.float32 if self.dtype is float else np.int32)

    def __getitem__(self, key):
        v = self.data[_idx(key)]
        return np.float32(v) if self.dtype is float else I32(int(v))

    def __setitem__(self, key, v):
        self.data[_idx(key)] = v

    def to_numpy(self): return self.data.copy()
    def from_numpy(self, a): self.data[...] = a
    def fill(self, v): self.data[...] = v


class VectorField(_FieldBase):
    def __init__(self, n, shape=None):
        self.n = n
        if shape is not None:
            self._place(shape if isinstance(shape, (tuple, list)) else (shape,))

    def _alloc(self):
        self.data = np.zeros(self.shape + (self.n,), dtype=np.float32)

    def __getitem__(self, key):
        return Vector(self.data[_idx(key)].copy())

    def __setitem__(self, key, v):
        if not isinstance(v, Vector):
            v = _VecType(self.n)(v)
        self.data[_idx(key)] = v._d.astype(np.float32)

    def to_numpy(self): return self.data.copy()
    def from_numpy(self, a): self.data[...] = np.asarray(a, dtype=np.float32)
    def fill(self, v): self.data[...] = (v._d.astype(np.float32) if isinstance(v, Vector) else v)


class StructField(_FieldBase):
    """elements are created lazily (a 768x432 Ray field costs nothing until touched)"""

    def __init__(self, cls, shape=None):
        self.cls = cls
        self.cells = {}
        if shape is not None:
            self._place(shape if isinstance(shape, (tuple, list)) else (shape,))

    def _alloc(self):
        self.cells = {}

    def __getitem__(self, key):
        k = _idx(key)
        c = self.cells.get(k)
        if c is None:
            c = self.cells[k] = self.cls()
        return c                      # live element: field[i].member = v writes through

    def __setitem__(self, key, v):
        if not isinstance(v, self.cls):
            raise TypeError("struct field element type")
        self.cells[_idx(key)] = copy.deepcopy(v)


class _Dense:
    def __init__(self, shape):
        self.shape = shape

    def place(self, *fields):
        for f in fields:
            f._place(self.shape)


class Root:
    def dense(self, axes, shape):
        if not isinstance(shape, (tuple, list)):
            shape = (shape,)
        return _Dense(tuple(shape))
