"""STAND-IN for the `taichi` package — this is NOT Taichi and shares no code with it.

Purpose (tools/ref_crosscheck.py): Taichi is not installable in the build container, so the
reference's own `@ti.func` / `@ti.kernel` bodies under /root/reference are executed by the Python
interpreter on top of this minimal runtime, and their inputs/outputs are written out as numeric
fixtures (tests/golden/ref_*.npz) that pin the repo's CPU oracle to the reference's actual code.

What the runtime provides — Taichi's *language rules*, nothing of the reference's arithmetic:
  * decorators: func/kernel (struct arguments are passed by value; values stored into local
    variables take Taichi's types: Python float -> f32, struct -> copied), dataclass /
    types.struct (typed fields, zero initialised), data_oriented, static, template;
  * f32 / i32 typing with weak Python literals (taichi/math.py), fields (taichi/_fields.py);
  * ti.random() -> a hook (the cross-check script plugs in the repo's counter-based stream so
    that the reference code and the oracle consume identical random numbers);
  * struct-for loops iterate in Python, optionally over a pixel subset (hook);
  * ui / tools stubs so that module-level GUI loops in the example scripts fall through.
Known differences from real Taichi: no parallelism, no fast-math (plain IEEE f32, left-to-right
sums), NumPy's float32 transcendentals."""
import ast
import copy
import functools
import inspect
import textwrap
import types as _pytypes

import numpy as np

from . import _rt
from . import math
from ._fields import I32, ScalarField, VectorField, StructField, Root
from .math import Matrix, _VecType, _MatType

f32, f64, i32, u32, u8 = float, float, int, int, int
cpu, gpu, cuda, vulkan = "cpu", "gpu", "cuda", "vulkan"
i, j, k, ij, ijk = "i", "j", "k", "ij", "ijk"
root = Root()


def init(*a, **kw):
    return None


def static(x, *rest):
    return x if not rest else (x,) + rest


def template():
    return object


def random(dtype=float):
    if _rt.rng is None:
        raise RuntimeError("stand-in ti.random(): no stream installed")
    return np.float32(_rt.rng())


# ---------------------------------------------------------------- local-variable typing
def _store(v):
    """value stored into a local variable inside a kernel: Taichi creates a typed variable"""
    t = type(v)
    if t is float:
        return np.float32(v)
    if t is tuple:
        return tuple(_store(x) for x in v)
    if isinstance(v, Struct):
        return copy.deepcopy(v)
    if t is np.float64:
        return np.float32(v)
    return v


class _Stores(ast.NodeTransformer):
    @staticmethod
    def _local(t):
        if isinstance(t, ast.Name):
            return True
        if isinstance(t, (ast.Tuple, ast.List)):
            return all(_Stores._local(e) for e in t.elts)
        return False

    @staticmethod
    def _is_static(node):
        return (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "static")

    def visit_Assign(self, node):
        self.generic_visit(node)
        if all(self._local(t) for t in node.targets) and not self._is_static(node.value):
            node.value = ast.Call(ast.Name("__ti_store__", ast.Load()), [node.value], [])
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.target, ast.Name):
            load = ast.Name(node.target.id, ast.Load())
            val = ast.Call(ast.Name("__ti_store__", ast.Load()), [ast.BinOp(load, node.op, node.value)], [])
            return ast.copy_location(ast.Assign([ast.Name(node.target.id, ast.Store())], val), node)
        return node

    def visit_FunctionDef(self, node):      # nested defs are left alone
        return node


def _retype(fn):
    try:
        src = textwrap.dedent(inspect.getsource(fn))
    except (OSError, TypeError):
        return fn
    tree = ast.parse(src)
    fdef = tree.body[0]
    if not isinstance(fdef, ast.FunctionDef):
        return fn
    fdef.decorator_list = []
    tr = _Stores()
    fdef.body = [tr.visit(s) for s in fdef.body]
    ast.fix_missing_locations(tree)
    g = fn.__globals__
    g["__ti_store__"] = _store
    loc = {}
    code = compile(tree, inspect.getsourcefile(fn) or "<ti-standin>", "exec")
    exec(code, g, loc)
    new = loc[fdef.name]
    new.__defaults__ = fn.__defaults__
    return new


def _scoped(fn):
    body = _retype(fn)

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        args = tuple(copy.deepcopy(a) if isinstance(a, Struct) else a for a in args)
        _rt.depth += 1
        try:
            return body(*args, **kw)
        finally:
            _rt.depth -= 1
    wrapper.__ti_body__ = body
    return wrapper


func = _scoped
kernel = _scoped
pyfunc = _scoped


def data_oriented(cls):
    return cls


# ---------------------------------------------------------------- structs
class Struct:
    _fields = ()          # ((name, type), ...)

    def __init__(self, *args, **kw):
        names = [n for n, _ in self._fields]
        if len(args) > len(names):
            raise TypeError("too many struct members")
        given = dict(zip(names, args))
        for key in kw:
            if key not in names or key in given:
                raise TypeError("bad struct member " + key)
        given.update(kw)
        for n, t in self._fields:
            if n in given:
                setattr(self, n, given[n])
            else:
                object.__setattr__(self, n, _zero(t))

    def __setattr__(self, name, value):
        for n, t in self._fields:
            if n == name:
                object.__setattr__(self, name, _cast(t, value))
                return
        raise AttributeError("struct has no member " + name)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % (n, getattr(self, n)) for n, _ in self._fields))

    @classmethod
    def field(cls, shape=None, **kw):
        return StructField(cls, shape)


def _zero(t):
    if t is float:
        return np.float32(0.0)
    if t is int:
        return I32(0)
    if t is bool:
        return False
    if isinstance(t, (_VecType, _MatType)):
        return t.zero()
    if isinstance(t, type) and issubclass(t, Struct):
        return t()
    raise TypeError("unsupported struct member type %r" % (t,))


def _cast(t, v):
    if t is float:
        return np.float32(v)
    if t is int:
        return I32(int(v))
    if t is bool:
        return bool(v)
    if isinstance(t, (_VecType, _MatType)):
        return t.cast(v)
    if isinstance(t, type) and issubclass(t, Struct):
        if not isinstance(v, t):
            raise TypeError("struct member type mismatch")
        return copy.deepcopy(v)
    raise TypeError("unsupported struct member type %r" % (t,))


def _make_struct(name, fields, methods):
    ns = dict(methods)
    ns["_fields"] = tuple(fields)
    return type(name, (Struct,), ns)


def dataclass(cls):
    fields = list(cls.__dict__.get("__annotations__", {}).items())
    methods = {k: v for k, v in cls.__dict__.items()
               if callable(v) and not (k.startswith("__") and k.endswith("__"))}
    return _make_struct(cls.__name__, fields, methods)


class types:
    @staticmethod
    def struct(**fields):
        return _make_struct("struct", list(fields.items()), {})

    @staticmethod
    def vector(n, dtype=float):
        return _VecType(n)

    @staticmethod
    def matrix(n, m, dtype=float):
        return _MatType(n)


# ---------------------------------------------------------------- fields
def field(dtype=float, shape=None, **kw):
    return ScalarField(float if dtype is float else int, shape)


class _VectorNS:
    @staticmethod
    def field(n, dtype=float, shape=None, **kw):
        return VectorField(n, shape)

    def __call__(self, comps, dt=None):
        return _VecType(len(comps))(*comps)


Vector = _VectorNS()


# ---------------------------------------------------------------- tools / ui stubs
class tools:
    @staticmethod
    def imread(path, channels=0):
        if _rt.imread is None:
            raise RuntimeError("stand-in ti.tools.imread(%r): no image source installed" % (path,))
        return _rt.imread(path)

    @staticmethod
    def imwrite(img, path):
        return None


from . import ui  # noqa: E402  (needs `math`)
