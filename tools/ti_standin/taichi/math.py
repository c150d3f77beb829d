"""`taichi.math` of the STAND-IN runtime (see taichi/__init__.py: this is NOT Taichi).

Vector / matrix algebra only, with Taichi's typing rules:
  * inside a @ti.func / @ti.kernel every real value is IEEE f32 (np.float32; correctly rounded
    + - * / sqrt), Python number literals are "weak" (they take the type of the other operand,
    and constant-only sub-expressions are evaluated by Python in double, as Taichi's AST
    transformer does), integers are i32;
  * in Python scope (module level of the reference's files) vectors hold Python doubles, like
    taichi.Matrix does outside kernels;
  * sums are evaluated left to right with no fused multiply-add (taichi.Matrix.sum,
    Matrix.__matmul__, dot = (a*b).sum(), norm = sqrt(norm_sqr), normalized = v / norm).
Transcendentals are NumPy's float32 ones (an implementation independent of the repo's own).
Nothing here restates arithmetic of the reference: the reference's own function bodies are
what runs on top of this."""
import builtins as _bi

import numpy as np

from . import _rt

np.seterr(all="ignore")
f32 = np.float32
pi = 3.141592653589793          # taichi.math.pi is the Python double
e = 2.718281828459045
_SW = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _dt():
    return np.float32 if _rt.depth > 0 else np.float64


def _is_real(x):
    return isinstance(x, (float, np.floating))


def _scal(x):
    """a scalar as it takes part in arithmetic in the current scope"""
    if _rt.depth > 0:
        if isinstance(x, (np.float32,)):
            return x
        if isinstance(x, (bool, np.bool_)):
            return np.float32(1.0 if x else 0.0)
        return np.float32(x)
    return float(x)


class Vector:
    """immutable n-vector"""
    __array_ufunc__ = None
    __slots__ = ("_d",)

    def __init__(self, arr):
        object.__setattr__(self, "_d", arr)

    # -- element access
    def __len__(self):
        return self._d.shape[0]

    def __getitem__(self, i):
        v = self._d[int(i)]
        return v if _rt.depth > 0 else float(v)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getattr__(self, name):
        try:
            idx = [_SW[ch] for ch in name]
        except KeyError:
            raise AttributeError(name) from None
        if len(idx) == 1:
            return self[idx[0]]
        return Vector(self._d[idx].copy())

    def __setattr__(self, name, value):
        if name in _SW and _rt.depth == 0:           # Python-scope convenience (src/main.py: direction.y = 0)
            b = self._d.copy(); b[_SW[name]] = value
            object.__setattr__(self, "_d", b)
            return
        raise AttributeError("stand-in vectors are immutable inside kernels")

    def _arr(self):
        dt = _dt()
        return self._d if self._d.dtype == dt else self._d.astype(dt)

    @staticmethod
    def _other(o, n):
        if isinstance(o, Vector):
            if len(o) != n:
                raise TypeError("vector size mismatch")
            return o._arr()
        if isinstance(o, (Matrix,)):
            raise TypeError("vector (op) matrix")
        return _dt()(o)

    def _bin(self, o, fn, rev=False):
        a = self._arr()
        b = Vector._other(o, len(self))
        return Vector(fn(b, a) if rev else fn(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __rpow__(self, o): return self._bin(o, np.power, True)
    def __neg__(self): return Vector(-self._arr())
    def __pos__(self): return self
    def __abs__(self): return Vector(np.abs(self._arr()))

    def __matmul__(self, m):
        # Vector @ Matrix: row vector times matrix, (v@M)_j = sum_i v_i M_ij (Taichi evaluates it
        # as transpose(M) @ v)
        if not isinstance(m, Matrix):
            return NotImplemented
        return m.transpose() @ self

    # -- reductions (taichi.Matrix.sum/max/min: left to right)
    def sum(self):
        a = self._arr()
        r = a[0]
        for i in range(1, a.shape[0]):
            r = r + a[i]
        return r if _rt.depth > 0 else float(r)

    def max(self):
        a = self._arr()
        r = a[0]
        for i in range(1, a.shape[0]):
            r = r if r >= a[i] else a[i]
        return r if _rt.depth > 0 else float(r)

    def min(self):
        a = self._arr()
        r = a[0]
        for i in range(1, a.shape[0]):
            r = r if r <= a[i] else a[i]
        return r if _rt.depth > 0 else float(r)

    def norm_sqr(self): return (self * self).sum()
    def norm(self): return sqrt(self.norm_sqr())
    def normalized(self): return self / self.norm()
    def dot(self, o): return (self * o).sum()
    def cross(self, o): return cross(self, o)

    def to_numpy(self): return np.array(self._d, dtype=np.float32)
    def __repr__(self): return "vec%d(%s)" % (len(self), ", ".join(repr(float(x)) for x in self._d))
    def __deepcopy__(self, memo): return self     # immutable


class Matrix:
    """immutable n x m matrix, row major"""
    __array_ufunc__ = None
    __slots__ = ("_d",)

    def __init__(self, arr):
        object.__setattr__(self, "_d", arr)

    def _arr(self):
        dt = _dt()
        return self._d if self._d.dtype == dt else self._d.astype(dt)

    def __getitem__(self, ij):
        i, j = ij
        v = self._d[int(i), int(j)]
        return v if _rt.depth > 0 else float(v)

    def transpose(self): return Matrix(np.ascontiguousarray(self._d.T))

    def __matmul__(self, o):
        a = self._arr()
        if isinstance(o, Vector):
            b = o._arr()
            acc = a[:, 0] * b[0]
            for k in range(1, a.shape[1]):
                acc = acc + a[:, k] * b[k]
            return Vector(acc)
        if isinstance(o, Matrix):
            b = o._arr()
            acc = a[:, 0:1] * b[0:1, :]
            for k in range(1, a.shape[1]):
                acc = acc + a[:, k:k + 1] * b[k:k + 1, :]
            return Matrix(acc)
        return NotImplemented

    def _bin(self, o, fn, rev=False):
        a = self._arr()
        b = o._arr() if isinstance(o, Matrix) else _dt()(o)
        return Matrix(fn(b, a) if rev else fn(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __neg__(self): return Matrix(-self._arr())
    def to_numpy(self): return np.array(self._d, dtype=np.float32)
    def __repr__(self): return "mat(%r)" % (self._d.tolist(),)
    def __deepcopy__(self, memo): return self


class _VecType:
    """vec2 / vec3 / vec4: constructor, annotation and .field() factory"""

    def __init__(self, n):
        self.n = n

    def __call__(self, *args):
        flat = []
        for x in args:
            if isinstance(x, Vector):
                flat.extend(x._arr().tolist() if _rt.depth == 0 else list(x._arr()))
            elif isinstance(x, (tuple, list)):
                flat.extend(x)
            else:
                flat.append(x)
        if len(flat) == 1:
            flat = flat * self.n
        if len(flat) != self.n:
            raise TypeError("vec%d from %d components" % (self.n, len(flat)))
        return Vector(np.array([_scal(v) for v in flat], dtype=_dt()))

    def field(self, shape=None, **kw):
        from . import _fields
        return _fields.VectorField(self.n, shape)

    def zero(self):
        return Vector(np.zeros(self.n, dtype=np.float32))

    def cast(self, v):
        if not isinstance(v, Vector):
            v = self(v)
        if len(v) != self.n:
            raise TypeError("vec%d <- vec%d" % (self.n, len(v)))
        return Vector(v._d.astype(np.float32))


class _MatType:
    def __init__(self, n):
        self.n = n

    def __call__(self, *args):
        n = self.n
        if len(args) == n and all(isinstance(r, Vector) for r in args):      # rows
            rows = [r._arr() for r in args]
            return Matrix(np.array(rows, dtype=_dt()))
        flat = []
        for x in args:
            if isinstance(x, (tuple, list)):
                flat.extend(x)
            else:
                flat.append(x)
        if len(flat) == 1:
            flat = flat * (n * n)
        if len(flat) != n * n:
            raise TypeError("mat%d from %d components" % (n, len(flat)))
        return Matrix(np.array([_scal(v) for v in flat], dtype=_dt()).reshape(n, n))

    def zero(self):
        return Matrix(np.zeros((self.n, self.n), dtype=np.float32))

    def cast(self, v):
        if not isinstance(v, Matrix):
            v = self(v)
        return Matrix(v._d.astype(np.float32))


vec2, vec3, vec4 = _VecType(2), _VecType(3), _VecType(4)
mat2, mat3, mat4 = _MatType(2), _MatType(3), _MatType(4)


# ---------------------------------------------------------------- element-wise functions
def _unary(npfn):
    def f(x):
        if isinstance(x, Vector):
            return Vector(npfn(x._arr()))
        if isinstance(x, Matrix):
            return Matrix(npfn(x._arr()))
        if _rt.depth > 0:
            return npfn(np.float32(x))
        return float(npfn(float(x)))
    return f


sin, cos, tan = _unary(np.sin), _unary(np.cos), _unary(np.tan)
asin, acos, atan = _unary(np.arcsin), _unary(np.arccos), _unary(np.arctan)
exp, log, sqrt = _unary(np.exp), _unary(np.log), _unary(np.sqrt)
floor, ceil, sign = _unary(np.floor), _unary(np.ceil), _unary(np.sign)


def atan2(y, x):
    if isinstance(y, Vector) or isinstance(x, Vector):
        n = len(y) if isinstance(y, Vector) else len(x)
        return Vector(np.arctan2(Vector._other(y, n), Vector._other(x, n)))
    if _rt.depth > 0:
        return np.arctan2(np.float32(y), np.float32(x))
    return float(np.arctan2(float(y), float(x)))


def _minmax(npfn, pyfn):
    def f(*args):
        if len(args) == 1:
            args = tuple(args[0])
        r = args[0]
        for o in args[1:]:
            if isinstance(r, Vector) or isinstance(o, Vector):
                n = len(r) if isinstance(r, Vector) else len(o)
                r = Vector(npfn(Vector._other(r, n), Vector._other(o, n)))
            elif isinstance(r, np.floating) or isinstance(o, np.floating):
                r = npfn(np.float32(r), np.float32(o))
            else:
                r = pyfn(r, o)          # Python numbers (module-level constants)
        return r
    return f


max = _minmax(np.maximum, _bi.max)
min = _minmax(np.minimum, _bi.min)


def clamp(x, xmin, xmax):       # taichi.math.clamp
    return max(xmin, min(xmax, x))


def mix(x, y, a):               # taichi.math.mix
    return x * (1.0 - a) + y * a


def radians(x):                 # taichi.math.radians
    return x * pi / 180


def degrees(x):
    return x * 180 / pi


def dot(a, b): return (a * b).sum()
def length(v): return sqrt((v * v).sum())
def normalize(v): return v / length(v)
def distance(a, b): return length(a - b)


def cross(a, b):
    if len(a) != 3 or len(b) != 3:
        raise TypeError("cross needs vec3")
    x, y = a._arr(), b._arr()
    return Vector(np.array([x[1] * y[2] - x[2] * y[1],
                            x[2] * y[0] - x[0] * y[2],
                            x[0] * y[1] - x[1] * y[0]], dtype=_dt()))


def reflect(i, n): return i - 2.0 * dot(n, i) * n
