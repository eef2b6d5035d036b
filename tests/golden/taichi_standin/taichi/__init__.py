'''
A pure-Python stand-in for the `taichi` package, just big enough to import the Taichi-free LOGIC of
archibate/ptina's per-sample math (`@ti.func` bodies are ordinary Python once the decorators are
identities) and evaluate it on numpy scalars.

TEST TOOLING, build container only.  Used by tests/golden/make_reference_l1_golden.py to run the
reference's own function bodies (materials/disney.py, materials/microfacet.py, geometries.py,
common.py, sampling/__init__.py, engine/path.py:11-15) on seeded inputs and to record inputs and
outputs in tests/golden/reference_l1.npz.  Neither this package nor the reference is needed at
test time: only the .npz travels.

What this is NOT: Taichi.  Scalars are numpy scalars of one chosen dtype (IEEE single operations,
numpy's libm), vectors are lists of such scalars combined entry by entry in the order Taichi's
Python-scope Matrix class (taichi/lang/matrix.py) unrolls them -- sum() left to right, dot = (a*b).sum(),
normalized = (1 / norm) * v, A @ v accumulating k = 0, 1, 2.  Taichi's compiler (type inference of
locals, fast-math, backend intrinsics) is not reproduced, so vectors made this way cross-check the
restatement's LOGIC against the reference's source; parity with real PTina output stays unpinned.
'''

import math
import sys
import types

import numpy as np

f32, f64 = np.float32, np.float64
i32, u32, i64, u64 = np.int32, np.uint32, np.int64, np.uint64
cpu, cuda, opengl, cc = 'cpu', 'cuda', 'opengl', 'cc'

_fp = [np.float32]


def set_default_fp(t):
    _fp[0] = t


def _s(x):
    '''a Python number entering a math function takes the default float type (Taichi: default_fp)'''
    if isinstance(x, (bool, int, float)):
        return _fp[0](x)
    return x


class _Runtime:
    materialized = True
    default_ip = i32
    default_fp = f32


_rt = _Runtime()


def get_runtime():
    return _rt


impl = types.SimpleNamespace(get_runtime=get_runtime)


def init(*a, **k):
    pass


def inside_kernel():
    return True


def get_os_name():
    return 'linux'


def materialize_callback(f):
    f()
    return f


def _identity(f=None, *a, **k):
    return f


func = kernel = pyfunc = data_oriented = _identity


def template():
    return None


ext_arr = template          # annotation of kernels that take numpy arrays (never called here)


def static(x, *xs):
    return [x] + list(xs) if xs else x


def static_assert(*a, **k):
    pass


def _unary(npf):
    def f(x):
        if isinstance(x, Matrix):
            return Matrix([f(e) for e in x.entries], x.n, x.m)
        with np.errstate(all='ignore'):
            return npf(_s(x))
    return f


sqrt, sin, cos, tan, exp, log = map(_unary, (np.sqrt, np.sin, np.cos, np.tan, np.exp, np.log))
floor, ceil = _unary(np.floor), _unary(np.ceil)
asin, acos, tanh = _unary(np.arcsin), _unary(np.arccos), _unary(np.tanh)


def atan2(y, x):
    return np.arctan2(_s(y), _s(x))


def pow(a, b):          # noqa: A001
    with np.errstate(all='ignore'):
        return _s(a) ** b


def _ew2(pyf):
    def f(a, b, *more):
        if more:
            return f(f(a, b), *more)
        if isinstance(a, Matrix) or isinstance(b, Matrix):
            A = a.entries if isinstance(a, Matrix) else [a] * len(b.entries)
            B = b.entries if isinstance(b, Matrix) else [b] * len(a.entries)
            ref = a if isinstance(a, Matrix) else b
            return Matrix([pyf(x, y) for x, y in zip(A, B)], ref.n, ref.m)
        return pyf(a, b)
    return f


min = _ew2(lambda x, y: x if x < y else y)      # noqa: A001  (kernel-scope min/max are element-wise)
max = _ew2(lambda x, y: x if x > y else y)      # noqa: A001


def cast(x, t):
    if isinstance(x, Matrix):
        return Matrix([cast(e, t) for e in x.entries], x.n, x.m)
    if t in (u32, i32, u64, i64):
        bits = np.dtype(t).itemsize * 8
        v = int(x) & ((1 << bits) - 1)
        if t in (i32, i64) and v >= 1 << (bits - 1):
            v -= 1 << bits
        return t(v)
    return t(x)


def bit_cast(x, t):
    return np.array(x).view(t)[()]


def random(*a):
    raise RuntimeError('ti.random() has no stand-in: pass explicit samples')


def expr_init(x):
    if isinstance(x, Matrix):
        return Matrix(list(x.entries), x.n, x.m)
    return x


expr_init_func = expr_init


def assign(a, b):
    raise RuntimeError('ti.assign outside a kernel')


class Matrix:
    '''entries in row-major order; a vector is n x 1 (Taichi: ti.Vector(xs) = Matrix of shape (len, 1))'''
    is_taichi_class = True
    __array_ufunc__ = None          # numpy scalars defer to our reflected operators

    def __init__(self, xs, n=None, m=None):
        if isinstance(xs, Matrix):
            xs, n, m = list(xs.entries), xs.n, xs.m
        xs = list(xs)
        if xs and isinstance(xs[0], (list, tuple)):
            n, m = len(xs), len(xs[0])
            xs = [e for row in xs for e in row]
        self.entries = xs
        self.n = len(xs) if n is None else n
        self.m = 1 if m is None else m

    # ---- construction helpers used by the reference
    @staticmethod
    def empty(n, m):
        return Matrix([None] * (n * m), n, m)

    @staticmethod
    def cols(cs):
        n, m = cs[0].n, len(cs)
        return Matrix([cs[j].entries[i] for i in range(n) for j in range(m)], n, m)

    @staticmethod
    def rows(rs):
        return Matrix([e for r in rs for e in r.entries], len(rs), rs[0].n)

    @staticmethod
    def unit(n, i):
        return Matrix([1 if k == i else 0 for k in range(n)])

    def is_global(self):
        return False

    def element_wise_writeback_binary(self, f, other):
        raise RuntimeError('no in-place Taichi assignment in the stand-in')

    # ---- access
    def __len__(self):
        return len(self.entries)

    def __iter__(self):
        return iter(self.entries)

    def __getitem__(self, i):
        if isinstance(i, tuple):
            return self.entries[i[0] * self.m + i[1]]
        return self.entries[i]

    def __setitem__(self, i, v):
        if isinstance(i, tuple):
            self.entries[i[0] * self.m + i[1]] = v
        else:
            self.entries[i] = v

    def __call__(self, i, j=0):
        return self.entries[i * self.m + j]

    x = property(lambda s: s.entries[0], lambda s, v: s.entries.__setitem__(0, v))
    y = property(lambda s: s.entries[1], lambda s, v: s.entries.__setitem__(1, v))
    z = property(lambda s: s.entries[2], lambda s, v: s.entries.__setitem__(2, v))
    w = property(lambda s: s.entries[3], lambda s, v: s.entries.__setitem__(3, v))

    # ---- element-wise arithmetic (Taichi unrolls entry by entry)
    def _bin(self, other, op, reflected=False):
        if isinstance(other, Matrix):
            assert len(other.entries) == len(self.entries)
            B = other.entries
        else:
            B = [other] * len(self.entries)
        with np.errstate(all='ignore'):
            if reflected:
                return Matrix([op(b, a) for a, b in zip(self.entries, B)], self.n, self.m)
            return Matrix([op(a, b) for a, b in zip(self.entries, B)], self.n, self.m)

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: a / b, True)
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __mod__(self, o): return self._bin(o, lambda a, b: a % b)
    def __floordiv__(self, o): return self._bin(o, lambda a, b: a // b)
    def __and__(self, o): return self._bin(o, lambda a, b: a & b)
    def __rand__(self, o): return self._bin(o, lambda a, b: a & b, True)
    def __or__(self, o): return self._bin(o, lambda a, b: a | b)
    def __xor__(self, o): return self._bin(o, lambda a, b: a ^ b)
    def __lshift__(self, o): return self._bin(o, lambda a, b: a << b)
    def __rshift__(self, o): return self._bin(o, lambda a, b: a >> b)
    def __neg__(self): return Matrix([-a for a in self.entries], self.n, self.m)
    def __pos__(self): return self
    def __abs__(self): return Matrix([abs(a) for a in self.entries], self.n, self.m)
    def __lt__(self, o): return self._bin(o, lambda a, b: int(a < b))
    def __le__(self, o): return self._bin(o, lambda a, b: int(a <= b))
    def __gt__(self, o): return self._bin(o, lambda a, b: int(a > b))
    def __ge__(self, o): return self._bin(o, lambda a, b: int(a >= b))
    def __eq__(self, o): return self._bin(o, lambda a, b: int(a == b))
    def __ne__(self, o): return self._bin(o, lambda a, b: int(a != b))
    __hash__ = None

    def __matmul__(self, o):        # taichi/lang/matrix.py: acc = A(i,0)*B(0,j); acc = acc + A(i,k)*B(k,j)
        assert self.m == o.n
        out = []
        for i in range(self.n):
            for j in range(o.m):
                acc = self(i, 0) * o(0, j)
                for k in range(1, self.m):
                    acc = acc + self(i, k) * o(k, j)
                out.append(acc)
        return Matrix(out, self.n, o.m)

    # ---- reductions
    def sum(self):
        ret = self.entries[0]
        for e in self.entries[1:]:
            ret = ret + e
        return ret

    def any(self):
        ret = False
        for e in self.entries:
            ret = ret or (e != 0)
        return int(ret)

    def all(self):
        ret = True
        for e in self.entries:
            ret = ret and (e != 0)
        return int(ret)

    def max(self):
        ret = self.entries[0]
        for e in self.entries[1:]:
            ret = ret if ret > e else e
        return ret

    def min(self):
        ret = self.entries[0]
        for e in self.entries[1:]:
            ret = ret if ret < e else e
        return ret

    def dot(self, o):
        return (self * o).sum()

    def cross(self, o):
        a, b = self, o
        return Matrix([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])

    def norm_sqr(self):
        return (self ** 2).sum()

    def norm(self, eps=0):
        return sqrt(self.norm_sqr() + eps)

    def normalized(self, eps=0):
        invlen = 1 / (self.norm() + eps)
        return invlen * self

    def __repr__(self):
        return 'Matrix(%r)' % (self.entries,)


def Vector(xs, *a, **k):
    return Matrix(list(xs))


Vector.unit = lambda n, i: Matrix.unit(n, i)


# ---------------------------------------------------------------- integers with Taichi's i32 behaviour
class I32(int):
    """a Python int that wraps like Taichi's default integer (i32) under arithmetic.  Being an int
    subclass it stays "weak" for numpy: np.float32 * I32 is float32, as f32 * i32 is in Taichi."""
    __slots__ = ()

    @staticmethod
    def wrap(v):
        v = int(v) & 0xFFFFFFFF
        return v - (1 << 32) if v & 0x80000000 else v

    def __new__(cls, v=0):
        return int.__new__(cls, I32.wrap(v))


def _i32_bin(name, f, int_result=True):
    def op(a, b):
        if isinstance(b, Matrix):
            return NotImplemented
        if isinstance(b, (int, np.integer)):
            r = f(int(a), int(b))
            return I32(r) if int_result else r
        return f(int(a), b)                     # a float operand: the plain int is "weak" for numpy

    def rop(a, b):
        if isinstance(b, (int, np.integer)):
            r = f(int(b), int(a))
            return I32(r) if int_result else r
        return f(b, int(a))
    setattr(I32, '__%s__' % name, op)
    setattr(I32, '__r%s__' % name, rop)


def _i32_ufunc(self, ufunc, method, *inputs, **kw):
    # numpy sees a plain Python int ("weak": f32 * i32 stays f32, as in Taichi)
    return getattr(ufunc, method)(*[int(x) if isinstance(x, I32) else x for x in inputs], **kw)


I32.__array_ufunc__ = _i32_ufunc
for _n, _f in (('add', lambda a, b: a + b), ('sub', lambda a, b: a - b), ('mul', lambda a, b: a * b),
               ('and', lambda a, b: a & b), ('or', lambda a, b: a | b), ('xor', lambda a, b: a ^ b),
               ('lshift', lambda a, b: a << b), ('rshift', lambda a, b: a >> b),
               ('floordiv', lambda a, b: a // b), ('mod', lambda a, b: a % b)):
    _i32_bin(_n, _f)
_i32_bin('truediv', lambda a, b: a / b, int_result=False)
_i32_bin('pow', lambda a, b: a ** b, int_result=False)
for _n, _f in (('lt', lambda a, b: a < b), ('le', lambda a, b: a <= b), ('gt', lambda a, b: a > b),
               ('ge', lambda a, b: a >= b), ('eq', lambda a, b: a == b), ('ne', lambda a, b: a != b)):
    setattr(I32, '__%s__' % _n, (lambda f: lambda a, b: NotImplemented if isinstance(b, Matrix) else bool(f(int(a), b)))(_f))
I32.__hash__ = lambda a: hash(int(a))
I32.__neg__ = lambda a: I32(-int(a))
I32.__pos__ = lambda a: a
I32.__invert__ = lambda a: I32(~int(a))
I32.__abs__ = lambda a: I32(abs(int(a)))


def ti_int(x):
    """kernel-scope int(): element-wise cast to i32 (u32 values keep their bit pattern)"""
    if isinstance(x, Matrix):
        return Matrix([ti_int(e) for e in x.entries], x.n, x.m)
    if isinstance(x, (np.floating, float)):
        return I32(int(x))                     # truncation toward zero
    return I32(int(x))


def ti_float(x):
    """kernel-scope float(): element-wise cast to the default float type"""
    if isinstance(x, Matrix):
        return Matrix([ti_float(e) for e in x.entries], x.n, x.m)
    return _fp[0](x)


def atomic_max(a, b):
    old = Matrix(a) if isinstance(a, Matrix) else a
    if isinstance(a, Matrix):
        B = b.entries if isinstance(b, Matrix) else [b] * len(a.entries)
        for k, e in enumerate(B):
            a.entries[k] = a.entries[k] if a.entries[k] > e else e
        return old
    raise RuntimeError('atomic on a Python scalar local has no stand-in')


def atomic_min(a, b):
    old = Matrix(a) if isinstance(a, Matrix) else a
    if isinstance(a, Matrix):
        B = b.entries if isinstance(b, Matrix) else [b] * len(a.entries)
        for k, e in enumerate(B):
            a.entries[k] = a.entries[k] if a.entries[k] < e else e
        return old
    raise RuntimeError('atomic on a Python scalar local has no stand-in')


def ndrange(*dims):
    import itertools
    return itertools.product(*[range(int(d)) if not isinstance(d, tuple) else range(int(d[0]), int(d[1])) for d in dims])


def grouped(x):
    raise RuntimeError('ti.grouped has no stand-in')


# ---------------------------------------------------------------- fields (numpy storage)
def _np_dtype(dt):
    if dt in (int, i32, ti_int):      # `int` inside the reference's modules may be the kernel-scope cast
        return np.int32
    if dt in (float, ti_float):
        return _fp[0]
    return dt


def _wrap_scalar(v, dtype):
    if np.issubdtype(dtype, np.integer):
        return I32(int(v))
    return dtype(v)


class ScalarField:
    def __init__(self, dtype, shape=None):
        self.dtype = _np_dtype(dtype)
        self.data = None
        if shape is not None:
            self._alloc(shape)

    def _alloc(self, shape):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(d) for d in shape)
        self.data = np.zeros(self.shape, self.dtype)

    @staticmethod
    def _ix(i):
        if i is None:
            return ()
        if isinstance(i, Matrix):
            return tuple(int(e) for e in i.entries)
        if isinstance(i, tuple):
            return tuple(int(e) for e in i)
        return (int(i),)

    def __getitem__(self, i):
        return _wrap_scalar(self.data[self._ix(i)], self.dtype)

    def __setitem__(self, i, v):
        if np.issubdtype(self.dtype, np.integer):
            v = I32.wrap(int(v))
        self.data[self._ix(i)] = v

    def fill(self, v):
        self.data[...] = v

    def from_numpy(self, a):
        a = np.asarray(a)
        if np.issubdtype(self.dtype, np.integer):
            a = (a.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(a.shape)
        self.data[tuple(slice(0, d) for d in a.shape)] = a

    def to_numpy(self):
        return self.data.copy()


class FieldMatrix(Matrix):
    """the value of a vector / matrix field element: reads are plain entries, item and attribute
    assignment writes through to the field (in Taichi a field subscript is an lvalue)"""

    def __init__(self, field, ix):
        vals = [_wrap_scalar(v, field.dtype) for v in field.data[ix].reshape(-1)]
        Matrix.__init__(self, vals, field.n, field.m)
        object.__setattr__(self, '_field', field)
        object.__setattr__(self, '_ix', ix)

    def __setitem__(self, i, v):
        Matrix.__setitem__(self, i, v)
        self._field.data[self._ix] = np.array(self.entries, self._field.dtype).reshape(self._field.data[self._ix].shape)


def _fm_set(k):
    def setter(self, v):
        self[k] = v
    return setter


for _k, _nm in enumerate('xyzw'):
    setattr(FieldMatrix, _nm, property(lambda s, _k=_k: s.entries[_k], _fm_set(_k)))


class MatrixField:
    def __init__(self, n, m, dtype, shape):
        self.n, self.m = n, m
        self.dtype = _np_dtype(dtype)
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(d) for d in shape)
        self.data = np.zeros(self.shape + ((n,) if m == 1 else (n, m)), self.dtype)

    def __getitem__(self, i):
        return FieldMatrix(self, ScalarField._ix(i))

    def __setitem__(self, i, v):
        if isinstance(v, Matrix):
            v = [e for e in v.entries]
        a = np.array(v, dtype=np.float64 if not np.issubdtype(self.dtype, np.integer) else np.int64)
        self.data[ScalarField._ix(i)] = a.reshape(self.data[ScalarField._ix(i)].shape).astype(self.dtype)

    def fill(self, v):
        self.data[...] = v

    def from_numpy(self, a):
        a = np.asarray(a)
        self.data[tuple(slice(0, d) for d in a.shape)] = a

    def to_numpy(self):
        return self.data.copy()


def field(dtype, shape=None, **k):
    return ScalarField(dtype, shape)


Vector.field = lambda n, dtype, shape=None, **k: MatrixField(n, 1, dtype, shape)
Matrix.field = staticmethod(lambda n, m, dtype, shape=None, **k: MatrixField(n, m, dtype, shape))


class _SNode:
    def __init__(self, dims=()):
        self.dims = dims

    def dense(self, axes, n):
        n = n if isinstance(n, (tuple, list)) else (n,)
        return _SNode(self.dims + tuple(int(x) for x in n))

    def place(self, *fields):
        for f in fields:
            f._alloc(self.dims)


root = _SNode()
i, j, k, l = 'i', 'j', 'k', 'l'
ij, ijk = 'ij', 'ijk'
cfg = types.SimpleNamespace(arch=cuda, cpu_max_num_threads=8)


# taichi.lang.common_ops.TaichiOperations: the reference adds __pos__ to it at import
lang = types.ModuleType('taichi.lang')
common_ops = types.ModuleType('taichi.lang.common_ops')


class TaichiOperations:
    pass


common_ops.TaichiOperations = TaichiOperations
lang.common_ops = common_ops
sys.modules['taichi.lang'] = lang
sys.modules['taichi.lang.common_ops'] = common_ops

pi = math.pi
tau = math.tau
