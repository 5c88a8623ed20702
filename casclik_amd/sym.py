"""CasADi-free symbolic front-end for skill scripts.

The reference writes constraint expressions as CasADi ``MX`` graphs
(reference: casclik/constraints.py:21-24; call sites e.g.
examples/notebooks/ur5_moe2016_example2.ipynb cells 4-8).  CasADi does not exist
on the target, so this module provides the small subset of the ``casadi``
module surface those scripts touch (``MX.sym``, ``vertcat``, ``mtimes``,
slicing, ``sin/cos``, ``norm_2``, ``norm_fro``, ``Function``, ``DM`` ...)
on top of a plain scalar expression DAG.  Nothing here runs on the per-tick hot
path: the DAG is *lowered once* to the flat device skill descriptor
(casclik_amd/lowering.py) that the HIP kernels consume.  The numeric evaluator
in this file only serves user-side conveniences (``Function.__call__`` used for
logging in the notebooks) - the controllers never call it.

An ``MX`` is a 2-D numpy object array of scalar nodes; all matrix algebra is
done on that array so the lowering only ever sees scalars.
"""
from __future__ import annotations

import math
import numpy as _np

np = _np  # scripts use ``cs.np`` (reference: ur5_moe2016_example2.ipynb cell 5)
inf = float("inf")
pi = math.pi


# --------------------------------------------------------------------------
# scalar nodes
# --------------------------------------------------------------------------
class Scalar(object):
    """Node of the scalar expression DAG."""
    __slots__ = ("op", "args", "value", "name", "index", "family", "aux")

    def __init__(self, op, args=(), value=None, name=None, index=None,
                 family=None, aux=None):
        self.op = op          # 'const','sym','add','sub','mul','div','neg',
        #                       'sin','cos','sqrt','exp','log','pow','fabs',
        #                       'sign','norm2','fk','ori_err','cmp_*','if_else'
        self.args = tuple(args)
        self.value = value    # for const
        self.name = name      # for sym
        self.index = index    # element index within symbol family
        self.family = family  # SymFamily for sym
        self.aux = aux        # op-specific payload (fk: (chain, i, j))

    def is_const(self):
        return self.op == "const"

    def __repr__(self):
        if self.op == "const":
            return repr(self.value)
        if self.op == "sym":
            return "%s_%d" % (self.name, self.index)
        return "%s(%s)" % (self.op, ",".join(repr(a) for a in self.args))


class SymFamily(object):
    """Identity of one ``MX.sym`` call (all its elements share the family)."""
    __slots__ = ("name", "n")

    def __init__(self, name, n):
        self.name = name
        self.n = n

    def __repr__(self):
        return "SymFamily(%s,%d)" % (self.name, self.n)


_ZERO = Scalar("const", value=0.0)
_ONE = Scalar("const", value=1.0)


def _c(v):
    v = float(v)
    if v == 0.0:
        return _ZERO
    if v == 1.0:
        return _ONE
    return Scalar("const", value=v)


def _s_add(a, b):
    if a.is_const() and b.is_const():
        return _c(a.value + b.value)
    if a.is_const() and a.value == 0.0:
        return b
    if b.is_const() and b.value == 0.0:
        return a
    return Scalar("add", (a, b))


def _s_sub(a, b):
    if a.is_const() and b.is_const():
        return _c(a.value - b.value)
    if b.is_const() and b.value == 0.0:
        return a
    if a.is_const() and a.value == 0.0:
        return _s_neg(b)
    return Scalar("sub", (a, b))


def _s_mul(a, b):
    if a.is_const() and b.is_const():
        return _c(a.value * b.value)
    if a.is_const():
        if a.value == 0.0:
            return _ZERO
        if a.value == 1.0:
            return b
    if b.is_const():
        if b.value == 0.0:
            return _ZERO
        if b.value == 1.0:
            return a
    return Scalar("mul", (a, b))


def _s_div(a, b):
    if a.is_const() and b.is_const():
        return _c(a.value / b.value)
    if b.is_const() and b.value == 1.0:
        return a
    if a.is_const() and a.value == 0.0:
        return _ZERO
    return Scalar("div", (a, b))


def _s_neg(a):
    if a.is_const():
        return _c(-a.value)
    if a.op == "neg":
        return a.args[0]
    return Scalar("neg", (a,))


def _nan_outside(f):
    """IEEE result like CasADi's (asin(2) is nan, not an exception)"""
    def g(v):
        try:
            return f(v)
        except ValueError:
            return float("nan")
    return g


_UNARY_NUMERIC = {
    "sin": math.sin, "cos": math.cos, "sqrt": math.sqrt, "exp": math.exp,
    "log": math.log, "fabs": abs, "tan": math.tan,
    "sign": lambda v: float(v > 0) - float(v < 0),
    "asin": _nan_outside(math.asin), "acos": _nan_outside(math.acos), "atan": math.atan, "tanh": math.tanh,
}
# two-argument functions (casadi.atan2 / fmin / fmax: angle-type task errors, saturations)
_BINARY_NUMERIC = {"atan2": math.atan2, "fmin": min, "fmax": max}


def _s_binary(op, a, b):
    if a.is_const() and b.is_const():
        return _c(_BINARY_NUMERIC[op](a.value, b.value))
    return Scalar(op, (a, b))


def _s_unary(op, a):
    if a.is_const():
        return _c(_UNARY_NUMERIC[op](a.value))
    return Scalar(op, (a,))


def _s_pow(a, b):
    if a.is_const() and b.is_const():
        return _c(a.value ** b.value)
    return Scalar("pow", (a, b))


def _s_norm2(items):
    items = tuple(items)
    if all(i.is_const() for i in items):
        return _c(math.sqrt(sum(i.value * i.value for i in items)))
    return Scalar("norm2", items)


# --------------------------------------------------------------------------
# matrix type
# --------------------------------------------------------------------------
def _as_array(x):
    """Return a 2-D object array of Scalar for anything matrix-like."""
    if isinstance(x, MX):
        return x._a
    if isinstance(x, DM):
        x = x._v
    if isinstance(x, Scalar):
        out = _np.empty((1, 1), dtype=object)
        out[0, 0] = x
        return out
    arr = _np.asarray(x)
    if arr.dtype == object:
        # list mixing MX and numbers (``cs.vertcat([...])`` style)
        flat = []
        for item in _np.ravel(arr):
            sub = _as_array(item)
            if sub.shape != (1, 1):
                raise ValueError("nested non-scalar entries")
            flat.append(sub[0, 0])
        out = _np.empty(len(flat), dtype=object)
        out[:] = flat
        arr_shape = arr.shape
        out = out.reshape(arr_shape)
    else:
        out = _np.empty(arr.shape, dtype=object)
        flat_in = _np.ravel(arr)
        flat = [_c(v) for v in flat_in]
        tmp = _np.empty(len(flat), dtype=object)
        tmp[:] = flat
        out = tmp.reshape(arr.shape)
    if out.ndim == 0:
        out = out.reshape(1, 1)
    elif out.ndim == 1:
        out = out.reshape(-1, 1)   # CasADi treats 1-D data as a column
    elif out.ndim != 2:
        raise ValueError("only 0/1/2-D data supported")
    return out


def _wrap(a):
    return MX(_array=a)


def _bcast(a, b):
    if a.shape == b.shape:
        return a, b
    if a.shape == (1, 1):
        out = _np.empty(b.shape, dtype=object)
        out[:] = a[0, 0]
        return out, b
    if b.shape == (1, 1):
        out = _np.empty(a.shape, dtype=object)
        out[:] = b[0, 0]
        return a, out
    raise ValueError("Dimension mismatch %s vs %s" % (a.shape, b.shape))


def _elementwise(f, a, b):
    a, b = _bcast(_as_array(a), _as_array(b))
    out = _np.empty(a.shape, dtype=object)
    for idx in _np.ndindex(a.shape):
        out[idx] = f(a[idx], b[idx])
    return _wrap(out)


def _map(f, a):
    a = _as_array(a)
    out = _np.empty(a.shape, dtype=object)
    for idx in _np.ndindex(a.shape):
        out[idx] = f(a[idx])
    return _wrap(out)


class MX(object):
    """Matrix of scalar expressions with the slice of the casadi.MX API the
    reference's skill scripts use."""
    __array_priority__ = 1000  # numpy defers binary ops to us

    def __init__(self, *args, **kw):
        a = kw.pop("_array", None)
        if a is not None:
            self._a = a
        elif len(args) == 0:
            self._a = _np.empty((0, 1), dtype=object)
        elif len(args) == 1:
            self._a = _as_array(args[0]).copy()
        elif len(args) == 2:
            self._a = _as_array(_np.zeros((int(args[0]), int(args[1]))))
        else:
            raise TypeError("MX(...)")

    # -- constructors ----------------------------------------------------
    @staticmethod
    def sym(name, n=1, m=1):
        if isinstance(n, (tuple, list)):
            n, m = n
        fam = SymFamily(name, int(n) * int(m))
        out = _np.empty((int(n), int(m)), dtype=object)
        k = 0
        for j in range(int(m)):          # column-major numbering like CasADi
            for i in range(int(n)):
                out[i, j] = Scalar("sym", name=name, index=k, family=fam)
                k += 1
        return MX(_array=out)

    @staticmethod
    def zeros(n=1, m=1):
        if isinstance(n, (tuple, list)):
            n, m = n
        return MX(_array=_as_array(_np.zeros((int(n), int(m)))))

    @staticmethod
    def ones(n=1, m=1):
        if isinstance(n, (tuple, list)):
            n, m = n
        return MX(_array=_as_array(_np.ones((int(n), int(m)))))

    @staticmethod
    def eye(n):
        return MX(_array=_as_array(_np.eye(int(n))))

    # -- shape -----------------------------------------------------------
    def size(self, axis=None):
        if axis is None:
            return self._a.shape
        return self._a.shape[axis - 1]    # casadi: size(1)=rows, size(2)=cols

    def size1(self):
        return self._a.shape[0]

    def size2(self):
        return self._a.shape[1]

    @property
    def shape(self):
        return self._a.shape

    def numel(self):
        return self._a.size

    def nnz(self):
        return sum(1 for s in self._a.flat
                   if not (s.is_const() and s.value == 0.0))

    def is_symbolic(self):
        return all(s.op == "sym" for s in self._a.flat) and self._a.size > 0

    def is_constant(self):
        return all(s.is_const() for s in self._a.flat)

    @property
    def T(self):
        return _wrap(self._a.T.copy())

    def __len__(self):
        return self._a.shape[0]

    # -- indexing --------------------------------------------------------
    def __getitem__(self, key):
        a = self._a
        if not isinstance(key, tuple):
            if a.shape[1] == 1:
                sub = a[key, 0]
            elif a.shape[0] == 1:
                sub = a[0, key]
            else:                         # linear (column-major) indexing
                sub = a.T.reshape(-1)[key]
            if isinstance(sub, Scalar):
                return _wrap(_as_array(sub))
            return _wrap(_np.asarray(sub, dtype=object).reshape(-1, 1))
        r, c = key
        sub = a[r, c]
        if isinstance(sub, Scalar):
            return _wrap(_as_array(sub))
        sub = _np.asarray(sub, dtype=object)
        if sub.ndim == 1:
            if isinstance(r, (int, _np.integer)):
                sub = sub.reshape(1, -1)
            else:
                sub = sub.reshape(-1, 1)
        return _wrap(sub.copy())

    def __setitem__(self, key, val):
        v = _as_array(val)
        a = self._a
        if not isinstance(key, tuple):
            key = (key, 0) if a.shape[1] == 1 else (0, key)
        tgt = a[key]
        if isinstance(tgt, Scalar):
            a[key] = v[0, 0]
        else:
            tgt_shape = _np.asarray(tgt, dtype=object).shape
            if v.shape == (1, 1):
                fill = _np.empty(tgt_shape, dtype=object)
                fill[...] = v[0, 0]
                a[key] = fill
            else:
                a[key] = v.reshape(tgt_shape)

    def __iter__(self):
        for i in range(self._a.shape[0]):
            yield self[i]

    # -- arithmetic ------------------------------------------------------
    def __add__(self, o):
        return _elementwise(_s_add, self, o)

    def __radd__(self, o):
        return _elementwise(_s_add, o, self)

    def __sub__(self, o):
        return _elementwise(_s_sub, self, o)

    def __rsub__(self, o):
        return _elementwise(_s_sub, o, self)

    def __mul__(self, o):
        return _elementwise(_s_mul, self, o)

    def __rmul__(self, o):
        return _elementwise(_s_mul, o, self)

    def __truediv__(self, o):
        return _elementwise(_s_div, self, o)

    def __rtruediv__(self, o):
        return _elementwise(_s_div, o, self)

    __div__ = __truediv__
    __rdiv__ = __rtruediv__

    def __neg__(self):
        return _map(_s_neg, self)

    def __pos__(self):
        return self

    def __pow__(self, o):
        return _elementwise(_s_pow, self, o)

    def __matmul__(self, o):
        return mtimes(self, o)

    def __rmatmul__(self, o):
        return mtimes(o, self)

    # comparisons produce 0/1 valued nodes (used by user-side Functions only)
    def __lt__(self, o):
        return _elementwise(lambda a, b: Scalar("cmp_lt", (a, b)), self, o)

    def __le__(self, o):
        return _elementwise(lambda a, b: Scalar("cmp_le", (a, b)), self, o)

    def __gt__(self, o):
        return _elementwise(lambda a, b: Scalar("cmp_lt", (b, a)), self, o)

    def __ge__(self, o):
        return _elementwise(lambda a, b: Scalar("cmp_le", (b, a)), self, o)

    # `==` / `!=` build 0/1 valued nodes too, as CasADi's MX does (the reference's own multidimensional tangent-cone
    # function is written with them, pseudo_inverse.py:229-236); so an MX hashes by identity and has no truth value
    # unless it is a constant
    def __eq__(self, o):
        if o is None:
            return False
        return _elementwise(lambda a, b: Scalar("cmp_eq", (a, b)), self, o)

    def __ne__(self, o):
        if o is None:
            return True
        return _elementwise(lambda a, b: Scalar("cmp_ne", (a, b)), self, o)

    def __hash__(self):
        return id(self)

    def __bool__(self):
        try:
            value = self.toarray()
        except Exception:
            raise TypeError("the truth value of a symbolic MX expression is undefined (compare with `is`, or "
                            "evaluate it through a cs.Function)")
        return bool(_np.all(value != 0.0))

    def __repr__(self):
        if self._a.size <= 12:
            return "MX(%s)" % (self._a.tolist(),)
        return "MX(%dx%d)" % self._a.shape

    # -- numeric ---------------------------------------------------------
    def toarray(self):
        """Numeric value of a constant expression (``DM.toarray`` parity)."""
        return evaluate(self, {})


class DM(object):
    """Numeric dense matrix with the ``casadi.DM`` methods callers use
    (reference usage: ``res[0].toarray()[:,0]``,
    ur5_moe2016_example2.ipynb:539)."""
    __array_priority__ = 900

    def __init__(self, *args):
        if len(args) == 2 and all(isinstance(a, (int, _np.integer)) for a in args):
            self._v = _np.zeros((args[0], args[1]))
        elif len(args) == 1:
            v = args[0]
            if isinstance(v, DM):
                v = v._v
            v = _np.array(v, dtype=float)
            if v.ndim == 0:
                v = v.reshape(1, 1)
            elif v.ndim == 1:
                v = v.reshape(-1, 1)
            self._v = v
        elif len(args) == 0:
            self._v = _np.zeros((0, 1))
        else:
            raise TypeError("DM(...)")

    @staticmethod
    def zeros(n=1, m=1):
        if isinstance(n, (tuple, list)):
            n, m = n
        return DM(_np.zeros((int(n), int(m))))

    @staticmethod
    def ones(n=1, m=1):
        if isinstance(n, (tuple, list)):
            n, m = n
        return DM(_np.ones((int(n), int(m))))

    @staticmethod
    def eye(n):
        return DM(_np.eye(int(n)))

    def toarray(self):
        return self._v.copy()

    def full(self):
        return self._v.copy()

    def size(self, axis=None):
        return self._v.shape if axis is None else self._v.shape[axis - 1]

    def size1(self):
        return self._v.shape[0]

    def size2(self):
        return self._v.shape[1]

    @property
    def shape(self):
        return self._v.shape

    @property
    def T(self):
        return DM(self._v.T)

    def __array__(self, dtype=None, copy=None):
        return self._v if dtype is None else self._v.astype(dtype)

    def __len__(self):
        return self._v.shape[0]

    def __float__(self):
        if self._v.size != 1:
            raise TypeError("only 1x1 DM converts to float")
        return float(self._v.reshape(-1)[0])

    def __int__(self):
        return int(float(self))

    def __bool__(self):
        return bool(float(self))

    __nonzero__ = __bool__

    def __getitem__(self, key):
        v = self._v
        if not isinstance(key, tuple):
            if v.shape[1] == 1:
                sub = v[key, 0]
            elif v.shape[0] == 1:
                sub = v[0, key]
            else:
                sub = v.T.reshape(-1)[key]
            return DM(sub)
        r, c = key
        sub = _np.asarray(v[r, c])
        if sub.ndim == 1 and isinstance(r, (int, _np.integer)):
            sub = sub.reshape(1, -1)
        return DM(sub)

    def __setitem__(self, key, val):
        val = val._v if isinstance(val, DM) else _np.asarray(val, dtype=float)
        if not isinstance(key, tuple):
            key = (key, 0) if self._v.shape[1] == 1 else (0, key)
        tgt = self._v[key]
        if _np.ndim(tgt) == 0:
            self._v[key] = float(_np.reshape(val, -1)[0])
        else:
            self._v[key] = _np.reshape(val, _np.shape(tgt)) if _np.size(val) > 1 else val

    def _bin(self, o, f, swap=False):
        if isinstance(o, MX):
            return NotImplemented
        ov = o._v if isinstance(o, DM) else _np.asarray(o, dtype=float)
        if _np.ndim(ov) == 1:
            ov = ov.reshape(-1, 1)
        return DM(f(ov, self._v) if swap else f(self._v, ov))

    def __add__(self, o):
        return self._bin(o, _np.add)

    def __radd__(self, o):
        return self._bin(o, _np.add, True)

    def __sub__(self, o):
        return self._bin(o, _np.subtract)

    def __rsub__(self, o):
        return self._bin(o, _np.subtract, True)

    def __mul__(self, o):
        return self._bin(o, _np.multiply)

    def __rmul__(self, o):
        return self._bin(o, _np.multiply, True)

    def __truediv__(self, o):
        return self._bin(o, _np.divide)

    def __rtruediv__(self, o):
        return self._bin(o, _np.divide, True)

    def __neg__(self):
        return DM(-self._v)

    def __repr__(self):
        return "DM(%s)" % (self._v.tolist(),)


SX = MX  # scripts occasionally name SX; one expression type serves both


# --------------------------------------------------------------------------
# casadi-module level functions
# --------------------------------------------------------------------------
def _is_numeric(x):
    return not isinstance(x, (MX, Scalar)) and not (
        isinstance(x, (list, tuple)) and any(isinstance(i, (MX, Scalar)) for i in x))


def vertcat(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    if len(args) == 0:
        return MX()
    numeric = all(_is_numeric(a) for a in args)
    parts = []
    for a in args:
        arr = _as_array(a)
        if arr.size == 0:
            continue
        parts.append(arr)
    if not parts:
        return DM() if numeric else MX()
    out = _wrap(_np.vstack(parts))
    return DM(evaluate(out, {})) if numeric else out


def horzcat(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    numeric = all(_is_numeric(a) for a in args)
    parts = [_as_array(a) for a in args if _as_array(a).size]
    out = _wrap(_np.hstack(parts))
    return DM(evaluate(out, {})) if numeric else out


def mtimes(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = tuple(args[0])
    res = args[0]
    for nxt in args[1:]:
        res = _mtimes2(res, nxt)
    return res


def _mtimes2(a, b):
    numeric = _is_numeric(a) and _is_numeric(b)
    A = _as_array(a)
    B = _as_array(b)
    if A.shape == (1, 1) or B.shape == (1, 1):
        out = _elementwise(_s_mul, _wrap(A), _wrap(B))
    else:
        if A.shape[1] != B.shape[0]:
            raise ValueError("mtimes: incompatible dimensions %s x %s"
                             % (A.shape, B.shape))
        out_a = _np.empty((A.shape[0], B.shape[1]), dtype=object)
        for i in range(A.shape[0]):
            for j in range(B.shape[1]):
                acc = _ZERO
                for k in range(A.shape[1]):
                    acc = _s_add(acc, _s_mul(A[i, k], B[k, j]))
                out_a[i, j] = acc
        out = _wrap(out_a)
    return DM(evaluate(out, {})) if numeric else out


def dot(a, b):
    A = _as_array(a).reshape(-1)
    B = _as_array(b).reshape(-1)
    if A.shape != B.shape:
        raise ValueError("dot: dimension mismatch")
    acc = _ZERO
    for x, y in zip(A, B):
        acc = _s_add(acc, _s_mul(x, y))
    return _wrap(_as_array(acc))


def cross(a, b):
    A = _as_array(a).reshape(-1)
    B = _as_array(b).reshape(-1)
    out = _np.empty((3, 1), dtype=object)
    out[0, 0] = _s_sub(_s_mul(A[1], B[2]), _s_mul(A[2], B[1]))
    out[1, 0] = _s_sub(_s_mul(A[2], B[0]), _s_mul(A[0], B[2]))
    out[2, 0] = _s_sub(_s_mul(A[0], B[1]), _s_mul(A[1], B[0]))
    return _wrap(out)


def transpose(a):
    return _wrap(_as_array(a).T.copy())


def sin(a):
    return _unary_dispatch("sin", a)


def cos(a):
    return _unary_dispatch("cos", a)


def tan(a):
    return _unary_dispatch("tan", a)


def sqrt(a):
    return _unary_dispatch("sqrt", a)


def exp(a):
    return _unary_dispatch("exp", a)


def log(a):
    return _unary_dispatch("log", a)


def fabs(a):
    return _unary_dispatch("fabs", a)


def asin(a):
    return _unary_dispatch("asin", a)


def acos(a):
    return _unary_dispatch("acos", a)


def atan(a):
    return _unary_dispatch("atan", a)


def tanh(a):
    return _unary_dispatch("tanh", a)


def _binary_dispatch(op, a, b):
    """element-wise two-argument function with casadi's broadcasting of a scalar operand"""
    if _is_numeric(a) and _is_numeric(b):
        aa = _np.asarray(a._v if isinstance(a, DM) else a, dtype=float)
        bb = _np.asarray(b._v if isinstance(b, DM) else b, dtype=float)
        res = _np.vectorize(_BINARY_NUMERIC[op], otypes=[float])(aa, bb)
        return DM(res) if (isinstance(a, DM) or isinstance(b, DM)) else (float(res) if res.ndim == 0 else res)
    A, B = _as_array(a), _as_array(b)
    if A.shape != B.shape:
        if A.size == 1:
            A = _np.full(B.shape, A.reshape(-1)[0], dtype=object)
        elif B.size == 1:
            B = _np.full(A.shape, B.reshape(-1)[0], dtype=object)
        else:
            raise ValueError("%s: shapes %s and %s do not match" % (op, A.shape, B.shape))
    out = _np.empty(A.shape, dtype=object)
    for idx in _np.ndindex(A.shape):
        out[idx] = _s_binary(op, A[idx], B[idx])
    return _wrap(out)


def atan2(y, x):
    return _binary_dispatch("atan2", y, x)


def arctan2(y, x):
    return _binary_dispatch("atan2", y, x)


def fmin(a, b):
    return _binary_dispatch("fmin", a, b)


def fmax(a, b):
    return _binary_dispatch("fmax", a, b)


def sign(a):
    return _unary_dispatch("sign", a)


def _unary_dispatch(op, a):
    if _is_numeric(a):
        arr = _np.asarray(a._v if isinstance(a, DM) else a, dtype=float)
        res = _np.vectorize(_UNARY_NUMERIC[op], otypes=[float])(arr)
        return float(res) if res.ndim == 0 else DM(res)
    return _map(lambda s: _s_unary(op, s), a)


def norm_2(a):
    A = _as_array(a)
    if A.shape[0] != 1 and A.shape[1] != 1:
        raise ValueError("norm_2 of a matrix is not supported; use norm_fro")
    out = _wrap(_as_array(_s_norm2(A.reshape(-1))))
    return float(evaluate(out, {})[0, 0]) if _is_numeric(a) else out


def norm_fro(a):
    A = _as_array(a)
    out = _wrap(_as_array(_s_norm2(A.reshape(-1))))
    return float(evaluate(out, {})[0, 0]) if _is_numeric(a) else out


def sumsqr(a):
    A = _as_array(a).reshape(-1)
    acc = _ZERO
    for x in A:
        acc = _s_add(acc, _s_mul(x, x))
    return _wrap(_as_array(acc))


def diag(a):
    A = _as_array(a)
    numeric = _is_numeric(a)
    if A.shape[1] == 1 or A.shape[0] == 1:
        v = A.reshape(-1)
        out = _as_array(_np.zeros((len(v), len(v))))
        for i, s in enumerate(v):
            out[i, i] = s
    else:
        n = min(A.shape)
        out = _np.empty((n, 1), dtype=object)
        for i in range(n):
            out[i, 0] = A[i, i]
    out = _wrap(out)
    return DM(evaluate(out, {})) if numeric else out


def inv(a):
    """Matrix inverse; only constant matrices (the notebooks invert the
    constant desired frame ``cs.inv(T_des)``,
    ur5_transformation_matrix_comparison_of_controllers.ipynb cell 36)."""
    m = MX(a) if not isinstance(a, MX) else a
    if not m.is_constant():
        raise NotImplementedError("inv() of a non-constant expression is not "
                                  "supported by the CasADi-free front-end")
    return DM(_np.linalg.inv(evaluate(m, {})))


def solve(a, b):
    """``cs.solve(A, B)`` = A^-1 B.  Constant operands: numpy.  Expressions: Gaussian elimination without
    pivoting written out on the expression graph - meant for the small symmetric positive definite systems the
    reference forms (``J J' + lam I`` in its pinv, pseudo_inverse.py:92-105; constraints.py:82-85)."""
    A = a if isinstance(a, MX) else MX(a)
    B = b if isinstance(b, MX) else MX(b)
    n = A.size()[0]
    if A.size()[1] != n or B.size()[0] != n:
        raise ValueError("solve: dimension mismatch %s \\ %s" % (A.size(), B.size()))
    if A.is_constant() and B.is_constant():
        return DM(_np.linalg.solve(evaluate(A, {}), evaluate(B, {})))
    k = B.size()[1]
    M = [[A[i, j] for j in range(n)] + [B[i, c] for c in range(k)] for i in range(n)]
    for p in range(n):
        piv = M[p][p]
        for i in range(p + 1, n):
            f = M[i][p] / piv
            for j in range(p + 1, n + k):
                M[i][j] = M[i][j] - f * M[p][j]
    X = [[None] * k for _ in range(n)]
    for c in range(k):
        for i in range(n - 1, -1, -1):
            acc = M[i][n + c]
            for j in range(i + 1, n):
                acc = acc - M[i][j] * X[j][c]
            X[i][c] = acc / M[i][i]
    return vertcat(*[horzcat(*row) for row in X])


def pinv(a):
    """``cs.pinv(A)`` as CasADi defines it: ``size2 >= size1``: ``solve(A A', A)'``, else ``solve(A'A, A')``."""
    A = a if isinstance(a, MX) else MX(a)
    r, c = A.size()
    if c >= r:
        return solve(mtimes(A, A.T), A).T
    return solve(mtimes(A.T, A), A.T)


def if_else(cond, a, b, short_circuit=False):
    c, x = _bcast(_as_array(cond), _as_array(a))
    c, y = _bcast(c, _as_array(b))
    x, y = _bcast(x, y)
    out = _np.empty(x.shape, dtype=object)
    for idx in _np.ndindex(x.shape):
        out[idx] = Scalar("if_else", (c[idx] if c.shape == x.shape else c[0, 0],
                                      x[idx], y[idx]))
    return _wrap(out)


def orientation_error(R, quat_des):
    """Three-vector orientation error  e_o = 1/2 * sum_i r_i(q) x r_i,des
    between the columns of the rotation ``R`` (3x3 expression) and of the
    rotation encoded by the unit quaternion ``quat_des`` = (x, y, z, w).

    The reference has no 3-vector orientation error of its own (its pose
    constraints are Frobenius / dual-quaternion / three-point forms,
    ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 18, 20); this is
    the form SURVEY.md section 8(d) fixes for the 6-D pose task of the benchmark
    configurations.  It is an ordinary expression of q, so its Jacobian is
    d e_o / d q exactly as CasADi AD would produce.
    """
    Ra = _as_array(R)
    qa = _as_array(quat_des).reshape(-1)
    if Ra.shape != (3, 3) or qa.shape != (4,):
        raise ValueError("orientation_error(R 3x3, quat 4)")
    out = _np.empty((3, 1), dtype=object)
    rn = tuple(Ra.reshape(-1))       # row-major 9 entries
    for k in range(3):
        out[k, 0] = Scalar("ori_err", rn + tuple(qa), aux=k)
    return _wrap(out)


def quat_to_rot(quat):
    """Rotation matrix (3x3 MX) of a unit quaternion (x, y, z, w)."""
    x, y, z, w = [MX(_array=_as_array(s)) for s in _as_array(quat).reshape(-1)]
    rows = [
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ]
    return vertcat(*[horzcat(*r) for r in rows])


# --------------------------------------------------------------------------
# forward kinematics atom
# --------------------------------------------------------------------------
def fk_matrix(chain, qvec):
    """4x4 homogeneous transform of ``chain`` (casclik_amd.urdf.Chain) as an
    MX whose entries are opaque 'fk' nodes of the joint variables in ``qvec``
    (one scalar per actuated joint of the chain)."""
    qa = _as_array(qvec).reshape(-1)
    if len(qa) != chain.n_actuated:
        raise ValueError("T_fk expects %d joint values, got %d"
                         % (chain.n_actuated, len(qa)))
    if all(s.is_const() for s in qa):
        return DM(chain.fk_numeric([s.value for s in qa]))
    out = _np.empty((4, 4), dtype=object)
    qargs = tuple(qa)
    for i in range(4):
        for j in range(4):
            if i == 3:
                out[i, j] = _c(1.0 if j == 3 else 0.0)
            else:
                out[i, j] = Scalar("fk", qargs, aux=(chain, i, j))
    return _wrap(out)


# --------------------------------------------------------------------------
# numeric evaluation (user-side convenience, not the hot path)
# --------------------------------------------------------------------------
def _eval_scalar(s, env, memo):
    key = id(s)
    if key in memo:
        return memo[key]
    op = s.op
    if op == "const":
        r = s.value
    elif op == "sym":
        try:
            r = env[id(s.family)][s.index]
        except KeyError:
            raise ValueError("symbol %s has no value" % s.name)
    elif op == "fk":
        chain, i, j = s.aux
        ck = ("fk", id(chain), tuple(id(a) for a in s.args))
        if ck not in memo:
            qv = [_eval_scalar(a, env, memo) for a in s.args]
            memo[ck] = chain.fk_numeric(qv)
        r = memo[ck][i, j]
    elif op == "fk_d":
        chain, i, j, k = s.aux
        ck = ("fk_d", id(chain), tuple(id(a) for a in s.args))
        if ck not in memo:
            qv = [_eval_scalar(a, env, memo) for a in s.args]
            memo[ck] = chain.fk_derivative_numeric(qv)
        r = memo[ck][k][i, j]
    elif op == "ori_err":
        ck = ("ori", tuple(id(a) for a in s.args))
        if ck not in memo:
            vals = [_eval_scalar(a, env, memo) for a in s.args]
            R = _np.array(vals[:9]).reshape(3, 3)
            Rd = _quat_to_rot_numeric(vals[9:13])
            e = _np.zeros(3)
            for i in range(3):
                e += 0.5 * _np.cross(R[:, i], Rd[:, i])
            memo[ck] = e
        r = memo[ck][s.aux]
    else:
        a = [_eval_scalar(x, env, memo) for x in s.args]
        if op == "add":
            r = a[0] + a[1]
        elif op == "sub":
            r = a[0] - a[1]
        elif op == "mul":
            r = a[0] * a[1]
        elif op == "div":
            # (IEEE semantics like CasADi's: x / 0 is inf or nan, not an exception - the reference's multidimensional
            # tangent-cone function divides by a zero norm on the branch an if_else discards)
            r = a[0] / a[1] if a[1] != 0.0 else (float("nan") if a[0] == 0.0 or a[0] != a[0]
                                                  else math.copysign(float("inf"), a[0]) * math.copysign(1.0, a[1]))
        elif op == "neg":
            r = -a[0]
        elif op == "pow":
            r = a[0] ** a[1]
        elif op in _BINARY_NUMERIC:
            r = _BINARY_NUMERIC[op](a[0], a[1])
        elif op == "norm2":
            r = math.sqrt(sum(v * v for v in a))
        elif op == "cmp_lt":
            r = 1.0 if a[0] < a[1] else 0.0
        elif op == "cmp_le":
            r = 1.0 if a[0] <= a[1] else 0.0
        elif op == "cmp_eq":
            r = 1.0 if a[0] == a[1] else 0.0
        elif op == "cmp_ne":
            r = 1.0 if a[0] != a[1] else 0.0
        elif op == "if_else":
            r = a[1] if a[0] != 0.0 else a[2]
        elif op in _UNARY_NUMERIC:
            r = float(_UNARY_NUMERIC[op](a[0]))
        else:
            raise NotImplementedError("evaluate: op %s" % op)
    memo[key] = r
    return r


def _quat_to_rot_numeric(qd):
    x, y, z, w = qd
    return _np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def evaluate(expr, env):
    """Numeric value (ndarray) of ``expr``; ``env`` maps id(SymFamily) ->
    flat list of values."""
    a = _as_array(expr)
    memo = {}
    out = _np.empty(a.shape, dtype=float)
    for idx in _np.ndindex(a.shape):
        out[idx] = _eval_scalar(a[idx], env, memo)
    return out


def _families_of(mx):
    fams = []
    for s in _as_array(mx).flat:
        if s.op != "sym":
            raise ValueError("Function inputs must be purely symbolic")
        if not any(f is s.family for f in fams):
            fams.append(s.family)
    return fams


class Function(object):
    """``casadi.Function`` look-alike for user-side numeric evaluation and
    symbolic re-use (reference usage: ``p_fk = cs.Function("p_fk",[t,q],
    [T_fk(q)[:3,3]])`` then ``p_fk(t,q)`` inside constraint expressions,
    ur5_moe2016_example2.ipynb cell 4)."""

    def __init__(self, name, inputs, outputs, *rest):
        self.name = name
        self._inputs = [i if isinstance(i, MX) else MX(i) for i in inputs]
        self._outputs = [o if isinstance(o, MX) else MX(o) for o in outputs]
        for i in self._inputs:
            _families_of(i)

    def __call__(self, *args):
        if len(args) != len(self._inputs):
            raise TypeError("%s expects %d arguments" % (self.name,
                                                         len(self._inputs)))
        symbolic = any(isinstance(a, MX) and not a.is_constant() for a in args)
        if symbolic:
            outs = [substitute(o, self._inputs, args) for o in self._outputs]
        else:
            env = {}
            for sym_in, val in zip(self._inputs, args):
                flat = _np.asarray(
                    val._v if isinstance(val, DM) else
                    (evaluate(val, {}) if isinstance(val, MX) else val),
                    dtype=float).T.reshape(-1)
                a = _as_array(sym_in)
                if flat.size != a.size:
                    raise ValueError("%s: argument size mismatch" % self.name)
                # column-major flattening matches the sym numbering
                for k, s in enumerate(a.T.reshape(-1)):
                    env.setdefault(id(s.family), {})[s.index] = flat[k]
            outs = [DM(evaluate(o, env)) for o in self._outputs]
        return outs[0] if len(outs) == 1 else tuple(outs)

    def __repr__(self):
        return "Function(%s)" % self.name


def substitute(expr, sym_list, val_list):
    """Replace the symbols in ``sym_list`` by the expressions ``val_list``."""
    table = {}
    for sym_in, val in zip(sym_list, val_list):
        sa = _as_array(sym_in)
        va = _as_array(val)
        if sa.size != va.size:
            raise ValueError("substitute: size mismatch")
        for s, v in zip(sa.T.reshape(-1), va.T.reshape(-1)):
            table[(id(s.family), s.index)] = v
    memo = {}

    def rec(s):
        k = id(s)
        if k in memo:
            return memo[k]
        if s.op == "const":
            r = s
        elif s.op == "sym":
            r = table.get((id(s.family), s.index), s)
        else:
            na = tuple(rec(a) for a in s.args)
            if all(x is y for x, y in zip(na, s.args)):
                r = s
            elif s.op == "add":
                r = _s_add(*na)
            elif s.op == "sub":
                r = _s_sub(*na)
            elif s.op == "mul":
                r = _s_mul(*na)
            elif s.op == "div":
                r = _s_div(*na)
            elif s.op == "neg":
                r = _s_neg(*na)
            elif s.op == "pow":
                r = _s_pow(*na)
            elif s.op in _BINARY_NUMERIC:
                r = _s_binary(s.op, *na)
            elif s.op == "norm2":
                r = _s_norm2(na)
            elif s.op in _UNARY_NUMERIC:
                r = _s_unary(s.op, na[0])
            else:
                r = Scalar(s.op, na, aux=s.aux)
        memo[k] = r
        return r

    a = _as_array(expr)
    out = _np.empty(a.shape, dtype=object)
    for idx in _np.ndindex(a.shape):
        out[idx] = rec(a[idx])
    return _wrap(out)


def depends_on(expr, var):
    """True when any entry of ``expr`` references a symbol of ``var``."""
    fams = [id(f) for f in _families_of(var)]
    seen = set()

    def rec(s):
        if id(s) in seen:
            return False
        seen.add(id(s))
        if s.op == "sym":
            return id(s.family) in fams
        return any(rec(a) for a in s.args)

    return any(rec(s) for s in _as_array(expr).flat)


def jacobian(expr, var):
    """``cs.jacobian`` (reference use: constraints.py:67-73; ur5_moe2016_example2.ipynb cell 4):
    symbolic m x n matrix of partial derivatives."""
    from . import autodiff
    return autodiff.jacobian(expr, var)


def jtimes(expr, var, vec):
    """``cs.jtimes(expr, var, vec)``: the Jacobian of ``expr`` w.r.t. ``var`` times ``vec`` (reference use:
    constraints.py:75-82, pseudo_inverse.py:155-160)."""
    return mtimes(jacobian(expr, var), vec)


def logic_and(a, b):
    """0/1 valued elementwise conjunction (pseudo_inverse.py:230)"""
    return (_wrap_mx(a) != 0.0) * (_wrap_mx(b) != 0.0)


def logic_or(a, b):
    return ((_wrap_mx(a) != 0.0) + (_wrap_mx(b) != 0.0)) != 0.0


def logic_not(a):
    return _wrap_mx(a) == 0.0


def _wrap_mx(a):
    return a if isinstance(a, MX) else MX(a)


# the common base of CasADi's matrix types, for `isinstance(w, cs.GenericMatrixCommon)` (reactive_qp.py:70)
GenericMatrixCommon = (MX, DM)


def trace(a):
    """sum of the diagonal of a square matrix expression"""
    a = _wrap_mx(a)
    rows, cols = a.size()
    if rows != cols:
        raise ValueError("trace of a %d x %d matrix" % (rows, cols))
    total = a[0, 0]
    for i in range(1, rows):
        total = total + a[i, i]
    return total


def skew(v):
    """the cross-product matrix of a 3-vector: skew(a) b = a x b"""
    v = _wrap_mx(v)
    return vertcat(horzcat(0.0, -v[2], v[1]), horzcat(v[2], 0.0, -v[0]), horzcat(-v[1], v[0], 0.0))

