"""Stand-in for the slice of the `casadi` module that the reference casclik package
(/root/reference/casclik, casadi==3.4.1 in its requirements.txt) touches on the
PseudoInverseController / ReactiveQPController path.

PURPOSE (tests/golden/make_ref_golden.py): CasADi cannot be installed here, so no fixture can come
from the reference running on its real back-end.  What CAN be executed is the reference's OWN
Python: its constraint classes, SkillSpecification (priority sort, _has_virtual / _has_input),
`get_problem_expressions` with its per-mode control flow, `get_in_tangent_cone_function[_multidim]`,
`get_cost_expr` / `get_constraints_expr` / `setup_initial_problem_solver` and both `solve`
methods.  This module supplies `import casadi as cs` for that: lazily evaluated dense matrix
expressions (numpy), forward-mode derivatives for `jacobian` / `jtimes`, numpy LU for `solve`,
and a small dense active-set QP for `conic`.  THE ARITHMETIC BACK-END IS THEREFORE A STAND-IN, not
CasADi: fixtures made with it pin the reference's control flow and formulas (which Jacobian is
stacked when, which branch of pinv, bounds, slicing), not CasADi's rounding.  It is test
infrastructure, it shares no code with casclik_amd or oracle/, and nothing in the product imports it.
"""
from __future__ import annotations

import math

import numpy
import numpy as np  # noqa: F401  (the reference reaches numpy through `cs.np`)

pi = math.pi
inf = float("inf")


def _arr(x):
    """numeric things -> 2-D float array (vectors are columns, as in CasADi)"""
    if isinstance(x, DM):
        return x.a
    if isinstance(x, bool):
        return numpy.array([[1.0 if x else 0.0]])
    a = numpy.asarray(x, dtype=float)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a.reshape(-1, 1)
    return a


class GenericMatrixCommon(object):
    __array_ufunc__ = None          # numpy defers to our reflected operators

    def size(self):
        return self.shape

    def size1(self):
        return self.shape[0]

    def size2(self):
        return self.shape[1]

    def numel(self):
        return self.shape[0] * self.shape[1]

    def sparsity(self):
        return ("dense",) + tuple(self.shape)

    # arithmetic (elementwise, scalars broadcast)
    def __add__(self, o): return _binary("add", self, o)
    def __radd__(self, o): return _binary("add", o, self)
    def __sub__(self, o): return _binary("sub", self, o)
    def __rsub__(self, o): return _binary("sub", o, self)
    def __mul__(self, o): return _binary("mul", self, o)
    def __rmul__(self, o): return _binary("mul", o, self)
    def __truediv__(self, o): return _binary("div", self, o)
    def __rtruediv__(self, o): return _binary("div", o, self)
    __div__ = __truediv__
    __rdiv__ = __rtruediv__
    def __neg__(self): return _unary("neg", self)
    def __pow__(self, o): return _binary("pow", self, o)
    def __lt__(self, o): return _binary("lt", self, o)
    def __le__(self, o): return _binary("le", self, o)
    def __gt__(self, o): return _binary("gt", self, o)
    def __ge__(self, o): return _binary("ge", self, o)
    def __eq__(self, o): return _binary("eq", self, o)
    def __ne__(self, o): return _binary("ne", self, o)
    __hash__ = object.__hash__

    @property
    def T(self):
        return _unary("transpose", self)

    def __getitem__(self, idx):
        return _index(self, idx)


def _norm_index(idx, shape):
    """CasADi-style indexing of a dense matrix: x[i] / x[a:b] address the entries of a vector,
    x[r, c] rows and columns; integers keep the dimension (everything stays 2-D)."""
    if not isinstance(idx, tuple):
        if shape[1] == 1:
            idx = (idx, slice(None))
        elif shape[0] == 1:
            idx = (slice(None), idx)
        else:
            raise NotImplementedError("linear indexing of a matrix")
    out = []
    for k, n in zip(idx, shape):
        if isinstance(k, (int, numpy.integer)):
            k = int(k)
            if k < 0:
                k += n
            if not 0 <= k < n:
                raise IndexError("index %d out of range %d" % (k, n))
            out.append(slice(k, k + 1))
        elif isinstance(k, slice):
            out.append(k)
        else:
            out.append(numpy.asarray(k, dtype=int))
    return tuple(out)


class DM(GenericMatrixCommon):
    def __init__(self, x=0.0, m=None):
        if m is not None:
            self.a = numpy.zeros((int(x), int(m)))
        else:
            self.a = numpy.array(_arr(x), dtype=float)

    @property
    def shape(self):
        return self.a.shape

    @staticmethod
    def zeros(n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        return DM(numpy.zeros((int(n), int(m))))

    @staticmethod
    def ones(n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        return DM(numpy.ones((int(n), int(m))))

    @staticmethod
    def eye(n):
        return DM(numpy.eye(int(n)))

    def nnz(self):
        return int(numpy.count_nonzero(self.a))

    def is_symbolic(self):
        return False

    def full(self):
        return self.a.copy()

    toarray = full

    def __setitem__(self, idx, val):
        self.a[_norm_index(idx, self.a.shape)] = _arr(val)

    def __float__(self):
        assert self.a.size == 1
        return float(self.a.reshape(-1)[0])

    def __int__(self):
        return int(float(self))

    def __bool__(self):
        assert self.a.size == 1
        return bool(self.a.reshape(-1)[0] != 0.0)

    __nonzero__ = __bool__

    def __len__(self):
        return self.a.shape[0]

    def __iter__(self):
        for i in range(self.a.shape[0]):
            yield DM(self.a[i:i + 1, :])

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __repr__(self):
        return "DM(%s)" % numpy.array2string(self.a, precision=17)


class MX(GenericMatrixCommon):
    """Lazy dense matrix expression."""

    def __init__(self, x=None, m=None, _op=None, _args=(), _shape=None, _data=None):
        if _op is None:
            a = numpy.zeros((int(x), int(m))) if m is not None else _arr(0.0 if x is None else x)
            _op, _shape, _data = "const", a.shape, a
        self.op, self.args, self.shape, self.data = _op, tuple(_args), tuple(_shape), _data

    @staticmethod
    def sym(name, n=1, m=1):
        return MX(_op="sym", _shape=(int(n), int(m)), _data=name)

    @staticmethod
    def zeros(n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        return MX(numpy.zeros((int(n), int(m))))

    @staticmethod
    def ones(n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        return MX(numpy.ones((int(n), int(m))))

    @staticmethod
    def eye(n):
        return MX(numpy.eye(int(n)))

    def is_symbolic(self):
        return self.op == "sym"

    def is_constant(self):
        return not _symbols_of(self)

    def nnz(self):
        """structural non-zeros, as far as the reference needs them: 0 for the Jacobian of an
        expression that does not depend on the variable (skill_specification.py:150-185,
        reactive_qp.py:346-347) and for constant zeros, dense otherwise"""
        if self.op == "jac":
            expr, var = self.args
            return self.numel() if (id(var) in _symbols_of(expr)) else 0
        if self.op == "const":
            return int(numpy.count_nonzero(self.data))
        return self.numel()

    def __setitem__(self, idx, val):
        raise NotImplementedError("MX item assignment is not on the path this stand-in serves")

    def __repr__(self):
        return "MX(%s %dx%d)" % (self.op, self.shape[0], self.shape[1])


SX = MX


def _lift(x):
    if isinstance(x, MX):
        return x
    return MX(_arr(x))


def _is_sym_expr(*xs):
    return any(isinstance(x, MX) for x in xs)


_ELEMENTWISE = {
    "add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b,
    "pow": lambda a, b: a ** b,
    "atan2": lambda a, b: numpy.arctan2(a, b), "fmin": lambda a, b: numpy.minimum(a, b), "fmax": lambda a, b: numpy.maximum(a, b),
    "lt": lambda a, b: (a < b).astype(float), "le": lambda a, b: (a <= b).astype(float),
    "gt": lambda a, b: (a > b).astype(float), "ge": lambda a, b: (a >= b).astype(float),
    "eq": lambda a, b: (a == b).astype(float), "ne": lambda a, b: (a != b).astype(float),
    "and": lambda a, b: ((a != 0) & (b != 0)).astype(float), "or": lambda a, b: ((a != 0) | (b != 0)).astype(float),
}
_UNARY = {
    "neg": lambda a: -a, "sin": numpy.sin, "cos": numpy.cos, "sqrt": numpy.sqrt, "fabs": numpy.abs,
    "sign": numpy.sign, "transpose": lambda a: a.T, "exp": numpy.exp, "log": numpy.log, "tan": numpy.tan,
    "arccos": numpy.arccos, "not": lambda a: (a == 0).astype(float),
    "arcsin": numpy.arcsin, "arctan": numpy.arctan, "tanh": numpy.tanh,
}


def _bshape(sa, sb):
    if sa == sb:
        return sa
    if sa == (1, 1):
        return sb
    if sb == (1, 1):
        return sa
    raise ValueError("dimension mismatch %s vs %s" % (sa, sb))


def _binary(op, a, b):
    if not _is_sym_expr(a, b):
        aa, bb = _arr(a), _arr(b)
        _bshape(aa.shape, bb.shape)
        return DM(_ELEMENTWISE[op](aa, bb))
    a, b = _lift(a), _lift(b)
    return MX(_op=op, _args=(a, b), _shape=_bshape(a.shape, b.shape))


def _unary(op, a):
    if not isinstance(a, MX):
        return DM(_UNARY[op](_arr(a)))
    shape = (a.shape[1], a.shape[0]) if op == "transpose" else a.shape
    return MX(_op=op, _args=(a,), _shape=shape)


def _index(x, idx):
    if isinstance(x, DM):
        return DM(x.a[_norm_index(idx, x.a.shape)])
    key = _norm_index(idx, x.shape)
    shape = numpy.empty(x.shape)[key].shape
    return MX(_op="index", _args=(x,), _shape=shape, _data=key)


def sin(x): return _unary("sin", x)
def cos(x): return _unary("cos", x)
def tan(x): return _unary("tan", x)
def sqrt(x): return _unary("sqrt", x)
def fabs(x): return _unary("fabs", x)
def sign(x): return _unary("sign", x)
def exp(x): return _unary("exp", x)
def log(x): return _unary("log", x)
def arccos(x): return _unary("arccos", x)
def acos(x): return _unary("arccos", x)
def asin(x): return _unary("arcsin", x)
def arcsin(x): return _unary("arcsin", x)
def atan(x): return _unary("arctan", x)
def arctan(x): return _unary("arctan", x)
def tanh(x): return _unary("tanh", x)
def atan2(y, x): return _binary("atan2", y, x)
def arctan2(y, x): return _binary("atan2", y, x)
def fmin(a, b): return _binary("fmin", a, b)
def fmax(a, b): return _binary("fmax", a, b)
def logic_not(x): return _unary("not", x)
def logic_and(a, b): return _binary("and", a, b)
def logic_or(a, b): return _binary("or", a, b)
def transpose(x): return _unary("transpose", x)


def _cat(op, args, axis):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        # cs.vertcat([1., 1., 1.]): a Python sequence is ONE argument, converted to a column
        return DM(_arr(list(args[0])))
    args = [a for a in args]
    if len(args) == 1 and isinstance(args[0], (MX, DM)):
        return args[0]                              # (CasADi returns the single operand itself)
    if not args:
        return DM(numpy.zeros((0, 1) if axis == 0 else (1, 0)))
    if not _is_sym_expr(*args):
        return DM(numpy.concatenate([_arr(a) for a in args], axis=axis))
    args = [_lift(a) for a in args]
    other = 1 - axis
    if len({a.shape[other] for a in args}) != 1:
        raise ValueError("%s: dimension mismatch %s" % (op, [a.shape for a in args]))
    shape = list(args[0].shape)
    shape[axis] = sum(a.shape[axis] for a in args)
    return MX(_op=op, _args=args, _shape=tuple(shape))


def vertcat(*args): return _cat("vertcat", args, 0)
def horzcat(*args): return _cat("horzcat", args, 1)


def mtimes(a, b=None):
    if b is None and isinstance(a, (list, tuple)):
        out = a[0]
        for x in a[1:]:
            out = mtimes(out, x)
        return out
    sa = _lift(a).shape if isinstance(a, MX) else _arr(a).shape
    sb = _lift(b).shape if isinstance(b, MX) else _arr(b).shape
    if sa == (1, 1) or sb == (1, 1):
        return _binary("mul", a, b)                 # a scalar operand: CasADi's mtimes multiplies elementwise
    if sa[1] != sb[0]:
        raise ValueError("mtimes: %s x %s" % (sa, sb))
    if not _is_sym_expr(a, b):
        return DM(_arr(a) @ _arr(b))
    return MX(_op="mtimes", _args=(_lift(a), _lift(b)), _shape=(sa[0], sb[1]))


def dot(a, b):
    if not _is_sym_expr(a, b):
        return DM(numpy.sum(_arr(a) * _arr(b)))
    a, b = _lift(a), _lift(b)
    if a.shape != b.shape:
        raise ValueError("dot: %s vs %s" % (a.shape, b.shape))
    return MX(_op="dot", _args=(a, b), _shape=(1, 1))


def norm_2(x):
    return sqrt(dot(x, x))


def norm_fro(x):
    return sqrt(dot(x, x))


def trace(x):
    n = x.shape[0]
    out = x[0, 0]
    for i in range(1, n):
        out = out + x[i, i]
    return out


def cross(a, b):
    return vertcat(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def skew(v):
    z = 0.0
    return vertcat(horzcat(z, -v[2], v[1]), horzcat(v[2], z, -v[0]), horzcat(-v[1], v[0], z))


def diag(x):
    if not isinstance(x, MX):
        a = _arr(x)
        return DM(numpy.diag(a.reshape(-1)) if 1 in a.shape else numpy.diag(a).reshape(-1, 1))
    if 1 in x.shape:
        n = x.numel()
        return MX(_op="diag", _args=(x,), _shape=(n, n))
    raise NotImplementedError("diag of a matrix expression")


def if_else(cond, a, b, short_circuit=False):
    if not _is_sym_expr(cond, a, b):
        return DM(numpy.where(_arr(cond) != 0, _arr(a), _arr(b)))
    cond, a, b = _lift(cond), _lift(a), _lift(b)
    shape = _bshape(_bshape(cond.shape, a.shape), b.shape)
    return MX(_op="if_else", _args=(cond, a, b), _shape=shape)


def solve(A, B, *unused):
    """A^-1 B (CasADi: a linear-solver node; here numpy's LU)"""
    if not _is_sym_expr(A, B):
        return DM(numpy.linalg.solve(_arr(A), _arr(B)))
    A, B = _lift(A), _lift(B)
    return MX(_op="solve", _args=(A, B), _shape=(A.shape[1], B.shape[1]))


def pinv(A, *unused):
    """CasADi's GenericMatrix::pinv:  size2 >= size1: solve(A A', A)'  else  solve(A'A, A')"""
    A = A if isinstance(A, MX) else DM(A)
    if A.shape[1] >= A.shape[0]:
        return solve(mtimes(A, A.T), A).T
    return solve(mtimes(A.T, A), A.T)


def jacobian(expr, var):
    if isinstance(var, MX) and var.op == "vertcat" and all(a.op == "sym" and a.shape[1] == 1 for a in var.args):
        # w.r.t. a stack of symbols (state_var = [robot_var; virtual_var], pseudo_inverse.py:79-86)
        return horzcat(*[jacobian(expr, a) for a in var.args])
    if not isinstance(var, MX) or var.op != "sym":
        raise NotImplementedError("jacobian w.r.t. a pure symbol only")
    if not isinstance(expr, MX):
        return DM(numpy.zeros((_arr(expr).size, var.numel())))
    return MX(_op="jac", _args=(expr, var), _shape=(expr.numel(), var.numel()))


def jtimes(expr, var, v, tr=False):
    if tr:
        return mtimes(jacobian(expr, var).T, v)
    return mtimes(jacobian(expr, var), v)


# ---- evaluation ----------------------------------------------------------------------------
_sym_cache = {}


def _symbols_of(node):
    key = id(node)
    got = _sym_cache.get(key)
    if got is not None and got[0] is node:
        return got[1]
    if node.op == "sym":
        out = frozenset([id(node)])
    else:
        out = frozenset()
        for a in node.args:
            if isinstance(a, MX):
                out = out | _symbols_of(a)
    _sym_cache[key] = (node, out)
    return out


def _value(node, env, cache):
    key = id(node)
    if key in cache:
        return cache[key]
    op = node.op
    if op == "const":
        v = node.data
    elif op == "sym":
        if id(node) not in env:
            raise KeyError("free symbol %r in an evaluated expression" % (node.data,))
        v = env[id(node)]
    elif op in _ELEMENTWISE:
        v = _ELEMENTWISE[op](_value(node.args[0], env, cache), _value(node.args[1], env, cache))
    elif op in _UNARY:
        v = _UNARY[op](_value(node.args[0], env, cache))
    elif op == "index":
        v = _value(node.args[0], env, cache)[node.data]
    elif op == "vertcat":
        v = numpy.concatenate([_value(a, env, cache) for a in node.args], axis=0)
    elif op == "horzcat":
        v = numpy.concatenate([_value(a, env, cache) for a in node.args], axis=1)
    elif op == "mtimes":
        v = _value(node.args[0], env, cache) @ _value(node.args[1], env, cache)
    elif op == "dot":
        v = numpy.array([[numpy.sum(_value(node.args[0], env, cache) * _value(node.args[1], env, cache))]])
    elif op == "diag":
        v = numpy.diag(_value(node.args[0], env, cache).reshape(-1))
    elif op == "if_else":
        c = _value(node.args[0], env, cache)
        v = numpy.where(c != 0, _value(node.args[1], env, cache), _value(node.args[2], env, cache))
    elif op == "solve":
        v = numpy.linalg.solve(_value(node.args[0], env, cache), _value(node.args[1], env, cache))
    elif op == "jac":
        expr, var = node.args
        if id(var) not in _symbols_of(expr):
            v = numpy.zeros(node.shape)
        else:
            _, tan = _dual(expr, var, env, {})
            # column-major vec of the expression, one column per seed direction
            v = tan.reshape(expr.shape[0] * expr.shape[1], var.numel(), order="F") if expr.shape[1] != 1 \
                else tan[:, 0, :]
    else:
        raise NotImplementedError(op)
    v = numpy.asarray(v, dtype=float)
    if v.shape != node.shape:
        v = numpy.broadcast_to(v, node.shape).copy()
    cache[key] = v
    return v


def _dual(node, var, env, cache):
    """(value [r,c], tangent [r,c,k]) of `node` for unit seeds on the k entries of `var`"""
    key = id(node)
    if key in cache:
        return cache[key]
    k = var.numel()
    op = node.op
    if id(var) not in _symbols_of(node):
        v = _value(node, env, {})
        out = (v, numpy.zeros(v.shape + (k,)))
        cache[key] = out
        return out
    if op == "sym":
        v = env[id(node)]
        t = numpy.zeros(v.shape + (k,))
        for i in range(k):          # column-major numbering of the entries
            t[i % v.shape[0], i // v.shape[0], i] = 1.0
        out = (v, t)
    elif op in ("add", "sub", "mul", "div"):
        (a, ta), (b, tb) = _dual(node.args[0], var, env, cache), _dual(node.args[1], var, env, cache)
        shape = node.shape
        a, b = numpy.broadcast_to(a, shape), numpy.broadcast_to(b, shape)
        ta, tb = numpy.broadcast_to(ta, shape + (k,)), numpy.broadcast_to(tb, shape + (k,))
        if op == "add":
            out = (a + b, ta + tb)
        elif op == "sub":
            out = (a - b, ta - tb)
        elif op == "mul":
            out = (a * b, ta * b[..., None] + a[..., None] * tb)
        else:
            out = (a / b, (ta - (a / b)[..., None] * tb) / b[..., None])
    elif op == "pow":
        (a, ta) = _dual(node.args[0], var, env, cache)
        b = _value(node.args[1], env, {})
        out = (a ** b, (b * a ** (b - 1))[..., None] * ta)
    elif op == "neg":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (-a, -ta)
    elif op == "sin":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.sin(a), numpy.cos(a)[..., None] * ta)
    elif op == "cos":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.cos(a), -numpy.sin(a)[..., None] * ta)
    elif op == "tan":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.tan(a), (1.0 / numpy.cos(a) ** 2)[..., None] * ta)
    elif op == "exp":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.exp(a), numpy.exp(a)[..., None] * ta)
    elif op == "log":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.log(a), ta / a[..., None])
    elif op == "sqrt":
        a, ta = _dual(node.args[0], var, env, cache)
        s = numpy.sqrt(a)
        out = (s, ta / (2.0 * s)[..., None])
    elif op == "fabs":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.abs(a), numpy.sign(a)[..., None] * ta)
    elif op in ("arccos", "arcsin"):
        a, ta = _dual(node.args[0], var, env, cache)
        sgn = -1.0 if op == "arccos" else 1.0
        out = (_UNARY[op](a), (sgn / numpy.sqrt(1.0 - a * a))[..., None] * ta)
    elif op == "arctan":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.arctan(a), (1.0 / (1.0 + a * a))[..., None] * ta)
    elif op == "tanh":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (numpy.tanh(a), (1.0 - numpy.tanh(a) ** 2)[..., None] * ta)
    elif op in ("atan2", "fmin", "fmax"):
        (a, ta), (b, tb) = _dual(node.args[0], var, env, cache), _dual(node.args[1], var, env, cache)
        shape = node.shape
        a, b = numpy.broadcast_to(a, shape), numpy.broadcast_to(b, shape)
        ta, tb = numpy.broadcast_to(ta, shape + (k,)), numpy.broadcast_to(tb, shape + (k,))
        if op == "atan2":           # d atan2(y, x) = (x dy - y dx) / (x^2 + y^2)
            out = (numpy.arctan2(a, b), (b[..., None] * ta - a[..., None] * tb) / (a * a + b * b)[..., None])
        else:                       # CasADi (casadi_math.hpp): d fmin = [x <= y, !(x <= y)], d fmax = [x >= y, !(x >= y)]
            first = (a <= b) if op == "fmin" else (a >= b)
            out = (numpy.where(first, a, b), numpy.where(first[..., None], ta, tb))
    elif op == "transpose":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (a.T, numpy.transpose(ta, (1, 0, 2)))
    elif op == "index":
        a, ta = _dual(node.args[0], var, env, cache)
        out = (a[node.data], ta[node.data])
    elif op in ("vertcat", "horzcat"):
        parts = [_dual(a, var, env, cache) for a in node.args]
        ax = 0 if op == "vertcat" else 1
        out = (numpy.concatenate([p[0] for p in parts], axis=ax), numpy.concatenate([p[1] for p in parts], axis=ax))
    elif op == "mtimes":
        (a, ta), (b, tb) = _dual(node.args[0], var, env, cache), _dual(node.args[1], var, env, cache)
        out = (a @ b, numpy.einsum("ijk,jl->ilk", ta, b) + numpy.einsum("ij,jlk->ilk", a, tb))
    elif op == "dot":
        (a, ta), (b, tb) = _dual(node.args[0], var, env, cache), _dual(node.args[1], var, env, cache)
        t = numpy.einsum("ijk,ij->k", ta, b) + numpy.einsum("ij,ijk->k", a, tb)
        out = (numpy.array([[numpy.sum(a * b)]]), t.reshape(1, 1, k))
    elif op == "if_else":
        c = _value(node.args[0], env, {})
        (a, ta), (b, tb) = _dual(node.args[1], var, env, cache), _dual(node.args[2], var, env, cache)
        shape = node.shape
        c = numpy.broadcast_to(c, shape)
        out = (numpy.where(c != 0, numpy.broadcast_to(a, shape), numpy.broadcast_to(b, shape)),
               numpy.where((c != 0)[..., None], numpy.broadcast_to(ta, shape + (k,)), numpy.broadcast_to(tb, shape + (k,))))
    else:
        raise NotImplementedError("derivative through %r" % op)
    cache[key] = out
    return out


class Function(object):
    def __init__(self, name, ins, outs, *rest):
        self.name = name
        self.ins = list(ins)
        self.outs = [o if isinstance(o, (MX, DM)) else DM(o) for o in outs]
        for s in self.ins:
            if not (isinstance(s, MX) and s.op == "sym"):
                raise NotImplementedError("Function inputs must be pure symbols")
        free = set()
        for o in self.outs:
            if isinstance(o, MX):
                free |= set(_symbols_of(o)) - {id(s) for s in self.ins}
        if free:
            raise RuntimeError("Function %s has free variables (CasADi raises at construction too)" % name)

    def __call__(self, *args):
        if len(args) != len(self.ins):
            raise TypeError("%s: %d arguments for %d inputs" % (self.name, len(args), len(self.ins)))
        env = {}
        for s, a in zip(self.ins, args):
            v = _arr(a)
            if v.shape != s.shape:
                if v.size == s.numel():
                    v = v.reshape(s.shape, order="F")
                else:
                    raise ValueError("%s: input %s has shape %s, expected %s" % (self.name, s.data, v.shape, s.shape))
            env[id(s)] = v
        cache = {}
        res = [DM(_value(o, env, cache)) if isinstance(o, MX) else DM(o.a) for o in self.outs]
        return res[0] if len(res) == 1 else res


# ---- conic: a small dense convex QP ----------------------------------------------------------
def _solve_qp(H, A, lb, ub, max_iter=200):
    """min 1/2 x'Hx  s.t.  lb <= A x <= ub,  H symmetric positive definite.
    Primal-dual active-set iteration on the KKT system of the working set (dense numpy), started
    from the equality rows; returns (x, multipliers per row (positive at ub, negative at lb))."""
    n = H.shape[0]
    m = A.shape[0]
    eq = [i for i in range(m) if lb[i] == ub[i]]
    work = {i: ub[i] for i in eq}                 # row -> bound value it is held at
    side = {i: 0 for i in eq}                     # 0 equality, +1 upper, -1 lower
    tol = 1e-11
    for _ in range(max_iter):
        rows = sorted(work)
        if rows:
            Aw = A[rows]
            K = numpy.block([[H, Aw.T], [Aw, numpy.zeros((len(rows), len(rows)))]])
            rhs = numpy.concatenate([numpy.zeros(n), numpy.array([work[i] for i in rows])])
            sol = numpy.linalg.lstsq(K, rhs, rcond=None)[0]
            x, nu = sol[:n], sol[n:]
            if numpy.abs(K @ sol - rhs).max() > 1e-9 * (1.0 + numpy.abs(rhs).max()):
                # the working rows contradict each other (least squares of an inconsistent KKT system): this
                # iteration cannot tell an infeasible problem from a bad working set - the fallback can
                return _solve_qp_ldp(H, A, lb, ub)
        else:
            x, nu = numpy.zeros(n), numpy.zeros(0)
        # drop the inequality whose multiplier has the wrong sign the most
        worst, wi = 0.0, None
        for r, i in enumerate(rows):
            if side[i] != 0 and side[i] * nu[r] < -tol * (1 + abs(nu[r])) and -side[i] * nu[r] > worst:
                worst, wi = -side[i] * nu[r], i
        if wi is not None:
            del work[wi], side[wi]
            continue
        ax = A @ x
        viol, vi, vs = 0.0, None, 0
        for i in range(m):
            if i in work:
                continue
            sc = 1e-9 * max(1.0, abs(ub[i]) if numpy.isfinite(ub[i]) else 1.0)
            if ax[i] - ub[i] > max(viol, sc):
                viol, vi, vs = ax[i] - ub[i], i, 1
            sc = 1e-9 * max(1.0, abs(lb[i]) if numpy.isfinite(lb[i]) else 1.0)
            if lb[i] - ax[i] > max(viol, sc):
                viol, vi, vs = lb[i] - ax[i], i, -1
        if vi is None:
            lam = numpy.zeros(m)
            for r, i in enumerate(rows):
                lam[i] = nu[r]
            return x, lam
        work[vi] = ub[vi] if vs > 0 else lb[vi]
        side[vi] = vs
    return _solve_qp_ldp(H, A, lb, ub)


def _nnls(E, f, max_iter=None):
    """Lawson & Hanson's NNLS (Solving Least Squares Problems, ch. 23): argmin ||E u - f||, u >= 0; finite."""
    m, n = E.shape
    max_iter = max_iter or 30 * n
    u = numpy.zeros(n)
    P = numpy.zeros(n, bool)
    w = E.T @ (f - E @ u)
    it = 0
    while (~P).any() and numpy.where(~P, w, -numpy.inf).max() > 1e-12 * (1.0 + numpy.abs(w).max()):
        P[int(numpy.argmax(numpy.where(~P, w, -numpy.inf)))] = True
        while True:
            it += 1
            if it > max_iter:
                raise RuntimeError("stand-in QP: NNLS iteration cap")
            s = numpy.zeros(n)
            s[P] = numpy.linalg.lstsq(E[:, P], f, rcond=None)[0]
            if (s[P] > 0).all():
                break
            neg = P & (s <= 0)
            alpha = numpy.min(u[neg] / (u[neg] - s[neg]))
            u = u + alpha * (s - u)
            P = P & (u > 1e-15)
            u[~P] = 0.0
        u = s
        w = E.T @ (f - E @ u)
    return u


def _solve_qp_ldp(H, A, lb, ub):
    """The same QP as a least-distance problem (Lawson & Hanson ch. 23: min ||z|| s.t. G z >= h through one NNLS),
    used when the working-set iteration above cycles: z = H^(1/2) x, one-sided rows C x <= d of the finite bounds.
    NNLS identifies the active rows (and proves infeasibility: zero residual); the minimiser and multipliers then
    come from the KKT system of those rows, solved densely.  Raises RuntimeError when no point satisfies the rows."""
    n, m = H.shape[0], A.shape[0]
    hd = numpy.diag(H)
    if numpy.abs(H - numpy.diag(hd)).max() > 0:
        raise NotImplementedError("stand-in QP fallback: diagonal H only (what casclik builds)")
    rows, sides = [], []
    for i in range(m):
        if numpy.isfinite(ub[i]):
            rows.append(i); sides.append(+1)
        if numpy.isfinite(lb[i]):
            rows.append(i); sides.append(-1)
    C = numpy.array([sides[k] * A[rows[k]] for k in range(len(rows))])
    d = numpy.array([ub[rows[k]] if sides[k] > 0 else -lb[rows[k]] for k in range(len(rows))])
    G = -C / numpy.sqrt(hd)[None, :]
    h = -d
    E = numpy.vstack([G.T, h[None, :]])
    f = numpy.zeros(n + 1); f[n] = 1.0
    u = _nnls(E, f)
    r = E @ u - f
    if numpy.linalg.norm(r) < 1e-10:
        raise RuntimeError("stand-in QP: infeasible (least-distance residual vanishes)")
    act = [k for k in range(len(rows)) if u[k] > 0]
    # polish on the active rows (an equality given as two opposite rows enters once)
    work, side = {}, {}
    for k in act:
        i = rows[k]
        if i in work:
            continue
        work[i] = ub[i] if sides[k] > 0 else lb[i]
        side[i] = 0 if lb[i] == ub[i] else sides[k]
    wr = sorted(work)
    if wr:
        Aw = A[wr]
        K = numpy.block([[H, Aw.T], [Aw, numpy.zeros((len(wr), len(wr)))]])
        sol = numpy.linalg.lstsq(K, numpy.concatenate([numpy.zeros(n), numpy.array([work[i] for i in wr])]), rcond=None)[0]
        x, nu = sol[:n], sol[n:]
    else:
        x, nu = numpy.zeros(n), numpy.zeros(0)
    lam = numpy.zeros(m)
    for k, i in enumerate(wr):
        lam[i] = nu[k]
    return x, lam


def conic(name, solver, structure, opts=None):
    def run(h=None, a=None, lba=None, uba=None, g=None, x0=None, lbx=None, ubx=None, lam_x0=None, lam_a0=None):
        H, A = _arr(h), _arr(a)
        lb, ub = _arr(lba).reshape(-1), _arr(uba).reshape(-1)
        if g is not None and numpy.any(_arr(g) != 0):
            raise NotImplementedError("linear cost term")
        x, lam = _solve_qp(H, A, lb, ub)
        return {"x": DM(x), "lam_a": DM(lam), "cost": DM(0.5 * x @ H @ x)}
    return run


qpsol = conic
