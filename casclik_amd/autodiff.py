"""Symbolic differentiation over the scalar DAG of casclik_amd.sym.

Serves ``BaseConstraint.jacobian`` (reference: casclik/constraints.py:67-73,
which calls ``cs.jacobian``) for inspection, and gives the lowering the exact
time derivative of time-only sub-expressions (the feed-forward term
``d e/d t`` of reference pseudo_inverse.py:285, reactive_qp.py:215-216).  The
per-tick Jacobians of the hot path are *not* produced here - the HIP kernels
build them from the structured task rows.
"""
from __future__ import annotations

import numpy as np

from . import sym as cs
from .sym import Scalar, _c, _s_add, _s_sub, _s_mul, _s_div, _s_neg, _s_unary, _s_pow, _ZERO

_ZERO = _c(0.0)


def _fk_d(node, k):
    chain, i, j = node.aux
    return Scalar("fk_d", node.args, aux=(chain, i, j, k))


def diff_scalar(node, key, memo):
    """d node / d symbol where ``key`` = (id(family), index)."""
    nid = id(node)
    if nid in memo:
        return memo[nid]
    op = node.op
    if op == "const":
        r = _ZERO
    elif op == "sym":
        r = _c(1.0) if (id(node.family), node.index) == key else _ZERO
    elif op in ("add", "sub"):
        da = diff_scalar(node.args[0], key, memo)
        db = diff_scalar(node.args[1], key, memo)
        r = _s_add(da, db) if op == "add" else _s_sub(da, db)
    elif op == "neg":
        r = _s_neg(diff_scalar(node.args[0], key, memo))
    elif op == "mul":
        a, b = node.args
        r = _s_add(_s_mul(diff_scalar(a, key, memo), b),
                   _s_mul(a, diff_scalar(b, key, memo)))
    elif op == "div":
        a, b = node.args
        da = diff_scalar(a, key, memo)
        db = diff_scalar(b, key, memo)
        r = _s_sub(_s_div(da, b), _s_mul(_s_div(a, _s_mul(b, b)), db))
    elif op == "pow":
        a, b = node.args
        da = diff_scalar(a, key, memo)
        db = diff_scalar(b, key, memo)
        r = _s_mul(_s_mul(b, _s_pow(a, _s_sub(b, _c(1.0)))), da)
        if not (db.is_const() and db.value == 0.0):
            r = _s_add(r, _s_mul(_s_mul(node, _s_unary("log", a)), db))
    elif op == "sin":
        r = _s_mul(_s_unary("cos", node.args[0]),
                   diff_scalar(node.args[0], key, memo))
    elif op == "cos":
        r = _s_neg(_s_mul(_s_unary("sin", node.args[0]),
                          diff_scalar(node.args[0], key, memo)))
    elif op == "tan":
        c = _s_unary("cos", node.args[0])
        r = _s_div(diff_scalar(node.args[0], key, memo), _s_mul(c, c))
    elif op == "sqrt":
        r = _s_div(diff_scalar(node.args[0], key, memo),
                   _s_mul(_c(2.0), node))
    elif op == "exp":
        r = _s_mul(node, diff_scalar(node.args[0], key, memo))
    elif op == "log":
        r = _s_div(diff_scalar(node.args[0], key, memo), node.args[0])
    elif op == "fabs":
        r = _s_mul(_s_unary("sign", node.args[0]),
                   diff_scalar(node.args[0], key, memo))
    elif op in ("asin", "acos"):
        a = node.args[0]
        r = _s_div(diff_scalar(a, key, memo), _s_unary("sqrt", _s_sub(_c(1.0), _s_mul(a, a))))
        if op == "acos":
            r = _s_neg(r)
    elif op == "atan":
        a = node.args[0]
        r = _s_div(diff_scalar(a, key, memo), _s_add(_c(1.0), _s_mul(a, a)))
    elif op == "tanh":
        r = _s_mul(_s_sub(_c(1.0), _s_mul(node, node)), diff_scalar(node.args[0], key, memo))
    elif op == "atan2":
        y, x = node.args
        dy, dx = diff_scalar(y, key, memo), diff_scalar(x, key, memo)
        r = _s_div(_s_sub(_s_mul(x, dy), _s_mul(y, dx)), _s_add(_s_mul(x, x), _s_mul(y, y)))
    elif op in ("fmin", "fmax"):
        # CasADi's rule: the first argument's derivative where it is the one taken (ties included)
        a, b = node.args
        first = Scalar("cmp_le", (a, b) if op == "fmin" else (b, a))
        da, db = diff_scalar(a, key, memo), diff_scalar(b, key, memo)
        r = _ZERO if (da.is_const() and db.is_const() and da.value == 0.0 and db.value == 0.0) else Scalar("if_else", (first, da, db))
    elif op == "sign" or op.startswith("cmp_"):
        r = _ZERO
    elif op == "norm2":
        acc = _ZERO
        for a in node.args:
            acc = _s_add(acc, _s_mul(a, diff_scalar(a, key, memo)))
        r = _ZERO if (acc.is_const() and acc.value == 0.0) else _s_div(acc, node)
    elif op == "if_else":
        c, a, b = node.args
        r = Scalar("if_else", (c, diff_scalar(a, key, memo),
                               diff_scalar(b, key, memo)))
    elif op == "fk":
        acc = _ZERO
        for k, a in enumerate(node.args):
            da = diff_scalar(a, key, memo)
            if not (da.is_const() and da.value == 0.0):
                acc = _s_add(acc, _s_mul(_fk_d(node, k), da))
        r = acc
    else:
        # opaque nodes (orientation error, FK derivative entries): zero when no argument depends on the
        # symbol (e.g. d/dt of a pose error whose target is not a function of time)
        ds = [diff_scalar(a, key, memo) for a in node.args]
        if all(d.is_const() and d.value == 0.0 for d in ds):
            r = _ZERO
        else:
            raise NotImplementedError("symbolic derivative of op '%s'" % op)
    memo[nid] = r
    return r


def jacobian(expr, var):
    """MX (m x n) of partial derivatives of the column ``expr`` w.r.t. the
    symbols of the column ``var``."""
    ea = cs._as_array(expr)
    va = cs._as_array(var)
    if ea.shape[1] != 1:
        raise ValueError("jacobian expects a column expression")
    syms = list(va.T.reshape(-1))
    out = np.empty((ea.shape[0], len(syms)), dtype=object)
    for j, s in enumerate(syms):
        if s.op != "sym":
            raise ValueError("jacobian: differentiation variable must be symbolic")
        memo = {}
        key = (id(s.family), s.index)
        for i in range(ea.shape[0]):
            out[i, j] = diff_scalar(ea[i, 0], key, memo)
    return cs.MX(_array=out)
