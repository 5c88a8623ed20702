"""Device code for constraint expressions outside the affine-in-features family.

The reference differentiates every constraint expression with CasADi
(``cs.jacobian``, casclik/constraints.py:67-73) and JIT-compiles the resulting
function per controller (casclik/controllers/pseudo_inverse.py:476-483,
reactive_qp.py:262-298).  Most skills lower to the device task table
(lowering.py) and run hand-written row code; a constraint that does not - products
or trigonometric functions of the state (``l*cos(q[0]+q[1])`` of
double_pendulum_2D_comparison_of_controllers.ipynb cell 4), functions of the tool
frame other than its entries and norms - is kept as an expression graph.  This
module turns such a constraint into one straight-line ``__device__`` function

    ExternTask<TI>::eval(z, ys, tv, K, e, J, Jt)

(value, state Jacobian and partial time derivative; clik_pinv_static.hpp) from the
scalar DAG and its symbolic derivatives (autodiff.py), with common sub-expressions
shared.  Time-only sub-expressions stay on the host as time slots (``tv``), like in
the row table.  The kernel templates around it are the hand-written ones; jit.py
compiles both for the skill.
"""
from __future__ import annotations

import math

from . import autodiff

_UNARY_C = {"tan": "tan", "sqrt": "sqrt", "exp": "exp", "log": "log",
            "fabs": "fabs", "asin": "asin", "acos": "acos", "atan": "atan", "tanh": "tanh"}
_BINARY_C = {"atan2": "atan2", "fmin": "fmin", "fmax": "fmax"}
_INFIX = {"add": "+", "sub": "-", "mul": "*", "div": "/"}


def _lit(v):
    v = float(v)
    if math.isinf(v):
        return "(-__builtin_inf())" if v < 0 else "__builtin_inf()"
    if math.isnan(v):
        return "__builtin_nan(\"\")"
    r = repr(v)
    return "(%s)" % r if v < 0 else r


class TaskEmitter(object):
    """Emits the body of one ExternTask<TI>::eval.  ``low`` is the skill's
    lowering._Lowerer (symbol families, chain bookkeeping, time slots)."""

    def __init__(self, low):
        self.low = low
        self.lines = []
        self._by_id = {}
        self._by_key = {}
        self._tslot_by_repr = {}
        self.uses_fk = False
        self._keep = []          # derivative trees must outlive the id()-keyed memo

    # -- leaves -------------------------------------------------------------
    def _sym(self, node):
        low = self.low
        fam = node.family
        if fam is low.fam_q:
            return "z[%d]" % node.index
        if fam is low.fam_x and fam is not None:
            return "z[%d]" % (low.desc.n_q + node.index)
        if fam is low.fam_y and fam is not None:
            return "ys[%d]" % node.index
        if fam is low.fam_dq or fam is low.fam_dx:
            raise NotImplementedError("constraint expressions must not contain velocity variables")
        raise NotImplementedError("symbol '%s' is not a variable of the skill specification" % node.name)

    def _time_slot(self, node):
        key = repr(node)
        if key not in self._tslot_by_repr:
            self._tslot_by_repr[key] = self.low._tslot(node)
        return "tv[%d]" % self._tslot_by_repr[key]

    def _temp(self, key, expr):
        if key in self._by_key:
            return self._by_key[key]
        name = "v%d" % len(self.lines)
        self.lines.append("const double %s = %s;" % (name, expr))
        self._by_key[key] = name
        return name

    # -- nodes --------------------------------------------------------------
    def ref(self, node):
        nid = id(node)
        if nid not in self._by_id:
            self._by_id[nid] = self._ref(node)
        return self._by_id[nid]

    def _ref(self, node):
        op = node.op
        low = self.low
        if op == "const":
            return _lit(node.value)
        if low._time_only(node):
            return self._time_slot(node)
        if op == "sym":
            return self._sym(node)
        if op == "fk":
            low._use_chain(node)
            self.uses_fk = True
            _, i, j = node.aux
            return "K.p[%d]" % i if j == 3 else "K.R[%d]" % (3 * i + j)
        if op == "fk_d":
            low._use_chain(node)
            self.uses_fk = True
            _, i, j, k = node.aux
            s = low._chain_state_index[k]
            if j == 3:
                return "K.Jv[%d][%d]" % (i, s)
            a, b = (i + 1) % 3, (i + 2) % 3
            # d R[:, j] / d z_s = w_s x R[:, j]
            expr = "K.Jw[%d][%d] * K.R[%d] - K.Jw[%d][%d] * K.R[%d]" % (a, s, 3 * b + j, b, s, 3 * a + j)
            return self._temp(("fk_d", i, j, s), expr)
        if op == "ori_err":
            raise NotImplementedError("orientation_error inside a non-affine constraint expression")
        args = [self.ref(a) for a in node.args]
        if op in ("sin", "cos"):
            # one evaluation serves both (a Jacobian needs the other one anyway); sincos_joint is the
            # kernels' own routine (straight-line up to 1e5 rad, library fallback beyond)
            key = ("sincos", args[0])
            if key not in self._by_key:
                k = len(self.lines)
                self.lines.append("double s%d, c%d; sincos_joint(%s, s%d, c%d);" % (k, k, args[0], k, k))
                self._by_key[key] = ("s%d" % k, "c%d" % k)
            return self._by_key[key][0 if op == "sin" else 1]
        if op in _INFIX:
            expr = "%s %s %s" % (args[0], _INFIX[op], args[1])
        elif op == "neg":
            expr = "-%s" % args[0]
        elif op in _UNARY_C:
            expr = "%s(%s)" % (_UNARY_C[op], args[0])
        elif op in _BINARY_C:
            expr = "%s(%s, %s)" % (_BINARY_C[op], args[0], args[1])
        elif op == "sign":
            expr = "(double)((%s > 0.0) - (%s < 0.0))" % (args[0], args[0])
        elif op == "pow":
            b = node.args[1]
            if b.is_const() and b.value == 2.0:
                expr = "%s * %s" % (args[0], args[0])
            elif b.is_const() and b.value == 1.0:
                return args[0]
            elif b.is_const() and b.value == 0.5:
                expr = "sqrt(%s)" % args[0]
            elif b.is_const() and float(b.value).is_integer() and abs(b.value) <= 64:
                expr = "__builtin_powi(%s, %d)" % (args[0], int(b.value))
            else:
                expr = "pow(%s, %s)" % (args[0], args[1])
        elif op == "norm2":
            expr = "sqrt(%s)" % " + ".join("%s * %s" % (a, a) for a in args)
        elif op == "cmp_lt":
            expr = "(%s < %s ? 1.0 : 0.0)" % (args[0], args[1])
        elif op == "cmp_le":
            expr = "(%s <= %s ? 1.0 : 0.0)" % (args[0], args[1])
        elif op == "cmp_eq":
            expr = "(%s == %s ? 1.0 : 0.0)" % (args[0], args[1])
        elif op == "cmp_ne":
            expr = "(%s != %s ? 1.0 : 0.0)" % (args[0], args[1])
        elif op == "if_else":
            expr = "(%s != 0.0 ? %s : %s)" % (args[0], args[1], args[2])
        else:
            raise NotImplementedError("no device code for operation '%s'" % op)
        return self._temp((op,) + tuple(args), expr)

    # -- attributes given as expressions ---------------------------------------
    def emit_attr(self, ti, nodes):
        """C++ source of ``template <> struct ExternAttr<ti>``: ``nodes`` are the Scalar nodes of the task's
        attribute slice in its layout order (clik_pinv_static.hpp: [gain | set_min | set_max | target], present
        parts only).  Values only - the reference never differentiates a gain or a bound
        (casclik/constraints.py:67-73 takes the Jacobian of ``expression``)."""
        n = self.low.desc.n_state
        out = ["a[%d] = %s;" % (k, self.ref(node)) for k, node in enumerate(nodes)]
        body = "\n        ".join(self.lines + out)
        return ("template <>\n"
                "struct ExternAttr<%d> {\n"
                "    template <int N>\n"
                "    __device__ __forceinline__ static void eval(const double (&z)[N], const double* ys, const double* tv,\n"
                "                                                const Kin<N>& K, double* a)\n"
                "    {\n"
                "        static_assert(N == %d, \"generated for another skill structure\");\n"
                "        (void)z; (void)ys; (void)tv; (void)K;\n"
                "        %s\n"
                "    }\n"
                "};\n" % (ti, n, body))

    # -- one constraint -------------------------------------------------------
    def emit_task(self, ti, nodes):
        """C++ source of ``template <> struct ExternTask<ti>`` for the column
        of Scalar nodes ``nodes``."""
        low = self.low
        n = low.desc.n_state
        keys = []
        for j in range(n):
            if j < low.desc.n_q:
                keys.append((id(low.fam_q), j))
            else:
                keys.append((id(low.fam_x), j - low.desc.n_q))
        tkey = (id(low.fam_t), 0) if low.fam_t is not None else None
        out = []
        memos = [dict() for _ in range(n + 1)]
        for i, node in enumerate(nodes):
            out.append("e[%d] = %s;" % (i, self.ref(node)))
            for j in range(n):
                d = autodiff.diff_scalar(node, keys[j], memos[j])
                self._keep.append(d)
                out.append("J[%d][%d] = %s;" % (i, j, self.ref(d)))
            if tkey is None:
                out.append("Jt[%d] = 0.0;" % i)
            else:
                d = autodiff.diff_scalar(node, tkey, memos[n])
                self._keep.append(d)
                out.append("Jt[%d] = %s;" % (i, self.ref(d)))
        body = "\n        ".join(self.lines + out)
        return ("template <>\n"
                "struct ExternTask<%d> {\n"
                "    template <int N, int M>\n"
                "    __device__ __forceinline__ static void eval(const double (&z)[N], const double* ys, const double* tv,\n"
                "                                                const Kin<N>& K, double (&e)[M], double (&J)[M][N],\n"
                "                                                double (&Jt)[M])\n"
                "    {\n"
                "        static_assert(N == %d && M == %d, \"generated for another skill structure\");\n"
                "        (void)ys; (void)tv; (void)K;\n"
                "        %s\n"
                "    }\n"
                "};\n" % (ti, n, len(nodes), body))
