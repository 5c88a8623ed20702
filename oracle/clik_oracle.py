"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by casclik_amd.

Line-by-line fp64 numpy restatement of the reference's per-tick hot path:

  * PseudoInverseController   /root/reference/casclik/controllers/pseudo_inverse.py
      pinv                       :92-105
      create_activation_map      :107-130
      in-tangent-cone (1-D)      :132-190
      in-tangent-cone (multidim) :192-257
      get_problem_expressions    :259-451   (incl. the first-equality double
                                             processing :317-326 + :382-396)
      solve (mode scan)          :512-556
  * ReactiveQPController      /root/reference/casclik/controllers/reactive_qp.py
      get_cost_expr              :175-189
      get_constraints_expr       :191-246
      solve / result slicing     :461-528

PARITY - WHAT IT IS PINNED TO.  Outputs of the reference itself: the closed-loop figures its notebooks store of
their real CasADi + qpOASES runs (some forty figures, every curve retraced within a pixel = 0.3 - 2 % of the plotted
range; tests/golden/make_figure_pins.py, tests/test_figure_pins.py), the print_constraints() texts and the UR5
forward-kinematics values the notebooks store.  WHAT THOSE FIGURES RESOLVE (profiles/r5_figure_resolution.md, made by
tools/figure_resolution.py; asserted by tests/test_figure_pins.py::test_what_the_*_resolve): a controller that
deviates from the reference's algorithm in ONE of the following misses a stored figure by the pixels given (the literal
algorithm: 0.16 - 0.56):
    first equality processed once instead of twice (quirk D1, :317-326 + :382-396)        319.5 px  dqc Q_dist2 / pinv
    damping factor 1e-5 or 1e-3 instead of 1e-7 (:47, :92-105); 1e-9                        324 px; 2.6 px  same figure
    pinv_method "standard" there                                                            no answer (J J' singular)
    pinv(J N) instead of N pinv(J) (:387-394)                                               31.6 px   Moe, three walls
    J instead of S J for an active multidimensional set (:289-298, :401-404)                30.5 px   Moe, multidim
    mode scan with the most active sets first (:107-130)                                    106 px; 25.6 px (cart)
    no feed-forward term (:320-321)                                                         68.2 px   Moe
    QP weight shifter 1e-2 instead of 1e-3 (reactive_qp.py:44)                              169 px    dqc quat_dist / qp
NOT resolved by any stored output, because no notebook exercises the difference (these rest on the line-by-line
reading above and on the fixtures made by the reference's own Python over the stand-in casadi, nothing else):
    the 1e-12 margins of the tangent cones and which side the boundary belongs to (D4, D5): measure-zero events
    the multidimensional cone's corner rule against the row-wise 1-D rule (:222-252): the stored runs never leave
        the box through a corner
    an active set that would push back instead of freezing its rows (D2): < 0.3 px on every stored run
    converge_final_set_to_max (D3), a leading VelocityEqualityConstraint, pinv_method "standard" away from
        singularities: 0 uses in the notebooks
    the slack weight mu + w against (1 + mu) w (D12): identical bit for bit at w = 1, the only weight ever used
Outputs of the reference's own CODE run in the build container over a stand-in casadi (tests/golden/make_ref_golden.py ->
ref_pins.npz, 1e-9; the stand-in itself must retrace the same figures before any fixture is written).  NOT pinned:
CasADi's rounding - no fixture comes from CasADi arithmetic at 1e-9 ("parity unpinned at the stated fp64 tolerance"):
the reference is pure Python on top of casadi==3.4.1 (requirements.txt:1) and urdf2casadi (un-vendored, unpinned);
neither exists in the build container nor on the GPU box, and the reference ships no tests or golden outputs for
``solve()``.  Besides the figures: the UR5 forward-kinematics KAT stored in the notebooks
(||p_tool0|| = 1.0192 at home), the constraint-order printouts, algebraic invariants of the damped pseudo-inverse,
KKT optimality of every QP answer, and agreement of the AD Jacobians below with finite differences.

CasADi's algorithmic differentiation (``cs.jacobian``, constraints.py:67-73) is
restated as forward-mode dual numbers over the same expression trees the
product lowers to its device tables, so the oracle does NOT share the
product's lowering or its closed-form geometric Jacobians.  ``cs.solve`` is
restated as ``numpy.linalg.solve`` (LU with partial pivoting); CasADi's own
linear solver differs in rounding, which is why parity is stated with a
tolerance (tests/tolerances.py).

Everything is vectorised over the instance batch where that is natural (the
expression evaluation); the controller algebra loops over instances.
"""
from __future__ import annotations

import math
import numpy as np


# ==========================================================================
# forward-mode AD over casclik_amd.sym scalar trees
# ==========================================================================
class Dual(object):
    """value [B] and derivative [B, nd] w.r.t. (t, z_0..z_{n-1})."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v = v
        self.d = d

    @staticmethod
    def const(val, B, nd):
        return Dual(np.full(B, float(val)), np.zeros((B, nd)))

    def __add__(self, o):
        return Dual(self.v + o.v, self.d + o.d)

    def __sub__(self, o):
        return Dual(self.v - o.v, self.d - o.d)

    def __mul__(self, o):
        return Dual(self.v * o.v, self.d * o.v[:, None] + o.d * self.v[:, None])

    def __truediv__(self, o):
        q = self.v / o.v
        return Dual(q, (self.d - o.d * q[:, None]) / o.v[:, None])

    def __neg__(self):
        return Dual(-self.v, -self.d)

    def scale(self, k):
        return Dual(self.v * k, self.d * k)


def _d_unary(x, f, df):
    return Dual(f(x.v), x.d * df(x.v)[:, None])


def _rpy_like_rot(axis, ang):
    """Rodrigues rotation with Dual angle -> 3x3 list of Duals."""
    x, y, z = axis
    c = _d_unary(ang, np.cos, lambda v: -np.sin(v))
    s = _d_unary(ang, np.sin, np.cos)
    B, nd = ang.d.shape
    one = Dual.const(1.0, B, nd)
    C = one - c
    def k(val):
        return Dual.const(val, B, nd)
    return [[c + C.scale(x * x), C.scale(x * y) - s.scale(z), C.scale(x * z) + s.scale(y)],
            [C.scale(y * x) + s.scale(z), c + C.scale(y * y), C.scale(y * z) - s.scale(x)],
            [C.scale(z * x) - s.scale(y), C.scale(z * y) + s.scale(x), c + C.scale(z * z)]]


def _fk_dual(chain, qd, B, nd):
    """Chain product  T = prod_j Trans(p_j) R_j Rot(axis_j, q_j)  with Dual
    joint values (URDF convention).  Returns 3x3 R and
    3-vector p as lists of Duals."""
    R = [[Dual.const(1.0 if i == j else 0.0, B, nd) for j in range(3)] for i in range(3)]
    p = [Dual.const(0.0, B, nd) for _ in range(3)]
    for jt in chain.joints:
        Rf = np.asarray(jt.R, dtype=float)
        pf = np.asarray(jt.p, dtype=float)
        # p <- p + R * pf ;  R <- R * Rf
        p = [p[i] + R[i][0].scale(pf[0]) + R[i][1].scale(pf[1]) + R[i][2].scale(pf[2])
             for i in range(3)]
        R = [[R[i][0].scale(Rf[0, j]) + R[i][1].scale(Rf[1, j]) + R[i][2].scale(Rf[2, j])
              for j in range(3)] for i in range(3)]
        if jt.type == 1:      # revolute
            M = _rpy_like_rot(jt.axis, qd[jt.q_index])
            R = [[R[i][0] * M[0][j] + R[i][1] * M[1][j] + R[i][2] * M[2][j]
                  for j in range(3)] for i in range(3)]
        elif jt.type == 2:    # prismatic
            ax = jt.axis
            d = qd[jt.q_index]
            p = [p[i] + (R[i][0].scale(ax[0]) + R[i][1].scale(ax[1]) + R[i][2].scale(ax[2])) * d
                 for i in range(3)]
    return R, p


def _quat_rot_dual(q4, B, nd):
    x, y, z, w = q4
    one = Dual.const(1.0, B, nd)
    two = 2.0
    return [[one - (y * y + z * z).scale(two), (x * y - z * w).scale(two), (x * z + y * w).scale(two)],
            [(x * y + z * w).scale(two), one - (x * x + z * z).scale(two), (y * z - x * w).scale(two)],
            [(x * z - y * w).scale(two), (y * z + x * w).scale(two), one - (x * x + y * y).scale(two)]]


class ExprEvaluator(object):
    """Evaluates expression trees for a batch with derivatives w.r.t.
    (time, state).  ``env`` maps id(SymFamily) -> ('t'|'z'|'y', offset)."""

    def __init__(self, spec, t, Z, Y):
        self.B = Z.shape[0]
        self.n = Z.shape[1]
        self.nd = 1 + self.n
        B, nd = self.B, self.nd
        self.memo = {}
        self.bind = {}
        tt = np.broadcast_to(np.asarray(t, dtype=float).reshape(-1), (B,)) \
            if np.ndim(t) else np.full(B, float(t))
        def fam(var):
            if var is None or var._a.size == 0:
                return None
            return next(s.family for s in var._a.flat)
        f_t, f_q, f_x, f_y = fam(spec.time_var), fam(spec.robot_var), \
            fam(spec.virtual_var), fam(spec.input_var)
        if f_t is not None:
            d = np.zeros((B, nd)); d[:, 0] = 1.0
            self.bind[id(f_t)] = [Dual(tt.copy(), d)]
        zs = []
        for k in range(self.n):
            d = np.zeros((B, nd)); d[:, 1 + k] = 1.0
            zs.append(Dual(Z[:, k].astype(float).copy(), d))
        nq = spec.n_robot_var
        self.bind[id(f_q)] = zs[:nq]
        if f_x is not None:
            self.bind[id(f_x)] = zs[nq:]
        if f_y is not None:
            if Y is None:
                raise ValueError("skill has input_var but no input values given")
            self.bind[id(f_y)] = [Dual(Y[:, k].astype(float).copy(), np.zeros((B, nd)))
                                  for k in range(Y.shape[1])]

    def ev(self, s):
        k = id(s)
        if k in self.memo:
            return self.memo[k]
        B, nd = self.B, self.nd
        op = s.op
        if op == "const":
            r = Dual.const(s.value, B, nd)
        elif op == "sym":
            try:
                r = self.bind[id(s.family)][s.index]
            except KeyError:
                raise ValueError("unbound symbol %s" % s.name)
        elif op == "fk":
            chain, i, j = s.aux
            ck = ("fk", id(chain), tuple(id(a) for a in s.args))
            if ck not in self.memo:
                qd = [self.ev(a) for a in s.args]
                self.memo[ck] = _fk_dual(chain, qd, B, nd)
            R, p = self.memo[ck]
            r = p[i] if j == 3 else R[i][j]
        elif op == "ori_err":
            ck = ("ori", tuple(id(a) for a in s.args))
            if ck not in self.memo:
                a = [self.ev(x) for x in s.args]
                R = [a[0:3], a[3:6], a[6:9]]
                Rd = _quat_rot_dual(a[9:13], B, nd)
                e = [Dual.const(0.0, B, nd) for _ in range(3)]
                for c in range(3):          # e += 1/2 * r_c x rd_c  (columns)
                    rc = [R[0][c], R[1][c], R[2][c]]
                    dc = [Rd[0][c], Rd[1][c], Rd[2][c]]
                    e[0] = e[0] + (rc[1] * dc[2] - rc[2] * dc[1]).scale(0.5)
                    e[1] = e[1] + (rc[2] * dc[0] - rc[0] * dc[2]).scale(0.5)
                    e[2] = e[2] + (rc[0] * dc[1] - rc[1] * dc[0]).scale(0.5)
                self.memo[ck] = e
            r = self.memo[ck][s.aux]
        else:
            a = [self.ev(x) for x in s.args]
            if op == "add":
                r = a[0] + a[1]
            elif op == "sub":
                r = a[0] - a[1]
            elif op == "mul":
                r = a[0] * a[1]
            elif op == "div":
                r = a[0] / a[1]
            elif op == "neg":
                r = -a[0]
            elif op == "sin":
                r = _d_unary(a[0], np.sin, np.cos)
            elif op == "cos":
                r = _d_unary(a[0], np.cos, lambda v: -np.sin(v))
            elif op == "tan":
                r = _d_unary(a[0], np.tan, lambda v: 1.0 / np.cos(v) ** 2)
            elif op == "sqrt":
                r = _d_unary(a[0], np.sqrt, lambda v: 0.5 / np.sqrt(v))
            elif op == "exp":
                r = _d_unary(a[0], np.exp, np.exp)
            elif op == "log":
                r = _d_unary(a[0], np.log, lambda v: 1.0 / v)
            elif op == "fabs":
                r = _d_unary(a[0], np.abs, np.sign)
            elif op == "sign":
                r = Dual(np.sign(a[0].v), np.zeros((B, nd)))
            elif op == "asin":
                r = _d_unary(a[0], np.arcsin, lambda v: 1.0 / np.sqrt(1.0 - v * v))
            elif op == "acos":
                r = _d_unary(a[0], np.arccos, lambda v: -1.0 / np.sqrt(1.0 - v * v))
            elif op == "atan":
                r = _d_unary(a[0], np.arctan, lambda v: 1.0 / (1.0 + v * v))
            elif op == "tanh":
                r = _d_unary(a[0], np.tanh, lambda v: 1.0 - np.tanh(v) ** 2)
            elif op == "atan2":
                y, x = a
                den = x.v * x.v + y.v * y.v
                r = Dual(np.arctan2(y.v, x.v), (y.d * x.v[:, None] - x.d * y.v[:, None]) / den[:, None])
            elif op in ("fmin", "fmax"):
                # CasADi: d fmin / d(first) = (first <= second), d fmax / d(first) = (first >= second)
                first = (a[0].v <= a[1].v) if op == "fmin" else (a[0].v >= a[1].v)
                r = Dual(np.where(first, a[0].v, a[1].v), np.where(first[:, None], a[0].d, a[1].d))
            elif op == "pow":
                if np.any(a[1].d != 0.0):
                    raise NotImplementedError("oracle: pow with variable exponent")
                ex = a[1].v
                r = Dual(a[0].v ** ex, a[0].d * (ex * a[0].v ** (ex - 1.0))[:, None])
            elif op == "norm2":
                ss = np.zeros(B)
                dd = np.zeros((B, nd))
                for x in a:
                    ss += x.v * x.v
                    dd += x.d * x.v[:, None]
                nrm = np.sqrt(ss)
                with np.errstate(divide="ignore", invalid="ignore"):
                    r = Dual(nrm, dd / nrm[:, None])
            else:
                raise NotImplementedError("oracle: op '%s'" % op)
        self.memo[k] = r
        return r

    def vector(self, mx):
        """(e [B,m], Jt [B,m], Jz [B,m,n]) of a column expression."""
        nodes = [mx._a[i, 0] for i in range(mx._a.shape[0])]
        ds = [self.ev(s) for s in nodes]
        e = np.stack([d.v for d in ds], axis=1)
        D = np.stack([d.d for d in ds], axis=1)
        return e, D[:, :, 0], D[:, :, 1:]


# ==========================================================================
# helpers
# ==========================================================================
class _Attrs(object):
    """The attributes of one constraint as numbers for one instance: the reference multiplies / subtracts
    ``cnstr.gain``, ``set_min``, ``set_max``, ``target`` inside its symbolic expressions, so an MX attribute that
    depends on (t, q, x, y) is simply evaluated with the rest (constraints.py:35-39, :90-92;
    pseudo_inverse.py:301-318; reactive_qp.py:199-232)."""
    __slots__ = ("gain", "set_min", "set_max", "target")


def _is_symbolic(val):
    return hasattr(val, "_a") and hasattr(val, "is_constant") and not val.is_constant()


def _attr_values(evaluator, val):
    """[B, rows, cols] values of an MX attribute"""
    a = val._a
    out = np.empty((evaluator.B,) + a.shape)
    for i in range(a.shape[0]):
        for j in range(a.shape[1]):
            out[:, i, j] = evaluator.ev(a[i, j]).v
    return out


def attribute_views(evaluator, constraints):
    """views[ci][b]: the constraint's attributes as the numbers instance b sees"""
    views = []
    for c in constraints:
        per = {}
        for name in _Attrs.__slots__:
            val = getattr(c, name, None)
            per[name] = _attr_values(evaluator, val) if _is_symbolic(val) else None
        row = []
        for b in range(evaluator.B):
            v = _Attrs()
            for name in _Attrs.__slots__:
                if per[name] is None:
                    setattr(v, name, getattr(c, name, None))
                else:
                    x = per[name][b]
                    setattr(v, name, x if (name == "gain" and x.shape[0] == x.shape[1] and x.shape[0] > 1)
                            else x.reshape(-1))
            row.append(v)
        views.append(row)
    return views


def _cls(cnstr):
    # (the Python class, as the reference's isinstance tests see it - `constraint_class` is "BaseConstraint" on the two
    # velocity constraints, constraints.py:299-368)
    for klass in type(cnstr).__mro__:
        if klass.__name__ in ("EqualityConstraint", "SetConstraint", "VelocityEqualityConstraint", "VelocitySetConstraint"):
            return klass.__name__
    return type(cnstr).__name__


def _num(val, m):
    """float / list / ndarray / DM / constant MX -> ndarray (m,)"""
    if hasattr(val, "toarray"):
        val = val.toarray()
    arr = np.asarray(val, dtype=float).reshape(-1)
    if arr.size == 1 and m > 1:
        arr = np.full(m, arr[0])
    return arr


def _gain_apply(gain, vec):
    """cs.mtimes(gain, vec) for float or square-matrix gains."""
    if hasattr(gain, "toarray"):
        gain = gain.toarray()
    g = np.asarray(gain, dtype=float)
    if g.ndim == 0 or g.size == 1:
        return float(g.reshape(-1)[0]) * vec
    return g.dot(vec)


def default_pinv_options(opt=None):
    """Defaults of pseudo_inverse.py:42-66."""
    opt = dict(opt or {})
    opt.setdefault("feedforward", True)
    opt.setdefault("multidim_sets", False)
    opt.setdefault("converge_final_set_to_max", False)
    opt.setdefault("pinv_method", "damped")
    opt.setdefault("damping_factor", 1e-7)
    return opt


def activation_map(n_sets):
    """pseudo_inverse.py:107-130: bit patterns of 0..2^n-1 with set 0 as the
    least significant bit, stably sorted by the number of active sets."""
    if n_sets == 0:
        return []
    binmaps = []
    for mode_idx in range(2 ** n_sets):
        bits = [0] * n_sets
        for idx, ch in enumerate(reversed(bin(mode_idx)[2:])):
            bits[-(idx + 1)] = int(ch)
        binmaps.append(list(reversed(bits)))
    return sorted(binmaps, key=lambda s: sum(s))


def dpinv(J, opt, cond=None):
    """pseudo_inverse.py:92-105.  `cond` (a one-element list, optional) accumulates the 2-norm condition numbers of the
    symmetric matrices handed to the linear solver: what the result is sensitive to (tests/tolerances.py; the rounding
    errors of the solves of one mode add up)."""
    rows, cols = J.shape
    if opt["pinv_method"] == "standard":
        # cs.pinv (CasADi GenericMatrix::pinv): size2 >= size1 -> solve(J J^T, J)^T, else solve(J^T J, J^T);
        # a square J takes the first form
        inner = J.dot(J.T) if cols >= rows else J.T.dot(J)
        if cond is not None:
            cond[0] += _sym_cond(inner)
        if cols >= rows:
            return np.linalg.solve(inner, J).T
        return np.linalg.solve(inner, J.T)
    lam = opt["damping_factor"]
    if cols >= rows:
        inner = J.dot(J.T) + lam * np.eye(rows)
        if cond is not None:
            cond[0] += _sym_cond(inner)
        return np.linalg.solve(inner, J).T
    inner = J.T.dot(J) + lam * np.eye(cols)
    if cond is not None:
        cond[0] += _sym_cond(inner)
    return np.linalg.solve(inner, J.T)


_LAST_CONDITION = {}


def _remember_condition(result, kappa):
    """the condition numbers of the most recent solve, keyed by the result array it returned (tests/tolerances.py looks
    them up for the array a test compares against - no stale numbers for another array)"""
    _LAST_CONDITION.clear()
    _LAST_CONDITION["id"] = id(result)
    _LAST_CONDITION["ref"] = result          # (keeps the id from being reused)
    _LAST_CONDITION["kappa"] = kappa


def condition_of(result):
    """condition numbers [B] of the solve that returned `result` (the array itself or a basic slice of it), or None"""
    if not _LAST_CONDITION:
        return None
    base = result
    while base is not None and id(base) != _LAST_CONDITION["id"]:
        base = getattr(base, "base", None)
    if base is None or len(result) != len(_LAST_CONDITION["kappa"]):
        return None
    return _LAST_CONDITION["kappa"]


def _sym_cond(A):
    w = np.linalg.eigvalsh(A)
    lo, hi = abs(w[0]), abs(w[-1])
    return float(hi / lo) if lo > 0.0 else float("inf")


def in_tangent_cone_1d(e, set_min, set_max, dexpr, eps=1e-12):
    """pseudo_inverse.py:162-185 (boundary counts as inside, 1e-12 margin; `eps` is that margin - only the tests that
    state what the stored figures resolve pass another)."""
    if set_min - e < eps:
        if e - set_max < eps:
            return True
        return bool(dexpr < 0.0)
    return bool(dexpr > 0.0)


def in_tangent_cone_multidim(e, set_min, set_max, dexpr, eps=1e-12, loose_inside=False):
    """pseudo_inverse.py:222-252 (`eps`, `loose_inside`: deliberate deviations for the resolution tests only -
    `loose_inside` is the 1-D function's inside test `min - e < eps`, quirk D5)."""
    le = e - set_min
    ue = e - set_max
    if loose_inside:
        inside = bool(np.all(-le < eps) and np.all(ue < eps))
    else:
        inside = bool(np.all(le >= eps) and np.all(ue <= eps))
    if inside:
        return True
    out_dir = (np.sign(le) + np.sign(ue)) / 2.0
    corner = bool(np.all(np.sign(le) == np.sign(ue)))
    od = float(out_dir.dot(dexpr))
    if corner:
        if od < 0.0:
            dists = (np.linalg.norm(dexpr) + 1e-10) * np.linalg.norm(out_dir)
            return bool(abs(-od) / dists < math.cos(math.pi / 4))
        return False
    return bool(od < 0.0)


def tangent_cone_margin(e, set_min, set_max, dexpr):
    """Distance of one tangent-cone decision (either function above) from flipping, as the smallest of
    the quantities it thresholds: |e - bound -/+ 1e-12| of every row (the inside test), and - outside -
    the normalised |out_dir . de| (or the distance of the corner ratio from cos(pi/4)).  A diagnostic for
    the parity tests: two correct fp64 evaluations of de differ by ~1e-9 relative, so a decision with a
    margin far above that cannot depend on which of them made it."""
    e = np.atleast_1d(np.asarray(e, dtype=float))
    set_min = np.atleast_1d(np.asarray(set_min, dtype=float))
    set_max = np.atleast_1d(np.asarray(set_max, dtype=float))
    dexpr = np.atleast_1d(np.asarray(dexpr, dtype=float))
    le, ue = e - set_min, e - set_max
    if e.shape[0] == 1:
        inside = (set_min[0] - e[0] < 1e-12) and (e[0] - set_max[0] < 1e-12)
        margin = min(abs(set_min[0] - e[0] - 1e-12), abs(e[0] - set_max[0] - 1e-12))
        return float(margin if inside else min(margin, abs(dexpr[0])))
    inside = bool(np.all(le >= 1e-12) and np.all(ue <= 1e-12))
    margin = float(min(np.abs(le - 1e-12).min(), np.abs(ue - 1e-12).min()))
    if inside:
        return margin
    out_dir = (np.sign(le) + np.sign(ue)) / 2.0
    od = float(out_dir.dot(dexpr))
    scale = (np.linalg.norm(dexpr) + 1e-10) * max(np.linalg.norm(out_dir), 1e-300)
    margin = min(margin, float(np.abs(le).min()), float(np.abs(ue).min()), abs(od) / scale)
    if bool(np.all(np.sign(le) == np.sign(ue))) and od < 0.0:
        margin = min(margin, abs(abs(od) / scale - math.cos(math.pi / 4)))
    return margin


# ==========================================================================
# PseudoInverseController
# ==========================================================================
def pinv_solve_batch(spec, options, t, Q, X=None, Y=None, return_all_modes=False, margins_out=None, _wrong=None,
                     cond_out=None):
    """Literal PseudoInverseController: returns (dZ [B,n_state], mode [B]).

    dZ[:, :n_q] is robot_vel, the rest virtual_vel.

    `cond_out` [B] (optional) receives, per instance, the SUM of the condition numbers of the matrices the reference's
    algorithm hands to its linear solver in the accepted mode (pseudo_inverse.py:92-105; the errors of successive solves add
    up): the yardstick of the stated parity tolerance (tests/tolerances.py).

    `_wrong` (None = the reference's algorithm) names ONE deliberate deviation, for the tests that state what the
    reference-held figure pins resolve (tests/test_figure_pins.py): "no_S" stacks J instead of S J for an active
    multidimensional set (:352-355, 401-404), "no_D1" processes the first equality once (the textbook reading of
    :317-396), "textbook_projection" uses pinv(J N) instead of N pinv(J) (:387-394), "active_first" scans the modes
    with the most active sets first (:107-130), "cone_1d_rows" tests a multidimensional set row by row with the 1-D
    rule (:162-185 in place of :222-252), "boundary_flipped" moves the 1e-12 margins of both cone functions to the
    other side of the bounds (the class docstring's reading, quirk D4), "multidim_loose" gives the multidimensional
    cone the 1-D function's inside test (quirk D5), "set_pushes_back" lets an active set drive its violated rows
    back to the bound with its gain instead of only freezing them (quirk D2)."""
    assert _wrong in (None, "no_S", "no_D1", "textbook_projection", "active_first", "cone_1d_rows", "boundary_flipped",
                      "multidim_loose", "set_pushes_back"), _wrong
    cone_eps = -1e-12 if _wrong == "boundary_flipped" else 1e-12
    opt = default_pinv_options(options)
    Q = np.atleast_2d(np.asarray(Q, dtype=float))
    B = Q.shape[0]
    Z = Q if X is None else np.hstack([Q, np.atleast_2d(np.asarray(X, dtype=float))])
    n = Z.shape[1]
    Yb = None if Y is None else np.atleast_2d(np.asarray(Y, dtype=float))
    evaluator = ExprEvaluator(spec, t, Z, Yb)
    cn = spec.constraints
    data = []
    for c in cn:
        e, Jt, Jz = evaluator.vector(c.expression)
        data.append((e, Jt, Jz))
    views = attribute_views(evaluator, cn)
    n_sets = sum(1 for c in cn if _cls(c) == "SetConstraint")
    amap = activation_map(n_sets)
    if _wrong == "active_first":
        amap = amap[::-1]
    n_modes = 2 ** n_sets
    ff = opt["feedforward"]
    multidim = opt["multidim_sets"]
    conv_last = opt["converge_final_set_to_max"]
    if not multidim:
        for c in cn:
            if _cls(c) == "SetConstraint" and c.expression.size()[0] > 1:
                raise NotImplementedError("multidimensional SetConstraint "
                                          "without multidim_sets")
    dZ = np.zeros((B, n))
    modes = -np.ones(B, dtype=np.int32)
    allv = np.zeros((B, n_modes, n)) if return_all_modes else None
    I = np.eye(n)
    cond_all = np.ones(B)
    for b in range(B):
        for mode_idx in range(n_modes):
            cnd = [1.0]                 # (per mode: only the accepted mode's solves shape the answer)
            set_idx = 0
            v = np.zeros(n)
            Ja, rJa, tc = [], [], []
            for ci, c in enumerate(cn):
                e = data[ci][0][b]
                Jt = data[ci][1][b]
                Ji = data[ci][2][b]
                m = e.shape[0]
                kind = _cls(c)
                a = views[ci][b]          # gain / bounds / target as numbers (the constraint's own unless symbolic)
                is_first = len(Ja) == 0
                is_last = ci == len(cn) - 1
                is_set = kind == "SetConstraint"
                is_eq = kind == "EqualityConstraint"
                is_veleq = kind == "VelocityEqualityConstraint"
                if multidim and is_set:
                    smin = _num(a.set_min, m)
                    smax = _num(a.set_max, m)
                    S = np.diag(((e - smax > 0.0) | (e - smin < 0.0)).astype(float))
                    if _wrong == "no_S":
                        S = np.eye(m)
                just_processed = False
                # chain 1 (:317-326)
                if is_first and is_eq:
                    just_processed = _wrong == "no_D1"
                    des = -_gain_apply(a.gain, e)
                    if ff:
                        des = des - Jt
                    v = v + dpinv(Ji, opt, cnd).dot(des)
                    Ja.append(Ji); rJa.append(Ji)
                # chain 2 (:327-443) - an independent if/elif ladder
                if is_first and is_veleq:
                    des = _num(a.target, m).copy()
                    if ff:
                        des = des - Jt
                    v = v + dpinv(Ji, opt, cnd).dot(des)
                    Ja.append(Ji); rJa.append(Ji)
                elif is_set and is_last and conv_last:
                    if amap[mode_idx][set_idx]:
                        des = _gain_apply(a.gain, _num(a.set_max, m) - e)
                        if ff:
                            des = des - Jt
                        if Ja:
                            N = I - dpinv(np.vstack(Ja), opt, cnd).dot(np.vstack(rJa))
                        else:
                            N = I   # never exercised by the reference (vertcat of nothing)
                        v = v + N.dot(dpinv(Ji, opt, cnd)).dot(des)
                        Ja.append(Ji)
                        rJa.append(S.dot(Ji) if multidim else Ji)
                    else:
                        tc.append(ci)
                    set_idx += 1
                elif is_eq and just_processed:
                    pass
                elif is_eq:
                    des = -_gain_apply(a.gain, e)
                    if ff:
                        des = des - Jt
                    N = I - dpinv(np.vstack(Ja), opt, cnd).dot(np.vstack(rJa))
                    if _wrong == "textbook_projection":
                        v = v + dpinv(Ji.dot(N), opt, cnd).dot(des - Ji.dot(v))
                    else:
                        v = v + N.dot(dpinv(Ji, opt, cnd)).dot(des)
                    Ja.append(Ji); rJa.append(Ji)
                elif is_set:
                    if amap[mode_idx][set_idx]:
                        if _wrong == "set_pushes_back":
                            smin_, smax_ = _num(a.set_min, m), _num(a.set_max, m)
                            des = _gain_apply(a.gain, np.clip(e, smin_, smax_) - e)
                            N = I - dpinv(np.vstack(Ja), opt, cnd).dot(np.vstack(rJa)) if Ja else I
                            v = v + N.dot(dpinv(Ji, opt, cnd)).dot(des)
                        Ja.append(Ji)
                        rJa.append(S.dot(Ji) if multidim else Ji)
                    else:
                        tc.append(ci)
                    set_idx += 1
                elif is_veleq:
                    des = _num(a.target, m).copy()
                    if ff:
                        des = des - Jt
                    N = I - dpinv(np.vstack(Ja), opt, cnd).dot(np.vstack(rJa))
                    v = v + N.dot(dpinv(Ji, opt, cnd)).dot(des)
                    Ja.append(Ji); rJa.append(Ji)
                # VelocitySetConstraint: no branch -> ignored
            if return_all_modes:
                allv[b, mode_idx] = v
            # solve(): accept the first mode whose inactive sets are all in
            # their tangent cone (:530-550)
            ok = True
            for ci in tc:
                a = views[ci][b]
                e = data[ci][0][b]
                m = e.shape[0]
                dexpr = data[ci][1][b] + data[ci][2][b].dot(v)
                smin = _num(a.set_min, m)
                smax = _num(a.set_max, m)
                if m == 1:
                    good = in_tangent_cone_1d(e[0], smin[0], smax[0], dexpr[0], cone_eps)
                elif _wrong == "cone_1d_rows":
                    good = all(in_tangent_cone_1d(e[k], smin[k], smax[k], dexpr[k]) for k in range(m))
                else:
                    good = in_tangent_cone_multidim(e, smin, smax, dexpr, cone_eps, _wrong == "multidim_loose")
                if margins_out is not None and modes[b] < 0:
                    # (every decision of the scan up to and including the accepted mode)
                    margins_out[b] = min(margins_out[b], tangent_cone_margin(e, smin, smax, dexpr))
                if not good:
                    ok = False
                    break
            if ok and modes[b] < 0:
                modes[b] = mode_idx
                dZ[b] = v
                cond_all[b] = cnd[0]
                if not return_all_modes:
                    break
        if modes[b] < 0:
            cond_all[b] = cnd[0]
    if cond_out is not None:
        cond_out[:] = cond_all
    _remember_condition(dZ, cond_all)
    if return_all_modes:
        return dZ, modes, allv
    return dZ, modes


# ==========================================================================
# ReactiveQPController
# ==========================================================================
def qp_weights(spec, robot_var_weights=None, virtual_var_weights=None,
               slack_var_weights=None):
    """Weight defaults of reactive_qp.py:65-133."""
    wr = np.ones(spec.n_robot_var) if robot_var_weights is None \
        else np.asarray(robot_var_weights, dtype=float).reshape(-1)
    wv = np.ones(spec.n_virtual_var) if virtual_var_weights is None \
        else np.asarray(virtual_var_weights, dtype=float).reshape(-1)
    if slack_var_weights is None:
        ws = []
        for c in spec.constraints:
            if c.constraint_type == "soft":
                ws += [float(c.slack_weight)] * c.expression.size()[0]
        ws = np.asarray(ws, dtype=float)
    else:
        ws = np.asarray(slack_var_weights, dtype=float).reshape(-1)
    return wr, wv, ws


def qp_data_batch(spec, t, Q, X=None, Y=None, weights=None, mu=0.001, _wrong=None):
    """H (diag), A, lbA, ubA per instance (reactive_qp.py:175-246).

    Returns Hdiag [B,nv], A [B,nc,nv], lb [B,nc], ub [B,nc].
    `_wrong` (resolution tests only): "slack_times" writes the slack weights as (1 + mu) w - the INITIAL problem's form
    (:331) - instead of mu + w (:187; quirk D12)."""
    assert _wrong in (None, "slack_times"), _wrong
    Q = np.atleast_2d(np.asarray(Q, dtype=float))
    B = Q.shape[0]
    Z = Q if X is None else np.hstack([Q, np.atleast_2d(np.asarray(X, dtype=float))])
    Yb = None if Y is None else np.atleast_2d(np.asarray(Y, dtype=float))
    evaluator = ExprEvaluator(spec, t, Z, Yb)
    wr, wv, ws = weights if weights is not None else qp_weights(spec)
    nq = spec.n_robot_var
    nvirt = spec.n_virtual_var
    nslack = spec.n_slack_var
    hd = [mu * wr]
    if nvirt > 0:
        hd.append(mu * wv)
    if nslack > 0:
        hd.append((1.0 + mu) * ws if _wrong == "slack_times" else mu + ws)
    hd = np.concatenate(hd)
    nv = hd.size
    A_blocks, lb_blocks, ub_blocks = [], [], []
    slack_ind = 0
    views = attribute_views(evaluator, spec.constraints)
    for ci, c in enumerate(spec.constraints):
        e, Jt, Jz = evaluator.vector(c.expression)
        m = e.shape[1]
        blk = np.zeros((B, m, nv))
        blk[:, :, :Jz.shape[2]] = Jz
        lb = -Jt.copy()
        ub = -Jt.copy()
        kind = _cls(c)
        av = views[ci]            # gain / bounds / target as numbers per instance
        if kind == "EqualityConstraint":
            ke = np.stack([_gain_apply(av[b].gain, e[b]) for b in range(B)])
            lb -= ke
            ub -= ke
        elif kind == "SetConstraint":
            lb += np.stack([_gain_apply(av[b].gain, _num(av[b].set_min, m) - e[b]) for b in range(B)])
            ub += np.stack([_gain_apply(av[b].gain, _num(av[b].set_max, m) - e[b]) for b in range(B)])
        elif kind == "VelocityEqualityConstraint":
            tg = np.stack([_num(av[b].target, m) for b in range(B)])
            lb += tg
            ub += tg
        elif kind == "VelocitySetConstraint":
            lb += np.stack([_num(av[b].set_min, m) for b in range(B)])
            ub += np.stack([_num(av[b].set_max, m) for b in range(B)])
        if nslack > 0 and c.constraint_type == "soft":
            for i in range(m):
                blk[:, i, nq + nvirt + slack_ind + i] = -1.0
            slack_ind += m
        A_blocks.append(blk)
        lb_blocks.append(lb)
        ub_blocks.append(ub)
    A = np.concatenate(A_blocks, axis=1)
    lbA = np.concatenate(lb_blocks, axis=1)
    ubA = np.concatenate(ub_blocks, axis=1)
    return np.broadcast_to(hd, (B, nv)).copy(), A, lbA, ubA


class QPInfeasible(Exception):
    pass


def qp_solve_dense(hdiag, A, lb, ub, max_iter=200, eq_tol=0.0):
    """Exact solution of   min 1/2 x'Hx  s.t. lb <= A x <= ub,  H = diag(hdiag) > 0
    (the problem ``cs.conic`` hands to qpOASES, reactive_qp.py:491-513: no
    linear term, no variable bounds) by the dual active-set method of
    Goldfarb & Idnani (1983), written with explicit dense solves - clarity over
    speed.  H is strictly positive so the minimiser is unique and
    solver-independent; ``kkt_residuals`` proves optimality of the answer."""
    hd = np.asarray(hdiag, dtype=float)
    A = np.asarray(A, dtype=float)
    nc, nv = A.shape
    Ginv = 1.0 / hd
    # inequality list: (row, sign) meaning sign*a_row.x >= sign*bound
    cons = []
    eq_rows = []
    for i in range(nc):
        if ub[i] - lb[i] <= eq_tol:
            eq_rows.append(i)
        else:
            # (an infinite bound is no constraint: notebooks pass set_max=cs.inf,
            # double_pendulum_2D_comparison_of_controllers.ipynb cell 10)
            if np.isfinite(lb[i]):
                cons.append((i, +1.0, lb[i]))
            if np.isfinite(ub[i]):
                cons.append((i, -1.0, -ub[i]))
    x = np.zeros(nv)
    act = []     # entries (normal vector, rhs, is_eq, tag)
    u = np.zeros(0)

    def normal(i, sgn):
        return sgn * A[i]

    def add_constraint(nvec, rhs, is_eq, tag):
        nonlocal x, u, act
        up = np.append(u, 0.0)
        for _ in range(4 * (nc + 2)):
            s = nvec.dot(x) - rhs
            if act:
                N = np.stack([a[0] for a in act], axis=1)          # nv x q
                GN = Ginv[:, None] * N
                M = N.T.dot(GN)
                rvec = np.linalg.solve(M, GN.T.dot(nvec))
                z = Ginv * (nvec - N.dot(rvec))
            else:
                rvec = np.zeros(0)
                z = Ginv * nvec
            zn = z.dot(nvec)
            # partial step: largest step keeping the active inequality
            # multipliers non-negative
            t1, drop = math.inf, -1
            for j, a in enumerate(act):
                if not a[2] and rvec[j] > 1e-14:
                    cand = up[j] / rvec[j]
                    if cand < t1:
                        t1, drop = cand, j
            scale = max(1.0, float(np.abs(nvec).max()) ** 2 * float(Ginv.max()))
            if zn > 1e-13 * scale:
                t2 = -s / zn
            else:
                t2 = math.inf
            if is_eq and s > 0:
                # equality approached from the other side: flip the normal
                nvec, rhs = -nvec, -rhs
                continue
            tstep = min(t1, t2)
            if tstep == math.inf:
                raise QPInfeasible("constraint %r cannot be satisfied" % (tag,))
            if t2 == math.inf:
                up[:-1] -= tstep * rvec
                up[-1] += tstep
                act.pop(drop)
                up = np.delete(up, drop)
                continue
            x = x + tstep * z
            up[:-1] -= tstep * rvec
            up[-1] += tstep
            if tstep == t2:
                act.append((nvec, rhs, is_eq, tag))
                u = up
                return
            act.pop(drop)
            up = np.delete(up, drop)
        raise QPInfeasible("active-set inner loop did not terminate")

    for i in eq_rows:
        rhs = 0.5 * (lb[i] + ub[i])
        nvec = A[i].copy()
        s = nvec.dot(x) - rhs
        if s > 0:
            nvec, rhs = -nvec, -rhs
        if abs(s) > 0 or True:
            add_constraint(nvec, rhs, True, (i, 0))
    for it in range(max_iter):
        worst, pick = -1e-11, None
        for (i, sgn, rhs) in cons:
            if any((a[3] == (i, sgn)) for a in act):
                continue
            viol = sgn * A[i].dot(x) - rhs
            nrm = max(1.0, abs(rhs))
            if viol / nrm < worst:
                worst, pick = viol / nrm, (i, sgn, rhs)
        if pick is None:
            # never hand out a point that is not primal feasible: on an
            # infeasible problem the scan above can run out of candidates
            Ax = A.dot(x)
            lo_s = np.where(np.isfinite(lb), (lb - Ax) / np.maximum(1.0, np.abs(np.where(np.isfinite(lb), lb, 0.0))), -1.0)
            hi_s = np.where(np.isfinite(ub), (Ax - ub) / np.maximum(1.0, np.abs(np.where(np.isfinite(ub), ub, 0.0))), -1.0)
            if np.any(lo_s > 1e-8) or np.any(hi_s > 1e-8):
                raise QPInfeasible("no feasible point found")
            return x
        i, sgn, rhs = pick
        add_constraint(normal(i, sgn), rhs, False, (i, sgn))
    raise QPInfeasible("iteration cap reached")


def kkt_residuals(hdiag, A, lb, ub, x, act_tol=1e-8):
    """(primal infeasibility, stationarity residual, worst multiplier sign
    violation) of ``x`` for the QP above.  Multipliers are recovered by least
    squares on the rows active at ``x``."""
    hd = np.asarray(hdiag, dtype=float)
    Ax = A.dot(x)
    scale = np.maximum(1.0, np.maximum(np.abs(lb), np.abs(ub)))
    prim = max(0.0, float(np.max((lb - Ax) / scale)), float(np.max((Ax - ub) / scale)))
    at_lb = np.abs(Ax - lb) <= act_tol * scale
    at_ub = np.abs(Ax - ub) <= act_tol * scale
    idx = np.where(at_lb | at_ub)[0]
    g = hd * x
    if idx.size == 0:
        return prim, float(np.abs(g).max()), 0.0
    N = A[idx].T
    lam, *_ = np.linalg.lstsq(N, g, rcond=None)
    stat = float(np.abs(N.dot(lam) - g).max())
    sign_bad = 0.0
    for k, i in enumerate(idx):
        if at_lb[i] and at_ub[i]:
            continue                      # equality: free sign
        if at_lb[i]:
            sign_bad = max(sign_bad, -lam[k])   # needs lam >= 0
        else:
            sign_bad = max(sign_bad, lam[k])    # needs lam <= 0
    return prim, stat, float(sign_bad)


def qp_condition(hdiag, A, lb, ub, x, act_tol=1e-8):
    """Sensitivity of the QP's minimiser x to rounding: cond(H) times the condition number of the Schur complement
    S = Aa H^-1 Aa' of the rows active at x (x = H^-1 Aa' S^+ b_a), taken over the singular values that are not zero to
    rounding (dependent active rows do not move x).  The yardstick of the stated QP tolerance (tests/tolerances.py)."""
    hd = np.asarray(hdiag, dtype=float)
    kh = float(hd.max() / hd.min())
    Ax = A.dot(x)
    scale = np.maximum(1.0, np.maximum(np.abs(lb), np.abs(ub)))
    idx = np.where((np.abs(Ax - lb) <= act_tol * scale) | (np.abs(Ax - ub) <= act_tol * scale))[0]
    if idx.size == 0:
        return kh
    Aa = A[idx]
    sv = np.linalg.svd((Aa / hd[None, :]).dot(Aa.T), compute_uv=False)
    sv = sv[sv > 1e-12 * sv[0]]
    return kh * float(sv[0] / sv[-1])


def qp_solve_batch(spec, t, Q, X=None, Y=None, weights=None, mu=0.001, cond_out=None, _wrong=None):
    """Literal ReactiveQPController.solve: (dq [B,nq], dx [B,nx] | None,
    slack [B,ns] | None, status [B]).  `cond_out` [B]: qp_condition of every solved instance."""
    hd, A, lbA, ubA = qp_data_batch(spec, t, Q, X, Y, weights, mu, _wrong)
    B, nv = hd.shape
    xs = np.zeros((B, nv))
    status = np.zeros(B, dtype=np.int32)
    kappa = np.ones(B)
    for b in range(B):
        try:
            xs[b] = qp_solve_dense(hd[b], A[b], lbA[b], ubA[b])
            kappa[b] = qp_condition(hd[b], A[b], lbA[b], ubA[b], xs[b])
        except QPInfeasible:
            status[b] = 2
            xs[b] = np.nan
    if cond_out is not None:
        cond_out[:] = kappa
    _remember_condition(xs, kappa)
    nq = spec.n_robot_var
    nvirt = spec.n_virtual_var if spec.virtual_var is not None else 0
    ns = spec.n_slack_var
    dq = xs[:, :nq]
    dx = xs[:, nq:nq + nvirt] if nvirt > 0 else None
    slack = xs[:, nq + nvirt:nq + nvirt + ns] if ns > 0 else None
    return dq, dx, slack, status


def qp_initial_problem(spec, t0, q0, x0=None, dq0=None, y0=None, weights=None, mu=0.001):
    """Literal ReactiveQPController.solve_initial_problem (reactive_qp.py:300-459): the QP over
    (virtual_vel, slack) with the robot velocity fixed to ``dq0`` (zeros when not given, :437-438):

        H = diag(mu w_virt, (1 + mu) w_slack)                                   (:321-331)
        rows of every constraint that depends on virtual_var or is soft         (:339-391):
            [J_virt | -I_slack] [dx; s]  in  [lb, ub] - d e/d t - J_q dq0       (:353-372)

    Returns (virtual_vel | None, slack | None); (None, None) without virtual and slack variables."""
    from casclik_amd import sym as cs
    nq = spec.n_robot_var
    nvirt = spec.n_virtual_var if spec.virtual_var is not None else 0
    nslack = spec.n_slack_var
    if nvirt == 0 and nslack == 0:
        return None, None
    q0 = np.asarray(q0, dtype=float).reshape(1, -1)
    Z = q0 if nvirt == 0 else np.hstack([q0, (np.zeros((1, nvirt)) if x0 is None
                                              else np.asarray(x0, dtype=float).reshape(1, -1))])
    Yb = None
    if spec.n_input_var > 0:
        Yb = np.zeros((1, spec.n_input_var)) if y0 is None else np.asarray(y0, dtype=float).reshape(1, -1)
    dq0 = np.zeros(nq) if dq0 is None else np.asarray(dq0, dtype=float).reshape(-1)
    evaluator = ExprEvaluator(spec, t0, Z, Yb)
    _, wv, ws = weights if weights is not None else qp_weights(spec)
    hd = np.concatenate(([mu * wv] if nvirt > 0 else []) + ([(1.0 + mu) * ws] if nslack > 0 else []))
    A_rows, lbs, ubs = [], [], []
    slack_ind = 0
    views = attribute_views(evaluator, spec.constraints)
    for ci, c_sym in enumerate(spec.constraints):
        e, Jt, Jz = evaluator.vector(c_sym.expression)
        e, Jt, Jz = e[0], Jt[0], Jz[0]
        m = e.size
        found_virt = nvirt > 0 and cs.depends_on(c_sym.expression, spec.virtual_var)     # structural, as J_virt.nnz()
        blk = np.zeros((m, nvirt + nslack))
        if found_virt:
            blk[:, :nvirt] = Jz[:, nq:nq + nvirt]
        lb = -Jt - Jz[:, :nq].dot(dq0)
        ub = lb.copy()
        kind = _cls(c_sym)
        c = views[ci][0]          # gain / bounds / target as numbers
        if kind == "EqualityConstraint":
            ke = _gain_apply(c.gain, e)
            lb, ub = lb - ke, ub - ke
        elif kind == "SetConstraint":
            lb = lb + _gain_apply(c.gain, _num(c.set_min, m) - e)
            ub = ub + _gain_apply(c.gain, _num(c.set_max, m) - e)
        elif kind == "VelocityEqualityConstraint":
            lb, ub = lb + _num(c.target, m), ub + _num(c.target, m)
        elif kind == "VelocitySetConstraint":
            lb, ub = lb + _num(c.set_min, m), ub + _num(c.set_max, m)
        found_slack = False
        if nslack > 0 and c_sym.constraint_type == "soft":
            for i in range(m):
                blk[i, nvirt + slack_ind + i] = -1.0
            slack_ind += m
            found_slack = True
        if found_virt or found_slack:
            A_rows.append(blk)
            lbs.append(lb)
            ubs.append(ub)
    if not A_rows:
        return None, None
    x = qp_solve_dense(hd, np.vstack(A_rows), np.concatenate(lbs), np.concatenate(ubs))
    return (x[:nvirt] if nvirt > 0 else None), (x[nvirt:nvirt + nslack] if nslack > 0 else None)
