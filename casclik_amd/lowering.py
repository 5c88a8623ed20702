"""Lower a SkillSpecification to the flat device skill descriptor.

The reference keeps constraint expressions as CasADi graphs and lets CasADi
differentiate and JIT them per controller (reference:
casclik/controllers/pseudo_inverse.py:285-286,453-483;
reactive_qp.py:210-216,262-298).  The MI355X kernels instead consume a table of
*affine rows over task features* (see include/clik.h, ``clik_row``):

    r = a.z + b.p(z) + g.vec(R(z)) + h.o(z,y) + sum yc*y[yi] + c + tval[t_slot]

with z = [robot_var; virtual_var], p / R the tool position / rotation of the
skill's serial chain and o the quaternion orientation error.  A constraint
expression row is either one such affine row or the 2-norm of a group of them
(``cs.norm_2`` / ``cs.norm_fro`` forms of the notebooks).  Time-only
sub-expressions (trajectories) become *time slots* whose value and exact time
derivative the host evaluates once per tick.

Expressions outside this family raise ``NotImplementedError`` naming the
offending operation - there is no silent fallback.
"""
from __future__ import annotations

import numpy as np

from . import sym as cs
from . import autodiff
from .constraints import (EqualityConstraint, SetConstraint,
                          VelocityEqualityConstraint, VelocitySetConstraint)
from .urdf import JOINT_FIXED

MAX_DOF = 14
MAX_JOINTS = 12
MAX_TASKS = 24
MAX_M = 14              # > DYN_MAX_M rows only in the shape-specialised kernels
DYN_MAX_M = 8
MAX_ROWS = 128
MAX_SETS = 10
MAX_TSLOTS = 32
MAX_YTERMS = 4
MAX_QPVARS = 46
MAX_QPROWS = 32
MAX_STATIC_TASKS = 8   # SHAPE_MAX_TASKS of csrc/clik_device.hpp

ROW_HAS_Q, ROW_HAS_P, ROW_HAS_R, ROW_HAS_O, ROW_HAS_Y, ROW_HAS_T = 1, 2, 4, 8, 16, 32
OUT_AFFINE, OUT_NORM2, OUT_EXTERN = 0, 1, 2
CLS_EQ, CLS_SET, CLS_VELEQ, CLS_VELSET = 0, 1, 2, 3
ATTR_GAIN, ATTR_SET_MIN, ATTR_SET_MAX, ATTR_TARGET = 1, 2, 4, 8     # clik_task::attr_ext (include/clik.h)


class NotAffine(Exception):
    pass


class _Affine(object):
    """Affine form over the task features; ``t`` is a Scalar sub-tree that
    depends on time only (or None)."""
    __slots__ = ("a", "b", "g", "h", "y", "t", "c")

    def __init__(self):
        self.a = {}
        self.b = np.zeros(3)
        self.g = np.zeros(9)
        self.h = np.zeros(3)
        self.y = {}
        self.t = None
        self.c = 0.0

    def is_const(self):
        return (not self.a and not self.y and self.t is None
                and not self.b.any() and not self.g.any() and not self.h.any())

    def is_time_only(self):
        return (not self.a and not self.y and not self.b.any()
                and not self.g.any() and not self.h.any())

    def scaled(self, k):
        out = _Affine()
        out.a = {i: v * k for i, v in self.a.items()}
        out.b = self.b * k
        out.g = self.g * k
        out.h = self.h * k
        out.y = {i: v * k for i, v in self.y.items()}
        out.t = None if self.t is None else cs._s_mul(cs._c(k), self.t)
        out.c = self.c * k
        return out

    def plus(self, o, sign=1.0):
        out = _Affine()
        out.a = dict(self.a)
        for i, v in o.a.items():
            out.a[i] = out.a.get(i, 0.0) + sign * v
        out.b = self.b + sign * o.b
        out.g = self.g + sign * o.g
        out.h = self.h + sign * o.h
        out.y = dict(self.y)
        for i, v in o.y.items():
            out.y[i] = out.y.get(i, 0.0) + sign * v
        if o.t is None:
            out.t = self.t
        else:
            ot = o.t if sign > 0 else cs._s_neg(o.t)
            out.t = ot if self.t is None else cs._s_add(self.t, ot)
        out.c = self.c + sign * o.c
        return out

    def time_tree(self):
        """Scalar tree of the time-only form (constant folded in)."""
        base = cs._c(self.c)
        return base if self.t is None else cs._s_add(self.t, base)


class SkillDescriptor(object):
    """Python-side mirror of ``clik_skill_desc`` (include/clik.h)."""

    def __init__(self):
        self.n_q = 0
        self.n_x = 0
        self.n_y = 0
        self.joints = []       # dicts: type, q_index, R(9), p(3), axis(3)
        self.tasks = []        # dicts, see _lower_task
        self.rows = []         # dicts: a,b,g,h,c,yc,yi,t_slot,flags
        self.tslots = []       # (value Scalar tree, derivative Scalar tree)
        self.uses_fk = False
        self.quat_src = 0
        self.quat_yi = [0, 0, 0, 0]
        self.quat = [0.0, 0.0, 0.0, 1.0]
        self.time_family = None
        self.extern_code = {}  # task index -> C++ source of its ExternTask specialisation (codegen.py)

    @property
    def n_state(self):
        return self.n_q + self.n_x

    @property
    def n_tslots(self):
        return len(self.tslots)

    @property
    def n_sets(self):
        return sum(1 for t in self.tasks if t["cls"] == CLS_SET)

    @property
    def n_slack(self):
        return sum(t["m"] for t in self.tasks if t["soft"])

    def extern_source(self):
        """Generated device code of the constraints outside the row-table family
        ('' when there is none); part of the run-time instantiated kernel."""
        return "".join(self.extern_code[k] for k in sorted(self.extern_code))

    def time_terms(self, t):
        """Host-side per-tick evaluation of the time slots:
        ndarray [values..., derivatives...] of length 2*n_tslots."""
        n = len(self.tslots)
        out = np.zeros(2 * n)
        if n == 0:
            return out
        env = {id(self.time_family): {0: float(t)}} if self.time_family is not None else {}
        memo = {}
        for k, (val, der) in enumerate(self.tslots):
            out[k] = cs._eval_scalar(val, env, memo)
            out[n + k] = cs._eval_scalar(der, env, memo)
        return out


class _Lowerer(object):
    def __init__(self, spec):
        self.spec = spec
        self.desc = SkillDescriptor()
        self.fam_t = self._family(spec.time_var)
        self.fam_q = self._family(spec.robot_var)
        self.fam_x = self._family(spec.virtual_var)
        self.fam_y = self._family(spec.input_var)
        self.fam_dq = self._family(spec.robot_vel_var)
        self.fam_dx = self._family(spec.virtual_vel_var)
        self.chain = None
        self.chain_args = None
        self.quat_nodes = None
        self.desc.time_family = self.fam_t
        self.desc.n_q = spec.n_robot_var
        self.desc.n_x = spec.n_virtual_var
        self.desc.n_y = spec.n_input_var
        self._tslot_index = {}
        self._dep_memo = {}
        self._rw, self._rw_all = {}, {}
        self._prim_fk = self._prim_quat = None

    @staticmethod
    def _family(var):
        if var is None or var.numel() == 0:
            return None
        fams = cs._families_of(var)
        if len(fams) != 1:
            raise ValueError("each skill variable must come from one MX.sym call")
        return fams[0]

    # -- dependency classes ---------------------------------------------
    def _time_only(self, node):
        """True when the sub-tree references no symbol other than time."""
        k = id(node)
        if k in self._dep_memo:
            return self._dep_memo[k]
        if node.op == "const":
            r = True
        elif node.op == "sym":
            r = node.family is self.fam_t
        elif node.op in ("fk", "fk_d", "ori_err"):
            r = False
        else:
            r = all(self._time_only(a) for a in node.args)
        self._dep_memo[k] = r
        return r

    # -- which kinematics atoms stay atoms -----------------------------------------------
    # The kernels carry ONE chain instance (T_fk of one chain on one argument vector) and ONE orientation target as
    # hand-written code; the first of each in priority order.  Every other 'fk' / 'ori_err' atom - a second chain or
    # tool frame, the same chain on other arguments, a second target, a target that is neither constant nor four
    # input_var entries, a rotation that is not the primary chain's - is rewritten into the explicit expression it
    # stands for (expand.py) and takes the generated-code route (the reference accepts any expression,
    # constraints.py:21-24).
    def _scan_primary(self):
        from . import expand
        self._prim_fk = None            # (chain, args tuple)
        self._prim_quat = None          # the four quaternion Scalars of the kept orientation target
        seen = set()
        for cn in self.spec.constraints:
            arr = cs._as_array(cn.expression)
            for i in range(arr.shape[0]):
                for a in expand.atoms(arr[i, 0], seen):
                    if a.op == "fk" and self._prim_fk is None and self._chain_args_ok(a.args):
                        self._prim_fk = (a.aux[0], a.args)
        if self._prim_fk is None:
            return
        seen = set()
        for cn in self.spec.constraints:
            arr = cs._as_array(cn.expression)
            for i in range(arr.shape[0]):
                for a in expand.atoms(arr[i, 0], seen):
                    if a.op == "ori_err" and self._prim_quat is None and self._ori_shape_ok(a):
                        self._prim_quat = a.args[9:13]

    def _chain_args_ok(self, args):
        return all(a.op == "sym" and a.family is not None and a.family in (self.fam_q, self.fam_x) for a in args)

    def _keep_fk(self, node):
        return (self._prim_fk is not None and node.aux[0] is self._prim_fk[0]
                and len(node.args) == len(self._prim_fk[1])
                and all(x is y for x, y in zip(node.args, self._prim_fk[1])))

    def _ori_shape_ok(self, node):
        for k, r in enumerate(node.args[:9]):
            if r.op != "fk" or r.aux[1] != k // 3 or r.aux[2] != k % 3 or not self._keep_fk(r):
                return False
        qn = node.args[9:13]
        return all(n.op == "const" for n in qn) or \
            all(n.op == "sym" and n.family is self.fam_y and n.family is not None for n in qn)

    def _keep_ori(self, node):
        return (self._prim_quat is not None and self._ori_shape_ok(node)
                and all(x is y for x, y in zip(node.args[9:13], self._prim_quat)))

    def _explicit(self, node, all_ori=False):
        """node with the atoms the kernels do not carry written out (all_ori: every orientation error - generated code
        has no 'ori_err' atom)"""
        from . import expand
        memo = self._rw_all if all_ori else self._rw
        return expand.rewrite(node, self._keep_fk, (lambda n: False) if all_ori else self._keep_ori, memo)

    # -- feature atoms ----------------------------------------------------
    def _use_chain(self, node):
        chain = node.aux[0]
        if self.chain is None:
            self.chain = chain
            self.chain_args = node.args
            idx = []
            for a in node.args:
                if a.op != "sym" or a.family not in (self.fam_q, self.fam_x) or a.family is None:
                    raise NotImplementedError(
                        "T_fk must be called with robot_var / virtual_var "
                        "symbols (got %r)" % (a,))
                idx.append(a.index if a.family is self.fam_q
                           else self.desc.n_q + a.index)
            self._chain_state_index = idx
        else:
            if chain is not self.chain or len(node.args) != len(self.chain_args) or \
                    any(x is not y for x, y in zip(node.args, self.chain_args)):
                raise NotImplementedError(
                    "a skill may reference one kinematic chain with one "
                    "argument vector; found a second T_fk instance")

    def _use_quat(self, node):
        rn = node.args[:9]
        for k, r in enumerate(rn):
            if r.op != "fk" or r.aux[1] != k // 3 or r.aux[2] != k % 3:
                raise NotImplementedError(
                    "orientation_error: R must be T_fk(q)[:3,:3] of the skill chain")
            self._use_chain(r)
        qn = node.args[9:13]
        if self.quat_nodes is not None:
            if any(x is not y for x, y in zip(qn, self.quat_nodes)):
                raise NotImplementedError(
                    "a skill may have one orientation target quaternion")
            return
        self.quat_nodes = qn
        d = self.desc
        if all(n.op == "const" for n in qn):
            d.quat_src = 1
            d.quat = [n.value for n in qn]
        elif all(n.op == "sym" and n.family is self.fam_y for n in qn):
            d.quat_src = 2
            d.quat_yi = [n.index for n in qn]
        else:
            raise NotImplementedError(
                "orientation target must be a constant quaternion or four "
                "input_var entries")

    # -- affine extraction --------------------------------------------------
    def affine(self, node, memo):
        k = id(node)
        if k in memo:
            return memo[k]
        r = self._affine(node, memo)
        memo[k] = r
        return r

    def _time_form(self, node):
        f = _Affine()
        f.t = node
        return f

    def _affine(self, node, memo):
        op = node.op
        f = _Affine()
        if op == "const":
            f.c = node.value
            return f
        if op == "sym":
            fam = node.family
            if fam is self.fam_t:
                return self._time_form(node)
            if fam is self.fam_q:
                f.a[node.index] = 1.0
                return f
            if fam is self.fam_x and fam is not None:
                f.a[self.desc.n_q + node.index] = 1.0
                return f
            if fam is self.fam_y and fam is not None:
                f.y[node.index] = 1.0
                return f
            if fam is self.fam_dq or fam is self.fam_dx:
                raise NotImplementedError(
                    "constraint expressions must not contain velocity variables")
            raise NotImplementedError(
                "symbol '%s' is not a variable of the skill specification" % node.name)
        if op == "fk":
            self._use_chain(node)
            _, i, j = node.aux
            if j == 3:
                f.b[i] = 1.0
            else:
                f.g[3 * i + j] = 1.0
            return f
        if op == "ori_err":
            self._use_quat(node)
            f.h[node.aux] = 1.0
            return f
        if op in ("add", "sub"):
            x = self.affine(node.args[0], memo)
            y = self.affine(node.args[1], memo)
            return x.plus(y, 1.0 if op == "add" else -1.0)
        if op == "neg":
            return self.affine(node.args[0], memo).scaled(-1.0)
        if op == "mul":
            x = self.affine(node.args[0], memo)
            y = self.affine(node.args[1], memo)
            if x.is_const():
                return y.scaled(x.c)
            if y.is_const():
                return x.scaled(y.c)
            if x.is_time_only() and y.is_time_only():
                return self._time_form(node)
            raise NotAffine("product of two state/input dependent terms")
        if op == "div":
            x = self.affine(node.args[0], memo)
            y = self.affine(node.args[1], memo)
            if y.is_const():
                return x.scaled(1.0 / y.c)
            if x.is_time_only() and y.is_time_only():
                return self._time_form(node)
            raise NotAffine("division by a state/input dependent term")
        # any other operation is fine when it only involves time
        if self._time_only(node):
            return self._time_form(node)
        raise NotAffine("operation '%s' of state/input dependent terms" % op)

    # -- rows -------------------------------------------------------------
    def _tslot(self, tree):
        key = id(tree)
        if key in self._tslot_index:
            return self._tslot_index[key]
        if self.fam_t is None:
            raise NotImplementedError("time-dependent term without time_var")
        der = autodiff.diff_scalar(tree, (id(self.fam_t), 0), {})
        self.desc.tslots.append((tree, der))
        idx = len(self.desc.tslots) - 1
        if idx >= MAX_TSLOTS:
            raise NotImplementedError("more than %d time-dependent terms" % MAX_TSLOTS)
        self._tslot_index[key] = idx
        return idx

    def _emit_row(self, form):
        n = self.desc.n_state
        row = {"a": np.zeros(MAX_DOF), "b": form.b.copy(), "g": form.g.copy(),
               "h": form.h.copy(), "c": form.c, "yc": np.zeros(MAX_YTERMS),
               "yi": [0] * MAX_YTERMS, "n_y": 0, "t_slot": -1, "flags": 0}
        for i, v in form.a.items():
            if v != 0.0:
                if i >= n:
                    raise ValueError("state index out of range")
                row["a"][i] = v
                row["flags"] |= ROW_HAS_Q
        if form.b.any():
            row["flags"] |= ROW_HAS_P
        if form.g.any():
            row["flags"] |= ROW_HAS_R
        if form.h.any():
            row["flags"] |= ROW_HAS_O
        ys = [(i, v) for i, v in sorted(form.y.items()) if v != 0.0]
        if len(ys) > MAX_YTERMS:
            raise NotImplementedError(
                "more than %d input_var terms in one expression row" % MAX_YTERMS)
        for k, (i, v) in enumerate(ys):
            row["yi"][k] = i
            row["yc"][k] = v
        row["n_y"] = len(ys)
        if ys:
            row["flags"] |= ROW_HAS_Y
        if form.t is not None:
            row["t_slot"] = self._tslot(form.t)
            row["flags"] |= ROW_HAS_T
        if row["flags"] & (ROW_HAS_P | ROW_HAS_R | ROW_HAS_O):
            self.desc.uses_fk = True
        self.desc.rows.append(row)
        if len(self.desc.rows) > MAX_ROWS:
            raise NotImplementedError("skill needs more than %d affine rows" % MAX_ROWS)
        return len(self.desc.rows) - 1

    def _const_vector(self, val, m, what, label, symbolic=None):
        """Numeric m-vector from float / list / ndarray / DM / constant MX.  An MX that depends on the skill's
        variables goes to ``symbolic`` (a list that receives its m Scalar nodes; the returned numbers are then
        placeholders) - or is refused when the caller passes none."""
        if isinstance(val, cs.MX):
            if not val.is_constant():
                if symbolic is None:
                    raise NotImplementedError(
                        "%s of '%s' must be numeric (symbolic bounds/targets are "
                        "not supported on the device path)" % (what, label))
                arr = cs._as_array(val)
                flat = [arr[i, j] for i in range(arr.shape[0]) for j in range(arr.shape[1])]
                if len(flat) == 1 and m > 1:
                    flat = flat * m
                if len(flat) != m:
                    raise ValueError("%s of '%s' has %d entries, expression has %d" % (what, label, len(flat), m))
                symbolic.extend(flat)
                return np.zeros(m)
            val = cs.evaluate(val, {})
        arr = np.asarray(val._v if isinstance(val, cs.DM) else val, dtype=float).reshape(-1)
        if arr.size == 1 and m > 1:
            arr = np.full(m, arr[0])
        if arr.size != m:
            raise ValueError("%s of '%s' has %d entries, expression has %d"
                             % (what, label, arr.size, m))
        return arr

    def _lower_task(self, cnstr):
        expr = cnstr.expression
        m, cols = expr.size()
        if cols != 1:
            raise ValueError("constraint expression must be a column")
        if m > MAX_M:
            raise NotImplementedError(
                "constraint '%s' has %d rows; the device path supports up to %d"
                % (cnstr.label, m, MAX_M))
        if isinstance(cnstr, EqualityConstraint):
            cls = CLS_EQ
        elif isinstance(cnstr, SetConstraint):
            cls = CLS_SET
        elif isinstance(cnstr, VelocityEqualityConstraint):
            cls = CLS_VELEQ
        elif isinstance(cnstr, VelocitySetConstraint):
            cls = CLS_VELSET
        else:
            raise TypeError("unknown constraint class for '%s'" % cnstr.label)
        task = {"cls": cls, "m": m, "label": cnstr.label,
                "soft": 1 if cnstr.constraint_type == "soft" else 0,
                "gain_is_matrix": 0, "attr_ext": 0, "gain": np.zeros(MAX_M * MAX_M),
                "out_kind": [0] * MAX_M, "out_row0": [0] * MAX_M,
                "out_nrows": [0] * MAX_M,
                "set_min": np.zeros(MAX_M), "set_max": np.zeros(MAX_M),
                "target": np.zeros(MAX_M),
                "slack_weight": float(cnstr.slack_weight)}
        # gain: float | square ndarray/DM | constant MX  (constraints.py:32-65)
        # An attribute that is an expression of (t, q, virtual, input) - the reference allows MX gains and bounds
        # (constraints.py:35-39, :90-92, :199-206) - becomes generated code (codegen.emit_attr): attr_nodes
        # collects its Scalar nodes in the layout [gain | set_min | set_max | target]
        attr_nodes = []
        g = cnstr.gain
        if isinstance(g, cs.MX):
            if not g.is_constant():
                ga = cs._as_array(g)
                if ga.shape == (1, 1):
                    attr_nodes.append(ga[0, 0])
                elif ga.shape == (m, m):
                    task["gain_is_matrix"] = 1
                    attr_nodes.extend(ga[i, j] for i in range(m) for j in range(m))
                else:
                    raise ValueError("gain shape %s does not fit '%s'" % (ga.shape, cnstr.label))
                task["attr_ext"] |= ATTR_GAIN
                g = 0.0
            else:
                g = cs.evaluate(g, {})
        if isinstance(g, cs.DM):
            g = g.toarray()
        if task["attr_ext"] & ATTR_GAIN:
            pass
        elif isinstance(g, (float, int)):
            task["gain"][0] = float(g)
        else:
            g = np.asarray(g, dtype=float)
            if g.ndim == 1:
                raise NotImplementedError(
                    "list gains fail in the reference (cs.mtimes(list, expr) "
                    "is dimension-inconsistent); pass a float or an m x m array")
            if g.shape == (1, 1):
                task["gain"][0] = float(g[0, 0])
            elif g.shape == (m, m):
                task["gain_is_matrix"] = 1
                task["gain"][:m * m] = g.reshape(-1)
            else:
                raise ValueError("gain shape %s does not fit '%s'" % (g.shape, cnstr.label))
        if cls in (CLS_SET, CLS_VELSET):
            # an infinite bound (set_max=cs.inf, double_pendulum_2D...ipynb cell 10) becomes the value the
            # reference itself uses for "no bound" (constraints.py:199-206): the device arithmetic stays finite
            for key, val, bit in (("set_min", cnstr.set_min, ATTR_SET_MIN), ("set_max", cnstr.set_max, ATTR_SET_MAX)):
                sym = []
                v = self._const_vector(val, m, key, cnstr.label, sym)
                task[key][:m] = np.where(np.isinf(v), np.sign(v) * 1e10, v)
                if sym:
                    task["attr_ext"] |= bit
                    attr_nodes.extend(sym)
        if cls == CLS_VELEQ:
            sym = []
            task["target"][:m] = self._const_vector(cnstr.target, m, "target", cnstr.label, sym)
            if sym:
                task["attr_ext"] |= ATTR_TARGET
                attr_nodes.extend(sym)
        arr = cs._as_array(expr)
        nodes = [self._explicit(arr[i, 0]) for i in range(m)]
        attr_nodes = [self._explicit(nd, all_ori=True) for nd in attr_nodes]
        memo = {}
        mark = (len(self.desc.rows), len(self.desc.tslots))
        try:
            for i in range(m):
                node = nodes[i]
                try:
                    form = self.affine(node, memo)
                    task["out_kind"][i] = OUT_AFFINE
                    task["out_row0"][i] = self._emit_row(form)
                    task["out_nrows"][i] = 1
                except NotAffine:
                    if node.op != "norm2":
                        raise
                    forms = [self.affine(a, memo) for a in node.args]
                    forms = [f for f in forms if not (f.is_const() and f.c == 0.0)]
                    first = None
                    for f in forms:
                        r = self._emit_row(f)
                        first = r if first is None else first
                    task["out_kind"][i] = OUT_NORM2
                    task["out_row0"][i] = first if first is not None else 0
                    task["out_nrows"][i] = len(forms)
        except NotAffine:
            # outside the row table: the whole constraint becomes generated code (codegen.py)
            self._rollback(mark)
            self._lower_extern(task, [self._explicit(arr[i, 0], all_ori=True) for i in range(m)], cnstr.label)
        if attr_nodes:
            self._lower_attr(task, attr_nodes, cnstr.label)
        return task

    def _lower_attr(self, task, nodes, label):
        from . import codegen
        ti = len(self.desc.tasks)
        if ti >= MAX_STATIC_TASKS:
            raise NotImplementedError(
                "constraint '%s' has a gain / bound given as an expression: generated device code, which only the "
                "shape-specialised kernels carry (at most %d constraints)" % (label, MAX_STATIC_TASKS))
        em = codegen.TaskEmitter(self)
        try:
            code = em.emit_attr(ti, nodes)
        except NotImplementedError as why:
            raise NotImplementedError("constraint '%s' (gain / bounds): %s" % (label, why))
        self.desc.extern_code[ti] = self.desc.extern_code.get(ti, "") + code
        self._extern_keep = getattr(self, "_extern_keep", []) + [em]
        if em.uses_fk:
            self.desc.uses_fk = True

    def _rollback(self, mark):
        n_rows, n_ts = mark
        del self.desc.rows[n_rows:]
        del self.desc.tslots[n_ts:]
        self._tslot_index = {k: v for k, v in self._tslot_index.items() if v < n_ts}

    def _lower_extern(self, task, nodes, label):
        from . import codegen
        ti = len(self.desc.tasks)
        if ti >= MAX_STATIC_TASKS:
            raise NotImplementedError(
                "constraint '%s' needs generated device code, which only the shape-specialised "
                "kernels carry (at most %d constraints)" % (label, MAX_STATIC_TASKS))
        em = codegen.TaskEmitter(self)
        try:
            self.desc.extern_code[ti] = em.emit_task(ti, nodes)
        except NotImplementedError as why:
            raise NotImplementedError("constraint '%s': %s" % (label, why))
        self._extern_keep = getattr(self, "_extern_keep", []) + [em]
        for i in range(len(nodes)):
            # placeholder row per output (keeps the row layout of the kernels); HAS_P: uses the tool frame
            form = _Affine()
            r = self._emit_row(form)
            if em.uses_fk:
                self.desc.rows[r]["flags"] |= ROW_HAS_P
                self.desc.uses_fk = True
            task["out_kind"][i] = OUT_EXTERN
            task["out_row0"][i] = r
            task["out_nrows"][i] = 1

    def run(self):
        spec = self.spec
        d = self.desc
        if d.n_state > MAX_DOF:
            raise NotImplementedError(
                "n_robot_var + n_virtual_var = %d exceeds the device limit %d"
                % (d.n_state, MAX_DOF))
        if d.n_state < 1:
            raise ValueError("skill has no robot variables")
        if len(spec.constraints) > MAX_TASKS:
            raise NotImplementedError("more than %d constraints" % MAX_TASKS)
        self._scan_primary()
        for cnstr in spec.constraints:
            d.tasks.append(self._lower_task(cnstr))
        if d.n_sets > MAX_SETS:
            raise NotImplementedError(
                "%d SetConstraints give %d modes; the device path supports up "
                "to %d sets" % (d.n_sets, 2 ** d.n_sets, MAX_SETS))
        if self.chain is not None:
            if len(self.chain.joints) > MAX_JOINTS:
                raise NotImplementedError("kinematic chain longer than %d joints" % MAX_JOINTS)
            for j in self.chain.joints:
                qi = -1 if j.type == JOINT_FIXED else self._chain_state_index[j.q_index]
                d.joints.append({"type": j.type, "q_index": qi,
                                 "R": np.asarray(j.R, dtype=float).reshape(-1).copy(),
                                 "p": np.asarray(j.p, dtype=float).copy(),
                                 "axis": np.asarray(j.axis, dtype=float).copy()})
        return d


def lower_skill(spec):
    """SkillSpecification -> SkillDescriptor (raises NotImplementedError for
    skills outside the device task family)."""
    return _Lowerer(spec).run()
