"""ReactiveQPController: the skill as a small dense QP per instance, batched
on the GPU.

Same constructor, weights, options, setup and ``solve`` surface as the
reference class (reference: casclik/controllers/reactive_qp.py:10-528).  The
QP  min 1/2 v'Hv  s.t.  lbA <= A v <= ubA  with H = diag(mu*w_rob, mu*w_virt,
mu + w_slack) (:175-189) and one row block per constraint (:191-246) is
assembled and solved inside one HIP kernel; ``cs.conic``/qpOASES (:248-260,
:491-513) is replaced by an exact dual active-set solve per lane.  H is
strictly positive, so the minimiser is unique and does not depend on the
solver or on warm starts: the ``warmstart_*`` arguments are accepted and
ignored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _capi
from .. import sym as cs
from ..constraints import (EqualityConstraint, SetConstraint, VelocityEqualityConstraint,
                           VelocitySetConstraint)
from ..lowering import lower_skill
from .base_controller import (BaseController, SingleSlot, current_stream, device_of, ptr,
                              to_device_matrix, check_out_tensor, free_stream, resident_wait, ResidentWatchdog,
                              _torch)
from .pseudo_inverse import _flat, _scalar


def _weights(weights, n, what):
    """Weight vectors (reactive_qp.py:65-133): None -> ones; list / ndarray of
    the right length; CasADi-typed weights always raise in the reference
    (it compares the bound method ``weights.size2`` to 1), so symbolic weights
    are rejected here too."""
    if weights is None:
        return np.ones(n)
    if isinstance(weights, cs.MX):
        raise ValueError(what + " must be a vector.")
    if isinstance(weights, cs.DM):
        weights = weights.toarray().reshape(-1)
    if isinstance(weights, (list, tuple, np.ndarray)):
        arr = np.asarray(weights, dtype=float).reshape(-1)
        if arr.size != n:
            # (reactive_qp.py:76-78, :100-102, :129-131)
            raise ValueError(what + " and " + what[:-len("_weights")] + " dimensions do not match")
        return arr
    raise TypeError("unsupported type for " + what)


class ReactiveQPController(BaseController):
    """Reactive QP controller.

    Args:
        skill_spec (SkillSpecification): skill specification
        robot_var_weights (list): cost weights of robot_vel_var
        virtual_var_weights (list): cost weights of virtual_vel_var
        slack_var_weights (list): cost weights of the slack variables
        options (dict): accepted for API parity; solver_name/solver_opts/
            function_opts configure CasADi in the reference and have no effect
            here; ``device`` selects the GPU; ``max_iter`` caps the active-set
            iterations.
    """
    controller_type = "ReactiveQPController"
    options_info = """solver_name / solver_opts / initial_solver_opts: accepted as the reference accepts them (the solver
    here is the device active set, clik_qp_static.hpp); function_opts: `jit` (instantiate a kernel for the skill's
    structure, default True), `jit_values` (compile the skill's numbers in), `device`."""
    weight_shifter = 0.001   # mu of the eTaSL paper (reactive_qp.py:44)

    def __init__(self, skill_spec, robot_var_weights=None,
                 virtual_var_weights=None, slack_var_weights=None, options=None):
        self._handle = None
        self._lib = None
        self._has_initial = False
        self.skill_spec = skill_spec
        self.robot_var_weights = robot_var_weights
        self.virtual_var_weights = virtual_var_weights
        self.slack_var_weights = slack_var_weights
        self.options = options

    def __del__(self):
        self._release()

    def _release(self):
        if getattr(self, "_handle", None) is not None and self._lib is not None:
            try:
                self._lib.clik_qp_destroy(self._handle)
            except Exception:
                pass
            self._handle = None
        self._slot = None

    # -- weights ------------------------------------------------------------
    @property
    def robot_var_weights(self):
        return self._robot_var_weights

    @robot_var_weights.setter
    def robot_var_weights(self, weights):
        self._robot_var_weights = _weights(weights, self.skill_spec.n_robot_var,
                                           "robot_var_weights")

    @property
    def virtual_var_weights(self):
        return self._virtual_var_weights

    @virtual_var_weights.setter
    def virtual_var_weights(self, weights):
        self._virtual_var_weights = _weights(weights, self.skill_spec.n_virtual_var,
                                             "virtual_var_weights")

    @property
    def slack_var_weights(self):
        return self._slack_var_weights

    @slack_var_weights.setter
    def slack_var_weights(self, weights):
        if weights is None:
            w = []
            for cnstr in self.skill_spec.constraints:
                if cnstr.constraint_type == "soft":
                    w += [float(cnstr.slack_weight)] * cnstr.expression.size()[0]
            self._slack_var_weights = np.asarray(w, dtype=float)
        else:
            self._slack_var_weights = _weights(weights, self.skill_spec.n_slack_var,
                                               "slack_var_weights")

    # -- options (reactive_qp.py:141-173) ---------------------------------------
    @property
    def options(self):
        return self._options

    @options.setter
    def options(self, opt):
        if opt is None or not isinstance(opt, dict):
            opt = {}
        # the dictionary a caller reads back has the reference's keys and defaults (reactive_qp.py:141-173): the solver
        # entries are accepted and kept (the solver here is the device active set), `function_opts["jit"]` decides
        # whether a kernel is instantiated for the skill's structure
        opt.setdefault("solver_name", "qpoases")
        sopts = opt.setdefault("solver_opts", {})
        sopts.setdefault("print_time", False)
        if opt["solver_name"] == "qpoases":
            sopts.setdefault("printLevel", "none")
        elif opt["solver_name"] == "ooqp":
            sopts.setdefault("print_level", 0)
        sopts.setdefault("jit", True)
        sopts.setdefault("jit_options", {"flags": "-O2"})
        opt.setdefault("initial_solver_opts", sopts)
        fopts = opt.setdefault("function_opts", {})
        fopts.setdefault("jit", True)
        fopts.setdefault("print_time", False)
        fopts.setdefault("jit_options", {"flags": "-O2"})
        self._options = opt

    # -- setup ------------------------------------------------------------------
    def setup_problem_functions(self):
        """Lower the skill, upload it with the cost weights (replaces the
        H/A/Blb/Bub ``cs.Function`` objects of reactive_qp.py:262-298)."""
        self._release()
        self._lib = _capi.load_library()
        spec = self.skill_spec
        self.descriptor = lower_skill(spec)
        d = self.descriptor
        cdesc = _capi.desc_to_c(d)
        state_w = np.concatenate([self._robot_var_weights,
                                  self._virtual_var_weights[:d.n_x]])
        if self._slack_var_weights.size != d.n_slack:
            raise ValueError("slack_var_weights and slack_var dimensions do not match")
        copts = _capi.qp_opts_to_c(self.weight_shifter, state_w,
                                   self._slack_var_weights,
                                   int(self.options.get("max_iter", 0)))
        self._device = device_of(self.options.get("device"))
        handle = C.c_void_p()
        torch = _torch()
        with torch.cuda.device(self._device):
            rc = self._lib.clik_qp_create(C.byref(cdesc), C.byref(copts), C.byref(handle))
        _capi.check(self._lib, rc)
        self._handle = handle
        self.n_qp_vars = self._lib.clik_qp_n_vars(handle)
        self.n_qp_rows = self._lib.clik_qp_n_rows(handle)
        self.kernel_name = self._lib.clik_qp_kernel_name(handle).decode()
        # no AOT shape for this skill: instantiate the shape-specialised QP kernel for
        # it (the reference JIT-compiles its H/A/lbA/ubA functions here, reactive_qp.py:283-298)
        import os
        fopts = self.options.get("function_opts") or {}
        want_jit = fopts.get("jit", True) and os.environ.get("CLIK_JIT", "1") != "0" \
            and os.environ.get("CLIK_FORCE_DYNAMIC", "0") != "1"
        if self.kernel_name in ("dynamic", "none") and want_jit:
            from .. import jit
            with torch.cuda.device(self._device):
                try:
                    name = jit.attach_qp(self._lib, handle, cdesc, extern=d.extern_source())
                except RuntimeError as exc:
                    # (a failed instantiation is not fatal when a built-in kernel serves the skill: say so
                    # and run that one - still the GPU path; skills only instantiated kernels can serve
                    # fail below)
                    import warnings
                    warnings.warn("run-time kernel instantiation failed, using the built-in dynamic-shape "
                                  "kernel: %s" % str(exc)[:400])
                    name = None
            if name:
                self.kernel_name = name
        if d.extern_code and not self.kernel_name.startswith("jit_"):
            # constraints outside the row-table family exist only as generated code inside a
            # run-time instantiated kernel; there is no other path (and no CPU fallback)
            raise NotImplementedError(
                "the skill has constraint expressions that need generated device code (%s), but no "
                "kernel could be instantiated for it (jit disabled, hipcc missing, or the skill is "
                "outside the shape-specialised family)" % ", ".join(
                    repr(d.tasks[k]["label"]) for k in sorted(d.extern_code)))
        if self.kernel_name == "none":
            raise NotImplementedError(
                "the QP of this skill (%d variables x %d rows) exceeds the built-in kernel and no "
                "shape-specialised kernel could be instantiated for it (jit disabled or hipcc missing)"
                % (self.n_qp_vars, self.n_qp_rows))
        # The per-tick kernel with this skill's numbers and QP options compiled in, as the functions CasADi generates
        # for the reference are (reactive_qp.py:283-298): one more hipcc run at set-up.  By default for the box
        # family (bound-constrained after the soft equalities are folded: config 4, the UR5 notebook stacks), whose
        # value-specialised kernel needs no LDS at all (config 4: 11.3 against 12.9 us per tick, hot-started 5.6
        # against 7.4); for other skills on request (function_opts["jit_values"] = True or CLIK_JIT_VALUES=2: it keeps
        # the LDS work area and gains about 1 %); function_opts["jit_values"] = False or CLIK_JIT_VALUES=0: never.
        self.value_kernel = None
        self._value_variant_fn = None
        jv, env_jv = fopts.get("jit_values", None), os.environ.get("CLIK_JIT_VALUES", "1")
        # (skills with more than ten state variables - two arms - keep the image-reading kernel unless asked: the
        # value-specialised one holds its n x n factor in registers)
        wanted = jv is True or env_jv == "2" or (jv is None and self._lib.clik_qp_is_box_family(handle) == 1
                                                  and d.n_state <= 10)
        if want_jit and wanted and jv is not False and env_jv != "0" and self.kernel_name not in ("dynamic", "none"):
            from .. import jit
            with torch.cuda.device(self._device):
                try:
                    self.value_kernel = jit.attach_qp_values(self._lib, handle, cdesc, extern=d.extern_source())
                    if self.value_kernel:
                        fn = jit.attach_qp_values.last_library.clik_jit_qp_value_variant
                        fn.restype, fn.argtypes = C.c_char_p, [C.c_longlong, C.c_int]
                        self._value_variant_fn = fn
                except RuntimeError as exc:
                    import warnings
                    warnings.warn("value-specialised QP kernel could not be built, using the image-reading one: %s"
                                  % str(exc)[:300])

    def setup_solver(self):
        """The solver lives inside the kernel; make sure the handle exists
        (reference: cs.conic construction, reactive_qp.py:248-260)."""
        if self._handle is None:
            self.setup_problem_functions()

    def _require_handle(self):
        if self._handle is None:
            raise RuntimeError("call setup_problem_functions() / setup_solver() first")

    def setup_initial_problem_solver(self):
        """The reference builds a second QP over (virtual_vel, slack) with the robot velocity
        fixed (reactive_qp.py:300-424); its result is only used as a warm start.  Nothing to
        compile here: see solve_initial_problem."""
        spec = self.skill_spec
        self._has_initial = (spec.n_slack_var > 0) or (spec.n_virtual_var > 0
                                                      and spec._has_virtual)

    def solve_initial_problem(self, time_var0, robot_var0, virtual_var0=None,
                              robot_vel_var0=None, input_var0=None):  # noqa: D401
        """(virtual_vel0, slack0) of reactive_qp.py:426-459: the minimiser of
        ``mu w_virt |dx|^2 + (1+mu) w_slack |s|^2`` over the rows that involve virtual or slack
        variables, with the robot velocity fixed to ``robot_vel_var0`` (zeros by default).

        Without virtual variables (every UR5 notebook) the rows decouple, ``-s_i in [lb_i, ub_i]``,
        and the minimiser is each slack clamped to its interval nearest zero - evaluated here from
        the expression graph at the one initial state (setup-time host arithmetic, like the
        lowering; no solver involved).  With virtual variables the reduced QP of :300-424 is solved
        on the device (``_initial_problem_with_virtual``)."""
        if not getattr(self, "_has_initial", True):
            return None, None
        spec = self.skill_spec
        if spec.n_slack_var == 0 and not (spec.n_virtual_var > 0 and spec._has_virtual):
            return None, None
        if spec.n_virtual_var > 0 and spec._has_virtual:
            return self._initial_problem_with_virtual(time_var0, robot_var0, virtual_var0, robot_vel_var0, input_var0)
        rows, dq0 = self._host_rows(time_var0, robot_var0, None, robot_vel_var0, input_var0)
        slack = []
        for (cn, m, Jq, Jx, lb, ub) in rows:
            if cn.constraint_type != "soft":
                continue
            shift = Jq(dq0) if dq0.any() else 0.0
            # -s in [lb, ub] - J_q dq0:  s in [-ub, -lb], the point nearest zero
            slack.append(np.clip(0.0, -(ub - shift), -(lb - shift)))
        return None, cs.DM(np.concatenate(slack).reshape(-1, 1))

    def get_cost_expr(self):
        """H of  min v' H v  over v = [robot_vel; virtual_vel; slack]: diag(mu w_robot, mu w_virtual, mu + w_slack)
        (reactive_qp.py:175-189).  The kernels carry the same diagonal (`clik_qp_data_batch`); this is the reference's
        public, symbolic form."""
        mu = self.weight_shifter
        spec = self.skill_spec
        parts = [mu * np.asarray(self._robot_var_weights, dtype=float).reshape(-1)]
        if spec.n_virtual_var > 0:
            parts.append(mu * np.asarray(self._virtual_var_weights, dtype=float).reshape(-1)[:spec.n_virtual_var])
        if spec.n_slack_var > 0:
            parts.append(mu + np.asarray(self._slack_var_weights, dtype=float).reshape(-1))
        return cs.diag(cs.MX(np.concatenate(parts)))

    def get_constraints_expr(self):
        """(A, lbA, ubA) of  lbA <= A v <= ubA  as expressions of (t, q[, x][, y]) (reactive_qp.py:191-246): per
        constraint the Jacobians w.r.t. the robot and virtual variables, a block of -I in the slack columns of a soft
        constraint, and bounds  -de/dt - K e  (equality),  -de/dt + K (set_min|max - e)  (set),  -de/dt + target
        (velocity equality),  -de/dt + set_min|max  (velocity set).  The kernels evaluate these rows in closed form
        (clik_qp_static.hpp); this is the reference's public, symbolic form - expressions holding an orientation-error
        node have no symbolic derivative here and raise NotImplementedError."""
        spec = self.skill_spec
        n_slack = spec.n_slack_var
        rows, lows, highs = [], [], []
        at = 0

        def times_gain(gain, vec, m):
            if isinstance(gain, (int, float)):
                return gain * vec
            if isinstance(gain, cs.MX):
                return cs.mtimes(gain, vec) if gain.size()[1] == m and gain.size()[0] == m and m > 1 else gain * vec
            arr = np.asarray(gain.toarray() if isinstance(gain, cs.DM) else gain, dtype=float)
            if arr.size == 1:
                return float(arr.reshape(-1)[0]) * vec
            if arr.size == m:
                return cs.MX(arr.reshape(m, 1)) * vec
            return cs.mtimes(cs.MX(arr.reshape(m, m)), vec)

        def column(val, m):
            if isinstance(val, cs.MX):
                return val
            arr = np.asarray(val.toarray() if isinstance(val, cs.DM) else val, dtype=float).reshape(-1)
            return cs.MX((np.full(m, arr[0]) if arr.size == 1 else arr).reshape(m, 1))

        for cn in spec.constraints:
            m = cn.expression.size()[0]
            block = cn.jacobian(spec.robot_var)
            if spec.virtual_var is not None:
                block = cs.horzcat(block, cn.jacobian(spec.virtual_var))
            low = high = -cn.jacobian(spec.time_var)
            if isinstance(cn, EqualityConstraint):
                low = high = low - times_gain(cn.gain, cn.expression, m)
            elif isinstance(cn, SetConstraint):
                low = low + times_gain(cn.gain, column(cn.set_min, m) - cn.expression, m)
                high = high + times_gain(cn.gain, column(cn.set_max, m) - cn.expression, m)
            elif isinstance(cn, VelocityEqualityConstraint):
                low = high = low + column(cn.target, m)
            else:
                low, high = low + column(cn.set_min, m), high + column(cn.set_max, m)
            if n_slack > 0:
                slack_cols = np.zeros((m, n_slack))
                if cn.constraint_type == "soft":
                    slack_cols[:, at:at + m] = -np.eye(m)
                    at += m
                block = cs.horzcat(block, cs.MX(slack_cols))
            rows.append(block)
            lows.append(low)
            highs.append(high)
        return cs.vertcat(*rows), cs.vertcat(*lows), cs.vertcat(*highs)

    def _host_rows(self, time_var0, robot_var0, virtual_var0, robot_vel_var0, input_var0):
        """Rows of the QP at ONE state, evaluated from the expression graph on the host (setup-time
        arithmetic for the initial problem, like the lowering; never on the per-tick path): per constraint
        ``(cn, m, Jq, Jx, lb, ub)`` with ``Jq(v) = (de/dq) v`` (a callable: orientation-error nodes have no
        symbolic q-derivative and take a directional difference), ``Jx = de/dx`` (matrix or None) and the
        bounds of reactive_qp.py:219-232.  Also returns the held robot velocity as a flat array."""
        from .. import autodiff
        spec = self.skill_spec
        env = {}

        def flat(val):
            return np.asarray(val.toarray() if hasattr(val, "toarray") else val, dtype=float).reshape(-1)

        def bind(var, val):
            if var is None or var.numel() == 0:
                return
            fl = flat(val)
            fam = cs._families_of(var)[0]
            env[id(fam)] = {k: float(fl[k]) for k in range(fl.size)}

        nq = spec.n_robot_var
        bind(spec.time_var, [time_var0])
        bind(spec.robot_var, robot_var0)
        if spec.n_virtual_var > 0:
            bind(spec.virtual_var, np.zeros(spec.n_virtual_var) if virtual_var0 is None else virtual_var0)
        if spec.n_input_var > 0:
            bind(spec.input_var, np.zeros(spec.n_input_var) if input_var0 is None else input_var0)
        dq0 = np.zeros(nq) if robot_vel_var0 is None else flat(robot_vel_var0)
        out = []
        for cn in spec.constraints:
            expr = cn.expression
            m = expr.size()[0]
            e = cs.evaluate(expr, env).reshape(-1)
            base = -cs.evaluate(autodiff.jacobian(expr, spec.time_var), env).reshape(-1)

            def Jq(v, expr=expr):
                try:
                    return cs.evaluate(autodiff.jacobian(expr, spec.robot_var), env).dot(v)
                except NotImplementedError:
                    # (opaque orientation-error nodes have no symbolic q-derivative: directional difference)
                    fam = id(cs._families_of(spec.robot_var)[0])
                    q_at = dict(env[fam])
                    h = 1e-6
                    try:
                        env[fam] = {k: q_at[k] + h * v[k] for k in q_at}
                        ep = cs.evaluate(expr, env).reshape(-1)
                        env[fam] = {k: q_at[k] - h * v[k] for k in q_at}
                        em = cs.evaluate(expr, env).reshape(-1)
                    finally:
                        env[fam] = q_at
                    return (ep - em) / (2.0 * h)

            Jx = None
            if spec.n_virtual_var > 0 and cs.depends_on(expr, spec.virtual_var):
                Jx = cs.evaluate(autodiff.jacobian(expr, spec.virtual_var), env).reshape(m, spec.n_virtual_var)

            def gained(v, g=cn.gain, m=m):
                if isinstance(g, cs.MX):
                    g = cs.evaluate(g, env)       # (an expression of (t, q, y): at the initial state)
                if isinstance(g, cs.DM):
                    g = g.toarray()
                g = np.asarray(g, dtype=float)
                return float(g.reshape(-1)[0]) * v if g.size == 1 else g.reshape(m, m).dot(v)

            def vec(val, m=m):
                if isinstance(val, cs.MX):
                    val = cs.evaluate(val, env)
                a = np.asarray(val.toarray() if isinstance(val, cs.DM) else val, dtype=float).reshape(-1)
                return np.full(m, a[0]) if a.size == 1 and m > 1 else a

            if isinstance(cn, EqualityConstraint):
                lb = ub = base - gained(e)
            elif isinstance(cn, SetConstraint):
                lb, ub = base + gained(vec(cn.set_min) - e), base + gained(vec(cn.set_max) - e)
            elif isinstance(cn, VelocityEqualityConstraint):
                lb = ub = base + vec(cn.target)
            else:
                lb, ub = base + vec(cn.set_min), base + vec(cn.set_max)
            out.append((cn, m, Jq, Jx, lb, ub))
        return out, dq0

    def _initial_problem_with_virtual(self, time_var0, robot_var0, virtual_var0, robot_vel_var0, input_var0):
        """The reference's initial problem with virtual variables (reactive_qp.py:300-459), on the device:

            min  mu w_virt |dx|^2 + (1 + mu) w_slack |s|^2                                   (:321-331, D12)
            s.t. [J_virt | -I_slack] [dx; s]  in  [lbA, ubA] - J_q dq0                       (:339-391)

        over the rows of the constraints that depend on the virtual variables or carry slack, the robot
        velocity held at ``robot_vel_var0`` (zeros by default, :437-438).  ``A = [J_q | J_virt | -I]``,
        ``lbA``, ``ubA`` of the full QP come from the device (``qp_data_batch``: the kernels' FK and
        Jacobians at the initial state); the reduced QP is then a skill of its own - linear
        velocity-level rows in the ``n_virtual`` unknowns with those numbers as coefficients - and is
        solved by a second ReactiveQPController, i.e. by the same device kernels."""
        from .. import sym as cs_
        from ..skill_specification import SkillSpecification
        spec = self.skill_spec
        nq, nx, ns = spec.n_robot_var, spec.n_virtual_var, spec.n_slack_var
        as1 = lambda v, n: np.zeros((1, n)) if v is None else np.asarray(  # noqa: E731
            v.toarray() if hasattr(v, "toarray") else v, dtype=float).reshape(1, n)
        q0, x0 = as1(robot_var0, nq), as1(virtual_var0, nx)
        y0 = as1(input_var0, spec.n_input_var) if spec.n_input_var > 0 and spec._has_input else None
        dq0 = as1(robot_vel_var0, nq).reshape(-1)
        mu = self.weight_shifter
        host_rows = None
        try:
            H, A, lb, ub = self.qp_data_batch(time_var0, q0, virtual_var=x0, input_var=y0)
            H, A, lb, ub = H[0], A[0], lb[0], ub[0]
            w_virt = H[nq:nq + nx] / mu
            w_slack = H[nq + nx:] - mu                  # main problem: mu + w  (:187)
        except NotImplementedError:
            # skills with generated rows / expression attributes exist only inside their instantiated kernel,
            # which has no data entry point (clik_qp_data_batch: CLIK_EUNSUPPORTED): the same rows from the
            # expression graph at the one initial state (setup-time host arithmetic, as without virtual variables)
            host_rows, _ = self._host_rows(time_var0, q0, x0, robot_vel_var0, y0)
            w_virt = np.asarray(self._virtual_var_weights, dtype=float).reshape(-1)[:nx]
            w_slack = np.asarray(self._slack_var_weights, dtype=float).reshape(-1)
        xs = cs_.MX.sym("dx_init", nx)
        cns, slack_w, row = [], [], 0
        sl = 0
        for ci, cn in enumerate(spec.constraints):
            m = cn.expression.size()[0]
            rows = slice(row, row + m)
            row += m
            soft = cn.constraint_type == "soft"
            found_virt = cs_.depends_on(cn.expression, spec.virtual_var)       # structural, as J_virt.nnz() (:346-347)
            if soft:
                wk = w_slack[sl:sl + m]
                sl += m
            if not (found_virt or soft):
                continue
            if host_rows is None:
                Jv = A[rows, nq:nq + nx] if found_virt else np.zeros((m, nx))
                shift = A[rows, :nq].dot(dq0)
                lo, hi = lb[rows] - shift, ub[rows] - shift
            else:
                _, _, Jq, Jx, lbr, ubr = host_rows[ci]
                Jv = Jx if Jx is not None else np.zeros((m, nx))
                shift = Jq(dq0) if dq0.any() else 0.0
                lo, hi = lbr - shift, ubr - shift
            kw = dict(label="init_" + cn.label, expression=cs_.mtimes(Jv, xs), priority=len(cns),
                      constraint_type="soft" if soft else "hard")
            if np.array_equal(lo, hi):
                cns.append(VelocityEqualityConstraint(target=lo, **kw))
            else:
                cns.append(VelocitySetConstraint(set_min=lo, set_max=hi, **kw))
            if soft:
                wd = (1.0 + mu) * wk - mu                                      # mu + w' = (1 + mu) w  (:331)
                if np.any(wd + mu <= 0.0):
                    raise ValueError("initial problem: slack weights must be positive")
                slack_w.extend(wd.tolist())
        if not cns:
            return None, None
        t_ = cs_.MX.sym("t_init")
        sub = SkillSpecification(label=spec.label + "_initial", time_var=t_, robot_var=xs, constraints=cns)
        ctrl = ReactiveQPController(sub, robot_var_weights=list(w_virt), slack_var_weights=slack_w or None,
                                    options={"device": self.options.get("device")} if self.options.get("device") is not None else None)
        ctrl.weight_shifter = mu                        # (the sub-problem's H is built with this instance's mu)
        ctrl.setup_problem_functions()
        dx, _, slack, status = ctrl.solve_batch(0.0, np.zeros((1, nx)))
        if int(status[0]) == 2:
            raise RuntimeError("initial problem infeasible")
        res_slack = cs.DM(slack[0].reshape(-1, 1)) if ns > 0 else None
        return cs.DM(dx[0].reshape(-1, 1)), res_slack

    # -- per tick -----------------------------------------------------------------
    def kernel_variant(self, batch, hot=False):
        """name of the kernel a batch of ``batch`` instances gets ("/v": with the skill's numbers compiled in; ``hot``:
        for a hot-started tick).  The suffix behind "/v" is the LAUNCHER's own decision: the instantiated library
        exports the predicate it launches by (clik_jit_qp_value_variant, csrc/clik_qp_static.hpp::qp_values_choice) -
        "/folio4": cold ticks of small batches, four waves per 64 instances with different starts of the passes;
        nothing: one lane per instance."""
        name = self.kernel_name + ("/v" if getattr(self, "value_kernel", None) else "")
        fn = getattr(self, "_value_variant_fn", None)
        if getattr(self, "value_kernel", None) and fn is not None:
            torch = _torch()
            with torch.cuda.device(self._device):
                name += fn(int(batch), 1 if hot else 0).decode()
        return name

    # -- resident ticks ----------------------------------------------------------------------------------------
    def workspace_bytes(self):
        """Device memory the controller's handle holds as work area of the global-workspace QP kernels (skills beyond 16
        rows or eight states on the built-in kernels; 0 for every other skill): sized for the blocks the largest batch so
        far needed, released when the controller goes (clik_qp_workspace_bytes)."""
        self._require_handle()
        return int(self._lib.clik_qp_workspace_bytes(self._handle))

    def resident_start(self, robot_var, input_var, n_ticks, time_var=0.0, timeout_s=2.0, stream=None, ring_depth=1,
                       publish_ahead=0):
        """Launch ONE kernel that stays on the device and solves tick k's QP as soon as ticket k is published
        (include/clik.h, clik_qp_resident_run - the QP's form of PseudoInverseController.resident_start: same ticket,
        same ``done`` slots, same rings).  ``robot_var`` / ``input_var`` are device tensors read in place, ``[B, n]`` or
        with ``ring_depth`` D > 1 rings ``[D, B, n]`` (tick k uses slot ``(k - 1) % D``, outputs likewise).  Every
        instance's working set stays in the kernel: tick 1 is a cold solve, every later tick is hot-started (the
        reference's qpOASES instance hot-starts the same way, reactive_qp.py:491-513).  For bound-constrained skills
        with forward kinematics and without virtual variables; ``NotImplementedError`` otherwise.  Returns a dict with
        ``ticket``, ``done``, ``waves``, ``out`` (velocities), ``slack``, ``status`` and the launch ``stream``."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        D = int(ring_depth)
        if D < 1:
            raise ValueError("ring_depth must be at least 1")
        if d.n_x > 0:
            raise NotImplementedError("resident QP ticks: skills without virtual variables only")
        # (the batch size is robot_var's; input_var must have THAT many rows - the kernel reads y[row * n_y + ...] for
        # every row of robot_var - and a tensor of the wrong rank is a ValueError, not an IndexError)
        B = None
        lead = (D,) if D > 1 else ()
        for name, tns, n in (("robot_var", robot_var, d.n_q), ("input_var", input_var, d.n_y)):
            if n == 0:
                continue
            if not isinstance(tns, torch.Tensor) or not tns.is_cuda or tns.dtype != torch.float64 or not tns.is_contiguous():
                raise ValueError("resident ticks: %s must be a contiguous float64 device tensor (read in place)" % name)
            if tns.dim() != len(lead) + 2:
                raise ValueError("resident ticks with ring_depth %d: %s must have %d dimensions %s, got shape %s"
                                 % (D, name, len(lead) + 2, "[D, B, n]" if D > 1 else "[B, n]", list(tns.shape)))
            if B is None:
                B = int(tns.shape[-2])
            want = lead + (B, n)
            if tuple(tns.shape) != tuple(want):
                raise ValueError("resident ticks with ring_depth %d: %s must have shape %s, got %s"
                                 % (D, name, list(want), list(tns.shape)))
        dQ = torch.zeros(lead + (B, d.n_q), dtype=torch.float64, device=dev)
        slack = torch.zeros(lead + (B, d.n_slack), dtype=torch.float64, device=dev) if d.n_slack > 0 else None
        status = torch.full(lead + (B,), -1, dtype=torch.int32, device=dev)
        ticket = torch.zeros(64, dtype=torch.int32, device=dev)
        ticket[16] = D if D > 1 else 0
        if publish_ahead:
            # tickets 1 .. publish_ahead are valid before the kernel starts (their inputs are in place): no producer
            # kernel has to run beside it (tools/resident_once.py under a counter run, which serialises kernels)
            ticket[0] = int(publish_ahead)
        waves = self._lib.clik_qp_resident_waves(self._handle, B)
        done = torch.zeros(max(waves, 1), dtype=torch.int32, device=dev)
        stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        tt, ttp = _capi.tterms_arg(d.time_terms(time_var))
        torch.cuda.current_stream(dev).synchronize()       # (ticket / outputs are initialised before the kernel starts)
        with torch.cuda.device(dev):
            rc = self._lib.clik_qp_resident_run(self._handle, B, int(n_ticks), ttp, ptr(robot_var),
                                                ptr(input_var) if d.n_y > 0 else None, ptr(dQ), ptr(slack), ptr(status),
                                                ptr(ticket), ptr(done), float(timeout_s), C.c_void_p(stream.cuda_stream))
        if rc == _capi.CLIK_EUNSUPPORTED:
            raise NotImplementedError(self._lib.clik_last_error().decode())
        _capi.check(self._lib, rc)
        return {"ticket": ticket, "done": done, "waves": waves, "out": dQ, "slack": slack, "status": status,
                "stream": stream, "keep": (robot_var, input_var, tt)}

    def resident_wait(self, run):
        """Wait for a resident run to leave; ticks finished, or ``ResidentWatchdog`` (``base_controller.resident_wait``)."""
        return resident_wait(run)

    def resident_feed_stream(self):
        """see PseudoInverseController.resident_feed_stream"""
        return free_stream(self._device)

    def resident_feed(self, run, n_ticks, closed_loop=False, timeout_s=2.0, stream=None):
        """The reference producer of resident ticks (clik_ticket_feed, see PseudoInverseController.resident_feed)."""
        torch = _torch()
        dev = self._device
        stream = stream if stream is not None else free_stream(dev)
        with torch.cuda.device(dev):
            rc = self._lib.clik_ticket_feed(ptr(run["ticket"]), ptr(run["done"]), int(n_ticks), 1 if closed_loop else 0,
                                            int(run["waves"]), float(timeout_s), C.c_void_p(stream.cuda_stream))
        _capi.check(self._lib, rc)
        return stream

    def solve_batch(self, time_var, robot_var, virtual_var=None, input_var=None,
                    return_status=True, hot_set=None, use_hot=True):
        """One QP tick for a batch: returns (robot_vel [B,n_q], virtual_vel |
        None, slack [B,n_slack] | None, status [B]) - status 0 optimal,
        1 iteration cap, 2 infeasible (then the velocities are NaN).

        ``hot_set``: optional int32 device tensor [B] carrying every instance's
        working set from tick to tick (the reference's qpOASES instance hot-starts
        the same way, reactive_qp.py:491-513).  It is read when ``use_hot`` and
        always overwritten with the final working set; the minimiser is the same
        with or without it.

        ``time_var`` may hold one time per instance (robots at different phases of their
        trajectories): one launch of the per-instance-time kernel (clik_qp_solve_batch_t)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        Q, was_np = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        B = Q.shape[0]
        times = None
        if np.ndim(time_var) > 0 and np.size(time_var) > 1:
            times = np.asarray(time_var, dtype=float).reshape(-1)
            if times.size != B:
                raise ValueError("time_var has %d entries, the batch %d instances" % (times.size, B))
            time_var = float(times[0])
        elif np.ndim(time_var) > 0:
            time_var = float(np.asarray(time_var).reshape(-1)[0])
        X = Y = None
        if d.n_x > 0:
            if virtual_var is None:
                raise ValueError("skill has virtual_var: pass virtual_var")
            X, _ = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)
        if d.n_y > 0:
            if input_var is None:
                raise ValueError("skill has input_var: pass input_var")
            Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
        dQ = torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev) if d.n_x else None
        ns = d.n_slack
        SL = torch.empty((B, ns), dtype=torch.float64, device=dev) if ns else None
        status = torch.empty((B,), dtype=torch.int32, device=dev) if return_status else None
        tt, ttp = _capi.tterms_arg(d.time_terms(time_var))
        if hot_set is not None and (hot_set.dtype != torch.int32 or hot_set.numel() != B or not hot_set.is_cuda):
            raise ValueError("hot_set must be an int32 device tensor with one entry per instance")
        hot_flag = 1 if (hot_set is not None and use_hot) else 0
        T = None
        if times is not None and np.size(tt) > 0:
            uniq, inverse = np.unique(times, return_inverse=True)
            terms = np.asarray([np.asarray(d.time_terms(float(tv)), dtype=float).reshape(-1) for tv in uniq])
            T = torch.from_numpy(np.ascontiguousarray(terms[inverse])).to(dev)
        with torch.cuda.device(dev):
            if T is not None:
                rc = self._lib.clik_qp_solve_batch_t(
                    self._handle, B, ptr(T), ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX),
                    ptr(SL), ptr(status), ptr(hot_set), hot_flag, current_stream(dev))
            else:
                rc = self._lib.clik_qp_solve_batch_hot(
                    self._handle, B, ttp, ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX),
                    ptr(SL), ptr(status), ptr(hot_set), hot_flag, current_stream(dev))
        if T is not None and rc == _capi.CLIK_EUNSUPPORTED:
            # (a skill on the dynamic fallback kernel: one launch per distinct time stamp)
            for k, tv in enumerate(uniq):
                rows = torch.from_numpy(np.nonzero(inverse == k)[0]).to(dev)
                hs = None if hot_set is None else hot_set.index_select(0, rows)
                res = self.solve_batch(float(tv), Q.index_select(0, rows),
                                       virtual_var=None if X is None else X.index_select(0, rows),
                                       input_var=None if Y is None else Y.index_select(0, rows),
                                       return_status=return_status, hot_set=hs, use_hot=use_hot)
                dQ.index_copy_(0, rows, res[0])
                if dX is not None:
                    dX.index_copy_(0, rows, res[1])
                if SL is not None:
                    SL.index_copy_(0, rows, res[2])
                if status is not None:
                    status.index_copy_(0, rows, res[3])
                if hs is not None:
                    hot_set.index_copy_(0, rows, hs)
            rc = 0
        _capi.check(self._lib, rc)
        if was_np:
            return (dQ.cpu().numpy(), None if dX is None else dX.cpu().numpy(),
                    None if SL is None else SL.cpu().numpy(),
                    None if status is None else status.cpu().numpy())
        return dQ, dX, SL, status

    def rollout_batch(self, time_vars, robot_var, input_var=None, dt=0.008, max_speed=0.0, virtual_var=None,
                      method="euler"):
        """``len(time_vars)`` ticks of QP solve -> clamp(+-max_speed) -> integrate in
        one launch, the working set hot-started from tick to tick (the host loop of
        ur5_moe2016_example2.ipynb:537-545 for this controller).  ``method="euler"``: ``q += dq*dt``;
        ``method="rk4"``: classical Runge-Kutta with the controller as the right-hand side
        (casclik/integration_methods.py:17-23: k1..k4 at t, t+dt/2, t+dt/2, t+dt, each clamped).  Returns
        (q_final, dq_last, slack_last | None, status [B] = worst status met); for a skill with
        virtual variables (pass ``virtual_var``; cart_on_track_1D...ipynb cell 60) they are integrated
        alongside, unclamped: (q_final, x_final, dq_last, dx_last, slack_last | None, status)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        Q, was_np = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        if not was_np:
            Q = Q.clone()
        B = Q.shape[0]
        X = dX = None
        if d.n_x > 0:
            if virtual_var is None:
                raise ValueError("skill has virtual_var: pass virtual_var")
            X, x_np = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)
            if not x_np:
                X = X.clone()
            dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev)
        Y = None
        if d.n_y > 0:
            Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
        if method not in ("euler", "rk4"):
            raise ValueError("method must be 'euler' or 'rk4'")
        times = np.asarray(time_vars, dtype=float).reshape(-1)
        n_ticks = int(times.size)
        if method == "rk4":
            times = np.stack([times, times + 0.5 * dt, times + 0.5 * dt, times + dt], axis=1).reshape(-1)
        tt = np.concatenate([d.time_terms(t) for t in times]) if d.n_tslots else np.zeros(0)
        tt, ttp = _capi.tterms_arg(tt)
        dQ = torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        SL = torch.empty((B, d.n_slack), dtype=torch.float64, device=dev) if d.n_slack else None
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = self._lib.clik_qp_rollout_batch_m(
                self._handle, B, n_ticks, 1 if method == "rk4" else 0, float(dt), float(max_speed), ttp,
                ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX), ptr(SL), ptr(status), current_stream(dev))
        _capi.check(self._lib, rc)
        outs = (Q, dQ, SL, status) if X is None else (Q, X, dQ, dX, SL, status)
        if was_np:
            return tuple(None if o is None else o.cpu().numpy() for o in outs)
        return outs

    def bind_batch(self, robot_var, input_var=None, virtual_var=None, out=None, hot_start=False, hot_set=None):
        """Pre-bind device tensors and return ``tick(time_var=0.0)``: one kernel
        launch per call (lean path for control loops, graph capture, benchmarks).
        ``hot_start``: keep each instance's working set between ticks (see solve_batch); ``hot_set``: an int32 device
        tensor [B] to keep them in - several bound ticks given the SAME tensor hand their working sets on to each other
        (a loop whose ticks rotate through input buffers), and a tick bound with one is hot-started from its first call
        on when the tensor already holds sets (``tick.primed = True``)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        Q, _ = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        B = Q.shape[0]
        X = Y = None
        if d.n_x > 0:
            X, _ = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)
        if d.n_y > 0:
            Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
        check_out_tensor(out, (B, d.n_q), "float64", dev, "out")
        dQ = out if out is not None else torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev) if d.n_x else None
        SL = torch.empty((B, d.n_slack), dtype=torch.float64, device=dev) if d.n_slack else None
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        fn, handle, lib = self._lib.clik_qp_solve_batch_hot, self._handle, self._lib
        if hot_set is not None:
            if not (isinstance(hot_set, torch.Tensor) and hot_set.is_cuda and hot_set.dtype == torch.int32
                    and hot_set.is_contiguous() and tuple(hot_set.shape) == (B,)):
                raise ValueError("hot_set must be a contiguous int32 device tensor of shape [%d]" % B)
            hot_start = True
        hot = hot_set if hot_set is not None else (torch.zeros((B,), dtype=torch.int32, device=dev) if hot_start else None)
        args = (ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX), ptr(SL), ptr(status), ptr(hot))
        static_tt = _capi.tterms_arg(np.zeros(0)) if d.n_tslots == 0 else None
        state = {"ticks": 0}

        def tick(time_var=0.0, stream_handle=None):
            tt, ttp = static_tt if static_tt is not None else _capi.tterms_arg(d.time_terms(time_var))
            sh = stream_handle if stream_handle is not None else current_stream(dev)
            rc = fn(handle, B, ttp, *args, 1 if (hot is not None and (state["ticks"] > 0 or tick.primed)) else 0, sh)
            state["ticks"] += 1
            if rc != 0:
                _capi.check(lib, rc)

        tick.primed = False
        tick.tensors = (Q, X, Y, dQ, dX, SL, status, hot)
        tick.hot_set = hot
        tick.out, tick.slack, tick.status = dQ, SL, status
        return tick

    def qp_data_batch(self, time_var, robot_var, virtual_var=None, input_var=None):
        """H diagonal, A, lbA, ubA per instance - what the reference's
        H_func/A_func/Blb_func/Bub_func return (reactive_qp.py:483-486)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        Q, was_np = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        B = Q.shape[0]
        X = Y = None
        if d.n_x > 0:
            X, _ = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)
        if d.n_y > 0:
            Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
        nv, nc = self.n_qp_vars, self.n_qp_rows
        Hd = torch.empty((B, nv), dtype=torch.float64, device=dev)
        A = torch.empty((B, nc, nv), dtype=torch.float64, device=dev)
        lb = torch.empty((B, nc), dtype=torch.float64, device=dev)
        ub = torch.empty((B, nc), dtype=torch.float64, device=dev)
        tt, ttp = _capi.tterms_arg(d.time_terms(time_var))
        with torch.cuda.device(dev):
            rc = self._lib.clik_qp_data_batch(
                self._handle, B, ttp, ptr(Q), ptr(X), ptr(Y), ptr(Hd), ptr(A),
                ptr(lb), ptr(ub), current_stream(dev))
        _capi.check(self._lib, rc)
        if was_np:
            return Hd.cpu().numpy(), A.cpu().numpy(), lb.cpu().numpy(), ub.cpu().numpy()
        return Hd, A, lb, ub

    def solve(self, time_var, robot_var, virtual_var=None, input_var=None,
              warmstart_robot_vel_var=None, warmstart_virtual_vel_var=None,
              warmstart_slack_var=None):
        """Single-instance tick, reference signature and return convention
        (reactive_qp.py:461-528): ``(robot_vel DM, virtual_vel DM | None,
        slack DM | None)``; raises RuntimeError when the QP is infeasible (the
        reference surfaces qpOASES failure as a CasADi RuntimeError)."""
        spec = self.skill_spec
        q = _flat(robot_var, spec.n_robot_var, "robot_var")
        x = y = None
        if spec.n_virtual_var > 0:
            x = _flat(virtual_var if virtual_var is not None
                      else np.zeros(spec.n_virtual_var), spec.n_virtual_var, "virtual_var")
        if spec.n_input_var > 0:
            y = _flat(input_var if input_var is not None
                      else np.zeros(spec.n_input_var), spec.n_input_var, "input_var")
        # B = 1 through persistent pinned / device staging (one copy each way); the working set of
        # the previous call hot-starts this one, like the reference's stateful qpOASES instance
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        nq, nx, ny, ns = d.n_q, d.n_x, d.n_y, d.n_slack
        slot = getattr(self, "_slot", None)
        if slot is None:
            slot = self._slot = SingleSlot(self._device, nq + nx + ny, nq + nx + ns, 2)
            self._slot_calls = 0
        slot.in_np[:nq] = q
        if nx:
            slot.in_np[nq:nq + nx] = x
        if ny:
            slot.in_np[nq + nx:nq + nx + ny] = y
        tt, ttp = _capi.tterms_arg(d.time_terms(float(_scalar(time_var))))
        with slot.guard():
            stream = slot.begin()
            rc = self._lib.clik_qp_solve_batch_hot(
                self._handle, 1, ttp, slot.in_ptr(0), slot.in_ptr(nq) if nx else None,
                slot.in_ptr(nq + nx) if ny else None, slot.out_ptr(0), slot.out_ptr(nq) if nx else None,
                slot.out_ptr(nq + nx) if ns else None, slot.int_ptr(0), slot.int_ptr(1),
                1 if self._slot_calls > 0 else 0, stream)
            _capi.check(self._lib, rc)
            slot.download()
        self._slot_calls += 1
        status = int(slot.out_i[0])
        if status != 0:
            raise RuntimeError("ReactiveQPController: QP %s"
                               % ("infeasible" if status == 2 else "hit the iteration cap"))
        out = slot.out_f.copy()
        res_robot_vel = cs.DM(out[:nq])
        res_virtual_vel = cs.DM(out[nq:nq + nx]) if (nx and spec._has_virtual) else None
        res_slack = cs.DM(out[nq + nx:nq + nx + ns]) if ns else None
        self.res = {"x": cs.DM(out[:nq + nx + ns])}
        return res_robot_vel, res_virtual_vel, res_slack
