"""PseudoInverseController: set-based task-priority CLIK, batched on the GPU.

Same constructor, options, setup and ``solve`` surface as the reference class
(reference: casclik/controllers/pseudo_inverse.py:10-556); ``solve_batch`` and
``rollout_batch`` are the additions that expose the batch dimension.  Where the
reference builds and JIT-compiles one CasADi function per mode
(:259-483), ``setup_problem_functions`` lowers the skill to the flat device
descriptor and uploads it; ``solve`` launches the HIP kernel (B = 1) instead
of scanning modes in Python (:512-556).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _capi
from .. import sym as cs
from ..lowering import lower_skill, DYN_MAX_M
from .base_controller import (BaseController, SingleSlot, current_stream, device_of, ptr,
                              to_device_matrix, check_out_tensor, free_stream, resident_wait, ResidentWatchdog,
                              _torch)


class PseudoInverseController(BaseController):
    """Pseudo inverse controller (Moe's set-based task priority scheme).

    Args:
        skill_spec (SkillSpecification): skill specification
        options (dict): see ``options`` setter for keys and defaults
    """
    controller_type = "PseudoInverseController"
    options_info = """feedforward (bool, True), multidim_sets (bool, False),
    converge_final_set_to_max (bool, False), pinv_method ("damped"|"standard"),
    damping_factor (float, 1e-7), function_opts (dict, accepted and ignored:
    there is no CasADi JIT on this path), device (torch device, optional)"""

    def __init__(self, skill_spec, options=None):
        self._handle = None
        self._lib = None
        self.skill_spec = skill_spec
        self.options = options
        self.current_mode = None
        self.modes = None

    def __del__(self):
        self._release()

    def _release(self):
        if getattr(self, "_handle", None) is not None and self._lib is not None:
            try:
                self._lib.clik_pinv_destroy(self._handle)
            except Exception:
                pass
            self._handle = None
        self._slot = None

    # -- options (pseudo_inverse.py:42-66) --------------------------------
    @property
    def options(self):
        return self._options

    @options.setter
    def options(self, opt):
        if opt is None:
            opt = {}
        opt.setdefault("feedforward", True)
        opt.setdefault("multidim_sets", False)
        opt.setdefault("converge_final_set_to_max", False)
        opt.setdefault("pinv_method", "damped")
        opt.setdefault("damping_factor", 1e-7)
        fopts = opt.setdefault("function_opts", {})
        fopts.setdefault("jit", True)
        fopts.setdefault("print_time", False)
        fopts.setdefault("jit_options", {"flags": "-O2"})
        self._options = opt

    # -- skill (pseudo_inverse.py:74-90) ----------------------------------
    @property
    def skill_spec(self):
        return self._skill_spec

    @skill_spec.setter
    def skill_spec(self, spec):
        cnt = spec.count_constraints()
        self.n_set_constraints = cnt["set"]
        self.n_modes = 2 ** cnt["set"]
        self.n_state_var = spec.n_robot_var + (spec.n_virtual_var
                                               if spec.virtual_var is not None else 0)
        # (pseudo_inverse.py:79-88: the stacked state and the list of its velocity symbols)
        self.state_var = (cs.vertcat(spec.robot_var, spec.virtual_var) if spec.virtual_var is not None
                          else cs.vertcat(spec.robot_var))
        self.cntrl_var = [spec.robot_vel_var] + ([spec.virtual_vel_var] if spec.virtual_var is not None else [])
        self._skill_spec = spec
        self.create_activation_map()

    def pinv(self, J):
        """The controller's pseudo-inverse of a symbolic (or numeric) matrix (pseudo_inverse.py:92-105): "standard" is
        cs.pinv; "damped" solves with J J' + lam I when J has at least as many columns as rows, else with J'J + lam I.
        The kernels evaluate the same rule (clik_pinv_static.hpp); this method is the reference's public helper."""
        J = J if isinstance(J, cs.MX) else cs.MX(J)
        if self.options["pinv_method"] == "standard":
            return cs.pinv(J)
        lam = self.options["damping_factor"]
        rows, cols = J.size()
        if cols >= rows:
            return cs.solve(cs.mtimes(J, J.T) + lam * cs.DM.eye(rows), J).T
        return cs.solve(cs.mtimes(J.T, J) + lam * cs.DM.eye(cols), J.T)

    def _tangent_cone_signature(self, cnstr):
        from ..constraints import SetConstraint
        if not isinstance(cnstr, SetConstraint):
            raise TypeError("in_tangent_cone is only available for SetConstraint")
        spec = self.skill_spec
        args, names = [spec.time_var, spec.robot_var], ["time_var", "robot_var"]
        rates, rate_names = [spec.robot_vel_var], ["robot_vel_var"]
        # de/dt along the motion: the partial time derivative plus the Jacobians times the velocity symbols
        # (pseudo_inverse.py:155-160)
        rate = cs.jacobian(cnstr.expression, spec.time_var) + cs.jtimes(cnstr.expression, spec.robot_var,
                                                                         spec.robot_vel_var)
        if spec.virtual_var is not None:
            args, names = args + [spec.virtual_var], names + ["virtual_var"]
            rates, rate_names = rates + [spec.virtual_vel_var], rate_names + ["virtual_vel_var"]
            rate = rate + cs.jtimes(cnstr.expression, spec.virtual_var, spec.virtual_vel_var)
        if spec.input_var is not None:
            args, names = args + [spec.input_var], names + ["input_var"]
        return rate, args + rates, names + rate_names

    def get_in_tangent_cone_function(self, cnstr):
        """cs.Function `(time_var, robot_var[, virtual_var][, input_var], robot_vel_var[, virtual_vel_var]) -> 0 / 1`:
        is a velocity inside the tangent cone of a one-dimensional SetConstraint (pseudo_inverse.py:132-190)?  Inside
        the set (with the 1e-12 margins): yes; below it: only when the expression increases; above: only when it
        decreases.  The kernels run this test after every mode (clik_pinv_static.hpp::cone_s); this is the reference's
        public, host-evaluated form of it."""
        rate, args, names = self._tangent_cone_signature(cnstr)
        e, lo, hi = cnstr.expression, cnstr.set_min, cnstr.set_max
        below_ok = cs.if_else(rate > 0.0, 1.0, 0.0)
        above_ok = cs.if_else(rate < 0.0, 1.0, 0.0)
        in_tc = cs.if_else(lo - e < 1e-12, cs.if_else(e - hi < 1e-12, 1.0, above_ok), below_ok)
        label = "in_tc_" + cnstr.label.replace(" ", "_")
        return cs.Function(label, args, [in_tc], names, ["in_tc_" + cnstr.label])

    def get_in_tangent_cone_function_multidim(self, cnstr):
        """The same for a multidimensional SetConstraint (pseudo_inverse.py:192-257): inside the box (margins 1e-12
        per row): yes; outside, the outward direction is the mean of the signs of the distances to both bounds; when
        every row has left its interval ("corner") the velocity must also make less than 45 degrees with the inward
        direction, otherwise a negative outward component is enough (clik_pinv_static.hpp::cone_s)."""
        rate, args, names = self._tangent_cone_signature(cnstr)
        le, ue = cnstr.expression - cnstr.set_min, cnstr.expression - cnstr.set_max
        row_above, row_below = le >= 1e-12, ue <= 1e-12
        inside = cs.logic_and(cs.dot(row_above - 1, row_above - 1) == 0, cs.dot(row_below - 1, row_below - 1) == 0)
        outward = (cs.sign(le) + cs.sign(ue)) / 2.0
        same = cs.sign(le) == cs.sign(ue)
        corner = cs.dot(same - 1, same - 1) == 0
        along = cs.dot(outward, rate)
        spread = (cs.norm_2(rate) + 1e-10) * cs.norm_2(outward)
        corner_ok = cs.if_else(along < 0.0, cs.fabs(cs.dot(-outward, rate)) / spread < np.cos(np.pi / 4), 0.0)
        going_in = cs.if_else(corner, corner_ok, along < 0.0)
        in_tc = cs.if_else(inside, 1.0, going_in)
        label = "in_tc_" + cnstr.label.replace(" ", "_")
        return cs.Function(label, args, [in_tc], names, ["in_tc_" + cnstr.label])

    def create_activation_map(self):
        """Mode order (pseudo_inverse.py:107-130): bit patterns with set 0 as
        least significant bit, stably sorted by the number of active sets."""
        n_sets = self.n_set_constraints
        if n_sets == 0:
            self.activation_map = []
            return
        patterns = [[(idx >> k) & 1 for k in range(n_sets)]
                    for idx in range(2 ** n_sets)]
        self.activation_map = sorted(patterns, key=sum)

    # -- setup --------------------------------------------------------------
    def get_problem_expressions(self):
        """Per-mode bookkeeping (which sets are active / tested); the
        arithmetic itself lives in the device kernel."""
        set_labels = [c.label for c in self.skill_spec.constraints
                      if c.constraint_class == "SetConstraint"]
        modes = []
        for mode_idx in range(self.n_modes):
            bits = self.activation_map[mode_idx] if self.activation_map else []
            modes.append({
                "active_set_names": [l for l, b in zip(set_labels, bits) if b],
                "in_tangent_cone_set_names": [l for l, b in zip(set_labels, bits) if not b],
            })
        self.modes = modes
        return modes

    def setup_problem_functions(self):
        """Lower the skill and create the device handle (replaces the
        per-mode ``cs.Function`` JIT of pseudo_inverse.py:453-483)."""
        self.get_problem_expressions()
        self._release()
        self._lib = _capi.load_library()
        self.descriptor = lower_skill(self.skill_spec)
        cdesc = _capi.desc_to_c(self.descriptor)
        copts = _capi.pinv_opts_to_c(self.options)
        self._device = device_of(self.options.get("device"))
        handle = C.c_void_p()
        torch = _torch()
        with torch.cuda.device(self._device):
            rc = self._lib.clik_pinv_create(C.byref(cdesc), C.byref(copts), C.byref(handle))
        _capi.check(self._lib, rc)
        self._handle = handle
        self.kernel_name = self._lib.clik_pinv_kernel_name(handle).decode()
        # no AOT shape for this skill: instantiate the static templates for it
        # (the reference JIT-compiles at this point too, function_opts["jit"])
        import os
        want_jit = self.options["function_opts"].get("jit", True) and os.environ.get("CLIK_JIT", "1") != "0" \
            and os.environ.get("CLIK_FORCE_DYNAMIC", "0") != "1"
        if self.kernel_name in ("dynamic", "none") and want_jit:
            from .. import jit
            with torch.cuda.device(self._device):
                try:
                    name = jit.attach(self._lib, handle, cdesc, copts, extern=self.descriptor.extern_source())
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
        # The skill's kernels with its own numbers compiled in (the reference's JIT compiles its functions with the
        # constants of the skill too): for the config-3 family (four lanes per instance at small batches, one lane per
        # instance above) and for single-mode skills without virtual variables (BASELINE configs 1 and 2: one lane
        # per instance, no LDS) - one more hipcc run at set-up.  function_opts["jit_values"] = False or
        # CLIK_JIT_VALUES=0 keeps the kernels that read the skill image from memory.
        self.value_kernel = None
        jv = self.options["function_opts"].get("jit_values", None)
        env_jv = os.environ.get("CLIK_JIT_VALUES", "1")
        variant1 = self._lib.clik_pinv_kernel_variant(handle, 1).decode()
        dd = self.descriptor
        single_mode = (dd.n_x == 0 and dd.n_sets == 0 and variant1 in ("lane", "split"))
        wanted = (variant1 == "team4" or single_mode) and jv is not False
        if want_jit and wanted and env_jv != "0":
            from .. import jit
            with torch.cuda.device(self._device):
                try:
                    self.value_kernel = jit.attach_values(self._lib, handle, cdesc, copts,
                                                          extern=self.descriptor.extern_source())
                except RuntimeError as exc:
                    import warnings
                    warnings.warn("value-specialised kernel could not be built, using the image-reading one: %s"
                                  % str(exc)[:300])
        if self.kernel_name == "none":
            raise NotImplementedError(
                "a constraint of this skill has more rows than the built-in kernels are wide (%d) and no "
                "shape-specialised kernel could be instantiated for it (jit disabled, hipcc missing, or the "
                "skill is outside the shape-specialised family)" % DYN_MAX_M)
        if self.descriptor.extern_code and not self.kernel_name.startswith("jit_"):
            # constraints outside the row-table family exist only as generated code inside a
            # run-time instantiated kernel; there is no other path (and no CPU fallback)
            raise NotImplementedError(
                "the skill has constraint expressions that need generated device code (%s), but no "
                "kernel could be instantiated for it (jit disabled, hipcc missing, or the skill is "
                "outside the shape-specialised family)" % ", ".join(
                    repr(self.descriptor.tasks[k]["label"]) for k in sorted(self.descriptor.extern_code)))

    def kernel_variant(self, batch):
        """``<kernel>/<variant>`` serving a batch of that many instances: ``team4`` (four lanes per
        instance), ``mp2`` / ``mp4`` (one wave per mode), ``split``, ``lane`` (one instance per lane)."""
        return "%s/%s" % (self.kernel_name, self._lib.clik_pinv_kernel_variant(self._handle, int(batch)).decode())

    def setup_solver(self):
        """Reference parity: re-runs the problem setup (pseudo_inverse.py:506-510)."""
        self.setup_problem_functions()

    def setup_initial_problem_solver(self):
        """Does nothing, as in the reference (pseudo_inverse.py:485-488)."""
        pass

    def solve_initial_problem(self, time_var0, robot_var0, virtual_var0=None,
                              robot_vel_var0=None, input_var0=None):
        """Zeros, as in the reference (pseudo_inverse.py:490-504)."""
        spec = self.skill_spec
        res_virt = cs.DM.zeros(spec.n_virtual_var, 1) if virtual_var0 is not None else None
        res_slack = cs.DM.zeros(spec.n_slack_var, 1) if spec.slack_var is not None else None
        return res_virt, res_slack

    # -- per-tick -------------------------------------------------------------
    def _require_handle(self):
        if self._handle is None:
            raise RuntimeError("call setup_problem_functions() / setup_solver() first")

    def solve_batch(self, time_var, robot_var, virtual_var=None, input_var=None,
                    out=None, return_mode=True):
        """One controller tick for a batch.

        robot_var [B, n_q], virtual_var [B, n_x], input_var [B, n_y] as numpy
        arrays or torch tensors (tensors on the controller's device are used
        in place).  ``time_var``: one time stamp for the batch, or an array with one per
        instance (then one launch per distinct time stamp).  Returns (robot_vel [B,n_q], virtual_vel | None, mode [B])
        in the container type of ``robot_var``.  The launch is asynchronous on
        torch's current stream when tensors are passed."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        if np.ndim(time_var) > 0 and np.size(time_var) > 1:
            return self._solve_batch_per_instance_time(time_var, robot_var, virtual_var, input_var, out, return_mode)
        time_var = float(np.asarray(time_var).reshape(-1)[0]) if np.ndim(time_var) > 0 else time_var
        Q, was_np = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        B = Q.shape[0]
        X = None
        if d.n_x > 0:
            if virtual_var is None:
                raise ValueError("skill has virtual_var: pass virtual_var")
            X, _ = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)
        Y = None
        if d.n_y > 0:
            if input_var is None:
                raise ValueError("skill has input_var: pass input_var")
            Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
        if out is not None:
            check_out_tensor(out, (B, d.n_q), "float64", dev, "out")
            dQ = out
        else:
            dQ = torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev) if d.n_x else None
        mode = torch.empty((B,), dtype=torch.int32, device=dev) if return_mode else None
        tt, ttp = _capi.tterms_arg(d.time_terms(time_var))
        with torch.cuda.device(dev):
            rc = self._lib.clik_pinv_solve_batch(
                self._handle, B, ttp, ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX),
                ptr(mode), current_stream(dev))
        _capi.check(self._lib, rc)
        if was_np:
            return (dQ.cpu().numpy(), None if dX is None else dX.cpu().numpy(),
                    None if mode is None else mode.cpu().numpy())
        return dQ, dX, mode

    def _solve_batch_per_instance_time(self, times, robot_var, virtual_var, input_var, out, return_mode):
        """``time_var`` with one entry per instance (robots at different phases of a trajectory): ONE launch of the
        per-instance-time kernel (clik_pinv_solve_batch_t: the time-only sub-expressions are evaluated on the host
        per distinct time stamp and travel as a [B, 2 * n_tslots] device array).  A skill served by the dynamic
        fallback kernel has no such variant: its batch is grouped by distinct time stamps, one launch per group."""
        torch = _torch()
        d = self.descriptor
        dev = self._device
        Q, was_np = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
        B = Q.shape[0]
        times = np.asarray(times, dtype=float).reshape(-1)
        if times.size != B:
            raise ValueError("time_var has %d entries, the batch %d instances" % (times.size, B))
        X = to_device_matrix(virtual_var, d.n_x, dev, "virtual_var", B)[0] if d.n_x > 0 else None
        Y = to_device_matrix(input_var, d.n_y, dev, "input_var", B)[0] if d.n_y > 0 else None
        if out is not None:
            check_out_tensor(out, (B, d.n_q), "float64", dev, "out")
        dQ = out if out is not None else torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev) if d.n_x else None
        mode = torch.empty((B,), dtype=torch.int32, device=dev) if return_mode else None
        uniq, inverse = np.unique(times, return_inverse=True)
        terms = np.asarray([np.asarray(d.time_terms(float(tv)), dtype=float).reshape(-1) for tv in uniq])
        rc = _capi.CLIK_EUNSUPPORTED
        if terms.shape[1] == 0:
            # no time-dependent sub-expression: every time stamp gives the same tick - the ordinary single launch
            res = self.solve_batch(float(times[0]), Q, virtual_var=X, input_var=Y, out=out, return_mode=return_mode)
            if was_np:
                return tuple(None if r is None else r.cpu().numpy() for r in res)
            return res
        if terms.shape[1] > 0:
            T = torch.from_numpy(np.ascontiguousarray(terms[inverse])).to(dev)
            with torch.cuda.device(dev):
                rc = self._lib.clik_pinv_solve_batch_t(self._handle, B, ptr(T), ptr(Q), ptr(X), ptr(Y), ptr(dQ),
                                                       ptr(dX), ptr(mode), current_stream(dev))
            if rc != _capi.CLIK_EUNSUPPORTED:
                _capi.check(self._lib, rc)
        for k, tv in enumerate(uniq if rc == _capi.CLIK_EUNSUPPORTED else ()):
            rows = torch.from_numpy(np.nonzero(inverse == k)[0]).to(dev)
            res = self.solve_batch(float(tv), Q.index_select(0, rows),
                                   virtual_var=None if X is None else X.index_select(0, rows),
                                   input_var=None if Y is None else Y.index_select(0, rows), return_mode=return_mode)
            dQ.index_copy_(0, rows, res[0])
            if dX is not None:
                dX.index_copy_(0, rows, res[1])
            if mode is not None:
                mode.index_copy_(0, rows, res[2])
        if was_np:
            return (dQ.cpu().numpy(), None if dX is None else dX.cpu().numpy(),
                    None if mode is None else mode.cpu().numpy())
        return dQ, dX, mode

    def bind_batch(self, robot_var, input_var=None, virtual_var=None, out=None,
                   mode_out=None, stream=None):
        """Pre-bind device tensors and return ``tick(time_var=0.0)``: one
        kernel launch per call with no per-call tensor handling (the lean path
        for control loops, CUDA-graph capture and benchmarks).  All tensors
        must already live on the controller's device."""
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
        check_out_tensor(mode_out, (B,), "int32", dev, "mode_out")
        dQ = out if out is not None else torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        dX = torch.empty((B, d.n_x), dtype=torch.float64, device=dev) if d.n_x else None
        mode = mode_out if mode_out is not None else torch.empty((B,), dtype=torch.int32, device=dev)
        fn = self._lib.clik_pinv_solve_batch
        handle = self._handle
        args = (ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX), ptr(mode))
        keep = (Q, X, Y, dQ, dX, mode)
        static_tt = None
        if d.n_tslots == 0:
            static_tt = _capi.tterms_arg(np.zeros(0))
        lib = self._lib

        def tick(time_var=0.0, stream_handle=None):
            tt, ttp = static_tt if static_tt is not None else _capi.tterms_arg(d.time_terms(time_var))
            sh = stream_handle if stream_handle is not None else current_stream(dev)
            rc = fn(handle, B, ttp, args[0], args[1], args[2], args[3], args[4], args[5], sh)
            if rc != 0:
                _capi.check(lib, rc)

        tick.tensors = keep
        tick.out = dQ
        tick.mode = mode
        return tick

    # -- resident ticks ----------------------------------------------------------------------------------------
    def resident_start(self, robot_var, input_var, n_ticks, time_var=0.0, out=None, mode_out=None, timeout_s=2.0,
                       stream=None, ring_depth=1, integrate_dt=0.0, max_speed=0.0, publish_ahead=0):
        """Launch ONE kernel that stays on the device and runs up to ``n_ticks`` ticks, each as soon as its ticket
        is published (include/clik.h, clik_pinv_resident_run): for closed loops whose inputs are produced on the
        device (or copied in behind a stream) every tick, at the price of a device-side hand-off instead of a launch.
        ``robot_var`` / ``input_var`` must be device tensors (the producer overwrites them in place); with
        ``ring_depth`` D > 1 they are rings ``[D, B, n]`` and tick k uses slot ``(k - 1) % D`` (outputs likewise), so
        that a producer can write the next tick's rows while this one runs.  ``integrate_dt`` > 0 keeps the state in
        the kernel: ``robot_var`` is read at tick 1 only and then stepped with ``q += clamp(dq, +-max_speed) * dt``
        after every tick (the notebooks' loop), so that only the targets ``input_var`` come from outside.  Returns a dict
        with the ``ticket`` (int32 device tensor of 64 words: [0] in_seq, [32] stop, [48] waves, [49] ticks_done),
        ``done`` (int32 device tensor, one slot per wave: the last tick that wave finished), ``waves`` per tick,
        ``out`` and ``mode`` tensors and the launch ``stream``.  The kernel
        leaves after ``n_ticks``, on ``ticket[32] != 0`` or when its poll budget (``timeout_s`` at a nominal 0.2 us per
        poll) is used up, whatever happens.  Whoever feeds it (copies, producer kernels) must use a stream that does not
        share a hardware queue with ``stream``: ``resident_feed_stream()`` hands one out (asking for another priority is
        not enough: which queue a new stream lands on depends on how many the process has made)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        D = int(ring_depth)
        if D < 1:
            raise ValueError("ring_depth must be at least 1")
        if not isinstance(robot_var, torch.Tensor) or not robot_var.is_cuda:
            raise ValueError("resident ticks: robot_var must be a device tensor")
        if D > 1:
            # a RING of D slots: robot_var [D, B, n_q], input_var [D, B, n_y]; tick k reads and writes slot (k - 1) % D
            # (include/clik.h): a producer fills slot k % D while tick k runs and publishes ticket k + 1 ahead
            if robot_var.dim() != 3 or robot_var.shape[0] != D or robot_var.shape[2] != d.n_q:
                raise ValueError("resident ticks with ring_depth %d: robot_var must have shape [%d, B, %d]" % (D, D, d.n_q))
            B = int(robot_var.shape[1])
            Q = robot_var
            Y = None
            if d.n_y > 0:
                if (not isinstance(input_var, torch.Tensor) or not input_var.is_cuda
                        or tuple(input_var.shape) != (D, B, d.n_y)):
                    raise ValueError("resident ticks with ring_depth %d: input_var must be a device tensor of shape "
                                     "[%d, %d, %d]" % (D, D, B, d.n_y))
                Y = input_var
            for tns in (Q, Y):
                if tns is not None and (tns.dtype != torch.float64 or not tns.is_contiguous()):
                    raise ValueError("resident ticks: inputs must be contiguous float64 device tensors (read in place)")
            out_shape, mode_shape = (D, B, d.n_q), (D, B)
        else:
            Q, _ = to_device_matrix(robot_var, d.n_q, dev, "robot_var")
            B = Q.shape[0]
            Y = None
            if d.n_y > 0:
                if not isinstance(input_var, torch.Tensor) or not input_var.is_cuda:
                    raise ValueError("resident ticks: input_var must be a device tensor")
                Y, _ = to_device_matrix(input_var, d.n_y, dev, "input_var", B)
            if Q.data_ptr() != robot_var.data_ptr() or (Y is not None and Y.data_ptr() != input_var.data_ptr()):
                raise ValueError("resident ticks: inputs must be contiguous float64 device tensors (they are read in place)")
            out_shape, mode_shape = (B, d.n_q), (B,)
        check_out_tensor(out, out_shape, "float64", dev, "out")
        check_out_tensor(mode_out, mode_shape, "int32", dev, "mode_out")
        dQ = out if out is not None else torch.zeros(out_shape, dtype=torch.float64, device=dev)
        mode = mode_out if mode_out is not None else torch.full(mode_shape, -1, dtype=torch.int32, device=dev)
        ticket = torch.zeros(64, dtype=torch.int32, device=dev)
        ticket[16] = D if D > 1 else 0
        if publish_ahead:
            # tickets 1 .. publish_ahead are valid before the kernel starts (the inputs of those ticks are in place):
            # no producer has to run next to it - what a profiler that serialises kernels needs
            ticket[0] = int(publish_ahead)

        waves = self._lib.clik_pinv_resident_waves(self._handle, B)
        done = torch.zeros(max(waves, 1), dtype=torch.int32, device=dev)
        stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        tt, ttp = _capi.tterms_arg(d.time_terms(time_var))
        torch.cuda.current_stream(dev).synchronize()       # (ticket / outputs are initialised before the kernel starts)
        with torch.cuda.device(dev):
            if integrate_dt > 0.0:
                # the state stays in the kernel (include/clik.h): q is read at tick 1 and stepped with
                # q += clamp(dq, +-max_speed) * integrate_dt after every tick; only input_var comes from outside
                rc = self._lib.clik_pinv_resident_run_state(
                    self._handle, B, int(n_ticks), ttp, ptr(Q), ptr(Y), ptr(dQ), ptr(mode), ptr(ticket), ptr(done),
                    float(integrate_dt), float(max_speed), float(timeout_s), C.c_void_p(stream.cuda_stream))
            else:
                rc = self._lib.clik_pinv_resident_run(self._handle, B, int(n_ticks), ttp, ptr(Q), ptr(Y), ptr(dQ),
                                                      ptr(mode), ptr(ticket), ptr(done), float(timeout_s),
                                                      C.c_void_p(stream.cuda_stream))
        _capi.check(self._lib, rc)
        return {"ticket": ticket, "done": done, "waves": waves, "out": dQ, "mode": mode, "stream": stream,
                "keep": (Q, Y, tt)}

    def resident_wait(self, run):
        """Wait for a resident run to leave; ticks finished, or ``ResidentWatchdog`` (``base_controller.resident_wait``)."""
        return resident_wait(run)

    def resident_feed_stream(self):
        """A stream for whoever feeds a resident run that is ALREADY launched (copies, producer kernels): one whose work
        makes progress beside the resident kernel (``base_controller.free_stream``; the runtime may have put a new stream
        onto the resident kernel's hardware queue, where it would wait for the kernel's watchdog)."""
        return free_stream(self._device)

    def resident_feed(self, run, n_ticks, closed_loop=False, timeout_s=2.0, stream=None):
        """The reference producer of resident ticks (clik_ticket_feed): one device thread that publishes tickets
        1 .. n_ticks on ``stream`` (a stream of its own by default), back to back or - ``closed_loop`` - each only after
        every wave has finished the previous tick."""
        torch = _torch()
        dev = self._device
        # (a stream of another PRIORITY: the runtime multiplexes streams of one priority onto a few hardware queues,
        # and a producer queued behind the resident kernel would wait for it to leave - see include/clik.h)
        stream = stream if stream is not None else free_stream(dev)
        with torch.cuda.device(dev):
            rc = self._lib.clik_ticket_feed(ptr(run["ticket"]), ptr(run["done"]), int(n_ticks), 1 if closed_loop else 0,
                                            int(run["waves"]),
                                            float(timeout_s), C.c_void_p(stream.cuda_stream))
        _capi.check(self._lib, rc)
        return stream

    def rollout_batch(self, time_vars, robot_var, input_var=None, dt=0.008,
                      max_speed=0.0, virtual_var=None, method="euler"):
        """``len(time_vars)`` ticks of solve -> clamp(+-max_speed) -> integrate in one launch.
        ``method="euler"``: ``q += dq*dt``, the host loop of ur5_moe2016_example2.ipynb:537-545;
        ``method="rk4"``: classical Runge-Kutta with the controller as the right-hand side
        (casclik/integration_methods.py:17-23: k1..k4 at t, t+dt/2, t+dt/2, t+dt, each clamped).
        Returns (q_final, dq_last, mode_last); for a skill with virtual variables (path following,
        cart_on_track_1D...ipynb cell 60: pass ``virtual_var``) the path parameters are integrated
        alongside, unclamped, and the result is (q_final, x_final, dq_last, dx_last, mode_last)."""
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        dev = self._device
        if method not in ("euler", "rk4"):
            raise ValueError("method must be 'euler' or 'rk4'")
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
        times = np.asarray(time_vars, dtype=float).reshape(-1)
        if method == "rk4":
            stage_times = np.stack([times, times + 0.5 * dt, times + 0.5 * dt, times + dt], axis=1).reshape(-1)
        else:
            stage_times = times
        tt = np.concatenate([d.time_terms(t) for t in stage_times]) if d.n_tslots else np.zeros(0)
        tt, ttp = _capi.tterms_arg(tt)
        dQ = torch.empty((B, d.n_q), dtype=torch.float64, device=dev)
        mode = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = self._lib.clik_pinv_rollout_batch_m(
                self._handle, B, int(times.size), 1 if method == "rk4" else 0, float(dt), float(max_speed), ttp,
                ptr(Q), ptr(X), ptr(Y), ptr(dQ), ptr(dX), ptr(mode), current_stream(dev))
        _capi.check(self._lib, rc)
        outs = (Q, X, dQ, dX, mode) if d.n_x > 0 else (Q, dQ, mode)
        return tuple(o.cpu().numpy() for o in outs) if was_np else outs

    def solve(self, time_var, robot_var, virtual_var=None, input_var=None,
              warmstart_robot_vel_var=None, warmstart_virtual_vel_var=None,
              warmstart_slack_var=None):
        """Single-instance tick with the reference's signature and return
        convention (pseudo_inverse.py:512-556): ``(robot_vel DM n x 1,
        virtual_vel DM | None, None)`` and ``self.current_mode``."""
        spec = self.skill_spec
        q = _flat(robot_var, spec.n_robot_var, "robot_var")
        x = None
        if spec.n_virtual_var > 0:
            # the reference forwards virtual_var only when it is used (:521-525);
            # the device descriptor always carries the full state vector
            x = _flat(virtual_var if virtual_var is not None
                      else np.zeros(spec.n_virtual_var), spec.n_virtual_var, "virtual_var")
        y = None
        if spec.n_input_var > 0:
            y = _flat(input_var if input_var is not None
                      else np.zeros(spec.n_input_var), spec.n_input_var, "input_var")
        # B = 1 through persistent pinned / device staging (one copy each way)
        self._require_handle()
        torch = _torch()
        d = self.descriptor
        nq, nx, ny = d.n_q, d.n_x, d.n_y
        slot = getattr(self, "_slot", None)
        if slot is None:
            slot = self._slot = SingleSlot(self._device, nq + nx + ny, nq + nx, 1)
        slot.in_np[:nq] = q
        if nx:
            slot.in_np[nq:nq + nx] = x
        if ny:
            slot.in_np[nq + nx:nq + nx + ny] = y
        tt, ttp = _capi.tterms_arg(d.time_terms(float(_scalar(time_var))))
        with slot.guard():
            stream = slot.begin()
            rc = self._lib.clik_pinv_solve_batch(
                self._handle, 1, ttp, slot.in_ptr(0), slot.in_ptr(nq) if nx else None,
                slot.in_ptr(nq + nx) if ny else None, slot.out_ptr(0), slot.out_ptr(nq) if nx else None,
                slot.int_ptr(0), stream)
            _capi.check(self._lib, rc)
            slot.download()
        self.current_mode = int(slot.out_i[0])
        cntrl_rob = cs.DM(slot.out_f[:nq].copy())
        cntrl_virt = None
        if spec.n_virtual_var > 0 and virtual_var is not None and spec._has_virtual:
            cntrl_virt = cs.DM(slot.out_f[nq:nq + nx].copy())
        return cntrl_rob, cntrl_virt, None


def _scalar(v):
    if hasattr(v, "toarray"):
        v = v.toarray()
    return np.asarray(v, dtype=float).reshape(-1)[0]


def _flat(v, n, what):
    if hasattr(v, "toarray"):
        v = v.toarray()
    arr = np.asarray(v, dtype=np.float64).reshape(-1)
    if arr.size != n:
        raise ValueError("%s must have %d entries, got %d" % (what, n, arr.size))
    return arr
