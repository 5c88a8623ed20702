"""ctypes front of oracle/clik_oracle_c.c - TEST INFRASTRUCTURE / CPU BASELINE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  It re-uses the descriptor struct mirrors of casclik_amd._capi (data
layout only) but none of the product's compute.  The descriptor it feeds the C
code is the one casclik_amd.lowering produces from the skill script (then the C
code witnesses the kernels' algebra, not the front-end) or, for the BASELINE
skills, the one oracle/baseline_desc.py writes down without the front-end
(``baseline=(robot, which)``: the full-size parity tests and bench.py's
cpu_baseline use that one).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from casclik_amd import _capi
from casclik_amd.lowering import lower_skill

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libclik_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "clik_oracle_c.c")
    hdr = os.path.join(_HERE, "..", "include", "clik.h")
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return LIB
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        build()         # (rebuilds when the source or include/clik.h is newer: the descriptor layout is shared)
        _lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        _lib.orc_pinv_solve_batch.restype = C.c_int
        _lib.orc_pinv_solve_batch.argtypes = [
            C.POINTER(_capi.clik_skill_desc), C.POINTER(_capi.clik_pinv_opts),
            C.c_int64, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_int32), C.c_int]
        _lib.orc_pinv_solve_batch_m.restype = C.c_int
        _lib.orc_pinv_solve_batch_m.argtypes = _lib.orc_pinv_solve_batch.argtypes + [dp]
        _lib.orc_task_eval.restype = C.c_int
        _lib.orc_task_eval.argtypes = [C.POINTER(_capi.clik_skill_desc), C.c_int,
                                       dp, dp, dp, dp, dp, dp]
        _lib.orc_qp_data_batch.restype = C.c_int
        _lib.orc_qp_data_batch.argtypes = [
            C.POINTER(_capi.clik_skill_desc), C.POINTER(_capi.clik_qp_opts),
            C.c_int64, dp, dp, dp, dp, dp, dp, dp, dp]
        _lib.orc_qp_solve_batch.restype = C.c_int
        _lib.orc_qp_solve_batch.argtypes = [
            C.POINTER(_capi.clik_skill_desc), C.POINTER(_capi.clik_qp_opts),
            C.c_int64, dp, dp, dp, dp, dp, C.POINTER(C.c_int32), C.c_int]
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(arr):
    return None if arr is None else arr.ctypes.data_as(C.POINTER(C.c_double))


def _full_pinv_options(options):
    opt = dict(options or {})
    opt.setdefault("feedforward", True)
    opt.setdefault("multidim_sets", False)
    opt.setdefault("converge_final_set_to_max", False)
    opt.setdefault("pinv_method", "damped")
    opt.setdefault("damping_factor", 1e-7)
    return opt


class _FlatDesc(object):
    """what the wrappers below need of a descriptor that did not come from the product's lowering"""
    extern_code = None

    def __init__(self, cdesc):
        self.n_q, self.n_x, self.n_y = int(cdesc.n_q), int(cdesc.n_x), int(cdesc.n_y)
        self.n_state = self.n_q + self.n_x
        self.tasks = [{"m": int(cdesc.tasks[k].m), "soft": int(cdesc.tasks[k].soft),
                       "slack_weight": float(cdesc.tasks[k].slack_weight)} for k in range(int(cdesc.n_tasks))]
        self.n_slack = sum(t["m"] for t in self.tasks if t["soft"])
        assert int(cdesc.n_tslots) == 0

    def time_terms(self, t):
        return np.zeros(0)


class CPinvOracle(object):
    """Literal PseudoInverseController on the CPU from a flat descriptor: the one the product's front-end
    lowers from ``spec``, or - ``baseline=(robot, which)`` - the independently written one of
    oracle/baseline_desc.py (BASELINE skills only), in which case nothing of casclik_amd's front-end is involved."""

    def __init__(self, spec, options=None, baseline=None):
        self.lib = load()
        if baseline is not None:
            from . import baseline_desc
            self.cdesc, _ = baseline_desc.baseline_descriptor(*baseline)
            self.desc = _FlatDesc(self.cdesc)
        else:
            self.desc = lower_skill(spec)
            if self.desc.extern_code:
                # (this restatement reads the lowered row table; expression-graph constraints are
                # covered by the numpy oracle only)
                raise NotImplementedError("the C oracle has no rows for constraints outside the row table")
            self.cdesc = _capi.desc_to_c(self.desc)
        self.copts = _capi.pinv_opts_to_c(_full_pinv_options(options))

    def solve_batch(self, t, Q, X=None, Y=None, nthreads=0, margins_out=None):
        """``margins_out`` [B] (optional): per instance the smallest distance of any tangent-cone decision of its
        mode scan from flipping (clik_oracle.tangent_cone_margin)"""
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        B = Q.shape[0]
        Xc = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
        Yc = None if Y is None else np.ascontiguousarray(Y, dtype=np.float64)
        tt = np.ascontiguousarray(self.desc.time_terms(t))
        dq = np.zeros((B, self.desc.n_q))
        dx = np.zeros((B, max(self.desc.n_x, 1)))
        mode = np.zeros(B, dtype=np.int32)
        if margins_out is not None:
            assert margins_out.dtype == np.float64 and margins_out.shape == (B,) and margins_out.flags.c_contiguous
        rc = self.lib.orc_pinv_solve_batch_m(
            C.byref(self.cdesc), C.byref(self.copts), B, _p(tt), _p(Q), _p(Xc),
            _p(Yc), _p(dq), _p(dx), mode.ctypes.data_as(C.POINTER(C.c_int32)),
            int(nthreads), _p(margins_out))
        if rc != 0:
            raise RuntimeError("C oracle failed (%d)" % rc)
        return dq, (dx[:, :self.desc.n_x] if self.desc.n_x else None), mode

    def task_eval(self, ti, t, z, y=None):
        n = self.desc.n_state
        m = self.desc.tasks[ti]["m"]
        z = np.ascontiguousarray(z, dtype=np.float64)
        yc = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
        tt = np.ascontiguousarray(self.desc.time_terms(t))
        e = np.zeros(m)
        J = np.zeros((m, n))
        Jt = np.zeros(m)
        self.lib.orc_task_eval(C.byref(self.cdesc), ti, _p(tt), _p(z), _p(yc),
                               _p(e), _p(J), _p(Jt))
        return e, J, Jt


def qp_data_batch(spec, t, Q, X=None, Y=None, mu=0.001, state_weights=None,
                  slack_weights=None):
    lib = load()
    desc = lower_skill(spec)
    cdesc = _capi.desc_to_c(desc)
    n = desc.n_state
    sw = np.ones(n) if state_weights is None else np.asarray(state_weights, float)
    if slack_weights is None:
        kw = []
        for tsk in desc.tasks:
            if tsk["soft"]:
                kw += [tsk["slack_weight"]] * tsk["m"]
    else:
        kw = slack_weights
    copts = _capi.qp_opts_to_c(mu, sw, kw)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    B = Q.shape[0]
    Xc = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
    Yc = None if Y is None else np.ascontiguousarray(Y, dtype=np.float64)
    tt = np.ascontiguousarray(desc.time_terms(t))
    nc = sum(tsk["m"] for tsk in desc.tasks)
    nv = n + desc.n_slack
    Hd = np.zeros((B, nv))
    A = np.zeros((B, nc, nv))
    lb = np.zeros((B, nc))
    ub = np.zeros((B, nc))
    lib.orc_qp_data_batch(C.byref(cdesc), C.byref(copts), B, _p(tt), _p(Q),
                          _p(Xc), _p(Yc), _p(Hd), _p(A), _p(lb), _p(ub))
    return Hd, A, lb, ub


class CQpOracle(object):
    """Literal ReactiveQPController.solve on the CPU (reactive_qp.py:461-528): H, A, lbA, ubA of
    orc_qp_data_batch handed to a dense Goldfarb-Idnani in C (the same method as
    clik_oracle.qp_solve_dense), OpenMP over instances.  Default weights only (what BASELINE config 4 uses)."""

    def __init__(self, spec, mu=0.001, baseline=None):
        self.lib = load()
        if baseline is not None:
            from . import baseline_desc
            self.cdesc, _ = baseline_desc.baseline_descriptor(*baseline)
            self.desc = _FlatDesc(self.cdesc)
        else:
            self.desc = lower_skill(spec)
            if self.desc.extern_code:
                raise NotImplementedError("the C oracle has no rows for constraints outside the row table")
            self.cdesc = _capi.desc_to_c(self.desc)
        kw = []
        for tsk in self.desc.tasks:
            if tsk["soft"]:
                kw += [tsk["slack_weight"]] * tsk["m"]
        self.copts = _capi.qp_opts_to_c(mu, np.ones(self.desc.n_state), kw)
        self.nv = self.desc.n_state + self.desc.n_slack

    def solve_batch(self, t, Q, X=None, Y=None, nthreads=0):
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        B = Q.shape[0]
        Xc = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
        Yc = None if Y is None else np.ascontiguousarray(Y, dtype=np.float64)
        tt = np.ascontiguousarray(self.desc.time_terms(t))
        xs = np.zeros((B, self.nv))
        status = np.zeros(B, dtype=np.int32)
        rc = self.lib.orc_qp_solve_batch(C.byref(self.cdesc), C.byref(self.copts), B, _p(tt), _p(Q), _p(Xc),
                                         _p(Yc), _p(xs), status.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads))
        if rc != 0:
            raise RuntimeError("C QP oracle failed (%d)" % rc)
        nq, nx = self.desc.n_q, self.desc.n_x
        return (xs[:, :nq], xs[:, nq:nq + nx] if nx else None,
                xs[:, nq + nx:] if self.desc.n_slack else None, status)


def num_threads():
    return load().orc_num_threads()
