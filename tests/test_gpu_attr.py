"""GPU: constraint attributes given as EXPRESSIONS - gains, set bounds, velocity targets that depend on
(t, q, input) - which casclik accepts as MX (casclik/constraints.py:35-39, :90-92, :199-206) and multiplies /
subtracts inside its symbolic expressions (pseudo_inverse.py:301-318, reactive_qp.py:199-232).  On the device they
are code generated next to the skill's kernel (codegen.emit_attr -> ExternAttr<TI>), evaluated per instance and tick.
The reference's own code run on such skills is pinned in tests/golden/ref_pins.npz (iiwa_sym_attrs,
ur5_qp_sym_attrs: tests/test_refpins.py, tests/test_gpu_refpins.py); here the other kernel families and the rollout."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import sym as cs
from oracle import clik_oracle
from tolerances import PINV_RTOL, QP_RTOL

pytestmark = pytest.mark.gpu


def _stack(fk, n, y_gain=False):
    """config-3 structure (multidimensional joint-limit set, pose-like task, centering) with limits that breathe
    with time, a set gain that depends on the state and a task gain that depends on time, state (and an input)"""
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    y = cs.MX.sym("y", 1) if y_gain else None
    lo, hi = np.asarray(fk["lower"], float), np.asarray(fk["upper"], float)
    shrink = 0.85 + 0.1 * cs.sin(0.5 * t)
    p = fk["T_fk"](q)[:3, 3]
    target = cs.vertcat(0.4 + 0.05 * cs.sin(t), 0.1, 0.5)
    kp = 2.0 + cs.cos(0.3 * t) + 0.4 * q[2] * q[2]
    if y_gain:
        kp = kp + y[0]
    cns = [cc.SetConstraint("joint_limits", q, set_min=cs.vertcat(*[float(v) * shrink for v in lo]),
                            set_max=cs.vertcat(*[float(v) * shrink for v in hi]), gain=1.0 + 0.2 * q[0] * q[0],
                            priority=0),
           cc.EqualityConstraint("tool_position", p - target, gain=kp, constraint_type="soft", priority=1),
           cc.EqualityConstraint("centering", q - 0.5 * (lo + hi), gain=0.5, constraint_type="soft", priority=2)]
    kw = {"input_var": y} if y_gain else {}
    return cc.SkillSpecification("sym_stack", t, q, constraints=cns, **kw), lo, hi


def _states(lo, hi, B, seed):
    rng = np.random.default_rng(seed)
    r = hi - lo
    return rng.uniform(lo - 0.05 * r, hi + 0.05 * r, size=(B, len(lo)))


@pytest.mark.parametrize("y_gain", [False, True])
def test_pinv_stack_with_expression_attributes(iiwa_fk, y_gain):
    spec, lo, hi = _stack(iiwa_fk, 7, y_gain)
    opts = {"multidim_sets": True}
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    ctrl.setup_problem_functions()
    assert ctrl.kernel_name.startswith("jit_")
    Q = _states(lo, hi, 333, 5)
    Y = np.random.default_rng(6).uniform(0.0, 2.0, size=(333, 1)) if y_gain else None
    for tv in (0.0, 2.2):
        dq, _, mode = ctrl.solve_batch(tv, Q, input_var=Y)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, tv, Q, Y=Y)
        assert np.array_equal(mode, rmode)
        assert set(np.unique(mode)) == {0, 1}
        err = np.abs(dq - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))
        assert err.max() < PINV_RTOL, err.max()


def test_large_batch_and_rollout_with_expression_attributes(ur5_fk):
    """the one-wave kernel (batch beyond the small-batch variants) and the on-device rollout against the host loop"""
    spec, lo, hi = _stack(ur5_fk, 6)
    opts = {"multidim_sets": True}
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    ctrl.setup_problem_functions()
    Q = _states(lo, hi, 40000, 7)
    dq, _, mode = ctrl.solve_batch(1.3, Q)
    sub = np.arange(0, 40000, 97)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 1.3, Q[sub])
    assert np.array_equal(mode[sub], rmode)
    assert (np.abs(dq[sub] - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))).max() < PINV_RTOL
    # rollout: 12 ticks in one launch = 12 host ticks
    Q0 = _states(lo, hi, 130, 8)
    times = 0.5 + 0.01 * np.arange(12)
    qf, dql, model = ctrl.rollout_batch(times, Q0, dt=0.01, max_speed=1.5)
    qh = Q0.copy()
    for tv in times:
        d, _, m = ctrl.solve_batch(float(tv), qh)
        d = np.clip(d, -1.5, 1.5)
        qh = qh + 0.01 * d
    assert np.allclose(qf, qh, rtol=1e-10, atol=1e-12) and np.array_equal(model, m)


def test_qp_with_expression_attributes(iiwa_fk):
    spec, lo, hi = _stack(iiwa_fk, 7)
    t, q = spec.time_var, spec.robot_var
    vmax = np.asarray(iiwa_fk["velocity"], float)
    vm = cs.vertcat(*[float(v) * (0.5 + 0.1 * cs.cos(t)) for v in vmax])
    cns = list(spec.constraints) + [cc.VelocitySetConstraint("speed", q, set_min=-vm, set_max=vm, priority=0)]
    spec = cc.SkillSpecification("sym_stack_qp", t, q, constraints=cns)
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    r = hi - lo
    Q = np.random.default_rng(9).uniform(lo + 0.1 * r, hi - 0.1 * r, size=(257, 7))
    for tv in (0.0, 1.9):
        dq, _, slack, status = ctrl.solve_batch(tv, Q)
        rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(spec, tv, Q)
        assert np.array_equal(status, rstatus) and (status == 0).all()
        assert (np.abs(dq - rdq).max(axis=1) / (1.0 + np.abs(rdq).max(axis=1))).max() < QP_RTOL
        assert (np.abs(slack - rslack).max(axis=1) / (1.0 + np.abs(rslack).max(axis=1))).max() < QP_RTOL


def test_expression_attributes_need_the_instantiated_kernel(ur5_fk, monkeypatch):
    """the dynamic fallback kernels read gains and bounds from the skill image: a skill whose attributes are
    expressions is refused there, loudly"""
    monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec, lo, hi = _stack(ur5_fk, 6)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options={"multidim_sets": True})
    with pytest.raises(NotImplementedError):
        ctrl.setup_problem_functions()
        ctrl.solve_batch(0.0, _states(lo, hi, 4, 1))


def test_expression_attributes_with_one_time_per_instance(ur5_fk):
    """time-dependent gains / bounds and time_var with one entry per instance: the generated attribute code reads
    the instance's own time-slot record (clik_pinv_solve_batch_t / clik_qp_solve_batch_t)"""
    spec, lo, hi = _stack(ur5_fk, 6)
    opts = {"multidim_sets": True}
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=opts)
    ctrl.setup_problem_functions()
    Q = _states(lo, hi, 90, 12)
    times = np.random.default_rng(13).uniform(0.0, 6.0, size=90)
    dq, _, mode = ctrl.solve_batch(times, Q)
    for b in range(0, 90, 4):
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, float(times[b]), Q[b:b + 1])
        assert mode[b] == rmode[0]
        assert np.abs(dq[b] - ref[0]).max() < PINV_RTOL * (1.0 + np.abs(ref).max())
    qp = cc.ReactiveQPController(skill_spec=spec)
    qp.setup_problem_functions()
    qp.setup_solver()
    r = hi - lo
    Qi = np.random.default_rng(14).uniform(lo + 0.15 * r, hi - 0.15 * r, size=(90, 6))
    dq, _, slack, status = qp.solve_batch(times, Qi)
    assert (status == 0).all()
    for b in range(0, 90, 6):
        rdq = clik_oracle.qp_solve_batch(spec, float(times[b]), Qi[b:b + 1])[0]
        assert np.abs(dq[b] - rdq[0]).max() < QP_RTOL * (1.0 + np.abs(rdq).max())
