"""GPU: integration schemes of the on-device rollout (casclik/integration_methods.py:11-23)
and re-entrancy of the rollout calls across streams (include/clik.h: handles are immutable)."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs

pytestmark = pytest.mark.gpu


def _tracking(fk, n):
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", n)
    p = fk["T_fk"](q)[:3, 3]
    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2, 0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    spec = cc.SkillSpecification("track", t, q, constraints=[
        cc.EqualityConstraint("move_point", p - path, gain=0.5, constraint_type="soft")])
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    return spec, ctrl


def _host_rk4(solve, times, Q, dt, vmax):
    """the scheme of integration_methods.py:17-23 with the controller as dx_function (each stage clamped)"""
    q = Q.copy()
    for tv in times:
        def f(tt, qq):
            d = solve(float(tt), qq)
            return np.clip(d, -vmax, vmax) if vmax > 0 else d
        k1 = f(tv, q)
        k2 = f(tv + dt / 2, q + dt / 2 * k1)
        k3 = f(tv + dt / 2, q + dt / 2 * k2)
        k4 = f(tv + dt, q + dt * k3)
        v = (k1 + 2 * k2 + 2 * k3 + k4) / 6.0
        q = q + dt * v
    return q, v


def test_rk4_rollout_matches_host_rk4_over_the_kernel_and_over_the_oracle(ur5_fk):
    from oracle import clik_oracle
    spec, ctrl = _tracking(ur5_fk, 6)
    rng = np.random.default_rng(8)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.1, size=(70, 6))
    dt, n_ticks, vmax = 0.05, 6, 0.4
    times = 3.0 + dt * np.arange(n_ticks)
    q_dev, dq_dev, _ = ctrl.rollout_batch(times, Q, dt=dt, max_speed=vmax, method="rk4")
    qh, vh = _host_rk4(lambda t, q: ctrl.solve_batch(t, q)[0], times, Q, dt, vmax)
    assert np.abs(q_dev - qh).max() < 1e-10 and np.abs(dq_dev - vh).max() < 1e-9
    qo, vo = _host_rk4(lambda t, q: clik_oracle.pinv_solve_batch(spec, None, t, q)[0], times, Q, dt, vmax)
    assert np.abs(q_dev - qo).max() < 1e-9 and np.abs(dq_dev - vo).max() < 1e-8
    # fourth order against first order: RK4 with the step doubled still beats Euler on the same horizon
    q_e, _, _ = ctrl.rollout_batch(times, Q, dt=dt, max_speed=vmax, method="euler")
    assert np.abs(q_e - q_dev).max() > 1e-7          # the schemes differ ...
    fine = 3.0 + (dt / 8) * np.arange(8 * n_ticks)
    q_f, _, _ = ctrl.rollout_batch(fine, Q, dt=dt / 8, max_speed=vmax, method="rk4")
    assert np.abs(q_dev - q_f).max() < 0.05 * np.abs(q_e - q_f).max()      # ... and RK4 is the closer one


def test_rk4_rollout_of_the_stack_with_sets(iiwa_fk):
    """config-3 stack (mode switches inside the stages), input targets, virtual-variable-free"""
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    Q, Y = skills.synthetic_inputs(iiwa_fk, 90, seed=5, distribution="mixed")
    dt, vmax, n = 0.008, np.pi / 5, 5
    q_dev, dq_dev, mode_dev = ctrl.rollout_batch(np.zeros(n), Q, input_var=Y, dt=dt, max_speed=vmax, method="rk4")
    modes = []

    def solve(t, q):
        d, _, m = ctrl.solve_batch(t, q, input_var=Y)
        modes.append(m)
        return d
    qh, vh = _host_rk4(solve, np.zeros(n), Q, dt, vmax)
    assert np.abs(q_dev - qh).max() < 1e-9 and np.abs(dq_dev - vh).max() < 1e-7
    assert np.array_equal(mode_dev, modes[-4])        # the mode of the last tick's first stage


@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_rollout_of_the_stack_beyond_the_team_kernels_range(iiwa_fk, method):
    """the same stack at 16500 instances: the one-lane value-specialised rollout (state and Runge-Kutta bookkeeping
    in registers, no LDS) against the host loop over the per-tick kernel of that batch size"""
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    B = 16500
    assert ctrl.kernel_variant(B).endswith("/lanev")
    Q, Y = skills.synthetic_inputs(iiwa_fk, B, seed=6, distribution="mixed")
    dt, vmax, n = 0.008, np.pi / 5, 3
    q_dev, dq_dev, mode_dev = ctrl.rollout_batch(np.zeros(n), Q, input_var=Y, dt=dt, max_speed=vmax, method=method)
    modes = []

    def solve(t, q):
        d, _, m = ctrl.solve_batch(t, q, input_var=Y)
        modes.append(m)
        return d
    if method == "rk4":
        qh, vh = _host_rk4(solve, np.zeros(n), Q, dt, vmax)
        first_stage_of_last_tick = modes[-4]
    else:
        qh = Q.copy()
        for _ in range(n):
            vh = np.clip(solve(0.0, qh), -vmax, vmax)
            qh = qh + dt * vh
        first_stage_of_last_tick = modes[-1]
    assert np.abs(q_dev - qh).max() < 1e-9 and np.abs(dq_dev - vh).max() < 1e-7
    assert np.array_equal(mode_dev, first_stage_of_last_tick)
    assert len(np.unique(mode_dev)) == 2


def test_qp_rk4_rollout_matches_host_rk4(ur5_fk):
    """the QP controller's rollout with method="rk4" (four QP solves per tick in one launch, hot-started from stage
    to stage) against the scheme of integration_methods.py:17-23 looped on the host over the same kernel and over
    the oracle"""
    from oracle import clik_oracle
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = ur5_fk["T_fk"](q)[:3, 3]
    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2, 0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    vmax = 0.6
    spec = cc.SkillSpecification("track_qp", t, q, constraints=[
        cc.EqualityConstraint("move_point", p - path, gain=0.5, constraint_type="soft"),
        cc.VelocitySetConstraint("speed", q, set_min=-vmax * np.ones(6), set_max=vmax * np.ones(6))])
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    rng = np.random.default_rng(21)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.1, size=(70, 6))
    dt, times = 0.05, 1.0 + 0.05 * np.arange(6)
    qf, dql, sl, status = ctrl.rollout_batch(times, Q, dt=dt, max_speed=0.5, method="rk4")
    assert (status == 0).all()
    qh, vh = _host_rk4(lambda tt, qq: ctrl.solve_batch(tt, qq)[0], times, Q, dt, 0.5)
    assert np.allclose(qf, qh, rtol=1e-9, atol=1e-11) and np.allclose(dql, vh, rtol=1e-7, atol=1e-9)
    qo, vo = _host_rk4(lambda tt, qq: clik_oracle.qp_solve_batch(spec, tt, qq)[0], times, Q[:9], dt, 0.5)
    assert np.allclose(qf[:9], qo, rtol=1e-7, atol=1e-9)
    # Euler through the same entry equals the Euler rollout
    qe, dqe, _, _ = ctrl.rollout_batch(times, Q, dt=dt, max_speed=0.5, method="euler")
    qe0, dqe0, _, _ = ctrl.rollout_batch(times, Q, dt=dt, max_speed=0.5)
    assert np.array_equal(qe, qe0) and np.array_equal(dqe, dqe0)
    assert np.abs(qe - qf).max() > 1e-7          # (and differs from Runge-Kutta)
    with pytest.raises(ValueError):
        ctrl.rollout_batch(times, Q, dt=dt, method="heun")


def test_rollouts_on_two_streams_do_not_share_state(ur5_fk):
    """two rollouts of ONE handle with different time stamps, enqueued on two streams before either is
    waited for: each must see its own time-slot records (they used to live in the handle)"""
    import torch
    spec, ctrl = _tracking(ur5_fk, 6)
    rng = np.random.default_rng(3)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = torch.from_numpy(home + rng.normal(scale=0.1, size=(4096, 6))).cuda()
    dt, n = 0.05, 40
    ta, tb = 1.0 + dt * np.arange(n), 9.0 + dt * np.arange(n)
    ref_a = ctrl.rollout_batch(ta, Q, dt=dt)[0].cpu().numpy()
    ref_b = ctrl.rollout_batch(tb, Q, dt=dt)[0].cpu().numpy()
    assert np.abs(ref_a - ref_b).max() > 1e-3
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(5):
        with torch.cuda.stream(sa):
            qa = ctrl.rollout_batch(ta, Q, dt=dt)[0]
        with torch.cuda.stream(sb):
            qb = ctrl.rollout_batch(tb, Q, dt=dt)[0]
        outs.append((qa, qb))
    torch.cuda.synchronize()
    for qa, qb in outs:
        assert np.array_equal(qa.cpu().numpy(), ref_a) and np.array_equal(qb.cpu().numpy(), ref_b)


@pytest.mark.parametrize("dynamic", [False, True])
def test_per_instance_time_stamps(ur5_fk, monkeypatch, dynamic):
    """time_var with one entry per instance (`t` of 1 or B values): one launch of the
    per-instance-time kernel (clik_pinv_solve_batch_t), or - a skill on the dynamic fallback kernel - one launch per
    distinct time stamp; both equal the per-time-stamp calls and the oracle evaluated at each instance's own time"""
    from oracle import clik_oracle as orc
    from tolerances import PINV_RTOL
    if dynamic:
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec, ctrl = _tracking(ur5_fk, 6)
    assert ("dynamic" in ctrl.kernel_name) == dynamic
    rng = np.random.default_rng(12)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.1, size=(97, 6))
    phases = np.array([0.0, 1.5, 4.0])
    times = phases[rng.integers(0, 3, size=97)]
    dq, _, mode = ctrl.solve_batch(times, Q)
    for tv in phases:
        rows = times == tv
        ref, _, rmode = ctrl.solve_batch(float(tv), Q[rows])
        assert np.allclose(dq[rows], ref, rtol=1e-12, atol=1e-14) and np.array_equal(mode[rows], rmode)
    # every instance at its own time
    times = rng.uniform(0.0, 30.0, size=97)
    dq, _, mode = ctrl.solve_batch(times, Q)
    for b in range(97):
        rdq, rmode = orc.pinv_solve_batch(spec, None, float(times[b]), Q[b:b + 1])
        assert mode[b] == rmode[0]
        assert np.allclose(dq[b], rdq[0], rtol=PINV_RTOL, atol=PINV_RTOL * max(1.0, np.abs(rdq).max()))
    with pytest.raises(ValueError):
        ctrl.solve_batch(times[:5], Q)


def test_per_instance_time_stamps_qp(ur5_fk):
    """the same through the QP controller (clik_qp_solve_batch_t)"""
    from oracle import clik_oracle as orc
    from tolerances import QP_RTOL
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = ur5_fk["T_fk"](q)[:3, 3]
    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2, 0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    spec = cc.SkillSpecification("track_qp", t, q, constraints=[
        cc.EqualityConstraint("move_point", p - path, gain=0.5, constraint_type="soft"),
        cc.VelocitySetConstraint("speed", q, set_min=-0.4 * np.ones(6), set_max=0.4 * np.ones(6))])
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    rng = np.random.default_rng(13)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.1, size=(70, 6))
    times = rng.uniform(0.0, 30.0, size=70)
    dq, _, sl, status = ctrl.solve_batch(times, Q)
    assert (status == 0).all()
    for b in range(0, 70, 3):
        rdq = orc.qp_solve_batch(spec, float(times[b]), Q[b:b + 1])[0]
        assert np.allclose(dq[b], rdq[0], rtol=QP_RTOL, atol=QP_RTOL * max(1.0, np.abs(rdq).max()))
    one = ctrl.solve_batch(float(times[7]), Q[7:8])[0]
    assert np.allclose(dq[7], one[0], rtol=1e-12, atol=1e-14)


def test_output_tensors_are_validated(iiwa_fk):
    """out / mode_out go to the kernels by pointer: a wrong shape, dtype, device or a strided view raises instead
    of writing out of bounds (solve_batch, bind_batch of both controllers)"""
    import torch
    ctrl = cc.PseudoInverseController(skill_spec=skills.stack_skill(iiwa_fk), options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    Q, Y = skills.synthetic_inputs(iiwa_fk, 10, seed=1)
    Qd, Yd = torch.from_numpy(Q).cuda(), torch.from_numpy(Y).cuda()
    good = torch.empty((10, 7), dtype=torch.float64, device="cuda")
    ctrl.solve_batch(0.0, Qd, input_var=Yd, out=good)
    bad = [torch.empty((10, 7), dtype=torch.float32, device="cuda"), torch.empty((9, 7), dtype=torch.float64, device="cuda"),
           torch.empty((10, 7), dtype=torch.float64), torch.empty((7, 10), dtype=torch.float64, device="cuda").T]
    for out in bad:
        with pytest.raises(ValueError):
            ctrl.solve_batch(0.0, Qd, input_var=Yd, out=out)
        with pytest.raises(ValueError):
            ctrl.bind_batch(Qd, input_var=Yd, out=out)
    with pytest.raises(ValueError):
        ctrl.bind_batch(Qd, input_var=Yd, mode_out=torch.empty((10,), dtype=torch.int64, device="cuda"))
    qp = cc.ReactiveQPController(skill_spec=skills.qp_skill(iiwa_fk))
    qp.setup_problem_functions()
    qp.setup_solver()
    with pytest.raises(ValueError):
        qp.bind_batch(Qd, input_var=Yd, out=bad[0])
