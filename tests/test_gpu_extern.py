"""GPU parity for constraints outside the affine-in-features row table: they run as device code
generated from the expression graph (casclik_amd/codegen.py, CLIK_OUT_EXTERN) inside the run-time
instantiated kernels.  Reference: the skills of double_pendulum_2D_comparison_of_controllers.ipynb
(explicit sin/cos kinematics, cells 3-16 and 31-36) through both controllers, and a 7-DoF skill with
products / functions of tool-frame entries, a virtual variable, input and time terms."""
import os

import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs
from extern_skills import double_pendulum_skill, mixed_frame_skill, dual_quaternion_skill
from tolerances import PINV_RTOL, QP_RTOL, pinv_close, qp_close

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


def _pendulum_states(B, seed):
    # the notebook's operating range (cell 15) widened so that both table constraints get active
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-0.3, np.pi + 0.3, B), rng.uniform(-2.0, 2.0, B)], axis=1)


@pytest.mark.parametrize("track", [False, True])
def test_double_pendulum_reactive_qp(track):
    """ReactiveQPController on the notebook's point / tracking skill (cells 14, 35)."""
    from oracle import clik_oracle
    spec = double_pendulum_skill(track)
    ctrl = cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0, 1.0])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    assert ctrl.kernel_name.startswith("jit_")
    Q = _pendulum_states(512, 3)
    t = 1.3
    dq, _, slack, status = ctrl.solve_batch(t, Q)
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(spec, t, Q, weights=clik_oracle.qp_weights(spec, [1.0, 1.0]))
    assert np.array_equal(status, rst)
    ok = rst == 0
    assert ok.sum() > 400
    assert qp_close(dq, rdq, rows=ok)
    assert qp_close(slack, rsl, rows=ok)
    assert np.abs(dq[ok]).max() <= 0.5 + 1e-9            # the hard speed limit of cell 9
    # the single-instance call of the notebook loop (cell 16)
    one = ctrl.solve(t, Q[7])
    assert np.abs(one[0].toarray()[:, 0] - rdq[7]).max() < 1e-8


def test_double_pendulum_notebook_loop_matches_the_rollout():
    """Cell 16: 800 ticks of solve -> Euler from q0 = [pi/2 - 1e-5, 0]; here 200 ticks, on-device rollout
    against the host loop over the oracle."""
    from oracle import clik_oracle
    spec = double_pendulum_skill(True)
    ctrl = cc.ReactiveQPController(skill_spec=spec, robot_var_weights=[1.0, 1.0])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    dt, n_ticks = 0.01, 200
    q0 = np.array([[np.pi / 2 - 1e-5, 0.0], [1.0, 0.5], [2.0, -0.7]])
    ts = dt * np.arange(n_ticks)
    qf, dq_last, _, status = ctrl.rollout_batch(ts, q0, dt=dt)
    w = clik_oracle.qp_weights(spec, [1.0, 1.0])
    q = q0.copy()
    for k in range(n_ticks):
        rdq, _, _, rst = clik_oracle.qp_solve_batch(spec, ts[k], q, weights=w)
        assert (rst == 0).all()
        q = q + rdq * dt
    assert (status == 0).all()
    assert np.abs(qf - q).max() < 1e-7
    assert np.abs(dq_last - rdq).max() < 1e-7


def test_double_pendulum_pseudo_inverse():
    """The same constraints through PseudoInverseController: two 1-D sets = four modes
    (pseudo_inverse.py:107-130), VelocitySetConstraint ignored."""
    from oracle import clik_oracle
    spec = double_pendulum_skill(False)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    assert ctrl.kernel_name.startswith("jit_")
    Q = _pendulum_states(2048, 5)
    dq, _, mode = ctrl.solve_batch(0.0, Q)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q)
    # near the kinematic singularity (q1 ~ 0, pi) sigma_min^2 approaches the damping and the two evaluations
    # of the damped inverse differ by cond * eps: compare where the 2x2 Jacobian is not degenerate
    sane = np.abs(np.sin(Q[:, 1])) > 1e-2
    assert np.array_equal(mode[sane], rmode[sane])
    assert len(np.unique(mode)) >= 2
    assert pinv_close(dq, ref, rows=sane), _rel(dq[sane], ref[sane]).max()


def test_tool_frame_products_virtual_input_and_time(iiwa_fk):
    from oracle import clik_oracle
    spec = mixed_frame_skill(iiwa_fk)
    rng = np.random.default_rng(2)
    Q, _ = skills.synthetic_inputs(iiwa_fk, 1024, seed=4)
    X = rng.uniform(-1.0, 1.0, size=(1024, 1))
    Y = rng.uniform(-1.0, 1.0, size=(1024, 3))
    t = 0.8
    opts = {"multidim_sets": False}
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
    ctrl.setup_problem_functions()
    assert ctrl.kernel_name.startswith("jit_")
    dq, dx, mode = ctrl.solve_batch(t, Q, virtual_var=X, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, t, Q, X=X, Y=Y)
    assert np.array_equal(mode, rmode)
    assert len(np.unique(mode)) >= 2
    got = np.hstack([dq, dx])
    assert pinv_close(got, ref), _rel(got, ref).max()
    # and through the QP controller (the sphere set soft, the rest as declared)
    qc = cc.ReactiveQPController(skill_spec=spec)
    qc.setup_problem_functions()
    qc.setup_solver()
    assert qc.kernel_name.startswith("jit_")
    qdq, qdx, qsl, st = qc.solve_batch(t, Q, virtual_var=X, input_var=Y)
    rdq, rdx, rsl, rst = clik_oracle.qp_solve_batch(spec, t, Q, X=X, Y=Y)
    assert np.array_equal(st, rst)
    ok = rst == 0
    assert ok.sum() > 900
    assert _rel(np.hstack([qdq, qdx])[ok], np.hstack([rdq, rdx])[ok]).max() < 1e-7


def test_initial_problem_of_a_skill_with_generated_rows_and_a_virtual_variable(iiwa_fk):
    """solve_initial_problem (reactive_qp.py:300-459) for a skill whose rows are non-affine in the virtual variable
    (exp(-x), cos(x + 0.3 t): generated code, no clik_qp_data_batch): the reduced QP is assembled from the
    expression graph at the initial state and solved on the device - equal to the oracle's literal restatement."""
    from oracle import clik_oracle
    spec = mixed_frame_skill(iiwa_fk)
    rng = np.random.default_rng(7)
    Q, _ = skills.synthetic_inputs(iiwa_fk, 6, seed=9)
    qc = cc.ReactiveQPController(skill_spec=spec)
    qc.setup_problem_functions()
    qc.setup_solver()
    qc.setup_initial_problem_solver()
    with pytest.raises(NotImplementedError):
        qc.qp_data_batch(0.4, Q[:1], virtual_var=np.zeros((1, 1)), input_var=np.zeros((1, 3)))
    for b in range(len(Q)):
        x0, y0 = rng.uniform(-1.0, 1.0, 1), rng.uniform(-1.0, 1.0, 3)
        dq0 = None if b % 2 == 0 else rng.uniform(-0.3, 0.3, 7)
        virt, slack = qc.solve_initial_problem(0.4, Q[b], virtual_var0=x0, robot_vel_var0=dq0, input_var0=y0)
        rvirt, rslack = clik_oracle.qp_initial_problem(spec, 0.4, Q[b], x0=x0, dq0=dq0, y0=y0)
        assert np.abs(np.asarray(virt.toarray()).reshape(-1) - rvirt).max() < 1e-8 * (1 + np.abs(rvirt).max())
        assert np.abs(np.asarray(slack.toarray()).reshape(-1) - rslack).max() < 1e-8 * (1 + np.abs(rslack).max())


def test_generated_constraints_need_the_instantiated_kernel(monkeypatch):
    """No silent path: with the run-time instantiation disabled the controller refuses the skill, and
    the C ABI refuses to solve it with a built-in kernel."""
    spec = double_pendulum_skill(False)
    monkeypatch.setenv("CLIK_JIT", "0")
    with pytest.raises(NotImplementedError, match="generated device code"):
        cc.PseudoInverseController(skill_spec=spec).setup_problem_functions()
    with pytest.raises(NotImplementedError, match="generated device code"):
        cc.ReactiveQPController(skill_spec=spec).setup_problem_functions()
    monkeypatch.delenv("CLIK_JIT")
    import ctypes as C
    import torch
    from casclik_amd import _capi
    from casclik_amd.lowering import lower_skill
    lib = _capi.load_library()
    d = lower_skill(spec)
    cdesc = _capi.desc_to_c(d)
    copts = _capi.pinv_opts_to_c(cc.PseudoInverseController(skill_spec=spec).options)
    h = C.c_void_p()
    assert lib.clik_pinv_create(C.byref(cdesc), C.byref(copts), C.byref(h)) == 0
    q = torch.zeros(64, 2, dtype=torch.float64, device="cuda")
    dq = torch.zeros_like(q)
    mode = torch.zeros(64, dtype=torch.int32, device="cuda")
    rc = lib.clik_pinv_solve_batch(h, 64, None, q.data_ptr(), None, None, dq.data_ptr(), None, mode.data_ptr(), None)
    assert rc == -2 and b"code-generated" in lib.clik_last_error()
    lib.clik_pinv_destroy(h)


@pytest.mark.parametrize("which", ["Q_dist1", "Q_dist2", "cart_dist", "quat_dist"])
def test_dual_quaternion_pose_error_reactive_qp(ur5_fk, which):
    """ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 16-18, 24: the dual-quaternion pose
    errors (8 rows built from Q_fk(q), generated code) with joint limits and the speed limit through
    ReactiveQPController, and the notebook's loop (cell 39: solve -> clamp -> Euler) as a rollout."""
    from oracle import clik_oracle
    spec = dual_quaternion_skill(ur5_fk, which)
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    assert ctrl.kernel_name.startswith("jit_")
    rng = np.random.default_rng(8)
    home = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])
    Q = home + rng.uniform(-1.0, 1.0, size=(512, 6))
    dq, _, slack, status = ctrl.solve_batch(0.0, Q)
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(spec, 0.0, Q)
    assert np.array_equal(status, rst) and (rst == 0).all()
    assert qp_close(dq, rdq) and qp_close(slack, rsl)
    assert np.abs(dq).max() <= np.pi / 5 + 1e-9
    # 100 ticks of the simulation loop from UR5_home
    dt, n_ticks, vmax = 0.01, 100, np.pi / 5
    ts = dt * np.arange(n_ticks)
    q0 = np.stack([home, home + 0.2])
    qf, dq_last, _, st = ctrl.rollout_batch(ts, q0, dt=dt, max_speed=vmax)
    q = q0.copy()
    for k in range(n_ticks):
        rdq, _, _, rst = clik_oracle.qp_solve_batch(spec, ts[k], q)
        q = q + np.clip(rdq, -vmax, vmax) * dt
    assert (st == 0).all() and np.abs(qf - q).max() < 1e-7


@pytest.mark.parametrize("which", ["Q_dist2", "quat_dist"])
def test_dual_quaternion_pose_error_pseudo_inverse(ur5_fk, which):
    """The comparison notebook's pinv skill (cells 37-38): six 1-D joint-limit sets (64 modes) in front of
    the 8-row dual-quaternion error, which is the first EqualityConstraint and TALL (8 rows, 6 joints): its
    double processing (pseudo_inverse.py:317-326, :382-396) runs in the Gram form."""
    from oracle import clik_oracle
    spec = dual_quaternion_skill(ur5_fk, which, for_pinv=True)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    assert ctrl.kernel_name.startswith("jit_")
    rng = np.random.default_rng(9)
    home = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])
    Q = home + rng.uniform(-1.0, 1.0, size=(1024, 6))
    Q[::7, 2] = rng.choice([-1.0, 1.0], size=Q[::7].shape[0]) * rng.uniform(3.0, 3.4, size=Q[::7].shape[0])  # elbow limit
    dq, _, mode = ctrl.solve_batch(0.0, Q)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q)
    assert np.array_equal(mode, rmode)
    assert len(np.unique(mode)) >= 2
    assert pinv_close(dq, ref), _rel(dq, ref).max()


def test_notebook_golden_vectors(ur5_fk):
    """Committed fixtures (tests/golden/notebook_golden.npz, oracle-of-record, see make_golden.py): the
    notebooks' skills on the device without importing the oracle."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_golden.npz"))
    ctrl = cc.ReactiveQPController(skill_spec=double_pendulum_skill(True), robot_var_weights=[1.0, 1.0])
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    dq, _, slack, status = ctrl.solve_batch(1.3, g["pendulum_Q"])
    assert np.array_equal(status, g["pendulum_qp_status"])
    ok = status == 0
    assert qp_close(dq[ok], g["pendulum_qp_dq"][ok]) and qp_close(slack[ok], g["pendulum_qp_slack"][ok])
    qc = cc.ReactiveQPController(skill_spec=dual_quaternion_skill(ur5_fk, "Q_dist2"))
    qc.setup_problem_functions()
    qc.setup_solver()
    dq, _, slack, status = qc.solve_batch(0.0, g["ur5_Q"])
    assert (status == 0).all()
    assert qp_close(dq, g["dq_qp_dq"]) and qp_close(slack, g["dq_qp_slack"])
    pc = cc.PseudoInverseController(skill_spec=dual_quaternion_skill(ur5_fk, "Q_dist2", for_pinv=True))
    pc.setup_problem_functions()
    dq, _, mode = pc.solve_batch(0.0, g["ur5_Q"])
    assert np.array_equal(mode, g["dq_pinv_mode"]) and pinv_close(dq, g["dq_pinv_dq"])


def test_failed_instantiation_is_loud_and_falls_back_only_where_a_builtin_kernel_can_serve(ur5_fk, monkeypatch):
    """A broken hipcc invocation at setup: a row-table skill runs on the built-in dynamic kernel (with a
    warning) and still matches the oracle; a skill with generated constraints has no other kernel and fails."""
    from casclik_amd import jit
    if jit._hipcc() is None:
        pytest.skip("no compiler on this box: nothing to break")
    from oracle import clik_oracle
    monkeypatch.setenv("CLIK_JIT_DEFINES", "-fthis-flag-does-not-exist-%d" % os.getpid())
    monkeypatch.delenv("CLIK_JIT_RECORD", raising=False)     # (a request that cannot compile is not one to replay at build())
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    T = ur5_fk["T_fk"](q)
    spec = cc.SkillSpecification("plain", t, q, constraints=[
        cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.35, 0.15, 0.45]), gain=1.7, priority=1),
        cc.SetConstraint("wrist", q[4], set_min=-0.3, set_max=0.3, priority=0)])
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    with pytest.warns(UserWarning, match="instantiation failed"):
        ctrl.setup_problem_functions()
    assert ctrl.kernel_name == "dynamic"
    Q, _ = skills.synthetic_inputs(ur5_fk, 100, seed=3)
    dq, _, mode = ctrl.solve_batch(0.0, 0.3 * Q)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, 0.3 * Q)
    assert np.array_equal(mode, rmode) and pinv_close(dq, ref)
    with pytest.warns(UserWarning, match="instantiation failed"):
        with pytest.raises(NotImplementedError, match="generated device code"):
            cc.ReactiveQPController(skill_spec=double_pendulum_skill(False)).setup_problem_functions()


def test_generated_constraints_in_every_constraint_class(iiwa_fk):
    """Generated rows as a multidimensional SetConstraint (activation matrix S, pseudo_inverse.py:289-298),
    as the converging final set (:337-379), as a VelocityEqualityConstraint and as a VelocitySetConstraint
    (QP only: the pinv controller skips it), next to table rows."""
    from oracle import clik_oracle
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 7)
    T = iiwa_fk["T_fk"](q)
    p = T[:3, 3]
    reach = cc.SetConstraint("reach", cs.vertcat(cs.dot(p, p), cs.sin(q[0]) * q[1]), set_min=np.array([0.15, -0.2]),
                             set_max=np.array([0.45, 0.2]), gain=1.5, priority=0)
    swirl = cc.VelocityEqualityConstraint("swirl", cs.cos(q[2]) * q[3] + 0.1 * cs.sin(t), target=0.05, priority=1)
    pos = cc.EqualityConstraint("pos", p - np.array([0.4, 0.1, 0.5]), gain=2.0, priority=2, constraint_type="soft")
    rest = cc.EqualityConstraint("rest", q - 0.2, gain=0.2, priority=3, constraint_type="soft")
    rate = cc.VelocitySetConstraint("rate", cs.vertcat(p[0] * p[1], q[5] * q[6]), set_min=-0.1 * np.ones(2),
                                    set_max=0.1 * np.ones(2), priority=4)
    Q, _ = skills.synthetic_inputs(iiwa_fk, 600, seed=15)
    Q = 0.6 * Q
    for cons, opts in (([reach, swirl, pos, rest, rate], {"multidim_sets": True}),
                       ([swirl, pos, rest, reach], {"multidim_sets": True, "converge_final_set_to_max": True})):
        reach.priority = 0 if cons[0] is reach else 9
        spec = cc.SkillSpecification("classes", t, q, constraints=cons)
        ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
        ctrl.setup_problem_functions()
        assert ctrl.kernel_name.startswith("jit_")
        dq, _, mode = ctrl.solve_batch(0.6, Q)
        ref, rmode = clik_oracle.pinv_solve_batch(spec, opts, 0.6, Q)
        assert np.array_equal(mode, rmode) and len(np.unique(mode)) >= 2
        assert pinv_close(dq, ref), _rel(dq, ref).max()
    reach.priority = 0
    reach.constraint_type = "soft"
    qspec = cc.SkillSpecification("classes_qp", t, q, constraints=[reach, swirl, pos, rest, rate])
    qc = cc.ReactiveQPController(skill_spec=qspec)
    qc.setup_problem_functions()
    qc.setup_solver()
    assert qc.kernel_name.startswith("jit_")
    qdq, _, qsl, st = qc.solve_batch(0.6, Q)
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(qspec, 0.6, Q)
    assert np.array_equal(st, rst)
    ok = rst == 0
    assert ok.sum() > 300 and _rel(qdq[ok], rdq[ok]).max() < 1e-7
