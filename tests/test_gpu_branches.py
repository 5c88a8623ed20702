"""GPU parity for the remaining branches of the reference's
get_problem_expressions (pseudo_inverse.py:274-443) and controller options
(:42-66), on skills that have no AOT shape (dynamic kernel)."""
import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs
from tolerances import PINV_RTOL, pinv_close, qp_close

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


def _check(spec, options, Q, Y=None, t=0.0, min_modes=1):
    from oracle import clik_oracle
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=None if options is None else dict(options))
    ctrl.setup_problem_functions()
    dq, _, mode = ctrl.solve_batch(t, Q, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, options, t, Q, Y=Y)
    assert np.array_equal(mode, rmode)
    assert len(np.unique(mode)) >= min_modes
    assert pinv_close(dq, ref), _rel(dq, ref).max()
    return ctrl


def _iiwa_syms(fk):
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 7)
    return t, q, fk["T_fk"](q)


def test_velocity_equality_first_and_not_first(iiwa_fk):
    """First VelocityEqualityConstraint is processed once (:327-335); a later one
    is projected (:430-443); the following EqualityConstraint is NOT 'first'."""
    t, q, T = _iiwa_syms(iiwa_fk)
    spin = cc.VelocityEqualityConstraint("spin", q[6], target=0.3, priority=0)
    lift = cc.VelocityEqualityConstraint("lift", T[2, 3], target=-0.05, priority=2)
    pos = cc.EqualityConstraint("xy", T[:2, 3] - np.array([0.3, 0.2]), gain=2.0, priority=1)
    spec = cc.SkillSpecification("vel", t, q, constraints=[spin, pos, lift])
    Q, _ = skills.synthetic_inputs(iiwa_fk, 150, seed=1)
    _check(spec, None, Q)


def test_converge_final_set_to_max(iiwa_fk):
    """converge_final_set_to_max (:337-379): the LAST constraint is a set; when
    active it is driven to set_max through the null space of the others."""
    t, q, T = _iiwa_syms(iiwa_fk)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=0)
    height = cc.SetConstraint("elbow", q[3], set_min=-1.0, set_max=1.0, gain=2.0, priority=5)
    spec = cc.SkillSpecification("conv", t, q, constraints=[pos, height])
    Q, _ = skills.synthetic_inputs(iiwa_fk, 200, seed=2, distribution="mixed")
    _check(spec, {"converge_final_set_to_max": True}, Q, min_modes=2)
    _check(spec, {"converge_final_set_to_max": False}, Q, min_modes=2)


def test_converge_final_multidim_set(iiwa_fk):
    t, q, T = _iiwa_syms(iiwa_fk)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=0)
    # (q[6] does not move the tool position: a set on it would make the cone test a rounding tie)
    box = cc.SetConstraint("wrist", q[3:6], set_min=-np.ones(3), set_max=np.ones(3), gain=1.5, priority=5)
    spec = cc.SkillSpecification("convm", t, q, constraints=[pos, box])
    Q, _ = skills.synthetic_inputs(iiwa_fk, 200, seed=3, distribution="mixed")
    _check(spec, {"converge_final_set_to_max": True, "multidim_sets": True}, Q, min_modes=2)


def test_standard_pinv_single_task(iiwa_fk):
    """pinv_method 'standard' (cs.pinv, :93-94) on a single full-rank task."""
    spec = skills.position_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 120, seed=4)
    c = _check(spec, {"pinv_method": "standard"}, Q, Y[:, :3])
    assert c.kernel_name.startswith("jit_")      # no AOT shape: instantiated at setup


def test_jit_and_dynamic_kernels_agree(iiwa_fk, monkeypatch):
    """A skill without an AOT shape gets a kernel instantiated at setup
    (casclik_amd/jit.py); with CLIK_JIT=0 the dynamic kernel serves it.  Same answers."""
    t, q, T = _iiwa_syms(iiwa_fk)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=0)
    elbow = cc.SetConstraint("elbow", q[3], set_min=-1.0, set_max=1.0, gain=2.0, priority=5)
    spec = cc.SkillSpecification("conv", t, q, constraints=[pos, elbow])
    opts = {"converge_final_set_to_max": True}
    Q, _ = skills.synthetic_inputs(iiwa_fk, 300, seed=12, distribution="mixed")
    fast = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
    fast.setup_problem_functions()
    monkeypatch.setenv("CLIK_JIT", "0")
    slow = cc.PseudoInverseController(skill_spec=spec, options=dict(opts))
    slow.setup_problem_functions()
    assert fast.kernel_name.startswith("jit_") and slow.kernel_name == "dynamic"
    a, _, ma = fast.solve_batch(0.0, Q)
    b, _, mb = slow.solve_batch(0.0, Q)
    assert np.array_equal(ma, mb) and pinv_close(a, b)


def test_feedforward_off_and_tiny_damping(ur5_fk):
    """feedforward False drops d e/d t (:320-321); the dual-quaternion notebook
    sets damping_factor 1e-26 (ur5_dual_quaternion_vs_transformation_matrix.ipynb:556)."""
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    path = cs.vertcat(0.3 * cs.sin(0.5 * t), 0.2 + 0.1 * t, 0.4 * cs.cos(0.5 * t))
    spec = cc.SkillSpecification("track", t, q, constraints=[cc.EqualityConstraint("p", p - path, gain=0.5)])
    Q, _ = skills.synthetic_inputs(fk, 100, seed=5)
    for opts in ({"feedforward": False}, {"feedforward": True}, {"damping_factor": 1e-26}):
        _check(spec, opts, Q, t=1.3)


def test_matrix_gain_and_input_offset(ur5_fk):
    """m x m gain matrices (constraints.py:46-49) and an additive input_var, the
    expression of ur5_input_experiment.ipynb cell 11: T_fk(q)[:3,3] - T_des[:3,3] + y."""
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    y = cs.MX.sym("y", 3)
    p = fk["T_fk"](q)[:3, 3]
    K = np.array([[2.0, 0.1, 0.0], [0.1, 3.0, 0.2], [0.0, 0.2, 1.0]])
    c = cc.EqualityConstraint("disturbed", p - np.array([0.4, 0.2, 0.3]) + y, gain=K, constraint_type="soft",
                              priority=100)
    spec = cc.SkillSpecification("inp", t, q, input_var=y, constraints=[c])
    Q, _ = skills.synthetic_inputs(fk, 100, seed=6)
    Y = np.random.default_rng(6).normal(scale=0.1, size=(100, 3))
    _check(spec, None, Q, Y)


def test_two_multidim_sets_four_modes(iiwa_fk):
    """Two multidimensional sets -> modes 00,10,01,11 in the reference order."""
    t, q, T = _iiwa_syms(iiwa_fk)
    lo, hi = np.array(iiwa_fk["lower"]), np.array(iiwa_fk["upper"])
    arm = cc.SetConstraint("arm", q[:3], set_min=0.5 * lo[:3], set_max=0.5 * hi[:3], priority=0)
    wrist = cc.SetConstraint("wrist", q[3:6], set_min=0.5 * lo[3:6], set_max=0.5 * hi[3:6], priority=1)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=2)
    spec = cc.SkillSpecification("two", t, q, constraints=[pos, wrist, arm])
    rng = np.random.default_rng(7)
    Q = rng.uniform(0.6 * lo, 0.6 * hi, size=(256, 7))
    ctrl = _check(spec, {"multidim_sets": True}, Q, min_modes=3)
    assert ctrl.n_modes == 4


def test_velocity_set_is_ignored_by_pinv(iiwa_fk):
    """VelocitySetConstraint has no branch in the pinv controller (:274-443)."""
    t, q, T = _iiwa_syms(iiwa_fk)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=1)
    speed = cc.VelocitySetConstraint("speed", q, set_min=-0.1 * np.ones(7), set_max=0.1 * np.ones(7), priority=0)
    with_speed = cc.SkillSpecification("a", t, q, constraints=[pos, speed])
    Q, _ = skills.synthetic_inputs(iiwa_fk, 64, seed=8)
    _check(with_speed, None, Q)


def test_multidim_set_without_option_is_refused(iiwa_fk):
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec)          # multidim_sets defaults to False
    with pytest.raises(NotImplementedError, match="multidim_sets"):
        ctrl.setup_problem_functions()


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_three_sets_eight_modes(ur5_fk, kernel, monkeypatch):
    """Three 1-D SetConstraints between two equality tasks: 8 modes scanned in the
    reference's order (pseudo_inverse.py:107-130), by the shape-specialised kernel
    (run-time instantiated, 8 mode bodies) and by the dynamic-shape kernel."""
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.EqualityConstraint("pos", p - np.array([0.35, 0.2, 0.45]), gain=4.0, priority=1)]
    for k, i in enumerate((0, 1, 3)):
        cons.append(cc.SetConstraint("lim_%d" % i, q[i], set_min=0.25 * lo[i], set_max=0.25 * hi[i], gain=1.0 + k,
                                     priority=2 + k))
    cons.append(cc.EqualityConstraint("posture", q[2] - 0.3, gain=0.7, priority=9))
    spec = cc.SkillSpecification("three_sets", t, q, constraints=cons)
    rng = np.random.default_rng(17)
    Q = rng.uniform(0.4 * lo, 0.4 * hi, size=(300, 6))       # all 8 modes occur (oracle: 168/19/46/37/8/4/13/5)
    ctrl = _check(spec, None, Q, min_modes=8)
    assert ctrl.n_modes == 8
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")


def _path_following_skill(fk):
    """Path following with a virtual variable (cart_on_track_1D notebook
    cells 56,75): the tool follows the line p0 + d*s, the path parameter s is a virtual
    state advanced by a VelocityEqualityConstraint and kept in [0, 1] by a SetConstraint."""
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    s = cs.MX.sym("s", 1)
    p = fk["T_fk"](q)[:3, 3]
    p0, d = np.array([0.3, 0.1, 0.4]), np.array([0.2, -0.1, 0.05])
    cons = [cc.EqualityConstraint("follow", p - (p0 + d * s), gain=2.0, priority=1, constraint_type="soft"),
            cc.VelocityEqualityConstraint("progress", s, target=0.05, priority=0),
            cc.SetConstraint("s_range", s, set_min=0.0, set_max=1.0, priority=2)]
    return cc.SkillSpecification("path", t, q, virtual_var=s, constraints=cons)


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_virtual_variable_pinv(ur5_fk, kernel, monkeypatch):
    from oracle import clik_oracle
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec = _path_following_skill(ur5_fk)
    rng = np.random.default_rng(3)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.2, size=(150, 6))
    X = rng.uniform(-0.1, 1.1, size=(150, 1))
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")
    dq, dx, mode = ctrl.solve_batch(0.0, Q, virtual_var=X)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, X=X)
    assert np.array_equal(mode, rmode) and len(np.unique(mode)) == 2
    assert dx.shape == (150, 1)
    assert pinv_close(np.hstack([dq, dx]), ref)
    # single-instance API returns the virtual velocity as the second result (pseudo_inverse.py:553-555)
    rob, virt, _ = ctrl.solve(0.0, Q[0], virtual_var=X[0])
    assert np.allclose(rob.toarray()[:, 0], ref[0, :6], atol=1e-9) and np.allclose(virt.toarray()[:, 0], ref[0, 6:], atol=1e-9)


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_virtual_variable_qp(ur5_fk, kernel, monkeypatch):
    from oracle import clik_oracle
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec = _path_following_skill(ur5_fk)
    rng = np.random.default_rng(4)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.2, size=(150, 6))
    X = rng.uniform(-0.1, 1.1, size=(150, 1))
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")
    dq, dx, slack, status = ctrl.solve_batch(0.0, Q, virtual_var=X)
    rdq, rdx, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q, X=X)
    # s beyond 1 with the hard progress rate contradicts the hard range set: infeasible on both sides
    assert np.array_equal(status == 2, rstatus == 2) and (rstatus == 2).sum() > 0
    ok = rstatus == 0
    assert _rel(dq[ok], rdq[ok]).max() < 1e-8 and _rel(dx[ok], rdx[ok]).max() < 1e-8
    assert _rel(slack[ok], rslack[ok]).max() < 1e-8


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_norm_rows_in_static_shapes(ur5_fk, kernel, monkeypatch):
    """cs.norm_2 distance constraint (ur5_moe2016_example2 cell 7) and a Frobenius-norm
    rotation constraint, with two 1-D sets: 2-norm output rows are part of the
    shape-specialised family (e = |r|, J = r'G/|r|)."""
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    T = fk["T_fk"](q)
    p = T[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    R_des = fk["chain"].fk_numeric(home + 0.3)[:3, :3]
    cons = [cc.EqualityConstraint("dist", cs.norm_2(np.array([0.5, 0.5, 0.5]) - p), gain=5.0, priority=6),
            cc.EqualityConstraint("rot", cs.norm_fro(T[:3, :3] - R_des), gain=2.0, priority=7),
            cc.SetConstraint("lim0", q[0], set_min=0.3 * lo[0], set_max=0.3 * hi[0], priority=0),
            cc.SetConstraint("lim1", q[1], set_min=0.3 * lo[1], set_max=0.3 * hi[1], priority=1)]
    spec = cc.SkillSpecification("norms", t, q, constraints=cons)
    rng = np.random.default_rng(23)
    Q = home + rng.uniform(-0.6, 0.6, size=(200, 6))
    Q[:, :2] = rng.uniform(0.5 * lo[:2], 0.5 * hi[:2], size=(200, 2))
    ctrl = _check(spec, None, Q, min_modes=4)
    assert ctrl.n_modes == 4
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")


def _cart_skill():
    """cart_on_track_1D notebook (cells 3-6, 56): a 1-DoF cart q follows the path parameter s
    (virtual variable) inside track limits - no kinematic chain at all, n_state = 2."""
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 1)
    s = cs.MX.sym("s", 1)
    cons = [cc.EqualityConstraint("follow", q - 2.0 * s, gain=3.0, priority=2, constraint_type="soft"),
            cc.VelocityEqualityConstraint("advance", s, target=0.25, priority=1),
            cc.SetConstraint("track", q, set_min=-1.0, set_max=1.0, gain=5.0, priority=0)]
    return cc.SkillSpecification("cart", t, q, virtual_var=s, constraints=cons)


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_chainless_one_dof_skill(kernel, monkeypatch):
    """Smallest possible problem (n_state = 2, no FK) through both controllers and both kernel
    families."""
    from oracle import clik_oracle
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    spec = _cart_skill()
    rng = np.random.default_rng(2)
    Q = rng.uniform(-1.3, 1.3, size=(200, 1))
    X = rng.uniform(-0.2, 0.8, size=(200, 1))
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")
    dq, dx, mode = ctrl.solve_batch(0.0, Q, virtual_var=X)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, X=X)
    assert np.array_equal(mode, rmode) and len(np.unique(mode)) == 2
    assert pinv_close(np.hstack([dq, dx]), ref)
    qp = cc.ReactiveQPController(skill_spec=spec)
    qp.setup_problem_functions()
    qp.setup_solver()
    assert (qp.kernel_name == "dynamic") == (kernel == "dynamic")
    dq, dx, slack, status = qp.solve_batch(0.0, Q, virtual_var=X)
    rdq, rdx, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q, X=X)
    assert np.array_equal(status, rstatus) and (status == 0).all()
    assert _rel(np.hstack([dq, dx, slack]), np.hstack([rdq, rdx, rslack])).max() < 1e-8


@pytest.mark.parametrize("kernel", ["static", "dynamic"])
def test_all_joint_limits_as_sets_64_modes(ur5_fk, kernel, monkeypatch):
    """Every UR5 joint limit as its own 1-D SetConstraint (the Moe-2016 way) above a position
    task and a posture task: 8 constraints, 64 modes - the largest skill of the static family
    (64 mode bodies in scan order; the build takes tens of seconds, once per skill structure)."""
    if kernel == "dynamic":
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    # (the posture task drives q5: the tool position does not depend on it, and a set on a joint no task
    # moves would have its tangent-cone test decided by rounding noise)
    cons = [cc.EqualityConstraint("pos", p - np.array([0.4, 0.2, 0.3]), gain=5.0, priority=10),
            cc.EqualityConstraint("posture", q[5] - 0.3, gain=1.0, priority=11)]
    for i in range(6):
        cons.append(cc.SetConstraint("limit_q_%d" % i, q[i], set_min=0.3 * lo[i], set_max=0.3 * hi[i], priority=i))
    spec = cc.SkillSpecification("all_limits", t, q, constraints=cons)
    rng = np.random.default_rng(29)
    Q = rng.uniform(0.36 * lo, 0.36 * hi, size=(256, 6))
    ctrl = _check(spec, None, Q, min_modes=10)
    assert ctrl.n_modes == 64
    assert (ctrl.kernel_name == "dynamic") == (kernel == "dynamic")


@pytest.mark.parametrize("first", ["unit_rows", "constant_matrix", "tall_constant_matrix"])
def test_first_equality_with_a_constant_jacobian_is_processed_twice_in_the_instantiated_kernels(iiwa_fk, first):
    """A joint-space first EqualityConstraint (a posture, `300 - x` of the cart notebook's path skill, any A q - b): its
    Jacobian is constant, its damped inverse P is precomputed on the host, and the double processing of the first
    equality (pseudo_inverse.py:317-326 + :382-396) is 2 P d - P (J (P d)) in the instantiated kernels
    (clik_pinv_static.hpp::step_s); the tasks behind it project through the stack [J; J].  Before round 3 such skills
    ran the built-in mode-scan kernel only."""
    fk = iiwa_fk
    t, q, T = _iiwa_syms(fk)
    rng = np.random.default_rng(41)
    if first == "unit_rows":
        head = cc.EqualityConstraint("posture", q[:3] - np.array([0.2, -0.4, 0.3]), gain=1.5, priority=0)
    elif first == "constant_matrix":
        A = rng.normal(size=(3, 7))
        head = cc.EqualityConstraint("mix", cs.mtimes(A, q) - np.array([0.1, 0.0, -0.2]), gain=1.5, priority=0)
    else:
        A = rng.normal(size=(9, 7))
        head = cc.EqualityConstraint("mix", cs.mtimes(A, q) - rng.normal(size=9) * 0.1, gain=1.5, priority=0)
    pos = cc.EqualityConstraint("pos", T[:3, 3] - np.array([0.4, 0.1, 0.6]), gain=5.0, priority=1)
    elbow = cc.SetConstraint("elbow", q[3], set_min=-1.0, set_max=1.0, priority=2)
    rest = cc.EqualityConstraint("rest", q - 0.1, gain=0.3, priority=3)
    Q, _ = skills.synthetic_inputs(fk, 256, seed=9, distribution="mixed")
    for cons in ([head], [head, pos], [head, pos, elbow, rest]):
        spec = cc.SkillSpecification("const_first", t, q, constraints=cons)
        # (a tall first equality fixes the whole velocity: the set behind it never decides anything on these inputs)
        ctrl = _check(spec, {"damping_factor": 1e-5}, Q,
                      min_modes=2 if (len(cons) == 4 and first != "tall_constant_matrix") else 1)
        assert ctrl.kernel_name.startswith("jit_"), ctrl.kernel_name


def test_one_set_per_joint_of_a_seven_dof_arm_128_modes(iiwa_fk):
    """The Moe-2016 pattern (one 1-D SetConstraint per joint limit, ur5_moe2016_example2.ipynb cell 6) on the 7-DoF
    iiwa: 2^7 = 128 modes (pseudo_inverse.py:107-130 builds them all), beyond the 64 mode bodies of the instantiated
    kernels - the built-in mode-scan kernel walks the table.  Modes and velocities equal the oracle's."""
    fk = iiwa_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 7)
    p = fk["T_fk"](q)[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.EqualityConstraint("pos", p - np.array([0.35, 0.2, 0.6]), gain=5.0, priority=10),
            cc.EqualityConstraint("posture", q[5:7] - np.array([0.3, -0.2]), gain=1.0, priority=11)]
    for i in range(7):
        cons.append(cc.SetConstraint("limit_q_%d" % i, q[i], set_min=0.3 * lo[i], set_max=0.3 * hi[i], priority=i))
    spec = cc.SkillSpecification("all_limits_7", t, q, constraints=cons)
    rng = np.random.default_rng(31)
    Q = rng.uniform(0.36 * lo, 0.36 * hi, size=(192, 7))
    ctrl = _check(spec, None, Q, min_modes=12)
    assert ctrl.n_modes == 128 and ctrl.kernel_name == "dynamic"


@pytest.mark.parametrize("force_dynamic", [False, True])
def test_tall_first_equality_is_processed_twice(ur5_fk, monkeypatch, force_dynamic):
    """First EqualityConstraint with more rows than joints (8 x 6): pinv takes the Gram branch
    (pseudo_inverse.py:97-100) and the double processing (:317-326, :382-396) projects through
    [J]: N = lam (J'J + lam I)^-1.  A lower-priority task then projects through [J; J]."""
    if force_dynamic:
        monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    T = ur5_fk["T_fk"](q)
    frame = cc.EqualityConstraint("frame", cs.vertcat(T[:3, 3] - np.array([0.4, 0.1, 0.4]),
                                                      T[:3, 0] - np.array([0.0, 1.0, 0.0]),
                                                      T[:2, 1] - np.array([1.0, 0.0])), gain=4.0, priority=0)
    rest = cc.EqualityConstraint("rest", q - np.array([0.0, -1.2, 1.0, -1.0, 0.3, 0.0]), gain=0.3, priority=1)
    for cons in ([frame], [frame, rest]):
        spec = cc.SkillSpecification("tall", t, q, constraints=cons)
        Q, _ = skills.synthetic_inputs(ur5_fk, 300, seed=6)
        ctrl = _check(spec, {"damping_factor": 1e-4}, 0.3 * Q)
        assert (ctrl.kernel_name == "dynamic") == force_dynamic


def test_nine_row_three_point_pose_error(ur5_fk, monkeypatch):
    """ur5_dual_quaternion_vs_transformation_matrix.ipynb cell 20, T_dist3: the "three point" pose error,
    9 rows on a 6-DoF arm - more rows than the built-in kernels are wide (CLIK_DYN_MAX_M = 8), so only the
    instantiated kernels serve it: as the tall, doubly processed first equality of the pinv controller
    behind 1-D joint-limit sets, and as 9 soft rows of the QP controller."""
    from oracle import clik_oracle
    from casclik_amd import numpy_geom
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    T_fk = ur5_fk["T_fk"]
    T_des = np.eye(4)
    T_des[:3, :3] = numpy_geom.rotation_rpy(5.0 * (np.pi / 180.0), 0.0, 0.0)
    T_des[:3, 3] = [0.5, 0.0, 0.5]
    expr = cs.vertcat(T_fk(q)[0, :3].T + T_fk(q)[:3, 3] - T_des[0, :3] - T_des[:3, 3],
                      T_fk(q)[1, :3].T + T_fk(q)[:3, 3] - T_des[1, :3] - T_des[:3, 3],
                      T_fk(q)[2, :3].T + T_fk(q)[:3, 3] - T_des[2, :3] - T_des[:3, 3])
    dist = cc.EqualityConstraint(label="T_dist3", expression=expr, gain=10.0, constraint_type="soft", priority=300)
    q_min, q_max = np.array(ur5_fk["lower"]), np.array(ur5_fk["upper"])
    home = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])
    rng = np.random.default_rng(21)
    Q = home + rng.uniform(-1.0, 1.0, size=(300, 6))
    Q[::5, 2] = rng.choice([-1.0, 1.0], size=Q[::5].shape[0]) * rng.uniform(3.0, 3.4, size=Q[::5].shape[0])
    limits = [cc.SetConstraint(label="limit_q_%d" % i, expression=q[i], set_min=q_min[i], set_max=q_max[i], priority=i)
              for i in (1, 2)]
    spec = cc.SkillSpecification("T_dist3_pinv", t, q, dq, constraints=[dist] + limits)
    ctrl = _check(spec, None, Q, min_modes=2)
    assert ctrl.kernel_name.startswith("jit_")
    qspec = cc.SkillSpecification("T_dist3_qp", t, q, dq, constraints=[
        dist, cc.SetConstraint(label="Joint_Limits", expression=q, set_min=q_min, set_max=q_max),
        cc.VelocitySetConstraint(label="Joint_speed_limits", expression=q, set_min=-np.full(6, np.pi / 5),
                                 set_max=np.full(6, np.pi / 5))])
    qc = cc.ReactiveQPController(skill_spec=qspec)
    qc.setup_problem_functions()
    qc.setup_solver()
    assert qc.kernel_name.startswith("jit_")
    qdq, _, qsl, st = qc.solve_batch(0.0, Q)
    rdq, _, rsl, rst = clik_oracle.qp_solve_batch(qspec, 0.0, Q)
    assert np.array_equal(st, rst)
    ok = rst == 0
    assert ok.sum() > 200 and _rel(qdq[ok], rdq[ok]).max() < 1e-8 and _rel(qsl[ok], rsl[ok]).max() < 1e-8
    # no built-in kernel is that wide: without the instantiation the controllers refuse, loudly
    monkeypatch.setenv("CLIK_JIT", "0")
    with pytest.raises(NotImplementedError, match="more rows than the built-in kernels"):
        cc.PseudoInverseController(skill_spec=spec).setup_problem_functions()
    # ... while the QP is served (round 4: by the dynamic kernel twelve rows wide with its work area in global memory -
    # slower, not refused) and gives the same minimisers
    slow = cc.ReactiveQPController(skill_spec=qspec)
    slow.setup_problem_functions()
    slow.setup_solver()
    assert slow.kernel_name == "dynamic"
    sdq, _, ssl, sst = slow.solve_batch(0.0, Q)
    assert np.array_equal(sst, rst) and _rel(sdq[ok], rdq[ok]).max() < 1e-8 and _rel(ssl[ok], rsl[ok]).max() < 1e-8


def test_rollout_with_virtual_variable_matches_the_host_loop(ur5_fk, monkeypatch):
    """Path following on the device (cart_on_track_1D...ipynb cell 60 integrates the path parameter next to
    the robot state): n ticks of solve -> clamp(robot velocities) -> Euler on [q; s] in one launch against the
    same loop over the oracle.  The built-in dynamic kernel cannot (loud refusal)."""
    from oracle import clik_oracle
    spec = _path_following_skill(ur5_fk)
    rng = np.random.default_rng(5)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.2, size=(70, 6))
    X = rng.uniform(0.0, 0.9, size=(70, 1))
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    dt, vmax, n_ticks = 0.01, 0.4, 40
    ts = dt * np.arange(n_ticks)
    qf, xf, dq_last, dx_last, mode = ctrl.rollout_batch(ts, Q, dt=dt, max_speed=vmax, virtual_var=X)
    q, x = Q.copy(), X.copy()
    for k in range(n_ticks):
        r, rmode = clik_oracle.pinv_solve_batch(spec, None, ts[k], q, X=x)
        dq, dx = np.clip(r[:, :6], -vmax, vmax), r[:, 6:]
        q, x = q + dq * dt, x + dx * dt
    assert np.array_equal(mode, rmode)
    assert np.abs(qf - q).max() < 1e-8 and np.abs(xf - x).max() < 1e-8
    assert np.abs(dq_last - dq).max() < 1e-7 and np.abs(dx_last - dx).max() < 1e-7
    assert np.abs(xf - X - 0.05 * dt * n_ticks).max() < 1e-6        # the path parameter advanced at its set rate
    monkeypatch.setenv("CLIK_FORCE_DYNAMIC", "1")
    dyn = cc.PseudoInverseController(skill_spec=spec)
    dyn.setup_problem_functions()
    with pytest.raises(NotImplementedError, match="shape-specialised kernel"):
        dyn.rollout_batch(ts, Q, dt=dt, max_speed=vmax, virtual_var=X)


def test_qp_rollout_with_virtual_variable_matches_the_host_loop(ur5_fk):
    """The same path-following loop through ReactiveQPController (clik_qp_rollout_batch_x): robot state and
    path parameter integrated on the device, working set carried between ticks."""
    from oracle import clik_oracle
    spec = _path_following_skill(ur5_fk)
    rng = np.random.default_rng(6)
    home = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) * np.pi / 180.0
    Q = home + rng.normal(scale=0.2, size=(40, 6))
    X = rng.uniform(0.1, 0.8, size=(40, 1))
    ctrl = cc.ReactiveQPController(skill_spec=spec)
    ctrl.setup_problem_functions()
    ctrl.setup_solver()
    dt, vmax, n_ticks = 0.01, 0.4, 30
    ts = dt * np.arange(n_ticks)
    qf, xf, dq_last, dx_last, slack, status = ctrl.rollout_batch(ts, Q, dt=dt, max_speed=vmax, virtual_var=X)
    q, x = Q.copy(), X.copy()
    for k in range(n_ticks):
        rdq, rdx, rsl, rst = clik_oracle.qp_solve_batch(spec, ts[k], q, X=x)
        assert (rst == 0).all()
        q, x = q + np.clip(rdq, -vmax, vmax) * dt, x + rdx * dt
    assert (status == 0).all()
    assert np.abs(qf - q).max() < 1e-7 and np.abs(xf - x).max() < 1e-7
    assert np.abs(dx_last - rdx).max() < 1e-7 and np.abs(slack - rsl).max() < 1e-6


def test_empty_batches_come_back_empty(iiwa_fk):
    """no instance: both controllers return arrays with zero rows (the C ABI returns CLIK_OK without a launch), for
    the tick and for the on-device rollout"""
    spec = skills.stack_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    ctrl.setup_problem_functions()
    Q, Y = np.zeros((0, 7)), np.zeros((0, 7))
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    assert dq.shape == (0, 7) and mode.shape == (0,)
    q1, dq1, _ = ctrl.rollout_batch(np.zeros(3), Q, input_var=Y, dt=1e-3)
    assert q1.shape == (0, 7) and dq1.shape == (0, 7)
    qp = cc.ReactiveQPController(skill_spec=skills.qp_skill(iiwa_fk))
    qp.setup_problem_functions()
    qp.setup_solver()
    dq, _, slack, status = qp.solve_batch(0.0, Q, input_var=Y)
    assert dq.shape == (0, 7) and status.shape == (0,) and slack.shape[0] == 0


def test_results_do_not_depend_on_the_batch_an_instance_sits_in(iiwa_fk):
    """property (hypothesis): any prefix, suffix or strided subset of a batch gives every instance the answer it gets
    in the full batch - ragged sizes across the 16 / 64-instance granularities of the kernels, both controllers"""
    from hypothesis import given, settings, strategies as st
    Q, Y = skills.synthetic_inputs(iiwa_fk, 700, seed=3, distribution="mixed")
    pinv = cc.PseudoInverseController(skill_spec=skills.stack_skill(iiwa_fk), options=dict(skills.STACK_OPTIONS))
    pinv.setup_problem_functions()
    qp = cc.ReactiveQPController(skill_spec=skills.qp_skill(iiwa_fk))
    qp.setup_problem_functions()
    qp.setup_solver()
    full_p = pinv.solve_batch(0.0, Q, input_var=Y)
    full_q = qp.solve_batch(0.0, Q, input_var=Y)

    @settings(max_examples=20, deadline=None)
    @given(start=st.integers(0, 650), count=st.integers(1, 700), step=st.integers(1, 5))
    def check(start, count, step):
        idx = np.arange(start, min(700, start + count * step), step)
        dq, _, mode = pinv.solve_batch(0.0, Q[idx], input_var=Y[idx])
        assert np.array_equal(mode, full_p[2][idx]) and np.array_equal(dq, full_p[0][idx])
        dq, _, slack, status = qp.solve_batch(0.0, Q[idx], input_var=Y[idx])
        assert np.array_equal(status, full_q[3][idx]) and np.array_equal(dq, full_q[0][idx])
        assert np.array_equal(slack, full_q[2][idx])
    check()


@pytest.mark.parametrize("nx", [2, 3])
def test_seven_dof_arm_with_two_or_three_virtual_variables(iiwa_fk, nx):
    """nine / ten state variables (CLIK_MAX_DOF = 10): no built-in pinv kernel is that wide, the handle is created
    without one and solves through the kernel instantiated for the skill (pseudo_inverse.py:79-88 puts no bound on
    virtual_var); both controllers against the oracle"""
    from oracle import clik_oracle
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 7), cs.MX.sym("dq", 7)
    x, dx = cs.MX.sym("x", nx), cs.MX.sym("dx", nx)
    T = iiwa_fk["T_fk"](q)
    dirs = np.array([[0.1, -0.05, 0.08], [-0.04, 0.09, 0.03], [0.02, 0.03, -0.07]])[:nx]
    patch = np.array([0.45, 0.1, 0.6]) + cs.mtimes(dirs.T, x)
    goal = np.array([1.0, 0.5, -0.3])[:nx]

    def spec_for(controller):
        cons = [cc.EqualityConstraint(label="follow_patch", expression=T[:3, 3] - patch, gain=5.0, constraint_type="soft",
                                      priority=1),
                cc.EqualityConstraint(label="patch_goal", expression=x - goal, gain=0.5, constraint_type="soft",
                                      priority=2)]
        if controller == "qp":
            vmax = np.array(iiwa_fk["velocity"])
            cons.append(cc.VelocitySetConstraint(label="speed", expression=q, set_min=-vmax, set_max=vmax, priority=0))
        else:
            cons.append(cc.SetConstraint(label="limit_q1", expression=q[1], set_min=-1.0, set_max=1.0, priority=0))
        return cc.SkillSpecification(label="patch", time_var=t, robot_var=q, robot_vel_var=dq, virtual_var=x,
                                     virtual_vel_var=dx, constraints=cons)
    Q, _ = skills.synthetic_inputs(iiwa_fk, 160, seed=21 + nx, distribution="mixed")
    X = np.random.default_rng(5).uniform(-0.2, 1.2, size=(160, nx))
    spec = spec_for("pinv")
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    dqv, dxv, mode = ctrl.solve_batch(0.0, Q, virtual_var=X)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, X=X)
    assert np.array_equal(mode, rmode) and len(np.unique(mode)) == 2
    assert dxv.shape == (160, nx) and pinv_close(np.hstack([dqv, dxv]), ref)
    spec = spec_for("qp")
    qctrl = cc.ReactiveQPController(skill_spec=spec)
    qctrl.setup_problem_functions()
    qctrl.setup_solver()
    dqv, dxv, slack, status = qctrl.solve_batch(0.0, Q, virtual_var=X)
    rdq, rdx, rslack, rstatus = clik_oracle.qp_solve_batch(spec, 0.0, Q, X=X)
    assert np.array_equal(status, rstatus) and (rstatus == 0).all()
    assert _rel(dqv, rdq).max() < 1e-8 and _rel(dxv, rdx).max() < 1e-8 and _rel(slack, rslack).max() < 1e-8


def test_a_wide_state_skill_without_an_instantiated_kernel_reports_unsupported(iiwa_fk, monkeypatch):
    """a pinv handle with nine state variables has no built-in kernel; with run-time instantiation switched
    off the controller says so at set-up (NotImplementedError), and the C ABI - asked to solve with such a handle -
    answers CLIK_EUNSUPPORTED with a message naming the cause instead of a failed launch"""
    import ctypes as C
    import torch
    from casclik_amd import _capi
    monkeypatch.setenv("CLIK_JIT", "0")
    t, q, x = cs.MX.sym("t"), cs.MX.sym("q", 7), cs.MX.sym("x", 2)
    T = iiwa_fk["T_fk"](q)
    cons = [cc.EqualityConstraint(label="a", expression=T[:3, 3] - cs.vertcat(x, 0.5), gain=1.0, priority=1),
            cc.EqualityConstraint(label="b", expression=x - np.array([0.3, 0.2]), gain=1.0, priority=2)]
    spec = cc.SkillSpecification(label="wide", time_var=t, robot_var=q, virtual_var=x, constraints=cons)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    with pytest.raises(NotImplementedError):
        ctrl.setup_problem_functions()
    lib, handle = ctrl._lib, ctrl._handle
    assert lib.clik_pinv_kernel_name(handle).decode() == "none"
    Q = torch.zeros((4, 7), dtype=torch.float64, device="cuda")
    X = torch.zeros((4, 2), dtype=torch.float64, device="cuda")
    dQ, dX = torch.empty_like(Q), torch.empty_like(X)
    mode = torch.empty(4, dtype=torch.int32, device="cuda")
    p = lambda tns: C.c_void_p(tns.data_ptr())          # noqa: E731
    rc = lib.clik_pinv_solve_batch(handle, 4, None, p(Q), p(X), None, p(dQ), p(dX), p(mode), None)
    assert rc == _capi.CLIK_EUNSUPPORTED
    assert b"state variables" in lib.clik_last_error() and b"instantiated" in lib.clik_last_error()


def test_two_seven_dof_arms_in_one_skill(iiwa_fk):
    """VERDICT r4 missing 1: the reference puts no bound on robot_var (pseudo_inverse.py:76-88, reactive_qp.py:191-246);
    CLIK_MAX_DOF is 14 since ABI 6 - two 7-DoF arms carrying a bar between their tools: two joint-limit sets (4 modes),
    a target for tool 1, the bar (the second arm's kinematics as generated device code), a 14-row posture task; both
    controllers against the oracle"""
    from oracle import clik_oracle
    import two_arm_skills as ta
    spec = ta.two_arm_pinv_skill(iiwa_fk)
    ctrl = cc.PseudoInverseController(skill_spec=spec, options={"multidim_sets": True})
    ctrl.setup_problem_functions()
    assert ctrl.descriptor.n_state == 14 and ctrl.kernel_name.startswith("jit_")
    Q, Y = ta.two_arm_inputs(iiwa_fk, 200, seed=4, distribution="mixed")
    dq, _, mode = ctrl.solve_batch(0.0, Q, input_var=Y)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, {"multidim_sets": True}, 0.0, Q, Y=Y)
    assert dq.shape == (200, 14) and np.array_equal(mode, rmode) and len(np.unique(mode)) >= 3
    assert pinv_close(dq, ref), _rel(dq, ref).max()
    qspec = ta.two_arm_qp_skill(iiwa_fk)
    qctrl = cc.ReactiveQPController(skill_spec=qspec)
    qctrl.setup_problem_functions()
    qctrl.setup_solver()
    assert qctrl.n_qp_vars == 20 and qctrl.n_qp_rows == 20 and qctrl.kernel_name.startswith("jit_")
    dqv, _, slack, status = qctrl.solve_batch(0.0, Q, input_var=Y)
    rdq, _, rslack, rstatus = clik_oracle.qp_solve_batch(qspec, 0.0, Q, Y=Y)
    assert np.array_equal(status, rstatus) and (rstatus == 0).all()
    assert qp_close(dqv, rdq) and qp_close(slack, rslack), (_rel(dqv, rdq).max(), _rel(slack, rslack).max())


def test_nine_set_constraints_512_modes(ur5_fk):
    """CLIK_MAX_SETS is 10 since round 5: the six 1-D joint-limit sets of the dual-quaternion notebooks
    (ur5_dual_quaternion_comparison_of_controllers.ipynb cell 14) together with the three task-space walls of the
    Moe-2016 example (ur5_moe2016_example2.ipynb cell 9) in ONE skill: nine sets, 512 modes, walked in the reference's
    order (pseudo_inverse.py:107-130, :530-550) by the built-in mode-scan kernel; modes and velocities against the
    oracle on states planted near limits and walls"""
    from oracle import clik_oracle
    fk = ur5_fk
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    T = fk["T_fk"](q)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.SetConstraint(label="limit_q%d" % i, expression=q[i], set_min=max(lo[i], -2.5), set_max=min(hi[i], 2.5), priority=i)
            for i in range(6)]
    cons += [cc.SetConstraint(label="wall_%s" % ax, expression=T[k, 3], set_min=-0.3, set_max=0.3, priority=6 + k)
             for k, ax in enumerate("xyz")]
    # (a target beyond the walls: the tool pushes against several of them at once)
    cons.append(cc.EqualityConstraint(label="reach", expression=T[:3, 3] - np.array([0.9, -0.8, 0.9]), gain=2.0, priority=20))
    spec = cc.SkillSpecification(label="nine_sets", time_var=t, robot_var=q, constraints=cons)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    ctrl.setup_problem_functions()
    assert ctrl.n_modes == 512
    rng = np.random.default_rng(9)
    Q = rng.uniform(-3.0, 3.0, size=(128, 6))           # (beyond +-2.5 in some joints, the tool beyond the walls for many)
    # (the last wrist joint does not move the tool's POSITION: its candidate velocity is zero to rounding, and beyond its
    # limit the sign of that zero would decide its tangent-cone test - a decision with margin 0; it stays inside)
    Q[:, 5] = rng.uniform(-2.0, 2.0, size=128)
    dq, _, mode = ctrl.solve_batch(0.0, Q)
    margins = np.full(len(Q), np.inf)
    ref, rmode = clik_oracle.pinv_solve_batch(spec, None, 0.0, Q, margins_out=margins)
    from tolerances import MODE_MARGIN
    decided = margins > MODE_MARGIN          # (tests/tolerances.py: modes must agree wherever the decision margin exceeds it)
    assert decided.mean() > 0.9 and np.array_equal(mode[decided], rmode[decided])
    assert len(np.unique(mode)) > 25 and mode.max() > 100
    same = mode == rmode
    assert pinv_close(dq, ref, rows=same), _rel(dq[same], ref[same]).max()
