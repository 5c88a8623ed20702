"""casclik API parity of the constraint / skill / front-end layer (CPU only)."""
import io
import sys

import numpy as np
import pytest

import casclik_amd as cc
from casclik_amd import skills
from casclik_amd import sym as cs
from casclik_amd.lowering import lower_skill, CLS_EQ, CLS_SET, CLS_VELSET, OUT_NORM2


def _syms(n=3):
    return cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("dq", n)


def test_import_surface():
    for name in ("EqualityConstraint", "SetConstraint", "VelocityEqualityConstraint",
                 "VelocitySetConstraint", "SkillSpecification", "PseudoInverseController",
                 "ReactiveQPController"):
        assert hasattr(cc, name)


def test_constraint_defaults_and_attributes():
    t, q, dq = _syms()
    e = cc.EqualityConstraint("e", q[0])
    assert (e.gain, e.constraint_type, e.priority, e.slack_weight) == (1.0, "hard", 1, 1.0)
    s = cc.SetConstraint("s", q)
    assert np.all(np.asarray(s.set_min) == -1e10) and np.all(np.asarray(s.set_max) == 1e10)
    v = cc.VelocityEqualityConstraint("v", q[1])
    assert v.target == 0.0
    vs = cc.VelocitySetConstraint("vs", q)
    assert vs.set_min == -1e10 and vs.set_max == 1e10
    assert repr(e).startswith("e<EqualityConstraint at 0x")
    assert e.size() == (1, 1) and s.size() == (3, 1)


def test_constraint_size_errors():
    t, q, dq = _syms()
    with pytest.raises(ValueError):
        cc.EqualityConstraint("bad", q, gain=np.eye(2))
    with pytest.raises(TypeError):
        cc.EqualityConstraint("bad", q, gain="high")
    with pytest.raises(ValueError):
        cc.SetConstraint("bad", q, set_min=np.zeros(2), set_max=np.ones(3))
    with pytest.raises(TypeError):
        cc.SetConstraint("bad", q, set_min="low")
    cc.EqualityConstraint("ok", q, gain=np.eye(3))
    cc.SetConstraint("ok", q[0], set_min=-1.0, set_max=1.0)


def test_skill_specification_bookkeeping():
    t, q, dq = _syms()
    y = cs.MX.sym("y", 2)
    x = cs.MX.sym("x", 1)
    cons = [cc.EqualityConstraint("a", q[0] - y[0], constraint_type="soft"),
            cc.SetConstraint("b", q, priority=0)]
    s = cc.SkillSpecification("demo", t, q, dq, virtual_var=x, input_var=y, constraints=cons)
    assert s.n_robot_var == 3 and s.n_virtual_var == 1 and s.n_input_var == 2
    assert s.n_slack_var == 1 and s.slack_var.size() == (1, 1)
    assert s._has_input and not s._has_virtual
    assert [c.label for c in s.constraints] == ["b", "a"]
    assert s.robot_vel_var is dq and s.virtual_vel_var.size() == (1, 1)
    buf, old = io.StringIO(), sys.stdout
    sys.stdout = buf
    try:
        s.print_constraints()
    finally:
        sys.stdout = old
    out = buf.getvalue()
    assert "SkillSpecification: demo" in out and "#0: b" in out and "Has input var: True" in out
    with pytest.raises(ValueError):
        cc.SkillSpecification("bad", t, q, robot_vel_var=cs.MX.sym("d", 2))
    with pytest.raises(TypeError):
        cc.SkillSpecification("bad", t, q, robot_vel_var=np.zeros(3))


def test_controller_options_defaults_and_repr():
    spec = skills.stack_skill()
    c = cc.PseudoInverseController(skill_spec=spec)
    assert c.options["feedforward"] is True and c.options["multidim_sets"] is False
    assert c.options["pinv_method"] == "damped" and c.options["damping_factor"] == 1e-7
    assert c.options["function_opts"]["jit"] is True
    assert repr(c) == "PseudoInverseController<stack>"
    qc = cc.ReactiveQPController(skill_spec=skills.qp_skill())
    assert qc.weight_shifter == 0.001 and qc.options["solver_name"] == "qpoases"
    assert np.allclose(qc.robot_var_weights, 1.0) and qc.slack_var_weights.shape == (6,)
    assert repr(qc) == "ReactiveQPController<qp_pose>"
    with pytest.raises(ValueError):
        cc.ReactiveQPController(skill_spec=skills.qp_skill(), robot_var_weights=[1.0, 2.0])


def test_solve_initial_problem_pinv_returns_zeros():
    c = cc.PseudoInverseController(skill_spec=skills.pose_skill())
    virt, slack = c.solve_initial_problem(0.0, np.zeros(7))
    assert virt is None and np.all(slack.toarray() == 0) and slack.size() == (6, 1)


# ------------------------------------------------------------------ front-end
def test_sym_algebra_and_function():
    t, q, _ = _syms()
    expr = cs.vertcat(cs.sin(q[0]) * 2 + t, cs.norm_2(q), cs.mtimes(np.array([[1.0, 2.0, 3.0]]), q))
    f = cs.Function("f", [t, q], [expr])
    val = f(0.5, [0.1, 0.2, 0.3]).toarray()[:, 0]
    assert np.allclose(val, [2 * np.sin(0.1) + 0.5, np.sqrt(0.14), 1.4])
    g = cs.Function("g", [q], [q[::-1] + 1.0])
    assert np.allclose(g(np.array([1.0, 2.0, 3.0])).toarray()[:, 0], [4.0, 3.0, 2.0])
    # symbolic call substitutes
    h = f(t, q * 2.0)
    assert isinstance(h, cs.MX) and h.size() == (3, 1)
    assert cs.inv(np.diag([2.0, 4.0])).toarray()[1, 1] == 0.25
    assert cs.vertcat([1.0] * 3).toarray().shape == (3, 1)
    d = cs.DM([1.0, 2.0])
    assert d.toarray().shape == (2, 1) and float(d[1]) == 2.0


def test_fk_atom_matches_chain(iiwa_fk):
    q = cs.MX.sym("q", 7)
    T = iiwa_fk["T_fk"](q)
    f = cs.Function("T", [q], [T])
    q0 = np.linspace(-1, 1, 7)
    assert np.allclose(f(q0).toarray(), iiwa_fk["chain"].fk_numeric(q0))
    J = cc.EqualityConstraint("p", T[:3, 3]).jacobian(q)
    Jn = cs.Function("J", [q], [J])(q0).toarray()
    h = 1e-6
    for k in range(7):
        qp, qm = q0.copy(), q0.copy()
        qp[k] += h
        qm[k] -= h
        fd = (iiwa_fk["chain"].fk_numeric(qp)[:3, 3] - iiwa_fk["chain"].fk_numeric(qm)[:3, 3]) / (2 * h)
        assert np.allclose(Jn[:, k], fd, atol=1e-8)


# ------------------------------------------------------------------ lowering
def test_lowering_of_baseline_skills(iiwa_fk):
    d = lower_skill(skills.stack_skill(iiwa_fk))
    assert (d.n_q, d.n_x, d.n_y) == (7, 0, 7) and len(d.joints) == 8 and d.n_sets == 1
    assert [t["cls"] for t in d.tasks] == [CLS_SET, CLS_EQ, CLS_EQ]
    assert [t["m"] for t in d.tasks] == [7, 6, 7]
    assert d.quat_src == 2 and d.quat_yi == [3, 4, 5, 6] and d.uses_fk
    assert np.allclose(d.tasks[0]["set_max"][:7], iiwa_fk["upper"])
    pose_rows = [d.rows[r] for r in d.tasks[1]["out_row0"][:6]]
    assert all(r["b"].any() for r in pose_rows[:3]) and all(r["h"].any() for r in pose_rows[3:])
    assert [r["yi"][0] for r in pose_rows[:3]] == [0, 1, 2] and all(r["yc"][0] == -1.0 for r in pose_rows[:3])
    dq = lower_skill(skills.qp_skill(iiwa_fk))
    assert [t["cls"] for t in dq.tasks] == [CLS_VELSET, CLS_EQ] and dq.n_slack == 6


def test_lowering_norm_and_time_terms(ur5_fk):
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    T = ur5_fk["T_fk"](q)
    dist = cc.EqualityConstraint("dist", cs.norm_2(np.array([0.5, 0.5, 0.5]) - T[:3, 3]), gain=50.0)
    track = cc.EqualityConstraint("track", T[:3, 3] - cs.vertcat(cs.sin(0.1 * t), 0.2 * t, 0.3), priority=2)
    d = lower_skill(cc.SkillSpecification("s", t, q, constraints=[dist, track]))
    assert d.tasks[0]["out_kind"][0] == OUT_NORM2 and d.tasks[0]["out_nrows"][0] == 3
    assert d.n_tslots == 2
    tt = d.time_terms(2.0)
    # the target enters the expression with a minus sign
    assert np.allclose(sorted(tt[:2]), sorted([-np.sin(0.2), -0.4]))
    assert np.allclose(sorted(tt[2:]), sorted([-0.1 * np.cos(0.2), -0.2]))


def test_lowering_rejects_what_the_device_cannot_do(iiwa_fk):
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 7), cs.MX.sym("dq", 7)
    T = iiwa_fk["T_fk"](q)
    # a product of state-dependent terms is outside the row table: it becomes generated code
    # (tests/test_codegen.py), not a refusal
    d = lower_skill(cc.SkillSpecification("s", t, q, constraints=[cc.EqualityConstraint("sq", T[0, 3] * T[1, 3])]))
    assert 0 in d.extern_code and d.uses_fk
    with pytest.raises(NotImplementedError, match="velocity variables"):
        lower_skill(cc.SkillSpecification("s", t, q, dq, constraints=[cc.EqualityConstraint("v", dq[0])]))
    with pytest.raises(NotImplementedError, match="rows"):
        lower_skill(cc.SkillSpecification("s", t, q, constraints=[
            cc.EqualityConstraint("fifteen", cs.vertcat(T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3], q[0], q[1], q[2]))]))
    # (thirteen rows fit since ABI 6: CLIK_MAX_M = 14)
    assert lower_skill(cc.SkillSpecification("s", t, q, constraints=[
        cc.EqualityConstraint("thirteen", cs.vertcat(T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3], q[0]))])).tasks[0]["m"] == 13
    # nine rows (the "three point" pose error of ur5_dual_quaternion_vs_transformation_matrix.ipynb cell 20)
    # fit the ABI; only the shape-specialised kernels are wide enough for them
    nine = lower_skill(cc.SkillSpecification("s", t, q, constraints=[
        cc.EqualityConstraint("nine", cs.vertcat(T[:3, 0], T[:3, 1], T[:3, 2]))]))
    assert nine.tasks[0]["m"] == 9
    # one 1-D set per joint of a 7-DoF arm (128 modes) lowers, and so do ten sets (1024 modes, CLIK_MAX_SETS since round 5);
    # eleven are beyond the device limit
    seven_sets = [cc.SetConstraint("s%d" % i, q[i], set_min=-1.0, set_max=1.0, priority=i) for i in range(7)]
    assert lower_skill(cc.SkillSpecification("s", t, q, constraints=seven_sets)).n_sets == 7
    ten_sets = seven_sets + [cc.SetConstraint("w%d" % i, T[i, 3], set_min=-1.0, set_max=1.0, priority=9) for i in range(3)]
    assert lower_skill(cc.SkillSpecification("s", t, q, constraints=ten_sets)).n_sets == 10
    eleven_sets = ten_sets + [cc.SetConstraint("w3", T[0, 0], set_min=-1.0, set_max=1.0, priority=9)]
    with pytest.raises(NotImplementedError, match="modes"):
        lower_skill(cc.SkillSpecification("s", t, q, constraints=eleven_sets))


def test_controller_needs_the_hip_library_or_gpu():
    """No CPU fallback: without a GPU the controller setup must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = cc.PseudoInverseController(skill_spec=skills.pose_skill())
    with pytest.raises(RuntimeError):
        c.setup_problem_functions()
    with pytest.raises(RuntimeError, match="setup"):
        c.solve(0.0, np.zeros(7), input_var=np.zeros(7))


def test_geometry_helpers_compose_like_the_joint_transforms():
    """casadi_geom / numpy_geom look-alikes (urdf2casadi call surface of the dual-quaternion notebooks,
    ur5_dual_quaternion_comparison_of_controllers.ipynb cell 4): the elementary dual quaternions compose to
    the joint transforms, numbers and expressions agree, and cs.jacobian differentiates FK expressions."""
    from casclik_amd import casadi_geom as g, numpy_geom as ng, skills
    xyz, rpy, ax = [0.3, -0.2, 0.5], [0.4, -0.3, 1.1], np.array([0.0, 0.0, 1.0])
    base = ng.dual_quaternion_revolute(xyz, rpy, [1, 0, 0], 0.0)
    assert np.abs(ng.dual_quaternion_product(ng.dual_quaternion_translation(xyz), ng.dual_quaternion_rpy(rpy)) - base).max() < 1e-15
    assert np.abs(ng.dual_quaternion_to_transformation_matrix(base) - ng.T_rpy(xyz, *rpy)).max() < 1e-15
    assert np.abs(ng.dual_quaternion_product(base, ng.dual_quaternion_axis_rotation(ax, 0.7))
                  - ng.dual_quaternion_revolute(xyz, rpy, ax, 0.7)).max() < 1e-15
    assert np.abs(ng.dual_quaternion_product(base, ng.dual_quaternion_axis_translation(ax, 0.7))
                  - ng.dual_quaternion_prismatic(xyz, rpy, ax, 0.7)).max() < 1e-15
    r = cs.SX.sym("rpy", 3)
    f = cs.Function("dqrpy", [r], [g.dual_quaternion_rpy(r)])
    assert np.abs(f(rpy).toarray().ravel() - ng.dual_quaternion_rpy(rpy)).max() < 1e-15
    fk = skills.ur5()
    q = cs.MX.sym("q", 6)
    J = cs.Function("J", [q], [cs.jacobian(fk["T_fk"](q)[:3, 3], q)])
    q0 = np.array([0.3, -1.2, 1.0, -0.5, 0.4, 0.1])
    num = np.zeros((3, 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        num[:, k] = (fk["T_fk"](q0 + d).toarray()[:3, 3] - fk["T_fk"](q0 - d).toarray()[:3, 3]) / 2e-6
    assert np.abs(J(q0).toarray() - num).max() < 1e-8


def test_initial_problem_slack_matches_the_reference_formulation(ur5_fk):
    """solve_initial_problem (reactive_qp.py:300-459) without virtual variables: the controller's
    closed form (each slack clamped to its interval nearest zero) against the oracle's literal QP over
    the slack variables, for the notebooks' call ``solve_initial_problem(0, UR5_home)[-1]``
    (ur5_dual_quaternion_comparison_of_controllers.ipynb cell 17) and with a robot velocity given."""
    from oracle import clik_oracle
    from extern_skills import dual_quaternion_skill
    home = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    T = ur5_fk["T_fk"](q)
    soft_set = cc.SkillSpecification("soft_set", t, q, constraints=[
        cc.SetConstraint("band", T[:2, 3], set_min=np.array([0.3, -0.1]), set_max=np.array([0.5, 0.1]), gain=2.0,
                         constraint_type="soft", priority=1),
        cc.EqualityConstraint("track", T[:3, 3] - cs.vertcat(0.4 + 0.1 * cs.sin(t), 0.0, 0.5), gain=np.diag([1.0, 2.0, 3.0]),
                              constraint_type="soft", priority=2, slack_weight=4.0),
        cc.VelocitySetConstraint("speed", q, set_min=-np.ones(6), set_max=np.ones(6), priority=0)])
    for spec in (dual_quaternion_skill(ur5_fk, "Q_dist2"), dual_quaternion_skill(ur5_fk, "cart_dist"), soft_set):
        ctrl = cc.ReactiveQPController(skill_spec=spec)
        ctrl.setup_initial_problem_solver()
        for dq0 in (None, np.array([0.1, -0.2, 0.3, 0.0, 0.05, -0.1])):
            virt, slack = ctrl.solve_initial_problem(0.7, home, robot_vel_var0=dq0)
            rvirt, rslack = clik_oracle.qp_initial_problem(spec, 0.7, home, dq0=dq0)
            assert virt is None and rvirt is None
            assert slack.toarray().shape == (spec.n_slack_var, 1)
            assert np.abs(slack.toarray()[:, 0] - rslack).max() < 1e-10
    hard = cc.SkillSpecification("hard", t, q, constraints=[cc.EqualityConstraint("p", T[:3, 3], constraint_type="hard")])
    ctrl = cc.ReactiveQPController(skill_spec=hard)
    ctrl.setup_initial_problem_solver()
    assert ctrl.solve_initial_problem(0.0, home) == (None, None)


def test_integration_method_factories():
    """get_euler_function / get_rk4_function with the reference's call shape (integration_methods.py:11-23)"""
    import numpy as np
    from casclik_amd import sym as cs
    from casclik_amd.integration_methods import get_euler_function, get_rk4_function
    x = cs.MX.sym("x", 2)
    rate = lambda v: cs.vertcat(v[1], -v[0])        # harmonic oscillator  # noqa: E731
    dt = 0.01
    fe, fr = get_euler_function(x, rate, dt), get_rk4_function(x, rate, dt)
    ve = vr = np.array([1.0, 0.0])
    for _ in range(100):
        ve = np.array(fe(ve)).reshape(-1)
        vr = np.array(fr(vr)).reshape(-1)
    exact = np.array([np.cos(1.0), -np.sin(1.0)])
    assert np.abs(vr - exact).max() < 1e-10                 # fourth order
    assert 1e-3 < np.abs(ve - exact).max() < 1e-2           # first order
    # one Euler step is exactly x + dx dt
    assert np.allclose(np.array(fe([2.0, 3.0])).reshape(-1), [2.0 + 3.0 * dt, 3.0 - 2.0 * dt], rtol=0, atol=1e-15)
    # symbolic call: the step function composes into an expression (as cs.Function does)
    y = cs.MX.sym("y", 2)
    two = cs.Function("two", [y], [fr(fr(y))])
    assert np.abs(np.array(two([1.0, 0.0])).reshape(-1) - [np.cos(2 * dt), -np.sin(2 * dt)]).max() < 1e-11


def test_nullspace_and_symbolic_pinv(ur5_fk):
    """BaseConstraint.nullspace = I - pinv(J) J (constraints.py:82-85) with cs.pinv / cs.solve of the shim"""
    import numpy as np
    import casclik_amd as cc
    from casclik_amd import sym as cs
    q = cs.MX.sym("q", 6)
    c = cc.EqualityConstraint("pos", ur5_fk["T_fk"](q)[:3, 3] - np.array([0.3, 0.2, 0.4]), gain=1.0)
    N = cs.Function("N", [q], [c.nullspace(q)])
    J = cs.Function("J", [q], [c.jacobian(q)])
    qv = np.array([0.3, -1.2, 1.0, -0.5, 0.7, 0.2])
    Nv, Jv = np.array(N(qv)), np.array(J(qv))
    assert np.abs(Jv @ Nv).max() < 1e-13 and np.abs(Nv @ Nv - Nv).max() < 1e-13
    assert np.abs(Nv - (np.eye(6) - np.linalg.pinv(Jv) @ Jv)).max() < 1e-12
    # tall matrices take the other branch of cs.pinv; constants are evaluated on the spot
    A = np.random.default_rng(0).normal(size=(5, 3))
    assert np.abs(np.array(cs.pinv(A)) - np.linalg.pinv(A)).max() < 1e-12
    assert np.abs(np.array(cs.solve(A.T @ A, np.eye(3))) - np.linalg.inv(A.T @ A)).max() < 1e-12


def test_out_tensor_validation_helper():
    """outputs handed to the kernels by pointer are validated (shape, dtype, device, contiguity)"""
    import pytest
    import torch
    from casclik_amd.controllers.base_controller import check_out_tensor
    dev = torch.device("cpu")
    check_out_tensor(torch.empty((4, 7), dtype=torch.float64), (4, 7), "float64", dev, "out")
    check_out_tensor(None, (4, 7), "float64", dev, "out")
    for bad in (torch.empty((4, 7), dtype=torch.float32), torch.empty((3, 7), dtype=torch.float64),
                torch.empty((7, 4), dtype=torch.float64).T, [[0.0] * 7] * 4):
        with pytest.raises(ValueError):
            check_out_tensor(bad, (4, 7), "float64", dev, "out")
    with pytest.raises(ValueError):
        check_out_tensor(torch.empty((4,), dtype=torch.int64), (4,), "int32", dev, "mode_out")


def test_print_constraints_matches_the_notebooks_stored_output():
    """SkillSpecification.print_constraints (skill_specification.py:202-217) against the text the reference
    notebooks STORE in their output cells (tests/golden/notebook_prints.json): stable priority sort, the
    virtual / input dependence flags and the per-class counts, verbatim."""
    import io
    import json
    import os
    import sys
    import casclik_amd as cc
    from casclik_amd import sym as cs
    data = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_prints.json")))
    for fx in data["prints"]:
        t, p, dp = cs.MX.sym("t"), cs.MX.sym("p"), cs.MX.sym("dp")
        x, dx = (cs.MX.sym("x"), cs.MX.sym("dx")) if fx["virtual"] else (None, None)
        sym = {"q": p, "x": x, "t": t}
        cons = []
        for c in fx["constraints"]:
            expr = 0.5
            for d in c["depends"]:
                expr = expr + (0.4 * cs.sin(0.3 * sym[d]) if d != "q" else sym[d])
            kw = dict(label=c["label"], expression=expr, priority=c["priority"], constraint_type=c["constraint_type"])
            if c["cls"] == "SetConstraint":
                cons.append(cc.SetConstraint(set_min=0.0, set_max=1.0, **kw))
            elif c["cls"] == "VelocitySetConstraint":
                cons.append(cc.VelocitySetConstraint(set_min=-0.275, set_max=0.275, **kw))
            else:
                cons.append(cc.EqualityConstraint(gain=1.0, **kw))
        spec = cc.SkillSpecification(label=fx["label"], time_var=t, robot_var=p, robot_vel_var=dp, virtual_var=x,
                                     virtual_vel_var=dx, constraints=cons)
        buf, old = io.StringIO(), sys.stdout
        sys.stdout = buf
        try:
            spec.print_constraints()
        finally:
            sys.stdout = old
        assert buf.getvalue() == fx["stdout"], (fx["notebook"], fx["cell"], buf.getvalue())


def test_expression_attributes_are_lowered_to_generated_code(ur5_fk):
    """MX gains / bounds / targets (casclik/constraints.py:35-39, :199-206): constants are folded, expressions of
    the skill's variables become ExternAttr code and the task records which attributes they replace"""
    from casclik_amd.lowering import lower_skill, ATTR_GAIN, ATTR_SET_MIN, ATTR_SET_MAX, ATTR_TARGET
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    p = ur5_fk["T_fk"](q)[:3, 3]
    gain = 1.0 + 0.5 * cs.sin(t) + q[0] * q[0]
    lim = cs.vertcat(*[1.0 + 0.1 * cs.cos(t) for _ in range(6)])
    spec = cc.SkillSpecification("attrs", t, q, constraints=[
        cc.SetConstraint("lims", q, gain=gain, set_min=-lim, set_max=lim, priority=0),
        cc.EqualityConstraint("move", p, gain=cs.MX(2.5) if hasattr(cs, "MX") else 2.5, constraint_type="soft", priority=1),
        cc.VelocityEqualityConstraint("spin", q[5], target=0.1 * cs.sin(t), constraint_type="soft", priority=2)])
    d = lower_skill(spec)
    assert d.tasks[0]["attr_ext"] == ATTR_GAIN | ATTR_SET_MIN | ATTR_SET_MAX
    assert d.tasks[1]["attr_ext"] == 0 and d.tasks[1]["gain"][0] == 2.5
    assert d.tasks[2]["attr_ext"] == ATTR_TARGET
    src = d.extern_source()
    assert "struct ExternAttr<0>" in src and "struct ExternAttr<2>" in src and "ExternAttr<1>" not in src
    assert src.count("a[") == 1 + 6 + 6 + 1        # scalar gain, 6 + 6 bounds; one target


def test_error_behaviour_equals_the_reference_packages():
    """tests/golden/error_cases.py through the product's classes against what the REFERENCE classes raised for the same
    constructions (tests/golden/ref_errors.json, recorded by make_ref_golden.py: exception class and a digest of the
    message).  Deliberate differences: the sum of two constraints passes the reference's checks and is then refused
    (its gain assembly needs CasADi's slice assignment; over the stand-in the reference itself stops there), and a 2-D
    array of weights is refused at construction instead of later."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import error_cases
    ref = json.load(open(os.path.join(here, "golden", "ref_errors.json")))
    got = error_cases.cases(cs, cc)
    assert set(got) == set(ref)
    deliberate = {"eq_add_ok": "NotImplementedError", "qp_w_matrix": "ValueError", "qp_slack_w_matrix": "ValueError"}
    for name, (kind, text) in got.items():
        if name in deliberate:
            assert kind == deliberate[name], (name, kind)
            continue
        assert [kind, error_cases.digest(text)] == ref[name], (name, kind, text, ref[name])


def test_small_public_methods_give_the_reference_packages_values():
    """tests/golden/make_ref_golden_cases.py::api_value_cases through the product against the dict the REFERENCE classes
    produced (tests/golden/ref_api_values.json): Jacobians, jtimes, nullspace, sizes, defaults (+-1e10 bounds, gains),
    class attributes (the two velocity constraints report "BaseConstraint", as in the reference), the skill's counters,
    dependence flags, priority sort and the re-sort when `constraints` is assigned."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ref_golden_cases
    ref = json.load(open(os.path.join(here, "golden", "ref_api_values.json")))
    got = json.loads(json.dumps(make_ref_golden_cases.api_value_cases(cs, cc), sort_keys=True))
    assert set(got) == set(ref)
    for key in ref:
        assert got[key] == ref[key], (key, got[key], ref[key])
