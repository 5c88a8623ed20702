"""Pins of the CPU oracle (CPU only).

The reference has no tests and cannot run here (no CasADi), so the oracle is
pinned by: the forward-kinematics value stored in the reference notebooks, the
mode ordering, algebraic invariants of the damped pseudo-inverse, agreement of
its AD Jacobians with finite differences, KKT optimality of its QP answers,
agreement of two independent implementations (numpy/dual numbers vs C/closed
forms), and the committed golden vectors.
"""
import os

import numpy as np
import pytest

from casclik_amd import skills
from casclik_amd import sym as cs
import casclik_amd as cc
from oracle import clik_oracle as orc
from tolerances import PINV_RTOL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clik_golden.npz")


def _rel(a, ref):
    return np.abs(a - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))


# ---------------------------------------------------------------- reference KATs
def test_ur5_fk_kat(ur5_fk):
    """||p_tool0|| = 1.0192 at UR5 home, the value printed by the reference
    notebooks (ur5_transformation_matrix_comparison_of_controllers.ipynb:92,
    ur5_input_experiment.ipynb:94)."""
    T = ur5_fk["T_fk"]([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0]).toarray()
    assert abs(np.linalg.norm(T[:3, 3]) - 1.0192) < 5e-5
    assert np.allclose(T[:3, 3], [0.0, 0.19145, 1.001059], atol=1e-9)
    assert np.allclose(T[:3, :3], [[1, 0, 0], [0, 0, 1], [0, -1, 0]], atol=1e-9)


def test_ur5_joint_limits_kat(ur5_fk):
    """Limits printed in ur5_moe2016_example2.ipynb:125-126."""
    up = np.array(ur5_fk["upper"])
    assert np.allclose(up, [6.28318531, 6.28318531, 3.14159265, 6.28318531, 6.28318531, 6.28318531], atol=1e-8)
    assert np.allclose(np.array(ur5_fk["lower"]), -up)


def test_iiwa_fk_zero(iiwa_fk):
    T = iiwa_fk["T_fk"]([0.0] * 7).toarray()
    assert np.allclose(T[:3, :3], np.eye(3))
    assert np.allclose(T[:3, 3], [0, 0, 0.36 + 0.42 + 0.4 + 0.126])


def test_activation_map_order():
    """pseudo_inverse.py:107-130: set 0 is the least significant bit, modes are
    stably sorted by the number of active sets."""
    assert orc.activation_map(0) == []
    assert orc.activation_map(1) == [[0], [1]]
    assert orc.activation_map(2) == [[0, 0], [1, 0], [0, 1], [1, 1]]
    m3 = orc.activation_map(3)
    assert m3[0] == [0, 0, 0] and m3[1:4] == [[1, 0, 0], [0, 1, 0], [0, 0, 1]] and m3[-1] == [1, 1, 1]
    spec = skills.stack_skill()
    ctrl = cc.PseudoInverseController(skill_spec=spec, options=dict(skills.STACK_OPTIONS))
    assert ctrl.activation_map == [[0], [1]] and ctrl.n_modes == 2


def test_priority_sort_is_stable():
    """skill_specification.py:139-142."""
    spec = skills.stack_skill()
    assert [c.label for c in spec.constraints] == ["joint_limits", "tool_pose", "joint_centering"]
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 2)
    cons = [cc.EqualityConstraint("a", q[0], priority=2), cc.EqualityConstraint("b", q[1], priority=1),
            cc.EqualityConstraint("c", q[0] + q[1], priority=2), cc.SetConstraint("d", q[0], priority=1,
                                                                                 set_min=-1.0, set_max=1.0)]
    s = cc.SkillSpecification("s", t, q, constraints=cons)
    assert [c.label for c in s.constraints] == ["b", "d", "a", "c"]
    cnt = s.count_constraints()
    assert cnt["all"] == 4 and cnt["equality"] == 3 and cnt["set"] == 1 and cnt["hard"] == 4


# ---------------------------------------------------------------- Jacobians
@pytest.mark.parametrize("robot", ["iiwa", "ur5"])
def test_ad_jacobian_vs_finite_differences(robot, iiwa_fk, ur5_fk):
    fk = iiwa_fk if robot == "iiwa" else ur5_fk
    spec = skills.pose_skill(fk)
    Q, Y = skills.synthetic_inputs(fk, 5, seed=1)
    ev = orc.ExprEvaluator(spec, 0.0, Q, Y)
    e, Jt, J = ev.vector(spec.constraints[0].expression)
    h = 1e-6
    for j in range(Q.shape[1]):
        Qp, Qm = Q.copy(), Q.copy()
        Qp[:, j] += h
        Qm[:, j] -= h
        ep = orc.ExprEvaluator(spec, 0.0, Qp, Y).vector(spec.constraints[0].expression)[0]
        em = orc.ExprEvaluator(spec, 0.0, Qm, Y).vector(spec.constraints[0].expression)[0]
        assert np.abs((ep - em) / (2 * h) - J[:, :, j]).max() < 1e-8
    assert np.abs(Jt).max() == 0.0


def test_time_derivative_feedforward():
    """d e/d t of a trajectory-tracking expression (moe2016 trajectory form)."""
    fk = skills.iiwa()
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 7)
    path = cs.vertcat(0.5 * cs.sin(0.1 * t) * cs.sin(0.1 * t) + 0.2,
                      0.5 * cs.cos(0.1 * t) + 0.25 * cs.sin(0.1 * t),
                      0.5 * cs.sin(0.1 * t) * cs.cos(0.1 * t) + 0.1)
    expr = fk["T_fk"](q)[:3, 3] - path
    spec = cc.SkillSpecification("track", t, q, constraints=[cc.EqualityConstraint("p", expr)])
    Q, _ = skills.synthetic_inputs(fk, 3, seed=2)
    t0, h = 3.7, 1e-6
    e, Jt, _ = orc.ExprEvaluator(spec, t0, Q, None).vector(expr)
    ep = orc.ExprEvaluator(spec, t0 + h, Q, None).vector(expr)[0]
    em = orc.ExprEvaluator(spec, t0 - h, Q, None).vector(expr)[0]
    assert np.abs((ep - em) / (2 * h) - Jt).max() < 1e-8


# ---------------------------------------------------------------- pinv invariants
def test_damped_pinv_singular_value_invariant():
    """For one full-rank task  J * pinv_damped(J) * xd  scales each singular
    direction by s^2/(s^2+lam)  (pseudo_inverse.py:92-105)."""
    rng = np.random.default_rng(0)
    opt = orc.default_pinv_options({"damping_factor": 1e-3})
    for rows, cols in [(6, 7), (3, 7), (7, 7), (9, 6)]:
        J = rng.normal(size=(rows, cols))
        P = orc.dpinv(J, opt)
        U, s, Vt = np.linalg.svd(J, full_matrices=False)
        expect = (Vt.T * (s / (s * s + 1e-3))) @ U.T
        assert np.abs(P - expect).max() < 1e-10


def test_quirk_matters(iiwa_fk):
    """The first EqualityConstraint is processed twice
    (pseudo_inverse.py:317-326 and :382-396): v = P d + (I - P J) P d, and the
    stack for lower priorities is [J; J].  A textbook implementation
    (v = P d, stack [J]) differs far beyond PINV_RTOL."""
    spec = skills.pose_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 64, seed=3)
    dq, _ = orc.pinv_solve_batch(spec, None, 0.0, Q, Y=Y)
    opt = orc.default_pinv_options()
    ev = orc.ExprEvaluator(spec, 0.0, Q, Y)
    e, Jt, J = ev.vector(spec.constraints[0].expression)
    worst = 0.0
    for b in range(len(Q)):
        P = orc.dpinv(J[b], opt)
        d = -10.0 * e[b]
        v1 = P @ d
        literal = v1 + (np.eye(7) - P @ J[b]) @ v1
        assert np.abs(literal - dq[b]).max() <= 1e-9 * (1 + np.abs(dq[b]).max())
        worst = max(worst, np.abs(v1 - dq[b]).max() / (1 + np.abs(dq[b]).max()))
    assert worst > 100 * PINV_RTOL


def test_tangent_cone_rules():
    """pseudo_inverse.py:162-185 (1-D, boundary counts as inside) and :222-252."""
    assert orc.in_tangent_cone_1d(0.5, 0.0, 1.0, -3.0)
    assert orc.in_tangent_cone_1d(1.0, 0.0, 1.0, +3.0)        # on the max bound: inside (1e-12 margin)
    assert not orc.in_tangent_cone_1d(1.1, 0.0, 1.0, +1.0)
    assert orc.in_tangent_cone_1d(1.1, 0.0, 1.0, -1.0)
    assert orc.in_tangent_cone_1d(-0.1, 0.0, 1.0, +1.0)
    assert not orc.in_tangent_cone_1d(-0.1, 0.0, 1.0, -1.0)
    lo, hi = np.zeros(2), np.ones(2)
    assert orc.in_tangent_cone_multidim(np.array([0.5, 0.5]), lo, hi, np.array([9.0, 9.0]))
    assert not orc.in_tangent_cone_multidim(np.array([0.0, 0.5]), lo, hi, np.array([0.0, 0.0]))  # strict inside
    assert orc.in_tangent_cone_multidim(np.array([1.2, 0.5]), lo, hi, np.array([-1.0, 0.0]))
    assert not orc.in_tangent_cone_multidim(np.array([1.2, 0.5]), lo, hi, np.array([+1.0, 0.0]))
    # corner: both components outside on the same side -> 45 degree rule
    assert not orc.in_tangent_cone_multidim(np.array([1.2, 1.2]), lo, hi, np.array([-1.0, -1.0]))
    assert orc.in_tangent_cone_multidim(np.array([1.2, 1.2]), lo, hi, np.array([-1.0, 0.2]))


# ---------------------------------------------------------------- two oracles agree
CASES = [("position", skills.position_skill, None, 3), ("pose", skills.pose_skill, None, 7),
         ("stack", skills.stack_skill, skills.STACK_OPTIONS, 7)]


@pytest.mark.parametrize("name,make,opts,ny", CASES)
@pytest.mark.parametrize("robot", ["iiwa", "ur5"])
def test_numpy_and_c_oracle_agree(name, make, opts, ny, robot, iiwa_fk, ur5_fk):
    from oracle.c_oracle import CPinvOracle
    fk = iiwa_fk if robot == "iiwa" else ur5_fk
    spec = make(fk)
    co = CPinvOracle(spec, opts)
    for dist in ("interior", "mixed"):
        Q, Y = skills.synthetic_inputs(fk, 96, seed=5, distribution=dist)
        Y = Y[:, :ny]
        ref, rmode = orc.pinv_solve_batch(spec, opts, 0.0, Q, Y=Y)
        dq, _, mode = co.solve_batch(0.0, Q, Y=Y)
        assert np.array_equal(mode, rmode)
        assert _rel(dq, ref).max() < 1e-8        # measured ~1e-9 on the [J;J] tall solve


def test_c_oracle_task_rows_match_ad(iiwa_fk):
    from oracle.c_oracle import CPinvOracle
    spec = skills.stack_skill(iiwa_fk)
    co = CPinvOracle(spec, skills.STACK_OPTIONS)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 4, seed=6, distribution="mixed")
    ev = orc.ExprEvaluator(spec, 0.0, Q, Y)
    for ti, c in enumerate(spec.constraints):
        e, Jt, J = ev.vector(c.expression)
        for b in range(len(Q)):
            e2, J2, Jt2 = co.task_eval(ti, 0.0, Q[b], Y[b])
            assert np.abs(e2 - e[b]).max() < 1e-13 and np.abs(J2 - J[b]).max() < 1e-13


def test_one_dim_sets_mode_scan(ur5_fk):
    """1-D joint-limit sets + a scalar distance task, the many-mode skill of
    ur5_transformation_matrix_comparison_of_controllers.ipynb cell 27.  The
    tool position does not depend on the last UR5 joint, so a limit on q[5]
    would make its tangent-cone test a tie decided by rounding noise; the sets
    are on joints 0-4 (32 modes)."""
    from oracle.c_oracle import CPinvOracle
    fk = ur5_fk
    t = cs.MX.sym("t")
    q = cs.MX.sym("q", 6)
    p = fk["T_fk"](q)[:3, 3]
    cons = [cc.EqualityConstraint("dist", cs.norm_2(np.array([0.5, 0.5, 0.5]) - p), gain=50.0,
                                  constraint_type="soft", priority=6)]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    for i in range(5):
        cons.append(cc.SetConstraint("limit_q_%d" % i, q[i], set_min=0.3 * lo[i], set_max=0.3 * hi[i],
                                     priority=i))
    spec = cc.SkillSpecification("point", t, q, constraints=cons)
    rng = np.random.default_rng(4)
    Q = rng.uniform(0.35 * lo, 0.35 * hi, size=(48, 6))
    ref, rmode = orc.pinv_solve_batch(spec, None, 0.0, Q)
    dq, _, mode = CPinvOracle(spec).solve_batch(0.0, Q)
    assert np.array_equal(mode, rmode)
    assert len(np.unique(rmode)) > 3              # several different modes are exercised
    assert _rel(dq, ref).max() < 1e-8


# ---------------------------------------------------------------- QP
def test_qp_solver_vs_enumeration():
    """Dual active-set answer == brute-force KKT enumeration on small QPs."""
    import itertools
    rng = np.random.default_rng(1)
    for trial in range(30):
        nv, nc = 4, 5
        hd = rng.uniform(0.01, 2.0, nv)
        A = rng.normal(size=(nc, nv))
        mid = A @ rng.normal(size=nv)             # feasible by construction
        lb, ub = mid - rng.uniform(0.0, 1, nc), mid + rng.uniform(0.0, 1, nc)
        lb[0] = ub[0] = mid[0]
        x = orc.qp_solve_dense(hd, A, lb, ub)
        best, best_cost = None, np.inf
        for pattern in itertools.product((0, 1, 2), repeat=nc):      # 0 free, 1 at lb, 2 at ub
            idx = [i for i, s in enumerate(pattern) if s]
            if len(idx) > nv:
                continue
            if idx:
                N = A[idx]
                b = np.array([lb[i] if pattern[i] == 1 else ub[i] for i in idx])
                K = np.block([[np.diag(hd), -N.T], [N, np.zeros((len(idx), len(idx)))]])
                try:
                    sol = np.linalg.solve(K, np.concatenate([np.zeros(nv), b]))
                except np.linalg.LinAlgError:
                    continue
                cand = sol[:nv]
            else:
                cand = np.zeros(nv)
            Ax = A @ cand
            if np.all(Ax >= lb - 1e-9) and np.all(Ax <= ub + 1e-9):
                cost = 0.5 * np.sum(hd * cand * cand)
                if cost < best_cost - 1e-12:
                    best, best_cost = cand, cost
        assert best is not None
        assert np.abs(x - best).max() < 1e-8


def test_qp_solver_reports_infeasible():
    hd = np.ones(2)
    A = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    with pytest.raises(orc.QPInfeasible):
        orc.qp_solve_dense(hd, A, np.array([1.0, -5.0, 0.0]), np.array([2.0, 0.0, 1.0]))
    with pytest.raises(orc.QPInfeasible):        # two contradicting equalities
        orc.qp_solve_dense(hd, A[:2], np.array([1.0, 2.0]), np.array([1.0, 2.0]))


def test_qp_rows_and_kkt(iiwa_fk):
    """reactive_qp.py:175-246: H = diag(mu w_rob, mu + w_slack), soft rows get
    -1 slack columns, bounds per constraint class; the answer satisfies KKT."""
    spec = skills.qp_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 16, seed=2)
    hd, A, lb, ub = orc.qp_data_batch(spec, 0.0, Q, Y=Y)
    assert hd.shape == (16, 13) and A.shape == (16, 13, 13)
    assert np.allclose(hd[:, :7], 1e-3) and np.allclose(hd[:, 7:], 1.001)
    # rows 0-6: VelocitySetConstraint (priority 0) on q: A = [I 0], bounds +-v_max
    assert np.allclose(A[:, :7, :7], np.eye(7)) and np.all(A[:, :7, 7:] == 0)
    assert np.allclose(ub[:, :7], np.array(iiwa_fk["velocity"])) and np.allclose(lb[:, :7], -ub[:, :7])
    # rows 7-12: soft pose equality: slack columns -I, lb == ub
    assert np.allclose(A[:, 7:, 7:], -np.eye(6)) and np.allclose(lb[:, 7:], ub[:, 7:])
    dq, _, slack, status = orc.qp_solve_batch(spec, 0.0, Q, Y=Y)
    assert (status == 0).all()
    for b in range(16):
        prim, stat, sign = orc.kkt_residuals(hd[b], A[b], lb[b], ub[b], np.concatenate([dq[b], slack[b]]))
        assert prim < 1e-9 and stat < 1e-9 and sign < 1e-9


def test_qp_data_c_oracle_matches(iiwa_fk):
    from oracle.c_oracle import qp_data_batch
    spec = skills.qp_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 8, seed=3)
    a = orc.qp_data_batch(spec, 0.0, Q, Y=Y)
    b = qp_data_batch(spec, 0.0, Q, Y=Y)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() < 1e-12


# ---------------------------------------------------------------- golden vectors
def test_oracle_reproduces_golden(iiwa_fk, ur5_fk):
    g = np.load(GOLDEN)
    for name, fk, make, opts in [("iiwa_position", iiwa_fk, skills.position_skill, None),
                                 ("iiwa_pose", iiwa_fk, skills.pose_skill, None),
                                 ("iiwa_stack", iiwa_fk, skills.stack_skill, skills.STACK_OPTIONS),
                                 ("ur5_stack", ur5_fk, skills.stack_skill, skills.STACK_OPTIONS)]:
        dq, mode = orc.pinv_solve_batch(make(fk), opts, 0.0, g[name + "_Q"], Y=g[name + "_Y"])
        assert np.array_equal(mode, g[name + "_mode"])
        assert _rel(dq, g[name + "_dq"]).max() < 1e-12
    dq, _, slack, status = orc.qp_solve_batch(skills.qp_skill(iiwa_fk), 0.0, g["iiwa_qp_Q"], Y=g["iiwa_qp_Y"])
    assert (status == 0).all()
    assert _rel(dq, g["iiwa_qp_dq"]).max() < 1e-11 and _rel(slack, g["iiwa_qp_slack"]).max() < 1e-11


def test_dual_quaternion_fk_matches_the_notebook_outputs(ur5_fk):
    """Known answers the reference notebooks print for urdf2casadi's dual-quaternion kinematics
    (ur5_dual_quaternion_comparison_of_controllers.ipynb cell 7: Q_fk(UR5_home); cell 33:
    dual_quaternion_revolute of the desired frame and the identity) - they pin layout, sign and the
    1/2 t (x) r convention of casclik_amd/geom.py - and agreement with the matrix kinematics."""
    from casclik_amd import numpy_geom
    home = [0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0]
    Q0 = ur5_fk["dual_quaternion_fk"](home).toarray().ravel()
    printed = np.array([-0.707107, -3.46237e-12, -3.46237e-12, 0.707107, -3.40946e-13, -0.28624, 0.421616, 3.21923e-13])
    assert np.abs(Q0 - printed).max() < 5e-7                      # printed with 6 significant digits
    T0 = numpy_geom.dual_quaternion_to_transformation_matrix(Q0)
    assert abs(np.linalg.norm(T0[:3, 3]) - 1.0192) < 5e-5         # "Distance to UR5_pome pos: 1.0192"
    assert np.allclose(numpy_geom.dual_quaternion_revolute([0.2, 0.2, 0.75], [0.0, 0.0, 0.0], [1, 0, 0], 0.0),
                       [0.0, 0.0, 0.0, 1.0, 0.1, 0.1, 0.375, 0.0], atol=1e-15)
    assert np.allclose(numpy_geom.dual_quaternion_revolute([0., 0., 0.], [0., 0., 0.], [1., 0., 0.], 0.0),
                       [0, 0, 0, 1, 0, 0, 0, 0], atol=0)
    rng = np.random.default_rng(0)
    for _ in range(20):
        qq = rng.uniform(-3.0, 3.0, 6)
        Q = ur5_fk["dual_quaternion_fk"](qq).toarray().ravel()
        assert abs(Q[:4].dot(Q[:4]) - 1.0) < 1e-14 and abs(Q[:4].dot(Q[4:])) < 1e-14       # unit dual quaternion
        assert np.abs(numpy_geom.dual_quaternion_to_transformation_matrix(Q) - ur5_fk["T_fk"](qq).toarray()).max() < 1e-14
        Qi = numpy_geom.dual_quaternion_inv(Q)
        assert np.abs(numpy_geom.dual_quaternion_product(Q, Qi) - [0, 0, 0, 1, 0, 0, 0, 0]).max() < 1e-14
        assert np.abs(numpy_geom.dual_quaternion_to_pos(Q) - ur5_fk["T_fk"](qq).toarray()[:3, 3]).max() < 1e-14


def test_denavit_hartenberg_chain_of_the_moe2016_notebook(ur5_fk):
    """converter.from_denavit_hartenberg with the notebook's classic UR5 DH table
    (ur5_moe2016_example2.ipynb cell 2).  Pins: the notebook starts "in the box" (cell 5: UR5_home, cell 7:
    box limits) - true for the classic DH frames, false for the URDF frames at the same joint angles - and
    the DH chain equals the URDF chain up to the UR5's known base rotation Rz(pi) and the rounding of the
    table (3 decimals)."""
    from casclik_amd import converter
    pi = np.pi
    fk = converter.from_denavit_hartenberg(
        joint_angles=["s" for _ in range(6)], link_lengths=[0., -0.425, -0.392, 0., 0., 0.],
        link_offsets=[0.089, 0., 0., 0.109, 0.095, 0.082], link_twists=[pi / 2, 0., 0., pi / 2, -pi / 2, 0.],
        joint_names=ur5_fk["joint_names"], upper_limits=ur5_fk["upper"], lower_limits=ur5_fk["lower"])
    assert fk["joint_names"] == ur5_fk["joint_names"] and fk["upper"] == ur5_fk["upper"]
    home = np.array([-(50.0 / 180.0) * pi, -(160.0 / 180.0) * pi, -(110.0 / 180.0) * pi, -(90.0 / 180.0) * pi,
                     -(90.0 / 180.0) * pi, 0.0])
    p = fk["T_fk"](home).toarray()[:3, 3]
    box_lo, box_hi = np.array([0.1, -0.5, -0.3]), np.array([0.6, 0.4, 0.25])
    assert (p > box_lo).all() and (p < box_hi).all()
    pu = ur5_fk["T_fk"](home).toarray()[:3, 3]
    assert not ((pu > box_lo).all() and (pu < box_hi).all())
    rng = np.random.default_rng(3)
    for _ in range(20):
        qq = rng.uniform(-3.0, 3.0, 6)
        T, Tu = fk["T_fk"](qq).toarray(), ur5_fk["T_fk"](qq).toarray()
        assert np.abs(T[:3, 3] - np.array([-1.0, -1.0, 1.0]) * Tu[:3, 3]).max() < 2e-3
        assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-12


def test_notebook_golden_vectors_are_the_oracles_answers(ur5_fk):
    """tests/golden/notebook_golden.npz (generated by make_golden.py) freezes the oracle's answers for the
    reference notebooks' own skills; an oracle edit that moves them shows up here."""
    from extern_skills import double_pendulum_skill, dual_quaternion_skill
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_golden.npz"))
    spec = double_pendulum_skill(True)
    dq, _, slack, status = orc.qp_solve_batch(spec, 1.3, g["pendulum_Q"], weights=orc.qp_weights(spec, [1.0, 1.0]))
    assert np.array_equal(status, g["pendulum_qp_status"])
    ok = status == 0
    assert np.abs(dq[ok] - g["pendulum_qp_dq"][ok]).max() < 1e-12 and np.abs(slack[ok] - g["pendulum_qp_slack"][ok]).max() < 1e-12
    dq, mode = orc.pinv_solve_batch(dual_quaternion_skill(ur5_fk, "Q_dist2", for_pinv=True), None, 0.0, g["ur5_Q"])
    assert np.array_equal(mode, g["dq_pinv_mode"]) and np.abs(dq - g["dq_pinv_dq"]).max() < 1e-12


@pytest.mark.parametrize("robot", ["iiwa", "ur5"])
@pytest.mark.parametrize("which", ["position", "pose", "stack", "qp"])
def test_independent_baseline_descriptors_equal_the_lowered_ones(robot, which, iiwa_fk, ur5_fk):
    """oracle/baseline_desc.py writes the flat descriptors of the BASELINE skills down from the URDF and SURVEY.md
    8(d) without the product's front-end; casclik_amd/lowering.py derives them from the skill scripts' expression
    graphs.  Two independent statements of the same skill must give the same struct, byte for byte - a check of the
    lowering, and the reason the C restatement fed with the former is a witness of more than the kernels' algebra."""
    from casclik_amd import _capi
    from casclik_amd.lowering import lower_skill
    from oracle import baseline_desc
    fk = iiwa_fk if robot == "iiwa" else ur5_fk
    mk = {"position": skills.position_skill, "pose": skills.pose_skill, "stack": skills.stack_skill,
          "qp": skills.qp_skill}[which]
    direct, dims = baseline_desc.baseline_descriptor(robot, which)
    lowered = _capi.desc_to_c(lower_skill(mk(fk)))
    assert bytes(direct) == bytes(lowered)
    assert dims[0] == len(fk["joint_names"])


def test_c_qp_port_equals_the_numpy_qp_oracle(iiwa_fk):
    """orc_qp_solve_batch (the compiled CPU baseline of the QP path) against oracle/clik_oracle.py::qp_solve_batch"""
    from oracle import c_oracle
    spec = skills.qp_skill(iiwa_fk)
    Q, Y = skills.synthetic_inputs(iiwa_fk, 96, seed=3, distribution="mixed")
    dq, _, slack, status = c_oracle.CQpOracle(None, baseline=("iiwa", "qp")).solve_batch(0.0, Q, Y=Y)
    rdq, _, rslack, rstatus = orc.qp_solve_batch(spec, 0.0, Q, Y=Y)
    assert np.array_equal(status, rstatus) and (status == 0).all()
    assert np.abs(dq - rdq).max() < 1e-10 and np.abs(slack - rslack).max() < 1e-10


def test_the_tolerance_rule_cannot_pass_without_comparing():
    """tests/tolerances.py: an all-ill-posed batch fails instead of passing vacuously, a device NaN on an
    instance the oracle solved fails, a row mask keeps every instance's own kappa, and the rule used is reported"""
    import tolerances as tol
    ref = np.array([[1.0, 2.0], [3.0, 4.0], [np.nan, np.nan], [5.0, 6.0]])
    good = ref.copy()
    kappa = np.array([1e3, 1e5, 1.0, 1e3])
    assert tol.qp_close(good, ref, kappa=kappa) and tol.LAST == {"rule": "kappa", "checked": 3, "left_out": 0, "worst": 0.0}
    off = good.copy(); off[1, 0] += 1e-9                      # beyond max(1e-12, 8 u 1e5 = 8.9e-11)
    assert not tol.qp_close(off, ref, kappa=kappa)
    assert tol.qp_close(off, ref, kappa=kappa, rows=np.array([True, False, True, True]))
    assert tol.qp_close(off, ref)                             # (no kappa: the flat ceiling 1e-8 - and it says so)
    assert tol.LAST["rule"] == "ceiling"
    nan_dev = good.copy(); nan_dev[0, 1] = np.nan
    assert not tol.qp_close(nan_dev, ref, kappa=kappa) and tol.LAST["rule"] == "nan"
    assert not tol.pinv_close(nan_dev, ref)
    assert tol.worst_over_tol(nan_dev, ref, kappa=kappa)[0] == float("inf")
    ill = np.full(4, 1e14)                                    # every bound beyond ILL_POSED: nothing would be compared
    assert not tol.pinv_close(good, ref, kappa=ill) and tol.LAST["checked"] == 0 and tol.LAST["left_out"] == 3
    some = np.array([1e3, 1e14, 1.0, 1e14])                   # two of three left out: more than MAX_LEFT_OUT
    assert not tol.pinv_close(good, ref, kappa=some)
    # the oracle's own array is recognised, a basic slice of it too; a fancy-indexed copy is not (use rows=)
    from oracle import clik_oracle
    from casclik_amd import skills
    fk = skills.iiwa()
    spec = skills.qp_skill(fk)
    Q, Y = skills.synthetic_inputs(fk, 6, seed=0)
    rdq, _, rslack, status = clik_oracle.qp_solve_batch(spec, 0.0, Q, Y=Y)
    assert tol.qp_close(rdq.copy(), rdq) and tol.LAST["rule"] == "kappa"
    assert tol.qp_close(rslack.copy(), rslack) and tol.LAST["rule"] == "kappa"
    ok = status == 0
    assert tol.qp_close(rdq.copy(), rdq, rows=ok) and tol.LAST["rule"] == "kappa"
    assert tol.qp_close(rdq[ok].copy(), rdq[ok]) and tol.LAST["rule"] == "ceiling"
