"""Property tests (hypothesis; CPU) of the invariants SURVEY.md section 8(c) lists for a path without reference outputs:

  * the damped pseudo-inverse of pseudo_inverse.py:92-105, through the controller's PUBLIC `pinv`, filters every singular
    direction by sigma / (sigma^2 + lam) - wide and tall matrices, any damping;
  * the mode scan (pseudo_inverse.py:512-556): the mode the oracle accepts is the FIRST mode in activation order whose
    velocity passes the tangent-cone test of every inactive SetConstraint - recomputed here from the oracle's per-mode
    velocities with the product's PUBLIC tangent-cone functions (themselves held to the reference's, test_refpins.py),
    for random 1-D sets, gains, targets and states;
  * every QP answer of the oracle is a KKT point of the reference's H, A, lbA, ubA (reactive_qp.py:175-246) and agrees
    with an independent solve (scipy SLSQP) of the same problem."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import casclik_amd as cc
from casclik_amd import sym as cs
from oracle import clik_oracle

SETTINGS = dict(max_examples=40, deadline=None)


@settings(**SETTINGS)
@given(rows=st.integers(1, 7), cols=st.integers(1, 7), lam=st.sampled_from([1e-7, 1e-4, 1e-2, 1.0]),
       seed=st.integers(0, 10 ** 6))
def test_damped_pinv_filters_every_singular_direction(rows, cols, lam, seed):
    rng = np.random.default_rng(seed)
    J = rng.normal(size=(rows, cols))
    t, q = cs.MX.sym("t"), cs.MX.sym("q", cols)
    spec = cc.SkillSpecification("s", t, q, constraints=[cc.EqualityConstraint(label="e", expression=q)])
    ctrl = cc.PseudoInverseController(skill_spec=spec, options={"damping_factor": lam})
    P = np.asarray(ctrl.pinv(cs.DM(J)).toarray())
    U, s, Vt = np.linalg.svd(J, full_matrices=False)
    want = (Vt.T * (s / (s * s + lam))).dot(U.T)
    assert P.shape == (cols, rows)
    assert np.abs(P - want).max() < 1e-9 * (1.0 + np.abs(want).max())


@settings(**SETTINGS)
@given(seed=st.integers(0, 10 ** 6), n_sets=st.integers(1, 3))
def test_accepted_mode_is_the_first_whose_inactive_sets_are_in_their_tangent_cones(seed, n_sets):
    rng = np.random.default_rng(seed)
    n = 4
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", n), cs.MX.sym("dq", n)
    A = rng.normal(size=(2, n))
    cons = [cc.EqualityConstraint(label="task", expression=cs.mtimes(A, q) - rng.normal(size=2) + 0.1 * cs.sin(t),
                                  gain=float(rng.uniform(0.5, 5.0)), priority=10)]
    sets = []
    for k in range(n_sets):
        row = rng.normal(size=n) if rng.random() < 0.5 else np.eye(n)[k]
        sets.append(cc.SetConstraint(label="set%d" % k, expression=cs.mtimes(row.reshape(1, n), q), set_min=-0.4,
                                     set_max=0.4, gain=float(rng.uniform(0.5, 3.0)), priority=k))
    spec = cc.SkillSpecification("s", t, q, dq, constraints=cons + sets)
    ctrl = cc.PseudoInverseController(skill_spec=spec)
    fns = [ctrl.get_in_tangent_cone_function(c) for c in spec.constraints if isinstance(c, cc.SetConstraint)]
    Q = rng.uniform(-0.7, 0.7, size=(24, n))
    t0 = float(rng.uniform(0, 2))
    dz, mode, allv = clik_oracle.pinv_solve_batch(spec, None, t0, Q, return_all_modes=True)
    amap = ctrl.activation_map
    for b in range(len(Q)):
        first = -1
        for m, active in enumerate(amap):
            ok = all(active[s] or float(np.asarray(fns[s](t0, Q[b], allv[b, m]).full()).ravel()[0]) != 0.0
                     for s in range(n_sets))
            if ok:
                first = m
                break
        assert first == mode[b], (b, first, mode[b])
        if first >= 0:
            assert np.array_equal(dz[b], allv[b, first])
        else:
            assert not dz[b].any()


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6))
def test_qp_answers_are_kkt_points_and_agree_with_an_independent_solver(seed):
    from scipy.optimize import minimize
    rng = np.random.default_rng(seed)
    n = 3
    t, q = cs.MX.sym("t"), cs.MX.sym("q", n)
    A = rng.normal(size=(2, n))
    cons = [cc.EqualityConstraint(label="task", expression=cs.mtimes(A, q) - rng.normal(size=2), gain=2.0,
                                  constraint_type="soft", slack_weight=float(rng.choice([1.0, 10.0]))),
            cc.VelocitySetConstraint(label="speed", expression=q, set_min=-np.full(n, 0.5), set_max=np.full(n, 0.5)),
            cc.SetConstraint(label="box", expression=q[0] + 0.3 * q[1], set_min=-0.3, set_max=0.3, gain=1.5)]
    spec = cc.SkillSpecification("s", t, q, constraints=cons)
    Q = rng.uniform(-0.5, 0.5, size=(6, n))
    H, Am, lb, ub = clik_oracle.qp_data_batch(spec, 0.0, Q)
    dq, _, slack, status = clik_oracle.qp_solve_batch(spec, 0.0, Q)
    for b in range(len(Q)):
        if status[b] != 0:
            continue
        x = np.concatenate([dq[b], slack[b]])
        infeasibility, stationarity, wrong_sign = clik_oracle.kkt_residuals(H[b], Am[b], lb[b], ub[b], x)
        assert infeasibility < 1e-8 and stationarity < 1e-7 and wrong_sign < 1e-7, (infeasibility, stationarity, wrong_sign)
        fun = lambda v: float((H[b] * v * v).sum())                                  # noqa: E731
        cons_ = [{"type": "ineq", "fun": lambda v, i=i: Am[b][i].dot(v) - lb[b][i]} for i in range(len(lb[b])) if lb[b][i] > -1e9]
        cons_ += [{"type": "ineq", "fun": lambda v, i=i: ub[b][i] - Am[b][i].dot(v)} for i in range(len(ub[b])) if ub[b][i] < 1e9]
        res = minimize(fun, np.zeros(len(x)), constraints=cons_, method="SLSQP", options={"ftol": 1e-14, "maxiter": 300})
        # (SLSQP sometimes stops with "positive directional derivative" AT the minimum: its point is judged by its rows
        # and its cost - no feasible point may undercut the oracle's, and SLSQP's own should not be worse by much)
        rows = Am[b].dot(res.x)
        feasible = (rows >= np.clip(lb[b], -1e9, 1e9) - 1e-7).all() and (rows <= np.clip(ub[b], -1e9, 1e9) + 1e-7).all()
        if feasible:
            assert fun(x) <= fun(res.x) + 1e-7 * (1.0 + fun(x)), (fun(res.x), fun(x))
        if res.success:
            assert abs(fun(res.x) - fun(x)) < 1e-6 * (1.0 + fun(x)), (fun(res.x), fun(x))
