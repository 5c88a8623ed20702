"""The closed-loop simulations whose FIGURES the reference's notebooks store, rebuilt with the product's front-end, and
the comparison of a simulated curve with the samples digitised from those figures (tests/golden/make_figure_pins.py
-> notebook_figures.npz):
  cart_on_track_1D_comparison_of_controllers.ipynb   cells 6-11, 18-22, 31-36, 50-54, 56-61, 75-78  (QP and pinv)
  double_pendulum_2D_comparison_of_controllers.ipynb  cells 7-19 (QP to a point above the table), 31-38 (QP on a circle)
  ur5_transformation_matrix_comparison_of_controllers.ipynb  cells 2-9, 27-32 (pinv, six 1-D limit sets, UR5)"""
import os

import numpy as np

import casclik_amd as cc
from casclik_amd import sym as cs

HERE = os.path.dirname(os.path.abspath(__file__))
FIGS = np.load(os.path.join(HERE, "golden", "notebook_figures.npz"))
MAX_SPEED, MIN_P, MAX_P = 0.275, 0.0, 1.0          # cell 6
MAX_DX, MIN_DX = 1.1, 0.0                            # cell 56
N_TICKS = 1200
CASES = ["qp_point", "pinv_point", "qp_traj", "pinv_traj", "qp_path", "pinv_path"]


def build(case):
    """-> (controller kind, SkillSpecification, dt, p0, has virtual variable)"""
    t, p, dp = cs.MX.sym("t"), cs.MX.sym("p"), cs.MX.sym("dp")
    kind, task = case.split("_")
    x = dx = None
    if task == "point":
        target, dt, p0 = 0.75, 0.01, 0.0
    elif task == "traj":
        target, dt, p0 = 0.4 * cs.sin(0.3 * t), 0.02, 0.0001
    else:
        x, dx = cs.MX.sym("x"), cs.MX.sym("dx")
        target, dt, p0 = 0.4 * cs.sin(0.3 * x), 0.02, 0.0001
    limits = dict(label="cart_limit_cnstr", expression=p, gain=1.0, set_min=MIN_P, set_max=MAX_P)
    if kind == "qp":
        cons = [cc.EqualityConstraint(label="min_dist_cnstr", expression=target - p, gain=1.0, constraint_type="soft",
                                      priority=1),
                cc.SetConstraint(**limits),
                cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=p, gain=10.0, set_min=-MAX_SPEED,
                                         set_max=MAX_SPEED)]
        if task == "path":
            cons = [cc.EqualityConstraint(label="move_up_path_cnstr", expression=300 - x, gain=1.0,
                                          constraint_type="soft"),
                    cc.VelocitySetConstraint(label="slow_path_cnstr", expression=x, set_min=MIN_DX, set_max=MAX_DX)] + cons
    else:
        # the pinv controller has no VelocitySetConstraints: the notebook re-prioritises and saturates in the loop
        if task == "path":
            cons = [cc.EqualityConstraint(label="move_up_path_cnstr", expression=300 - x, gain=1.0, priority=1),
                    cc.EqualityConstraint(label="min_dist_cnstr", expression=target - p, gain=1.0,
                                          constraint_type="soft", priority=3),
                    cc.SetConstraint(priority=1, **limits)]
        else:
            cons = [cc.EqualityConstraint(label="min_dist_cnstr", expression=target - p, gain=1.0,
                                          constraint_type="soft", priority=2),
                    cc.SetConstraint(priority=1, **limits)]
    kw = dict(virtual_var=x, virtual_vel_var=dx) if x is not None else {}
    spec = cc.SkillSpecification(label=case, time_var=t, robot_var=p, robot_vel_var=dp, constraints=cons, **kw)
    return kind, spec, dt, p0, x is not None


def simulate(case, solve):
    """the notebook's loop (explicit Euler; the pinv runs saturate the speeds): `solve(t, p, x | None)` ->
    (dp, dx | None).  Returns t_sim, p_sim, dp_sim."""
    kind, _, dt, p0, virt = build(case)
    t_sim = np.array([dt * i for i in range(N_TICKS)])
    p_sim, dp_sim, x_sim = np.zeros(N_TICKS), np.zeros(N_TICKS), np.zeros(N_TICKS)
    p_sim[0] = p0
    for i in range(N_TICKS - 1):
        v, w = solve(t_sim[i], p_sim[i], x_sim[i] if virt else None)
        if kind == "pinv":
            v = max(min(v, MAX_SPEED), -MAX_SPEED)
            if virt:
                w = max(min(w, MAX_DX), MIN_DX)
        dp_sim[i] = v
        p_sim[i + 1] = p_sim[i] + v * dt
        if virt:
            x_sim[i + 1] = x_sim[i] + w * dt
    return t_sim, p_sim, dp_sim


def deviation_in_pixels(case, curve, t_sim, values, above=None):
    """For every sample digitised from the stored figure: how far (in pixel rows) the simulated curve misses it, where
    the simulated curve may be taken anywhere within a pixel and a half in t (a figure cannot place a jump more
    exactly).  Returns (worst deviation, samples)."""
    key = "%s_%s" % (case, curve)
    ft, fv = FIGS[key + "_t"], FIGS[key + "_v"]
    px_t, px_v = FIGS[key + "_pixel"]
    worst = 0.0
    if above is not None:
        ft, fv = ft[fv > above], fv[fv > above]
    for tk, vk in zip(ft, fv):
        near = (t_sim >= tk - 1.5 * px_t) & (t_sim <= tk + 1.5 * px_t)
        lo, hi = values[near].min(), values[near].max()
        miss = max(lo - vk, vk - hi, 0.0) / px_v
        worst = max(worst, miss)
    return worst, len(ft)


# ---- double pendulum (QP with the table SetConstraints: general inequality rows) ---------------------------------
PENDULUM_CASES = ["pend_point", "pend_track"]


def simulate_pendulum(case, solve):
    """cells 16 / 36: `solve(t, q)` -> dq [2].  Returns t_sim, q_sim, dq_sim, p_sim (sample 0 of p_sim stays zero, as
    in the notebook: the figures show that stroke)."""
    n = 800 if case == "pend_point" else 2000
    dt = 0.01
    t_sim = np.array([dt * i for i in range(n)])
    q_sim, dq_sim, p_sim = np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 2))
    q_sim[0] = [np.pi / 2 - 1e-5, 0.0]
    for i in range(n - 1):
        dq_sim[i] = solve(t_sim[i], q_sim[i])
        q_sim[i + 1] = q_sim[i] + dq_sim[i] * dt
        a, b = q_sim[i + 1]
        p_sim[i + 1] = [np.cos(a) + 0.75 * np.cos(a + b), np.sin(a) + 0.75 * np.sin(a + b)]
    return t_sim, q_sim, dq_sim, p_sim


# ---- UR5 move-to-point, PseudoInverseController --------------------------------------------------------------------
UR5_HOME = np.array([0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0])


def ur5_pinv_point_skill(fk):
    """cells 8, 27: the distance to (0.5, 0.5, 0.5) as ONE norm_2 row, least priority, behind a 1-D SetConstraint per
    joint limit"""
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    p_fk = fk["T_fk"](q)[:3, 3]
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    cons = [cc.EqualityConstraint(label="Minimize_point_error", expression=cs.norm_2(np.array([0.5, 0.5, 0.5]) - p_fk),
                                  gain=50.0, constraint_type="soft", priority=6)]
    cons += [cc.SetConstraint(label="limit_q_%d" % i, expression=q[i], set_min=lo[i], set_max=hi[i], priority=i)
             for i in range(6)]
    return cc.SkillSpecification(label="point_skill_pinv", time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons)


def simulate_ur5(fk, solve):
    """cell 29: 1000 ticks of 0.01 s from UR5_home, joint speeds saturated at pi / 5.  Returns t_sim, p_sim."""
    n, dt, max_speed = 1000, 0.01, np.pi / 5
    t_sim = np.array([dt * i for i in range(n)])
    q_sim, p_sim = np.zeros((n, 6)), np.zeros((n, 3))
    q_sim[0] = UR5_HOME
    p_sim[0] = fk["chain"].fk_numeric(UR5_HOME)[:3, 3]
    for i in range(n - 1):
        dq = np.clip(solve(t_sim[i], q_sim[i]), -max_speed, max_speed)
        q_sim[i + 1] = q_sim[i] + dq * dt
        p_sim[i + 1] = fk["chain"].fk_numeric(q_sim[i + 1])[:3, 3]
    return t_sim, p_sim


# ---- UR5 with a simulated input at the end effector, ReactiveQPController ---------------------------------------------
def ur5_input_skill(fk):
    """ur5_input_experiment.ipynb cells 7-13: the tool position follows `T_des[:3, 3] - y` (soft, gain 1) under the
    multidimensional joint limits and the joint-speed limits; y is the skill's input_var"""
    t, q, dq, y = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6), cs.MX.sym("y", 3)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    max_speed = np.pi / 5
    cons = [cc.EqualityConstraint(label="transfmat_dist_id", expression=fk["T_fk"](q)[:3, 3] - np.array([0.5, 0.5, 0.5]) + y,
                                  constraint_type="soft", gain=1.0, priority=100),
            cc.SetConstraint(label="Joint_Limits", expression=q, set_min=lo, set_max=hi),
            cc.VelocitySetConstraint(label="Joint_speed_limits", expression=q, set_min=-np.full(6, max_speed),
                                     set_max=np.full(6, max_speed))]
    return cc.SkillSpecification(label="linear_input_skill", time_var=t, robot_var=q, robot_vel_var=dq, input_var=y,
                                 constraints=cons)


def ur5_input_signal(n=4501, dt=0.01):
    """cell 16: nothing for 10 s, then 0.1 (cos(0.1 (t - 10)), sin(0.1 (t - 10 + pi / 2)), 0)"""
    y = np.zeros((n, 3))
    for i in range(n - 1):
        if dt * i > 10.0:
            y[i] = [0.1 * np.cos(0.1 * (dt * i - 10)), 0.1 * np.sin(0.1 * (dt * i - 10 + np.pi / 2)), 0.0]
    return y


def simulate_ur5_input(fk, solve):
    """cell 16: 4500 ticks of 0.01 s from UR5_home; `solve(t, q, y)` -> dq [6] (clamped like the notebook does)."""
    n, dt, max_speed = 4501, 0.01, np.pi / 5
    y_sim = ur5_input_signal(n, dt)
    t_sim = np.array([dt * i for i in range(n)])
    q_sim, p_sim = np.zeros((n, 6)), np.zeros((n, 3))
    q_sim[0] = UR5_HOME
    p_sim[0] = fk["chain"].fk_numeric(UR5_HOME)[:3, 3]
    for i in range(n - 1):
        dq = np.clip(solve(t_sim[i], q_sim[i], y_sim[i]), -max_speed, max_speed)
        q_sim[i + 1] = q_sim[i] + dq * dt
        p_sim[i + 1] = fk["chain"].fk_numeric(q_sim[i + 1])[:3, 3]
    return t_sim, p_sim


# ---- UR5 from home to a frame: error norms on a log axis --------------------------------------------------------------
def frame_error_skill(fk, which, controller):
    """ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 14-24: the tool frame's deviation from a desired frame
    (5 degrees of roll at (0.5, 0, 0.5)), gain 10, soft.  QP: behind the multidimensional joint limits (no speed-limit
    constraint: the loop saturates); pinv: the error alone, options multidim_sets / damped / damping_factor 1e-26.
    Returns (spec, options, error norm as a function of q)."""
    from casclik_amd import numpy_geom, casadi_geom
    t, q = cs.MX.sym("t"), cs.MX.sym("q", 6)
    rpy, xyz = [5.0 * (np.pi / 180.0), 0.0, 0.0], [0.5, 0.0, 0.5]
    if which == "Q_dist1":
        q1, q2 = cs.SX.sym("q1", 8), cs.SX.sym("q2", 8)
        product = cs.Function("dualquatprod", [q1, q2], [casadi_geom.dual_quaternion_product(q1, q2)])
        conj = cs.Function("dualquatconj", [q1], [casadi_geom.dual_quaternion_conj(q1)])
        Q_des = numpy_geom.dual_quaternion_revolute(xyz, rpy, [1, 0, 0], 0.0)
        Q_id = numpy_geom.dual_quaternion_revolute([0., 0., 0.], [0., 0., 0.], [1., 0., 0.], 0.0)
        expr = product(fk["dual_quaternion_fk"](q), conj(Q_des)) - Q_id
        norm = cs.norm_2(expr)
    else:
        raise ValueError(which)
    error = cc.EqualityConstraint(label=which + "_cnstr", expression=expr, constraint_type="soft", gain=10.0, priority=301)
    if controller == "qp":
        cons = [cc.SetConstraint(label="Joint_Limits", expression=q, set_min=np.array(fk["lower"]),
                                 set_max=np.array(fk["upper"])), error]
        options = None
    else:
        cons = [error]
        options = {"multidim_sets": True, "pinv_method": "damped", "damping_factor": 1e-26}
    spec = cc.SkillSpecification(label=which + "_skill", time_var=t, robot_var=q, constraints=cons)
    return spec, options, cs.Function("e", [t, q], [norm])


def simulate_frame_error(eval_norm, solve):
    """cell 25: 1000 ticks of 0.008 s from UR5_home, speeds saturated at pi / 5; e_sim[i + 1] is the error norm at
    q_sim[i + 1].  Returns t_sim, log10(e_sim)."""
    n, dt, max_speed = 1001, 0.008, np.pi / 5
    t_sim = np.array([dt * i for i in range(n)])
    q = UR5_HOME.copy()
    e_sim = np.zeros(n)
    e_sim[0] = float(np.asarray(eval_norm(0.0, q).toarray()).ravel()[0])
    for i in range(n - 1):
        q = q + np.clip(solve(t_sim[i], q), -max_speed, max_speed) * dt
        e_sim[i + 1] = float(np.asarray(eval_norm(t_sim[i], q).toarray()).ravel()[0])
    return t_sim, np.log10(np.maximum(e_sim, 1e-300))
