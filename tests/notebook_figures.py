"""The simulations of tests/golden/figure_skills.py bound to the product's front-end, plus the UR5 runs (which need the
product's URDF kinematics)."""
import os
import sys

import numpy as np

import casclik_amd as cc
from casclik_amd import sym as cs

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import figure_skills                                    # noqa: E402
from figure_skills import (FIGS, CASES, PENDULUM_CASES, MAX_SPEED, MIN_P, MAX_P, N_TICKS, simulate, simulate_pendulum,   # noqa: E402,F401
                           deviation_in_pixels)


def build(case):
    return figure_skills.build(case, cs, cc)


def pendulum_skill(case):
    return figure_skills.pendulum_skill(case, cs, cc)


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
    elif which == "Q_dist2":
        # cell 18: the dual Hamilton operator "minus" of Q_des (cell 3) times the conjugation signs times Q_des - Q_fk(q)
        Q_des = np.asarray(numpy_geom.dual_quaternion_revolute(xyz, rpy, [1, 0, 0], 0.0), dtype=float).ravel()
        a = Q_des
        Hm = np.array([[a[3], a[2], -a[1], a[0], 0, 0, 0, 0], [-a[2], a[3], a[0], a[1], 0, 0, 0, 0],
                       [a[1], -a[0], a[3], a[2], 0, 0, 0, 0], [-a[0], -a[1], -a[2], a[3], 0, 0, 0, 0],
                       [a[7], a[6], -a[5], a[4], a[3], a[2], -a[1], a[0]], [-a[6], a[7], a[4], a[5], -a[2], a[3], a[0], a[1]],
                       [a[5], -a[4], a[7], a[6], a[1], -a[0], a[3], a[2]], [-a[4], -a[5], -a[6], a[7], -a[0], -a[1], -a[2], a[3]]])
        expr = cs.mtimes(Hm.dot(np.diag([-1.0, -1, -1, 1, -1, -1, -1, 1])), Q_des - fk["dual_quaternion_fk"](q))
        norm = cs.norm_2(expr)
    elif which in ("T_dist1", "T_dist2"):
        T_des = np.eye(4)
        T_des[:3, :3] = numpy_geom.rotation_rpy(*rpy)
        T_des[:3, 3] = xyz
        if which == "T_dist1":
            expr = cs.norm_fro(cs.mtimes(np.linalg.inv(T_des), fk["T_fk"](q)) - np.eye(4))
            norm = expr
        else:
            # cell 20: position deviation, and the Frobenius norm of the rotation's deviation from the desired one
            T = fk["T_fk"](q)
            expr = cs.vertcat(T[:3, 3] - T_des[:3, 3], cs.norm_fro(cs.mtimes(np.linalg.inv(T_des[:3, :3]), T[:3, :3]) - np.eye(3)))
            norm = cs.norm_2(expr)
    else:
        raise ValueError(which)
    error = cc.EqualityConstraint(label=which + "_cnstr", expression=expr, constraint_type="soft", gain=10.0,
                                  priority=300 if which == "T_dist2" else 301)
    if controller == "qp":
        cons = [cc.SetConstraint(label="Joint_Limits", expression=q, set_min=np.array(fk["lower"]),
                                 set_max=np.array(fk["upper"])), error]
        options = None
    else:
        cons = [error]
        options = {"multidim_sets": True, "pinv_method": "damped", "damping_factor": 1e-26}
    spec = cc.SkillSpecification(label=which + "_skill", time_var=t, robot_var=q, constraints=cons)
    return spec, options, cs.Function("e", [t, q], [norm])


def simulate_frame_error(eval_norm, solve, return_q=False):
    """cell 25: 1000 ticks of 0.008 s from UR5_home, speeds saturated at pi / 5; e_sim[i + 1] is the error norm at
    q_sim[i + 1].  Returns t_sim, log10(e_sim) (and q_sim with `return_q`)."""
    n, dt, max_speed = 1001, 0.008, np.pi / 5
    t_sim = np.array([dt * i for i in range(n)])
    q = UR5_HOME.copy()
    e_sim, q_sim = np.zeros(n), np.zeros((n, 6))
    q_sim[0] = q
    e_sim[0] = float(np.asarray(eval_norm(0.0, q).toarray()).ravel()[0])
    for i in range(n - 1):
        q = q + np.clip(solve(t_sim[i], q), -max_speed, max_speed) * dt
        q_sim[i + 1] = q
        e_sim[i + 1] = float(np.asarray(eval_norm(t_sim[i], q).toarray()).ravel()[0])
    if return_q:
        return t_sim, np.log10(np.maximum(e_sim, 1e-300)), q_sim
    return t_sim, np.log10(np.maximum(e_sim, 1e-300))


# ---- UR5, Moe-2016 example 2 (ur5_moe2016_example2.ipynb): pinv with 8 modes / an active multidimensional set; QP ------
from figure_skills import (MOE_CASES, MOE_DT, MOE_TICKS, MOE_HOME, moe_path, simulate_moe, interval_deviation,   # noqa: E402,F401
                           fill_deviation, moe_pins)


def moe_fk():
    """cell 2: joint names and limits from the URDF, kinematics from the UR5's Denavit-Hartenberg table"""
    from casclik_amd import skills
    from casclik_amd.urdf import converter
    fk = skills.ur5()
    return converter.from_denavit_hartenberg(joint_angles=["s"] * 6, joint_names=fk["joint_names"],
                                             upper_limits=fk["upper"], lower_limits=fk["lower"], **figure_skills.MOE_DH)


def moe_skill(fk, situation):
    return figure_skills.moe_skill(situation, cs, cc, fk["T_fk"])[0]


def moe_options(case):
    return {"multidim_sets": True} if case == "pinv_multidim" else None            # cell 11


# ---- ur5_dual_quaternion_comparison_of_controllers.ipynb: error norms on log axes (cells 10-17, 33-39; figures 19, 20, 41, 42)
DQC_CASES = [(which, kind) for which in ("cart_dist", "quat_dist", "Q_dist1", "Q_dist2") for kind in ("pinv", "qp")]


def dqc_skill(fk, which, kind):
    """cells 10-14 / 33-38: the constraint `which` (gain 1, soft; cart_dist / quat_dist with priority 300) in front of six
    1-D joint-limit sets (pinv) or beside the multidimensional limits and the joint-speed limits (QP).
    -> (spec, error norm as a Function of (t, q))"""
    from casclik_amd import numpy_geom, casadi_geom
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    Q_fk = fk["dual_quaternion_fk"]
    q1, q2 = cs.SX.sym("q1", 8), cs.SX.sym("q2", 8)
    dual_quaternion_product = cs.Function("dualquatprod", [q1, q2], [casadi_geom.dual_quaternion_product(q1, q2)])
    dual_quaternion_conj = cs.Function("dualquatconj", [q1], [casadi_geom.dual_quaternion_conj(q1)])
    p1, p2 = cs.SX.sym("p1", 4), cs.SX.sym("p2", 4)
    quaternion_product = cs.Function("quatprod", [p1, p2], [casadi_geom.quaternion_product(p1, p2)])
    quaternion_conj = cs.Function("quatconj", [p1], [casadi_geom.quaternion_conj(p1)])
    Q_des = numpy_geom.dual_quaternion_revolute([0.2, 0.2, 0.75], [0.0, 0.0, 0.0], [1, 0, 0], 0.0)            # cell 33
    Q_id = numpy_geom.dual_quaternion_revolute([0., 0., 0.], [0., 0., 0.], [1., 0., 0.], 0.0)
    Q_r, Q_d = Q_fk(q)[:4], Q_fk(q)[4:8]
    kw = {}
    if which == "cart_dist":
        expr, kw = 2 * quaternion_product(Q_d, quaternion_conj(Q_r))[:3], dict(priority=300)                  # cell 12
    elif which == "quat_dist":
        expr, kw = Q_d[:3] - 0.5 * quaternion_product(np.array([0.5, 0.5, 0.5, 0.0]), Q_r)[:3], dict(priority=300)
    elif which == "Q_dist1":
        expr = dual_quaternion_product(Q_fk(q), dual_quaternion_conj(Q_des)) - Q_id                           # cell 34
    else:
        expr = dual_quaternion_product(dual_quaternion_conj(Q_des - Q_fk(q)), Q_des)
    task = cc.EqualityConstraint(label=which, expression=expr, gain=1.0, constraint_type="soft", **kw)
    lo, hi = np.array(fk["lower"]), np.array(fk["upper"])
    max_speed = np.pi / 5
    if kind == "pinv":
        cons = [task] + [cc.SetConstraint(label="limit_q_%d" % i, expression=q[i], set_min=lo[i], set_max=hi[i], priority=i)
                         for i in range(6)]
    else:
        cons = [task, cc.SetConstraint(label="Joint_Limits", expression=q, set_min=lo, set_max=hi),
                cc.VelocitySetConstraint(label="Joint_speed_limits", expression=q, set_min=-np.full(6, max_speed),
                                         set_max=np.full(6, max_speed))]
    spec = cc.SkillSpecification(label=which, time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons)
    return spec, cs.Function("e", [t, q], [cs.norm_2(expr)])


def simulate_dqc(eval_norm, solve, n_ticks=4500, return_q=False):
    """cells 17 / 39: 4500 ticks of 0.01 s from UR5_home, speeds saturated at pi / 5; e_sim[i + 1] is the error norm at
    q_sim[i + 1].  Returns t_sim, log10(e_sim) (and q_sim [n_ticks + 1, 6] with `return_q`)."""
    dt, max_speed = 0.01, np.pi / 5
    t_sim = np.array([dt * i for i in range(n_ticks + 1)])
    q = UR5_HOME.copy()
    e_sim = np.zeros(n_ticks + 1)
    q_sim = np.zeros((n_ticks + 1, len(q)))
    q_sim[0] = q
    e_sim[0] = float(np.asarray(eval_norm(0.0, q).toarray()).ravel()[0])
    for i in range(n_ticks):
        q = q + np.clip(solve(t_sim[i], q), -max_speed, max_speed) * dt
        q_sim[i + 1] = q
        e_sim[i + 1] = float(np.asarray(eval_norm(t_sim[i], q).toarray()).ravel()[0])
    if return_q:
        return t_sim, np.log10(np.maximum(e_sim, 1e-300)), q_sim
    return t_sim, np.log10(np.maximum(e_sim, 1e-300))


DQC_FLOOR = -12.5      # log10: below, the stored curves are the rounding floor of CasADi's arithmetic, not the controllers


def dqc_pins(which, kind, t_sim, log_e):
    """[(pin, worst deviation in pixels, columns, t)]: the controller's own colour where it is visible, and the band of
    coloured pixels it otherwise hides under; columns whose stored values lie below 1e-12.5 are left out"""
    out = []
    for key in ("dqc_%s_%s" % (which, kind), "dqc_%s_union" % which):
        keep = (FIGS[key + "_lo"] >= DQC_FLOOR)
        ft = FIGS[key + "_t"]
        ok_cols = np.array([tk for tk in np.unique(ft) if keep[ft == tk].all()])
        if len(ok_cols) == 0:
            continue
        # columns are contiguous in t from the start until the curves sink below the floor (Q_dist2's pinv never does)
        out.append((key,) + interval_deviation(key, t_sim, log_e, within=(0.0, float(ok_cols.max()))))
    return out


# ---- ... and its frame_3d figures (cells 22-27, 44-48): the tool's path and the tips of its frame's axes in 3-D ----------
FRAME_PIXELS = 2.0       # a stored pixel of a curve lies within this of the simulated curve (line half-width 1.04 px + 1)
FRAME_COVERED = 0.9      # ... and this share of the simulated curve's length has ink within 1.5 px


def dqc_frame_pins(fk, which, kind, q_sim):
    """{colour: (worst distance of a stored pixel from the simulated curve [px], share of the simulated curve drawn,
    pixels compared)} for the figure of that run; matplotlib's projection and autoscaled view limits are restated in
    tests/golden/frame3d_pins.py"""
    import frame3d_pins
    T = np.array([fk["chain"].fk_numeric(q) for q in q_sim])
    axes, curves, dots = frame3d_pins.axes_of(T[:, :3, 3], T[:, :3, :3], **frame3d_pins.dqc_target(which))
    return frame3d_pins.deviations(frame3d_pins.stored_frames(FIGS, which, kind), axes, curves, dots)



def frame_error_frame_pins(fk, which, q_sim):
    """the same for ur5_dual_quaternion_vs_transformation_matrix.ipynb cells 33 / 34 (the QP's runs on Q_dist2 / T_dist2;
    350 x 216 canvas, the view limits the cells set: one pixel = 7 mm)"""
    import frame3d_pins
    T = np.array([fk["chain"].fk_numeric(q) for q in q_sim])
    axes, curves, dots = frame3d_pins.axes_of(T[:, :3, 3], T[:, :3, :3], T_des=frame3d_pins.dqtm_target(),
                                              limits=frame3d_pins.DQTM_LIMITS, width=frame3d_pins.DQTM_CANVAS[0],
                                              height=frame3d_pins.DQTM_CANVAS[1])
    return frame3d_pins.deviations(frame3d_pins.stored_frames(FIGS, which, "qp", prefix="f3d_dqtm_"), axes, curves, dots)


def ur5_qp_point_skill(fk):
    """ur5_transformation_matrix_comparison_of_controllers.ipynb cells 8, 9: the position error as three rows (gain 50),
    the multidimensional joint limits and the joint-speed limits"""
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    max_speed = np.pi / 5
    cons = [cc.EqualityConstraint(label="Minimize_point_error", expression=np.array([0.5, 0.5, 0.5]) - fk["T_fk"](q)[:3, 3],
                                  gain=50.0, constraint_type="soft"),
            cc.SetConstraint(label="Joint_Limits", expression=q, set_min=np.array(fk["lower"]), set_max=np.array(fk["upper"])),
            cc.VelocitySetConstraint(label="Joint_speed_limits", expression=q, set_min=-np.full(6, max_speed),
                                     set_max=np.full(6, max_speed))]
    return cc.SkillSpecification(label="Move to point", time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons)


def simulate_ur5_joints(solve, clamp, return_dq=False):
    """cells 14 / 29: 1000 samples of 0.01 s from UR5_home (the pinv loop saturates the speeds at pi / 5, the QP's
    carries them as rows).  Returns q_sim [1000, 6] (and dq_sim as the notebooks keep it: the applied speeds, the last
    row left at zero)."""
    n, dt, max_speed = 1000, 0.01, np.pi / 5
    q_sim, dq_sim = np.zeros((n, 6)), np.zeros((n, 6))
    q_sim[0] = UR5_HOME
    for i in range(n - 1):
        dq = solve(dt * i, q_sim[i])
        dq_sim[i] = np.clip(dq, -max_speed, max_speed) if clamp else dq
        q_sim[i + 1] = q_sim[i] + dq_sim[i] * dt
    return (q_sim, dq_sim) if return_dq else q_sim


def ur5_point_joint_pins(q_sim, dq_sim):
    """cell 31 of that notebook (`common_plots.joints` of the pinv point run): {"q": [...], "dq": [...]} per joint (worst
    distance of a stored pixel of the joint's colour from the simulated curve, share of the curve under ink, pixels);
    the legend hides the top axes from column 328 on"""
    import frame3d_pins as f3
    t = 0.01 * np.arange(len(q_sim))
    frames = FIGS["j2d_tm_pinv_frames"]
    out = {}
    for a, (name, values) in enumerate((("q", q_sim), ("dq", dq_sim))):
        stored = [FIGS["j2d_tm_pinv_%s_%d" % (name, k)].astype(float) + 0.5 for k in range(6)]
        out[name] = f3.deviations_2d(stored, frames[a], t, values, last_column=f3.TM_JOINTS_LEGEND_FROM if a == 0 else None)
    return out


def ur5_point_frame_pins(fk, kind, q_sim):
    """cells 17 / 33 of that notebook: the frame_3d figure of the point run of `kind` ("qp" / "pinv"); the crop offset of
    the stored image from the black dot at p_des (tests/golden/frame3d_pins.py)"""
    import frame3d_pins as f3
    T = np.array([fk["chain"].fk_numeric(q) for q in q_sim])
    axes, curves, dots = f3.axes_of(T[:, :3, 3], T[:, :3, :3], p_des=f3.TM_P_DES, limits=f3.TM_LIMITS,
                                    width=f3.TM_CANVAS[0], height=f3.TM_CANVAS[1])
    offset = FIGS["f3d_tm_point_%s_dot" % kind] - axes.pixels(dots["k"])[0]
    assert 7.2 <= offset[0] <= 14.0 and 7.2 <= offset[1] <= 14.0, offset       # (the pad, plus what the labels overhang)
    stored = {c: v - offset for c, v in f3.stored_frames(FIGS, "point", kind, prefix="f3d_tm_").items()}
    return f3.deviations(stored, axes, curves, dots, dot_radius=5.0)

