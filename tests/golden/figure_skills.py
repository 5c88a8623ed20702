"""The closed-loop simulations whose FIGURES the reference's notebooks store - skills parametrised by the symbolic module
and the constraint classes (the product's front-end in the tests; the REFERENCE package over the stand-in casadi in
tests/golden/make_ref_golden.py) - and the comparison of a simulated curve with the samples digitised from those figures
(tests/golden/make_figure_pins.py -> notebook_figures.npz).  Imports nothing of the product:
  cart_on_track_1D_comparison_of_controllers.ipynb   cells 6-11, 18-22, 31-36, 50-54, 56-61, 75-78  (QP and pinv)
  double_pendulum_2D_comparison_of_controllers.ipynb  cells 7-19 (QP to a point above the table), 31-38 (QP on a circle)
  ur5_transformation_matrix_comparison_of_controllers.ipynb  cells 2-9, 27-32 (pinv, six 1-D limit sets, UR5)"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIGS = np.load(os.path.join(HERE, "notebook_figures.npz"))
MAX_SPEED, MIN_P, MAX_P = 0.275, 0.0, 1.0          # cell 6
MAX_DX, MIN_DX = 1.1, 0.0                            # cell 56
N_TICKS = 1200
CASES = ["qp_point", "pinv_point", "qp_traj", "pinv_traj", "qp_path", "pinv_path"]


def build(case, cs, cc):
    """-> (controller kind, SkillSpecification, dt, p0, has virtual variable); `cs`, `cc`: the symbolic module and the
    constraint / skill classes to build with (the product's, or the reference's over the stand-in casadi)"""
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
    kind, task = case.split("_")
    dt, p0 = (0.01, 0.0) if task == "point" else (0.02, 0.0001)
    virt = task == "path"
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


def pendulum_skill(case, cs, cc):
    """double_pendulum_2D_comparison_of_controllers.ipynb cells 3-11 (to the point (0.75, 0.5)) and 31-33 (the circle):
    a soft 2-row tool-position task, joint-speed limits, and the table as two SetConstraints on task-space heights"""
    l_1, l_2, table_height = 1.0, 0.75, -0.05
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 2), cs.MX.sym("dq", 2)
    p_mid = cs.vertcat(l_1 * cs.cos(q[0]), l_1 * cs.sin(q[0]))
    p = cs.vertcat(l_1 * cs.cos(q[0]) + l_2 * cs.cos(q[0] + q[1]),
                   l_1 * cs.sin(q[0]) + l_2 * cs.sin(q[0] + q[1]))
    if case == "pend_track":
        p_des = 0.25 * cs.vertcat(cs.cos(0.5 * t), cs.sin(0.5 * t)) + cs.vertcat(1.0, 1.0)
    else:
        p_des = cs.vertcat(0.75, 0.5)
    cons = [cc.EqualityConstraint(label="min_dist_cnstr", expression=p_des - p, gain=1.0, constraint_type="soft"),
            cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=q, set_min=-cs.vertcat(0.5, 0.5),
                                     set_max=cs.vertcat(0.5, 0.5)),
            cc.SetConstraint(label="table_midpoint_cnstr", expression=p_mid[1] - table_height, set_min=0.0,
                             set_max=cs.inf),
            cc.SetConstraint(label="table_endpoint_cnstr", expression=p[1] - table_height, set_min=0.0,
                             set_max=cs.inf)]
    return cc.SkillSpecification(label="move_to_point_skill", time_var=t, robot_var=q, robot_vel_var=dq,
                                 constraints=cons)


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




# ---- UR5, Moe-2016 example 2: a trajectory that leaves a box (pinv with 8 modes / with a multidimensional set; QP) ---
MOE_DT, MOE_TICKS, MOE_MAX_SPEED = 0.008, 10000, np.pi / 5                  # ur5_moe2016_example2.ipynb cells 5, 12
MOE_HOME = np.array([-50.0, -160.0, -110.0, -90.0, -90.0, 0.0]) / 180.0 * np.pi
MOE_WALLS = np.array([[0.1, 0.6], [-0.5, 0.4], [-0.3, 0.25]])             # cell 7: x, y, z (min, max)
MOE_DH = dict(link_lengths=[0., -0.425, -0.392, 0., 0., 0.], link_twists=[np.pi / 2, 0., 0., np.pi / 2, -np.pi / 2, 0.],
              link_offsets=[0.089, 0., 0., 0.109, 0.095, 0.082])           # cell 2: the UR5's classic DH table
MOE_CASES = ["pinv_singular", "pinv_multidim", "qp_singular", "qp_multidim"]


def moe_path(t, np_or_cs=np):
    """cell 7: the desired tool trajectory (omega = 0.1)"""
    m = np_or_cs
    return [0.5 * m.sin(0.1 * t) * m.sin(0.1 * t) + 0.2, 0.5 * m.cos(0.1 * t) + 0.25 * m.sin(0.1 * t),
            0.5 * m.sin(0.1 * t) * m.cos(0.1 * t) + 0.1]


def moe_skill(situation, cs, cc, T_fk):
    """ur5_moe2016_example2.ipynb cells 4-8.  'singular': three 1-D wall SetConstraints on the tool position (hard, gain
    5e2, priorities y 7 < x 8 < z 9) in front of the soft tracking equality (gain 0.15, priority 10); 'multidim': ONE
    3-row SetConstraint for the box.  -> (SkillSpecification, t, q)"""
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    p = T_fk(q)[:3, 3]
    path = cs.vertcat(*moe_path(t, cs))
    (x_min, x_max), (y_min, y_max), (z_min, z_max) = MOE_WALLS
    track = cc.EqualityConstraint(label="move_point2", expression=p - path, priority=10, constraint_type="soft", gain=0.15)
    if situation == "singular":
        cons = [cc.SetConstraint(label="colav_x", expression=p[0], set_min=x_min, set_max=x_max, priority=8,
                                 constraint_type="hard", gain=5e2),
                cc.SetConstraint(label="colav_y", expression=p[1], set_min=y_min, set_max=y_max, priority=7,
                                 constraint_type="hard", gain=5e2),
                cc.SetConstraint(label="colav_z", expression=p[2], set_min=z_min, set_max=z_max, priority=9,
                                 constraint_type="hard", gain=5e2),
                track]
        label = "box_move"
    else:
        cons = [cc.SetConstraint(label="colav_box", expression=p, set_min=MOE_WALLS[:, 0].copy(),
                                 set_max=MOE_WALLS[:, 1].copy(), priority=7, constraint_type="hard", gain=5e2),
                track]
        label = "box_move_multidim"
    spec = cc.SkillSpecification(label=label, time_var=t, robot_var=q, robot_vel_var=dq, constraints=cons)
    return spec, t, q


def simulate_moe(solve, fk_position, n_ticks=MOE_TICKS):
    """cell 12: `solve(t, q)` -> (dq [6], mode | None); `fk_position(q)` -> tool position [3].  Returns t_sim, q_sim, p_sim,
    e_sim, mode_sim as the notebook fills them: e_sim[i + 1] is the tracking error at q_sim[i + 1] against the
    trajectory at t_sim[i]; mode_sim[i + 1] the mode of tick i."""
    t_sim = np.array([MOE_DT * i for i in range(n_ticks + 1)])
    q_sim, p_sim = np.zeros((n_ticks + 1, 6)), np.zeros((n_ticks + 1, 3))
    e_sim, mode_sim = np.zeros(n_ticks + 1), np.zeros(n_ticks + 1)
    q_sim[0] = MOE_HOME
    p_sim[0] = fk_position(MOE_HOME)
    e_sim[0] = np.linalg.norm(p_sim[0] - np.array(moe_path(t_sim[0])))
    for i in range(n_ticks):
        dq, mode = solve(t_sim[i], q_sim[i])
        dq = np.clip(dq, -MOE_MAX_SPEED, MOE_MAX_SPEED)
        q_sim[i + 1] = q_sim[i] + dq * MOE_DT
        p_sim[i + 1] = fk_position(q_sim[i + 1])
        e_sim[i + 1] = np.linalg.norm(p_sim[i + 1] - np.array(moe_path(t_sim[i])))
        if mode is not None:
            mode_sim[i + 1] = mode
    return t_sim, q_sim, p_sim, e_sim, mode_sim


def interval_deviation(key, t_sim, values, reach=1.5, within=None):
    """Interval pins (tests/golden/moe_figure_pins.py): per pixel column the stored figure allows the curve's centre one or
    more intervals of values.  For every column: how far (in pixel rows) the simulated curve - taken anywhere within
    `reach` pixels in t - stays away from the nearest of them.  Returns (worst deviation, columns, t of the worst)."""
    ft, flo, fhi = FIGS[key + "_t"], FIGS[key + "_lo"], FIGS[key + "_hi"]
    px_t, px_v = FIGS[key + "_pixel"]
    order = np.argsort(ft, kind="stable")
    ft, flo, fhi = ft[order], flo[order], fhi[order]
    starts = np.nonzero(np.diff(ft, prepend=-np.inf) > 0)[0]
    worst, where = 0.0, None
    n = 0
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(ft)
        if ft[s] + reach * px_t > t_sim[-1] or (within is not None and not within[0] <= ft[s] <= within[1]):
            continue                                           # (a shortened simulation: the columns it covers)
        n += 1
        i0, i1 = np.searchsorted(t_sim, [ft[s] - reach * px_t, ft[s] + reach * px_t])
        smin, smax = values[i0:max(i1, i0 + 1)].min(), values[i0:max(i1, i0 + 1)].max()
        miss = min(max(lo - smax, smin - hi, 0.0) for lo, hi in zip(flo[s:e], fhi[s:e])) / px_v
        if miss > worst:
            worst, where = miss, float(ft[s])
    return worst, n, where


def fill_deviation(key, t_sim, values, reach=1.5):
    """columns a chattering run FILLS between two levels in the stored figure (rows: t, low level, high level): the
    simulated run has to visit both levels within `reach` pixels of the column.  Returns (columns missed, columns)."""
    fill = FIGS[key + "_fill"]
    px_t = FIGS[key + "_pixel"][0]
    missed = 0
    for tk, lo, hi in fill:
        if tk + reach * px_t > t_sim[-1]:
            continue
        i0, i1 = np.searchsorted(t_sim, [tk - reach * px_t, tk + reach * px_t])
        seen = values[i0:max(i1, i0 + 1)]
        missed += not (seen.min() <= lo + 0.25 and seen.max() >= hi - 0.25)
    return missed, len(fill)


def moe_pins(case, t_sim, p_sim, e_sim, mode_sim, within=None):
    """every pin the stored figures hold for one of the four runs -> [(pin, worst deviation in pixels, columns, t)]"""
    kind, sit = case.split("_")
    own = kind == "pinv"                        # the pinv curves are visible in their own colour; the QP's lie under the others
    rows = []
    for k, axis in enumerate("xyz"):
        rows.append(("moe_%s_%s_union" % (axis, sit), p_sim[:, k]))
        if own:
            rows.append(("moe_%s_%s_pinv" % (axis, sit), p_sim[:, k]))
    if sit == "multidim":
        rows.append(("moe_y_multidim_inset_union", p_sim[:, 1]))
        if own:
            rows.append(("moe_y_multidim_inset_pinv", p_sim[:, 1]))
    for fig in ("moe_e_%s" % sit, "moe_e_%s_small" % sit, "moe_e_%s_small_inset" % sit):
        rows.append((fig + "_union", e_sim))
        if own:
            rows.append((fig + "_pinv", e_sim))
    if own:
        rows += [("moe_modes_multidim", mode_sim), ("moe_modes_full_multidim", mode_sim)] if sit == "multidim" else [
            ("moe_modes_separate", mode_sim)]
    return [(key,) + interval_deviation(key, t_sim, values, within=within) for key, values in rows]
