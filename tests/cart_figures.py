"""The six simulations of cart_on_track_1D_comparison_of_controllers.ipynb whose figures the notebook stores (cells 6-11,
18-22, 31-36, 50-54, 56-61, 75-78), rebuilt with the product's front-end, and the comparison of a simulated curve with
the samples digitised from those figures (tests/golden/make_figure_pins.py -> notebook_figures.npz)."""
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


def deviation_in_pixels(case, curve, t_sim, values):
    """For every sample digitised from the stored figure: how far (in pixel rows) the simulated curve misses it, where
    the simulated curve may be taken anywhere within a pixel and a half in t (a figure cannot place a jump more
    exactly).  Returns (worst deviation, samples)."""
    key = "%s_%s" % (case, curve)
    ft, fv = FIGS[key + "_t"], FIGS[key + "_v"]
    px_t, px_v = FIGS[key + "_pixel"]
    worst = 0.0
    for tk, vk in zip(ft, fv):
        near = (t_sim >= tk - 1.5 * px_t) & (t_sim <= tk + 1.5 * px_t)
        lo, hi = values[near].min(), values[near].max()
        miss = max(lo - vk, vk - hi, 0.0) / px_v
        worst = max(worst, miss)
    return worst, len(ft)
