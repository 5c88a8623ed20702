#!/usr/bin/env python3
"""Three of the reference's notebooks as scripts, with the imports a user changes and nothing else.

    casclik                                   casclik_amd
    ----------------------------------------  ---------------------------------------------------------
    import casclik as cc                      import casclik_amd as cc
    import casadi as cs                       from casclik_amd import sym as cs
    from urdf2casadi import converter         from casclik_amd import converter

The loops below are the notebooks' own (`solve(t_sim[i], q_sim[i])[0].toarray()`, explicit Euler, one robot): they run
on the GPU through the reference-shaped single-instance API.  Each result is compared with the figure the notebook
stores of the reference's run (tests/golden/notebook_figures.npz, digitised by tests/golden/make_figure_pins.py) - the
printed number is the worst miss in pixel rows of that figure.  For fleets of robots use `solve_batch` / `rollout_batch`
(README.md).        python examples/notebook_loops.py            (needs an AMD GPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import casclik_amd as cc                      # noqa: E402
from casclik_amd import sym as cs             # noqa: E402
from casclik_amd import converter, skills     # noqa: E402
import figure_skills as figures               # noqa: E402  (the digitised figures and the pixel comparison)


def cart_on_track():
    """cart_on_track_1D_comparison_of_controllers.ipynb cells 31-36: track 0.4 sin(0.3 t) with the ReactiveQPController;
    the trajectory leaves the rail [0, 1], the SetConstraint holds the cart at its end"""
    t, p, dp = cs.MX.sym("t"), cs.MX.sym("p"), cs.MX.sym("dp")
    p_des = 0.4 * cs.sin(0.3 * t)
    max_speed, min_p, max_p = 0.275, 0.0, 1.0
    min_dist_cnstr = cc.EqualityConstraint(label="min_dist_cnstr", expression=p_des - p, gain=1.0,
                                           constraint_type="soft", priority=1)
    cart_limit_cnstr = cc.SetConstraint(label="cart_limit_cnstr", expression=p, gain=1.0, set_min=min_p, set_max=max_p)
    speed_limit_cnstr = cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=p, gain=10.0,
                                                 set_min=-max_speed, set_max=max_speed)
    trajectory_skill = cc.SkillSpecification(label="track_trajectory_skill", time_var=t, robot_var=p, robot_vel_var=dp,
                                             constraints=[min_dist_cnstr, cart_limit_cnstr, speed_limit_cnstr])
    trajectory_skill.print_constraints()
    reactiveQPcntrllr = cc.ReactiveQPController(skill_spec=trajectory_skill, robot_var_weights=[1.0])
    reactiveQPcntrllr.setup_problem_functions()
    reactiveQPcntrllr.setup_solver()
    dt = 0.02
    t_sim = np.array([dt * i for i in range(1200)])
    p_sim = np.zeros(len(t_sim))
    p_sim[0] = 0.0001
    dp_sim = np.zeros(len(t_sim))
    for i in range(len(t_sim) - 1):
        dp_sim[i] = reactiveQPcntrllr.solve(t_sim[i], p_sim[i])[0].toarray()[0, 0]
        p_sim[i + 1] = p_sim[i] + dp_sim[i] * dt
    print("cart on track, QP: position ends at %.4f m; against the notebook's stored figure: position %.2f px, speed %.2f px"
          % (p_sim[-1], figures.deviation_in_pixels("qp_traj", "p", t_sim, p_sim)[0],
             figures.deviation_in_pixels("qp_traj", "dp", t_sim, dp_sim)[0]))


def double_pendulum():
    """double_pendulum_2D_comparison_of_controllers.ipynb cells 3-19: the tool to (0.75, 0.5) above a table, joint speeds
    limited to 0.5 rad/s, ReactiveQPController"""
    l_1, l_2 = 1.0, 0.75
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 2), cs.MX.sym("dq", 2)
    p_mid = cs.vertcat(l_1 * cs.cos(q[0]), l_1 * cs.sin(q[0]))
    p = cs.vertcat(l_1 * cs.cos(q[0]) + l_2 * cs.cos(q[0] + q[1]), l_1 * cs.sin(q[0]) + l_2 * cs.sin(q[0] + q[1]))
    fp = cs.Function("fp", [q], [p], ["q"], ["p"])
    p_des = cs.vertcat(0.75, 0.5)
    max_speed, table_height = 0.5, -0.05
    min_dist_cnstr = cc.EqualityConstraint(label="min_dist_cnstr", expression=p_des - p, gain=1.0, constraint_type="soft")
    speed_limit_cnstr = cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=q,
                                                 set_min=-cs.vertcat(max_speed, max_speed),
                                                 set_max=cs.vertcat(max_speed, max_speed))
    table_midpoint_cnstr = cc.SetConstraint(label="table_midpoint_cnstr", expression=p_mid[1] - table_height,
                                            set_min=0.0, set_max=cs.inf)
    table_endpoint_cnstr = cc.SetConstraint(label="table_endpoint_cnstr", expression=p[1] - table_height,
                                            set_min=0.0, set_max=cs.inf)
    point_skill = cc.SkillSpecification(label="move_to_point_skill", time_var=t, robot_var=q, robot_vel_var=dq,
                                        constraints=[min_dist_cnstr, speed_limit_cnstr, table_midpoint_cnstr,
                                                     table_endpoint_cnstr])
    qpcntrllr = cc.ReactiveQPController(skill_spec=point_skill, robot_var_weights=[1.0, 1.0])
    qpcntrllr.setup_problem_functions()
    qpcntrllr.setup_solver()
    dt = 0.01
    t_sim = np.array([dt * i for i in range(800)])
    q_sim = np.zeros((len(t_sim), 2))
    q_sim[0, :] = [np.pi / 2 - 1e-5, 0.0]
    dq_sim = np.zeros((len(t_sim), 2))
    p_sim = np.zeros((len(t_sim), 2))
    for i in range(len(t_sim) - 1):
        dq_sim[i, :] = qpcntrllr.solve(t_sim[i], q_sim[i, :])[0].toarray()[:, 0]
        q_sim[i + 1, :] = q_sim[i, :] + dq_sim[i, :] * dt
        p_sim[i + 1, :] = fp(q_sim[i + 1, :]).toarray()[:, 0]
    print("double pendulum, QP: tool ends at (%.4f, %.4f); against the stored figures: joint speeds %.2f / %.2f px, tool "
          "%.2f / %.2f px" % (p_sim[-1, 0], p_sim[-1, 1],
                              figures.deviation_in_pixels("pend_point_dq", "dq0", t_sim, dq_sim[:, 0])[0],
                              figures.deviation_in_pixels("pend_point_dq", "dq1", t_sim, dq_sim[:, 1])[0],
                              figures.deviation_in_pixels("pend_point_p", "px", t_sim, p_sim[:, 0])[0],
                              figures.deviation_in_pixels("pend_point_p", "py", t_sim, p_sim[:, 1])[0]))


def ur5_to_a_point():
    """ur5_transformation_matrix_comparison_of_controllers.ipynb cells 2-9, 27-32: the UR5's tool to (0.5, 0.5, 0.5) with
    the PseudoInverseController, one 1-D SetConstraint per joint limit, speeds saturated at pi / 5"""
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 6), cs.MX.sym("dq", 6)
    max_speed = np.pi / 5
    UR5_home = [0.0, -np.pi / 2, 0.0, -np.pi / 2, 0.0, 0.0]
    fk_dict = converter.from_file(root="base_link", tip="tool0", filename=skills.UR5_URDF)
    T_fk = fk_dict["T_fk"]
    print("Distance to UR5Home pos: " + str(cs.norm_2(T_fk(UR5_home)[:3, 3])))
    q_max, q_min = np.array(fk_dict["upper"]), np.array(fk_dict["lower"])
    p_des = np.array([0.5, 0.5, 0.5])
    p_fk = T_fk(q)[:3, 3]
    min_dist_cnstr = cc.EqualityConstraint(label="Minimize_point_error", expression=cs.norm_2(p_des - p_fk), gain=50.,
                                           constraint_type="soft", priority=6)
    joint_limits_cnstr_list = [cc.SetConstraint(label="limit_q_" + str(i), expression=q[i], set_min=q_min[i],
                                                set_max=q_max[i], priority=i) for i in range(6)]
    point_skill_pinv = cc.SkillSpecification(label="point_skill_pinv", time_var=t, robot_var=q, robot_vel_var=dq,
                                             constraints=[min_dist_cnstr] + joint_limits_cnstr_list)
    pinvcntrllr = cc.PseudoInverseController(skill_spec=point_skill_pinv)
    pinvcntrllr.setup_problem_functions()
    pinvcntrllr.setup_solver()
    dt = 0.01
    t_sim = np.array([dt * i for i in range(1000)])
    q_sim = np.zeros((len(t_sim), 6))
    q_sim[0, :] = UR5_home
    p_sim = np.zeros((len(t_sim), 3))
    p_sim[0, :] = T_fk(UR5_home)[:3, 3].toarray()[:, 0]
    for i in range(len(t_sim) - 1):
        dq_i = pinvcntrllr.solve(t_sim[i], q_sim[i, :])[0].toarray()[:, 0]
        dq_i = np.clip(dq_i, -max_speed, max_speed)
        q_sim[i + 1, :] = q_sim[i, :] + dq_i * dt
        p_sim[i + 1, :] = T_fk(q_sim[i + 1, :])[:3, 3].toarray()[:, 0]
    print("UR5 to a point, pinv: tool ends at %s; against the stored figure: x %.2f, y %.2f, z %.2f px"
          % (np.round(p_sim[-1], 4).tolist(), figures.deviation_in_pixels("ur5_pinv_p", "x", t_sim, p_sim[:, 0])[0],
             figures.deviation_in_pixels("ur5_pinv_p", "y", t_sim, p_sim[:, 1])[0],
             figures.deviation_in_pixels("ur5_pinv_p", "z", t_sim, p_sim[:, 2])[0]))


if __name__ == "__main__":
    cart_on_track()
    double_pendulum()
    ur5_to_a_point()
