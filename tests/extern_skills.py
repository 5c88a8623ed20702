"""Skills whose constraints lie outside the affine-in-features row table (they run as generated
device code, casclik_amd/codegen.py); shared by the CPU code-generation tests and the GPU parity tests."""
import casclik_amd as cc
from casclik_amd import sym as cs


def double_pendulum_skill(track=False):
    """double_pendulum_2D_comparison_of_controllers.ipynb cells 3-11 (and 31-33 with track=True)."""
    l_1, l_2 = 1.0, 0.75
    t, q, dq = cs.MX.sym("t"), cs.MX.sym("q", 2), cs.MX.sym("dq", 2)
    p_mid = cs.vertcat(l_1 * cs.cos(q[0]), l_1 * cs.sin(q[0]))
    p = cs.vertcat(l_1 * cs.cos(q[0]) + l_2 * cs.cos(q[0] + q[1]),
                   l_1 * cs.sin(q[0]) + l_2 * cs.sin(q[0] + q[1]))
    if track:
        p_des = 0.25 * cs.vertcat(cs.cos(0.5 * t), cs.sin(0.5 * t)) + cs.vertcat(1.0, 1.0)
    else:
        p_des = cs.vertcat(0.75, 0.5)
    table_height = -0.05
    cn = [cc.EqualityConstraint(label="min_dist_cnstr", expression=p_des - p, gain=1.0, constraint_type="soft"),
          cc.VelocitySetConstraint(label="speed_limit_cnstr", expression=q, set_min=-cs.vertcat(0.5, 0.5),
                                   set_max=cs.vertcat(0.5, 0.5)),
          cc.SetConstraint(label="table_midpoint_cnstr", expression=p_mid[1] - table_height, set_min=0.0,
                           set_max=cs.inf),
          cc.SetConstraint(label="table_endpoint_cnstr", expression=p[1] - table_height, set_min=0.0,
                           set_max=cs.inf)]
    return cc.SkillSpecification(label="move_to_point_skill", time_var=t, robot_var=q, robot_vel_var=dq,
                                 constraints=cn)



def mixed_frame_skill(fk):
    """7-DoF arm: products / functions of tool-frame entries, a virtual variable, input and time terms."""
    t, q, x, y = cs.MX.sym("t"), cs.MX.sym("q", 7), cs.MX.sym("x", 1), cs.MX.sym("y", 3)
    T = fk["T_fk"](q)
    p = T[:3, 3]
    c = cs.vertcat(0.4, 0.1, 0.5)
    cn = [cc.SetConstraint("sphere", cs.dot(p - c, p - c), set_min=0.25, set_max=1e10, priority=0),
          cc.EqualityConstraint("mix", cs.vertcat(T[0, 3] * T[1, 3] - y[0] * cs.sin(t),
                                                  T[2, 2] * q[1] + cs.exp(-x[0]) * y[1],
                                                  cs.sqrt(1.5 + T[0, 0] * T[1, 1]) - cs.cos(x[0] + 0.3 * t) / (2.0 + y[2] ** 2)),
                                gain=2.0, priority=1, constraint_type="soft"),
          cc.EqualityConstraint("rest", q - 0.1, gain=0.5, priority=2, constraint_type="soft")]
    return cc.SkillSpecification("mixed", t, q, virtual_var=x, input_var=y, constraints=cn)
